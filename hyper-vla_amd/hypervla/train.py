"""Host side of the fine-tune step (reference: scripts/train.py:405-542 ``train_step_pmap``,
octo/utils/train_utils.py:295-443 ``create_optimizer``), frozen-image-encoder variant.

    ft = FineTuner(model, batch=32)
    loss = ft.step(instruction_dict, initial_state, images, batch)       # fwd + bwd + all-reduce + AdamW + EMA

All arithmetic is in libhvla (csrc/train.hip); torch owns the device buffers and, when a process group is
initialised, all-reduces the flat gradient (RCCL over xGMI on the GPU box — the `pmean` of train.py:460).
"""
from __future__ import annotations

from typing import Dict, List, Tuple

import numpy as np

from . import _native
from .config import Geometry, generated_leaves


def train_param_layout(g: Geometry) -> Tuple[List[Tuple[str, int, Tuple[int, ...]]], int]:
    """[(name, offset, shape)] of the flat trainable-parameter vector == make_train_layout() in csrc/train.hip.
    The 73 output heads are the fused entries "W_cat" [C, G] and "b_cat" [G] (columns in pytree leaf order)."""
    C, F = g.ctx_dim, g.ctx_mlp
    out, off = [], 0

    def add(name, shape):
        nonlocal off
        out.append((name, off, tuple(shape)))
        off += int(np.prod(shape))

    add("task_token_projection/kernel", (g.lang_dim, C)); add("task_token_projection/bias", (C,))
    add("initial_image_projection/kernel", (g.enc_dim, C)); add("initial_image_projection/bias", (C,))
    add("task_pos_embedding", (1, g.lang_tokens, C)); add("initial_image_pos_embedding", (1, 1, C))
    add("layer_pos_embedding", (1, 1, C))
    for l in range(g.ctx_layers):
        b = f"Transformer_0/encoderblock_{l}/"
        a = b + "MultiHeadDotProductAttention_0/"
        hc = C // g.ctx_heads
        add(b + "LayerNorm_0/scale", (C,)); add(b + "LayerNorm_0/bias", (C,))
        add(b + "LayerNorm_1/scale", (C,)); add(b + "LayerNorm_1/bias", (C,))
        for nm in ("query", "key", "value"):
            add(a + nm + "/kernel", (C, g.ctx_heads, hc)); add(a + nm + "/bias", (g.ctx_heads, hc))
        add(a + "out/kernel", (g.ctx_heads, hc, C)); add(a + "out/bias", (C,))
        add(b + "MlpBlock_0/Dense_0/kernel", (C, F)); add(b + "MlpBlock_0/Dense_0/bias", (F,))
        add(b + "MlpBlock_0/Dense_1/kernel", (F, C)); add(b + "MlpBlock_0/Dense_1/bias", (C,))
    add("Transformer_0/encoder_norm/scale", (C,)); add("Transformer_0/encoder_norm/bias", (C,))
    G = generated_leaves(g)[-1].offset + generated_leaves(g)[-1].size
    add("W_cat", (C, G)); add("b_cat", (G,))
    return out, off


def pack_params(g: Geometry, params: Dict[str, np.ndarray]) -> np.ndarray:
    layout, total = train_param_layout(g)
    flat = np.zeros(total, np.float32)
    leaves = generated_leaves(g)
    for name, off, shape in layout:
        n = int(np.prod(shape))
        if name == "W_cat":
            w = np.concatenate([np.asarray(params[l.head_name + "/kernel"], np.float32) for l in leaves], axis=1)
            flat[off:off + n] = w.reshape(-1)
        elif name == "b_cat":
            flat[off:off + n] = np.concatenate([np.asarray(params[l.head_name + "/bias"], np.float32).reshape(-1) for l in leaves])
        else:
            flat[off:off + n] = np.asarray(params[name], np.float32).reshape(-1)
    return flat


def unpack_params(g: Geometry, flat: np.ndarray) -> Dict[str, np.ndarray]:
    """flat vector (parameters or gradients) -> reference-named tensors."""
    layout, _ = train_param_layout(g)
    leaves = generated_leaves(g)
    out: Dict[str, np.ndarray] = {}
    for name, off, shape in layout:
        v = np.asarray(flat[off:off + int(np.prod(shape))]).reshape(shape)
        if name == "W_cat":
            for l in leaves:
                out[l.head_name + "/kernel"] = v[:, l.offset:l.offset + l.size].copy()
        elif name == "b_cat":
            for l in leaves:
                out[l.head_name + "/bias"] = v[l.offset:l.offset + l.size].copy()
        else:
            out[name] = v.copy()
    return out


def lr_rsqrt(step: int, peak: float, warmup: int = 2000, timescale: int = 10000, init: float = 0.0) -> float:
    """octo/utils/train_utils.py:212-225 ("rsqrt": linear warm-up joined to peak / sqrt((s + ts) / ts))."""
    if step < warmup:
        return init + (peak - init) * step / warmup
    s = step - warmup
    return peak / float(np.sqrt((s + timescale) / timescale))


class FineTuner:
    def __init__(self, model, batch: int, peak_lr: float = 3e-4, weight_decay: float = 0.05, clip: float = 1.0,
                 ema_decay: float = 0.999, b1: float = 0.9, b2: float = 0.999, eps: float = 1e-8):
        import torch
        self.torch, self.model, self.g, self.B = torch, model, model.geometry, batch
        dev = model.device
        n, G, work = model._ctx.train_sizes(batch)
        layout, total = train_param_layout(self.g)
        assert n == total, (n, total)
        self.n, self.G = n, G
        f32 = dict(dtype=torch.float32, device=dev)
        self.params = torch.as_tensor(pack_params(self.g, model.params)).to(dev)
        self.grads = torch.zeros(n, **f32)
        self.mu = torch.zeros(n, dtype=torch.bfloat16, device=dev)
        self.nu = torch.zeros(n, **f32)
        self.ema = self.params.clone()
        self.theta = torch.empty(batch, G, **f32)
        self.dtheta = torch.empty(batch, G, **f32)
        self.work = torch.empty(work, **f32)
        self.loss = torch.zeros(batch, **f32)
        self.actions = torch.zeros(batch, self.g.horizon, self.g.action_dim, **f32)
        self.logits = torch.zeros(batch, self.g.horizon, **f32)
        self.sqsum = torch.zeros(1, **f32)
        mask = np.zeros(G, np.uint8)
        for l in generated_leaves(self.g):                     # weight_decay_strategy v5 (train_utils.py:354-363)
            if "kernel" in l.flat_name:
                mask[l.offset:l.offset + l.size] = 1
        self.wd_mask = torch.as_tensor(mask).to(dev)
        self.buf = _native.hvla_train_buffers(*[t.data_ptr() for t in (
            self.params, self.grads, self.mu, self.nu, self.ema, self.theta, self.dtheta, self.work, self.loss,
            self.actions, self.logits, self.sqsum, self.wd_mask)])
        self.hy = dict(b1=b1, b2=b2, eps=eps, weight_decay=weight_decay, clip=clip, ema_decay=ema_decay)
        self.peak_lr, self.step_count = peak_lr, 0

    def _hyper(self, lr, forward_only=False):
        h = self.hy
        return _native.hvla_train_hyper(lr, h["b1"], h["b2"], h["eps"], h["weight_decay"], h["clip"], h["ema_decay"],
                                        self.step_count, int(forward_only))

    def forward_backward(self, instruction_dict, initial_state, tokens, batch, forward_only=False):
        """loss [B] (device) after writing self.grads = d mean(loss) / d params."""
        torch, m, g = self.torch, self.model, self.g
        li = instruction_dict["language_instruction"]
        tok = m._dev(li["token_embedding"], torch.float32)
        msk = m._dev(li["attention_mask"], torch.int64)
        cls = m._dev(np.asarray(initial_state["patch_embeddings"])[:, 0], torch.float32)
        tkn = m._dev(tokens, torch.float32)
        tgt = m._dev(np.asarray(batch["action"])[:, 0], torch.float32)
        am = m._dev(np.asarray(batch["action_pad_mask"])[:, 0].astype(np.uint8), torch.uint8)
        tm = m._dev(np.asarray(batch["timestep_pad_mask"])[:, 0].astype(np.uint8), torch.uint8)
        assert tok.shape[0] == self.B and tuple(tkn.shape) == (self.B, g.patches, g.enc_dim)
        self._keep = (tok, msk, cls, tkn, tgt, am, tm)
        ptrs = [t.data_ptr() for t in (tok, msk, cls, tkn, tgt, tm, am)]
        m._ctx.train_step(self.buf, ptrs, self.B, self._hyper(0.0, forward_only), m._stream())
        return self.loss

    def apply(self, lr=None):
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            dist.all_reduce(self.grads)                               # RCCL; pmean(grads) of scripts/train.py:460
            self.grads /= dist.get_world_size()
        lr = lr_rsqrt(self.step_count, self.peak_lr) if lr is None else lr
        self.model._ctx.train_apply(self.buf, self._hyper(lr), self.model._stream())
        self.step_count += 1

    def step(self, instruction_dict, initial_state, images, batch, lr=None):
        tokens = self.model.encode_images(images)                     # frozen encoder
        loss = self.forward_backward(instruction_dict, initial_state, tokens, batch)
        self.apply(lr)
        return loss.mean()
