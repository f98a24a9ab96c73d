"""Host side of the fine-tune step (reference: scripts/train.py:405-542 ``train_step_pmap``,
octo/utils/train_utils.py:295-443 ``create_optimizer``).

    ft = FineTuner(model, batch=32)                         # image encoder frozen (the config default)
    ft = FineTuner(model, batch=32, train_encoder=True)     # README.md:55 fine_tune_pretrained_image_encoder=True
    loss = ft.step(instruction_dict, initial_state, images, batch)       # fwd + bwd + all-reduce + AdamW + EMA

All arithmetic is in libhvla (csrc/train.hip); torch owns the device buffers and, when a process group is
initialised, all-reduces the flat gradient (RCCL over xGMI on the GPU box — the `pmean` of train.py:460).
"""
from __future__ import annotations

from typing import Dict, List, Tuple

import numpy as np

from . import _native
from .config import Geometry, encoder_leaves, generated_leaves, shared_name


def train_param_layout(g: Geometry, train_encoder: bool = False) -> Tuple[List[Tuple[str, int, Tuple[int, ...]]], int]:
    """[(name, offset, shape)] of the flat trainable-parameter vector == make_train_layout() in csrc/train.hip.
    The 73 output heads are the fused entries "W_cat" [C, G] and "b_cat" [G] (columns in pytree leaf order); with
    `train_encoder` the shared DINOv2 leaves follow as flat vectors under their checkpoint names."""
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
    if train_encoder:
        for path, shape in encoder_leaves(g):
            add(shared_name(path), (int(np.prod(shape)),))
    return out, off


def gradient_buckets(g: Geometry, train_encoder: bool = False) -> List[Tuple[str, int, int]]:
    """[(name, offset, length)] of the flat gradient in the order hvla_train_step finishes them (include/hvla.h,
    hvla_train_bucket_ranges): the shared DINOv2 leaves after the image encoder's backward, the output heads (W_cat,
    b_cat) after the weight-generation backward, the context encoder at the end of the step.  Contiguous, disjoint, and
    together the whole vector -- what `pmean(grads)` (scripts/train.py:460) is cut into so that each all-reduce runs
    under the rest of the backward pass."""
    layout, total = train_param_layout(g, train_encoder)
    at = {name: off for name, off, _ in layout}
    wcat = at["W_cat"]
    n_hyper = at["b_cat"] + generated_leaves(g)[-1].offset + generated_leaves(g)[-1].size
    out = []
    if train_encoder:
        out.append(("image_encoder", n_hyper, total - n_hyper))
    out.append(("output_heads", wcat, n_hyper - wcat))
    out.append(("context_encoder", 0, wcat))
    return out


def pack_params(g: Geometry, params: Dict[str, np.ndarray], train_encoder: bool = False) -> np.ndarray:
    layout, total = train_param_layout(g, train_encoder)
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


def unpack_params(g: Geometry, flat: np.ndarray, train_encoder: bool = False) -> Dict[str, np.ndarray]:
    """flat vector (parameters or gradients) -> reference-named tensors."""
    layout, _ = train_param_layout(g, train_encoder)
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


def weight_decay_mask(g: Geometry, strategy: str, train_encoder: bool = False) -> np.ndarray:
    """uint8 [n_params]: where `create_optimizer`'s decoupled weight decay applies (octo/utils/train_utils.py:325-382).
    The reference tests `jax.tree_util.keystr(path)` of every leaf of the hypernetwork's parameter tree; an output head is
    the module `output_head_<flat base-net leaf name>` (hypervla/model.py:342) with leaves `kernel` and `bias`, so the
    NAME of the generated leaf is part of the path of both.

    "v1" (the config default, hypervla_pretrain_config.py:290; train_utils.py:378-382): `"kernel" in keystr(path)` -- every
    Dense kernel of the hypernetwork, every output head's kernel, the BIAS of every head that generates a base-net
    *kernel* leaf (its path contains `..._kernel`), and the encoder leaves named *kernel*.
    "v2" (:326-330): everything except leaves with "norm" in the path that are not output heads.
    "v3" (:335-350): output heads that generate *kernel* leaves (kernel and bias), every leaf of the shared image
    encoder, and the remaining *kernel* leaves (context encoder, projections).
    "v5" (the README run, README.md:29; :354-363): as v3 without the context-encoder kernels.
    "v4" adds a second backward pass through a weight-decay loss (scripts/train.py:473-480) and is not built."""
    if strategy not in ("v1", "v2", "v3", "v5"):
        raise ValueError(f"weight_decay_strategy {strategy!r}: 'v1', 'v2', 'v3' and 'v5' are built (v4 is not)")
    layout, total = train_param_layout(g, train_encoder)
    leaves = generated_leaves(g)
    G = leaves[-1].offset + leaves[-1].size
    mask = np.zeros(total, np.uint8)
    cols = np.zeros(G, np.uint8)
    for l in leaves:
        if "kernel" in l.flat_name:
            cols[l.offset:l.offset + l.size] = 1
    for name, off, shape in layout:
        n = int(np.prod(shape))
        if name == "W_cat":                       # [C, G]: the kernels of the 73 output heads
            mask[off:off + n] = 1 if strategy in ("v1", "v2") else np.tile(cols, shape[0])
        elif name == "b_cat":                     # their biases: `output_head_<..._kernel>/bias` contains "kernel"
            mask[off:off + n] = 1 if strategy == "v2" else cols
        elif name.startswith("encoder_image_encoder_"):
            if strategy == "v1":
                mask[off:off + n] = 1 if "kernel" in name else 0
            elif strategy == "v2":
                mask[off:off + n] = 0 if "norm" in name.lower() else 1
            else:
                mask[off:off + n] = 1
        elif strategy == "v2":
            mask[off:off + n] = 0 if "norm" in name.lower() else 1
        elif strategy in ("v1", "v3") and "kernel" in name:
            mask[off:off + n] = 1
    return mask


class FineTuner:
    """One optimizer state of the fine-tune step.  Defaults are the README run's (README.md:29-31,61 and
    scripts/configs/hypervla_pretrain_config.py:286-321): weight_decay_strategy v5, learning rate 3e-4 (rsqrt) for the
    hypernetwork and 3e-5 for the shared image encoder, base_weight_decay 0, no gradient accumulation, EMA 0.999 started at
    update 5000 (`ema_start_step`; the EMA is a copy of the parameters at that update and an average afterwards,
    scripts/train.py:681-690)."""

    def __init__(self, model, batch: int, peak_lr: float = 3e-4, weight_decay: float = 0.05, clip: float = 1.0,
                 ema_decay: float = 0.999, b1: float = 0.9, b2: float = 0.999, eps: float = 1e-8,
                 train_encoder: bool = False, base_lr: float = 3e-5, base_weight_decay: float = 0.0,
                 weight_decay_strategy: str = "v5", ema_start_step: int = 5000, grad_accumulation_steps: int = 1,
                 accept_baked_position_table: bool = False):
        import torch
        self.torch, self.model, self.g, self.B = torch, model, model.geometry, batch
        self.train_encoder = bool(train_encoder)
        baked = (model.config or {}).get("position_embeddings_baked_from")
        if self.train_encoder and baked and not accept_baked_position_table:
            raise ValueError(
                f"this checkpoint's DINOv2 position table was baked from {baked} to the run-time grid at conversion time "
                "(hypervla/convert.py); the reference trains the ORIGINAL table through interpolate_pos_encoding, so its "
                "gradient and Adam state differ and the result cannot be exported back into a reference-shaped "
                "checkpoint.  Pass accept_baked_position_table=True to train the baked table anyway (INTEGRATION.md).")
        dev = model.device
        n, G, work, n_hyper = model._ctx.train_sizes(batch, self.train_encoder)
        layout, total = train_param_layout(self.g, self.train_encoder)
        assert n == total, (n, total)
        self.n, self.G, self.n_hyper = n, G, n_hyper
        f32 = dict(dtype=torch.float32, device=dev)
        self.params = torch.as_tensor(pack_params(self.g, model.params, self.train_encoder)).to(dev)
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
        self.weight_decay_strategy = weight_decay_strategy
        self.wd_mask = torch.as_tensor(weight_decay_mask(self.g, weight_decay_strategy, self.train_encoder)).to(dev)
        # the pull towards the pretrained encoder only exists for base_weight_decay > 0 (scripts/train.py:469)
        self.params0 = self.params[n_hyper:].clone() if self.train_encoder and base_weight_decay > 0 else None
        self.accum_k = int(grad_accumulation_steps)
        if self.accum_k < 1:
            raise ValueError("grad_accumulation_steps >= 1")
        self.acc = torch.zeros(n, **f32) if self.accum_k > 1 else None
        self.micro = 0                                  # micro-batches since the last update
        self.buf = self._buffers(self.grads)
        self.buf_acc = self._buffers(self.acc) if self.acc is not None else None
        self.hy = dict(b1=b1, b2=b2, eps=eps, weight_decay=weight_decay, clip=clip, ema_decay=ema_decay,
                       base_weight_decay=base_weight_decay)
        self.peak_lr, self.base_peak_lr, self.step_count = peak_lr, base_lr, 0
        self.ema_start_step = int(ema_start_step)
        self._bucket_setup()

    def _bucket_setup(self):
        self.buckets = gradient_buckets(self.g, self.train_encoder)
        ranges = self.model._ctx.train_bucket_ranges(self.train_encoder)
        self._bucket_id = [ranges.index((off, n)) for _, off, n in self.buckets]   # the library's numbering (0 encoder, 1 heads, 2 context)
        self._comm = None

    def _buffers(self, grads):
        return _native.hvla_train_buffers(*[t.data_ptr() if t is not None else None for t in (
            self.params, grads, self.mu, self.nu, self.ema, self.theta, self.dtheta, self.work, self.loss,
            self.actions, self.logits, self.sqsum, self.wd_mask, self.params0)])

    def _hyper(self, lr, forward_only=False, base_lr=None, clip=None, ema=True):
        h = self.hy
        return _native.hvla_train_hyper(lr, h["b1"], h["b2"], h["eps"], h["weight_decay"], h["clip"] if clip is None else clip,
                                        h["ema_decay"] if ema else 0.0,
                                        self.step_count, int(forward_only), lr if base_lr is None else base_lr,
                                        h["base_weight_decay"], int(self.train_encoder))

    def forward_backward(self, instruction_dict, initial_state, tokens_or_images, batch, forward_only=False):
        """loss [B] (device) after writing self.grads = d mean(loss) / d params.  The fourth argument is the frozen
        encoder's patch tokens f32 [B, P, E], or -- with train_encoder -- the uint8 observations [B, (1,) H, W, 3]."""
        torch, m, g = self.torch, self.model, self.g
        li = instruction_dict["language_instruction"]
        if "token_embedding" not in li:                    # frozen T5 inside the step, as scripts/train.py:407-415
            li = m.encode_instructions(li)
        tok = m._dev(li["token_embedding"], torch.float32)
        msk = m._dev(li["attention_mask"], torch.int64)
        cls = m._dev(np.asarray(initial_state["patch_embeddings"])[:, 0], torch.float32)
        if self.train_encoder:
            img = tokens_or_images
            if not torch.is_tensor(img):
                img = np.asarray(img)
            if img.ndim == 5:
                img = img[:, 0]
            obs = m._dev(img, torch.uint8)
            assert tuple(obs.shape) == (self.B, g.image_size, g.image_size, 3), obs.shape
            tkn_ptr, img_ptr = None, obs.data_ptr()
        else:
            obs = m._dev(tokens_or_images, torch.float32)
            assert tuple(obs.shape) == (self.B, g.patches, g.enc_dim), obs.shape
            tkn_ptr, img_ptr = obs.data_ptr(), None
        tgt = m._dev(np.asarray(batch["action"])[:, 0], torch.float32)
        am = m._dev(np.asarray(batch["action_pad_mask"])[:, 0].astype(np.uint8), torch.uint8)
        tm = m._dev(np.asarray(batch["timestep_pad_mask"])[:, 0].astype(np.uint8), torch.uint8)
        assert tok.shape[0] == self.B
        self._keep = (tok, msk, cls, obs, tgt, am, tm)
        ptrs = [tok.data_ptr(), msk.data_ptr(), cls.data_ptr(), tkn_ptr, img_ptr, tgt.data_ptr(), tm.data_ptr(), am.data_ptr()]
        m._ctx.train_step(self.buf, ptrs, self.B, self._hyper(0.0, forward_only), m._stream())
        return self.loss

    def all_reduce_gradient(self, single_rank_too: bool = False):
        """`pmean(grads)` of scripts/train.py:460, bucketed: RCCL over xGMI, one all-reduce per gradient bucket, each
        enqueued on a communication stream that waits (on the device, hvla_train_wait_bucket) only for the event
        hvla_train_step recorded when that bucket became final -- the 343 MB DINOv2 bucket is on the links while the
        weight-generation and context-encoder backward still run.  Nothing here blocks the host; the compute stream
        waits for the reductions before the optimizer reads `grads`.  (`single_rank_too`: run the same path in a
        one-rank process group -- the tests' way of exercising the events and streams on a one-GPU box.)"""
        import torch.distributed as dist
        if not (dist.is_available() and dist.is_initialized() and (dist.get_world_size() > 1 or single_rank_too)):
            return
        torch, m = self.torch, self.model
        world = dist.get_world_size()
        if self._comm is None:
            self._comm = torch.cuda.Stream(device=m.device)
        cur = torch.cuda.current_stream(m.device)
        works = []
        with torch.cuda.stream(self._comm):
            for i, (_, off, n) in enumerate(self.buckets):
                m._ctx.train_wait_bucket(self._bucket_id[i], self._comm.cuda_stream)
                works.append(dist.all_reduce(self.grads[off:off + n], async_op=True))
            for w in works:
                w.wait()                                  # the communication stream waits for the collective
            self.grads /= world                           # ... and scales behind it
        cur.wait_stream(self._comm)

    def apply(self, lr=None, base_lr=None):
        """Gradient all-reduce, then the optimizer: chain(clip_by_global_norm, [MultiSteps](adamw)) and the EMA.  With
        grad_accumulation_steps = k the clipped gradient of every micro-batch goes into a running mean and parameters move
        on every k-th call only (optax.MultiSteps, octo/utils/train_utils.py:420-421); returns True when they moved."""
        self.all_reduce_gradient()
        ctx, st = self.model._ctx, self.model._stream()
        if self.accum_k > 1:
            if self.micro == 0:
                self.acc.zero_()
            ctx.train_accumulate(self.buf, self.acc.data_ptr(), 1.0 / self.accum_k, self._hyper(0.0), st)
            self.micro += 1
            if self.micro < self.accum_k:
                return False
            self.micro = 0
        lr = lr_rsqrt(self.step_count, self.peak_lr) if lr is None else lr
        base_lr = lr_rsqrt(self.step_count, self.base_peak_lr) if base_lr is None else base_lr
        update = self.step_count + 1                   # scripts/train.py:679 current_update_step
        ema_on = update > self.ema_start_step
        if self.accum_k > 1:                           # the micro-gradients were clipped one by one
            ctx.train_apply(self.buf_acc, self._hyper(lr, base_lr=base_lr, clip=float("inf"), ema=ema_on), st)
        else:
            ctx.train_apply(self.buf, self._hyper(lr, base_lr=base_lr, ema=ema_on), st)
        if update == self.ema_start_step:              # scripts/train.py:682-688: the EMA starts as a copy
            self.ema.copy_(self.params)
        self.step_count += 1
        return True

    def step(self, instruction_dict, initial_state, images, batch, lr=None, base_lr=None):
        obs = images if self.train_encoder else self.model.encode_images(images)
        loss = self.forward_backward(instruction_dict, initial_state, obs, batch)
        self.apply(lr, base_lr)
        return loss.mean()
