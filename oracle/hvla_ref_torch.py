"""ORACLE (second, independent restatement) — torch-CPU graph of the same path.

TEST INFRASTRUCTURE ONLY (see oracle/hvla_ref_np.py for the rules).  Written separately from the
numpy oracle and on different primitives so that agreement between the two is evidence:

* the image encoder is HF transformers' own torch ``Dinov2Model`` (loaded with the same synthetic
  weights, ``image_size`` = run-time size so no position-embedding interpolation happens),
* attention goes through ``F.scaled_dot_product_attention`` with a boolean keep-mask,
* LayerNorm through ``F.layer_norm`` (two-pass variance), GELU through ``F.gelu``.

It is also the "CPU restatement (not JAX)" timed by ``bench.py``'s ``cpu_baseline`` leg
(BASELINE.md §3): float32, ``torch.set_num_threads(cores)``.

``quant`` lets tests emulate the GPU path's rounding (bf16 MFMA operands, f32 accumulate) to
predict its error budget on the CPU.
"""
from __future__ import annotations

import math
from typing import Callable, Dict, Optional

import numpy as np
import torch
import torch.nn.functional as Fn


def _t(a, dtype):
    return torch.as_tensor(np.asarray(a)).to(dtype)


def bf16_round(x: torch.Tensor) -> torch.Tensor:
    return x.to(torch.bfloat16).to(x.dtype)


def _mm(x, w, quant):
    if quant is not None:
        return quant(x) @ quant(w)
    return x @ w


# ---------------------------------------------------------------- hypernetwork
class HyperNetRef:
    """HyperNetwork.__call__ (hypervla/components/hypernetwork.py:99-233)."""

    def __init__(self, params: Dict[str, np.ndarray], g, leaves, dtype=torch.float32):
        self.g, self.leaves, self.dt = g, leaves, dtype
        self.p = {k: _t(v, dtype) for k, v in params.items() if not k.startswith("encoder_image_encoder_")}
        # all 73 heads fused once: W_cat [C, G], b_cat [G]  (share_layer_index => same ctx token)
        self.w_cat = torch.cat([self.p[l.head_name + "/kernel"] for l in leaves], dim=1)
        self.b_cat = torch.cat([self.p[l.head_name + "/bias"] for l in leaves], dim=0)

    def _block(self, x, keep, l, heads, quant):
        p, b = self.p, f"Transformer_0/encoderblock_{l}/"
        C = x.shape[-1]
        hd = C // heads
        h = Fn.layer_norm(x, (C,), p[b + "LayerNorm_0/scale"], p[b + "LayerNorm_0/bias"], eps=1e-6)
        a = b + "MultiHeadDotProductAttention_0/"
        B, S, _ = h.shape
        q = _mm(h, p[a + "query/kernel"].reshape(C, C), quant) + p[a + "query/bias"].reshape(C)
        k = _mm(h, p[a + "key/kernel"].reshape(C, C), quant) + p[a + "key/bias"].reshape(C)
        v = _mm(h, p[a + "value/kernel"].reshape(C, C), quant) + p[a + "value/bias"].reshape(C)
        q, k, v = (t.reshape(B, S, heads, hd).transpose(1, 2) for t in (q, k, v))
        o = Fn.scaled_dot_product_attention(q, k, v, attn_mask=keep)      # scale = 1/sqrt(hd)
        o = o.transpose(1, 2).reshape(B, S, C)
        x = x + _mm(o, p[a + "out/kernel"].reshape(C, C), quant) + p[a + "out/bias"]
        y = Fn.layer_norm(x, (C,), p[b + "LayerNorm_1/scale"], p[b + "LayerNorm_1/bias"], eps=1e-6)
        y = Fn.gelu(_mm(y, p[b + "MlpBlock_0/Dense_0/kernel"], quant) + p[b + "MlpBlock_0/Dense_0/bias"],
                    approximate="tanh")
        return x + _mm(y, p[b + "MlpBlock_0/Dense_1/kernel"], quant) + p[b + "MlpBlock_0/Dense_1/bias"]

    def context(self, token_embedding, attention_mask, init_cls, quant=None):
        g, p = self.g, self.p
        tok = _t(token_embedding, self.dt)
        B, T, _ = tok.shape
        x_lang = _mm(tok, p["task_token_projection/kernel"], quant) + p["task_token_projection/bias"] + p["task_pos_embedding"]
        x_img = _mm(_t(init_cls, self.dt).reshape(B, 1, -1), p["initial_image_projection/kernel"], quant) \
            + p["initial_image_projection/bias"] + p["initial_image_pos_embedding"]
        x_layer = p["layer_pos_embedding"].expand(B, 1, -1)
        x = torch.cat([x_lang, x_img, x_layer], 1)
        S = T + 2
        keep = torch.zeros(B, 1, S, S, dtype=torch.bool)
        keep[:, 0, :, :T] = torch.as_tensor(np.asarray(attention_mask)).bool()[:, None, :]
        keep[:, 0, :, T] = True
        keep[:, 0, S - 1, S - 1] = True
        for l in range(g.ctx_layers):
            x = self._block(x, keep, l, g.ctx_heads, quant)
        x = Fn.layer_norm(x, (g.ctx_dim,), p["Transformer_0/encoder_norm/scale"], p["Transformer_0/encoder_norm/bias"], eps=1e-6)
        ctx = x[:, -1]
        return ctx / math.sqrt(g.ctx_dim) if g.scale_context else ctx

    def generate(self, ctx, quant=None):
        """flat generated parameters [B, G] in reference leaf order."""
        return _mm(ctx, self.w_cat, quant) + self.b_cat


# ---------------------------------------------------------------- DINOv2 (HF torch model)
def build_hf_dinov2(params: Dict[str, np.ndarray], g, enc_shapes, dtype=torch.float32):
    from transformers import Dinov2Config, Dinov2Model
    cfg = Dinov2Config(hidden_size=g.enc_dim, num_hidden_layers=g.enc_layers,
                       num_attention_heads=g.enc_heads, mlp_ratio=g.enc_mlp // g.enc_dim,
                       image_size=g.image_size, patch_size=g.patch, layer_norm_eps=1e-6,
                       hidden_act="gelu", layerscale_value=1.0, qkv_bias=True,
                       use_swiglu_ffn=False, attn_implementation="eager")
    m = Dinov2Model(cfg).eval()

    def get(path):
        return torch.as_tensor(params["encoder_image_encoder_" + "_".join(path)]).reshape(enc_shapes[path]).float()

    sd = {}
    sd["embeddings.cls_token"] = get(("embeddings", "cls_token"))
    sd["embeddings.mask_token"] = get(("embeddings", "mask_token"))
    sd["embeddings.position_embeddings"] = get(("embeddings", "position_embeddings"))
    # flax conv kernel HWIO -> torch OIHW
    sd["embeddings.patch_embeddings.projection.weight"] = get(("embeddings", "patch_embeddings", "projection", "kernel")).permute(3, 2, 0, 1).contiguous()
    sd["embeddings.patch_embeddings.projection.bias"] = get(("embeddings", "patch_embeddings", "projection", "bias"))
    for i in range(g.enc_layers):
        L, T = ("encoder", "layer", str(i)), f"encoder.layer.{i}."
        for fx, tc in (("norm1", "norm1"), ("norm2", "norm2")):
            sd[T + tc + ".weight"] = get(L + (fx, "scale"))
            sd[T + tc + ".bias"] = get(L + (fx, "bias"))
        for nm in ("query", "key", "value"):
            sd[T + f"attention.attention.{nm}.weight"] = get(L + ("attention", "attention", nm, "kernel")).t().contiguous()
            sd[T + f"attention.attention.{nm}.bias"] = get(L + ("attention", "attention", nm, "bias"))
        sd[T + "attention.output.dense.weight"] = get(L + ("attention", "output", "dense", "kernel")).t().contiguous()
        sd[T + "attention.output.dense.bias"] = get(L + ("attention", "output", "dense", "bias"))
        sd[T + "layer_scale1.lambda1"] = get(L + ("layer_scale1", "lambda1"))
        sd[T + "layer_scale2.lambda1"] = get(L + ("layer_scale2", "lambda1"))
        for fc in ("fc1", "fc2"):
            sd[T + f"mlp.{fc}.weight"] = get(L + ("mlp", fc, "kernel")).t().contiguous()
            sd[T + f"mlp.{fc}.bias"] = get(L + ("mlp", fc, "bias"))
    sd["layernorm.weight"] = get(("layernorm", "scale"))
    sd["layernorm.bias"] = get(("layernorm", "bias"))
    missing, unexpected = m.load_state_dict(sd, strict=False)
    assert not unexpected and all("mask_token" in k or False for k in missing) or not missing, (missing, unexpected)
    return m.to(dtype)


_MEAN = torch.tensor([0.485, 0.456, 0.406])
_STD = torch.tensor([0.229, 0.224, 0.225])


def encode_images(model, images_u8, dtype=torch.float32):
    """base_vit.py:111-122: normalise NHWC uint8, run DINOv2 (torch model wants NCHW), drop CLS."""
    x = torch.as_tensor(np.asarray(images_u8)).to(dtype)
    if x.ndim == 5:
        x = x[:, 0]
    x = (x / 255.0 - _MEAN.to(dtype)) / _STD.to(dtype)
    with torch.no_grad():
        hs = model(pixel_values=x.permute(0, 3, 1, 2).contiguous()).last_hidden_state
    return hs[:, 1:]


# ---------------------------------------------------------------- generated policy
class PolicyRef:
    """ViT.__call__ (DINOv2 branch, base_vit.py:130-227) + MixActionHead (action_heads.py:431-472,
    524-538) evaluated for a batch of episodes that each carry their own weights (``theta`` is the
    flat [B, G] vector in reference leaf order; batched matmuls play the role of jax.vmap in
    scripts/train.py:453-454)."""

    def __init__(self, g, leaves):
        self.g = g
        self.idx = {l.flat_name: l for l in leaves}

    def leaf(self, theta, name):
        l = self.idx[name]
        return theta[:, l.offset:l.offset + l.size].reshape((theta.shape[0],) + tuple(l.shape))

    def __call__(self, theta, tokens, quant=None):
        g = self.g
        B, P, E = tokens.shape
        D, H, hd = g.dim, g.heads, g.head_dim
        lf = lambda n: self.leaf(theta, n)
        bmm = (lambda a, b: torch.bmm(quant(a), quant(b))) if quant is not None else torch.bmm
        x = bmm(tokens, lf("encoder_image_embedding_projection_kernel")) + lf("encoder_image_embedding_projection_bias")[:, None]
        x = torch.cat([x, torch.zeros(B, 1, D, dtype=x.dtype)], 1) + lf("encoder_pos_embedding")[:, 0]
        S = P + 1
        keep = torch.ones(S, S, dtype=torch.bool)
        keep[:-1, -1] = False

        def ln(x, pre):
            s, b = lf(pre + "_scale")[:, None], lf(pre + "_bias")[:, None]
            return Fn.layer_norm(x, (D,), None, None, eps=1e-6) * s + b

        for l in range(g.layers):
            pb = f"encoder_Transformer_0_encoderblock_{l}_"
            pa = pb + "MultiHeadDotProductAttention_0_"
            h = ln(x, pb + "LayerNorm_0")
            q = bmm(h, lf(pa + "query_kernel").reshape(B, D, D)) + lf(pa + "query_bias").reshape(B, 1, D)
            k = bmm(h, lf(pa + "key_kernel").reshape(B, D, D)) + lf(pa + "key_bias").reshape(B, 1, D)
            v = bmm(h, lf(pa + "value_kernel").reshape(B, D, D)) + lf(pa + "value_bias").reshape(B, 1, D)
            q, k, v = (t.reshape(B, S, H, hd).transpose(1, 2) for t in (q, k, v))
            if quant is not None:
                q, k, v = quant(q), quant(k), quant(v)
            o = Fn.scaled_dot_product_attention(q, k, v, attn_mask=keep)
            o = o.transpose(1, 2).reshape(B, S, D)
            x = x + bmm(o, lf(pa + "out_kernel").reshape(B, D, D)) + lf(pa + "out_bias")[:, None]
            y = ln(x, pb + "LayerNorm_1")
            y = Fn.gelu(bmm(y, lf(pb + "MlpBlock_0_Dense_0_kernel")) + lf(pb + "MlpBlock_0_Dense_0_bias")[:, None], approximate="tanh")
            x = x + bmm(y, lf(pb + "MlpBlock_0_Dense_1_kernel")) + lf(pb + "MlpBlock_0_Dense_1_bias")[:, None]
        x = ln(x, "encoder_Transformer_0_encoder_norm")
        emb = x[:, -1:]
        cont = torch.bmm(emb, lf("action_head_continuous_head_kernel"))[:, 0] + lf("action_head_continuous_head_bias")
        logit = torch.bmm(emb, lf("action_head_discrete_head_kernel"))[:, 0] + lf("action_head_discrete_head_bias")
        cont = torch.tanh(cont.reshape(B, g.horizon, g.action_dim - 1) / g.tanh_scale) * g.max_action
        act = torch.cat([cont, (logit >= 0).to(cont.dtype)[..., None]], -1)
        return act, logit, emb[:, 0]


class FullRef:
    """create_tasks + sample_actions end to end (the CPU baseline object)."""

    def __init__(self, params, g, leaves, enc_shapes, dtype=torch.float32, with_encoder=True):
        self.g, self.dt = g, dtype
        self.hn = HyperNetRef(params, g, leaves, dtype)
        self.pol = PolicyRef(g, leaves)
        self.enc = build_hf_dinov2(params, g, enc_shapes, dtype) if with_encoder else None

    @torch.no_grad()
    def create_tasks(self, instruction_dict, initial_state, quant=None):
        li = instruction_dict["language_instruction"]
        ctx = self.hn.context(li["token_embedding"], li["attention_mask"],
                              np.asarray(initial_state["patch_embeddings"])[:, 0], quant)
        return self.hn.generate(ctx, quant), ctx

    @torch.no_grad()
    def sample_actions(self, theta, images_u8, quant=None):
        tokens = encode_images(self.enc, images_u8, self.dt)
        return self.pol(theta, tokens, quant) + (tokens,)


# ---------------------------------------------------------------- A13: loss and its gradient (round-2 oracle)
def mix_loss(g, cont, logits, actions, timestep_pad_mask, action_pad_mask, clip_target=True):
    """Per-sample MixActionHead.loss (action_heads.py:474-522) and the batch mean (scripts/train.py:453-457)."""
    a = torch.as_tensor(np.asarray(actions)).to(cont.dtype)
    if clip_target:
        a = a.clamp(-g.max_action, g.max_action)
    a = a[:, 0]                                                       # window 1
    m = (torch.as_tensor(np.asarray(timestep_pad_mask)).bool()[:, :, None, None]
         & torch.as_tensor(np.asarray(action_pad_mask)).bool())[:, 0].to(cont.dtype)
    sq = (cont - a[..., :-1]) ** 2
    mc, md = m[..., :-1], m[..., -1]
    cont_term = (sq * mc).mean((1, 2)) / mc.mean((1, 2)).clamp_min(1e-5)
    bce = Fn.binary_cross_entropy_with_logits(logits, a[..., -1], reduction="none")
    disc_term = (bce * md).mean(1) / md.mean(1).clamp_min(1e-5)
    per = cont_term * (g.action_dim - 1) + disc_term
    return per, per.mean()


def train_loss_and_grads(params, g, leaves, instruction_dict, initial_state, tokens, batch, dtype=torch.float64,
                         images=None, enc_shapes=None, clip_target=None):
    """Loss and d(loss)/d(HN params): autograd through hypernetwork -> generated theta -> per-sample policy -> mix
    loss.  With `images` (and `enc_shapes`) the DINOv2 encoder is part of the graph
    (base_vit.py:128 `fine_tune_pretrained_image_encoder=True`) and the returned dict also holds the gradient of every
    shared `encoder_image_encoder_*` leaf as a flat vector in the checkpoint's (flax) layout; otherwise the encoder is
    frozen and `tokens` are given.  This is the gradient oracle for the fine-tune step."""
    hn = HyperNetRef(params, g, leaves, dtype)
    names = sorted(hn.p)
    for k in names:
        hn.p[k] = hn.p[k].clone().requires_grad_(True)
    hn.w_cat = torch.cat([hn.p[l.head_name + "/kernel"] for l in leaves], dim=1)
    hn.b_cat = torch.cat([hn.p[l.head_name + "/bias"] for l in leaves], dim=0)
    li = instruction_dict["language_instruction"]
    ctx = hn.context(li["token_embedding"], li["attention_mask"], np.asarray(initial_state["patch_embeddings"])[:, 0])
    theta = hn.generate(ctx)
    enc = None
    if images is not None:
        enc = build_hf_dinov2(params, g, enc_shapes, dtype).train(False)
        for q in enc.parameters():
            q.requires_grad_(True)
        x = torch.as_tensor(np.asarray(images)).to(dtype)
        if x.ndim == 5:
            x = x[:, 0]
        x = (x / 255.0 - _MEAN.to(dtype)) / _STD.to(dtype)
        tok_t = enc(pixel_values=x.permute(0, 3, 1, 2).contiguous()).last_hidden_state[:, 1:]
    else:
        tok_t = torch.as_tensor(np.asarray(tokens)).to(dtype)
    act, logit, _ = PolicyRef(g, leaves)(theta, tok_t)
    if clip_target is None:
        clip_target = bool(getattr(g, "clip_target", True))
    per, loss = mix_loss(g, act[..., :-1], logit, batch["action"], batch["timestep_pad_mask"], batch["action_pad_mask"],
                         clip_target=clip_target)
    wrt = [hn.p[k] for k in names]
    enc_named = list(enc.named_parameters()) if enc is not None else []
    grads = torch.autograd.grad(loss, wrt + [q for _, q in enc_named], allow_unused=True)
    out = {k: (gr.detach() if gr is not None else torch.zeros_like(hn.p[k])) for k, gr in zip(names, grads)}
    if enc is not None:
        hf = {n: (gr.detach() if gr is not None else torch.zeros_like(q)) for (n, q), gr in zip(enc_named, grads[len(wrt):])}
        out.update(_hf_grads_to_flax(hf, g))
    return per.detach(), loss.detach(), out


def _hf_grads_to_flax(hf, g):
    """torch Dinov2Model parameter names / layouts -> the checkpoint's flat `encoder_image_encoder_*` vectors
    (inverse of build_hf_dinov2: Linear weights transposed back to [in, out], conv OIHW -> HWIO)."""
    pre, out = "encoder_image_encoder_", {}

    def put(path, t):
        out[pre + "_".join(path)] = t.contiguous().reshape(-1)

    put(("embeddings", "cls_token"), hf["embeddings.cls_token"])
    put(("embeddings", "mask_token"), hf["embeddings.mask_token"])
    put(("embeddings", "position_embeddings"), hf["embeddings.position_embeddings"])
    put(("embeddings", "patch_embeddings", "projection", "kernel"), hf["embeddings.patch_embeddings.projection.weight"].permute(2, 3, 1, 0))
    put(("embeddings", "patch_embeddings", "projection", "bias"), hf["embeddings.patch_embeddings.projection.bias"])
    for i in range(g.enc_layers):
        L, T = ("encoder", "layer", str(i)), f"encoder.layer.{i}."
        for nm in ("norm1", "norm2"):
            put(L + (nm, "scale"), hf[T + nm + ".weight"])
            put(L + (nm, "bias"), hf[T + nm + ".bias"])
        for nm in ("query", "key", "value"):
            put(L + ("attention", "attention", nm, "kernel"), hf[T + f"attention.attention.{nm}.weight"].t())
            put(L + ("attention", "attention", nm, "bias"), hf[T + f"attention.attention.{nm}.bias"])
        put(L + ("attention", "output", "dense", "kernel"), hf[T + "attention.output.dense.weight"].t())
        put(L + ("attention", "output", "dense", "bias"), hf[T + "attention.output.dense.bias"])
        put(L + ("layer_scale1", "lambda1"), hf[T + "layer_scale1.lambda1"])
        put(L + ("layer_scale2", "lambda1"), hf[T + "layer_scale2.lambda1"])
        for fc in ("fc1", "fc2"):
            put(L + ("mlp", fc, "kernel"), hf[T + f"mlp.{fc}.weight"].t())
            put(L + ("mlp", fc, "bias"), hf[T + f"mlp.{fc}.bias"])
    put(("layernorm", "scale"), hf["layernorm.weight"])
    put(("layernorm", "bias"), hf["layernorm.bias"])
    return out
