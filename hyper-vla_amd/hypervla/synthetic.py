"""Deterministic synthetic weights and OXE-shaped inputs (no checkpoint / dataset exists offline).

Definitions follow SURVEY.md §8(d) "Synthetic inputs": everything is drawn from
``numpy.random.Generator(PCG64(seed))`` with seeds fixed here, so the GPU path, the oracle and the
golden fixtures all see the same numbers.  Every one of the 73 generated leaves is made
context-dependent (non-zero head kernels) and every bias / LayerNorm parameter is non-trivial so
that a dropped bias or a swapped scale shows up in parity tests.
"""
from __future__ import annotations

from typing import Dict, Tuple

import numpy as np

from .config import (FULL, T5_BASE, Geometry, T5Geometry, encoder_leaves, generated_leaves, hypernet_param_shapes,
                     shared_name, t5_param_shapes)

SEED_IMAGES, SEED_TOKENS, SEED_CLS, SEED_STATS, SEED_WEIGHTS = 1000, 2000, 3000, 4000, 5000


def _rng(seed: int) -> np.random.Generator:
    return np.random.Generator(np.random.PCG64(seed))


def _xavier_std(fan_in: int, fan_out: int) -> float:
    return float(np.sqrt(2.0 / (fan_in + fan_out)))


def _leaf_natural(path: Tuple[str, ...], shape: Tuple[int, ...], g: Geometry):
    """(std of a 'standard init' of this base-net leaf, mean) — the head *bias* is drawn from it."""
    last = path[-1]
    if last == "scale":
        return 0.05, 1.0
    if last == "bias":
        return 0.02, 0.0
    if last == "pos_embedding":
        return 0.02, 0.0
    # kernels: xavier over (fan_in, fan_out) of the logical matmul
    if path[-2] == "out":                       # [H, hd, D]
        fi, fo = shape[0] * shape[1], shape[2]
    elif path[-2] in ("query", "key", "value"):  # [D, H, hd]
        fi, fo = shape[0], shape[1] * shape[2]
    else:
        fi, fo = shape[0], shape[1]
    return _xavier_std(fi, fo), 0.0


def synthetic_params(g: Geometry = FULL, seed: int = SEED_WEIGHTS) -> Dict[str, np.ndarray]:
    """HN checkpoint (reference naming) with float32 leaves."""
    rng = _rng(seed)
    shapes = hypernet_param_shapes(g)
    p: Dict[str, np.ndarray] = {}

    def normal(shape, std, mean=0.0):
        return (mean + std * rng.standard_normal(shape)).astype(np.float32)

    C = g.ctx_dim
    p["task_token_projection/kernel"] = normal(shapes["task_token_projection/kernel"],
                                               _xavier_std(g.lang_dim, C))
    p["task_token_projection/bias"] = normal((C,), 0.02)
    p["initial_image_projection/kernel"] = normal(shapes["initial_image_projection/kernel"],
                                                  _xavier_std(g.enc_dim, C))
    p["initial_image_projection/bias"] = normal((C,), 0.02)
    for nm in ("task_pos_embedding", "initial_image_pos_embedding", "layer_pos_embedding"):
        # layer token is all-zeros + pos-emb (hypernetwork.py:144-145): give it unit scale so the
        # context embedding is not degenerate
        p[nm] = normal(shapes[nm], 1.0 if nm == "layer_pos_embedding" else 0.02)
    for l in range(g.ctx_layers):
        b = f"Transformer_0/encoderblock_{l}/"
        for ln in ("LayerNorm_0", "LayerNorm_1"):
            p[b + ln + "/scale"] = normal((C,), 0.05, 1.0)
            p[b + ln + "/bias"] = normal((C,), 0.02)
        a = b + "MultiHeadDotProductAttention_0/"
        for nm in ("query", "key", "value"):
            p[a + nm + "/kernel"] = normal(shapes[a + nm + "/kernel"], _xavier_std(C, C))
            p[a + nm + "/bias"] = normal(shapes[a + nm + "/bias"], 0.02)
        p[a + "out/kernel"] = normal(shapes[a + "out/kernel"], _xavier_std(C, C))
        p[a + "out/bias"] = normal((C,), 0.02)
        p[b + "MlpBlock_0/Dense_0/kernel"] = normal((C, g.ctx_mlp), _xavier_std(C, g.ctx_mlp))
        p[b + "MlpBlock_0/Dense_0/bias"] = normal((g.ctx_mlp,), 0.02)
        p[b + "MlpBlock_0/Dense_1/kernel"] = normal((g.ctx_mlp, C), _xavier_std(g.ctx_mlp, C))
        p[b + "MlpBlock_0/Dense_1/bias"] = normal((C,), 0.02)
    p["Transformer_0/encoder_norm/scale"] = normal((C,), 0.05, 1.0)
    p["Transformer_0/encoder_norm/bias"] = normal((C,), 0.02)

    # 73 output heads: bias = a standard init of the base net, kernel = context-dependent delta
    k_scale = 0.5 / np.sqrt(C) if g.scale_context else 0.5 / C
    for lf in generated_leaves(g):
        std, mean = _leaf_natural(lf.path, lf.shape, g)
        p[lf.head_name + "/bias"] = normal((lf.size,), std, mean)
        # ctx = LN(.)/sqrt(C) has |ctx|~1 => delta ~ 0.5*std of the leaf's own scale
        p[lf.head_name + "/kernel"] = normal((C, lf.size), k_scale * std * np.sqrt(C))

    # shared DINOv2 leaves (flat vectors)
    for path, shape in encoder_leaves(g):
        last = path[-1]
        n = int(np.prod(shape))
        if last == "scale":
            v = normal((n,), 0.05, 1.0)
        elif last == "lambda1":
            v = normal((n,), 0.05, 1.0)
        elif last == "bias":
            v = normal((n,), 0.02)
        else:   # kernels, cls/mask token, position embeddings
            v = normal((n,), 0.02)
        p[shared_name(path)] = v
    assert set(p) == set(shapes)
    for k, v in p.items():
        assert v.shape == tuple(shapes[k]), (k, v.shape, shapes[k])
    return p


def synthetic_params_trained_like(g: Geometry = FULL, seed: int = SEED_WEIGHTS + 1) -> Dict[str, np.ndarray]:
    """`synthetic_params` with the image encoder re-drawn to the statistics a *trained* DINOv2 shows and random-init
    weights do not (no pretrained checkpoint exists offline): LayerScale in [0.05, 1] instead of 1.0, a few outlier
    channels of magnitude >= 100 in the residual stream from the second layer on (the 'massive activations' of ViTs:
    injected by fc2's bias plus token-dependent fc2 columns, and damped by small LayerNorm scales on those channels as
    trained models do), heavy-tailed LayerNorm scales and fc1 columns (pre-activations up to a few tens), sharper
    attention.  Used by the parity tests to check that 16-bit operands neither overflow nor lose the tolerance on such
    weights; the hypernetwork half is unchanged."""
    p = synthetic_params(g, SEED_WEIGHTS)
    rng = _rng(seed)
    E, F, nl = g.enc_dim, g.enc_mlp, g.enc_layers
    outl = rng.choice(E, size=4, replace=False)                       # the outlier channels
    sign = rng.choice([-1.0, 1.0], size=4)

    def put(path, v):
        p[shared_name(path)] = np.asarray(v, np.float32).reshape(-1)

    def leaf(path, shape):
        return p[shared_name(path)].reshape(shape).astype(np.float64)

    for i in range(nl):
        L = ("encoder", "layer", str(i))
        lam1, lam2 = rng.uniform(0.05, 1.0, size=E), rng.uniform(0.05, 1.0, size=E)
        put(L + ("layer_scale1", "lambda1"), lam1)
        put(L + ("layer_scale2", "lambda1"), lam2)
        for nm in ("norm1", "norm2"):
            sc = np.exp(0.5 * rng.standard_normal(E))
            sc[outl] = 0.05 * rng.uniform(0.5, 1.5, size=4)            # trained models damp the outlier channels
            put(L + (nm, "scale"), sc)
            put(L + (nm, "bias"), 0.1 * rng.standard_normal(E))
        w1 = leaf(L + ("mlp", "fc1", "kernel"), (E, F)) * np.exp(0.8 * rng.standard_normal(F))[None, :]
        put(L + ("mlp", "fc1", "kernel"), w1)
        put(L + ("mlp", "fc1", "bias"), 0.5 * rng.standard_normal(F))
        for nm, k in (("query", 2.0), ("key", 2.0)):
            put(L + ("attention", "attention", nm, "kernel"), leaf(L + ("attention", "attention", nm, "kernel"), (E, E)) * k)
        if i == 1:                                                     # the layer that creates the outliers
            w2 = leaf(L + ("mlp", "fc2", "kernel"), (F, E))
            b2 = leaf(L + ("mlp", "fc2", "bias"), (E,))
            w2[:, outl] *= 30.0                                        # token-dependent part
            b2[outl] = sign * 130.0 / lam2[outl]                       # constant part: |x| >= 100 afterwards
            put(L + ("mlp", "fc2", "kernel"), w2)
            put(L + ("mlp", "fc2", "bias"), b2)
    return p


def synthetic_images_structured(batch: int, g: Geometry = FULL, rank: int = 0) -> np.ndarray:
    """uint8 [B, 1, H, W, 3] observations that differ from image to image the way camera frames do and iid noise does
    not: per-image brightness / contrast / colour cast, a smooth low-frequency scene (a few random blobs and a
    gradient) and mild sensor noise.  Parity tests use them next to `synthetic_images` so that per-image activation
    statistics are not all alike."""
    rng = _rng(SEED_IMAGES + 500 + rank)
    S = g.image_size
    yy, xx = np.meshgrid(np.linspace(-1, 1, S), np.linspace(-1, 1, S), indexing="ij")
    out = np.empty((batch, 1, S, S, 3), np.uint8)
    for b in range(batch):
        img = np.zeros((S, S, 3))
        gx, gy = rng.normal(0, 0.4, size=2)
        img += (gx * xx + gy * yy)[..., None]
        for _ in range(int(rng.integers(3, 9))):
            cx, cy, r = rng.uniform(-1, 1), rng.uniform(-1, 1), rng.uniform(0.1, 0.6)
            col = rng.normal(0, 0.6, size=3)
            img += np.exp(-((xx - cx) ** 2 + (yy - cy) ** 2) / (2 * r * r))[..., None] * col
        img = img * rng.uniform(0.3, 1.2) + rng.uniform(-0.5, 0.5) + rng.normal(0, 0.15, size=3)
        img += rng.normal(0, rng.uniform(0.01, 0.08), size=img.shape)
        out[b, 0] = np.clip(np.rint(127.5 + 100.0 * img), 0, 255).astype(np.uint8)
    return out


def synthetic_images(batch: int, g: Geometry = FULL, rank: int = 0) -> np.ndarray:
    """uint8 [B, 1, H, W, 3] observations (OXE ``image_primary`` with window 1)."""
    rng = _rng(SEED_IMAGES + rank)
    return rng.integers(0, 256, size=(batch, 1, g.image_size, g.image_size, 3), dtype=np.uint8)


def synthetic_instructions(batch: int, g: Geometry = FULL, rank: int = 0) -> Dict:
    """``instruction_dict`` as the evaluators build it (data/simpler/evaluate.py:235-254)."""
    rng = _rng(SEED_TOKENS + rank)
    emb = (0.2 * rng.standard_normal((batch, g.lang_tokens, g.lang_dim))).astype(np.float32)
    lo, hi = 3, max(4, min(20, g.lang_tokens))
    n = rng.integers(lo, hi + 1, size=batch)
    mask = (np.arange(g.lang_tokens)[None, :] < n[:, None]).astype(np.int64)
    ids = rng.integers(2, 32000, size=(batch, g.lang_tokens)).astype(np.int64) * mask
    return {"language_instruction": {"input_ids": ids, "attention_mask": mask,
                                     "token_embedding": emb}}


def synthetic_initial_state(batch: int, g: Geometry = FULL, rank: int = 0) -> Dict:
    """``initial_state`` with frozen-DINOv2 ``patch_embeddings`` [B, 1+P, E]; only the CLS row is
    read by the hypernetwork (hypernetwork.py:118-128) so the other rows are left zero."""
    rng = _rng(SEED_CLS + rank)
    pe = np.zeros((batch, g.patches + 1, g.enc_dim), np.float32)
    pe[:, 0] = rng.standard_normal((batch, g.enc_dim)).astype(np.float32)
    return {"patch_embeddings": pe, "pad_mask_dict": {"image_primary": np.ones((batch, 1))}}


def synthetic_dataset_statistics(g: Geometry = FULL) -> Dict:
    rng = _rng(SEED_STATS)
    a = g.action_dim
    mean = (0.1 * rng.standard_normal(a)).astype(np.float32)
    std = rng.uniform(0.05, 0.5, size=a).astype(np.float32)
    mask = np.array([True] * (a - 1) + [False])
    p01 = (mean - 2.3 * std).astype(np.float32)
    p99 = (mean + 2.3 * std).astype(np.float32)
    stats = {"action": {"mean": mean, "std": std, "mask": mask, "p01": p01, "p99": p99}}
    return {"bridge_dataset": stats, "fractal20220817_data": stats, "libero": stats}


def synthetic_action_batch(batch: int, g: Geometry = FULL, rank: int = 0) -> Dict:
    """OXE-shaped training targets (octo/data/traj_transforms.py:11-99): action f32[B,1,H,7] (normalised, a few
    beyond +-max_action so clip_target matters), action_pad_mask bool[B,1,H,7], timestep_pad_mask bool[B,1]."""
    rng = _rng(6000 + rank)
    a = rng.normal(0.0, 2.5, size=(batch, 1, g.horizon, g.action_dim)).astype(np.float32)
    a[..., -1] = (rng.uniform(size=(batch, 1, g.horizon)) > 0.5).astype(np.float32)
    apm = rng.uniform(size=a.shape) > 0.15
    tpm = rng.uniform(size=(batch, 1)) > 0.1
    return {"action": a, "action_pad_mask": apm, "timestep_pad_mask": tpm}


def synthetic_t5_params(t: T5Geometry = T5_BASE, seed: int = 7000) -> Dict[str, np.ndarray]:
    """Random T5 encoder weights under the FlaxT5EncoderModel names (no checkpoint exists offline).  Scales follow
    T5's own initialisation closely enough that activations stay O(1) through 12 layers."""
    rng = _rng(seed)
    p: Dict[str, np.ndarray] = {}
    for k, shp in t5_param_shapes(t).items():
        if k.endswith("layer_norm/weight"):
            v = 1.0 + 0.05 * rng.standard_normal(shp)
        elif k.endswith("relative_attention_bias/embedding"):
            v = 0.5 * rng.standard_normal(shp)
        elif k == "shared/embedding":
            v = rng.standard_normal(shp)
        elif k.endswith("q/kernel"):
            v = rng.standard_normal(shp) * (t.d_model * t.d_kv) ** -0.5
        elif k.endswith("wo/kernel"):
            v = rng.standard_normal(shp) * t.d_ff ** -0.5
        elif k.endswith("o/kernel"):
            v = rng.standard_normal(shp) * t.inner ** -0.5
        else:
            v = rng.standard_normal(shp) * t.d_model ** -0.5
        p[k] = v.astype(np.float32)
    return p


def synthetic_token_ids(batch: int, t: T5Geometry = T5_BASE, tokens: int = 32, rank: int = 0) -> Dict:
    """input_ids / attention_mask as HFTokenizer(max_length=32, padding="max_length") returns them: ids, EOS (1), pads (0)."""
    rng = _rng(SEED_TOKENS + 500 + rank)
    n = rng.integers(3, min(20, tokens) + 1, size=batch)
    ids = rng.integers(2, t.vocab, size=(batch, tokens)).astype(np.int64)
    mask = (np.arange(tokens)[None, :] < n[:, None]).astype(np.int64)
    ids = ids * mask
    ids[np.arange(batch), n - 1] = 1
    return {"input_ids": ids, "attention_mask": mask}


# ------------------------------------------------------------------------------------------------ reference pin (tools/make_reference_fixtures.py)
SEED_POS37 = 5100


def synthetic_position_table_hub(g: Geometry = FULL, n: int = 37, seed: int = SEED_POS37) -> np.ndarray:
    """A position table of the SHAPE the reference's DINOv2 module declares -- [1, 1 + 37*37, E], the hub config's
    image_size 518 / patch 14 (hypervla/components/base_vit.py:76-77) -- which `FlaxDinov2Embeddings` resizes to the
    16 x 16 grid inside every forward pass.  Seeded, never stored: the reference runs on this table
    (tools/make_reference_fixtures.py) and this build on `convert.bake_position_embeddings` of it."""
    return _rng(seed).normal(0.0, 0.02, size=(1, 1 + n * n, g.enc_dim)).astype(np.float32)


def synthetic_params_for_reference_pin(g: Geometry = FULL) -> Dict[str, np.ndarray]:
    """`synthetic_params(g)` with the DINOv2 position table replaced by the hub-shaped one baked to this geometry's grid:
    the parameters the HIP path / the oracle run on when they are compared with tests/golden/reference_full_b4.npz."""
    from .convert import bake_position_embeddings
    P = dict(synthetic_params(g))
    key = next(k for k in P if k.endswith("embeddings_position_embeddings"))
    P[key] = bake_position_embeddings(synthetic_position_table_hub(g), g.image_size // g.patch).reshape(-1).astype(np.float32)
    return P
