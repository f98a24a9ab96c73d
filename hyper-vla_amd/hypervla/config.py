"""Static configuration + base-network metadata for the HyperVLA action-prediction path.

The reference keeps every model hyper-parameter in ``config.json`` (hypervla/model.py:152-163) and
derives the generated-leaf metadata in ``HyperVLA.init_base_net`` (hypervla/model.py:370-515).  This
module restates the README run's effective values (README.md:33-61) and the leaf list
the hypernetwork generates: 73 leaves, G = 201 500 parameters per episode for
E = 768.

Nothing here touches the GPU; it is pure host metadata shared by the product path, the tests and
the synthetic generators.
"""
from __future__ import annotations

import copy
from dataclasses import dataclass, field
from typing import Dict, List, Tuple

import numpy as np


# --------------------------------------------------------------------------------------
# Geometry of the path (all sizes are parameters so a tiny config can exercise every branch)
# --------------------------------------------------------------------------------------
@dataclass(frozen=True)
class Geometry:
    # frozen image encoder (HF Dinov2, reference: "facebook/dinov2-base", base_vit.py:76)
    image_size: int = 224
    patch: int = 14
    enc_dim: int = 768          # E
    enc_layers: int = 12
    enc_heads: int = 12
    enc_mlp: int = 3072
    # generated policy ("vit_t": README.md:47-50)
    dim: int = 64               # D
    layers: int = 4             # L
    heads: int = 4              # H
    mlp: int = 128              # M
    horizon: int = 4            # action_horizon
    action_dim: int = 7
    tanh_scale: float = 5.0
    max_action: float = 5.0
    clip_target: bool = True    # MixActionHead.loss clips the action target to +-max_action (action_heads.py:499-500)
    # hypernetwork (README.md:33-44)
    ctx_dim: int = 128          # C
    ctx_layers: int = 6
    ctx_heads: int = 4
    ctx_mlp: int = 512
    lang_tokens: int = 32       # T5 max_length (hypervla_pretrain_config.py:302-312)
    lang_dim: int = 768         # T5-base hidden size
    scale_context: bool = True

    @property
    def grid(self) -> int:
        return self.image_size // self.patch

    @property
    def patches(self) -> int:      # P
        return self.grid * self.grid

    @property
    def seq(self) -> int:          # S = patches + 1 action token (policy) / + CLS (encoder)
        return self.patches + 1

    @property
    def ctx_seq(self) -> int:      # language tokens + initial-image CLS + 1 layer token
        return self.lang_tokens + 2

    @property
    def head_dim(self) -> int:
        return self.dim // self.heads

    @property
    def patch_in(self) -> int:     # 14*14*3
        return self.patch * self.patch * 3


FULL = Geometry()
# DINOv2-small token width named by BASELINE.json config 2 (policy-only variant)
SMALL_E = Geometry(enc_dim=384, enc_heads=6, enc_mlp=1536)
# Reduced geometry inside the HIP kernels' specialisation (P % 32 == 0, encoder head_dim 64, policy
# 64-d / 4 heads): fast full-tensor GPU parity runs against the live oracle
MID = Geometry(image_size=112, patch=14, enc_dim=128, enc_layers=2, enc_heads=2, enc_mlp=512,
               layers=2, ctx_layers=2, ctx_mlp=256, lang_tokens=12, lang_dim=64)
# A tiny geometry that still walks every code path (used for exhaustive intermediate fixtures)
TINY = Geometry(image_size=56, patch=14, enc_dim=32, enc_layers=2, enc_heads=2, enc_mlp=64,
                dim=16, layers=2, heads=2, mlp=32, ctx_dim=16, ctx_layers=2, ctx_heads=2, ctx_mlp=32,
                lang_tokens=8, lang_dim=24)


# --------------------------------------------------------------------------------------
# Generated-leaf metadata  (reference: init_base_net, hypervla/model.py:370-515)
# --------------------------------------------------------------------------------------
@dataclass(frozen=True)
class T5Geometry:
    """Frozen instruction encoder (``LanguageTokenizer('t5-base')``, data/utils/language_tokenizer.py:9-28;
    octo/model/components/tokenizers.py:186-211): HF ``T5Config`` fields of "t5-base"."""
    vocab: int = 32128
    d_model: int = 768
    d_kv: int = 64
    heads: int = 12
    d_ff: int = 3072
    layers: int = 12
    buckets: int = 32            # relative_attention_num_buckets
    max_distance: int = 128      # relative_attention_max_distance
    eps: float = 1e-6

    @property
    def inner(self) -> int:
        return self.heads * self.d_kv


T5_BASE = T5Geometry()
T5_TINY = T5Geometry(vocab=300, d_model=32, d_kv=8, heads=2, d_ff=64, layers=2)      # pairs with TINY (lang_dim 32)
T5_MID = T5Geometry(vocab=1000, d_model=768, d_kv=64, heads=12, d_ff=3072, layers=2)  # full widths, 2 layers (tests)


def t5_param_shapes(t: T5Geometry) -> Dict[str, Tuple[int, ...]]:
    """FlaxT5EncoderModel parameter tree, '/'-joined (kernels are [in, out])."""
    s: Dict[str, Tuple[int, ...]] = {"shared/embedding": (t.vocab, t.d_model)}
    for i in range(t.layers):
        b = f"encoder/block/{i}/layer/"
        for nm in ("q", "k", "v"):
            s[b + f"0/SelfAttention/{nm}/kernel"] = (t.d_model, t.inner)
        s[b + "0/SelfAttention/o/kernel"] = (t.inner, t.d_model)
        s[b + "0/layer_norm/weight"] = (t.d_model,)
        s[b + "1/DenseReluDense/wi/kernel"] = (t.d_model, t.d_ff)
        s[b + "1/DenseReluDense/wo/kernel"] = (t.d_ff, t.d_model)
        s[b + "1/layer_norm/weight"] = (t.d_model,)
    s["encoder/block/0/layer/0/SelfAttention/relative_attention_bias/embedding"] = (t.buckets, t.heads)
    s["encoder/final_layer_norm/weight"] = (t.d_model,)
    return s


@dataclass(frozen=True)
class Leaf:
    path: Tuple[str, ...]       # base-net pytree path
    shape: Tuple[int, ...]
    offset: int                 # offset into the per-episode flat parameter vector (reference order)

    @property
    def size(self) -> int:
        return int(np.prod(self.shape))

    @property
    def flat_name(self) -> str:     # hypervla/model.py:532-540 (flatten_dict, sep='_')
        return "_".join(self.path)

    @property
    def head_name(self) -> str:     # flax auto-name of the per-leaf Dense: hypernetwork.py:65-67
        return "output_head_" + self.flat_name


def generated_leaves(g: Geometry) -> List[Leaf]:
    """The 73 HN-generated leaves in jax pytree order (dict keys sorted at every level).

    ``shared_modules=("image_encoder",)`` removes the DINOv2 leaves (hypervla/model.py:439-451);
    ``share_layer_index=True`` makes every leaf read context token 0 (model.py:400-402).
    """
    D, M, H, hd, E = g.dim, g.mlp, g.heads, g.head_dim, g.enc_dim
    A = g.horizon * (g.action_dim - 1)
    items: List[Tuple[Tuple[str, ...], Tuple[int, ...]]] = []
    items += [(("action_head", "continuous_head", "bias"), (A,)),
              (("action_head", "continuous_head", "kernel"), (D, A)),
              (("action_head", "discrete_head", "bias"), (g.horizon,)),
              (("action_head", "discrete_head", "kernel"), (D, g.horizon))]
    T = ("encoder", "Transformer_0")
    items += [(T + ("encoder_norm", "bias"), (D,)), (T + ("encoder_norm", "scale"), (D,))]
    for l in range(g.layers):
        B = T + (f"encoderblock_{l}",)
        items += [(B + ("LayerNorm_0", "bias"), (D,)), (B + ("LayerNorm_0", "scale"), (D,)),
                  (B + ("LayerNorm_1", "bias"), (D,)), (B + ("LayerNorm_1", "scale"), (D,)),
                  (B + ("MlpBlock_0", "Dense_0", "bias"), (M,)),
                  (B + ("MlpBlock_0", "Dense_0", "kernel"), (D, M)),
                  (B + ("MlpBlock_0", "Dense_1", "bias"), (D,)),
                  (B + ("MlpBlock_0", "Dense_1", "kernel"), (M, D))]
        A_ = B + ("MultiHeadDotProductAttention_0",)
        items += [(A_ + ("key", "bias"), (H, hd)), (A_ + ("key", "kernel"), (D, H, hd)),
                  (A_ + ("out", "bias"), (D,)), (A_ + ("out", "kernel"), (H, hd, D)),
                  (A_ + ("query", "bias"), (H, hd)), (A_ + ("query", "kernel"), (D, H, hd)),
                  (A_ + ("value", "bias"), (H, hd)), (A_ + ("value", "kernel"), (D, H, hd))]
    items += [(("encoder", "image_embedding_projection", "bias"), (D,)),
              (("encoder", "image_embedding_projection", "kernel"), (E, D)),
              (("encoder", "pos_embedding"), (1, g.seq, D))]
    leaves, off = [], 0
    for path, shape in items:
        leaves.append(Leaf(path, tuple(shape), off))
        off += int(np.prod(shape))
    return leaves


def total_generated(g: Geometry) -> int:
    lv = generated_leaves(g)
    return lv[-1].offset + lv[-1].size


assert total_generated(FULL) == 201_500 and len(generated_leaves(FULL)) == 73


# --------------------------------------------------------------------------------------
# config.json the evaluators read (data/utils/hypervla_interface.py:76-87, model.py:152-163)
# --------------------------------------------------------------------------------------
def default_config(g: Geometry = FULL, dataset_name: str = "bridge_dataset") -> Dict:
    cfg = dict(
        window_size=1,
        dataset_kwargs=dict(
            dataset_kwargs_list=[
                dict(name="bridge_dataset", action_proprio_normalization_type="normal"),
                dict(name="fractal20220817_data", action_proprio_normalization_type="normal"),
                dict(name="libero", action_proprio_normalization_type="normal"),
            ],
            frame_transform_kwargs=dict(resize_size=dict(primary=(g.image_size, g.image_size))),
            batch_size=256,
        ),
        text_processor=dict(kwargs=dict(
            tokenizer_name="t5-base",
            tokenizer_kwargs=dict(max_length=g.lang_tokens, padding="max_length", truncation=True,
                                  return_tensors="np"))),
        hypernet_kwargs=dict(
            encoder_type="transformer", context_embedding_dim=g.ctx_dim,
            context_encoder_kwargs=dict(num_layers=g.ctx_layers, mlp_dim=g.ctx_mlp,
                                        num_attention_heads=g.ctx_heads, dropout_rate=0.0,
                                        attention_dropout_rate=0.0, add_position_embedding=False),
            attend_to_padding=False, task_attend_to_layer=False, embedding_dropout_rate=0.0,
            scale_context_embedding=g.scale_context, output_head_bias=True,
            generation_strategy="block", shared_modules=("image_encoder",),
            include_goal_image=False, use_initial_image=True, use_all_image_tokens=False,
            share_TF_output_head=False, init_strategy=0, share_all_params=False,
            share_layer_index=True, image_dropout=0.0),
        base_net_kwargs=dict(
            model_type="vit", action_head_type="mix", action_horizon=g.horizon,
            action_dim=g.action_dim,
            vit_kwargs=dict(encoder_type="DINOv2", patch_size=16, hidden_dim=g.dim,
                            num_layers=g.layers, num_heads=g.heads, mlp_dim=g.mlp,
                            dropout_rate=0.0, use_language_token=False,
                            fine_tune_pretrained_image_encoder=True, image_embedding_noise=0.0,
                            use_differential_transformer=False, return_attention_map=False,
                            add_positional_embedding=True, include_class_token=False),
            action_head_kwargs=dict(token_per_horizon=False, squash_continuous_action=True,
                                    tanh_scaling_factor=g.tanh_scale, clip_target=g.clip_target,
                                    max_action=g.max_action, hidden_dims=())),
        geometry=dict(image_size=g.image_size, patch=g.patch, enc_dim=g.enc_dim,
                      enc_layers=g.enc_layers, enc_heads=g.enc_heads, enc_mlp=g.enc_mlp,
                      lang_tokens=g.lang_tokens, lang_dim=g.lang_dim),
    )
    return copy.deepcopy(cfg)


def geometry_from_config(cfg: Dict) -> Geometry:
    """Inverse of :func:`default_config` (rejects branches the path does not build)."""
    b, h = cfg["base_net_kwargs"], cfg["hypernet_kwargs"]
    v, a = b["vit_kwargs"], b["action_head_kwargs"]
    if b["model_type"] != "vit" or b["action_head_type"] != "mix" or v["encoder_type"] != "DINOv2":
        raise ValueError("only model_type=vit / encoder_type=DINOv2 / action_head_type=mix is built "
                         "(README.md:45-56); got %r/%r/%r" % (b["model_type"], v["encoder_type"],
                                                              b["action_head_type"]))
    for key, want in (("generation_strategy", "block"), ("share_layer_index", True),
                      ("use_initial_image", True), ("use_all_image_tokens", False),
                      ("attend_to_padding", False), ("task_attend_to_layer", False)):
        if h.get(key) != want:
            raise ValueError(f"hypernet_kwargs.{key}={h.get(key)!r} is outside the built path "
                             f"(README.md:33-44 uses {want!r})")
    if a.get("token_per_horizon") or a.get("hidden_dims"):
        raise ValueError("token_per_horizon / hidden_dims action heads are not built")
    if not a.get("squash_continuous_action", True):
        raise ValueError("squash_continuous_action=False is not built: the mix head kernels apply tanh(x / s) * max_action "
                         "(action_heads.py:469-470)")
    ge = cfg.get("geometry", {})
    ce = h["context_encoder_kwargs"]
    return Geometry(
        image_size=ge.get("image_size", 224), patch=ge.get("patch", 14),
        enc_dim=ge.get("enc_dim", 768), enc_layers=ge.get("enc_layers", 12),
        enc_heads=ge.get("enc_heads", 12), enc_mlp=ge.get("enc_mlp", 3072),
        dim=v["hidden_dim"], layers=v["num_layers"], heads=v["num_heads"], mlp=v["mlp_dim"],
        horizon=b["action_horizon"], action_dim=b["action_dim"],
        tanh_scale=a.get("tanh_scaling_factor", 5.0), max_action=a.get("max_action", 5.0),
        clip_target=bool(a.get("clip_target", True)),                  # action_heads.py:408 default True
        ctx_dim=h["context_embedding_dim"], ctx_layers=ce["num_layers"],
        ctx_heads=ce["num_attention_heads"], ctx_mlp=ce["mlp_dim"],
        lang_tokens=ge.get("lang_tokens", 32), lang_dim=ge.get("lang_dim", 768),
        scale_context=bool(h.get("scale_context_embedding", False)))


# --------------------------------------------------------------------------------------
# Shared (not generated) leaves: the DINOv2 image encoder inside the base net
# (transformers FlaxDinov2Module param tree).  In the HN checkpoint each is a
# flat vector named "encoder_image_encoder_<path joined by _>" (hypernetwork.py:88-97).
# --------------------------------------------------------------------------------------
def encoder_leaves(g: Geometry) -> List[Tuple[Tuple[str, ...], Tuple[int, ...]]]:
    E, F, p = g.enc_dim, g.enc_mlp, g.patch
    out: List[Tuple[Tuple[str, ...], Tuple[int, ...]]] = [
        (("embeddings", "cls_token"), (1, 1, E)),
        (("embeddings", "mask_token"), (1, E)),
        (("embeddings", "patch_embeddings", "projection", "bias"), (E,)),
        (("embeddings", "patch_embeddings", "projection", "kernel"), (p, p, 3, E)),
        # baked to the run-time grid at conversion time (never interpolated on device)
        (("embeddings", "position_embeddings"), (1, g.patches + 1, E)),
    ]
    for i in range(g.enc_layers):
        L = ("encoder", "layer", str(i))
        for nm in ("key", "query", "value"):
            out += [(L + ("attention", "attention", nm, "bias"), (E,)),
                    (L + ("attention", "attention", nm, "kernel"), (E, E))]
        out += [(L + ("attention", "output", "dense", "bias"), (E,)),
                (L + ("attention", "output", "dense", "kernel"), (E, E)),
                (L + ("layer_scale1", "lambda1"), (E,)),
                (L + ("layer_scale2", "lambda1"), (E,)),
                (L + ("mlp", "fc1", "bias"), (F,)), (L + ("mlp", "fc1", "kernel"), (E, F)),
                (L + ("mlp", "fc2", "bias"), (E,)), (L + ("mlp", "fc2", "kernel"), (F, E)),
                (L + ("norm1", "bias"), (E,)), (L + ("norm1", "scale"), (E,)),
                (L + ("norm2", "bias"), (E,)), (L + ("norm2", "scale"), (E,))]
    out += [(("layernorm", "bias"), (E,)), (("layernorm", "scale"), (E,))]
    return out


def shared_name(path: Tuple[str, ...]) -> str:
    return "encoder_image_encoder_" + "_".join(path)


def hypernet_param_shapes(g: Geometry) -> Dict[str, Tuple[int, ...]]:
    """Every tensor of the HN checkpoint, keyed by '/'-joined flax path."""
    C, Hc, hc = g.ctx_dim, g.ctx_heads, g.ctx_dim // g.ctx_heads
    s: Dict[str, Tuple[int, ...]] = {
        "task_token_projection/kernel": (g.lang_dim, C), "task_token_projection/bias": (C,),
        "initial_image_projection/kernel": (g.enc_dim, C), "initial_image_projection/bias": (C,),
        "task_pos_embedding": (1, g.lang_tokens, C),
        "initial_image_pos_embedding": (1, 1, C),
        "layer_pos_embedding": (1, 1, C),
    }
    for l in range(g.ctx_layers):
        b = f"Transformer_0/encoderblock_{l}/"
        for ln in ("LayerNorm_0", "LayerNorm_1"):
            s[b + ln + "/scale"] = (C,)
            s[b + ln + "/bias"] = (C,)
        a = b + "MultiHeadDotProductAttention_0/"
        for nm in ("query", "key", "value"):
            s[a + nm + "/kernel"] = (C, Hc, hc)
            s[a + nm + "/bias"] = (Hc, hc)
        s[a + "out/kernel"] = (Hc, hc, C)
        s[a + "out/bias"] = (C,)
        s[b + "MlpBlock_0/Dense_0/kernel"] = (C, g.ctx_mlp)
        s[b + "MlpBlock_0/Dense_0/bias"] = (g.ctx_mlp,)
        s[b + "MlpBlock_0/Dense_1/kernel"] = (g.ctx_mlp, C)
        s[b + "MlpBlock_0/Dense_1/bias"] = (C,)
    s["Transformer_0/encoder_norm/scale"] = (C,)
    s["Transformer_0/encoder_norm/bias"] = (C,)
    for lf in generated_leaves(g):
        s[lf.head_name + "/kernel"] = (C, lf.size)
        s[lf.head_name + "/bias"] = (lf.size,)
    for path, shape in encoder_leaves(g):
        s[shared_name(path)] = (int(np.prod(shape)),)
    return s
