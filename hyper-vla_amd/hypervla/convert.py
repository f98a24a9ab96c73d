"""Checkpoint conversion for the built path: reference artefacts -> ``params_<step>.npz``.

What can be read without JAX / Orbax in the process:

* ``EMA_params.pkl`` (``scripts/train.py:697-699``: ``pickle.dump({"EMA_0.999": params})`` of **jax arrays**) — the
  file the evaluators actually load (``data/simpler/evaluate.py:440-444``).  jax pickles an array as
  ``jax._src.array._reconstruct_array(numpy_reduce_fn, numpy_reduce_args, array_state, aval_state)``; the shim
  unpickler below rebuilds the numpy value and drops the device placement, so neither jax nor jaxlib is needed.
* a flax parameter tree already on the host (nested dict of arrays; e.g. ``orbax`` restore output handed over by a
  process that has JAX) — :func:`params_from_tree`.
* the pretrained DINOv2 weights as a Hugging Face **torch** ``state_dict`` (``facebook/dinov2-base``) —
  :func:`dinov2_from_hf_state_dict`, including the position-embedding table baked from HF's 37 x 37 grid to the run-time
  16 x 16 grid exactly as ``FlaxDinov2Embeddings.interpolate_pos_encoding`` computes it on every call
  (``jax.image.scale_and_translate(method="bicubic", antialias=False)`` with the ``+0.1`` size offset).

Orbax step directories themselves (OCDBT / tensorstore) are not parsed here; export them where JAX exists with
``tools/export_reference_checkpoint.py``.
"""
from __future__ import annotations

import io
import json
import os
import pickle
import shutil
from typing import Any, Dict, Mapping, Optional

import numpy as np

from .config import Geometry, encoder_leaves, geometry_from_config, hypernet_param_shapes, shared_name


# ------------------------------------------------------------------------------------------------ pickles of jax arrays
def _reconstruct_array(fun, args, arr_state, aval_state):
    """Stand-in for ``jax._src.array._reconstruct_array``: the first three arguments are numpy's own pickle recipe."""
    value = fun(*args)
    value.__setstate__(arr_state)
    return value


class _JaxFreeUnpickler(pickle.Unpickler):
    def find_class(self, module, name):
        if name == "_reconstruct_array" and module.startswith("jax"):
            return _reconstruct_array
        if module.startswith("jax") or module.startswith("jaxlib"):
            raise pickle.UnpicklingError(f"unsupported jax object in pickle: {module}.{name}")
        return super().find_class(module, name)


def load_jax_pickle(path_or_bytes) -> Any:
    """``pickle.load`` for files that contain jax arrays, without jax."""
    if isinstance(path_or_bytes, (bytes, bytearray)):
        return _JaxFreeUnpickler(io.BytesIO(path_or_bytes)).load()
    with open(path_or_bytes, "rb") as f:
        return _JaxFreeUnpickler(f).load()


# ------------------------------------------------------------------------------------------------ flax msgpack (example_batch)
# `example_batch.msgpack` is written with flax.serialization.msgpack_serialize (hypervla/model.py:270-274) and read back in
# load_pretrained (:165-169).  The wire format (flax/serialization.py, flax 0.8.1): plain msgpack maps / lists / scalars;
# a numpy or jax array is ExtType 1 whose payload is msgpack((shape, dtype.name, raw bytes)); a numpy scalar is ExtType 3
# with the payload of a 0-d array; a Python complex is ExtType 2 (packed (real, imag)); arrays above ~2 GB are split into
# a {"__msgpack_chunked_array__": True, "shape": ..., "chunks": [...]} map.  No flax / jax needed.
_EXT_NDARRAY, _EXT_COMPLEX, _EXT_NPSCALAR = 1, 2, 3


def _unpack_ndarray(payload: bytes) -> np.ndarray:
    import msgpack
    shape, dtype_name, buf = msgpack.unpackb(payload, raw=True)
    dtype_name = dtype_name.decode() if isinstance(dtype_name, bytes) else dtype_name
    if dtype_name == "bfloat16":                       # not a numpy dtype: widen to float32
        u = np.frombuffer(buf, dtype=np.uint16).astype(np.uint32) << 16
        return u.view(np.float32).reshape(shape)
    return np.frombuffer(buf, dtype=np.dtype(dtype_name)).reshape(shape).copy()


def _ext_hook(code: int, data: bytes):
    import msgpack
    if code == _EXT_NDARRAY:
        return _unpack_ndarray(data)
    if code == _EXT_NPSCALAR:
        return _unpack_ndarray(data)[()]
    if code == _EXT_COMPLEX:
        re, im = msgpack.unpackb(data)
        return complex(re, im)
    return msgpack.ExtType(code, data)


def _unchunk(tree):
    if isinstance(tree, dict):
        if tree.get("__msgpack_chunked_array__"):
            chunks = [tree["chunks"][str(i)] if isinstance(tree["chunks"], dict) else tree["chunks"][i]
                      for i in range(len(tree["chunks"]))]
            shape = tree["shape"]                       # flax writes tuples through _tuple_to_dict: {"0": d0, "1": d1, ...}
            if isinstance(shape, dict):
                shape = tuple(int(shape[str(i)]) for i in range(len(shape)))
            return np.concatenate([np.asarray(c).reshape(-1) for c in chunks]).reshape(shape)
        return {k: _unchunk(v) for k, v in tree.items()}
    return tree


def msgpack_restore(data: bytes) -> Any:
    """`flax.serialization.msgpack_restore` without flax: bytes -> nested dict of numpy arrays / Python scalars."""
    import msgpack
    return _unchunk(msgpack.unpackb(data, ext_hook=_ext_hook, raw=False, strict_map_key=False))


def msgpack_serialize(tree: Any) -> bytes:
    """`flax.serialization.msgpack_serialize` without flax (arrays below the 2 GB chunking threshold)."""
    import msgpack

    def default(o):
        if isinstance(o, np.ndarray):
            return msgpack.ExtType(_EXT_NDARRAY, msgpack.packb((list(o.shape), o.dtype.name, o.tobytes()), use_bin_type=True))
        if isinstance(o, np.generic):
            a = np.asarray(o)
            return msgpack.ExtType(_EXT_NPSCALAR, msgpack.packb((list(a.shape), a.dtype.name, a.tobytes()), use_bin_type=True))
        if isinstance(o, complex):
            return msgpack.ExtType(_EXT_COMPLEX, msgpack.packb((o.real, o.imag)))
        raise TypeError(f"cannot serialise {type(o)}")

    def prep(t):                      # torch tensors / anything array-like -> numpy
        if isinstance(t, dict):
            return {str(k): prep(v) for k, v in t.items()}
        if isinstance(t, (list, tuple)):
            return [prep(v) for v in t]
        if isinstance(t, (np.ndarray, np.generic, int, float, bool, str, bytes, complex)) or t is None:
            return t
        return np.asarray(t)

    return msgpack.packb(prep(tree), default=default, use_bin_type=True)


def load_example_batch(checkpoint_path: str) -> Optional[Dict[str, Any]]:
    """The `example_batch` of a checkpoint directory (hypervla/model.py:165-169), or None when the file is absent.  As the
    reference does (:190-192), a missing language `token_embedding` is added as zeros [B, T, 768]."""
    fp = os.path.join(checkpoint_path, "example_batch.msgpack")
    if not os.path.exists(fp):
        return None
    with open(fp, "rb") as f:
        eb = msgpack_restore(f.read())
    try:
        li = eb["task"]["language_instruction"]
        if "token_embedding" not in li:
            li["token_embedding"] = np.zeros(tuple(np.shape(li["input_ids"])) + (768,), np.float32)
    except (KeyError, TypeError):
        pass
    return eb


def load_ema_pickle(path: str, coefficient: float = 0.999) -> Dict[str, Any]:
    """The tree the evaluators swap in with ``model.replace(params=EMA_params[f"EMA_{coefficient}"])``."""
    trees = load_jax_pickle(path)
    key = f"EMA_{coefficient}"
    if key not in trees:
        raise KeyError(f"{key} not in {sorted(trees)}")
    return trees[key]


# ------------------------------------------------------------------------------------------------ flax tree -> flat names
def flatten_tree(tree: Mapping, sep: str = "/") -> Dict[str, np.ndarray]:
    """``flax.traverse_util.flatten_dict(tree, sep=sep)`` for nested dicts (FrozenDict iterates the same way)."""
    out: Dict[str, np.ndarray] = {}

    def walk(node, prefix):
        if isinstance(node, Mapping) or hasattr(node, "items"):
            for k, v in node.items():
                walk(v, prefix + [str(k)])
        else:
            out[sep.join(prefix)] = np.asarray(node)

    walk(tree, [])
    return out


def params_from_tree(tree: Mapping, g: Geometry) -> Dict[str, np.ndarray]:
    """Hypernetwork parameter tree (SURVEY.md §5.4 naming) -> the flat float32 dict `HyperVLA(params=...)` takes.
    Every tensor the built path needs must be present with the expected shape; tensors the path does not use are
    reported, not silently dropped."""
    flat = flatten_tree(tree)
    if flat and all(k.startswith("params/") for k in flat):             # a variables dict {"params": ...}
        flat = {k[len("params/"):]: v for k, v in flat.items()}
    shapes = hypernet_param_shapes(g)
    missing = sorted(set(shapes) - set(flat))
    if missing:
        raise KeyError(f"{len(missing)} tensors missing from the checkpoint, e.g. {missing[:4]}")
    extra = sorted(set(flat) - set(shapes))
    if extra:
        raise ValueError(f"{len(extra)} tensors of the checkpoint are outside the built path (other generation strategy / "
                         f"encoder?), e.g. {extra[:4]}")
    out = {}
    for k, shp in shapes.items():
        v = np.asarray(flat[k], np.float32)
        if k.endswith("embeddings_position_embeddings") and v.size != int(np.prod(shp)):
            # the checkpoint carries HF's 37 x 37 table; the reference resizes it inside every forward pass
            v = bake_position_embeddings(v.reshape(1, -1, g.enc_dim), g.image_size // g.patch)
        if int(np.prod(v.shape)) != int(np.prod(shp)):
            raise ValueError(f"{k}: checkpoint shape {v.shape} != expected {tuple(shp)}")
        out[k] = v.reshape(shp)
    return out


# ------------------------------------------------------------------------------------------------ DINOv2 position table
def _cubic_kernel(x: np.ndarray) -> np.ndarray:
    """Keys cubic convolution kernel, a = -0.5 (jax.image 'bicubic' == 'cubic'; torch's bicubic uses a = -0.75)."""
    x = np.abs(x)
    out = ((1.5 * x - 2.5) * x) * x + 1.0
    out = np.where(x >= 1.0, ((-0.5 * x + 2.5) * x - 4.0) * x + 2.0, out)
    return np.where(x >= 2.0, 0.0, out)


def _scale_and_translate_weights(in_size: int, out_size: int, scale: float, translation: float = 0.0) -> np.ndarray:
    """[in_size, out_size] weight matrix of jax.image.scale_and_translate for one spatial axis, antialias=False
    (`jax/_src/image/scale.py::compute_weight_mat`): sample position of output i is (i + 0.5) / scale - t / scale - 0.5,
    weights renormalised over the taps that exist, zero for samples outside [-0.5, in_size - 0.5]."""
    inv = np.float32(1.0) / np.float32(scale)
    sample_f = (np.arange(out_size, dtype=np.float32) + np.float32(0.5)) * inv - np.float32(translation) * inv - np.float32(0.5)
    x = np.abs(sample_f[None, :] - np.arange(in_size, dtype=np.float32)[:, None])      # kernel_scale = 1 without antialias
    w = _cubic_kernel(x.astype(np.float32)).astype(np.float32)
    total = w.sum(axis=0, keepdims=True)
    w = np.where(np.abs(total) > 1000.0 * np.finfo(np.float32).eps, w / np.where(total != 0, total, 1), 0.0)
    inside = (sample_f >= -0.5) & (sample_f <= in_size - 0.5)
    return np.where(inside[None, :], w, 0.0).astype(np.float32)


def bake_position_embeddings(table: np.ndarray, grid: int) -> np.ndarray:
    """HF ``embeddings.position_embeddings`` [1, 1 + n*n, E] -> [1, 1 + grid*grid, E] as
    ``FlaxDinov2Embeddings.interpolate_pos_encoding`` produces it for a (14*grid)^2 image: the patch part is resized
    with scale (grid + 0.1) / n per axis, translation 0, bicubic, no antialiasing, in float32; the class row is kept."""
    table = np.asarray(table, np.float32)
    if table.ndim == 2:
        table = table[None]
    n2, E = table.shape[1] - 1, table.shape[2]
    n = int(round(np.sqrt(n2)))
    if n * n != n2:
        raise ValueError(f"position table has {n2} patch rows, not a square")
    if n == grid:
        return table.copy()
    w = _scale_and_translate_weights(n, grid, np.float32((grid + 0.1) / n))           # same for rows and columns
    patch = table[0, 1:].reshape(n, n, E)
    # jax contracts the height axis first, then the width axis (spatial_dims order), each in float32
    tmp = np.einsum("hwe,hi->iwe", patch, w, dtype=np.float32)
    out = np.einsum("iwe,wj->ije", tmp, w, dtype=np.float32)
    return np.concatenate([table[:, :1], out.reshape(1, grid * grid, E)], axis=1)


def tree_from_params(params: Mapping[str, np.ndarray]) -> Dict[str, Any]:
    """Inverse of :func:`params_from_tree`: the flat '/'-named dict -> the nested hypernetwork parameter tree of the
    reference (`HyperVLA.params`, hypervla/model.py:330-346): modules nest by '/', the shared image-encoder leaves are flat
    vectors under their one-level names `encoder_image_encoder_<path>`.  Values keep their shapes (the shared leaves are
    ravelled); reshape against the target tree's leaves where the reference's shapes are at hand
    (tools/make_reference_fixtures.py does)."""
    tree: Dict[str, Any] = {}
    for name, v in params.items():
        v = np.asarray(v, np.float32)
        if name.startswith("encoder_image_encoder_"):
            v = v.reshape(-1)
        node = tree
        keys = name.split("/")
        for k in keys[:-1]:
            node = node.setdefault(k, {})
        node[keys[-1]] = v
    return tree


# ------------------------------------------------------------------------------------------------ HF torch DINOv2 -> shared leaves
def dinov2_from_hf_state_dict(sd: Mapping[str, Any], g: Geometry) -> Dict[str, np.ndarray]:
    """``Dinov2Model.state_dict()`` (torch layout) -> the checkpoint's shared leaves: flat float32 vectors named
    ``encoder_image_encoder_<path>`` in flax layout (Linear weights transposed to [in, out], conv OIHW -> HWIO),
    position table baked to the geometry's grid (`hypervla/model.py:330-346,543-565`)."""
    def arr(name):
        v = sd[name]
        v = v.detach().cpu().numpy() if hasattr(v, "detach") else np.asarray(v)
        return np.asarray(v, np.float32)

    out: Dict[str, np.ndarray] = {}

    def put(path, v, shape):
        v = np.ascontiguousarray(v, np.float32)
        if tuple(v.shape) != tuple(shape):
            raise ValueError(f"{'/'.join(path)}: {v.shape} != {tuple(shape)}")
        out[shared_name(path)] = v.reshape(-1)

    shapes = dict(encoder_leaves(g))
    E = g.enc_dim
    put(("embeddings", "cls_token"), arr("embeddings.cls_token"), shapes[("embeddings", "cls_token")])
    put(("embeddings", "mask_token"), arr("embeddings.mask_token"), shapes[("embeddings", "mask_token")])
    put(("embeddings", "position_embeddings"), bake_position_embeddings(arr("embeddings.position_embeddings"), g.image_size // g.patch),
        shapes[("embeddings", "position_embeddings")])
    put(("embeddings", "patch_embeddings", "projection", "kernel"),
        arr("embeddings.patch_embeddings.projection.weight").transpose(2, 3, 1, 0), (g.patch, g.patch, 3, E))
    put(("embeddings", "patch_embeddings", "projection", "bias"), arr("embeddings.patch_embeddings.projection.bias"), (E,))
    for i in range(g.enc_layers):
        L, T = ("encoder", "layer", str(i)), f"encoder.layer.{i}."
        for nm in ("norm1", "norm2"):
            put(L + (nm, "scale"), arr(T + nm + ".weight"), (E,))
            put(L + (nm, "bias"), arr(T + nm + ".bias"), (E,))
        for nm in ("query", "key", "value"):
            put(L + ("attention", "attention", nm, "kernel"), arr(T + f"attention.attention.{nm}.weight").T, (E, E))
            put(L + ("attention", "attention", nm, "bias"), arr(T + f"attention.attention.{nm}.bias"), (E,))
        put(L + ("attention", "output", "dense", "kernel"), arr(T + "attention.output.dense.weight").T, (E, E))
        put(L + ("attention", "output", "dense", "bias"), arr(T + "attention.output.dense.bias"), (E,))
        put(L + ("layer_scale1", "lambda1"), arr(T + "layer_scale1.lambda1"), (E,))
        put(L + ("layer_scale2", "lambda1"), arr(T + "layer_scale2.lambda1"), (E,))
        put(L + ("mlp", "fc1", "kernel"), arr(T + "mlp.fc1.weight").T, (E, g.enc_mlp))
        put(L + ("mlp", "fc1", "bias"), arr(T + "mlp.fc1.bias"), (g.enc_mlp,))
        put(L + ("mlp", "fc2", "kernel"), arr(T + "mlp.fc2.weight").T, (g.enc_mlp, E))
        put(L + ("mlp", "fc2", "bias"), arr(T + "mlp.fc2.bias"), (E,))
    put(("layernorm", "scale"), arr("layernorm.weight"), (E,))
    put(("layernorm", "bias"), arr("layernorm.bias"), (E,))
    assert set(out) == {shared_name(p) for p in shapes}
    return out


# ------------------------------------------------------------------------------------------------ HF torch T5 -> flax names
def t5_from_hf_state_dict(sd: Mapping[str, Any], t) -> Dict[str, np.ndarray]:
    """``T5EncoderModel.state_dict()`` ("t5-base", torch) -> the FlaxT5EncoderModel tree `load_language_encoder` takes
    ('/'-joined names, Linear weights transposed to [in, out]; octo/utils/train_utils.py:542-568 `hf_weights_loader`)."""
    from .config import t5_param_shapes

    def arr(name):
        v = sd[name]
        v = v.detach().cpu().numpy() if hasattr(v, "detach") else np.asarray(v)
        return np.asarray(v, np.float32)

    out = {"shared/embedding": arr("shared.weight"), "encoder/final_layer_norm/weight": arr("encoder.final_layer_norm.weight"),
           "encoder/block/0/layer/0/SelfAttention/relative_attention_bias/embedding":
               arr("encoder.block.0.layer.0.SelfAttention.relative_attention_bias.weight")}
    for i in range(t.layers):
        b, T = f"encoder/block/{i}/layer/", f"encoder.block.{i}.layer."
        for nm in "qkvo":
            out[b + f"0/SelfAttention/{nm}/kernel"] = np.ascontiguousarray(arr(T + f"0.SelfAttention.{nm}.weight").T)
        out[b + "0/layer_norm/weight"] = arr(T + "0.layer_norm.weight")
        out[b + "1/DenseReluDense/wi/kernel"] = np.ascontiguousarray(arr(T + "1.DenseReluDense.wi.weight").T)
        out[b + "1/DenseReluDense/wo/kernel"] = np.ascontiguousarray(arr(T + "1.DenseReluDense.wo.weight").T)
        out[b + "1/layer_norm/weight"] = arr(T + "1.layer_norm.weight")
    shapes = t5_param_shapes(t)
    if set(out) != set(shapes):
        raise KeyError(sorted(set(out) ^ set(shapes))[:4])
    for k, shp in shapes.items():
        if tuple(out[k].shape) != tuple(shp):
            raise ValueError(f"{k}: {out[k].shape} != {tuple(shp)}")
    return out


# ------------------------------------------------------------------------------------------------ whole checkpoint directories
def convert_checkpoint(src_dir: str, dst_dir: str, step: int, ema: Optional[float] = 0.999,
                       tree: Optional[Mapping] = None) -> str:
    """Reference run directory -> a directory `HyperVLA.load_pretrained` reads.

    Parameters come from `tree` when given (a host-side flax tree), else from ``<src_dir>/<step>/EMA_params.pkl``.
    ``dataset_statistics.json`` and ``example_batch.msgpack`` are copied as they are (`hypervla/model.py:260-284`);
    ``config.json`` is written with the `action_head_kwargs` defaults of `hypervla/model.py:157-163` filled in and, when the
    position table was resized, the key `position_embeddings_baked_from`.  Converting in place would overwrite the
    reference run's own ``config.json`` with that augmented one, so `dst_dir` must differ from `src_dir`."""
    if os.path.abspath(src_dir) == os.path.abspath(dst_dir):
        raise ValueError("convert_checkpoint: dst_dir must not be the reference run directory itself "
                         "(its config.json would be overwritten)")
    with open(os.path.join(src_dir, "config.json")) as f:
        config = json.load(f)
    if "action_head_kwargs" not in config["base_net_kwargs"]:            # hypervla/model.py:157-163
        config["base_net_kwargs"]["action_head_kwargs"] = dict(
            token_per_horizon=False, squash_continuous_action=True, clip_target=False, max_action=5.0)
    g = geometry_from_config(config)
    if tree is None:
        if ema is None:
            raise ValueError("pass the restored parameter tree, or ema=<coefficient> to read EMA_params.pkl")
        tree = load_ema_pickle(os.path.join(src_dir, str(step), "EMA_params.pkl"), ema)
    params = params_from_tree(tree, g)
    os.makedirs(dst_dir, exist_ok=True)
    out = os.path.join(dst_dir, f"params_{step}.npz")
    np.savez(out, **params)
    for name in ("dataset_statistics.json", "example_batch.msgpack"):
        sp = os.path.join(src_dir, name)
        if os.path.exists(sp):
            shutil.copyfile(sp, os.path.join(dst_dir, name))
    # the reference keeps HF's 37 x 37 DINOv2 position table and resizes it inside every forward pass; here it is baked to
    # the run-time grid once.  The config says so: FineTuner refuses to train a baked table unless told to
    flat = flatten_tree(tree)
    pos = [v for k, v in flat.items() if k.endswith("embeddings_position_embeddings")]
    rows = int(np.asarray(pos[0]).size // g.enc_dim) - 1 if pos else g.patches
    if rows != g.patches:
        n = int(round(np.sqrt(rows)))
        config["position_embeddings_baked_from"] = [n, n]
    with open(os.path.join(dst_dir, "config.json"), "w") as f:
        json.dump(config, f)
    return out
