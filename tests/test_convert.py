"""Checkpoint conversion: jax-free EMA pickle reader, flax tree -> flat names, HF torch DINOv2
state_dict -> shared leaves, position-table baking."""
import pickle
import sys
import types

import numpy as np
import pytest

from hypervla import synthetic as syn
from hypervla.config import MID, TINY, default_config, encoder_leaves, generated_leaves, hypernet_param_shapes, shared_name
from hypervla import convert as cv


def _nest(flat):
    tree = {}
    for k, v in flat.items():
        node = tree
        parts = k.split("/")
        for p in parts[:-1]:
            node = node.setdefault(p, {})
        node[parts[-1]] = v
    return tree


class _FakeJaxArray:
    """Pickles exactly like jax's ArrayImpl.__reduce__: (jax._src.array._reconstruct_array, (fun, args, state, aval))."""

    def __init__(self, value):
        self.value = np.asarray(value)

    def __reduce__(self):
        fun, args, arr_state = self.value.__reduce__()
        return sys.modules["jax._src.array"]._reconstruct_array, (fun, args, arr_state, {"weak_type": False, "named_shape": {}})


@pytest.fixture
def fake_jax():
    """A module named like jax's so that pickle.dumps emits the real GLOBAL opcode; removed before loading."""
    names = ["jax", "jax._src", "jax._src.array"]
    saved = {n: sys.modules.get(n) for n in names}
    mods = {n: types.ModuleType(n) for n in names}

    def _reconstruct_array(*a):                      # never called: the reader must not need it
        raise AssertionError("the jax-free reader called into jax")

    _reconstruct_array.__module__, _reconstruct_array.__qualname__ = "jax._src.array", "_reconstruct_array"
    mods["jax._src.array"]._reconstruct_array = _reconstruct_array
    sys.modules.update(mods)
    yield
    for n in names:
        if saved[n] is None:
            sys.modules.pop(n, None)
        else:
            sys.modules[n] = saved[n]


def test_ema_pickle_without_jax(fake_jax, tmp_path):
    g = TINY
    P = syn.synthetic_params(g)
    tree = _nest({k: _FakeJaxArray(v) for k, v in P.items()})
    blob = pickle.dumps({"EMA_0.999": tree})
    assert b"_reconstruct_array" in blob and b"jax._src.array" in blob
    for n in ("jax", "jax._src", "jax._src.array"):
        sys.modules.pop(n, None)                     # the reading process has no jax at all
    (tmp_path / "7").mkdir()
    (tmp_path / "7" / "EMA_params.pkl").write_bytes(blob)
    got = cv.params_from_tree(cv.load_ema_pickle(str(tmp_path / "7" / "EMA_params.pkl"), 0.999), g)
    assert set(got) == set(P)
    for k in P:
        np.testing.assert_array_equal(got[k], P[k])
    with pytest.raises(KeyError):
        cv.load_ema_pickle(str(tmp_path / "7" / "EMA_params.pkl"), 0.99)


def test_convert_checkpoint_directory_round_trip(fake_jax, tmp_path):
    import json
    from hypervla.model import HyperVLA
    g = TINY
    P = syn.synthetic_params(g)
    src, dst = tmp_path / "run", tmp_path / "out"
    (src / "100").mkdir(parents=True)
    (src / "config.json").write_text(json.dumps(default_config(g)))
    stats = syn.synthetic_dataset_statistics(g)
    (src / "dataset_statistics.json").write_text(json.dumps(
        {k: {kk: {n: np.asarray(a).tolist() for n, a in vv.items()} for kk, vv in v.items()} for k, v in stats.items()}))
    (src / "100" / "EMA_params.pkl").write_bytes(pickle.dumps({"EMA_0.999": _nest({k: _FakeJaxArray(v) for k, v in P.items()})}))
    for n in ("jax", "jax._src", "jax._src.array"):
        sys.modules.pop(n, None)
    out = cv.convert_checkpoint(str(src), str(dst), 100, ema=0.999)
    with np.load(out) as z:
        assert set(z.files) == set(P)
        np.testing.assert_array_equal(z["task_token_projection/kernel"], P["task_token_projection/kernel"])
    assert (dst / "config.json").exists() and (dst / "dataset_statistics.json").exists()


def test_converter_records_a_baked_position_table(fake_jax, tmp_path):
    """A checkpoint that carries HF's un-resized position table (the reference resizes it in every forward pass,
    SURVEY App. A) is baked to the run-time grid and the converted config says so; example_batch.msgpack is carried over."""
    import json
    g = TINY                                                    # 4 x 4 patches
    P = syn.synthetic_params(g)
    key = "encoder_image_encoder_embeddings_position_embeddings"
    big = np.random.default_rng(0).standard_normal((1, 1 + 9 * 9, g.enc_dim)).astype(np.float32)
    tree = _nest({k: (big.reshape(-1) if k == key else v) for k, v in P.items()})
    src, dst = tmp_path / "run", tmp_path / "out"
    src.mkdir()
    (src / "config.json").write_text(json.dumps(default_config(g)))
    (src / "example_batch.msgpack").write_bytes(cv.msgpack_serialize({"task": {"language_instruction": {"input_ids": np.zeros((1, 8), np.int64)}}}))
    out = cv.convert_checkpoint(str(src), str(dst), 3, tree=tree)
    with np.load(out) as z:
        np.testing.assert_allclose(z[key].reshape(1, 17, g.enc_dim), cv.bake_position_embeddings(big, 4))
    assert json.loads((dst / "config.json").read_text())["position_embeddings_baked_from"] == [9, 9]
    assert cv.load_example_batch(str(dst))["task"]["language_instruction"]["input_ids"].shape == (1, 8)
    # a table that already has the run-time grid is not marked
    out2 = cv.convert_checkpoint(str(src), str(tmp_path / "out2"), 3, tree=_nest(P))
    assert "position_embeddings_baked_from" not in json.loads((tmp_path / "out2" / "config.json").read_text())


def test_tree_errors_name_the_problem():
    g = TINY
    P = syn.synthetic_params(g)
    bad = dict(P)
    bad.pop("layer_pos_embedding")
    with pytest.raises(KeyError, match="missing"):
        cv.params_from_tree(_nest(bad), g)
    extra = dict(P, **{"language_encoder/kernel": np.zeros(3, np.float32)})
    with pytest.raises(ValueError, match="outside the built path"):
        cv.params_from_tree(_nest(extra), g)
    wrong = dict(P, **{"task_token_projection/bias": np.zeros(3, np.float32)})
    with pytest.raises(ValueError, match="task_token_projection/bias"):
        cv.params_from_tree(_nest(wrong), g)
    np.testing.assert_array_equal(cv.params_from_tree({"params": _nest(P)}, g)["layer_pos_embedding"], P["layer_pos_embedding"])


def test_position_table_baking_properties():
    """Bicubic (Keys a = -0.5) resize of the n x n table to the run-time grid with scale (grid + 0.1) / n."""
    rng = np.random.default_rng(0)
    n, grid, E = 37, 16, 8
    table = rng.standard_normal((1, 1 + n * n, E)).astype(np.float32)
    out = cv.bake_position_embeddings(table, grid)
    assert out.shape == (1, 1 + grid * grid, E) and out.dtype == np.float32
    np.testing.assert_array_equal(out[:, 0], table[:, 0])                          # class row untouched
    np.testing.assert_array_equal(cv.bake_position_embeddings(table, n), table)      # same grid: returned as is
    w = cv._scale_and_translate_weights(n, grid, np.float32((grid + 0.1) / n))
    np.testing.assert_allclose(w.sum(0), 1.0, atol=1e-6)                           # every output is an affine combination
    assert (np.count_nonzero(w, axis=0) <= 4).all()                                # 4-tap cubic
    # cubic convolution reproduces polynomials of degree <= 2 away from the borders: resample f(y, x) = 2 + 3y - x
    yy, xx = np.meshgrid(np.arange(n, dtype=np.float32), np.arange(n, dtype=np.float32), indexing="ij")
    lin = (2 + 3 * yy - xx).reshape(1, n * n, 1)
    got = cv.bake_position_embeddings(np.concatenate([np.zeros((1, 1, 1), np.float32), lin], 1), grid)[0, 1:, 0].reshape(grid, grid)
    pos = (np.arange(grid, dtype=np.float64) + 0.5) * n / (grid + 0.1) - 0.5
    want = 2 + 3 * pos[:, None] - pos[None, :]
    inner = (pos >= 1) & (pos <= n - 2)
    np.testing.assert_allclose(got[np.ix_(inner, inner)], want[np.ix_(inner, inner)], rtol=0, atol=2e-4)
    # constant tables stay constant everywhere (weights renormalised at the border)
    const = cv.bake_position_embeddings(np.full((1, 1 + n * n, 2), 0.25, np.float32), grid)
    np.testing.assert_allclose(const, 0.25, atol=1e-6)


def test_hf_state_dict_round_trip():
    """build the torch Dinov2Model from the checkpoint's shared leaves (oracle), read its state_dict back."""
    torch = pytest.importorskip("torch")
    pytest.importorskip("transformers")
    from oracle import hvla_ref_torch as ot
    g = MID
    P = syn.synthetic_params(g)
    model = ot.build_hf_dinov2(P, g, dict(encoder_leaves(g)))
    got = cv.dinov2_from_hf_state_dict(model.state_dict(), g)
    for path, shape in encoder_leaves(g):
        k = shared_name(path)
        np.testing.assert_array_equal(got[k], P[k].reshape(-1), err_msg=k)
    # a 3x3 source table is baked to the geometry's grid on the way
    sd = dict(model.state_dict())
    sd["embeddings.position_embeddings"] = torch.randn(1, 1 + 9, g.enc_dim)
    baked = cv.dinov2_from_hf_state_dict(sd, g)[shared_name(("embeddings", "position_embeddings"))]
    assert baked.size == (g.patches + 1) * g.enc_dim


def test_t5_state_dict_names():
    torch = pytest.importorskip("torch")
    tr = pytest.importorskip("transformers")
    from hypervla.config import T5_TINY as t, t5_param_shapes
    cfg = tr.T5Config(vocab_size=t.vocab, d_model=t.d_model, d_kv=t.d_kv, d_ff=t.d_ff, num_layers=t.layers, num_heads=t.heads,
                      feed_forward_proj="relu")
    m = tr.T5EncoderModel(cfg)
    got = cv.t5_from_hf_state_dict(m.state_dict(), t)
    assert set(got) == set(t5_param_shapes(t))
    np.testing.assert_array_equal(got["encoder/block/1/layer/1/DenseReluDense/wi/kernel"],
                                  m.state_dict()["encoder.block.1.layer.1.DenseReluDense.wi.weight"].numpy().T)


def test_example_batch_msgpack_without_flax(tmp_path):
    """`example_batch.msgpack` (hypervla/model.py:165-169,270-274) in flax's msgpack dialect: arrays are ExtType 1 with the
    payload msgpack((shape, dtype name, raw bytes)).  The first literal below is that layout written out by hand for
    np.array([1, 2, 3], int32) inside {"a": ...}; then a round trip of an OXE-shaped batch, and the token_embedding the
    reference adds when the file lacks it (:190-192)."""
    from hypervla.convert import load_example_batch, msgpack_restore, msgpack_serialize
    wire = bytes.fromhex("81" "a161"                      # map of 1: "a"
                         "c7" "17" "01"                   # ext8, 23 payload bytes, type 1 (ndarray)
                         "93" "91" "03"                   # (shape = [3],
                         "a5" "696e743332"                #  "int32",
                         "c4" "0c" "010000000200000003000000")   # bin8, 12 bytes)
    got = msgpack_restore(wire)
    assert got["a"].dtype == np.int32 and got["a"].tolist() == [1, 2, 3]
    assert msgpack_serialize({"a": np.array([1, 2, 3], np.int32)}) == wire
    eb = {"observation": {"image_primary": np.arange(2 * 1 * 4 * 4 * 3, dtype=np.uint8).reshape(2, 1, 4, 4, 3),
                          "timestep_pad_mask": np.ones((2, 1), bool)},
          "task": {"language_instruction": {"input_ids": np.arange(64).reshape(2, 32), "attention_mask": np.ones((2, 32), np.int64)}},
          "initial_state": {"image_primary": np.zeros((2, 1, 4, 4, 3), np.uint8)},
          "action": np.linspace(-1, 1, 2 * 1 * 4 * 7, dtype=np.float32).reshape(2, 1, 4, 7), "dataset_name": "bridge_dataset"}
    (tmp_path / "example_batch.msgpack").write_bytes(msgpack_serialize(eb))
    back = load_example_batch(str(tmp_path))
    np.testing.assert_array_equal(back["observation"]["image_primary"], eb["observation"]["image_primary"])
    np.testing.assert_array_equal(back["action"], eb["action"])
    assert back["observation"]["timestep_pad_mask"].dtype == bool and back["dataset_name"] == "bridge_dataset"
    assert back["task"]["language_instruction"]["token_embedding"].shape == (2, 32, 768)
    assert load_example_batch(str(tmp_path / "nowhere")) is None
    # bfloat16 leaves (not a numpy dtype) come back as float32
    bf = msgpack_restore(bytes.fromhex("81a162" "c7" "10" "01" "93" "91" "01" "a8" + "bfloat16".encode().hex() + "c4" "02" "803f"))
    assert bf["b"].dtype == np.float32 and bf["b"].tolist() == [1.0]


def test_chunked_array_map_and_in_place_conversion(tmp_path):
    """Arrays above flax's chunk size are written as a map {"__msgpack_chunked_array__": True, "shape": {"0": d0, ...},
    "chunks": {"0": c0, ...}} (flax/serialization.py `_chunk`, tuples go through `_tuple_to_dict`); and converting a run
    directory onto itself is refused (it would overwrite the reference's config.json)."""
    import msgpack
    from hypervla.convert import _EXT_NDARRAY, convert_checkpoint, msgpack_restore
    a = np.arange(24, dtype=np.float32).reshape(2, 3, 4)
    flat = a.reshape(-1)

    def nd(x):
        return msgpack.ExtType(_EXT_NDARRAY, msgpack.packb((list(x.shape), x.dtype.name, x.tobytes()), use_bin_type=True))

    wire = msgpack.packb({"w": {"__msgpack_chunked_array__": True, "shape": {"0": 2, "1": 3, "2": 4},
                                "chunks": {"0": nd(flat[:10]), "1": nd(flat[10:20]), "2": nd(flat[20:])}}}, use_bin_type=True)
    np.testing.assert_array_equal(msgpack_restore(wire)["w"], a)
    wire_list = msgpack.packb({"w": {"__msgpack_chunked_array__": True, "shape": [2, 3, 4],
                                     "chunks": [nd(flat[:12]), nd(flat[12:])]}}, use_bin_type=True)
    np.testing.assert_array_equal(msgpack_restore(wire_list)["w"], a)
    (tmp_path / "config.json").write_text("{}")
    with pytest.raises(ValueError, match="dst_dir"):
        convert_checkpoint(str(tmp_path), str(tmp_path), 1)
