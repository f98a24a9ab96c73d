"""CPU: host-side mirror of the reference interface (config, leaf metadata, wrapper post-processing)."""
import json

import numpy as np
import pytest

from hypervla import synthetic as syn
from hypervla.config import (FULL, MID, SMALL_E, TINY, default_config, encoder_leaves, generated_leaves,
                             geometry_from_config, hypernet_param_shapes, total_generated)
from hypervla.interface import ActionEnsembler, InferenceWrapper, euler2axangle
from oracle import hvla_ref_np as onp


def test_leaf_table_matches_survey_appendix_b():
    lv = generated_leaves(FULL)
    assert len(lv) == 73 and total_generated(FULL) == 201_500
    names = [l.flat_name for l in lv]
    assert names == sorted(names, key=lambda n: [p for p in lv if p.flat_name == n][0].path)   # pytree order
    per_block = sum(l.size for l in lv if "encoderblock_2" in l.flat_name)
    assert per_block == 33_472
    assert lv[-1].flat_name == "encoder_pos_embedding" and lv[-1].shape == (1, 257, 64)
    assert lv[0].head_name == "output_head_action_head_continuous_head_bias"
    assert total_generated(SMALL_E) == 201_500 - 384 * 64
    n_enc = sum(int(np.prod(s)) for _, s in encoder_leaves(FULL))
    assert 85_000_000 < n_enc < 87_000_000       # 86.6 M with the un-baked 37x37 pos-emb


def test_config_roundtrip_and_rejections():
    for g in (FULL, MID, TINY, SMALL_E):
        cfg = json.loads(json.dumps(default_config(g)))
        assert geometry_from_config(cfg) == g
    cfg = default_config(FULL)
    cfg["base_net_kwargs"]["action_head_type"] = "diffusion"
    with pytest.raises(ValueError):
        geometry_from_config(cfg)
    cfg = default_config(FULL)
    cfg["hypernet_kwargs"]["share_layer_index"] = False
    with pytest.raises(ValueError):
        geometry_from_config(cfg)


def test_synthetic_params_cover_checkpoint_schema():
    p = syn.synthetic_params(TINY)
    shapes = hypernet_param_shapes(TINY)
    assert set(p) == set(shapes)
    assert all(p[k].dtype == np.float32 and p[k].shape == tuple(shapes[k]) for k in p)
    q = syn.synthetic_params(TINY)
    assert all(np.array_equal(p[k], q[k]) for k in p)                       # deterministic


def test_euler2axangle_against_rotation_matrices():
    rng = np.random.default_rng(3)
    for _ in range(20):
        r, pch, y = rng.uniform(-1.5, 1.5, 3)
        ax, ang = euler2axangle(r, pch, y)
        ax2, ang2 = onp.euler2axangle(r, pch, y)
        np.testing.assert_allclose(ax * ang, ax2 * ang2, atol=1e-12)
        cx, sx, cy, sy, cz, sz = np.cos(r), np.sin(r), np.cos(pch), np.sin(pch), np.cos(y), np.sin(y)
        Rx = np.array([[1, 0, 0], [0, cx, -sx], [0, sx, cx]])
        Ry = np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]])
        Rz = np.array([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1]])
        R = Rz @ Ry @ Rx                                                       # static xyz
        K = np.array([[0, -ax[2], ax[1]], [ax[2], 0, -ax[0]], [-ax[1], ax[0], 0]])
        Rod = np.eye(3) + np.sin(ang) * K + (1 - np.cos(ang)) * K @ K
        np.testing.assert_allclose(Rod, R, atol=1e-12)
    assert euler2axangle(0.0, 0.0, 0.0)[1] == 0.0


class _FakeModel:
    """create_tasks / sample_actions stand-in so the wrapper's host logic runs without a GPU."""

    def __init__(self, raw):
        self.raw, self.t = raw, 0
        self.config = default_config(FULL)
        self.dataset_statistics = syn.synthetic_dataset_statistics(FULL)

    def create_tasks(self, instruction_dict=None, initial_state=None):
        self.t = 0
        return object(), {"language_instruction": instruction_dict}, {}

    def sample_actions(self, images, instruction_dict, task, pad_mask, base_params, rng=None, image_embeddings=None):
        assert images.shape == (1, 1, 224, 224, 3) and images.dtype == np.uint8 and pad_mask.shape == (1, 1)
        a = self.raw[self.t][None]
        self.t += 1
        return a, {}


@pytest.mark.parametrize("setup", ["google_robot", "widowx_bridge", "libero"])
@pytest.mark.parametrize("ens", [True, False])
def test_wrapper_chain_matches_golden(golden_dir, setup, ens):
    z = np.load(golden_dir + "/wrapper.npz")
    raw = z["raw_actions"]
    w = InferenceWrapper(_FakeModel(raw), policy_setup=setup, horizon=1, pred_action_horizon=4, image_size=224,
                         action_ensemble=ens)
    w.reset("task", {"language_instruction": {}}, {"patch_embeddings": None})
    img = np.zeros((224, 224, 3), np.uint8)
    raws, acts = [], []
    for t in range(len(raw)):
        ra, act, im, (desc, task), dt = w.step(img)
        raws.append(ra), acts.append(act)
    np.testing.assert_allclose(np.stack(raws), z[f"{setup}_{int(ens)}_raw"], atol=1e-12)
    np.testing.assert_allclose(np.stack(acts), z[f"{setup}_{int(ens)}_act"], atol=1e-6)
    if setup == "google_robot" and not ens:
        assert (np.abs(np.stack(acts)[:, -1]) > 0.5).sum() >= 15             # sticky gripper exercised


def test_wrapper_rejects_bad_setup_and_sends_padded_resize_to_the_device():
    m = _FakeModel(np.zeros((1, 4, 7)))
    with pytest.raises(ValueError):
        InferenceWrapper(m, policy_setup="metaworld")
    w = InferenceWrapper(m, policy_setup="libero", pred_action_horizon=4, image_size=224, padded_resize=True)
    w.reset("t", {"language_instruction": {}}, {})
    seen = {}

    class _T:
        def __init__(self, a): self.a = a
        def __getitem__(self, i): return _T(self.a[i])
        def cpu(self): return self
        def numpy(self): return self.a

    def fake_preprocess(frames, crop=False, padded_resize=False):
        seen.update(crop=crop, padded_resize=padded_resize, shape=frames.shape)
        return _T(np.zeros((1, 224, 224, 3), np.uint8))
    m.preprocess_images = fake_preprocess
    m.geometry = type("G", (), {"image_size": 224})()
    w.step(np.zeros((480, 640, 3), np.uint8))
    assert seen == {"crop": False, "padded_resize": True, "shape": (480, 640, 3)}


def test_batched_ensembler_equals_unbatched():
    rng = np.random.default_rng(2)
    a, b = ActionEnsembler(4), [ActionEnsembler(4) for _ in range(3)]
    for t in range(6):
        x = rng.normal(size=(3, 4, 7))
        got = a.ensemble_action(x)
        for i in range(3):
            np.testing.assert_allclose(got[i], b[i].ensemble_action(x[i]), atol=1e-12)


def test_weight_decay_masks_follow_the_reference_strategies():
    """octo/utils/train_utils.py:325-382.  The reference's mask functions look at `jax.tree_util.keystr(path)` of every leaf
    of the hypernetwork's parameter tree and, for v3 / v5, at the top-level key `path[0].key`.  Here every leaf of the
    checkpoint schema gets its key path back ("output_head_<leaf>/bias" -> ["output_head_<leaf>", "bias"]), the four
    predicates are evaluated on it as the reference writes them, and the flat mask is compared slice by slice."""
    from hypervla.config import MID, generated_leaves
    from hypervla.train import train_param_layout, unpack_params, weight_decay_mask
    g = MID
    layout, total = train_param_layout(g, True)

    def keystr(keys):                                                   # jax.tree_util.keystr of DictKeys
        return "".join(f"['{k}']" for k in keys)

    def v1(keys):                                                       # :378-382
        return "kernel" in keystr(keys)

    def v2(keys):                                                       # :326-330
        ps = keystr(keys)
        return not ("norm" in ps.lower() and "output_head" not in ps)

    def v3(keys):                                                       # :335-350
        if "output_head" in keys[0]:
            return "kernel" in keys[0]
        return "image_encoder" in keystr(keys) or "kernel" in keystr(keys)

    def v5(keys):                                                       # :354-363
        if "output_head" in keys[0]:
            return "kernel" in keys[0]
        return "image_encoder" in keystr(keys)

    for strategy, pred in (("v1", v1), ("v2", v2), ("v3", v3), ("v5", v5)):
        mask = weight_decay_mask(g, strategy, True)
        assert mask.shape == (total,) and mask.dtype == np.uint8
        named = unpack_params(g, mask, True)                            # reference-named tensors of 0 / 1
        assert len(named) > 100
        for name, m in named.items():
            want = pred(name.split("/"))
            assert m.size and bool(m.all()) == want and (m.all() or not m.any()), (strategy, name, want)
    # the case the v1 mask is easy to get wrong on: the bias of a head that generates a base-net kernel is decayed
    lf = next(l for l in generated_leaves(g) if "kernel" in l.flat_name)
    assert unpack_params(g, weight_decay_mask(g, "v1", True), True)[lf.head_name + "/bias"].all()
    for bad in ("v4", "v6"):
        with pytest.raises(ValueError):
            weight_decay_mask(g, bad)


def test_gradient_buckets_tile_the_flat_vector():
    """The three all-reduce buckets (image encoder | output heads | context encoder, in the order the backward pass
    finishes them) are contiguous, disjoint and cover the flat gradient; the frozen-encoder layout has two."""
    from hypervla.config import FULL, MID
    from hypervla.train import gradient_buckets, train_param_layout
    for g in (MID, FULL):
        for enc in (False, True):
            layout, total = train_param_layout(g, enc)
            b = gradient_buckets(g, enc)
            assert [n for n, _, _ in b] == (["image_encoder"] if enc else []) + ["output_heads", "context_encoder"]
            spans = sorted((off, off + n) for _, off, n in b)
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(a[1] == c[0] for a, c in zip(spans, spans[1:]))
            at = {name: off for name, off, _ in layout}
            heads = dict((n, (o, l)) for n, o, l in b)["output_heads"]
            assert heads[0] == at["W_cat"] and heads[0] + heads[1] == (total if not enc else dict((n, (o, l)) for n, o, l in b)["image_encoder"][0])
    enc_bucket = gradient_buckets(FULL, True)[0]
    assert enc_bucket[2] > 85_000_000 and enc_bucket[2] * 4 > 340e6            # the 343 MB bucket that overlaps


def test_device_unnormalization_pairs():
    """BOUNDS as the affine pair hvla_ensemble takes: a * std + mean == (a + 1) (p99 - p01 + 1e-8) / 2 + p01."""
    from hypervla.interface import device_unnormalization
    rng = np.random.default_rng(1)
    p01 = rng.uniform(-2, 0, 7)
    stats = {"p01": p01, "p99": p01 + rng.uniform(0.1, 3, 7), "mean": rng.normal(size=7), "std": rng.uniform(0.1, 2, 7)}
    a = rng.uniform(-1, 1, (5, 7))
    mean, std, mask = device_unnormalization(stats, "bounds")
    np.testing.assert_allclose(a * std + mean, (a + 1) * (stats["p99"] - stats["p01"] + 1e-8) / 2 + stats["p01"], rtol=2e-6, atol=2e-6)
    assert mask.dtype == np.uint8 and mask.all()
    mean, std, _ = device_unnormalization(stats, "normal")
    np.testing.assert_allclose(mean, stats["mean"].astype(np.float32))
    np.testing.assert_allclose(std, stats["std"].astype(np.float32))


def test_gelu_polynomial_in_the_kernel_source_against_erf():
    """csrc/common.h gelu_erf2: gelu(x) = max(x, 0) - a 2^q(a), a = min(|x|, 8), q of degree 5 (round 5; degree 6 before).  The
    coefficients are read out of the kernel source and evaluated in f32 the way the kernel does (one rounding per fma): within 6e-7 of
    the erf form that HF's ACT2FN["gelu"] computes (transformers' Dinov2MLP), over the whole range fc1's pre-activations can reach --
    against an output that the epilogue rounds to 16 bits."""
    import os
    import re
    from scipy.special import erf
    src = open(os.path.join(os.path.dirname(__file__), "..", "hyper-vla_amd", "csrc", "common.h")).read()
    body = src[src.index("__device__ __forceinline__ f32x2 gelu_erf2"):]
    body = body[:body.index("__device__ __forceinline__ float gelu_erf(")]
    clamp = [float(c) for c in re.findall(r"elementwise_min\(ax, f32x2\{(-?[0-9.e+-]+)f,", body)][0]
    poly = body[body.index("f32x2 q = "):body.index("const f32x2 e = ")]
    coef = [float(c) for c in re.findall(r"f32x2\{(-?[0-9.e+-]+)f,", poly)]
    assert clamp == 8.0 and len(coef) == 6, (clamp, coef)                       # c5 .. c0
    assert "elementwise_fma(-a, e, m)" in body                                   # the result on the CLAMPED magnitude
    c = np.array(coef, np.float32)                                               # highest degree first
    x = np.concatenate([np.linspace(-70, 70, 400001), np.linspace(-1, 1, 100001)]).astype(np.float32)
    a = np.minimum(np.abs(x), np.float32(clamp))
    q = np.full_like(a, c[0])
    for ck in c[1:]:
        q = (q.astype(np.float64) * a + ck).astype(np.float32)
    e = np.exp2(q.astype(np.float64)).astype(np.float32)
    got = (np.maximum(x, 0).astype(np.float64) - a.astype(np.float64) * e).astype(np.float32)
    want = 0.5 * x.astype(np.float64) * (1.0 + erf(x.astype(np.float64) / np.sqrt(2.0)))
    err = np.abs(got - want)
    assert err.max() < 6e-7 + 1e-7 * 70, err.max()          # |error of the form| <= 4.7e-7; the rest is one f32 rounding of a value up to 70
    assert np.abs(got - want)[np.abs(x) <= 8].max() < 1e-6
    assert np.abs(got)[x < -8].max() < 1e-13                # beyond the clamp: zero in every 16-bit format


def test_bench_names_the_kernel_symbols_the_committed_profile_lists():
    """bench.py::big_gemm_symbols restates run_encoder's launch choices so that the bench line can NAME the instantiation it timed and look
    its counters up in the committed PMC passes.  The names it produces for the headline configuration must be symbols
    of the committed rocprofv3 statistics of that very command -- and the one with the largest share must be their top row."""
    import csv
    import importlib.util
    import os
    root = os.path.join(os.path.dirname(__file__), "..")
    spec = importlib.util.spec_from_file_location("bench_module", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    from hypervla.config import FULL, SMALL_E
    rows = list(csv.DictReader(open(os.path.join(root, "profiles", f"{bench.PMC_ROUND}_bench_b256_kernel_stats.csv"))))
    share = {r["Name"]: float(r["Percentage"]) for r in rows}
    sym = bench.big_gemm_symbols(256, FULL, "f16")
    assert set(sym) == {"qkv_gemm", "out_gemm", "fc1_gemm", "fc2_gemm"}
    for cat, (name, flops) in sym.items():
        assert name in share, (cat, name)
        assert flops > 0
    assert sym["out_gemm"][0] == sym["fc2_gemm"][0] == rows[0]["Name"]            # one instantiation, the profile's top row
    assert sym["fc1_gemm"][1] == 2.0 * 256 * 256 * 768 * 3072
    # the PMC lookup of that symbol finds all four counters' files
    pm = bench.pmc_rows(rows[0]["Name"], bench.PMC_ROUND)
    assert {"FETCH_SIZE", "WRITE_SIZE", "SQ_VALU_MFMA_BUSY_CYCLES", "GRBM_GUI_ACTIVE"} <= set(pm)
    # small batches have no dominant 256 x 256 symbol; DINOv2-small and two half-batches name other instantiations
    assert bench.big_gemm_symbols(4, FULL, "f16") is None
    assert "OpBF16" in bench.big_gemm_symbols(256, FULL, "bf16")["qkv_gemm"][0]
    assert bench.big_gemm_symbols(128, FULL, "f16")["out_gemm"][0].endswith("3, false, true, true>(hvla::GemmArgs)")
    assert bench.big_gemm_symbols(256, SMALL_E, "f16") is not None
