"""CPU: the numpy oracle against the committed golden fixtures and against the independent torch
restatement (SURVEY.md §8c substitute pins 1 and 2)."""
import numpy as np
import pytest

from hypervla import synthetic as syn
from hypervla.config import FULL, TINY, encoder_leaves, generated_leaves
from oracle import hvla_ref_np as onp


def _run(g, B, sink=None):
    P = syn.synthetic_params(g)
    leaves, enc_shapes = generated_leaves(g), dict(encoder_leaves(g))
    ins, st, im = syn.synthetic_instructions(B, g), syn.synthetic_initial_state(B, g), syn.synthetic_images(B, g)
    bp, ctx = onp.create_tasks(P, g, leaves, ins, st, sink)
    act, logit, emb, tok = onp.sample_actions(P, g, enc_shapes, bp, im, sink)
    theta = np.concatenate([bp[l.flat_name].reshape(B, -1) for l in leaves], 1)
    return P, leaves, enc_shapes, ins, st, im, dict(ctx=ctx[:, 0], theta=theta, actions=act, logits=logit, emb=emb, tokens=tok)


def test_tiny_every_intermediate_matches_golden(golden_dir):
    z = np.load(golden_dir + "/tiny_all.npz")
    sink = {}
    *_, r = _run(TINY, 3, sink)
    assert len(z.files) > 20
    for k in z.files:
        if k.startswith("out__"):
            got = r[k[5:]]
        else:
            got = np.asarray(sink[k.replace("__", "/")])
        np.testing.assert_allclose(got, z[k], rtol=0, atol=1e-12, err_msg=k)


def test_tiny_numpy_vs_independent_torch_restatement():
    torch = pytest.importorskip("torch")
    from oracle import hvla_ref_torch as ot
    P, leaves, enc_shapes, ins, st, im, r = _run(TINY, 3)
    ref = ot.FullRef(P, TINY, leaves, enc_shapes, torch.float64)
    theta, ctx = ref.create_tasks(ins, st)
    act, logit, emb, tok = ref.sample_actions(theta, im)
    assert np.abs(ctx.numpy() - r["ctx"]).max() <= 1e-12
    assert np.abs(theta.numpy() - r["theta"]).max() <= 1e-12
    assert np.abs(tok.numpy() - r["tokens"]).max() <= 1e-5      # f32 mean/std constants in the torch graph
    assert np.abs(act.numpy() - r["actions"]).max() <= 1e-5
    assert np.abs(logit.numpy() - r["logits"]).max() <= 1e-5


@pytest.mark.timeout(600)
def test_full_geometry_matches_golden_and_torch(golden_dir):
    """README geometry (DINOv2-base, vit_t) B=2 prefix of the B=4 fixture; HF's own torch Dinov2Model is
    the encoder of the second restatement."""
    torch = pytest.importorskip("torch")
    from oracle import hvla_ref_torch as ot
    z = np.load(golden_dir + "/full_b4.npz")
    B = 2
    P, leaves, enc_shapes, ins, st, im, r = _run(FULL, B)
    # synthetic inputs for B=2 are not a prefix of B=4 (one RNG stream per call) -> regenerate at B=4 cheaply
    ins4, st4 = syn.synthetic_instructions(4, FULL), syn.synthetic_initial_state(4, FULL)
    bp4, ctx4 = onp.create_tasks(P, FULL, leaves, ins4, st4)
    theta4 = np.concatenate([bp4[l.flat_name].reshape(4, -1) for l in leaves], 1)
    np.testing.assert_allclose(ctx4[:, 0], z["ctx"], atol=1e-12)
    np.testing.assert_allclose(theta4[:, z["theta_idx"]], z["theta_samples"], atol=1e-12)
    np.testing.assert_allclose(theta4.sum(1), z["theta_sum"], atol=1e-9)
    ref = ot.FullRef(P, FULL, leaves, enc_shapes, torch.float32)
    theta, ctx = ref.create_tasks(ins, st)
    act, logit, emb, tok = ref.sample_actions(theta, im)
    assert np.abs(ctx.numpy() - r["ctx"]).max() <= 1e-5
    assert np.abs(theta.numpy() - r["theta"]).max() <= 1e-5
    assert np.abs(tok.numpy() - r["tokens"]).max() <= 1e-4
    assert np.abs(act.numpy()[..., :6] - r["actions"][..., :6]).max() <= 1e-4
