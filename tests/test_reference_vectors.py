"""Vectors produced by running the reference's own `BatchActionEnsembler` (tests/golden/make_reference_vectors.py):
the one piece of the path that is plain numpy in the reference and therefore pins against the reference itself."""
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


def _cases():
    z = np.load(os.path.join(HERE, "golden", "reference_action_ensemble.npz"))
    for k in sorted(z.files):
        if k.endswith("_cfg"):
            n = k[:-4]
            B, horizon, temp, steps = z[k]
            yield n, int(B), int(horizon), float(temp), int(steps), z[n + "_in"], z[n + "_out"]


def test_host_ensembler_reproduces_the_reference():
    from hypervla.interface import ActionEnsembler
    seen = 0
    for n, B, horizon, temp, steps, x, y in _cases():
        ens = ActionEnsembler(horizon, temp)
        ens.reset()
        for t in range(steps):
            np.testing.assert_allclose(ens.ensemble_action(x[t]), y[t], rtol=0, atol=1e-14, err_msg=n)
        one = ActionEnsembler(horizon, temp)                  # the unbatched form InferenceWrapper uses per episode
        for t in range(steps):
            np.testing.assert_allclose(one.ensemble_action(x[t, 0]), y[t, 0], rtol=0, atol=1e-14, err_msg=n)
        seen += 1
    assert seen == 4


def test_oracle_ensembler_reproduces_the_reference():
    from oracle import hvla_ref_np as onp
    for n, B, horizon, temp, steps, x, y in _cases():
        ens = onp.Ensembler(horizon, temp)
        for t in range(steps):
            np.testing.assert_allclose(ens(x[t]), y[t], rtol=0, atol=1e-14, err_msg=n)


@pytest.mark.gpu
def test_device_ensemble_ring_reproduces_the_reference():
    """hvla_ensemble (temperature 0, un-normalisation with mean 0 / std 1 = identity) against the reference's outputs."""
    import torch
    from hypervla.config import FULL
    from hypervla.model import HyperVLA
    from hypervla.synthetic import synthetic_initial_state, synthetic_instructions
    g = FULL
    for n, B, horizon, temp, steps, x, y in _cases():
        if temp != 0.0 or horizon != g.horizon:
            continue
        m = HyperVLA.from_synthetic(g, max_batch=8)
        w, _, _ = m.create_tasks(instruction_dict=synthetic_instructions(B, g), initial_state=synthetic_initial_state(B, g))
        dev = m.device
        mean, std = torch.zeros(7, device=dev), torch.ones(7, device=dev)
        mask = torch.ones(7, dtype=torch.uint8, device=dev)
        out = torch.empty(B, 7, device=dev)
        m._ctx.ensemble_reset(w._h, m._stream())
        for t in range(steps):
            a = torch.tensor(x[t].astype(np.float32), device=dev)
            m._ctx.ensemble(w._h, a.data_ptr(), mean.data_ptr(), std.data_ptr(), mask.data_ptr(), out.data_ptr(), m._stream())
            np.testing.assert_allclose(out.cpu().numpy(), y[t], rtol=0, atol=2e-6, err_msg=f"{n} step {t}")
