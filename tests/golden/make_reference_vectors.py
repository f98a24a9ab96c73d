"""Golden vectors produced by EXECUTING THE REFERENCE'S OWN CODE (not the oracle).

    python tests/golden/make_reference_vectors.py        # needs /root/reference; run in the build container only

The action-prediction path of the reference is JAX / Flax and cannot run here, but one piece of the caller-side chain is
plain numpy and imports cleanly: `data/utils/action_ensemble.py::BatchActionEnsembler`, the temporal ensemble that
`InferenceWrapper.step` applies to every un-normalised action chunk (data/utils/hypervla_interface.py:250-253).  This
script loads that file from /root/reference, drives it with seeded action chunks and stores inputs and outputs in
`reference_action_ensemble.npz`; tests/test_reference_vectors.py checks the host `ActionEnsembler`, the oracle's
restatement and the device-side `hvla_ensemble` ring against them.  Only data is committed, never the reference's source.
"""
import importlib.util
import os

import numpy as np

REF = "/root/reference/data/utils/action_ensemble.py"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "reference_action_ensemble.npz")


def main():
    spec = importlib.util.spec_from_file_location("ref_action_ensemble", REF)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    rng = np.random.default_rng(20240917)
    out = {}
    for name, (B, horizon, temp, steps) in {"b1_h4_t0": (1, 4, 0.0, 7), "b5_h4_t0": (5, 4, 0.0, 9),
                                            "b3_h4_t05": (3, 4, 0.5, 6), "b2_h1_t0": (2, 1, 0.0, 3)}.items():
        ens = mod.BatchActionEnsembler(horizon, temp)
        ens.reset()
        x = rng.uniform(-2.0, 2.0, size=(steps, B, horizon, 7))
        y = np.stack([ens.ensemble_action(x[t]) for t in range(steps)])
        out[name + "_in"], out[name + "_out"] = x, y
        out[name + "_cfg"] = np.array([B, horizon, temp, steps], np.float64)
    np.savez_compressed(OUT, **out)
    print("wrote", OUT, {k: v.shape for k, v in out.items() if k.endswith("_out")})


if __name__ == "__main__":
    main()
