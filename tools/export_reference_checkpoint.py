"""Run this WHERE THE REFERENCE'S ENVIRONMENT EXISTS (jax + flax + orbax; not in the build container, not on the GPU
box): restores one Orbax step of a HyperVLA run and writes the flat ``params_<step>.npz`` that
``hypervla.model.HyperVLA.load_pretrained`` (this repo) reads.  The reference's own loader needs the model definition to
build a restore target (hypervla/model.py:196-210); restoring without a target returns the same nested dict of arrays.

    python tools/export_reference_checkpoint.py <run_dir> <step> <out_dir> [--ema 0.999]
"""
import argparse
import json
import os
import pickle
import shutil
import sys

import numpy as np


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("run_dir"), ap.add_argument("step", type=int), ap.add_argument("out_dir")
    ap.add_argument("--ema", type=float, default=None, help="export EMA_params.pkl[EMA_<x>] instead of the raw step")
    a = ap.parse_args()
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "hyper-vla_amd"))
    from hypervla.config import geometry_from_config
    from hypervla.convert import params_from_tree
    if a.ema is not None:
        with open(os.path.join(a.run_dir, str(a.step), "EMA_params.pkl"), "rb") as f:
            tree = pickle.load(f)[f"EMA_{a.ema}"]            # needs jax importable (or use hypervla.convert.load_ema_pickle)
    else:
        import orbax.checkpoint as ocp
        tree = ocp.CheckpointManager(a.run_dir, ocp.PyTreeCheckpointer()).restore(a.step)
    with open(os.path.join(a.run_dir, "config.json")) as f:
        config = json.load(f)
    config["base_net_kwargs"].setdefault("action_head_kwargs", dict(token_per_horizon=False, squash_continuous_action=True,
                                                                    clip_target=False, max_action=5.0))
    params = params_from_tree(tree, geometry_from_config(config))
    os.makedirs(a.out_dir, exist_ok=True)
    np.savez(os.path.join(a.out_dir, f"params_{a.step}.npz"), **params)
    for name in ("config.json", "dataset_statistics.json"):
        if os.path.exists(os.path.join(a.run_dir, name)):
            shutil.copyfile(os.path.join(a.run_dir, name), os.path.join(a.out_dir, name))
    print("wrote", os.path.join(a.out_dir, f"params_{a.step}.npz"), len(params), "tensors")


if __name__ == "__main__":
    main()
