"""One-command pin of this build against the REFERENCE ITSELF.  Run it WHERE THE REFERENCE'S ENVIRONMENT EXISTS
(jax 0.4.20 + flax 0.8.1 + transformers 4.50.0 + the Hyper-VLA checkout; not in the build container and never on the GPU
box -- nothing here travels there, only the .npz it writes does):

    python tools/make_reference_fixtures.py --reference /path/to/Hyper-VLA [--episodes 4] [--out tests/golden/reference_full_b4.npz]

What it does, at the README geometry (DINOv2-base -> generated vit_t 4L/64d -> mix head):
  1. builds the reference's `hypervla.model.HyperVLA` with `HyperVLA.from_config` (hypervla/model.py:286-365) on an
     OXE-shaped example batch, then swaps in THIS repo's seeded `synthetic_params` (inverse of
     `hypervla.convert.params_from_tree`; every leaf is reshaped to the reference leaf it replaces, and a leaf without a
     counterpart on either side is an error).  The DINOv2 position table is the hub-shaped [1, 1370, 768] one
     (`synthetic_position_table_hub`), which the reference resizes to 16 x 16 inside its forward pass; this build bakes the
     same table once (`convert.bake_position_embeddings`), so the file also pins that restatement of JAX's bicubic kernel.
  2. for each of the seeded episodes (the reference is batch 1 only, hypervla/model.py:81) runs `create_tasks` and
     `sample_actions` (hypervla/model.py:35-137) on the seeded `synthetic_instructions` / `synthetic_initial_state` /
     `synthetic_images`, on the CPU backend in float32 (the reference's default precision),
  3. writes actions, the context embedding, samples + sums of the generated parameter vector (the index set of
     tests/golden/full_b4.npz), and the two attention maps the evaluators pickle
     (data/utils/hypervla_interface.py:208-217): DINOv2 CLS-row attention [12, 12, 256] and the generated policy's
     action-row attention [4, 4, 256].
`tests/test_reference_vectors.py` consumes the file when it exists (CPU: the float64 oracle against it; `-m gpu`: the HIP
path against it) and reports "parity unpinned" while it does not.

This repo's host package is also called `hypervla` (it shadows the reference's: INTEGRATION.md), so it is imported here
under the alias `hvla_amd` and the reference's own `hypervla` under its real name.
"""
import argparse
import importlib
import importlib.util
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def import_this_repo():
    """hyper-vla_amd/hypervla as the package `hvla_amd` (its modules only use relative imports)."""
    pkg = os.path.join(ROOT, "hyper-vla_amd", "hypervla")
    spec = importlib.util.spec_from_file_location("hvla_amd", os.path.join(pkg, "__init__.py"), submodule_search_locations=[pkg])
    mod = importlib.util.module_from_spec(spec)
    sys.modules["hvla_amd"] = mod
    spec.loader.exec_module(mod)
    return (importlib.import_module("hvla_amd.synthetic"), importlib.import_module("hvla_amd.config"),
            importlib.import_module("hvla_amd.convert"))


def reference_config(ref_root, ours):
    """The README run's config (README.md:14-62).  The reference's own config file supplies every key (`model`, the octo
    kwargs `BaseNetwork` stores, ...); the README's flag overrides are this repo's `default_config(FULL)` sections."""
    cfg = None
    try:
        sys.path.insert(0, os.path.join(ref_root, "scripts", "configs"))
        cfg_mod = importlib.import_module("hypervla_pretrain_config")
        cfg = cfg_mod.get_config("vit_t,oxe").to_dict()            # the README run: hypervla_pretrain_config.py:vit_t,oxe
    except Exception as e:                                  # tf / dlimp imports of the data config may be absent
        print(f"[make_reference_fixtures] reference config file not importable ({type(e).__name__}: {e}); "
              "using this repo's default_config + an empty `model` section", file=sys.stderr)
        cfg = {"model": {}}

    def update(dst, src):
        for k, v in src.items():
            if isinstance(v, dict) and isinstance(dst.get(k), dict):
                update(dst[k], v)
            else:
                dst[k] = v

    update(cfg, {k: ours[k] for k in ("hypernet_kwargs", "base_net_kwargs", "window_size")})
    cfg.setdefault("model", {})
    return cfg


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reference", required=True, help="root of the MasterXiong/Hyper-VLA checkout")
    ap.add_argument("--episodes", type=int, default=4)
    ap.add_argument("--out", default=os.path.join(ROOT, "tests", "golden", "reference_full_b4.npz"))
    a = ap.parse_args()
    os.environ.setdefault("JAX_PLATFORMS", "cpu")           # the north star's "JAX-CPU reference"
    syn, cfgm, conv = import_this_repo()
    g, B = cfgm.FULL, a.episodes

    sys.path.insert(0, a.reference)
    import flax
    import jax
    import jax.numpy as jnp
    import transformers
    from hypervla.components.hypernetwork import HyperNetwork          # the REFERENCE's package
    from hypervla.model import HyperVLA

    # ---- seeded inputs and weights of this repo
    ins, st, im = syn.synthetic_instructions(B, g), syn.synthetic_initial_state(B, g), syn.synthetic_images(B, g)
    P = dict(syn.synthetic_params(g))
    pos_key = next(k for k in P if k.endswith("embeddings_position_embeddings"))
    P[pos_key] = syn.synthetic_position_table_hub(g).reshape(-1)
    ours = conv.tree_from_params(P)

    # ---- the reference model, freshly initialised, then our parameters in its tree
    li = ins["language_instruction"]
    example_batch = {
        "observation": {"image_primary": im[:1], "timestep_pad_mask": np.ones((1, 1), bool)},
        "task": {"language_instruction": {k: v[:1] for k, v in li.items()},
                 "pad_mask_dict": {"language_instruction": np.ones(1, bool)}},
        "initial_state": {"image_primary": im[:1], "patch_embeddings": st["patch_embeddings"][:1],
                          "pad_mask_dict": {"image_primary": np.ones((1, 1))}},
        "action": np.zeros((1, 1, g.horizon, g.action_dim), np.float32),
        "action_pad_mask": np.ones((1, 1, g.horizon, g.action_dim), bool),
    }
    config = reference_config(a.reference, cfgm.default_config(g))
    model = HyperVLA.from_config(config, example_batch, rng=jax.random.PRNGKey(0))
    ref_flat = flax.traverse_util.flatten_dict(flax.core.unfreeze(model.params), sep="/")
    our_flat = flax.traverse_util.flatten_dict(ours, sep="/")
    missing, extra = sorted(set(ref_flat) - set(our_flat)), sorted(set(our_flat) - set(ref_flat))
    if missing or extra:
        raise SystemExit(f"parameter trees differ: {len(missing)} reference leaves without a counterpart (e.g. {missing[:3]}), "
                         f"{len(extra)} of this repo's leaves the reference does not have (e.g. {extra[:3]})")
    new = {}
    for k, v in ref_flat.items():
        w = np.asarray(our_flat[k], np.float32)
        if w.size != np.asarray(v).size:
            raise SystemExit(f"{k}: reference leaf has {np.asarray(v).shape}, this repo's {w.shape}")
        new[k] = jnp.asarray(w.reshape(np.asarray(v).shape))
    model = model.replace(params=flax.traverse_util.unflatten_dict(new, sep="/"))

    # ---- per episode: create_tasks + sample_actions, exactly as InferenceWrapper.reset / .step call them
    leaves = cfgm.generated_leaves(g)
    rng77 = np.random.Generator(np.random.PCG64(77))        # the index set of tests/golden/make_golden.py (full_b4.npz)
    idx = np.concatenate([l.offset + np.sort(rng77.choice(l.size, size=min(64, l.size), replace=False)) for l in leaves])
    out = {k: [] for k in ("actions", "ctx", "theta_samples", "theta_sum", "theta_abs_sum", "dino_cls_attention", "head_attention")}
    key = jax.random.PRNGKey(0)
    for b in range(B):
        idict = {"language_instruction": {k: v[b:b + 1] for k, v in li.items()}}
        init = {"patch_embeddings": st["patch_embeddings"][b:b + 1], "pad_mask_dict": {"image_primary": np.ones((1, 1))}}
        base_params, tasks, _ = model.create_tasks(instruction_dict=idict, initial_state=init)
        bp = flax.traverse_util.flatten_dict(flax.core.unfreeze(base_params), sep="/")
        theta = np.concatenate([np.asarray(bp["/".join(l.path)], np.float64).reshape(-1) for l in leaves])
        assert theta.size == leaves[-1].offset + leaves[-1].size, theta.size
        ctx = model.hypernet.apply({"params": model.params}, tasks, False, init, method=HyperNetwork.generate_context_embedding)
        key, sub = jax.random.split(key)
        actions, inter = model.sample_actions(im[b:b + 1], idict, tasks, np.ones((1, 1), bool), base_params, rng=sub)
        enc = inter["intermediates"]["encoder"]
        dino = np.stack([np.asarray(x)[0, :, 0, 1:] for x in enc["DINO_attention_map"][0]])          # hypervla_interface.py:210-211
        tf0 = enc["Transformer_0"]
        try:
            head = np.stack([np.asarray(tf0[f"encoderblock_{i}"]["MultiHeadDotProductAttention_0"]["attention_weights"][0])[0, :, -1, :-1]
                             for i in range(g.layers)])                                               # :213-215
        except (KeyError, TypeError):
            head = np.stack([np.asarray(tf0[f"encoderblock_{i}"]["attention_map"][0])[0, :, -1, :-1] for i in range(g.layers)])
        out["actions"].append(np.asarray(actions)[0])
        out["ctx"].append(np.asarray(ctx).reshape(-1)[-g.ctx_dim:])
        out["theta_samples"].append(theta[idx]), out["theta_sum"].append(theta.sum()), out["theta_abs_sum"].append(np.abs(theta).sum())
        out["dino_cls_attention"].append(dino), out["head_attention"].append(head)
        print(f"episode {b}: actions[0] = {np.asarray(actions)[0, 0]}", flush=True)
    np.savez_compressed(
        a.out, theta_idx=idx, **{k: np.stack(v) for k, v in out.items()},
        versions=np.array([f"jax {jax.__version__}", f"flax {flax.__version__}", f"transformers {transformers.__version__}",
                           f"backend {jax.default_backend()}"]))
    print("wrote", a.out, os.path.getsize(a.out), "bytes")


if __name__ == "__main__":
    main()
