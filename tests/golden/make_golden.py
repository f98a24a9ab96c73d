"""Generate the committed golden fixtures with the numpy float64 oracle.

    python tests/golden/make_golden.py

The reference holds no golden vectors for this path and cannot be imported here
(no jax/flax), so these fixtures pin the *oracle's* outputs on the seeded synthetic weights/inputs of
``hypervla.synthetic`` (SURVEY.md §8c "substitute pins").  Inputs are never stored: they are
regenerated from the fixed seeds.  Files:

  tiny_all.npz    tiny geometry, B=3: every intermediate of create_tasks + sample_actions
  full_b4.npz     README geometry (DINOv2-base, vit_t), B=4: ctx, generated-parameter checksum and
                  64 sampled entries per leaf, sampled encoder tokens, action-token embedding,
                  gripper logits, actions
  full_b64*.npz   README geometry, 64 episodes (the sample the end-to-end tolerance is asserted on): actions, gripper
                  logits, action-token embedding, 256 sampled tokens per episode, for
                    full_b64.npz             synthetic weights, iid-noise images
                    full_b64_trained.npz     trained-DINOv2-like encoder statistics (LayerScale 0.05..1, outlier channels
                                             >= 100 in the residual stream, heavy-tailed fc1), iid-noise images
                    full_b64_structured.npz  synthetic weights, camera-like images (smooth scenes, per-image brightness /
                                             contrast): neighbouring tokens are similar, so rounding errors are correlated
  wrapper.npz     caller-side chain (un-normalise, ensemble, axis-angle, gripper rules) for the three
                  policy setups on a seeded raw-action sequence

    python tests/golden/make_golden.py            # everything (about ten minutes on 8 cores)
    python tests/golden/make_golden.py b64        # only the three 64-episode fixtures
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "hyper-vla_amd"))
sys.path.insert(0, ROOT)

from hypervla import synthetic as syn                               # noqa: E402
from hypervla.config import FULL, TINY, encoder_leaves, generated_leaves   # noqa: E402
from oracle import hvla_ref_np as onp                               # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


def run(g, B, sink=None, params=None, images=None):
    P = syn.synthetic_params(g) if params is None else params
    leaves, enc_shapes = generated_leaves(g), dict(encoder_leaves(g))
    ins, st = syn.synthetic_instructions(B, g), syn.synthetic_initial_state(B, g)
    im = syn.synthetic_images(B, g) if images is None else images
    bp, ctx = onp.create_tasks(P, g, leaves, ins, st, sink)
    act, logit, emb, tok = onp.sample_actions(P, g, enc_shapes, bp, im, sink)
    theta = np.concatenate([bp[l.flat_name].reshape(B, -1) for l in leaves], 1)
    return dict(ctx=ctx[:, 0], theta=theta, actions=act, logits=logit, emb=emb, tokens=tok)


def b64():
    """The three 64-episode fixtures at the README geometry (in chunks of 16 episodes: the float64 encoder needs about
    1 GB per 16 images)."""
    B, CH = 64, 16
    cases = {"full_b64.npz": (None, syn.synthetic_images(B, FULL)),
             "full_b64_trained.npz": (syn.synthetic_params_trained_like(FULL), syn.synthetic_images(B, FULL)),
             "full_b64_structured.npz": (None, syn.synthetic_images_structured(B, FULL))}
    rng = np.random.Generator(np.random.PCG64(79))
    tok_idx = np.sort(rng.choice(FULL.patches * FULL.enc_dim, size=256, replace=False))
    leaves, enc_shapes = generated_leaves(FULL), dict(encoder_leaves(FULL))
    for name, (params, images) in cases.items():
        P = syn.synthetic_params(FULL) if params is None else params
        ins, st = syn.synthetic_instructions(B, FULL), syn.synthetic_initial_state(B, FULL)
        bp, ctx = onp.create_tasks(P, FULL, leaves, ins, st)
        acts, logits, embs, toks = [], [], [], []
        for c in range(0, B, CH):
            sub = {k: v[c:c + CH] for k, v in bp.items()}
            a, l, e, t = onp.sample_actions(P, FULL, enc_shapes, sub, images[c:c + CH])
            acts.append(a), logits.append(l), embs.append(e), toks.append(t.reshape(CH, -1)[:, tok_idx])
        np.savez_compressed(os.path.join(OUT, name), actions=np.concatenate(acts), logits=np.concatenate(logits),
                            emb=np.concatenate(embs), tok_idx=tok_idx, tok_samples=np.concatenate(toks))
        print(name, os.path.getsize(os.path.join(OUT, name)), flush=True)


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "b64":
        return b64()
    b64()
    # ---- tiny: everything
    sink = {}
    r = run(TINY, 3, sink)
    keep = {k.replace("/", "__"): np.asarray(v) for k, v in sink.items()}
    keep.update({"out__" + k: v for k, v in r.items()})
    np.savez_compressed(os.path.join(OUT, "tiny_all.npz"), **keep)

    # ---- full: sampled
    B = 4
    r = run(FULL, B)
    rng = np.random.Generator(np.random.PCG64(77))
    leaves = generated_leaves(FULL)
    idx = np.concatenate([l.offset + np.sort(rng.choice(l.size, size=min(64, l.size), replace=False))
                          for l in leaves])
    tok_idx = np.sort(rng.choice(r["tokens"][0].size, size=4096, replace=False))
    np.savez_compressed(
        os.path.join(OUT, "full_b4.npz"),
        ctx=r["ctx"], theta_idx=idx, theta_samples=r["theta"][:, idx],
        theta_sum=r["theta"].sum(1), theta_abs_sum=np.abs(r["theta"]).sum(1),
        tok_idx=tok_idx, tok_samples=r["tokens"].reshape(B, -1)[:, tok_idx],
        tok_mean=r["tokens"].mean((1, 2)), tok_sq=np.square(r["tokens"]).mean((1, 2)),
        emb=r["emb"], logits=r["logits"], actions=r["actions"])

    # ---- wrapper chain
    rng = np.random.Generator(np.random.PCG64(88))
    T = 40
    raw = rng.uniform(-2.0, 2.0, size=(T, 4, 7))
    raw[..., 6] = (rng.uniform(size=(T, 4)) > 0.5) if True else 0
    # make the gripper switch rarely so the sticky logic (15 repeats) is exercised
    raw[..., 6] = (np.arange(T)[:, None] // 11 % 2)
    stats = syn.synthetic_dataset_statistics(FULL)["bridge_dataset"]["action"]
    out = {"raw_actions": raw}
    for setup in ("google_robot", "widowx_bridge", "libero"):
        for ens in (True, False):
            a, b = onp.postprocess_episode(raw, stats, setup, ens)
            out[f"{setup}_{int(ens)}_raw"] = a
            out[f"{setup}_{int(ens)}_act"] = b
    np.savez_compressed(os.path.join(OUT, "wrapper.npz"), **out)
    for f in ("tiny_all.npz", "full_b4.npz", "wrapper.npz"):
        print(f, os.path.getsize(os.path.join(OUT, f)))


if __name__ == "__main__":
    main()
