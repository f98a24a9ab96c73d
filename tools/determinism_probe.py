"""Diagnostic (GPU box): run-to-run bitwise stability of every stage of the hot path on identical inputs."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "hyper-vla_amd"))
import numpy as np, torch
from hypervla import synthetic as syn
from hypervla.config import FULL
from hypervla.model import HyperVLA
g = FULL
m = HyperVLA.from_synthetic(g, max_batch=256)
for B, runs in ((1, 300), (8, 100), (256, 12)):
    ins, st, im = syn.synthetic_instructions(B, g), syn.synthetic_initial_state(B, g), syn.synthetic_images(B, g)
    ref = None
    bad = {"ctx": 0, "theta": 0, "tokens": 0, "actions": 0}
    worst = dict(bad)
    for it in range(runs):
        w, _, _ = m.create_tasks(instruction_dict=ins, initial_state=st)
        theta, ctx = w.export()
        tok = m.encode_images(im)
        a, l = m.policy_from_tokens(tok, w)
        cur = {"ctx": ctx.clone(), "theta": theta.clone(), "tokens": tok.clone(), "actions": a.clone()}
        if ref is None:
            ref = cur
            continue
        for k in cur:
            d = float((cur[k] - ref[k]).abs().max())
            if d > 0:
                bad[k] += 1
                worst[k] = max(worst[k], d)
    print(f"B={B} runs={runs}: runs differing from the first ->", {k: (bad[k], f"{worst[k]:.1e}") for k in bad})
