"""Diagnostic (GPU box): is the policy megakernel run-to-run deterministic on identical inputs?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "hyper-vla_amd"))
import numpy as np, torch
from hypervla import synthetic as syn
from hypervla.config import FULL
from hypervla.model import HyperVLA
g = FULL
Bs = [int(v) for v in sys.argv[1:]] or [256]
m = HyperVLA.from_synthetic(g, max_batch=max(Bs))
for B in Bs:
    ins, st, im = syn.synthetic_instructions(B, g), syn.synthetic_initial_state(B, g), syn.synthetic_images(B, g)
    w, _, _ = m.create_tasks(instruction_dict=ins, initial_state=st)
    tok = m.encode_images(im)
    ref_a, ref_l = [t.clone() for t in m.policy_from_tokens(tok, w)]
    runs = max(30, int(os.environ.get('HVLA_PROBE_EPISODE_RUNS', '7680')) // B)
    bad, worst = 0, 0.0
    for it in range(runs):
        a, l = m.policy_from_tokens(tok, w)
        d = (a - ref_a).abs().reshape(B, -1).max(1).values
        bad += int((d > 0).sum())
        worst = max(worst, float(d.max()))
    torch.cuda.synchronize()
    import time
    t0 = time.perf_counter()
    for it in range(50):
        m.policy_from_tokens(tok, w)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 50 * 1e3
    print(f"B={B}: {runs} runs, episode-runs that differ from the first run: {bad} of {runs * B}, worst |d action| {worst:.2e}; {ms:.4f} ms per launch")
