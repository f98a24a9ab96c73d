"""Diagnostic (GPU box): tokens of the last episode of a B=256 batch (split-K tail tiles) against the same episode alone."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "hyper-vla_amd"))
import numpy as np, torch
from hypervla import synthetic as syn
from hypervla.config import FULL
from hypervla.model import HyperVLA
g, B = FULL, 256
m = HyperVLA.from_synthetic(g, max_batch=B)
im = syn.synthetic_images(B, g)
big = m.encode_images(im).cpu().numpy()
for e in (0, 100, 254, 255):
    one = m.encode_images(im[e:e + 1]).cpu().numpy()[0]
    d = np.abs(big[e] - one)
    print("episode", e, "max |tokens(B=256) - tokens(B=1)| =", d.max(), "rms", np.sqrt((d * d).mean()), "rows>1e-3:", (d.max(1) > 1e-3).sum())
big2 = m.encode_images(im).cpu().numpy()
print("run-to-run max diff", np.abs(big2 - big).max(), "episodes differing", (np.abs(big2 - big).reshape(B, -1).max(1) > 0).sum())
