"""Diagnostic (GPU box): encoder token error vs the float64 oracle, by depth and operand type."""
import dataclasses, sys, os
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "hyper-vla_amd")); sys.path.insert(0, ROOT)
from hypervla import synthetic as syn
from hypervla.config import MID, encoder_leaves, generated_leaves
from hypervla.model import HyperVLA
from oracle import hvla_ref_np as onp

B = 4
for layers in (0, 1, 2):
    g = dataclasses.replace(MID, enc_layers=layers)
    P = syn.synthetic_params(g)
    im = syn.synthetic_images(B, g)
    sink = {}
    hs = onp.dinov2(P, g, dict(encoder_leaves(g)), onp.normalize_images(im[:, 0]), sink)
    tok = hs[:, 1:]
    for dt in ("f16", "bf16"):
        m = HyperVLA.from_synthetic(g, max_batch=8, enc_dtype=dt)
        t = m.encode_images(im).cpu().numpy().astype(np.float64)
        d = t - tok
        per_tok = np.sqrt((d * d).mean(-1))
        print(f"layers={layers} {dt}: rms {np.sqrt((d*d).mean()):.3e} max {np.abs(d).max():.3e} "
              f"worst-token rms {per_tok.max():.3e} at {np.unravel_index(per_tok.argmax(), per_tok.shape)} "
              f"median-token rms {np.median(per_tok):.3e}; per-feature rms max {np.sqrt((d*d).mean((0,1))).max():.3e}")
        del m
