"""Diagnostic (GPU box): race screen for the production encoder GEMM schedule -- many repetitions of the encoder at several
batch sizes (persistent whole rounds, one workgroup per tile, multi-round grids, the 64x64 kernel), every run compared bit
for bit with the first; the sha1 of image 0's tokens must also be the same at every batch size (a row gets the same bits
whichever kernel and schedule computes it)."""
import os, sys, hashlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "hyper-vla_amd"))
import numpy as np, torch
from hypervla import synthetic as syn
from hypervla.config import FULL
from hypervla.model import HyperVLA
g = FULL
m = HyperVLA.from_synthetic(g, max_batch=1024)
for B, runs in ((256, 300), (37, 300), (96, 200), (1024, 60), (4, 500)):
    im = torch.as_tensor(syn.synthetic_images(B, g)[:, 0]).to(m.device).contiguous()
    ref = m.encode_images(im).clone()
    bad = 0
    for _ in range(runs):
        bad += int(not torch.equal(m.encode_images(im), ref))
    h = hashlib.sha1(ref.cpu().numpy().tobytes()).hexdigest()[:16]
    h0 = hashlib.sha1(ref[0].cpu().numpy().tobytes()).hexdigest()[:16]
    print(f"B={B}: {runs} runs, differing from the first: {bad}; tokens sha1 {h}; image 0 sha1 {h0}")
