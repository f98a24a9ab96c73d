"""Diagnostic (GPU box): do the LayerNorms fused into the residual GEMMs (gemm256p_kernel<..., LNX>, B >= 8) give every image the
bytes of the stand-alone kernels (B = 1)?  With `spin` = 0 (libhvla_bench.so) nobody waits: every tile but an image's last arriver
is normalised from memory.

    python tools/lnx_check.py [spin ticks of 10 ns] [B ...]
"""
import ctypes as C, os, sys
PRODUCT = os.environ.get("HVLA_LNX_PRODUCT") == "1"        # the product library (no spin override): for A/B variants copied over libhvla.so
if not PRODUCT:
    os.environ["HVLA_LIBRARY_FLAVOUR"] = "bench"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "hyper-vla_amd")); sys.path.insert(0, ROOT)
import numpy as np, torch
from hypervla import synthetic as syn
from hypervla.config import FULL
from hypervla.model import HyperVLA

spin = int(sys.argv[1]) if len(sys.argv) > 1 else 800
Bs = [int(x) for x in sys.argv[2:]] or [8, 9, 40, 64, 85, 86, 255, 256, 512]
BM = max(Bs)
m = HyperVLA.from_synthetic(FULL, max_batch=BM)
lib = m._ctx.lib
if not PRODUCT:
    lib.hvla_debug_lnx_spin.argtypes = [C.c_void_p, C.c_uint32]
    assert lib.hvla_debug_lnx_spin(m._ctx.h, spin) == 0
im = syn.synthetic_images(BM, FULL)[:, 0]
ref = torch.stack([m.encode_images(im[i:i + 1]).cpu()[0] for i in (0, 1, 7, BM - 1)])
for B in Bs:
    bad = []
    for rep in range(3):
        tok = m.encode_images(im[:B]).cpu()
        for k, i in enumerate((0, 1, 7, BM - 1)):
            if i < B and not torch.equal(tok[i], ref[k]):
                bad.append((rep, i, float((tok[i] - ref[k]).abs().max())))
    print(f"spin {spin} B = {B}: {'same bytes' if not bad else bad[:6]}", flush=True)
