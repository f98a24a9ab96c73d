"""Diagnostic (GPU box): how the LayerNorm tail jobs of gemm256p_kernel<..., LNT> spread over the workgroups of one step
(libhvla_bench.so, hvla_debug_lnt_stats): jobs per workgroup and launch, shader-clock ticks inside jobs and inside the
ticket / claim protocol.

    python tools/lnt_stats.py [B]
"""
import ctypes as C, os, sys
os.environ["HVLA_LIBRARY_FLAVOUR"] = "bench"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "hyper-vla_amd")); sys.path.insert(0, ROOT)
import numpy as np
from hypervla import synthetic as syn
from hypervla.config import FULL
from hypervla.model import HyperVLA

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
m = HyperVLA.from_synthetic(FULL, max_batch=B)
im = syn.synthetic_images(B, FULL)
lib = m._ctx.lib
lib.hvla_debug_lnt_stats.argtypes = [C.c_void_p, C.POINTER(C.c_ulonglong), C.c_int]
buf = (C.c_ulonglong * (4 * 256))()
for _ in range(3):
    m.encode_images(im)
assert lib.hvla_debug_lnt_stats(m._ctx.h, buf, 1) == 0
steps = 5
for _ in range(steps):
    m.encode_images(im)
assert lib.hvla_debug_lnt_stats(m._ctx.h, buf, 1) == 0
a = np.array(list(buf), dtype=np.float64).reshape(4, 256)
launches = a[3].max()
print(f"B = {B}: {launches:.0f} tail-carrying launches in {steps} steps; per workgroup and launch:")
jobs = a[0] / launches
print(f"  jobs         mean {jobs.mean():.3f}  min {jobs.min():.3f}  max {jobs.max():.3f}")
print(f"  histogram of jobs per workgroup summed over the launches: {np.bincount(a[0].astype(int))[:]}")
tj = a[1] / np.maximum(a[0], 1)
print(f"  ticks per job (100 MHz shader-clock counter? printed raw) mean {tj[a[0] > 0].mean():.0f}  max {tj.max():.0f}")
print(f"  ticks in jobs per launch   mean {(a[1] / launches).mean():.0f}  max {(a[1] / launches).max():.0f}")
print(f"  ticks in protocol / launch mean {(a[2] / launches).mean():.0f}  max {(a[2] / launches).max():.0f}")
