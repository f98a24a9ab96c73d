"""Diagnostic (GPU box): where a workgroup's time goes inside attention_kernel -- shader-clock stamps of wave 0 at the phase
boundaries (libhvla_bench.so, hvla_debug_attention_stamps) for a few workgroups of a B x 12 launch.

    python tools/attention_timeline.py [B]
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
g = FULL
m = HyperVLA.from_synthetic(g, max_batch=B)
m.encode_images(syn.synthetic_images(B, g))          # leaves real q / k / v in the workspace
lib = m._ctx.lib
n_wg = B * g.enc_heads
wgs = [0, 1, 255, 511, 512, 1024, n_wg // 2, n_wg - 513, n_wg - 1]
wgs = [w for w in wgs if 0 <= w < n_wg]
arr = (C.c_int32 * len(wgs))(*wgs)
out = (C.c_ulonglong * (8 * len(wgs)))()
lib.hvla_debug_attention_stamps.argtypes = [C.c_void_p, C.c_int32, C.POINTER(C.c_int32), C.c_int32, C.POINTER(C.c_ulonglong)]
rc = lib.hvla_debug_attention_stamps(m._ctx.h, B, arr, len(wgs), out)
assert rc == 0, rc
t = np.array(list(out), dtype=np.float64).reshape(len(wgs), 8)
names = ["requests issued (LDS-DMA, q)", "address set-up", "wait for chunk 0", "pass over the keys (+ 4 hand-overs)", "normalise + column sums + stores", "last query partials", "combine + end"]
print(f"B = {B}: {n_wg} workgroups; clock ticks per phase (wave 0)")
print("workgroup".ljust(10) + "".join(n[:22].rjust(24) for n in names) + "total".rjust(10))
for i, w in enumerate(wgs):
    d = np.diff(t[i])
    print(str(w).ljust(10) + "".join(f"{v:24.0f}" for v in d) + f"{t[i, -1] - t[i, 0]:10.0f}")
d = np.diff(t, axis=1).mean(0)
print("mean share".ljust(10) + "".join(f"{100 * v / d.sum():23.1f}%" for v in d))
