"""Diagnostic (GPU box): what the LayerNorm inside the residual GEMMs' epilogue (gemm256p_kernel<..., LNX>) costs per tile
(libhvla_bench.so, hvla_debug_lnx_stats): shader-clock ticks of the epilogue's phases per workgroup and tile, how long a tile
waits for the image's other column tiles, how many tiles are left to the last arriver.

    python tools/lnx_stats.py [B] [spin ticks of 10 ns]
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
lib.hvla_debug_lnx_stats.argtypes = [C.c_void_p, C.POINTER(C.c_ulonglong), C.c_int]
lib.hvla_debug_lnx_spin.argtypes = [C.c_void_p, C.c_uint32]
if len(sys.argv) > 2:
    assert lib.hvla_debug_lnx_spin(m._ctx.h, int(sys.argv[2])) == 0
buf = (C.c_ulonglong * (8 * 256))()
for _ in range(3):
    m.encode_images(im)
assert lib.hvla_debug_lnx_stats(m._ctx.h, buf, 1) == 0
steps = 5
for _ in range(steps):
    m.encode_images(im)
assert lib.hvla_debug_lnx_stats(m._ctx.h, buf, 1) == 0
a = np.array(list(buf), dtype=np.float64).reshape(8, 256)
tiles = a[0]
n = np.maximum(tiles, 1)
print(f"B = {B}: {tiles.sum():.0f} tiles in {steps} steps ({tiles.sum() / steps / 25:.0f} per launch); shader-clock ticks per tile and workgroup (mean / max over workgroups):")
for name, row in (("whole epilogue", 1), ("[A] x into registers", 5), ("[B] statistics + publish + drain", 6), ("[C] wait for the partners (wave 0)", 2),
                  ("[D-F] mean / rstd, normalise, store h", 7)):
    v = a[row] / n
    print(f"  {name:42s} {v[tiles > 0].mean():9.0f} {v.max():9.0f}")
print(f"  tiles abandoned {a[3].sum():.0f}, normalised from memory by the last arriver {a[4].sum():.0f}")
w = a[2] / n
print("  wait by workgroup id (first 40):", np.round(w[:40]).astype(int).tolist())
slow = a[4]
print("  tiles normalised from memory, by workgroup id (nonzero):", {int(i): int(v) for i, v in enumerate(slow) if v})
