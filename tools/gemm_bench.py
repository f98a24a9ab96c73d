"""Diagnostic (GPU box): time the encoder's four GEMM shapes in isolation through hvla_debug_gemm.

    python tools/gemm_bench.py [B]        HVLA_VARIANTS=3,2,1,0 selects kernels: 3 gemm256p_kernel persistent (production),
                                          2 gemm256p_kernel one workgroup per tile, 1 gemm64_kernel, 0 gemm_kernel (128x128)
M = B x 256 rows (whole image-aligned tiles, as in the encoder)."""
import ctypes as C, os, sys
os.environ["HVLA_LIBRARY_FLAVOUR"] = "bench"        # libhvla_bench.so: the product library has no hvla_debug_* entry points
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "hyper-vla_amd")); sys.path.insert(0, ROOT)
import torch
from hypervla.config import FULL
from hypervla.model import HyperVLA
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
m = HyperVLA.from_synthetic(FULL, max_batch=B)
lib = m._ctx.lib
lib.hvla_debug_gemm.argtypes = [C.c_void_p] + [C.c_int] * 6 + [C.POINTER(C.c_float)]
M = B * 256
shapes = {"qkv": (M, 2304, 768, 1), "out": (M, 768, 768, 3), "fc1": (M, 3072, 768, 2), "fc2": (M, 768, 3072, 3)}
names = {0: "128x128 register-staged", 1: "64x64 (small M)", 2: "256x256, one WG per tile", 3: "256x256 persistent"}
for nm, (M_, N, K, epi) in shapes.items():
    for variant in (tuple(int(v) for v in os.environ["HVLA_VARIANTS"].split(",")) if "HVLA_VARIANTS" in os.environ else (3, 2)):
        ms = C.c_float()
        rc = lib.hvla_debug_gemm(m._ctx.h, M_, N, K, epi, variant, 20, C.byref(ms))
        tf = 2.0 * M_ * N * K / (ms.value * 1e-3) / 1e12
        print(f"{nm:4s} M={M_} N={N} K={K} {names[variant]:26s} rc={rc} {ms.value*1e3:8.1f} us  {tf:7.1f} TF/s")
