"""Diagnostic (GPU box): time the encoder GEMM shapes / ablations through hvla_debug_gemm."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "hyper-vla_amd")); sys.path.insert(0, ROOT)
import torch
from hypervla.config import FULL
from hypervla.model import HyperVLA
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
m = HyperVLA.from_synthetic(FULL, max_batch=B)
lib = m._ctx.lib
lib.hvla_debug_gemm.argtypes = [C.c_void_p] + [C.c_int] * 6 + [C.POINTER(C.c_float)]
M = B * 257 - int(os.environ.get('HVLA_DBG_MSHRINK', '0'))
shapes = {"qkv": (M, 2304, 768, 1), "out": (M, 768, 768, 3), "fc1": (M, 3072, 768, 2), "fc2": (M, 768, 3072, 3)}
names = {0: "128x128 regstage", 4: "ring5 8 waves", 5: "256 ring4", 6: "ring4 no-DMA", 7: "ring4 no-MFMA", 8: "ring4 DMA-only", 9: "phased, peeled tail", 10: "phased, clamped tail", 11: "phased no stagger", 12: "phased with setprio", 13: "phased no DMA in loop", 14: "phased no MFMA", 15: "phased no frag reads", 16: "phased barriers only", 17: "phased MFMA only", 18: "phased frag reads only", 19: "phased DMA only", 20: "phased MFMA only, no stagger", 21: "phased no epilogue", 22: "phased old epilogue (QKV)", 23: "phased stores kept in L2", 24: "phased old epilogue (GELU)", 25: "phased old epilogue (RES)", 27: "phased, persistent"}
for nm, (M_, N, K, epi) in shapes.items():
    for variant in (tuple(int(v) for v in os.environ["HVLA_VARIANTS"].split(",")) if "HVLA_VARIANTS" in os.environ else (5, 4)):
        if 6 <= variant <= 8 and nm != "qkv" and nm != "fc2":
            continue
        if 11 <= variant <= 23 and nm != "qkv":
            continue
        if (variant == 24 and nm != "fc1") or (variant == 25 and nm not in ("out", "fc2")):
            continue
        ms = C.c_float()
        e = 1 if (6 <= variant <= 8 or 11 <= variant <= 23) else epi
        if 'HVLA_DBG_EPI' in os.environ: e = int(os.environ['HVLA_DBG_EPI'])
        rc = lib.hvla_debug_gemm(m._ctx.h, M_, N, K, e, variant, 20, C.byref(ms))
        tf = 2.0 * M_ * N * K / (ms.value * 1e-3) / 1e12
        print(f"{nm:4s} M={M_} N={N} K={K} {names[variant]:20s} rc={rc} {ms.value*1e3:8.1f} us  {tf:7.1f} TF/s")
