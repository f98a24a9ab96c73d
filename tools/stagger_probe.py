"""Diagnostic (GPU box): start-stagger sweep of the phased GEMM (first-round workgroups of later XCD groups start late)."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "hyper-vla_amd")); sys.path.insert(0, ROOT)
from hypervla.config import FULL
from hypervla.model import HyperVLA
B = 256
m = HyperVLA.from_synthetic(FULL, max_batch=B)
lib = m._ctx.lib
lib.hvla_debug_gemm.argtypes = [C.c_void_p] + [C.c_int] * 6 + [C.POINTER(C.c_float)]
M = B * 257 - 256          # whole rounds: 256 M-tiles
shapes = {"qkv": (M, 2304, 768, 1), "out": (M, 768, 768, 3), "fc1": (M, 3072, 768, 2), "fc2": (M, 768, 3072, 3)}
def run(nm):
    M_, N, K, epi = shapes[nm]
    ms = C.c_float()
    best = 1e9
    for _ in range(3):
        lib.hvla_debug_gemm(m._ctx.h, M_, N, K, epi, 9, 20, C.byref(ms))
        best = min(best, ms.value * 1e3)
    return best
for nm in shapes:
    os.environ.pop("HVLA_STAGGER", None)
    base = run(nm)
    out = [f"{nm}: base {base:.1f} us |"]
    for groups in (2, 4, 8):
        os.environ["HVLA_STAGGER_GROUPS"] = str(groups)
        for us in (8, 16, 24, 32, 48, 64, 96):
            os.environ["HVLA_STAGGER"] = str(us * 100)
            out.append(f"g{groups}/{us}us {run(nm):.1f}")
        out.append("|")
    print(" ".join(out), flush=True)
