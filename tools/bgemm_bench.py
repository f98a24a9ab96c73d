"""Diagnostic (GPU box): time shapes of the fine-tune path's batched split-bf16 GEMM in isolation (hvla_debug_bgemm)."""
import ctypes as C, os, sys
os.environ["HVLA_LIBRARY_FLAVOUR"] = "bench"        # libhvla_bench.so: the product library has no hvla_debug_* entry points
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "hyper-vla_amd"))
import torch
from hypervla import _native
lib = _native.load_library()
lib.hvla_debug_bgemm.argtypes = [C.c_void_p] * 3 + [C.c_int] * 8 + [C.POINTER(C.c_float)]
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 8224
shapes = [("NN 768", rows, 768, 768, 0, 0, 0), ("NT 768", rows, 768, 768, 0, 1, 0), ("TN 768 (dW, split-K)", 768, 768, rows, 1, 0, 1),
          ("NN fc1", rows, 3072, 768, 0, 0, 0), ("NT fc2-dX", rows, 3072, 768, 0, 1, 0), ("TN fc dW", 768, 3072, rows, 1, 0, 1),
          ("NN fc2", rows, 768, 3072, 0, 0, 0)]
for nm, M, N, K, ta, tb, acc in shapes:
    a = torch.randn(M * K, device="cuda"); b = torch.randn(N * K, device="cuda"); c = torch.zeros(M * N, device="cuda")
    ms = C.c_float()
    rc = lib.hvla_debug_bgemm(a.data_ptr(), b.data_ptr(), c.data_ptr(), M, N, K, ta, tb, 1, acc, 20, C.byref(ms))
    print(f"{nm:24s} M={M} N={N} K={K} rc={rc} {ms.value*1e3:8.1f} us {2.0*M*N*K/ms.value/1e9:7.1f} TF/s (f32-equivalent)")
