"""Diagnostic (GPU box): attention_kernel's two mean rows per (image, head) -- the out-projection's compensation operand -- against
the means of the output rows it stored (libhvla_bench.so: hvla_debug_attention_stamps re-runs the launch on the workspace's q / k / v,
hvla_debug_workspace reads o and the mean rows back).  The kernel adds the f32 outputs up (in-lane, then over the waves of a half),
this check the 16-bit rows it stored: agreement to a few 1e-3 of the row scale is what the arithmetic allows.

    python tools/attention_omean_check.py [B]
"""
import ctypes as C, os, sys
os.environ["HVLA_LIBRARY_FLAVOUR"] = "bench"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "hyper-vla_amd")); sys.path.insert(0, ROOT)
import numpy as np, torch
from hypervla import synthetic as syn
from hypervla.config import FULL
from hypervla.model import HyperVLA

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
g = FULL
m = HyperVLA.from_synthetic(g, max_batch=B)
m.encode_images(syn.synthetic_images(B, g))          # leaves the last layer's q / k / v in the workspace
lib = m._ctx.lib
lib.hvla_debug_workspace.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]
lib.hvla_debug_attention_stamps.argtypes = [C.c_void_p, C.c_int32, C.POINTER(C.c_int32), C.c_int32, C.POINTER(C.c_ulonglong)]
hip = C.CDLL("libamdhip64.so")
arr = (C.c_int32 * 1)(0)
out = (C.c_ulonglong * 8)()
assert lib.hvla_debug_attention_stamps(m._ctx.h, B, arr, 1, out) == 0


def grab(which, nbytes, dtype):
    p, n = C.c_void_p(), C.c_size_t()
    assert lib.hvla_debug_workspace(m._ctx.h, which, C.byref(p), C.byref(n)) == 0
    torch.cuda.synchronize()
    buf = np.empty(nbytes, np.uint8)
    assert hip.hipMemcpy(buf.ctypes.data_as(C.c_void_p), p, C.c_size_t(nbytes), 2) == 0
    return buf.view(dtype)


S, E = g.patches + 1, g.enc_dim
o = grab(1, B * S * E * 2, np.float16).reshape(B, S, E).astype(np.float64)
om = grab(5, B * 2 * E * 2, np.float16).reshape(B, 2, E).astype(np.float64)
NW = (S - 1) // 32
NH = NW // 2
want = np.stack([o[:, :NH * 32].mean(1), o[:, NH * 32:].mean(1)], 1)          # the kernel's split: tokens [0, 32 NW / 2) and the rest
d = np.abs(om - want)
scale = np.abs(want).max()
print(f"B = {B}: mean rows max |diff| {d.max():.3e} (row scale {scale:.3f}); worst (image, half, column) {np.unravel_index(d.argmax(), d.shape)}")
print(f"   mean |diff| {d.mean():.3e}; values above 1e-3: {(d > 1e-3).sum()} of {d.size}; per half max {d.max(axis=(0, 2))}; "
      f"per head max {d.reshape(B, 2, -1, 64).max(axis=(0, 1, 3)).round(4).tolist()}")
i, hf, c = np.unravel_index(d.argmax(), d.shape)
print(f"   worst: kernel {om[i, hf, c]:.5f} rows {want[i, hf, c]:.5f}; the column's |o| max {np.abs(o[i, :, c]).max():.3f}")
print("ok" if d.max() < 4e-3 * max(scale, 1.0) else "MISMATCH")
