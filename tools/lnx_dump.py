"""Diagnostic (GPU box): a ONE-layer DINOv2 geometry; read the workspace back (libhvla_bench.so, hvla_debug_workspace) after a
batch of B images (norm2 inside the out-projection's epilogue: gemm256p_kernel<..., LNX>) and after each image alone (stand-alone
LayerNorm kernels): the 16-bit h = norm2(x) rows and the mean rows must be the same bytes.

    python tools/lnx_dump.py [B]
"""
import ctypes as C, os, sys, dataclasses
os.environ["HVLA_LIBRARY_FLAVOUR"] = "bench"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "hyper-vla_amd")); sys.path.insert(0, ROOT)
import numpy as np, torch
from hypervla import synthetic as syn
from hypervla.config import FULL
from hypervla.model import HyperVLA

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
G = dataclasses.replace(FULL, enc_layers=1)
m = HyperVLA.from_synthetic(G, max_batch=B)
lib = m._ctx.lib
lib.hvla_debug_workspace.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]
hip = C.CDLL("libamdhip64.so")

def grab(which, nbytes, dtype):
    p, n = C.c_void_p(), C.c_size_t()
    assert lib.hvla_debug_workspace(m._ctx.h, which, C.byref(p), C.byref(n)) == 0
    torch.cuda.synchronize()
    out = np.empty(nbytes, np.uint8)
    assert hip.hipMemcpy(out.ctypes.data_as(C.c_void_p), p, C.c_size_t(nbytes), 2) == 0
    return out.view(dtype)

S, E = G.patches + 1, G.enc_dim
im = syn.synthetic_images(B, G)[:, 0]
m.encode_images(im)
h_b = grab(1, B * S * E * 2, np.float16).reshape(B, S, E).copy()
ab_b = grab(5, B * 2 * G.enc_mlp * 2, np.float16).copy()
part = grab(7, B * 3 * 256 * 16, np.uint32).reshape(B, 3, 256, 4).copy()
cnt = grab(6, B * 4, np.uint32).copy()
print("ln_cnt", cnt[:8], "tags", np.unique(part[..., 1]), np.unique(part[..., 3]))
for i in range(min(B, 3)):
    m.encode_images(im[i:i + 1])
    h_1 = grab(1, S * E * 2, np.float16).reshape(S, E).copy()
    d = np.abs(h_b[i].astype(np.float32) - h_1.astype(np.float32))
    rows = np.where(d.max(axis=1) > 0)[0]
    print(f"image {i}: rows that differ {len(rows)} of {S}; first {rows[:8]}; max |diff| {d.max():.4f}; CLS row diff {d[0].max():.4f}; "
          f"max |h| batch {np.abs(h_b[i]).max():.3f} alone {np.abs(h_1).max():.3f}")
    if len(rows):
        r = rows[min(1, len(rows) - 1)]
        ratio = h_b[i, r].astype(np.float32) / np.where(h_1[r] == 0, 1, h_1[r]).astype(np.float32)
        print(f"   row {r}: ratio batch/alone by 256-column block {[float(np.median(ratio[c * 256:(c + 1) * 256])) for c in range(3)]}")
        sq = part[i, :, r - 1 if r else 0, :]
        print("   entries of that row:", sq.view(np.float32)[:, [0, 2]].tolist())
