"""Diagnostic (GPU box): the SECOND-ORDER inputs of the encoder -- the mean rows and the bias-row (`corr`) tables of the weight-rounding
compensation -- checked directly against the rows / operands they are made from.  A wrong mean row moves the
actions by a second-order term that the end-to-end tolerances forgive (round 5: a reduce-scatter that lost a sixteenth of the rows was
4e-3 wrong in the mean rows and passed every parity test), so these tables have checks of their own:

  1. ln_abar   the two mean rows per image that the LayerNorm fused into a residual GEMM's epilogue writes (gemm256p_kernel<..., LNX>)
               against the column means of the 16-bit `h` rows the SAME launch stored (tokens 1..128 | 129..256, CLS excluded)
  2. corr      the table gemm64_kernel's second problem writes from those mean rows ([B][2][3E] for the QKV product): against
               bias + abar . dW / 4096 in float64, dW = the 16-bit rounding residue of the weights restated from the f32 parameters
  3. colmean   the GELU epilogue's two column means per image of its rounded outputs (fc2's compensation operand) against the
               column means of the `g` rows the launch stored
  4. corr      fc2's table against bias + colmean . dW2 / 4096

libhvla_bench.so only: hvla_debug_encode_stop returns from hvla_encode behind the n-th dense product, hvla_debug_workspace hands out
the workspace buffers.  No oracle involved: every check is the kernel's output against the kernel's own inputs.

    python tools/second_order_check.py [B] [f16|bf16]
"""
import ctypes as C
import os
import sys

os.environ["HVLA_LIBRARY_FLAVOUR"] = "bench"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "hyper-vla_amd"))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402
from hypervla import synthetic as syn  # noqa: E402
from hypervla.config import FULL, shared_name  # noqa: E402
from hypervla.model import HyperVLA  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
dt = sys.argv[2] if len(sys.argv) > 2 else "f16"
g = FULL
m = HyperVLA.from_synthetic(g, max_batch=B, enc_dtype=dt)
lib = m._ctx.lib
lib.hvla_debug_workspace.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]
lib.hvla_debug_encode_stop.argtypes = [C.c_void_p, C.c_int32]
hip = C.CDLL("libamdhip64.so")
S, E, F, P = g.patches + 1, g.enc_dim, g.enc_mlp, g.patches
images = syn.synthetic_images(B, g)


def to64(raw):
    """16-bit device values -> float64 (fp16, or bf16 = the upper half of an f32)."""
    if dt == "f16":
        return raw.view(np.float16).astype(np.float64)
    return (raw.view(np.uint16).astype(np.uint32) << 16).view(np.float32).astype(np.float64)


def round16(x):
    """float32 -> the operand type and back (round to nearest even), as csrc/pack.h does."""
    x = np.asarray(x, np.float32)
    if dt == "f16":
        with np.errstate(over="ignore"):
            return x.astype(np.float16).astype(np.float32)
    u = x.view(np.uint32).astype(np.uint64)
    u = ((u + 0x7FFF + ((u >> 16) & 1)) >> 16) << 16
    return u.astype(np.uint32).view(np.float32)


def grab(which, nbytes):
    p, n = C.c_void_p(), C.c_size_t()
    assert lib.hvla_debug_workspace(m._ctx.h, which, C.byref(p), C.byref(n)) == 0 and n.value >= nbytes, (which, n.value, nbytes)
    torch.cuda.synchronize()
    buf = np.empty(nbytes, np.uint8)
    assert hip.hipMemcpy(buf.ctypes.data_as(C.c_void_p), p, C.c_size_t(nbytes), 2) == 0
    return buf


def run_until(n):
    assert lib.hvla_debug_encode_stop(m._ctx.h, n) == 0
    m.encode_images(images)
    torch.cuda.synchronize()


def residue(kernels):
    """[N][K] float64 of the device's dW: (W - W16) x 4096 rounded to the operand type, flax [K][N] kernels side by side along N."""
    w = np.concatenate([np.asarray(k, np.float32) for k in kernels], axis=1)
    return round16((w - round16(w)) * np.float32(4096.0)).astype(np.float64).T


def leaf(layer, *path, shape=None):
    a = np.asarray(m.params[shared_name(("encoder", "layer", str(layer)) + path)])      # (synthetic parameters are stored flat)
    return a.reshape(shape) if shape is not None else a


def half_means(rows):
    """[B][S][N] -> [B][2][N]: means over tokens 1 .. P/2 and P/2 + 1 .. P (the CLS row belongs to neither)."""
    return np.stack([rows[:, 1:1 + P // 2].mean(1), rows[:, 1 + P // 2:].mean(1)], 1)


ok = True


def report(name, got, want, tol):
    global ok
    d = np.abs(got - want)
    scale = max(np.abs(want).max(), 1e-30)
    good = d.max() <= tol * max(scale, 1.0)
    ok &= bool(good)
    i = np.unravel_index(d.argmax(), d.shape)
    print(f"{name}: max |diff| {d.max():.3e} (scale {scale:.3f}, bound {tol * max(scale, 1.0):.1e}) at {tuple(int(v) for v in i)}: "
          f"kernel {got[i]:.6f} expected {want[i]:.6f}; mean |diff| {d.mean():.2e} {'ok' if good else 'MISMATCH'}")


# ---- 1 + 2: behind the QKV product of layer 1 -- h = norm1 of layer 1 (written by layer 0's fc2 epilogue), abar = its mean rows,
# corr = the QKV table made from them
run_until(5)
h = to64(grab(1, B * S * E * 2)).reshape(B, S, E)
abar = to64(grab(5, B * 2 * E * 2)).reshape(B, 2, E)
corr = grab(4, B * 2 * 3 * E * 4).view(np.float32).astype(np.float64).reshape(B, 2, 3 * E)
eps16 = 2.0 ** -11 if dt == "f16" else 2.0 ** -8
# the kernel adds the f32 values up and rounds the mean once; the rows it stored are rounded one by one: the two differ by the mean of
# P/2 independent roundings of values of the rows' size plus one rounding of the mean
report("ln_abar vs the column means of the h rows the same launch stored", abar, half_means(h), 2.0 * eps16)
dqkv = residue([leaf(1, "attention", "attention", nm, "kernel", shape=(E, E)) for nm in ("query", "key", "value")])
bqkv = np.concatenate([leaf(1, "attention", "attention", nm, "bias") for nm in ("query", "key", "value")]).astype(np.float64)
report("corr (QKV) vs bias + abar . dW / 4096 in float64", corr, bqkv + abar @ dqkv.T / 4096.0, 5e-8)
# the size of what the table corrects, for scale: how far it is from the plain bias
print(f"   (the table moves the bias by up to {np.abs(corr - bqkv).max():.2e}; a table that ignored the mean rows would be off by that)")

# ---- 3 + 4: the whole encoder -- g = the last layer's GELU outputs, abar = their column means, corr = fc2's table
run_until(0)
L = g.enc_layers - 1
gl = to64(grab(3, B * S * F * 2)).reshape(B, S, F)
cm = to64(grab(5, B * 2 * F * 2)).reshape(B, 2, F)
corr2 = grab(4, B * 2 * E * 4).view(np.float32).astype(np.float64).reshape(B, 2, E)
# the epilogue adds the ROUNDED outputs up in the operand type itself (two v_pk_add_f16 per row; colmean_kernel restates it): the
# sum of 32 values per lane carries the operand type's rounding at every step, then f32 over the lanes / waves
report("GELU column means vs the column means of the g rows the same launch stored", cm, half_means(gl), 16.0 * eps16)
dw2 = residue([leaf(L, "mlp", "fc2", "kernel", shape=(F, E))])
b2 = leaf(L, "mlp", "fc2", "bias").astype(np.float64)
report("corr (fc2) vs bias + colmean . dW2 / 4096 in float64", corr2, b2 + cm @ dw2.T / 4096.0, 5e-8)
print(f"   (the table moves the bias by up to {np.abs(corr2 - b2).max():.2e})")
print("ok" if ok else "MISMATCH")
sys.exit(0 if ok else 1)
