"""ctypes binding of libhvla (include/hvla.h).  There is no CPU or eager fallback: if the HIP library
is missing or the device is not gfx950 every entry point raises."""
from __future__ import annotations

import ctypes as C
import os
from typing import Dict, Optional

import numpy as np

_LIB_DIR = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "lib")
# HVLA_LIBRARY_FLAVOUR=bench: tools/ load libhvla_bench.so (the same sources with the hvla_debug_* timing hooks compiled in)
_LIB_PATH = os.path.join(_LIB_DIR, "libhvla_bench.so" if os.environ.get("HVLA_LIBRARY_FLAVOUR") == "bench" else "libhvla.so")

HVLA_ENC_F16, HVLA_ENC_BF16 = 0, 1
_ERRORS = {-1: "HVLA_E_SHAPE", -2: "HVLA_E_DTYPE", -3: "HVLA_E_DEVICE", -4: "HVLA_E_ARENA_FULL",
           -5: "HVLA_E_HIP", -6: "HVLA_E_WEIGHTS", -7: "HVLA_E_STATE"}

EXPORTS = ["hvla_create", "hvla_destroy", "hvla_last_error", "hvla_load_weights", "hvla_num_generated",
           "hvla_generate", "hvla_weights_free", "hvla_release_pooled_arenas", "hvla_weights_batch", "hvla_weights_export",
           "hvla_encode", "hvla_policy", "hvla_step", "hvla_ensemble_reset", "hvla_ensemble",
           "hvla_selftest", "hvla_profile", "hvla_profile_read", "hvla_loss",
           "hvla_train_sizes", "hvla_train_step", "hvla_train_apply", "hvla_encode_hidden", "hvla_t5_load",
           "hvla_t5_encode", "hvla_preprocess", "hvla_encode_audit", "hvla_train_accumulate", "hvla_train_bucket_ranges",
           "hvla_train_wait_bucket", "hvla_set_attention_outputs", "hvla_train_profile", "hvla_train_profile_read",
           "hvla_launches", "hvla_box_probe", "hvla_profile_select"]
PROF_NAMES = ["patch_embed", "layernorm", "qkv_gemm", "attention", "out_gemm", "fc1_gemm", "fc2_gemm", "policy",
              "small_row_gemms"]      # mean rows + the 2 B latency-bound rows per GEMM: CLS rows and weight-rounding compensation rows


class hvla_config(C.Structure):
    _fields_ = [("struct_size", C.c_uint32)] + \
               [(n, C.c_int32) for n in ("image_size", "patch", "enc_dim", "enc_layers", "enc_heads",
                                         "enc_mlp", "dim", "layers", "heads", "mlp", "horizon",
                                         "action_dim")] + \
               [("tanh_scale", C.c_float), ("max_action", C.c_float)] + \
               [(n, C.c_int32) for n in ("ctx_dim", "ctx_layers", "ctx_heads", "ctx_mlp", "lang_tokens",
                                         "lang_dim", "scale_context", "max_batch", "enc_dtype", "streams", "clip_target")]


class hvla_tensor_desc(C.Structure):
    _fields_ = [("name", C.c_char_p), ("data", C.POINTER(C.c_float)), ("numel", C.c_int64)]


class hvla_t5_config(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("vocab", "d_model", "d_kv", "heads", "d_ff", "layers", "buckets", "max_distance")] + \
               [("eps", C.c_float), ("max_tokens", C.c_int32), ("max_batch", C.c_int32)]


class hvla_train_buffers(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("params", "grads", "mu", "nu", "ema", "theta", "dtheta", "work", "loss",
                                          "actions", "logits", "sqsum", "wd_mask", "params0")]


class hvla_train_hyper(C.Structure):
    _fields_ = [(n, C.c_float) for n in ("lr", "b1", "b2", "eps", "weight_decay", "clip", "ema_decay")] + \
               [("step", C.c_int32), ("forward_only", C.c_int32), ("base_lr", C.c_float),
                ("base_weight_decay", C.c_float), ("train_encoder", C.c_int32)]


_lib = None


def lib_path() -> str:
    return _LIB_PATH


def load_library():
    """dlopen libhvla.so and declare every prototype of include/hvla.h."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(_LIB_PATH):
        raise RuntimeError(f"libhvla.so not built at {_LIB_PATH}: run `python -c 'import __graft_entry__ as g; "
                           f"g.build()'` (make -C hyper-vla_amd/csrc). There is no CPU fallback.")
    lib = C.CDLL(_LIB_PATH)
    vp, i32, i64 = C.c_void_p, C.c_int32, C.c_int64
    lib.hvla_create.argtypes = [C.POINTER(hvla_config), C.c_int, C.POINTER(vp)]
    lib.hvla_create.restype = C.c_int
    lib.hvla_destroy.argtypes = [vp]
    lib.hvla_destroy.restype = None
    lib.hvla_last_error.argtypes = [vp]
    lib.hvla_last_error.restype = C.c_char_p
    lib.hvla_load_weights.argtypes = [vp, C.POINTER(hvla_tensor_desc), i32]
    lib.hvla_load_weights.restype = C.c_int
    lib.hvla_num_generated.argtypes = [vp]
    lib.hvla_num_generated.restype = i64
    lib.hvla_generate.argtypes = [vp, vp, vp, vp, i32, C.POINTER(vp), vp]
    lib.hvla_generate.restype = C.c_int
    lib.hvla_weights_free.argtypes = [vp, vp]
    lib.hvla_weights_free.restype = C.c_int
    lib.hvla_release_pooled_arenas.argtypes = [vp]
    lib.hvla_release_pooled_arenas.restype = C.c_int
    lib.hvla_weights_batch.argtypes = [vp]
    lib.hvla_weights_batch.restype = i32
    lib.hvla_launches.argtypes = [vp]
    lib.hvla_launches.restype = i64
    lib.hvla_box_probe.argtypes = [vp, C.POINTER(C.c_float), vp]
    lib.hvla_box_probe.restype = C.c_int
    lib.hvla_weights_export.argtypes = [vp, vp, vp, vp, vp]
    lib.hvla_weights_export.restype = C.c_int
    lib.hvla_encode.argtypes = [vp, vp, vp, i32, vp]
    lib.hvla_encode.restype = C.c_int
    lib.hvla_encode_hidden.argtypes = [vp, vp, vp, i32, vp]
    lib.hvla_encode_hidden.restype = C.c_int
    lib.hvla_preprocess.argtypes = [vp, vp, i32, i32, i32, i32, vp, vp]
    lib.hvla_preprocess.restype = C.c_int
    lib.hvla_t5_load.argtypes = [vp, C.POINTER(hvla_t5_config), C.POINTER(hvla_tensor_desc), i32]
    lib.hvla_t5_load.restype = C.c_int
    lib.hvla_t5_encode.argtypes = [vp, vp, vp, vp, i32, i32, vp]
    lib.hvla_t5_encode.restype = C.c_int
    lib.hvla_policy.argtypes = [vp, vp, vp, vp, vp, i32, vp]
    lib.hvla_policy.restype = C.c_int
    lib.hvla_step.argtypes = [vp, vp, vp, vp, vp, i32, vp]
    lib.hvla_step.restype = C.c_int
    lib.hvla_ensemble_reset.argtypes = [vp, vp, vp]
    lib.hvla_ensemble_reset.restype = C.c_int
    lib.hvla_ensemble.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp]
    lib.hvla_ensemble.restype = C.c_int
    lib.hvla_loss.argtypes = [vp, vp, vp, vp, vp, vp, vp, i32, vp]
    lib.hvla_loss.restype = C.c_int
    lib.hvla_train_sizes.argtypes = [vp, i32, i32, C.POINTER(i64)]
    lib.hvla_train_sizes.restype = C.c_int
    lib.hvla_train_step.argtypes = [vp, C.POINTER(hvla_train_buffers), vp, vp, vp, vp, vp, vp, vp, vp, i32,
                                    C.POINTER(hvla_train_hyper), vp]
    lib.hvla_train_step.restype = C.c_int
    lib.hvla_train_apply.argtypes = [vp, C.POINTER(hvla_train_buffers), C.POINTER(hvla_train_hyper), vp]
    lib.hvla_train_accumulate.argtypes = [vp, C.POINTER(hvla_train_buffers), vp, C.c_float, C.POINTER(hvla_train_hyper), vp]
    lib.hvla_train_accumulate.restype = C.c_int
    lib.hvla_train_bucket_ranges.argtypes = [vp, i32, C.POINTER(C.c_int64)]
    lib.hvla_train_bucket_ranges.restype = C.c_int
    lib.hvla_train_wait_bucket.argtypes = [vp, i32, vp]
    lib.hvla_train_wait_bucket.restype = C.c_int
    lib.hvla_train_profile.argtypes = [vp, i32]
    lib.hvla_train_profile.restype = C.c_int
    lib.hvla_train_profile_read.argtypes = [vp, C.POINTER(C.c_float), C.POINTER(C.c_double), C.POINTER(i32)]
    lib.hvla_train_profile_read.restype = C.c_int
    lib.hvla_train_apply.restype = C.c_int
    lib.hvla_profile.argtypes = [vp, i32]
    lib.hvla_profile.restype = C.c_int
    lib.hvla_profile_select.argtypes = [vp, C.c_uint32]
    lib.hvla_profile_select.restype = C.c_int
    lib.hvla_profile_read.argtypes = [vp, C.POINTER(C.c_float), C.POINTER(i32)]
    lib.hvla_profile_read.restype = C.c_int
    lib.hvla_encode_audit.argtypes = [vp, vp, i32, C.POINTER(C.c_float), C.POINTER(i32), vp]
    lib.hvla_encode_audit.restype = C.c_int
    lib.hvla_set_attention_outputs.argtypes = [vp, vp, vp]
    lib.hvla_set_attention_outputs.restype = C.c_int
    lib.hvla_selftest.argtypes = [vp, vp]
    lib.hvla_selftest.restype = C.c_int
    _lib = lib
    return lib


class NativeError(RuntimeError):
    pass


class Context:
    """One hvla_ctx (one device)."""

    def __init__(self, geometry, device: int = 0, max_batch: int = 256, enc_dtype: str = "f16", streams: int = 1):
        self.lib = load_library()
        g = geometry
        if enc_dtype not in ("f16", "bf16"):
            raise ValueError("enc_dtype must be 'f16' or 'bf16'")
        self.cfg = hvla_config(C.sizeof(hvla_config), g.image_size, g.patch, g.enc_dim, g.enc_layers, g.enc_heads, g.enc_mlp,
                               g.dim, g.layers, g.heads, g.mlp, g.horizon, g.action_dim,
                               g.tanh_scale, g.max_action, g.ctx_dim, g.ctx_layers, g.ctx_heads, g.ctx_mlp,
                               g.lang_tokens, g.lang_dim, int(g.scale_context), int(max_batch),
                               HVLA_ENC_BF16 if enc_dtype == "bf16" else HVLA_ENC_F16, int(streams),
                               int(getattr(g, "clip_target", True)))
        self.geometry, self.device, self.max_batch, self.enc_dtype = g, device, max_batch, enc_dtype
        h = C.c_void_p()
        rc = self.lib.hvla_create(C.byref(self.cfg), device, C.byref(h))
        if rc != 0:
            raise NativeError(f"hvla_create failed: {_ERRORS.get(rc, rc)} (geometry outside the hand-written "
                              f"kernels' specialisation, or device {device} is not a gfx950 GPU)")
        self.h = h

    def _check(self, rc: int, what: str):
        if rc != 0:
            msg = self.lib.hvla_last_error(self.h)
            raise NativeError(f"{what}: {_ERRORS.get(rc, rc)}: {msg.decode() if msg else ''}")

    def close(self):
        if getattr(self, "h", None):
            self.lib.hvla_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def num_generated(self) -> int:
        return int(self.lib.hvla_num_generated(self.h))

    def load_weights(self, params: Dict[str, np.ndarray]):
        keep = []
        descs = (hvla_tensor_desc * len(params))()
        for i, (k, v) in enumerate(params.items()):
            a = np.ascontiguousarray(np.asarray(v), dtype=np.float32)
            keep.append(a)
            descs[i] = hvla_tensor_desc(k.encode(), a.ctypes.data_as(C.POINTER(C.c_float)), a.size)
        self._check(self.lib.hvla_load_weights(self.h, descs, len(params)), "hvla_load_weights")

    def t5_load(self, t, params: Dict[str, np.ndarray], max_tokens: int, max_batch: int):
        keep = []
        descs = (hvla_tensor_desc * len(params))()
        for i, (k, v) in enumerate(params.items()):
            a = np.ascontiguousarray(np.asarray(v), dtype=np.float32)
            keep.append(a)
            descs[i] = hvla_tensor_desc(k.encode(), a.ctypes.data_as(C.POINTER(C.c_float)), a.size)
        cfg = hvla_t5_config(t.vocab, t.d_model, t.d_kv, t.heads, t.d_ff, t.layers, t.buckets, t.max_distance, t.eps,
                             max_tokens, max_batch)
        self._check(self.lib.hvla_t5_load(self.h, C.byref(cfg), descs, len(params)), "hvla_t5_load")

    def preprocess(self, src_ptr, B, H, W, crop, dst_ptr, stream=0, padded_resize=False):
        flags = (1 if crop else 0) | (2 if padded_resize else 0)          # HVLA_PREPROCESS_CROP | HVLA_PREPROCESS_PAD
        self._check(self.lib.hvla_preprocess(self.h, src_ptr, B, H, W, flags, dst_ptr, C.c_void_p(stream)), "hvla_preprocess")

    def t5_encode(self, ids_ptr, mask_ptr, out_ptr, B, T, stream=0):
        self._check(self.lib.hvla_t5_encode(self.h, ids_ptr, mask_ptr, out_ptr, B, T, C.c_void_p(stream)), "hvla_t5_encode")

    def encode_audit(self, images_ptr, B, stream: int = 0):
        """{site: (largest finite |16-bit operand|, number of inf / NaN)} over all encoder layers (tests)."""
        mx, bad = (C.c_float * 4)(), (C.c_int32 * 4)()
        self._check(self.lib.hvla_encode_audit(self.h, C.c_void_p(images_ptr), B, mx, bad, C.c_void_p(stream)), "hvla_encode_audit")
        return {k: (float(mx[i]), int(bad[i])) for i, k in enumerate(("layernorm_out", "qkv", "attention_out", "gelu_out"))}

    def set_attention_outputs(self, dino_ptr: int = 0, head_ptr: int = 0):
        """Device buffers the following encode / policy / step calls write the two attention maps into (0 = off)."""
        self._check(self.lib.hvla_set_attention_outputs(self.h, C.c_void_p(dino_ptr or None), C.c_void_p(head_ptr or None)),
                    "hvla_set_attention_outputs")

    def train_profile(self, on: bool):
        self._check(self.lib.hvla_train_profile(self.h, int(bool(on))), "hvla_train_profile")

    def train_profile_read(self):
        """(ms summed over the fine-tune step's batched GEMM launches, their f32-equivalent FLOPs, launches) since the last read."""
        ms, fl, n = C.c_float(), C.c_double(), C.c_int32()
        self._check(self.lib.hvla_train_profile_read(self.h, C.byref(ms), C.byref(fl), C.byref(n)), "hvla_train_profile_read")
        return ms.value, fl.value, n.value

    def release_pooled_arenas(self):
        """Give the weight arenas parked by `weights_free` back to the device (up to 4 per context stay resident otherwise)."""
        self._check(self.lib.hvla_release_pooled_arenas(self.h), "hvla_release_pooled_arenas")

    def selftest(self, stream: int = 0):
        self._check(self.lib.hvla_selftest(self.h, C.c_void_p(stream)), "hvla_selftest")

    def profile(self, mode: int):
        self._check(self.lib.hvla_profile(self.h, mode), "hvla_profile")

    def profile_select(self, names):
        """Categories (PROF_NAMES) that mode 1 times."""
        mask = 0
        for n in names:
            mask |= 1 << PROF_NAMES.index(n)
        self._check(self.lib.hvla_profile_select(self.h, mask), "hvla_profile_select")

    def profile_read(self):
        """{category: (total ms, launches)} since the last read."""
        ms = (C.c_float * len(PROF_NAMES))()
        n = (C.c_int32 * len(PROF_NAMES))()
        self._check(self.lib.hvla_profile_read(self.h, ms, n), "hvla_profile_read")
        return {k: (float(ms[i]), int(n[i])) for i, k in enumerate(PROF_NAMES)}

    def launches(self) -> int:
        """Kernel launches (and memset nodes) enqueued by this ctx since the last call."""
        return int(self.lib.hvla_launches(self.h))

    def box_probe(self, stream=0):
        """(sustained shader clock MHz under a chip-wide MFMA loop, that loop's TFLOP/s, its duration in ms)."""
        out = (C.c_float * 3)()
        self._check(self.lib.hvla_box_probe(self.h, out, C.c_void_p(stream)), "hvla_box_probe")
        return float(out[0]), float(out[1]), float(out[2])

    # raw pointer-level calls (device pointers as ints)
    def generate(self, tok_ptr, mask_ptr, cls_ptr, B, stream=0):
        w = C.c_void_p()
        self._check(self.lib.hvla_generate(self.h, tok_ptr, mask_ptr, cls_ptr, B, C.byref(w), C.c_void_p(stream)),
                    "hvla_generate")
        return w

    def weights_free(self, w):
        self.lib.hvla_weights_free(self.h, w)

    def weights_export(self, w, theta_ptr, ctx_ptr, stream=0):
        self._check(self.lib.hvla_weights_export(self.h, w, theta_ptr, ctx_ptr, C.c_void_p(stream)), "hvla_weights_export")

    def encode(self, img_ptr, tok_ptr, B, stream=0):
        self._check(self.lib.hvla_encode(self.h, img_ptr, tok_ptr, B, C.c_void_p(stream)), "hvla_encode")

    def encode_hidden(self, img_ptr, hidden_ptr, B, stream=0):
        self._check(self.lib.hvla_encode_hidden(self.h, img_ptr, hidden_ptr, B, C.c_void_p(stream)), "hvla_encode_hidden")

    def policy(self, w, tok_ptr, act_ptr, logit_ptr, B, stream=0):
        self._check(self.lib.hvla_policy(self.h, w, tok_ptr, act_ptr, logit_ptr, B, C.c_void_p(stream)), "hvla_policy")

    def step(self, w, img_ptr, act_ptr, logit_ptr, B, stream=0):
        self._check(self.lib.hvla_step(self.h, w, img_ptr, act_ptr, logit_ptr, B, C.c_void_p(stream)), "hvla_step")

    def loss(self, act_ptr, logit_ptr, target_ptr, tmask_ptr, amask_ptr, loss_ptr, B, stream=0):
        self._check(self.lib.hvla_loss(self.h, act_ptr, logit_ptr, target_ptr, tmask_ptr, amask_ptr, loss_ptr, B,
                                       C.c_void_p(stream)), "hvla_loss")

    def train_sizes(self, B, train_encoder=False):
        """(n_params, G, workspace_floats, n_hypernet)"""
        out = (C.c_int64 * 4)()
        self._check(self.lib.hvla_train_sizes(self.h, B, int(train_encoder), out), "hvla_train_sizes")
        return int(out[0]), int(out[1]), int(out[2]), int(out[3])

    def train_step(self, buf, ptrs, B, hyper, stream=0):
        self._check(self.lib.hvla_train_step(self.h, C.byref(buf), *ptrs, B, C.byref(hyper), C.c_void_p(stream)),
                    "hvla_train_step")

    def train_apply(self, buf, hyper, stream=0):
        self._check(self.lib.hvla_train_apply(self.h, C.byref(buf), C.byref(hyper), C.c_void_p(stream)), "hvla_train_apply")

    def train_bucket_ranges(self, train_encoder):
        out = (C.c_int64 * 6)()
        self._check(self.lib.hvla_train_bucket_ranges(self.h, int(train_encoder), out), "hvla_train_bucket_ranges")
        return [(int(out[2 * i]), int(out[2 * i + 1])) for i in range(3)]

    def train_wait_bucket(self, bucket, stream):
        self._check(self.lib.hvla_train_wait_bucket(self.h, int(bucket), C.c_void_p(stream)), "hvla_train_wait_bucket")

    def train_accumulate(self, buf, acc_ptr, inv_k, hyper, stream=0):
        self._check(self.lib.hvla_train_accumulate(self.h, C.byref(buf), C.c_void_p(acc_ptr), C.c_float(inv_k), C.byref(hyper),
                                                   C.c_void_p(stream)), "hvla_train_accumulate")

    def ensemble_reset(self, w, stream=0):
        self._check(self.lib.hvla_ensemble_reset(self.h, w, C.c_void_p(stream)), "hvla_ensemble_reset")

    def ensemble(self, w, act_ptr, mean_ptr, std_ptr, mask_ptr, out_ptr, stream=0):
        self._check(self.lib.hvla_ensemble(self.h, w, act_ptr, mean_ptr, std_ptr, mask_ptr, out_ptr, C.c_void_p(stream)),
                    "hvla_ensemble")
