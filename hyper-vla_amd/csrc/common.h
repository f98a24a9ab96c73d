// common.h — shared device helpers for libhvla (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace hvla {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) _Float16 f16x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2;

constexpr int WAVE = 64;

// ---- 16-bit operand families for the encoder (fp16 default, bf16 selectable) -------------------
struct OpF16 {
  using elem = _Float16;
  using x8 = f16x8;
  using x4 = f16x4;
  static __device__ __forceinline__ f32x4 mma16(x8 a, x8 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
  }
  static __device__ __forceinline__ f32x16 mma32(x8 a, x8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
  }
};
struct OpBF16 {
  using elem = __bf16;
  using x8 = bf16x8;
  using x4 = bf16x4;
  static __device__ __forceinline__ f32x4 mma16(x8 a, x8 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
  }
  static __device__ __forceinline__ f32x16 mma32(x8 a, x8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
  }
};

// ---- split-bf16 ("bf16x3") arithmetic: x = hi + lo with hi = bf16(x), lo = bf16(x - hi) ---------
// a*b ~= a_hi*b_hi + a_hi*b_lo + a_lo*b_hi, f32 accumulate: ~2^-16 relative instead of 2^-9.
struct Split8 {
  bf16x8 hi, lo;
};
__device__ __forceinline__ void split1(float x, __bf16& h, __bf16& l) {
  h = (__bf16)x;
  l = (__bf16)(x - (float)h);
}
__device__ __forceinline__ Split8 split8(const float* v) {
  Split8 s;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    __bf16 h, l;
    split1(v[j], h, l);
    s.hi[j] = h;
    s.lo[j] = l;
  }
  return s;
}
// 32x32x16: D[32x32] += A[32x16] * B[16x32]
__device__ __forceinline__ f32x16 mma32_x3(const Split8& a, const Split8& b, f32x16 c) {
  c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.lo, b.hi, c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.hi, b.lo, c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.hi, b.hi, c, 0, 0, 0);
  return c;
}

// C-layout of v_mfma_f32_32x32x16_*: lane l holds column (l & 31); register r holds row
//   crow(r, l >> 5) = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5)              (guide §3)
__device__ __forceinline__ constexpr int crow(int r, int half) {
  return (r & 3) + 8 * (r >> 2) + 4 * half;
}

__device__ __forceinline__ float wave_xor_sum32(float v) {
  // combine the two 32-lane halves (lanes l and l^32)
  return v + __shfl_xor(v, 32, 64);
}

__device__ __forceinline__ float gelu_tanh(float x) {
  // flax nn.gelu(approximate=True): 0.5 x (1 + tanh(u)),  u = sqrt(2/pi) (x + 0.044715 x^3).
  // 0.5 (1 + tanh u) = 1 / (1 + exp(-2 u)): one v_exp_f32 and one v_rcp_f32 instead of libm's tanhf (about 25 instructions,
  // and this sits on every hidden value of the generated policy, whose kernel is VALU-bound).  |error| ~ 1e-7 relative:
  // exp2 overflows to +inf for u < -44 (result -0.0 x ... = 0) and underflows to 0 for large u (result x), both the limits.
  const float k0 = 0.7978845608028654f, k1 = 0.044715f;
  const float u = k0 * (x + k1 * x * x * x);
  return x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-2.0f * 1.4426950408889634f * u));
}
// exact-erf GELU (HF ACT2FN["gelu"]): the fc1 epilogue applies it to B*257*3072 values per layer with the matrix pipe idle, so its
// instruction count is step time (measured: an fc1 tile is 3.4 us longer than a QKV tile of the same K).  Even form, no sign
// transfer and no cancellation for negative x:
//   gelu(x) = max(x, 0) - |x| Phi(-|x|),   Phi(-a) = erfc(a / sqrt 2) / 2 = exp2(q(a)),
// q = a minimax fit of log2 Phi(-a) on [0, 8] weighted by a Phi(-a) (tools/fit_gelu.py) -- degree 5 since round 5: |error of the product|
// <= 4.7e-7 evaluated in f32, against an output that is rounded to 16 bits (half an ulp of a value of 0.1 is 3e-5; degree 6, rounds
// 4-5: 8.7e-8; Abramowitz-Stegun 7.1.26, rounds 1-3: 2.1e-7 and a reciprocal as well, two transcendentals per element instead of
// one).  One packed fma per element pair less is 0.05 ms of the step (the epilogue runs with the matrix pipe idle: every vector
// instruction per element costs 0.07 ms); the 64-episode fixtures moved from 8.5e-4 / 7.2e-4 / 4.2e-5 to 8.0e-4 / 7.2e-4 / 4.4e-5.
// Beyond 8 the exponent is held (the polynomial is not monotone out there): the product is then below 2^-50 |x|.  Five packed fmas +
// one for the result, one v_exp_f32, min / max per element.
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 gelu_erf2(f32x2 x) {
  const f32x2 ax = __builtin_elementwise_abs(x);
  const f32x2 a = __builtin_elementwise_min(ax, f32x2{8.f, 8.f});
  f32x2 q = __builtin_elementwise_fma(a, f32x2{-0.000473309017252177f, -0.000473309017252177f}, f32x2{0.007084553595632315f, 0.007084553595632315f});
  q = __builtin_elementwise_fma(q, a, f32x2{-0.05182736739516258f, -0.05182736739516258f});
  q = __builtin_elementwise_fma(q, a, f32x2{-0.4599924683570862f, -0.4599924683570862f});
  q = __builtin_elementwise_fma(q, a, f32x2{-1.150787830352783f, -1.150787830352783f});
  q = __builtin_elementwise_fma(q, a, f32x2{-1.000037670135498f, -1.000037670135498f});
  const f32x2 e = {__builtin_amdgcn_exp2f(q[0]), __builtin_amdgcn_exp2f(q[1])};
  const f32x2 m = __builtin_elementwise_max(x, f32x2{0.f, 0.f});
  // -min(|x|, 8) 2^q instead of -|x| 2^q: beyond |x| = 8 the product is below 2^-47 either way -- nothing next to x for x > 8, and zero in
  // every 16-bit output format for x < -8 (every caller rounds to one) -- and |x| is then only an operand MODIFIER of the v_min_f32
  // above: the packed fma cannot take one, and the v_and_b32 per element it needed was a tenth of the GELU epilogue's vector instructions.
  return __builtin_elementwise_fma(-a, e, m);
}
__device__ __forceinline__ float gelu_erf(float x) { return gelu_erf2(f32x2{x, x})[0]; }

}  // namespace hvla
