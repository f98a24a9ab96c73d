// selftest.hip — exact-integer probes of the MFMA fragment layouts every kernel in this library relies
// on (guide §3: "always A = I-check with ASYMMETRIC B").  flags[i] counts mismatching lanes of probe i.
#include "common.h"
#include "kernels.h"

namespace hvla {

__device__ __forceinline__ float pa(int i, int k) { return (float)((i * 3 + k * 5) % 7 - 3); }
__device__ __forceinline__ float pb(int k, int j) { return (float)((k * 2 + j * 7) % 9 - 4); }

template <typename Op>
__device__ void probe32(int* flag) {
  const int lane = threadIdx.x & 63, rc = lane & 31, half = lane >> 5;
  typename Op::x8 a, b;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    a[j] = (typename Op::elem)pa(rc, 8 * half + j);      // A[row = l & 31][k = 8 (l >> 5) + j]
    b[j] = (typename Op::elem)pb(8 * half + j, rc);      // B[k = 8 (l >> 5) + j][col = l & 31]
  }
  f32x16 c;
#pragma unroll
  for (int r = 0; r < 16; ++r) c[r] = 0.f;
  c = Op::mma32(a, b, c);
  int bad = 0;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int row = crow(r, half);
    float e = 0.f;
    for (int k = 0; k < 16; ++k) e += pa(row, k) * pb(k, rc);
    bad += (c[r] != e);
  }
  if (bad) atomicAdd(flag, 1);
}

template <typename Op>
__device__ void probe16(int* flag) {
  const int lane = threadIdx.x & 63, rc = lane & 15, q = lane >> 4;
  typename Op::x8 a, b;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    a[j] = (typename Op::elem)pa(rc, 8 * q + j);         // A[row = l & 15][k = 8 (l >> 4) + j]
    b[j] = (typename Op::elem)pb(8 * q + j, rc);         // B[k = 8 (l >> 4) + j][col = l & 15]
  }
  f32x4 c = {0.f, 0.f, 0.f, 0.f};
  c = Op::mma16(a, b, c);
  int bad = 0;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int row = 4 * q + r;
    float e = 0.f;
    for (int k = 0; k < 32; ++k) e += pa(row, k) * pb(k, rc);
    bad += (c[r] != e);
  }
  if (bad) atomicAdd(flag, 1);
}

// accumulator tile -> next product's B operand (policy.hip's whole structure): Y = A2 . X where
// X = A1 . B1 is taken straight from registers; checks the k-permutation kphi of layout.h.
__device__ void probe_chain(int* flag) {
  const int lane = threadIdx.x & 63, rc = lane & 31, half = lane >> 5;
  bf16x8 a, b;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    a[j] = (__bf16)pa(rc, 8 * half + j);
    b[j] = (__bf16)pb(8 * half + j, rc);
  }
  f32x16 x;
#pragma unroll
  for (int r = 0; r < 16; ++r) x[r] = 0.f;
  x = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, x, 0, 0, 0);   // X[32 x 32], |X| <= 16*3*4 = 192 (exact in bf16? no:
  // keep it exact: reduce to small integers first
#pragma unroll
  for (int r = 0; r < 16; ++r) x[r] = (float)(((int)x[r] % 5 + 5) % 5 - 2);
  // Y[m][col] = sum_k A2[m][k] X[k][col], k = 0..31: two k-steps; element j of half h in step s is row
  // 16 s + 8 (j >> 2) + 4 h + (j & 3) of X
  f32x16 y;
#pragma unroll
  for (int r = 0; r < 16; ++r) y[r] = 0.f;
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    bf16x8 xb, a2;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      xb[j] = (__bf16)x[8 * s + j];
      const int k = 16 * s + 8 * (j >> 2) + 4 * half + (j & 3);
      a2[j] = (__bf16)pa(rc + 1, k + 2);
    }
    y = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, xb, y, 0, 0, 0);
  }
  int bad = 0;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int row = crow(r, half);
    float e = 0.f;
    for (int k = 0; k < 32; ++k) {
      float xe = 0.f;
      for (int kk = 0; kk < 16; ++kk) xe += pa(k, kk) * pb(kk, rc);
      xe = (float)(((int)xe % 5 + 5) % 5 - 2);
      e += pa(row + 1, k + 2) * xe;
    }
    bad += (y[r] != e);
  }
  if (bad) atomicAdd(flag, 1);
}

__global__ void selftest_kernel(int* flags) {
  probe32<OpBF16>(flags + 0);
  probe32<OpF16>(flags + 1);
  probe16<OpBF16>(flags + 2);
  probe16<OpF16>(flags + 3);
  probe_chain(flags + 4);
  // split-bf16 sanity: hi + lo reproduces an f32 to ~2^-17 relative
  const float v = 1.2345678f + 0.001f * threadIdx.x;
  __bf16 h, l;
  split1(v, h, l);
  if (fabsf(((float)h + (float)l) - v) > 1.6e-5f * fabsf(v)) atomicAdd(flags + 5, 1);
}

hipError_t launch_selftest(int* flags, hipStream_t st) {
  hipLaunchKernelGGL(selftest_kernel, dim3(1), dim3(64), 0, st, flags);
  return hipGetLastError();
}

}  // namespace hvla
