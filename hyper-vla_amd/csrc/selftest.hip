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

// ---- the box probe (hvla_box_probe; bench.py's `box` block): what THIS device sustains, measured in a kernel of its own right
// before a timed region, so that a step time can be read against the box it ran on (the boxes of a pool differ by several per
// cent: clocks under load, memory).  One workgroup of eight waves per CU, every wave `iters` rounds of four independent
// v_mfma_f32_32x32x16_f16 (two waves per SIMD keep the matrix pipe full: the encoder GEMM's occupancy); wave 0 of workgroup 0
// reads the shader clock (s_memtime) and the constant 100 MHz clock (s_memrealtime) around its loop: their ratio is the
// sustained shader clock under matrix load (MI355X_MICROARCH.md, DVFS).  out[0..1] = ticks of the two clocks.
__global__ __launch_bounds__(512) void box_mfma_probe_kernel(float* __restrict__ sink, unsigned long long* __restrict__ out, int iters) {
  const int lane = threadIdx.x & 63;
  f16x8 a, b;
#pragma unroll
  for (int j = 0; j < 8; ++j) a[j] = (_Float16)(0.001f * (float)((lane + j) & 7)), b[j] = (_Float16)(0.002f * (float)((lane * 3 + j) & 7));
  f32x16 c[4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) c[i][r] = 0.f;
  const unsigned long long t0 = __builtin_readcyclecounter(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 4; ++i) c[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c[i], 0, 0, 0);
  }
  const unsigned long long t1 = __builtin_readcyclecounter(), r1 = __builtin_amdgcn_s_memrealtime();
  float acc = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc += c[i][r];
  sink[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = acc;
  if (blockIdx.x == 0 && threadIdx.x == 0) out[0] = t1 - t0, out[1] = r1 - r0;
}
// host: out[0] sustained shader clock in MHz, out[1] the probe's TFLOP/s (dense fp16 MFMA, whole chip), out[2] its duration in ms
hipError_t run_box_probe(float* sink, unsigned long long* ticks, float out[3], hipStream_t st) {
  int dev = 0, ncu = 256;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return e;
  (void)hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev);
  const int iters = 20000;                              // ~ 4 ms: long enough for the clocks to settle under load
  hipEvent_t e0, e1;
  if ((e = hipEventCreate(&e0)) != hipSuccess) return e;
  if ((e = hipEventCreate(&e1)) != hipSuccess) { (void)hipEventDestroy(e0); return e; }
  hipLaunchKernelGGL(box_mfma_probe_kernel, dim3(ncu), dim3(512), 0, st, sink, ticks, 2000);     // warm-up
  (void)hipEventRecord(e0, st);
  hipLaunchKernelGGL(box_mfma_probe_kernel, dim3(ncu), dim3(512), 0, st, sink, ticks, iters);
  (void)hipEventRecord(e1, st);
  e = hipEventSynchronize(e1);
  float ms = 0.f;
  if (e == hipSuccess) e = hipEventElapsedTime(&ms, e0, e1);
  unsigned long long h[2] = {0, 0};
  if (e == hipSuccess) e = hipMemcpy(h, ticks, sizeof h, hipMemcpyDeviceToHost);
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  if (e != hipSuccess) return e;
  out[0] = h[1] ? (float)((double)h[0] / (double)h[1] * 100.0) : 0.f;
  out[1] = ms > 0.f ? (float)(2.0 * 32 * 32 * 16 * 4.0 * iters * 8.0 * ncu / (ms * 1e-3) / 1e12) : 0.f;
  out[2] = ms;
  return hipGetLastError();
}

hipError_t launch_selftest(int* flags, hipStream_t st) {
  hipLaunchKernelGGL(selftest_kernel, dim3(1), dim3(64), 0, st, flags);
  return hipGetLastError();
}

}  // namespace hvla
