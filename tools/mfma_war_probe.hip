// Diagnostic (GPU box): stand-alone repro of the hazard behind the policy megakernel's -O3 nondeterminism.
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/mfma_war_probe.hip -o /tmp/mfma_war_probe && /tmp/mfma_war_probe
//
// Question: may a vector-memory load overwrite the VGPRs an MFMA reads as its A operand while that MFMA is still QUEUED
// behind MFMAs it depends on (an accumulation chain)?  hipcc (ROCm 7.2) schedules exactly that in policy_kernel at -O3:
//     v_mfma_f32_32x32x16_bf16 v[2:17], v[74:77], v[18:21], v[2:17]
//     v_mfma_f32_32x32x16_bf16 v[2:17], v[74:77], v[26:29], v[2:17]      <- waits for the one above
//     global_load_dwordx4 v[74:77], v[78:79], off                         <- next weight fragment into the SAME registers
// and guards only SrcC against such writes (the documented late read).  Each wave here issues a chain of NDEP dependent
// MFMAs whose A operand is all ones, then at once reloads the A registers with zeros from an L1/L2-resident line, and
// checks whether every MFMA of the chain still saw the ones (expected: acc = NDEP * 16 in every element).
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

template <int NDEP, int GAP>
__global__ void war_kernel(const u32x4* __restrict__ zeros, float* __restrict__ out, int iters) {
  const int lane = threadIdx.x & 63;
  __shared__ u32x4 lz[64];
  if (threadIdx.x < 64) lz[threadIdx.x] = u32x4{0u, 0u, 0u, 0u};
  __syncthreads();
  bf16x8 ones;
#pragma unroll
  for (int j = 0; j < 8; ++j) ones[j] = (__bf16)1.0f;
  const u32x4* src = zeros + lane;            // 1 KiB that stays in L1 / L2: short load latency is the worst case
  float bad = 0.f;
  for (int it = 0; it < iters; ++it) {
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    u32x4 a = __builtin_bit_cast(u32x4, ones);
    const bf16x8 b = ones;
    // NDEP dependent MFMAs reading `a`, then the load that overwrites `a`, all in one statement so that the compiler can
    // neither separate them nor insert its own waits (the leading s_nop: operands written by the compiler's v_mov just
    // before the statement need their VALU -> MFMA wait states, which hipcc does not insert inside asm)
    if constexpr (NDEP == 2 && GAP == 0)
      asm volatile("s_nop 7\n\tv_mfma_f32_32x32x16_bf16 %0, %1, %2, %0\n\tv_mfma_f32_32x32x16_bf16 %0, %1, %2, %0\n\t"
                   "global_load_dwordx4 %1, %3, off\n\ts_waitcnt vmcnt(0)\n\ts_nop 15\n\ts_nop 15"
                   : "+v"(acc), "+v"(a) : "v"(b), "v"(src) : "memory");
    if constexpr (NDEP == 4 && GAP == 0)
      asm volatile("s_nop 7\n\tv_mfma_f32_32x32x16_bf16 %0, %1, %2, %0\n\tv_mfma_f32_32x32x16_bf16 %0, %1, %2, %0\n\t"
                   "v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0\n\tv_mfma_f32_32x32x16_bf16 %0, %1, %2, %0\n\t"
                   "global_load_dwordx4 %1, %3, off\n\ts_waitcnt vmcnt(0)\n\ts_nop 15\n\ts_nop 15"
                   : "+v"(acc), "+v"(a) : "v"(b), "v"(src) : "memory");
    if constexpr (NDEP == 8 && GAP == 0)
      asm volatile("s_nop 7\n\tv_mfma_f32_32x32x16_bf16 %0, %1, %2, %0\n\tv_mfma_f32_32x32x16_bf16 %0, %1, %2, %0\n\t"
                   "v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0\n\tv_mfma_f32_32x32x16_bf16 %0, %1, %2, %0\n\t"
                   "v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0\n\tv_mfma_f32_32x32x16_bf16 %0, %1, %2, %0\n\t"
                   "v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0\n\tv_mfma_f32_32x32x16_bf16 %0, %1, %2, %0\n\t"
                   "global_load_dwordx4 %1, %3, off\n\ts_waitcnt vmcnt(0)\n\ts_nop 15\n\ts_nop 15"
                   : "+v"(acc), "+v"(a) : "v"(b), "v"(src) : "memory");
    if constexpr (NDEP == 8 && GAP == 1)          // the same chain, but the load is held back until the chain has drained
      asm volatile("s_nop 7\n\tv_mfma_f32_32x32x16_bf16 %0, %1, %2, %0\n\tv_mfma_f32_32x32x16_bf16 %0, %1, %2, %0\n\t"
                   "v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0\n\tv_mfma_f32_32x32x16_bf16 %0, %1, %2, %0\n\t"
                   "v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0\n\tv_mfma_f32_32x32x16_bf16 %0, %1, %2, %0\n\t"
                   "v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0\n\tv_mfma_f32_32x32x16_bf16 %0, %1, %2, %0\n\t"
                   "s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\t"
                   "s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\t"
                   "global_load_dwordx4 %1, %3, off\n\ts_waitcnt vmcnt(0)\n\ts_nop 15\n\ts_nop 15"
                   : "+v"(acc), "+v"(a) : "v"(b), "v"(src) : "memory");
    if constexpr (GAP == 2) {                     // the same question for an LDS read (shorter latency than a global load)
      const unsigned laddr = (unsigned)(lane * 16);
      if constexpr (NDEP == 2)
        asm volatile("s_nop 7\n\tv_mfma_f32_32x32x16_bf16 %0, %1, %2, %0\n\tv_mfma_f32_32x32x16_bf16 %0, %1, %2, %0\n\t"
                     "ds_read_b128 %1, %3\n\ts_waitcnt lgkmcnt(0)\n\ts_nop 15\n\ts_nop 15"
                     : "+v"(acc), "+v"(a) : "v"(b), "v"(laddr) : "memory");
      if constexpr (NDEP == 4)
        asm volatile("s_nop 7\n\tv_mfma_f32_32x32x16_bf16 %0, %1, %2, %0\n\tv_mfma_f32_32x32x16_bf16 %0, %1, %2, %0\n\t"
                     "v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0\n\tv_mfma_f32_32x32x16_bf16 %0, %1, %2, %0\n\t"
                     "ds_read_b128 %1, %3\n\ts_waitcnt lgkmcnt(0)\n\ts_nop 15\n\ts_nop 15"
                     : "+v"(acc), "+v"(a) : "v"(b), "v"(laddr) : "memory");
      if constexpr (NDEP == 8)
        asm volatile("s_nop 7\n\tv_mfma_f32_32x32x16_bf16 %0, %1, %2, %0\n\tv_mfma_f32_32x32x16_bf16 %0, %1, %2, %0\n\t"
                     "v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0\n\tv_mfma_f32_32x32x16_bf16 %0, %1, %2, %0\n\t"
                     "v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0\n\tv_mfma_f32_32x32x16_bf16 %0, %1, %2, %0\n\t"
                     "v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0\n\tv_mfma_f32_32x32x16_bf16 %0, %1, %2, %0\n\t"
                     "ds_read_b128 %1, %3\n\ts_waitcnt lgkmcnt(0)\n\ts_nop 15\n\ts_nop 15"
                     : "+v"(acc), "+v"(a) : "v"(b), "v"(laddr) : "memory");
    }
    const float want = (float)(NDEP * 16);
#pragma unroll
    for (int r = 0; r < 16; ++r) bad += acc[r] != want ? 1.f : 0.f;
    if (a[0] != 0u) bad += 1e6f;              // the load itself must have happened
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = bad;
}

template <int NDEP, int GAP>
static void run(const u32x4* zeros, float* out, int blocks, int threads, int iters, const char* what) {
  hipLaunchKernelGGL((war_kernel<NDEP, GAP>), dim3(blocks), dim3(threads), 0, 0, zeros, out, iters);
  std::vector<float> h((size_t)blocks * threads);
  (void)hipMemcpy(h.data(), out, h.size() * 4, hipMemcpyDeviceToHost);
  double bad = 0;
  for (float v : h) bad += v;
  printf("%-46s %d waves/CU: wrong accumulator elements %.0f of %.0f\n", what, threads / 64, bad, (double)h.size() * iters * 16);
  fflush(stdout);
}

int main() {
  u32x4* zeros;
  float* out;
  (void)hipMalloc(&zeros, 64 * sizeof(u32x4));
  (void)hipMemset(zeros, 0, 64 * sizeof(u32x4));
  (void)hipMalloc(&out, (size_t)256 * 1024 * 4);
  for (int threads : {64, 256, 576, 1024}) {
    run<2, 0>(zeros, out, 256, threads, 2000, "2 dependent MFMAs, then load over A");
    run<4, 0>(zeros, out, 256, threads, 2000, "4 dependent MFMAs, then load over A");
    run<8, 0>(zeros, out, 256, threads, 2000, "8 dependent MFMAs, then load over A");
    run<8, 1>(zeros, out, 256, threads, 2000, "8 dependent MFMAs, 256 wait states, then load");
    run<2, 2>(zeros, out, 256, threads, 2000, "2 dependent MFMAs, then ds_read over A");
    run<4, 2>(zeros, out, 256, threads, 2000, "4 dependent MFMAs, then ds_read over A");
    run<8, 2>(zeros, out, 256, threads, 2000, "8 dependent MFMAs, then ds_read over A");
  }
  return 0;
}
