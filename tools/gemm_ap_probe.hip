// Diagnostic (GPU box): can ONE wave per SIMD keep the matrix pipe busy on a 128 x 256 output tile?
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/gemm_ap_probe.hip -o /tmp/gemm_ap_probe && /tmp/gemm_ap_probe
//
// Feasibility probe for hiding the GEMM epilogues (known gap 1): two 4-wave groups per workgroup in
// antiphase, one in its K loop while the other runs the previous tile's epilogue, LDS stages time-shared.  That only pays
// if a single 4-wave group (one wave per SIMD, nobody to hide its LDS reads behind) runs the K loop of a 128 x 256 tile
// at a rate close to what the production kernel's eight waves reach on 256 x 256 (0.85 us per 64-deep K-tile = 85 % of
// the MFMA rate), although it moves 1.5 x the bytes per flop from L2 to LDS.  This file times exactly that K loop --
// C[M, N] = A[M, K] . W[N, K]^T, fp16, f32 accumulate, 16-bit store without epilogue arithmetic -- and checks the result
// against a plain kernel.
//   * 4 waves, wave j owns columns [64 j, +64) of all 128 rows: 128 accumulator VGPRs (the production wave tile)
//   * K-steps of 32 through SIX 24 KB LDS stages (A 128 x 32, W 256 x 32; 64-byte rows, chunk ^= (row >> 2) & 3 applied to
//     the DMA source address and to the ds_read_b128 address): the DMA of step s + 5 is issued in step s, four steps
//     (2048 MFMA cycles = 0.85 us) before its fragments are read
//   * per step and wave: counted vmcnt, one raw barrier, 6 DMA instructions, 12 fragment reads for the NEXT step into the
//     other fragment register set, 32 MFMAs on this step's set
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

typedef _Float16 half_t;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) _Float16 f16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;

constexpr int TM = 128, TN = 256, BK = 32, NST = 6, STAGE = (TM + TN) * BK * 2;   // 24 576 B

#define BAR() do { __builtin_amdgcn_sched_barrier(0); __builtin_amdgcn_s_barrier(); __builtin_amdgcn_sched_barrier(0); } while (0)

// ABL (timing ablations, wrong results): 1 no DMA and no vmcnt waits, 2 no fragment reads inside the loop, 3 no barriers, 4 no stores
template <int ABL>
__global__ __launch_bounds__(256) void ap_kloop_kernel(const half_t* __restrict__ A, const half_t* __restrict__ W,
                                                       half_t* __restrict__ C, int M, int N, int K, int ntm, int ntn) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
  const int fr = lane & 15, fq = lane >> 4;
  // DMA pieces of 1 KiB = 16 rows x 64 B: lane -> (row = lane >> 2, LDS chunk c = lane & 3), source chunk c ^ ((row >> 2) & 3).
  // A: 8 pieces per step (wave j: pieces j, j + 4), W: 16 pieces (wave j: j, j + 4, j + 8, j + 12).
  const int prow = lane >> 2, pch = (lane & 3) ^ ((prow >> 2) & 3);
  const uint32_t lane_off = ((uint32_t)prow * (uint32_t)K + pch * 8) * 2u;   // bytes
  // fragment reads: row fr of a 16-row tile, k-chunk fq (8 halves = 16 B), swizzled
  const int fsw = (fq ^ ((fr >> 2) & 3)) << 4;
  const int a_off = fr * 64 + fsw, w_off = TM * 64 + (wave * 64 + fr) * 64 + fsw;
  const int KS = K / BK, ntiles = ntm * ntn;
  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int tm = tile / ntn, tn = tile % ntn;          // n fastest: the workgroups of a wave front share A panels
    const half_t* abase = A + (size_t)tm * TM * K;
    const half_t* wbase = W + (size_t)tn * TN * K;
    auto dma = [&](int s) {                              // K-step s into stage s % NST
      const uint32_t st = lds0 + (uint32_t)((s % NST) * STAGE);
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int piece = wave + 4 * u;
        const half_t* sb = abase + (size_t)piece * 16 * K + s * BK;
        const uint32_t dst = st + piece * 1024;
        uint32_t keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(lane_off), "s"(sb), "s"(dst));
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int piece = wave + 4 * u;
        const half_t* sb = wbase + (size_t)piece * 16 * K + s * BK;
        const uint32_t dst = st + TM * 64 + piece * 1024;
        uint32_t keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(lane_off), "s"(sb), "s"(dst));
      }
    };
    f16x8 fa0[8], fw0[4], fa1[8], fw1[4];            // two fragment sets, compile-time names (a runtime index would put them in scratch)
    auto rd = [&](f16x8 (&fa)[8], f16x8 (&fw)[4], int s) {
      const char* lb = smem + (s % NST) * STAGE;
#pragma unroll
      for (int t = 0; t < 4; ++t) fw[t] = *reinterpret_cast<const f16x8*>(lb + w_off + t * 1024);
#pragma unroll
      for (int t = 0; t < 8; ++t) fa[t] = *reinterpret_cast<const f16x8*>(lb + a_off + t * 1024);
    };
    f32x4 acc[4][8];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    auto mm = [&](const f16x8 (&fa)[8], const f16x8 (&fw)[4]) {
#pragma unroll
      for (int mt = 0; mt < 8; ++mt)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) acc[nt][mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa[mt], fw[nt], acc[nt][mt], 0, 0, 0);
    };
    // one K-step: compute on (fa, fw), read step s + 1 into (na, nw)
    auto step = [&](int s, const f16x8 (&fa)[8], const f16x8 (&fw)[4], f16x8 (&na)[8], f16x8 (&nw)[4]) {
      // step s + 1 must have landed for everyone before its fragments are read below; steps s + 2 .. s + 4 stay in flight
      if constexpr (ABL != 1) {
        if (s + 4 < KS) asm volatile("s_waitcnt vmcnt(18)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      if constexpr (ABL != 3) BAR();
      if constexpr (ABL != 1) if (s + NST - 1 < KS) dma(s + NST - 1);            // into the stage of step s - 1, which nobody reads any more
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (ABL != 2) if (s + 1 < KS) rd(na, nw, s + 1);
      __builtin_amdgcn_sched_barrier(0);
      mm(fa, fw);
      __builtin_amdgcn_sched_barrier(0);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    };
    // prologue: steps 0 .. 4 in flight, fragments of step 0 read
    BAR();                                               // everyone is done with the previous tile's stages
#pragma unroll
    for (int s = 0; s < NST - 1; ++s)
      if (s < KS) dma(s);
    asm volatile("s_waitcnt vmcnt(24)" ::: "memory");    // step 0 landed (KS >= 5 assumed: 4 younger steps x 6 instructions)
    BAR();
    rd(fa0, fw0, 0);
    if constexpr (ABL == 2) rd(fa1, fw1, 1);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    for (int s = 0; s < KS; s += 2) {                    // KS even
      step(s, fa0, fw0, fa1, fw1);
      step(s + 1, fa1, fw1, fa0, fw0);
    }
    // no epilogue arithmetic: round and store (lane (fr, fq) holds rows 4 fq + r of m-tile mt, column fr of n-tile nt)
    half_t* cb = C + (size_t)tm * TM * N + tn * TN + wave * 64;
    if constexpr (ABL == 4) { if (acc[0][0][0] != 12345.678f) continue; }
#pragma unroll
    for (int mt = 0; mt < 8; ++mt)
#pragma unroll
      for (int nt = 0; nt < 4; ++nt)
#pragma unroll
        for (int r = 0; r < 4; ++r) cb[(size_t)(mt * 16 + 4 * fq + r) * N + nt * 16 + fr] = (half_t)acc[nt][mt][r];
  }
}

__global__ void naive_kernel(const half_t* A, const half_t* W, float* C, int M, int N, int K, int rows) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x, m = blockIdx.y;
  if (n >= N || m >= rows) return;
  float s = 0.f;
  for (int k = 0; k < K; ++k) s += (float)A[(size_t)m * K + k] * (float)W[(size_t)n * K + k];
  C[(size_t)m * N + n] = s;
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

int main() {
  const int M = 65536;
  struct Shape { const char* name; int N, K; } shapes[] = {{"qkv", 2304, 768}, {"fc1", 3072, 768}, {"out", 768, 768}, {"fc2", 768, 3072}};
  half_t *A, *W, *C;
  float* R;
  CK(hipMalloc(&A, (size_t)M * 3072 * 2)); CK(hipMalloc(&W, (size_t)3072 * 3072 * 2)); CK(hipMalloc(&C, (size_t)M * 3072 * 2));
  CK(hipMalloc(&R, (size_t)256 * 3072 * 4));
  std::vector<half_t> h((size_t)M * 3072);
  srand(1);
  for (auto& v : h) v = (half_t)((rand() % 2001 - 1000) * 1e-3f);
  CK(hipMemcpy(A, h.data(), h.size() * 2, hipMemcpyHostToDevice));
  CK(hipMemcpy(W, h.data() + 12345, (size_t)3072 * 3072 * 2, hipMemcpyHostToDevice));
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(ap_kloop_kernel<0>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(ap_kloop_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(ap_kloop_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(ap_kloop_kernel<3>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(ap_kloop_kernel<4>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  int ncu = 256;
  (void)hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, 0);
  for (const Shape& sh : shapes) {
    const int ntm = M / TM, ntn = sh.N / TN;
    auto launch = [&] { hipLaunchKernelGGL(ap_kloop_kernel<0>, dim3(ncu), dim3(256), NST * STAGE, 0, A, W, C, M, sh.N, sh.K, ntm, ntn); };
    launch();
    CK(hipDeviceSynchronize());
    // check the first 256 rows and the last 128
    double worst = 0;
    for (int part = 0; part < 2; ++part) {
      const int r0 = part ? M - 128 : 0, rows = part ? 128 : 256;
      hipLaunchKernelGGL(naive_kernel, dim3((sh.N + 255) / 256, rows), dim3(256), 0, 0, A + (size_t)r0 * sh.K, W, R, M, sh.N, sh.K, rows);
      std::vector<float> ref((size_t)rows * sh.N);
      std::vector<half_t> got((size_t)rows * sh.N);
      CK(hipMemcpy(ref.data(), R, ref.size() * 4, hipMemcpyDeviceToHost));
      CK(hipMemcpy(got.data(), C + (size_t)r0 * sh.N, got.size() * 2, hipMemcpyDeviceToHost));
      for (size_t i = 0; i < ref.size(); ++i) {
        const double d = fabs((double)(float)got[i] - ref[i]) / (1.0 + fabs(ref[i]));
        if (d > worst) worst = d;
      }
    }
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0, 0));
    const int iters = 20;
    for (int i = 0; i < iters; ++i) launch();
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    ms /= iters;
    const double tf = 2.0 * M * sh.N * sh.K / (ms * 1e-3) / 1e12;
    printf("%-4s M=%d N=%d K=%d: %8.1f us  %7.1f TFLOP/s (%.0f %% of 2.5 PF)   worst relative error %.2e\n", sh.name, M, sh.N, sh.K,
           ms * 1e3, tf, tf / 25.0, worst);
    auto time_abl = [&](auto fn, const char* what) {
      fn();
      (void)hipEventRecord(e0, 0);
      for (int i = 0; i < 10; ++i) fn();
      (void)hipEventRecord(e1, 0);
      (void)hipEventSynchronize(e1);
      float t = 0;
      (void)hipEventElapsedTime(&t, e0, e1);
      printf("      %-28s %8.1f us\n", what, t * 100.f);
    };
    time_abl([&] { hipLaunchKernelGGL(ap_kloop_kernel<1>, dim3(ncu), dim3(256), NST * STAGE, 0, A, W, C, M, sh.N, sh.K, ntm, ntn); }, "no DMA / vmcnt waits");
    time_abl([&] { hipLaunchKernelGGL(ap_kloop_kernel<2>, dim3(ncu), dim3(256), NST * STAGE, 0, A, W, C, M, sh.N, sh.K, ntm, ntn); }, "no fragment reads");
    time_abl([&] { hipLaunchKernelGGL(ap_kloop_kernel<3>, dim3(ncu), dim3(256), NST * STAGE, 0, A, W, C, M, sh.N, sh.K, ntm, ntn); }, "no barriers");
    time_abl([&] { hipLaunchKernelGGL(ap_kloop_kernel<4>, dim3(ncu), dim3(256), NST * STAGE, 0, A, W, C, M, sh.N, sh.K, ntm, ntn); }, "no stores");
  }
  return 0;
}
