// Diagnostic (GPU box): when can ANOTHER wave read the bytes of an LDS-DMA (global_load_lds_dwordx4)?
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/lds_dma_visibility_probe.hip -o /tmp/ldsdma_probe && /tmp/ldsdma_probe
//
// policy_kernel (csrc/policy.hip, acquire()) and gemm64_body (csrc/encoder.hip) stage tiles by LDS-DMA and consume them as
//     issuing wave: s_waitcnt vmcnt(N)  ->  every wave: s_barrier  ->  every wave: ds_read of the staged tile, at once.
// Both showed rare run-to-run differences (profiles/r3_policy_race.txt; the three-stage small-row GEMM of round 3) that a
// delay behind the barrier removes and a stricter vmcnt in front of it does not.  This probe is that protocol and nothing
// else: eight waves, wave w moves fragment w (1 KiB) of tile t from a COLD line of a big buffer (every element carries its
// own global index, so a stale byte is recognisable) into a ring of LDS slots DEPTH - 1 tiles ahead, so that the counted
// wait really waits for the transfer (a tile that landed long ago cannot show anything); behind the barrier every wave
// reads all eight fragments after DELAY x `s_nop 15` and counts the words that are not the expected ones, separately for
// the fragment it moved itself and for the other waves' fragments.  Variants: the read straight behind the barrier, behind
// a delay, behind a SECOND barrier, and with vmcnt(0) in front of the barrier.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int DEPTH, int DELAY, int BARS, bool DRAIN, int DWELL, bool NOWAIT = false, int NREAD = 8>
__global__ __launch_bounds__(512) void probe_kernel(const u32x4* __restrict__ big, size_t tile_stride, int tiles,
                                                    unsigned long long* __restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) char smem[];          // [DEPTH][8 KiB]
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
  const u32x4* base = big + (size_t)blockIdx.x * tiles * tile_stride;
  auto dma = [&](int t) {
    if (t >= tiles) return;
    const u32x4* src = base + (size_t)t * tile_stride + wave * 64 + lane;
    const uint32_t dst = __builtin_amdgcn_readfirstlane(lds0 + (uint32_t)((t % DEPTH) * 8192 + wave * 1024));
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(src), "s"(dst) : "memory");
  };
  unsigned long long bad_own = 0, bad_other = 0;
#pragma unroll
  for (int t = 0; t < DEPTH - 1; ++t) dma(t);
  for (int t = 0; t < tiles; ++t) {
    // all but the DEPTH - 2 youngest transfers of this wave are done => its fragment of tile t has landed
    if (NOWAIT) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // negative control: nothing waits for the transfer
    else if (DRAIN || t + DEPTH - 2 >= tiles) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(DEPTH - 2) : "memory");
#pragma unroll
    for (int b = 0; b < BARS; ++b) __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int d = 0; d < DELAY; ++d) asm volatile("s_nop 15" ::: "memory");
    dma(t + DEPTH - 1);                                                 // into the slot of tile t - 1 (every wave is done with it)
    const u32x4* slot = reinterpret_cast<const u32x4*>(smem + (t % DEPTH) * 8192);
    const size_t tile0 = ((size_t)blockIdx.x * tiles + t) * tile_stride;
#pragma unroll
    for (int ff = 0; ff < NREAD; ++ff) {                                // NREAD < 8: this wave's own fragment and the next waves' (a short
      const int f = NREAD == 8 ? ff : (wave + ff) & 7;                  // iteration: the counted wait then blocks on every tile)
      const u32x4 v = slot[f * 64 + lane];
      const unsigned long long idx = tile0 + f * 64 + lane;             // what element (f, lane) of this tile must hold
      const bool ok = v[0] == (unsigned)idx && v[1] == (unsigned)(idx >> 32) && v[2] == ~(unsigned)idx && v[3] == 0x5eed0000u + (unsigned)(idx % 977);
      if (!ok) { if (f == wave) ++bad_own; else ++bad_other; }
    }
    // some work on the tile, so that the next tile's transfer is not simply behind this one in the queue
#pragma unroll 1
    for (int d = 0; d < DWELL; ++d) asm volatile("s_nop 15" ::: "memory");
  }
  atomicAdd(out, bad_own);
  atomicAdd(out + 1, bad_other);
}

__global__ void fill_kernel(u32x4* big, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    big[i] = u32x4{(unsigned)i, (unsigned)(i >> 32), ~(unsigned)i, 0x5eed0000u + (unsigned)(i % 977)};
}

template <int DEPTH, int DELAY, int BARS, bool DRAIN, int DWELL, bool NOWAIT = false, int NREAD = 8>
static void run(const char* what, const u32x4* big, size_t tile_stride, int blocks, int tiles, unsigned long long* d_out, int reps) {
  unsigned long long tot[2] = {0, 0};
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(probe_kernel<DEPTH, DELAY, BARS, DRAIN, DWELL, NOWAIT, NREAD>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  for (int r = 0; r < reps; ++r) {
    (void)hipMemset(d_out, 0, 16);
    hipLaunchKernelGGL((probe_kernel<DEPTH, DELAY, BARS, DRAIN, DWELL, NOWAIT, NREAD>), dim3(blocks), dim3(512), DEPTH * 8192, 0, big, tile_stride, tiles, d_out);
    unsigned long long h[2];
    (void)hipMemcpy(h, d_out, 16, hipMemcpyDeviceToHost);
    tot[0] += h[0], tot[1] += h[1];
  }
  const double words = (double)reps * blocks * tiles * (double)NREAD * 512;     // (reader wave, fragment, lane) checks
  printf("  %-78s stale 16-B words: own fragment %8llu, other waves' fragments %10llu   of %.3g\n", what, tot[0], tot[1], words);
}

int main() {
  const int blocks = 256, tiles = 96, reps = 40;
  const size_t tile_stride = 8192;                                      // 128 KiB between tiles: every tile comes from cold lines
  const size_t n = (size_t)blocks * tiles * tile_stride;
  u32x4* big = nullptr;
  unsigned long long* d_out = nullptr;
  if (hipMalloc(&big, n * sizeof(u32x4)) != hipSuccess || hipMalloc(&d_out, 16) != hipSuccess) { printf("alloc failed\n"); return 1; }
  hipLaunchKernelGGL(fill_kernel, dim3(2048), dim3(256), 0, 0, big, n);
  (void)hipDeviceSynchronize();
  printf("LDS-DMA visibility probe: %d workgroups x 8 waves x %d tiles x %d launches; tile t + DEPTH - 1 is requested when tile t is read\n", blocks, tiles, reps);
  printf("one tile ahead (DEPTH 2: the counted wait really waits for the transfer)\n");
  run<2, 0, 1, false, 0>("vmcnt(N) -> s_barrier -> ds_read at once", big, tile_stride, blocks, tiles, d_out, reps);
  run<2, 0, 1, true, 0>("vmcnt(0) -> s_barrier -> ds_read at once", big, tile_stride, blocks, tiles, d_out, reps);
  run<2, 0, 2, false, 0>("vmcnt(N) -> s_barrier, s_barrier -> ds_read", big, tile_stride, blocks, tiles, d_out, reps);
  run<2, 1, 1, false, 0>("vmcnt(N) -> s_barrier -> 1 x s_nop 15 -> ds_read", big, tile_stride, blocks, tiles, d_out, reps);
  run<2, 4, 1, false, 0>("vmcnt(N) -> s_barrier -> 4 x s_nop 15 -> ds_read", big, tile_stride, blocks, tiles, d_out, reps);
  run<2, 16, 1, false, 0>("vmcnt(N) -> s_barrier -> 16 x s_nop 15 -> ds_read", big, tile_stride, blocks, tiles, d_out, reps);
  run<2, 0, 1, false, 0, true>("NEGATIVE CONTROL: no vmcnt wait at all -> s_barrier -> ds_read at once", big, tile_stride, blocks, tiles, d_out, reps);
  printf("one tile ahead, two fragments read per wave and tile (own + the next wave's): an iteration is shorter than a transfer, the wait blocks every time\n");
  run<2, 0, 1, false, 0, true, 2>("NEGATIVE CONTROL: no vmcnt wait at all -> s_barrier -> ds_read at once", big, tile_stride, blocks, tiles, d_out, reps);
  run<2, 0, 1, false, 0, false, 2>("vmcnt(N) -> s_barrier -> ds_read at once", big, tile_stride, blocks, tiles, d_out, reps);
  run<2, 0, 1, true, 0, false, 2>("vmcnt(0) -> s_barrier -> ds_read at once", big, tile_stride, blocks, tiles, d_out, reps);
  run<2, 1, 1, false, 0, false, 2>("vmcnt(N) -> s_barrier -> 1 x s_nop 15 -> ds_read", big, tile_stride, blocks, tiles, d_out, reps);
  run<3, 0, 1, false, 0, false, 2>("two ahead: vmcnt(1) -> s_barrier -> ds_read at once", big, tile_stride, blocks, tiles, d_out, reps);
  run<4, 0, 1, false, 0, false, 2>("three ahead: vmcnt(2) -> s_barrier -> ds_read at once", big, tile_stride, blocks, tiles, d_out, reps);
  printf("two tiles ahead (DEPTH 3), 32 x s_nop 15 of work per tile\n");
  run<3, 0, 1, false, 32>("vmcnt(N) -> s_barrier -> ds_read at once", big, tile_stride, blocks, tiles, d_out, reps);
  run<3, 4, 1, false, 32>("vmcnt(N) -> s_barrier -> 4 x s_nop 15 -> ds_read", big, tile_stride, blocks, tiles, d_out, reps);
  printf("three tiles ahead (DEPTH 4: policy_kernel's ring), 32 x s_nop 15 of work per tile\n");
  run<4, 0, 1, false, 32>("vmcnt(N) -> s_barrier -> ds_read at once", big, tile_stride, blocks, tiles, d_out, reps);
  run<4, 4, 1, false, 32>("vmcnt(N) -> s_barrier -> 4 x s_nop 15 -> ds_read", big, tile_stride, blocks, tiles, d_out, reps);
  (void)hipFree(big);
  (void)hipFree(d_out);
  return 0;
}
