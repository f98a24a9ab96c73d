// Diagnostic (GPU box): do scratch (private-segment) loads and global loads of one wave retire in issue order?
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/vmcnt_order_probe.hip -o /tmp/vmcnt_order_probe && /tmp/vmcnt_order_probe
//
// hipcc's s_waitcnt insertion treats every vector-memory load of a wave (global_*, scratch_*, buffer_*) as returning in
// issue order and waits with COUNTED vmcnt(N).  The old policy megakernel (60 spilled VGPRs: scratch reloads interleaved
// with the global loads of weight fragments that fed MFMAs directly) was run-to-run nondeterministic at -O3 unless its
// memory-sourced MFMA operands went through a VALU copy (tools/policy_hazard_probe.hip).  This probe asks the hardware
// directly: a SLOW global load (a line no cache holds) is followed by a FAST scratch load (L1-hot), then `s_waitcnt vmcnt(1)`
// -- which in-order retirement makes "the global load has landed" -- and the global load's destination is read at once.
// A stale value there means the scratch load retired first and the counted wait let the wave through.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

__global__ void order_kernel(const unsigned* __restrict__ big, size_t stride_words, int iters, unsigned* __restrict__ out,
                             int mode) {
  volatile unsigned priv[8];                       // forces a private segment (scratch) for this kernel
  const int lane = threadIdx.x & 63;
  for (int i = 0; i < 8; ++i) priv[i] = 0x5c5c0000u + i;
  const size_t wave = (size_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  unsigned stale = 0, ok = 0;
  for (int it = 0; it < iters; ++it) {
    const unsigned* src = big + ((wave * iters + it) * stride_words) + lane;   // a cold line per wave and iteration
    unsigned early, late, tmp;
    if (mode == 0) {          // global (slow) then scratch (fast), wait for all but the youngest
      asm volatile(
          "v_mov_b32 %0, 0xdead\n\t"
          "s_nop 4\n\t"
          "global_load_dword %0, %3, off\n\t"
          "scratch_load_dword %2, off, off offset:0\n\t"
          "s_waitcnt vmcnt(1)\n\t"
          "v_mov_b32 %1, %0\n\t"
          "s_waitcnt vmcnt(0)\n\t"
          "s_nop 4"
          : "=&v"(late), "=&v"(early), "=&v"(tmp) : "v"(src) : "memory");
    } else {                  // control: two global loads (slow first, then an L1-hot one)
      const unsigned* hot = big + lane;
      asm volatile(
          "v_mov_b32 %0, 0xdead\n\t"
          "s_nop 4\n\t"
          "global_load_dword %0, %3, off\n\t"
          "global_load_dword %2, %4, off\n\t"
          "s_waitcnt vmcnt(1)\n\t"
          "v_mov_b32 %1, %0\n\t"
          "s_waitcnt vmcnt(0)\n\t"
          "s_nop 4"
          : "=&v"(late), "=&v"(early), "=&v"(tmp) : "v"(src), "v"(hot) : "memory");
    }
    if (early != late) ++stale; else ++ok;
    if (tmp == 0x12345678u) ++ok;                  // keep tmp alive
  }
  out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = stale;
  if (priv[3] == 1u) out[0] = ok;                  // keep the private array
}

int main() {
  const int blocks = 256, threads = 576, iters = 64;
  const size_t stride_words = 4096;                // 16 KiB apart: every access its own cold line
  const size_t waves = (size_t)blocks * (threads / 64);
  const size_t words = waves * iters * stride_words + 64;
  unsigned *big, *out;
  if (hipMalloc(&big, words * 4) != hipSuccess) { printf("alloc of %.1f GB failed\n", words * 4 / 1e9); return 1; }
  (void)hipMemset(big, 0x11, words * 4);
  (void)hipMalloc(&out, (size_t)blocks * threads * 4);
  for (int mode = 0; mode < 2; ++mode) {
    for (int rep = 0; rep < 3; ++rep) {
      (void)hipMemset(big, 0x11 + rep + 3 * mode, words * 4);      // rewritten: nothing of it is in a cache the kernel reads through
      hipLaunchKernelGGL(order_kernel, dim3(blocks), dim3(threads), 0, 0, big, stride_words, iters, out, mode);
      std::vector<unsigned> h((size_t)blocks * threads);
      (void)hipMemcpy(h.data(), out, h.size() * 4, hipMemcpyDeviceToHost);
      unsigned long long stale = 0;
      for (unsigned v : h) stale += v;
      printf("%s: stale reads behind `s_waitcnt vmcnt(1)`: %llu of %llu\n",
             mode == 0 ? "global load, then scratch load" : "global load, then global load ", stale,
             (unsigned long long)h.size() * iters);
    }
  }
  return 0;
}
