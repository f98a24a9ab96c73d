// plan.h -- the host-side decisions of run_encoder that are pure arithmetic (no HIP types), kept apart so that a CPU test can
// drive them at sizes no test batch reaches (tests/native/plan_check.cpp, built with g++ under ASan / UBSan by
// tests/test_host_sanitizers.py), plus the tile-order arithmetic the 64 x 64 and 128 x 128 GEMM kernels share.
#pragma once
#include <stddef.h>
#include <stdint.h>

#ifdef __HIPCC__
#define HVLA_HD __host__ __device__ __forceinline__
#else
#define HVLA_HD inline
#endif

namespace hvla {

// Workgroups are dealt round-robin over the chip's 8 XCDs (block b runs on XCD b % 8), each with an L2 of its own.  xcd_run maps
// the block id to a linear tile index such that XCD x owns the x-th of 8 CONTIGUOUS runs of tile indices (the first nwg % 8 runs one
// longer): a bijection of [0, nwg) for every nwg >= 1.  Tiles that share an operand panel are made neighbours in the linear order,
// so the panel crosses the fabric once per XCD that a run boundary puts it on (at most two) instead of once per L2.
HVLA_HD int xcd_run(int bid, int nwg) {
  const int q = nwg / 8, r = nwg % 8, xcd = bid % 8;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + bid / 8;
}

// A GEMM output of 96 MB or more is not found in L2 / the memory-side cache by the kernel that reads it next: the 16-bit outputs
// of QKV / fc1 are then stored non-temporally and the f32 residual rows read non-temporally (gemm256p_kernel<..., NT>).
HVLA_HD bool big_output(size_t rows, size_t cols, size_t elem_bytes) { return rows * cols * elem_bytes >= ((size_t)96 << 20); }

// The non-temporal 16-bit stores of the QKV / GELU epilogues are buffer stores with ONE 32-bit BYTE offset per lane (the rows as
// scalar offsets): an output of 4 GiB or more cannot be addressed that way and takes the flat-store form (element offsets, good to
// 2^32 ELEMENTS, which run_encoder checks for every batch).  fc1 at S = 257, F = 3072: from B = 2721 on.
HVLA_HD bool nt16_addressable(size_t rows, size_t cols, size_t elem_bytes) { return rows * cols * elem_bytes < ((size_t)1 << 32); }

}  // namespace hvla
