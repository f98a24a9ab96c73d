// train.hip — fine-tune step of the hypernetwork (BASELINE config 5):
//   forward + backward of  loss(params) = mean_b MixLoss(policy(theta_b(params), tokens_b), action_b)
//   (scripts/train.py:326-346,453-460; hypervla/components/action_heads.py:474-522), fused
//   clip-by-global-norm + AdamW (bf16 first moment) + EMA (octo/utils/train_utils.py:411-426,
//   scripts/train.py:618-625).  The DINOv2 image encoder is FROZEN in this first version
//   (`fine_tune_pretrained_image_encoder=False`, the reference's config default): its tokens come from
//   hvla_encode; gradients flow to every hypernetwork parameter (73 output heads = W_cat / b_cat, the
//   context encoder, the projections and position embeddings).
//
// Correctness-first f32 implementation: one generic strided/batched f32 GEMM kernel (per-episode weights
// are just a batch stride into theta[B, G]) plus small LayerNorm / softmax / GELU / bias-gradient kernels,
// sequenced by host code.  Activations are kept in a workspace; cheap things (LN output, GELU) are
// recomputed in the backward pass.  Parity: every gradient leaf against autograd on the float64 CPU
// restatement (tests/test_gpu_train.py).
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "common.h"
#include "kernels.h"
#include "train.h"

namespace hvla {

// ------------------------------------------------------------------------------------------------
// generic batched GEMM  C[b0,b1] (+)= alpha * op(A)[b0,b1] * op(B)[b0,b1] (+ bias[b0][n])
// ------------------------------------------------------------------------------------------------

// 64x64 output tile per workgroup, 4 waves x (32x32) on the exact-f32 matrix instruction
// v_mfma_f32_32x32x2_f32 (bitwise an fmaf chain, guide §3): lane l supplies A[i = l & 31][k = l >> 5] and
// B[k = l >> 5][j = l & 31], so both operands are conflict-free row reads of the k-major LDS tiles.
template <bool TA, bool TB>
__global__ __launch_bounds__(256) void bgemm_kernel(BG g) {
  __shared__ float As[16][64];
  __shared__ float Bs[16][64];
  const int zb = blockIdx.z / g.ksplit, kc = blockIdx.z % g.ksplit;
  const int b0 = zb / g.nb1, b1 = zb % g.nb1;
  const int kchunk = ((g.K + g.ksplit - 1) / g.ksplit + 15) & ~15;
  const int kbeg = kc * kchunk, kend = kbeg + kchunk < g.K ? kbeg + kchunk : g.K;
  const float* A = g.A + b0 * g.sA0 + b1 * g.sA1;
  const float* B = g.B + b0 * g.sB0 + b1 * g.sB1;
  float* C = g.C + b0 * g.sC0 + b1 * g.sC1;
  const int m0 = blockIdx.y * 64, n0 = blockIdx.x * 64;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int wm = wave >> 1, wn = wave & 1, col = lane & 31, half = lane >> 5;
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  for (int k0 = kbeg; k0 < kend; k0 += 16) {
    for (int i = threadIdx.x; i < 64 * 16; i += 256) {
      int mm, kk;
      if (TA) { mm = i & 63; kk = i >> 6; } else { kk = i & 15; mm = i >> 4; }
      const int m = m0 + mm, k = k0 + kk;
      float v = 0.f;
      if (m < g.M && k < kend) v = TA ? A[(long)k * g.lda + m] : A[(long)m * g.lda + k];
      As[kk][mm] = v;
    }
    for (int i = threadIdx.x; i < 64 * 16; i += 256) {
      int nn, kk;
      if (TB) { kk = i & 15; nn = i >> 4; } else { nn = i & 63; kk = i >> 6; }
      const int n = n0 + nn, k = k0 + kk;
      float v = 0.f;
      if (n < g.N && k < kend) v = TB ? B[(long)n * g.ldb + k] : B[(long)k * g.ldb + n];
      Bs[kk][nn] = v;
    }
    __syncthreads();
#pragma unroll
    for (int kk = 0; kk < 16; kk += 2)
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(As[kk + half][wm * 32 + col], Bs[kk + half][wn * 32 + col], acc, 0, 0, 0);
    __syncthreads();
  }
  const float* bias = g.bias ? g.bias + b0 * g.sBias0 + b1 * g.sBias1 : nullptr;
  const int n = n0 + wn * 32 + col;
  if (n >= g.N) return;
  const float bn = bias && kc == 0 ? bias[n] : 0.f;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int m = m0 + wm * 32 + crow(r, half);
    if (m >= g.M) continue;
    const float v = g.alpha * acc[r] + bn;
    float* c = C + (long)m * g.ldc + n;
    if (g.accumulate == 2 || g.ksplit > 1) unsafeAtomicAdd(c, v);          // several batches reduce into one C
    else *c = g.accumulate ? *c + v : v;
  }
}

// ------------------------------------------------------------------------------------------------
// The same product on the bf16 matrix cores at f32-class accuracy: every f32 operand element is split while it is
// staged into LDS, x = hi + lo (two bf16, 16 mantissa bits together), and each 16-deep k chunk costs three
// v_mfma_f32_32x32x16_bf16 (lo*hi + hi*lo + hi*hi; the dropped lo*lo term is 2^-16 relative) -- about 5x the f32
// instruction's rate.  Operands are staged without transposition whatever their layout in memory (see stage_load).
// 4 waves as 2 x 2, BM x BN in {128 x 128, 64 x 64}, k steps of 32 with three steps of global loads in flight.
// ------------------------------------------------------------------------------------------------
// Staging of one BR x 32 operand tile (f32 in memory -> hi / lo bf16 planes in LDS), two flavours:
//   T = false  source is k-contiguous (rows of 32 floats): LDS image [row][32] with XOR-swizzled 16-byte chunks; fragments by ds_read_b128
//   T = true   source is row-contiguous (memory [k][row]): LDS image [k][BR + 32], no transposition while staging;
//              the fragment's k-in-lane order comes from ds_read_b64_tr_b16 (two per plane and k chunk)
// Either way a thread moves float4's along the contiguous dimension (scalar fallback for unaligned / edge pieces).
// Loads are branch-free so that the compiler keeps counted s_waitcnt vmcnt: VEC = one float4 per piece (host guarantees
// 16-byte alignment and that pieces never straddle the K / row edge), else 4 dwords.
// Out-of-range pieces are read from this block of zeros: selecting the ADDRESS keeps the load free of any instruction that
// consumes its result (a `v = ok ? v : 0` after the load would make the wave wait for the data right there and undo the
// prefetch).
__device__ __attribute__((aligned(16))) float hvla_zero_piece[4] = {0.f, 0.f, 0.f, 0.f};

// The loads are written as inline asm so that the compiler does not track them: it would otherwise place its own
// (conservative, often vmcnt(0)) waits wherever it schedules the consumers and drain the younger stages with them.  The
// kernel waits for exactly one stage with a constant s_waitcnt vmcnt and pins the consumers behind it with tie().
// One stage of one operand in registers: P pieces of 4 consecutive elements, as float4 tuples (VEC) or as 4P scalars.
// Each register is written by exactly one asm load and first touched again by tie() after the wait -- never copied in
// between (the hardware does not interlock a VGPR read against a load still in flight).
template <int P, bool VEC> struct StageRegs;
template <int P> struct StageRegs<P, true> {
  f32x4 v[P];
  __device__ __forceinline__ float get(int j, int c) const { return v[j][c]; }
};
template <int P> struct StageRegs<P, false> {
  float f[4 * P];
  __device__ __forceinline__ float get(int j, int c) const { return f[4 * j + c]; }
};
#define HVLA_T(x) "+v"(x)
__device__ __forceinline__ void tie(StageRegs<4, true>& r) { asm volatile("" : HVLA_T(r.v[0]), HVLA_T(r.v[1]), HVLA_T(r.v[2]), HVLA_T(r.v[3])::"memory"); }
__device__ __forceinline__ void tie(StageRegs<2, true>& r) { asm volatile("" : HVLA_T(r.v[0]), HVLA_T(r.v[1])::"memory"); }
__device__ __forceinline__ void tie(StageRegs<2, false>& r) {
  asm volatile("" : HVLA_T(r.f[0]), HVLA_T(r.f[1]), HVLA_T(r.f[2]), HVLA_T(r.f[3]), HVLA_T(r.f[4]), HVLA_T(r.f[5]), HVLA_T(r.f[6]), HVLA_T(r.f[7])::"memory");
}
__device__ __forceinline__ void tie(StageRegs<4, false>& r) {
  asm volatile("" : HVLA_T(r.f[0]), HVLA_T(r.f[1]), HVLA_T(r.f[2]), HVLA_T(r.f[3]), HVLA_T(r.f[4]), HVLA_T(r.f[5]), HVLA_T(r.f[6]), HVLA_T(r.f[7]),
               HVLA_T(r.f[8]), HVLA_T(r.f[9]), HVLA_T(r.f[10]), HVLA_T(r.f[11]), HVLA_T(r.f[12]), HVLA_T(r.f[13]), HVLA_T(r.f[14]), HVLA_T(r.f[15])::"memory");
}
#undef HVLA_T

template <bool T, int BR, bool VEC, int NTH = 256>
__device__ __forceinline__ void stage_load(const float* __restrict__ P, int ld, int r0, int R, int k0, int kend,
                                           StageRegs<BR * 8 / NTH, VEC>& out) {
#pragma unroll
  for (int j = 0; j < BR * 8 / NTH; ++j) {
    const int idx = threadIdx.x + j * NTH;
    const int r = r0 + (T ? (idx % (BR / 4)) * 4 : idx >> 3), k = k0 + (T ? idx / (BR / 4) : (idx & 7) * 4);
    const bool ok = r < R && k < kend;
    const float* p = P + (T ? (long)k * ld + r : (long)r * ld + k);     // the 4 elements of a piece are adjacent in memory
    if constexpr (VEC) {
      const float* q = ok ? p : hvla_zero_piece;
      asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(out.v[j]) : "v"(q) : "memory");
    } else {
      const int lim = T ? R - r : kend - k;               // valid elements in this piece (<= 0: none)
      const float* z = hvla_zero_piece;
      const float *q0 = ok ? p : z, *q1 = ok && lim > 1 ? p + 1 : z, *q2 = ok && lim > 2 ? p + 2 : z, *q3 = ok && lim > 3 ? p + 3 : z;
      asm volatile("global_load_dword %0, %1, off" : "=v"(out.f[4 * j + 0]) : "v"(q0) : "memory");
      asm volatile("global_load_dword %0, %1, off" : "=v"(out.f[4 * j + 1]) : "v"(q1) : "memory");
      asm volatile("global_load_dword %0, %1, off" : "=v"(out.f[4 * j + 2]) : "v"(q2) : "memory");
      asm volatile("global_load_dword %0, %1, off" : "=v"(out.f[4 * j + 3]) : "v"(q3) : "memory");
    }
  }
}
template <bool T, int BR, bool VEC, int NTH = 256>
__device__ __forceinline__ void stage_store(__bf16* __restrict__ hi, __bf16* __restrict__ lo, const StageRegs<BR * 8 / NTH, VEC>& in) {
#pragma unroll
  for (int j = 0; j < BR * 8 / NTH; ++j) {
    const int idx = threadIdx.x + j * NTH;
    const int o = T ? (idx / (BR / 4)) * (BR + 32) + (idx % (BR / 4)) * 4
                    : (idx >> 3) * 32 + ((((idx & 7) >> 1) ^ ((idx >> 4) & 3)) << 3) + (idx & 1) * 4;   // row = idx >> 3, see load_frag
    bf16x4 h, l;
    __bf16 a, b;
    split1(in.get(j, 0), a, b); h[0] = a; l[0] = b;
    split1(in.get(j, 1), a, b); h[1] = a; l[1] = b;
    split1(in.get(j, 2), a, b); h[2] = a; l[2] = b;
    split1(in.get(j, 3), a, b); h[3] = a; l[3] = b;
    *reinterpret_cast<bf16x4*>(hi + o) = h;
    *reinterpret_cast<bf16x4*>(lo + o) = l;
  }
}
typedef short hvla_tr4 __attribute__((__vector_size__(4 * sizeof(short))));
__device__ __forceinline__ bf16x8 tr_frag(const __bf16* p0, const __bf16* p1) {   // p1 = 4 k rows below p0
  const hvla_tr4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) hvla_tr4*)p0);
  const hvla_tr4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) hvla_tr4*)p1);
  typedef short s8 __attribute__((ext_vector_type(8)));
  const s8 v = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
  return __builtin_bit_cast(bf16x8, v);
}
// fragment of the 32-row block starting at row rb, k chunk kk (0 / 16): lane l holds row rb + (l & 31), k = kk + 8 (l >> 5) + 0..7
template <bool T, int BR>
__device__ __forceinline__ Split8 load_frag(const __bf16* hi, const __bf16* lo, int rb, int kk, int lane) {
  Split8 f;
  if (!T) {
    // 64-byte rows, the 16-byte chunk index XORed with (row >> 1) & 3: the 8-byte stores of a half-wave (4 rows) and the
    // 16-byte reads of 8 consecutive rows both spread over all 64 banks (the padded [row][40] image had 2-way store conflicts)
    const int row = rb + (lane & 31);
    const int o = row * 32 + ((((kk >> 3) + (lane >> 5)) ^ ((row >> 1) & 3)) << 3);
    f.hi = *reinterpret_cast<const bf16x8*>(hi + o);
    f.lo = *reinterpret_cast<const bf16x8*>(lo + o);
  } else {
    // 16-lane group g = lane >> 4 covers rows rb + 16 (g & 1) .. + 15 at k half (g >> 1); lane 4q + p of the group
    // addresses LDS row (k) q, columns 4p .. 4p + 3 and receives its own column of the four rows
    constexpr int LDT = BR + 32;
    const int i = lane & 15, q = i >> 2, pp = i & 3;
    const int o = (kk + 8 * (lane >> 5) + q) * LDT + rb + 16 * ((lane >> 4) & 1) + 4 * pp;
    f.hi = tr_frag(hi + o, hi + o + 4 * LDT);
    f.lo = tr_frag(lo + o, lo + o + 4 * LDT);
  }
  return f;
}

// BM = 256 (with BN = 128): EIGHT waves as 4 x 2, the same 64 x 64 per wave -- for problems whose 128 x 128 grid is between one and
// two workgroups per CU (N = 768 at 8 224 rows: 390 tiles, 134 CUs with two and 122 with one; as 256 x 128 it is 198 tiles, one
// round, and a thread splits 24 instead of 32 values per k step).
template <bool TA, bool TB, int BM, int BN, bool VEC>
__global__ __launch_bounds__(BM == 256 ? 512 : 256, BM == 256 ? 1 : 2) void bgemm3_kernel(BG g) {
  constexpr int NTH = BM == 256 ? 512 : 256;               // threads
  constexpr int WM = BM == 256 ? 64 : BM / 2, WN = BN / 2, IM = WM / 32, IN = WN / 32;
  constexpr bool SA = TA, SB = !TB;                       // staging flavour: row-contiguous source?
  constexpr int NA = SA ? 32 * (BM + 32) : BM * 32, NB = SB ? 32 * (BN + 32) : BN * 32;
  // k steps of global loads in flight (register stages): three float4 stages, or two when every piece is 4 dwords
  // (the 6-bit vmcnt could not count three of those)
  constexpr int NST = VEC ? 3 : 2;
  constexpr int PA = BM * 8 / NTH, PB = BN * 8 / NTH;     // float4 pieces per thread and stage
  constexpr int LPS = (PA + PB) * (VEC ? 1 : 4);          // load instructions per stage
  __shared__ __attribute__((aligned(16))) __bf16 Ah[NA], Al[NA], Bh[NB], Bl[NB];
  const int zb = blockIdx.z / g.ksplit, kc = blockIdx.z % g.ksplit;
  const int b0 = zb / g.nb1, b1 = zb % g.nb1;
  const int kchunk = ((g.K + g.ksplit - 1) / g.ksplit + 31) & ~31;
  const int kbeg = kc * kchunk, kend = kbeg + kchunk < g.K ? kbeg + kchunk : g.K;
  const float* A = g.A + b0 * g.sA0 + b1 * g.sA1;
  const float* B = g.B + b0 * g.sB0 + b1 * g.sB1;
  float* C = g.C + b0 * g.sC0 + b1 * g.sC1;
  const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int wm = wave >> 1, wn = wave & 1, col = lane & 31, half = lane >> 5;
  f32x16 acc[IM][IN];
#pragma unroll
  for (int a = 0; a < IM; ++a)
#pragma unroll
    for (int b = 0; b < IN; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
  StageRegs<PA, VEC> ra[NST];
  StageRegs<PB, VEC> rb[NST];
#pragma unroll
  for (int s2 = 0; s2 < NST; ++s2) {                      // steps past kend read the block of zeros
    stage_load<SA, BM, VEC, NTH>(A, g.lda, m0, g.M, kbeg + 32 * s2, kend, ra[s2]);
    stage_load<SB, BN, VEC, NTH>(B, g.ldb, n0, g.N, kbeg + 32 * s2, kend, rb[s2]);
  }
  // Whole groups of NST steps (a step past kend multiplies staged zeros), every step issues its refill: the number of
  // loads in flight is the same at every wait, so the wait for the oldest stage is the constant vmcnt((NST - 1) * LPS).
  for (int k0 = kbeg; k0 < kend; k0 += 32 * NST) {
#pragma unroll
    for (int s2 = 0; s2 < NST; ++s2) {
      const int kcur = k0 + 32 * s2;
      // raw barriers: a __syncthreads() would also drain the younger register stages' global loads (vmcnt(0))
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");     // the previous step's fragment reads are done
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NST - 1) * LPS) : "memory");     // the oldest stage has landed
      tie(ra[s2]);
      tie(rb[s2]);
      stage_store<SA, BM, VEC, NTH>(Ah, Al, ra[s2]);
      stage_store<SB, BN, VEC, NTH>(Bh, Bl, rb[s2]);
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");     // the LDS image is complete
      // refill this register stage NST steps ahead
      stage_load<SA, BM, VEC, NTH>(A, g.lda, m0, g.M, kcur + 32 * NST, kend, ra[s2]);
      stage_load<SB, BN, VEC, NTH>(B, g.ldb, n0, g.N, kcur + 32 * NST, kend, rb[s2]);
#pragma unroll
      for (int kk = 0; kk < 32; kk += 16) {
        Split8 fa[IM], fb[IN];
#pragma unroll
        for (int a = 0; a < IM; ++a) fa[a] = load_frag<SA, BM>(Ah, Al, wm * WM + a * 32, kk, lane);
#pragma unroll
        for (int b = 0; b < IN; ++b) fb[b] = load_frag<SB, BN>(Bh, Bl, wn * WN + b * 32, kk, lane);
#pragma unroll
        for (int a = 0; a < IM; ++a)
#pragma unroll
          for (int b = 0; b < IN; ++b) {
            acc[a][b] = mma32_x3(fa[a], fb[b], acc[a][b]);
          }
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // the refills past kend (zeros) are still in flight
  const float* bias = g.bias && kc == 0 ? g.bias + b0 * g.sBias0 + b1 * g.sBias1 : nullptr;
  // Straight-line epilogue per (full tile?, store mode): with per-element `m < M` branches and a run-time mode inside the
  // loops the compiler drains the memory counter in every predicated block, i.e. one store round trip after the other.
  auto store = [&](auto fullc, auto modec) {
    constexpr bool FULL = decltype(fullc)::value;
    constexpr int MODE = decltype(modec)::value;    // 0 store, 1 read-modify-write, 2 atomic add
#pragma unroll
    for (int b = 0; b < IN; ++b) {
      const int n = n0 + wn * WN + b * 32 + col;
      if (!FULL && n >= g.N) continue;
      const float bn = bias ? bias[n] : 0.f;
#pragma unroll
      for (int a = 0; a < IM; ++a) {
        float old[16];
        if constexpr (MODE == 1) {
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int m = m0 + wm * WM + a * 32 + crow(r, half);
            old[r] = (FULL || m < g.M) ? C[(long)m * g.ldc + n] : 0.f;
          }
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int m = m0 + wm * WM + a * 32 + crow(r, half);
          if (!FULL && m >= g.M) continue;
          const float v = g.alpha * acc[a][b][r] + bn;
          float* c = C + (long)m * g.ldc + n;
          if constexpr (MODE == 2) unsafeAtomicAdd(c, v);
          else if constexpr (MODE == 1) *c = old[r] + v;
          else *c = v;
        }
      }
    }
  };
  using Yes = std::integral_constant<bool, true>;
  using No = std::integral_constant<bool, false>;
  using M0 = std::integral_constant<int, 0>;
  using M1 = std::integral_constant<int, 1>;
  using M2 = std::integral_constant<int, 2>;
  const bool full = m0 + BM <= g.M && n0 + BN <= g.N;
  const int mode = (g.accumulate == 2 || g.ksplit > 1) ? 2 : (g.accumulate ? 1 : 0);
  if (full) {
    if (mode == 2) store(Yes{}, M2{}); else if (mode == 1) store(Yes{}, M1{}); else store(Yes{}, M0{});
  } else {
    if (mode == 2) store(No{}, M2{}); else if (mode == 1) store(No{}, M1{}); else store(No{}, M0{});
  }
}

template <int BM, int BN, bool VEC>
static void launch_bgemm3(hipStream_t st, bool ta, bool tb, const BG& g, dim3 grid) {
  constexpr int NTH = BM == 256 ? 512 : 256;
  if (!ta && !tb) hipLaunchKernelGGL((bgemm3_kernel<false, false, BM, BN, VEC>), grid, dim3(NTH), 0, st, g);
  else if (!ta && tb) hipLaunchKernelGGL((bgemm3_kernel<false, true, BM, BN, VEC>), grid, dim3(NTH), 0, st, g);
  else if (ta && !tb) hipLaunchKernelGGL((bgemm3_kernel<true, false, BM, BN, VEC>), grid, dim3(NTH), 0, st, g);
  else hipLaunchKernelGGL((bgemm3_kernel<true, true, BM, BN, VEC>), grid, dim3(NTH), 0, st, g);
}

// the exact-f32 matrix instruction (bitwise fmaf chains) instead of the split-bf16 path: libhvla_bench.so only
// (hvla_debug_train_gemm_exact); the product library has no switch that changes its arithmetic
#ifdef HVLA_BENCH_HOOKS
static bool g_train_gemm_exact = false;
void set_train_gemm_exact(bool on) { g_train_gemm_exact = on; }
static bool train_gemm_exact() { return g_train_gemm_exact; }
#else
static constexpr bool train_gemm_exact() { return false; }
#endif

// Optional live timing of the batched GEMM launches (hvla_train_profile: bench.py --finetune's `roofline` block): HIP events on
// the launch stream around every bgemm() call and the f32-equivalent work of the call (2 M N K per batch entry; the split-bf16
// kernel spends three matrix instructions per product).  Off by default.  One timer per DEVICE (a context belongs to one device and
// the API entry points make it current before they launch; ADVICE r4: a process-wide timer mixed the contexts of two devices and
// never gave its events back): events are created on that device, destroyed by train_gemm_timer_release() from hvla_destroy, and a
// launch into a stream that is being captured is not timed (events recorded inside a capture cannot be read).  Not thread-safe
// (one training stream per device).
namespace {
struct GemmTimer {
  bool on = false;
  int users = 0;                      // contexts on this device that switched the timer on and are still alive (the events
                                      // go back when the LAST of them is destroyed, not when any of them is)
  std::vector<hipEvent_t> a, b;
  size_t used = 0;
  double flops = 0.0;
};
GemmTimer g_gemm_timer[64];
int g_gemm_timers_on = 0;             // devices whose timer is on: bgemm() asks for the current device only when this is not zero
GemmTimer* gemm_timer_here() {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
  return &g_gemm_timer[dev];
}
}  // namespace
void train_gemm_timer(bool on, bool first_use_by_this_context) {
  GemmTimer* t = gemm_timer_here();
  if (!t) return;
  if (first_use_by_this_context) ++t->users;
  if (t->on != on) g_gemm_timers_on += on ? 1 : -1;
  t->on = on;
}
void train_gemm_timer_release() {
  GemmTimer* t = gemm_timer_here();
  if (!t || t->users <= 0 || --t->users > 0) return;      // another live context of this device still uses the timer
  if (t->on) --g_gemm_timers_on;
  for (hipEvent_t e : t->a) (void)hipEventDestroy(e);
  for (hipEvent_t e : t->b) (void)hipEventDestroy(e);
  t->a.clear(), t->b.clear();
  t->used = 0, t->flops = 0.0, t->on = false;
}
hipError_t train_gemm_timer_read(float* ms, double* flops, int* launches) {
  *ms = 0.f; *flops = 0.0; *launches = 0;
  GemmTimer* tp = gemm_timer_here();
  if (!tp) return hipErrorInvalidDevice;
  GemmTimer& t = *tp;
  *flops = t.flops; *launches = (int)t.used;
  for (size_t i = 0; i < t.used; ++i) {
    hipError_t e = hipEventSynchronize(t.b[i]);
    if (e != hipSuccess) return e;
    float x = 0.f;
    if ((e = hipEventElapsedTime(&x, t.a[i], t.b[i])) != hipSuccess) return e;
    *ms += x;
  }
  t.used = 0; t.flops = 0.0;
  return hipSuccess;
}
static void bgemm_launch(hipStream_t st, bool ta, bool tb, BG g, int nb0);
void bgemm(hipStream_t st, bool ta, bool tb, BG g, int nb0) {
  if (g_gemm_timers_on == 0) { bgemm_launch(st, ta, tb, g, nb0); return; }
  GemmTimer* tp = gemm_timer_here();
  hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
  if (!tp || !tp->on || hipStreamIsCapturing(st, &cap) != hipSuccess || cap != hipStreamCaptureStatusNone) { bgemm_launch(st, ta, tb, g, nb0); return; }
  GemmTimer& t = *tp;
  if (t.used == t.a.size()) {
    hipEvent_t x, y;
    if (hipEventCreate(&x) != hipSuccess) { bgemm_launch(st, ta, tb, g, nb0); return; }
    if (hipEventCreate(&y) != hipSuccess) { (void)hipEventDestroy(x); bgemm_launch(st, ta, tb, g, nb0); return; }
    t.a.push_back(x), t.b.push_back(y);
  }
  (void)hipEventRecord(t.a[t.used], st);
  bgemm_launch(st, ta, tb, g, nb0);
  (void)hipEventRecord(t.b[t.used], st);
  ++t.used;
  t.flops += 2.0 * g.M * g.N * g.K * (double)nb0 * g.nb1;
}
static void bgemm_launch(hipStream_t st, bool ta, bool tb, BG g, int nb0) {
  const bool exact = train_gemm_exact();
  const int T = !exact && g.M >= 96 && g.N >= 96 ? 128 : 64;
  // deep-K products onto few output tiles (shared-weight gradients: K = all rows of the batch) would leave most CUs
  // idle: cut K so that the grid has >= ~512 workgroups (two per CU; more only adds atomic traffic); legal whenever the result is accumulated (C zeroed before)
  const long tiles = (long)((g.N + T - 1) / T) * ((g.M + T - 1) / T) * nb0 * g.nb1;
  constexpr long want = 256;
  // two 128 x 128 workgroups fit on a CU: 512 workgroups (same box, alternating: step 37.70 -> 37.47 ms; 1 024: 39.2)
  constexpr long want_split = 512;
  if (g.allow_split && g.accumulate != 0 && g.ksplit == 1 && tiles < want_split && g.K >= 256) {
    long ks = (want_split + tiles - 1) / tiles, kmax = g.K / 128;
    g.ksplit = (int)(ks < kmax ? ks : kmax);
    if (g.ksplit < 1) g.ksplit = 1;
  }
  dim3 grid((g.N + T - 1) / T, (g.M + T - 1) / T, nb0 * g.nb1 * g.ksplit);
  if (!exact) {
    // float4 staging: 16-byte aligned pieces that never straddle an edge (k-contiguous operand: K % 4 == 0;
    // row-contiguous operand: its row count % 4 == 0); otherwise the all-dword variant
    auto al = [](const float* p, int ld, long s0, long s1, int extent) {
      return ((uintptr_t)p % 16 == 0) && ld % 4 == 0 && s0 % 4 == 0 && s1 % 4 == 0 && extent % 4 == 0;
    };
    const int a_extent = ta ? g.M : g.K;
    const bool vec = al(g.A, g.lda, g.sA0, g.sA1, g.a_padded && g.lda >= ((a_extent + 3) & ~3) ? 4 : a_extent) && al(g.B, g.ldb, g.sB0, g.sB1, tb ? g.K : g.N);
    // between one and two 128 x 128 workgroups per CU: 256 x 128 tiles (eight waves) make it one round
    const long t128 = (long)grid.x * grid.y * grid.z, t256 = (long)grid.x * ((g.M + 255) / 256) * grid.z;
    if (T == 128 && vec && t128 > want && t128 <= 2 * want && t256 <= want) {
      launch_bgemm3<256, 128, true>(st, ta, tb, g, dim3(grid.x, (g.M + 255) / 256, grid.z));
      return;
    }
    if (T == 128) { if (vec) launch_bgemm3<128, 128, true>(st, ta, tb, g, grid); else launch_bgemm3<128, 128, false>(st, ta, tb, g, grid); }
    else { if (vec) launch_bgemm3<64, 64, true>(st, ta, tb, g, grid); else launch_bgemm3<64, 64, false>(st, ta, tb, g, grid); }
    return;
  }
  if (!ta && !tb) hipLaunchKernelGGL((bgemm_kernel<false, false>), grid, dim3(256), 0, st, g);
  else if (!ta && tb) hipLaunchKernelGGL((bgemm_kernel<false, true>), grid, dim3(256), 0, st, g);
  else if (ta && !tb) hipLaunchKernelGGL((bgemm_kernel<true, false>), grid, dim3(256), 0, st, g);
  else hipLaunchKernelGGL((bgemm_kernel<true, true>), grid, dim3(256), 0, st, g);
}

// ------------------------------------------------------------------------------------------------
// row kernels (one wave per row).  Rows are [nb][S][D]; per-row parameters come from p + b * pstride.
// ------------------------------------------------------------------------------------------------
__global__ void ln_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, float* __restrict__ mean,
                              float* __restrict__ rstd, const float* __restrict__ scale,
                              const float* __restrict__ bias, long pstride, int rows, int S, int D) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= rows) return;
  const float* xr = x + (long)row * D;
  float s = 0.f, q = 0.f;
  for (int c = lane; c < D; c += 64) { s += xr[c]; q += xr[c] * xr[c]; }
  for (int o = 32; o; o >>= 1) { s += __shfl_xor(s, o, 64); q += __shfl_xor(q, o, 64); }
  const float m = s / D, var = fmaxf(0.f, q / D - m * m), r = rsqrtf(var + 1e-6f);
  const float* sc = scale + (long)(row / S) * pstride;
  const float* bi = bias + (long)(row / S) * pstride;
  for (int c = lane; c < D; c += 64) y[(long)row * D + c] = (xr[c] - m) * r * sc[c] + bi[c];
  if (lane == 0) { mean[row] = m; rstd[row] = r; }
}

// dx (+)= LN backward; per-row contributions to dscale / dbias are accumulated with atomics into
// dscale + b * pstride (per-episode parameters: pstride = G; shared parameters: pstride = 0).
__global__ void ln_bwd_kernel(const float* __restrict__ x, const float* __restrict__ dy, const float* __restrict__ mean,
                              const float* __restrict__ rstd, const float* __restrict__ scale, float* __restrict__ dx,
                              float* __restrict__ dscale, float* __restrict__ dbias, long pstride, int rows, int S,
                              int D, int accumulate) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= rows) return;
  const float m = mean[row], r = rstd[row];
  const float* xr = x + (long)row * D;
  const float* dyr = dy + (long)row * D;
  const float* sc = scale + (long)(row / S) * pstride;
  float s1 = 0.f, s2 = 0.f;                               // sum(dxhat), sum(dxhat * xhat)
  for (int c = lane; c < D; c += 64) {
    const float xh = (xr[c] - m) * r, dxh = dyr[c] * sc[c];
    s1 += dxh; s2 += dxh * xh;
  }
  for (int o = 32; o; o >>= 1) { s1 += __shfl_xor(s1, o, 64); s2 += __shfl_xor(s2, o, 64); }
  float* ds = dscale + (long)(row / S) * pstride;
  float* db = dbias + (long)(row / S) * pstride;
  for (int c = lane; c < D; c += 64) {
    const float xh = (xr[c] - m) * r, dxh = dyr[c] * sc[c];
    const float v = r * (dxh - s1 / D - xh * s2 / D);
    float* d = dx + (long)row * D + c;
    *d = accumulate ? *d + v : v;
    if (dscale) {
      unsafeAtomicAdd(ds + c, dyr[c] * xh);
      unsafeAtomicAdd(db + c, dyr[c]);
    }
  }
}

// LN backward for SHARED parameters in one pass (round 3; before: ln_bwd_kernel + ln_pgrad_kernel, each reading x and dy: 28 + 22 us
// for 8 224 x 768): a workgroup of eight waves takes 64 rows, a wave its rows one after the other with x and dy of a row in
// registers (float4, read once), dx (+)= r (dxhat - mean(dxhat) - xhat mean(dxhat xhat)), and every lane keeps the dscale / dbias
// sums of its columns; the waves are combined through LDS, one atomic per (workgroup, column).  D % 4 == 0, D <= 1024.
constexpr int LNB_WAVES = 8;
__global__ __launch_bounds__(LNB_WAVES * 64) void ln_bwd_shared_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                            const float* __restrict__ mean, const float* __restrict__ rstd,
                                                            const float* __restrict__ scale, float* __restrict__ dx,
                                                            float* __restrict__ dscale, float* __restrict__ dbias, int rows, int D,
                                                            int accumulate) {
  __shared__ f32x4 red[2][LNB_WAVES][256];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, n4 = D / 4;
  f32x4 sc[4], ga[4], gb[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = lane + 64 * i;
    ga[i] = gb[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    sc[i] = c < n4 ? reinterpret_cast<const f32x4*>(scale)[c] : ga[i];
  }
  const int r0 = blockIdx.x * 64, r1 = r0 + 64 < rows ? r0 + 64 : rows;
  const float invD = 1.f / (float)D;
  for (int row = r0 + wave; row < r1; row += LNB_WAVES) {
    const f32x4* xr = reinterpret_cast<const f32x4*>(x + (long)row * D);
    const f32x4* dr = reinterpret_cast<const f32x4*>(dy + (long)row * D);
    f32x4* ox = reinterpret_cast<f32x4*>(dx + (long)row * D);
    f32x4 xv[4], dv[4], old[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int c = lane + 64 * i;
      xv[i] = dv[i] = old[i] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (c < n4) {
        xv[i] = xr[c];
        dv[i] = dr[c];
        if (accumulate) old[i] = ox[c];
      }
    }
    const float m = mean[row], r = rstd[row];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if (lane + 64 * i < n4) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float xh = (xv[i][j] - m) * r, dxh = dv[i][j] * sc[i][j];
          s1 += dxh;
          s2 = fmaf(dxh, xh, s2);
          ga[i][j] = fmaf(dv[i][j], xh, ga[i][j]);
          gb[i][j] += dv[i][j];
          xv[i][j] = xh;                     // keep xhat and dxhat for the second half
          dv[i][j] = dxh;
        }
      }
    }
    for (int o = 32; o; o >>= 1) { s1 += __shfl_xor(s1, o, 64); s2 += __shfl_xor(s2, o, 64); }
    s1 *= invD;
    s2 *= invD;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int c = lane + 64 * i;
      if (c < n4) {
        f32x4 v;
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = old[i][j] + r * (dv[i][j] - s1 - xv[i][j] * s2);
        ox[c] = v;
      }
    }
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) red[0][wave][lane + 64 * i] = ga[i], red[1][wave][lane + 64 * i] = gb[i];
  __syncthreads();
  for (int c = threadIdx.x; c < 2 * n4; c += LNB_WAVES * 64) {
    const int which = c >= n4, cc = c - which * n4;
    f32x4 t = red[which][0][cc];
#pragma unroll
    for (int w = 1; w < LNB_WAVES; ++w) t += red[which][w][cc];
    float* o = (which ? dbias : dscale) + 4 * cc;
#pragma unroll
    for (int j = 0; j < 4; ++j) unsafeAtomicAdd(o + j, t[j]);
  }
}

// shared LayerNorm parameters: dscale[c] += sum_r dy[r][c] xhat[r][c], dbias[c] += sum_r dy[r][c]; 64 rows per
// block (blockIdx.y) reduced in registers, one atomic per (block, column) instead of one per element.
__global__ void ln_pgrad_kernel(const float* __restrict__ x, const float* __restrict__ dy, const float* __restrict__ mean,
                                const float* __restrict__ rstd, float* __restrict__ dscale, float* __restrict__ dbias,
                                int rows, int D) {
  const int c = blockIdx.x * 64 + threadIdx.x;
  if (c >= D) return;
  const int r0 = blockIdx.y * 64, r1 = r0 + 64 < rows ? r0 + 64 : rows;
  float a = 0.f, b = 0.f;
  for (int r = r0; r < r1; ++r) {
    const float d = dy[(long)r * D + c];
    a = fmaf(d, (x[(long)r * D + c] - mean[r]) * rstd[r], a);
    b += d;
  }
  unsafeAtomicAdd(dscale + c, a);
  unsafeAtomicAdd(dbias + c, b);
}

// LayerScale (DINOv2 layer_scale{1,2}.lambda1): out = res + ls (.) y
__global__ void scale_add_kernel(float* __restrict__ out, const float* __restrict__ res, const float* __restrict__ y,
                                 const float* __restrict__ ls, long n, int D) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
    out[i] = res[i] + ls[i % D] * y[i];
}
// float4 form (D % 4 == 0, 16-byte aligned): no 64-bit modulo per element
__global__ void scale_add4_kernel(f32x4* __restrict__ out, const f32x4* __restrict__ res, const f32x4* __restrict__ y,
                                  const f32x4* __restrict__ ls, long n4, int d4) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
    const f32x4 l = ls[(int)(i % d4)], r = res[i], v = y[i];
    out[i] = f32x4{fmaf(l[0], v[0], r[0]), fmaf(l[1], v[1], r[1]), fmaf(l[2], v[2], r[2]), fmaf(l[3], v[3], r[3])};
  }
}
// backward: dy = dx (.) ls ; dls[c] += sum_r dx[r][c] y[r][c]  (row chunks of 64 as in ln_pgrad_kernel)
__global__ void ls_bwd_kernel(const float* __restrict__ dx, const float* __restrict__ y, const float* __restrict__ ls,
                              float* __restrict__ dy, float* __restrict__ dls, int rows, int D) {
  const int c = blockIdx.x * 64 + threadIdx.x;
  if (c >= D) return;
  const int r0 = blockIdx.y * 64, r1 = r0 + 64 < rows ? r0 + 64 : rows;
  const float l = ls[c];
  float a = 0.f;
  for (int r = r0; r < r1; ++r) {
    const float d = dx[(long)r * D + c];
    a = fmaf(d, y[(long)r * D + c], a);
    dy[(long)r * D + c] = d * l;
  }
  unsafeAtomicAdd(dls + c, a);
}

// the same with float4 columns, eight waves x 8 rows x 2 batches per workgroup (all loads of a batch in flight), the dls partials
// of the waves combined through LDS: 34 -> ~15 us for 8 224 x 768.  D % 4 == 0, 16-byte aligned rows.
__global__ __launch_bounds__(512) void ls_bwd4_kernel(const float* __restrict__ dx, const float* __restrict__ y, const float* __restrict__ ls,
                                                      float* __restrict__ dy, float* __restrict__ dls, int rows, int D) {
  __shared__ f32x4 part[8][64];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int c4 = blockIdx.x * 64 + lane, n4 = D / 4;
  f32x4 a = f32x4{0.f, 0.f, 0.f, 0.f};
  if (c4 < n4) {
    const f32x4 l = reinterpret_cast<const f32x4*>(ls)[c4];
#pragma unroll
    for (int bt = 0; bt < 2; ++bt) {
      const int r0 = (blockIdx.y * 2 + bt) * 64 + wave * 8;
      f32x4 d[8], v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int r = r0 + u < rows ? r0 + u : rows - 1;
        d[u] = reinterpret_cast<const f32x4*>(dx + (long)r * D)[c4];
        v[u] = reinterpret_cast<const f32x4*>(y + (long)r * D)[c4];
      }
#pragma unroll
      for (int u = 0; u < 8; ++u)
        if (r0 + u < rows) {
#pragma unroll
          for (int j = 0; j < 4; ++j) a[j] = fmaf(d[u][j], v[u][j], a[j]);
          reinterpret_cast<f32x4*>(dy + (long)(r0 + u) * D)[c4] = d[u] * l;
        }
    }
  }
  part[wave][lane] = a;
  __syncthreads();
  if (wave == 0 && c4 < n4) {
    f32x4 t = part[0][lane];
#pragma unroll
    for (int w = 1; w < 8; ++w) t += part[w][lane];
#pragma unroll
    for (int j = 0; j < 4; ++j) unsafeAtomicAdd(dls + 4 * c4 + j, t[j]);
  }
}

// DINOv2 input side (base_vit.py:111-122 + HF Dinov2Embeddings): normalised patch matrix [B*P][p*p*3] in the flax
// conv kernel's (dy, dx, c) order, then x[b][0] = cls + pos[0], x[b][1+t] = patch_t W + b + pos[1+t].
__global__ void im2col_f32_kernel(const uint8_t* __restrict__ img, float* __restrict__ out, int B, int HW, int p) {
  const int gp = HW / p, K = p * p * 3;
  const long n = (long)B * gp * gp * K;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const int k = (int)(i % K);
    const long t = i / K;
    const int px = (int)(t % gp), py = (int)((t / gp) % gp), b = (int)(t / ((long)gp * gp));
    const int c = k % 3, dx = (k / 3) % p, dy = k / (3 * p);
    const float mean = c == 0 ? 0.485f : c == 1 ? 0.456f : 0.406f, sd = c == 0 ? 0.229f : c == 1 ? 0.224f : 0.225f;
    const uint8_t v = img[(((long)b * HW + py * p + dy) * HW + px * p + dx) * 3 + c];
    out[i] = ((float)v / 255.f - mean) / sd;
  }
}
__global__ void enc_x0_kernel(float* __restrict__ x, const float* __restrict__ cls, const float* __restrict__ pos, int B,
                              int S, int E) {
  const long n = (long)B * S * E;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const int c = (int)(i % E), t = (int)((i / E) % S);
    x[i] = (t == 0 ? cls[c] : x[i]) + pos[(long)t * E + c];
  }
}

// masked softmax over the last dim of [nmat][R][Cc] in place.  mode 2: no mask; mode 0: policy mask (rows < R-1 cannot see
// column Cc-1, base_vit.py:209-214); mode 1: context mask (hypernetwork.py:149-181) from attn_mask[b][T].
__global__ void softmax_fwd_kernel(float* __restrict__ p, int nmat, int R, int Cc, int mode,
                                   const int64_t* __restrict__ am, int heads, int ld) {      // ld >= Cc: row stride; [Cc, ld) is set to 0
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= nmat * R) return;
  const int mat = row / R, q = row % R;
  float* pr = p + (long)row * ld;
  if (lane < ld - Cc) pr[Cc + lane] = 0.f;
  auto keep = [&](int k) {
    if (mode == 2) return true;                                    // DINOv2 encoder: dense attention
    if (mode == 0) return !(q < R - 1 && k == Cc - 1);
    const int T = Cc - 2;
    if (k < T) return am[(long)(mat / heads) * T + k] != 0;
    if (k == T) return true;
    return q == T + 1;
  };
  float mx = -3.4e38f;
  for (int k = lane; k < Cc; k += 64) if (keep(k)) mx = fmaxf(mx, pr[k]);
  for (int o = 32; o; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
  float sum = 0.f;
  for (int k = lane; k < Cc; k += 64) {
    const float e = keep(k) ? __expf(pr[k] - mx) : 0.f;
    pr[k] = e;
    sum += e;
  }
  for (int o = 32; o; o >>= 1) sum += __shfl_xor(sum, o, 64);
  const float inv = 1.f / sum;
  for (int k = lane; k < Cc; k += 64) pr[k] *= inv;
}

// ds = p * (dp - sum_k dp p), in place on dp
__global__ void softmax_bwd_kernel(const float* __restrict__ p, float* __restrict__ dp, int rows, int Cc, int ld) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= rows) return;
  const float* pr = p + (long)row * ld;
  float* dr = dp + (long)row * ld;
  if (lane < ld - Cc) dr[Cc + lane] = 0.f;                 // ds is the A operand of two float4-staged products: zeros behind the row
  float s = 0.f;
  for (int k = lane; k < Cc; k += 64) s += pr[k] * dr[k];
  for (int o = 32; o; o >>= 1) s += __shfl_xor(s, o, 64);
  for (int k = lane; k < Cc; k += 64) dr[k] = pr[k] * (dr[k] - s);
}

__device__ __forceinline__ float dgelu_tanh(float x) {
  const float k0 = 0.7978845608028654f, k1 = 0.044715f;
  const float u = k0 * (x + k1 * x * x * x), t = tanhf(u);
  return 0.5f * (1.f + t) + 0.5f * x * (1.f - t * t) * k0 * (1.f + 3.f * k1 * x * x);
}
// erf form (torch nn.GELU(), the DINOv2 MLP): exact erff here, the training path is f32 end to end
__device__ __forceinline__ float gelu_erf_exact(float x) { return 0.5f * x * (1.f + erff(x * 0.7071067811865476f)); }
__device__ __forceinline__ float dgelu_erf(float x) {
  return 0.5f * (1.f + erff(x * 0.7071067811865476f)) + x * 0.3989422804014327f * __expf(-0.5f * x * x);
}
__global__ void gelu_fwd_kernel(const float* __restrict__ u, float* __restrict__ g, long n, int erf_form) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
    g[i] = erf_form ? gelu_erf_exact(u[i]) : gelu_tanh(u[i]);
}
__global__ void gelu_bwd_kernel(const float* __restrict__ u, float* __restrict__ dg, long n, int erf_form) {   // dg -> du in place
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
    dg[i] *= erf_form ? dgelu_erf(u[i]) : dgelu_tanh(u[i]);
}

// column sums: out[b * ostride + n] += sum_{r < rows_used} x[b][r][n]  (bias gradients).  blockIdx.z splits the rows
// into chunks of 64 that are reduced with one atomic each, so long shared-parameter reductions stay parallel.
__global__ void colsum_kernel(const float* __restrict__ x, float* __restrict__ out, long ostride, int R, int rows_used,
                              int N, int nb) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x, b = blockIdx.y;
  if (n >= N || b >= nb) return;
  const int r0 = blockIdx.z * 64, r1 = r0 + 64 < rows_used ? r0 + 64 : rows_used;
  float s = 0.f;
  for (int r = r0; r < r1; ++r) s += x[((long)b * R + r) * N + n];
  unsafeAtomicAdd(out + (long)b * ostride + n, s);
}

// the same for 16-byte aligned rows with N % 4 == 0: a thread owns 4 columns (1 KiB per wave and row), 32 rows per chunk,
// eight row loads in flight
__global__ void colsum4_kernel(const float* __restrict__ x, float* __restrict__ out, long ostride, int R, int rows_used,
                               int N, int nb) {
  const int n = (blockIdx.x * blockDim.x + threadIdx.x) * 4, b = blockIdx.y;
  if (n >= N || b >= nb) return;
  const int r0 = blockIdx.z * 32, r1 = r0 + 32 < rows_used ? r0 + 32 : rows_used;
  const float* p = x + ((long)b * R + r0) * N + n;
  f32x4 s = f32x4{0.f, 0.f, 0.f, 0.f};
  int r = r0;
  for (; r + 8 <= r1; r += 8, p += 8L * N) {
    f32x4 v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const f32x4*>(p + (long)u * N);
#pragma unroll
    for (int u = 0; u < 8; ++u) s += v[u];
  }
  for (; r < r1; ++r, p += N) s += *reinterpret_cast<const f32x4*>(p);
  float* o = out + (long)b * ostride + n;
#pragma unroll
  for (int c = 0; c < 4; ++c) unsafeAtomicAdd(o + c, s[c]);
}

// The long shared-parameter reductions of the encoder (8 224 rows x 768 / 3 072 columns): the two kernels above have 800-1 500
// waves of 8 KB in flight for 25-100 MB (17-23 us for the narrow matrices: 1.5 TB/s).  Here a workgroup is eight waves x 8 rows
// x 2 batches, every row load of a batch in flight at once, the eight partial rows combined through LDS, one atomic per column
// and workgroup (four waves x 8 rows x 1 batch, twice the atomics of before, was SLOWER than the old kernels: 37 us).
constexpr int CSB_WAVES = 8, CSB_BATCH = 2;          // 128 rows per workgroup: half the atomics of the 64-row chunks above
__global__ __launch_bounds__(CSB_WAVES * 64) void colsum4b_kernel(const float* __restrict__ x, float* __restrict__ out, int rows, int N) {
  __shared__ f32x4 part[CSB_WAVES][64];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int n = (blockIdx.x * 64 + lane) * 4;
  f32x4 s = f32x4{0.f, 0.f, 0.f, 0.f};
  if (n < N) {
#pragma unroll
    for (int bt = 0; bt < CSB_BATCH; ++bt) {
      const int r0 = (blockIdx.z * CSB_BATCH + bt) * (CSB_WAVES * 8) + wave * 8;
      f32x4 v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int r = r0 + u < rows ? r0 + u : rows - 1;
        v[u] = *reinterpret_cast<const f32x4*>(x + (long)r * N + n);
      }
#pragma unroll
      for (int u = 0; u < 8; ++u)
        if (r0 + u < rows) s += v[u];
    }
  }
  part[wave][lane] = s;
  __syncthreads();
  if (wave == 0 && n < N) {
    f32x4 t = part[0][lane];
#pragma unroll
    for (int w = 1; w < CSB_WAVES; ++w) t += part[w][lane];
#pragma unroll
    for (int c = 0; c < 4; ++c) unsafeAtomicAdd(out + n + c, t[c]);
  }
}

// x0 assembly: rows < P already hold tokens.Wp + bp; row P = 0; += pos  (base_vit.py:182-204)
__global__ void x0_finish_kernel(float* __restrict__ x, const float* __restrict__ pos, long pstride, int S, int D, int nb) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long)nb * S * D) return;
  const int c = (int)(i % D), t = (int)((i / D) % S), b = (int)(i / ((long)S * D));
  const float base = t == S - 1 ? 0.f : x[i];
  x[i] = base + pos[(long)b * pstride + (long)t * D + c];
}
__global__ void add_kernel(float* __restrict__ dst, const float* __restrict__ src, long n) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) dst[i] += src[i];
}
// dst[b * dstride + i] += src[b * n + i]
__global__ void add_strided_kernel(float* __restrict__ dst, long dstride, const float* __restrict__ src, long n, int nb) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long)nb * n) return;
  dst[(i / n) * dstride + (i % n)] += src[i];
}

// mix head + loss, forward and backward for the action-token row (one thread block of 64 per episode).
// e = LN_f(x[S-1]); z = e.Wc + bc (A values), l = e.Wd + bd (Hz); cont = tanh(z / ts) * ma.
// Writes loss[b]; if dxrow != null also d/d(theta head leaves) and d/dx[S-1] for loss_total = mean_b loss[b].
struct HeadP {
  const float* x; long xstride;             // [B][S][D], row S-1 used
  const float* theta; float* dtheta; long G;
  int o_wc, o_bc, o_wd, o_bd, o_ns, o_nb;   // leaf offsets in theta
  const float* target; const uint8_t* tmask; const uint8_t* amask;
  float* loss; float* dxrow;                // [B][D]
  float* actions; float* logits;            // optional outputs
  int B, S, D, Hz, ad; float tanh_scale, max_action;
  int clip_target;                          // action_heads.py:499-500
};
__global__ void head_loss_kernel(HeadP p) {
  const int b = blockIdx.x, lane = threadIdx.x;            // 64 lanes, D == 64
  __shared__ float e[64], xh[64], dz[32], sh[4];
  const float* xr = p.x + (long)b * p.xstride + (long)(p.S - 1) * p.D;
  const float* th = p.theta + (long)b * p.G;
  float v = xr[lane], s = v, q = v * v;
  for (int o = 32; o; o >>= 1) { s += __shfl_xor(s, o, 64); q += __shfl_xor(q, o, 64); }
  const float m = s / 64.f, var = fmaxf(0.f, q / 64.f - m * m), r = rsqrtf(var + 1e-6f);
  xh[lane] = (v - m) * r;
  e[lane] = xh[lane] * th[p.o_ns + lane] + th[p.o_nb + lane];
  __syncthreads();
  const int A = p.Hz * (p.ad - 1), NO = A + p.Hz;
  float z = 0.f;
  if (lane < NO) {
    const float* W = lane < A ? th + p.o_wc : th + p.o_wd;
    const int n = lane < A ? lane : lane - A, ld = lane < A ? A : p.Hz;
    for (int k = 0; k < 64; ++k) z = fmaf(e[k], W[k * ld + n], z);
    z += lane < A ? th[p.o_bc + n] : th[p.o_bd + n];
  }
  // ---- loss terms
  const bool tm = p.tmask[b] != 0;
  float cs = 0.f, cm = 0.f, ds = 0.f, dm = 0.f, dzl = 0.f, mk = 0.f, pred = 0.f, tgt = 0.f;
  if (lane < NO) {
    int h, a;
    if (lane < A) { h = lane / (p.ad - 1); a = lane % (p.ad - 1); } else { h = lane - A; a = p.ad - 1; }
    const long o = ((long)b * p.Hz + h) * p.ad + a;
    mk = (tm && p.amask[o]) ? 1.f : 0.f;
    tgt = p.clip_target ? fminf(fmaxf(p.target[o], -p.max_action), p.max_action) : p.target[o];
    if (lane < A) {
      pred = tanhf(z / p.tanh_scale) * p.max_action;
      cs = (pred - tgt) * (pred - tgt) * mk; cm = mk;
      if (p.actions) p.actions[o] = pred;
    } else {
      const float sp = log1pf(expf(-fabsf(z)));
      ds = (tgt * (fmaxf(-z, 0.f) + sp) + (1.f - tgt) * (fmaxf(z, 0.f) + sp)) * mk; dm = mk;
      if (p.actions) p.actions[o] = z >= 0.f ? 1.f : 0.f;
      if (p.logits) p.logits[(long)b * p.Hz + h] = z;
    }
  }
  for (int o = 32; o; o >>= 1) {
    cs += __shfl_xor(cs, o, 64); cm += __shfl_xor(cm, o, 64); ds += __shfl_xor(ds, o, 64); dm += __shfl_xor(dm, o, 64);
  }
  const float nc = (float)A, nd = (float)p.Hz;
  const float dc = fmaxf(cm / nc, 1e-5f), dd = fmaxf(dm / nd, 1e-5f);
  if (lane == 0) p.loss[b] = (cs / nc) / dc * (float)(p.ad - 1) + (ds / nd) / dd;
  if (!p.dxrow) return;
  // ---- backward (d/dz of loss[b] / B)
  const float gb = 1.f / (float)p.B;
  if (lane < A) {
    const float t = pred / p.max_action;                                   // tanh(z / ts)
    dzl = gb * (float)(p.ad - 1) / (nc * dc) * 2.f * (pred - tgt) * mk * p.max_action * (1.f - t * t) / p.tanh_scale;
  } else if (lane < NO) {
    const float sg = 1.f / (1.f + expf(-z));
    dzl = gb / (nd * dd) * (sg - tgt) * mk;
  }
  if (lane < 32) dz[lane] = lane < NO ? dzl : 0.f;
  __syncthreads();
  float* dth = p.dtheta + (long)b * p.G;
  if (lane < NO) {                                                           // bias grads
    if (lane < A) dth[p.o_bc + lane] += dzl; else dth[p.o_bd + lane - A] += dzl;
  }
  // kernel grads dW[k][n] = e[k] dz[n]; de[k] = sum_n W[k][n] dz[n]   (lane = k)
  float de = 0.f;
  for (int n = 0; n < A; ++n) { dth[p.o_wc + lane * A + n] += e[lane] * dz[n]; de = fmaf(th[p.o_wc + lane * A + n], dz[n], de); }
  for (int n = 0; n < p.Hz; ++n) { dth[p.o_wd + lane * p.Hz + n] += e[lane] * dz[A + n]; de = fmaf(th[p.o_wd + lane * p.Hz + n], dz[A + n], de); }
  dth[p.o_ns + lane] += de * xh[lane];
  dth[p.o_nb + lane] += de;
  const float dxh = de * th[p.o_ns + lane];
  float s1 = dxh, s2 = dxh * xh[lane];
  for (int o = 32; o; o >>= 1) { s1 += __shfl_xor(s1, o, 64); s2 += __shfl_xor(s2, o, 64); }
  p.dxrow[(long)b * p.D + lane] = r * (dxh - s1 / 64.f - xh[lane] * s2 / 64.f);
}

// context-token assembly: rows t < T += pos_tok[t]; row T += pos_img; row T+1 = pos_layer (hypernetwork.py:112-145)
__global__ void ctx_rows_kernel(float* __restrict__ x, const float* __restrict__ pos_tok, const float* __restrict__ pos_img,
                                const float* __restrict__ pos_layer, int B, int T, int C) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const int Sc = T + 2;
  if (i >= (long)B * Sc * C) return;
  const int c = (int)(i % C), t = (int)((i / C) % Sc);
  if (t < T) x[i] += pos_tok[(long)t * C + c];
  else if (t == T) x[i] += pos_img[c];
  else x[i] = pos_layer[c];
}
__global__ void ctx_rows_bwd_kernel(const float* __restrict__ dx, float* __restrict__ dpos_tok, float* __restrict__ dpos_img,
                                    float* __restrict__ dpos_layer, float* __restrict__ db_tok, float* __restrict__ db_img,
                                    int B, int T, int C) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const int Sc = T + 2;
  if (i >= (long)B * Sc * C) return;
  const int c = (int)(i % C), t = (int)((i / C) % Sc);
  const float v = dx[i];
  if (t < T) { unsafeAtomicAdd(dpos_tok + (long)t * C + c, v); unsafeAtomicAdd(db_tok + c, v); }
  else if (t == T) { unsafeAtomicAdd(dpos_img + c, v); unsafeAtomicAdd(db_img + c, v); }
  else unsafeAtomicAdd(dpos_layer + c, v);
}
// ctx[b] = (LN(x_b) * scale + bias) * post   (encoder_norm on the layer-token row, hypernetwork.py:188-192)
__global__ void ctx_final_fwd_kernel(const float* __restrict__ x, long xstride, float* __restrict__ xhat,
                                     float* __restrict__ mean, float* __restrict__ rstd, const float* __restrict__ scale,
                                     const float* __restrict__ bias, float* __restrict__ ctx, int B, int C, float post) {
  const int b = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (b >= B) return;
  const float* xr = x + (long)b * xstride;
  float s = 0.f, q = 0.f;
  for (int c = lane; c < C; c += 64) { s += xr[c]; q += xr[c] * xr[c]; }
  for (int o = 32; o; o >>= 1) { s += __shfl_xor(s, o, 64); q += __shfl_xor(q, o, 64); }
  const float m = s / C, var = fmaxf(0.f, q / C - m * m), r = rsqrtf(var + 1e-6f);
  for (int c = lane; c < C; c += 64) {
    const float xh = (xr[c] - m) * r;
    xhat[(long)b * C + c] = xh;
    ctx[(long)b * C + c] = (xh * scale[c] + bias[c]) * post;
  }
  if (lane == 0) { mean[b] = m; rstd[b] = r; }
}
__global__ void ctx_final_bwd_kernel(const float* __restrict__ x, long xstride, const float* __restrict__ dctx,
                                     const float* __restrict__ mean, const float* __restrict__ rstd,
                                     const float* __restrict__ scale, float* __restrict__ dx, float* __restrict__ dscale,
                                     float* __restrict__ dbias, int B, int C, float post) {
  const int b = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (b >= B) return;
  const float* xr = x + (long)b * xstride;
  const float m = mean[b], r = rstd[b];
  float s1 = 0.f, s2 = 0.f;
  for (int c = lane; c < C; c += 64) {
    const float xh = (xr[c] - m) * r, dy = dctx[(long)b * C + c] * post, dxh = dy * scale[c];
    s1 += dxh; s2 += dxh * xh;
  }
  for (int o = 32; o; o >>= 1) { s1 += __shfl_xor(s1, o, 64); s2 += __shfl_xor(s2, o, 64); }
  for (int c = lane; c < C; c += 64) {
    const float xh = (xr[c] - m) * r, dy = dctx[(long)b * C + c] * post, dxh = dy * scale[c];
    dx[(long)b * xstride + c] += r * (dxh - s1 / C - xh * s2 / C);
    unsafeAtomicAdd(dscale + c, dy * xh);
    unsafeAtomicAdd(dbias + c, dy);
  }
}

// ---- optimizer ------------------------------------------------------------------------------------
__global__ void sqsum_kernel(const float* __restrict__ g, long n, float* __restrict__ out) {
  float s = 0.f;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) s += g[i] * g[i];
  for (int o = 32; o; o >>= 1) s += __shfl_xor(s, o, 64);
  if ((threadIdx.x & 63) == 0) unsafeAtomicAdd(out, s);
}
// MultiSteps micro-step (octo/utils/train_utils.py:420-426: chain(clip_by_global_norm, MultiSteps(adamw))): the clipped
// gradient of this micro-batch goes into the running mean, acc += clip(g) / k
__global__ void accumulate_kernel(float* __restrict__ acc, const float* __restrict__ g, long n, const float* __restrict__ sq,
                                  float clip, float inv_k) {
  const float norm = sqrtf(sq[0]);
  const float sc = (norm < clip ? 1.f : clip / norm) * inv_k;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) acc[i] = fmaf(g[i], sc, acc[i]);
}
// clip-by-global-norm -> AdamW (mu stored bf16, optax mu_dtype) -> EMA; decoupled weight decay where wd_mask[i] != 0
__global__ void adamw_kernel(float* __restrict__ p, const float* __restrict__ g, __bf16* __restrict__ mu,
                             float* __restrict__ nu, float* __restrict__ ema, long n, const float* __restrict__ sq,
                             float clip, float lr, float b1, float b2, float eps, float wd,
                             const uint8_t* __restrict__ wd_mask, float bc1, float bc2, float ema_decay) {
  const float norm = sqrtf(sq[0]);
  const float sc = norm < clip ? 1.f : clip / norm;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const float gi = g[i] * sc;
    const float m = b1 * (float)mu[i] + (1.f - b1) * gi;
    const float v = b2 * nu[i] + (1.f - b2) * gi * gi;
    mu[i] = (__bf16)m;
    nu[i] = v;
    float upd = (m / bc1) / (sqrtf(v / bc2) + eps);
    if (wd_mask && wd_mask[i]) upd += wd * p[i];     // the host builds the mask of the selected weight_decay_strategy
    const float pn = p[i] - lr * upd;
    p[i] = pn;
    if (ema) ema[i] = ema_decay * ema[i] + (1.f - ema_decay) * pn;
  }
}

// optimizer group "shared" (train_utils.py:414-419 + scripts/train.py:465-471): AdamW at base_lr, decay base_wd on the
// masked leaves (every image_encoder leaf under weight_decay_strategy v5, the *kernel* leaves under v1), and -- as the
// reference does for every shared leaf when base_wd > 0 -- the update gets + base_lr * base_wd * p0 (the pull back towards
// the pretrained weights p0).
__global__ void adamw_shared_kernel(float* __restrict__ p, const float* __restrict__ g, __bf16* __restrict__ mu,
                                    float* __restrict__ nu, float* __restrict__ ema, long n, const float* __restrict__ sq,
                                    float clip, float lr, float b1, float b2, float eps, float wd,
                                    const uint8_t* __restrict__ mask, const float* __restrict__ p0, float bc1, float bc2,
                                    float ema_decay) {
  const float norm = sqrtf(sq[0]);
  const float sc = norm < clip ? 1.f : clip / norm;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const float gi = g[i] * sc;
    const float m = b1 * (float)mu[i] + (1.f - b1) * gi;
    const float v = b2 * nu[i] + (1.f - b2) * gi * gi;
    mu[i] = (__bf16)m;
    nu[i] = v;
    float upd = (m / bc1) / (sqrtf(v / bc2) + eps);
    if (wd > 0.f) {
      if (mask && mask[i]) upd += wd * p[i];
      if (p0) upd -= wd * p0[i];
    }
    const float pn = p[i] - lr * upd;
    p[i] = pn;
    if (ema) ema[i] = ema_decay * ema[i] + (1.f - ema_decay) * pn;
  }
}

// ------------------------------------------------------------------------------------------------
// host sequencing
// ------------------------------------------------------------------------------------------------
#define KL(kernel, grid, block, ...) hipLaunchKernelGGL(kernel, grid, block, 0, st, __VA_ARGS__)
static inline dim3 g1(long n, int bs = 256) { long b = (n + bs - 1) / bs; return dim3((unsigned)(b > 65535 ? 65535 : b)); }

TrainLayout make_train_layout(const Geom& g) {
  TrainLayout L;
  long o = 0;
  auto add = [&](long& slot, long n) { slot = o; o += n; };
  const int C = g.C, F = g.ctx_mlp;
  add(L.w_tok, (long)g.lang_dim * C); add(L.b_tok, C); add(L.w_img, (long)g.E * C); add(L.b_img, C);
  add(L.pos_tok, (long)g.T * C); add(L.pos_img, C); add(L.pos_layer, C);
  for (int l = 0; l < g.ctx_layers; ++l) {
    TrainLayout::CL& c = L.layer[l];
    add(c.ln0_s, C); add(c.ln0_b, C); add(c.ln1_s, C); add(c.ln1_b, C);
    add(c.wq, (long)C * C); add(c.bq, C); add(c.wk, (long)C * C); add(c.bk, C); add(c.wv, (long)C * C); add(c.bv, C);
    add(c.wo, (long)C * C); add(c.bo, C); add(c.w1, (long)C * F); add(c.b1, F); add(c.w2, (long)F * C); add(c.b2, C);
  }
  add(L.norm_s, C); add(L.norm_b, C);
  L.G = generated_leaves(g).back().offset + generated_leaves(g).back().size;
  add(L.wcat, (long)C * L.G); add(L.bcat, L.G);
  L.total = o;
  // shared DINOv2 leaves, hypervla.config.encoder_leaves order, offsets relative to L.total
  o = 0;
  const long E = g.E, Fe = g.enc_mlp, Se = g.P() + 1;
  add(L.e_cls, E); add(L.e_mask, E); add(L.e_pb, E); add(L.e_pk, (long)g.patch * g.patch * 3 * E); add(L.e_pos, Se * E);
  for (int l = 0; l < g.enc_layers; ++l) {
    TrainLayout::EL& y = L.enc[l];
    add(y.kb, E); add(y.kk, E * E); add(y.qb, E); add(y.qk, E * E); add(y.vb, E); add(y.vk, E * E); add(y.ob, E); add(y.ok, E * E);
    add(y.ls1, E); add(y.ls2, E); add(y.f1b, Fe); add(y.f1k, E * Fe); add(y.f2b, E); add(y.f2k, Fe * E);
    add(y.n1b, E); add(y.n1s, E); add(y.n2b, E); add(y.n2s, E);
  }
  add(L.e_lnb, E); add(L.e_lns, E);
  L.enc_total = o;
  return L;
}

struct Off { int wp, bp, pos, ns, nb, wc, bc, wd, bd; struct Lyr { int l0s, l0b, l1s, l1b, w1, b1, w2, b2, wk, bk, wo, bo, wq, bq, wv, bv; } l[16]; };
static Off leaf_offsets(const Geom& g) {
  Off o{};
  auto lv = generated_leaves(g);
  auto f = [&](const std::string& n) { for (auto& l : lv) if (l.flat == n) return (int)l.offset; return -1; };
  o.bc = f("action_head_continuous_head_bias"); o.wc = f("action_head_continuous_head_kernel");
  o.bd = f("action_head_discrete_head_bias"); o.wd = f("action_head_discrete_head_kernel");
  o.nb = f("encoder_Transformer_0_encoder_norm_bias"); o.ns = f("encoder_Transformer_0_encoder_norm_scale");
  o.bp = f("encoder_image_embedding_projection_bias"); o.wp = f("encoder_image_embedding_projection_kernel");
  o.pos = f("encoder_pos_embedding");
  for (int l = 0; l < g.L; ++l) {
    const std::string B = "encoder_Transformer_0_encoderblock_" + std::to_string(l) + "_", A = B + "MultiHeadDotProductAttention_0_";
    Off::Lyr& y = o.l[l];
    y.l0b = f(B + "LayerNorm_0_bias"); y.l0s = f(B + "LayerNorm_0_scale"); y.l1b = f(B + "LayerNorm_1_bias"); y.l1s = f(B + "LayerNorm_1_scale");
    y.b1 = f(B + "MlpBlock_0_Dense_0_bias"); y.w1 = f(B + "MlpBlock_0_Dense_0_kernel");
    y.b2 = f(B + "MlpBlock_0_Dense_1_bias"); y.w2 = f(B + "MlpBlock_0_Dense_1_kernel");
    y.bk = f(A + "key_bias"); y.wk = f(A + "key_kernel"); y.bo = f(A + "out_bias"); y.wo = f(A + "out_kernel");
    y.bq = f(A + "query_bias"); y.wq = f(A + "query_kernel"); y.bv = f(A + "value_bias"); y.wv = f(A + "value_kernel");
  }
  return o;
}

// Transformer block forward / backward on rows [nb][S][D] with weights at W + b * wstride (wstride = G for
// the per-episode policy, 0 for the shared context encoder).  Buffers for one layer:
// h, h2, g (the two LayerNorm outputs and the GELU output): kept per block when the plan has room for them (the backward pass
// then reads them instead of recomputing them: two LayerNorm passes and one GELU pass per block), else nullptr
struct BlkBuf { float *x_in, *mean0, *rstd0, *q, *k, *v, *p, *o, *x_mid, *mean1, *rstd1, *u, *y1, *y2, *h, *h2, *g; };
struct BlkW { const float *l0s, *l0b, *wq, *bq, *wk, *bk, *wv, *bv, *wo, *bo, *l1s, *l1b, *w1, *b1, *w2, *b2, *ls1, *ls2; };
struct BlkG { float *l0s, *l0b, *wq, *bq, *wk, *bk, *wv, *bv, *wo, *bo, *l1s, *l1b, *w1, *b1, *w2, *b2, *ls1, *ls2; };
// block flavour: the flax Encoder1DBlock of the context encoder / generated policy (tanh GELU, masked attention) or the
// HF Dinov2Layer (erf GELU, LayerScale on both residual branches, dense attention)
struct BlkOpt { int mask_mode; const int64_t* am; int gelu_erf; };
struct BlkTmp { float *h, *g, *d, *dq, *dk, *dv, *dp, *y; };

// Y[nb][S][N] (+)= X[nb][S][K] W + bias with W per episode (ws = G) or shared (ws = 0: the batch folds into M, so
// S = 257 does not pay a mostly empty 64-row tile per sample)
static void linear(hipStream_t st, int nb, int S, long ws, const float* X, const float* W, const float* bias, float* Y, int K, int N, int acc) {
  if (ws == 0) bgemm(st, false, false, BG{X, W, Y, bias, nb * S, N, K, K, N, N, 0, 0, 0, 0, 0, 0, 0, 1, 1.f, acc}, 1);
  else bgemm(st, false, false, BG{X, W, Y, bias, S, N, K, K, N, N, (long)S * K, 0, ws, 0, (long)S * N, 0, ws, 1, 1.f, acc}, nb);
}
// dX[nb][S][K] (+)= dY[nb][S][N] W^T
static void linear_dx(hipStream_t st, int nb, int S, long ws, const float* dY, const float* W, float* dX, int K, int N, int acc) {
  if (ws == 0) bgemm(st, false, true, BG{dY, W, dX, nullptr, nb * S, K, N, N, N, K, 0, 0, 0, 0, 0, 0, 0, 1, 1.f, acc}, 1);
  else bgemm(st, false, true, BG{dY, W, dX, nullptr, S, K, N, N, N, K, (long)S * N, 0, ws, 0, (long)S * K, 0, 0, 1, 1.f, acc}, nb);
}

static void scale_add(hipStream_t st, float* out, const float* res, const float* y, const float* ls, long n, int D) {
  const bool v4 = D % 4 == 0 && ((reinterpret_cast<uintptr_t>(out) | reinterpret_cast<uintptr_t>(res) | reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(ls)) & 15) == 0;
  if (v4) KL(scale_add4_kernel, g1(n / 4), dim3(256), reinterpret_cast<f32x4*>(out), reinterpret_cast<const f32x4*>(res), reinterpret_cast<const f32x4*>(y), reinterpret_cast<const f32x4*>(ls), n / 4, D / 4);
  else KL(scale_add_kernel, g1(n), dim3(256), out, res, y, ls, n, D);
}

static void block_fwd(hipStream_t st, int nb, int S, int D, int H, int F, long ws, const BlkW& w, const BlkBuf& a,
                      float* x_out, const BlkTmp& t, const BlkOpt& op) {
  const int hd = D / H, rows = nb * S;
  float* const h = a.h ? a.h : t.h;
  float* const h2 = a.h2 ? a.h2 : t.h;
  float* const gg = a.g ? a.g : t.g;
  KL(ln_fwd_kernel, dim3((rows + 3) / 4), dim3(256), a.x_in, h, a.mean0, a.rstd0, w.l0s, w.l0b, ws, rows, S, D);
  {
    // q, k, v = h Wq, h Wk, h Wv as ONE batched product when weights, biases and outputs are equally spaced in memory order
    // (shared weights): 3 x 390 tiles in one launch instead of three launches of 198 double tiles that fill 77 % of the chip
    const float* Wm[3] = {w.wq, w.wk, w.wv};
    const float* Bm[3] = {w.bq, w.bk, w.bv};
    float* Ym[3] = {a.q, a.k, a.v};
    int ord[3] = {0, 1, 2};
    for (int i = 0; i < 2; ++i)
      for (int j = 0; j < 2 - i; ++j)
        if (Wm[ord[j]] > Wm[ord[j + 1]]) { const int tsw = ord[j]; ord[j] = ord[j + 1]; ord[j + 1] = tsw; }
    const long wsd = Wm[ord[1]] - Wm[ord[0]], bsd = Bm[ord[1]] - Bm[ord[0]], ysd = Ym[ord[1]] - Ym[ord[0]];
    const bool one = ws == 0 && Wm[ord[2]] - Wm[ord[1]] == wsd && Bm[ord[2]] - Bm[ord[1]] == bsd && Ym[ord[2]] - Ym[ord[1]] == ysd &&
                     wsd % 4 == 0 && ysd % 4 == 0;
    if (one) {
      BG g{h, Wm[ord[0]], Ym[ord[0]], Bm[ord[0]], nb * S, D, D, D, D, D, 0, 0, 0, wsd, 0, ysd, 0, 3, 1.f, 0};
      g.sBias1 = bsd;
      bgemm(st, false, false, g, 1);
    } else {
      for (int i = 0; i < 3; ++i) linear(st, nb, S, ws, h, Wm[i], Bm[i], Ym[i], D, D, 0);
    }
  }
  // scores[b][h] = q_h k_h^T / sqrt(hd)
  // the S x S attention matrices have row stride Sp = S rounded up to 4 floats, zeros behind every row (written by the softmax
  // kernels): as A operands they are staged with float4 loads like everything else (S = 257: the all-dword staging these four
  // products per layer were left with cost 3.6 ms per step)
  const int Sp = (S + 3) & ~3;
  bgemm(st, false, true, BG{a.q, a.k, a.p, nullptr, S, S, hd, D, D, Sp, (long)S * D, hd, (long)S * D, hd, (long)H * S * Sp, (long)S * Sp, 0, H, 1.f / sqrtf((float)hd), 0}, nb);
  KL(softmax_fwd_kernel, dim3((nb * H * S + 3) / 4), dim3(256), a.p, nb * H, S, S, op.mask_mode, op.am, H, Sp);
  {
    BG pv{a.p, a.v, a.o, nullptr, S, hd, S, Sp, D, D, (long)H * S * Sp, (long)S * Sp, (long)S * D, hd, (long)S * D, hd, 0, H, 1.f, 0};
    pv.a_padded = 1;
    bgemm(st, false, false, pv, nb);
  }
  if (w.ls1) {
    linear(st, nb, S, ws, a.o, w.wo, w.bo, a.y1, D, D, 0);
    scale_add(st, a.x_mid, a.x_in, a.y1, w.ls1, (long)rows * D, D);
  } else {
    (void)hipMemcpyAsync(a.x_mid, a.x_in, (size_t)rows * D * 4, hipMemcpyDeviceToDevice, st);
    linear(st, nb, S, ws, a.o, w.wo, w.bo, a.x_mid, D, D, 1);
  }
  KL(ln_fwd_kernel, dim3((rows + 3) / 4), dim3(256), a.x_mid, h2, a.mean1, a.rstd1, w.l1s, w.l1b, ws, rows, S, D);
  linear(st, nb, S, ws, h2, w.w1, w.b1, a.u, D, F, 0);
  KL(gelu_fwd_kernel, g1((long)rows * F), dim3(256), a.u, gg, (long)rows * F, op.gelu_erf);
  if (w.ls2) {
    linear(st, nb, S, ws, gg, w.w2, w.b2, a.y2, F, D, 0);
    scale_add(st, x_out, a.x_mid, a.y2, w.ls2, (long)rows * D, D);
  } else {
    (void)hipMemcpyAsync(x_out, a.x_mid, (size_t)rows * D * 4, hipMemcpyDeviceToDevice, st);
    linear(st, nb, S, ws, gg, w.w2, w.b2, x_out, F, D, 1);
  }
}

// dx (in/out: gradient wrt the block output on entry, wrt its input on exit).  Weight gradients: per-episode
// (gs = G) written per b; shared (gs = 0) reduced over the batch by folding it into the GEMM's M/K dimension.
static void block_bwd(hipStream_t st, int nb, int S, int D, int H, int F, long ws, long gs, const BlkW& w, const BlkG& gw,
                      const BlkBuf& a, float* dx, const BlkTmp& t, const BlkOpt& op) {
  const int hd = D / H, rows = nb * S;
  const bool shared = gs == 0;
  auto wgrad = [&](const float* X, int K, const float* dY, int N, float* dW) {   // dW[K][N] (+)= X^T dY
    if (shared) bgemm(st, true, false, BG{X, dY, dW, nullptr, K, N, rows, K, N, N, 0, 0, 0, 0, 0, 0, 0, 1, 1.f, 1, 1, 1}, 1);
    else bgemm(st, true, false, BG{X, dY, dW, nullptr, K, N, S, K, N, N, (long)S * K, 0, (long)S * N, 0, gs, 0, 0, 1, 1.f, 1}, nb);
  };
  auto bgrad = [&](const float* dY, int N, float* dB) {
    // the 4-column form wins on the narrow matrices of the policy / hypernetwork; on the encoder's 768- and 3072-wide ones
    // its fewer, fatter waves lose to the one-column form (17 vs 23 us)
    const bool v4 = N % 4 == 0 && N <= 512 && (reinterpret_cast<uintptr_t>(dY) & 15) == 0;
    if (shared && N % 4 == 0 && N > 512 && rows >= 2048 && (reinterpret_cast<uintptr_t>(dY) & 15) == 0)
      KL(colsum4b_kernel, dim3((N / 4 + 63) / 64, 1, (rows + CSB_WAVES * 8 * CSB_BATCH - 1) / (CSB_WAVES * 8 * CSB_BATCH)), dim3(CSB_WAVES * 64), dY, dB, rows, N);
    else if (shared && v4) KL(colsum4_kernel, dim3((N / 4 + 63) / 64, 1, (rows + 31) / 32), dim3(64), dY, dB, 0, rows, rows, N, 1);
    else if (shared) KL(colsum_kernel, dim3((N + 63) / 64, 1, (rows + 63) / 64), dim3(64), dY, dB, 0, rows, rows, N, 1);
    else if (v4) KL(colsum4_kernel, dim3((N / 4 + 63) / 64, nb, (S + 31) / 32), dim3(64), dY, dB, gs, S, S, N, nb);
    else KL(colsum_kernel, dim3((N + 63) / 64, nb, (S + 63) / 64), dim3(64), dY, dB, gs, S, S, N, nb);
  };
  auto ln_bwd = [&](const float* x, const float* dy, const float* mean, const float* rstd, const float* sc, float* gs_, float* gb_) {
    if (shared && D % 4 == 0 && D <= 1024 && ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(dy) | reinterpret_cast<uintptr_t>(dx)) & 15) == 0) {
      KL(ln_bwd_shared_kernel, dim3((rows + 63) / 64), dim3(LNB_WAVES * 64), x, dy, mean, rstd, sc, dx, gs_, gb_, rows, D, 1);
    } else if (shared) {
      KL(ln_bwd_kernel, dim3((rows + 3) / 4), dim3(256), x, dy, mean, rstd, sc, dx, (float*)nullptr, (float*)nullptr, 0, rows, S, D, 1);
      KL(ln_pgrad_kernel, dim3((D + 63) / 64, (rows + 63) / 64), dim3(64), x, dy, mean, rstd, gs_, gb_, rows, D);
    } else {
      KL(ln_bwd_kernel, dim3((rows + 3) / 4), dim3(256), x, dy, mean, rstd, sc, dx, gs_, gb_, gs, rows, S, D, 1);
    }
  };
  auto ls_bwd = [&](const float* dxp, const float* y, const float* ls, float* dy, float* dls) {
    if (D % 4 == 0 && rows >= 1024 && ((reinterpret_cast<uintptr_t>(dxp) | reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(dy)) & 15) == 0)
      KL(ls_bwd4_kernel, dim3((D / 4 + 63) / 64, (rows + 127) / 128), dim3(512), dxp, y, ls, dy, dls, rows, D);
    else
      KL(ls_bwd_kernel, dim3((D + 63) / 64, (rows + 63) / 64), dim3(64), dxp, y, ls, dy, dls, rows, D);
  };
  // ---- MLP: x_out = x_mid + [ls2 (.)] (gelu(LN1(x_mid) W1 + b1) W2 + b2)
  const float* h2 = a.h2 ? a.h2 : t.h;
  const float* gg = a.g ? a.g : t.g;
  if (!a.h2) KL(ln_fwd_kernel, dim3((rows + 3) / 4), dim3(256), a.x_mid, t.h, a.mean1, a.rstd1, w.l1s, w.l1b, ws, rows, S, D);   // recompute h2
  if (!a.g) KL(gelu_fwd_kernel, g1((long)rows * F), dim3(256), a.u, t.g, (long)rows * F, op.gelu_erf);                           // recompute g
  const float* dy = dx;
  if (w.ls2) {
    ls_bwd(dx, a.y2, w.ls2, t.y, gw.ls2);
    dy = t.y;
  }
  wgrad(gg, F, dy, D, gw.w2);
  bgrad(dy, D, gw.b2);
  linear_dx(st, nb, S, ws, dy, w.w2, t.d, F, D, 0);                                                                     // dg = dy W2^T
  KL(gelu_bwd_kernel, g1((long)rows * F), dim3(256), a.u, t.d, (long)rows * F, op.gelu_erf);                           // du
  wgrad(h2, D, t.d, F, gw.w1);
  bgrad(t.d, F, gw.b1);
  linear_dx(st, nb, S, ws, t.d, w.w1, t.g, D, F, 0);                                                                    // dh2 -> t.g[rows][D]
  ln_bwd(a.x_mid, t.g, a.mean1, a.rstd1, w.l1s, gw.l1s, gw.l1b);
  //   (dx now = gradient wrt x_mid)
  // ---- attention: x_mid = x_in + [ls1 (.)] (o Wo + bo)
  dy = dx;
  if (w.ls1) {
    ls_bwd(dx, a.y1, w.ls1, t.y, gw.ls1);
    dy = t.y;
  }
  wgrad(a.o, D, dy, D, gw.wo);
  bgrad(dy, D, gw.bo);
  linear_dx(st, nb, S, ws, dy, w.wo, t.d, D, D, 0);                                                                     // do
  // dp = do_h v_h^T ; dv_h = p^T do_h
  const int Sp = (S + 3) & ~3;                                                                                          // row stride of p / dp (block_fwd)
  const long ss0 = (long)H * S * Sp, ss1 = (long)S * Sp;
  auto padded = [](BG g) { g.a_padded = 1; return g; };
  bgemm(st, false, true, BG{t.d, a.v, t.dp, nullptr, S, S, hd, D, D, Sp, (long)S * D, hd, (long)S * D, hd, ss0, ss1, 0, H, 1.f, 0}, nb);
  bgemm(st, true, false, padded(BG{a.p, t.d, t.dv, nullptr, S, hd, S, Sp, D, D, ss0, ss1, (long)S * D, hd, (long)S * D, hd, 0, H, 1.f, 0}), nb);
  KL(softmax_bwd_kernel, dim3((nb * H * S + 3) / 4), dim3(256), a.p, t.dp, nb * H * S, S, Sp);                             // ds
  const float sc = 1.f / sqrtf((float)hd);
  bgemm(st, false, false, padded(BG{t.dp, a.k, t.dq, nullptr, S, hd, S, Sp, D, D, ss0, ss1, (long)S * D, hd, (long)S * D, hd, 0, H, sc, 0}), nb);   // dq = ds k / sqrt(hd)
  bgemm(st, true, false, padded(BG{t.dp, a.q, t.dk, nullptr, S, hd, S, Sp, D, D, ss0, ss1, (long)S * D, hd, (long)S * D, hd, 0, H, sc, 0}), nb);    // dk = ds^T q / sqrt(hd)
  const float* h = a.h ? a.h : t.h;
  if (!a.h) KL(ln_fwd_kernel, dim3((rows + 3) / 4), dim3(256), a.x_in, t.h, a.mean0, a.rstd0, w.l0s, w.l0b, ws, rows, S, D);      // recompute h
  const float* dqkv[3] = {t.dq, t.dk, t.dv};
  float* gWm[3] = {gw.wq, gw.wk, gw.wv};
  float* gBm[3] = {gw.bq, gw.bk, gw.bv};
  const float* Wm[3] = {w.wq, w.wk, w.wv};
  // the three weight gradients h^T dq, h^T dk, h^T dv as ONE batched product when the three dY buffers and the three gradient
  // leaves are equally spaced (shared weights; any order): one launch whose split-K needs a third of the atomic adds per element
  int ord[3] = {0, 1, 2};                                    // q, k, v in the order of their gradient leaves in memory
  for (int i = 0; i < 2; ++i)
    for (int j = 0; j < 2 - i; ++j)
      if (gWm[ord[j]] > gWm[ord[j + 1]]) { const int tsw = ord[j]; ord[j] = ord[j + 1]; ord[j + 1] = tsw; }
  const long dys = dqkv[ord[1]] - dqkv[ord[0]], gws = gWm[ord[1]] - gWm[ord[0]];
  const bool one = shared && dqkv[ord[2]] - dqkv[ord[1]] == dys && gWm[ord[2]] - gWm[ord[1]] == gws && dys % 4 == 0;
  if (one) bgemm(st, true, false, BG{h, dqkv[ord[0]], gWm[ord[0]], nullptr, D, D, rows, D, D, D, 0, 0, 0, dys, 0, gws, 0, 3, 1.f, 1, 1, 1}, 1);
  for (int i = 0; i < 3; ++i) {
    if (!one) wgrad(h, D, dqkv[i], D, gWm[i]);
    bgrad(dqkv[i], D, gBm[i]);
    linear_dx(st, nb, S, ws, dqkv[i], Wm[i], t.g, D, D, i > 0);                                                         // dh
  }
  ln_bwd(a.x_in, t.g, a.mean0, a.rstd0, w.l0s, gw.l0s, gw.l0b);
}

// ---- workspace plan (shared by the size query and the step so the two cannot drift) ----------------------------
struct Plan {
  std::vector<BlkBuf> cb, pb, eb;
  float *cx_fin, *cdx, *px_fin, *pdx, *ex_fin, *edx, *eh, *patches;
  BlkTmp t;
  float *cmean, *crstd, *ctx, *dctx, *ctxn, *dxrow, *emean, *erstd;
  long used;
};
static Plan make_plan(const Geom& g, int B, bool enc, float* base) {
  Plan pl;
  float* ws = base;
  auto take = [&](long n) { float* p = ws; ws += (n + 3) / 4 * 4; return p; };
  auto take_blk = [&](long nb, long s, long d, long h, long f, bool ls) {
    BlkBuf b; b.x_in = take(nb * s * d); b.mean0 = take(nb * s); b.rstd0 = take(nb * s); b.k = take(nb * s * d); b.q = take(nb * s * d);      // k, q, v: the order of the weight leaves (block_fwd batches the three)
    b.v = take(nb * s * d); b.p = take(nb * h * s * ((s + 3) & ~3L)); b.o = take(nb * s * d); b.x_mid = take(nb * s * d); b.mean1 = take(nb * s);
    b.rstd1 = take(nb * s); b.u = take(nb * s * f);
    b.y1 = ls ? take(nb * s * d) : nullptr; b.y2 = ls ? take(nb * s * d) : nullptr;
    // 288 GB of HBM: the LayerNorm and GELU outputs are kept (B = 32 with the encoder trained: 1.8 GB) instead of recomputed
    b.h = take(nb * s * d); b.h2 = take(nb * s * d); b.g = take(nb * s * f);
    return b;
  };
  const long S = g.S(), D = g.D, H = g.H, F = g.M, Sc = g.T + 2, C = g.C, Hc = g.ctx_heads, Fc = g.ctx_mlp;
  const long Se = g.P() + 1, E = g.E, He = g.enc_heads, Fe = g.enc_mlp;
  pl.cb.resize(g.ctx_layers); pl.pb.resize(g.L); pl.eb.resize(enc ? g.enc_layers : 0);
  for (auto& b : pl.cb) b = take_blk(B, Sc, C, Hc, Fc, false);
  for (auto& b : pl.pb) b = take_blk(B, S, D, H, F, false);
  for (auto& b : pl.eb) b = take_blk(B, Se, E, He, Fe, true);
  pl.cx_fin = take(B * Sc * C); pl.cdx = take(B * Sc * C);
  pl.px_fin = take(B * S * D); pl.pdx = take(B * S * D);
  pl.ex_fin = pl.edx = pl.eh = pl.patches = pl.emean = pl.erstd = nullptr;
  if (enc) {
    pl.ex_fin = take(B * Se * E); pl.edx = take(B * Se * E); pl.eh = take(B * Se * E);
    pl.patches = take((long)B * g.P() * g.patch * g.patch * 3);
    pl.emean = take(B * Se); pl.erstd = take(B * Se);
  }
  // temporaries are used by one transformer at a time: size them for the largest
  auto mx = [&](long a, long b, long c) { c = enc ? c : 0; return a > b ? (a > c ? a : c) : (b > c ? b : c); };
  const long rd = mx(B * S * D, B * Sc * C, B * Se * E), rf = mx(B * S * F, B * Sc * Fc, B * Se * Fe), hss = mx(B * H * S * ((S + 3) & ~3L), B * Hc * Sc * ((Sc + 3) & ~3L), B * He * Se * ((Se + 3) & ~3L));
  pl.t.h = take(rd); pl.t.g = take(rf); pl.t.d = take(rf); pl.t.dk = take(rd); pl.t.dq = take(rd); pl.t.dv = take(rd);      /* k, q, v: the order of the leaves (block_bwd batches the three) */ pl.t.dp = take(hss);
  pl.t.y = take(rd);
  pl.cmean = take(B); pl.crstd = take(B); pl.ctx = take(B * C); pl.dctx = take(B * C); pl.ctxn = take(B * C);
  pl.dxrow = take(B * D);
  pl.used = ws - base;
  return pl;
}
size_t train_workspace_floats(const Geom& g, int B, bool train_encoder) {
  return (size_t)make_plan(g, B, train_encoder, nullptr).used + 64;
}

hipError_t train_step(const Geom& g, const TrainLayout& L, const TrainBuffers& tb, const TrainInputs& in, int B,
                      const TrainHyper& hp, hipStream_t st, hipEvent_t* bucket_done) {
  const int S = g.S(), P = g.P(), D = g.D, H = g.H, F = g.M, E = g.E;
  const int Sc = g.T + 2, C = g.C, Hc = g.ctx_heads, Fc = g.ctx_mlp, T = g.T;
  const int Se = P + 1, He = g.enc_heads, Fe = g.enc_mlp, Kp = g.patch * g.patch * 3;
  const bool enc = in.images != nullptr;
  const long G = L.G;
  const Off off = leaf_offsets(g);
  const Plan pl = make_plan(g, B, enc, tb.work);
  const std::vector<BlkBuf>&cb = pl.cb, &pb = pl.pb, &eb = pl.eb;
  float *cx_fin = pl.cx_fin, *cdx = pl.cdx, *px_fin = pl.px_fin, *pdx = pl.pdx;
  float *cmean = pl.cmean, *crstd = pl.crstd, *ctx = pl.ctx, *dctx = pl.dctx, *ctxn = pl.ctxn, *dxrow = pl.dxrow;
  const BlkTmp& t = pl.t;
  const float* Pm = tb.params;
  float* Gm = tb.grads;
  const float* Pe = Pm + L.total;          // shared DINOv2 leaves (only touched when enc)
  float* Ge = Gm + L.total;
  (void)hipMemsetAsync(Gm, 0, (size_t)(L.total + (enc ? L.enc_total : 0)) * 4, st);
  (void)hipMemsetAsync(tb.dtheta, 0, (size_t)B * G * 4, st);
  const BlkOpt ctx_opt{1, in.attn_mask, 0}, pol_opt{0, nullptr, 0}, enc_opt{2, nullptr, 1};

  // =============================== DINOv2 forward in f32, activations kept (HF Dinov2Model; base_vit.py:111-131) =====
  auto ew = [&](int l, const float* b) { const TrainLayout::EL& y = L.enc[l]; return BlkW{b + y.n1s, b + y.n1b, b + y.qk, b + y.qb, b + y.kk, b + y.kb, b + y.vk, b + y.vb, b + y.ok, b + y.ob, b + y.n2s, b + y.n2b, b + y.f1k, b + y.f1b, b + y.f2k, b + y.f2b, b + y.ls1, b + y.ls2}; };
  auto eg = [&](int l, float* b) { const TrainLayout::EL& y = L.enc[l]; return BlkG{b + y.n1s, b + y.n1b, b + y.qk, b + y.qb, b + y.kk, b + y.kb, b + y.vk, b + y.vb, b + y.ok, b + y.ob, b + y.n2s, b + y.n2b, b + y.f1k, b + y.f1b, b + y.f2k, b + y.f2b, b + y.ls1, b + y.ls2}; };
  const float* tokens = in.tokens;         // [B][P][E] rows, episode stride tok_stride
  long tok_stride = (long)P * E;
  if (enc) {
    float* ex0 = eb.empty() ? pl.ex_fin : eb[0].x_in;
    KL(im2col_f32_kernel, g1((long)B * P * Kp), dim3(256), in.images, pl.patches, B, g.image_size, g.patch);
    bgemm(st, false, false, BG{pl.patches, Pe + L.e_pk, ex0 + E, Pe + L.e_pb, P, E, Kp, Kp, E, E, (long)P * Kp, 0, 0, 0, (long)Se * E, 0, 0, 1, 1.f, 0}, B);
    KL(enc_x0_kernel, g1((long)B * Se * E), dim3(256), ex0, Pe + L.e_cls, Pe + L.e_pos, B, Se, E);
    for (int l = 0; l < g.enc_layers; ++l)
      block_fwd(st, B, Se, E, He, Fe, 0, ew(l, Pe), eb[l], l + 1 < g.enc_layers ? eb[l + 1].x_in : pl.ex_fin, t, enc_opt);
    KL(ln_fwd_kernel, dim3((B * Se + 3) / 4), dim3(256), pl.ex_fin, pl.eh, pl.emean, pl.erstd, Pe + L.e_lns, Pe + L.e_lnb, 0, B * Se, Se, E);
    tokens = pl.eh + E;                    // drop the CLS row through the pointer: rows 1.. of every [Se][E] block
    tok_stride = (long)Se * E;
  }

  // =============================== context encoder forward (hypernetwork.py:99-197) ===============================
  float* cx0 = cb.empty() ? cx_fin : cb[0].x_in;
  // tokens: [B][T][lang] . Wt + bt + pos_t  (rows 0..T-1 of each episode's [Sc][C] block)
  bgemm(st, false, false, BG{in.tok, Pm + L.w_tok, cx0, Pm + L.b_tok, T, C, g.lang_dim, g.lang_dim, C, C, (long)T * g.lang_dim, 0, 0, 0, (long)Sc * C, 0, 0, 1, 1.f, 0}, B);
  bgemm(st, false, false, BG{in.cls, Pm + L.w_img, cx0 + (long)T * C, Pm + L.b_img, 1, C, E, E, C, C, (long)E, 0, 0, 0, (long)Sc * C, 0, 0, 1, 1.f, 0}, B);
  KL(ctx_rows_kernel, g1((long)B * Sc * C), dim3(256), cx0, Pm + L.pos_tok, Pm + L.pos_img, Pm + L.pos_layer, B, T, C);
  auto cw = [&](int l) { const TrainLayout::CL& c = L.layer[l]; return BlkW{Pm + c.ln0_s, Pm + c.ln0_b, Pm + c.wq, Pm + c.bq, Pm + c.wk, Pm + c.bk, Pm + c.wv, Pm + c.bv, Pm + c.wo, Pm + c.bo, Pm + c.ln1_s, Pm + c.ln1_b, Pm + c.w1, Pm + c.b1, Pm + c.w2, Pm + c.b2, nullptr, nullptr}; };
  auto cg = [&](int l) { const TrainLayout::CL& c = L.layer[l]; return BlkG{Gm + c.ln0_s, Gm + c.ln0_b, Gm + c.wq, Gm + c.bq, Gm + c.wk, Gm + c.bk, Gm + c.wv, Gm + c.bv, Gm + c.wo, Gm + c.bo, Gm + c.ln1_s, Gm + c.ln1_b, Gm + c.w1, Gm + c.b1, Gm + c.w2, Gm + c.b2, nullptr, nullptr}; };
  for (int l = 0; l < g.ctx_layers; ++l)
    block_fwd(st, B, Sc, C, Hc, Fc, 0, cw(l), cb[l], l + 1 < g.ctx_layers ? cb[l + 1].x_in : cx_fin, t, ctx_opt);
  // ctx = LN_f(x[:, -1]) (/ sqrt(C)); rows gathered through a stride: x = cx_fin + (Sc-1)*C, row stride Sc*C
  KL(ctx_final_fwd_kernel, dim3((B + 3) / 4), dim3(256), cx_fin + (long)(Sc - 1) * C, (long)Sc * C, ctxn, cmean, crstd, Pm + L.norm_s, Pm + L.norm_b, ctx, B, C, g.scale_context ? 1.f / sqrtf((float)C) : 1.f);
  // =============================== theta = ctx W_cat + b_cat (hypernetwork.py:205-233) ===============================
  bgemm(st, false, false, BG{ctx, Pm + L.wcat, tb.theta, Pm + L.bcat, B, (int)G, C, C, (int)G, (int)G, 0, 0, 0, 0, 0, 0, 0, 1, 1.f, 0}, 1);
  // =============================== policy forward with per-episode weights ===============================
  const float* TH = tb.theta;
  float* px0 = pb[0].x_in;
  bgemm(st, false, false, BG{tokens, TH + off.wp, px0, TH + off.bp, P, D, E, E, D, D, tok_stride, 0, G, 0, (long)S * D, 0, G, 1, 1.f, 0}, B);
  KL(x0_finish_kernel, g1((long)B * S * D), dim3(256), px0, TH + off.pos, G, S, D, B);
  auto pw = [&](int l, const float* b) { const Off::Lyr& y = off.l[l]; return BlkW{b + y.l0s, b + y.l0b, b + y.wq, b + y.bq, b + y.wk, b + y.bk, b + y.wv, b + y.bv, b + y.wo, b + y.bo, b + y.l1s, b + y.l1b, b + y.w1, b + y.b1, b + y.w2, b + y.b2, nullptr, nullptr}; };
  auto pg = [&](int l, float* b) { const Off::Lyr& y = off.l[l]; return BlkG{b + y.l0s, b + y.l0b, b + y.wq, b + y.bq, b + y.wk, b + y.bk, b + y.wv, b + y.bv, b + y.wo, b + y.bo, b + y.l1s, b + y.l1b, b + y.w1, b + y.b1, b + y.w2, b + y.b2, nullptr, nullptr}; };
  for (int l = 0; l < g.L; ++l)
    block_fwd(st, B, S, D, H, F, G, pw(l, TH), pb[l], l + 1 < g.L ? pb[l + 1].x_in : px_fin, t, pol_opt);
  // =============================== head + loss (+ backward seed) ===============================
  (void)hipMemsetAsync(pdx, 0, (size_t)B * S * D * 4, st);
  HeadP hpp{px_fin, (long)S * D, TH, tb.dtheta, G, off.wc, off.bc, off.wd, off.bd, off.ns, off.nb, in.target, in.tmask, in.amask,
            tb.loss, hp.forward_only ? nullptr : dxrow, tb.actions, tb.logits, B, S, D, g.horizon, g.action_dim, g.tanh_scale, g.max_action, g.clip_target};
  KL(head_loss_kernel, dim3(B), dim3(64), hpp);
  if (hp.forward_only) return hipGetLastError();
  KL(add_strided_kernel, g1((long)B * D), dim3(256), pdx + (long)(S - 1) * D, (long)S * D, dxrow, (long)D, B);
  // =============================== policy backward ===============================
  for (int l = g.L - 1; l >= 0; --l)
    block_bwd(st, B, S, D, H, F, G, G, pw(l, TH), pg(l, tb.dtheta), pb[l], pdx, t, pol_opt);
  // x0 = [tokens Wp + bp ; 0] + pos : dpos = dx0, dbp = sum_{t<P} dx0, dWp = tokens^T dx0[:P]
  KL(add_strided_kernel, g1((long)B * S * D), dim3(256), tb.dtheta + off.pos, G, pdx, (long)S * D, B);
  KL(colsum_kernel, dim3((D + 63) / 64, B, (P + 63) / 64), dim3(64), pdx, tb.dtheta + off.bp, G, S, P, D, B);
  bgemm(st, true, false, BG{tokens, pdx, tb.dtheta + off.wp, nullptr, E, D, P, E, D, D, tok_stride, 0, (long)S * D, 0, G, 0, 0, 1, 1.f, 1}, B);
  // =============================== DINOv2 backward ===============================
  if (enc) {
    // d tokens_b = dx0_b[:P] Wp_b^T -> rows 1.. of the final-LayerNorm output gradient (CLS row gets none)
    float* dh = t.y;                                                      // [B][Se][E]; t.y is free between blocks
    (void)hipMemsetAsync(dh, 0, (size_t)B * Se * E * 4, st);
    bgemm(st, false, true, BG{pdx, TH + off.wp, dh + E, nullptr, P, E, D, D, D, E, (long)S * D, 0, G, 0, (long)Se * E, 0, 0, 1, 1.f, 0}, B);
    float* edx = pl.edx;
    KL(ln_bwd_kernel, dim3((B * Se + 3) / 4), dim3(256), pl.ex_fin, dh, pl.emean, pl.erstd, Pe + L.e_lns, edx, (float*)nullptr, (float*)nullptr, 0, B * Se, Se, E, 0);
    KL(ln_pgrad_kernel, dim3((E + 63) / 64, (B * Se + 63) / 64), dim3(64), pl.ex_fin, dh, pl.emean, pl.erstd, Ge + L.e_lns, Ge + L.e_lnb, B * Se, E);
    for (int l = g.enc_layers - 1; l >= 0; --l)
      block_bwd(st, B, Se, E, He, Fe, 0, 0, ew(l, Pe), eg(l, Ge), eb[l], edx, t, enc_opt);
    // x0[b][0] = cls + pos[0]; x0[b][1+t] = patch_t Wk + bk + pos[1+t]
    KL(colsum_kernel, dim3((Se * E + 63) / 64, 1, (B + 63) / 64), dim3(64), edx, Ge + L.e_pos, 0, B, B, Se * E, 1);       // dpos = sum_b dx0
    (void)hipMemcpyAsync(Ge + L.e_cls, Ge + L.e_pos, (size_t)E * 4, hipMemcpyDeviceToDevice, st);                          // dcls = dpos[0]
    KL(colsum_kernel, dim3((E + 63) / 64, 1, (P + 63) / 64), dim3(64), Ge + L.e_pos + E, Ge + L.e_pb, 0, P, P, E, 1);     // dbias = sum_{t>=1} dpos[t]
    bgemm(st, true, false, BG{pl.patches, edx + E, Ge + L.e_pk, nullptr, Kp, E, P, Kp, E, E, (long)P * Kp, 0, (long)Se * E, 0, 0, 0, 0, 1, 1.f, 2}, B);
    if (bucket_done) (void)hipEventRecord(bucket_done[0], st);           // the shared DINOv2 leaves are final
  }
  // =============================== weight generation backward ===============================
  bgemm(st, true, false, BG{ctx, tb.dtheta, Gm + L.wcat, nullptr, C, (int)G, B, C, (int)G, (int)G, 0, 0, 0, 0, 0, 0, 0, 1, 1.f, 1}, 1);   // dW_cat = ctx^T dtheta
  KL(colsum_kernel, dim3((unsigned)((G + 63) / 64), 1, (B + 63) / 64), dim3(64), tb.dtheta, Gm + L.bcat, 0, B, B, (int)G, 1);                  // db_cat
  (void)hipMemsetAsync(dctx, 0, (size_t)B * C * 4, st);
  // dctx = dtheta W_cat^T: [B, G] x [G, C], a K = 201 500 product onto a B x C output -> split-K over the whole chip
  bgemm(st, false, true, BG{tb.dtheta, Pm + L.wcat, dctx, nullptr, B, C, (int)G, (int)G, (int)G, C, 0, 0, 0, 0, 0, 0, 0, 1, 1.f, 1, 1, 1}, 1);
  if (bucket_done) (void)hipEventRecord(bucket_done[1], st);             // W_cat, b_cat are final
  // =============================== context encoder backward ===============================
  (void)hipMemsetAsync(cdx, 0, (size_t)B * Sc * C * 4, st);
  KL(ctx_final_bwd_kernel, dim3((B + 3) / 4), dim3(256), cx_fin + (long)(Sc - 1) * C, (long)Sc * C, dctx, cmean, crstd, Pm + L.norm_s, cdx + (long)(Sc - 1) * C, Gm + L.norm_s, Gm + L.norm_b, B, C, g.scale_context ? 1.f / sqrtf((float)C) : 1.f);
  for (int l = g.ctx_layers - 1; l >= 0; --l)
    block_bwd(st, B, Sc, C, Hc, Fc, 0, 0, cw(l), cg(l), cb[l], cdx, t, ctx_opt);
  // inputs: tokens rows -> w_tok, b_tok, pos_tok ; image row -> w_img, b_img, pos_img ; layer row -> pos_layer
  KL(ctx_rows_bwd_kernel, g1((long)B * Sc * C), dim3(256), cdx, Gm + L.pos_tok, Gm + L.pos_img, Gm + L.pos_layer, Gm + L.b_tok, Gm + L.b_img, B, T, C);
  bgemm(st, true, false, BG{in.tok, cdx, Gm + L.w_tok, nullptr, g.lang_dim, C, T, g.lang_dim, C, C, (long)T * g.lang_dim, 0, (long)Sc * C, 0, 0, 0, 0, 1, 1.f, 2}, B);
  bgemm(st, true, false, BG{in.cls, cdx + (long)T * C, Gm + L.w_img, nullptr, E, C, 1, E, C, C, (long)E, 0, (long)Sc * C, 0, 0, 0, 0, 1, 1.f, 2}, B);
  if (bucket_done) (void)hipEventRecord(bucket_done[2], st);             // the context encoder's leaves: everything is final
  return hipGetLastError();
}

hipError_t train_apply(const TrainLayout& L, const TrainBuffers& tb, const TrainHyper& hp, bool train_encoder, hipStream_t st) {
  const long n = L.total + (train_encoder ? L.enc_total : 0);
  (void)hipMemsetAsync(tb.sqsum, 0, 4, st);
  KL(sqsum_kernel, dim3(1024), dim3(256), tb.grads, n, tb.sqsum);          // one global norm over both optimizer groups
  const float t = (float)(hp.step + 1);
  const float bc1 = 1.f - powf(hp.b1, t), bc2 = 1.f - powf(hp.b2, t);
  KL(adamw_kernel, dim3(2048), dim3(256), tb.params, tb.grads, tb.mu, tb.nu, hp.ema_decay > 0.f ? tb.ema : nullptr, L.total, tb.sqsum,
     hp.clip, hp.lr, hp.b1, hp.b2, hp.eps, hp.weight_decay, tb.wd_mask, bc1, bc2, hp.ema_decay);
  if (train_encoder)
    KL(adamw_shared_kernel, dim3(2048), dim3(256), tb.params + L.total, tb.grads + L.total, tb.mu + L.total, tb.nu + L.total,
       hp.ema_decay > 0.f ? tb.ema + L.total : nullptr, L.enc_total, tb.sqsum, hp.clip, hp.base_lr, hp.b1, hp.b2, hp.eps,
       hp.base_weight_decay, tb.wd_mask ? tb.wd_mask + L.total : nullptr, tb.params0, bc1, bc2, hp.ema_decay);
  return hipGetLastError();
}

hipError_t train_accumulate(const TrainLayout& L, const TrainBuffers& tb, float* acc, float inv_k, const TrainHyper& hp,
                            bool train_encoder, hipStream_t st) {
  const long n = L.total + (train_encoder ? L.enc_total : 0);
  (void)hipMemsetAsync(tb.sqsum, 0, 4, st);
  KL(sqsum_kernel, dim3(1024), dim3(256), tb.grads, n, tb.sqsum);
  KL(accumulate_kernel, dim3(2048), dim3(256), acc, tb.grads, n, tb.sqsum, hp.clip, inv_k);
  return hipGetLastError();
}

}  // namespace hvla
