// encoder.hip — the frozen DINOv2 image encoder inside sample_actions (reference:
// hypervla/components/base_vit.py:109-122 calling transformers' FlaxDinov2Module).
// 99.7 % of the per-step FLOPs.  Shared weights, so these are plain dense contractions with
// M = B * 257 rows:
//
//   im2col_kernel      u8 NHWC image -> [B*P, 2*Kp1] 16-bit patch matrix of (pixel - 128), twice (exact integers;
//                      the /255, mean, std normalisation is folded into the patch weights/bias at load and the
//                      weights are split hi|lo along K)
//   gemm_kernel        C = A[M,K] x W[N,K]^T on v_mfma_f32_16x16x32_{f16,bf16}: 128x128x64 block tile,
//                      4 waves x (64x64), LDS double-buffered with register prefetch, 144-B padded rows
//                      (conflict-free ds_read_b128), fused epilogues:
//                        PATCH  + bias + position embedding  -> f32 residual stream (row remap b*P+p -> b*S+1+p)
//                        QKV    + bias, q * 1/sqrt(hd)        -> 16-bit
//                        GELU   + bias, exact erf GELU        -> 16-bit
//                        RES    x += (acc + bias) * layerscale -> f32 residual stream (in place)
//   layernorm_kernel   f32 rows -> 16-bit rows (eps 1e-6), one wavefront per row; final variant drops the
//                      CLS row and writes the f32 patch tokens the policy consumes
//   attention_kernel   S = 257, head_dim 64: one workgroup per (image, head), one wavefront per 32-query
//                      block; K and V resident in LDS (V consumed through ds_read_b64_tr_b16); transposed scores (keys on accumulator rows,
//                      query on the lane) so softmax is in-lane and P feeds the PV MFMA from registers.
//
// Residual stream, LayerNorm statistics, softmax and GELU are f32; only MFMA operands are 16-bit.
#include <cstdlib>
#include <cstring>
#include <type_traits>

#include "common.h"
#include "kernels.h"

namespace hvla {

// ------------------------------------------------------------------------------------------------
// Row m = [a | a] with a[k] = pixel - 128 (k < patch*patch*3, else 0): centred integers are exact in
// fp16/bf16, and the weight matrix is [W_hi | W_lo] so the single 16-bit GEMM over K = 2*Kp1 computes
// a . (W_hi + W_lo): the patch embedding is then accurate to ~2^-20 instead of 2^-11 (it dominated
// the encoder's error when W was rounded once).
template <typename Op>
__global__ void im2col_kernel(const uint8_t* __restrict__ img, typename Op::elem* __restrict__ out, int B,
                              int image, int patch, int grid, int Kp) {
  // one thread = 8 consecutive k of one patch row
  const int chunks = Kp / 8, Kp1 = Kp / 2;
  const size_t total = (size_t)B * grid * grid * chunks;
  const int kreal = patch * patch * 3;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int ch = (int)(i % chunks);
    const size_t m = i / chunks;
    const int px = (int)(m % grid), py = (int)((m / grid) % grid);
    const size_t b = m / ((size_t)grid * grid);
    // k = (dy * patch + dx) * 3 + c: a patch row is 3 * patch consecutive bytes of the image row, so one division finds
    // (dy, byte in row) of the chunk's first element and the other seven follow by compare / subtract
    typename Op::x8 v;
    const int rowb = 3 * patch;
    int k0 = ch * 8;
    k0 = k0 >= Kp1 ? k0 - Kp1 : k0;                  // Kp1 is a multiple of 8: a chunk never straddles the two halves
    int dy = k0 / rowb, rb = k0 - dy * rowb;
    const uint8_t* src = img + ((b * image + (size_t)(py * patch + dy)) * image + (size_t)px * patch) * 3;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      float f = 0.f;
      if (k0 + j < kreal) f = (float)src[rb] - 128.f;
      v[j] = (typename Op::elem)f;
      if (++rb == rowb) rb = 0, src += (size_t)image * 3;
    }
    *reinterpret_cast<typename Op::x8*>(out + m * Kp + ch * 8) = v;
  }
}

__global__ void cls_rows_kernel(float* __restrict__ x, const float* __restrict__ pos, int B, int S, int E) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < B * E) x[(size_t)(i / E) * S * E + (i % E)] = pos[i % E];
}

// ------------------------------------------------------------------------------------------------
enum { EPI_PATCH = 0, EPI_QKV = 1, EPI_GELU = 2, EPI_RES = 3 };

struct GemmArgs {
  const void* A;   // [M][K] 16-bit
  const void* W;   // [N][K] 16-bit
  int M, N, K;
  const float* bias;     // [N]
  const float* aux;      // PATCH: pos [S][E];  RES: layerscale [N]
  void* out;             // 16-bit [M][N] or f32 [M'][N]
  int P, S;              // PATCH row remap
  int qcols;             // QKV: columns < qcols are scaled by qscale
  float qscale;
  int lda = 0, ldw = 0;  // row strides (elements) of A and W; 0 = K
  int split_from = 0;    // RES only: logical tiles >= split_from are split along K into split_parts
  int split_parts = 0;   // workgroups that accumulate into x with f32 atomics (tail-round fix)
  int no_dma_epilogue = 0;   // diagnostics: residual tile through registers instead of LDS-DMA
};

// Epilogue shared by both GEMM kernels.  acc[nt][mt]: lane holds column m = m_base + 16 mt + fr and rows
// n = n_base + 16 nt + 4 fq + {0..3}.  Loads of the residual tile are issued in batches ahead of the
// stores (the compiler cannot prove the in-place x += ... stores do not alias the next loads and would
// otherwise serialise 32 dependent round trips to HBM per lane).
template <typename Op, int EPI, int NT, int MT>
__device__ __forceinline__ void gemm_epilogue(const f32x4 (&acc)[NT][MT], const GemmArgs& g, int m_base, int n_base,
                                              int fr, int fq) {
  using T = typename Op::elem;
  f32x4 b4[NT], l4[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    b4[nt] = *reinterpret_cast<const f32x4*>(g.bias + n_base + nt * 16 + fq * 4);
    if constexpr (EPI == EPI_RES) l4[nt] = *reinterpret_cast<const f32x4*>(g.aux + n_base + nt * 16 + fq * 4);
  }
#pragma unroll
  for (int mp = 0; mp < MT; mp += 2) {
    f32x4 xin[2][NT];
    bool ok[2];
    size_t row[2];
    int prow[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int m = m_base + (mp + u) * 16 + fr;
      ok[u] = m < g.M;
      row[u] = (size_t)m;
      prow[u] = 0;
      if constexpr (EPI == EPI_PATCH) {
        prow[u] = 1 + (m % g.P);
        row[u] = (size_t)(m / g.P) * g.S + prow[u];
      }
      if constexpr (EPI == EPI_RES || EPI == EPI_PATCH) {
        if (ok[u]) {
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) {
            const int n = n_base + nt * 16 + fq * 4;
            if constexpr (EPI == EPI_RES)
              xin[u][nt] = *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(g.out) + row[u] * g.N + n);
            else
              xin[u][nt] = *reinterpret_cast<const f32x4*>(g.aux + (size_t)prow[u] * g.N + n);
          }
        }
      }
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      if (!ok[u]) continue;
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const int n = n_base + nt * 16 + fq * 4;
        f32x4 v = acc[nt][mp + u];
        if constexpr (EPI == EPI_PATCH) {
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] *= g.qscale;     // patch weights are stored x256 (16-bit range)
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] += b4[nt][r];
        if constexpr (EPI == EPI_QKV) {
          if (n < g.qcols) {
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] *= g.qscale;
          }
          typename Op::x4 o;
#pragma unroll
          for (int r = 0; r < 4; ++r) o[r] = (T)v[r];
          *reinterpret_cast<typename Op::x4*>(reinterpret_cast<T*>(g.out) + row[u] * g.N + n) = o;
        } else if constexpr (EPI == EPI_GELU) {
          typename Op::x4 o;
#pragma unroll
          for (int r = 0; r < 4; ++r) o[r] = (T)gelu_erf(v[r]);
          *reinterpret_cast<typename Op::x4*>(reinterpret_cast<T*>(g.out) + row[u] * g.N + n) = o;
        } else if constexpr (EPI == EPI_RES) {
          f32x4 x = xin[u][nt];
#pragma unroll
          for (int r = 0; r < 4; ++r) x[r] = fmaf(v[r], l4[nt][r], x[r]);
          *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(g.out) + row[u] * g.N + n) = x;
        } else {
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] += xin[u][nt][r];
          *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(g.out) + row[u] * g.N + n) = v;
        }
      }
    }
  }
}

// Row-major epilogue of the production kernels (gemm256p_kernel, gemm_kernel).  Those run the MFMA with the activation
// fragment as the first operand and stage W so that the LDS row 16 nt + c of a wave's 64 columns holds global column
// 4 c + nt (wperm() below): lane (fr, fq) then owns, for each of its rows m = m_base + 16 mt + 4 fq + r, the FOUR
// CONSECUTIVE columns n_base + 4 fr + {0..3} (one from each accumulator tile nt).  A store instruction therefore covers
// 4 rows x 64 consecutive columns -- four full 128-B lines of a 16-bit output, eight of an f32 one -- instead of 64
// scattered 8-B pieces (measured on the QKV shape: the old per-lane-column layout spent 8.6 us of a 24 us tile in its
// epilogue even on an otherwise idle chip; the memory pipe handles one line per cycle, not one instruction).
__device__ __forceinline__ int wperm(int rho) { return (rho & ~63) + 4 * (rho & 15) + ((rho >> 4) & 3); }

template <typename Op, int EPI, int MT, bool FULL>
__device__ __forceinline__ void gemm_epilogue_rows_impl(const f32x4 (&acc)[4][MT], const GemmArgs& g, int m_base, int n_base,
                                                        int fr, int fq, const f32x4* pre_b4, const f32x4* pre_l4) {
  using T = typename Op::elem;
  const int n = n_base + 4 * fr;
  const f32x4 b4 = pre_b4 ? *pre_b4 : *reinterpret_cast<const f32x4*>(g.bias + n);
  f32x4 l4 = f32x4{1.f, 1.f, 1.f, 1.f};
  if constexpr (EPI == EPI_RES) l4 = pre_l4 ? *pre_l4 : *reinterpret_cast<const f32x4*>(g.aux + n);
  const float q = EPI == EPI_PATCH ? g.qscale : ((EPI == EPI_QKV && n_base < g.qcols) ? g.qscale : 1.f);
  // FULL (every row of the wave tile exists): straight-line code, 32-bit element offsets from the uniform base.  With the
  // per-row `m < M` branches the compiler has to put an s_waitcnt vmcnt(0) into every predicated block (for the bias load),
  // which also waits for the previous STORE: 32 serialised store round trips per wave, 6.5 us per 256x256 tile.
  constexpr int RB = MT < 2 ? 1 : (EPI == EPI_RES && MT >= 4 ? 4 : 2);   // m-tiles per batch: residual loads of a batch are issued together
#pragma unroll
  for (int mp = 0; mp < MT; mp += RB) {
    f32x4 xin[RB][4];
    uint32_t off[RB][4];
    bool ok[RB][4];
#pragma unroll
    for (int u = 0; u < RB; ++u)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int m = m_base + (mp + u) * 16 + 4 * fq + r;
        ok[u][r] = FULL || m < g.M;
        off[u][r] = (uint32_t)m * (uint32_t)g.N + (uint32_t)n;
        if constexpr (EPI == EPI_PATCH) {
          const int prow = 1 + (m % g.P);
          off[u][r] = (uint32_t)((m / g.P) * g.S + prow) * (uint32_t)g.N + (uint32_t)n;
          if (ok[u][r]) xin[u][r] = *reinterpret_cast<const f32x4*>(g.aux + (uint32_t)prow * (uint32_t)g.N + (uint32_t)n);
        }
        if constexpr (EPI == EPI_RES) {
          if (ok[u][r]) xin[u][r] = *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(g.out) + off[u][r]);
        }
      }
#pragma unroll
    for (int u = 0; u < RB; ++u) {
      f32x4 t[4];                                    // [column c] over the four rows r: the accumulators' own register pairs
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        if constexpr (EPI == EPI_PATCH) t[c] = acc[c][mp + u] * q + b4[c];      // patch weights are stored x256 (16-bit range)
        else t[c] = acc[c][mp + u] + b4[c];
        if constexpr (EPI == EPI_QKV) t[c] *= q;
        if constexpr (EPI == EPI_GELU) {
          const f32x2 g0 = gelu_erf2(f32x2{t[c][0], t[c][1]}), g1 = gelu_erf2(f32x2{t[c][2], t[c][3]});
          t[c] = f32x4{g0[0], g0[1], g1[0], g1[1]};
        }
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        if (!ok[u][r]) continue;
        if constexpr (EPI == EPI_QKV || EPI == EPI_GELU) {
          typename Op::x4 o;
#pragma unroll
          for (int c = 0; c < 4; ++c) o[c] = (T)t[c][r];
          *reinterpret_cast<typename Op::x4*>(reinterpret_cast<T*>(g.out) + off[u][r]) = o;
        } else if constexpr (EPI == EPI_RES) {
          f32x4 x = xin[u][r];
#pragma unroll
          for (int c = 0; c < 4; ++c) x[c] = fmaf(t[c][r], l4[c], x[c]);
          *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(g.out) + off[u][r]) = x;
        } else {
          f32x4 x = xin[u][r];
#pragma unroll
          for (int c = 0; c < 4; ++c) x[c] += t[c][r];
          *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(g.out) + off[u][r]) = x;
        }
      }
    }
  }
}

template <typename Op, int EPI, int MT>
__device__ __forceinline__ void gemm_epilogue_rows(const f32x4 (&acc)[4][MT], const GemmArgs& g, int m_base, int n_base,
                                                   int fr, int fq, const f32x4* pre_b4 = nullptr,
                                                   const f32x4* pre_l4 = nullptr) {
  if (m_base + 16 * MT <= g.M) gemm_epilogue_rows_impl<Op, EPI, MT, true>(acc, g, m_base, n_base, fr, fq, pre_b4, pre_l4);
  else gemm_epilogue_rows_impl<Op, EPI, MT, false>(acc, g, m_base, n_base, fr, fq, pre_b4, pre_l4);
}

// split-K tail tiles of a RES GEMM: x += (acc + [part 0] bias) * layerscale with f32 atomics
// (global_atomic_add_f32; order-dependent in the last bits, only the <= 1 % of rows of the tail tiles)
template <int NT, int MT>
__device__ __forceinline__ void gemm_epilogue_atomic(const f32x4 (&acc)[NT][MT], const GemmArgs& g, int m_base,
                                                     int n_base, int fr, int fq, bool add_bias) {
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    const int n = n_base + nt * 16 + fq * 4;
    f32x4 b4 = *reinterpret_cast<const f32x4*>(g.bias + n);
    const f32x4 l4 = *reinterpret_cast<const f32x4*>(g.aux + n);
    if (!add_bias) b4 = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      const int m = m_base + mt * 16 + fr;
      if (m >= g.M) continue;
      float* xp = reinterpret_cast<float*>(g.out) + (size_t)m * g.N + n;
#pragma unroll
      for (int r = 0; r < 4; ++r) unsafeAtomicAdd(xp + r, (acc[nt][mt][r] + b4[r]) * l4[r]);
    }
  }
}

constexpr int GBM = 128, GBN = 128, GBK = 64, GLD = 72;   // GLD: padded LDS row (halves) = 144 B

template <typename Op, int EPI>
__global__ __launch_bounds__(256, 2) void gemm_kernel(GemmArgs g) {
  using T = typename Op::elem;
  using X8 = typename Op::x8;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  T* As = reinterpret_cast<T*>(smem);                 // [2][GBM][GLD]
  T* Ws = As + 2 * GBM * GLD;                         // [2][GBN][GLD]
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int wm = wave >> 1, wn = wave & 1;            // wave tile: rows m [64*wm, +64), cols n [64*wn, +64)
  // XCD-aware tile order: blocks b and b+8 share an XCD (guide T1); give each XCD a contiguous run of
  // M-tiles for one N-tile column so the W panel and neighbouring A panels stay in its L2.
  const int nbm = (g.M + GBM - 1) / GBM, nbn = g.N / GBN;
  int bid = blockIdx.x;
  {
    const int nwg = nbm * nbn, q = nwg / 8, r = nwg % 8, xcd = bid % 8;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + bid / 8;
  }
  const int bm = bid % nbm, bn = bid / nbm;
  const int m0 = bm * GBM, n0 = bn * GBN;
  const T* A = reinterpret_cast<const T*>(g.A);
  const T* W = reinterpret_cast<const T*>(g.W);

  // global -> register staging: thread loads 4 x 16 B of A and of W per k-tile
  const int lrow = tid >> 3, lch = tid & 7;
  const T* ap[4];
  const T* wp[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    int m = m0 + lrow + 32 * i;
    m = m < g.M ? m : g.M - 1;
    ap[i] = A + (size_t)m * g.K + lch * 8;
    wp[i] = W + (size_t)(n0 + wperm(lrow + 32 * i)) * g.K + lch * 8;   // LDS row -> column permutation of the epilogue
  }
  X8 ra[4], rw[4];
  auto gload = [&](int kt) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      ra[i] = *reinterpret_cast<const X8*>(ap[i] + kt * GBK);
      rw[i] = *reinterpret_cast<const X8*>(wp[i] + kt * GBK);
    }
  };
  auto sstore = [&](int buf) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      *reinterpret_cast<X8*>(As + (buf * GBM + lrow + 32 * i) * GLD + lch * 8) = ra[i];
      *reinterpret_cast<X8*>(Ws + (buf * GBN + lrow + 32 * i) * GLD + lch * 8) = rw[i];
    }
  };

  f32x4 acc[4][4];   // [n-tile][m-tile]: D = A-tile (16 m rows) x W-tile^T (16 n cols)
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int KT = g.K / GBK;
  gload(0);
  sstore(0);
  __syncthreads();
  const int fr = lane & 15, fq = lane >> 4;
  for (int kt = 0; kt < KT; ++kt) {
    const int buf = kt & 1;
    if (kt + 1 < KT) gload(kt + 1);
    const T* as = As + (buf * GBM + wm * 64 + fr) * GLD + fq * 8;
    const T* ws = Ws + (buf * GBN + wn * 64 + fr) * GLD + fq * 8;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      X8 fa[4], fw[4];
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        fa[t] = *reinterpret_cast<const X8*>(as + t * 16 * GLD + kk * 32);
        fw[t] = *reinterpret_cast<const X8*>(ws + t * 16 * GLD + kk * 32);
      }
#pragma unroll
      for (int nt = 0; nt < 4; ++nt)
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) acc[nt][mt] = Op::mma16(fa[mt], fw[nt], acc[nt][mt]);
    }
    if (kt + 1 < KT) sstore(buf ^ 1);
    __syncthreads();
  }

  gemm_epilogue_rows<Op, EPI, 4>(acc, g, m0 + wm * 64, n0 + wn * 64, fr, fq);
}

// ------------------------------------------------------------------------------------------------
// gemm64_kernel -- 64x64x64 tiles for SMALL row counts (the peeled tail rows of a big GEMM, B = 1..3): such a problem
// is pure latency, so it is cut into many small workgroups (256 rows x 768 columns -> 48) and each keeps SNS - 1 = 5
// K-tiles of LDS-DMA in flight (6 stages x 16 KB, counted vmcnt, one raw barrier per K-tile).  Four waves, wave w owns
// rows [16 w, +16) x all 64 columns: 10 ds_read_b128 and 8 MFMAs per K-tile.  LDS image, swizzle, W-row permutation, MFMA
// and k order are those of gemm256p_kernel, so a row gets the same bits whichever kernel computes it.
constexpr int SBM = 64, SBN = 64, SNS = 6;
template <typename Op, int EPI>
__global__ __launch_bounds__(256) void gemm64_kernel(GemmArgs g) {
  using T = typename Op::elem;
  using X8 = typename Op::x8;
  extern __shared__ __attribute__((aligned(16))) char smem[];   // SNS x (A 8 KB | W 8 KB)
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nbm = (g.M + SBM - 1) / SBM;
  const int bm = blockIdx.x % nbm, bn = blockIdx.x / nbm;
  const int m0 = bm * SBM, n0 = bn * SBN;
  const T* A = reinterpret_cast<const T*>(g.A);
  const T* W = reinterpret_cast<const T*>(g.W);
  // LDS-DMA pieces: instruction j = 0, 1 of wave w fills rows [32 j + 8 w, +8); lane -> (row = lane >> 3, LDS chunk = lane & 7),
  // source chunk (lane & 7) ^ (row & 7)
  const int sch = ((lane & 7) ^ (lane >> 3)) * 8;
  uint32_t aoff[2], woff[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int rl = 32 * j + 8 * wave + (lane >> 3);
    int m = m0 + rl;
    m = m < g.M ? m : g.M - 1;
    aoff[j] = (uint32_t)m * (uint32_t)g.K + sch;
    woff[j] = (uint32_t)(n0 + wperm(rl)) * (uint32_t)g.K + sch;
  }
  const int KT = g.K / 64;
  auto issue = [&](int kt) {
    char* base = smem + (kt % SNS) * 16384 + wave * 1024;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(A + aoff[j] + kt * 64),
                                       (__attribute__((address_space(3))) void*)(base + j * 4096), 16, 0, 0);
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(W + woff[j] + kt * 64),
                                       (__attribute__((address_space(3))) void*)(base + 8192 + j * 4096), 16, 0, 0);
    }
  };
  f32x4 acc[4][1];
#pragma unroll
  for (int i = 0; i < 4; ++i) acc[i][0] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int fr = lane & 15, fq = lane >> 4;
  const int sw0 = (fq ^ (fr & 7)) << 4, sw1 = ((fq + 4) ^ (fr & 7)) << 4;
  const int a_off = (wave * 16 + fr) * 128, w_off = 8192 + fr * 128;
#pragma unroll
  for (int s = 0; s < SNS - 1; ++s)
    if (s < KT) issue(s);
  for (int kt = 0; kt < KT; ++kt) {
    const int issued = kt + SNS - 1 < KT ? kt + SNS - 1 : KT;
    switch (issued - kt - 1) {                       // K-tiles issued after tile kt may stay in flight (4 pieces each)
      case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
      case 1: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
      case 2: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
      case 3: asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); break;
      default: asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); break;
    }
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();                    // tile kt landed for every wave; everyone is done reading tile kt - 1
    __builtin_amdgcn_sched_barrier(0);
    if (kt + SNS - 1 < KT) issue(kt + SNS - 1);      // into the stage of tile kt - 1
    const char* lb = smem + (kt % SNS) * 16384;
    X8 fa[2], fw[4][2];
    fa[0] = *reinterpret_cast<const X8*>(lb + a_off + sw0);
    fa[1] = *reinterpret_cast<const X8*>(lb + a_off + sw1);
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
      fw[nt][0] = *reinterpret_cast<const X8*>(lb + w_off + nt * 2048 + sw0);
      fw[nt][1] = *reinterpret_cast<const X8*>(lb + w_off + nt * 2048 + sw1);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) acc[nt][0] = Op::mma16(fa[kk], fw[nt][kk], acc[nt][0]);
  }
  gemm_epilogue_rows<Op, EPI, 1>(acc, g, m0 + wave * 16, n0, fr, fq);
}

// ------------------------------------------------------------------------------------------------
// 256x256x64 block tiles, 8 waves (2 along M x 4 along N, 128x64 each), operands staged global -> LDS by LDS-DMA
// (global_load_lds_dwordx4: no VGPR round trip, 1 KiB per wave-instruction).  The LDS image is lane-linear (hardware
// writes base + lane*16), so the bank swizzle chunk ^= (row & 7) is applied to the per-lane SOURCE address and again on
// the ds_read_b128 address (guide §5.4 rule 21): 128-B rows then read conflict-free.
constexpr int HBM_ = 256, HBN_ = 256;

// ------------------------------------------------------------------------------------------------
// 256x256x64 tile, 8 waves, FOUR PHASES per K-tile with the two wave rows running half a phase apart (guide §5 "256^2
// 8-phase template": counted vmcnt, raw s_barrier, staggered wave groups).
//   * a phase = { ds_read fragments | issue one 16 KB half-tile of LDS-DMA } barrier { 16 MFMAs = one quadrant of the
//     wave's 128x64 tile over K = 64 } barrier.  Waves 4-7 execute one extra barrier up front, so while waves 0-3 are in
//     their MFMA cluster waves 4-7 (the other wave on each SIMD) read LDS / issue DMA, and vice versa: the matrix pipe and
//     the LDS / memory pipes are both busy all the time instead of all eight waves wanting the same pipe at once.
//   * all fragments of a K-tile are read in its first two phases (A rows 0-63 + W rows 0-31, then A rows 64-127 + W rows
//     32-63 of the wave's tile), so its LDS buffer is free from phase 3 on; tile t+2's A halves are staged there in phases
//     3 and 4 of tile t, its W halves in phases 1 and 2 of tile t+1.  The wait in phase 4 is the constant vmcnt(4): tile
//     t+1 has landed, the two A halves of tile t+2 stay in flight across the barriers.
//   * the last two K-tiles are peeled (nothing is staged past the end, the wait constants stay immediates).
// LDS image and swizzle as described above (128-byte rows, chunk ^= row & 7 on the source and on the read).
// FL (diagnostics): 1 no stagger, 2 s_setprio(1) around the MFMA clusters (measured 10 % SLOWER here, off by default),
// 4 no DMA in the loop, 8 no MFMA, 16 no fragment reads in the loop, 32 no epilogue
template <typename Op, int EPI, bool PEEL = true, int FL = 0, bool PERSIST = false>
__global__ __launch_bounds__(512) void gemm256p_kernel(GemmArgs g) {
  using T = typename Op::elem;
  using X8 = typename Op::x8;
  extern __shared__ __attribute__((aligned(16))) char smem[];   // 2 buffers x (A 32 KB | W 32 KB)
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  const int nbm = (g.M + HBM_ - 1) / HBM_, nbn = g.N / HBN_;
  const int ntiles = nbm * nbn;
  const int GN = nbn % 4 == 0 ? 4 : (nbn % 3 == 0 ? 3 : (nbn % 2 == 0 ? 2 : 1));
  // virtual block id v (= blockIdx.x, + k * gridDim.x in the persistent form: gridDim.x is a multiple of 8, so a workgroup
  // stays in its XCD's id range) -> tile origin
  auto tile_origin = [&](int v, int& m0, int& n0) {
    const int q = ntiles / 8, r = ntiles % 8, xcd = v % 8;
    const int bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + v / 8;
    constexpr int CH = 8;
    const int per_chunk = CH * nbn;
    const int chunk = bid / per_chunk, rc = bid % per_chunk;
    const int rows = (nbm - chunk * CH) < CH ? (nbm - chunk * CH) : CH;
    const int sc = rc / (rows * GN), r2 = rc % (rows * GN);
    m0 = (chunk * CH + r2 / GN) * HBM_;
    n0 = (sc * GN + r2 % GN) * HBN_;
  };
  int vb = blockIdx.x, m0, n0;
  tile_origin(vb, m0, n0);
  const T* A = reinterpret_cast<const T*>(g.A);
  const T* W = reinterpret_cast<const T*>(g.W);
  // LDS-DMA pieces: instruction j of wave w fills rows [64 j + 8 w, +8) of A (or W); lane -> (row = lane >> 3, LDS chunk =
  // lane & 7), source chunk (lane & 7) ^ (row & 7).  Half-tile h = 0,1: A rows [128 h, +128) (j = 2h, 2h + 1); h = 2,3: W.
  const int srow = wave * 8 + (lane >> 3);
  const int sch = ((lane & 7) ^ (lane >> 3)) * 8;
  // PERSIST (every tile is full: the host peels the tail rows): one per-lane offset for A and one for W plus wave-uniform
  // row bases, i.e. 2 VGPRs instead of 8 and SGPR-base addressing -- the tile loop has no registers to spare
  uint32_t aoff[4], woff[4];
  const uint32_t lane_a = ((uint32_t)srow * (uint32_t)g.K + sch) * (uint32_t)sizeof(T);                 // bytes
  const uint32_t lane_w = ((uint32_t)(wperm(srow)) * (uint32_t)g.K + sch) * (uint32_t)sizeof(T);        // wperm(64 j + srow) = 64 j + wperm(srow)
  const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
  const T* abase = A;
  const T* wbase = W;
  auto offsets = [&](int tm0, int tn0) {
    if constexpr (PERSIST) {
      abase = A + (size_t)tm0 * g.K;
      wbase = W + (size_t)tn0 * g.K;
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        int m = tm0 + 64 * j + srow;
        m = m < g.M ? m : g.M - 1;
        aoff[j] = (uint32_t)m * (uint32_t)g.K + sch;
        woff[j] = (uint32_t)(tn0 + ((FL & 64) ? 64 * j + srow : wperm(64 * j + srow))) * (uint32_t)g.K + sch;   // column permutation of the epilogue
      }
    }
  };
  offsets(m0, n0);
  const int KT = g.K / 64;
  auto stage = [&](auto hc, int kt) {              // half-tile hc of K-tile kt (clamped) into buffer kt & 1
    constexpr int h = decltype(hc)::value;
    if ((FL & 4) && kt > 1) return;
    const int buf = kt & 1, kc = PEEL ? kt : (kt < KT ? kt : KT - 1);
    char* base = smem + buf * 65536 + wave * 1024 + (h >> 1) * 32768;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int j = 2 * (h & 1) + u;
      if constexpr (PERSIST) {
        // SGPR row base + 32-bit per-lane byte offset, issued by hand: the builtin always takes a 64-bit VGPR address
        // (8 loop-invariant pairs that the register allocator spills, and every spill reload waits vmcnt(0))
        const T* sb = (h < 2 ? abase : wbase) + ((size_t)j * 64 * g.K + kc * 64);
        const uint32_t dst = lds0 + (uint32_t)(buf * 65536 + wave * 1024 + (h >> 1) * 32768 + j * 8192);
        const uint32_t vo = h < 2 ? lane_a : lane_w;
        asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1"
                     :: "v"(vo), "s"(sb), "s"(dst) );        // no "memory" clobber: it would make every issue wait for the fragment reads in
                                                       // flight (barriers / counted waits order it); M0 is reserved by the
                                                       // compiler, which has no use of its own for it in this instantiation
      } else {
        const T* src = (h < 2 ? A + aoff[j] : W + woff[j]) + kc * 64;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)(base + j * 8192), 16, 0, 0);
      }
    }
  };
  f32x4 acc[4][8];   // [n-tile][m-tile]
  const int fr = lane & 15, fq = lane >> 4;
  const int sw0 = ((fq) ^ (fr & 7)) << 4, sw1 = ((fq + 4) ^ (fr & 7)) << 4;
  const int a_off = (wm * 128 + fr) * 128, w_off = 32768 + (wn * 64 + fr) * 128;
  X8 fa0[8], fa1[8], fw0[4], fw1[4];               // [tile * 2 + kk]
  auto rd_a = [&](X8 (&f)[8], const char* lb, int ah) {
    if ((FL & 16) && lb != smem) return;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      f[2 * t] = *reinterpret_cast<const X8*>(lb + a_off + (4 * ah + t) * 2048 + sw0);
      f[2 * t + 1] = *reinterpret_cast<const X8*>(lb + a_off + (4 * ah + t) * 2048 + sw1);
    }
  };
  auto rd_w = [&](X8 (&f)[4], const char* lb, int wh) {
    if ((FL & 16) && lb != smem) return;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      f[2 * t] = *reinterpret_cast<const X8*>(lb + w_off + (2 * wh + t) * 2048 + sw0);
      f[2 * t + 1] = *reinterpret_cast<const X8*>(lb + w_off + (2 * wh + t) * 2048 + sw1);
    }
  };
  auto quad = [&](const X8 (&fa)[8], const X8 (&fw)[4], auto ahc, auto whc) {   // 16 MFMAs, 8 accumulators x 2 k-chunks
    constexpr int ah = decltype(ahc)::value, wh = decltype(whc)::value;
    if constexpr (FL & 8) {
#pragma unroll
      for (int i = 0; i < 8; ++i) asm volatile("" ::"v"(fa[i]));
#pragma unroll
      for (int i = 0; i < 4; ++i) asm volatile("" ::"v"(fw[i]));
      return;
    }
    if constexpr (FL & 2) __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
      for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
          if constexpr (FL & 64) acc[2 * wh + nt][4 * ah + mt] = Op::mma16(fw[2 * nt + kk], fa[2 * mt + kk], acc[2 * wh + nt][4 * ah + mt]);
          else acc[2 * wh + nt][4 * ah + mt] = Op::mma16(fa[2 * mt + kk], fw[2 * nt + kk], acc[2 * wh + nt][4 * ah + mt]);
    if constexpr (FL & 2) __builtin_amdgcn_s_setprio(0);
  };
  using H0 = std::integral_constant<int, 0>;
  using H1 = std::integral_constant<int, 1>;
  using H2 = std::integral_constant<int, 2>;
  using H3 = std::integral_constant<int, 3>;
#define HVLA_BAR() do { __builtin_amdgcn_sched_barrier(0); __builtin_amdgcn_s_barrier(); __builtin_amdgcn_sched_barrier(0); } while (0)
  // ---- prologue: tile 0 complete in buffer 0, the A halves of tile 1 in flight in buffer 1
  stage(H0{}, 0); stage(H1{}, 0); stage(H2{}, 0); stage(H3{}, 0);
  if (KT > 1 || !PEEL) {
    stage(H0{}, 1); stage(H1{}, 1);
    asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  } else {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  // one K-tile = four phases.  S1: tile kt+1 exists (stage its W halves in phases 1, 2); S2: tile kt+2 exists (stage its A
  // halves in phases 3, 4).  The last two K-tiles are peeled so that nothing is staged past the end and the phase-4 wait
  // stays an immediate: vmcnt(4) with both A halves of tile kt+2 in flight, vmcnt(0) without.
  auto ktile = [&](int kt, auto s1c, auto s2c) {
    constexpr bool S1 = decltype(s1c)::value, S2 = decltype(s2c)::value;
    const char* lb = smem + (kt & 1) * 65536;
    // phase 1: fragments A0, W0
    rd_w(fw0, lb, 0);
    rd_a(fa0, lb, 0);
    if constexpr (S1) stage(H2{}, kt + 1);
    HVLA_BAR();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    quad(fa0, fw0, H0{}, H0{});
    HVLA_BAR();
    // phase 2: fragments A1, W1 -- the last reads of this buffer, retired BEFORE the barrier: the other wave row restages
    // the buffer right after it
    rd_w(fw1, lb, 1);
    rd_a(fa1, lb, 1);
    if constexpr (S1) stage(H3{}, kt + 1);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    HVLA_BAR();
    quad(fa0, fw1, H0{}, H1{});
    HVLA_BAR();
    // phase 3
    if constexpr (S2) stage(H0{}, kt + 2);
    HVLA_BAR();
    quad(fa1, fw1, H1{}, H1{});
    HVLA_BAR();
    // phase 4: tile kt+1 must have landed (its last piece was issued in phase 2)
    if constexpr (S2) stage(H1{}, kt + 2);
    if constexpr (S2) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else if constexpr (S1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    HVLA_BAR();
    quad(fa1, fw0, H1{}, H0{});
    HVLA_BAR();
  };
  using Yes = std::integral_constant<bool, true>;
  using No = std::integral_constant<bool, false>;
  // PERSIST: gridDim.x workgroups walk the tiles v, v + gridDim.x, ...; the prologue DMA of the next tile (K-tile 0 and the
  // A halves of K-tile 1) is issued BEFORE this tile's epilogue, so its latency and the workgroup launch disappear under
  // the epilogue's VALU work and stores.  The wait for it counts the epilogue's memory operations, which were issued later
  // (vmcnt retires in order): at least 32 stores per wave, so vmcnt(36) leaves those and the two A halves in flight.
  while (true) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    HVLA_BAR();
    if (wm == 1 && !(FL & 1)) HVLA_BAR();             // waves 4-7 run half a phase behind
    if constexpr (PEEL) {
      int kt = 0;
      for (; kt + 2 < KT; ++kt) ktile(kt, Yes{}, Yes{});
      if (kt + 1 < KT) { ktile(kt, Yes{}, No{}); ++kt; }
      ktile(kt, No{}, No{});
    } else {      // diagnostics: stage past the end (clamped re-loads of the last tile), one loop body
      for (int kt = 0; kt < KT; ++kt) ktile(kt, Yes{}, Yes{});
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    if (wm == 0 && !(FL & 1)) HVLA_BAR();             // same number of barriers in both wave rows
    const int cm0 = m0, cn0 = n0;
    bool more = false;
    f32x4 pb4, pl4;
    if constexpr (PERSIST) {
      // bias / LayerScale of this tile are fetched and WAITED FOR before the next tile's DMA goes out: the compiler does not
      // see the hand-issued DMA, so a wait it places after that point is a vmcnt(0) that would drain the prefetch
      pb4 = *reinterpret_cast<const f32x4*>(g.bias + cn0 + wn * 64 + 4 * fr);
      pl4 = pb4;
      if constexpr (EPI == EPI_RES) pl4 = *reinterpret_cast<const f32x4*>(g.aux + cn0 + wn * 64 + 4 * fr);
      asm volatile("" : "+v"(pb4), "+v"(pl4));
      vb += gridDim.x;
      more = vb < ntiles;
      if (more) {
        tile_origin(vb, m0, n0);
        offsets(m0, n0);
        __builtin_amdgcn_sched_barrier(0);            // the wait below counts the epilogue's stores as issued AFTER this DMA:
        stage(H0{}, 0); stage(H1{}, 0); stage(H2{}, 0); stage(H3{}, 0);
        stage(H0{}, 1); stage(H1{}, 1);
        __builtin_amdgcn_sched_barrier(0);            // nothing may be scheduled across it in either direction
      }
    }
    if constexpr (FL & 32) {                          // diagnostics: no epilogue (one lane keeps the accumulators alive)
      float s = 0.f;
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) s += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
      if (s == 12345.678f) reinterpret_cast<float*>(g.out)[0] = s;
    } else if constexpr (FL & 64) {
      gemm_epilogue<Op, EPI, 4, 8>(acc, g, cm0 + wm * 128, cn0 + wn * 64, fr, fq);
    } else if constexpr (FL & 128) {                  // diagnostics: every CU stores into the first tile rows (stays in L2)
      gemm_epilogue_rows<Op, EPI, 8>(acc, g, wm * 128, (blockIdx.x % (g.N / HBN_)) * HBN_ + wn * 64, fr, fq);
    } else {
      if constexpr (PERSIST) gemm_epilogue_rows<Op, EPI, 8>(acc, g, cm0 + wm * 128, cn0 + wn * 64, fr, fq, &pb4, &pl4);
      else gemm_epilogue_rows<Op, EPI, 8>(acc, g, cm0 + wm * 128, cn0 + wn * 64, fr, fq);
    }
    if (!more) break;
    if constexpr (EPI == EPI_QKV || EPI == EPI_GELU) {
      if (cm0 + HBM_ <= g.M) asm volatile("s_waitcnt vmcnt(36)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // an edge tile issues fewer stores
    } else {
      if (cm0 + HBM_ <= g.M) asm volatile("s_waitcnt vmcnt(63)" ::: "memory");   // 64+ loads and stores behind the DMA
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
  }
#undef HVLA_BAR
}

// In-place residual epilogue of the ring kernel with the x tile PREFETCHED by LDS-DMA.  Loading x through
// registers costs ~18 us per 256x256 tile (32 dependent 16-B round trips per lane interleaved with stores
// that may alias them); here the tile comes in as four 64-row quarters of 64 KB through the (now idle) LDS
// ring, two quarters in flight, each row one 1-KiB piece with its 16-B chunks XOR-swizzled by (row & 15) so
// that the accumulator-layout reads (16 lanes = 16 different rows, same column chunk) are conflict-free.
template <int MT>
__device__ __forceinline__ void res_epilogue_dma(const f32x4 (&acc)[4][MT], const GemmArgs& g, char* smem, int m0,
                                                 int n0, int wave, int lane) {
  static_assert(MT == 8, "8-wave layout");
  const int wm = wave >> 2, wn = wave & 3, fr = lane & 15, fq = lane >> 4;
  float* X = reinterpret_cast<float*>(g.out);
  const bool edge = m0 + 256 > g.M;                 // some rows clamped / stores skipped: use full drains
  f32x4 b4[4], l4[4];
#pragma unroll
  for (int nt = 0; nt < 4; ++nt) {
    b4[nt] = *reinterpret_cast<const f32x4*>(g.bias + n0 + wn * 64 + nt * 16 + fq * 4);
    l4[nt] = *reinterpret_cast<const f32x4*>(g.aux + n0 + wn * 64 + nt * 16 + fq * 4);
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");     // every wave is done with the ring
  auto dma_q = [&](int qi) {
    char* dst = smem + (qi & 1) * 65536 + wave * 8192;
#pragma unroll
    for (int jj = 0; jj < 8; ++jj) {
      const int p = wave * 8 + jj;                                          // local row 0..63 (wave-uniform)
      int m = m0 + (p >> 5) * 128 + (2 * qi + ((p >> 4) & 1)) * 16 + (p & 15);
      m = m < g.M ? m : g.M - 1;
      const float* src = X + (size_t)m * g.N + n0 + ((lane ^ (p & 15)) << 2);
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                       (__attribute__((address_space(3))) void*)(dst + jj * 1024), 16, 0, 0);
    }
  };
  auto proc_q = [&](int qi) {
    const char* lb = smem + (qi & 1) * 65536;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int mt = 2 * qi + u;
      const int rl = wm * 32 + u * 16 + fr;
      const int m = m0 + wm * 128 + mt * 16 + fr;
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) {
        const int c = wn * 16 + nt * 4 + fq;
        f32x4 x = *reinterpret_cast<const f32x4*>(lb + rl * 1024 + ((c ^ fr) << 4));
#pragma unroll
        for (int r = 0; r < 4; ++r) x[r] = fmaf(acc[nt][mt][r] + b4[nt][r], l4[nt][r], x[r]);
        if (m < g.M) *reinterpret_cast<f32x4*>(X + (size_t)m * g.N + n0 + wn * 64 + nt * 16 + fq * 4) = x;
      }
    }
  };
  dma_q(0);
  dma_q(1);
  // outstanding, oldest first: Q0 x8 | Q1 x8
  if (edge) asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  asm volatile("s_barrier" ::: "memory");
  proc_q(0);                                                                  // + 8 stores
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");              // buffer 0 free
  dma_q(2);
  // Q1 x8 | st x8 | Q2 x8
  if (edge) asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
  asm volatile("s_barrier" ::: "memory");
  proc_q(1);
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");              // buffer 1 free
  dma_q(3);
  // st x8 | Q2 x8 | st x8 | Q3 x8
  if (edge) asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
  asm volatile("s_barrier" ::: "memory");
  proc_q(2);
  // st x8 | Q3 x8 | st x8
  if (edge) asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  asm volatile("s_barrier" ::: "memory");
  proc_q(3);
}

// ------------------------------------------------------------------------------------------------
// gemm256r_kernel — the production 256x256 GEMM.  Measured on the simpler kernel above: with one K-tile
// in flight the LDS-DMA side alone needs 1.36 us per 64-deep K-tile (latency-bound) and the MFMA side
// 1.33 us (every group of 8 MFMAs waits on its ds_reads), and the two overlap only half.  This version
//   * stages K in 32-deep SLOTS (A 16 KB + W 16 KB) through a ring of 4 slots: three slots are landed or
//     in flight ahead of the one being consumed (counted s_waitcnt vmcnt(8), raw s_barrier — a
//     __syncthreads() would drain the DMA queue, guide §5 "Pipelining across barriers");
//   * issues the DMA one piece per MFMA group (4 pieces per slot and wave), never as a burst;
//   * rolls the fragment reads one MFMA group ahead (2 + 4 ping-pong register sets, 48 VGPRs) so the
//     matrix pipe does not wait on LDS; the barrier of phase i sits BEFORE its last MFMA group, whose
//     fragments are already in registers, so slot i is free for the DMA of slot i+4 right behind it and
//     the first fragments of slot i+1 are fetched under that last group.
// LDS rows are 64 B here; chunk' = chunk ^ LUT[(row >> 2) & 3], LUT = {0,2,3,1}, applied to the DMA source
// address and to the read address, makes every ds_read_b128 lane group hit 16 distinct 16-B slots.
template <int N>
__device__ __forceinline__ void wait_vmcnt() {      // counted drain of the LDS-DMA queue (immediate operand)
  static_assert(N >= 0 && N <= 17, "vmcnt immediate");
#define HVLA_W(K) else if constexpr (N == K) asm volatile("s_waitcnt vmcnt(" #K ")" ::: "memory");
  if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  HVLA_W(1) HVLA_W(2) HVLA_W(3) HVLA_W(4) HVLA_W(5) HVLA_W(6) HVLA_W(7) HVLA_W(8) HVLA_W(9) HVLA_W(10) HVLA_W(11)
  HVLA_W(12) HVLA_W(13) HVLA_W(14) HVLA_W(15) HVLA_W(16) HVLA_W(17)
#undef HVLA_W
}

template <typename Op, int EPI, int WM = 2, int ABL = 0, int RING = 4>   // ABL (diagnostics): 1 no DMA in loop, 2 no MFMA, 3 DMA only
__global__ __launch_bounds__(WM * 256) void gemm256r_kernel(GemmArgs g) {
  constexpr int D = RING - 1;                        // slots landed or in flight ahead of the one being consumed
  // WM waves along M x 4 along N; per-wave tile (256 / WM) x 64 = MT x 4 MFMA tiles.
  //   WM = 2:  8 waves, 128x64 per wave (128 accumulator VGPRs, 2 waves per SIMD)
  //   WM = 4: 16 waves,  64x64 per wave ( 64 accumulator VGPRs, 4 waves per SIMD: more issue interleave)
  constexpr int NWV = WM * 4, MT = 16 / WM, G = MT / 2, PPW = 32 / NWV;   // groups / DMA pieces per phase
  static_assert(G == PPW, "one DMA piece per MFMA group");
  using T = typename Op::elem;
  using X8 = typename Op::x8;
  extern __shared__ __attribute__((aligned(16))) char smem[];   // RING slots x 32 KB
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  const int nbm = (g.M + HBM_ - 1) / HBM_, nbn = g.N / HBN_;
  int bid = blockIdx.x, part = 0, nparts = 1;
  {
    const int nfull = g.split_parts > 1 ? g.split_from : nbm * nbn;
    if (bid < nfull) {
      const int q = nfull / 8, r = nfull % 8, xcd = bid % 8;
      bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + bid / 8;
    } else {                      // tail tile, split along K
      const int e = bid - nfull;
      nparts = g.split_parts;
      part = e % nparts;
      bid = nfull + e / nparts;
    }
  }
  // Tile order inside an XCD's contiguous id range: chunks of 8 M-tiles; inside a chunk the N super-columns
  // (GN N-tiles each) one after the other.  The ~32 tiles an XCD runs at once are then an 8 x GN patch whose
  // A and W K-slices share its 4 MiB L2, and the next patch reuses the SAME 8 A panels (still in L2), so the
  // activation matrix is fetched from HBM once instead of once per super-column.
  const int GN = nbn % 4 == 0 ? 4 : (nbn % 3 == 0 ? 3 : (nbn % 2 == 0 ? 2 : 1));
  constexpr int CH = 8;
  const int per_chunk = CH * nbn;
  const int chunk = bid / per_chunk, rc = bid % per_chunk;
  const int rows = (nbm - chunk * CH) < CH ? (nbm - chunk * CH) : CH;      // last chunk may be short
  const int sc = rc / (rows * GN), r2 = rc % (rows * GN);
  const int bm = chunk * CH + r2 / GN, bn = sc * GN + r2 % GN;
  const int m0 = bm * HBM_, n0 = bn * HBN_;
  const T* A = reinterpret_cast<const T*>(g.A);
  const T* W = reinterpret_cast<const T*>(g.W);
  // ---- DMA pieces: a slot has 32 pieces of 1 KiB (0..15: A rows 16 q.., 16..31: W rows); wave w moves
  // pieces q = w + NWV * j.  lane -> row + (lane >> 2), LDS chunk lane & 3, source chunk (lane & 3) ^ LUT
  const int lut = (0x1320 >> (((lane >> 4) & 3) * 4)) & 3;          // {0,2,3,1}[(row >> 2) & 3]
  const int sch = ((lane & 3) ^ lut) * 8;
  const int NP = g.K / 32 / nparts;            // phases of this workgroup: [part * NP, (part + 1) * NP)
  uint32_t poff[PPW];
#pragma unroll
  for (int j = 0; j < PPW; ++j) {
    const int q = wave + NWV * j;
    if (q < 16) {
      int m = m0 + q * 16 + (lane >> 2);
      m = m < g.M ? m : g.M - 1;
      poff[j] = (uint32_t)m * (uint32_t)(g.lda ? g.lda : g.K) + sch;
    } else {
      poff[j] = (uint32_t)(n0 + (q - 16) * 16 + (lane >> 2)) * (uint32_t)(g.ldw ? g.ldw : g.K) + sch;
    }
    poff[j] += (uint32_t)(part * NP * 32);
  }
  auto dma = [&](int j, int slot_k /* phase index */) {
    if (ABL == 1 && slot_k > D) return;
    const int q = wave + NWV * j;                        // wave-uniform
    char* dst = smem + (slot_k % RING) * 32768 + q * 1024;
    const T* src = (q < 16 ? A : W) + poff[j] + slot_k * 32;
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
  };
  f32x4 acc[4][MT];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < MT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int fr = lane & 15, fq = lane >> 4;
  const int rlut = (0x1320 >> (((fr >> 2) & 3) * 4)) & 3;
  const int a_off = (wm * (16 * MT) + fr) * 64 + ((fq ^ rlut) << 4);
  const int w_off = 16384 + (wn * 64 + fr) * 64 + ((fq ^ rlut) << 4);
  auto rd_a = [&](X8 (&fa)[2], int slot_k, int mp) {
    if (ABL == 3) return;
    const char* lb = smem + (slot_k % RING) * 32768 + a_off + mp * 2048;
    fa[0] = *reinterpret_cast<const X8*>(lb);
    fa[1] = *reinterpret_cast<const X8*>(lb + 1024);
  };
  auto rd_w = [&](X8 (&fw)[4], int slot_k) {
    if (ABL == 3) return;
    const char* lb = smem + (slot_k % RING) * 32768 + w_off;
#pragma unroll
    for (int t = 0; t < 4; ++t) fw[t] = *reinterpret_cast<const X8*>(lb + t * 1024);
  };
  auto mma_g = [&](const X8 (&fa)[2], const X8 (&fw)[4], int mp) {
    if constexpr (ABL == 2) {
      asm volatile("" ::"v"(fa[0]), "v"(fa[1]), "v"(fw[0]), "v"(fw[1]), "v"(fw[2]), "v"(fw[3]));
      return;
    }
    if constexpr (ABL == 3) return;
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) acc[nt][2 * mp + u] = Op::mma16(fw[nt], fa[u], acc[nt][2 * mp + u]);
  };
  X8 aA[2], aB[2], wA[4], wB[4];
  // ---- prologue: slots 0..D-1 and the first piece of slot D in flight; slot 0 landed
#pragma unroll
  for (int sl = 0; sl < D; ++sl)
#pragma unroll
    for (int j = 0; j < PPW; ++j) dma(j, sl);
  dma(0, D);
  wait_vmcnt<(D - 1) * PPW + 1>();
  asm volatile("s_barrier" ::: "memory");
  rd_w(wA, 0);
  rd_a(aA, 0, 0);
  // One phase = one 32-deep slot = G MFMA groups of 8.  MODE 1: steady state (pieces 1.. of slot i+3 at the
  // first groups, piece 0 of slot i+4 behind the barrier); MODE 2: only finish slot i+3; MODE 0: no DMA.
  // VM: slots that may still be in flight when slot i+1 must have landed.  LAST: no next slot to read.
  auto phase = [&](int i, auto& wc, auto& wnx, auto mode, auto vm, auto last) {
    constexpr int MODE = decltype(mode)::value, VM = decltype(vm)::value;
    constexpr bool LAST = decltype(last)::value;
    // program order == issue order we want; the sched_group_barrier chains pin it (the default schedule
    // sinks each ds_read to just in front of its MFMAs and then waits for it: LDS latency fully exposed)
#define HVLA_GRP(NDS, NVM)                                                         \
  __builtin_amdgcn_sched_group_barrier(0x100, NDS, 0); /* DS reads, next group */ \
  if constexpr (NVM) __builtin_amdgcn_sched_group_barrier(0x010, 1, 0);           \
  __builtin_amdgcn_sched_group_barrier(0x008, 8, 0); /* 8 MFMA, this group     */
    rd_a(aB, i, 1);
    if constexpr (MODE != 0) dma(1, i + D);
    mma_g(aA, wc, 0);
    HVLA_GRP(2, MODE != 0)
    if constexpr (G == 4) {
      rd_a(aA, i, 2);
      if constexpr (MODE != 0) dma(2, i + D);
      mma_g(aB, wc, 1);
      HVLA_GRP(2, MODE != 0)
      rd_a(aB, i, 3);
      if constexpr (MODE != 0) dma(3, i + D);
      mma_g(aA, wc, 2);
      HVLA_GRP(2, MODE != 0)
    }
    if constexpr (!LAST) {
      wait_vmcnt<VM * PPW>();
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
      rd_w(wnx, i + 1);
      rd_a(aA, i + 1, 0);
      if constexpr (MODE == 1) dma(0, i + D + 1);
      mma_g(aB, wc, G - 1);
      HVLA_GRP(6, MODE == 1)
    } else {
      mma_g(aB, wc, G - 1);
    }
#undef HVLA_GRP
  };
  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;
  using I2 = std::integral_constant<int, 2>;
  using Yes = std::integral_constant<bool, true>;
  using No = std::integral_constant<bool, false>;
  using ID1 = std::integral_constant<int, D - 1>;
  // NP is even (K % 64 == 0), phases go in (wA, wB) pairs; the tail is a fixed, branch-free sequence so that
  // every register array keeps a static name (a run-time parity switch here spills 600 B per lane)
  int i = 0;
  if constexpr (D == 3) {
    for (; i + 4 < NP; i += 2) {
      phase(i, wA, wB, I1{}, ID1{}, No{});
      phase(i + 1, wB, wA, I1{}, ID1{}, No{});
    }
    phase(i, wA, wB, I2{}, I2{}, No{});          // NP-4: finish slot NP-1
    phase(i + 1, wB, wA, I0{}, I1{}, No{});
    phase(i + 2, wA, wB, I0{}, I0{}, No{});
    phase(i + 3, wB, wA, I0{}, I0{}, Yes{});
  } else {
    for (; i + 6 < NP; i += 2) {
      phase(i, wA, wB, I1{}, ID1{}, No{});
      phase(i + 1, wB, wA, I1{}, ID1{}, No{});
    }
    phase(i, wA, wB, I1{}, ID1{}, No{});         // NP-6: last steady phase
    phase(i + 1, wB, wA, I2{}, ID1{}, No{});     // NP-5: finish slot NP-1
    phase(i + 2, wA, wB, I0{}, I2{}, No{});
    phase(i + 3, wB, wA, I0{}, I1{}, No{});
    phase(i + 4, wA, wB, I0{}, I0{}, No{});
    phase(i + 5, wB, wA, I0{}, I0{}, Yes{});
  }
  if constexpr (EPI == EPI_RES) {
    if (nparts > 1) {
      gemm_epilogue_atomic<4, MT>(acc, g, m0 + wm * (16 * MT), n0 + wn * 64, fr, fq, part == 0);
      return;
    }
    if constexpr (WM == 2 && ABL == 0) {
      if (!g.no_dma_epilogue) {
        res_epilogue_dma<MT>(acc, g, smem, m0, n0, wave, lane);
        return;
      }
    }
  }
  gemm_epilogue<Op, EPI, 4, MT>(acc, g, m0 + wm * (16 * MT), n0 + wn * 64, fr, fq);
}

// ------------------------------------------------------------------------------------------------
// LayerNorm: one wave per row of E f32 (E % 4 == 0, E <= 1024); FINAL drops row 0 of every image and
// writes f32.
// FINAL: 0 = 16-bit output for the next GEMM; 1 = f32 patch tokens with the CLS row dropped (base_vit.py:120-122);
//        2 = f32 last_hidden_state with every row (the evaluators' initial-image embedding)
template <typename Op, int FINAL>
__global__ __launch_bounds__(256) void layernorm_kernel(const float* __restrict__ x, void* __restrict__ out,
                                                        const float* __restrict__ scale,
                                                        const float* __restrict__ bias, int M, int E, int S) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= M) return;
  if (FINAL == 1 && (row % S) == 0) return;
  const f32x4* xr = reinterpret_cast<const f32x4*>(x + (size_t)row * E);
  const int n4 = E / 4;
  f32x4 v[4];
  float sum = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = lane + 64 * i;
    v[i] = c < n4 ? xr[c] : f32x4{0.f, 0.f, 0.f, 0.f};
    sum += v[i][0] + v[i][1] + v[i][2] + v[i][3];
  }
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) sum += __shfl_xor(sum, o, 64);
  const float mean = sum / E;
  float sq = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = lane + 64 * i;
    if (c < n4) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float d = v[i][j] - mean;
        sq += d * d;
      }
    }
  }
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) sq += __shfl_xor(sq, o, 64);
  const float rstd = rsqrtf(sq / E + 1e-6f);
  size_t orow = row;
  if (FINAL == 1) orow = (size_t)(row / S) * (S - 1) + (row % S) - 1;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = lane + 64 * i;
    if (c < n4) {
      const f32x4 s4 = reinterpret_cast<const f32x4*>(scale)[c], b4 = reinterpret_cast<const f32x4*>(bias)[c];
      f32x4 y;
#pragma unroll
      for (int j = 0; j < 4; ++j) y[j] = (v[i][j] - mean) * rstd * s4[j] + b4[j];
      if (FINAL) {
        reinterpret_cast<f32x4*>(reinterpret_cast<float*>(out) + orow * E)[c] = y;
      } else {
        typename Op::x4 o;
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = (typename Op::elem)y[j];
        reinterpret_cast<typename Op::x4*>(reinterpret_cast<typename Op::elem*>(out) + orow * E)[c] = o;
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// Attention, head_dim 64.  qkv [B*S][3E] 16-bit (q already scaled by log2(e)/sqrt(64)); out o [B*S][E].
constexpr int AKLD = 72;          // K row stride in LDS (halves): 144 B
constexpr int AVLD = 64;          // V row stride (halves): 128 B, 64-B halves swapped on rows with bit 1 set

// transposed LDS read (ds_read_b64_tr_b16, guide T10): per 16-lane group a 4-row x 16-column block of 16-bit
// elements comes back column-major; lane 4q+p supplies the address of row q, columns 4p..4p+3 and lane i
// receives column i of the 4 rows.  EXEC must be all ones.
typedef short hvla_s4 __attribute__((__vector_size__(4 * sizeof(short))));
template <typename X8>
__device__ __forceinline__ X8 tr_read2(const void* p0, const void* p1) {
  const hvla_s4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) hvla_s4*)p0);
  const hvla_s4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) hvla_s4*)p1);
  typedef short s8 __attribute__((ext_vector_type(8)));
  const s8 v = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
  return __builtin_bit_cast(X8, v);
}

// 16-lane row reductions by DPP (quad swaps, then the half-row and row mirrors) and an SGPR broadcast of one lane
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ float row16_max(float v) {
  v = fmaxf(v, dpp_mov<0xB1>(v));
  v = fmaxf(v, dpp_mov<0x4E>(v));
  v = fmaxf(v, dpp_mov<0x141>(v));
  return fmaxf(v, dpp_mov<0x140>(v));
}
__device__ __forceinline__ float row16_sum(float v) {
  v += dpp_mov<0xB1>(v);
  v += dpp_mov<0x4E>(v);
  v += dpp_mov<0x141>(v);
  return v + dpp_mov<0x140>(v);
}
__device__ __forceinline__ float lane_bcast(float v, int l) {
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), l));
}

template <typename Op>
__global__ void attention_kernel(const typename Op::elem* __restrict__ qkv, typename Op::elem* __restrict__ o,
                                 int S, int E, int H) {
  // S = 32 * NW + 1 tokens.  NW waves of 64 lanes: wave w owns queries [32 w, 32 w + 32) on the matrix cores;
  // the one remaining query (the last token) is done co-operatively on the VALU, wave w taking key tile w,
  // and combined through LDS.  8 waves per workgroup at S = 257 (2 per SIMD) so that two workgroups share a
  // CU (the 9-wave version only ever had one resident).
  using T = typename Op::elem;
  using X8 = typename Op::x8;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int NW = (S - 1) / 32, KT = NW + 1, SP = KT * 32;
  T* Ks = reinterpret_cast<T*>(smem);                      // [SP][64]  keys, 16-B chunks XOR-swizzled by key & 7
  T* Vs = Ks + SP * AVLD;                                  // [SP][64]  values, row-major (read transposed)
  T* qxs = Vs + SP * AVLD;                                 // [64]       the last query (parked here, not in registers)
  float* part = reinterpret_cast<float*>(qxs + 64);        // [KT][66]   partial (max, sum, O[64]) of the last query
  const int b = blockIdx.x / H, head = blockIdx.x % H;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, nthr = blockDim.x;
  const size_t rowstride = (size_t)3 * E;
  const T* base = qkv + (size_t)b * S * rowstride + head * 64;
  // ---- this wave's 32 queries as B fragments (natural d order): 4 k-steps of 16 -- requested first, used last
  const int col = lane & 31, half = lane >> 5;
  const int q = wave * 32 + col;                           // always < S - 1
  X8 qf[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) qf[ks] = *reinterpret_cast<const X8*>(base + (size_t)q * rowstride + ks * 16 + half * 8);
  // the extra query (token S-1) goes to LDS: 16 VGPRs less across the MFMA loops (the kernel runs at the 128-VGPR limit
  // of 4 waves per SIMD)
  if (tid < 8) *reinterpret_cast<X8*>(qxs + tid * 8) = *reinterpret_cast<const X8*>(base + (size_t)(S - 1) * rowstride + tid * 8);
  // ---- stage K and V: 16 B per thread and chunk (zero rows for the padding keys).  SP * 8 / nthr = 4 (NW + 1) / NW <= 8
  // chunks per thread; ALL their loads are requested before the first LDS store (a rolled loop pays one HBM round trip
  // per iteration: 5 in a row at S = 257, a third of the workgroup's life time).
  // K first, then V: V is only needed in the second pass, so its loads stay in flight (in registers) under the first.
  constexpr int STG = 8;
  X8 kreg[STG], vreg[STG];
#pragma unroll
  for (int it = 0; it < STG; ++it) {
    if (it * nthr >= SP * 8) break;                        // uniform
    const int i = tid + it * nthr, key = i >> 3, ch = i & 7;
#pragma unroll
    for (int j = 0; j < 8; ++j) kreg[it][j] = (T)0.f;
    if (i < SP * 8 && key < S) kreg[it] = *reinterpret_cast<const X8*>(base + (size_t)key * rowstride + E + ch * 8);
  }
#pragma unroll
  for (int it = 0; it < STG; ++it) {
    if (it * nthr >= SP * 8) break;
    const int i = tid + it * nthr, key = i >> 3, ch = i & 7;
#pragma unroll
    for (int j = 0; j < 8; ++j) vreg[it][j] = (T)0.f;
    if (i < SP * 8 && key < S) vreg[it] = *reinterpret_cast<const X8*>(base + (size_t)key * rowstride + 2 * E + ch * 8);
  }
#pragma unroll
  for (int it = 0; it < STG; ++it) {
    if (it * nthr >= SP * 8) break;
    const int i = tid + it * nthr, key = i >> 3, ch = i & 7;
    if (i < SP * 8) *reinterpret_cast<X8*>(Ks + key * AVLD + ((ch ^ (key & 7)) * 8)) = kreg[it];
  }
  __syncthreads();
  auto stage_v = [&] {
#pragma unroll
    for (int it = 0; it < STG; ++it) {
      if (it * nthr >= SP * 8) break;
      const int i = tid + it * nthr, key = i >> 3, ch = i & 7;
      if (i < SP * 8) *reinterpret_cast<X8*>(Vs + key * AVLD + ((ch ^ (((key >> 1) & 1) << 2)) * 8)) = vreg[it];
    }
    __syncthreads();
  };

  // per-lane part of the transposed-read address: row (half * 4 + q), column 16 * dgrp + 4 p, and the 64-B
  // half swap of rows with bit 1 set (q >= 2)
  const int g16 = lane >> 4, i16 = lane & 15;
  const int vsw = ((i16 >> 3) & 1) << 5;
  const T* vtr = Vs + ((g16 >> 1) * 4 + (i16 >> 2)) * AVLD + (g16 & 1) * 16 + (i16 & 3) * 4;
  const int kswz = col & 7;                                // K chunk swizzle of this lane's key row
  // Two passes over the resident K tiles instead of an online softmax: the matrix pipe is idle most of the
  // time here (head_dim 64: 8 MFMAs per 1024 scores), so recomputing K.Q^T (4 MFMAs) is cheaper than the
  // per-tile rescale of O and the running-max bookkeeping.
  //   pass 1: row max (scores are in the log2 domain: q carries 1/sqrt(64) * log2 e from the QKV epilogue)
  //   pass 2: accumulator initialised to -max, so p = exp2(acc) with no subtract; invalid keys get -1e30
  auto kfrag = [&](int kt, int ks) {
    return *reinterpret_cast<const X8*>(Ks + (kt * 32 + col) * AVLD + (((2 * ks + half) ^ kswz) * 8));
  };
  // This kernel is VALU-issue-bound (it spent about 4 VALU cycles per MFMA cycle): the accumulator input of a score tile
  // is the inline constant 0 and the padding mask only exists in the peeled last key tile, so the loops carry no
  // per-element initialisation or select; -max goes into the exp2 argument.
  auto qk = [&](int kt, f32x16 c) {
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) c = Op::mma32(kfrag(kt, ks), qf[ks], c);
    return c;
  };
  auto zero16 = [] {
    f32x16 z;
#pragma unroll
    for (int r = 0; r < 16; ++r) z[r] = 0.f;
    return z;
  };
  auto mask16 = [&] {                // 0 for real keys, -1e30 for the padding keys of the last tile
    f32x16 z;
#pragma unroll
    for (int r = 0; r < 16; ++r) z[r] = ((KT - 1) * 32 + crow(r, half) < S) ? 0.f : -1e30f;
    return z;
  };
  float mx = -1e30f;
  auto rowmax = [&](const f32x16& sc) {
#pragma unroll
    for (int r = 0; r < 16; ++r) mx = fmaxf(mx, sc[r]);
  };
  for (int kt = 0; kt < KT - 1; ++kt) rowmax(qk(kt, zero16()));
  rowmax(qk(KT - 1, mask16()));
  mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
  stage_v();
  typedef float f32x2v __attribute__((ext_vector_type(2)));
  f32x2v lsum2 = {0.f, 0.f};        // this half's partial denominator (two interleaved partial sums)
  f32x16 O[2];
#pragma unroll
  for (int r = 0; r < 16; ++r) O[0][r] = 0.f, O[1][r] = 0.f;
  auto pv = [&](int kt, const f32x16& sc) {
    X8 pf[2];
#pragma unroll
    for (int r = 0; r < 16; r += 2) {
      const f32x2v p2 = {__builtin_amdgcn_exp2f(sc[r] - mx), __builtin_amdgcn_exp2f(sc[r + 1] - mx)};
      lsum2 += p2;
      pf[r >> 3][r & 7] = (T)p2[0];
      pf[r >> 3][(r & 7) + 1] = (T)p2[1];
    }
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
#pragma unroll
      for (int sstep = 0; sstep < 2; ++sstep) {
        // A operand = V^T: lane (d = 32 mt + (l & 31), half) needs keys 16 s + 8 (j >> 2) + 4 half + (j & 3)
        const T* v0 = vtr + ((kt * 32 + sstep * 16) * AVLD) + ((mt * 32) ^ vsw);
        O[mt] = Op::mma32(tr_read2<X8>(v0, v0 + 8 * AVLD), pf[sstep], O[mt]);
      }
    }
  };
  for (int kt = 0; kt < KT - 1; ++kt) pv(kt, qk(kt, zero16()));
  pv(KT - 1, qk(KT - 1, mask16()));
  const float lsum = lsum2[0] + lsum2[1];
  const float inv = 1.f / (lsum + __shfl_xor(lsum, 32, 64));
  {
    T* op = o + ((size_t)b * S + q) * E + head * 64;
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        typename Op::x4 v4;
#pragma unroll
        for (int r = 0; r < 4; ++r) v4[r] = (T)(O[mt][g4 * 4 + r] * inv);
        *reinterpret_cast<typename Op::x4*>(op + mt * 32 + g4 * 8 + half * 4) = v4;
      }
  }
  // ---- the last query, VALU: wave w scores key tile w (lane = key, the two halves split d), the last wave also the
  // final key S-1; partial softmax + partial P.V (lane = d); combine across waves through LDS.
  {
    X8 qx[4];                            // lane holds d = 32 * half .. +32 of the last query
#pragma unroll
    for (int c = 0; c < 4; ++c) qx[c] = *reinterpret_cast<const X8*>(qxs + half * 32 + c * 8);
    auto score = [&](int key) {          // sum over this half's 32 d
      float a = 0.f;
      const T* kr = Ks + key * AVLD;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const X8 kv = *reinterpret_cast<const X8*>(kr + (((half * 4 + c) ^ (key & 7)) * 8));
#pragma unroll
        for (int j = 0; j < 8; ++j) a = fmaf((float)qx[c][j], (float)kv[j], a);
      }
      return a + __shfl_xor(a, 32, 64);
    };
    // wave w: key tile w.  Reductions over the 32 keys by DPP inside the 16-lane rows + two readlanes (a ds_bpermute
    // butterfly is five dependent LDS round trips), the P.V row by readlane broadcasts of p.
    {
      const int tile = wave;
      const float sc = score(tile * 32 + col);
      float m = row16_max(sc);
      m = fmaxf(lane_bcast(m, 0), lane_bcast(m, 16));
      const float p = __builtin_amdgcn_exp2f(sc - m);
      float l = row16_sum(p);
      l = lane_bcast(l, 0) + lane_bcast(l, 16);
      float od = 0.f;                    // lane = d
#pragma unroll
      for (int i = 0; i < 32; ++i) {
        const int k2 = tile * 32 + i;
        od = fmaf(lane_bcast(p, i), (float)Vs[k2 * AVLD + (lane ^ (((k2 >> 1) & 1) << 5))], od);
      }
      float* pp = part + tile * 66;
      if (lane == 0) pp[0] = m, pp[1] = l;
      pp[2 + lane] = od;
    }
    if (wave == NW - 1) {                // the one real key of tile KT-1 (= S-1): p = 1, sum = 1, P.V = its V row
      const int key = S - 1;
      const float sc = score(key);
      float* pp = part + (KT - 1) * 66;
      if (lane == 0) pp[0] = sc, pp[1] = 1.f;
      pp[2 + lane] = (float)Vs[key * AVLD + (lane ^ (((key >> 1) & 1) << 5))];
    }
  }
  __syncthreads();
  if (wave == 0) {
    float M = -1e30f;
    for (int t = 0; t < KT; ++t) M = fmaxf(M, part[t * 66]);
    float L = 0.f, od = 0.f;
    for (int t = 0; t < KT; ++t) {
      const float f = __builtin_amdgcn_exp2f(part[t * 66] - M);
      L = fmaf(part[t * 66 + 1], f, L);
      od = fmaf(part[t * 66 + 2 + lane], f, od);
    }
    o[((size_t)b * S + (S - 1)) * E + head * 64 + lane] = (T)(od / L);
  }
}

// ------------------------------------------------------------------------------------------------
template <typename Op>
static hipError_t run_encoder(const Geom& g, const EncWeights& w, const EncWorkspace& ws, const uint8_t* images,
                              float* tokens, int B, hipStream_t st, Profiler* prof, bool keep_cls) {
  Profiler none;
  Profiler& pf = prof ? *prof : none;
  using T = typename Op::elem;
  const int P = g.P(), S = g.S(), E = g.E, F = g.enc_mlp, H = g.enc_heads;
  const int Kp = 2 * ((g.patch * g.patch * 3 + 63) / 64 * 64);   // [a | a] x [W_hi | W_lo]
  const int M = B * S;
  // the GEMM epilogues address their outputs with 32-bit element offsets
  if ((size_t)M * (size_t)(F > 3 * E ? F : 3 * E) >= (1ull << 32)) return hipErrorInvalidValue;
  const size_t gsm = (size_t)2 * (GBM + GBN) * GLD * sizeof(T);
  static bool attr = false;
  if (!attr) {
    hipError_t e;
#define SETA(K) \
  if ((e = hipFuncSetAttribute(reinterpret_cast<const void*>(K), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)) != hipSuccess) return e;
    SETA((gemm_kernel<Op, EPI_PATCH>)) SETA((gemm_kernel<Op, EPI_QKV>)) SETA((gemm_kernel<Op, EPI_GELU>))
    SETA((gemm_kernel<Op, EPI_RES>)) SETA((attention_kernel<Op>))
    SETA((gemm256r_kernel<Op, EPI_PATCH>)) SETA((gemm256r_kernel<Op, EPI_QKV>)) SETA((gemm256r_kernel<Op, EPI_GELU>))
    SETA((gemm256r_kernel<Op, EPI_RES>))
    SETA((gemm256p_kernel<Op, EPI_PATCH>)) SETA((gemm256p_kernel<Op, EPI_QKV>)) SETA((gemm256p_kernel<Op, EPI_GELU>))
    SETA((gemm256p_kernel<Op, EPI_RES>))
    SETA((gemm256p_kernel<Op, EPI_PATCH, true, 0, true>)) SETA((gemm256p_kernel<Op, EPI_QKV, true, 0, true>))
    SETA((gemm256p_kernel<Op, EPI_GELU, true, 0, true>)) SETA((gemm256p_kernel<Op, EPI_RES, true, 0, true>))
    SETA((gemm64_kernel<Op, EPI_PATCH>)) SETA((gemm64_kernel<Op, EPI_QKV>)) SETA((gemm64_kernel<Op, EPI_GELU>))
    SETA((gemm64_kernel<Op, EPI_RES>))
#undef SETA
    attr = true;
  }
  static const char* gsel = getenv("HVLA_GEMM");     // diagnostics: "128" | "ring" select another kernel
  static const bool phased = !(gsel && !strcmp(gsel, "ring"));
  static const bool nopeel = getenv("HVLA_NO_PEEL") != nullptr;
  static const bool nopersist = getenv("HVLA_NO_PERSIST") != nullptr;
  static const int g64_maxm = getenv("HVLA_G64_MAXM") ? atoi(getenv("HVLA_G64_MAXM")) : 2047;   // rows up to which the 64x64 kernel is used
  // split-K of the tail-round tiles of the residual GEMMs is opt-in: it buys < 1 % of the step, and its f32 atomic adds make
  // the one episode that owns those rows run-to-run different by up to 2e-3 in its tokens (tools/tail_probe.py)
  static const bool nosplit = getenv("HVLA_SPLIT_TAIL") == nullptr;
  static const bool no_dma_epi = getenv("HVLA_NO_DMA_EPILOGUE") != nullptr;
  static int ncu = 0;
  if (!ncu) {
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || ncu <= 0) ncu = 256;
  }
  bool fc1_main_closed = false;   // profiler mode 1: the fc1 bracket covers the 256x256 launch only (see gemm)
  auto gemm = [&](auto kern, auto kern64, auto kern256r, auto kern256p, auto kern256pp, const void* A, const void* Wt, int Mm, int N, int K,
                  const float* bias, const float* aux, void* out, int qcols, bool is_res = false, bool peel = true,
                  int main_cat = -1) {
    GemmArgs a{A, Wt, Mm, N, K, bias, aux, out, P, S, qcols, qcols ? 0.125f * 1.4426950408889634f : 1.f / 256.f};   // q: 1/sqrt(64) and exp -> exp2
    a.no_dma_epilogue = no_dma_epi;
    const bool big = N % HBN_ == 0 && Mm > g64_maxm && Mm >= 1024 && !(gsel && !strcmp(gsel, "128"));
    const bool fits32 = (size_t)Mm * K < (1ull << 31) && (size_t)N * K < (1ull << 31);
    if (big && K >= 256 && fits32 && phased) {
      const int nbn = N / HBN_;
      int nbm = (Mm + HBM_ - 1) / HBM_;
      // One workgroup per CU: the grid runs in rounds of `ncu` tiles, and B x 257 tokens leave a handful of tiles
      // (257 = 256 + 1 M-tiles) that start an extra round (launch time steps with ceil(tiles / ncu): fc2 85 us per round).
      // Peel the last r M-tile rows off so that the main grid is a whole number of rounds and give them to the 128x128
      // kernel (four times the workgroups per tile, same k order and MFMA, so the same bits) right behind it.  Worth 1.2 %
      // of the step (fc2 -3 %, out-proj -4 %): the small kernel is latency-bound (48 K-steps for fc2), so most of the round
      // comes back as its run time.
      const int rem = (nbm * nbn) % ncu, r = (rem + nbn - 1) / nbn;
      if (peel && !nopeel && rem > 0 && r * 8 <= nbm && ((nbm - r) * nbn) % ncu == 0 && Mm % HBM_ == 0) {
        const int m_main = (nbm - r) * HBM_, m_tail = Mm - m_main;
        GemmArgs t = a;
        t.A = reinterpret_cast<const T*>(A) + (size_t)m_main * K;
        t.out = is_res ? static_cast<void*>(reinterpret_cast<float*>(out) + (size_t)m_main * N)
                       : static_cast<void*>(reinterpret_cast<T*>(out) + (size_t)m_main * N);
        t.M = m_tail;
        a.M = m_main;
        // whole rounds of full tiles: one persistent workgroup per CU walks them, with the next tile's first DMA in flight
        // under the current tile's epilogue
        if (!nopersist && K >= 128) hipLaunchKernelGGL(kern256pp, dim3(ncu), dim3(512), 131072, st, a);
        else hipLaunchKernelGGL(kern256p, dim3((nbm - r) * nbn), dim3(512), 131072, st, a);
        if (main_cat >= 0 && pf.mode == 1) { pf.end(main_cat, st); fc1_main_closed = true; }
        hipLaunchKernelGGL(kern64, dim3(((m_tail + SBM - 1) / SBM) * (N / SBN)), dim3(256), SNS * 16384, st, t);
      } else if (!nopersist && K >= 128 && Mm % HBM_ == 0 && (nbm * nbn) % ncu == 0) {
        hipLaunchKernelGGL(kern256pp, dim3(ncu), dim3(512), 131072, st, a);
      } else {
        hipLaunchKernelGGL(kern256p, dim3(nbm * nbn), dim3(512), 131072, st, a);
      }
    } else if (big && K >= 256 && fits32) {
      int nb = ((Mm + HBM_ - 1) / HBM_) * (N / HBN_);
      // tail-round fix (in-place residual epilogue only): when a few tiles spill into an extra round on the
      // 256 CUs, split those along K over otherwise idle CUs
      const int remt = nb % ncu;
      if (is_res && nb > ncu && remt > 0 && remt <= ncu / 8 && !nosplit) {
        const int NPt = K / 32;
        int parts = 1;
        for (int c = NPt / 4; c >= 2; --c)
          if (NPt % c == 0 && (NPt / c) % 2 == 0 && remt * c <= ncu) { parts = c; break; }
        if (parts > 1) {
          a.split_from = nb - remt;
          a.split_parts = parts;
          nb = a.split_from + remt * parts;
        }
      }
      hipLaunchKernelGGL(kern256r, dim3(nb), dim3(512), 131072, st, a);
    } else if (Mm <= g64_maxm && fits32 && N % SBN == 0 && K % 64 == 0) {
      hipLaunchKernelGGL(kern64, dim3(((Mm + SBM - 1) / SBM) * (N / SBN)), dim3(256), SNS * 16384, st, a);
    } else {
      const int nb = ((Mm + GBM - 1) / GBM) * (N / GBN);
      hipLaunchKernelGGL(kern, dim3(nb), dim3(256), gsm, st, a);
    }
  };
  // patch embedding
  pf.begin(0, st);
  {
    const size_t total = (size_t)B * P * (Kp / 8);
    int blocks = (int)((total + 255) / 256);
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(im2col_kernel<Op>, dim3(blocks), dim3(256), 0, st, images, reinterpret_cast<T*>(ws.g), B,
                       g.image_size, g.patch, g.grid(), Kp);
    hipLaunchKernelGGL(cls_rows_kernel, dim3((B * E + 255) / 256), dim3(256), 0, st, ws.x, w.pos, B, S, E);
    gemm(gemm_kernel<Op, EPI_PATCH>, gemm64_kernel<Op, EPI_PATCH>, gemm256r_kernel<Op, EPI_PATCH>, gemm256p_kernel<Op, EPI_PATCH>, gemm256p_kernel<Op, EPI_PATCH, true, 0, true>, ws.g, w.w_patch, B * P, E, Kp, w.b_patch, w.pos, ws.x, 0, false, false);
  }
  pf.end(0, st);
  const int KT = (S + 31) / 32;     // S = 32 * (KT - 1) + 1
  const size_t asm_bytes = (size_t)KT * 32 * 2 * AVLD * sizeof(T) + (size_t)KT * 66 * sizeof(float) + 64 * sizeof(T);
  for (int l = 0; l < g.enc_layers; ++l) {
    const EncLayerW& L = w.layer[l];
    pf.begin(1, st);
    hipLaunchKernelGGL((layernorm_kernel<Op, 0>), dim3((M + 3) / 4), dim3(256), 0, st, ws.x, ws.h, L.ln1_s,
                       L.ln1_b, M, E, S);
    pf.end(1, st);
    pf.begin(2, st);
    gemm(gemm_kernel<Op, EPI_QKV>, gemm64_kernel<Op, EPI_QKV>, gemm256r_kernel<Op, EPI_QKV>, gemm256p_kernel<Op, EPI_QKV>, gemm256p_kernel<Op, EPI_QKV, true, 0, true>, ws.h, L.wqkv, M, 3 * E, E, L.bqkv, nullptr, ws.qkv, E);
    pf.end(2, st);
    pf.begin(3, st);
    hipLaunchKernelGGL(attention_kernel<Op>, dim3(B * H), dim3((KT - 1) * 64), asm_bytes, st,
                       reinterpret_cast<const T*>(ws.qkv), reinterpret_cast<T*>(ws.h), S, E, H);
    pf.end(3, st);
    pf.begin(4, st);
    gemm(gemm_kernel<Op, EPI_RES>, gemm64_kernel<Op, EPI_RES>, gemm256r_kernel<Op, EPI_RES>, gemm256p_kernel<Op, EPI_RES>, gemm256p_kernel<Op, EPI_RES, true, 0, true>, ws.h, L.wo, M, E, E, L.bo, L.ls1, ws.x, 0, true);
    pf.end(4, st);
    pf.begin(1, st);
    hipLaunchKernelGGL((layernorm_kernel<Op, 0>), dim3((M + 3) / 4), dim3(256), 0, st, ws.x, ws.h, L.ln2_s,
                       L.ln2_b, M, E, S);
    pf.end(1, st);
    fc1_main_closed = false;
    pf.begin(5, st);   // in "dominant kernel only" mode the bracket is closed right behind the main launch (inside gemm)
    gemm(gemm_kernel<Op, EPI_GELU>, gemm64_kernel<Op, EPI_GELU>, gemm256r_kernel<Op, EPI_GELU>, gemm256p_kernel<Op, EPI_GELU>, gemm256p_kernel<Op, EPI_GELU, true, 0, true>, ws.h, L.w1, M, F, E, L.b1, nullptr, ws.g, 0, false,
         true, 5);
    if (!(pf.mode == 1 && fc1_main_closed)) pf.end(5, st);
    pf.begin(6, st);
    gemm(gemm_kernel<Op, EPI_RES>, gemm64_kernel<Op, EPI_RES>, gemm256r_kernel<Op, EPI_RES>, gemm256p_kernel<Op, EPI_RES>, gemm256p_kernel<Op, EPI_RES, true, 0, true>, ws.g, L.w2, M, E, F, L.b2, L.ls2, ws.x, 0, true);
    pf.end(6, st);
  }
  pf.begin(1, st);
  if (keep_cls)
    hipLaunchKernelGGL((layernorm_kernel<Op, 2>), dim3((M + 3) / 4), dim3(256), 0, st, ws.x, tokens, w.lnf_s, w.lnf_b, M, E, S);
  else
    hipLaunchKernelGGL((layernorm_kernel<Op, 1>), dim3((M + 3) / 4), dim3(256), 0, st, ws.x, tokens, w.lnf_s, w.lnf_b, M, E, S);
  pf.end(1, st);
  return hipGetLastError();
}

// diagnostics: time `iters` launches of one GEMM shape on workspace buffers (contents irrelevant)
hipError_t debug_gemm(const void* A, const void* W, const float* bias, const float* aux, void* out, int M, int N,
                      int K, int epi, int variant, int iters, float* ms, hipStream_t st) {
  using Op = OpF16;
  GemmArgs a{A, W, M, N, K, bias, aux, out, 256, 257, epi == EPI_QKV ? N / 3 : 0, 0.125f};
  if (const char* e = getenv("HVLA_DBG_LDA")) a.lda = K + atoi(e);
  if (const char* e = getenv("HVLA_DBG_LDW")) a.ldw = K + atoi(e);
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  auto launch = [&]() {
    const int nb256 = ((M + 255) / 256) * (N / 256), nb128 = ((M + 127) / 128) * (N / 128);
    const size_t gsm = (size_t)2 * (GBM + GBN) * GLD * 2;
    if (variant == 0) {
      if (epi == EPI_QKV) hipLaunchKernelGGL((gemm_kernel<Op, EPI_QKV>), dim3(nb128), dim3(256), gsm, st, a);
      else if (epi == EPI_GELU) hipLaunchKernelGGL((gemm_kernel<Op, EPI_GELU>), dim3(nb128), dim3(256), gsm, st, a);
      else hipLaunchKernelGGL((gemm_kernel<Op, EPI_RES>), dim3(nb128), dim3(256), gsm, st, a);
    } else if (variant >= 6 && variant <= 8) {
      if (variant == 6) hipLaunchKernelGGL((gemm256r_kernel<Op, EPI_QKV, 2, 1>), dim3(nb256), dim3(512), 131072, st, a);
      if (variant == 7) hipLaunchKernelGGL((gemm256r_kernel<Op, EPI_QKV, 2, 2>), dim3(nb256), dim3(512), 131072, st, a);
      if (variant == 8) hipLaunchKernelGGL((gemm256r_kernel<Op, EPI_QKV, 2, 3>), dim3(nb256), dim3(512), 131072, st, a);
    } else if (variant == 5) {
#define L256R(E) hipLaunchKernelGGL((gemm256r_kernel<Op, E>), dim3(nb256), dim3(512), 131072, st, a)
      if (epi == EPI_QKV) L256R(EPI_QKV); else if (epi == EPI_GELU) L256R(EPI_GELU); else L256R(EPI_RES);
#undef L256R
    } else if (variant == 9 || variant == 10) {
#define L256P(E) do { if (variant == 9) hipLaunchKernelGGL((gemm256p_kernel<Op, E, true>), dim3(nb256), dim3(512), 131072, st, a); \
                      else hipLaunchKernelGGL((gemm256p_kernel<Op, E, false>), dim3(nb256), dim3(512), 131072, st, a); } while (0)
      if (epi == EPI_QKV) L256P(EPI_QKV); else if (epi == EPI_GELU) L256P(EPI_GELU); else L256P(EPI_RES);
#undef L256P
    } else if (variant >= 11 && variant <= 15) {
      if (variant == 11) hipLaunchKernelGGL((gemm256p_kernel<Op, EPI_QKV, true, 1>), dim3(nb256), dim3(512), 131072, st, a);
      if (variant == 12) hipLaunchKernelGGL((gemm256p_kernel<Op, EPI_QKV, true, 2>), dim3(nb256), dim3(512), 131072, st, a);
      if (variant == 13) hipLaunchKernelGGL((gemm256p_kernel<Op, EPI_QKV, true, 4>), dim3(nb256), dim3(512), 131072, st, a);
      if (variant == 14) hipLaunchKernelGGL((gemm256p_kernel<Op, EPI_QKV, true, 8>), dim3(nb256), dim3(512), 131072, st, a);
      if (variant == 15) hipLaunchKernelGGL((gemm256p_kernel<Op, EPI_QKV, true, 16>), dim3(nb256), dim3(512), 131072, st, a);
    } else if (variant >= 16 && variant <= 21) {
      if (variant == 16) hipLaunchKernelGGL((gemm256p_kernel<Op, EPI_QKV, true, 28>), dim3(nb256), dim3(512), 131072, st, a);
      if (variant == 17) hipLaunchKernelGGL((gemm256p_kernel<Op, EPI_QKV, true, 20>), dim3(nb256), dim3(512), 131072, st, a);
      if (variant == 18) hipLaunchKernelGGL((gemm256p_kernel<Op, EPI_QKV, true, 12>), dim3(nb256), dim3(512), 131072, st, a);
      if (variant == 19) hipLaunchKernelGGL((gemm256p_kernel<Op, EPI_QKV, true, 24>), dim3(nb256), dim3(512), 131072, st, a);
      if (variant == 20) hipLaunchKernelGGL((gemm256p_kernel<Op, EPI_QKV, true, 21>), dim3(nb256), dim3(512), 131072, st, a);
      if (variant == 21) hipLaunchKernelGGL((gemm256p_kernel<Op, EPI_QKV, true, 32>), dim3(nb256), dim3(512), 131072, st, a);
    } else if (variant == 27) {
      const int grid = nb256 < 256 ? nb256 : 256;
#define L256PP(E) hipLaunchKernelGGL((gemm256p_kernel<Op, E, true, 0, true>), dim3(grid), dim3(512), 131072, st, a)
      if (epi == EPI_QKV) L256PP(EPI_QKV); else if (epi == EPI_GELU) L256PP(EPI_GELU); else L256PP(EPI_RES);
#undef L256PP
    } else if (variant >= 22 && variant <= 25) {
      if (variant == 22) hipLaunchKernelGGL((gemm256p_kernel<Op, EPI_QKV, true, 64>), dim3(nb256), dim3(512), 131072, st, a);
      if (variant == 23) hipLaunchKernelGGL((gemm256p_kernel<Op, EPI_QKV, true, 128>), dim3(nb256), dim3(512), 131072, st, a);
      if (variant == 24) hipLaunchKernelGGL((gemm256p_kernel<Op, EPI_GELU, true, 64>), dim3(nb256), dim3(512), 131072, st, a);
      if (variant == 25) hipLaunchKernelGGL((gemm256p_kernel<Op, EPI_RES, true, 64>), dim3(nb256), dim3(512), 131072, st, a);
    } else if (variant == 4) {
#define L256R4(E) hipLaunchKernelGGL((gemm256r_kernel<Op, E, 2, 0, 5>), dim3(nb256), dim3(512), 163840, st, a)
      if (epi == EPI_QKV) L256R4(EPI_QKV); else if (epi == EPI_GELU) L256R4(EPI_GELU); else L256R4(EPI_RES);
#undef L256R4
    }
  };
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm256r_kernel<Op, EPI_QKV>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm256r_kernel<Op, EPI_QKV, 2, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm256r_kernel<Op, EPI_QKV, 2, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm256r_kernel<Op, EPI_QKV, 2, 3>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm256r_kernel<Op, EPI_GELU>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm256r_kernel<Op, EPI_RES>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm256p_kernel<Op, EPI_QKV>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm256p_kernel<Op, EPI_QKV, true, 0, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm256p_kernel<Op, EPI_GELU, true, 0, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm256p_kernel<Op, EPI_RES, true, 0, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm256p_kernel<Op, EPI_GELU>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm256p_kernel<Op, EPI_RES>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm256p_kernel<Op, EPI_QKV, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm256p_kernel<Op, EPI_QKV, true, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm256p_kernel<Op, EPI_QKV, true, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm256p_kernel<Op, EPI_QKV, true, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm256p_kernel<Op, EPI_QKV, true, 8>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm256p_kernel<Op, EPI_QKV, true, 16>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm256p_kernel<Op, EPI_GELU, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm256p_kernel<Op, EPI_RES, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm256r_kernel<Op, EPI_QKV, 2, 0, 5>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm256r_kernel<Op, EPI_GELU, 2, 0, 5>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm256r_kernel<Op, EPI_RES, 2, 0, 5>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr = true;
  }
  launch();
  (void)hipEventRecord(e0, st);
  for (int i = 0; i < iters; ++i) launch();
  (void)hipEventRecord(e1, st);
  (void)hipEventSynchronize(e1);
  (void)hipEventElapsedTime(ms, e0, e1);
  *ms /= iters;
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  return hipGetLastError();
}

hipError_t launch_encoder(const Geom& g, int dtype, const EncWeights& w, const EncWorkspace& ws,
                          const uint8_t* images, float* tokens, int B, hipStream_t st, Profiler* prof, bool keep_cls) {
  if (dtype == 1) return run_encoder<OpBF16>(g, w, ws, images, tokens, B, st, prof, keep_cls);
  return run_encoder<OpF16>(g, w, ws, images, tokens, B, st, prof, keep_cls);
}

}  // namespace hvla
