// hypernet.hip — per-episode weight generation (reference: HyperNetwork.__call__,
// hypervla/components/hypernetwork.py:99-233; Transformer, transformer.py:127-262).
//
//   ctx_encoder_kernel  one workgroup per episode: token/image projections + position embeddings,
//                       the masked 6-layer context Transformer and the final scale, all in exact f32 with the
//                       34 x 128 token block resident in LDS: the dense layers and the attention on the
//                       matrix cores (v_mfma_f32_16x16x4_f32 multiplies and accumulates in f32), LayerNorm
//                       and softmax on the VALU (84 MFLOP/episode, once per episode -> latency, not
//                       throughput, is what matters here).
//   weightgen_kernel    the 73 output heads collapsed into ONE GEMM  theta = ctx @ W_cat + b_cat,
//                       computed transposed (theta^T tile = W_cat^T tile x ctx^T) with split-bf16
//                       MFMA (3 x v_mfma_f32_32x32x16_bf16, ~2^-16 relative) so that each lane ends up
//                       holding 16 consecutive packed positions of one episode and stores them
//                       straight into the policy kernel's arena layout (layout.h).  HBM-bound:
//                       reads W_cat once (hi+lo planes), writes the arena once; the ctx tiles reach the
//                       four waves of a workgroup through LDS (LDS-DMA, one tile ahead).
//   export_theta_kernel arena -> reference-order theta[B, G] (parity tests / `base_params` views).
#include "common.h"
#include "kernels.h"

namespace hvla {

// ------------------------------------------------------------------------------------------------
// context encoder
// ------------------------------------------------------------------------------------------------
constexpr int CTX_THREADS = 1024;   // 16 waves: one workgroup (one episode) per CU, so the waves of that one workgroup have to hide each other's latencies
constexpr int CTX_WAVES = CTX_THREADS / 64;
constexpr int CTX_RG = 20;   // S <= 2 * CTX_RG = 40 rows: three 16-row MFMA tiles at most

// y[s][n] = sum_k xs[s][k] * W[k][n]   for s in [0,S), n in [0,N); xs in LDS (row stride xs_ld, == 4 mod 64 words so that
// the 16 rows x 4 k of one A operand fall in 64 different banks), W global [K][N] (flax kernel layout), result handed to
// `sink(s, n, value)`.  Exact f32 on the matrix cores: v_mfma_f32_16x16x4_f32 multiplies and accumulates in f32 (lane l
// supplies A[i = l & 15][k = l >> 4] and B[k = l >> 4][j = l & 15] and receives C[4 (l >> 4) + r][l & 15], r = 0..3).
// One wave takes a 16-column tile of W for ALL row tiles (W is read once per workgroup, the B operand is reused for the two
// or three row tiles); when N has fewer than 8 column tiles for the 16 waves the k range is cut in two, the upper half's
// partial sums go through `part` (LDS, [S][part_ld]) and the lower half's waves add them: the order of the additions is
// fixed, so the bits are reproducible and independent of the batch.
// an opaque copy of the lane id: address arithmetic derived from it stays inside the block that uses it (hoisted out of the
// layer loop it becomes a kernel-long live range per call site and the allocator spills)
__device__ __forceinline__ int lane_here() {
  int v = threadIdx.x & 63;
  asm volatile("" : "+v"(v));
  return v;
}
struct Acc3 {
  f32x4 a0, a1, a2;   // rows 0-15, 16-31, 32-47 of one 16-column tile
};
__device__ __forceinline__ void acc3_zero(Acc3& a) {
  a.a0 = f32x4{0.f, 0.f, 0.f, 0.f};
  a.a1 = a.a0;
  a.a2 = a.a0;
}
// acc += xs[0:S, kx : kx + klen] @ Wk[0:klen, 16 tc : 16 tc + 16]   (Wk = the first of the klen rows of W; NT row tiles)
template <int NT>
__device__ __forceinline__ void mfma_cols_nt(const float* xs, int xs_ld, int S, int kx, const float* __restrict__ Wk, int N,
                                             int tc, int klen, Acc3& acc) {
  const int lane = lane_here(), c = lane & 15, kq = lane >> 4;
  const int r0 = c < S ? c : S - 1, r1 = 16 + c < S ? 16 + c : S - 1, r2 = 32 + c < S ? 32 + c : S - 1;   // clamped rows are never stored
  const float* wp = Wk + (size_t)kq * N + tc * 16 + c;
  const float* x0 = xs + r0 * xs_ld + kx + kq;
  const float* x1 = xs + r1 * xs_ld + kx + kq;
  const float* x2 = xs + r2 * xs_ld + kx + kq;
  auto step = [&](float xa, float xb, float xc, float bv) {
    acc.a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(xa, bv, acc.a0, 0, 0, 0);
    if (NT > 1) acc.a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(xb, bv, acc.a1, 0, 0, 0);
    if (NT > 2) acc.a2 = __builtin_amdgcn_mfma_f32_16x16x4f32(xc, bv, acc.a2, 0, 0, 0);
  };
  int k = 0;
  if (klen >= 32) {
    // eight k steps per batch; the next batch's W values are requested (L2) before this batch's x values (LDS), and all of a
    // batch's x values before its first MFMA: the 16 waves of the one workgroup per CU are all there is to hide the round trips
    float cur[8], nxt[8], xa[8], xb[8], xc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) cur[j] = wp[(size_t)(4 * j) * N];
    for (; k + 32 <= klen; k += 32) {
      const bool more = k + 64 <= klen;
      if (more) {
#pragma unroll
        for (int j = 0; j < 8; ++j) nxt[j] = wp[(size_t)(k + 32 + 4 * j) * N];
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        xa[j] = x0[k + 4 * j];
        if (NT > 1) xb[j] = x1[k + 4 * j];
        if (NT > 2) xc[j] = x2[k + 4 * j];
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) step(xa[j], xb[j], xc[j], cur[j]);
      if (more) {
#pragma unroll
        for (int j = 0; j < 8; ++j) cur[j] = nxt[j];
      }
    }
  }
  for (; k < klen; k += 4) step(x0[k], x1[k], x2[k], wp[(size_t)k * N]);
}
__device__ __forceinline__ void mfma_cols(const float* xs, int xs_ld, int S, int kx, const float* __restrict__ Wk, int N,
                                          int tc, int klen, Acc3& acc) {
  if (S > 32) mfma_cols_nt<3>(xs, xs_ld, S, kx, Wk, N, tc, klen, acc);
  else if (S > 16) mfma_cols_nt<2>(xs, xs_ld, S, kx, Wk, N, tc, klen, acc);
  else mfma_cols_nt<1>(xs, xs_ld, S, kx, Wk, N, tc, klen, acc);
}
// f(row, n, value, tile) for the rows < S of a finished tile
template <typename F>
__device__ __forceinline__ void acc3_rows(const Acc3& acc, int S, int tc, F f) {
  const int lane = lane_here(), c = lane & 15, kq = lane >> 4;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int row = 4 * kq + r, n = tc * 16 + c;
    if (row < S) f(row, n, acc.a0[r]);
    if (16 + row < S) f(16 + row, n, acc.a1[r]);
    if (32 + row < S) f(32 + row, n, acc.a2[r]);
  }
}
// the upper-k-half waves leave their sums in `part`, the lower-half waves add them and hand the total to the sink
template <typename Sink>
__device__ __forceinline__ void acc3_combine(const Acc3& acc, int S, int tc, int kh, float* part, int part_ld, Sink sink) {
  if (kh == 1) acc3_rows(acc, S, tc, [&](int row, int n, float v) { part[row * part_ld + n] = v; });
  __syncthreads();
  if (kh == 0) acc3_rows(acc, S, tc, [&](int row, int n, float v) { sink(row, n, v + part[row * part_ld + n]); });
}

// sink(row, n, value + bias[n]): the bias is requested with the tile's first W values, not after its last MFMA
template <typename Sink>
__device__ __forceinline__ void dense_mfma(const float* xs, int xs_ld, int S, int K, const float* __restrict__ W, int N,
                                           const float* __restrict__ bias, float* part, int part_ld, Sink sink) {
  const int wave = threadIdx.x >> 6;
  const int ntc = N >> 4;
  const bool split = part != nullptr && ntc * 2 <= CTX_WAVES && (K & 7) == 0;     // uniform over the workgroup
  if (!split) {
    for (int tc = wave; tc < ntc; tc += CTX_WAVES) {
      Acc3 acc;
      acc3_zero(acc);
      const float bn = bias[tc * 16 + (lane_here() & 15)];
      mfma_cols(xs, xs_ld, S, 0, W, N, tc, K, acc);
      acc3_rows(acc, S, tc, [&](int row, int n, float v) { sink(row, n, v + bn); });
    }
  } else {
    // ntc * 2 <= CTX_WAVES: one task per wave, its sums stay in registers across the barrier
    const int tc = wave % ntc, kh = wave < 2 * ntc ? wave / ntc : -1, klen = K >> 1;
    Acc3 acc;
    acc3_zero(acc);
    const float bn = bias[tc * 16 + (lane_here() & 15)];
    if (kh >= 0) mfma_cols(xs, xs_ld, S, kh * klen, W + (size_t)kh * klen * N, N, tc, klen, acc);
    acc3_combine(acc, S, tc, kh, part, part_ld, [&](int row, int n, float v) { sink(row, n, v + bn); });
  }
}

// flax LayerNorm (eps 1e-6, fast variance) over rows of xs -> ys; one wave per row.
__device__ __forceinline__ void ln_rows(const float* xs, float* ys, int ld, int S, int C,
                                        const float* __restrict__ scale, const float* __restrict__ bias) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, nw = CTX_THREADS >> 6;
  for (int s = wave; s < S; s += nw) {
    float sum = 0.f, sq = 0.f;
    for (int c = lane; c < C; c += 64) {
      const float v = xs[s * ld + c];
      sum += v;
      sq += v * v;
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
      sum += __shfl_xor(sum, o, 64);
      sq += __shfl_xor(sq, o, 64);
    }
    const float mean = sum / C;
    const float var = fmaxf(0.f, sq / C - mean * mean);
    const float rstd = rsqrtf(var + 1e-6f);
    for (int c = lane; c < C; c += 64) ys[s * ld + c] = (xs[s * ld + c] - mean) * rstd * scale[c] + bias[c];
  }
}

#ifdef HVLA_BENCH_HOOKS
// libhvla_bench.so only: shader-clock stamps of workgroup 0 at the phase boundaries (tools/ctx_phase_times.py)
__device__ unsigned long long g_ctx_stamps[64];
#define CTX_STAMP(i)                                                                    \
  do {                                                                                  \
    if (blockIdx.x == 0 && threadIdx.x == 0 && (i) < 64) g_ctx_stamps[(i)] = __builtin_amdgcn_s_memtime(); \
  } while (0)
#else
#define CTX_STAMP(i) do { } while (0)
#endif

__global__ __launch_bounds__(CTX_THREADS) void ctx_encoder_kernel(CtxParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int b = blockIdx.x;
  const int T = p.T, S = p.T + 2, C = p.C, F = p.F, Hc = p.heads, hc = C / Hc;
  const int ldx = C + 4, ldq = 3 * C + 4, ldf = F + 4;
  float* x = reinterpret_cast<float*>(smem);            // [S][ldx] residual stream
  float* h = x + S * ldx;                                // [S][ldx] LN output / attention output
  float* big = h + S * ldx;                              // [S][max(ldq, ldf)] qkv or mlp hidden
  CTX_STAMP(0);
  // ---- token projection: x[s] = tok[s] @ Wt + bt + pos_t[s]     (hypernetwork.py:112-115)
  {
    const float* tok = p.tok + (size_t)b * T * p.lang_dim;
    // the token rows go through `big` in K-chunks (row stride == 4 mod 64 words); every wave keeps its tile's sums over the chunks
    const int kc = T * 388 <= p.big_elems ? 384 : 128, ldt = kc + 4;
    const int wave = threadIdx.x >> 6, ntc = C >> 4;
    const bool split = ntc * 2 <= CTX_WAVES && (p.lang_dim & 7) == 0;
    const int tc = wave % ntc, kh = split ? (wave < 2 * ntc ? wave / ntc : -1) : (wave < ntc ? 0 : -1);
    Acc3 acc;
    acc3_zero(acc);
    for (int k0 = 0; k0 < p.lang_dim; k0 += kc) {
      const int kw = min(kc, p.lang_dim - k0);
      __syncthreads();
      for (int i = threadIdx.x; i < T * (kw >> 2); i += CTX_THREADS) {
        const int s = i / (kw >> 2), k = (i % (kw >> 2)) * 4;
        *reinterpret_cast<f32x4*>(big + s * ldt + k) = *reinterpret_cast<const f32x4*>(tok + (size_t)s * p.lang_dim + k0 + k);
      }
      __syncthreads();
      const int klen = split ? kw >> 1 : kw;
      if (kh >= 0) mfma_cols(big, ldt, T, kh * klen, p.w_tok + (size_t)(k0 + kh * klen) * C, C, tc, klen, acc);
    }
    auto put = [&](int row, int n, float v) { x[row * ldx + n] = v + p.b_tok[n] + p.pos_tok[row * C + n]; };
    if (split) acc3_combine(acc, T, tc, kh, h, ldx, put);
    else if (kh == 0) acc3_rows(acc, T, tc, put);
    __syncthreads();
    CTX_STAMP(1);
    // ---- initial-image CLS projection (hypernetwork.py:118-128) and layer token (:144-145): the E products of a column are
    // summed in CTX_THREADS / C parts, then the parts in order
    const float* cls = p.cls + (size_t)b * p.E;
    float* parts = big + p.E;
    for (int i = threadIdx.x; i < p.E; i += CTX_THREADS) big[i] = cls[i];
    __syncthreads();
    {
      const int np = CTX_THREADS / C, kper = (p.E + np - 1) / np;
      const int c = threadIdx.x % C, pi = threadIdx.x / C;
      const int kb = pi * kper, ke = min(p.E, kb + kper);
      float a = 0.f;
#pragma unroll 8
      for (int k = kb; k < ke; ++k) a = fmaf(big[k], p.w_img[(size_t)k * C + c], a);
      parts[pi * C + c] = a;
      __syncthreads();
      if (threadIdx.x < C) {
        float t = 0.f;
        for (int i = 0; i < np; ++i) t += parts[i * C + c];
        x[T * ldx + c] = t + p.b_img[c] + p.pos_img[c];
        x[(T + 1) * ldx + c] = p.pos_layer[c];
      }
    }
    __syncthreads();
  }
  CTX_STAMP(2);
  int* kmask = reinterpret_cast<int*>(big + p.big_elems);          // [T] language-token mask (read in the first layer after several barriers)
  for (int k = threadIdx.x; k < T; k += CTX_THREADS) kmask[k] = p.attn_mask[(size_t)b * T + k] != 0;
  // ---- context Transformer (transformer.py:127-262)
  for (int l = 0; l < p.layers; ++l) {
    const CtxLayer& w = p.layer[l];
    ln_rows(x, h, ldx, S, C, w.ln0_s, w.ln0_b);
    __syncthreads();
    CTX_STAMP(3 + 7 * l);
    // q | k | v = h @ W{q,k,v} + b    (flax DenseGeneral kernel [C, H, hd] == [C, C] row-major)
    for (int task = threadIdx.x >> 6; task < 3 * (C >> 4); task += CTX_WAVES) {     // 24 column tiles over 16 waves: 6 per SIMD
      const int which = task / (C >> 4), tc = task % (C >> 4);
      const float* W = which == 0 ? w.wq : which == 1 ? w.wk : w.wv;
      const float* B = which == 0 ? w.bq : which == 1 ? w.bk : w.bv;
      const float sc = which == 0 ? rsqrtf((float)hc) : 1.f;     // query pre-scaling
      Acc3 acc;
      acc3_zero(acc);
      const float bn = B[tc * 16 + (lane_here() & 15)];
      mfma_cols(h, ldx, S, 0, W, C, tc, C, acc);
      acc3_rows(acc, S, tc, [&](int s, int n, float v) { big[s * ldq + which * C + n] = (v + bn) * sc; });
    }
    __syncthreads();
    CTX_STAMP(4 + 7 * l);
    // attention on the matrix cores, transposed: one wave per (head, 16-query tile).  S^T = K Q^T leaves lane (query = l & 15,
    // kq = l >> 4) with the logits of keys 16 kt + 4 kq + r of its query -- exactly the B operand P^T[key slot][query] that
    // O^T = V^T P^T wants when MFMA k slot kq of step (kt, r) is read as key 16 kt + 4 kq + r, so the probabilities never leave
    // their registers; the softmax reduces in-lane and over the four lanes of a query (xor 16, 32).
    {
      const int nqt = (S + 15) >> 4;
      for (int task = threadIdx.x >> 6; task < Hc * nqt; task += CTX_WAVES) {
        const int hh = task / nqt, qt = task % nqt;
        const int lane = lane_here(), j = lane & 15, kq = lane >> 4;
        const int q = qt * 16 + j, qr = q < S ? q : S - 1;
        const float* qp = big + qr * ldq + hh * hc + kq;
        f32x4 sc[3];
#pragma unroll
        for (int kt = 0; kt < 3; ++kt) {
          sc[kt] = f32x4{0.f, 0.f, 0.f, 0.f};
          if (kt < nqt) {
            const int kr = kt * 16 + j < S ? kt * 16 + j : S - 1;
            const float* kp = big + kr * ldq + C + hh * hc + kq;
            for (int d = 0; d < hc; d += 4) sc[kt] = __builtin_amdgcn_mfma_f32_16x16x4f32(kp[d], qp[d], sc[kt], 0, 0, 0);
          }
        }
        float mx = -3.4028234663852886e38f;        // finfo(float32).min stands in for a masked logit (transformer.py)
#pragma unroll
        for (int kt = 0; kt < 3; ++kt)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int key = kt * 16 + 4 * kq + r;
            const bool keep = key < S && (key < T ? kmask[key] != 0 : (key == T ? true : q == T + 1));
            sc[kt][r] = keep ? sc[kt][r] : -3.4028234663852886e38f;
            mx = fmaxf(mx, sc[kt][r]);
          }
        mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        float den = 0.f;
#pragma unroll
        for (int kt = 0; kt < 3; ++kt)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int key = kt * 16 + 4 * kq + r;
            const float e = key < S ? __expf(sc[kt][r] - mx) : 0.f;     // a masked key of the sequence gives exp(min - mx) = 0
            sc[kt][r] = e;
            den += e;
          }
        den += __shfl_xor(den, 16, 64);
        den += __shfl_xor(den, 32, 64);
        const float inv = 1.f / den;
        for (int dt = 0; dt * 16 < hc; ++dt) {
          const int dd = dt * 16 + j < hc ? dt * 16 + j : hc - 1;
          const float* vp = big + 2 * C + hh * hc + dd;
          f32x4 o = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int kt = 0; kt < 3; ++kt)
#pragma unroll
            for (int r = 0; r < 4; ++r)
              if (kt * 16 + r < S) {                 // uniform: the step's smallest key (kq = 0)
                const int key = kt * 16 + 4 * kq + r < S ? kt * 16 + 4 * kq + r : S - 1;     // past the end: probability 0 times a real row
                o = __builtin_amdgcn_mfma_f32_16x16x4f32(vp[key * ldq], sc[kt][r], o, 0, 0, 0);
              }
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int d = dt * 16 + 4 * kq + r;
            if (d < hc && q < S) h[q * ldx + hh * hc + d] = o[r] * inv;
          }
        }
      }
    }
    __syncthreads();
    CTX_STAMP(5 + 7 * l);
    // out projection + residual   (kernel [H, hd, C] == [C, C] row-major)
    dense_mfma(h, ldx, S, C, w.wo, C, w.bo, big, ldq, [&](int s, int n, float v) { x[s * ldx + n] += v; });     // q|k|v are dead: `big` takes the partial sums
    __syncthreads();
    CTX_STAMP(6 + 7 * l);
    ln_rows(x, h, ldx, S, C, w.ln1_s, w.ln1_b);
    __syncthreads();
    CTX_STAMP(7 + 7 * l);
    dense_mfma(h, ldx, S, C, w.w1, F, w.b1, nullptr, 0, [&](int s, int n, float v) { big[s * ldf + n] = gelu_tanh(v); });
    __syncthreads();
    CTX_STAMP(8 + 7 * l);
    dense_mfma(big, ldf, S, F, w.w2, C, w.b2, h, ldx, [&](int s, int n, float v) { x[s * ldx + n] += v; });      // h (LN output) is dead
    __syncthreads();
    CTX_STAMP(9 + 7 * l);
  }
  // ---- encoder_norm on the layer-token row, scale, publish (hypernetwork.py:188-192)
  if (threadIdx.x < 64) {
    const int lane = threadIdx.x;
    const float* xr = x + (S - 1) * ldx;
    float sum = 0.f, sq = 0.f;
    for (int c = lane; c < C; c += 64) {
      sum += xr[c];
      sq += xr[c] * xr[c];
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
      sum += __shfl_xor(sum, o, 64);
      sq += __shfl_xor(sq, o, 64);
    }
    const float mean = sum / C, var = fmaxf(0.f, sq / C - mean * mean), rstd = rsqrtf(var + 1e-6f);
    const float post = p.scale_context ? rsqrtf((float)C) : 1.f;
    for (int c = lane; c < C; c += 64) {
      const float v = ((xr[c] - mean) * rstd * p.norm_s[c] + p.norm_b[c]) * post;
      p.ctx[(size_t)b * C + c] = v;
      __bf16 hi, lo;
      split1(v, hi, lo);
      p.ctx_hi[(size_t)b * C + c] = hi;
      p.ctx_lo[(size_t)b * C + c] = lo;
    }
  }
}

// ------------------------------------------------------------------------------------------------
// weight generation GEMM
// ------------------------------------------------------------------------------------------------
// grid.x = ceil(ntiles / 4); each wave owns one 32-position tile and loops over all episode tiles.
// A fragments (W_cat^T, hi/lo) live in registers for the whole loop (KS * 2 * 4 VGPRs).
// Stores: a lane ends up with 16 consecutive packed positions of ONE episode, so written straight from registers a store
// instruction is 64 pieces of 16 bytes in 32 different rows of the arena (rows are 356 KB apart).  Workgroups whose four
// tiles all lie in the matrix region therefore go through LDS: the four waves park their [32 episodes x 32 positions] hi and
// lo tiles side by side ([plane][episode][128 positions] = 16 KB), and every wave then writes eight episode rows as 256-byte
// runs (16 lanes x 16 B per row).  The vector region (f32, 10 % of the bytes) and a ragged last workgroup keep the direct form.
// Round 3: the B operand (the 32 episodes' ctx rows, hi and lo) no longer comes from global memory per wave and k-step -- 16
// loads per wave and episode tile whose 64 lanes pick 16-byte pieces out of 32 different rows, four waves fetching the same
// rows: that was most of what the CU's vector memory pipe did, and every iteration began with their round trip -- but through
// LDS: the tile's 32 rows are contiguous in memory, so the workgroup stages them with KS / 2 LDS-DMA instructions per wave
// (1 KB each, linear; the 16-byte chunk index XORed with the row on the SOURCE address so that the ds_read_b128 of 16 rows at
// one k chunk spread over all banks), one tile ahead (double buffer), and the four waves read their fragments from there.
template <int KS>
__global__ __launch_bounds__(256, 3) void weightgen_kernel(WeightGenParams p) {      // 48 KB of LDS: three workgroups per CU
  constexpr int CH = 2 * KS;                       // 16-byte chunks per ctx row (C = 16 KS bf16)
  constexpr int IPW = KS / 2 > 0 ? KS / 2 : 1;     // DMA instructions per wave and episode tile (2 planes x 32 rows x CH chunks / 64 lanes / 4 waves)
  __shared__ __attribute__((aligned(16))) __bf16 stage[2][32][128];
  __shared__ __attribute__((aligned(16))) __bf16 cbuf[2][2][32 * KS * 16];      // [buffer][hi / lo][row][chunk ^ row][8]
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int wg = blockIdx.x;
  const bool active = wg * 4 + wave < p.ntiles;      // a ragged last workgroup: its spare waves still stage and go to the barriers
  const int tile = active ? wg * 4 + wave : p.ntiles - 1;
  const bool via_lds = wg * 4 + 3 < p.ntiles && (wg * 4 + 4) * 32 <= p.Gm;     // workgroup-uniform
  const int col = lane & 31, half = lane >> 5;
  const uint32_t lds_c = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)&cbuf[0][0][0];
  auto stage_ctx = [&](int b0, int buf) {
#pragma unroll
    for (int u = 0; u < IPW; ++u) {
      const int j = wave * IPW + u;                  // instruction of the tile: plane j / KS, 1 KB piece j % KS
      if (KS < 2 && j >= 2 * KS) break;
      const int plane = j / KS, piece = j % KS;
      const int pos = piece * 64 + lane;             // 16-byte position inside the plane's tile
      const int r = pos / CH, c = pos % CH;
      int b = b0 + r;
      b = b < p.B ? b : p.B - 1;
      const __bf16* src = (plane ? p.ctx_lo : p.ctx_hi) + (size_t)b * (KS * 16) + ((c ^ (r & (CH - 1))) * 8);
      const uint32_t dst = __builtin_amdgcn_readfirstlane(lds_c + (uint32_t)(((buf * 2 + plane) * 32 * KS * 16) * 2 + piece * 1024));
      uint32_t keep;
      asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                   : "=&s"(keep) : "v"(src), "s"(dst));
    }
  };
  stage_ctx(0, 0);
  bf16x8 ah[KS], al[KS];
  {
    const bf16x8* Ah = reinterpret_cast<const bf16x8*>(p.wcat_hi) + ((size_t)tile * KS) * 64 + lane;
    const bf16x8* Al = reinterpret_cast<const bf16x8*>(p.wcat_lo) + ((size_t)tile * KS) * 64 + lane;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      ah[ks] = __builtin_nontemporal_load(Ah + ks * 64);
      al[ks] = __builtin_nontemporal_load(Al + ks * 64);
    }
  }
  // this lane's 16 consecutive packed positions and their bias
  const int pos0 = tile * 32 + half * 16;
  float bias[16];
#pragma unroll
  for (int r = 0; r < 16; r += 4) {
    const f32x4 v = *reinterpret_cast<const f32x4*>(p.bcat + pos0 + r);
    bias[r] = v[0], bias[r + 1] = v[1], bias[r + 2] = v[2], bias[r + 3] = v[3];
  }
  int it = 0;
  for (int b0 = 0; b0 < p.B; b0 += 32, ++it) {
    const int b = b0 + col;
    const bool more = b0 + 32 < p.B;
    // the other buffer was last read two barriers ago (every iteration has at least one behind its MFMAs)
    if (more) {
      stage_ctx(b0 + 32, (it + 1) & 1);
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(IPW) : "memory");     // this tile has landed (the A fragments and the last stores are older still)
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();                               // ... for every wave's share of it
    const __bf16* ch = &cbuf[it & 1][0][0] + col * (KS * 16);
    const __bf16* cl = &cbuf[it & 1][1][0] + col * (KS * 16);
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = bias[r];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const int sw = ((ks * 2 + half) ^ (col & (CH - 1))) * 8;
      const bf16x8 bh = *reinterpret_cast<const bf16x8*>(ch + sw), bl = *reinterpret_cast<const bf16x8*>(cl + sw);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[ks], bh, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[ks], bl, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[ks], bh, acc, 0, 0, 0);
    }
    if (via_lds) {
      bf16x8 h0, h1, l0, l1;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        __bf16 hi, lo;
        split1(acc[j], hi, lo);
        h0[j] = hi, l0[j] = lo;
        split1(acc[8 + j], hi, lo);
        h1[j] = hi, l1[j] = lo;
      }
      __syncthreads();                           // the previous episode tile's rows have been read
      // A row is 256 B = twice the 32 write banks, and the eight lanes of a ds_write_b128 group hold eight consecutive rows at ONE
      // position: eight-way conflicts on every write (65 % of this kernel's LDS cycles, profiles/r3_pmc_sq_by_kernel.csv).  The
      // 16-byte chunk index is XORed with the row, here and where the rows are read back: both sides are conflict-free.
      const int c0 = wave * 4 + half * 2, sx = col & 15;
      *reinterpret_cast<bf16x8*>(&stage[0][col][((c0 + 0) ^ sx) * 8]) = h0;
      *reinterpret_cast<bf16x8*>(&stage[0][col][((c0 + 1) ^ sx) * 8]) = h1;
      *reinterpret_cast<bf16x8*>(&stage[1][col][((c0 + 0) ^ sx) * 8]) = l0;
      *reinterpret_cast<bf16x8*>(&stage[1][col][((c0 + 1) ^ sx) * 8]) = l1;
      __syncthreads();
      const size_t gpos = (size_t)wg * 128 + (lane & 15) * 8;          // 16 lanes x 16 B = one episode's 256-byte run
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int row = wave * 8 + u * 4 + (lane >> 4);
        if (b0 + row < p.B) {
          const bf16x8 vh = *reinterpret_cast<const bf16x8*>(&stage[0][row][((lane & 15) ^ (row & 15)) * 8]);
          const bf16x8 vl = *reinterpret_cast<const bf16x8*>(&stage[1][row][((lane & 15) ^ (row & 15)) * 8]);
          *reinterpret_cast<bf16x8*>(p.wh + (size_t)(b0 + row) * p.Gm + gpos) = vh;
          *reinterpret_cast<bf16x8*>(p.wl + (size_t)(b0 + row) * p.Gm + gpos) = vl;
        }
      }
    } else {
      if (active && b < p.B) {
        if (pos0 < p.Gm) {
          bf16x8 h0, h1, l0, l1;
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            __bf16 hi, lo;
            split1(acc[j], hi, lo);
            h0[j] = hi, l0[j] = lo;
            split1(acc[8 + j], hi, lo);
            h1[j] = hi, l1[j] = lo;
          }
          bf16x8* dh = reinterpret_cast<bf16x8*>(p.wh + (size_t)b * p.Gm + pos0);
          bf16x8* dl = reinterpret_cast<bf16x8*>(p.wl + (size_t)b * p.Gm + pos0);
          dh[0] = h0, dh[1] = h1, dl[0] = l0, dl[1] = l1;
        } else {
          f32x4* dv = reinterpret_cast<f32x4*>(p.vf + (size_t)b * p.Gv + (pos0 - p.Gm));
#pragma unroll
          for (int r = 0; r < 16; r += 4) dv[r >> 2] = f32x4{acc[r], acc[r + 1], acc[r + 2], acc[r + 3]};
        }
      }
      __syncthreads();                             // (the via_lds form has its two: here one, so that nobody stages over a tile still being read)
    }
  }
}

__global__ void export_theta_kernel(const __bf16* wh, const __bf16* wl, const float* vf, const int32_t* perm,
                                    int Gm, int Gv, int G, int B, float* theta) {
  const int b = blockIdx.y;
  for (int pos = blockIdx.x * blockDim.x + threadIdx.x; pos < Gm + Gv; pos += gridDim.x * blockDim.x) {
    const int ref = perm[pos];
    if (ref < 0) continue;
    float v;
    if (pos < Gm)
      v = (float)wh[(size_t)b * Gm + pos] + (float)wl[(size_t)b * Gm + pos];
    else
      v = vf[(size_t)b * Gv + pos - Gm];
    theta[(size_t)b * G + ref] = v;
  }
}

#ifdef HVLA_BENCH_HOOKS
hipError_t debug_ctx_stamps(unsigned long long* out) { return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_ctx_stamps), sizeof(unsigned long long) * 64); }
#endif

// ---- launchers -----------------------------------------------------------------------------------
// LDS the context encoder needs for a geometry (T task tokens, width C, MLP width F, image width E): hvla_create refuses a
// configuration that does not fit instead of letting the first hvla_generate fail with a bare HIP error
static size_t ctx_big_elems(int T, int C, int F, int E) {
  const int S = T + 2;
  const int ldq = 3 * C + 4, ldf = F + 4;
  int bigld = ldq > ldf ? ldq : ldf;
  if (bigld < 132) bigld = 132;
  size_t big_elems = (size_t)S * bigld;
  if (big_elems < (size_t)E + CTX_THREADS) big_elems = (size_t)E + CTX_THREADS;
  return big_elems;
}
size_t ctx_encoder_lds_bytes(int T, int C, int F, int E) {
  return ((size_t)2 * (T + 2) * (C + 4) + ctx_big_elems(T, C, F, E) + 64) * sizeof(float);
}

hipError_t launch_ctx_encoder(const CtxParams& p, int B, hipStream_t st) {
  const int S = p.T + 2;
  const int ldx = p.C + 4;
  const size_t big_elems = ctx_big_elems(p.T, p.C, p.F, p.E);
  CtxParams q = p;
  q.big_elems = (int)big_elems;
  const size_t smem = ((size_t)2 * S * ldx + big_elems + 64) * sizeof(float);      // + the key mask (= ctx_encoder_lds_bytes)
  static bool attr_done[64] = {};                              // per device: the attribute belongs to the device's code object
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return e;
  if (dev < 0 || dev >= 64) return hipErrorInvalidDevice;
  if (!attr_done[dev]) {
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(ctx_encoder_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return e;
    attr_done[dev] = true;
  }
  if (smem > 160 * 1024) return hipErrorInvalidValue;
  hipLaunchKernelGGL(ctx_encoder_kernel, dim3(B), dim3(CTX_THREADS), smem, st, q);
  return hipGetLastError();
}

hipError_t launch_weightgen(const WeightGenParams& p, int C, hipStream_t st) {
  const int blocks = (p.ntiles + 3) / 4;
  if (C == 128)
    hipLaunchKernelGGL(weightgen_kernel<8>, dim3(blocks), dim3(256), 0, st, p);
  else if (C == 64)
    hipLaunchKernelGGL(weightgen_kernel<4>, dim3(blocks), dim3(256), 0, st, p);
  else if (C == 32)
    hipLaunchKernelGGL(weightgen_kernel<2>, dim3(blocks), dim3(256), 0, st, p);
  else
    return hipErrorInvalidValue;
  return hipGetLastError();
}

hipError_t launch_export_theta(const __bf16* wh, const __bf16* wl, const float* vf, const int32_t* perm,
                               int Gm, int Gv, int G, int B, float* theta, hipStream_t st) {
  hipLaunchKernelGGL(export_theta_kernel, dim3(128, B), dim3(256), 0, st, wh, wl, vf, perm, Gm, Gv, G, B, theta);
  return hipGetLastError();
}

}  // namespace hvla
