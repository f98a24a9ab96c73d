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
//                      block; K and V^T resident in LDS; transposed scores (keys on accumulator rows,
//                      query on the lane) so softmax is in-lane and P feeds the PV MFMA from registers.
//
// Residual stream, LayerNorm statistics, softmax and GELU are f32; only MFMA operands are 16-bit.
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
    typename Op::x8 v;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      int k = ch * 8 + j;
      k = k >= Kp1 ? k - Kp1 : k;
      float f = 0.f;
      if (k < kreal) {
        const int c = k % 3, dx = (k / 3) % patch, dy = k / (3 * patch);
        f = (float)img[((b * image + (size_t)(py * patch + dy)) * image + (px * patch + dx)) * 3 + c] - 128.f;
      }
      v[j] = (typename Op::elem)f;
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
};

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
    wp[i] = W + (size_t)(n0 + lrow + 32 * i) * g.K + lch * 8;
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

  f32x4 acc[4][4];   // [n-tile][m-tile]: D = W-tile(16 n rows) x A-tile^T (16 m cols)
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
        for (int mt = 0; mt < 4; ++mt) acc[nt][mt] = Op::mma16(fw[nt], fa[mt], acc[nt][mt]);
    }
    if (kt + 1 < KT) sstore(buf ^ 1);
    __syncthreads();
  }

  // epilogue: lane holds, per (nt, mt), column m = m0+64wm+16mt+fr and rows n = n0+64wn+16nt+4fq+{0..3}
#pragma unroll
  for (int mt = 0; mt < 4; ++mt) {
    const int m = m0 + wm * 64 + mt * 16 + fr;
    if (m >= g.M) continue;
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
      const int n = n0 + wn * 64 + nt * 16 + fq * 4;
      const f32x4 b4 = *reinterpret_cast<const f32x4*>(g.bias + n);
      f32x4 v = acc[nt][mt];
      if constexpr (EPI == EPI_PATCH) {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] *= g.qscale;     // patch weights are stored x256 (16-bit range)
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] += b4[r];
      if constexpr (EPI == EPI_QKV) {
        if (n < g.qcols) {
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] *= g.qscale;
        }
        typename Op::x4 o;
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r] = (T)v[r];
        *reinterpret_cast<typename Op::x4*>(reinterpret_cast<T*>(g.out) + (size_t)m * g.N + n) = o;
      } else if constexpr (EPI == EPI_GELU) {
        typename Op::x4 o;
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r] = (T)gelu_erf(v[r]);
        *reinterpret_cast<typename Op::x4*>(reinterpret_cast<T*>(g.out) + (size_t)m * g.N + n) = o;
      } else if constexpr (EPI == EPI_RES) {
        const f32x4 ls = *reinterpret_cast<const f32x4*>(g.aux + n);
        f32x4* xp = reinterpret_cast<f32x4*>(reinterpret_cast<float*>(g.out) + (size_t)m * g.N + n);
        f32x4 x = *xp;
#pragma unroll
        for (int r = 0; r < 4; ++r) x[r] = fmaf(v[r], ls[r], x[r]);
        *xp = x;
      } else {  // EPI_PATCH
        const int b = m / g.P, p = m % g.P;
        const f32x4 pe = *reinterpret_cast<const f32x4*>(g.aux + (size_t)(1 + p) * g.N + n);
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] += pe[r];
        *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(g.out) + ((size_t)b * g.S + 1 + p) * g.N + n) = v;
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// LayerNorm: one wave per row of E f32 (E % 4 == 0, E <= 1024); FINAL drops row 0 of every image and
// writes f32.
template <typename Op, bool FINAL>
__global__ __launch_bounds__(256) void layernorm_kernel(const float* __restrict__ x, void* __restrict__ out,
                                                        const float* __restrict__ scale,
                                                        const float* __restrict__ bias, int M, int E, int S) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= M) return;
  if (FINAL && (row % S) == 0) return;
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
  if (FINAL) orow = (size_t)(row / S) * (S - 1) + (row % S) - 1;
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
// Attention, head_dim 64.  qkv [B*S][3E] 16-bit (q already scaled by 1/sqrt(64)); out o [B*S][E].
constexpr int AKLD = 72;          // K row stride in LDS (halves): 144 B
template <typename Op>
__global__ void attention_kernel(const typename Op::elem* __restrict__ qkv, typename Op::elem* __restrict__ o,
                                 int S, int E, int H) {
  using T = typename Op::elem;
  using X8 = typename Op::x8;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int KT = (S + 31) / 32, SP = KT * 32;
  const int VLD = SP + 8;                                 // V^T row stride (halves)
  T* Ks = reinterpret_cast<T*>(smem);                      // [SP][AKLD]
  T* Vt = Ks + SP * AKLD;                                  // [64][VLD]
  const int b = blockIdx.x / H, head = blockIdx.x % H;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, nthr = blockDim.x;
  const size_t rowstride = (size_t)3 * E;
  const T* base = qkv + (size_t)b * S * rowstride + head * 64;
  // ---- stage K (row-major) and V^T (keys permuted: bits 2 and 3 of key&31 swapped)
  for (int i = tid; i < SP * 8; i += nthr) {
    const int key = i >> 3, ch = i & 7;
    X8 kv, vv;
    if (key < S) {
      kv = *reinterpret_cast<const X8*>(base + (size_t)key * rowstride + E + ch * 8);
      vv = *reinterpret_cast<const X8*>(base + (size_t)key * rowstride + 2 * E + ch * 8);
    } else {
#pragma unroll
      for (int j = 0; j < 8; ++j) kv[j] = (T)0.f, vv[j] = (T)0.f;
    }
    *reinterpret_cast<X8*>(Ks + key * AKLD + ch * 8) = kv;
    const int kl = key & 31;
    const int vpos = (key & ~31) | (kl & 0x13) | ((kl & 4) << 1) | ((kl & 8) >> 1);
#pragma unroll
    for (int j = 0; j < 8; ++j) Vt[(ch * 8 + j) * VLD + vpos] = vv[j];
  }
  // ---- this wave's 32 queries as B fragments (natural d order): 4 k-steps of 16
  const int col = lane & 31, half = lane >> 5;
  int q = wave * 32 + col;
  const bool qvalid = q < S;
  q = qvalid ? q : S - 1;
  X8 qf[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) qf[ks] = *reinterpret_cast<const X8*>(base + (size_t)q * rowstride + ks * 16 + half * 8);
  __syncthreads();

  const float LOG2E = 1.4426950408889634f;
  float m2 = -1e30f, lsum = 0.f;     // running max (log2 domain), this half's partial denominator
  f32x16 O[2];
#pragma unroll
  for (int r = 0; r < 16; ++r) O[0][r] = 0.f, O[1][r] = 0.f;
  for (int kt = 0; kt < KT; ++kt) {
    f32x16 s;
#pragma unroll
    for (int r = 0; r < 16; ++r) s[r] = 0.f;
    const T* kp = Ks + (kt * 32 + col) * AKLD + half * 8;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) s = Op::mma32(*reinterpret_cast<const X8*>(kp + ks * 16), qf[ks], s);
    float tmax = -1e30f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      float v = s[r] * LOG2E;
      if (kt == KT - 1 && kt * 32 + crow(r, half) >= S) v = -1e30f;
      s[r] = v;
      tmax = fmaxf(tmax, v);
    }
    tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
    const float mn = fmaxf(m2, tmax);
    const float alpha = __builtin_amdgcn_exp2f(m2 - mn);
    m2 = mn;
    float psum = 0.f;
    X8 pf[2];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float p = __builtin_amdgcn_exp2f(s[r] - mn);
      psum += p;
      pf[r >> 3][r & 7] = (T)p;
    }
    lsum = lsum * alpha + psum;
#pragma unroll
    for (int r = 0; r < 16; ++r) O[0][r] *= alpha, O[1][r] *= alpha;
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
      const T* vp = Vt + (mt * 32 + col) * VLD + kt * 32 + half * 8;
#pragma unroll
      for (int sstep = 0; sstep < 2; ++sstep)
        O[mt] = Op::mma32(*reinterpret_cast<const X8*>(vp + sstep * 16), pf[sstep], O[mt]);
    }
  }
  const float inv = 1.f / (lsum + __shfl_xor(lsum, 32, 64));
  if (qvalid) {
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
}

// ------------------------------------------------------------------------------------------------
template <typename Op>
static hipError_t run_encoder(const Geom& g, const EncWeights& w, const EncWorkspace& ws, const uint8_t* images,
                              float* tokens, int B, hipStream_t st, Profiler* prof) {
  Profiler none;
  Profiler& pf = prof ? *prof : none;
  using T = typename Op::elem;
  const int P = g.P(), S = g.S(), E = g.E, F = g.enc_mlp, H = g.enc_heads;
  const int Kp = 2 * ((g.patch * g.patch * 3 + 63) / 64 * 64);   // [a | a] x [W_hi | W_lo]
  const int M = B * S;
  const size_t gsm = (size_t)2 * (GBM + GBN) * GLD * sizeof(T);
  static bool attr = false;
  if (!attr) {
    hipError_t e;
#define SETA(K) \
  if ((e = hipFuncSetAttribute(reinterpret_cast<const void*>(K), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)) != hipSuccess) return e;
    SETA((gemm_kernel<Op, EPI_PATCH>)) SETA((gemm_kernel<Op, EPI_QKV>)) SETA((gemm_kernel<Op, EPI_GELU>))
    SETA((gemm_kernel<Op, EPI_RES>)) SETA((attention_kernel<Op>))
#undef SETA
    attr = true;
  }
  auto gemm = [&](auto kern, const void* A, const void* Wt, int Mm, int N, int K, const float* bias, const float* aux,
                  void* out, int qcols) {
    GemmArgs a{A, Wt, Mm, N, K, bias, aux, out, P, S, qcols, qcols ? 0.125f : 1.f / 256.f};
    const int nb = ((Mm + GBM - 1) / GBM) * (N / GBN);
    hipLaunchKernelGGL(kern, dim3(nb), dim3(256), gsm, st, a);
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
    gemm(gemm_kernel<Op, EPI_PATCH>, ws.g, w.w_patch, B * P, E, Kp, w.b_patch, w.pos, ws.x, 0);
  }
  pf.end(0, st);
  const int KT = (S + 31) / 32;
  const size_t asm_bytes = ((size_t)KT * 32 * AKLD + (size_t)64 * (KT * 32 + 8)) * sizeof(T);
  for (int l = 0; l < g.enc_layers; ++l) {
    const EncLayerW& L = w.layer[l];
    pf.begin(1, st);
    hipLaunchKernelGGL((layernorm_kernel<Op, false>), dim3((M + 3) / 4), dim3(256), 0, st, ws.x, ws.h, L.ln1_s,
                       L.ln1_b, M, E, S);
    pf.end(1, st);
    pf.begin(2, st);
    gemm(gemm_kernel<Op, EPI_QKV>, ws.h, L.wqkv, M, 3 * E, E, L.bqkv, nullptr, ws.qkv, E);
    pf.end(2, st);
    pf.begin(3, st);
    hipLaunchKernelGGL(attention_kernel<Op>, dim3(B * H), dim3(KT * 64), asm_bytes, st,
                       reinterpret_cast<const T*>(ws.qkv), reinterpret_cast<T*>(ws.h), S, E, H);
    pf.end(3, st);
    pf.begin(4, st);
    gemm(gemm_kernel<Op, EPI_RES>, ws.h, L.wo, M, E, E, L.bo, L.ls1, ws.x, 0);
    pf.end(4, st);
    pf.begin(1, st);
    hipLaunchKernelGGL((layernorm_kernel<Op, false>), dim3((M + 3) / 4), dim3(256), 0, st, ws.x, ws.h, L.ln2_s,
                       L.ln2_b, M, E, S);
    pf.end(1, st);
    pf.begin(5, st);
    gemm(gemm_kernel<Op, EPI_GELU>, ws.h, L.w1, M, F, E, L.b1, nullptr, ws.g, 0);
    pf.end(5, st);
    pf.begin(6, st);
    gemm(gemm_kernel<Op, EPI_RES>, ws.g, L.w2, M, E, F, L.b2, L.ls2, ws.x, 0);
    pf.end(6, st);
  }
  pf.begin(1, st);
  hipLaunchKernelGGL((layernorm_kernel<Op, true>), dim3((M + 3) / 4), dim3(256), 0, st, ws.x, tokens, w.lnf_s,
                     w.lnf_b, M, E, S);
  pf.end(1, st);
  return hipGetLastError();
}

hipError_t launch_encoder(const Geom& g, int dtype, const EncWeights& w, const EncWorkspace& ws,
                          const uint8_t* images, float* tokens, int B, hipStream_t st, Profiler* prof) {
  if (dtype == 1) return run_encoder<OpBF16>(g, w, ws, images, tokens, B, st, prof);
  return run_encoder<OpF16>(g, w, ws, images, tokens, B, st, prof);
}

}  // namespace hvla
