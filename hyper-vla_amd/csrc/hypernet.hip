// hypernet.hip — per-episode weight generation (reference: HyperNetwork.__call__,
// hypervla/components/hypernetwork.py:99-233; Transformer, transformer.py:127-262).
//
//   ctx_encoder_kernel  one workgroup per episode: token/image projections + position embeddings,
//                       the masked 6-layer context Transformer and the final scale, all in exact f32
//                       on the VALU with the 34 x 128 token block resident in LDS (84 MFLOP/episode,
//                       once per episode -> latency, not throughput, is what matters here).
//   weightgen_kernel    the 73 output heads collapsed into ONE GEMM  theta = ctx @ W_cat + b_cat,
//                       computed transposed (theta^T tile = W_cat^T tile x ctx^T) with split-bf16
//                       MFMA (3 x v_mfma_f32_32x32x16_bf16, ~2^-16 relative) so that each lane ends up
//                       holding 16 consecutive packed positions of one episode and stores them
//                       straight into the policy kernel's arena layout (layout.h).  HBM-bound:
//                       reads W_cat once (hi+lo planes), writes the arena once.
//   export_theta_kernel arena -> reference-order theta[B, G] (parity tests / `base_params` views).
#include "common.h"
#include "kernels.h"

namespace hvla {

// ------------------------------------------------------------------------------------------------
// context encoder
// ------------------------------------------------------------------------------------------------
constexpr int CTX_THREADS = 1024;   // 16 waves: one workgroup (one episode) per CU, so the waves of that one workgroup have to hide each other's latencies
constexpr int CTX_RG = 20;   // max rows per row-group (S <= 40)

// y[s][n] (+)= sum_k xs[s][k] * W[k][n]   for s in [0,S), n in [0,N); xs in LDS (row stride xs_ld),
// W global [K][N] (flax kernel layout), result handed to `sink(s, n, value)`.
template <typename Sink>
__device__ __forceinline__ void dense_rows(const float* __restrict__ xs, int xs_ld, int S, int K,
                                           const float* __restrict__ W, int N, Sink sink) {
  // the S rows are cut into nrg row groups, as many as there are threads for (at least two: CTX_RG rows per thread at most);
  // an output element is one thread's k = 0 .. K-1 chain whatever the grouping, so the bits do not depend on it
  int nrg = CTX_THREADS / N;
  if (nrg < 2) nrg = 2;
  const int rg = (S + nrg - 1) / nrg;
  const int items = nrg * N;
  for (int it = threadIdx.x; it < items; it += CTX_THREADS) {
    const int n = it % N, g0 = (it / N) * rg;
    float acc[CTX_RG];
#pragma unroll
    for (int r = 0; r < CTX_RG; ++r) acc[r] = 0.f;
    for (int k = 0; k < K; k += 4) {
      const float w0 = W[(size_t)(k + 0) * N + n], w1 = W[(size_t)(k + 1) * N + n];
      const float w2 = W[(size_t)(k + 2) * N + n], w3 = W[(size_t)(k + 3) * N + n];
#pragma unroll
      for (int r = 0; r < CTX_RG; ++r) {
        if (r < rg) {                          // uniform
          int row = g0 + r;
          row = row < S ? row : S - 1;          // clamp: discarded at the sink
          const f32x4 x = *reinterpret_cast<const f32x4*>(xs + row * xs_ld + k);
          acc[r] = fmaf(x[0], w0, fmaf(x[1], w1, fmaf(x[2], w2, fmaf(x[3], w3, acc[r]))));
        }
      }
    }
#pragma unroll
    for (int r = 0; r < CTX_RG; ++r) {
      const int row = g0 + r;
      if (r < rg && row < S) sink(row, n, acc[r]);
    }
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

__global__ __launch_bounds__(CTX_THREADS) void ctx_encoder_kernel(CtxParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int b = blockIdx.x;
  const int T = p.T, S = p.T + 2, C = p.C, F = p.F, Hc = p.heads, hc = C / Hc;
  const int ldx = C + 4, ldq = 3 * C + 4, ldf = F + 4;
  float* x = reinterpret_cast<float*>(smem);            // [S][ldx] residual stream
  float* h = x + S * ldx;                                // [S][ldx] LN output / attention output
  float* big = h + S * ldx;                              // [S][max(ldq, ldf)] qkv or mlp hidden
  // ---- token projection: x[s] = tok[s] @ Wt + bt + pos_t[s]     (hypernetwork.py:112-115)
  {
    const float* tok = p.tok + (size_t)b * T * p.lang_dim;
    // stage the token rows in K-chunks of `kc` through `big`
    const int kc = 128, ldt = kc + 4;
    const int nrg = CTX_THREADS / C, rg = (T + nrg - 1) / nrg;       // C divides CTX_THREADS (128, 64, 32)
    float acc[CTX_RG];
    const int n = threadIdx.x % C, g0 = (threadIdx.x / C) * rg;
    const bool active = threadIdx.x < nrg * C;
#pragma unroll
    for (int r = 0; r < CTX_RG; ++r) acc[r] = 0.f;
    for (int k0 = 0; k0 < p.lang_dim; k0 += kc) {
      const int kw = min(kc, p.lang_dim - k0);
      __syncthreads();
      for (int i = threadIdx.x; i < T * kc; i += CTX_THREADS) {
        const int s = i / kc, k = i % kc;
        big[s * ldt + k] = k < kw ? tok[(size_t)s * p.lang_dim + k0 + k] : 0.f;
      }
      __syncthreads();
      if (active) {
        for (int k = 0; k < kw; k += 4) {
          const float* W = p.w_tok + (size_t)(k0 + k) * C + n;
          const float w0 = W[0], w1 = W[C], w2 = W[2 * C], w3 = W[3 * C];
#pragma unroll
          for (int r = 0; r < CTX_RG; ++r) {
            if (r < rg) {                                            // uniform
              int row = g0 + r;
              row = row < T ? row : T - 1;
              const f32x4 xv = *reinterpret_cast<const f32x4*>(big + row * ldt + k);
              acc[r] = fmaf(xv[0], w0, fmaf(xv[1], w1, fmaf(xv[2], w2, fmaf(xv[3], w3, acc[r]))));
            }
          }
        }
      }
    }
    if (active) {
#pragma unroll
      for (int r = 0; r < CTX_RG; ++r) {
        const int row = g0 + r;
        if (r < rg && row < T) x[row * ldx + n] = acc[r] + p.b_tok[n] + p.pos_tok[row * C + n];
      }
    }
    __syncthreads();
    // ---- initial-image CLS projection (hypernetwork.py:118-128) and layer token (:144-145)
    const float* cls = p.cls + (size_t)b * p.E;
    for (int i = threadIdx.x; i < p.E; i += CTX_THREADS) big[i] = cls[i];
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += CTX_THREADS) {
      float a = 0.f;
      for (int k = 0; k < p.E; ++k) a = fmaf(big[k], p.w_img[(size_t)k * C + c], a);
      x[T * ldx + c] = a + p.b_img[c] + p.pos_img[c];
      x[(T + 1) * ldx + c] = p.pos_layer[c];
    }
    __syncthreads();
  }
  const int64_t* am = p.attn_mask + (size_t)b * T;
  // ---- context Transformer (transformer.py:127-262)
  for (int l = 0; l < p.layers; ++l) {
    const CtxLayer& w = p.layer[l];
    ln_rows(x, h, ldx, S, C, w.ln0_s, w.ln0_b);
    __syncthreads();
    // q | k | v = h @ W{q,k,v} + b    (flax DenseGeneral kernel [C, H, hd] == [C, C] row-major)
    for (int which = 0; which < 3; ++which) {
      const float* W = which == 0 ? w.wq : which == 1 ? w.wk : w.wv;
      const float* B = which == 0 ? w.bq : which == 1 ? w.bk : w.bv;
      const float sc = which == 0 ? rsqrtf((float)hc) : 1.f;     // query pre-scaling
      dense_rows(h, ldx, S, C, W, C, [&](int s, int n, float v) { big[s * ldq + which * C + n] = (v + B[n]) * sc; });
    }
    __syncthreads();
    // attention: one thread per (head, query)
    for (int it = threadIdx.x; it < Hc * S; it += CTX_THREADS) {
      const int hh = it / S, q = it % S;
      const float* qp = big + q * ldq + hh * hc;
      float sc[2 * CTX_RG];
      float mx = -3.4028234663852886e38f;
#pragma unroll
      for (int k = 0; k < 2 * CTX_RG; ++k) {
        float s = -3.4028234663852886e38f;      // finfo(float32).min for masked logits
        if (k < S) {
          const bool keep = k < T ? (am[k] != 0) : (k == T ? true : (q == T + 1));
          if (keep) {
            const float* kp = big + k * ldq + C + hh * hc;
            float a = 0.f;
            for (int d = 0; d < hc; ++d) a = fmaf(qp[d], kp[d], a);
            s = a;
          }
          mx = fmaxf(mx, s);
        }
        sc[k] = s;
      }
      float den = 0.f;
#pragma unroll
      for (int k = 0; k < 2 * CTX_RG; ++k) {
        const float e = k < S ? __expf(sc[k] - mx) : 0.f;
        sc[k] = e;
        den += e;
      }
      const float inv = 1.f / den;
      for (int d = 0; d < hc; ++d) {
        float a = 0.f;
#pragma unroll
        for (int k = 0; k < 2 * CTX_RG; ++k)
          if (k < S) a = fmaf(sc[k], big[k * ldq + 2 * C + hh * hc + d], a);
        h[q * ldx + hh * hc + d] = a * inv;
      }
    }
    __syncthreads();
    // out projection + residual   (kernel [H, hd, C] == [C, C] row-major)
    dense_rows(h, ldx, S, C, w.wo, C, [&](int s, int n, float v) { x[s * ldx + n] += v + w.bo[n]; });
    __syncthreads();
    ln_rows(x, h, ldx, S, C, w.ln1_s, w.ln1_b);
    __syncthreads();
    dense_rows(h, ldx, S, C, w.w1, F, [&](int s, int n, float v) { big[s * ldf + n] = gelu_tanh(v + w.b1[n]); });
    __syncthreads();
    dense_rows(big, ldf, S, F, w.w2, C, [&](int s, int n, float v) { x[s * ldx + n] += v + w.b2[n]; });
    __syncthreads();
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
template <int KS>
__global__ __launch_bounds__(256) void weightgen_kernel(WeightGenParams p) {
  __shared__ __attribute__((aligned(16))) __bf16 stage[2][32][128];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int tile = blockIdx.x * 4 + wave;
  const bool via_lds = blockIdx.x * 4 + 3 < p.ntiles && (blockIdx.x * 4 + 4) * 32 <= p.Gm;     // workgroup-uniform
  if (tile >= p.ntiles) return;                  // (only in a workgroup that is not via_lds: no barrier is skipped)
  const int col = lane & 31, half = lane >> 5;
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
  const int C = KS * 16;
  for (int b0 = 0; b0 < p.B; b0 += 32) {
    const int b = b0 + col;
    const int bc = b < p.B ? b : p.B - 1;
    const bf16x8* Bh = reinterpret_cast<const bf16x8*>(p.ctx_hi + (size_t)bc * C + half * 8);
    const bf16x8* Bl = reinterpret_cast<const bf16x8*>(p.ctx_lo + (size_t)bc * C + half * 8);
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = bias[r];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const bf16x8 bh = Bh[ks * 2], bl = Bl[ks * 2];       // 16 bf16 per k-step = 2 x bf16x8
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
      bf16x8* sh = reinterpret_cast<bf16x8*>(&stage[0][col][wave * 32 + half * 16]);
      bf16x8* sl = reinterpret_cast<bf16x8*>(&stage[1][col][wave * 32 + half * 16]);
      sh[0] = h0, sh[1] = h1, sl[0] = l0, sl[1] = l1;
      __syncthreads();
      const size_t gpos = (size_t)blockIdx.x * 128 + (lane & 15) * 8;          // 16 lanes x 16 B = one episode's 256-byte run
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int row = wave * 8 + u * 4 + (lane >> 4);
        if (b0 + row < p.B) {
          const bf16x8 vh = *reinterpret_cast<const bf16x8*>(&stage[0][row][(lane & 15) * 8]);
          const bf16x8 vl = *reinterpret_cast<const bf16x8*>(&stage[1][row][(lane & 15) * 8]);
          *reinterpret_cast<bf16x8*>(p.wh + (size_t)(b0 + row) * p.Gm + gpos) = vh;
          *reinterpret_cast<bf16x8*>(p.wl + (size_t)(b0 + row) * p.Gm + gpos) = vl;
        }
      }
    } else if (b < p.B) {
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

// ---- launchers -----------------------------------------------------------------------------------
hipError_t launch_ctx_encoder(const CtxParams& p, int B, hipStream_t st) {
  const int S = p.T + 2;
  const int ldx = p.C + 4, ldq = 3 * p.C + 4, ldf = p.F + 4;
  int bigld = ldq > ldf ? ldq : ldf;
  if (bigld < 132) bigld = 132;
  size_t big_elems = (size_t)S * bigld;
  if (big_elems < (size_t)p.E) big_elems = p.E;
  const size_t smem = ((size_t)2 * S * ldx + big_elems) * sizeof(float);
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
  hipLaunchKernelGGL(ctx_encoder_kernel, dim3(B), dim3(CTX_THREADS), smem, st, p);
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
