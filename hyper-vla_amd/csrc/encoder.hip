// encoder.hip — the frozen DINOv2 image encoder inside sample_actions (reference:
// hypervla/components/base_vit.py:109-122 calling transformers' FlaxDinov2Module).
// 99.7 % of the per-step FLOPs.  Shared weights, so these are plain dense contractions with
// M = B * 257 rows:
//
//   im2col_kernel      u8 NHWC image -> [B*P, 2*Kp1] 16-bit patch matrix of (pixel - 128), twice (exact integers;
//                      the /255, mean, std normalisation is folded into the patch weights/bias at load and the
//                      weights are split hi|lo along K)
//   gemm256p_kernel    production GEMM C = A[M,K] x W[N,K]^T on v_mfma_f32_16x16x32_{f16,bf16}: 256x256x64 tiles, LDS-DMA,
//                      four phases per K-tile, persistent.  A 256-row tile is exactly the 256 patch rows of ONE image
//                      (rows b*257 + 1 .. b*257 + 256; the B CLS rows go to gemm64_kernel with a row stride): no ragged
//                      tile rows, and everything that is per image (the bias row below, column sums) is per tile.
//   gemm64_kernel      64x64 tiles for small row counts (B <= 7, and the CLS rows);  gemm_kernel: 128x128 tiles for
//                      widths that are not multiples of 256 (DINOv2-small).  Fused epilogues:
//                        PATCH  + bias + position embedding  -> f32 residual stream (row remap b*P+p -> b*S+1+p)
//                        QKV    + bias, q * 1/sqrt(hd)        -> 16-bit
//                        GELU   + bias, exact erf GELU        -> 16-bit
//                        RES    x += (acc + bias) * layerscale -> f32 residual stream (in place)
//   mean rows / corr   first-order compensation of the WEIGHT rounding: A W = A W16 + A dW with
//                      dW = W - W16; A dW is replaced by (per-image mean row of A) dW, a [B, N] table (gemm64_kernel's
//                      second problem) that the epilogues add instead of the bias.  The mean rows come out of the kernels
//                      that write A: layernorm_img_kernel, attention_kernel, the GELU epilogue.  Rounding a shared weight perturbs every token of an image the same way,
//                      which the generated policy (it pools the 256 tokens) feels about sqrt(257) times more than the
//                      independent rounding of activations; the mean row carries most of that coherent part.
//   layernorm_kernel   f32 rows -> 16-bit rows (eps 1e-6), one wavefront per row; final variant drops the
//                      CLS row and writes the f32 patch tokens the policy consumes
//   attention_kernel   S = 257, head_dim 64: one workgroup per (image, head), one wavefront per 32-query
//                      block; K and V resident in LDS (V consumed through ds_read_b64_tr_b16); transposed scores (keys on accumulator rows,
//                      query on the lane) so softmax is in-lane and P feeds the PV MFMA from registers.
//
// Residual stream, LayerNorm statistics, softmax and GELU are f32; only MFMA operands are 16-bit.
#include <cstring>
#include <type_traits>

#include "common.h"
#include "kernels.h"
#include "plan.h"

namespace hvla {

// ------------------------------------------------------------------------------------------------
// Row m = [a | a] with a[k] = pixel - 128 (k < patch*patch*3, else 0): centred integers are exact in
// fp16/bf16, and the weight matrix is [W_hi | W_lo] so the single 16-bit GEMM over K = 2*Kp1 computes
// a . (W_hi + W_lo): the patch embedding is then accurate to ~2^-20 instead of 2^-11 (it dominated
// the encoder's error when W was rounded once).
// The same launch also writes the B CLS rows of the residual stream (x[b * S] = pos[0], which carries cls_token + position 0: one
// launch less per step -- at B = 1 a launch is ~5 us of a 1.2 ms step).
template <typename Op>
__global__ void im2col_kernel(const uint8_t* __restrict__ img, typename Op::elem* __restrict__ out, int B,
                              int image, int patch, int grid, int Kp, float* __restrict__ x, const float* __restrict__ pos, int S, int E) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < (size_t)B * E; i += (size_t)gridDim.x * blockDim.x)
    x[(i / E) * S * E + (i % E)] = pos[i % E];
  // one thread = 8 consecutive k of one patch row, converted once and stored into both halves of the row (Kp1 is a multiple of 8)
  const int Kp1 = Kp / 2, chunks = Kp1 / 8;
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
    const int k0 = ch * 8;
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
    *reinterpret_cast<typename Op::x8*>(out + m * Kp + Kp1 + ch * 8) = v;
  }
}

// ------------------------------------------------------------------------------------------------
enum { EPI_PATCH = 0, EPI_QKV = 1, EPI_GELU = 2, EPI_RES = 3, EPI_CORR = 4 };   // CORR: out f32 = bias + acc / 4096 (gemm64_kernel's second problem)

struct GemmArgs {
  const void* A;   // 16-bit; logical row m lives at global row row0 + m * row_step, K contiguous
  const void* W;   // [N][K] 16-bit
  int M, N, K;
  const float* bias;     // [N] (PATCH; the other epilogues when corr == nullptr)
  const float* aux;      // PATCH: pos [S][E];  RES: layerscale [N]
  void* out;             // 16-bit [rows][N] or f32 [rows][N], same row map as A (PATCH: its own remap)
  int P, S;              // patches / tokens per image (PATCH row remap; row -> image)
  int qcols;             // QKV: columns < qcols are scaled by qscale
  float qscale;
  const float* corr = nullptr;   // [B][2][N] bias rows per image HALF = bias + (mean row of A over that half of the image) . dW;
                                 // rows that are an image's CLS token keep the plain bias (one token of 257: nothing coherent
                                 // to compensate).  Two halves (round 3): on camera-like frames the coherent part of the weight
                                 // rounding differs between the upper and the lower half of the picture; the float64 study
                                 // (tests/studies/precision_budget.py --study hilo) puts the worst action error of 64 structured
                                 // episodes at 7.1e-4 with two mean rows against 1.0e-3 with one (finer bands add nothing)
  // gemm64_kernel, second problem in the same launch (M2 > 0): corr2[M2][N] = bias + abar2[M2][K] . dW2[N][K] / 4096
  const void* abar2 = nullptr;
  const void* dW2 = nullptr;
  float* corr2 = nullptr;
  int M2 = 0;
  int row0 = 0, row_step = 1;    // gemm64_kernel: the B CLS rows are rows b * S of the activation matrix
  int hsplit = 1 << 30;          // corr tables have TWO rows per image: [image][half][N], half = (token index >= hsplit); the
                                 // host sets 1 + P / 2 (tokens 1 .. P/2 | P/2 + 1 .. P: the two wave rows of an image-aligned tile)
  // gemm256p_kernel: tile row t covers global rows tile_row0 + t * tile_stride .. + 255 (image-aligned: 1, S; the patch
  // GEMM's contiguous rows: 0, 256); nbm tile rows, every tile full
  int tile_row0 = 0, tile_stride = 256, nbm = 0;
  void* colmean = nullptr;       // GELU, image-aligned tiles only: [B][2][N] 16-bit mean over each wave row's 128 rows of its rounded outputs
  // gemm256p_kernel<..., LNX = true> (RES / PATCH on image-aligned tiles): the LayerNorm that follows this GEMM (norm1 / norm2,
  // base_vit.py:109-122 -> HF Dinov2Layer) inside its epilogue -- nobody reads x again.  Every column tile of an image keeps its
  // 256 x 256 block of the new residual stream in the dead accumulators, publishes per-row (sum, sum of squares) over its 256
  // columns to ln_part, counts itself in ln_cnt[image], waits for the image's other column tiles (same round, same XCD: a few
  // microseconds of skew), combines the partials in column-tile order and normalises from registers -> ln_out (16-bit) and the
  // two mean rows -> ln_abar.  A tile whose partners are not due soon (another round of a persistent grid, or the bound ln_spin
  // ran out) does not wait: it marks itself abandoned in the image's word and goes on; the image's last arriver then normalises
  // that tile too, from the x the abandoner stored (write-through).  Same bits either way, and nothing can hang.
  void* ln_out = nullptr;        // 16-bit [B*S][N]
  const float* ln_scale = nullptr;
  const float* ln_bias = nullptr;
  void* ln_abar = nullptr;       // 16-bit [B][2][N] (nullable)
  uint32_t* ln_cnt = nullptr;    // [B]: bits 0-15 column tiles arrived since the call's memset (ln_target = launch number x nbn),
                                 //      bit 16 + c: column tile c of this launch was abandoned
  float* ln_part = nullptr;      // [B][nbn][256] entries {sum, tag, sum of squares, tag}: 16 bytes, zero when the call starts
  uint32_t ln_target = 0;        // launch number (1, 2, ...) x nbn: the count of a complete image
  uint32_t ln_tag = 0;           // launch number: the tag of this launch's entries in ln_part
  uint32_t ln_spin = 0;          // bound of the wait, in ticks of the 100 MHz constant clock
  uint32_t out_bytes = 0;        // size of `out` in bytes (buffer descriptor of the write-through stores)
  uint32_t part_bytes = 0;       // size of ln_part in bytes
  // persistent form (tile_origin_x): G = workgroups per XCD label / nbn, rows_aligned = rounds x G, rounds_div = rounds / nbn,
  // rows_xcd = images per XCD label, rcp = ceil(65536 / nbn)
  int lnx_G = 0, lnx_rows_aligned = 0, lnx_rounds_div = 0, lnx_rows_xcd = 0, lnx_rcp = 0;
  // gemm64c_kernel<..., FOLD = true> (the GEMM behind a LayerNorm at a small batch): the mean rows are not read from abar2 but
  // added up by every workgroup from layernorm_split_kernel's partial column sums [image][half][LNW][K] f32, with
  // layernorm_mean_kernel's arithmetic -- that launch (two per layer, 4.5 us + a kernel boundary each at B = 1) is gone
  const float* ln_partial = nullptr;
};

// Row-major epilogue shared by the three GEMM kernels.  They run the MFMA with the activation fragment as the first
// operand and stage W so that the LDS row 16 nt + c of a wave's 64 columns holds global column 4 c + nt (wperm() below):
// lane (fr, fq) then owns, for each of its rows m = m_base + 16 mt + 4 fq + r, the FOUR CONSECUTIVE columns
// n_base + 4 fr + {0..3} (one from each accumulator tile nt).  A store instruction therefore covers 4 rows x 64
// consecutive columns -- four full 128-B lines of a 16-bit output, eight of an f32 one -- instead of 64 scattered 8-B
// pieces (measured on the QKV shape: the per-lane-column layout spent 8.6 us of a 24 us tile in its epilogue even on an
// otherwise idle chip; the memory pipe handles one line per cycle, not one instruction).
__device__ __forceinline__ int wperm(int rho) { return (rho & ~63) + 4 * (rho & 15) + ((rho >> 4) & 3); }
// one entry of the per-image bias row: bias + (mean row . dW) / 4096 (dW is stored x 4096).  ONE definition, an explicit fma,
// for the stand-alone table (EPI_CORR) and for the form fused into gemm64c_kernel: the same bits from either.
__device__ __forceinline__ f32x4 corr_value(f32x4 acc, float b) {
  return f32x4{fmaf(acc[0], 1.f / 4096.f, b), fmaf(acc[1], 1.f / 4096.f, b), fmaf(acc[2], 1.f / 4096.f, b), fmaf(acc[3], 1.f / 4096.f, b)};
}

// ROWBIAS: the bias row depends on the row's image (rows of several images in one tile: gemm64_kernel / gemm_kernel with a
// corr table); otherwise one bias row per call (pre_b4, or g.bias).  FULL: every row of the wave tile exists -- straight-
// line code with 32-bit element offsets (with per-row `m < M` branches the compiler puts an s_waitcnt vmcnt(0) into every
// predicated block, which also waits for the previous STORE: 32 serialised store round trips per wave, 6.5 us per tile).
// CS (GELU, FULL only): also returns in cs this lane's sums over its rows of the ROUNDED outputs of its four columns,
// accumulated in the operand type itself (fp16: two v_pk_add_f16 per row instead of four converts and four adds; 32 values
// per lane, and the mean row only needs a few per cent: colmean_kernel restates exactly this arithmetic).
// WT (RES / PATCH): the f32 rows are stored write-through (buffer stores with sc1: they reach the memory side, no dirty line
// stays in this XCD's L2), so that another workgroup may read them inside this launch behind the stores' vmcnt(0), a
// ticket and an agent-scope acquire (the LayerNorm tail of gemm256p_kernel) without a release fence, which would write
// back the whole L2 once per tile (measured in round 2: 830 us per launch).
template <typename Op, int EPI, int MT, bool FULL, bool ROWBIAS, bool CS = false, bool WT = false, bool NT = false>
__device__ __forceinline__ void gemm_epilogue_rows_impl(const f32x4 (&acc)[4][MT], const GemmArgs& g, int m_base, int n_base,
                                                        int fr, int fq, const f32x4* pre_b4, const f32x4* pre_l4,
                                                        typename Op::x4* cs = nullptr,
                                                        const __attribute__((address_space(3))) float* lds_corr = nullptr,   // ROWBIAS, lds_img0 >= 0: [image - lds_img0][64] for this
                                                        int lds_img0 = -1) {                                                 // block's columns (LDS offset 0 is a valid address)
  using T = typename Op::elem;
  const int n = n_base + 4 * fr;
  f32x4 b4 = f32x4{0.f, 0.f, 0.f, 0.f};
  if constexpr (!ROWBIAS) b4 = pre_b4 ? *pre_b4 : *reinterpret_cast<const f32x4*>(g.bias + n);
  f32x4 l4 = f32x4{1.f, 1.f, 1.f, 1.f};
  if constexpr (EPI == EPI_RES) l4 = pre_l4 ? *pre_l4 : *reinterpret_cast<const f32x4*>(g.aux + n);
  const float q = EPI == EPI_PATCH ? g.qscale : ((EPI == EPI_QKV && n_base < g.qcols) ? g.qscale : 1.f);
  __amdgpu_buffer_rsrc_t wt_rsrc;
  if constexpr (WT) wt_rsrc = __builtin_amdgcn_make_buffer_rsrc(g.out, 0, (int)g.out_bytes, 0x00020000);
  auto store_f32x4 = [&](uint32_t elem_off, f32x4 v) {
    if constexpr (WT) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), wt_rsrc, (int)(elem_off * 4u), 0, 16);   // aux 16 = sc1
    else *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(g.out) + elem_off) = v;
  };
  __amdgpu_buffer_rsrc_t nt_rsrc;
  uint32_t nt_vo = 0;
  if constexpr (NT && FULL && (EPI == EPI_QKV || EPI == EPI_GELU)) {
    nt_rsrc = __builtin_amdgcn_make_buffer_rsrc(g.out, 0, -1, 0x00020000);     // (32-bit BYTE offsets: the host takes this form for outputs below 4 GiB only, plan.h nt16_addressable)
    nt_vo = ((uint32_t)(g.row0 + (m_base + 4 * fq) * g.row_step) * (uint32_t)g.N + (uint32_t)n) * (uint32_t)sizeof(T);
  }
  constexpr int RB = MT < 2 ? 1 : (EPI == EPI_RES && MT >= 4 ? 4 : 2);   // m-tiles per batch: residual loads of a batch are issued together
#pragma unroll
  for (int mp = 0; mp < MT; mp += RB) {
    f32x4 xin[RB][4], brow[RB][4];
    uint32_t off[RB][4];
    bool ok[RB][4];
#pragma unroll
    for (int u = 0; u < RB; ++u)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int m = m_base + (mp + u) * 16 + 4 * fq + r;
        ok[u][r] = FULL || m < g.M;
        const int grow = g.row0 + m * g.row_step;
        off[u][r] = (uint32_t)grow * (uint32_t)g.N + (uint32_t)n;
        if constexpr (EPI == EPI_PATCH) {
          const int prow = 1 + (m % g.P);
          off[u][r] = (uint32_t)((m / g.P) * g.S + prow) * (uint32_t)g.N + (uint32_t)n;
          if (ok[u][r]) xin[u][r] = *reinterpret_cast<const f32x4*>(g.aux + (uint32_t)prow * (uint32_t)g.N + (uint32_t)n);
        }
        if constexpr (EPI == EPI_RES) {
          // NT (big batches): the residual rows are read once here and written back; read non-temporally they do not push the
          // running GEMM's panels out of L2 either (same box: out 1.345 -> 1.330, fc2 3.238 -> 3.206 ms per step)
          if (ok[u][r]) {
            if constexpr (NT) xin[u][r] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(g.out) + off[u][r]));
            else xin[u][r] = *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(g.out) + off[u][r]);
          }
        }
        if constexpr (ROWBIAS) {
          brow[u][r] = f32x4{0.f, 0.f, 0.f, 0.f};
          const int img = grow / g.S;
          const int tok = grow - img * g.S, hf = tok >= g.hsplit ? 1 : 0;      // table row = (image, half)
          if (lds_img0 >= 0) {                             // wave-uniform: the table was computed by this workgroup (gemm64c_kernel)
            if (ok[u][r]) {
              if (tok) brow[u][r] = *reinterpret_cast<const __attribute__((address_space(3))) f32x4*>(lds_corr + ((img - lds_img0) * 2 + hf) * 64 + 4 * fr);
              else brow[u][r] = *reinterpret_cast<const f32x4*>(g.bias + (uint32_t)n);      // CLS row: plain bias
            }
          } else {
            const float* bsrc = tok ? g.corr + (uint32_t)(img * 2 + hf) * (uint32_t)g.N : g.bias;   // CLS row: plain bias
            if (ok[u][r]) brow[u][r] = *reinterpret_cast<const f32x4*>(bsrc + (uint32_t)n);
          }
        }
      }
#pragma unroll
    for (int u = 0; u < RB; ++u) {
      f32x4 t[4];                                    // [column c] over the four rows r: the accumulators' own register pairs
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        if constexpr (EPI == EPI_PATCH) t[c] = __builtin_elementwise_fma(acc[c][mp + u], f32x4{q, q, q, q}, f32x4{b4[c], b4[c], b4[c], b4[c]});   // patch weights are stored x256 (16-bit range)
        else if constexpr (EPI == EPI_CORR) t[c] = corr_value(acc[c][mp + u], b4[c]);
        else if constexpr (ROWBIAS) t[c] = acc[c][mp + u] + f32x4{brow[u][0][c], brow[u][1][c], brow[u][2][c], brow[u][3][c]};
        else t[c] = acc[c][mp + u] + b4[c];
        if constexpr (EPI == EPI_QKV) t[c] *= q;
        if constexpr (EPI == EPI_GELU) {
          // (the same chain on plain instead of packed fmas -- a packed f32 instruction costs attention_kernel three plain ones --
          // changes nothing here: fc1 3.81 / 3.76 against 3.81 / 3.80 ms per step, same box; two waves per SIMD, stores in between)
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
          if constexpr (CS) *cs += o;                      // rows in ascending order: the order colmean_kernel restates
          // NT (gemm256p_kernel on a big batch): q / k / v and the GELU outputs are read once, a kernel later, by other CUs
          // (attention's loads are non-temporal too; fc2 fetches its A panels from the memory side anyway) -- kept out of L2 they
          // leave the GEMM's own A / W panels there.  Same box, B = 256: QKV 2.52 -> 2.40, fc1 3.80 -> 3.72, fc2 3.31 -> 3.27 ms
          // per step, the step 15.46 -> 15.25.  Small batches, whose outputs the next kernel finds in L2 / Infinity Cache, lose
          // (B = 1: 1.246 -> 1.266 ms, B = 16: +0.4 %): the host asks for it from 96 MB of output on.
          // (round 5: as a buffer store -- ONE byte offset per lane, the row as a scalar offset -- instead of a 64-bit address per
          // row: 32 x (v_mad_u64_u32 + v_lshl_add_u64 + ...) = a hundred of an epilogue's vector instructions per wave)
          if constexpr (NT && FULL) {
            int rowb = g.row_step * g.N * (int)sizeof(T);
            asm volatile("" : "+s"(rowb));                 // (opaque per site: shared, all 32 row offsets are kept in SGPRs at once)
            __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, o), nt_rsrc, (int)nt_vo, ((mp + u) * 16 + r) * rowb, 2);   // aux 2 = nt
          } else if constexpr (NT) __builtin_nontemporal_store(o, reinterpret_cast<typename Op::x4*>(reinterpret_cast<T*>(g.out) + off[u][r]));
          else *reinterpret_cast<typename Op::x4*>(reinterpret_cast<T*>(g.out) + off[u][r]) = o;
        } else if constexpr (EPI == EPI_RES) {
          f32x4 x = xin[u][r];
#pragma unroll
          for (int c = 0; c < 4; ++c) x[c] = fmaf(t[c][r], l4[c], x[c]);
          store_f32x4(off[u][r], x);
        } else if constexpr (EPI == EPI_CORR) {
          *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(g.out) + off[u][r]) = f32x4{t[0][r], t[1][r], t[2][r], t[3][r]};
        } else {
          f32x4 x = xin[u][r];
#pragma unroll
          for (int c = 0; c < 4; ++c) x[c] += t[c][r];
          store_f32x4(off[u][r], x);
        }
      }
    }
  }
}

// gemm64_kernel / gemm_kernel: ragged last tile, bias row per image when a corr table is given
template <typename Op, int EPI, int MT>
__device__ __forceinline__ void gemm_epilogue_rows(const f32x4 (&acc)[4][MT], const GemmArgs& g, int m_base, int n_base,
                                                   int fr, int fq) {
  const bool full = m_base + 16 * MT <= g.M;
  if constexpr (EPI == EPI_PATCH || EPI == EPI_CORR) {
    if (full) gemm_epilogue_rows_impl<Op, EPI, MT, true, false>(acc, g, m_base, n_base, fr, fq, nullptr, nullptr);
    else gemm_epilogue_rows_impl<Op, EPI, MT, false, false>(acc, g, m_base, n_base, fr, fq, nullptr, nullptr);
  } else {
    if (g.corr) {
      if (full) gemm_epilogue_rows_impl<Op, EPI, MT, true, true>(acc, g, m_base, n_base, fr, fq, nullptr, nullptr);
      else gemm_epilogue_rows_impl<Op, EPI, MT, false, true>(acc, g, m_base, n_base, fr, fq, nullptr, nullptr);
    } else {
      if (full) gemm_epilogue_rows_impl<Op, EPI, MT, true, false>(acc, g, m_base, n_base, fr, fq, nullptr, nullptr);
      else gemm_epilogue_rows_impl<Op, EPI, MT, false, false>(acc, g, m_base, n_base, fr, fq, nullptr, nullptr);
    }
  }
}

// ------------------------------------------------------------------------------------------------
// gemm_kernel -- 128x128x64 tiles, 4 waves x (64x64), LDS double-buffered through registers with 144-B padded rows.
// Serves widths that are not multiples of 256 (DINOv2-small) and geometries whose images are not 256 patches.
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
  const int bid = xcd_run(blockIdx.x, nbm * nbn);
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
    ap[i] = A + (size_t)(g.row0 + m * g.row_step) * g.K + lch * 8;
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
// gemm64_kernel -- 64x64x64 tiles for SMALL row counts (B <= 7, and the B CLS rows of a big batch, which sit S rows apart:
// row0 / row_step): such a problem is pure latency, so it is cut into many small workgroups (256 rows x 768 columns -> 48)
// and each keeps SNS - 1 = 5 K-tiles of LDS-DMA in flight (6 stages x 16 KB, counted vmcnt, one raw barrier per K-tile).
// Four waves, wave w owns rows [16 w, +16) x all 64 columns: 10 ds_read_b128 and 8 MFMAs per K-tile.  LDS image, swizzle,
// W-row permutation, MFMA and k order are those of gemm256p_kernel, so a row gets the same bits whichever kernel
// computes it.
constexpr int SBM = 64, SBN = 64, SNS = 6;
// NS stages of 16 KB: 6 (5 K-tiles in flight, one workgroup per CU) or 4 (64 KB: two workgroups per CU, for the launches
// with more blocks than CUs -- the QKV and fc1 shapes -- which otherwise run as two rounds).  Same bits either way.
template <typename Op, int EPI, int NS = SNS>
__device__ __forceinline__ void gemm64_body(const GemmArgs& g, int bm, int bn) {
  using T = typename Op::elem;
  using X8 = typename Op::x8;
  extern __shared__ __attribute__((aligned(16))) char smem[];   // SNS x (A 8 KB | W 8 KB)
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
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
    aoff[j] = (uint32_t)(g.row0 + m * g.row_step) * (uint32_t)g.K + sch;
    woff[j] = (uint32_t)(n0 + wperm(rl)) * (uint32_t)g.K + sch;
  }
  const int KT = g.K / 64;
  auto issue = [&](int kt) {
    char* base = smem + (kt % NS) * 16384 + wave * 1024;
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
  for (int s = 0; s < NS - 1; ++s)
    if (s < KT) issue(s);
  for (int kt = 0; kt < KT; ++kt) {
    const int issued = kt + NS - 1 < KT ? kt + NS - 1 : KT;
    switch (issued - kt - 1) {                       // K-tiles issued after tile kt may stay in flight (4 pieces each)
      case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
      case 1: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
      case 2: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
      case 3: asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); break;
      default: asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); break;
    }
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();                    // tile kt landed for every wave; everyone is done reading tile kt - 1
    // compiler barrier: the fragment reads below must not be placed in front of the s_barrier (the builtin is "no memory" to
    // the compiler; while the DMA issue stood here it kept them behind -- with the reads first nothing did, and with two
    // or three workgroups per CU the batch of 1024 showed it: a wave read the tile before the other waves' pieces had landed)
    asm volatile("" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    const char* lb = smem + (kt % NS) * 16384;
    X8 fa[2], fw[4][2];
    fa[0] = *reinterpret_cast<const X8*>(lb + a_off + sw0);
    fa[1] = *reinterpret_cast<const X8*>(lb + a_off + sw1);
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
      fw[nt][0] = *reinterpret_cast<const X8*>(lb + w_off + nt * 2048 + sw0);
      fw[nt][1] = *reinterpret_cast<const X8*>(lb + w_off + nt * 2048 + sw1);
    }
    __builtin_amdgcn_sched_barrier(0);
    if (kt + NS - 1 < KT) issue(kt + NS - 1);      // into the stage of tile kt - 1; issued while the fragment reads are in flight
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) acc[nt][0] = Op::mma16(fa[kk], fw[nt][kk], acc[nt][0]);
  }
  gemm_epilogue_rows<Op, EPI, 1>(acc, g, m0 + wave * 16, n0, fr, fq);
}

// One launch, up to two independent problems of the same N and K: the GEMM itself (M rows) and, when M2 > 0, the per-image bias
// rows of the big GEMM that follows (corr2 = bias + abar2 . dW2 / 4096, two rows per image), so that the B CLS rows and the 2 B mean
// rows, both pure latency, cost one launch instead of two.  A column tile has ceil(M / 64) + ceil(M2 / 64) row tiles.
// Tile order (round 6): workgroups go round-robin over the 8 XCDs, each with an L2 of its own, so the linear tile index
// t = column tile x RT + row tile (RT = the row tiles of BOTH problems: every workgroup that reads one 64-column slice of W or
// of dW) is cut into 8 contiguous runs, XCD x taking run x (gemm_kernel's arithmetic): a slice then crosses the fabric once --
// twice where a run boundary falls inside a column tile -- and the other row tiles find it in that XCD's L2.  With
// bm = id % nbm on the raw block id the 4 CLS row tiles of a column landed on 4 L2s and the 8 mean-row tiles on 8: 58.7 MB
// fetched per fc1-shaped launch at B = 256 for 10.6 MB of operands (profiles/r5_pmc_fetch_size_by_kernel.csv).  No k order
// changes, so no bit does.  (xcd_run: plan.h)
template <typename Op, int EPI, int NS = SNS>
__global__ __launch_bounds__(256) void gemm64_kernel(GemmArgs g) {
  const int nbm1 = (g.M + SBM - 1) / SBM, nbm2 = (g.M2 + SBM - 1) / SBM, rt_all = nbm1 + nbm2;
  const int t = xcd_run(blockIdx.x, gridDim.x);
  const int bn = t / rt_all, bm = t - bn * rt_all;
  if (bm < nbm1) {
    gemm64_body<Op, EPI, NS>(g, bm, bn);
  } else {
    GemmArgs c = g;
    c.A = g.abar2; c.W = g.dW2; c.out = g.corr2; c.M = g.M2; c.row0 = 0; c.row_step = 1; c.corr = nullptr;
    gemm64_body<Op, EPI_CORR, NS>(c, bm - nbm1, bn);
  }
}

// gemm64c_kernel -- gemm64_kernel for a SMALL batch (all rows, up to 2047) with the per-image bias rows computed inside:
// the workgroup's 64 rows belong to at most 8 images (two at S = 257: four (image, half) mean rows); those against the block's 64 columns of
// dW are one more 16-row MFMA tile over the same K loop (wave w takes n-tile w: two MFMAs per K-tile and wave), staged
// beside A and W (dW 8 KB, mean rows 4 KB per stage).  The accumulators go through corr_value() into a 16 x 64 table in LDS
// that the epilogue reads instead of a table in memory: the separate table launch in front of every GEMM (48 per step,
// 12 us each at B = 1) is gone.  Same operands, same MFMA, same k order as gemm64_body<EPI_CORR> => the same bias rows.
constexpr int SNSC = 5, SSTC = 28672;   // stages x (A 8 KB | W 8 KB | dW 8 KB | mean rows 4 KB)
constexpr int LNW = 16;                 // waves of a LayerNorm workgroup
constexpr int LNG = 8;                  // row groups per image of layernorm_group_kernel = partial column sums per image: [half][fq]
// abar[(image, half)] = ((p0 + p1) + (p2 + p3)) / (P / 2): the four partials of a half (layernorm_group_kernel), n4 float4 apart
__device__ __forceinline__ f32x4 ln_half_sum(const f32x4* p, int n4) { return (p[0] + p[n4]) + (p[2 * n4] + p[3 * n4]); }
// FOLD: the mean rows of ALL K-tiles sit in a table behind the stages ([K-tile][16 rows][128 B], a stage's mean-row image), written
// once in the prologue from the LayerNorm's partial sums (GemmArgs::ln_partial); a stage is then A | W | dW = 24 KB, six pieces.
constexpr int SSTF = 24576;
constexpr size_t gemm64c_fold_lds(int K) { return (size_t)SNSC * SSTF + (size_t)(K / 64) * 2048; }
template <typename Op, int EPI, bool FOLD = false>
__global__ __launch_bounds__(256) void gemm64c_kernel(GemmArgs g) {
  constexpr int SST = FOLD ? SSTF : SSTC, PCS = FOLD ? 6 : 7;        // stage stride, LDS-DMA pieces per wave and K-tile
  using T = typename Op::elem;
  using X8 = typename Op::x8;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nbm = (g.M + SBM - 1) / SBM;
  // (round 6) the row tiles of a column tile on ONE XCD, as in gemm64_kernel: at B = 1 the five row tiles of a 64-column slice of W
  // and dW sat on five L2s and every slice crossed the fabric five times
  const int tix = xcd_run(blockIdx.x, gridDim.x), bm = tix % nbm, bn = tix / nbm;
  const int m0 = bm * SBM, n0 = bn * SBN;
  const T* A = reinterpret_cast<const T*>(g.A);
  const T* W = reinterpret_cast<const T*>(g.W);
  const T* dW = reinterpret_cast<const T*>(g.dW2);
  const T* AB = reinterpret_cast<const T*>(g.abar2);
  const int img0 = (g.row0 + m0 * g.row_step) / g.S;          // first image of this block's rows
  const int sch = ((lane & 7) ^ (lane >> 3)) * 8;
  uint32_t aoff[2], woff[2], boff;
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int rl = 32 * j + 8 * wave + (lane >> 3);
    int m = m0 + rl;
    m = m < g.M ? m : g.M - 1;
    aoff[j] = (uint32_t)(g.row0 + m * g.row_step) * (uint32_t)g.K + sch;
    woff[j] = (uint32_t)(n0 + wperm(rl)) * (uint32_t)g.K + sch;
  }
  {
    int mr = img0 * 2 + 8 * wave + (lane >> 3);               // rows 0..31 of the mean-row tile = (image, half) pairs from image
    mr = mr < g.M2 ? mr : g.M2 - 1;                           // img0 on (M2 = 2 B rows); the MFMA reads rows 0..15
    boff = (uint32_t)mr * (uint32_t)g.K + sch;
  }
  const int KT = g.K / 64;
  // LDS-DMA issued by hand (SGPR base + 32-bit lane offset, M0 saved / written / restored inside the statement: see
  // gemm256p_kernel): the builtin is a FLAT-encoded instruction for the compiler's wait-count model, and with one of those
  // pending it waits for EVERY LDS read in flight (lgkmcnt(0)) before an MFMA that needs the older fragment set only.
  const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
  auto dma = [&](const T* sb, uint32_t elem_off, uint32_t dst) {
    const uint32_t vo = elem_off * (uint32_t)sizeof(T);
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(vo), "s"(sb), "s"(dst));
  };
  auto issue_at = [&](int kt, int slot) {           // K-tile kt into the stage of tile `slot`
    const uint32_t base = lds0 + (uint32_t)((slot % SNSC) * SST + wave * 1024);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      dma(A, aoff[j] + kt * 64, base + j * 4096);
      dma(W, woff[j] + kt * 64, base + 8192 + j * 4096);
      dma(dW, woff[j] + kt * 64, base + 16384 + j * 4096);
    }
    if constexpr (!FOLD) dma(AB, boff + kt * 64, base + 24576);
  };
  f32x4 acc[4][1], acc2 = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int i = 0; i < 4; ++i) acc[i][0] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int fr = lane & 15, fq = lane >> 4;
  const int sw0 = (fq ^ (fr & 7)) << 4, sw1 = ((fq + 4) ^ (fr & 7)) << 4;
  const int a_off = (wave * 16 + fr) * 128, w_off = 8192 + fr * 128;
  const int d_off = 16384 + (wave * 16 + fr) * 128, b_off = 24576 + fr * 128;
#pragma unroll
  for (int s = 0; s < SNSC - 1; ++s) issue_at(s < KT ? s : KT - 1, s);
  char* mtab = smem + SNSC * SST;                     // FOLD: [KT][16 rows][128 B]
  if constexpr (FOLD) {
    // mean row r of the table = (image, half) img0 * 2 + r, for the images this block's rows belong to (the MFMA's other rows are
    // never written: their products land in accumulator rows nobody reads).  layernorm_mean_kernel's arithmetic: the LNW partials
    // in wave order, x 1 / rows of the half, rounded to the operand type.  Element (r, k) goes where the LDS-DMA of a [row][K]
    // matrix would have put it: K-tile k / 64, 16-byte chunk ((k % 64) / 8) ^ (r & 7) of the row's 128 bytes.
    int mlast = m0 + SBM - 1;
    mlast = mlast < g.M ? mlast : g.M - 1;
    const int nr = 2 * ((g.row0 + mlast * g.row_step) / g.S - img0 + 1);       // <= 16 (the host checks S >= 9)
    const int n4 = g.K / 4;
    for (int it = tid; it < nr * n4; it += 256) {
      const int r = it / n4, c4 = it - r * n4, bh = img0 * 2 + r;
      const f32x4 t = ln_half_sum(reinterpret_cast<const f32x4*>(g.ln_partial + (size_t)bh * 4 * g.K) + c4, n4);
      const float inv = 1.f / (float)(g.P / 2);
      typename Op::x4 o;
#pragma unroll
      for (int j = 0; j < 4; ++j) o[j] = (T)(t[j] * inv);
      const int k = 4 * c4;
      *reinterpret_cast<typename Op::x4*>(mtab + (k >> 6) * 2048 + r * 128 + ((((k & 63) >> 3) ^ (r & 7)) << 4) + (k & 7) * 2) = o;
    }
  }
  // One workgroup per CU (140 KB of stages) and at B = 1 only 60-240 workgroups in all: nothing but this workgroup's own waves
  // can hide its LDS round trip, so the fragments of K-tile kt + 1 are read while the MFMAs of tile kt run (two register
  // sets).  The MFMA chain of every accumulator is the one it always was: same bits.
  struct Frags {
    X8 fa[2], fw[4][2], fb[2], fd[2];
  };
  auto read_frags = [&](int t, Frags& f) {
    const char* lb = smem + (t % SNSC) * SST;
    f.fa[0] = *reinterpret_cast<const X8*>(lb + a_off + sw0);
    f.fa[1] = *reinterpret_cast<const X8*>(lb + a_off + sw1);
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
      f.fw[nt][0] = *reinterpret_cast<const X8*>(lb + w_off + nt * 2048 + sw0);
      f.fw[nt][1] = *reinterpret_cast<const X8*>(lb + w_off + nt * 2048 + sw1);
    }
    const char* mb = FOLD ? mtab + (t < KT ? t : KT - 1) * 2048 + fr * 128 : lb + b_off;   // (past the end: the last K-tile again, unused)
    f.fb[0] = *reinterpret_cast<const X8*>(mb + sw0);
    f.fb[1] = *reinterpret_cast<const X8*>(mb + sw1);
    f.fd[0] = *reinterpret_cast<const X8*>(lb + d_off + sw0);
    f.fd[1] = *reinterpret_cast<const X8*>(lb + d_off + sw1);
  };
  // No branch inside a step (behind a join the compiler's wait for `cur` becomes lgkmcnt(0), i.e. also waits for `nxt`): past
  // the end the reads fetch a stale stage and the DMA stages the last K-tile once more, both unused -- and every step has the
  // same SNSC - 3 younger tiles in flight when it waits for tile kt + 1, so that wait is the constant vmcnt(14).
  static_assert(SNSC == 5, "the wait constants below are for five stages of PCS pieces");
  auto step = [&](int kt, const Frags& cur, Frags& nxt) {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * PCS) : "memory");
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();                    // tile kt + 1 is in LDS for every wave, and every wave holds tile kt - 1's fragments (its MFMAs needed them)
    asm volatile("" ::: "memory");                   // (see gemm64_body)
    __builtin_amdgcn_sched_barrier(0);
    read_frags(kt + 1, nxt);
    __builtin_amdgcn_sched_barrier(0);
    issue_at(kt + SNSC - 1 < KT ? kt + SNSC - 1 : KT - 1, kt + SNSC - 1);    // into the stage of tile kt - 1
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) acc[nt][0] = Op::mma16(cur.fa[kk], cur.fw[nt][kk], acc[nt][0]);
      acc2 = Op::mma16(cur.fb[kk], cur.fd[kk], acc2);
    }
  };
  Frags f0, f1;
  // tile 0 (the prologue issued four tiles, past the end the last one again); FOLD: this wave's table entries are in LDS
  asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(3 * PCS) : "memory");
  __builtin_amdgcn_sched_barrier(0);
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  __builtin_amdgcn_sched_barrier(0);
  read_frags(0, f0);
  // (behind the loop header's join the compiler waits for every LDS read in flight before the first MFMA, i.e. step kt of
  // each group of four does not overlap its reads; the other three do)
  int kt = 0;
  for (; kt + 4 <= KT; kt += 4) {
    step(kt, f0, f1);
    step(kt + 1, f1, f0);
    step(kt + 2, f0, f1);
    step(kt + 3, f1, f0);
  }
  for (; kt < KT; ++kt) {
    step(kt, f0, f1);
    f0 = f1;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the DMA past the end still writes into the stages: the table below takes their place
  // acc2[r] = (mean row 2 img0 + 4 fq + r) . dW[:, n0 + 4 fr + wave] x 4096  ->  table[(image - img0) * 2 + half][4 fr + wave]
  __builtin_amdgcn_s_barrier();                      // every wave is done with the last stage: its space takes the table
  float* table = reinterpret_cast<float*>(smem);
  {
    const f32x4 cv = corr_value(acc2, g.bias[n0 + 4 * fr + wave]);
#pragma unroll
    for (int r = 0; r < 4; ++r) table[(4 * fq + r) * 64 + 4 * fr + wave] = cv[r];
  }
  __syncthreads();
  const auto lt = reinterpret_cast<const __attribute__((address_space(3))) float*>((__attribute__((address_space(3))) char*)smem);
  if (m0 + wave * 16 + 16 <= g.M) gemm_epilogue_rows_impl<Op, EPI, 1, true, true>(acc, g, m0 + wave * 16, n0, fr, fq, nullptr, nullptr, nullptr, lt, img0);
  else gemm_epilogue_rows_impl<Op, EPI, 1, false, true>(acc, g, m0 + wave * 16, n0, fr, fq, nullptr, nullptr, nullptr, lt, img0);
}

// gemm64c32_kernel -- gemm64c_kernel<EPI_RES> on 64 x 32 tiles, for the two residual GEMMs of a layer (out-projection, fc2:
// N = E) when the 64 x 64 grid would fill less than half of the chip (60 workgroups at B = 1).  The K loop of that kernel
// runs at what one CU's vector memory pipe moves (28 KB per K-tile); this one stages 20 KB per K-tile (A 8 | W 4 | dW 4 |
// mean rows 4) on twice as many CUs, seven stages deep.  W rows are staged so that LDS row 16 nt + c holds global column
// 2 c + nt: lane (fr, fq) owns the two adjacent columns n0 + 2 fr + {0, 1} of its four rows.  Same operands, same MFMA,
// same k order per accumulator, the same epilogue arithmetic (acc + bias row, then fma with the layer scale into x) as
// gemm_epilogue_rows_impl<EPI_RES>: the same bits as every other path (the batch-invariance tests cross them).
constexpr int SNS32 = 7, SST32 = 20480;
template <typename Op>
__global__ __launch_bounds__(256) void gemm64c32_kernel(GemmArgs g) {
  using T = typename Op::elem;
  using X8 = typename Op::x8;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nbm = (g.M + SBM - 1) / SBM;
  const int tix = xcd_run(blockIdx.x, gridDim.x), bm = tix % nbm, bn = tix / nbm;      // (see gemm64c_kernel)
  const int m0 = bm * SBM, n0 = bn * 32;
  const T* A = reinterpret_cast<const T*>(g.A);
  const T* W = reinterpret_cast<const T*>(g.W);
  const T* dW = reinterpret_cast<const T*>(g.dW2);
  const T* AB = reinterpret_cast<const T*>(g.abar2);
  const int img0 = (g.row0 + m0 * g.row_step) / g.S;
  const int sch = ((lane & 7) ^ (lane >> 3)) * 8;
  uint32_t aoff[2], woff, boff;
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    int m = m0 + 32 * j + 8 * wave + (lane >> 3);
    m = m < g.M ? m : g.M - 1;
    aoff[j] = (uint32_t)(g.row0 + m * g.row_step) * (uint32_t)g.K + sch;
  }
  {
    const int rl = 8 * wave + (lane >> 3);                    // LDS row of the W / dW stage -> global column 2 (rl & 15) + (rl >> 4)
    woff = (uint32_t)(n0 + 2 * (rl & 15) + (rl >> 4)) * (uint32_t)g.K + sch;
    int mr = img0 * 2 + rl;
    mr = mr < g.M2 ? mr : g.M2 - 1;
    boff = (uint32_t)mr * (uint32_t)g.K + sch;
  }
  const int KT = g.K / 64;
  const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
  auto dma = [&](const T* sb, uint32_t elem_off, uint32_t dst) {
    const uint32_t vo = elem_off * (uint32_t)sizeof(T);
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(vo), "s"(sb), "s"(dst));
  };
  auto issue_at = [&](int kt, int slot) {           // five pieces per wave and K-tile
    const uint32_t base = lds0 + (uint32_t)((slot % SNS32) * SST32 + wave * 1024);
    dma(A, aoff[0] + kt * 64, base);
    dma(A, aoff[1] + kt * 64, base + 4096);
    dma(W, woff + kt * 64, base + 8192);
    dma(dW, woff + kt * 64, base + 12288);
    dma(AB, boff + kt * 64, base + 16384);
  };
  f32x4 acc[2], acc2 = f32x4{0.f, 0.f, 0.f, 0.f};
  acc[0] = acc2;
  acc[1] = acc2;
  const int fr = lane & 15, fq = lane >> 4;
  const int sw0 = (fq ^ (fr & 7)) << 4, sw1 = ((fq + 4) ^ (fr & 7)) << 4;
  const int a_off = (wave * 16 + fr) * 128, w_off = 8192 + fr * 128;
  const int d_off = 12288 + ((wave & 1) * 16 + fr) * 128, b_off = 16384 + fr * 128;      // waves 2, 3 repeat n-tiles 0, 1 and drop the result
#pragma unroll
  for (int s = 0; s < SNS32 - 1; ++s) issue_at(s < KT ? s : KT - 1, s);
  struct Frags {
    X8 fa[2], fw[2][2], fb[2], fd[2];
  };
  auto read_frags = [&](int t, Frags& f) {
    const char* lb = smem + (t % SNS32) * SST32;
    f.fa[0] = *reinterpret_cast<const X8*>(lb + a_off + sw0);
    f.fa[1] = *reinterpret_cast<const X8*>(lb + a_off + sw1);
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
      f.fw[nt][0] = *reinterpret_cast<const X8*>(lb + w_off + nt * 2048 + sw0);
      f.fw[nt][1] = *reinterpret_cast<const X8*>(lb + w_off + nt * 2048 + sw1);
    }
    f.fb[0] = *reinterpret_cast<const X8*>(lb + b_off + sw0);
    f.fb[1] = *reinterpret_cast<const X8*>(lb + b_off + sw1);
    f.fd[0] = *reinterpret_cast<const X8*>(lb + d_off + sw0);
    f.fd[1] = *reinterpret_cast<const X8*>(lb + d_off + sw1);
  };
  // as in gemm64c_kernel: branch-free steps, constant waits (SNS32 - 3 = 4 younger tiles of 5 pieces in flight behind tile kt + 1)
  static_assert(SNS32 == 7, "the wait constants below are for seven stages of five pieces");
  auto step = [&](int kt, const Frags& cur, Frags& nxt) {
    asm volatile("s_waitcnt vmcnt(20)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    read_frags(kt + 1, nxt);
    __builtin_amdgcn_sched_barrier(0);
    issue_at(kt + SNS32 - 1 < KT ? kt + SNS32 - 1 : KT - 1, kt + SNS32 - 1);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      acc[0] = Op::mma16(cur.fa[kk], cur.fw[0][kk], acc[0]);
      acc[1] = Op::mma16(cur.fa[kk], cur.fw[1][kk], acc[1]);
      acc2 = Op::mma16(cur.fb[kk], cur.fd[kk], acc2);
    }
  };
  Frags f0, f1;
  asm volatile("s_waitcnt vmcnt(25)" ::: "memory");  // tile 0: five younger tiles in flight
  __builtin_amdgcn_sched_barrier(0);
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  __builtin_amdgcn_sched_barrier(0);
  read_frags(0, f0);
  int kt = 0;
  for (; kt + 4 <= KT; kt += 4) {
    step(kt, f0, f1);
    step(kt + 1, f1, f0);
    step(kt + 2, f0, f1);
    step(kt + 3, f1, f0);
  }
  for (; kt < KT; ++kt) {
    step(kt, f0, f1);
    f0 = f1;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  // acc2[r] = (mean row 2 img0 + 4 fq + r) . dW[:, n0 + 2 fr + wave] x 4096 (waves 0, 1)  ->  table[(image - img0) * 2 + half][2 fr + wave]
  __builtin_amdgcn_s_barrier();
  float* table = reinterpret_cast<float*>(smem);
  if (wave < 2) {
    const f32x4 cv = corr_value(acc2, g.bias[n0 + 2 * fr + wave]);
#pragma unroll
    for (int r = 0; r < 4; ++r) table[(4 * fq + r) * 32 + 2 * fr + wave] = cv[r];
  }
  __syncthreads();
  typedef float f32x2e __attribute__((ext_vector_type(2)));
  const int n = n0 + 2 * fr;
  const f32x2e l2 = *reinterpret_cast<const f32x2e*>(g.aux + n);
  const f32x2e b2 = *reinterpret_cast<const f32x2e*>(g.bias + n);
  f32x2e xin[4], brow[4];
  uint32_t off[4];
  bool ok[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {                      // the four residual loads are requested together
    const int m = m0 + wave * 16 + 4 * fq + r;
    ok[r] = m < g.M;
    const int grow = g.row0 + (ok[r] ? m : g.M - 1) * g.row_step;
    off[r] = (uint32_t)grow * (uint32_t)g.N + (uint32_t)n;
    xin[r] = *reinterpret_cast<const f32x2e*>(reinterpret_cast<const float*>(g.out) + off[r]);
    const int img = grow / g.S, tok = grow - img * g.S, hf = tok >= g.hsplit ? 1 : 0;
    brow[r] = tok ? *reinterpret_cast<const f32x2e*>(table + ((img - img0) * 2 + hf) * 32 + 2 * fr) : b2;      // CLS row: plain bias
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    if (!ok[r]) continue;
    f32x2e x = xin[r];
    x[0] = fmaf(acc[0][r] + brow[r][0], l2[0], x[0]);
    x[1] = fmaf(acc[1][r] + brow[r][1], l2[1], x[1]);
    *reinterpret_cast<f32x2e*>(reinterpret_cast<float*>(g.out) + off[r]) = x;
  }
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
// v + (the value of lane ^ 16) / v + (the value of lane ^ 32), without a lane id and without the LDS crossbar: v_permlane16_swap /
// v_permlane32_swap on the value and a copy of it leave [row 0, row 0, row 2, row 2] | [row 1, row 1, row 3, row 3] (lower half twice |
// upper half twice) in the two registers.  Same bits as v + __shfl_xor(v, 16 / 32) (an addition commutes).  Issued by hand: hipcc 7.2's
// builtin returns its first result twice (tools/permlane_swap_probe.hip); `s_nop 1`: two wait states behind the VALU write of v.
// (__shfl_xor needs __lane_id(), which the compiler computes once in front of a persistent kernel's tile loop and keeps -- or spills.)
__device__ __forceinline__ float xor16_sum(float v) {
  float u = v;
  asm("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(v), "+v"(u));
  return v + u;
}
__device__ __forceinline__ float xor32_sum(float v) {
  float u = v;
  asm("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(v), "+v"(u));
  return v + u;
}
__device__ __forceinline__ float lane_bcast(float v, int l) {
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), l));
}

// sum over all 64 lanes, the same value in every lane: DPP inside the four 16-lane rows, then the four row sums through
// SGPRs -- about a hundred cycles of dependent latency instead of six ds_bpermute round trips
__device__ __forceinline__ float wave64_sum(float v) {
  const float t = row16_sum(v);
  return (lane_bcast(t, 0) + lane_bcast(t, 16)) + (lane_bcast(t, 32) + lane_bcast(t, 48));
}

// ---- the CANONICAL arithmetic of norm1 / norm2 (HF Dinov2Layer.norm1 / .norm2 as called at base_vit.py:109-122; flax
// nn.LayerNorm, eps 1e-6, one-pass statistics E[x^2] - E[x]^2 clipped at 0: SURVEY.md Appendix A).  Round 5: the LayerNorm
// behind a residual GEMM runs inside that GEMM's epilogue (gemm256p_kernel<..., LNX>) on the accumulator layout, where a row's
// E columns are spread over column tiles (workgroups), waves and lanes; the stand-alone kernels (small batches, the CLS rows,
// geometries without image-aligned tiles) restate the same additions in the same order, so that a row of h and a mean row are
// the same bits whichever kernel wrote them (the batch-invariance tests cross them):
//   * a lane holds four consecutive columns v[0..3]:  s = (v0 + v1) + (v2 + v3),  q = fma(v1, v1, v0 v0) + fma(v3, v3, v2 v2);
//   * 16 lanes = 64 columns (a wave's columns of a GEMM tile / a 16-lane row of a 64-lane chunk): the DPP tree of row16_sum;
//   * four such groups = 256 columns (a GEMM column tile / a 64-lane chunk of four columns per lane): (g0 + g1) + (g2 + g3);
//   * the 256-column blocks in ascending order; columns past E count as zeros;
//   * mean = S / E as S * (1 / E), var = max(fma(Q, 1 / E, -(mean mean)), 0), rstd = rsqrt(var + 1e-6),
//     y = fma((x - mean) rstd, scale, bias), h = y rounded to the operand type;
//   * mean rows (the operand of the next GEMM's weight-rounding compensation): one per image HALF over its P / 2 patch rows
//     (the CLS row is left out, as in colmean_kernel and attention_kernel), f32 sums of y: row 16 mt + 4 fq + r of the half
//     goes to partial fq, each partial adds its rows in ascending order, the half is (p0 + p1) + (p2 + p3), x 1 / (P / 2),
//     rounded to the operand type.
__device__ __forceinline__ void ln_lane_stats(f32x4 v, float& s, float& q) {
  s = (v[0] + v[1]) + (v[2] + v[3]);
  q = fmaf(v[1], v[1], v[0] * v[0]) + fmaf(v[3], v[3], v[2] * v[2]);
}
__device__ __forceinline__ void ln_finish(float S, float Q, float invE, float& mean, float& rstd) {
  mean = S * invE;
  const float var = fmaxf(fmaf(Q, invE, -(mean * mean)), 0.f);
  rstd = rsqrtf(var + 1e-6f);
}
__device__ __forceinline__ float ln_value(float x, float mean, float rstd, float sc, float bi) { return fmaf((x - mean) * rstd, sc, bi); }

// a whole row held by ONE wave as NCH 64-lane chunks of four columns per lane (lane l of chunk i: columns 256 i + 4 l ..):
// statistics in the canonical order, the same value in every lane
template <int NCH>
__device__ __forceinline__ void ln_row_stats(const f32x4 (&cur)[NCH], float invE, float& mean, float& rstd) {
  float S = 0.f, Q = 0.f;
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    float s, q;
    ln_lane_stats(cur[i], s, q);
    s = row16_sum(s);
    q = row16_sum(q);
    const float cs = (lane_bcast(s, 0) + lane_bcast(s, 16)) + (lane_bcast(s, 32) + lane_bcast(s, 48));
    const float cq = (lane_bcast(q, 0) + lane_bcast(q, 16)) + (lane_bcast(q, 32) + lane_bcast(q, 48));
    S = i ? S + cs : cs;
    Q = i ? Q + cq : cq;
  }
  ln_finish(S, Q, invE, mean, rstd);
}

// Stand-alone norm1 / norm2 (+ the partial column sums of the mean rows): grid (8, B), workgroup (image b, group g = half * 4 + fq)
// normalises the 4 nmt rows  token 1 + half P/2 + 16 mt + 4 fq + r  (index i = 4 mt + r, ascending = the order partial fq adds
// them in), wave w the indices w, w + 16, ...; their y go to LDS and one thread per four columns adds them in index order ->
// partial[b][g][E] f32.  Group 0 also normalises the CLS row (token 0; in no mean row).  layernorm_mean_kernel (or the consumer
// GEMM itself: gemm64c_kernel<..., FOLD>) then combines (p0 + p1) + (p2 + p3) per half.  E % 4 == 0, E <= 1024, P % 32 == 0.
template <typename Op>
__global__ __launch_bounds__(LNW * 64) void layernorm_group_kernel(const float* __restrict__ x, typename Op::elem* __restrict__ out,
                                                                   const float* __restrict__ scale, const float* __restrict__ bias,
                                                                   float* __restrict__ partial, int S, int P, int E) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  f32x4* ys = reinterpret_cast<f32x4*>(smem);                        // [4 nmt][E / 4]
  const int grp = blockIdx.x, b = blockIdx.y;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int n4 = E / 4, nrows = P / 8;                               // rows of a group = 4 nmt = 4 (P / 2) / 16
  const int tok0 = 1 + (grp >> 2) * (P / 2) + 4 * (grp & 3);
  const float invE = 1.f / (float)E;
  f32x4 s4[4], b4[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int col = lane + 64 * i;
    const f32x4 z = f32x4{0.f, 0.f, 0.f, 0.f};
    s4[i] = col < n4 ? reinterpret_cast<const f32x4*>(scale)[col] : z;
    b4[i] = col < n4 ? reinterpret_cast<const f32x4*>(bias)[col] : z;
  }
  const int nwork = nrows + (grp == 0 ? 1 : 0);                      // index nrows (group 0 only) = the CLS row
  auto row_in = [&](int i, f32x4 (&r)[4]) {
    const int tok = i < nrows ? tok0 + 16 * (i >> 2) + (i & 3) : 0;
    const f32x4* xr = reinterpret_cast<const f32x4*>(x + ((size_t)b * S + tok) * E);
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int col = lane + 64 * c;
      r[c] = col < n4 ? xr[col] : f32x4{0.f, 0.f, 0.f, 0.f};
    }
  };
  // the wave's next row is requested before this one is worked on (at B = 1 a launch is two rows per wave and nothing else on
  // the chip: the second row's memory round trip used to start after the first row's stores)
  f32x4 cur[4], nxt[4];
  if (wave < nwork) row_in(wave, cur);
  for (int i = wave; i < nwork; i += LNW) {                          // wave-uniform
    const int tok = i < nrows ? tok0 + 16 * (i >> 2) + (i & 3) : 0;
    if (i + LNW < nwork) row_in(i + LNW, nxt);
    float mean, rstd;
    ln_row_stats<4>(cur, invE, mean, rstd);
    typename Op::elem* orow = out + ((size_t)b * S + tok) * E;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int col = lane + 64 * c;
      if (col < n4) {
        f32x4 y;
        typename Op::x4 o;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          y[j] = ln_value(cur[c][j], mean, rstd, s4[c][j], b4[c][j]);
          o[j] = (typename Op::elem)y[j];
        }
        reinterpret_cast<typename Op::x4*>(orow)[col] = o;
        if (partial && i < nrows) ys[i * n4 + col] = y;
      }
    }
#pragma unroll
    for (int c = 0; c < 4; ++c) cur[c] = nxt[c];
  }
  if (!partial) return;
  __syncthreads();
  if ((int)threadIdx.x < n4) {
    f32x4 cs = ys[threadIdx.x];
    int i = 1;
    for (; i + 8 <= nrows; i += 8) {                                   // eight LDS reads in flight, added in index order
      f32x4 t[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) t[j] = ys[(i + j) * n4 + threadIdx.x];
#pragma unroll
      for (int j = 0; j < 8; ++j) cs += t[j];
    }
    for (; i < nrows; ++i) cs += ys[i * n4 + threadIdx.x];
    reinterpret_cast<f32x4*>(partial + ((size_t)b * LNG + grp) * E)[threadIdx.x] = cs;
  }
}
// abar[(image, half)] = ((p0 + p1) + (p2 + p3)) / (P / 2), rounded to the operand type
template <typename Op>
__global__ __launch_bounds__(256) void layernorm_mean_kernel(const float* __restrict__ partial, typename Op::elem* __restrict__ abar,
                                                             int P, int E) {
  const int bh = blockIdx.x, n4 = E / 4;                             // (image, half)
  if ((int)threadIdx.x >= n4) return;
  const f32x4 t = ln_half_sum(reinterpret_cast<const f32x4*>(partial + (size_t)bh * 4 * E) + threadIdx.x, n4);
  const float inv = 1.f / (float)(P / 2);
  typename Op::x4 o;
#pragma unroll
  for (int j = 0; j < 4; ++j) o[j] = (typename Op::elem)(t[j] * inv);
  reinterpret_cast<typename Op::x4*>(abar + (size_t)bh * E)[threadIdx.x] = o;
}
// ------------------------------------------------------------------------------------------------
// gemm256p_kernel -- the production GEMM.  256x256x64 block tiles, 8 waves (2 along M x 4 along N, 128x64 each), operands
// staged global -> LDS by LDS-DMA (global_load_lds_dwordx4: no VGPR round trip, 1 KiB per wave-instruction).  The LDS
// image is lane-linear (hardware writes base + lane*16), so the bank swizzle chunk ^= (row & 7) is applied to the
// per-lane SOURCE address and again on the ds_read_b128 address (guide 5.4 rule 21): 128-B rows then read conflict-free.
// FOUR PHASES per K-tile with the two wave rows running half a phase apart (guide 5 "256^2 8-phase template": counted
// vmcnt, raw s_barrier, staggered wave groups).
//   * a phase = { ds_read fragments | issue one 16 KB half-tile of LDS-DMA } barrier { 16 MFMAs = one quadrant of the
//     wave's 128x64 tile over K = 64 } barrier.  Waves 4-7 execute one extra barrier up front, so while waves 0-3 are in
//     their MFMA cluster waves 4-7 (the other wave on each SIMD) read LDS / issue DMA, and vice versa: the matrix pipe and
//     the LDS / memory pipes are both busy all the time instead of all eight waves wanting the same pipe at once.
//   * all fragments of a K-tile are read in its first two phases (A rows 0-63 + W rows 0-31, then A rows 64-127 + W rows
//     32-63 of the wave's tile), so its LDS buffer is free from phase 3 on; tile t+2's A halves are staged there in phases
//     3 and 4 of tile t, its W halves in phases 1 and 2 of tile t+1.  The wait in phase 4 is the constant vmcnt(4): tile
//     t+1 has landed, the two A halves of tile t+2 stay in flight across the barriers.
//   * the last two K-tiles are peeled (nothing is staged past the end, the wait constants stay immediates).
// Every tile is full (the host only sends whole tile rows here: GemmArgs::nbm, tile_row0, tile_stride).
// PERSIST: gridDim.x workgroups walk the tiles v, v + gridDim.x, ...; the prologue DMA of the next tile (K-tile 0 and the
// A halves of K-tile 1 = PRO_DMA instructions per wave) is issued BEFORE this tile's epilogue, so its latency and the
// workgroup launch disappear under the epilogue's VALU work and stores.
constexpr int HBM_ = 256, HBN_ = 256;
constexpr int PRO_DMA = 12, PRO_DMA_KT0 = 8;        // LDS-DMA instructions per wave in that prologue / of them K-tile 0
// vector-memory instructions every wave issues in a FULL tile's epilogue, all of them behind the prologue DMA (vmcnt counts
// loads, stores and DMA together and retires in order): MT = 8 m-tiles x 4 rows of stores, plus as many residual loads.
template <int EPI> struct EpiVmem { static constexpr int min_ops = (EPI == EPI_RES || EPI == EPI_PATCH) ? 64 : 32; };

// LNX (RES / PATCH, image-aligned tiles): the LayerNorm behind this GEMM inside its epilogue (GemmArgs::ln_*).  Its LDS -- the
// four waves' row statistics, (mean, rstd) per row, a control word -- lies in the W half of buffer 1, which the next tile's
// prologue DMA (in flight during the epilogue) does not touch and nobody stages before phase 1 of the next K loop.
#ifdef HVLA_BENCH_HOOKS
// libhvla_bench.so (tools/lnx_stats.py): per workgroup id, summed over the launches since the last reset: [0] tiles, [1] shader-clock
// ticks of the whole epilogue, [2] ticks between "partials published" and "partners known", [3] tiles abandoned, [4] tiles
// normalised from memory by the image's last arriver, [5] ticks of [A] (x into registers), [6] of [B] (statistics, publish, drain),
// [7] of [D] + [E] (+ [F]) (mean / rstd, normalise, store h)
__device__ unsigned long long g_lnx_dbg[8][256];
#endif
constexpr int LNX_STAT = 98304, LNX_MR = LNX_STAT + 8192, LNX_CTRL = LNX_MR + 2048;   // [4][256][2] f32 | [256][2] f32 | 16 B
typedef __attribute__((address_space(3))) char lds_char;
template <typename Op, int EPI, bool PERSIST, bool LNX = false, bool NTOUT = false>
__global__ __launch_bounds__(512) void gemm256p_kernel(GemmArgs g) {
  static_assert(!NTOUT || EPI != EPI_PATCH, "NTOUT: non-temporal stores of the 16-bit outputs (QKV, GELU) / loads of the residual rows (RES)");
  using T = typename Op::elem;
  using X8 = typename Op::x8;
  static_assert(!LNX || EPI == EPI_RES || EPI == EPI_PATCH, "the fused LayerNorm follows a GEMM that writes the residual stream");
  extern __shared__ __attribute__((aligned(16))) char smem[];   // 2 buffers x (A 32 KB | W 32 KB)
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  const int nbm = g.nbm, nbn = (g.N + HBN_ - 1) / HBN_;     // N % 64 == 0; a last tile column narrower than 256 (DINOv2-small: N = 384, 1152)
  const int ntiles = nbm * nbn;                             // stages valid W rows again for the missing ones and its surplus waves store nothing
  const int GN = nbn % 4 == 0 ? 4 : (nbn % 3 == 0 ? 3 : (nbn % 2 == 0 ? 2 : 1));
  // virtual block id v (= blockIdx.x, + k * gridDim.x in the persistent form: gridDim.x is a multiple of 8, so a workgroup
  // stays in its XCD's id range) -> tile row tm, tile column origin n0.  Inside an XCD's contiguous id range: chunks of 8
  // tile rows, inside a chunk the N super-columns (GN tiles each) one after the other, so that the ~32 tiles an XCD runs at
  // once are an 8 x GN patch whose A and W K-slices share its 4 MiB L2.
  auto tile_origin = [&](int v, int& tm, int& n0) {
    const int q = ntiles / 8, r = ntiles % 8, xcd = v % 8;
    const int bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + v / 8;
    constexpr int CH = 8;
    const int per_chunk = CH * nbn;
    const int chunk = bid / per_chunk, rc = bid % per_chunk;
    const int rows = (nbm - chunk * CH) < CH ? (nbm - chunk * CH) : CH;
    const int sc = rc / (rows * GN), r2 = rc % (rows * GN);
    tm = chunk * CH + r2 / GN;
    n0 = (sc * GN + r2 % GN) * HBN_;
  };
  // LNX, persistent: the nbn column tiles of an image must run in the SAME round (they wait for each other's row statistics).
  // An XCD label (blockIdx % 8) has Wx = gridDim / 8 workgroups and nbm / 8 images; its first G nbn workgroups (G = Wx / nbn) form
  // G groups, group k takes image r G + k in round r, one column tile per member.  The Wx % nbn workgroups that are left over
  // (two of 32 at nbn = 3) take one image each per nbn rounds, its column tiles one after the other: all but the last of those
  // tiles are `later` -- abandoned at once (GemmArgs::ln_cnt) and normalised from memory when the last one is done.  The host
  // only asks for this form when the counts divide (run_encoder).
  // (no division here: the quotients by nbn go through the host's reciprocal GemmArgs::lnx_rcp = ceil(65536 / nbn), exact for the
  // few dozen workgroups per XCD / rounds there are; G, R / nbn and the images per XCD label come from the host as well)
  bool later = false;
  int rnd = 0;                                       // tiles this workgroup has started
  int pend_img = -1;                                 // LNX: the workgroup's previous tile has not counted itself in yet (image, its mark)
  uint32_t pend_mark = 0;
  auto tile_origin_x = [&](int& tm, int& n0, bool& lat) {
    typedef const __attribute__((address_space(4))) GemmArgs* karg_ptr;     // (re-read from the kernarg segment: not kept in SGPRs across the K loops)
    karg_ptr ga = (karg_ptr)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(ga));
    const int xcd = (int)blockIdx.x & 7, w = (int)blockIdx.x >> 3;
    const int G = ga->lnx_G, rcp = ga->lnx_rcp, wq = (w * rcp) >> 16, rq = (rnd * rcp) >> 16;
    int row, col;
    if (wq < G) {
      row = rnd * G + wq; col = w - wq * nbn; lat = false;
    } else {
      row = ga->lnx_rows_aligned + (w - G * nbn) * ga->lnx_rounds_div + rq; col = rnd - rq * nbn; lat = col != nbn - 1;
    }
    tm = xcd * ga->lnx_rows_xcd + row;
    n0 = col * HBN_;
    ++rnd;
  };
  int vb = blockIdx.x, tm, n0;
  if constexpr (LNX && PERSIST) tile_origin_x(tm, n0, later);
  else tile_origin(vb, tm, n0);
  int m0 = g.tile_row0 + tm * g.tile_stride;       // first global row of the tile
  const T* A = reinterpret_cast<const T*>(g.A);
  const T* W = reinterpret_cast<const T*>(g.W);
  // LDS-DMA pieces: instruction j of wave w fills rows [64 j + 8 w, +8) of A (or W); lane -> (row = lane >> 3, LDS chunk =
  // lane & 7), source chunk (lane & 7) ^ (row & 7).  Half-tile h = 0,1: A rows [128 h, +128) (j = 2h, 2h + 1); h = 2,3: W.
  // One per-lane byte offset for A and one for W plus wave-uniform (SGPR) row bases: the tile loop has no registers to
  // spare for per-piece 64-bit addresses.
  const int srow = wave * 8 + (lane >> 3);
  const int sch = ((lane & 7) ^ (lane >> 3)) * 8;
  const uint32_t lane_a = ((uint32_t)srow * (uint32_t)g.K + sch) * (uint32_t)sizeof(T);                 // bytes
  const uint32_t lane_w = ((uint32_t)(wperm(srow)) * (uint32_t)g.K + sch) * (uint32_t)sizeof(T);        // wperm(64 j + srow) = 64 j + wperm(srow)
  const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
  const T* abase = A + (size_t)m0 * g.K;
  const T* wbase = W + (size_t)n0 * g.K;
  int wn0 = n0;                                       // first column of the tile whose W rows `stage` is fetching
  const int KT = g.K / 64;
  auto stage = [&](auto hc, int kt) {              // half-tile hc of K-tile kt into buffer kt & 1
    constexpr int h = decltype(hc)::value;
    const int buf = kt & 1;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int j = 2 * (h & 1) + u;
      // SGPR row base + 32-bit per-lane byte offset, issued by hand: the builtin always takes a 64-bit VGPR address
      // (8 loop-invariant pairs that the register allocator spills, and every spill reload waits vmcnt(0)).  M0 (the LDS
      // destination) is compiler-reserved: it is saved, written, used and restored inside the one statement (guide 5.7;
      // an "m0" clobber is only a warning).  No "memory" clobber: it would make every issue wait for the fragment reads
      // in flight (barriers / counted waits order the DMA).
      const int jw = (h < 2 || wn0 + 64 * j < g.N) ? j : 0;           // (wave-uniform) a W row group past N: group 0 again, its products are dropped
      const T* sb = (h < 2 ? abase : wbase) + ((size_t)jw * 64 * g.K + kt * 64);
      const uint32_t dst = lds0 + (uint32_t)(buf * 65536 + wave * 1024 + (h >> 1) * 32768 + j * 8192);
      const uint32_t vo = h < 2 ? lane_a : lane_w;
      uint32_t keep;
      asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                   : "=&s"(keep) : "v"(vo), "s"(sb), "s"(dst));
    }
  };
  f32x4 acc[4][8];   // [n-tile][m-tile]
  const int fr = lane & 15, fq = lane >> 4;
  const int sw0 = ((fq) ^ (fr & 7)) << 4, sw1 = ((fq + 4) ^ (fr & 7)) << 4;
  const int a_off = (wm * 128 + fr) * 128, w_off = 32768 + (wn * 64 + fr) * 128;
  X8 fa0[8], fa1[8], fw0[4], fw1[4];               // [tile * 2 + kk]
  auto rd_a = [&](X8 (&f)[8], const char* lb, int ah) {
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      f[2 * t] = *reinterpret_cast<const X8*>(lb + a_off + (4 * ah + t) * 2048 + sw0);
      f[2 * t + 1] = *reinterpret_cast<const X8*>(lb + a_off + (4 * ah + t) * 2048 + sw1);
    }
  };
  auto rd_w = [&](X8 (&f)[4], const char* lb, int wh) {
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      f[2 * t] = *reinterpret_cast<const X8*>(lb + w_off + (2 * wh + t) * 2048 + sw0);
      f[2 * t + 1] = *reinterpret_cast<const X8*>(lb + w_off + (2 * wh + t) * 2048 + sw1);
    }
  };
  auto quad = [&](const X8 (&fa)[8], const X8 (&fw)[4], auto ahc, auto whc) {   // 16 MFMAs, 8 accumulators x 2 k-chunks
    constexpr int ah = decltype(ahc)::value, wh = decltype(whc)::value;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
      for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
          acc[2 * wh + nt][4 * ah + mt] = Op::mma16(fa[2 * mt + kk], fw[2 * nt + kk], acc[2 * wh + nt][4 * ah + mt]);
  };
  using H0 = std::integral_constant<int, 0>;
  using H1 = std::integral_constant<int, 1>;
  using H2 = std::integral_constant<int, 2>;
  using H3 = std::integral_constant<int, 3>;
#define HVLA_BAR() do { __builtin_amdgcn_sched_barrier(0); __builtin_amdgcn_s_barrier(); __builtin_amdgcn_sched_barrier(0); } while (0)
  // ---- prologue: tile 0 complete in buffer 0, the A halves of tile 1 in flight in buffer 1
  stage(H0{}, 0); stage(H1{}, 0); stage(H2{}, 0); stage(H3{}, 0);
  if (KT > 1) {
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
  while (true) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    HVLA_BAR();
    if (wm == 1) HVLA_BAR();                          // waves 4-7 run half a phase behind
    {
      int kt = 0;
      for (; kt + 2 < KT; ++kt) ktile(kt, Yes{}, Yes{});
      if (kt + 1 < KT) { ktile(kt, Yes{}, No{}); ++kt; }
      ktile(kt, No{}, No{});
    }
    if (wm == 0) HVLA_BAR();                          // same number of barriers in both wave rows
    if constexpr (LNX) {
      // the accumulators start new live ranges here: the register allocator then treats the K loop (at the register limit) and the
      // long epilogue below separately (without this it spilled four accumulator tiles INSIDE the last K-tile)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) asm volatile("" : "+v"(acc[i][j]));
    }
    const int cm0 = m0, cn0 = n0, ctm = tm;
    bool more = false;
    // bias row (per image when a corr table is given: an image-aligned tile row IS an image) / LayerScale of this tile
    // are fetched and WAITED FOR before the next tile's DMA goes out: the compiler does not see the hand-issued DMA, so a
    // wait it places after that point is a vmcnt(0) that would drain the prefetch
    const float* brow = (EPI != EPI_PATCH && g.corr) ? g.corr + ((size_t)ctm * 2 + wm) * g.N : g.bias;   // this wave row's half of the image
    const bool wvalid = cn0 + wn * 64 < g.N;          // (wave-uniform) this wave's 64 columns exist
    f32x4 pb4 = f32x4{0.f, 0.f, 0.f, 0.f}, pl4 = pb4;
    // LNX: the lane id comes from the hardware again (mbcnt) -- nothing per-lane of this long epilogue is a register across the K loops
    int lane_e = lane;
    if constexpr (LNX) asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lane_e));   // (volatile: recomputed per tile, not hoisted)
    const int fr_e = LNX ? (lane_e & 15) : fr;
    if (wvalid) {
      const uint32_t vcol = (uint32_t)(cn0 + wn * 64 + 4 * fr_e);     // 32-bit lane offset on a uniform base: no 64-bit lane address kept across the K loop
      pb4 = *reinterpret_cast<const f32x4*>(brow + vcol);
      pl4 = pb4;
      if constexpr (EPI == EPI_RES) pl4 = *reinterpret_cast<const f32x4*>(g.aux + vcol);
    }
    asm volatile("" : "+v"(pb4), "+v"(pl4));
    const bool clater = later;
    if constexpr (PERSIST) {
      vb += gridDim.x;
      more = vb < ntiles;
      if (more) {
        if constexpr (LNX) tile_origin_x(tm, n0, later);
        else tile_origin(vb, tm, n0);
        m0 = g.tile_row0 + tm * g.tile_stride;
        abase = A + (size_t)m0 * g.K;
        wbase = W + (size_t)n0 * g.K;
        wn0 = n0;
        __builtin_amdgcn_sched_barrier(0);            // the wait below counts the epilogue's stores as issued AFTER this DMA:
        stage(H0{}, 0); stage(H1{}, 0); stage(H2{}, 0); stage(H3{}, 0);
        stage(H0{}, 1); stage(H1{}, 1);
        __builtin_amdgcn_sched_barrier(0);            // nothing may be scheduled across it in either direction
      }
    }
    if constexpr (LNX) {
      // (below)
    } else if (!wvalid) {
      // nothing to store
    } else if constexpr (EPI == EPI_GELU) {
      if (g.colmean) {
        // mean rows of this tile's rounded outputs for the fc2 compensation, one per WAVE ROW (= half of the image): lane sums
        // its 32 rows (ascending, in the operand type), the four lanes that share the columns combine in f32 as
        // (fq0 + fq1) + (fq2 + fq3), and the wave writes its 64 columns of row (image, wave row) itself -- no LDS, no barrier.
        typename Op::x4 csh;
#pragma unroll
        for (int c = 0; c < 4; ++c) csh[c] = (T)0.f;
        gemm_epilogue_rows_impl<Op, EPI, 8, true, false, true, false, NTOUT>(acc, g, cm0 + wm * 128, cn0 + wn * 64, fr, fq, &pb4, &pl4, &csh);
        typename Op::x4 mo;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          float cs = (float)csh[c];
          cs += __shfl_xor(cs, 16, 64);
          cs += __shfl_xor(cs, 32, 64);
          mo[c] = (T)(cs * (1.f / 128.f));
        }
        if (fq == 0) *reinterpret_cast<typename Op::x4*>(reinterpret_cast<T*>(g.colmean) + ((size_t)ctm * 2 + wm) * g.N + cn0 + wn * 64 + 4 * fr) = mo;
      } else {
        gemm_epilogue_rows_impl<Op, EPI, 8, true, false, false, false, NTOUT>(acc, g, cm0 + wm * 128, cn0 + wn * 64, fr, fq, &pb4, &pl4);
      }
    } else {
      gemm_epilogue_rows_impl<Op, EPI, 8, true, false, false, false, NTOUT>(acc, g, cm0 + wm * 128, cn0 + wn * 64, fr, fq, &pb4, &pl4);
    }
    if constexpr (LNX) {
      // ---- the LayerNorm of this tile's 256 x 256 block of the new residual stream (GemmArgs::ln_*; canonical arithmetic: see
      // ln_lane_stats).  Barriers below are the plain kind: both wave rows are in step again.
      typedef __attribute__((address_space(3))) float lds_f;
      typedef __attribute__((address_space(3))) f32x4 lds_f4;
      typedef __attribute__((address_space(3))) uint32_t lds_u;
      // (opaque copies: what derives from them is computed inside this block, not in front of the tile loop -- hoisted, even the
      // constant LDS addresses below end up as spilled SGPRs)
      uint32_t lds_x = LNX_STAT;
      int nbn_x = nbn, wm_x = wm, wn_x = wn;
      asm volatile("" : "+s"(lds_x), "+s"(nbn_x), "+s"(wm_x), "+s"(wn_x));
      lds_f* stat = (lds_f*)((lds_char*)smem + lds_x);
      lds_f* mr = (lds_f*)((lds_char*)smem + lds_x + (LNX_MR - LNX_STAT));
      volatile lds_u* ctrl = (volatile lds_u*)((lds_char*)smem + lds_x + (LNX_CTRL - LNX_STAT));
      const int img = ctm, ct = cn0 / HBN_;
      // the arguments are re-read from the kernarg segment behind an opaque pointer: none of them is kept in an SGPR across the K loops
      typedef const __attribute__((address_space(4))) GemmArgs* karg_ptr;
      karg_ptr gp = (karg_ptr)__builtin_amdgcn_kernarg_segment_ptr();
      asm volatile("" : "+s"(gp));
      // an opaque copy of the thread id: nothing this block derives from it can be computed in front of the K loop and kept in
      // registers across it (the K loop has none to spare)
      int tid_x = wave * 64 + lane_e;
      asm volatile("" : "+v"(tid_x));
      const int lane_x = tid_x & 63, fr_x = lane_x & 15, fq_x = lane_x >> 4;
      const uint32_t row0 = (uint32_t)(img * gp->S + 1 + wm_x * 128 + 4 * fq_x);        // global row of this lane's (mt, r) = (0, 0)
      // the previous tile of this workgroup counts itself in now (see [F]): its stores completed during the K loop
      uint32_t pend_old = 0;
      if constexpr (PERSIST) {
        if (pend_img >= 0 && tid_x == 0)
          pend_old = __hip_atomic_fetch_add(gp->ln_cnt + pend_img, 1u + pend_mark, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      const float invE = 1.f / (float)gp->N;
#ifdef HVLA_BENCH_HOOKS
      const unsigned long long dbg_t0 = __builtin_readcyclecounter();
      unsigned long long dbg_wait = 0;
#endif
#define HVLA_LBAR() do { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory"); } while (0)
      // [A] x = residual + (acc + bias row) * layer scale (PATCH: position embedding + acc + bias) -> back into the accumulators
      // x and h are addressed as buffers: ONE per-lane_x offset (the lane_x's row (mt, r) = (0, 0), its four columns) plus a scalar
      // row offset per access -- no per-row address registers beside the 128 of xk
      const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(gp->out, 0, (int)gp->out_bytes, 0x00020000);
      const uint32_t lane_el = row0 * (uint32_t)gp->N + (uint32_t)(wn_x * 64 + 4 * fr_x);     // element offset of (row0, this lane_x's columns of a tile at column 0)
      f32x4 xk[8][4];                                    // [mt][r]: x of row 16 mt + 4 fq + r of the wave's 128, columns 4 fr .. 4 fr + 3 of its 64
      {                                                  // (every wave: one whose 64 columns do not exist computes on zeros / whatever the
                                                         // next row holds, counts as zeros below and stores nothing -- a branch here would
                                                         // make xk a phi of 128 zeros that are live beside the accumulators)
        // gemm_epilogue_rows_impl's arithmetic (RES: fma(acc + bias row, layer scale, x); PATCH: position + fma(acc, 1/256, bias)),
        // one m-tile (four rows) at a time, two m-tiles requested ahead: 8-12 loads in flight.  The accumulators of an m-tile die
        // where its four rows of xk are born (the K loop runs at the register limit: there is no room for both arrays).
        __amdgpu_buffer_rsrc_t irs = xrs;
        uint32_t ivo = (lane_el + (uint32_t)cn0) * 4u;
        if constexpr (EPI == EPI_PATCH) {
          irs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(gp->aux), 0, gp->S * gp->N * 4, 0x00020000);
          ivo = ((uint32_t)(1 + wm_x * 128 + 4 * fq_x) * (uint32_t)gp->N + (uint32_t)(cn0 + wn_x * 64 + 4 * fr_x)) * 4u;
        }
        const float q = gp->qscale;
        const float vmask = wvalid ? 1.f : 0.f;
        // (a ring of four: same step within the noise of two same-box runs -- out 1.665 / 1.740 against 1.680 / 1.697 ms -- once the spill
        // it caused was gone; five spills 16-23 registers in the persistent form.  The phase is not limited by the bytes in flight.)
        constexpr int XR = 3;
        f32x4 xin[XR][4];                                // a ring of XR m-tiles
        int rowb = gp->N * 4;                              // bytes per row; opaque at every site that forms the 32 scalar row offsets: shared, the
        asm volatile("" : "+s"(rowb));                   // compiler keeps all of them in SGPRs through the epilogue and spills a hundred others
        auto request = [&](int mt) {
#pragma unroll
          for (int r = 0; r < 4; ++r)
            xin[mt % XR][r] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(irs, (int)ivo, (16 * mt + r) * rowb, (EPI == EPI_RES && NTOUT) ? 2 : 0));   // aux 2 = nt
        };
#pragma unroll
        for (int mt = 0; mt < XR - 1; ++mt) request(mt);
#pragma unroll
        for (int mt = 0; mt < 8; ++mt) {
          if (mt + XR - 1 < 8) request(mt + XR - 1);
          f32x4 t[4];
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            if constexpr (EPI == EPI_PATCH) t[c] = __builtin_elementwise_fma(acc[c][mt], f32x4{q, q, q, q}, f32x4{pb4[c], pb4[c], pb4[c], pb4[c]});
            else t[c] = acc[c][mt] + pb4[c];
          }
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            f32x4 x = xin[mt % XR][r];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
              if constexpr (EPI == EPI_RES) x[c] = fmaf(t[c][r], pl4[c], x[c]);
              else x[c] += t[c][r];
            }
            xk[mt][r] = x;
            asm volatile("" : "+v"(xk[mt][r]));          // computed HERE (left alone, the arithmetic is sunk to its first use, far below,
          }                                              // and the 32 loads' destinations all stay live: spills)
          // [B] (sum, sum of squares) of these four rows over this wave's 64 columns -> LDS, here, under the loads of the next
          // m-tiles (this loop waits for memory with the VALU idle; as a phase of its own the reductions were 5 us per tile)
          {
            float sv[4], qv[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              ln_lane_stats(xk[mt][r], sv[r], qv[r]);
              sv[r] = row16_sum(sv[r]) * vmask;          // (x 1.0f is exact; x 0.0f of a finite number: an absent wave counts as zeros)
              qv[r] = row16_sum(qv[r]) * vmask;
            }
            if (fr_x == 0) {
              lds_f4* d = (lds_f4*)(stat + (wn_x * 256 + wm_x * 128 + 16 * mt + 4 * fq_x) * 2);
              d[0] = f32x4{sv[0], qv[0], sv[1], qv[1]};
              d[1] = f32x4{sv[2], qv[2], sv[3], qv[3]};
            }
          }
          __builtin_amdgcn_sched_barrier(0);             // (m-tile mt + 3 is not requested before this one is done: registers)
        }
      }
#ifdef HVLA_BENCH_HOOKS
      const unsigned long long dbg_tA = __builtin_readcyclecounter();
#endif
      // the four waves of a wave row -> the tile's partial of the row, published as ONE 16-byte entry {S, tag, Q, tag}
      // (write-through; tag = this launch's number in the call, the table is zeroed when the call starts): whoever reads an entry
      // with both tags right has the row's two sums
      HVLA_LBAR();
      typedef float f2 __attribute__((ext_vector_type(2)));
      const __amdgpu_buffer_rsrc_t prs = __builtin_amdgcn_make_buffer_rsrc(gp->ln_part, 0, (int)gp->part_bytes, 0x00020000);
      const uint32_t tag = gp->ln_tag;
      auto ent = [](u32x4 e) { const f32x4 f = __builtin_bit_cast(f32x4, e); return f2{f[0], f[2]}; };   // (sum, sum of squares) of an entry
      // thread t and thread t + 256 both hold the tile's partial of row t: the lower wave row publishes it, the upper one reads
      // the other tiles' (a wave that has just stored gets no load back before its store has completed: vmcnt retires in order)
      const int prow = tid_x & 255;
      f2 mine;
      {
        typedef __attribute__((address_space(3))) f2 lds_f2;
        const f2 p0 = *(lds_f2*)(stat + (0 * 256 + prow) * 2), p1 = *(lds_f2*)(stat + (1 * 256 + prow) * 2);
        const f2 p2 = *(lds_f2*)(stat + (2 * 256 + prow) * 2), p3 = *(lds_f2*)(stat + (3 * 256 + prow) * 2);
        mine = (p0 + p1) + (p2 + p3);
      }
      if (tid_x < 256) {
        // (as a float vector: u32x4{bit_cast(mine[0]), tag, bit_cast(mine[1]), tag} came out of hipcc 7.2 with mine[0] twice)
        const float tagf = __builtin_bit_cast(float, tag);
        const f32x4 ef = f32x4{mine[0], tagf, mine[1], tagf};
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, ef), prs, (int)((uint32_t)((img * nbn_x + ct) * 256 + prow) * 16u), 0, 16);   // aux 16 = sc1
      }
      // scale / bias of the LayerNorm for this lane's four columns (the residual rows' registers are free again)
      f32x4 lg4 = f32x4{0.f, 0.f, 0.f, 0.f}, lb4 = lg4;
      if (wvalid) {
        const uint32_t vcol = (uint32_t)(cn0 + wn_x * 64 + 4 * fr_x);
        lg4 = *reinterpret_cast<const f32x4*>(gp->ln_scale + vcol);
        lb4 = *reinterpret_cast<const f32x4*>(gp->ln_bias + vcol);
      }
      const __amdgpu_buffer_rsrc_t hrs = __builtin_amdgcn_make_buffer_rsrc(gp->ln_out, 0, (int)(gp->out_bytes / 4u * (uint32_t)sizeof(T)), 0x00020000);
      auto store_x = [&]() {                             // the new residual rows, write-through: an abandoned tile is read back by another workgroup
        if (!wvalid) return;
        const uint32_t vo = (lane_el + (uint32_t)cn0) * 4u;
        int rowb = gp->N * 4;
        asm volatile("" : "+s"(rowb));
#pragma unroll
        for (int mt = 0; mt < 8; ++mt)
#pragma unroll
          for (int r = 0; r < 4; ++r)
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, xk[mt][r]), xrs, (int)vo, (16 * mt + r) * rowb,
                                                   // aux 16 = sc1 (write-through: the route through memory reads these rows), + 2 = nt at a big batch: the
                                                   // 202 MB of x are next read a GEMM later and do not fit beside anything; streamed, they leave the
                                                   // 101 MB of h this epilogue also writes -- the next GEMM's A operand -- in the memory-side cache
                                                   // (same box: step 14.37 / 14.44 -> 14.23 / 14.28 ms; fc2 3.70 -> 3.62, QKV 2.50 -> 2.48)
                                                   (NTOUT ? 18 : 16));
      };
#ifdef HVLA_BENCH_HOOKS
      const unsigned long long dbg_tB = __builtin_readcyclecounter();
#endif
      // [C] threads 256-511 (the upper wave row) read the other column tiles' entries of their row until the tags are right -- a few
      // microseconds of skew between workgroups that started together -- or the bound runs out / the partners are known to be a
      // round away (`later`): then this tile does not wait (its mark in [F]).  The lower wave row stores x meanwhile, and wave 0
      // normalises this tile's 256 columns of the image's CLS row (x[img * S] was finished by an earlier launch).
      if (wave >= 4) {
        bool ok = !clater;
        f2 t = mine;
        if (ok) {
          const uint64_t t0 = __builtin_amdgcn_s_memrealtime();
          const uint32_t eo = (uint32_t)(img * nbn_x * 256 + prow) * 16u;
          while (true) {
            u32x4 pe[4];
#pragma unroll
            for (int c = 0; c < 4; ++c)
              if (c < nbn_x) pe[c] = __builtin_amdgcn_raw_buffer_load_b128(prs, (int)eo, c * 4096, 16);      // (this tile's own entry may not have landed: `mine`)
            ok = true;
#pragma unroll
            for (int c = 0; c < 4; ++c)
              if (c < nbn_x && c != ct) ok = ok && pe[c][1] == tag && pe[c][3] == tag;
            if (__builtin_amdgcn_ballot_w64(ok) == ~0ull) {
              f2 pv[4];
#pragma unroll
              for (int c = 0; c < 4; ++c) pv[c] = c == ct ? mine : ent(pe[c]);
              t = pv[0];                                                                               // column tiles in ascending order
#pragma unroll
              for (int c = 1; c < 4; ++c)
                if (c < nbn_x) t += pv[c];
              break;
            }
            if (__builtin_amdgcn_s_memrealtime() - t0 > (uint64_t)gp->ln_spin) break;
            __builtin_amdgcn_s_sleep(1);
          }
        }
        const bool wave_ok = __builtin_amdgcn_ballot_w64(ok) == ~0ull;
        if (wave_ok) {
          float mean, rstd;
          ln_finish(t[0], t[1], invE, mean, rstd);
          *(__attribute__((address_space(3))) f2*)(mr + prow * 2) = f2{mean, rstd};
        }
        if (lane_x == 0) ctrl[wave - 4] = wave_ok ? 1u : 0u;
#ifdef HVLA_BENCH_HOOKS
        if (tid_x == 256) ctrl[6] = (uint32_t)(__builtin_readcyclecounter() - dbg_tB);
#endif
      } else {
        store_x();
        if (wave == 0) {
          const int n4 = gp->N / 4;
          const f32x4* xr = reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(gp->out) + (size_t)img * gp->S * gp->N);
          f32x4 cur[4];
#pragma unroll
          for (int i = 0; i < 4; ++i) cur[i] = lane_x + 64 * i < n4 ? xr[lane_x + 64 * i] : f32x4{0.f, 0.f, 0.f, 0.f};
          float mean, rstd;
          ln_row_stats<4>(cur, invE, mean, rstd);
          const f32x4 mv = ct == 0 ? cur[0] : (ct == 1 ? cur[1] : (ct == 2 ? cur[2] : cur[3]));
          const int col = lane_x + 64 * ct;
          if (col < n4) {
            const f32x4 s4 = reinterpret_cast<const f32x4*>(gp->ln_scale)[col], b4 = reinterpret_cast<const f32x4*>(gp->ln_bias)[col];
            typename Op::x4 o;
#pragma unroll
            for (int j = 0; j < 4; ++j) o[j] = (T)ln_value(mv[j], mean, rstd, s4[j], b4[j]);
            reinterpret_cast<typename Op::x4*>(reinterpret_cast<T*>(gp->ln_out) + (size_t)img * gp->S * gp->N)[col] = o;
          }
        }
      }
      asm volatile("" : "+v"(lg4), "+v"(lb4));
      HVLA_LBAR();
      const bool go = (ctrl[0] & ctrl[1] & ctrl[2] & ctrl[3]) != 0u;   // every row of the tile has its statistics
      if (wave >= 4) store_x();
#ifdef HVLA_BENCH_HOOKS
      dbg_wait = ctrl[6];
      unsigned long long dbg_slow = 0;
      const unsigned long long dbg_tC = __builtin_readcyclecounter();
#endif
      // [E] normalise from the registers: h (16-bit) and this wave's 64 columns of the wave row's mean row.  `im`: the image the
      // rows in xk and (mean, rstd) in LDS belong to.
      auto sweep = [&](int im, int ctile, f32x4 gm4, f32x4 bt4) {
        const uint32_t ncol = (uint32_t)(ctile * HBN_ + wn_x * 64 + 4 * fr_x);
        const uint32_t vo = ((uint32_t)(im * gp->S + 1 + wm_x * 128 + 4 * fq_x) * (uint32_t)gp->N + ncol) * (uint32_t)sizeof(T);
        int rowb = gp->N * (int)sizeof(T);
        asm volatile("" : "+s"(rowb));
        f32x4 cs = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int mt = 0; mt < 8; ++mt) {
          const lds_f4* mp = (const lds_f4*)(mr + (wm_x * 128 + 16 * mt + 4 * fq_x) * 2);
          const f32x4 m01 = mp[0], m23 = mp[1];
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float mean = r < 2 ? m01[2 * r] : m23[2 * r - 4], rstd = r < 2 ? m01[2 * r + 1] : m23[2 * r - 3];
            typename Op::x4 o;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
              const float y = ln_value(xk[mt][r][c], mean, rstd, gm4[c], bt4[c]);
              cs[c] = (mt == 0 && r == 0) ? y : cs[c] + y;                     // ascending rows
              o[c] = (T)y;
            }
            __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, o), hrs, (int)vo, (16 * mt + r) * rowb, 0);
          }
          __builtin_amdgcn_sched_barrier(0);
        }
        if (gp->ln_abar) {
          typename Op::x4 mo;
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            float t = cs[c];
            t = xor32_sum(xor16_sum(t));                                       // (p0 + p1) + (p2 + p3)
            mo[c] = (T)(t * (1.f / 128.f));
          }
          if (fq_x == 0) *reinterpret_cast<typename Op::x4*>(reinterpret_cast<T*>(gp->ln_abar) + ((size_t)im * 2 + wm_x) * gp->N + ncol) = mo;
        }
      };
      if (go && wvalid) sweep(img, ct, lg4, lb4);
      // [F] book-keeping, off the critical path.  Every tile adds itself ONCE to its image's word (bits 0-15: tiles arrived; bit
      // 16 + c: column tile c did not wait) when its stores -- partial, x, h -- have completed: in the persistent form at the start
      // of the workgroup's NEXT epilogue (a whole K loop later: nothing to wait for), its answer read at the end of that epilogue; a
      // workgroup's last tile drains and adds at once.  The tile whose add completes the image finds the marks of the tiles that
      // did not wait and normalises them from memory (their x and every partial are there: each add follows its tile's stores).
      auto from_memory = [&](int im, uint32_t bits) {      // workgroup-uniform; the image's (mean, rstd) again, then the marked tiles
        if (wave == 0) {
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        HVLA_LBAR();
        if (tid_x < 256) {
          const uint32_t eo = (uint32_t)(im * nbn_x * 256 + tid_x) * 16u;
          u32x4 pe[4];
#pragma unroll
          for (int c = 0; c < 4; ++c)
            if (c < nbn_x) pe[c] = __builtin_amdgcn_raw_buffer_load_b128(prs, (int)eo, c * 4096, 0);
          f2 t = ent(pe[0]);
#pragma unroll
          for (int c = 1; c < 4; ++c)
            if (c < nbn_x) t += ent(pe[c]);
          float mean, rstd;
          ln_finish(t[0], t[1], invE, mean, rstd);
          *(__attribute__((address_space(3))) f2*)(mr + tid_x * 2) = f2{mean, rstd};
        }
        HVLA_LBAR();
        for (int c = 0; c < nbn_x; ++c) {
          if (!((bits >> (16 + c)) & 1u) || c * HBN_ + wn_x * 64 >= gp->N) continue;       // (wave-uniform)
          const uint32_t ncol = (uint32_t)(c * HBN_ + wn_x * 64 + 4 * fr_x);
          const f32x4 gm4 = *reinterpret_cast<const f32x4*>(gp->ln_scale + ncol), bt4 = *reinterpret_cast<const f32x4*>(gp->ln_bias + ncol);
          // the rows stream through a ring of three m-tiles (as in [A]): the loads of m-tile mt + 2 are in flight while m-tile mt is
          // normalised and stored -- this route runs on ONE CU at the end of a launch, and as "all rows in, then the sweep" it was two
          // exposed memory latencies per tile.  The arithmetic and the order of the column sums are sweep()'s.
          {
            const uint32_t vox = ((uint32_t)(im * gp->S + 1 + wm_x * 128 + 4 * fq_x) * (uint32_t)gp->N + ncol) * 4u;
            const uint32_t voh = ((uint32_t)(im * gp->S + 1 + wm_x * 128 + 4 * fq_x) * (uint32_t)gp->N + ncol) * (uint32_t)sizeof(T);
            int rowbx = gp->N * 4, rowbh = gp->N * (int)sizeof(T);
            asm volatile("" : "+s"(rowbx), "+s"(rowbh));
            f32x4 ring[3][4];
            auto req = [&](int mt) {
#pragma unroll
              for (int r = 0; r < 4; ++r)
                ring[mt % 3][r] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xrs, (int)vox, (16 * mt + r) * rowbx, 0));
            };
            req(0);
            req(1);
            f32x4 cs = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int mt = 0; mt < 8; ++mt) {
              if (mt + 2 < 8) req(mt + 2);
              const lds_f4* mp = (const lds_f4*)(mr + (wm_x * 128 + 16 * mt + 4 * fq_x) * 2);
              const f32x4 m01 = mp[0], m23 = mp[1];
#pragma unroll
              for (int r = 0; r < 4; ++r) {
                const float mean = r < 2 ? m01[2 * r] : m23[2 * r - 4], rstd = r < 2 ? m01[2 * r + 1] : m23[2 * r - 3];
                const f32x4 xv = ring[mt % 3][r];
                typename Op::x4 o;
#pragma unroll
                for (int cc = 0; cc < 4; ++cc) {
                  const float y = ln_value(xv[cc], mean, rstd, gm4[cc], bt4[cc]);
                  cs[cc] = (mt == 0 && r == 0) ? y : cs[cc] + y;                 // ascending rows
                  o[cc] = (T)y;
                }
                __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, o), hrs, (int)voh, (16 * mt + r) * rowbh, 0);
              }
              __builtin_amdgcn_sched_barrier(0);
            }
            if (gp->ln_abar) {
              typename Op::x4 mo;
#pragma unroll
              for (int cc = 0; cc < 4; ++cc) {
                float t = cs[cc];
                t = xor32_sum(xor16_sum(t));                                     // (p0 + p1) + (p2 + p3)
                mo[cc] = (T)(t * (1.f / 128.f));
              }
              if (fq_x == 0) *reinterpret_cast<typename Op::x4*>(reinterpret_cast<T*>(gp->ln_abar) + ((size_t)im * 2 + wm_x) * gp->N + ncol) = mo;
            }
          }
#ifdef HVLA_BENCH_HOOKS
          ++dbg_slow;
#endif
        }
        if (tid_x == 0) __hip_atomic_fetch_and(gp->ln_cnt + im, 0xffffu, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // the marks, for the next launch
      };
      auto settle = [&](int im, uint32_t mark, uint32_t old) {   // workgroup-uniform: `old` = what the add of tile (im, mark) returned (wave 0)
        if (wave == 0) {
          old = __builtin_amdgcn_readfirstlane(old);
          const uint32_t bits = (old + 1u + mark);
          if (lane_x == 0) ctrl[4] = ((bits & 0xffffu) >= gp->ln_target && (bits >> 16)) ? bits : 0u;
        }
        HVLA_LBAR();
        const uint32_t bits = ctrl[4];
        HVLA_LBAR();                                       // (read by every wave before it is written again)
        if (bits) from_memory(im, bits);
      };
      const uint32_t my_mark = go ? 0u : (0x10000u << ct);
      if constexpr (PERSIST) {
        if (pend_img >= 0) settle(pend_img, pend_mark, pend_old);
        pend_img = img;
        pend_mark = my_mark;
      }
      if (!PERSIST || !more) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // this tile's partial, x and h have completed
        HVLA_LBAR();
        uint32_t old = 0;
        if (tid_x == 0) old = __hip_atomic_fetch_add(gp->ln_cnt + img, 1u + my_mark, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        settle(img, my_mark, old);
      }
#ifdef HVLA_BENCH_HOOKS
      if (tid_x == 0 && blockIdx.x < 256) {
        atomicAdd(&g_lnx_dbg[0][blockIdx.x], 1ull);
        atomicAdd(&g_lnx_dbg[1][blockIdx.x], (unsigned long long)__builtin_readcyclecounter() - dbg_t0);
        atomicAdd(&g_lnx_dbg[2][blockIdx.x], dbg_wait);
        if (!go) atomicAdd(&g_lnx_dbg[3][blockIdx.x], 1ull);
        if (dbg_slow) atomicAdd(&g_lnx_dbg[4][blockIdx.x], dbg_slow);
        atomicAdd(&g_lnx_dbg[5][blockIdx.x], dbg_tA - dbg_t0);
        atomicAdd(&g_lnx_dbg[6][blockIdx.x], dbg_tB - dbg_tA);
        atomicAdd(&g_lnx_dbg[7][blockIdx.x], (unsigned long long)__builtin_readcyclecounter() - dbg_tC);
      }
#endif
#undef HVLA_LBAR
      if (!more) break;
      // K-tile 0 of the next tile must have landed: its DMA is older than everything this epilogue issued -- 32 loads and 32 stores
      // of x at least in a wave that has columns
      if (wvalid) asm volatile("s_waitcnt vmcnt(63)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      continue;
    }
    if (!more) break;
    // K-tile 0 of the next tile must have landed: its PRO_DMA_KT0 instructions are the oldest of the PRO_DMA + (epilogue)
    // operations issued since, so at most (PRO_DMA - PRO_DMA_KT0) + min_ops may still be outstanding
    // (the 6-bit immediate caps the count at 63, which only makes the wait stricter)
    constexpr int WAITN = (PRO_DMA - PRO_DMA_KT0) + EpiVmem<EPI>::min_ops;
    static_assert(EpiVmem<EPI>::min_ops >= 32, "an epilogue issues at least one store per row of the wave tile");
    static_assert(WAITN >= 63 || WAITN == 36, "the s_waitcnt immediates below are written for these two counts");
    if (!wvalid) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PRO_DMA - PRO_DMA_KT0) : "memory");     // (no epilogue operations behind the DMA)
    else if constexpr (WAITN >= 63) asm volatile("s_waitcnt vmcnt(63)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(36)" ::: "memory");
  }
#undef HVLA_BAR
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
constexpr int APS = 68;           // floats per partial of the last query: max, sum, 2 unused, O[64] (16-byte aligned)
constexpr int AVLD = 64;          // V row stride (halves): 128 B, 64-B halves swapped on rows with bit 1 set
// K rows are 128 B = half of the 64 banks, so a row's bank half is its row parity, and the 16-B chunk swizzle (chunk ^ key & 7)
// alone leaves every ds_read_b128 lane group ({0-3, 12-15, 20-27}, {4-11, 16-19, 28-31}: guide, LDS table) with two keys
// per 4-bank slot (SQ_LDS_BANK_CONFLICT was 27 % of the LDS cycles of this kernel).  Storing key k in row kperm(k) -- bits 0
// and 3 exchanged -- makes the bank half bit 3 of the key, which differs inside every such pair: conflict-free.
__device__ __forceinline__ constexpr int kperm(int k) { return (k & ~9) | ((k & 1) << 3) | ((k >> 3) & 1); }

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

// omean (nullable): [B][E] 16-bit mean over all S tokens of the image of the output, the operand of the out-projection's
// weight-rounding compensation (the corr rows of gemm64_kernel): the workgroup owns every row of its 64 columns.
// NWC: the number of 32-query waves as a compile-time constant (8 at S = 257: the key-tile loops unroll, their LDS addresses become
// immediate offsets), 0 = taken from S at run time (the small test geometries).
template <typename Op, bool AMAP = false, int NWC = 0>      // AMAP: the opt-in instantiation that also exports the CLS query's attention row
__global__ void attention_kernel(const typename Op::elem* __restrict__ qkv, typename Op::elem* __restrict__ o,
                                 int S, int E, int H, typename Op::elem* __restrict__ omean,
                                 float* __restrict__ amap     // nullable: this layer's slice [B][..][H][S - 1] of the CLS query's attention over the patch keys
                                 , int amap_stride            // floats between two images in amap (= layers * H * (S - 1))
#ifdef HVLA_BENCH_HOOKS
                                 , unsigned long long* stamps = nullptr      // libhvla_bench.so: shader-clock stamps of workgroup `stamp_wg`, wave 0
                                 , int stamp_wg = 0
#endif
                                 ) {
#ifdef HVLA_BENCH_HOOKS
  int nstamp = 0;
#define HVLA_ASTAMP() do { if (stamps && (int)blockIdx.x == stamp_wg && threadIdx.x == 0) stamps[nstamp] = __builtin_readcyclecounter(); ++nstamp; } while (0)
#else
#define HVLA_ASTAMP() do { } while (0)
#endif
  HVLA_ASTAMP();                                           // 0 start
  // S = 32 * NW + 1 tokens.  NW waves of 64 lanes: wave w owns queries [32 w, 32 w + 32) on the matrix cores;
  // the one remaining query (the last token) is done co-operatively on the VALU, wave w taking key tile w,
  // and combined through LDS.  8 waves per workgroup at S = 257 (2 per SIMD) so that two workgroups share a
  // CU (the 9-wave version only ever had one resident).
  using T = typename Op::elem;
  using X8 = typename Op::x8;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int NW = NWC ? NWC : (S - 1) / 32, KT = NW + 1, SP = KT * 32;
  T* Ks = reinterpret_cast<T*>(smem);                      // [SP][64]  keys, 16-B chunks XOR-swizzled by key & 7
  T* Vs = Ks + SP * AVLD;                                  // [SP][64]  values, row-major (read transposed)
  T* qxs = Vs + SP * AVLD;                                 // [64]       the last query (parked here, not in registers)
  float* part = reinterpret_cast<float*>(qxs + 64);        // [KT][APS]  partial (max, sum, -, -, O[64]) of the last query
  float* csum = part + KT * APS;                            // [NW][64]   per-wave column sums of the output
  float* clsm = csum + NW * 64;                            // [32]       AMAP only: the running maximum every key tile's entries of clsrow are relative to
  float* clsrow = clsm + 32;                               // [SP]       AMAP only: the CLS query's unnormalised probabilities (a region of
                                                           // its own: `part` is written by the other waves' tails while wave 0 may still be in its pass over the keys)
  const int b = blockIdx.x / H, head = blockIdx.x % H;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, nthr = blockDim.x;
  const size_t rowstride = (size_t)3 * E;
  const T* base = qkv + (size_t)b * S * rowstride + head * 64;
  // ---- this wave's 32 queries as B fragments (natural d order): 4 k-steps of 16 -- requested first, used last
  const int col = lane & 31, half = lane >> 5;
  const int q = wave * 32 + col;                           // always < S - 1
  X8 qf[4];
  constexpr bool STREAM = NWC == 8 && !AMAP;               // S = 257, 512 threads: K / V streamed in five 64-key chunks by LDS-DMA (below)
  if constexpr (STREAM) {
    // requested by hand: an ordinary load in flight beside the LDS-DMAs below makes the compiler wait with vmcnt(0) -- for every
    // chunk -- before the first score MFMA (it does not count through the DMA builtins).  The registers are written by the
    // hardware behind the compiler's back until the counted wait of the first hand-over, which is tied to them ("+v").
    const T* qp = base + (size_t)q * rowstride + half * 8;
    asm volatile("global_load_dwordx4 %0, %4, off\n\tglobal_load_dwordx4 %1, %4, off offset:32\n\t"
                 "global_load_dwordx4 %2, %4, off offset:64\n\tglobal_load_dwordx4 %3, %4, off offset:96"
                 : "=&v"(qf[0]), "=&v"(qf[1]), "=&v"(qf[2]), "=&v"(qf[3]) : "v"(qp) : "memory");
  } else {
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) qf[ks] = *reinterpret_cast<const X8*>(base + (size_t)q * rowstride + ks * 16 + half * 8);
  }
  // the extra query (token S-1) goes to LDS: 16 VGPRs less across the MFMA loops (the kernel runs at the 128-VGPR limit
  // of 4 waves per SIMD)
  if constexpr (!STREAM) { if (tid < 8) *reinterpret_cast<X8*>(qxs + tid * 8) = *reinterpret_cast<const X8*>(base + (size_t)(S - 1) * rowstride + tid * 8); }
  // ---- STREAM (round 6): K and V straight into their LDS images by LDS-DMA, chunk by chunk -- K0 V0 K1 V1 ... (64 keys each: one
  // 16-byte piece per thread and image), all requested at once; the key tiles of chunk c are worked on behind `vmcnt(pieces issued
  // after chunk c)` + one barrier, while the later chunks are still on their way: the first score MFMA waits for 16 KB, not for
  // 132 KB (the one-pass softmax needs nothing from later keys).  No staging registers (40 VGPRs less at the head of the kernel).
  // The images' permutations are applied on the SOURCE side: LDS row rho of a chunk takes key kperm(rho) (K) / rho (V), its 16-byte
  // slot gamma the source piece gamma ^ (key & 7) (K) / gamma ^ 4 on rows with bit 1 set (V).  Chunk 4 is the single key 256: its
  // K row by lanes 0-7 of wave 0, its V row by lanes 0-7 of wave 4 (one partial-EXEC instruction each; vmcnt is counted per wave),
  // the padding rows 257 .. 287 of both images are zeroed by ordinary stores up front.
  X8 qlast;                                                // STREAM: the last query, wave 0's LAST request (stored behind the final hand-over)
  auto stream_wait = [&](auto rest) {                      // rest = pieces of chunks 1-3 that may stay in flight; wave 4: + its piece of chunk 4; wave 0: + its piece and the last query; -1: nothing at all
    constexpr int R = decltype(rest)::value;
    if (R >= 0 && wave == 0) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(R < 0 ? 0 : R + 2) : "memory");
    else if (R >= 0 && wave == 4) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(R < 0 ? 0 : R + 1) : "memory");
    else asm volatile("s_waitcnt vmcnt(%0)" :: "n"(R < 0 ? 0 : R) : "memory");
    // the queries' registers are DEFINED here for the compiler (one statement behind all three waits: a copy it makes for this tie
    // reads them after the data has landed; tied inside the branches, the wave-4 path got its copies IN FRONT of the wait)
    asm volatile("" : "+v"(qf[0]), "+v"(qf[1]), "+v"(qf[2]), "+v"(qf[3]));
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");                         // (nothing below may be placed in front of the barrier: see gemm64_body)
    __builtin_amdgcn_sched_barrier(0);
  };
  if constexpr (STREAM) {
    __builtin_amdgcn_sched_barrier(0);                     // the queries' loads are requested FIRST (vmcnt retires in order: their wait must not cover a chunk)
    const int rho = tid >> 3, slot = tid & 7;              // this thread's LDS row inside a chunk (wave w: rows 8 w .. 8 w + 7) and 16-byte slot
    const int ksrc = kperm(rho);
    // byte offsets from `base` (uniform: an SGPR pair) of this thread's piece of chunk 0
    const uint32_t ko = (uint32_t)(((size_t)ksrc * rowstride + E + ((slot ^ (ksrc & 7)) * 8)) * sizeof(T));
    const uint32_t vo = (uint32_t)(((size_t)rho * rowstride + 2 * E + ((slot ^ (((rho >> 1) & 1) << 2)) * 8)) * sizeof(T));
    const uint32_t cbytes = (uint32_t)(64 * rowstride * sizeof(T));       // between two chunks
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
    const uint32_t kl = __builtin_amdgcn_readfirstlane(lds0 + wave * 1024), vl = kl + (uint32_t)(SP * AVLD * sizeof(T));   // (wave-uniform: M0)
    // issued by hand (SGPR base + 32-bit lane offset, M0 saved / written / restored inside the statement, as in gemm64c_kernel): behind
    // the BUILTIN the compiler waits with vmcnt(0) in front of the first ds_read_b64_tr_b16 of V (it cannot tell the images apart)
    auto dma = [&](uint32_t off, uint32_t dst) {
      uint32_t keep;
      asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2 nt\n\ts_mov_b32 m0, %0"
                   : "=&s"(keep) : "v"(off), "s"(base), "s"(dst) : "memory");     // nt: K and V are read once per step
    };
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      dma(ko + c * cbytes, kl + c * 8192);
      dma(vo + c * cbytes, vl + c * 8192);
    }
    if (wave == 0) {
      if (lane < 8) {
        dma(ko + 4 * cbytes, kl + 4 * 8192);
        qlast = *reinterpret_cast<const X8*>(base + (size_t)(S - 1) * rowstride + tid * 8);
      }
    } else if (wave == 4) {
      // V row 256, slots 0 .. 7 (no half swap: bit 1 of the row is clear); wave 4's own rows are 32 .. 39, so neither vo nor vl fit
      if (lane < 8) dma((uint32_t)(((size_t)256 * rowstride + 2 * E + slot * 8) * sizeof(T)), vl - 4 * 1024 + 4 * 8192);
    }
    if (tid >= 8 && tid < 256) {                           // rows 257 .. 287 of both images: the padding keys
      X8 z;
#pragma unroll
      for (int j = 0; j < 8; ++j) z[j] = (T)0.f;
      *reinterpret_cast<X8*>(Ks + (256 + rho) * AVLD + slot * 8) = z;
      *reinterpret_cast<X8*>(Vs + (256 + rho) * AVLD + slot * 8) = z;
    }
  }
  // ---- (not STREAM) stage K and V: 16 B per thread and chunk (zero rows for the padding keys).  SP * 8 / nthr = 4 (NW + 1) / NW <= 8
  // chunks per thread; ALL their loads are requested before the first LDS store (a rolled loop pays one HBM round trip
  // per iteration: 5 in a row at S = 257, a third of the workgroup's life time).
  // K first, then V.
  constexpr int STG = STREAM ? 0 : 8;
  X8 kreg[STG ? STG : 1], vreg[STG ? STG : 1];
  // Round 5: the kernel is bound by instruction issue (four waves per SIMD, ~2 500 instructions per wave and item), and a fifth
  // of them were this staging: per chunk a 64-bit address, a zero fill, two compares and a branch.  Now one 32-bit offset per
  // thread into a buffer resource that ends with the image's last row: chunk `it` is 64 keys further on (a scalar offset), K
  // and V are E and 2 E elements behind q (scalar too), the padding keys are out of the resource's range and come back as
  // zeros, and the LDS addresses of a thread's chunks differ by constants.  (The range check of a raw buffer on gfx9 does not
  // see the scalar offset, so the padding chunks get their out-of-range offset in the VGPR.)
  const __amdgpu_buffer_rsrc_t krs = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(base), 0, (int)((size_t)S * rowstride * sizeof(T)) - head * 128, 0x00020000);
  const int key0 = tid >> 3;
  const uint32_t kvo = (uint32_t)key0 * (uint32_t)(rowstride * sizeof(T)) + (tid & 7) * 16;
  const int kround = nthr >> 3;                                                     // keys per round of chunks
  const uint32_t kstep = (uint32_t)kround * (uint32_t)(rowstride * sizeof(T));      // (scalar) bytes between a thread's chunks
  auto chunk_vo = [&](int it) -> int { return key0 < S - it * kround ? (int)kvo : 0x7fff0000; };   // chunk `it` of this thread: key key0 + it * kround
#pragma unroll
  for (int it = 0; it < STG; ++it) {
    if (it * nthr >= SP * 8) break;                        // uniform
    // (K and V are read exactly once per step: non-temporal loads, 1.88 -> 1.84 ms per step)
    kreg[it] = __builtin_bit_cast(X8, __builtin_amdgcn_raw_buffer_load_b128(krs, chunk_vo(it), (int)(it * kstep) + E * (int)sizeof(T), 2));        // aux 2 = nt
  }
#pragma unroll
  for (int it = 0; it < STG; ++it) {
    if (it * nthr >= SP * 8) break;
    vreg[it] = __builtin_bit_cast(X8, __builtin_amdgcn_raw_buffer_load_b128(krs, chunk_vo(it), (int)(it * kstep) + 2 * E * (int)sizeof(T), 2));
  }
  // LDS side: chunk `it` lies kround rows behind chunk it - 1 in both images when kround is a multiple of 16 (kperm() exchanges
  // bits 0 and 3 of the key, the swizzles use bits 0-2); one workgroup wave (the tiny test geometries) takes the general form.
  const int ch0 = tid & 7;
  T* const kdst = Ks + kperm(key0) * AVLD + ((ch0 ^ (key0 & 7)) * 8);
  T* const vdst = Vs + key0 * AVLD + ((ch0 ^ (((key0 >> 1) & 1) << 2)) * 8);
  const bool klin = (kround & 15) == 0;                    // uniform
  if (klin) {
#pragma unroll
    for (int it = 0; it < STG; ++it) {
      if (it * nthr >= SP * 8) break;
      if (key0 + it * kround < SP) *reinterpret_cast<X8*>(kdst + it * kround * AVLD) = kreg[it];
    }
  } else {
#pragma unroll
    for (int it = 0; it < STG; ++it) {
      if (it * nthr >= SP * 8) break;
      const int key = key0 + it * kround;
      if (key < SP) *reinterpret_cast<X8*>(Ks + kperm(key) * AVLD + ((ch0 ^ (key & 7)) * 8)) = kreg[it];
    }
  }
  HVLA_ASTAMP();                                           // 1 K staged / STREAM: every request issued
  auto stage_v = [&] {
#pragma unroll
    for (int it = 0; it < STG; ++it) {
      if (it * nthr >= SP * 8) break;
      if (key0 + it * kround < SP) *reinterpret_cast<X8*>(vdst + it * kround * AVLD) = vreg[it];
    }
    __syncthreads();
  };

  // per-lane part of the transposed-read address: row (half * 4 + q), column 16 * dgrp + 4 p, and the 64-B
  // half swap of rows with bit 1 set (q >= 2)
  const int g16 = lane >> 4, i16 = lane & 15;
  const int vsw = ((i16 >> 3) & 1) << 5;
  const T* vtr = Vs + ((g16 >> 1) * 4 + (i16 >> 2)) * AVLD + (g16 & 1) * 16 + (i16 & 3) * 4;
  const int kswz = col & 7;                                // K chunk swizzle of this lane's key row
  const int colp = kperm(col);                             // its row inside a 32-key tile
  // One pass over the key tiles (round 5; the pass is described at pv() below): scores are in the log2 domain (q carries
  // 1/sqrt(64) * log2 e from the QKV epilogue), the padding keys of the last tile get -1e30 through the accumulator input.
  auto kfrag = [&](int kt, int ks) {
    return *reinterpret_cast<const X8*>(Ks + (kt * 32 + colp) * AVLD + (((2 * ks + half) ^ kswz) * 8));
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
  HVLA_ASTAMP();                                           // 2 (rounds 1-4: the first pass over the keys)
  if constexpr (!STREAM) {
    stage_v();
    HVLA_ASTAMP();                                         // 3 V staged
  }
  typedef float f32x2v __attribute__((ext_vector_type(2)));
  f32x2v lsum2 = {0.f, 0.f};        // this half's partial denominator (two interleaved partial sums)
  f32x16 O[2];
#pragma unroll
  for (int r = 0; r < 16; ++r) O[0][r] = 0.f, O[1][r] = 0.f;
  // ONE pass over the keys (round 5; rounds 1-4 ran K Q^T twice: row maxima first, then exp2 against the final maximum).  The
  // maximum runs along, and O / the denominator are rescaled LAZILY: only when some query of the wave meets a tile whose maximum
  // lies more than 8 (a factor 256 -- p stays below 2^8, far inside the 16-bit formats' range) above the maximum it is using, which
  // after the first tile or two does not happen -- a wave-uniform branch that is not taken.  Against the second K Q^T that is
  // 36 MFMAs, 36 ds_read_b128 (a third of the kernel's LDS bytes) and ~90 VALU instructions less per wave and item; a tile's
  // maximum costs eight v_max3 and one exchange between the two halves of the query's lanes (v_permlane32_swap, by hand: see
  // the column sums below).  Deterministic and batch-invariant as before: the branch depends on the (image, head)'s data only.
  // a value of the two halves of a query's lanes combined, the same in both (v_permlane32_swap on the value and a copy of it
  // leaves [lower, lower] in one register and [upper, upper] in the other; a ds_bpermute round trip otherwise)
  auto half_exchange_max = [](float t) {
    float u = t;
    asm("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(t), "+v"(u));
    return fmaxf(t, u);
  };
  auto half_exchange_sum = [](float t) {
    float u = t;
    asm("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(t), "+v"(u));
    return t + u;
  };
  auto tile_max = [&](const f32x16& sc) {
    float t = sc[0];
#pragma unroll
    for (int r = 1; r < 16; ++r) t = fmaxf(t, sc[r]);
    float u = t;
    asm("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(t), "+v"(u));     // t = [lower half's, lower half's], u = [upper, upper]
    // rounded up to an integer: maxima then differ by whole numbers, every rescale factor (and the factor between the p of two
    // rescale policies) is an exact power of two, and o does not depend on when which maximum was in use
    return __builtin_ceilf(fmaxf(t, u));
  };
  auto pv = [&](int kt, const f32x16& sc) {
    X8 pf[2];
    {
      const float tm = tile_max(sc);
      if (kt == 0) {
        mx = tm;
      } else if (__builtin_amdgcn_ballot_w64(tm > mx + 8.f) != 0) {
        const float mn = fmaxf(mx, tm);
        const float al = __builtin_amdgcn_exp2f(mx - mn);
#pragma unroll
        for (int r = 0; r < 16; ++r) O[0][r] *= al, O[1][r] *= al;
        lsum2[0] *= al, lsum2[1] *= al;
        mx = mn;
      }
      if (AMAP && wave == 0 && lane == 0) clsm[kt] = mx;   // the maximum this tile's row of the attention map is relative to
    }
#pragma unroll
    for (int r = 0; r < 16; r += 2) {
      const f32x2v p2 = {__builtin_amdgcn_exp2f(sc[r] - mx), __builtin_amdgcn_exp2f(sc[r + 1] - mx)};
      if (AMAP && wave == 0 && col == 0) {                 // query 0 = the CLS token: its unnormalised row waits in LDS for the denominator
        clsrow[kt * 32 + crow(r, half)] = p2[0];
        clsrow[kt * 32 + crow(r + 1, half)] = p2[1];
      }
      // two plain adds, kept apart by the asm statements: as ONE v_pk_add_f32 (what `lsum2 += p2` compiles to, and what the
      // compiler makes of two adjacent adds by itself) the kernel is 3.5 % SLOWER with 72 instructions fewer -- a packed f32
      // instruction costs this issue-bound kernel about three plain ones (same box: attention 1.572 -> 1.517 ms per step)
      { float la = lsum2[0], lb = lsum2[1];
        la += p2[0]; asm volatile("" : "+v"(la));
        lb += p2[1]; asm volatile("" : "+v"(lb));
        lsum2[0] = la, lsum2[1] = lb; }
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
  // (The scores relative to the maximum in use -- -max as the accumulator input of a tile's first MFMA, one tuple of 16 registers
  // rewritten only when the maximum moves, p = exp2(score) with no subtraction: 16 instructions less per key tile -- was built: 1 %.
  // After the one-pass form this kernel is no longer bound by its VALU instruction count.)
  if constexpr (STREAM) {
    using std::integral_constant;
    stream_wait(integral_constant<int, 6>{});              // chunk 0 (keys 0 .. 63) is in LDS for every wave
    HVLA_ASTAMP();                                         // 3 (STREAM) the first hand-over: what the head of the workgroup waits for
    pv(0, qk(0, zero16()));
    pv(1, qk(1, zero16()));
    stream_wait(integral_constant<int, 4>{});
    pv(2, qk(2, zero16()));
    pv(3, qk(3, zero16()));
    stream_wait(integral_constant<int, 2>{});
    pv(4, qk(4, zero16()));
    pv(5, qk(5, zero16()));
    stream_wait(integral_constant<int, 0>{});
    pv(6, qk(6, zero16()));
    pv(7, qk(7, zero16()));
    if (tid < 8) *reinterpret_cast<X8*>(qxs + tid * 8) = qlast;      // (the compiler waits for it with vmcnt(0): everything has landed)
    stream_wait(integral_constant<int, -1>{});             // key 256, the zero rows, the last query
  } else if constexpr (NWC != 0) {
#pragma unroll
    for (int kt = 0; kt < NWC; ++kt) pv(kt, qk(kt, zero16()));
  } else {
    for (int kt = 0; kt < KT - 1; ++kt) pv(kt, qk(kt, zero16()));
  }
  pv(KT - 1, qk(KT - 1, mask16()));
  const float inv = 1.f / half_exchange_sum(lsum2[0] + lsum2[1]);
  HVLA_ASTAMP();                                           // 4 the pass over the keys done
  if (AMAP && wave == 0) {       // outputs.attentions[layer][b, head, 0, 1:] (base_vit.py:117-118, hypervla_interface.py:210-211)
    const float i0 = lane_bcast(inv, 0);
    float* am = amap + (size_t)b * amap_stride + (size_t)head * (S - 1);
    const float m0 = lane_bcast(mx, 0);
    for (int j = lane; j < S - 1; j += 64) am[j] = clsrow[1 + j] * __builtin_amdgcn_exp2f(clsm[(1 + j) >> 5] - m0) * i0;
  }
  {
    T* op = o + ((size_t)b * S + q) * E + head * 64;
    // The lane's 32 outputs (item j = 16 mt + 4 g4 + r: column d = 32 mt + 8 g4 + 4 half + r) are to be summed over the 32 lanes
    // (queries) of its half.  Round 5: a reduce-scatter over the lane bits instead of 32 full 32-lane reductions (each 4 DPP
    // adds + a ds_bpermute + a masked LDS store: a fifth of an item's clock ticks in a kernel that is bound by VALU issue).
    // Lane bit 4 (the two 16-lane rows of a half): v_permlane16_swap exchanges the odd rows of the mt = 0 value with the even rows
    // of the mt = 1 value, one add leaves item j in the even row and item j + 16 in the odd one; bits 3 .. 0: the lanes whose
    // bit is clear keep the lower half of the remaining items, the others the upper half, and add the partner's share (row
    // mirror, half-row mirror and the two quad permutations: each pairs lanes that differ in that bit).  Lane l ends with the
    // sum of item l & 31: 77 instructions and one LDS store.  (The order of the additions differs from rounds 1-4; these mean
    // rows only feed the weight-rounding compensation, to per cents.)
    // (The instruction is issued by hand: hipcc 7.2 returns the FIRST result of __builtin_amdgcn_permlane16_swap in both elements
    // of its pair -- tools/permlane_swap_probe.hip.  `s_nop 1`: two wait states between a VALU write and the swap that reads it,
    // which the hazard recognizer cannot place inside an asm statement.)
    float w16[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) O[0][j] *= inv, O[1][j] *= inv;
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        typename Op::x4 v4;
#pragma unroll
        for (int r = 0; r < 4; ++r) v4[r] = (T)O[mt][g4 * 4 + r];
        *reinterpret_cast<typename Op::x4*>(op + mt * 32 + g4 * 8 + half * 4) = v4;
      }
    if (omean) {
#pragma unroll
      for (int j = 0; j < 16; j += 4) {
        float a0 = O[0][j], a1 = O[0][j + 1], a2 = O[0][j + 2], a3 = O[0][j + 3];
        float b0 = O[1][j], b1 = O[1][j + 1], b2 = O[1][j + 2], b3 = O[1][j + 3];
        asm("s_nop 1\n\tv_permlane16_swap_b32 %0, %4\n\tv_permlane16_swap_b32 %1, %5\n\tv_permlane16_swap_b32 %2, %6\n\tv_permlane16_swap_b32 %3, %7"
            : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3));
        w16[j] = a0 + b0, w16[j + 1] = a1 + b1, w16[j + 2] = a2 + b2, w16[j + 3] = a3 + b3;
      }
    }
    if (omean) {
      const bool b3 = lane & 8, b2 = lane & 4, b1 = lane & 2, b0 = lane & 1;
      float w8[8], w4[4], w2[2];
#pragma unroll
      for (int j = 0; j < 8; ++j) w8[j] = (b3 ? w16[j + 8] : w16[j]) + dpp_mov<0x140>(b3 ? w16[j] : w16[j + 8]);     // row_mirror
#pragma unroll
      for (int j = 0; j < 4; ++j) w4[j] = (b2 ? w8[j + 4] : w8[j]) + dpp_mov<0x141>(b2 ? w8[j] : w8[j + 4]);         // row_half_mirror
#pragma unroll
      for (int j = 0; j < 2; ++j) w2[j] = (b1 ? w4[j + 2] : w4[j]) + dpp_mov<0x4E>(b1 ? w4[j] : w4[j + 2]);          // lanes ^ 2
      const float w1 = (b0 ? w2[1] : w2[0]) + dpp_mov<0xB1>(b0 ? w2[0] : w2[1]);                                     // lanes ^ 1
      const int j = lane & 31;                                                                                       // = 16 mt + 4 g4 + r
      csum[wave * 64 + (j >> 4) * 32 + ((j >> 2) & 3) * 8 + half * 4 + (j & 3)] = w1;
    }
  }
  HVLA_ASTAMP();                                           // 5 normalised, column sums, stores issued
  // ---- the last query, through the matrix pipe (round 5).  Its scores against key tile `wave` are one 32 x 32 tile of S^T = K Q^T
  // whose 32 columns are all THIS query (the B fragment is the parked query, the same in every lane column), so every lane holds
  // the 16 keys of its half in registers: max / sum are in-lane + one exchange between the halves, P feeds the P.V tile from
  // registers exactly as in the pass over the keys, and the partial (max, sum, O[64]) goes to LDS from one lane column.  On the VALU (rounds 1-4:
  // 32 fma per key with converts, a 32-step readlane loop for P.V) this was 9 400 of an item's 34 000 clock ticks in a kernel whose
  // two workgroups per CU contend for VALU issue (profiles/r5_experiments_not_kept.txt 5).  The last wave also takes the final key
  // tile (one real key, the rest masked).
  {
    X8 ql[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) ql[ks] = *reinterpret_cast<const X8*>(qxs + ks * 16 + half * 8);
    auto lastq = [&](int tile, bool masked) {
      f32x16 c = zero16();
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) c = Op::mma32(kfrag(tile, ks), ql[ks], c);
      if (masked) {
#pragma unroll
        for (int r = 0; r < 16; ++r) c[r] = (tile * 32 + crow(r, half) < S) ? c[r] : -1e30f;
      }
      float m = c[0];
#pragma unroll
      for (int r = 1; r < 16; ++r) m = fmaxf(m, c[r]);
      m = half_exchange_max(m);
      X8 pf[2];
      float la = 0.f, lb = 0.f;
#pragma unroll
      for (int r = 0; r < 16; r += 2) {
        const float p0 = __builtin_amdgcn_exp2f(c[r] - m), p1 = __builtin_amdgcn_exp2f(c[r + 1] - m);
        la += p0; asm volatile("" : "+v"(la));       // (plain adds: see pv())
        lb += p1; asm volatile("" : "+v"(lb));
        pf[r >> 3][r & 7] = (T)p0;
        pf[r >> 3][(r & 7) + 1] = (T)p1;
      }
      float l = la + lb;
      l = half_exchange_sum(l);
      f32x16 ol[2];
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) {
        ol[mt] = zero16();
#pragma unroll
        for (int sstep = 0; sstep < 2; ++sstep) {
          const T* v0 = vtr + ((tile * 32 + sstep * 16) * AVLD) + ((mt * 32) ^ vsw);
          ol[mt] = Op::mma32(tr_read2<X8>(v0, v0 + 8 * AVLD), pf[sstep], ol[mt]);
        }
      }
      float* pp = part + tile * APS;
      if (col == 0) {                      // one lane column: lane (0, half) holds d = 32 mt + crow(r, half) = 32 mt + 8 (r >> 2) + 4 half + (r & 3)
        if (half == 0) pp[0] = m, pp[1] = l;
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
          for (int g4 = 0; g4 < 4; ++g4)
            *reinterpret_cast<f32x4*>(pp + 4 + 32 * mt + 8 * g4 + 4 * half) = f32x4{ol[mt][4 * g4], ol[mt][4 * g4 + 1], ol[mt][4 * g4 + 2], ol[mt][4 * g4 + 3]};
      }
    };
    lastq(wave, false);
    // The final key tile holds ONE real key (token S - 1): p = 1, sum = 1, P.V = its V row, the score one 64-term dot product
    // (lane = d) -- a dozen VALU instructions on the first wave instead of a second masked 32-key tile on the last one, which
    // every other wave then waited for at the barrier below.
    if (wave == 0) {
      const int key = S - 1;
      const float sc = wave64_sum((float)qxs[lane] * (float)Ks[kperm(key) * AVLD + (((lane >> 3) ^ (key & 7)) * 8) + (lane & 7)]);
      float* pp = part + (KT - 1) * APS;
      if (lane == 0) pp[0] = sc, pp[1] = 1.f;
      pp[4 + lane] = (float)Vs[key * AVLD + (lane ^ (((key >> 1) & 1) << 5))];
    }
  }
  __syncthreads();
  HVLA_ASTAMP();                                           // 6 last-query partials done
  if (wave == 0) {
    // (every value is requested before the first is used: KT and NW are run-time numbers, and a rolled loop pays an LDS round
    // trip per partial while the workgroup's other waves have already left)
    constexpr int KTM = 9;
    float last;
    if (KT <= KTM) {
      float pm[KTM], pl[KTM], po[KTM];
#pragma unroll
      for (int t = 0; t < KTM; ++t) {
        const bool in = t < KT;
        pm[t] = in ? part[t * APS] : -1e30f;
        pl[t] = in ? part[t * APS + 1] : 0.f;
        po[t] = in ? part[t * APS + 4 + lane] : 0.f;
      }
      float M = -1e30f;
#pragma unroll
      for (int t = 0; t < KTM; ++t) M = fmaxf(M, pm[t]);
      float L = 0.f, od = 0.f;
#pragma unroll
      for (int t = 0; t < KTM; ++t) {
        const float f = __builtin_amdgcn_exp2f(pm[t] - M);
        L = fmaf(pl[t], f, L);
        od = fmaf(po[t], f, od);
      }
      last = od / L;
    } else {
      float M = -1e30f;
      for (int t = 0; t < KT; ++t) M = fmaxf(M, part[t * APS]);
      float L = 0.f, od = 0.f;
      for (int t = 0; t < KT; ++t) {
        const float f = __builtin_amdgcn_exp2f(part[t * APS] - M);
        L = fmaf(part[t * APS + 1], f, L);
        od = fmaf(part[t * APS + 4 + lane], f, od);
      }
      last = od / L;
    }
    o[((size_t)b * S + (S - 1)) * E + head * 64 + lane] = (T)last;
    if (omean) {               // [image][half][E]: tokens [0, 32 NW / 2) and the rest (the wave split; half a wave's width off the
      const int NH = NW / 2;   // consumer's token split for odd NW, one token off at S = 257: the mean only needs per cents)
      float t0 = 0.f, t1 = last;
      if (NW == 8) {
        float c[8];
#pragma unroll
        for (int w = 0; w < 8; ++w) c[w] = csum[w * 64 + lane];
#pragma unroll
        for (int w = 0; w < 4; ++w) t0 += c[w], t1 += c[4 + w];
      } else {
        for (int w = 0; w < NH; ++w) t0 += csum[w * 64 + lane];
        for (int w = NH; w < NW; ++w) t1 += csum[w * 64 + lane];
      }
      if (NH == 0) t0 = t1;                                  // one wave (tiny geometries): both rows carry the whole image's mean
      omean[((size_t)b * 2 + 0) * E + head * 64 + lane] = (T)(t0 / (float)(NH ? NH * 32 : S));
      omean[((size_t)b * 2 + 1) * E + head * 64 + lane] = (T)(t1 / (float)(NH ? S - NH * 32 : S));
    }
  }
  HVLA_ASTAMP();                                           // 7 end
}

// ------------------------------------------------------------------------------------------------
// Weight-rounding compensation.  For a GEMM  Y = A W  run as  A W16  (W16 = W rounded to 16 bits),
// the missing term A (W - W16) is approximated per image by  abar_b (W - W16),  abar_b = the mean over the image's patch
// rows of A: a [B, N] table added by the epilogue in place of the bias.
//
// colmean_kernel: mean row of a 16-bit activation matrix over the P patch rows of every image (the CLS row is left out:
// the table only needs the mean to a few per cent), summed as two halves so that gemm256p_kernel's GELU epilogue can produce the
// same numbers for its own outputs (same values, same order of additions => same bits; the batch-invariance tests cross
// the two): half wm = rows [wm P/2, +P/2) of the image; inside a half, quad q (rows 4q..4q+3) goes to partial q & 3, each
// partial adds its values in ascending row order IN THE 16-BIT TYPE, the half is (p0 + p1) + (p2 + p3) in f32, and each
// half's mean (half / (P / 2), rounded to the operand type) is one row of the table [image][half][K].
// Only small batches come here (a large batch gets the sums from the GELU epilogue), so the launch is one latency chain and is
// laid out for that: the eight independent partial sums of a column (2 halves x 4 partials) go to eight 32-lane groups, a lane
// takes eight adjacent columns with 16-byte loads and has 16 rows in flight (23 -> 4 us at B = 1: twelve workgroups walking
// 256 rows with four 2-byte loads in flight each).
template <typename T>
__global__ __launch_bounds__(256) void colmean_kernel(const T* __restrict__ a, T* __restrict__ abar, int S, int P, int K) {
  typedef T T8 __attribute__((ext_vector_type(8)));
  __shared__ float part[8][256];
  const int b = blockIdx.y, chain = threadIdx.x >> 5, l32 = threadIdx.x & 31;
  const int wm = chain >> 2, pq = chain & 3;
  const int n0 = blockIdx.x * 256 + l32 * 8;
  const int half = P / 2, nq = half / 4;
  T8 p;
#pragma unroll
  for (int i = 0; i < 8; ++i) p[i] = (T)0.f;                 // accumulated in the operand type, as the GELU epilogue does
  if (n0 < K) {
    const T* base = a + ((size_t)b * S + 1 + wm * half) * K + n0;
    for (int q = pq; q < nq; q += 16) {                       // quads q, q + 4, q + 8, q + 12 of this partial: ascending rows
      T8 v[4][4];
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (q + 4 * u < nq) {
#pragma unroll
          for (int r = 0; r < 4; ++r) v[u][r] = *reinterpret_cast<const T8*>(base + (size_t)(4 * (q + 4 * u) + r) * K);
        }
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (q + 4 * u < nq) {
#pragma unroll
          for (int r = 0; r < 4; ++r) p = p + v[u][r];
        }
    }
  }
#pragma unroll
  for (int i = 0; i < 8; ++i) part[chain][l32 * 8 + i] = (float)p[i];
  __syncthreads();
  const int t = threadIdx.x, n = blockIdx.x * 256 + t;
  if (n >= K) return;
  const float h0 = (part[0][t] + part[1][t]) + (part[2][t] + part[3][t]);
  const float h1 = (part[4][t] + part[5][t]) + (part[6][t] + part[7][t]);
  abar[((size_t)b * 2 + 0) * K + n] = (T)(h0 * (1.f / (float)half));       // [image][half][K], as the GELU epilogue writes it
  abar[((size_t)b * 2 + 1) * K + n] = (T)(h1 * (1.f / (float)half));
}

// ------------------------------------------------------------------------------------------------
// Range audit of the 16-bit operands (hvla_encode_audit; tests only): largest |value| and number of non-finite values
// of a buffer, accumulated into slot[0] (float bits, non-negative floats order like unsigned integers) and slot[1].
template <typename T>
__global__ __launch_bounds__(256) void absmax_kernel(const T* __restrict__ p, size_t n8, uint32_t* __restrict__ slot) {
  typedef T x8 __attribute__((ext_vector_type(8)));
  float mx = 0.f;
  uint32_t bad = 0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (size_t)gridDim.x * blockDim.x) {
    const x8 v = reinterpret_cast<const x8*>(p)[i];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float f = fabsf((float)v[j]);
      if (!(f <= 3.0e38f)) ++bad;                    // inf or NaN
      else mx = fmaxf(mx, f);
    }
  }
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) {
    mx = fmaxf(mx, __shfl_xor(mx, o, 64));
    bad += __shfl_xor(bad, o, 64);
  }
  if ((threadIdx.x & 63) == 0) {
    atomicMax(slot, __float_as_uint(mx));
    if (bad) atomicAdd(slot + 1, bad);
  }
}

// ------------------------------------------------------------------------------------------------
namespace {
struct DeviceInfo {       // per device: a second HyperVLA on another GPU of the same process needs its own
  bool attr[2] = {false, false};
  int ncu = 0;
};
DeviceInfo g_dev[64];
}  // namespace

// every launch of an encoder call is counted (Profiler::nlaunch; hvla_launches): bench.py reports launches per step from here
#define HVLA_LAUNCH(...) do { ++pf.nlaunch; hipLaunchKernelGGL(__VA_ARGS__); } while (0)
template <typename Op>
static hipError_t run_encoder(const Geom& g, const EncWeights& w, const EncWorkspace& ws, const uint8_t* images,
                              float* tokens, int B, hipStream_t st, Profiler* prof, bool keep_cls, uint32_t* audit) {
  Profiler none;
  Profiler& pf = prof ? *prof : none;
  using T = typename Op::elem;
  constexpr int opi = std::is_same<Op, OpBF16>::value ? 1 : 0;
  const int P = g.P(), S = g.S(), E = g.E, F = g.enc_mlp, H = g.enc_heads;
  const int Kp = 2 * ((g.patch * g.patch * 3 + 63) / 64 * 64);   // [a | a] x [W_hi | W_lo]
  const int M = B * S;
  // the GEMM epilogues address their outputs with 32-bit element offsets
  if ((size_t)M * (size_t)(F > 3 * E ? F : 3 * E) >= (1ull << 32)) return hipErrorInvalidValue;
  const size_t gsm = (size_t)2 * (GBM + GBN) * GLD * sizeof(T);
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return e;
  if (dev < 0 || dev >= 64) return hipErrorInvalidDevice;
  DeviceInfo& di = g_dev[dev];
  if (!di.attr[opi]) {
#define SETA(K) \
  if ((e = hipFuncSetAttribute(reinterpret_cast<const void*>(K), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)) != hipSuccess) return e;
    SETA((gemm_kernel<Op, EPI_PATCH>)) SETA((gemm_kernel<Op, EPI_QKV>)) SETA((gemm_kernel<Op, EPI_GELU>))
    SETA((gemm_kernel<Op, EPI_RES>)) SETA((attention_kernel<Op, false>)) SETA((attention_kernel<Op, true>)) SETA((attention_kernel<Op, false, 8>))
    SETA((gemm256p_kernel<Op, EPI_PATCH, false>)) SETA((gemm256p_kernel<Op, EPI_QKV, false>))
    SETA((gemm256p_kernel<Op, EPI_GELU, false>)) SETA((gemm256p_kernel<Op, EPI_RES, false>))
    SETA((gemm256p_kernel<Op, EPI_PATCH, true>)) SETA((gemm256p_kernel<Op, EPI_QKV, true>))
    SETA((gemm256p_kernel<Op, EPI_GELU, true>)) SETA((gemm256p_kernel<Op, EPI_RES, true>))
    SETA((gemm256p_kernel<Op, EPI_PATCH, false, true>)) SETA((gemm256p_kernel<Op, EPI_RES, false, true>))
    SETA((gemm256p_kernel<Op, EPI_PATCH, true, true>)) SETA((gemm256p_kernel<Op, EPI_RES, true, true>))
    SETA((gemm256p_kernel<Op, EPI_RES, false, true, true>)) SETA((gemm256p_kernel<Op, EPI_RES, true, true, true>))
    SETA((gemm64_kernel<Op, EPI_PATCH>)) SETA((gemm64_kernel<Op, EPI_QKV>)) SETA((gemm64_kernel<Op, EPI_GELU>))
    SETA((gemm64_kernel<Op, EPI_RES>)) SETA((gemm64_kernel<Op, EPI_CORR>))
    SETA((gemm64_kernel<Op, EPI_QKV, 4>)) SETA((gemm64_kernel<Op, EPI_GELU, 4>)) SETA((gemm64_kernel<Op, EPI_RES, 4>))
    SETA((gemm64c_kernel<Op, EPI_QKV>)) SETA((gemm64c_kernel<Op, EPI_GELU>)) SETA((gemm64c_kernel<Op, EPI_RES>)) SETA((gemm64c32_kernel<Op>))
    SETA((gemm64c_kernel<Op, EPI_QKV, true>)) SETA((gemm64c_kernel<Op, EPI_GELU, true>)) SETA((layernorm_group_kernel<Op>))
    SETA((gemm256p_kernel<Op, EPI_QKV, false, false, true>)) SETA((gemm256p_kernel<Op, EPI_GELU, false, false, true>))
    SETA((gemm256p_kernel<Op, EPI_QKV, true, false, true>)) SETA((gemm256p_kernel<Op, EPI_GELU, true, false, true>))
    SETA((gemm256p_kernel<Op, EPI_RES, false, false, true>)) SETA((gemm256p_kernel<Op, EPI_RES, true, false, true>))
#undef SETA
    di.attr[opi] = true;
  }
  if (!di.ncu && (hipDeviceGetAttribute(&di.ncu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || di.ncu <= 0)) di.ncu = 256;
  const int ncu = di.ncu;
  constexpr int G64_MAXM = 2047;       // rows up to which a GEMM is cut into 64x64 tiles (pure latency below that)
  constexpr int CAT_COMP = 8;          // HVLA_PROF_COMP
  const bool comp = w.layer[0].dqkv != nullptr && ws.corr && ws.abar;
  const int hsplit = 1 + P / 2;        // tokens [1, hsplit) | [hsplit, S): the two wave rows of an image-aligned tile; the CLS row takes the plain bias
  // ---- one GEMM of the encoder: activations [B*S rows][K] -> [B*S rows][N], with the per-image bias rows that compensate
  // the rounding of W (dW = the rounding residue x 4096; the mean row of the activation operand is in ws.abar, written by
  // the kernel that produced the operand)
  //  * images of 256 patches, batch >= 8, N % 256 == 0: ONE gemm64_kernel launch for the 2 B latency-bound rows (the B CLS
  //    rows, stride S, plain bias; the B mean rows against dW -> ws.corr), then gemm256p_kernel over image-aligned tiles
  //    (tile row b = rows b*S + 1 .. b*S + 256 = the patch rows of image b) with ws.corr as its bias rows;
  //  * otherwise a corr-only gemm64_kernel launch, then gemm64_kernel (<= 2047 rows) or gemm_kernel (128x128 tiles) over all rows, bias row
  //    looked up per row.
  // Returns whether the image-aligned form ran (then a GELU epilogue writes the mean row of its output itself).
  // ln_s / ln_b (RES only): scale and bias of the LayerNorm that follows this GEMM; the image-aligned form runs it inside its
  // epilogue (GemmArgs::ln_*, gemm256p_kernel<..., LNX>) and sets ln_fused, otherwise the caller launches it.
  bool ln_fused = false;
  uint32_t ln_launch = 0;              // LNX launches of this call so far: the count an image's word reaches is launch number x nbn
  // small batch: does the GEMM [M][K] x [N][K] run as gemm64c_kernel (bias rows computed inside)?  ...and may it also add up the
  // mean rows from the LayerNorm's partials (FOLD)?  ln_partial: set by layernorm() when it left the mean rows to its consumer.
  auto small_fused = [&](int N, int K) {
    return comp && M <= G64_MAXM && (size_t)M * K < (1ull << 31) && (size_t)N * K < (1ull << 31) && N % SBN == 0 && K % 64 == 0 && S >= 9;
  };
  const float* ln_partial = nullptr;
  const bool can_fuse_ln = ws.ln_cnt != nullptr && ws.ln_part != nullptr && (size_t)M * E * 4 < (1ull << 32) && E <= 1024;
  // the persistent form of an LNX launch keeps the nbn column tiles of an image in one round (tile_origin_x): the counts must divide
  auto lnx_persistent = [&](int nbm_, int nbn_) {
    const int nt = nbm_ * nbn_;
    if (ncu % 8 || nt % ncu || nbm_ % 8) return false;
    const int Wx = ncu / 8, odd = Wx % nbn_, R = nt / ncu;
    return Wx >= nbn_ && (odd == 0 || R % nbn_ == 0);
  };
  auto lnx_args = [&](GemmArgs& a, const float* ln_s, const float* ln_b, int nbn_) {
    a.ln_out = ws.h; a.ln_scale = ln_s; a.ln_bias = ln_b; a.ln_abar = comp ? ws.abar : nullptr; a.ln_cnt = ws.ln_cnt; a.ln_part = ws.ln_part;
    a.out_bytes = (uint32_t)((size_t)M * E * 4);
    a.part_bytes = (uint32_t)((size_t)B * nbn_ * 256 * 16);
    a.ln_tag = ++ln_launch;
    a.ln_target = ln_launch * (uint32_t)nbn_;
    a.ln_spin = ws.ln_spin;
    if (lnx_persistent((int)a.nbm, nbn_)) {
      const int Wx = ncu / 8, R = a.nbm * nbn_ / ncu;
      a.lnx_G = Wx / nbn_; a.lnx_rows_aligned = R * a.lnx_G; a.lnx_rounds_div = R / nbn_; a.lnx_rows_xcd = a.nbm / 8;
      a.lnx_rcp = (65536 + nbn_ - 1) / nbn_;
    }
  };
  auto gemm = [&](auto epic, const void* A, const void* Wt, const void* dW, int N, int K, const float* bias, const float* aux,
                  void* out, int qcols, int cat, void* colmean = nullptr, const float* ln_s = nullptr, const float* ln_b = nullptr) -> bool {
    constexpr int EPI = decltype(epic)::value;
    ln_fused = false;
    GemmArgs a{A, Wt, M, N, K, bias, aux, out, P, S, qcols, qcols ? 0.125f * 1.4426950408889634f : 1.f / 256.f};   // q: 1/sqrt(64) and exp -> exp2
    a.hsplit = hsplit;
    const bool fits32 = (size_t)M * K < (1ull << 31) && (size_t)N * K < (1ull << 31);
    // N % 256 != 0 (DINOv2-small: 384, 1152): the last tile column is 64, 128 or 192 wide, the rest of its MFMAs wasted (33 % at
    // N = 384) -- still twice as fast as the 128 x 128 register-staged kernel
    const bool aligned = P == HBM_ && N % SBN == 0 && N >= HBN_ && M > G64_MAXM && K >= 128 && K % 64 == 0 && fits32;
    if (aligned) {
      pf.begin(CAT_COMP, st);
      GemmArgs c = a;                                  // the B CLS rows (+ the B mean rows -> ws.corr)
      c.M = B; c.row0 = 0; c.row_step = S;
      int nblocks = ((B + SBM - 1) / SBM) * (N / SBN);
      if (comp) { c.abar2 = ws.abar; c.dW2 = dW; c.corr2 = ws.corr; c.M2 = 2 * B; nblocks += ((2 * B + SBM - 1) / SBM) * (N / SBN); }   // two mean rows per image
      if constexpr (EPI != EPI_PATCH) {
        // (a three-stage form, three workgroups per CU for the launches above 512 blocks, was tried: run-to-run different
        // tokens in ~4 % of the episodes of a 1024-episode batch, whichever order the reads and the DMA issue are in, while
        // the four- and six-stage forms are clean -- not understood, not used: profiles/r3_experiments_not_kept.txt)
        if (nblocks > ncu) HVLA_LAUNCH((gemm64_kernel<Op, EPI, 4>), dim3(nblocks), dim3(256), 4 * 16384, st, c);   // two workgroups per CU
        else HVLA_LAUNCH((gemm64_kernel<Op, EPI>), dim3(nblocks), dim3(256), SNS * 16384, st, c);
      } else {
        HVLA_LAUNCH((gemm64_kernel<Op, EPI>), dim3(nblocks), dim3(256), SNS * 16384, st, c);
      }
      pf.end(CAT_COMP, st);
      const int nbn = (N + HBN_ - 1) / HBN_;
      a.corr = comp ? ws.corr : nullptr;
      a.nbm = B; a.tile_row0 = 1; a.tile_stride = S; a.colmean = colmean;
      const size_t lds = 131072;
      pf.begin(cat, st);
      if constexpr (EPI == EPI_RES) {
        if (ln_s && can_fuse_ln) {                     // the LayerNorm behind this GEMM inside its epilogue
          lnx_args(a, ln_s, ln_b, nbn);
          const bool nt = big_output(M, N, sizeof(float));      // a big batch: the residual rows are read past L2 (below)
          if (lnx_persistent(B, nbn)) {
            if (nt) HVLA_LAUNCH((gemm256p_kernel<Op, EPI, true, true, true>), dim3(ncu), dim3(512), lds, st, a);
            else HVLA_LAUNCH((gemm256p_kernel<Op, EPI, true, true>), dim3(ncu), dim3(512), lds, st, a);
          } else {
            if (nt) HVLA_LAUNCH((gemm256p_kernel<Op, EPI, false, true, true>), dim3(B * nbn), dim3(512), lds, st, a);
            else HVLA_LAUNCH((gemm256p_kernel<Op, EPI, false, true>), dim3(B * nbn), dim3(512), lds, st, a);
          }
          pf.end(cat, st);
          ln_fused = true;
          return true;
        }
      }
      if constexpr (EPI != EPI_PATCH) {
        // a big batch: the 16-bit output goes past L2, the f32 residual rows are read past it (see the epilogue)
        // (the 16-bit NT stores carry 32-bit BYTE offsets: an output of 4 GiB or more takes the plain form -- ADVICE r5)
        if (big_output(M, N, EPI == EPI_RES ? sizeof(float) : sizeof(T)) && (EPI == EPI_RES || nt16_addressable(M, N, sizeof(T)))) {
          if ((B * nbn) % ncu == 0) HVLA_LAUNCH((gemm256p_kernel<Op, EPI, true, false, true>), dim3(ncu), dim3(512), lds, st, a);
          else HVLA_LAUNCH((gemm256p_kernel<Op, EPI, false, false, true>), dim3(B * nbn), dim3(512), lds, st, a);
          pf.end(cat, st);
          return true;
        }
      }
      if ((B * nbn) % ncu == 0) HVLA_LAUNCH((gemm256p_kernel<Op, EPI, true>), dim3(ncu), dim3(512), lds, st, a);
      else HVLA_LAUNCH((gemm256p_kernel<Op, EPI, false>), dim3(B * nbn), dim3(512), lds, st, a);
      pf.end(cat, st);
      return true;
    }
    if constexpr (EPI != EPI_PATCH) {
      // (S >= 9: a block's 64 rows then belong to at most 8 images = the 16 rows of the kernels' bias-row tile)
      if (small_fused(N, K)) {                           // small batch: the bias rows are computed inside the GEMM
        a.abar2 = ws.abar; a.dW2 = dW; a.M2 = 2 * B;
        pf.begin(cat, st);                               // (no launch of its own for the bias rows: nothing is timed as HVLA_PROF_COMP)
        const int nb64 = ((M + SBM - 1) / SBM) * (N / SBN);
        if constexpr (EPI != EPI_RES) {
          if (ln_partial) {                              // the LayerNorm in front left the mean rows to this launch
            a.ln_partial = ln_partial;
            ln_partial = nullptr;
            HVLA_LAUNCH((gemm64c_kernel<Op, EPI, true>), dim3(nb64), dim3(256), gemm64c_fold_lds(K), st, a);
            pf.end(cat, st);
            return false;
          }
        }
        if constexpr (EPI == EPI_RES) {
          if (2 * nb64 <= ncu) {                     // less than half of the chip: 64 x 32 tiles on twice as many CUs
            HVLA_LAUNCH((gemm64c32_kernel<Op>), dim3(2 * nb64), dim3(256), SNS32 * SST32, st, a);
            pf.end(cat, st);
            return false;
          }
        }
        HVLA_LAUNCH((gemm64c_kernel<Op, EPI>), dim3(nb64), dim3(256), SNSC * SSTC, st, a);
        pf.end(cat, st);
        return false;
      }
    }
    if (comp) {                                        // the same gemm64_body<EPI_CORR> arithmetic as the fused launch above
      pf.begin(CAT_COMP, st);
      GemmArgs c{ws.abar, dW, 2 * B, N, K, bias, nullptr, ws.corr, P, S, 0, 1.f};
      HVLA_LAUNCH((gemm64_kernel<Op, EPI_CORR>), dim3(((2 * B + SBM - 1) / SBM) * (N / SBN)), dim3(256), SNS * 16384, st, c);
      a.corr = ws.corr;
      pf.end(CAT_COMP, st);
    }
    pf.begin(cat, st);
    if (M <= G64_MAXM && fits32 && N % SBN == 0 && K % 64 == 0)
      HVLA_LAUNCH((gemm64_kernel<Op, EPI>), dim3(((M + SBM - 1) / SBM) * (N / SBN)), dim3(256), SNS * 16384, st, a);
    else
      HVLA_LAUNCH((gemm_kernel<Op, EPI>), dim3(((M + GBM - 1) / GBM) * (N / GBN)), dim3(256), gsm, st, a);
    pf.end(cat, st);
    return false;
  };
  // Nnext: width of the GEMM that reads the output.  scratch / scratch_cols: a 16-bit [M][scratch_cols] workspace buffer that is free
  // from this launch until its consumer GEMM has ENDED (norm1: ws.g, QKV writes ws.qkv; norm2: ws.qkv, fc1 writes ws.g), for the
  // partial column sums [B][LNG][E] f32
  auto layernorm = [&](const float* sc, const float* bi, int Nnext, void* scratch, int scratch_cols) {   // norm1 / norm2 (+ the column sums of the output)
    const bool room = (size_t)LNG * E * sizeof(float) <= (size_t)S * scratch_cols * sizeof(T);
    float* partial = comp && room ? reinterpret_cast<float*>(scratch) : nullptr;
    HVLA_LAUNCH((layernorm_group_kernel<Op>), dim3(LNG, B), dim3(LNW * 64), (size_t)(P / 8) * E * sizeof(float), st, ws.x,
                       reinterpret_cast<T*>(ws.h), sc, bi, partial, S, P, E);
    if (!partial) return;
    // (one launch with the image's last workgroup to arrive -- ticket by atomicAdd behind a __threadfence -- adding the
    // partials was tried: 14.0 us against 5.2 + 4.7 for the two launches, profiles/r3_experiments_not_kept.txt)
    // B = 1: every workgroup of the consumer GEMM adds the partials up itself (gemm64c_kernel, FOLD) -- while that GEMM is ONE round
    // of workgroups (round 4, same box: B = 1 1.250 against 1.270 ms per step with the separate launch; B = 4 1.93 against 1.85).
    if (small_fused(Nnext, E) && gemm64c_fold_lds(E) <= 160 * 1024 && ((M + SBM - 1) / SBM) * (Nnext / SBN) <= ncu) {
      ln_partial = partial;
      return;
    }
    HVLA_LAUNCH((layernorm_mean_kernel<Op>), dim3(2 * B), dim3(256), 0, st, partial, reinterpret_cast<T*>(ws.abar), P, E);
  };
  auto colmean_of = [&](const void* act, int K) {      // mean row of a GEMM output whose epilogue did not write it (no image-aligned tiles)
    if (!comp) return;
    HVLA_LAUNCH((colmean_kernel<T>), dim3((K + 255) / 256, B), dim3(256), 0, st, reinterpret_cast<const T*>(act),
                       reinterpret_cast<T*>(ws.abar), S, P, K);
  };
  auto audit_of = [&](const void* buf, size_t n, int site) {      // site: 0 LayerNorm out, 1 q/k/v, 2 attention out, 3 GELU out
    if (!audit) return;
    HVLA_LAUNCH((absmax_kernel<T>), dim3(1024), dim3(256), 0, st, reinterpret_cast<const T*>(buf), n / 8, audit + 2 * site);
  };
  // the images' arrival words of the fused LayerNorms: zero before the first launch of every call (a memset node when the call is captured)
  if (can_fuse_ln && P == HBM_ && M > G64_MAXM) {
    if ((e = hipMemsetAsync(ws.ln_cnt, 0, (size_t)((B * 4 + 15) / 16 * 16), st)) != hipSuccess) return e;
    if ((e = hipMemsetAsync(ws.ln_part, 0, (size_t)B * ((E + HBN_ - 1) / HBN_) * 256 * 16, st)) != hipSuccess) return e;   // the entries' tags
    pf.nlaunch += 2;
  }
  using EQ = std::integral_constant<int, EPI_QKV>;
  using EG = std::integral_constant<int, EPI_GELU>;
  using ER = std::integral_constant<int, EPI_RES>;
  // patch embedding
  pf.begin(0, st);
  {
    const size_t total = (size_t)B * P * (Kp / 16);
    int blocks = (int)((total + 255) / 256);
    if (blocks > 8192) blocks = 8192;
    HVLA_LAUNCH(im2col_kernel<Op>, dim3(blocks), dim3(256), 0, st, images, reinterpret_cast<T*>(ws.g), B,
                       g.image_size, g.patch, g.grid(), Kp, ws.x, w.pos, S, E);
    const int Mp = B * P;
    GemmArgs a{ws.g, w.w_patch, Mp, E, Kp, w.b_patch, w.pos, ws.x, P, S, 0, 1.f / 256.f};
    const bool fits32 = (size_t)Mp * Kp < (1ull << 31);
    if (Mp % HBM_ == 0 && E % SBN == 0 && E >= HBN_ && Mp > G64_MAXM && fits32) {
      a.nbm = Mp / HBM_; a.tile_row0 = 0; a.tile_stride = HBM_;
      const int nbn = (E + HBN_ - 1) / HBN_;
      if (P == HBM_ && can_fuse_ln && g.enc_layers > 0) {   // a tile row is an image: norm1 of layer 0 inside the epilogue (the CLS rows are written above)
        a.hsplit = hsplit;
        lnx_args(a, w.layer[0].ln1_s, w.layer[0].ln1_b, nbn);
        if (lnx_persistent(a.nbm, nbn)) HVLA_LAUNCH((gemm256p_kernel<Op, EPI_PATCH, true, true>), dim3(ncu), dim3(512), 131072, st, a);
        else HVLA_LAUNCH((gemm256p_kernel<Op, EPI_PATCH, false, true>), dim3(a.nbm * nbn), dim3(512), 131072, st, a);
        ln_fused = true;
      } else if ((a.nbm * nbn) % ncu == 0) HVLA_LAUNCH((gemm256p_kernel<Op, EPI_PATCH, true>), dim3(ncu), dim3(512), 131072, st, a);
      else HVLA_LAUNCH((gemm256p_kernel<Op, EPI_PATCH, false>), dim3(a.nbm * nbn), dim3(512), 131072, st, a);
    } else if (Mp <= G64_MAXM && fits32 && E % SBN == 0) {
      HVLA_LAUNCH((gemm64_kernel<Op, EPI_PATCH>), dim3(((Mp + SBM - 1) / SBM) * (E / SBN)), dim3(256), SNS * 16384, st, a);
    } else {
      HVLA_LAUNCH((gemm_kernel<Op, EPI_PATCH>), dim3(((Mp + GBM - 1) / GBM) * (E / GBN)), dim3(256), gsm, st, a);
    }
  }
  pf.end(0, st);
  const int KT = (S + 31) / 32;     // S = 32 * (KT - 1) + 1
  const size_t asm_bytes = (size_t)KT * 32 * 2 * AVLD * sizeof(T) + (size_t)KT * APS * sizeof(float) + 64 * sizeof(T) +
                           (size_t)(KT - 1) * 64 * sizeof(float);
#ifdef HVLA_BENCH_HOOKS
  int products = 0;
#define HVLA_STOP_HERE() do { if (ws.stop_after > 0 && ++products == ws.stop_after) return hipGetLastError(); } while (0)
#else
#define HVLA_STOP_HERE() do { } while (0)
#endif
  for (int l = 0; l < g.enc_layers; ++l) {
    const EncLayerW& L = w.layer[l];
    if (!ln_fused) {                    // norm1: otherwise done as the tail of the previous layer's fc2 / of the patch embedding
      pf.begin(1, st);
      layernorm(L.ln1_s, L.ln1_b, 3 * E, ws.g, F);
      pf.end(1, st);
    }
    audit_of(ws.h, (size_t)M * E, 0);
    gemm(EQ{}, ws.h, L.wqkv, L.dqkv, 3 * E, E, L.bqkv, nullptr, ws.qkv, E, 2);                       // the LayerNorm wrote the mean row itself
    audit_of(ws.qkv, (size_t)M * 3 * E, 1);
    HVLA_STOP_HERE();
    pf.begin(3, st);
    const bool att_unrolled = KT == 9;
    auto attn = [&](auto kern, size_t lds) {
      HVLA_LAUNCH(kern, dim3(B * H), dim3((KT - 1) * 64), lds, st, reinterpret_cast<const T*>(ws.qkv), reinterpret_cast<T*>(ws.h), S, E, H,
                  comp ? reinterpret_cast<T*>(ws.abar) : nullptr, ws.amap ? ws.amap + (size_t)l * H * (S - 1) : nullptr, g.enc_layers * H * (S - 1)
#ifdef HVLA_BENCH_HOOKS
                  , (unsigned long long*)nullptr, 0     // (default arguments do not travel through a function pointer)
#endif
                  );
    };
    if (ws.amap) {                     // (unrolled, the attention-map instantiation spills: it stays rolled)
      attn(attention_kernel<Op, true>, asm_bytes + (size_t)(KT * 32 + 32) * sizeof(float));
    } else {
      if (att_unrolled) attn(attention_kernel<Op, false, 8>, asm_bytes);
      else attn(attention_kernel<Op, false>, asm_bytes);
    }
    pf.end(3, st);
    audit_of(ws.h, (size_t)M * E, 2);
    gemm(ER{}, ws.h, L.wo, L.dwo, E, E, L.bo, L.ls1, ws.x, 0, 4, nullptr, L.ln2_s, L.ln2_b);   // the attention kernel wrote the mean row itself; norm2 as the tail
    if (!ln_fused) {
      pf.begin(1, st);
      layernorm(L.ln2_s, L.ln2_b, F, ws.qkv, 3 * E);
      pf.end(1, st);
    }
    HVLA_STOP_HERE();
    audit_of(ws.h, (size_t)M * E, 0);
    const bool summed = gemm(EG{}, ws.h, L.w1, L.dw1, F, E, L.b1, nullptr, ws.g, 0, 5, comp ? ws.abar : nullptr);
    audit_of(ws.g, (size_t)M * F, 3);
    if (!summed) colmean_of(ws.g, F);                                                 // aligned tiles: the GELU epilogue wrote the mean row
    HVLA_STOP_HERE();
    if (l + 1 < g.enc_layers) gemm(ER{}, ws.g, L.w2, L.dw2, E, F, L.b2, L.ls2, ws.x, 0, 6, nullptr, w.layer[l + 1].ln1_s, w.layer[l + 1].ln1_b);   // the next layer's norm1 as the tail
    else gemm(ER{}, ws.g, L.w2, L.dw2, E, F, L.b2, L.ls2, ws.x, 0, 6);
    HVLA_STOP_HERE();
  }
#undef HVLA_STOP_HERE
  pf.begin(1, st);
  if (keep_cls)
    HVLA_LAUNCH((layernorm_kernel<Op, 2>), dim3((M + 3) / 4), dim3(256), 0, st, ws.x, tokens, w.lnf_s, w.lnf_b, M, E, S);
  else
    HVLA_LAUNCH((layernorm_kernel<Op, 1>), dim3((M + 3) / 4), dim3(256), 0, st, ws.x, tokens, w.lnf_s, w.lnf_b, M, E, S);
  pf.end(1, st);
  return hipGetLastError();
}

#undef HVLA_LAUNCH
#ifdef HVLA_BENCH_HOOKS
hipError_t debug_lnx_stats(unsigned long long* out, int reset) {   // out: host [8][256]
  hipError_t e = hipDeviceSynchronize();
  if (e == hipSuccess && out) e = hipMemcpyFromSymbol(out, HIP_SYMBOL(g_lnx_dbg), sizeof(unsigned long long) * 8 * 256);
  if (e == hipSuccess && reset) {
    static unsigned long long zero[8][256];
    e = hipMemcpyToSymbol(HIP_SYMBOL(g_lnx_dbg), zero, sizeof zero);
  }
  return e;
}
// diagnostics, libhvla_bench.so only (tools/gemm_bench.py): time `iters` launches of one encoder GEMM shape on workspace buffers (contents
// irrelevant).  variant 0: gemm_kernel (128x128), 1: gemm64_kernel, 2: gemm256p_kernel one workgroup per tile,
// 3: gemm256p_kernel persistent.  M is rounded down to whole 256-row tiles for variants 2 and 3.
hipError_t debug_gemm(const void* A, const void* W, const float* bias, const float* aux, void* out, int M, int N,
                      int K, int epi, int variant, int iters, float* ms, hipStream_t st) {
  using Op = OpF16;
  GemmArgs a{A, W, M, N, K, bias, aux, out, 256, 257, epi == EPI_QKV ? N / 3 : 0, 0.125f};
  a.nbm = M / HBM_; a.tile_row0 = 0; a.tile_stride = HBM_;
  if (epi < EPI_QKV || epi > EPI_RES || variant < 0 || variant > 3 || N % HBN_ || K % 64 || (variant >= 2 && a.nbm < 1)) return hipErrorInvalidValue;
  int dev = 0, ncu = 256;
  (void)hipGetDevice(&dev);
  (void)hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev);
  const size_t gsm = (size_t)2 * (GBM + GBN) * GLD * 2;
  auto launch_e = [&](auto epic) {
    constexpr int EPI = decltype(epic)::value;
    static bool attr = false;
    if (!attr) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_kernel<Op, EPI>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm64_kernel<Op, EPI>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm256p_kernel<Op, EPI, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm256p_kernel<Op, EPI, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      attr = true;
    }
    const int nt = a.nbm * (N / HBN_);
    if (variant == 0) hipLaunchKernelGGL((gemm_kernel<Op, EPI>), dim3(((M + GBM - 1) / GBM) * (N / GBN)), dim3(256), gsm, st, a);
    else if (variant == 1) hipLaunchKernelGGL((gemm64_kernel<Op, EPI>), dim3(((M + SBM - 1) / SBM) * (N / SBN)), dim3(256), SNS * 16384, st, a);
    else if (variant == 2) hipLaunchKernelGGL((gemm256p_kernel<Op, EPI, false>), dim3(nt), dim3(512), 131072, st, a);
    else hipLaunchKernelGGL((gemm256p_kernel<Op, EPI, true>), dim3(nt < ncu ? nt : ncu), dim3(512), 131072, st, a);
  };
  auto launch = [&]() {
    if (epi == EPI_QKV) launch_e(std::integral_constant<int, EPI_QKV>{});
    else if (epi == EPI_GELU) launch_e(std::integral_constant<int, EPI_GELU>{});
    else launch_e(std::integral_constant<int, EPI_RES>{});
  };
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
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
// phase stamps of attention_kernel (tools/attention_timeline.py): one launch over B * H workgroups on workspace-shaped buffers
hipError_t debug_attention_stamps(const void* qkv, void* o, void* omean, int B, int S, int E, int H, int wg, unsigned long long* stamps,
                                  hipStream_t st) {
  using Op = OpF16;
  using T = Op::elem;
  const int KT = (S + 31) / 32;
  const size_t asm_bytes = (size_t)KT * 32 * 2 * AVLD * sizeof(T) + (size_t)KT * APS * sizeof(float) + 64 * sizeof(T) +
                           (size_t)(KT - 1) * 64 * sizeof(float);
  auto go = [&](auto kern) {           // the instantiation the step launches at this S (run_encoder)
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipLaunchKernelGGL(kern, dim3(B * H), dim3((KT - 1) * 64), asm_bytes, st, reinterpret_cast<const T*>(qkv),
                       reinterpret_cast<T*>(o), S, E, H, reinterpret_cast<T*>(omean), (float*)nullptr, 0, stamps, wg);
  };
  if (KT == 9) go(attention_kernel<Op, false, 8>);
  else go(attention_kernel<Op, false>);
  return hipGetLastError();
}
#endif  // HVLA_BENCH_HOOKS

hipError_t launch_encoder(const Geom& g, int dtype, const EncWeights& w, const EncWorkspace& ws,
                          const uint8_t* images, float* tokens, int B, hipStream_t st, Profiler* prof, bool keep_cls,
                          uint32_t* audit) {
  if (dtype == 1) return run_encoder<OpBF16>(g, w, ws, images, tokens, B, st, prof, keep_cls, audit);
  return run_encoder<OpF16>(g, w, ws, images, tokens, B, st, prof, keep_cls, audit);
}

}  // namespace hvla
