// kernels.h — parameter blocks + launchers shared between the kernel files and api.hip.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <vector>

#include "layout.h"

namespace hvla {

// ---------------------------------------------------------------- hypernetwork
struct CtxLayer {
  const float *ln0_s, *ln0_b, *wq, *bq, *wk, *bk, *wv, *bv, *wo, *bo, *ln1_s, *ln1_b, *w1, *b1, *w2, *b2;
};
constexpr int CTX_MAX_LAYERS = 8;
struct CtxParams {
  int T, C, F, heads, layers, lang_dim, E, scale_context;
  int big_elems;             // floats in the kernel's third LDS buffer (set by launch_ctx_encoder)
  const float* tok;          // [B, T, lang_dim]
  const int64_t* attn_mask;  // [B, T]
  const float* cls;          // [B, E]
  const float *w_tok, *b_tok, *w_img, *b_img, *pos_tok, *pos_img, *pos_layer, *norm_s, *norm_b;
  CtxLayer layer[CTX_MAX_LAYERS];
  float* ctx;                // [B, C]
  __bf16 *ctx_hi, *ctx_lo;   // [B, C]
};
struct WeightGenParams {
  const __bf16 *wcat_hi, *wcat_lo;   // [ntiles][KS][64 lanes][8]
  const float* bcat;                 // [Gm + Gv]
  const __bf16 *ctx_hi, *ctx_lo;     // [B, C]
  __bf16 *wh, *wl;                   // [B, Gm]
  float* vf;                         // [B, Gv]
  int B, Gm, Gv, ntiles;
};
hipError_t launch_ctx_encoder(const CtxParams& p, int B, hipStream_t st);
size_t ctx_encoder_lds_bytes(int T, int C, int F, int E);       // dynamic LDS of ctx_encoder_kernel for a geometry (<= 160 KiB or refused at create)
hipError_t launch_weightgen(const WeightGenParams& p, int C, hipStream_t st);
hipError_t launch_export_theta(const __bf16* wh, const __bf16* wl, const float* vf, const int32_t* perm,
                               int Gm, int Gv, int G, int B, float* theta, hipStream_t st);

// ---------------------------------------------------------------- image encoder (DINOv2)
struct EncLayerW {
  const void *wqkv, *wo, *w1, *w2;               // 16-bit [N][K] (K contiguous)
  const void *dqkv, *dwo, *dw1, *dw2;            // 16-bit [N][K]: (W - the 16-bit W above) x 4096, the rounding the corr table compensates
  const float *bqkv, *bo, *b1, *b2;              // f32
  const float *ln1_s, *ln1_b, *ln2_s, *ln2_b, *ls1, *ls2;
};
constexpr int ENC_MAX_LAYERS = 24;
struct EncWeights {
  const void* w_patch;        // 16-bit [E][Kp] (normalisation folded in)
  const float* b_patch;       // [E] (bias - sum W mean/std)
  const float* pos;           // [S, E] position embeddings (row 0 already has cls_token added)
  const float *lnf_s, *lnf_b;
  EncLayerW layer[ENC_MAX_LAYERS];
};
struct EncWorkspace {
  float* x;        // [B*S, E] residual stream f32
  void* h;         // [B*S, E] 16-bit LN output / attention output
  void* qkv;       // [B*S, 3E] 16-bit
  void* g;         // [B*S, F] 16-bit MLP hidden; also holds the im2col matrix [B*P, Kp]
  float* corr;     // [B, 2, max(3E, F)] bias rows of the current GEMM per image half (bias + mean row of that half . dW)
  void* abar;      // [B, 2, max(E, F)] 16-bit mean rows (upper / lower half of the image) of the current GEMM's activation operand
  float* amap = nullptr;   // nullable: [B, enc_layers, enc_heads, P] attention of the CLS query over the patch keys (opt-in export)
  uint32_t* ln_cnt = nullptr;   // nullable: [B rounded up to 4] arrival words of the LayerNorms fused into the residual GEMMs (encoder.hip GemmArgs::ln_cnt)
  float* ln_part = nullptr;     // nullable: [B][<= 4][256] 16-byte entries, the per-row (sum, sum of squares) of every column tile (GemmArgs::ln_part); either null = LayerNorm as its own launch
  uint32_t ln_spin = 800;       // how long a column tile waits for the image's other tiles before it leaves its share to the last arriver: ticks of 10 ns
#ifdef HVLA_BENCH_HOOKS
  int stop_after = 0;           // libhvla_bench.so (hvla_debug_encode_stop): return behind the n-th dense product of the call (QKV, out, fc1, fc2 of
                                // layer 0 = 1 .. 4, ...), so that a test can read the workspace as that product's consumers would find it
#endif
};
// optional live timing: a pool of hipEvent pairs tagged with a category (include/hvla.h HVLA_PROF_*)
struct Profiler {
  int mode = 0;                       // 0 off, 1 the selected categories only, 2 all
  uint32_t select = 1u << 5;          // mode 1: bit c = category c (hvla_profile_select; default the fc1 GEMM)
  std::vector<hipEvent_t> start, stop;
  std::vector<int> cat;
  size_t used = 0;
  uint64_t nlaunch = 0;               // kernel launches since the last hvla_launches() (counted in run_encoder / hvla_policy / hvla_ensemble)
  bool want(int c) const { return c >= 0 && (mode == 2 || (mode == 1 && ((select >> c) & 1u))); }
  void begin(int c, hipStream_t st) {
    if (!want(c)) return;
    if (used == start.size()) {
      hipEvent_t a, b;
      if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) return;
      start.push_back(a), stop.push_back(b), cat.push_back(c);
    }
    cat[used] = c;
    (void)hipEventRecord(start[used], st);
  }
  void end(int c, hipStream_t st) {
    if (!want(c) || used >= start.size()) return;
    (void)hipEventRecord(stop[used], st);
    ++used;
  }
};
hipError_t launch_encoder(const Geom& g, int dtype, const EncWeights& w, const EncWorkspace& ws,
                          const uint8_t* images, float* tokens, int B, hipStream_t st, Profiler* prof,
                          bool keep_cls = false,    // keep_cls: tokens = f32 [B, S, E] last_hidden_state
                          uint32_t* audit = nullptr);   // [4 sites][2]: max |16-bit operand| (float bits), non-finite count

#ifdef HVLA_BENCH_HOOKS
hipError_t debug_gemm(const void* A, const void* W, const float* bias, const float* aux, void* out, int M, int N,
                      int K, int epi, int variant, int iters, float* ms, hipStream_t st);
hipError_t debug_lnx_stats(unsigned long long* out, int reset);   // [8][256]: fused-LayerNorm counters per workgroup id (encoder.hip g_lnx_dbg)
hipError_t debug_ctx_stamps(unsigned long long* out);   // [64] shader-clock stamps of the last context-encoder launch
hipError_t debug_attention_stamps(const void* qkv, void* o, void* omean, int B, int S, int E, int H, int wg, unsigned long long* stamps,
                                  hipStream_t st);
#endif

// ---------------------------------------------------------------- observation preprocessing (resize.hip)
void build_resize_spans(int in_size, int out_size, std::vector<int>& start, std::vector<int>& count, std::vector<float>& w,
                        int& maxspan);
hipError_t launch_resize(const uint8_t* src, uint8_t* dst, float* tmp_rows, float* tmp_img, const int* row_start,
                         const int* row_count, const float* row_w, int row_span, const int* col_start, const int* col_count,
                         const float* col_w, int col_span, int B, int H, int W, int S, int crop, hipStream_t st,
                         float* padded = nullptr, int src_h = 0, int src_w = 0);   // padded: f32 [B][H][W][3] scratch of resize_with_pad

// ---------------------------------------------------------------- generated policy
struct PolicyParams {
  PolicyLayout pl;
  const __bf16 *wh, *wl;     // [B, Gm]
  const float* vf;           // [B, Gv]
  const float* tokens;       // [B, P, E]
  float* actions;            // [B, horizon, action_dim]
  float* logits;             // [B, horizon] or null
  int B, E, P, L, M, horizon, action_dim;
  float tanh_scale, max_action;
  float* amap = nullptr;     // [B, L, heads, P] or null: attention of the action token over the patch keys, every layer
#ifdef HVLA_BENCH_HOOKS
  unsigned long long* stamps = nullptr;   // libhvla_bench.so: shader-clock time stamps of episode 0 / wave 0 at the phase boundaries
#endif
};
hipError_t launch_policy(const PolicyParams& p, hipStream_t st);

// ---------------------------------------------------------------- caller-side device helpers
hipError_t launch_ensemble(const float* actions, float* ring, int* count, const float* mean,
                           const float* std, const uint8_t* mask, float* out, int B, int horizon,
                           int action_dim, hipStream_t st);

hipError_t launch_loss(const float* actions, const float* logits, const float* target, const uint8_t* tmask,
                       const uint8_t* amask, float* loss, int B, int horizon, int action_dim, float max_action,
                       bool clip_target, hipStream_t st);

// ---------------------------------------------------------------- self test
hipError_t launch_selftest(int* fail_flags, hipStream_t st);
hipError_t run_box_probe(float* sink, unsigned long long* ticks, float out[3], hipStream_t st);   // selftest.hip: sustained clock / MFMA rate of this box

}  // namespace hvla
