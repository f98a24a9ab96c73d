// api.hip — the C ABI of libhvla (include/hvla.h): context, weight packing, launch sequencing.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/hvla.h"
#include "common.h"
#include "kernels.h"
#include "layout.h"
#include "pack.h"
#include "t5.h"
#include "train.h"

using namespace hvla;

namespace {

struct DevBuf {
  void* p = nullptr;
  size_t bytes = 0;
  ~DevBuf() { if (p) (void)hipFree(p); }
  hipError_t alloc(size_t n) {
    if (p) { (void)hipFree(p); p = nullptr; }
    bytes = n;
    return n ? hipMalloc(&p, n) : hipSuccess;
  }
  template <typename T> T* as() const { return reinterpret_cast<T*>(p); }
};

using hvla::pack::bf2f;
using hvla::pack::f2bf;
using hvla::pack::f2h;

}  // namespace

struct hvla_weights {
  int B = 0;
  DevBuf wh, wl, vf, ctx, ring, count;
};

struct hvla_ctx {
  hvla_config cfg{};
  Geom g{};
  int device = 0;
  std::string err;
  bool loaded = false;
  bool train_timer = false;      // this context switched its device's GEMM timer on (hvla_train_profile): hvla_destroy gives the events back
  PackedLayout lay;
  int Kp = 0;
  // device weights
  DevBuf hn_f32;                 // context-encoder parameters, natural flax layout
  CtxParams ctxp{};
  DevBuf wcat_hi, wcat_lo, bcat, perm;
  DevBuf enc16, encd16, encf32;  // encoder matrices (16-bit), their rounding residues x 4096 (16-bit) and vectors (f32)
  EncWeights encw{};
  // workspaces (sized for cfg.max_batch)
  DevBuf ctx_hi, ctx_lo, ctx_f32, ws_x, ws_h, ws_qkv, ws_g, ws_corr, ws_abar, ws_lncnt, ws_lnpart, tokens, flags;
  uint32_t ln_spin = 800;        // EncWorkspace::ln_spin (8 us); hvla_debug_lnx_spin of the bench library changes it
#ifdef HVLA_BENCH_HOOKS
  int enc_stop = 0;              // EncWorkspace::stop_after (hvla_debug_encode_stop)
#endif
  Profiler prof;
  float *amap_dino = nullptr, *amap_head = nullptr;     // hvla_set_attention_outputs: caller-owned device buffers (opt-in)
  // cfg.streams == 2: helper stream and fork / join events of hvla_step
  hipStream_t side = nullptr;
  hipEvent_t ev_fork = nullptr, ev_join = nullptr;
  // weight arenas handed back with hvla_weights_free wait here for the next hvla_generate of the same batch size: after
  // the first episode batch hvla_generate does not allocate (include/hvla.h: it can then be captured / does not sync)
  std::vector<hvla_weights*> arena_pool;
  std::mutex pool_mu;                        // hvla_weights_free can arrive from another thread (Python's GC) than hvla_generate
  static constexpr size_t ARENA_POOL_MAX = 4;
  hipEvent_t ev_bucket[3] = {nullptr, nullptr, nullptr};   // hvla_train_step: gradient buckets final (created on first use)
  bool bucket_recorded[3] = {false, false, false};
  ~hvla_ctx() {
    for (hvla_weights* w : arena_pool) delete w;
    for (hipEvent_t e : ev_bucket)
      if (e) (void)hipEventDestroy(e);
    if (ev_fork) (void)hipEventDestroy(ev_fork);
    if (ev_join) (void)hipEventDestroy(ev_join);
    if (side) (void)hipStreamDestroy(side);
  }
  // observation preprocessing (hvla_preprocess): span tables of the last (H, W) and scratch
  int rs_H = 0, rs_W = 0, rs_row_span = 0, rs_col_span = 0;
  DevBuf rs_tab, rs_rows, rs_img, rs_pad;
  // optional frozen T5 instruction encoder (hvla_t5_load)
  bool t5_loaded = false;
  T5Dims t5d{};
  T5Weights t5w{};
  DevBuf t5_f32, t5_work;
  int t5_max_batch = 0;
};

#define FAIL(ctx, code, ...)                       \
  do {                                             \
    char _b[512];                                  \
    snprintf(_b, sizeof _b, __VA_ARGS__);          \
    (ctx)->err = _b;                               \
    return (code);                                 \
  } while (0)
#define HIPCHK(ctx, call)                                                               \
  do {                                                                                  \
    hipError_t _e = (call);                                                             \
    if (_e != hipSuccess) FAIL(ctx, HVLA_E_HIP, "%s: %s", #call, hipGetErrorString(_e)); \
  } while (0)

extern "C" {

const char* hvla_last_error(const hvla_ctx* ctx) { return ctx ? ctx->err.c_str() : "null ctx"; }

int hvla_create(const hvla_config* c, int device, hvla_ctx** out) {
  if (!c || !out) return HVLA_E_SHAPE;
  *out = nullptr;
  if (c->struct_size != sizeof(hvla_config)) return HVLA_E_SHAPE;   // the caller's header is not this library's: never read past its struct
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) return HVLA_E_DEVICE;
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, device) != hipSuccess) return HVLA_E_DEVICE;
  if (!strstr(prop.gcnArchName, "gfx950")) return HVLA_E_DEVICE;   // CDNA4 only: no fallback path
  std::unique_ptr<hvla_ctx> ctx(new hvla_ctx);
  ctx->cfg = *c;
  ctx->device = device;
  Geom& g = ctx->g;
  g = Geom{c->image_size, c->patch, c->enc_dim, c->enc_layers, c->enc_heads, c->enc_mlp,
           c->dim, c->layers, c->heads, c->mlp, c->horizon, c->action_dim, c->tanh_scale, c->max_action,
           c->ctx_dim, c->ctx_layers, c->ctx_heads, c->ctx_mlp, c->lang_tokens, c->lang_dim, c->scale_context,
           c->clip_target != 0};
  // what the hand-written kernels are specialised for (anything else is refused, never emulated)
  const int P = g.P();
  const bool ok = c->dim == 64 && c->heads == 4 && c->mlp % 32 == 0 && c->mlp >= 32 && c->enc_dim % 128 == 0 &&
                  c->enc_dim <= 1024 && c->enc_mlp % 128 == 0 && c->enc_dim / c->enc_heads == 64 &&
                  (P == 256 || P == 64 || P == 32) && c->image_size % c->patch == 0 &&
                  (c->ctx_dim == 128 || c->ctx_dim == 64 || c->ctx_dim == 32) && c->ctx_dim % c->ctx_heads == 0 &&
                  c->ctx_mlp % 16 == 0 && c->lang_tokens + 2 <= 40 && c->lang_tokens >= 2 && c->lang_dim % 4 == 0 &&
                  c->ctx_layers <= CTX_MAX_LAYERS && c->enc_layers <= ENC_MAX_LAYERS && c->layers >= 1 &&
                  c->horizon * (c->action_dim - 1) + c->horizon <= 32 && c->max_batch >= 1 &&
                  (c->enc_dtype == HVLA_ENC_F16 || c->enc_dtype == HVLA_ENC_BF16);
  if (!ok) return c->enc_dtype != HVLA_ENC_F16 && c->enc_dtype != HVLA_ENC_BF16 ? HVLA_E_DTYPE : HVLA_E_SHAPE;
  // the context encoder keeps its token block, q / k / v and the MLP hidden rows in LDS: a geometry that does not fit is
  // refused here, not at the first hvla_generate
  if (ctx_encoder_lds_bytes(c->lang_tokens, c->ctx_dim, c->ctx_mlp, c->enc_dim) > 160 * 1024) return HVLA_E_SHAPE;
  if (hipSetDevice(device) != hipSuccess) return HVLA_E_DEVICE;
  ctx->lay = build_layout(g);
  ctx->Kp = 2 * ((g.patch * g.patch * 3 + 63) / 64 * 64);     // [W_hi | W_lo] along K (encoder.hip)
  {  // the packed order must be a bijection onto the reference parameter vector
    std::vector<uint8_t> seen(ctx->lay.pl.G, 0);
    for (int32_t r : ctx->lay.perm)
      if (r >= 0) {
        if (r >= ctx->lay.pl.G || seen[r]) return HVLA_E_STATE;
        seen[r] = 1;
      }
    for (uint8_t s : seen)
      if (!s) return HVLA_E_STATE;
  }
  const size_t Bm = c->max_batch, S = g.S(), E = g.E, F = g.enc_mlp;
  size_t gbytes = Bm * S * F * 2;
  if (gbytes < Bm * P * ctx->Kp * 2) gbytes = Bm * P * ctx->Kp * 2;
  hipError_t e = hipSuccess;
  auto A = [&](DevBuf& b, size_t n) { if (e == hipSuccess) e = b.alloc(n); };
  A(ctx->ctx_hi, Bm * g.C * 2); A(ctx->ctx_lo, Bm * g.C * 2); A(ctx->ctx_f32, Bm * g.C * 4);
  A(ctx->ws_x, Bm * S * E * 4); A(ctx->ws_h, Bm * S * E * 2); A(ctx->ws_qkv, Bm * S * 3 * E * 2);
  A(ctx->ws_g, gbytes); A(ctx->tokens, Bm * P * E * 4); A(ctx->flags, 64 * sizeof(int));
  A(ctx->ws_corr, 2 * Bm * (F > 3 * E ? F : 3 * E) * 4);      // two rows per image (upper / lower half): encoder.hip GemmArgs::corr
  A(ctx->ws_abar, 2 * Bm * (F > E ? F : E) * 2);
  A(ctx->ws_lncnt, (Bm + 4) * 4 + 64);                          // arrival words of the fused LayerNorms, one per image (16-byte multiples per half batch)
  A(ctx->ws_lnpart, Bm * 4 * 256 * 16);                         // their per-row partial statistics: [image][column tile <= 4][256] entries of 16 bytes
  if (e != hipSuccess) return HVLA_E_ARENA_FULL;
  if (hipMemset(ctx->ws_lncnt.p, 0, ctx->ws_lncnt.bytes) != hipSuccess) return HVLA_E_HIP;
  if (c->streams == 2) {
    if (hipStreamCreateWithFlags(&ctx->side, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreateWithFlags(&ctx->ev_fork, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&ctx->ev_join, hipEventDisableTiming) != hipSuccess)
      return HVLA_E_HIP;
  } else if (c->streams != 0 && c->streams != 1) {
    return HVLA_E_SHAPE;
  }
  *out = ctx.release();
  return HVLA_OK;
}

void hvla_destroy(hvla_ctx* ctx) {
  if (!ctx) return;
  (void)hipSetDevice(ctx->device);
  if (ctx->train_timer) train_gemm_timer_release();
  delete ctx;
}

int64_t hvla_num_generated(const hvla_ctx* ctx) { return ctx ? ctx->lay.pl.G : 0; }

int hvla_load_weights(hvla_ctx* ctx, const hvla_tensor_desc* t, int32_t n) {
  if (!ctx) return HVLA_E_STATE;
  HIPCHK(ctx, hipSetDevice(ctx->device));
  const Geom& g = ctx->g;
  std::map<std::string, const hvla_tensor_desc*> m;
  for (int i = 0; i < n; ++i) {
    if (!t[i].name || !t[i].data) FAIL(ctx, HVLA_E_WEIGHTS, "tensor %d has null name/data", i);
    m[t[i].name] = &t[i];
  }
  const char* missing = nullptr;
  std::string missing_s;
  auto get = [&](const std::string& name, int64_t numel) -> const float* {
    auto it = m.find(name);
    if (it == m.end() || it->second->numel != numel) {
      if (!missing) {
        missing_s = name + (it == m.end() ? " (absent)" : " (wrong size)");
        missing = missing_s.c_str();
      }
      return nullptr;
    }
    return it->second->data;
  };
  const int C = g.C, Hc = g.ctx_heads, F = g.ctx_mlp, T = g.T, E = g.E;
  // ------------------------------------------------------------ context encoder (f32, natural layout)
  std::vector<float> hn;
  std::vector<std::pair<const float**, size_t>> fix;     // pointer slot -> offset
  auto push = [&](const float** slot, const std::string& name, int64_t numel) {
    const float* src = get(name, numel);
    const size_t off = hn.size();
    hn.resize(off + ((numel + 3) / 4) * 4, 0.f);
    if (src) memcpy(hn.data() + off, src, numel * 4);
    fix.push_back({slot, off});
  };
  CtxParams& cp = ctx->ctxp;
  cp = CtxParams{};
  cp.T = T; cp.C = C; cp.F = F; cp.heads = Hc; cp.layers = g.ctx_layers; cp.lang_dim = g.lang_dim; cp.E = E;
  cp.scale_context = g.scale_context;
  push(&cp.w_tok, "task_token_projection/kernel", (int64_t)g.lang_dim * C);
  push(&cp.b_tok, "task_token_projection/bias", C);
  push(&cp.w_img, "initial_image_projection/kernel", (int64_t)E * C);
  push(&cp.b_img, "initial_image_projection/bias", C);
  push(&cp.pos_tok, "task_pos_embedding", (int64_t)T * C);
  push(&cp.pos_img, "initial_image_pos_embedding", C);
  push(&cp.pos_layer, "layer_pos_embedding", C);
  push(&cp.norm_s, "Transformer_0/encoder_norm/scale", C);
  push(&cp.norm_b, "Transformer_0/encoder_norm/bias", C);
  for (int l = 0; l < g.ctx_layers; ++l) {
    const std::string b = "Transformer_0/encoderblock_" + std::to_string(l) + "/";
    const std::string a = b + "MultiHeadDotProductAttention_0/";
    CtxLayer& L = cp.layer[l];
    push(&L.ln0_s, b + "LayerNorm_0/scale", C); push(&L.ln0_b, b + "LayerNorm_0/bias", C);
    push(&L.wq, a + "query/kernel", (int64_t)C * C); push(&L.bq, a + "query/bias", C);
    push(&L.wk, a + "key/kernel", (int64_t)C * C); push(&L.bk, a + "key/bias", C);
    push(&L.wv, a + "value/kernel", (int64_t)C * C); push(&L.bv, a + "value/bias", C);
    push(&L.wo, a + "out/kernel", (int64_t)C * C); push(&L.bo, a + "out/bias", C);
    push(&L.ln1_s, b + "LayerNorm_1/scale", C); push(&L.ln1_b, b + "LayerNorm_1/bias", C);
    push(&L.w1, b + "MlpBlock_0/Dense_0/kernel", (int64_t)C * F); push(&L.b1, b + "MlpBlock_0/Dense_0/bias", F);
    push(&L.w2, b + "MlpBlock_0/Dense_1/kernel", (int64_t)F * C); push(&L.b2, b + "MlpBlock_0/Dense_1/bias", C);
  }
  // ------------------------------------------------------------ W_cat / b_cat in packed order
  const PolicyLayout& pl = ctx->lay.pl;
  const int Gtot = pl.Gm + pl.Gv;
  auto leaves = generated_leaves(g);
  std::vector<const float*> lk(leaves.size()), lb(leaves.size());
  for (size_t i = 0; i < leaves.size(); ++i) {
    lk[i] = get("output_head_" + leaves[i].flat + "/kernel", (int64_t)C * leaves[i].size);
    lb[i] = get("output_head_" + leaves[i].flat + "/bias", leaves[i].size);
  }
  // ------------------------------------------------------------ DINOv2 (shared leaves, flat vectors)
  const std::string ep = "encoder_image_encoder_";
  const int Fe = g.enc_mlp, p = g.patch, S = g.S(), Kp = ctx->Kp, Kreal = p * p * 3;
  const float* e_cls = get(ep + "embeddings_cls_token", E);
  (void)get(ep + "embeddings_mask_token", E);
  const float* e_pk = get(ep + "embeddings_patch_embeddings_projection_kernel", (int64_t)Kreal * E);
  const float* e_pb = get(ep + "embeddings_patch_embeddings_projection_bias", E);
  const float* e_pos = get(ep + "embeddings_position_embeddings", (int64_t)S * E);
  const float* e_lns = get(ep + "layernorm_scale", E);
  const float* e_lnb = get(ep + "layernorm_bias", E);
  struct LSrc { const float *q, *qb, *k, *kb, *v, *vb, *o, *ob, *l1, *l2, *f1, *f1b, *f2, *f2b, *n1s, *n1b, *n2s, *n2b; };
  std::vector<LSrc> ls(g.enc_layers);
  for (int i = 0; i < g.enc_layers; ++i) {
    const std::string L = ep + "encoder_layer_" + std::to_string(i) + "_";
    LSrc& s = ls[i];
    s.q = get(L + "attention_attention_query_kernel", (int64_t)E * E); s.qb = get(L + "attention_attention_query_bias", E);
    s.k = get(L + "attention_attention_key_kernel", (int64_t)E * E); s.kb = get(L + "attention_attention_key_bias", E);
    s.v = get(L + "attention_attention_value_kernel", (int64_t)E * E); s.vb = get(L + "attention_attention_value_bias", E);
    s.o = get(L + "attention_output_dense_kernel", (int64_t)E * E); s.ob = get(L + "attention_output_dense_bias", E);
    s.l1 = get(L + "layer_scale1_lambda1", E); s.l2 = get(L + "layer_scale2_lambda1", E);
    s.f1 = get(L + "mlp_fc1_kernel", (int64_t)E * Fe); s.f1b = get(L + "mlp_fc1_bias", Fe);
    s.f2 = get(L + "mlp_fc2_kernel", (int64_t)Fe * E); s.f2b = get(L + "mlp_fc2_bias", E);
    s.n1s = get(L + "norm1_scale", E); s.n1b = get(L + "norm1_bias", E);
    s.n2s = get(L + "norm2_scale", E); s.n2b = get(L + "norm2_bias", E);
  }
  if (missing) FAIL(ctx, HVLA_E_WEIGHTS, "checkpoint tensor %s", missing);

  // ---- upload the context encoder
  HIPCHK(ctx, ctx->hn_f32.alloc(hn.size() * 4));
  HIPCHK(ctx, hipMemcpy(ctx->hn_f32.p, hn.data(), hn.size() * 4, hipMemcpyHostToDevice));
  for (auto& f : fix) *f.first = ctx->hn_f32.as<float>() + f.second;

  // ---- pack W_cat^T fragments (layout.h): tile pt, k-step ks, lane (rho = l & 31, hk = l >> 5), j
  {
    std::vector<uint16_t> hi, lo;
    std::vector<float> bc;
    for (size_t i = 0; i < leaves.size(); ++i)
      if (!lk[i] || !lb[i]) FAIL(ctx, HVLA_E_WEIGHTS, "output head %s missing", leaves[i].flat.c_str());
    pack::pack_wcat(ctx->lay, leaves, lk, lb, C, hi, lo, bc);
    HIPCHK(ctx, ctx->wcat_hi.alloc(hi.size() * 2));
    HIPCHK(ctx, ctx->wcat_lo.alloc(lo.size() * 2));
    HIPCHK(ctx, ctx->bcat.alloc(bc.size() * 4));
    HIPCHK(ctx, ctx->perm.alloc(ctx->lay.perm.size() * 4));
    HIPCHK(ctx, hipMemcpy(ctx->wcat_hi.p, hi.data(), hi.size() * 2, hipMemcpyHostToDevice));
    HIPCHK(ctx, hipMemcpy(ctx->wcat_lo.p, lo.data(), lo.size() * 2, hipMemcpyHostToDevice));
    HIPCHK(ctx, hipMemcpy(ctx->bcat.p, bc.data(), bc.size() * 4, hipMemcpyHostToDevice));
    HIPCHK(ctx, hipMemcpy(ctx->perm.p, ctx->lay.perm.data(), ctx->lay.perm.size() * 4, hipMemcpyHostToDevice));
  }

  // ---- pack the image encoder: 16-bit [N][K] matrices, f32 vectors
  {
    const bool bf = ctx->cfg.enc_dtype == HVLA_ENC_BF16;
    auto cv = [&](float f) { return bf ? f2bf(f) : f2h(f); };
    const size_t per_layer16 = (size_t)3 * E * E + (size_t)E * E + (size_t)2 * E * Fe;
    std::vector<uint16_t> w16((size_t)E * Kp + per_layer16 * g.enc_layers), d16(w16.size(), 0);
    const size_t per_layerf = (size_t)3 * E + E + Fe + E + 6 * (size_t)E;
    std::vector<float> wf((size_t)E + (size_t)S * E + 2 * (size_t)E + per_layerf * g.enc_layers);
    size_t o16 = 0, of = 0;
    std::vector<size_t> off16, offf;
    auto mark16 = [&](size_t n) { off16.push_back(o16); o16 += n; };
    auto markf = [&](size_t n) { offf.push_back(of); of += n; };
    // patch embedding: ((p/255 - mean)/std) . w  ==  (p - 128) . w' / 256 + const with w' = 256 w / (255 std)
    // (x256 keeps small weights in the 16-bit normal range); w' is stored split [hi | lo] along K.
    const double mean[3] = {0.485, 0.456, 0.406}, sd[3] = {0.229, 0.224, 0.225};
    const int Kp1 = Kp / 2;
    mark16((size_t)E * Kp);
    markf(E);
    auto back = [&](uint16_t h) -> float { return pack::from16(h, bf); };
    for (int nn = 0; nn < E; ++nn) {
      double bacc = e_pb[nn];
      for (int k = 0; k < Kp1; ++k) {
        uint16_t hi = 0, lo = 0;
        if (k < Kreal) {
          const int c = k % 3;
          const double wk = e_pk[(size_t)k * E + nn];
          const float w = (float)(wk * 256.0 / (255.0 * sd[c]));
          hi = cv(w);
          lo = cv(w - back(hi));
          bacc += wk * (128.0 / 255.0 - mean[c]) / sd[c];
        }
        w16[off16.back() + (size_t)nn * Kp + k] = hi;
        w16[off16.back() + (size_t)nn * Kp + Kp1 + k] = lo;
      }
      wf[offf.back() + nn] = (float)bacc;
    }
    markf((size_t)S * E);
    for (size_t i = 0; i < (size_t)S * E; ++i) wf[offf.back() + i] = e_pos[i] + (i < (size_t)E ? e_cls[i] : 0.f);
    markf(E); memcpy(&wf[offf.back()], e_lns, E * 4);
    markf(E); memcpy(&wf[offf.back()], e_lnb, E * 4);
    // flax [K][N] -> [N][K] 16-bit, and what the rounding dropped (x 4096: stays in the normal range of fp16) for the
    // per-image compensation of the encoder GEMMs (encoder.hip corr_kernel)
    auto tr = [&](const float* src, int K, int N, size_t dst) { pack::pack_matrix_t(src, K, N, bf, &w16[dst], &d16[dst]); };
    for (int i = 0; i < g.enc_layers; ++i) {
      const LSrc& s = ls[i];
      mark16((size_t)3 * E * E);
      tr(s.q, E, E, off16.back()); tr(s.k, E, E, off16.back() + (size_t)E * E); tr(s.v, E, E, off16.back() + (size_t)2 * E * E);
      mark16((size_t)E * E); tr(s.o, E, E, off16.back());
      mark16((size_t)E * Fe); tr(s.f1, E, Fe, off16.back());
      mark16((size_t)Fe * E); tr(s.f2, Fe, E, off16.back());
      markf(3 * E);
      memcpy(&wf[offf.back()], s.qb, E * 4); memcpy(&wf[offf.back() + E], s.kb, E * 4); memcpy(&wf[offf.back() + 2 * E], s.vb, E * 4);
      markf(E); memcpy(&wf[offf.back()], s.ob, E * 4);
      markf(Fe); memcpy(&wf[offf.back()], s.f1b, Fe * 4);
      markf(E); memcpy(&wf[offf.back()], s.f2b, E * 4);
      const float* six[6] = {s.n1s, s.n1b, s.n2s, s.n2b, s.l1, s.l2};
      for (int q = 0; q < 6; ++q) { markf(E); memcpy(&wf[offf.back()], six[q], E * 4); }
    }
    HIPCHK(ctx, ctx->enc16.alloc(w16.size() * 2));
    HIPCHK(ctx, ctx->encd16.alloc(d16.size() * 2));
    HIPCHK(ctx, ctx->encf32.alloc(wf.size() * 4));
    HIPCHK(ctx, hipMemcpy(ctx->enc16.p, w16.data(), w16.size() * 2, hipMemcpyHostToDevice));
    HIPCHK(ctx, hipMemcpy(ctx->encd16.p, d16.data(), d16.size() * 2, hipMemcpyHostToDevice));
    HIPCHK(ctx, hipMemcpy(ctx->encf32.p, wf.data(), wf.size() * 4, hipMemcpyHostToDevice));
    const uint16_t* e16 = ctx->enc16.as<uint16_t>();
    const float* df = ctx->encf32.as<float>();
    EncWeights& w = ctx->encw;
    size_t i16 = 0, iff = 0;
    w.w_patch = e16 + off16[i16++];
    w.b_patch = df + offf[iff++];
    w.pos = df + offf[iff++];
    w.lnf_s = df + offf[iff++];
    w.lnf_b = df + offf[iff++];
    for (int i = 0; i < g.enc_layers; ++i) {
      EncLayerW& L = w.layer[i];
      const uint16_t* dd = ctx->encd16.as<uint16_t>();
      L.dqkv = dd + off16[i16]; L.wqkv = e16 + off16[i16++]; L.dwo = dd + off16[i16]; L.wo = e16 + off16[i16++];
      L.dw1 = dd + off16[i16]; L.w1 = e16 + off16[i16++]; L.dw2 = dd + off16[i16]; L.w2 = e16 + off16[i16++];
      L.bqkv = df + offf[iff++]; L.bo = df + offf[iff++]; L.b1 = df + offf[iff++]; L.b2 = df + offf[iff++];
      L.ln1_s = df + offf[iff++]; L.ln1_b = df + offf[iff++]; L.ln2_s = df + offf[iff++]; L.ln2_b = df + offf[iff++];
      L.ls1 = df + offf[iff++]; L.ls2 = df + offf[iff++];
    }
  }
  HIPCHK(ctx, hipDeviceSynchronize());
  ctx->loaded = true;
  return HVLA_OK;
}

int hvla_generate(hvla_ctx* ctx, const float* tok, const int64_t* mask, const float* cls, int32_t B,
                  hvla_weights** out, void* stream) {
  if (!ctx || !out) return HVLA_E_STATE;
  *out = nullptr;
  if (!ctx->loaded) FAIL(ctx, HVLA_E_STATE, "hvla_generate before hvla_load_weights");
  if (B < 1 || B > ctx->cfg.max_batch) FAIL(ctx, HVLA_E_SHAPE, "batch %d outside [1, %d]", B, ctx->cfg.max_batch);
  if (!tok || !mask || !cls) FAIL(ctx, HVLA_E_SHAPE, "null input pointer");
  HIPCHK(ctx, hipSetDevice(ctx->device));
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const PolicyLayout& pl = ctx->lay.pl;
  const Geom& g = ctx->g;
  std::unique_ptr<hvla_weights> w;
  {
    std::lock_guard<std::mutex> lk(ctx->pool_mu);
    for (size_t i = 0; i < ctx->arena_pool.size(); ++i)        // an arena of this batch size handed back earlier
      if (ctx->arena_pool[i]->B == B) {
        w.reset(ctx->arena_pool[i]);
        ctx->arena_pool.erase(ctx->arena_pool.begin() + i);
        break;
      }
  }
  if (!w) {
    w.reset(new hvla_weights);
    w->B = B;
    hipError_t e = hipSuccess;
    auto A = [&](DevBuf& b, size_t n) { if (e == hipSuccess) e = b.alloc(n); };
    A(w->wh, (size_t)B * pl.Gm * 2); A(w->wl, (size_t)B * pl.Gm * 2); A(w->vf, (size_t)B * pl.Gv * 4);
    A(w->ctx, (size_t)B * g.C * 4);
    A(w->ring, (size_t)g.horizon * B * g.horizon * g.action_dim * 4); A(w->count, 16);
    if (e != hipSuccess) {
      {
        std::lock_guard<std::mutex> lk(ctx->pool_mu);
        for (hvla_weights* q : ctx->arena_pool) delete q;      // give the pooled arenas back to the device and retry once
        ctx->arena_pool.clear();
      }
      e = hipSuccess;
      A(w->wh, (size_t)B * pl.Gm * 2); A(w->wl, (size_t)B * pl.Gm * 2); A(w->vf, (size_t)B * pl.Gv * 4);
      A(w->ctx, (size_t)B * g.C * 4);
      A(w->ring, (size_t)g.horizon * B * g.horizon * g.action_dim * 4); A(w->count, 16);
    }
    if (e != hipSuccess) FAIL(ctx, HVLA_E_ARENA_FULL, "weight arena for %d episodes: %s", B, hipGetErrorString(e));
  }
  HIPCHK(ctx, hipMemsetAsync(w->count.p, 0, 16, st));
  CtxParams cp = ctx->ctxp;
  cp.tok = tok; cp.attn_mask = mask; cp.cls = cls;
  cp.ctx = w->ctx.as<float>(); cp.ctx_hi = ctx->ctx_hi.as<__bf16>(); cp.ctx_lo = ctx->ctx_lo.as<__bf16>();
  HIPCHK(ctx, launch_ctx_encoder(cp, B, st));
  WeightGenParams wp{ctx->wcat_hi.as<__bf16>(), ctx->wcat_lo.as<__bf16>(), ctx->bcat.as<float>(),
                     ctx->ctx_hi.as<__bf16>(), ctx->ctx_lo.as<__bf16>(), w->wh.as<__bf16>(), w->wl.as<__bf16>(),
                     w->vf.as<float>(), B, pl.Gm, pl.Gv, (pl.Gm + pl.Gv) / 32};
  HIPCHK(ctx, launch_weightgen(wp, g.C, st));
  *out = w.release();
  return HVLA_OK;
}

int hvla_weights_free(hvla_ctx* ctx, hvla_weights* w) {
  if (!w) return HVLA_OK;
  if (!ctx) { delete w; return HVLA_OK; }
  (void)hipSetDevice(ctx->device);
  // hipFree waited for the device; a pooled arena must give the same guarantee before another stream's hvla_generate
  // writes into it (this call has no stream of its own: frees happen at episode resets, not in the step loop)
  (void)hipDeviceSynchronize();
  std::lock_guard<std::mutex> lk(ctx->pool_mu);
  if (ctx->arena_pool.size() < hvla_ctx::ARENA_POOL_MAX) ctx->arena_pool.push_back(w);
  else delete w;
  return HVLA_OK;
}

int hvla_release_pooled_arenas(hvla_ctx* ctx) {
  if (!ctx) return HVLA_E_STATE;
  (void)hipSetDevice(ctx->device);
  (void)hipDeviceSynchronize();
  std::lock_guard<std::mutex> lk(ctx->pool_mu);
  for (hvla_weights* q : ctx->arena_pool) delete q;
  ctx->arena_pool.clear();
  return HVLA_OK;
}

int32_t hvla_weights_batch(const hvla_weights* w) { return w ? w->B : 0; }

int hvla_weights_export(hvla_ctx* ctx, const hvla_weights* w, float* theta, float* context, void* stream) {
  if (!ctx || !w) return HVLA_E_STATE;
  HIPCHK(ctx, hipSetDevice(ctx->device));
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const PolicyLayout& pl = ctx->lay.pl;
  if (theta)
    HIPCHK(ctx, launch_export_theta(w->wh.as<__bf16>(), w->wl.as<__bf16>(), w->vf.as<float>(), ctx->perm.as<int32_t>(),
                                    pl.Gm, pl.Gv, pl.G, w->B, theta, st));
  if (context)
    HIPCHK(ctx, hipMemcpyAsync(context, w->ctx.p, (size_t)w->B * ctx->g.C * 4, hipMemcpyDeviceToDevice, st));
  return HVLA_OK;
}

static int check_step(hvla_ctx* ctx, int32_t B) {
  if (!ctx->loaded) FAIL(ctx, HVLA_E_STATE, "called before hvla_load_weights");
  if (B < 1 || B > ctx->cfg.max_batch) FAIL(ctx, HVLA_E_SHAPE, "batch %d outside [1, %d]", B, ctx->cfg.max_batch);
  return HVLA_OK;
}

// episodes [b0, b0 + nb) of the batch: every per-episode buffer is offset, the workspace slices are disjoint
static int encode_range(hvla_ctx* ctx, const uint8_t* images, float* out, int b0, int nb, bool keep_cls, hipStream_t st) {
  const Geom& g = ctx->g;
  const size_t S = g.S(), E = g.E, F = g.enc_mlp, rows = (size_t)b0 * S;
  // ws_g holds an episode's MLP hidden rows [S][F] and, before that, its im2col rows [P][Kp]: the slices of two concurrent
  // halves must be disjoint for the larger of the two
  const size_t gper = S * F > (size_t)g.P() * ctx->Kp ? S * F : (size_t)g.P() * ctx->Kp;
  EncWorkspace ws{ctx->ws_x.as<float>() + rows * E, static_cast<char*>(ctx->ws_h.p) + rows * E * 2,
                  static_cast<char*>(ctx->ws_qkv.p) + rows * 3 * E * 2, static_cast<char*>(ctx->ws_g.p) + (size_t)b0 * gper * 2,
                  ctx->ws_corr.as<float>() + (size_t)2 * b0 * (F > 3 * E ? F : 3 * E),
                  static_cast<char*>(ctx->ws_abar.p) + (size_t)2 * b0 * (F > E ? F : E) * 2};
  if (ctx->amap_dino) ws.amap = ctx->amap_dino + (size_t)b0 * g.enc_layers * g.enc_heads * g.P();
  ws.ln_cnt = ctx->ws_lncnt.as<uint32_t>() + (size_t)((b0 + 3) / 4 * 4);   // (a second half starts on a 16-byte boundary)
  ws.ln_part = ctx->ws_lnpart.as<float>() + (size_t)b0 * 4 * 256 * 4;
  ws.ln_spin = ctx->ln_spin;
#ifdef HVLA_BENCH_HOOKS
  ws.stop_after = ctx->enc_stop;
#endif
  const size_t img = (size_t)g.image_size * g.image_size * 3, per = (keep_cls ? S : (size_t)g.P()) * E;
  HIPCHK(ctx, launch_encoder(g, ctx->cfg.enc_dtype, ctx->encw, ws, images + (size_t)b0 * img, out + (size_t)b0 * per, nb, st,
                             &ctx->prof, keep_cls));
  return HVLA_OK;
}

int hvla_encode(hvla_ctx* ctx, const uint8_t* images, float* tokens, int32_t B, void* stream) {
  if (!ctx) return HVLA_E_STATE;
  if (int r = check_step(ctx, B)) return r;
  if (!images || !tokens) FAIL(ctx, HVLA_E_SHAPE, "null pointer");
  HIPCHK(ctx, hipSetDevice(ctx->device));
  return encode_range(ctx, images, tokens, 0, B, false, reinterpret_cast<hipStream_t>(stream));
}

int hvla_encode_hidden(hvla_ctx* ctx, const uint8_t* images, float* hidden, int32_t B, void* stream) {
  if (!ctx) return HVLA_E_STATE;
  if (int r = check_step(ctx, B)) return r;
  if (!images || !hidden) FAIL(ctx, HVLA_E_SHAPE, "null pointer");
  HIPCHK(ctx, hipSetDevice(ctx->device));
  return encode_range(ctx, images, hidden, 0, B, true, reinterpret_cast<hipStream_t>(stream));
}

// test instrumentation: the encoder with a range audit of every 16-bit MFMA operand it writes (LayerNorm outputs, q / k /
// v, attention outputs, GELU outputs; all layers).  maxabs f32 [4], nonfinite i32 [4]: HOST pointers.
int hvla_encode_audit(hvla_ctx* ctx, const uint8_t* images, int32_t B, float* maxabs, int32_t* nonfinite, void* stream) {
  if (!ctx) return HVLA_E_STATE;
  if (int r = check_step(ctx, B)) return r;
  if (!images || !maxabs || !nonfinite) FAIL(ctx, HVLA_E_SHAPE, "null pointer");
  HIPCHK(ctx, hipSetDevice(ctx->device));
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  uint32_t* slots = reinterpret_cast<uint32_t*>(ctx->flags.p);
  HIPCHK(ctx, hipMemsetAsync(slots, 0, 8 * sizeof(uint32_t), st));
  const Geom& g = ctx->g;
  const size_t F = g.enc_mlp, E = g.E;
  EncWorkspace ws{ctx->ws_x.as<float>(), ctx->ws_h.p, ctx->ws_qkv.p, ctx->ws_g.p, ctx->ws_corr.as<float>(), ctx->ws_abar.p};
  (void)F; (void)E;
  ws.ln_cnt = ctx->ws_lncnt.as<uint32_t>(); ws.ln_part = ctx->ws_lnpart.as<float>(); ws.ln_spin = ctx->ln_spin;
  HIPCHK(ctx, launch_encoder(g, ctx->cfg.enc_dtype, ctx->encw, ws, images, ctx->tokens.as<float>(), B, st, nullptr, false, slots));
  uint32_t h[8];
  HIPCHK(ctx, hipMemcpyAsync(h, slots, sizeof h, hipMemcpyDeviceToHost, st));
  HIPCHK(ctx, hipStreamSynchronize(st));
  for (int i = 0; i < 4; ++i) {
    memcpy(&maxabs[i], &h[2 * i], 4);
    nonfinite[i] = (int32_t)h[2 * i + 1];
  }
  return HVLA_OK;
}

static int policy_range(hvla_ctx* ctx, const hvla_weights* w, const float* tokens, float* actions, float* logits, int b0,
                        int nb, hipStream_t st) {
  const Geom& g = ctx->g;
  const PolicyLayout& pl = ctx->lay.pl;
  PolicyParams p{pl, w->wh.as<__bf16>() + (size_t)b0 * pl.Gm, w->wl.as<__bf16>() + (size_t)b0 * pl.Gm,
                 w->vf.as<float>() + (size_t)b0 * pl.Gv, tokens + (size_t)b0 * g.P() * g.E,
                 actions + (size_t)b0 * g.horizon * g.action_dim, logits ? logits + (size_t)b0 * g.horizon : nullptr,
                 nb, g.E, g.P(), g.L, g.M, g.horizon, g.action_dim, g.tanh_scale, g.max_action};
  if (ctx->amap_head) p.amap = ctx->amap_head + (size_t)b0 * g.L * g.H * g.P();
  ctx->prof.begin(HVLA_PROF_POLICY, st);
  ++ctx->prof.nlaunch;
  HIPCHK(ctx, launch_policy(p, st));
  ctx->prof.end(HVLA_PROF_POLICY, st);
  return HVLA_OK;
}

int hvla_policy(hvla_ctx* ctx, const hvla_weights* w, const float* tokens, float* actions, float* logits, int32_t B,
                void* stream) {
  if (!ctx || !w) return HVLA_E_STATE;
  if (int r = check_step(ctx, B)) return r;
  if (B != w->B) FAIL(ctx, HVLA_E_SHAPE, "batch %d != arena batch %d", B, w->B);
  if (!tokens || !actions) FAIL(ctx, HVLA_E_SHAPE, "null pointer");
  HIPCHK(ctx, hipSetDevice(ctx->device));
  return policy_range(ctx, w, tokens, actions, logits, 0, B, reinterpret_cast<hipStream_t>(stream));
}

int hvla_set_attention_outputs(hvla_ctx* ctx, float* dino_cls_attention, float* head_attention) {
  if (!ctx) return HVLA_E_STATE;
  ctx->amap_dino = dino_cls_attention;
  ctx->amap_head = head_attention;
  return HVLA_OK;
}

int hvla_step(hvla_ctx* ctx, const hvla_weights* w, const uint8_t* images, float* actions, float* logits, int32_t B,
              void* stream) {
  if (!ctx || !w) return HVLA_E_STATE;
  if (int r = check_step(ctx, B)) return r;
  if (B != w->B) FAIL(ctx, HVLA_E_SHAPE, "batch %d != arena batch %d", B, w->B);
  if (!images || !actions) FAIL(ctx, HVLA_E_SHAPE, "null pointer");
  HIPCHK(ctx, hipSetDevice(ctx->device));
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  float* tokens = ctx->tokens.as<float>();
  if (!ctx->side || B < 64) {
    if (int r = encode_range(ctx, images, tokens, 0, B, false, st)) return r;
    return policy_range(ctx, w, tokens, actions, logits, 0, B, st);
  }
  // two halves on two streams: while one half is in an HBM-bound kernel (LayerNorm, a residual epilogue, attention
  // staging) or in the ragged end of a grid, the other half's GEMM has the matrix cores.  Episodes are independent and
  // every kernel is batch-invariant, so the bytes are those of the single-stream step.
  const int b0 = B / 2;
  HIPCHK(ctx, hipEventRecord(ctx->ev_fork, st));
  HIPCHK(ctx, hipStreamWaitEvent(ctx->side, ctx->ev_fork, 0));
  if (int r = encode_range(ctx, images, tokens, 0, b0, false, st)) return r;
  if (int r = encode_range(ctx, images, tokens, b0, B - b0, false, ctx->side)) return r;
  if (int r = policy_range(ctx, w, tokens, actions, logits, 0, b0, st)) return r;
  if (int r = policy_range(ctx, w, tokens, actions, logits, b0, B - b0, ctx->side)) return r;
  HIPCHK(ctx, hipEventRecord(ctx->ev_join, ctx->side));
  HIPCHK(ctx, hipStreamWaitEvent(st, ctx->ev_join, 0));
  return HVLA_OK;
}

int hvla_ensemble_reset(hvla_ctx* ctx, hvla_weights* w, void* stream) {
  if (!ctx || !w) return HVLA_E_STATE;
  HIPCHK(ctx, hipMemsetAsync(w->count.p, 0, 16, reinterpret_cast<hipStream_t>(stream)));
  return HVLA_OK;
}

int hvla_ensemble(hvla_ctx* ctx, hvla_weights* w, const float* actions, const float* mean, const float* std,
                  const uint8_t* mask, float* out, void* stream) {
  if (!ctx || !w) return HVLA_E_STATE;
  if (!actions || !mean || !std || !mask || !out) FAIL(ctx, HVLA_E_SHAPE, "null pointer");
  HIPCHK(ctx, hipSetDevice(ctx->device));
  ++ctx->prof.nlaunch;
  HIPCHK(ctx, launch_ensemble(actions, w->ring.as<float>(), w->count.as<int>(), mean, std, mask, out, w->B,
                              ctx->g.horizon, ctx->g.action_dim, reinterpret_cast<hipStream_t>(stream)));
  return HVLA_OK;
}

int hvla_loss(hvla_ctx* ctx, const float* actions, const float* logits, const float* target, const uint8_t* tmask,
              const uint8_t* amask, float* loss, int32_t B, void* stream) {
  if (!ctx) return HVLA_E_STATE;
  if (B < 1) FAIL(ctx, HVLA_E_SHAPE, "batch %d", B);
  if (!actions || !logits || !target || !tmask || !amask || !loss) FAIL(ctx, HVLA_E_SHAPE, "null pointer");
  HIPCHK(ctx, hipSetDevice(ctx->device));
  HIPCHK(ctx, launch_loss(actions, logits, target, tmask, amask, loss, B, ctx->g.horizon, ctx->g.action_dim,
                          ctx->g.max_action, ctx->g.clip_target != 0, reinterpret_cast<hipStream_t>(stream)));
  return HVLA_OK;
}

int hvla_train_sizes(hvla_ctx* ctx, int32_t B, int32_t train_encoder, int64_t out[4]) {
  if (!ctx || !out) return HVLA_E_STATE;
  if (B < 1) FAIL(ctx, HVLA_E_SHAPE, "batch %d", B);
  if (ctx->g.ctx_layers > 8 || ctx->g.L > 16 || ctx->g.enc_layers > 24) FAIL(ctx, HVLA_E_SHAPE, "too many layers for the training path");
  const TrainLayout L = make_train_layout(ctx->g);
  out[0] = L.total + (train_encoder ? L.enc_total : 0); out[1] = L.G;
  out[2] = (int64_t)train_workspace_floats(ctx->g, B, train_encoder != 0); out[3] = L.total;
  return HVLA_OK;
}

static TrainBuffers to_tb(const hvla_train_buffers* b) {
  return TrainBuffers{b->params, b->grads, reinterpret_cast<__bf16*>(b->mu), b->nu, b->ema, b->theta, b->dtheta, b->work,
                      b->loss, b->actions, b->logits, b->sqsum, b->wd_mask, b->params0};
}
static TrainHyper to_hp(const hvla_train_hyper* hy) {
  return TrainHyper{hy->lr, hy->b1, hy->b2, hy->eps, hy->weight_decay, hy->clip, hy->ema_decay, hy->step, hy->forward_only,
                    hy->base_lr, hy->base_weight_decay};
}

int hvla_train_step(hvla_ctx* ctx, const hvla_train_buffers* buf, const float* tok, const int64_t* mask, const float* cls,
                    const float* tokens, const uint8_t* images, const float* target, const uint8_t* tmask,
                    const uint8_t* amask, int32_t B, const hvla_train_hyper* hy, void* stream) {
  if (!ctx || !buf || !hy) return HVLA_E_STATE;
  if (B < 1) FAIL(ctx, HVLA_E_SHAPE, "batch %d", B);
  if (!buf->params || !buf->grads || !buf->theta || !buf->dtheta || !buf->work || !buf->loss || !tok || !mask || !cls ||
      !target || !tmask || !amask)
    FAIL(ctx, HVLA_E_SHAPE, "null pointer");
  if ((tokens != nullptr) == (images != nullptr)) FAIL(ctx, HVLA_E_SHAPE, "pass exactly one of tokens (frozen encoder) / images (trained encoder)");
  if ((images != nullptr) != (hy->train_encoder != 0)) FAIL(ctx, HVLA_E_STATE, "hyper.train_encoder does not match the inputs");
  if (ctx->g.ctx_layers > 8 || ctx->g.L > 16 || ctx->g.enc_layers > 24) FAIL(ctx, HVLA_E_SHAPE, "too many layers for the training path");
  HIPCHK(ctx, hipSetDevice(ctx->device));
  const TrainLayout L = make_train_layout(ctx->g);
  TrainInputs in{tok, mask, cls, tokens, images, target, tmask, amask};
  const TrainHyper hp = to_hp(hy);
  for (hipEvent_t& e : ctx->ev_bucket)
    if (!e) HIPCHK(ctx, hipEventCreateWithFlags(&e, hipEventDisableTiming));
  HIPCHK(ctx, train_step(ctx->g, L, to_tb(buf), in, B, hp, reinterpret_cast<hipStream_t>(stream), hp.forward_only ? nullptr : ctx->ev_bucket));
  // ONE pending backward per ctx: the bucket events belong to the last hvla_train_step that ran a backward pass, and
  // hvla_train_wait_bucket refers to that step.  A forward-only step (evaluation between a step and its apply) records
  // nothing and leaves the pending step's events alone.
  if (!hp.forward_only) {
    ctx->bucket_recorded[0] = images != nullptr;
    ctx->bucket_recorded[1] = ctx->bucket_recorded[2] = true;
  }
  return HVLA_OK;
}

int hvla_train_bucket_ranges(hvla_ctx* ctx, int32_t train_encoder, int64_t out[6]) {
  if (!ctx || !out) return HVLA_E_STATE;
  const TrainLayout L = make_train_layout(ctx->g);
  out[0] = L.total; out[1] = train_encoder ? L.enc_total : 0;     // the shared DINOv2 leaves
  out[2] = L.wcat; out[3] = L.total - L.wcat;                     // the output heads (W_cat, b_cat)
  out[4] = 0; out[5] = L.wcat;                                    // the context encoder
  return HVLA_OK;
}

int hvla_train_wait_bucket(hvla_ctx* ctx, int32_t bucket, void* stream) {
  if (!ctx) return HVLA_E_STATE;
  if (bucket < 0 || bucket > 2) FAIL(ctx, HVLA_E_SHAPE, "bucket %d outside [0, 2]", bucket);
  if (!ctx->bucket_recorded[bucket]) FAIL(ctx, HVLA_E_STATE, "bucket %d was not produced by the last hvla_train_step", bucket);
  HIPCHK(ctx, hipSetDevice(ctx->device));
  HIPCHK(ctx, hipStreamWaitEvent(reinterpret_cast<hipStream_t>(stream), ctx->ev_bucket[bucket], 0));
  return HVLA_OK;
}

int hvla_train_profile(hvla_ctx* ctx, int32_t on) {
  if (!ctx) return HVLA_E_STATE;
  HIPCHK(ctx, hipSetDevice(ctx->device));
  const bool first = on && !ctx->train_timer;
  train_gemm_timer(on != 0, first);
  if (on) ctx->train_timer = true;
  return HVLA_OK;
}

int hvla_train_profile_read(hvla_ctx* ctx, float* gemm_ms, double* gemm_flops, int32_t* launches) {
  if (!ctx || !gemm_ms || !gemm_flops || !launches) return HVLA_E_STATE;
  HIPCHK(ctx, hipSetDevice(ctx->device));
  int n = 0;
  HIPCHK(ctx, train_gemm_timer_read(gemm_ms, gemm_flops, &n));
  *launches = n;
  return HVLA_OK;
}

int hvla_train_apply(hvla_ctx* ctx, const hvla_train_buffers* buf, const hvla_train_hyper* hy, void* stream) {
  if (!ctx || !buf || !hy) return HVLA_E_STATE;
  if (!buf->params || !buf->grads || !buf->mu || !buf->nu || !buf->sqsum) FAIL(ctx, HVLA_E_SHAPE, "null pointer");
  HIPCHK(ctx, hipSetDevice(ctx->device));
  const TrainLayout L = make_train_layout(ctx->g);
  HIPCHK(ctx, train_apply(L, to_tb(buf), to_hp(hy), hy->train_encoder != 0, reinterpret_cast<hipStream_t>(stream)));
  return HVLA_OK;
}

int hvla_train_accumulate(hvla_ctx* ctx, const hvla_train_buffers* buf, float* acc, float inv_k, const hvla_train_hyper* hy,
                          void* stream) {
  if (!ctx || !buf || !hy) return HVLA_E_STATE;
  if (!buf->grads || !buf->sqsum || !acc) FAIL(ctx, HVLA_E_SHAPE, "null pointer");
  HIPCHK(ctx, hipSetDevice(ctx->device));
  const TrainLayout L = make_train_layout(ctx->g);
  HIPCHK(ctx, train_accumulate(L, to_tb(buf), acc, inv_k, to_hp(hy), hy->train_encoder != 0, reinterpret_cast<hipStream_t>(stream)));
  return HVLA_OK;
}

int hvla_preprocess(hvla_ctx* ctx, const uint8_t* frames, int32_t B, int32_t H, int32_t W, int32_t flags, uint8_t* images,
                    void* stream) {
  if (!ctx) return HVLA_E_STATE;
  if (B < 1 || H < 2 || W < 2 || H > 8192 || W > 8192) FAIL(ctx, HVLA_E_SHAPE, "frames [%d, %d, %d, 3]", B, H, W);
  if (!frames || !images) FAIL(ctx, HVLA_E_SHAPE, "null pointer");
  if (flags & ~3) FAIL(ctx, HVLA_E_SHAPE, "flags %d (1 = crop, 2 = padded resize)", flags);
  HIPCHK(ctx, hipSetDevice(ctx->device));
  const int S = ctx->g.image_size;
  const bool crop = flags & HVLA_PREPROCESS_CROP, pad = flags & HVLA_PREPROCESS_PAD;
  const int LH = pad ? 256 : H, LW = pad ? 320 : W;           // what the lanczos3 stage sees (hypervla_interface.py:90-95)
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (LH != ctx->rs_H || LW != ctx->rs_W) {                    // new source shape: rebuild the span tables
    std::vector<int> rs, rc, cs, cc;
    std::vector<float> rw, cw;
    build_resize_spans(LH, S, rs, rc, rw, ctx->rs_row_span);
    build_resize_spans(LW, S, cs, cc, cw, ctx->rs_col_span);
    std::vector<float> host;                                    // [row start | row count | col start | col count] as ints, then weights
    host.resize((size_t)4 * S + rw.size() + cw.size());
    memcpy(host.data(), rs.data(), (size_t)S * 4); memcpy(host.data() + S, rc.data(), (size_t)S * 4);
    memcpy(host.data() + 2 * S, cs.data(), (size_t)S * 4); memcpy(host.data() + 3 * S, cc.data(), (size_t)S * 4);
    memcpy(host.data() + 4 * S, rw.data(), rw.size() * 4); memcpy(host.data() + 4 * S + rw.size(), cw.data(), cw.size() * 4);
    HIPCHK(ctx, hipStreamSynchronize(st));                      // a previous call may still read the old tables
    HIPCHK(ctx, ctx->rs_tab.alloc(host.size() * 4));
    HIPCHK(ctx, hipMemcpy(ctx->rs_tab.p, host.data(), host.size() * 4, hipMemcpyHostToDevice));
    ctx->rs_H = LH; ctx->rs_W = LW;
  }
  const size_t need_rows = (size_t)B * S * LW * 3 * 4, need_img = (size_t)B * S * S * 3 * 4;
  const size_t need_pad = pad ? (size_t)B * LH * LW * 3 * 4 : 0;
  if (ctx->rs_rows.bytes < need_rows || ctx->rs_img.bytes < need_img || ctx->rs_pad.bytes < need_pad) {
    HIPCHK(ctx, hipStreamSynchronize(st));
    if (ctx->rs_rows.bytes < need_rows) HIPCHK(ctx, ctx->rs_rows.alloc(need_rows));
    if (ctx->rs_img.bytes < need_img) HIPCHK(ctx, ctx->rs_img.alloc(need_img));
    if (ctx->rs_pad.bytes < need_pad) HIPCHK(ctx, ctx->rs_pad.alloc(need_pad));
  }
  const int* ti = ctx->rs_tab.as<int>();
  const float* tw = ctx->rs_tab.as<float>() + 4 * S;
  HIPCHK(ctx, launch_resize(frames, images, ctx->rs_rows.as<float>(), ctx->rs_img.as<float>(), ti, ti + S, tw,
                            ctx->rs_row_span, ti + 2 * S, ti + 3 * S, tw + (size_t)S * ctx->rs_row_span, ctx->rs_col_span, B, LH,
                            LW, S, crop, st, pad ? ctx->rs_pad.as<float>() : nullptr, H, W));
  return HVLA_OK;
}

// bucket of relative position rel = key - query, bidirectional (transformers `_relative_position_bucket`; the log is
// evaluated in float32 as both the torch and the flax model do)
static int t5_bucket(int rel, int buckets, int max_distance) {
  const int nb = buckets / 2, max_exact = nb / 2;
  int out = rel > 0 ? nb : 0;
  const int n = rel < 0 ? -rel : rel;
  if (n < max_exact) return out + n;
  const float v = logf((float)n / (float)max_exact) / (float)log((double)max_distance / (double)max_exact) * (float)(nb - max_exact);
  int large = max_exact + (int)v;
  if (large > nb - 1) large = nb - 1;
  return out + large;
}

int hvla_t5_load(hvla_ctx* ctx, const hvla_t5_config* c, const hvla_tensor_desc* t, int32_t n) {
  if (!ctx || !c) return HVLA_E_STATE;
  if (c->layers < 1 || c->layers > 24 || c->heads < 1 || c->d_model < 1 || c->d_kv < 1 || c->d_ff < 1 || c->vocab < 1 ||
      c->buckets < 4 || c->buckets % 2 || c->max_tokens < 1 || c->max_batch < 1)
    FAIL(ctx, HVLA_E_SHAPE, "bad T5 configuration");
  if (c->d_model != ctx->g.lang_dim) FAIL(ctx, HVLA_E_SHAPE, "T5 d_model %d != lang_dim %d of the hypernetwork", c->d_model, ctx->g.lang_dim);
  HIPCHK(ctx, hipSetDevice(ctx->device));
  std::map<std::string, const hvla_tensor_desc*> m;
  for (int i = 0; i < n; ++i) {
    if (!t[i].name || !t[i].data) FAIL(ctx, HVLA_E_WEIGHTS, "tensor %d has null name/data", i);
    std::string nm = t[i].name;
    if (nm.rfind("hf_model/", 0) == 0) nm = nm.substr(9);          // the reference nests the HF module under `hf_model`
    m[nm] = &t[i];
  }
  const int D = c->d_model, I = c->heads * c->d_kv, F = c->d_ff, TM = c->max_tokens;
  std::vector<float> host;
  std::vector<std::pair<const float**, size_t>> fix;
  std::string bad;
  auto push = [&](const float** slot, const std::string& name, int64_t numel) {
    auto it = m.find(name);
    if (it == m.end() || it->second->numel != numel) {
      if (bad.empty()) bad = name + (it == m.end() ? " (absent)" : " (wrong size)");
      return;
    }
    const size_t off = (host.size() + 3) / 4 * 4;
    host.resize(off + numel);
    memcpy(host.data() + off, it->second->data, (size_t)numel * 4);
    fix.emplace_back(slot, off);
  };
  T5Weights w{};
  w.max_tokens = TM;
  push(&w.shared, "shared/embedding", (int64_t)c->vocab * D);
  push(&w.final_ln, "encoder/final_layer_norm/weight", D);
  for (int l = 0; l < c->layers; ++l) {
    const std::string b = "encoder/block/" + std::to_string(l) + "/layer/";
    T5LayerW& L = w.layer[l];
    push(&L.ln0, b + "0/layer_norm/weight", D);
    push(&L.wq, b + "0/SelfAttention/q/kernel", (int64_t)D * I);
    push(&L.wk, b + "0/SelfAttention/k/kernel", (int64_t)D * I);
    push(&L.wv, b + "0/SelfAttention/v/kernel", (int64_t)D * I);
    push(&L.wo, b + "0/SelfAttention/o/kernel", (int64_t)I * D);
    push(&L.ln1, b + "1/layer_norm/weight", D);
    push(&L.wi, b + "1/DenseReluDense/wi/kernel", (int64_t)D * F);
    push(&L.wo2, b + "1/DenseReluDense/wo/kernel", (int64_t)F * D);
  }
  // relative-position bias table [heads][2 TM - 1] from the first block's bucket embedding [buckets][heads]
  {
    auto it = m.find("encoder/block/0/layer/0/SelfAttention/relative_attention_bias/embedding");
    if (it == m.end() || it->second->numel != (int64_t)c->buckets * c->heads) {
      if (bad.empty()) bad = "encoder/block/0/layer/0/SelfAttention/relative_attention_bias/embedding";
    } else {
      const float* emb = it->second->data;
      const size_t off = (host.size() + 3) / 4 * 4;
      host.resize(off + (size_t)c->heads * (2 * TM - 1));
      for (int h = 0; h < c->heads; ++h)
        for (int r = -(TM - 1); r <= TM - 1; ++r)
          host[off + (size_t)h * (2 * TM - 1) + (r + TM - 1)] = emb[(size_t)t5_bucket(r, c->buckets, c->max_distance) * c->heads + h];
      fix.emplace_back(&w.relbias, off);
    }
  }
  if (!bad.empty()) FAIL(ctx, HVLA_E_WEIGHTS, "T5 tensor %s", bad.c_str());
  HIPCHK(ctx, ctx->t5_f32.alloc(host.size() * 4));
  HIPCHK(ctx, hipMemcpy(ctx->t5_f32.p, host.data(), host.size() * 4, hipMemcpyHostToDevice));
  for (auto& f : fix) *f.first = ctx->t5_f32.as<float>() + f.second;
  ctx->t5d = T5Dims{c->vocab, D, c->d_kv, c->heads, F, c->layers, c->buckets, c->max_distance, c->eps};
  ctx->t5w = w;
  ctx->t5_max_batch = c->max_batch;
  HIPCHK(ctx, ctx->t5_work.alloc(t5_workspace_floats(ctx->t5d, c->max_batch, TM) * 4));
  ctx->t5_loaded = true;
  return HVLA_OK;
}

int hvla_t5_encode(hvla_ctx* ctx, const int64_t* input_ids, const int64_t* attention_mask, float* token_embedding, int32_t B,
                   int32_t T, void* stream) {
  if (!ctx) return HVLA_E_STATE;
  if (!ctx->t5_loaded) FAIL(ctx, HVLA_E_STATE, "hvla_t5_load has not been called");
  if (B < 1 || B > ctx->t5_max_batch || T < 1 || T > ctx->t5w.max_tokens)
    FAIL(ctx, HVLA_E_SHAPE, "batch %d / tokens %d outside [1, %d] x [1, %d]", B, T, ctx->t5_max_batch, ctx->t5w.max_tokens);
  if (!input_ids || !attention_mask || !token_embedding) FAIL(ctx, HVLA_E_SHAPE, "null pointer");
  HIPCHK(ctx, hipSetDevice(ctx->device));
  HIPCHK(ctx, t5_encode(ctx->t5d, ctx->t5w, ctx->t5_work.as<float>(), input_ids, attention_mask, token_embedding, B, T,
                        reinterpret_cast<hipStream_t>(stream)));
  return HVLA_OK;
}

int hvla_profile(hvla_ctx* ctx, int32_t mode) {
  if (!ctx) return HVLA_E_STATE;
  if (mode < 0 || mode > 2) FAIL(ctx, HVLA_E_SHAPE, "profile mode %d", mode);
  ctx->prof.mode = mode;
  return HVLA_OK;
}

int hvla_profile_select(hvla_ctx* ctx, uint32_t category_mask) {
  if (!ctx) return HVLA_E_STATE;
  if (category_mask == 0 || (category_mask >> HVLA_PROF_N) != 0) FAIL(ctx, HVLA_E_SHAPE, "profile category mask 0x%x", category_mask);
  ctx->prof.select = category_mask;
  return HVLA_OK;
}

int hvla_profile_read(hvla_ctx* ctx, float* ms, int32_t* launches) {
  if (!ctx || !ms || !launches) return HVLA_E_STATE;
  HIPCHK(ctx, hipSetDevice(ctx->device));
  for (int i = 0; i < HVLA_PROF_N; ++i) ms[i] = 0.f, launches[i] = 0;
  Profiler& p = ctx->prof;
  for (size_t i = 0; i < p.used; ++i) {
    HIPCHK(ctx, hipEventSynchronize(p.stop[i]));
    float t = 0.f;
    HIPCHK(ctx, hipEventElapsedTime(&t, p.start[i], p.stop[i]));
    ms[p.cat[i]] += t;
    launches[p.cat[i]] += 1;
  }
  p.used = 0;
  return HVLA_OK;
}

#ifdef HVLA_BENCH_HOOKS
// diagnostics (libhvla_bench.so only; not part of include/hvla.h): one encoder GEMM shape on the ctx's workspace
int hvla_debug_gemm(hvla_ctx* ctx, int M, int N, int K, int epi, int variant, int iters, float* ms) {
  if (!ctx || !ctx->loaded) return HVLA_E_STATE;
  HIPCHK(ctx, hipSetDevice(ctx->device));
  const void* A = K > ctx->g.E ? ctx->ws_g.p : ctx->ws_h.p;
  void* out = epi == 3 ? ctx->ws_x.p : (epi == 1 ? ctx->ws_qkv.p : ctx->ws_g.p);
  if (epi == 2 && K > ctx->g.E) return HVLA_E_SHAPE;
  const EncLayerW& L = ctx->encw.layer[0];
  const void* W = epi == 1 ? L.wqkv : (epi == 2 ? L.w1 : (K > ctx->g.E ? L.w2 : L.wo));
  const float* bias = epi == 1 ? L.bqkv : (epi == 2 ? L.b1 : L.b2);
  HIPCHK(ctx, debug_gemm(A, W, bias, L.ls1, out, M, N, K, epi, variant, iters, ms, nullptr));
  return HVLA_OK;
}

// diagnostics (not part of include/hvla.h): time one shape of the fine-tune path's batched GEMM on caller buffers
int hvla_debug_bgemm(const float* A, const float* B, float* C, int M, int N, int K, int ta, int tb, int nb, int accumulate,
                     int iters, float* ms) {
  const int lda = ta ? M : K, ldb = tb ? K : N;
  BG g{A, B, C, nullptr, M, N, K, lda, ldb, N, (long)M * K, 0, (long)N * K, 0, (long)M * N, 0, 0, 1, 1.f, accumulate, 1, accumulate != 0};
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) bgemm(nullptr, ta != 0, tb != 0, g, nb);
  (void)hipEventRecord(e0, nullptr);
  for (int i = 0; i < iters; ++i) bgemm(nullptr, ta != 0, tb != 0, g, nb);
  (void)hipEventRecord(e1, nullptr);
  (void)hipEventSynchronize(e1);
  (void)hipEventElapsedTime(ms, e0, e1);
  *ms /= iters;
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  return hipGetLastError() == hipSuccess ? HVLA_OK : HVLA_E_HIP;
}

// phase time stamps of attention_kernel for workgroups `wgs[i]` (8 stamps each), on the ctx's workspace (contents irrelevant)
int hvla_debug_attention_stamps(hvla_ctx* ctx, int32_t B, const int32_t* wgs, int32_t nwg, unsigned long long* out) {
  if (!ctx || !out || !wgs) return HVLA_E_STATE;
  HIPCHK(ctx, hipSetDevice(ctx->device));
  const Geom& g = ctx->g;
  DevBuf st;
  HIPCHK(ctx, st.alloc((size_t)8 * 8));
  for (int i = 0; i < nwg; ++i) {
    HIPCHK(ctx, hipMemset(st.p, 0, 64));
    for (int rep = 0; rep < 2; ++rep)
      HIPCHK(ctx, debug_attention_stamps(ctx->ws_qkv.p, ctx->ws_h.p, ctx->ws_abar.p, B, g.S(), g.E, g.enc_heads, wgs[i], st.as<unsigned long long>(), nullptr));
    HIPCHK(ctx, hipDeviceSynchronize());
    HIPCHK(ctx, hipMemcpy(out + (size_t)i * 8, st.p, 64, hipMemcpyDeviceToHost));
  }
  return HVLA_OK;
}

// phase time stamps of the policy kernel (episode 0, wave 0; shader clock): 27 values at the README geometry, policy.hip HVLA_STAMP
int hvla_debug_policy_stamps(hvla_ctx* ctx, const hvla_weights* w, const float* tokens, float* actions, float* logits, int32_t B,
                             unsigned long long* out, int32_t n) {
  if (!ctx || !w || !out) return HVLA_E_STATE;
  HIPCHK(ctx, hipSetDevice(ctx->device));
  const Geom& g = ctx->g;
  const PolicyLayout& pl = ctx->lay.pl;
  DevBuf st;
  HIPCHK(ctx, st.alloc((size_t)n * 8));
  HIPCHK(ctx, hipMemset(st.p, 0, (size_t)n * 8));
  PolicyParams p{pl, w->wh.as<__bf16>(), w->wl.as<__bf16>(), w->vf.as<float>(), tokens, actions, logits,
                 B, g.E, g.P(), g.L, g.M, g.horizon, g.action_dim, g.tanh_scale, g.max_action};
  p.stamps = st.as<unsigned long long>();
  for (int i = 0; i < 3; ++i) HIPCHK(ctx, launch_policy(p, nullptr));
  HIPCHK(ctx, hipDeviceSynchronize());
  HIPCHK(ctx, hipMemcpy(out, st.p, (size_t)n * 8, hipMemcpyDeviceToHost));
  return HVLA_OK;
}

// the fine-tune GEMM on the exact-f32 matrix instruction (bitwise fmaf chains) instead of split-bf16: a debugging aid for
// gradient comparisons, so it is a call in the bench library and not an environment switch of the product
int hvla_debug_train_gemm_exact(int on) {
  set_train_gemm_exact(on != 0);
  return HVLA_OK;
}
// shader-clock stamps of the last context-encoder launch (workgroup 0): see hypernet.hip CTX_STAMP
int hvla_debug_ctx_stamps(unsigned long long* out) { return debug_ctx_stamps(out) == hipSuccess ? HVLA_OK : HVLA_E_HIP; }
// device pointer and size of one encoder workspace buffer (tools/ read intermediates back with hipMemcpy):
// 0 x (f32 residual stream), 1 h (16-bit LayerNorm / attention output), 2 qkv, 3 g (MLP hidden), 4 corr, 5 abar, 6 ln_cnt, 7 ln_part
int hvla_debug_workspace(hvla_ctx* ctx, int which, void** ptr, size_t* bytes) {
  if (!ctx || !ptr || !bytes) return HVLA_E_STATE;
  DevBuf* b[8] = {&ctx->ws_x, &ctx->ws_h, &ctx->ws_qkv, &ctx->ws_g, &ctx->ws_corr, &ctx->ws_abar, &ctx->ws_lncnt, &ctx->ws_lnpart};
  if (which < 0 || which > 7) return HVLA_E_SHAPE;
  *ptr = b[which]->p;
  *bytes = b[which]->bytes;
  return HVLA_OK;
}
int hvla_debug_lnx_stats(hvla_ctx* ctx, unsigned long long* out, int reset) {
  if (!ctx) return HVLA_E_STATE;
  HIPCHK(ctx, hipSetDevice(ctx->device));
  HIPCHK(ctx, debug_lnx_stats(out, reset));
  return HVLA_OK;
}
// hvla_encode then returns behind its n-th dense product (QKV, out, fc1, fc2 of layer 0 = 1 .. 4, ...; 0 = the whole encoder): the
// workspace is read back (hvla_debug_workspace) as that product's consumers would find it -- tests of the mean rows and the corr table
int hvla_debug_encode_stop(hvla_ctx* ctx, int32_t n) {
  if (!ctx || n < 0) return HVLA_E_STATE;
  ctx->enc_stop = n;
  return HVLA_OK;
}
// how long a column tile of a residual GEMM waits for the image's other tiles (ticks of 10 ns; the product's value is 800).  0 = nobody
// waits: every tile but an image's last arriver is normalised from memory -- the test that the two routes give the same bytes
int hvla_debug_lnx_spin(hvla_ctx* ctx, uint32_t ticks) {
  if (!ctx) return HVLA_E_STATE;
  ctx->ln_spin = ticks;
  return HVLA_OK;
}
#endif  // HVLA_BENCH_HOOKS

int64_t hvla_launches(hvla_ctx* ctx) {
  if (!ctx) return 0;
  const int64_t n = (int64_t)ctx->prof.nlaunch;
  ctx->prof.nlaunch = 0;
  return n;
}

int hvla_box_probe(hvla_ctx* ctx, float out[3], void* stream) {
  if (!ctx || !out) return HVLA_E_STATE;
  HIPCHK(ctx, hipSetDevice(ctx->device));
  DevBuf sink, ticks;
  int ncu = 256;
  (void)hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, ctx->device);
  if (sink.alloc((size_t)ncu * 512 * sizeof(float)) != hipSuccess || ticks.alloc(64) != hipSuccess) FAIL(ctx, HVLA_E_ARENA_FULL, "box probe buffers");
  HIPCHK(ctx, run_box_probe(sink.as<float>(), ticks.as<unsigned long long>(), out, reinterpret_cast<hipStream_t>(stream)));
  return HVLA_OK;
}

int hvla_selftest(hvla_ctx* ctx, void* stream) {
  if (!ctx) return HVLA_E_STATE;
  HIPCHK(ctx, hipSetDevice(ctx->device));
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  HIPCHK(ctx, hipMemsetAsync(ctx->flags.p, 0, 64 * sizeof(int), st));
  HIPCHK(ctx, launch_selftest(ctx->flags.as<int>(), st));
  int h[64];
  HIPCHK(ctx, hipMemcpyAsync(h, ctx->flags.p, sizeof h, hipMemcpyDeviceToHost, st));
  HIPCHK(ctx, hipStreamSynchronize(st));
  for (int i = 0; i < 8; ++i)
    if (h[i]) FAIL(ctx, HVLA_E_STATE, "MFMA layout probe %d failed (%d mismatches)", i, h[i]);
  return HVLA_OK;
}

}  // extern "C"
