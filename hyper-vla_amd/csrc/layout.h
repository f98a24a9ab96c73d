// layout.h — how one episode's generated parameters are laid out in the device weight arena, and
// the map back to the reference's pytree order (hypervla/model.py:370-515).
//
// The hypernetwork's 73 output heads are one GEMM  theta[B, G] = ctx[B, C] @ W_cat[C, G] + b_cat
// (share_layer_index=True => every head reads context token 0, hypernetwork.py:205-217).  Because
// the columns of W_cat can be permuted freely at load time, the GEMM writes each parameter straight
// to the place the policy kernel reads it from:
//
//   matrix region  Wh/Wl[B][Gm]  bf16 hi / lo planes (x = hi + lo), in MFMA A-fragment order:
//                  fragment f = 64 lanes x 8 elements; lane (row = l & 31, half = l >> 5) holds
//                  W[k(step, half, j)][n = 32 * mtile + row], j = 0..7 (the K order of each GEMM is the
//                  order in which the kernel produces the matching B fragments, see policy.hip)
//   vector region  Vf[B][Gv]     f32 biases / LayerNorm / position embedding, in accumulator
//                  ("C-layout") order: [tile][half][r] <-> feature 32 * tile + crow(r, half)
//
// perm[pos] is the reference flat index (leaf offset + C-order index in the flax leaf) of packed
// position pos in [0, Gm + Gv), or -1 for padding.
#pragma once
#include <stdint.h>

#include <string>
#include <vector>

namespace hvla {

struct Geom {
  int image_size, patch, E, enc_layers, enc_heads, enc_mlp;
  int D, L, H, M, horizon, action_dim;
  float tanh_scale, max_action;
  int C, ctx_layers, ctx_heads, ctx_mlp, T, lang_dim, scale_context;
  int clip_target = 1;       // MixActionHead.loss: clip the action target to +-max_action (action_heads.py:499-500)
  int grid() const { return image_size / patch; }
  int P() const { return grid() * grid(); }
  int S() const { return P() + 1; }
  int hd() const { return D / H; }
  int A() const { return horizon * (action_dim - 1); }
};

// one generated leaf in reference (jax pytree) order
struct LeafInfo {
  std::string flat;          // "encoder_Transformer_0_encoderblock_0_LayerNorm_0_bias"
  std::vector<int> shape;
  int64_t offset, size;
};

inline std::vector<LeafInfo> generated_leaves(const Geom& g) {
  std::vector<LeafInfo> v;
  int64_t off = 0;
  auto add = [&](const std::string& name, std::vector<int> shape) {
    int64_t n = 1;
    for (int s : shape) n *= s;
    v.push_back({name, shape, off, n});
    off += n;
  };
  const int D = g.D, M = g.M, H = g.H, hd = g.hd();
  add("action_head_continuous_head_bias", {g.A()});
  add("action_head_continuous_head_kernel", {D, g.A()});
  add("action_head_discrete_head_bias", {g.horizon});
  add("action_head_discrete_head_kernel", {D, g.horizon});
  const std::string T = "encoder_Transformer_0_";
  add(T + "encoder_norm_bias", {D});
  add(T + "encoder_norm_scale", {D});
  for (int l = 0; l < g.L; ++l) {
    const std::string B = T + "encoderblock_" + std::to_string(l) + "_";
    add(B + "LayerNorm_0_bias", {D});
    add(B + "LayerNorm_0_scale", {D});
    add(B + "LayerNorm_1_bias", {D});
    add(B + "LayerNorm_1_scale", {D});
    add(B + "MlpBlock_0_Dense_0_bias", {M});
    add(B + "MlpBlock_0_Dense_0_kernel", {D, M});
    add(B + "MlpBlock_0_Dense_1_bias", {D});
    add(B + "MlpBlock_0_Dense_1_kernel", {M, D});
    const std::string A = B + "MultiHeadDotProductAttention_0_";
    add(A + "key_bias", {H, hd});
    add(A + "key_kernel", {D, H, hd});
    add(A + "out_bias", {D});
    add(A + "out_kernel", {H, hd, D});
    add(A + "query_bias", {H, hd});
    add(A + "query_kernel", {D, H, hd});
    add(A + "value_bias", {H, hd});
    add(A + "value_kernel", {D, H, hd});
  }
  add("encoder_image_embedding_projection_bias", {D});
  add("encoder_image_embedding_projection_kernel", {g.E, D});
  add("encoder_pos_embedding", {1, g.S(), D});
  return v;
}

// Offsets the policy kernel uses (all in elements; frag offsets are multiples of 512).
struct PolicyLayout {
  int Gm, Gv;              // padded sizes of the matrix / vector regions (per episode)
  int G;                   // reference parameter count
  // matrix region (bf16 elements)
  int m_proj;              // [D/32 mtile][E/16 kstep] frags
  int m_layer0, m_layer_stride;   // per layer: qkv, out, fc1, fc2
  int m_qkv, m_out, m_fc1, m_fc2;  // offsets inside a layer
  int m_head;
  // vector region (f32 elements)
  int v_proj_bias, v_pos, v_layer0, v_layer_stride;
  int v_ln0_s, v_ln0_b, v_qkv_b, v_out_b, v_ln1_s, v_ln1_b, v_fc1_b, v_fc2_b;  // inside a layer
  int v_norm_s, v_norm_b, v_head_b;
};

inline int crow_h(int r, int half) { return (r & 3) + 8 * (r >> 2) + 4 * half; }

// K order of B fragments made from 32-row accumulator tiles (guide §3 "accumulator tile as the next
// MFMA's operand"): element j of lane-half h in k-step ks is feature
//   32 * (ks >> 1) + 16 * (ks & 1) + 8 * (j >> 2) + 4 * h + (j & 3)
inline int kphi(int ks, int half, int j) {
  return 32 * (ks >> 1) + 16 * (ks & 1) + 8 * (j >> 2) + 4 * half + (j & 3);
}
// K order of the image-projection GEMM: each lane loads 16 consecutive floats of its token row per
// pair of k-steps:  32 * (ks >> 1) + 16 * h + 8 * (ks & 1) + j
inline int kproj(int ks, int half, int j) { return 32 * (ks >> 1) + 16 * half + 8 * (ks & 1) + j; }

struct PackedLayout {
  PolicyLayout pl;
  std::vector<int32_t> perm;   // [Gm + Gv] -> reference flat index or -1
};

inline PackedLayout build_layout(const Geom& g) {
  PackedLayout out;
  PolicyLayout& p = out.pl;
  const int D = g.D, M = g.M, H = g.H, hd = g.hd(), E = g.E, S = g.S();
  const int TD = D / 32, TM = M / 32;
  auto leaves = generated_leaves(g);
  auto find = [&](const std::string& n) -> const LeafInfo& {
    for (auto& l : leaves)
      if (l.flat == n) return l;
    static LeafInfo none;
    return none;
  };
  p.G = (int)(leaves.back().offset + leaves.back().size);
  // ---- matrix region offsets
  int m = 0;
  p.m_proj = m;  m += TD * (E / 16) * 512;
  p.m_qkv = 0;
  p.m_out = p.m_qkv + (3 * D / 32) * (D / 16) * 512;
  p.m_fc1 = p.m_out + H * TD * 512;
  p.m_fc2 = p.m_fc1 + TM * (D / 16) * 512;
  p.m_layer_stride = p.m_fc2 + TM * TD * 2 * 512;
  p.m_layer0 = m;  m += g.L * p.m_layer_stride;
  p.m_head = m;  m += (D / 16) * 512;
  p.Gm = m;
  // ---- vector region offsets
  int v = 0;
  p.v_proj_bias = v;  v += D;
  p.v_pos = v;  v += S * D;
  p.v_ln0_s = 0;  p.v_ln0_b = D;  p.v_qkv_b = 2 * D;  p.v_out_b = 5 * D;
  p.v_ln1_s = 6 * D;  p.v_ln1_b = 7 * D;  p.v_fc1_b = 8 * D;  p.v_fc2_b = 8 * D + M;
  p.v_layer_stride = 9 * D + M;
  p.v_layer0 = v;  v += g.L * p.v_layer_stride;
  p.v_norm_s = v;  v += D;
  p.v_norm_b = v;  v += D;
  p.v_head_b = v;  v += 32;
  p.Gv = (v + 31) / 32 * 32;

  out.perm.assign((size_t)p.Gm + p.Gv, -1);
  int32_t* pm = out.perm.data();
  int32_t* pv = out.perm.data() + p.Gm;

  // A-fragment writer: W^T rows n (32 per mtile), k given by korder(ks, half, j); ref(k, n) gives the
  // reference flat index or -1.
  auto put_frag = [&](int base, int frag, int mtile, int ks, int (*korder)(int, int, int), int kbase,
                      auto ref) {
    for (int lane = 0; lane < 64; ++lane)
      for (int j = 0; j < 8; ++j) {
        int n = 32 * mtile + (lane & 31);
        int k = kbase + korder(ks, lane >> 5, j);
        pm[base + frag * 512 + lane * 8 + j] = ref(k, n);
      }
  };
  auto cvec = [&](int base, int ntiles, auto ref) {   // C-layout vector [tile][half][r]
    for (int t = 0; t < ntiles; ++t)
      for (int h = 0; h < 2; ++h)
        for (int r = 0; r < 16; ++r) pv[base + (t * 2 + h) * 16 + r] = ref(32 * t + crow_h(r, h));
  };

  {  // image_embedding_projection: kernel [E, D]
    const LeafInfo& K = find("encoder_image_embedding_projection_kernel");
    const LeafInfo& Bv = find("encoder_image_embedding_projection_bias");
    int f = 0;
    for (int mt = 0; mt < TD; ++mt)
      for (int ks = 0; ks < E / 16; ++ks, ++f)
        put_frag(p.m_proj, f, mt, ks, kproj, 0, [&](int k, int n) { return (int32_t)(K.offset + (int64_t)k * D + n); });
    cvec(p.v_proj_bias, TD, [&](int n) { return (int32_t)(Bv.offset + n); });
    const LeafInfo& Pe = find("encoder_pos_embedding");
    for (int t = 0; t < S; ++t)
      cvec(p.v_pos + t * D, TD, [&](int n) { return (int32_t)(Pe.offset + (int64_t)t * D + n); });
  }
  for (int l = 0; l < g.L; ++l) {
    const std::string Bn = "encoder_Transformer_0_encoderblock_" + std::to_string(l) + "_";
    const std::string An = Bn + "MultiHeadDotProductAttention_0_";
    const int mb = p.m_layer0 + l * p.m_layer_stride, vb = p.v_layer0 + l * p.v_layer_stride;
    // fused QKV: rows n in [0,D) = query (head-major h*hd+d == flax [D,H,hd] flattened), [D,2D) key,
    // [2D,3D) value
    const LeafInfo* W[3] = {&find(An + "query_kernel"), &find(An + "key_kernel"), &find(An + "value_kernel")};
    const LeafInfo* Bq[3] = {&find(An + "query_bias"), &find(An + "key_bias"), &find(An + "value_bias")};
    int f = 0;
    for (int mt = 0; mt < 3 * D / 32; ++mt)
      for (int ks = 0; ks < D / 16; ++ks, ++f)
        put_frag(mb + p.m_qkv, f, mt, ks, kphi, 0, [&](int k, int n) {
          return (int32_t)(W[n / D]->offset + (int64_t)k * D + (n % D));
        });
    cvec(vb + p.v_qkv_b, 3 * D / 32, [&](int n) { return (int32_t)(Bq[n / D]->offset + (n % D)); });
    // out projection: flax kernel [H, hd, D]; per head one k-step of hd = 16, K order kphi(ks=0)
    const LeafInfo& Wo = find(An + "out_kernel");
    f = 0;
    for (int hh = 0; hh < H; ++hh)
      for (int mt = 0; mt < TD; ++mt, ++f)
        put_frag(mb + p.m_out, f, mt, 0, kphi, 0, [&](int k, int n) {
          return (int32_t)(Wo.offset + ((int64_t)hh * hd + k) * D + n);
        });
    cvec(vb + p.v_out_b, TD, [&](int n) { return (int32_t)(find(An + "out_bias").offset + n); });
    cvec(vb + p.v_ln0_s, TD, [&](int n) { return (int32_t)(find(Bn + "LayerNorm_0_scale").offset + n); });
    cvec(vb + p.v_ln0_b, TD, [&](int n) { return (int32_t)(find(Bn + "LayerNorm_0_bias").offset + n); });
    cvec(vb + p.v_ln1_s, TD, [&](int n) { return (int32_t)(find(Bn + "LayerNorm_1_scale").offset + n); });
    cvec(vb + p.v_ln1_b, TD, [&](int n) { return (int32_t)(find(Bn + "LayerNorm_1_bias").offset + n); });
    // fc1: kernel [D, M]
    const LeafInfo& W1 = find(Bn + "MlpBlock_0_Dense_0_kernel");
    f = 0;
    for (int mt = 0; mt < TM; ++mt)
      for (int ks = 0; ks < D / 16; ++ks, ++f)
        put_frag(mb + p.m_fc1, f, mt, ks, kphi, 0, [&](int k, int n) { return (int32_t)(W1.offset + (int64_t)k * M + n); });
    cvec(vb + p.v_fc1_b, TM, [&](int n) { return (int32_t)(find(Bn + "MlpBlock_0_Dense_0_bias").offset + n); });
    // fc2: kernel [M, D]; consumed per hidden tile t: [t][mt][s]
    const LeafInfo& W2 = find(Bn + "MlpBlock_0_Dense_1_kernel");
    f = 0;
    for (int t = 0; t < TM; ++t)
      for (int mt = 0; mt < TD; ++mt)
        for (int s = 0; s < 2; ++s, ++f)
          put_frag(mb + p.m_fc2, f, mt, s, kphi, 32 * t, [&](int k, int n) { return (int32_t)(W2.offset + (int64_t)k * D + n); });
    cvec(vb + p.v_fc2_b, TD, [&](int n) { return (int32_t)(find(Bn + "MlpBlock_0_Dense_1_bias").offset + n); });
  }
  cvec(p.v_norm_s, TD, [&](int n) { return (int32_t)(find("encoder_Transformer_0_encoder_norm_scale").offset + n); });
  cvec(p.v_norm_b, TD, [&](int n) { return (int32_t)(find("encoder_Transformer_0_encoder_norm_bias").offset + n); });
  {  // mix head: rows [0, A) continuous_head, [A, A + horizon) discrete_head, rest padding
    const LeafInfo& Wc = find("action_head_continuous_head_kernel");
    const LeafInfo& Wd = find("action_head_discrete_head_kernel");
    const LeafInfo& Bc = find("action_head_continuous_head_bias");
    const LeafInfo& Bd = find("action_head_discrete_head_bias");
    const int A = g.A(), Hz = g.horizon;
    for (int ks = 0; ks < D / 16; ++ks)
      put_frag(p.m_head, ks, 0, ks, kphi, 0, [&](int k, int n) -> int32_t {
        if (n < A) return (int32_t)(Wc.offset + (int64_t)k * A + n);
        if (n < A + Hz) return (int32_t)(Wd.offset + (int64_t)k * Hz + (n - A));
        return -1;
      });
    cvec(p.v_head_b, 1, [&](int n) -> int32_t {
      if (n < A) return (int32_t)(Bc.offset + n);
      if (n < A + Hz) return (int32_t)(Bd.offset + (n - A));
      return -1;
    });
  }
  return out;
}

}  // namespace hvla
