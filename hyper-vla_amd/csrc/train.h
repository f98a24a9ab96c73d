// train.h — fine-tune step structures (train.hip)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "layout.h"

namespace hvla {

// generic batched f32 GEMM on the matrix cores (split-bf16, f32-class accuracy; train.hip):
//   C[b0,b1] (+)= alpha * op(A)[b0,b1] * op(B)[b0,b1] (+ bias[b0][n]),  batch = blockIdx.z = b0 * nb1 + b1
struct BG {
  const float* A;
  const float* B;
  float* C;
  const float* bias;          // nullable, indexed [n], batch stride sBias0
  int M, N, K, lda, ldb, ldc;
  long sA0, sA1, sB0, sB1, sC0, sC1, sBias0;
  int nb1;                    // batch = blockIdx.z = b0 * nb1 + b1
  float alpha;
  int accumulate;             // 0 store, 1 C += (one writer per element), 2 atomic C += (batches share C)
  int ksplit = 1;             // > 1: K is cut into ksplit chunks over blockIdx.z, reduced with atomics (accumulate != 0)
  int allow_split = 0;        // 1: the launcher may choose ksplit > 1 itself (deep-K gradient products; result then
                              //    depends on the order of the atomic adds in its last bits)
  int a_padded = 0;           // 1: every row of A is followed by zeros up to a multiple of 4 elements inside lda (the attention
                              //    matrices, row stride S rounded up): float4 staging may run over the row's end
  long sBias1 = 0;            // bias stride of the inner batch index b1
};
void bgemm(hipStream_t st, bool ta, bool tb, BG g, int nb0);
void train_gemm_timer(bool on, bool first_use_by_this_context);                  // hvla_train_profile (the current device's timer; counts its users)
void train_gemm_timer_release();                                                 // hvla_destroy of a context that used it: the events go with the last user
hipError_t train_gemm_timer_read(float* ms, double* flops, int* launches);      // since the last read

// flat layout of the trainable hypernetwork parameters (float32 elements)
// `total` = the hypernetwork's own parameters; the shared DINOv2 leaves follow at [total, total + enc_total) when the
// image encoder is trained too (`fine_tune_pretrained_image_encoder=True`), in hypervla.config.encoder_leaves order.
struct TrainLayout {
  long w_tok, b_tok, w_img, b_img, pos_tok, pos_img, pos_layer, norm_s, norm_b, wcat, bcat, total, G;
  struct CL { long ln0_s, ln0_b, ln1_s, ln1_b, wq, bq, wk, bk, wv, bv, wo, bo, w1, b1, w2, b2; } layer[8];
  long e_cls, e_mask, e_pb, e_pk, e_pos, e_lnb, e_lns, enc_total;
  struct EL { long kb, kk, qb, qk, vb, vk, ob, ok, ls1, ls2, f1b, f1k, f2b, f2k, n1b, n1s, n2b, n2s; } enc[24];
};
TrainLayout make_train_layout(const Geom& g);
#ifdef HVLA_BENCH_HOOKS
void set_train_gemm_exact(bool on);
#endif
size_t train_workspace_floats(const Geom& g, int B, bool train_encoder);

struct TrainBuffers {        // all device memory, owned by the caller
  float* params;             // [total]
  float* grads;              // [total]
  __bf16* mu;                // [total]  AdamW first moment (optax mu_dtype = bfloat16)
  float* nu;                 // [total]
  float* ema;                // [total] or null
  float* theta;              // [B, G]
  float* dtheta;             // [B, G]
  float* work;               // [train_workspace_floats]
  float* loss;               // [B]
  float* actions;            // [B, horizon, action_dim] or null
  float* logits;             // [B, horizon] or null
  float* sqsum;              // [1]
  const uint8_t* wd_mask;    // [total (+ enc_total)] 1 where decoupled weight decay applies: built by the host from the
                             //     selected weight_decay_strategy (hypervla/train.py), flat parameter order
  const float* params0;      // [enc_total] pretrained encoder weights for the delta decay (train.py:465-471) or null
};
struct TrainInputs {
  const float* tok;          // [B, T, lang_dim]
  const int64_t* attn_mask;  // [B, T]
  const float* cls;          // [B, E]
  const float* tokens;       // [B, P, E]  frozen-encoder patch tokens (hvla_encode); null when `images` is given
  const uint8_t* images;     // [B, H, W, 3] -> the encoder runs (and is differentiated) inside the step
  const float* target;       // [B, horizon, action_dim]
  const uint8_t* tmask;      // [B]
  const uint8_t* amask;      // [B, horizon, action_dim]
};
struct TrainHyper {
  float lr, b1, b2, eps, weight_decay, clip, ema_decay;
  int step, forward_only;
  float base_lr, base_weight_decay;      // optimizer group of the shared (DINOv2) leaves, train_utils.py:411-419
};
// bucket_done (nullable, [3]): events recorded on `st` when a contiguous range of `grads` is final -- [0] the shared DINOv2
// leaves [total, total + enc_total) after the image encoder's backward (trained encoder only), [1] the output heads
// [wcat, total) after the weight-generation backward, [2] the context encoder [0, wcat) at the end -- so that the caller's
// all-reduce of a bucket runs under the rest of the backward pass.
hipError_t train_step(const Geom& g, const TrainLayout& L, const TrainBuffers& tb, const TrainInputs& in, int B,
                      const TrainHyper& hp, hipStream_t st, hipEvent_t* bucket_done = nullptr);
hipError_t train_apply(const TrainLayout& L, const TrainBuffers& tb, const TrainHyper& hp, bool train_encoder, hipStream_t st);
hipError_t train_accumulate(const TrainLayout& L, const TrainBuffers& tb, float* acc, float inv_k, const TrainHyper& hp,
                            bool train_encoder, hipStream_t st);

}  // namespace hvla
