// train.h — fine-tune step structures (train.hip)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "layout.h"

namespace hvla {

// flat layout of the trainable hypernetwork parameters (float32 elements)
struct TrainLayout {
  long w_tok, b_tok, w_img, b_img, pos_tok, pos_img, pos_layer, norm_s, norm_b, wcat, bcat, total, G;
  struct CL { long ln0_s, ln0_b, ln1_s, ln1_b, wq, bq, wk, bk, wv, bv, wo, bo, w1, b1, w2, b2; } layer[8];
};
TrainLayout make_train_layout(const Geom& g);
size_t train_workspace_floats(const Geom& g, int B);

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
  const uint8_t* wd_mask;    // [G] 1 where the generated leaf is a base-net kernel (weight_decay_strategy v5)
};
struct TrainInputs {
  const float* tok;          // [B, T, lang_dim]
  const int64_t* attn_mask;  // [B, T]
  const float* cls;          // [B, E]
  const float* tokens;       // [B, P, E]  frozen-encoder patch tokens (hvla_encode)
  const float* target;       // [B, horizon, action_dim]
  const uint8_t* tmask;      // [B]
  const uint8_t* amask;      // [B, horizon, action_dim]
};
struct TrainHyper {
  float lr, b1, b2, eps, weight_decay, clip, ema_decay;
  int step, forward_only;
};
hipError_t train_step(const Geom& g, const TrainLayout& L, const TrainBuffers& tb, const TrainInputs& in, int B,
                      const TrainHyper& hp, hipStream_t st);
hipError_t train_apply(const TrainLayout& L, const TrainBuffers& tb, const TrainHyper& hp, hipStream_t st);

}  // namespace hvla
