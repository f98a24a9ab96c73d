/*
 * hvla.h — C ABI of libhvla: the MI355X-native HyperVLA action-prediction path.
 *
 * Drop-in boundary.  Each entry point names the reference interface it replaces
 * (paths relative to the reference repository).  The reference is pure Python/JAX and has no FFI
 * of its own; a maintainer binds this library with ctypes from hypervla/model.py (the stub is in
 * INTEGRATION.md; the binding this repo ships is hyper-vla_amd/hypervla/_native.py).
 *
 * Conventions
 *   - plain C: pointers + sizes, no torch / HIP types in signatures (`stream` is a hipStream_t
 *     passed as void*; NULL = the device's default stream).
 *   - every data pointer of generate/encode/policy/step/export is a DEVICE pointer owned by the
 *     caller; hvla_load_weights takes HOST pointers.
 *   - every call returns HVLA_OK (0) or a negative HVLA_E_* code; no C++ exception crosses the
 *     boundary; hvla_last_error(ctx) gives the message of the last failing call on that ctx.
 *   - a ctx is bound to one device; calls on one ctx are not re-entrant; different ctxs are
 *     independent.  After hvla_load_weights, encode/policy/step do not allocate, do not
 *     synchronise and launch only on `stream`, so a step can be captured into a hipGraph.
 *     hvla_generate allocates its weight arena only when the ctx holds no arena of that batch size
 *     handed back by hvla_weights_free (the first episode batch of a size; afterwards it only
 *     launches on `stream`); hvla_weights_free waits for the device, as hipFree would.
 */
#ifndef HVLA_H_
#define HVLA_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define HVLA_OK 0
#define HVLA_E_SHAPE (-1)      /* batch / geometry outside what the ctx was created for        */
#define HVLA_E_DTYPE (-2)      /* unsupported operand type selector                            */
#define HVLA_E_DEVICE (-3)     /* bad device ordinal, or not a gfx950 device                   */
#define HVLA_E_ARENA_FULL (-4) /* device allocation for a weight arena / workspace failed      */
#define HVLA_E_HIP (-5)        /* a HIP runtime call failed (message has the hipError string)  */
#define HVLA_E_WEIGHTS (-6)    /* missing / mis-sized / unknown named tensor                   */
#define HVLA_E_STATE (-7)      /* call order violated (e.g. step before load_weights)          */

#define HVLA_ENC_F16 0  /* encoder MFMA operands fp16 (default: meets the 1e-3 action tolerance) */
#define HVLA_ENC_BF16 1 /* encoder MFMA operands bf16 (same rate, ~7x larger action error)       */

typedef struct hvla_ctx hvla_ctx;
typedef struct hvla_weights hvla_weights; /* per-batch generated-weight arena (opaque)          */

/* Geometry of the path == the values the reference reads from config.json
 * (hypervla/model.py:152-163,197-200; README.md:33-61).                                        */
typedef struct hvla_config {
  uint32_t struct_size;                      /* = sizeof(hvla_config) of the header the caller was built against: hvla_create
                                                returns HVLA_E_SHAPE when it is not this library's (the struct is passed by pointer
                                                and grows at its end; a binder built against another header is refused, never
                                                read past its end)                                                      */
  int32_t image_size, patch;                 /* 224, 14                                         */
  int32_t enc_dim, enc_layers, enc_heads, enc_mlp; /* DINOv2-base: 768, 12, 12, 3072            */
  int32_t dim, layers, heads, mlp;           /* generated vit_t: 64, 4, 4, 128                  */
  int32_t horizon, action_dim;               /* 4, 7                                            */
  float tanh_scale, max_action;              /* 5, 5 (action_heads.py:469-470)                  */
  int32_t ctx_dim, ctx_layers, ctx_heads, ctx_mlp; /* hypernet: 128, 6, 4, 512.  ctx_mlp % 16 == 0 (the f32 MFMA's column
                                                tiles), and the context encoder's LDS working set -- (lang_tokens + 2) rows of
                                                max(3 ctx_dim, ctx_mlp) + 2 ctx_dim floats -- must fit 160 KiB: hvla_create
                                                returns HVLA_E_SHAPE otherwise                                          */
  int32_t lang_tokens, lang_dim;             /* 32, 768                                         */
  int32_t scale_context;                     /* hypernetwork.py:191-192                         */
  int32_t max_batch;                         /* workspace is sized for this many episodes       */
  int32_t enc_dtype;                         /* HVLA_ENC_F16 | HVLA_ENC_BF16                    */
  int32_t streams;                           /* 1 (default, 0 = 1) or 2: hvla_step runs the two halves of a batch of
                                                >= 64 episodes on two streams, forked from / joined to the caller's  */
  int32_t clip_target;                       /* action_head_kwargs.clip_target (action_heads.py:408,499-500): the loss clips
                                                the action target to +-max_action iff non-zero                        */
} hvla_config;

/* One named float32 tensor of the hypernetwork checkpoint, HOST memory, reference naming
 * (SURVEY.md §5.4): "task_token_projection/kernel", "Transformer_0/encoderblock_3/LayerNorm_0/scale",
 * "output_head_<flat leaf>/kernel", flat shared vectors "encoder_image_encoder_<path>", ...      */
typedef struct hvla_tensor_desc {
  const char* name;
  const float* data;
  int64_t numel;
} hvla_tensor_desc;

/* Replaces: HyperVLA.load_pretrained's model construction (hypervla/model.py:197-208).          */
int hvla_create(const hvla_config* cfg, int device, hvla_ctx** out);
void hvla_destroy(hvla_ctx* ctx);
const char* hvla_last_error(const hvla_ctx* ctx);

/* Replaces: orbax restore of HN params + `.replace(params=EMA)` (hypervla/model.py:210-214,
 * data/simpler/evaluate.py:438-444).  Packs every tensor into its device layout.
 * Must be given every tensor of the checkpoint in one call; may be called again to swap weights. */
int hvla_load_weights(hvla_ctx* ctx, const hvla_tensor_desc* tensors, int32_t n);

/* Number of generated parameters per episode (G = 201500 for the README geometry).              */
int64_t hvla_num_generated(const hvla_ctx* ctx);

/* Replaces: HyperVLA.create_tasks -> HyperNetwork.__call__ (hypervla/model.py:35-83,
 * hypervla/components/hypernetwork.py:99-233) for B episodes at once.
 *   token_embedding f32 [B, lang_tokens, lang_dim]; attention_mask i64 [B, lang_tokens];
 *   initial_cls f32 [B, enc_dim]  (= initial_state["patch_embeddings"][:, 0]).
 * On success *out owns a device arena with B episodes' policy weights.                          */
int hvla_generate(hvla_ctx* ctx, const float* token_embedding, const int64_t* attention_mask,
                  const float* initial_cls, int32_t B, hvla_weights** out, void* stream);
int hvla_weights_free(hvla_ctx* ctx, hvla_weights* w);
/* hvla_weights_free parks up to 4 arenas per ctx for the next hvla_generate of the same batch size (0.8 MB per episode at the
 * README geometry: ~0.8 GB stay resident after batches of 256).  This gives them back to the device (waits for the device).
 * The pool is locked: hvla_weights_free / hvla_release_pooled_arenas may be called from another thread than hvla_generate
 * (a garbage collector), everything else on a ctx stays single-threaded.                                                   */
int hvla_release_pooled_arenas(hvla_ctx* ctx);
int32_t hvla_weights_batch(const hvla_weights* w);

/* Reference-order views for the caller / parity tests (the reference returns these from
 * create_tasks as `base_params` and the context embedding).  theta f32 [B, G] in pytree leaf
 * order; context f32 [B, ctx_dim].  Either pointer may be NULL.          */
int hvla_weights_export(hvla_ctx* ctx, const hvla_weights* w, float* theta, float* context,
                        void* stream);

/* Replaces: ViT.__call__'s DINOv2 branch up to `last_hidden_state[:, 1:]`
 * (hypervla/components/base_vit.py:109-122).  images u8 [B, H, W, 3] -> tokens f32 [B, P, E].   */
int hvla_encode(hvla_ctx* ctx, const uint8_t* images, float* tokens, int32_t B, void* stream);

/* Replaces: the evaluators' `DINO_encode_image(initial_image).last_hidden_state`, the source of
 * initial_state["patch_embeddings"] whose CLS row conditions the hypernetwork
 * (data/simpler/evaluate.py:155-163,264-274; data/libero/evaluate.py:175; scripts/train.py:417-419), run with the
 * DINOv2 weights loaded into this ctx.  images u8 [B, H, W, 3] -> hidden f32 [B, 1 + P, E] (row 0 = CLS).   */
int hvla_encode_hidden(hvla_ctx* ctx, const uint8_t* images, float* hidden, int32_t B, void* stream);

/* Replaces: the rest of BaseNetwork.predict_action with per-episode weights
 * (base_vit.py:130-227, transformer.py:127-262, action_heads.py:431-472,524-538).
 *   tokens f32 [B, P, E] -> actions f32 [B, horizon, action_dim]; gripper_logits f32 [B, horizon]
 *   (nullable; the threshold `logit >= 0` is what column action_dim-1 of `actions` holds).     */
int hvla_policy(hvla_ctx* ctx, const hvla_weights* w, const float* tokens, float* actions,
                float* gripper_logits, int32_t B, void* stream);

/* Replaces: the `intermediates` that sample_actions returns next to the actions (hypervla/model.py:126-137: every attention
 * map, sown at hypervla/components/base_vit.py:117-118 and by flax's attention modules), as far as a caller of the path
 * reads them: InferenceWrapper(save_attention_map=True) keeps exactly two slices
 * (data/utils/hypervla_interface.py:208-217), which the evaluators pickle (data/simpler/evaluate.py:358-378):
 *   dino_cls_attention f32 [B, enc_layers, enc_heads, P]  DINOv2: attention of the CLS query over the P patch keys
 *                                                          (`DINO_attention_map[0][layer][b, :, 0, 1:]`)
 *   head_attention     f32 [B, layers, heads, P]          generated policy: attention of the action token over the P patch
 *                                                          keys (`attention_weights[0][b, :, -1, :-1]`)
 * Opt-in: the DEVICE buffers registered here (either may be NULL) are written by every following hvla_encode /
 * hvla_policy / hvla_step on this ctx until the call is repeated with NULLs; rows are the episodes of that call.   */
int hvla_set_attention_outputs(hvla_ctx* ctx, float* dino_cls_attention, float* head_attention);

/* Replaces: HyperVLA.sample_actions (hypervla/model.py:85-137): hvla_encode + hvla_policy through
 * the ctx's own token workspace.  B must equal hvla_weights_batch(w).                           */
int hvla_step(hvla_ctx* ctx, const hvla_weights* w, const uint8_t* images, float* actions,
              float* gripper_logits, int32_t B, void* stream);

/* Replaces: the un-normalise + temporal ensemble of InferenceWrapper.step for a batch of
 * episodes (data/utils/hypervla_interface.py:219-253, data/utils/action_ensemble.py:15-27,
 * temperature 0).  Keeps a device ring of the last `horizon` predictions inside `w`.
 *   actions f32 [B, horizon, action_dim] (as written by hvla_policy/step);
 *   mean/std f32 [action_dim], mask u8 [action_dim] (device); out f32 [B, action_dim].
 * The un-normalisation is the affine map a * std + mean on the masked columns.  NormalizationType.NORMAL (:219-230) passes
 * the dataset's mean / std; NormalizationType.BOUNDS (:231-242), (a + 1) (p99 - p01 + 1e-8) / 2 + p01, is the same map with
 * std = (p99 - p01 + 1e-8) / 2 and mean = p01 + std (hypervla.interface.device_unnormalization builds either pair).  */
int hvla_ensemble_reset(hvla_ctx* ctx, hvla_weights* w, void* stream);
int hvla_ensemble(hvla_ctx* ctx, hvla_weights* w, const float* actions, const float* mean,
                  const float* std, const uint8_t* mask, float* out, void* stream);

/* Replaces: MixActionHead.loss evaluated per sample (vmap of sample_loss_fn) on the policy's outputs
 * (hypervla/components/action_heads.py:474-522, scripts/train.py:326-346): the forward half of the
 * fine-tune step.  actions / logits as written by hvla_policy / hvla_step; target f32 [B, horizon,
 * action_dim] (clipped to +-max_action inside iff hvla_config.clip_target); timestep_mask u8 [B];
 * action_mask u8 [B, horizon, action_dim]; loss f32 [B] = 6 * masked-MSE + masked sigmoid-BCE.     */
int hvla_loss(hvla_ctx* ctx, const float* actions, const float* gripper_logits, const float* target,
              const uint8_t* timestep_mask, const uint8_t* action_mask, float* loss, int32_t B,
              void* stream);

/* Replaces: one fine-tune step of the hypernetwork (scripts/train.py:405-542 `train_step_pmap` without the
 * frozen T5 / initial-image encoders that feed it; octo/utils/train_utils.py:295-443 `create_optimizer`):
 * forward + backward of mean_b MixLoss(policy(theta_b(params), tokens_b), action_b), then (hvla_train_apply)
 * clip-by-global-norm -> AdamW with bf16 first moment and the v5 weight-decay mask -> EMA.  The image encoder is
 * either frozen (`fine_tune_pretrained_image_encoder=False`, the config default: `tokens` come from hvla_encode) or
 * trained (README.md:55; its leaves form the "shared" optimizer group at base_lr / base_weight_decay).  Between the
 * two calls the caller all-reduces `grads` over ranks (RCCL; scripts/train.py:460 `pmean`).
 * Every buffer is DEVICE memory owned by the caller; the flat parameter order is make_train_layout()
 * (csrc/train.hip) == hypervla.train.train_param_layout(); sizes from hvla_train_sizes.                  */
typedef struct hvla_train_buffers {
  float* params;           /* [n_params]                                                     */
  float* grads;            /* [n_params]  written by hvla_train_step                          */
  void* mu;                /* [n_params]  bf16 first moment                                   */
  float* nu;               /* [n_params]                                                     */
  float* ema;              /* [n_params] or NULL                                              */
  float* theta;            /* [B, G]                                                         */
  float* dtheta;           /* [B, G]                                                         */
  float* work;             /* [workspace_floats]                                              */
  float* loss;             /* [B] per-sample loss                                             */
  float* actions;          /* [B, horizon, action_dim] or NULL                                */
  float* logits;           /* [B, horizon] or NULL                                            */
  float* sqsum;            /* [1] scratch for the global gradient norm                        */
  const uint8_t* wd_mask;  /* [n_params] 1 where decoupled weight decay applies (the caller builds  */
                           /*   the mask of its weight_decay_strategy, train_utils.py:330-375)     */
  const float* params0;    /* [n_encoder] pretrained encoder weights (delta decay) or NULL     */
} hvla_train_buffers;
typedef struct hvla_train_hyper {
  float lr, b1, b2, eps, weight_decay, clip, ema_decay;
  int32_t step, forward_only;
  float base_lr, base_weight_decay; /* optimizer group of the shared DINOv2 leaves               */
  int32_t train_encoder;            /* base_vit.py:67 fine_tune_pretrained_image_encoder        */
} hvla_train_hyper;
/* out = { n_params, G, workspace_floats, n_hypernet }.  With train_encoder the shared DINOv2 leaves
 * (hypervla.config.encoder_leaves order) occupy params[n_hypernet, n_params).                     */
int hvla_train_sizes(hvla_ctx* ctx, int32_t B, int32_t train_encoder, int64_t out[4]);
/* Exactly one of `tokens` (f32 [B, P, E] from hvla_encode: frozen encoder) and `images` (u8 [B, H, W, 3]:
 * DINOv2 runs in f32 inside the step and receives gradients, README.md:55) is non-NULL.          */
int hvla_train_step(hvla_ctx* ctx, const hvla_train_buffers* buf, const float* token_embedding,
                    const int64_t* attention_mask, const float* initial_cls, const float* tokens,
                    const uint8_t* images, const float* target, const uint8_t* timestep_mask,
                    const uint8_t* action_mask, int32_t B, const hvla_train_hyper* hyper, void* stream);
int hvla_train_apply(hvla_ctx* ctx, const hvla_train_buffers* buf, const hvla_train_hyper* hyper, void* stream);
/* Replaces: the point where scripts/train.py:460 `jax.lax.pmean(grads, "batch")` lets XLA overlap the gradient
 * all-reduce with the backward pass.  hvla_train_step finishes `grads` in three contiguous buckets, in this order:
 *   0  the shared DINOv2 leaves  [n_hyper, n_hyper + n_encoder)   (trained encoder only; after the image encoder's backward)
 *   1  the output heads          [offset of W_cat, n_hyper)       (after the weight-generation backward)
 *   2  the context encoder       [0, offset of W_cat)             (end of the step)
 * hvla_train_bucket_ranges writes (offset, length) of buckets 0, 1, 2 into out[6] (length 0 = not produced);
 * hvla_train_wait_bucket makes `stream` (the caller's communication stream) wait until bucket `bucket` of the LAST
 * hvla_train_step is final, without blocking the host: the caller then enqueues its all-reduce of that range there.     */
int hvla_train_bucket_ranges(hvla_ctx* ctx, int32_t train_encoder, int64_t out[6]);
int hvla_train_wait_bucket(hvla_ctx* ctx, int32_t bucket, void* stream);
/* Measurement only (bench.py --finetune `roofline`): while on, every batched GEMM launch of hvla_train_step (the split-bf16
 * kernel that carries ~80 % of the step: forward, input-gradient and weight-gradient products) is bracketed by HIP events on the
 * launch stream.  hvla_train_profile_read returns, since the last read, the summed milliseconds of those launches, their
 * f32-equivalent work (2 M N K each; three bf16 matrix instructions per product on the hardware) and their number.           */
int hvla_train_profile(hvla_ctx* ctx, int32_t on);
int hvla_train_profile_read(hvla_ctx* ctx, float* gemm_ms, double* gemm_flops, int32_t* launches);
/* Replaces: one micro-step of optax.MultiSteps under the reference's chain(clip_by_global_norm, MultiSteps(adamw))
 * (octo/utils/train_utils.py:420-426, grad_accumulation_steps > 1): acc += clip_by_global_norm(buf->grads) * inv_k.
 * After k micro-steps the caller runs hvla_train_apply with `grads` pointing at acc and hyper.clip = +inf.        */
int hvla_train_accumulate(hvla_ctx* ctx, const hvla_train_buffers* buf, float* acc, float inv_k,
                          const hvla_train_hyper* hyper, void* stream);

/* Replaces: InferenceWrapper._resize_image (data/utils/hypervla_interface.py:89-121): optionally
 * tf.image.resize_with_pad(image, 256, 320) (bilinear, zero padding; `padded_resize`), then
 * tf.image.resize(lanczos3, antialias=True) to image_size x image_size, optionally the centred sqrt(0.9)
 * tf.image.crop_and_resize (bilinear), then round / clip / uint8.  frames u8 [B, H, W, 3] (device) ->
 * images u8 [B, image_size, image_size, 3] (device), the input of hvla_step / hvla_encode_hidden.
 * flags: HVLA_PREPROCESS_CROP | HVLA_PREPROCESS_PAD.                                                              */
#define HVLA_PREPROCESS_CROP 1
#define HVLA_PREPROCESS_PAD 2
int hvla_preprocess(hvla_ctx* ctx, const uint8_t* frames, int32_t B, int32_t H, int32_t W, int32_t flags, uint8_t* images,
                    void* stream);

/* Replaces: the frozen instruction encoder `LanguageTokenizer('t5-base')` that produces
 * task["language_instruction"]["token_embedding"] (octo/model/components/tokenizers.py:186-211;
 * data/utils/language_tokenizer.py:9-28; scripts/train.py:407-415 runs it inside every training step):
 * FlaxT5EncoderModel(config).module(input_ids, attention_mask).last_hidden_state.  Optional: load once, then
 * input_ids / attention_mask i64 [B, T] (device) -> token_embedding f32 [B, T, d_model] (device), ready for
 * hvla_generate.  Tensors are named as in the flax tree ('/'-joined, kernels [in, out]; an `hf_model/` prefix is
 * accepted); d_model must equal the hypernetwork's lang_dim.  Tokenisation (SentencePiece) stays on the host.      */
typedef struct hvla_t5_config {
  int32_t vocab, d_model, d_kv, heads, d_ff, layers, buckets, max_distance;
  float eps;
  int32_t max_tokens, max_batch;
} hvla_t5_config;
int hvla_t5_load(hvla_ctx* ctx, const hvla_t5_config* cfg, const hvla_tensor_desc* tensors, int32_t n);
int hvla_t5_encode(hvla_ctx* ctx, const int64_t* input_ids, const int64_t* attention_mask, float* token_embedding,
                   int32_t B, int32_t T, void* stream);

/* Live per-kernel timing with HIP events recorded on the launch stream (bench.py's roofline leg).
 *   mode 0: off (default);  1: only the selected categories (hvla_profile_select; the encoder fc1 GEMM unless told otherwise);  2: every category.
 * hvla_profile_read synchronises the recorded events, adds their durations per category into
 * ms[HVLA_PROF_N] / launches[HVLA_PROF_N], and clears the event pool.                            */
#define HVLA_PROF_PATCH 0   /* im2col + patch-embedding GEMM + CLS rows */
#define HVLA_PROF_LN 1      /* encoder LayerNorms                       */
#define HVLA_PROF_QKV 2     /* encoder QKV GEMM                         */
#define HVLA_PROF_ATTN 3    /* encoder attention                        */
#define HVLA_PROF_OUT 4     /* encoder out-projection GEMM (+residual)  */
#define HVLA_PROF_FC1 5     /* encoder fc1 GEMM (+erf GELU)             */
#define HVLA_PROF_FC2 6     /* encoder fc2 GEMM (+residual)             */
#define HVLA_PROF_POLICY 7  /* generated-policy megakernel              */
#define HVLA_PROF_COMP 8    /* encoder: the small launches in front of each big GEMM -- mean rows of its activation operand and the 2 B
                               latency-bound rows (B CLS rows + B rows that compensate the weight rounding)                      */
#define HVLA_PROF_N 9
int hvla_profile(hvla_ctx* ctx, int32_t mode);
int hvla_profile_read(hvla_ctx* ctx, float* ms, int32_t* launches);
/* Which categories mode 1 times: bit c = HVLA_PROF_c (default 1 << HVLA_PROF_FC1).  bench.py first reads a mode-2 pass, then selects
 * the categories of the kernel symbol with the largest share of the step (round 6: out + fc2 run one instantiation) for the events
 * inside its timed region.  A mask with a bit at or above HVLA_PROF_N, or zero, is HVLA_E_SHAPE.                                   */
int hvla_profile_select(hvla_ctx* ctx, uint32_t category_mask);

/* Measurement only (bench.py).  hvla_launches: kernel launches (and memset nodes) the ctx has enqueued since the last call
 * -- hvla_encode / hvla_policy / hvla_step / hvla_ensemble count themselves -- so that "launches per step" is what the library
 * did, not a restatement of its host logic.  hvla_box_probe: what THIS device sustains right now, in a probe kernel of its own
 * (never inside a product kernel): out[0] = shader clock in MHz under a chip-wide matrix-core loop (shader-clock ticks per tick of
 * the constant 100 MHz clock), out[1] = that loop's TFLOP/s (dense fp16 v_mfma_f32_32x32x16, eight waves per CU), out[2] = its
 * duration in ms.  The boxes of a pool differ by several per cent: a step time is read against these.                        */
int64_t hvla_launches(hvla_ctx* ctx);
int hvla_box_probe(hvla_ctx* ctx, float out[3], void* stream);

/* Test instrumentation: hvla_encode with a range audit of every 16-bit MFMA operand the encoder writes, over all layers:
 * site 0 LayerNorm outputs, 1 stored q / k / v, 2 attention outputs, 3 GELU outputs.  maxabs f32 [4] = largest finite
 * |value| per site, nonfinite i32 [4] = number of inf / NaN values (an fp16 overflow shows up here).  HOST pointers.    */
int hvla_encode_audit(hvla_ctx* ctx, const uint8_t* images, int32_t B, float* maxabs, int32_t* nonfinite, void* stream);

/* Primitive self-check used by tests: runs the MFMA fragment-layout probes on the ctx's device
 * and returns HVLA_OK only if every probe matches its exact integer expectation.                */
int hvla_selftest(hvla_ctx* ctx, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* HVLA_H_ */
