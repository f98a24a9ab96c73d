// t5.hip — the frozen T5-base encoder that turns tokenised instructions into `token_embedding`
// (reference: octo/model/components/tokenizers.py:186-211 `LanguageTokenizer`,
// data/utils/language_tokenizer.py:9-28; arithmetic in transformers 4.50.0 `FlaxT5EncoderModel`, un-vendored):
//   x = shared[input_ids]
//   12 x { h = RMSNorm(x); q, k, v = h Wq, h Wk, h Wv (no bias, NO 1/sqrt(d) scaling);
//          p = softmax(q k^T + relative_position_bias[h][j - i] + key mask); x += (p v) Wo;
//          h = RMSNorm(x); x += relu(h Wi) Wo2 }
//   last_hidden_state = RMSNorm(x)
// Runs once per episode on B x 32 tokens, so it is built from the generic f32-accurate batched GEMM of the fine-tune path
// (train.hip: split-bf16 on the matrix cores) plus four small kernels; parity target is the float64 restatement at 1e-4.
#include <cmath>

#include "common.h"
#include "t5.h"
#include "train.h"

namespace hvla {

#define KL(kernel, grid, block, ...) hipLaunchKernelGGL(kernel, grid, block, 0, st, __VA_ARGS__)

__global__ void t5_gather_kernel(const float* __restrict__ table, const int64_t* __restrict__ ids, float* __restrict__ x,
                                 long rows, int d, int vocab) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < rows * d; i += (long)gridDim.x * blockDim.x) {
    long id = ids[i / d];
    id = id < 0 ? 0 : (id >= vocab ? vocab - 1 : id);
    x[i] = table[id * d + i % d];
  }
}

// T5LayerNorm: y = x * rsqrt(mean(x^2) + eps) * w   (no mean subtraction, no bias); one wave per row
__global__ void t5_rmsnorm_kernel(const float* __restrict__ x, const float* __restrict__ w, float* __restrict__ y, int rows,
                                  int d, float eps) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= rows) return;
  const float* xr = x + (long)row * d;
  float q = 0.f;
  for (int c = lane; c < d; c += 64) q += xr[c] * xr[c];
  for (int o = 32; o; o >>= 1) q += __shfl_xor(q, o, 64);
  const float r = rsqrtf(q / d + eps);
  for (int c = lane; c < d; c += 64) y[(long)row * d + c] = xr[c] * r * w[c];
}

// p[b][h][i][:] = softmax_j(s[b][h][i][j] + relbias[h][j - i + TM - 1] + (mask[b][j] ? 0 : -inf)); one wave per row
__global__ void t5_softmax_kernel(float* __restrict__ s, const float* __restrict__ relbias, const int64_t* __restrict__ mask,
                                  int B, int H, int T, int TM) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= B * H * T) return;
  const int i = row % T, h = (row / T) % H, b = row / (T * H);
  float* sr = s + (long)row * T;
  const float* rb = relbias + (long)h * (2 * TM - 1) + (TM - 1 - i);
  const int64_t* mk = mask + (long)b * T;
  float mx = -3.4e38f;
  for (int j = lane; j < T; j += 64) {
    const float v = mk[j] != 0 ? sr[j] + rb[j] : -3.4e38f;
    sr[j] = v;
    mx = fmaxf(mx, v);
  }
  for (int o = 32; o; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
  float sum = 0.f;
  for (int j = lane; j < T; j += 64) {
    const float e = __expf(sr[j] - mx);      // a fully padded sequence degrades to uniform weights, as finfo.min does
    sr[j] = e;
    sum += e;
  }
  for (int o = 32; o; o >>= 1) sum += __shfl_xor(sum, o, 64);
  const float inv = 1.f / sum;
  for (int j = lane; j < T; j += 64) sr[j] *= inv;
}

__global__ void t5_relu_kernel(float* __restrict__ u, long n) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) u[i] = fmaxf(u[i], 0.f);
}

static inline dim3 g1(long n) { long b = (n + 255) / 256; return dim3((unsigned)(b > 65535 ? 65535 : b)); }

size_t t5_workspace_floats(const T5Dims& d, int B, int T) {
  const long rows = (long)B * T, inner = (long)d.heads * d.d_kv;
  return (size_t)(rows * d.d_model * 2 + rows * inner * 4 + (long)B * d.heads * T * T + rows * d.d_ff + 64);
}

hipError_t t5_encode(const T5Dims& d, const T5Weights& w, float* work, const int64_t* ids, const int64_t* mask, float* out,
                     int B, int T, hipStream_t st) {
  const int rows = B * T, D = d.d_model, H = d.heads, dk = d.d_kv, I = H * dk, F = d.d_ff;
  float* p = work;
  auto take = [&](long n) { float* r = p; p += (n + 3) / 4 * 4; return r; };
  float *x = take((long)rows * D), *h = take((long)rows * D), *q = take((long)rows * I), *k = take((long)rows * I),
        *v = take((long)rows * I), *o = take((long)rows * I), *s = take((long)B * H * T * T), *u = take((long)rows * F);
  auto lin = [&](const float* X, const float* W, float* Y, int K, int N, int acc) {       // Y[rows][N] (+)= X[rows][K] W[K][N]
    bgemm(st, false, false, BG{X, W, Y, nullptr, rows, N, K, K, N, N, 0, 0, 0, 0, 0, 0, 0, 1, 1.f, acc}, 1);
  };
  KL(t5_gather_kernel, g1((long)rows * D), dim3(256), w.shared, ids, x, (long)rows, D, d.vocab);
  for (int l = 0; l < d.layers; ++l) {
    const T5LayerW& L = w.layer[l];
    KL(t5_rmsnorm_kernel, dim3((rows + 3) / 4), dim3(256), x, L.ln0, h, rows, D, d.eps);
    lin(h, L.wq, q, D, I, 0);
    lin(h, L.wk, k, D, I, 0);
    lin(h, L.wv, v, D, I, 0);
    // s[b][h] = q_h k_h^T (batch b0 = sequence, b1 = head)
    bgemm(st, false, true, BG{q, k, s, nullptr, T, T, dk, I, I, T, (long)T * I, dk, (long)T * I, dk, (long)H * T * T, (long)T * T, 0, H, 1.f, 0}, B);
    KL(t5_softmax_kernel, dim3((B * H * T + 3) / 4), dim3(256), s, w.relbias, mask, B, H, T, w.max_tokens);
    bgemm(st, false, false, BG{s, v, o, nullptr, T, dk, T, T, I, I, (long)H * T * T, (long)T * T, (long)T * I, dk, (long)T * I, dk, 0, H, 1.f, 0}, B);
    lin(o, L.wo, x, I, D, 1);
    KL(t5_rmsnorm_kernel, dim3((rows + 3) / 4), dim3(256), x, L.ln1, h, rows, D, d.eps);
    lin(h, L.wi, u, D, F, 0);
    KL(t5_relu_kernel, g1((long)rows * F), dim3(256), u, (long)rows * F);
    lin(u, L.wo2, x, F, D, 1);
  }
  KL(t5_rmsnorm_kernel, dim3((rows + 3) / 4), dim3(256), x, w.final_ln, out, rows, D, d.eps);
  return hipGetLastError();
}

}  // namespace hvla
