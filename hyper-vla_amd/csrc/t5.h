// t5.h — frozen T5 instruction encoder on the device (t5.hip)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace hvla {

struct T5Dims { int vocab, d_model, d_kv, heads, d_ff, layers, buckets, max_distance; float eps; };
struct T5LayerW { const float *ln0, *wq, *wk, *wv, *wo, *ln1, *wi, *wo2; };
struct T5Weights {
  const float* shared;        // [vocab, d_model]
  const float* relbias;       // [heads, 2 * max_tokens - 1]  bias of relative position (j - i) + max_tokens - 1
  const float* final_ln;      // [d_model]
  T5LayerW layer[24];
  int max_tokens;
};
size_t t5_workspace_floats(const T5Dims& d, int B, int T);
hipError_t t5_encode(const T5Dims& d, const T5Weights& w, float* work, const int64_t* ids, const int64_t* mask, float* out,
                     int B, int T, hipStream_t st);

}  // namespace hvla
