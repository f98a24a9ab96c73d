// resize.hip — observation preprocessing on the device (reference
// data/utils/hypervla_interface.py:89-121 `InferenceWrapper._resize_image` without padded_resize):
//   tf.image.resize(image, (S, S), method="lanczos3", antialias=True)   [scale_and_translate_op.cc: per output index a
//     span of source pixels with Lanczos-3 weights stretched by max(in/out, 1), renormalised; rows, then columns; f32]
//   optional tf.image.crop_and_resize with the centred sqrt(0.9) box, bilinear (crop_and_resize_op.cc)
//   round-half-even, clip to [0, 255], uint8.
// Byte work on a few hundred KB per frame: three small kernels, span tables built once per (H, W, S) on the host.
// Products and sums are kept un-fused (__fmul_rn / __fadd_rn) so that the f32 results follow the stated operation
// order of the CPU restatement; residual differences come only from sinf in the weights.
#include <cmath>
#include <vector>

#include "kernels.h"

namespace hvla {

static float lanczos3(float x) {
  const float pi = 3.14159265359f;
  x = fabsf(x);
  if (x > 3.0f) return 0.f;
  if (x <= 1e-3f) return 1.f;
  return 3.0f * sinf(pi * x) * sinf(pi * x / 3.0f) / (pi * pi * x * x);
}

void build_resize_spans(int in_size, int out_size, std::vector<int>& start, std::vector<int>& count, std::vector<float>& w,
                        int& maxspan) {
  const float scale = (float)out_size / (float)in_size, inv_scale = 1.0f / scale;
  const float kernel_scale = inv_scale > 1.0f ? inv_scale : 1.0f, radius = 3.0f;
  maxspan = (int)ceilf(2.f * radius * kernel_scale) + 2;
  start.assign(out_size, 0);
  count.assign(out_size, 0);
  w.assign((size_t)out_size * maxspan, 0.f);
  for (int x = 0; x < out_size; ++x) {
    const float sample_f = ((float)x + 0.5f) * inv_scale;
    if (sample_f < 0.f || sample_f > (float)in_size) continue;
    int s = (int)ceilf(sample_f - radius * kernel_scale - 0.5f), e = (int)floorf(sample_f + radius * kernel_scale - 0.5f);
    s = s < 0 ? 0 : (s > in_size - 1 ? in_size - 1 : s);
    e = (e < 0 ? 0 : (e > in_size - 1 ? in_size - 1 : e)) + 1;
    float tot = 0.f;
    for (int k = s; k < e; ++k) {
      const float v = lanczos3(((float)k + 0.5f - sample_f) / kernel_scale);
      w[(size_t)x * maxspan + (k - s)] = v;
      tot += v;
    }
    if (fabsf(tot) >= 1000.0f * 1.17549435e-38f) {
      const float inv = 1.0f / tot;
      for (int k = 0; k < e - s; ++k) w[(size_t)x * maxspan + k] *= inv;
    }
    start[x] = s;
    count[x] = e - s;
  }
}

// optional first stage, tf.image.resize_with_pad(image, PH, PW) (hypervla_interface.py:90-95): legacy bilinear resize with
// half-pixel centres to (rh, rw) = floor(size / max(W / PW, H / PH)), zero padding around it; f32 [B][PH][PW][3]
__global__ void pad_bilinear_kernel(const uint8_t* __restrict__ src, float* __restrict__ dst, int B, int H, int W, int PH,
                                    int PW, int rh, int rw, int ph, int pw) {
  const long n = (long)B * PH * PW * 3;
  const float ys = __fdiv_rn((float)H, (float)rh), xs = __fdiv_rn((float)W, (float)rw);
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const int c = (int)(i % 3), x = (int)((i / 3) % PW) - pw, y = (int)((i / (3L * PW)) % PH) - ph;
    const long b = i / (3L * PW * PH);
    float v = 0.f;
    if (x >= 0 && x < rw && y >= 0 && y < rh) {
      const float in_y = __fadd_rn(__fmul_rn(__fadd_rn((float)y, 0.5f), ys), -0.5f);
      const float in_x = __fadd_rn(__fmul_rn(__fadd_rn((float)x, 0.5f), xs), -0.5f);
      const float fy = floorf(in_y), fx = floorf(in_x);
      int y0 = (int)fy, y1 = (int)ceilf(in_y), x0 = (int)fx, x1 = (int)ceilf(in_x);
      y0 = y0 < 0 ? 0 : y0; x0 = x0 < 0 ? 0 : x0;
      y1 = y1 > H - 1 ? H - 1 : y1; x1 = x1 > W - 1 ? W - 1 : x1;
      const float ly = in_y - fy, lx = in_x - fx;
      const uint8_t* base = src + b * H * W * 3 + c;
      const float tl = (float)base[((long)y0 * W + x0) * 3], tr = (float)base[((long)y0 * W + x1) * 3];
      const float bl = (float)base[((long)y1 * W + x0) * 3], br = (float)base[((long)y1 * W + x1) * 3];
      const float top = __fadd_rn(tl, __fmul_rn(tr - tl, lx)), bot = __fadd_rn(bl, __fmul_rn(br - bl, lx));
      v = __fadd_rn(top, __fmul_rn(bot - top, ly));
    }
    dst[i] = v;
  }
}

// rows: tmp[b][y][x][c] = sum_k w[y][k] * src[b][start[y] + k][x][c]       (src u8 or f32 [B][H][W][3], tmp f32 [B][S][W][3])
template <typename SRC>
__global__ void resize_rows_kernel(const SRC* __restrict__ src, float* __restrict__ tmp, const int* __restrict__ start,
                                   const int* __restrict__ count, const float* __restrict__ w, int maxspan, int B, int H,
                                   int W, int S) {
  const long n = (long)B * S * W * 3;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const int xc = (int)(i % (W * 3)), y = (int)((i / (W * 3)) % S), b = (int)(i / ((long)W * 3 * S));
    const SRC* p = src + ((long)b * H + start[y]) * W * 3 + xc;
    const float* wy = w + (long)y * maxspan;
    float acc = 0.f;
    for (int k = 0; k < count[y]; ++k) acc = __fadd_rn(acc, __fmul_rn(wy[k], (float)p[(long)k * W * 3]));
    tmp[i] = acc;
  }
}
// columns: out[b][y][x][c] = sum_k w[x][k] * tmp[b][y][start[x] + k][c]    (out f32 [B][S][S][3])
__global__ void resize_cols_kernel(const float* __restrict__ tmp, float* __restrict__ out, const int* __restrict__ start,
                                   const int* __restrict__ count, const float* __restrict__ w, int maxspan, int B, int W,
                                   int S) {
  const long n = (long)B * S * S * 3;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const int c = (int)(i % 3), x = (int)((i / 3) % S);
    const long by = i / (3L * S);
    const float* p = tmp + (by * W + start[x]) * 3 + c;
    const float* wx = w + (long)x * maxspan;
    float acc = 0.f;
    for (int k = 0; k < count[x]; ++k) acc = __fadd_rn(acc, __fmul_rn(wx[k], p[(long)k * 3]));
    out[i] = acc;
  }
}
// optional centred crop resized back bilinearly, then round-half-even / clip / uint8
__global__ void resize_finish_kernel(const float* __restrict__ img, uint8_t* __restrict__ dst, int B, int S, int crop,
                                     float box_lo, float box_hi) {
  const long n = (long)B * S * S * 3;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    float v;
    if (!crop) {
      v = img[i];
    } else {
      const int c = (int)(i % 3), x = (int)((i / 3) % S), y = (int)((i / (3L * S)) % S);
      const long b = i / (3L * S * S);
      const float sc = __fdiv_rn(__fmul_rn(box_hi - box_lo, (float)(S - 1)), (float)(S - 1));
      const float in_y = __fadd_rn(__fmul_rn(box_lo, (float)(S - 1)), __fmul_rn((float)y, sc));
      const float in_x = __fadd_rn(__fmul_rn(box_lo, (float)(S - 1)), __fmul_rn((float)x, sc));
      if (in_y < 0.f || in_y > (float)(S - 1) || in_x < 0.f || in_x > (float)(S - 1)) {
        v = 0.f;
      } else {
        const int top = (int)floorf(in_y), bot = (int)ceilf(in_y), l = (int)floorf(in_x), r = (int)ceilf(in_x);
        const float ly = in_y - (float)top, lx = in_x - (float)l;
        const float* base = img + b * S * S * 3 + c;
        const float tl = base[((long)top * S + l) * 3], tr = base[((long)top * S + r) * 3];
        const float bl = base[((long)bot * S + l) * 3], br = base[((long)bot * S + r) * 3];
        const float t = __fadd_rn(tl, __fmul_rn(tr - tl, lx)), bb = __fadd_rn(bl, __fmul_rn(br - bl, lx));
        v = __fadd_rn(t, __fmul_rn(bb - t, ly));
      }
    }
    v = rintf(v);                                   // round half to even (tf.round)
    v = v < 0.f ? 0.f : (v > 255.f ? 255.f : v);
    dst[i] = (uint8_t)v;
  }
}

static inline dim3 g1(long n) { long b = (n + 255) / 256; return dim3((unsigned)(b > 16384 ? 16384 : b)); }

void resize_with_pad_dims(int H, int W, int PH, int PW, int& rh, int& rw, int& ph, int& pw) {
  const float r1 = (float)W / (float)PW, r2 = (float)H / (float)PH, ratio = r1 > r2 ? r1 : r2;
  const float rhf = (float)H / ratio, rwf = (float)W / ratio;
  rh = (int)floorf(rhf);
  rw = (int)floorf(rwf);
  ph = (int)floorf(((float)PH - rhf) / 2.f);
  pw = (int)floorf(((float)PW - rwf) / 2.f);
  ph = ph < 0 ? 0 : ph;
  pw = pw < 0 ? 0 : pw;
}

hipError_t launch_resize(const uint8_t* src, uint8_t* dst, float* tmp_rows, float* tmp_img, const int* row_start,
                         const int* row_count, const float* row_w, int row_span, const int* col_start, const int* col_count,
                         const float* col_w, int col_span, int B, int H, int W, int S, int crop, hipStream_t st,
                         float* padded, int src_h, int src_w) {
  if (padded) {                                      // (H, W) = padded size, (src_h, src_w) = camera frame
    int rh, rw, ph, pw;
    resize_with_pad_dims(src_h, src_w, H, W, rh, rw, ph, pw);
    hipLaunchKernelGGL(pad_bilinear_kernel, g1((long)B * H * W * 3), dim3(256), 0, st, src, padded, B, src_h, src_w, H, W, rh,
                       rw, ph, pw);
    hipLaunchKernelGGL(resize_rows_kernel<float>, g1((long)B * S * W * 3), dim3(256), 0, st, (const float*)padded, tmp_rows,
                       row_start, row_count, row_w, row_span, B, H, W, S);
  } else {
    hipLaunchKernelGGL(resize_rows_kernel<uint8_t>, g1((long)B * S * W * 3), dim3(256), 0, st, src, tmp_rows, row_start, row_count,
                       row_w, row_span, B, H, W, S);
  }
  hipLaunchKernelGGL(resize_cols_kernel, g1((long)B * S * S * 3), dim3(256), 0, st, tmp_rows, tmp_img, col_start, col_count,
                     col_w, col_span, B, W, S);
  const double scale = sqrt(0.9), off = (1.0 - scale) / 2.0;         // python floats in the reference, cast with the box
  hipLaunchKernelGGL(resize_finish_kernel, g1((long)B * S * S * 3), dim3(256), 0, st, tmp_img, dst, B, S, crop, (float)off,
                     (float)(off + scale));
  return hipGetLastError();
}

}  // namespace hvla
