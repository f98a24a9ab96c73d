// pack.h — host-only packing of checkpoint tensors into the device layouts (used by api.hip; no HIP types, so that
// tests/native/pack_asan.cpp can run it under -fsanitize=address,undefined on the CPU).
#pragma once
#include <stdint.h>
#include <string.h>

#include <vector>

#include "layout.h"

namespace hvla {
namespace pack {

inline uint16_t f2bf(float f) {          // round-to-nearest-even, NaN preserved
  uint32_t u;
  memcpy(&u, &f, 4);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);
  return (uint16_t)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
}
inline float bf2f(uint16_t h) {
  uint32_t u = (uint32_t)h << 16;
  float f;
  memcpy(&f, &u, 4);
  return f;
}
// IEEE binary16, round-to-nearest-even, subnormals kept, overflow to infinity (== the hardware conversion)
inline uint16_t f2h(float f) {
  uint32_t u;
  memcpy(&u, &f, 4);
  const uint32_t sign = (u >> 16) & 0x8000u;
  u &= 0x7fffffffu;
  if (u >= 0x7f800000u) return (uint16_t)(sign | 0x7c00u | (u > 0x7f800000u ? 0x200u : 0u));   // inf / NaN
  if (u >= 0x477ff000u) return (uint16_t)(sign | 0x7c00u);                                       // rounds to >= 65520: inf
  if (u < 0x38800000u) {                                                                         // below 2^-14: subnormal
    if (u < 0x33000000u) return (uint16_t)sign;                                                  // below 2^-25: zero
    const int shift = 126 - (int)(u >> 23);                          // 14 .. 24
    const uint32_t mant = (u & 0x7fffffu) | 0x800000u;
    const uint32_t q = mant >> shift, rem = mant & ((1u << shift) - 1u), halfway = 1u << (shift - 1);
    return (uint16_t)(sign | (q + ((rem > halfway || (rem == halfway && (q & 1u))) ? 1u : 0u)));
  }
  const uint32_t v = u - 0x38000000u;                                // rebias 127 -> 15
  return (uint16_t)(sign | ((v + 0xfffu + ((v >> 13) & 1u)) >> 13));
}
inline float h2f(uint16_t h) {
  const uint32_t sign = (uint32_t)(h & 0x8000u) << 16, e = (h >> 10) & 0x1fu, m = h & 0x3ffu;
  uint32_t u;
  if (e == 0) {
    if (m == 0) u = sign;
    else {
      int s = 0;
      uint32_t mm = m;
      while (!(mm & 0x400u)) mm <<= 1, ++s;
      u = sign | ((uint32_t)(113 - s) << 23) | ((mm & 0x3ffu) << 13);
    }
  } else if (e == 31) u = sign | 0x7f800000u | (m << 13);
  else u = sign | ((e + 112u) << 23) | (m << 13);
  float f;
  memcpy(&f, &u, 4);
  return f;
}
inline uint16_t to16(float f, bool bf) { return bf ? f2bf(f) : f2h(f); }
inline float from16(uint16_t h, bool bf) { return bf ? bf2f(h) : h2f(h); }

// W_cat^T as MFMA A fragments (layout.h): tile pt, k-step ks, lane (rho = l & 31, hk = l >> 5), j; hi / lo bf16 planes, and
// b_cat in packed order.  lk[i] / lb[i]: kernel [C][size_i] / bias [size_i] of generated leaf i.
inline void pack_wcat(const PackedLayout& lay, const std::vector<LeafInfo>& leaves, const std::vector<const float*>& lk,
                      const std::vector<const float*>& lb, int C, std::vector<uint16_t>& hi, std::vector<uint16_t>& lo,
                      std::vector<float>& bc) {
  const PolicyLayout& pl = lay.pl;
  const int Gtot = pl.Gm + pl.Gv, ntiles = Gtot / 32, KS = C / 16;
  std::vector<int32_t> leaf_of(pl.G);
  for (size_t i = 0; i < leaves.size(); ++i)
    for (int64_t j = 0; j < leaves[i].size; ++j) leaf_of[leaves[i].offset + j] = (int32_t)i;
  hi.assign((size_t)ntiles * KS * 512, 0);
  lo.assign(hi.size(), 0);
  bc.assign(Gtot, 0.f);
  const int32_t* perm = lay.perm.data();
  for (int pos = 0; pos < Gtot; ++pos)
    if (perm[pos] >= 0) {
      const int ref = perm[pos], li = leaf_of[ref];
      bc[pos] = lb[li][ref - leaves[li].offset];
    }
  for (int pt = 0; pt < ntiles; ++pt)
    for (int lane = 0; lane < 64; ++lane) {
      const int rho = lane & 31, hk = lane >> 5;
      const int tau = 16 * ((rho >> 2) & 1) + (rho & 3) + 4 * (rho >> 3);
      const int ref = perm[pt * 32 + tau];
      const float* col = nullptr;
      int64_t n_leaf = 0;
      if (ref >= 0) {
        const int li = leaf_of[ref];
        col = lk[li] + (ref - leaves[li].offset);
        n_leaf = leaves[li].size;
      }
      for (int ks = 0; ks < KS; ++ks)
        for (int j = 0; j < 8; ++j) {
          const int k = 16 * ks + 8 * hk + j;
          const float w = col ? col[(int64_t)k * n_leaf] : 0.f;
          const uint16_t h = f2bf(w);
          const size_t o = ((size_t)(pt * KS + ks) * 64 + lane) * 8 + j;
          hi[o] = h;
          lo[o] = f2bf(w - bf2f(h));
        }
    }
}

// flax [K][N] -> [N][K] 16-bit, and what the rounding dropped (x 4096: stays in the normal range of fp16) for the
// per-image compensation of the encoder GEMMs
inline void pack_matrix_t(const float* src, int K, int N, bool bf, uint16_t* w16, uint16_t* d16) {
  for (int n = 0; n < N; ++n)
    for (int k = 0; k < K; ++k) {
      const float wv = src[(size_t)k * N + n];
      const uint16_t h = to16(wv, bf);
      w16[(size_t)n * K + k] = h;
      d16[(size_t)n * K + k] = to16((wv - from16(h, bf)) * 4096.f, bf);
    }
}

}  // namespace pack
}  // namespace hvla
