// CPU-only check of the host-side layout and packing code (csrc/layout.h, csrc/pack.h) under AddressSanitizer and
// UndefinedBehaviorSanitizer.  Built and run by tests/test_host_sanitizers.py:
//   g++ -std=c++17 -O1 -g -fsanitize=address,undefined -fno-sanitize-recover=all -I hyper-vla_amd/csrc tests/native/pack_sanitize.cpp
// Exercises the index arithmetic api.hip runs at hvla_create time -- the permutation from reference flat order to the
// MFMA fragment order, the W_cat fragment packing and the 16-bit transposes of the encoder matrices -- on the README
// geometry and a small one, and checks what the packing promises: the permutation is a bijection onto the generated
// parameters, every packed value is the source value (hi + lo reconstructs it to 2^-16 relative), paddings are zero.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>

#include "pack.h"

using namespace hvla;

#define REQUIRE(c, ...) do { if (!(c)) { printf("FAILED %s:%d: ", __FILE__, __LINE__); printf(__VA_ARGS__); printf("\n"); return 1; } } while (0)

static int check_geometry(const Geom& g, const char* name) {
  const PackedLayout lay = build_layout(g);
  const PolicyLayout& pl = lay.pl;
  const int Gtot = pl.Gm + pl.Gv, C = g.C;
  REQUIRE((int)lay.perm.size() == Gtot, "%s: perm has %zu entries for %d slots", name, lay.perm.size(), Gtot);
  REQUIRE(Gtot % 32 == 0 && C % 16 == 0, "%s: tile shapes", name);
  std::vector<int> seen(pl.G, 0);
  for (int pos = 0; pos < Gtot; ++pos) {
    const int ref = lay.perm[pos];
    REQUIRE(ref >= -1 && ref < pl.G, "%s: perm[%d] = %d outside [−1, %d)", name, pos, ref, pl.G);
    if (ref >= 0) ++seen[ref];
  }
  for (int i = 0; i < pl.G; ++i) REQUIRE(seen[i] == 1, "%s: reference parameter %d packed %d times", name, i, seen[i]);

  auto leaves = generated_leaves(g);
  int64_t total = 0;
  for (const auto& l : leaves) {
    REQUIRE(l.offset == total, "%s: leaf %s offset", name, l.flat.c_str());
    total += l.size;
  }
  REQUIRE(total == pl.G, "%s: leaves cover %lld of %d parameters", name, (long long)total, pl.G);

  std::mt19937 rng(7);
  std::normal_distribution<float> nd(0.f, 0.05f);
  std::vector<std::vector<float>> K(leaves.size()), Bv(leaves.size());
  std::vector<const float*> lk(leaves.size()), lb(leaves.size());
  for (size_t i = 0; i < leaves.size(); ++i) {
    K[i].resize((size_t)C * leaves[i].size);      // exact-size allocations: an out-of-range read is an ASAN report
    Bv[i].resize(leaves[i].size);
    for (auto& v : K[i]) v = nd(rng);
    for (auto& v : Bv[i]) v = nd(rng);
    lk[i] = K[i].data();
    lb[i] = Bv[i].data();
  }
  std::vector<uint16_t> hi, lo;
  std::vector<float> bc;
  pack::pack_wcat(lay, leaves, lk, lb, C, hi, lo, bc);
  const int KS = C / 16;
  REQUIRE(hi.size() == (size_t)(Gtot / 32) * KS * 512 && lo.size() == hi.size() && (int)bc.size() == Gtot, "%s: packed sizes", name);
  std::vector<int> leaf_of(pl.G);
  for (size_t i = 0; i < leaves.size(); ++i)
    for (int64_t j = 0; j < leaves[i].size; ++j) leaf_of[leaves[i].offset + j] = (int)i;
  for (int pt = 0; pt < Gtot / 32; ++pt)
    for (int lane = 0; lane < 64; ++lane) {
      const int rho = lane & 31, hk = lane >> 5;
      const int tau = 16 * ((rho >> 2) & 1) + (rho & 3) + 4 * (rho >> 3);
      const int ref = lay.perm[pt * 32 + tau];
      for (int ks = 0; ks < KS; ++ks)
        for (int j = 0; j < 8; ++j) {
          const size_t o = ((size_t)(pt * KS + ks) * 64 + lane) * 8 + j;
          const float got = pack::bf2f(hi[o]) + pack::bf2f(lo[o]);
          float want = 0.f;
          if (ref >= 0) {
            const int li = leaf_of[ref];
            want = K[li][(size_t)(16 * ks + 8 * hk + j) * leaves[li].size + (ref - leaves[li].offset)];
          }
          REQUIRE(std::fabs(got - want) <= std::ldexp(std::fabs(want), -15), "%s: W_cat tile %d lane %d ks %d j %d: %g vs %g", name,
                  pt, lane, ks, j, got, want);
        }
    }
  for (int pos = 0; pos < Gtot; ++pos) {
    const int ref = lay.perm[pos];
    const float want = ref >= 0 ? Bv[leaf_of[ref]][ref - leaves[leaf_of[ref]].offset] : 0.f;
    REQUIRE(bc[pos] == want, "%s: b_cat[%d]", name, pos);
  }
  printf("%s: %d generated parameters in %d packed slots, %zu W_cat fragment values checked\n", name, pl.G, Gtot, hi.size());
  return 0;
}

static int check_matrix(int K, int N, bool bf) {
  std::mt19937 rng(K * 131 + N);
  std::normal_distribution<float> nd(0.f, 0.04f);
  std::vector<float> src((size_t)K * N);
  for (auto& v : src) v = nd(rng);
  src[0] = 0.f; src[1] = 65504.f; src[2] = 6e-8f; src[3] = -1e-5f; src[4] = 3e-5f;     // zero, max, subnormals
  std::vector<uint16_t> w16((size_t)K * N), d16(w16.size());
  pack::pack_matrix_t(src.data(), K, N, bf, w16.data(), d16.data());
  for (int n = 0; n < N; ++n)
    for (int k = 0; k < K; ++k) {
      const float w = src[(size_t)k * N + n], h = pack::from16(w16[(size_t)n * K + k], bf);
      const float d = pack::from16(d16[(size_t)n * K + k], bf) / 4096.f;
      const float ulp = bf ? std::ldexp(std::fabs(w), -8) : std::fmax(std::ldexp(std::fabs(w), -11), std::ldexp(1.f, -25));
      REQUIRE(std::fabs(h - w) <= ulp, "matrix %dx%d: [%d][%d] %g rounds to %g", K, N, k, n, w, h);
      REQUIRE(std::fabs(h + d - w) <= ulp * (bf ? 0x1p-7f : 0x1p-9f) + 0x1p-37f, "matrix %dx%d: [%d][%d] residue %g + %g vs %g", K, N, k, n, h, d, w);
    }
  return 0;
}

// the software binary16 conversion against the compiler's (when it has one) and against round trips
static int check_half() {
  for (uint32_t h = 0; h < 0x10000u; ++h) {
    if (((h >> 10) & 0x1f) == 0x1f && (h & 0x3ff)) continue;                 // NaNs: payload is free
    const float f = pack::h2f((uint16_t)h);
    REQUIRE(pack::f2h(f) == h, "f2h(h2f(%#x)) = %#x", h, pack::f2h(f));
    if (((h >> 10) & 0x1f) == 0x1f) continue;
    // halfway cases between h and its successor go to the even one
    const uint16_t nx = (uint16_t)(h + 1);
    if ((nx & 0x7fff) > 0x7c00 || (h & 0x7fff) == 0x7bff) continue;
    const float mid = 0.5f * (f + pack::h2f(nx));
    REQUIRE(pack::f2h(mid) == ((h & 1) ? nx : h), "tie at %#x", h);
    REQUIRE(pack::f2h(std::nextafterf(mid, f)) == h && pack::f2h(std::nextafterf(mid, pack::h2f(nx))) == nx, "around the tie at %#x", h);
  }
  REQUIRE(pack::f2h(65519.99f) == 0x7bff && pack::f2h(65520.f) == 0x7c00 && pack::f2h(1e9f) == 0x7c00, "overflow");
  REQUIRE(pack::f2h(-1e9f) == 0xfc00 && pack::f2h(0x1p-25f) == 0 && pack::f2h(std::nextafterf(0x1p-25f, 1.f)) == 1, "edges");
#if defined(__FLT16_MANT_DIG__)
  std::mt19937 rng(3);
  for (int i = 0; i < 2000000; ++i) {
    uint32_t u = rng();
    float f;
    memcpy(&f, &u, 4);
    if (f != f) continue;
    const _Float16 x = (_Float16)f;
    uint16_t hw;
    memcpy(&hw, &x, 2);
    REQUIRE(pack::f2h(f) == hw, "f2h(%a) = %#x, compiler says %#x", f, pack::f2h(f), hw);
  }
  printf("binary16 conversion: all 63488 finite halves round-trip, ties to even, 2000000 random floats match _Float16\n");
#else
  printf("binary16 conversion: all 63488 finite halves round-trip, ties to even (no _Float16 in this compiler)\n");
#endif
  return 0;
}

int main(int argc, char** argv) {
  if (argc > 1) {            // dump (float bits, f2h) pairs for the Python side to compare with numpy's float16
    FILE* f = fopen(argv[1], "wb");
    if (!f) return 2;
    std::mt19937 rng(11);
    for (int i = 0; i < 1000000; ++i) {
      uint32_t u = rng();
      if (i & 1) u = (u & 0x807fffffu) | ((96u + (u >> 23) % 48u) << 23);    // half of them inside / around the binary16 range
      float x;
      memcpy(&x, &u, 4);
      const uint32_t rec[2] = {u, pack::f2h(x)};
      fwrite(rec, 4, 2, f);
    }
    fclose(f);
  }
  Geom full{224, 14, 768, 12, 12, 3072, 64, 4, 4, 128, 4, 7, 5.f, 5.f, 128, 6, 4, 512, 32, 768, 1};
  Geom small{28, 14, 64, 2, 2, 128, 32, 2, 2, 64, 2, 3, 5.f, 5.f, 32, 2, 2, 64, 8, 64, 1};
  if (check_half()) return 1;
  if (check_geometry(full, "README geometry")) return 1;
  if (check_geometry(small, "small geometry")) return 1;
  for (bool bf : {false, true}) {
    if (check_matrix(768, 96, bf)) return 1;
    if (check_matrix(33, 7, bf)) return 1;
    if (check_matrix(588, 768, bf)) return 1;
  }
  printf("OK\n");
  return 0;
}
