// CPU-only check of csrc/plan.h (the arithmetic-only launch decisions of run_encoder and the XCD tile order of the small GEMM
// kernels).  Built and run by tests/test_host_sanitizers.py with g++ -fsanitize=address,undefined.
#include <cstdio>
#include <vector>

#include "plan.h"

using namespace hvla;

#define REQUIRE(c, ...) do { if (!(c)) { printf("FAILED %s:%d: ", __FILE__, __LINE__); printf(__VA_ARGS__); printf("\n"); return 1; } } while (0)

int main() {
  // xcd_run: a bijection of [0, nwg) for every grid size, XCD x (= block id % 8) owning ONE contiguous run of tile indices
  for (int nwg = 1; nwg <= 4100; ++nwg) {
    std::vector<int> seen(nwg, 0), lo(8, nwg), hi(8, -1), cnt(8, 0);
    for (int b = 0; b < nwg; ++b) {
      const int t = xcd_run(b, nwg);
      REQUIRE(t >= 0 && t < nwg, "nwg %d: block %d -> tile %d", nwg, b, t);
      ++seen[t];
      const int x = b % 8;
      if (t < lo[x]) lo[x] = t;
      if (t > hi[x]) hi[x] = t;
      ++cnt[x];
    }
    for (int t = 0; t < nwg; ++t) REQUIRE(seen[t] == 1, "nwg %d: tile %d taken %d times", nwg, t, seen[t]);
    for (int x = 0; x < 8; ++x)
      if (cnt[x]) REQUIRE(hi[x] - lo[x] + 1 == cnt[x], "nwg %d: XCD %d owns %d tiles spread over [%d, %d]", nwg, x, cnt[x], lo[x], hi[x]);
  }
  // the small-row launch at B = 256 (RT = 4 CLS + 8 mean-row tiles per column tile): a 64-column slice of W or dW is read by
  // the row tiles of one problem of one column tile -- on at most two XCDs (one where the runs are whole columns: fc1, out / fc2)
  const int RT = 12, ncols[3] = {36, 48, 12};
  for (int nc : ncols) {
    const int nwg = RT * nc;
    std::vector<unsigned> xcds(2 * nc, 0);
    for (int b = 0; b < nwg; ++b) {
      const int t = xcd_run(b, nwg), bn = t / RT, bm = t % RT;
      xcds[2 * bn + (bm >= 4)] |= 1u << (b % 8);
    }
    int worst = 0;
    for (unsigned m : xcds) { const int c = __builtin_popcount(m); if (c > worst) worst = c; }
    REQUIRE(worst <= 2, "%d column tiles: a slice is read on %d XCDs", nc, worst);
    if ((nwg / 8) % RT == 0) REQUIRE(worst == 1, "%d column tiles: whole columns per XCD, yet a slice is on %d XCDs", nc, worst);
  }
  // non-temporal 16-bit outputs: 32-bit BYTE offsets.  fc1 at S = 257, F = 3072, 2-byte elements
  const size_t S = 257, F = 3072;
  REQUIRE(big_output(256 * S, F, 2) && nt16_addressable(256 * S, F, 2), "B = 256 takes the non-temporal form");
  REQUIRE(!big_output(16 * S, F, 2), "B = 16 does not (25 MB)");
  REQUIRE(nt16_addressable(2720 * S, F, 2), "B = 2720: %zu bytes", 2720 * S * F * 2);
  REQUIRE(!nt16_addressable(2721 * S, F, 2), "B = 2721: %zu bytes wrap a 32-bit byte offset", 2721 * S * F * 2);
  REQUIRE(!nt16_addressable((size_t)1 << 16, (size_t)1 << 15, 2), "exactly 4 GiB");
  REQUIRE(nt16_addressable(((size_t)1 << 16) - 1, (size_t)1 << 15, 2), "one row below 4 GiB");
  printf("OK\n");
  return 0;
}
