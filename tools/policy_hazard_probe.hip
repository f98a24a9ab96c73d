// Diagnostic (GPU box): which operand path makes the policy megakernel run-to-run nondeterministic at -O3.
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/policy_hazard_probe.hip -o /tmp/policy_hazard_probe && /tmp/policy_hazard_probe [launches]
//
// Compiles csrc/policy.hip's kernel in its TIE variants (see mfma_tied there), runs each on one fixed random arena /
// token set `launches` times (default 3000) with the chip otherwise idle and with a memory-streaming kernel beside it, and
// counts launches whose action bytes differ from the first launch of that variant.  Numbers are random (finite), only
// bit-stability matters.  Not part of the product or the tests.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>

#include "../hyper-vla_amd/csrc/policy.hip"

using namespace hvla;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void stream_kernel(const float4* __restrict__ a, float4* __restrict__ b, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) b[i] = a[i];
}

static uint16_t f2bf(float f) {
  uint32_t u;
  memcpy(&u, &f, 4);
  return (uint16_t)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
}

template <int TIE>
static int run_variant(const PolicyParams& p, int launches, bool loaded, const float4* sa, float4* sb, size_t sn) {
  constexpr int NW = 8, SP = NW * 32, VLD = SP + 8;           // the eight-wave kernel of round 3 (launch_policy_nw in csrc/policy.hip)
  const size_t smem = (size_t)PRING * 8192 + ((size_t)2 * 2 * SP * 16 + (size_t)2 * 32 * VLD) * sizeof(__bf16) + (size_t)NW * 4096 +
                      (size_t)(p.pl.Gv - p.pl.v_layer0) * sizeof(float) + (size_t)(((384 + 2 * 9 * 20 + NW * 32 + 2 * SP) * 4 + 1023) & ~1023);
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(policy_kernel<NW, TIE>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  const size_t nact = (size_t)p.B * p.horizon * p.action_dim;
  std::vector<float> first(nact), cur(nact);
  hipStream_t s2;
  CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
  int differing = 0;
  for (int it = 0; it < launches; ++it) {
    if (loaded) hipLaunchKernelGGL(stream_kernel, dim3(512), dim3(256), 0, s2, sa, sb, sn);
    hipLaunchKernelGGL((policy_kernel<NW, TIE>), dim3(p.B), dim3(NW * 64), smem, 0, p);
    CK(hipMemcpy(it ? cur.data() : first.data(), p.actions, nact * 4, hipMemcpyDeviceToHost));
    if (it && memcmp(cur.data(), first.data(), nact * 4)) ++differing;
  }
  CK(hipStreamSynchronize(s2));
  CK(hipStreamDestroy(s2));
  for (float v : first)
    if (!(v == v)) { printf("  (NaN in the outputs: bad probe data)\n"); break; }
  printf("TIE %d, %s: %d of %d launches differ from the first\n", TIE, loaded ? "HBM stream beside it" : "idle chip       ", differing, launches - 1);
  fflush(stdout);
  return 0;
}

int main(int argc, char** argv) {
  const int launches = argc > 1 ? atoi(argv[1]) : 3000;
  Geom g{224, 14, 768, 12, 12, 3072, 64, 4, 4, 128, 4, 7, 5.f, 5.f, 128, 6, 4, 512, 32, 768, 1};
  const PackedLayout lay = build_layout(g);
  const PolicyLayout& pl = lay.pl;
  const int B = 64, P = g.P(), E = g.E;
  std::mt19937 rng(1234);
  std::normal_distribution<float> nd(0.f, 1.f);
  std::vector<uint16_t> wh((size_t)B * pl.Gm), wl(wh.size());
  for (size_t i = 0; i < wh.size(); ++i) {
    const float w = 0.08f * nd(rng);
    uint32_t hb = (uint32_t)f2bf(w) << 16;
    float hf;
    memcpy(&hf, &hb, 4);
    wh[i] = f2bf(w);
    wl[i] = f2bf(w - hf);
  }
  std::vector<float> vf((size_t)B * pl.Gv), tok((size_t)B * P * E);
  for (auto& v : vf) v = 0.05f * nd(rng);
  for (int b = 0; b < B; ++b)                  // LayerNorm scales around 1
    for (int l = 0; l < g.L; ++l)
      for (int i = 0; i < 64; ++i) {
        vf[(size_t)b * pl.Gv + pl.v_layer0 + l * pl.v_layer_stride + pl.v_ln0_s + i] += 1.f;
        vf[(size_t)b * pl.Gv + pl.v_layer0 + l * pl.v_layer_stride + pl.v_ln1_s + i] += 1.f;
      }
  for (auto& v : tok) v = nd(rng);
  void *dwh, *dwl, *dvf, *dtok, *dact, *dlog, *sa, *sb;
  const size_t sn = (size_t)64 << 20;          // 1 GiB in, 1 GiB out per streaming launch
  CK(hipMalloc(&dwh, wh.size() * 2)); CK(hipMalloc(&dwl, wl.size() * 2)); CK(hipMalloc(&dvf, vf.size() * 4));
  CK(hipMalloc(&dtok, tok.size() * 4)); CK(hipMalloc(&dact, (size_t)B * 28 * 4)); CK(hipMalloc(&dlog, (size_t)B * 4 * 4));
  CK(hipMalloc(&sa, sn * 16)); CK(hipMalloc(&sb, sn * 16));
  CK(hipMemcpy(dwh, wh.data(), wh.size() * 2, hipMemcpyHostToDevice));
  CK(hipMemcpy(dwl, wl.data(), wl.size() * 2, hipMemcpyHostToDevice));
  CK(hipMemcpy(dvf, vf.data(), vf.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(dtok, tok.data(), tok.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemset(sa, 0, sn * 16));
  PolicyParams p{pl, (const __bf16*)dwh, (const __bf16*)dwl, (const float*)dvf, (const float*)dtok, (float*)dact, (float*)dlog,
                 B, E, P, g.L, g.M, g.horizon, g.action_dim, g.tanh_scale, g.max_action};
  for (int loaded = 0; loaded < 2; ++loaded) {
    if (run_variant<0>(p, launches, loaded, (const float4*)sa, (float4*)sb, sn)) return 1;
    if (run_variant<1>(p, launches, loaded, (const float4*)sa, (float4*)sb, sn)) return 1;
    if (run_variant<2>(p, launches, loaded, (const float4*)sa, (float4*)sb, sn)) return 1;
    if (run_variant<3>(p, launches, loaded, (const float4*)sa, (float4*)sb, sn)) return 1;
    if (run_variant<4>(p, launches, loaded, (const float4*)sa, (float4*)sb, sn)) return 1;
  }
  return 0;
}
