// Diagnostic (GPU box): what v_permlane16_swap_b32 / v_permlane32_swap_b32 do on gfx950, and what hipcc 7.2 makes of the builtins.
//   hipcc --offload-arch=gfx950 -O3 tools/permlane_swap_probe.hip -o /tmp/psp && /tmp/psp
// Finding (round 5): the instruction swaps the odd 16-lane rows of its first operand with the even rows of its second (32-lane form:
// the upper half of the first with the lower half of the second), as documented -- but __builtin_amdgcn_permlane16_swap(a, b, 0, 0)
// returns the FIRST operand's new value in BOTH elements of its pair (the ISA stores the same register twice).  Issued by inline asm
// both registers come back right, and a reduce-scatter of 32 values per lane over the 32 lanes of a wave half built on it
// (attention_kernel's column sums) adds up exactly.
#include <hip/hip_runtime.h>
#include <cstdio>
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
__global__ void swaps(float* out) {
  const int lane = threadIdx.x;
  float a = lane, b = 100 + lane;
  asm volatile("" : "+v"(a), "+v"(b));
  const auto sw = __builtin_amdgcn_permlane16_swap(__builtin_bit_cast(unsigned, a), __builtin_bit_cast(unsigned, b), false, false);
  out[lane] = __builtin_bit_cast(float, sw[0]);
  out[64 + lane] = __builtin_bit_cast(float, sw[1]);
  float c = a, d = b;
  asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(c), "+v"(d));
  out[128 + lane] = c;
  out[192 + lane] = d;
}
// every lane: 32 values (item j of lane l = 1000 j + (l & 31)); lane l must end with the sum of item l & 31 over its half
__global__ void reduce_scatter(const float* in, float* out) {
  const int lane = threadIdx.x;
  float O0[16], O1[16], w16[16];
  for (int j = 0; j < 16; ++j) O0[j] = in[lane * 32 + j], O1[j] = in[lane * 32 + 16 + j];
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    float a = O0[j], b = O1[j];
    asm("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));
    w16[j] = a + b;
  }
  const bool b3 = lane & 8, b2 = lane & 4, b1 = lane & 2, b0 = lane & 1;
  float w8[8], w4[4], w2[2];
#pragma unroll
  for (int j = 0; j < 8; ++j) w8[j] = (b3 ? w16[j + 8] : w16[j]) + dpp_mov<0x140>(b3 ? w16[j] : w16[j + 8]);
#pragma unroll
  for (int j = 0; j < 4; ++j) w4[j] = (b2 ? w8[j + 4] : w8[j]) + dpp_mov<0x141>(b2 ? w8[j] : w8[j + 4]);
#pragma unroll
  for (int j = 0; j < 2; ++j) w2[j] = (b1 ? w4[j + 2] : w4[j]) + dpp_mov<0x4E>(b1 ? w4[j] : w4[j + 2]);
  out[lane] = (b0 ? w2[1] : w2[0]) + dpp_mov<0xB1>(b0 ? w2[0] : w2[1]);
}
int main() {
  float *o, h[256];
  (void)hipMalloc(&o, sizeof h);
  swaps<<<1, 64>>>(o);
  (void)hipMemcpy(h, o, sizeof h, hipMemcpyDeviceToHost);
  const char* n[4] = {"builtin [0]", "builtin [1]", "asm, first operand", "asm, second operand"};
  printf("operands: a = lane, b = 100 + lane; lanes 0, 15, 16, 31, 32, 47, 48, 63 afterwards\n");
  for (int r = 0; r < 4; ++r) {
    printf("%-20s", n[r]);
    for (int l = 0; l < 64; ++l) if (l % 16 == 0 || l % 16 == 15) printf(" %4.0f", h[r * 64 + l]);
    printf("\n");
  }
  static float in[64 * 32], res[64];
  for (int l = 0; l < 64; ++l) for (int j = 0; j < 32; ++j) in[l * 32 + j] = (float)(j * 1000 + (l & 31));
  float *d;
  (void)hipMalloc(&d, sizeof in);
  (void)hipMemcpy(d, in, sizeof in, hipMemcpyHostToDevice);
  reduce_scatter<<<1, 64>>>(d, o);
  (void)hipMemcpy(res, o, sizeof res, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int l = 0; l < 64; ++l) bad += res[l] != 32000.f * (l & 31) + 496.f;
  printf("reduce-scatter over the lane bits (swap, row mirror, half-row mirror, two quad permutations): %d of 64 lanes wrong\n", bad);
  return bad != 0;
}
