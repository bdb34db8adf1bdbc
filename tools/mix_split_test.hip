// The two-term split's low term through v_fma_mixlo_f16 / v_fma_mixhi_f16 (fno_dev.h: split2_low) against the compiler's form
// (v_cvt_f32_f16, v_sub_f32, v_cvt_pk_f16_f32), bit for bit: (1) 2^23 random values over the exponent range 2^-37 .. 2^22 (fp16
// denormals, overflow to infinity) plus zeros, alone on the chip; (2) the same comparison in tester waves whose SIMD partners
// issue fp16 MFMAs back to back (the situation of every kernel that uses it; the packed-fp32 op_sel forms of tools/pk_opsel_hazard.hip
// fail exactly there).
//   hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize -o tools/mix_split_test.bin tools/mix_split_test.hip && tools/mix_split_test.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "../pde_policylearning_amd/csrc/fno_dev.h"
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

__global__ void k_alone(const float* x, unsigned* out, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const f32x2 v = {x[2 * i], x[2 * i + 1]};
  const f16x2 h = __builtin_convertvector(v, f16x2);
  const f16x2 ref = __builtin_convertvector(v - __builtin_convertvector(h, f32x2), f16x2);
  out[2 * i] = __builtin_bit_cast(unsigned, ref);
  out[2 * i + 1] = __builtin_bit_cast(unsigned, split2_low(v, h));
}
__global__ void __launch_bounds__(512, 2) k_beside_mfma(unsigned* out, int iters) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (wave < 4) {
    unsigned seed = (blockIdx.x * 512u + tid) * 2654435761u + 12345u, bad = 0;
    for (int it = 0; it < iters; ++it) {
      seed = seed * 1664525u + 1013904223u;
      const unsigned ua = (seed & 0x807fffffu) | ((100u + (seed >> 24) % 44u) << 23);
      seed = seed * 1664525u + 1013904223u;
      const unsigned ub = (seed & 0x807fffffu) | ((100u + (seed >> 24) % 44u) << 23);
      f32x2 v = {__builtin_bit_cast(float, ua), __builtin_bit_cast(float, ub)};
      asm volatile("" : "+v"(v));
      const f16x2 h = __builtin_convertvector(v, f16x2);
      const f16x2 ref = __builtin_convertvector(v - __builtin_convertvector(h, f32x2), f16x2);
      const f16x2 got = split2_low(v, h);
      bad += __builtin_bit_cast(unsigned, ref) != __builtin_bit_cast(unsigned, got);
    }
    if (bad) atomicAdd(&out[0], bad);
  } else {
    f32x16 acc;
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    f16x8 fa, fb;
    for (int j = 0; j < 8; ++j) { fa[j] = (_Float16)(lane * 0.01f + j); fb[j] = (_Float16)(j * 0.5f - lane * 0.02f); }
    for (int it = 0; it < iters / 4; ++it) {
#pragma unroll
      for (int q = 0; q < 8; ++q) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa, fb, acc, 0, 0, 0);
    }
    if (acc[0] + acc[5] == 123.456f) out[1] = 1;
  }
}
int main() {
  const int n = 1 << 22;
  std::vector<float> x(2 * n);
  srand(1);
  for (auto& v : x) { unsigned u = ((unsigned)rand() << 16) ^ (unsigned)rand(); u = (u & 0x807fffffu) | ((unsigned)(90 + rand() % 60) << 23); memcpy(&v, &u, 4); }
  for (int i = 0; i < 64; ++i) x[i] = (i % 8 == 0) ? 0.f : ldexpf(1.f + i * 0.013f, -20 - i % 12);
  float* dx; unsigned* dout;
  CK(hipMalloc(&dx, (size_t)2 * n * 4)); CK(hipMalloc(&dout, (size_t)2 * n * 4));
  CK(hipMemcpy(dx, x.data(), (size_t)2 * n * 4, hipMemcpyHostToDevice));
  k_alone<<<n / 256, 256>>>(dx, dout, n);
  std::vector<unsigned> o(2 * n);
  CK(hipMemcpy(o.data(), dout, (size_t)2 * n * 4, hipMemcpyDeviceToHost));
  size_t bad = 0;
  for (int i = 0; i < n; ++i) bad += o[2 * i] != o[2 * i + 1];
  printf("alone: %d pairs, %zu different\n", n, bad);
  CK(hipMemset(dout, 0, 64));
  hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
  const int iters = 20000;
  k_beside_mfma<<<2 * prop.multiProcessorCount, 512>>>(dout, iters);
  unsigned h[2];
  CK(hipMemcpy(h, dout, 8, hipMemcpyDeviceToHost));
  printf("beside MFMA partners: %.3g pairs, %u different\n", (double)2 * prop.multiProcessorCount * 256 * iters, h[0]);
  return bad || h[0];
}
