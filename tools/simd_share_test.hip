// What do two waves on one SIMD share?  512-thread workgroups (waves w and w + 4 sit on the same SIMD), one per CU:
// VALU stream alone (1 and 2 waves per SIMD), bf16 MFMA stream alone, and VALU beside MFMA.  Cycles per instruction from
// s_memtime around an unrolled loop.   hipcc --offload-arch=gfx950 -O3 tools/simd_share_test.hip -o tools/simd_share_test.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

// role: 0 = idle, 1 = plain VALU (v_fma_f32), 2 = packed VALU (v_pk_fma_f32), 3 = bf16 MFMA 32x32x16, 4 = v_exp_f32,
//       5 = fp32 MFMA 32x32x2
__global__ void __launch_bounds__(512) k(const int role_lo, const int role_hi, const int iters, float* out, long long* cyc) {
  const int wave = threadIdx.x >> 6;
  const int role = wave < 4 ? role_lo : role_hi;
  float v[8];
  for (int i = 0; i < 8; ++i) v[i] = threadIdx.x * 0.001f + i;
  f32x2 p[8];
  for (int i = 0; i < 8; ++i) { p[i][0] = v[i]; p[i][1] = v[i] + 0.5f; }
  f32x16 acc0, acc1;
  for (int i = 0; i < 16; ++i) { acc0[i] = 0.f; acc1[i] = 0.f; }
  bf16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (short)(0x3f80 + threadIdx.x % 7); b[i] = (short)(0x3f80 + i); }
  __syncthreads();
  const long long t0 = __builtin_readcyclecounter();
  if (role == 1) {
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int u = 0; u < 8; ++u)
#pragma unroll
        for (int i = 0; i < 8; ++i) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(v[i]) : "v"(1.0001f));
    }
  } else if (role == 2) {
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int u = 0; u < 8; ++u)
#pragma unroll
        for (int i = 0; i < 8; ++i) asm volatile("v_pk_fma_f32 %0, %0, %1, %0" : "+v"(p[i]) : "v"(p[(i + 1) & 7]));
    }
  } else if (role == 3) {
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc1, 0, 0, 0);
      }
    }
  } else if (role == 4) {
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int u = 0; u < 8; ++u)
#pragma unroll
        for (int i = 0; i < 8; ++i) asm volatile("v_exp_f32 %0, %0" : "+v"(v[i]));
    }
  } else if (role == 5) {
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(v[0], v[1], acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(v[2], v[3], acc1, 0, 0, 0);
      }
    }
  }
  const long long t1 = __builtin_readcyclecounter();
  float s = 0.f;
  for (int i = 0; i < 8; ++i) s += v[i] + p[i][0] + p[i][1];
  for (int i = 0; i < 16; ++i) s += acc0[i] + acc1[i];
  out[blockIdx.x * 512 + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
}
int main() {
  float* out; long long* cyc;
  hipMalloc(&out, 256 * 512 * 4); hipMalloc(&cyc, 256 * 8 * 8);
  const char* names[6] = {"idle", "v_fma_f32", "v_pk_fma_f32", "mfma_bf16_32x32x16", "v_exp_f32", "mfma_f32_32x32x2"};
  const int per_iter[6] = {0, 64, 64, 8, 64, 8};
  const int iters = 2000;
  const int combos[][2] = {{1, 0}, {1, 1}, {2, 0}, {2, 2}, {4, 0}, {4, 4}, {3, 0}, {3, 3}, {5, 0}, {5, 5}, {1, 3}, {2, 3}, {4, 3}, {1, 5}, {2, 5}, {3, 5}};
  for (auto& c : combos) {
    hipLaunchKernelGGL(k, dim3(256), dim3(512), 0, 0, c[0], c[1], iters, out, cyc);
    hipDeviceSynchronize();
    std::vector<long long> h(256 * 8);
    hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
    double lo = 0, hi = 0;
    for (int b = 0; b < 256; ++b) for (int w = 0; w < 8; ++w) (w < 4 ? lo : hi) += h[b * 8 + w];
    lo /= 1024; hi /= 1024;
    printf("waves 0-3: %-20s waves 4-7: %-20s | cycles per instruction: lo %.2f  hi %.2f\n", names[c[0]], names[c[1]],
           c[0] ? lo / (iters * (double)per_iter[c[0]]) : 0.0, c[1] ? hi / (iters * (double)per_iter[c[1]]) : 0.0);
  }
  return 0;
}
