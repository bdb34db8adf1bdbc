// How does the vector-instruction rate of one SIMD scale with the number of resident waves (1, 2, 3, 4 per SIMD), alone and
// beside fp16 MFMA streams?  256-thread workgroups (one wave per SIMD each), W workgroups per CU; in every workgroup waves
// run `role` (1 v_fma_f32, 2 v_pk_fma_f32, 4 v_exp_f32, 6 a GELU-like mix), and M of the W workgroups run fp16 MFMAs
// instead.  Prints vector instructions per cycle per SIMD.   hipcc --offload-arch=gfx950 -O3 tools/occupancy_valu_test.hip -o tools/occupancy_valu_test.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

__global__ void __launch_bounds__(256) k(int role, int nmfma, int W, int iters, float* out, long long* cyc) {
  const bool mf = (int)(blockIdx.x / 256) < nmfma;      // blocks b, b + 256, .. share a CU under round-robin dispatch (speed only)
  float v[8];
  for (int i = 0; i < 8; ++i) v[i] = threadIdx.x * 0.001f + i;
  f32x2 p[8];
  for (int i = 0; i < 8; ++i) { p[i][0] = v[i]; p[i][1] = v[i] + 0.5f; }
  f32x16 acc0 = {0}, acc1 = {0};
  f16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(1 + threadIdx.x % 7); b[i] = (_Float16)(1 + i); }
  __syncthreads();
  const long long t0 = __builtin_readcyclecounter();
  if (mf) {
    for (int it = 0; it < iters; ++it)
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc1, 0, 0, 0);
      }
  } else if (role == 1) {
    for (int it = 0; it < iters; ++it)
#pragma unroll
      for (int u = 0; u < 8; ++u)
#pragma unroll
        for (int i = 0; i < 8; ++i) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(v[i]) : "v"(1.0001f));
  } else if (role == 2) {
    for (int it = 0; it < iters; ++it)
#pragma unroll
      for (int u = 0; u < 8; ++u)
#pragma unroll
        for (int i = 0; i < 8; ++i) asm volatile("v_pk_fma_f32 %0, %0, %1, %0" : "+v"(p[i]) : "v"(p[(i + 1) & 7]));
  } else if (role == 4) {
    for (int it = 0; it < iters; ++it)
#pragma unroll
      for (int u = 0; u < 8; ++u)
#pragma unroll
        for (int i = 0; i < 8; ++i) asm volatile("v_exp_f32 %0, %0" : "+v"(v[i]));
  } else {      // the packed GELU's mix per pair: 7 v_pk_fma, 4 v_med3, 2 v_exp, 2 v_fma  (x 4 pairs = 60 instructions)
    for (int it = 0; it < iters; ++it)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
#pragma unroll
        for (int q = 0; q < 7; ++q) asm volatile("v_pk_fma_f32 %0, %0, %1, %0" : "+v"(p[i]) : "v"(p[i + 4]));
        asm volatile("v_med3_f32 %0, %0, 0, %1\n\tv_med3_f32 %2, %2, 0, %1" : "+v"(v[i]), "+v"(v[i + 4]) : "v"(6.0f));
        asm volatile("v_med3_f32 %0, %0, 0, %1\n\tv_med3_f32 %2, %2, 0, %1" : "+v"(v[i]), "+v"(v[i + 4]) : "v"(6.0f));
        asm volatile("v_exp_f32 %0, %0\n\tv_exp_f32 %1, %1" : "+v"(v[i]), "+v"(v[i + 4]));
        asm volatile("v_fma_f32 %0, %0, %1, %0\n\tv_fma_f32 %2, %2, %1, %2" : "+v"(v[i]), "+v"(v[i + 4]) : "v"(1.0001f));
      }
  }
  const long long t1 = __builtin_readcyclecounter();
  float s = 0.f;
  for (int i = 0; i < 8; ++i) s += v[i] + p[i][0] + p[i][1];
  for (int i = 0; i < 16; ++i) s += acc0[i] + acc1[i];
  out[(size_t)blockIdx.x * 256 + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}
int main() {
  float* out; long long* cyc;
  const int maxw = 8;
  if (hipMalloc(&out, (size_t)256 * maxw * 256 * 4) != hipSuccess || hipMalloc(&cyc, 256 * maxw * 4 * 8) != hipSuccess) return 1;
  const char* names[7] = {"", "v_fma_f32", "v_pk_fma_f32", "", "v_exp_f32", "", "GELU mix"};
  const int per_iter[7] = {0, 64, 64, 0, 64, 0, 60};
  const int iters = 4000;
  for (int role : {1, 2, 4, 6})
    for (int nm : {0, 1, 2})
      for (int W : {1, 2, 3, 4, 6, 8}) {
        if (nm >= W) continue;
        hipLaunchKernelGGL(k, dim3(256 * W), dim3(256), 0, 0, role, nm, W, iters, out, cyc);
        if (hipDeviceSynchronize() != hipSuccess) return 1;
        std::vector<long long> h(256 * W * 4);
        hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
        double vs = 0, ms = 0;
        for (int b = 0; b < 256 * W; ++b) for (int w = 0; w < 4; ++w) (b / 256 < nm ? ms : vs) += h[b * 4 + w];
        const int nv = (W - nm) * 1024, nmw = nm * 1024;
        const double cpi = vs / nv / (iters * (double)per_iter[role]);
        printf("%-13s %d VALU + %d MFMA waves per SIMD: %.2f cycles per vector instruction and wave -> %.3f instructions / cycle / SIMD%s\n",
               names[role], W - nm, nm, cpi, (W - nm) / cpi, nm ? "" : "");
        if (nm) printf("%13s    MFMA waves: %.1f cycles per MFMA\n", "", ms / nmw / (iters * 8.0));
      }
  return 0;
}
