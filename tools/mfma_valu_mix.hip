// Does VALU work of one wave overlap MFMA work of other waves on the same SIMD?
// Each wave loops { 32 dependent MFMA 32x32x2 ; NV independent-ish v_fma }.  Reports the time
// against the pure-MFMA and pure-VALU times.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int NM, int NV>
__global__ void __launch_bounds__(512) kmix(float* out, int iters, float a, float b) {
  f32x16 acc;
  for (int r = 0; r < 16; ++r) acc[r] = threadIdx.x * 1e-9f;
  float v[16];
  for (int r = 0; r < 16; ++r) v[r] = threadIdx.x * 1e-3f + r;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int m = 0; m < NM; ++m) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
#pragma unroll
    for (int k = 0; k < NV / 16; ++k)
#pragma unroll
      for (int r = 0; r < 16; ++r) v[r] = __builtin_fmaf(v[r], a, b);
  }
  float s = 0.f;
  for (int r = 0; r < 16; ++r) s += acc[r] + v[r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <typename K> float run(K kern, int blocks, int threads, int iters, float* d) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), 0, 0, d, iters, 1.0f, 1e-9f);
  hipEventRecord(e0);
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), 0, 0, d, iters, 1.0f, 1e-9f);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); return ms;
}
int main() {
  float* d; hipMalloc(&d, 64 << 20);
  const int it = 2000;
  for (int wg : {256, 512, 1024}) {   // x 512 threads = 2, 4, 8 waves per SIMD
    float m = run(kmix<32, 0>, wg, 512, it, d);
    float v = run(kmix<0, 320>, wg, 512, it, d);
    float x = run(kmix<32, 320>, wg, 512, it, d);
    float x2 = run(kmix<32, 160>, wg, 512, it, d);
    printf("waves/SIMD=%d: mfma-only %.3f ms, valu-only(320) %.3f ms, mixed(32 mfma + 320 valu) %.3f ms, mixed(160 valu) %.3f ms\n",
           wg * 8 / 1024, m, v, x, x2);
  }
  return 0;
}
