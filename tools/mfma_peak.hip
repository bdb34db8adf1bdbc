// Calibration micro-benchmark: sustained fp32 MFMA rate on this GPU for the instruction
// shapes / chain structures the engine uses.  hipcc --offload-arch=gfx950 -O3 tools/mfma_peak.hip -o /tmp/mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int CHAINS>
__global__ void __launch_bounds__(512) k32(float* out, int iters, float a, float b) {
  f32x16 acc[CHAINS];
  for (int c = 0; c < CHAINS; ++c) for (int r = 0; r < 16; ++r) acc[c][r] = threadIdx.x * 1e-9f;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int c = 0; c < CHAINS; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[c], 0, 0, 0);
  }
  float s = 0.f;
  for (int c = 0; c < CHAINS; ++c) for (int r = 0; r < 16; ++r) s += acc[c][r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int CHAINS>
__global__ void __launch_bounds__(512) k16(float* out, int iters, float a, float b) {
  f32x4 acc[CHAINS];
  for (int c = 0; c < CHAINS; ++c) for (int r = 0; r < 4; ++r) acc[c][r] = threadIdx.x * 1e-9f;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int c = 0; c < CHAINS; ++c) acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[c], 0, 0, 0);
  }
  float s = 0.f;
  for (int c = 0; c < CHAINS; ++c) for (int r = 0; r < 4; ++r) s += acc[c][r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <typename K>
void run(const char* name, K kern, int blocks, int threads, int iters, double flop_per_iter_per_wave, float* d) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), 0, 0, d, iters, 1.0f, 1e-9f);
  hipEventRecord(e0);
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), 0, 0, d, iters, 1.0f, 1e-9f);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  double waves = (double)blocks * threads / 64;
  printf("%-34s blocks=%d thr=%d: %.3f ms  %.1f TFLOP/s\n", name, blocks, threads, ms, waves * iters * flop_per_iter_per_wave / ms / 1e9);
}
int main() {
  float* d; hipMalloc(&d, 64 << 20);
  const int it = 20000;
  for (int wg : {256, 512, 1024, 2048}) {
    for (int thr : {256, 512}) {
      run("32x32x2 1 chain", k32<1>, wg, thr, it, 4096.0, d);
      run("32x32x2 2 chains", k32<2>, wg, thr, it, 2 * 4096.0, d);
      run("16x16x4 2 chains", k16<2>, wg, thr, it * 2, 2 * 2048.0, d);
    }
  }
  return 0;
}
