// Verifies (a) the bf16 32x32x16 MFMA fragment layout assumed by the engine and (b) the accuracy of the
// 3-term bf16 split GEMM (6 products, fp32 accumulate) against an fp64 host reference, next to the
// fp32 MFMA 32x32x2 result.   hipcc --offload-arch=gfx950 -O3 tools/mfma_bf16x3_test.hip -o /tmp/t && /tmp/t
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));

__device__ inline unsigned short f2bf(float x) {   // round-to-nearest-even (finite inputs)
  unsigned u = __builtin_bit_cast(unsigned, x);
  u += 0x7FFFu + ((u >> 16) & 1u);
  return (unsigned short)(u >> 16);
}
__device__ inline float bf2f(unsigned short h) { return __builtin_bit_cast(float, (unsigned)h << 16); }
__device__ inline void split3(float x, unsigned short& h, unsigned short& m, unsigned short& l) {
  h = f2bf(x); float r = x - bf2f(h);
  m = f2bf(r); r -= bf2f(m);
  l = f2bf(r);
}

// C[32][32] = A[32][K] * B[K][32], one wave; A row-major, B row-major
__global__ void k_test(const float* A, const float* B, float* C3, float* C32, int K) {
  const int lane = threadIdx.x, r = lane & 31, hh = lane >> 5;
  f32x16 acc, accf;
  for (int i = 0; i < 16; ++i) { acc[i] = 0.f; accf[i] = 0.f; }
  for (int kb = 0; kb < K / 16; ++kb) {
    bf16x8 a[3], b[3];
    for (int j = 0; j < 8; ++j) {
      const int k = kb * 16 + 8 * hh + j;
      unsigned short h, m, l;
      split3(A[r * K + k], h, m, l); a[0][j] = h; a[1][j] = m; a[2][j] = l;
      split3(B[k * 32 + r], h, m, l); b[0][j] = h; b[1][j] = m; b[2][j] = l;
    }
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2], b[0], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[2], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[1], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[0], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[1], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[0], acc, 0, 0, 0);
  }
  for (int s = 0; s < K / 2; ++s)
    accf = __builtin_amdgcn_mfma_f32_32x32x2f32(A[r * K + 2 * s + hh], B[(2 * s + hh) * 32 + r], accf, 0, 0, 0);
  for (int i = 0; i < 16; ++i) {
    const int row = (i & 3) + 8 * (i >> 2) + 4 * hh;
    C3[row * 32 + r] = acc[i];
    C32[row * 32 + r] = accf[i];
  }
}
int main() {
  for (int K : {16, 64, 256}) {
    std::vector<float> A(32 * K), B(K * 32);
    srand(1);
    for (auto& v : A) v = (rand() / (float)RAND_MAX - 0.5f) * 0.6f;
    for (auto& v : B) v = (rand() / (float)RAND_MAX - 0.5f) * 2.f;
    float *dA, *dB, *d3, *d32;
    hipMalloc(&dA, A.size() * 4); hipMalloc(&dB, B.size() * 4); hipMalloc(&d3, 4096); hipMalloc(&d32, 4096);
    hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_test, dim3(1), dim3(64), 0, 0, dA, dB, d3, d32, K);
    std::vector<float> C3(1024), C32(1024);
    hipMemcpy(C3.data(), d3, 4096, hipMemcpyDeviceToHost);
    hipMemcpy(C32.data(), d32, 4096, hipMemcpyDeviceToHost);
    double e3 = 0, e32 = 0, nrm = 0;
    for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) {
      double ref = 0; for (int k = 0; k < K; ++k) ref += (double)A[i * K + k] * B[k * 32 + j];
      e3 += (C3[i * 32 + j] - ref) * (C3[i * 32 + j] - ref);
      e32 += (C32[i * 32 + j] - ref) * (C32[i * 32 + j] - ref);
      nrm += ref * ref;
    }
    printf("K=%3d: rel-L2 error  bf16x3(6 products) %.3e   fp32 MFMA %.3e\n", K, sqrt(e3 / nrm), sqrt(e32 / nrm));
  }
  return 0;
}
