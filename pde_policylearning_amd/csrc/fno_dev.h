// Device-side helpers shared by the fnoengine HIP kernels (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define FNO_DEV __device__ __forceinline__

// v_mfma_f32_32x32x2_f32: A[i=l&31][k=l>>5], B[k=l>>5][j=l&31];
// D: col = l&31, row = (r&3) + 8*(r>>2) + 4*(l>>5)   (exact fp32 FMA chain)
FNO_DEV f32x16 mfma32(float a, float b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}
// v_mfma_f32_16x16x4_f32: A[i=l&15][k=l>>4], B[k=l>>4][j=l&15];
// D: col = l&15, row = (l>>4)*4 + r
FNO_DEV f32x4 mfma16(float a, float b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}
// row index inside a 32x32 accumulator tile held by lane-half `half`, register r
FNO_DEV int acc_row32(int r, int half) { return (r & 3) + 8 * (r >> 2) + 4 * half; }

// exact-erf GELU (torch F.gelu default) and its derivative
FNO_DEV float gelu_f(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }
FNO_DEV float gelu_grad_f(float x) {
  const float cdf = 0.5f * (1.0f + erff(x * 0.70710678118654752440f));
  const float pdf = 0.39894228040143267794f * expf(-0.5f * x * x);
  return cdf + x * pdf;
}

FNO_DEV float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
FNO_DEV void st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }
