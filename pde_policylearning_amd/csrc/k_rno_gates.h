// GRU-style gates of the recurrent neural operator cell (neuralop/models/rno.py:254-260), fused:
//   z  = sigmoid(f1(x) + f2(h) + b1)      z2 = sigmoid(f7(x) + f8(h) + b4)
//   r  = sigmoid(f3(x) + f4(h) + b2)      h^ = selu(f5(x) + f6(r h) + b3)
//   h' = (1 - z) h + z2 h^
// The eight f_i are Fourier layers (fused engine layers); what is left between them is streaming
// elementwise work on (B, C, X, Y) tensors with four SCALAR biases.  Two kernels forward (reset gate,
// output gate), two backward; every tensor is read / written once with 16-B accesses and the scalar-bias
// gradients are reduced per workgroup IN DOUBLE (33 M terms of either sign at BASELINE config 3: float32 partial sums cost
// the four scalars 3 - 14x the error of the rest of the gradient; the adds hide behind the memory traffic), partials
// (double) summed by the caller in a fixed order.
#pragma once
#include "fno_dev.h"

FNO_DEV float sigmoid_f(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
// torch.nn.functional.selu constants
#define SELU_ALPHA 1.6732632423543772848170429916717f
#define SELU_SCALE 1.0507009873554804934193349852946f

// r = sigmoid(a3 + a4 + b2), rh = r * h
__global__ void __launch_bounds__(256) k_rno_reset_fwd(const float4* __restrict__ a3, const float4* __restrict__ a4,
                                                       const float* __restrict__ b2, const float4* __restrict__ h,
                                                       float4* __restrict__ r, float4* __restrict__ rh, size_t n4) {
  const float b = b2[0];
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
    const float4 p = a3[i], q = a4[i], hv = h[i];
    float4 rv;
    rv.x = sigmoid_f(p.x + q.x + b); rv.y = sigmoid_f(p.y + q.y + b);
    rv.z = sigmoid_f(p.z + q.z + b); rv.w = sigmoid_f(p.w + q.w + b);
    r[i] = rv;
    rh[i] = make_float4(rv.x * hv.x, rv.y * hv.y, rv.z * hv.z, rv.w * hv.w);
  }
}
// given d(rh): ds = d(rh) h r (1 - r) (gradient of a3, a4 and, summed, of b2); dh = d(rh) r
__global__ void __launch_bounds__(256) k_rno_reset_bwd(const float4* __restrict__ drh, const float4* __restrict__ r,
                                                       const float4* __restrict__ h, float4* __restrict__ ds,
                                                       float4* __restrict__ dh, double* __restrict__ db_part, size_t n4) {
  double acc = 0.0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
    const float4 g = drh[i], rv = r[i], hv = h[i];
    float4 s;
    s.x = g.x * hv.x * rv.x * (1.f - rv.x); s.y = g.y * hv.y * rv.y * (1.f - rv.y);
    s.z = g.z * hv.z * rv.z * (1.f - rv.z); s.w = g.w * hv.w * rv.w * (1.f - rv.w);
    ds[i] = s;
    dh[i] = make_float4(g.x * rv.x, g.y * rv.y, g.z * rv.z, g.w * rv.w);
    acc += (double)((s.x + s.y) + (s.z + s.w));
  }
  for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 64);
  __shared__ double sh[4];
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) db_part[blockIdx.x] = (sh[0] + sh[1]) + (sh[2] + sh[3]);
}

struct RnoOutArgs {
  const float4 *a1, *a2, *a7, *a8, *a5, *a6, *h;   // Fourier-layer outputs and the previous state
  const float *b1, *b4, *b3;                      // scalar biases (device)
  float4 *z, *z2, *s3, *hn;                       // saved gates, pre-SELU sum, new state
  size_t n4;
};
__global__ void __launch_bounds__(256) k_rno_out_fwd(RnoOutArgs a) {
  const float b1 = a.b1[0], b4 = a.b4[0], b3 = a.b3[0];
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < a.n4; i += (size_t)gridDim.x * blockDim.x) {
    const float4 p1 = a.a1[i], p2 = a.a2[i], p7 = a.a7[i], p8 = a.a8[i], p5 = a.a5[i], p6 = a.a6[i], hv = a.h[i];
    float4 z, z2, s3, hn;
    const float* q1 = &p1.x; const float* q2 = &p2.x; const float* q7 = &p7.x; const float* q8 = &p8.x;
    const float* q5 = &p5.x; const float* q6 = &p6.x; const float* qh = &hv.x;
    float* oz = &z.x; float* oz2 = &z2.x; float* os = &s3.x; float* oh = &hn.x;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      oz[j] = sigmoid_f(q1[j] + q2[j] + b1);
      oz2[j] = sigmoid_f(q7[j] + q8[j] + b4);
      os[j] = q5[j] + q6[j] + b3;
      const float hh = SELU_SCALE * (os[j] > 0.f ? os[j] : SELU_ALPHA * (__expf(os[j]) - 1.0f));
      oh[j] = (1.0f - oz[j]) * qh[j] + oz2[j] * hh;
    }
    a.z[i] = z; a.z2[i] = z2; a.s3[i] = s3; a.hn[i] = hn;
  }
}
struct RnoOutBwdArgs {
  const float4 *g, *z, *z2, *s3, *h;
  float4 *ds1, *ds7, *ds3, *dh;     // gradients of (a1, a2), (a7, a8), (a5, a6) and the direct path to h
  double* db_part;                  // [3][gridDim]: b1, b4, b3 partial sums
  size_t n4;
};
__global__ void __launch_bounds__(256) k_rno_out_bwd(RnoOutBwdArgs a) {
  double acc1 = 0.0, acc7 = 0.0, acc3 = 0.0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < a.n4; i += (size_t)gridDim.x * blockDim.x) {
    const float4 gv = a.g[i], zv = a.z[i], z2v = a.z2[i], sv = a.s3[i], hv = a.h[i];
    float4 d1, d7, d3, dh;
    const float* g = &gv.x; const float* z = &zv.x; const float* z2 = &z2v.x; const float* s = &sv.x; const float* h = &hv.x;
    float* o1 = &d1.x; float* o7 = &d7.x; float* o3 = &d3.x; float* oh = &dh.x;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float e = s[j] > 0.f ? 0.f : SELU_ALPHA * __expf(s[j]);          // alpha exp(s) on the negative branch
      const float hh = SELU_SCALE * (s[j] > 0.f ? s[j] : e - SELU_ALPHA);
      const float dsel = SELU_SCALE * (s[j] > 0.f ? 1.0f : e);
      o1[j] = -g[j] * h[j] * z[j] * (1.0f - z[j]);
      o7[j] = g[j] * hh * z2[j] * (1.0f - z2[j]);
      o3[j] = g[j] * z2[j] * dsel;
      oh[j] = g[j] * (1.0f - z[j]);
    }
    acc1 += (double)((o1[0] + o1[1]) + (o1[2] + o1[3]));
    acc7 += (double)((o7[0] + o7[1]) + (o7[2] + o7[3]));
    acc3 += (double)((o3[0] + o3[1]) + (o3[2] + o3[3]));
    a.ds1[i] = d1; a.ds7[i] = d7; a.ds3[i] = d3; a.dh[i] = dh;
  }
  for (int off = 32; off > 0; off >>= 1) {
    acc1 += __shfl_xor(acc1, off, 64); acc7 += __shfl_xor(acc7, off, 64); acc3 += __shfl_xor(acc3, off, 64);
  }
  __shared__ double sh[12];
  if ((threadIdx.x & 63) == 0) { sh[threadIdx.x >> 6] = acc1; sh[4 + (threadIdx.x >> 6)] = acc7; sh[8 + (threadIdx.x >> 6)] = acc3; }
  __syncthreads();
  if (threadIdx.x == 0) {
    a.db_part[blockIdx.x] = (sh[0] + sh[1]) + (sh[2] + sh[3]);
    a.db_part[gridDim.x + blockIdx.x] = (sh[4] + sh[5]) + (sh[6] + sh[7]);
    a.db_part[2 * gridDim.x + blockIdx.x] = (sh[8] + sh[9]) + (sh[10] + sh[11]);
  }
}
