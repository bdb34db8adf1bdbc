// Training-step tail: decode + relative-L2 loss + its gradient,
// and the Adam update on the flat parameter bucket.  All HBM-bound streaming kernels.
//   decode      libs/utilities3.py:115-129   x * (std + eps) + mean
//   LpLoss.rel  libs/utilities3.py:323-334   sum_b ||x_b - y_b||_2 / ||y_b||_2   (or the mean over b)
//   Adam        run_pde_observers.py:134     torch.optim.Adam(lr, weight_decay) - L2 decay added to the gradient
#pragma once
#include "fno_dev.h"

// stat index of element e of a sample: statistics are one scalar (SL == 1) or one value per element
FNO_DEV float stat_at(const float* s, size_t e, int SL) { return s ? s[SL == 1 ? 0 : e] : 0.f; }

// partial[(b * nsplit + sp) * 2 + {0,1}] = sum over the split of (pd - td)^2 and td^2
__global__ void __launch_bounds__(256) k_lploss_partial(const float* __restrict__ pred, const float* __restrict__ tgt,
                                                        const float* __restrict__ mean, const float* __restrict__ stdv,
                                                        int SL, float eps, size_t n, float* __restrict__ partial) {
  const int b = blockIdx.y, sp = blockIdx.x, nsplit = gridDim.x;
  const float* p = pred + (size_t)b * n;
  const float* t = tgt + (size_t)b * n;
  float sd = 0.f, sy = 0.f;
  for (size_t e = (size_t)sp * blockDim.x + threadIdx.x; e < n; e += (size_t)nsplit * blockDim.x) {
    const float sc = stdv ? stat_at(stdv, e, SL) + eps : 1.0f;
    const float mu = stat_at(mean, e, SL);
    const float pd = fmaf(p[e], sc, mu), td = fmaf(t[e], sc, mu);
    const float d = pd - td;
    sd = fmaf(d, d, sd);
    sy = fmaf(td, td, sy);
  }
  for (int off = 32; off > 0; off >>= 1) { sd += __shfl_xor(sd, off, 64); sy += __shfl_xor(sy, off, 64); }
  __shared__ float sh[8];
  if ((threadIdx.x & 63) == 0) { sh[(threadIdx.x >> 6) * 2] = sd; sh[(threadIdx.x >> 6) * 2 + 1] = sy; }
  __syncthreads();
  if (threadIdx.x == 0) {
    float* o = partial + ((size_t)b * nsplit + sp) * 2;
    o[0] = (sh[0] + sh[2]) + (sh[4] + sh[6]);
    o[1] = (sh[1] + sh[3]) + (sh[5] + sh[7]);
  }
}

// one workgroup: per-sample norms in a fixed order, loss = scale * sum_b diff_b / yn_b,
// coef[b] = scale / (diff_b * yn_b)  (0 where the difference vanishes: torch.norm's subgradient)
__global__ void __launch_bounds__(256) k_lploss_finish(const float* __restrict__ partial, int B, int nsplit, float scale,
                                                       float* __restrict__ loss, float* __restrict__ coef) {
  __shared__ float sh[256];
  float acc = 0.f;
  for (int b = threadIdx.x; b < B; b += blockDim.x) {
    float sd = 0.f, sy = 0.f;
    for (int s = 0; s < nsplit; ++s) { sd += partial[((size_t)b * nsplit + s) * 2]; sy += partial[((size_t)b * nsplit + s) * 2 + 1]; }
    const float dn = sqrtf(sd), yn = sqrtf(sy);
    acc += dn / yn;
    coef[b] = dn > 0.f ? scale / (dn * yn) : 0.f;
  }
  sh[threadIdx.x] = acc;
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1) {
    if ((int)threadIdx.x < off) sh[threadIdx.x] += sh[threadIdx.x + off];
    __syncthreads();
  }
  if (threadIdx.x == 0) loss[0] = scale * sh[0];
}

// dpred[b][e] = gout * coef[b] * (pred - tgt) * (std + eps)^2     (gout: upstream scalar on the device, or NULL = 1)
__global__ void __launch_bounds__(256) k_lploss_grad(const float* __restrict__ pred, const float* __restrict__ tgt,
                                                     const float* __restrict__ stdv, int SL, float eps, size_t n,
                                                     const float* __restrict__ coef, const float* __restrict__ gout,
                                                     float* __restrict__ dpred) {
  const int b = blockIdx.y;
  const float k = coef[b] * (gout ? gout[0] : 1.0f);
  const size_t base = (size_t)b * n;
  for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (size_t)gridDim.x * blockDim.x) {
    const float sc = stdv ? stat_at(stdv, e, SL) + eps : 1.0f;
    dpred[base + e] = k * (pred[base + e] - tgt[base + e]) * sc * sc;
  }
}

// torch.optim.Adam._single_tensor_adam restated element-wise (same operation order):
//   g += wd * p;  m = lerp(m, g, 1 - b1);  v = b2 * v + (1 - b2) g g;
//   p -= (lr / bc1) * m / (sqrt(v) / sqrt(bc2) + eps)
struct AdamArgs {
  float* p; const float* g; float* m; float* v;
  size_t n;
  float lr, beta1, beta2, eps, wd, step_size, bc2_sqrt;
  const float* dyn;   // graph mode: {step_size, bc2_sqrt} written by k_adam_prep on the device, else null
};
// graph-replayable step counter: ++*step, then the two bias-correction scalars of that step (double arithmetic)
__global__ void k_adam_prep(int* step, float* dyn, float lr, float beta1, float beta2) {
  const int t = ++step[0];
  const double bc1 = 1.0 - pow((double)beta1, (double)t), bc2 = 1.0 - pow((double)beta2, (double)t);
  dyn[0] = (float)((double)lr / bc1);
  dyn[1] = (float)sqrt(bc2);
}
#ifndef FNO_ADAM_NT
#define FNO_ADAM_NT 1      // 1 = nontemporal stores of p / m / v (nothing re-reads them before the next step: 1.34 -> 1.21 ms for the
                           // full-field observer's 906 MB bucket, and the step's later kernels keep their cache lines); 0 = plain stores;
                           // 2 = + nontemporal gradient loads (measured slower: 1.30 ms)
#endif
#ifndef FNO_ADAM_UNROLL
#define FNO_ADAM_UNROLL 1
#endif
FNO_DEV void adam4(const AdamArgs& a, size_t i) {
  float4 p = ld4(a.p + 4 * i), m = ld4(a.m + 4 * i), v = ld4(a.v + 4 * i), g;
  if (FNO_ADAM_NT >= 2) {
    g.x = __builtin_nontemporal_load(a.g + 4 * i); g.y = __builtin_nontemporal_load(a.g + 4 * i + 1);
    g.z = __builtin_nontemporal_load(a.g + 4 * i + 2); g.w = __builtin_nontemporal_load(a.g + 4 * i + 3);
  } else g = ld4(a.g + 4 * i);
  float* pp = &p.x; float* gp = &g.x; float* mp = &m.x; float* vp = &v.x;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const float gg = fmaf(a.wd, pp[j], gp[j]);
    mp[j] = fmaf(gg - mp[j], 1.0f - a.beta1, mp[j]);
    vp[j] = fmaf(gg * gg, 1.0f - a.beta2, a.beta2 * vp[j]);
    const float denom = sqrtf(vp[j]) / a.bc2_sqrt + a.eps;
    pp[j] = fmaf(-a.step_size, mp[j] / denom, pp[j]);
  }
  if (FNO_ADAM_NT >= 1) {
    typedef float f4 __attribute__((ext_vector_type(4)));
    __builtin_nontemporal_store(f4{p.x, p.y, p.z, p.w}, reinterpret_cast<f4*>(a.p + 4 * i));
    __builtin_nontemporal_store(f4{m.x, m.y, m.z, m.w}, reinterpret_cast<f4*>(a.m + 4 * i));
    __builtin_nontemporal_store(f4{v.x, v.y, v.z, v.w}, reinterpret_cast<f4*>(a.v + 4 * i));
  } else {
    st4(a.p + 4 * i, p); st4(a.m + 4 * i, m); st4(a.v + 4 * i, v);
  }
}
__global__ void __launch_bounds__(256) k_adam(AdamArgs a) {
  if (a.dyn) { a.step_size = a.dyn[0]; a.bc2_sqrt = a.dyn[1]; }
  const size_t n4 = a.n / 4;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (FNO_ADAM_UNROLL == 2) {
    for (; i + stride < n4; i += 2 * stride) { adam4(a, i); adam4(a, i + stride); }
  }
  for (; i < n4; i += stride) adam4(a, i);
  if (blockIdx.x == 0 && threadIdx.x < (a.n & 3)) {
    const size_t i = n4 * 4 + threadIdx.x;
    const float gg = fmaf(a.wd, a.p[i], a.g[i]);
    const float m = fmaf(gg - a.m[i], 1.0f - a.beta1, a.m[i]);
    const float v = fmaf(gg * gg, 1.0f - a.beta2, a.beta2 * a.v[i]);
    a.m[i] = m; a.v[i] = v;
    a.p[i] = fmaf(-a.step_size, m / (sqrtf(v) / a.bc2_sqrt + a.eps), a.p[i]);
  }
}
