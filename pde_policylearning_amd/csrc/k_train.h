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
  float omb1, omb2;   // 1 - beta in DOUBLE, then rounded (torch passes the Python double 1 - beta2 = 0.001 to addcmul_: 1.0f - 0.999f
                      // is 1.3e-5 away from it, and exp_avg_sq with it)
  const float* dyn;   // graph mode: {step_size, bc2_sqrt} written by k_adam_prep on the device, else null
};
// graph-replayable step counter: ++*step, then the two bias-correction scalars of that step (double arithmetic)
__global__ void k_adam_prep(int* step, float* dyn, double lr, double beta1, double beta2) {
  const int t = ++step[0];
  const double bc1 = 1.0 - pow(beta1, (double)t), bc2 = 1.0 - pow(beta2, (double)t);
  dyn[0] = (float)(lr / bc1);
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
    mp[j] = fmaf(gg - mp[j], a.omb1, mp[j]);
    vp[j] = fmaf(gg * gg, a.omb2, a.beta2 * vp[j]);
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
    const float m = fmaf(gg - a.m[i], a.omb1, a.m[i]);
    const float v = fmaf(gg * gg, a.omb2, a.beta2 * a.v[i]);
    a.m[i] = m; a.v[i] = v;
    a.p[i] = fmaf(-a.step_size, m / (sqrtf(v) / a.bc2_sqrt + a.eps), a.p[i]);
  }
}

// ---- row-sliced parameter blocks (dialect-C spectral weights of which only [..., :k] ever sees data) --------------------
// A block is rows x row_len floats; the first live_len floats of every row are live, the rest ("dead") receive an exactly
// zero gradient every step.  k_adam_live steps the live part only (param / grad in the full layout, moments compact:
// rows x live_len); what Adam does to a dead element is a recurrence on (p, m, v) alone (g = wd p), so it is not stepped
// but REPLAYED - k_adam_replay_dead takes it through any number of skipped steps in registers, one read and one write -
// when somebody needs the slice (checkpoint, state_dict, a longer last dimension).  Same operations in the same order
// as adam4 with g = 0: bit-identical to stepping it every time.
struct AdamLiveArgs {
  float* p; const float* g; float* m; float* v;      // p, g: full layout; m, v: compact live moments
  size_t rows; int row_len, live_len;
  float beta1, beta2, eps, wd, step_size, bc2_sqrt, omb1, omb2;
  const float* dyn;
};
__global__ void __launch_bounds__(256) k_adam_live(AdamLiveArgs a) {
  if (a.dyn) { a.step_size = a.dyn[0]; a.bc2_sqrt = a.dyn[1]; }
  const int lp = a.live_len / 2;                       // float2 (one complex number) per thread and trip
  const size_t n2 = a.rows * (size_t)lp;
  for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < n2; e += (size_t)gridDim.x * blockDim.x) {
    const size_t row = lp == 1 ? e : e / (size_t)lp;
    const size_t full = row * (size_t)a.row_len + 2 * (e - row * (size_t)lp);
    float2 p = *reinterpret_cast<const float2*>(a.p + full), g = *reinterpret_cast<const float2*>(a.g + full);
    float2 m = *reinterpret_cast<const float2*>(a.m + 2 * e), v = *reinterpret_cast<const float2*>(a.v + 2 * e);
    float* pp = &p.x; float* gp = &g.x; float* mp = &m.x; float* vp = &v.x;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const float gg = fmaf(a.wd, pp[j], gp[j]);
      mp[j] = fmaf(gg - mp[j], a.omb1, mp[j]);
      vp[j] = fmaf(gg * gg, a.omb2, a.beta2 * vp[j]);
      const float denom = sqrtf(vp[j]) / a.bc2_sqrt + a.eps;
      pp[j] = fmaf(-a.step_size, mp[j] / denom, pp[j]);
    }
    *reinterpret_cast<float2*>(a.p + full) = p;
    *reinterpret_cast<float2*>(a.m + 2 * e) = m;
    *reinterpret_cast<float2*>(a.v + 2 * e) = v;
  }
}
// scal[2 j] = step_size, scal[2 j + 1] = sqrt(bias correction 2) of step step_from + j (k_adam_prep's arithmetic)
__global__ void k_adam_replay_prep(float* scal, int step_from, int nsteps, double lr, double beta1, double beta2) {
  for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < nsteps; j += gridDim.x * blockDim.x) {
    const int t = step_from + j;
    const double bc1 = 1.0 - pow(beta1, (double)t), bc2 = 1.0 - pow(beta2, (double)t);
    scal[2 * j] = (float)(lr / bc1);
    scal[2 * j + 1] = (float)sqrt(bc2);
  }
}
struct AdamReplayArgs {
  float* p;                 // full layout
  float* dm; float* dv;     // compact dead moments (rows x (row_len - live_len)): read unless moments_zero, always written
  size_t rows; int row_len, live_len, moments_zero;
  const float* scal; int nsteps;
  float beta1, beta2, eps, wd, omb1, omb2;
};
__global__ void __launch_bounds__(256) k_adam_replay_dead(AdamReplayArgs a) {
  const int dl = a.row_len - a.live_len;
  const size_t n = a.rows * (size_t)dl;
  for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (size_t)gridDim.x * blockDim.x) {
    const size_t row = e / (size_t)dl;
    const size_t full = row * (size_t)a.row_len + a.live_len + (e - row * (size_t)dl);
    float p = a.p[full], m = a.moments_zero ? 0.f : a.dm[e], v = a.moments_zero ? 0.f : a.dv[e];
    for (int j = 0; j < a.nsteps; ++j) {
      const float step_size = a.scal[2 * j], bc2_sqrt = a.scal[2 * j + 1];      // (uniform: scalar loads)
      const float gg = fmaf(a.wd, p, 0.0f);
      m = fmaf(gg - m, a.omb1, m);
      v = fmaf(gg * gg, a.omb2, a.beta2 * v);
      const float denom = sqrtf(v) / bc2_sqrt + a.eps;
      p = fmaf(-step_size, m / denom, p);
    }
    a.p[full] = p; a.dm[e] = m; a.dv[e] = v;
  }
}
