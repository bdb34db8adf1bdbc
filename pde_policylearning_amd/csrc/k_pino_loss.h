// PINO residual loss: the spectral Navier-Stokes vorticity residual
//   FDM_NS_vorticity + Channelflow_PINO_loss   libs/envs/diff_control_env.py:5-60
//   (== libs/pino_utils/losses.py:68-104, 246-262), called by train_pino.py:98-101
// forward and backward, fused per (sample, time level) plane.
//
// One workgroup owns one N x N plane of the (B, N, N, T) field and keeps it in LDS as a complex grid
// (rows padded to N + 1 float2).  All transforms are in-LDS radix-2 FFTs, one line per wave pass:
// forward = decimation in frequency (natural -> bit-reversed), inverse = decimation in time
// (bit-reversed -> natural), so no reordering pass exists; the spectral multipliers are evaluated at the
// bit-reversed positions.  The reference's `irfft2(spec[:, :, :n/2+1])` equals
// Re(ifft2(Hermitian extension of those columns)); on the full grid that extension is the multiplier
// conj(M(-kx, -ky)) for column indices above n/2, which is what `pino_mult` returns there.
// The spectrum w_h of the plane stays in registers (2 floats x N*N/512 per thread) while the five derived
// fields (u_x, w_x, u_y, w_y, lap w) are produced one after the other through the single LDS grid.
#pragma once
#include "fno_dev.h"

FNO_DEV float2 cmulf(float2 a, float2 b) { return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }
FNO_DEV void lds_wave_sync() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
// The thread index as a value the optimiser cannot see through: what is derived from it INSIDE the loop over the five derived
// fields (grid offsets, bit-reversed wave numbers, the multiplier's kx / ky / lap of 16 items per thread at 128 x 128) is
// then recomputed per field instead of being hoisted out of the loop and kept live across it - that hoisting is what spilled
// 248 / 572 bytes per lane at the 128 registers a 16-wave workgroup gets (round 4 register table).
FNO_DEV int opaque_tid(int tid) { asm volatile("" : "+v"(tid)); return tid; }

// N-point FFTs of the N lines `G + line * line_stride` (elements `elem_stride` apart) by NWV waves.
// tw[k] = exp(-2 pi i k / N), k < N/2.
template <int N, bool INVERSE, int NWV>
FNO_DEV void fft_lines(float2* G, int line_stride, int elem_stride, const float2* tw, int wave, int lane) {
  constexpr int LPL = N / 2;            // butterflies (lanes) per line
  constexpr int LW = 64 / LPL;          // lines per wave pass
  static_assert(LW >= 1, "N <= 128");
  const int sub = lane / LPL;
  // (N = 128: the butterfly offsets of all seven stages are the same for every line; hoisted out of the line loop they hold ~25
  // registers per direction for the whole kernel, which has 128 per lane - the lane's butterfly index is made opaque per line so that
  // they are recomputed per line: a handful of integer operations beside four LDS accesses per stage)
  const int jl = lane % LPL;
  for (int line = wave * LW + sub; line < N; line += NWV * LW) {
    float2* L = G + line * line_stride;
    int j = jl;
    if (N >= 128) asm volatile("" : "+v"(j));
    if (!INVERSE) {
#pragma unroll
      for (int h = N / 2; h >= 1; h >>= 1) {
        const int pos = j & (h - 1), i0 = ((j - pos) << 1) + pos, i1 = i0 + h;
        const float2 a = L[i0 * elem_stride], b = L[i1 * elem_stride];
        const float2 w = tw[pos * (N / 2 / h)];
        L[i0 * elem_stride] = make_float2(a.x + b.x, a.y + b.y);
        L[i1 * elem_stride] = cmulf(make_float2(a.x - b.x, a.y - b.y), w);
        lds_wave_sync();
      }
    } else {
#pragma unroll
      for (int h = 1; h <= N / 2; h <<= 1) {
        const int pos = j & (h - 1), i0 = ((j - pos) << 1) + pos, i1 = i0 + h;
        float2 w = tw[pos * (N / 2 / h)];
        w.y = -w.y;
        const float2 a = L[i0 * elem_stride], b = cmulf(L[i1 * elem_stride], w);
        L[i0 * elem_stride] = make_float2(a.x + b.x, a.y + b.y);
        L[i1 * elem_stride] = make_float2(a.x - b.x, a.y - b.y);
        lds_wave_sync();
      }
    }
  }
}
template <int N, bool INVERSE, int NWV>
FNO_DEV void fft2_grid(float2* G, const float2* tw, int wave, int lane) {
  constexpr int P = N + 1;
  fft_lines<N, INVERSE, NWV>(G, P, 1, tw, wave, lane);     // along y (rows)
  __syncthreads();
  fft_lines<N, INVERSE, NWV>(G, 1, P, tw, wave, lane);     // along x (columns)
  __syncthreads();
}
// threads per plane: 16 waves for 128 x 128 (16 pixels and 16 spectrum values per thread stay in registers)
template <int N> struct PinoCfg { static constexpr int NT = N == 128 ? 1024 : 512; };

// Spectral multiplier of derived field f at full-grid index (ix, iy)  (diff_control_env.py:15-30):
//   0: u_x = i ky / lap   1: w_x = i kx   2: u_y = -i kx / lap   3: w_y = i ky   4: lap w = -lap
// k(index) = index < n/2 ? index : index - n (index n/2 carries -n/2); lap(0, 0) := 1.
// Columns iy > n/2 carry the Hermitian extension conj(M(-ix, -iy)).
template <int N>
FNO_DEV float2 pino_mult(int f, int ix, int iy) {
  const bool ext = iy > N / 2;
  const int jx = ext ? ((N - ix) & (N - 1)) : ix;
  const int jy = ext ? N - iy : iy;
  const float kx = (float)(jx < N / 2 ? jx : jx - N);
  const float ky = (float)(jy < N / 2 ? jy : jy - N);
  float lap = kx * kx + ky * ky;
  if (jx == 0 && jy == 0) lap = 1.0f;
  // branch-free in f (uniform selects): the switch over f inside the fully unrolled per-thread loops was unswitched into ~400
  // basic blocks per kernel.  Same values: the numerator is exactly +-kx or ky, the quotient the same division.
  const float cx = f == 1 ? 1.f : (f == 2 ? -1.f : 0.f), cy = (f == 0 || f == 3) ? 1.f : 0.f;      // (uniform: scalar registers)
  const float num = fmaf(cx, kx, cy * ky);                       // exactly +-kx, ky or 0: one of the coefficients is zero
  const float im = num / ((f == 0 || f == 2) ? lap : 1.0f);      // (x / 1 is x: no branch around the division)
  const float re = f == 4 ? -lap : 0.f;
  return make_float2(re, ext ? -im : im);
}
template <int N>
FNO_DEV int brev_n(int v) {
  constexpr int LOG = N == 256 ? 8 : N == 128 ? 7 : N == 64 ? 6 : 5;
  return (int)(__brev((unsigned)v) >> (32 - LOG));
}

struct PinoArgs {
  const float* u;        // (B, N, N, T)
  const float* forcing;  // (N, N)
  const float* visc;     // (B)
  float* fields;         // 5 arrays of (B*(T-2), N, N): u_x, w_x, u_y, w_y, residual
  float* dws;            // backward: (B*(T-2), N, N) spectral part of dL/du
  float* partial;        // forward: (B*(T-2)) sums of residual^2
  const float* coef_f;   // backward: (B) = 1 / (B ||r_b|| ||f_b||)
  const float* g_f;      // backward: upstream gradient of loss_f (device scalar) or null
  int B, T;
  float inv2dt;
};

template <int N>
__global__ void __launch_bounds__(PinoCfg<N>::NT) k_pino_plane_fwd(PinoArgs a) {
  constexpr int P = N + 1, NT = PinoCfg<N>::NT, NWV = NT / 64, PPT = N * N / NT;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float2* G = reinterpret_cast<float2*>(smem);
  float2* tw = G + N * P;
  float* red = reinterpret_cast<float*>(tw + N / 2);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int plane = blockIdx.x, b = plane / (a.T - 2), t = plane % (a.T - 2) + 1;
  const size_t np = (size_t)a.B * (a.T - 2) * N * N;
  for (int k = tid; k < N / 2; k += NT) {
    float sn, cs;
    sincospif(-2.0f * (float)k / (float)N, &sn, &cs);      // exact argument (k / N is dyadic)
    tw[k] = make_float2(cs, sn);
  }
  float acc[PPT], wre[PPT], wim[PPT];
#pragma unroll
  for (int j = 0; j < PPT; ++j) {
    const int e = tid + NT * j, x = e / N, y = e % N;
    const float* up = a.u + (((size_t)b * N + x) * N + y) * a.T + t;
    G[x * P + y] = make_float2(up[0], 0.f);
    acc[j] = (up[1] - up[-1]) * a.inv2dt;                   // w_t, central difference (diff_control_env.py:38-39)
  }
  __syncthreads();
  fft2_grid<N, false, NWV>(G, tw, wave, lane);
#pragma unroll
  for (int j = 0; j < PPT; ++j) {
    const int e = tid + NT * j;
    const float2 v = G[(e / N) * P + e % N];
    wre[j] = v.x; wim[j] = v.y;
  }
  const float nu = a.visc[b];
  const float inv_n2 = 1.0f / (float)(N * N);
#pragma unroll 1
  for (int f = 0; f < 5; ++f) {
    __syncthreads();
    const int tf = opaque_tid(tid);
#pragma unroll
    for (int j = 0; j < PPT; ++j) {
      const int e = tf + NT * j, p = e / N, q = e % N;
      const float2 m = pino_mult<N>(f, brev_n<N>(p), brev_n<N>(q));
      G[p * P + q] = cmulf(m, make_float2(wre[j], wim[j]));
      if ((j & 3) == 3) __builtin_amdgcn_sched_barrier(0);      // four items' temporaries in flight, not sixteen
    }
    __syncthreads();
    fft2_grid<N, true, NWV>(G, tw, wave, lane);
#pragma unroll
    for (int j = 0; j < PPT; ++j) {
      const int e = tf + NT * j;
      const float val = G[(e / N) * P + e % N].x * inv_n2;
      float* fo = a.fields + (size_t)plane * N * N + e;
      if (f < 4) fo[(size_t)f * np] = val;
      // u_x / u_y of the step before come back from the field array this thread has just written (a register array carried
      // across the loop became a 16-wide vector value the allocator spilled wholesale: 1144 bytes per lane)
      if (f == 1 || f == 3) acc[j] = fmaf(fo[(size_t)(f - 1) * np], val, acc[j]);   // + u_x w_x, + u_y w_y
      else if (f == 4) acc[j] = fmaf(-nu, val, acc[j]);    // - nu lap(w)
    }
  }
  float ss = 0.f;
#pragma unroll
  for (int j = 0; j < PPT; ++j) {
    const int e = tid + NT * j;
    const float r = acc[j] - a.forcing[e];
    a.fields[(size_t)4 * np + (size_t)plane * N * N + e] = r;
    ss = fmaf(r, r, ss);
  }
  for (int off = 32; off > 0; off >>= 1) ss += __shfl_xor(ss, off, 64);
  if (lane == 0) red[wave] = ss;
  __syncthreads();
  if (tid == 0) {
    float tot = 0.f;
#pragma unroll
    for (int k = 0; k < NWV; ++k) tot += red[k];
    a.partial[plane] = tot;
  }
}

// dL/dw (spectral part) of one plane: g = dL/dDu; the five real-linear maps W -> field_f have the
// adjoint  Re(ifft2(conj(M_f) . fft2(.))) / n^2, so their sum costs five forward FFTs and one inverse.
template <int N>
__global__ void __launch_bounds__(PinoCfg<N>::NT) k_pino_plane_bwd(PinoArgs a) {
  constexpr int P = N + 1, NT = PinoCfg<N>::NT, NWV = NT / 64, PPT = N * N / NT;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float2* G = reinterpret_cast<float2*>(smem);
  float2* tw = G + N * P;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int plane = blockIdx.x, b = plane / (a.T - 2);
  const size_t np = (size_t)a.B * (a.T - 2) * N * N;
  for (int k = tid; k < N / 2; k += NT) {
    float sn, cs;
    sincospif(-2.0f * (float)k / (float)N, &sn, &cs);      // exact argument (k / N is dyadic)
    tw[k] = make_float2(cs, sn);
  }
  const float gs = a.coef_f[b] * (a.g_f ? a.g_f[0] : 1.0f);
  const float nu = a.visc[b];
  // (g = gs * residual is re-read with every field's partner instead of living in 16 more registers per thread across the
  // loop: the 128 x 128 instantiation has 128 registers per lane and spilled exactly this array)
  float dre[PPT], dim_[PPT];
#pragma unroll
  for (int j = 0; j < PPT; ++j) { dre[j] = 0.f; dim_[j] = 0.f; }
#pragma unroll 1
  for (int f = 0; f < 5; ++f) {
    __syncthreads();
    // dL/dfield_f: u_x <- g w_x, w_x <- g u_x, u_y <- g w_y, w_y <- g u_y, lap w <- -nu g
    const int partner = f ^ 1;
    const int tf = opaque_tid(tid);
#pragma unroll
    for (int j = 0; j < PPT; ++j) {
      const int e = tf + NT * j;
      const float* fp = a.fields + (size_t)plane * N * N + e;
      const float gj = gs * fp[(size_t)4 * np];
      const float pv = fp[(size_t)(partner & 3) * np];      // (always a load, of a valid field: no branch per item; unused at f = 4)
      const float gy = gj * (f < 4 ? pv : -nu);
      G[(e / N) * P + e % N] = make_float2(gy, 0.f);
      if ((j & 3) == 3) __builtin_amdgcn_sched_barrier(0);
    }
    __syncthreads();
    fft2_grid<N, false, NWV>(G, tw, wave, lane);
#pragma unroll
    for (int j = 0; j < PPT; ++j) {
      const int e = tf + NT * j, p = e / N, q = e % N;
      float2 m = pino_mult<N>(f, brev_n<N>(p), brev_n<N>(q));
      m.y = -m.y;
      const float2 v = cmulf(m, G[p * P + q]);
      dre[j] += v.x; dim_[j] += v.y;
      if ((j & 3) == 3) __builtin_amdgcn_sched_barrier(0);      // four items' temporaries in flight, not sixteen
    }
  }
  __syncthreads();
#pragma unroll
  for (int j = 0; j < PPT; ++j) {
    const int e = tid + NT * j;
    G[(e / N) * P + e % N] = make_float2(dre[j], dim_[j]);
  }
  __syncthreads();
  fft2_grid<N, true, NWV>(G, tw, wave, lane);
  const float inv_n2 = 1.0f / (float)(N * N);
#pragma unroll
  for (int j = 0; j < PPT; ++j) {
    const int e = tid + NT * j;
    a.dws[(size_t)plane * N * N + e] = G[(e / N) * P + e % N].x * inv_n2;
  }
}

// initial-condition term: partial[(b*S + s)*2 + {0,1}] = sum (u[b,:,:,0] - u0)^2, sum u0^2   (diff_control_env.py:53-54)
__global__ void __launch_bounds__(256) k_pino_ic_partial(const float* __restrict__ u, const float* __restrict__ u0, int nn, int T,
                                                         float* __restrict__ partial) {
  const int b = blockIdx.y, S = gridDim.x;
  float sd = 0.f, sy = 0.f;
  for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < nn; e += S * blockDim.x) {
    const float y = u0[(size_t)b * nn + e];
    const float d = u[((size_t)b * nn + e) * T] - y;
    sd = fmaf(d, d, sd); sy = fmaf(y, y, sy);
  }
  for (int off = 32; off > 0; off >>= 1) { sd += __shfl_xor(sd, off, 64); sy += __shfl_xor(sy, off, 64); }
  __shared__ float sh[8];
  if ((threadIdx.x & 63) == 0) { sh[(threadIdx.x >> 6) * 2] = sd; sh[(threadIdx.x >> 6) * 2 + 1] = sy; }
  __syncthreads();
  if (threadIdx.x == 0) {
    partial[((size_t)b * S + blockIdx.x) * 2] = (sh[0] + sh[2]) + (sh[4] + sh[6]);
    partial[((size_t)b * S + blockIdx.x) * 2 + 1] = (sh[1] + sh[3]) + (sh[5] + sh[7]);
  }
}

// one workgroup: LpLoss(size_average=True).rel of both terms and the per-sample gradient coefficients
//   loss_f = mean_b ||Du_b - f|| / ||f repeated over T-2||,  loss_ic = mean_b ||u_b(t=0) - u0_b|| / ||u0_b||
__global__ void __launch_bounds__(256) k_pino_finish(const float* __restrict__ part_f, const float* __restrict__ part_ic,
                                                     const float* __restrict__ forcing, int B, int T, int nn, int S,
                                                     int PP /* partial sums per plane in part_f */,
                                                     float* __restrict__ loss_ic, float* __restrict__ loss_f,
                                                     float* __restrict__ coef_ic, float* __restrict__ coef_f) {
  __shared__ float sh[256];
  float s = 0.f;
  for (int e = threadIdx.x; e < nn; e += blockDim.x) s = fmaf(forcing[e], forcing[e], s);
  sh[threadIdx.x] = s;
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1) {
    if ((int)threadIdx.x < off) sh[threadIdx.x] += sh[threadIdx.x + off];
    __syncthreads();
  }
  const float fnorm = sqrtf(sh[0] * (float)(T - 2));
  __syncthreads();
  if (threadIdx.x == 0) {
    float lf = 0.f, lic = 0.f;
    for (int b = 0; b < B; ++b) {
      float sf = 0.f;
      for (int k = 0; k < (T - 2) * PP; ++k) sf += part_f[(size_t)b * (T - 2) * PP + k];
      float sd = 0.f, sy = 0.f;
      for (int k = 0; k < S; ++k) { sd += part_ic[((size_t)b * S + k) * 2]; sy += part_ic[((size_t)b * S + k) * 2 + 1]; }
      const float rn = sqrtf(sf), dn = sqrtf(sd), yn = sqrtf(sy);
      lf += rn / fnorm;
      lic += dn / yn;
      coef_f[b] = rn > 0.f ? 1.0f / ((float)B * rn * fnorm) : 0.f;
      coef_ic[b] = dn > 0.f ? 1.0f / ((float)B * dn * yn) : 0.f;
    }
    loss_f[0] = lf / (float)B;
    loss_ic[0] = lic / (float)B;
  }
}

// du[b,x,y,t] = spectral part (interior t) + adjoint of the central time difference + initial-condition term
__global__ void __launch_bounds__(256) k_pino_assemble(const float* __restrict__ u, const float* __restrict__ u0,
                                                       const float* __restrict__ dws, const float* __restrict__ resid,
                                                       const float* __restrict__ coef_f, const float* __restrict__ coef_ic,
                                                       const float* __restrict__ g_f, const float* __restrict__ g_ic, int B,
                                                       int nn, int T, float inv2dt, float* __restrict__ du) {
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (size_t)B * nn) return;
  const int b = (int)(idx / nn), e = (int)(idx % nn);
  const float gf = coef_f[b] * (g_f ? g_f[0] : 1.0f);
  const float gi = coef_ic[b] * (g_ic ? g_ic[0] : 1.0f);
  const size_t pl = (size_t)b * (T - 2);
  float* o = du + idx * T;
  float gprev = 0.f;                                         // g at level t - 1
  float gcur = 0.f;                                          // g at level t
  for (int t = 0; t < T; ++t) {
    const float gnext = (t + 1 >= 1 && t + 1 <= T - 2) ? gf * resid[(pl + t) * nn + e] : 0.f;   // level t + 1 -> plane index t
    float v = (gprev - gnext) * inv2dt;
    if (t >= 1 && t <= T - 2) v += dws[(pl + t - 1) * nn + e];
    if (t == 0) v += gi * (u[idx * T] - u0[idx]);
    o[t] = v;
    gprev = gcur;
    gcur = gnext;
  }
}
