// PINO residual loss for planes that do not fit one CU's LDS as a full complex grid (256 x 256: the grid BASELINE
// config 5 names): the same arithmetic as k_pino_loss.h (libs/envs/diff_control_env.py:5-60) with the 2-D transforms
// split into a row pass, a column pass and a row pass that hand complex slabs to each other through HBM.
//
//   forward    rows_fwd : w(x, :) real           -> FFT along y             -> S1[plane][x][q]
//              cols_fwd : S1[:, q-block]         -> FFT along x, x M_f, inverse FFT along x (f = 0..4) -> S2[f][plane][x][q]
//              rows_inv : S2[f][x-block, :]      -> inverse FFT along y, real part -> u_x, w_x, u_y, w_y, lap w -> residual
//   backward   rows_bwd : g . partner_f real     -> FFT along y             -> S2[f][plane][x][q]
//              cols_bwd : S2[f][:, q-block]      -> FFT along x, sum_f conj(M_f) ., inverse FFT along x -> S1
//              rows_out : S1[x-block, :]         -> inverse FFT along y, real part -> dws
//
// As in the single-workgroup kernels the forward transforms are decimation-in-frequency (natural -> bit-reversed) and
// the inverse ones decimation-in-time (bit-reversed -> natural): slab position (p, q) carries the mode
// (brev(p), brev(q)), which is where the multipliers are evaluated; no reordering pass exists.
// A slab block is 32 rows (or 32 columns) of one plane: 66 KB of LDS, two workgroups per CU.  Planes are processed in
// chunks (host side) so that S1 / S2 stay bounded (64 planes: 34 MB + 168 MB at 256 x 256).
#pragma once
#include "k_pino_loss.h"

// N-point FFTs of `nlines` lines by NWV waves; a line is handled by min(64, N/2) lanes of ONE wave (lanes take N/128
// butterflies per stage when N > 128), so the stages of a line only need the wave-level LDS ordering.
template <int N, bool INVERSE, int NWV>
FNO_DEV void fft_lines_n(float2* G, int nlines, int line_stride, int elem_stride, const float2* tw, int wave, int lane) {
  constexpr int BPL = N / 2;
  constexpr int LANES = BPL < 64 ? BPL : 64;
  constexpr int LW = 64 / LANES;
  constexpr int BPT = BPL / LANES;
  const int jl = lane % LANES, sub = lane / LANES;
  for (int line = wave * LW + sub; line < nlines; line += NWV * LW) {
    float2* L = G + line * line_stride;
    if (!INVERSE) {
#pragma unroll
      for (int h = N / 2; h >= 1; h >>= 1) {
#pragma unroll
        for (int bb = 0; bb < BPT; ++bb) {
          const int j = jl + LANES * bb;
          const int pos = j & (h - 1), i0 = ((j - pos) << 1) + pos, i1 = i0 + h;
          const float2 a = L[i0 * elem_stride], b = L[i1 * elem_stride];
          const float2 w = tw[pos * (N / 2 / h)];
          L[i0 * elem_stride] = make_float2(a.x + b.x, a.y + b.y);
          L[i1 * elem_stride] = cmulf(make_float2(a.x - b.x, a.y - b.y), w);
        }
        lds_wave_sync();
      }
    } else {
#pragma unroll
      for (int h = 1; h <= N / 2; h <<= 1) {
#pragma unroll
        for (int bb = 0; bb < BPT; ++bb) {
          const int j = jl + LANES * bb;
          const int pos = j & (h - 1), i0 = ((j - pos) << 1) + pos, i1 = i0 + h;
          float2 w = tw[pos * (N / 2 / h)];
          w.y = -w.y;
          const float2 a = L[i0 * elem_stride], b = cmulf(L[i1 * elem_stride], w);
          L[i0 * elem_stride] = make_float2(a.x + b.x, a.y + b.y);
          L[i1 * elem_stride] = make_float2(a.x - b.x, a.y - b.y);
        }
        lds_wave_sync();
      }
    }
  }
}

constexpr int PINO2_RB = 32;     // rows (columns) of a plane per workgroup
constexpr int PINO2_NT = 512;

struct Pino2Args {
  PinoArgs p;          // u, forcing, visc, fields, dws, partial (PINO2 parts per plane), coef_f, g_f, B, T, inv2dt
  float2* s1;          // (chunk planes, N, N)
  float2* s2;          // (5, chunk planes, N, N)
  int plane0;          // first plane of this chunk
};

template <int N>
FNO_DEV void pino2_twiddles(float2* tw, int tid) {
  for (int k = tid; k < N / 2; k += PINO2_NT) {
    float sn, cs;
    sincospif(-2.0f * (float)k / (float)N, &sn, &cs);
    tw[k] = make_float2(cs, sn);
  }
}

// ---- forward -------------------------------------------------------------------------------------------------
template <int N>
__global__ void __launch_bounds__(PINO2_NT) k_pino2_rows_fwd(Pino2Args a) {
  constexpr int RB = PINO2_RB, P = N + 1, NT = PINO2_NT, NWV = NT / 64, PPT = RB * N / NT;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float2* G = reinterpret_cast<float2*>(smem);
  float2* tw = G + RB * P;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int x0 = blockIdx.x * RB, pl = blockIdx.y, plane = a.plane0 + pl;
  const int b = plane / (a.p.T - 2), t = plane % (a.p.T - 2) + 1;
  pino2_twiddles<N>(tw, tid);
#pragma unroll
  for (int j = 0; j < PPT; ++j) {
    const int e = tid + NT * j, xl = e / N, y = e % N;
    G[xl * P + y] = make_float2(a.p.u[(((size_t)b * N + x0 + xl) * N + y) * a.p.T + t], 0.f);
  }
  __syncthreads();
  fft_lines_n<N, false, NWV>(G, RB, P, 1, tw, wave, lane);
  __syncthreads();
#pragma unroll
  for (int j = 0; j < PPT; ++j) {
    const int e = tid + NT * j, xl = e / N, q = e % N;
    a.s1[((size_t)pl * N + x0 + xl) * N + q] = G[xl * P + q];
  }
}

template <int N>
__global__ void __launch_bounds__(PINO2_NT) k_pino2_cols_fwd(Pino2Args a) {
  constexpr int CB = PINO2_RB, PC = CB + 1, NT = PINO2_NT, NWV = NT / 64, PPT = N * CB / NT;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float2* G = reinterpret_cast<float2*>(smem);
  float2* tw = G + N * PC;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int q0 = blockIdx.x * CB, pl = blockIdx.y;
  const size_t chunk = (size_t)gridDim.y * N * N;
  pino2_twiddles<N>(tw, tid);
#pragma unroll
  for (int j = 0; j < PPT; ++j) {
    const int e = tid + NT * j, x = e / CB, ql = e % CB;
    G[x * PC + ql] = a.s1[((size_t)pl * N + x) * N + q0 + ql];
  }
  __syncthreads();
  fft_lines_n<N, false, NWV>(G, CB, 1, PC, tw, wave, lane);
  __syncthreads();
  float wre[PPT], wim[PPT];
#pragma unroll
  for (int j = 0; j < PPT; ++j) {
    const int e = tid + NT * j;
    const float2 v = G[(e / CB) * PC + e % CB];
    wre[j] = v.x; wim[j] = v.y;
  }
#pragma unroll 1
  for (int f = 0; f < 5; ++f) {
    __syncthreads();
#pragma unroll
    for (int j = 0; j < PPT; ++j) {
      const int e = tid + NT * j, p = e / CB, ql = e % CB;
      const float2 m = pino_mult<N>(f, brev_n<N>(p), brev_n<N>(q0 + ql));
      G[p * PC + ql] = cmulf(m, make_float2(wre[j], wim[j]));
    }
    __syncthreads();
    fft_lines_n<N, true, NWV>(G, CB, 1, PC, tw, wave, lane);
    __syncthreads();
#pragma unroll
    for (int j = 0; j < PPT; ++j) {
      const int e = tid + NT * j, x = e / CB, ql = e % CB;
      a.s2[(size_t)f * chunk + ((size_t)pl * N + x) * N + q0 + ql] = G[x * PC + ql];
    }
  }
}

template <int N>
__global__ void __launch_bounds__(PINO2_NT) k_pino2_rows_inv(Pino2Args a) {
  constexpr int RB = PINO2_RB, P = N + 1, NT = PINO2_NT, NWV = NT / 64, PPT = RB * N / NT;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float2* G = reinterpret_cast<float2*>(smem);
  float2* tw = G + RB * P;
  float* red = reinterpret_cast<float*>(tw + N / 2);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int x0 = blockIdx.x * RB, pl = blockIdx.y, plane = a.plane0 + pl;
  const int b = plane / (a.p.T - 2), t = plane % (a.p.T - 2) + 1;
  const size_t np = (size_t)a.p.B * (a.p.T - 2) * N * N;
  const size_t chunk = (size_t)gridDim.y * N * N;
  pino2_twiddles<N>(tw, tid);
  float acc[PPT], keep[PPT];
#pragma unroll
  for (int j = 0; j < PPT; ++j) {
    const int e = tid + NT * j, xl = e / N, y = e % N;
    const float* up = a.p.u + (((size_t)b * N + x0 + xl) * N + y) * a.p.T + t;
    acc[j] = (up[1] - up[-1]) * a.p.inv2dt;                   // w_t, central difference (diff_control_env.py:38-39)
  }
  const float nu = a.p.visc[b];
  const float inv_n2 = 1.0f / (float)(N * N);
#pragma unroll 1
  for (int f = 0; f < 5; ++f) {
    __syncthreads();
#pragma unroll
    for (int j = 0; j < PPT; ++j) {
      const int e = tid + NT * j, xl = e / N, q = e % N;
      G[xl * P + q] = a.s2[(size_t)f * chunk + ((size_t)pl * N + x0 + xl) * N + q];
    }
    __syncthreads();
    fft_lines_n<N, true, NWV>(G, RB, P, 1, tw, wave, lane);
    __syncthreads();
#pragma unroll
    for (int j = 0; j < PPT; ++j) {
      const int e = tid + NT * j, xl = e / N, y = e % N;
      const float val = G[xl * P + y].x * inv_n2;
      if (f < 4) a.p.fields[(size_t)f * np + ((size_t)plane * N + x0 + xl) * N + y] = val;
      if (f == 0 || f == 2) keep[j] = val;                 // u_x, u_y
      else if (f == 1 || f == 3) acc[j] = fmaf(keep[j], val, acc[j]);   // + u_x w_x, + u_y w_y
      else acc[j] = fmaf(-nu, val, acc[j]);                // - nu lap(w)
    }
  }
  float ss = 0.f;
#pragma unroll
  for (int j = 0; j < PPT; ++j) {
    const int e = tid + NT * j, xl = e / N, y = e % N;
    const float r = acc[j] - a.p.forcing[(x0 + xl) * N + y];
    a.p.fields[(size_t)4 * np + ((size_t)plane * N + x0 + xl) * N + y] = r;
    ss = fmaf(r, r, ss);
  }
  for (int off = 32; off > 0; off >>= 1) ss += __shfl_xor(ss, off, 64);
  if (lane == 0) red[wave] = ss;
  __syncthreads();
  if (tid == 0) {
    float tot = 0.f;
#pragma unroll
    for (int k = 0; k < NWV; ++k) tot += red[k];
    a.p.partial[(size_t)plane * gridDim.x + blockIdx.x] = tot;
  }
}

// ---- backward ------------------------------------------------------------------------------------------------
template <int N>
__global__ void __launch_bounds__(PINO2_NT) k_pino2_rows_bwd(Pino2Args a) {
  constexpr int RB = PINO2_RB, P = N + 1, NT = PINO2_NT, NWV = NT / 64, PPT = RB * N / NT;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float2* G = reinterpret_cast<float2*>(smem);
  float2* tw = G + RB * P;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int x0 = blockIdx.x * RB, pl = blockIdx.y, plane = a.plane0 + pl;
  const int b = plane / (a.p.T - 2);
  const size_t np = (size_t)a.p.B * (a.p.T - 2) * N * N;
  const size_t chunk = (size_t)gridDim.y * N * N;
  pino2_twiddles<N>(tw, tid);
  const float gs = a.p.coef_f[b] * (a.p.g_f ? a.p.g_f[0] : 1.0f);
  const float nu = a.p.visc[b];
  float g[PPT];
#pragma unroll
  for (int j = 0; j < PPT; ++j) {
    const int e = tid + NT * j;
    g[j] = gs * a.p.fields[(size_t)4 * np + ((size_t)plane * N + x0) * N + e];
  }
#pragma unroll 1
  for (int f = 0; f < 5; ++f) {
    __syncthreads();
    const int partner = f ^ 1;       // dL/dfield_f: u_x <- g w_x, w_x <- g u_x, u_y <- g w_y, w_y <- g u_y, lap w <- -nu g
#pragma unroll
    for (int j = 0; j < PPT; ++j) {
      const int e = tid + NT * j;
      const float gy = f < 4 ? g[j] * a.p.fields[(size_t)partner * np + ((size_t)plane * N + x0) * N + e] : -nu * g[j];
      G[(e / N) * P + e % N] = make_float2(gy, 0.f);
    }
    __syncthreads();
    fft_lines_n<N, false, NWV>(G, RB, P, 1, tw, wave, lane);
    __syncthreads();
#pragma unroll
    for (int j = 0; j < PPT; ++j) {
      const int e = tid + NT * j, xl = e / N, q = e % N;
      a.s2[(size_t)f * chunk + ((size_t)pl * N + x0 + xl) * N + q] = G[xl * P + q];
    }
  }
}

template <int N>
__global__ void __launch_bounds__(PINO2_NT) k_pino2_cols_bwd(Pino2Args a) {
  constexpr int CB = PINO2_RB, PC = CB + 1, NT = PINO2_NT, NWV = NT / 64, PPT = N * CB / NT;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float2* G = reinterpret_cast<float2*>(smem);
  float2* tw = G + N * PC;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int q0 = blockIdx.x * CB, pl = blockIdx.y;
  const size_t chunk = (size_t)gridDim.y * N * N;
  pino2_twiddles<N>(tw, tid);
  float dre[PPT], dim_[PPT];
#pragma unroll
  for (int j = 0; j < PPT; ++j) { dre[j] = 0.f; dim_[j] = 0.f; }
#pragma unroll 1
  for (int f = 0; f < 5; ++f) {
    __syncthreads();
#pragma unroll
    for (int j = 0; j < PPT; ++j) {
      const int e = tid + NT * j, x = e / CB, ql = e % CB;
      G[x * PC + ql] = a.s2[(size_t)f * chunk + ((size_t)pl * N + x) * N + q0 + ql];
    }
    __syncthreads();
    fft_lines_n<N, false, NWV>(G, CB, 1, PC, tw, wave, lane);
    __syncthreads();
#pragma unroll
    for (int j = 0; j < PPT; ++j) {
      const int e = tid + NT * j, p = e / CB, ql = e % CB;
      float2 m = pino_mult<N>(f, brev_n<N>(p), brev_n<N>(q0 + ql));
      m.y = -m.y;
      const float2 v = cmulf(m, G[p * PC + ql]);
      dre[j] += v.x; dim_[j] += v.y;
    }
  }
  __syncthreads();
#pragma unroll
  for (int j = 0; j < PPT; ++j) {
    const int e = tid + NT * j;
    G[(e / CB) * PC + e % CB] = make_float2(dre[j], dim_[j]);
  }
  __syncthreads();
  fft_lines_n<N, true, NWV>(G, CB, 1, PC, tw, wave, lane);
  __syncthreads();
#pragma unroll
  for (int j = 0; j < PPT; ++j) {
    const int e = tid + NT * j, x = e / CB, ql = e % CB;
    a.s1[((size_t)pl * N + x) * N + q0 + ql] = G[x * PC + ql];
  }
}

template <int N>
__global__ void __launch_bounds__(PINO2_NT) k_pino2_rows_out(Pino2Args a) {
  constexpr int RB = PINO2_RB, P = N + 1, NT = PINO2_NT, NWV = NT / 64, PPT = RB * N / NT;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float2* G = reinterpret_cast<float2*>(smem);
  float2* tw = G + RB * P;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int x0 = blockIdx.x * RB, pl = blockIdx.y, plane = a.plane0 + pl;
  pino2_twiddles<N>(tw, tid);
#pragma unroll
  for (int j = 0; j < PPT; ++j) {
    const int e = tid + NT * j, xl = e / N, q = e % N;
    G[xl * P + q] = a.s1[((size_t)pl * N + x0 + xl) * N + q];
  }
  __syncthreads();
  fft_lines_n<N, true, NWV>(G, RB, P, 1, tw, wave, lane);
  __syncthreads();
  const float inv_n2 = 1.0f / (float)(N * N);
#pragma unroll
  for (int j = 0; j < PPT; ++j) {
    const int e = tid + NT * j;
    a.p.dws[((size_t)plane * N + x0) * N + e] = G[(e / N) * P + e % N].x * inv_n2;
  }
}
