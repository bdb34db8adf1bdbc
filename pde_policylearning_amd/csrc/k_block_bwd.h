// Backward of one fused FNO block (the adjoint of k_pw_fwd with CIN == COUT == C).
//
// Inputs per tile: g = dL/du_{l+1} and the block's own pre-activation input u_l.
//   dx[i][px]  = sum_o W[o][i] g[o][px] + sum_j Zg[b,row,j,i] * Tinv[j][w]      (MFMA 32x32x2)
//   gout       = dx * gelu'(u_l)            (when the block input was activated)
//   dW[o][i]  += sum_px g[o][px] * a_l[i][px],  a_l = gelu(u_l) or u_l           (MFMA, K = pixels)
//   dbias[o]  += sum_px g[o][px]
//   X1g        = truncated row DFT of gout  (MFMA 16x16x4)  -> feeds the spectral
//                backward of block l-1;   or, for block 0, the lifting gradients
//                dWl[c][i] = sum_px gout[c][px] x_in[i][px], dbl[c] = sum_px gout[c][px].
// Weight/bias gradients are accumulated in registers across the persistent
// tile loop and written once per workgroup as partial slabs (summed by
// k_reduce_slabs: deterministic, no float atomics).
//
// Reference semantics: autograd of fno_block.py:123-170 + spectral_convolution.py:303-347
// (formulas: SURVEY.md Appendix A, oracle/fno_oracle.py::spectral_conv_A_backward).
#pragma once
#include "fno_dev.h"

struct BlkBwdArgs {
  const float* g;      // (B, C, PW)
  const float* uin;    // (B, C, PW)
  const float* w;      // (C, C) [o][i]
  const float* zg;     // (B, P, K2in, C, 2) or null
  const float* tinv;   // (2*K2in, W)
  float* gout;         // (B, C, PW) or null
  float* x1g;          // (B, P, K2out, C, 2) or null
  const float* tfwd;   // (16*NJ, W)
  float* dw_part;      // (gridDim * KSPLIT, C, C)
  float* db_part;      // (gridDim, C)
  const float* xin;    // (B, CL, PW) or null   (block 0: lifting input)
  float* dwl_part;     // (gridDim, C, 16): col i < CL = dWl, col CL = dbl
  int CL;
  int PW, W, P, K2in, K2out, NJ;
  int act_in;
  int tiles_per_plane, ntiles;
};

template <int C, int NPX>
__global__ void __launch_bounds__(NPX * 2) k_block_bwd(BlkBwdArgs a) {
  constexpr int NW = NPX / 32;
  constexpr int NT = NW * 64;
  constexpr int MT = C / 32;
  constexpr int KS = C / 2;
  constexpr int PITCH = NPX + 4;
  constexpr int TILES = MT * MT;                       // 32x32 tiles of dW
  constexpr int KSPLIT = (NW >= TILES) ? NW / TILES : 1;
  constexpr int TPW = (TILES + NW - 1) / NW;           // dW tiles per wave when NW < TILES
  constexpr int LJ = (C / 16 + NW - 1) / NW;           // lifting-epilogue jobs per wave
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* gs = smem;                 // C x PITCH
  float* us = smem + C * PITCH;     // C x PITCH
  float* xls = us + C * PITCH;      // 4 x PITCH (lifting input rows)

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, half = lane >> 5;
  const int l15 = lane & 15, quad = lane >> 4;

  // A fragments of W^T: A[i][k = o] = W[o][i]
  float afrag[MT][KS];
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int s = 0; s < KS; ++s) afrag[m][s] = a.w[(2 * s + half) * C + m * 32 + l31];

  f32x16 dwacc[TPW];
#pragma unroll
  for (int t = 0; t < TPW; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) dwacc[t][r] = 0.0f;
  float dbacc[C / 8];
#pragma unroll
  for (int i = 0; i < C / 8; ++i) dbacc[i] = 0.0f;
  f32x4 dl[LJ];
#pragma unroll
  for (int j = 0; j < LJ; ++j) dl[j] = f32x4{0.f, 0.f, 0.f, 0.f};

  for (int tile = blockIdx.x; tile < a.ntiles; tile += gridDim.x) {
    const int b = tile / a.tiles_per_plane;
    const int px0 = (tile % a.tiles_per_plane) * NPX;
    const float* gb = a.g + (size_t)b * C * a.PW + px0;
    const float* ub = a.uin + (size_t)b * C * a.PW + px0;
#pragma unroll
    for (int i = 0; i < C / 8; ++i) {
      const int idx = tid + i * NT;
      const int c = idx / (NPX / 4), q = idx % (NPX / 4);
      const float4 gv = ld4(gb + (size_t)c * a.PW + 4 * q);
      const float4 uv = ld4(ub + (size_t)c * a.PW + 4 * q);
      dbacc[i] += (gv.x + gv.y) + (gv.z + gv.w);
      st4(gs + c * PITCH + 4 * q, gv);
      st4(us + c * PITCH + 4 * q, uv);
    }
    if (a.xin) {
      const float* xb = a.xin + (size_t)b * a.CL * a.PW + px0;
      for (int idx = tid; idx < a.CL * (NPX / 4); idx += NT) {
        const int c = idx / (NPX / 4), q = idx % (NPX / 4);
        st4(xls + c * PITCH + 4 * q, ld4(xb + (size_t)c * a.PW + 4 * q));
      }
    }
    __syncthreads();

    // ---- dx GEMM (+ row inverse DFT of the spectral gradient) -------------
    f32x16 acc[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[m][r] = 0.0f;
    const int n0 = wave * 32;
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      const float bf = gs[(2 * s + half) * PITCH + n0 + l31];
#pragma unroll
      for (int m = 0; m < MT; ++m) acc[m] = mfma32(afrag[m][s], bf, acc[m]);
    }
    if (a.zg) {
      const int prow = (px0 + n0) / a.W;
      const int wcol = (px0 + n0) % a.W + l31;
      const float* zr = a.zg + ((size_t)b * a.P + prow) * a.K2in * C * 2;
      for (int s = 0; s < a.K2in; ++s) {
        const float bf = a.tinv[(2 * s + half) * a.W + wcol];
#pragma unroll
        for (int m = 0; m < MT; ++m) {
          const float af = zr[(s * C + m * 32 + l31) * 2 + half];
          acc[m] = mfma32(af, bf, acc[m]);
        }
      }
    }
    // ---- activation derivative; us becomes a_l = act(u_l) in place --------
    // (each (i, px) element of us is touched only by the lane that owns it)
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int i = m * 32 + acc_row32(r, half);
        float v = acc[m][r];
        if (a.act_in) {
          const float uu = us[i * PITCH + n0 + l31];
          v *= gelu_grad_f(uu);
          us[i * PITCH + n0 + l31] = gelu_f(uu);
        }
        acc[m][r] = v;
        if (a.gout) a.gout[((size_t)b * C + i) * a.PW + px0 + n0 + l31] = v;
      }
    __syncthreads();

    // ---- dW[o][i] += sum_px g[o][px] a[i][px] ------------------------------
    // k index of MFMA #t in group q: lane-half h <-> pixel 8q + 4h + t (one b128 per 4 MFMAs)
#pragma unroll
    for (int t = 0; t < TPW; ++t) {
      const int job = wave + t * NW;
      const int tl = job % TILES, kp = (job / TILES) % KSPLIT;
      if (TPW > 1 && job >= TILES) break;
      const int mt = tl / MT, nt = tl % MT;
      constexpr int PXK = NPX / KSPLIT;
      const float* ga = gs + (mt * 32 + l31) * PITCH + kp * PXK + 4 * half;
      const float* ab = us + (nt * 32 + l31) * PITCH + kp * PXK + 4 * half;
#pragma unroll 4
      for (int q = 0; q < PXK / 8; ++q) {
        const float4 av = ld4(ga + 8 * q);
        const float4 bv = ld4(ab + 8 * q);
        dwacc[t] = mfma32(av.x, bv.x, dwacc[t]);
        dwacc[t] = mfma32(av.y, bv.y, dwacc[t]);
        dwacc[t] = mfma32(av.z, bv.z, dwacc[t]);
        dwacc[t] = mfma32(av.w, bv.w, dwacc[t]);
      }
    }

    if (a.x1g || a.xin) {
      __syncthreads();  // dW GEMM done with gs
#pragma unroll
      for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r)
          gs[(m * 32 + acc_row32(r, half)) * PITCH + n0 + l31] = acc[m][r];
      __syncthreads();
      if (a.x1g) {
        const int R = NPX / a.W;
        const int njobs = (C / 16) * R * a.NJ;
        for (int job = wave; job < njobs; job += NW) {
          const int nt = job % (C / 16);
          const int rr = (job / (C / 16)) % R;
          const int jt = job / ((C / 16) * R);
          f32x4 d = {0.f, 0.f, 0.f, 0.f};
          const float* tf = a.tfwd + (size_t)(jt * 16 + l15) * a.W + quad;
          const float* xr = gs + (nt * 16 + l15) * PITCH + rr * a.W + quad;
          for (int s = 0; s < a.W / 4; ++s) d = mfma16(tf[4 * s], xr[4 * s], d);
          const int prow = px0 / a.W + rr;
          const int o = nt * 16 + l15;
#pragma unroll
          for (int pr = 0; pr < 2; ++pr) {
            const int k2 = jt * 8 + quad * 2 + pr;
            if (k2 < a.K2out)
              *reinterpret_cast<float2*>(a.x1g + ((((size_t)b * a.P + prow) * a.K2out + k2) * C + o) * 2) =
                  make_float2(d[2 * pr], d[2 * pr + 1]);
          }
        }
      }
      if (a.xin) {
        // dl[c][n] += sum_px gout[c][px] * xext[n][px],  xext = [x_in rows | ones | 0..]
#pragma unroll
        for (int j = 0; j < LJ; ++j) {
          const int mt = wave + j * NW;
          if (mt < C / 16) {
            const float* ar = gs + (mt * 16 + l15) * PITCH + quad;
            const float* br = xls + l15 * PITCH + quad;
            for (int s = 0; s < NPX / 4; ++s) {
              const float bf = (l15 < a.CL) ? br[4 * s] : (l15 == a.CL ? 1.0f : 0.0f);
              dl[j] = mfma16(ar[4 * s], bf, dl[j]);
            }
          }
        }
      }
    }
    __syncthreads();
  }

  // ---- write partial slabs ---------------------------------------------------
#pragma unroll
  for (int t = 0; t < TPW; ++t) {
    const int job = wave + t * NW;
    if (TPW > 1 && job >= TILES) break;
    const int tl = job % TILES, kp = (job / TILES) % KSPLIT;
    const int mt = tl / MT, nt = tl % MT;
    float* dst = a.dw_part + ((size_t)blockIdx.x * KSPLIT + kp) * C * C;
#pragma unroll
    for (int r = 0; r < 16; ++r)
      dst[(mt * 32 + acc_row32(r, half)) * C + nt * 32 + l31] = dwacc[t][r];
  }
#pragma unroll
  for (int i = 0; i < C / 8; ++i) {
    float v = dbacc[i];
    for (int off = (NPX / 4) / 2; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    const int c = (tid + i * NT) / (NPX / 4);
    if ((tid % (NPX / 4)) == 0) a.db_part[(size_t)blockIdx.x * C + c] = v;
  }
  if (a.xin) {
#pragma unroll
    for (int j = 0; j < LJ; ++j) {
      const int mt = wave + j * NW;
      if (mt < C / 16) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
          a.dwl_part[((size_t)blockIdx.x * C + mt * 16 + quad * 4 + r) * 16 + l15] = dl[j][r];
      }
    }
  }
}

// out[e] = sum_s part[s*n + e]   (deterministic slab reduction; optionally accumulates)
__global__ void k_reduce_slabs(const float* __restrict__ part, float* __restrict__ out, int nslab, int n,
                               int ld_out, int ncols, int ld_in) {
  // element e = (row, col) with col < ncols; input row stride ld_in, output row stride ld_out
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n) return;
  const int row = e / ncols, col = e % ncols;
  float s = 0.f;
  const size_t slab = (size_t)(n / ncols) * ld_in;
  for (int k = 0; k < nslab; ++k) s += part[k * slab + (size_t)row * ld_in + col];
  out[(size_t)row * ld_out + col] = s;
}
