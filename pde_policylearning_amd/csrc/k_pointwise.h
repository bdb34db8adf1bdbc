// Fused pointwise (1x1 conv) + truncated-spectrum row passes, forward direction.
//
// One workgroup owns a tile of NPX consecutive pixels of one sample's (P x W)
// plane, for ALL channels.  Per tile (persistent grid-stride loop):
//   1. stage x[b, 0:CIN, tile] HBM -> LDS (16 B/lane coalesced), applying the
//      previous block's GELU on load when act_in (activations are stored
//      PRE-activation so that backward never needs a second copy);
//   2. D[o][px] = sum_c W[o][c] a[c][px]                      (fp32 MFMA 32x32x2)
//               + sum_j Z[b,row,j,o] * Tinv[j][w]             (row inverse DFT folded in
//                                                              as a K-extension of the GEMM)
//               + bias[o];
//   3. u = D is written to HBM; act_out(u) goes back to LDS;
//   4. row forward DFT of the tile, truncated to the kept bins, on fp32 MFMA
//      16x16x4:  X1[b,row,k2,o] = sum_w a[o][w] * Tfwd[j][w]   -> HBM (small).
// So a whole FNO block costs one read and one write of the activation.
//
// Replaces, for the reference's default FNO path: Lifting (tfno.py:19-20), the
// skip conv + residual add + GELU of FNOBlocks.forward (fno_block.py:131,147-150),
// the last-dim pass of irfftn + bias (spectral_convolution.py:342-345) and the
// last-dim pass of the NEXT block's rfftn (spectral_convolution.py:324).
#pragma once
#include "fno_dev.h"

struct PwFwdArgs {
  const float* x;     // (B, CIN, PW) pre-activation input, or null (no conv part)
  const float* w;     // (COUT, CIN) row-major
  const float* bias;  // (COUT) or null
  const float* z;     // (B, P, K2in, COUT, 2) column-inverse-transformed spectrum, or null
  const float* tinv;  // (2*K2in, W) row inverse table (scale and gamma folded in)
  float* u;           // (B, COUT, PW) output (pre-activation), or null
  float* x1;          // (B, P, K2out, COUT, 2) row spectra of act_out(u), or null
  const float* tfwd;  // (16*NJ, W) row forward table, zero rows past 2*K2out
  int PW, W, P, K2in, K2out, NJ;
  int act_in, act_out;
  int tiles_per_plane, ntiles;
};

template <int CIN, int COUT, int NPX>
__global__ void __launch_bounds__(NPX * 2) k_pw_fwd(PwFwdArgs a) {
  constexpr int NW = NPX / 32;
  constexpr int NT = NW * 64;
  constexpr int CINP = (CIN + 1) & ~1;
  constexpr int MT = COUT / 32;
  constexpr int KS = CINP / 2;
  constexpr int PITCH = NPX + 4;
  static_assert(COUT % 32 == 0 && NPX % 32 == 0, "tile shape");
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* xs = smem;  // max(CINP, COUT) rows of PITCH floats

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, half = lane >> 5;
  const int l15 = lane & 15, quad = lane >> 4;
  const bool has_conv = (CIN > 0) && (a.x != nullptr);

  // weight fragments: A[i = o][k = c], constant over all tiles of this workgroup
  float afrag[MT][KS > 0 ? KS : 1];
  if (has_conv) {
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int s = 0; s < KS; ++s) {
        const int c = 2 * s + half;
        afrag[m][s] = (c < CIN) ? a.w[(m * 32 + l31) * CIN + c] : 0.0f;
      }
  }
  float bias_r[MT][16];
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int r = 0; r < 16; ++r)
      bias_r[m][r] = a.bias ? a.bias[m * 32 + acc_row32(r, half)] : 0.0f;

  for (int tile = blockIdx.x; tile < a.ntiles; tile += gridDim.x) {
    const int b = tile / a.tiles_per_plane;
    const int px0 = (tile % a.tiles_per_plane) * NPX;

    if (has_conv) {
      const float* xb = a.x + (size_t)b * CIN * a.PW + px0;
      for (int idx = tid; idx < CINP * (NPX / 4); idx += NT) {
        const int c = idx / (NPX / 4), q = idx % (NPX / 4);
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (c < CIN) {
          v = ld4(xb + (size_t)c * a.PW + 4 * q);
          if (a.act_in) { v.x = gelu_f(v.x); v.y = gelu_f(v.y); v.z = gelu_f(v.z); v.w = gelu_f(v.w); }
        }
        st4(xs + c * PITCH + 4 * q, v);
      }
    }
    __syncthreads();

    f32x16 acc[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[m][r] = 0.0f;
    const int n0 = wave * 32;
    if (has_conv) {
#pragma unroll
      for (int s = 0; s < KS; ++s) {
        const float bf = xs[(2 * s + half) * PITCH + n0 + l31];
#pragma unroll
        for (int m = 0; m < MT; ++m) acc[m] = mfma32(afrag[m][s], bf, acc[m]);
      }
    }
    if (a.z) {
      const int prow = (px0 + n0) / a.W;
      const int wcol = (px0 + n0) % a.W + l31;
      const float* zr = a.z + ((size_t)b * a.P + prow) * a.K2in * COUT * 2;
      for (int s = 0; s < a.K2in; ++s) {
        const float bf = a.tinv[(2 * s + half) * a.W + wcol];
#pragma unroll
        for (int m = 0; m < MT; ++m) {
          const float af = zr[(s * COUT + m * 32 + l31) * 2 + half];
          acc[m] = mfma32(af, bf, acc[m]);
        }
      }
    }
    __syncthreads();  // all waves are done reading the staged input

#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int o = m * 32 + acc_row32(r, half);
        const float v = acc[m][r] + bias_r[m][r];
        if (a.u) a.u[((size_t)b * COUT + o) * a.PW + px0 + n0 + l31] = v;
        if (a.x1) xs[o * PITCH + n0 + l31] = a.act_out ? gelu_f(v) : v;
      }

    if (a.x1) {
      __syncthreads();
      const int R = NPX / a.W;
      const int njobs = (COUT / 16) * R * a.NJ;
      for (int job = wave; job < njobs; job += NW) {
        const int nt = job % (COUT / 16);
        const int rr = (job / (COUT / 16)) % R;
        const int jt = job / ((COUT / 16) * R);
        f32x4 d = {0.f, 0.f, 0.f, 0.f};
        const float* tf = a.tfwd + (size_t)(jt * 16 + l15) * a.W + quad;
        const float* xr = xs + (nt * 16 + l15) * PITCH + rr * a.W + quad;
        for (int s = 0; s < a.W / 4; ++s) d = mfma16(tf[4 * s], xr[4 * s], d);
        // D[row = j_local][col = o_local]: lane holds j = jt*16 + quad*4 + r, o = nt*16 + l15
        const int prow = px0 / a.W + rr;
        const int o = nt * 16 + l15;
#pragma unroll
        for (int pr = 0; pr < 2; ++pr) {
          const int k2 = jt * 8 + quad * 2 + pr;
          if (k2 < a.K2out) {
            float2 v2 = make_float2(d[2 * pr], d[2 * pr + 1]);
            *reinterpret_cast<float2*>(a.x1 + ((((size_t)b * a.P + prow) * a.K2out + k2) * COUT + o) * 2) = v2;
          }
        }
      }
    }
    __syncthreads();  // xs is restaged by the next tile
  }
}
