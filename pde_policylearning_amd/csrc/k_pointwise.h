// Fused pointwise (1x1 conv) + truncated-spectrum row passes, forward direction.
//
// One workgroup owns a tile of NPX consecutive pixels of one sample's (P x W)
// plane, for ALL channels; wave (mt, nt) owns the 32x32 output sub-tile
// (channels 32*mt.., pixels 32*nt..).  Per tile (persistent grid-stride loop):
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

#ifndef FNO_OCC_PW
#define FNO_OCC_PW 4
#endif

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
__global__ void __launch_bounds__((COUT / 32) * (NPX / 32) * 64, FNO_OCC_PW) k_pw_fwd(PwFwdArgs a) {
  constexpr int NTN = NPX / 32;          // pixel sub-tiles
  constexpr int MT = COUT / 32;          // channel sub-tiles
  constexpr int NW = MT * NTN;
  constexpr int NT = NW * 64;
  constexpr int CINP = (CIN + 1) & ~1;
  constexpr int KS = CINP / 2;
  constexpr int PITCH = NPX + 4;
  static_assert(COUT % 32 == 0 && NPX % 32 == 0, "tile shape");
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* xs = smem;  // max(CINP, COUT) rows of PITCH floats

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, half = lane >> 5;
  const int mt = wave / NTN, nt = wave % NTN;
  const int n0 = nt * 32;
  const bool has_conv = a.x != nullptr;

  // weight fragments: A[i = o][k = c], constant over all tiles of this workgroup
  float afrag[KS];
#pragma unroll
  for (int s = 0; s < KS; ++s) {
    const int c = 2 * s + half;
    afrag[s] = (has_conv && c < CIN) ? a.w[(mt * 32 + l31) * CIN + c] : 0.0f;
  }

  for (int tile = blockIdx.x; tile < a.ntiles; tile += gridDim.x) {
    const int b = tile / a.tiles_per_plane;
    const int px0 = (tile % a.tiles_per_plane) * NPX;

    if (has_conv) {
      if (CIN == CINP)
        stage_rows_t<NPX, NT, CINP>(xs, a.x + (size_t)b * CIN * a.PW + px0, a.PW, a.act_in != 0, tid);
      else
        stage_rows<NPX, NT>(xs, a.x + (size_t)b * CIN * a.PW + px0, a.PW, CIN, CINP, a.act_in != 0, tid);
    }
    __syncthreads();

    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
    if (has_conv) {
#pragma unroll
      for (int s = 0; s < KS; ++s) acc = mfma32(afrag[s], xs[(2 * s + half) * PITCH + n0 + l31], acc);
    }
    if (a.z) {
      const int prow = (px0 + n0) / a.W;
      const int wcol = (px0 + n0) % a.W + l31;
      const float* zr = a.z + (((size_t)b * a.P + prow) * a.K2in * COUT + mt * 32 + l31) * 2 + half;
      const float* tv = a.tinv + (size_t)half * a.W + wcol;
      for (int s = 0; s < a.K2in; ++s) acc = mfma32(zr[(size_t)s * COUT * 2], tv[(size_t)2 * s * a.W], acc);
    }
    __syncthreads();  // all waves are done reading the staged input

    {
      // one base pointer + small row offsets (keeps 16 x 64-bit addresses out of registers)
      float* up = a.u ? a.u + ((size_t)b * COUT + mt * 32 + 4 * half) * a.PW + px0 + n0 + l31 : nullptr;
      float* xp = xs + (mt * 32 + 4 * half) * PITCH + n0 + l31;
      const float* bp = a.bias ? a.bias + mt * 32 + 4 * half : nullptr;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int ro = (r & 3) + 8 * (r >> 2);          // acc_row32(r, half) - 4 * half
        const float v = acc[r] + (bp ? bp[ro] : 0.0f);
        if (up) up[(size_t)ro * a.PW] = v;
        if (a.x1) xp[ro * PITCH] = a.act_out ? gelu_f(v) : v;
      }
    }
    if (a.x1) {
      __syncthreads();
      row_dft_epilogue<COUT, NPX, NW>(xs, a.tfwd, a.x1, b, px0, a.P, a.W, a.K2out, a.NJ, wave, lane);
    }
    __syncthreads();  // xs is restaged by the next tile
  }
}
