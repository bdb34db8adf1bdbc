// Fused projection MLP  y = W2 . gelu(W1 . a + b1) + b2   (neuralop/models/tfno.py:23-38)
// forward and backward.  The HID-wide hidden tensor (1.07 GB at the benchmark
// shape in the reference) never leaves the CU: it lives in MFMA accumulators,
// 64 hidden rows at a time.  Backward recomputes it instead of storing it.
#pragma once
#include "fno_dev.h"

#ifndef FNO_OCC_PF
#define FNO_OCC_PF 4
#endif
#ifndef FNO_OCC_PB
#define FNO_OCC_PB 2   // measured: 2 (no spills, 1 workgroup/CU) beats 4 (spills) on MI355X
#endif

// W1 (HID, C) row-major -> MFMA A-fragment order so that a wave reads its
// fragments as one fully coalesced 256-B load per k-step:
//   w1p[((ch*2 + m)*(C/2) + s)*64 + lane] = W1[ch*64 + m*32 + (lane&31)][2*s + (lane>>5)]
__global__ void k_pack_w1(const float* __restrict__ w1, float* __restrict__ w1p, int HID, int C) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= HID * C) return;
  const int lane = e & 63;
  const int s = (e >> 6) % (C / 2);
  const int m = ((e >> 6) / (C / 2)) & 1;
  const int ch = (e >> 6) / (C / 2) / 2;
  w1p[e] = w1[(ch * 64 + m * 32 + (lane & 31)) * C + 2 * s + (lane >> 5)];
}

struct ProjFwdArgs {
  const float* x;    // (B, C, PW) pre-activation u_L
  const float* w1p;  // packed W1
  const float* b1;   // (HID)
  const float* w2;   // (CO, HID)
  const float* b2;   // (CO)
  float* y;          // (B, CO, PW)
  int PW, CO, act_in, tiles_per_plane, ntiles;
};

constexpr int PROJ_MAXCO = 4;   // largest supported projection output width

// wave (hm, nt): hidden rows [64*ch + 32*hm, +32) of every chunk ch, pixels [32*nt, +32)
template <int C, int HID, int NPX, int NCO>
__global__ void __launch_bounds__(NPX * 4, FNO_OCC_PF) k_proj_fwd(ProjFwdArgs a) {
  constexpr int NTN = NPX / 32;
  constexpr int NW = 2 * NTN;
  constexpr int NT = NW * 64;
  constexpr int KS = C / 2;
  constexpr int PITCH = NPX + 4;
  constexpr int NCH = HID / 64;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* xs = smem;                      // C x PITCH
  float* b1s = xs + C * PITCH;           // HID
  float* w2s = b1s + HID;                // MAXCO x HID
  float* ysh = w2s + NCO * HID;   // MAXCO x NPX: partial sums of the hm = 1 waves
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, half = lane >> 5;
  const int hm = wave / NTN, nt = wave % NTN;
  const int n0 = nt * 32;

  for (int i = tid; i < HID; i += NT) b1s[i] = a.b1[i];
  for (int i = tid; i < NCO * HID; i += NT) w2s[i] = (i < a.CO * HID) ? a.w2[i] : 0.f;

  for (int tile = blockIdx.x; tile < a.ntiles; tile += gridDim.x) {
    const int b = tile / a.tiles_per_plane;
    const int px0 = (tile % a.tiles_per_plane) * NPX;
    stage_rows_t<NPX, NT, C>(xs, a.x + (size_t)b * C * a.PW + px0, a.PW, a.act_in != 0, tid);
    __syncthreads();
    const float* w1p_t = a.w1p;
    asm volatile("" : "+s"(w1p_t));   // keep the fragment loads inside the tile loop (L2-resident)
    float ysum[NCO];
#pragma unroll
    for (int co = 0; co < NCO; ++co) ysum[co] = 0.f;
#pragma unroll 1
    for (int ch = 0; ch < NCH; ++ch) {
      f32x16 acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.f;
      const float* wp = w1p_t + ((size_t)(ch * 2 + hm) * KS) * 64 + lane;
#pragma unroll 8
      for (int s = 0; s < KS; ++s) acc = mfma32(wp[s * 64], xs[(2 * s + half) * PITCH + n0 + l31], acc);
      const float* b1p = b1s + ch * 64 + hm * 32 + 4 * half;
      const float* w2p = w2s + ch * 64 + hm * 32 + 4 * half;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int ro = (r & 3) + 8 * (r >> 2);
        const float gl = gelu_f(acc[r] + b1p[ro]);
#pragma unroll
        for (int co = 0; co < NCO; ++co) ysum[co] = fmaf(w2p[co * HID + ro], gl, ysum[co]);
      }
    }
#pragma unroll
    for (int co = 0; co < NCO; ++co) {
      ysum[co] += __shfl_xor(ysum[co], 32, 64);
      if (hm == 1 && half == 0) ysh[co * NPX + n0 + l31] = ysum[co];
    }
    __syncthreads();
    if (hm == 0 && half == 0) {
#pragma unroll
      for (int co = 0; co < NCO; ++co)
        if (co < a.CO)
          a.y[((size_t)b * a.CO + co) * a.PW + px0 + n0 + l31] = ysum[co] + ysh[co * NPX + n0 + l31] + a.b2[co];
    }
    __syncthreads();
  }
}

struct ProjBwdArgs {
  const float* x;     // (B, C, PW) u_L
  const float* dy;    // (B, CO, PW)
  const float* w1;    // (HID, C) row-major
  const float* w1p;   // packed W1
  const float* b1;    // (HID)
  const float* w2;    // (CO, HID)
  float* gout;        // (B, C, PW): dL/du_L
  float* x1g;         // (B, P, K2out, C, 2) or null
  const float* tfwd;  // (16*NJ, W)
  float* dw1_part;    // (gridDim, HID, C)
  float* db1_part;    // (gridDim, HID)
  float* dw2_part;    // (gridDim, CO, HID)
  int PW, W, P, K2out, NJ, CO, act_in, tiles_per_plane, ntiles;
};

template <int C, int HID, int NPX>
struct ProjBwdCfg {
  static constexpr int NTN = NPX / 32;
  static constexpr int NW = 2 * NTN;
  static constexpr int MT = C / 32;
  static constexpr int NCH = HID / 64;
  static constexpr int TILES = 2 * MT;          // 32x32 tiles of one chunk's dW1 (64 x C)
  static constexpr int GW = NW / TILES;         // wave groups available
  static constexpr int G = GW < NCH ? GW : NCH; // groups used; chunk ch is owned by group ch % G
  static constexpr int CPW = NCH / G;           // chunks per owning wave
  static_assert(NW % TILES == 0 && G >= 1 && NCH % G == 0, "projection backward tiling");
};

// wave (hm, nt) as in k_proj_fwd.  Per 64-row hidden chunk:
//   A1  recompute P1 (MFMA), gl = gelu(P1) -> LDS, dP1 = gelu'(P1) * (W2^T dy) in registers
//   A2  dW2 += sum_px gl * dy                      (thread sums over the LDS tile)
//   A3  dP1 -> LDS;  dx += W1^T dP1, dP1 fed to the MFMA straight from the accumulator
//       registers (its row index is the k index of this product)
//   B   db1 += sum_px dP1 (thread sums);  dW1[chunk] += dP1 . a^T  (MFMA, K = pixels) by the
//       wave group that owns this chunk
template <int C, int HID, int NPX, int NCO>
__global__ void __launch_bounds__(NPX * 4, FNO_OCC_PB) k_proj_bwd(ProjBwdArgs a) {
  using Cfg = ProjBwdCfg<C, HID, NPX>;
  constexpr int NTN = Cfg::NTN, NW = Cfg::NW, MT = Cfg::MT, NCH = Cfg::NCH, TILES = Cfg::TILES, G = Cfg::G,
                CPW = Cfg::CPW;
  constexpr int NT = NW * 64;
  constexpr int KS = C / 2;
  constexpr int PITCH = NPX + 4;
  constexpr int TPX = NPX / NW;             // pixels per thread-sum group
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* xs = smem;                        // C x PITCH   : a = act(u_L); later the gout tile
  float* dps = xs + C * PITCH;             // 64 x PITCH  : gelu(P1) then dP1 of the current chunk;
                                           //               after the chunk loop: dx partials of hm = 1
  float* douts = dps + 64 * PITCH;         // MAXCO x NPX
  float* b1s = douts + NCO * NPX;   // HID
  float* w2s = b1s + HID;                  // MAXCO x HID
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, half = lane >> 5;
  const int hm = wave / NTN, nt = wave % NTN;
  const int n0 = nt * 32;
  const int hl = tid & 63, qt = tid >> 6;   // thread-sum mapping: hidden row, pixel group
  const int dgrp = wave / TILES, dtl = wave % TILES;
  const int dmt = dtl / MT, dnt = dtl % MT;  // dW1 tile: hidden 32-block, channel 32-block

  for (int i = tid; i < HID; i += NT) b1s[i] = a.b1[i];
  for (int i = tid; i < NCO * HID; i += NT) w2s[i] = (i < a.CO * HID) ? a.w2[i] : 0.f;

  f32x16 dw1acc[CPW];
#pragma unroll
  for (int k = 0; k < CPW; ++k)
#pragma unroll
    for (int r = 0; r < 16; ++r) dw1acc[k][r] = 0.f;
  float sdb1[NCH], sdw2[NCH][NCO];
#pragma unroll
  for (int ch = 0; ch < NCH; ++ch) {
    sdb1[ch] = 0.f;
#pragma unroll
    for (int co = 0; co < NCO; ++co) sdw2[ch][co] = 0.f;
  }

  for (int tile = blockIdx.x; tile < a.ntiles; tile += gridDim.x) {
    const int b = tile / a.tiles_per_plane;
    const int px0 = (tile % a.tiles_per_plane) * NPX;
    stage_rows_t<NPX, NT, C>(xs, a.x + (size_t)b * C * a.PW + px0, a.PW, a.act_in != 0, tid);
    for (int idx = tid; idx < NCO * NPX; idx += NT) {
      const int co = idx / NPX, p = idx % NPX;
      douts[idx] = (co < a.CO) ? a.dy[((size_t)b * a.CO + co) * a.PW + px0 + p] : 0.f;
    }
    __syncthreads();
    float dyl[NCO];
#pragma unroll
    for (int co = 0; co < NCO; ++co) dyl[co] = douts[co * NPX + n0 + l31];
    const float* w1p_t = a.w1p;
    const float* w1_t = a.w1;
    asm volatile("" : "+s"(w1p_t), "+s"(w1_t));   // keep weight loads inside the tile loop

    f32x16 acc2[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc2[m][r] = 0.f;

#pragma unroll 1
    for (int ch = 0; ch < NCH; ++ch) {
      // ---- A1 ------------------------------------------------------------
      f32x16 acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.f;
      const float* wp = w1p_t + ((size_t)(ch * 2 + hm) * KS) * 64 + lane;
#pragma unroll 8
      for (int s = 0; s < KS; ++s) acc = mfma32(wp[s * 64], xs[(2 * s + half) * PITCH + n0 + l31], acc);
      {
        float* dpp = dps + (hm * 32 + 4 * half) * PITCH + n0 + l31;
        const float* b1p = b1s + ch * 64 + hm * 32 + 4 * half;
        const float* w2p = w2s + ch * 64 + hm * 32 + 4 * half;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int ro = (r & 3) + 8 * (r >> 2);
          float t = 0.f;
#pragma unroll
          for (int co = 0; co < NCO; ++co) t = fmaf(w2p[co * HID + ro], dyl[co], t);
          float gl, dg;
          gelu_both(acc[r] + b1p[ro], gl, dg);
          dpp[ro * PITCH] = gl;
          acc[r] = dg * t;
        }
      }
      __syncthreads();
      // ---- A2: dW2[co][hid] += sum_px gl[hid][px] * dy[co][px] -----------
      {
        const float* gr = dps + hl * PITCH + qt * TPX;
        float t2[NCO];
#pragma unroll
        for (int co = 0; co < NCO; ++co) t2[co] = 0.f;
#pragma unroll
        for (int j = 0; j < TPX / 4; ++j) {
          const float4 gv = ld4(gr + 4 * j);
#pragma unroll
          for (int co = 0; co < NCO; ++co) {
            const float4 dv = ld4(douts + co * NPX + qt * TPX + 4 * j);
            t2[co] += gv.x * dv.x + gv.y * dv.y + gv.z * dv.z + gv.w * dv.w;
          }
        }
#pragma unroll
        for (int k = 0; k < NCH; ++k)
          if (k == ch) {
#pragma unroll
            for (int co = 0; co < NCO; ++co) sdw2[k][co] += t2[co];
          }
      }
      __syncthreads();
      // ---- A3 --------------------------------------------------------------
      {
        float* dpp = dps + (hm * 32 + 4 * half) * PITCH + n0 + l31;
        const float* wr = w1_t + (size_t)(ch * 64 + hm * 32 + 4 * half) * C + l31;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int ro = (r & 3) + 8 * (r >> 2);
          dpp[ro * PITCH] = acc[r];
#pragma unroll
          for (int mc = 0; mc < MT; ++mc) acc2[mc] = mfma32(wr[ro * C + mc * 32], acc[r], acc2[mc]);
        }
      }
      __syncthreads();
      // ---- B -----------------------------------------------------------------
      {
        const float* gr = dps + hl * PITCH + qt * TPX;
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < TPX / 4; ++j) {
          const float4 gv = ld4(gr + 4 * j);
          s += (gv.x + gv.y) + (gv.z + gv.w);
        }
#pragma unroll
        for (int k = 0; k < NCH; ++k)
          if (k == ch) sdb1[k] += s;
        if (dgrp == ch % G) {
          const float* ga = dps + (dmt * 32 + l31) * PITCH + 4 * half;
          const float* ab = xs + (dnt * 32 + l31) * PITCH + 4 * half;
#pragma unroll
          for (int k = 0; k < CPW; ++k)
            if (k == ch / G) {
              f32x16 dacc = dw1acc[k];
#pragma unroll 2
              for (int q = 0; q < NPX / 8; ++q) {
                const float4 av = ld4(ga + 8 * q);
                const float4 bv = ld4(ab + 8 * q);
                dacc = mfma32(av.x, bv.x, dacc);
                dacc = mfma32(av.y, bv.y, dacc);
                dacc = mfma32(av.z, bv.z, dacc);
                dacc = mfma32(av.w, bv.w, dacc);
              }
              dw1acc[k] = dacc;
            }
        }
      }
      __syncthreads();
    }

    // ---- dx: add the two hidden halves, (x act'), store, row DFT -------------
    if (hm == 1) {
      float* dpp = dps + (4 * half) * PITCH + n0 + l31;
#pragma unroll
      for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) dpp[(m * 32 + (r & 3) + 8 * (r >> 2)) * PITCH] = acc2[m][r];
    }
    __syncthreads();
    if (hm == 0) {
      const float* dpp = dps + (4 * half) * PITCH + n0 + l31;
      float* xp = xs + (4 * half) * PITCH + n0 + l31;
      const size_t goff = ((size_t)b * C + 4 * half) * a.PW + px0 + n0 + l31;
#pragma unroll
      for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int ro = m * 32 + (r & 3) + 8 * (r >> 2);
          float v = acc2[m][r] + dpp[ro * PITCH];
          if (a.act_in) v *= gelu_grad_f(a.x[goff + (size_t)ro * a.PW]);
          a.gout[goff + (size_t)ro * a.PW] = v;
          if (a.x1g) xp[ro * PITCH] = v;
        }
    }
    if (a.x1g) {
      __syncthreads();
      row_dft_epilogue<C, NPX, NW>(xs, a.tfwd, a.x1g, b, px0, a.P, a.W, a.K2out, a.NJ, wave, lane);
    }
    __syncthreads();
  }

  // ---- partial slabs -------------------------------------------------------
#pragma unroll
  for (int k = 0; k < CPW; ++k) {
    const int ch = dgrp + k * G;
    if (dgrp >= G) break;
    float* dst = a.dw1_part + (size_t)blockIdx.x * HID * C;
#pragma unroll
    for (int r = 0; r < 16; ++r)
      dst[(size_t)(ch * 64 + dmt * 32 + acc_row32(r, half)) * C + dnt * 32 + l31] = dw1acc[k][r];
  }
  // thread sums: reduce the NW pixel-group threads of each hidden row through LDS
  __syncthreads();
  float* red = smem;  // NW x HID x (1 + MAXCO)
#pragma unroll
  for (int ch = 0; ch < NCH; ++ch) {
    float* rp = red + ((size_t)(qt * HID + ch * 64 + hl)) * (1 + NCO);
    rp[0] = sdb1[ch];
#pragma unroll
    for (int co = 0; co < NCO; ++co) rp[1 + co] = sdw2[ch][co];
  }
  __syncthreads();
  for (int h = tid; h < HID; h += NT) {
    float s1 = 0.f, s2[NCO];
#pragma unroll
    for (int co = 0; co < NCO; ++co) s2[co] = 0.f;
    for (int q = 0; q < NW; ++q) {
      const float* rp = red + ((size_t)(q * HID + h)) * (1 + NCO);
      s1 += rp[0];
#pragma unroll
      for (int co = 0; co < NCO; ++co) s2[co] += rp[1 + co];
    }
    a.db1_part[(size_t)blockIdx.x * HID + h] = s1;
#pragma unroll
    for (int co = 0; co < NCO; ++co)
      if (co < a.CO) a.dw2_part[((size_t)blockIdx.x * a.CO + co) * HID + h] = s2[co];
  }
}
