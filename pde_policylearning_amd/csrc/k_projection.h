// Fused projection MLP  y = W2 . gelu(W1 . a + b1) + b2   (neuralop/models/tfno.py:23-38)
// forward and backward.  The HID-wide hidden tensor (1.07 GB at the benchmark
// shape in the reference) never leaves the CU: it lives in MFMA accumulators,
// 64 hidden rows at a time.  Backward recomputes it instead of storing it.
#pragma once
#include "fno_dev.h"

#ifndef FNO_OCC_PF
#define FNO_OCC_PF 2
#endif
#ifndef FNO_OCC_PB
#define FNO_OCC_PB 2   // measured: 2 (no spills, 1 workgroup/CU) beats 4 (spills) on MI355X
#endif

struct ProjFwdArgs {
  const float* x;    // (B, C, PW) pre-activation u_L
  const float* w1;   // (HID, C) row-major
  const float* b1;   // (HID)
  const float* w2;   // (CO, HID)
  const float* b2;   // (CO)
  float* y;          // (B, CO, PW)
  int PW, CO, act_in, tiles_per_plane, ntiles;
  const float* xmax; // k_proj_fwd_h2: device scalar, a bound of |x| (published by the kernel that stored x)
  int share32 = 0;   // k_proj_fwd_w: 32nds of a CU's columns for the workgroup dispatched first (0 = even; pair_share, fno_dev.h)
};

constexpr int PROJ_MAXCO = 4;   // largest supported projection output width

// wave (hm, nt): hidden rows [64*ch + 32*hm, +32) of every chunk ch, pixels [32*nt, +32).
// W1 stays in LDS for the lifetime of the (persistent) workgroup, rows padded to C+1 floats so
// that both the row-per-lane fragment reads here and the channel-per-lane reads of the backward
// kernel are bank-conflict-free; the next tile's activations are prefetched into registers.
template <int C, int HID, int NPX, int NCO>
__global__ void __launch_bounds__(NPX * 4, FNO_OCC_PF) k_proj_fwd(ProjFwdArgs a) {
  constexpr int NTN = NPX / 32;
  constexpr int NW = 2 * NTN;
  constexpr int NT = NW * 64;
  constexpr int KS = C / 2;
  constexpr int PITCH = NPX + 4;
  constexpr int WP = C + 1;
  constexpr int NCH = HID / 64;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* xs = smem;                      // C x PITCH
  float* b1s = xs + C * PITCH;           // HID
  float* w2s = b1s + HID;                // NCO x HID
  float* ysh = w2s + NCO * HID;          // NCO x NPX: partial sums of the hm = 1 waves
  float* w1s = ysh + NCO * NPX;          // HID x WP
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, half = lane >> 5;
  const int hm = wave / NTN, nt = wave % NTN;
  const int n0 = nt * 32;

  for (int i = tid; i < HID; i += NT) b1s[i] = a.b1[i];
  for (int i = tid; i < NCO * HID; i += NT) w2s[i] = (i < a.CO * HID) ? a.w2[i] : 0.f;
  for (int i = tid; i < HID * C; i += NT) w1s[(i / C) * WP + i % C] = a.w1[i];

  TilePrefetch<NPX, NT, C, C> pfx;
  if ((int)blockIdx.x < a.ntiles)
    pfx.issue(a.x + (size_t)(blockIdx.x / a.tiles_per_plane) * C * a.PW + (blockIdx.x % a.tiles_per_plane) * NPX, a.PW, tid);

  for (int tile = blockIdx.x; tile < a.ntiles; tile += gridDim.x) {
    const int b = tile / a.tiles_per_plane;
    const int px0 = (tile % a.tiles_per_plane) * NPX;
    pfx.commit(xs, a.act_in != 0, tid);
    __syncthreads();
    {
      const int nt2 = tile + gridDim.x;
      if (nt2 < a.ntiles) {
        int t_ = tid;
        asm volatile("" : "+v"(t_));      // (no hoisted per-lane 64-bit prefetch addresses: k_pw_fwd_x3)
        pfx.issue(a.x + (size_t)(nt2 / a.tiles_per_plane) * C * a.PW + (nt2 % a.tiles_per_plane) * NPX, a.PW, t_);
      }
    }
    float ysum[NCO];
#pragma unroll
    for (int co = 0; co < NCO; ++co) ysum[co] = 0.f;
#pragma unroll 1
    for (int ch = 0; ch < NCH; ++ch) {
      f32x16 acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.f;
      const float* wp = w1s + (ch * 64 + hm * 32 + l31) * WP + half;
      const float* xp = xs + half * PITCH + n0 + l31;
#pragma unroll
      for (int s = 0; s < KS; ++s) acc = mfma32(wp[2 * s], xp[2 * s * PITCH], acc);
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
  const float* b1;    // (HID)
  const float* w2;    // (CO, HID)
  float* gout;        // (B, C, PW): dL/du_L
  float* x1g;         // (B, P, K2out, C, 2) or null
  const float* tfwd;  // (16*NJ, W)
  const unsigned short* wa1;   // split-precision A1 fragments (k_pack_w1_x3), or null
  const unsigned short* wa3;   // split-precision A3 fragments
  float* dw1_part;    // (gridDim, HID, C)
  float* db1_part;    // (gridDim * NPX/32, HID)
  float* dw2_part;    // (gridDim * NPX/32, CO, HID)
  int PW, W, P, K2out, NJ, CO, act_in, tiles_per_plane, ntiles;
  const float* amax;  // k_proj_bwd_t<.., 2>: {-, max |dy|, max |W1|, max |w2|} (device scalars)
  const float* xmax;  // ... and the bound of |x| the forward pass published
  float* gmax_out;    // k_proj_bwd_t: max |gout| is published here (bound for the next kernel's fp16 operand scale)
  int rev = 0;        // k_proj_bwd_t: walk the tiles from the last to the first (u_L's most recently read part - what the
                      // Infinity Cache still holds of it behind the forward pass - is read first)
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

// wave (hm, nt) as in k_proj_fwd.  Per 64-row hidden chunk (ONE barrier per chunk, the dP1
// chunk is double-buffered in LDS so wave groups run up to a chunk apart):
//   A1  recompute P1 (MFMA 32x32x2)
//   E   gl = gelu(P1), dP1 = gelu'(P1) * (W2^T dy); dW2 / db1 contributions are reduced over
//       the 32 pixels of the wave with DPP adds (no LDS round trip); dP1 -> LDS
//   A3  dx += W1^T dP1, dP1 fed to the MFMA straight from the accumulator registers (its row
//       index is the k index of this product)
//   --- barrier ---
//   B   dW1[chunk] += dP1 . a^T  (MFMA, K = pixels) by the wave group that owns this chunk,
//       while the other group already recomputes the next chunk
template <int C, int HID, int NPX, int NCO>
__global__ void __launch_bounds__(NPX * 4, FNO_OCC_PB) k_proj_bwd(ProjBwdArgs a) {
  // W1 is resident in LDS (rows padded to C+1 floats: conflict-free both for the row-per-lane
  // fragments of the recompute and the channel-per-lane fragments of the dx product); the dP1
  // chunk is double-buffered when two 64 x PITCH buffers still fit beside it
  constexpr int WP = C + 1;
  constexpr int kSmall = NCO * NPX + HID + NCO * HID;
  constexpr bool W1LDS = ((C + 64) * (NPX + 4) + kSmall + HID * WP) * 4 <= 160 * 1024;   // else: W1 from L2
  constexpr bool DBUF = ((C + 128) * (NPX + 4) + kSmall + (W1LDS ? HID * WP : 0)) * 4 <= 160 * 1024;
  using Cfg = ProjBwdCfg<C, HID, NPX>;
  constexpr int NTN = Cfg::NTN, NW = Cfg::NW, MT = Cfg::MT, NCH = Cfg::NCH, TILES = Cfg::TILES, G = Cfg::G,
                CPW = Cfg::CPW;
  constexpr int NT = NW * 64;
  constexpr int KS = C / 2;
  constexpr int PITCH = NPX + 4;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* xs = smem;                        // C x PITCH        : a = act(u_L); later the gout tile
  float* dps = xs + C * PITCH;             // 2 x 64 x PITCH   : dP1 chunks (double-buffered);
                                           //                    after the chunk loop: dx partials of hm = 1
  float* douts = dps + (DBUF ? 2 : 1) * 64 * PITCH;     // NCO x NPX
  float* b1s = douts + NCO * NPX;          // HID
  float* w2s = b1s + HID;                  // NCO x HID
  float* w1s = w2s + NCO * HID;            // HID x WP
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, half = lane >> 5;
  const int l15 = lane & 15;
  const int hm = wave / NTN, nt = wave % NTN;
  const int n0 = nt * 32;
  const int dgrp = wave / TILES, dtl = wave % TILES;
  const int dmt = dtl / MT, dnt = dtl % MT;  // dW1 tile: hidden 32-block, channel 32-block

  for (int i = tid; i < HID; i += NT) b1s[i] = a.b1[i];
  for (int i = tid; i < NCO * HID; i += NT) w2s[i] = (i < a.CO * HID) ? a.w2[i] : 0.f;
  if (W1LDS)
    for (int i = tid; i < HID * C; i += NT) w1s[(i / C) * WP + i % C] = a.w1[i];

  f32x16 dw1acc[CPW];
#pragma unroll
  for (int k = 0; k < CPW; ++k)
#pragma unroll
    for (int r = 0; r < 16; ++r) dw1acc[k][r] = 0.f;
  // lane (l15, half) accumulates hidden row  ch*64 + hm*32 + acc_row32(l15, half)  of every chunk
  float sdb1[NCH], sdw2[NCH][NCO];
#pragma unroll
  for (int ch = 0; ch < NCH; ++ch) {
    sdb1[ch] = 0.f;
#pragma unroll
    for (int co = 0; co < NCO; ++co) sdw2[ch][co] = 0.f;
  }

  TilePrefetch<NPX, NT, C, C> pfx;      // next tile's u_L rows, in flight during this tile
  if ((int)blockIdx.x < a.ntiles)
    pfx.issue(a.x + (size_t)(blockIdx.x / a.tiles_per_plane) * C * a.PW + (blockIdx.x % a.tiles_per_plane) * NPX, a.PW, tid);

  for (int tile = blockIdx.x; tile < a.ntiles; tile += gridDim.x) {
    const int b = tile / a.tiles_per_plane;
    const int px0 = (tile % a.tiles_per_plane) * NPX;
    pfx.commit(xs, a.act_in != 0, tid);
    for (int idx = tid; idx < NCO * NPX; idx += NT) {
      const int co = idx / NPX, p = idx % NPX;
      douts[idx] = (co < a.CO) ? a.dy[((size_t)b * a.CO + co) * a.PW + px0 + p] : 0.f;
    }
    __syncthreads();
    {
      const int nt2 = tile + gridDim.x;
      if (nt2 < a.ntiles) {
        int t_ = tid;
        asm volatile("" : "+v"(t_));      // (no hoisted per-lane 64-bit prefetch addresses: k_pw_fwd_x3)
        pfx.issue(a.x + (size_t)(nt2 / a.tiles_per_plane) * C * a.PW + (nt2 % a.tiles_per_plane) * NPX, a.PW, t_);
      }
    }
    float dyl[NCO];
#pragma unroll
    for (int co = 0; co < NCO; ++co) dyl[co] = douts[co * NPX + n0 + l31];

    f32x16 acc2[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc2[m][r] = 0.f;

#pragma unroll 1
    for (int ch = 0; ch < NCH; ++ch) {
      float* dpb = dps + (DBUF ? (ch & 1) : 0) * 64 * PITCH;
      // ---- A1 ------------------------------------------------------------
      f32x16 acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.f;
      if constexpr (W1LDS) {
        const float* wp = w1s + (ch * 64 + hm * 32 + l31) * WP + half;
        const float* xp = xs + half * PITCH + n0 + l31;
#pragma unroll
        for (int s = 0; s < KS; ++s) acc = mfma32(wp[2 * s], xp[2 * s * PITCH], acc);
      } else {
        const float* wp = a.w1 + (size_t)(ch * 64 + hm * 32 + l31) * C + half;
        const float* xp = xs + half * PITCH + n0 + l31;
#pragma unroll 8
        for (int s = 0; s < KS; ++s) acc = mfma32(wp[2 * s], xp[2 * s * PITCH], acc);
      }
      // ---- E ---------------------------------------------------------------
      {
        float* dpp = dpb + (hm * 32 + 4 * half) * PITCH + n0 + l31;
        const float* b1p = b1s + ch * 64 + hm * 32 + 4 * half;
        const float* w2p = w2s + ch * 64 + hm * 32 + 4 * half;
        float rdb = 0.f, rdw[NCO];
#pragma unroll
        for (int co = 0; co < NCO; ++co) rdw[co] = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int ro = (r & 3) + 8 * (r >> 2);
          float t = 0.f;
#pragma unroll
          for (int co = 0; co < NCO; ++co) t = fmaf(w2p[co * HID + ro], dyl[co], t);
          float gl, dg;
          gelu_both(acc[r] + b1p[ro], gl, dg);
          const float dp = dg * t;
          acc[r] = dp;
          dpp[ro * PITCH] = dp;
          const float sdp = half_reduce_sum(dp);
          rdb = (l15 == r) ? sdp : rdb;
#pragma unroll
          for (int co = 0; co < NCO; ++co) {
            const float sg = half_reduce_sum(gl * dyl[co]);
            rdw[co] = (l15 == r) ? sg : rdw[co];
          }
        }
#pragma unroll
        for (int k = 0; k < NCH; ++k)
          if (k == ch) {
            sdb1[k] += rdb;
#pragma unroll
            for (int co = 0; co < NCO; ++co) sdw2[k][co] += rdw[co];
          }
      }
      // ---- A3 --------------------------------------------------------------
      {
        constexpr int WS = W1LDS ? WP : C;
        const float* wr;
        if constexpr (W1LDS) wr = w1s + (ch * 64 + hm * 32 + 4 * half) * WP + l31;
        else wr = a.w1 + (size_t)(ch * 64 + hm * 32 + 4 * half) * C + l31;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int ro = (r & 3) + 8 * (r >> 2);
#pragma unroll
          for (int mc = 0; mc < MT; ++mc) acc2[mc] = mfma32(wr[ro * WS + mc * 32], acc[r], acc2[mc]);
        }
      }
      __syncthreads();
      // ---- B -----------------------------------------------------------------
      if (dgrp == ch % G) {
        const float* ga = dpb + (dmt * 32 + l31) * PITCH + 4 * half;
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
      if (!DBUF) __syncthreads();   // single buffer: the next chunk overwrites it
    }
    __syncthreads();   // all dW1 GEMMs done with xs / dps

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
      row_dft_epilogue<C, NPX, NW>(xs, a.tfwd, a.W, a.x1g, b, px0, a.P, a.W, a.K2out, a.NJ, wave, lane);
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
  if ((lane & 16) == 0) {      // lanes 16-31 / 48-63 hold duplicates
    const size_t slab = (size_t)blockIdx.x * NTN + nt;
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch) {
      const int hid = ch * 64 + hm * 32 + acc_row32(l15, half);
      a.db1_part[slab * HID + hid] = sdb1[ch];
#pragma unroll
      for (int co = 0; co < NCO; ++co)
        if (co < a.CO) a.dw2_part[(slab * a.CO + co) * HID + hid] = sdw2[ch][co];
    }
  }
}

// ---------------------------------------------------------------------------
// Split-precision (bf16x3 on the matrix cores, see fno_dev.h) variant of k_proj_fwd.
// W1 is split once per workgroup into MFMA A-fragment order and stays in LDS:
//   w1b[((mt*KB + kb)*3 + t)*64 + lane][8] = term t of W1[mt*32 + (lane&31)][kb*16 + 8*(lane>>5) + j]
// The activation tile is pixel-major bf16x3 (SplitTilePrefetch); each wave keeps its 32 pixels'
// B fragments (KB x 3 x 4 VGPRs) in registers for all hidden chunks.
template <int C, int HID, int NPX, int NCO, bool RELU = false>
__global__ void __launch_bounds__(NPX * 4, 2) k_proj_fwd_x3(ProjFwdArgs a) {
  constexpr int NTN = NPX / 32;
  constexpr int NW = 2 * NTN;
  constexpr int NT = NW * 64;
  constexpr int KB = C / 16;
  constexpr int NCH = HID / 64;
  using PF = SplitTilePrefetch<NPX, NT, C>;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  unsigned short* xb = reinterpret_cast<unsigned short*>(smem);          // 3 x NPX x (C+8) halfs
  unsigned short* w1b = xb + 3 * PF::TERM;                                // (HID/32) x KB x 3 x 64 x 8 halfs
  float* b1s = reinterpret_cast<float*>(w1b + (HID / 32) * KB * 3 * 64 * 8);   // HID
  float* w2s = b1s + HID;                                                 // NCO x HID
  float* ysh = w2s + NCO * HID;                                           // NCO x NPX
  float gk_six, gk_inf;                    // clamp constants of the packed GELU (fno_dev.h), in SGPRs
  gelu_consts(gk_six, gk_inf);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, half = lane >> 5;
  const int hm = wave / NTN, nt = wave % NTN;
  const int n0 = nt * 32;

  for (int i = tid; i < HID; i += NT) b1s[i] = a.b1[i];
  for (int i = tid; i < NCO * HID; i += NT) w2s[i] = (i < a.CO * HID) ? a.w2[i] : 0.f;
  for (int it = tid; it < (HID / 32) * KB * 64; it += NT) {      // item = (mt, kb, lane)
    const int ln = it & 63, kb = (it >> 6) % KB, mt = (it >> 6) / KB;
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = a.w1[(size_t)(mt * 32 + (ln & 31)) * C + kb * 16 + 8 * (ln >> 5) + j];
    bf16x8 h, m, l;
    split3x8(v, h, m, l);
    unsigned short* dst = w1b + ((size_t)((mt * KB + kb) * 3) * 64 + ln) * 8;
    st8h(dst, h);
    st8h(dst + 64 * 8, m);
    st8h(dst + 2 * 64 * 8, l);
  }

  PF pfx;
  if ((int)blockIdx.x < a.ntiles)
    pfx.issue(a.x + (size_t)(blockIdx.x / a.tiles_per_plane) * C * a.PW + (blockIdx.x % a.tiles_per_plane) * NPX, a.PW, tid);

  for (int tile = blockIdx.x; tile < a.ntiles; tile += gridDim.x) {
    const int b = tile / a.tiles_per_plane;
    const int px0 = (tile % a.tiles_per_plane) * NPX;
    pfx.commit(xb, a.act_in != 0, tid);
    __syncthreads();
    {
      const int nt2 = tile + gridDim.x;
      if (nt2 < a.ntiles) {
        int t_ = tid;
        asm volatile("" : "+v"(t_));      // (no hoisted per-lane 64-bit prefetch addresses: k_pw_fwd_x3)
        pfx.issue(a.x + (size_t)(nt2 / a.tiles_per_plane) * C * a.PW + (nt2 % a.tiles_per_plane) * NPX, a.PW, t_);
      }
    }
#ifndef FNO_PFWD_STAGGER
#define FNO_PFWD_STAGGER 0
#endif
    // the chunk loop below has no barrier: both waves of a SIMD would run their matrix phases and their GELU phases together;
    // holding one partner back by about half a chunk puts one's GELU beside the other's MFMAs
    if (FNO_PFWD_STAGGER > 0 && hm == 1) __builtin_amdgcn_s_sleep(FNO_PFWD_STAGGER);
    // this wave's activation fragments: B[k = c][n = px], 8 consecutive channels per lane
    bf16x8 bfrag[KB][3];
#pragma unroll
    for (int kb = 0; kb < KB; ++kb)
#pragma unroll
      for (int t = 0; t < 3; ++t)
        bfrag[kb][t] = ld8h(xb + t * PF::TERM + (n0 + l31) * PF::PBH + kb * 16 + 8 * half);

    float ysum[NCO];
#pragma unroll
    for (int co = 0; co < NCO; ++co) ysum[co] = 0.f;
#pragma unroll 1
    for (int ch = 0; ch < NCH; ++ch) {
      f32x16 acc, lo;      // hh products / cross terms of the split (fno_dev.h: mfma_x3s)
#pragma unroll
      for (int r = 0; r < 16; ++r) { acc[r] = 0.f; lo[r] = 0.f; }
      const unsigned short* wa = w1b + ((size_t)((ch * 2 + hm) * KB * 3) * 64 + lane) * 8;
#pragma unroll
      for (int kb = 0; kb < KB; ++kb) {
        bf16x8 af[3];
#pragma unroll
        for (int t = 0; t < 3; ++t) af[t] = ld8h(wa + (size_t)(kb * 3 + t) * 64 * 8);
        mfma_x3s(af, bfrag[kb], acc, lo);
      }
      const float* b1p = b1s + ch * 64 + hm * 32 + 4 * half;
      const float* w2p = w2s + ch * 64 + hm * 32 + 4 * half;
      // the hidden activation on pairs (fno_dev.h: gelu_pairs): 16 hidden rows of this lane's pixel
      f32x2 hp[8];
#pragma unroll
      for (int r = 0; r < 16; r += 2) {
        hp[r >> 1][0] = acc[r] + lo[r] + b1p[(r & 3) + 8 * (r >> 2)];
        hp[r >> 1][1] = acc[r + 1] + lo[r + 1] + b1p[((r + 1) & 3) + 8 * ((r + 1) >> 2)];
      }
      if constexpr (RELU) {
#pragma unroll
        for (int k = 0; k < 8; ++k) { hp[k][0] = fmaxf(hp[k][0], 0.f); hp[k][1] = fmaxf(hp[k][1], 0.f); }
      } else gelu_pairs<8>(hp, gk_six, gk_inf);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int ro = (r & 3) + 8 * (r >> 2);
        const float gl = hp[r >> 1][r & 1];
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
