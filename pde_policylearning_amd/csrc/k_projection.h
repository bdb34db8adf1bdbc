// Fused projection MLP  y = W2 . gelu(W1 . a + b1) + b2   (neuralop/models/tfno.py:23-38)
// forward and backward.  The HID-wide hidden tensor (1.07 GB at the benchmark
// shape in the reference) never leaves the CU: it lives in MFMA accumulators,
// 64 hidden rows at a time.  Backward recomputes it instead of storing it.
#pragma once
#include "fno_dev.h"

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

constexpr int PROJ_MAXCO = 4;

template <int C, int HID, int NPX>
__global__ void __launch_bounds__(NPX * 2) k_proj_fwd(ProjFwdArgs a) {
  constexpr int NW = NPX / 32;
  constexpr int NT = NW * 64;
  constexpr int KS = C / 2;
  constexpr int PITCH = NPX + 4;
  constexpr int NCH = HID / 64;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* xs = smem;                    // C x PITCH
  float* b1s = xs + C * PITCH;         // HID
  float* w2s = b1s + HID;              // CO x HID
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, half = lane >> 5;

  for (int i = tid; i < HID; i += NT) b1s[i] = a.b1[i];
  for (int i = tid; i < a.CO * HID; i += NT) w2s[i] = a.w2[i];

  for (int tile = blockIdx.x; tile < a.ntiles; tile += gridDim.x) {
    const int b = tile / a.tiles_per_plane;
    const int px0 = (tile % a.tiles_per_plane) * NPX;
    const float* xb = a.x + (size_t)b * C * a.PW + px0;
    for (int idx = tid; idx < C * (NPX / 4); idx += NT) {
      const int c = idx / (NPX / 4), q = idx % (NPX / 4);
      float4 v = ld4(xb + (size_t)c * a.PW + 4 * q);
      if (a.act_in) { v.x = gelu_f(v.x); v.y = gelu_f(v.y); v.z = gelu_f(v.z); v.w = gelu_f(v.w); }
      st4(xs + c * PITCH + 4 * q, v);
    }
    __syncthreads();
    const int n0 = wave * 32;
    float ysum[PROJ_MAXCO] = {0.f, 0.f, 0.f, 0.f};
    for (int ch = 0; ch < NCH; ++ch) {
      f32x16 acc[2];
#pragma unroll
      for (int r = 0; r < 16; ++r) { acc[0][r] = 0.f; acc[1][r] = 0.f; }
      const float* wp = a.w1p + (size_t)ch * 2 * KS * 64 + lane;
#pragma unroll 8
      for (int s = 0; s < KS; ++s) {
        const float bf = xs[(2 * s + half) * PITCH + n0 + l31];
        acc[0] = mfma32(wp[s * 64], bf, acc[0]);
        acc[1] = mfma32(wp[(KS + s) * 64], bf, acc[1]);
      }
#pragma unroll
      for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int hid = ch * 64 + m * 32 + acc_row32(r, half);
          const float gl = gelu_f(acc[m][r] + b1s[hid]);
          for (int co = 0; co < a.CO; ++co) ysum[co] += w2s[co * HID + hid] * gl;
        }
    }
    for (int co = 0; co < a.CO; ++co) {
      const float v = ysum[co] + __shfl_xor(ysum[co], 32, 64);
      if (half == 0) a.y[((size_t)b * a.CO + co) * a.PW + px0 + n0 + l31] = v + a.b2[co];
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
  float* dw1_part;    // (gridDim*KSPLIT, HID, C)
  float* db1_part;    // (gridDim, HID)
  float* dw2_part;    // (gridDim, CO, HID)
  int PW, W, P, K2out, NJ, CO, act_in, tiles_per_plane, ntiles;
};

template <int C, int HID, int NPX>
__global__ void __launch_bounds__(NPX * 2) k_proj_bwd(ProjBwdArgs a) {
  constexpr int NW = NPX / 32;
  constexpr int NT = NW * 64;
  constexpr int KS = C / 2;
  constexpr int MT = C / 32;
  constexpr int PITCH = NPX + 4;
  constexpr int NCH = HID / 64;
  constexpr int TILES = 2 * MT;                              // 32x32 tiles of one chunk's dW1
  constexpr int KSPLIT = (NW >= TILES) ? NW / TILES : 1;
  static_assert(NW >= TILES, "one dW1 tile job per wave");
  constexpr int PXK = NPX / KSPLIT;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* xs = smem;                      // C x PITCH   : a = act(u_L); later the gout tile
  float* dps = xs + C * PITCH;           // 64 x PITCH  : gelu(P1) then dP1 of the current chunk
  float* douts = dps + 64 * PITCH;       // CO x NPX
  float* b1s = douts + PROJ_MAXCO * NPX; // HID
  float* w2s = b1s + HID;                // CO x HID
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, half = lane >> 5;
  const int l15 = lane & 15, quad = lane >> 4;
  const int hl = tid & 63, qt = tid >> 6;   // thread-sum mapping: hidden row, 32-px quarter

  for (int i = tid; i < HID; i += NT) b1s[i] = a.b1[i];
  for (int i = tid; i < a.CO * HID; i += NT) w2s[i] = a.w2[i];

  f32x16 dw1acc[NCH];
#pragma unroll
  for (int ch = 0; ch < NCH; ++ch)
#pragma unroll
    for (int r = 0; r < 16; ++r) dw1acc[ch][r] = 0.f;
  float sdb1[NCH], sdw2[NCH][PROJ_MAXCO];
#pragma unroll
  for (int ch = 0; ch < NCH; ++ch) {
    sdb1[ch] = 0.f;
#pragma unroll
    for (int co = 0; co < PROJ_MAXCO; ++co) sdw2[ch][co] = 0.f;
  }
  const int tl = wave % TILES, kp = wave / TILES;
  const int dmt = tl / MT, dnt = tl % MT;   // dW1 tile: hidden 32-block, channel 32-block

  for (int tile = blockIdx.x; tile < a.ntiles; tile += gridDim.x) {
    const int b = tile / a.tiles_per_plane;
    const int px0 = (tile % a.tiles_per_plane) * NPX;
    const float* xb = a.x + (size_t)b * C * a.PW + px0;
    for (int idx = tid; idx < C * (NPX / 4); idx += NT) {
      const int c = idx / (NPX / 4), q = idx % (NPX / 4);
      float4 v = ld4(xb + (size_t)c * a.PW + 4 * q);
      if (a.act_in) { v.x = gelu_f(v.x); v.y = gelu_f(v.y); v.z = gelu_f(v.z); v.w = gelu_f(v.w); }
      st4(xs + c * PITCH + 4 * q, v);
    }
    for (int idx = tid; idx < a.CO * NPX; idx += NT) {
      const int co = idx / NPX, p = idx % NPX;
      const float v = a.dy[((size_t)b * a.CO + co) * a.PW + px0 + p];
      douts[co * NPX + p] = v;
    }
    __syncthreads();
    const int n0 = wave * 32;
    float dyl[PROJ_MAXCO];
#pragma unroll
    for (int co = 0; co < PROJ_MAXCO; ++co) dyl[co] = (co < a.CO) ? douts[co * NPX + n0 + l31] : 0.f;

    f32x16 acc2[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc2[m][r] = 0.f;

#pragma unroll
    for (int ch = 0; ch < NCH; ++ch) {
      // ---- A1: recompute P1 chunk; gl -> LDS; dP1 stays in registers -----
      f32x16 acc[2];
#pragma unroll
      for (int r = 0; r < 16; ++r) { acc[0][r] = 0.f; acc[1][r] = 0.f; }
      const float* wp = a.w1p + (size_t)ch * 2 * KS * 64 + lane;
#pragma unroll 8
      for (int s = 0; s < KS; ++s) {
        const float bf = xs[(2 * s + half) * PITCH + n0 + l31];
        acc[0] = mfma32(wp[s * 64], bf, acc[0]);
        acc[1] = mfma32(wp[(KS + s) * 64], bf, acc[1]);
      }
#pragma unroll
      for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int hrow = m * 32 + acc_row32(r, half);
          const int hid = ch * 64 + hrow;
          const float p = acc[m][r] + b1s[hid];
          float t = 0.f;
#pragma unroll
          for (int co = 0; co < PROJ_MAXCO; ++co)
            if (co < a.CO) t += w2s[co * HID + hid] * dyl[co];
          dps[hrow * PITCH + n0 + l31] = gelu_f(p);
          acc[m][r] = gelu_grad_f(p) * t;
        }
      __syncthreads();
      // ---- A2: dW2[co][hid] += sum_px gl[hid][px] * dy[co][px] -----------
      {
        const float* gr = dps + hl * PITCH + qt * 32;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float4 gv = ld4(gr + 4 * j);
#pragma unroll
          for (int co = 0; co < PROJ_MAXCO; ++co)
            if (co < a.CO) {
              const float4 dv = ld4(douts + co * NPX + qt * 32 + 4 * j);
              sdw2[ch][co] += gv.x * dv.x + gv.y * dv.y + gv.z * dv.z + gv.w * dv.w;
            }
        }
      }
      __syncthreads();
      // ---- A3: dP1 -> LDS; dx += W1^T dP1 with dP1 taken straight from the
      //      accumulator registers (k = hidden row = accumulator row index) ---
#pragma unroll
      for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int hrow = m * 32 + acc_row32(r, half);
          dps[hrow * PITCH + n0 + l31] = acc[m][r];
          const float* wr = a.w1 + (size_t)(ch * 64 + hrow) * C + l31;
#pragma unroll
          for (int mc = 0; mc < MT; ++mc) acc2[mc] = mfma32(wr[mc * 32], acc[m][r], acc2[mc]);
        }
      __syncthreads();
      // ---- B: db1, and dW1[hid][c] += sum_px dP1[hid][px] a[c][px] -------
      {
        const float* gr = dps + hl * PITCH + qt * 32;
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float4 gv = ld4(gr + 4 * j);
          s += (gv.x + gv.y) + (gv.z + gv.w);
        }
        sdb1[ch] += s;
        const float* ga = dps + (dmt * 32 + l31) * PITCH + kp * PXK + 4 * half;
        const float* ab = xs + (dnt * 32 + l31) * PITCH + kp * PXK + 4 * half;
#pragma unroll 4
        for (int q = 0; q < PXK / 8; ++q) {
          const float4 av = ld4(ga + 8 * q);
          const float4 bv = ld4(ab + 8 * q);
          dw1acc[ch] = mfma32(av.x, bv.x, dw1acc[ch]);
          dw1acc[ch] = mfma32(av.y, bv.y, dw1acc[ch]);
          dw1acc[ch] = mfma32(av.z, bv.z, dw1acc[ch]);
          dw1acc[ch] = mfma32(av.w, bv.w, dw1acc[ch]);
        }
      }
      __syncthreads();
    }

    // ---- dx epilogue: (x act') , store, row DFT for the last block's spectral backward
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int c = m * 32 + acc_row32(r, half);
        float v = acc2[m][r];
        if (a.act_in) v *= gelu_grad_f(a.x[((size_t)b * C + c) * a.PW + px0 + n0 + l31]);
        a.gout[((size_t)b * C + c) * a.PW + px0 + n0 + l31] = v;
        if (a.x1g) xs[c * PITCH + n0 + l31] = v;
      }
    if (a.x1g) {
      __syncthreads();
      const int R = NPX / a.W;
      const int njobs = (C / 16) * R * a.NJ;
      for (int job = wave; job < njobs; job += NW) {
        const int nt = job % (C / 16);
        const int rr = (job / (C / 16)) % R;
        const int jt = job / ((C / 16) * R);
        f32x4 d = {0.f, 0.f, 0.f, 0.f};
        const float* tf = a.tfwd + (size_t)(jt * 16 + l15) * a.W + quad;
        const float* xr = xs + (nt * 16 + l15) * PITCH + rr * a.W + quad;
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
    __syncthreads();
  }

  // ---- partial slabs -------------------------------------------------------
#pragma unroll
  for (int ch = 0; ch < NCH; ++ch) {
    float* dst = a.dw1_part + ((size_t)blockIdx.x * KSPLIT + kp) * HID * C;
#pragma unroll
    for (int r = 0; r < 16; ++r)
      dst[(size_t)(ch * 64 + dmt * 32 + acc_row32(r, half)) * C + dnt * 32 + l31] = dw1acc[ch][r];
  }
  // thread-sums: reduce the NW quarter-threads of each hidden row through LDS
  __syncthreads();
  float* red = smem;  // NW x (NCH*64) x (1 + CO)
  for (int ch = 0; ch < NCH; ++ch) {
    red[(qt * NCH * 64 + ch * 64 + hl) * (1 + PROJ_MAXCO)] = sdb1[ch];
    for (int co = 0; co < PROJ_MAXCO; ++co)
      red[(qt * NCH * 64 + ch * 64 + hl) * (1 + PROJ_MAXCO) + 1 + co] = sdw2[ch][co];
  }
  __syncthreads();
  for (int h = tid; h < HID; h += NT) {
    float s1 = 0.f, s2[PROJ_MAXCO] = {0.f, 0.f, 0.f, 0.f};
    for (int q = 0; q < NW; ++q) {
      s1 += red[(q * HID + h) * (1 + PROJ_MAXCO)];
      for (int co = 0; co < PROJ_MAXCO; ++co) s2[co] += red[(q * HID + h) * (1 + PROJ_MAXCO) + 1 + co];
    }
    a.db1_part[(size_t)blockIdx.x * HID + h] = s1;
    for (int co = 0; co < a.CO; ++co) a.dw2_part[((size_t)blockIdx.x * a.CO + co) * HID + h] = s2[co];
  }
}

// db2[co] = sum over (b, px) of dy  -- tiny, memory-trivial (CO planes)
__global__ void k_sum_planes(const float* __restrict__ dy, float* __restrict__ part, int B, int CO, int PW) {
  // grid = (nblk, CO); each block sums a strided share of (b, px) for channel co
  const int co = blockIdx.y;
  float s = 0.f;
  const size_t n = (size_t)B * PW;
  for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (size_t)gridDim.x * blockDim.x) {
    const size_t b = e / PW, p = e % PW;
    s += dy[(b * CO + co) * PW + p];
  }
  for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
  __shared__ float sh[16];
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    float t = 0.f;
    for (int i = 0; i < (int)(blockDim.x >> 6); ++i) t += sh[i];
    part[(size_t)blockIdx.x * CO + co] = t;
  }
}
