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
// Wave (mt, nt) owns the 32x32 sub-tile (channels 32*mt.., pixels 32*nt..) of dx and one
// (32x32 tile, pixel-range) job of dW.  Weight/bias gradients are accumulated in
// registers across the persistent tile loop and written once per workgroup as partial
// slabs (summed by k_reduce_slabs: deterministic, no float atomics).
//
// Reference semantics: autograd of fno_block.py:123-170 + spectral_convolution.py:303-347
// (formulas: SURVEY.md Appendix A, oracle/fno_oracle.py::spectral_conv_A_backward).
#pragma once
#include "fno_dev.h"

#ifndef FNO_OCC_BB
#define FNO_OCC_BB 2   // measured: 2 (no spills, 1 workgroup/CU) beats 4 (spills) on MI355X
#endif

struct BlkBwdArgs {
  const float* g;      // (B, C, PW)
  const float* uin;    // (B, C, PW)
  const float* w;      // (C, C) [o][i]
  const float* zg;     // (B, P, K2in, C, 2) or null
  const float* gadd;   // (B, C, PW) added to dx before the activation derivative, or null
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
  int loose;           // rows do not tile the pixel tile (see PwFwdArgs.loose); k_block_bwd_x3 only
  const float* lw;     // LIFT variant of k_block_bwd_x3 (block 0 of a model with a lifting layer): u_0 = lw xin + lb is
  const float* lb;     // recomputed from the model input instead of read (lw (C, CL), lb (C)); uin is ignored
  int kch;             // loose rows: kept last-dim modes per K-extension chunk (0 = all at once); the spectral rows and
                       // the table of a chunk are staged right before it is applied, so many kept modes still fit LDS
  int tiles_per_plane, ntiles;
  const unsigned* drop_seed;   // k_block_bwd_t<.., DROPK = true>: the forward's dropout words and rate (fno_dev.h: drop_cfg)
  float drop_p;
  // two-term fp16 GEMMs (k_block_bwd_t / _g2 with NT3 = 2; fno_dev.h "h2"): device scalars bounding |g| and |u| (|u_0| with a
  // recomputed lifting); gmax_out (any variant): max |gout| is published there for the next kernel of the chain
  const float* gmax_in;
  const float* umax;
  float* gmax_out;
  int kx16;           // k_block_bwd_g2: the spectral K-extension as ONE 16-deep bf16x3 block on the matrix pipe (<= 8 kept last-dim
                      // modes; table / spectral-row images in LDS) instead of 2-deep fp32 MFMAs on the VALU lanes
  int rev = 0;        // k_block_bwd_g2: walk the tiles from the last to the first (the most recently written part of g - what the
                      // Infinity Cache still holds of the producer's output - is read first)
  int lines = 0;      // k_block_bwd_g2: u is loaded and gout stored in whole 128 / 256-byte lines (the host adds 16 KB of staging
                      // to the LDS size; round 6)
};

template <int C, int NPX>
struct BlkBwdCfg {
  static constexpr int NTN = NPX / 32;
  static constexpr int MT = C / 32;
  static constexpr int NW = MT * NTN;
  static constexpr int TILES = MT * MT;
  static constexpr int KSPLIT = NW / TILES;
  static_assert(NW % TILES == 0 && KSPLIT >= 1, "C <= NPX required");
};

template <int C, int NPX>
__global__ void __launch_bounds__((C / 32) * (NPX / 32) * 64, FNO_OCC_BB) k_block_bwd(BlkBwdArgs a) {
  using Cfg = BlkBwdCfg<C, NPX>;
  constexpr int NTN = Cfg::NTN, MT = Cfg::MT, NW = Cfg::NW, TILES = Cfg::TILES, KSPLIT = Cfg::KSPLIT;
  constexpr int NT = NW * 64;
  constexpr int KS = C / 2;
  constexpr int PITCH = NPX + 4;
  constexpr int DBPX = NPX / (NT / C);                 // pixels per thread in the dbias sums
  static_assert(NT % C == 0 && DBPX % 4 == 0, "dbias thread mapping");
  constexpr int PXK = NPX / KSPLIT;
  constexpr int LJ = (C / 16 + NW - 1) / NW;           // lifting-epilogue jobs per wave
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* gs = smem;                 // C x PITCH : g, later the gout tile
  float* us = smem + C * PITCH;     // C x PITCH : u_l, then a_l in place
  float* xls = us + C * PITCH;      // 8 x PITCH : lifting input rows (block 0 only)
  float* tinv_s = xls + (a.xin ? 8 * PITCH : 0);   // 2*K2in x W : row inverse table (if zg)
  const int R = NPX / a.W;
  float* zs = tinv_s + (a.zg ? 2 * a.K2in * a.W : 0);   // R x K2in x C x 2 : this tile's spectral gradient rows
  float* tfwd_s = zs + (a.zg ? R * a.K2in * C * 2 : 0);  // 16*NJ x W : row forward table (if x1g)

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, half = lane >> 5;
  const int l15 = lane & 15, quad = lane >> 4;
  const int mt = wave / NTN, nt = wave % NTN;
  const int n0 = nt * 32;
  const int dtl = wave % TILES, dkp = wave / TILES;     // dW job
  const int dmt = dtl / MT, dnt = dtl % MT;

  if (a.zg)
    for (int i = tid; i < 2 * a.K2in * a.W; i += NT) tinv_s[i] = a.tinv[i];
  if (a.x1g)
    for (int i = tid; i < 16 * a.NJ * a.W; i += NT) tfwd_s[(i / a.W) * (a.W + 4) + i % a.W] = a.tfwd[i];
  const int zcount4 = a.zg ? R * a.K2in * C / 2 : 0;

  // A fragments of W^T: A[i][k = o] = W[o][i]
  float afrag[KS];
#pragma unroll
  for (int s = 0; s < KS; ++s) afrag[s] = a.w[(2 * s + half) * C + mt * 32 + l31];

  f32x16 dwacc;
#pragma unroll
  for (int r = 0; r < 16; ++r) dwacc[r] = 0.0f;
  float dbsum = 0.0f;                                  // thread (c = tid % C, part = tid / C)
  f32x4 dl[LJ];
#pragma unroll
  for (int j = 0; j < LJ; ++j) dl[j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // register-staged prefetch: the NEXT tile's g, u and Zg rows are in flight during this tile
  TilePrefetch<NPX, NT, C, C> pfg, pfu;
  float4 zv = make_float4(0.f, 0.f, 0.f, 0.f);
  auto issue = [&](int tile) {
    const int b = tile / a.tiles_per_plane;
    const int px0 = (tile % a.tiles_per_plane) * NPX;
    int t = tid;
    asm volatile("" : "+v"(t));      // (keeps the per-lane 64-bit prefetch addresses out of the loop-invariant registers: k_pw_fwd_x3)
    pfg.issue(a.g + (size_t)b * C * a.PW + px0, a.PW, t);
    pfu.issue(a.uin + (size_t)b * C * a.PW + px0, a.PW, t);
    if (t < zcount4) zv = ld4(a.zg + ((size_t)b * a.P + px0 / a.W) * a.K2in * C * 2 + 4 * t);
  };
  if ((int)blockIdx.x < a.ntiles) issue(blockIdx.x);

  int tslot = 0;
  FNO_TRACE_IF(false);
  for (int tile = blockIdx.x; tile < a.ntiles; tile += gridDim.x) {
    const int b = tile / a.tiles_per_plane;
    const int px0 = (tile % a.tiles_per_plane) * NPX;
    FNO_STAMP(tslot + 0);
    pfg.commit(gs, false, tid);
    pfu.commit(us, false, tid);
    if (tid < zcount4) st4(zs + 4 * tid, zv);
    for (int i = tid + NT; i < zcount4; i += NT)      // more spectral rows than threads (short rows, many modes)
      st4(zs + 4 * i, ld4(a.zg + ((size_t)b * a.P + px0 / a.W) * a.K2in * C * 2 + 4 * i));
    if (a.xin) stage_rows<NPX, NT>(xls, a.xin + (size_t)b * a.CL * a.PW + px0, a.PW, a.CL, a.CL, false, tid);
    FNO_STAMP(tslot + 1);
    __syncthreads();
    FNO_STAMP(tslot + 2);
    if (tile + (int)gridDim.x < a.ntiles) issue(tile + gridDim.x);

    {  // dbias[c] partial: this thread's TPX pixels of row c
      const float* gr = gs + (tid % C) * PITCH + (tid / C) * DBPX;
#pragma unroll
      for (int j = 0; j < DBPX / 4; ++j) {
        const float4 gv = ld4(gr + 4 * j);
        dbsum += (gv.x + gv.y) + (gv.z + gv.w);
      }
    }
    FNO_STAMP(tslot + 3);
    // ---- dx GEMM (+ row inverse DFT of the spectral gradient) -------------
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
#pragma unroll
    for (int s = 0; s < KS; ++s) acc = mfma32(afrag[s], gs[(2 * s + half) * PITCH + n0 + l31], acc);
    if (a.zg) {
      const float* zr = zs + (((n0 / a.W) * a.K2in) * C + mt * 32 + l31) * 2 + half;
      const float* tv = tinv_s + half * a.W + n0 % a.W + l31;
#pragma unroll 2
      for (int s = 0; s < a.K2in; ++s) acc = mfma32(zr[s * C * 2], tv[2 * s * a.W], acc);
    }
    if (a.gadd) {
      const float* ap = a.gadd + ((size_t)b * C + mt * 32 + 4 * half) * a.PW + px0 + n0 + l31;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] += ap[(size_t)((r & 3) + 8 * (r >> 2)) * a.PW];
    }
    FNO_STAMP(tslot + 4);
    // ---- activation derivative; us becomes a_l = act(u_l) in place --------
    // (each (i, px) element of us is touched only by the lane that owns it)
    {
      float* up = us + (mt * 32 + 4 * half) * PITCH + n0 + l31;
      float* gp = a.gout ? a.gout + ((size_t)b * C + mt * 32 + 4 * half) * a.PW + px0 + n0 + l31 : nullptr;
      if (a.act_in) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int ro = (r & 3) + 8 * (r >> 2);          // acc_row32(r, half) - 4 * half
          float gl, dg;
          gelu_both(up[ro * PITCH], gl, dg);
          acc[r] *= dg;
          up[ro * PITCH] = gl;
        }
      }
      if (gp) {
#pragma unroll
        for (int r = 0; r < 16; ++r) gp[(size_t)((r & 3) + 8 * (r >> 2)) * a.PW] = acc[r];
      }
    }
    FNO_STAMP(tslot + 5);
    __syncthreads();
    FNO_STAMP(tslot + 6);

    // ---- dW[o][i] += sum_px g[o][px] a[i][px] ------------------------------
    // k index of MFMA #t in group q: lane-half h <-> pixel 8q + 4h + t (one b128 per 4 MFMAs)
    {
      const float* ga = gs + (dmt * 32 + l31) * PITCH + dkp * PXK + 4 * half;
      const float* ab = us + (dnt * 32 + l31) * PITCH + dkp * PXK + 4 * half;
#pragma unroll 2
      for (int q = 0; q < PXK / 8; ++q) {
        const float4 av = ld4(ga + 8 * q);
        const float4 bv = ld4(ab + 8 * q);
        dwacc = mfma32(av.x, bv.x, dwacc);
        dwacc = mfma32(av.y, bv.y, dwacc);
        dwacc = mfma32(av.z, bv.z, dwacc);
        dwacc = mfma32(av.w, bv.w, dwacc);
      }
    }

    FNO_STAMP(tslot + 7);
    if (a.x1g || a.xin) {
      __syncthreads();  // dW GEMM done with gs
      FNO_STAMP(tslot + 8);
      {
        float* gq = gs + (mt * 32 + 4 * half) * PITCH + n0 + l31;
#pragma unroll
        for (int r = 0; r < 16; ++r) gq[((r & 3) + 8 * (r >> 2)) * PITCH] = acc[r];
      }
      FNO_STAMP(tslot + 9);
      __syncthreads();
      FNO_STAMP(tslot + 10);
      if (a.x1g) row_dft_epilogue<C, NPX, NW>(gs, tfwd_s, a.W + 4, a.x1g, b, px0, a.P, a.W, a.K2out, a.NJ, wave, lane);
      if (a.xin) {
        // dl[c][n] += sum_px gout[c][px] * xext[n][px],  xext = [x_in rows | ones | 0..]
#pragma unroll
        for (int j = 0; j < LJ; ++j) {
          const int jm = wave + j * NW;
          if (jm < C / 16) {
            const float* ar = gs + (jm * 16 + l15) * PITCH + quad;
            const float* br = xls + (l15 < a.CL ? l15 : 0) * PITCH + quad;
            const float cst = l15 == a.CL ? 1.0f : 0.0f;
            for (int s = 0; s < NPX / 4; ++s) {
              const float bf = (l15 < a.CL) ? br[4 * s] : cst;
              dl[j] = mfma16(ar[4 * s], bf, dl[j]);
            }
          }
        }
      }
    }
    FNO_STAMP(tslot + 11);
    __syncthreads();
    tslot += 12;
  }

  // ---- write partial slabs ---------------------------------------------------
  {
    float* dst = a.dw_part + ((size_t)blockIdx.x * KSPLIT + dkp) * C * C;
#pragma unroll
    for (int r = 0; r < 16; ++r) dst[(dmt * 32 + acc_row32(r, half)) * C + dnt * 32 + l31] = dwacc[r];
  }
  __syncthreads();
  smem[tid] = dbsum;                         // [part][c]
  __syncthreads();
  if (tid < C) {
    float v = 0.f;
    for (int k = 0; k < NT / C; ++k) v += smem[k * C + tid];
    a.db_part[(size_t)blockIdx.x * C + tid] = v;
  }
  if (a.xin) {
#pragma unroll
    for (int j = 0; j < LJ; ++j) {
      const int jm = wave + j * NW;
      if (jm < C / 16) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
          a.dwl_part[((size_t)blockIdx.x * C + jm * 16 + quad * 4 + r) * 16 + l15] = dl[j][r];
      }
    }
  }
}

// ---------------------------------------------------------------------------
// Gradients of a lifting layer on its own (y = W x + b, x (B, CL <= 4, PW) -> y (B, C, PW); tfno.py:11-20,
// and the composed `fc0` + Re-conditioning front of the PINO observers, pinobserver.py:205-207):
//   dW[c][i] = sum_{b,px} dy[c][px] x[i][px],  db[c] = sum dy[c][px]      (no input gradient: x is data)
// Same MFMA 16x16x4 scheme as the block-0 epilogue of k_block_bwd: B operand = [x rows | ones | 0..].
struct LiftBwdArgs {
  const float* dy;       // (B, C, PW)
  const float* xin;      // (B, CL, PW)
  float* dwl_part;       // (gridDim, C, 16): col i < CL = dW, col CL = db
  int CL, PW, tiles_per_plane, ntiles;
};
template <int C, int NPX>
__global__ void __launch_bounds__(256) k_lift_bwd(LiftBwdArgs a) {
  constexpr int NW = 4, NT = NW * 64, PITCH = NPX + 4;
  static_assert(C / 16 <= NW, "one 16-channel job per wave");
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* gs = smem;                 // C x PITCH
  float* xls = gs + C * PITCH;      // 8 x PITCH
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l15 = lane & 15, quad = lane >> 4;
  f32x4 dl = {0.f, 0.f, 0.f, 0.f};
  TilePrefetch<NPX, NT, C, C> pf;
  auto issue = [&](int tile) {
    pf.issue(a.dy + (size_t)(tile / a.tiles_per_plane) * C * a.PW + (tile % a.tiles_per_plane) * NPX, a.PW, tid);
  };
  if ((int)blockIdx.x < a.ntiles) issue(blockIdx.x);
  for (int tile = blockIdx.x; tile < a.ntiles; tile += gridDim.x) {
    const int b = tile / a.tiles_per_plane;
    const int px0 = (tile % a.tiles_per_plane) * NPX;
    pf.commit(gs, false, tid);
    stage_rows<NPX, NT>(xls, a.xin + (size_t)b * a.CL * a.PW + px0, a.PW, a.CL, a.CL, false, tid);
    __syncthreads();
    if (tile + (int)gridDim.x < a.ntiles) issue(tile + gridDim.x);
    if (wave < C / 16) {
      const float* ar = gs + (wave * 16 + l15) * PITCH + quad;
      const float* br = xls + (l15 < a.CL ? l15 : 0) * PITCH + quad;
      const float cst = l15 == a.CL ? 1.0f : 0.0f;
#pragma unroll 8
      for (int s = 0; s < NPX / 4; ++s) {
        const float bf = (l15 < a.CL) ? br[4 * s] : cst;
        dl = mfma16(ar[4 * s], bf, dl);
      }
    }
    __syncthreads();
  }
  if (wave < C / 16) {
#pragma unroll
    for (int r = 0; r < 4; ++r) a.dwl_part[((size_t)blockIdx.x * C + wave * 16 + quad * 4 + r) * 16 + l15] = dl[r];
  }
}

// out[row][col] = sum_s part[s][row][col]: deterministic two-level slab reduction.
// block = (64 elements, 16 slab lanes); element e = (row, col), col < ncols.
__global__ void __launch_bounds__(1024) k_reduce_slabs(const float* __restrict__ part, float* __restrict__ out,
                                                       int nslab, int n, int ld_out, int ncols, int ld_in) {
  __shared__ float sh[16][65];
  const int e = blockIdx.x * 64 + threadIdx.x;
  const int sy = threadIdx.y;
  float s = 0.f;
  if (e < n) {
    const int row = e / ncols, col = e % ncols;
    const size_t slab = (size_t)(n / ncols) * ld_in;
    const float* p = part + (size_t)row * ld_in + col;
    for (int k = sy; k < nslab; k += 16) s += p[k * slab];
  }
  sh[sy][threadIdx.x] = s;
  __syncthreads();
  if (sy == 0 && e < n) {
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) t += sh[k][threadIdx.x];
    out[(size_t)(e / ncols) * ld_out + e % ncols] = t;
  }
}

// All slab reductions of one backward pass in ONE launch: blockIdx.y selects the job.
struct ReduceJob { const float* part; float* out; int nslab, n, ld_out, ncols, ld_in; };
struct ReduceJobs { ReduceJob j[24]; int count; };
// NSY slab lanes per column group: 64 (1024 threads, 4 slab loads per thread, a 6-step tree).  16 (256 threads, 16 loads per
// thread, a 4-step tree) was measured in round 4 and is SLOWER: 28 vs 19.5 us at BASELINE config 2, 43 vs 23 us at config 4.
template <int NSY>
__global__ void __launch_bounds__(16 * NSY) k_reduce_jobs(ReduceJobs jobs) {
  // block (16, NSY): 16 column groups x NSY slab lanes.  Many small workgroups (a 64x64 weight gradient
  // alone gives 64 of them) keep every CU loading; each thread has <= 8 independent 16-B loads in flight.
  __shared__ float4 sh[NSY][17];
  const ReduceJob jb = jobs.j[blockIdx.y];
  const int tx = threadIdx.x, sy = threadIdx.y;
  const size_t slab = (size_t)(jb.n / jb.ncols) * jb.ld_in;
  if (jb.ncols % 4 == 0 && jb.ld_in % 4 == 0) {
    const int ng = jb.n / 4;
    for (int g0 = blockIdx.x * 16; g0 < ng; g0 += gridDim.x * 16) {
      const int g = g0 + tx;
      const int row = (4 * g) / jb.ncols, col = (4 * g) % jb.ncols;
      float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
      if (g < ng) {
        const float* p = jb.part + (size_t)row * jb.ld_in + col;
#pragma unroll 8
        for (int k = sy; k < jb.nslab; k += NSY) {
          const float4 v = ld4(p + k * slab);
          s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
        }
      }
      sh[sy][tx] = s;
      __syncthreads();
      for (int off = NSY / 2; off > 0; off >>= 1) {       // fixed-order tree over the slab lanes
        if (sy < off) {
          const float4 v = sh[sy + off][tx];
          float4 t = sh[sy][tx];
          t.x += v.x; t.y += v.y; t.z += v.z; t.w += v.w;
          sh[sy][tx] = t;
        }
        __syncthreads();
      }
      if (sy == 0 && g < ng) {
        const float4 t = sh[0][tx];
        float* o = jb.out + (size_t)row * jb.ld_out + col;     // destination rows may be unaligned (flat bucket offsets)
        o[0] = t.x; o[1] = t.y; o[2] = t.z; o[3] = t.w;
      }
      __syncthreads();
    }
    return;
  }
  // scalar columns (bias-type jobs): 16 elements x NSY slab lanes
  for (int e0 = blockIdx.x * 16; e0 < jb.n; e0 += gridDim.x * 16) {
    const int e = e0 + tx;
    float s = 0.f;
    if (e < jb.n) {
      const int row = e / jb.ncols, col = e % jb.ncols;
      const float* p = jb.part + (size_t)row * jb.ld_in + col;
#pragma unroll 8
      for (int k = sy; k < jb.nslab; k += NSY) s += p[k * slab];
    }
    sh[sy][tx].x = s;
    __syncthreads();
    for (int off = NSY / 2; off > 0; off >>= 1) {
      if (sy < off) sh[sy][tx].x += sh[sy + off][tx].x;
      __syncthreads();
    }
    if (sy == 0 && e < jb.n) jb.out[(size_t)(e / jb.ncols) * jb.ld_out + e % jb.ncols] = sh[0][tx].x;
    __syncthreads();
  }
}

// Input gradient of a lifting layer (y = W x + b, W (C, CL <= 4)): dx[k][px] = sum_c W[c][k] g[c][px]
// (run_control.py:186-224 differentiates the observer down to its input field).  One thread = 4 pixels of one sample.
__global__ void __launch_bounds__(256) k_lift_dx(const float* __restrict__ g, const float* __restrict__ lw, float* __restrict__ dx,
                                                 int C, int CL, size_t PW, size_t n4) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
    const size_t b = (4 * i) / PW, px = 4 * i - b * PW;
    float4 acc[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) acc[k] = make_float4(0.f, 0.f, 0.f, 0.f);
    const float* gp = g + b * C * PW + px;
#pragma unroll 8
    for (int c = 0; c < C; ++c) {
      const float4 gv = ld4(gp + (size_t)c * PW);
#pragma unroll
      for (int k = 0; k < 4; ++k)
        if (k < CL) {
          const float w = lw[c * CL + k];
          acc[k].x = fmaf(w, gv.x, acc[k].x); acc[k].y = fmaf(w, gv.y, acc[k].y);
          acc[k].z = fmaf(w, gv.z, acc[k].z); acc[k].w = fmaf(w, gv.w, acc[k].w);
        }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k)
      if (k < CL) st4(dx + (b * CL + k) * PW + px, acc[k]);
  }
}
