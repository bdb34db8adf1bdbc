// Fused pointwise (1x1 conv) + truncated-spectrum row passes, forward direction.
//
// One workgroup owns a tile of NPX consecutive pixels of one sample's (P x W)
// plane, for ALL channels; wave (mt, nt) owns the 32x32 output sub-tile
// (channels 32*mt.., pixels 32*nt..).  Per tile (persistent grid-stride loop, the
// NEXT tile's HBM loads are in flight while the current one is computed):
//   1. stage x[b, 0:CIN, tile] HBM -> registers -> LDS (16 B/lane coalesced), applying
//      the previous block's GELU on the way when act_in (activations are stored
//      PRE-activation so that backward never needs a second copy);
//   2. D[o][px] = sum_c W[o][c] a[c][px]                      (fp32 MFMA 32x32x2)
//               + sum_j Z[b,row,j,o] * Tinv[j][w]             (row inverse DFT folded in
//                                                              as a K-extension of the GEMM)
//               + bias[o];
//   3. u = D is written to HBM; act_out(u) goes back to LDS;
//   4. row forward DFT of the tile, truncated to the kept bins, on fp32 MFMA
//      16x16x4:  X1[b,row,k2,o] = sum_w a[o][w] * Tfwd[j][w]   -> HBM (small).
// So a whole FNO block costs one read and one write of the activation.  Twiddle
// tables live in LDS for the lifetime of the workgroup.
//
// Replaces, for the reference's default FNO path: Lifting (tfno.py:19-20), the
// skip conv + residual add + GELU of FNOBlocks.forward (fno_block.py:131,147-150),
// the last-dim pass of irfftn + bias (spectral_convolution.py:342-345) and the
// last-dim pass of the NEXT block's rfftn (spectral_convolution.py:324).
#pragma once
#include "fno_dev.h"

#ifndef FNO_OCC_PW
#define FNO_OCC_PW 4   // lifting / no-GEMM variants (CIN <= 4) ask for 6 waves per SIMD: 3 workgroups per CU fit their 42 KB of LDS
#endif
#ifndef FNO_OCC_PWX
#define FNO_OCC_PWX 2   // measured: 1 workgroup/CU without spills (0.175 ms) beats 2 with spills (0.27 ms)
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
  const float* add;   // (B, COUT, PW) tensor added to the output before it is stored, or null
  int PW, W, P, K2in, K2out, NJ;
  int act_in, act_out;
  int relu_out;       // k_pw_fwd_x3<.., RELU = true>: store max(u, 0)
  int loose;          // rows do not tile the pixel tile (W >= 32, any remainder): spectral rows per tile vary, no x1 epilogue
  const float* lw;    // LIFT variant of k_pw_fwd_x3 (block 0 computes u_0 = lw x + lb itself): lifting weight (C, CL) ...
  const float* lb;    // ... and bias (C); x is then the (B, CL, PW) model input
  int CL;
  int tiles_per_plane, ntiles;
  const float* xmax;  // k_blk_fwd_t<.., NT3 = 2>: device scalar, a bound of |x| (of the model input with a fused lifting)
  float* ubound;      // k_blk_fwd_t<LIFT, NT3 = 2>: the bound of |u_0| it derived from xmax and the lifting parameters is left here
  float* umax;        // if set: max |u| of what this launch stores is published here (atomic max of the float pattern; the
                      // two-term fp16 GEMMs of the consumer scale their operand by it: fno_dev.h, "h2")
  int share32 = 0;    // k_blk_fwd_t with two workgroups per CU: 32nds of a CU's tiles that go to the workgroup dispatched FIRST
                      // (0 = even split; pair_share() in fno_dev.h)
  int rev = 0;        // k_blk_fwd_s: walk the tiles from the last to the first (the input's most recently written part - what the
                      // Infinity Cache still holds of the producer's output - is read first)
};

// dynamic LDS bytes needed by k_pw_fwd<CIN, COUT, NPX>
static inline size_t pw_fwd_lds_bytes(int cin, int cout, int npx, int W, int K2in, int NJ, bool has_z, bool has_x1) {
  const int rows = ((cin + 1) & ~1) > cout ? ((cin + 1) & ~1) : cout;
  size_t fl = (size_t)rows * (npx + 4);
  if (has_z) fl += (size_t)2 * K2in * W + (size_t)(npx / W) * K2in * cout * 2;
  if (has_x1) fl += (size_t)16 * NJ * (W + 4);
  return fl * 4;
}

template <int CIN, int COUT, int NPX>
__global__ void __launch_bounds__((COUT / 32) * (NPX / 32) * 64, CIN <= 4 ? (COUT * NPX >= 64 * 128 ? 2 : 4) : FNO_OCC_PW) k_pw_fwd(PwFwdArgs a) {
  constexpr int NTN = NPX / 32;          // pixel sub-tiles
  constexpr int MT = COUT / 32;          // channel sub-tiles
  constexpr int NW = MT * NTN;
  constexpr int NT = NW * 64;
  constexpr int CINP = (CIN + 1) & ~1;
  constexpr int KS = CINP / 2;
  constexpr int PITCH = NPX + 4;
  constexpr int ROWS = CINP > COUT ? CINP : COUT;
  static_assert(COUT % 32 == 0 && NPX % 32 == 0, "tile shape");
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* xs = smem;                                   // ROWS x PITCH
  float* tinv_s = xs + ROWS * PITCH;                  // 2*K2in x W          (if z)
  const int R = NPX / a.W;
  float* zs = tinv_s + (a.z ? 2 * a.K2in * a.W : 0);  // R x K2in x COUT x 2 (if z)
  float* tfwd_s = zs + (a.z ? R * a.K2in * COUT * 2 : 0);   // 16*NJ x W     (if x1)

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, half = lane >> 5;
  const int mt = wave / NTN, nt = wave % NTN;
  const int n0 = nt * 32;
  const bool has_conv = a.x != nullptr;

  if (a.z)
    for (int i = tid; i < 2 * a.K2in * a.W; i += NT) tinv_s[i] = a.tinv[i];
  if (a.x1)
    for (int i = tid; i < 16 * a.NJ * a.W; i += NT) tfwd_s[(i / a.W) * (a.W + 4) + i % a.W] = a.tfwd[i];

  // weight fragments: A[i = o][k = c], constant over all tiles of this workgroup
  float afrag[KS];
#pragma unroll
  for (int s = 0; s < KS; ++s) {
    const int c = 2 * s + half;
    afrag[s] = (has_conv && c < CIN) ? a.w[(mt * 32 + l31) * CIN + c] : 0.0f;
  }

  const int zcount4 = a.z ? R * a.K2in * COUT / 2 : 0;   // float4 pieces of one tile's Z rows
  TilePrefetch<NPX, NT, CIN, CINP> pf;
  float4 zpf = make_float4(0.f, 0.f, 0.f, 0.f);
  auto issue = [&](int tile) {
    const int b = tile / a.tiles_per_plane;
    const int px0 = (tile % a.tiles_per_plane) * NPX;
    if (has_conv) pf.issue(a.x + (size_t)b * CIN * a.PW + px0, a.PW, tid);
    if (tid < zcount4) zpf = ld4(a.z + ((size_t)b * a.P + px0 / a.W) * a.K2in * COUT * 2 + 4 * tid);
  };
  if ((int)blockIdx.x < a.ntiles) issue(blockIdx.x);

  for (int tile = blockIdx.x; tile < a.ntiles; tile += gridDim.x) {
    const int b = tile / a.tiles_per_plane;
    const int px0 = (tile % a.tiles_per_plane) * NPX;

    if (has_conv) pf.commit(xs, a.act_in != 0, tid);
    if (tid < zcount4) st4(zs + 4 * tid, zpf);
    for (int i = tid + NT; i < zcount4; i += NT)     // more Z rows than threads (short rows): straight from L2
      st4(zs + 4 * i, ld4(a.z + ((size_t)b * a.P + px0 / a.W) * a.K2in * COUT * 2 + 4 * i));
    __syncthreads();
    if (tile + (int)gridDim.x < a.ntiles) issue(tile + gridDim.x);   // overlaps everything below

    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
    if (has_conv) {
#pragma unroll
      for (int s = 0; s < KS; ++s) acc = mfma32(afrag[s], xs[(2 * s + half) * PITCH + n0 + l31], acc);
    }
    if (a.z) {
      const int rr = n0 / a.W;
      const float* zr = zs + ((rr * a.K2in) * COUT + mt * 32 + l31) * 2 + half;
      const float* tv = tinv_s + half * a.W + n0 % a.W + l31;
#pragma unroll 2
      for (int s = 0; s < a.K2in; ++s) acc = mfma32(zr[s * COUT * 2], tv[2 * s * a.W], acc);
    }
    __syncthreads();  // all waves are done reading the staged input

    {
      // one base pointer + small row offsets (keeps 16 x 64-bit addresses out of registers)
      float* up = a.u ? a.u + ((size_t)b * COUT + mt * 32 + 4 * half) * a.PW + px0 + n0 + l31 : nullptr;
      float* xp = xs + (mt * 32 + 4 * half) * PITCH + n0 + l31;
      const float* bp = a.bias ? a.bias + mt * 32 + 4 * half : nullptr;
      if (bp) {
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] += bp[(r & 3) + 8 * (r >> 2)];   // row acc_row32(r, half)
      }
      if (a.add) {
        const float* ap = a.add + ((size_t)b * COUT + mt * 32 + 4 * half) * a.PW + px0 + n0 + l31;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] += ap[(size_t)((r & 3) + 8 * (r >> 2)) * a.PW];
      }
      if (up) {
#pragma unroll
        for (int r = 0; r < 16; ++r) up[(size_t)((r & 3) + 8 * (r >> 2)) * a.PW] = acc[r];
      }
      if (a.x1) {
        if (a.act_out) {
#pragma unroll
          for (int r = 0; r < 16; ++r) xp[((r & 3) + 8 * (r >> 2)) * PITCH] = gelu_f(acc[r]);
        } else {
#pragma unroll
          for (int r = 0; r < 16; ++r) xp[((r & 3) + 8 * (r >> 2)) * PITCH] = acc[r];
        }
      }
    }
    if (a.x1) {
      __syncthreads();
      row_dft_epilogue<COUT, NPX, NW>(xs, tfwd_s, a.W + 4, a.x1, b, px0, a.P, a.W, a.K2out, a.NJ, wave, lane);
    }
    __syncthreads();  // xs is restaged by the next tile
  }
}

// ---------------------------------------------------------------------------
// Last-dim forward pass of the STANDALONE spectral convolution (fno_spec_*: dialects B / C and the
// unfused FNO): X1[b,row,k2,c] = sum_w x[b,c,row,w] * Tfwd[k2][w].  Same tile / prefetch / MFMA
// row-DFT machinery as the fused kernels, no GEMM.  4 waves, ~50 KB LDS -> 3 workgroups per CU.
struct RowDftArgs {
  const float* x;      // (B, C, PW)
  float* x1;           // (B, P, K2out, C, 2)
  const float* tfwd;   // (16*NJ, W)
  int PW, W, P, K2out, NJ;
  int tiles_per_plane, ntiles;
  // MOD = 1: the rows of drop(x) (counter-based dropout, fno_dev.h; rno.py:98 spec_conv(dropout(x)))
  const unsigned* drop_seed;
  float drop_p;
  // MOD = 2: the rows of g = x * (ymask > 0) (ReLU derivative read off the forward's output); g itself is also written
  const float* ymask;  // (B, C, PW)
  float* gmasked;      // (B, C, PW)
};
template <int C, int NPX, int MOD = 0>
__global__ void __launch_bounds__(256) k_rowdft_tile(RowDftArgs a) {
  constexpr int NW = 4, NT = NW * 64, PITCH = NPX + 4;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* xs = smem;                      // C x PITCH
  float* tfwd_s = xs + C * PITCH;        // 16*NJ x (W + 4)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int i = tid; i < 16 * a.NJ * a.W; i += NT) tfwd_s[(i / a.W) * (a.W + 4) + i % a.W] = a.tfwd[i];
  TilePrefetch<NPX, NT, C, C> pf;
  TilePrefetch<NPX, NT, MOD == 2 ? C : 1, MOD == 2 ? C : 1> pfy;
  DropCfg dc = {0u, 0u, 1.f};
  if constexpr (MOD == 1) dc = drop_cfg(a.drop_seed, a.drop_p);
  auto issue = [&](int tile) {
    const size_t off = (size_t)(tile / a.tiles_per_plane) * C * a.PW + (tile % a.tiles_per_plane) * NPX;
    pf.issue(a.x + off, a.PW, tid);
    if constexpr (MOD == 2) pfy.issue(a.ymask + off, a.PW, tid);
  };
  if ((int)blockIdx.x < a.ntiles) issue(blockIdx.x);
  for (int tile = blockIdx.x; tile < a.ntiles; tile += gridDim.x) {
    const int b = tile / a.tiles_per_plane;
    const int px0 = (tile % a.tiles_per_plane) * NPX;
    if constexpr (MOD == 1) {
      const size_t e0 = (size_t)b * C * a.PW + px0;
      pf.commit_with(xs, tid, [&](float4& t, int c, int q, int) {
        const size_t e = e0 + (size_t)c * a.PW + 4 * q;
        t.x *= drop_scale(dc, e); t.y *= drop_scale(dc, e + 1); t.z *= drop_scale(dc, e + 2); t.w *= drop_scale(dc, e + 3);
      });
    } else if constexpr (MOD == 2) {
      float* gdst = a.gmasked + (size_t)b * C * a.PW + px0;
      pf.commit_with(xs, tid, [&](float4& t, int c, int q, int i) {
        const float4 y = pfy.v[i];
        t.x = y.x <= 0.f ? 0.f : t.x; t.y = y.y <= 0.f ? 0.f : t.y; t.z = y.z <= 0.f ? 0.f : t.z; t.w = y.w <= 0.f ? 0.f : t.w;   // threshold_backward
        st4(gdst + (size_t)c * a.PW + 4 * q, t);
      });
    } else {
      pf.commit(xs, false, tid);
    }
    __syncthreads();
    if (tile + (int)gridDim.x < a.ntiles) issue(tile + gridDim.x);
    row_dft_epilogue<C, NPX, NW>(xs, tfwd_s, a.W + 4, a.x1, b, px0, a.P, a.W, a.K2out, a.NJ, wave, lane);
    __syncthreads();
  }
}

// ---------------------------------------------------------------------------
// Split-precision variant of the block kernel (CIN == COUT == C): the 1x1 skip convolution runs
// as six bf16 MFMAs per 16 channels on the matrix cores (fno_dev.h, fp32-grade accuracy), the
// spectral K-extension and the row-DFT epilogue stay fp32.  The activation tile is staged
// pixel-major as three bf16 arrays; the fp32 output tile for the DFT epilogue reuses that LDS.
static inline size_t pw_fwd_x3_lds_bytes(int c, int npx, int W, int K2in, int NJ, bool has_z, bool has_x1) {
  size_t bytes = (size_t)3 * npx * (c + 8) * 2;
  const size_t out_tile = (size_t)c * (npx + 4) * 4;
  if (out_tile > bytes) bytes = out_tile;
  size_t fl = 0;
  if (has_z) fl += (size_t)2 * K2in * W + (size_t)(npx / W) * K2in * c * 2;
  if (has_x1) fl += (size_t)16 * NJ * (W + 4);
  return bytes + fl * 4;
}

#ifndef FNO_TRACE_SEL
#define FNO_TRACE_SEL false
#endif
// NTW = 32-pixel column tiles per wave.  NTW = 2 halves the workgroup (4 waves at C = 64, NPX = 128)
// so that TWO workgroups share a CU at the same 256-VGPR budget per wave: their phases (split /
// MFMA / epilogue / row DFT) drift apart and the matrix pipe of one overlaps the VALU work of the other.
// RELU: the stored tensor is max(u, 0) (the one-layer stacks of the RNO regressor, rno.py:92-106, whose backward reads the
// ReLU mask off this output)
template <int C, int NPX, int NTW, bool LOOSE = false, bool LIFT = false, bool RELU = false>
__global__ void __launch_bounds__((C / 32) * (NPX / 32 / NTW) * 64, FNO_OCC_PWX) k_pw_fwd_x3(PwFwdArgs a) {
  constexpr int NTN = NPX / 32;
  constexpr int NTG = NTN / NTW;          // wave groups along the pixel dimension
  constexpr int MT = C / 32;
  constexpr int NW = MT * NTG;
  constexpr int NT = NW * 64;
  constexpr int KB = C / 16;
  constexpr int PITCH = NPX + 4;
  using PF = SplitTilePrefetch<NPX, NT, C>;
  constexpr size_t XB_BYTES = (size_t)3 * PF::TERM * 2;
  constexpr size_t OUT_BYTES = (size_t)C * PITCH * 4;
  constexpr size_t REGION = XB_BYTES > OUT_BYTES ? XB_BYTES : OUT_BYTES;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  unsigned short* xb = reinterpret_cast<unsigned short*>(smem);     // 3 x NPX x (C+8) halfs ...
  float* xs = smem;                                                  // ... reused as the C x PITCH fp32 output tile
  float* tinv_s = smem + REGION / 4;
  const int R = LOOSE ? NPX / a.W + 2 : NPX / a.W;
  float* zs = tinv_s + (a.z ? 2 * a.K2in * a.W : 0);
  float* tfwd_s = zs + (a.z ? R * a.K2in * C * 2 : 0);

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, half = lane >> 5;
  const int mt = wave / NTG, ng = wave % NTG;

  if (a.z)
    for (int i = tid; i < 2 * a.K2in * a.W; i += NT) tinv_s[i] = a.tinv[i];
  if (a.x1)
    for (int i = tid; i < 16 * a.NJ * a.W; i += NT) tfwd_s[(i / a.W) * (a.W + 4) + i % a.W] = a.tfwd[i];

  // weight fragments A[i = o][k = c] split into (h, m, l), constant over all tiles
  bf16x8 afrag[KB][3];
#pragma unroll
  for (int kb = 0; kb < KB; ++kb) {
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = a.w[(mt * 32 + l31) * C + kb * 16 + 8 * half + j];
    split3x8(v, afrag[kb][0], afrag[kb][1], afrag[kb][2]);
  }

  // float4s of spectral rows a tile needs: R rows when rows tile it, else the rows that overlap [px0, px0 + NPX)
  auto zc4 = [&](int px0) {
    const int nrows = LOOSE ? (px0 + NPX - 1) / a.W - px0 / a.W + 1 : R;
    return a.z ? nrows * a.K2in * C / 2 : 0;
  };
  PF pf;
  LiftSplitTilePrefetch<NPX, NT, C> pfl;
  __shared__ __attribute__((aligned(16))) float lws[LIFT ? 5 * C : 4];
  if constexpr (LIFT) stage_lift_params<C>(lws, a.lw, a.lb, a.CL, tid, NT);     // visible after the first tile's barrier? no: sync here
  if constexpr (LIFT) __syncthreads();
  float4 zpf = make_float4(0.f, 0.f, 0.f, 0.f);
  auto issue = [&](int tile_) {
    const int tile = a.rev ? a.ntiles - 1 - tile_ : tile_;      // (zigzag along the kernel chain: k_blk_fwd_s)
    const int b = tile / a.tiles_per_plane;
    const int px0 = (tile % a.tiles_per_plane) * NPX;
    // the thread index goes through an opaque move: hoisted out of the tile loop, the per-lane 64-bit base addresses of the
    // two prefetches were the four registers that did not fit (20-24 bytes of scratch per lane at 64 channels)
    int t = tid;
    asm volatile("" : "+v"(t));
    if constexpr (LIFT) pfl.issue(a.x + (size_t)b * a.CL * a.PW + px0, a.PW, a.CL, t);
    else pf.issue(a.x + (size_t)b * C * a.PW + px0, a.PW, t);
    if (t < zc4(px0)) zpf = ld4(a.z + ((size_t)b * a.P + px0 / a.W) * a.K2in * C * 2 + 4 * t);
  };
  if ((int)blockIdx.x < a.ntiles) issue(blockIdx.x);

  int tslot = 0;
  FNO_TRACE_IF(FNO_TRACE_SEL);
  for (int tile_ = blockIdx.x; tile_ < a.ntiles; tile_ += gridDim.x) {
    const int tile = a.rev ? a.ntiles - 1 - tile_ : tile_;
    const int b = tile / a.tiles_per_plane;
    const int px0 = (tile % a.tiles_per_plane) * NPX;
    FNO_STAMP(tslot + 0);
    if constexpr (LIFT) pfl.commit(xb, lws, tid);
    else pf.commit(xb, a.act_in != 0, tid);
    const int zcount4 = zc4(px0);
    if (tid < zcount4) st4(zs + 4 * tid, zpf);
    for (int i = tid + NT; i < zcount4; i += NT)   // rare tail (small workgroups): straight from L2
      st4(zs + 4 * i, ld4(a.z + ((size_t)b * a.P + px0 / a.W) * a.K2in * C * 2 + 4 * i));
    FNO_STAMP(tslot + 1);
    __syncthreads();
    FNO_STAMP(tslot + 2);
    if (tile_ + (int)gridDim.x < a.ntiles) issue(tile_ + gridDim.x);

    f32x16 acc[NTW];
#pragma unroll
    for (int q = 0; q < NTW; ++q) {
      const int n0 = (ng * NTW + q) * 32;
      f32x16 lo;          // cross terms of the split: own accumulator (fno_dev.h: mfma_x3s)
#pragma unroll
      for (int r = 0; r < 16; ++r) { acc[q][r] = 0.0f; lo[r] = 0.0f; }
#pragma unroll
      for (int kb = 0; kb < KB; ++kb) {
        bf16x8 bf[3];
#pragma unroll
        for (int t = 0; t < 3; ++t) bf[t] = ld8h(xb + t * PF::TERM + (n0 + l31) * PF::PBH + kb * 16 + 8 * half);
        mfma_x3s(afrag[kb], bf, acc[q], lo);
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[q][r] += lo[r];
      if constexpr (LOOSE) {
        if (a.z) acc[q] = kext_loose_rows<C>(acc[q], zs, tinv_s, a.K2in, a.W, px0 + n0, px0 / a.W, mt, l31, half);
      } else if (a.z) {
        const int rr = n0 / a.W;
        const float* zr = zs + ((rr * a.K2in) * C + mt * 32 + l31) * 2 + half;
        const float* tv = tinv_s + half * a.W + n0 % a.W + l31;
#pragma unroll 2
        for (int s = 0; s < a.K2in; ++s) acc[q] = mfma32(zr[s * C * 2], tv[2 * s * a.W], acc[q]);
      }
    }
    FNO_STAMP(tslot + 3);
    __syncthreads();  // all waves are done reading the staged input (the output tile reuses it)
    FNO_STAMP(tslot + 4);

    {
      float vmax = 0.f;          // max |u| this thread stores for the tile (a.umax; published per tile: a value that lived
                                 // across the tile loop cost 12 bytes of scratch per lane at the 256-register limit)
      const float* bp = a.bias ? a.bias + mt * 32 + 4 * half : nullptr;
      float bv[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) bv[r] = bp ? bp[(r & 3) + 8 * (r >> 2)] : 0.f;
#pragma unroll
      for (int q = 0; q < NTW; ++q) {
        const int n0 = (ng * NTW + q) * 32;
        float* up = a.u ? a.u + ((size_t)b * C + mt * 32 + 4 * half) * a.PW + px0 + n0 + l31 : nullptr;
        float* xp = xs + (mt * 32 + 4 * half) * PITCH + n0 + l31;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[q][r] += bv[r];
        if (a.add) {
          const float* ap = a.add + ((size_t)b * C + mt * 32 + 4 * half) * a.PW + px0 + n0 + l31;
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[q][r] += ap[(size_t)((r & 3) + 8 * (r >> 2)) * a.PW];
        }
        if constexpr (RELU) {
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[q][r] = acc[q][r] < 0.f ? 0.f : acc[q][r];      // (NaN stays NaN, as torch's relu)
        }
        if (up) {
#pragma unroll
          for (int r = 0; r < 16; ++r) up[(size_t)((r & 3) + 8 * (r >> 2)) * a.PW] = acc[q][r];
        }
        if (a.umax) {
#pragma unroll
          for (int r = 0; r < 16; ++r) vmax = fmaxf(vmax, fabsf(acc[q][r]));
        }
        if (a.x1) {
          if (a.act_out) {      // (on pairs, fno_dev.h: the same values in half the instructions)
            float six, inf;
            gelu_consts(six, inf);
#pragma unroll
            for (int h8 = 0; h8 < 2; ++h8) {
              float t[8];
#pragma unroll
              for (int j = 0; j < 8; ++j) t[j] = acc[q][8 * h8 + j];
              gelu8(t, six, inf);
#pragma unroll
              for (int j = 0; j < 8; ++j) xp[((j & 3) + 8 * (2 * h8 + (j >> 2))) * PITCH] = t[j];
            }
          } else {
#pragma unroll
            for (int r = 0; r < 16; ++r) xp[((r & 3) + 8 * (r >> 2)) * PITCH] = acc[q][r];
          }
        }
      }
      if (a.umax) absmax_publish(vmax, a.umax);
    }
    FNO_STAMP(tslot + 5);
    if (a.x1) {
      __syncthreads();
      FNO_STAMP(tslot + 6);
      row_dft_epilogue<C, NPX, NW>(xs, tfwd_s, a.W + 4, a.x1, b, px0, a.P, a.W, a.K2out, a.NJ, wave, lane);
    }
    FNO_STAMP(tslot + 7);
    __syncthreads();
    tslot += 8;
  }
}

// ---------------------------------------------------------------------------
// Row spectra of the lifting output WITHOUT the lifting output: x1 = rowDFT(W_l x + b_l) is linear in x, so
//   x1[b, row, k2, c] = sum_k W_l[c][k] * rowDFT(x_k)[row, k2] + b_l[c] * sum_w T[k2][w]
// needs the truncated DFT of the <= 4 input channels only (the fused block 0 recomputes u_0 itself and never reads it:
// k_pw_fwd_x3<.., LIFT>).  The tile kernel it replaces transformed all C channels on the fp32 matrix pipe: 52 us at BASELINE
// config 2 for 12.6 MB in and 25 MB out; this one is a streaming pass.
//   block 256, one workgroup per LR_ROWS rows of W pixels; LDS: table 2 K2 x W, LR_ROWS x CL x W inputs, spectra
constexpr int LR_ROWS = 8;
__global__ void __launch_bounds__(256) k_lift_rowdft(const float* __restrict__ x, const float* __restrict__ lw,
                                                     const float* __restrict__ lb, const float* __restrict__ tfwd,
                                                     float2* __restrict__ x1, int CL, int C, int PW, int W, int P, int K2,
                                                     int nrows, float* __restrict__ xmax) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int WP = W + 4;                               // row pitch: threads of a wave read different rows at the same w, four
                                                      // floats per read (rows 4 banks apart: 8 lanes cover the 32 banks)
  float* ts = smem;                                   // [2 K2][WP]
  float* xs = ts + 2 * K2 * WP;                       // [LR_ROWS][CL][WP]
  float2* xh = reinterpret_cast<float2*>(xs + ((LR_ROWS * CL * WP + 1) & ~1));     // [LR_ROWS][K2][CL + 1]: spectra; entry CL = sum of the table row
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int row0 = blockIdx.x * LR_ROWS;              // rows are (b, prow) pairs, b-major
  const int nr = min(LR_ROWS, nrows - row0);
  float vmax = 0.f;
  const int c = lane;                                  // the output stage's lane <-> channel weights: requested up front
  float wv[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) wv[k] = (c < C && k < CL) ? lw[c * CL + k] : 0.f;
  const float bc = (lb && c < C) ? lb[c] : 0.f;
  // Eight loads per thread in flight before the first LDS write (round 4: the row-per-wave loops issued one load, waited,
  // wrote, ... - 12 + 7 dependent round trips per workgroup, 30 us per launch for 38 MB at BASELINE config 2)
  {
    const int nt_tab = 2 * K2 * W, nt_x = LR_ROWS * CL * W;
    for (int base = 0; base < nt_tab; base += 8 * 256) {
      float v[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) { const int i = base + q * 256 + tid; v[q] = i < nt_tab ? tfwd[i] : 0.f; }
#pragma unroll
      for (int q = 0; q < 8; ++q) { const int i = base + q * 256 + tid; if (i < nt_tab) ts[(i / W) * WP + i % W] = v[q]; }
    }
    for (int base = 0; base < nt_x; base += 8 * 256) {
      float v[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const int i = base + q * 256 + tid;
        const int rk = i / W, w = i - rk * W, r = rk / CL, k = rk - r * CL;
        const int row = row0 + r;
        const int b = row / P, prow = row - b * P;
        v[q] = (i < nt_x && r < nr) ? x[((size_t)b * CL + k) * PW + (size_t)prow * W + w] : 0.f;
      }
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const int i = base + q * 256 + tid;
        if (i < nt_x) { xs[(i / W) * WP + i % W] = v[q]; vmax = fmaxf(vmax, fabsf(v[q])); }
      }
    }
  }
  if (xmax) {      // max |x| of the model input (bound for the fused block 0's fp16 operand scale): one publish per workgroup
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) vmax = fmaxf(vmax, __shfl_xor(vmax, o, 64));
    __shared__ float wmax[4];
    if (lane == 0) wmax[wave] = vmax;
    __syncthreads();
    // All workgroups of the launch reach this point together, see the slot still at zero and would ALL issue the atomic:
    // 1024 same-address atomics serialise to ~24 us (found in round 4: the kernel took 30 us with or without its DFT and
    // its stores).  One workgroup in sixteen publishes here; the others look again at the end of the kernel, when the
    // slot already holds (nearly) the maximum and almost none of them has anything to add.  Same result: a maximum.
    vmax = fmaxf(fmaxf(wmax[0], wmax[1]), fmaxf(wmax[2], wmax[3]));      // the workgroup's maximum, in every thread
    if (wave == 0 && (blockIdx.x & 15) == 0) absmax_publish(vmax, xmax);
  }
  __syncthreads();
  for (int i = tid; i < LR_ROWS * K2 * (CL + 1); i += 256) {
    const int r = i / (K2 * (CL + 1)), k2 = (i / (CL + 1)) % K2, k = i % (CL + 1);
    const float* tr = ts + (2 * k2) * WP;
    const float* ti = tr + WP;
    float sr = 0.f, si = 0.f;
    if (k < CL) {
      const float* xr = xs + (r * CL + k) * WP;
      float sr2 = 0.f, si2 = 0.f;                       // two chains: the sums are latency-bound otherwise
      // 16-byte LDS reads (round 5: the stage was bound by LDS issue - three 4-byte reads per product pair); W is a multiple
      // of 32 on this path.  Same products in the same two chains (even / odd w) as before.
#pragma unroll 4
      for (int w = 0; w < W; w += 4) {
        const float4 xv = *reinterpret_cast<const float4*>(xr + w), tv = *reinterpret_cast<const float4*>(tr + w),
                     uv = *reinterpret_cast<const float4*>(ti + w);
        sr = fmaf(xv.x, tv.x, sr); si = fmaf(xv.x, uv.x, si);
        sr2 = fmaf(xv.y, tv.y, sr2); si2 = fmaf(xv.y, uv.y, si2);
        sr = fmaf(xv.z, tv.z, sr); si = fmaf(xv.z, uv.z, si);
        sr2 = fmaf(xv.w, tv.w, sr2); si2 = fmaf(xv.w, uv.w, si2);
      }
      sr += sr2; si += si2;
    } else {
      for (int w = 0; w < W; ++w) { sr += tr[w]; si += ti[w]; }
    }
    xh[i] = make_float2(sr, si);
  }
  __syncthreads();
  // lane <-> channel (two channels per lane pair of passes when C > 64 never happens: C is 32 or 64); a wave per (row, bin)
  if (c < C) {
    float2* dst = x1 + (size_t)row0 * K2 * C + c;
    for (int p = wave; p < nr * K2; p += 4) {           // p = r * K2 + k2
      const float2* h = xh + p * (CL + 1);
      float sr = bc * h[CL].x, si = bc * h[CL].y;
#pragma unroll
      for (int k = 0; k < 4; ++k)
        if (k < CL) { sr = fmaf(wv[k], h[k].x, sr); si = fmaf(wv[k], h[k].y, si); }
      dst[(size_t)p * C] = make_float2(sr, si);
    }
  }
  if (xmax && wave == 0 && (blockIdx.x & 15) != 0) absmax_publish(vmax, xmax);      // (see above: the late look)
}
