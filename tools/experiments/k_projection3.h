// Backward of the fused projection MLP, third generation ("q": FOUR waves per SIMD; round 5).  Same arguments, mathematics
// and partial-slab outputs as k_proj_bwd_t (k_projection2.h; reference: autograd of neuralop/models/tfno.py:23-38), two fp16
// terms per operand (fno_dev.h "h2").  Why: tools/occupancy_valu_test.hip measured that a SIMD issues vector instructions at
// 0.13-0.15 per cycle from two waves beside matrix work and at 0.23-0.27 from four; k_proj_bwd_t is pinned to two waves per
// SIMD by the 256 x 64 dW1 accumulators of its one 8-wave workgroup (245 VGPRs) and spends 1.08e8 vector instructions per
// launch at 0.09 per cycle and SIMD (profiles/r04_pmc_sq.csv).  Here:
//   * ONE 16-wave workgroup per CU (<= 128 VGPRs), 64-pixel half tiles (two per 128-pixel tile of the plan);
//   * wave (hb, pb) OWNS hidden rows 32 hb .. 32 hb + 31 for the whole kernel and pixel block pb of every half tile: b1 / w2 /
//     the db1 / dW2 sums are per-lane scalars, its dW1 slice (64 x 32: two accumulator tiles) never leaves its registers;
//   * the hidden chunk is computed TRANSPOSED (P1^T[pixel][hidden]: lane <-> hidden row) so that dW1^T[c][hid] = a[c][px] X[px][hid]
//     takes the accumulator registers straight as the B operand (cdna_hip_programming.md "an accumulator tile as the next
//     MFMA's operand": k order 16 s + 8 (j >> 2) + 4 half + (j & 3)) - no LDS round trip, no barrier for dW1;
//   * dx^T[px][c] sums over ALL hidden rows: the dP1 values go to a [hidden][pixel] fp16 image once (8-byte stores), and after
//     ONE barrier eight waves contract it (transposed reads) against a [hidden][channel] image of W1 that lives in LDS for the
//     whole kernel (row reads feed the recompute, transposed reads the dx product: one image, two uses); the K range is split
//     in two, the partner waves exchange half of their partial tile through LDS and each finishes 8 of the 16 pixel groups;
//   * the row DFT of the gradient runs per half tile on the waves that have no dx work, a 128-float row's two halves are
//     added in registers (same wave, same lanes).
//   per half tile:  commit a = act(u) | B1 | recompute (12 MFMAs), GELU / GELU' / dP1 split, dW1 (12) | B2 | dx (24, waves 0-7)
//                   | B4 | partner add, x act'(u), gout store | B5 | row DFT (waves 8-15)
// LDS: a image 16 KB (reused for the partial exchange) | dP1 image 64 KB (reused for the fp32 gout half tile) | W1 image 64 KB
//      | dy | forward row table.
#pragma once
#include "fno_dev.h"
#include "k_block_bwd2.h"
#include "k_projection.h"

// [rows][64 x 16 bit] image with 128-byte rows addressed in 8-byte groups g = column / 4 (0..15): byte offset of group g of
// `row`.  Conflict-free for ds_read_b64_tr_b16 blocks (4 rows x 8 groups per 32-lane half, rows 4-aligned, groups 8-aligned)
// and for ds_write_b64 of one row's 16 groups; ds_read_b64 by 32 consecutive rows at one g is 2-way (16 such reads per wave and
// half tile).  The XOR only looks at row bits 1-3: rows 16 apart differ by a CONSTANT 2048 bytes, so the k blocks of a product
// share one per-lane base and take their offsets as immediates (with row bit 4 in the XOR the compiler kept one address
// register per k block and spilled them: 540 bytes of scratch).
FNO_DEV int off8_x(int row) { return (((row >> 1) & 1) << 3) | ((row >> 2) & 3); }
FNO_DEV int off8(int row, int g) { return 128 * row + 8 * (g ^ off8_x(row)); }
// the compiler must not hoist what depends on v out of the enclosing loop (it would trade a v_xor for a spilled register)
FNO_DEV int opaque(int v) { asm volatile("" : "+v"(v)); return v; }

FNO_DEV void split2x4(const float (&v)[4], float s, unsigned& h0, unsigned& h1, unsigned& l0, unsigned& l1) {
  const f32x2 v0 = natural_pair(v[0], v[1]) * f32x2{s, s}, v1 = natural_pair(v[2], v[3]) * f32x2{s, s};
  const f16x2 a0 = __builtin_convertvector(v0, f16x2), a1 = __builtin_convertvector(v1, f16x2);
  const f16x2 b0 = __builtin_convertvector(v0 - __builtin_convertvector(a0, f32x2), f16x2);
  const f16x2 b1 = __builtin_convertvector(v1 - __builtin_convertvector(a1, f32x2), f16x2);
  h0 = __builtin_bit_cast(unsigned, a0); h1 = __builtin_bit_cast(unsigned, a1);
  l0 = __builtin_bit_cast(unsigned, b0); l1 = __builtin_bit_cast(unsigned, b1);
}
typedef unsigned u32x4v __attribute__((ext_vector_type(4)));
FNO_DEV f16x8 frag_from(unsigned a, unsigned b, unsigned c, unsigned d) { return __builtin_bit_cast(f16x8, u32x4v{a, b, c, d}); }
FNO_DEV f16x8 lds_rd2x8(const unsigned char* p0, const unsigned char* p1) {      // two 8-byte reads -> one 8 x fp16 fragment
  const uint2 u = *reinterpret_cast<const uint2*>(p0), v = *reinterpret_cast<const uint2*>(p1);
  return frag_from(u.x, u.y, v.x, v.y);
}
FNO_DEV f16x8 lds_tr2x8(const unsigned char* p0, const unsigned char* p1) {      // two transposed reads (rows r .. r+3, r+4 .. r+7)
  return __builtin_bit_cast(f16x8, cat4(lds_tr16(p0), lds_tr16(p1)));
}

// -DPBQ_TRACE (tools/pbq_bench.hip): shader-clock stamps of workgroup 0, every wave, the first 32 half tiles
#ifndef PBQ_SKIP
#define PBQ_SKIP 0      // timing experiments (results wrong): 1 no GELU, 2 no dW1 products, 4 no dP1 image stores, 8 no dx products,
#endif                  // 16 no row DFT, 32 no recompute products, 64 no next-tile prefetch, 128 no dP1 split
#ifndef PBQ_VAR
#define PBQ_VAR 0       // structure experiments (results right): 1 next-tile loads issued behind the epilogue, 2 dx loop unrolled by two
#endif                  // without scheduling fences, 4 recompute loop without scheduling fences
#ifdef PBQ_TRACE
__device__ unsigned long long g_pbq[16 * 32 * 16];
#define PBQ_STAMP(slot) do { if (blockIdx.x == 0 && pbq_ht < 32 && (threadIdx.x & 63) == 0) \
    g_pbq[((threadIdx.x >> 6) * 32 + pbq_ht) * 16 + (slot)] = __builtin_readcyclecounter(); } while (0)
#else
#define PBQ_STAMP(slot) do { } while (0)
#endif
#if PBQ_SKIP & 512
#define PBQ_SYNC() do { } while (0)
#else
#define PBQ_SYNC() __syncthreads()
#endif
template <int HID, bool RELU = false>
__global__ void __launch_bounds__(1024) k_proj_bwd_q(ProjBwdArgs a) {
  constexpr int C = 64, NPX = 64, NT = 1024, PITCH = NPX + 4;
  constexpr int ATERM = C * 128, DTERM = HID * 128, WTERM = HID * 128;      // bytes per term plane
  static_assert(HID == 256, "16 waves = 8 hidden 32-blocks x 2 pixel blocks");
  static_assert((size_t)C * PITCH * 4 <= (size_t)2 * DTERM && (size_t)4 * 8 * 64 * 4 * 2 <= (size_t)2 * ATERM, "aliases fit");
  extern __shared__ __attribute__((aligned(16))) float smem[];
  unsigned char* aimg = reinterpret_cast<unsigned char*>(smem);      // [2][64 c][64 px] fp16, off8
  unsigned char* dimg = aimg + 2 * ATERM;                            // [2][256 hid][64 px] fp16, off8
  unsigned char* wimg = dimg + 2 * DTERM;                            // [2][256 hid][64 c] fp16, swz64_off
  float* douts = reinterpret_cast<float*>(wimg + 2 * WTERM);         // dy of the half tile (64)
  float* tfwd_s = douts + NPX;                                       // 16 NJ x (W + 4): forward row table (if x1g)
  // behind the table: the row-DFT sums of the first half of a 128-float row, [8 jobs][64 lanes] float4 (same wave, same lanes
  // write and read them: no barrier)
  float* part = reinterpret_cast<float*>(aimg);                      // dx partial exchange [2 kh][4 tiles][8 regs][64 lanes]
  float* r3 = reinterpret_cast<float*>(dimg);                        // gout half tile [64 c][PITCH]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, half = lane >> 5, l15 = lane & 15, quad = lane >> 4;
  const int hb = wave >> 1, pb = wave & 1, n0 = pb * 32;
  const int tq = l15 >> 2, tp = l15 & 3;
  const int trow = 8 * (quad >> 1) + tq;            // transposed-read roles: row of the 8-row block, 4-column group
  const int tcol = 16 * (quad & 1) + 4 * tp;

  const float sa = h2_scale(*a.xmax), sw = h2_scale(a.amax[2]), sd = h2_scale(1.13f * a.amax[3] * a.amax[1]);
  const float inv_aw = 1.f / (sa * sw), inv_dw = 1.f / (sd * sw), inv_da = 1.f / (sd * sa);
  float gk_six, gk_inf;
  gelu_consts(gk_six, gk_inf);

  // ---- prologue: W1 -> two fp16 term planes [hid][c] (16-byte chunks, swz64_off); row table; per-lane constants ----
#pragma unroll 1
  for (int it = 0; it < HID * 8 / NT; ++it) {
    const int item = tid + it * NT, row = item >> 3, ch = item & 7;
    const float4 w0 = ld4(a.w1 + (size_t)row * C + 8 * ch), w1v = ld4(a.w1 + (size_t)row * C + 8 * ch + 4);
    const float v[8] = {w0.x, w0.y, w0.z, w0.w, w1v.x, w1v.y, w1v.z, w1v.w};
    f16x8 h, l;
    split2x8(v, sw, h, l);
    *reinterpret_cast<f16x8*>(wimg + swz64_off(row, ch)) = h;
    *reinterpret_cast<f16x8*>(wimg + WTERM + swz64_off(row, ch)) = l;
  }
  if (a.x1g)
    for (int i = tid; i < 16 * a.NJ * a.W; i += NT) tfwd_s[(i / a.W) * (a.W + 4) + i % a.W] = a.tfwd[i];
  const int hrow = hb * 32 + l31;
  const float b1v = a.b1[hrow], w2v = a.w2[hrow];
  float sdb1 = 0.f, sdw2 = 0.f, gvmax = 0.f;
  f32x16 dw1[2];
#pragma unroll
  for (int k = 0; k < 2; ++k)
#pragma unroll
    for (int r = 0; r < 16; ++r) dw1[k][r] = 0.f;

  // the half tile's rows of u_L: thread (c = tid / 16, g = tid % 16) loads 16 bytes
  float4 xq;
  auto issue_x = [&](int ht, int c, int g) {
    const int tile = ht >> 1;
    const int b = tile / a.tiles_per_plane;
    const int px0 = (tile % a.tiles_per_plane) * 128 + (ht & 1) * NPX;
    xq = ld4(a.x + ((size_t)b * C + c) * a.PW + px0 + 4 * g);
  };
  const int nht = 2 * a.ntiles;
  if (2 * (int)blockIdx.x < nht) issue_x(2 * blockIdx.x, tid >> 4, tid & 15);

#ifdef PBQ_TRACE
  int pbq_ht = -1;
#endif
  for (int tile = blockIdx.x; tile < a.ntiles; tile += gridDim.x) {
    const int b = tile / a.tiles_per_plane;
#pragma unroll 1
    for (int hf = 0; hf < 2; ++hf) {
#ifdef PBQ_TRACE
      ++pbq_ht;
#endif
      PBQ_STAMP(0);
      const int px0 = (tile % a.tiles_per_plane) * 128 + hf * NPX;
      // every lane-derived index of the loop body comes from an OPAQUE copy of the lane id: the compiler otherwise hoists
      // dozens of per-lane address terms out of the tile loop and spills them (128-register budget)
      const int ln = opaque(lane);
      const int l31 = ln & 31, half = ln >> 5, l15 = ln & 15, quad = ln >> 4;
      const int tq = l15 >> 2, tp = l15 & 3;
      const int trow = 8 * (quad >> 1) + tq, tcol = 16 * (quad & 1) + 4 * tp;
      const int hrow = hb * 32 + l31;
      const int aT0 = off8(trow, (n0 + tcol) >> 2), aT1 = off8(trow + 4, (n0 + tcol) >> 2);      // a image, transposed reads
      const int wfx = (((hrow >> 1) & 1) << 2) | ((hrow >> 2) & 3);                                 // swz64_off's XOR of row hrow
      const int wRb = 128 * hrow + 16 * (half ^ (wfx & 1)), wfx6 = wfx & 6;                         // W1 image, row reads
      const int dfx = off8_x(hrow), afx = off8_x(l31);
      const int xc = (wave << 2) | (ln >> 4), xg = ln & 15;
      // ---- commit: a = act(u) -> [c][px] image, one split --------------------------------------------------------------
      {
        float4 t = xq;
        if (a.act_in) t = gelu4(t, gk_six, gk_inf);
        const float tv[4] = {t.x, t.y, t.z, t.w};
        unsigned h0, h1, l0, l1;
        split2x4(tv, sa, h0, h1, l0, l1);
        *reinterpret_cast<uint2*>(aimg + off8(xc, xg)) = make_uint2(h0, h1);
        *reinterpret_cast<uint2*>(aimg + ATERM + off8(xc, xg)) = make_uint2(l0, l1);
      }
      if (tid < NPX) douts[tid] = a.dy[(size_t)b * a.PW + px0 + tid];
      PBQ_STAMP(1);
      PBQ_SYNC();                                                                                                  // B1
      PBQ_STAMP(2);
      // ---- A1: P1^T[px][hid] = sum_c a[c][px] W1[hid][c] (+ b1): A = transposed reads of the a image, B = row reads of W1 ----
      f32x16 hi, lo;
#pragma unroll
      for (int r = 0; r < 16; ++r) { hi[r] = 0.f; lo[r] = 0.f; }
#pragma unroll
      for (int kb = 0; kb < ((PBQ_SKIP & 32) ? 0 : 4); ++kb) {
        f16x8 af[2], bf[2];
        const int o0 = aT0 + 2048 * kb, o1 = aT1 + 2048 * kb;         // rows 16 kb + trow (+ 4) of the a image
        const int ow = wRb + 16 * ((2 * kb) ^ wfx6);                   // chunk (2 kb + half) ^ swizzle of W1 row hrow
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          af[t] = lds_tr2x8(aimg + t * ATERM + o0, aimg + t * ATERM + o1);
          bf[t] = *reinterpret_cast<const f16x8*>(wimg + t * WTERM + ow);
        }
        mfma_h2s(af, bf, hi, lo);
        if (!(PBQ_VAR & 4)) __builtin_amdgcn_sched_barrier(0);      // at most one k block of operand fragments live (the budget is 128 registers)
      }
      // ---- E: lane <-> hidden row hrow; registers <-> pixels n0 + (r & 3) + 8 (r >> 2) + 4 half ----------------------------
      // ---- E + B per k step s of the dW1 product (16 pixels): dP1 = act'(P1) w2 dy as packed fp16 pairs - what goes to the image
      //      AND the B fragments of dW1^T[c][hid] += sum_px a[c][px] dP1[px][hid] (the accumulator's own registers, k order of the idiom)
      PBQ_STAMP(3);
      float p1[16];               // hh + cross terms: 16 instead of 32 live registers through the vector phase
#pragma unroll
      for (int r = 0; r < 16; ++r) p1[r] = (PBQ_SKIP & 256) ? (float)(ln * 16 + r) * 1e-3f * (float)(hf + 1) : hi[r] + lo[r];
      {
        float sdb = 0.f, sdw = 0.f;
        const int dWb = 128 * hrow, aRb = 128 * l31;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
          unsigned dh[4], dl[4];
#pragma unroll
          for (int ii = 0; ii < 2; ++ii) {
            const int i = 2 * s + ii;
            const float4 dy4 = ld4(douts + n0 + 8 * i + 4 * half);
            const float dyv[4] = {dy4.x, dy4.y, dy4.z, dy4.w};
            float4 glv = make_float4(fmaf(p1[4 * i], inv_aw, b1v), fmaf(p1[4 * i + 1], inv_aw, b1v),
                                     fmaf(p1[4 * i + 2], inv_aw, b1v), fmaf(p1[4 * i + 3], inv_aw, b1v)), dgv;
            if constexpr (RELU) {
              dgv = make_float4(glv.x > 0.f ? 1.f : 0.f, glv.y > 0.f ? 1.f : 0.f, glv.z > 0.f ? 1.f : 0.f, glv.w > 0.f ? 1.f : 0.f);
              glv = make_float4(fmaxf(glv.x, 0.f), fmaxf(glv.y, 0.f), fmaxf(glv.z, 0.f), fmaxf(glv.w, 0.f));
            } else if (PBQ_SKIP & 1) dgv = glv;
            else gelu_both4(glv, dgv);
            const float gl4[4] = {glv.x, glv.y, glv.z, glv.w}, dg4[4] = {dgv.x, dgv.y, dgv.z, dgv.w};
            float dp[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              dp[j] = dg4[j] * (w2v * dyv[j]);
              sdw = fmaf(gl4[j], dyv[j], sdw);
              sdb += dp[j];
            }
            if (PBQ_SKIP & 128) {
              dh[2 * ii] = __builtin_bit_cast(unsigned, dp[0]); dh[2 * ii + 1] = __builtin_bit_cast(unsigned, dp[1]);
              dl[2 * ii] = __builtin_bit_cast(unsigned, dp[2]); dl[2 * ii + 1] = __builtin_bit_cast(unsigned, dp[3]);
            } else
            split2x4(dp, sd, dh[2 * ii], dh[2 * ii + 1], dl[2 * ii], dl[2 * ii + 1]);
            const int od = dWb + 8 * (((n0 >> 2) + 2 * i + half) ^ dfx);
            if (!(PBQ_SKIP & 4)) {
            *reinterpret_cast<uint2*>(dimg + od) = make_uint2(dh[2 * ii], dh[2 * ii + 1]);
            *reinterpret_cast<uint2*>(dimg + DTERM + od) = make_uint2(dl[2 * ii], dl[2 * ii + 1]);
            }
            asm volatile("" : "+v"(sdb), "+v"(sdw));      // the two sums are finished HERE (the optimiser otherwise sinks both
                                                           // chains to the end of the tile loop and parks 24 operands in scratch)
          }
          PBQ_STAMP(11 + 2 * s);
          const f16x8 bf[2] = {frag_from(dh[0], dh[1], dh[2], dh[3]), frag_from(dl[0], dl[1], dl[2], dl[3])};
#pragma unroll
          for (int cb = 0; cb < ((PBQ_SKIP & 2) ? 0 : 2); ++cb) {
            f16x8 af[2];
            const int g0 = (n0 >> 2) + 4 * s + half;
            const int oa0 = aRb + 4096 * cb + 8 * (g0 ^ afx), oa1 = aRb + 4096 * cb + 8 * ((g0 + 2) ^ afx);      // row 32 cb + l31
#pragma unroll
            for (int t = 0; t < 2; ++t) af[t] = lds_rd2x8(aimg + t * ATERM + oa0, aimg + t * ATERM + oa1);
            dw1[cb] = mfma_h2(af, bf, dw1[cb]);
          }
          __builtin_amdgcn_sched_barrier(0);
          PBQ_STAMP(12 + 2 * s);
        }
        sdb1 += sdb; sdw2 += sdw;
      }
      // the next half tile's rows: in flight during the dx phase
      {
        const int nh = hf == 0 ? 2 * tile + 1 : 2 * (tile + (int)gridDim.x);
        if (nh < nht && !(PBQ_SKIP & 64) && !(PBQ_VAR & 1)) issue_x(nh, xc, xg);
      }
      PBQ_STAMP(4);
      PBQ_SYNC();                                                                                                  // B2
      PBQ_STAMP(5);
      // ---- A3 (waves 0-7): dx^T[px][c] = sum_hid dP1[hid][px] W1[hid][c]; wave = (K half kh, pixel block pt, channel block ct) ----
      const int kh = (wave >> 2) & 1, pt = (wave >> 1) & 1, ct = wave & 1;
      if (wave < 8) {
        f32x16 dxh, dxl;
#pragma unroll
        for (int r = 0; r < 16; ++r) { dxh[r] = 0.f; dxl[r] = 0.f; }
        const int crow = ct * 32 + l31;
        const int dT0 = off8(kh * 128 + trow, (pt * 32 + tcol) >> 2), dT1 = off8(kh * 128 + trow + 4, (pt * 32 + tcol) >> 2);
        const int wT0 = swz64_off(kh * 128 + trow, (ct * 32 + tcol) >> 3) + 2 * (tcol & 7);
        const int wT1 = swz64_off(kh * 128 + trow + 4, (ct * 32 + tcol) >> 3) + 2 * (tcol & 7);
#if PBQ_VAR & 2
#pragma unroll 2
#else
#pragma unroll 1
#endif
        for (int kb = 0; kb < ((PBQ_SKIP & 8) ? 0 : 8); ++kb) {
          f16x8 af[2], bf[2];
          const int o0 = dT0 + 2048 * kb, o1 = dT1 + 2048 * kb, w0 = wT0 + 2048 * kb, w1o = wT1 + 2048 * kb;      // rows + 16 kb
#pragma unroll
          for (int t = 0; t < 2; ++t) {
            af[t] = lds_tr2x8(dimg + t * DTERM + o0, dimg + t * DTERM + o1);
            bf[t] = lds_tr2x8(wimg + t * WTERM + w0, wimg + t * WTERM + w1o);
          }
          mfma_h2s(af, bf, dxh, dxl);
          if (!(PBQ_VAR & 2)) __builtin_amdgcn_sched_barrier(0);
        }
        // the pair (kh = 0, 1) of a tile: each hands the other the 8 registers (two pixel groups) the other finishes -
        // kh = 0 keeps registers 0-7 (i = 0, 1), kh = 1 registers 8-15 (i = 2, 3); selects on the wave-uniform kh, no branches
        // u of this wave's share of the tile (pixel groups i = 2 kh, 2 kh + 1) for gelu'(u): requested behind the products (L2: the
        // commit read these lines), the barrier below hides the latency
        float4 uq[2];
        if (a.act_in) {
#pragma unroll
          for (int j = 0; j < 2; ++j)
            uq[j] = ld4(a.x + ((size_t)b * C + crow) * a.PW + px0 + pt * 32 + 8 * (2 * kh + j) + 4 * half);
        }
        float own[8], give[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          const float s0 = dxh[q] + dxl[q], s1 = dxh[8 + q] + dxl[8 + q];
          own[q] = kh ? s1 : s0;
          give[q] = kh ? s0 : s1;
        }
        float* pw = part + (((kh * 4 + pt * 2 + ct) * 8) * 64) + lane;
#pragma unroll
        for (int q = 0; q < 8; ++q) pw[q * 64] = give[q];
        PBQ_STAMP(6);
        PBQ_SYNC();                                                                                                // B4
        PBQ_STAMP(7);
        const float* pr = part + ((((1 - kh) * 4 + pt * 2 + ct) * 8) * 64) + lane;
        const size_t ro = ((size_t)b * C + crow) * a.PW + px0 + pt * 32 + 16 * kh + 4 * half;
        float* r3p = r3 + crow * PITCH + pt * 32 + 16 * kh + 4 * half;
#pragma unroll
        for (int ii = 0; ii < 2; ++ii) {
          float4 v = make_float4(own[4 * ii] + pr[(4 * ii) * 64], own[4 * ii + 1] + pr[(4 * ii + 1) * 64],
                                 own[4 * ii + 2] + pr[(4 * ii + 2) * 64], own[4 * ii + 3] + pr[(4 * ii + 3) * 64]);
          v.x *= inv_dw; v.y *= inv_dw; v.z *= inv_dw; v.w *= inv_dw;
          if (a.act_in) {
            float4 uu = uq[ii], dd;
            gelu_both4(uu, dd);
            v.x *= dd.x; v.y *= dd.y; v.z *= dd.z; v.w *= dd.w;
          }
          st4(a.gout + ro + 8 * ii, v);
          if (a.gmax_out) gvmax = fmaxf(fmaxf(gvmax, fabsf(v.x)), fmaxf(fmaxf(fabsf(v.y), fabsf(v.z)), fabsf(v.w)));
          if (a.x1g) st4(r3p + 8 * ii, v);
        }
      } else {
        PBQ_STAMP(6);
        PBQ_SYNC();                                                                                                // B4
        PBQ_STAMP(7);
      }
      if (PBQ_VAR & 1) {
        const int nh = hf == 0 ? 2 * tile + 1 : 2 * (tile + (int)gridDim.x);
        if (nh < nht && !(PBQ_SKIP & 64)) issue_x(nh, xc, xg);
      }
      PBQ_STAMP(8);
      PBQ_SYNC();                  // B5: the exchange buffer (= the a image) and the dP1 image are free; the gout half tile is complete
      PBQ_STAMP(9);
      if (a.x1g && !(PBQ_SKIP & 16)) {
        // ---- row DFT of the gout half tile (waves 8-15): X1[b, row, k, c] = sum_w g[c][w] (tfwd[2k][w] + i tfwd[2k+1][w]) ----
        // job = (16-channel block nt, row segment rr of SEG floats, 16-output block jt); rows of 128 floats span both halves:
        // the first half's sums wait in dcar (same wave, same lanes in both halves)
        const int SEG = a.W < NPX ? a.W : NPX, R = NPX / SEG;
        const int njobs = 4 * R * a.NJ;
        const int job = wave - 8;       // (use_pbwd_q admits at most 8 jobs per half tile)
        if (wave >= 8 && job < njobs) {
          const int nt = job & 3, rr = (job >> 2) % R, jt = (job >> 2) / R;
          const int wofs = a.W > NPX ? hf * NPX : 0;           // position of the segment in its row
          f32x4 d0 = {0.f, 0.f, 0.f, 0.f}, d1 = {0.f, 0.f, 0.f, 0.f};
          const float* tf = tfwd_s + (size_t)(jt * 16 + l15) * (a.W + 4) + wofs + 4 * quad;
          const float* xr = r3 + (nt * 16 + l15) * PITCH + rr * SEG + 4 * quad;
          for (int q0 = 0; q0 < SEG / 16; q0 += 2) {
            float4 av[2], bv[2];
#pragma unroll
            for (int j = 0; j < 2; ++j) { av[j] = ld4(tf + 16 * (q0 + j)); bv[j] = ld4(xr + 16 * (q0 + j)); }
#pragma unroll
            for (int j = 0; j < 2; ++j) {
              d0 = mfma16(av[j].x, bv[j].x, d0);
              d1 = mfma16(av[j].y, bv[j].y, d1);
              d0 = mfma16(av[j].z, bv[j].z, d0);
              d1 = mfma16(av[j].w, bv[j].w, d1);
            }
          }
          f32x4 d = {d0[0] + d1[0], d0[1] + d1[1], d0[2] + d1[2], d0[3] + d1[3]};
          if (a.W > NPX) {
            float* dc = tfwd_s + 16 * a.NJ * (a.W + 4) + (job * 64 + ln) * 4;
            if (hf == 0) st4(dc, make_float4(d[0], d[1], d[2], d[3]));
            else { const float4 c4 = ld4(dc); d[0] += c4.x; d[1] += c4.y; d[2] += c4.z; d[3] += c4.w; }
          }
          if (a.W <= NPX || hf == 1) {
            const int prow = (a.W > NPX ? (px0 - NPX) : px0) / a.W + rr;
            const int c = nt * 16 + l15;
#pragma unroll
            for (int pr2 = 0; pr2 < 2; ++pr2) {
              const int k2 = jt * 8 + quad * 2 + pr2;
              if (k2 < a.K2out)
                *reinterpret_cast<float2*>(a.x1g + ((((size_t)b * a.P + prow) * a.K2out + k2) * C + c) * 2) = make_float2(d[2 * pr2], d[2 * pr2 + 1]);
            }
          }
        }
        PBQ_STAMP(10);
        // (no barrier: the gout half tile sits in the dP1 image, which the next half tile writes only behind its commit barrier)
      }
    }
  }

  // ---- partial slabs: one per workgroup ------------------------------------------------------------------------------------
  if (a.gmax_out) absmax_publish(gvmax, a.gmax_out);
  __syncthreads();
  float* sc = reinterpret_cast<float*>(dimg);      // [8 hb][2 cb][16][64] dW1 of the pb = 1 waves, then [8][2][32] sums
  float* ss = sc + 8 * 2 * 16 * 64;
  {
    const float vb = sdb1 + __shfl_xor(sdb1, 32, 64), vw = sdw2 + __shfl_xor(sdw2, 32, 64);
    if (pb == 1) {
#pragma unroll
      for (int k = 0; k < 2; ++k)
#pragma unroll
        for (int r = 0; r < 16; ++r) sc[((hb * 2 + k) * 16 + r) * 64 + lane] = dw1[k][r];
      if (half == 0) { ss[(hb * 2 + 0) * 32 + l31] = vb; ss[(hb * 2 + 1) * 32 + l31] = vw; }
    }
    __syncthreads();
    if (pb == 0) {
      float* dst = a.dw1_part + (size_t)blockIdx.x * HID * C;
#pragma unroll
      for (int k = 0; k < 2; ++k)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          float4 v;
          v.x = (dw1[k][4 * i] + sc[((hb * 2 + k) * 16 + 4 * i) * 64 + lane]) * inv_da;
          v.y = (dw1[k][4 * i + 1] + sc[((hb * 2 + k) * 16 + 4 * i + 1) * 64 + lane]) * inv_da;
          v.z = (dw1[k][4 * i + 2] + sc[((hb * 2 + k) * 16 + 4 * i + 2) * 64 + lane]) * inv_da;
          v.w = (dw1[k][4 * i + 3] + sc[((hb * 2 + k) * 16 + 4 * i + 3) * 64 + lane]) * inv_da;
          // D[row = channel k * 32 + 8 i + 4 half + (0..3)][col = hidden row hrow]
          st4(dst + (size_t)hrow * C + k * 32 + 8 * i + 4 * half, v);
        }
      if (half == 0) {
        a.db1_part[(size_t)blockIdx.x * HID + hrow] = vb + ss[(hb * 2 + 0) * 32 + l31];
        a.dw2_part[(size_t)blockIdx.x * HID + hrow] = vw + ss[(hb * 2 + 1) * 32 + l31];
      }
    }
  }
}
static inline size_t proj_bwd_q_lds(int HID, int W, int NJ, bool x1g) {
  return (size_t)2 * 64 * 128 + (size_t)4 * HID * 128 + 64 * 4 +
         (x1g ? (size_t)16 * NJ * (W + 4) * 4 + (W > 64 ? (size_t)4 * NJ * 64 * 16 : 0) : 0);
}
