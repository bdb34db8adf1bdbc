// EXPERIMENT (round 5; NOT part of the library, timing only - its results were never checked against the oracle): backward of the
// fused projection MLP with wave ROLES (C = 64, hidden 256, one output channel, two fp16 terms).  Measured with tools/pbr_bench.hip at
// BASELINE config 2's shape: 0.657 ms per launch (scalar vector phase) / 0.634 (packed) against 0.61-0.62 for k_proj_bwd_t in the
// same harness: the single vector wave of a SIMD runs its dependent GELU' chains at one instruction per 9-10 cycles (3.7 k cycles per
// 32 x 32 sub-tile, where the stand-alone phase takes 2.3-2.5 k), so hiding the matrix products buys less than the lost second wave
// costs.  DESIGN.md section 4f.
// Same arguments, mathematics and partial-slab outputs as k_proj_bwd_t (k_projection2.h; reference: autograd of
// neuralop/models/tfno.py:23-38).
// Why roles.  Measured this round (tools/role_split_test.hip, profiles/r05_role_split_microbench.txt): on one SIMD a wave that
// runs the GELU' / split phase of this kernel takes 1615 cycles per 32 x 32 tile alone and 2766 beside a wave that keeps the
// matrix pipe busy - exactly the 36 x 32 cycles of that tile's products more: the packed-fp32 vector instructions and the
// matrix instructions of two waves do NOT run concurrently, whichever wave issues them (and four waves per SIMD change nothing:
// k_proj_bwd_q, k_projection3.h).  The same phase in scalar fp32 instructions takes 2309 cycles alone and 2547 beside the
// saturated matrix pipe: scalar vector work DOES run under matrix products.  k_proj_bwd_t's two waves per SIMD both alternate
// between the two kinds of work between the same barriers, so its time is their sum (SQ counters: 1.08e8 vector instructions,
// 1.05e7 products, nothing overlaps).  Here each SIMD has ONE vector wave and ONE matrix wave:
//   vector wave 4 + m (pixel block m of the 128-pixel tile): for the eight (chunk, hidden half) sub-tiles in turn it ISSUES the 12
//     recompute products of the NEXT sub-tile (P1^T[px][hid]; they run in the matrix pipe while it goes on), then runs GELU / GELU' /
//     dP1 = act'(P1) w2 dy / two-term split / db1, dW2 sums on the accumulators of the CURRENT one in SCALAR fp32, and writes dP1
//     into the swizzled [hidden][pixel] image of its chunk;
//   matrix wave m (0-3): one chunk behind, the dx product (24) and one 32 x 32 tile of the chunk's dW1 (24) from that image -
//     exactly k_proj_bwd_t's products - and, while the vector waves are in a tile's first chunk, the row DFT of the previous tile.
// One workgroup barrier per chunk (the dP1 image is double-buffered), one before the epilogue.
//   LDS: a image 32 KB | two dP1 images 64 KB | dy | gout tile 33 KB | row table.
#pragma once
#include "../../pde_policylearning_amd/csrc/fno_dev.h"
#include "../../pde_policylearning_amd/csrc/k_block_bwd2.h"
#include "../../pde_policylearning_amd/csrc/k_projection.h"
#include "../../pde_policylearning_amd/csrc/k_projection2.h"

// two-term split, element-wise (no packed fp32 instruction: those stall beside the matrix pipe)
FNO_DEV void split2_scalar4(const float (&v)[4], float s, unsigned& h0, unsigned& h1, unsigned& l0, unsigned& l1) {
  _Float16 hh[4], ll[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    float t = v[j] * s;
    asm volatile("" : "+v"(t));            // (keeps any packing pass from re-pairing the elements)
    hh[j] = (_Float16)t;
    float r = t - (float)hh[j];
    asm volatile("" : "+v"(r));
    ll[j] = (_Float16)r;
  }
  h0 = __builtin_bit_cast(unsigned, f16x2{hh[0], hh[1]}); h1 = __builtin_bit_cast(unsigned, f16x2{hh[2], hh[3]});
  l0 = __builtin_bit_cast(unsigned, f16x2{ll[0], ll[1]}); l1 = __builtin_bit_cast(unsigned, f16x2{ll[2], ll[3]});
}

#ifndef PBR_PACKED_E
#define PBR_PACKED_E 0      // 1: the vector waves use the packed GELU / split forms (A/B arm)
#endif
// -DPBR_TRACE (tools/pbr_bench.hip): shader-clock stamps of workgroup 0, every wave: [wave][tile < 8][slot < 8][4]: at the slot's
// barrier, behind it, at the end of the slot's work; slot 6 = epilogue
#ifdef PBR_TRACE
__device__ unsigned long long g_pbr[8 * 8 * 8 * 4];
#define PBR_STAMP(slot, k) do { if (blockIdx.x == 0 && pbr_t < 8 && (threadIdx.x & 63) == 0) \
    g_pbr[(((threadIdx.x >> 6) * 8 + pbr_t) * 8 + (slot)) * 4 + (k)] = __builtin_readcyclecounter(); } while (0)
#else
#define PBR_STAMP(slot, k) do { } while (0)
#endif

// one matrix product, then PBR_VPM vector instructions, twelve times (sched_group_barrier masks: 0x8 MFMA, 0x2 VALU, 0x100 DS read)
#ifndef PBR_VPM
#define PBR_VPM 22
#endif
#define PBR_SPREAD() do { _Pragma("unroll") for (int q_ = 0; q_ < 12; ++q_) { \
    __builtin_amdgcn_sched_group_barrier(0x100, 2, 0); __builtin_amdgcn_sched_group_barrier(0x8, 1, 0); \
    __builtin_amdgcn_sched_group_barrier(0x2, PBR_VPM, 0); } } while (0)
template <int HID, bool RELU = false>
__global__ void __launch_bounds__(512) k_proj_bwd_r(ProjBwdArgs a) {
  constexpr int C = 64, NPX = 128, NT = 512, KB = 4, NCH = HID / 64, PITCH = NPX + 4;
  constexpr int ATERM = C * 256, DTERM = 64 * 256;          // bytes per term plane of the a image / one dP1 image
  extern __shared__ __attribute__((aligned(16))) float smem[];
  unsigned char* aimg = reinterpret_cast<unsigned char*>(smem);             // [2][64 c][128 px] fp16, swz_off
  unsigned char* dr0 = aimg + 2 * ATERM;                                    // two dP1 images [2][64 hid][128 px] fp16, swz_off
  float* douts = reinterpret_cast<float*>(dr0 + 2 * 2 * DTERM);             // dy of the tile (128)
  unsigned* gcnt = reinterpret_cast<unsigned*>(douts + NPX);                // [0] matrix-wave, [1] vector-wave group barrier counters
  float* r3 = douts + NPX + 4;                                              // gout tile C x PITCH (its own region: the matrix waves
                                                                            // transform it while the vector waves fill the dP1 images)
  float* tfwd_s = r3 + C * PITCH;                                           // 16 NJ x (W + 4): forward row table (if x1g)
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool matrix = wave < 4;
  const int nt = wave & 3, n0 = nt * 32;
  const int l31 = lane & 31, half = lane >> 5, l15 = lane & 15, quad = lane >> 4;
  const int tq = l15 >> 2, tp = l15 & 3;
  const int tpx = n0 + 16 * (quad & 1) + 4 * tp, trow = 8 * (quad >> 1) + tq;      // transposed-read roles (k_block_bwd2.h)
  const int dmt = nt >> 1, dnt = nt & 1;                   // dW1 tile of matrix wave nt: hidden 32-block, channel 32-block of the chunk

  const float sa = h2_scale(*a.xmax), sw = h2_scale(a.amax[2]), sd = h2_scale(1.13f * a.amax[3] * a.amax[1]);
  const float inv_aw = 1.f / (sa * sw), inv_dw = 1.f / (sd * sw), inv_da = 1.f / (sd * sa);
  float gk_six, gk_inf;
  gelu_consts(gk_six, gk_inf);

  float gvmax = 0.f;

  const __amdgpu_buffer_rsrc_t rs_wa1 = make_rsrc(a.wa1, (unsigned)((HID / 32) * KB * 2 * 64 * 16));
  const __amdgpu_buffer_rsrc_t rs_wb3 = make_rsrc(a.wa3, (unsigned)((HID / 16) * 2 * 2 * 64 * 16));
  if (a.x1g)
    for (int i = tid; i < 16 * a.NJ * a.W; i += NT) tfwd_s[(i / a.W) * (a.W + 4) + i % a.W] = a.tfwd[i];
  if (tid < 2) gcnt[tid] = 0u;
  __syncthreads();
  unsigned gepoch = 0;               // arrivals so far at this role's group barrier (k_block_bwd2.h::group_barrier: four waves)
#ifdef PBR_TRACE
  int pbr_t = -1;
#endif

  // The two roles run their OWN tile loops (same barrier sequence in both: one per chunk slot, one before the epilogue, one behind the last tile): a register of
  // one role is never live in the other's code, so the allocation is the larger of the two, not their union.
  if (matrix) {
    f32x16 dw1acc[NCH];                 // dW1 tile (dmt, dnt) of every chunk
#pragma unroll
    for (int k = 0; k < NCH; ++k)
#pragma unroll
      for (int r = 0; r < 16; ++r) dw1acc[k][r] = 0.f;
    for (int tile = blockIdx.x; tile < a.ntiles; tile += gridDim.x) {
#ifdef PBR_TRACE
      ++pbr_t;
#endif
      const int b = tile / a.tiles_per_plane;
      const int px0 = (tile % a.tiles_per_plane) * NPX;
      f32x16 dxh[2], dxl[2];         // dx^T[px block nt][channel block cb]: hh products / cross terms
#pragma unroll
      for (int cb = 0; cb < 2; ++cb)
#pragma unroll
        for (int r = 0; r < 16; ++r) { dxh[cb][r] = 0.f; dxl[cb][r] = 0.f; }
#pragma unroll 1
      for (int s = 1; s <= NCH; ++s) {
        PBR_STAMP(s, 0);
        __syncthreads();                             // slot s: dP1 of chunk s - 1 is complete
        PBR_STAMP(s, 1);
        {
          // ---- dx and dW1 of chunk s - 1.  W1 fragments from L2 (buffer loads, 1 KB each) into one set of eight:
          //   [W <- dx kk = 0, 1]  dW1 (operands from LDS only: hides the load)  dx(0, 1)  [W <- dx kk = 2, 3]  dx(2, 3)
          const int ch = s - 1;
          const unsigned char* dr = dr0 + (ch & 1) * 2 * DTERM;
          bf16x8 w0[4][2];
          auto load_x = [&](int k0) {
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
              for (int t = 0; t < 2; ++t)
                w0[q][t] = buf_ld8h(rs_wb3, lane * 16, ((((ch * 4 + k0 + (q >> 1)) * 2 + (q & 1)) * 2) + t) * 1024);
          };
          auto dx_pair = [&](int k0) {
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
              bf16x8 af[2];
              const int o0 = swz_off((k0 + kk) * 16 + trow, tpx >> 3) + 2 * (tpx & 7);
              const int o1 = swz_off((k0 + kk) * 16 + trow + 4, tpx >> 3) + 2 * (tpx & 7);
#pragma unroll
              for (int t = 0; t < 2; ++t) af[t] = cat4(lds_tr16(dr + t * DTERM + o0), lds_tr16(dr + t * DTERM + o1));
#pragma unroll
              for (int cb = 0; cb < 2; ++cb) mfma_split_s<2>(af, w0[kk * 2 + cb], dxh[cb], dxl[cb]);
            }
          };
          load_x(0);
          {
            const int ro = dmt * 32 + l31, rc = dnt * 32 + l31;
#pragma unroll
            for (int k = 0; k < NCH; ++k)
              if (k == ch) {
                f32x16 dacc = dw1acc[k];
#pragma unroll 1
                for (int kq = 0; kq < NPX / 16; ++kq) {
                  const int chn = 2 * kq + half;
                  const int od = swz_off(ro, chn), oa = swz_off(rc, chn);
                  bf16x8 af[2], bf[2];
#pragma unroll
                  for (int t = 0; t < 2; ++t) {
                    af[t] = *reinterpret_cast<const bf16x8*>(dr + t * DTERM + od);
                    bf[t] = *reinterpret_cast<const bf16x8*>(aimg + t * ATERM + oa);
                  }
                  dacc = mfma_split<2>(af, bf, dacc);
                }
                dw1acc[k] = dacc;
              }
          }
          __builtin_amdgcn_sched_barrier(0);
          dx_pair(0);
          __builtin_amdgcn_sched_barrier(0);
          load_x(2);
          dx_pair(2);
        }
        PBR_STAMP(s, 2);
      }
      PBR_STAMP(6, 0);
      // ---- epilogue: x act'(u), gout store; then the gout tile for the row DFT (next tile's slot 0) ------------------------------
      __syncthreads();               // E1: every dx / dW1 read of the dP1 images and of the a image is done
#pragma unroll
      for (int cb = 0; cb < 2; ++cb) {
        const int crow = cb * 32 + l31;
        const size_t ro = ((size_t)b * C + crow) * a.PW + px0 + n0 + 4 * half;
        float4 v[4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
          v[i] = make_float4((dxh[cb][4 * i] + dxl[cb][4 * i]) * inv_dw, (dxh[cb][4 * i + 1] + dxl[cb][4 * i + 1]) * inv_dw,
                             (dxh[cb][4 * i + 2] + dxl[cb][4 * i + 2]) * inv_dw, (dxh[cb][4 * i + 3] + dxl[cb][4 * i + 3]) * inv_dw);
        if (a.act_in) {
          const __amdgpu_buffer_rsrc_t rs = make_rsrc(a.x + (size_t)b * C * a.PW, (unsigned)(C * a.PW * 4));
          float4 uq[4];
#pragma unroll
          for (int i = 0; i < 4; ++i) uq[i] = buf_ld4(rs, (crow * a.PW + n0 + 4 * half) * 4, (px0 + 8 * i) * 4);
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            float4 uu = uq[i], dd;
            gelu_both4(uu, dd);
            v[i].x *= dd.x; v[i].y *= dd.y; v[i].z *= dd.z; v[i].w *= dd.w;
          }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) st4(a.gout + ro + 8 * i, v[i]);
        if (a.gmax_out) {
#pragma unroll
          for (int i = 0; i < 4; ++i) gvmax = fmaxf(fmaxf(gvmax, fabsf(v[i].x)), fmaxf(fmaxf(fabsf(v[i].y), fabsf(v[i].z)), fabsf(v[i].w)));
        }
        if (a.x1g) {
          float* r3p = r3 + crow * PITCH + n0 + 4 * half;
#pragma unroll
          for (int i = 0; i < 4; ++i) st4(r3p + 8 * i, v[i]);
        }
      }
      // the row DFT of this tile's gradient, by the matrix waves alone (their own barrier: the vector waves are already in
      // the next tile's first chunk)
      if (a.x1g) {
        group_barrier(gcnt, gepoch, lane);
        row_dft_epilogue<C, NPX, 4>(r3, tfwd_s, a.W + 4, a.x1g, b, px0, a.P, a.W, a.K2out, a.NJ, wave, lane);
      }
    }
    // partial slabs (layout of k_proj_bwd_t): dW1 one slab per workgroup
    if (a.gmax_out) absmax_publish(gvmax, a.gmax_out);
    float* dst = a.dw1_part + (size_t)blockIdx.x * HID * C;
#pragma unroll
    for (int k = 0; k < NCH; ++k)
#pragma unroll
      for (int r = 0; r < 16; ++r)
        dst[(size_t)(k * 64 + dmt * 32 + acc_row32(r, half)) * C + dnt * 32 + l31] = dw1acc[k][r] * inv_da;
  } else {
    float sdb1[2 * NCH], sdw2[2 * NCH];      // db1 / dW2 sums of sub-tile j = 2 ch + hm over this wave's pixel block
#pragma unroll
    for (int j = 0; j < 2 * NCH; ++j) { sdb1[j] = 0.f; sdw2[j] = 0.f; }
    // the tile's rows of u_L, fetched and committed by the vector waves: thread (c = vt / 32 + 8 i, q = vt % 32) of the 256
    // vector threads loads 16 bytes, i = 0..7
    float4 xq[8];
    const int vt = tid & 255;
    const int xvoff = ((vt >> 5) * a.PW + 4 * (vt & 31)) * 4;
    auto issue_x = [&](int tile) {
      const int b = tile / a.tiles_per_plane;
      const int px0 = (tile % a.tiles_per_plane) * NPX;
      const __amdgpu_buffer_rsrc_t rs = make_rsrc(a.x + (size_t)b * C * a.PW, (unsigned)(C * a.PW * 4));
#pragma unroll
      for (int i = 0; i < 8; ++i) xq[i] = buf_ld4(rs, xvoff, (8 * i * a.PW + px0) * 4);
    };
    // a = act(u) -> swizzled [c][px] image (one split) and the dy row of a tile
    auto commit = [&](int tile) {
      const int b = tile / a.tiles_per_plane;
      const int px0 = (tile % a.tiles_per_plane) * NPX;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int c = (vt >> 5) + 8 * i, q = vt & 31;
        float4 t = xq[i];
        if (a.act_in) t = gelu4(t, gk_six, gk_inf);
        put_split4_n<2>(aimg, ATERM, swz_off(c, q >> 1) + 8 * (q & 1), t, sa);
      }
      if (vt < NPX) douts[vt] = a.dy[(size_t)b * a.PW + px0 + vt];
    };
    if ((int)blockIdx.x < a.ntiles) { issue_x(blockIdx.x); commit(blockIdx.x); }
    // per-lane constants of every (chunk, hidden half): b1 and w2 of row 64 ch + 32 hm + l31
    float b1r[2 * NCH], w2r[2 * NCH];
#pragma unroll
    for (int j = 0; j < 2 * NCH; ++j) { b1r[j] = a.b1[j * 32 + l31]; w2r[j] = a.w2[j * 32 + l31]; }
    // recompute of sub-tile j = 2 ch + hm: P1^T[px block nt][hidden rows 32 j ..]: A = transposed reads of the a image, B = W1
    // fragments from L2 (requested one sub-tile ahead into `wn`)
    bf16x8 wn[KB][2];
    auto load_w = [&](int j) {
#pragma unroll
      for (int kb = 0; kb < KB; ++kb)
#pragma unroll
        for (int t = 0; t < 2; ++t) wn[kb][t] = buf_ld8h(rs_wa1, lane * 16, ((j * KB + kb) * 2 + t) * 1024);
    };
    auto recompute = [&](f32x16& hi, f32x16& lo) {
#pragma unroll
      for (int r = 0; r < 16; ++r) { hi[r] = 0.f; lo[r] = 0.f; }
#pragma unroll
      for (int kb = 0; kb < KB; ++kb) {
        bf16x8 af[2];
        const int o0 = swz_off(kb * 16 + trow, tpx >> 3) + 2 * (tpx & 7);
        const int o1 = swz_off(kb * 16 + trow + 4, tpx >> 3) + 2 * (tpx & 7);
#pragma unroll
        for (int t = 0; t < 2; ++t) af[t] = cat4(lds_tr16(aimg + t * ATERM + o0), lds_tr16(aimg + t * ATERM + o1));
        mfma_split_s<2>(af, wn[kb], hi, lo);
      }
    };
    load_w(0);
    for (int tile = blockIdx.x; tile < a.ntiles; tile += gridDim.x) {
#ifdef PBR_TRACE
      ++pbr_t;
#endif
      f32x16 hiA, loA, hiB, loB;     // accumulators of the current / next sub-tile (they alternate)
      PBR_STAMP(0, 0);
      group_barrier(gcnt + 1, gepoch, lane);      // the a image and the dy row of this tile are committed (by the four vector waves)
      PBR_STAMP(0, 1);
      recompute(hiA, loA);           // sub-tile 0 (nothing to run beside it)
      load_w(1);
      // ---- E of sub-tile j on (hc, lc) while the products of sub-tile j + 1 run.  lane <-> hidden row; registers <-> pixels
      //      n0 + (r & 3) + 8 (r >> 2) + 4 half
      auto vector_phase = [&](int j, const f32x16& hc, const f32x16& lc) {
        const int ch = j >> 1, hm = j & 1;
        unsigned char* dr = dr0 + (ch & 1) * 2 * DTERM;
        const int hrow = hm * 32 + l31;
        float b1v = 0.f, w2v = 0.f;
#pragma unroll
        for (int k = 0; k < 2 * NCH; ++k)
          if (k == j) { b1v = b1r[k]; w2v = w2r[k]; }
        float sdb = 0.f, sdw = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float4 dy4 = ld4(douts + n0 + 8 * i + 4 * half);
          const float dyv[4] = {dy4.x, dy4.y, dy4.z, dy4.w};
          float pr[4];
#pragma unroll
          for (int q = 0; q < 4; ++q) pr[q] = fmaf(hc[4 * i + q] + lc[4 * i + q], inv_aw, b1v);
          float gl4[4], dg4[4], dp[4];
#if PBR_PACKED_E
          { float4 glv = make_float4(pr[0], pr[1], pr[2], pr[3]), dgv;
            if constexpr (RELU) {
              dgv = make_float4(glv.x > 0.f ? 1.f : 0.f, glv.y > 0.f ? 1.f : 0.f, glv.z > 0.f ? 1.f : 0.f, glv.w > 0.f ? 1.f : 0.f);
              glv = make_float4(fmaxf(glv.x, 0.f), fmaxf(glv.y, 0.f), fmaxf(glv.z, 0.f), fmaxf(glv.w, 0.f));
            } else gelu_both4(glv, dgv);
            gl4[0] = glv.x; gl4[1] = glv.y; gl4[2] = glv.z; gl4[3] = glv.w; dg4[0] = dgv.x; dg4[1] = dgv.y; dg4[2] = dgv.z; dg4[3] = dgv.w; }
#else
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            if constexpr (RELU) { dg4[q] = pr[q] > 0.f ? 1.f : 0.f; gl4[q] = fmaxf(pr[q], 0.f); }
            else gelu_both(pr[q], gl4[q], dg4[q]);
          }
#endif
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            dp[q] = dg4[q] * (w2v * dyv[q]);
            sdw = fmaf(gl4[q], dyv[q], sdw);
            sdb += dp[q];
          }
          const int od = swz_off(hrow, (n0 >> 3) + i) + 8 * half;
#if PBR_PACKED_E
          put_split4_n<2>(dr, DTERM, od, make_float4(dp[0], dp[1], dp[2], dp[3]), sd);
#else
          unsigned h0, h1, l0, l1;
          split2_scalar4(dp, sd, h0, h1, l0, l1);
          *reinterpret_cast<uint2*>(dr + od) = make_uint2(h0, h1);
          *reinterpret_cast<uint2*>(dr + DTERM + od) = make_uint2(l0, l1);
#endif
          asm volatile("" : "+v"(sdb), "+v"(sdw));
        }
#pragma unroll
        for (int k = 0; k < 2 * NCH; ++k)
          if (k == j) { sdb1[k] += sdb; sdw2[k] += sdw; }
      };
#pragma unroll 1
      for (int ch = 0; ch < NCH; ++ch) {
        if (ch > 0) {
          PBR_STAMP(ch, 0);
          __syncthreads();           // slot ch: dP1 of chunk ch - 1 is complete (and the image of chunk ch - 2 is free again)
          PBR_STAMP(ch, 1);
        }
        // sub-tile 2 ch on A while 2 ch + 1 is produced into B, then 2 ch + 1 on B while 2 ch + 2 is produced into A
        // (the products are SPREAD through the vector phase by the scheduler directives below: a wave issues in order, and twelve
        // dependent products in a row would hold its vector stream for ~700 cycles)
        __builtin_amdgcn_sched_barrier(0);
        recompute(hiB, loB);
        vector_phase(2 * ch, hiA, loA);
        PBR_SPREAD();
        __builtin_amdgcn_sched_barrier(0);
        load_w(2 * ch + 2 < 2 * NCH ? 2 * ch + 2 : 0);
        if (ch == NCH - 1) {                         // the next tile's rows: in flight behind the last sub-tile
          const int nt2 = tile + gridDim.x;
          if (nt2 < a.ntiles) issue_x(nt2);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (ch + 1 < NCH) {
          recompute(hiA, loA);
          vector_phase(2 * ch + 1, hiB, loB);
          PBR_SPREAD();
          __builtin_amdgcn_sched_barrier(0);
          load_w(2 * ch + 3);
        } else {
          vector_phase(2 * ch + 1, hiB, loB);
        }
        __builtin_amdgcn_sched_barrier(0);
        PBR_STAMP(ch, 2);
      }
      PBR_STAMP(NCH, 0);
      __syncthreads();               // slot NCH: dP1 of the last chunk is complete (the matrix waves finish dx / dW1)
      PBR_STAMP(NCH, 1);
      PBR_STAMP(NCH, 2);
      PBR_STAMP(6, 0);
      __syncthreads();               // E1: the a image, the dy row and both dP1 images are free
      {
        const int nt2 = tile + gridDim.x;
        if (nt2 < a.ntiles) commit(nt2);             // (its rows were requested behind the last sub-tile)
      }
    }
    // partial slabs (layout of k_proj_bwd_t): db1 / dW2 one per pixel block
    const size_t slab = (size_t)blockIdx.x * 4 + nt;
#pragma unroll
    for (int j = 0; j < 2 * NCH; ++j) {
      const float vb = sdb1[j] + __shfl_xor(sdb1[j], 32, 64);
      const float vw = sdw2[j] + __shfl_xor(sdw2[j], 32, 64);
      if (half == 0) {
        a.db1_part[slab * HID + j * 32 + l31] = vb;
        a.dw2_part[slab * HID + j * 32 + l31] = vw;
      }
    }
  }
}
static inline size_t proj_bwd_r_lds(int W, int NJ, bool x1g) {
  return (size_t)2 * 64 * 256 + (size_t)4 * 64 * 256 + 128 * 4 + 16 + (size_t)64 * 132 * 4 + (x1g ? (size_t)16 * NJ * (W + 4) * 4 : 0);
}
