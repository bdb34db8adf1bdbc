// Backward of the fused projection MLP, second generation (C = 64; same arguments, partial-slab outputs and mathematics as
// k_proj_bwd_x3 in k_projection.h; reference: autograd of neuralop/models/tfno.py:23-38).  Built on the block backward's
// findings (k_block_bwd2.h): every GEMM whose result feeds the dx chain keeps the hh products and the cross terms of the
// 3-way split in separate accumulators; ONE [channel][pixel] image of a = act(u_L) serves the recompute (transposed
// reads) and dW1 (row reads); the hidden chunk is computed TRANSPOSED (P1^T[pixel][hidden]: lane <-> hidden row), so
//   * b1 / W2 are per-lane scalars, the dW2 / db1 reductions are lane-local sums (no DPP butterflies, no LDS reads),
//   * the dP1 chunk goes to its row-major image as 8-byte stores of 4 consecutive pixels (was: 48 two-byte stores),
//   * dx^T[pixel][channel] accumulates in ONE tile per wave (pixel block nt, channel block hm): no partial-sum exchange.
// The dP1 image is double-buffered: ONE barrier per 64-row chunk; the wave group that owns a chunk's dW1 (48 MFMAs) runs
// beside its SIMD partners' recompute + GELU phase of the next chunk.
//   per chunk:  A1 recompute (24 MFMAs) | E gelu, gelu', dP1 split -> dr[ch & 1] | barrier | A3 dx += (24) | owners: dW1 (48)
#pragma once
#include "fno_dev.h"
#include "k_block_bwd2.h"
#include "k_projection.h"

// W1 (HID, C) fp32 -> bf16x3 fragments for k_proj_bwd_t: wa1 as k_pack_w1_x3; the dx product's B fragments in natural k order
//   wb3[(((ch*4 + kb)*MT + cb)*3 + t)*64 + lane][j] = term t of W1[ch*64 + kb*16 + 8*(lane>>5) + j][cb*32 + (lane&31)]
// NTERM = 2: two fp16 terms of h2_scale(*wmax) * W1 (fno_dev.h "h2"; wmax = device scalar max |W1|, k_absmax)
template <int NTERM>
FNO_DEV void pack_w1_t_item(const float* __restrict__ w1, unsigned short* __restrict__ wa1, unsigned short* __restrict__ wb3,
                            int HID, int C, float sw, int it) {
  const int KB = C / 16, MT = C / 32;
  const int n1 = (HID / 32) * KB * 64, n3 = (HID / 16) * MT * 64;
  float v[8];
  bf16x8 f[NTERM];
  if (it < n1) {
    const int ln = it & 63, kb = (it >> 6) % KB, mt = (it >> 6) / KB;
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = w1[(size_t)(mt * 32 + (ln & 31)) * C + kb * 16 + 8 * (ln >> 5) + j];
    split_n_x8<NTERM>(v, sw, f);
    unsigned short* dst = wa1 + ((size_t)((mt * KB + kb) * NTERM) * 64 + ln) * 8;
#pragma unroll
    for (int t = 0; t < NTERM; ++t) st8h(dst + t * 64 * 8, f[t]);
  } else if (it < n1 + n3) {
    const int i3 = it - n1;
    const int ln = i3 & 63, cb = (i3 >> 6) % MT, kh = (i3 >> 6) / MT;      // kh = ch*4 + kb: 16-row block of W1
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = w1[(size_t)(kh * 16 + 8 * (ln >> 5) + j) * C + cb * 32 + (ln & 31)];
    split_n_x8<NTERM>(v, sw, f);
    unsigned short* dst = wb3 + ((size_t)((kh * MT + cb) * NTERM) * 64 + ln) * 8;
#pragma unroll
    for (int t = 0; t < NTERM; ++t) st8h(dst + t * 64 * 8, f[t]);
  }
}
template <int NTERM>
__global__ void k_pack_w1_t(const float* __restrict__ w1, unsigned short* __restrict__ wa1, unsigned short* __restrict__ wb3,
                            int HID, int C, const float* __restrict__ wmax) {
  const float sw = NTERM == 2 ? h2_scale(*wmax) : 1.f;
  pack_w1_t_item<NTERM>(w1, wa1, wb3, HID, C, sw, blockIdx.x * blockDim.x + threadIdx.x);
}
// max |x| of three arrays in one launch -> dst[0..2] (atomic max of the float pattern; zeroed by the caller): blocks
// [0, g0) scan x0, [g0, g0 + g1) x1, the rest x2
// sum0 (or null): the blocks of job 0 also leave the SUM of their share of x0 in sum0[block] (g0 partial sums, fixed order:
// the bias gradient of a one-channel projection is the sum of dy, and this launch reads dy anyway)
FNO_DEV void absmax3_block(const float* __restrict__ x0, size_t n0, int g0, const float* __restrict__ x1, size_t n1, int g1,
                           const float* __restrict__ x2, size_t n2, int g2, float* __restrict__ dst, float* __restrict__ sum0) {
  const int bi = blockIdx.x;
  const int job = bi < g0 ? 0 : (bi < g0 + g1 ? 1 : 2);
  const float* x = job == 0 ? x0 : (job == 1 ? x1 : x2);
  const size_t n = job == 0 ? n0 : (job == 1 ? n1 : n2);
  const int b0 = job == 0 ? 0 : (job == 1 ? g0 : g0 + g1), nb = job == 0 ? g0 : (job == 1 ? g1 : g2);
  float m = 0.f, sm = 0.f;
  if ((n & 3) == 0 && (reinterpret_cast<uintptr_t>(x) & 15) == 0) {      // 16-byte loads, four in flight per thread (the scalar
    const size_t n4 = n / 4, stride = (size_t)nb * blockDim.x;             // loop was 16 dependent 4-byte loads: 21 us per launch)
    size_t i = (size_t)(bi - b0) * blockDim.x + threadIdx.x;
    for (; i + 3 * stride < n4; i += 4 * stride) {
      const float4 a = ld4(x + 4 * i), b = ld4(x + 4 * (i + stride)), c = ld4(x + 4 * (i + 2 * stride)), d = ld4(x + 4 * (i + 3 * stride));
      m = fmaxf(m, fmaxf(fmaxf(fmaxf(fabsf(a.x), fabsf(a.y)), fmaxf(fabsf(a.z), fabsf(a.w))),
                         fmaxf(fmaxf(fabsf(b.x), fabsf(b.y)), fmaxf(fabsf(b.z), fabsf(b.w)))));
      m = fmaxf(m, fmaxf(fmaxf(fmaxf(fabsf(c.x), fabsf(c.y)), fmaxf(fabsf(c.z), fabsf(c.w))),
                         fmaxf(fmaxf(fabsf(d.x), fabsf(d.y)), fmaxf(fabsf(d.z), fabsf(d.w)))));
      sm += (((a.x + a.y) + (a.z + a.w)) + ((b.x + b.y) + (b.z + b.w))) + (((c.x + c.y) + (c.z + c.w)) + ((d.x + d.y) + (d.z + d.w)));
    }
    for (; i < n4; i += stride) {
      const float4 a = ld4(x + 4 * i);
      m = fmaxf(m, fmaxf(fmaxf(fabsf(a.x), fabsf(a.y)), fmaxf(fabsf(a.z), fabsf(a.w))));
      sm += (a.x + a.y) + (a.z + a.w);
    }
  } else {
    for (size_t i = (size_t)(bi - b0) * blockDim.x + threadIdx.x; i < n; i += (size_t)nb * blockDim.x) { m = fmaxf(m, fabsf(x[i])); sm += x[i]; }
  }
  // one publish per workgroup (same-address atomics serialise)
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) { m = fmaxf(m, __shfl_xor(m, o, 64)); sm += __shfl_xor(sm, o, 64); }
  __shared__ float wm[16], ws[16];
  if ((threadIdx.x & 63) == 0) { wm[threadIdx.x >> 6] = m; ws[threadIdx.x >> 6] = sm; }
  __syncthreads();
  if (sum0 && job == 0 && threadIdx.x == 0) {
    float t = 0.f;
    for (int k = 0; k < (int)(blockDim.x >> 6); ++k) t += ws[k];
    sum0[bi - b0] = t;
  }
  if (threadIdx.x < 64) {
    const int nw = (int)(blockDim.x >> 6);
    float r = threadIdx.x < nw ? wm[threadIdx.x] : 0.f;
    absmax_publish(r, dst + job);
  }
}
// the scan (max |dy|, max |W1|, max |w2|, partial sums of dy) and k_pack_w1_t<2> in ONE launch (round 5: two ~5 us launches at the head of every backward pass): blocks
// [0, g0 + g1 + g2) are k_absmax3's, the rest split x1 = W1 (HID x C) into its two fp16 terms.  The split needs max |W1|, which
// the scan blocks of the same launch are still producing: every split block takes the maximum over W1 itself (64 KB out of
// L2, 16 loads per thread) - the same float the scan publishes in dst[1], so the kernels that read the bound see the scale
// the fragments were made with.
__global__ void __launch_bounds__(256) k_absmax3_pack_w1(const float* __restrict__ x0, size_t n0, int g0, const float* __restrict__ w1,
                                                         int g1, const float* __restrict__ x2, size_t n2, int g2,
                                                         float* __restrict__ dst, float* __restrict__ sum0,
                                                         unsigned short* __restrict__ wa1, unsigned short* __restrict__ wb3,
                                                         int HID, int C) {
  const int nscan = g0 + g1 + g2;
  if ((int)blockIdx.x < nscan) {
    absmax3_block(x0, n0, g0, w1, (size_t)HID * C, g1, x2, n2, g2, dst, sum0);
    return;
  }
  // max of the BIT PATTERNS of |w| - as absmax_publish forms the published bound (a NaN orders above every number there; fmaxf
  // would drop it here and the fragments' scale would disagree with the bound the consumers read: ADVICE r05)
  unsigned mu = 0u;
  auto upd = [&](float v) { const unsigned u = __builtin_bit_cast(unsigned, v) & 0x7fffffffu; mu = u > mu ? u : mu; };
  const int n4 = HID * C / 4;                        // (C is 32 or 64: whole float4s)
  if ((reinterpret_cast<uintptr_t>(w1) & 15) == 0) {
    for (int i = threadIdx.x; i < n4; i += 256) {
      const float4 a = ld4(w1 + 4 * i);
      upd(a.x); upd(a.y); upd(a.z); upd(a.w);
    }
  } else {                                           // a parameter that is a view at an odd offset of a flat bucket
    for (int i = threadIdx.x; i < 4 * n4; i += 256) upd(w1[i]);
  }
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) { const unsigned t = (unsigned)__shfl_xor((int)mu, o, 64); mu = t > mu ? t : mu; }
  __shared__ unsigned wmx[4];
  if ((threadIdx.x & 63) == 0) wmx[threadIdx.x >> 6] = mu;
  __syncthreads();
  mu = max(max(wmx[0], wmx[1]), max(wmx[2], wmx[3]));
  const float m = __builtin_bit_cast(float, mu);
  pack_w1_t_item<2>(w1, wa1, wb3, HID, C, h2_scale(m), ((int)blockIdx.x - nscan) * 256 + threadIdx.x);
}

// C = 32: the dx product's K (hidden rows) is split between the two wave groups (wave (hm, nt) contracts over rows
// 32 hm .. 32 hm + 31 of every chunk: one channel block, two partial tiles per pixel block, added through LDS once per tile) and
// a chunk's dW1 (two 32 x 32 tiles) is split over the owner group's four waves by pixel halves (added through LDS at the end).
// NT3 = 3: three bf16 terms, six products per k block; NT3 = 2: two fp16 terms, three products (fno_dev.h "h2"), operands
// scaled by powers of two from a.amax = {max |x|, max |dy|, max |W1|, max |w2|} (device scalars)
#ifndef FNO_PB_SDFOLD
#define FNO_PB_SDFOLD 1      // 0: the split multiplies by the fp16 scale itself (A/B arm)
#endif
template <int C, int HID, bool RELU = false, int NT3 = 3>
__global__ void __launch_bounds__(512, 2) k_proj_bwd_t(ProjBwdArgs a) {
  FNO_CLK_ENTRY();
  constexpr int NPX = 128, NTN = 4, NT = 512, KB = C / 16, MT = C / 32, XI = C / 16;
  constexpr bool LINE_ST = NT3 == 2 || MT == 1;      // gout leaves in whole lines from the LDS tile (not the three-term 64-channel
                                                       // kernel: it has no register to spare for it)
  static_assert(C == 32 || C == 64, "32 or 64 channels");
  constexpr int NCH = HID / 64, CPW = NCH / 2;
  constexpr int PITCH = NPX + 4;
  constexpr int ATERM = C * 256, DTERM = 64 * 256;          // bytes per term plane of the a image / one dP1 buffer
  static_assert(NCH % 2 == 0, "two owner groups");
  static_assert((size_t)64 * PITCH * 4 <= (size_t)2 * NT3 * DTERM, "the gout tile / the partial-sum exchange alias the dP1 buffers");
  // C = 32: the partial-sum exchange [2][32][PITCH] sits at the start of the dP1 buffers, the gout tile behind it
  constexpr int R3B_OFF = NT3 * DTERM;
  static_assert(C == 64 || (size_t)2 * 32 * (NT3 == 3 ? PITCH : NPX) * 4 <= (size_t)NT3 * DTERM, "C = 32: the exchange buffer fits dP1 buffer 0");
  extern __shared__ __attribute__((aligned(16))) float smem[];
  unsigned char* aimg = reinterpret_cast<unsigned char*>(smem);             // a = act(u_L): [3][C][128] bf16, swizzled
  unsigned char* dr0 = aimg + NT3 * ATERM;                                   // dP1 chunk, two buffers [NT3][64][128] x 16 bit, swizzled
  float* douts = reinterpret_cast<float*>(dr0 + 2 * NT3 * DTERM);            // dy row of the tile (NPX)
  float* r3 = reinterpret_cast<float*>(dr0);                                 // after the chunk loop: gout tile C x PITCH
  float* tfwd_s = douts + NPX;                                               // 16*NJ x (W + 4): forward row table (if x1g)
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, half = lane >> 5;
  const int l15 = lane & 15, quad = lane >> 4;
  const int hm = wave >> 2, nt = wave & 3;
  float gk_six, gk_inf;
  gelu_consts(gk_six, gk_inf);
  const int n0 = nt * 32;
  const int dmt = nt >> 1, dnt = MT == 2 ? (nt & 1) : 0;   // dW1 tile of an owner wave: hidden 32-block, channel 32-block
  const int dkh = MT == 2 ? 0 : (nt & 1);                   // C = 32: pixel half of the tile's K range
  // transposed-read lane roles (k_block_bwd2.h): lane 4q + p of a 16-lane group supplies row q, pixels 4p .. 4p + 3
  const int tq = l15 >> 2, tp = l15 & 3;
  const int tpx = n0 + 16 * (quad & 1) + 4 * tp;
  const int trow = 8 * (quad >> 1) + tq;

  // operand scales of the fp16 form (powers of two; 1 for bf16x3): a = act(u_L), W1, dP1 = act'(P1) w2 dy
  float sa = 1.f, sw = 1.f, sd = 1.f;
  if constexpr (NT3 == 2) {
    sa = h2_scale(*a.xmax); sw = h2_scale(a.amax[2]);
    sd = h2_scale(1.13f * a.amax[3] * a.amax[1]);         // |gelu'| <= 1.13 (ReLU: 1)
  }
  const float inv_aw = 1.f / (sa * sw), inv_dw = 1.f / (sd * sw), inv_da = 1.f / (sd * sa), inv_sd = 1.f / sd;
  f32x16 dw1acc[CPW];
#pragma unroll
  for (int k = 0; k < CPW; ++k)
#pragma unroll
    for (int r = 0; r < 16; ++r) dw1acc[k][r] = 0.f;
  float sdb1[NCH], sdw2[NCH];
#pragma unroll
  for (int ch = 0; ch < NCH; ++ch) { sdb1[ch] = 0.f; sdw2[ch] = 0.f; }
  float gvmax = 0.f;      // max |gout| of this thread (a.gmax_out)

  // ONE set of weight fragments, time-shared: the recompute's (wa1) during A1, the dx product's (wb3) during A3.  Buffer loads:
  // descriptor + fragment offset in SGPRs, one 32-bit lane offset
  const __amdgpu_buffer_rsrc_t rs_wa1 = make_rsrc(a.wa1, (unsigned)((HID / 32) * KB * NT3 * 64 * 16));
  const __amdgpu_buffer_rsrc_t rs_wb3 = make_rsrc(a.wa3, (unsigned)((HID / 16) * MT * NT3 * 64 * 16));
  bf16x8 wf[4][NT3];      // (KB used by the recompute, 4 / 2 by the dx product)
  auto load_wa1 = [&](int ch) {
#pragma unroll
    for (int kb = 0; kb < KB; ++kb)
#pragma unroll
      for (int t = 0; t < NT3; ++t) wf[kb][t] = buf_ld8h(rs_wa1, lane * 16, (((ch * 2 + hm) * KB + kb) * NT3 + t) * 1024);
  };
  constexpr int NK3 = MT == 2 ? 4 : 2;            // 16-row k blocks of the dx product per wave and chunk
  auto load_wb3 = [&](int ch) {
#pragma unroll
    for (int kk = 0; kk < NK3; ++kk) {
      const int kb = MT == 2 ? kk : 2 * hm + kk, cb = MT == 2 ? hm : 0;
#pragma unroll
      for (int t = 0; t < NT3; ++t) wf[kk][t] = buf_ld8h(rs_wb3, lane * 16, ((((ch * 4 + kb) * MT + cb) * NT3) + t) * 1024);
    }
  };
  // the tile's rows of u_L: thread (c = tid / 32 + 16 i, q = tid % 32) loads 16 bytes; per-sample descriptor
  float4 xq[XI];
  const int xvoff = ((tid >> 5) * a.PW + 4 * (tid & 31)) * 4;
  auto issue_x = [&](int tile_) {
    const int tile = a.rev ? a.ntiles - 1 - tile_ : tile_;
    const int b = tile / a.tiles_per_plane;
    const int px0 = (tile % a.tiles_per_plane) * NPX;
    const __amdgpu_buffer_rsrc_t rs = make_rsrc(a.x + (size_t)b * C * a.PW, (unsigned)(C * a.PW * 4));
#ifdef PBT_GLOBAL_X
#pragma unroll
    for (int i = 0; i < XI; ++i) xq[i] = ld4(a.x + ((size_t)b * C + (tid >> 5) + 16 * i) * a.PW + px0 + 4 * (tid & 31));
#else
#pragma unroll
    for (int i = 0; i < XI; ++i) xq[i] = buf_ld4s(rs, xvoff, (16 * i * a.PW + px0) * 4);
#endif
  };
  if ((int)blockIdx.x < a.ntiles) issue_x(blockIdx.x);
  if (a.x1g)
    for (int i = tid; i < 16 * a.NJ * a.W; i += NT) tfwd_s[(i / a.W) * (a.W + 4) + i % a.W] = a.tfwd[i];
  load_wa1(0);

#ifndef PBT_BASE_PRIO
#define PBT_BASE_PRIO 1
#endif
  // static priority for the later-dispatched half (MI355X_MICROARCH.md, two waves per SIMD, item 4): waves 4-7 lose every
  // VALU arbitration to their older partners otherwise (GELU phase 4 k vs 1.9 k cycles in the phase trace)
  const int base_prio = (PBT_BASE_PRIO && wave >= 4) ? 1 : 0;
  if (base_prio) __builtin_amdgcn_s_setprio(1);
  int tslot = 0;
  FNO_TRACE_IF(FNO_TRACE_WHICH == 1);
  FNO_CLK_BEGIN();
  for (int tile_ = blockIdx.x; tile_ < a.ntiles; tile_ += gridDim.x) {
    const int tile = a.rev ? a.ntiles - 1 - tile_ : tile_;
    const int b = tile / a.tiles_per_plane;
    const int px0 = (tile % a.tiles_per_plane) * NPX;
    FNO_STAMP(tslot + 0);
    // ---- commit: a = act(u) -> swizzled [c][px] image (one split) -----------------------------------------------------
#pragma unroll
    for (int i = 0; i < XI; ++i) {
      const int c = (tid >> 5) + 16 * i, q = tid & 31;
      float4 t = xq[i];
      if (a.act_in) t = gelu4(t, gk_six, gk_inf);      // (on pairs: every wave is in this phase together, nobody's matrix products to hide behind)
      put_split4_n<NT3>(aimg, ATERM, swz_off(c, q >> 1) + 8 * (q & 1), t, sa);
    }
    if (tid < NPX) douts[tid] = a.dy[(size_t)b * a.PW + px0 + tid];
    FNO_STAMP(tslot + 1);
    __syncthreads();
    FNO_STAMP(tslot + 2);
    f32x16 dxh, dxl;         // dx^T: hh products / cross terms (summed once, in the epilogue)
#pragma unroll
    for (int r = 0; r < 16; ++r) { dxh[r] = 0.f; dxl[r] = 0.f; }

#pragma unroll 1
    for (int ch = 0; ch < NCH; ++ch) {
      unsigned char* dr = dr0 + (ch & 1) * NT3 * DTERM;
      // per-lane constants of this chunk: hidden row ch*64 + hm*32 + l31 (L2-resident; used after the recompute)
      const float b1v = a.b1[ch * 64 + hm * 32 + l31], w2v = a.w2[ch * 64 + hm * 32 + l31];
      if (ch == 1) FNO_STAMP(tslot + 3);
      // ---- A1: P1^T[px][hid] = sum_c a[c][px] W1[hid][c]: A = transposed reads of the a image, B = W1 fragments ---------
      f32x16 acc;
      {
        f32x16 hi, lo;
#pragma unroll
        for (int r = 0; r < 16; ++r) { hi[r] = 0.f; lo[r] = 0.f; }
#pragma unroll
        for (int kb = 0; kb < KB; ++kb) {
          bf16x8 af[NT3];
          const int o0 = swz_off(kb * 16 + trow, tpx >> 3) + 2 * (tpx & 7);
          const int o1 = swz_off(kb * 16 + trow + 4, tpx >> 3) + 2 * (tpx & 7);
#pragma unroll
          for (int t = 0; t < NT3; ++t) af[t] = cat4(lds_tr16(aimg + t * ATERM + o0), lds_tr16(aimg + t * ATERM + o1));
          mfma_split_s<NT3>(af, wf[kb], hi, lo);
          __builtin_amdgcn_sched_barrier(0);      // keep at most one k block of operand fragments live
        }
        // (pairs: with the SLP vectorizer off - build.py - element-wise source is element-wise code; the accumulator registers
        // are consecutive, so the pairs are natural ones.  Round 5: the vector phase's bookkeeping around the GELU by hand on pairs)
#pragma unroll
        for (int r = 0; r < 16; r += 2) {
          f32x2 sm = f32x2{hi[r], hi[r + 1]} + f32x2{lo[r], lo[r + 1]};
          if (NT3 == 2) sm = sm * f32x2{inv_aw, inv_aw};
          acc[r] = sm[0]; acc[r + 1] = sm[1];
        }
      }
      load_wb3(ch);            // the dx product's fragments arrive while the GELU phase runs
      if (ch == 1) FNO_STAMP(tslot + 4);
      // ---- E: lane <-> hidden row; registers <-> pixels n0 + (r&3) + 8 (r>>2) + 4 half ------------------------------------
      {
        f32x2 sdb = {0.f, 0.f}, sdw = {0.f, 0.f};      // even / odd pixels; added at the end of the chunk
        const int hrow = hm * 32 + l31;
        // w2 carries the fp16 scale of dP1 (a power of two: every product below is the unscaled one times sd, bit for bit),
        // so the split needs no multiply of its own; the bias-gradient sum is unscaled once per chunk
        const float w2s = FNO_PB_SDFOLD ? w2v * sd : w2v;
        const f32x2 b1p = {b1v, b1v}, w2p = {w2s, w2s};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float4 dy4 = ld4(douts + n0 + 8 * i + 4 * half);
          const f32x2 dyp[2] = {f32x2{dy4.x, dy4.y}, f32x2{dy4.z, dy4.w}};
          f32x2 dpp[2], pv[2], gv[2], dv[2];
#pragma unroll
          for (int h2 = 0; h2 < 2; ++h2) pv[h2] = f32x2{acc[4 * i + 2 * h2], acc[4 * i + 2 * h2 + 1]} + b1p;
          if constexpr (RELU) {
#pragma unroll
            for (int h2 = 0; h2 < 2; ++h2) {
              dv[h2] = f32x2{pv[h2][0] > 0.f ? 1.f : 0.f, pv[h2][1] > 0.f ? 1.f : 0.f};
              gv[h2] = f32x2{fmaxf(pv[h2][0], 0.f), fmaxf(pv[h2][1], 0.f)};
            }
          } else gelu_both_pairs<2>(pv, gv, dv);          // value and derivative, two pairs in lock step (fno_dev.h)
#pragma unroll
          for (int h2 = 0; h2 < 2; ++h2) {
            dpp[h2] = dv[h2] * (w2p * dyp[h2]);
            sdw = __builtin_elementwise_fma(gv[h2], dyp[h2], sdw);
            sdb = sdb + dpp[h2];
          }
          put_split4_n<NT3>(dr, DTERM, swz_off(hrow, (n0 >> 3) + i) + 8 * half, make_float4(dpp[0][0], dpp[0][1], dpp[1][0], dpp[1][1]), FNO_PB_SDFOLD ? 1.f : sd);
        }
#pragma unroll
        for (int k = 0; k < NCH; ++k)
          if (k == ch) { sdb1[k] += (sdb[0] + sdb[1]) * (FNO_PB_SDFOLD ? inv_sd : 1.f); sdw2[k] += sdw[0] + sdw[1]; }
      }
      if (ch == 1) FNO_STAMP(tslot + 5);
      __syncthreads();         // dr[ch & 1] is complete; every reader of dr[(ch + 1) & 1] (chunk ch - 1) is done
      if (ch == 1) FNO_STAMP(tslot + 6);
      // the next tile's rows: issued in the last chunk, so that their 16 registers are not live through the whole tile
      if (ch == NCH - 1) {
        const int nt2 = tile_ + gridDim.x;
        if (nt2 < a.ntiles) issue_x(nt2);
      }
      // ---- A3: dx^T[px][c] += sum_hid dP1[hid][px] W1[hid][c]: A = transposed reads of the dP1 image ----------------------
      {
#pragma unroll
        for (int kk = 0; kk < NK3; ++kk) {
          bf16x8 af[NT3];
          const int kb = MT == 2 ? kk : 2 * hm + kk;       // C = 32: this wave group's half of the chunk's hidden rows
          const int o0 = swz_off(kb * 16 + trow, tpx >> 3) + 2 * (tpx & 7);
          const int o1 = swz_off(kb * 16 + trow + 4, tpx >> 3) + 2 * (tpx & 7);
#pragma unroll
          for (int t = 0; t < NT3; ++t) af[t] = cat4(lds_tr16(dr + t * DTERM + o0), lds_tr16(dr + t * DTERM + o1));
          mfma_split_s<NT3>(af, wf[kk], dxh, dxl);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      // the next chunk's recompute fragments (the owners' dW1 phase hides their latency; the other group waits for them)
      load_wa1(ch + 1 < NCH ? ch + 1 : 0);      // (last chunk: the next tile's chunk 0)
      if (ch == 1) FNO_STAMP(tslot + 7);
      // ---- B: dW1[hid][c] += sum_px dP1[hid][px] a[c][px] by the group that owns this chunk --------------------------------
      if (hm == (ch & 1)) {
        __builtin_amdgcn_s_setprio(2);
        const int ro = dmt * 32 + l31, rc = dnt * 32 + l31;
#pragma unroll
        for (int k = 0; k < CPW; ++k)
          if (k == (ch >> 1)) {
        f32x16 dacc = dw1acc[k];
#pragma unroll 1
        for (int kq = 0; kq < (MT == 2 ? NPX / 16 : NPX / 32); ++kq) {
          const int chn = 2 * (kq + dkh * (NPX / 32)) + half;
          const int od = swz_off(ro, chn), oa = swz_off(rc, chn);
          bf16x8 af[NT3], bf[NT3];
#pragma unroll
          for (int t = 0; t < NT3; ++t) {
            af[t] = *reinterpret_cast<const bf16x8*>(dr + t * DTERM + od);
            bf[t] = *reinterpret_cast<const bf16x8*>(aimg + t * ATERM + oa);
          }
          dacc = mfma_split<NT3>(af, bf, dacc);
        }
        dw1acc[k] = dacc;
          }
        if (base_prio) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(0);
      }
      if (ch == 1) FNO_STAMP(tslot + 8);
    }
    FNO_STAMP(tslot + 9);
    if constexpr (NT3 == 2) {      // undo the operand scales of the dx product (exact: powers of two)
#pragma unroll
      for (int r = 0; r < 16; ++r) { dxh[r] *= inv_dw; dxl[r] *= inv_dw; }
    }
    // ---- epilogue: x act'(u), gout store (before the barrier: the last chunk's dW1 owners are still on the matrix pipe),
    //      then the gout tile for the row DFT ----------------------------------------------------------------------------
    if constexpr (MT == 2) {
      const int crow = hm * 32 + l31;
      const size_t ro = ((size_t)b * C + crow) * a.PW + px0 + n0 + 4 * half;
      float4 v[4];
#pragma unroll
      for (int i = 0; i < 4; ++i)
        v[i] = make_float4(dxh[4 * i] + dxl[4 * i], dxh[4 * i + 1] + dxl[4 * i + 1], dxh[4 * i + 2] + dxl[4 * i + 2],
                           dxh[4 * i + 3] + dxl[4 * i + 3]);
      if (a.act_in) {
        const __amdgpu_buffer_rsrc_t rs = make_rsrc(a.x + (size_t)b * C * a.PW, (unsigned)(C * a.PW * 4));
        float4 uq[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) uq[i] = buf_ld4(rs, (crow * a.PW + n0 + 4 * half) * 4, (px0 + 8 * i) * 4);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          { float4 uu = uq[i], dd; gelu_both4(uu, dd); v[i].x *= dd.x; v[i].y *= dd.y; v[i].z *= dd.z; v[i].w *= dd.w; }
        }
      }
      // (with a gout tile in LDS for the row DFT the tile leaves in whole lines behind the barrier, below: stored from the
      // accumulator layout - lane <-> channel row, 16 bytes - one instruction touches 32 lines for 32 bytes each)
      if (!(LINE_ST && a.x1g)) {
#pragma unroll
        for (int i = 0; i < 4; ++i) st4(a.gout + ro + 8 * i, v[i]);
      }
      if (a.gmax_out) {
#pragma unroll
        for (int i = 0; i < 4; ++i) gvmax = fmaxf(fmaxf(gvmax, fabsf(v[i].x)), fmaxf(fmaxf(fabsf(v[i].y), fabsf(v[i].z)), fabsf(v[i].w)));
      }
      FNO_STAMP(tslot + 10);
      __syncthreads();       // every wave is done with dP1 buffer 0 (= r3) and with the a image
      if (a.x1g) {
        float* r3p = r3 + crow * PITCH + n0 + 4 * half;
#pragma unroll
        for (int i = 0; i < 4; ++i) st4(r3p + 8 * i, v[i]);
      }
    } else {
      // C = 32: the two wave groups hold partial sums over their hidden halves of the SAME tile.  Wave (hm, nt) finishes the
      // pixel groups i = 2 hm, 2 hm + 1 of its 32 pixels and hands the other two to its partner through dP1 buffer 0 (free
      // since the last chunk's barrier); the gout tile goes to buffer 1 (free behind the barrier below).
      const size_t ro = ((size_t)b * C + l31) * a.PW + px0 + n0 + 4 * half;
      // the exchange buffer [hm][32 rows][128 px] must fit dP1 buffer 0 (the owners of the last chunk still read buffer 1 for
      // their dW1): rows of PITCH floats with three term planes (48 KB), unpadded rows with the 16-byte pieces XOR-swizzled by
      // the row with two (32 KB)
      float* part = reinterpret_cast<float*>(dr0);
      auto poff = [&](int prow, int col) {      // float offset of (row of the [2 x 32] block, pixel), col % 4 == 0
        if constexpr (NT3 == 3) return prow * PITCH + col;
        else return prow * NPX + 4 * ((col >> 2) ^ (prow & 7));
      };
      float* r3b = reinterpret_cast<float*>(dr0 + R3B_OFF);
      {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const int i = 2 * (1 - hm) + j;
#pragma unroll
          for (int ii = 0; ii < 4; ++ii)
            if (ii == i) st4(part + poff(hm * 32 + l31, n0 + 4 * half + 8 * ii),
                             make_float4(dxh[4 * ii] + dxl[4 * ii], dxh[4 * ii + 1] + dxl[4 * ii + 1],
                                         dxh[4 * ii + 2] + dxl[4 * ii + 2], dxh[4 * ii + 3] + dxl[4 * ii + 3]));
        }
      }
      float4 uq[2];
      if (a.act_in) {
        const __amdgpu_buffer_rsrc_t rs = make_rsrc(a.x + (size_t)b * C * a.PW, (unsigned)(C * a.PW * 4));
#pragma unroll
        for (int j = 0; j < 2; ++j) uq[j] = buf_ld4(rs, (l31 * a.PW + n0 + 4 * half) * 4, (px0 + 8 * (2 * hm + j)) * 4);
      }
      FNO_STAMP(tslot + 10);
      __syncthreads();
      float* r3p = r3b + l31 * PITCH + n0 + 4 * half;
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int i = 2 * hm + j;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int ii = 0; ii < 4; ++ii)
          if (ii == i) v = make_float4(dxh[4 * ii] + dxl[4 * ii], dxh[4 * ii + 1] + dxl[4 * ii + 1],
                                       dxh[4 * ii + 2] + dxl[4 * ii + 2], dxh[4 * ii + 3] + dxl[4 * ii + 3]);
        const float4 o = ld4(part + poff((1 - hm) * 32 + l31, n0 + 4 * half + 8 * i));
        v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w;
        if (a.act_in) {
          { float4 uu = uq[j], dd; gelu_both4(uu, dd); v.x *= dd.x; v.y *= dd.y; v.z *= dd.z; v.w *= dd.w; }
        }
        if (!(LINE_ST && a.x1g)) st4(a.gout + ro + 8 * i, v);
        if (a.gmax_out) gvmax = fmaxf(fmaxf(gvmax, fabsf(v.x)), fmaxf(fmaxf(fabsf(v.y), fabsf(v.z)), fabsf(v.w)));
        if (a.x1g) st4(r3p + 8 * i, v);
      }
    }
    if (a.x1g) {
      __syncthreads();
      if constexpr (LINE_ST) {      // gout: row tid / 32 + 16 i of the tile, 16-byte piece tid % 32 - whole 512-byte rows
        // (buffer stores on the x loads' lane offset; the LDS offset from an opaque copy of the thread index: derived per tile,
        // not hoisted into registers that would be live through the chunk loop - the kernel sits at its 256-register cap)
        int t_ = tid;
        asm volatile("" : "+v"(t_));
        const float* r3l = (MT == 2 ? r3 : reinterpret_cast<float*>(dr0 + R3B_OFF)) + (t_ >> 5) * PITCH + 4 * (t_ & 31);
        const __amdgpu_buffer_rsrc_t rg = make_rsrc(a.gout + (size_t)b * C * a.PW, (unsigned)(C * a.PW * 4));
#pragma unroll
        for (int i = 0; i < C / 16; ++i)
          buf_st4(rg, xvoff + (16 * i * a.PW + px0) * 4, 0, ld4(r3l + 16 * i * PITCH));
      }
      row_dft_epilogue<C, NPX, 8>(MT == 2 ? r3 : reinterpret_cast<float*>(dr0 + R3B_OFF), tfwd_s, a.W + 4, a.x1g, b, px0, a.P,
                                  a.W, a.K2out, a.NJ, wave, lane);
      FNO_STAMP(tslot + 11);
      // no barrier here: the gout tile (a dP1 buffer) is rewritten by the next tile's chunks, behind the commit barrier, which
      // a wave only reaches after it has left the row DFT; the commit itself writes the a image, which nobody reads any more
    }
    tslot += 12;
  }

  FNO_CLK_END(3);
  // ---- partial slabs ---------------------------------------------------------------------------------------------------------
  if (a.gmax_out) absmax_publish(gvmax, a.gmax_out);
  if constexpr (NT3 == 2) {
#pragma unroll
    for (int k = 0; k < CPW; ++k)
#pragma unroll
      for (int r = 0; r < 16; ++r) dw1acc[k][r] *= inv_da;
  }
  if constexpr (MT == 1) {      // C = 32: add the two pixel halves of every dW1 tile (waves nt, nt ^ 1 of the owner group)
    __syncthreads();
    float* sc = smem;           // [hm][k][dmt][16][64]
#pragma unroll
    for (int k = 0; k < CPW; ++k)
      if (dkh == 1) {
#pragma unroll
        for (int r = 0; r < 16; ++r) sc[(((hm * CPW + k) * 2 + dmt) * 16 + r) * 64 + lane] = dw1acc[k][r];
      }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < CPW; ++k)
      if (dkh == 0) {
#pragma unroll
        for (int r = 0; r < 16; ++r) dw1acc[k][r] += sc[(((hm * CPW + k) * 2 + dmt) * 16 + r) * 64 + lane];
      }
  }
#pragma unroll
  for (int k = 0; k < CPW; ++k) {
    const int ch = 2 * k + hm;
    float* dst = a.dw1_part + (size_t)blockIdx.x * HID * C;
    if (MT == 2 || dkh == 0) {
#pragma unroll
      for (int r = 0; r < 16; ++r)
        dst[(size_t)(ch * 64 + dmt * 32 + acc_row32(r, half)) * C + dnt * 32 + l31] = dw1acc[k][r];
    }
  }
  {
    const size_t slab = (size_t)blockIdx.x * NTN + nt;
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch) {
      const float vb = sdb1[ch] + __shfl_xor(sdb1[ch], 32, 64);
      const float vw = sdw2[ch] + __shfl_xor(sdw2[ch], 32, 64);
      if (half == 0) {
        const int hid = ch * 64 + hm * 32 + l31;
        a.db1_part[slab * HID + hid] = vb;
        a.dw2_part[slab * HID + hid] = vw;
      }
    }
  }
}
