// Backward of one fused FNO block, second generation ("transposed" layout).  Same mathematics, arguments and partial-slab
// outputs as k_block_bwd_x3 (k_block_bwd.h; reference semantics: autograd of fno_block.py:123-170 +
// spectral_convolution.py:303-347), restructured around three facts measured on gfx950:
//   * the bf16 MFMA's fp32 accumulation is biased (tools/mfma_bias_test.hip: -1e-8 of |result| per element when small
//     and large products share one accumulator), which the DC-type gradients (bias, k = 0 modes) amplify by sqrt(#pixels):
//     the hh products and the five cross terms of the 3-way split now go to SEPARATE accumulators, summed once in fp32;
//   * ds_read_b64_tr_b16 delivers a [channel][pixel] bf16 image column-wise: ONE split of g serves both GEMMs (row reads for
//     dW, transposed reads for dx) - the pixel-major image, its fp32 staging tile, the second split pass and its two
//     barriers are gone;
//   * dx is computed TRANSPOSED (D[pixel][channel]: lane <-> channel, registers <-> 4-pixel runs), which is the layout the
//     tile is loaded in (lane <-> channel row, 16-byte runs of 4 pixels): gelu'(u) stays in registers from the commit to
//     the epilogue (no LDS round trip), the bias gradient is a lane-local sum, gout leaves as 16-byte stores.
// Per tile: commit (GELU, one split per operand -> two swizzled images) | barrier | dW GEMM, dx GEMM + spectral
// K-extension, x gelu', gout store, gout tile -> LDS | barrier | row DFT (or lifting gradients): two barriers instead of five.
#pragma once
#include "fno_dev.h"
#include "k_block_bwd.h"

#ifndef FNO_BBT_LINE_ST
#define FNO_BBT_LINE_ST 1      // k_block_bwd_t: gout in whole lines from the LDS tile (A/B arm 0: 16 bytes per channel row from registers)
#endif
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) s16x4* lds_s16x4_ptr;

// [rows][128 x bf16] image with 256-byte rows: byte offset of 16-byte chunk `ch` (8 pixels) of row `row`.  The XOR makes the
// b128 row reads (lane <-> row), the transposed b64 reads (4 rows x 16 pixels per 16 lanes) and the b64 stores of 16
// consecutive rows spread over the banks.
FNO_DEV int swz_off(int row, int ch) { return 256 * row + 16 * (ch ^ (((row & 3) << 2) | ((row >> 2) & 3))); }

FNO_DEV s16x4 lds_tr16(const unsigned char* p) {
  return __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)p);
}
FNO_DEV bf16x8 cat4(s16x4 lo, s16x4 hi) {
  bf16x8 r;
  r[0] = lo[0]; r[1] = lo[1]; r[2] = lo[2]; r[3] = lo[3];
  r[4] = hi[0]; r[5] = hi[1]; r[6] = hi[2]; r[7] = hi[3];
  return r;
}
// 4 consecutive pixels of one row -> the three bf16 terms, 8 bytes each, at byte offset `off` of the three term planes
FNO_DEV void put_split4(unsigned char* img, int term_bytes, int off, const float4& t) {
  const float tv[4] = {t.x, t.y, t.z, t.w};
  unsigned short hh[4], mm[4], ll[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) split3(tv[j], hh[j], mm[j], ll[j]);
  *reinterpret_cast<uint2*>(img + off) = make_uint2(hh[0] | ((unsigned)hh[1] << 16), hh[2] | ((unsigned)hh[3] << 16));
  *reinterpret_cast<uint2*>(img + term_bytes + off) = make_uint2(mm[0] | ((unsigned)mm[1] << 16), mm[2] | ((unsigned)mm[3] << 16));
  *reinterpret_cast<uint2*>(img + 2 * term_bytes + off) = make_uint2(ll[0] | ((unsigned)ll[1] << 16), ll[2] | ((unsigned)ll[3] << 16));
}

// term-count generic form (fno_dev.h): NTERM = 2 writes the two fp16 terms of scale * t
template <int NTERM>
FNO_DEV void put_split4_n(unsigned char* img, int term_bytes, int off, const float4& t, float scale) {
  if constexpr (NTERM == 3) put_split4(img, term_bytes, off, t);
  else {
    const f32x2 v0 = f32x2{t.x, t.y} * f32x2{scale, scale}, v1 = f32x2{t.z, t.w} * f32x2{scale, scale};
    const f16x2 h0 = __builtin_convertvector(v0, f16x2), h1 = __builtin_convertvector(v1, f16x2);
    const f16x2 l0 = split2_low(v0, h0), l1 = split2_low(v1, h1);      // (one mixed-precision FMA per element: fno_dev.h)
    *reinterpret_cast<uint2*>(img + off) = make_uint2(__builtin_bit_cast(unsigned, h0), __builtin_bit_cast(unsigned, h1));
    *reinterpret_cast<uint2*>(img + term_bytes + off) = make_uint2(__builtin_bit_cast(unsigned, l0), __builtin_bit_cast(unsigned, l1));
  }
}

// kext_loose_rows (fno_dev.h) for the transposed accumulator: the table value rides on the A operand (lane <-> pixel), the
// spectral row on the B operand (lane <-> channel); each lane supplies the same two values as before.
template <int C>
FNO_DEV f32x16 kext_loose_rows_t(f32x16 acc, const float* zs, const float* tinv_s, int K2, int W, int f, int r_lo, int mt,
                                 int l31, int half) {
  const int ra = f / W;
  const int w0 = f - ra * W;
  const int wl = w0 + l31;
  {
    const float* zr = zs + ((size_t)((ra - r_lo) * K2) * C + mt * 32 + l31) * 2 + half;
    const bool in = wl < W;
    const float* tv = tinv_s + half * W + (in ? wl : 0);
#pragma unroll 2
    for (int s = 0; s < K2; ++s) acc = mfma32(in ? tv[2 * s * W] : 0.f, zr[s * C * 2], acc);
  }
  if (w0 + 31 >= W) {
    const float* zr = zs + ((size_t)((ra + 1 - r_lo) * K2) * C + mt * 32 + l31) * 2 + half;
    const bool in = wl >= W;
    const float* tv = tinv_s + half * W + (in ? wl - W : 0);
#pragma unroll 2
    for (int s = 0; s < K2; ++s) acc = mfma32(in ? tv[2 * s * W] : 0.f, zr[s * C * 2], acc);
  }
  return acc;
}

// DROPK: the spectral branch saw drop(x) in the forward pass (rno.py:98): its gradient, the K-extension part of dx, is
// multiplied by the regenerated dropout scale before the skip branch's W^T g is accumulated on top of it.
// NT3: terms of the two channel GEMMs' operands: 3 = bf16 (six products per k block), 2 = fp16 (three; fno_dev.h "h2") with g, a
// and W scaled by powers of two from a.gmax_in, a.umax and the weights' own maximum.
template <int C, int NPX, bool LOOSE = false, bool LIFT = false, bool DROPK = false, int NT3 = 3>
__global__ void __launch_bounds__((C / 32) * (NPX / 32) * 64, 2) k_block_bwd_t(BlkBwdArgs a) {
  static_assert(!(NT3 == 2 && DROPK), "dropout of the spectral branch: three-term kernel only");
  using Cfg = BlkBwdCfg<C, NPX>;
  static_assert(NPX == 128, "images are 128 pixels wide");
  static_assert(!DROPK || (!LOOSE && !LIFT), "dropout of the spectral branch: whole rows, no lifting");
  constexpr int NTN = Cfg::NTN, MT = Cfg::MT, NW = Cfg::NW, TILES = Cfg::TILES, KSPLIT = Cfg::KSPLIT;
  constexpr int NT = NW * 64;
  constexpr int KB = C / 16;
  constexpr int PITCH = NPX + 4;
  constexpr int TERM = C * 256;                         // bytes per term plane of an image
  constexpr int PXK = NPX / KSPLIT;
  static_assert(PXK % 16 == 0, "dW k blocks");
  constexpr int LJ = (C / 16 + NW - 1) / NW;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  unsigned char* gimg = reinterpret_cast<unsigned char*>(smem);       // g,          [NT3][C][128] x 16 bit, swizzled
  unsigned char* aimg = gimg + NT3 * TERM;                            // a = act(u), [NT3][C][128] x 16 bit, swizzled
  float* r3 = reinterpret_cast<float*>(aimg + NT3 * TERM);            // C x PITCH fp32: the gout tile (row DFT / lifting gradients)
  float* xls = r3 + C * PITCH;                                         // 2 x 8 x PITCH: lifting input rows (block 0), by tile parity
  float* tinv_s = xls + (a.xin ? 2 * 8 * PITCH : 0);
  const int R = LOOSE ? NPX / a.W + 2 : NPX / a.W;
  const int KC = (LOOSE && a.kch > 0 && a.kch < a.K2in) ? a.kch : a.K2in;
  const bool chunked = KC < a.K2in;
  float* zs = tinv_s + (a.zg ? 2 * KC * a.W : 0);
  float* tfwd_s = zs + (a.zg ? R * KC * C * 2 : 0);

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, half = lane >> 5;
  const int l15 = lane & 15, quad = lane >> 4;
  const int mt = wave / NTN, nt = wave % NTN;
  const int n0 = nt * 32;
  const int crow = mt * 32 + l31;                       // this lane's channel row (loads, images, dx columns)
  const int dtl = wave % TILES, dkp = wave / TILES;     // dW job
  const int dmt = dtl / MT, dnt = dtl % MT;

  if (a.zg && !chunked)
    for (int i = tid; i < 2 * a.K2in * a.W; i += NT) tinv_s[i] = a.tinv[i];
  if (a.x1g)
    for (int i = tid; i < 16 * a.NJ * a.W; i += NT) tfwd_s[(i / a.W) * (a.W + 4) + i % a.W] = a.tfwd[i];
  auto zc4 = [&](int px0) {
    const int nrows = LOOSE ? (px0 + NPX - 1) / a.W - px0 / a.W + 1 : R;
    return (a.zg && !chunked) ? nrows * a.K2in * C / 2 : 0;
  };

  // operand scales of the two-term fp16 GEMMs (powers of two; 1 with three bf16 terms)
  float sg = 1.f, sa = 1.f, sw = 1.f;
  if constexpr (NT3 == 2) {
    float mw = 0.f;
    for (int i = tid; i < C * C; i += NT) mw = fmaxf(mw, fabsf(a.w[i]));
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) mw = fmaxf(mw, __shfl_xor(mw, o, 64));
    __shared__ float red[NT / 64];
    if (lane == 0) red[tid >> 6] = mw;
    __syncthreads();
    mw = 0.f;
#pragma unroll
    for (int k = 0; k < NT / 64; ++k) mw = fmaxf(mw, red[k]);
    sw = h2_scale(mw); sg = h2_scale(*a.gmax_in); sa = h2_scale(*a.umax);
  }
  const float inv_gw = 1.f / (sg * sw), inv_ga = 1.f / (sg * sa);
  float vmax = 0.f;                                     // max |gout| of this thread (a.gmax_out)
  // B fragments of the dx GEMM: B[k = o][n = i] = W[o][i], lane <-> input channel i, split into terms
  bf16x8 wfrag[KB][NT3];
#pragma unroll
  for (int kb = 0; kb < KB; ++kb) {
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = a.w[(kb * 16 + 8 * half + j) * C + crow];
    split_n_x8<NT3>(v, sw, wfrag[kb]);
  }
  // LIFT: u_0[px][c] = sum_k x[k][px] Wl[c][k] + bl[c] as fp32 MFMAs; B[k][n = c] with k = half + 2 s, the bias rides on k = CL
  constexpr int NKL = 3;
  float wl[NKL];
  if constexpr (LIFT) {
#pragma unroll
    for (int s = 0; s < NKL; ++s) {
      const int k = half + 2 * s;
      wl[s] = k < a.CL ? a.lw[crow * a.CL + k] : (k == a.CL ? a.lb[crow] : 0.f);
    }
  }

  f32x16 dwtot;
#pragma unroll
  for (int r = 0; r < 16; ++r) dwtot[r] = 0.0f;
  float dbsum[4] = {0.f, 0.f, 0.f, 0.f};                // channel grow0 + 8 i, this lane's 4 pixels of every tile
  f32x4 dl[LJ];
#pragma unroll
  for (int j = 0; j < LJ; ++j) dl[j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // the NEXT tile's operands.  u in the dx accumulator's layout (row crow, pixels n0 + 8 i + 4 half .. + 3: gelu' stays in
  // registers); g only feeds the image and the bias sums, so it comes in whole lines (8 lanes per 128-byte row segment)
  float4 gq[4], uq[4];
  const int grow0 = mt * 32 + (lane >> 3);
  float xa[NKL];                // LIFT: x[k = half + 2 s][pixel n0 + l31] of the next tile
  float4 zv = make_float4(0.f, 0.f, 0.f, 0.f);
  auto issue = [&](int tile_) {
    const int tile = a.rev ? a.ntiles - 1 - tile_ : tile_;      // (zigzag along the kernel chain: fno_abi.hip)
    const int b = tile / a.tiles_per_plane;
    const int px0 = (tile % a.tiles_per_plane) * NPX;
    const size_t ro = ((size_t)b * C + crow) * a.PW + px0 + n0 + 4 * half;
#pragma unroll
    for (int i = 0; i < 4; ++i)      // g: whole 128-byte lines per 8 lanes (row grow0 + 8 i, pixels n0 + 4 (lane & 7) ..); read once: streaming
      gq[i] = ld4s(a.g + ((size_t)b * C + grow0 + 8 * i) * a.PW + px0 + n0 + 4 * (lane & 7));
    if constexpr (LIFT) {
#pragma unroll
      for (int s = 0; s < NKL; ++s) {
        const int k = half + 2 * s;
        xa[s] = k < a.CL ? a.xin[((size_t)b * a.CL + k) * a.PW + px0 + n0 + l31] : (k == a.CL ? 1.f : 0.f);
      }
    } else {
#pragma unroll
      // (default cache policy, NOT streaming: in this layout a 128-byte line is touched by four instructions, and a line that
      // the L2 does not keep is fetched again for each of them - 0.36 -> 0.46 ms per launch at FNO3d, round 6)
      for (int i = 0; i < 4; ++i) uq[i] = ld4(a.uin + ro + 8 * i);
    }
    if (tid < zc4(px0)) zv = ld4(a.zg + ((size_t)b * a.P + px0 / a.W) * a.K2in * C * 2 + 4 * tid);
  };
  if ((int)blockIdx.x < a.ntiles) issue(blockIdx.x);

  int tslot = 0, par = 0;
  FNO_TRACE_IF(FNO_TRACE_WHICH == 2 && a.x1g != nullptr);
  for (int tile_ = blockIdx.x; tile_ < a.ntiles; tile_ += gridDim.x, par ^= 1) {
    const int tile = a.rev ? a.ntiles - 1 - tile_ : tile_;
    const int b = tile / a.tiles_per_plane;
    const int px0 = (tile % a.tiles_per_plane) * NPX;
    float* xlt = xls + par * 8 * PITCH;
    FNO_STAMP(tslot + 0);
    // ---- commit: one split per operand into the swizzled images; gelu'(u) stays in registers ----------------------
    float4 dg[4];
    {
      f32x16 u0;
      if constexpr (LIFT) {
#pragma unroll
        for (int r = 0; r < 16; ++r) u0[r] = 0.f;
#pragma unroll
        for (int s = 0; s < NKL; ++s)
          if (2 * s <= a.CL) u0 = mfma32(xa[s], wl[s], u0);
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int off = swz_off(crow, (n0 >> 3) + i) + 8 * half;
        const float4 gv = gq[i];
        dbsum[i] += (gv.x + gv.y) + (gv.z + gv.w);
        put_split4_n<NT3>(gimg, TERM, swz_off(grow0 + 8 * i, (n0 >> 3) + ((lane & 7) >> 1)) + 8 * (lane & 1), gv, sg);
        float4 uv;
        if constexpr (LIFT) uv = make_float4(u0[4 * i], u0[4 * i + 1], u0[4 * i + 2], u0[4 * i + 3]);
        else uv = uq[i];
        if (a.act_in) {
          gelu_both4(uv, dg[i]);          // value and derivative on pairs (fno_dev.h)
        }
        put_split4_n<NT3>(aimg, TERM, off, uv, sa);
      }
    }
    const int zcount4 = zc4(px0);
    if (tid < zcount4) st4(zs + 4 * tid, zv);
    for (int i = tid + NT; i < zcount4; i += NT)      // more spectral rows than threads (short rows, many modes)
      st4(zs + 4 * i, ld4(a.zg + ((size_t)b * a.P + px0 / a.W) * a.K2in * C * 2 + 4 * i));
    if (a.xin) stage_rows<NPX, NT>(xlt, a.xin + (size_t)b * a.CL * a.PW + px0, a.PW, a.CL, a.CL, false, tid);
    FNO_STAMP(tslot + 1);
    __syncthreads();
    FNO_STAMP(tslot + 2);
    if (tile_ + (int)gridDim.x < a.ntiles) issue(tile_ + gridDim.x);

    // ---- dW[o][i] += sum_px g[o][px] a[i][px]: row reads of both images, 8 consecutive pixels per lane ----------------
    // (one accumulator for all six products: a per-element bias of 1e-8 is harmless here, nothing sums dW any further)
    {
      const int ro = dmt * 32 + l31, ri = dnt * 32 + l31;
#pragma unroll 2
      for (int kq = 0; kq < PXK / 16; ++kq) {
        const int ch = dkp * (PXK / 8) + 2 * kq + half;
        const int og = swz_off(ro, ch), oa = swz_off(ri, ch);
        bf16x8 af[NT3], bf[NT3];
#pragma unroll
        for (int t = 0; t < NT3; ++t) {
          af[t] = *reinterpret_cast<const bf16x8*>(gimg + t * TERM + og);
          bf[t] = *reinterpret_cast<const bf16x8*>(aimg + t * TERM + oa);
        }
        dwtot = mfma_split<NT3>(af, bf, dwtot);
      }
    }
    FNO_STAMP(tslot + 3);
    // ---- dx^T[px][i] = sum_o g[o][px] W[o][i]: A = transposed reads of the g image, B = the W fragments -----------------
    f32x16 acc;
    {
      f32x16 hi, lo;
#pragma unroll
      for (int r = 0; r < 16; ++r) { hi[r] = 0.f; lo[r] = 0.f; }
      if constexpr (DROPK) {
        if (a.zg) {
          const float* zr = zs + (((n0 / a.W) * a.K2in) * C + crow) * 2 + half;
          const float* tv = tinv_s + half * a.W + n0 % a.W + l31;
#pragma unroll 2
          for (int s = 0; s < a.K2in; ++s) hi = mfma32(tv[2 * s * a.W], zr[s * C * 2], hi);
          const DropCfg dc = drop_cfg(a.drop_seed, a.drop_p);
          const size_t e0 = ((size_t)b * C + crow) * a.PW + px0 + n0 + 4 * half;      // hi[4 i + j] <-> element e0 + 8 i + j
#pragma unroll
          for (int r = 0; r < 16; ++r) hi[r] *= drop_scale(dc, e0 + 8 * (r >> 2) + (r & 3));
        }
      }
      // lane 4q + p of a 16-lane group supplies row q, pixels 4p .. 4p + 3 of its 4 x 16 block; groups 0 / 1 = pixels
      // 0-15 / 16-31 of k half 0, groups 2 / 3 the same pixels of k half 1
      const int tq = l15 >> 2, tp = l15 & 3;
      const int px = n0 + 16 * (quad & 1) + 4 * tp;
      const int orow = 8 * (quad >> 1) + tq;
#pragma unroll
      for (int kb = 0; kb < KB; ++kb) {      // (full unroll: wfrag must be indexed statically)
        bf16x8 af[NT3];
        const int o0 = swz_off(kb * 16 + orow, px >> 3) + 2 * (px & 7);
        const int o1 = swz_off(kb * 16 + orow + 4, px >> 3) + 2 * (px & 7);
#pragma unroll
        for (int t = 0; t < NT3; ++t) af[t] = cat4(lds_tr16(gimg + t * TERM + o0), lds_tr16(gimg + t * TERM + o1));
        mfma_split_s<NT3>(af, wfrag[kb], hi, lo);
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = NT3 == 2 ? (hi[r] + lo[r]) * inv_gw : hi[r] + lo[r];
    }
    if constexpr (LOOSE) {
      if (a.zg && !chunked) acc = kext_loose_rows_t<C>(acc, zs, tinv_s, a.K2in, a.W, px0 + n0, px0 / a.W, mt, l31, half);
      else if (a.zg) {
        const int r_lo = px0 / a.W, nrows = (px0 + NPX - 1) / a.W - r_lo + 1;
        for (int k0 = 0; k0 < a.K2in; k0 += KC) {
          const int kc = min(KC, a.K2in - k0);
          __syncthreads();                        // every wave is done with the previous chunk (or the previous tile's last)
          for (int i = tid; i < 2 * kc * a.W; i += NT) tinv_s[i] = a.tinv[2 * k0 * a.W + i];
          const int per_row4 = kc * C / 2;
          for (int i = tid; i < nrows * per_row4; i += NT) {
            const int r = i / per_row4, rem = i - r * per_row4;
            st4(zs + 4 * i, ld4(a.zg + (((size_t)b * a.P + r_lo + r) * a.K2in + k0) * C * 2 + 4 * rem));
          }
          __syncthreads();
          acc = kext_loose_rows_t<C>(acc, zs, tinv_s, kc, a.W, px0 + n0, r_lo, mt, l31, half);
        }
      }
    } else if (a.zg && !DROPK) {
      const float* zr = zs + (((n0 / a.W) * a.K2in) * C + crow) * 2 + half;
      const float* tv = tinv_s + half * a.W + n0 % a.W + l31;
#pragma unroll 2
      for (int s = 0; s < a.K2in; ++s) acc = mfma32(tv[2 * s * a.W], zr[s * C * 2], acc);
    }
    FNO_STAMP(tslot + 4);
    // ---- epilogue: (+ gradient addend) x gelu'(u), gout store, gout tile for the row DFT ---------------------------------
    {
      const size_t ro = ((size_t)b * C + crow) * a.PW + px0 + n0 + 4 * half;
      float* r3p = r3 + crow * PITCH + n0 + 4 * half;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        float4 v = make_float4(acc[4 * i], acc[4 * i + 1], acc[4 * i + 2], acc[4 * i + 3]);
        if (a.gadd) {
          const float4 ad = ld4(a.gadd + ro + 8 * i);
          v.x += ad.x; v.y += ad.y; v.z += ad.z; v.w += ad.w;
        }
        if (a.act_in) { v.x *= dg[i].x; v.y *= dg[i].y; v.z *= dg[i].z; v.w *= dg[i].w; }
        // (with a gout tile in LDS the tile leaves in whole lines behind the barrier, below: from this layout - lane <-> channel
        // row, 16 bytes - one store instruction touches 32 lines for 32 bytes each)
        if (a.gout && !(FNO_BBT_LINE_ST && (a.x1g || a.xin))) st4(a.gout + ro + 8 * i, v);
        if (a.gmax_out) vmax = fmaxf(fmaxf(vmax, fabsf(v.x)), fmaxf(fmaxf(fabsf(v.y), fabsf(v.z)), fabsf(v.w)));
        if (a.x1g || a.xin) st4(r3p + 8 * i, v);
      }
    }
    FNO_STAMP(tslot + 5);
    __syncthreads();          // images are free for the next commit; the gout tile is complete
    FNO_STAMP(tslot + 6);
    if (FNO_BBT_LINE_ST && a.gout && (a.x1g || a.xin)) {      // gout: row tid / 32 + (NT / 32) i of the tile, 16-byte piece tid % 32: whole 512-byte rows
      int t_ = tid;
      asm volatile("" : "+v"(t_));         // (opaque: the offsets are derived per tile, not hoisted into registers live through the GEMMs)
      const float* r3l = r3 + (t_ >> 5) * PITCH + 4 * (t_ & 31);
      float* gl = a.gout + ((size_t)b * C + (t_ >> 5)) * a.PW + px0 + 4 * (t_ & 31);
#pragma unroll
      for (int i = 0; i < C * 32 / NT; ++i) st4(gl + (size_t)i * (NT / 32) * a.PW, ld4(r3l + i * (NT / 32) * PITCH));
    }
    if (a.x1g) row_dft_epilogue<C, NPX, NW>(r3, tfwd_s, a.W + 4, a.x1g, b, px0, a.P, a.W, a.K2out, a.NJ, wave, lane);
    if (a.xin) {
      // dl[c][n] += sum_px gout[c][px] * xext[n][px],  xext = [x_in rows | ones | 0..]
#pragma unroll
      for (int j = 0; j < LJ; ++j) {
        const int jm = wave + j * NW;
        if (jm < C / 16) {
          const float* arow = r3 + (jm * 16 + l15) * PITCH + quad;
          const float* br = xlt + (l15 < a.CL ? l15 : 0) * PITCH + quad;
          const float cst = l15 == a.CL ? 1.0f : 0.0f;
          for (int s = 0; s < NPX / 4; ++s) {
            const float bf = (l15 < a.CL) ? br[4 * s] : cst;
            dl[j] = mfma16(arow[4 * s], bf, dl[j]);
          }
        }
      }
    }
    FNO_STAMP(tslot + 7);
    // no barrier here: the next commit writes the images, zs and the other xls buffer, none of which this phase reads;
    // r3 is rewritten only behind the next tile's first barrier
    tslot += 8;
  }

  // ---- write partial slabs ---------------------------------------------------
  if (a.gmax_out) absmax_publish(vmax, a.gmax_out);
  {
    float* dst = a.dw_part + ((size_t)blockIdx.x * KSPLIT + dkp) * C * C;
#pragma unroll
    for (int r = 0; r < 16; ++r) dst[(dmt * 32 + acc_row32(r, half)) * C + dnt * 32 + l31] = NT3 == 2 ? dwtot[r] * inv_ga : dwtot[r];
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 4; ++i) smem[(nt * 8 + (lane & 7)) * C + grow0 + 8 * i] = dbsum[i];
  __syncthreads();
  if (tid < C) {
    float v = 0.f;
#pragma unroll
    for (int k = 0; k < 8 * NTN; ++k) v += smem[k * C + tid];
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
// Two-group variant (C = 64, rows of 32 / 64 / 128 pixels).  The 8 waves of a workgroup form TWO independent groups of four
// (waves 0-3 and 4-7: one wave of each group per SIMD).  Each group walks its own 128-pixel tiles in two 64-pixel halves
// with its own images and its own barrier (a counter in LDS: s_barrier would tie the groups together), so that the groups
// drift apart and one group's commit (VALU) runs beside the other's GEMMs (matrix pipe) on every SIMD - what two
// workgroups per CU would give, at one copy of the tables and within the 160 KB of LDS.
//   LDS: per group  g image | a image ([3][64][64] bf16 each, 128-byte rows, swizzled) | fp32 gout half-tile 64 x 68 |
//        spectral rows | 2 x lifting-input rows;  shared: inverse / forward row tables, two barrier counters.
// ---------------------------------------------------------------------------
// [rows][64 x bf16] image with 128-byte rows: byte offset of 16-byte chunk `ch` (0..7) of row `row`.  Two rows share a
// 256-byte bank row; the XOR keeps b128 row reads (16-lane groups), transposed b64 reads (4 rows x 16 pixels) conflict-free
// and b64 stores of 16 consecutive rows 2-way.
FNO_DEV int swz64_off(int row, int ch) { return 128 * row + 16 * (ch ^ ((((row >> 1) & 1) << 2) | ((row >> 2) & 3))); }

// barrier of one 4-wave group: arrive on an LDS counter, poll until all four have (epoch counts arrivals so far)
FNO_DEV void group_barrier(unsigned* cnt, unsigned& epoch, int lane) {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // this wave's LDS reads / writes are done
  epoch += 4;
  if (lane == 0) __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  while (__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < epoch) __builtin_amdgcn_s_sleep(1);
  asm volatile("" ::: "memory");
}

// LIFT: block 0 of a model with a lifting layer (u_0 recomputed from the model input, lifting gradients instead of a row DFT);
// GADD: a gradient addend is added to dx (fan-out chains); NJP: 16-output blocks of the row DFT per wave when a row spans both halves
// LINES: u is loaded and gout stored in whole lines (a.lines; the host adds the staging to the LDS size)
template <bool LIFT = false, bool GADD = false, int NJP = 1, int NT3 = 3, bool LINES = false>
__global__ void __launch_bounds__(512, 2) k_block_bwd_g2(BlkBwdArgs a) {
  FNO_CLK_ENTRY();
  constexpr int C = 64, GPX = 64, KB = C / 16, PITCH = GPX + 4, XPITCH = GPX + 4;
  constexpr int TERM = C * 128;                  // bytes per term plane
  constexpr int IMG = NT3 * TERM;                // one image
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);     // wave-uniform by construction: keep tile / address math scalar
  const int grp = wave >> 2, wg = wave & 3, gtid = tid & 255;
  const int l31 = lane & 31, half = lane >> 5;
  const int l15 = lane & 15, quad = lane >> 4;
  const int mt = wg >> 1, nt = wg & 1;
  const int n0 = nt * 32;
  const int crow = mt * 32 + l31;
  const int R2 = 128 / a.W;                      // rows per 128-pixel tile (1, 2 or 4)
  const int zrow_f = a.K2in * C * 2;             // floats per spectral row
  // ---- LDS carve ----
  unsigned char* base = reinterpret_cast<unsigned char*>(smem);
  unsigned char* gimg = base + grp * 2 * IMG;
  unsigned char* aimg = gimg + IMG;
  float* r3 = reinterpret_cast<float*>(base + 4 * IMG) + grp * C * PITCH;
  float* after = reinterpret_cast<float*>(base + 4 * IMG) + 2 * C * PITCH;
  // spectral K-extension operands: fp32 rows + table (2-deep fp32 MFMAs), or - kx - bf16x3 images with k = 2 s + (re, im)
  // padded to ONE 16-deep block: zimg [3][row][channel][16], timg [3][pixel of the row][16] (k_block_fwd2.h has the same pair)
  const bool kx = a.kx16 != 0 && a.zg != nullptr;
  const int ZT = R2 * C * 32, TT = a.W * 32;     // bytes per term plane of the two images
  float* zs = after + (a.zg && !kx ? grp * R2 * zrow_f : 0);
  after += a.zg && !kx ? 2 * R2 * zrow_f : 0;
  unsigned char* zimg = reinterpret_cast<unsigned char*>(after) + (kx ? grp * 3 * ZT : 0);
  after += kx ? 2 * 3 * ZT / 4 : 0;
  float* xls = after + (a.xin ? grp * 2 * 8 * XPITCH : 0);
  after += a.xin ? 2 * 2 * 8 * XPITCH : 0;
  float* tinv_s = after;
  after += a.zg && !kx ? 2 * a.K2in * a.W : 0;
  unsigned char* timg = reinterpret_cast<unsigned char*>(after);
  after += kx ? 3 * TT / 4 : 0;
  float* tfwd_s = after;
  after += a.x1g ? 16 * a.NJ * (a.W + 4) : 0;
  unsigned* bar = reinterpret_cast<unsigned*>(after) + grp;
  // a.lines: 2 KB per wave behind the counters (16-byte aligned): gelu'(u), computed where u arrives in whole lines (lane <-> row
  // 8 i + lane / 8, 16-byte chunk lane % 8), takes a turn through it into the accumulators' layout (lane <-> row, registers <->
  // 4-pixel runs) for the epilogue: 16 rows x 128 bytes at a time, chunks XOR-swizzled by the row, private to the wave
  unsigned char* dgst = reinterpret_cast<unsigned char*>(after + 4) + wave * 2048;

  // Every global value the prologue needs is REQUESTED first (weights, bounds, both tables), then the images are cleared and
  // the values used: the prologue was a chain of dependent L2 round trips behind barriers (9.2 us of a 170-200 us launch,
  // tools/kernel_clock.py), now it is one.
  constexpr int NTI = 4;
  float wraw[KB][8];                 // B fragments of the dx GEMM: B[k = o][n = i] = W[o][i], lane <-> input channel i
#pragma unroll
  for (int kb = 0; kb < KB; ++kb)
#pragma unroll
    for (int j = 0; j < 8; ++j) wraw[kb][j] = a.w[(kb * 16 + 8 * half + j) * C + crow];
  float bg = 0.f, bu = 0.f;
  if constexpr (NT3 == 2) { bg = *a.gmax_in; bu = *a.umax; }
  const int nti = a.zg ? 2 * a.K2in * a.W : 0, ntf = a.x1g ? 16 * a.NJ * a.W : 0;
  float tiv[NTI], tfv[NTI];
#pragma unroll
  for (int k = 0; k < NTI; ++k) {
    tiv[k] = tid + k * 512 < nti ? a.tinv[tid + k * 512] : 0.f;
    tfv[k] = tid + k * 512 < ntf ? a.tfwd[tid + k * 512] : 0.f;
  }
  constexpr int NKL = 3;
  float4 gq[4], uq[4];
  const int grow0 = mt * 32 + (lane >> 3);
  float xa[NKL];
  float4 zv = make_float4(0.f, 0.f, 0.f, 0.f);
  const int zc4 = a.zg ? R2 * a.K2in * C / 2 : 0;     // float4s of spectral rows per 128-pixel tile
  auto issue = [&](int tile_, int h) {
    const int tile = a.rev ? a.ntiles - 1 - tile_ : tile_;
    const int b = tile / a.tiles_per_plane;
    const int px0 = (tile % a.tiles_per_plane) * 128 + 64 * h;
#pragma unroll
    for (int i = 0; i < 4; ++i)
      gq[i] = ld4s(a.g + ((size_t)b * C + grow0 + 8 * i) * a.PW + px0 + n0 + 4 * (lane & 7));
    if constexpr (LIFT) {
#pragma unroll
      for (int s = 0; s < NKL; ++s) {
        const int k = half + 2 * s;
        xa[s] = k < a.CL ? a.xin[((size_t)b * a.CL + k) * a.PW + px0 + n0 + l31] : (k == a.CL ? 1.f : 0.f);
      }
    } else {
      // (as loaded by rounds 2-5 - lane <-> channel row, 16 bytes - one instruction touches 32 lines for 32 bytes each; in
      // whole lines like g the block backward takes 0.181 instead of 0.205 ms per launch at config 2 and, unlike every
      // schedule change of rounds 3-5, the step gets ALL of it: fewer requests are less energy, not just less waiting)
      if constexpr (LINES) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
          uq[i] = ld4s(a.uin + ((size_t)b * C + grow0 + 8 * i) * a.PW + px0 + n0 + 4 * (lane & 7));
      } else {
        const size_t ro = ((size_t)b * C + crow) * a.PW + px0 + n0 + 4 * half;
#pragma unroll
        for (int i = 0; i < 4; ++i) uq[i] = ld4(a.uin + ro + 8 * i);
      }
    }
    if (h == 0 && gtid < zc4) zv = ld4(a.zg + ((size_t)b * a.P + px0 / a.W) * zrow_f + 4 * gtid);
  };
  const int tile0 = blockIdx.x * 2 + grp, tstep = 2 * gridDim.x;
  if (tile0 < a.ntiles) issue(tile0, 0);      // the first tile travels while the tables and fragments are set up
  auto put_tinv = [&](int i, float v) {
    const int k = i / a.W, w = i - k * a.W;
    unsigned short h, m, l;
    split3(v, h, m, l);
    unsigned short* d = reinterpret_cast<unsigned short*>(timg) + w * 16 + k;
    d[0] = h; d[TT / 2] = m; d[TT] = l;
  };
  if (kx) {
    unsigned* zi = reinterpret_cast<unsigned*>(zimg - grp * 3 * ZT);      // both groups' images: the k pads stay zero
    for (int i = tid; i < 2 * 3 * ZT / 4; i += 512) zi[i] = 0u;
    for (int i = tid; i < 3 * TT / 4; i += 512) reinterpret_cast<unsigned*>(timg)[i] = 0u;
    __syncthreads();
#pragma unroll
    for (int k = 0; k < NTI; ++k)
      if (tid + k * 512 < nti) put_tinv(tid + k * 512, tiv[k]);
    for (int i = tid + NTI * 512; i < nti; i += 512) put_tinv(i, a.tinv[i]);
  } else if (a.zg) {
#pragma unroll
    for (int k = 0; k < NTI; ++k)
      if (tid + k * 512 < nti) tinv_s[tid + k * 512] = tiv[k];
    for (int i = tid + NTI * 512; i < nti; i += 512) tinv_s[i] = a.tinv[i];
  }
  if (a.x1g) {
#pragma unroll
    for (int k = 0; k < NTI; ++k) {
      const int i = tid + k * 512;
      if (i < ntf) tfwd_s[(i / a.W) * (a.W + 4) + i % a.W] = tfv[k];
    }
    for (int i = tid + NTI * 512; i < ntf; i += 512) tfwd_s[(i / a.W) * (a.W + 4) + i % a.W] = a.tfwd[i];
  }
  if (tid < 2) reinterpret_cast<unsigned*>(after)[tid] = 0u;
  // operand scales of the two-term fp16 GEMMs (powers of two; 1 with three bf16 terms)
  float sg = 1.f, sa = 1.f, sw = 1.f;
  __shared__ float red[8];
  if constexpr (NT3 == 2) {
    // max |W|: the fragments of the workgroup's waves cover every element of W (column crow, all rows over the two halves)
    float mw = 0.f;
#pragma unroll
    for (int kb = 0; kb < KB; ++kb)
#pragma unroll
      for (int j = 0; j < 8; ++j) mw = fmaxf(mw, fabsf(wraw[kb][j]));
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) mw = fmaxf(mw, __shfl_xor(mw, o, 64));
    if (lane == 0) red[tid >> 6] = mw;
  }
  __syncthreads();
  unsigned epoch = 0;
  if constexpr (NT3 == 2) {
    float mw = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) mw = fmaxf(mw, red[k]);
    sw = h2_scale(mw); sg = h2_scale(bg); sa = h2_scale(bu);
  }
  const float inv_gw = 1.f / (sg * sw), inv_ga = 1.f / (sg * sa);
  float vmax = 0.f;                                     // max |gout| of this thread (a.gmax_out)
  bf16x8 wfrag[KB][NT3];
#pragma unroll
  for (int kb = 0; kb < KB; ++kb) split_n_x8<NT3>(wraw[kb], sw, wfrag[kb]);
  float wl[NKL];
  if constexpr (LIFT) {
#pragma unroll
    for (int s = 0; s < NKL; ++s) {
      const int k = half + 2 * s;
      wl[s] = k < a.CL ? a.lw[crow * a.CL + k] : (k == a.CL ? a.lb[crow] : 0.f);
    }
  }

  f32x16 dwtot;
#pragma unroll
  for (int r = 0; r < 16; ++r) dwtot[r] = 0.0f;
  float dbsum[4] = {0.f, 0.f, 0.f, 0.f};
  f32x4 dl = {0.f, 0.f, 0.f, 0.f};                   // lifting gradients: job wg (16 channels)
  f32x4 dft0[NJP], dft1[NJP];                         // W = 128: row-DFT accumulators of jobs wg (, wg + 4), carried over the halves

#ifndef FNO_G2_STAGGER
#define FNO_G2_STAGGER 55
#endif
  // identical programs started together stay in lockstep (both groups in their VALU phase, then both on the matrix pipe);
  // starting group 1 about half a period late puts its commits beside group 0's GEMMs
  if (FNO_G2_STAGGER > 0 && grp == 1) __builtin_amdgcn_s_sleep(FNO_G2_STAGGER);

  int par = 0;
  FNO_TRACE_IF(FNO_TRACE_WHICH == 2 && a.x1g != nullptr);
  int tslot = 0;
  FNO_CLK_BEGIN();
  for (int tile_ = tile0; tile_ < a.ntiles; tile_ += tstep) {
    const int tile = a.rev ? a.ntiles - 1 - tile_ : tile_;
    const int b = tile / a.tiles_per_plane;
    const int pxt = (tile % a.tiles_per_plane) * 128;
#pragma unroll 1
    for (int h = 0; h < 2; ++h, par ^= 1) {
      const int px0 = pxt + 64 * h;
      float* xlt = xls + par * 8 * XPITCH;
      FNO_STAMP(tslot + 0);
      // ---- commit ---------------------------------------------------------------------------------------------------
      float4 dg[4];
      {
        f32x16 u0;
        if constexpr (LIFT) {
#pragma unroll
          for (int r = 0; r < 16; ++r) u0[r] = 0.f;
#pragma unroll
          for (int s = 0; s < NKL; ++s)
            if (2 * s <= a.CL) u0 = mfma32(xa[s], wl[s], u0);
        }
        constexpr bool ULINES = !LIFT && LINES;      // u in g's layout (whole lines)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float4 gv = gq[i];
          dbsum[i] += (gv.x + gv.y) + (gv.z + gv.w);
          put_split4_n<NT3>(gimg, TERM, swz64_off(grow0 + 8 * i, (n0 >> 3) + ((lane & 7) >> 1)) + 8 * (lane & 1), gv, sg);
          float4 uv;
          if constexpr (LIFT) uv = make_float4(u0[4 * i], u0[4 * i + 1], u0[4 * i + 2], u0[4 * i + 3]);
          else uv = uq[i];
          if constexpr (ULINES) {
            // the derivative takes a turn through the wave's staging rows into the accumulator layout: rows 16 p .. 16 p + 15
            // of the wave's 32 (i = 2 p, 2 p + 1) per pass, read back by the half of the lanes that own those rows
            float4 dw_ = make_float4(1.f, 1.f, 1.f, 1.f);
            if (a.act_in) gelu_both4(uv, dw_);
            put_split4_n<NT3>(aimg, TERM, swz64_off(grow0 + 8 * i, (n0 >> 3) + ((lane & 7) >> 1)) + 8 * (lane & 1), uv, sa);
            if (a.act_in) {
              const int r = 8 * (i & 1) + (lane >> 3);
              st4(reinterpret_cast<float*>(dgst + r * 128 + (((lane & 7) ^ (r & 7)) * 16)), dw_);
              if ((i & 1) && (l31 >> 4) == (i >> 1)) {
                const int rr = l31 & 15;
#pragma unroll
                for (int k = 0; k < 4; ++k) dg[k] = ld4(reinterpret_cast<const float*>(dgst + rr * 128 + (((2 * k + half) ^ (rr & 7)) * 16)));
              }
            }
          } else {
            if (a.act_in) {
              gelu_both4(uv, dg[i]);          // value and derivative on pairs (fno_dev.h)
            }
            put_split4_n<NT3>(aimg, TERM, swz64_off(crow, (n0 >> 3) + i) + 8 * half, uv, sa);
          }
        }
      }
      if (h == 0) {
        if (kx) {
          // float4 piece i of the tile's spectral rows = (re, im) of channels c0, c0 + 1 of one (row, mode s): two bf16 pairs
          // per term at k = 2 s, 2 s + 1 of those channels
          auto put_z = [&](int i, const float4& v) {
            const int e = 2 * i, c0 = e % C, rs = e / C, sm = rs % a.K2in, row = rs / a.K2in;
            unsigned char* d = zimg + ((row * C + c0) * 16 + 2 * sm) * 2;
            unsigned short hh[4], mm[4], ll[4];
            split3(v.x, hh[0], mm[0], ll[0]); split3(v.y, hh[1], mm[1], ll[1]);
            split3(v.z, hh[2], mm[2], ll[2]); split3(v.w, hh[3], mm[3], ll[3]);
            *reinterpret_cast<unsigned*>(d) = hh[0] | ((unsigned)hh[1] << 16);
            *reinterpret_cast<unsigned*>(d + 32) = hh[2] | ((unsigned)hh[3] << 16);
            *reinterpret_cast<unsigned*>(d + ZT) = mm[0] | ((unsigned)mm[1] << 16);
            *reinterpret_cast<unsigned*>(d + ZT + 32) = mm[2] | ((unsigned)mm[3] << 16);
            *reinterpret_cast<unsigned*>(d + 2 * ZT) = ll[0] | ((unsigned)ll[1] << 16);
            *reinterpret_cast<unsigned*>(d + 2 * ZT + 32) = ll[2] | ((unsigned)ll[3] << 16);
          };
          if (gtid < zc4) put_z(gtid, zv);
          for (int i = gtid + 256; i < zc4; i += 256) put_z(i, ld4(a.zg + ((size_t)b * a.P + pxt / a.W) * zrow_f + 4 * i));
        } else {
          if (gtid < zc4) st4(zs + 4 * gtid, zv);
          for (int i = gtid + 256; i < zc4; i += 256)
            st4(zs + 4 * i, ld4(a.zg + ((size_t)b * a.P + pxt / a.W) * zrow_f + 4 * i));
        }
      }
      if constexpr (LIFT) {      // lifting input rows of this half: [k][64 px]
        for (int idx = gtid; idx < a.CL * (GPX / 4); idx += 256) {
          const int k = idx / (GPX / 4), q = idx % (GPX / 4);
          st4(xlt + k * XPITCH + 4 * q, ld4(a.xin + ((size_t)b * a.CL + k) * a.PW + px0 + 4 * q));
        }
      }
      FNO_STAMP(tslot + 1);
      group_barrier(bar, epoch, lane);
      FNO_STAMP(tslot + 2);
#ifndef FNO_G2_PRIO
#define FNO_G2_PRIO 1
#endif
      // the matrix-pipe phase gets issue priority over the other group's VALU phase on this SIMD: its MFMAs and LDS reads are
      // latency-bound, the VALU stream fills the gaps (tools/simd_share_test.hip: VALU beside bf16 MFMA runs at 87 %)
      if (FNO_G2_PRIO) __builtin_amdgcn_s_setprio(2);
      // ---- dW[o][i] += sum_px g[o][px] a[i][px]: wave wg owns one 32 x 32 tile over the 64 pixels of the half -------------
      {
        const int ro = mt * 32 + l31, ri = nt * 32 + l31;
#pragma unroll 2
        for (int kq = 0; kq < GPX / 16; ++kq) {
          const int ch = 2 * kq + half;
          const int og = swz64_off(ro, ch), oa = swz64_off(ri, ch);
          bf16x8 af[NT3], bf[NT3];
#pragma unroll
          for (int t = 0; t < NT3; ++t) {
            af[t] = *reinterpret_cast<const bf16x8*>(gimg + t * TERM + og);
            bf[t] = *reinterpret_cast<const bf16x8*>(aimg + t * TERM + oa);
          }
          dwtot = mfma_split<NT3>(af, bf, dwtot);
        }
      }
      FNO_STAMP(tslot + 3);
      // ---- dx^T[px][i] = sum_o g[o][px] W[o][i] ---------------------------------------------------------------------------
      f32x16 acc;
      {
        f32x16 hi, lo;
#pragma unroll
        for (int r = 0; r < 16; ++r) { hi[r] = 0.f; lo[r] = 0.f; }
        const int tq = l15 >> 2, tp = l15 & 3;
        const int px = n0 + 16 * (quad & 1) + 4 * tp;
        const int orow = 8 * (quad >> 1) + tq;
#pragma unroll
        for (int kb = 0; kb < KB; ++kb) {
          bf16x8 af[NT3];
          const int o0 = swz64_off(kb * 16 + orow, px >> 3) + 2 * (px & 7);
          const int o1 = swz64_off(kb * 16 + orow + 4, px >> 3) + 2 * (px & 7);
#pragma unroll
          for (int t = 0; t < NT3; ++t) af[t] = cat4(lds_tr16(gimg + t * TERM + o0), lds_tr16(gimg + t * TERM + o1));
          mfma_split_s<NT3>(af, wfrag[kb], hi, lo);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = NT3 == 2 ? (hi[r] + lo[r]) * inv_gw : hi[r] + lo[r];
      }
      if (kx) {        // spectral K-extension as one 16-deep bf16x3 block: D^T[px][c] += T^T[px][k] Z[c][k] (six products)
        const int pl = 64 * h + n0;                    // position of this wave's 32 pixels in the 128-pixel tile
        const unsigned char* ta = timg + ((pl % a.W + l31) * 16 + 8 * half) * 2;
        const unsigned char* zb = zimg + (((pl / a.W) * C + crow) * 16 + 8 * half) * 2;
        bf16x8 fa3[3], fb3[3];
#pragma unroll
        for (int t = 0; t < 3; ++t) {
          fa3[t] = *reinterpret_cast<const bf16x8*>(ta + t * TT);
          fb3[t] = *reinterpret_cast<const bf16x8*>(zb + t * ZT);
        }
        f32x16 eh, el;
#pragma unroll
        for (int r = 0; r < 16; ++r) { eh[r] = 0.f; el[r] = 0.f; }
        mfma_split_s<3>(fa3, fb3, eh, el);
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] += eh[r] + el[r];
      } else if (a.zg) {      // spectral K-extension: all table / spectrum values of a group of 4 modes are in flight together
        const int pl = 64 * h + n0;                    // position of this wave's 32 pixels in the 128-pixel tile
        const float* zr = zs + ((pl / a.W) * a.K2in * C + crow) * 2 + half;
        const float* tv = tinv_s + half * a.W + pl % a.W + l31;
        for (int s0 = 0; s0 < a.K2in; s0 += 4) {
          float zq[4], tq4[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const bool in = s0 + j < a.K2in;
            zq[j] = in ? zr[(s0 + j) * C * 2] : 0.f;
            tq4[j] = in ? tv[2 * (s0 + j) * a.W] : 0.f;
          }
#pragma unroll
          for (int j = 0; j < 4; ++j)
            if (s0 + j < a.K2in) acc = mfma32(tq4[j], zq[j], acc);
        }
      }
      if (FNO_G2_PRIO) __builtin_amdgcn_s_setprio(0);
      FNO_STAMP(tslot + 4);
      // the next half's operands: issued behind the GEMMs (32 registers that must not be live beside the weight fragments and
      // the accumulators); their latency is covered by the epilogue, the row DFT and the other group's work on this SIMD
      {
        const int nh = h ^ 1, ntile = h ? tile_ + tstep : tile_;
        if (ntile < a.ntiles) issue(ntile, nh);
      }
      // ---- epilogue -------------------------------------------------------------------------------------------------------
      {
        const size_t ro = ((size_t)b * C + crow) * a.PW + px0 + n0 + 4 * half;
        float* r3p = r3 + crow * PITCH + n0 + 4 * half;
        float4 ad[4];
        if constexpr (GADD) {
#pragma unroll
          for (int i = 0; i < 4; ++i) ad[i] = ld4(a.gadd + ro + 8 * i);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          float4 v = make_float4(acc[4 * i], acc[4 * i + 1], acc[4 * i + 2], acc[4 * i + 3]);
          if constexpr (GADD) { v.x += ad[i].x; v.y += ad[i].y; v.z += ad[i].z; v.w += ad[i].w; }
          if (a.act_in) { v.x *= dg[i].x; v.y *= dg[i].y; v.z *= dg[i].z; v.w *= dg[i].w; }
          // (a.lines with a gout tile in LDS: the tile leaves in whole lines behind the barrier, below)
          if (a.gout && !(LINES && (LIFT || a.x1g))) st4(a.gout + ro + 8 * i, v);
          if (a.gmax_out) vmax = fmaxf(fmaxf(vmax, fabsf(v.x)), fmaxf(fmaxf(fabsf(v.y), fabsf(v.z)), fabsf(v.w)));
          if (LIFT || a.x1g) st4(r3p + 8 * i, v);
        }
      }
      FNO_STAMP(tslot + 5);
      group_barrier(bar, epoch, lane);
      FNO_STAMP(tslot + 6);
      if (LINES && a.gout && (LIFT || a.x1g)) {
        // gout from the fp32 tile in whole 256-byte lines: wave wg stores rows 16 wg .. 16 wg + 15, four rows per instruction
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int row = 16 * wg + 4 * i + (lane >> 4);
          st4(a.gout + ((size_t)b * C + row) * a.PW + px0 + 4 * (lane & 15), ld4(r3 + row * PITCH + 4 * (lane & 15)));
        }
      }
      // ---- row DFT of gout (truncated, fp32 MFMA 16x16x4) -----------------------------------------------------------------
      if (!LIFT && a.x1g) {
        if (a.W == 128) {
          // a row spans both halves: jobs (16-channel block, 16-output block) = wg, wg + 4; this half adds 4 of the 8 k steps
#pragma unroll
          for (int jj = 0; jj < NJP; ++jj) {
            const int job = wg + 4 * jj;
            if (job < 4 * a.NJ) {
              const int n16 = job & 3, jt = job >> 2;
              if (h == 0) { dft0[jj] = f32x4{0.f, 0.f, 0.f, 0.f}; dft1[jj] = f32x4{0.f, 0.f, 0.f, 0.f}; }
              const float* tf = tfwd_s + (size_t)(jt * 16 + l15) * (a.W + 4) + 4 * quad + 64 * h;
              const float* xr = r3 + (n16 * 16 + l15) * PITCH + 4 * quad;
              float4 av[4], bv[4];        // all four k steps' operands in flight before the first MFMA (fno_dev.h: row_dft_epilogue)
#pragma unroll
              for (int q = 0; q < 4; ++q) { av[q] = ld4(tf + 16 * q); bv[q] = ld4(xr + 16 * q); }
#ifndef FNO_DFT_NOPIPE
              __builtin_amdgcn_sched_barrier(0);
#endif
#pragma unroll
              for (int q = 0; q < 4; ++q) {
                dft0[jj] = mfma16(av[q].x, bv[q].x, dft0[jj]);
                dft1[jj] = mfma16(av[q].y, bv[q].y, dft1[jj]);
                dft0[jj] = mfma16(av[q].z, bv[q].z, dft0[jj]);
                dft1[jj] = mfma16(av[q].w, bv[q].w, dft1[jj]);
              }
              if (h == 1) {
                const int prow = pxt / a.W;
                const int c = n16 * 16 + l15;
#pragma unroll
                for (int pr = 0; pr < 2; ++pr) {
                  const int k2 = jt * 8 + quad * 2 + pr;
                  if (k2 < a.K2out)
                    *reinterpret_cast<float2*>(a.x1g + ((((size_t)b * a.P + prow) * a.K2out + k2) * C + c) * 2) =
                        make_float2(dft0[jj][2 * pr] + dft1[jj][2 * pr], dft0[jj][2 * pr + 1] + dft1[jj][2 * pr + 1]);
                }
              }
            }
          }
        } else {
          // rows of 32 / 64 pixels lie inside the half: complete jobs (channel block, row, output block)
          const int RH = GPX / a.W;
          const int njobs = 4 * RH * a.NJ;
          for (int job = wg; job < njobs; job += 4) {
            const int n16 = job & 3, rr = (job >> 2) % RH, jt = (job >> 2) / RH;
            f32x4 d0 = {0.f, 0.f, 0.f, 0.f}, d1 = {0.f, 0.f, 0.f, 0.f};
            const float* tf = tfwd_s + (size_t)(jt * 16 + l15) * (a.W + 4) + 4 * quad;
            const float* xr = r3 + (n16 * 16 + l15) * PITCH + rr * a.W + 4 * quad;
            for (int q = 0; q < a.W / 16; ++q) {
              const float4 av = ld4(tf + 16 * q);
              const float4 bv = ld4(xr + 16 * q);
              d0 = mfma16(av.x, bv.x, d0);
              d1 = mfma16(av.y, bv.y, d1);
              d0 = mfma16(av.z, bv.z, d0);
              d1 = mfma16(av.w, bv.w, d1);
            }
            const int prow = px0 / a.W + rr;
            const int c = n16 * 16 + l15;
#pragma unroll
            for (int pr = 0; pr < 2; ++pr) {
              const int k2 = jt * 8 + quad * 2 + pr;
              if (k2 < a.K2out)
                *reinterpret_cast<float2*>(a.x1g + ((((size_t)b * a.P + prow) * a.K2out + k2) * C + c) * 2) =
                    make_float2(d0[2 * pr] + d1[2 * pr], d0[2 * pr + 1] + d1[2 * pr + 1]);
            }
          }
        }
      }
      if constexpr (LIFT) {      // lifting gradients: job wg = 16 channels, dl[c][n] += sum_px gout[c][px] xext[n][px]
        const float* arow = r3 + (wg * 16 + l15) * PITCH + quad;
        const float* br = xlt + (l15 < a.CL ? l15 : 0) * XPITCH + quad;
        const float cst = l15 == a.CL ? 1.0f : 0.0f;
#pragma unroll 4
        for (int s = 0; s < GPX / 4; ++s) {
          const float bf = (l15 < a.CL) ? br[4 * s] : cst;
          dl = mfma16(arow[4 * s], bf, dl);
        }
      }
      FNO_STAMP(tslot + 7);
      tslot += 8;
    }
  }

  FNO_CLK_END(1);
  // ---- partial slabs: one dW slab per group; bias and lifting gradients summed over both groups ---------------------------
  if (a.gmax_out) absmax_publish(vmax, a.gmax_out);
  {
    float* dst = a.dw_part + ((size_t)blockIdx.x * 2 + grp) * C * C;
#pragma unroll
    for (int r = 0; r < 16; ++r) dst[(mt * 32 + acc_row32(r, half)) * C + nt * 32 + l31] = NT3 == 2 ? dwtot[r] * inv_ga : dwtot[r];
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 4; ++i) smem[((grp * 2 + nt) * 8 + (lane & 7)) * C + grow0 + 8 * i] = dbsum[i];
  float* dls = smem + 32 * C;                      // [grp][64 channels][16]
  if constexpr (LIFT) {
#pragma unroll
    for (int r = 0; r < 4; ++r) dls[(grp * C + wg * 16 + quad * 4 + r) * 16 + l15] = dl[r];
  }
  __syncthreads();
  if (tid < C) {
    float v = 0.f;
#pragma unroll
    for (int k = 0; k < 32; ++k) v += smem[k * C + tid];
    a.db_part[(size_t)blockIdx.x * C + tid] = v;
  }
  if constexpr (LIFT)
    for (int i = tid; i < C * 16; i += 512) a.dwl_part[(size_t)blockIdx.x * C * 16 + i] = dls[i] + dls[C * 16 + i];
}
