
// Split-precision variant of k_proj_bwd: ALL THREE GEMMs run as bf16x3 MFMAs on the matrix cores
// (fno_dev.h) with W1 fragments pre-split by k_pack_w1_x3 (L2-resident); the fp32 lanes only do the
// GELU / reductions / splits.  Wave (hm, nt) as in k_proj_fwd.  LDS holds three bf16x3 images:
//   xb [3][NPX][C+8]    a, pixel-major   -> B operand of the P1 recompute (contraction over channels)
//   xr [3][C][NPX+8]    a, row-major     -> B operand of dW1           (contraction over pixels)
//   dr [3][64][NPX+8]   dP1 chunk, row-major -> A operand of dW1
// Per 64-row hidden chunk:
//   A1  recompute P1                                            (24 bf16 MFMAs per wave)
//   E   gl = gelu(P1), dP1 = gelu'(P1) * (W2^T dy); DPP reductions for dW2 / db1; dP1 is split
//       once: the three terms go to `dr` AND stay in registers as the B fragments of A3
//   A3  dx += W1^T dP1 straight from those registers            (24 bf16 MFMAs)
//   --- barrier ---
//   B   dW1[chunk] += dP1 . a^T by the wave group that owns the chunk   (48 bf16 MFMAs)
//   --- barrier --- (dr is single-buffered)
template <int C, int HID, int NPX, int NCO, bool RELU = false>
__global__ void __launch_bounds__(NPX * 4, FNO_OCC_PB) k_proj_bwd_x3(ProjBwdArgs a) {
  using Cfg = ProjBwdCfg<C, HID, NPX>;
  constexpr int NTN = Cfg::NTN, NW = Cfg::NW, MT = Cfg::MT, NCH = Cfg::NCH, TILES = Cfg::TILES, G = Cfg::G,
                CPW = Cfg::CPW;
  constexpr int NT = NW * 64;
  constexpr int PITCH = NPX + 4;
  constexpr int KB = C / 16;
  using SP = SplitTilePrefetch<NPX, NT, C>;      // layout constants of the pixel-major image
  constexpr int RP = NPX + 8;                    // halfs per row of the row-major images
  constexpr int XR_TERM = C * RP, DR_TERM = 64 * RP;
  static_assert((size_t)C * PITCH * 4 <= (size_t)3 * DR_TERM * 2 && (size_t)C * PITCH * 4 <= (size_t)3 * XR_TERM * 2,
                "fp32 tiles alias the bf16 images");
  extern __shared__ __attribute__((aligned(16))) float smem[];
  unsigned short* xb = reinterpret_cast<unsigned short*>(smem);
  unsigned short* xr = xb + 3 * SP::TERM;
  unsigned short* dr = xr + 3 * XR_TERM;
  float* douts = reinterpret_cast<float*>(dr + 3 * DR_TERM);   // NCO x NPX
  float* b1s = douts + NCO * NPX;                               // HID
  float* w2s = b1s + HID;                                       // NCO x HID
  float* tmpf = reinterpret_cast<float*>(dr);    // C x PITCH fp32: staging tile for the split pass, later the gout tile
  float* part = reinterpret_cast<float*>(xr);    // C x PITCH fp32: dx partials of the hm = 1 waves (after the chunk loop)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, half = lane >> 5;
  const int l15 = lane & 15;
  const int hm = wave / NTN, nt = wave % NTN;
  const int n0 = nt * 32;
  const int dgrp = wave / TILES, dtl = wave % TILES;
  const int dmt = dtl / MT, dnt = dtl % MT;  // dW1 tile: hidden 32-block, channel 32-block

  for (int i = tid; i < HID; i += NT) b1s[i] = a.b1[i];
  for (int i = tid; i < NCO * HID; i += NT) w2s[i] = (i < a.CO * HID) ? a.w2[i] : 0.f;
  f32x16 dw1acc[CPW];
#pragma unroll
  for (int k = 0; k < CPW; ++k)
#pragma unroll
    for (int r = 0; r < 16; ++r) dw1acc[k][r] = 0.f;
  // lane accumulates hidden row  ch*64 + hm*32 + acc_row32(reduce16_id(lane), half)  of every chunk
  float sdb1[NCH], sdw2[NCH][NCO];
#pragma unroll
  for (int ch = 0; ch < NCH; ++ch) {
    sdb1[ch] = 0.f;
#pragma unroll
    for (int co = 0; co < NCO; ++co) sdw2[ch][co] = 0.f;
  }

  bf16x8 afn[KB][3];     // A fragments (W1 rows of this wave) of the chunk about to be recomputed
  auto load_w1 = [&](int ch) {
    const unsigned short* wa = a.wa1 + ((size_t)((ch * 2 + hm) * KB * 3) * 64 + lane) * 8;
#pragma unroll
    for (int kb = 0; kb < KB; ++kb)
#pragma unroll
      for (int t = 0; t < 3; ++t) afn[kb][t] = ld8h(wa + (size_t)(kb * 3 + t) * 64 * 8);
  };
  using PFX = TilePrefetch<NPX, NT, C, C>;
  PFX pfx;      // next tile's u_L rows, in flight during this tile
  if ((int)blockIdx.x < a.ntiles)
    pfx.issue(a.x + (size_t)(blockIdx.x / a.tiles_per_plane) * C * a.PW + (blockIdx.x % a.tiles_per_plane) * NPX, a.PW, tid);

  int tslot = 0;
  FNO_TRACE_IF(FNO_TRACE_WHICH == 1);
  FNO_SIMD_PARTNER_PRIO(wave, NW);
  for (int tile = blockIdx.x; tile < a.ntiles; tile += gridDim.x) {
    const int b = tile / a.tiles_per_plane;
    const int px0 = (tile % a.tiles_per_plane) * NPX;
    FNO_STAMP(tslot + 0);
    // commit: a = act(u) -> fp32 staging tile + row-major bf16x3 image
#pragma unroll
    for (int i = 0; i < PFX::ITER; ++i) {
      const int idx = tid + i * NT;
      const int c = idx / (NPX / 4), q = idx % (NPX / 4);
      float4 t = pfx.v[i];
      if (a.act_in) { t.x = gelu_f(t.x); t.y = gelu_f(t.y); t.z = gelu_f(t.z); t.w = gelu_f(t.w); }
      st4(tmpf + c * PITCH + 4 * q, t);
      const float tv[4] = {t.x, t.y, t.z, t.w};
      unsigned short hh[4], mm[4], ll[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) split3(tv[j], hh[j], mm[j], ll[j]);
      unsigned short* dst = xr + c * RP + 4 * q;
      *reinterpret_cast<uint2*>(dst) = make_uint2(hh[0] | ((unsigned)hh[1] << 16), hh[2] | ((unsigned)hh[3] << 16));
      *reinterpret_cast<uint2*>(dst + XR_TERM) = make_uint2(mm[0] | ((unsigned)mm[1] << 16), mm[2] | ((unsigned)mm[3] << 16));
      *reinterpret_cast<uint2*>(dst + 2 * XR_TERM) = make_uint2(ll[0] | ((unsigned)ll[1] << 16), ll[2] | ((unsigned)ll[3] << 16));
    }
    for (int idx = tid; idx < NCO * NPX; idx += NT) {
      const int co = idx / NPX, p = idx % NPX;
      douts[idx] = (co < a.CO) ? a.dy[((size_t)b * a.CO + co) * a.PW + px0 + p] : 0.f;
    }
    FNO_STAMP(tslot + 1);
    __syncthreads();
    FNO_STAMP(tslot + 2);
    {
      const int nt2 = tile + gridDim.x;
      if (nt2 < a.ntiles) {
        int t_ = tid;
        asm volatile("" : "+v"(t_));      // (no hoisted per-lane 64-bit prefetch addresses: k_pw_fwd_x3)
        pfx.issue(a.x + (size_t)(nt2 / a.tiles_per_plane) * C * a.PW + (nt2 % a.tiles_per_plane) * NPX, a.PW, t_);
      }
    }
    // split pass: fp32 tile [c][px] -> pixel-major bf16x3 image (A1's B operand)
    for (int it = tid; it < NPX * (C / 8); it += NT) {
      const int px = it % NPX, cg = it / NPX;
      float v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = tmpf[(cg * 8 + j) * PITCH + px];
      bf16x8 h, m, l;
      split3x8(v, h, m, l);
      unsigned short* dst = xb + px * SP::PBH + cg * 8;
      st8h(dst, h);
      st8h(dst + SP::TERM, m);
      st8h(dst + 2 * SP::TERM, l);
    }
    FNO_STAMP(tslot + 3);
    __syncthreads();            // tmpf (= dr) is free from here on
    FNO_STAMP(tslot + 4);
    const unsigned short* xbp = xb + (n0 + l31) * SP::PBH + 8 * half;   // this lane's pixel row
    float dyl[NCO];
#pragma unroll
    for (int co = 0; co < NCO; ++co) dyl[co] = douts[co * NPX + n0 + l31];

    f32x16 acc2[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc2[m][r] = 0.f;
    if (tile == (int)blockIdx.x) load_w1(0);     // later tiles: chunk 0 was prefetched by the previous tile's last chunk

#pragma unroll 1
    for (int ch = 0; ch < NCH; ++ch) {
      if (ch == 1) FNO_STAMP(tslot + 5);
      // ---- A1 ------------------------------------------------------------
      f32x16 acc, lo1;     // hh products / cross terms of the split (fno_dev.h: mfma_x3s)
#pragma unroll
      for (int r = 0; r < 16; ++r) { acc[r] = 0.f; lo1[r] = 0.f; }
      {
#pragma unroll
        for (int kb = 0; kb < KB; ++kb) {
          bf16x8 bf[3];
#pragma unroll
          for (int t = 0; t < 3; ++t) bf[t] = ld8h(xbp + t * SP::TERM + kb * 16);
          mfma_x3s(afn[kb], bf, acc, lo1);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] += lo1[r];
      }
      if (ch == 1) FNO_STAMP(tslot + 6);
      // ---- E ---------------------------------------------------------------
      bf16x8 bd[2][3];      // dP1 split: accumulator registers 8s..8s+7 = B fragment of hidden k-block s
      {
        unsigned short* drp = dr + (hm * 32 + 4 * half) * RP + n0 + l31;
        const float* b1p = b1s + ch * 64 + hm * 32 + 4 * half;
        const float* w2p = w2s + ch * 64 + hm * 32 + 4 * half;
        float dpv[16], glv[NCO][16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int ro = (r & 3) + 8 * (r >> 2);
          float t = 0.f;
#pragma unroll
          for (int co = 0; co < NCO; ++co) t = fmaf(w2p[co * HID + ro], dyl[co], t);
          float gl, dg;
          if constexpr (RELU) {          // hidden ReLU (rno.py:136-137 regressor head): relu'(0) = 0 as torch
            const float p1 = acc[r] + b1p[ro];
            gl = fmaxf(p1, 0.f);
            dg = p1 > 0.f ? 1.f : 0.f;
          } else {
            gelu_both(acc[r] + b1p[ro], gl, dg);
          }
          const float dp = dg * t;
          unsigned short ph, pm, pl;
          split3(dp, ph, pm, pl);
          bd[r >> 3][0][r & 7] = (short)ph;
          bd[r >> 3][1][r & 7] = (short)pm;
          bd[r >> 3][2][r & 7] = (short)pl;
          drp[ro * RP] = ph;
          drp[ro * RP + DR_TERM] = pm;
          drp[ro * RP + 2 * DR_TERM] = pl;
          dpv[r] = dp;
#pragma unroll
          for (int co = 0; co < NCO; ++co) glv[co][r] = gl * dyl[co];
        }
        // pixel sums of this wave's 32 columns: lane -> accumulator register reduce16_id(lane)
        const float rdb = half_reduce16(dpv, lane);
        float rdw[NCO];
#pragma unroll
        for (int co = 0; co < NCO; ++co) rdw[co] = half_reduce16(glv[co], lane);
#pragma unroll
        for (int k = 0; k < NCH; ++k)
          if (k == ch) {
            sdb1[k] += rdb;
#pragma unroll
            for (int co = 0; co < NCO; ++co) sdw2[k][co] += rdw[co];
          }
      }
      if (ch == 1) FNO_STAMP(tslot + 7);
      // ---- A3: W1^T fragments come in the accumulator's k order (k_pack_w1_x3) ------------
      {
        const unsigned short* wa = a.wa3 + ((size_t)((ch * 2 + hm) * 2 * MT * 3) * 64 + lane) * 8;
#pragma unroll
        for (int mc = 0; mc < MT; ++mc) {
          f32x16 lo3;
#pragma unroll
          for (int r = 0; r < 16; ++r) lo3[r] = 0.f;
#pragma unroll
          for (int s = 0; s < 2; ++s) {
            bf16x8 af[3];
#pragma unroll
            for (int t = 0; t < 3; ++t) af[t] = ld8h(wa + (size_t)((s * MT + mc) * 3 + t) * 64 * 8);
            mfma_x3s(af, bd[s], acc2[mc], lo3);
          }
#pragma unroll
          for (int r = 0; r < 16; ++r) acc2[mc][r] += lo3[r];
        }
      }
      // W1 fragments of the NEXT chunk: L2 latency hides behind the barrier and the dW1 phase
      load_w1(ch + 1 < NCH ? ch + 1 : 0);
      if (ch == 1) FNO_STAMP(tslot + 8);
      __syncthreads();
      if (ch == 1) FNO_STAMP(tslot + 9);
      // ---- B: dW1[hid][c] += sum_px dP1[hid][px] a[c][px], both operands row-major bf16x3 -----
      if (dgrp == ch % G) {
        const unsigned short* ga = dr + (dmt * 32 + l31) * RP + 8 * half;
        const unsigned short* ab = xr + (dnt * 32 + l31) * RP + 8 * half;
#pragma unroll
        for (int k = 0; k < CPW; ++k)
          if (k == ch / G) {
            f32x16 dacc = dw1acc[k];
#pragma unroll 2
            for (int kq = 0; kq < NPX / 16; ++kq) {
              bf16x8 af[3], bf[3];
#pragma unroll
              for (int t = 0; t < 3; ++t) {
                af[t] = ld8h(ga + t * DR_TERM + kq * 16);
                bf[t] = ld8h(ab + t * XR_TERM + kq * 16);
              }
              dacc = mfma_x3(af, bf, dacc);
            }
            dw1acc[k] = dacc;
          }
      }
      if (ch == 1) FNO_STAMP(tslot + 10);
      __syncthreads();   // dr is rewritten by the next chunk
      if (ch == 1) FNO_STAMP(tslot + 11);
    }

    FNO_STAMP(tslot + 12);
    // ---- dx: add the two hidden halves, (x act'), store, row DFT -------------
    // C = 64: wave (hm, nt) finalises channel block m = hm of its 32 pixels and hands the other block to its
    // partner; C = 32: the hm = 0 wave finalises the single block
    {
      float* pp = part + (4 * half) * PITCH + n0 + l31;
#pragma unroll
      for (int m = 0; m < MT; ++m)
        if (!((MT == 2) ? (m == hm) : (hm == 0))) {
#pragma unroll
          for (int r = 0; r < 16; ++r) pp[(m * 32 + (r & 3) + 8 * (r >> 2)) * PITCH] = acc2[m][r];
        }
    }
    __syncthreads();
    {
      const float* pp = part + (4 * half) * PITCH + n0 + l31;
      float* xp = tmpf + (4 * half) * PITCH + n0 + l31;
      const size_t goff = ((size_t)b * C + 4 * half) * a.PW + px0 + n0 + l31;
#pragma unroll
      for (int m = 0; m < MT; ++m)
        if ((MT == 2) ? (m == hm) : (hm == 0)) {
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int ro = m * 32 + (r & 3) + 8 * (r >> 2);
            float v = acc2[m][r] + pp[ro * PITCH];
            if (a.act_in) v *= gelu_grad_f(a.x[goff + (size_t)ro * a.PW]);
            a.gout[goff + (size_t)ro * a.PW] = v;
            if (a.x1g) xp[ro * PITCH] = v;
          }
        }
    }
    FNO_STAMP(tslot + 13);
    if (a.x1g) {
      __syncthreads();
      FNO_STAMP(tslot + 14);
      row_dft_epilogue<C, NPX, NW>(tmpf, a.tfwd, a.W, a.x1g, b, px0, a.P, a.W, a.K2out, a.NJ, wave, lane);
    }
    FNO_STAMP(tslot + 15);
    __syncthreads();
    tslot += 16;
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
      const int hid = ch * 64 + hm * 32 + acc_row32(reduce16_id(lane), half);
      a.db1_part[slab * HID + hid] = sdb1[ch];
#pragma unroll
      for (int co = 0; co < NCO; ++co)
        if (co < a.CO) a.dw2_part[(slab * a.CO + co) * HID + hid] = sdw2[ch][co];
    }
  }
}


// W1 (HID, C) fp32 -> bf16x3 MFMA A-fragments for k_proj_bwd_x3 (once per step, 2 x 96 KB at C = 64):
//   wa1[((mt*KB + kb)*3 + t)*64 + lane][j] = term t of W1[mt*32 + (lane&31)][kb*16 + 8*(lane>>5) + j]
//   wa3[(((mt*2 + s)*MT + mc)*3 + t)*64 + lane][j] = term t of W1[mt*32 + 16s + 8(j>>2) + 4(lane>>5) + (j&3)][mc*32 + (lane&31)]
__global__ void k_pack_w1_x3(const float* __restrict__ w1, unsigned short* __restrict__ wa1,
                             unsigned short* __restrict__ wa3, int HID, int C) {
  const int KB = C / 16, MT = C / 32;
  const int n1 = (HID / 32) * KB * 64, n3 = (HID / 32) * 2 * MT * 64;
  const int it = blockIdx.x * blockDim.x + threadIdx.x;
  float v[8];
  bf16x8 h, m, l;
  if (it < n1) {
    const int ln = it & 63, kb = (it >> 6) % KB, mt = (it >> 6) / KB;
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = w1[(size_t)(mt * 32 + (ln & 31)) * C + kb * 16 + 8 * (ln >> 5) + j];
    split3x8(v, h, m, l);
    unsigned short* dst = wa1 + ((size_t)((mt * KB + kb) * 3) * 64 + ln) * 8;
    st8h(dst, h); st8h(dst + 64 * 8, m); st8h(dst + 2 * 64 * 8, l);
  } else if (it < n1 + n3) {
    const int i3 = it - n1;
    const int ln = i3 & 63, mc = (i3 >> 6) % MT, s = ((i3 >> 6) / MT) & 1, mt = (i3 >> 6) / MT / 2;
#pragma unroll
    for (int j = 0; j < 8; ++j)
      v[j] = w1[(size_t)(mt * 32 + 16 * s + 8 * (j >> 2) + 4 * (ln >> 5) + (j & 3)) * C + mc * 32 + (ln & 31)];
    split3x8(v, h, m, l);
    unsigned short* dst = wa3 + ((size_t)(((mt * 2 + s) * MT + mc) * 3) * 64 + ln) * 8;
    st8h(dst, h); st8h(dst + 64 * 8, m); st8h(dst + 2 * 64 * 8, l);
  }
}


// k_proj_fwd_x3 with two fp16 terms.  a.xmax: device scalar, a bound of |x| (required)
template <int C, int HID, int NPX, int NCO, bool RELU = false>
__global__ void __launch_bounds__(NPX * 4, 2) k_proj_fwd_h2(ProjFwdArgs a) {
  FNO_CLK_ENTRY();
  constexpr int NTN = NPX / 32;
  constexpr int NW = 2 * NTN;
  constexpr int NT = NW * 64;
  constexpr int KB = C / 16;
  constexpr int NCH = HID / 64;
  using PF = SplitTilePrefetchH2<NPX, NT, C>;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  unsigned short* xb = reinterpret_cast<unsigned short*>(smem);          // 2 x NPX x (C+8) halfs
  unsigned short* w1b = xb + 2 * PF::TERM;                                // (HID/32) x KB x 2 x 64 x 8 halfs
  float* b1s = reinterpret_cast<float*>(w1b + (HID / 32) * KB * 2 * 64 * 8);   // HID
  float* w2s = b1s + HID;                                                 // NCO x HID
  float* ysh = w2s + NCO * HID;                                           // NCO x NPX (first floats: reduction scratch)
  float gk_six, gk_inf;
  gelu_consts(gk_six, gk_inf);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, half = lane >> 5;
  const int hm = wave / NTN, nt = wave % NTN;
  const int n0 = nt * 32;

  const float sx = h2_scale(*a.xmax);                                     // activation scale (|gelu(x)| <= |x|)
  const float sw = h2_scale(wg_absmax<NT>(a.w1, HID * C, ysh, tid));      // weight scale
  const float inv = 1.0f / (sx * sw);                                     // exact: powers of two
  for (int i = tid; i < HID; i += NT) b1s[i] = a.b1[i];
  for (int i = tid; i < NCO * HID; i += NT) w2s[i] = (i < a.CO * HID) ? a.w2[i] : 0.f;
  for (int it = tid; it < (HID / 32) * KB * 64; it += NT) {      // item = (mt, kb, lane)
    const int ln = it & 63, kb = (it >> 6) % KB, mt = (it >> 6) / KB;
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = a.w1[(size_t)(mt * 32 + (ln & 31)) * C + kb * 16 + 8 * (ln >> 5) + j];
    f16x8 h, l;
    split2x8(v, sw, h, l);
    unsigned short* dst = w1b + ((size_t)((mt * KB + kb) * 2) * 64 + ln) * 8;
    *reinterpret_cast<f16x8*>(dst) = h;
    *reinterpret_cast<f16x8*>(dst + 64 * 8) = l;
  }

  PF pfx;
  if ((int)blockIdx.x < a.ntiles)
    pfx.issue(a.x + (size_t)(blockIdx.x / a.tiles_per_plane) * C * a.PW + (blockIdx.x % a.tiles_per_plane) * NPX, a.PW, tid);

  FNO_CLK_BEGIN();
  for (int tile = blockIdx.x; tile < a.ntiles; tile += gridDim.x) {
    const int b = tile / a.tiles_per_plane;
    const int px0 = (tile % a.tiles_per_plane) * NPX;
    pfx.commit(xb, a.act_in != 0, sx, gk_six, gk_inf, tid);
    __syncthreads();
    {
      const int nt2 = tile + gridDim.x;
      if (nt2 < a.ntiles)
        pfx.issue(a.x + (size_t)(nt2 / a.tiles_per_plane) * C * a.PW + (nt2 % a.tiles_per_plane) * NPX, a.PW, tid);
    }
    // this wave's activation fragments: B[k = c][n = px], 8 consecutive channels per lane
    f16x8 bfrag[KB][2];
#pragma unroll
    for (int kb = 0; kb < KB; ++kb)
#pragma unroll
      for (int t = 0; t < 2; ++t)
        bfrag[kb][t] = *reinterpret_cast<const f16x8*>(xb + t * PF::TERM + (n0 + l31) * PF::PBH + kb * 16 + 8 * half);

    float ysum[NCO];
#pragma unroll
    for (int co = 0; co < NCO; ++co) ysum[co] = 0.f;
    // One-chunk software pipeline (round 4 experiment, -DPFWD_PIPE=1): the products of chunk ch + 1 ISSUED before the GELU of
    // chunk ch, so that a wave's matrix work runs under its own vector work.  Measured 0.194-0.199 vs 0.200-0.212 ms per
    // launch at config 2 (within the noise of the boxes) for 208 instead of 168 VGPRs, and the four-output variant spills with
    // it: off.  (One output channel takes k_proj_fwd_w below by default.)
#ifndef PFWD_PIPE
#define PFWD_PIPE 0
#endif
    auto products = [&](int ch, f32x16& acc, f32x16& lo) {
#pragma unroll
      for (int r = 0; r < 16; ++r) { acc[r] = 0.f; lo[r] = 0.f; }
      const unsigned short* wa = w1b + ((size_t)((ch * 2 + hm) * KB * 2) * 64 + lane) * 8;
#pragma unroll
      for (int kb = 0; kb < KB; ++kb) {
        f16x8 af[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) af[t] = *reinterpret_cast<const f16x8*>(wa + (size_t)(kb * 2 + t) * 64 * 8);
        mfma_h2s(af, bfrag[kb], acc, lo);
      }
    };
    auto activate = [&](int ch, const f32x16& acc, const f32x16& lo) {
      const float* b1p = b1s + ch * 64 + hm * 32 + 4 * half;
      const float* w2p = w2s + ch * 64 + hm * 32 + 4 * half;
      f32x2 hp[8];
#pragma unroll
      for (int r = 0; r < 16; r += 2) {
        hp[r >> 1][0] = fmaf(acc[r] + lo[r], inv, b1p[(r & 3) + 8 * (r >> 2)]);
        hp[r >> 1][1] = fmaf(acc[r + 1] + lo[r + 1], inv, b1p[((r + 1) & 3) + 8 * ((r + 1) >> 2)]);
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
    };
#if PFWD_PIPE
    static_assert(NCH % 2 == 0, "two accumulator sets alternate");
    f32x16 accA, loA, accB, loB;
    products(0, accA, loA);
#pragma unroll
    for (int ch = 0; ch < NCH; ch += 2) {
      products(ch + 1, accB, loB);
      __builtin_amdgcn_sched_barrier(0);        // (the products above stay ahead of the vector work below)
      activate(ch, accA, loA);
      if (ch + 2 < NCH) products(ch + 2, accA, loA);
      __builtin_amdgcn_sched_barrier(0);
      activate(ch + 1, accB, loB);
    }
#else
#pragma unroll 1
    for (int ch = 0; ch < NCH; ++ch) {
      f32x16 acc, lo;      // hh products / cross terms
      products(ch, acc, lo);
      activate(ch, acc, lo);
    }
#endif
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
  FNO_CLK_END(2);
}


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
__global__ void k_absmax3(const float* __restrict__ x0, size_t n0, int g0, const float* __restrict__ x1, size_t n1, int g1,
                          const float* __restrict__ x2, size_t n2, float* __restrict__ dst, float* __restrict__ sum0) {
  absmax3_block(x0, n0, g0, x1, n1, g1, x2, n2, (int)gridDim.x - g0 - g1, dst, sum0);
}


// ---------------------------------------------------------------------------
// Split-precision variant: both GEMMs of the block backward run as bf16x3 MFMAs on the matrix
// cores (fno_dev.h); the fp32 lanes keep the GELU derivative, the splits and the row DFT.
// LDS (C = 64, NPX = 128: 152 KB with all tables):
//   R12  gr[3][C][NPX+8] | ar[3][C][NPX+8]   row-major bf16x3 of g and a_l = act(u_l): dW operands
//        ... after the dW GEMM the same bytes hold  gb[3][NPX][C+8] (pixel-major g, the dx GEMM's
//        B operand) | dg tile fp32 C x PITCH (gelu'(u_l) in the accumulator's layout)
//   R3   fp32 C x PITCH: g as loaded (for dbias and the pixel-major split pass), later the gout tile
// Per tile:  commit (GELU, row-major splits) | dW GEMM + dbias | split pass + dg | dx GEMM, x gelu',
// gout store | row DFT / lifting gradients - five barriers, as in the fp32 kernel.
template <int C, int NPX, bool LOOSE = false, bool LIFT = false>
__global__ void __launch_bounds__((C / 32) * (NPX / 32) * 64, FNO_OCC_BB) k_block_bwd_x3(BlkBwdArgs a) {
  using Cfg = BlkBwdCfg<C, NPX>;
  constexpr int NTN = Cfg::NTN, MT = Cfg::MT, NW = Cfg::NW, TILES = Cfg::TILES, KSPLIT = Cfg::KSPLIT;
  constexpr int NT = NW * 64;
  constexpr int KB = C / 16;
  constexpr int PITCH = NPX + 4;
  constexpr int RP = NPX + 8;                          // halfs per row of the row-major images
  constexpr int RTERM = C * RP;
  constexpr int PBH = C + 8, PTERM = NPX * PBH;        // pixel-major image
  constexpr int DBPX = NPX / (NT / C);
  static_assert(NT % C == 0 && DBPX % 4 == 0, "dbias thread mapping");
  constexpr int PXK = NPX / KSPLIT;
  static_assert(PXK % 16 == 0, "dW k blocks");
  constexpr int LJ = (C / 16 + NW - 1) / NW;
  static_assert((size_t)3 * PTERM * 2 + (size_t)C * PITCH * 4 <= (size_t)6 * RTERM * 2, "gb + dg alias the row-major images");
  extern __shared__ __attribute__((aligned(16))) float smem[];
  unsigned short* gr = reinterpret_cast<unsigned short*>(smem);
  unsigned short* ar = gr + 3 * RTERM;
  unsigned short* gb = gr;                                            // after the dW GEMM
  float* dgs = reinterpret_cast<float*>(gr + 3 * PTERM);             // after the dW GEMM
  float* r3 = reinterpret_cast<float*>(gr + 6 * RTERM);              // C x PITCH fp32
  float* xls = r3 + C * PITCH;
  float* tinv_s = xls + (a.xin ? 8 * PITCH : 0);
  const int R = LOOSE ? NPX / a.W + 2 : NPX / a.W;
  const int KC = (LOOSE && a.kch > 0 && a.kch < a.K2in) ? a.kch : a.K2in;     // modes resident in LDS at a time
  const bool chunked = KC < a.K2in;
  float* zs = tinv_s + (a.zg ? 2 * KC * a.W : 0);
  float* tfwd_s = zs + (a.zg ? R * KC * C * 2 : 0);

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, half = lane >> 5;
  const int l15 = lane & 15, quad = lane >> 4;
  const int mt = wave / NTN, nt = wave % NTN;
  const int n0 = nt * 32;
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

  // A fragments of W^T: A[i][k = o] = W[o][i], split into (h, m, l)
  bf16x8 afrag[KB][3];
#pragma unroll
  for (int kb = 0; kb < KB; ++kb) {
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = a.w[(kb * 16 + 8 * half + j) * C + mt * 32 + l31];
    split3x8(v, afrag[kb][0], afrag[kb][1], afrag[kb][2]);
  }

  f32x16 dwacc;
#pragma unroll
  for (int r = 0; r < 16; ++r) dwacc[r] = 0.0f;
  float dbsum = 0.0f;
  f32x4 dl[LJ];
#pragma unroll
  for (int j = 0; j < LJ; ++j) dl[j] = f32x4{0.f, 0.f, 0.f, 0.f};

  using PF = TilePrefetch<NPX, NT, C, C>;
  PF pfg, pfu;
  float4 xl[4];                  // LIFT: the lifting input under this thread's pixel group
  __shared__ __attribute__((aligned(16))) float lws[LIFT ? 5 * C : 4];
  if constexpr (LIFT) { stage_lift_params<C>(lws, a.lw, a.lb, a.CL, tid, NT); __syncthreads(); }
  static_assert(!LIFT || NT % (NPX / 4) == 0, "LIFT: one pixel group per thread");
  float4 zv = make_float4(0.f, 0.f, 0.f, 0.f);
  auto issue = [&](int tile) {
    const int b = tile / a.tiles_per_plane;
    const int px0 = (tile % a.tiles_per_plane) * NPX;
    int t = tid;
    asm volatile("" : "+v"(t));      // (see k_pw_fwd_x3: no hoisted per-lane 64-bit prefetch addresses)
    pfg.issue(a.g + (size_t)b * C * a.PW + px0, a.PW, t);
    if constexpr (LIFT) {     // this thread's 4 pixels of the <= 4 input rows (q = tid % (NPX / 4) for all its items)
      const float* xb = a.xin + (size_t)b * a.CL * a.PW + px0 + 4 * (t % (NPX / 4));
#pragma unroll
      for (int k = 0; k < 4; ++k) xl[k] = k < a.CL ? ld4(xb + (size_t)k * a.PW) : make_float4(0.f, 0.f, 0.f, 0.f);
    } else {
      pfu.issue(a.uin + (size_t)b * C * a.PW + px0, a.PW, t);
    }
    if (t < zc4(px0)) zv = ld4(a.zg + ((size_t)b * a.P + px0 / a.W) * a.K2in * C * 2 + 4 * t);
  };
  if ((int)blockIdx.x < a.ntiles) issue(blockIdx.x);
  auto put_row4 = [&](unsigned short* img, int c, int q, const float4& t) {   // 4 pixels of row c -> 3 terms
    const float tv[4] = {t.x, t.y, t.z, t.w};
    unsigned short hh[4], mm[4], ll[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) split3(tv[j], hh[j], mm[j], ll[j]);
    unsigned short* dst = img + c * RP + 4 * q;
    *reinterpret_cast<uint2*>(dst) = make_uint2(hh[0] | ((unsigned)hh[1] << 16), hh[2] | ((unsigned)hh[3] << 16));
    *reinterpret_cast<uint2*>(dst + RTERM) = make_uint2(mm[0] | ((unsigned)mm[1] << 16), mm[2] | ((unsigned)mm[3] << 16));
    *reinterpret_cast<uint2*>(dst + 2 * RTERM) = make_uint2(ll[0] | ((unsigned)ll[1] << 16), ll[2] | ((unsigned)ll[3] << 16));
  };

  int tslot = 0;
  FNO_TRACE_IF(false);
  FNO_SIMD_PARTNER_PRIO(wave, (C / 32) * (NPX / 32));
  for (int tile = blockIdx.x; tile < a.ntiles; tile += gridDim.x) {
    const int b = tile / a.tiles_per_plane;
    const int px0 = (tile % a.tiles_per_plane) * NPX;
    FNO_STAMP(tslot + 0);
    // ---- commit: g -> fp32 tile + row-major image; a_l = act(u_l) -> row-major image; gelu' stays in registers
    float4 dgv[PF::ITER];
#pragma unroll
    for (int i = 0; i < PF::ITER; ++i) {
      const int idx = tid + i * NT;
      const int c = idx / (NPX / 4), q = idx % (NPX / 4);
      const float4 gv = pfg.v[i];
      st4(r3 + c * PITCH + 4 * q, gv);
      put_row4(gr, c, q, gv);
      float4 uv;
      if constexpr (LIFT) {      // u_0 = W_l x + b_l, never stored by the forward pass
        const float bc = lws[4 * C + c];
        const float4 wv = ld4(lws + 4 * c);
        const float wk[4] = {wv.x, wv.y, wv.z, wv.w};
        uv = make_float4(bc, bc, bc, bc);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          uv.x = fmaf(wk[k], xl[k].x, uv.x); uv.y = fmaf(wk[k], xl[k].y, uv.y);
          uv.z = fmaf(wk[k], xl[k].z, uv.z); uv.w = fmaf(wk[k], xl[k].w, uv.w);
        }
      } else {
        uv = pfu.v[i];
      }
      if (a.act_in) {
        gelu_both(uv.x, uv.x, dgv[i].x);
        gelu_both(uv.y, uv.y, dgv[i].y);
        gelu_both(uv.z, uv.z, dgv[i].z);
        gelu_both(uv.w, uv.w, dgv[i].w);
      }
      put_row4(ar, c, q, uv);
    }
    const int zcount4 = zc4(px0);
    if (tid < zcount4) st4(zs + 4 * tid, zv);
    for (int i = tid + NT; i < zcount4; i += NT)      // more spectral rows than threads (short rows, many modes)
      st4(zs + 4 * i, ld4(a.zg + ((size_t)b * a.P + px0 / a.W) * a.K2in * C * 2 + 4 * i));
    if (a.xin) stage_rows<NPX, NT>(xls, a.xin + (size_t)b * a.CL * a.PW + px0, a.PW, a.CL, a.CL, false, tid);
    FNO_STAMP(tslot + 1);
    __syncthreads();
    FNO_STAMP(tslot + 2);
    if (tile + (int)gridDim.x < a.ntiles) issue(tile + gridDim.x);

    {  // dbias[c] partial
      const float* gq = r3 + (tid % C) * PITCH + (tid / C) * DBPX;
#pragma unroll
      for (int j = 0; j < DBPX / 4; ++j) {
        const float4 gv = ld4(gq + 4 * j);
        dbsum += (gv.x + gv.y) + (gv.z + gv.w);
      }
    }
    FNO_STAMP(tslot + 3);
    // ---- dW[o][i] += sum_px g[o][px] a[i][px]: both operands row-major, 8 consecutive pixels per lane
    {
      const unsigned short* ga = gr + (dmt * 32 + l31) * RP + dkp * PXK + 8 * half;
      const unsigned short* ab = ar + (dnt * 32 + l31) * RP + dkp * PXK + 8 * half;
#pragma unroll
      for (int kq = 0; kq < PXK / 16; ++kq) {
        bf16x8 af[3], bf[3];
#pragma unroll
        for (int t = 0; t < 3; ++t) {
          af[t] = ld8h(ga + t * RTERM + kq * 16);
          bf[t] = ld8h(ab + t * RTERM + kq * 16);
        }
        dwacc = mfma_x3(af, bf, dwacc);
      }
    }
    FNO_STAMP(tslot + 4);
    __syncthreads();          // row-major images are dead: their bytes become gb + dg
    FNO_STAMP(tslot + 5);
    // ---- pixel-major image of g (B operand of the dx GEMM) from the fp32 tile; gelu' to its tile
    for (int it = tid; it < NPX * (C / 8); it += NT) {
      const int px = it % NPX, cg = it / NPX;
      float v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = r3[(cg * 8 + j) * PITCH + px];
      bf16x8 h, m, l;
      split3x8(v, h, m, l);
      unsigned short* dst = gb + px * PBH + cg * 8;
      st8h(dst, h);
      st8h(dst + PTERM, m);
      st8h(dst + 2 * PTERM, l);
    }
    if (a.act_in) {
#pragma unroll
      for (int i = 0; i < PF::ITER; ++i) {
        const int idx = tid + i * NT;
        st4(dgs + (idx / (NPX / 4)) * PITCH + 4 * (idx % (NPX / 4)), dgv[i]);
      }
    }
    FNO_STAMP(tslot + 6);
    __syncthreads();
    FNO_STAMP(tslot + 7);
    // ---- dx GEMM (+ row inverse DFT of the spectral gradient) -------------
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
    {
      const unsigned short* gp = gb + (n0 + l31) * PBH + 8 * half;
#pragma unroll
      for (int kb = 0; kb < KB; ++kb) {
        bf16x8 bf[3];
#pragma unroll
        for (int t = 0; t < 3; ++t) bf[t] = ld8h(gp + t * PTERM + kb * 16);
        acc = mfma_x3(afrag[kb], bf, acc);
      }
    }
    if constexpr (LOOSE) {
      if (a.zg && !chunked) acc = kext_loose_rows<C>(acc, zs, tinv_s, a.K2in, a.W, px0 + n0, px0 / a.W, mt, l31, half);
      else if (a.zg) {
        const int r_lo = px0 / a.W, nrows = (px0 + NPX - 1) / a.W - r_lo + 1;
        for (int k0 = 0; k0 < a.K2in; k0 += KC) {
          const int kc = min(KC, a.K2in - k0);
          __syncthreads();                        // every wave is done with the previous chunk (or the previous tile's last)
          for (int i = tid; i < 2 * kc * a.W; i += NT) tinv_s[i] = a.tinv[2 * k0 * a.W + i];
          const int per_row4 = kc * C / 2;        // float4s of one row's chunk: modes k0 .. k0 + kc are contiguous in a row
          for (int i = tid; i < nrows * per_row4; i += NT) {
            const int r = i / per_row4, rem = i - r * per_row4;
            st4(zs + 4 * i, ld4(a.zg + (((size_t)b * a.P + r_lo + r) * a.K2in + k0) * C * 2 + 4 * rem));
          }
          __syncthreads();
          acc = kext_loose_rows<C>(acc, zs, tinv_s, kc, a.W, px0 + n0, r_lo, mt, l31, half);
        }
      }
    } else if (a.zg) {
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
    FNO_STAMP(tslot + 8);
    {
      const float* dq = dgs + (mt * 32 + 4 * half) * PITCH + n0 + l31;
      float* gq = r3 + (mt * 32 + 4 * half) * PITCH + n0 + l31;      // the fp32 g tile is dead since the split pass
      float* gp = a.gout ? a.gout + ((size_t)b * C + mt * 32 + 4 * half) * a.PW + px0 + n0 + l31 : nullptr;
      if (a.act_in) {
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] *= dq[((r & 3) + 8 * (r >> 2)) * PITCH];
      }
      if (gp) {
#pragma unroll
        for (int r = 0; r < 16; ++r) gp[(size_t)((r & 3) + 8 * (r >> 2)) * a.PW] = acc[r];
      }
      if (a.x1g || a.xin) {
#pragma unroll
        for (int r = 0; r < 16; ++r) gq[((r & 3) + 8 * (r >> 2)) * PITCH] = acc[r];
      }
    }
    FNO_STAMP(tslot + 9);
    if (a.x1g || a.xin) {
      __syncthreads();
      FNO_STAMP(tslot + 10);
      if (a.x1g) row_dft_epilogue<C, NPX, NW>(r3, tfwd_s, a.W + 4, a.x1g, b, px0, a.P, a.W, a.K2out, a.NJ, wave, lane);
      if (a.xin) {
        // dl[c][n] += sum_px gout[c][px] * xext[n][px],  xext = [x_in rows | ones | 0..]
#pragma unroll
        for (int j = 0; j < LJ; ++j) {
          const int jm = wave + j * NW;
          if (jm < C / 16) {
            const float* arow = r3 + (jm * 16 + l15) * PITCH + quad;
            const float* br = xls + (l15 < a.CL ? l15 : 0) * PITCH + quad;
            const float cst = l15 == a.CL ? 1.0f : 0.0f;
            for (int s = 0; s < NPX / 4; ++s) {
              const float bf = (l15 < a.CL) ? br[4 * s] : cst;
              dl[j] = mfma16(arow[4 * s], bf, dl[j]);
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
