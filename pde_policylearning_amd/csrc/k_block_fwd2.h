// Forward of one fused FNO block, second generation (whole rows, 128-pixel tiles).  Same mathematics and arguments as
// k_pw_fwd_x3 (k_pointwise.h; reference semantics: fno_block.py:123-150 + the last-dim passes of
// spectral_convolution.py:324,342-345), rebuilt around what bounds that kernel on gfx950: VALU issue, not HBM and not the
// matrix pipe (2.3 k vector instructions per tile and wave, a third of them address arithmetic, and 1.8 k cycles of fp32
// MFMAs that execute on the same lanes).
//   * the tile comes in through BUFFER loads (descriptor + row offset in SGPRs, one 32-bit lane offset) and the output leaves
//     through buffer stores: no 64-bit per-lane address arithmetic at all;
//   * the GEMM is computed TRANSPOSED (D[pixel][channel] - the same LDS fragments with the MFMA operands swapped): a lane
//     owns one output channel and runs of 4 consecutive pixels, so the bias is one register, u leaves as 16-byte stores and
//     the tile for the row DFT as 16-byte LDS writes;
//   * the spectral K-extension (the last-dim inverse DFT folded into the GEMM) runs on the bf16 matrix pipe as well: the
//     inverse table is split once per workgroup, the tile's spectral rows once per tile (a few hundred values), and the
//     extension becomes one more 16-deep k block of the split-precision GEMM instead of 2 x K2in fp32 MFMAs on the VALU lanes;
//   * GELU is evaluated on pairs (v_pk_fma_f32 Horner steps, v_med3 clamps: no canonicalisation moves);
//   * the pixel-major split image is XOR-swizzled instead of padded (48 instead of 55 KB at 64 channels), which pays for the
//     split tables: two 4-wave workgroups still share a CU.
// Per tile: commit (GELU, split -> image; spectral rows -> image) | barrier | GEMM + K-extension | barrier | bias, u store,
// act_out -> fp32 tile | barrier | row DFT of the tile (fp32 MFMA) | barrier.
#pragma once
#include "fno_dev.h"
#include "k_pointwise.h"
#include <type_traits>

// Pixel-major split image [term][pixel][C x bf16], rows of 2 C bytes, 16-byte chunks XOR-swizzled by the pixel index so that
// b128 reads with lanes <-> consecutive pixels (one chunk index per wave half) and the b128 writes of the commit (lanes <->
// consecutive pixels) are bank-conflict-free without row padding.
template <int C>
FNO_DEV int pimg_off(int px, int ch) {
  if constexpr (C == 64) return 128 * px + 16 * (ch ^ ((px >> 1) & 7));
  else return 64 * px + 16 * (ch ^ ((px >> 2) & 3));
}

static inline size_t blk_fwd_t_lds_bytes(int c, int W, int K2in, int NJ, bool has_z, bool has_x1) {
  size_t bytes = (size_t)3 * 128 * c * 2;
  const size_t out_tile = (size_t)c * (128 + 4) * 4;
  if (out_tile > bytes) bytes = out_tile;
  const int KZ = (2 * K2in + 15) / 16;
  if (has_z) bytes += (size_t)3 * W * KZ * 32 + (size_t)3 * (128 / W) * c * KZ * 32;
  if (has_x1) bytes += (size_t)16 * NJ * (W + 4) * 4;
  return bytes;
}


// LIFT: block 0 of a model with a lifting layer: the tile is u_0 = W_l x + b_l of the <= 4-channel model input, computed on
// commit (k_pointwise.h, LiftSplitTilePrefetch).  RELU: the stored tensor is max(u, 0) (rno.py:92-106 regressor layers).
// ACT_IN: GELU on load (a.act_in).  EPI: 0 = store only, 1 = store + row DFT of the output, 2 = store + row DFT of gelu(output)
// (a.x1 / a.act_out).  ADD: a tensor is added to the output before the store (a.add; EPI 0 only).  Compile-time so that the
// unrolled commit / epilogue bodies carry no per-iteration branches.  KZ: 16-deep k blocks of the spectral extension
// (0 = no spectral branch, 1 = up to 8 kept last-dim modes, 2 = up to 16).  NT3: terms of the channel GEMM's operands: 3 = bf16,
// six products per k block; 2 = fp16 (fno_dev.h "h2"), three products, the activation scaled by the bound a.xmax of |x|
// (LIFT: of the model input) and the weights by their own maximum; the spectral extension keeps three bf16 terms.
template <int C, bool LIFT, bool RELU, bool ACT_IN, int EPI, bool ADD, int KZ, int NT3 = 3>
__global__ void __launch_bounds__((C / 32) * 2 * 64, 2) k_blk_fwd_t(PwFwdArgs a) {
  FNO_CLK_ENTRY();
  static_assert(!(LIFT && ACT_IN) && !(ADD && EPI != 0), "variants");
  constexpr int NPX = 128, MT = C / 32, NTG = 2, NTW = 2, NW = MT * NTG, NT = NW * 64, KB = C / 16;
  constexpr int TERM = NPX * C * 2;                      // bytes per term plane of the activation image
  constexpr int PITCH = NPX + 4;
  constexpr int ITER = NPX * (C / 8) / NT;               // (pixel, 8-channel group) items per thread: 4
  constexpr int CGS = NT / NPX;                          // channel groups covered per pass: 2 (C = 64) or 1 (C = 32)
  static_assert(ITER * NT == NPX * (C / 8) && NT % NPX == 0, "one pixel per thread");
  extern __shared__ __attribute__((aligned(16))) float smem[];
  unsigned char* xb = reinterpret_cast<unsigned char*>(smem);
  float* xs = smem;                                      // fp32 output tile C x PITCH for the row DFT: reuses the image
  constexpr size_t REGION = (size_t)NT3 * TERM > (size_t)C * PITCH * 4 ? (size_t)NT3 * TERM : (size_t)C * PITCH * 4;
  const int R = NPX / a.W;
  const int TT = a.W * KZ * 32, ZT = R * C * KZ * 32;    // bytes per term plane of the table / spectral-row images
  unsigned char* timg = xb + REGION;                     // [3][W][KZ * 16] bf16: Tinv^T, k = 2 s + (re, im)
  unsigned char* zimg = timg + 3 * TT;                   // [3][R][C][KZ * 16] bf16: the tile's spectral rows
  float* tfwd_s = reinterpret_cast<float*>(zimg + 3 * ZT);

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, half = lane >> 5;
  const int mt = wave / NTG, ng = wave % NTG;
  const int opx = tid & (NPX - 1);                       // the pixel this thread stages
  const int cg0 = __builtin_amdgcn_readfirstlane(tid / NPX);
  float six, inf;                                        // clamp constants of the packed GELU, kept in SGPRs
  gelu_consts(six, inf);

  const int zc4 = KZ > 0 ? R * a.K2in * C / 2 : 0;          // float4 pieces of one tile's spectral rows
  const unsigned PWb = (unsigned)a.PW * 4u;              // bytes per channel row
  float pv[ITER][8];                                     // the NEXT tile, 8 channels of this thread's pixel per item
  float xin[4];                                          // LIFT: the model input at this thread's pixel
  constexpr int ZP = 2;                                  // float4 pieces of spectral rows prefetched per thread
  float4 zpf[ZP];
  auto issue = [&](int tile_) {
    const int tile = a.rev ? a.ntiles - 1 - tile_ : tile_;      // (zigzag along the kernel chain: k_blk_fwd_s)
    const int b = tile / a.tiles_per_plane;
    const int px0 = (tile % a.tiles_per_plane) * NPX;
    if constexpr (LIFT) {
      const __amdgpu_buffer_rsrc_t rx = make_rsrc(a.x + (size_t)b * a.CL * a.PW + px0, (unsigned)(a.CL - 1) * PWb + NPX * 4);
#pragma unroll
      for (int k = 0; k < 4; ++k) xin[k] = k < a.CL ? buf_ld1(rx, opx * 4, k * PWb) : 0.f;
    } else {
      const __amdgpu_buffer_rsrc_t rx = make_rsrc(a.x + (size_t)b * C * a.PW + px0, (unsigned)(C - 1) * PWb + NPX * 4);
      unsigned so = (unsigned)cg0 * 8u * PWb;              // running row offset: one scalar add per load
#pragma unroll
      for (int i = 0; i < ITER; ++i) {
#pragma unroll
        for (int j = 0; j < 8; ++j) { pv[i][j] = buf_ld1(rx, opx * 4, so); so += PWb; }
        so += (unsigned)(CGS - 1) * 8u * PWb;
      }
    }
#pragma unroll
    for (int k = 0; k < ZP; ++k)
      if (tid + k * NT < zc4) zpf[k] = ld4(a.z + ((size_t)b * a.P + px0 / a.W) * a.K2in * C * 2 + 4 * (tid + k * NT));
  };
  const TileShare ts = pair_share(a.ntiles, a.share32);
  if (ts.first < ts.end) issue(ts.first);      // the first tile travels while the tables and fragments are set up

  // ---- once per workgroup: tables, weight fragments --------------------------------------------------------------------------
  // Every global value the prologue needs is REQUESTED first, then the images are cleared and the values are used: the
  // prologue was a chain of five dependent L2 round trips (table, forward table, weight scan, bound, weight fragments,
  // each behind a barrier: 10.8 us of a 120-150 us launch, tools/kernel_clock.py), now it is one.
  const int orow = mt * 32 + l31;
  constexpr int NTI = 8;                                   // table values per thread held in registers (more: a second pass)
  float tiv[KZ > 0 ? NTI : 1], tfv[EPI != 0 ? NTI : 1];
  float wraw[KB][8];                                       // B fragments of the transposed GEMM: B[k = c][n = o] = W[o][c]
  float bx = 0.f;
  if constexpr (NT3 == 2) bx = *a.xmax;                    // |x| <= bx; |gelu(x)| <= |x|
#pragma unroll
  for (int kb = 0; kb < KB; ++kb)
#pragma unroll
    for (int j = 0; j < 8; ++j) wraw[kb][j] = a.w[orow * C + kb * 16 + 8 * half + j];
  const int nti = KZ > 0 ? 2 * a.K2in * a.W : 0, ntf = EPI != 0 ? 16 * a.NJ * a.W : 0;
  if constexpr (KZ > 0) {
#pragma unroll
    for (int k = 0; k < NTI; ++k) tiv[k] = tid + k * NT < nti ? a.tinv[tid + k * NT] : 0.f;
  }
  if constexpr (EPI != 0) {
#pragma unroll
    for (int k = 0; k < NTI; ++k) tfv[k] = tid + k * NT < ntf ? a.tfwd[tid + k * NT] : 0.f;
  }
  auto put_tinv = [&](int i, float v) {
    const int k = i / a.W, w = i - k * a.W;
    unsigned short h, m, l;
    split3(v, h, m, l);
    unsigned short* d = reinterpret_cast<unsigned short*>(timg) + w * KZ * 16 + k;
    d[0] = h; d[TT / 2] = m; d[TT] = l;
  };
  if constexpr (KZ > 0) {
    for (int i = tid; i < 3 * (TT + ZT) / 4; i += NT) reinterpret_cast<unsigned*>(timg)[i] = 0u;   // k pads stay zero
    __syncthreads();
#pragma unroll
    for (int k = 0; k < NTI; ++k)
      if (tid + k * NT < nti) put_tinv(tid + k * NT, tiv[k]);
    for (int i = tid + NTI * NT; i < nti; i += NT) put_tinv(i, a.tinv[i]);
  }
  if constexpr (EPI != 0) {
#pragma unroll
    for (int k = 0; k < NTI; ++k) {
      const int i = tid + k * NT;
      if (i < ntf) tfwd_s[(i / a.W) * (a.W + 4) + i % a.W] = tfv[k];
    }
    for (int i = tid + NTI * NT; i < ntf; i += NT) tfwd_s[(i / a.W) * (a.W + 4) + i % a.W] = a.tfwd[i];
  }

  __shared__ __attribute__((aligned(16))) float lws[LIFT ? 5 * C : 4];
  if constexpr (LIFT) stage_lift_params<C>(lws, a.lw, a.lb, a.CL, tid, NT);
  // operand scales of the two-term fp16 GEMM (powers of two; 1 with three bf16 terms)
  float sx = 1.f, sw = 1.f;
  if constexpr (NT3 == 2) {
    __shared__ float red[NW];
    auto wg_max = [&](float m) {
#pragma unroll
      for (int o = 32; o >= 1; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
      __syncthreads();
      if (lane == 0) red[wave] = m;
      __syncthreads();
      float r = 0.f;
#pragma unroll
      for (int k = 0; k < NW; ++k) r = fmaxf(r, red[k]);
      return r;
    };
    // max |W|: the fragments of the workgroup's waves cover every element of W (row orow, all columns over the two halves)
    float mw = 0.f;
#pragma unroll
    for (int kb = 0; kb < KB; ++kb)
#pragma unroll
      for (int j = 0; j < 8; ++j) mw = fmaxf(mw, fabsf(wraw[kb][j]));
    sw = h2_scale(wg_max(mw));
    if constexpr (LIFT) {                                  // |u_0[c]| <= sum_k |lw[c][k]| bx + |lb[c]|
      float m = 0.f;
      for (int c = tid; c < C; c += NT) {
        const float4 wv = ld4(lws + 4 * c);                // (staged above; the first wg_max barrier made it visible)
        m = fmaxf(m, (fabsf(wv.x) + fabsf(wv.y) + fabsf(wv.z) + fabsf(wv.w)) * bx + fabsf(lws[4 * C + c]));
      }
      (void)wg_max(0.f);
      bx = wg_max(m);
      if (a.ubound && blockIdx.x == 0 && tid == 0) *a.ubound = bx;      // (the same value in every workgroup) for the backward pass
    }
    sx = h2_scale(bx);
  }
  const float inv_xw = 1.f / (sx * sw);
  bf16x8 wfrag[KB][NT3];
#pragma unroll
  for (int kb = 0; kb < KB; ++kb) split_n_x8<NT3>(wraw[kb], sw, wfrag[kb]);
  // With a row-DFT epilogue (EPI != 0) the accumulators are TRANSPOSED (lane <-> channel, registers <-> 4-pixel runs: one bias
  // register, 16-byte LDS writes of the tile).  Without one the only consumer is the store, and the plain orientation
  // (lane <-> pixel, registers <-> channel rows) leaves as whole 128-byte lines per wave half: 118 vs 127 us at config-2 size.
  constexpr bool TR = EPI != 0;
  const float bias_o = a.bias ? a.bias[orow] : 0.f;
  float bias_r[TR ? 1 : 16];
  if constexpr (!TR) {
#pragma unroll
    for (int r = 0; r < 16; ++r) bias_r[r] = a.bias ? a.bias[mt * 32 + 4 * half + (r & 3) + 8 * (r >> 2)] : 0.f;
  }

  __syncthreads();

  // one float4 of spectral rows = (o, re), (o, im), (o + 1, re), (o + 1, im) of row-mode rs -> k = 2 s, 2 s + 1 of two channels
  auto put_z = [&](int f, const float4& zq) {
    const int rs = (4 * f) / (2 * C), o = ((4 * f) % (2 * C)) >> 1;
    const int row = rs / a.K2in, s = rs - row * a.K2in;
    unsigned short h[4], m[4], l[4];
    split3(zq.x, h[0], m[0], l[0]); split3(zq.y, h[1], m[1], l[1]);
    split3(zq.z, h[2], m[2], l[2]); split3(zq.w, h[3], m[3], l[3]);
    unsigned char* d = zimg + ((row * C + o) * KZ * 16 + 2 * s) * 2;
    *reinterpret_cast<unsigned*>(d) = h[0] | ((unsigned)h[1] << 16);
    *reinterpret_cast<unsigned*>(d + KZ * 32) = h[2] | ((unsigned)h[3] << 16);
    *reinterpret_cast<unsigned*>(d + ZT) = m[0] | ((unsigned)m[1] << 16);
    *reinterpret_cast<unsigned*>(d + ZT + KZ * 32) = m[2] | ((unsigned)m[3] << 16);
    *reinterpret_cast<unsigned*>(d + 2 * ZT) = l[0] | ((unsigned)l[1] << 16);
    *reinterpret_cast<unsigned*>(d + 2 * ZT + KZ * 32) = l[2] | ((unsigned)l[3] << 16);
  };

  // per-lane offsets that do not change from tile to tile
  // output / addend: row orow (the descriptor starts at row mt * 32), pixels ng * 64 + 4 half ..; the (q, g) part of the
  // offset is a compile-time constant that lands in the instruction's immediate field.  soffset stays 0 on purpose: with
  // a REGISTER soffset the compiler assumes there is no "store data overwritten behind a > 8-byte store" hazard and pads
  // nothing, and on gfx950 that lost 5 % of the outputs when the next float4 was formed in the same registers.
  const int st_voff = (l31 * a.PW + 4 * half + ng * NTW * 32) * 4;

#ifdef FNO_ELIM      // phase elimination (tools/bf2_test.hip -DFNO_ELIM, timing only: results are wrong): bits of a.loose switch phases off
  const int elim = a.loose;
#else
  constexpr int elim = 0;
#endif
  int tslot = 0;
  float vmax = 0.f;          // max |u| stored by this thread (a.umax)
  FNO_TRACE_IF(true);
  FNO_CLK_BEGIN();
  for (int tile_ = ts.first; tile_ < ts.end; tile_ += ts.step) {
    const int tile = a.rev ? a.ntiles - 1 - tile_ : tile_;
    const int b = tile / a.tiles_per_plane;
    const int px0 = (tile % a.tiles_per_plane) * NPX;
    FNO_STAMP(tslot + 0);

    // ---- commit: (lifting |) GELU, split, pixel-major image; spectral rows -> image -----------------------------------------
#pragma unroll
    for (int i = 0; i < ITER; ++i) {
      const int cg = cg0 + CGS * i;
      float v[8];
      if constexpr (LIFT) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int c = cg * 8 + j;
          const float4 wv = ld4(lws + 4 * c);
          v[j] = fmaf(wv.w, xin[3], fmaf(wv.z, xin[2], fmaf(wv.y, xin[1], fmaf(wv.x, xin[0], lws[4 * C + c]))));
        }
      } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = pv[i][j];
        if constexpr (ACT_IN) { if (!(elim & 2)) gelu8(v, six, inf); }
      }
      bf16x8 f[NT3];
      split_n_x8<NT3>(v, sx, f);
      unsigned char* dst = xb + pimg_off<C>(opx, cg);
#pragma unroll
      for (int t = 0; t < NT3; ++t) *reinterpret_cast<bf16x8*>(dst + t * TERM) = f[t];
    }
#pragma unroll
    for (int k = 0; k < ZP; ++k)
      if (tid + k * NT < zc4) put_z(tid + k * NT, zpf[k]);
    for (int i = tid + ZP * NT; i < zc4; i += NT)        // still more spectral rows (short rows, many modes): straight from L2
      put_z(i, ld4(a.z + ((size_t)b * a.P + px0 / a.W) * a.K2in * C * 2 + 4 * i));
    FNO_STAMP(tslot + 1);
    __syncthreads();
    FNO_STAMP(tslot + 2);

    // ---- D^T[px][o] = sum_c act[px][c] W[o][c] + sum_k Tinv[k][w(px)] Z[row(px)][k][o] --------------------------------------
    // Software-pipelined over the k blocks: the fragments of block k + 1 are read into a SECOND register set while the
    // products of block k run (left to itself the compiler reads, waits for lgkmcnt(0) and multiplies: three exposed LDS
    // latencies per block, 5.8 k cycles per tile for 1.9 k cycles of MFMAs).  An earlier version reloaded each term's
    // registers in place right behind the term's last product; with two workgroups per CU that corrupted single 32 x 32
    // sub-tiles now and then (an MFMA queued behind the SIMD partner's MFMAs reads its operands later than it issues, and the
    // LDS data had already landed in them) - so the sets alternate.  sched_group_barrier pins the MFMA / DS-read interleaving.
    f32x16 acc[NTW];
    {
      auto frag = [&](const unsigned char* p) { return *reinterpret_cast<const bf16x8*>(p); };
      auto act_src = [&](int q, int kb) { return xb + pimg_off<C>((ng * NTW + q) * 32 + l31, 2 * kb + half); };
      bf16x8 fa[2][3];                                     // two sets of A-fragment terms: (h, m, l) or (h, l, -)
      {
        const unsigned char* s0 = act_src(0, 0);
#pragma unroll
        for (int t = 0; t < NT3; ++t) fa[0][t] = frag(s0 + t * TERM);
      }
      auto qtile = [&](auto qc, auto set0) {
        constexpr int q = decltype(qc)::value;
        constexpr int S0 = decltype(set0)::value;          // the register set that holds this tile's first fragment
        const int n0 = (ng * NTW + q) * 32;
        f32x16 hi, lo;                                     // hh products / cross terms of the split (fno_dev.h: mfma_x3s)
#pragma unroll
        for (int r = 0; r < 16; ++r) { hi[r] = 0.0f; lo[r] = 0.0f; }
        const unsigned char* tsrc = timg + ((n0 % a.W + l31) * KZ * 16 + 8 * half) * 2;
        const unsigned char* zsrc = zimg + (((n0 / a.W) * C + orow) * KZ * 16 + 8 * half) * 2;
        bf16x8 zb[KZ > 0 ? KZ : 1][3];                      // B fragments of the extension blocks: this tile's spectral rows
#pragma unroll
        for (int kz = 0; kz < KZ; ++kz)
#pragma unroll
          for (int t = 0; t < 3; ++t) zb[kz][t] = frag(zsrc + t * ZT + kz * 32);
        // One k block with NB terms per operand: A from register set S, channel-side fragments b[]; meanwhile the NEXT
        // block's fragment (NN terms at `nx`, plane stride `ns`; NN = 0: nothing follows) is read into set 1 - S.
        auto block = [&](auto sc, auto nbc, auto nnc, const bf16x8* b, const unsigned char* nx, int ns) {
          constexpr int S = decltype(sc)::value, NB = decltype(nbc)::value, NN = decltype(nnc)::value;
          // (pixel-side fragment x, channel-side fragment y) -> D^T[px][o] (TR) or D[o][px]: the same registers, swapped operands
          auto mm = [&](const bf16x8& x, const bf16x8& y, const f32x16& c) {
            if constexpr (NB == 3) {
              if constexpr (TR) return __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, c, 0, 0, 0);
              else return __builtin_amdgcn_mfma_f32_32x32x16_bf16(y, x, c, 0, 0, 0);
            } else {
              const f16x8 xh = __builtin_bit_cast(f16x8, x), yh = __builtin_bit_cast(f16x8, y);
              if constexpr (TR) return __builtin_amdgcn_mfma_f32_32x32x16_f16(xh, yh, c, 0, 0, 0);
              else return __builtin_amdgcn_mfma_f32_32x32x16_f16(yh, xh, c, 0, 0, 0);
            }
          };
          const bf16x8 (&cur)[3] = fa[S];
          bf16x8 (&nxt)[3] = fa[1 - S];
          if constexpr (NB == 3) {
            lo = mm(cur[2], b[0], lo);
            if constexpr (NN >= 1) nxt[0] = frag(nx);
            lo = mm(cur[1], b[1], lo);
            lo = mm(cur[1], b[0], lo);
            if constexpr (NN >= 2) nxt[1] = frag(nx + ns);
            lo = mm(cur[0], b[2], lo);
            lo = mm(cur[0], b[1], lo);
            hi = mm(cur[0], b[0], hi);
            if constexpr (NN == 3) nxt[2] = frag(nx + 2 * ns);
            if constexpr (NN == 3) {
              __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
              __builtin_amdgcn_sched_group_barrier(0x008, 2, 0); __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
              __builtin_amdgcn_sched_group_barrier(0x008, 3, 0); __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            }
          } else {
            lo = mm(cur[1], b[0], lo);
            if constexpr (NN >= 1) nxt[0] = frag(nx);
            lo = mm(cur[0], b[1], lo);
            if constexpr (NN >= 2) nxt[1] = frag(nx + ns);
            hi = mm(cur[0], b[0], hi);
            if constexpr (NN == 3) nxt[2] = frag(nx + 2 * ns);
            if constexpr (NN == 2) {
              __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
              __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
              __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            }
          }
        };
        using I0 = std::integral_constant<int, 0>;
        using I1 = std::integral_constant<int, 1>;
        using I3 = std::integral_constant<int, 3>;
        using IT = std::integral_constant<int, NT3>;
        constexpr bool LASTQ = q + 1 == NTW;
        const unsigned char* qnext = act_src(LASTQ ? q : q + 1, 0);
        // block j of this tile (j = 0 .. KB + KZ - 1) reads set (S0 + j) % 2
        auto run = [&](auto jc, auto nbc, auto nnc, const bf16x8* b, const unsigned char* nx, int ns) {
          constexpr int J = decltype(jc)::value;
          if constexpr ((S0 + J) % 2 == 0) block(I0{}, nbc, nnc, b, nx, ns);
          else block(I1{}, nbc, nnc, b, nx, ns);
        };
        auto act_blocks = [&](auto self, auto kbc) {
          constexpr int kb = decltype(kbc)::value;
          if constexpr (kb < KB) {
            using J = std::integral_constant<int, kb>;
            if constexpr (kb + 1 < KB) run(J{}, IT{}, IT{}, wfrag[kb], act_src(q, kb + 1), TERM);
            else if constexpr (KZ > 0) run(J{}, IT{}, I3{}, wfrag[kb], tsrc, TT);
            else if constexpr (!LASTQ) run(J{}, IT{}, IT{}, wfrag[kb], qnext, TERM);
            else run(J{}, IT{}, I0{}, wfrag[kb], nullptr, 0);
            self(self, std::integral_constant<int, kb + 1>{});
          }
        };
        act_blocks(act_blocks, I0{});
        if constexpr (NT3 == 2) {      // undo the operand scales (exact) before the unscaled extension products are added
#pragma unroll
          for (int r = 0; r < 16; ++r) { hi[r] = (hi[r] + lo[r]) * inv_xw; lo[r] = 0.f; }
        }
        auto ext_blocks = [&](auto self, auto kzc) {
          constexpr int kz = decltype(kzc)::value;
          if constexpr (kz < KZ) {
            using J = std::integral_constant<int, KB + kz>;
            if constexpr (kz + 1 < KZ) run(J{}, I3{}, I3{}, zb[kz], tsrc + (kz + 1) * 32, TT);
            else if constexpr (!LASTQ) run(J{}, I3{}, IT{}, zb[kz], qnext, TERM);
            else run(J{}, I3{}, I0{}, zb[kz], nullptr, 0);
            self(self, std::integral_constant<int, kz + 1>{});
          }
        };
        ext_blocks(ext_blocks, I0{});
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[q][r] = hi[r] + lo[r];
      };
      static_assert(NTW == 2, "two 32-pixel column tiles per wave");
      if (!(elim & 1)) {
        qtile(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{});
        qtile(std::integral_constant<int, 1>{}, std::integral_constant<int, (KB + KZ) % 2>{});      // q = 0 left its successor's first fragment there
      } else {
#pragma unroll
        for (int q = 0; q < NTW; ++q)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[q][r] = (float)fa[0][0][r & 3];
      }
    }
    // the next tile's loads go out behind the GEMM (their 32 registers must not be live beside the weight fragments, the
    // accumulators and the fragment double buffer); epilogue, row DFT and the other workgroup's phases cover their latency
    if (tile_ + ts.step < ts.end) issue(tile_ + ts.step);
    FNO_STAMP(tslot + 3);
    __syncthreads();        // every wave is done with the images (the fp32 output tile reuses them)
    FNO_STAMP(tslot + 4);

    // ---- epilogue: bias (+ addend), store, activation -> fp32 tile -------------------------------------------------------------
    {
      const size_t obase = ((size_t)b * C + mt * 32) * a.PW + px0;
      const unsigned obytes = 31u * PWb + NPX * 4;
      const __amdgpu_buffer_rsrc_t ru = make_rsrc(a.u ? a.u + obase : nullptr, a.u ? obytes : 0u);
      const __amdgpu_buffer_rsrc_t ra = make_rsrc(ADD ? a.add + obase : nullptr, ADD ? obytes : 0u);
      if constexpr (!TR) {
        // plain orientation: acc[q][r] = channel mt * 32 + 4 half + (r & 3) + 8 (r >> 2), pixel (ng * 2 + q) * 32 + l31:
        // dword stores, a whole 128-byte line per wave half; the row offset rides in the scalar offset
        const int vo = (4 * half * a.PW + ng * NTW * 32 + l31) * 4;
#pragma unroll
        for (int q = 0; q < NTW; ++q) {
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const unsigned so = (unsigned)((r & 3) + 8 * (r >> 2)) * PWb + q * 128u;
            float v = acc[q][r] + bias_r[r];
            if constexpr (ADD) v += buf_ld1(ra, vo, so);
            if constexpr (RELU) v = v < 0.f ? 0.f : v;      // (NaN stays NaN, as torch's relu)
            if (a.u) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), ru, vo, so, 0);
            if (a.umax) vmax = fmaxf(vmax, fabsf(v));
          }
        }
      } else {
      // (soffset 0 + per-lane voffset: see st_voff - the compiler then pads the store-data hazard itself)
#pragma unroll
      for (int q = 0; q < NTW; ++q) {
        float* xp = xs + orow * PITCH + (ng * NTW + q) * 32 + 4 * half;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          float4 v = make_float4(acc[q][4 * g] + bias_o, acc[q][4 * g + 1] + bias_o, acc[q][4 * g + 2] + bias_o,
                                 acc[q][4 * g + 3] + bias_o);
          if constexpr (ADD) {
            const float4 ad = buf_ld4(ra, st_voff + (q * 32 + 8 * g) * 4, 0);
            v.x += ad.x; v.y += ad.y; v.z += ad.z; v.w += ad.w;
          }
          if constexpr (RELU) {      // (NaN stays NaN, as torch's relu)
            v.x = v.x < 0.f ? 0.f : v.x; v.y = v.y < 0.f ? 0.f : v.y; v.z = v.z < 0.f ? 0.f : v.z; v.w = v.w < 0.f ? 0.f : v.w;
          }
          if (a.u && !(elim & 16)) buf_st4(ru, st_voff + (q * 32 + 8 * g) * 4, 0, v);
          if (a.umax) vmax = fmaxf(fmaxf(vmax, fabsf(v.x)), fmaxf(fmaxf(fabsf(v.y), fabsf(v.z)), fabsf(v.w)));
          if constexpr (EPI != 0) {
            if constexpr (EPI == 2) { if (!(elim & 4)) v = gelu4(v, six, inf); }
            st4(xp + 8 * g, v);
          }
        }
      }
      }
    }
    FNO_STAMP(tslot + 5);
    if constexpr (EPI != 0) {
      __syncthreads();
      FNO_STAMP(tslot + 6);
      if (!(elim & 8)) row_dft_epilogue<C, NPX, NW>(xs, tfwd_s, a.W + 4, a.x1, b, px0, a.P, a.W, a.K2out, a.NJ, wave, lane);
      FNO_STAMP(tslot + 7);
      __syncthreads();      // the next commit rewrites the images under the tile
    }
    tslot += 8;
  }
  FNO_CLK_END(0);
  if (a.umax) absmax_publish(vmax, a.umax);
}
