// Projection MLP with the channel GEMMs as TWO fp16 terms and THREE products (fno_dev.h: "h2") instead of three bf16 terms
// and six products: the same kernels as k_proj_fwd_x3 (k_projection.h) and k_proj_bwd_t (k_projection2.h) - tiling,
// phases, partial-slab outputs - with half the matrix-pipe work, two instead of three LDS planes per operand and 3
// instead of 4.5 vector instructions per split element.  Reference semantics: neuralop/models/tfno.py:23-38 and its autograd.
// Operand scales: the activation's bound comes from the kernel that stored it (PwFwdArgs.umax -> ProjFwdArgs.xmax), the
// weights are scanned by the kernel that splits them, dP1 = gelu'(P1) w2 dy is bounded by 1.13 max|w2| max|dy|.
#pragma once
#include "fno_dev.h"
#include "k_projection.h"
#include "k_projection2.h"

// Pixel-major two-term image: xb[t][px][C + 8 halfs] (rows 16-B aligned, b128 reads with lanes <-> pixels conflict-free),
// register-staged prefetch as SplitTilePrefetch (fno_dev.h)
template <int NPX, int NT, int C>
struct SplitTilePrefetchH2 {
  static constexpr int PBH = C + 8;
  static constexpr int ITEMS = NPX * (C / 8);
  static constexpr int ITER = (ITEMS + NT - 1) / NT;
  static constexpr int TERM = NPX * PBH;             // halfs per term array
  float v[ITER][8];
  FNO_DEV void issue(const float* src, size_t row_stride, int tid) {
#pragma unroll
    for (int i = 0; i < ITER; ++i) {
      const int idx = tid + i * NT;
      const int px = idx % NPX, cg = idx / NPX;
#pragma unroll
      for (int j = 0; j < 8; ++j)
        v[i][j] = (ITEMS % NT == 0 || idx < ITEMS) ? src[(size_t)(cg * 8 + j) * row_stride + px] : 0.f;
    }
  }
  FNO_DEV void commit(unsigned short* xb, bool act, float scale, float six, float inf, int tid) {
#pragma unroll
    for (int i = 0; i < ITER; ++i) {
      const int idx = tid + i * NT;
      const int px = idx % NPX, cg = idx / NPX;
      if (act) gelu8(v[i], six, inf);
      f16x8 h, l;
      split2x8(v[i], scale, h, l);
      if (ITEMS % NT == 0 || idx < ITEMS) {
        unsigned short* dst = xb + px * PBH + cg * 8;
        *reinterpret_cast<f16x8*>(dst) = h;
        *reinterpret_cast<f16x8*>(dst + TERM) = l;
      }
    }
  }
};

// max |w| over n values, by every thread of the workgroup (LDS scratch: one float per wave)
template <int NT>
FNO_DEV float wg_absmax(const float* __restrict__ w, int n, float* scratch, int tid) {
  float m = 0.f;
  for (int i = tid; i < n; i += NT) m = fmaxf(m, fabsf(w[i]));
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
  if ((tid & 63) == 0) scratch[tid >> 6] = m;
  __syncthreads();
  float r = 0.f;
  for (int k = 0; k < NT / 64; ++k) r = fmaxf(r, scratch[k]);
  __syncthreads();
  return r;
}


// ---------------------------------------------------------------------------------------------------------------------------
// Projection forward, third generation ("w": independent waves; round 4).  tools/occupancy_valu_test.hip measured what bounds
// the fused kernels: one SIMD issues 0.14-0.19 vector instructions per cycle from ONE wave, 0.23-0.28 from two, 0.30-0.37
// from four and 0.51-0.59 from eight - and beside a wave that keeps the matrix pipe busy only 0.09-0.11 / 0.15-0.18 / 0.23-0.27
// from one / two / four vector waves (profiles/r04_valu_rate_vs_waves_per_simd.txt).  The 8-wave kernels with one workgroup
// per CU (two waves per SIMD, 160-250 VGPRs) therefore run their GELU-heavy vector phases at a third of the SIMD's issue rate,
// whatever their structure.  This kernel is built for SIX to EIGHT waves per SIMD instead (two 12- / 16-wave workgroups per CU,
// 80 / 64 VGPRs, 67 / 34 KB of LDS each) and has NO barrier in its main loop:
//   * a wave owns a COLUMN of 32 consecutive pixels for all channels and all hidden rows; it loads the column straight from HBM
//     into MFMA operand layout (lane = (pixel, k half), eight channels 16 kb + 8 half + j in eight registers: every load
//     instruction moves two whole 128-byte lines), applies GELU and splits in registers - no LDS round trip, no commit barrier;
//   * only the weights live in LDS (two fp16 terms of W1, split once per workgroup), read as A fragments by every wave;
//   * per chunk of 32 hidden rows: 12 fp16 MFMAs into (hh, cross) accumulators, then bias + GELU + the w2 dot on the 16
//     accumulator values of a lane; the waves of a SIMD drift apart on their own, so one wave's matrix burst runs under the
//     others' vector work.
// Same arithmetic as k_proj_fwd_h2 (bitwise: same products, same order); reference semantics neuralop/models/tfno.py:23-38.
#ifdef PFW_TRACE
__device__ unsigned long long g_pfw_trace[64 * 16 * 4];
#endif
// Two 12-wave workgroups per CU (six waves per SIMD, <= 80 VGPRs), static shares of the columns.  Measured on the way
// (profiles/r04_proj_fwd_w_workgroup_times.txt, tools/kernel_clock.py): the workgroup that arrives second on a CU loses the
// vector-issue arbitration to the older one (median 83 us vs 137 us for workgroups [0, 256) / [256, 512) of one launch, both
// started within 5 us) - alternating the issue priority per column in opposite phase evens that out (107 / 137 us) but the
// launch ends at the same time (150 us from first start to last end: the SIMDs' issue capacity, not the split, bounds it;
// the old 8-wave tile kernel: 176 us); ONE 16-wave workgroup per CU with an LDS work queue is balanced but slower (159-174 us:
// four waves per SIMD); eight global work queues (returning atomics on eight addresses) serialise: 465 us.  The chip holds
// 1.9-2.0 GHz inside this kernel where the 8-wave kernels hold 2.3-2.4 (denser issue, MI355X_MICROARCH.md DVFS item 4).
// Round 5: this is the one kernel of the engine whose waves are independent (six per SIMD, no barrier), so at any moment some
// of a SIMD's waves are in their products and some in their vector phase - and on gfx950 SCALAR fp32 vector instructions run
// beside another wave's products while packed ones serialise with them.  With the hidden activation
// (gelu_s: 11 instructions per value against 7.5 packed) and the column's own activation + split in scalar form the launch takes
// 0.1606 ms against 0.1742 (level 1, hidden activation only: 0.1679; six interleaved runs each on one box, +-0.0005) - more
// instructions, hidden behind the matrix pipe.  Same operations in the same order: bit-identical results.  A MIX of the forms
// (three or two of a chunk's four groups of hidden values scalar, the rest packed) is slower than all-scalar: 0.1678 / 0.1690
// against 0.1646 ms on one box - every packed instruction holds the matrix pipe off.  Issue priorities (none, or raised
// around the products instead of alternating per column): 0.1654 / 0.1674 against 0.1650.
#ifndef FNO_PFW_SCALAR_ACT
#define FNO_PFW_SCALAR_ACT 2      // 0 = packed forms (A/B arm), 1 = scalar hidden activation, 2 = + scalar input activation and split
#endif
template <int C, int HID, int NWAVE>
__global__ void __launch_bounds__(NWAVE * 64, NWAVE / 2) k_proj_fwd_w(ProjFwdArgs a) {
  FNO_CLK_ENTRY();
  constexpr int KB = C / 16, NCHK = HID / 32, NT = NWAVE * 64;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  unsigned short* w1b = reinterpret_cast<unsigned short*>(smem);          // [HID/32][KB][2 terms][64 lanes][8 halfs]
  float* b1s = reinterpret_cast<float*>(w1b + (size_t)NCHK * KB * 2 * 64 * 8);
  float* w2s = b1s + HID;
  float* scratch = w2s + HID;                                              // NWAVE floats
  int* colq = reinterpret_cast<int*>(scratch + 16);                        // the workgroup's column counter
  float gk_six, gk_inf;
  gelu_consts(gk_six, gk_inf);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, half = lane >> 5;
  FNO_CLK_BEGIN();

#ifdef PFW_TRACE
  const unsigned long long tr_k0 = __builtin_readcyclecounter();
#endif
  // ONE pass over W1: a thread fetches its fragment items (hidden 32-block, k block, lane: eight consecutive channels of one
  // row), the workgroup maximum of |W1| comes from the same registers (the items cover W1 exactly once), then they are split.
  // (Was a scan followed by a second, dependent read of the same 64 KB: 21 k cycles of prologue per workgroup.)
  constexpr int NIT = (NCHK * KB * 64 + NT - 1) / NT;
  const float bxv = *a.xmax;
  float wv[NIT][8];
  float mw = 0.f;
#pragma unroll
  for (int q = 0; q < NIT; ++q) {
    const int it = tid + q * NT;
    const int ln = it & 63, kb = (it >> 6) % KB, mt = (it >> 6) / KB;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      wv[q][j] = it < NCHK * KB * 64 ? a.w1[(size_t)(mt * 32 + (ln & 31)) * C + kb * 16 + 8 * (ln >> 5) + j] : 0.f;
      mw = fmaxf(mw, fabsf(wv[q][j]));
    }
  }
  for (int i = tid; i < HID; i += NT) { b1s[i] = a.b1[i]; w2s[i] = a.w2[i]; }
  if (tid == 0) *colq = 0;
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) mw = fmaxf(mw, __shfl_xor(mw, o, 64));
  if (lane == 0) scratch[wave] = mw;
  __syncthreads();
  mw = 0.f;
#pragma unroll
  for (int k = 0; k < NWAVE; ++k) mw = fmaxf(mw, scratch[k]);
  const float sx = h2_scale(bxv), sw = h2_scale(mw);
#ifdef PFW_TRACE
  const unsigned long long tr_k1 = __builtin_readcyclecounter();
#endif
  const float inv = 1.0f / (sx * sw);
#pragma unroll
  for (int q = 0; q < NIT; ++q) {
    const int it = tid + q * NT;
    if (it < NCHK * KB * 64) {
      const int ln = it & 63, kb = (it >> 6) % KB, mt = (it >> 6) / KB;
      f16x8 h, l;
      split2x8(wv[q], sw, h, l);
      unsigned short* dst = w1b + ((size_t)((mt * KB + kb) * 2) * 64 + ln) * 8;
      *reinterpret_cast<f16x8*>(dst) = h;
      *reinterpret_cast<f16x8*>(dst + 64 * 8) = l;
    }
  }
  __syncthreads();

  const int cols_per_plane = a.PW / 32, ncols = a.ntiles * 4;
  const unsigned PWb = (unsigned)a.PW * 4u;
#ifdef PFW_TRACE      // diagnostic build: where a wave's cycles go (stamps cost an s_waitcnt lgkmcnt(0) each)
  unsigned long long tr_load = 0, tr_mfma = 0, tr_valu = 0, tr_cols = 0, tr_t = __builtin_readcyclecounter(), tr_n;
  const unsigned long long tr_k2 = tr_t;
#define PFW_STAMP(acc) do { tr_n = __builtin_readcyclecounter(); acc += tr_n - tr_t; tr_t = tr_n; } while (0)
#else
#define PFW_STAMP(acc) do { } while (0)
#endif
  const int voff = (8 * half * a.PW + l31) * 4;
  // A workgroup owns a static share of the columns (pair_share: the one dispatched first on its CU the larger one) and its
  // waves POP them from a counter in LDS: the waves of a SIMD are served oldest first, so with one static share per wave the
  // youngest wave of the launch finished its 5-6 columns 35 us after the median one (kernel 192 us, wave 0 of the workgroups
  // done at 103 / 125 / 158 us at the 10 / 50 / 100 % quantiles; profiles/r04_two_workgroups_per_cu_tail.txt).  The issue
  // priority alternates from column to column, in opposite phase for the two workgroups of a CU.
  const TileShare ts = pair_share(ncols, a.share32);
  const int nmine = ts.first < ts.end ? (ts.end - ts.first + ts.step - 1) / ts.step : 0;
  int kcol = (2 * (int)blockIdx.x >= (int)gridDim.x) ? 1 : 0;
  for (;; ++kcol) {
    int jq = 0;
    if (lane == 0) jq = __hip_atomic_fetch_add(colq, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    jq = __builtin_amdgcn_readfirstlane(jq);
    if (jq >= nmine) break;
    const int col = ts.first + ts.step * jq;
    if (kcol & 1) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(0);
    const int b = col / cols_per_plane;
    const int px0 = (col - b * cols_per_plane) * 32;
    const __amdgpu_buffer_rsrc_t rx = make_rsrc(a.x + (size_t)b * C * a.PW + px0, (unsigned)(C - 1) * PWb + 128u);
    // B[k = c][n = px] fragments of the column: lane (px = l31, half) holds channels 16 kb + 8 half + j
    f16x8 bfrag[KB][2];
#pragma unroll
    for (int kb = 0; kb < KB; ++kb) {
      float v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = buf_ld1(rx, voff, (unsigned)(16 * kb + j) * PWb);      // (default policy: the projection backward re-reads u_L, last part first)
#if FNO_PFW_SCALAR_ACT >= 2      // the column's own activation and split with scalar instructions too
      if (a.act_in) {
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = gelu_s(v[j], gk_six, gk_inf);
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float sv = v[j] * sx;
        const _Float16 hh = (_Float16)sv;
        bfrag[kb][0][j] = hh; bfrag[kb][1][j] = (_Float16)(sv - (float)hh);
      }
#else
      if (a.act_in) gelu8(v, gk_six, gk_inf);
      split2x8(v, sx, bfrag[kb][0], bfrag[kb][1]);
#endif
    }
#ifdef PFW_TRACE
    asm volatile("" :: "v"(bfrag[KB - 1][1]));
#endif
    PFW_STAMP(tr_load);
    float ysum = 0.f;
#pragma unroll 1
    for (int ch = 0; ch < NCHK; ++ch) {
      f32x16 acc, lo;      // hh products / cross terms
#pragma unroll
      for (int r = 0; r < 16; ++r) { acc[r] = 0.f; lo[r] = 0.f; }
      const unsigned short* wa = w1b + ((size_t)(ch * KB * 2) * 64 + lane) * 8;
#pragma unroll
      for (int kb = 0; kb < KB; ++kb) {
        f16x8 af[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) af[t] = *reinterpret_cast<const f16x8*>(wa + (size_t)(kb * 2 + t) * 64 * 8);
        mfma_h2s(af, bfrag[kb], acc, lo);
      }
#ifdef PFW_TRACE
      asm volatile("s_nop 11" :: "v"(acc), "v"(lo));
#endif
      PFW_STAMP(tr_mfma);
      // D[row = hidden 32 ch + (r & 3) + 8 (r >> 2) + 4 half][col = pixel l31]
      const float* b1p = b1s + ch * 32 + 4 * half;
      const float* w2p = w2s + ch * 32 + 4 * half;
#if FNO_PFW_SCALAR_ACT
#pragma unroll
      for (int q = 0; q < 4; ++q) {            // four values at a time (the scalar form's temporaries: 80 registers per wave)
        float hv[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const int r = 4 * q + k;
          hv[k] = gelu_s(fmaf(acc[r] + lo[r], inv, b1p[(r & 3) + 8 * (r >> 2)]), gk_six, gk_inf);
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const int r = 4 * q + k;
          ysum = fmaf(w2p[(r & 3) + 8 * (r >> 2)], hv[k], ysum);
        }
      }
#else
#pragma unroll
      for (int q = 0; q < 2; ++q) {            // eight values at a time: half the live temporaries of the packed GELU
        f32x2 hp[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const int r = 8 * q + 2 * k;
          hp[k][0] = fmaf(acc[r] + lo[r], inv, b1p[(r & 3) + 8 * (r >> 2)]);
          hp[k][1] = fmaf(acc[r + 1] + lo[r + 1], inv, b1p[((r + 1) & 3) + 8 * ((r + 1) >> 2)]);
        }
        gelu_pairs<4>(hp, gk_six, gk_inf);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const int r = 8 * q + 2 * k;
          ysum = fmaf(w2p[(r & 3) + 8 * (r >> 2)], hp[k][0], ysum);
          ysum = fmaf(w2p[((r + 1) & 3) + 8 * ((r + 1) >> 2)], hp[k][1], ysum);
        }
      }
#endif
#ifdef PFW_TRACE
      asm volatile("" :: "v"(ysum));
#endif
      PFW_STAMP(tr_valu);
    }
    ysum += __shfl_xor(ysum, 32, 64);
    if (half == 0) a.y[(size_t)b * a.PW + px0 + l31] = ysum + a.b2[0];      // (a scalar load per column: held in a register across the
                                                                              // column loop it was the kernel's one spill)
#ifdef PFW_TRACE
    ++tr_cols;
#endif
  }
  FNO_CLK_END(2);
#ifdef PFW_TRACE
  if (lane == 0 && blockIdx.x < 64) {
    unsigned long long* q = g_pfw_trace + (blockIdx.x * NWAVE + wave) * 4;
    q[0] = tr_load; q[1] = tr_mfma; q[2] = tr_valu; q[3] = tr_cols | ((tr_k1 - tr_k0) << 16) | ((tr_k2 - tr_k1) << 40);
  }
#endif
}
static inline size_t proj_fwd_w_lds(int C, int HID) { return (size_t)(HID / 32) * (C / 16) * 2 * 64 * 16 + (size_t)(2 * HID + 16 + 4) * 4; }
