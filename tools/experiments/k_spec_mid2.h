// NOT part of the product library (round 6 experiment, measured and not adopted): k_spec_mid with two samples per workgroup.
// Dropped into csrc/k_spectral_mid.h behind k_spec_mid and launched from spec_mid_t (fno_abi.hip) as
//   launch("k_spec_mid", k_spec_mid2<12, 64>, dim3(inner / 64, (samples + 1) / 2), dim3(64, 16), 2 * lds, st, x1, hat, wm, z, twT, twi,
//          n, inner, K2, conj_w, Bm, w_ms, samples)      // when C == 64, samples >= 2, (Bm % 2 == 0 || Bm >= samples)
// it passes tests/test_parity_gpu.py (190 cases) at 122 VGPRs / 0 scratch / 120 KB LDS.  BASELINE config 2, four interleaved pairs
// on one box: 25.5-25.9 us per launch against 27.0-27.1 (the weight rows are fetched once per pair of samples: half the L2 -> L1
// traffic, every CU at most one workgroup) - and the STEP is unchanged (2.1003 vs 2.0922 ms mean, the other kernels 0.2-0.5 %
// slower beside it).  5 % of the launch is what the weight stream was worth; the rest is the chain of dependent latencies
// (DESIGN.md section 4g): VERDICT r05 item 3's 18 us are not behind this restructuring.
// Two samples per workgroup: 1024 threads = two 8-wave halves.  Each half runs phases 1 and 3 for its own
// sample exactly as k_spec_mid does; in the contraction a half takes every second group of modes for BOTH samples, so a weight
// row is fetched once per pair of samples (half the L2 -> L1 traffic of the launch: the weights are its largest stream) and
// every CU holds at most one workgroup - the launch is no longer set by the CUs that hold two.  grid (inner / 64, ceil(samples / 2)),
// block (64, 16), LDS 2 x (8 + 2) x NK x 64 float2.  Both samples of a workgroup belong to one member (Bm even or one member).
template <int NK, int C>
__global__ void __launch_bounds__(1024, 1) k_spec_mid2(const float2* __restrict__ x1, float2* __restrict__ hat,
                                                     const float2* __restrict__ wm, float2* __restrict__ z,
                                                     const float2* __restrict__ twT, const float2* __restrict__ twi, int n,
                                                     int inner, int K2, int conj_w, int Bm, size_t w_ms, int samples) {
  static_assert(C == 32 || C == 64, "whole bins per workgroup");
  static_assert(FNO_MID_SMEM_TABLE && NK <= 12, "table rows in scalar registers");
  constexpr bool SREG = true;
  constexpr int SEGS = 8, JW = C / SEGS, NTH = 64 * SEGS;
  constexpr int RB = 2, NG = NK / RB;               // RB modes' weight rows per load group, double-buffered
  static_assert(NK % (2 * RB) == 0, "mode groups, an even number of them");
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float2* sh2 = reinterpret_cast<float2*>(smem);    // [2 samples][SEGS][NK][64] partial sums (phases 1 and 2)
  float2* ys2 = sh2 + 2 * SEGS * NK * 64;           // [2][NK][64] truncated spectra of the inputs
  float2* os2 = ys2 + 2 * NK * 64;                  // [2][NK][64] contracted spectra
  const int ql = threadIdx.x;
  const int yy = __builtin_amdgcn_readfirstlane(threadIdx.y);
  const int seg = yy & (SEGS - 1), sp = yy >> 3;    // sp: which of the workgroup's two samples this half owns in phases 1 and 3
  const int q = blockIdx.x * 64 + ql;
  const bool live = 2 * (int)blockIdx.y + sp < samples;      // (an odd sample count: the last workgroup's second half idles)
  const int o = live ? 2 * blockIdx.y + sp : samples - 1;
  float2* sh = sh2 + sp * SEGS * NK * 64;
  float2* ys = ys2 + sp * NK * 64;
  float2* os = os2 + sp * NK * 64;
  const int k2 = q / C, ch = q - k2 * C, lb = (ql / C) * C;     // lb: first lane of this lane's bin
  const float2* wb = wm + (size_t)((2 * blockIdx.y) / Bm) * w_ms + ((size_t)k2 * C + seg * JW) * C + ch;      // (both samples: one member)
  MID_STAMP(0);
  // Table rows are the same for every lane.  Round 2 staged them in LDS (through the scalar cache, one row at a time, a wave
  // waited ~1000 cycles per row); round 5 found what that costs: a broadcast ds_read_b64 still occupies the LDS unit for four
  // cycles, 192 of them per wave and phase = 12 k cycles per phase on the CUs that hold two workgroups - the phases were bound
  // by LDS issue, which is why halving their vector instructions alone changed nothing.  Now a row's NK entries come through
  // scalar loads one row AHEAD of their use (2 x NK scalar register pairs), and the multiply-adds are packed.
  // the first weight group does not depend on phase 1: in flight from here on
  float2 wv[2][RB][JW];
  auto load_w = [&](int g, int buf) {
#pragma unroll
    for (int rr = 0; rr < RB; ++rr)
#pragma unroll
      for (int j = 0; j < JW; ++j) wv[buf][rr][j] = wb[((size_t)(g * RB + rr) * K2 * C + j) * C];
  };
  if (!(FNO_MID_SKIP & 1)) load_w(sp, 0);      // this half's groups: sp, sp + 2, ...
  // ---- phase 1 ----
  {
    f32x2 acc[NK];
#pragma unroll
    for (int r = 0; r < NK; ++r) acc[r] = f32x2{0.f, 0.f};
    const float2* src = x1 + (size_t)o * n * inner + q;
    // 16 rows per lane in flight (two batches of 8) before the first use: one HBM round trip for n <= 128
    constexpr int NBR = 16;
    bool first = true;
    for (int nb = seg; nb < n; nb += NBR * SEGS) {
      float2 v[NBR];
#pragma unroll
      for (int j = 0; j < NBR; ++j) {
        const int nn = nb + SEGS * j;
        v[j] = (nn < n && live && !(FNO_MID_SKIP & 2)) ? src[(size_t)nn * inner] : make_float2(0.f, 0.f);
      }
      if constexpr (SREG) {
      (void)first;
      // (GUARD: whether rows beyond n exist in this batch.  Without the per-row branch the batch is ONE basic block and the
      // scalar loads of row j + 1 stay in flight under row j's multiply-adds; with it every row's loads are waited for at the
      // block boundary in front of them)
      auto rows = [&](auto guard) {
        constexpr bool GUARD = decltype(guard)::value;
        float2 tw[2][NK];                                 // row nb and the next one: uniform addresses, scalar loads
#pragma unroll
        for (int r = 0; r < NK; ++r) tw[0][r] = twT[(size_t)min(nb, n - 1) * NK + r];
#pragma unroll
        for (int j = 0; j < NBR; ++j) {
          const int nn = nb + SEGS * j;
          if (j + 1 < NBR) {
#pragma unroll
            for (int r = 0; r < NK; ++r) tw[(j + 1) & 1][r] = twT[(size_t)(GUARD ? min(nn + SEGS, n - 1) : nn + SEGS) * NK + r];
          }
          if (!GUARD || nn < n) {
            const f32x2 vv = {v[j].x, v[j].y}, vs = natural_pair(-v[j].y, v[j].x);
#pragma unroll
            for (int r = 0; r < NK; ++r) cfma_uniform<true>(acc[r], tw[j & 1][r], vv, vs);
          }
        }
      };
      if (nb + SEGS * (NBR - 1) < n) rows(MidFlag<false>{}); else rows(MidFlag<true>{});
      } else {
      if (first) { __syncthreads(); first = false; }      // the table is staged (n >= 1: every wave passes here once)
#pragma unroll
      for (int j = 0; j < NBR; ++j) {
        const int nn = nb + SEGS * j;
        if (nn < n) {
          const float2* t = sh + nn * NK;
          const f32x2 vv = {v[j].x, v[j].y}, vs = natural_pair(-v[j].y, v[j].x);
#pragma unroll
          for (int r = 0; r < NK; ++r) cfma_uniform<false>(acc[r], t[r], vv, vs);
        }
      }
      }
    }
    if constexpr (!SREG) { if (first) __syncthreads(); }
    MID_STAMP(1);
    if constexpr (!SREG) __syncthreads();               // every wave is done reading the table
#pragma unroll
    for (int r = 0; r < NK; ++r) sh[(seg * NK + r) * 64 + ql] = make_float2(acc[r][0], acc[r][1]);
  }
  __syncthreads();
  MID_STAMP(2);
  for (int r = seg; r < NK; r += SEGS) {
    float sx = 0.f, sy = 0.f;
#pragma unroll
    for (int k = 0; k < SEGS; ++k) { sx += sh[(k * NK + r) * 64 + ql].x; sy += sh[(k * NK + r) * 64 + ql].y; }
    ys[r * 64 + ql] = make_float2(sx, sy);
    if (hat && live) hat[((size_t)o * NK + r) * inner + q] = make_float2(sx, sy);
  }
  __syncthreads();
  MID_STAMP(3);
  // ---- phase 2: this half takes every second group of modes, for BOTH samples, with ONE copy of the weight rows ----
  if (!(FNO_MID_SKIP & 1))
  {
    const float sg = conj_w ? -1.f : 1.f;
#pragma unroll
    for (int gi = 0; gi < NG / 2; ++gi) {
      const int g = 2 * gi + sp;
      if (gi + 1 < NG / 2) load_w(g + 2, (gi + 1) & 1);
#pragma unroll
      for (int rr = 0; rr < RB; ++rr) {
        float ax0 = 0.f, ay0 = 0.f, ax1 = 0.f, ay1 = 0.f;
#pragma unroll
        for (int j = 0; j < JW; ++j) {
          const float2 y0 = ys2[(g * RB + rr) * 64 + lb + seg * JW + j];
          const float2 y1 = ys2[NK * 64 + (g * RB + rr) * 64 + lb + seg * JW + j];
          const float2 w = wv[gi & 1][rr][j];
          const float wy = sg * w.y;
          ax0 = fmaf(y0.x, w.x, ax0); ax0 = fmaf(-y0.y, wy, ax0);
          ay0 = fmaf(y0.x, wy, ay0); ay0 = fmaf(y0.y, w.x, ay0);
          ax1 = fmaf(y1.x, w.x, ax1); ax1 = fmaf(-y1.y, wy, ax1);
          ay1 = fmaf(y1.x, wy, ay1); ay1 = fmaf(y1.y, w.x, ay1);
        }
        sh2[(seg * NK + g * RB + rr) * 64 + ql] = make_float2(ax0, ay0);
        sh2[SEGS * NK * 64 + (seg * NK + g * RB + rr) * 64 + ql] = make_float2(ax1, ay1);
      }
    }
  }
  MID_STAMP(4);
  __syncthreads();
  MID_STAMP(5);
  for (int r = seg; r < NK; r += SEGS) {
    float sx = 0.f, sy = 0.f;
#pragma unroll
    for (int k = 0; k < SEGS; ++k) { sx += sh[(k * NK + r) * 64 + ql].x; sy += sh[(k * NK + r) * 64 + ql].y; }
    os[r * 64 + ql] = make_float2(sx, sy);
  }
  __syncthreads();
  MID_STAMP(6);
  // ---- phase 3 ----
  if (!(FNO_MID_SKIP & 4))
  {
    f32x2 v[NK], vs[NK];
#pragma unroll
    for (int k = 0; k < NK; ++k) {
      const float2 t = os[k * 64 + ql];
      v[k] = f32x2{t.x, t.y};
      vs[k] = natural_pair(-t.y, t.x);
    }
    float2* dst = z + (size_t)o * n * inner + q;
    if constexpr (SREG) {
    // two rows of the table, ping-pong (static names: no indexed register array); PAIRS: this wave's row count is even, the
    // loop body is one basic block and a row's scalar loads stay in flight under the row before it
    auto rows3 = [&](auto pairs) {
      constexpr bool PAIRS = decltype(pairs)::value;
      float2 ta[NK], tb[NK];
#pragma unroll
      for (int k = 0; k < NK; ++k) ta[k] = twi[(size_t)min(seg, n - 1) * NK + k];
      for (int r = seg; r < n; r += 2 * SEGS) {
#pragma unroll
        for (int k = 0; k < NK; ++k) tb[k] = twi[(size_t)(PAIRS ? r + SEGS : min(r + SEGS, n - 1)) * NK + k];
        f32x2 s2 = {0.f, 0.f};
#pragma unroll
        for (int k = 0; k < NK; ++k) cfma_uniform<true>(s2, ta[k], v[k], vs[k]);
        if (live) dst[(size_t)r * inner] = make_float2(s2[0], s2[1]);
        if (PAIRS || r + SEGS < n) {
#pragma unroll
          for (int k = 0; k < NK; ++k) ta[k] = twi[(size_t)min(r + 2 * SEGS, n - 1) * NK + k];
          f32x2 s3 = {0.f, 0.f};
#pragma unroll
          for (int k = 0; k < NK; ++k) cfma_uniform<true>(s3, tb[k], v[k], vs[k]);
          if (live) dst[(size_t)(r + SEGS) * inner] = make_float2(s3[0], s3[1]);
        }
      }
    };
    const int nrows3 = seg < n ? (n - seg + SEGS - 1) / SEGS : 0;
    if ((nrows3 & 1) == 0) rows3(MidFlag<true>{}); else rows3(MidFlag<false>{});
    } else {
#pragma unroll 2
    for (int r = seg; r < n; r += SEGS) {
      const float2* t = sh + r * NK;
      f32x2 s2 = {0.f, 0.f};
#pragma unroll
      for (int k = 0; k < NK; ++k) cfma_uniform<false>(s2, t[k], v[k], vs[k]);
      dst[(size_t)r * inner] = make_float2(s2[0], s2[1]);
    }
    }
  }
  MID_STAMP(7);
}
