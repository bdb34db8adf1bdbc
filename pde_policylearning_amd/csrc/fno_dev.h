// Device-side helpers shared by the fnoengine HIP kernels (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define FNO_DEV __device__ __forceinline__

// v_mfma_f32_32x32x2_f32: A[i=l&31][k=l>>5], B[k=l>>5][j=l&31];
// D: col = l&31, row = (r&3) + 8*(r>>2) + 4*(l>>5)   (exact fp32 FMA chain)
FNO_DEV f32x16 mfma32(float a, float b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}
// v_mfma_f32_16x16x4_f32: A[i=l&15][k=l>>4], B[k=l>>4][j=l&15];
// D: col = l&15, row = (l>>4)*4 + r
FNO_DEV f32x4 mfma16(float a, float b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}
// 8-wave workgroups place waves w and w + 4 on the same SIMD.  Both run the same phase sequence between the same
// barriers, so without help they reach their MFMA bursts and their VALU bursts together and the two pipes take turns
// instead of overlapping (phase trace: A1 + E + A3 of the projection backward = MFMA time + VALU time).  Raising the
// issue priority of one partner lets it run ahead: its VALU phase then overlaps the other's matrix phase.
// FNO_PRIO: 0 = off, 1 = waves [0, n/2) high, 2 = waves [n/2, n) high.
#ifndef FNO_PRIO
#define FNO_PRIO 0
#endif
#if FNO_PRIO == 1
#define FNO_SIMD_PARTNER_PRIO(wave, nwaves) do { if ((nwaves) == 8 && (wave) < 4) __builtin_amdgcn_s_setprio(2); } while (0)
#elif FNO_PRIO == 2
#define FNO_SIMD_PARTNER_PRIO(wave, nwaves) do { if ((nwaves) == 8 && (wave) >= 4) __builtin_amdgcn_s_setprio(2); } while (0)
#else
#define FNO_SIMD_PARTNER_PRIO(wave, nwaves) do { } while (0)
#endif
// FNO_TRACE_WHICH selects the traced kernel: 1 = projection backward, 2 = block backward (a middle block)
#ifndef FNO_TRACE_WHICH
#define FNO_TRACE_WHICH 1
#endif
#ifdef FNO_TRACE
// Debug build only (-DFNO_TRACE): per-phase shader-clock stamps of workgroup 0, read back with
// fno_debug_trace_dump (tools/trace_phases.py).  g_trace[wave][slot]
__device__ unsigned long long g_trace[16 * 256];
#define FNO_STAMP(slot)                                                                          \
  do {                                                                                           \
    if (trace_on && blockIdx.x == 0 && (threadIdx.x & 63) == 0 && (slot) < 256)                            \
      g_trace[(threadIdx.x >> 6) * 256 + (slot)] = __builtin_readcyclecounter();                 \
  } while (0)
#define FNO_TRACE_IF(cond) const bool trace_on = (cond)
#else
#define FNO_TRACE_IF(cond)
#define FNO_STAMP(slot) do { } while (0)
#endif

// Diagnostic build only (-DFNO_CLOCK, tools/kernel_clock.py): shader-clock and 100 MHz real-time stamps around the tile loop
// of the four hot kernels, one record per workgroup -> the clock the chip holds INSIDE the kernel (MI355X_MICROARCH.md, DVFS
// give-back item 6).  The stamps go to a buffer nothing else reads; no stamp executes in the product build.
#ifdef FNO_CLOCK
__device__ unsigned long long g_clk[4 * 1024 * 8];
// record: {cycles at loop begin, at loop end, real time (100 MHz) at loop begin, at loop end, at kernel entry, 0, 0, 0}
#define FNO_CLK_ENTRY() const unsigned long long clk_re_ = __builtin_amdgcn_s_memrealtime()
#define FNO_CLK_BEGIN() const unsigned long long clk_c0_ = __builtin_amdgcn_s_memtime(), clk_r0_ = __builtin_amdgcn_s_memrealtime()
#define FNO_CLK_END(id)                                                                              \
  do {                                                                                               \
    const unsigned long long c1_ = __builtin_amdgcn_s_memtime(), r1_ = __builtin_amdgcn_s_memrealtime();  \
    if (threadIdx.x == 0 && blockIdx.x < 1024) {                                                     \
      unsigned long long* q_ = g_clk + ((id) * 1024 + blockIdx.x) * 8;                                \
      q_[0] = clk_c0_; q_[1] = c1_; q_[2] = clk_r0_; q_[3] = r1_; q_[4] = clk_re_;                    \
    }                                                                                                \
  } while (0)
#else
#define FNO_CLK_ENTRY() do { } while (0)
#define FNO_CLK_BEGIN() do { } while (0)
#define FNO_CLK_END(id) do { } while (0)
#endif

// Uneven tile shares for kernels that run TWO workgroups per CU (round 4).  The SIMD issues the OLDER wave first, so of the two
// workgroups that share a CU the one dispatched first (blockIdx < gridDim / 2 under in-order dispatch) runs at almost the
// speed it would have alone and the second at 0.55-0.6 of it; with equal static shares the first finishes at 65-70 % of the
// launch and its half of the CU idles (profiles/r04_two_workgroups_per_cu_tail.txt: end times 77 / 91 / 112 / 118 us at the 10 /
// 50 / 90 / 100 % quantiles of k_blk_fwd_t).  The pair (b, b + G/2) owns the tiles p + (G/2) j and splits the j range
// share32 : 32 - share32.  Only speed depends on the dispatch order; every tile is owned exactly once whatever it is.
// Returns first tile, step and end (exclusive, in tile index) of the calling workgroup's loop.
struct TileShare { int first, step, end; };
FNO_DEV TileShare pair_share(int ntiles, int share32) {
  const int G = gridDim.x, b = blockIdx.x;
  if (share32 <= 0 || (G & 1) || ntiles < 2 * G) return TileShare{b, G, ntiles};
  const int H = G >> 1, p = b >= H ? b - H : b;
  const int J = (ntiles - p + H - 1) / H;             // tiles p, p + H, ... of the pair
  const int J1 = (J * share32 + 16) >> 5;
  return b < H ? TileShare{p, H, p + H * J1} : TileShare{p + H * J1, H, p + H * J};
}

// row index inside a 32x32 accumulator tile held by lane-half `half`, register r
FNO_DEV int acc_row32(int r, int half) { return (r & 3) + 8 * (r >> 2) + 4 * half; }

// GELU with the exact-erf definition (torch F.gelu default), branch-free.
// gelu_both (value + derivative, backward kernels): erf via Abramowitz-Stegun 7.1.26, one v_exp and
// one v_rcp shared by value and derivative.
// gelu_f (value only, forward kernels): gelu(x) = max(x, 0) - |x| Phi(-|x|) with
// Phi(-s) = exp2(r(s)), r = degree-7 minimax fit of log2 Phi(-s) on [0, 6] (oracle-side derivation in
// tools/gelu_bench.hip) - one v_exp, no reciprocal, and the Horner chain packs into v_pk_fma_f32.
// Measured on MI355X (tools/gelu_bench.hip): 10.8 issue slots per element vs 14.8, N(0,1.5^2)-weighted
// rel-L2 error vs fp64 4.3e-8 (A&S form: 7.8e-8; torch's own fp32 gelu: 8.8e-8), max abs error 3.0e-7.
FNO_DEV void gelu_both(float x, float& g, float& dg) {
  const float ax = fabsf(x);
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f * 0.70710678118654752440f, ax, 1.0f));
  float poly = fmaf(t, 1.061405429f, -1.453152027f);
  poly = fmaf(t, poly, 1.421413741f);
  poly = fmaf(t, poly, -0.284496736f);
  poly = fmaf(t, poly, 0.254829592f);
  poly *= t;
  const float e = __expf(-0.5f * x * x);
  const float q = 0.5f * poly * e;              // = 0.5 * erfc(|x|/sqrt2)
  const float cdf = x >= 0.0f ? 1.0f - q : q;
  g = x * cdf;
  dg = fmaf(x * 0.39894228040143267794f, e, cdf);
}
FNO_DEV float gelu_f(float x) {
  const float ax = fabsf(x);
  const float s = fminf(ax, 6.0f);
  float r = fmaf(s, 6.119213594502071e-06f, -3.2478157663717866e-05f);
  r = fmaf(s, r, -0.0004947108100168407f);
  r = fmaf(s, r, 0.0075082844123244286f);
  r = fmaf(s, r, -0.052784692496061325f);
  r = fmaf(s, r, -0.4591203033924103f);
  r = fmaf(s, r, -1.1511149406433105f);
  r = fmaf(s, r, -0.9999998211860657f);
  return fmaf(-ax, __builtin_amdgcn_exp2f(r), fmaxf(x, 0.0f));
}
FNO_DEV float gelu_grad_f(float x) { float g, d; gelu_both(x, g, d); return d; }

// ---- GELU on pairs -----------------------------------------------------------------------------------------------------------
// The kernels are bound by VALU issue, and gelu_f / gelu_both compile to one instruction per scalar step plus canonicalisation
// moves around fminf / fmaxf (3 v_max + 1 v_min per element).  On pairs the polynomial steps are v_pk_fma_f32 / v_pk_mul_f32
// (half the issue slots), the clamps v_med3_f32 (a target intrinsic: no canonicalisation) and the sign select of the
// derivative form a v_bfi: 15 instead of 26 instructions per pair for the value, 22 instead of ~36 for value + derivative.
// Same polynomials and the same final operations as the scalar forms.  `six` / `inf` = 6.0f / +infinity in SGPRs
// (gelu_consts: VOP3 takes no literal on gfx9).  -DFNO_GELU_PK=0 restores the scalar forms (A/B arm).
#ifndef FNO_GELU_PK
#define FNO_GELU_PK 1
#endif
typedef float f32x2 __attribute__((ext_vector_type(2)));
FNO_DEV void gelu_consts(float& six, float& inf) {
  asm volatile("s_mov_b32 %0, 0x40c00000\n\ts_mov_b32 %1, 0x7f800000" : "=s"(six), "=s"(inf));
}
template <int NP>
FNO_DEV void gelu_pairs(f32x2 (&x)[NP], float six, float inf) {
#if !FNO_GELU_PK
#pragma unroll
  for (int p = 0; p < NP; ++p) { x[p][0] = gelu_f(x[p][0]); x[p][1] = gelu_f(x[p][1]); }
  return;
#endif
  f32x2 s[NP], r[NP];
#pragma unroll
  for (int p = 0; p < NP; ++p) {
    s[p][0] = __builtin_amdgcn_fmed3f(__builtin_fabsf(x[p][0]), 0.0f, six);
    s[p][1] = __builtin_amdgcn_fmed3f(__builtin_fabsf(x[p][1]), 0.0f, six);
  }
  constexpr float cf[8] = {6.119213594502071e-06f, -3.2478157663717866e-05f, -0.0004947108100168407f, 0.0075082844123244286f,
                           -0.052784692496061325f, -0.4591203033924103f, -1.1511149406433105f, -0.9999998211860657f};
#pragma unroll
  for (int p = 0; p < NP; ++p) r[p] = __builtin_elementwise_fma(s[p], f32x2{cf[0], cf[0]}, f32x2{cf[1], cf[1]});
#pragma unroll
  for (int k = 2; k < 8; ++k)
#pragma unroll
    for (int p = 0; p < NP; ++p) r[p] = __builtin_elementwise_fma(s[p], r[p], f32x2{cf[k], cf[k]});
#pragma unroll
  for (int p = 0; p < NP; ++p) {
    r[p][0] = __builtin_amdgcn_exp2f(r[p][0]);
    r[p][1] = __builtin_amdgcn_exp2f(r[p][1]);
  }
#pragma unroll
  for (int p = 0; p < NP; ++p) {
    x[p][0] = __builtin_fmaf(-__builtin_fabsf(x[p][0]), r[p][0], __builtin_amdgcn_fmed3f(x[p][0], 0.0f, inf));
    x[p][1] = __builtin_fmaf(-__builtin_fabsf(x[p][1]), r[p][1], __builtin_amdgcn_fmed3f(x[p][1], 0.0f, inf));
  }
}
// gelu_pairs' arithmetic on ONE value with scalar fp32 instructions (same operations in the same order: bit-identical).  Eleven
// instructions per value where the packed form spends 7.5 - but scalar fp32 instructions run beside another wave's matrix
// products and packed ones do not, so a kernel with many independent waves per SIMD (k_proj_fwd_w: six,
// no barrier) hides its products behind this form.
FNO_DEV float gelu_s(float x, float six, float inf) {
  const float ax = __builtin_fabsf(x);
  const float s = __builtin_amdgcn_fmed3f(ax, 0.0f, six);
  float r = __builtin_fmaf(s, 6.119213594502071e-06f, -3.2478157663717866e-05f);
  r = __builtin_fmaf(s, r, -0.0004947108100168407f);
  r = __builtin_fmaf(s, r, 0.0075082844123244286f);
  r = __builtin_fmaf(s, r, -0.052784692496061325f);
  r = __builtin_fmaf(s, r, -0.4591203033924103f);
  r = __builtin_fmaf(s, r, -1.1511149406433105f);
  r = __builtin_fmaf(s, r, -0.9999998211860657f);
  r = __builtin_amdgcn_exp2f(r);
  return __builtin_fmaf(-ax, r, __builtin_amdgcn_fmed3f(x, 0.0f, inf));
}
FNO_DEV void gelu8(float (&v)[8], float six, float inf) {
  f32x2 x[4] = {f32x2{v[0], v[1]}, f32x2{v[2], v[3]}, f32x2{v[4], v[5]}, f32x2{v[6], v[7]}};
  gelu_pairs<4>(x, six, inf);
#pragma unroll
  for (int p = 0; p < 4; ++p) { v[2 * p] = x[p][0]; v[2 * p + 1] = x[p][1]; }
}
FNO_DEV float4 gelu4(const float4& v, float six, float inf) {
  f32x2 x[2] = {f32x2{v.x, v.y}, f32x2{v.z, v.w}};
  gelu_pairs<2>(x, six, inf);
  return make_float4(x[0][0], x[0][1], x[1][0], x[1][1]);
}
// value and derivative of a pair (gelu_both's formulas: erfc by Abramowitz-Stegun 7.1.26, one v_exp and one v_rcp per element
// shared by both).  The branch `x >= 0 ? 1 - q : q` becomes 0.5 + copysign(0.5 - q, x).
FNO_DEV void gelu_both2(f32x2 x, f32x2& g, f32x2& dg) {
#if !FNO_GELU_PK
  { float g0, d0, g1, d1; gelu_both(x[0], g0, d0); gelu_both(x[1], g1, d1); g = f32x2{g0, g1}; dg = f32x2{d0, d1}; }
  return;
#endif
  const f32x2 ax = {__builtin_fabsf(x[0]), __builtin_fabsf(x[1])};
  f32x2 t = __builtin_elementwise_fma(ax, f32x2{0.3275911f * 0.70710678118654752440f, 0.3275911f * 0.70710678118654752440f}, f32x2{1.0f, 1.0f});
  t[0] = __builtin_amdgcn_rcpf(t[0]); t[1] = __builtin_amdgcn_rcpf(t[1]);
  // 0.5 * (((((a5 t + a4) t + a3) t + a2) t + a1) t): the 0.5 of q = 0.5 poly e is folded into the coefficients
  f32x2 poly = __builtin_elementwise_fma(t, f32x2{0.5f * 1.061405429f, 0.5f * 1.061405429f}, f32x2{0.5f * -1.453152027f, 0.5f * -1.453152027f});
  poly = __builtin_elementwise_fma(t, poly, f32x2{0.5f * 1.421413741f, 0.5f * 1.421413741f});
  poly = __builtin_elementwise_fma(t, poly, f32x2{0.5f * -0.284496736f, 0.5f * -0.284496736f});
  poly = __builtin_elementwise_fma(t, poly, f32x2{0.5f * 0.254829592f, 0.5f * 0.254829592f});
  poly = poly * t;
  f32x2 e = (x * x) * f32x2{-0.5f * 1.4426950408889634f, -0.5f * 1.4426950408889634f};      // exp(-x^2 / 2) = exp2(-x^2 log2(e) / 2)
  e[0] = __builtin_amdgcn_exp2f(e[0]); e[1] = __builtin_amdgcn_exp2f(e[1]);
  const f32x2 q = poly * e;                                 // = 0.5 erfc(|x| / sqrt 2)
  const f32x2 hq = f32x2{0.5f, 0.5f} - q;
  f32x2 sg;
  sg[0] = __builtin_copysignf(hq[0], x[0]); sg[1] = __builtin_copysignf(hq[1], x[1]);
  const f32x2 cdf = f32x2{0.5f, 0.5f} + sg;
  g = x * cdf;
  dg = __builtin_elementwise_fma(x * f32x2{0.39894228040143267794f, 0.39894228040143267794f}, e, cdf);
}
// NP pairs in lock step (gelu_both2's operations per element, bit-identical): gfx950 needs one wait state between a vector
// instruction and a packed one that reads its result through op_sel (the broadcast constants), and the compiler schedules ONE
// pair's chain depth first and fills the gaps with s_nop 0 (k_proj_bwd_t: 103 of them per chunk, a wave alone on its SIMD pays 4
// cycles each).  Written step by step over the pairs, the next pair's instruction sits in every gap.
#ifndef FNO_GELU_LOCKSTEP
#define FNO_GELU_LOCKSTEP 1      // 0: one pair after the other (A/B arm)
#endif
template <int NP>
FNO_DEV void gelu_both_pairs(const f32x2 (&x)[NP], f32x2 (&g)[NP], f32x2 (&dg)[NP]) {
#if !FNO_GELU_PK || !FNO_GELU_LOCKSTEP
#pragma unroll
  for (int p = 0; p < NP; ++p) gelu_both2(x[p], g[p], dg[p]);
  return;
#endif
  constexpr float ca = 0.3275911f * 0.70710678118654752440f;
  constexpr float cp[5] = {0.5f * 1.061405429f, 0.5f * -1.453152027f, 0.5f * 1.421413741f, 0.5f * -0.284496736f, 0.5f * 0.254829592f};
  f32x2 t[NP], poly[NP], e[NP];
  // (1 + ca |x| on single lanes: |x| is a source modifier of v_fma_f32 and costs nothing there; the packed form has no such
  // modifier and spends two v_and_b32 per pair on it)
#pragma unroll
  for (int p = 0; p < NP; ++p) {
    t[p][0] = __builtin_fmaf(__builtin_fabsf(x[p][0]), ca, 1.0f);
    t[p][1] = __builtin_fmaf(__builtin_fabsf(x[p][1]), ca, 1.0f);
  }
#pragma unroll
  for (int p = 0; p < NP; ++p) e[p] = (x[p] * x[p]) * f32x2{-0.5f * 1.4426950408889634f, -0.5f * 1.4426950408889634f};
#pragma unroll
  for (int p = 0; p < NP; ++p) { t[p][0] = __builtin_amdgcn_rcpf(t[p][0]); t[p][1] = __builtin_amdgcn_rcpf(t[p][1]); }
#pragma unroll
  for (int p = 0; p < NP; ++p) { e[p][0] = __builtin_amdgcn_exp2f(e[p][0]); e[p][1] = __builtin_amdgcn_exp2f(e[p][1]); }
#pragma unroll
  for (int p = 0; p < NP; ++p) poly[p] = __builtin_elementwise_fma(t[p], f32x2{cp[0], cp[0]}, f32x2{cp[1], cp[1]});
#pragma unroll
  for (int k = 2; k < 5; ++k)
#pragma unroll
    for (int p = 0; p < NP; ++p) poly[p] = __builtin_elementwise_fma(t[p], poly[p], f32x2{cp[k], cp[k]});
#pragma unroll
  for (int p = 0; p < NP; ++p) poly[p] = poly[p] * t[p];
#pragma unroll
  for (int p = 0; p < NP; ++p) poly[p] = poly[p] * e[p];                                   // q = 0.5 erfc(|x| / sqrt 2)
#pragma unroll
  for (int p = 0; p < NP; ++p) poly[p] = f32x2{0.5f, 0.5f} - poly[p];
#pragma unroll
  for (int p = 0; p < NP; ++p) { poly[p][0] = __builtin_copysignf(poly[p][0], x[p][0]); poly[p][1] = __builtin_copysignf(poly[p][1], x[p][1]); }
#pragma unroll
  for (int p = 0; p < NP; ++p) poly[p] = f32x2{0.5f, 0.5f} + poly[p];                      // cdf
#pragma unroll
  for (int p = 0; p < NP; ++p) t[p] = x[p] * f32x2{0.39894228040143267794f, 0.39894228040143267794f};
#pragma unroll
  for (int p = 0; p < NP; ++p) { g[p] = x[p] * poly[p]; dg[p] = __builtin_elementwise_fma(t[p], e[p], poly[p]); }
}
// four values at once (a float4 of activations): value back in v, derivative in d
FNO_DEV void gelu_both4(float4& v, float4& d) {
  const f32x2 x[2] = {f32x2{v.x, v.y}, f32x2{v.z, v.w}};
  f32x2 g[2], dd[2];
  gelu_both_pairs<2>(x, g, dd);
  v = make_float4(g[0][0], g[0][1], g[1][0], g[1][1]);
  d = make_float4(dd[0][0], dd[0][1], dd[1][0], dd[1][1]);
}

// sum over the 32 lanes of each wave half (lanes 0-31 / 32-63); every lane gets its half's sum.
// 4 DPP adds (quad xor 1, quad xor 2, row_half_mirror, row_mirror) + one cross-row exchange.
FNO_DEV float dpp_add_(float v, const int ctrl_sel) {
  int t;
  const int iv = __builtin_bit_cast(int, v);
  switch (ctrl_sel) {
    case 0: t = __builtin_amdgcn_update_dpp(0, iv, 0xB1, 0xf, 0xf, true); break;    // quad_perm [1,0,3,2]
    case 1: t = __builtin_amdgcn_update_dpp(0, iv, 0x4E, 0xf, 0xf, true); break;    // quad_perm [2,3,0,1]
    case 2: t = __builtin_amdgcn_update_dpp(0, iv, 0x141, 0xf, 0xf, true); break;   // row_half_mirror
    default: t = __builtin_amdgcn_update_dpp(0, iv, 0x140, 0xf, 0xf, true); break;  // row_mirror
  }
  return v + __builtin_bit_cast(float, t);
}
FNO_DEV float half_reduce_sum(float v) {
  v = dpp_add_(v, 0);
  v = dpp_add_(v, 1);
  v = dpp_add_(v, 2);
  v = dpp_add_(v, 3);
  return v + __shfl_xor(v, 16, 64);
}

// Transposed reduction: every lane holds 16 values v[0..15] (one per accumulator register); on return
// each lane holds ONE total, sum over the 32 lanes of its wave half of v[rid], rid = reduce16_id(lane).
// Each butterfly step halves the number of live values (the lane keeps the half selected by one of its
// lane-id bits and adds the partner's partial of that half), so the whole reduction costs 3 VALU ops per
// input value instead of the 6 of sixteen independent all-lane reductions.
//   step (xor mask, select bit): (8, b3) (7, b2) (2, b1) (1, b0), then one cross-row add (xor 16).
template <int CTRL>
FNO_DEV float dpp_xchg_(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
FNO_DEV int reduce16_id(int lane) { return ((lane >> 3) & 1) | (((lane >> 2) & 1) << 1) | (((lane >> 1) & 1) << 2) | ((lane & 1) << 3); }
FNO_DEV float half_reduce16(const float (&v)[16], int lane) {
  const bool b3 = lane & 8, b2 = lane & 4, b1 = lane & 2, b0 = lane & 1;
  float w[8], u[4], t[2];
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const float keep = b3 ? v[2 * k + 1] : v[2 * k], send = b3 ? v[2 * k] : v[2 * k + 1];
    w[k] = keep + dpp_xchg_<0x128>(send);          // row_ror:8  (lane ^ 8)
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const float keep = b2 ? w[2 * k + 1] : w[2 * k], send = b2 ? w[2 * k] : w[2 * k + 1];
    u[k] = keep + dpp_xchg_<0x141>(send);          // row_half_mirror (lane ^ 7)
  }
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const float keep = b1 ? u[2 * k + 1] : u[2 * k], send = b1 ? u[2 * k] : u[2 * k + 1];
    t[k] = keep + dpp_xchg_<0x4E>(send);           // quad_perm [2,3,0,1] (lane ^ 2)
  }
  const float keep = b0 ? t[1] : t[0], send = b0 ? t[0] : t[1];
  const float r = keep + dpp_xchg_<0xB1>(send);    // quad_perm [1,0,3,2] (lane ^ 1)
  return r + __shfl_xor(r, 16, 64);
}

// ---------------------------------------------------------------------------
// fp32-grade GEMMs on the bf16 matrix cores: every fp32 operand is split into three bf16 terms
// x = h + m + l (8 + 8 + 8 significant bits) and the product keeps the six terms of weight
// >= 2^-16 (lh, hl, mm, mh, hm, hh; the dropped ml / lm / ll terms are <= 2^-24 relative), all
// accumulated in the MFMA's fp32 accumulator.  Measured on MI355X (tools/mfma_bf16x3_test.hip):
// rel-L2 error vs fp64 3.4e-8 / 1.1e-7 / 2.5e-7 at K = 16 / 64 / 256, BELOW the fp32 MFMA's
// 7.4e-8 / 1.4e-7 / 2.9e-7.  Six 32x32x16 bf16 MFMAs (192 cycles per 16 k) replace eight
// 32x32x2 fp32 MFMAs (512 cycles), and - unlike the fp32 MFMA, which executes on the VALU's
// fp32 lanes (tools/mfma_valu_mix.hip) - they run on the matrix pipe concurrently with VALU work.
typedef short bf16x8 __attribute__((ext_vector_type(8)));
FNO_DEV unsigned short f2bf(float x) { return __builtin_bit_cast(unsigned short, (__bf16)x); }
FNO_DEV float bf2f(unsigned short h) { return __builtin_bit_cast(float, (unsigned)h << 16); }
FNO_DEV void split3(float x, unsigned short& h, unsigned short& m, unsigned short& l) {
  h = f2bf(x);
  float r = x - bf2f(h);
  m = f2bf(r);
  r -= bf2f(m);
  l = f2bf(r);
}
// split 8 floats into three packed 8 x bf16 fragments
FNO_DEV void split3x8(const float (&x)[8], bf16x8& h, bf16x8& m, bf16x8& l) {
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    unsigned short a, b, c;
    split3(x[j], a, b, c);
    h[j] = (short)a; m[j] = (short)b; l[j] = (short)c;
  }
}
// acc += A * B for one 16-deep k block; a[0..2] / b[0..2] = (h, m, l) fragments; small terms first
FNO_DEV f32x16 mfma_x3(const bf16x8 (&a)[3], const bf16x8 (&b)[3], f32x16 acc) {
#ifndef FNO_EXP_HALF_MFMA      // (timing experiment: what three instead of six products per k block would buy; results are wrong)
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2], b[0], acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[2], acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[1], acc, 0, 0, 0);
#endif
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[0], acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[1], acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[0], acc, 0, 0, 0);
  return acc;
}
// Buffer loads: the 128-bit resource descriptor and the scalar offset live in SGPRs, each lane supplies one 32-bit byte
// offset - no 64-bit per-lane pointers (which the compiler otherwise keeps, one pair per unrolled load, in VGPRs).
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
FNO_DEV __amdgpu_buffer_rsrc_t make_rsrc(const void* p, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), (short)0, (int)bytes, 0x00020000);
}
FNO_DEV float4 buf_ld4(__amdgpu_buffer_rsrc_t r, int voff, int soff) {
  // (bit_cast of the WHOLE result: picking elements out of the builtin's own vector type compiles to a one-dword load)
  const f32x4 v = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0));
  return make_float4(v[0], v[1], v[2], v[3]);
}
// (16-byte buffer stores: keep `soff` 0 and the offset in `voff` - with an SGPR soffset the compiler pads no store-data hazard and
// on gfx950 data registers rewritten right behind the store lost 5 % of the outputs)
FNO_DEV void buf_st4(__amdgpu_buffer_rsrc_t r, int voff, int soff, const float4& v) {
  __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, f32x4{v.x, v.y, v.z, v.w}), r, voff, soff, 0);
}
FNO_DEV float buf_ld1(__amdgpu_buffer_rsrc_t r, int voff, int soff) {
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 0));
}
// STREAMING loads (round 6): activations a kernel reads exactly once are loaded non-temporally - no allocation in L2 / the
// Infinity Cache - so that what stays resident is what the kernel WRITES, which the next kernel of the chain (walking its
// tiles in the opposite direction: "zigzag", fno_abi.hip) reads first.  Measured on the strip kernel alone: 110 -> 101 us per
// launch (issue -> landed is shorter for nt loads, MI355X_MICROARCH.md, nt-weights).  -DFNO_NT_LOADS=0: default policy (A/B arm).
// Only for loads that take WHOLE 128-byte lines per instruction: a line the L2 does not keep is fetched again by every later
// instruction that touches another part of it (k_block_bwd_t's u loads, 16 bytes of 32 rows per instruction: 0.36 -> 0.46 ms).
#ifndef FNO_NT_LOADS
#define FNO_NT_LOADS 1
#endif
FNO_DEV float buf_ld1s(__amdgpu_buffer_rsrc_t r, int voff, int soff) {
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, FNO_NT_LOADS ? 2 : 0));
}
FNO_DEV float4 buf_ld4s(__amdgpu_buffer_rsrc_t r, int voff, int soff) {
  const f32x4 v = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, FNO_NT_LOADS ? 2 : 0));
  return make_float4(v[0], v[1], v[2], v[3]);
}
FNO_DEV float4 ld4s(const float* p) {
#if FNO_NT_LOADS
  const f32x4 v = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p));
  return make_float4(v[0], v[1], v[2], v[3]);
#else
  return *reinterpret_cast<const float4*>(p);
#endif
}
FNO_DEV bf16x8 buf_ld8h(__amdgpu_buffer_rsrc_t r, int voff, int soff) {
  return __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0));
}
// The same step with the hh products and the five cross terms in SEPARATE accumulators.  The bf16 MFMA truncates the
// aligned addends of its fp32 accumulation: with small and large products in one accumulator every element carries a bias
// of about -1e-8 of its magnitude (tools/mfma_bias_test.hip) - invisible in a relative-L2 check of the GEMM, but DC-type
// reductions further down (bias gradients, k = 0 modes, lifting weights: sums over ~1e6 pixels of a zero-mean signal)
// amplify it by sqrt(#pixels).  Split, the bias drops 50x and the GEMM error halves (5e-8 at K = 64).  Every GEMM whose
// result feeds activations or the dx chain uses this form; `hi` may be a long-running accumulator (products of one size
// class), `lo` is summed into the result once.
FNO_DEV void mfma_x3s(const bf16x8 (&a)[3], const bf16x8 (&b)[3], f32x16& hi, f32x16& lo) {
#ifndef FNO_EXP_HALF_MFMA
  lo = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2], b[0], lo, 0, 0, 0);
  lo = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[2], lo, 0, 0, 0);
  lo = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[1], lo, 0, 0, 0);
#endif
  lo = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[0], lo, 0, 0, 0);
  lo = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[1], lo, 0, 0, 0);
  hi = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[0], hi, 0, 0, 0);
}
FNO_DEV bf16x8 ld8h(const unsigned short* p) { return *reinterpret_cast<const bf16x8*>(p); }
FNO_DEV void st8h(unsigned short* p, bf16x8 v) { *reinterpret_cast<bf16x8*>(p) = v; }

// ---------------------------------------------------------------------------
// fp32-grade GEMMs from TWO fp16 terms and THREE products ("h2").  The training step runs at the board's power cap
// (1400 W, tools/smi_sample.sh): time follows energy, and halving the matrix-pipe work of the split-precision GEMMs took
// 10 % off the step in a timing experiment (-DFNO_EXP_HALF_MFMA).  x = h + l with h = fp16(s x), l = fp16(s x - h) keeps
// 22 significant bits when s (a power of two) brings the operand's largest magnitude to 2^13: values above 2^-16 of the
// maximum keep a relative error of 2^-23, smaller ones an absolute error of 2^-38 of the maximum.  The products hh,
// hl, lh (ll <= 2^-22) in fp32 accumulators give a GEMM error equal to the fp32 MFMA's own (1.5e-7 at K = 64, measured
// against fp64 on the host: DESIGN.md section 4d); hh and the cross terms keep separate accumulators as in mfma_x3s.
// The scale needs a BOUND of the operand's magnitude before it is split: producers publish max |x| of what they store
// (absmax_publish), weights are scanned by the kernel that splits them, products of known factors use the factors' bounds.
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
// power of two s with s * amax < 2^13 (amax >= 0; 0 and denormals -> 1)
FNO_DEV float h2_scale(float amax) {
  const int e = (int)((__builtin_bit_cast(unsigned, amax) >> 23) & 0xffu);      // amax < 2^(e - 126)
  int f = 266 - e;                                                              // exponent field of 2^(13 - (e - 126))
  f = f < 1 ? 1 : (f > 254 ? 254 : f);
  return e == 0 ? 1.0f : __builtin_bit_cast(float, (unsigned)f << 23);
}
// gfx950 HAZARD (found in round 4; reproducer tools/pk_opsel_hazard.hip): a packed-fp32 VOP3P
// instruction (v_pk_mul_f32 / v_pk_add_f32 / v_pk_fma_f32) whose op_sel takes the LOW result's operand from the HIGH dword of
// a VGPR pair in src1 now and then computes lanes 48-63 with that operand read as ZERO while the SIMD's matrix pipe is busy
// with another wave's MFMAs (no wait state helps; the natural order, the op_sel_hi broadcast forms, SGPR sources and
// v_pk_mov_b32 never fail).  hipcc emits the form by itself whenever the register allocator holds a pair in swapped order
// (x[j + 1] below x[j]) - as it did for the prefetch registers of k_blk_fwd_t<.., NT3 = 2>.  An empty asm on the PAIR makes it
// an opaque 64-bit value in natural order: nothing is left for the instruction selector to fold a swap from.
// tools/check_opsel.py lints the built code object for the form (tests/test_abi_and_host.py runs it).
FNO_DEV f32x2 natural_pair(float lo, float hi) {
  f32x2 p = {lo, hi};
  asm volatile("" : "+v"(p));
  return p;
}
// FNO_SPLIT2_VARIANT (build flag): 0 = the product; 1 = element-wise (A/B arm); 6 = the HAZARDOUS form spelled out (a pair
// held in swapped order, un-swapped by op_sel inside the packed operations - what hipcc generated before round 4), kept so
// that the detectors can be shown to fail on it (tools/h2_rate.py, tests/test_fullsize_gpu.py)
#ifndef FNO_SPLIT2_VARIANT
#define FNO_SPLIT2_VARIANT 0
#endif
// low term of the two-term split, l = fp16(v - float(h)) for both halves of a packed h: ONE v_fma_mix{lo,hi}_f16 each
// (fma(v, 1.0, -h) with h read as fp16 straight from its half of the packed register; the difference is exact in fp32, so the
// single rounding to fp16 is the one that v_cvt_f32_f16 / v_sub_f32 / v_cvt_f16_f32 make - 2 instructions per pair instead of
// 4, and the split is paid per element of every GEMM operand.  tools/mix_split_test.hip: bit-identical on 2^23 values incl.
// the fp16 denormal and overflow ranges, alone and beside another wave's MFMAs).  -DFNO_SPLIT2_MIX=0: the compiler's form.
#ifndef FNO_SPLIT2_MIX
#define FNO_SPLIT2_MIX 1
#endif
FNO_DEV f16x2 split2_low(f32x2 v, f16x2 h) {
#if FNO_SPLIT2_MIX
  unsigned l;
  const unsigned hu = __builtin_bit_cast(unsigned, h);
  asm("v_fma_mixlo_f16 %0, %1, 1.0, -%3 op_sel:[0,0,0] op_sel_hi:[0,0,1]\n\t"
      "v_fma_mixhi_f16 %0, %2, 1.0, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "=&v"(l) : "v"(v[0]), "v"(v[1]), "v"(hu));
  return __builtin_bit_cast(f16x2, l);
#else
  return __builtin_convertvector(v - __builtin_convertvector(h, f32x2), f16x2);
#endif
}
FNO_DEV void split2x8(const float (&x)[8], float s, f16x8& h, f16x8& l) {
#if FNO_SPLIT2_VARIANT == 1
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    float v = x[j] * s;
    asm volatile("" : "+v"(v));
    const _Float16 hh = (_Float16)v;
    float r = v - (float)hh;
    asm volatile("" : "+v"(r));
    h[j] = hh; l[j] = (_Float16)r;
  }
#else
#pragma unroll
  for (int j = 0; j < 8; j += 2) {
#if FNO_SPLIT2_VARIANT == 0
    const f32x2 v = natural_pair(x[j], x[j + 1]) * f32x2{s, s};
    const f16x2 hh = __builtin_convertvector(v, f16x2);
    const f16x2 ll = split2_low(v, hh);
#else
    const f32x2 ss = {s, s};
    f32x2 xin = {x[j + 1], x[j]}, v;
    asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0]" : "=&v"(v) : "v"(ss), "v"(xin));
    const f16x2 hh = __builtin_convertvector(v, f16x2);
    const f32x2 hf = __builtin_convertvector(hh, f32x2);
    asm volatile("v_pk_fma_f32 %0, %1, %0, %2 op_sel:[0,1,0] op_sel_hi:[1,0,1] neg_lo:[0,0,1] neg_hi:[0,0,1]" : "+v"(xin) : "v"(ss), "v"(hf));
    const f16x2 ll = __builtin_convertvector(xin, f16x2);
#endif
    h[j] = hh[0]; h[j + 1] = hh[1]; l[j] = ll[0]; l[j + 1] = ll[1];
  }
#endif
}
// acc += A * B for one 16-deep k block; a / b = (h, l) fragments; cross terms first
FNO_DEV void mfma_h2s(const f16x8 (&a)[2], const f16x8 (&b)[2], f32x16& hi, f32x16& lo) {
  lo = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[1], b[0], lo, 0, 0, 0);
  lo = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[0], b[1], lo, 0, 0, 0);
  hi = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[0], b[0], hi, 0, 0, 0);
}
FNO_DEV f32x16 mfma_h2(const f16x8 (&a)[2], const f16x8 (&b)[2], f32x16 acc) {      // one accumulator (weight-gradient GEMMs)
  acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[1], b[0], acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[0], b[1], acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[0], b[0], acc, 0, 0, 0);
  return acc;
}
// ---- term-count generic forms: NTERM = 3 (bf16, six products) or 2 (fp16 "h2", three products).  Fragments are carried
// as 8 x 16-bit vectors (bf16x8) either way; the fp16 forms re-type them at the MFMA.
template <int NTERM>
FNO_DEV void mfma_split_s(const bf16x8 (&a)[NTERM], const bf16x8 (&b)[NTERM], f32x16& hi, f32x16& lo) {
  if constexpr (NTERM == 3) mfma_x3s(a, b, hi, lo);
  else {
    const f16x8 a0 = __builtin_bit_cast(f16x8, a[0]), a1 = __builtin_bit_cast(f16x8, a[1]);
    const f16x8 b0 = __builtin_bit_cast(f16x8, b[0]), b1 = __builtin_bit_cast(f16x8, b[1]);
    lo = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, b0, lo, 0, 0, 0);
    lo = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b1, lo, 0, 0, 0);
    hi = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b0, hi, 0, 0, 0);
  }
}
template <int NTERM>
FNO_DEV f32x16 mfma_split(const bf16x8 (&a)[NTERM], const bf16x8 (&b)[NTERM], f32x16 acc) {
  if constexpr (NTERM == 3) return mfma_x3(a, b, acc);
  else {
    const f16x8 a0 = __builtin_bit_cast(f16x8, a[0]), a1 = __builtin_bit_cast(f16x8, a[1]);
    const f16x8 b0 = __builtin_bit_cast(f16x8, b[0]), b1 = __builtin_bit_cast(f16x8, b[1]);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, b0, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b1, acc, 0, 0, 0);
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b0, acc, 0, 0, 0);
  }
}
// 8 values -> NTERM fragments (the fp16 form multiplies by `scale` first; the bf16 form ignores it)
template <int NTERM>
FNO_DEV void split_n_x8(const float (&x)[8], float scale, bf16x8 (&f)[NTERM]) {
  if constexpr (NTERM == 3) split3x8(x, f[0], f[1], f[2]);
  else {
    f16x8 h, l;
    split2x8(x, scale, h, l);
    f[0] = __builtin_bit_cast(bf16x8, h); f[1] = __builtin_bit_cast(bf16x8, l);
  }
}
// max |v| over the wave -> *dst (float bits, atomic max of the non-negative pattern: order-independent, so deterministic)
FNO_DEV void absmax_publish(float vmax, float* dst) {
  unsigned u = __builtin_bit_cast(unsigned, vmax) & 0x7fffffffu;
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) { const unsigned t = (unsigned)__shfl_xor((int)u, o, 64); u = t > u ? t : u; }
  // same-address atomics serialise at ~12 ns each (32 k of them cost k_lift_rowdft 0.33 ms at 64^3): only a wave that would
  // raise the published value issues one - a handful per launch once the first waves have finished
  if ((threadIdx.x & 63) == 0) {
    unsigned* d = reinterpret_cast<unsigned*>(dst);
    if (u > __hip_atomic_load(d, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(d, u);
  }
}

// Pixel-major split-precision activation tile in LDS: xb[t][px][c] (t = h, m, l), rows of
// C + 8 halfs (16-B aligned, b128 reads with lanes <-> pixels are bank-conflict-free).
// Register-staged prefetch: item (px, cg) = 8 consecutive channels of one pixel, loaded as 8
// dword loads (lanes <-> consecutive pixels: coalesced), split on commit.
template <int NPX, int NT, int C>
struct SplitTilePrefetch {
  static constexpr int PBH = C + 8;                  // halfs per pixel row
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
  FNO_DEV void commit(unsigned short* xb, bool act, int tid) {
#pragma unroll
    for (int i = 0; i < ITER; ++i) {
      const int idx = tid + i * NT;
      const int px = idx % NPX, cg = idx / NPX;
      if (act) {      // (on pairs: 7.5 instead of 15 instructions per value, fno_dev.h "GELU on pairs"; same polynomial, same results)
        float six, inf;
        gelu_consts(six, inf);
        gelu8(v[i], six, inf);
      }
      bf16x8 h, m, l;
      split3x8(v[i], h, m, l);
      if (ITEMS % NT == 0 || idx < ITEMS) {
        unsigned short* dst = xb + px * PBH + cg * 8;
        st8h(dst, h);
        st8h(dst + TERM, m);
        st8h(dst + 2 * TERM, l);
      }
    }
  }
};

FNO_DEV float4 ld4(const float* p);
// lifting parameters lw (C, CL), lb (C) -> LDS [C][4] (zero-padded columns) + [C]
template <int C>
FNO_DEV void stage_lift_params(float* lws, const float* lw, const float* lb, int CL, int tid, int nt) {
  for (int i = tid; i < 4 * C; i += nt) lws[i] = (i & 3) < CL ? lw[(i >> 2) * CL + (i & 3)] : 0.f;
  for (int i = tid; i < C; i += nt) lws[4 * C + i] = lb[i];
}
// The same LDS image for block 0 of a model with a lifting layer, computed instead of loaded: the tile is
// u_0 = W_l x + b_l of the <= 4-channel model input (tfno.py:11-20), so the 64-channel u_0 is never written to or read from
// HBM.  px = idx % NPX with NT a multiple of NPX: a thread's items all sit on ONE pixel, whose CL input values are loaded
// once; lw (C, CL) and lb (C) are wave-uniform per item (cg) and come through the scalar cache.
template <int NPX, int NT, int C>
struct LiftSplitTilePrefetch {
  using S = SplitTilePrefetch<NPX, NT, C>;
  static_assert(NT % NPX == 0 && S::ITEMS % NT == 0, "one pixel per thread");
  float xin[4];
  FNO_DEV void issue(const float* src, size_t row_stride, int CL, int tid) {
    const int px = tid % NPX;
#pragma unroll
    for (int k = 0; k < 4; ++k) xin[k] = k < CL ? src[(size_t)k * row_stride + px] : 0.f;
  }
  // lws: LDS copy of the lifting parameters, [C][4] weights (columns >= CL zero) followed by [C] biases (stage_lift_params)
  FNO_DEV void commit(unsigned short* xb, const float* lws, int tid) {
#pragma unroll
    for (int i = 0; i < S::ITER; ++i) {
      const int idx = tid + i * NT;
      const int px = idx % NPX, cg = idx / NPX;
      float v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int c = cg * 8 + j;
        const float4 wv = ld4(lws + 4 * c);
        v[j] = fmaf(wv.w, xin[3], fmaf(wv.z, xin[2], fmaf(wv.y, xin[1], fmaf(wv.x, xin[0], lws[4 * C + c]))));
      }
      bf16x8 h, m, l;
      split3x8(v, h, m, l);
      unsigned short* dst = xb + px * S::PBH + cg * 8;
      st8h(dst, h);
      st8h(dst + S::TERM, m);
      st8h(dst + 2 * S::TERM, l);
    }
  }
};

// Spectral K-extension for rows that do not tile the workgroup's pixel tile ("loose rows": odd row lengths such as the
// PINO observers' padded T axis, W >= 32): acc[c][px] += sum_s Z[row(px)][s][c] . T[s][w(px)] for one 32-pixel block that
// starts at flattened plane index f.  A block overlaps at most two rows; each gets its own pass with the table column
// masked to the lanes (pixels) that belong to it.  zs = [rows of the tile][K2][C][2] from row r_lo, tinv_s = [2 K2][W].
template <int C>
FNO_DEV f32x16 kext_loose_rows(f32x16 acc, const float* zs, const float* tinv_s, int K2, int W, int f, int r_lo, int mt,
                               int l31, int half) {
  const int ra = f / W;
  const int w0 = f - ra * W;                 // wave-uniform: position of the block's first pixel in its row
  const int wl = w0 + l31;
  {
    const float* zr = zs + ((size_t)((ra - r_lo) * K2) * C + mt * 32 + l31) * 2 + half;
    const bool in = wl < W;
    const float* tv = tinv_s + half * W + (in ? wl : 0);
#pragma unroll 2
    for (int s = 0; s < K2; ++s) acc = mfma32(zr[s * C * 2], in ? tv[2 * s * W] : 0.f, acc);
  }
  if (w0 + 31 >= W) {
    const float* zr = zs + ((size_t)((ra + 1 - r_lo) * K2) * C + mt * 32 + l31) * 2 + half;
    const bool in = wl >= W;
    const float* tv = tinv_s + half * W + (in ? wl - W : 0);
#pragma unroll 2
    for (int s = 0; s < K2; ++s) acc = mfma32(zr[s * C * 2], in ? tv[2 * s * W] : 0.f, acc);
  }
  return acc;
}

FNO_DEV float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
FNO_DEV void st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }

// ---------------------------------------------------------------------------
// Shared tile helpers.  A workgroup tile is NPX consecutive pixels of one sample's plane
// for all channels, staged in LDS as rows of PITCH = NPX + 4 floats (16-B aligned rows;
// row-wise b32 reads and column-wise b128 reads are both bank-conflict-free).
// ---------------------------------------------------------------------------

// HBM -> LDS: rows [0, nrows) of a (channels, PW) plane slice, 16 B per lane, optional GELU.
template <int NPX, int NT>
FNO_DEV void stage_rows(float* dst, const float* src, size_t row_stride, int nrows, int nrows_pad, bool act,
                        int tid) {
  constexpr int PITCH = NPX + 4;
  for (int idx = tid; idx < nrows_pad * (NPX / 4); idx += NT) {
    const int c = idx / (NPX / 4), q = idx % (NPX / 4);
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (c < nrows) {
      v = ld4(src + (size_t)c * row_stride + 4 * q);
      if (act) { v.x = gelu_f(v.x); v.y = gelu_f(v.y); v.z = gelu_f(v.z); v.w = gelu_f(v.w); }
    }
    st4(dst + c * PITCH + 4 * q, v);
  }
}

// Same for a compile-time row count: all loads are issued before the first LDS write so the
// HBM latency is paid once per tile, not once per pass.
template <int NPX, int NT, int NROWS>
FNO_DEV void stage_rows_t(float* dst, const float* src, size_t row_stride, bool act, int tid) {
  constexpr int PITCH = NPX + 4;
  constexpr int TOTAL = NROWS * (NPX / 4);
  constexpr int ITER = (TOTAL + NT - 1) / NT;
  float4 v[ITER];
#pragma unroll
  for (int i = 0; i < ITER; ++i) {
    const int idx = tid + i * NT;
    const int c = idx / (NPX / 4), q = idx % (NPX / 4);
    v[i] = (TOTAL % NT == 0 || idx < TOTAL) ? ld4(src + (size_t)c * row_stride + 4 * q) : make_float4(0.f, 0.f, 0.f, 0.f);
  }
#pragma unroll
  for (int i = 0; i < ITER; ++i) {
    const int idx = tid + i * NT;
    const int c = idx / (NPX / 4), q = idx % (NPX / 4);
    if (act) { v[i].x = gelu_f(v[i].x); v[i].y = gelu_f(v[i].y); v[i].z = gelu_f(v[i].z); v[i].w = gelu_f(v[i].w); }
    if (TOTAL % NT == 0 || idx < TOTAL) st4(dst + c * PITCH + 4 * q, v[i]);
  }
}

// Truncated row DFT of the tile held in LDS (rows = channels), fp32 MFMA 16x16x4:
//   X1[b, prow, k2, c] = sum_w tile[c][r*W + w] * (tfwd[2k2][w] + i tfwd[2k2+1][w])
// D[row = j][col = c]: a lane ends up with (re, im) pairs -> float2 stores, 128-B runs.
// `tfwd` rows are `tpitch` floats apart (W in HBM; W + 4 for the 16-B aligned, bank-conflict-free LDS copy).
template <int NCH, int NPX, int NW>
FNO_DEV void row_dft_epilogue(const float* tile, const float* __restrict__ tfwd, int tpitch, float* __restrict__ x1,
                              int b, int px0, int P, int W, int K2out, int NJ, int wave, int lane) {
  constexpr int PITCH = NPX + 4;
  const int l15 = lane & 15, quad = lane >> 4;
  const int R = NPX / W;
  const int njobs = (NCH / 16) * R * NJ;
  for (int job = wave; job < njobs; job += NW) {
    const int nt = job % (NCH / 16);
    const int rr = (job / (NCH / 16)) % R;
    const int jt = job / ((NCH / 16) * R);
    f32x4 d0 = {0.f, 0.f, 0.f, 0.f}, d1 = {0.f, 0.f, 0.f, 0.f};
    // k index of MFMA #t in group q: quad <-> pixel 16q + 4*quad + t, so each lane feeds four
    // MFMAs from ONE b128 read per operand (two independent accumulation chains)
    const float* tf = tfwd + (size_t)(jt * 16 + l15) * tpitch + 4 * quad;
    const float* xr = tile + (nt * 16 + l15) * PITCH + rr * W + 4 * quad;
    // operands of FOUR k steps are in flight together (left to itself the compiler reuses one register set: read, wait,
    // four MFMAs, read ... - the phase then runs at LDS latency, 2.9 k instead of 1 k cycles per 128-pixel row)
    for (int q0 = 0; q0 < W / 16; q0 += 4) {
      float4 av[4], bv[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const bool in = q0 + j < W / 16;          // rows of 32 pixels: two k steps
        av[j] = in ? ld4(tf + 16 * (q0 + j)) : make_float4(0.f, 0.f, 0.f, 0.f);
        bv[j] = in ? ld4(xr + 16 * (q0 + j)) : make_float4(0.f, 0.f, 0.f, 0.f);
      }
#ifndef FNO_DFT_NOPIPE
      __builtin_amdgcn_sched_barrier(0);
#endif
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        d0 = mfma16(av[j].x, bv[j].x, d0);
        d1 = mfma16(av[j].y, bv[j].y, d1);
        d0 = mfma16(av[j].z, bv[j].z, d0);
        d1 = mfma16(av[j].w, bv[j].w, d1);
      }
    }
    const int prow = px0 / W + rr;
    const int c = nt * 16 + l15;
#pragma unroll
    for (int pr = 0; pr < 2; ++pr) {
      const int k2 = jt * 8 + quad * 2 + pr;
      if (k2 < K2out)
        *reinterpret_cast<float2*>(x1 + ((((size_t)b * P + prow) * K2out + k2) * NCH + c) * 2) =
            make_float2(d0[2 * pr] + d1[2 * pr], d0[2 * pr + 1] + d1[2 * pr + 1]);
    }
  }
}

// Counter-based dropout (rno.py:89,98: nn.Dropout on the spectral branch's input): the keep / drop decision of element e
// is a hash of (e, two 32-bit words the caller draws per call and keeps in device memory), so the backward kernels
// regenerate the mask instead of reading one.  drop_scale = 0 with probability p, 1 / (1 - p) otherwise.
FNO_DEV unsigned mix32(unsigned x) {      // (lowbias32: full-avalanche 32-bit mixer)
  x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
  return x;
}
struct DropCfg { unsigned key, thr; float inv_keep; };        // thr = p * 2^32: dropped iff hash < thr
FNO_DEV DropCfg drop_cfg(const unsigned* seed, float p) {
  DropCfg d;
  d.key = seed[0] ^ (seed[1] * 0x85ebca6bU);
  d.thr = (unsigned)fminf(p * 4294967296.f, 4294967040.f);
  d.inv_keep = 1.f / (1.f - p);
  return d;
}
FNO_DEV float drop_scale(const DropCfg& d, size_t e) {
  const unsigned x = ((unsigned)e ^ d.key) * 0x9E3779B1U + (unsigned)(e >> 32);
  return mix32(x) < d.thr ? 0.f : d.inv_keep;
}

// Register-staged prefetch of a tile's rows: issue() starts the HBM loads for the NEXT tile,
// commit() (one iteration later) applies the optional GELU and writes them to LDS.
template <int NPX, int NT, int NROWS, int NROWS_PAD>
struct TilePrefetch {
  static constexpr int PITCH = NPX + 4;
  static constexpr int TOTAL = NROWS_PAD * (NPX / 4);
  static constexpr int ITER = (TOTAL + NT - 1) / NT;
  float4 v[ITER];
  FNO_DEV void issue(const float* src, size_t row_stride, int tid) {
#pragma unroll
    for (int i = 0; i < ITER; ++i) {
      const int idx = tid + i * NT;
      const int c = idx / (NPX / 4), q = idx % (NPX / 4);
      v[i] = (idx < TOTAL && c < NROWS) ? ld4(src + (size_t)c * row_stride + 4 * q) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  }
  FNO_DEV void commit(float* dst, bool act, int tid) {
#pragma unroll
    for (int i = 0; i < ITER; ++i) {
      const int idx = tid + i * NT;
      const int c = idx / (NPX / 4), q = idx % (NPX / 4);
      float4 t = v[i];
      if (act) { t.x = gelu_f(t.x); t.y = gelu_f(t.y); t.z = gelu_f(t.z); t.w = gelu_f(t.w); }
      if (idx < TOTAL) st4(dst + c * PITCH + 4 * q, t);
    }
  }
  // commit with a per-piece transform f(t, row c, float4 index q within the row)
  template <typename F>
  FNO_DEV void commit_with(float* dst, int tid, F f) {
#pragma unroll
    for (int i = 0; i < ITER; ++i) {
      const int idx = tid + i * NT;
      const int c = idx / (NPX / 4), q = idx % (NPX / 4);
      float4 t = v[i];
      if (idx < TOTAL) { f(t, c, q, i); st4(dst + c * PITCH + 4 * q, t); }
    }
  }
};
