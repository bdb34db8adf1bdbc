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
// row index inside a 32x32 accumulator tile held by lane-half `half`, register r
FNO_DEV int acc_row32(int r, int half) { return (r & 3) + 8 * (r >> 2) + 4 * half; }

// GELU with the exact-erf definition (torch F.gelu default), branch-free:
// erf via Abramowitz-Stegun 7.1.26 (|err| <= 1.5e-7, the fp32 rounding level; measured
// rel-L2 8.6e-8 vs fp64 on N(0,1) inputs, torch's own fp32 gelu: 6.2e-8).  One v_exp,
// one v_rcp and ~8 FMAs; value and derivative share the exponential.
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
FNO_DEV float gelu_f(float x) { float g, d; gelu_both(x, g, d); return g; }
FNO_DEV float gelu_grad_f(float x) { float g, d; gelu_both(x, g, d); return d; }

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
// `tfwd` rows are `tpitch` floats apart (W in HBM; W + 2 for the bank-conflict-free LDS copy).
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
    const float* tf = tfwd + (size_t)(jt * 16 + l15) * tpitch + quad;
    const float* xr = tile + (nt * 16 + l15) * PITCH + rr * W + quad;
#pragma unroll 4
    for (int s = 0; s < W / 4; s += 2) {       // two independent accumulation chains
      d0 = mfma16(tf[4 * s], xr[4 * s], d0);
      d1 = mfma16(tf[4 * s + 4], xr[4 * s + 4], d1);
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
};
