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

// k_proj_fwd_x3 with two fp16 terms.  a.xmax: device scalar, a bound of |x| (required)
template <int C, int HID, int NPX, int NCO, bool RELU = false>
__global__ void __launch_bounds__(NPX * 4, 2) k_proj_fwd_h2(ProjFwdArgs a) {
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
#pragma unroll 1
    for (int ch = 0; ch < NCH; ++ch) {
      f32x16 acc, lo;      // hh products / cross terms
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
    }
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
