// The "spectral middle": everything that happens on the truncated spectrum.
// All tensors here are O(kept modes) small (<= ~10 % of one activation), complex
// interleaved (float2), batch-major:  [B][K1][(K2)][Klast][C].
//
//   k_axis_pass        truncated DFT / inverse DFT along one leading axis, as a
//                      table-driven complex mat-vec: out[o][r][q] = sum_n tw[r][n] in[o][n][q]
//   k_pack_w / k_unpack_dw   reference corner-weight layout (Cin,Cout,m..) <-> mode-major [K][Cin][Cout]
//   k_mode_gemm_*      the complex contraction 'bixy,ioxy->boxy' over kept modes and its two adjoints (32 / 64 channels:
//                      one real GEMM per mode on the fp32 matrix cores, k_mode_gemm_mfma / k_mode_gemm_dw_mfma)
//                      (neuralop/models/spectral_convolution.py:15-36, rno.py:51-58, basics.py:14-24)
//   k_rowdft_generic / k_rowidft_generic   last-dim passes for shapes the fused MFMA
//                      kernels do not cover (W not a multiple of 32, odd channel counts)
#pragma once
#include "fno_dev.h"

// ---------------------------------------------------------------------------
// Truncating pass (many -> few):  out[o][r][q] = sum_n twT[n][r] * in[o][n][q],  n_out = NR small.
// Each thread keeps all NR complex outputs of one column q in registers; the n range is split over
// SEGS waves (8; 4 for the widest kept extents, whose partial sums would not fit LDS otherwise; each wave's loads
// are all in flight together) and combined through LDS.  twT rows are
// wave-uniform and NR is a compile-time constant -> wide scalar loads, no per-element guards.
//   block (64, SEGS), grid (ceil(inner/64), outer)
template <int NR, int SEGS = 8, bool TLDS = false>
__global__ void __launch_bounds__(64 * SEGS) k_axis_fwd(const float2* __restrict__ in, float2* __restrict__ out,
                                                        const float2* __restrict__ twT_g, int n_in, int inner) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float2* sh = reinterpret_cast<float2*>(smem);     // [SEGS][NR][64]
  const int ql = threadIdx.x;
  const int seg = __builtin_amdgcn_readfirstlane(threadIdx.y);
  const int q = blockIdx.x * 64 + ql;
  const int o = blockIdx.y;
  // TLDS (wide kept extents): the (n_in, NR) table is copied into LDS first - it shares the space with the partial sums,
  // which are only written after the sweep.  A table row of 40 complex values is five 16-dword scalar loads; with ~100
  // SGPRs only one row is in flight per wave and the sweep waits on the scalar cache for every n (the table does not fit
  // it), whereas broadcast LDS reads pipeline freely.
  const float2* twT = twT_g;
  if constexpr (TLDS) {
    for (int i = threadIdx.y * 64 + threadIdx.x; i < n_in * NR; i += 64 * SEGS) sh[i] = twT_g[i];
    __syncthreads();
    twT = sh;
  }
  float2 acc[NR];
#pragma unroll
  for (int r = 0; r < NR; ++r) acc[r] = make_float2(0.f, 0.f);
  if (q < inner) {
    const float2* src = in + (size_t)o * n_in * inner + q;
    for (int nb = seg; nb < n_in; nb += 8 * SEGS) {
      float2 v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int n = nb + SEGS * j;
        v[j] = (n < n_in) ? src[(size_t)n * inner] : make_float2(0.f, 0.f);
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int n = nb + SEGS * j;
        if (n < n_in) {
          const float2* t = twT + (size_t)n * NR;
#pragma unroll
          for (int r = 0; r < NR; ++r) {
            const float2 w = t[r];
            acc[r].x = fmaf(w.x, v[j].x, acc[r].x); acc[r].x = fmaf(-w.y, v[j].y, acc[r].x);
            acc[r].y = fmaf(w.x, v[j].y, acc[r].y); acc[r].y = fmaf(w.y, v[j].x, acc[r].y);
          }
        }
      }
    }
  }
  if constexpr (TLDS) __syncthreads();              // every wave is done reading the table
#pragma unroll
  for (int r = 0; r < NR; ++r) sh[(seg * NR + r) * 64 + ql] = acc[r];
  __syncthreads();
  if (q < inner) {
    for (int r = seg; r < NR; r += SEGS) {
      float sx = 0.f, sy = 0.f;
#pragma unroll
      for (int k = 0; k < SEGS; ++k) { sx += sh[(k * NR + r) * 64 + ql].x; sy += sh[(k * NR + r) * 64 + ql].y; }
      out[((size_t)o * NR + r) * inner + q] = make_float2(sx, sy);
    }
  }
}

// Expanding pass (few -> many):  out[o][r][q] = sum_k tw[r][k] * in[o][k][q],  n_in = NK small.
// Each thread holds its column's NK inputs in registers and sweeps its share of r (16 waves per
// block share the sweep); tw rows are wave-uniform, NK compile-time -> wide scalar loads.
//   block (64, SEGS), grid (ceil(inner/64), outer); SEGS = 16, or 8 when NK > 24 (register budget of the NK inputs)
template <int NK, int SEGS = 16, bool TLDS = false>
__global__ void __launch_bounds__(64 * SEGS) k_axis_inv(const float2* __restrict__ in, float2* __restrict__ out,
                                                   const float2* __restrict__ tw_g, int n_out, int inner) {
  const int ql = threadIdx.x;
  const int seg = __builtin_amdgcn_readfirstlane(threadIdx.y);
  const int q = blockIdx.x * 64 + ql;
  const int o = blockIdx.y;
  const float2* tw = tw_g;
  if constexpr (TLDS) {        // the (n_out, NK) table through LDS instead of the scalar cache (see k_axis_fwd)
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float2* tws = reinterpret_cast<float2*>(smem);
    for (int i = threadIdx.y * 64 + threadIdx.x; i < n_out * NK; i += 64 * SEGS) tws[i] = tw_g[i];
    __syncthreads();
    tw = tws;
  }
  if (q >= inner) return;
  float2 v[NK];
#pragma unroll
  for (int k = 0; k < NK; ++k) v[k] = in[((size_t)o * NK + k) * inner + q];
#pragma unroll 2
  for (int r = seg; r < n_out; r += SEGS) {
    const float2* t = tw + (size_t)r * NK;
    float sr = 0.f, si = 0.f;
#pragma unroll
    for (int k = 0; k < NK; ++k) {
      const float2 w = t[k];
      sr = fmaf(w.x, v[k].x, sr); sr = fmaf(-w.y, v[k].y, sr);
      si = fmaf(w.x, v[k].y, si); si = fmaf(w.y, v[k].x, si);
    }
    out[((size_t)o * n_out + r) * inner + q] = make_float2(sr, si);
  }
}

// generic fallbacks for kept counts without a specialisation (any n_in / n_out)
__global__ void __launch_bounds__(256) k_axis_generic(const float2* __restrict__ in, float2* __restrict__ out,
                                                      const float2* __restrict__ tw, int n_in, int n_out, int inner,
                                                      int tw_transposed) {
  const int q = blockIdx.x * blockDim.x + threadIdx.x;
  const int r = blockIdx.y, o = blockIdx.z;
  if (q >= inner) return;
  const float2* src = in + (size_t)o * n_in * inner + q;
  float sr = 0.f, si = 0.f;
  for (int n = 0; n < n_in; ++n) {
    const float2 v = src[(size_t)n * inner];
    const float2 w = tw_transposed ? tw[(size_t)n * n_out + r] : tw[(size_t)r * n_in + n];
    sr = fmaf(w.x, v.x, sr); sr = fmaf(-w.y, v.y, sr);
    si = fmaf(w.x, v.y, si); si = fmaf(w.y, v.x, si);
  }
  out[((size_t)o * n_out + r) * inner + q] = make_float2(sr, si);
}

// ---------------------------------------------------------------------------
struct ModeMap {
  int nlead;         // number of leading (two-sided) dims: 1 (2-D) or 2 (3-D)
  int K[3];          // kept extent per dim: 2*m for leading dims, m_last for the last
  int m[3];          // per-corner extent per dim
  int wl_stride;     // last-dim extent of the stored corner weight (>= m[last]; dialect C 3-D: modes3)
  int Cin, Cout, Ktot;
};

struct CornerPtrs { const float2* p[4]; };
struct CornerPtrsMut { float2* p[4]; };
struct CornerPtrsL { const float2* p[16][4]; };      // [layer][corner]
struct CornerPtrsMutL { float2* p[16][4]; };

__device__ __forceinline__ void mode_decompose(const ModeMap& mm, int k, int& corner, size_t& loc) {
  // k = (k1 * K[1] + k2) * K[2] + k3 (3-D) or k1 * K[1] + klast (2-D)
  int kl, k1, k2 = 0;
  if (mm.nlead == 2) { kl = k % mm.K[2]; k2 = (k / mm.K[2]) % mm.K[1]; k1 = k / (mm.K[2] * mm.K[1]); }
  else { kl = k % mm.K[1]; k1 = k / mm.K[1]; }
  const int h1 = k1 >= mm.m[0];
  const int l1 = k1 - h1 * mm.m[0];
  if (mm.nlead == 2) {
    const int h2 = k2 >= mm.m[1];
    const int l2 = k2 - h2 * mm.m[1];
    corner = 2 * h1 + h2;
    loc = ((size_t)l1 * mm.m[1] + l2) * mm.wl_stride + kl;
  } else {
    corner = h1;
    loc = (size_t)l1 * mm.wl_stride + kl;
  }
}

// LDS-tiled forms of the two layout changes.  The corner tensors are (i, o)-major with the kept modes innermost, the
// packed arrays mode-major with (i, o) innermost: a thread-per-element copy is coalesced on one side only and moves 8 bytes
// per 64-byte sector on the other (the PINO observers carry 0.2 - 1 GB of spectral weights per layer).  A workgroup owns
// one (layer, corner, leading-mode position `rest`, input channel i): the tile [o][kl] (Cout x wl_stride complex) is read
// along its contiguous side, transposed through LDS and written along the other's.
//   grid (ncorner * nrest * Cin, layers), block 256, LDS Cout * (wl_stride + 1) float2
FNO_DEV int tile_mode_base(const ModeMap& mm, int corner, int rest) {        // packed mode index of (corner, rest, kl = 0)
  if (mm.nlead == 2) {
    const int l1 = rest / mm.m[1], l2 = rest - l1 * mm.m[1];
    return ((l1 + (corner >> 1) * mm.m[0]) * mm.K[1] + l2 + (corner & 1) * mm.m[1]) * mm.K[2];
  }
  return (rest + corner * mm.m[0]) * mm.K[1];
}
// zero64 (or null): 64 floats cleared by the first workgroup - the magnitude-bound slots the kernels BEHIND this launch publish
// into (one fill launch less per forward pass)
__global__ void __launch_bounds__(256) k_pack_w_tiled(CornerPtrsL cw, float2* __restrict__ wp, float2* __restrict__ wpt,
                                                      ModeMap mm, size_t stride, int nrest, int TI, float* __restrict__ zero64) {
  if (zero64 && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x < 64) zero64[threadIdx.x] = 0.f;
  // TI consecutive input channels per workgroup: the transposed copy wpt[k][o][i] is then written in TI * 8-byte pieces
  // (whole 64-byte sectors at TI = 8) instead of lone 8-byte elements
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float2* t = reinterpret_cast<float2*>(smem);     // [TI][Cout][wl + 1]
  const int wl = mm.wl_stride, WLP = wl + 1, klive = mm.K[mm.nlead];
  const int l = blockIdx.y;
  const int nib = mm.Cin / TI;
  const int i0 = (blockIdx.x % nib) * TI, rest = (blockIdx.x / nib) % nrest, corner = blockIdx.x / (nib * nrest);
  const size_t per = (size_t)nrest * wl;
  const float2* src = cw.p[l][corner] + (size_t)i0 * mm.Cout * per + (size_t)rest * wl;
  const int nio = TI * mm.Cout;
  for (int idx = threadIdx.x; idx < nio * klive; idx += 256) {      // live modes only (dialect C stores modes3 >= the live count)
    const int io = idx / klive, kl = idx - io * klive;
    t[io * WLP + kl] = src[(size_t)io * per + kl];
  }
  __syncthreads();
  const int k0 = tile_mode_base(mm, corner, rest);
  if (wp)
    for (int idx = threadIdx.x; idx < klive * nio; idx += 256) {
      const int kl = idx / nio, io = idx - kl * nio;
      wp[l * stride + ((size_t)(k0 + kl) * mm.Cin + i0) * mm.Cout + io] = t[io * WLP + kl];
    }
  if (wpt)
    for (int idx = threadIdx.x; idx < klive * nio; idx += 256) {
      const int kl = idx / nio, r = idx - kl * nio, o = r / TI, ii = r - o * TI;
      wpt[l * stride + ((size_t)(k0 + kl) * mm.Cout + o) * mm.Cin + i0 + ii] = t[(ii * mm.Cout + o) * WLP + kl];
    }
}
// corner-layout gradients from mode-major dWp[k][i][o]; stored entries beyond the kept last-dim range get zero
__global__ void __launch_bounds__(256) k_unpack_dw_tiled(const float2* __restrict__ dwp, CornerPtrsMutL gw, ModeMap mm,
                                                         size_t stride, int nrest) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float2* t = reinterpret_cast<float2*>(smem);
  const int wl = mm.wl_stride, WLP = wl + 1, klive = mm.K[mm.nlead];
  const int l = blockIdx.y;
  const int i = blockIdx.x % mm.Cin, rest = (blockIdx.x / mm.Cin) % nrest, corner = blockIdx.x / (mm.Cin * nrest);
  const size_t per = (size_t)nrest * wl;
  const int k0 = tile_mode_base(mm, corner, rest);
  for (int idx = threadIdx.x; idx < klive * mm.Cout; idx += 256) {
    const int kl = idx / mm.Cout, o = idx - kl * mm.Cout;
    t[o * WLP + kl] = dwp[l * stride + ((size_t)(k0 + kl) * mm.Cin + i) * mm.Cout + o];
  }
  __syncthreads();
  float2* dst = gw.p[l][corner] + (size_t)i * mm.Cout * per + (size_t)rest * wl;
  for (int idx = threadIdx.x; idx < mm.Cout * wl; idx += 256) {
    const int o = idx / wl, kl = idx - o * wl;
    dst[(size_t)o * per + kl] = kl < klive ? t[o * WLP + kl] : make_float2(0.f, 0.f);
  }
}

// PLANE-MAJOR corner weights (FnoSpecDesc / FnoModelDesc .weight_planes): the stored tensor is [wl][Cin][Cout][rest]
// (rest = the leading per-corner mode positions, innermost) instead of [Cin][Cout][rest][wl] - the live last-dim slices
// [0, klive) of a dialect-C weight are then ONE contiguous prefix of every tensor (PINObserverFullField at T = 1: 1 plane
// of 12), which is what the layout kernels, the optimizer and the gradient exchange each touch.  A workgroup owns a tile of
// TI input x TO output channels x RC rest positions of one (layer, corner, live plane): read along rest (RC * 8 bytes
// contiguous), written with o innermost (wp) and with i innermost (wpt): whole 64-byte sectors at TI = TO = 8.
//   grid (ncorner * klive * ceil(Cin / 8) * ceil(Cout / 8) * ceil(nrest / RC), layers), block 256, LDS 64 (RC + 1) float2
#define PL_T 8
FNO_DEV void plane_tile(const ModeMap& mm, int nrest, int RC, int bx, int& corner, int& kl, int& i0, int& o0, int& r0) {
  const int nrc = (nrest + RC - 1) / RC, nob = (mm.Cout + PL_T - 1) / PL_T, nib = (mm.Cin + PL_T - 1) / PL_T;
  r0 = (bx % nrc) * RC; bx /= nrc;
  o0 = (bx % nob) * PL_T; bx /= nob;
  i0 = (bx % nib) * PL_T; bx /= nib;
  const int klive = mm.K[mm.nlead];
  kl = bx % klive; corner = bx / klive;
}
__global__ void __launch_bounds__(256) k_pack_w_planes(CornerPtrsL cw, float2* __restrict__ wp, float2* __restrict__ wpt,
                                                       ModeMap mm, size_t stride, int nrest, int RC) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float2* t = reinterpret_cast<float2*>(smem);     // [PL_T i][PL_T o][RC + 1]
  const int RCP = RC + 1, l = blockIdx.y;
  int corner, kl, i0, o0, r0;
  plane_tile(mm, nrest, RC, blockIdx.x, corner, kl, i0, o0, r0);
  const int nr = min(RC, nrest - r0);
  const float2* src = cw.p[l][corner] + (size_t)kl * mm.Cin * mm.Cout * nrest;
  for (int idx = threadIdx.x; idx < PL_T * PL_T * RC; idx += 256) {
    const int r = idx % RC, io = idx / RC, ii = io / PL_T, oo = io - ii * PL_T;
    if (r < nr && i0 + ii < mm.Cin && o0 + oo < mm.Cout)
      t[io * RCP + r] = src[((size_t)(i0 + ii) * mm.Cout + o0 + oo) * nrest + r0 + r];
  }
  __syncthreads();
  for (int idx = threadIdx.x; idx < PL_T * PL_T * nr; idx += 256) {
    const int a = idx % PL_T, b = (idx / PL_T) % PL_T, r = idx / (PL_T * PL_T);
    const size_t k = (size_t)tile_mode_base(mm, corner, r0 + r) + kl;
    if (wp && i0 + b < mm.Cin && o0 + a < mm.Cout)           // a = o innermost
      wp[l * stride + (k * mm.Cin + i0 + b) * mm.Cout + o0 + a] = t[(b * PL_T + a) * RCP + r];
    if (wpt && i0 + a < mm.Cin && o0 + b < mm.Cout)          // a = i innermost
      wpt[l * stride + (k * mm.Cout + o0 + b) * mm.Cin + i0 + a] = t[(a * PL_T + b) * RCP + r];
  }
}
// plane-major corner gradients from mode-major dWp[k][i][o]: the live planes only (the caller keeps the dead ones zero)
__global__ void __launch_bounds__(256) k_unpack_dw_planes(const float2* __restrict__ dwp, CornerPtrsMutL gw, ModeMap mm,
                                                          size_t stride, int nrest, int RC) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float2* t = reinterpret_cast<float2*>(smem);
  const int RCP = RC + 1, l = blockIdx.y;
  int corner, kl, i0, o0, r0;
  plane_tile(mm, nrest, RC, blockIdx.x, corner, kl, i0, o0, r0);
  const int nr = min(RC, nrest - r0);
  for (int idx = threadIdx.x; idx < PL_T * PL_T * nr; idx += 256) {
    const int oo = idx % PL_T, ii = (idx / PL_T) % PL_T, r = idx / (PL_T * PL_T);
    const size_t k = (size_t)tile_mode_base(mm, corner, r0 + r) + kl;
    if (i0 + ii < mm.Cin && o0 + oo < mm.Cout)
      t[(ii * PL_T + oo) * RCP + r] = dwp[l * stride + (k * mm.Cin + i0 + ii) * mm.Cout + o0 + oo];
  }
  __syncthreads();
  float2* dst = gw.p[l][corner] + (size_t)kl * mm.Cin * mm.Cout * nrest;
  for (int idx = threadIdx.x; idx < PL_T * PL_T * RC; idx += 256) {
    const int r = idx % RC, io = idx / RC, ii = io / PL_T, oo = io - ii * PL_T;
    if (r < nr && i0 + ii < mm.Cin && o0 + oo < mm.Cout)
      dst[((size_t)(i0 + ii) * mm.Cout + o0 + oo) * nrest + r0 + r] = t[io * RCP + r];
  }
}

// ---------------------------------------------------------------------------
// O[b][k][o] = sum_i X[b][k][i] * W[k][i][o]          (conj_w: use conj(W), for the adjoint
// with wpt[k][o][i] passed as w and Cin/Cout swapped)
__global__ void __launch_bounds__(256) k_mode_gemm(const float2* __restrict__ x, const float2* __restrict__ w,
                                                   float2* __restrict__ out, int B, int Ktot, int Cin, int Cout,
                                                   int conj_w) {
  const int o = threadIdx.x % Cout;
  const int bs = threadIdx.x / Cout;
  const int nb = blockDim.x / Cout;
  const int k = blockIdx.x;
  const int b = blockIdx.y * nb + bs;
  if (b >= B || bs >= nb) return;
  const float2* xr = x + ((size_t)b * Ktot + k) * Cin;
  const float2* wr = w + (size_t)k * Cin * Cout + o;
  float sr = 0.f, si = 0.f;
  const float sg = conj_w ? -1.f : 1.f;
#pragma unroll 4
  for (int i = 0; i < Cin; ++i) {
    const float2 a = xr[i];
    float2 ww = wr[(size_t)i * Cout];
    ww.y *= sg;
    sr = fmaf(a.x, ww.x, sr); sr = fmaf(-a.y, ww.y, sr);
    si = fmaf(a.x, ww.y, si); si = fmaf(a.y, ww.x, si);
  }
  out[((size_t)b * Ktot + k) * Cout + o] = make_float2(sr, si);
}

// LDS-tiled variant for the hot case (Cout a divisor of 512): a workgroup owns one mode and a
// batch tile of BT = J * (512 / Cout) samples; the mode's weight matrix and the tile's spectra are
// staged once in LDS and every thread accumulates J batch rows of one output channel.
//   grid (Ktot, ceil(B / BT)), block 512, LDS (Cin*Cout + BT*Cin) float2
template <int J>
__global__ void __launch_bounds__(512) k_mode_gemm_lds(const float2* __restrict__ x, const float2* __restrict__ w,
                                                       float2* __restrict__ out, int B, int Ktot, int Cin, int Cout,
                                                       int conj_w) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float2* ws = reinterpret_cast<float2*>(smem);      // Cin x Cout
  float2* xs = ws + (size_t)Cin * Cout;              // BT x Cin
  const int ng = 512 / Cout, BT = J * ng;
  const int k = blockIdx.x, b0 = blockIdx.y * BT;
  const int nb = min(BT, B - b0);
  const int tid = threadIdx.x;
  const float2* wk = w + (size_t)k * Cin * Cout;
  for (int i = tid; i < Cin * Cout; i += 512) ws[i] = wk[i];
  for (int i = tid; i < nb * Cin; i += 512) {
    const int bb = i / Cin, ci = i % Cin;
    xs[i] = x[((size_t)(b0 + bb) * Ktot + k) * Cin + ci];
  }
  __syncthreads();
  const int o = tid % Cout, bg = tid / Cout;
  const float sg = conj_w ? -1.f : 1.f;
  float sr[J], si[J];
#pragma unroll
  for (int j = 0; j < J; ++j) { sr[j] = 0.f; si[j] = 0.f; }
#pragma unroll 4
  for (int i = 0; i < Cin; ++i) {
    float2 wv = ws[(size_t)i * Cout + o];
    wv.y *= sg;
#pragma unroll
    for (int j = 0; j < J; ++j) {
      const int bb = bg + j * ng;
      const float2 a = xs[(bb < nb ? bb : 0) * Cin + i];
      sr[j] = fmaf(a.x, wv.x, sr[j]); sr[j] = fmaf(-a.y, wv.y, sr[j]);
      si[j] = fmaf(a.x, wv.y, si[j]); si[j] = fmaf(a.y, wv.x, si[j]);
    }
  }
#pragma unroll
  for (int j = 0; j < J; ++j) {
    const int bb = bg + j * ng;
    if (bb < nb) out[((size_t)(b0 + bb) * Ktot + k) * Cout + o] = make_float2(sr[j], si[j]);
  }
}

// dW[k][i][o] = sum_b conj(X[b][k][i]) * G[b][k][o]: a workgroup owns one mode and J * (512/Cout)
// input channels; the batch is staged in LDS in chunks of 64.
//   grid (Ktot, ceil(Cin / (J * 512/Cout))), block 512, LDS 64*(J*ng + Cout) float2
template <int J>
__global__ void __launch_bounds__(512) k_mode_gemm_dw_lds(const float2* __restrict__ x, const float2* __restrict__ g,
                                                          float2* __restrict__ dw, int B, int Ktot, int Cin,
                                                          int Cout) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int k = blockIdx.x, tid = threadIdx.x;
  const int o = tid % Cout, ig = tid / Cout, ng = 512 / Cout;
  const int IT = J * ng, i0 = blockIdx.y * IT;
  float2* xs = reinterpret_cast<float2*>(smem);      // 64 x IT
  float2* gs = xs + (size_t)64 * IT;                 // 64 x Cout
  float sr[J], si[J];
#pragma unroll
  for (int j = 0; j < J; ++j) { sr[j] = 0.f; si[j] = 0.f; }
  for (int b0 = 0; b0 < B; b0 += 64) {
    const int nb = min(64, B - b0);
    __syncthreads();
    for (int i = tid; i < nb * IT; i += 512) {
      const int bb = i / IT, ii = i0 + i % IT;
      xs[i] = (ii < Cin) ? x[((size_t)(b0 + bb) * Ktot + k) * Cin + ii] : make_float2(0.f, 0.f);
    }
    for (int i = tid; i < nb * Cout; i += 512) gs[i] = g[((size_t)(b0 + i / Cout) * Ktot + k) * Cout + i % Cout];
    __syncthreads();
#pragma unroll 4
    for (int bb = 0; bb < nb; ++bb) {
      const float2 gg = gs[bb * Cout + o];
#pragma unroll
      for (int j = 0; j < J; ++j) {
        const float2 a = xs[bb * IT + ig + j * ng];
        sr[j] = fmaf(a.x, gg.x, sr[j]); sr[j] = fmaf(a.y, gg.y, sr[j]);      // conj(a) * g
        si[j] = fmaf(a.x, gg.y, si[j]); si[j] = fmaf(-a.y, gg.x, si[j]);
      }
    }
  }
#pragma unroll
  for (int j = 0; j < J; ++j) {
    const int i = i0 + ig + j * ng;
    if (i < Cin) dw[((size_t)k * Cin + i) * Cout + o] = make_float2(sr[j], si[j]);
  }
}

// ---------------------------------------------------------------------------
// Matrix-core variants of the mode contraction for 32 / 64 channels: per kept mode k the complex product
// 'bi,io->bo' is ONE real GEMM on the interleaved (re, im) storage,
//   [.. xr_i xi_i ..] (B x 2Cin)  .  [[ wr  wi ], [ -wi  wr ]]_(i,o) (2Cin x 2Cout)  =  [.. yr_o yi_o ..] (B x 2Cout),
// run as exact-fp32 v_mfma_f32_32x32x2_f32 tiles (K step 2 = one complex input channel).  A workgroup owns one mode and
// 64 batch rows; the expanded real weight block and the spectra are staged once in LDS (weight rows padded by 32 floats so
// that the two lane halves, which read adjacent k rows, fall on disjoint banks; spectrum rows by 1 float for the
// row-per-lane A reads).  Wave (mt, nt): batch rows [32 mt, +32), real output columns [32 nt, +32).
//   grid (Ktot, ceil(B / MG_BR)), block (MG_BR / 32) * (2 CO / 32) waves, LDS MG_BR (2 CI + 1) floats + max(CI (CO + 1), CO (CI + 1)) float2
// MG_BR = batch rows per workgroup: 64, or 32 when the batch has no more (RNO2d / PINO observers at batch 32: half of the
// 64-row tile was padding; with 32 rows the workgroup is four waves and 50 KB, three per CU instead of two).
template <int CI, int CO, int MG_BR>
__global__ void __launch_bounds__((MG_BR / 32) * (2 * CO / 32) * 64) k_mode_gemm_mfma(const float2* __restrict__ x,
                                                                           const float2* __restrict__ w,
                                                                           float2* __restrict__ out, int B, int Ktot,
                                                                           int conj_w, size_t x_ms, size_t w_ms, size_t o_ms,
                                                                           int trans_w) {
  // trans_w: w holds the (CO, CI) matrix of the FORWARD contraction and this launch contracts with its transpose
  // (out[b][n] = sum_m x[b][m] w[n][m]): the adjoint reads the forward's packed weights, no transposed copy exists
  // blockIdx.z = member of a batch of independent contractions (fan-outs); *_ms = member strides in float2 (0: shared)
  x += blockIdx.z * x_ms; w += blockIdx.z * w_ms; out += blockIdx.z * o_ms;
  // The weight block stays COMPLEX in LDS (rows padded by one float2) and the real operand [[wr wi], [-wi wr]] is formed
  // per lane while it is read: 33 KB instead of the 80 KB expanded block, so two workgroups share a CU (the launch is
  // bound by staging latency, not arithmetic: 28 -> 16 us for the RNO cell's batched contractions).
  constexpr int NTN = 2 * CO / 32, NT = (MG_BR / 32) * NTN * 64, PA = 2 * CI + 1;
  constexpr int PW_N = CO + 1, PW_T = CI + 1;         // row pitch (float2) of the stored block: (CI, CO) or, trans_w, (CO, CI)
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* xs = smem;                                              // [MG_BR][PA]
  float2* wc = reinterpret_cast<float2*>(smem + MG_BR * PA);     // (MG_BR * PA floats: 8-byte aligned, MG_BR even)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, half = lane >> 5;
  const int mt = wave / NTN, nt = wave % NTN;
  const int k = blockIdx.x, b0 = blockIdx.y * MG_BR;
  const int nb = min(MG_BR, B - b0);
  const float sg = conj_w ? -1.f : 1.f;
  const float2* wk = w + (size_t)k * CI * CO;
  if (trans_w) {
    for (int i = tid; i < CI * CO; i += NT) wc[(i / CI) * PW_T + i % CI] = wk[i];      // stored (CO, CI): row = output channel
  } else {
    for (int i = tid; i < CI * CO; i += NT) wc[(i / CO) * PW_N + i % CO] = wk[i];      // stored (CI, CO): row = input channel
  }
  for (int i = tid; i < MG_BR * CI; i += NT) {
    const int bb = i / CI, ci = i % CI;
    const float2 v = bb < nb ? x[((size_t)(b0 + bb) * Ktot + k) * CI + ci] : make_float2(0.f, 0.f);
    xs[bb * PA + 2 * ci] = v.x;
    xs[bb * PA + 2 * ci + 1] = v.y;
  }
  __syncthreads();
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  const float* ap = xs + (mt * 32 + l31) * PA + half;
  // this lane's real output column n = 2 o + c and k row 2 s + half:  half 0: (wr, wi)_c,  half 1: (-wi, wr)_c
  const int n = nt * 32 + l31, o = n >> 1, c = n & 1;
  const float2* wp = trans_w ? wc + o * PW_T : wc + o;
  const int wstep = trans_w ? 1 : PW_N;
  const bool pick_y = (c ^ half) != 0;                  // the imaginary part feeds (half 0, c 1) and (half 1, c 0)
  const float sgn = pick_y ? (half ? -sg : sg) : 1.f;
#pragma unroll 8
  for (int s = 0; s < CI; ++s) {
    const float2 wv = wp[s * wstep];
    acc = mfma32(ap[2 * s], (pick_y ? wv.y : wv.x) * sgn, acc);
  }
  float* op = reinterpret_cast<float*>(out);
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int bb = mt * 32 + acc_row32(r, half);
    if (bb < nb) op[((size_t)(b0 + bb) * Ktot + k) * (2 * CO) + nt * 32 + l31] = acc[r];
  }
}

// dW[k][i][o] = sum_b conj(X[b][k][i]) G[b][k][o] the same way: rows i, real columns 2o + c, K = 2B with
//   A[i][2b + h] = (xr, xi)_h,   B[2b][2o + c] = (gr, gi)_c,   B[2b + 1][2o + c] = (gi, -gr)_c.
// The batch is staged in chunks of DW_BC samples (32: 57 KB of LDS at 64 channels, two workgroups per CU, so that the 288
// workgroups of BASELINE config 2 - 72 modes x 4 layers - are resident at once; chunks of 64 samples, 114 KB, ran them as
// one workgroup per CU in two rounds: 30 us per launch).  Wave (mt, nt): input channels [32 mt, +32), columns [32 nt, +32).
//   grid (Ktot), block (CI / 32) * (2 CO / 32) waves, LDS DW_BC * 2 CI + 2 DW_BC (2 CO + 32) floats
// DW_BC is 32 for batches above 32 samples, 16 below (four workgroups per CU: 17 vs 18 us at batch 32, 11 vs 13 at batch 16).
template <int CI, int CO, int DW_BC>
__global__ void __launch_bounds__((CI / 32) * (2 * CO / 32) * 64) k_mode_gemm_dw_mfma(const float2* __restrict__ x,
                                                                                      const float2* __restrict__ g,
                                                                                      float2* __restrict__ dw, int B,
                                                                                      int Ktot, size_t x_ms, size_t g_ms, size_t d_ms) {
  x += blockIdx.y * x_ms; g += blockIdx.y * g_ms; dw += blockIdx.y * d_ms;        // blockIdx.y = member (see k_mode_gemm_mfma)
  constexpr int NTN = 2 * CO / 32, NT = (CI / 32) * NTN * 64, PA = 2 * CI, PB = 2 * CO + 32;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* xs = smem;                 // [DW_BC][PA]   raw interleaved spectra of the chunk
  float* gs = smem + DW_BC * PA;    // [2 DW_BC][PB]  rows 2b: (gr, gi), rows 2b + 1: (gi, -gr)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, half = lane >> 5;
  const int mt = wave / NTN, nt = wave % NTN;
  const int k = blockIdx.x;
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  for (int b0 = 0; b0 < B; b0 += DW_BC) {
    const int nb = min(DW_BC, B - b0);
    __syncthreads();
    for (int i = tid; i < DW_BC * CI; i += NT) {
      const int bb = i / CI, ci = i % CI;
      const float2 v = bb < nb ? x[((size_t)(b0 + bb) * Ktot + k) * CI + ci] : make_float2(0.f, 0.f);
      xs[bb * PA + 2 * ci] = v.x;
      xs[bb * PA + 2 * ci + 1] = v.y;
    }
    for (int i = tid; i < DW_BC * CO; i += NT) {
      const int bb = i / CO, o = i % CO;
      const float2 v = bb < nb ? g[((size_t)(b0 + bb) * Ktot + k) * CO + o] : make_float2(0.f, 0.f);
      float* r0 = gs + (2 * bb) * PB + 2 * o;
      r0[0] = v.x; r0[1] = v.y;
      r0[PB] = v.y; r0[PB + 1] = -v.x;
    }
    __syncthreads();
    const float* ap = xs + 2 * (mt * 32 + l31) + half;
    const float* bp = gs + half * PB + nt * 32 + l31;
#pragma unroll 8
    for (int s = 0; s < DW_BC; ++s) acc = mfma32(ap[s * PA], bp[(2 * s) * PB], acc);
  }
  float* op = reinterpret_cast<float*>(dw);
#pragma unroll
  for (int r = 0; r < 16; ++r)
    op[((size_t)k * CI + mt * 32 + acc_row32(r, half)) * (2 * CO) + nt * 32 + l31] = acc[r];
}

// ---------------------------------------------------------------------------
// Tiny batches (B <= 4: one or two samples per GPU of the PINO fine-tuning configurations, whose spectral weights are
// 0.2 - 1 GB per layer): the contraction is a stream over the weights with almost no arithmetic, and the tile kernels
// above keep one workgroup per CU busy staging 32 KB per mode.  Here a wave owns one mode: lane o streams column o of the
// mode's (Cin, Cout) block (consecutive lanes -> consecutive addresses), the mode's spectrum values are wave-uniform
// (readlane), nothing goes through LDS except the transposed variant's tile.  Cin, Cout <= 64.
//   grid ceil(Ktot / 4), block 256 (4 modes)
template <int NB>
__global__ void __launch_bounds__(256) k_mode_gemv(const float2* __restrict__ x, const float2* __restrict__ w,
                                                   float2* __restrict__ out, int Ktot, int Cin, int Cout, int conj_w) {
  const int lane = threadIdx.x & 63, k = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (k >= Ktot) return;
  const float sg = conj_w ? -1.f : 1.f;
  float2 xv[NB];
#pragma unroll
  for (int b = 0; b < NB; ++b) xv[b] = lane < Cin ? x[((size_t)b * Ktot + k) * Cin + lane] : make_float2(0.f, 0.f);
  float sr[NB], si[NB];
#pragma unroll
  for (int b = 0; b < NB; ++b) { sr[b] = 0.f; si[b] = 0.f; }
  const float2* wk = w + (size_t)k * Cin * Cout + (lane < Cout ? lane : 0);
#pragma unroll 8
  for (int i = 0; i < Cin; ++i) {
    float2 wv = wk[(size_t)i * Cout];
    wv.y *= sg;
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      const float ax = __shfl(xv[b].x, i, 64), ay = __shfl(xv[b].y, i, 64);
      sr[b] = fmaf(ax, wv.x, sr[b]); sr[b] = fmaf(-ay, wv.y, sr[b]);
      si[b] = fmaf(ax, wv.y, si[b]); si[b] = fmaf(ay, wv.x, si[b]);
    }
  }
  if (lane < Cout) {
#pragma unroll
    for (int b = 0; b < NB; ++b) out[((size_t)b * Ktot + k) * Cout + lane] = make_float2(sr[b], si[b]);
  }
}
// adjoint with the forward's weights: out[b][k][i] = sum_o g[b][k][o] conj(W[k][i][o]).  A workgroup owns one mode; the
// (Cin, Cout) block is copied into a padded LDS tile along its rows (coalesced), then lane i walks its own row.
//   grid Ktot, block 256, LDS Cin * (Cout + 1) float2
template <int NB>
__global__ void __launch_bounds__(256) k_mode_gemv_t(const float2* __restrict__ g, const float2* __restrict__ w,
                                                     float2* __restrict__ out, int Ktot, int Cin, int Cout) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float2* t = reinterpret_cast<float2*>(smem);       // [Cin][Cout + 1]
  const int k = blockIdx.x, tid = threadIdx.x;
  const float2* wk = w + (size_t)k * Cin * Cout;
  for (int e = tid; e < Cin * Cout; e += 256) t[(e / Cout) * (Cout + 1) + e % Cout] = wk[e];
  __syncthreads();
  if (tid >= 64) return;
  float2 gv[NB];
#pragma unroll
  for (int b = 0; b < NB; ++b) gv[b] = tid < Cout ? g[((size_t)b * Ktot + k) * Cout + tid] : make_float2(0.f, 0.f);
  float sr[NB], si[NB];
#pragma unroll
  for (int b = 0; b < NB; ++b) { sr[b] = 0.f; si[b] = 0.f; }
  const float2* row = t + (tid < Cin ? tid : 0) * (Cout + 1);
#pragma unroll 8
  for (int o = 0; o < Cout; ++o) {
    const float2 wv = row[o];
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      const float ax = __shfl(gv[b].x, o, 64), ay = __shfl(gv[b].y, o, 64);
      sr[b] = fmaf(ax, wv.x, sr[b]); sr[b] = fmaf(ay, wv.y, sr[b]);        // g * conj(w)
      si[b] = fmaf(ay, wv.x, si[b]); si[b] = fmaf(-ax, wv.y, si[b]);
    }
  }
  if (tid < Cin) {
#pragma unroll
    for (int b = 0; b < NB; ++b) out[((size_t)b * Ktot + k) * Cin + tid] = make_float2(sr[b], si[b]);
  }
}
// dW[k][i][o] = sum_b conj(X[b][k][i]) G[b][k][o]: an outer product per mode, pure write streaming.
//   grid ceil(Ktot / 4), block 256 (wave = mode, lane = o)
template <int NB>
__global__ void __launch_bounds__(256) k_mode_outer_dw(const float2* __restrict__ x, const float2* __restrict__ g,
                                                       float2* __restrict__ dw, int Ktot, int Cin, int Cout) {
  const int lane = threadIdx.x & 63, k = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (k >= Ktot) return;
  float2 xv[NB], gv[NB];
#pragma unroll
  for (int b = 0; b < NB; ++b) {
    xv[b] = lane < Cin ? x[((size_t)b * Ktot + k) * Cin + lane] : make_float2(0.f, 0.f);
    gv[b] = lane < Cout ? g[((size_t)b * Ktot + k) * Cout + lane] : make_float2(0.f, 0.f);
  }
  float2* dk = dw + (size_t)k * Cin * Cout + lane;
#pragma unroll 8
  for (int i = 0; i < Cin; ++i) {
    float sr = 0.f, si = 0.f;
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      const float ax = __shfl(xv[b].x, i, 64), ay = __shfl(xv[b].y, i, 64);
      sr = fmaf(ax, gv[b].x, sr); sr = fmaf(ay, gv[b].y, sr);              // conj(x) * g
      si = fmaf(ax, gv[b].y, si); si = fmaf(-ay, gv[b].x, si);
    }
    if (lane < Cout) dk[(size_t)i * Cout] = make_float2(sr, si);
  }
}

// dW[k][i][o] = sum_b conj(X[b][k][i]) * G[b][k][o]
__global__ void __launch_bounds__(256) k_mode_gemm_dw(const float2* __restrict__ x, const float2* __restrict__ g,
                                                      float2* __restrict__ dw, int B, int Ktot, int Cin, int Cout) {
  const int o = threadIdx.x % Cout;
  const int is = threadIdx.x / Cout;
  const int ni = blockDim.x / Cout;
  const int k = blockIdx.x;
  const int i = blockIdx.y * ni + is;
  if (i >= Cin || is >= ni) return;
  float sr = 0.f, si = 0.f;
  for (int b = 0; b < B; ++b) {
    const float2 a = x[((size_t)b * Ktot + k) * Cin + i];
    const float2 gg = g[((size_t)b * Ktot + k) * Cout + o];
    // conj(a) * g
    sr = fmaf(a.x, gg.x, sr); sr = fmaf(a.y, gg.y, sr);
    si = fmaf(a.x, gg.y, si); si = fmaf(-a.y, gg.x, si);
  }
  dw[((size_t)k * Cin + i) * Cout + o] = make_float2(sr, si);
}

// ---------------------------------------------------------------------------
// generic last-dim passes (any W, any C: odd row lengths such as PINO's padded T, 34-channel RNO).
// A workgroup owns RB consecutive rows of one sample for all channels (rows are adjacent in memory, so the
// channel segments it reads / writes are RB*W floats long); twiddle tables and the tile live in LDS.
// x (B, C, P, W) -> x1 (B, P, K2, C, 2);  tfwd (>= 2*K2 rows, W)
//   LDS: table [W][2*K2] | tile [C][RB*W + 1]
__global__ void __launch_bounds__(256) k_rowdft_generic(const float* __restrict__ x, float2* __restrict__ x1,
                                                        const float* __restrict__ tfwd, int C, int P, int W, int K2,
                                                        int RB) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int K2E = (K2 + 1) & ~1;                   // table rows padded to a multiple of 4 floats
  float* tab = smem;                                // [w][2*K2E]: (re, im) pairs of every kept bin
  float* xs = smem + W * 2 * K2E;                   // [c][RB*W + 1]
  const int nblk = (P + RB - 1) / RB;
  const int b = blockIdx.x / nblk, p0 = (blockIdx.x % nblk) * RB;
  const int nr = min(RB, P - p0);
  const int pitch = RB * W + 1;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < W * 2 * K2E; i += blockDim.x) {
    const int w = i / (2 * K2E), j = i % (2 * K2E);
    tab[i] = j < 2 * K2 ? tfwd[(size_t)j * W + w] : 0.f;
  }
  // staging: a wave owns channels wave, wave + 4, ...; lanes run along the nr*W contiguous floats of a channel.
  // Four channels (up to 4 x ceil(seg/64) loads per lane) are in flight before the first LDS write; no divisions.
  const int seg = nr * W;
  const size_t cstride = (size_t)P * W;
  const float* xb = x + ((size_t)b * C * P + p0) * W;
  for (int c0 = wave; c0 < C; c0 += 16) {
    float v[4][4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int c = c0 + 4 * k;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int o = lane + 64 * j;
        v[k][j] = (c < C && o < seg) ? xb[(size_t)c * cstride + o] : 0.f;
      }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int c = c0 + 4 * k;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int o = lane + 64 * j;
        if (c < C && o < seg) xs[c * pitch + o] = v[k][j];
      }
    }
    for (int o = lane + 256; o < seg; o += 64)        // rows longer than 256 floats in total: plain tail
#pragma unroll
      for (int k = 0; k < 4; ++k)
        if (c0 + 4 * k < C) xs[(c0 + 4 * k) * pitch + o] = xb[(size_t)(c0 + 4 * k) * cstride + o];
  }
  __syncthreads();
  // item = (row r, bin pair kp, channel c): two bins = 4 accumulators, channel fastest (lanes <-> channels)
  const int nkp = K2E / 2;
  for (int it = threadIdx.x; it < nr * nkp * C; it += blockDim.x) {
    const int c = it % C, kp = (it / C) % nkp, r = it / (C * nkp);
    const float* xr = xs + c * pitch + r * W;
    const float* tp = tab + 4 * kp;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
#pragma unroll 4
    for (int w = 0; w < W; ++w) {
      const float v = xr[w];
      const float4 t = ld4(tp + w * 2 * K2E);       // wave-uniform address: LDS broadcast
      a0 = fmaf(v, t.x, a0); a1 = fmaf(v, t.y, a1); a2 = fmaf(v, t.z, a2); a3 = fmaf(v, t.w, a3);
    }
    const size_t row = (size_t)b * P + p0 + r;
    x1[(row * K2 + 2 * kp) * C + c] = make_float2(a0, a1);
    if (2 * kp + 1 < K2) x1[(row * K2 + 2 * kp + 1) * C + c] = make_float2(a2, a3);
  }
}

// z (B, P, K2, C, 2) -> y (B, C, P, W) (+ bias[c]);  tinv (2*K2, W)
//   LDS: table [2*K2][W] | spectra [RB][K2][C][2]
template <int K2C>
__global__ void __launch_bounds__(256) k_rowidft_generic(const float2* __restrict__ z, float* __restrict__ y,
                                                         const float* __restrict__ tinv,
                                                         const float* __restrict__ bias, int C, int P, int W,
                                                         int K2, int RB) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* tab = smem;                                            // [2*K2][W]
  float2* zs = reinterpret_cast<float2*>(smem + ((2 * K2 * W + 1) & ~1));   // [r][k2][c]
  const int nblk = (P + RB - 1) / RB;
  const int b = blockIdx.x / nblk, p0 = (blockIdx.x % nblk) * RB;
  const int nr = min(RB, P - p0);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < 2 * K2 * W; i += blockDim.x) tab[i] = tinv[i];
  const float2* zr = z + ((size_t)b * P + p0) * K2 * C;
  for (int i = threadIdx.x; i < nr * K2 * C; i += blockDim.x) zs[i] = zr[i];
  __syncthreads();
  // a wave owns channels wave, wave + 4, ...; lanes run along the nr*W contiguous output floats of the channel
  // (coalesced stores, conflict-free table reads, broadcast spectra); the row of an element comes from compares.
  const int seg = nr * W;
  const size_t cstride = (size_t)P * W;
  float* yb = y + ((size_t)b * C * P + p0) * W;
  for (int o = lane; o < seg; o += 64) {
    int r = 0;
    for (int k = 1; k < nr; ++k) r += (o >= k * W);
    const int w = o - r * W;
    const float2* zrow = zs + (size_t)r * K2 * C;
    if constexpr (K2C > 0) {
      // this lane's table column stays in registers for all channels (K2 <= K2C bins, zero beyond K2)
      float tr[K2C], ti[K2C];
#pragma unroll
      for (int k2 = 0; k2 < K2C; ++k2) {
        tr[k2] = k2 < K2 ? tab[(2 * k2) * W + w] : 0.f;
        ti[k2] = k2 < K2 ? tab[(2 * k2 + 1) * W + w] : 0.f;
      }
      for (int c = wave; c < C; c += 4) {
        float s0 = bias ? bias[c] : 0.f, s1 = 0.f;
#pragma unroll
        for (int k2 = 0; k2 < K2C; ++k2) {
          // wave-uniform address: LDS broadcast; bins past K2 re-read the last one against a zero table entry (no branch)
          const float2 v = zrow[(k2 < K2 ? k2 : K2 - 1) * C + c];
          s0 = fmaf(v.x, tr[k2], s0);
          s1 = fmaf(v.y, ti[k2], s1);
        }
        yb[(size_t)c * cstride + o] = s0 + s1;
      }
    } else {
      // many kept bins (full-spectrum plans): table column from LDS
      for (int c = wave; c < C; c += 4) {
        float s0 = bias ? bias[c] : 0.f, s1 = 0.f;
#pragma unroll 4
        for (int k2 = 0; k2 < K2; ++k2) {
          const float2 v = zrow[k2 * C + c];
          s0 = fmaf(v.x, tab[(2 * k2) * W + w], s0);
          s1 = fmaf(v.y, tab[(2 * k2 + 1) * W + w], s1);
        }
        yb[(size_t)c * cstride + o] = s0 + s1;
      }
    }
  }
}

// ---------------------------------------------------------------------------
// lanes <-> channels variants of the generic last-dim passes (K2 <= 32 kept bins).  One item = (row r, channel c) keeps all
// its bins in registers; the table row of the current w is wave-uniform and comes through the scalar cache
// (tT = [W][2*K2P], zero-padded bins), so the LDS only carries the activation tile: one 4-byte access per 2*K2P FMAs
// instead of one broadcast read per FMA pair.  Tile [C][RB*W + 1]: odd pitch, conflict-free along c.
// Persistent workgroups: while the current tile is transformed out of the LDS, the whole next tile (80 registers per lane) is
// already in flight from HBM, so the load latency overlaps the FMA loop.  A tile is CG = 4 * CPW channels x RB rows: a wave
// owns CPW channels and 80 / CPW 64-float slots of each channel's contiguous RB*W run (CPW = 16: all of <= 64 channels, runs
// of <= 320 floats; used when the rows do not allow the 16-byte variant below).
template <int K2P, int CPW>
__global__ void __launch_bounds__(256) k_rowdft_chan(const float* __restrict__ x, float2* __restrict__ x1,
                                                     const float* __restrict__ tT, int C, int P, int W, int K2, int RB,
                                                     int ntiles, int act_in) {
  constexpr int SL = 80 / CPW, CG = 4 * CPW;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* xs = smem;                                 // [CG][RB*W + 1]
  const int nblk = (P + RB - 1) / RB, ncg = (C + CG - 1) / CG;
  const int pitch = RB * W + 1;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;       // 4 waves
  const size_t cstride = (size_t)P * W;
  float v[CPW][SL];                                 // channel wave + 4k of the group, floats lane + 64 j of the nr*W run
  auto fetch = [&](int tile) {
    const int pb = tile % nblk, cg = (tile / nblk) % ncg, b = tile / (nblk * ncg);
    const int p0 = pb * RB, seg = min(RB, P - p0) * W;
    const float* xb = x + ((size_t)b * C * P + p0) * W;
#pragma unroll
    for (int k = 0; k < CPW; ++k)
#pragma unroll
      for (int j = 0; j < SL; ++j)
        v[k][j] = xb[(size_t)min(cg * CG + wave + 4 * k, C - 1) * cstride + min(lane + 64 * j, seg - 1)];   // clamped: no branches
  };
  int tile = blockIdx.x;
  if (tile < ntiles) fetch(tile);
  for (; tile < ntiles; tile += gridDim.x) {
    const int pb = tile % nblk, cg = (tile / nblk) % ncg, b = tile / (nblk * ncg);
    const int p0 = pb * RB, nr = min(RB, P - p0), seg = nr * W;
    const int ncl = min(CG, C - cg * CG);           // channels of this group
    __syncthreads();                                // previous tile's readers are done
#pragma unroll
    for (int k = 0; k < CPW; ++k)
#pragma unroll
      for (int j = 0; j < SL; ++j)
        if (wave + 4 * k < ncl && lane + 64 * j < seg) xs[(wave + 4 * k) * pitch + lane + 64 * j] = act_in ? gelu_f(v[k][j]) : v[k][j];
    __syncthreads();
    if (tile + (int)gridDim.x < ntiles) fetch(tile + gridDim.x);
    for (int it = threadIdx.x; it < nr * ncl; it += blockDim.x) {
      const int r = it / ncl, lc = it - r * ncl;
      const float* xr = xs + lc * pitch + r * W;
      float acc[2 * K2P];
#pragma unroll
      for (int j = 0; j < 2 * K2P; ++j) acc[j] = 0.f;
#pragma unroll 4
      for (int w = 0; w < W; ++w) {
        const float xv = xr[w];
        const float* t = tT + (size_t)w * 2 * K2P;  // wave-uniform: scalar loads
#pragma unroll
        for (int j = 0; j < 2 * K2P; ++j) acc[j] = fmaf(xv, t[j], acc[j]);
      }
      const size_t row = (size_t)b * P + p0 + r;
#pragma unroll
      for (int k2 = 0; k2 < K2P; ++k2)
        if (k2 < K2) x1[(row * K2 + k2) * C + cg * CG + lc] = make_float2(acc[2 * k2], acc[2 * k2 + 1]);
    }
  }
}

// Long-run variant: tile = 8 channels x RB rows with RB*W % 4 == 0 and P % RB == 0, so every channel run starts 16-byte
// aligned and is a whole number of float4: a wave owns 2 channels x 10 slots of 64 float4 (runs of <= 2560 floats).
// 16-byte requests: four times fewer outstanding misses per byte in flight than the dword staging above.
// smallest pitch > n that is 4 mod 32
static __host__ __device__ inline int rowdft4_pitch(int n) { return n + 1 + ((4 - (n + 1) % 32) + 32) % 32; }

template <int K2P>
__global__ void __launch_bounds__(256) k_rowdft_chan4(const float* __restrict__ x, float2* __restrict__ x1,
                                                      const float* __restrict__ tfwd, int C, int P, int W, int K2, int RB,
                                                      int ntiles, int act_in, int trows) {
  constexpr int CPW = 2, SL = 10, CG = 8, NJT = K2P / 8;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int seg = RB * W, seg4 = seg / 4;
  const int pitch = rowdft4_pitch(seg), TP = rowdft4_pitch((W + 3) & ~3);   // both = 4 mod 32: 2-way (minimal) conflicts on the MFMA operand reads
  float* xs = smem;                                 // [CG][pitch] + 4 floats of slack (k-padding reads)
  float* tab = smem + CG * pitch + 4;               // [16 * NJT][TP]   rows (2 k2, 2 k2 + 1) = (cos, -sin) of bin k2
  const int nblk = P / RB, ncg = (C + CG - 1) / CG;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int l15 = lane & 15, quad = lane >> 4;
  const size_t cstride = (size_t)P * W;
  for (int i = threadIdx.x; i < 16 * NJT * TP; i += blockDim.x) {
    const int j = i / TP, w = i - j * TP;
    tab[i] = (j < trows && w < W) ? tfwd[(size_t)j * W + w] : 0.f;
  }
  if (threadIdx.x < 4) xs[CG * pitch + threadIdx.x] = 0.f;
  for (int i = threadIdx.x; i < CG * (pitch - seg); i += blockDim.x)   // the pad of every channel run is read (times a zero table entry)
    xs[(i / (pitch - seg)) * pitch + seg + i % (pitch - seg)] = 0.f;
  float4 v[CPW][SL];
  auto fetch = [&](int tile) {
    const int pb = tile % nblk, cg = (tile / nblk) % ncg, b = tile / (nblk * ncg);
    const float* xb = x + ((size_t)b * C * P + (size_t)pb * RB) * W;
#pragma unroll
    for (int k = 0; k < CPW; ++k)
#pragma unroll
      for (int j = 0; j < SL; ++j)
        v[k][j] = ld4(xb + (size_t)min(cg * CG + wave + 4 * k, C - 1) * cstride + 4 * min(lane + 64 * j, seg4 - 1));
  };
  int tile = blockIdx.x;
  if (tile < ntiles) fetch(tile);
  const int nsteps = (W + 3) / 4;
  for (; tile < ntiles; tile += gridDim.x) {
    const int pb = tile % nblk, cg = (tile / nblk) % ncg, b = tile / (nblk * ncg);
    const int p0 = pb * RB;
    const int ncl = min(CG, C - cg * CG);
    __syncthreads();
#pragma unroll
    for (int k = 0; k < CPW; ++k)
#pragma unroll
      for (int j = 0; j < SL; ++j)
        if (lane + 64 * j < seg4) {                 // channels past ncl hold a clamped copy: finite filler, never stored
          float* d = xs + (wave + 4 * k) * pitch + 4 * (lane + 64 * j);
          float4 q = v[k][j];
          if (act_in) { q.x = gelu_f(q.x); q.y = gelu_f(q.y); q.z = gelu_f(q.z); q.w = gelu_f(q.w); }
          d[0] = q.x; d[1] = q.y; d[2] = q.z; d[3] = q.w;
        }
    __syncthreads();
    if (tile + (int)gridDim.x < ntiles) fetch(tile + gridDim.x);
    // X1^T[j][item] = sum_w T[j][w] x_item[w] on the fp32 matrix pipe: 16 table rows x 16 (row, channel) items per MFMA,
    // k = 4 consecutive... (quad <-> w = 4 s + quad); both operands are 4-byte LDS reads, no per-w scalar traffic
    const int nitems = RB * ncl;
    for (int g = wave; g * 16 < nitems; g += 4) {
      const int it = min(g * 16 + l15, nitems - 1);
      const int r = it / ncl, lc = it - r * ncl;
      const float* xr = xs + lc * pitch + r * W + quad;
#pragma unroll
      for (int jt = 0; jt < NJT; ++jt) {
        const float* tr = tab + (jt * 16 + l15) * TP + quad;
        f32x4 d0 = {0.f, 0.f, 0.f, 0.f}, d1 = {0.f, 0.f, 0.f, 0.f};
        int sidx = 0;
        for (; sidx + 1 < nsteps; sidx += 2) {
          d0 = mfma16(tr[4 * sidx], xr[4 * sidx], d0);
          d1 = mfma16(tr[4 * sidx + 4], xr[4 * sidx + 4], d1);
        }
        if (sidx < nsteps) d0 = mfma16(tr[4 * sidx], xr[4 * sidx], d0);
        if (g * 16 + l15 < nitems) {
          const size_t row = (size_t)b * P + p0 + r;
#pragma unroll
          for (int pr = 0; pr < 2; ++pr) {
            const int k2 = jt * 8 + quad * 2 + pr;
            if (k2 < K2) x1[(row * K2 + k2) * C + cg * CG + lc] = make_float2(d0[2 * pr] + d1[2 * pr], d0[2 * pr + 1] + d1[2 * pr + 1]);
          }
        }
      }
    }
  }
}

template <int K2P>
__global__ void __launch_bounds__(256) k_rowidft_chan(const float2* __restrict__ z, float* __restrict__ y,
                                                      const float* __restrict__ tT, const float* __restrict__ bias, int C,
                                                      int P, int W, int K2, int RB) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* ys = smem;                                 // [c][RB*W + 1]
  const int nblk = (P + RB - 1) / RB;
  const int b = blockIdx.x / nblk, p0 = (blockIdx.x % nblk) * RB;
  const int nr = min(RB, P - p0);
  const int pitch = RB * W + 1;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwave = blockDim.x >> 6;
  for (int it = threadIdx.x; it < nr * C; it += blockDim.x) {
    const int r = it / C, c = it - r * C;
    const float2* zr = z + ((size_t)b * P + p0 + r) * K2 * C + c;
    float zz[2 * K2P];
#pragma unroll
    for (int k2 = 0; k2 < K2P; ++k2) {
      const float2 v = k2 < K2 ? zr[(size_t)k2 * C] : make_float2(0.f, 0.f);
      zz[2 * k2] = v.x; zz[2 * k2 + 1] = v.y;
    }
    const float b0 = bias ? bias[c] : 0.f;
    float* yr = ys + c * pitch + r * W;
#pragma unroll 4
    for (int w = 0; w < W; ++w) {
      const float* t = tT + (size_t)w * 2 * K2P;    // wave-uniform: scalar loads
      float s0 = b0, s1 = 0.f, s2 = 0.f, s3 = 0.f;
#pragma unroll
      for (int j = 0; j < 2 * K2P; j += 4) {
        s0 = fmaf(zz[j], t[j], s0); s1 = fmaf(zz[j + 1], t[j + 1], s1);
        s2 = fmaf(zz[j + 2], t[j + 2], s2); s3 = fmaf(zz[j + 3], t[j + 3], s3);
      }
      yr[w] = (s0 + s1) + (s2 + s3);
    }
  }
  __syncthreads();
  const int seg = nr * W;
  const size_t cstride = (size_t)P * W;
  float* yb = y + ((size_t)b * C * P + p0) * W;
  for (int c = wave; c < C; c += nwave)
    for (int o = lane; o < seg; o += 64) yb[(size_t)c * cstride + o] = ys[c * pitch + o];
}

// The same pass with the truncated inverse DFT on the fp32 matrix cores (C a multiple of 32, W <= 128): per workgroup
// one real GEMM  Y[(r, c)][w] = sum_j Z[(r, c)][j] T[j][w]  (M = RB * C items, N = W, K = 2 * K2; j = 2 * bin + re/im)
// as v_mfma_f32_32x32x2_f32 tiles.  A operand straight from global memory: lane (l31, half) of M-tile (r, 32-channel
// block) loads the float2 spectrum value of its channel and keeps the re (half 0) or im (half 1) part - K step s = kept
// bin s.  B operand from an LDS copy of the (2 K2, W) table, zero-padded to whole 32-column tiles.  The VALU variant
// above paces its FMA loop by scalar table loads that share the lgkm counter with its LDS writes; here a wave issues
// K2 * ceil(W / 32) MFMAs per M-tile and the rest of the kernel is the store phase.
//   LDS: ys [C][RB*W + 1] | tab [2 K2][WP],  WP = 32 * ceil(W / 32) + 4
template <int K2P>
__global__ void __launch_bounds__(256) k_rowidft_chan_mfma(const float2* __restrict__ z, float* __restrict__ y,
                                                           const float* __restrict__ tinv, const float* __restrict__ bias,
                                                           int C, int P, int W, int K2, int RB) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int pitch = RB * W + 1;
  const int NTN = (W + 31) / 32, WP = NTN * 32 + 4;
  float* ys = smem;                                 // [c][pitch]
  float* tab = smem + C * pitch;                    // [2 K2][WP]
  const int nblk = (P + RB - 1) / RB;
  const int b = blockIdx.x / nblk, p0 = (blockIdx.x % nblk) * RB;
  const int nr = min(RB, P - p0);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwave = blockDim.x >> 6;
  const int l31 = lane & 31, half = lane >> 5;
  for (int i = threadIdx.x; i < 2 * K2 * WP; i += blockDim.x) {
    const int j = i / WP, w = i - j * WP;
    tab[i] = w < W ? tinv[(size_t)j * W + w] : 0.f;
  }
  __syncthreads();
  const int cb = C / 32;                            // 32-channel blocks per row
  for (int m = wave; m < nr * cb; m += nwave) {
    const int r = m / cb, c0 = (m - r * cb) * 32;
    const float2* zr = z + ((size_t)b * P + p0 + r) * K2 * C + c0 + l31;
    float av[K2P];
#pragma unroll
    for (int s = 0; s < K2P; ++s) {
      const float2 v = s < K2 ? zr[(size_t)s * C] : make_float2(0.f, 0.f);
      av[s] = half ? v.y : v.x;
    }
    float bv[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) bv[q] = bias ? bias[c0 + acc_row32(q, half)] : 0.f;
    for (int nt = 0; nt < NTN; ++nt) {
      f32x16 acc;
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[q] = bv[q];
      const float* tp = tab + half * WP + nt * 32 + l31;
#pragma unroll
      for (int s = 0; s < K2P; ++s)
        if (s < K2) acc = mfma32(av[s], tp[(2 * s) * WP], acc);
      const int w = nt * 32 + l31;
      if (w < W) {
        float* yp = ys + (c0 + 4 * half) * pitch + r * W + w;
#pragma unroll
        for (int q = 0; q < 16; ++q) yp[((q & 3) + 8 * (q >> 2)) * pitch] = acc[q];
      }
    }
  }
  __syncthreads();
  const int seg = nr * W;
  const size_t cstride = (size_t)P * W;
  float* yb = y + ((size_t)b * C * P + p0) * W;
  for (int c = wave; c < C; c += nwave)
    for (int o = lane; o < seg; o += 64) yb[(size_t)c * cstride + o] = ys[c * pitch + o];
}

// Same arithmetic, tiled along the FLATTENED plane instead of by rows: a workgroup owns floats [CH * j, CH * (j + 1)) of
// every channel plane (CH = 256 floats = eight 128-byte lines) and the <= ceil(CH / W) + 1 rows that overlap them.  Row
// runs of odd length (W = 73: 292 bytes) start at arbitrary byte offsets, so row tiles are written as unaligned dword
// stores whose first and last lines are shared with the neighbouring workgroups (on other XCDs); flat tiles are whole
// lines written once, by one 16-byte store per lane and channel.  Needs PW % 4 == 0 (16-byte aligned planes).
//   grid (ceil(PW / CH), B), LDS: ys [C][CH + 4] | tab [2 K2][WP]
constexpr int ROWFLAT_CH = 256;
template <int K2P>
__global__ void __launch_bounds__(512) k_rowidft_flat_mfma(const float2* __restrict__ z, float* __restrict__ y,
                                                           const float* __restrict__ tinv, const float* __restrict__ bias,
                                                           int C, int P, int W, int K2) {
  constexpr int CH = ROWFLAT_CH, YP = CH + 4;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int NTN = (W + 31) / 32, WP = NTN * 32 + 4;
  float* ys = smem;                                 // [c][YP]
  float* tab = smem + C * YP;                       // [2 K2][WP]
  const int b = blockIdx.y;
  const int PW = P * W;
  const int f0 = blockIdx.x * CH, f1 = min(f0 + CH, PW);
  const int r_lo = f0 / W, r_hi = (f1 - 1) / W;     // rows overlapping the tile
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwave = blockDim.x >> 6;
  const int l31 = lane & 31, half = lane >> 5;
  for (int i = threadIdx.x; i < 2 * K2 * WP; i += blockDim.x) {
    const int j = i / WP, w = i - j * WP;
    tab[i] = w < W ? tinv[(size_t)j * W + w] : 0.f;
  }
  __syncthreads();
  const int cb = C / 32;
  const int nm = (r_hi - r_lo + 1) * cb;
  float zn[K2P];                                    // next M-tile's spectrum values (this lane's re or im part), in
  {                                                 // flight during this tile's MFMAs
    const int m0 = wave < nm ? wave : 0;
    const float* zr = reinterpret_cast<const float*>(z + ((size_t)b * P + r_lo + m0 / cb) * K2 * C + (m0 % cb) * 32 + l31) + half;
#pragma unroll
    for (int s = 0; s < K2P; ++s) zn[s] = s < K2 ? zr[(size_t)s * C * 2] : 0.f;
  }
  for (int m = wave; m < nm; m += nwave) {
    const int r = r_lo + m / cb, c0 = (m % cb) * 32;
    float av[K2P];
#pragma unroll
    for (int s = 0; s < K2P; ++s) av[s] = zn[s];
    {
      const int mn = m + nwave < nm ? m + nwave : m;
      const float* zr = reinterpret_cast<const float*>(z + ((size_t)b * P + r_lo + mn / cb) * K2 * C + (mn % cb) * 32 + l31) + half;
#pragma unroll
      for (int s = 0; s < K2P; ++s) zn[s] = s < K2 ? zr[(size_t)s * C * 2] : 0.f;
    }
    float bv[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) bv[q] = bias ? bias[c0 + acc_row32(q, half)] : 0.f;
    for (int nt = 0; nt < NTN; ++nt) {
      const int w = nt * 32 + l31;
      const int f = r * W + w - f0;                 // position of (r, w) inside the tile
      if (r * W + nt * 32 >= f1 || r * W + nt * 32 + 31 < f0) continue;     // whole column block outside (wave-uniform)
      f32x16 acc;
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[q] = bv[q];
      const float* tp = tab + half * WP + nt * 32 + l31;
#pragma unroll
      for (int s = 0; s < K2P; ++s)
        if (s < K2) acc = mfma32(av[s], tp[(2 * s) * WP], acc);
      if (w < W && f >= 0 && f < f1 - f0) {
        float* yp = ys + (c0 + 4 * half) * YP + f;
#pragma unroll
        for (int q = 0; q < 16; ++q) yp[((q & 3) + 8 * (q >> 2)) * YP] = acc[q];
      }
    }
  }
  __syncthreads();
  float* yb = y + (size_t)b * C * PW + f0;
  if (f1 - f0 == CH) {
    for (int c = wave; c < C; c += nwave) st4(yb + (size_t)c * PW + 4 * lane, ld4(ys + c * YP + 4 * lane));
  } else {
    for (int c = wave; c < C; c += nwave)
      for (int o = lane; o < f1 - f0; o += 64) yb[(size_t)c * PW + o] = ys[c * YP + o];
  }
}

// dbias[c] partials: sum over (b, pixels) of dy (B, C, PW) -> part[blk][c]
__global__ void __launch_bounds__(256) k_channel_sums(const float* __restrict__ dy, float* __restrict__ part, int B,
                                                      int C, int PW) {
  // grid = (nchunk, C); block sums its chunk of (b, px) for channel c; 16-B loads when rows allow it
  const int c = blockIdx.y;
  float s = 0.f;
  if (PW % 4 == 0) {
    const size_t n4 = (size_t)B * (PW / 4);
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < n4; e += (size_t)gridDim.x * blockDim.x) {
      const size_t b = e / (PW / 4), p = e % (PW / 4);
      const float4 v = ld4(dy + (b * C + c) * PW + 4 * p);
      acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
    s = (acc.x + acc.y) + (acc.z + acc.w);
  } else {
    const size_t n = (size_t)B * PW;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (size_t)gridDim.x * blockDim.x) {
      const size_t b = e / PW, p = e % PW;
      s += dy[(b * C + c) * PW + p];
    }
  }
  for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
  __shared__ float sh[4];
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) part[(size_t)blockIdx.x * C + c] = sh[0] + sh[1] + sh[2] + sh[3];
}

// ---------------------------------------------------------------------------
// The whole middle of a block with ONE leading dim (2-D grids) in one launch: leading-axis truncating DFT -> per-mode
// channel contraction -> leading-axis inverse DFT (k_axis_fwd -> k_mode_gemm* -> k_axis_inv above).
// A workgroup owns one sample and 64 consecutive (last-dim bin, channel) columns = whole bins of all C channels, so the
// contraction over channels stays inside the workgroup; the spectrum of the kept modes never leaves LDS except for the
// copy `hat` the weight-gradient contraction reads later.
//   phase 1  Y[r][q]   = sum_n twT[n][r] X1[b][n][q]              the n range split over the 8 waves, combined through LDS
//   phase 2  O[r][k2,o] = sum_j Y[r][k2,j] (conj) M[(r,k2)][j][o]  M = packed weights [k][i][o] (forward) or their
//                                                                  transposed copy [k][o][i] with conj (adjoint): the lane
//                                                                  runs along the contiguous index, the j range is split
//                                                                  over the waves; the weights stream through L2 (they are
//                                                                  shared by all samples), two modes' rows in flight
//                                                                  while the previous two are used
//   phase 3  Z[b][n][q] = sum_r twi[n][r] O[r][q]                  rows n split over the waves
// Measured (BASELINE config 2, 384 workgroups, two per CU; in-kernel stamps, tools/trace_mid.py): every workgroup starts
// together, so the three phases run in lock step over the whole chip - 25 MB of HBM reads, then 151 MB of weight rows out
// of L2, then 25 MB of HBM writes, ~9 + 6 + 4 us with nothing overlapping: 29 us per launch against 32 us for the three
// separate launches (rocprofv3; ~4.5 us of each launch is the fixed cost of a launch on this stack).  What it buys is
// launches: BASELINE config 1 (launch-bound, hipGraph) runs 0.217 instead of 0.255 ms per step.
// Round 5: the complex multiply-adds of the three phases as two v_pk_fma_f32 each (table entry broadcast through op_sel on
// src0, bit-identical results) change nothing - phase 1 14.0-14.9 k cycles against 14.4-14.6 k, contraction 12.5-14.4 k
// against 14.3-16 k, phase 3 7.3 k against 7.1 k, the launch 33.0 against 31.7 us on one box (HIP events): phases 1 and 3
// wait for HBM (50 MB in ~9 us), the contraction for L2 (151 MB of weight rows in ~6 us = 25 TB/s); none of them for the
// vector unit.  The scalar form stays.
//   block (64, 8), grid (inner / 64, samples), LDS (8 + 2) * NK * 64 float2
#ifndef FNO_MID_SMEM_TABLE
#define FNO_MID_SMEM_TABLE 1      // 1 = DFT table rows through scalar loads (round 5; up to 12 kept modes: 16 need more scalar
                                  // registers than a wave has); 0 = staged in LDS, broadcast reads (A/B arm)
#endif
#ifndef FNO_MID_SKIP
#define FNO_MID_SKIP 0        // timing experiments: 1 = no contraction, 2 = no phase 1 loads, 4 = no phase 3
#endif
#ifdef FNO_TRACE
#define MID_STAMP(slot) do { if (blockIdx.x == 2 && blockIdx.y == gridDim.y / 2 && threadIdx.x == 0) \
    g_trace[threadIdx.y * 256 + (slot)] = __builtin_readcyclecounter(); \
    if (blockIdx.x == 0 && threadIdx.x == 0 && threadIdx.y == 0 && ((slot) == 0 || (slot) == 7)) \
    g_trace[8 * 256 + blockIdx.y * 2 + ((slot) == 7)] = __builtin_readcyclecounter(); \
    if (blockIdx.x == 1 && threadIdx.x == 0 && threadIdx.y == 0 && blockIdx.y < 64) /* every sample's workgroup 1, wave 0: all stamps */ \
    g_trace[10 * 256 + blockIdx.y * 8 + (slot)] = __builtin_readcyclecounter(); } while (0)
#else
#define MID_STAMP(slot) do { } while (0)
#endif
template <bool B> struct MidFlag { static constexpr bool value = B; };
// acc += w * v as complex numbers, w UNIFORM (a table entry in a scalar register pair): two packed FMAs with the factor taken
// from the pair's low / high word through op_sel on src0 (SGPR sources and src0 selections are outside the gfx950 hazard:
// fno_dev.h).  vs = (-v.y, v.x).  Same products in the same order as four scalar FMAs: bit-identical.
template <bool SREG>
FNO_DEV void cfma_uniform(f32x2& acc, float2 w, f32x2 v, f32x2 vs) {
  if constexpr (SREG) {
    const f32x2 wp = {w.x, w.y};
    asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]" : "+v"(acc) : "s"(wp), "v"(v));
    asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,1,1]" : "+v"(acc) : "s"(wp), "v"(vs));
  } else {
    acc[0] = fmaf(w.x, v[0], acc[0]); acc[0] = fmaf(w.y, vs[0], acc[0]);
    acc[1] = fmaf(w.x, v[1], acc[1]); acc[1] = fmaf(w.y, vs[1], acc[1]);
  }
}
template <int NK, int C>
__global__ void __launch_bounds__(512, 4) k_spec_mid(const float2* __restrict__ x1, float2* __restrict__ hat,
                                                     const float2* __restrict__ wm, float2* __restrict__ z,
                                                     const float2* __restrict__ twT, const float2* __restrict__ twi, int n,
                                                     int inner, int K2, int conj_w, int Bm, size_t w_ms) {
  static_assert(C == 32 || C == 64, "whole bins per workgroup");
  constexpr bool SREG = FNO_MID_SMEM_TABLE && NK <= 12;      // table rows in scalar registers
  constexpr int SEGS = 8, JW = C / SEGS, NTH = 64 * SEGS;
  constexpr int RB = 2, NG = NK / RB;               // RB modes' weight rows per load group, double-buffered
  static_assert(NK % RB == 0, "mode groups");
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float2* sh = reinterpret_cast<float2*>(smem);     // [SEGS][NK][64] partial sums (phases 1 and 2); the (n, NK) tables
                                                    // of phases 1 and 3 are staged here while the sums are not live
  float2* ys = sh + SEGS * NK * 64;                 // [NK][64] truncated spectrum of the input
  float2* os = ys + NK * 64;                        // [NK][64] contracted spectrum
  const int ql = threadIdx.x;
  const int seg = __builtin_amdgcn_readfirstlane(threadIdx.y);
  const int tid = seg * 64 + ql;
  const int q = blockIdx.x * 64 + ql;
  const int o = blockIdx.y;
  const int k2 = q / C, ch = q - k2 * C, lb = (ql / C) * C;     // lb: first lane of this lane's bin
  const float2* wb = wm + (size_t)(o / Bm) * w_ms + ((size_t)k2 * C + seg * JW) * C + ch;
  MID_STAMP(0);
  // Table rows are the same for every lane.  Round 2 staged them in LDS (through the scalar cache, one row at a time, a wave
  // waited ~1000 cycles per row); round 5 found what that costs: a broadcast ds_read_b64 still occupies the LDS unit for four
  // cycles, 192 of them per wave and phase = 12 k cycles per phase on the CUs that hold two workgroups - the phases were bound
  // by LDS issue, which is why halving their vector instructions alone changed nothing.  Now a row's NK entries come through
  // scalar loads one row AHEAD of their use (2 x NK scalar register pairs), and the multiply-adds are packed.
  if constexpr (!SREG) for (int i = tid; i < n * NK; i += NTH) sh[i] = twT[i];
  // the first weight group does not depend on phase 1: in flight from here on
  float2 wv[2][RB][JW];
  auto load_w = [&](int g, int buf) {
#pragma unroll
    for (int rr = 0; rr < RB; ++rr)
#pragma unroll
      for (int j = 0; j < JW; ++j) wv[buf][rr][j] = wb[((size_t)(g * RB + rr) * K2 * C + j) * C];
  };
  if (!(FNO_MID_SKIP & 1)) load_w(0, 0);
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
        v[j] = (nn < n && !(FNO_MID_SKIP & 2)) ? src[(size_t)nn * inner] : make_float2(0.f, 0.f);
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
    if (hat) hat[((size_t)o * NK + r) * inner + q] = make_float2(sx, sy);
  }
  __syncthreads();
  MID_STAMP(3);
  // ---- phase 2 ----
  if (!(FNO_MID_SKIP & 1))
  {
    const float sg = conj_w ? -1.f : 1.f;
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      if (g + 1 < NG) load_w(g + 1, (g + 1) & 1);
#pragma unroll
      for (int rr = 0; rr < RB; ++rr) {
        float ax = 0.f, ay = 0.f;
#pragma unroll
        for (int j = 0; j < JW; ++j) {
          const float2 y = ys[(g * RB + rr) * 64 + lb + seg * JW + j];
          const float2 w = wv[g & 1][rr][j];
          const float wy = sg * w.y;
          ax = fmaf(y.x, w.x, ax); ax = fmaf(-y.y, wy, ax);
          ay = fmaf(y.x, wy, ay); ay = fmaf(y.y, w.x, ay);
        }
        sh[(seg * NK + g * RB + rr) * 64 + ql] = make_float2(ax, ay);
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
  if constexpr (!SREG) {
    for (int i = tid; i < n * NK; i += NTH) sh[i] = twi[i];
    __syncthreads();
  }
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
        dst[(size_t)r * inner] = make_float2(s2[0], s2[1]);
        if (PAIRS || r + SEGS < n) {
#pragma unroll
          for (int k = 0; k < NK; ++k) ta[k] = twi[(size_t)min(r + 2 * SEGS, n - 1) * NK + k];
          f32x2 s3 = {0.f, 0.f};
#pragma unroll
          for (int k = 0; k < NK; ++k) cfma_uniform<true>(s3, tb[k], v[k], vs[k]);
          dst[(size_t)(r + SEGS) * inner] = make_float2(s3[0], s3[1]);
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
