// Channel-flow Navier-Stokes right-hand side on the staggered grid and the physics-informed loss built on it
// (reference: NSControlEnvMatlab.compute_rhs_py libs/envs/control_env.py:429-530, pde_loss :627-633).
//
// One field: U, W (Nx, Ny+1, Nz), V (Nx, Ny, Nz), z contiguous; x and z periodic, y wall-normal with the non-uniform
// metrics y (faces), ym (centres), yg (ghost-extended centres).  The reference walks y in Python (six loops of ~Ny slice
// updates per call, two calls per sample); here one workgroup owns an (sample, x) slab and every output point is one
// gather over its 3x3x3 neighbourhood, so the whole RHS is a single pass over three fields.  HBM-bound by construction:
// 3 fields in, 3 out per sample, neighbours come from L1/L2 (a slab and its two x-neighbours are 3 x 17 KB).
//
// The loss never forms the two right-hand sides: F(U, Vgt, W) - F(U, V, W) only keeps the terms that contain V, and every one
// of them is linear in E = Vgt - V except d(vv)/dy, which is (E_j + E_j+1)(S_j + S_j+1)/4 with S = Vgt + V.  Working on E
// directly avoids the cancellation the reference's subtraction of two full fp32 right-hand sides suffers.
#pragma once
#include <hip/hip_runtime.h>

#include "fno_dev.h"

FNO_DEV float chanflow_wave_sum(float v) {
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

// metrics: three zero-padded arrays of MP = Ny + 2 doubles:  rdy[j] = 1/(y[j]-y[j-1]) (1 <= j <= Ny-1),
// rdym[j] = 1/(ym[j]-ym[j-1]) (1 <= j <= Ny-2), rdyg[j] = 1/(yg[j]-yg[j-1]) (1 <= j <= Ny); zero elsewhere.
struct ChanflowGeo {
  int B, Nx, Ny, Nz;
  double rdx, rdz, nu;
  const double* metrics;
};

template <typename T>
struct ChanflowRhsArgs {
  const T *U, *V, *W, *dPdx;      // dPdx: per-sample device array or null
  T *Fu, *Fv, *Fw;
  T dPdx_default;
};

template <typename T>
__global__ __launch_bounds__(256) void k_chanflow_rhs(ChanflowGeo g, ChanflowRhsArgs<T> a) {
  const int Nx = g.Nx, Ny = g.Ny, Nz = g.Nz, MP = Ny + 2;
  const int b = blockIdx.x / Nx, i = blockIdx.x % Nx;
  const int ixm = (i + Nx - 1) % Nx, ixp = (i + 1) % Nx;
  const size_t su = (size_t)(Ny + 1) * Nz, sv = (size_t)Ny * Nz;
  const T* U = a.U + (size_t)b * Nx * su;
  const T* V = a.V + (size_t)b * Nx * sv;
  const T* W = a.W + (size_t)b * Nx * su;
  const T *U0 = U + i * su, *Um = U + ixm * su, *Up = U + ixp * su;
  const T *V0 = V + i * sv, *Vm = V + ixm * sv, *Vp = V + ixp * sv;
  const T *W0 = W + i * su, *Wm = W + ixm * su, *Wp = W + ixp * su;
  T* Fu = a.Fu + ((size_t)b * Nx + i) * su;
  T* Fv = a.Fv + ((size_t)b * Nx + i) * sv;
  T* Fw = a.Fw + ((size_t)b * Nx + i) * su;
  const T rdx = (T)g.rdx, rdz = (T)g.rdz, nu = (T)g.nu, h = (T)0.5;
  const T nx2 = nu * rdx * rdx, nz2 = nu * rdz * rdz;
  const T half_dpdx = h * (a.dPdx ? a.dPdx[b] : a.dPdx_default);
  const double *rdy = g.metrics, *rdym = g.metrics + MP, *rdyg = g.metrics + 2 * MP;

  const int npts = (Ny + 1) * Nz, per = (npts + gridDim.y - 1) / gridDim.y, hi = min(npts, (int)(blockIdx.y + 1) * per);
  for (int idx = blockIdx.y * per + threadIdx.x; idx < hi; idx += blockDim.x) {
    const int j = idx / Nz, k = idx - j * Nz;
    const int kzm = k ? k - 1 : Nz - 1, kzp = (k + 1 == Nz) ? 0 : k + 1;
    const int r = j * Nz;                       // row offset (same for the U/W and V layouts)
    const bool yin = j >= 1 && j <= Ny - 1;     // rows that carry the wall-normal terms of Fu, Fw
    const T u = U0[r + k], w = W0[r + k];
    // ---- Fu
    {
      const T uxp = Up[r + k], uxm = Um[r + k], uzp = U0[r + kzp], uzm = U0[r + kzm];
      const T uu1 = h * (u + uxp), uu0 = h * (uxm + u);
      T f = -(uu1 * uu1 - uu0 * uu0) * rdx;
      const T uw0 = (h * (w + Wm[r + k])) * (h * (u + uzm));
      const T uw1 = (h * (W0[r + kzp] + Wm[r + kzp])) * (h * (uzp + u));
      f -= (uw1 - uw0) * rdz;
      f += nx2 * (uxp - 2 * u + uxm) + nz2 * (uzp - 2 * u + uzm);
      if (yin) {
        const T ujp = U0[r + Nz + k], ujm = U0[r - Nz + k];
        const T uv1 = (h * (V0[r + k] + Vm[r + k])) * (h * (u + ujp));
        const T uv0 = (h * (V0[r - Nz + k] + Vm[r - Nz + k])) * (h * (ujm + u));
        const T ry = (T)rdy[j];
        f -= (uv1 - uv0) * ry;
        f += nu * ((ujp - u) * (T)rdyg[j + 1] - (u - ujm) * (T)rdyg[j]) * ry;
      }
      Fu[r + k] = f + half_dpdx;
    }
    // ---- Fw
    {
      const T wxp = Wp[r + k], wxm = Wm[r + k], wzp = W0[r + kzp], wzm = W0[r + kzm];
      const T uw0 = (h * (w + wxm)) * (h * (u + U0[r + kzm]));
      const T uw1 = (h * (wxp + w)) * (h * (Up[r + k] + Up[r + kzm]));
      T f = -(uw1 - uw0) * rdx;
      const T ww1 = h * (w + wzp), ww0 = h * (wzm + w);
      f -= (ww1 * ww1 - ww0 * ww0) * rdz;
      f += nx2 * (wxp - 2 * w + wxm) + nz2 * (wzp - 2 * w + wzm);
      if (yin) {
        const T wjp = W0[r + Nz + k], wjm = W0[r - Nz + k];
        const T vw1 = (h * (V0[r + k] + V0[r + kzm])) * (h * (w + wjp));
        const T vw0 = (h * (V0[r - Nz + k] + V0[r - Nz + kzm])) * (h * (wjm + w));
        const T ry = (T)rdy[j];
        f -= (vw1 - vw0) * ry;
        f += nu * ((wjp - w) * (T)rdyg[j + 1] - (w - wjm) * (T)rdyg[j]) * ry;
      }
      Fw[r + k] = f;
    }
    // ---- Fv (Ny rows)
    if (j < Ny) {
      const T v = V0[r + k], vxp = Vp[r + k], vxm = Vm[r + k], vzp = V0[r + kzp], vzm = V0[r + kzm];
      const T ujp = U0[r + Nz + k], wjp = W0[r + Nz + k];
      const T uv0 = (h * (v + vxm)) * (h * (u + ujp));
      const T uv1 = (h * (vxp + v)) * (h * (Up[r + k] + Up[r + Nz + k]));
      T f = -(uv1 - uv0) * rdx;
      const T vw0 = (h * (v + vzm)) * (h * (w + wjp));
      const T vw1 = (h * (vzp + v)) * (h * (W0[r + kzp] + W0[r + Nz + kzp]));
      f -= (vw1 - vw0) * rdz;
      f += nx2 * (vxp - 2 * v + vxm) + nz2 * (vzp - 2 * v + vzm);
      if (j >= 1 && j <= Ny - 2) {
        const T vjp = V0[r + Nz + k], vjm = V0[r - Nz + k];
        const T vv1 = h * (v + vjp), vv0 = h * (vjm + v);
        const T rm = (T)rdym[j];
        f -= (vv1 * vv1 - vv0 * vv0) * rm;
        f += nu * ((vjp - v) * (T)rdy[j + 1] - (v - vjm) * (T)rdy[j]) * rm;
      }
      Fv[r + k] = f;
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------------
// pde_loss = sum_b ( ||Du_b|| + ||Dv_b|| + ||Dw_b|| ),  D = F(U, Vgt, W) - F(U, V, W)
// ---------------------------------------------------------------------------------------------------------------------------
struct ChanflowLossArgs {
  const float *U, *Vgt, *V, *W;
  float *Du, *Dv, *Dw;           // saved difference fields: Du, Dw (B, Nx, Ny+1, Nz), Dv (B, Nx, Ny, Nz)
  float* partial;                // [B*Nx][3] sums of squares
  float* inv_norm;               // [B][3]
  float* loss;
  const float* gloss;            // upstream gradient (device scalar) or null
  float* dV;
};

__global__ __launch_bounds__(256) void k_chanflow_diff(ChanflowGeo g, ChanflowLossArgs a) {
  const int Nx = g.Nx, Ny = g.Ny, Nz = g.Nz, MP = Ny + 2;
  const int b = blockIdx.x / Nx, i = blockIdx.x % Nx;
  const int ixm = (i + Nx - 1) % Nx, ixp = (i + 1) % Nx;
  const size_t su = (size_t)(Ny + 1) * Nz, sv = (size_t)Ny * Nz;
  const float* U = a.U + (size_t)b * Nx * su;
  const float* W = a.W + (size_t)b * Nx * su;
  const float *U0 = U + i * su, *Up = U + ixp * su, *W0 = W + i * su;
  const size_t vb = (size_t)b * Nx * sv;
  const float *G0 = a.Vgt + vb + i * sv, *Gm = a.Vgt + vb + ixm * sv, *Gp = a.Vgt + vb + ixp * sv;
  const float *P0 = a.V + vb + i * sv, *Pm = a.V + vb + ixm * sv, *Pp = a.V + vb + ixp * sv;
  float* Du = a.Du + ((size_t)b * Nx + i) * su;
  float* Dw = a.Dw + ((size_t)b * Nx + i) * su;
  float* Dv = a.Dv + ((size_t)b * Nx + i) * sv;
  const float rdx = (float)g.rdx, rdz = (float)g.rdz, nu = (float)g.nu;
  const float nx2 = nu * rdx * rdx, nz2 = nu * rdz * rdz;
  const double *rdy = g.metrics, *rdym = g.metrics + MP;
  float su2 = 0.f, sv2 = 0.f, sw2 = 0.f;

  const int npts = (Ny + 1) * Nz, per = (npts + gridDim.y - 1) / gridDim.y, hi = min(npts, (int)(blockIdx.y + 1) * per);
  for (int idx = blockIdx.y * per + threadIdx.x; idx < hi; idx += blockDim.x) {
    const int j = idx / Nz, k = idx - j * Nz;
    const int kzm = k ? k - 1 : Nz - 1, kzp = (k + 1 == Nz) ? 0 : k + 1;
    const int r = j * Nz;
    float du = 0.f, dw = 0.f;
    if (j >= 1 && j <= Ny - 1) {
      const float e = G0[r + k] - P0[r + k], em = G0[r - Nz + k] - P0[r - Nz + k];
      const float u = U0[r + k], w = W0[r + k];
      const float ry = (float)rdy[j];
      const float uv1 = 0.5f * (e + (Gm[r + k] - Pm[r + k])) * (0.5f * (u + U0[r + Nz + k]));
      const float uv0 = 0.5f * (em + (Gm[r - Nz + k] - Pm[r - Nz + k])) * (0.5f * (U0[r - Nz + k] + u));
      du = -(uv1 - uv0) * ry;
      const float vw1 = 0.5f * (e + (G0[r + kzm] - P0[r + kzm])) * (0.5f * (w + W0[r + Nz + k]));
      const float vw0 = 0.5f * (em + (G0[r - Nz + kzm] - P0[r - Nz + kzm])) * (0.5f * (W0[r - Nz + k] + w));
      dw = -(vw1 - vw0) * ry;
    }
    Du[r + k] = du;
    Dw[r + k] = dw;
    su2 += du * du;
    sw2 += dw * dw;
    if (j < Ny) {
      const float gv = G0[r + k], pv = P0[r + k];
      const float e = gv - pv;
      const float exp_ = Gp[r + k] - Pp[r + k], exm = Gm[r + k] - Pm[r + k];
      const float ezp = G0[r + kzp] - P0[r + kzp], ezm = G0[r + kzm] - P0[r + kzm];
      const float ub0 = 0.5f * (U0[r + k] + U0[r + Nz + k]), ub1 = 0.5f * (Up[r + k] + Up[r + Nz + k]);
      const float wb0 = 0.5f * (W0[r + k] + W0[r + Nz + k]), wb1 = 0.5f * (W0[r + kzp] + W0[r + Nz + kzp]);
      float d = -(0.5f * (exp_ + e) * ub1 - 0.5f * (e + exm) * ub0) * rdx;
      d -= (0.5f * (ezp + e) * wb1 - 0.5f * (e + ezm) * wb0) * rdz;
      d += nx2 * (exp_ - 2.f * e + exm) + nz2 * (ezp - 2.f * e + ezm);
      if (j >= 1 && j <= Ny - 2) {
        const float gp = G0[r + Nz + k], pp = P0[r + Nz + k], gm = G0[r - Nz + k], pm = P0[r - Nz + k];
        const float ep = gp - pp, em = gm - pm;
        const float rm = (float)rdym[j];
        const float vv1 = 0.25f * (e + ep) * ((gv + pv) + (gp + pp));
        const float vv0 = 0.25f * (em + e) * ((gm + pm) + (gv + pv));
        d -= (vv1 - vv0) * rm;
        d += nu * ((ep - e) * (float)rdy[j + 1] - (e - em) * (float)rdy[j]) * rm;
      }
      Dv[r + k] = d;
      sv2 += d * d;
    }
  }
  __shared__ float red[3][4];
  su2 = chanflow_wave_sum(su2);
  sv2 = chanflow_wave_sum(sv2);
  sw2 = chanflow_wave_sum(sw2);
  const int wv = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) { red[0][wv] = su2; red[1][wv] = sv2; red[2][wv] = sw2; }
  __syncthreads();
  if (threadIdx.x < 3)
    a.partial[((size_t)blockIdx.x * gridDim.y + blockIdx.y) * 3 + threadIdx.x] = red[threadIdx.x][0] + red[threadIdx.x][1] + red[threadIdx.x][2] + red[threadIdx.x][3];
}

// one workgroup: per-sample norms (double accumulation over the Nx slab partials), their reciprocals for the backward, the loss
__global__ __launch_bounds__(256) void k_chanflow_finish(int B, int Nx /* partials per sample */, const float* partial, float* inv_norm, float* loss) {
  double acc = 0.0;
  for (int t = threadIdx.x; t < B * 3; t += blockDim.x) {
    const int b = t / 3, c = t - 3 * b;
    double s = 0.0;
    for (int i = 0; i < Nx; ++i) s += (double)partial[((size_t)b * Nx + i) * 3 + c];
    const double n = sqrt(s);
    inv_norm[t] = n > 0.0 ? (float)(1.0 / n) : 0.f;      // torch: the subgradient of ||.|| at 0 is 0
    acc += n;
  }
  __shared__ double red[256];
  red[threadIdx.x] = acc;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
    __syncthreads();
  }
  if (threadIdx.x == 0) *loss = (float)red[0];
}

// dV = gloss * sum_c  (dD_c/dV)^T (D_c / ||D_c||): the adjoint stencil, gathered per V point
__global__ __launch_bounds__(256) void k_chanflow_diff_bwd(ChanflowGeo g, ChanflowLossArgs a) {
  const int Nx = g.Nx, Ny = g.Ny, Nz = g.Nz, MP = Ny + 2;
  const int b = blockIdx.x / Nx, i = blockIdx.x % Nx;
  const int ixm = (i + Nx - 1) % Nx, ixp = (i + 1) % Nx;
  const size_t su = (size_t)(Ny + 1) * Nz, sv = (size_t)Ny * Nz;
  const float* U = a.U + (size_t)b * Nx * su;
  const float* W = a.W + (size_t)b * Nx * su;
  const float *U0 = U + i * su, *Up = U + ixp * su, *W0 = W + i * su;
  const size_t vb = (size_t)b * Nx * sv, ub = (size_t)b * Nx * su;
  const float *G0 = a.Vgt + vb + i * sv, *P0 = a.V + vb + i * sv;
  const float *Au0 = a.Du + ub + i * su, *Aup = a.Du + ub + ixp * su;
  const float *Aw0 = a.Dw + ub + i * su;
  const float *Av0 = a.Dv + vb + i * sv, *Avm = a.Dv + vb + ixm * sv, *Avp = a.Dv + vb + ixp * sv;
  float* dV = a.dV + vb + i * sv;
  const float gl = a.gloss ? *a.gloss : 1.f;
  const float cu = a.inv_norm[b * 3 + 0], cv = a.inv_norm[b * 3 + 1], cw = a.inv_norm[b * 3 + 2];
  const float rdx = (float)g.rdx, rdz = (float)g.rdz, nu = (float)g.nu;
  const float nx2 = nu * rdx * rdx, nz2 = nu * rdz * rdz;
  const double *rdy = g.metrics, *rdym = g.metrics + MP;

  const int npts = Ny * Nz, per = (npts + gridDim.y - 1) / gridDim.y, hi = min(npts, (int)(blockIdx.y + 1) * per);
  for (int idx = blockIdx.y * per + threadIdx.x; idx < hi; idx += blockDim.x) {
    const int j = idx / Nz, k = idx - j * Nz;
    const int kzm = k ? k - 1 : Nz - 1, kzp = (k + 1 == Nz) ? 0 : k + 1;
    const int r = j * Nz;
    const int rm = j ? r - Nz : r, rp = (j + 1 < Ny) ? r + Nz : r;        // clamped V-layout neighbours (their weights vanish)
    const float ry0 = (float)rdy[j], ry1 = (float)rdy[j + 1];           // rdy[0] = rdy[Ny] = 0
    const float rm0 = (float)rdym[j], rm1 = (float)rdym[j + 1];         // rdym[0] = rdym[Ny-1] = rdym[Ny] = 0
    const float rmm = j ? (float)rdym[j - 1] : 0.f;
    const float av = cv * Av0[r + k];
    const float avxm = cv * Avm[r + k], avxp = cv * Avp[r + k], avzm = cv * Av0[r + kzm], avzp = cv * Av0[r + kzp];
    // adjoint of dUV at (i, j, k) and (xp, j, k)
    const float auv0 = cu * (-Au0[r + k] * ry0 + Au0[r + Nz + k] * ry1) + (av - avxm) * rdx;
    const float auv1 = cu * (-Aup[r + k] * ry0 + Aup[r + Nz + k] * ry1) + (avxp - av) * rdx;
    const float ub0 = 0.5f * (U0[r + k] + U0[r + Nz + k]), ub1 = 0.5f * (Up[r + k] + Up[r + Nz + k]);
    float ge = 0.5f * (ub0 * auv0 + ub1 * auv1);
    // adjoint of dVW at (i, j, k) and (i, j, zp)
    const float avw0 = cw * (-Aw0[r + k] * ry0 + Aw0[r + Nz + k] * ry1) + (av - avzm) * rdz;
    const float avw1 = cw * (-Aw0[r + kzp] * ry0 + Aw0[r + Nz + kzp] * ry1) + (avzp - av) * rdz;
    const float wb0 = 0.5f * (W0[r + k] + W0[r + Nz + k]), wb1 = 0.5f * (W0[r + kzp] + W0[r + Nz + kzp]);
    ge += 0.5f * (wb0 * avw0 + wb1 * avw1);
    ge += nx2 * (avxp - 2.f * av + avxm) + nz2 * (avzp - 2.f * av + avzm);
    // wall-normal viscous term: c[j] = nu a_v[j] rdym[j]
    const float avjm = cv * Av0[rm + k], avjp = cv * Av0[rp + k];
    const float c0 = nu * av * rm0, cm = nu * avjm * rmm, cp = nu * avjp * rm1;
    ge += cm * ry0 - c0 * (ry1 + ry0) + cp * ry1;
    // d(vv)/dy: the only term that sees S = Vgt + V
    float gs = 0.f;
    const float e = G0[r + k] - P0[r + k], s = G0[r + k] + P0[r + k];
    if (j <= Ny - 2) {
      const float avv = -av * rm0 + avjp * rm1;                              // adjoint of dVV(j)
      const float ep = G0[r + Nz + k] - P0[r + Nz + k], sp = G0[r + Nz + k] + P0[r + Nz + k];
      ge += 0.25f * (s + sp) * avv;
      gs += 0.25f * (e + ep) * avv;
    }
    if (j >= 1) {
      const float avv = -avjm * rmm + av * rm0;                              // adjoint of dVV(j-1)
      const float em = G0[r - Nz + k] - P0[r - Nz + k], sm = G0[r - Nz + k] + P0[r - Nz + k];
      ge += 0.25f * (sm + s) * avv;
      gs += 0.25f * (em + e) * avv;
    }
    dV[r + k] = gl * (gs - ge);                                              // dE/dV = -1, dS/dV = +1
  }
}
