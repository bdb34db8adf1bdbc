"""CPU oracle for the PINO residual loss.

TEST INFRASTRUCTURE ONLY - never imported by the product path.

Restates, in our own words, libs/envs/diff_control_env.py:5-60 (the same code appears in
libs/pino_utils/losses.py:68-104, 246-262) as called by train_pino.py:79-111:
  FDM_NS_vorticity(w, v, t_interval) -> Du   and   Channelflow_PINO_loss / PINO_loss3d -> (loss_ic, loss_f).
Pinned by tests/golden/pino_loss_*.npz, produced by importing the reference file itself
(oracle/make_golden.py::gen_pino_loss).
"""
import math

import torch


def wavenumbers(n, device=None, dtype=torch.float32):
    """Index -> signed wavenumber exactly as the reference builds it (diff_control_env.py:15-21):
    [0, 1, .., n/2 - 1, -n/2, .., -1]; index n/2 carries -n/2."""
    k_max = n // 2
    return torch.cat((torch.arange(0, k_max, device=device), torch.arange(-k_max, 0, device=device))).to(dtype)


def ns_vorticity_residual(w, visc, t_interval=1.0):
    """Du = w_t + u . grad(w) - nu * lap(w) on interior time levels   (diff_control_env.py:5-41).
    w (B, N, N, T) real, visc (B,).  Spectral derivatives over (x, y), central difference in t.
      w_h = fft2(w) over dims (1, 2)                                         :12
      lap = kx^2 + ky^2 with lap[0, 0] := 1 ; psi_h = w_h / lap              :22-24
      ux_h = i ky psi_h, uy_h = -i kx psi_h, wx_h = i kx w_h, wy_h = i ky w_h, wlap_h = -lap w_h   :26-30
      irfft2 over dims (1, 2) of the columns [0, n/2]                         :32-36
    """
    B, n, _, nt = w.shape
    w = w.reshape(B, n, n, nt)
    k = wavenumbers(n, w.device, w.dtype)
    kx = k.reshape(1, n, 1, 1)
    ky = k.reshape(1, 1, n, 1)
    lap = kx ** 2 + ky ** 2
    lap = lap.clone()
    lap[0, 0, 0, 0] = 1.0
    w_h = torch.fft.fft2(w, dim=[1, 2])
    psi_h = w_h / lap
    half = n // 2 + 1

    def back(spec):
        return torch.fft.irfft2(spec[:, :, :half], dim=[1, 2])

    ux = back(1j * ky * psi_h)
    uy = back(-1j * kx * psi_h)
    wx = back(1j * kx * w_h)
    wy = back(1j * ky * w_h)
    wlap = back(-lap * w_h)
    dt = t_interval / (nt - 1)
    wt = (w[..., 2:] - w[..., :-2]) / (2 * dt)                                # :38-39
    return wt + (ux * wx + uy * wy - visc.reshape(B, 1, 1, 1) * wlap)[..., 1:-1]   # :41


def lp_rel_mean(x, y):
    """LpLoss(size_average=True).rel (libs/pino_utils/losses.py LpLoss): mean_b ||x_b - y_b|| / ||y_b||."""
    B = x.shape[0]
    d = torch.norm(x.reshape(B, -1) - y.reshape(B, -1), 2, 1)
    return torch.mean(d / torch.norm(y.reshape(B, -1), 2, 1))


def forcing(n, device=None):
    """get_forcing (libs/pino_utils/losses.py:288-291): -4 cos(4 y), y = 2 pi j / n, shape (1, n, n, 1)."""
    y = torch.arange(n, device=device, dtype=torch.float32) * (2 * math.pi / n)
    return (-4 * torch.cos(4 * y)).reshape(1, 1, n, 1).repeat(1, n, 1, 1)


def pino_loss(u, u0, f, visc, t_interval=1.0):
    """(loss_ic, loss_f) of Channelflow_PINO_loss (diff_control_env.py:44-60)."""
    B, n, _, nt = u.shape
    loss_ic = lp_rel_mean(u[..., 0], u0)
    du = ns_vorticity_residual(u, visc, t_interval)
    loss_f = lp_rel_mean(du, f.repeat(B, 1, 1, nt - 2))
    return loss_ic, loss_f
