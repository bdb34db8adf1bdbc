"""PINO residual loss fwd+bwd: engine vs the torch.fft op sequence of the reference on the same GPU
(shipped config 5 shape: B 4, 128 x 128, T 65).  GPU box: python tools/pino_loss_bench.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pde_policylearning_amd import functional as F, _lib
import math


class P:
    """The reference's op sequence (libs/envs/diff_control_env.py:5-60) in torch ops, for the timing comparison only."""

    @staticmethod
    def forcing(n, device=None):
        y = torch.arange(n, device=device, dtype=torch.float32) * (2 * math.pi / n)
        return (-4 * torch.cos(4 * y)).reshape(1, 1, n, 1).repeat(1, n, 1, 1)

    @staticmethod
    def pino_loss(u, u0, f, visc, t_interval):
        B, n, _, nt = u.shape
        k = torch.cat((torch.arange(0, n // 2, device=u.device), torch.arange(-(n // 2), 0, device=u.device))).float()
        kx, ky = k.reshape(1, n, 1, 1), k.reshape(1, 1, n, 1)
        lap = (kx ** 2 + ky ** 2).clone()
        lap[0, 0, 0, 0] = 1.0
        w_h = torch.fft.fft2(u, dim=[1, 2])
        psi = w_h / lap
        back = lambda sp: torch.fft.irfft2(sp[:, :, :n // 2 + 1], dim=[1, 2])
        ux, uy = back(1j * ky * psi), back(-1j * kx * psi)
        wx, wy, wlap = back(1j * kx * w_h), back(1j * ky * w_h), back(-lap * w_h)
        dt = t_interval / (nt - 1)
        du = (u[..., 2:] - u[..., :-2]) / (2 * dt) + (ux * wx + uy * wy - visc.reshape(B, 1, 1, 1) * wlap)[..., 1:-1]
        rel = lambda a, b: torch.mean(torch.norm((a - b).reshape(B, -1), 2, 1) / torch.norm(b.reshape(B, -1), 2, 1))
        return rel(u[..., 0], u0), rel(du, f.repeat(B, 1, 1, nt - 2))
dev = torch.device("cuda", 0)
def timeit(fn, n=20, w=3):
    for _ in range(w): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
for (B, n, nt) in ((4, 128, 65), (4, 64, 65), (16, 128, 17)):
    u = torch.randn(B, n, n, nt, device=dev, requires_grad=True)
    u0 = torch.randn(B, n, n, device=dev)
    visc = torch.rand(B, device=dev) * 0.01 + 0.002
    f = P.forcing(n, dev)
    def eng():
        u.grad = None
        a, b = F.pino_loss(u, u0, f, visc, 0.5); (5 * a + b).backward()
    def ref():
        u.grad = None
        a, b = P.pino_loss(u, u0, f, visc, 0.5); (5 * a + b).backward()
    te, tr = timeit(eng), timeit(ref)
    L = _lib.lib(); L.fno_profile_reset(); L.fno_profile_enable(1); eng(); torch.cuda.synchronize(); L.fno_profile_enable(0)
    prof = ", ".join(f"{k} {ms/c*1e3:.0f}us" for k, ms, c in _lib.profile_summary()); L.fno_profile_reset()
    mb = u.numel() * 4 / 1e6
    print(f"B{B} {n}x{n} T{nt} ({mb:.0f} MB): engine {te:.3f} ms   torch.fft on GPU {tr:.3f} ms   x{tr/te:.1f}   [{prof}]")
