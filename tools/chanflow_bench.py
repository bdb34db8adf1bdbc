"""Channel-flow RHS / physics-informed loss: engine kernels vs the same arithmetic as torch ops on the GPU.
GPU box:  python tools/chanflow_bench.py [B]
Two torch baselines: `vectorised` (every y-loop of the reference written as one slice expression: the best plain torch
can do) and `row loops` (the reference's structure, control_env.py:448-449 etc.: one small kernel group per y row, per sample,
two RHS evaluations per sample)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from pde_policylearning_amd import functional as F
from pde_policylearning_amd.libs.envs.control_env import ChannelFlowRHS

dev = torch.device("cuda", 0)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
Nx, Ny, Nz = 32, 130, 32
env = ChannelFlowRHS.tanh_channel(Nx, Ny, Nz)
torch.manual_seed(0)
U = 1 + 0.5 * torch.randn(B, Nx, Ny + 1, Nz, device=dev)
W = 0.3 * torch.randn(B, Nx, Ny + 1, Nz, device=dev)
Vgt = 0.3 * torch.randn(B, Nx, Ny, Nz, device=dev)
V = (Vgt + 0.05 * torch.randn_like(Vgt)).requires_grad_(True)
dy = torch.tensor(np.diff(env.y), device=dev, dtype=torch.float32)[None, None, :, None]
dym = torch.tensor(np.diff(env.ym), device=dev, dtype=torch.float32)[None, None, :, None]
dyg = torch.tensor(np.diff(env.yg), device=dev, dtype=torch.float32)[None, None, :, None]
dx, dz, nu = env.dx, env.dz, env.nu
r = torch.roll


def rhs_vec(U, V, W, dpdx):
    ub, wb = 0.5 * (U[:, :, :-1] + U[:, :, 1:]), 0.5 * (W[:, :, :-1] + W[:, :, 1:])
    UV, VW = 0.5 * (V + r(V, 1, 1)) * ub, 0.5 * (V + r(V, 1, 3)) * wb
    lap = lambda A: nu * (r(A, -1, 1) - 2 * A + r(A, 1, 1)) / dx ** 2 + nu * (r(A, -1, 3) - 2 * A + r(A, 1, 3)) / dz ** 2
    pad = lambda c, like: torch.nn.functional.pad(c, (0, 0, 1, like.shape[2] - 1 - c.shape[2]))
    UU = (0.5 * (U + r(U, -1, 1))) ** 2
    UW = 0.5 * (W + r(W, 1, 1)) * 0.5 * (U + r(U, 1, 3))
    Fu = -(UU - r(UU, 1, 1)) / dx - (r(UW, -1, 3) - UW) / dz + lap(U) + dpdx / 2
    Fu = Fu + pad(-(UV[:, :, 1:] - UV[:, :, :-1]) / dy + nu * ((U[:, :, 2:] - U[:, :, 1:-1]) / dyg[:, :, 1:] - (U[:, :, 1:-1] - U[:, :, :-2]) / dyg[:, :, :-1]) / dy, Fu)
    VV = (0.5 * (V[:, :, :-1] + V[:, :, 1:])) ** 2
    Fv = -(r(UV, -1, 1) - UV) / dx - (r(VW, -1, 3) - VW) / dz + lap(V)
    Fv = Fv + pad(-(VV[:, :, 1:] - VV[:, :, :-1]) / dym + nu * ((V[:, :, 2:] - V[:, :, 1:-1]) / dy[:, :, 1:] - (V[:, :, 1:-1] - V[:, :, :-2]) / dy[:, :, :-1]) / dym, Fv)
    WW = (0.5 * (W + r(W, -1, 3))) ** 2
    Fw = -(r(UW, -1, 1) - UW) / dx - (WW - r(WW, 1, 3)) / dz + lap(W)
    Fw = Fw + pad(-(VW[:, :, 1:] - VW[:, :, :-1]) / dy + nu * ((W[:, :, 2:] - W[:, :, 1:-1]) / dyg[:, :, 1:] - (W[:, :, 1:-1] - W[:, :, :-2]) / dyg[:, :, :-1]) / dy, Fw)
    return Fu, Fv, Fw


def rhs_rowloop(U, V, W, dpdx):
    """one field; the wall-normal terms added row by row like the reference does"""
    U, V, W = U[None], V[None], W[None]
    ub, wb = 0.5 * (U[:, :, :-1] + U[:, :, 1:]), 0.5 * (W[:, :, :-1] + W[:, :, 1:])
    UV, VW = 0.5 * (V + r(V, 1, 1)) * ub, 0.5 * (V + r(V, 1, 3)) * wb
    lap = lambda A: nu * (r(A, -1, 1) - 2 * A + r(A, 1, 1)) / dx ** 2 + nu * (r(A, -1, 3) - 2 * A + r(A, 1, 3)) / dz ** 2
    UU = (0.5 * (U + r(U, -1, 1))) ** 2
    UW = 0.5 * (W + r(W, 1, 1)) * 0.5 * (U + r(U, 1, 3))
    Fu = -(UU - r(UU, 1, 1)) / dx - (r(UW, -1, 3) - UW) / dz + lap(U) + dpdx / 2
    VV = (0.5 * (V[:, :, :-1] + V[:, :, 1:])) ** 2
    Fv = -(r(UV, -1, 1) - UV) / dx - (r(VW, -1, 3) - VW) / dz + lap(V)
    WW = (0.5 * (W + r(W, -1, 3))) ** 2
    Fw = -(r(UW, -1, 1) - UW) / dx - (WW - r(WW, 1, 3)) / dz + lap(W)
    y, ym, yg = env.y, env.ym, env.yg
    for A, F_, X in ((U, Fu, UV), (W, Fw, VW)):
        for i in range(1, Ny):
            F_[:, :, i] -= (X[:, :, i] - X[:, :, i - 1]) / (y[i] - y[i - 1])
        for i in range(1, Ny):
            F_[:, :, i] += nu * ((A[:, :, i + 1] - A[:, :, i]) / (yg[i + 1] - yg[i]) - (A[:, :, i] - A[:, :, i - 1]) / (yg[i] - yg[i - 1])) / (y[i] - y[i - 1])
    for i in range(1, Ny - 1):
        Fv[:, :, i] -= (VV[:, :, i] - VV[:, :, i - 1]) / (ym[i] - ym[i - 1])
    for i in range(1, Ny - 1):
        Fv[:, :, i] += nu * ((V[:, :, i + 1] - V[:, :, i]) / (y[i + 1] - y[i]) - (V[:, :, i] - V[:, :, i - 1]) / (y[i] - y[i - 1])) / (ym[i] - ym[i - 1])
    return Fu[0], Fv[0], Fw[0]


def timeit(fn, n):
    fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e3


def loss_vec():
    V.grad = None
    a, b = rhs_vec(U, Vgt, W, 0.0), rhs_vec(U, V, W, 0.0)
    sum((p - q).flatten(1).norm(dim=1).sum() for p, q in zip(a, b)).backward()


def loss_rowloop():
    V.grad = None
    tot = 0
    for s in range(B):
        a, b = rhs_rowloop(U[s], Vgt[s], W[s], 0.0), rhs_rowloop(U[s], V[s], W[s], 0.0)
        tot = tot + sum((p - q).norm() for p, q in zip(a, b))
    tot.backward()


def loss_engine():
    V.grad = None
    F.chanflow_pde_loss(env.grid, U, Vgt, V, W).backward()


fbytes = 4 * B * Nx * Nz
rhs_bytes = fbytes * (3 * (Ny + 1) + 3 * Ny + 2)            # U, V, W in; Fu, Fv, Fw out
loss_bytes = fbytes * ((4 * Ny + 2) + (3 * Ny + 2)) + fbytes * ((4 * Ny + 2) + (3 * Ny + 2) + Ny)   # fwd: 4 in, 3 D out; bwd: 4 + 3 D in, dV out
if os.environ.get("ENGINE_ONLY"):          # for rocprofv3 --kernel-trace --stats: just the engine launches
    with torch.no_grad():
        Ud, Vd, Wd = U.double(), V.detach().double(), W.double()
        t32 = timeit(lambda: F.chanflow_rhs(env.grid, U, V, W, env.dPdx), 50)
        t64 = timeit(lambda: F.chanflow_rhs(env.grid, Ud, Vd, Wd, env.dPdx), 10)
    print(f"engine only: rhs f32 {t32*1e3:.1f} us, rhs f64 {t64*1e3:.1f} us, pde_loss fwd+bwd {timeit(loss_engine, 50)*1e3:.1f} us")
    sys.exit(0)
with torch.no_grad():
    t_e = timeit(lambda: F.chanflow_rhs(env.grid, U, V, W, env.dPdx), 50)
    t_v = timeit(lambda: rhs_vec(U, V, W, env.dPdx), 20)
    t_r = timeit(lambda: [rhs_rowloop(U[s], V[s], W[s], env.dPdx) for s in range(B)], 1)
    got, want = F.chanflow_rhs(env.grid, U, V, W, env.dPdx), rhs_vec(U, V, W, env.dPdx)
    err = max(float((a - b).norm() / b.norm()) for a, b in zip(got, want))
print(f"RHS  B={B} {Nx}x{Ny}x{Nz} fp32: engine {t_e*1e3:.1f} us ({rhs_bytes/t_e/1e6:.0f} GB/s of algorithmic traffic) | torch vectorised {t_v:.3f} ms | "
      f"torch row loops {t_r:.1f} ms | engine vs vectorised rel-L2 {err:.1e}")
l_e = timeit(loss_engine, 50)
g_e = V.grad.clone()
l_v = timeit(loss_vec, 10)
g_v = V.grad.clone()
l_r = timeit(loss_rowloop, 1)
print(f"pde_loss fwd+bwd: engine {l_e*1e3:.1f} us ({loss_bytes/l_e/1e6:.0f} GB/s) | torch vectorised {l_v:.3f} ms | torch row loops {l_r:.1f} ms | "
      f"dV engine vs vectorised rel-L2 {float((g_e-g_v).norm()/g_v.norm()):.1e}")
