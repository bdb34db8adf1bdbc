"""libs/envs/control_env.py surface, the part the observer training loop calls: the channel-flow right-hand side and the
physics-informed loss (NSControlEnvMatlab.compute_rhs_py :429-530, pde_loss :627-633, load_state :149-180).

The reference class is a MATLAB-backed CFD environment (RK3 stepper, Poisson projection, opposition control, rewards); none
of that is on the training hot path and none of it is here.  What the `pde_loss_weight` branch of the loop needs
(run_pde_observers.py:53-57, 226-231) is an object with the grid metrics and these two methods; they run on the HIP
engine (fno_chanflow_*), one launch per batch instead of ~1500 slice kernels per sample."""
import numpy as np
import torch

from ... import functional as F


class ChannelFlowRHS:
    """Grid metrics + compute_rhs_py / pde_loss with the reference's names and argument meaning.

    Build it from the same `.mat` initial condition the reference loads (`from_mat`, control_env.py:149-168) or from an
    analytic tanh grid (`tanh_channel`, libs/matlab_codes/main.m:13-22)."""
    default_nu = 3.076923076923077e-04      # control_env.py:26
    default_re = 178.1899                   # :27
    default_dPdx = 0.57231059E-01 ** 2      # :161

    def __init__(self, Nx, Nz, dx, dz, y, ym, yg=None, Re=-1.0, dPdx=None):
        y, ym = np.asarray(y, dtype=np.float64).reshape(-1), np.asarray(ym, dtype=np.float64).reshape(-1)
        if yg is None:
            yg = np.concatenate(([-ym[0]], ym, [2 + ym[0]]))                 # :165
        self.nu = self.default_nu * (self.default_re / Re) if Re > 0 else self.default_nu      # :26-29
        self.dPdx = self.default_dPdx if dPdx is None else float(dPdx)
        self.Nx, self.Ny, self.Nz = int(Nx), int(y.shape[0]), int(Nz)
        self.dx, self.dz = float(np.asarray(dx).reshape(-1)[0]), float(np.asarray(dz).reshape(-1)[0])
        self.y, self.ym, self.yg = y, ym, np.asarray(yg, dtype=np.float64).reshape(-1)
        self.grid = F.ChannelGrid(self.Nx, self.Nz, self.dx, self.dz, self.y, self.ym, self.yg, self.nu)

    @classmethod
    def from_mat(cls, load_path, Re=-1.0):
        """The grid of a reference initial-condition file (x, y, z, ym; Nx = len(x) - 2, Nz = len(z) - 2)."""
        import scipy.io
        m = scipy.io.loadmat(load_path, mat_dtype=True)
        x, y, z, ym = (np.asarray(m[k], dtype=np.float64).reshape(-1) for k in ("x", "y", "z", "ym"))
        return cls(len(x) - 2, len(z) - 2, x[1] - x[0], z[1] - z[0], y, ym, Re=Re)

    @classmethod
    def tanh_channel(cls, Nx=32, Ny=130, Nz=32, Lx=2 * np.pi, Lz=2 * np.pi, stretch=2.6, Re=-1.0):
        y = 1 + np.tanh(stretch * np.linspace(-1, 1, Ny)) / np.tanh(stretch)
        return cls(Nx, Nz, Lx / Nx, Lz / Nz, y, 0.5 * (y[1:] + y[:-1]), Re=Re)

    # -- the reference's per-field signatures (fields (Nx, Ny[+1], Nz)); a leading batch dimension is accepted too
    def compute_rhs_py(self, U, V, W, dPdx=None):
        if dPdx is None:
            dPdx = self.dPdx
        if U.dim() == 3:
            return tuple(f[0] for f in F.chanflow_rhs(self.grid, U[None], V[None], W[None], dPdx))
        return F.chanflow_rhs(self.grid, U, V, W, dPdx)

    def pde_loss(self, U, Vgt, V, W, dPdx=None):
        """||Fu_gt - Fu_pred|| + ||Fv_gt - Fv_pred|| + ||Fw_gt - Fw_pred||, summed over the batch when the fields carry
        one (the loop at run_pde_observers.py:228-230 in one call).  dPdx cancels in the difference; the argument is kept
        for signature parity."""
        if U.dim() == 3:
            U, Vgt, V, W = U[None], Vgt[None], V[None], W[None]
        return F.chanflow_pde_loss(self.grid, U.float(), Vgt.float(), V.float(), W.float())
