"""PINO residual loss with the reference surface (libs/pino_utils/losses.py:68-104, 246-262, 288-291):
FDM-in-time / spectral-in-space Navier-Stokes vorticity residual and the two relative-L2 terms the
fine-tuning loop (train_pino.py:98-101) combines.  The arithmetic runs in the HIP engine
(functional.pino_loss -> fno_pino_loss_*); there is no CPU path."""
import math

import torch

from ... import functional as F


def get_forcing(S):
    """-4 cos(4 y) on the periodic grid y_j = 2 pi j / S, shape (1, S, S, 1)  (losses.py:288-291)."""
    y = torch.arange(S, dtype=torch.float32) * (2 * math.pi / S)
    return (-4 * torch.cos(4 * y)).reshape(1, 1, S, 1).repeat(1, S, 1, 1)


def PINO_loss3d(u, u0, forcing, v=1 / 40, t_interval=1.0):
    """(loss_ic, loss_f): LpLoss(size_average=True) of u[..., 0] vs u0 and of the residual vs the forcing."""
    B, nx, ny, nt = u.size(0), u.size(1), u.size(2), u.size(3)
    u = u.reshape(B, nx, ny, nt)
    if not torch.is_tensor(v):
        v = torch.full((B,), float(v), dtype=torch.float32, device=u.device)
    return F.pino_loss(u, u0.reshape(B, nx, ny), forcing.to(u.device), v.reshape(B).to(u.device), t_interval)


Channelflow_PINO_loss = PINO_loss3d      # libs/envs/diff_control_env.py:44-60 is the same computation
