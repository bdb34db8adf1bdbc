"""Dialect-C spectral convolutions with the reference surface
(libs/models/pino_models/basics.py:64-96 SpectralConv2d, :99-143 SpectralConv3d): complex64
parameters weights1..N initialised scale * U[0,1), torch-default ('backward') FFT norm, no bias.
forward() runs in the HIP engine."""
import torch
from torch import nn

from .... import functional as F


def _cweight(cin, cout, *modes, planes=False):
    """the reference's initial values (scale * U[0, 1) drawn in the reference's order); planes: stored with the last dim
    outermost in memory (functional.plane_major) - shape, values and every tensor operation on it are unchanged"""
    w = (1.0 / (cin * cout)) * torch.rand(cin, cout, *modes, dtype=torch.cfloat)
    return nn.Parameter(F.to_plane_major(w) if planes else w)


class SpectralConv2d(nn.Module):
    def __init__(self, in_channels, out_channels, modes1, modes2):
        super().__init__()
        self.in_channels, self.out_channels, self.modes1, self.modes2 = in_channels, out_channels, modes1, modes2
        self.scale = 1 / (in_channels * out_channels)
        self.weights1 = _cweight(in_channels, out_channels, modes1, modes2)
        self.weights2 = _cweight(in_channels, out_channels, modes1, modes2)

    def direct_grad_params(self):
        """parameters whose gradient the engine may write straight into a trainer.FlatGradBucket (one use per step)"""
        return [self.weights1, self.weights2]

    def forward(self, x):
        return F.spectral_conv(x, [self.weights1, self.weights2], None, (self.modes1, self.modes2), "backward",
                               direct_grads=getattr(self, "_direct_grads", False))


class SpectralConv3d(nn.Module):
    def __init__(self, in_channels, out_channels, modes1, modes2, modes3):
        super().__init__()
        self.in_channels, self.out_channels = in_channels, out_channels
        self.modes1, self.modes2, self.modes3 = modes1, modes2, modes3
        self.scale = 1 / (in_channels * out_channels)
        for i in range(1, 5):
            # plane-major: only min(Nz/2+1, modes3) last-dim slices ever see data (:119-139 of the reference file); with the
            # last dim outermost they are one contiguous prefix of the tensor - the part the engine packs, Adam steps and
            # the gradient exchange sends (PINObserverFullField at T = 1: 1/12 of 906 MB)
            setattr(self, f"weights{i}", _cweight(in_channels, out_channels, modes1, modes2, modes3, planes=True))

    def direct_grad_params(self):
        return [self.weights1, self.weights2, self.weights3, self.weights4]

    def _announce_live(self, k3):
        """k3 last-dim modes are about to be read (and receive a gradient): trainer.live_last_of plans the live-slice
        gradient exchange from it, and an optimizer that skips the dead slices (trainer.FusedAdam) brings them up to date
        BEFORE a longer last dimension reads them."""
        guard = self.__dict__.get("_dead_slice_guard")
        if guard is not None:
            guard(self, k3)
        self._live_last = k3

    def engine_call(self, x):
        """(corner weights in the engine's order, kept modes, stored last-dim extent) for an input of x's shape"""
        k3 = min(x.shape[-1] // 2 + 1, self.modes3)
        self._announce_live(k3)
        return [self.weights1, self.weights3, self.weights2, self.weights4], (self.modes1, self.modes2, k3), self.modes3

    def forward(self, x):
        # reference corner order: weights1 (lo,lo), weights2 (hi,lo), weights3 (lo,hi), weights4 (hi,hi)
        # (basics.py:125-139); the engine takes (lo,lo), (lo,hi), (hi,lo), (hi,hi).  Only
        # min(Nz/2+1, modes3) last-dim modes are live (:119,:125-126); the rest get zero gradient.
        k3 = min(x.shape[-1] // 2 + 1, self.modes3)
        self._announce_live(k3)
        return F.spectral_conv(x, [self.weights1, self.weights3, self.weights2, self.weights4], None,
                               (self.modes1, self.modes2, k3), "backward", weight_last_extent=self.modes3,
                               direct_grads=getattr(self, "_direct_grads", False))
