"""Dialect-A spectral convolution with the reference's constructor and state_dict
surface (neuralop/models/spectral_convolution.py:143-347, dense path)."""
import torch
from torch import nn

from ... import functional as F


class DenseComplexWeight(nn.Module):
    """Complex (Cin, Cout, m1..mN) weight stored as a real (.., 2) parameter named
    `tensor` (the role tltorch's ComplexDense FactorizedTensor plays in the reference,
    spectral_convolution.py:253-268; tltorch's own leaf name is unpinned)."""
    name = "ComplexDense"

    def __init__(self, shape):
        super().__init__()
        self.shape = tuple(shape)
        self.tensor = nn.Parameter(torch.zeros(*shape, 2))

    def normal_(self, mean=0.0, std=1.0):
        with torch.no_grad():
            self.tensor.normal_(mean, std)
        return self

    def to_tensor(self):
        return torch.view_as_complex(self.tensor)


def _unsupported(what):
    raise NotImplementedError(
        f"fnoengine SpectralConv: {what} is outside the accelerated hot path "
        "(only the dense, non-separable, fixed-mode configuration that FNO2d/FNO3d use is built)")


class SpectralConv(nn.Module):
    """Drop-in for FactorizedSpectralConv(in_channels, out_channels, n_modes, ...) on its
    dense configuration (factorization=None).  forward(x, indices) computes
    irfftn(pad(W . rfftn(x)[corners])) + bias[indices] in the HIP engine."""

    def __init__(self, in_channels, out_channels, n_modes, incremental_n_modes=None, bias=True,
                 n_layers=1, separable=False, output_scaling_factor=None,
                 rank=0.5, factorization=None, implementation='reconstructed',
                 fixed_rank_modes=False, joint_factorization=False, decomposition_kwargs=dict(),
                 init_std='auto', fft_norm='backward'):
        super().__init__()
        if factorization is not None and 'dense' not in str(factorization).lower():
            _unsupported(f"factorization={factorization!r}")
        if separable:
            _unsupported("separable=True")
        if incremental_n_modes is not None:
            _unsupported("incremental_n_modes")
        if output_scaling_factor is not None:
            _unsupported("output_scaling_factor")
        if joint_factorization:
            _unsupported("joint_factorization=True")
        self.in_channels = in_channels
        self.out_channels = out_channels
        if isinstance(n_modes, int):
            n_modes = [n_modes]
        self.n_modes = list(n_modes)
        self.order = len(self.n_modes)
        # kept extent per corner = n_modes // 2 on every dim (spectral_convolution.py:202-203)
        self.half_n_modes = [m // 2 for m in self.n_modes]
        self.half_total_n_modes = list(self.half_n_modes)
        self.incremental_n_modes = None
        self.output_scaling_factor = None
        self.n_layers = n_layers
        self.fft_norm = fft_norm
        self.separable = False
        self.n_weights_per_layer = 2 ** (self.order - 1)
        init_std = (1 / (in_channels * out_channels)) if init_std == 'auto' else 0.02
        shape = (in_channels, out_channels, *self.half_n_modes)
        self.weight = nn.ModuleList([DenseComplexWeight(shape)
                                     for _ in range(self.n_weights_per_layer * n_layers)])
        for w in self.weight:
            w.normal_(0, init_std)
        if bias:
            self.bias = nn.Parameter(init_std * torch.randn(*((n_layers, out_channels) + (1,) * self.order)))
        else:
            self.bias = None

    def layer_weights(self, indices):
        n = self.n_weights_per_layer
        return [self.weight[n * indices + i].tensor for i in range(n)]

    def forward(self, x, indices=0):
        b = self.bias[indices] if self.bias is not None else None
        return F.spectral_conv(x.float(), self.layer_weights(indices), b, self.half_n_modes, self.fft_norm)


FactorizedSpectralConv = SpectralConv
