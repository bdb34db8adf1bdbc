"""Dialect-A spectral convolution with the reference's constructor and state_dict
surface (neuralop/models/spectral_convolution.py:143-347, dense path)."""
import itertools
import warnings

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

    def _load_from_state_dict(self, state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys, error_msgs):
        """Checkpoints of the reference carry this weight under whatever leaf name its tltorch version gives the dense
        factorized tensor (unpinned: tltorch is neither vendored nor version-locked, spectral_convolution.py:253-268) and
        possibly as a complex tensor.  Any SINGLE entry below this module's prefix is taken as the weight; a complex value
        is stored as its real (.., 2) view."""
        key = prefix + "tensor"
        if key not in state_dict:
            cands = [k for k in state_dict if k.startswith(prefix)]
            if len(cands) == 1:
                state_dict[key] = state_dict.pop(cands[0])
        v = state_dict.get(key)
        if torch.is_tensor(v) and v.is_complex():
            state_dict[key] = torch.view_as_real(v.resolve_conj().contiguous())
        super()._load_from_state_dict(state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys, error_msgs)


def _unsupported(what):
    raise NotImplementedError(
        f"fnoengine SpectralConv: {what} is outside the accelerated hot path "
        "(only the dense, non-separable, fixed-mode configuration that FNO2d/FNO3d use is built)")


class SpectralConv(nn.Module):
    """Drop-in for FactorizedSpectralConv(in_channels, out_channels, n_modes, ...) on its
    dense configuration (factorization=None).  forward(x, indices) computes
    irfftn(pad(W . rfftn(x)[corners])) + bias[indices] in the HIP engine.

    Constructor combinations beside the accelerated one:
    * `incremental_n_modes` stays on the engine: the kept extent shrinks and the corner weights are sliced
      (spectral_convolution.py:270-298);
    * `separable=True` (per-channel weights, `x * w`, :38-41) and `output_scaling_factor` (inverse transform onto a
      resized grid, :338-342) have no HIP kernel: forward composes them from torch operations on the same device
      (`_torch_composition`; rocFFT-backed, several times slower than the engine path - a warning says so once);
    * CP / Tucker / TT factorizations and `joint_factorization` raise: their parameters live in tltorch containers this
      image does not have, so there is nothing a torch composition could be checked against."""

    def __init__(self, in_channels, out_channels, n_modes, incremental_n_modes=None, bias=True,
                 n_layers=1, separable=False, output_scaling_factor=None,
                 rank=0.5, factorization=None, implementation='reconstructed',
                 fixed_rank_modes=False, joint_factorization=False, decomposition_kwargs=dict(),
                 init_std='auto', fft_norm='backward'):
        super().__init__()
        if factorization is not None and 'dense' not in str(factorization).lower():
            _unsupported(f"factorization={factorization!r}")
        if joint_factorization:
            _unsupported("joint_factorization=True")
        if separable and in_channels != out_channels:
            raise ValueError('To use separable Fourier Conv, in_channels must be equal to out_channels, '
                             f'but got in_channels={in_channels} and out_channels={out_channels}.')
        self.in_channels = in_channels
        self.out_channels = out_channels
        if isinstance(n_modes, int):
            n_modes = [n_modes]
        self.n_modes = list(n_modes)
        self.order = len(self.n_modes)
        # kept extent per corner = n_modes // 2 on every dim (spectral_convolution.py:202-203)
        self.half_total_n_modes = [m // 2 for m in self.n_modes]
        self.incremental_n_modes = incremental_n_modes
        if output_scaling_factor is not None:
            if isinstance(output_scaling_factor, (float, int)):
                output_scaling_factor = [[float(output_scaling_factor)] * self.order] * n_layers
            elif isinstance(output_scaling_factor[0], (float, int)):
                output_scaling_factor = [[s] * self.order for s in output_scaling_factor]
        self.output_scaling_factor = output_scaling_factor
        self.n_layers = n_layers
        self.fft_norm = fft_norm
        self.separable = bool(separable)
        self.n_weights_per_layer = 2 ** (self.order - 1)
        init_std = (1 / (in_channels * out_channels)) if init_std == 'auto' else 0.02
        shape = ((in_channels,) if separable else (in_channels, out_channels)) + tuple(self.half_total_n_modes)
        self.weight = nn.ModuleList([DenseComplexWeight(shape)
                                     for _ in range(self.n_weights_per_layer * n_layers)])
        for w in self.weight:
            w.normal_(0, init_std)
        if bias:
            self.bias = nn.Parameter(init_std * torch.randn(*((n_layers, out_channels) + (1,) * self.order)))
        else:
            self.bias = None

    @property
    def incremental_n_modes(self):
        return self._incremental_n_modes

    @incremental_n_modes.setter
    def incremental_n_modes(self, value):
        # may be changed while training (spectral_convolution.py:276-298)
        if value is None:
            self._incremental_n_modes = None
            self.half_n_modes = [m // 2 for m in self.n_modes]
            return
        if isinstance(value, int):
            value = [value] * len(self.n_modes)
        if len(value) != len(self.n_modes):
            raise ValueError(f'Provided {value} for actual n_modes={self.n_modes}.')
        self._incremental_n_modes = list(value)
        self.half_n_modes = [m // 2 for m in value]

    def layer_weights(self, indices):
        n = self.n_weights_per_layer
        ws = [self.weight[n * indices + i].tensor for i in range(n)]
        if self._incremental_n_modes is not None:
            lead = 1 if self.separable else 2
            cut = (slice(None),) * lead + tuple(slice(None, h) for h in self.half_n_modes)
            ws = [w[cut].contiguous() for w in ws]       # autograd scatters the gradient back into the full weight
        return ws

    def _torch_composition(self, x, indices):
        """forward for `separable` / `output_scaling_factor` from torch operations on x's device (no HIP kernel covers
        them): real FFT, one contraction per kept corner, inverse FFT at the (possibly resized) output grid."""
        F._require_cuda(x, "x")          # still no CPU path
        global _WARNED
        if not _WARNED:
            _WARNED = True
            warnings.warn("fnoengine SpectralConv: separable / output_scaling_factor run as a torch composition "
                          "(no fused HIP kernel for these options)", RuntimeWarning, stacklevel=3)
        sizes = list(x.shape[2:])
        dims = list(range(-self.order, 0))
        xf = torch.fft.rfftn(x.float(), norm=self.fft_norm, dim=dims)
        out = torch.zeros(x.shape[0], self.out_channels, *sizes[:-1], sizes[-1] // 2 + 1,
                          device=x.device, dtype=torch.cfloat)
        h = self.half_n_modes
        ws = self.layer_weights(indices)
        for i, hi in enumerate(itertools.product(*([(False, True)] * (self.order - 1)))):
            sl = (slice(None), slice(None)) + tuple(slice(-m, None) if up else slice(None, m)
                                                    for m, up in zip(h[:-1], hi)) + (slice(None, h[-1]),)
            w = torch.view_as_complex(ws[i])
            if self.separable:
                out[sl] = xf[sl] * w
            else:
                out[sl] = torch.einsum("bi...,io...->bo...", xf[sl], w)
        if self.output_scaling_factor is not None:
            sizes = [int(round(s * r)) for s, r in zip(sizes, self.output_scaling_factor[indices])]
        y = torch.fft.irfftn(out, s=sizes, norm=self.fft_norm)
        if self.bias is not None:
            y = y + self.bias[indices]
        return y

    def forward(self, x, indices=0):
        if self.separable or self.output_scaling_factor is not None:
            return self._torch_composition(x, indices)
        b = self.bias[indices] if self.bias is not None else None
        return F.spectral_conv(x.float(), self.layer_weights(indices), b, self.half_n_modes, self.fft_norm)


_WARNED = False
FactorizedSpectralConv = SpectralConv
