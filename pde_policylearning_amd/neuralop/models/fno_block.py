"""FNOBlocks with the reference surface (neuralop/models/fno_block.py:10-170), default
path only: linear (bias-free 1x1 conv) skip, no MLP, no norm, no preactivation."""
import itertools

import torch
import torch.nn.functional as TF
from torch import nn

from .spectral_convolution import SpectralConv, _unsupported


def _resample(x, scales):
    """The skip branch on the resized grid of `output_scaling_factor` (fno_block.py:132-134 -> resample.py:6-56):
    linear / antialiased bicubic interpolation with aligned corners in 1-D / 2-D, Fourier zero-padding or truncation
    in 3-D.  Torch operations: this option has no HIP kernel (SpectralConv._torch_composition)."""
    old = x.shape[-len(scales):]
    new = tuple(int(round(s * r)) for s, r in zip(old, scales))
    if len(new) == 1:
        return TF.interpolate(x, size=new[0], mode='linear', align_corners=True)
    if len(new) == 2:
        return TF.interpolate(x, size=new, mode='bicubic', align_corners=True, antialias=True)
    dims = list(range(-len(new), 0))
    X = torch.fft.rfftn(x.float(), norm='forward', dim=dims)
    fsz = list(new[:-1]) + [new[-1] // 2 + 1]
    keep = [min(a, b) for a, b in zip(fsz, X.shape[-len(new):])]
    out = torch.zeros(x.shape[0], x.shape[1], *fsz, device=x.device, dtype=torch.cfloat)
    corners = [((None, m // 2), (-m // 2, None)) for m in keep[:-1]] + [((None, keep[-1]),)]
    for bounds in itertools.product(*corners):
        sl = (slice(None), slice(None)) + tuple(slice(*b) for b in bounds)
        out[sl] = X[sl]
    return torch.fft.irfftn(out, s=new, norm='forward', dim=dims)


class FNOBlocks(nn.Module):
    def __init__(self, in_channels, out_channels, n_modes, output_scaling_factor=None, n_layers=1,
                 incremental_n_modes=None, use_mlp=False, mlp_dropout=0, mlp_expansion=0.5,
                 non_linearity=TF.gelu, norm=None, ada_in_features=None, preactivation=False,
                 fno_skip='linear', mlp_skip='soft-gating', separable=False, factorization=None,
                 rank=1.0, SpectralConv=SpectralConv, joint_factorization=False,
                 fixed_rank_modes=False, implementation='factorized', decomposition_kwargs=dict(),
                 fft_norm='forward', **kwargs):
        super().__init__()
        if use_mlp:
            _unsupported("use_mlp=True")
        if norm is not None:
            _unsupported(f"norm={norm!r}")
        if preactivation:
            _unsupported("preactivation=True")
        if fno_skip != 'linear':
            _unsupported(f"fno_skip={fno_skip!r}")
        if non_linearity is not TF.gelu:
            _unsupported("non_linearity other than F.gelu")
        if isinstance(n_modes, int):
            n_modes = [n_modes]
        self.n_modes = list(n_modes)
        self.n_dim = len(self.n_modes)
        self.in_channels, self.out_channels, self.n_layers = in_channels, out_channels, n_layers
        self.non_linearity = non_linearity
        self.fft_norm = fft_norm
        if output_scaling_factor is not None:           # fno_block.py:37-42
            if isinstance(output_scaling_factor, (float, int)):
                output_scaling_factor = [[float(output_scaling_factor)] * self.n_dim] * n_layers
            elif isinstance(output_scaling_factor[0], (float, int)):
                output_scaling_factor = [[s] * self.n_dim for s in output_scaling_factor]
        self.output_scaling_factor = output_scaling_factor
        self.convs = SpectralConv(in_channels, out_channels, self.n_modes,
                                  output_scaling_factor=output_scaling_factor,
                                  incremental_n_modes=incremental_n_modes, rank=rank, fft_norm=fft_norm,
                                  fixed_rank_modes=fixed_rank_modes, implementation=implementation,
                                  separable=separable, factorization=factorization,
                                  decomposition_kwargs=decomposition_kwargs,
                                  joint_factorization=joint_factorization, n_layers=n_layers)
        Conv = getattr(nn, f'Conv{self.n_dim}d')
        self.fno_skips = nn.ModuleList([Conv(in_channels, out_channels, kernel_size=1, bias=False)
                                        for _ in range(n_layers)])

    def gelu_after(self, index):
        return index < (self.n_layers - index)          # fno_block.py:149

    def forward(self, x, index=0):
        """Unfused composition (one block on its own); FNO.forward uses the fused engine path."""
        x_skip = self.fno_skips[index](x)
        if self.convs.output_scaling_factor is not None:
            x_skip = _resample(x_skip, self.output_scaling_factor[index])
        x = self.convs(x, index) + x_skip
        if self.gelu_after(index):
            x = self.non_linearity(x)
        return x
