"""Three-dimensional spectral regressor with the reference surface (neuralop/models/spectral_regressor.py:
SpectralConv3d :17-61, SpectralConvWithFC3d :64-90, SpectralRegressor :93-201) - the class `neuralop.models` exports
under the name `SpectralRegressor` (models/__init__.py:6).  The spectral convolutions run in the HIP engine (dialect C:
four corner weights, last-dim spectrum cut or zero-padded to modes3); the Linear beside each of them is the engine's
pointwise channel mix with the spectral branch as its addend whenever the width tiles (32 / 64 channels), torch otherwise.
Only spacial_dim = 3 can be built: the reference's own 2-D branch names a class its module never defines (:117-118)."""
import torch
from torch import nn

from ... import functional as F
from ...libs.models.pino_models.basics import SpectralConv3d  # noqa: F401  (same operator as the reference's :17-61)


def default(value, d):
    return d if value is None else value


class SpectralConvWithFC3d(nn.Module):
    def __init__(self, in_channels, out_channels, modes1, modes2, modes3, n_grid=None, dropout=0.1, norm='ortho',
                 activation='silu', return_freq=False, debug=False):
        super().__init__()
        self.in_channels, self.out_channels = in_channels, out_channels
        self.spec_conv = SpectralConv3d(in_channels, out_channels, modes1, modes2, modes3)
        self.linear = nn.Linear(in_channels, out_channels)
        self.activation = nn.SiLU() if activation == 'silu' else nn.ReLU()
        self.dropout = nn.Dropout(dropout)
        self.return_freq = return_freq

    def forward(self, x):
        """(B, X, Y, Z, in) -> (B, X, Y, Z, out): activation(spec_conv(dropout(x)) + linear(x))"""
        if self.return_freq:
            raise RuntimeError("Not supported return freq")
        a = x.permute(0, 4, 1, 2, 3).contiguous()
        s = self.spec_conv(self.dropout(a))
        if self.in_channels == self.out_channels and F.pointwise_supported(a):
            y = F.pointwise_conv_add(a, self.linear.weight, self.linear.bias, addend=s)
            return self.activation(y).permute(0, 2, 3, 4, 1)
        return self.activation(s.permute(0, 2, 3, 4, 1) + self.linear(x))


class SpectralRegressor(nn.Module):
    def __init__(self, in_dim, n_hidden, freq_dim, out_dim, modes: int, num_spectral_layers: int = 2, n_grid=None,
                 dim_feedforward=None, spacial_fc=False, spacial_dim=2, return_freq=False, return_latent=False,
                 normalizer=None, activation='silu', last_activation=True, dropout=0.1, debug=False):
        super().__init__()
        if spacial_dim != 3:
            raise NotImplementedError("neuralop.models.SpectralRegressor: only spacial_dim=3 exists "
                                      "(the 2-D regressor RNO2d uses is neuralop.models.rno.SpectralRegressor)")
        if return_freq:
            raise RuntimeError("Not supported return freq")
        activation = default(activation, 'silu')
        dropout = default(dropout, 0.1)
        self.activation = nn.SiLU() if activation == 'silu' else nn.ReLU()
        self.spacial_fc = spacial_fc
        if spacial_fc:
            self.fc = nn.Linear(in_dim + spacial_dim, n_hidden)
        widths = [n_hidden] + [freq_dim] * num_spectral_layers
        self.spectral_conv = nn.ModuleList([
            SpectralConvWithFC3d(widths[i], widths[i + 1], modes, modes, modes, n_grid=n_grid, dropout=dropout,
                                 activation=activation) for i in range(num_spectral_layers)])
        if not last_activation:
            self.spectral_conv[-1].activation = nn.Identity()
        self.n_grid = n_grid
        self.dim_feedforward = default(dim_feedforward, 2 * spacial_dim * freq_dim)
        self.regressor = nn.Sequential(nn.Linear(freq_dim, self.dim_feedforward), self.activation,
                                       nn.Linear(self.dim_feedforward, out_dim))
        self.normalizer = normalizer
        self.return_freq, self.return_latent, self.debug = return_freq, return_latent, debug

    def forward(self, x, edge=None, pos=None, grid=None):
        latent = []
        if self.spacial_fc:
            x = self.fc(torch.cat([x, grid], dim=-1))
        for layer in self.spectral_conv:
            x = layer(x)
            if self.return_latent:
                latent.append(x.contiguous())
        x = self.regressor(x)
        if self.normalizer:
            x = self.normalizer.inverse_transform(x)
        if self.return_latent:
            return x, dict(preds_freq=[], preds_latent=latent)
        return x
