"""Recurrent neural operator with the reference surface (neuralop/models/rno.py): dialect-B
SpectralConv2d (:34-77), SpectralConvWithFC (:80-106), SpectralRegressor (:109-212),
FourierLayer2d (:215-228), RNO_cell (:231-260), RNO_layer (:263-290), RNO2d (:293-392).

Every spectral convolution runs in the HIP engine (functional.spectral_conv, norm 'ortho',
full modes1 x modes2 per corner); Fourier layers, cell gates, the input projection and the spectral
regressor (channel mixes + ReLU head) run on the engine's fused / pointwise kernels whenever the shape
allows (32 / 64 channels), torch ops otherwise.
Parameter names and shapes match the reference state_dict."""
import numpy as np
import torch
import torch.nn.functional as TF
from torch import nn

from ... import functional as F


class SpectralConv2d(nn.Module):
    def __init__(self, in_channels, out_channels, modes1, modes2, norm='ortho'):
        super().__init__()
        self.in_channels, self.out_channels = in_channels, out_channels
        self.modes1, self.modes2, self.norm = modes1, modes2, norm
        self.fourier_weight = nn.ParameterList(
            [nn.Parameter(torch.empty(in_channels, out_channels, modes1, modes2, 2)) for _ in range(2)])
        gain = np.sqrt(in_channels + out_channels) / (in_channels * out_channels)     # rno.py:42-48
        for w in self.fourier_weight:
            nn.init.xavier_normal_(w, gain=gain)

    def direct_grad_params(self):
        """parameters whose gradient the engine may write straight into a trainer.FlatGradBucket - honoured only while the
        enclosing model declares one use per step (functional.single_use; RNO2d.forward: one time step)"""
        return list(self.fourier_weight)

    def forward(self, x):
        if x.shape[-1] != x.shape[-2]:
            raise RuntimeError("rno.SpectralConv2d transforms with s=(n, n), n = x.shape[-1] (rno.py:66-67): "
                               "square grids only")
        return F.spectral_conv(x, list(self.fourier_weight), None, (self.modes1, self.modes2), self.norm,
                               direct_grads=getattr(self, "_direct_grads", False))


class SpectralConvWithFC(nn.Module):
    """channels-last: act(spec_conv(dropout(x)) + Linear(x))  (rno.py:92-106)."""

    def __init__(self, in_channels, out_channels, modes1, modes2, n_grid=None, dropout=0.1, norm='ortho',
                 activation='silu', return_freq=False, debug=False):
        super().__init__()
        if return_freq:
            raise RuntimeError("Not supported return freq")
        self.in_channels, self.out_channels = in_channels, out_channels
        self.spec_conv = SpectralConv2d(in_channels, out_channels, modes1, modes2, norm)
        self.linear = nn.Linear(in_channels, out_channels)
        self.activation = nn.SiLU() if activation == 'silu' else nn.ReLU()
        self.dropout = nn.Dropout(dropout)
        self.return_freq = False

    def forward(self, x):
        a = x.permute(0, 3, 1, 2)
        if F.pointwise_supported(a) and self.in_channels == self.out_channels:
            return self.forward_channels_first(a).permute(0, 2, 3, 1)
        res = self.linear(x)
        y = self.spec_conv(self.dropout(x).permute(0, 3, 1, 2).contiguous())
        return self.activation(y.permute(0, 2, 3, 1) + res)

    def forward_channels_first(self, a):
        """The same layer on a (B, C, X, Y) tensor, no layout changes: the Linear is a 1x1 channel mix, so
        `linear(a) + bias + spec_conv(dropout(a))` is ONE fno_pointwise_* launch with the spectral branch as its
        addend (the dropout mask is drawn on the channels-first tensor: same distribution, different element order
        than the reference's channels-last draw)."""
        sc = self.spec_conv
        relu, ident = isinstance(self.activation, nn.ReLU), isinstance(self.activation, nn.Identity)
        if ((relu or ident) and not getattr(self, "no_engine_tail", False) and a.shape[-1] == a.shape[-2]
                and F.block_tail_supported(a, (sc.modes1, sc.modes2), sc.norm)):
            # ONE fused engine layer incl. the dropout of the spectral branch (counter-based mask, regenerated in the
            # backward), the ReLU and its derivative (fno_model_forward_tail / _backward_tail); the mask comes from the
            # engine's hash of torch-drawn seed words: same distribution as nn.Dropout, a different stream
            p = self.dropout.p if self.training else 0.0
            seed = F.draw_dropout_seed(a.device) if p > 0 else None
            return F.fno_block_tail(a.contiguous(), self.linear.weight, list(sc.fourier_weight), self.linear.bias.view(1, -1),
                                    (sc.modes1, sc.modes2), sc.norm, relu_out=relu, drop_p=p, seed=seed,
                                    direct_grads=getattr(sc, "_direct_grads", False))
        s = self.spec_conv(self.dropout(a))
        return self.activation(F.pointwise_conv_add(a, self.linear.weight, self.linear.bias, addend=s))


class SpectralRegressor(nn.Module):
    def __init__(self, in_dim, n_hidden, freq_dim, out_dim, modes, num_spectral_layers=2, n_grid=None,
                 dim_feedforward=None, spacial_fc=False, spacial_dim=2, return_freq=False,
                 return_latent=False, normalizer=None, activation='silu', last_activation=True,
                 dropout=0.1, debug=False):
        super().__init__()
        if spacial_dim != 2:
            raise NotImplementedError("3D not implemented.")
        if return_freq or return_latent or spacial_fc or normalizer is not None:
            raise NotImplementedError("SpectralRegressor: only the configuration RNO2d uses is built")
        activation = 'silu' if activation is None else activation
        dropout = 0.1 if dropout is None else dropout
        self.activation = nn.SiLU() if activation == 'silu' else nn.ReLU()
        dims = [n_hidden] + [freq_dim] * num_spectral_layers
        self.spectral_conv = nn.ModuleList([
            SpectralConvWithFC(dims[i], dims[i + 1], modes, modes, n_grid=n_grid, dropout=dropout,
                               activation=activation) for i in range(num_spectral_layers)])
        if not last_activation:
            self.spectral_conv[-1].activation = nn.Identity()
        self.dim_feedforward = 2 * spacial_dim * freq_dim if dim_feedforward is None else dim_feedforward
        self.regressor = nn.Sequential(nn.Linear(freq_dim, self.dim_feedforward), self.activation,
                                       nn.Linear(self.dim_feedforward, out_dim))

    def forward(self, x, edge=None, pos=None, grid=None):
        a = x.permute(0, 3, 1, 2)            # RNO2d hands over a permuted channels-first tensor: this is a view of it
        fc1, fc2 = self.regressor[0], self.regressor[2]
        if (x.dim() == 4 and F.pointwise_supported(a)
                and all(l.in_channels == l.out_channels for l in self.spectral_conv)
                and isinstance(self.activation, nn.ReLU)
                and F.projection_supported(a, fc1.out_features, fc2.out_features, "relu")):
            # whole regressor channels-first on the engine: two pointwise launches + the fused ReLU head
            # (hidden (B, X, Y, 256) tensor never materialised; backward recomputes it)
            for layer in self.spectral_conv:
                a = layer.forward_channels_first(a)
            y = F.projection_head(a, fc1.weight, fc1.bias, fc2.weight, fc2.bias, act="relu")
            return y.permute(0, 2, 3, 1)
        for layer in self.spectral_conv:
            x = layer(x)
        return self.regressor(x)


class FourierLayer2d(nn.Module):
    def __init__(self, modes1, modes2, width):
        super().__init__()
        self.modes1, self.modes2, self.width = modes1, modes2, width
        self.spec_conv = SpectralConv2d(width, width, modes1, modes2, norm='ortho')
        self.norm_conv1d = nn.Conv1d(width, width, 1)

    def forward(self, x):
        b, c, n1, n2 = x.shape
        if n1 == n2 and F.blocks_supported(x, 1, (self.spec_conv.modes1, self.spec_conv.modes2), self.spec_conv.norm):
            # one fused engine layer: spectral conv + Conv1d(k=1) + bias (fno_model_* block stack, L = 1)
            sc = self.spec_conv
            return F.fno_blocks(x, [self.norm_conv1d.weight], list(sc.fourier_weight), self.norm_conv1d.bias.view(1, c),
                                (sc.modes1, sc.modes2), sc.norm, direct_grads=getattr(sc, "_direct_grads", False))
        return self.spec_conv(x) + self.norm_conv1d(x.reshape(b, c, n1 * n2)).view(b, self.width, n1, n2)


class RNO_cell(nn.Module):
    """GRU cell whose eight linear maps are Fourier layers (rno.py:254-260)."""

    def __init__(self, in_dim, out_dim, modes1, modes2, width):
        super().__init__()
        self.modes1, self.modes2, self.width, self.in_dim, self.out_dim = modes1, modes2, width, in_dim, out_dim
        for i in range(1, 9):
            setattr(self, f"f{i}", FourierLayer2d(modes1, modes2, width))
        for i in range(1, 5):
            setattr(self, f"b{i}", nn.Parameter(torch.normal(torch.tensor(0.), torch.tensor(1.))))

    def _fan(self, layers, t):
        """Fourier layers that share the input t as one engine fan-out (fno_fanout_*): t is transformed once, and the
        layers' input gradients are summed inside the backward kernels."""
        sc = layers[0].spec_conv
        return F.fourier_fanout(t, [l.norm_conv1d.weight for l in layers], [l.norm_conv1d.bias for l in layers],
                                [w for l in layers for w in l.spec_conv.fourier_weight], (sc.modes1, sc.modes2), sc.norm,
                                direct_grads=all(getattr(l.spec_conv, "_direct_grads", False) for l in layers))

    def forward(self, x, h):
        sc = self.f1.spec_conv
        if (F.gates_supported(x, h) and x.shape == h.shape and x.shape[-1] == x.shape[-2]
                and F.fanout_supported(x, 4, (sc.modes1, sc.modes2), sc.norm)):
            # three distinct inputs feed the eight Fourier layers (rno.py:254-260): x -> f1, f3, f5, f7; h -> f2, f4, f8;
            # r * h -> f6
            a1, a3, a5, a7 = self._fan((self.f1, self.f3, self.f5, self.f7), x)
            a2, a4, a8 = self._fan((self.f2, self.f4, self.f8), h)
            rh = F.rno_reset_gate(a3, a4, self.b2, h)
            return F.rno_output_gate(a1, a2, self.b1, a7, a8, self.b4, a5, self.f6(rh), self.b3, h)
        if F.gates_supported(x, h):
            # two fused engine kernels for everything between the eight Fourier layers (fno_rno_*_gate_*)
            rh = F.rno_reset_gate(self.f3(x), self.f4(h), self.b2, h)
            return F.rno_output_gate(self.f1(x), self.f2(h), self.b1, self.f7(x), self.f8(h), self.b4,
                                     self.f5(x), self.f6(rh), self.b3, h)
        z = torch.sigmoid(self.f1(x) + self.f2(h) + self.b1)
        z2 = torch.sigmoid(self.f7(x) + self.f8(h) + self.b4)
        r = torch.sigmoid(self.f3(x) + self.f4(h) + self.b2)
        h_hat = TF.selu(self.f5(x) + self.f6(r * h) + self.b3)
        return (1. - z) * h + z2 * h_hat


class RNO_layer(nn.Module):
    def __init__(self, in_dim, out_dim, modes1, modes2, width, return_sequences=False):
        super().__init__()
        self.modes1, self.modes2, self.width = modes1, modes2, width
        self.in_dim, self.out_dim, self.return_sequences = in_dim, out_dim, return_sequences
        self.cell = RNO_cell(in_dim, out_dim, modes1, modes2, width)
        self.bias_h = nn.Parameter(torch.normal(torch.tensor(0.), torch.tensor(1.)))

    def forward(self, x, h=None):
        b, steps, _, n1, n2 = x.shape
        if h is None:
            # zeros + bias_h (rno.py:279) in one pass: the broadcast scalar materialised once (same values, same gradient)
            h = self.bias_h.expand(b, self.width, n1, n2).contiguous()
        seq = []
        for t in range(steps):
            # one time step (the shipped configuration): a view, so the backward needs no zero-filled (B, T, ...) scatter
            xt = x.reshape(b, x.shape[2], n1, n2) if steps == 1 else x[:, t]
            h = self.cell(xt, h)
            if self.return_sequences:
                seq.append(h)
        if not self.return_sequences:
            return h
        return torch.stack(seq, dim=1) if steps > 1 else seq[0].unsqueeze(1)


# wider twins (32 or 64 channels) of narrow models (see RNO2d._wide_twin): keyed by the configuration, parameters on the meta device
_WIDE_TWINS = {}
ENGINE_WIDTHS = (32, 64)


def _pad_to(t, shape):
    """Zero-pad every dimension of t at its end up to `shape` (differentiable: the gradient is the leading slice)."""
    if tuple(t.shape) == tuple(shape):
        return t
    pads = []
    for have, want in zip(reversed(t.shape), reversed(shape)):
        pads += [0, want - have]
    return TF.pad(t, pads)


class RNO2d(nn.Module):
    def __init__(self, modes1, modes2, width, recurrent_index, layer_num=3, pad_amount=None, pad_dim='1'):
        super().__init__()
        pad_amount = tuple(pad_amount) if pad_amount is not None else None       # a list (YAML / JSON) is not hashable
        self._ctor = (modes1, modes2, recurrent_index, layer_num, pad_amount, pad_dim)
        self.modes1 = modes2           # the reference overwrites modes1 with modes2 (rno.py:301-302)
        self.width, self.pad_amount, self.pad_dim = width, pad_amount, pad_dim
        self.recurrent_index = recurrent_index
        self.in_dim, self.out_dim, self.layer_num = 1, 1, layer_num
        self.input_projection_layer = nn.Linear(self.in_dim, width)
        nn.init.normal_(self.input_projection_layer.weight, mean=0, std=1)
        self.layers = nn.ModuleList(
            [RNO_layer(width, width, modes1, modes2, width, return_sequences=True) for _ in range(layer_num - 1)]
            + [RNO_layer(width, width, modes1, modes2, width, return_sequences=False)])
        self.regressor = SpectralRegressor(in_dim=width, n_hidden=width, freq_dim=width, out_dim=self.out_dim,
                                           modes=modes2, activation='relu', dropout=0.3)

    def _pad(self, x):
        if not self.pad_amount:
            return x
        if self.pad_dim in ('1', 'both'):
            x = TF.pad(x.transpose(-1, -2), [0, self.pad_amount[0]]).transpose(-1, -2)
        if self.pad_dim in ('2', 'both'):
            x = TF.pad(x, [0, self.pad_amount[1]])
        return x

    def _unpad(self, h):
        if not self.pad_amount:
            return h
        if self.pad_dim in ('1', 'both'):
            h = h[:, :, :-self.pad_amount[0]]
        if self.pad_dim in ('2', 'both'):
            h = h[..., :-self.pad_amount[1]]
        return h

    def forward_one_step(self, x, v_plane=None, init_hidden_states=None):
        if init_hidden_states is None:
            init_hidden_states = [None] * self.layer_num
        b, t, n1, n2, _ = x.shape
        lin = self.input_projection_layer
        xc = x.reshape(b * t, 1, n1, n2)     # in_dim = 1: channels-last == channels-first
        if not x.requires_grad and F.lifting_supported(xc, self.width):
            x = F.lifting(xc, lin.weight, lin.bias).view(b, t, self.width, n1, n2)       # engine 1 -> C channel mix
        else:
            x = lin(x).permute(0, 1, 4, 2, 3)
        x = self._pad(x)                     # (B, T, C, X, Y)
        finals = []
        for i, layer in enumerate(self.layers):
            out = layer(x, init_hidden_states[i])
            if i < self.layer_num - 1:
                x = x + out                      # residual over the whole sequence (rno.py:343-345)
                finals.append(x[:, -1])
            else:
                x = out
                finals.append(x)
        h = self._unpad(finals[-1]).permute(0, 2, 3, 1)
        return self.regressor(h), finals

    def predict(self, x, num_steps):
        outs, states = [], [None] * self.layer_num
        for _ in range(num_steps):
            pred, states = self.forward_one_step(x, init_hidden_states=states)
            outs.append(pred)
            x = pred.reshape(pred.shape[0], 1, pred.shape[1], pred.shape[2], pred.shape[3])
        return torch.stack(outs, dim=1)

    def _wide_twin(self):
        """A width the fused kernels do not tile (the shipped YAML's 34, configs/matlab_rno.yaml:80) runs EXACTLY as the same
        network embedded in the next width the kernels tile (32 or 64 channels): every parameter zero-padded to the twin's shapes.  Padded input columns
        are zero, so the padded channels (which the scalar gate biases make non-zero) never reach a real channel, and the
        gradient of a padded weight entry is discarded by the pad's adjoint (a slice); the regressor head widens its 4 * width hidden units the
        same way.  The twin is a structure only (meta-device parameters, shared per configuration)."""
        wide = min(w for w in ENGINE_WIDTHS if w >= self.width)      # the narrowest width the fused kernels tile
        key = self._ctor + (wide,)
        if key not in _WIDE_TWINS:
            m1, m2, ri, ln, pa, pd = self._ctor
            with torch.device("meta"):
                _WIDE_TWINS[key] = RNO2d(m1, m2, wide, ri, layer_num=ln, pad_amount=pa, pad_dim=pd)
        return _WIDE_TWINS[key]

    def forward(self, x, v_plane=None, timestep=2):
        if (x.is_cuda and self.width not in ENGINE_WIDTHS and self.width < 64
                and not getattr(self, "no_width_padding", False)):
            twin = self._wide_twin()
            twin.train(self.training)          # dropout of the regressor follows THIS module's mode
            shapes = {k: v.shape for k, v in twin.named_parameters()}
            padded = {k: _pad_to(v, shapes[k]) for k, v in self.named_parameters()}
            return torch.func.functional_call(twin, padded, (x,), {"v_plane": v_plane, "timestep": timestep})
        # every parameter is used once per time step and per predicted step (both = x.shape[1]): gradients may be written in
        # place only for a single step
        with F.single_use(x.shape[1] == 1):
            return self.predict(x, num_steps=x.shape[1])[:, self.recurrent_index]

    def count_params(self):
        return int(sum(p.numel() for p in self.parameters() if p.requires_grad))
