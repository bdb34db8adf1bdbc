"""PINO observers with the reference surface (libs/models/pino_models/pinobserver.py:
MultiplicativeNet :14-63, PINObserver2d :129-233, PlanePredHead :236-273,
PINObserverFullField :276-375).  SpectralConv3d runs in the HIP engine; the channels-last
Linear / Conv1d(k=1) glue is torch."""
import math

import torch
import torch.nn.functional as TF
from torch import nn
from torch.nn import init

from .basics import SpectralConv3d
from .... import functional as F


def _activation(name):
    table = {"tanh": torch.tanh, "gelu": TF.gelu, "relu": TF.relu_, "elu": TF.elu_, "leaky_relu": TF.leaky_relu_}
    if name not in table:
        raise ValueError(f"{name} is not supported")
    return table[name]


def _pad_last(x, num_pad):
    return TF.pad(x, (num_pad[0], num_pad[1]), "constant", 0) if max(num_pad) > 0 else x


def _unpad_last(x, num_pad):
    return x[..., num_pad[0]:-num_pad[1]] if max(num_pad) > 0 else x


class MultiplicativeNet(nn.Module):
    """out = B x1 + A x2 + bias, x1 (N, X, Y, T, in1) channels-last, x2 (N, in2) a per-sample code."""

    def __init__(self, in1_features, in2_features, out_features, device=None, dtype=None):
        super().__init__()
        self.in1_features, self.in2_features, self.out_features = in1_features, in2_features, out_features
        self.A = nn.Parameter(torch.empty(out_features, in2_features))
        self.B = nn.Parameter(torch.empty(out_features, in1_features))
        self.bias = nn.Parameter(torch.empty(out_features))
        self.reset_parameters()

    def reset_parameters(self):
        init.kaiming_uniform_(self.A, a=math.sqrt(5))
        init.kaiming_uniform_(self.B, a=math.sqrt(5))
        bound = 1 / math.sqrt(self.in1_features)
        init.uniform_(self.bias, -bound, bound)

    def forward(self, input1, input2):
        if input2.dim() < 2:
            input2 = input2.unsqueeze(-1)
        code = (input2 @ self.A.t())[:, None, None, None, :]
        return input1 @ self.B.t() + code + self.bias


def _lift_front(fc0, mn, x, re, width):
    """multiplicative_net1(fc0(x), re) as a channels-first tensor (pinobserver.py:205-207, 356-358).  fc0 followed by the
    Re-conditioning affine is ONE linear map of the <= 4 input channels: when the shape allows it the two small matrices are
    composed (autograd differentiates the composition) and applied by the engine's lifting kernels."""
    xc = x.permute(0, 4, 1, 2, 3)
    if not x.requires_grad and F.lifting_supported(xc, width):
        w = mn.B @ fc0.weight                                             # (C, in_dim)
        bias = mn.B @ fc0.bias + mn.bias
        code = (re if re.dim() >= 2 else re.unsqueeze(-1)) @ mn.A.t()     # (B, C)
        return F.lifting(xc.contiguous(), w, bias) + code[:, :, None, None, None]
    return mn(fc0(x), re).permute(0, 4, 1, 2, 3)


class _SpectralStack(nn.Module):
    """layers of  x <- act(SpectralConv3d(x) + Conv1d_{k=1}(x)), no activation after the last."""

    def _build_stack(self, layers, modes1, modes2, modes3):
        self.sp_convs = nn.ModuleList([SpectralConv3d(i, o, m1, m2, m3)
                                       for i, o, m1, m2, m3 in zip(layers, layers[1:], modes1, modes2, modes3)])
        self.ws = nn.ModuleList([nn.Conv1d(i, o, 1) for i, o in zip(layers, layers[1:])])

    def _chain_supported(self, x):
        if self.act is not TF.gelu or len(set(self.layers)) != 1:
            return False
        for i, conv in enumerate(self.sp_convs):
            ws, modes, wle = conv.engine_call(x)
            if not F.spectral_layer_supported(x, len(ws), modes, "backward", wle, i > 0):
                return False
        return True

    def _fused_stack(self, x):
        """The whole stack as ONE engine block stack (fno_model_* with Cin = Cout = 0: per layer one fused kernel each way that
        applies the previous GELU on load, mixes the channels, adds the inverse last-dim transform of the spectral branch
        as a K-extension of the same GEMM and the bias; no `SpectralConv(x)` tensor, no addend / gradient-addend passes)
        when every layer has the same width and kept modes and the weights store exactly the live last-dim modes."""
        if self.act is not TF.gelu or len(set(self.layers)) != 1:
            return None
        calls = [conv.engine_call(x) for conv in self.sp_convs]
        modes = calls[0][1]
        if any(c[1] != modes or c[2] != modes[-1] for c in calls):
            return None
        n = len(self.ws)
        gelu_mask = (1 << (n - 1)) - 1                       # GELU after every layer but the last
        if not F.blocks_supported(x, n, modes, "backward", gelu_mask):
            return None
        bias = torch.stack([w.bias for w in self.ws])
        direct = all(getattr(conv, "_direct_grads", False) for conv in self.sp_convs)
        return F.fno_blocks(x, [w.weight for w in self.ws], [t for c in calls for t in c[0]], bias, modes, "backward",
                            gelu_mask=gelu_mask, direct_grads=direct)

    def _run_stack(self, x):
        y = self._fused_stack(x)
        if y is not None:
            return y
        if self._chain_supported(x):
            # the stack chained on PRE-activation tensors: each layer applies the previous layer's GELU while it loads its
            # input (spectral rows and channel mix alike) and its backward folds gelu' and the two-branch gradient sum in
            for i, (conv, w) in enumerate(zip(self.sp_convs, self.ws)):
                ws, modes, wle = conv.engine_call(x)
                x = F.spectral_pointwise_layer(x, ws, modes, "backward", w.weight, w.bias, input_gelu=i > 0,
                                               weight_last_extent=wle, direct_grads=getattr(conv, "_direct_grads", False))
            return x
        b = x.shape[0]
        sx, sy, sz = x.shape[-3:]
        last = len(self.ws) - 1
        for i, (conv, w) in enumerate(zip(self.sp_convs, self.ws)):
            if self.layers[i] == self.layers[i + 1] and F.pointwise_supported(x):
                # Conv1d(k=1) + bias + the residual add in one engine kernel each way (fno_pointwise_*)
                x = F.pointwise_conv_add(x, w.weight, w.bias, conv(x))
            else:
                x = conv(x) + w(x.reshape(b, self.layers[i], -1)).view(b, self.layers[i + 1], sx, sy, sz)
            if i != last:
                x = self.act(x)
        return x


def _pad_ratio(pad_ratio):
    if isinstance(pad_ratio, float):
        return [pad_ratio, pad_ratio]
    assert len(pad_ratio) == 2, 'Cannot add padding in more than 2 directions.'
    return pad_ratio


class PINObserver2d(_SpectralStack):
    def __init__(self, modes1, modes2, modes3, width=16, fc_dim=128, layers=None, in_dim=4, out_dim=1,
                 act='gelu', pad_ratio=[0., 0.], use_fourier_layer=False):
        super().__init__()
        if use_fourier_layer:
            raise NotImplementedError("use_fourier_layer=True is outside the accelerated hot path")
        self.pad_ratio = _pad_ratio(pad_ratio)
        self.modes1, self.modes2, self.modes3, self.in_dim = modes1, modes2, modes3, in_dim
        self.layers = [width] * 4 if layers is None else layers
        self.use_fourier_layer, self.fourier_layer1 = False, None
        self.fc0 = nn.Linear(in_dim, self.layers[0])
        self.multiplicative_net1 = MultiplicativeNet(self.layers[0], 1, self.layers[0])
        self._build_stack(self.layers, modes1, modes2, modes3)
        self.multiplicative_net2 = MultiplicativeNet(self.layers[-1], 1, self.layers[-1])
        self.fc1 = nn.Linear(self.layers[-1], fc_dim)
        self.fc2 = nn.Linear(fc_dim, out_dim)
        self.act = _activation(act)

    def forward(self, x, re):
        re = re.float()
        size_z = x.shape[-2]
        num_pad = [round(size_z * r) for r in self.pad_ratio] if max(self.pad_ratio) > 0 else [0., 0.]
        y = self._forward_padded(x, re, [int(p) for p in num_pad])
        if y is not None:
            return y
        x = _lift_front(self.fc0, self.multiplicative_net1, x, re, self.layers[0])
        x = _unpad_last(self._run_stack(_pad_last(x, num_pad).contiguous()), num_pad)
        if (self.act is TF.gelu and self.layers[-1] in (32, 64)
                and F.projection_supported(x, self.fc1.out_features, self.fc2.out_features)):
            # channels-first tail on the engine: the Re-conditioning affine as a pointwise mix, then the projection kernels
            mn = self.multiplicative_net2
            code = (re if re.dim() >= 2 else re.unsqueeze(-1)) @ mn.A.t()                     # (B, C)
            h = F.pointwise_conv_add(x.contiguous(), mn.B, mn.bias, None) + code[:, :, None, None, None]
            y = F.projection_head(h, self.fc1.weight, self.fc1.bias, self.fc2.weight, self.fc2.bias)
            return y.permute(0, 2, 3, 4, 1)
        x = self.multiplicative_net2(x.permute(0, 2, 3, 4, 1), re)
        return self.fc2(self.act(self.fc1(x)))


    PER_SAMPLE_MAX = 16      # the per-sample-bias kernels are launched once per sample

    def _forward_padded(self, x, re, num_pad):
        """The whole forward on the PADDED grid, channels-first, with no layout or padding copies of the 64-channel tensors:
        the in_dim-channel INPUT is padded (C / in_dim times cheaper), lifted per sample with the Re-conditioning code folded
        into the bias, and the pad columns are zeroed in place (= F.pad of the lifted tensor, pinobserver.py:208-213); the
        tail (second Re-conditioning affine with its code as a per-sample bias, fc1 -> GELU -> fc2) runs on the padded tensor
        too - it is pointwise, the pad columns are dropped from the 1-channel output (:228).  None when the engine's
        lifting / pointwise / projection kernels do not cover the shape."""
        B = x.shape[0]
        if (x.requires_grad or self.act is not TF.gelu or len(set(self.layers)) != 1 or self.layers[0] not in (32, 64)
                or B > self.PER_SAMPLE_MAX or self.fc2.out_features != 1):
            return None
        p0, p1 = num_pad
        xc = TF.pad(x.permute(0, 4, 1, 2, 3), (p0, p1)) if p0 + p1 > 0 else x.permute(0, 4, 1, 2, 3).contiguous()
        C = self.layers[0]
        like = xc[:, :1].expand(-1, C, -1, -1, -1)          # a (B, C, X, Y, T') view for the shape predicates (no memory)
        if not (F.lifting_supported(xc, C) and F.pointwise_supported(like)
                and F.projection_supported(like, self.fc1.out_features, self.fc2.out_features)):
            return None
        fc0, mn1, mn2 = self.fc0, self.multiplicative_net1, self.multiplicative_net2
        re2 = re if re.dim() >= 2 else re.unsqueeze(-1)
        w = mn1.B @ fc0.weight                                             # (C, in_dim): fc0 then the Re-conditioning mix
        bias = (mn1.B @ fc0.bias + mn1.bias)[None, :] + re2 @ mn1.A.t()    # (B, C): + the per-sample code
        h = F.zero_last_pads_(F.lifting_per_sample_bias(xc, w, bias), p0, p1)
        h = self._run_stack(h)
        h = F.pointwise_conv_per_sample_bias(h, mn2.B, mn2.bias[None, :] + re2 @ mn2.A.t())
        y = F.projection_head(h, self.fc1.weight, self.fc1.bias, self.fc2.weight, self.fc2.bias)
        if p0 + p1 > 0:
            y = y[..., p0:y.shape[-1] - p1]
        return y.permute(0, 2, 3, 4, 1)


class PlanePredHead(_SpectralStack):
    def __init__(self, layers, modes1, modes2, modes3, fc_dim, out_dim, act):
        super().__init__()
        self.layers, self.modes1, self.modes2, self.modes3 = layers, modes1, modes2, modes3
        self._build_stack(layers, modes1, modes2, modes3)
        self.fc1 = nn.Linear(layers[-1], fc_dim)
        self.fc2 = nn.Linear(fc_dim, out_dim)
        self.act = _activation(act)

    def forward(self, x, num_pad, re, multiplicative_net2):
        x = _unpad_last(self._run_stack(x), num_pad)
        if (self.act is TF.gelu and self.layers[-1] in (32, 64) and not getattr(self, "no_engine_tail", False)
                and F.projection_supported(x, self.fc1.out_features, self.fc2.out_features)):
            # channels-first tail on the engine (as PINObserver2d's): the Re-conditioning affine as a pointwise mix with the
            # per-sample code added, then fc1 -> GELU -> fc2 for all planes in the projection kernels (hidden tensor never
            # materialised; pinobserver.py:257-273)
            mn = multiplicative_net2
            code = (re if re.dim() >= 2 else re.unsqueeze(-1)) @ mn.A.t()                     # (B, C)
            h = F.pointwise_conv_add(x.contiguous(), mn.B, mn.bias, None) + code[:, :, None, None, None]
            y = F.projection_head(h, self.fc1.weight, self.fc1.bias, self.fc2.weight, self.fc2.bias)
            return y.permute(0, 2, 3, 4, 1)
        x = multiplicative_net2(x.permute(0, 2, 3, 4, 1), re)
        return self.fc2(self.act(self.fc1(x)))


class PINObserverFullField(nn.Module):
    def __init__(self, plane_num, modes1, modes2, modes3, width=16, fc_dim=128, layers=None, in_dim=4, out_dim=1,
                 act='gelu', pad_ratio=[0., 0.], use_fourier_layer=False):
        super().__init__()
        if use_fourier_layer:
            raise NotImplementedError("use_fourier_layer=True is outside the accelerated hot path")
        self.plane_num, self.pad_ratio = plane_num, _pad_ratio(pad_ratio)
        self.modes1, self.modes2, self.modes3 = modes1, modes2, modes3
        self.max_re, self.in_dim = 1000, in_dim
        self.layers = [width] * 4 if layers is None else layers
        self.use_fourier_layer, self.fourier_layer1 = False, None
        self.fc0 = nn.Linear(in_dim, self.layers[0])
        self.multiplicative_net1 = MultiplicativeNet(self.layers[0], 1, self.layers[0])
        self.multiplicative_net2 = MultiplicativeNet(self.layers[-1], 1, self.layers[-1])
        self.observer_head = PlanePredHead(self.layers, modes1, modes2, modes3, fc_dim, out_dim * plane_num, act)

    def forward(self, x, re):
        re = re.float() / self.max_re
        size_z = x.shape[-2]                  # the reference takes the pad size from dim -2 (:353)
        num_pad = [round(size_z * r) for r in self.pad_ratio] if max(self.pad_ratio) > 0 else [0., 0.]
        x = _lift_front(self.fc0, self.multiplicative_net1, x, re, self.layers[0])
        pred = self.observer_head(_pad_last(x, num_pad).contiguous(), num_pad, re, self.multiplicative_net2)
        return pred.permute(0, 4, 1, 2, 3)     # (B, planes, X, Y, T)
