"""CPU oracle, part 2: RNO2d and the PINO observers (TEST INFRASTRUCTURE ONLY, see
fno_oracle.py).  Functional restatements over a {reference state_dict name: tensor} dict;
pinned by tests/golden/{rno2d_*, pino_*}.npz generated from the real reference."""
import torch
import torch.nn.functional as F

from .fno_oracle import spectral_conv_B, spectral_conv_C3d


def _sub(p, prefix):
    return {k[len(prefix):]: v for k, v in p.items() if k.startswith(prefix)}


# ReLU decisions of the RNO regressor (rno.py:104, 170-174), exposed for the mask-conditioned gradient comparison of
# tests/test_fullsize_gpu.py: two float evaluations of the same network that decide a ReLU input within rounding of zero
# differently differentiate DIFFERENT piecewise-linear functions, and everything upstream of the decision moves by ~1e-5 of its
# norm.  RELU_HOOK(tag, pre_activation) -> activated tensor; None = F.relu.
RELU_HOOK = None


class ReluMasks(object):
    """RELU_HOOK that records every decision (`seen[tag]`: list of bool tensors in call order) and, for tags in `impose`
    ({tag: bool tensor over the whole batch, channels last}), applies THAT decision instead of its own: consecutive calls
    with one tag take consecutive sample ranges (the full-size tests evaluate the oracle in chunks of samples)."""

    def __init__(self, impose=None):
        self.impose = impose or {}
        self.seen, self._next = {}, {}

    def __call__(self, tag, t):
        own = t > 0
        self.seen.setdefault(tag, []).append(own)
        if tag in self.impose:
            lo = self._next.get(tag, 0)
            self._next[tag] = lo + t.shape[0]
            return t * self.impose[tag][lo:lo + t.shape[0]].to(t.dtype)
        return F.relu(t)


def _relu(tag, t):
    return F.relu(t) if RELU_HOOK is None else RELU_HOOK(tag, t)


# ---- neuralop/models/rno.py ---------------------------------------------------------
def fourier_layer2d(p, x, m1, m2):
    """rno.py:224-228: SpectralConv2d(x) + Conv1d(k=1)(x)."""
    b, c, n1, n2 = x.shape
    spec = spectral_conv_B(x, p["spec_conv.fourier_weight.0"], p["spec_conv.fourier_weight.1"], m1, m2)
    w = p["norm_conv1d.weight"][:, :, 0]
    lin = torch.einsum("oi,bin->bon", w, x.reshape(b, c, n1 * n2)) + p["norm_conv1d.bias"][None, :, None]
    return spec + lin.view(b, -1, n1, n2)


def rno_cell(p, x, h, m1, m2):
    """rno.py:254-260."""
    f = lambda i, t: fourier_layer2d(_sub(p, f"f{i}."), t, m1, m2)
    z = torch.sigmoid(f(1, x) + f(2, h) + p["b1"])
    z2 = torch.sigmoid(f(7, x) + f(8, h) + p["b4"])
    r = torch.sigmoid(f(3, x) + f(4, h) + p["b2"])
    h_hat = F.selu(f(5, x) + f(6, r * h) + p["b3"])
    return (1.0 - z) * h + z2 * h_hat


def rno_layer(p, x, h, m1, m2, width, return_sequences):
    """rno.py:275-290."""
    b, steps, _, n1, n2 = x.shape
    if h is None:
        h = torch.zeros(b, width, n1, n2, dtype=x.dtype) + p["bias_h"]
    seq = []
    for t in range(steps):
        h = rno_cell(_sub(p, "cell."), x[:, t], h, m1, m2)
        seq.append(h)
    return torch.stack(seq, dim=1) if return_sequences else h


def spectral_conv_with_fc(p, x, m, tag="relu"):
    """rno.py:92-106 in eval mode (dropout = identity), activation ReLU as RNO2d builds it."""
    res = x @ p["linear.weight"].t() + p["linear.bias"]
    y = spectral_conv_B(x.permute(0, 3, 1, 2), p["spec_conv.fourier_weight.0"], p["spec_conv.fourier_weight.1"], m, m)
    return _relu(tag, y.permute(0, 2, 3, 1) + res)


def rno2d_forward(p, x, modes1, modes2, width, recurrent_index, layer_num):
    """RNO2d.forward -> predict -> forward_one_step (rno.py:320-379), no padding, eval mode."""
    def one_step(xs, states):
        h = (xs @ p["input_projection_layer.weight"].t() + p["input_projection_layer.bias"]).permute(0, 1, 4, 2, 3)
        finals = []
        for i in range(layer_num):
            out = rno_layer(_sub(p, f"layers.{i}."), h, states[i], modes1, modes2, width, i < layer_num - 1)
            if i < layer_num - 1:
                h = h + out
                finals.append(h[:, -1])
            else:
                h = out
                finals.append(h)
        t = finals[-1].permute(0, 2, 3, 1)
        for j in range(2):
            t = spectral_conv_with_fc(_sub(p, f"regressor.spectral_conv.{j}."), t, modes2, f"regressor.spectral_conv.{j}")
        t = _relu("regressor.head", t @ p["regressor.regressor.0.weight"].t() + p["regressor.regressor.0.bias"])
        return t @ p["regressor.regressor.2.weight"].t() + p["regressor.regressor.2.bias"], finals

    outs, states = [], [None] * layer_num
    for _ in range(x.shape[1]):
        pred, states = one_step(x, states)
        outs.append(pred)
        x = pred.reshape(pred.shape[0], 1, pred.shape[1], pred.shape[2], pred.shape[3])
    return torch.stack(outs, dim=1)[:, recurrent_index]


# ---- libs/models/pino_models/pinobserver.py -----------------------------------------------
def multiplicative_net(p, x1, x2):
    """pinobserver.py:41-59: B x1 + A x2 + bias (x2 a per-sample code broadcast over the grid)."""
    if x2.dim() < 2:
        x2 = x2.unsqueeze(-1)
    return x1 @ p["B"].t() + (x2 @ p["A"].t())[:, None, None, None, :] + p["bias"]


def _spectral_stack(p, x, layers, modes):
    """4 x [SpectralConv3d + Conv1d(k=1)], GELU except after the last (pinobserver.py:221-226, 261-266)."""
    b = x.shape[0]
    n = len(layers) - 1
    for i in range(n):
        sx, sy, sz = x.shape[-3:]
        ws = [p[f"sp_convs.{i}.weights{j}"] for j in (1, 2, 3, 4)]
        x1 = spectral_conv_C3d(x, *ws, *modes[i])
        w = p[f"ws.{i}.weight"][:, :, 0]
        x2 = torch.einsum("oi,bin->bon", w, x.reshape(b, layers[i], -1)) + p[f"ws.{i}.bias"][None, :, None]
        x = x1 + x2.view(b, layers[i + 1], sx, sy, sz)
        if i != n - 1:
            x = F.gelu(x)
    return x


def _pads(size_z, pad_ratio):
    return [round(size_z * r) for r in pad_ratio] if max(pad_ratio) > 0 else [0, 0]


def pinobserver2d_forward(p, x, re, layers, modes, pad_ratio):
    """pinobserver.py:192-233."""
    re = re if re.dtype == torch.float64 else re.float()      # (float64 passes through: error-budget runs)
    num_pad = _pads(x.shape[-2], pad_ratio)
    x = x @ p["fc0.weight"].t() + p["fc0.bias"]
    x = multiplicative_net(_sub(p, "multiplicative_net1."), x, re).permute(0, 4, 1, 2, 3)
    if max(num_pad) > 0:
        x = F.pad(x, (num_pad[0], num_pad[1]))
    x = _spectral_stack(p, x, layers, modes)
    if max(num_pad) > 0:
        x = x[..., num_pad[0]:-num_pad[1]]
    x = multiplicative_net(_sub(p, "multiplicative_net2."), x.permute(0, 2, 3, 4, 1), re)
    x = F.gelu(x @ p["fc1.weight"].t() + p["fc1.bias"])
    return x @ p["fc2.weight"].t() + p["fc2.bias"]


def pinobserver_fullfield_forward(p, x, re, layers, modes, pad_ratio, max_re=1000):
    """pinobserver.py:341-375 + PlanePredHead.forward :257-273."""
    re = (re if re.dtype == torch.float64 else re.float()) / max_re
    num_pad = _pads(x.shape[-2], pad_ratio)
    x = x @ p["fc0.weight"].t() + p["fc0.bias"]
    x = multiplicative_net(_sub(p, "multiplicative_net1."), x, re).permute(0, 4, 1, 2, 3)
    if max(num_pad) > 0:
        x = F.pad(x, (num_pad[0], num_pad[1]))
    hp = _sub(p, "observer_head.")
    x = _spectral_stack(hp, x, layers, modes)
    if max(num_pad) > 0:
        x = x[..., num_pad[0]:-num_pad[1]]
    x = multiplicative_net(_sub(p, "multiplicative_net2."), x.permute(0, 2, 3, 4, 1), re)
    x = F.gelu(x @ hp["fc1.weight"].t() + hp["fc1.bias"])
    x = x @ hp["fc2.weight"].t() + hp["fc2.bias"]
    return x.permute(0, 4, 1, 2, 3)
