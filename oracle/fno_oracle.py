"""CPU oracle: a plain restatement of the reference's FNO hot path.

TEST INFRASTRUCTURE ONLY.  Only tests/, __graft_entry__.smoke() and the
`cpu_baseline` leg of bench.py may import this module.  The product path
(pde_policylearning_amd/) never imports it and has no CPU fallback.

Every function restates, in our own words, what one piece of
neuraloperator/pde-policylearning computes, and cites the file:line it follows
(paths relative to the reference checkout).  It is written with torch CPU ops
(torch.fft + einsum) so that (a) autograd gives the oracle gradients and (b) the
same code is what bench.py times as the CPU baseline ("port").

PINNING: oracle/make_golden.py imports the real reference in the build
container and dumps input/output/gradient vectors into tests/golden/*.npz;
tests/test_oracle_golden.py checks this module against every one of them.
Dialects B (neuralop/models/rno.py) and C (libs/models/pino_models/basics.py)
are pure-torch in the reference and are pinned directly.  Dialect A
(neuralop/models/spectral_convolution.py) routes its dense contraction through
tensorly/tltorch, third-party packages that are NOT vendored under the
reference, are unpinned in requirements.txt:1-8 and are absent from the image;
the generator substitutes a dense-only weight container and torch.einsum for
`tl.einsum` (for factorization=None the call is a plain 2-operand einsum,
spectral_convolution.py:31-36).  Forward/backward numerics for explicit weights
are therefore pinned; tltorch's init distribution and state_dict leaf names are
"parity unpinned".
"""
import itertools
import math

import torch
import torch.nn.functional as F


# --------------------------------------------------------------------------
# dialect A: neuralop.models.spectral_convolution.FactorizedSpectralConv (dense)
# --------------------------------------------------------------------------
def spectral_conv_A(x, weights, bias, half_modes, fft_norm="forward"):
    """y = irfftn(pad(W_c . rfftn(x)[corner_c])) + bias.

    Follows neuralop/models/spectral_convolution.py:303-347.
      x        (B, Cin, d1..dN) real
      weights  list of 2^(N-1) complex tensors (Cin, Cout, m1..mN), corner order =
               itertools.product over the leading dims of (low, high), last dim always
               [:m_last]  (:330-337)
      bias     (Cout, 1, .., 1) or None, added after the inverse FFT (:344-345)
      half_modes  kept extent per dim = n_modes // 2 on EVERY dim (:202-203)
    """
    order = len(half_modes)
    B = x.shape[0]
    sizes = list(x.shape[2:])
    fft_size = list(sizes)
    fft_size[-1] = fft_size[-1] // 2 + 1                      # :320-321
    dims = list(range(-order, 0))
    xf = torch.fft.rfftn(x.float() if x.dtype != torch.float64 else x,
                         norm=fft_norm, dim=dims)             # :324 (x.float())
    cout = weights[0].shape[1]
    out = torch.zeros([B, cout, *fft_size], dtype=xf.dtype, device=x.device)   # :326
    sel = [((None, m), (-m, None)) for m in half_modes[:-1]] + [((None, half_modes[-1]),)]  # :330
    for i, bnd in enumerate(itertools.product(*sel)):
        idx = (slice(None), slice(None)) + tuple(slice(*b) for b in bnd)
        # dense contraction 'b i x y.., i o x y.. -> b o x y..'   (:15-36)
        out[idx] = _mode_einsum(xf[idx], weights[i])
    y = torch.fft.irfftn(out, s=sizes, norm=fft_norm)          # :342 (trailing dims)
    if bias is not None:
        y = y + bias                                          # :344-345
    return y


def _mode_einsum(a, w):
    nd = a.dim() - 2
    sp = "xyzt"[:nd]
    return torch.einsum(f"bi{sp},io{sp}->bo{sp}", a, w)


# --------------------------------------------------------------------------
# dialect B: neuralop.models.rno.SpectralConv2d
# --------------------------------------------------------------------------
def spectral_conv_B(x, w0, w1, modes1, modes2, norm="ortho"):
    """Follows neuralop/models/rno.py:60-77.  w0, w1 real (Cin, Cout, m1, m2, 2);
    FFT size is (n, n) with n = x.shape[-1] (square grids only, :66-67); full
    modes1 x modes2 kept per corner (:71-74); the complex product is written as four
    real einsums "bixy,ioxy->boxy" (:51-58) - mathematically the complex einsum."""
    n = x.shape[-1]
    xf = torch.fft.rfft2(x, s=(n, n), norm=norm)
    cout = w0.shape[1]
    out = torch.zeros(x.shape[0], cout, n, n // 2 + 1, dtype=xf.dtype, device=x.device)
    w0c = torch.view_as_complex(w0.contiguous())
    w1c = torch.view_as_complex(w1.contiguous())
    out[:, :, :modes1, :modes2] = _mode_einsum(xf[:, :, :modes1, :modes2], w0c)
    out[:, :, -modes1:, :modes2] = _mode_einsum(xf[:, :, -modes1:, :modes2], w1c)
    return torch.fft.irfft2(out, s=(n, n), norm=norm)


# --------------------------------------------------------------------------
# dialect C: libs.models.pino_models.basics.SpectralConv2d / SpectralConv3d
# --------------------------------------------------------------------------
def spectral_conv_C2d(x, w1, w2, modes1, modes2):
    """Follows libs/models/pino_models/basics.py:79-96 (default 'backward' norm)."""
    xf = torch.fft.rfftn(x, dim=[2, 3])
    cout = w1.shape[1]
    out = torch.zeros(x.shape[0], cout, x.size(-2), x.size(-1) // 2 + 1,
                      dtype=xf.dtype, device=x.device)
    out[:, :, :modes1, :modes2] = _mode_einsum(xf[:, :, :modes1, :modes2], w1)
    out[:, :, -modes1:, :modes2] = _mode_einsum(xf[:, :, -modes1:, :modes2], w2)
    return torch.fft.irfftn(out, s=(x.size(-2), x.size(-1)), dim=[2, 3])


def spectral_conv_C3d(x, w1, w2, w3, w4, modes1, modes2, modes3):
    """Follows libs/models/pino_models/basics.py:114-143: 4 corners over (x, y), last
    dim [:min(Nz/2+1, modes3)]; the truncated spectrum is zero-extended to modes3
    (:125-139), out_ft has last extent modes3 (:122) and irfftn(s=..) crops / pads it
    (:142)."""
    xf = torch.fft.rfftn(x, dim=[2, 3, 4])
    zd = min(xf.shape[4], modes3)
    cout = w1.shape[1]
    out = torch.zeros(x.shape[0], cout, xf.shape[2], xf.shape[3], modes3,
                      dtype=xf.dtype, device=x.device)

    def corner(sx, sy, w):
        coeff = torch.zeros(x.shape[0], x.shape[1], modes1, modes2, modes3,
                            dtype=xf.dtype, device=x.device)
        coeff[..., :zd] = xf[:, :, sx, sy, :zd]
        return _mode_einsum(coeff, w)

    lo1, hi1 = slice(None, modes1), slice(-modes1, None)
    lo2, hi2 = slice(None, modes2), slice(-modes2, None)
    out[:, :, lo1, lo2, :] = corner(lo1, lo2, w1)
    out[:, :, hi1, lo2, :] = corner(hi1, lo2, w2)
    out[:, :, lo1, hi2, :] = corner(lo1, hi2, w3)
    out[:, :, hi1, hi2, :] = corner(hi1, hi2, w4)
    return torch.fft.irfftn(out, s=(x.size(2), x.size(3), x.size(4)), dim=[2, 3, 4])


# --------------------------------------------------------------------------
# pointwise pieces and the FNO model (neuralop/models/tfno.py, fno_block.py)
# --------------------------------------------------------------------------
def conv1x1(x, w, b=None):
    """nn.Conv{N}d(kernel_size=1): w (Cout, Cin, 1, ..), b (Cout,) or None."""
    w2 = w.reshape(w.shape[0], w.shape[1])
    y = torch.einsum("oi,bi...->bo...", w2, x)
    if b is not None:
        y = y + b.reshape(1, -1, *([1] * (x.dim() - 2)))
    return y


def fno_gelu_gate(index, n_layers):
    """fno_block.py:149: GELU after the residual add iff index < n_layers - index
    (default path: use_mlp=False, preactivation=False)."""
    return index < (n_layers - index)


def complex_weight(p, key):
    """Spectral weight stored as a real (.., 2) tensor -> complex view."""
    t = p[key]
    return t if t.is_complex() else torch.view_as_complex(t.contiguous())


def fno_forward(p, x, n_modes, n_layers=4, fft_norm="forward",
                weight_leaf="tensor", return_intermediates=False):
    """neuralop.models.FNO.forward (tfno.py:195-211) on the default path:
    lifting (tfno.py:11-20) -> n_layers x FNOBlocks.forward (fno_block.py:123-170,
    skip = bias-free 1x1 conv, skip_connections.py:31) -> projection (tfno.py:23-38,
    F.gelu exact-erf).  `p` maps reference state_dict names to tensors."""
    order = len(n_modes)
    half = [m // 2 for m in n_modes]
    nw = 2 ** (order - 1)
    h = conv1x1(x, p["lifting.fc.weight"], p["lifting.fc.bias"])
    inter = [h]
    for l in range(n_layers):
        ws = [complex_weight(p, f"fno_blocks.convs.weight.{nw * l + i}.{weight_leaf}")
              for i in range(nw)]
        skip = conv1x1(h, p[f"fno_blocks.fno_skips.{l}.weight"])
        bias = p.get("fno_blocks.convs.bias")
        spec = spectral_conv_A(h, ws, None if bias is None else bias[l], half, fft_norm)
        h = spec + skip
        if fno_gelu_gate(l, n_layers):
            h = F.gelu(h)
        inter.append(h)
    h = conv1x1(h, p["projection.fc1.weight"], p["projection.fc1.bias"])
    h = F.gelu(h)
    y = conv1x1(h, p["projection.fc2.weight"], p["projection.fc2.bias"])
    if return_intermediates:
        return y, inter
    return y


def fno2d_observer_forward(p, p_plane, v_plane=None, n_modes=(12, 12), n_layers=4,
                           use_v_plane=False, prefix="fno2d."):
    """libs/models/fno_models.py:41-57: append an inclusive linspace(0,1,n) grid
    (x then y), NHWC -> NCHW, run FNO2d.  p_plane (B, X, Y, 1)."""
    B, sx, sy = p_plane.shape[0], p_plane.shape[1], p_plane.shape[2]
    gx = torch.linspace(0, 1, sx, dtype=torch.float64).to(p_plane.dtype)
    gy = torch.linspace(0, 1, sy, dtype=torch.float64).to(p_plane.dtype)
    gx = gx.reshape(1, sx, 1, 1).expand(B, sx, sy, 1)
    gy = gy.reshape(1, 1, sy, 1).expand(B, sx, sy, 1)
    parts = [p_plane, v_plane, gx, gy] if use_v_plane else [p_plane, gx, gy]
    xin = torch.cat(parts, dim=-1).permute(0, 3, 1, 2)
    sub = {k[len(prefix):]: v for k, v in p.items() if k.startswith(prefix)}
    return fno_forward(sub, xin, n_modes, n_layers)


def lp_loss_rel_sum(x, y):
    """libs/utilities3.py:323-334 with size_average=False: sum_b ||x-y||_2/||y||_2."""
    n = x.shape[0]
    d = torch.norm(x.reshape(n, -1) - y.reshape(n, -1), 2, 1)
    yn = torch.norm(y.reshape(n, -1), 2, 1)
    return torch.sum(d / yn)


# --------------------------------------------------------------------------
# explicit backward formulas, used to cross-check autograd and
# to document what the HIP backward kernels implement.
# --------------------------------------------------------------------------
def norm_scales(norm, n_total):
    """(s_f, s_i): scale applied by the forward / inverse transform for a torch.fft
    norm string."""
    if norm == "forward":
        return 1.0 / n_total, 1.0
    if norm == "ortho":
        return 1.0 / math.sqrt(n_total), 1.0 / math.sqrt(n_total)
    return 1.0, 1.0 / n_total


def spectral_conv_A_backward(x, weights, dy, half_modes, fft_norm="forward"):
    """Hand-derived gradients of spectral_conv_A (bias excluded): returns
    (dx, [dW_c]).  G = gamma * s_i * F(dy) on the kept modes; dW_c = sum_b conj(X) G;
    GX = G conj(W_c); dx = s_f * Re(F^H zero-extended(GX))."""
    order = len(half_modes)
    sizes = list(x.shape[2:])
    n_total = 1
    for s in sizes:
        n_total *= s
    s_f, s_i = norm_scales(fft_norm, n_total)
    dims = list(range(-order, 0))
    xf = torch.fft.rfftn(x, dim=dims) * s_f
    g = torch.fft.rfftn(dy, dim=dims) * s_i
    wl = sizes[-1]
    gamma = torch.full((wl // 2 + 1,), 2.0, dtype=x.dtype)
    gamma[0] = 1.0
    if wl % 2 == 0:
        gamma[-1] = 1.0
    g = g * gamma
    sel = [((None, m), (-m, None)) for m in half_modes[:-1]] + [((None, half_modes[-1]),)]
    gx_full = torch.zeros(x.shape[0], x.shape[1], *sizes, dtype=xf.dtype)
    dws = []
    for i, bnd in enumerate(itertools.product(*sel)):
        idx = (slice(None), slice(None)) + tuple(slice(*b) for b in bnd)
        xs, gs = xf[idx], g[idx]
        sp = "xyzt"[:order]
        dws.append(torch.einsum(f"bi{sp},bo{sp}->io{sp}", xs.conj(), gs))
        gx_full[idx] = torch.einsum(f"bo{sp},io{sp}->bi{sp}", gs, weights[i].conj())
    dx = torch.fft.ifftn(gx_full, dim=dims, norm="forward").real * s_f
    return dx, dws
