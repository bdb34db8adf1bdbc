"""Generate tests/golden/*.npz by running the REAL reference on CPU.

Runs only in the build container (needs /root/reference, which never travels to
the GPU box).  Usage:  python oracle/make_golden.py [--ref /root/reference]

What is committed is DATA ONLY: inputs, per-parameter scales, outputs and
gradients.  Weights are not stored: each parameter is rebuilt as
scale * oracle.detfill.unit_fill(shape, crc32(name)) on both sides.

Third-party packages the reference imports but the image lacks (no network) are
replaced by in-memory stand-ins below.  Only two of them touch arithmetic on the
path: `tensorly.einsum` (-> torch.einsum; for factorization=None the reference
issues a plain two-operand einsum, neuralop/models/spectral_convolution.py:31-36)
and `tltorch.FactorizedTensor` (-> a dense complex parameter container; it only
stores the weight, spectral_convolution.py:253-268).  Everything else is an empty
placeholder so that `import neuralop` succeeds.
"""
import argparse
import os
import sys
import types

import numpy as np
import torch
from torch import nn

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)

from oracle.detfill import fill_named  # noqa: E402


# ----------------------------------------------------------------------------
# stand-ins for absent third-party modules
# ----------------------------------------------------------------------------
class _Anything:
    def __init__(self, *a, **k):
        pass

    def __call__(self, *a, **k):
        return _Anything()

    def __getattr__(self, k):
        return _Anything()


class _Placeholder(types.ModuleType):
    def __getattr__(self, k):
        if k.startswith("__"):
            raise AttributeError(k)
        return _Anything


class DenseComplexWeight(nn.Module):
    """Dense-only stand-in for tltorch's ComplexDense FactorizedTensor: a complex
    tensor stored as a real (.., 2) nn.Parameter."""
    name = "ComplexDense"

    def __init__(self, shape):
        super().__init__()
        self.shape = tuple(shape)
        self.tensor = nn.Parameter(torch.zeros(*shape, 2))

    @classmethod
    def new(cls, shape, rank=None, factorization="ComplexDense", fixed_rank_modes=None, **kw):
        if "dense" not in factorization.lower():
            raise NotImplementedError("stand-in supports factorization=None only")
        return cls(shape)

    def normal_(self, mean=0.0, std=1.0):
        with torch.no_grad():
            self.tensor.normal_(mean, std)
        return self

    def to_tensor(self):
        return torch.view_as_complex(self.tensor)

    def __getitem__(self, idx):
        return self.to_tensor()[idx]


def install_standins():
    tl = _Placeholder("tensorly")
    tl.set_backend = lambda *_: None
    tl.einsum = torch.einsum
    tl.ndim = lambda t: t.dim()
    plugins = _Placeholder("tensorly.plugins")
    plugins.use_opt_einsum = lambda *_: None
    tl.plugins = plugins
    tlt = _Placeholder("tltorch")
    ft = _Placeholder("tltorch.factorized_tensors")
    core = _Placeholder("tltorch.factorized_tensors.core")
    core.FactorizedTensor = DenseComplexWeight
    tlt.FactorizedTensor = DenseComplexWeight
    mods = {
        "tensorly": tl, "tensorly.plugins": plugins,
        "tltorch": tlt, "tltorch.factorized_tensors": ft,
        "tltorch.factorized_tensors.core": core,
        "tltorch.utils": _Placeholder("tltorch.utils"),
    }
    for name in ["torch_harmonics", "torch_harmonics.examples", "torchvision",
                 "torchvision.transforms", "h5py", "zarr", "wandb", "configmypy",
                 "mpi4py", "opt_einsum", "cv2", "imageio", "torchdiffeq", "pympler",
                 "matlab", "matlab.engine"]:
        mods.setdefault(name, _Placeholder(name))
    for k, v in mods.items():
        if k not in sys.modules:
            sys.modules[k] = v


# ----------------------------------------------------------------------------
# helpers
# ----------------------------------------------------------------------------
def refill_parameters(model, std_override=None):
    """Replace every parameter by scale * unit_fill(crc32(name)); scale = RMS of the
    reference's own init (so magnitudes stay realistic).  Returns {name: scale}."""
    scales = {}
    with torch.no_grad():
        for name, prm in model.named_parameters():
            if prm.is_complex():
                rms = float(torch.view_as_real(prm.detach()).pow(2).mean().sqrt())
            else:
                rms = float(prm.detach().float().pow(2).mean().sqrt()) if prm.numel() else 1.0
            if not np.isfinite(rms) or rms == 0.0:
                rms = 1.0
            scale = float(np.float32(rms * 1.7))      # unit_fill has RMS 1/sqrt(3)
            if std_override and name in std_override:
                scale = std_override[name]
            scales[name] = scale
            if prm.is_complex():
                v = fill_named(name, tuple(prm.shape), scale, complex_=True)
            else:
                v = fill_named(name, tuple(prm.shape), scale)
            prm.copy_(torch.from_numpy(np.asarray(v)).reshape(prm.shape))
    return scales


def input_fill(name, shape, scale=1.0):
    return torch.from_numpy(fill_named("input:" + name, shape, scale))


def grads_of(model):
    out = {}
    for name, prm in model.named_parameters():
        g = prm.grad
        if g is None:
            continue
        g = g.detach()
        out[name] = (torch.view_as_real(g) if g.is_complex() else g).numpy().copy()
    return out


def save(path, **arrs):
    flat = {}
    for k, v in arrs.items():
        if isinstance(v, dict):
            for kk, vv in v.items():
                flat[f"{k}/{kk}"] = np.asarray(vv)
        else:
            flat[k] = np.asarray(v.detach().numpy() if torch.is_tensor(v) else v)
    np.savez_compressed(path, **flat)
    print(f"  wrote {os.path.relpath(path, ROOT)}  ({os.path.getsize(path) / 1024:.0f} KiB)")


# ----------------------------------------------------------------------------
# fixture generators
# ----------------------------------------------------------------------------
def gen_specconv_A(outdir):
    from neuralop.models.spectral_convolution import FactorizedSpectralConv
    cases = {
        # name: (cin, cout, n_modes, spatial, n_layers, fft_norm, batch)
        "A2d": (4, 6, (6, 8), (16, 20), 2, "forward", 3),
        "A2d_ortho_odd": (3, 3, (4, 6), (12, 15), 1, "ortho", 2),
        "A2d_backward": (5, 2, (8, 4), (8, 32), 1, "backward", 2),
        "A3d": (3, 4, (4, 6, 4), (8, 10, 12), 2, "forward", 2),
    }
    for cname, (cin, cout, n_modes, sp, nl, norm, B) in cases.items():
        torch.manual_seed(0)
        conv = FactorizedSpectralConv(cin, cout, n_modes, n_layers=nl, fft_norm=norm,
                                      factorization=None, implementation="factorized",
                                      rank=1.0)
        scales = refill_parameters(conv)
        x = input_fill(cname + ".x", (B, cin, *sp)).requires_grad_(True)
        dy = input_fill(cname + ".dy", (B, cout, *sp))
        idx = nl - 1
        y = conv(x, idx)
        y.backward(dy)
        save(os.path.join(outdir, f"specconv_{cname}.npz"),
             x=x, dy=dy, y=y, dx=x.grad, grads=grads_of(conv), scales=scales,
             meta=np.array([cin, cout, nl, idx, B, len(n_modes), *n_modes, *sp]),
             fft_norm=np.array(norm))


def gen_specconv_A_overlap(outdir):
    """Overlapping corners (2 * half_modes[0] > H): the reference assigns the corners in order into one zero-filled spectrum, so
    rows that belong to both take the SECOND corner's product (spectral_convolution.py:330-337).  The fixture pins what the
    reference does for the oracle (tests/test_oracle_golden.py) and for the engine, which since round 6 follows the same
    order on 2-D grids (tests/test_parity_gpu.py::test_specconv_A_golden[A2d_overlap])."""
    from neuralop.models.spectral_convolution import FactorizedSpectralConv
    cin, cout, n_modes, sp, B = 3, 4, (12, 6), (8, 16), 2
    torch.manual_seed(0)
    conv = FactorizedSpectralConv(cin, cout, n_modes, n_layers=1, fft_norm="forward", factorization=None,
                                  implementation="factorized", rank=1.0)
    scales = refill_parameters(conv)
    x = input_fill("A2d_overlap.x", (B, cin, *sp)).requires_grad_(True)
    dy = input_fill("A2d_overlap.dy", (B, cout, *sp))
    y = conv(x, 0)
    y.backward(dy)
    save(os.path.join(outdir, "specconv_A2d_overlap.npz"), x=x, dy=dy, y=y, dx=x.grad, grads=grads_of(conv), scales=scales,
         meta=np.array([cin, cout, 1, 0, B, len(n_modes), *n_modes, *sp]), fft_norm=np.array("forward"))


def gen_specconv_B(outdir):
    from neuralop.models.rno import SpectralConv2d
    for cname, (cin, cout, m1, m2, n, B) in {"B2d": (4, 5, 3, 5, 16, 3),
                                             "B2d_full": (2, 2, 6, 7, 12, 2)}.items():
        torch.manual_seed(0)
        conv = SpectralConv2d(cin, cout, m1, m2)
        scales = refill_parameters(conv)
        x = input_fill(cname + ".x", (B, cin, n, n)).requires_grad_(True)
        dy = input_fill(cname + ".dy", (B, cout, n, n))
        y = conv(x)
        y.backward(dy)
        save(os.path.join(outdir, f"specconv_{cname}.npz"), x=x, dy=dy, y=y, dx=x.grad,
             grads=grads_of(conv), scales=scales, meta=np.array([cin, cout, m1, m2, n, B]))


def gen_specconv_C(outdir):
    from libs.models.pino_models.basics import SpectralConv2d, SpectralConv3d
    torch.manual_seed(0)
    cin, cout, m1, m2, B = 4, 3, 3, 4, 2
    conv = SpectralConv2d(cin, cout, m1, m2)
    scales = refill_parameters(conv)
    x = input_fill("C2d.x", (B, cin, 10, 16)).requires_grad_(True)
    dy = input_fill("C2d.dy", (B, cout, 10, 16))
    y = conv(x)
    y.backward(dy)
    save(os.path.join(outdir, "specconv_C2d.npz"), x=x, dy=dy, y=y, dx=x.grad,
         grads=grads_of(conv), scales=scales, meta=np.array([cin, cout, m1, m2, 10, 16, B]))
    # 3-D: regular; Nz/2+1 < modes3; T = 1 (PINObserverFullField, model_timestep 1)
    for cname, (cin, cout, m1, m2, m3, sp, B) in {
            "C3d": (3, 4, 2, 3, 3, (8, 8, 10), 2),
            "C3d_shortz": (3, 3, 3, 2, 6, (8, 6, 6), 2),
            "C3d_T1": (4, 4, 3, 3, 3, (8, 8, 1), 2)}.items():
        torch.manual_seed(0)
        conv = SpectralConv3d(cin, cout, m1, m2, m3)
        scales = refill_parameters(conv)
        x = input_fill(cname + ".x", (B, cin, *sp)).requires_grad_(True)
        dy = input_fill(cname + ".dy", (B, cout, *sp))
        y = conv(x)
        y.backward(dy)
        save(os.path.join(outdir, f"specconv_{cname}.npz"), x=x, dy=dy, y=y, dx=x.grad,
             grads=grads_of(conv), scales=scales,
             meta=np.array([cin, cout, m1, m2, m3, *sp, B]))


def _lp_rel_sum(x, y):
    from libs.utilities3 import LpLoss
    return LpLoss(size_average=False)(x, y)


def gen_fno_models(outdir):
    from neuralop.models import FNO2d, FNO3d
    cfgs = {
        # name: (ctor, args, input shape, keep_all_grads)
        "fno2d_cfg1": (FNO2d, (8, 8, 32), (4, 3, 64, 64), True),
        "fno2d_cfg2small": (FNO2d, (12, 12, 64), (2, 3, 128, 128), False),
        "fno3d_small": (FNO3d, (8, 8, 8, 32), (1, 3, 32, 32, 32), False),
    }
    for cname, (ctor, args, shp, keep_all) in cfgs.items():
        torch.manual_seed(0)
        model = ctor(*args, in_channels=3, out_channels=1)
        scales = refill_parameters(model)
        x = input_fill(cname + ".x", shp)
        tgt = input_fill(cname + ".target", (shp[0], 1, *shp[2:]))
        y = model(x)
        loss = _lp_rel_sum(y, tgt)
        loss.backward()
        g = grads_of(model)
        gnorm = {k: np.array([np.sqrt((v.astype(np.float64) ** 2).sum())]) for k, v in g.items()}
        if not keep_all:
            # keep small params whole, and a leading slab of the big spectral weights
            g = {k: (v if v.size <= 20000 else v.reshape(-1)[:4096].copy()) for k, v in g.items()}
        save(os.path.join(outdir, f"{cname}.npz"), x=x, target=tgt, y=y,
             loss=np.array([float(loss.detach())]), grads=g, gnorm=gnorm, scales=scales,
             shapes={k: np.array(v.shape) for k, v in model.state_dict().items()})


def gen_fno_models_fp64(outdir):
    """Error-budget vectors for the three full-model fixtures: the same inputs and parameters evaluated in float64.

    The reference itself cannot run in float64: FactorizedSpectralConv.forward hard-codes `x.float()` before the
    transform and a `torch.cfloat` output spectrum (neuralop/models/spectral_convolution.py:324, :326).  The float64
    evaluation is therefore oracle/fno_oracle.py::fno_forward (pinned bit-for-bit to the reference in float32 by the
    fixtures above) on float64 copies of the SAME float32 inputs / parameters.  Stored next to it: how far the
    reference's OWN float32 result is from that float64 value, per parameter (`ref32_err`), which is the budget the
    GPU tests compare the engine's error against (tests/test_parity_gpu.py::test_fno_model_fp64_error_budget)."""
    from neuralop.models import FNO2d, FNO3d
    from oracle import fno_oracle as O
    cfgs = {
        "fno2d_cfg1": (FNO2d, (8, 8, 32), (4, 3, 64, 64)),
        "fno2d_cfg2small": (FNO2d, (12, 12, 64), (2, 3, 128, 128)),
        "fno3d_small": (FNO3d, (8, 8, 8, 32), (1, 3, 32, 32, 32)),
    }

    def rel(a, b):
        a = np.asarray(a, np.float64).reshape(-1)
        b = np.asarray(b, np.float64).reshape(-1)
        return float(np.sqrt(((a - b) ** 2).sum()) / np.sqrt((b ** 2).sum()))
    for cname, (ctor, args, shp) in cfgs.items():
        torch.manual_seed(0)
        model = ctor(*args, in_channels=3, out_channels=1)
        refill_parameters(model)
        x = input_fill(cname + ".x", shp)
        tgt = input_fill(cname + ".target", (shp[0], 1, *shp[2:]))
        y32 = model(x)
        l32 = _lp_rel_sum(y32, tgt)
        l32.backward()
        g32 = grads_of(model)
        p64 = {k: v.detach().double().requires_grad_(True) for k, v in model.state_dict().items()}
        y64 = O.fno_forward(p64, x.double(), args[:-1])
        l64 = O.lp_loss_rel_sum(y64, tgt.double())
        l64.backward()
        g64 = {k: p64[k].grad.numpy() for k in g32}
        ref32_err = {k: np.array([rel(g32[k], g64[k])]) for k in g32}
        gnorm64 = {k: np.array([np.sqrt((g64[k] ** 2).sum())]) for k in g64}
        # same subsets as the float32 fixtures: small parameters whole, a leading slab of the big spectral weights
        keep = {k: (v if v.size <= 20000 else v.reshape(-1)[:4096].copy()) for k, v in g64.items()}
        print(f"    {cname}: y ref32-vs-fp64 {rel(y32.detach().numpy(), y64.detach().numpy()):.2e}; "
              f"worst grad {max(float(v[0]) for v in ref32_err.values()):.2e}")
        save(os.path.join(outdir, f"{cname}_fp64.npz"), y64=y64, loss64=np.array([float(l64.detach())]), grads64=keep,
             gnorm64=gnorm64, ref32_err=ref32_err,
             y_ref32_err=np.array([rel(y32.detach().numpy(), y64.detach().numpy())]))


def gen_observer_adam(outdir):
    """Trainer counterpart of run_pde_observers.py:185-193 (FNO2dObserver, LpLoss sum,
    Adam lr 1e-3 wd 1e-4): 3 steps, record the loss trajectory."""
    from libs.models.fno_models import FNO2dObserver
    torch.manual_seed(0)
    model = FNO2dObserver(8, 8, 16)
    scales = refill_parameters(model)
    B, S = 4, 32
    p_plane = input_fill("obs.p", (B, S, S, 1))
    target = input_fill("obs.t", (B, S, S, 1))
    mean = input_fill("obs.mean", (S, S), 0.3).numpy()
    std = np.abs(input_fill("obs.std", (S, S), 0.5).numpy()) + 0.5
    opt = torch.optim.Adam(model.parameters(), lr=1e-3, weight_decay=1e-4)
    mean_t, std_t = torch.from_numpy(mean), torch.from_numpy(std)
    losses, y0 = [], None
    for step in range(3):
        opt.zero_grad()
        pred = model(p_plane, None)                       # (B,1,S,S)
        pred = pred.reshape(B, S, S)
        # NormalizerGivenMeanStd.cuda_decode (libs/utilities3.py:115-129): x*(std+eps)+mean
        pd = pred * (std_t + 1e-5) + mean_t
        td = target.reshape(B, S, S) * (std_t + 1e-5) + mean_t
        loss = _lp_rel_sum(pd, td)
        if step == 0:
            y0 = pred.detach().clone()
        loss.backward()
        opt.step()
        losses.append(float(loss))
    save(os.path.join(outdir, "observer_adam3.npz"), p_plane=p_plane, target=target,
         mean=mean, std=std, y0=y0, losses=np.array(losses), scales=scales)




# ----------------------------------------------------------------------------
# RNO2d (neuralop/models/rno.py) and the PINO observers (libs/models/pino_models)
# ----------------------------------------------------------------------------
def gen_rno(outdir):
    from neuralop.models import RNO2d
    cfgs = {"rno2d_small": dict(args=(4, 4, 8, 1), kw=dict(layer_num=2), shp=(2, 2, 16, 16, 1)),
            "rno2d_shipped": dict(args=(12, 12, 34, 0), kw=dict(layer_num=1), shp=(2, 1, 32, 32, 1))}
    for cname, c in cfgs.items():
        torch.manual_seed(0)
        model = RNO2d(*c["args"], **c["kw"]).eval()       # eval: dropout off (rno.py:89,98,317)
        scales = refill_parameters(model)
        x = input_fill(cname + ".x", c["shp"])
        y = model(x)
        tgt = input_fill(cname + ".t", tuple(y.shape))
        loss = _lp_rel_sum(y, tgt)
        loss.backward()
        g = grads_of(model)
        gnorm = {k: np.array([np.sqrt((v.astype(np.float64) ** 2).sum())]) for k, v in g.items()}
        g = {k: (v if v.size <= 20000 else v.reshape(-1)[:2048].copy()) for k, v in g.items()}
        save(os.path.join(outdir, f"{cname}.npz"), x=x, target=tgt, y=y, loss=np.array([float(loss.detach())]),
             grads=g, gnorm=gnorm, scales=scales,
             shapes={k: np.array(v.shape) for k, v in model.state_dict().items()})


def gen_pino(outdir):
    from libs.models.pino_models.pinobserver import PINObserver2d, PINObserverFullField
    # PINObserverFullField as run_pde_observers.py:201-207 feeds it: x (B,X,Y,T,1), re (B,1)
    torch.manual_seed(0)
    m = PINObserverFullField(plane_num=3, modes1=[4] * 4, modes2=[4] * 4, modes3=[4] * 4, fc_dim=16,
                             layers=[8] * 5, in_dim=1, out_dim=1, act="gelu", pad_ratio=[0.0, 0.0625])
    scales = refill_parameters(m)
    x = input_fill("pinoff.x", (2, 16, 16, 1, 1))
    re = torch.from_numpy(np.array([[180.0], [395.0]], dtype=np.float32))
    y = m(x, re)
    tgt = input_fill("pinoff.t", tuple(y.shape))
    loss = _lp_rel_sum(y, tgt)
    loss.backward()
    g = grads_of(m)
    save(os.path.join(outdir, "pino_fullfield_small.npz"), x=x, re=re, target=tgt, y=y,
         loss=np.array([float(loss.detach())]), grads=g, scales=scales,
         shapes={k: np.array(v.shape) for k, v in m.state_dict().items()})
    # PINObserver2d as train_pino.py:154-160 builds it (T padded by round(T * 0.0625))
    torch.manual_seed(0)
    m = PINObserver2d(modes1=[3] * 4, modes2=[3] * 4, modes3=[3] * 4, fc_dim=16, layers=[8] * 5, in_dim=4,
                      out_dim=1, act="gelu", pad_ratio=[0.0, 0.0625])
    scales = refill_parameters(m)
    x = input_fill("pino2d.x", (2, 12, 12, 16, 4))
    re = torch.from_numpy(np.array([[0.4], [0.9]], dtype=np.float32))
    y = m(x, re)
    tgt = input_fill("pino2d.t", tuple(y.shape))
    loss = _lp_rel_sum(y, tgt)
    loss.backward()
    g = grads_of(m)
    save(os.path.join(outdir, "pino2d_small.npz"), x=x, re=re, target=tgt, y=y,
         loss=np.array([float(loss.detach())]), grads=g, scales=scales,
         shapes={k: np.array(v.shape) for k, v in m.state_dict().items()})


def gen_pino_loss(outdir):
    """PINO residual loss straight from the reference file (libs/pino_utils/losses.py; the copy in
    libs/envs/diff_control_env.py is line-for-line the same but sits in a package whose __init__ needs MATLAB)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("ref_pino_losses", os.path.join(sys.path[0], "libs", "pino_utils", "losses.py"))
    L = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(L)
    for tag, (B, n, nt) in {"n32": (2, 32, 6), "n64": (2, 64, 5), "n128": (1, 128, 4), "n256": (1, 256, 3)}.items():
        u = input_fill("pinoloss.u." + tag, (B, n, n, nt)).requires_grad_(True)
        u0 = input_fill("pinoloss.u0." + tag, (B, n, n))
        re = torch.from_numpy(np.array([180.0, 395.0][:B], dtype=np.float32))
        v = 1.0 / re
        f = L.get_forcing(n)
        t_interval = 0.5
        du = L.FDM_NS_vorticity(u, v, t_interval)
        loss_ic, loss_f = L.PINO_loss3d(u, u0, f, v, t_interval)          # == Channelflow_PINO_loss
        (5.0 * loss_ic + 1.0 * loss_f).backward()                        # configs/pino-observer-finetune-1s.yaml weights
        save(os.path.join(outdir, f"pino_loss_{tag}.npz"), meta=np.array([B, n, nt]), re=re, t_interval=np.float32(t_interval),
             forcing=f, loss_ic=loss_ic.detach(), loss_f=loss_f.detach(), du_residual=du.detach(), grad_u=u.grad)


def gen_pde_dataset(outdir):
    """Plane datasets read by the reference's own classes (libs/pde_data_loader.py) from a small folder in its on-disk
    format.  The normaliser constructor calls `.cuda()` on its statistics (libs/utilities3.py:88-90); this container has
    no GPU, so `.cuda()` is made the identity for the duration of the call (it only creates device copies)."""
    import tempfile
    import types
    from libs.pde_data_loader import PDEDataset, SequentialPDEDataset
    rng = np.random.default_rng(7)
    n_files, full = 6, (12, 10)
    planes = {"P_planes": rng.standard_normal((n_files,) + full).astype(np.float32) * 3 + 1,
              "V_planes": rng.standard_normal((n_files,) + full).astype(np.float32) * 0.5 - 2}
    meta = {k: dict(mean=v.mean(0), std=v.std(0)) for k, v in planes.items()}
    orig_cuda = torch.Tensor.cuda
    torch.Tensor.cuda = lambda self, *a, **k: self
    try:
        with tempfile.TemporaryDirectory() as d:
            for k, v in planes.items():
                for i in range(n_files):
                    np.save(os.path.join(d, f"{k}_{i:06d}.npy"), v[i])
            np.save(os.path.join(d, "metadata.npy"), meta, allow_pickle=True)
            args = types.SimpleNamespace(model_timestep=2)
            idx = [4, 0, 3, 5, 1, 2]
            ds = PDEDataset(args, d, idx, 2, 5, 4)                       # downsample 2, crop to 5 x 4
            items = [ds[i] for i in range(len(ds))]
            ds_patch = PDEDataset(args, d, idx[:3], 1, 6, 5, use_patch=True)   # 12x10 -> four 6x5 patches
            patch_items = [ds_patch[i] for i in range(len(ds_patch))]
            # SequentialPDEDataset cannot be constructed in the reference (it reads self.p_plane_mean before anything
            # sets it, pde_data_loader.py:97-105): no vectors for it
    finally:
        torch.Tensor.cuda = orig_cuda
    save(os.path.join(outdir, "pde_dataset.npz"), p_raw=planes["P_planes"], v_raw=planes["V_planes"],
         data_index=np.array(idx), p_items=torch.stack([a for a, _ in items]), v_items=torch.stack([b for _, b in items]),
         p_patch=torch.stack([a for a, _ in patch_items]), v_patch=torch.stack([b for _, b in patch_items]))


def fullfield_folder(d, n_files=4, Nx=6, Ny=8, Nz=5, seed=11):
    """A tiny folder in the FullFieldNSDataset on-disk format (U/W (Nx, Ny+1, Nz), V (Nx, Ny, Nz) per timestep + metadata.npy)."""
    rng = np.random.default_rng(seed)
    f = {"U_field": rng.standard_normal((n_files, Nx, Ny + 1, Nz)).astype(np.float32) + 1,
         "V_field": rng.standard_normal((n_files, Nx, Ny, Nz)).astype(np.float32) * 0.4 - 0.1,
         "W_field": rng.standard_normal((n_files, Nx, Ny + 1, Nz)).astype(np.float32) * 0.3}
    meta = {k: dict(mean=v.mean(0), std=v.std(0)) for k, v in f.items()}
    meta["U_field"]["dpdx"] = [float(x) for x in rng.uniform(0.002, 0.004, n_files)]
    meta["re"] = 178.1899
    p = rng.standard_normal((n_files, Nx, Nz)).astype(np.float32)
    meta["P_planes"] = dict(mean=p.mean(0), std=p.std(0))
    for k, v in f.items():
        for i in range(n_files):
            np.save(os.path.join(d, f"{k}_{i:06d}.npy"), v[i])
    np.save(os.path.join(d, "metadata.npy"), meta, allow_pickle=True)
    return f, meta


def gen_fullfield_dataset(outdir):
    """FullFieldNSDataset items from the reference's own class (libs/pde_data_loader.py:135-198); `.cuda()` patched to the
    identity as in gen_pde_dataset."""
    import tempfile
    import types
    from libs.pde_data_loader import FullFieldNSDataset
    orig_cuda = torch.Tensor.cuda
    torch.Tensor.cuda = lambda self, *a, **k: self
    try:
        with tempfile.TemporaryDirectory() as d:
            f, meta = fullfield_folder(d)
            idx, planes = [2, 0, 3, 1], [-3, -2, 1]
            ds = FullFieldNSDataset(types.SimpleNamespace(model_timestep=2), d, idx, planes, 1, 6, 5)
            items = [ds[i] for i in range(len(ds))]
    finally:
        torch.Tensor.cuda = orig_cuda
    names = ["v_plane", "v_field", "u", "v", "w", "re", "dpdx"]
    save(os.path.join(outdir, "fullfield_dataset.npz"), data_index=np.array(idx), plane_indexs=np.array(planes),
         **{n: torch.stack([it[k] for it in items]) for k, n in enumerate(names)})


def gen_kf_dataset(outdir):
    """MultipleReynoldsKFaDataset items from the reference's own class (libs/pino_utils/datasets.py:548-617) on a tiny
    multi-Reynolds .npz; its leftover pdb.set_trace() (:588) is patched to a no-op."""
    import pdb
    import tempfile
    from libs.pino_utils.datasets import MultipleReynoldsKFaDataset
    pdb.set_trace = lambda *a, **k: None
    rng = np.random.default_rng(3)
    raw = rng.standard_normal((4, 9, 8, 8)).astype(np.float32)
    res = np.array([300.0, 400.0, 500.0, 600.0], dtype=np.float32)
    out = {"raw": raw, "re_file": res}
    with tempfile.TemporaryDirectory() as d:
        path = os.path.join(d, "multi_reynolds_tiny.npz")
        np.savez(path, data1=raw, data2=res)
        for tag, kw in {"half": dict(data_res=[8, 8, 9], pde_res=[8, 8, 9], t_duration=0.5, n_samples=2, offset=1),
                        "quarter_sub": dict(data_res=[4, 4, 9], pde_res=[4, 4, 9], t_duration=0.25, n_samples=3, offset=0)}.items():
            ds = MultipleReynoldsKFaDataset(paths=[path], raw_res=[8, 8, 9], **kw)
            items = [ds[i] for i in range(len(ds))]
            out[f"{tag}_u"] = torch.stack([it[0] for it in items])
            out[f"{tag}_a"] = torch.stack([it[1] for it in items])
            out[f"{tag}_re"] = np.array([float(it[2]) for it in items])
    save(os.path.join(outdir, "kf_dataset.npz"), **out)


def chanflow_inputs(tag, Nx, Ny, Nz, dtype=torch.float32):
    """U, Vgt, V, W of one deterministic channel-flow sample (shared with the tests via oracle.detfill)."""
    U = 1.0 + input_fill(f"chanflow.U.{tag}", (Nx, Ny + 1, Nz), 0.5)
    Vgt = input_fill(f"chanflow.Vgt.{tag}", (Nx, Ny, Nz), 0.3)
    V = Vgt + input_fill(f"chanflow.dV.{tag}", (Nx, Ny, Nz), 0.1)
    W = input_fill(f"chanflow.W.{tag}", (Nx, Ny + 1, Nz), 0.3)
    return [a.to(dtype) for a in (U, Vgt, V, W)]


def gen_chanflow(outdir):
    """Channel-flow RHS and physics-informed loss from the reference's own NSControlEnvMatlab.compute_rhs_py / pde_loss
    (libs/envs/control_env.py:429-530, 627-633).  The class constructor starts MATLAB, so the two methods are called unbound
    on a namespace carrying exactly the attributes they read (dx, dz, y, ym, yg, Ny, nu, dPdx); pde_loss's two
    pdb.set_trace() calls are patched to no-ops.  dx, dz are numpy float64 scalars and y, ym, yg 1-D float64 arrays, so every
    metric enters as a 0-dim float64 tensor and the arithmetic stays in the dtype of the fields (fp32 and fp64 runs are both
    stored).  With the (N, 1) arrays scipy.io.loadmat hands the real constructor (control_env.py:151-165) the same code
    promotes fp32 fields to fp64 at the first division."""
    import pdb
    from oracle.chanflow_oracle import tanh_grid
    import libs.envs.control_env as ce
    pdb.set_trace = lambda *a, **k: None
    nu = 3.076923076923077e-04                                             # control_env.py:26
    for tag, (Nx, Ny, Nz, stride) in {"small": (8, 10, 6, 1), "odd": (6, 7, 10, 1), "shipped": (32, 130, 32, 7)}.items():
        y, ym, yg = tanh_grid(Ny)
        ns = types.SimpleNamespace(dx=np.float64(2 * np.pi / Nx), dz=np.float64(2 * np.pi / Nz), y=y, ym=ym, yg=yg, Ny=Ny, nu=nu, dPdx=0.57231059E-01 ** 2)
        ns.compute_rhs_py = lambda *a, **k: ce.NSControlEnvMatlab.compute_rhs_py(ns, *a, **k)
        out = {"meta": np.array([Nx, Ny, Nz, stride]), "nu": np.float64(nu), "dpdx": np.float64(ns.dPdx)}
        for dt, dn in ((torch.float32, "f32"), (torch.float64, "f64")):
            U, Vgt, V, W = chanflow_inputs(tag, Nx, Ny, Nz, dt)
            V.requires_grad_(True)
            F = ce.NSControlEnvMatlab.compute_rhs_py(ns, U, V.detach(), W, torch.tensor(ns.dPdx, dtype=dt))
            loss = ce.NSControlEnvMatlab.pde_loss(ns, U, Vgt, V, W, torch.tensor(ns.dPdx, dtype=dt))
            loss.backward()
            sub = lambda a: a.detach().reshape(-1)[::stride]
            out.update({f"Fu_{dn}": sub(F[0]), f"Fv_{dn}": sub(F[1]), f"Fw_{dn}": sub(F[2]), f"loss_{dn}": loss.detach(),
                        f"gradV_{dn}": sub(V.grad), f"norms_{dn}": np.array([float(a.norm()) for a in F])})
        save(os.path.join(outdir, f"chanflow_{tag}.npz"), **out)


def gen_options(outdir):
    """Constructor options beside the accelerated configuration: separable weights, incremental
    modes, output scaling (one convolution, and a block with its resampled skip), all on the dense weight container."""
    from neuralop.models.spectral_convolution import FactorizedSpectralConv
    from neuralop.models.fno_block import FNOBlocks
    cases = {
        # name: (cin, cout, n_modes, spatial, n_layers, fft_norm, batch, ctor options)
        "A2d_separable": (4, 4, (6, 8), (16, 20), 2, "forward", 2, dict(separable=True)),
        "A2d_incremental": (4, 6, (8, 8), (16, 16), 1, "forward", 2, dict(incremental_n_modes=(4, 6))),
        "A2d_scaled": (3, 5, (6, 6), (12, 16), 2, "forward", 2, dict(output_scaling_factor=[2.0, 0.5])),
        "A3d_separable": (3, 3, (4, 4, 4), (8, 8, 10), 1, "ortho", 2, dict(separable=True)),
    }
    for cname, (cin, cout, n_modes, sp, nl, norm, B, opts) in cases.items():
        torch.manual_seed(0)
        conv = FactorizedSpectralConv(cin, cout, n_modes, n_layers=nl, fft_norm=norm, factorization=None,
                                      implementation="factorized", rank=1.0, **opts)
        scales = refill_parameters(conv)
        x = input_fill(cname + ".x", (B, cin, *sp)).requires_grad_(True)
        idx = nl - 1
        y = conv(x, idx)
        dy = input_fill(cname + ".dy", tuple(y.shape))
        y.backward(dy)
        save(os.path.join(outdir, f"specconv_{cname}.npz"), x=x, dy=dy, y=y, dx=x.grad, grads=grads_of(conv),
             scales=scales, meta=np.array([cin, cout, nl, idx, B, len(n_modes), *n_modes, *sp]), fft_norm=np.array(norm))
    for cname, (sp, scale) in {"blocks2d_scaled": ((12, 16), [[1.5, 1.5], [0.5, 1.0]]),
                               "blocks3d_scaled": ((8, 8, 8), [[1.5, 1.0, 1.0], [1.0, 1.0, 1.0]])}.items():
        torch.manual_seed(0)
        n_modes = (4,) * len(sp)
        blk = FNOBlocks(4, 4, n_modes, output_scaling_factor=scale, n_layers=2, fft_norm="forward",
                        factorization=None, implementation="factorized")
        scales = refill_parameters(blk)
        x = input_fill(cname + ".x", (2, 4, *sp)).requires_grad_(True)
        y = blk(blk(x, 0), 1)
        dy = input_fill(cname + ".dy", tuple(y.shape))
        y.backward(dy)
        save(os.path.join(outdir, f"{cname}.npz"), x=x, dy=dy, y=y, dx=x.grad, grads=grads_of(blk), scales=scales,
             scale=np.array(scale), sp=np.array(sp))


def gen_regressor3d(outdir):
    """neuralop.models.SpectralRegressor (the 3-D one, spectral_regressor.py:93-201), eval mode (dropout off)."""
    from neuralop.models import SpectralRegressor
    cases = {"regressor3d_small": dict(kw=dict(in_dim=5, n_hidden=5, freq_dim=6, out_dim=2, modes=3, spacial_dim=3),
                                       shp=(2, 8, 8, 10, 5)),
             # width 32 tiles the engine's pointwise kernel; modes 6 > Nz/2+1 = 5: zero-padded last-dim spectrum
             "regressor3d_w32": dict(kw=dict(in_dim=32, n_hidden=32, freq_dim=32, out_dim=1, modes=6, spacial_dim=3,
                                             activation='relu', last_activation=True), shp=(2, 16, 16, 8, 32))}
    for cname, c in cases.items():
        torch.manual_seed(0)
        model = SpectralRegressor(**c["kw"]).eval()
        scales = refill_parameters(model)
        x = input_fill(cname + ".x", c["shp"]).requires_grad_(True)
        y = model(x)
        dy = input_fill(cname + ".dy", tuple(y.shape))
        y.backward(dy)
        g = grads_of(model)
        gnorm = {k: np.array([np.sqrt((v.astype(np.float64) ** 2).sum())]) for k, v in g.items()}
        g = {k: (v if v.size <= 20000 else v.reshape(-1)[:2048].copy()) for k, v in g.items()}
        save(os.path.join(outdir, f"{cname}.npz"), x=x, dy=dy, y=y, dx=x.grad, grads=g, gnorm=gnorm, scales=scales,
             shapes={k: np.array(v.shape) for k, v in model.state_dict().items()})


def gen_rno_predict(outdir):
    """RNO2d.predict (rno.py:370-379): autoregressive roll-out, every prediction fed back as the next one-step input."""
    from neuralop.models import RNO2d
    torch.manual_seed(0)
    model = RNO2d(4, 4, 8, 0, layer_num=2).eval()
    scales = refill_parameters(model)
    x = input_fill("rno2d_predict.x", (2, 1, 16, 16, 1))
    y = model.predict(x, num_steps=3)
    tgt = input_fill("rno2d_predict.t", tuple(y.shape))
    loss = _lp_rel_sum(y, tgt)
    loss.backward()
    save(os.path.join(outdir, "rno2d_predict.npz"), x=x, target=tgt, y=y, loss=np.array([float(loss.detach())]),
         grads=grads_of(model), scales=scales, shapes={k: np.array(v.shape) for k, v in model.state_dict().items()})


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ref", default="/root/reference")
    ap.add_argument("--out", default=os.path.join(ROOT, "tests", "golden"))
    ap.add_argument("--only", default="")
    args = ap.parse_args()
    install_standins()
    sys.path.insert(0, args.ref)
    os.makedirs(args.out, exist_ok=True)
    torch.set_num_threads(8)
    gens = [gen_specconv_A, gen_specconv_A_overlap, gen_specconv_B, gen_specconv_C, gen_fno_models, gen_fno_models_fp64, gen_observer_adam, gen_rno, gen_pino, gen_pino_loss, gen_pde_dataset, gen_fullfield_dataset, gen_kf_dataset, gen_chanflow, gen_options, gen_regressor3d, gen_rno_predict]
    for g in gens:
        if args.only and args.only not in g.__name__:
            continue
        print(g.__name__)
        g(args.out)


if __name__ == "__main__":
    main()
