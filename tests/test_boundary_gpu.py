"""GPU tests of the drop-in boundary beside the accelerated configuration: constructor options
that run as torch compositions or on sliced weights, the 3-D `neuralop.models.SpectralRegressor`, `RNO2d.predict` -
every one against vectors generated from the reference itself (oracle/make_golden.py::gen_options / gen_regressor3d /
gen_rno_predict)."""
import warnings

import numpy as np
import pytest
import torch

from oracle import fno_oracle as O
from tests.util import load_golden, rebuild_params, rel_l2

pytestmark = pytest.mark.gpu

TOL = 1e-5          # north star: relative L2, outputs and gradients


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    from pde_policylearning_amd import _lib
    _lib.lib()
    return torch.device("cuda:0")


def _t(a, dev, grad=False):
    t = torch.from_numpy(np.array(a)).to(dev)
    return t.requires_grad_(True) if grad else t


def _cpu(t):
    return t.detach().cpu().numpy()


def _load(model, g, dev, complex_names=()):
    shapes = g["shapes"] if "shapes" in g else {k: tuple(v.shape) for k, v in model.state_dict().items()}
    model.load_state_dict(rebuild_params(g["scales"], shapes, complex_names=complex_names), strict=True)
    return model.to(dev)


def _check_grads(model, g, tol=TOL):
    seen = 0
    for name, prm in model.named_parameters():
        if name not in g["grads"]:
            continue
        ref, got = g["grads"][name], prm.grad
        got = _cpu(torch.view_as_real(got) if got.is_complex() else got)
        if ref.shape != got.shape:                    # large tensors are stored as their first 2048 values + the norm
            assert abs(np.sqrt((got.astype(np.float64) ** 2).sum()) / float(g["gnorm"][name][0]) - 1) < tol, name
            got = got.reshape(-1)[:ref.size]
        assert rel_l2(got, ref) < tol, name
        seen += 1
    assert seen


OPTS = {"A2d_separable": dict(separable=True), "A3d_separable": dict(separable=True),
        "A2d_incremental": dict(incremental_n_modes=(4, 6)), "A2d_scaled": dict(output_scaling_factor=[2.0, 0.5])}


@pytest.mark.parametrize("case", sorted(OPTS))
def test_specconv_constructor_options_golden(dev, case):
    """separable / output_scaling_factor (torch compositions) and incremental_n_modes (engine, sliced weights)."""
    from pde_policylearning_amd.neuralop.models import FactorizedSpectralConv
    g = load_golden("specconv_" + case)
    cin, cout, nl, idx, B, order = [int(v) for v in g["meta"][:6]]
    n_modes = [int(v) for v in g["meta"][6:6 + order]]
    with warnings.catch_warnings():
        warnings.simplefilter("ignore", RuntimeWarning)
        conv = FactorizedSpectralConv(cin, cout, n_modes, n_layers=nl, fft_norm=str(g["fft_norm"]), factorization=None,
                                      implementation="factorized", rank=1.0, **OPTS[case])
        conv = _load(conv, g, dev)
        x = _t(g["x"], dev, True)
        y = conv(x, idx)
    assert y.shape == g["y"].shape
    assert rel_l2(_cpu(y), g["y"]) < TOL
    y.backward(_t(g["dy"], dev))
    assert rel_l2(_cpu(x.grad), g["dx"]) < TOL
    nw = 2 ** (order - 1)
    for name, prm in conv.named_parameters():
        layer_of = int(name.split(".")[1]) // nw if name.startswith("weight.") else idx
        if layer_of != idx:
            continue
        ref = g["grads"][name]
        got = _cpu(prm.grad)
        if name == "bias":
            got, ref = got[idx], ref[idx]
        assert rel_l2(got, ref) < TOL, name


@pytest.mark.parametrize("case", ["blocks2d_scaled", "blocks3d_scaled"])
def test_fno_blocks_output_scaling_golden(dev, case):
    from pde_policylearning_amd.neuralop.models import FNOBlocks
    g = load_golden(case)
    sp = [int(v) for v in g["sp"]]
    with warnings.catch_warnings():
        warnings.simplefilter("ignore", RuntimeWarning)
        blk = FNOBlocks(4, 4, (4,) * len(sp), output_scaling_factor=g["scale"].tolist(), n_layers=2, fft_norm="forward",
                        factorization=None, implementation="factorized")
        blk = _load(blk, g, dev)
        x = _t(g["x"], dev, True)
        y = blk(blk(x, 0), 1)
    assert rel_l2(_cpu(y), g["y"]) < TOL
    y.backward(_t(g["dy"], dev))
    assert rel_l2(_cpu(x.grad), g["dx"]) < TOL
    _check_grads(blk, g)


def test_incremental_modes_fused_model_equals_small_model(dev):
    """FNO2d(incremental_n_modes) on the fused engine path == an FNO2d built with the smaller n_modes whose weights
    are the leading slices (spectral_convolution.py:270-298), gradients scattered back into the full weights."""
    from pde_policylearning_amd.neuralop.models import FNO2d
    torch.manual_seed(3)
    big = FNO2d(12, 12, 32, incremental_n_modes=(8, 8)).to(dev)
    small = FNO2d(8, 8, 32).to(dev)
    sd = big.state_dict()
    for k, v in small.state_dict().items():
        src = sd[k]
        v.copy_(src[:, :, :4, :4] if "convs.weight" in k else src)
    x = torch.randn(4, 3, 64, 64, device=dev)
    yb, ys = big(x), small(x)
    assert rel_l2(_cpu(yb), _cpu(ys)) < 1e-6
    yb.square().sum().backward()
    ys.square().sum().backward()
    gb, gs = dict(big.named_parameters()), dict(small.named_parameters())
    for k in gs:
        a = gb[k].grad
        if "convs.weight" in k:
            assert float(a[:, :, 4:].abs().max()) == 0 and float(a[:, :, :, 4:].abs().max()) == 0, k
            a = a[:, :, :4, :4]
        assert rel_l2(_cpu(a), _cpu(gs[k].grad)) < 1e-6, k


@pytest.mark.parametrize("case,kw", [
    ("regressor3d_small", dict(in_dim=5, n_hidden=5, freq_dim=6, out_dim=2, modes=3, spacial_dim=3)),
    ("regressor3d_w32", dict(in_dim=32, n_hidden=32, freq_dim=32, out_dim=1, modes=6, spacial_dim=3, activation='relu'))])
def test_spectral_regressor3d_golden(dev, case, kw):
    from pde_policylearning_amd.neuralop.models import SpectralRegressor
    g = load_golden(case)
    cn = {k for k in g["shapes"] if "weights" in k}
    model = _load(SpectralRegressor(**kw).eval(), g, dev, complex_names=cn)
    x = _t(g["x"], dev, True)
    y = model(x)
    assert rel_l2(_cpu(y), g["y"]) < TOL
    y.backward(_t(g["dy"], dev))
    assert rel_l2(_cpu(x.grad), g["dx"]) < TOL
    _check_grads(model, g)


def test_rno2d_predict_golden(dev):
    """RNO2d.predict (rno.py:370-379): three autoregressive steps, hidden states carried, loss on the stacked roll-out."""
    from pde_policylearning_amd.neuralop.models import RNO2d
    g = load_golden("rno2d_predict")
    model = _load(RNO2d(4, 4, 8, 0, layer_num=2).eval(), g, dev)
    y = model.predict(_t(g["x"], dev), num_steps=3)
    assert y.shape == g["y"].shape
    assert rel_l2(_cpu(y), g["y"]) < TOL
    loss = O.lp_loss_rel_sum(y, _t(g["target"], dev))
    assert abs(float(loss) / float(g["loss"][0]) - 1) < TOL
    loss.backward()
    _check_grads(model, g)


def test_train_observer_yaml_rno_sequences_vs_oracle(dev, tmp_path, monkeypatch):
    """`train_observer --train_yaml` on a reference-style RNO YAML (minchan_rno.yaml's keys: `timestep: 2`, recurrent_model,
    recurrent_index, width given twice) over a SequentialPDEDataset folder: two epochs of two steps of RNO2dObserver at
    model_timestep 2 (run_pde_observers.py:75-82, 170-193: sequences (B, T, X, Y, 1), scored on time step recurrent_index)
    against the CPU oracle (observers_oracle.rno2d_forward + torch.optim.Adam) on the same files, initial weights and batch
    order.  Dropout is switched off on both sides (the oracle restates eval mode)."""
    import yaml
    from oracle import observers_oracle as OO
    from pde_policylearning_amd import train_observer
    from pde_policylearning_amd.libs.models.rno_models import RNO2dObserver
    rng = np.random.default_rng(7)
    n, S, T, B = 20, 32, 2, 4
    xs = np.linspace(0, 2 * np.pi, S, endpoint=False)
    p = (rng.standard_normal((n, 1, 1)) * np.sin(xs)[None, :, None] + rng.standard_normal((n, 1, 1)) * np.cos(2 * xs)[None, None, :]
         + 0.1 * rng.standard_normal((n, S, S))).astype(np.float32)
    v = (0.7 * np.roll(p, 3, axis=1) - 0.2 * p + 0.5).astype(np.float32)
    for i in range(n):
        np.save(tmp_path / f"P_planes_{i:06d}.npy", p[i])
        np.save(tmp_path / f"V_planes_{i:06d}.npy", v[i])
    meta = {"P_planes": dict(mean=p.mean(0), std=p.std(0)), "V_planes": dict(mean=v.mean(0), std=v.std(0))}
    np.save(tmp_path / "metadata.npy", meta, allow_pickle=True)
    cfg = tmp_path / "rno.yaml"
    cfg.write_text("DATA_FOLDER: './data/somewhere_else'\nntrain: 16\nntest: 4\nmodel_name: RNO2dObserver\nlearning_rate: 0.001\n"
                   "weight_decay: 0.0001\nmodes: 12\nwidth: 32\ndownsample_rate: 1\nx_range: 32\ny_range: 32\nuse_patch: false\n"
                   "timestep: 2\nrecurrent_model: true\nrecurrent_index: 1\nrandom_split: false\nwidth: 64\nbatch_size: 4\n"
                   "layer_num: 1\nclose_wandb: true\nepochs: 2\n")

    def no_dropout(*a, **k):
        m = RNO2dObserver(*a, **k)
        for mod in m.modules():
            if isinstance(mod, torch.nn.Dropout):
                mod.p = 0.0
        return m
    monkeypatch.setattr(train_observer, "RNO2dObserver", no_dropout)
    args = train_observer.plan_from_yaml(train_observer.build_parser().parse_args(
        ["--train_yaml", str(cfg), "--data-folder", str(tmp_path), "--no-shuffle"]))
    assert (args.width, args.model_timestep, args.recurrent_index, args.dataset) == (64, 2, 1, "SequentialPDEDataset")
    hist = train_observer.run(args, log=lambda *_: None)
    torch.set_num_threads(min(torch.get_num_threads(), 16))      # (torch's CPU FFT / einsum do not scale past that on a many-core host)

    # ---- the oracle's trajectory --------------------------------------------------------------------------------------
    torch.manual_seed(args.seed)
    ref = RNO2dObserver(12, 12, 64, recurrent_index=1, layer_num=1)
    prm = {k: v.detach().clone().requires_grad_(True) for k, v in ref.state_dict().items()}
    opt = torch.optim.Adam(list(prm.values()), lr=1e-3, weight_decay=1e-4)
    eps = 1e-5
    enc = lambda a, s: (torch.from_numpy(a) - torch.from_numpy(s["mean"])) / (torch.from_numpy(s["std"]) + eps)
    pn, vn = enc(p, meta["P_planes"]), enc(v, meta["V_planes"])
    dec = lambda t: t * (torch.from_numpy(meta["V_planes"]["std"]) + eps) + torch.from_numpy(meta["V_planes"]["mean"])
    seq = lambda a, lo, hi: a[lo:hi].reshape(-1, T, S, S)

    def loss_of(lo, hi):
        x, tgt = seq(pn, lo, hi).unsqueeze(-1), seq(vn, lo, hi)[:, 1]
        y = OO.rno2d_forward(prm, x, 12, 12, 64, 1, 1).reshape(tgt.shape)
        return O.lp_loss_rel_sum(dec(y), dec(tgt))
    for ep in range(2):
        tot = 0.0
        for lo in (0, 8):                          # 16 training frames = 8 sequences = two batches of 4
            opt.zero_grad()
            l = loss_of(lo, lo + 8)
            l.backward()
            opt.step()
            tot += float(l)
        with torch.no_grad():
            test = float(loss_of(16, 20))
        assert abs(hist[ep]["train_l2"] - tot / 8) < 2e-4 * (tot / 8), (ep, hist[ep], tot / 8)
        assert abs(hist[ep]["test_l2"] - test / 2) < 2e-4 * (test / 2), (ep, hist[ep], test / 2)


def test_backward_survives_relocated_saved_buffer(dev):
    """The forward pass records what it left in its `saved` buffer (u_0 skipped by the fused lifting, magnitude bounds published)
    under the buffer's ADDRESS (csrc/fno_abi.hip FnoModelPlan::CallState).  autograd may hand the buffer back at another address
    (saved_tensors_hooks: offload, checkpoint repack): the backward must then still know that u_0 was never written - block 0 would
    read uninitialised memory otherwise - and fall back to the kernels that need no published bounds.  Every gradient
    against the undisturbed pass; config 2's shape, where the lifting IS fused and the two-term GEMMs ARE on."""
    from oracle.detfill import fill_named
    from pde_policylearning_amd.neuralop.models import FNO2d
    torch.manual_seed(0)
    model = FNO2d(12, 12, 64, in_channels=3, out_channels=1).to(dev)
    x = torch.from_numpy(fill_named("reloc.x", (8, 3, 128, 128), 1.0)).to(dev)
    tgt = torch.from_numpy(fill_named("reloc.t", (8, 1, 128, 128), 1.0)).to(dev)

    def grads(relocate):
        model.zero_grad(set_to_none=True)
        if relocate:
            with torch.autograd.graph.saved_tensors_hooks(lambda t: t.clone(), lambda t: t):
                y = model(x)
        else:
            y = model(x)
        O.lp_loss_rel_sum(y, tgt).backward()
        return y.detach(), {n: p.grad.detach().clone() for n, p in model.named_parameters()}
    y0, g0 = grads(False)
    y1, g1 = grads(True)
    assert torch.equal(y0, y1)
    for n in g0:
        a, b = g1[n].double(), g0[n].double()
        assert torch.isfinite(a).all() and float((a - b).norm() / b.norm()) < 5e-6, n


def test_second_backward_on_one_forward_resets_the_bounds(dev):
    """A forward pass clears the 64 magnitude-bound slots behind its `saved` buffer and the FIRST backward pass on that buffer
    skips its own fill of slots [32, 64) (round 5: one launch less per step, csrc/fno_abi.hip CallState::bwd_clean).  A second
    backward on the same forward (retain_graph) must clear them again: its dy is 1e-6 of the first one's here, and with the first
    pass's bounds left in place the two-term fp16 operands of the whole gradient chain would be scaled a million times too small
    (second terms in the fp16 subnormals).  The step is linear in dy: every gradient of the second pass against 1e-6 times the
    first pass's, at config 2's shape (two-term GEMMs on, lifting fused)."""
    from oracle.detfill import fill_named
    from pde_policylearning_amd.neuralop.models import FNO2d
    torch.manual_seed(0)
    model = FNO2d(12, 12, 64, in_channels=3, out_channels=1).to(dev)
    x = torch.from_numpy(fill_named("twice.x", (8, 3, 128, 128), 1.0)).to(dev)
    dy = torch.from_numpy(fill_named("twice.dy", (8, 1, 128, 128), 1.0)).to(dev)
    y = model(x)
    y.backward(dy, retain_graph=True)
    g1 = {n: p.grad.detach().clone() for n, p in model.named_parameters()}
    model.zero_grad(set_to_none=True)
    y.backward(1e-6 * dy)
    for n, p in model.named_parameters():
        a, b = p.grad.double(), 1e-6 * g1[n].double()
        assert torch.isfinite(a).all() and float((a - b).norm() / b.norm()) < 1e-5, (n, float((a - b).norm() / b.norm()))


def test_projection_weight_at_an_unaligned_address(dev):
    """Parameters may be views into a flat bucket (trainer.FusedAdam): W1 of the projection then starts at an address that is
    not a multiple of 16 bytes.  The backward's prologue (k_absmax3_pack_w1: the scan of dy / W1 / w2 and the two-term split of
    W1 in one launch, round 5) reads W1 with 16-byte loads when it can and element-wise when it cannot - same bound, same
    fragments: every gradient bitwise equal to the aligned case."""
    from oracle.detfill import fill_named
    from pde_policylearning_amd.neuralop.models import FNO2d
    torch.manual_seed(0)
    model = FNO2d(12, 12, 64, in_channels=3, out_channels=1).to(dev)
    x = torch.from_numpy(fill_named("unal.x", (8, 3, 128, 128), 1.0)).to(dev)
    tgt = torch.from_numpy(fill_named("unal.t", (8, 1, 128, 128), 1.0)).to(dev)

    def grads():
        model.zero_grad(set_to_none=True)
        O.lp_loss_rel_sum(model(x), tgt).backward()
        return {n: p.grad.detach().clone() for n, p in model.named_parameters()}
    g0 = grads()
    w = model.projection.fc1.weight
    assert w.data_ptr() % 16 == 0
    flat = torch.empty(w.numel() + 1, dtype=w.dtype, device=dev)
    flat[1:].copy_(w.detach().reshape(-1))
    w.data = flat[1:].view_as(w)                 # 4 bytes past a 16-byte boundary
    assert w.data_ptr() % 16 == 4
    g1 = grads()
    for n in g0:
        assert torch.equal(g0[n], g1[n]), n


def test_corner_weight_of_the_wrong_extent_is_an_error_not_a_fault(dev):
    """The C ABI takes raw pointers, so a corner weight smaller than the plan's kept modes would be read out of bounds on the
    device (round 5: a tool did exactly that and took a GPU memory fault).  The operator layer refuses it, as the reference's
    einsum does (neuralop/models/spectral_convolution.py:15-36)."""
    from pde_policylearning_amd import functional as F
    C, modes = 32, (8, 8)
    x = torch.randn(2, C, 64, 64, device=dev)
    skip = [0.1 * torch.randn(C, C, 1, device=dev)]
    bias = 0.1 * torch.randn(1, C, device=dev)
    good = [0.05 * torch.randn((C, C) + modes + (2,), device=dev) for _ in range(2)]
    small = [w[:, :, :4, :4].contiguous() for w in good]
    assert torch.isfinite(F.fno_blocks(x, skip, good, bias, modes, "ortho")).all()
    with pytest.raises(RuntimeError, match="spectral weight"):
        F.fno_blocks(x, skip, small, bias, modes, "ortho")


def test_split2_low_mixed_precision_fma_is_bit_identical(tmp_path):
    """fno_dev.h::split2_low forms the low term of every two-term fp16 split with v_fma_mixlo_f16 / v_fma_mixhi_f16 (hipcc does
    not select them by itself).  tools/mix_split_test.hip compares it bit for bit with the compiler's form (v_cvt_f32_f16,
    v_sub_f32, v_cvt_pk_f16_f32) on 2^23 values over the fp16 denormal .. overflow range and on 2.6e9 pairs in tester waves whose
    SIMD partners issue MFMAs back to back (where the packed-fp32 op_sel forms of round 4 misread); exit code = mismatches."""
    import os
    import shutil
    import subprocess
    from pde_policylearning_amd import build
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    hipcc = build.hipcc_path()
    assert hipcc and (os.path.exists(hipcc) or shutil.which(hipcc)), "hipcc is part of the image"
    exe = str(tmp_path / "mix_split_test.bin")
    r = subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fno-slp-vectorize", "-Wno-unused-value",
                        "-Wno-unused-result", "-o", exe, os.path.join(root, "tools", "mix_split_test.hip")],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, (r.stdout, r.stderr)
    assert "alone: 4194304 pairs, 0 different" in r.stdout and ", 0 different" in r.stdout.splitlines()[-1], r.stdout
