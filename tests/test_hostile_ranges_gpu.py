"""Hostile dynamic range.  Every other parity input is a uniform [-1, 1) hash fill (oracle/detfill.py); the two-term fp16 GEMMs
(fno_dev.h "h2") keep full relative precision only for elements above 2^-16 of the PUBLISHED maximum of their tensor, and the
chain that publishes those maxima (absmax_publish, the gradient-bound chain of the backward pass) is exercised here where it
matters: one channel 10^4 times the others, heavy-tailed (log-normal) fields, an all-zero sample, a tensor whose maximum
sits in a single element, one sample whose target norm is 1e-6 (its LpLoss gradient is 1e6 x its batch mates': the outlier dy
drives the gradient-bound chain of the whole backward pass).  Sizes at which every kernel runs its two-term variant (>= 1024 tiles).  Reference: the oracle in
float64 on float64 copies of the same float32 numbers; tolerance 1e-5 relative L2 on the output (BASELINE.json north_star),
gradients within the budget of tests/test_parity_gpu.py (1e-5, or BUDGET_SLACK x the float32 oracle's own distance from
float64 where that is larger).  Outputs are held to plain 1e-5 wherever the float32 oracle itself is under 5e-6; every case
appends its achieved numbers (engine / float32 oracle against float64, per tensor) to gpurun_out/hostile_errors.txt - the
committed table is profiles/r06_hostile_errors.txt (round 5: r05_hostile_errors.txt)."""
import os
import numpy as np
import pytest
import torch

from oracle import fno_oracle as O
from oracle.detfill import fill_named
from tests.test_parity_gpu import _fno_params, _run_fused, _within_budget
from tests.util import rel_l2

pytestmark = pytest.mark.gpu
TOL_Y = 1e-5
_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _record(test, case, name, e, e32):
    """one table row per (case, tensor): engine and float32-oracle distance from the float64 value"""
    try:
        os.makedirs(os.path.join(_ROOT, "gpurun_out"), exist_ok=True)
        with open(os.path.join(_ROOT, "gpurun_out", "hostile_errors.txt"), "a") as f:
            f.write(f"{test:12s} {case:22s} {name:40s} engine {e:.3e}   float32 oracle {e32:.3e}   ratio {e / max(e32, 1e-30):6.2f}\n")
    except OSError:
        pass


def _check_output(case, ey, ey32):
    # plain 1e-5 wherever the float32 evaluation of the reference is itself comfortably inside it; otherwise twice its distance
    assert np.isfinite(ey) and ey < (TOL_Y if ey32 < 5e-6 else max(TOL_Y, 2.0 * ey32)), (case, ey, ey32)


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    from pde_policylearning_amd import _lib
    _lib.lib()
    return torch.device("cuda:0")


def _hostile(case, B, S, C, L, half):
    p = _fno_params(C, L, half)
    x = torch.from_numpy(fill_named("hx", (B, 3, S, S), 1.0))
    tgt = torch.from_numpy(fill_named("ht", (B, 1, S, S), 1.0))
    if case == "hidden_channel_x1e4":
        # ONE hidden channel 10^4 times the others after the lifting; block 0 reads it back with 10^-4 weights, so the sums
        # stay O(1) while the operand tile's maximum is 10^4 x its typical element
        c = 7
        p["lifting.fc.weight"][c] *= 1e4
        p["lifting.fc.bias"][c] *= 1e4
        p["fno_blocks.fno_skips.0.weight"][:, c] *= 1e-4
        for i in (0, 1):
            p[f"fno_blocks.convs.weight.{i}.tensor"][c] *= 1e-4
    elif case == "input_channel_x1e4":
        x[:, 1] *= 1e4
        p["lifting.fc.weight"][:, 1] *= 1e-4
    elif case == "lognormal":
        g = torch.Generator().manual_seed(11)
        x = torch.exp(2.5 * torch.randn(x.shape, generator=g)) * torch.sign(torch.randn(x.shape, generator=g))
        x = (x / x.abs().mean()).float()
    elif case == "zero_sample":
        x[1] = 0.0
    elif case == "single_spike":
        x *= 1e-3
        x[2, 0, 17, 93] = 1e3
    elif case == "tiny":
        x *= 1e-6
    elif case == "target_norm_1e-6":
        tgt[3] *= 1e-6          # dL/dy of sample 3 is 1e6 x the other samples' (LpLoss divides by the target's norm)
    else:
        raise ValueError(case)
    return p, x, tgt


@pytest.mark.parametrize("case", ["hidden_channel_x1e4", "input_channel_x1e4", "lognormal", "zero_sample", "single_spike", "tiny",
                                  "target_norm_1e-6"])
def test_fno_model_hostile_dynamic_range(dev, case):
    B, S, C, L, modes = 8, 128, 64, 4, (12, 12)
    p, x, tgt = _hostile(case, B, S, C, L, [m // 2 for m in modes])
    torch.set_num_threads(min(torch.get_num_threads(), 16))
    p64 = {k: v.double().clone().requires_grad_(True) for k, v in p.items()}
    y64 = O.fno_forward(p64, x.double(), modes, n_layers=L)
    O.lp_loss_rel_sum(y64, tgt.double()).backward()
    p32 = {k: v.clone().requires_grad_(True) for k, v in p.items()}
    y32 = O.fno_forward(p32, x, modes, n_layers=L)
    O.lp_loss_rel_sum(y32, tgt).backward()
    y, pg = _run_fused(p, x, modes, dev, n_layers=L)
    ey, ey32 = rel_l2(y.detach().cpu().numpy(), y64.detach().numpy()), rel_l2(y32.detach().numpy(), y64.detach().numpy())
    _record("fno_model", case, "y", ey, ey32)
    _check_output(case, ey, ey32)
    O.lp_loss_rel_sum(y, tgt.to(dev)).backward()
    torch.cuda.synchronize()
    errs = {k: (rel_l2(pg[k].grad.cpu().numpy(), p64[k].grad.numpy()), rel_l2(p32[k].grad.numpy(), p64[k].grad.numpy())) for k in p}
    for k, (e, e32) in errs.items():
        _record("fno_model", case, k, e, e32)
    for k, (e, e32) in errs.items():
        _within_budget(e, e32, (case, k))


@pytest.mark.parametrize("case", ["channel_x1e4", "lognormal", "zero_sample", "single_spike", "dy_outlier_sample"])
def test_projection_head_hostile_dynamic_range(dev, case):
    from pde_policylearning_amd import functional as F
    C, hid, shape = 64, 256, (8, 64, 128, 128)
    x = torch.from_numpy(fill_named("hpx", shape, 1.0))
    w1 = torch.from_numpy(fill_named("hpw1", (hid, C), 0.15))
    b1 = torch.from_numpy(fill_named("hpb1", (hid,), 0.1))
    w2 = torch.from_numpy(fill_named("hpw2", (1, hid), 0.1))
    b2 = torch.from_numpy(fill_named("hpb2", (1,), 0.1))
    dy = torch.from_numpy(fill_named("hpd", (shape[0], 1) + shape[2:], 1.0))
    if case == "channel_x1e4":
        x[:, 5] *= 1e4
        w1[:, 5] *= 1e-4
    elif case == "lognormal":
        g = torch.Generator().manual_seed(12)
        x = (torch.exp(2.5 * torch.randn(shape, generator=g)) * torch.sign(torch.randn(shape, generator=g))).float()
        x = x / x.abs().mean()
        w1 *= 0.05
    elif case == "zero_sample":
        x[3] = 0.0
        dy[5] = 0.0
    elif case == "dy_outlier_sample":
        dy[2] *= 1e6            # one sample's output gradient 1e6 x the others' (what a target of norm 1e-6 does to LpLoss)
    else:
        x *= 1e-3
        x[1, 9, 100, 3] = 1e3
    def ref(dtype):
        t = [v.to(dtype).clone().requires_grad_(True) for v in (x, w1, b1, w2, b2)]
        yr = (torch.nn.functional.gelu(t[0].movedim(1, -1) @ t[1].t() + t[2]) @ t[3].t() + t[4]).movedim(-1, 1)
        yr.backward(dy.to(dtype))
        return yr.detach().numpy(), [v.grad.numpy() for v in t]
    torch.set_num_threads(min(torch.get_num_threads(), 16))
    y64, g64 = ref(torch.float64)
    y32, g32 = ref(torch.float32)
    eng = [t.to(dev).requires_grad_(True) for t in (x, w1, b1, w2, b2)]
    ye = F.projection_head(*eng, act="gelu")
    ey, ey32 = rel_l2(ye.detach().cpu().numpy(), y64), rel_l2(y32, y64)
    _record("projection", case, "y", ey, ey32)
    _check_output(case, ey, ey32)
    ye.backward(dy.to(dev))
    errs = [(name, rel_l2(a.grad.cpu().numpy(), r64), rel_l2(r32, r64)) for a, r64, r32, name in zip(eng, g64, g32, ("x", "w1", "b1", "w2", "b2"))]
    for name, e, e32 in errs:
        _record("projection", case, name, e, e32)
    for name, e, e32 in errs:
        _within_budget(e, e32, (case, name))


@pytest.mark.parametrize("B", [9, 17])
def test_uneven_tile_shares_cover_every_tile(dev, B):
    """Two workgroups per CU split a CU's tiles unevenly (pair_share, fno_dev.h: the block forward's tiles, the projection
    forward's pixel columns popped from an LDS counter).  Batch sizes whose tile count is not a multiple of the grid
    (9 x 128 = 1152 and 17 x 128 = 2176 tiles on 512 workgroups: pairs own 4-5 or 8-9 tiles, split 2/2, 3/2, 4/4, 5/4) must
    still write every pixel exactly once: output and every gradient against float64."""
    S, C, L, modes = 128, 64, 4, (12, 12)
    p = _fno_params(C, L, [m // 2 for m in modes])
    x = torch.from_numpy(fill_named(f"ux{B}", (B, 3, S, S), 1.0))
    tgt = torch.from_numpy(fill_named(f"ut{B}", (B, 1, S, S), 1.0))
    torch.set_num_threads(min(torch.get_num_threads(), 16))
    p64 = {k: v.double().clone().requires_grad_(True) for k, v in p.items()}
    y64 = O.fno_forward(p64, x.double(), modes, n_layers=L)
    O.lp_loss_rel_sum(y64, tgt.double()).backward()
    p32 = {k: v.clone().requires_grad_(True) for k, v in p.items()}
    O.lp_loss_rel_sum(O.fno_forward(p32, x, modes, n_layers=L), tgt).backward()
    y, pg = _run_fused(p, x, modes, dev, n_layers=L)
    assert rel_l2(y.detach().cpu().numpy(), y64.detach().numpy()) < TOL_Y
    O.lp_loss_rel_sum(y, tgt.to(dev)).backward()
    torch.cuda.synchronize()
    for k in p:
        g64 = p64[k].grad.numpy()
        _within_budget(rel_l2(pg[k].grad.cpu().numpy(), g64), rel_l2(p32[k].grad.numpy(), g64), (B, k))
