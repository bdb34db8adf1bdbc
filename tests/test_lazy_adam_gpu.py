"""FusedAdam's dead-slice plan (trainer.FusedAdam.skip_dead_slices) against stepping everything and against
torch.optim.Adam (run_pde_observers.py:134: Adam(lr, weight_decay)): a dialect-C 3-D spectral weight only sees data in
[..., :min(Nz/2+1, modes3)] (libs/models/pino_models/basics.py:119-139); the rest is not stepped but replayed.  The replay
must leave exactly what stepping leaves - bit for bit against the engine's own full step (same operations in the same
order), to float32 rounding against torch.optim.Adam - through learning-rate changes, checkpoints in the middle, a longer
last dimension in the middle, and a state_dict round trip."""
import copy

import numpy as np
import pytest
import torch
from torch import nn

from tests.util import rel_l2

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    from pde_policylearning_amd import _lib
    _lib.lib()
    return torch.device("cuda:0")


class Tiny(nn.Module):
    """lift -> dialect-C spectral convolution -> project, on (B, X, Y, T, 3) fields"""

    def __init__(self, modes3=6, planes=True):
        super().__init__()
        from pde_policylearning_amd.libs.models.pino_models.basics import SpectralConv3d
        self.fc0 = nn.Linear(3, 8)
        self.conv = SpectralConv3d(8, 8, 4, 4, modes3)
        self.conv2 = SpectralConv3d(8, 8, 4, 4, modes3)
        self.fc1 = nn.Linear(8, 1)
        if not planes:      # the reference's own memory order (e.g. parameters swapped in from elsewhere): row-sliced plan
            for c in (self.conv, self.conv2):
                for i in range(1, 5):
                    setattr(c, f"weights{i}", nn.Parameter(getattr(c, f"weights{i}").detach().contiguous()))

    def forward(self, x):
        h = self.fc0(x).permute(0, 4, 1, 2, 3).contiguous()
        h = torch.tanh(self.conv(h))
        h = self.conv2(h)
        return self.fc1(h.permute(0, 2, 3, 4, 1))


def _real(t):
    return torch.view_as_real(t) if t.is_complex() else t


def _same_bits(a, b):
    return torch.equal(_real(a).contiguous().view(torch.int32), _real(b).contiguous().view(torch.int32))


def _batch(step, T, dev):
    g = torch.Generator().manual_seed(1000 + step)
    return torch.randn(2, 16, 16, T, 3, generator=g).to(dev), torch.randn(2, 16, 16, T, 1, generator=g).to(dev)


def _loss(model, step, T, dev):
    x, t = _batch(step, T, dev)
    return ((model(x) - t) ** 2).sum()


@pytest.mark.parametrize("planes", [True, False])
@pytest.mark.parametrize("capturable", [False, True])
@pytest.mark.parametrize("wd", [1e-4, 0.0])
def test_dead_slices_replayed_equal_stepped(dev, capturable, wd, planes):
    from pde_policylearning_amd import functional as F
    from pde_policylearning_amd.trainer import FlatGradBucket, FusedAdam
    torch.manual_seed(5)
    base = Tiny(planes=planes).to(dev)
    assert F.plane_major(base.conv.weights1) == planes
    lazy_m, full_m, ref_m = copy.deepcopy(base), copy.deepcopy(base), copy.deepcopy(base)
    opts = {}
    for name, m, skip in (("lazy", lazy_m, True), ("full", full_m, False)):
        bucket = FlatGradBucket(m.parameters(), direct_module=m)
        opts[name] = (bucket, FusedAdam(bucket, lr=1e-3, weight_decay=wd, capturable=capturable, skip_dead_slices=skip))
    ref_opt = torch.optim.Adam(ref_m.parameters(), lr=1e-3, weight_decay=wd)
    T = 2                                  # k3 = min(T/2+1, 6) = 2: four of six last-dim slices are dead

    def one(step):
        for name, m in (("lazy", lazy_m), ("full", full_m)):
            bucket, opt = opts[name]
            bucket.zero()
            _loss(m, step, T, dev).backward()
            opt.step()
        ref_opt.zero_grad()
        _loss(ref_m, step, T, dev).backward()
        ref_opt.step()

    def check(tag, bits=True):
        sl, sf, sr = lazy_m.state_dict(), full_m.state_dict(), ref_m.state_dict()       # (the pre-hook replays)
        for k in sl:
            if bits:
                assert _same_bits(sl[k], sf[k]), (tag, k)
            # (two float32 Adam implementations drift apart over 55 steps by more than one step's rounding - the slices
            # that come alive at step 50 start from moments of ~1e-12; the engine's step itself is held to torch.optim.Adam
            # at 1e-6 in test_parity_gpu.test_fused_adam_matches_torch_adam, the replay to the engine's step bit for bit)
            assert rel_l2(_real(sl[k]).cpu().numpy(), _real(sr[k]).cpu().numpy()) < 5e-5, (tag, k)

    for step in range(20):
        one(step)
    assert opts["lazy"][1]._runs is not None and any(r[0] == "rows" for r in opts["lazy"][1]._runs)
    assert opts["lazy"][1].exp_avg.numel() < opts["full"][1].exp_avg.numel()
    # the dead slice has not been stepped yet: with weight decay it lags, the replay (state_dict below) brings it level
    if wd:
        assert not _same_bits(lazy_m.conv.weights1.data[..., 2:], full_m.conv.weights1.data[..., 2:])
    assert _same_bits(lazy_m.conv.weights1.data[..., :2], full_m.conv.weights1.data[..., :2])
    for _, o in opts.values():             # a scheduler's milestone (trainer.MultiStepLR sets optimizer.lr)
        o.lr = 5e-4
    ref_opt.param_groups[0]["lr"] = 5e-4
    for step in range(20, 30):
        one(step)
    check("after the lr change")
    so_l, so_f = opts["lazy"][1].state_dict(), opts["full"][1].state_dict()
    for i in so_f["state"]:
        for key in ("exp_avg", "exp_avg_sq"):
            assert _same_bits(so_l["state"][i][key], so_f["state"][i][key]), (i, key)
        assert float(so_l["state"][i]["step"]) == float(so_f["state"][i]["step"]) == 30.0
    for step in range(30, 40):
        one(step)
    # a longer last dimension (validation at T = 8: k3 = 5) reads slices 2..4: they must be current BEFORE the read
    with torch.no_grad():
        xl, _ = _batch(777, 8, dev)
        yl, yf = lazy_m(xl), full_m(xl)
    assert _same_bits(yl, yf)
    run0 = opts["lazy"][1]._runs[0]                    # the re-planned block: five of six slices live
    assert run0[0] == "rows" and run0[4] * 6 == run0[3] * 5
    for step in range(40, 50):
        one(step)
    T = 8
    for step in range(50, 55):             # ... and then trains there
        one(step)
    check("after the longer last dimension")
    # state_dict round trip into fresh optimizers (lazy -> lazy, lazy -> torch.optim.Adam), then five more steps
    sd = opts["lazy"][1].state_dict()
    again_m = copy.deepcopy(lazy_m)
    bucket2 = FlatGradBucket(again_m.parameters(), direct_module=again_m)
    opt2 = FusedAdam(bucket2, lr=1.0, weight_decay=0.5, capturable=capturable, skip_dead_slices=True)
    opt2.load_state_dict(sd)
    for step in range(55, 60):
        one(step)
        bucket2.zero()
        _loss(again_m, step, T, dev).backward()
        opt2.step()
    check("at the end")
    s2, sl = again_m.state_dict(), lazy_m.state_dict()
    for k in sl:
        assert _same_bits(s2[k], sl[k]), k


def test_replay_many_steps_matches_torch_adam(dev):
    """200 skipped steps in one replay against torch.optim.Adam stepping a zero gradient 200 times."""
    from pde_policylearning_amd import functional as F
    torch.manual_seed(3)
    rows, row_len, live_len = 37, 12, 2
    p0 = (0.5 + torch.rand(rows, row_len, device=dev)) * torch.where(torch.rand(rows, row_len, device=dev) < 0.5, -1.0, 1.0)
    ref = torch.nn.Parameter(p0.clone())
    ref_opt = torch.optim.Adam([ref], lr=1e-3, weight_decay=1e-4)
    for _ in range(200):
        ref.grad = torch.zeros_like(ref)
        ref_opt.step()
    mine = p0.clone().view(-1)
    dm, dv = (torch.empty(rows * (row_len - live_len), device=dev) for _ in range(2))
    for on_device in (False, True):
        q = mine.clone()
        scal = F.adam_replay_scalars(1, 200, 1e-3, (0.9, 0.999), dev, on_device=on_device)
        F.adam_replay_dead(rows, row_len, live_len, q, dm, dv, True, scal, (0.9, 0.999), 1e-8, 1e-4)
        q = q.view(rows, row_len)
        assert torch.equal(q[:, :live_len], p0[:, :live_len])           # the live part is not the replay's business
        assert rel_l2(q[:, live_len:].cpu().numpy(), ref.data[:, live_len:].cpu().numpy()) < 1e-6
        st = ref_opt.state[ref]
        assert rel_l2(dm.view(rows, -1).cpu().numpy(), st["exp_avg"][:, live_len:].cpu().numpy()) < 1e-6
        assert rel_l2(dv.view(rows, -1).cpu().numpy(), st["exp_avg_sq"][:, live_len:].cpu().numpy()) < 1e-6


@pytest.mark.parametrize("cin,cout,m12,m3,T", [(8, 8, (4, 4), 6, 2), (5, 7, (3, 5), 4, 16), (16, 16, (8, 7), 3, 2)])
def test_plane_major_weights_equal_contiguous(dev, cin, cout, m12, m3, T):
    """The engine on plane-major corner weights (last dim outermost in memory: libs SpectralConv3d's own layout) against
    the same values in the reference's memory order: outputs and gradients bit for bit (the packed arrays the GEMMs read
    are the same), gradients returned in the weights' layout with zeros in the dead planes."""
    from pde_policylearning_amd import functional as F
    torch.manual_seed(2)
    k3 = min(T // 2 + 1, m3)
    ws = [torch.randn(cin, cout, *m12, m3, dtype=torch.cfloat, device=dev) for _ in range(4)]
    x = torch.randn(2, cin, 16, 16, T, device=dev)
    outs = []
    for planes in (False, True):
        wl = [(F.to_plane_major(w) if planes else w.clone()).requires_grad_(True) for w in ws]
        assert all(F.plane_major(w) == (planes and m3 > 1) for w in wl)
        xi = x.clone().requires_grad_(True)
        y = F.spectral_conv(xi, wl, None, (*m12, k3), "backward", weight_last_extent=m3)
        (y * torch.cos(y)).sum().backward()
        outs.append((y.detach(), xi.grad, [w.grad for w in wl]))
    (y0, dx0, g0), (y1, dx1, g1) = outs
    assert _same_bits(y0, y1) and _same_bits(dx0, dx1)
    for a, b in zip(g0, g1):
        assert b.stride() == F.to_plane_major(a).stride()
        assert _same_bits(a, b)
        assert float(b[..., k3:].abs().max()) == 0.0 if k3 < m3 else True
