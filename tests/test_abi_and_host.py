"""CPU-side checks (no GPU): the C-ABI library builds/loads and exports every symbol
include/fnoengine.h declares; the host modules mirror the reference surface; the product
path refuses to run without a GPU instead of silently falling back."""
import ctypes
import io
import os
import pickle
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    from pde_policylearning_amd import build, _lib
    build.build()
    return _lib.lib()


def test_header_symbols_exported(lib):
    hdr = open(os.path.join(ROOT, "include", "fnoengine.h")).read()
    declared = set(re.findall(r"\b(fno_[a-z0-9_]+)\s*\(", hdr))
    assert len(declared) >= 18
    from pde_policylearning_amd import _lib
    assert declared == set(_lib.EXPORTED_SYMBOLS)
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.fno_version() >= 100


def test_struct_layout_matches_header(lib):
    from pde_policylearning_amd import _lib
    # FnoSpecDesc: 3 + 3 + 3 + 4 ints ; FnoModelDesc: 6 + 3 + 3 + 3 ints
    assert ctypes.sizeof(_lib.FnoSpecDesc) == 13 * 4
    assert ctypes.sizeof(_lib.FnoModelDesc) == 15 * 4
    assert ctypes.sizeof(_lib.FnoModelParams) == 8 * (2 + 16 + 64 + 1 + 4)


def test_plan_rejects_bad_arguments_without_gpu(lib):
    from pde_policylearning_amd import _lib
    d = _lib.FnoSpecDesc()
    d.ndim, d.Cin, d.Cout = 4, 2, 2          # unsupported ndim: refused before any HIP call
    h = ctypes.c_void_p()
    rc = lib.fno_spec_plan_create(ctypes.byref(d), ctypes.byref(h))
    assert rc < 0 and b"ndim" in lib.fno_last_error()
    d.ndim = 2
    d.dims[0], d.dims[1] = 8, 8
    d.modes[0], d.modes[1] = 9, 3            # more kept rows than the grid has
    rc = lib.fno_spec_plan_create(ctypes.byref(d), ctypes.byref(h))
    assert rc < 0 and b"exceeds" in lib.fno_last_error()
    d.ndim = 3
    d.dims[0], d.dims[1], d.dims[2] = 8, 8, 8
    d.modes[0], d.modes[1], d.modes[2] = 5, 3, 3     # 2*5 > 8: overlapping corners are supported on 2-D grids only
    rc = lib.fno_spec_plan_create(ctypes.byref(d), ctypes.byref(h))
    assert rc < 0 and b"overlapping" in lib.fno_last_error()


def test_fno2d_surface_matches_reference():
    from pde_policylearning_amd.neuralop.models import FNO2d, FNO3d
    m = FNO2d(12, 12, 64, in_channels=3, out_channels=1)
    # parameter count of the reference model
    assert sum(p.numel() for p in m.parameters()) == 2393089
    keys = set(m.state_dict())
    assert {"lifting.fc.weight", "lifting.fc.bias", "fno_blocks.convs.bias",
            "fno_blocks.fno_skips.0.weight", "fno_blocks.convs.weight.7.tensor",
            "projection.fc1.weight", "projection.fc2.bias"} <= keys
    assert m.fno_blocks.convs.weight[0].tensor.shape == (64, 64, 6, 6, 2)   # n_modes // 2 per dim
    assert [m.fno_blocks.gelu_after(l) for l in range(4)] == [True, True, False, False]
    assert sum(p.numel() for p in FNO3d(8, 8, 8, 32).parameters()) == 2110209
    # torch.save(model) round trip (run_pde_observers.py:313-314)
    buf = io.BytesIO()
    pickle.dump(m, buf)
    m2 = pickle.loads(buf.getvalue())
    assert torch.equal(m2.lifting.fc.weight, m.lifting.fc.weight)


def test_whole_module_checkpoint_round_trip(tmp_path):
    """run_pde_observers.py:313-314 saves the best observer with torch.save(model) and run_control.py loads it back with
    torch.load: every observer class of the path must survive that (state, class identity, engine hooks), and
    train_observer.save_if_best must write only on improvement."""
    from pde_policylearning_amd.libs.models.fno_models import FNO2dObserver
    from pde_policylearning_amd.libs.models.pino_models import PINObserver2d, PINObserverFullField
    from pde_policylearning_amd.libs.models.rno_models import RNO2dObserver
    from pde_policylearning_amd.train_observer import save_if_best
    models = [FNO2dObserver(8, 8, 32), RNO2dObserver(4, 4, 32, 0, layer_num=1),
              PINObserver2d(modes1=[4] * 4, modes2=[4] * 4, modes3=[4] * 4, fc_dim=16, layers=[8] * 5, in_dim=4, out_dim=1,
                            act="gelu", pad_ratio=0.0625),
              PINObserverFullField(plane_num=3, modes1=[4] * 4, modes2=[4] * 4, modes3=[4] * 4, fc_dim=16, layers=[8] * 5,
                                   in_dim=1, out_dim=1, act="gelu", pad_ratio=[0.0, 0.0625])]
    logs = []
    for i, m in enumerate(models):
        path = str(tmp_path / "outputs" / f"m{i}.pth")
        assert save_if_best(m, 0.5, 1e10, path, 0, logs.append) == 0.5
        m2 = torch.load(path, weights_only=False)
        assert type(m2) is type(m)
        sd, sd2 = m.state_dict(), m2.state_dict()
        assert list(sd) == list(sd2) and all(torch.equal(sd[k], sd2[k]) for k in sd)
        os.remove(path)
        assert save_if_best(m, 0.7, 0.5, path, 0, logs.append) == 0.5 and not os.path.exists(path)      # no improvement
        assert save_if_best(m, 0.3, 0.5, path, 1, logs.append) == 0.3 and not os.path.exists(path)      # rank != 0 tracks only
    assert len(logs) == len(models)


def test_unsupported_configurations_fail_loudly():
    from pde_policylearning_amd.neuralop.models import FNO, SpectralConv
    with pytest.raises(NotImplementedError):
        SpectralConv(4, 4, (8, 8), factorization="tucker")
    with pytest.raises(NotImplementedError):
        FNO((8, 8), 32, use_mlp=True)
    with pytest.raises(NotImplementedError):
        FNO((8, 8), 32, norm="group_norm")


def test_reference_import_lines_resolve_against_the_package():
    """The reference's own import statements (libs/models/fno_models.py:9, libs/models/rno_models.py:8,
    neuralop/__init__.py:3-5, neuralop/models/__init__.py:1-7), executed verbatim with the package aliased as `neuralop`."""
    import importlib
    import sys
    import pde_policylearning_amd.neuralop as pkg
    saved = {k: sys.modules.get(k) for k in ("neuralop", "neuralop.models")}
    sys.modules["neuralop"], sys.modules["neuralop.models"] = pkg, pkg.models
    try:
        ns = {}
        for line in ("from neuralop.models import FNO2d", "from neuralop.models import RNO2d",
                     "from neuralop.models import TFNO3d, TFNO2d, TFNO1d, TFNO", "from neuralop.models import get_model",
                     "from neuralop.models import FNO, FNO1d, FNO2d, FNO3d", "from neuralop.models import SFNO",
                     "from neuralop.models import UNO", "from neuralop.models import SpectralRegressor",
                     "from neuralop import TFNO3d, TFNO2d, TFNO1d, TFNO", "from neuralop import RNO2d",
                     "from neuralop import get_model"):
            exec(line, ns)
        from pde_policylearning_amd.neuralop.models import rno, tfno
        assert ns["RNO2d"] is rno.RNO2d and ns["FNO2d"] is tfno.FNO2d
        # the classes outside the accelerated path keep name and signature and say so when constructed
        for cls, args in ((ns["TFNO2d"], (4, 4, 8)), (ns["TFNO"], ((4, 4), 8)), (ns["FNO1d"], (4, 8)),
                          (ns["SFNO"], ((4, 4), 8)), (ns["UNO"], (3, 1, 8))):
            with pytest.raises(NotImplementedError):
                cls(*args)
        with pytest.raises(ImportError, match="out of scope"):
            exec("from neuralop import Trainer", {})
        # get_model builds the accelerated classes from a reference-style config (model_dispatcher.py:26-62)
        cfg = {"arch": "FNO2d", "fno2d": dict(data_channels=3, n_modes_height=8, n_modes_width=8, hidden_channels=32),
               "patching": {"levels": 1}}
        m = ns["get_model"](cfg)
        assert type(m) is tfno.FNO2d and m.in_channels == 6
        with pytest.raises(ValueError):
            ns["get_model"]({"arch": "nope", "nope": {}})
    finally:
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v


def test_constructor_options_beside_the_accelerated_path():
    """separable / incremental_n_modes / output_scaling_factor build (torch compositions or sliced engine weights, GPU
    parity in tests/test_boundary_gpu.py); shapes follow spectral_convolution.py:243-268."""
    from pde_policylearning_amd.neuralop.models import FNO, SpectralConv
    c = SpectralConv(4, 4, (8, 6), separable=True, n_layers=2)
    assert tuple(c.weight[0].tensor.shape) == (4, 4, 3, 2) and len(c.weight) == 4
    with pytest.raises(ValueError):
        SpectralConv(4, 6, (8, 6), separable=True)
    c = SpectralConv(4, 6, (8, 8), incremental_n_modes=(4, 6))
    assert c.half_n_modes == [2, 3] and tuple(c.layer_weights(0)[0].shape) == (4, 6, 2, 3, 2)
    c.incremental_n_modes = None
    assert c.half_n_modes == [4, 4]
    m = FNO((8, 8), 16, output_scaling_factor=[2, 1, 1, 0.5])
    assert m.fno_blocks.convs.output_scaling_factor == [[2, 2], [1, 1], [1, 1], [0.5, 0.5]]
    with pytest.raises(RuntimeError, match="GPU"):       # still no CPU path
        c(torch.zeros(1, 4, 16, 16))


def test_no_cpu_fallback():
    from pde_policylearning_amd.neuralop.models import FNO2d
    m = FNO2d(8, 8, 32)
    with pytest.raises(RuntimeError, match="GPU"):
        m(torch.zeros(1, 3, 32, 32))


def test_observer_grid_matches_reference_definition():
    from pde_policylearning_amd.libs.models.fno_models import FNO2dObserver
    import numpy as np
    ob = FNO2dObserver(8, 8, 32)
    g = ob.get_grid((2, 5, 7, 1), torch.device("cpu"))
    assert g.shape == (2, 5, 7, 2)
    assert np.allclose(g[0, :, 0, 0].numpy(), np.linspace(0, 1, 5).astype(np.float32))
    assert np.allclose(g[1, 0, :, 1].numpy(), np.linspace(0, 1, 7).astype(np.float32))


def _write_plane_folder(tmp_path, g):
    import numpy as np
    planes = {"P_planes": g["p_raw"], "V_planes": g["v_raw"]}
    for k, v in planes.items():
        for i in range(v.shape[0]):
            np.save(tmp_path / f"{k}_{i:06d}.npy", v[i])
    np.save(tmp_path / "metadata.npy", {k: dict(mean=v.mean(0), std=v.std(0)) for k, v in planes.items()}, allow_pickle=True)


def test_plane_dataset_matches_reference_items(tmp_path):
    """libs.pde_data_loader.PDEDataset on the reference's on-disk format (per-timestep .npy planes + pickled metadata.npy)
    vs items produced by the reference class itself (tests/golden/pde_dataset.npz, oracle/make_golden.py::gen_pde_dataset)."""
    import types
    import numpy as np
    from tests.util import load_golden
    from pde_policylearning_amd.libs.pde_data_loader import PDEDataset, SequentialPDEDataset
    g = load_golden("pde_dataset")
    _write_plane_folder(tmp_path, g)
    args = types.SimpleNamespace(model_timestep=2)
    idx = [int(i) for i in g["data_index"]]
    ds = PDEDataset(args, str(tmp_path), idx, 2, 5, 4)
    assert len(ds) == 6
    for i in range(len(ds)):
        p, v = ds[i]
        assert p.shape == (5, 4, 1) and np.array_equal(p.numpy(), g["p_items"][i]) and np.array_equal(v.numpy(), g["v_items"][i])
    dsp = PDEDataset(args, str(tmp_path), idx[:3], 1, 6, 5, use_patch=True)
    for i in range(len(dsp)):
        p, v = dsp[i]
        assert np.array_equal(p.numpy(), g["p_patch"][i]) and np.array_equal(v.numpy(), g["v_patch"][i])
    seq = SequentialPDEDataset(args, str(tmp_path), idx, 2, 5, 4)
    assert len(seq) == 3
    ps, vs = seq[1]                                   # planes idx[2], idx[3]
    assert ps.shape == (2, 5, 4) and np.array_equal(ps[0].numpy(), g["p_items"][2][..., 0]) and np.array_equal(vs[1].numpy(), g["v_items"][3][..., 0])
    # decode(encode(x)) round trip of the normaliser
    raw = torch.tensor(g["p_raw"][idx[0]][::2, ::2][:5, :4])
    assert torch.allclose(ds.p_norm.decode(ds.p_norm.encode(raw)), raw, atol=1e-5)


def test_chanflow_host_side_without_gpu(lib):
    """fno_chanflow_pack_metrics is pure host code: check the packed reciprocal spacings against numpy, the argument
    checks of the device entry points (refused before any HIP call), and that CPU tensors are refused loudly."""
    import numpy as np
    from pde_policylearning_amd import _lib
    from pde_policylearning_amd.libs.envs.control_env import ChannelFlowRHS
    env = ChannelFlowRHS.tanh_channel(8, 10, 6)
    Ny, MP = 10, 12
    assert env.yg.shape == (Ny + 1,) and env.yg[0] == -env.ym[0] and env.yg[-1] == 2 + env.ym[0]      # control_env.py:165
    packed = np.zeros(3 * MP)
    dp = ctypes.POINTER(ctypes.c_double)
    assert lib.fno_chanflow_pack_metrics(Ny, env.y.ctypes.data_as(dp), env.ym.ctypes.data_as(dp), env.yg.ctypes.data_as(dp),
                                         packed.ctypes.data_as(dp)) == 0
    want = np.zeros(3 * MP)
    want[1:Ny] = 1 / np.diff(env.y)
    want[MP + 1:MP + Ny - 1] = 1 / np.diff(env.ym)
    want[2 * MP + 1:2 * MP + Ny + 1] = 1 / np.diff(env.yg)
    assert np.array_equal(packed, want)
    y_bad = env.y.copy()
    y_bad[3] = y_bad[2]
    assert lib.fno_chanflow_pack_metrics(Ny, y_bad.ctypes.data_as(dp), env.ym.ctypes.data_as(dp), env.yg.ctypes.data_as(dp),
                                         packed.ctypes.data_as(dp)) < 0
    assert b"repeated" in lib.fno_last_error()
    g = _lib.FnoChanflowGrid(8, 2, 6, 0.1, 0.1, 1e-3)                         # Ny < 3
    assert lib.fno_chanflow_rhs(ctypes.byref(g), 1, 0, packed.ctypes.data, None, None, None, None, 0.0, None, None, None, None) < 0
    assert ctypes.sizeof(_lib.FnoChanflowGrid) == 3 * 4 + 4 + 3 * 8           # 3 ints, padding, 3 doubles
    U, V, W = torch.zeros(8, 11, 6), torch.zeros(8, 10, 6), torch.zeros(8, 11, 6)
    with pytest.raises(RuntimeError, match="GPU"):
        env.compute_rhs_py(U, V, W)
    with pytest.raises(RuntimeError, match="GPU"):
        env.pde_loss(U, V, V, W)
    with pytest.raises(RuntimeError, match="expected U, W"):
        env.pde_loss(U, V, V, W[:, :-1])


def _write_fullfield_folder(d, g):
    """rebuild the folder the fixture was generated from out of the raw fields the reference's items carry"""
    import numpy as np
    idx = [int(i) for i in g["data_index"]]
    T = g["u"].shape[1]
    raw = {k: [None] * len(idx) for k in ("u", "v", "w")}
    dpdx = [0.0] * len(idx)
    for item in range(g["u"].shape[0]):
        for t in range(T):
            i = idx[item * T + t]
            for k in raw:
                raw[k][i] = g[k][item, t]
            dpdx[i] = float(g["dpdx"][item, t])
    f = {"U_field": np.stack(raw["u"]), "V_field": np.stack(raw["v"]), "W_field": np.stack(raw["w"])}
    meta = {k: dict(mean=v.mean(0), std=v.std(0)) for k, v in f.items()}
    meta["U_field"]["dpdx"] = dpdx
    meta["re"] = float(g["re"][0, 0])
    meta["P_planes"] = dict(mean=np.zeros(f["V_field"].shape[1::2], np.float32), std=np.ones(f["V_field"].shape[1::2], np.float32))
    for k, v in f.items():
        for i in range(v.shape[0]):
            np.save(os.path.join(d, f"{k}_{i:06d}.npy"), v[i])
    np.save(os.path.join(d, "metadata.npy"), meta, allow_pickle=True)
    return idx


def test_fullfield_dataset_matches_reference_items(tmp_path):
    """libs.pde_data_loader.FullFieldNSDataset vs items of the reference class (tests/golden/fullfield_dataset.npz,
    oracle/make_golden.py::gen_fullfield_dataset): wall plane + target planes normalised by the wall-plane statistics,
    raw U / V / W, Re and dPdx per timestep."""
    import types
    import numpy as np
    from tests.util import load_golden
    from pde_policylearning_amd.libs.pde_data_loader import FullFieldNSDataset
    g = load_golden("fullfield_dataset")
    idx = _write_fullfield_folder(str(tmp_path), g)
    ds = FullFieldNSDataset(types.SimpleNamespace(model_timestep=2), str(tmp_path), idx, [int(p) for p in g["plane_indexs"]], 1, 6, 5)
    assert len(ds) == g["u"].shape[0]
    for i in range(len(ds)):
        item = ds[i]
        for got, name in zip(item, ["v_plane", "v_field", "u", "v", "w", "re", "dpdx"]):
            want = g[name][i]
            assert tuple(got.shape) == want.shape, name
            assert np.allclose(got.numpy(), want, rtol=0, atol=0 if name in ("u", "v", "w") else 1e-6), name


def test_kf_dataset_matches_reference_items(tmp_path):
    """libs.pino_utils.datasets.MultipleReynoldsKFaDataset vs items of the reference class (tests/golden/kf_dataset.npz):
    windowing by t_duration, spatial subsampling, the (x, y, t, u0) input grid, the per-window Reynolds number."""
    import numpy as np
    from tests.util import load_golden
    from pde_policylearning_amd.libs.pino_utils.datasets import MultipleReynoldsKFaDataset, sample_data
    g = load_golden("kf_dataset")
    path = os.path.join(tmp_path, "multi_reynolds_tiny.npz")
    np.savez(path, data1=g["raw"], data2=g["re_file"])
    for tag, kw in {"half": dict(data_res=[8, 8, 9], pde_res=[8, 8, 9], t_duration=0.5, n_samples=2, offset=1),
                    "quarter_sub": dict(data_res=[4, 4, 9], pde_res=[4, 4, 9], t_duration=0.25, n_samples=3, offset=0)}.items():
        ds = MultipleReynoldsKFaDataset(paths=[path], raw_res=[8, 8, 9], **kw)
        assert len(ds) == g[f"{tag}_u"].shape[0]
        for i in range(len(ds)):
            u, a, re = ds[i]
            assert np.array_equal(u.numpy(), g[f"{tag}_u"][i]) and np.array_equal(a.numpy(), g[f"{tag}_a"][i])
            assert float(re) == g[f"{tag}_re"][i]
    it = sample_data([1, 2])
    assert [next(it) for _ in range(5)] == [1, 2, 1, 2, 1]


def test_multistep_lr_matches_torch():
    """trainer.MultiStepLR (for FusedAdam) follows torch.optim.lr_scheduler.MultiStepLR step for step, including a resume."""
    import types
    from pde_policylearning_amd.trainer import MultiStepLR
    p = torch.nn.Parameter(torch.zeros(1))
    topt = torch.optim.Adam([p], lr=0.0025)
    tsch = torch.optim.lr_scheduler.MultiStepLR(topt, milestones=[3, 5, 9], gamma=0.5)
    opt = types.SimpleNamespace(lr=0.0025)
    sch = MultiStepLR(opt, milestones=[5, 3, 9], gamma=0.5)
    for i in range(12):
        assert abs(opt.lr - topt.param_groups[0]["lr"]) < 1e-15, i
        topt.step()
        tsch.step()
        sch.step()
        if i == 6:
            opt2 = types.SimpleNamespace(lr=123.0)
            sch2 = MultiStepLR(opt2, milestones=[1], gamma=0.1)
            sch2.load_state_dict(sch.state_dict())
            assert opt2.lr == opt.lr and sch2.last_epoch == 7


def test_bench_self_launch_spawns_one_process_per_rank(tmp_path):
    """`python bench.py --gpus N` without a launcher starts N ranks itself (bench.spawn_ranks): each child sees its RANK /
    LOCAL_RANK / WORLD_SIZE and a common rendezvous address, only rank 0's stdout comes back, a failing rank fails the job.
    Exercised here with a stand-in script that joins a gloo group (no GPU)."""
    import json
    import sys
    import bench
    script = tmp_path / "rank.py"
    script.write_text(
        "import os, json, sys\n"
        "import torch, torch.distributed as dist\n"
        "dist.init_process_group('gloo')\n"
        "t = torch.tensor([float(os.environ['RANK']) + 1.0])\n"
        "dist.all_reduce(t)\n"
        "print(json.dumps({'rank': dist.get_rank(), 'world': dist.get_world_size(), 'sum': float(t), 'local': os.environ['LOCAL_RANK']}))\n"
        "r = dist.get_rank()\n"
        "dist.destroy_process_group()\n"
        "sys.exit(int(os.environ.get('FAIL_RANK', '-1')) == r)\n")
    out, rc = bench.spawn_ranks(2, [sys.executable, str(script)])
    rec = json.loads(out.strip().splitlines()[-1])
    assert rc == 0 and rec == {"rank": 0, "world": 2, "sum": 3.0, "local": "0"}
    out, rc = bench.spawn_ranks(2, [sys.executable, str(script)], extra_env={"FAIL_RANK": "1"})
    assert rc != 0


def test_bucket_hooks_are_released_and_live_extent_growth_is_refused():
    """a segmented bucket registers parameter hooks and a direct-write hook; close() (or dropping the bucket)
    releases both, and a batch with a longer last dimension than the sample the exchange was planned on is refused."""
    import gc
    from torch import nn
    from pde_policylearning_amd import functional as F
    from pde_policylearning_amd.trainer import FlatGradBucket

    class Conv(nn.Module):                       # stand-in for basics.SpectralConv3d: four complex corner weights
        def __init__(self):
            super().__init__()
            self.modes3 = 6
            for i in range(1, 5):
                setattr(self, f"weights{i}", nn.Parameter(torch.zeros(2, 2, 3, 3, 6, dtype=torch.cfloat)))
            self._live_last = 2
    m = Conv()
    n0 = len(F.DIRECT_WRITE_HOOKS)
    b = FlatGradBucket(m.parameters())
    b._live_modules = [(m, 2)]
    b.enable_segmented_exchange(min_bytes=64, live_last={w: 2 for w in m.parameters()})
    assert len(F.DIRECT_WRITE_HOOKS) == n0 + 1 and len(b._hook_handles) == 4
    assert b.planned_wire_bytes() == 4 * 4 * (2 * 2 * 3 * 3 * 2 * 2)          # live slices only
    b.all_reduce()                                # single process: nothing to exchange, extents unchanged
    m._live_last = 3
    with pytest.raises(RuntimeError, match="live last-dim"):
        b.all_reduce()
    b.close()
    assert len(F.DIRECT_WRITE_HOOKS) == n0 and not b._hook_handles
    b2 = FlatGradBucket(m.parameters()).enable_segmented_exchange(min_bytes=64)
    assert len(F.DIRECT_WRITE_HOOKS) == n0 + 1
    del b2
    gc.collect()
    F._notify_direct([])                          # a dead bucket's entry is dropped, not called
    assert len(F.DIRECT_WRITE_HOOKS) == n0


def test_rno2d_accepts_list_pad_amount_and_picks_the_narrowest_twin():
    from pde_policylearning_amd.neuralop.models import RNO2d
    m = RNO2d(4, 4, 20, 0, layer_num=1, pad_amount=[2, 2], pad_dim='both')
    assert m._wide_twin().width == 32
    assert RNO2d(4, 4, 34, 0, layer_num=1)._wide_twin().width == 64


def test_fused_adam_dead_slice_plan_on_cpu():
    """FusedAdam.skip_dead_slices: the layout plan (row-sliced blocks merged, compact moment offsets 16-byte aligned), the
    full <-> compact moment round trip, re-planning when a module announces a longer live extent, and hook release -
    host logic only (the kernels: tests/test_lazy_adam_gpu.py)."""
    from torch import nn
    from pde_policylearning_amd.trainer import FlatGradBucket, FusedAdam

    class Conv(nn.Module):
        def __init__(self):
            super().__init__()
            self.modes3 = 6
            for i in range(1, 5):
                setattr(self, f"weights{i}", nn.Parameter(torch.rand(2, 2, 3, 3, 6, dtype=torch.cfloat)))
            self._live_last = 2

        def direct_grad_params(self):
            return [self.weights1, self.weights2, self.weights3, self.weights4]

    class Net(nn.Module):
        def __init__(self):
            super().__init__()
            self.a, self.c, self.b = nn.Linear(4, 4), Conv(), nn.Linear(4, 3)

    m = Net()
    bucket = FlatGradBucket(m.parameters(), direct_module=m)
    opt = FusedAdam(bucket, weight_decay=1e-4, skip_dead_slices=True)
    assert opt.skip_dead_slices() is True
    assert opt._runs == [("rows", 0, 144, 12, 4, 0), ("dense", 1728, 35, 576)]
    assert opt.exp_avg.numel() == 576 + 35 and bucket.flat.numel() == 1728 + 35
    opt.exp_avg.copy_(torch.arange(opt.exp_avg.numel(), dtype=torch.float32))
    opt.exp_avg_sq.copy_(torch.arange(opt.exp_avg.numel(), dtype=torch.float32) * 2)
    before = opt.exp_avg.clone(), opt.exp_avg_sq.clone()
    full = opt._full_moments()
    assert full[0].numel() == bucket.flat.numel()
    assert torch.equal(full[0][:12], torch.tensor([0., 1, 2, 3] + [0.] * 8))          # live pairs, then the dead part of row 0
    opt._scatter_full(*full)
    assert torch.equal(opt.exp_avg, before[0]) and torch.equal(opt.exp_avg_sq, before[1]) and opt._dead == {}
    # dead moments that are not zero (a loaded checkpoint) are kept beside the compact ones
    full[0][4] = 7.0
    opt._scatter_full(*full)
    assert 0 in opt._dead and float(opt._dead[0][0][0]) == 7.0
    assert float(opt._full_moments()[0][4]) == 7.0
    # a module about to read five slices re-plans (nothing is pending: no kernel runs), and a plan never shrinks
    m.c._dead_slice_guard(m.c, 5)
    assert opt._runs[0] == ("rows", 0, 144, 12, 10, 0) and float(opt._full_moments()[0][4]) == 7.0
    m.c._live_last = 1
    opt.skip_dead_slices()
    assert opt._runs[0][4] == 10
    # every slice live: nothing to skip, the optimizer is back on the full layout
    m.c._dead_slice_guard(m.c, 6)
    assert opt._runs is None and opt.exp_avg.numel() == bucket.flat.numel() and float(opt.exp_avg[4]) == 7.0
    assert "_dead_slice_guard" not in m.c.__dict__
    opt.close()


def test_no_packed_fp32_op_sel_hazard_in_any_kernel():
    """gfx950 hazard (tools/pk_opsel_hazard.hip): a v_pk_{mul,add,fma}_f32 that takes its LOW result's src1
    operand from the HIGH dword of a VGPR pair reads it as zero in lanes 48-63 now and then while the SIMD's matrix pipe is
    busy - with ANOTHER wave's matrix instructions, so a kernel without any (the complex arithmetic of the spectral middle, the
    PINO loss FFTs) is exposed as soon as a matrix kernel shares its CU (a caller's second stream, a DP overlap).  hipcc's SLP
    vectorizer is what picks the form: the library is built with -fno-slp-vectorize (build.py; round 5: 0.5 % faster to 1.4 %
    slower per workload) and fno_dev.h::natural_pair keeps it out of the explicit packed code, and the built code object is
    linted here: NO kernel may carry it."""
    import importlib.util
    tool = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "check_opsel.py")
    spec = importlib.util.spec_from_file_location("check_opsel", tool)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    if not os.path.exists(mod.OBJDUMP):
        pytest.skip("llvm-objdump of the ROCm toolchain not present")
    from pde_policylearning_amd import _lib
    _lib.lib()
    res = mod.scan(_lib.LIB_PATH)
    names = " ".join(res)
    for must in ("k_blk_fwd_t", "k_block_bwd_g2", "k_proj_bwd_t", "k_proj_fwd_w", "k_spec_mid", "k_pino_plane_fwd", "k_adam"):
        assert must in names, f"the scan did not see {must}"
    assert any(v[0] for k, v in res.items() if "k_proj_bwd_t" in k), "matrix instructions not recognised"
    bad = {k: v[1] for k, v in res.items() if v[1]}
    assert not bad, f"packed-fp32 op_sel hazard forms: {bad}"


def test_reference_yamls_become_the_run_plan(tmp_path):
    """`train_observer --train_yaml configs/base_fno.yaml` (BASELINE.json north_star; reference: libs/arguments.py:10-39 merge,
    run_pde_observers.py:336-346): the YAML's keys win over the flags, a key given twice keeps its last value, --set_epoch /
    --set_re apply after the merge, and the settings the loop reads come out as the reference's main() would see them."""
    from pde_policylearning_amd import train_observer as T
    from tests import yaml_fixtures as Y
    paths = {}
    for name, text in (("base_fno", Y.BASE_FNO), ("minchan_rno", Y.MINCHAN_RNO), ("matlab_rno", Y.MATLAB_RNO)):
        paths[name] = tmp_path / (name + ".yaml")
        paths[name].write_text(text)
    parse = lambda *argv: T.plan_from_yaml(T.build_parser().parse_args(list(argv)))
    p = parse("--train_yaml", str(paths["base_fno"]), "--batch-size", "7", "--modes", "99")
    assert (p.model, p.dataset, p.modes, p.width, p.batch_size, p.x_range, p.y_range) == ("FNO2dObserver", "SequentialPDEDataset", 12, 32, 20, 32, 32)
    assert (p.ntrain, p.ntest, p.epochs, p.random_split, p.recurrent_model, p.model_timestep) == (7500, 2500, 500, True, False, 1)
    assert (p.learning_rate, p.weight_decay, p.data_folder) == (0.001, 0.0001, "./data/planes_channel180_minchan")
    p = parse("--train_yaml", str(paths["base_fno"]), "--set_epoch", "3", "--data-folder", "/data/mine")
    assert p.epochs == 3 and p.data_folder == "/data/mine"
    p = parse("--train_yaml", str(paths["minchan_rno"]))
    assert (p.model, p.dataset, p.width, p.layer_num, p.batch_size, p.epochs) == ("RNO2dObserver", "SequentialPDEDataset", 34, 3, 32, 200)
    assert (p.recurrent_model, p.recurrent_index, p.model_timestep, p.random_split) == (True, 0, 2, False)      # `timestep: 2`
    p = parse("--train_yaml", str(paths["matlab_rno"]), "--set_re", "180")
    assert (p.model, p.dataset, p.plane_indexs, p.pde_loss_weight, p.model_timestep) == ("PINObserverFullField", "FullFieldNSDataset", [-10, -8, -6], 1.0, 1)
    assert (p.width, p.fullfield_width, p.batch_size, p.layer_num, p.epochs, p.Re) == (34, 64, 32, 1, 100, 180)
    assert (p.ntrain, p.ntest, p.init_cond_path) == (280, 20, "./data/channel180_minchan_mf.mat")
    with pytest.raises(ValueError, match="ntrain"):
        parse("--data-folder", "/x")                    # no YAML, no counts
    q = parse("--data-folder", "/x", "--ntrain", "8", "--ntest", "2", "--model", "RNO2dObserver")
    assert q.recurrent_model and q.dataset == "PDEDataset"


def test_reference_checkpoint_leaf_names_are_tolerated():
    """The spectral weights of a reference checkpoint sit under a leaf name that its tltorch version decides (unpinned) and may
    be complex: any single leaf below `fno_blocks.convs.weight.N.` loads (spectral_convolution.py:253-268)."""
    from pde_policylearning_amd.neuralop.models import FNO2d
    torch.manual_seed(1)
    src = FNO2d(8, 8, 32, in_channels=3, out_channels=1)
    sd = src.state_dict()
    foreign = {}
    for k, v in sd.items():
        if ".convs.weight." in k and k.endswith(".tensor"):
            n = int(k.split(".convs.weight.")[1].split(".")[0])
            # two spellings a tltorch release could use: a real view under another name, a complex tensor
            foreign[k[:-len("tensor")] + ("_tensor_real" if n % 2 else "weight")] = v.clone() if n % 2 else torch.view_as_complex(v.clone())
        else:
            foreign[k] = v.clone()
    dst = FNO2d(8, 8, 32, in_channels=3, out_channels=1)
    missing, unexpected = dst.load_state_dict(foreign, strict=True)
    assert not missing and not unexpected
    for (k, a), (_, b) in zip(sd.items(), dst.state_dict().items()):
        assert torch.equal(a, b), k


def test_library_shares_the_hip_runtime_of_torch():
    """One HIP runtime per process: the library works on torch's pointers and streams, so it must resolve libamdhip64.so.7 to
    the copy torch loaded (pde_policylearning_amd/_lib.py lib()).  A fresh interpreter that has NOT imported torch loads the
    library through _lib.lib(), then torch: /opt/rocm's runtime must not appear beside the wheel's."""
    import subprocess, sys
    code = ("import sys; sys.path.insert(0, %r)\n"
            "from pde_policylearning_amd import _lib\n_lib.lib()\nimport torch, os\n"
            "tl = os.path.realpath(os.path.join(os.path.dirname(torch.__file__), 'lib'))\n"
            "libs = sorted({os.path.realpath(l.split()[-1]) for l in open('/proc/self/maps') if 'libamdhip64' in l or 'libhsa-runtime64' in l})\n"
            "own = [l for l in libs if l.startswith(tl)]\n"
            "print(libs)\n"
            "assert not own or own == libs, libs\n"
            "assert len({os.path.basename(l).split('.so')[0] for l in libs}) == len(libs), libs\n") % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
