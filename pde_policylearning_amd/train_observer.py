"""Observer training loop: the counterpart of run_pde_observers.py:66-240 (dataset split, loaders, model choice, Adam,
LpLoss on decoded fields, per-epoch train / test relative L2) for the `PDEDataset` path and the `FullFieldNSDataset` path
(plane-prediction observer + physics-informed channel-flow term, :200-231), with every hot step in the engine:
asynchronous input staging (trainer.DevicePrefetcher), the fused FNO model or the engine-backed RNO, fused decode + loss,
flat-bucket Adam, optional data parallelism (one process per GPU, overlapped RCCL all-reduce).  No W&B, no MATLAB control
environment (out of scope).

  python -m pde_policylearning_amd.train_observer --data-folder DIR --ntrain 800 --ntest 200 --modes 12 --width 64 \\
         --x-range 128 --y-range 128 --batch-size 64 --epochs 5 [--model FNO2dObserver|RNO2dObserver]
  python -m pde_policylearning_amd.train_observer --data-folder DIR --ntrain 800 --ntest 200 --dataset FullFieldNSDataset \
         --model PINObserverFullField --modes 12 --width 64 --plane-indexs -10 -8 -6 --pde-loss-weight 1.0 [--init-cond-path F.mat]
  (N GPUs: python -m torch.distributed.run --nproc-per-node N -m pde_policylearning_amd.train_observer ...)

The reference's own YAMLs drop in (run_pde_observers.py:336-346 + libs/arguments.py:10-39): `--train_yaml configs/base_fno.yaml`
is loaded with the reference's merge rule (YAML keys OVER command-line values; duplicate YAML keys: the last one wins, as
yaml.safe_load does), `--set_epoch` / `--set_re` are applied after the merge as upstream, and the keys the loop reads
(model_name, dataset_name, modes, width, batch_size, x_range, y_range, learning_rate, weight_decay, epochs, recurrent_model,
recurrent_index, layer_num, model_timestep, plane_indexs, pde_loss_weight, random_split, ntrain, ntest, DATA_FOLDER,
downsample_rate, init_cond_path, Re) become the run plan (`plan_from_yaml`); keys of the out-of-scope control loop / W&B are
carried along and ignored:
  python -m pde_policylearning_amd.train_observer --train_yaml configs/base_fno.yaml [--data-folder DIR] [--set_epoch 5]
"""
import argparse
import os
import time
import types

import torch
import torch.distributed as dist
from torch.utils.data import DataLoader

from .libs.models.fno_models import FNO2dObserver
from .libs.models.rno_models import RNO2dObserver
from .libs.pde_data_loader import FullFieldNSDataset, PDEDataset, SequentialPDEDataset
from .trainer import (enable_dp_exchange, DevicePrefetcher, FlatGradBucket, FullFieldObjective, FusedAdam, FusedLpLoss, MeanStdDecoder,
                      broadcast_parameters, shard_batch, train_step)


def build_parser():
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--train_yaml", "--train-yaml", dest="train_yaml", default=None,
                    help="a reference YAML (configs/base_fno.yaml, matlab_rno.yaml, minchan_rno.yaml ...): its keys win over the flags")
    ap.add_argument("--set_epoch", "--set-epoch", dest="set_epoch", type=int, default=-1)        # run_pde_observers.py:344-345
    ap.add_argument("--set_re", "--set-re", dest="set_re", type=int, default=-1)                  # :342-343
    ap.add_argument("--data-folder", default=None, help="required without --train_yaml; with it: replaces the YAML's DATA_FOLDER")
    ap.add_argument("--ntrain", type=int, default=None)
    ap.add_argument("--ntest", type=int, default=None)
    ap.add_argument("--random-split", action="store_true")          # run_pde_observers.py:69-72
    ap.add_argument("--model", default="FNO2dObserver", choices=["FNO2dObserver", "RNO2dObserver", "PINObserverFullField"])
    ap.add_argument("--dataset", default="PDEDataset", choices=["PDEDataset", "SequentialPDEDataset", "FullFieldNSDataset"])   # configs/matlab_rno.yaml:21-22
    ap.add_argument("--recurrent-model", action="store_true")        # run_pde_observers.py:174-178
    ap.add_argument("--recurrent-index", type=int, default=0)
    ap.add_argument("--plane-indexs", type=int, nargs="+", default=[-10, -8, -6])                     # :62
    ap.add_argument("--pde-loss-weight", type=float, default=0.0)                                     # :56
    ap.add_argument("--init-cond-path", default=None, help=".mat initial condition holding the grid (x, y, z, ym); "
                    "default: the analytic tanh channel grid matching the data's shape")
    ap.add_argument("--Re", type=float, default=-1.0)
    ap.add_argument("--model-timestep", type=int, default=1)
    ap.add_argument("--modes", type=int, default=12)
    ap.add_argument("--width", type=int, default=32)
    ap.add_argument("--layer-num", type=int, default=1)
    ap.add_argument("--downsample-rate", type=int, default=1)
    ap.add_argument("--x-range", type=int, default=32)
    ap.add_argument("--y-range", type=int, default=32)
    ap.add_argument("--batch-size", type=int, default=20)
    ap.add_argument("--epochs", type=int, default=1)
    ap.add_argument("--learning-rate", type=float, default=1e-3)
    ap.add_argument("--weight-decay", type=float, default=1e-4)
    ap.add_argument("--seed", type=int, default=0)                  # run_pde_observers.py:25
    ap.add_argument("--no-shuffle", action="store_true", help="keep the (time-ordered) sample order; the reference shuffles")
    ap.add_argument("--graph", action="store_true", help="PINObserverFullField on one GPU: capture the training step of a full "
                    "batch into a hipGraph and replay it (the step is launch-bound: 1.9 vs 2.4 ms at the YAML's sizes); a "
                    "smaller last batch runs eagerly")
    ap.add_argument("--save-path", default=None, help="whole-module checkpoint written whenever the test rel-L2 improves "
                    "(run_pde_observers.py:307-315: torch.save(observer_model, './outputs/<path>_<exp>.pth'))")
    return ap


# YAML key -> run-plan attribute (everything else of the YAML is carried under its own name)
_YAML_KEYS = {"DATA_FOLDER": "data_folder", "model_name": "model", "dataset_name": "dataset"}


def load_train_yaml(path):
    """libs/arguments.py:10-13: yaml.safe_load of the whole file (a key given twice keeps its LAST value: matlab_rno.yaml
    and minchan_rno.yaml both set `width` twice)."""
    import yaml
    with open(path, "r") as f:
        d = yaml.safe_load(f)
    if not isinstance(d, dict):
        raise ValueError(f"{path}: a mapping of settings is expected")
    return d


def plan_from_yaml(args, yaml_dict=None):
    """The run plan of `train_observer` from parsed flags + (optionally) a reference YAML, with the reference's semantics:
    merge_args_with_yaml (libs/arguments.py:16-26: the YAML's keys replace the flags'), then --set_re / --set_epoch
    (run_pde_observers.py:342-345).  An explicit --data-folder still replaces DATA_FOLDER (the YAMLs hold paths of the
    authors' machines).  Returns a new namespace; `args` is not modified."""
    plan = dict(vars(args))
    explicit_folder = plan.get("data_folder")
    if yaml_dict is None and plan.get("train_yaml"):
        yaml_dict = load_train_yaml(plan["train_yaml"])
    y = dict(yaml_dict or {})
    for k, v in y.items():
        plan[_YAML_KEYS.get(k, k)] = v
    if y:
        if "dataset_name" not in y:
            # the upstream loop only trains SequentialPDEDataset / FullFieldNSDataset (run_pde_observers.py:170,200); YAMLs
            # without the key (base_fno.yaml, minchan_rno.yaml) mean the plane sequences
            plan["dataset"] = "SequentialPDEDataset"
        if "model_timestep" not in y:
            ts = y.get("timestep", 1)       # (minchan_rno.yaml: `timestep: 2`; base_fno.yaml: -1)
            plan["model_timestep"] = int(ts) if isinstance(ts, int) and ts > 0 else 1
        if explicit_folder:
            plan["data_folder"] = explicit_folder
        # run_pde_observers.py:104-107 builds the full-field observer with layers [64] * 5 whatever `width` says
        # (matlab_rno.yaml: width 34 belongs to the commented-out RNO2dObserver)
        plan["fullfield_width"] = 64
    if plan.get("set_re", -1) > 0:
        plan["Re"] = plan["set_re"]
    if plan.get("set_epoch", -1) > 0:
        plan["epochs"] = plan["set_epoch"]
    plan["recurrent_model"] = bool(plan.get("recurrent_model", False)) or plan.get("model") == "RNO2dObserver"
    ns = argparse.Namespace(**plan)
    missing = [k for k in ("data_folder", "ntrain", "ntest") if getattr(ns, k, None) is None]
    if missing:
        raise ValueError("train_observer: " + ", ".join(missing) + " not given (flags or --train_yaml)")
    if ns.model not in ("FNO2dObserver", "RNO2dObserver", "PINObserverFullField"):
        raise NotImplementedError(f"model_name {ns.model!r} is outside the accelerated hot path (FNO2dObserver, RNO2dObserver, "
                                  "PINObserverFullField)")
    return ns


def save_if_best(model, test_l2, best, path, rank, log):
    """run_pde_observers.py:307-315: keep the best-so-far model as a whole pickled module (the format run_control.py
    reads back with torch.load).  Rank 0 writes; every rank tracks the same `best` (test_l2 is evaluated on identical
    replicas and the full test set by every rank)."""
    if test_l2 >= best:
        return best
    if path and rank == 0:
        os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
        # the trainer hangs run-time state on the modules (the overlap hook = the FlatGradBucket with its gradient buffer,
        # process group and pending Work handle; direct-write flags; cached device grids): not part of a checkpoint
        stripped = []
        for m in model.modules():
            for k in ("_grad_overlap", "_direct_grads", "_grid_cache", "_dead_slice_guard", "_dead_slice_k"):
                if k in m.__dict__:
                    stripped.append((m, k, m.__dict__.pop(k)))
        # FusedAdam keeps every parameter as a view of ONE float32 buffer - complex weights as complex views of it, which
        # torch.save refuses ("tensors that view the same data as different types"): the pickle gets private copies
        views = [(p, p.data) for p in model.parameters()]
        try:
            for p, d in views:
                p.data = d.clone()
            torch.save(model, path)
        finally:
            for p, d in views:
                p.data = d
            for m, k, v in stripped:
                m.__dict__[k] = v
        log(f"Best model saved at {path}!")
    return test_l2


def make_train_loader(train_ds, args, world):
    """run_pde_observers.py:29,90: DataLoader(train, batch_size, shuffle=train_shuffle (default True), drop_last=False).
    The permutation comes from a generator seeded with args.seed, so every rank draws the same order and shard_batch cuts
    the same global batch; only the data-parallel case drops the ragged tail (the global batch must split evenly)."""
    shuffle = not getattr(args, "no_shuffle", False)
    gen = torch.Generator().manual_seed(int(args.seed)) if shuffle else None
    loader = DataLoader(train_ds, batch_size=args.batch_size * world, shuffle=shuffle, drop_last=world > 1, generator=gen)
    if len(loader) == 0:
        raise ValueError(f"empty training epoch: {len(train_ds)} samples < global batch {args.batch_size * world} "
                         f"(batch_size {args.batch_size} x {world} ranks, the ragged tail is dropped under data parallelism)")
    return loader


def run(args, log=print):
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group("nccl", device_id=dev)
    torch.manual_seed(args.seed)
    n = args.ntrain + args.ntest
    idx = torch.randperm(n) if args.random_split else torch.arange(n)
    if args.dataset == "FullFieldNSDataset":
        return run_full_field(args, idx, dev, rank, world, log)
    T = max(int(getattr(args, "model_timestep", 1) or 1), 1)
    recurrent = bool(getattr(args, "recurrent_model", False)) or args.model == "RNO2dObserver"
    ri = int(getattr(args, "recurrent_index", 0) or 0)
    X, Y = args.x_range, args.y_range
    if args.dataset == "SequentialPDEDataset":
        # run_pde_observers.py:75-82,170-183: items are `model_timestep` consecutive planes (T, X, Y); a recurrent model sees
        # the sequence (B, T, X, Y, 1) and is scored on time step `recurrent_index`, every other model sees T independent planes
        ds_args = types.SimpleNamespace(model_timestep=T)
        mk = lambda ix: SequentialPDEDataset(ds_args, args.data_folder, ix, args.downsample_rate, X, Y,
                                             use_patch=bool(getattr(args, "use_patch", False)))
        if recurrent:
            if not 0 <= ri < T:
                raise ValueError(f"recurrent_index {ri} outside the sequence of model_timestep {T}")
            def make_batch(p, v):
                return p.reshape(-1, T, X, Y, 1), v.reshape(-1, T, X, Y)[:, ri]
        else:
            def make_batch(p, v):
                return p.reshape(-1, X, Y, 1), v.reshape(-1, X, Y)
    else:
        ds_args = types.SimpleNamespace(model_timestep=1)
        mk = lambda ix: PDEDataset(ds_args, args.data_folder, ix, args.downsample_rate, X, Y)
        def make_batch(p, v):
            return (p.unsqueeze(1) if args.model == "RNO2dObserver" else p), v.squeeze(-1)
    train_ds, test_ds = mk(idx[:args.ntrain].tolist()), mk(idx[-args.ntest:].tolist())
    train_loader = make_train_loader(train_ds, args, world)
    test_loader = DataLoader(test_ds, batch_size=args.batch_size, shuffle=False, drop_last=False)
    if args.model == "FNO2dObserver":
        model = FNO2dObserver(args.modes, args.modes, args.width, use_v_plane=bool(getattr(args, "use_v_plane", False))).to(dev)
        forward = lambda p: model(p, None)
    elif args.model == "RNO2dObserver":
        model = RNO2dObserver(args.modes, args.modes, args.width, recurrent_index=ri, layer_num=args.layer_num).to(dev)
        forward = lambda p: model(p)                             # (B, T, X, Y, 1) -> time step `recurrent_index`
    else:
        raise NotImplementedError(f"{args.model} on plane datasets (PINObserverFullField trains on FullFieldNSDataset)")
    broadcast_parameters(model)
    fused = args.model == "FNO2dObserver"
    if fused and world > 1:
        bucket = FlatGradBucket.for_fno(model, split_layer=1)
    else:
        # RNO2d: the spectral weights' gradients are written in place while a step uses every parameter once (one time
        # step: functional.single_use); cleared in full for the accumulating fallback
        bucket = FlatGradBucket(model.parameters(), direct_module=model, zero_all=not fused)
        if world > 1:
            bucket.enable_segmented_exchange()       # RNO2d: 95 MB at width 64 leave in layer-ordered segments during backward
    opt = FusedAdam(bucket, lr=args.learning_rate, weight_decay=args.weight_decay)
    decoder = MeanStdDecoder(train_ds.v_norm.mean.numpy(), train_ds.v_norm.std.numpy(), eps=train_ds.v_norm.eps, device=dev)
    loss_fn = FusedLpLoss(size_average=False, decoder=decoder)   # myloss = LpLoss(size_average=False), :138
    history, best = [], float("inf")            # best_loss = 1e10 in the reference (:64)
    for ep in range(args.epochs):
        model.train()
        t0 = time.perf_counter()
        tot = torch.zeros((), device=dev)
        cnt = 0
        for p_plane, v_plane in DevicePrefetcher(train_loader, dev):
            if world > 1:
                p_plane, v_plane = shard_batch(p_plane, rank, world), shard_batch(v_plane, rank, world)
            p_plane, tgt = make_batch(p_plane, v_plane)
            tot += train_step(forward, bucket, opt, (p_plane,), tgt, loss_fn)     # :185-193
            cnt += tgt.shape[0]
        if world > 1:
            dist.all_reduce(tot)
            cnt *= world
        model.eval()
        test_tot, test_cnt = torch.zeros((), device=dev), 0
        with torch.no_grad():
            for p_plane, v_plane in DevicePrefetcher(test_loader, dev):
                p_plane, tgt = make_batch(p_plane, v_plane)
                test_tot += loss_fn(forward(p_plane).reshape(tgt.shape), tgt)
                test_cnt += tgt.shape[0]
        rec = dict(epoch=ep, train_l2=float(tot) / max(cnt, 1), test_l2=float(test_tot) / max(test_cnt, 1),
                   seconds=time.perf_counter() - t0)
        history.append(rec)
        opt.sync_dead_slices()       # (whole-module pickles read the parameters directly)
        best = save_if_best(model, rec["test_l2"], best, getattr(args, "save_path", None), rank, log)
        if rank == 0:
            log(f"epoch {ep}: train rel-L2 {rec['train_l2']:.5f}  test rel-L2 {rec['test_l2']:.5f}  {rec['seconds']:.2f} s")
    return history


def run_full_field(args, idx, dev, rank, world, log):
    """run_pde_observers.py:200-231 + its eval twin: PINObserverFullField predicts `plane_indexs` planes of v from the wall
    plane; loss = LpLoss(decoded planes) + pde_loss_weight * channel-flow physics term."""
    from .libs.envs.control_env import ChannelFlowRHS
    from .libs.models.pino_models import PINObserverFullField
    ds_args = types.SimpleNamespace(model_timestep=args.model_timestep)
    mk = lambda ix: FullFieldNSDataset(ds_args, args.data_folder, ix.tolist(), args.plane_indexs, args.downsample_rate,
                                       args.x_range, args.y_range)
    train_ds, test_ds = mk(idx[:args.ntrain]), mk(idx[-args.ntest:])
    train_loader = make_train_loader(train_ds, args, world)
    test_loader = DataLoader(test_ds, batch_size=args.batch_size, shuffle=False, drop_last=False)
    P, L = len(args.plane_indexs), 4
    model = PINObserverFullField(plane_num=P, modes1=[args.modes] * L, modes2=[args.modes] * L, modes3=[args.modes] * L,
                                 fc_dim=128, layers=[getattr(args, "fullfield_width", None) or args.width] * (L + 1), in_dim=1, out_dim=1, act="gelu",
                                 pad_ratio=[0.0, 0.0625]).to(dev)                   # run_pde_observers.py:117-131
    broadcast_parameters(model)
    bucket = FlatGradBucket(model.parameters(), direct_module=model)      # spectral-weight gradients written in place
    if world > 1:      # segments go on the wire as they complete; only the live last-dim slice of the spectral weights (1/12 at T = 1)
        s0 = train_ds[0][0]
        enable_dp_exchange(bucket, model, (torch.as_tensor(s0)[None].to(dev).float().permute(0, 2, 3, 1).unsqueeze(-1),
                                           torch.full((1, 1), 180.0, device=dev)))
    # the dead last-dim slices of the spectral weights (11/12 of them at T = 1) are replayed, not stepped: the loop reads the
    # weights through model.state_dict() / optimizer.state_dict() only, which bring them up to date first
    use_graph = bool(getattr(args, "graph", False)) and world == 1
    opt = FusedAdam(bucket, lr=args.learning_rate, weight_decay=args.weight_decay, skip_dead_slices=True, capturable=use_graph)
    env = None
    if args.pde_loss_weight > 0:
        Nx, Ny, Nz = train_ds[0][3].shape[1:]
        env = (ChannelFlowRHS.from_mat(args.init_cond_path, Re=args.Re) if args.init_cond_path
               else ChannelFlowRHS.tanh_channel(Nx, Ny, Nz, Re=args.Re))
        if (env.Nx, env.Ny, env.Nz) != (Nx, Ny, Nz):
            raise RuntimeError(f"grid {env.Nx, env.Ny, env.Nz} of {args.init_cond_path} does not match the fields {Nx, Ny, Nz}")
    norm = train_ds.bound_v_norm
    decoder = MeanStdDecoder(norm.mean.numpy(), norm.std.numpy(), eps=norm.eps, device=dev)
    objective = FullFieldObjective(decoder, args.plane_indexs, env, args.pde_loss_weight)
    forward = lambda plane, re: model(plane.permute(0, 2, 3, 1).unsqueeze(-1), re)     # 'btxy -> bxyt', + channel  (:204)
    history, best = [], float("inf")
    graphed, gshape = None, None
    for ep in range(args.epochs):
        model.train()
        t0 = time.perf_counter()
        tot, cnt = torch.zeros((), device=dev), 0
        for batch in DevicePrefetcher(train_loader, dev):
            if world > 1:
                batch = [shard_batch(t, rank, world) for t in batch]
            v_plane, v_field, U, V, W, re, _dpdx = [t.float() for t in batch]
            if use_graph and graphed is None and v_plane.shape[0] == args.batch_size:
                from .trainer import GraphedTrainStep      # the whole step (zero_grad .. Adam) captured once on a full batch
                graphed = GraphedTrainStep(forward, bucket, opt, (v_plane, re), (v_field, U, V, W), objective)
                gshape = tuple(v_plane.shape)
            if graphed is not None and tuple(v_plane.shape) == gshape:
                tot += graphed((v_plane, re), (v_field, U, V, W))
            else:
                tot += train_step(forward, bucket, opt, (v_plane, re), (v_field, U, V, W), objective)
            cnt += v_plane.shape[0]
        if world > 1:
            dist.all_reduce(tot)
            cnt *= world
        model.eval()
        test_tot, test_cnt = torch.zeros((), device=dev), 0
        with torch.no_grad():
            for batch in DevicePrefetcher(test_loader, dev):
                v_plane, v_field, U, V, W, re, _dpdx = [t.float() for t in batch]
                objective(forward(v_plane, re), (v_field, U, V, W))
                test_tot += objective.last_terms[0]                                  # test metric: the data term
                test_cnt += v_plane.shape[0]
        rec = dict(epoch=ep, train_l2=float(tot) / max(cnt, 1), test_l2=float(test_tot) / max(test_cnt, 1),
                   seconds=time.perf_counter() - t0)
        history.append(rec)
        opt.sync_dead_slices()       # (whole-module pickles read the parameters directly)
        best = save_if_best(model, rec["test_l2"], best, getattr(args, "save_path", None), rank, log)
        if rank == 0:
            log(f"epoch {ep}: train loss {rec['train_l2']:.5f}  test rel-L2 {rec['test_l2']:.5f}  {rec['seconds']:.2f} s")
    return history


def main():
    run(plan_from_yaml(build_parser().parse_args()))


if __name__ == "__main__":
    main()
