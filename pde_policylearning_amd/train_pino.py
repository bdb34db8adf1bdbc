"""PINO pre-training / fine-tuning loop: the counterpart of train_pino.py (eval_ns :24-38, train_ns :41-137, subprocess
:140-221) with the step on the engine: PINObserver2d (spectral convolutions, pointwise layers, lifting and projection
kernels), the Navier-Stokes residual loss (fno_pino_loss_*), fused LpLoss, flat-bucket Adam with the MultiStepLR schedule,
asynchronous input staging.  Same YAML keys as the reference configs (configs/pino-observer-finetune-1s.yaml).  No W&B.

  python -m pde_policylearning_amd.train_pino --config configs/pino-observer-finetune-1s.yaml [--ckpt F] [--test]
"""
import argparse
import os
import random

import numpy as np
import torch
import torch.distributed as dist
import yaml
from torch.utils.data import DataLoader

from .libs.models.pino_models import PINObserver2d
from .libs.pino_utils.datasets import MultipleReynoldsKFaDataset, sample_data
from .libs.pino_utils.losses import get_forcing
from .libs.pino_utils.utils import count_params, dict2str, save_ckpt
from .trainer import (enable_dp_exchange, DevicePrefetcher, FlatGradBucket, FusedAdam, FusedLpLoss, MultiStepLR, PinoObjective, broadcast_parameters,
                      shard_batch, train_step)


@torch.no_grad()
def eval_ns(model, val_loader, criterion, device):
    """mean and standard error of the per-batch relative L2 error (train_pino.py:24-38)."""
    model.eval()
    errs = []
    for u, a, re in DevicePrefetcher(val_loader, device):
        errs.append(float(criterion(model(a, re).reshape(u.shape), u)))
    model.train()
    n = len(errs)
    return float(np.mean(errs)), (float(np.std(errs, ddof=1) / np.sqrt(n)) if n > 1 else float("nan"))


def train_ns(model, train_u_loader, val_loader, optimizer, scheduler, device, config, args, log=print):
    """iterations start_iter .. num_iter: one batch, loss = xy * data + f * PDE + ic * IC, Adam step, scheduler step;
    evaluation every eval_step, checkpoint every save_step (train_pino.py:41-137).  Returns the per-iteration log dicts
    (device tensors are turned into floats only every `log_every` iterations: no host sync in between)."""
    tcfg = config['train']
    world = dist.get_world_size() if dist.is_initialized() else 1
    rank = dist.get_rank() if dist.is_initialized() else 0
    ckpt_dir = os.path.join('exp', config['log']['logdir'], 'ckpts')
    S = config['data']['pde_res'][0]
    objective = PinoObjective(get_forcing(S).to(device), config['data']['t_duration'], tcfg['ic_loss'], tcfg['f_loss'],
                              tcfg['xy_loss'], scale=1.0 / world)       # mean-reduced losses: ranks average
    lploss = FusedLpLoss(size_average=True)
    bucket = optimizer.bucket
    batches = sample_data(DevicePrefetcher(train_u_loader, device))
    log_every = max(1, int(getattr(args, "log_every", 100)))
    history = []
    for e in range(tcfg['start_iter'], tcfg['num_iter']):
        u, a_in, re = next(batches)
        if world > 1:
            u, a_in, re = (shard_batch(t, rank, world) for t in (u, a_in, re))
        loss = train_step(model, bucket, optimizer, (a_in, re), (u, a_in, re), objective)
        scheduler.step()
        evaluate = val_loader is not None and e % tcfg['eval_step'] == 0       # train_pino.py:112-118: every eval_step iterations,
        if e % log_every == 0 or e == tcfg['num_iter'] - 1 or evaluate:          # whatever the logging cadence is
            rec = {k: float(v) for k, v in objective.last_terms.items()}
            rec.update({'iter': e, 'train loss': float(loss) * world})
            if evaluate:
                rec['val error'] = eval_ns(model, val_loader, lploss, device)[0]
            history.append(rec)
            if rank == 0:
                log(dict2str(rec))
        if rank == 0 and e % tcfg['save_step'] == 0 and e > 0:
            os.makedirs(ckpt_dir, exist_ok=True)
            save_ckpt(os.path.join(ckpt_dir, f'model-{e}.pt'), model, optimizer, scheduler)
    return history


def build_model(config, device):
    m = config['model']
    return PINObserver2d(modes1=m['modes1'], modes2=m['modes2'], modes3=m['modes3'], fc_dim=m['fc_dim'], layers=m['layers'],
                         act=m['act'], pad_ratio=m['pad_ratio']).to(device)


def _dataset(config, paths, res, n_samples, offset):
    d = config['data']
    return MultipleReynoldsKFaDataset(paths=paths, raw_res=d['raw_res'], data_res=res, pde_res=res, n_samples=n_samples,
                                      offset=offset, t_duration=d['t_duration'])


def run(config, args, log=print):
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group("nccl", device_id=device)
    torch.manual_seed(args.seed)
    random.seed(args.seed)
    model = build_model(config, device)
    log(f'Number of parameters: {count_params(model)}')
    ckpt = torch.load(args.ckpt, map_location=device) if args.ckpt else None
    if ckpt:
        model.load_state_dict(ckpt['model'])
    d, t = config['data'], config['train']
    if args.test:
        testset = _dataset(config, d.get('paths', d['test_paths']), config['test']['data_res'], d['n_test_samples'], d['testoffset'])
        err, std = eval_ns(model, DataLoader(testset, batch_size=config['test']['batchsize']), FusedLpLoss(size_average=True), device)
        log(f'Averaged test relative L2 error: {err}; Standard error: {std}')
        return err, std
    u_set = _dataset(config, d['train_paths'], d['data_res'], d['n_data_samples'], d['offset'])
    valset = _dataset(config, d['test_paths'], config['test']['data_res'], d['n_test_samples'], d['testoffset'])
    u_loader = DataLoader(u_set, batch_size=t['batchsize'] * world, shuffle=True, drop_last=world > 1)
    val_loader = DataLoader(valset, batch_size=t['batchsize'])
    broadcast_parameters(model)
    bucket = FlatGradBucket(model.parameters(), direct_module=model)
    if world > 1:      # 268 MB (modes 8) / 4.2 GB (modes 20) of spectral weights: per-layer segments overlap the backward pass
        a0 = torch.as_tensor(u_set[0][1])[None].to(device).float()
        enable_dp_exchange(bucket, model, (a0, torch.full((1,), 300.0, device=device)))
    optimizer = FusedAdam(bucket, lr=t['base_lr'])
    scheduler = MultiStepLR(optimizer, milestones=t['milestones'], gamma=t['scheduler_gamma'])
    if ckpt and ckpt.get('optim') is not None:
        optimizer.load_state_dict(ckpt['optim'])
        scheduler.load_state_dict(ckpt['scheduler'])
        t['start_iter'] = scheduler.last_epoch
    return train_ns(model, u_loader, val_loader, optimizer, scheduler, device, config, args, log=log)


def build_parser():
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument('--config', type=str, required=True)
    ap.add_argument('--seed', type=int, default=0)
    ap.add_argument('--ckpt', type=str, default=None)
    ap.add_argument('--test', action='store_true')
    ap.add_argument('--log-every', type=int, default=100)
    return ap


def main():
    args = build_parser().parse_args()
    with open(args.config, 'r') as f:
        config = yaml.load(f, yaml.FullLoader)
    run(config, args)


if __name__ == '__main__':
    main()
