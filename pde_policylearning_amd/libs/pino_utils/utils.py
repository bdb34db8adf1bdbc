"""libs/pino_utils/utils.py surface used by the PINO fine-tuning loop: the space-time input grid, checkpointing, small helpers
(get_grid3d :107-115, save_ckpt :178-194, count_params, dict2str :197-201)."""
import numpy as np
import torch


def get_grid3d(S, T, time_scale=1.0, device='cpu'):
    """x, y: S points of [0, 1) (endpoint excluded); t: T points of [0, time_scale] (endpoint included); each (1, S, S, T, 1)."""
    xs = torch.tensor(np.linspace(0, 1, S + 1)[:-1], dtype=torch.float, device=device)
    ts = torch.tensor(np.linspace(0, 1 * time_scale, T), dtype=torch.float, device=device)
    gridx = xs.reshape(1, S, 1, 1, 1).repeat([1, 1, S, T, 1])
    gridy = xs.reshape(1, 1, S, 1, 1).repeat([1, S, 1, T, 1])
    gridt = ts.reshape(1, 1, 1, T, 1).repeat([1, S, S, 1, 1])
    return gridx, gridy, gridt


def count_params(model):
    return sum(p.numel() * (2 if p.is_complex() else 1) for p in model.parameters())


def save_ckpt(path, model, optimizer=None, scheduler=None):
    """{'model', 'optim', 'scheduler'} state dicts, the layout train_pino.py:165-168, 206-209 reads back."""
    torch.save({'model': model.state_dict(),
                'optim': optimizer.state_dict() if optimizer else None,
                'scheduler': scheduler.state_dict() if scheduler else None}, path)
    print(f'Checkpoint is saved to {path}')


def dict2str(log_dict):
    return ''.join(f'{k}: {v}|' for k, v in log_dict.items())
