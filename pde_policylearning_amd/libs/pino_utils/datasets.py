"""libs/pino_utils/datasets.py surface used by the PINO fine-tuning loop (train_pino.py:188-204):
`MultipleReynoldsKFaDataset` (:548-617) on the reference's files - an `.npz` with `data1` (N, T, S, S) vorticity
trajectories and `data2` (N,) Reynolds numbers when the path contains "multi_reynolds", otherwise one `.npy` whose
name carries `Re<digits>` - and `sample_data` (:24-27)."""
import re as _re

import numpy as np
import torch
from torch.utils.data import Dataset

from .utils import get_grid3d


def sample_data(loader):
    """endless batches"""
    while True:
        yield from loader


class MultipleReynoldsKFaDataset(Dataset):
    """Items (u [S, S, T], a [S, S, T, 4] = (x, y, t, u(t = 0) repeated), Re).  With `t_duration` = 1/K every trajectory
    is cut into K windows that share their end points; the initial condition of window j is the raw frame j * step.
    As in the reference, window (i, j) gets `re[i]` counted from the START of the file, not from `offset` (:607-610)."""

    def __init__(self, paths, data_res, pde_res, raw_res, n_samples=None, total_samples=None, idx=0, offset=0, t_duration=1.0):
        super().__init__()
        self.data_res, self.pde_res, self.raw_res = data_res, pde_res, raw_res
        self.t_duration, self.paths, self.offset, self.n_samples = t_duration, paths, offset, n_samples
        self.T = self.pde_res[2] if t_duration == 1.0 else int(self.pde_res[2] * t_duration) + 1
        self.re = None
        self.load()
        if total_samples is not None:
            self.data = self.data[idx:idx + total_samples]
            self.a_data = self.a_data[idx:idx + total_samples]

    def load(self):
        path = self.paths[0]
        if 'multi_reynolds' in path:
            f = np.load(path)
            raw, self.re = f['data1'], f['data2']
        else:
            raw = np.load(path, mmap_mode='r')
            self.re = torch.tensor([int(_re.search(r'Re(\d+)', path).group(1))] * raw.shape[0]).float()
        sub_x = self.raw_res[0] // self.data_res[0]
        sub_t = (self.raw_res[2] - 1) // (self.data_res[2] - 1)
        a_sub_x = self.raw_res[0] // self.pde_res[0]
        rows = slice(self.offset, self.offset + self.n_samples)
        data = raw[rows, ::sub_t, ::sub_x, ::sub_x]
        if self.t_duration != 0.:
            end_t = self.raw_res[2] - 1
            K = int(1 / self.t_duration)
            data, self.re = self.partition(data)
            a = raw[rows, 0:end_t:end_t // K, ::a_sub_x, ::a_sub_x].reshape(self.n_samples * K, 1, self.pde_res[0], self.pde_res[1])
        else:
            a = raw[rows, 0:1, ::a_sub_x, ::a_sub_x]
        self.data = torch.from_numpy(np.ascontiguousarray(data)).to(torch.float32).permute(0, 2, 3, 1)        # [N, S, S, T]
        self.a_data = torch.from_numpy(np.ascontiguousarray(a)).to(torch.float32).permute(0, 2, 3, 1)[:, :, :, :, None]
        gx, gy, gt = get_grid3d(self.pde_res[1], self.T)
        self.grid = torch.cat((gx[0], gy[0], gt[0]), dim=-1)                                                  # S x S x T x 3

    def partition(self, data):
        """(N, T, S, S) -> (K N, T // K + 1, S, S) overlapping windows and their Reynolds numbers."""
        N, T, S = data.shape[:3]
        K = int(1 / self.t_duration)
        step = T // K
        out = np.zeros((K * N, step + 1, S, S))
        res = np.ones((K * N,))
        for i in range(N):
            for j in range(K):
                out[i * K + j] = data[i, j * step:(j + 1) * step + 1]
                res[i * K + j] = self.re if type(self.re) == int else self.re[i]
        return out, res

    def __getitem__(self, idx):
        a = torch.cat((self.grid, self.a_data[idx].repeat(1, 1, self.T, 1)), dim=-1)
        return self.data[idx], a, self.re[idx]

    def __len__(self):
        return self.data.shape[0]
