"""Plane datasets with the reference surface and on-disk format (libs/pde_data_loader.py:8-127):
a folder of per-timestep `P_planes_######.npy` / `V_planes_######.npy` files (or `P_plane` / `V_plane`) and a pickled
`metadata.npy` dict holding per-field `mean` / `std` arrays (run_control.py:234-292 writes them).
Items are normalised on the host exactly as the reference does; trainer.DevicePrefetcher moves batches to the GPU
asynchronously (pinned staging + a copy stream) instead of the reference's per-step `.cuda().float()`
(run_pde_observers.py:173)."""
import os

import numpy as np
import torch
from torch.utils.data import Dataset

from .utilities3 import NormalizerGivenMeanStd


def _plane_names(metadata):
    if 'P_planes' in metadata:
        return 'P_planes', 'V_planes'
    if 'P_plane' in metadata:
        return 'P_plane', 'V_plane'
    raise RuntimeError("Not recognized key name!")


class _PlaneFolder(Dataset):
    def _open(self, data_folder, data_index, downsample_rate, x_range, y_range, use_patch):
        self.data_folder = data_folder
        self.downsample_rate, self.x_range, self.y_range = downsample_rate, x_range, y_range
        self.metadata = np.load(os.path.join(data_folder, 'metadata.npy'), allow_pickle=True).tolist()
        self.file_list = os.listdir(data_folder)
        p_name, v_name = _plane_names(self.metadata)
        self.p_plane_files = sorted(f for f in self.file_list if p_name in f)
        self.v_plane_files = sorted(f for f in self.file_list if v_name in f)
        self.p_plane_mean, self.p_plane_std = self.metadata[p_name]['mean'], self.metadata[p_name]['std']
        self.v_plane_mean, self.v_plane_std = self.metadata[v_name]['mean'], self.metadata[v_name]['std']
        self.data_index = data_index
        self.data_length = len(data_index)
        self.use_patch = use_patch
        crop = self._crop_stats
        self.p_norm = NormalizerGivenMeanStd(crop(self.p_plane_mean), crop(self.p_plane_std))
        self.v_norm = NormalizerGivenMeanStd(crop(self.v_plane_mean), crop(self.v_plane_std))

    def _crop_stats(self, a):
        if self.use_patch:                                           # pde_data_loader.py:31-35
            return a.reshape(-1, self.x_range, self.y_range).mean(0)
        d = self.downsample_rate                                      # :37-40
        return a[::d, ::d][:self.x_range, :self.y_range]

    def _load_plane(self, name, norm):
        t = torch.tensor(np.load(os.path.join(self.data_folder, name)))
        if self.use_patch:                                            # :54-57
            t = t.reshape(-1, self.x_range, self.y_range)
        else:
            d = self.downsample_rate
            t = t[::d, ::d][:self.x_range, :self.y_range]
        return norm.encode(t)


class PDEDataset(_PlaneFolder):
    """(p_plane, v_plane), each (X, Y, 1), normalised  (pde_data_loader.py:8-69)."""

    def __init__(self, args, data_folder, data_index, downsample_rate, x_range, y_range, use_patch=False, full_field=False):
        super().__init__()
        self._open(data_folder, data_index, downsample_rate, x_range, y_range, use_patch)

    def __len__(self):
        return self.data_length

    def __getitem__(self, index):
        i = self.data_index[index]
        return (self._load_plane(self.p_plane_files[i], self.p_norm).unsqueeze(-1),
                self._load_plane(self.v_plane_files[i], self.v_norm).unsqueeze(-1))


class SequentialPDEDataset(_PlaneFolder):
    """(p, v), each (timestep, X, Y): `args.model_timestep` consecutive planes  (pde_data_loader.py:72-127;
    the reference reads self.p_plane_* without ever setting them in this class - they are set here)."""

    def __init__(self, args, data_folder, data_index, downsample_rate, x_range, y_range, use_patch=False, full_field=True):
        super().__init__()
        self.timestep = args.model_timestep
        self.full_field = full_field
        self._open(data_folder, data_index, downsample_rate, x_range, y_range, use_patch)

    def __len__(self):
        return self.data_length // self.timestep

    def __getitem__(self, index):
        ps, vs = [], []
        for t in range(self.timestep):
            i = self.data_index[index * self.timestep + t]
            ps.append(self._load_plane(self.p_plane_files[i], self.p_norm))
            vs.append(self._load_plane(self.v_plane_files[i], self.v_norm))
        return torch.stack(ps), torch.stack(vs)


class FullFieldNSDataset(Dataset):
    """Full-field channel-flow samples for the plane-prediction observer with the physics-informed term
    (libs/pde_data_loader.py:135-198; consumer run_pde_observers.py:200-231).  Folder format: per-timestep
    `U_field_######.npy`, `W_field_######.npy` (Nx, Ny+1, Nz), `V_field_######.npy` (Nx, Ny, Nz) and a pickled `metadata.npy`
    with `re`, `U_field.dpdx[i]`, `V_field.{mean,std}` and `P_planes.{mean,std}`.

    Item: (wall plane of v, normalised [T, X, Z]; target planes of v, normalised [T, P, X, Z]; U, V, W [T, ...];
    Re [T]; dPdx [T]).  The input plane and every target plane share ONE normaliser, the statistics of the wall plane
    V[:, -1, :] (:154-161)."""

    def __init__(self, args, data_folder, data_index, plane_indexs, downsample_rate, x_range, y_range, use_patch=False,
                 full_field=True):
        super().__init__()
        self.timestep = args.model_timestep
        self.data_folder, self.full_field = data_folder, full_field
        self.downsample_rate, self.x_range, self.y_range = downsample_rate, x_range, y_range      # kept, unused (as upstream)
        self.metadata = np.load(os.path.join(data_folder, 'metadata.npy'), allow_pickle=True).tolist()
        self.re = torch.tensor(self.metadata['re'])
        self.dpdx_all = self.metadata['U_field']['dpdx']
        self.file_list = os.listdir(data_folder)
        self.u_field_files, self.v_field_files, self.w_field_files = (
            sorted(f for f in self.file_list if tag in f) for tag in ('U_field', 'V_field', 'W_field'))
        self.scale_factor = 1
        v_stats = self.metadata['V_field']
        self.bound_v_mean, self.bound_v_std = v_stats['mean'][:, -1, :], v_stats['std'][:, -1, :] / self.scale_factor
        self.v_field_mean, self.v_field_std = v_stats['mean'][:, 1:-1, :], v_stats['std'][:, 1:-1, :]
        self.data_index = data_index
        self.data_length = len(data_index)
        self.plane_indexs = plane_indexs
        self.bound_v_norm = NormalizerGivenMeanStd(self.bound_v_mean, self.bound_v_std)
        self.v_field_norm = self.bound_v_norm
        self.p_plane_mean, self.p_plane_std = self.metadata['P_planes']['mean'], self.metadata['P_planes']['std']
        self.p_plane_norm = NormalizerGivenMeanStd(self.p_plane_mean, self.p_plane_std)

    def __len__(self):
        return self.data_length // self.timestep

    def _field(self, names, i):
        return torch.tensor(np.load(os.path.join(self.data_folder, names[i])))

    def __getitem__(self, index):
        planes, targets, us, vs, ws, res, dpdxs = [], [], [], [], [], [], []
        for t in range(self.timestep):
            i = self.data_index[index * self.timestep + t]
            v = self._field(self.v_field_files, i)
            us.append(self._field(self.u_field_files, i))
            vs.append(v)
            ws.append(self._field(self.w_field_files, i))
            planes.append(self.bound_v_norm.encode(v[:, -1, :]))
            targets.append(torch.stack([self.v_field_norm.encode(v[:, p, :]) for p in self.plane_indexs]))
            res.append(self.re)
            dpdxs.append(self.dpdx_all[i])
        return (torch.stack(planes), torch.stack(targets), torch.stack(us), torch.stack(vs), torch.stack(ws),
                torch.tensor(res), torch.tensor(dpdxs))
