"""FNO2dObserver with the reference surface (libs/models/fno_models.py:16-57)."""
import torch
from torch import nn

from ...neuralop.models import FNO2d


class FNO2dObserver(nn.Module):
    def __init__(self, modes1, modes2, width, use_v_plane=False):
        super().__init__()
        self.modes1, self.modes2, self.width = modes1, modes2, width
        self.use_v_plane = use_v_plane
        self.padding = 9
        self.input_channel_num = 4 if use_v_plane else 3
        self.fno2d = FNO2d(modes1, modes2, width, in_channels=self.input_channel_num, out_channels=1)
        self._grid_cache = {}

    def forward(self, p_plane, v_plane=None):
        grid = self.get_grid(p_plane.shape, p_plane.device)
        parts = (p_plane, v_plane, grid) if self.use_v_plane else (p_plane, grid)
        x = torch.cat(parts, dim=-1).permute(0, 3, 1, 2)        # NHWC -> NCHW (:46-47)
        return self.fno2d(x.contiguous())

    def get_grid(self, shape, device):
        """inclusive linspace(0,1,n) grids, x then y (:51-57); cached on the device instead of
        being rebuilt on the CPU and copied every call."""
        b, sx, sy = shape[0], shape[1], shape[2]
        key = (sx, sy, str(device))
        cache = self.__dict__.setdefault("_grid_cache", {})      # (dropped from whole-module checkpoints)
        g = cache.get(key)
        if g is None:
            gx = torch.linspace(0, 1, sx, dtype=torch.float64).to(torch.float32).reshape(1, sx, 1, 1)
            gy = torch.linspace(0, 1, sy, dtype=torch.float64).to(torch.float32).reshape(1, 1, sy, 1)
            g = torch.cat((gx.expand(1, sx, sy, 1), gy.expand(1, sx, sy, 1)), dim=-1).to(device)
            cache[key] = g
        return g.expand(b, sx, sy, 2)
