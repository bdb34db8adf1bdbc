"""Shared helpers for the tests (fixtures -> tensors, error metrics)."""
import os

import numpy as np
import torch

from oracle.detfill import fill_named

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_golden(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    flat = {k: z[k] for k in z.files}
    out, groups = {}, {}
    for k, v in flat.items():
        if "/" in k:
            g, kk = k.split("/", 1)
            groups.setdefault(g, {})[kk] = v
        else:
            out[k] = v
    out.update(groups)
    return out


def rebuild_params(scales, shapes, dtype=np.float32, complex_names=()):
    """{name: torch tensor} = scale * unit_fill(crc32(name)) (see oracle/detfill.py)."""
    p = {}
    for name, sc in scales.items():
        if name not in shapes:
            continue
        shp = tuple(int(s) for s in shapes[name])
        cplx = name in complex_names
        p[name] = torch.from_numpy(fill_named(name, shp, float(sc), dtype=dtype, complex_=cplx))
    return p


def rel_l2(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    den = np.sqrt((b ** 2).sum())
    return float(np.sqrt(((a - b) ** 2).sum()) / (den if den > 0 else 1.0))
