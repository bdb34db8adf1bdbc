#!/usr/bin/env python3
"""Error budget of the fused FNO model against float64 (GPU box): for the three full-model fixtures, per parameter,
the reference's own float32 error vs float64 (tests/golden/*_fp64.npz: ref32_err) next to the engine's error vs float64 in
both GEMM modes.  Usage: python tools/fp64_budget.py > gpurun_out/fp64_budget.txt"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import fno_oracle as O  # noqa: E402
from pde_policylearning_amd import _lib  # noqa: E402
from tests.test_parity_gpu import _run_fused  # noqa: E402
from tests.util import load_golden, rebuild_params, rel_l2  # noqa: E402

dev = torch.device("cuda:0")
L = _lib.lib()
for case, n_modes in [("fno2d_cfg1", (8, 8)), ("fno2d_cfg2small", (12, 12)), ("fno3d_small", (8, 8, 8))]:
    g, g64 = load_golden(case), load_golden(case + "_fp64")
    res = {}
    for mode in (1, 0):
        L.fno_set_gemm_mode(mode)
        p = rebuild_params(g["scales"], g["shapes"])
        y, pg = _run_fused(p, torch.from_numpy(g["x"]), n_modes, dev)
        ey = rel_l2(y.detach().cpu().numpy(), g64["y64"])
        O.lp_loss_rel_sum(y, torch.from_numpy(g["target"]).to(dev)).backward()
        torch.cuda.synchronize()
        eg = {}
        for name, ref in g64["grads64"].items():
            got = pg[name].grad.detach().cpu().numpy().astype(np.float64)
            if ref.shape != got.shape:
                got = got.reshape(-1)[:ref.size]
            eg[name] = rel_l2(got, ref)
        res[mode] = (ey, eg)
    L.fno_set_gemm_mode(1)
    print(f"== {case}: rel-L2 error vs float64 ==")
    print(f"{'tensor':44s} {'reference fp32':>15s} {'engine bf16x3':>15s} {'engine f32':>15s}")
    print(f"{'y':44s} {float(g64['y_ref32_err'][0]):15.2e} {res[1][0]:15.2e} {res[0][0]:15.2e}")
    for name in g64["grads64"]:
        print(f"{'d ' + name:44s} {float(g64['ref32_err'][name][0]):15.2e} {res[1][1][name]:15.2e} {res[0][1][name]:15.2e}")
