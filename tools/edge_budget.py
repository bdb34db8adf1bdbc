#!/usr/bin/env python3
"""Per-parameter error vs the float64 oracle for one FNO2d shape in both GEMM modes (GPU box):
   python tools/edge_budget.py C cin cout L S B"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import fno_oracle as O
from oracle.detfill import fill_named
from pde_policylearning_amd import _lib
from tests.test_parity_gpu import _fno_params, _run_fused, _oracle_fno_fp64
from tests.util import rel_l2
C, cin, cout, L, S, B = [int(v) for v in sys.argv[1:7]]
modes = (8, 6)
half = [m // 2 for m in modes]
p = _fno_params(C, L, half, cin=cin, cout=cout, seed_tag="e")
x = torch.from_numpy(fill_named("xe", (B, cin, S, S), 1.0))
tgt = torch.from_numpy(fill_named("te", (B, cout, S, S), 1.0))
y64, g64 = _oracle_fno_fp64(p, x, tgt, modes, L)
pc = {k: v.clone().requires_grad_(True) for k, v in p.items()}
yc = O.fno_forward(pc, x, modes, n_layers=L)
O.lp_loss_rel_sum(yc, tgt).backward()
dev = torch.device("cuda:0")
res = {}
for mode in (1, 0):
    _lib.lib().fno_set_gemm_mode(mode)
    y, pg = _run_fused(p, x, modes, dev, n_layers=L)
    O.lp_loss_rel_sum(y, tgt.to(dev)).backward()
    res[mode] = {k: rel_l2(pg[k].grad.cpu().numpy(), g64[k]) for k in p}
    res[mode]["y"] = rel_l2(y.detach().cpu().numpy(), y64)
print(f"{'tensor':44s} {'oracle fp32':>12s} {'engine x3':>12s} {'engine f32':>12s} {'|g|':>10s}")
print(f"{'y':44s} {rel_l2(yc.detach().numpy(), y64):12.2e} {res[1]['y']:12.2e} {res[0]['y']:12.2e}")
for k in p:
    print(f"{k:44s} {rel_l2(pc[k].grad.numpy(), g64[k]):12.2e} {res[1][k]:12.2e} {res[0][k]:12.2e} {np.linalg.norm(g64[k]):10.2e}")
