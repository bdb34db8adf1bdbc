"""How many ReLU mask decisions of the RNO head's backward differ from float64?  With w2 = 1 and dy = 1 the bias gradient of
the hidden layer, db1[h] = sum_px [P1[h, px] > 0], is an INTEGER count per hidden unit (exact in float32 below 2^24), so its
difference from the float64 count is the net number of flipped decisions; sum |diff| over units bounds the flips from below.
Compared: the engine's projection backward (the mode given by the environment: default two fp16 terms where the bounds
exist, FNO_NO_H2=1 three bf16 terms, FNO_GEMM_F32=1 fp32 MFMA) and torch float32 on the CPU.  Usage (GPU box):
   [FNO_NO_H2=1 | FNO_GEMM_F32=1] python tools/relu_flip_check.py"""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle.detfill import fill_named
from pde_policylearning_amd import functional as F

C, hid, shape = 64, 256, (32, 64, 128, 128)
x = torch.from_numpy(fill_named("rfx", shape, 1.0))
w1 = torch.from_numpy(fill_named("rfw1", (hid, C), 0.15))
b1 = torch.from_numpy(fill_named("rfb1", (hid,), 0.1))
w2 = torch.ones(1, hid)
b2 = torch.zeros(1)
torch.set_num_threads(min(torch.get_num_threads(), 16))
def counts(dtype):
    p1 = (x.to(dtype).movedim(1, -1).reshape(-1, C) @ w1.to(dtype).t() + b1.to(dtype))
    return (p1 > 0).sum(0).double().numpy(), p1
c64, p64 = counts(torch.float64)
c32, _ = counts(torch.float32)
near = [(p64.abs() < t).sum().item() for t in (1e-7, 3e-7, 1e-6)]
dev = torch.device("cuda:0")
eng = [t.to(dev).requires_grad_(True) for t in (x, w1, b1, w2, b2)]
y = F.projection_head(*eng, act="relu")
y.backward(torch.ones_like(y))
ce = eng[2].grad.double().cpu().numpy()
n = p64.numel()
print(f"{n} decisions; |P1| < 1e-7 / 3e-7 / 1e-6 (float64): {near}")
print(f"torch float32 (CPU): sum |count - count64| = {np.abs(c32 - c64).sum():.0f}, net {np.sum(c32 - c64):+.0f}")
print(f"engine ({'fp32 MFMA' if os.environ.get('FNO_GEMM_F32') else 'bf16 x 3' if os.environ.get('FNO_NO_H2') else 'default'}): "
      f"sum |count - count64| = {np.abs(ce - c64).sum():.0f}, net {np.sum(ce - c64):+.0f}; non-integer part max {np.abs(ce - np.round(ce)).max():.2e}")
