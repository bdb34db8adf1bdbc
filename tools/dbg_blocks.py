import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pde_policylearning_amd import functional as F
from pde_policylearning_amd.neuralop.models.rno import FourierLayer2d
dev = torch.device("cuda", 0)
torch.manual_seed(0)
layer = FourierLayer2d(12, 12, 64).to(dev)
x = torch.randn(2, 64, 64, 64, device=dev, requires_grad=True)
dy = torch.randn(2, 64, 64, 64, device=dev)
def run(fused):
    orig = F.blocks_supported
    if not fused: F.blocks_supported = lambda *a, **k: False
    try:
        for p in layer.parameters(): p.grad = None
        x.grad = None
        y = layer(x); y.backward(dy)
    finally:
        F.blocks_supported = orig
    return [y.detach().clone(), x.grad.clone()] + [p.grad.clone() for p in layer.parameters()]
a = run(True); b = run(True); c = run(False)
names = ["y", "dx"] + [n for n, _ in layer.named_parameters()]
for n, u, v, w in zip(names, a, b, c):
    r = lambda p, q: float((p - q).norm() / q.norm())
    print(f"{n:30s} fused-vs-fused {r(u, v):.2e}   fused-vs-unfused {r(u, w):.2e}")
