import sys, torch, numpy as np
sys.path.insert(0, ".")
from pde_policylearning_amd.neuralop.models import FNO2d
torch.manual_seed(0)
dev = torch.device("cuda:0")
m = FNO2d(12, 12, 64, in_channels=3, out_channels=1).to(dev)
g = torch.Generator().manual_seed(1)
x = torch.randn(64, 3, 128, 128, generator=g).to(dev)
for B in (2, 8):
    y = m(x[:B])
    saved = y.grad_fn.saved_tensors[1]
    sf = saved.view(torch.float32)
    n_act = B * 64 * 128 * 128
    amax = sf[-64:].cpu().numpy()
    print("B", B, "max|x|", float(x[:B].abs().max()), "amax[7]", amax[7], "ubound[8]", amax[8])
    for l in range(1, 5):
        u = sf[l * n_act:(l + 1) * n_act]
        print("  l", l, "true max|u_l|", float(u.abs().max()), "published", amax[8 + l], " per-sample max:", [round(float(v), 3) for v in u.view(B, -1).abs().max(dim=1).values[:8]])
