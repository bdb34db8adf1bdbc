"""Experiment: forward + backward of BASELINE config 2 as one batch of 64 vs two half batches of 32 (each half's activations,
134 MB, fit the 256 MB Infinity Cache between producer and consumer kernels).  Times with torch events; also 4 quarters."""
import sys, time, torch
sys.path.insert(0, ".")
from pde_policylearning_amd.neuralop.models import FNO2d
from oracle import fno_oracle as O
dev = torch.device("cuda:0")
torch.manual_seed(0)
model = FNO2d(12, 12, 64, in_channels=3, out_channels=1).to(dev)
x = torch.randn(64, 3, 128, 128, device=dev)
t = torch.randn(64, 1, 128, 128, device=dev)
def step(nsplit):
    for p in model.parameters(): p.grad = None
    n = 64 // nsplit
    for i in range(nsplit):
        y = model(x[i * n:(i + 1) * n])
        O.lp_loss_rel_sum(y, t[i * n:(i + 1) * n]).backward()
for nsplit in (1, 2, 4, 1, 2, 4):
    for _ in range(5): step(nsplit)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(200): step(nsplit)
    e1.record(); torch.cuda.synchronize()
    print(f"nsplit {nsplit}: {e0.elapsed_time(e1) / 200:.3f} ms per fwd+bwd of 64 fields", flush=True)
