"""Where the bench step's wall time goes outside the engine kernels (GPU box only)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pde_policylearning_amd.neuralop.models import FNO2d
from pde_policylearning_amd.trainer import FlatGradBucket, LpLoss, train_step

dev = torch.device("cuda", 0)
torch.manual_seed(0)
model = FNO2d(12, 12, 64, in_channels=3, out_channels=1).to(dev)
B = 64
x = torch.randn(B, 3, 128, 128, device=dev)
tgt = torch.randn(B, 1, 128, 128, device=dev)
bucket = FlatGradBucket(model.parameters(), direct_module=model)
opt = torch.optim.Adam(model.parameters(), lr=1e-3, weight_decay=1e-4)
loss_fn = LpLoss(size_average=False)

def timeit(fn, n=30, w=5):
    for _ in range(w): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3

def full(): train_step(model, bucket, opt, (x,), tgt, loss_fn)
def no_opt(): train_step(model, bucket, None, (x,), tgt, loss_fn)
dy = torch.randn(B, 1, 128, 128, device=dev)
def fwd_bwd_only():
    y = model(x); y.backward(dy)
def fwd_only():
    with torch.no_grad(): model(x)
pred = model(x).detach().requires_grad_(True)
def loss_only():
    l = loss_fn(pred, tgt); l.backward()
def adam_only(): opt.step()
for name, fn in (("full step", full), ("no optimizer", no_opt), ("fwd+bwd given dy", fwd_bwd_only), ("fwd only (no_grad)", fwd_only),
                 ("loss fwd+bwd only", loss_only), ("adam only", adam_only)):
    print(f"{name:24s} {timeit(fn):8.3f} ms")
