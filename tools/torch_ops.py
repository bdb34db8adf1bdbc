"""Which torch (non-engine) GPU kernels an observer training step still launches, by aten op and input shape.
usage (GPU box): python tools/torch_ops.py [rno2d|rno2d_shipped|pino_ff|pino2d]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity
from pde_policylearning_amd import trainer
from pde_policylearning_amd.libs.models.rno_models import RNO2dObserver
kind = sys.argv[1] if len(sys.argv) > 1 else "rno2d"
dev = torch.device("cuda", 0)
torch.manual_seed(0)
gen = torch.Generator().manual_seed(0)
if kind.startswith("rno2d"):
    width, (X, Y), B = (64, (128, 128), 32) if kind == "rno2d" else (34, (32, 32), 32)
    model = RNO2dObserver(12, 12, width, 0, layer_num=1).to(dev)          # configs/matlab_rno.yaml:67,80-82 (bench.py's workload)
    inputs = (torch.randn(B, 1, X, Y, 1, device=dev),)
    tgt = torch.randn(B, X, Y, 1, device=dev)
else:
    from pde_policylearning_amd.libs.models.pino_models import PINObserver2d, PINObserverFullField
    if kind == "pino_ff":                                                  # bench.py: pino_fullfield_32x32_w64_m12_b32
        B = 32
        model = PINObserverFullField(plane_num=3, modes1=[12] * 4, modes2=[12] * 4, modes3=[12] * 4, fc_dim=128,
                                     layers=[64] * 5, in_dim=1, out_dim=1, act="gelu", pad_ratio=[0.0, 0.0625]).to(dev)
        x = torch.randn(B, 32, 32, 1, 1, device=dev)
    else:                                                                  # bench.py: pinobserver2d_128x128x65_w64_m8_b2
        B = 2
        model = PINObserver2d(modes1=[8] * 4, modes2=[8] * 4, modes3=[8] * 4, fc_dim=128, layers=[64] * 5, in_dim=4,
                              out_dim=1, act="gelu", pad_ratio=0.0625).to(dev)
        x = torch.randn(B, 128, 128, 65, 4, device=dev)
    re = (torch.rand(B, 1) * 100 + 100).to(dev)
    inputs = (x, re)
    with torch.no_grad():
        tgt = torch.randn(model(*inputs).shape, device=dev)
bucket = trainer.FlatGradBucket(model.parameters(), direct_module=model, zero_all=kind.startswith("rno2d"))      # as bench.py
opt = trainer.FusedAdam(bucket, lr=1e-3, weight_decay=1e-4)
loss_fn = trainer.FusedLpLoss(size_average=False)
step = lambda: trainer.train_step(model, bucket, opt, inputs, tgt, loss_fn)
for _ in range(3):
    step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    for _ in range(3):
        step()
    torch.cuda.synchronize()
rows = []
for e in prof.key_averages(group_by_input_shape=True):
    t = getattr(e, "self_device_time_total", None)
    if t is None:
        t = getattr(e, "self_cuda_time_total", 0)
    k = e.key
    if t > 0 and not (k.startswith("k_") or k.startswith("void k_") or k.startswith("_") or k.startswith("void at::") or k.startswith("__amd")
                      or k.startswith("Memcpy") or k.startswith("Memset") or k.startswith("void (anonymous")):
        rows.append((t / 3, e.count / 3, e.key, str(e.input_shapes)[:110]))
rows.sort(reverse=True)
tot = sum(r[0] for r in rows)
print(f"non-engine GPU time per step: {tot:.0f} us")
for t, n, k, sh in rows[:60]:
    print(f"{t:8.1f} us {n:6.1f}x {k:45s} {sh}")
