"""The clock the chip holds INSIDE the four hot kernels of the headline step (MI355X_MICROARCH.md, DVFS give-back item 6):
a -DFNO_CLOCK build stamps s_memtime (shader cycles) and s_memrealtime (100 MHz) around each workgroup's tile loop; after
>= 2 s of back-to-back training steps on random data the median over the workgroups of d(cycles) / d(real time) is the
in-kernel clock.  Usage (GPU box):
   FNO_LIB_PATH=$PWD/tools/exp_clock.so FNO_EXTRA_FLAGS=-DFNO_CLOCK python -m pde_policylearning_amd.build --force
   FNO_LIB_PATH=$PWD/tools/exp_clock.so python tools/kernel_clock.py"""
import ctypes as C
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
import bench
from pde_policylearning_amd import _lib
from pde_policylearning_amd.trainer import FlatGradBucket, FusedAdam, FusedLpLoss, train_step

cfg = dict(bench.CONFIGS["fno2d_128x128_w64_m12_b64"])
dev = torch.device("cuda:0")
model, inputs, tgt = bench.make_workload(cfg, 0, dev)
bucket = FlatGradBucket(model.parameters(), direct_module=model)
opt = FusedAdam(bucket, lr=1e-3, weight_decay=1e-4)
loss_fn = FusedLpLoss(size_average=False)
t0 = time.time()
n = 0
while time.time() - t0 < 3.0:
    for _ in range(50):
        train_step(model, bucket, opt, inputs, tgt, loss_fn)
    torch.cuda.synchronize()
    n += 50
L = _lib.lib()
# whole-launch durations (HIP events on the launch stream) of the same process, for the time OUTSIDE the stamped section
L.fno_profile_reset()
L.fno_profile_enable(1)
for _ in range(5):
    train_step(model, bucket, opt, inputs, tgt, loss_fn)
torch.cuda.synchronize()
L.fno_profile_enable(0)
launch_ms = {name: ms / max(cnt, 1) for name, ms, cnt in _lib.profile_summary()}
L.fno_debug_clock_dump.argtypes = [C.c_void_p, C.c_size_t]
N = 4 * 1024 * 8
buf = (C.c_ulonglong * N)()
assert L.fno_debug_clock_dump(buf, N) == 0
a = np.frombuffer(buf, np.uint64).reshape(4, 1024, 8).astype(np.float64)
print("per-launch times by HIP events:", {k: round(v * 1e3, 1) for k, v in launch_ms.items() if v > 0.05})
print(f"{n} steps in {time.time() - t0:.2f} s; in-kernel clock = d(s_memtime) / d(s_memrealtime) x 100 MHz around the tile loop, last launch of each kernel")
for kid, name in enumerate(("k_blk_fwd_t (block forward)", "k_block_bwd_g2 (block backward)", "k_proj_fwd_h2 (projection forward)",
                            "k_proj_bwd_t (projection backward)")):
    r = a[kid]
    ok = (r[:, 3] > r[:, 2]) & (r[:, 1] > r[:, 0])
    if not ok.any():
        print(f"   {name}: no records")
        continue
    ghz = (r[ok, 1] - r[ok, 0]) / (r[ok, 3] - r[ok, 2]) * 0.1
    us = (r[ok, 3] - r[ok, 2]) / 100.0
    span = (r[ok, 3].max() - r[ok, 2].min()) / 100.0
    late = (r[ok, 2] - r[ok, 2].min()) / 100.0
    print(f"   {name:38s} workgroups {int(ok.sum()):4d}   clock median {np.median(ghz):.3f} GHz  (min {ghz.min():.3f}, max {ghz.max():.3f})"
          f"   stamped-section time median {np.median(us):.1f} us; first start to last end {span:.1f} us; "
          f"workgroups starting > 5 us after the first: {int((late > 5).sum())}")
    entry = r[ok, 4]
    print(f"      kernel entry -> stamped section: median {np.median(r[ok, 2] - entry) / 100.0:.1f} us (max {(r[ok, 2] - entry).max() / 100.0:.1f}); "
          f"first entry to last end {(r[ok, 3].max() - entry.min()) / 100.0:.1f} us; entries spread over {(entry.max() - entry.min()) / 100.0:.1f} us; "
          f"end-time quantiles from the first entry [10, 50, 90, 100] %: " + " ".join(f"{np.percentile(r[ok, 3] - entry.min(), q) / 100.0:.1f}" for q in (10, 50, 90, 100)))

    if kid == 2:
        d = (r[ok, 3] - r[ok, 2]) / 100.0
        idx = np.nonzero(ok)[0]
        print("      duration quantiles [0, 10, 50, 90, 100] %:", " ".join(f"{np.percentile(d, q):.1f}" for q in (0, 10, 50, 90, 100)))
        print("      median by blockIdx % 8 (XCD under round-robin):", " ".join(f"{np.median(d[idx % 8 == x]):.1f}" for x in range(8)))
        print("      median of workgroups [0, 256) / [256, 512):", f"{np.median(d[idx < 256]):.1f} / {np.median(d[idx >= 256]):.1f}")
        end = (r[ok, 3] - r[ok, 2].min()) / 100.0
        print("      end time quantiles [0, 10, 50, 90, 100] %:", " ".join(f"{np.percentile(end, q):.1f}" for q in (0, 10, 50, 90, 100)))
