"""Where a wave of k_proj_fwd_w spends its cycles (-DPFW_TRACE build): load + GELU + split of the column / matrix section of
the chunks (LDS fragment reads, 12 MFMAs, drain) / vector section (bias, GELU, w2 dot).  Usage (GPU box):
   FNO_LIB_PATH=$PWD/tools/exp_pfwtrace.so FNO_EXTRA_FLAGS=-DPFW_TRACE python -m pde_policylearning_amd.build --force
   FNO_LIB_PATH=$PWD/tools/exp_pfwtrace.so python tools/pfw_trace.py"""
import ctypes as C, sys
import numpy as np, torch
sys.path.insert(0, ".")
import bench
from pde_policylearning_amd import _lib
cfg = dict(bench.CONFIGS["fno2d_128x128_w64_m12_b64"])
dev = torch.device("cuda:0")
model, inputs, tgt = bench.make_workload(cfg, 0, dev)
for _ in range(5):
    y = model(*inputs)
torch.cuda.synchronize()
L = _lib.lib()
L.fno_debug_pfw_dump.argtypes = [C.c_void_p, C.c_size_t]
N = 64 * 16 * 4
buf = (C.c_ulonglong * N)()
assert L.fno_debug_pfw_dump(buf, N) == 0
raw = np.frombuffer(buf, np.uint64).reshape(64 * 16, 4)
raw = raw[raw[:, 3] > 0]
print(f"prologue: weight scan {float(((raw[:, 3] >> 16) & 0xffffff).mean()):.0f} cycles, weight split + tables {float((raw[:, 3] >> 40).mean()):.0f} cycles")
a = raw.astype(np.float64)
a[:, 3] = (raw[:, 3] & 0xffff).astype(np.float64)
per = a[:, :3] / a[:, 3:4]
print(f"{len(a)} waves, columns per wave {a[:, 3].mean():.2f}; cycles per column and wave: load+gelu+split {per[:, 0].mean():.0f}, "
      f"matrix sections {per[:, 1].mean():.0f} ({per[:, 1].mean() / 8:.0f} per chunk), vector sections {per[:, 2].mean():.0f} "
      f"({per[:, 2].mean() / 8:.0f} per chunk), total {per.sum(1).mean():.0f}")
