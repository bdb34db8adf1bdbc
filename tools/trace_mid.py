"""Shader-clock stamps of one workgroup of k_spec_mid (debug build: FNO_EXTRA_FLAGS="-DFNO_TRACE -DFNO_TRACE_WHICH=3").
Prints per wave the cycles between the stamps: phase-1 sweep | barrier | reduce | contraction | barrier | reduce | phase 3."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pde_policylearning_amd import _lib
from pde_policylearning_amd.neuralop.models import FNO2d
dev = torch.device("cuda", 0)
torch.manual_seed(0)
model = FNO2d(12, 12, 64, in_channels=3, out_channels=1).to(dev)
x = torch.randn(64, 3, 128, 128, device=dev)
for _ in range(3):
    with torch.no_grad():
        model(x)
torch.cuda.synchronize()
L = _lib.lib()
buf = (C.c_ulonglong * (16 * 256))()
L.fno_debug_trace_dump.argtypes = [C.c_void_p, C.c_size_t]
assert L.fno_debug_trace_dump(buf, 16 * 256) == 0
t0 = min(buf[w * 256] for w in range(8))
for w in range(8):
    st = [buf[w * 256 + i] for i in range(8)]
    print(f"wave {w}: start +{st[0] - t0:5d} total {st[7] - st[0]:6d}  " + " ".join(f"{st[i + 1] - st[i]:6d}" for i in range(7)))
# start / end of workgroup (0, y) for every sample y, relative to the earliest start
se = [(buf[8 * 256 + 2 * y], buf[8 * 256 + 2 * y + 1]) for y in range(64)]
t0 = min(a for a, b in se)
print("workgroup (0, y): start / end  " + " ".join(f"{a - t0}/{b - t0}" for a, b in se[::4]))

d = sorted(b - a for a, b in se)
print("durations of workgroup (0, y), cycles: min %d  median %d  max %d   (n = %d)" % (d[0], d[len(d) // 2], d[-1], len(d)))

# every sample's workgroup (1, y), wave 0: cycles per phase, sorted by total (the workgroups that share a CU are the slow ones)
rows = []
for y in range(64):
    st = [buf[10 * 256 + y * 8 + i] for i in range(8)]
    if st[7] > st[0] > 0:
        rows.append((st[7] - st[0], [st[i + 1] - st[i] for i in range(7)]))
rows.sort()
for name, sel in (("fastest quarter", rows[:len(rows) // 4]), ("slowest quarter", rows[-(len(rows) // 4):])):
    if sel:
        n = len(sel)
        print(f"workgroups (1, y), {name} (n = {n}): total {sum(r[0] for r in sel) // n:6d}  phases " +
              " ".join(f"{sum(r[1][i] for r in sel) // n:6d}" for i in range(7)))
