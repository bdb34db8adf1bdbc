"""Per-phase shader-clock trace of workgroup 0 of a fused kernel (debug build, -DFNO_TRACE).
usage (GPU box): FNO_EXTRA_FLAGS=-DFNO_TRACE python -m pde_policylearning_amd.build --force; python tools/trace_phases.py [layer]
Prints, for waves 0 and 7, the cycles spent between consecutive stamps for tiles 4..7 of the LAST launch
of the traced kernel (forward of one model call; the traced kernel is the last block)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pde_policylearning_amd import _lib
from pde_policylearning_amd.neuralop.models import FNO2d
nphase = int(os.environ.get("NPHASE", "8"))
dev = torch.device("cuda", 0)
torch.manual_seed(0)
model = FNO2d(12, 12, 64, in_channels=3, out_channels=1).to(dev)
x = torch.randn(64, 3, 128, 128, device=dev)
mode = sys.argv[1] if len(sys.argv) > 1 else "fwd"
for _ in range(3):
    if mode == "fwd":
        with torch.no_grad():
            model(x)
    else:
        model(x).sum().backward()
torch.cuda.synchronize()
L = _lib.lib()
buf = (C.c_ulonglong * (16 * 256))()
L.fno_debug_trace_dump.argtypes = [C.c_void_p, C.c_size_t]
assert L.fno_debug_trace_dump(buf, 16 * 256) == 0
for w in [int(v) for v in os.environ.get("WAVES", "0,3").split(",")]:
    row = [buf[w * 256 + i] for i in range(256)]
    print(f"wave {w}:")
    for t in range(4, 10):
        st = row[t * nphase:(t + 1) * nphase + 1]
        if 0 in st: continue
        d = [st[i + 1] - st[i] for i in range(nphase)]
        print(f"  tile {t}: total {st[nphase] - st[0]:6d}  phases " + " ".join(f"{v:6d}" for v in d))
