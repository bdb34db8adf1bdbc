"""Per-kernel time of the standalone spectral-conv path (GPU box): python tools/spec_profile.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pde_policylearning_amd import functional as F, _lib
dev = torch.device("cuda", 0)
def run(name, shape, modes, norm, wle=None, live=None):
    C = shape[1]; nd = len(shape) - 2
    x = torch.randn(*shape, device=dev, requires_grad=True)
    ws = [(torch.randn(C, C, *modes, 2, device=dev) * 0.01).requires_grad_(True) for _ in range(2 ** (nd - 1))]
    dy = torch.randn(*shape, device=dev)
    L = _lib.lib()
    for it in range(3):
        if it == 2: L.fno_profile_reset(); L.fno_profile_enable(1)
        y = F.spectral_conv(x, ws, None, live or modes, norm=norm, weight_last_extent=wle); y.backward(dy)
    torch.cuda.synchronize(); L.fno_profile_enable(0)
    print(name)
    for n, ms, k in _lib.profile_summary(): print(f"   {n:24s} {ms/k*1e3:9.1f} us x{k}")
    L.fno_profile_reset()
only = os.environ.get("SPEC_ONLY", "")
if only in ("", "rno"):
    run("RNO 32x64x128x128 m12", (32, 64, 128, 128), (12, 12), "ortho")
if only in ("", "pino2d"):
    run("PINO2d 1x64x128x128x73 m8", (1, 64, 128, 128, 73), (8, 8, 8), "backward", 8, (8, 8, 8))
if only in ("", "pinoff"):
    run("PINO fullfield 32x64x32x32x1 m12", (32, 64, 32, 32, 1), (12, 12, 12), "backward", 12, (12, 12, 1))
if os.environ.get("ROW_ALIGN_PROBE"):
    run("probe W=80 (row 320 B, 128 B-aligned tiles)", (1, 64, 128, 128, 80), (8, 8, 8), "backward", 8, (8, 8, 8))
    run("probe W=72 (row 288 B)", (1, 64, 128, 128, 72), (8, 8, 8), "backward", 8, (8, 8, 8))
    run("probe W=65", (1, 64, 128, 128, 65), (8, 8, 8), "backward", 8, (8, 8, 8))
