"""Random-shape cross-check of the fused FNO model against the unfused composition (engine spectral conv + torch glue).
GPU box: python tools/fuzz_fused.py [n_cases]"""
import os, sys, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pde_policylearning_amd.neuralop.models import FNO2d, FNO3d

dev = torch.device("cuda", 0)
random.seed(int(os.environ.get("SEED", "0")))
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
worst = 0.0
for case in range(n):
    three_d = random.random() < 0.25
    C = random.choice([32, 64])
    L = random.randint(1, 5)
    cin, cout = random.randint(1, 4), random.choice([1, 1, 2, 3])
    if three_d:
        dims = (random.choice([8, 16]), random.choice([8, 16, 24]), random.choice([32, 64, 48, 40]))
        B = random.randint(1, 2)
    else:
        dims = (random.choice([16, 32, 48, 64, 96, 128, 160]), random.choice([32, 64, 96, 128, 256, 48, 160, 40, 50]))
        B = random.randint(1, 3)
    pw = 1
    for d in dims: pw *= d
    if pw % 128:                      # rows that tile the 128 / 256-pixel tile, and "loose" ones (48, 96, 160, 40, 50 ...)
        continue
    modes = tuple(2 * random.randint(1, min(8, d // 2 - 1 if i < len(dims) - 1 else d // 2)) for i, d in enumerate(dims))
    torch.manual_seed(case)
    ctor = FNO3d if three_d else FNO2d
    model = ctor(*modes, C, in_channels=cin, out_channels=cout, n_layers=L).to(dev)
    x = torch.randn((B, cin) + dims, device=dev)
    dy = torch.randn((B, cout) + dims, device=dev)
    def run(fused):
        for p in model.parameters(): p.grad = None
        if not fused:
            orig = model.fused_supported
            model.fused_supported = lambda *_: False
        try:
            y = model(x); y.backward(dy)
        finally:
            if not fused: model.fused_supported = orig
        return [y.detach().clone()] + [p.grad.clone() for p in model.parameters()]
    try:
        a, b = run(True), run(False)
    except RuntimeError as e:
        print(f"case {case}: dims {dims} C {C} L {L} modes {modes} cin {cin} cout {cout} B {B}: {str(e)[:120]}")
        continue
    errs = [float((u - v).norm() / (v.norm() + 1e-30)) for u, v in zip(a, b)]
    worst = max(worst, errs[0], max(errs[1:]))
    flag = "  <-- MISMATCH" if errs[0] > 1e-5 or max(errs[1:]) > 5e-4 else ""
    print(f"case {case}: dims {dims} C {C} L {L} modes {modes} cin {cin} cout {cout} B {B}: y {errs[0]:.1e} grads max {max(errs[1:]):.1e}{flag}")
print("worst", worst)
