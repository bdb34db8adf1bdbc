"""Random block stacks (fno_blocks: fused Fourier layers with optional GELU, 2-D / 3-D, three norms, input gradient) against
the unfused composition (fno_spec_* + fno_pointwise_* + torch GELU).  GPU box: python tools/fuzz_blocks.py [n]"""
import os, random, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pde_policylearning_amd import functional as F

dev = torch.device("cuda", 0)
random.seed(int(os.environ.get("SEED", "0")))
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
worst = 0.0
done = 0
while done < n:
    nd = random.choice([2, 2, 3])
    C = random.choice([32, 64])
    L = random.randint(1, 4)
    B = random.randint(1, 2)
    # last dims that tile the 128-pixel workgroup tile, and "loose" ones (33..96, not multiples of 32: spectral rows gathered per tile)
    last2 = random.choice([32, 64, 128, 33, 40, 73, 96 - 1, 50])
    last3 = random.choice([32, 64, 40, 73, 36])
    dims = (random.choice([8, 16, 32, 64, 128]), last2) if nd == 2 else (random.choice([4, 8, 16]), random.choice([8, 16]), last3)
    pw = 1
    for d in dims: pw *= d
    if pw % 128 or (dims[-1] % 32 == 0 and 128 % dims[-1]):
        continue
    modes = tuple(random.randint(1, min(8, d // 2)) for d in dims[:-1]) + (random.randint(1, min(12, dims[-1] // 2)),)
    norm = random.choice(["backward", "forward", "ortho"])
    gelu_mask = random.randrange(1 << (L - 1)) if L > 1 else 0          # never after the last layer
    if not F.blocks_supported(torch.empty((B, C) + dims, device=dev), L, modes, norm, gelu_mask):
        continue
    done += 1
    torch.manual_seed(done)
    nc = 2 ** (nd - 1)
    x = torch.randn((B, C) + dims, device=dev)
    skips = [torch.randn(C, C, 1, device=dev) * 0.1 for _ in range(L)]
    specs = [torch.randn((C, C) + modes + (2,), device=dev) * 0.03 for _ in range(nc * L)]
    bias = torch.randn(L, C, device=dev) * 0.1
    dy = torch.randn((B, C) + dims, device=dev)
    a = [t.clone().requires_grad_(True) for t in [x] + skips + specs + [bias]]
    y = F.fno_blocks(a[0], a[1:1 + L], a[1 + L:1 + L + nc * L], a[-1], modes, norm, gelu_mask)
    y.backward(dy)
    b = [t.clone().requires_grad_(True) for t in [x] + skips + specs + [bias]]
    h = b[0]
    for l in range(L):
        sp = F.spectral_conv(h, b[1 + L + nc * l:1 + L + nc * (l + 1)], None, modes, norm)
        h = F.pointwise_conv_add(h, b[1 + l], b[-1][l], sp)
        if (gelu_mask >> l) & 1:
            h = torch.nn.functional.gelu(h)
    h.backward(dy)
    errs = [float((y - h).norm() / h.norm())] + [float((u.grad - v.grad).norm() / (v.grad.norm() + 1e-30)) for u, v in zip(a, b)]
    worst = max(worst, max(errs))
    flag = "  <-- MISMATCH" if errs[0] > 1e-5 or max(errs[1:]) > 5e-4 else ""
    print(f"{nd}-D {dims} C{C} L{L} B{B} modes {modes} {norm} gelu {gelu_mask:b}: y {errs[0]:.1e} grads max {max(errs[1:]):.1e}{flag}")
print("worst", worst)
