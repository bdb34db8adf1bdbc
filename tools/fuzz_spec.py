"""Random-shape check of the standalone spectral convolution (fno_spec_*: any size, any channel count, three norms,
2-D / 3-D, live last-dim extent smaller than the stored one) against an fp64 torch.fft evaluation on the same GPU.
GPU box: python tools/fuzz_spec.py [n_cases]"""
import itertools, os, random, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pde_policylearning_amd import functional as F

dev = torch.device("cuda", 0)
random.seed(int(os.environ.get("SEED", "0")))
n = int(sys.argv[1]) if len(sys.argv) > 1 else 30


def ref_conv(x, ws, bias, modes, norm, wle):
    """corner order (lo..), (hi..) over the leading dims (itertools.product), last dim [:m_last]; weights stored with last
    extent wle >= modes[-1]."""
    nd = x.dim() - 2
    dims = list(range(2, 2 + nd))
    xf = torch.fft.rfftn(x, dim=dims, norm=norm)
    out = torch.zeros(x.shape[0], ws[0].shape[1], *xf.shape[2:], dtype=xf.dtype, device=x.device)
    sel = [((None, m), (-m, None)) for m in modes[:-1]] + [((None, modes[-1]),)]
    for w, bnd in zip(ws, itertools.product(*sel)):
        idx = (slice(None), slice(None)) + tuple(slice(*b) for b in bnd)
        wc = torch.view_as_complex(w)[..., :modes[-1]]
        out[idx] = torch.einsum("bi...,io...->bo...", xf[idx], wc)
    y = torch.fft.irfftn(out, s=x.shape[2:], dim=dims, norm=norm)
    return y if bias is None else y + bias


worst = 0.0
for case in range(n):
    nd = random.choice([2, 2, 3])
    if nd == 2:
        dims = (random.randint(4, 70), random.randint(4, 140))
    else:
        dims = (random.randint(4, 20), random.randint(4, 24), random.choice([1, 1, 5, 16, 33, 64, 73]))
    cin, cout = random.randint(1, 70), random.randint(1, 70)
    if random.random() < 0.3:
        cin = cout = random.choice([32, 64])
    B = random.randint(1, 3)
    lead = [random.randint(1, max(1, d // 2)) for d in dims[:-1]]
    klive = random.randint(1, dims[-1] // 2 + 1)
    wle = klive + random.choice([0, 0, 3]) if nd == 3 else klive
    modes = tuple(lead + [klive])
    norm = random.choice(["backward", "forward", "ortho"])
    use_bias = random.random() < 0.5
    torch.manual_seed(case)
    x = torch.randn((B, cin) + dims, device=dev, requires_grad=True)
    ws = [(torch.randn((cin, cout) + tuple(lead) + (wle, 2), device=dev) * 0.1).requires_grad_(True) for _ in range(2 ** (nd - 1))]
    bias = (torch.randn((cout,) + (1,) * nd, device=dev) * 0.1).requires_grad_(True) if use_bias else None
    dy = torch.randn((B, cout) + dims, device=dev)
    tag = f"case {case}: {nd}-D dims {dims} {cin}->{cout} B {B} modes {modes} wle {wle} {norm} bias {use_bias}"
    try:
        y = F.spectral_conv(x, ws, bias, modes, norm, weight_last_extent=wle)
        y.backward(dy)
    except RuntimeError as e:
        print(tag, "ENGINE ERROR", str(e)[:140])
        continue
    got = [y.detach().double(), x.grad.double()] + [w.grad.double() for w in ws] + ([bias.grad.double()] if use_bias else [])
    xd = x.detach().double().requires_grad_(True)
    wd = [w.detach().double().requires_grad_(True) for w in ws]
    bd = bias.detach().double().requires_grad_(True) if use_bias else None
    yr = ref_conv(xd, wd, bd, modes, norm, wle)
    yr.backward(dy.double())
    ref = [yr.detach(), xd.grad] + [w.grad for w in wd] + ([bd.grad] if use_bias else [])
    errs = [float((a - b).norm() / (b.norm() + 1e-300)) for a, b in zip(got, ref)]
    worst = max(worst, max(errs))
    flag = "  <-- MISMATCH" if max(errs) > 2e-5 else ""
    print(tag, " ".join(f"{e:.1e}" for e in errs), flag)
print("worst", worst)
