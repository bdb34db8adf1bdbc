"""Random-shape checks of the smaller entry points against fp64 torch on the same GPU: pointwise conv + add, projection
head, lifting, RNO gates, fused loss / Adam, PINO residual loss.  GPU box: python tools/fuzz_misc.py [n_cases]"""
import math, os, random, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pde_policylearning_amd import functional as F

dev = torch.device("cuda", 0)
random.seed(int(os.environ.get("SEED", "0")))
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10
worst = {}


def rel(a, b):
    return float((a.double() - b).norm() / (b.norm() + 1e-300))


def check(kind, tag, got, ref, tol):
    errs = [rel(a, b) for a, b in zip(got, ref)]
    worst[kind] = max(worst.get(kind, 0.0), max(errs))
    if max(errs) > tol:
        print(kind, tag, " ".join(f"{e:.1e}" for e in errs), "<-- MISMATCH")


def shape_with_plane_multiple_of_128():
    while True:
        dims = tuple(random.choice([1, 2, 4, 8, 16, 32, 64, 73, 65, 24]) for _ in range(random.randint(1, 3)))
        pw = math.prod(dims)
        if pw % 128 == 0 and pw <= 1 << 18:
            return dims


for case in range(n):
    torch.manual_seed(1000 + case)
    C = random.choice([32, 64])
    B = random.randint(1, 3)
    dims = shape_with_plane_multiple_of_128()
    x = torch.randn((B, C) + dims, device=dev)
    # ---- pointwise conv + bias + add ----
    w, b, add = torch.randn(C, C, 1, device=dev) * 0.1, torch.randn(C, device=dev) * 0.1, torch.randn_like(x)
    dy = torch.randn_like(x)
    leaves = [t.clone().requires_grad_(True) for t in (x, w, b, add)]
    y = F.pointwise_conv_add(*leaves); y.backward(dy)
    ld = [t.detach().double().requires_grad_(True) for t in (x, w, b, add)]
    yr = torch.einsum("oi,bi...->bo...", ld[1][:, :, 0], ld[0]) + ld[2].view(1, C, *([1] * len(dims))) + ld[3]
    yr.backward(dy.double())
    check("pointwise", f"C{C} B{B} {dims}", [y] + [t.grad for t in leaves], [yr.detach()] + [t.grad for t in ld], 2e-5)
    # ---- projection head ----
    hid = random.choice([128, 256])
    ps = [torch.randn(hid, C, device=dev) * 0.15, torch.randn(hid, device=dev) * 0.1, torch.randn(1, hid, device=dev) * 0.1,
          torch.randn(1, device=dev) * 0.1]
    dyp = torch.randn((B, 1) + dims, device=dev)
    leaves = [t.clone().requires_grad_(True) for t in [x] + ps]
    y = F.projection_head(*leaves); y.backward(dyp)
    ld = [t.detach().double().requires_grad_(True) for t in [x] + ps]
    h = torch.einsum("hi,bi...->bh...", ld[1], ld[0]) + ld[2].view(1, hid, *([1] * len(dims)))
    yr = torch.einsum("oh,bh...->bo...", ld[3], torch.nn.functional.gelu(h)) + ld[4].view(1, 1, *([1] * len(dims)))
    yr.backward(dyp.double())
    check("projection", f"C{C} hid{hid} B{B} {dims}", [y] + [t.grad for t in leaves], [yr.detach()] + [t.grad for t in ld], 1e-4)
    # ---- lifting ----
    cin = random.randint(1, 4)
    xi = torch.randn((B, cin) + dims, device=dev)
    wl, bl = torch.randn(C, cin, device=dev) * 0.3, torch.randn(C, device=dev) * 0.1
    le = [wl.clone().requires_grad_(True), bl.clone().requires_grad_(True)]
    y = F.lifting(xi, *le); y.backward(dy)
    ldd = [wl.double().requires_grad_(True), bl.double().requires_grad_(True)]
    yr = torch.einsum("oi,bi...->bo...", ldd[0], xi.double()) + ldd[1].view(1, C, *([1] * len(dims)))
    yr.backward(dy.double())
    check("lifting", f"{cin}->{C} B{B} {dims}", [y] + [t.grad for t in le], [yr.detach()] + [t.grad for t in ldd], 2e-5)
    # ---- fused relative L2 loss with decode ----
    S = random.choice([7, 16, 40])
    pred, tgt = torch.randn(B + 1, S, S, device=dev), torch.randn(B + 1, S, S, device=dev)
    mean, std = torch.randn(S, S, device=dev), torch.rand(S, S, device=dev) + 0.5
    pl = pred.clone().requires_grad_(True)
    loss = F.lp_loss_rel(pl, tgt, mean, std, 1e-5, size_average=bool(case % 2)); loss.backward()
    pd = pred.double().requires_grad_(True)
    dec = lambda t: t * (std.double() + 1e-5) + mean.double()
    d = torch.norm((dec(pd) - dec(tgt.double())).reshape(B + 1, -1), 2, 1) / torch.norm(dec(tgt.double()).reshape(B + 1, -1), 2, 1)
    lr = d.mean() if case % 2 else d.sum()
    lr.backward()
    check("lploss", f"B{B + 1} S{S}", [loss.detach().reshape(1), pl.grad], [lr.detach().reshape(1), pd.grad], 2e-5)
    # ---- PINO residual loss ----
    nn_, nt = random.choice([32, 64]), random.randint(3, 7)
    u = torch.randn(B, nn_, nn_, nt, device=dev)
    u0 = torch.randn(B, nn_, nn_, device=dev)
    visc = torch.rand(B, device=dev) * 0.02 + 0.002
    yy = torch.arange(nn_, device=dev, dtype=torch.float32) * (2 * math.pi / nn_)
    f = (-4 * torch.cos(4 * yy)).reshape(1, 1, nn_, 1).repeat(1, nn_, 1, 1)
    ul = u.clone().requires_grad_(True)
    lic, lf = F.pino_loss(ul, u0, f, visc, 0.7); (2 * lic + lf).backward()
    ud = u.double().requires_grad_(True)
    k = torch.cat((torch.arange(0, nn_ // 2, device=dev), torch.arange(-(nn_ // 2), 0, device=dev))).double()
    kx, ky = k.reshape(1, nn_, 1, 1), k.reshape(1, 1, nn_, 1)
    lap = (kx ** 2 + ky ** 2).clone(); lap[0, 0, 0, 0] = 1.0
    w_h = torch.fft.fft2(ud, dim=[1, 2]); psi = w_h / lap
    back = lambda sp: torch.fft.irfft2(sp[:, :, :nn_ // 2 + 1], dim=[1, 2])
    du = (ud[..., 2:] - ud[..., :-2]) / (2 * 0.7 / (nt - 1)) + (back(1j * ky * psi) * back(1j * kx * w_h) + back(-1j * kx * psi) * back(1j * ky * w_h)
                                                        - visc.double().reshape(B, 1, 1, 1) * back(-lap * w_h))[..., 1:-1]
    relm = lambda a, b: torch.mean(torch.norm((a - b).reshape(B, -1), 2, 1) / torch.norm(b.reshape(B, -1), 2, 1))
    ric, rf = relm(ud[..., 0], u0.double()), relm(du, f.double().repeat(B, 1, 1, nt - 2))
    (2 * ric + rf).backward()
    check("pino_loss", f"B{B} n{nn_} nt{nt}", [lic.detach().reshape(1), lf.detach().reshape(1), ul.grad],
          [ric.detach().reshape(1), rf.detach().reshape(1), ud.grad], 2e-5)
print("worst per kind:", {k: f"{v:.1e}" for k, v in worst.items()})
