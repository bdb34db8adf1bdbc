"""A/B of the projection backward kernels (k_proj_bwd_q vs k_proj_bwd_t): every gradient of a config-2-shaped FNO2d step from two
processes (FNO_PBWD_Q=1 / unset), compared with each other and timed through the library's per-kernel events.
   python tools/pbq_check.py [batch]"""
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def child(out, B):
    import torch
    from pde_policylearning_amd import _lib
    from pde_policylearning_amd.neuralop.models import FNO2d
    from pde_policylearning_amd.trainer import FusedLpLoss
    torch.manual_seed(0)
    dev = torch.device("cuda:0")
    model = FNO2d(12, 12, 64, in_channels=3, out_channels=1).to(dev)
    g = torch.Generator(device="cpu").manual_seed(5)
    x = torch.randn((B, 3, 128, 128), generator=g).to(dev)
    t = torch.randn((B, 1, 128, 128), generator=g).to(dev)
    loss_fn = FusedLpLoss(size_average=False)
    L = _lib.lib()
    for it in range(3):
        model.zero_grad(set_to_none=True)
        loss = loss_fn(model(x), t)
        loss.backward()
    torch.cuda.synchronize()
    L.fno_profile_reset(); L.fno_profile_enable(1)
    for it in range(5):
        model.zero_grad(set_to_none=True)
        loss = loss_fn(model(x), t)
        loss.backward()
    torch.cuda.synchronize()
    L.fno_profile_enable(0)
    prof = {n: ms / k for n, ms, k in _lib.profile_summary()}
    torch.save({"grads": {n: p.grad.detach().cpu() for n, p in model.named_parameters()}, "loss": float(loss),
                "prof": prof}, out)


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--child":
        child(sys.argv[2], int(sys.argv[3]))
        sys.exit(0)
    import torch
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    res = {}
    with tempfile.TemporaryDirectory() as tmp:
        for arm, env in (("q", {"FNO_PBWD_Q": "1"}), ("t", {})):
            f = os.path.join(tmp, arm + ".pt")
            subprocess.check_call([sys.executable, os.path.abspath(__file__), "--child", f, str(B)], env=dict(os.environ, **env))
            res[arm] = torch.load(f)
    print(f"loss q {res['q']['loss']:.7f}  t {res['t']['loss']:.7f}")
    print(f"k_proj_bwd per launch: q {res['q']['prof'].get('k_proj_bwd', 0):.4f} ms   t {res['t']['prof'].get('k_proj_bwd', 0):.4f} ms")
    worst = 0.0
    for n, gq in res["q"]["grads"].items():
        gt = res["t"]["grads"][n]
        e = float((gq.double() - gt.double()).norm() / gt.double().norm())
        worst = max(worst, e)
        print(f"  {n:45s} rel-L2(q, t) {e:.3e}   |g| {float(gt.norm()):.3e}")
    print("worst", worst)
    sys.exit(0 if worst < 2e-5 else 1)
