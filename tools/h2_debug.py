import os, sys, subprocess, torch, numpy as np
sys.path.insert(0, ".")
if len(sys.argv) > 1:
    from pde_policylearning_amd.neuralop.models import FNO2d
    torch.manual_seed(0)
    dev = torch.device("cuda:0")
    m = FNO2d(12, 12, 64, in_channels=3, out_channels=1).to(dev)
    g = torch.Generator().manual_seed(1)
    x = torch.randn(64, 3, 128, 128, generator=g).to(dev)
    outs = {}
    for B in (1, 2, 8, 64):
        y = m(x[:B])
        y.square().sum().backward()
        outs[B] = (y.detach().cpu().numpy(), {k: p.grad.detach().cpu().numpy().copy() for k, p in m.named_parameters()})
        m.zero_grad(set_to_none=True)
    np.save(sys.argv[1], np.array([outs], dtype=object), allow_pickle=True)
else:
    res = {}
    for tag, env in (("ref", {"FNO_NO_H2": "1"}), ("proj", {"FNO_NO_H2_BLOCKS": "1"}), ("all", {})):
        subprocess.check_call([sys.executable, __file__, f"/tmp/h2dbg_{tag}.npy"], env=dict(os.environ, **env))
        res[tag] = np.load(f"/tmp/h2dbg_{tag}.npy", allow_pickle=True)[0]
    rel = lambda a, b: float(np.linalg.norm(a - b) / np.linalg.norm(b))
    for tag in ("proj", "all"):
        for B in (1, 2, 8, 64):
            y, g = res[tag][B]; yr, gr = res["ref"][B]
            worst = max((rel(g[k], gr[k]), k) for k in g)
            print(f"{tag:5s} B={B:3d}: y vs bf16x3 {rel(y, yr):.2e}   y[0] {rel(y[:1], yr[:1]):.2e}  worst grad {worst[0]:.2e} ({worst[1]})")
