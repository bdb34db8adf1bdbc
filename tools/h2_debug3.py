import os, sys, subprocess, torch, numpy as np
sys.path.insert(0, ".")
if len(sys.argv) > 1:
    from pde_policylearning_amd.neuralop.models import FNO2d
    torch.manual_seed(0)
    dev = torch.device("cuda:0")
    m = FNO2d(12, 12, 64, in_channels=3, out_channels=1).to(dev)
    g = torch.Generator().manual_seed(1)
    x = torch.randn(64, 3, 128, 128, generator=g).to(dev)
    B = 8
    if os.environ.get("H2DBG_PERM"):      # the same eight fields in another order: do wrong patches follow the data or the position?
        x = x[[2, 3, 0, 1, 6, 7, 4, 5] + list(range(8, 64))]
    y = m(x[:B])
    sf = y.grad_fn.saved_tensors[1].view(torch.float32)
    n_act = B * 64 * 128 * 128
    out = {l: sf[l * n_act:(l + 1) * n_act].view(B, 64, 128 * 128).cpu().numpy() for l in range(1, 5)}
    out["y"] = y.detach().cpu().numpy()
    np.save(sys.argv[1], np.array([out], dtype=object), allow_pickle=True)
else:
    res = {}
    for tag, env in (("ref", {"FNO_NO_H2": "1"}), ("all", {})):
        subprocess.check_call([sys.executable, __file__, f"/tmp/h2dbg3_{tag}.npy"], env=dict(os.environ, **env))
        res[tag] = np.load(f"/tmp/h2dbg3_{tag}.npy", allow_pickle=True)[0]
    for l in range(1, 5):
        a, b = res["all"][l], res["ref"][l]
        per = [float(np.linalg.norm(a[i] - b[i]) / np.linalg.norm(b[i])) for i in range(a.shape[0])]
        print("u_%d per-sample rel diff:" % l, " ".join(f"{v:.1e}" for v in per))
        if max(per) > 1e-5:
            i = int(np.argmax(per)); d = np.abs(a[i] - b[i])
            c, px = np.unravel_index(np.argmax(d), d.shape)
            print("   worst sample", i, "channel", c, "pixel", px, "tile", px // 128, "val", a[i][c, px], "ref", b[i][c, px])
            bad = (d > 1e-4 * np.abs(b[i]).max())
            cc, pp = np.where(bad)
            print("   bad box: channels", cc.min(), "-", cc.max(), " pixels-in-tile", (pp % 128).min(), "-", (pp % 128).max(), " distinct channels", len(set(cc.tolist())), "distinct px", len(set((pp % 128).tolist())))
            print("   bad elements", int(bad.sum()), "channels", sorted(set(np.where(bad)[0]))[:12], "tiles", sorted(set((np.where(bad)[1] // 128).tolist()))[:12])
            break
