"""What is wrong inside a bad patch of the two-term fp16 block forward (FNO_H2_FWD_BLOCKS=1)?  The layer's output is
skip GEMM + spectral K-extension + bias.  From the reference run's (correct) input of the first bad layer the skip part is
recomputed on the host in float64; ratio = (bad - skip - bias) / (ref - skip - bias) over the bad elements says what became of
the extension there: 0 = missing, 1 = fine (then the skip part is wrong), anything else = garbage.
Usage (GPU box): python tools/h2_debug4.py   (repeats the bad run until a patch appears, at most 12 times)"""
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
    y = m(x[:B])
    sf = y.grad_fn.saved_tensors[1].view(torch.float32)
    n_act = B * 64 * 128 * 128
    out = {l: sf[l * n_act:(l + 1) * n_act].view(B, 64, 128 * 128).cpu().numpy() for l in range(1, 5)}
    out["skip"] = [w.weight.detach().reshape(64, 64).cpu().numpy() for w in m.fno_blocks.fno_skips]
    out["bias"] = m.fno_blocks.convs.bias.detach().reshape(4, 64).cpu().numpy()
    np.save(sys.argv[1], np.array([out], dtype=object), allow_pickle=True)
else:
    from scipy.special import erf
    run = lambda tag, env: (subprocess.check_call([sys.executable, __file__, f"/tmp/h2dbg4_{tag}.npy"], env=dict(os.environ, **env)),
                            np.load(f"/tmp/h2dbg4_{tag}.npy", allow_pickle=True)[0])[1]
    ref = run("ref", {})
    gelu = lambda v: 0.5 * v * (1 + erf(v / np.sqrt(2)))
    for attempt in range(12):
        bad = run("bad", {"FNO_H2_FWD_BLOCKS": "1"})
        hit = None
        for l in range(1, 5):
            d = np.abs(bad[l] - ref[l])
            if d.max() > 1e-3 * np.abs(ref[l]).max():
                hit = l
                break
        if hit is None:
            print("attempt", attempt, ": no bad patch")
            continue
        l = hit                                   # u_l is the output of block l - 1, its input u_{l-1} (l >= 2 here)
        bsel, csel, psel = np.where(d > 1e-3 * np.abs(ref[l]).max())
        b = bsel[0]
        t0 = psel[bsel == b].min() // 128           # the first bad tile of that sample only
        px = np.unique(psel[(bsel == b) & (psel // 128 == t0)])
        if l >= 2:                                  # everything about that tile's row for offline analysis
            r0 = t0 * 128
            np.savez("gpurun_out/h2_bad_patch.npz", u_in=ref[l - 1][b][:, r0:r0 + 128], ref=ref[l][b][:, r0:r0 + 128],
                     bad=bad[l][b][:, r0:r0 + 128], skip=ref["skip"][l - 1], bias=ref["bias"][l - 1], layer=l, px=px - r0,
                     u_in_prev_tile=ref[l - 1][b][:, max(r0 - 128, 0):max(r0 - 128, 0) + 128],
                     u_in_next_tile=ref[l - 1][b][:, min(r0 + 128, 16384 - 128):min(r0 + 128, 16384 - 128) + 128])
        print(f"attempt {attempt}: first bad tensor u_{l}, sample {b}, {len(px)} pixels {px.min()}..{px.max()} (in tile {px.min() % 128}..{px.max() % 128})")
        if l < 2:
            print("   (block 0 fuses the lifting: not decomposed here)")
            break
        a = ref[l - 1][b][:, px].astype(np.float64)
        if l - 2 < 4 - (l - 2):                   # GELU after block l - 2 (fno_block.py:149)
            a = gelu(a)
        skip = ref["skip"][l - 1].astype(np.float64) @ a + ref["bias"][l - 1].astype(np.float64)[:, None]
        ext_ref = ref[l][b][:, px] - skip
        ext_bad = bad[l][b][:, px] - skip
        ratio = ext_bad / np.where(np.abs(ext_ref) > 1e-6, ext_ref, np.nan)
        print("   |ext_ref| mean %.3e   |ext_bad| mean %.3e   |bad - skip - bias| / |ref - skip - bias| (norms) %.3f" %
              (np.abs(ext_ref).mean(), np.abs(ext_bad).mean(), np.linalg.norm(ext_bad) / np.linalg.norm(ext_ref)))
        print("   ratio ext_bad / ext_ref: median %.3f, 10%% %.3f, 90%% %.3f" % tuple(np.nanpercentile(ratio, [50, 10, 90])))
        # is the bad extension the extension of OTHER pixels?  compare with the reference extension of every 16-pixel group of the row
        row0 = (px.min() // 128) * 128
        ext_row = ref[l][b][:, row0:row0 + 128] - (ref["skip"][l - 1].astype(np.float64) @ (gelu(ref[l - 1][b][:, row0:row0 + 128].astype(np.float64))
                  if l - 2 < 4 - (l - 2) else ref[l - 1][b][:, row0:row0 + 128].astype(np.float64)) + ref["bias"][l - 1].astype(np.float64)[:, None])
        for g0 in range(0, 128, 16):
            if g0 + len(px) <= 128:
                e = np.linalg.norm(ext_bad - ext_row[:, g0:g0 + len(px)]) / np.linalg.norm(ext_bad)
                print(f"      vs reference extension of pixels {g0}..{g0 + len(px) - 1}: rel diff {e:.3f}")
        break
