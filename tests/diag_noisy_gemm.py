"""Diagnostic, not a collected test (run by hand: `python -m tests.diag_noisy_gemm`; lives under tests/ because it calls the oracle).

Which arithmetic owns the ratios between 1.4 and 1.7 of profiles/r06_hostile_errors.txt?  The two cases that produce them
(`target_norm_1e-6`, `tiny`) are run under every arithmetic arm the library has - split-precision channel GEMMs (default), the
exact fp32 matrix instruction for every GEMM (fno_set_gemm_mode(0)), the mode contraction on vector kernels
(fno_set_mode_gemm(0)), the three-launch spectral middle (fno_set_fused_mid(0)) - and the engine's distance from the float64
oracle is printed beside the float32 oracle's for every spectral weight gradient.  A tensor whose ratio does not move between
the arms is not noisy because of a GEMM's split arithmetic."""
import sys
import numpy as np
import torch

from oracle import fno_oracle as O
from tests.test_hostile_ranges_gpu import _hostile
from tests.test_parity_gpu import _run_fused
from tests.util import rel_l2


def main():
    from pde_policylearning_amd import _lib
    L = _lib.lib()
    dev = torch.device("cuda:0")
    B, S, C, NL, modes = 8, 128, 64, 4, (12, 12)
    arms = [("split (default)", 1, 1, 1), ("exact fp32 GEMMs", 0, 1, 1), ("split, vector mode contraction", 1, 0, 1),
            ("split, three-launch middle", 1, 1, 0), ("exact fp32, vector contraction, three launches", 0, 0, 0)]
    for case in ("target_norm_1e-6", "tiny"):
        p, x, tgt = _hostile(case, B, S, C, NL, [m // 2 for m in modes])
        p64 = {k: v.double().clone().requires_grad_(True) for k, v in p.items()}
        O.lp_loss_rel_sum(O.fno_forward(p64, x.double(), modes, n_layers=NL), tgt.double()).backward()
        p32 = {k: v.clone().requires_grad_(True) for k, v in p.items()}
        O.lp_loss_rel_sum(O.fno_forward(p32, x, modes, n_layers=NL), tgt).backward()
        names = [k for k in p if "convs.weight" in k]
        e32 = {k: rel_l2(p32[k].grad.numpy(), p64[k].grad.numpy()) for k in names}
        print(f"== {case}: float32 oracle vs float64, then engine / float32-oracle ratio per arm")
        print("   " + " ".join(f"{k.split('.')[-2]:>8s}" for k in names))
        print("   " + " ".join(f"{e32[k]:8.1e}" for k in names) + "   <- float32 oracle's own relative error")
        for label, gm, mg, fm in arms:
            L.fno_set_gemm_mode(gm)
            L.fno_set_mode_gemm(mg)
            L.fno_set_fused_mid(fm)
            y, pg = _run_fused(p, x, modes, dev, n_layers=NL)
            O.lp_loss_rel_sum(y, tgt.to(dev)).backward()
            torch.cuda.synchronize()
            r = [rel_l2(pg[k].grad.cpu().numpy(), p64[k].grad.numpy()) / max(e32[k], 1e-30) for k in names]
            print("   " + " ".join(f"{v:8.2f}" for v in r) + f"   {label}")
        L.fno_set_gemm_mode(1)
        L.fno_set_mode_gemm(1)
        L.fno_set_fused_mid(1)
    return 0


if __name__ == "__main__":
    sys.exit(main())
