import time, torch, sys
sys.path.insert(0, '.')
from oracle import fno_oracle as O
from tests.test_parity_gpu import _fno_params
p = _fno_params(64, 4, [6, 6])
x = torch.randn(4, 3, 128, 128); t = torch.randn(4, 1, 128, 128)
for nt in (8, 16, 32, 64):
    torch.set_num_threads(nt)
    pc = {k: v.clone().requires_grad_(True) for k, v in p.items()}
    def step():
        y = O.fno_forward(pc, x, (12, 12)); O.lp_loss_rel_sum(y, t).backward()
    step(); t0 = time.perf_counter(); step(); step(); dt = (time.perf_counter() - t0) / 2
    print(f"threads {nt}: {dt:.2f} s/step, {4/dt:.2f} fields/s", flush=True)
