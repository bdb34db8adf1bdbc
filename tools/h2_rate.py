"""Failure RATE of the two-term fp16 block forward under the packed-fp32 op_sel hazard: repeats the FNO2d
forward pass and counts the repetitions whose saved activations differ from an element-wise median reference, and how many
tiles deviate.  The shipped library gives 0; a build with the hazardous form spelled out gives 60 of 60
(profiles/r04_h2_block_forward_failure_rates.txt):
   FNO_LIB_PATH=$PWD/tools/exp_v6.so FNO_EXTRA_FLAGS=-DFNO_SPLIT2_VARIANT=6 python -m pde_policylearning_amd.build --force
Usage (GPU box): [FNO_LIB_PATH=$PWD/tools/exp_v6.so] python tools/h2_rate.py [reps] [batch]"""
import os, sys, torch
sys.path.insert(0, ".")
from pde_policylearning_amd.neuralop.models import FNO2d
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 60
B = int(sys.argv[2]) if len(sys.argv) > 2 else 24
torch.manual_seed(0)
dev = torch.device("cuda:0")
m = FNO2d(12, 12, 64, in_channels=3, out_channels=1).to(dev)
x = torch.randn(B, 3, 128, 128, generator=torch.Generator().manual_seed(1)).to(dev)
n_act = B * 64 * 128 * 128
def run():
    y = m(x)
    return y.grad_fn.saved_tensors[1].view(torch.float32)[:5 * n_act].view(5, B, 64, 128, 128).clone()
ref = torch.stack([run() for _ in range(3)]).median(dim=0).values
scale = ref.abs().amax(dim=(1, 2, 3, 4), keepdim=True)
bad_reps = bad_tiles = 0
for rep in range(reps):
    d = ((run() - ref).abs() > 1e-4 * scale)
    n = int(d.any(dim=2).any(dim=-1).sum())       # (layer, sample, row = tile) triples with a deviation
    bad_reps += n > 0
    bad_tiles += n
print(f"{os.environ.get('FNO_LIB_PATH', 'main').split('/')[-1]}: {bad_reps} of {reps} repetitions deviate, {bad_tiles} (layer, tile) deviations in all")
