#!/usr/bin/env python3
"""Headline benchmark: FNO2d forward+backward fields/sec (BASELINE.json).

Workload (config 2): FNO2d(n_modes 12x12, width 64, in 3, out 1) on 128x128 fields,
batch 64 PER GPU (weak scaling), fp32, synthetic N(0,1) inputs/targets, default init.
One step = the reference training step (run_pde_observers.py:185-193): zero_grad ->
forward -> LpLoss(sum) -> backward -> [N>1: one RCCL all-reduce(SUM) of the flat gradient
bucket] -> Adam.  All arithmetic of forward/backward runs in the HIP engine.

  python bench.py [--gpus N] [--steps K] [--warmup W]
  (N > 1: launched by `python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...`)

Prints ONE JSON line on rank 0 (see README / DESIGN.md section "Measurement").
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PEAK_HBM_GBS = 8000.0        # MI355X HBM3E spec (MI355X_MICROARCH.md)
PEAK_MFMA_F32_TFLOPS = 157.3  # dense fp32 MFMA (v_mfma_f32_32x32x2_f32), same guide
PEAK_MFMA_16BIT_TFLOPS = 2500.0   # dense fp16 / bf16 MFMA, same guide
# A kernel's fp32-equivalent algorithmic FLOP/s are priced against the peak of the pipe its GEMMs RUN on
# (fno_profile_get_terms): fp32 MFMA; two fp16 terms per operand = 3 products per k block on the 16-bit pipe; three bf16 terms
# = 6 products.
PIPE = {1: ("fp32 MFMA", PEAK_MFMA_F32_TFLOPS), 2: ("fp16 MFMA, 3 products per k block", PEAK_MFMA_16BIT_TFLOPS / 3.0),
        3: ("bf16 MFMA, 6 products per k block", PEAK_MFMA_16BIT_TFLOPS / 6.0)}


def algorithmic_bytes_per_field(cfg, model=None):
    """Algorithmic HBM bytes of one training step per field (DESIGN.md section 5 states each model).
    * fused FNO models: SURVEY section 8(d), bytes = 4 HW [(Cin + Cout + (2L + 2) C) + ((Cout + 2C) + 3LC + n_gelu C + (C + Cin))]
      (fwd + bwd, every activation moved the minimum number of times; weight traffic amortised over the batch and excluded);
    * PINO observers (pinobserver.py:192-233, 341-375): the same formula on the padded grid with Cin = in_dim, Cout = output
      planes, L = 4, n_gelu = 3, PLUS 40 bytes per LIVE real spectral-weight element per step divided by the batch (forward
      read, adjoint read, gradient write, Adam p/g/m/v read + p/m/v write): their weights are not amortised (4.2 GB at modes 20);
    * RNO2d (rno.py:231-260, 293-392; one cell, one time step): 61 activation planes-of-C moved per step (20 forward: input
      projection 1, three gates 2 + 3, spectrum of r h 2, candidate + state update 5 + 2, regressor 2 + 2 + 1; 41 backward)
      + 40 bytes per real weight element / batch."""
    hw = 1
    size = cfg["size"]
    if cfg["kind"] in ("pino2d", "pino2d_train"):
        size = size[:2] + (size[2] + 2 * round(size[2] * 0.0625),)      # T padded both ends (pad_ratio 0.0625)
    for n in size:
        hw *= n
    B = cfg["batch"]
    if cfg["kind"] in ("2d", "3d"):
        C, L, cin, cout = cfg["width"], 4, 3, 1
        n_gelu = sum(1 for l in range(L) if l < L - l)
        return 4.0 * hw * ((cin + cout + (2 * L + 2) * C) + ((cout + 2 * C) + 3 * L * C + n_gelu * C + (C + cin)))
    if cfg["kind"].startswith("pino"):
        C, L, n_gelu = 64, 4, 3
        cin, cout = (1, 3) if cfg["kind"].startswith("pino_ff") else (4, 1)
        m = cfg.get("modes", 12 if cfg["kind"].startswith("pino_ff") else 8)
        k3 = min(size[2] // 2 + 1, m) if len(size) > 2 else 1      # live last-dim slices (basics.py:119-139)
        live = L * 4 * C * C * m * m * k3 * 2
        return (4.0 * hw * ((cin + cout + (2 * L + 2) * C) + ((cout + 2 * C) + 3 * L * C + n_gelu * C + (C + cin)))
                + 40.0 * live / B)
    if cfg["kind"].startswith("rno2d"):
        C = 64 if cfg["kind"] == "rno2d" else 34
        n_w = (8 + 2) * 2 * C * C * 12 * 12 * 2 + 10 * C * C + 8 * C * C      # 8 cell + 2 regressor spectral layers, their 1x1 layers, the head
        return 61 * 4.0 * hw * C + 40.0 * n_w / B
    return None

CONFIGS = {
    # name: (ctor args, input shape per GPU)
    "fno2d_128x128_w64_m12_b64": dict(kind="2d", modes=(12, 12), width=64, batch=64, size=(128, 128)),
    "fno2d_64x64_w32_m8_b4": dict(kind="2d", modes=(8, 8), width=32, batch=4, size=(64, 64), graph=True),   # launch-bound
    "fno3d_64_w32_m8_b16": dict(kind="3d", modes=(8, 8, 8), width=32, batch=16, size=(64, 64, 64)),
    # a grid whose rows do not tile the kernels' 128-pixel tile ("loose rows": spectral rows gathered per tile)
    "fno2d_96x96_w64_m12_b64": dict(kind="2d", modes=(12, 12), width=64, batch=64, size=(96, 96)),
    # observer models of BASELINE configs 3 / 5.  Secondary workloads (same roofline / cpu_baseline
    # legs, kernel model at their own batch / plane size).
    "rno2d_128x128_w64_m12_b32": dict(kind="rno2d", batch=32, size=(128, 128)),            # cfg 3 as named (256 / 8 GPUs)
    "rno2d_32x32_w34_m12_b32": dict(kind="rno2d_shipped", batch=32, size=(32, 32), graph=True),   # configs/matlab_rno.yaml values; launch-bound
    # the YAML's active model; launch-bound since its optimizer skips the dead weight slices (round 3): hipGraph replay
    "pino_fullfield_32x32_w64_m12_b32": dict(kind="pino_ff", batch=32, size=(32, 32), graph=True),
    # the same model with the loop's loss: decode + LpLoss on the planes + pde_loss_weight 1.0 * channel-flow term (matlab_rno.yaml:56,62)
    "pino_fullfield_pde_32x130x32_w64_m12_b32": dict(kind="pino_ff_pde", batch=32, size=(32, 32), graph=True),      # launch-bound as well
    "pinobserver2d_128x128x65_w64_m8_b2": dict(kind="pino2d", batch=2, size=(128, 128, 65)),  # configs/pino-observer-finetune-1s.yaml
    # the fine-tuning step of that YAML as train_pino.py runs it: batch 4, loss = 5 * IC + 1 * PDE residual (xy_loss 0)
    "pino_finetune_128x128x65_w64_m8_b4": dict(kind="pino2d_train", batch=4, size=(128, 128, 65)),
    # BASELINE config 5 AS NAMED (256 x 256, width 64, modes 20, PDE-residual loss): one sample per GPU; the residual loss runs
    # through the row / column / row slab kernels (k_pino_loss2.h), the spectral weights are 4 layers x 4 corners x 262 MB
    "pino_finetune_256x256x65_w64_m20_b1": dict(kind="pino2d_train", batch=1, size=(256, 256, 65), modes=20),
}


PMC_TRAFFIC_JSON = "r06_pmc_traffic.json"
PMC_SQ_CSV = "r06_pmc_sq.csv"          # tools/pmc_sq.py summary of the SQ passes (tools/collect_profiles.sh)


# profile label -> the instantiation rocprofv3 --kernel-trace lists for it (headline workload, default GEMM mode)
INSTANTIATION = {
    "fno2d_128x128_w64_m12_b64": {
        "k_block_bwd": "k_block_bwd_g2<false, false, 1, 2, true>",
        "k_block_bwd0": "k_block_bwd_g2<true, false, 1, 2, true>",
        "k_proj_bwd": "k_proj_bwd_t<64, 256, false, 2>",
        "k_proj_fwd": "k_proj_fwd_w<64, 256, 12>",
        "k_pw_fwd_block": "k_blk_fwd_s<true, 2, false> / <true, 1, false> / <false, 0, false> (blocks 1, 2, 3)",
        "k_pw_fwd_block0": "k_blk_fwd_s<false, 2, true>",
        "k_spec_mid": "k_spec_mid<12, 64>",
    },
}


def source_hash():
    """sha1 over the engine's device + ABI sources: ties a committed counter file to the kernels it was collected on
    (the git sha cannot: committing the profile changes it)."""
    import hashlib
    h = hashlib.sha1()
    src = os.path.join(ROOT, "pde_policylearning_amd", "csrc")
    for f in sorted(os.listdir(src)) + ["../../include/fnoengine.h"]:
        with open(os.path.join(src, f), "rb") as fh:
            h.update(f.encode() + b"\0" + fh.read())
    return h.hexdigest()[:16]


def git_sha():
    try:
        import subprocess
        env = {k: v for k, v in os.environ.items() if k not in ("LD_PRELOAD",) and not k.startswith("ROCP")}
        return subprocess.run(["git", "-C", ROOT, "rev-parse", "--short=12", "HEAD"], capture_output=True, text=True,
                              timeout=5, env=env).stdout.strip() or None
    except Exception:
        return None


def kernel_model(cfg):
    """Algorithmic FLOPs and HBM bytes PER LAUNCH of each hot kernel.  The observer workloads run the
    same block / projection kernels on (batch, 64 channels, plane) activations: same formulas, their plane size."""
    B = cfg["batch"]
    C = cfg.get("width", 64)
    size = cfg["size"]
    if cfg["kind"] in ("pino2d", "pino2d_train"):
        size = size[:2] + (size[2] + 2 * round(size[2] * 0.0625),)      # T padded both ends (pad_ratio 0.0625)
    if cfg["kind"] in ("pino_ff", "pino_ff_pde"):
        size = size + (1,)
    PW = 1
    for s in size:
        PW *= s
    HID, CO, CIN = (128 if cfg["kind"].startswith("pino") else 256), 1, 3
    act = 4.0 * B * C * PW                       # bytes of one (B, C, ...) activation
    gemm = 2.0 * B * PW * C * C                  # one 1x1 conv
    proj = 2.0 * B * PW * (C * HID + HID * CO)
    return {
        "k_pw_fwd_lift": dict(bytes=4.0 * B * CIN * PW + act, flops=2.0 * B * PW * CIN * C),
        "k_pw_fwd_block": dict(bytes=2 * act, flops=gemm),
        # block 0 behind a fused lifting layer: u_0 is recomputed from the CIN-channel model input (no read of a 64-channel tensor)
        "k_pw_fwd_block0": dict(bytes=4.0 * B * CIN * PW + act, flops=gemm + 2.0 * B * PW * CIN * C),
        "k_proj_fwd": dict(bytes=act + 4.0 * B * CO * PW, flops=proj),
        "k_proj_bwd": dict(bytes=2 * act + 4.0 * B * CO * PW, flops=2 * proj),
        "k_block_bwd": dict(bytes=3 * act, flops=2 * gemm),
        # block 0 behind a lifting layer: reads dL/du_1 and the model input, writes weight-gradient slabs only
        "k_block_bwd0": dict(bytes=act + 4.0 * B * CIN * PW, flops=gemm + 2.0 * B * PW * CIN * C),
        # Adam on the flat bucket: reads p, g, m, v and writes p, m, v
        "k_adam": dict(bytes=28.0 * cfg.get("n_params", 0), flops=12.0 * cfg.get("n_params", 0)),
    }


def make_workload(cfg, rank, dev, tgt_shape=None):
    """(model, inputs, target) of a workload on `dev`: default init under torch.manual_seed(0) (run_pde_observers.py:25),
    N(0,1) inputs from a per-rank CPU generator.  With `tgt_shape` (the pinned CPU-baseline process: no engine there) the
    target is drawn at that shape instead of at the model output's."""
    import torch
    from pde_policylearning_amd.neuralop.models import FNO2d, FNO3d
    torch.manual_seed(0)
    B = cfg["batch"]
    gen = torch.Generator(device="cpu").manual_seed(1234 + rank)
    if cfg["kind"] in ("2d", "3d"):
        ctor = FNO2d if cfg["kind"] == "2d" else FNO3d
        model = ctor(*cfg["modes"], cfg["width"], in_channels=3, out_channels=1).to(dev)
        x = torch.randn((B, 3) + cfg["size"], generator=gen).to(dev)
        tgt = torch.randn((B, 1) + cfg["size"], generator=gen).to(dev)
        return model, (x,), tgt
    if cfg["kind"].startswith("rno2d"):
        from pde_policylearning_amd.libs.models.rno_models import RNO2dObserver
        width = 64 if cfg["kind"] == "rno2d" else 34
        model = RNO2dObserver(12, 12, width, 0, layer_num=1).to(dev)          # configs/matlab_rno.yaml:67,80-82
        x = torch.randn((B, 1) + cfg["size"] + (1,), generator=gen).to(dev)    # (B, T, X, Y, 1), model_timestep 1
        tgt = torch.randn((B,) + cfg["size"] + (1,), generator=gen).to(dev)
        return model, (x,), tgt
    from pde_policylearning_amd.libs.models.pino_models import PINObserver2d, PINObserverFullField
    if cfg["kind"] == "pino2d_train":   # train_pino.py:79-106; a = (x, y, t, u0) grid as libs/pino_utils/datasets.py:612-617 builds it
        from pde_policylearning_amd.libs.pino_utils.utils import get_grid3d
        S, T = cfg["size"][0], cfg["size"][2]
        m = cfg.get("modes", 8)
        model = PINObserver2d(modes1=[m] * 4, modes2=[m] * 4, modes3=[m] * 4, fc_dim=128, layers=[64] * 5, in_dim=4,
                              out_dim=1, act="gelu", pad_ratio=0.0625).to(dev)
        u0 = torch.randn((B, S, S, 1, 1), generator=gen)
        grid = torch.cat([g[0] for g in get_grid3d(S, T)], dim=-1)
        x = torch.cat((grid.expand(B, -1, -1, -1, -1), u0.repeat(1, 1, 1, T, 1)), dim=-1).to(dev)
    elif cfg["kind"] in ("pino_ff", "pino_ff_pde"):     # run_pde_observers.py:201-207: x (B, X, Y, T=1, 1), re (B, 1)
        model = PINObserverFullField(plane_num=3, modes1=[12] * 4, modes2=[12] * 4, modes3=[12] * 4, fc_dim=128,
                                     layers=[64] * 5, in_dim=1, out_dim=1, act="gelu", pad_ratio=[0.0, 0.0625]).to(dev)
        x = torch.randn((B,) + cfg["size"] + (1, 1), generator=gen).to(dev)
    else:                             # train_pino.py:154-160: x (B, X, Y, T, 4), T padded by round(T * 0.0625)
        model = PINObserver2d(modes1=[8] * 4, modes2=[8] * 4, modes3=[8] * 4, fc_dim=128, layers=[64] * 5, in_dim=4,
                              out_dim=1, act="gelu", pad_ratio=0.0625).to(dev)      # YAML: a float pads both ends, T 65 -> 73
        x = torch.randn((B,) + cfg["size"] + (4,), generator=gen).to(dev)
    re = (torch.rand((B, 1), generator=gen) * 100 + 100).to(dev)
    inputs = (x, re)
    if tgt_shape is None:
        with torch.no_grad():
            tgt_shape = model(*inputs).shape
    tgt = torch.randn(tuple(tgt_shape), generator=gen).to(dev)
    return model, inputs, tgt


def cpu_loss_builder(cfg, pc, cin, ts, bs):
    """The oracle's (CPU restatement of the reference) loss of one step on `bs` fields, as a closure."""
    from oracle import fno_oracle as O
    from oracle import observers_oracle as OO
    kind = cfg["kind"]
    if kind in ("2d", "3d"):
        return lambda: O.lp_loss_rel_sum(O.fno_forward(pc, cin[0], cfg["modes"]), ts)
    if kind.startswith("rno2d"):
        width = 64 if kind == "rno2d" else 34
        return lambda: O.lp_loss_rel_sum(OO.rno2d_forward(pc, cin[0], 12, 12, width, 0, 1), ts)
    if kind in ("pino_ff", "pino_ff_pde"):
        # (the channel-flow physics term of pino_ff_pde is not part of the CPU sample: model + LpLoss only, said in `sample`)
        def f():
            y = OO.pinobserver_fullfield_forward(pc, cin[0], cin[1], [64] * 5, [(12, 12, 12)] * 4, [0.0, 0.0625])
            return O.lp_loss_rel_sum(y.reshape(bs, -1), ts.reshape(bs, -1)) if y.numel() == ts.numel() else y.square().sum()
        return f
    from oracle import pino_loss_oracle as P
    m = cfg.get("modes", 8)
    S = cfg["size"][0]

    def g():
        y = OO.pinobserver2d_forward(pc, cin[0], cin[1], [64] * 5, [(m, m, m)] * 4, [0.0625, 0.0625])
        if kind == "pino2d_train":     # 5 * IC + PDE residual (train_pino.py:86-106)
            lic, lf = P.pino_loss(y.reshape(y.shape[:4]), cin[0][:, :, :, 0, -1], P.forcing(S), 1.0 / cin[1].reshape(bs), 0.5)
            return 5.0 * lic + lf
        return O.lp_loss_rel_sum(y, ts)
    return g


def physical_cores(node=None):
    """One hardware thread per physical core this process may run on, in CPU-number order; node = a NUMA node number
    restricts the list to that node (falls back to all cores when sysfs does not describe it)."""
    allowed = sorted(os.sched_getaffinity(0))

    def parse(txt):
        out = []
        for part in txt.strip().split(","):
            if part:
                a, _, b = part.partition("-")
                out += list(range(int(a), int(b or a) + 1))
        return out
    if node is not None:
        try:
            on_node = set(parse(open(f"/sys/devices/system/node/node{node}/cpulist").read()))
            if on_node & set(allowed):
                allowed = [c for c in allowed if c in on_node]
        except OSError:
            pass
    seen, cores = set(), []
    for c in allowed:
        try:
            sib = tuple(parse(open(f"/sys/devices/system/cpu/cpu{c}/topology/thread_siblings_list").read()))
        except OSError:
            sib = (c,)
        if sib not in seen:
            seen.add(sib)
            cores.append(c)
    return cores


def cpu_child(spec):
    """The CPU-baseline process: pins itself to `threads` physical cores (of NUMA node 0 unless `all_nodes`) BEFORE torch
    creates its thread pool, rebuilds the workload on the CPU, runs `warmups` untimed + up to `iters` timed oracle steps
    (zero_grad + forward + loss + backward) within `budget` seconds and prints one JSON line per timed step (cumulative), so
    that a parent which has to stop it early still has the completed steps."""
    cores = physical_cores(None if spec.get("all_nodes") else 0)
    nthr = min(spec["threads"], len(cores))
    os.sched_setaffinity(0, cores[:nthr])
    os.environ["OMP_NUM_THREADS"] = str(nthr)
    import torch
    torch.set_num_threads(nthr)
    cfg = dict(CONFIGS[spec["config"]], batch=spec["bs"])
    model, inputs, tgt = make_workload(cfg, 0, "cpu", tgt_shape=spec["tgt_shape"])
    pc = {k: v.detach().clone().requires_grad_(True) for k, v in model.state_dict().items()}
    loss = cpu_loss_builder(cfg, pc, [t.detach() for t in inputs], None if cfg["kind"] == "pino2d_train" else tgt, spec["bs"])

    def one():
        for v in pc.values():
            v.grad = None
        loss().backward()
    for _ in range(spec["warmups"]):
        one()
    t0, n = time.perf_counter(), 0
    while n < spec["iters"]:
        one()
        n += 1
        el = time.perf_counter() - t0
        print(json.dumps(dict(iters=n, seconds=round(el, 4), threads=nthr, pinned_cpus=cores[:nthr][:4] + ["..."] * (nthr > 4))),
              flush=True)
        if el > spec["budget"]:
            break
    return 0


def run_cpu_child(spec, timeout):
    """Run cpu_child(spec) in its own process (never initialises the GPU); returns its last progress record or None."""
    import subprocess
    env = {k: v for k, v in os.environ.items() if k != "LD_PRELOAD" and not k.startswith(("ROCP", "OMP_", "MKL_"))}
    env["HIP_VISIBLE_DEVICES"] = ""
    cmd = [sys.executable, os.path.abspath(__file__), "--cpu-child", json.dumps(spec)]
    proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True)
    try:
        out, _ = proc.communicate(timeout=timeout)
    except subprocess.TimeoutExpired:
        proc.kill()                               # exactly the process started here
        out, _ = proc.communicate()
    last = None
    for line in (out or "").splitlines():
        try:
            last = json.loads(line)
        except ValueError:
            pass
    return last


def spawn_ranks(n, cmd, extra_env=None):
    """Start `cmd` n times, one process per rank (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in the environment, a free
    rendezvous port on 127.0.0.1); returns (rank 0's stdout, worst return code).  The other ranks' stdout is discarded, every
    rank's stderr passes through."""
    import socket
    import subprocess
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        env.update(extra_env or {})
        procs.append(subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    out0 = procs[0].communicate()[0].decode()
    rcs = [procs[0].returncode] + [q.wait() for q in procs[1:]]
    return out0, max(abs(rc) for rc in rcs)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", default="fno2d_128x128_w64_m12_b64", choices=sorted(CONFIGS))
    ap.add_argument("--repeats", type=int, default=15,
                    help="the timed block of --steps steps is run this many times (each bracketed by barrier + synchronize, "
                         "MAX over ranks); ms_per_step / value are the MEDIAN block, value_min / value_max the extremes")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-child", default=None, help=argparse.SUPPRESS)     # internal: the pinned CPU-baseline process
    ap.add_argument("--profile-steps", type=int, default=5)
    ap.add_argument("--no-overlap", action="store_true",
                    help="N > 1: exchange all gradients in one all-reduce after the backward pass instead of starting "
                         "the late layers' segment while the early layers are still being differentiated")
    ap.add_argument("--overlap", default="auto", choices=("auto", "on"),
                    help="N > 1, fused FNO models: auto = time three steps with and without the overlapped exchange at start-up "
                         "and keep the faster arm; on = always overlap")
    ap.add_argument("--no-exact-fp32", action="store_true", help="skip the second timed block in the exact-fp32 GEMM mode")
    ap.add_argument("--graph", action="store_true",
                    help="capture the whole step once into a hipGraph and replay it (single GPU; the default for the "
                         "launch-bound small configurations, marked graph=True in CONFIGS)")
    ap.add_argument("--eager", action="store_true", help="never replay a hipGraph, also where it is the workload's default")
    args = ap.parse_args()
    if args.cpu_child:
        return cpu_child(json.loads(args.cpu_child))
    if CONFIGS[args.config].get("graph") and not args.eager:
        # (N > 1 too since round 6: the step replays as two graphs with the all-reduce of a plain bucket between them)
        args.graph = True

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # self-launch: one child process per GPU, started BEFORE anything here touches the GPU (never exec / re-exec a process
        # that has initialised it); rank 0's JSON line is the children's only stdout and becomes ours
        out0, rc = spawn_ranks(args.gpus, [sys.executable, os.path.abspath(__file__)] + sys.argv[1:])
        sys.stdout.write(out0)
        sys.stdout.flush()
        sys.exit(rc)

    sha, src_hash = git_sha(), source_hash()          # child process + file reads: before anything touches the GPU

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # FNO_BENCH_FORCE_DIST=1: run the multi-rank code path (process group, broadcast, overlapped all-reduce, barriers)
    # with a single rank - a self-test of the N > 1 path on a one-GPU box
    force_dist = os.environ.get("FNO_BENCH_FORCE_DIST") == "1" and world == 1
    dist_on = world > 1 or force_dist
    if force_dist:
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
    if dist_on:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if args.gpus != world and not force_dist:
        sys.exit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}")
    assert torch.cuda.is_available(), "bench.py needs a GPU (the engine has no CPU path)"
    # FNO_BENCH_ONE_DEVICE=1 + FNO_BENCH_BACKEND=gloo: every rank on device 0, gloo instead of RCCL - a self-test of the WHOLE N > 1
    # flow (probe of the exchange arms, timed blocks, extra arms, profiled steps with their collectives) on a one-GPU box, where RCCL
    # cannot place two ranks; the numbers of such a run mean nothing
    backend = os.environ.get("FNO_BENCH_BACKEND", "nccl")
    if os.environ.get("FNO_BENCH_ONE_DEVICE") == "1":
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if dist_on:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    from pde_policylearning_amd import _lib
    from pde_policylearning_amd.neuralop.models import FNO2d, FNO3d
    from pde_policylearning_amd.trainer import FlatGradBucket, FusedAdam, FusedLpLoss, broadcast_parameters, train_step

    cfg = CONFIGS[args.config]
    B = cfg["batch"]
    fused_model = cfg["kind"] in ("2d", "3d")
    model, inputs, tgt = make_workload(cfg, rank, dev)
    x = inputs[0]
    broadcast_parameters(model)
    overlap = fused_model and dist_on and not args.no_overlap and not args.graph      # (graph replay: plain bucket, one all-reduce)
    dp_choice = None
    if overlap and args.overlap == "auto":
        # let the data decide: three timed steps after two warm-ups with the overlapped two-part exchange and
        # with the single all-reduce, MAX over ranks each; the faster arm runs the benchmark, both timings go into the JSON line
        def probe(make_bucket):
            bk = make_bucket()
            bk.force_collective = force_dist
            op = FusedAdam(bk, lr=1e-3, weight_decay=1e-4, skip_dead_slices=True)
            lf = FusedLpLoss(size_average=False)
            for _ in range(2):
                train_step(model, bk, op, inputs, tgt, lf)
            torch.cuda.synchronize(); dist.barrier(); torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(3):
                train_step(model, bk, op, inputs, tgt, lf)
            torch.cuda.synchronize(); dist.barrier(); torch.cuda.synchronize()
            t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            bk.close()      # (an overlapped bucket takes itself off the fused module: the other arm must not inherit it - ADVICE r05)
            return float(t.item()) / 3 * 1e3
        state0 = {k: v.detach().clone() for k, v in model.state_dict().items()}
        t_ov = probe(lambda: FlatGradBucket.for_fno(model, split_layer=1))
        model.load_state_dict(state0)
        t_single = probe(lambda: FlatGradBucket(model.parameters(), direct_module=model))
        model.load_state_dict(state0)            # the probes stepped the weights: the timed run starts from the initial ones
        overlap = t_ov <= t_single
        dp_choice = dict(overlapped_ms=round(t_ov, 4), single_allreduce_ms=round(t_single, 4),
                         chosen="overlapped" if overlap else "single all-reduce")
    if overlap:
        # [projection | blocks L-1..1] go on the wire (async RCCL all-reduce) while block 0 and the lifting are differentiated
        bucket = FlatGradBucket.for_fno(model, split_layer=1)
    else:
        # fused FNO: every gradient is written in place; PINO observers: their spectral weights (> 99 % of the bytes) are
        # RNO2d: its spectral weights too, while the sequence is one time step long (functional.single_use decides per call;
        # the bucket is cleared in full so that the accumulating fallback starts from zeros)
        bucket = FlatGradBucket(model.parameters(), direct_module=model, zero_all=cfg["kind"].startswith("rno2d"))
    bucket.force_collective = force_dist
    if not fused_model and dist_on and args.graph:
        args.graph = False      # the observers' segmented, dead-slice-skipping exchange starts collectives inside the backward pass: eager under DP
    if not fused_model:
        # layer-ordered segments go on the wire as their gradients complete; dead last-dim slices of the dialect-C weights
        # (PINObserverFullField at T = 1: 11/12 of 906 MB) are never exchanged.  Single GPU: the plan is only reported.
        from pde_policylearning_amd.trainer import enable_dp_exchange
        enable_dp_exchange(bucket, model, tuple(t[:1] for t in inputs))
    # run_pde_observers.py:134; dead last-dim slices of dialect-C weights replayed instead of stepped (as train_observer does)
    opt = FusedAdam(bucket, lr=1e-3, weight_decay=1e-4, capturable=args.graph, skip_dead_slices=True)
    loss_fn = FusedLpLoss(size_average=False)                  # run_pde_observers.py:138
    if cfg["kind"] == "pino2d_train":
        from pde_policylearning_amd.libs.pino_utils.losses import get_forcing
        from pde_policylearning_amd.trainer import PinoObjective
        loss_fn = PinoObjective(get_forcing(cfg["size"][0]).to(dev), 0.5, 5.0, 1.0, 0.0)     # ic_loss 5, f_loss 1, xy_loss 0
        tgt = (tgt.reshape(tgt.shape[:4]), x, inputs[1].reshape(B))
    if cfg["kind"] == "pino_ff_pde":                           # run_pde_observers.py:207-231 on a 32 x 130 x 32 channel
        from pde_policylearning_amd.libs.envs.control_env import ChannelFlowRHS
        from pde_policylearning_amd.trainer import FullFieldObjective, MeanStdDecoder
        env = ChannelFlowRHS.tanh_channel(32, 130, 32)
        gen = torch.Generator(device="cpu").manual_seed(4321 + rank)
        rnd = lambda *sh: torch.randn(sh, generator=gen).to(dev)
        decoder = MeanStdDecoder(0.1 * rnd(32, 32), 0.5 + torch.rand((32, 32), generator=gen).to(dev), device=dev)
        loss_fn = FullFieldObjective(decoder, [-10, -8, -6], env, 1.0)
        tgt = (rnd(B, 1, 3, 32, 32), 1 + 0.5 * rnd(B, 1, 32, 131, 32), 0.3 * rnd(B, 1, 32, 130, 32), 0.3 * rnd(B, 1, 32, 131, 32))
    cfg = dict(cfg, n_params=sum(p.numel() * (2 if p.is_complex() else 1) for p in model.parameters()))

    def step():
        return train_step(model, bucket, opt, inputs, tgt, loss_fn)

    eager_step = step
    if args.graph:
        from pde_policylearning_amd.trainer import GraphedTrainStep
        graphed = GraphedTrainStep(model, bucket, opt, inputs, tgt, loss_fn)

        def step():
            return graphed()

    def sync():
        torch.cuda.synchronize()
        if dist_on:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    blocks = []
    for _ in range(max(1, args.repeats)):
        sync()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            loss = step()
        if getattr(opt, "_runs", None) is not None:
            opt.sync_dead_slices()       # the deferred replay of the skipped (dead) weight slices belongs to these steps' work
        sync()
        blocks.append(time.perf_counter() - t0)
    if dist_on:
        tmax = torch.tensor(blocks, dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)          # every block: the slowest rank's time
        blocks = [float(v) for v in tmax.tolist()]
    assert torch.isfinite(loss).all(), "non-finite loss"
    srt = sorted(blocks)
    dt = srt[len(srt) // 2] if len(srt) % 2 else 0.5 * (srt[len(srt) // 2 - 1] + srt[len(srt) // 2])
    fields_per_s = B * world * args.steps / dt

    # ---- the precision-matched arm: the same step with every channel GEMM on the exact-fp32 matrix instructions
    # (fno_set_gemm_mode(0): v_mfma_f32_32x32x2_f32, no split-precision products) - a short second timed block, same bracketing
    exact_fp32 = None
    if not args.no_exact_fp32 and _lib.lib().fno_get_gemm_mode() == 1:
        # (every workload since round 6: the observers' modules take the unfused compositions where a fused path exists in
        # split precision only - loose rows, the regressor tails; eager steps: a captured graph holds the split-precision launches)
        Lb = _lib.lib()
        Lb.fno_set_gemm_mode(0)
        try:
            for _ in range(3):
                eager_step()
            eb = []
            nst = max(3, args.steps // 2)
            for _ in range(3):
                sync()
                t0 = time.perf_counter()
                for _ in range(nst):
                    eager_step()
                sync()
                eb.append(time.perf_counter() - t0)
            if dist_on:
                tm = torch.tensor(eb, dtype=torch.float64, device=dev)
                dist.all_reduce(tm, op=dist.ReduceOp.MAX)
                eb = [float(v) for v in tm.tolist()]
            em = sorted(eb)[1]
            exact_fp32 = dict(value=round(B * world * nst / em, 2), ms_per_step=round(1e3 * em / nst, 4), steps=nst, repeats=3,
                              gemm_mode="f32 (v_mfma_f32_32x32x2_f32 everywhere: fno_set_gemm_mode(0))",
                              step_hbm_frac=round(algorithmic_bytes_per_field(cfg) * B / (em / nst) / (PEAK_HBM_GBS * 1e9), 4),
                              launch="eager")
        except RuntimeError as e:
            exact_fp32 = dict(unavailable=str(e)[:300])
        finally:
            Lb.fno_set_gemm_mode(1)
        eager_step()      # back on the default kernels before the profiled steps

    # ---- the mode contraction on the matrix cores, beside the default (north_star: 'bixy,ioxy->boxy' as a batched complex GEMM on
    # MFMA).  The default 2-D path runs the contraction inside k_spec_mid as a mat-vec per (sample, bin) workgroup on the vector
    # lanes (384 workgroups per launch keep the two DFT phases on every CU); the three-launch
    # sequence k_axis_fwd -> k_mode_gemm (v_mfma_f32_32x32x2_f32, exact fp32, one workgroup per mode x 64 batch rows) ->
    # k_axis_inv is the matrix-core arm: the same step timed with fno_set_fused_mid(0), its kernels profiled below.
    mode_contraction = None
    if fused_model and not args.graph and not args.no_exact_fp32 and cfg["kind"] == "2d" and _lib.lib().fno_get_fused_mid() == 1:
        Lb = _lib.lib()
        Lb.fno_set_fused_mid(0)
        try:
            for _ in range(3):
                step()
            mb = []
            nst = max(5, args.steps // 2)
            for _ in range(3):
                sync()
                t0 = time.perf_counter()
                for _ in range(nst):
                    step()
                sync()
                mb.append(time.perf_counter() - t0)
            if dist_on:
                tm = torch.tensor(mb, dtype=torch.float64, device=dev)
                dist.all_reduce(tm, op=dist.ReduceOp.MAX)
                mb = [float(v) for v in tm.tolist()]
            mm = sorted(mb)[1]
            arm = {}
            if args.profile_steps > 0:       # (every rank runs the profiled steps: they contain the gradient exchange)
                Lb.fno_profile_reset(); Lb.fno_profile_enable(1)
                for _ in range(2 * args.profile_steps):
                    eager_step()
                torch.cuda.synchronize()
                Lb.fno_profile_reset()
                for _ in range(args.profile_steps):
                    eager_step()
                torch.cuda.synchronize()
                Lb.fno_profile_enable(0)
                arm = {n: dict(launches_per_step=k / args.profile_steps, avg_ms=round(ms / k, 4))
                       for n, ms, k in _lib.profile_summary() if n.startswith(("k_axis", "k_mode_gemm", "k_spec_mid"))}
                Lb.fno_profile_reset()
            mode_contraction = dict(
                default="k_spec_mid: leading-axis DFT + contraction (vector lanes, mat-vec per (sample, bin)) + inverse DFT in one launch",
                mfma_arm=dict(value=round(B * world * nst / mm, 2), ms_per_step=round(1e3 * mm / nst, 4), kernels=arm,
                              what="fno_set_fused_mid(0): k_axis_fwd -> k_mode_gemm on v_mfma_f32_32x32x2_f32 -> k_axis_inv per block "
                                   "and direction; SQ_INSTS_MFMA of k_mode_gemm: profiles/r06_pmc_sq_mfma_arm.csv"))
        finally:
            Lb.fno_set_fused_mid(1)
        step()

    # ---- N > 1: what the gradient exchange costs, and how much of it the overlap hides ----
    exchange = None
    if dist_on:
        bucket.time_exchange(True)
        for _ in range(max(5, args.profile_steps)):
            eager_step()
        torch.cuda.synchronize()
        ms = bucket.exchange_ms()
        bucket.time_exchange(False)
        if ms is not None:
            tm = torch.tensor(ms, dtype=torch.float64, device=dev)
            dist.all_reduce(tm, op=dist.ReduceOp.MAX)
            exchange = [round(float(v), 4) for v in tm.tolist()]

    # ---- per-kernel timing with HIP events on the launch stream (separate profiled steps) ----
    roofline = None
    kernels = []
    # (every rank runs the profiled steps - for N > 1 they contain the collective of the gradient exchange; rank 0 reports)
    if args.profile_steps > 0:
        L = _lib.lib()
        L.fno_profile_reset()
        L.fno_profile_enable(1)
        for _ in range(args.profile_steps):      # fills the library's event pool (creation is slow enough to open gaps
            eager_step()                         # between the kernels); these records are dropped
        torch.cuda.synchronize()
        L.fno_profile_reset()
        for _ in range(args.profile_steps):
            eager_step()           # per-kernel events need individual launches
        torch.cuda.synchronize()
        L.fno_profile_enable(0)
        prof = _lib.profile_summary(with_terms=True)
        L.fno_profile_reset()
        km = kernel_model(cfg)
        tot = sum(ms for _, ms, _, _ in prof)
        mode_terms = 3 if L.fno_get_gemm_mode() == 1 else 1
        # The per-launch event pairs run long (r04: their sum 2.46 ms against a 2.33 ms step and 2.31 ms of rocprofv3 kernel
        # time): every launch carries the two event records.  The engine's kernels are > 95 % of an eager fused-FNO step, so
        # their event times are scaled to sum to the timed step; avg_ms is the scaled value (avg_ms_events the raw one) and the
        # roofline fractions use it.  Never scaled UP, and not at all for workloads with sizeable torch-side kernels.
        ev_scale = 1.0
        if fused_model and not args.graph and not dist_on and tot > 0:
            ev_scale = min(1.0, (1e3 * dt / args.steps) / (tot / args.profile_steps))
        for name, ms, n, terms in sorted(prof, key=lambda r: -r[1]):
            avg_ev = ms / n
            # (the RAW event time is `avg_ms` and the basis of every rate below; the value scaled to the timed step
            # is reported beside it - the event pairs' overhead is a per-launch constant, not proportional to the duration)
            avg = avg_ev
            rec = dict(name=name, launches_per_step=n / args.profile_steps, avg_ms=round(avg, 4), avg_ms_events=round(avg_ev, 4),
                       avg_ms_scaled=round(avg_ev * ev_scale, 4), share=round(ms / tot, 3))
            if name in INSTANTIATION.get(args.config, {}):
                rec["instantiation"] = INSTANTIATION[args.config][name]
            if name in km:
                pipe_name, pipe_peak = PIPE[terms if terms in PIPE else mode_terms]
                rec["GBps"] = round(km[name]["bytes"] / avg / 1e6, 1)
                rec["TFLOPs"] = round(km[name]["flops"] / avg / 1e9, 2)
                rec["hbm_frac"] = round(rec["GBps"] / PEAK_HBM_GBS, 4)
                rec["pipe"] = pipe_name
                rec["pipe_peak_TFLOPs"] = round(pipe_peak, 1)
                rec["pipe_frac"] = round(rec["TFLOPs"] / pipe_peak, 4)
            kernels.append(rec)
        dom = next((k for k in kernels if k["name"] in km), None)
        if dom is not None:
            f_h, f_m = dom["hbm_frac"], dom["pipe_frac"]
            if f_m >= f_h:
                roofline = dict(kernel=dom["name"], bound="mfma", achieved=dom["TFLOPs"], peak=dom["pipe_peak_TFLOPs"],
                                unit="TFLOP/s", frac=round(f_m, 4), traffic=None, pipe=dom["pipe"], hbm_frac=f_h)
            else:
                roofline = dict(kernel=dom["name"], bound="hbm", achieved=dom["GBps"], peak=PEAK_HBM_GBS,
                                unit="GB/s", frac=round(f_h, 4), traffic=None, pipe=dom["pipe"], pipe_frac=f_m)
            roofline["note"] = ("frac = the larger of (algorithmic bytes / launch time / 8 TB/s) and (fp32-equivalent algorithmic "
                                "FLOP/s / the dense peak of the matrix pipe the kernel's GEMMs run on divided by its products per k block)")
            # HBM bytes per launch from the committed PMC passes (profiles/, tools/pmc_traffic.py): quoted only when the file
            # was collected on THESE kernel sources (source_hash) and this workload
            try:
                pj = json.load(open(os.path.join(ROOT, "profiles", PMC_TRAFFIC_JSON)))
                if pj.get("source_hash") != src_hash:
                    roofline["traffic_source"] = (f"profiles/{PMC_TRAFFIC_JSON} was collected on other kernel sources "
                                                  f"({pj.get('source_hash')} != {src_hash}): not quoted")
                elif pj.get("workload", "fno2d_128x128_w64_m12_b64") == args.config:
                    pt = pj["kernels"]
                    base = {"k_pw_fwd_block": "k_pw_fwd", "k_pw_fwd_lift": "k_pw_fwd"}.get(dom["name"], dom["name"])
                    cands = [k for k in pt if k.split("<")[0] in (base, base + "_x3", base + "_t", base + "_g2")]
                    inst = INSTANTIATION.get(args.config, {}).get(dom["name"])
                    key = inst if inst in pt else next((k for k in cands if "<true" not in k), cands[0])   # not the block-0 LIFT variants
                    roofline["traffic"] = round(pt[key]["fetch_bytes"] + pt[key]["write_bytes"])
                    roofline["traffic_source"] = (f"profiles/{PMC_TRAFFIC_JSON} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in "
                                                  "separate passes, corrected; same command and workload)")
                    roofline["algorithmic_bytes"] = round(km[dom["name"]]["bytes"])
            except Exception:
                pass
            # matrix-pipe occupancy of that kernel from the committed SQ pass: busy cycles summed over the 1024 SIMDs /
            # (1024 x kernel cycles); GRBM_GUI_ACTIVE is summed over the 8 XCDs
            try:
                import csv
                if args.config == "fno2d_128x128_w64_m12_b64" and roofline.get("traffic") is not None:
                    base = {"k_pw_fwd_block": "k_pw_fwd", "k_pw_fwd_lift": "k_pw_fwd"}.get(dom["name"], dom["name"])
                    for r in csv.DictReader(open(os.path.join(ROOT, "profiles", PMC_SQ_CSV))):
                        if r["kernel"] in (base + "_x3", base, base + "_t", base + "_g2"):
                            roofline["matrix_pipe_busy"] = round(float(r["SQ_VALU_MFMA_BUSY_CYCLES"]) / 1024.0
                                                                 / (float(r["GRBM_GUI_ACTIVE"]) / 8.0), 3)
                            roofline["matrix_pipe_busy_source"] = f"profiles/{PMC_SQ_CSV} (rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE, own pass)"
                            if r["kernel"].endswith(("_x3", "_t", "_g2")):
                                break
            except Exception:
                pass
            roofline["avg_launch_ms"] = dom["avg_ms"]
            roofline["event_scale"] = round(ev_scale, 4)
            if "instantiation" in dom:
                roofline["instantiation"] = dom["instantiation"]      # the name rocprofv3 --kernel-trace shows for it
            roofline["measured"] = (f"HIP events on the launch stream, {args.profile_steps} profiled steps after the timed region, "
                                    "RAW event durations (event_scale = timed step / sum of event times, for reference only)")
            # the kernel FURTHEST below its roof among those with >= 5 % of the step: `roofline` above is
            # the one with the largest share
            big = [k for k in kernels if k["name"] in km and k["share"] >= 0.05]
            if big:
                worst = min(big, key=lambda k: max(k["hbm_frac"], k["pipe_frac"]))
                roofline["furthest_below_roof"] = dict(kernel=worst["name"], instantiation=worst.get("instantiation"),
                                                        frac=max(worst["hbm_frac"], worst["pipe_frac"]), hbm_frac=worst["hbm_frac"],
                                                        pipe_frac=worst["pipe_frac"], avg_launch_ms=worst["avg_ms"], share=worst["share"])

    # ---- CPU baseline: the oracle (validated restatement of the reference) on the host cores ----
    # Each measurement is its own process, pinned to physical cores of NUMA node 0 before torch creates its thread pool
    # (2 warm-ups, >= 5 timed steps).  torch's CPU FFT / einsum stop scaling long before a many-core
    # host is full, so `value` is the best of an 8 / 16 / 32-thread sweep and `all_cores` the figure SURVEY section 8d asks
    # for: every physical core of the host, one field, bounded at 120 s.
    cpu_baseline = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        kind = cfg["kind"]
        bs = min({"2d": 8, "rno2d": 2, "rno2d_shipped": 8, "pino_ff": 4, "pino_ff_pde": 4}.get(kind, 1), B)
        tshape = [bs] + list((tgt[0] if isinstance(tgt, tuple) else tgt).shape[1:])
        if kind == "pino_ff_pde":
            tshape = [bs, 3 * cfg["size"][0] * cfg["size"][1]]
        node0 = len(physical_cores(0))
        heavy = cfg.get("n_params", 0) > 50_000_000       # weight-dominated observers: seconds per step, one thread count
        sweep = [16] if heavy else [t for t in (8, 16, 32) if t <= node0] or [node0]
        runs = []
        for nthr in sweep:
            spec = dict(config=args.config, bs=bs, tgt_shape=tshape, threads=nthr, warmups=1 if heavy else 2,
                        iters=2 if heavy else 8, budget=20.0 if heavy else 6.0)
            r = run_cpu_child(spec, timeout=240 if heavy else 90)
            if r:
                runs.append(dict(threads=r["threads"], iters=r["iters"], seconds=r["seconds"],
                                 value=round(bs * r["iters"] / r["seconds"], 3)))
        allc = None
        nall = len(physical_cores(None))
        if not heavy and nall > max(sweep):
            r = run_cpu_child(dict(config=args.config, bs=1, tgt_shape=[1] + tshape[1:], threads=nall, all_nodes=True,
                                   warmups=1, iters=5, budget=40.0), timeout=120)
            allc = (dict(value=round(r["iters"] / r["seconds"], 3), cores=r["threads"], iters=r["iters"], seconds=r["seconds"],
                         sample="1 field per step, 1 warm-up") if r else
                    dict(value=None, cores=nall, note="not one timed step of 1 field finished within 120 s"))
        if runs:
            best = max(runs, key=lambda r: r["value"])
            cpu_baseline = dict(value=best["value"], unit="fields/s", cores=best["threads"], kind="port",
                                sample=f"oracle (CPU restatement of the reference, torch ops) zero_grad+fwd+loss+bwd on "
                                       f"{bs} fields of the same shape, own process pinned to {best['threads']} physical cores "
                                       f"of NUMA node 0 ({node0} there, {nall} on the host): {best['iters']} timed steps in "
                                       f"{best['seconds']:.1f}s after {1 if heavy else 2} warm-ups; best of the thread sweep"
                                       + ("; physics term not included" if kind == "pino_ff_pde" else ""),
                                sweep=runs, all_cores=allc)

    if rank == 0:
        out = {
            "metric": {"2d": "FNO2d", "3d": "FNO3d", "rno2d": "RNO2d", "rno2d_shipped": "RNO2d", "pino_ff": "PINObserverFullField", "pino_ff_pde": "PINObserverFullField+pde_loss",
                       "pino2d": "PINObserver2d", "pino2d_train": "PINObserver2d+PINO_loss"}[cfg["kind"]] + " fwd+bwd fields/sec",
            "value": round(fields_per_s, 2),
            "unit": "fields/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(1e3 * dt / args.steps, 4),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "gemm_mode": ("split precision, fp32 in / fp32 accumulate / fp32 out: channel GEMMs on two fp16 terms per operand "
                          "scaled by published magnitude bounds (3 products per k block), small problems and the spectral "
                          "K-extension on three bf16 terms (6 products); parity held at 1e-5 against the float64 oracle; "
                          "per kernel: kernels[].pipe")
                         if _lib.lib().fno_get_gemm_mode() == 1 else "f32 (v_mfma_f32_32x32x2_f32)",
            "data": "synthetic",
            "repeats": len(blocks),
            "value_min": round(B * world * args.steps / max(blocks), 2),
            "value_max": round(B * world * args.steps / min(blocks), 2),
            "ms_per_step_all": [round(1e3 * b / args.steps, 4) for b in blocks],
            "n_ranks": dist.get_world_size() if dist_on else 1,      # RCCL ranks seen by torch.distributed
            "rccl_version": ".".join(str(v) for v in torch.cuda.nccl.version()) if dist_on and backend == "nccl" else None,
            "backend": backend if dist_on else None,
            "allreduce_ms_total": exchange[0] if exchange else None,
            "allreduce_ms_exposed": exchange[1] if exchange else None,
            "gradient_exchange": {"bucket_bytes": 4 * bucket.flat.numel(), "wire_bytes_per_rank_per_step": bucket.planned_wire_bytes(),
                                  "segments": len(getattr(bucket, "_segments", None) or [1]),
                                  "kind": "overlapped late-layer segment + rest" if overlap else
                                          ("segmented, live slices of dialect-C weights only" if not fused_model else "single all-reduce")},
            "git_sha": sha,
            "source_hash": src_hash,
            "config": {"workload": args.config, "batch_per_gpu": B, "global_batch": B * world,
                       "step": "zero_grad+fwd+" + {"pino_ff_pde": "decode+LpLoss(sum)+channel-flow pde_loss", "pino2d_train": "5*IC+PDE residual loss"}.get(cfg["kind"], "LpLoss(sum)") + "+bwd" +
                               ("+allreduce(sum" + (", overlapped with bwd)" if overlap else ")") if dist_on else "") + "+Adam",
                       "parallelism": f"dp{world}", "launch": "hipGraph replay" if args.graph else "eager"},
            "roofline": roofline,
            "step_hbm_frac": (round(algorithmic_bytes_per_field(cfg) * B / (dt / args.steps) / (PEAK_HBM_GBS * 1e9), 4)
                              if algorithmic_bytes_per_field(cfg) else None),
            "step_hbm_note": "whole step: algorithmic bytes per field (bench.algorithmic_bytes_per_field: SURVEY 8(d) for the fused FNO models, the models stated in DESIGN.md section 5 for the observers) x batch / ms_per_step / 8 TB/s",
            "exact_fp32": exact_fp32,
            "mode_contraction": mode_contraction,
            "dp_exchange_choice": dp_choice,
            "cpu_baseline": cpu_baseline,
            "kernels": kernels,
        }
    if dist_on:
        dist.destroy_process_group()
    if rank == 0:
        # RCCL's version banner (NCCL_DEBUG=VERSION) sits in the C stdio buffer: flush it first so that the JSON line
        # is the last thing this rank writes
        try:
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
