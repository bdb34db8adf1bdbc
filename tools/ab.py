"""usage (GPU box): python tools/ab.py [--config NAME] [--n 9] ENV1=1 'ENV2=1 ENV3=x' ...  -> one bench line per environment
(A/B runs inside ONE gpurun call: the pool's boxes differ by +-8 %).  '-' = no extra environment."""
import json, os, subprocess, sys
args = sys.argv[1:]
cfg, n = None, 9
while args and args[0].startswith("--"):
    if args[0] == "--config": cfg = args[1]
    if args[0] == "--n": n = int(args[1])
    args = args[2:]
for e in args:
    env = dict(os.environ)
    if e != "-":
        for kv in e.split():
            k, v = kv.split("=", 1)
            env[k] = v
    cmd = [sys.executable, "bench.py", "--steps", "30", "--warmup", "5", "--no-cpu-baseline"] + (["--config", cfg] if cfg else [])
    try:
        out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=200).stdout
        d = json.loads(out.strip().splitlines()[-1])
        print(f"{e:28s} {d['value']:9.1f} {d['ms_per_step']:7.4f} ms  " +
              " ".join(f"{k['name']}={k['avg_ms']}x{k['launches_per_step']:g}" for k in d["kernels"][:n]), flush=True)
    except Exception as ex:
        print(e, "FAILED", repr(ex)[:200], flush=True)
