#!/bin/bash
mkdir -p gpurun_out/exp
run() {   # name, lib, env...
  name=$1; lib=$2; shift 2
  ( for kv in "$@"; do export "$kv"; done
    if [ -n "$lib" ]; then export FNO_LIB_PATH=$PWD/tools/$lib; fi
    python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/exp/bench_$name.json 2> gpurun_out/exp/bench_$name.err ) || { tail -n 5 gpurun_out/exp/bench_$name.err; return 1; }
  python - "$name" <<'PY'
import json, sys
d = json.load(open(f"gpurun_out/exp/bench_{sys.argv[1]}.json"))
ks = {k["name"]: k["avg_ms"] for k in d.get("kernels", [])}
print(sys.argv[1], d["value"], d["ms_per_step"], {k: v for k, v in ks.items() if "proj" in k or "bl" in k or "mid" in k})
PY
}
( time timeout -k 10 900 python -m pytest tests/ -x -q -m gpu ) > gpurun_out/exp/gputest.txt 2>&1; tail -n 5 gpurun_out/exp/gputest.txt
FNO_LIB_PATH=$PWD/tools/exp_clock.so python tools/kernel_clock.py 2>&1 | grep -v amdgpu.ids > gpurun_out/exp/clock4.txt; cat gpurun_out/exp/clock4.txt
run new "" &&
run new_notw3 "" FNO_MID_TW3=0 &&
run new2 ""
