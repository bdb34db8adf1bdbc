# usage: tools/ab_cfg.sh <config> <lib|default> ...   one bench run per lib on the same box; prints step time and the named kernels
cfg=$1; shift
for l in "$@"; do
  [ "$l" = default ] && lp="" || lp=$PWD/tools/_libs/$l
  FNO_LIB_PATH=$lp timeout -k 10 300 python bench.py --config $cfg --no-cpu-baseline --repeats 7 --steps 10 --warmup 3 --no-exact-fp32 > gpurun_out/ab_tmp.json 2>> gpurun_out/ab.err
  python - "$l" <<PY
import json,sys
d=json.loads(open("gpurun_out/ab_tmp.json").read().strip().splitlines()[-1])
ks={k["name"]:k["avg_ms"] for k in d["kernels"]}
print(sys.argv[1], d["ms_per_step"], "min", round(min(d["ms_per_step_all"]),4), {k:v for k,v in ks.items() if "pino" in k or "absmax" in k or "pack_w" in k})
PY
done
