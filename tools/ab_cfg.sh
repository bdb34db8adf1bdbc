# usage: tools/ab_cfg.sh <config> <rounds> "label:ENV=.." ...   interleaved bench runs of one workload on one box
cfg=$1; n=$2; shift 2
for i in $(seq $n); do for arm in "$@"; do
  l=${arm%%:*}; e=${arm#*:}; [ "$e" = "$arm" ] && e=""
  env $e timeout -k 10 300 python bench.py --config $cfg --no-cpu-baseline --repeats 5 --steps 10 --warmup 3 --no-exact-fp32 --profile-steps 10 > gpurun_out/ab_tmp.json 2>> gpurun_out/ab.err
  python - "$l" <<PY
import json,sys
d=json.loads(open("gpurun_out/ab_tmp.json").read().strip().splitlines()[-1])
print("%-10s %.4f"%(sys.argv[1], d["ms_per_step"]), " ".join("%s %.4f"%(k["name"][2:],k["avg_ms_events"]) for k in d["kernels"][:7]), flush=True)
PY
done; done
