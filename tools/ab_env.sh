# usage: tools/ab_env.sh <rounds> "label:ENV=1 ENV2=.." ...   interleaved bench runs on one box, one arm per environment
# (label "-" = no extra environment); prints step time and the big kernels' event times
n=$1; shift
for i in $(seq $n); do for arm in "$@"; do
  l=${arm%%:*}; e=${arm#*:}; [ "$e" = "$arm" ] && e=""
  env $e timeout -k 10 300 python bench.py --no-cpu-baseline --repeats 5 --steps 20 --warmup 5 --no-exact-fp32 --profile-steps 20 > gpurun_out/ab_tmp.json 2>> gpurun_out/ab.err
  python - "$l" <<PY
import json,sys
d=json.loads(open("gpurun_out/ab_tmp.json").read().strip().splitlines()[-1])
ks={k["name"]:k["avg_ms_events"] for k in d["kernels"]}
print("%-12s %.4f"%(sys.argv[1], d["ms_per_step"]), " ".join("%s %.4f"%(n[2:],ks[n]) for n in ("k_proj_bwd","k_block_bwd","k_block_bwd0","k_pw_fwd_block","k_pw_fwd_block0","k_proj_fwd","k_spec_mid") if n in ks), flush=True)
PY
done; done
