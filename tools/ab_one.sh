# usage: tools/ab_one.sh <lib under tools/_libs | ""> ...   one bench run per argument (same box), prints step time and the k_spec_mid / tail kernels
for l in "$@"; do
  [ "$l" = default ] && lp="" || lp=$PWD/tools/_libs/$l
  FNO_LIB_PATH=$lp timeout -k 10 300 python bench.py --no-cpu-baseline --repeats 9 --steps 20 --warmup 5 --no-exact-fp32 > gpurun_out/ab_tmp.json 2>> gpurun_out/ab.err
  python - "$l" <<PY
import json,sys
d=json.loads(open("gpurun_out/ab_tmp.json").read().strip().splitlines()[-1])
ks={k["name"]:k["avg_ms"] for k in d["kernels"]}
print(sys.argv[1], d["ms_per_step"], "min", round(min(d["ms_per_step_all"]),4), "spec_mid", ks.get("k_spec_mid"), "absmax", ks.get("k_absmax"), "pack_w", ks.get("k_pack_w"))
PY
done
