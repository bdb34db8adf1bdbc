# usage: tools/ab_step.sh <rounds> <lib|default> ...   interleaved bench runs on one box (step time only); mean and spread per lib at the end
n=$1; shift
rm -f gpurun_out/ab_step.txt
for i in $(seq $n); do for l in "$@"; do
  [ "$l" = default ] && lp="" || lp=$PWD/tools/_libs/$l
  FNO_LIB_PATH=$lp timeout -k 10 300 python bench.py --no-cpu-baseline --repeats 7 --steps 20 --warmup 5 --no-exact-fp32 --profile-steps 0 > gpurun_out/ab_tmp.json 2>> gpurun_out/ab.err
  python - "$l" >> gpurun_out/ab_step.txt <<PY
import json,sys
d=json.loads(open("gpurun_out/ab_tmp.json").read().strip().splitlines()[-1])
print(sys.argv[1], d["ms_per_step"], min(d["ms_per_step_all"]))
PY
done; done
python - <<PY
import collections, statistics as st
r=collections.OrderedDict()
for ln in open("gpurun_out/ab_step.txt"):
    n,a,b=ln.split(); r.setdefault(n,[]).append((float(a),float(b)))
for n,v in r.items():
    med=[x[0] for x in v]; mn=[x[1] for x in v]
    print("%-20s median-of-blocks: mean %.4f sd %.4f  | min-of-blocks: mean %.4f  | runs %s"%(n, st.mean(med), st.pstdev(med), st.mean(mn), " ".join("%.4f"%x for x in med)))
PY
