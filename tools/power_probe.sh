# usage (GPU box): tools/power_probe.sh [lib under tools/_libs | default] ...   - package power / shader clock / temperatures from rocm-smi while
# bench.py runs ~4 s of timed steps of the headline workload; one block per library
for l in "${@:-default}"; do
  [ "$l" = default ] && lp="" || lp=$PWD/tools/_libs/$l
  FNO_LIB_PATH=$lp python bench.py --steps 350 --warmup 20 --repeats 5 --no-cpu-baseline --no-exact-fp32 --profile-steps 0 > gpurun_out/power_bench.json 2>/dev/null &
  pid=$!
  sleep 4.5
  echo "== $l"
  for i in 1 2 3 4 5 6 7 8; do
    rocm-smi --showpower --showclocks --showtemp 2>/dev/null | grep -E "Power \(W\)|sclk|Sensor (junction|memory)" | sed -E 's/^GPU\[0\]\s*: //; s/Current Socket Graphics Package Power \(W\)/W/; s/Temperature \(Sensor ([a-z]+)\) \(C\)/T_\1/; s/sclk clock level: [0-9]+: \(([0-9]+)Mhz\)/sclk \1 MHz/' | tr '\n' ' '; echo
    sleep 0.25
  done
  wait $pid
  python - <<PY
import json
d=json.loads(open("gpurun_out/power_bench.json").read().strip().splitlines()[-1])
print("   bench: %.1f fields/s, %.4f ms per step, blocks %s" % (d["value"], d["ms_per_step"], d["ms_per_step_all"]))
PY
done
