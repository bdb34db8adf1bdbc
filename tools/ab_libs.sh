for cfg in fno2d_128x128_w64_m12_b64 rno2d_128x128_w64_m12_b32 pino_finetune_128x128x65_w64_m8_b4 fno3d_64_w32_m8_b16; do
for l in "" tools/_libs/lib_noslp2.so; do
  FNO_LIB_PATH=${l:+$PWD/$l} timeout -k 10 200 python bench.py --config $cfg --no-cpu-baseline --repeats 7 --steps 10 --warmup 3 --no-exact-fp32 --profile-steps 0 > gpurun_out/ab_tmp.json 2>> gpurun_out/ab.err
  python - "$cfg" "${l:-default}" <<PY
import json,sys
d=json.loads(open("gpurun_out/ab_tmp.json").read().strip().splitlines()[-1])
print(sys.argv[1], sys.argv[2], d["value"], d["ms_per_step"], d["value_min"], d["value_max"])
PY
done
done
