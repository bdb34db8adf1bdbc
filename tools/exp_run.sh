#!/bin/bash
mkdir -p gpurun_out/r04
for e in "X=1" "FNO_NO_PFWD_W=1"; do
env $e python bench.py --config fno3d_64_w32_m8_b16 --no-cpu-baseline --steps 10 --repeats 5 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); k={x['name']:(x['launches_per_step'],x['avg_ms']) for x in d['kernels']}; print('$e fno3d', d['ms_per_step'], d['value'], {n:k[n] for n in list(k)[:8]})"
done
exit 0
