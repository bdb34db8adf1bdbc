#!/bin/bash
mkdir -p gpurun_out/r04
timeout -k 10 300 python -m pytest tests/test_parity_gpu.py -x -q -k "projection or fno_model_golden" 2>&1 | tail -n 2
FNO_LIB_PATH=$PWD/tools/exp_clock.so python tools/kernel_clock.py 2>&1 | tail -n 8
for i in 1 2; do
python bench.py --no-cpu-baseline --repeats 5 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); k={x['name']:x['avg_ms'] for x in d['kernels']}; print('new', d['ms_per_step'], 'proj_fwd', k.get('k_proj_fwd'))"
FNO_NO_PFWD_W=1 python bench.py --no-cpu-baseline --repeats 5 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); k={x['name']:x['avg_ms'] for x in d['kernels']}; print('old', d['ms_per_step'], 'proj_fwd', k.get('k_proj_fwd'))"
done
exit 0
