#!/bin/bash
mkdir -p gpurun_out/exp
( time timeout -k 10 900 python -m pytest tests/ -x -q -m gpu ) > gpurun_out/exp/gputest.txt 2>&1; tail -n 5 gpurun_out/exp/gputest.txt
python bench.py --no-cpu-baseline > gpurun_out/exp/bench_new4.json 2>/dev/null
python -c "
import json; d=json.load(open('gpurun_out/exp/bench_new4.json')); print(d['value'], d['ms_per_step'], [(k['name'],k['avg_ms']) for k in d['kernels'][:16]])"
python bench.py --no-cpu-baseline --config fno3d_64_w32_m8_b16 --steps 10 > gpurun_out/exp/bench_new4_fno3d.json 2>/dev/null
python -c "
import json; d=json.load(open('gpurun_out/exp/bench_new4_fno3d.json')); print(d['value'], d['ms_per_step'], [(k['name'],k['avg_ms']) for k in d['kernels'][:12]])"
