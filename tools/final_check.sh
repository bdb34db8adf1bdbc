#!/bin/bash
mkdir -p gpurun_out/r04
( time timeout -k 10 900 python -m pytest tests/ -x -q -m gpu ) > gpurun_out/r04/gputest_final.txt 2>&1
tail -n 6 gpurun_out/r04/gputest_final.txt
python __graft_entry__.py smoke 2>&1 | tail -n 2      # build() then smoke() in one interpreter
python bench.py > gpurun_out/r04/bench_final.json 2> gpurun_out/r04/bench_final.err
python -c "
import json; d=json.load(open('gpurun_out/r04/bench_final.json')); print(d['value'], d['ms_per_step'], d['step_hbm_frac']); print(d['roofline']); print(d['cpu_baseline']['value'], d['cpu_baseline']['cores'])"
