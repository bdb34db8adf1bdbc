#!/bin/bash
mkdir -p gpurun_out/r04
( time timeout -k 10 600 python -m pytest tests/test_fullsize_gpu.py -x -q --durations=8 ) > gpurun_out/r04/gputest2.txt 2>&1
tail -n 16 gpurun_out/r04/gputest2.txt
python bench.py > gpurun_out/r04/bench2.json 2> gpurun_out/r04/bench2.err; cut -c1-330 gpurun_out/r04/bench2.json
FNO_GEMM_F32=1 python bench.py --no-cpu-baseline > gpurun_out/r04/bench_f32mode.json 2> gpurun_out/r04/bench_f32.err; cut -c1-330 gpurun_out/r04/bench_f32mode.json
FNO_LIB_PATH=$PWD/tools/exp_clock.so python tools/kernel_clock.py > gpurun_out/r04/kernel_clock.txt 2>&1; cat gpurun_out/r04/kernel_clock.txt | tail -n 6
exit 0
