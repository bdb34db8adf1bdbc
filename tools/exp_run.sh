#!/bin/bash
mkdir -p gpurun_out/r04
( time timeout -k 10 1000 python -m pytest tests/ -x -q -m gpu --durations=30 ) > gpurun_out/r04/gputest1.txt 2>&1
tail -n 45 gpurun_out/r04/gputest1.txt
exit 0
