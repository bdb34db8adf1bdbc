#!/bin/bash
# usage (GPU box): tools/prof_one.sh <tag> <name> <bench args...>  -> gpurun_out/<tag>_kernel_stats_<name>.csv (rocprofv3 --kernel-trace --stats)
tag="$1"; name="$2"; shift 2; R=$PWD; out=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp; cd $R
timeout -k 5 240 rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_stats_$name -- python3 bench.py "$@" --no-cpu-baseline > $out/${tag}_stats_$name.log 2>&1
cp $(ls $out/${tag}_stats_$name/*/*kernel_stats.csv | head -1) $out/${tag}_kernel_stats_$name.csv
rm -rf $out/${tag}_stats_$name
