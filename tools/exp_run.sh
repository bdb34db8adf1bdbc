#!/bin/bash
mkdir -p gpurun_out/r04
out=gpurun_out/r04/rno_dbg2.txt
: > $out
for e in "X=1" "FNO_GEMM_F32=1"; do echo "== $e" >> $out; env $e timeout -k 10 400 python tools/rno_debug.py 2>&1 | grep -v amdgpu.ids | head -n 62 >> $out; done
cut -c1-200 $out | sed -n 1,62p
exit 0
