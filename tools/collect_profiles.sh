#!/bin/bash
# usage (GPU box, from the repo root): tools/collect_profiles.sh <tag>   -> gpurun_out/<tag>_*
# rocprofv3 kernel-trace stats, the bench JSON lines and the two HBM-traffic PMC passes (separate runs).
tag="$1"; R=$PWD; out=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp; cd $R
python3 bench.py > $out/${tag}_bench.json 2> $out/${tag}_bench.err
python3 bench.py --config fno2d_64x64_w32_m8_b4 --no-cpu-baseline --steps 50 > $out/${tag}_bench_cfg1.json 2>> $out/${tag}_bench.err
python3 bench.py --config fno3d_64_w32_m8_b16 --no-cpu-baseline --steps 10 > $out/${tag}_bench_fno3d.json 2>> $out/${tag}_bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_stats -- python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline > $out/${tag}_stats.log 2>&1
cp $(ls $out/${tag}_stats/*/*kernel_stats.csv | head -1) $out/${tag}_kernel_stats.csv
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $out/${tag}_pmc_$c -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --profile-steps 0 > /dev/null 2>&1
  cp $(ls $out/${tag}_pmc_$c/*/*counter_collection.csv | head -1) $out/${tag}_pmc_$(echo $c | tr A-Z a-z).csv
done
python3 tools/pmc_traffic.py $out/${tag}_pmc_FETCH_SIZE $out/${tag}_pmc_WRITE_SIZE $out/${tag}_pmc_traffic.json
tail -c 600 $out/${tag}_bench.json
