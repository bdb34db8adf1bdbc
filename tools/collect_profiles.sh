#!/bin/bash
# usage (GPU box, from the repo root): tools/collect_profiles.sh <tag> [quick|bench|prof]   -> gpurun_out/<tag>_*
# (the whole set takes longer than one gpurun call may: `bench` = the bench lines of every workload, `prof` = headline line +
# kernel-trace stats of four workloads + the PMC passes, `quick` = headline line + its stats + the PMC passes)
# bench JSON lines (headline + secondary workloads), rocprofv3 kernel-trace stats, the two HBM-traffic PMC passes and
# two SQ passes (matrix-pipe busy cycles, VALU / wait cycles) - every --pmc pass is its own run with --kernel-trace only.
# Every step runs under `timeout` and appends to files (a silent hang would otherwise eat the GPU budget).
tag="$1"; quick="$2"; R=$PWD; out=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp; cd $R
T="timeout -k 5 240"
if [ "$quick" != "bench" ]; then
$T python3 bench.py > $out/${tag}_bench.json 2> $out/${tag}_bench.err
echo "headline done" >> $out/${tag}_progress.log
fi
if [ -z "$quick" ] || [ "$quick" = "bench" ]; then
$T python3 bench.py --config fno2d_64x64_w32_m8_b4 --steps 50 > $out/${tag}_bench_cfg1.json 2>> $out/${tag}_bench.err
$T python3 bench.py --config fno2d_64x64_w32_m8_b4 --no-cpu-baseline --steps 50 --eager > $out/${tag}_bench_cfg1_eager.json 2>> $out/${tag}_bench.err
$T python3 bench.py --config fno3d_64_w32_m8_b16 --steps 10 > $out/${tag}_bench_fno3d.json 2>> $out/${tag}_bench.err
for c in rno2d_128x128_w64_m12_b32 rno2d_32x32_w34_m12_b32 pino_fullfield_32x32_w64_m12_b32 pino_fullfield_pde_32x130x32_w64_m12_b32 \
         pinobserver2d_128x128x65_w64_m8_b2 pino_finetune_128x128x65_w64_m8_b4 pino_finetune_256x256x65_w64_m20_b1; do
  $T python3 bench.py --config $c --steps 10 --warmup 3 > $out/${tag}_bench_$c.json 2>> $out/${tag}_bench.err
  echo "$c done" >> $out/${tag}_progress.log
done
$T python3 bench.py --config rno2d_128x128_w64_m12_b32 --steps 10 --warmup 3 --graph --no-cpu-baseline > $out/${tag}_bench_rno2d_128x128_w64_m12_b32_graph.json 2>> $out/${tag}_bench.err
$T python3 bench.py --config pino_fullfield_32x32_w64_m12_b32 --steps 10 --warmup 3 --eager --no-cpu-baseline > $out/${tag}_bench_pino_fullfield_32x32_w64_m12_b32_eager.json 2>> $out/${tag}_bench.err
$T python3 bench.py --config pino_fullfield_pde_32x130x32_w64_m12_b32 --steps 10 --warmup 3 --eager --no-cpu-baseline > $out/${tag}_bench_pino_fullfield_pde_32x130x32_w64_m12_b32_eager.json 2>> $out/${tag}_bench.err
fi
[ "$quick" = "bench" ] && exit 0
prof() {   # prof <name> <bench args...>: kernel-trace stats of one workload
  local name=$1; shift
  $T rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_stats_$name -- python3 bench.py "$@" --no-cpu-baseline --no-exact-fp32 > $out/${tag}_stats_$name.log 2>&1
  cp $(ls $out/${tag}_stats_$name/*/*kernel_stats.csv | head -1) $out/${tag}_kernel_stats_$name.csv
  echo "stats $name done" >> $out/${tag}_progress.log
}
prof headline --steps 20 --warmup 3
cp $out/${tag}_kernel_stats_headline.csv $out/${tag}_kernel_stats.csv
if [ -z "$quick" ] || [ "$quick" = "prof" ]; then
prof pino_fullfield --config pino_fullfield_32x32_w64_m12_b32 --steps 10 --warmup 3 --eager
prof fno3d --config fno3d_64_w32_m8_b16 --steps 10 --warmup 3
prof rno2d --config rno2d_128x128_w64_m12_b32 --steps 10 --warmup 3
prof pino_finetune_256 --config pino_finetune_256x256x65_w64_m20_b1 --steps 5 --warmup 2
fi
for c in FETCH_SIZE WRITE_SIZE; do
  $T rocprofv3 --kernel-trace --pmc $c --output-format csv -d $out/${tag}_pmc_$c -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-exact-fp32 --profile-steps 0 > /dev/null 2>&1
  cp $(ls $out/${tag}_pmc_$c/*/*counter_collection.csv | head -1) $out/${tag}_pmc_$(echo $c | tr A-Z a-z).csv
  echo "pmc $c done" >> $out/${tag}_progress.log
done
python3 tools/pmc_traffic.py $out/${tag}_pmc_FETCH_SIZE $out/${tag}_pmc_WRITE_SIZE $out/${tag}_pmc_traffic.json fno2d_128x128_w64_m12_b64
i=0
for grp in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA GRBM_GUI_ACTIVE" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_LDS SQ_WAVES"; do
  i=$((i+1))
  $T rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $out/${tag}_pmc_sq$i -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-exact-fp32 --profile-steps 0 > /dev/null 2>&1
  echo "pmc sq$i done" >> $out/${tag}_progress.log
done
python3 tools/pmc_sq.py $out/${tag}_pmc_sq1 $out/${tag}_pmc_sq2 > $out/${tag}_pmc_sq.csv 2>> $out/${tag}_bench.err
# the matrix-core arm of the mode contraction (three launches per block and direction, k_mode_gemm on v_mfma_f32_32x32x2_f32):
# SQ_INSTS_MFMA / pipe-busy of k_mode_gemm at the headline workload (bench.py reports the arm's step time as mode_contraction.mfma_arm)
FNO_NO_FUSED_MID=1 $T rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA GRBM_GUI_ACTIVE --output-format csv -d $out/${tag}_pmc_sq3 -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-exact-fp32 --profile-steps 0 > /dev/null 2>&1
python3 tools/pmc_sq.py $out/${tag}_pmc_sq3 > $out/${tag}_pmc_sq_mfma_arm.csv 2>> $out/${tag}_bench.err
echo "pmc mfma arm done" >> $out/${tag}_progress.log
# the raw rocprofv3 trees are large: keep the summaries only
rm -rf $out/${tag}_stats_* $out/${tag}_pmc_FETCH_SIZE $out/${tag}_pmc_WRITE_SIZE $out/${tag}_pmc_sq1 $out/${tag}_pmc_sq2 $out/${tag}_pmc_sq3 2>/dev/null
tail -c 400 $out/${tag}_bench.json
