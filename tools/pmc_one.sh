#!/bin/bash
# usage (GPU box): tools/pmc_one.sh <tag> <binary> [args...]  -> gpurun_out/<tag>_pmc.csv: SQ counters per kernel (average per
# launch) of a standalone binary.  Every --pmc pass is its own run with --kernel-trace only.
tag="$1"; shift
R=$PWD; out=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp; cd $R
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU GRBM_GUI_ACTIVE" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SALU" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_SMEM SQ_WAVES SQ_VALU_MFMA_COEXEC_CYCLES"; do
  i=$((i+1))
  timeout -k 5 120 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $out/${tag}_p$i -- "$@" > $out/${tag}_p$i.log 2>&1
  echo "pass $i rc=$?" >> $out/${tag}_progress.log
done
python3 tools/pmc_sq.py $out/${tag}_p1 $out/${tag}_p2 $out/${tag}_p3 > $out/${tag}_pmc.csv
rm -rf $out/${tag}_p1 $out/${tag}_p2 $out/${tag}_p3
cat $out/${tag}_pmc.csv
