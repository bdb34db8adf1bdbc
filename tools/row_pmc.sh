#!/bin/bash
# usage (GPU box, repo root): tools/row_pmc.sh  -> gpurun_out/rowpmc_*  SQ / TCP / TCC counters of the standalone row passes
# (tools/spec_profile.py workload), one rocprofv3 --pmc pass per counter group, summarised by tools/pmc_sq.py
R=$PWD; out=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp; cd $R
i=0
for grp in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU" "SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
           "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_SMEM" \
           "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum" \
           "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum" "GRBM_GUI_ACTIVE TCC_REQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_64B_sum"; do
  i=$((i+1))
  timeout 250 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $out/rowpmc_$i -- python3 tools/spec_profile.py > /dev/null 2>&1 < /dev/null
done
python3 tools/pmc_sq.py $out/rowpmc_* > $out/rowpmc_summary.csv
grep -E "^kernel|rowdft|rowidft|k_pw_fwd" $out/rowpmc_summary.csv
