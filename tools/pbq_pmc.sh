cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R/tools
hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wno-unused-value pbq_bench.hip -o /tmp/pbq_plain 2>/dev/null && /tmp/pbq_plain
hipcc --offload-arch=gfx950 -O3 -std=c++17 -DPBQ_TRACE -Wno-unused-value pbq_bench.hip -o /tmp/pbq_tr 2>/dev/null && /tmp/pbq_tr
cd /tmp
i=0
for grp in "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC"; do
  i=$((i+1))
  timeout -k 5 120 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d /tmp/pmc$i -- /tmp/pbq_plain > /dev/null 2>&1
  f=$(ls /tmp/pmc$i/*/*counter_collection.csv | head -1)
  python3 - "$f" <<'PY'
import csv,sys,collections
agg=collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if "k_proj_bwd_q" in r["Kernel_Name"]: agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,v in agg.items(): print(f"{k:28s} {sum(v)/len(v):.4g}  (n={len(v)})")
PY
done
