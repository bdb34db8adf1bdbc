# phase elimination / structure variants of k_proj_bwd_q (GPU box): one build per (PBQ_SKIP, PBQ_VAR), HIP-event time per launch
cd $GRAFT_REPO_ROOT/tools
for v in ${VARS:-0}; do
for m in ${MASKS:-0 1 2 4 8 16 32 64 128 255}; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -DPBQ_SKIP=$m -DPBQ_VAR=$v -Wno-unused-value pbq_bench.hip -o /tmp/pbq_${m}_$v 2>/dev/null && echo "PBQ_SKIP=$m PBQ_VAR=$v: $(/tmp/pbq_${m}_$v $GRID)"
done
done
