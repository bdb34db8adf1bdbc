FNO_EXTRA_FLAGS="-DFNO_TRACE -DFNO_TRACE_WHICH=3" python -m pde_policylearning_amd.build --force > /dev/null 2>&1
python tools/trace_mid.py 2>&1 | grep "^wave"
python -m pde_policylearning_amd.build --force > /dev/null 2>&1
python tools/ab.py --n 6 - FNO_NO_FUSED_MID=1
