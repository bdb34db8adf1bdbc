#!/bin/bash
# usage (GPU box): tools/kexp2.sh "<label>" "<hipcc -D flags>" ["ENV=1 ENV2=.."]  -> rebuild with the flags, print step time and the top kernels
FNO_EXTRA_FLAGS="$2" python -m pde_policylearning_amd.build --force > /dev/null 2>&1
env $3 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$1', d['value'], d['ms_per_step'], ' '.join(f\"{k['name']}={k['avg_ms']}\" for k in d['kernels'][:4]))"
