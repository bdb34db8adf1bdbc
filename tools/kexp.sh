#!/bin/bash
# usage: tools/kexp.sh "<label>" "<extra hipcc flags>"  -> rebuilds and prints per-kernel ms
label="$1"; flags="$2"
FNO_EXTRA_FLAGS="$flags" python -m pde_policylearning_amd.build --force >/dev/null 2>&1 || { echo "$label: BUILD FAILED"; exit 1; }
python bench.py --steps 10 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
ks={k['name']:k['avg_ms'] for k in d['kernels']}
print('$label', 'fields/s', d['value'], ' '.join(f'{n}={ks.get(n)}' for n in ('k_proj_fwd','k_proj_bwd','k_block_bwd','k_pw_fwd_block','k_pw_fwd_lift')))
"
