#!/bin/bash
# usage: tools/regs.sh <kernel-name-substring>   -> VGPRs / scratch / LDS of every instantiation (offline compile, no GPU)
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -fno-slp-vectorize -Wno-unused-value -Wno-unused-result $FNO_EXTRA_FLAGS \
  -Rpass-analysis=kernel-resource-usage -o /tmp/regs_probe.so pde_policylearning_amd/csrc/fno_abi.hip 2>&1 |
  grep -A8 "Function Name: .*$1" | grep "Function Name\|VGPRs:\|ScratchSize\|SGPRs:" |
  sed 's/.*remark: *//; s/ \[-Rpass.*//' | paste - - - - | sed 's/Function Name: //'
