#!/bin/bash
# usage (GPU box): tools/smi_sample.sh <seconds> <outfile> : sample clocks / power / temperature every 0.2 s while something runs
end=$((SECONDS + $1))
while [ $SECONDS -lt $end ]; do
  rocm-smi --showclocks --showpower --showtemp --csv 2>/dev/null | tail -n +1 | tr '\n' ' ' >> $2
  echo >> $2
  sleep 0.2
done
