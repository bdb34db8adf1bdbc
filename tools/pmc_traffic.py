#!/usr/bin/env python3
"""Summarise rocprofv3 PMC passes into profiles/<tag>_pmc_traffic.json (HBM bytes per launch).

  cd /tmp && export TMPDIR=/tmp
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_FETCH_SIZE -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --profile-steps 0
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_WRITE_SIZE -- python3 bench.py ...   (separate pass)
  python tools/pmc_traffic.py gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE profiles/r01_pmc_traffic.json

Units / corrections (MI355X_MICROARCH.md, HBM section): both counters are in units of 1024 B; on gfx950
FETCH_SIZE reports exactly half of a wide coalesced streaming read -> doubled; WRITE_SIZE is exact.
"""
import collections
import csv
import glob
import json
import sys


def load(directory, counter):
    files = glob.glob(directory + "/*/*counter_collection.csv")
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(files[0])):
        if r["Counter_Name"] == counter:
            agg[r["Kernel_Name"].split("(")[0].replace("void ", "")].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in agg.items()}


def main():
    fdir, wdir, out = sys.argv[1:4]
    fe, wr = load(fdir, "FETCH_SIZE"), load(wdir, "WRITE_SIZE")
    kernels = {k: dict(fetch_bytes=2 * fe[k] * 1024, write_bytes=wr.get(k, 0.0) * 1024,
                       raw_FETCH_SIZE=fe[k], raw_WRITE_SIZE=wr.get(k, 0.0))
               for k in fe if k.startswith("k_")}
    sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
    import bench
    workload = sys.argv[4] if len(sys.argv) > 4 else "fno2d_128x128_w64_m12_b64"
    json.dump(dict(workload=workload, source_hash=bench.source_hash(), git_sha=bench.git_sha(), note="per-launch averages; FETCH_SIZE doubled (gfx950 correction), WRITE_SIZE exact; separate --pmc passes",
                   kernels=kernels), open(out, "w"), indent=1)


if __name__ == "__main__":
    main()
