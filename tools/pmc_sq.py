#!/usr/bin/env python3
"""Aggregate rocprofv3 SQ counter passes per kernel (average per launch).
  python tools/pmc_sq.py gpurun_out/pmc_sq1 gpurun_out/pmc_sq2 ...  -> table on stdout"""
import collections, csv, glob, sys
tab = collections.defaultdict(dict)
for d in sys.argv[1:]:
    for f in glob.glob(d + "/*/*counter_collection.csv"):
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0].replace("void ", "").split("<")[0]
            agg[(k, r["Counter_Name"])].append(float(r["Counter_Value"]))
        for (k, c), v in agg.items():
            tab[k][c] = sum(v) / len(v)
names = sorted({c for k in tab for c in tab[k]})
print("kernel," + ",".join(names))
for k in sorted(tab):
    if k.startswith("k_"):
        print(k + "," + ",".join(f"{tab[k].get(c, float('nan')):.4g}" for c in names))
