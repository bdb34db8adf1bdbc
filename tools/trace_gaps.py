"""Timeline of one training step from a rocprofv3 --kernel-trace CSV: per kernel start / duration / gap to the end of the
previous kernel (negative = overlap), for the last complete step.  usage: python tools/trace_gaps.py <kernel_trace.csv> [anchor]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
anchor = sys.argv[2] if len(sys.argv) > 2 else "k_adam"
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if r["Kernel_Name"].startswith(anchor)]
a, b = idx[-3], idx[-2]
step = rows[a + 1:b + 1]
t0 = int(step[0]["Start_Timestamp"])
prev_end = int(rows[a]["End_Timestamp"])
busy = 0
tot_gap = 0
for r in step:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = s - prev_end
    print(f"{(s - t0) / 1e3:9.1f} us  dur {(e - s) / 1e3:7.1f}  gap {gap / 1e3:7.1f}  q{r.get('Queue_Id', '?'):>2s}  {r['Kernel_Name'][:70]}")
    tot_gap += max(gap, 0)
    prev_end = max(prev_end, e)
print(f"step span {(prev_end - int(rows[a]['End_Timestamp'])) / 1e3:.1f} us, sum of positive gaps {tot_gap / 1e3:.1f} us")
