"""Which kernel instantiations of the code object does anything launch?  Compares the kernels of the built library
(profiles/<tag>_regs_table.txt: one line per instantiation) with the kernel names in rocprofv3 kernel-stats CSVs - the GPU test
suite run under `rocprofv3 --kernel-trace --stats` plus the bench workloads' committed stats.
usage: python tools/kernel_coverage.py <regs_table.txt> <kernel_stats.csv> [more.csv ...]"""
import collections, csv, re, subprocess, sys
tab = open(sys.argv[1]).read().splitlines()
syms = [l.split("\t")[0] for l in tab]
dem = subprocess.run(["c++filt"], input="\n".join(syms), capture_output=True, text=True).stdout.splitlines()
norm = lambda n: re.sub(r"\s+", "", re.sub(r"^void ", "", n))
launched = collections.Counter()
for f in sys.argv[2:]:
    for r in csv.DictReader(open(f)):
        launched[norm(r["Name"])] += int(r["Calls"])
never = [d for d in dem if norm(d) not in launched]
print(f"{len(dem)} kernels in the code object; {len(dem) - len(never)} launched; {len(never)} never launched")
fam = collections.Counter(re.match(r"(void )?(\w+)", d).group(2) for d in never)
allf = collections.Counter(re.match(r"(void )?(\w+)", d).group(2) for d in dem)
for k, v in fam.most_common():
    print(f"  {v:3d} of {allf[k]:3d}  {k}")
print()
for d in never:
    print("never:", d)
