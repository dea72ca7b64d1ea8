#!/usr/bin/env python3
"""tools/kstat1.py <rocprof_dir> [pattern]: per-kernel count / avg / min duration (us) from a --kernel-trace csv."""
import csv, glob, re, sys, collections
d = sys.argv[1]; pat = sys.argv[2] if len(sys.argv) > 2 else ""
f = (glob.glob(d + '/*/*_kernel_trace.csv') + glob.glob(d + '/*_kernel_trace.csv'))[0]
agg = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    n = re.sub(r"^void ", "", r["Kernel_Name"]); n = re.sub(r"\(anonymous namespace\)::", "", n).split("(")[0]
    if pat in n:
        agg[n].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for n, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
    v2 = sorted(v)
    print(f"{len(v):5d} x  avg {sum(v)/len(v):8.2f}  med {v2[len(v)//2]:8.2f}  min {v2[0]:8.2f} us   {n[:100]}")
