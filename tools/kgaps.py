#!/usr/bin/env python3
"""tools/kgaps.py <trace_dir>: largest idle gaps between consecutive kernels inside the last training steps."""
import csv, sys, glob
d = sys.argv[1]
f = (glob.glob(d + '/*/*_kernel_trace.csv') + glob.glob(d + '/*_kernel_trace.csv'))[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
marks = [i for i, r in enumerate(rows) if "adamw_kernel" in r["Kernel_Name"]]
for a, b in zip(marks[-4:-1], marks[-3:]):
    seg = rows[a:b + 1]
    gaps = sorted(((int(seg[i + 1]["Start_Timestamp"]) - int(seg[i]["End_Timestamp"])) / 1e3, i, seg[i]["Kernel_Name"][:50], seg[i + 1]["Kernel_Name"][:50])
                  for i in range(len(seg) - 1))
    wall = (int(seg[-1]["End_Timestamp"]) - int(seg[0]["End_Timestamp"])) / 1e3
    busy = sum((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in seg[1:])
    print(f"step: wall {wall:.0f} us, kernels {busy:.0f} us, {len(seg)-1} launches; largest gaps:")
    for g, i, n0, n1 in gaps[-4:]:
        print(f"   {g:8.1f} us after launch #{i}: {n0} -> {n1}")
