#!/usr/bin/env python3
"""tools/kseq.py <trace_dir>: ordered kernel sequence of the last training step (name, blocks, us)."""
import csv, re, sys, glob
d = sys.argv[1]
f = (glob.glob(d + '/*/*_kernel_trace.csv') + glob.glob(d + '/*_kernel_trace.csv'))[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
marks = [i for i, r in enumerate(rows) if "adamw_kernel" in r["Kernel_Name"] or "adamw_seg_kernel" in r["Kernel_Name"]]
lo, hi = marks[-2] + 1, marks[-1] + 1
for r in rows[lo:hi]:
    wg = int(r["Workgroup_Size_X"]) * int(r["Workgroup_Size_Y"]) * int(r["Workgroup_Size_Z"])
    nb = int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"]) // max(wg, 1)
    n = re.sub(r"^void ", "", r["Kernel_Name"]); n = re.sub(r"\(anonymous namespace\)::", "", n); n = re.sub(r"at::native::", "", n).split("(")[0][:70]
    print(f"{(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3:7.1f} us {nb:6d} blk  {n}")
