#!/usr/bin/env python3
"""tools/ktop.py <rocprof_dir> [n] [pattern]: the n longest individual kernel launches of the last training step (name, grid, us)."""
import csv, glob, re, sys
d = sys.argv[1]; n = int(sys.argv[2]) if len(sys.argv) > 2 else 40; pat = sys.argv[3] if len(sys.argv) > 3 else ""
f = (glob.glob(d + '/*/*_kernel_trace.csv') + glob.glob(d + '/*_kernel_trace.csv'))[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
marks = [i for i, r in enumerate(rows) if "adamw_kernel" in r["Kernel_Name"]]
seg = rows[marks[-2] + 1:marks[-1] + 1]
out = []
for r in seg:
    wg = int(r["Workgroup_Size_X"]) * int(r["Workgroup_Size_Y"]) * int(r["Workgroup_Size_Z"])
    nb = int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"]) // max(wg, 1)
    nm = re.sub(r"^void ", "", r["Kernel_Name"]); nm = re.sub(r"\(anonymous namespace\)::", "", nm).split("(")[0][:60]
    if pat in nm:
        out.append(((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, nb, wg, nm))
print(f"{len(seg)} launches in the step; total {sum(int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in seg) / 1e6:.3f} ms of kernel time")
for us, nb, wg, nm in sorted(out, reverse=True)[:n]:
    print(f"{us:8.1f} us  {nb:6d} blk x {wg:4d}  {nm}")
