#!/usr/bin/env python3
"""tools/kclass.py <trace_dir> [max_blocks]: kernels of the steady-state window whose grid has <= max_blocks workgroups."""
import csv, re, sys, glob, collections
d = sys.argv[1]; mx = int(sys.argv[2]) if len(sys.argv) > 2 else 64
f = (glob.glob(d + '/*/*_kernel_trace.csv') + glob.glob(d + '/*_kernel_trace.csv'))[0]
rows = list(csv.DictReader(open(f)))
marks = sorted(int(r["Start_Timestamp"]) for r in rows if "ssd_apply_kernel" in r["Kernel_Name"])
lo, hi = marks[-40], marks[-10]
rows = [r for r in rows if lo <= int(r["Start_Timestamp"]) < hi]
agg = collections.defaultdict(lambda: [0, 0])
for r in rows:
    wg = int(r["Workgroup_Size_X"]) * int(r["Workgroup_Size_Y"]) * int(r["Workgroup_Size_Z"])
    nb = int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"]) // max(wg, 1)
    if nb > mx: continue
    n = re.sub(r"^void ", "", r["Kernel_Name"]); n = re.sub(r"\(anonymous namespace\)::", "", n).split("(")[0][:80]
    a = agg[n]; a[0] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3; a[1] += 1
tot = sum(v[0] for v in agg.values()); cnt = sum(v[1] for v in agg.values())
print(f"grids <= {mx} blocks: {cnt/3:.0f} launches/step, {tot/3/1e3:.2f} ms/step")
for n, v in sorted(agg.items(), key=lambda kv: -kv[1][0])[:70]:
    print(f"{v[0]/3/1e3:7.3f} ms {v[1]/3:6.1f} calls {v[0]/v[1]:6.1f} us  {n}")
