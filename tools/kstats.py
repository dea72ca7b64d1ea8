#!/usr/bin/env python3
"""Steady-state summary of a rocprofv3 --kernel-trace run of bench.py: looks only at the LAST `frac` of the
dispatches' time range (skips MIOpen find / first-touch effects) and normalises per training step using a
once-per-step marker kernel.   tools/kstats.py <dir> [top_n] [frac]"""
import csv, glob, sys
d = sys.argv[1]
top = int(sys.argv[2]) if len(sys.argv) > 2 else 30
f = (glob.glob(d + '/*/*_kernel_trace.csv') + glob.glob(d + '/*_kernel_trace.csv'))[0]
rows = list(csv.DictReader(open(f)))
nsteps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
marks = sorted(int(r['Start_Timestamp']) for r in rows if 'ssd_apply_kernel' in r['Kernel_Name'])  # 10 launches per step
per_step = 10
assert len(marks) >= per_step * (nsteps + 1), "not enough steps in the trace"
lo, hi = marks[-per_step * (nsteps + 1)], marks[-per_step]
rows = [r for r in rows if lo <= int(r['Start_Timestamp']) < hi]
steps = nsteps
def cat(name):
    if 'anonymous namespace)::' in name and 'at::native' not in name and 'ck::' not in name: return 'adnm_hip'
    if name.startswith('Cijk'): return 'tensile_gemm'
    if 'ck16tensor' in name or 'ck::' in name: return 'ck_conv'
    if 'naive_conv' in name or 'miopen' in name.lower() or 'igemm' in name or 'gridwise' in name.lower() or 'Im2' in name or 'SubTensor' in name: return 'miopen'
    if 'elementwise' in name: return 'torch_elementwise'
    if 'reduce_kernel' in name: return 'torch_reduce'
    if 'multi_tensor' in name or 'lpnorm' in name: return 'torch_foreach'
    if 'rocclr' in name: return 'copy/fill'
    if 'pool' in name: return 'torch_pool'
    if 'Cat' in name or 'gather' in name or 'index' in name: return 'torch_index/cat'
    return 'other'
agg, per = {}, {}
for r in rows:
    n = r['Kernel_Name']; dur = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6
    a = agg.setdefault(cat(n), [0, 0]); a[0] += dur; a[1] += 1
    p = per.setdefault(n, [0, 0]); p[0] += dur; p[1] += 1
tot = sum(v[0] for v in agg.values())
wall = (max(int(r['End_Timestamp']) for r in rows) - min(int(r['Start_Timestamp']) for r in rows)) / 1e6
print(f"window: {steps} steps, wall {wall/steps:.2f} ms/step, kernel time {tot/steps:.2f} ms/step, {len(rows)/steps:.0f} launches/step")
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][0]): print(f"  {k:20s} {v[0]/steps:7.2f} ms/step {v[1]/steps:7.0f} launches")
for n, v in sorted(per.items(), key=lambda kv: -kv[1][0])[:top]:
    print(f"{v[0]/steps:8.3f} ms/step {v[1]/steps:7.1f} calls {1e3*v[0]/v[1]:8.1f} us  {n[:130]}")
# ---- per (family, grid size): where in the pyramid the time goes; `real` subtracts ~2 us/launch of trace overhead
import re
fam = {}
for r in rows:
    n = re.sub(r"^void ", "", r['Kernel_Name']); n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = n.split("(")[0][:70]
    k = (n, int(r.get('Grid_Size', 0) or 0))
    a = fam.setdefault(k, [0, 0]); a[0] += (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6; a[1] += 1
print("\n(kernel, grid) by time:")
for (n, g), v in sorted(fam.items(), key=lambda kv: -kv[1][0])[:int(sys.argv[4]) if len(sys.argv) > 4 else 90]:
    print(f"{v[0]/steps:8.3f} ms/step {v[1]/steps:6.1f} calls {1e3*v[0]/v[1]:8.1f} us grid={g:9d}  {n}")
