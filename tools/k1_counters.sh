#!/bin/bash
# tools/k1_counters.sh <out.txt> : SQ counters of the K1 kernels at the refiner shape (tools/kbench_ssd.py under two rocprofv3 --pmc passes,
# per-dispatch sums over the 8 XCDs), the same counters as profiles/r02_k1_counters.txt.  Run on the GPU box from the repo root.
set -o pipefail
out=${1:-gpurun_out/k1_counters.txt}
root=$(pwd)
export TMPDIR=/tmp
tmp=$root/gpurun_out/k1pmc
mkdir -p $tmp
( cd /tmp && REPS=4 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU --output-format csv -d $tmp/a -o a -- python3 $root/tools/kbench_ssd.py > $tmp/a.log 2>&1 )
( cd /tmp && REPS=4 rocprofv3 --kernel-trace --pmc SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_MFMA --output-format csv -d $tmp/b -o b -- python3 $root/tools/kbench_ssd.py > $tmp/b.log 2>&1 )
python3 - "$tmp" > $out <<'PY'
import csv, glob, sys, collections
tmp = sys.argv[1]
print("# K1 kernels at the refiner shape (B=4, L=16384, H=16, P=4, N=16, G=2): rocprofv3 --pmc over tools/kbench_ssd.py, per-dispatch sums over the 8 XCDs")
for tag in ("a", "b"):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(f"{tmp}/{tag}/**/*counter_collection.csv", recursive=True):
        per = collections.defaultdict(float)
        for row in csv.DictReader(open(f)):
            name = row["Kernel_Name"]
            if "ssd_" not in name:
                continue
            short = name.replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0]
            per[(row["Dispatch_Id"], short, row["Counter_Name"])] += float(row["Counter_Value"])
        for (disp, name, ctr), v in per.items():
            acc[name][ctr].append(v)
    for name in sorted(acc):
        print(f"{name}:")
        for ctr in sorted(acc[name]):
            vals = acc[name][ctr]
            print(f"   {ctr:28s} {sum(vals) / len(vals):14.0f} per dispatch")
PY
rm -rf $tmp
tail -n 5 $out
