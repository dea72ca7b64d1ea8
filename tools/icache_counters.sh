#!/bin/bash
# tools/icache_counters.sh <out.txt> : instruction-cache behaviour of every kernel of the replayed training step (bench.py under two
# rocprofv3 --pmc passes): requests / hits / misses of the instruction cache and the mean instruction-fetch latency per kernel, averaged
# over the dispatches of the run, largest miss totals first.  Run on the GPU box from the repo root.
set -o pipefail
out=${1:-gpurun_out/icache_counters.txt}
root=$(pwd)
export TMPDIR=/tmp
tmp=$root/gpurun_out/icpmc
mkdir -p $tmp
( cd /tmp && rocprofv3 --kernel-trace --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE --output-format csv -d $tmp/a -o a -- python3 $root/bench.py --steps 2 --warmup 2 --prof-steps 0 --no-cpu-baseline > $tmp/a.log 2>&1 ) || { echo "pass a failed"; tail -n 5 $tmp/a.log; }
( cd /tmp && rocprofv3 --kernel-trace --pmc SQ_IFETCH SQ_IFETCH_LEVEL SQ_BUSY_CYCLES SQ_WAVES --output-format csv -d $tmp/b -o b -- python3 $root/bench.py --steps 2 --warmup 2 --prof-steps 0 --no-cpu-baseline > $tmp/b.log 2>&1 ) || { echo "pass b failed"; tail -n 5 $tmp/b.log; }
python3 - "$tmp" > $out <<'PY'
import csv, glob, sys, collections
tmp = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for tag in ("a", "b"):
    for f in glob.glob(f"{tmp}/{tag}/**/*counter_collection.csv", recursive=True):
        per = collections.defaultdict(float)
        for row in csv.DictReader(open(f)):
            short = row["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0]
            per[(row["Dispatch_Id"], short, row["Counter_Name"])] += float(row["Counter_Value"])
        for (disp, name, ctr), v in per.items():
            acc[name][ctr].append(v)
print("# per kernel, mean per dispatch over the run (sums over the XCD / SE instances): instruction-cache requests, hits, misses (+ duplicates), hit rate;")
print("# SQ_IFETCH_LEVEL / SQ_IFETCH = mean fetch latency in cycles; SQ_BUSY_CYCLES for scale.  Sorted by total misses.")
rows = []
for name, c in acc.items():
    m = lambda k: sum(c[k]) / len(c[k]) if c.get(k) else 0.0
    n = len(c.get("SQC_ICACHE_REQ", c.get("SQ_IFETCH", [])))
    rows.append((m("SQC_ICACHE_MISSES") * n, name, n, m("SQC_ICACHE_REQ"), m("SQC_ICACHE_HITS"), m("SQC_ICACHE_MISSES"), m("SQC_ICACHE_MISSES_DUPLICATE"),
                 m("SQ_IFETCH"), m("SQ_IFETCH_LEVEL"), m("SQ_BUSY_CYCLES"), m("SQ_WAVES")))
rows.sort(reverse=True)
print(f"{'kernel':60s} {'disp':>5s} {'req':>9s} {'hits':>9s} {'miss':>8s} {'dup':>8s} {'hit%':>6s} {'ifetch':>9s} {'lat cyc':>8s} {'busy cyc':>10s} {'waves':>7s}")
for tot, name, n, req, hit, mis, dup, ife, lev, busy, waves in rows[:90]:
    print(f"{name[:60]:60s} {n:5d} {req:9.0f} {hit:9.0f} {mis:8.0f} {dup:8.0f} {100 * hit / max(req, 1):6.1f} {ife:9.0f} {lev / max(ife, 1):8.1f} {busy:10.0f} {waves:7.0f}")
PY
rm -rf $tmp
head -n 30 $out
