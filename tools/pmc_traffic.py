#!/usr/bin/env python3
"""Per-kernel HBM traffic from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; they do not fit one pass on gfx950).
   tools/pmc_traffic.py <fetch_dir> <write_dir> <out.json>
Corrections as /opt/skills/guides/MI355X_MICROARCH.md prescribes: the counters are in KiB-like units of 1024 B; on gfx950
FETCH_SIZE tallies 128-B requests at 64 B, so wide streaming reads are doubled; WRITE_SIZE is exact."""
import csv, glob, json, re, sys, collections


def load(d, counter):
    f = (glob.glob(d + "/*counter_collection.csv") + glob.glob(d + "/*/*counter_collection.csv"))[0]
    per = collections.defaultdict(list)
    acc = collections.defaultdict(float)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != counter:
            continue
        acc[(r["Dispatch_Id"], r["Kernel_Name"])] += float(r["Counter_Value"])  # one row per XCD / dimension instance
    for (_, name), v in acc.items():
        n = re.sub(r"^void ", "", name)
        n = re.sub(r"\(anonymous namespace\)::", "", n).split("(")[0]
        per[n].append(v)
    return per


fetch, write = load(sys.argv[1], "FETCH_SIZE"), load(sys.argv[2], "WRITE_SIZE")
out = {}
for k in sorted(set(fetch) | set(write)):
    f, w = fetch.get(k, []), write.get(k, [])
    fb = 2.0 * 1024.0 * (sum(f) / len(f)) if f else None   # x2: gfx950 FETCH_SIZE correction
    wb = 1024.0 * (sum(w) / len(w)) if w else None
    out[k] = {"launches_seen": max(len(f), len(w)), "fetch_bytes_per_launch": fb, "write_bytes_per_launch": wb,
              "hbm_bytes_per_launch": (fb or 0.0) + (wb or 0.0)}
json.dump(out, open(sys.argv[3], "w"), indent=1)
for k, v in sorted(out.items(), key=lambda kv: -kv[1]["hbm_bytes_per_launch"] * kv[1]["launches_seen"])[:25]:
    print(f"{v['hbm_bytes_per_launch']/1e6:10.2f} MB/launch x{v['launches_seen']:5d}  {k[:90]}")
