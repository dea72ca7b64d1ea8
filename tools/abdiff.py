#!/usr/bin/env python3
"""A/B of two bench.py --dump-json records: step time and the kernels whose time moved (HIP-event ms per step)."""
import json
import sys
a, b = (json.load(open(f)) for f in sys.argv[1:3])
thr = float(sys.argv[3]) if len(sys.argv) > 3 else 0.01
print(f"step: {a['ms_per_step']} -> {b['ms_per_step']} ms")
ka, kb = a["kernels"], b["kernels"]
rows = []
for k in sorted(set(ka) | set(kb)):
    x, y = ka.get(k, {"ms_per_step": 0, "launches_per_step": 0}), kb.get(k, {"ms_per_step": 0, "launches_per_step": 0})
    d = y["ms_per_step"] - x["ms_per_step"]
    if abs(d) >= thr:
        rows.append((d, k, x["launches_per_step"], y["launches_per_step"], x["ms_per_step"], y["ms_per_step"]))
for d, k, la, lb, xa, xb in sorted(rows):
    print(f"  {k:24s} {la:5.0f} -> {lb:5.0f} launches  {xa:.3f} -> {xb:.3f} ms  ({d:+.3f})")
print(f"sum of kernels: {sum(v['ms_per_step'] for v in ka.values()):.3f} -> {sum(v['ms_per_step'] for v in kb.values()):.3f}")
