#!/usr/bin/env python3
"""Per-kernel HBM traffic per training step from two rocprofv3 --pmc passes of bench.py (FETCH_SIZE, WRITE_SIZE: they do not
fit one pass on gfx950).

   tools/pmc_traffic.py <fetch_dir> <write_dir> <out.json> [commit]

Corrections as /opt/skills/guides/MI355X_MICROARCH.md prescribes: both counters count units of 1024 B; on gfx950 FETCH_SIZE
tallies 128-B requests at 64 B, so the streaming reads of these kernels are doubled; WRITE_SIZE is exact.  Values are summed
over the per-XCD counter instances of a dispatch.  Whole steps are cut out of the dispatch sequence at the once-per-step
`adamw_kernel` launches (the first step — lazy allocations, graph capture — is dropped).

Output (the format bench.py's pmc_table() reads):
   {"csrc_hash": ..., "commit": ..., "steps_measured": n,
    "kernels": {<profiler scope>: {"hbm_bytes_per_step", "fetch_bytes_per_step", "write_bytes_per_step", "launches_per_step"}},
    "symbols": {<kernel symbol>: the same, per kernel function}}
A profiler scope (csrc/*.hip ADNM_PROF name, the key of bench.py's `kernels` table) is matched to kernel symbols with SCOPES
below; every second-stage fold runs in fold_rows_kernel and is reported under the one scope "fold_batch"."""
import collections
import csv
import glob
import hashlib
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# every scope whose launch is the shared fold_rows_kernel: one PMC row for the group ("a|b|c" keys = the kernel symbol serves all of them)
FOLDS = "|".join(["fold_batch", "skgemm_fold", "ssd_head_fold", "lincomb_bwd_fold", "mixnorm_bwd_fold", "igate_bwd_fold", "rainloss_fold", "grad_sumsq_fold",
                  "dwconv_wgrad_fold", "rownorm_bwd_fold", "tsgemm_tn_fold", "conv3_wgrad_fold", "catmix_bwd_fold", "colsum", "adn_prep_bwd_fold",
                  "skip_vec_fold", "skip_scal_fold", "skip_wgrad_fold", "bridge_heads_fold", "bridge_pool_fold", "instnorm_bwd_scalar", "swish_bwd_fold"])
# (regex on the demangled kernel symbol, profiler scope); first match wins
SCOPES = [
    (r"^skgemm_kernel<true, true", "skgemm_nt"), (r"^skgemm_kernel<true, false", "skgemm_nn"), (r"^skgemm_kernel<false, false|^skgemm_tn_multi_kernel", "skgemm_tn"),
    (r"^lgemm_kernel<false", "skgemm_nt"), (r"^lgemm_kernel<true", "skgemm_nn"),   # the LDS-tiled half of adnm_skgemm (B_OC = op NN)
    (r"^colsum_partial_kernel", "colsum_partial"),
    (r"^tsgemm_nt_kernel", "tsgemm_nt"), (r"^tsgemm_tn_kernel|^tsgemm_tn_multi_kernel", "tsgemm_tn"),
    (r"^dwconv_kernel<float, 3", "dwconv_k3"), (r"^dwconv_kernel<float, 5", "dwconv_k5"),
    (r"^dwconv_wgrad3_roll_kernel|^dwconv_wgrad_kernel<float, 3|^dwconv_wgrad_multi_kernel<3", "dwconv_wgrad_k3"),
    (r"^dwconv_wgrad_kernel<float, 5|^dwconv_wgrad_multi_kernel<5|^dwconv_wgrad5_walk", "dwconv_wgrad_k5"),
    (r"^adn_prep_fwd_multi_kernel", "adn_prep_fwd"), (r"^adn_prep_bwd_multi_kernel", "adn_prep_bwd"),
    (r"^wt_prep_fwd_multi_kernel", "wt_prep_fwd"), (r"^wt_prep_bwd_multi_kernel", "wt_prep_bwd"),
    (r"^igate_res_fwd_kernel", "igate_fwd"), (r"^igate_res_bwd_kernel", "igate_bwd"), (r"^conv1d3_kernel", "conv1d3|conv1d3_bwd"),
    (r"^ssd_kv_kernel<.*true>$", "ssd_kv"), (r"^ssd_kv_kernel<.*false>$", "ssd_dkv"),
    (r"^ssd_apply_kernel<.*true>$", "ssd_apply_ln"), (r"^ssd_apply_kernel<.*false>$", "ssd_apply"),
    (r"^ssd_bwd_kernel", "ssd_bwd"), (r"^ssd_fold_kernel", "ssd_fold"), (r"^ssd_bc_fold_kernel", "ssd_bc_fold"),
    (r"^fold_rows_kernel", FOLDS),
    (r"^conv3_kernel", "conv3_fwd|conv3_dgrad"), (r"^conv3_wgrad_kernel", "conv3_wgrad"), (r"^conv3_join_kernel", "conv3_join"),
    (r"^adamw_(seg_)?kernel", "adamw_update"), (r"^sumsq_partial_kernel", "grad_sumsq"),
    (r"^haar_dwt_kernel", "haar_dwt"), (r"^haar_idwt_kernel", "haar_idwt"),
    (r"^(\w+?)_kernel", None),   # default: the symbol's stem is the scope (rownorm_fwd, lincomb_bwd, gate_fwd, instnorm_apply, ...)
]


def scope_of(sym):
    for rx, scope in SCOPES:
        m = re.search(rx, sym)
        if m:
            return scope if scope else m.group(1)
    return None


def clean(name):
    n = re.sub(r"^void ", "", name)
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    return n.split("(")[0].strip()


def load(d, counter):
    """-> list of (dispatch id, symbol, counter value summed over instances), in dispatch order"""
    fs = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
    acc = collections.OrderedDict()
    for f in fs:
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter:
                continue
            k = (int(r["Dispatch_Id"]), clean(r["Kernel_Name"]))
            acc[k] = acc.get(k, 0.0) + float(r["Counter_Value"])
    return sorted((d_, s, v) for (d_, s), v in acc.items())


def per_step(rows, scale):
    """cut whole steps at the adamw launches, drop the first step; -> (steps, {symbol: [bytes per step, launches per step]})"""
    cuts = [i for i, (_, s, _) in enumerate(rows) if s.startswith(("adamw_kernel", "adamw_seg_kernel"))]
    if len(cuts) < 3:
        raise SystemExit("need at least 3 training steps in the PMC pass")
    lo, hi = cuts[0] + 1, cuts[-1] + 1
    steps = len(cuts) - 1
    out = collections.defaultdict(lambda: [0.0, 0.0])
    for _, s, v in rows[lo:hi]:
        out[s][0] += scale * v / steps
        out[s][1] += 1.0 / steps
    return steps, out


def csrc_hash():
    h = hashlib.sha256()
    d = os.path.join(ROOT, "adnm-unet_amd", "csrc")
    for f in sorted(os.listdir(d)):
        if f.endswith((".hip", ".h", ".inc")):
            h.update(f.encode())
            h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]


def by_scope(symbols):
    kernels = {}
    for s, row in symbols.items():
        sc = scope_of(s)
        if sc:
            k = kernels.setdefault(sc, {"hbm_bytes_per_step": 0.0, "fetch_bytes_per_step": 0.0, "write_bytes_per_step": 0.0, "launches_per_step": 0.0})
            for f in k:
                k[f] += row[f]
    return kernels


def report(out):
    symbols, kernels = out["symbols"], out["kernels"]
    tot = sum(v["hbm_bytes_per_step"] for v in symbols.values())
    print(f"{out['steps_measured']} steps; {tot / 1e9:.3f} GB of HBM traffic per step over {sum(v['launches_per_step'] for v in symbols.values()):.0f} launches")
    for s, v in sorted(kernels.items(), key=lambda kv: -kv[1]["hbm_bytes_per_step"])[:40]:
        print(f"{v['hbm_bytes_per_step'] / 1e6:10.2f} MB/step  {v['launches_per_step']:7.1f} launches  {s[:100]}")


def main():
    if sys.argv[1] == "--rescope":   # re-derive the per-scope table of an existing file from its per-symbol table (SCOPES changed)
        out = json.load(open(sys.argv[2]))
        out["kernels"] = by_scope(out["symbols"])
        json.dump(out, open(sys.argv[2], "w"), indent=1)
        report(out)
        return
    fdir, wdir, outp = sys.argv[1:4]
    commit = sys.argv[4] if len(sys.argv) > 4 else None
    nf, fetch = per_step(load(fdir, "FETCH_SIZE"), 2.0 * 1024.0)   # x2: the gfx950 FETCH_SIZE correction
    nw, write = per_step(load(wdir, "WRITE_SIZE"), 1024.0)
    symbols = {}
    for s in sorted(set(fetch) | set(write)):
        fb, fl = fetch.get(s, [0.0, 0.0])
        wb, wl = write.get(s, [0.0, 0.0])
        symbols[s] = {"hbm_bytes_per_step": fb + wb, "fetch_bytes_per_step": fb, "write_bytes_per_step": wb, "launches_per_step": max(fl, wl)}
    out = {"csrc_hash": csrc_hash(), "commit": commit, "steps_measured": min(nf, nw),
           "method": "rocprofv3 --pmc FETCH_SIZE (x2 on gfx950) and --pmc WRITE_SIZE in separate passes, x1024 B, summed over XCD instances, "
                     "whole steps cut at adamw_kernel, first step dropped",
           "kernels": by_scope(symbols), "symbols": symbols}
    json.dump(out, open(outp, "w"), indent=1)
    report(out)


if __name__ == "__main__":
    main()
