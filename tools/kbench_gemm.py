#!/usr/bin/env python3
"""Short-GEMM micro-benchmark: tools/kbench_gemm.py  [op M N K]...   (op in NT NN TN; default: the deep-level shapes of config 2).
Each (shape, precision) is captured as a hipGraph of REPS back-to-back launches of adnm_skgemm and replayed: device-side time per
launch without host overhead (as inside the trainer's graph).  The library GEMM is timed the same way for comparison, and every
result is checked against it (fp32: rel-L2 <= 2e-6; bf16 operands: <= 1e-2)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "adnm-unet_amd"))
import torch
from adnm_hip import ops, lib

dev = "cuda"
args = sys.argv[1:]
# environment: SHAPES=<file of "op M N K [count]" lines>  OUT=<json>  NOLIB=1 (skip the library GEMM timing)  PRECS=f32,bf16
file_shapes = []
if os.environ.get("SHAPES"):
    for ln in open(os.environ["SHAPES"]):
        f = ln.split()
        if len(f) >= 4 and (not os.environ.get("OPS") or f[0] in os.environ["OPS"].split(",")):
            file_shapes.append((f[0], int(f[1]), int(f[2]), int(f[3])))
precs = os.environ.get("PRECS", "f32,bf16").split(",")
nolib = os.environ.get("NOLIB") == "1"
shapes = file_shapes or [(args[i], int(args[i + 1]), int(args[i + 2]), int(args[i + 3])) for i in range(0, len(args), 4)] or [
    ("NT", 4, 1024, 2144), ("NN", 4, 1024, 2144),
    ("NT", 64, 512, 2048), ("NT", 64, 1024, 2048), ("NN", 64, 1024, 2048), ("TN", 64, 1024, 2048),
    ("NT", 64, 4096, 1024), ("NN", 64, 4096, 1024), ("NT", 64, 4672, 1024), ("NN", 64, 4672, 1024), ("TN", 64, 4672, 1024),
    ("NT", 256, 512, 1024), ("NN", 256, 512, 1024), ("NT", 256, 1024, 4096), ("NN", 256, 1024, 4096), ("TN", 256, 1024, 4096),
    ("NT", 256, 4672, 1024), ("NN", 256, 4672, 1024),
    ("NT", 1024, 128, 256), ("NN", 1024, 256, 128), ("NT", 1024, 512, 2048), ("NN", 1024, 512, 2048), ("TN", 1024, 512, 2048),
    ("NT", 1024, 2368, 512), ("NN", 1024, 2368, 512), ("TN", 1024, 2368, 512),
]
OPC = {"NT": ops.SK_NT, "NN": ops.SK_NN, "TN": ops.SK_TN}
reps = int(os.environ.get("REPS", "20"))
# COLD=1: the weight operand of every launch comes from HBM, as in a training step (72 M parameters are streamed once per pass, AdamW
# rewrites them in between): the launches of a graph cycle over enough copies of the weight to exceed the 256 MB Infinity Cache.
# Without it the 20 launches re-read one weight that stays in L2 / the Infinity Cache — 2-3x faster than the step ever sees.
cold = os.environ.get("COLD") == "1"
FLUSH = torch.empty(1 << 28, dtype=torch.float32, device="cuda") if cold else None


def graph_time(fn, nvar=1, defer=False):
    """fn(i): launch with operand copy i (i < nvar)"""
    fn(0)
    torch.cuda.synchronize()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        fn(0)
        g = torch.cuda.CUDAGraph()
        scope = ops.SPLITWS.open_scope(torch.device(dev, torch.cuda.current_device()))   # the split launches' uncached workspace, kept with the graph
        with ops.SPLITWS.capturing(scope), torch.cuda.graph(g, stream=s):
            # the weight-gradient op defers its split-K folds exactly as under the trainer: queued, batched, flushed at the end
            with ops.FOLDS.active(torch.device(dev, torch.cuda.current_device()), defer):
                for r in range(reps):
                    fn(r % nvar)
    torch.cuda.current_stream().wait_stream(s)
    g.replay()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        if cold:   # evict L2 and the Infinity Cache: 1 GB of writes
            FLUSH.zero_()
            torch.cuda.synchronize()
        e0.record()
        g.replay()
        e1.record()
        e1.synchronize()
        best = min(best, 1e3 * e0.elapsed_time(e1) / reps)
    return best


tot = {pr: 0.0 for pr in precs}
tot["lib"] = 0.0
bad = 0
rows = []
for op, M, N, K in shapes:
    gen = lambda *s: torch.randn(*s, device=dev)
    wshape = (N, K) if op in ("NT", "NN") else (M, K)
    nvar = 1
    if cold and op != "TN":
        nvar = reps   # every launch of a replay reads its own copy; the caches are flushed between replays (graph_time)
    bs = [gen(*wshape) for _ in range(nvar)]
    if nvar > 1:
        for t in bs[1:]:
            t.copy_(bs[0])
    b = bs[0]
    if op == "NT":
        a, c = gen(M, K), torch.empty(M, N, device=dev)
        libf = lambda i=0: torch.mm(a, bs[i].t())
    elif op == "NN":
        a, c = gen(M, N), torch.empty(M, K, device=dev)
        libf = lambda i=0: torch.mm(a, bs[i])
    else:
        a, c = gen(M, N), torch.empty(N, K, device=dev)
        libf = lambda i=0: torch.mm(a.t(), bs[i])
    ref = libf().double()
    row = {"op": op, "M": M, "N": N, "K": K}
    txt = []
    for prec in precs:
        ops.set_mfma_precision(prec)
        c.fill_(float("nan"))
        try:
            t = graph_time(lambda i: ops._skgemm(OPC[op], a, bs[i], None, c, None, M, N, K, defer=(op == "TN")), nvar, defer=(op == "TN"))
        except RuntimeError as e:   # a forced configuration this op has no kernel for
            txt.append(f"sk {prec:4s}    n/a")
            continue
        err = ((c.double() - ref).norm() / ref.norm()).item()
        red = K if op == "NT" else (N if op == "NN" else M)   # fp32 summation noise grows with the reduction length
        ok = err <= ((2e-6 if red <= 4096 else 1e-5) if prec == "f32" else 1e-2)
        bad += 0 if ok else 1
        row[prec] = t if ok else None
        tot[prec] += t
        txt.append(f"sk {prec:4s} {t:6.1f} us (err {err:.1e}{'' if ok else ' BAD'})")
    ops.set_mfma_precision("f32")
    if not nolib:
        row["lib"] = graph_time(libf, nvar)
        tot["lib"] += row["lib"]
        txt.append(f"rocBLAS {row['lib']:6.1f} us")
    rows.append(row)
    flops = 2.0 * M * N * K
    byts = 4.0 * (M * K + N * K + M * N)
    print(f"{op} M={M:5d} N={N:5d} K={K:5d}: " + "  ".join(txt) + f" | ideal f32-MFMA {flops / 157e6:5.1f} us, HBM {byts / 6.3e6:4.1f} us", flush=True)
print("sum: " + ", ".join(f"{k} {v:.1f} us" for k, v in tot.items()) + f"; {bad} results out of tolerance")
if os.environ.get("OUT"):
    import json
    json.dump({"force": os.environ.get("ADNM_SK_FORCE"), "rows": rows}, open(os.environ["OUT"], "w"))
sys.exit(1 if bad else 0)
