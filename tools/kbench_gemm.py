#!/usr/bin/env python3
"""Short-GEMM micro-benchmark: tools/kbench_gemm.py  [op M N K]...   (op in NT NN TN; default: the deep-level shapes of config 2).
Back-to-back launches of adnm_skgemm in both MFMA precisions and of the library GEMM; run under rocprofv3 --kernel-trace for true
kernel durations (tools/kstat1.py)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "adnm-unet_amd"))
import torch
from adnm_hip import ops, lib

dev = "cuda"
args = sys.argv[1:]
shapes = [(args[i], int(args[i + 1]), int(args[i + 2]), int(args[i + 3])) for i in range(0, len(args), 4)] or [
    ("NT", 256, 1024, 512), ("NN", 256, 1024, 512), ("TN", 256, 1024, 512),
    ("NT", 64, 4672, 1024), ("NN", 64, 4672, 1024), ("TN", 64, 4672, 1024),
    ("NT", 1024, 512, 2048), ("NN", 1024, 512, 2048), ("TN", 1024, 512, 2048),
    ("NT", 1024, 2368, 512), ("NT", 64, 512, 2048), ("NT", 4096, 256, 512),
]
OPC = {"NT": ops.SK_NT, "NN": ops.SK_NN, "TN": ops.SK_TN}
reps = int(os.environ.get("REPS", "30"))


def timeit(fn):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    e1.synchronize()
    return 1e3 * e0.elapsed_time(e1) / reps


for op, M, N, K in shapes:
    g = lambda *s: torch.randn(*s, device=dev)
    if op == "NT":
        a, b, c = g(M, K), g(N, K), torch.empty(M, N, device=dev)
        libf = lambda: torch.mm(a, b.t())
    elif op == "NN":
        a, b, c = g(M, N), g(N, K), torch.empty(M, K, device=dev)
        libf = lambda: torch.mm(a, b)
    else:
        a, b, c = g(M, N), g(M, K), torch.empty(N, K, device=dev)
        libf = lambda: torch.mm(a.t(), b)
    res = []
    for prec in ("f32", "bf16"):
        ops.set_mfma_precision(prec)
        res.append(timeit(lambda: ops._skgemm(OPC[op], a, b, None, c, None, M, N, K)))
    ops.set_mfma_precision("f32")
    t_lib = timeit(libf)
    flops = 2.0 * M * N * K
    byts = 4.0 * (M * K + N * K + M * N)
    print(f"{op} M={M:5d} N={N:5d} K={K:5d}: sk f32 {res[0]:7.1f} us  sk bf16 {res[1]:7.1f} us  rocBLAS {t_lib:7.1f} us   | ideal f32-MFMA {flops / 157e6:6.1f} us, HBM {byts / 6.3e6:5.1f} us")
