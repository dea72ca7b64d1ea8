#!/usr/bin/env python3
"""Tall-skinny GEMM micro-benchmark (tsgemm, M >= 2048 token rows): tools/kbench_ts.py [M N K]...  Graph-replayed launches of the forward
(NT), input-gradient (NT through swapped strides) and weight-gradient (TN) kernels against the library GEMM; GB/s of algorithmic bytes."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "adnm-unet_amd"))
import torch
from adnm_hip import ops, lib

dev = "cuda"
args = sys.argv[1:]
shapes = [(int(args[i]), int(args[i + 1]), int(args[i + 2])) for i in range(0, len(args), 3)] or [
    (65536, 128, 32), (65536, 208, 32), (65536, 32, 128), (65536, 32, 64), (65536, 128, 64), (65536, 64, 128), (65536, 32, 32),
    (16384, 128, 256), (16384, 256, 128), (16384, 64, 32), (4096, 256, 512), (4096, 64, 128)]
reps = int(os.environ.get("REPS", "20"))


def graph_time(fn):
    fn()
    torch.cuda.synchronize()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        fn()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            for _ in range(reps):
                fn()
    torch.cuda.current_stream().wait_stream(s)
    g.replay()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0.record()
        g.replay()
        e1.record()
        e1.synchronize()
        best = min(best, 1e3 * e0.elapsed_time(e1) / reps)
    return best


ops.set_mfma_precision(os.environ.get("PREC", "f32"))
print("precision of our kernels:", os.environ.get("PREC", "f32"))
tot = [0.0] * 6
for M, N, K in shapes:
    x, w, dy = torch.randn(M, K, device=dev), torch.randn(N, K, device=dev) * 0.1, torch.randn(M, N, device=dev)
    byts = 4.0 * (M * (K + N) + N * K)
    t = [graph_time(lambda: ops.k_linear(x, w, None)), graph_time(lambda: ops.k_linear_dx(dy, w)), graph_time(lambda: ops.k_linear_dw(dy, x, True)),
         graph_time(lambda: torch.mm(x, w.t())), graph_time(lambda: torch.mm(dy, w)), graph_time(lambda: torch.mm(dy.t(), x))]
    e = [((ops.k_linear(x, w, None).double() - x.double() @ w.double().t()).norm() / (x.double() @ w.double().t()).norm()).item(),
         ((ops.k_linear_dx(dy, w).double() - dy.double() @ w.double()).norm() / (dy.double() @ w.double()).norm()).item()]
    for i in range(6):
        tot[i] += t[i]
    print(f"M={M:6d} N={N:4d} K={K:4d} ({byts / 1e6:5.1f} MB): fwd {t[0]:6.1f} us {byts / t[0] / 1e3:6.0f} GB/s | dx {t[1]:6.1f} us {byts / t[1] / 1e3:6.0f} | dw {t[2]:6.1f} us "
          f"{byts / t[2] / 1e3:6.0f} || library {t[3]:6.1f} {t[4]:6.1f} {t[5]:6.1f} us   err {e[0]:.1e} {e[1]:.1e}", flush=True)
print("sum: fwd %.1f dx %.1f dw %.1f | library %.1f %.1f %.1f us" % tuple(tot))
