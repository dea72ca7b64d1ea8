#!/usr/bin/env python3
"""Does a kernel cost more when its code is cold?  The same small launch (row-norm backward over 256 rows of 2048 features: 36 KB of
straight-line code, 2 MB of data) replayed (a) back to back — instruction cache warm after the first — and (b) with other kernels of the
step in between, as inside a training step where ~700 different launches run once each.   python tools/kbench_icache.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "adnm-unet_amd"))
import torch
from adnm_hip import ops, lib

dev = "cuda"
REPS = 20


def graph_of(fn):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        fn()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            fn()
    torch.cuda.synchronize()
    return g


def time_graph(g, n=5):
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


for M, d in [(256, 2048), (1024, 1024), (64, 4096)]:
    x = torch.randn(M, d, device=dev, requires_grad=True)
    w = torch.ones(d, device=dev, requires_grad=True)
    cot = torch.randn(M, d, device=dev)
    # the kernel under test through the C-ABI wrapper the autograd node uses (no autograd engine inside the capture)
    xd, wd = x.detach(), w.detach()
    _, mu, rstd = ops.k_rownorm_fwd(xd, wd, None, None, None, 1e-5, False)

    def bwd():
        ops.k_rownorm_bwd(cot, xd, wd, None, None, mu, rstd, False, False, False)

    # fillers: other kernels with sizeable code (short GEMMs of different shapes, a depthwise stencil, a dense conv), tiny problems
    a1, w1 = torch.randn(64, 1024, device=dev), torch.randn(2048, 1024, device=dev)
    a2, w2 = torch.randn(256, 512, device=dev), torch.randn(1024, 512, device=dev)
    xt = torch.randn(1, 64, 64, device=dev)
    taps = torch.randn(64, 1, 3, 3, device=dev)
    cw = torch.randn(64, 64, 3, 3, device=dev) * 0.1

    def fillers():
        ops.linear(a1, w1, None)
        ops.linear(a2, w2, None)
        ops.dwconv(xt, taps, None, 8, 8, lib.ACT_GELU)
        ops.conv3(xt, cw, None, 8, 8, lib.ACT_GELU)

    def same():
        for _ in range(REPS):
            bwd()

    def mixed():
        for _ in range(REPS):
            bwd()
            fillers()

    def only_fill():
        for _ in range(REPS):
            fillers()

    t_same = time_graph(graph_of(same)) / REPS
    t_mixed, t_fill = time_graph(graph_of(mixed)) / REPS, time_graph(graph_of(only_fill)) / REPS
    print(f"rownorm_bwd M={M} d={d}: back to back {t_same:6.1f} us per launch (backward = 1-2 launches) | between other kernels {t_mixed - t_fill:6.1f} us "
          f"(mixed {t_mixed:.1f} - fillers alone {t_fill:.1f})")
