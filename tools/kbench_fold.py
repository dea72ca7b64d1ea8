#!/usr/bin/env python3
"""The shared second-stage fold on partial sets of the sizes a training step folds (rows x columns), four columns per lane (16-byte aligned
set) against one column per lane (the same set 4 bytes off alignment): us per launch and GB/s, graph-replayed.   python tools/kbench_fold.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "adnm-unet_amd"))
import torch
from adnm_hip import ops

dev = "cuda"
REPS = 10


def timed(fn):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        fn()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            for _ in range(REPS):
                fn()
    torch.cuda.synchronize()
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / 5 / REPS


for rows, n in [(2, 4784128), (4, 4784128), (8, 1048576), (16, 262144), (64, 65536), (256, 4800), (256, 73728), (1024, 2048)]:
    buf = torch.randn(rows * n + 8, device=dev)
    a = buf[:rows * n].view(rows, n)
    b = buf[1:1 + rows * n].view(rows, n)
    out = torch.empty(n, device=dev)
    ta, tb = timed(lambda: ops.colsum(a, out)), timed(lambda: ops.colsum(b, out))
    mb = 4.0 * (rows + 1) * n / 1e6
    print(f"{rows:5d} x {n:8d} ({mb:6.1f} MB): aligned {ta:6.1f} us = {mb / ta / 1e3 * 1e3:6.0f} GB/s | 4 bytes off (one column per lane) {tb:6.1f} us = {mb / tb / 1e3 * 1e3:6.0f} GB/s")
