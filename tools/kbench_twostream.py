#!/usr/bin/env python3
"""Do two hipGraphs replayed on two streams run side by side on this box?  A = a chain of small dependent launches (the deep levels'
input-gradient chain), B = a few large streaming launches (the weight-gradient leaves)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "adnm-unet_amd"))
import torch
from adnm_hip import ops, lib
dev = "cuda"
one = torch.ones(1, device=dev)
small = [torch.randn(256, 512, device=dev) for _ in range(2)]
big = [torch.randn(65536, 192, device=dev) for _ in range(2)]


def capture(fn, stream):
    fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(stream):
        fn()
        with torch.cuda.graph(g, stream=stream):
            fn()
    torch.cuda.synchronize()
    return g


def chain():
    x = small[0]
    for _ in range(200):
        x = ops.lincomb([x, small[1]], [one, one])
    return x


def stream_work():
    for _ in range(40):
        ops.lincomb([big[0], big[1]], [one, one])


s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
gA, gB = capture(chain, s1), capture(stream_work, s2)


def timed(fn, n=10):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(n):
        torch.cuda.synchronize()
        e0.record()
        fn()
        e1.record()
        e1.synchronize()
        best = min(best, e0.elapsed_time(e1))
    return best


def both():
    cur = torch.cuda.current_stream()
    s1.wait_stream(cur); s2.wait_stream(cur)
    with torch.cuda.stream(s1):
        gA.replay()
    with torch.cuda.stream(s2):
        gB.replay()
    cur.wait_stream(s1); cur.wait_stream(s2)


def only(g, s):
    cur = torch.cuda.current_stream()
    s.wait_stream(cur)
    with torch.cuda.stream(s):
        g.replay()
    cur.wait_stream(s)


ta, tb, tab = timed(lambda: only(gA, s1)), timed(lambda: only(gB, s2)), timed(both)
print(f"chain alone {ta:.3f} ms | streaming alone {tb:.3f} ms | both on two streams {tab:.3f} ms (sum {ta + tb:.3f}, max {max(ta, tb):.3f})")
