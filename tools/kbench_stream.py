#!/usr/bin/env python3
"""Streaming-kernel ceiling on this box: graph-replayed launches of torch's copy / add and of this library's elementwise kernels on
token tensors of the full-resolution level, warm (same buffers every launch: Infinity-Cache resident when small) and cold (a ring of
buffers larger than the cache).  GB/s of bytes read + written."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "adnm-unet_amd"))
import torch
from adnm_hip import ops, lib

dev = "cuda"
reps = 20


def graph_time(fn, flush=None):
    fn(0)
    torch.cuda.synchronize()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        fn(0)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            for r in range(reps):
                fn(r)
    torch.cuda.current_stream().wait_stream(s)
    g.replay()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(3):
        if flush is not None:
            flush.zero_()
        torch.cuda.synchronize()
        e0.record()
        g.replay()
        e1.record()
        e1.synchronize()
        best = min(best, 1e3 * e0.elapsed_time(e1) / reps)
    return best


FLUSH = torch.empty(1 << 28, dtype=torch.float32, device=dev)
one = torch.ones(1, device=dev)
for M, C in [(65536, 32), (65536, 64), (65536, 128), (65536, 256), (16384, 256), (4096, 512)]:
    n = M * C
    for mode in ("warm", "cold"):
        nv = 1 if mode == "warm" else reps
        xs = [torch.randn(M, C, device=dev) for _ in range(nv)]
        ys = [torch.randn(M, C, device=dev) for _ in range(nv)]
        zs = [torch.empty(M, C, device=dev) for _ in range(nv)]
        fl = FLUSH if mode == "cold" else None
        with torch.no_grad():
            t_copy = graph_time(lambda r: zs[r % nv].copy_(xs[r % nv]), fl)
            t_add = graph_time(lambda r: torch.add(xs[r % nv], ys[r % nv], out=zs[r % nv]), fl)
            t_lin = graph_time(lambda r: ops.lincomb([xs[r % nv], ys[r % nv]], [one, one]), fl)
            t_act = graph_time(lambda r: ops.act(xs[r % nv], lib.ACT_GELU), fl)
        print(f"M={M:6d} C={C:4d} ({4 * n / 1e6:5.1f} MB/tensor) {mode}: copy {t_copy:6.1f} us {8 * n / t_copy / 1e3:6.0f} GB/s | torch add {t_add:6.1f} us "
              f"{12 * n / t_add / 1e3:6.0f} | lincomb(2) {t_lin:6.1f} us {12 * n / t_lin / 1e3:6.0f} | gelu {t_act:6.1f} us {8 * n / t_act / 1e3:6.0f}", flush=True)
