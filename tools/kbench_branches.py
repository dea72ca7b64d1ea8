#!/usr/bin/env python3
"""Three INDEPENDENT chains of small dependent launches (the three EncoderToDecoder modules of the model are such chains): (a) one captured graph,
the chains one after the other; (b) one captured graph with the chains as parallel branches (forked / joined with events inside the capture);
(c) three captured graphs replayed on three streams.   python tools/kbench_branches.py [launches per chain]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "adnm-unet_amd"))
import torch
from adnm_hip import ops

dev = "cuda"
N = int(sys.argv[1]) if len(sys.argv) > 1 else 30
one = torch.ones(1, device=dev)
data = [[torch.randn(256, 512, device=dev) for _ in range(2)] for _ in range(3)]
w = [torch.randn(512, 512, device=dev) * 0.05 for _ in range(3)]


def chain(k):
    x = data[k][0]
    for i in range(N):
        x = ops.lincomb([x, data[k][1]], [one, one]) if i % 3 else ops.linear(x, w[k], None)   # elementwise and short-GEMM launches
    return x


def timed(replay, n=20):
    replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


with torch.no_grad():
    for k in range(3):
        chain(k)
    torch.cuda.synchronize()
    main = torch.cuda.Stream()
    side = [torch.cuda.Stream() for _ in range(3)]
    scope = ops.SPLITWS.open_scope(torch.device(dev, torch.cuda.current_device()))
    # (a) serial
    ga = torch.cuda.CUDAGraph()
    with ops.SPLITWS.capturing(scope), torch.cuda.stream(main):
        with torch.cuda.graph(ga, stream=main):
            for k in range(3):
                chain(k)
    # (b) branches inside one capture
    gb = torch.cuda.CUDAGraph()
    with ops.SPLITWS.capturing(scope), torch.cuda.stream(main):
        with torch.cuda.graph(gb, stream=main):
            for k in (1, 2):
                side[k].wait_stream(main)
            chain(0)
            for k in (1, 2):
                with torch.cuda.stream(side[k]):
                    chain(k)
            for k in (1, 2):
                main.wait_stream(side[k])
    # (c) three graphs, three streams (each its own split-K scope: they run at the same time)
    gc, scopes = [], [ops.SPLITWS.open_scope(torch.device(dev, torch.cuda.current_device())) for _ in range(3)]
    for k in range(3):
        g = torch.cuda.CUDAGraph()
        with ops.SPLITWS.capturing(scopes[k]), torch.cuda.stream(side[k]):
            with torch.cuda.graph(g, stream=side[k]):
                chain(k)
        gc.append(g)
    torch.cuda.synchronize()

    def rep_c():
        for k in range(3):
            side[k].wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side[k]):
                gc[k].replay()
        for k in range(3):
            torch.cuda.current_stream().wait_stream(side[k])

    ta, tb, tc = timed(ga.replay), timed(gb.replay), timed(rep_c)
    t1 = timed(lambda: gc[0].replay())
    print(f"{N} launches per chain: one chain alone {t1:.1f} us | three chains in one serial graph {ta:.1f} us | as branches of one graph {tb:.1f} us | "
          f"three graphs on three streams {tc:.1f} us")
