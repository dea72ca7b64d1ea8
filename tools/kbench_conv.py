#!/usr/bin/env python3
"""The dense 3x3 convolutions of one training step (config 2: B = 4, 128 x 128) one by one: forward and input gradient through the C ABI,
graph-replayed with cold operands (each launch of a replay reads its own copy of the tokens).   python tools/kbench_conv.py [f32|bf16 ...]
ADNM_CONV3_FP32_IMAGES=1 selects the round-3 kernel (fp32 LDS images) in the bf16 configuration."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "adnm-unet_amd"))
import torch
from adnm_hip import lib

dev = "cuda"
SHAPES = [(128, 128, 5, 64), (128, 128, 64, 64), (128, 128, 128, 32), (128, 128, 20, 20), (64, 64, 64, 128), (64, 64, 128, 32),
          (32, 32, 128, 256), (32, 32, 256, 64)]
B, REPS = 4, 20
PREC = {"f32": 0, "bf16": 1}


def timed(fn):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        fn(s.cuda_stream)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            fn(s.cuda_stream)
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


for name in (sys.argv[1:] or ["f32", "bf16"]):
    prec = PREC[name]
    print(f"--- {name}: us per launch (graph of {REPS} launches on {REPS} operand copies)", flush=True)
    tot = [0.0, 0.0, 0.0]
    for H, W, K, N in SHAPES:
        M = B * H * W
        xs = [torch.randn(M, K, device=dev) for _ in range(REPS)]
        dys = [torch.randn(M, N, device=dev) for _ in range(REPS)]
        w = torch.randn(N, 3, 3, K, device=dev) * 0.1          # channels-last, as the flat trainer keeps it: strides (9K, K, 1) for (n, tap, k)
        b = torch.zeros(N, device=dev)
        y, pre, dx = torch.empty(M, N, device=dev), torch.empty(M, N, device=dev), torch.empty(M, K, device=dev)
        act = lib.ACT_GELU if K != 20 else lib.ACT_NONE
        nbf, nbd = lib.query("adnm_conv3_ws_bytes", B, H, W, K, N), lib.query("adnm_conv3_ws_bytes", B, H, W, N, K)
        wsf, wsd = torch.empty(max(nbf, 16), dtype=torch.uint8, device=dev), torch.empty(max(nbd, 16), dtype=torch.uint8, device=dev)

        def fwd(st):
            for x in xs:
                lib.call("adnm_conv3_fwd", x.data_ptr(), K, w.data_ptr(), 9 * K, K, 1, b.data_ptr(), y.data_ptr(), N, pre.data_ptr() if act else None, N,
                         wsf.data_ptr(), nbf, B, H, W, K, N, act, prec, None, st)

        def dgrad(st):
            for dy in dys:
                lib.call("adnm_conv3_dgrad", dy.data_ptr(), N, pre.data_ptr() if act else None, N, act, w.data_ptr(), 9 * K, K, 1, dx.data_ptr(), K,
                         wsd.data_ptr(), nbd, B, H, W, K, N, prec, None, st)

        nbw = lib.query("adnm_conv3_wgrad_ws_bytes", B, H, W, K, N)
        wsw = torch.empty(max(nbw, 16), dtype=torch.uint8, device=dev)
        dw, db = torch.empty(N, 3, 3, K, device=dev), torch.empty(N, device=dev)

        def wgrad(st):   # (kernel + its fold: launched directly here; a queued leaf + a batched fold under a trainer)
            for x, dy in zip(xs, dys):
                lib.call("adnm_conv3_wgrad", dy.data_ptr(), N, pre.data_ptr() if act else None, N, act, x.data_ptr(), K, dw.data_ptr(), db.data_ptr(),
                         wsw.data_ptr(), nbw, B, H, W, K, N, prec, st)

        t_f, t_d, t_w = timed(fwd), timed(dgrad), timed(wgrad)
        mb = 4.0 * M * (K + N) / 1e6
        print(f"{H:4d}x{W:<4d} {K:4d} -> {N:<4d} {mb:6.1f} MB   fwd {t_f:6.1f}   dgrad {t_d:6.1f}   wgrad + fold {t_w:6.1f}", flush=True)
        tot[0] += t_f
        tot[1] += t_d
        tot[2] += t_w
    print(f"sum: fwd {tot[0]:.1f} us, dgrad {tot[1]:.1f} us, wgrad + fold {tot[2]:.1f} us")
