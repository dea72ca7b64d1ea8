#!/usr/bin/env python3
"""Depthwise 3x3 stencil at the full-resolution level, fp32 vs bf16 token storage: device time per launch of a replayed graph of REPS launches
(warm: the same buffers every launch, as inside a training step where the producer has just written them) next to a plain copy of the same
bytes.  Run on the GPU box: python tools/kbench_dw.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "adnm-unet_amd"))
import torch
from adnm_hip import ops, lib

dev = "cuda"
reps = 20


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
    best = 1e9
    for _ in range(5):
        torch.cuda.synchronize()
        e0.record()
        g.replay()
        e1.record()
        e1.synchronize()
        best = min(best, 1e3 * e0.elapsed_time(e1) / reps)
    return best


B, H, W = 4, 128, 128
M = B * H * W
for C, ld_in, ld_out, act, wcm in [(192, 208, 320, lib.ACT_SILU, False), (192, 208, 320, lib.ACT_NONE, False), (128, 128, 128, lib.ACT_NONE, True)]:
    for dt in (torch.float32, torch.bfloat16):
        src = torch.randn(M, ld_in, device=dev).to(dt)
        dst = torch.empty(M, ld_out, device=dev, dtype=dt)
        taps = torch.randn(C, 9, device=dev) if wcm else torch.randn(9, C, device=dev)
        x, y = src[:, :C], dst[:, ld_out - C:]
        es = src.element_size()
        t_f = graph_time(lambda: ops.k_dwconv_fwd(x, taps, None, B, H, W, C, 3, act, y=y, chan_major=wcm))
        dy = torch.randn(M, C, device=dev).to(dt)
        dx = torch.empty(M, C, device=dev, dtype=dt)
        dpre = torch.empty(M, C, device=dev, dtype=dt)
        nb = lib.query("adnm_dwconv_bwd_ws_bytes", B, H, W, C, 3, 3)
        ws = torch.empty(nb, dtype=torch.uint8, device=dev)
        dwt = torch.empty_like(taps)

        def bwd():
            lib.call("adnm_dwconv_bwd", dy.data_ptr(), C, x.data_ptr(), x.stride(0), taps.data_ptr(), None, dpre.data_ptr() if act else None, dx.data_ptr(), C, None, None,
                     ws.data_ptr(), nb, B, H, W, C, 3, 3, act, int(wcm), ops._dt(x), torch.cuda.current_stream().cuda_stream)

        def wgrad():
            lib.call("adnm_dwconv_wgrad", dy.data_ptr(), C, x.data_ptr(), x.stride(0), dwt.data_ptr(), None, ws.data_ptr(), nb, B, H, W, C, 3, 3, int(wcm), ops._dt(x),
                     torch.cuda.current_stream().cuda_stream)
        t_b, t_w = graph_time(bwd), graph_time(wgrad)
        a, b2 = torch.randn(M, C, device=dev).to(dt), torch.empty(M, C, device=dev, dtype=dt)
        t_c = graph_time(lambda: b2.copy_(a))
        mb = 2 * M * C * es / 1e6
        print(f"C={C:4d} act={act} wcm={int(wcm)} {str(dt)[6:]:9s} fwd {t_f:6.1f} us ({mb / t_f / 1e3 * 1e3:6.0f} GB/s) | bwd(dpre+dx) {t_b:6.1f} us | wgrad {t_w:6.1f} us | "
              f"copy {t_c:6.1f} us ({mb / t_c / 1e3 * 1e3:6.0f} GB/s)", flush=True)
