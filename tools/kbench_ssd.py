#!/usr/bin/env python3
"""K1 micro-benchmark at the refiner shape (or tools/kbench_ssd.py B L H): back-to-back launches of the forward pair (kv+apply[+LN]) and
the backward pair (dkv+bwd) on the mixer's real operand layout (column slices of the (M,128) xBC and (M,208) proj buffers).
Run it under rocprofv3 --kernel-trace --stats (durations) or --pmc (counters)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "adnm-unet_amd"))
import torch
from adnm_hip import ops, lib

B, L, H = (int(a) for a in sys.argv[1:4]) if len(sys.argv) > 3 else (4, 16384, 16)
P, N, G = 4, 16, 2
reps = int(os.environ.get("REPS", "20"))
dev = "cuda"
M, di = B * L, H * P
torch.manual_seed(0)
xbc = torch.randn(M, di + 2 * G * N, device=dev)
proj = torch.randn(M, 2 * di + 2 * G * N + H, device=dev)
dxbc, dproj = torch.empty_like(xbc), torch.empty_like(proj)
cat = torch.empty(M, 2 * di, device=dev)
dy = torch.randn(M, di, device=dev)
dt_bias, A_log, D = torch.randn(H, device=dev) * 0.1, torch.rand(H, device=dev), torch.ones(H, device=dev)
lw, lb = torch.ones(di, device=dev), torch.zeros(di, device=dev)
x, Bm, Cm, dt = xbc[:, :di], xbc[:, di:di + G * N], xbc[:, di + G * N:], proj[:, -H:]


def fwd():
    if di == 64:
        return ops.k_ssd_fwd(x, Bm, Cm, dt, dt_bias, A_log, D, B, L, H, P, N, G, ln=(lw, lb, cat[:, :di], 1e-5))
    return ops.k_ssd_fwd(x, Bm, Cm, dt, dt_bias, A_log, D, B, L, H, P, N, G)


y, kv = fwd()[:2]


def bwd():
    return ops.k_ssd_bwd(dy, x, Bm, Cm, dt, dt_bias, A_log, D, kv, dxbc[:, :di], dxbc[:, di:di + G * N], dxbc[:, di + G * N:], dproj[:, -H:], B, L, H, P, N, G)


for name, fn in (("fwd", fwd), ("bwd", bwd)):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    e1.synchronize()
    print(f"{name}: {1e3 * e0.elapsed_time(e1) / reps:.1f} us per call (B={B} L={L} H={H})")
