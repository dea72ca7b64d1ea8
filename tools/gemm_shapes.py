#!/usr/bin/env python3
"""Every Linear-shaped GEMM of one training step (config 2 by default): which kernel takes it, and the time of libadnm_hip's
short-GEMM kernel against the library GEMM for the same operands.   python tools/gemm_shapes.py [size] [batch]
Prints one row per (op, M, N, K): calls per step, skgemm us, rocBLAS us."""
import os, sys, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "adnm-unet_amd"))
import torch
from adnm_hip import ops, recipe, lib
from models.ADNMUNet import create_ADNMUNet
from models.loss import enRainfallLoss

size = int(sys.argv[1]) if len(sys.argv) > 1 else 128
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 4
dev = "cuda"
torch.backends.cuda.preferred_blas_library("hipblas")
shapes = collections.Counter()
orig = (ops.k_linear, ops.k_linear_dx, ops.k_linear_dw)


def rec_fwd(x2, w, bias, out=None):
    shapes[("NT", x2.shape[0], w.shape[0], x2.shape[1], bias is not None)] += 1
    return orig[0](x2, w, bias, out)


def rec_dx(dy2, w, out=None):
    shapes[("NN", dy2.shape[0], dy2.shape[1], w.shape[1], False)] += 1
    return orig[1](dy2, w, out)


def rec_dw(dy2, x2, want_bias, *a, **k):
    shapes[("TN", dy2.shape[0], dy2.shape[1], x2.shape[1], bool(want_bias))] += 1
    return orig[2](dy2, x2, want_bias, *a, **k)


ops.k_linear, ops.k_linear_dx, ops.k_linear_dw = rec_fwd, rec_dx, rec_dw
m = create_ADNMUNet(5, 20, 6, img_size=size).to(dev)
recipe.fill_parameters(m)
fr = recipe.radar_batch(batch, 25, size, name="bench").to(dev)
enRainfallLoss(0.57, 0.25, 0.)(m(fr[:, :5]), fr[:, 5:]).backward()
ops.k_linear, ops.k_linear_dx, ops.k_linear_dw = orig


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    e1.synchronize()
    return 1e3 * e0.elapsed_time(e1) / n


OPC = {"NT": ops.SK_NT, "NN": ops.SK_NN, "TN": ops.SK_TN}
tot = {"ts": 0.0, "sk": 0.0, "lib": 0.0, "best": 0.0}
print(f"{'op':3s} {'M':>6s} {'N':>5s} {'K':>5s} b  calls  {'ts us':>8s} {'sk us':>8s} {'lib us':>8s}")
for (op, M, N, K, hb), cnt in sorted(shapes.items(), key=lambda kv: (kv[0][1], kv[0][0], kv[0][2])):
    g = lambda *s: torch.randn(*s, device=dev)
    t_ts = t_sk = None
    if op == "NT":
        a, w, b = g(M, K), g(N, K), (g(N) if hb else None)
        libf = (lambda: torch.addmm(b, a, w.t())) if hb else (lambda: torch.mm(a, w.t()))
        c = torch.empty(M, N, device=dev)
        if ops.ts_ok_nt(M, N, K, a):
            t_ts = timeit(lambda: lib.call("adnm_tsgemm_nt", a.data_ptr(), K, w.data_ptr(), K, 1, ops._p(b), c.data_ptr(), N, M, N, K, ops.MFMA_PREC[0], ops._stream()))
        if lib.query("adnm_skgemm_supported", OPC[op], M, N, K) == 1:
            t_sk = timeit(lambda: ops._skgemm(OPC[op], a, w, b, c, None, M, N, K))
    elif op == "NN":
        a, w = g(M, N), g(N, K)
        libf = lambda: torch.mm(a, w)
        c = torch.empty(M, K, device=dev)
        if ops.ts_ok_nt(M, K, N, a):
            t_ts = timeit(lambda: lib.call("adnm_tsgemm_nt", a.data_ptr(), N, w.data_ptr(), 1, K, None, c.data_ptr(), K, M, K, N, ops.MFMA_PREC[0], ops._stream()))
        if lib.query("adnm_skgemm_supported", OPC[op], M, N, K) == 1:
            t_sk = timeit(lambda: ops._skgemm(OPC[op], a, w, None, c, None, M, N, K))
    else:
        a, x = g(M, N), g(M, K)
        c, db = torch.empty(N, K, device=dev), (torch.empty(N, device=dev) if hb else None)
        libf = (lambda: (torch.mm(a.t(), x), a.sum(0))) if hb else (lambda: torch.mm(a.t(), x))
        if ops.ts_ok_tn(M, N, K, x):
            nb = lib.query("adnm_tsgemm_tn_ws_bytes", M, N, K)
            ws = torch.empty(nb, dtype=torch.uint8, device=dev)
            t_ts = timeit(lambda: lib.call("adnm_tsgemm_tn", a.data_ptr(), N, x.data_ptr(), K, c.data_ptr(), ops._p(db), ws.data_ptr(), nb, M, N, K, ops._stream()))
        if lib.query("adnm_skgemm_supported", OPC[op], M, N, K) == 1:
            t_sk = timeit(lambda: ops._skgemm(OPC[op], a, x, None, c, db, M, N, K))
    t_lib = timeit(libf)
    f = lambda t: f"{t:8.1f}" if t is not None else "       -"
    print(f"{op:3s} {M:6d} {N:5d} {K:5d} {int(hb)} {cnt:5d}  {f(t_ts)} {f(t_sk)} {f(t_lib)}")
    own = min(t for t in (t_ts, t_sk) if t is not None) if (t_ts is not None or t_sk is not None) else None
    tot["lib"] += cnt * t_lib
    tot["sk"] += cnt * (own if own is not None else t_lib)
    tot["best"] += cnt * min(t for t in (t_ts, t_sk, t_lib) if t is not None)
print(f"per step: own kernels everywhere {tot['sk'] / 1e3:.3f} ms, library everywhere {tot['lib'] / 1e3:.3f} ms, best-of {tot['best'] / 1e3:.3f} ms")
