#!/usr/bin/env python3
"""Which short-GEMM launches of one eager training step are slow FOR THEIR SIZE, and what their operands look like: every adnm_skgemm call
of a step timed with events on its stream, grouped by (op, M, N, K, lda, ldc, contiguous output, weight storage), next to the same shape
launched alone on dense operands.   python tools/gemm_probe.py [min_us]"""
import os, sys, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "adnm-unet_amd"))
import torch
from adnm_hip import ops, recipe
from adnm_hip.trainer import FlatTrainer
from models.ADNMUNet import create_ADNMUNet
from models.loss import enRainfallLoss

dev = "cuda"
min_us = float(sys.argv[1]) if len(sys.argv) > 1 else 0.0
ops.set_mfma_precision("bf16")
m = create_ADNMUNet(5, 20, 6, img_size=128).to(dev)
recipe.fill_parameters(m)
fr = recipe.radar_batch(4, 25, 128, name="bench").to(dev)
x, t = fr[:, :5].contiguous(), fr[:, 5:].contiguous()
tr = FlatTrainer(m, enRainfallLoss(0.57, 0.25, 0.), lr=1e-4, max_norm=1.0, use_graph=False)
for _ in range(3):
    tr.step(x, t, eager=True)
torch.cuda.synchronize()

orig = ops._skgemm
log = []


def probe(op, a, b, bias, c, dbias, M, N, K, defer=False, side=False, q=None, role="f", narrow=None):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    orig(op, a, b, bias, c, dbias, M, N, K, defer=defer, side=side, q=q, role=role, narrow=narrow)
    e1.record()
    log.append(((("NT", "NN", "TN")[op], M, N, K, a.stride(0), c.stride(0), bool(c.is_contiguous()), type(b).__name__, bias is not None), e0, e1))


ops._skgemm = probe
for _ in range(3):
    log.clear()
    tr.step(x, t, eager=True)
torch.cuda.synchronize()
ops._skgemm = orig
agg = collections.defaultdict(list)
for key, e0, e1 in log:
    agg[key].append(e0.elapsed_time(e1) * 1e3)
print("op      M     N     K    lda    ldc  c-dense  weight        bias  calls   avg us")
for key, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
    if key[0] == "TN" or sum(v) / len(v) < min_us:
        continue
    print("%s %6d %5d %5d %6d %6d  %-7s  %-12s %-5s %4d   %6.1f" % (*key, len(v), sum(v) / len(v)))
