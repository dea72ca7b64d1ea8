#!/usr/bin/env python3
"""Measured error of the bf16-MFMA mode (bench.py --dtype bf16) on the whole model: rel-L2 of the outputs against the reference's fp32
fixture samples and against this build's fp32 path, loss and total gradient norm — the numbers behind the tolerances asserted in
tests/test_model_gpu.py::test_visionmamba_bf16_mfma_vs_reference.   python tools/bf16_error.py > profiles/rNN_bf16_parity.txt"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "adnm-unet_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import torch
from adnm_hip import ops, recipe
from models.ADNMUNet import create_ADNMUNet
from models.loss import enRainfallLoss
from util import load_npz

DEV = "cuda"
rel = lambda a, b: float((a.double() - b.double()).norm() / b.double().norm())
for name, size, batch, radar in [("visionmamba_128_b4", 128, 4, "bench"), ("visionmamba_64_b2", 64, 2, "radar64"), ("visionmamba_128_b1", 128, 1, "radar128")]:
    try:
        z = load_npz(name)
    except Exception as e:
        print(f"{name}: fixture not loadable here ({e})")
        continue
    model = create_ADNMUNet(5, 20, 6, img_size=size)
    recipe.fill_parameters(model)
    model = model.to(DEV).train()
    frames = recipe.radar_batch(batch, 25, size, name=radar).to(DEV)
    x, tgt = frames[:, :5], frames[:, 5:]
    res = {}
    for prec in ("f32", "bf16"):
        ops.set_mfma_precision(prec)
        for p in model.parameters():
            p.grad = None
        out = model(x)
        loss = enRainfallLoss(0.57, 0.25, gamma=0.0)(out, tgt)
        loss.backward()
        total = sum(float(p.grad.double().pow(2).sum()) for p in model.parameters() if p.grad is not None) ** 0.5
        res[prec] = (out.detach(), float(loss), total)
    ops.set_mfma_precision("f32")
    idx = z["out_idx"].to(DEV)
    ref = z["out_samples"].to(DEV)
    print(f"{name} (B={batch}, {size}x{size}):")
    for prec in ("f32", "bf16"):
        o, l, t = res[prec]
        print(f"  {prec:4s}: outputs vs reference fixture rel-L2 {rel(o.flatten()[idx], ref):.2e}; loss {l:.6f} (reference {float(z['loss']):.6f}, "
              f"rel {abs(l - float(z['loss'])) / abs(float(z['loss'])):.1e}); total gradient norm rel {abs(t - float(z['grad_total_norm'])) / float(z['grad_total_norm']):.1e}")
    print(f"  bf16 vs this build's fp32 path: rel-L2 {rel(res['bf16'][0], res['f32'][0]):.2e}, max abs {float((res['bf16'][0] - res['f32'][0]).abs().max()):.2e}")
