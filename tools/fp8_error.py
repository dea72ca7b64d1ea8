#!/usr/bin/env python3
"""fp8 (BASELINE config 5) accuracy of the whole model under exemption policies: rel-L2 of the fp8 forward against this build's fp32
forward on the benchmarked batch (parameters = the recipe, calibration on the same batch).  Run on the GPU box:
    python tools/fp8_error.py            -> one line per policy (which module prefixes keep bf16 operands)"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "adnm-unet_amd"))
from adnm_hip import ops, recipe  # noqa: E402
from models.ADNMUNet import create_ADNMUNet  # noqa: E402

POLICIES = {
    "default rule: fp8 for GEMMs over <= QUANT.max_rows token rows": None,
    "all fp8": [],
    "out_proj bf16": ["refiner.out_proj"],
    "out_proj + encoder1 bf16": ["refiner.out_proj", "encoder.encoder1"],
    "out_proj + encoder1 + decoder6_s bf16": ["refiner.out_proj", "encoder.encoder1", "decoder.decoder6_s"],
    "refiner + encoder1 bf16": ["refiner.", "encoder.encoder1"],
    "refiner + decoder6 + encoder1/2 bf16 (the whole 128x128 level)": ["refiner.", "encoder.encoder1", "encoder.encoder2", "decoder.decoder6", "decoder.decoder5"],
    "deep levels only fp8 (encoder4-6, attn2, decoder1-3, e2ds, fusion)": ["refiner.", "encoder.encoder1", "encoder.encoder2", "encoder.encoder3", "encoder.attn.",
                                                                          "decoder.decoder4", "decoder.decoder5", "decoder.decoder6", "decoder.attn", "decoder.up_sample"],
}


def main():
    dev = torch.device("cuda")
    model = create_ADNMUNet(5, 20, 6, img_size=128)
    recipe.fill_parameters(model)
    model = model.to(dev).train()
    x = recipe.radar_batch(4, 25, 128, name="bench").to(dev)[:, :5].contiguous()
    with torch.no_grad():
        y32 = model(x)
        ops.set_mfma_precision("bf16")
        y16 = model(x)
        print(f"{'bf16':70s} rel-L2 {float((y16 - y32).norm() / y32.norm()):.4e}")
        default_rows = ops.QUANT.max_rows
        for name, prefixes in POLICIES.items():
            ops.QUANT.max_rows = default_rows if prefixes is None else 1 << 30
            keys = [p.data_ptr() for n, p in model.named_parameters() if any(n.startswith(pre) for pre in (prefixes or []))]
            ops.QUANT.keep_bf16(keys)
            ops.fp8_calibrate(dev, lambda: model(x))
            y8 = model(x)
            ops.set_mfma_precision("f32")
            nsites = len(ops.QUANT.dump(dev))
            print(f"{name:70s} rel-L2 {float((y8 - y32).norm() / y32.norm()):.4e}   ({nsites} fp8 call sites)")
    ops.QUANT.max_rows = default_rows
    ops.QUANT.keep_bf16([])
    ops.QUANT.reset()
    del model
    train_margin(dev)


def train_margin(dev):
    """the measured margin of tests/test_model_gpu.py::test_visionmamba_fp8_vs_reference's training bar (loss of 50 FlatTrainer steps within
    5 % of the fp32 HIP path's at every step): the default rule, and the variant with every call site on fp8"""
    from adnm_hip.trainer import FlatTrainer
    from models.loss import enRainfallLoss
    frames = recipe.radar_batch(4, 25, 128, name="bench").to(dev)
    x, tgt = frames[:, :5].contiguous(), frames[:, 5:].contiguous()
    crit = enRainfallLoss(0.57, 0.25, gamma=0.0)

    def run(prec, max_rows=None):
        default_rows = ops.QUANT.max_rows
        if max_rows is not None:
            ops.QUANT.max_rows = max_rows
        ops.set_mfma_precision(prec)
        try:
            m = create_ADNMUNet(5, 20, 6, img_size=128)
            recipe.fill_parameters(m)
            tr = FlatTrainer(m.to(dev).train(), crit, lr=1e-3, betas=(0.9, 0.999), eps=1e-9, weight_decay=1e-2, max_norm=0.025, use_graph=True)
            losses = [float(tr.step(x, tgt)) for _ in range(50)]
            nsites = len(ops.QUANT.dump(dev)) if prec == "fp8" else 0
            tr.close()
            return losses, nsites
        finally:
            ops.set_mfma_precision("f32")
            ops.QUANT.max_rows = default_rows
            ops.QUANT.reset()

    l32, _ = run("f32")
    for label, rows in (("default rule (GEMMs over > QUANT.max_rows token rows keep bf16 operands)", None), ("every GEMM / conv call site on fp8", 1 << 30)):
        l8, n = run("fp8", rows)
        worst = max(abs(a - b) / abs(a) for a, b in zip(l32, l8))
        print(f"50 training steps, {label}: loss {l8[0]:.5f} -> {l8[-1]:.5f} (fp32 {l32[0]:.5f} -> {l32[-1]:.5f}), worst relative deviation "
              f"{worst:.3e} (bar 5e-2), {n} quantisation records (call sites + weight records)")


if __name__ == "__main__":
    main()
