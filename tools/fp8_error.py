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


if __name__ == "__main__":
    main()
