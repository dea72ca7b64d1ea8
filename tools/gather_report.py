#!/usr/bin/env python3
"""tools/gather_report.py: which parameter gradients of the benchmarked step are NOT born inside the trainer's flat gradient buffer
(FlatTrainer._gather copies those with torch._foreach_copy_ once per stage).  Prints name, numel and the stage."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "adnm-unet_amd"))
import torch
from adnm_hip import lib, recipe, ops
from adnm_hip.trainer import FlatTrainer
from models.ADNMUNet import create_ADNMUNet
from models.loss import enRainfallLoss
lib.load()
ops.set_mfma_precision("bf16")
os.environ["ADNM_AUTO_DDP"] = "0"
dev = torch.device("cuda", 0)
model = create_ADNMUNet(5, 20, 6, img_size=128)
recipe.fill_parameters(model)
model = model.to(dev).train()
crit = enRainfallLoss(omega_t=0.57, alpha=0.25, gamma=0.).to(dev)
tr = FlatTrainer(model, crit, lr=1e-3, betas=(0.9, 0.999), eps=1e-9, weight_decay=1e-2, max_norm=0.025, use_graph=True)
frames = recipe.radar_batch(4, 25, 128, salt=0, name="bench").to(dev)
x, t = frames[:, :5].contiguous(), frames[:, 5:].contiguous()
tr.prepare(x, t)
tr.step(x, t)
names = {id(p): n for n, p in model.named_parameters()}
tot = 0
for i in tr.gathered:
    p = tr.used[i]
    print(f"{names[id(p)]:70s} {p.numel():8d} {tuple(p.shape)}")
    tot += p.numel()
print("gathered by copy:", len(tr.gathered), "tensors,", tot, "elements of", sum(p.numel() for p in tr.used))
