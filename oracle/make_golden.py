"""Generates tests/golden/*.npz by running the REFERENCE's own Python (through
oracle/ref_harness.py) on recipe-filled parameters and recipe inputs.

CONTAINER-ONLY (needs /root/reference).  Fixtures are data: inputs, parameters of the
small modules, expected outputs / gradients.  Run:  python oracle/make_golden.py
"""
import os
import sys
import json
import re
import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(ROOT, "adnm-unet_amd"))

import ref_harness as H  # noqa: E402
from adnm_hip import recipe  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
os.makedirs(OUT, exist_ok=True)
torch.manual_seed(0)


def npy(t):
    return t.detach().cpu().numpy()


ONLY = None  # regex from --only: fixtures whose name does not match are left as they are on disk


def wanted(name):
    return ONLY is None or re.search(ONLY, name) is not None


def save(name, **arrays):
    if not wanted(name):
        return
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **{k: (npy(v) if torch.is_tensor(v) else np.asarray(v)) for k, v in arrays.items()})
    print(f"{name}: {os.path.getsize(path) / 1024:.1f} KiB")


def module_case(name, mod, inputs, call=None, grad_inputs=()):
    """Fill `mod` by recipe, run fwd + bwd against a recipe cotangent, save params, inputs,
    output, input grads and parameter grads."""
    if not wanted(name):
        return
    recipe.fill_parameters(mod)
    ins = {k: v.clone().requires_grad_(k in grad_inputs) for k, v in inputs.items()}
    out = call(mod, **ins) if call else mod(**ins)
    outs = out if isinstance(out, (tuple, list)) else (out,)
    arrays = {}
    loss = 0
    for i, o in enumerate(outs):
        cot = recipe.tensor(f"{name}.cot{i}", tuple(o.shape))
        arrays[f"out{i}"] = o
        arrays[f"cot{i}"] = cot
        loss = loss + (o * cot).sum()
    loss.backward()
    for k, v in ins.items():
        arrays["in." + k] = v
        if k in grad_inputs:
            arrays["gin." + k] = v.grad
    for k, v in mod.state_dict().items():
        arrays["p." + k] = v
    for k, p in mod.named_parameters():
        if p.grad is not None:
            arrays["g." + k] = p.grad
    save(name, **arrays)


def main():
    ref = H.load_reference()
    A, S, V, Wt, U, Ls = ref.ADNMUNet, ref.ADNssd, ref.Vssd, ref.WTConv2d, ref.model_untils, ref.loss
    T = recipe.tensor

    # G2 WTConv2d (WTConv2d.py:63-153): even and odd (padded/cropped) sizes
    module_case("wtconv_c5_l3_16x16", Wt.WTConv2d(5, 5, 5, 1, False, wt_levels=3),
                {"x": T("wt1.x", (2, 5, 16, 16))}, grad_inputs=("x",))
    module_case("wtconv_c8_l2_20x28", Wt.WTConv2d(8, 8, 5, 1, True, wt_levels=2),
                {"x": T("wt2.x", (2, 8, 20, 28))}, grad_inputs=("x",))
    module_case("wtconv_c4_l3_11x13_k3", Wt.WTConv2d(4, 4, 3, 1, True, wt_levels=3),
                {"x": T("wt3.x", (1, 4, 11, 13))}, grad_inputs=("x",))

    # G5 K1 alone (ADNssd.py:252-299 single-group branch; Vssd.py:161-208 grouped branch)
    m = S.Mamba2(d_model=32, headdim=4)
    x, dt = T("k1.x", (2, 96, 8, 4)), T("k1.dt", (2, 96, 8), 0.1, positive=True)
    Aneg = -torch.exp(T("k1.alog", (8,), 1.5))
    Bm, Cm, D = T("k1.B", (2, 96, 16)), T("k1.C", (2, 96, 16)), 1 + 0.1 * T("k1.D", (8,))
    y = m.non_casual_linear_attn(x, dt, Aneg, Bm, Cm, D, 8, 12)
    save("k1_single_group", x=x, dt=dt, A=Aneg, B=Bm, C=Cm, D=D, y=y)
    mv = V.Mamba2(d_model=32, headdim=4)
    Bg, Cg = T("k1g.B", (2, 96, 32)), T("k1g.C", (2, 96, 32))
    yg = mv.non_casual_linear_attn(x, dt, Aneg, Bg, Cg, D, 8, 12)
    save("k1_grouped", x=x, dt=dt, A=Aneg, B=Bg, C=Cg, D=D, y=yg)

    # G3 / G4 mixers
    module_case("adn_mamba2_d32", S.Mamba2(d_model=32, headdim=4), {"u": T("adn.u", (2, 144, 32))},
                call=lambda mod, u: mod(u, 12, 12), grad_inputs=("u",))
    module_case("adn_mamba2_d64_rect", S.Mamba2(d_model=64, headdim=4), {"u": T("adn2.u", (1, 60, 64))},
                call=lambda mod, u: mod(u, 6, 10), grad_inputs=("u",))
    module_case("vssd_mamba2_d32", V.Mamba2(d_model=32, headdim=4), {"u": T("vssd.u", (2, 144, 32))},
                call=lambda mod, u: mod(u, 12, 12), grad_inputs=("u",))

    # G6 Block (ADNMUNet.py:51-168) through create_block (:243-292), eps 1e-6 as create_ADNMUNet
    module_case("block_32_32", A.create_block(32, 32, headdim=4, norm_epsilon=1e-6), {"x": T("blk.x", (2, 64, 32))},
                call=lambda mod, x: mod(x), grad_inputs=("x",))
    module_case("block_res_feat_64_32", A.create_block(64, 32, headdim=4, norm_epsilon=1e-6),
                {"x": T("blk2.x", (1, 64, 32)), "r": T("blk2.r", (1, 64, 32)), "f": T("blk2.f", (1, 64, 32))},
                call=lambda mod, x, r, f: mod(x, residual=r, features=f), grad_inputs=("x", "r", "f"))
    module_case("block_feat_32_16", A.create_block(32, 16, headdim=4, norm_epsilon=1e-6),
                {"x": T("blk3.x", (1, 16, 32)), "f": T("blk3.f", (1, 16, 32))},
                call=lambda mod, x, f: mod(x, features=f), grad_inputs=("x", "f"))

    # G7 conv-stack / glue modules (model_untils.py)
    module_case("attention_d32", A.Attention(32, headdim=4), {"x": T("att.x", (2, 16, 32))},
                call=lambda mod, x: mod(x), grad_inputs=("x",))
    module_case("patch_embed_5_16", U.PatchEmbed(img_size=16, patch_size=2, in_channels=5, embed_dim=16, kernel=5, wt_levels=3),
                {"x": T("pe.x", (2, 256, 5), positive=True)}, call=lambda mod, x: mod(x), grad_inputs=("x",))
    module_case("wtlayer_16_24", U.WTLayer(16, 24, kernel=5, wt_levels=2), {"x": T("wl.x", (2, 144, 16))},
                call=lambda mod, x: mod(x), grad_inputs=("x",))
    module_case("wtlayer_res_16_8", U.WTLayer(16, 8, kernel=5, wt_levels=1, if_res=True),
                {"x": T("wl2.x", (1, 64, 8)), "r": T("wl2.r", (1, 64, 8)), "f": T("wl2.f", (1, 64, 8))},
                call=lambda mod, x, r, f: mod(x, residual=r, features=f), grad_inputs=("x", "r"))
    module_case("outproj_16_6", U.OutProj(num_frames=6, embed_dim=16, img_size=[16, 16], wt_levels=3, out_expand=2),
                {"x": T("op.x", (2, 256, 16)), "res": T("op.r", (2, 16, 16), positive=True)},
                call=lambda mod, x, res: mod(x, res), grad_inputs=("x",))
    module_case("upsample_8", U.UpSample(dim=8, ratio=2), {"x": T("up.x", (2, 16, 8))},
                call=lambda mod, x: mod(x), grad_inputs=("x",))
    module_case("downsample_8", U.DownSample(dim=8, ratio=2), {"x": T("dn.x", (2, 64, 8))},
                call=lambda mod, x: mod(x), grad_inputs=("x",))
    module_case("e2d_16", U.EncoderToDecoder(embed_dim=16), {"x": T("e2d.x", (2, 64, 16)), "res": T("e2d.r", (2, 64, 16))},
                call=lambda mod, x, res: mod(x, res), grad_inputs=("x", "res"))
    dims = [4, 8, 8, 8, 12, 16, 20]
    sizes = [64, 16, 16, 4, 4, 4, 4]

    def bridge(mod, **kw):
        d = {i: kw[f"t{i}"] for i in range(7)}
        out = mod(d)
        return tuple(out[i] for i in range(7))

    module_case("bridge_small", U.Channel_Att_Bridge(c_list=dims),
                {f"t{i}": T(f"br.t{i}", (2, sizes[i], dims[i])) for i in range(7)}, call=bridge,
                grad_inputs=tuple(f"t{i}" for i in range(7)))

    # G8 loss (loss.py:30-57): targets cross the 0.7 heavy-rain threshold; gamma 0 and 0.1
    pred, tgt = T("loss.p", (2, 20, 1, 16, 16), positive=True), T("loss.t", (2, 20, 1, 16, 16), positive=True)
    pr = pred.clone().requires_grad_(True)
    l0 = Ls.enRainfallLoss(0.57, 0.25, gamma=0.0)(pr, tgt)
    l0.backward()
    l1 = Ls.enRainfallLoss(0.57, 0.25, gamma=0.1)(pred, tgt)
    save("en_rainfall_loss", pred=pred, target=tgt, loss_g0=l0, grad_g0=pr.grad, loss_g01=l1)

    # G9 whole model, create_ADNMUNet(5,20,6) configuration (+ config 4's 10->40 at 256x256, + the benchmarked B=4 batch)
    if wanted("state_dict_manifest"):
        model = H.build_visionmamba(64)
        consts = {}
        for k, v in model.state_dict().items():   # per-tensor init constant (None for randomly initialised tensors): lets the
            f = v.double().flatten()              # recipe rebuild the exact state_dict from this manifest alone
            consts[k] = float(f[0]) if bool((f == f[0]).all()) else None
        recipe.fill_parameters(model)
        trainable = {k: p.requires_grad for k, p in model.named_parameters()}
        manifest = {k: {"shape": list(v.shape), "const": consts[k], "trainable": bool(trainable[k]),
                        "sum": float(v.double().sum()), "abs": float(v.double().abs().sum())}
                    for k, v in model.state_dict().items()}
        with open(os.path.join(OUT, "state_dict_manifest.json"), "w") as f:
            json.dump(manifest, f, separators=(",", ":"))
        del model
    whole_model_case("visionmamba_64_b2", 64, 2, radar="radar64", full_out=True)
    whole_model_case("visionmamba_128_b1", 128, 1, radar="radar128")
    whole_model_case("visionmamba_256_b1", 256, 1, radar="radar256")
    # round 2: the exact benchmarked workload (bench.py: B=4, radar_batch(name="bench", salt=rank 0)), BASELINE config 2's shape
    whole_model_case("visionmamba_128_b4", 128, 4, radar="bench", deltas=True)
    # round 2: BASELINE config 4 (10 -> 40 frames, 256x256): channels=10, out_channels=40 (ADNMUNet.py:906-940)
    whole_model_case("visionmamba_256_10to40_b1", 256, 1, radar="radar256x", cin=10, cout=40, deltas=True)
    convlstm_case()
    evaluator_case()
    resize_case()


def whole_model_case(name, size, batch, radar, cin=5, cout=20, full_out=False, deltas=False):
    if not wanted(name):
        return
    ref = H.load_reference()
    Ls = ref.loss
    model = H.build_visionmamba(size, channels=cin, out_channels=cout)
    recipe.fill_parameters(model)
    frames = recipe.radar_batch(batch, cin + cout, size, name=radar)
    x, tgt = frames[:, :cin], frames[:, cin:]
    taps = {}
    hooks = [model.encoder.register_forward_hook(lambda m, i, o: taps.__setitem__("encoder", o[0])),
             model.decoder.register_forward_hook(lambda m, i, o: taps.__setitem__("decoder", o)),
             model.refiner.refiner4.register_forward_hook(lambda m, i, o: taps.__setitem__("refiner4", o))]
    out = model(x)
    loss = Ls.enRainfallLoss(0.57, 0.25, gamma=0.0)(out, tgt)
    loss.backward()
    for h in hooks:
        h.remove()
    names = [k for k, _ in model.named_parameters()]
    gnorm = np.array([float(p.grad.double().norm()) if p.grad is not None else -1.0 for _, p in model.named_parameters()])
    total = torch.sqrt(sum((p.grad.double() ** 2).sum() for p in model.parameters() if p.grad is not None))
    flat = out.flatten()
    idx = torch.from_numpy((recipe.uniform01(f"sample{name if deltas else size}", 4096) * flat.numel()).astype(np.int64))
    arrays = dict(out_idx=idx, out_samples=flat[idx], out_norm=out.double().norm(), out_mean=out.double().mean(),
                  loss=loss, grad_total_norm=total, grad_norms=gnorm, names=np.array(names))
    for k, v in taps.items():
        fl = v.flatten()
        ii = torch.from_numpy((recipe.uniform01(f"tap{k}{name if deltas else size}", 1024) * fl.numel()).astype(np.int64))
        arrays[f"tap.{k}.idx"], arrays[f"tap.{k}.val"], arrays[f"tap.{k}.norm"] = ii, fl[ii], v.double().norm()
    if full_out:
        arrays["out_full"] = out
    if deltas:
        # gradient DIRECTIONS, not only norms: the projection of every parameter gradient on a recipe probe vector
        arrays["grad_probe"] = np.array([float((p.grad.double().flatten() * torch.from_numpy(recipe.sym("probe." + k, p.numel()))).sum())
                                         if p.grad is not None else 0.0 for k, p in model.named_parameters()])
        before = [p.detach().double().clone() for p in model.parameters()]
    # one AdamW step with train.py's recipe (train_untils.py:35-42, train.py:140 clip at norm_max 0.025)
    opt = torch.optim.AdamW(model.parameters(), lr=1e-3, betas=(0.9, 0.999), eps=1e-9, weight_decay=1e-2)
    pre = torch.nn.utils.clip_grad_norm_(model.parameters(), 0.025)
    opt.step()
    arrays["clip_pre_norm"] = pre
    arrays["param_sum_after_step"] = np.array([float(p.double().sum()) for _, p in model.named_parameters()])
    if deltas:
        # per-parameter update (p_after - p_before): its sum, and every element of the <= 8-element tensors (the 360 scalars)
        d = [p.detach().double() - b for p, b in zip(model.parameters(), before)]
        arrays["delta_sum"] = np.array([float(t.sum()) for t in d])
        arrays["delta_abs"] = np.array([float(t.abs().sum()) for t in d])
        small = [t.flatten() for t in d if t.numel() <= 8]
        arrays["delta_small"] = torch.cat(small)
        arrays["delta_small_sizes"] = np.array([t.numel() for t in small])
    save(name, **arrays)
    del model


def evaluator_case():
    """SimplifiedEvaluator (datasets/Shanghai_metrics.py:14-290) on a recipe batch: the contingency counts, MSE / MAE per frame and the
    aggregated CSI / POD / HSS / FAR / RMSE of done().  lpips (a weight download) and cv2 are absent here: lpips.LPIPS is a stub
    returning zeros, so LPIPS is NOT recorded.  SSIM (round 3) IS recorded, from the reference's own cal_ssim code (:132-152) with
    the two cv2 calls it makes replaced by their documented formulas — cv2.getGaussianKernel(11, 1.5) = the normalised sampled Gaussian
    (OpenCV's formula for ksize > 7), cv2.filter2D = correlation (the border mode is irrelevant: cal_ssim crops to the valid region) —
    i.e. pinned to the reference's arithmetic around those two calls, not to OpenCV's binaries."""
    name = "evaluator_b3_t5"
    if not wanted(name):
        return
    import importlib.util
    import types
    import scipy.ndimage
    H.load_reference()
    lp = types.ModuleType("lpips")

    class LPIPS(torch.nn.Module):
        def __init__(self, **kw):
            super().__init__()

        def forward(self, a, b):
            return torch.zeros(a.shape[0], 1, 1, 1)
    lp.LPIPS = LPIPS
    sys.modules["lpips"] = lp
    cv2 = sys.modules["cv2"]

    def gk(ksize, sigma):
        x = np.arange(ksize) - (ksize - 1) / 2.0
        k = np.exp(-(x ** 2) / (2 * sigma ** 2))
        return (k / k.sum()).reshape(-1, 1)
    cv2.getGaussianKernel = gk
    cv2.filter2D = lambda img, ddepth, kern: scipy.ndimage.correlate(img, kern, mode="mirror")
    # the reference's datasets/ has no __init__.py and an unrelated installed package owns that name: load the file itself
    spec = importlib.util.spec_from_file_location("_adnm_ref_shanghai_metrics", os.path.join(H.REF_ROOT, "datasets", "Shanghai_metrics.py"))
    M = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(M)
    B, T, S = 3, 5, 48
    truth = recipe.tensor("eval.t", (B, T, S, S), 0.9, positive=True)
    pred = (truth + 0.25 * recipe.tensor("eval.e", (B, T, S, S))).clamp(-0.1, 1.1)
    ev = M.SimplifiedEvaluator(seq_len=T, value_scale=90, thresholds=[20, 30, 35, 40])
    ev.evaluate(truth.numpy(), pred.numpy())
    res = ev.done()
    arrays = dict(truth=truth, pred=pred, value_scale=np.array(90.0), thresholds=np.array([20, 30, 35, 40]),
                  mse=np.array(ev.losses["mse"]), mae=np.array(ev.losses["mae"]), FAR=np.array(res["FAR"]), RMSE=np.array(res["RMSE"]),
                  ssim=np.array(ev.losses["ssim"]), SSIM=np.array(res["SSIM"]))
    for thr in (20, 30, 35, 40):
        for k in ("hits", "misses", "falsealarms", "correctnegs"):
            arrays[f"{k}.{thr}"] = np.array(ev.metrics[thr][k])
        for k in ("TP", "TN", "FP", "FN", "CSI", "POD", "HSS"):
            arrays[f"{k}.{thr}"] = np.array(float(res["threshold_metrics"][thr][k]))
    save(name, **arrays)


def resize_case():
    """datasets/Shanghai.py:52-59,121: uint8 (T, H0, W0) -> float / 255 -> transforms.Resize((S, S)).  torchvision is not installed here;
    its tensor Resize (v0.16, antialias unset -> False) is torch.nn.functional.interpolate(mode='bilinear', align_corners=False), which
    is what this fixture records — pinned to that torch call, not to torchvision itself."""
    name = "radar_resize_565x784_to_128"
    if not wanted(name):
        return
    rng = (recipe.uniform01("resize.u8", 3 * 57 * 79) * 71).astype(np.uint8).reshape(3, 57, 79)
    small = torch.from_numpy(rng).float() / 255.0
    out_small = torch.nn.functional.interpolate(small[None], size=(16, 16), mode="bilinear", align_corners=False)[0]
    big = (recipe.uniform01("resize.big", 2 * 565 * 784) * 71).astype(np.uint8).reshape(2, 565, 784)
    out_big = torch.nn.functional.interpolate(torch.from_numpy(big).float()[None] / 255.0, size=(128, 128), mode="bilinear", align_corners=False)[0]
    idx = torch.from_numpy((recipe.uniform01("resize.idx", 2048) * out_big.numel()).astype(np.int64))
    save(name, small_u8=rng, small_out=out_small, big_idx=idx, big_samples=out_big.flatten()[idx], big_sum=out_big.double().sum())


def convlstm_case():
    """BASELINE config 1 (plumbing): the reference's ConvLSTM encoder-forecaster, 5 -> 5 frames, batch 2, on the CPU
    (ConvLSTM.py:219-256; it only accepts 256x256 input, its first state grid is the 64x64 of :219)."""
    name = "convlstm_5to5_b2"
    if not wanted(name):
        return
    import importlib
    H.load_reference()
    saved = list(sys.path)
    sys.path.insert(0, H.REF_ROOT)
    try:
        C = importlib.import_module("models.ConvLSTM")
    finally:
        sys.path[:] = saved
        for k in [k for k in sys.modules if k == "models" or k.startswith("models.")]:
            sys.modules["_adnm_ref_" + k] = sys.modules.pop(k)
    ref = H.load_reference()
    model = C.create_ConvLSTM(5)
    recipe.fill_parameters(model)
    frames = recipe.radar_batch(2, 10, 256, name="convlstm")
    x, tgt = frames[:, :5], frames[:, 5:]
    out = model(x)
    crit = ref.loss.Weighted_mse_mae(thresholds=[20, 30, 35, 40])   # train_untils.py:25,61 (Shanghai thresholds)
    loss = crit(out, tgt)
    loss.backward()
    names = [k for k, _ in model.named_parameters()]
    gnorm = np.array([float(p.grad.double().norm()) if p.grad is not None else -1.0 for _, p in model.named_parameters()])
    flat = out.flatten()
    idx = torch.from_numpy((recipe.uniform01("convlstm.sample", 4096) * flat.numel()).astype(np.int64))
    shapes = {k: list(v.shape) for k, v in model.state_dict().items()}
    save(name, out_idx=idx, out_samples=flat[idx], out_norm=out.double().norm(), loss=loss, grad_norms=gnorm, names=np.array(names),
         state_keys=np.array(list(shapes)), state_shapes=np.array([str(v) for v in shapes.values()]),
         n_params=np.array(sum(p.numel() for p in model.parameters())))


if __name__ == "__main__":
    import argparse
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default=None, help="regex: regenerate only the fixtures whose name matches")
    ONLY = ap.parse_args().only
    main()
