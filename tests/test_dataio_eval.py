"""The data formats on either side of the hot path: input ingest (uint8 -> /255 -> bilinear resize),
the evaluator's contingency counts / error sums, and checkpoint compatibility.  CPU: the oracle against the fixtures
oracle/make_golden.py generated (evaluator: the reference's own SimplifiedEvaluator; resize: torch's F.interpolate, which is
what torchvision's tensor Resize calls — torchvision itself is not installed in the build container).  GPU: the HIP kernels
against the same fixtures."""
import numpy as np
import pytest
import torch

import adnm_oracle as O
from util import load_npz, assert_close

THR = [20, 30, 35, 40]


def test_oracle_resize_vs_fixture():
    z = load_npz("radar_resize_565x784_to_128")
    out = O.radar_resize(np.asarray(z["small_u8"]), 16)
    assert_close(out, z["small_out"], 1e-6, "small resize")


def test_oracle_evaluator_vs_reference():
    z = load_npz("evaluator_b3_t5")
    counts, mae, mse = O.evaluator_counts(z["truth"], z["pred"], float(z["value_scale"]), THR)
    for thr in THR:
        for i, k in enumerate(("hits", "misses", "falsealarms", "correctnegs")):
            assert torch.equal(counts[thr][..., i].long(), z[f"{k}.{thr}"].long()), (thr, k)
    assert_close(mse, z["mse"], 1e-5, "mse")
    assert_close(mae, z["mae"], 1e-5, "mae")
    res, far, rmse = O.evaluator_done(counts, mse)
    for thr in THR:
        for k in ("CSI", "POD", "HSS"):
            assert abs(res[thr][k] - float(z[f"{k}.{thr}"])) <= 1e-9, (thr, k)
    assert abs(far - float(z["FAR"])) <= 1e-9 and abs(rmse - float(z["RMSE"])) <= 1e-5 * float(z["RMSE"])
    # SSIM: the reference's cal_ssim, run by oracle/make_golden.py with the two cv2 calls replaced by their documented formulas
    ssim = O.evaluator_ssim(z["truth"], z["pred"], float(z["value_scale"]))
    assert_close(ssim, z["ssim"], 1e-9, "per-frame SSIM")
    assert abs(float(ssim.mean()) - float(z["SSIM"])) <= 1e-9


def test_checkpoint_roundtrip_with_dataparallel_prefix(tmp_path):
    """train.py:169-178 saves state_dict() (keys prefixed `module.` under nn.DataParallel); validate.py:86 loads it."""
    from adnm_hip import checkpoint, recipe
    from models.ADNMUNet import create_block
    a, b = create_block(32, 16, headdim=4), create_block(32, 16, headdim=4)
    recipe.fill_parameters(a)
    path = str(tmp_path / "Block_best.pth")
    n = checkpoint.save_reference_checkpoint(a, path, data_parallel_prefix=True)
    assert all(k.startswith("module.") for k in torch.load(path))
    assert checkpoint.load_reference_checkpoint(b, path) == n == len(a.state_dict())
    for (k, v), (_, w) in zip(a.state_dict().items(), b.state_dict().items()):
        assert torch.equal(v, w), k
    bad = {k: v for k, v in torch.load(path).items()}
    bad.pop(next(iter(bad)))
    with pytest.raises(RuntimeError, match="missing"):
        checkpoint.load_reference_checkpoint(b, bad)
    wrong = checkpoint.strip_prefix(torch.load(path))
    k0 = "norm1_layers.0.weight"
    wrong[k0] = torch.zeros(7)
    with pytest.raises(RuntimeError, match="shapes differ"):
        checkpoint.load_reference_checkpoint(b, wrong)


def test_checkpoint_full_model_keys():
    """The 992-key manifest recorded from the reference loads into this package's model, plain and `module.`-prefixed."""
    import json, os
    from adnm_hip import checkpoint, recipe
    from models.ADNMUNet import create_ADNMUNet
    from util import GOLDEN
    with open(os.path.join(GOLDEN, "state_dict_manifest.json")) as f:
        manifest = json.load(f)
    sd = recipe.state_dict_from_manifest(manifest)
    model = create_ADNMUNet(5, 20, 6, img_size=64)
    frozen = {k: v.clone() for k, v in model.state_dict().items() if not manifest[k]["trainable"]}
    sd.update(frozen)   # the manifest rebuilds frozen Haar filters as zeros: keep the model's own
    assert checkpoint.load_reference_checkpoint(model, {"module." + k: v for k, v in sd.items()}) == 992
    got = model.state_dict()
    assert all(torch.equal(got[k], sd[k]) for k in sd)


# ------------------------------------------------------------------------------------------------ GPU
@pytest.mark.gpu
def test_radar_ingest_vs_fixture():
    from adnm_hip import dataio
    z = load_npz("radar_resize_565x784_to_128")
    small = torch.as_tensor(np.asarray(z["small_u8"])).to(torch.uint8).cuda()[None]      # (1, 3, 57, 79)
    out = dataio.ingest(small, 16)
    assert out.shape == (1, 3, 1, 16, 16)
    assert_close(out[0, :, 0], z["small_out"], 1e-6, "small resize")
    from adnm_hip import recipe
    big = torch.from_numpy((recipe.uniform01("resize.big", 2 * 565 * 784) * 71).astype(np.uint8).reshape(1, 2, 565, 784)).cuda()
    ob = dataio.ingest(big, 128)
    assert_close(ob.flatten()[z["big_idx"].cuda()], z["big_samples"], 1e-6, "565x784 -> 128x128 samples")
    assert abs(float(ob.double().sum()) - float(z["big_sum"])) <= 1e-6 * float(z["big_sum"])


@pytest.mark.gpu
def test_radar_ingest_double_buffered_stream():
    """RadarIngest: pinned staging + copy stream, two batches in flight, split 5 / 20 as train.py:133."""
    from adnm_hip import dataio
    rng = np.random.default_rng(0)
    ing = dataio.RadarIngest(2, 25, 40, 56, 32, "cuda", in_frames=5)
    batches = [rng.integers(0, 71, size=(2, 25, 40, 56), dtype=np.uint8) for _ in range(4)]
    ing.submit(batches[0])
    for i in range(4):
        imgs, tgts = ing.take()
        if i + 1 < 4:
            ing.submit(batches[i + 1])
        assert imgs.shape == (2, 5, 1, 32, 32) and tgts.shape == (2, 20, 1, 32, 32)
        ref = torch.stack([O.radar_resize(b, 32) for b in batches[i]])
        assert_close(torch.cat((imgs, tgts), 1)[:, :, 0], ref, 1e-6, f"batch {i}")


@pytest.mark.gpu
def test_gpu_evaluator_vs_reference():
    from adnm_hip.evaluator import GpuEvaluator
    z = load_npz("evaluator_b3_t5")
    ev = GpuEvaluator(seq_len=5, value_scale=float(z["value_scale"]), thresholds=THR)
    t, p = z["truth"].cuda(), z["pred"].cuda()
    ev.evaluate(t[:2], p[:2])
    ev.evaluate(t[2:].unsqueeze(2), p[2:].unsqueeze(2))   # (B, T, 1, H, W) as the model emits it
    res = ev.done()
    for thr in THR:
        m = res["threshold_metrics"][thr]
        for k in ("TP", "TN", "FP", "FN"):
            assert m[k] == float(z[f"{k}.{thr}"]), (thr, k)
        for k in ("CSI", "POD", "HSS"):
            assert abs(m[k] - float(z[f"{k}.{thr}"])) <= 1e-9, (thr, k)
    assert abs(res["FAR"] - float(z["FAR"])) <= 1e-9
    assert abs(res["SSIM"] - float(z["SSIM"])) <= 1e-6, (res["SSIM"], float(z["SSIM"]))   # float64 moments on the device; per-tile sums folded in fp32
    assert abs(res["RMSE"] - float(z["RMSE"])) <= 1e-5 * float(z["RMSE"])
    assert abs(res["MSE"] - float(z["mse"].mean())) <= 1e-5 * float(z["mse"].mean())


@pytest.mark.gpu
def test_graphed_eval_forward_matches_eager():
    from adnm_hip import recipe
    from adnm_hip.evaluator import GraphedForward
    from models.ADNMUNet import create_ADNMUNet
    model = create_ADNMUNet(5, 20, 6, img_size=64)
    recipe.fill_parameters(model)
    model = model.cuda().eval()
    fwd = GraphedForward(model)
    for salt in (0, 1):
        x = recipe.radar_batch(2, 5, 64, salt=salt, name="evalfwd").cuda()
        with torch.no_grad():
            ref = model(x)
        out = fwd(x)
        assert out.shape == (2, 20, 1, 64, 64) and not out.requires_grad
        assert torch.equal(out, ref)
