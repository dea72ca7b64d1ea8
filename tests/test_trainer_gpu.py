"""FlatTrainer on the GPU: the fused clip + AdamW kernels against torch.nn.utils.clip_grad_norm_ +
torch.optim.AdamW (the reference's recipe, train.py:140, train_untils.py:35-42), eager and hipGraph replay."""
import copy
import pytest
import torch

from adnm_hip import recipe
from adnm_hip.trainer import FlatTrainer
from util import assert_close

pytestmark = pytest.mark.gpu
DEV = "cuda"


def small_model():
    from models.ADNMUNet import create_block
    torch.manual_seed(0)
    m = create_block(32, 16, headdim=4, norm_epsilon=1e-6)
    recipe.fill_parameters(m)
    return m.to(DEV)


@pytest.mark.parametrize("use_graph", [False, True])
@pytest.mark.parametrize("max_norm", [0.0, 0.05])
def test_fused_step_matches_torch(use_graph, max_norm):
    ref = small_model()
    mine = copy.deepcopy(ref)
    x, tgt = recipe.tensor("tr.x", (2, 64, 32)).to(DEV), recipe.tensor("tr.t", (2, 64, 16)).to(DEV)
    loss_fn = lambda o, t: ((o - t) ** 2).mean()
    opt = torch.optim.AdamW(ref.parameters(), lr=1e-3, betas=(0.9, 0.999), eps=1e-9, weight_decay=1e-2)
    tr = FlatTrainer(mine, loss_fn, lr=1e-3, betas=(0.9, 0.999), eps=1e-9, weight_decay=1e-2, max_norm=max_norm, use_graph=use_graph)
    for step in range(4):
        loss_ref = loss_fn(ref(x), tgt)
        loss_ref.backward()
        if max_norm > 0:
            norm_ref = torch.nn.utils.clip_grad_norm_(ref.parameters(), max_norm)
        opt.step()
        opt.zero_grad(set_to_none=True)
        loss = tr.step(x, tgt)
        assert abs(float(loss) - float(loss_ref)) <= 1e-5 * abs(float(loss_ref)) + 1e-7
        if max_norm > 0:
            assert abs(float(tr.grad_norm()) - float(norm_ref)) <= 1e-4 * float(norm_ref)
    for (k, a), (_, b) in zip(mine.named_parameters(), ref.named_parameters()):
        assert_close(a, b, 2e-5, k, atol=1e-6)
    # parameters that never receive a gradient are untouched (no decay), as torch.optim.AdamW leaves them
    unused = [k for k, p in mine.named_parameters() if all(p is not q for q in tr.used)]
    assert "alpha1" in unused and "act.beta" in unused


@pytest.mark.parametrize("use_graph,stages", [(False, False), (True, False), (False, True), (True, True)])
def test_whole_model_step_vs_reference_fixture(use_graph, stages):
    """FlatTrainer (flat buffers, channels-last conv weights, in-place gradient destinations, hipGraph replay, fused
    clip + AdamW) on the whole ADNM-UNet: loss, pre-clip gradient norm and every parameter after one step against the values the
    reference produced with clip_grad_norm_(0.025) + torch.optim.AdamW (tests/golden, train.py:140, train_untils.py:35-42)."""
    from models.ADNMUNet import create_ADNMUNet
    from models.loss import enRainfallLoss
    from util import load_npz
    z = load_npz("visionmamba_64_b2")
    model = create_ADNMUNet(5, 20, 6, img_size=64)
    recipe.fill_parameters(model)
    model = model.to(DEV).train()
    frames = recipe.radar_batch(2, 25, 64, name="radar64").to(DEV)
    x, tgt = frames[:, :5], frames[:, 5:]
    # stages=True: the two-stage backward (encoder | decoder + refiner, two graphs) that multi-GPU runs use to overlap the gradient
    # all-reduce with the encoder's backward — same numbers required
    tr = FlatTrainer(model, enRainfallLoss(0.57, 0.25, gamma=0.0), lr=1e-3, betas=(0.9, 0.999), eps=1e-9, weight_decay=1e-2, max_norm=0.025,
                     use_graph=use_graph, stages=stages)
    assert tr.staged == stages
    before = {k: p.detach().clone() for k, p in model.named_parameters()}
    loss = tr.step(x, tgt)
    assert abs(float(loss) - float(z["loss"])) <= 1e-4 * abs(float(z["loss"]))
    assert abs(float(tr.grad_norm()) - float(z["clip_pre_norm"])) <= 1e-3 * float(z["clip_pre_norm"])
    if stages:   # five stages (refiner | decoder blocks | e2ds + fusion | encoder4-6 | encoder1-3): five graphs, five buckets
        assert len(tr.stage_defs) == 5 and len(tr.buckets) == 5 and len(tr.graphs) == (4 if use_graph else 0)
        assert 0 < tr.n_late < tr.n and all(len(g) > 20 for g in tr.groups)
    names, ref_sums, gn = [str(n) for n in z["names"]], z["param_sum_after_step"].numpy(), z["grad_norms"].numpy()
    named = dict(model.named_parameters())
    for i, k in enumerate(names):
        p = named[k]
        s = float(p.double().sum())
        assert abs(s - ref_sums[i]) <= 2.5e-3 * p.numel() ** 0.5 + 1e-5 * abs(ref_sums[i]) + 1e-6, k
        if gn[i] < 0:   # the reference leaves these without a gradient: neither updated nor decayed
            assert torch.equal(p, before[k]), k
    # state_dict still has the reference's logical shapes (conv weights are channels-last views of the flat buffer)
    assert model.state_dict()["refiner.out_proj.conv.0.conv.weight"].shape == (64, 32, 3, 3)


def test_benchmarked_workload_step_vs_reference_fixture():
    """The exact workload bench.py times (B=4, 128x128, 5->20, recipe batch "bench", hipGraph replay, fused clip + AdamW): loss,
    pre-clip norm and the UPDATE of every parameter (p_after - p_before; element-wise for the 360 scalar mixes) against what the
    reference produced for the same batch (fixture visionmamba_128_b4, oracle/make_golden.py)."""
    from models.ADNMUNet import create_ADNMUNet
    from models.loss import enRainfallLoss
    from util import load_npz, check_update_deltas
    z = load_npz("visionmamba_128_b4")
    model = create_ADNMUNet(5, 20, 6, img_size=128)
    recipe.fill_parameters(model)
    model = model.to(DEV).train()
    frames = recipe.radar_batch(4, 25, 128, name="bench").to(DEV)
    x, tgt = frames[:, :5].contiguous(), frames[:, 5:].contiguous()
    tr = FlatTrainer(model, enRainfallLoss(0.57, 0.25, gamma=0.0), lr=1e-3, betas=(0.9, 0.999), eps=1e-9, weight_decay=1e-2, max_norm=0.025,
                     use_graph=True)
    before = [p.detach().double().clone() for p in model.parameters()]
    loss = tr.step(x, tgt)
    assert abs(float(loss) - float(z["loss"])) <= 1e-4 * abs(float(z["loss"]))
    assert abs(float(tr.grad_norm()) - float(z["clip_pre_norm"])) <= 1e-3 * float(z["clip_pre_norm"])
    names = [str(n) for n in z["names"]]
    assert names == [k for k, _ in model.named_parameters()]
    check_update_deltas(z, names, [p.detach().double() - b for p, b in zip(model.parameters(), before)])


def test_deferred_folds_are_neutral_and_deterministic():
    """Batching the second-stage fold launches of the parameter gradients and grouping the weight-gradient launches (ops.FOLDS /
    adnm_foldq_* / adnm_leafq_*) must not change the step beyond fp32 summation order — a queued weight gradient splits its reduction
    into fewer slices than one launched alone, the fold arithmetic itself is the same — including Block.beta1/beta2, which receive two
    contributions; and the deferred step is deterministic: two runs agree bit for bit (no atomics anywhere)."""
    from models.ADNMUNet import create_ADNMUNet
    from models.loss import enRainfallLoss
    frames = recipe.radar_batch(1, 25, 64, name="defer").to(DEV)
    x, tgt = frames[:, :5].contiguous(), frames[:, 5:].contiguous()
    flats = []
    for defer in (False, True, True):
        model = create_ADNMUNet(5, 20, 6, img_size=64)
        recipe.fill_parameters(model)
        model = model.to(DEV).train()
        tr = FlatTrainer(model, enRainfallLoss(0.57, 0.25, gamma=0.0), max_norm=0.025, use_graph=False, defer_folds=defer)
        tr.prepare(x, tgt)
        from adnm_hip import lib
        lib.query("adnm_prof_enable", 1)
        tr._run_eager(x, tgt)
        torch.cuda.synchronize()
        lib.query("adnm_prof_enable", 0)
        import ctypes
        buf = ctypes.create_string_buffer(1 << 20)
        lib.query("adnm_prof_collect", buf, len(buf))
        nfold = sum(int(l.split("\t")[1]) for l in buf.value.decode().splitlines() if "fold" in l.split("\t")[0])
        flats.append((tr.flat_g.clone(), nfold))
        tr.close()
        del tr
    assert torch.equal(flats[1][0], flats[2][0]), "the deferred step is not deterministic"
    a, b = flats[0][0].double(), flats[1][0].double()
    assert float((a - b).norm()) <= 1e-6 * float(a.norm()), float((a - b).norm() / a.norm())
    assert float((a - b).abs().max()) <= 1e-5 * float(a.abs().max())
    assert flats[1][1] < 0.5 * flats[0][1], (flats[0][1], flats[1][1])


def test_bf16_wire_casts_on_the_real_bucket_boundaries():
    """The bf16 wire format of the gradient all-reduce (reduce_dtype="bf16"): the HIP cast kernels need 16-byte aligned ranges on both
    sides, so every stage boundary of the staged layout must fall on a multiple of 8 elements — checked on the benchmarked model's
    real five buckets, with the round trip through the wire buffer (a single process cannot run the collective itself)."""
    from models.ADNMUNet import create_ADNMUNet
    from models.loss import enRainfallLoss
    model = create_ADNMUNet(5, 20, 6, img_size=64)
    recipe.fill_parameters(model)
    model = model.to(DEV).train()
    frames = recipe.radar_batch(1, 25, 64, name="wire").to(DEV)
    x, tgt = frames[:, :5].contiguous(), frames[:, 5:].contiguous()
    tr = FlatTrainer(model, enRainfallLoss(0.57, 0.25, gamma=0.0), max_norm=0.025, use_graph=False, overlap=True, reduce_dtype="bf16")
    tr.prepare(x, tgt)
    tr._run_eager(x, tgt)
    assert len(tr.buckets) == 5 and all(lo % 8 == 0 for lo, _ in tr.buckets)
    comm = torch.zeros(tr.n, dtype=torch.bfloat16, device=DEV)   # (zeros: the alignment gaps between buckets belong to no bucket)
    back = torch.empty_like(tr.flat_g)
    for lo, hi in tr.buckets:
        tr._cast(tr.flat_g[lo:hi], comm[lo:hi])
        tr._cast(comm[lo:hi], back[lo:hi], 0.5)
    torch.cuda.synchronize()
    for lo, hi in tr.buckets:
        assert torch.equal(comm[lo:hi], tr.flat_g[lo:hi].to(torch.bfloat16))
        assert torch.equal(back[lo:hi], comm[lo:hi].float() * 0.5)


@pytest.mark.parametrize("use_graph", [False, True])
def test_side_stream_weight_gradients_are_bitwise_neutral(use_graph):
    """Launching the weight-gradient kernels on a second stream beside the input-gradient chain (ops.SIDE: fork / join events, parallel
    branches of the captured graph) must not change a bit of the parameters after several steps — any missing dependency would."""
    from models.ADNMUNet import create_ADNMUNet
    from models.loss import enRainfallLoss
    frames = recipe.radar_batch(2, 25, 64, name="side").to(DEV)
    x, tgt = frames[:, :5].contiguous(), frames[:, 5:].contiguous()
    flats = []
    for side in (False, True):
        model = create_ADNMUNet(5, 20, 6, img_size=64)
        recipe.fill_parameters(model)
        model = model.to(DEV).train()
        tr = FlatTrainer(model, enRainfallLoss(0.57, 0.25, gamma=0.0), lr=1e-3, max_norm=0.025, use_graph=use_graph, side_stream=side)
        for _ in range(4):
            tr.step(x, tgt)
        torch.cuda.synchronize()
        flats.append((tr.flat_p.clone(), tr.flat_g.clone()))
        del tr
    assert torch.equal(flats[0][1], flats[1][1]), "gradients differ with the side stream on"
    assert torch.equal(flats[0][0], flats[1][0]), "parameters differ with the side stream on"


def test_bf16_shadow_weights_are_bitwise_neutral(monkeypatch):
    """The bf16 configuration reads its GEMM weights from the bf16 shadow the optimiser pass writes (and the mixer prep's bf16 copies):
    rounding the fp32 weight on the way into the MFMA or reading its pre-rounded twin is the same arithmetic — parameters and gradients
    after several steps must agree bit for bit with ADNM_NARROW_WEIGHTS=0."""
    from models.ADNMUNet import create_ADNMUNet
    from models.loss import enRainfallLoss
    from adnm_hip import ops
    frames = recipe.radar_batch(2, 25, 64, name="shadow").to(DEV)
    x, tgt = frames[:, :5].contiguous(), frames[:, 5:].contiguous()
    flats = []
    ops.set_mfma_precision("bf16")
    try:
        for narrow in ("0", "1"):
            monkeypatch.setenv("ADNM_NARROW_WEIGHTS", narrow)
            model = create_ADNMUNet(5, 20, 6, img_size=64)
            recipe.fill_parameters(model)
            model = model.to(DEV).train()
            tr = FlatTrainer(model, enRainfallLoss(0.57, 0.25, gamma=0.0), lr=1e-3, max_norm=0.025, use_graph=True)
            for _ in range(4):
                tr.step(x, tgt)
            torch.cuda.synchronize()
            assert (getattr(tr, "shadow_mode", 0) == 1) == (narrow == "1")
            if narrow == "1":
                assert torch.equal(tr.shadow, tr.flat_p.to(torch.bfloat16)), "the shadow is bf16(p) after every step"
            flats.append((tr.flat_p.clone(), tr.flat_g.clone()))
            tr.close()
            del tr
    finally:
        ops.set_mfma_precision("f32")
    assert torch.equal(flats[0][1], flats[1][1]), "gradients differ with the shadow weights"
    assert torch.equal(flats[0][0], flats[1][0]), "parameters differ with the shadow weights"


def test_tail_graph_follows_host_schedules():
    """The staged trainer (the multi-GPU form) replays its optimiser as a captured TAIL graph whose learning rate and clip threshold live in
    device memory: changing tr.lr / tr.max_norm between steps (train.py's warm-up + cosine schedule, its adaptive clip threshold) must give
    bit for bit what the eagerly launched optimiser gives."""
    from models.ADNMUNet import create_ADNMUNet
    from models.loss import enRainfallLoss
    frames = recipe.radar_batch(2, 25, 64, name="tail").to(DEV)
    x, tgt = frames[:, :5].contiguous(), frames[:, 5:].contiguous()
    flats = []
    for use_graph in (False, True):
        model = create_ADNMUNet(5, 20, 6, img_size=64)
        recipe.fill_parameters(model)
        model = model.to(DEV).train()
        tr = FlatTrainer(model, enRainfallLoss(0.57, 0.25, gamma=0.0), lr=1e-3, max_norm=0.025, use_graph=use_graph, overlap=True)
        for k in range(5):
            tr.lr, tr.max_norm = 1e-3 * (1.0 - 0.15 * k), 0.025 * (1.0 + 0.5 * k)
            tr.step(x, tgt)
        torch.cuda.synchronize()
        assert tr.staged and (getattr(tr, "tail", None) is not None) == use_graph
        flats.append((tr.flat_p.clone(), tr.exp_avg.clone()))
        tr.close()
        del tr
    assert torch.equal(flats[0][1], flats[1][1]) and torch.equal(flats[0][0], flats[1][0])
