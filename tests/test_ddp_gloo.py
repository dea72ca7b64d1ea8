"""N>1 path on CPU: 2 gloo ranks, the gradient bucketer (adnm_hip.ddp.GradBuckets) must reproduce the
average of the per-rank gradients, leave never-used parameters with grad=None, survive
optimizer.zero_grad(set_to_none=True) between steps, and keep replicas bit-identical after AdamW steps."""
import os
import socket
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
import torch.nn as nn


def _plain(o):
    """tensors -> numpy before crossing the process boundary (no shared-memory handles that outlive the worker)."""
    if torch.is_tensor(o):
        return o.detach().numpy().copy()
    if isinstance(o, dict):
        return {k: _plain(v) for k, v in o.items()}
    if isinstance(o, (list, tuple)):
        return [_plain(v) for v in o]
    return o


def _torchify(o):
    import numpy as np
    if isinstance(o, np.ndarray):
        return torch.from_numpy(o)
    if isinstance(o, dict):
        return {k: _torchify(v) for k, v in o.items()}
    if isinstance(o, list):
        return [_torchify(v) for v in o]
    return o


class Toy(nn.Module):
    def __init__(self):
        super().__init__()
        torch.manual_seed(0)
        self.a = nn.Linear(8, 16)
        self.dead = nn.Linear(16, 16)   # never used in forward: must keep grad None (like e2ds[3..6])
        self.b = nn.Linear(16, 4)
        self.s = nn.Parameter(torch.tensor(1.0))

    def forward(self, x):
        return self.forward_stage2(*self.forward_stage1(x))

    # the cut FlatTrainer uses for its two-stage backward on > 1 rank (all-reduce of stage 2's gradients while stage 1's backward
    # runs); the hidden tensor is handed over twice on purpose: the trainer must differentiate it once
    def forward_stage1(self, x):
        h = torch.tanh(self.a(x))
        return (h, h)

    def forward_stage2(self, h1, h2):
        return self.b(0.5 * (h1 + h2)) * self.s

    def stage1_parameters(self):
        return self.a.parameters()


class Toy3(nn.Module):
    """three stages through forward_stages() (the form ADNM-UNet offers: five stages); `skip` crosses TWO cuts unchanged, as the U-Net's
    skip tensors do, and is used again in the last stage"""

    def __init__(self):
        super().__init__()
        torch.manual_seed(0)
        self.a, self.m, self.b = nn.Linear(8, 16), nn.Linear(16, 16), nn.Linear(16, 4)
        self.dead = nn.Linear(16, 16)
        self.s = nn.Parameter(torch.tensor(1.0))

    def forward(self, x):
        a = (x,)
        for fn, _ in self.forward_stages():
            a = fn(*a)
        return a[0]

    def forward_stages(self):
        s0 = lambda x: (torch.tanh(self.a(x)),)
        s1 = lambda h: (torch.tanh(self.m(h)) + h, h)             # (new, skip)
        s2 = lambda h, skip: (self.b(h + 0.5 * skip) * self.s,)
        return [(s0, [self.a]), (s1, [self.m]), (s2, [self.b, _Holder(self.s)])]


class _Holder(nn.Module):
    def __init__(self, p):
        super().__init__()
        self.p = p


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "adnm-unet_amd"))
    from adnm_hip.ddp import GradBuckets
    model = Toy()
    buckets = GradBuckets(model, bucket_mb=0.0005)  # force several buckets
    opt = torch.optim.AdamW(model.parameters(), lr=1e-2, eps=1e-9, weight_decay=1e-2)
    torch.manual_seed(100 + rank)
    out = {}
    for step in range(3):
        x = torch.randn(5, 8)
        loss = model(x).pow(2).mean()
        loss.backward()
        local = {k: p.grad.clone() for k, p in model.named_parameters() if p.grad is not None}
        buckets.finalize()
        if step == 0:
            out["local"], out["avg"] = local, {k: p.grad.clone() for k, p in model.named_parameters() if p.grad is not None}
            out["dead_none"] = model.dead.weight.grad is None
            out["nbuckets"] = len(buckets.buckets)
        opt.step()
        if step == 1:
            opt.zero_grad(set_to_none=True)   # train.py's call: drops the flat views, hooks must re-attach
        else:
            buckets.zero_grad()
    out["final"] = {k: p.detach().clone() for k, p in model.named_parameters()}
    q.put((rank, _plain(out)))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gradient_average():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = {r: _torchify(o) for r, o in (q.get(timeout=120) for _ in range(2))}
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    a, b = res[0], res[1]
    assert a["dead_none"] and b["dead_none"]
    assert a["nbuckets"] > 1
    for k in a["local"]:
        expect = (a["local"][k] + b["local"][k]) / 2
        assert torch.allclose(a["avg"][k], expect, atol=1e-7), k
        assert torch.equal(a["avg"][k], b["avg"][k]), k
    for k in a["final"]:
        assert torch.equal(a["final"][k], b["final"][k]), f"replicas diverged at {k}"


def _trainer_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "adnm-unet_amd"))
    from adnm_hip.trainer import FlatTrainer
    model = Toy()
    tr = FlatTrainer(model, lambda o, t: (o - t).pow(2).mean(), lr=1e-2, eps=1e-9, weight_decay=1e-2, max_norm=0.5,
                     use_graph=False, fused=False, stages=True)  # CPU: flattening + two-stage backward + all-reduce (HIP optimiser needs a GPU)
    torch.manual_seed(100 + rank)
    xs = [torch.randn(5, 8) for _ in range(3)]
    ts = [torch.randn(5, 4) for _ in range(3)]
    for x, t in zip(xs, ts):
        tr.step(x, t)
    assert tr.staged and len(tr.early) == 2 and len(tr.late) == 3 and 0 < tr.n_late < tr.n   # 2 ranks -> two-stage backward
    q.put((rank, _plain({"final": {k: p.detach().clone() for k, p in model.named_parameters()}, "xs": xs, "ts": ts,
                         "n_used": len(tr.used)})))
    dist.barrier()
    dist.destroy_process_group()


def test_flat_trainer_two_ranks_equals_global_batch():
    """2 ranks x batch 5 with averaged gradients == 1 process with the concatenated batch of 10 (mean loss)."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_trainer_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = {r: _torchify(o) for r, o in (q.get(timeout=120) for _ in range(2))}
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    a, b = res[0], res[1]
    for k in a["final"]:
        assert torch.equal(a["final"][k], b["final"][k]), f"replicas diverged at {k}"
    assert a["n_used"] == 5  # a.weight, a.bias, b.weight, b.bias, s — not the two `dead` tensors
    # single-process reference on the global batch
    model = Toy()
    opt = torch.optim.AdamW(model.parameters(), lr=1e-2, eps=1e-9, weight_decay=1e-2)
    for i in range(3):
        x = torch.cat([a["xs"][i], b["xs"][i]])
        t = torch.cat([a["ts"][i], b["ts"][i]])
        (model(x) - t).pow(2).mean().backward()
        torch.nn.utils.clip_grad_norm_(model.parameters(), 0.5)
        opt.step()
        opt.zero_grad(set_to_none=True)
    for k, p in model.named_parameters():
        assert torch.allclose(a["final"][k], p, atol=2e-6, rtol=1e-5), k


def _attach_worker(rank, world, port, q):
    """The reference's train.py loop, unmodified (train.py:132-146): forward, loss, backward, clip_grad_norm_, optimizer.step,
    zero_grad — on a model that only went through adnm_hip.ddp.attach (what create_ADNMUNet does when WORLD_SIZE > 1)."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "adnm-unet_amd"))
    from adnm_hip import ddp
    torch.manual_seed(7 + rank)          # different initial weights per rank: attach() must broadcast rank 0's
    model = Toy()
    with torch.no_grad():
        for p in model.parameters():
            p.add_(0.01 * rank)
    ddp.attach(model)                    # no process group yet: created (gloo on CPU) on the first forward
    opt = torch.optim.AdamW(model.parameters(), lr=1e-2, eps=1e-9, weight_decay=1e-2)
    torch.manual_seed(100 + rank)
    xs = [torch.randn(5, 8) for _ in range(3)]
    ts = [torch.randn(5, 4) for _ in range(3)]
    for x, t in zip(xs, ts):
        loss = (model(x) - t).pow(2).mean()
        loss.backward()
        torch.nn.utils.clip_grad_norm_(model.parameters(), 0.5)
        opt.step()
        opt.zero_grad()
    q.put((rank, _plain({"final": {k: p.detach().clone() for k, p in model.named_parameters()}, "xs": xs, "ts": ts,
                         "dead_none": model.dead.weight.grad is None})))
    dist.barrier()
    dist.destroy_process_group()


def test_unmodified_training_loop_with_attach():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_attach_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = {r: _torchify(o) for r, o in (q.get(timeout=120) for _ in range(2))}
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    a, b = res[0], res[1]
    assert a["dead_none"] and b["dead_none"]
    for k in a["final"]:
        assert torch.equal(a["final"][k], b["final"][k]), f"replicas diverged at {k}"
    # single process on the global batch, started from rank 0's weights (seed 7, no offset)
    torch.manual_seed(7)
    model = Toy()
    opt = torch.optim.AdamW(model.parameters(), lr=1e-2, eps=1e-9, weight_decay=1e-2)
    for i in range(3):
        x = torch.cat([a["xs"][i], b["xs"][i]])
        t = torch.cat([a["ts"][i], b["ts"][i]])
        (model(x) - t).pow(2).mean().backward()
        torch.nn.utils.clip_grad_norm_(model.parameters(), 0.5)
        opt.step()
        opt.zero_grad()
    for k, p in model.named_parameters():
        assert torch.allclose(a["final"][k], p, atol=2e-6, rtol=1e-5), k


def _bf16_trainer_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "adnm-unet_amd"))
    from adnm_hip.trainer import FlatTrainer
    model = Toy()
    # the default for > 1 rank: overlap="auto" -> two-stage backward, late bucket reduced beside the early stage; bf16 wire format
    tr = FlatTrainer(model, lambda o, t: (o - t).pow(2).mean(), lr=1e-2, eps=1e-9, weight_decay=1e-2, max_norm=0.5,
                     use_graph=False, fused=False, reduce_dtype="bf16")
    torch.manual_seed(100 + rank)
    for _ in range(3):
        tr.step(torch.randn(5, 8), torch.randn(5, 4))
    assert tr.staged and len(tr.buckets) == 2
    q.put((rank, _plain({"final": {k: p.detach().clone() for k, p in model.named_parameters()}, "g": tr.flat_g.clone()})))
    dist.barrier()
    dist.destroy_process_group()


def test_flat_trainer_default_overlap_and_bf16_wire():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_bf16_trainer_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = {r: _torchify(o) for r, o in (q.get(timeout=120) for _ in range(2))}
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for k in res[0]["final"]:
        assert torch.equal(res[0]["final"][k], res[1]["final"][k]), f"replicas diverged at {k}"
    assert torch.equal(res[0]["g"], res[1]["g"])


def _trainer3_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "adnm-unet_amd"))
    from adnm_hip.trainer import FlatTrainer
    model = Toy3()
    tr = FlatTrainer(model, lambda o, t: (o - t).pow(2).mean(), lr=1e-2, eps=1e-9, weight_decay=1e-2, max_norm=0.5, use_graph=False, fused=False)
    torch.manual_seed(100 + rank)
    xs = [torch.randn(5, 8) for _ in range(3)]
    ts = [torch.randn(5, 4) for _ in range(3)]
    for x, t in zip(xs, ts):
        tr.step(x, t)
    assert tr.staged and len(tr.stage_defs) == 3 and len(tr.buckets) == 3
    assert [len(g) for g in tr.groups] == [3, 2, 2]            # backward order: (b.weight, b.bias, s) | m | a
    assert all(lo % 8 == 0 for lo, _ in tr.buckets)           # bucket boundaries are 16-byte aligned in the bf16 wire buffer too
    q.put((rank, _plain({"final": {k: p.detach().clone() for k, p in model.named_parameters()}, "xs": xs, "ts": ts, "buckets": list(tr.buckets)})))
    dist.barrier()
    dist.destroy_process_group()


def test_flat_trainer_three_stages_four_ranks():
    """4 ranks x batch 5, staged backward over THREE stages (a bucket per stage, all-reduced while the earlier stages' backward runs)
    == 1 process with the concatenated batch of 20; the bucket map is identical on every rank."""
    world = 4
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_trainer3_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = {r: _torchify(o) for r, o in (q.get(timeout=180) for _ in range(world))}
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for r in range(1, world):
        assert res[r]["buckets"] == res[0]["buckets"]
        for k in res[0]["final"]:
            assert torch.equal(res[0]["final"][k], res[r]["final"][k]), f"replicas diverged at {k}"
    model = Toy3()
    opt = torch.optim.AdamW(model.parameters(), lr=1e-2, eps=1e-9, weight_decay=1e-2)
    for i in range(3):
        x = torch.cat([res[r]["xs"][i] for r in range(world)])
        t = torch.cat([res[r]["ts"][i] for r in range(world)])
        (model(x) - t).pow(2).mean().backward()
        torch.nn.utils.clip_grad_norm_(model.parameters(), 0.5)
        opt.step()
        opt.zero_grad(set_to_none=True)
    for k, p in model.named_parameters():
        if k.startswith("dead"):
            continue
        assert torch.allclose(res[0]["final"][k], p, atol=2e-6, rtol=1e-5), k


def test_real_model_bucket_map_is_deterministic():
    """The flat layout / bucket map of the REAL model (669 used, 307 unused tensors at config 2; five stages in backward order:
    refiner | decoder blocks | e2ds + fusion | encoder4-6 | encoder1-3), derived on the CPU from the reference fixture's list of
    parameters that receive a gradient: identical for two independently built replicas, every tensor and every bucket boundary
    16-byte aligned (fp32 and bf16 wire buffers), every used parameter in exactly one bucket."""
    import sys
    import numpy as np
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "adnm-unet_amd"))
    from adnm_hip.trainer import FlatTrainer
    z = np.load(os.path.join(root, "tests", "golden", "visionmamba_64_b2.npz"), allow_pickle=False)
    names, gn = [str(n) for n in z["names"]], z["grad_norms"]
    used_names = {n for n, g in zip(names, gn) if g >= 0}

    def plan():
        from models.ADNMUNet import create_ADNMUNet
        model = create_ADNMUNet(5, 20, 6, img_size=64)
        for n, p in model.named_parameters():
            p.grad = torch.zeros_like(p) if n in used_names else None      # a marker: _flatten only asks "is there a gradient"
        tr = FlatTrainer(model, None, use_graph=False, fused=False, overlap=True)
        tr._flatten()
        name_of = {id(p): n for n, p in model.named_parameters()}
        offs = {name_of[id(p)]: int((p.data_ptr() - tr.flat_p.data_ptr()) // 4) for p in tr.used}
        return offs, list(tr.buckets), [len(g) for g in tr.groups], tr.n

    a, b = plan(), plan()
    assert a == b
    offs, buckets, sizes, n = a
    assert len(offs) == len(used_names) == 669 and sum(sizes) == 669 and len(buckets) == 5
    assert all(o % 4 == 0 for o in offs.values()) and all(lo % 8 == 0 for lo, _ in buckets)
    assert buckets[0][0] == 0 and buckets[-1][1] == n and all(buckets[i][1] <= buckets[i + 1][0] for i in range(4))
    bucket_of = lambda o: next(i for i, (lo, hi) in enumerate(buckets) if lo <= o < hi)
    assert bucket_of(offs["refiner.refiner1.mixer_layers.0.in_proj.weight"]) == 0
    assert bucket_of(offs["decoder.decoder1.mixer_layers.0.in_proj.weight"]) == 1
    assert bucket_of(offs["decoder.e2ds.0.ffd.project_in.conv.weight"]) == 2 and bucket_of(offs["decoder.fusion.att7.weight"]) == 2
    assert bucket_of(offs["encoder.encoder6.mixer_layers.0.in_proj.weight"]) == 3 and bucket_of(offs["encoder.attn2.attn_mlp.fc1.weight"]) == 3
    assert bucket_of(offs["encoder.encoder1.conv2.0.conv.weight"]) == 4 and bucket_of(offs["encoder.attn.attn_mlp.fc1.weight"]) == 4


def _frozen_mid_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "adnm-unet_amd"))
    from adnm_hip.trainer import FlatTrainer
    model = Toy3()
    for p in model.m.parameters():   # the MIDDLE stage is frozen: its bucket is empty, the buckets either side keep their parts
        p.requires_grad_(False)
    tr = FlatTrainer(model, lambda o, t: (o - t).pow(2).mean(), lr=1e-2, eps=1e-9, weight_decay=1e-2, max_norm=0.5, use_graph=False, fused=False)
    torch.manual_seed(100 + rank)
    xs = [torch.randn(5, 8) for _ in range(3)]
    ts = [torch.randn(5, 4) for _ in range(3)]
    for x, t in zip(xs, ts):
        tr.step(x, t)
    assert tr.staged and len(tr.buckets) == len(tr.stage_defs) == 3
    assert [len(g) for g in tr.groups] == [3, 0, 2] and tr.buckets[1][0] == tr.buckets[1][1]
    q.put((rank, _plain({"final": {k: p.detach().clone() for k, p in model.named_parameters()}, "xs": xs, "ts": ts})))
    dist.barrier()
    dist.destroy_process_group()


def test_flat_trainer_frozen_middle_stage():
    """a stage without trainable parameters must keep its (empty) bucket, so that bucket j still belongs to backward part j:
    2 ranks x batch 5 with the middle stage frozen == 1 process on the concatenated batch."""
    world = 2
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_frozen_mid_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = {r: _torchify(o) for r, o in (q.get(timeout=180) for _ in range(world))}
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for k in res[0]["final"]:
        assert torch.equal(res[0]["final"][k], res[1]["final"][k]), f"replicas diverged at {k}"
    model = Toy3()
    for p in model.m.parameters():
        p.requires_grad_(False)
    opt = torch.optim.AdamW([p for p in model.parameters() if p.requires_grad], lr=1e-2, eps=1e-9, weight_decay=1e-2)
    for i in range(3):
        x = torch.cat([res[r]["xs"][i] for r in range(world)])
        t = torch.cat([res[r]["ts"][i] for r in range(world)])
        (model(x) - t).pow(2).mean().backward()
        torch.nn.utils.clip_grad_norm_([p for p in model.parameters() if p.requires_grad], 0.5)
        opt.step()
        opt.zero_grad(set_to_none=True)
    for k, p in model.named_parameters():
        if k.startswith("dead"):
            continue
        assert torch.allclose(res[0]["final"][k], p, atol=2e-6, rtol=1e-5), k
