"""Host-side logic that needs no GPU: the trainer's ownership / error path, the gradient-destination registry's
finaliser safety, the supervised multi-rank launcher of bench.py, RadarIngest's slot bookkeeping."""
import gc
import os
import subprocess
import sys
import time

import pytest
import torch
import torch.nn as nn

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class _Toy(nn.Module):
    def __init__(self):
        super().__init__()
        torch.manual_seed(0)
        self.a, self.b = nn.Linear(8, 16), nn.Linear(16, 4)

    def forward(self, x):
        return self.b(torch.tanh(self.a(x)))


def test_prepare_exception_path_restores_state():
    """An exception inside prepare() (here: a loss function that raises on the second call, i.e. after the dry run) must leave the
    collector enabled, no graph objects behind and the trainer reusable; close() is idempotent and drops the registration."""
    from adnm_hip import ops
    from adnm_hip.trainer import FlatTrainer
    calls = [0]

    def loss_fn(o, t):
        calls[0] += 1
        if calls[0] == 2:
            raise ValueError("boom")
        return (o - t).pow(2).mean()

    model = _Toy()
    tr = FlatTrainer(model, loss_fn, use_graph=False, fused=False)
    x, t = torch.randn(5, 8), torch.randn(5, 4)
    assert gc.isenabled()
    tr.prepare(x, t)            # first call of loss_fn: fine
    with pytest.raises(ValueError):
        tr.step(x, t)           # second call raises inside the eager step
    assert gc.isenabled()
    tr2 = FlatTrainer(_Toy(), loss_fn, use_graph=False, fused=False)
    calls[0] = 1
    with pytest.raises(ValueError):
        tr2.prepare(x, t)       # raises inside prepare's dry run
    assert gc.isenabled() and tr2.graph is None and tr2.graph2 is None and tr2.static_loss is None
    calls[0] = 10
    tr2.prepare(x, t)           # and the same object can be prepared again
    tr2.step(x, t)
    owner = id(tr2)
    tr2.close()
    tr2.close()
    ops.GRADS.reset_claims(owner)   # drains the finaliser queue
    assert owner not in ops.GRADS._claimed and not any(o == owner for o, _ in ops.GRADS._dst.values())


def test_grad_registry_drop_is_lock_free_and_reentrant():
    """drop() is what a finaliser calls: it must not take the registry lock (the cyclic collector can run while register() / take()
    hold it on the same thread)."""
    from adnm_hip.ops import GradRegistry
    reg = GradRegistry()
    g = torch.zeros(4)
    reg.register(1, {g.data_ptr(): g})
    with reg._lock:            # as if the collector fired inside take()
        reg.drop(1)            # must return at once
        with reg._lock:        # and the lock is re-entrant for the same thread
            pass
    t = reg.take(g.data_ptr(), (4,), g.device)
    assert t.data_ptr() != g.data_ptr()   # owner 1 was forgotten before the lookup
    # a slice whose layout differs from what the caller writes is refused WITHOUT being claimed
    w = torch.zeros(2, 3, 3, 5).permute(0, 3, 1, 2)   # (2, 5, 3, 3) logical, channels-last memory
    reg.register(2, {w.data_ptr(): w})
    got = reg.take(w.data_ptr(), (2, 5, 3, 3), w.device, strides=(45, 9, 3, 1))
    assert got.data_ptr() != w.data_ptr() and w.data_ptr() not in reg.born_in_place(2)
    got = reg.take(w.data_ptr(), (2, 5, 3, 3), w.device, strides=tuple(w.stride()))
    assert got.data_ptr() == w.data_ptr() and w.data_ptr() in reg.born_in_place(2)


def test_grad_registry_hands_out_dense_slices_under_another_shape():
    """A (4d, d, 1, 1) conv weight whose gradient a GEMM writes as (4d, d): the same dense memory, so the claim succeeds with the slice
    viewed the caller's way (and is recorded: the trainer's gather then skips the copy); a non-dense layout is still refused."""
    from adnm_hip.ops import GradRegistry
    reg = GradRegistry()
    g = torch.zeros(8, 4, 1, 1)
    reg.register(7, {g.data_ptr(): g})
    t = reg.take(g.data_ptr(), (8, 4), g.device)
    assert t.shape == (8, 4) and t.data_ptr() == g.data_ptr() and g.data_ptr() in reg.born_in_place(7)
    h = torch.zeros(6, 5)[:, :4]   # not dense
    reg.register(8, {h.data_ptr(): h})
    u = reg.take(h.data_ptr(), (24,), h.device)
    assert u.data_ptr() != h.data_ptr() and h.data_ptr() not in reg.born_in_place(8)
    k = torch.zeros(8, 4)
    reg.register(9, {k.data_ptr(): k})
    v = reg.take(k.data_ptr(), (4, 8), k.device, strides=(1, 4))   # the caller would write a transposed layout: refused
    assert v.data_ptr() != k.data_ptr() and k.data_ptr() not in reg.born_in_place(9)


def test_bench_multi_rank_launcher_reports_a_dead_rank():
    """`python bench.py --gpus 2` on a box without GPUs: both ranks exit at once ("needs a GPU"); the parent must notice, stop the
    job and exit non-zero within seconds instead of waiting for a rendezvous that never happens."""
    env = dict(os.environ, ADNM_BENCH_TIMEOUT="120")
    env.pop("WORLD_SIZE", None)
    if torch.cuda.is_available():
        pytest.skip("needs a box without a GPU (on a GPU box the 2-rank job would really start)")
    t0 = time.time()
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"], env=env,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode != 0
    assert "exited with status" in r.stderr and "needs a GPU" in r.stderr
    assert time.time() - t0 < 240


def test_radar_ingest_slot_bookkeeping():
    """submit, submit, take, take works; a third submit without a take raises (no staging buffer is overwritten under a copy)."""
    from adnm_hip import dataio
    ing = dataio.RadarIngest.__new__(dataio.RadarIngest)   # the bookkeeping alone: no pinned memory / device here
    ing._pending, ing._slot = [], 0
    ing._pending.append(0), ing._pending.append(1)
    with pytest.raises(RuntimeError):
        dataio.RadarIngest.submit(ing, None)
    assert ing._pending.pop(0) == 0 and ing._pending.pop(0) == 1


def test_bench_stdout_line_is_small_and_parses():
    """the driver parses ONE stdout line; it must stay under 4 kB however many kernels the profile holds.
    Canned profile = round 3's per-(kernel, shape) table (profiles/r03_kernel_shapes_bf16.tsv)."""
    import json
    import bench
    prof = {}
    with open(os.path.join(ROOT, "profiles", "r03_kernel_shapes_bf16.tsv")) as f:
        next(f)
        for row in f:
            key, n, avg_us, ms, _ = row.rstrip("\n").split("\t")
            n = max(1, round(float(n)))
            prof[key] = {"launches": 3 * n, "ms": 3 * float(ms), "bytes": 3 * n * float(key.split("@")[1])}
    assert len(prof) > 100
    pmc = {"kernels": {k.split("@")[0]: {"hbm_bytes_per_step": 1.5e8, "launches_per_step": 1} for k in prof}}
    kernels, families, roofline = bench.summarise(prof, 3, pmc, "canned " + "x" * 200, 8.5e-3, 11.0)
    assert roofline["kernel"] in kernels and roofline["traffic"] is not None and 0 < roofline["frac"] < 1
    res = {"metric": "sequences/sec training ADNM-UNet 5->20x128x128", "value": 468.4, "unit": "sequences/s", "n_gpus": 1, "steps": 20, "warmup": 10,
           "ms_per_step": 8.54, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "dtype_detail": "d" * 300,
           "data": "synthetic", "config": {"workload": "w" * 250, "per_gpu_batch": 4, "protocol": "p" * 150},
           "windows_ms_per_step": [8.5] * 5, "roofline": roofline,
           "cpu_baseline": {"value": 3.7, "unit": "sequences/s", "cores": 16, "kind": "port", "sample": "s" * 200}}
    line = bench.stdout_line(res, families, "bench_tables.json")
    assert "\n" not in line and len(line) < 4096
    back = json.loads(line)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "dtype", "config", "roofline", "cpu_baseline"):
        assert k in back
    assert back["roofline"]["frac"] == roofline["frac"] and 1 <= len(back["families_top"]) <= 8
    assert "kernels" not in back and "families" not in back
    # a pathological record still comes out under the cap (families dropped from the tail) ...
    fat = dict(res, dtype_detail="d" * 1500)
    assert len(bench.stdout_line(fat, families, "t.json")) < 4096
    # ... and one that cannot fit raises instead of printing a line the driver will not parse
    with pytest.raises(RuntimeError):
        bench.stdout_line(dict(res, dtype_detail="d" * 5000), families, "t.json")


def test_flat_trainer_refuses_unclaimed_and_doubly_claimed_parameters():
    """a parameter that receives a gradient but is named by no stage (it would never be trained, only decayed), or by two
    stages, is an error, not a silent layout decision."""
    from adnm_hip.trainer import FlatTrainer

    class M(nn.Module):
        def __init__(self, stages):
            super().__init__()
            torch.manual_seed(0)
            self.a, self.b = nn.Linear(4, 4), nn.Linear(4, 2)
            self.extra = nn.Parameter(torch.ones(()))
            self._stages = stages

        def forward(self, x):
            return self.b(torch.tanh(self.a(x))) * self.extra

        def forward_stages(self):
            s0 = lambda x: (torch.tanh(self.a(x)),)
            s1 = lambda h: (self.b(h) * self.extra,)
            return [(s0, self._stages[0](self)), (s1, self._stages[1](self))]

    x, t = torch.randn(3, 4), torch.randn(3, 2)
    loss = lambda o, tt: (o - tt).pow(2).mean()
    tr = FlatTrainer(M((lambda m: [m.a], lambda m: [m.b])), loss, use_graph=False, fused=False, overlap=True)
    with pytest.raises(RuntimeError, match="belong to no stage.*extra"):
        tr.prepare(x, t)
    tr = FlatTrainer(M((lambda m: [m.a, m.b], lambda m: [m.b])), loss, use_graph=False, fused=False, overlap=True)
    with pytest.raises(RuntimeError, match="more than one stage"):
        tr.prepare(x, t)
