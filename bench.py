#!/usr/bin/env python3
"""ADNM-UNet training-step benchmark (BASELINE.json metric: training sequences/sec, 5->20 x 128x128).

  python bench.py --gpus N --steps K --warmup W            (N=1: single process)
  python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...   (one rank per GPU, RCCL)

One step = forward -> enRainfallLoss -> backward -> (gradient all-reduce) -> clip_grad_norm_ -> AdamW.step ->
zero_grad on a synthetic radar batch already resident in HBM (train.py:133-146 of the reference).  Rank 0
prints ONE JSON line.  `roofline` is measured live with HIP events recorded by libadnm_hip around every one of
its kernel launches (adnm_prof_enable/collect) during extra instrumented steps after the timed region;
`cpu_baseline` times the oracle (oracle/adnm_oracle.py, the CPU restatement of the reference's algorithm) on
the host cores for the same workload, rank 0 / N=1 only.
"""
import argparse
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "adnm-unet_amd"))

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6.3 TB/s is the measured copy ceiling


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=4, help="per-GPU batch (BASELINE config 2: 4)")
    ap.add_argument("--size", type=int, default=128)
    ap.add_argument("--in-frames", type=int, default=5)
    ap.add_argument("--out-frames", type=int, default=20)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-steps", type=int, default=8, help="timed oracle steps of the CPU baseline (~1.5 s each on 16 threads: ~12-15 s in all)")
    ap.add_argument("--prof-steps", type=int, default=3)
    ap.add_argument("--stages", default="auto", choices=["auto", "0", "1"],
                    help="two-stage backward (encoder | decoder+refiner) that overlaps the gradient all-reduce with the encoder's backward: "
                         "auto = the ADNM_STAGES environment variable (default off: on one MI355X the second graph replay costs "
                         "more than an 8-rank ring takes)")
    ap.add_argument("--dump-prof", default="", help="write the per-(kernel, shape) HIP-event table of the instrumented steps to this file")
    ap.add_argument("--graph", type=int, default=1, help="1 (default): replay fwd+bwd as a captured hipGraph; 0: eager launches")
    ap.add_argument("--blas", default="hipblas", choices=["default", "hipblas", "hipblaslt"],
                    help="library used for the plain GEMMs (default rocBLAS: its long-reduction weight-grad GEMMs are 6x faster here)")
    return ap.parse_args()


def collect_profile(lib):
    buf = ctypes.create_string_buffer(1 << 20)  # one call: collecting also clears the records
    lib.query("adnm_prof_collect", buf, len(buf))
    rows = {}  # key "kernel@bytes_per_launch": one entry per kernel AND shape
    for line in buf.value.decode().splitlines():
        name, cnt, ms, nbytes = line.split("\t")
        rows[name] = {"launches": int(cnt), "ms": float(ms), "bytes": float(nbytes)}
    return rows


# profiler scope name -> device kernel name, for the kernels whose every launch has ONE shape (so the per-kernel PMC
# average of profiles/r01_pmc_traffic.json IS the per-launch traffic of that shape)
PMC_KERNEL = {"adamw_update": "adamw_kernel", "grad_sumsq": "sumsq_partial_kernel"}


def pmc_traffic(scope_name):
    """HBM bytes per launch from the committed rocprofv3 PMC passes (FETCH_SIZE x2 + WRITE_SIZE, tools/pmc_traffic.py) of this
    same command; None when that kernel runs several shapes or no PMC pass is committed."""
    path = os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")
    dev = PMC_KERNEL.get(scope_name)
    if dev is None or not os.path.exists(path):
        return None
    with open(path) as f:
        rec = json.load(f).get(dev)
    return round(rec["hbm_bytes_per_launch"]) if rec else None


def by_kernel(rows):
    out = {}
    for key, r in rows.items():
        a = out.setdefault(key.split("@")[0], {"launches": 0, "ms": 0.0, "bytes": 0.0})
        for f in ("launches", "ms", "bytes"):
            a[f] += r[f]
    return out


def host_cores():
    """CPU threads this process may really use: affinity mask, cgroup quota, and the 16-per-GPU share of the box."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return max(1, min(n, int(os.environ.get("ADNM_CPU_THREADS", "16"))))


def cpu_baseline(args, steps):
    """The oracle's full training step on the host CPU (kind 'port': the reference's Python cannot travel)."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import adnm_oracle as O
    from adnm_hip import recipe
    with open(os.path.join(ROOT, "tests", "golden", "state_dict_manifest.json")) as f:
        manifest = json.load(f)
    if (args.in_frames, args.out_frames) != (5, 20):
        return None
    cores = host_cores()
    torch.set_num_threads(cores)
    print(f"[bench] cpu_baseline: oracle on {cores} host threads ...", file=sys.stderr, flush=True)
    sd = recipe.state_dict_from_manifest(manifest)
    params = {k: v.clone().requires_grad_(True) for k, v in sd.items() if manifest[k]["trainable"]}
    full = dict(sd)
    full.update(params)
    opt = torch.optim.AdamW(list(params.values()), lr=1e-3, betas=(0.9, 0.999), eps=1e-9, weight_decay=1e-2)
    frames = recipe.radar_batch(args.batch, args.in_frames + args.out_frames, args.size, name="bench")
    x, tgt = frames[:, :args.in_frames], frames[:, args.in_frames:]
    times = []
    for i in range(steps + 1):
        t0 = time.perf_counter()
        out = O.vision_mamba(full, x)
        loss = O.en_rainfall_loss(out, tgt)
        loss.backward()
        torch.nn.utils.clip_grad_norm_([p for p in params.values() if p.grad is not None], 0.025)
        opt.step()
        opt.zero_grad(set_to_none=True)
        times.append(time.perf_counter() - t0)
        print(f"[bench] cpu_baseline step {i}: {times[-1]:.2f} s", file=sys.stderr, flush=True)
    t = sorted(times[1:])[len(times[1:]) // 2]
    return {"value": args.batch / t, "unit": "sequences/s", "cores": cores, "kind": "port",
            "sample": f"{steps} timed full training steps (+1 warm-up) of the oracle at B={args.batch}, {args.size}x{args.size}, fp32, "
                      f"torch.set_num_threads({cores}); median {t:.2f} s/step"}


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the product path has no CPU fallback")
    local = local % torch.cuda.device_count()  # (a 1-GPU rehearsal box runs every rank on device 0, backend gloo)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        backend = os.environ.get("ADNM_DIST_BACKEND", "nccl")  # "nccl" IS RCCL on ROCm
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    from adnm_hip import lib, recipe
    from adnm_hip.trainer import FlatTrainer
    from models.ADNMUNet import create_ADNMUNet
    from models.loss import enRainfallLoss
    lib.load()
    if args.blas != "default":
        torch.backends.cuda.preferred_blas_library(args.blas)  # plain library GEMMs: rocBLAS ("hipblas") or hipBLASLt

    model = create_ADNMUNet(args.in_frames, args.out_frames, 6, img_size=args.size)
    recipe.fill_parameters(model)  # identical replicas on every rank, same parameters as the parity fixtures
    model = model.to(dev).train()
    criterion = enRainfallLoss(omega_t=0.57, alpha=0.25, gamma=0.).to(dev)  # train_untils.py:43
    # AdamW recipe of train_untils.py:35-42; clip threshold = norm_max of the warm-up epochs (train.py:87,122-124)
    trainer = FlatTrainer(model, criterion, lr=1e-3, betas=(0.9, 0.999), eps=1e-9, weight_decay=1e-2, max_norm=0.025,
                          use_graph=bool(args.graph), stages="auto" if args.stages == "auto" else bool(int(args.stages)))
    frames = recipe.radar_batch(args.batch, args.in_frames + args.out_frames, args.size, salt=rank, name="bench").to(dev)
    x, tgt = frames[:, :args.in_frames].contiguous(), frames[:, args.in_frames:].contiguous()
    trainer.prepare(x, tgt)

    def step():
        return trainer.step(x, tgt)

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    sync()
    if rank == 0:
        print("[bench] warm-up done, timing ...", file=sys.stderr, flush=True)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    sync()
    dt = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([dt], device=dev, dtype=torch.float32)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax)
    loss_val = float(loss.detach())
    if rank == 0:
        print(f"[bench] timed region: {1e3 * dt / args.steps:.2f} ms/step", file=sys.stderr, flush=True)

    def eager_step():  # same work, launched eagerly: per-launch HIP events cannot be recorded inside a graph replay
        g = trainer.graph
        trainer.graph = None
        for p in trainer.used:
            p.grad = None
        try:
            return trainer.step(x, tgt)
        finally:
            trainer.graph = g

    # ---- instrumented steps (outside the timed region): per-kernel HIP-event timing inside libadnm_hip
    prof = {}
    if rank != 0 and args.prof_steps > 0:
        for _ in range(args.prof_steps):
            eager_step()  # every rank takes part: the step contains the gradient all-reduce
        torch.cuda.synchronize()
    if rank == 0 and args.prof_steps > 0:
        lib.query("adnm_prof_enable", 1)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(args.prof_steps):
            eager_step()
        e1.record()
        torch.cuda.synchronize()
        lib.query("adnm_prof_enable", 0)
        prof = collect_profile(lib)
        prof_step_ms = e0.elapsed_time(e1) / args.prof_steps
    if world > 1:
        dist.barrier()

    if rank == 0:
        total_ms = sum(r["ms"] for r in prof.values()) or 1.0
        if args.dump_prof:
            with open(args.dump_prof, "w") as f:
                f.write("kernel@bytes_per_launch\tlaunches_per_step\tavg_us\tms_per_step\tGB/s\n")
                for key, r in sorted(prof.items(), key=lambda kv: -kv[1]["ms"]):
                    f.write(f"{key}\t{r['launches'] / args.prof_steps:.1f}\t{1e3 * r['ms'] / r['launches']:.2f}\t{r['ms'] / args.prof_steps:.4f}\t"
                            f"{r['bytes'] / max(r['ms'], 1e-9) / 1e6:.1f}\n")
        # the dominant kernel = the (kernel, shape) instance with the largest share of the HIP-kernel time
        dom = max(prof.items(), key=lambda kv: kv[1]["ms"]) if prof else None
        roofline = None
        kernels = {}
        for name, r in sorted(by_kernel(prof).items(), key=lambda kv: -kv[1]["ms"]):
            kernels[name] = {"launches_per_step": r["launches"] / args.prof_steps, "ms_per_step": round(r["ms"] / args.prof_steps, 4),
                             "avg_us": round(1e3 * r["ms"] / r["launches"], 2),
                             "GBps": round(r["bytes"] / (r["ms"] * 1e-3) / 1e9, 1) if r["ms"] > 0 else None}
        if dom:
            name, r = dom
            ach = r["bytes"] / (r["ms"] * 1e-3) / 1e9
            roofline = {"kernel": name.split("@")[0], "shape_bytes": int(float(name.split("@")[1])), "launches_per_step": r["launches"] / args.prof_steps,
                        "bound": "hbm", "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": round(ach / HBM_PEAK_GBS, 4), "traffic": pmc_traffic(name.split("@")[0]),
                        "avg_launch_us": round(1e3 * r["ms"] / r["launches"], 2),
                        "algorithmic_bytes_per_launch": round(r["bytes"] / r["launches"]),
                        "hip_kernels_ms_per_step": round(total_ms / args.prof_steps, 3),
                        "instrumented_step_ms": round(prof_step_ms, 3)}
        res = {
            "metric": "sequences/sec training ADNM-UNet 5->20x128x128", "value": round(world * args.batch * args.steps / dt, 3),
            "unit": "sequences/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1e3 * dt / args.steps, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"ADNM-UNet create_ADNMUNet({args.in_frames},{args.out_frames},6) {args.size}x{args.size} full training step "
                                   "(fwd + enRainfallLoss + bwd + clip_grad_norm_ + AdamW), recipe parameters, synthetic radar frames in HBM",
                       "per_gpu_batch": args.batch, "global_batch": world * args.batch, "frames": f"{args.in_frames}->{args.out_frames}",
                       "parallelism": f"dp{world}" if world > 1 else "single", "launch": ("hipGraph replay" if args.graph else "eager") + (", two-stage backward with overlapped all-reduce" if trainer.staged else ""),
                       "loss": round(loss_val, 6)},
            "roofline": roofline,
        }
        if world == 1 and not args.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline(args, args.cpu_steps)
        else:
            res["cpu_baseline"] = None
        res["kernels"] = kernels
        print(json.dumps(res), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
