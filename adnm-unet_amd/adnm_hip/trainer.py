"""The training step around the hot path, MI355X-style:

    [hipGraph 0]   forward -> loss -> backward of the LAST stage (refiner)       -> its gradients land in flat_g[bucket 0]
    [RCCL]         all-reduce of bucket 0 on RCCL's stream over xGMI ............  runs beside:
    [hipGraph 1]   backward of the stage before it (decoder blocks)             -> flat_g[bucket 1]
    [RCCL]         all-reduce of bucket 1 ........................................  beside graph 2 (e2ds + fusion), and so on down to the
    [hipGraph K-1] backward of the FIRST stage (encoder1-3)                     -> the last, smallest bucket: the only exposed all-reduce
    [3 HIP launches]   global grad-norm, clip_grad_norm_ scaling, AdamW on flat p / g / m / v

replacing the reference's nn.DataParallel scatter/replicate/gather (train.py:99-102), its per-tensor
clip_grad_norm_ (train.py:140) and torch.optim.AdamW over 669 tensors (train_untils.py:35-42).  On one GPU (or with
overlap=False) there is ONE graph and one bucket.  The stage cuts come from the model (forward_stages(): SURVEY.md §8e's order
refiner -> decoder -> e2ds / fusion -> encoder4-6 -> encoder1-3; or the older two-stage forward_stage1 / forward_stage2).

  * Parameters that receive gradients are re-homed as views into one flat buffer (state_dict unchanged), so the
    optimiser is a single streaming kernel and the gradient collective is a few LARGE messages in reverse execution order
    (refiner -> decoder -> encoder) — the right shape for xGMI rings.  The 307 parameters the reference never
    gives a gradient (e2ds[3..6], att1..4, ...) are discovered by a dry-run backward and left out: they are never decayed nor
    updated, exactly like torch.optim.AdamW skipping p.grad is None.
  * fwd+bwd is captured once as hipGraphs (our kernels are enqueued through ctypes on torch's capture stream; all memory
    comes from torch's graph-private pool) and replayed.  The collectives and the optimiser stay outside the graphs, so
    learning-rate schedules and the adaptive clip threshold of train.py:122-130 remain ordinary host-side floats, and the
    collectives are plain torch.distributed calls between graph launches (nothing RCCL-specific is captured).
  * reduce_dtype="bf16": the wire format of the collective is bf16 (146 MB instead of 292 MB); sums are formed by RCCL in
    bf16, the 1/world average and the return to fp32 happen in one HIP pass.
"""
import gc

import torch
import torch.distributed as dist

from . import lib, ops


class _Both:
    """enter a then b, leave b then a"""

    def __init__(self, a, b):
        self.a, self.b = a, b

    def __enter__(self):
        self.a.__enter__()
        self.b.__enter__()

    def __exit__(self, *exc):
        self.b.__exit__(*exc)
        self.a.__exit__(*exc)
        return False


class FlatTrainer:
    def __init__(self, model, loss_fn, lr=1e-3, betas=(0.9, 0.999), eps=1e-9, weight_decay=1e-2, max_norm=0.0,
                 process_group=None, use_graph=True, fused=True, overlap="auto", reduce_dtype="f32", stages=None, defer_folds=True, side_stream=False,
                 nstages=None):
        """overlap: cut the backward at the model's stage boundaries and all-reduce every stage's gradients while the backward of the
        stages before it runs.  "auto" = whenever there is more than one rank.  `stages` is the older name of the same switch
        (True / False).  nstages: None = every stage the model offers (forward_stages(): 5 for ADNM-UNet); 2 = the two-stage cut
        (encoder | decoder + refiner)."""
        self.model, self.loss_fn = model, loss_fn
        self.lr, self.betas, self.eps, self.wd, self.max_norm = lr, betas, eps, weight_decay, max_norm
        self.group = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self.use_graph, self.fused = use_graph, fused
        if stages is not None:
            overlap = bool(stages) if stages != "auto" else "auto"
        two = hasattr(model, "forward_stage1") and hasattr(model, "forward_stage2") and hasattr(model, "stage1_parameters")
        multi = hasattr(model, "forward_stages")
        self.staged = (two or multi) and (self.world > 1 if overlap == "auto" else bool(overlap))
        self.stage_defs = None   # [(function, set of parameter ids)] in forward order
        if self.staged:
            if multi and nstages != 2:
                defs = [(fn, {id(p) for m in mods for p in m.parameters()}) for fn, mods in model.forward_stages()]
                if nstages is not None and nstages < len(defs):   # merge the FIRST stages (the last buckets) down to nstages
                    k = len(defs) - nstages + 1
                    fns, ids = [f for f, _ in defs[:k]], set().union(*[i for _, i in defs[:k]])

                    def merged(*a, _fns=fns):
                        for f in _fns:
                            a = f(*a)
                        return a
                    defs = [(merged, ids)] + defs[k:]
                self.stage_defs = defs
            else:
                first = {id(p) for p in model.stage1_parameters()}
                self.stage_defs = [(model.forward_stage1, first), (model.forward_stage2, None)]   # None: every other parameter
        assert reduce_dtype in ("f32", "bf16")
        self.reduce_dtype = reduce_dtype
        self.defer_folds = defer_folds
        # weight-gradient kernels on a second stream beside the input-gradient chain (ops.SIDE).  Off by default: measured on MI355X /
        # ROCm 7.2 the forked branches of the replayed hipGraph buy nothing and every fork costs (11.4 ms/step off, 12.1 with a fork per
        # 16 weight gradients, 13.1 with one per weight gradient); bitwise neutral either way (tests/test_trainer_gpu.py)
        self.side_stream = side_stream
        self.used = None
        self.graph = self.graph2 = None
        self.graphs = []
        self.static_loss = self._carry = self._cuts = self.sx = self.st = None
        self.flat_g = None
        self.buckets = []
        self._steps = 0

    # ------------------------------------------------------------------ one-time setup
    def _fwd_bwd(self, x, tgt):
        ops.GRADS.reset_claims(id(self))
        out = self.model(x)
        loss = self.loss_fn(out, tgt)
        with self._deferred():   # the second-stage folds of the parameter gradients: batched, flushed on the way out
            loss.backward()
        return loss

    def _deferred(self):
        """the context of a backward pass: second-stage folds of the parameter gradients batched (ops.FOLDS), weight-gradient kernels
        on a side stream beside the input-gradient chain (ops.SIDE); both are joined / flushed on the way out"""
        on = self.defer_folds and self.used is not None and self.flat_g.is_cuda
        dev = self.flat_g.device if self.used is not None else torch.device("cpu")
        return _Both(ops.SIDE.active(dev, on and self.side_stream), ops.FOLDS.active(dev, on))

    @torch.no_grad()
    def _flatten(self):
        used = [p for p in self.model.parameters() if p.requires_grad and p.grad is not None]
        # flat layout = the stages in BACKWARD order (last stage first): every stage's gradients are one contiguous range = one bucket,
        # in the order the buckets become ready.  groups[j] = parameters of the j-th bucket.
        groups = [used]
        if self.staged:
            K = len(self.stage_defs)
            claimed = set().union(*[ids for _, ids in self.stage_defs if ids is not None])
            owners = {}
            for k, (_, ids) in enumerate(self.stage_defs):
                for i in (ids or ()):
                    owners.setdefault(i, []).append(k)
            twice = [n for n, p in self.model.named_parameters() if len(owners.get(id(p), ())) > 1]
            if twice:
                raise RuntimeError(f"FlatTrainer: parameters claimed by more than one stage of forward_stages(): {twice[:8]}")
            catch_all = any(ids is None for _, ids in self.stage_defs)
            groups = []
            for k in range(K - 1, -1, -1):
                ids = self.stage_defs[k][1]
                groups.append([p for p in used if (id(p) in ids if ids is not None else id(p) not in claimed)])
            left = {id(p) for p in used} - {id(p) for g in groups for p in g}
            if left and not catch_all:
                # a parameter no stage names would be asked for in no part's backward(inputs=...): it would never get a gradient and
                # only be weight-decayed — refuse instead of training a model with silently frozen parameters
                names = [n for n, p in self.model.named_parameters() if id(p) in left]
                raise RuntimeError(f"FlatTrainer: parameters that receive gradients but belong to no stage of forward_stages(): {names[:8]}")
            used = [p for g in groups for p in g]
        self.late, self.early = groups[0], [p for g in groups[1:] for p in g]
        dev, dt = used[0].device, used[0].dtype
        # one (lo, hi) bucket PER GROUP, empty groups included (a frozen or parameter-free stage): buckets[j] belongs to backward part j
        offs, total, buckets = [], 0, []
        for g in groups:
            total = (total + 7) // 8 * 8  # a stage boundary is also a bucket boundary of the bf16 wire buffer (16-byte aligned there too)
            lo = total
            for p in g:
                offs.append(total)
                total += (p.numel() + 3) // 4 * 4  # keep every tensor 16-byte aligned inside the flat buffers
            buckets.append((lo, total))
        assert len(buckets) == len(groups) and (not self.staged or len(buckets) == len(self.stage_defs))
        self.groups = groups
        self.flat_p = torch.zeros(total, dtype=dt, device=dev)
        self.flat_g = torch.zeros(total, dtype=dt, device=dev)
        self.exp_avg = torch.zeros(total, dtype=dt, device=dev)
        self.exp_avg_sq = torch.zeros(total, dtype=dt, device=dev)
        self.state = torch.zeros(4, dtype=torch.float32, device=dev)
        self.g_views = []

        # modules name the weights their NHWC kernels want in (out, kh, kw, in) memory order (dense 3x3 convs, transposed convs)
        nhwc = {id(p) for m in self.model.modules() if hasattr(m, "adnm_nhwc_parameters") for p in m.adnm_nhwc_parameters()}

        def shaped(flat, o, p):
            """view of the flat slice with p's logical shape.  The weights of the dense NHWC convs get channels-last strides: the
            kernels then read them (and write their gradients) as they lie, with the reduction axis contiguous."""
            t = flat[o:o + p.numel()]
            if id(p) in nhwc and p.dim() == 4 and p.is_cuda:
                co, ci, kh, kw = p.shape
                return t.view(co, kh, kw, ci).permute(0, 3, 1, 2)
            return t.view_as(p)
        for p, o in zip(used, offs):
            view = shaped(self.flat_p, o, p)
            view.copy_(p.data)
            p.data = view
            self.g_views.append(shaped(self.flat_g, o, p))
            p.grad = None
        self.used, self.n = used, total
        self.offs = offs
        # gradient buckets in the order they become ready (= the order they are all-reduced); an empty stage has an empty bucket
        # (lo == hi), which _reduce_begin skips
        self.buckets = buckets
        self.n_late = self.buckets[0][1]
        self.group_ranges, lo = [], 0
        for g in groups:
            self.group_ranges.append((lo, lo + len(g)))
            lo += len(g)
        if self.world > 1 and self.reduce_dtype == "bf16":
            self.comm = torch.empty(total, dtype=torch.bfloat16, device=dev)
        # backward functions that allocate parameter gradients write them straight into these slices (ops.GRADS)
        ops.GRADS.register(id(self), {p.data_ptr(): gv for p, gv in zip(used, self.g_views)})
        if self.fused:
            self.ws = torch.empty(int(lib.query("adnm_adamw_ws_bytes")), dtype=torch.uint8, device=dev)

    def bucket_report(self):
        """[(bucket index in all-reduce order, first element, last element + 1, bytes on the wire)]: what each stage's collective moves —
        the sizes to price against the xGMI ring."""
        eb = 2 if self.reduce_dtype == "bf16" else 4
        return [(j, lo, hi, (hi - lo) * eb) for j, (lo, hi) in enumerate(self.buckets)]

    def _setup_shadows(self, mode):
        """The narrow SHADOW of the flat parameter buffer (ops.ShadowRegistry; include/adnm_hip.h: adnm_adamw_step): mode 1 = bf16, 2 = per-tensor
        scaled e4m3 (weight records in ops.QUANT's table, one per matrix-shaped parameter), rewritten by every optimiser pass.  The
        weight-streaming GEMMs read it instead of the fp32 values.  mode 0 (exact fp32, CPU, ADNM_NARROW_WEIGHTS=0): none."""
        self.shadow, self.shadow_mode = None, 0
        import os
        if mode == 0 or not self.fused or not self.flat_p.is_cuda or os.environ.get("ADNM_NARROW_WEIGHTS", "1") == "0":
            return
        dev = self.flat_p.device
        self.shadow_mode = mode
        self.shadow = torch.zeros(self.n, dtype=torch.bfloat16 if mode == 1 else torch.uint8, device=dev)
        rows = []
        for p in self.used:   # fp8: a record per GEMM-shaped weight (every other tensor: no scale, its shadow bytes are never read)
            rows.append(ops.QUANT.weight_row(dev, p.data_ptr()) if (mode == 2 and p.dim() >= 2 and p.numel() >= 1024) else -1)
        ends = [o // 4 for o in self.offs[1:]] + [self.n // 4]
        self.seg_end = torch.tensor(ends, dtype=torch.int32, device=dev)
        self.seg_rec = torch.tensor(rows, dtype=torch.int32, device=dev)
        ops.SHADOWS.register(id(self), self.flat_p, self.shadow, mode, self.offs, rows)

    def refresh_shadows(self, collect_only=False):
        """rewrite the shadow from the parameters as they are (after they moved into the flat buffer; after a checkpoint was loaded into a
        prepared trainer).  collect_only (fp8): only gather max |w| per weight record — the first calibration."""
        if not getattr(self, "shadow_mode", 0):
            return
        dev = self.flat_p.device
        tab = ops.QUANT.table_ptr(dev) if self.shadow_mode == 2 else None
        lib.call("adnm_shadow_refresh", self.flat_p.data_ptr(), self.n, None if collect_only else self.shadow.data_ptr(), self.shadow_mode,
                 self.seg_end.data_ptr(), self.seg_rec.data_ptr(), self.seg_end.numel(), tab, int(collect_only or self.shadow_mode == 2),
                 torch.cuda.current_stream().cuda_stream)
        if not collect_only:
            ops.SHADOWS.set_valid(id(self), True)

    def close(self):
        """Give back everything this trainer owns on the device, in a fixed order, NOW (not whenever the cyclic collector gets to
        it): the captured graphs first (they reference graph-pool memory), then the tensors that live in that pool (the static loss
        with its autograd graph, the stage hand-over, the static inputs), then the gradient-destination registration.  Idempotent;
        __del__ calls it.  A round-2 run aborted (rc 134) because dead trainers of failed tests — kept alive by their tracebacks —
        were finalised by the cyclic GC in the middle of a NEW trainer's warm-up / stream capture: destroying a hipGraph (and
        freeing its private pool) while another capture is in flight on the device is not allowed by the runtime.  prepare()
        therefore also collects BEFORE it starts and keeps the collector off until both captures have ended."""
        self.tail = None
        self.graphs = []
        self.graph2 = None
        self.graph = None
        self.static_loss = None
        self._carry = self._cuts = None
        self.sx = self.st = None
        self._splitws = None   # (after the graphs: their launches point into it)
        if getattr(self, "_quant_pinned", False):
            self._quant_pinned = False
            try:
                ops.QUANT.unpin(self.flat_g.device)
            except Exception:
                pass
        try:
            ops.SHADOWS.drop(id(self))
        except Exception:
            pass
        try:
            ops.GRADS.drop(id(self))   # lock-free for a finaliser: queued, drained by the next register / take
        except Exception:
            pass

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _gather(self, lo=0, hi=None):
        """Copy the gradients of used[lo:hi] that were not born inside the flat buffer (same data_ptr = already in place)."""
        hi = len(self.used) if hi is None else hi
        # a slice that ONE backward node claimed holds that gradient itself (whatever tensor object autograd kept for p.grad)
        born = ops.GRADS.born_in_place(id(self))
        pairs = [(gv, p.grad) for gv, p in zip(self.g_views[lo:hi], self.used[lo:hi])
                 if p.grad is not None and p.grad.data_ptr() != gv.data_ptr() and p.data_ptr() not in born]
        self.gathered = [i for i, (gv, p) in enumerate(zip(self.g_views[lo:hi], self.used[lo:hi]), lo)
                         if p.grad is not None and p.grad.data_ptr() != gv.data_ptr() and p.data_ptr() not in born]   # (tools/gather_report.py)
        if pairs:
            torch._foreach_copy_([d for d, _ in pairs], [s for _, s in pairs])

    # staged backward.  part 0 = the whole forward + the backward of the LAST stage (down to its input cut); part j = the backward of
    # stage K-1-j, started from the gradients the part before left at its output cut.
    def _stage_part0(self, x, tgt):
        ops.GRADS.reset_claims(id(self))
        K = len(self.stage_defs)
        self._cuts = [None] * K   # stage k >= 1: (tensors the stage before produced, their detached twins)
        args = (x,)
        for k, (fn, _) in enumerate(self.stage_defs):
            if k > 0:
                # a true cut: the stage runs on detached twins, so a later stage's backward stops there (a skip tensor also reaches the
                # loss THROUGH later layers of its own stage; that path belongs to the earlier part, which starts from the originals
                # with the twins' gradients)
                twins, origs, targs = {}, [], []
                for t in args:
                    if torch.is_tensor(t) and t.requires_grad:
                        if id(t) not in twins:   # a tensor handed over twice gets one twin: its gradient is the sum over both uses
                            twins[id(t)] = t.detach().requires_grad_(True)
                            origs.append(t)
                        targs.append(twins[id(t)])
                    else:
                        targs.append(t)
                self._cuts[k] = (origs, [twins[id(t)] for t in origs])
                args = tuple(targs)
            args = fn(*args)
            args = args if isinstance(args, tuple) else (args,)
        loss = self.loss_fn(args[0], tgt)
        self._backward_stage(K - 1, [loss], None)
        return loss

    def _backward_stage(self, k, roots, grads):
        """backward of stage k from `roots` (the loss, or the stage's outputs with the gradients carried over the cut) down to its own
        parameters and the twins at its input cut; leaves the twins' gradients as the next part's carry"""
        twins = self._cuts[k][1] if k > 0 else []
        # backward(inputs=...) accumulates through the ordinary AccumulateGrad path (which keeps a fresh gradient without copying it;
        # autograd.grad() was measured to cost ~290 extra device copies per step here)
        for t in twins:
            t.grad = None
        j = len(self.stage_defs) - 1 - k
        wanted = self.groups[j] + twins
        if wanted and roots:   # (a frozen FIRST stage has neither parameters nor an input cut: nothing to differentiate)
            with self._deferred():
                torch.autograd.backward(roots, grad_tensors=grads, inputs=wanted)
        self._carry = [(o, tw.grad) for o, tw in zip(self._cuts[k][0], twins) if tw.grad is not None] if k > 0 else []
        self._gather(*self.group_ranges[j])

    def _stage_part(self, j):
        k = len(self.stage_defs) - 1 - j
        self._backward_stage(k, [t for t, _ in self._carry], [g for _, g in self._carry])

    def prepare(self, x, tgt):
        """Dry-run backward (finds the parameters that receive gradients), flatten, and capture the graph(s).
        Ownership: objects with device-side destructors that died earlier (old trainers, graphs of failed runs) are collected
        BEFORE anything starts, and the cyclic collector stays off until the last capture has ended (see close()); an exception
        on the way — inside a capture included — unbinds the fold queue, drops the half-built graphs and re-raises."""
        gc.collect()
        gc_was_on = gc.isenabled()
        gc.disable()
        try:
            self._prepare(x, tgt)
        except BaseException:
            self._abort_prepare()
            raise
        finally:
            if gc_was_on:
                gc.enable()

    def _abort_prepare(self):
        """After an exception in prepare(): torch.cuda.graph's __exit__ has already ended a capture in flight; what is left is
        OUR state — the thread's fold-queue binding (a kernel wrapper may have died between bind and unbind), the deferral /
        side-stream switches of this device, and the graph objects, which must not survive half-captured."""
        try:
            lib.load().adnm_foldq_bind(None)
        except Exception:
            pass
        dev = self.flat_g.device if getattr(self, "flat_g", None) is not None else None
        if dev is not None and dev.type == "cuda":
            for reg in (ops.FOLDS, ops.SIDE):
                try:
                    reg.abort(dev)
                except Exception:
                    pass
        self.graphs = []
        self.graph2 = None
        self.graph = None
        self.static_loss = None
        self._carry = self._cuts = None
        self._splitws = None

    def _prepare(self, x, tgt):
        self.model.zero_grad(set_to_none=True)
        # fp8 (BASELINE config 5): the dry run and the calibration step run on bf16 operands; the calibration step — after the
        # parameters have moved into the flat buffer, so that the call-site keys are the final addresses — collects every GEMM
        # operand's amax, from which adnm_quant_update makes the first scales.  From then on the steps run on fp8 operands and
        # re-calibrate themselves every ops.QUANT.period steps (delayed per-tensor scaling, all on the device: graph-replayable).
        self.fp8 = ops.mfma_precision() == "fp8" and x.is_cuda
        if self.fp8:
            ops.set_mfma_precision("bf16")
        try:
            self._fwd_bwd(x, tgt)
            self._flatten()
            if self.fp8:
                ops.QUANT.reset(x.device)
                self._setup_shadows(2)                        # weight records (after the reset: it may forget the keys)
                self.refresh_shadows(collect_only=True)      # max |w| of every GEMM weight
                ops.QUANT.calibrating = True
                for p in self.used:
                    p.grad = None
                self._run_eager(x, tgt)                       # bf16 operands, fp32 weights: every call site's activation / gradient maxima
                ops.QUANT.calibrating = False
                ops.QUANT.update(x.device)                    # -> the first scales, weights included
                self.refresh_shadows()                        # the e4m3 shadow, written with them
                for p in self.used:
                    p.grad = None
            else:
                self._setup_shadows(1 if (x.is_cuda and ops.mfma_precision() == "bf16") else 0)
                self.refresh_shadows()
        finally:
            ops.QUANT.calibrating = False
            if self.fp8:
                ops.set_mfma_precision("fp8")
        if not self.use_graph:
            return
        self.sx, self.st = x.clone(), tgt.clone()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(2):
                for p in self.used:
                    p.grad = None
                self._run_eager(self.sx, self.st)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        for p in self.used:
            p.grad = None
        self.graph = torch.cuda.CUDAGraph()
        # what the graphs hold besides torch's pool: the uncached [arrival counters | slabs] region of their split GEMM launches (one
        # scope for all of this trainer's graphs: they replay one after the other on one stream) and, in the fp8 configuration, pointers
        # into the device's quantisation table (pinned: a later calibration re-uses the rows instead of re-assigning them)
        self._splitws = ops.SPLITWS.open_scope(x.device)
        if self.fp8 and not getattr(self, "_quant_pinned", False):
            ops.QUANT.pin(x.device)
            self._quant_pinned = True
        # thread_local: RCCL's watchdog thread may query events while we capture; only this thread's calls are checked
        with ops.SPLITWS.capturing(self._splitws):
            if not self.staged:
                with torch.cuda.graph(self.graph, capture_error_mode="thread_local"):
                    self.static_loss = self._fwd_bwd(self.sx, self.st)
                    self._gather()
            else:
                # each stage graph ends with the wire cast of ITS bucket (bf16 wire): a step at N > 1 is then K graph replays + K collective
                # calls + the tail graph below, nothing else is launched from the host
                with torch.cuda.graph(self.graph, capture_error_mode="thread_local"):
                    self.static_loss = self._stage_part0(self.sx, self.st)
                    self._wire_cast(0)
                self.graphs = []   # same memory pool: every part reads what the parts before saved for it
                for j in range(1, len(self.stage_defs)):
                    g = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(g, pool=self.graph.pool(), capture_error_mode="thread_local"):
                        self._stage_part(j)
                        self._wire_cast(j)
                    self.graphs.append(g)
                self.graph2 = self.graphs[0] if self.graphs else None
            # the tail of a step: (bf16 wire: averages back to fp32) -> fp8 scale update -> clip + AdamW (+ shadow), with the learning rate and
            # the clip threshold in device memory (self.hyper) so that the captured launches follow the host's schedules
            self.tail = None
            if self.fused and self.staged:
                self.hyper = torch.tensor([self.lr, self.max_norm], dtype=torch.float32, device=x.device)
                self._hyper_host = (float(self.lr), float(self.max_norm))
                keep = (self.flat_p.clone(), self.exp_avg.clone(), self.exp_avg_sq.clone(), self.state.clone(), self.flat_g.clone(),
                        self.shadow.clone() if getattr(self, "shadow", None) is not None else None)
                qsave = ops.QUANT.snapshot(x.device) if self.fp8 else None
                side2 = torch.cuda.Stream()
                side2.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(side2):
                    self._tail_body()   # warm-up outside capture
                torch.cuda.current_stream().wait_stream(side2)
                self.tail = torch.cuda.CUDAGraph()
                with torch.cuda.graph(self.tail, pool=self.graph.pool(), capture_error_mode="thread_local"):
                    self._tail_body()
                # the warm-up advanced the optimiser once: put everything back
                for dst, src in zip((self.flat_p, self.exp_avg, self.exp_avg_sq, self.state, self.flat_g, getattr(self, "shadow", None)), keep):
                    if src is not None:
                        dst.copy_(src)
                if qsave is not None:
                    ops.QUANT.restore(x.device, qsave)

    def _wire_cast(self, j):
        """fp32 -> bf16 copy of bucket j into the wire buffer (captured at the end of stage graph j)"""
        if self.world > 1 and self.reduce_dtype == "bf16":
            lo, hi = self.buckets[j]
            if hi > lo:
                self._cast(self.flat_g[lo:hi], self.comm[lo:hi])

    def _tail_body(self):
        if self.world > 1 and self.reduce_dtype == "bf16":
            for lo, hi in self.buckets:
                if hi > lo:
                    self._cast(self.comm[lo:hi], self.flat_g[lo:hi], 1.0 / self.world)
        if getattr(self, "fp8", False):
            ops.QUANT.update(self.flat_g.device)
        self._optimizer_step(hyper=True)

    def _run_eager(self, x, tgt, between=None):
        """between(j): called after part j (its bucket is complete) while parts remain — the N > 1 flow starts the bucket's all-reduce there"""
        if not self.staged:
            loss = self._fwd_bwd(x, tgt)
            self._gather()
            return loss
        loss = self._stage_part0(x, tgt)
        for j in range(1, len(self.stage_defs)):
            if between is not None:
                between(j - 1)
            self._stage_part(j)
        return loss

    # ------------------------------------------------------------------ the collective
    def _cast(self, src, dst, scale=1.0):
        """fp32 <-> bf16 copy of a gradient range (HIP pass on the GPU; torch on the CPU test path)."""
        if self.fused and src.is_cuda:
            name = "adnm_cast_f32_bf16" if src.dtype == torch.float32 else "adnm_cast_bf16_f32"
            lib.call(name, src.data_ptr(), dst.data_ptr(), src.numel(), float(scale), torch.cuda.current_stream().cuda_stream)
        else:
            dst.copy_(src.to(dst.dtype) if scale == 1.0 else (src.float() * scale).to(dst.dtype))

    def _reduce_begin(self, lo, hi, pending, cast_done=False):
        """Start averaging flat_g[lo:hi] over the ranks.  RCCL: asynchronously on its own stream (it first waits for what the
        compute stream has enqueued so far, i.e. the graph that produced the range), so the next graph runs beside the ring."""
        if self.world == 1 or hi <= lo:
            return
        nccl = dist.get_backend(self.group) == "nccl"
        if self.reduce_dtype == "bf16":
            t = self.comm[lo:hi]
            if not cast_done:   # (the stage graphs end with their bucket's cast)
                self._cast(self.flat_g[lo:hi], t)
            op = dist.ReduceOp.SUM
        else:
            t = self.flat_g[lo:hi]
            op = dist.ReduceOp.AVG if nccl else dist.ReduceOp.SUM
        work = dist.all_reduce(t, op=op, group=self.group, async_op=True)
        pending.append((work, lo, hi, nccl))

    def _reduce_end(self, pending, tail=False):
        """tail: the casts back / the division are part of the captured tail graph"""
        for work, lo, hi, nccl in pending:
            work.wait()   # the compute stream waits for the ring; the host does not block (RCCL)
            if tail and self.reduce_dtype == "bf16":
                continue
            if self.reduce_dtype == "bf16":
                self._cast(self.comm[lo:hi], self.flat_g[lo:hi], 1.0 / self.world)
            elif not nccl:
                self.flat_g[lo:hi].div_(self.world)

    # ------------------------------------------------------------------ per step
    def step(self, x, tgt, eager=False):
        """One training step.  eager=True launches the same work without the captured graphs (bench.py's instrumented steps:
        per-launch HIP events cannot be recorded inside a graph replay)."""
        if self.used is None:
            self.prepare(x, tgt)
        pending = []
        nb = len(self.buckets)
        graphed = False
        if self.graph is not None and not eager:
            if x.data_ptr() != self.sx.data_ptr():
                self.sx.copy_(x, non_blocking=True)
                self.st.copy_(tgt, non_blocking=True)
            self.graph.replay()
            self._reduce_begin(*self.buckets[0], pending, cast_done=self.staged)
            if self.staged:
                for j, g in enumerate(self.graphs, start=1):
                    g.replay()
                    if j < nb:
                        self._reduce_begin(*self.buckets[j], pending, cast_done=True)
            loss = self.static_loss
            graphed = True
        else:
            for p in self.used:
                p.grad = None
            if self.staged:
                loss = self._run_eager(x, tgt, between=lambda j: self._reduce_begin(*self.buckets[j], pending))
                self._reduce_begin(*self.buckets[nb - 1], pending)
            else:
                loss = self._run_eager(x, tgt)
                self._reduce_begin(*self.buckets[0], pending)
            for p in self.used:
                p.grad = None
        tail = graphed and getattr(self, "tail", None) is not None and (self.world == 1 or self.reduce_dtype == "bf16" or dist.get_backend(self.group) == "nccl")
        self._reduce_end(pending, tail=tail)
        if tail:
            if (float(self.lr), float(self.max_norm)) != self._hyper_host:   # a schedule moved them: two floats to the device
                self._hyper_host = (float(self.lr), float(self.max_norm))
                self.hyper.copy_(torch.tensor(self._hyper_host, dtype=torch.float32), non_blocking=True)
            self.tail.replay()
        else:
            if getattr(self, "fp8", False):
                # one launch: amax -> scales on calibration steps, the next step's record flags.  BEFORE the optimiser pass: that pass writes
                # the e4m3 shadow of the updated weights with the scales the next step's GEMMs will read
                ops.QUANT.update(self.flat_g.device)
            self._optimizer_step()
        self._steps += 1
        return loss

    def _optimizer_step(self, hyper=False):
        if self.fused:
            mode = getattr(self, "shadow_mode", 0)
            sh = (self.shadow.data_ptr(), mode, self.seg_end.data_ptr(), self.seg_rec.data_ptr(), self.seg_end.numel(),
                  ops.QUANT.table_ptr(self.flat_p.device) if mode == 2 else None) if mode else (None, 0, None, None, 0, None)
            lib.call("adnm_adamw_step", self.flat_p.data_ptr(), self.flat_g.data_ptr(), self.exp_avg.data_ptr(),
                     self.exp_avg_sq.data_ptr(), self.n, self.state.data_ptr(), float(self.lr), float(self.betas[0]), float(self.betas[1]),
                     float(self.eps), float(self.wd), float(self.max_norm), self.ws.data_ptr(), self.ws.numel(), *sh,
                     self.hyper.data_ptr() if hyper else None, torch.cuda.current_stream().cuda_stream)
        else:
            self._torch_adamw_for_tests()

    @torch.no_grad()
    def _torch_adamw_for_tests(self):
        """Same arithmetic in torch ops — exists only so the CPU (gloo) tests can exercise the N>1 logic and the
        GPU test has an independent statement of the fused kernel's update rule.  bench.py never takes it."""
        g = self.flat_g
        if self.max_norm > 0:
            g = g * torch.clamp(self.max_norm / (g.norm() + 1e-6), max=1.0)
        step = self._steps + 1
        b1, b2 = self.betas
        self.flat_p.mul_(1 - self.lr * self.wd)
        self.exp_avg.lerp_(g, 1 - b1)
        self.exp_avg_sq.mul_(b2).addcmul_(g, g, value=1 - b2)
        denom = (self.exp_avg_sq.sqrt() / (1 - b2 ** step) ** 0.5).add_(self.eps)
        self.flat_p.addcdiv_(self.exp_avg, denom, value=-self.lr / (1 - b1 ** step))

    def grad_norm(self):
        """Pre-clip total gradient norm of the last step (device scalar; train.py:141 reads it with .item())."""
        return self.state[1].sqrt() if self.fused else self.flat_g.norm()
