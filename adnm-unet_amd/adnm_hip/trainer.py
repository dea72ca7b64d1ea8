"""The training step around the hot path, MI355X-style:

    [hipGraph replay]  forward -> loss -> backward -> gather gradients into ONE flat fp32 buffer
    [RCCL]             all-reduce(AVG) of that flat buffer over xGMI            (world > 1 only)
    [3 HIP launches]   global grad-norm, clip_grad_norm_ scaling, AdamW on flat p / g / m / v

replacing the reference's nn.DataParallel scatter/replicate/gather (train.py:99-102), its per-tensor
clip_grad_norm_ (train.py:140) and torch.optim.AdamW over 669 tensors (train_untils.py:35-42).

  * Parameters that receive gradients are re-homed as views into one flat buffer (state_dict unchanged), so the
    optimiser is a single streaming kernel and the gradient collective is ONE large message — the right shape for
    xGMI rings (few, large transfers).  The 307 parameters the reference never gives a gradient (e2ds[3..6], att1..4,
    ...) are discovered by a dry-run backward and left out: they are never decayed nor updated, exactly like
    torch.optim.AdamW skipping p.grad is None.
  * The whole fwd+bwd is captured once as a hipGraph (our kernels are enqueued through ctypes on torch's capture
    stream; all memory comes from torch's graph-private pool) and replayed: the ~3.5k launches of a step no longer
    cost host time.  The collective and the optimiser stay outside the graph, so learning-rate schedules and the
    adaptive clip threshold of train.py:122-130 remain ordinary host-side floats.
"""
import os

import torch
import torch.distributed as dist

from . import lib, ops


class FlatTrainer:
    def __init__(self, model, loss_fn, lr=1e-3, betas=(0.9, 0.999), eps=1e-9, weight_decay=1e-2, max_norm=0.0,
                 process_group=None, use_graph=True, fused=True, stages="auto"):
        """stages: two-stage backward cut at model.forward_stage1 / forward_stage2 (ADNM-UNet: encoder | decoder + refiner), so that
        the second stage's gradients are all-reduced while the first stage's backward still runs.  True / False force it; "auto"
        reads ADNM_STAGES (default off).  Measured on one MI355X: the two captured graphs cost 14.0 ms of GPU time against 12.95 ms
        for the single graph (stalls inside the second replay), while the encoder's backward — the window that hides the ring — is
        only 2.4 ms long, so the cut pays at 2 ranks (xGMI: one link, ~3.7 ms ring) and not at 8 (~0.9 ms); hence opt-in."""
        self.model, self.loss_fn = model, loss_fn
        self.lr, self.betas, self.eps, self.wd, self.max_norm = lr, betas, eps, weight_decay, max_norm
        self.group = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self.use_graph, self.fused = use_graph, fused
        can_stage = hasattr(model, "forward_stage1") and hasattr(model, "forward_stage2") and hasattr(model, "stage1_parameters")
        self.staged = can_stage and (os.environ.get("ADNM_STAGES", "0") == "1" if stages == "auto" else bool(stages))
        self.used = None
        self.graph = self.graph2 = None
        self._steps = 0

    # ------------------------------------------------------------------ one-time setup
    def _fwd_bwd(self, x, tgt):
        ops.grad_claims_reset()
        out = self.model(x)
        loss = self.loss_fn(out, tgt)
        loss.backward()
        return loss

    @torch.no_grad()
    def _flatten(self):
        used = [p for p in self.model.parameters() if p.requires_grad and p.grad is not None]
        if self.staged:   # flat layout [stage-2 (late) parameters | stage-1 (early) parameters]: each stage's gradients are one range
            first = {id(p) for p in self.model.stage1_parameters()}
            self.late = [p for p in used if id(p) not in first]
            self.early = [p for p in used if id(p) in first]
            used = self.late + self.early
        dev, dt = used[0].device, used[0].dtype
        offs, total = [], 0
        for p in used:
            offs.append(total)
            total += (p.numel() + 3) // 4 * 4  # keep every tensor 16-byte aligned inside the flat buffers
        self.flat_p = torch.zeros(total, dtype=dt, device=dev)
        self.flat_g = torch.zeros(total, dtype=dt, device=dev)
        self.exp_avg = torch.zeros(total, dtype=dt, device=dev)
        self.exp_avg_sq = torch.zeros(total, dtype=dt, device=dev)
        self.state = torch.zeros(4, dtype=torch.float32, device=dev)
        self.g_views = []
        def shaped(flat, o, p):
            """view of the flat slice with p's logical shape.  Dense conv weights (Cout, Cin>1, kh, kw) get channels-last strides:
            MIOpen's NHWC kernels then read them (and write their gradients) as they lie, instead of re-laying them out in a
            copy kernel on every call (13 convs x 3 passes per step)."""
            t = flat[o:o + p.numel()]
            if p.dim() == 4 and p.shape[1] > 1 and p.is_cuda:
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
        self.n_late = offs[len(self.late)] if self.staged and self.early else total
        # backward functions that allocate parameter gradients write them straight into these slices (ops.grad_dst)
        ops.GRAD_DST.clear()
        ops.GRAD_DST.update({p.data_ptr(): gv for p, gv in zip(used, self.g_views)})
        ops.GRAD_DST_OWNER[0] = id(self)
        if self.fused:
            self.ws = torch.empty(int(lib.query("adnm_adamw_ws_bytes")), dtype=torch.uint8, device=dev)

    def __del__(self):
        try:
            if ops.GRAD_DST_OWNER[0] == id(self):   # do not leave destinations of a dead trainer behind
                ops.GRAD_DST.clear()
        except Exception:
            pass

    def _gather(self, lo=0, hi=None):
        """Copy the gradients of used[lo:hi] that were not born inside the flat buffer (same data_ptr = already in place)."""
        hi = len(self.used) if hi is None else hi
        pairs = [(gv, p.grad) for gv, p in zip(self.g_views[lo:hi], self.used[lo:hi]) if p.grad.data_ptr() != gv.data_ptr()]
        if pairs:
            torch._foreach_copy_([d for d, _ in pairs], [s for _, s in pairs])

    # two-stage backward: A = forward + backward of stage 2 (down to the cut), B = backward of stage 1
    def _stage_a(self, x, tgt):
        ops.grad_claims_reset()
        cut = self.model.forward_stage1(x)
        # a true cut: stage 2 runs on detached twins, so stage A's backward stops there (a skip tensor also reaches the loss THROUGH the
        # rest of the encoder; that path belongs to stage B, which starts from the originals with the twins' gradients)
        twins, origs, args = {}, [], []
        for t in cut:
            if torch.is_tensor(t) and t.requires_grad:
                if id(t) not in twins:   # a tensor handed over twice gets one twin, so its gradient is the sum over both uses
                    twins[id(t)] = t.detach().requires_grad_(True)
                    origs.append(t)
                args.append(twins[id(t)])
            else:
                args.append(t)
        loss = self.loss_fn(self.model.forward_stage2(*args), tgt)
        uniq = [twins[id(t)] for t in origs]
        # backward(inputs=...) accumulates through the ordinary AccumulateGrad path (which keeps a fresh gradient without copying it;
        # autograd.grad() was measured to cost ~290 extra device copies per step here)
        for t in uniq:
            t.grad = None
        torch.autograd.backward(loss, inputs=self.late + uniq)
        self._carry = [(t, tw.grad) for t, tw in zip(origs, uniq) if tw.grad is not None]
        self._gather(0, len(self.late))
        return loss

    def _stage_b(self):
        ts, gs = [t for t, _ in self._carry], [g for _, g in self._carry]
        torch.autograd.backward(ts, grad_tensors=gs, inputs=self.early)
        self._gather(len(self.late), len(self.used))

    def prepare(self, x, tgt):
        """Dry-run backward (finds the parameters that receive gradients), flatten, and capture the graph."""
        self.model.zero_grad(set_to_none=True)
        self._fwd_bwd(x, tgt)
        self._flatten()
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
        # thread_local: RCCL's watchdog thread may query events while we capture; only this thread's calls are checked
        if not self.staged:
            with torch.cuda.graph(self.graph, capture_error_mode="thread_local"):
                self.static_loss = self._fwd_bwd(self.sx, self.st)
                self._gather()
        else:
            with torch.cuda.graph(self.graph, capture_error_mode="thread_local"):
                self.static_loss = self._stage_a(self.sx, self.st)
            self.graph2 = torch.cuda.CUDAGraph()   # same memory pool: stage B reads what stage A saved for it
            with torch.cuda.graph(self.graph2, pool=self.graph.pool(), capture_error_mode="thread_local"):
                self._stage_b()

    def _run_eager(self, x, tgt, between=None):
        if not self.staged:
            loss = self._fwd_bwd(x, tgt)
            self._gather()
            return loss
        loss = self._stage_a(x, tgt)
        if between is not None:
            between()
        self._stage_b()
        return loss

    # ------------------------------------------------------------------ per step
    def step(self, x, tgt):
        if self.used is None:
            self.prepare(x, tgt)
        nccl = self.world > 1 and dist.get_backend(self.group) == "nccl"
        works = []

        def reduce(lo, hi):
            """average flat_g[lo:hi] over the ranks; RCCL: asynchronously on its own stream (it first waits for what this stream has
            enqueued so far, i.e. the stage that produced the range), so the next stage's kernels run beside the ring"""
            if self.world == 1 or hi <= lo:
                return
            t = self.flat_g[lo:hi]
            if nccl:
                works.append(dist.all_reduce(t, op=dist.ReduceOp.AVG, group=self.group, async_op=True))
            else:
                dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
                t.div_(self.world)

        if self.graph is not None:
            if x.data_ptr() != self.sx.data_ptr():
                self.sx.copy_(x, non_blocking=True)
                self.st.copy_(tgt, non_blocking=True)
            self.graph.replay()
            if self.staged:
                reduce(0, self.n_late)
                self.graph2.replay()
                reduce(self.n_late, self.n)
            else:
                reduce(0, self.n)
            loss = self.static_loss
        else:
            if self.staged:
                loss = self._run_eager(x, tgt, between=lambda: reduce(0, self.n_late))
                reduce(self.n_late, self.n)
            else:
                loss = self._run_eager(x, tgt)
                reduce(0, self.n)
            for p in self.used:
                p.grad = None
        for w in works:
            w.wait()   # the compute stream waits for the rings; the host does not block
        self._optimizer_step()
        self._steps += 1
        return loss

    def _optimizer_step(self):
        if self.fused:
            lib.call("adnm_adamw_step", self.flat_p.data_ptr(), self.flat_g.data_ptr(), self.exp_avg.data_ptr(),
                     self.exp_avg_sq.data_ptr(), self.n, self.state.data_ptr(), float(self.lr), float(self.betas[0]), float(self.betas[1]),
                     float(self.eps), float(self.wd), float(self.max_norm), self.ws.data_ptr(), self.ws.numel(),
                     torch.cuda.current_stream().cuda_stream)
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
