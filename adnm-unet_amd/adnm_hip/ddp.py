"""One-process-per-GPU data parallelism for the ADNM-UNet step: flat gradient buckets all-reduced by RCCL
(torch.distributed backend "nccl" on ROCm) over xGMI, launched from post-accumulate-grad hooks so the
collectives overlap the rest of backward.  Replaces the reference's single-process nn.DataParallel
(train.py:99-102: per-step parameter broadcast + gradient reduce to device 0, and a shared-dict race).

Design points:
  * buckets are laid out in reverse registration order (~ reverse execution order: refiner first, encoder last),
    so the first buckets complete early in backward; sized for xGMI rings (few, large messages);
  * the set of parameters that never receive a gradient is static for this model (307 tensors): it is
    discovered on the first backward and those parameters are left with grad=None — they are never
    zero-filled, so AdamW keeps skipping them exactly as in the reference (no spurious weight decay);
  * gradients live as views into the flat bucket: the all-reduce result IS p.grad, no unflatten copy;
  * finalize() (call before clip_grad_norm_ / optimizer.step) waits for the collectives; averaging uses
    ReduceOp.AVG on RCCL, sum + scale on gloo.
"""
import os

import torch
import torch.distributed as dist
from torch.autograd import Variable


class GradBuckets:
    def __init__(self, module, process_group=None, bucket_mb=64.0, auto_finalize=False):
        self.group = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self.params = [p for p in module.parameters() if p.requires_grad]
        self.bucket_bytes = int(bucket_mb * (1 << 20))
        self.ready = False
        self.buckets = []      # dicts: flat, views{param:view}, pending, work
        self.where = {}        # param -> (bucket index)
        self._hooks = []
        self._avg = None
        # auto_finalize: finalize() runs by itself at the end of every backward pass (an autograd engine callback queued from the first
        # gradient hook of the pass, the mechanism torch's DistributedDataParallel uses), so an UNMODIFIED training loop — the
        # reference's train.py:136-145: backward, clip_grad_norm_, optimizer.step — sees averaged gradients when backward returns
        self.auto_finalize = auto_finalize
        self._cb_queued = False
        if auto_finalize:
            self._first_hooks = [p.register_post_accumulate_grad_hook(self._queue_finalize) for p in self.params]

    # ---------------------------------------------------------------- construction (after the first backward)
    def _build(self):
        used = [p for p in reversed(self.params) if p.grad is not None]
        cur, cur_bytes = [], 0
        groups = []
        for p in used:
            nb = p.numel() * p.element_size()
            if cur and cur_bytes + nb > self.bucket_bytes:
                groups.append(cur)
                cur, cur_bytes = [], 0
            cur.append(p)
            cur_bytes += nb
        if cur:
            groups.append(cur)
        for bi, ps in enumerate(groups):
            total = sum((p.numel() + 3) // 4 * 4 for p in ps)
            flat = torch.zeros(total, dtype=ps[0].dtype, device=ps[0].device)
            views, off = {}, 0
            for p in ps:
                v = flat[off:off + p.numel()].view_as(p)
                v.copy_(p.grad)
                p.grad = v
                views[p] = v
                self.where[p] = bi
                off += (p.numel() + 3) // 4 * 4
            self.buckets.append({"flat": flat, "views": views, "pending": len(ps), "n": len(ps), "work": None})
        for p in used:
            self._hooks.append(p.register_post_accumulate_grad_hook(self._on_grad))
        backend = dist.get_backend(self.group) if dist.is_initialized() else None
        self._avg = backend == "nccl"
        self.ready = True

    def _launch(self, b):
        if self.world > 1:
            op = dist.ReduceOp.AVG if self._avg else dist.ReduceOp.SUM
            b["work"] = dist.all_reduce(b["flat"], op=op, group=self.group, async_op=True)

    def _queue_finalize(self, p=None):
        if self.auto_finalize and not self._cb_queued:
            self._cb_queued = True
            Variable._execution_engine.queue_callback(self.finalize)

    def _on_grad(self, p):
        self._queue_finalize()
        b = self.buckets[self.where[p]]
        v = b["views"][p]
        if p.grad.data_ptr() != v.data_ptr():  # optimizer.zero_grad(set_to_none=True) dropped the view
            v.copy_(p.grad)
            p.grad = v
        b["pending"] -= 1
        if b["pending"] == 0:
            self._launch(b)

    # ---------------------------------------------------------------- per step
    def finalize(self):
        """Call after loss.backward() (or let auto_finalize do it): returns when every gradient holds the rank average."""
        self._cb_queued = False
        if not self.ready:
            self._build()
            for b in self.buckets:
                self._launch(b)
        for b in self.buckets:
            if b["pending"] != 0 and b["work"] is None and self.world > 1:
                raise RuntimeError("a bucket did not fill: the set of parameters receiving gradients changed between steps")
            if b["work"] is not None:
                b["work"].wait()
                if not self._avg:
                    b["flat"].div_(self.world)
                b["work"] = None
            b["pending"] = b["n"]

    def zero_grad(self):
        """Keeps the flat layout (cheaper than optimizer.zero_grad(): one memset per bucket)."""
        if not self.ready:
            for p in self.params:
                p.grad = None
            return
        for b in self.buckets:
            b["flat"].zero_()

    def grads(self):
        return [b["flat"] for b in self.buckets]

    def nbytes(self):
        return sum(b["flat"].numel() * b["flat"].element_size() for b in self.buckets)


def attach(model, bucket_mb=64.0):
    """Make `model` data-parallel for an UNMODIFIED single-process training script started once per GPU (torchrun / one process per
    rank with WORLD_SIZE, RANK, LOCAL_RANK, MASTER_ADDR, MASTER_PORT set and HIP_VISIBLE_DEVICES pinned so that train.py:99-102 sees
    one device and does not wrap the model in nn.DataParallel).  The process group is created on first use (RCCL = backend "nccl" when
    the parameters live on a GPU, gloo on the CPU) and gradients are averaged in flat buckets during backward (GradBuckets,
    auto_finalize).  models.ADNMUNet.create_ADNMUNet calls this by itself when WORLD_SIZE > 1 (ADNM_AUTO_DDP=0 turns that off).
    Replaces the reference's nn.DataParallel scatter / replicate / gather (train.py:99-102)."""
    state = {"buckets": None}

    def first_forward(module, args):
        if state["buckets"] is not None:
            return
        p0 = next(module.parameters())
        if not dist.is_initialized():
            os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
            if p0.is_cuda:
                dist.init_process_group("nccl", device_id=p0.device)
            else:
                dist.init_process_group("gloo")
        # replicas must start identical (train.py builds the model from an unseeded initialiser on every rank): rank 0's values win
        with torch.no_grad():
            for t in list(module.parameters()) + list(module.buffers()):
                dist.broadcast(t.data, src=0)
        state["buckets"] = GradBuckets(module, bucket_mb=bucket_mb, auto_finalize=True)
        module._adnm_grad_buckets = state["buckets"]

    model.register_forward_pre_hook(first_forward)
    return model
