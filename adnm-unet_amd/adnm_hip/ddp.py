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
import torch
import torch.distributed as dist


class GradBuckets:
    def __init__(self, module, process_group=None, bucket_mb=64.0):
        self.group = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self.params = [p for p in module.parameters() if p.requires_grad]
        self.bucket_bytes = int(bucket_mb * (1 << 20))
        self.ready = False
        self.buckets = []      # dicts: flat, views{param:view}, pending, work
        self.where = {}        # param -> (bucket index)
        self._hooks = []
        self._avg = None

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

    def _on_grad(self, p):
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
        """Call after loss.backward(): returns when every gradient holds the rank average."""
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
