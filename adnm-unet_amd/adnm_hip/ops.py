"""Raw kernel wrappers (`k_*`, no autograd) and torch.autograd.Function wrappers over
libadnm_hip.so.  Everything here launches on torch's current stream and allocates through
torch's caching allocator only, so a whole training step is hipGraph-capturable.

Layout convention: token tensors are (B, L, C) / (M, C) channels-last; a "row view" is a 2-D
tensor with stride (ld, 1) — kernels take the row stride, so column slices of wide buffers are
passed without copies."""
import math
import os
import threading

import torch

from . import lib

_DT = {torch.float32: lib.F32, torch.bfloat16: lib.BF16}

# Matrix-core precision of the GEMM-shaped kernels (short / tall-skinny GEMMs, dense 3x3 convs; include/adnm_hip.h ADNM_MFMA_*):
#   "f32"  exact fp32 MFMA (the parity path);
#   "bf16" operands rounded to bf16 into v_mfma_f32_16x16x32_bf16, fp32 accumulation (BASELINE configs 2-4);
#   "fp8"  BASELINE config 5: per-tensor scaled OCP fp8 operands (e4m3 activations / weights, e5m2 output gradients) into
#          v_mfma_f32_16x16x32_{fp8,bf8}_fp8 for the forward and input-gradient GEMMs / convs; weight gradients stay on bf16 operands.
#          Scales are delayed (QUANT below): a call site's scale comes from the amax its kernel saw in an earlier step.
# A process-wide setting like torch's autocast state; kernels take it as an explicit argument.
MFMA_PREC = [0]
_PREC_NAMES = {"f32": 0, "fp32": 0, "bf16": 1, "fp8": 2}


def set_mfma_precision(name):
    MFMA_PREC[0] = _PREC_NAMES[name]


def mfma_precision():
    return ("f32", "bf16", "fp8")[MFMA_PREC[0]]


class QuantTable:
    """Quantisation records of the fp8 configuration (include/adnm_hip.h: `q` of the GEMM-shaped entry points, adnm_quant_update): one
    32-byte device record per GEMM call site, keyed by (stable key of the weight, role), role "f" = forward (first operand = activation
    rows, e4m3) or "g" = input-gradient (first operand = output gradient, e5m2).  Delayed per-tensor scaling: while a record's `record`
    flag is set the kernels collect max |value| of both operands; update() — one launch per training step — turns those into the next
    scales every `period` steps.  Records are created eagerly (a host -> device write): every call site must have run once before a
    hipGraph capture (FlatTrainer.prepare's calibration step does that, on bf16 operands).
    The table is per device; lookups are lock-guarded (autograd's engine threads)."""
    CAP = 2048

    def __init__(self):
        self._lock = threading.Lock()
        self._dev = {}        # device index -> {"tab": (CAP, 8) fp32, "state": (2,), "keys": {key: row}}
        self.calibrating = False
        self.bf16_keys = set()   # weights (stable keys) whose GEMMs stay on bf16 operands in the fp8 configuration
        # GEMMs / convs over more token rows than this keep bf16 operands in the fp8 configuration.  At config 2 (B = 4) that is the
        # 128x128 and 64x64 levels (65 536 / 16 384 rows: refiner, encoder1-2, decoder5-6, the output head): their weights are tiny
        # (<= 26 k elements: the activations are the byte stream, fp8 operands save nothing) and their reductions short (K = 32 .. 128:
        # no averaging of the 2^-4 rounding steps), and they touch the data directly.  Measured (tools/fp8_error.py,
        # profiles/r03_fp8_error.txt): output rel-L2 vs fp32 0.20 with every GEMM on fp8, 0.11 with this rule, 0.06 fp8 below 32x32 only.
        self.max_rows = int(os.environ.get("ADNM_FP8_MAX_ROWS", "8192"))
        self.period = int(os.environ.get("ADNM_FP8_PERIOD", "16"))
        self.headroom = float(os.environ.get("ADNM_FP8_HEADROOM", "2.0"))

    def active(self):
        return MFMA_PREC[0] == 2 or self.calibrating

    def _ent(self, device):
        ent = self._dev.get(self._idx(device))
        if ent is None:
            ent = self._dev[self._idx(device)] = {"tab": torch.zeros((self.CAP, 8), dtype=torch.float32, device=device),
                                            "state": torch.tensor([0.0, float(self.period)], dtype=torch.float32, device=device), "keys": {}}
        return ent

    def keep_bf16(self, keys):
        """the GEMMs / convs of these weights (data_ptr keys) run on bf16 operands in the fp8 configuration: the customary exemption of
        the layers that touch the data directly (VisionMamba.fp8_exempt_parameters names the input embedding and the output head)"""
        self.bf16_keys = set(keys)

    @staticmethod
    def _idx(device):
        return device.index if device.index is not None else torch.cuda.current_device()

    def record(self, device, key, role, rows=0):
        """-> device pointer of the call site's record, or None when neither fp8 nor a calibration pass is on, or the call site is exempt
        (more than max_rows token rows, or a weight named by keep_bf16): it then runs on bf16 operands."""
        if not self.active() or rows > self.max_rows or key in self.bf16_keys:
            return None
        with self._lock:
            ent = self._ent(device)
            row = ent["keys"].get((key, role))
            if row is None:
                if torch.cuda.is_current_stream_capturing():
                    raise RuntimeError("adnm_hip fp8: a GEMM call site ran for the first time inside a hipGraph capture (no quantisation record yet); "
                                       "run the step eagerly once before capturing (FlatTrainer.prepare does)")
                row = len(ent["keys"])
                if row >= self.CAP:
                    raise RuntimeError(f"adnm_hip fp8: more than {self.CAP} GEMM call sites")
                ent["keys"][(key, role)] = row
                ent["tab"][row] = torch.tensor([1.0, 1.0, 0.0, 0.0, 57344.0 if role[0] == "g" else 448.0, 448.0, 1.0, 0.0])
            return ent["tab"].data_ptr() + 32 * row

    def explicit(self, device, scale_a, scale_b, grad_a=False, record=False):
        """a stand-alone record with given scales (tests / callers that manage their own scaling): (8,) fp32 tensor, pass its data_ptr"""
        return torch.tensor([scale_a, scale_b, 0.0, 0.0, 57344.0 if grad_a else 448.0, 448.0, 1.0 if record else 0.0, 0.0], dtype=torch.float32,
                            device=device)

    def weight_row(self, device, key):
        """-> row of the record of WEIGHT `key` (a parameter's data_ptr) in this device's table: the b-slots hold the scale its fp8 shadow is
        written with (adnm_adamw_step) and the max |w| the optimiser pass collects for the next one; the a-slots are unused."""
        with self._lock:
            ent = self._ent(device)
            row = ent["keys"].get((key, "w"))
            if row is None:
                if torch.cuda.is_current_stream_capturing():
                    raise RuntimeError("adnm_hip fp8: weight records are created before capture (FlatTrainer.prepare)")
                row = len(ent["keys"])
                if row >= self.CAP:
                    raise RuntimeError(f"adnm_hip fp8: more than {self.CAP} records")
                ent["keys"][(key, "w")] = row
                ent["tab"][row] = torch.tensor([1.0, 1.0, 0.0, 0.0, 0.0, 448.0, 1.0, 0.0])
            return row

    def table_ptr(self, device):
        return self._ent(device)["tab"].data_ptr()

    def scale_b_view(self, device, row):
        """1-element view of a record's scale_b (what the fp8 shadow of a weight was / will be multiplied by)"""
        return self._ent(device)["tab"][row, 1:2]

    ROLES = {"linear_fwd": "fnt", "linear_dgrad": "gnn", "conv3_fwd": "fc3", "conv3_dgrad": "gc3", "convt_fwd": "fnn", "convt_dgrad": "gnt"}

    def set(self, device, key, site, scale_a, scale_b, record=False):
        """create / overwrite the record of call site (key, site) with explicit scales (tests, callers that manage their own scaling);
        site: one of ROLES.  Needs active() (fp8 mode or a calibration pass)."""
        role = self.ROLES[site]
        with self._lock:
            ent = self._ent(device)
            row = ent["keys"].get((key, role))
            if row is None:
                row = len(ent["keys"])
                ent["keys"][(key, role)] = row
            ent["tab"][row] = torch.tensor([scale_a, scale_b, 0.0, 0.0, 57344.0 if role[0] == "g" else 448.0, 448.0, 1.0 if record else 0.0, 0.0])
            return ent["tab"][row]

    def update(self, device):
        """once per training step, after it: amax -> scales on calibration steps, set the record flags of the next step"""
        ent = self._dev.get(self._idx(device))
        if ent is None:
            return
        lib.call("adnm_quant_update", ent["tab"].data_ptr(), len(ent["keys"]), ent["state"].data_ptr(), float(self.headroom), _stream())

    def reset(self, device=None):
        """Start a new calibration.  The table's memory is NEVER given back (captured hipGraphs have `tab.data_ptr() + 32 * row` baked
        into their kernel arguments).  While a capture that may hold record pointers is alive on the device (pin() / unpin(), taken by
        FlatTrainer and GraphedForward around the lifetime of their graphs) the key -> row map is kept as well and only the VALUES go
        back to their defaults (scale 1, recording on), so a replayed graph keeps reading the record of ITS call site — freshly
        calibrated by whoever asked for the reset.  With nothing pinned the keys are forgotten too (tests building model after model)."""
        with self._lock:
            ents = list(self._dev.values()) if device is None else [e for e in (self._dev.get(self._idx(device)),) if e is not None]
            for ent in ents:
                if ent.get("pins", 0) > 0:
                    n = len(ent["keys"])
                    ent["tab"][:n, 0:2] = 1.0
                    ent["tab"][:n, 2:4] = 0.0
                    ent["tab"][:n, 6] = 1.0
                else:
                    ent["keys"].clear()
                    ent["tab"].zero_()
                ent["state"].copy_(torch.tensor([0.0, float(self.period)]))

    def snapshot(self, device):
        ent = self._ent(device)
        return ent["tab"].clone(), ent["state"].clone()

    def restore(self, device, snap):
        ent = self._ent(device)
        ent["tab"].copy_(snap[0])
        ent["state"].copy_(snap[1])

    def pin(self, device):
        """a captured graph that may hold record pointers of this device's table now exists (see reset())"""
        with self._lock:
            ent = self._ent(device)
            ent["pins"] = ent.get("pins", 0) + 1

    def unpin(self, device):
        with self._lock:
            ent = self._dev.get(self._idx(device))
            if ent is not None and ent.get("pins", 0) > 0:
                ent["pins"] -= 1

    def dump(self, device):
        ent = self._dev.get(self._idx(device))
        if ent is None:
            return {}
        tab = ent["tab"].cpu()
        return {k: tab[r].tolist() for k, r in ent["keys"].items()}


QUANT = QuantTable()


def fp8_calibrate(device, fn):
    """Run fn() once on bf16 operands with every GEMM call site collecting its operands' amax, then turn those into fp8 scales (what
    FlatTrainer.prepare does for a training step; this is the stand-alone form for a forward-only / evaluation use).  Leaves the
    precision at "fp8"."""
    QUANT.reset(device)
    set_mfma_precision("bf16")
    QUANT.calibrating = True
    try:
        out = fn()
    finally:
        QUANT.calibrating = False
    QUANT.update(device)
    set_mfma_precision("fp8")
    return out


def _gemm_prec(q, role):
    """(prec, q pointer) of a GEMM-shaped call: the fp8 modes need a record; a calibration pass runs bf16 operands WITH the record."""
    p = MFMA_PREC[0]
    if p == 2:
        if q is None:   # an exempt weight (QUANT.keep_bf16)
            return 1, None
        return (3 if role == "g" else 2), q
    return (1 if QUANT.calibrating else p), q


def _dt(t):
    try:
        return _DT[t.dtype]
    except KeyError:
        raise RuntimeError(f"adnm_hip: unsupported activation dtype {t.dtype}")


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _rows(t):
    """(M, C) row view -> (data_ptr, ld). Leading dims must collapse to one row stride."""
    assert t.stride(-1) == 1 or t.shape[-1] == 1, "last dim must be contiguous"
    if t.dim() == 2:
        return t.data_ptr(), t.stride(0)
    ld = t.stride(-2)
    for i in range(t.dim() - 2):  # (B, L, C) with stride(0) == L * ld
        assert t.stride(i) == t.stride(i + 1) * t.shape[i + 1], "leading dims are not row-collapsible"
    return t.data_ptr(), ld


def _p(t):
    return None if t is None else t.data_ptr()


def _f32(t):
    assert t is None or (t.dtype == torch.float32 and t.is_contiguous()), "parameters must be contiguous fp32"
    return t


def _ws(nbytes, device):
    return torch.empty(max(int(nbytes), 16), dtype=torch.uint8, device=device)


class GradRegistry:
    """Gradient destinations.  A FlatTrainer registers, for every parameter (keyed by data_ptr — device pointers are unique
    across the GPUs of a process), the slice of ITS flat gradient buffer.  Backward functions that allocate a parameter gradient
    themselves take that slice instead of fresh memory, so the value is born in place and the trainer's gather copy for it
    disappears.  A second claim of the same slice inside one backward pass (a weight used by two autograd nodes) gets fresh
    memory: autograd then adds the two as usual.

    State is per owner (= per trainer, hence per device / per replica) and guarded by a lock: autograd runs backward on its own
    engine threads, one per device, so under the reference's nn.DataParallel flow (train.py:99-102) several threads come
    through here at once and must not see each other's claims."""

    def __init__(self):
        # re-entrant: register() / take() allocate GC-tracked objects under the lock, so the cyclic collector can run there and
        # finalise a dead trainer whose __del__ comes back in through drop() on the SAME thread
        self._lock = threading.RLock()
        self._dst = {}       # data_ptr -> (owner id, view into the owner's flat gradient buffer)
        self._claimed = {}   # owner id -> set of data_ptrs claimed in the running backward pass
        self._multi = {}     # owner id -> data_ptrs claimed more than once (autograd summed several contributions)
        self._dead = []      # owner ids dropped from a finaliser: list.append is atomic, no lock taken there

    def _drain(self):
        """(lock held) forget the owners whose trainers have been finalised since the last call"""
        while self._dead:
            owner = self._dead.pop()
            for ptr in [k for k, (o, _) in self._dst.items() if o == owner]:
                del self._dst[ptr]
            self._claimed.pop(owner, None)
            self._multi.pop(owner, None)

    def register(self, owner, mapping):
        with self._lock:
            self._dead.append(owner)
            self._drain()
            for ptr, view in mapping.items():
                self._dst[ptr] = (owner, view)
            self._claimed[owner] = set()
            self._multi[owner] = set()

    def drop(self, owner):
        """Lock-free (callable from __del__, i.e. from inside the collector on any thread): the owner is queued and its
        entries disappear at the next register / take / reset_claims."""
        self._dead.append(owner)

    def reset_claims(self, owner):
        with self._lock:
            self._drain()
            if owner in self._claimed:
                self._claimed[owner].clear()
                self._multi[owner].clear()

    def born_in_place(self, owner):
        """data_ptrs of the parameters whose slice holds the complete gradient of the running step: claimed exactly once."""
        with self._lock:
            return set(self._claimed.get(owner, ())) - self._multi.get(owner, set())

    def take_accumulating(self, ptr, shape, device):
        """-> (tensor, accumulate).  Like take(), for a kernel whose gradient goes through a DEFERRED fold: a second claim of a registered
        slice inside one backward pass gets the slice again with accumulate = True — the caller marks its queued fold as accumulating
        (adnm_foldq_accumulate_next) and hands autograd None for that input, so no separate add (and no flush) is needed.  Outside a
        deferring trainer this is take()."""
        with self._lock:
            self._drain()
            ent = self._dst.get(ptr)
            if ent is not None:
                owner, g = ent
                if ptr in self._claimed[owner] and ptr not in self._multi[owner] and tuple(g.shape) == tuple(shape) and FOLDS.deferring(device):
                    return g.detach(), True
        return self.take(ptr, shape, device), False

    def take(self, ptr, shape, device, dtype=torch.float32, strides=None):
        """strides: the memory layout the caller is going to WRITE (element strides of `shape`); a registered slice laid out
        differently is refused WITHOUT being recorded as claimed, so the trainer's gather still copies that gradient."""
        second = False
        with self._lock:
            self._drain()
            ent = self._dst.get(ptr)
            if ent is not None:
                owner, g = ent
                fits = tuple(g.shape) == tuple(shape) and g.dtype == dtype and (strides is None or tuple(g.stride()) == tuple(strides))
                # the same dense memory under another shape — a (4d, d, 1, 1) conv weight whose gradient a GEMM writes as (4d, d), a
                # (C, 1, 3, 3) depthwise weight written as (C, 9): the caller gets the slice viewed its way (25 M of config 2's 72 M
                # gradient elements used to be refused here and copied by the trainer's gather)
                dense = (not fits and g.dtype == dtype and g.is_contiguous() and g.numel() == math.prod(shape)
                         and (strides is None or tuple(strides) == tuple(torch.empty(tuple(shape), device="meta").stride())))
                if ptr not in self._claimed[owner] and (fits or dense):
                    self._claimed[owner].add(ptr)
                    # a fresh tensor object on the same memory: AccumulateGrad only keeps ("steals") a gradient nobody else references
                    return g.detach() if fits else g.detach().view(tuple(shape))
                second = ptr in self._claimed[owner]
                if second:
                    self._multi[owner].add(ptr)
        if second:   # autograd is about to ADD this contribution to the first one: both must be complete when it does
            FOLDS.flush(device, on_main=True)
            FOLDS.hold(device)   # ... so the producer of this one must not defer its fold either
        return torch.empty(tuple(shape), dtype=dtype, device=device)


class FoldRegistry:
    """Deferred second-stage folds (include/adnm_hip.h: adnm_foldq_*).  A trainer enables it for its device; the *_bwd wrappers that
    only produce PARAMETER gradients (or the single-consumer intermediates of the parameter-prep nodes) then bind the device's
    queue around their library call, so their fold launches are queued instead of issued, and keep their partial workspaces alive
    here.  flush() issues one launch per 16 queued folds; it is called before anything can read the results: at the head of the
    parameter-prep backward nodes, on the second claim of a gradient slice (autograd is about to ADD to it), and by the trainer
    after backward.  Disabled (the default, plain autograd use) every fold is launched where it is produced."""

    def __init__(self):
        self._lock = threading.Lock()
        self._q = {}   # device index -> {"h": queue handle, "keep": [...], "on": bool}

    def enable(self, device, on=True):
        import os
        if os.environ.get("ADNM_FOLD_DEFER", "1") == "0":
            on = False
        with self._lock:
            ent = self._q.get(device.index)
            if ent is None:
                # "lh": the leaf queue (grouped weight-gradient GEMMs, include/adnm_hip.h adnm_leafq_*): bound / flushed together with the
                # fold queue — its launches come first, the folds of their partials behind them
                ent = self._q[device.index] = {"h": lib.load().adnm_foldq_create(), "keep": [], "on": False,
                                               "lh": lib.load().adnm_leafq_create() if os.environ.get("ADNM_LEAF_DEFER", "1") != "0" else None}
        if not on:
            self.flush(device)
        ent["on"] = bool(on)

    def active(self, device, on=True):
        """with FOLDS.active(dev): ... — deferral is on for the kernels launched inside (from any thread: autograd runs backward on
        its own), the queue is flushed on the way out.  Scoped in time, so plain autograd users of the device are never deferred."""
        return _Active(self, device, on)

    def deferring(self, device):
        ent = self._q.get(device.index)
        return ent is not None and ent["on"] and not ent.get("hold", False)

    def hold(self, device):
        """The next defer() on this device (the call that asked for the gradient slice just refused) runs undeferred."""
        ent = self._q.get(device.index)
        if ent is not None:
            ent["hold"] = True

    def defer(self, device, *keep):
        """keep: the partial WORKSPACES (and other inputs of the queued folds) — never the destinations: a gradient tensor that
        something else references is not adopted by autograd's AccumulateGrad but cloned on the spot, i.e. before the fold ran."""
        ent = self._q.get(device.index)
        if ent is not None and ent.pop("hold", False):
            ent = None   # (a call that is not deferred is not moved to the side stream either: SIDE.submit checks deferring())
        return _Deferred(ent, keep, self._lock)

    def abort(self, device):
        """after an exception inside a backward pass / capture: deferral off, nothing launched, queued entries and kept workspaces
        dropped (the C-side queue is emptied by a flush into a throw-away state only if something is still pending)"""
        ent = self._q.get(device.index)
        if ent is None:
            return
        ent["on"] = False
        ent.pop("hold", None)
        try:
            lib.load().adnm_foldq_bind(None)
            lib.load().adnm_leafq_bind(None)
            if ent["lh"] is not None:
                lib.load().adnm_leafq_clear(ent["lh"])
            if lib.query("adnm_foldq_pending", ent["h"]) > 0:
                lib.load().adnm_foldq_clear(ent["h"])
        finally:
            with self._lock:
                ent["keep"].clear()

    def flush(self, device, on_main=False):
        """Issue the queued leaf launches and folds.  With the side stream on (SideStreams) they go THERE — forked from the current stream's
        present, joined only at the end of the backward pass — unless the caller reads the results on the current stream right away
        (on_main: a second claim of a gradient slice); the parameter-prep backward nodes, the other readers, move to the side stream
        themselves (late())."""
        ent = self._q.get(device.index)
        if ent is None:
            return
        sent = None if on_main else SIDE.live(device)
        with torch.cuda.device(device):
            if sent is not None:
                SIDE._launch(sent)   # what was collected for the side stream first, in submission order
                SIDE.fork(sent)      # everything the queued launches read exists on the current stream by now
                with torch.cuda.stream(sent["stream"]):
                    self._issue(ent)
                with self._lock:     # the workspaces / operands of what now runs beside the main stream: alive until the join
                    sent["keep"].extend(ent["keep"])
                    ent["keep"].clear()
                return
            SIDE.join(device)   # queued folds (and whoever asked for the flush) read what weight-gradient kernels wrote on the side stream
            self._issue(ent)
        with self._lock:
            ent["keep"].clear()

    @staticmethod
    def _issue(ent):
        if ent["lh"] is not None and lib.query("adnm_leafq_pending", ent["lh"]) > 0:
            lib.call("adnm_leafq_flush", ent["lh"], _stream())   # the grouped leaf launches first: the folds read their partials
        if lib.query("adnm_foldq_pending", ent["h"]) > 0:
            lib.call("adnm_foldq_flush", ent["h"], _stream())

    def late(self, device):
        """with FOLDS.late(dev): <launch a kernel that reads deferred fold results and writes only parameter gradients> — the flush and the
        body run on the side stream when it is on (the parameter-prep backward nodes), else on the current stream."""
        return _Late(self, device)


class _Late:
    def __init__(self, reg, device):
        self.reg, self.device, self.ctx = reg, device, None

    def __enter__(self):
        self.reg.flush(self.device)
        sent = SIDE.live(self.device)
        if sent is not None:
            self.ctx = torch.cuda.stream(sent["stream"])
            self.ctx.__enter__()
            sent["dirty"] = True

    def __exit__(self, *exc):
        if self.ctx is not None:
            self.ctx.__exit__(*exc)
        return False


class _Active:
    def __init__(self, reg, device, on):
        self.reg, self.device, self.on = reg, device, on and device.type == "cuda"

    def __enter__(self):
        if self.on:
            self.reg.enable(self.device, True)

    def __exit__(self, *exc):
        if self.on:
            self.reg.enable(self.device, False)   # flushes
        return False


class _Deferred:
    def __init__(self, ent, keep, lock):
        self.ent, self.keep, self.lock = (ent if ent is not None and ent["on"] else None), keep, lock

    def deferring(self):
        return self.ent is not None

    def __enter__(self):
        if self.ent is not None:
            lib.load().adnm_foldq_bind(self.ent["h"])
            if self.ent["lh"] is not None:
                lib.load().adnm_leafq_bind(self.ent["lh"])
        return self.ent is not None

    def __exit__(self, *exc):
        if self.ent is not None:
            lib.load().adnm_foldq_bind(None)
            lib.load().adnm_leafq_bind(None)
            with self.lock:
                self.ent["keep"].extend(t for t in self.keep if t is not None)
        return False


class SideStreams:
    """Weight-gradient kernels off the critical path.  In backward the input-gradient chain (dY -> dX -> the layer below) is one long
    dependency chain of mostly small grids, while each layer's weight gradient (dW = dY^T X, depthwise / dense conv tap gradients) is
    a leaf that nothing reads before the optimiser.  A trainer turns this on around its backward pass; the weight-gradient wrappers then
    SUBMIT their library call instead of making it: calls are collected and, every `batch` of them, launched together on a second HIP
    stream that forks from the current stream there (one event: every operand of the batch exists by then) and joins it again before
    anything reads the results: every FOLDS.flush (the second-stage folds read the partials those kernels wrote) and the end of the
    pass.  Under hipGraph capture the fork / join events become graph edges, i.e. parallel branches of the replayed graph; a fork costs
    several microseconds on both streams, hence the batching.
    Operands and workspaces are allocated on the MAIN stream before the fork and kept alive here until the join, so the caching
    allocator cannot hand their memory to a later main-stream tensor while the side kernel still runs.  The OUTPUTS are never kept (a
    second reference would make autograd's AccumulateGrad clone a gradient instead of adopting it — on the main stream, before the side
    kernel ran): they are parameter gradients that stay referenced by autograd, or tap gradients consumed by a parameter-prep node
    whose first action is FOLDS.flush.
    Off (the default, plain autograd use): every call is made at once on the current stream."""

    def __init__(self):
        self._lock = threading.Lock()
        self._s = {}   # device index -> {"stream", "on", "dirty", "keep", "pending"}
        self.batch = int(os.environ.get("ADNM_SIDE_BATCH", "16"))

    def active(self, device, on=True):
        return _SideActive(self, device, on and device.type == "cuda" and os.environ.get("ADNM_SIDE_STREAM", "1") != "0")

    def _ent(self, device):
        with self._lock:
            ent = self._s.get(device.index)
            if ent is None:
                ent = self._s[device.index] = {"stream": torch.cuda.Stream(device=device), "on": False, "dirty": False, "keep": [], "pending": []}
        return ent

    def live(self, device):
        """the device's entry while the side stream is ON (inside a trainer's backward pass), else None"""
        ent = self._s.get(device.index)
        return ent if ent is not None and ent["on"] else None

    def keep(self, device, *tensors):
        """hold tensors a side-stream kernel reads until the join (no-op with the side stream off)"""
        ent = self.live(device)
        if ent is not None:
            with self._lock:
                for t in tensors:
                    if isinstance(t, (list, tuple)):
                        ent["keep"].extend(x for x in t if x is not None)
                    elif t is not None:
                        ent["keep"].append(t)

    def fork(self, ent):
        """the side stream waits for everything enqueued on the current stream so far"""
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(ent["stream"].device))
        ent["stream"].wait_event(ev)
        ent["dirty"] = True

    def submit(self, device, keep, deferred, fn):
        """fn() makes the library call on torch's CURRENT stream (it must read _stream() itself) inside `deferred` (a FOLDS.defer(...)
        context, created by the caller so that a pending hold is honoured now)."""
        ent = self._s.get(device.index)
        if ent is None or not ent["on"] or not deferred.deferring():
            with deferred:
                fn()
            return
        with self._lock:
            ent["pending"].append((deferred, fn))
            ent["keep"].extend(t for t in keep if t is not None)
            full = len(ent["pending"]) >= self.batch
        if full:
            self._launch(ent)

    def _launch(self, ent):
        with self._lock:
            todo, ent["pending"] = ent["pending"], []
        if not todo:
            return
        side = ent["stream"]
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(side.device))   # fork: the operands of every collected call exist on the main stream by now
        side.wait_event(ev)
        with torch.cuda.stream(side):
            for deferred, fn in todo:
                with deferred:
                    fn()
        ent["dirty"] = True

    def abort(self, device):
        """after an exception: forget what was collected (nothing is launched), switch off"""
        ent = self._s.get(device.index)
        if ent is None:
            return
        with self._lock:
            ent["on"] = False
            ent["pending"] = []
            ent["keep"].clear()

    def join(self, device):
        """launch what is still collected, then the current stream waits for everything on the side stream"""
        ent = self._s.get(device.index)
        if ent is None:
            return
        self._launch(ent)
        if not ent["dirty"]:
            return
        ev = torch.cuda.Event()
        ev.record(ent["stream"])
        torch.cuda.current_stream(device).wait_event(ev)
        with self._lock:
            ent["dirty"] = False
            ent["keep"].clear()


class _SideActive:
    def __init__(self, reg, device, on):
        self.reg, self.device, self.on = reg, device, on

    def __enter__(self):
        if self.on:
            self.reg._ent(self.device)["on"] = True

    def __exit__(self, *exc):
        if self.on:
            self.reg._ent(self.device)["on"] = False
            self.reg.join(self.device)
        return False


class _UncachedRegion:
    """`nbytes` of zero-filled uncached device memory (include/adnm_hip.h: adnm_uncached_alloc); freed with the object."""

    def __init__(self, device, nbytes):
        with torch.cuda.device(device):
            self.ptr = lib.load().adnm_uncached_alloc(int(nbytes))
        if not self.ptr:
            raise RuntimeError(f"adnm_hip: {lib.last_error()}")
        self.nbytes, self.device = int(nbytes), device

    def __del__(self):
        try:
            if self.ptr:
                with torch.cuda.device(self.device):
                    lib.load().adnm_uncached_free(self.ptr)
                self.ptr = None
        except Exception:
            pass


class _SplitRegion:
    """what one unit of ordering owns for its split GEMM launches: zeroed arrival counters in ordinary device memory (a torch tensor from
    the regular allocator — allocated OUTSIDE any capture) and `nbytes` of uncached slab space"""
    COUNTER_BYTES = 1 << 20   # 256 k output tiles per launch

    def __init__(self, device, nbytes):
        self.counters = torch.zeros(self.COUNTER_BYTES // 4, dtype=torch.int32, device=device)
        self.slabs = _UncachedRegion(device, nbytes)
        self.nbytes = self.slabs.nbytes


class SplitWorkspaces:
    """The caller's side of the in-launch split-K combine (include/adnm_hip.h, adnm_skgemm): the arrival counters and the UNCACHED slab
    space of a split NT / NN launch, owned here per unit of ordering — never shared by launches that could run at the same time:
      * eager launches: one region per (device, stream); launches on a stream are ordered, so one region serves them all.  It grows
        (a new, larger region; outgrown ones are kept until release(), an in-flight launch may still use them);
      * stream capture: one region per capture SCOPE.  The owner of the graphs (FlatTrainer, GraphedForward) opens a scope before it
        captures and keeps it as long as its graphs live; every split launch captured inside uses the scope's region, sized by the
        largest request this device has seen (the eager warm-up steps).  All graphs of one scope must replay on one stream, one at a
        time (FlatTrainer's stage graphs do).  A capture nobody opened a scope for gets an ordinary torch workspace with freshly
        zeroed counters and the fenced protocol."""
    MIN_BYTES = 16 << 20

    def __init__(self):
        self._lock = threading.Lock()
        self._streams = {}   # (device index, stream handle) -> [regions, newest last]
        self._seen = {}      # device index -> largest request so far
        self._scope = {}     # device index -> the open capture scope (a dict holding its region)

    def take(self, device, nbytes, counter_bytes):
        """-> the region (counters + uncached slab space) of one split launch on torch's current stream, or None (an un-scoped capture)"""
        idx = device.index if device.index is not None else torch.cuda.current_device()
        if counter_bytes > _SplitRegion.COUNTER_BYTES:
            raise RuntimeError(f"adnm_hip: a split GEMM with {counter_bytes // 4} output tiles (more than {_SplitRegion.COUNTER_BYTES // 4})")
        with self._lock:
            self._seen[idx] = max(self._seen.get(idx, 0), nbytes)
            if torch.cuda.is_current_stream_capturing():
                scope = self._scope.get(idx)
                if scope is None:
                    return None
                reg = scope["region"]
                if reg.nbytes < nbytes:
                    raise RuntimeError(f"adnm_hip: a split GEMM inside a hipGraph capture needs {nbytes} slab bytes, the capture scope "
                                       f"holds {reg.nbytes}: run the step eagerly once before capturing (FlatTrainer.prepare does)")
                return reg
            key = (idx, _stream())
            regs = self._streams.setdefault(key, [])
            if not regs or regs[-1].nbytes < nbytes:
                regs.append(_SplitRegion(device, max(self.MIN_BYTES, 2 * nbytes if regs else nbytes)))
            return regs[-1]

    def open_scope(self, device):
        """Before a capture (NOT inside one: it allocates).  -> the scope; keep it alive as long as the captured graphs."""
        idx = device.index if device.index is not None else torch.cuda.current_device()
        with self._lock:
            need = max(self.MIN_BYTES, self._seen.get(idx, 0))
        return {"region": _SplitRegion(device, need), "device": idx}

    def capturing(self, scope):
        """with SPLITWS.capturing(scope): <torch.cuda.graph(...)> — split launches captured inside use the scope's region"""
        return _ScopeActive(self, scope)

    def release(self):
        """forget every eager region (tests; nothing may be in flight)"""
        with self._lock:
            self._streams.clear()


class _ScopeActive:
    def __init__(self, reg, scope):
        self.reg, self.scope = reg, scope

    def __enter__(self):
        if self.scope is not None:
            with self.reg._lock:
                self.reg._scope[self.scope["device"]] = self.scope

    def __exit__(self, *exc):
        if self.scope is not None:
            with self.reg._lock:
                self.reg._scope.pop(self.scope["device"], None)
        return False


class ShadowRegistry:
    """Narrow SHADOW copies of the GEMM weights (include/adnm_hip.h: adnm_adamw_step `shadow`, adnm_skgemm `b_dtype`).  A FlatTrainer keeps,
    beside its flat fp32 parameter buffer, one buffer of the same element layout in bf16 (bf16 configuration) or per-tensor scaled OCP
    e4m3 (fp8 configuration), rewritten by the optimiser pass that updates the parameters.  The weight-streaming GEMMs (k_linear /
    k_linear_dx on the short-GEMM kernels) look their weight up here and read the shadow instead of the fp32 values: half / a quarter of
    the bytes of the two passes that stream the 72 M parameters every step, bit for bit the result of rounding the fp32 operand the
    same way.  Outside a trainer (plain autograd use) nothing is registered and the kernels convert the fp32 weight on the fly."""

    def __init__(self):
        self._lock = threading.Lock()
        self._own = {}   # owner id -> entry

    def register(self, owner, flat_p, shadow, b_dtype, seg_start, seg_row):
        """seg_start: sorted element offsets of the tensors inside flat_p; seg_row[k]: QUANT row of tensor k's weight record (fp8) or -1"""
        with self._lock:
            self._own[owner] = {"lo": flat_p.data_ptr(), "hi": flat_p.data_ptr() + 4 * flat_p.numel(), "shadow": shadow, "dt": b_dtype,
                                "starts": list(seg_start), "rows": list(seg_row), "dev": flat_p.device, "valid": False}

    def set_valid(self, owner, ok=True):
        with self._lock:
            if owner in self._own:
                self._own[owner]["valid"] = ok

    def drop(self, owner):
        self._own.pop(owner, None)   # (atomic: callable from a finaliser)

    def lookup(self, w, prec):
        """-> (shadow pointer of w, b_dtype, scale tensor or None) when the weight tensor `w` lies in a registered flat buffer whose shadow
        matches the matrix-core precision `prec` of the call (1: bf16 shadow; 2 / 3: fp8 shadow and a weight record), else None."""
        if not self._own or prec == 0 or QUANT.calibrating:
            return None
        ptr = w.data_ptr()
        for ent in list(self._own.values()):
            if ent["lo"] <= ptr < ent["hi"] and ent["valid"]:
                if (ent["dt"] == 1) != (prec == 1):
                    return None
                off = (ptr - ent["lo"]) // 4
                if ent["dt"] == 1:
                    return ent["shadow"].data_ptr() + 2 * off, 1, None
                import bisect
                k = bisect.bisect_right(ent["starts"], off) - 1
                row = ent["rows"][k]
                if row < 0:
                    return None
                return ent["shadow"].data_ptr() + off, 2, QUANT.scale_b_view(ent["dev"], row)
        return None


FOLDS = FoldRegistry()
SPLITWS = SplitWorkspaces()
SHADOWS = ShadowRegistry()
GRADS = GradRegistry()
SIDE = SideStreams()
grad_dst = GRADS.take


class _NoDefer:
    def deferring(self):
        return False

    def __enter__(self):
        return False

    def __exit__(self, *exc):
        return False


_NODEFER = _NoDefer()


def _need_gpu(t):
    if not t.is_cuda:
        raise RuntimeError("adnm_hip kernels run on the GPU only (there is no CPU fallback); got a CPU tensor")


# ======================================================================================= raw kernels
def k_rownorm_fwd(x2, w, b, scale, shift, eps, mean, out=None):
    _need_gpu(x2)
    M, d = x2.shape
    y = out if out is not None else torch.empty((M, d), dtype=x2.dtype, device=x2.device)
    mu = torch.empty(M if mean else 1, dtype=torch.float32, device=x2.device)
    rstd = torch.empty(M, dtype=torch.float32, device=x2.device)
    px, ldx = _rows(x2)
    py, ldy = _rows(y)
    lib.call("adnm_rownorm_fwd", px, ldx, _p(_f32(w)), _p(_f32(b)), _p(scale), _p(shift), py, ldy, mu.data_ptr(),
             rstd.data_ptr(), M, d, float(eps), int(mean), _dt(x2), _stream())
    return y, mu, rstd


def k_rownorm_bwd(dy2, x2, w, b, scale, mu, rstd, mean, want_b, want_affine, dx_out=None, dres=None, shift=None, defer=False):
    """defer: dw / db / dscale / dshift are parameter gradients (or prep intermediates) nobody reads before the next flush point."""
    M, d = x2.shape
    dev = x2.device
    dx = dx_out if dx_out is not None else torch.empty((M, d), dtype=x2.dtype, device=dev)
    dw = grad_dst(w.data_ptr(), (d,), dev)
    db = grad_dst(b.data_ptr(), (d,), dev) if want_b else None
    dsc = grad_dst(scale.data_ptr() if scale is not None else 0, (), dev) if want_affine else None
    dsh = grad_dst(shift.data_ptr() if shift is not None else 0, (), dev) if want_affine else None
    nb = lib.query("adnm_rownorm_bwd_ws_bytes", M, d)
    ws = _ws(nb, dev)
    pdy, lddy = _rows(dy2)
    px, ldx = _rows(x2)
    pdx, lddx = _rows(dx)
    pres, ldres = _rows(dres) if dres is not None else (None, 0)
    with FOLDS.defer(dev, ws) if defer else _NODEFER:
        lib.call("adnm_rownorm_bwd", pdy, lddy, px, ldx, _p(w), _p(b), _p(scale), mu.data_ptr(), rstd.data_ptr(), pdx, lddx,
                 dw.data_ptr(), _p(db), _p(dsc), _p(dsh), pres, ldres, ws.data_ptr(), nb, M, d, int(mean), _dt(x2), _stream())
    return dx, dw, db, dsc, dsh


def k_ssd_fwd(x, Bm, Cm, dt_raw, dt_bias, A_log, D, B, L, H, P, N, G, y=None, ln=None):
    """x (M,H*P) row view, Bm/Cm (M,G*N) row views, dt_raw (M,H) row view; M = B*L.
    ln = (weight, bias, out row view (M,H*P), eps): the mixer's LayerNorm (ADNssd.py:456) as the epilogue of pass 2 — only
    when H*P == 64 (the token row is one head block); returns (y, kv, mu, rstd) then."""
    _need_gpu(x)
    dev = x.device
    M = B * L
    if y is None:
        y = torch.empty((M, H * P), dtype=x.dtype, device=dev)
    kv = torch.empty((B, H, N, P), dtype=torch.float32, device=dev)
    nb = lib.query("adnm_ssd_ws_bytes", B, L, H, P, N, G)
    ws = _ws(nb, dev)
    px, ldx = _rows(x)
    pb, ldb = _rows(Bm)
    pc, ldc = _rows(Cm)
    pt, ldt = _rows(dt_raw)
    py, ldy = _rows(y)
    if ln is not None:
        lw, lb, yn, eps = ln
        mu = torch.empty(M, dtype=torch.float32, device=dev)
        rstd = torch.empty(M, dtype=torch.float32, device=dev)
        pn, ldn = _rows(yn)
        extra = (_p(_f32(lw)), _p(_f32(lb)), pn, ldn, mu.data_ptr(), rstd.data_ptr(), float(eps))
    else:
        extra = (None, None, None, 0, None, None, 0.0)
    lib.call("adnm_ssd_reduce_fwd", px, ldx, pb, ldb, pc, ldc, pt, ldt, 1, _p(_f32(dt_bias)), _p(_f32(A_log)), _p(_f32(D)), 1, py,
             ldy, kv.data_ptr(), *extra, ws.data_ptr(), nb, B, L, H, P, N, G, _dt(x), _stream())
    return (y, kv, mu, rstd) if ln is not None else (y, kv)


def k_ssd_bwd(dy, x, Bm, Cm, dt_raw, dt_bias, A_log, D, kv, dx, dBm, dCm, ddt, B, L, H, P, N, G):
    dev = x.device
    dbias = torch.empty(H, dtype=torch.float32, device=dev)
    dA = torch.empty(H, dtype=torch.float32, device=dev)
    dD = torch.empty(H, dtype=torch.float32, device=dev)
    nb = lib.query("adnm_ssd_ws_bytes", B, L, H, P, N, G)
    ws = _ws(nb, dev)
    a = []
    for t in (dy, x, Bm, Cm, dt_raw):
        a += list(_rows(t))
    g = []
    for t in (dx, dBm, dCm, ddt):
        g += list(_rows(t))
    with FOLDS.defer(dev, ws):   # the per-head statistics [dD | ddt_bias | dA_log] are parameter gradients
        lib.call("adnm_ssd_reduce_bwd", a[0], a[1], a[2], a[3], a[4], a[5], a[6], a[7], a[8], a[9], 1, _p(dt_bias), _p(A_log), _p(D), 1,
                 kv.data_ptr(), g[0], g[1], g[2], g[3], g[4], g[5], g[6], g[7], dbias.data_ptr(), dA.data_ptr(), dD.data_ptr(),
                 ws.data_ptr(), nb, B, L, H, P, N, G, _dt(x), _stream())
    return dbias, dA, dD


def k_dwconv_fwd(x, wt, bias, B, H, W, C, K, act, y=None, addend=None, chan_major=False):
    """x (M,C) row view (M=B*H*W); wt tap-major (K*K, C) fp32, or nn.Conv2d's (C, K*K) with chan_major."""
    _need_gpu(x)
    if y is None:
        y = torch.empty((B * H * W, C), dtype=x.dtype, device=x.device)
    px, ldx = _rows(x)
    py, ldy = _rows(y)
    pa, lda = _rows(addend) if addend is not None else (None, 0)
    lib.call("adnm_dwconv_fwd", px, ldx, _p(_f32(wt)), _p(_f32(bias)), pa, lda, py, ldy, B, H, W, C, K, K, act, int(chan_major), _dt(x), _stream())
    return y


def k_dwconv_bwd(dy, x, wt, bias, B, H, W, C, K, act, dx=None, want_bias=False, want_w=True, chan_major=False, want_dx=True):
    """want_dx=False (the input needs no gradient; only without activation, where the tap gradient reads dy itself): tap / bias gradients only."""
    dev = x.device
    if not want_dx:
        assert act == lib.ACT_NONE and want_w
    if dx is None and want_dx:
        dx = torch.empty((B * H * W, C), dtype=x.dtype, device=dev)
    dwt = torch.empty((C, K * K) if chan_major else (K * K, C), dtype=torch.float32, device=dev) if want_w else None
    db = torch.empty(C, dtype=torch.float32, device=dev) if want_bias else None
    dpre = torch.empty((B * H * W, C), dtype=x.dtype, device=dev) if act != lib.ACT_NONE else None
    nb = lib.query("adnm_dwconv_bwd_ws_bytes", B, H, W, C, K, K)
    ws = _ws(nb, dev)
    pdy, lddy = _rows(dy)
    px, ldx = _rows(x)
    pdx, lddx = _rows(dx) if dx is not None else (None, 0)
    if not want_w:
        lib.call("adnm_dwconv_bwd", pdy, lddy, px, ldx, _p(wt), _p(bias), _p(dpre), pdx, lddx, None, None, ws.data_ptr(),
                 nb, B, H, W, C, K, K, act, int(chan_major), _dt(x), _stream())
        return dx, dwt, db
    # input gradient (and the pre-activation gradient it needs) on the current stream, the tap / bias gradients as a leaf beside it
    if want_dx:
        lib.call("adnm_dwconv_bwd", pdy, lddy, px, ldx, _p(wt), _p(bias), _p(dpre), pdx, lddx, None, None, ws.data_ptr(),
                 nb, B, H, W, C, K, K, act, int(chan_major), _dt(x), _stream())
    g, ldg = (dpre.data_ptr(), C) if dpre is not None else (pdy, lddy)
    pdw, pdb, dt_ = _p(dwt), _p(db), _dt(x)
    # taps / bias gradients: parameters, or prep intermediates flushed at the prep node
    # (the operands are kept with the workspace: under a bound leaf queue the launch itself waits for the grouped flush)
    SIDE.submit(dev, (dy, x, dpre), FOLDS.defer(dev, ws, dy, x, dpre), lambda: lib.call(
        "adnm_dwconv_wgrad", g, ldg, px, ldx, pdw, pdb, ws.data_ptr(), nb, B, H, W, C, K, K, int(chan_major), dt_, _stream()))
    return dx, dwt, db


def k_haar_dwt(x, B, H, W, C, cx=1):
    """x: (B*H*W, C*cx) row view holding channel c at column c*cx -> (B*h2*w2, 4C)."""
    _need_gpu(x)
    h2, w2 = (H + 1) // 2, (W + 1) // 2
    y = torch.empty((B * h2 * w2, 4 * C), dtype=x.dtype, device=x.device)
    px, ldx = _rows(x)
    lib.call("adnm_haar_dwt", px, ldx, cx, y.data_ptr(), B, H, W, C, _dt(x), _stream())
    return y


def k_wt_level(x, B, H, W, C, cx, taps, K, flip=False):
    """One analysis level of WTConv2d in one launch: x (B*H*W, C*cx) fp32 row view -> (sub, tag), both (B*h2*w2, 4C); tag = depthwise KxK conv
    of sub = DWT(x) (flip: with the flipped taps, the backward form)."""
    _need_gpu(x)
    h2, w2 = (H + 1) // 2, (W + 1) // 2
    sub = torch.empty((B * h2 * w2, 4 * C), dtype=torch.float32, device=x.device)
    tag = torch.empty_like(sub)
    px, ldx = _rows(x)
    lib.call("adnm_wt_level", px, ldx, cx, _f32(taps).data_ptr(), sub.data_ptr(), tag.data_ptr(), B, H, W, C, K, int(flip), _stream())
    return sub, tag


def k_haar_idwt(s, ll_add, B, H, W, C, y_add=(), up=()):
    """s: (B*h2*w2, 4C) contiguous (+ ll_add (B*h2*w2, C)) -> (B*H*W, C) [+ up to two contiguous (B*H*W, C) addends].
    up: the sub-band tensors of the next one / two coarser levels — the cascade in one launch; ll_add then belongs to the coarsest."""
    _need_gpu(s)
    assert s.is_contiguous() and (ll_add is None or ll_add.is_contiguous()) and len(up) <= 2 and all(u.is_contiguous() and u.dtype == s.dtype for u in up)
    adds = [a if a.is_contiguous() else a.contiguous() for a in y_add if a is not None]
    assert len(adds) <= 2
    adds += [None] * (2 - len(adds))
    ups = list(up) + [None] * (2 - len(up))
    y = torch.empty((B * H * W, C), dtype=s.dtype, device=s.device)
    lib.call("adnm_haar_idwt", s.data_ptr(), _p(ups[0]), _p(ups[1]), _p(ll_add), _p(adds[0]), _p(adds[1]), y.data_ptr(), B, H, W, C, _dt(s), _stream())
    return y


def k_haar_synthesis(bands, shapes, B, C, y_add=()):
    """The synthesis cascade of WTConv2d (WTConv2d.py:128-141): bands[i] = level i's (B*h*w, 4C) sub-band tensor, shapes[i] = the (H, W) level
    i reconstructs.  Three levels per launch (the finest launch takes y_add); deeper pyramids chain launches from the coarsest end."""
    n = len(bands)
    nxt, top = None, n   # levels [top, n) are already folded into nxt, which sits on level top - 1's LL band
    while top > 0:
        lo = max(top - 3, 0)
        hh, ww = shapes[lo]
        nxt = k_haar_idwt(bands[lo], nxt, B, hh, ww, C, y_add=y_add if lo == 0 else (), up=tuple(bands[lo + 1:top]))
        top = lo
    return nxt


def k_instnorm_fwd(x, scale, shift, B, HW, C, eps, act):
    _need_gpu(x)
    assert x.is_contiguous()
    dev = x.device
    y = torch.empty_like(x)
    mu = torch.empty((B, C), dtype=torch.float32, device=dev)
    rstd = torch.empty((B, C), dtype=torch.float32, device=dev)
    nb = lib.query("adnm_instnorm_ws_bytes", B, HW, C)
    ws = _ws(nb, dev)
    lib.call("adnm_instnorm_fwd", x.data_ptr(), _p(scale), _p(shift), y.data_ptr(), mu.data_ptr(), rstd.data_ptr(), ws.data_ptr(), nb,
             B, HW, C, float(eps), act, _dt(x), _stream())
    return y, mu, rstd


def k_instnorm_bwd(dy, x, scale, shift, mu, rstd, B, HW, C, act, want_affine=True):
    dev = x.device
    assert dy.is_contiguous() and x.is_contiguous()
    dx = torch.empty_like(x)
    dsc = grad_dst(scale.data_ptr() if scale is not None else 0, (), dev) if want_affine else None
    dsh = grad_dst(shift.data_ptr() if shift is not None else 0, (), dev) if want_affine else None
    nb = lib.query("adnm_instnorm_ws_bytes", B, HW, C)
    ws = _ws(nb, dev)
    with FOLDS.defer(dev, ws):   # d scale / d shift: parameter gradients (per-workgroup partials + the shared fold)
        lib.call("adnm_instnorm_bwd", dy.data_ptr(), x.data_ptr(), _p(scale), _p(shift), mu.data_ptr(), rstd.data_ptr(), dx.data_ptr(),
                 _p(dsc), _p(dsh), ws.data_ptr(), nb, B, HW, C, act, _dt(x), _stream())
    return dx, dsc, dsh


def k_gate_fwd(h, F):
    _need_gpu(h)
    M = h.shape[0]
    y = torch.empty((M, F), dtype=h.dtype, device=h.device)
    ph, ldh = _rows(h)
    lib.call("adnm_gate_fwd", ph, ldh, y.data_ptr(), F, M, F, _dt(h), _stream())
    return y


def k_gate_bwd(dy, h, F):
    M = h.shape[0]
    dh = torch.empty((M, 2 * F), dtype=h.dtype, device=h.device)
    pdy, lddy = _rows(dy)
    ph, ldh = _rows(h)
    lib.call("adnm_gate_bwd", pdy, lddy, ph, ldh, dh.data_ptr(), 2 * F, M, F, _dt(h), _stream())
    return dh


def tap_major(w):
    """nn.Conv2d depthwise weight (C,1,KH,KW) -> tap-major (KH*KW, C) fp32: the layout the parameter-prep kernels emit (used by the tests
    to feed the raw tap-major entry points)."""
    c = w.shape[0]
    return w.reshape(c, -1).t().contiguous().float()


# ======================================================================================= autograd
class RowNormFn(torch.autograd.Function):
    """y = scale * ((x-mu)*rstd*w + b) + shift over the last dim (RMSNorm when mean=False)."""

    @staticmethod
    def forward(ctx, x, w, b, scale, shift, eps, mean):
        shp = x.shape
        x2 = x.reshape(-1, shp[-1])
        if x2.stride(-1) != 1:
            x2 = x2.contiguous()
        y, mu, rstd = k_rownorm_fwd(x2, w, b, scale, shift, eps, mean)
        ctx.save_for_backward(x2, w, b, scale, mu, rstd)
        ctx.mean, ctx.shp, ctx.has_shift, ctx.shift = mean, shp, shift is not None, shift
        return y.view(shp)

    @staticmethod
    def backward(ctx, dy):
        x2, w, b, scale, mu, rstd = ctx.saved_tensors
        dy2 = dy.reshape(-1, ctx.shp[-1])
        if dy2.stride(-1) != 1:
            dy2 = dy2.contiguous()
        dx, dw, db, dsc, dsh = k_rownorm_bwd(dy2, x2, w, b, scale, mu, rstd, ctx.mean, b is not None,
                                            scale is not None or ctx.has_shift, shift=ctx.shift, defer=True)
        return (dx.view(ctx.shp), dw, db, dsc if scale is not None else None, dsh if ctx.has_shift else None, None, None)


def rownorm(x, w, b=None, scale=None, shift=None, eps=1e-5, mean=True):
    return RowNormFn.apply(x, w, b, scale, shift, eps, mean)


class RowNormTapFn(torch.autograd.Function):
    """(norm(x), x): the head of a pre-norm residual block (ADNMUNet.py:149-152).  Handing x back as a second output lets
    backward add the residual path's gradient inside the row-norm kernel instead of in a separate autograd add."""

    @staticmethod
    def forward(ctx, x, w, b, scale, shift, eps, mean):
        shp = x.shape
        x2 = x.reshape(-1, shp[-1])
        if x2.stride(-1) != 1:
            x2 = x2.contiguous()
        y, mu, rstd = k_rownorm_fwd(x2, w, b, scale, shift, eps, mean)
        ctx.save_for_backward(x2, w, b, scale, mu, rstd)
        ctx.mean, ctx.shp, ctx.has_shift, ctx.shift = mean, shp, shift is not None, shift
        ctx.set_materialize_grads(False)
        return y.view(shp), x

    @staticmethod
    def backward(ctx, dy, dres):
        x2, w, b, scale, mu, rstd = ctx.saved_tensors
        d = ctx.shp[-1]
        if dy is None:
            return (dres, None, None, None, None, None, None)
        dy2 = dy.reshape(-1, d)
        if dy2.stride(-1) != 1:
            dy2 = dy2.contiguous()
        if dres is not None:
            dres = dres.reshape(-1, d)
            if dres.stride(-1) != 1:
                dres = dres.contiguous()
        dx, dw, db, dsc, dsh = k_rownorm_bwd(dy2, x2, w, b, scale, mu, rstd, ctx.mean, b is not None,
                                            scale is not None or ctx.has_shift, dres=dres, shift=ctx.shift, defer=True)
        return (dx.view(ctx.shp), dw, db, dsc if scale is not None else None, dsh if ctx.has_shift else None, None, None)


def rownorm_tap(x, w, b=None, scale=None, shift=None, eps=1e-5, mean=True):
    return RowNormTapFn.apply(x, w, b, scale, shift, eps, mean)


class CatMixFn(torch.autograd.Function):
    """cat((a1*x, a2*r), -1) [+ cat((a3*f, a4*f), -1)] in one pass each way (csrc/elementwise.hip)."""

    @staticmethod
    def forward(ctx, x, r, f, a1, a2, a3, a4):
        shp = x.shape
        d = shp[-1]
        rows = lambda t: t.reshape(-1, d) if t.reshape(-1, d).stride(-1) == 1 else t.reshape(-1, d).contiguous()
        x2, r2 = rows(x), rows(r)
        f2 = rows(f) if f is not None else None
        _need_gpu(x2)
        M = x2.shape[0]
        y = torch.empty((M, 2 * d), dtype=x.dtype, device=x.device)
        (px, ldx), (pr, ldr) = _rows(x2), _rows(r2)
        pf, ldf = _rows(f2) if f2 is not None else (None, 0)
        lib.call("adnm_catmix_fwd", px, ldx, pr, ldr, pf, ldf, _p(a1), _p(a2), _p(a3), _p(a4), y.data_ptr(), M, d, _dt(y), _stream())
        ctx.save_for_backward(x2, r2, f2, a1, a2, a3, a4)
        ctx.shp = shp
        return y.view(*shp[:-1], 2 * d)

    @staticmethod
    def backward(ctx, dy):
        x2, r2, f2, a1, a2, a3, a4 = ctx.saved_tensors
        M, d = x2.shape
        g = dy.reshape(M, 2 * d)
        g = g if g.stride(-1) == 1 else g.contiguous()
        need = ctx.needs_input_grad
        mk = lambda on: torch.empty((M, d), dtype=x2.dtype, device=x2.device) if on else None
        dx, dr, df = mk(need[0]), mk(need[1]), mk(need[2] and f2 is not None)
        da = torch.empty(4, dtype=torch.float32, device=x2.device)
        nb = lib.query("adnm_catmix_bwd_ws_bytes", M, d)
        ws = _ws(nb, x2.device)
        (pg, ldg), (px, ldx), (pr, ldr) = _rows(g), _rows(x2), _rows(r2)
        pf, ldf = _rows(f2) if f2 is not None else (None, 0)
        with FOLDS.defer(x2.device, ws):   # alpha1..4: one merge node per module
            lib.call("adnm_catmix_bwd", pg, ldg, px, ldx, pr, ldr, pf, ldf, _p(a1), _p(a2), _p(a3), _p(a4), _p(dx), _p(dr), _p(df), da.data_ptr(),
                     ws.data_ptr(), nb, M, d, _dt(g), _stream())
        v = lambda t: t.view(ctx.shp) if t is not None else None
        s = lambda a, i: da[i].view_as(a) if a is not None else None
        return v(dx), v(dr), v(df), s(a1, 0), s(a2, 1), s(a3 if f2 is not None else None, 2), s(a4 if f2 is not None else None, 3)


def _unsupported(op, why):
    raise RuntimeError(f"adnm_hip {op}: {why} (the HIP path has no PyTorch fallback)")


def catmix(x, r, f, a1, a2, a3=None, a4=None):
    """Head merge of Block / Attention / WTLayer: cat((a1 x, a2 r)) [+ cat((a3 f, a4 f))]."""
    _need_gpu(x)
    d = x.shape[-1]
    if r.shape != x.shape or (f is not None and f.shape != x.shape):
        _unsupported("catmix", f"operands must share one shape, got {tuple(x.shape)}, {tuple(r.shape)}" + (f", {tuple(f.shape)}" if f is not None else ""))
    if d % 4 or x.dtype not in _DT or any(a is not None and a.numel() != 1 for a in (a1, a2, a3, a4)):
        _unsupported("catmix", f"needs a channel count that is a multiple of 4 (got {d}), fp32/bf16 tokens and 1-element mixing parameters")
    return CatMixFn.apply(x, r, f, a1, a2, a3, a4)


class SSDReduceFn(torch.autograd.Function):
    """Stand-alone K1 (used by tests and by callers that hold x/B/C/dt as separate tensors)."""

    @staticmethod
    def forward(ctx, x, Bm, Cm, dt_raw, dt_bias, A_log, D, G):
        B, L, H, P = x.shape
        N = Bm.shape[-1] // G
        M = B * L
        xs, bs, cs, ts = x.reshape(M, H * P), Bm.reshape(M, -1), Cm.reshape(M, -1), dt_raw.reshape(M, H)
        xs, bs, cs, ts = [t if t.stride(-1) == 1 else t.contiguous() for t in (xs, bs, cs, ts)]
        y, kv = k_ssd_fwd(xs, bs, cs, ts, dt_bias, A_log, D, B, L, H, P, N, G)
        ctx.save_for_backward(xs, bs, cs, ts, dt_bias, A_log, D, kv)
        ctx.dims = (B, L, H, P, N, G)
        return y.view(B, L, H, P)

    @staticmethod
    def backward(ctx, dy):
        xs, bs, cs, ts, dt_bias, A_log, D, kv = ctx.saved_tensors
        B, L, H, P, N, G = ctx.dims
        M = B * L
        dy2 = dy.reshape(M, H * P)
        dy2 = dy2 if dy2.stride(-1) == 1 else dy2.contiguous()
        dx, dB, dC, ddt = (torch.empty_like(t, memory_format=torch.contiguous_format) for t in (xs, bs, cs, ts))
        dbias, dA, dD = k_ssd_bwd(dy2, xs, bs, cs, ts, dt_bias, A_log, D, kv, dx, dB, dC, ddt, B, L, H, P, N, G)
        return dx.view(B, L, H, P), dB.view(B, L, -1), dC.view(B, L, -1), ddt.view(B, L, H), dbias, dA, dD, None


def ssd_reduce(x, Bm, Cm, dt_raw, dt_bias, A_log, D, groups=1):
    return SSDReduceFn.apply(x, Bm, Cm, dt_raw, dt_bias, A_log, D, groups)


class DWConvFn(torch.autograd.Function):
    """Depthwise KxK 'same' conv (+bias, +activation) on (B, H*W, C) tokens; w is nn.Conv2d's (C,1,K,K)."""

    @staticmethod
    def forward(ctx, x, w, bias, H, W, act):
        B, L, C = x.shape
        K = w.shape[-1]
        x2 = x.reshape(B * L, C)
        x2 = x2 if x2.stride(-1) == 1 else x2.contiguous()
        wt = w.reshape(C, K * K)   # nn.Conv2d's own layout: the kernels read it as is (no tap-major copy either way)
        if wt.dtype != torch.float32 or not wt.is_contiguous():
            wt = wt.contiguous().float()
        y = k_dwconv_fwd(x2, wt, bias, B, H, W, C, K, act, chan_major=True)
        ctx.save_for_backward(x2, wt, bias)
        ctx.dims = (B, H, W, C, K, act, w.shape)
        return y.view(B, L, C)

    @staticmethod
    def backward(ctx, dy):
        x2, wt, bias = ctx.saved_tensors
        B, H, W, C, K, act, wshape = ctx.dims
        dy2 = dy.reshape(B * H * W, C)
        dy2 = dy2 if dy2.stride(-1) == 1 else dy2.contiguous()
        want_w = ctx.needs_input_grad[1]
        dx, dwt, db = k_dwconv_bwd(dy2, x2, wt, bias, B, H, W, C, K, act, want_bias=bias is not None, want_w=want_w, chan_major=True)
        return dx.view(B, H * W, C), dwt.view(wshape) if want_w else None, db, None, None, None


def dwconv(x, w, bias, H, W, act=lib.ACT_NONE):
    return DWConvFn.apply(x, w, bias, H, W, act)


class InstNormFn(torch.autograd.Function):
    """act(scale * InstanceNorm2d(x) + shift) on (B, H*W, C) tokens."""

    @staticmethod
    def forward(ctx, x, scale, shift, eps, act):
        B, L, C = x.shape
        x = x.contiguous()
        y, mu, rstd = k_instnorm_fwd(x, scale, shift, B, L, C, eps, act)
        ctx.save_for_backward(x, scale, shift, mu, rstd)
        ctx.act = act
        return y

    @staticmethod
    def backward(ctx, dy):
        x, scale, shift, mu, rstd = ctx.saved_tensors
        B, L, C = x.shape
        dx, dsc, dsh = k_instnorm_bwd(dy.contiguous(), x, scale, shift, mu, rstd, B, L, C, ctx.act)
        return dx, dsc if scale is not None else None, dsh if shift is not None else None, None, None


def instnorm(x, scale=None, shift=None, eps=1e-5, act=lib.ACT_NONE):
    return InstNormFn.apply(x, scale, shift, eps, act)


class GateFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, h):
        F = h.shape[-1] // 2
        h2 = h.reshape(-1, 2 * F)
        h2 = h2 if h2.stride(-1) == 1 else h2.contiguous()
        ctx.save_for_backward(h2)
        ctx.shp = h.shape
        return k_gate_fwd(h2, F).view(*h.shape[:-1], F)

    @staticmethod
    def backward(ctx, dy):
        (h2,) = ctx.saved_tensors
        F = h2.shape[-1] // 2
        dy2 = dy.reshape(-1, F)
        dy2 = dy2 if dy2.stride(-1) == 1 else dy2.contiguous()
        return k_gate_bwd(dy2, h2, F).view(ctx.shp)


def gate(h):
    return GateFn.apply(h)


class WTConvFn(torch.autograd.Function):
    """WTConv2d.forward (WTConv2d.py:100-153) on (B, H*W, C) tokens: Haar pyramid, depthwise KxK on every
    level's 4C sub-bands (per-channel wavelet_scale folded into the taps by the caller), inverse pyramid,
    plus the base depthwise conv (base_scale folded) — all by HIP kernels, manual backward.
    Arguments: x, base taps (K*K,C), base bias (C)|None, *level taps (K*K,4C).
    Returns (y, x): the second value is an autograd alias of the input for its OTHER consumer (the residual mixes of PatchEmbed /
    WTLayer / OutProj read x again: model_untils.py:306,310,418,881 of the reference) — its gradient is added inside the last synthesis
    kernel of the backward pass instead of by a separate autograd add."""

    @staticmethod
    def forward(ctx, x, H, W, K, base_wt, base_bias, *level_wt):
        B, L, C = x.shape
        x2 = x.reshape(B * L, C)
        x2 = x2 if x2.stride(-1) == 1 else x2.contiguous()
        levels = len(level_wt)
        shapes, subs, tags = [], [], []
        cur, cx, h, w = x2, 1, H, W
        fused = x2.dtype == torch.float32 and K in (3, 5)   # DWT + the level's stencil in one launch (csrc/wtlevel.hip)
        for i in range(levels):
            shapes.append((h, w))
            if fused:
                sub, tag = k_wt_level(cur, B, h, w, C, cx, level_wt[i], K)
            else:
                sub = k_haar_dwt(cur, B, h, w, C, cx)  # (B*h2*w2, 4C); its LL band (column c*4) feeds the next level
                tag = k_dwconv_fwd(sub, level_wt[i], None, B, (h + 1) // 2, (w + 1) // 2, 4 * C, K, lib.ACT_NONE)
            h, w = (h + 1) // 2, (w + 1) // 2
            subs.append(sub)
            tags.append(tag)
            cur, cx = sub, 4
        nxt = k_haar_synthesis(tags, shapes, B, C) if levels else None
        y = k_dwconv_fwd(x2, base_wt, base_bias, B, H, W, C, K, lib.ACT_NONE, addend=nxt)
        ctx.save_for_backward(x2, base_wt, base_bias, *level_wt, *subs)
        ctx.dims = (B, H, W, C, K, levels, shapes)
        ctx.set_materialize_grads(False)
        return y.view(B, L, C), x

    @staticmethod
    def backward(ctx, dy, dalias):
        B, H, W, C, K, levels, shapes = ctx.dims
        saved = ctx.saved_tensors
        x2, base_wt, base_bias = saved[0], saved[1], saved[2]
        level_wt, subs = saved[3:3 + levels], saved[3 + levels:]
        need_dx = ctx.needs_input_grad[0]
        if dy is None:   # only the alias was used
            return (dalias if need_dx else None, None, None, None, None, None, *([None] * levels))
        dy2 = dy.reshape(B * H * W, C)
        dy2 = dy2 if dy2.stride(-1) == 1 else dy2.contiguous()
        dxb, dbase, dbb = k_dwconv_bwd(dy2, x2, base_wt, base_bias, B, H, W, C, K, lib.ACT_NONE, want_bias=base_bias is not None, want_dx=need_dx)
        # the reconstruction's backward walks down: d(merged_i) = DWT(d r_{i-1}); its LL band is d r_i.  With an input gradient wanted, the
        # same launch also yields d sub_i = conv^T(d merged_i) (the fused level kernel on the flipped taps)
        dtags, dsubs = [], []
        cur, cx = dy2, 1
        fused = need_dx and dy2.dtype == torch.float32 and K in (3, 5)
        for i in range(levels):
            hh, ww = shapes[i]
            if fused:
                dm, dsub = k_wt_level(cur, B, hh, ww, C, cx, level_wt[i], K, flip=True)
                dsubs.append(dsub)
            else:
                dm = k_haar_dwt(cur, B, hh, ww, C, cx)
            dtags.append(dm)
            cur, cx = dm, 4
        # the analysis side's backward walks up: d(ll_{i-1}) = IDWT(d sub_i + [d ll_i on the LL band]); the last step adds the base conv's
        # input gradient and the alias's gradient in the same pass.  An input that needs no gradient (PatchEmbed.conv1: the radar frames)
        # skips every input-gradient kernel.  The tap gradients (dtags_i x sub_i) are leaves.
        dlw = [None] * levels
        for i in range(levels - 1, -1, -1):
            hh, ww = shapes[i]
            h2, w2 = (hh + 1) // 2, (ww + 1) // 2
            if fused:
                _, dlw[i], _ = k_dwconv_bwd(dtags[i], subs[i], level_wt[i], None, B, h2, w2, 4 * C, K, lib.ACT_NONE, want_dx=False)
            else:
                dsub, dlw[i], _ = k_dwconv_bwd(dtags[i], subs[i], level_wt[i], None, B, h2, w2, 4 * C, K, lib.ACT_NONE, want_dx=need_dx)
                dsubs.insert(0, dsub)
        if not need_dx:
            dx = None
        elif levels == 0:
            dx = dxb if dalias is None else dxb + dalias.reshape(B * H * W, C)
        else:   # the cascade, three levels per launch; the finest launch adds the base conv's and the alias's gradients
            dx = k_haar_synthesis(dsubs, shapes, B, C, y_add=(dxb, dalias.reshape(B * H * W, C) if dalias is not None else None))
        return (dx.view(B, H * W, C) if dx is not None else None, None, None, None, dbase, dbb, *dlw)


def wtconv(x, H, W, K, base_wt, base_bias, level_wts, tap=False):
    """tap=True: -> (y, alias of x) — hand the alias to x's other consumer (see WTConvFn)."""
    y, xa = WTConvFn.apply(x, H, W, K, base_wt, base_bias, *level_wts)
    return (y, xa) if tap else y


class ADNMixerFn(torch.autograd.Function):
    """ADNssd.Mamba2.forward (ADNssd.py:302-462) as one autograd node with a hand-written backward, so the
    wide intermediate buffers are written once and never sliced/zero-filled by autograd.

    Inputs are prepared by models.ADNssd.Mamba2 (tiny differentiable index ops on the PARAMETERS):
      w_in   (d_in_proj, dm): in_proj rows reordered to [z | x' | B' | C' | dt]; x' pairs the even/odd
             channel halves as alternating heads (head 2j+e <- half e, head j), B'/C' = [even | odd], so the
             reference's index_select gathers (ADNssd.py:329-341,375-386) vanish and K1 runs ONCE with G=2.
      taps   (9, 2di+2gN) = [czw | cw], tap-major: czw (9, di) the taps of conv2d_z; cw (9, di+2gN) the effective 3x3 taps of
             every xBC channel in kernel order: conv2d taps for the even channels, outer(conv_31, conv_13) for the four
             asymmetric chains (ADNssd.py:343-346; with no bias and zero padding a 3x1 o 1x3 chain IS a separable 3x3).
             z and xBC are adjacent column ranges of in_proj's output, so one depthwise launch applies both tap sets.
      tb     the matching (2di+2gN) bias or None (the reference configuration has conv_bias=False);
      ln_w/ln_b permuted like x';  w_out (dm, 2di) = alpha1 * out_proj
             with its y-columns permuted alike (ADNssd.py:459: alpha1 scales both halves).
    """

    @staticmethod
    def forward(ctx, u, w_in, taps, tb, dt_bias, A_log, D, ln_w, ln_b, w_out, H, W, P, N, scan_chunk=0, scan_groups=2, qkeys=(None, None),
                narrow=(None, None, None, None)):
        """narrow = (w_in_n, s_in, w_out_n, s_out): the prep's narrow copies of the two projections (AdnPrepMultiFn); w_in / w_out are then
        storage-less handles"""
        n_in = (narrow[0].data_ptr(), 1 if narrow[0].dtype == torch.bfloat16 else 2, narrow[1]) if narrow[0] is not None else None
        n_out = (narrow[2].data_ptr(), 1 if narrow[2].dtype == torch.bfloat16 else 2, narrow[3]) if narrow[2] is not None else None
        Bsz, L, dm = u.shape
        M = Bsz * L
        di = w_out.shape[1] // 2
        cx = taps.shape[1] - di  # di + 2gN
        nh = di // P
        u2 = u.reshape(M, dm)
        u2 = u2 if u2.is_contiguous() else u2.contiguous()
        # storage type of the node's wide internal tensors (proj, wide, y and their gradients): bf16 at the full-resolution level in the
        # bf16 configuration (low_storage), fp32 otherwise; u, the result and every parameter gradient stay fp32
        st = low_storage(M, [(w_in.shape[0], dm), (dm, 2 * di), (2 * di, dm), (dm, w_in.shape[0])]) if scan_chunk == 0 else u.dtype
        proj = k_linear(u2, w_in, None, qkey=qkeys[0], out_dtype=st, narrow=n_in)  # (M, 2di+2gN+nh) = [z | xBC | dt]
        # one wide buffer [LN(y) | silu(conv_z(z)) | silu(conv(xBC))]: its first 2di columns are out_proj's input, the rest K1's operands;
        # z and xBC are adjacent in `proj` and their conv outputs adjacent here, so ONE depthwise launch (taps = [czw | cw]) does both
        wide = torch.empty((M, 2 * di + cx), dtype=st, device=u.device)
        cat, xbc = wide[:, :2 * di], wide[:, 2 * di:]
        k_dwconv_fwd(proj[:, :di + cx], taps, tb, Bsz, H, W, di + cx, 3, lib.ACT_SILU, y=wide[:, di:])
        if scan_chunk == 0:   # linear_attn_duality=True: the global reduction (K1), both halves in one launch
            if di == 64:   # the token row is one head block: LayerNorm(y) (ADNssd.py:456) rides in pass 2's epilogue (refiner mixers)
                y, kv, mu, rstd = k_ssd_fwd(xbc[:, :di], xbc[:, di:di + 2 * N], xbc[:, di + 2 * N:], proj[:, di + cx:], dt_bias, A_log, D,
                                            Bsz, L, nh, P, N, 2, ln=(ln_w, ln_b, cat[:, :di], 1e-5))
            else:
                y, kv = k_ssd_fwd(xbc[:, :di], xbc[:, di:di + 2 * N], xbc[:, di + 2 * N:], proj[:, di + cx:], dt_bias, A_log, D,
                                  Bsz, L, nh, P, N, 2)
                mu = None
        else:                 # chunked scan (K1b): even half forward in time, odd half backward (ADNssd.py:416-439)
            y = torch.empty((M, di), dtype=u.dtype, device=u.device)
            Ns = N // scan_groups
            kv = torch.stack([k_ssd_scan_fwd(xbc[:, e * P:di], 2 * P, xbc[:, di + e * N:di + (e + 1) * N], xbc[:, di + 2 * N + e * N:di + 2 * N + (e + 1) * N],
                                             proj[:, di + cx + e:], 2, dt_bias[e:], A_log[e:], D[e:], 2, y[:, e * P:], 2 * P, Bsz, L, nh // 2, P, Ns,
                                             scan_groups, scan_chunk, e == 1) for e in (0, 1)])
            mu = None
        if mu is None:
            _, mu, rstd = k_rownorm_fwd(y, ln_w, ln_b, None, None, 1e-5, True, out=cat[:, :di])
        out = k_linear(cat, w_out, None, qkey=qkeys[1], out_dtype=u.dtype, narrow=n_out)
        ctx.save_for_backward(u2, w_in, taps, tb, dt_bias, A_log, D, ln_w, ln_b, w_out, proj, wide, y, kv, mu, rstd)
        ctx.dims = (Bsz, L, dm, H, W, P, N, di, cx, nh, scan_chunk, scan_groups)
        ctx.qkeys = qkeys
        ctx.narrow = (n_in, n_out, narrow)   # (the tuples hold raw pointers: `narrow` keeps the tensors alive)
        return out.view(Bsz, L, dm)

    @staticmethod
    def backward(ctx, dout):
        (u2, w_in, taps, tb, dt_bias, A_log, D, ln_w, ln_b, w_out, proj, wide, y, kv, mu, rstd) = ctx.saved_tensors
        Bsz, L, dm, H, W, P, N, di, cx, nh, scan_chunk, scan_groups = ctx.dims
        M = Bsz * L
        cat, xbc = wide[:, :2 * di], wide[:, 2 * di:]
        do = dout.reshape(M, dm)
        do = do if do.is_contiguous() else do.contiguous()
        dw_out, _ = k_linear_dw(do, cat, False)
        # the gradient of `wide`, laid out alike: out_proj's input gradient fills the first 2di columns, K1's backward the rest, and the
        # merged depthwise backward reads [d zc | d xbc] as one column range
        dwide = torch.empty_like(wide)
        dcat, dxbc = dwide[:, :2 * di], dwide[:, 2 * di:]
        k_linear_dx(do, w_out, out=dcat, qkey=ctx.qkeys[1], narrow=ctx.narrow[1])
        dy, dln_w, dln_b, _, _ = k_rownorm_bwd(dcat[:, :di], y, ln_w, ln_b, None, mu, rstd, True, True, False, defer=True)
        dproj = torch.empty_like(proj)
        if scan_chunk == 0:
            ddtb, dA, dD = k_ssd_bwd(dy, xbc[:, :di], xbc[:, di:di + 2 * N], xbc[:, di + 2 * N:], proj[:, di + cx:], dt_bias, A_log, D, kv,
                                     dxbc[:, :di], dxbc[:, di:di + 2 * N], dxbc[:, di + 2 * N:], dproj[:, di + cx:], Bsz, L, nh, P, N, 2)
        else:
            Ns = N // scan_groups
            parts = [k_ssd_scan_bwd(dy[:, e * P:], 2 * P, xbc[:, e * P:di], 2 * P, xbc[:, di + e * N:di + (e + 1) * N],
                                    xbc[:, di + 2 * N + e * N:di + 2 * N + (e + 1) * N], proj[:, di + cx + e:], 2, dt_bias[e:], A_log[e:], D[e:], 2,
                                    kv[e], dxbc[:, e * P:di], 2 * P, dxbc[:, di + e * N:di + (e + 1) * N],
                                    dxbc[:, di + 2 * N + e * N:di + 2 * N + (e + 1) * N], dproj[:, di + cx + e:], 2, Bsz, L, nh // 2, P, Ns,
                                    scan_groups, scan_chunk, e == 1) for e in (0, 1)]
            ddtb, dA, dD = (torch.stack((parts[0][k], parts[1][k]), dim=1).reshape(nh) for k in range(3))
        _, dtaps, dtb = k_dwconv_bwd(dwide[:, di:], proj[:, :di + cx], taps, tb, Bsz, H, W, di + cx, 3, lib.ACT_SILU, dx=dproj[:, :di + cx],
                                     want_bias=tb is not None)
        du = k_linear_dx(dproj, w_in, qkey=ctx.qkeys[0], out_dtype=u2.dtype, narrow=ctx.narrow[0])
        dw_in, _ = k_linear_dw(dproj, u2, False)
        return (du.view(Bsz, L, dm), dw_in, dtaps, dtb, ddtb, dA, dD, dln_w, dln_b, dw_out, None, None, None, None, None, None, None, None)


def adn_mixer(u, w_in, taps, tb, dt_bias, A_log, D, ln_w, ln_b, w_out, H, W, P, N, scan_chunk=0, scan_groups=2, qkeys=(None, None),
              narrow=(None, None, None, None)):
    """qkeys: stable identities (data_ptr of in_proj.weight / out_proj.weight) of the two projections for the fp8 call-site records —
    w_in / w_out themselves are per-step temporaries.  narrow: the prep's narrow copies (see ADNMixerFn.forward)."""
    return ADNMixerFn.apply(u, w_in, taps, tb, dt_bias, A_log, D, ln_w, ln_b, w_out, H, W, P, N, scan_chunk, scan_groups, qkeys, narrow)


class FeedForwardFn(torch.autograd.Function):
    """FeedForward.forward (model_untils.py:192-197 of the reference) as ONE autograd node with a hand-written backward:
    1x1 d -> 4d (+bias), depthwise 3x3 on 4d (+bias), gelu(x1) * sigmoid(x2), 1x1 2d -> d (+bias) on (B, H*W, d) tokens.  The three wide
    intermediates (4d, 4d, 2d columns) are internal: at the full-resolution level of the bf16 configuration they are stored in bf16
    (low_storage), which halves the bytes of the kernels that are bound by them; input, output and parameter gradients stay fp32.
    Weights: w_in (4d, d), dw (4d, 1, 3, 3) in nn.Conv2d's own layout, w_out (d, 2d)."""

    @staticmethod
    def forward(ctx, x, w_in, b_in, dw, b_dw, w_out, b_out, H, W):
        B, L, d = x.shape
        M = B * L
        F4, F2 = w_in.shape[0], w_out.shape[1]
        x2 = x.reshape(M, d)
        x2 = x2 if x2.is_contiguous() else x2.contiguous()
        w_in, w_out = w_in.contiguous(), w_out.contiguous()
        wt = dw.reshape(F4, 9)
        if wt.dtype != torch.float32 or not wt.is_contiguous():
            wt = wt.contiguous().float()
        st = low_storage(M, [(F4, d), (d, F4), (d, F2), (F2, d)])
        h1 = k_linear(x2, w_in, b_in, out_dtype=st)                                    # project_in
        h2 = k_dwconv_fwd(h1, wt, b_dw, B, H, W, F4, 3, lib.ACT_NONE, chan_major=True)   # dwconv
        g = k_gate_fwd(h2, F2)                                                          # gelu(x1) * sigmoid(x2)
        out = k_linear(g, w_out, b_out, out_dtype=x.dtype)                              # project_out
        ctx.save_for_backward(x2, w_in, b_in, wt, b_dw, w_out, b_out, h1, h2, g)
        ctx.meta = (B, L, d, H, W, F4, F2, dw.shape)
        return out.view(B, L, w_out.shape[0])

    @staticmethod
    def backward(ctx, dout):
        x2, w_in, b_in, wt, b_dw, w_out, b_out, h1, h2, g = ctx.saved_tensors
        B, L, d, H, W, F4, F2, dwshape = ctx.meta
        M = B * L
        do = dout.reshape(M, w_out.shape[0])
        do = do if do.is_contiguous() else do.contiguous()
        dg = k_linear_dx(do, w_out, out_dtype=g.dtype)
        dw_out, db_out = k_linear_dw(do, g, b_out is not None, w_out.data_ptr(), b_out.data_ptr() if b_out is not None else 0)
        dh2 = k_gate_bwd(dg, h2, F2)
        dh1, dwt, db_dw = k_dwconv_bwd(dh2, h1, wt, b_dw, B, H, W, F4, 3, lib.ACT_NONE, want_bias=b_dw is not None, chan_major=True)
        dx = k_linear_dx(dh1, w_in, out_dtype=x2.dtype) if ctx.needs_input_grad[0] else None
        dw_in, db_in = k_linear_dw(dh1, x2, b_in is not None, w_in.data_ptr(), b_in.data_ptr() if b_in is not None else 0)
        return (dx.view(B, L, d) if dx is not None else None, dw_in, db_in, dwt.view(dwshape), db_dw, dw_out, db_out, None, None)


def feedforward(x, w_in, b_in, dw, b_dw, w_out, b_out, H, W):
    _need_gpu(x)
    F4, F2 = w_in.shape[0], w_out.shape[1]
    if (x.dtype != torch.float32 or x.dim() != 3 or x.shape[1] != H * W or F4 != 2 * F2 or F2 % 4 or tuple(dw.shape) != (F4, 1, 3, 3)
            or w_in.shape[1] != x.shape[-1]):
        _unsupported("feedforward", f"is 1x1 d->4d, depthwise 3x3, gelu*sigmoid gate, 1x1 2d->d on fp32 (B, H*W, d) tokens with 8 | 4d, got "
                                    f"{x.dtype} {tuple(x.shape)}, weights {tuple(w_in.shape)}, {tuple(dw.shape)}, {tuple(w_out.shape)}")
    return FeedForwardFn.apply(x, w_in, b_in, dw, b_dw, w_out, b_out, H, W)


class LinCombFn(torch.autograd.Function):
    """y = gamma * (s0*x0 + s1*x1 + s2*x2): scalars are 1-element parameters, gamma a per-channel vector."""

    @staticmethod
    def forward(ctx, gamma, *args):
        n = len(args) // 2
        xs, ss = args[:n], args[n:]
        shp = xs[0].shape
        C = shp[-1]
        x2 = []
        for x in xs:
            t = x.reshape(-1, C)
            x2.append(t if t.stride(-1) == 1 else t.contiguous())
        M = x2[0].shape[0]
        y = torch.empty((M, C), dtype=xs[0].dtype, device=xs[0].device)
        ptr = [(None, 0)] * 3
        for i, t in enumerate(x2):
            ptr[i] = _rows(t)
        sp = [_p(s) for s in ss] + [None] * (3 - n)
        _need_gpu(x2[0])
        lib.call("adnm_lincomb_fwd", ptr[0][0], ptr[0][1], ptr[1][0], ptr[1][1], ptr[2][0], ptr[2][1], sp[0], sp[1], sp[2], _p(gamma),
                 y.data_ptr(), C, M, C, _dt(y), _stream())
        ctx.save_for_backward(gamma, *x2, *[s for s in ss if s is not None])
        ctx.meta = (n, shp, [s is not None for s in ss])
        return y.view(shp)

    @staticmethod
    def backward(ctx, dy):
        n, shp, has_s = ctx.meta
        saved = ctx.saved_tensors
        gamma, x2 = saved[0], saved[1:1 + n]
        it = iter(saved[1 + n:])
        ss = [next(it) if h else None for h in has_s]
        C = shp[-1]
        M = x2[0].shape[0]
        dev = x2[0].device
        g = dy.reshape(M, C)
        g = g if g.stride(-1) == 1 else g.contiguous()
        need_x = ctx.needs_input_grad[1:1 + n]
        dxs = [torch.empty((M, C), dtype=x2[0].dtype, device=dev) if need_x[i] else None for i in range(n)]
        # Block's beta1 / beta2 feed two mixes: the second claim of their slice lets the deferred fold ADD to it (fold segments: 0 = gamma,
        # 1 + i = scalar i) and hands autograd None — no flush, no separate add (GRADS.take_accumulating)
        dss, accmask = [], 0
        for i in range(n):
            if ss[i] is None:
                dss.append(None)
                continue
            t, acc = GRADS.take_accumulating(ss[i].data_ptr(), ss[i].shape, dev)
            dss.append(t)
            accmask |= (1 << (1 + i)) if acc else 0
        dgamma = grad_dst(gamma.data_ptr(), gamma.shape, dev) if gamma is not None else None
        nb = lib.query("adnm_lincomb_bwd_ws_bytes", M, C)
        ws = _ws(nb, dev)
        xp = [_rows(t) for t in x2] + [(None, 0)] * (3 - n)
        dxp = [(_p(t), C) for t in dxs] + [(None, 0)] * (3 - n)
        sp = [_p(s) for s in ss] + [None] * (3 - n)
        dsp = [_p(s) for s in dss] + [None] * (3 - n)
        pg, ldg = _rows(g)
        with FOLDS.defer(dev, ws) as deferred:
            if accmask:
                assert deferred, "an accumulating gradient claim needs the deferred fold queue"
                lib.load().adnm_foldq_accumulate_next(accmask)
            lib.call("adnm_lincomb_bwd", pg, ldg, xp[0][0], xp[0][1], xp[1][0], xp[1][1], xp[2][0], xp[2][1], sp[0], sp[1], sp[2], _p(gamma),
                     dxp[0][0], dxp[0][1], dxp[1][0], dxp[1][1], dxp[2][0], dxp[2][1], dsp[0], dsp[1], dsp[2], _p(dgamma), ws.data_ptr(), nb,
                     M, C, _dt(g), _stream())
        return (dgamma, *[d.view(shp) if d is not None else None for d in dxs],
                *[None if (accmask >> (1 + i)) & 1 else d for i, d in enumerate(dss)])


def lincomb(xs, scalars, gamma=None):
    """xs: 1-3 same-shape token tensors; scalars: matching list of 1-element parameters (or None = 1)."""
    C = xs[0].shape[-1]
    _need_gpu(xs[0])
    if gamma is None and C % 4 and xs[0].numel() % 4 == 0 and all(x.shape == xs[0].shape and x.is_contiguous() for x in xs):
        # a purely scalar mix is elementwise: run it on the flat arrays viewed as rows of 4 (PatchEmbed's 5-channel frames)
        shp = xs[0].shape
        return lincomb([x.view(-1, 4) for x in xs], scalars).view(shp)
    if C % 4 or C > 2048 or not 1 <= len(xs) <= 3 or any(s is not None and s.numel() != 1 for s in scalars):
        _unsupported("lincomb", f"needs 1-3 operands with 4 | C <= 2048 channels and 1-element scalars, got {len(xs)} operands, C={C}")
    return LinCombFn.apply(gamma, *xs, *scalars)


class MixNormFn(torch.autograd.Function):
    """(xn, x) with x = gamma * (s0*x0 + s1*x1) and xn = scale * norm(x) + shift: a residual mix and the NEXT sub-block's pre-norm
    (ADNMUNet.py:149-158) in one launch each way (csrc/mixnorm.hip).  x comes back as the second output, so the gradient the later
    residual mix sends to it is added inside this node's backward kernel — d x never exists as a tensor."""

    @staticmethod
    def forward(ctx, gamma, w, b, scale, shift, eps, mean, x0, x1, s0, s1):
        shp = x0.shape
        d = shp[-1]
        rows = lambda t: t.reshape(-1, d) if t.reshape(-1, d).stride(-1) == 1 else t.reshape(-1, d).contiguous()
        a2, b2 = rows(x0), rows(x1)
        _need_gpu(a2)
        M, dev = a2.shape[0], a2.device
        x = torch.empty((M, d), dtype=torch.float32, device=dev)
        xn = torch.empty((M, d), dtype=torch.float32, device=dev)
        mu = torch.empty(M, dtype=torch.float32, device=dev) if mean else None
        rstd = torch.empty(M, dtype=torch.float32, device=dev)
        (pa, lda), (pb, ldb) = _rows(a2), _rows(b2)
        lib.call("adnm_mixnorm_fwd", pa, lda, pb, ldb, _p(s0), _p(s1), _p(gamma), w.data_ptr(), _p(b), _p(scale), _p(shift), x.data_ptr(), d,
                 xn.data_ptr(), d, _p(mu), rstd.data_ptr(), M, d, float(eps), int(mean), _stream())
        ctx.save_for_backward(gamma, w, b, scale, a2, b2, s0, s1, mu, rstd)
        ctx.meta = (shp, mean, shift)
        ctx.set_materialize_grads(False)
        return xn.view(shp), x.view(shp)

    @staticmethod
    def backward(ctx, dxn, dres):
        gamma, w, b, scale, a2, b2, s0, s1, mu, rstd = ctx.saved_tensors
        shp, mean, shift = ctx.meta
        M, d = a2.shape
        dev = a2.device
        rows = lambda t: t.reshape(M, d) if t.reshape(M, d).stride(-1) == 1 else t.reshape(M, d).contiguous()
        if dxn is None:   # the normed output fed nothing that needs a gradient: a plain mix
            dxn = torch.zeros((M, d), dtype=torch.float32, device=dev)
        dxn = rows(dxn)
        dres = rows(dres) if dres is not None else None
        need = ctx.needs_input_grad
        dx0 = torch.empty((M, d), dtype=torch.float32, device=dev) if need[7] else None
        dx1 = torch.empty((M, d), dtype=torch.float32, device=dev) if need[8] else None
        dss, accmask = [], 0
        for i, sk in enumerate((s0, s1)):   # Block's beta1 / beta2 feed two mixes (see LinCombFn.backward)
            if sk is None:
                dss.append(None)
                continue
            t, acc = GRADS.take_accumulating(sk.data_ptr(), sk.shape, dev)
            dss.append(t)
            accmask |= (1 << (1 + i)) if acc else 0
        dgamma = grad_dst(gamma.data_ptr(), gamma.shape, dev) if gamma is not None else None
        dw = grad_dst(w.data_ptr(), (d,), dev)
        db = grad_dst(b.data_ptr(), (d,), dev) if b is not None else None
        affine = scale is not None or shift is not None
        dsc = grad_dst(scale.data_ptr() if scale is not None else 0, (), dev) if affine else None
        dsh = grad_dst(shift.data_ptr() if shift is not None else 0, (), dev) if affine else None
        nb = lib.query("adnm_mixnorm_bwd_ws_bytes", M, d)
        ws = _ws(nb, dev)
        (pg, ldg), (pa, lda), (pb, ldb) = _rows(dxn), _rows(a2), _rows(b2)
        pr, ldr = _rows(dres) if dres is not None else (None, 0)
        with FOLDS.defer(dev, ws) as deferred:
            if accmask:
                assert deferred, "an accumulating gradient claim needs the deferred fold queue"
                lib.load().adnm_foldq_accumulate_next(accmask)
            lib.call("adnm_mixnorm_bwd", pg, ldg, pr, ldr, pa, lda, pb, ldb, _p(s0), _p(s1), _p(gamma), w.data_ptr(), _p(b), _p(scale), _p(mu),
                     rstd.data_ptr(), _p(dx0), d, _p(dx1), d, _p(dss[0]), _p(dss[1]), _p(dgamma), dw.data_ptr(), _p(db), _p(dsc), _p(dsh),
                     ws.data_ptr(), nb, M, d, int(mean), _stream())
        ds = [None if (accmask >> (1 + i)) & 1 else t for i, t in enumerate(dss)]
        return (dgamma, dw, db, dsc if scale is not None else None, dsh if shift is not None else None, None, None,
                dx0.view(shp) if dx0 is not None else None, dx1.view(shp) if dx1 is not None else None, ds[0], ds[1])


def mixnorm(xs, scalars, gamma, w, b=None, scale=None, shift=None, eps=1e-5, mean=True):
    """-> (scale * norm(x) + shift, x) with x = gamma * (s0*xs[0] + s1*xs[1]); see MixNormFn."""
    _need_gpu(xs[0])
    d = xs[0].shape[-1]
    if (len(xs) != 2 or xs[0].shape != xs[1].shape or any(x.dtype != torch.float32 for x in xs) or d % 4 or d > 1024
            or any(sk is not None and sk.numel() != 1 for sk in scalars)):
        _unsupported("mixnorm", f"mixes two same-shape fp32 token tensors with 4 | d <= 1024 channels and 1-element scalars, got "
                                f"{[tuple(x.shape) for x in xs]} {[x.dtype for x in xs]}")
    return MixNormFn.apply(gamma, w, b, scale, shift, eps, mean, xs[0], xs[1], scalars[0], scalars[1])


class AdnPrepFn(torch.autograd.Function):
    """Reference-layout ADN-SSD parameters -> the kernel-layout tensors ADNMixerFn consumes (one HIP launch each way;
    see csrc/paramprep.hip).  Argument order = the `params[15]` table of include/adnm_hip.h.
    Returns (w_in, taps, ln_w, ln_b, w_out): `taps` (9, 2di+2gN) = [czw | cw], the tap-major 3x3 taps of conv2d_z and of every xBC
    channel side by side, so that z and xBC — adjacent column ranges of in_proj's output — go through ONE depthwise launch."""

    @staticmethod
    def forward(ctx, dm, di, gn, P, *params):
        params = [p.contiguous() for p in params]
        dev = params[0].device
        _need_gpu(params[0])
        nh, cx = di // P, di + 2 * gn
        f = lambda *s: torch.empty(s, dtype=torch.float32, device=dev)
        w_in, taps, ln_w, ln_b, w_out = f(2 * di + 2 * gn + nh, dm), f(9, di + cx), f(di), f(di), f(dm, 2 * di)
        outs = [w_in, taps[:, di:], taps[:, :di], ln_w, ln_b, w_out]   # the ABI's table: {w_in, cw, czw, ln_w, ln_b, w_out}
        lib.call("adnm_adnprep_fwd", lib.ptr_table(params), lib.ptr_table(outs), dm, di, gn, P, di + cx, _stream())
        ctx.save_for_backward(*params)
        ctx.dims = (dm, di, gn, P)
        return w_in, taps, ln_w, ln_b, w_out

    @staticmethod
    def backward(ctx, g_w_in, g_taps, g_ln_w, g_ln_b, g_w_out):
        params = ctx.saved_tensors
        dm, di, gn, P = ctx.dims
        cx = di + 2 * gn
        with FOLDS.late(params[0].device):   # the incoming gradients are (deferred) fold results of the mixer's backward
            g_w_in, g_taps, g_ln_w, g_ln_b, g_w_out = (g.contiguous() for g in (g_w_in, g_taps, g_ln_w, g_ln_b, g_w_out))
            gouts = [g_w_in, g_taps[:, di:], g_taps[:, :di], g_ln_w, g_ln_b, g_w_out]
            dparams = [grad_dst(p.data_ptr(), p.shape, p.device, p.dtype) for p in params]
            nb = lib.query("adnm_adnprep_bwd_ws_bytes")
            ws = _ws(nb, params[0].device)
            with FOLDS.defer(params[0].device, ws, g_w_in, g_taps, g_ln_w, g_ln_b, g_w_out):
                lib.call("adnm_adnprep_bwd", lib.ptr_table(params), lib.ptr_table(gouts), lib.ptr_table(dparams), dm, di, gn, P, di + cx,
                         ws.data_ptr(), nb, _stream())
        return (None, None, None, None, *dparams)


def adn_prep(dm, di, gn, P, params):
    return AdnPrepFn.apply(dm, di, gn, P, *params)


class WtPrepFn(torch.autograd.Function):
    """WTConv2d parameters -> tap-major taps with the per-channel scales folded in and channels zero-padded to Cp.
    args: C, Cp, K, levels, bias|None, then (1+levels) conv weights, then (1+levels) scale tensors."""

    @staticmethod
    def forward(ctx, C, Cp, K, levels, bias, *ws_):
        n = 1 + levels
        w = [t.contiguous() for t in ws_[:n]]
        s = [t.contiguous() for t in ws_[n:]]
        dev = w[0].device
        _need_gpu(w[0])
        taps = [torch.empty((K * K, Cp if k == 0 else 4 * Cp), dtype=torch.float32, device=dev) for k in range(n)]
        bias_t = torch.empty(Cp, dtype=torch.float32, device=dev) if bias is not None else None
        lib.call("adnm_wtprep_fwd", lib.ptr_table(w), lib.ptr_table(s), _p(bias), lib.ptr_table(taps), _p(bias_t), C, Cp, K, levels, _stream())
        ctx.save_for_backward(bias, *w, *s)
        ctx.dims = (C, Cp, K, levels)
        return (bias_t, *taps)

    @staticmethod
    def backward(ctx, gbias_t, *gtaps):
        C, Cp, K, levels = ctx.dims
        n = 1 + levels
        saved = ctx.saved_tensors
        bias, w, s = saved[0], saved[1:1 + n], saved[1 + n:]
        with FOLDS.late(w[0].device):   # the incoming tap gradients are (deferred) fold results of the depthwise weight-gradient kernels
            gtaps = [g.contiguous() for g in gtaps]
            dw = [torch.empty_like(t) for t in w]
            ds = [torch.empty_like(t) for t in s]
            dbias = torch.empty_like(bias) if bias is not None else None
            if bias is not None:
                gbias_t = gbias_t.contiguous() if gbias_t is not None else torch.zeros(Cp, dtype=torch.float32, device=bias.device)
            lib.call("adnm_wtprep_bwd", lib.ptr_table(w), lib.ptr_table(s), _p(bias), lib.ptr_table(gtaps), _p(gbias_t) if bias is not None else None,
                     lib.ptr_table(dw), lib.ptr_table(ds), _p(dbias), C, Cp, K, levels, _stream())
            SIDE.keep(w[0].device, gtaps, gbias_t)
        return (None, None, None, None, dbias, *dw, *ds)


def wt_prep(C, Cp, K, levels, bias, weights, scales):
    out = WtPrepFn.apply(C, Cp, K, levels, bias, *weights, *scales)
    return out[0], out[1], list(out[2:])


def narrow_weight_mode():
    """1 / 2: the big matrices the mixer prep emits are written as bf16 / scaled e4m3 copies only (the weight-streaming GEMMs read those:
    adnm_skgemm b_dtype); 0: fp32 (exact mode, calibration passes, ADNM_NARROW_WEIGHTS=0)"""
    if QUANT.calibrating or os.environ.get("ADNM_NARROW_WEIGHTS", "1") == "0":
        return 0
    return MFMA_PREC[0] if MFMA_PREC[0] in (1, 2) else 0


class AdnPrepMultiFn(torch.autograd.Function):
    """AdnPrepFn for every mixer of a model stage in ONE launch each way (include/adnm_hip.h: adnm_adnprep_*_multi).
    args: dims = [(dm, di, gn, P), ...], then 15 parameters per mixer; returns 9 entries per mixer:
    (w_in, taps, ln_w, ln_b, w_out, w_in_n, s_in, w_out_n, s_out).  In the narrow modes a matrix too large for the tall-skinny kernel
    (N K > 8192: it goes through the weight-streaming short GEMMs) is written ONLY as its bf16 / scaled-e4m3 copy w_*_n (s_*: the fp8
    scale, a 1-element tensor); the fp32 entry of that name is then a storage-less (N, K) handle that carries shape and autograd edge
    (its gradient arrives as an ordinary fp32 matrix).  fp8 copies need the weight records of a FlatTrainer; without them (plain autograd
    use) the fp32 matrices are written as before."""

    @staticmethod
    def forward(ctx, dims, *params):
        params = [p.contiguous() for p in params]
        n = len(dims)
        dev = params[0].device
        _need_gpu(params[0])
        f = lambda *s: torch.empty(s, dtype=torch.float32, device=dev)
        mode = narrow_weight_mode()
        ndt = {1: torch.bfloat16, 2: torch.uint8}.get(mode)
        handle = lambda N, K: torch.empty(1, dtype=torch.float32, device=dev).expand(N, K)
        outs, table, dflat, ntab, nondiff = [], [], [], [], []
        any_narrow = False
        for i, (dm, di, gn, P) in enumerate(dims):
            nh, cx = di // P, di + 2 * gn
            n_in, n_out = 2 * di + 2 * gn + nh, dm
            taps, ln_w, ln_b = f(9, di + cx), f(di), f(di)
            w_in_p, w_out_p = params[15 * i], params[15 * i + 13]
            s_in = s_out_src = None
            ok = mode != 0
            if mode == 2:   # the fp8 scales come from the weight records the optimiser pass maintains
                a, b = SHADOWS.lookup(w_in_p, 2), SHADOWS.lookup(w_out_p, 2)
                ok = a is not None and b is not None
                if ok:
                    s_in, s_out_src = a[2], b[2]
            big_in, big_out = ok and n_in * dm > 8192, ok and n_out * 2 * di > 8192
            w_in_n = torch.empty((n_in, dm), dtype=ndt, device=dev) if big_in else None
            w_out_n = torch.empty((n_out, 2 * di), dtype=ndt, device=dev) if big_out else None
            w_in = handle(n_in, dm) if big_in else f(n_in, dm)
            w_out = handle(n_out, 2 * di) if big_out else f(n_out, 2 * di)
            s_out = torch.empty(1, dtype=torch.float32, device=dev) if (big_out and mode == 2) else None
            any_narrow = any_narrow or big_in or big_out
            outs += [w_in, taps, ln_w, ln_b, w_out, w_in_n, s_in if big_in else None, w_out_n, s_out]
            nondiff += [t for t in (w_in_n, w_out_n, s_out) if t is not None]
            table += [None if big_in else w_in, taps[:, di:], taps[:, :di], ln_w, ln_b, None if big_out else w_out]
            ntab += [w_in_n, w_out_n, s_in if big_in else None, s_out_src if big_out else None, s_out]
            dflat += [dm, di, gn, P, di + cx]
        lib.call("adnm_adnprep_fwd_multi", n, lib.ptr_table(params), lib.ptr_table(table), lib.i64_table(dflat),
                 lib.ptr_table(ntab) if any_narrow else None, mode if any_narrow else 0, _stream())
        ctx.save_for_backward(*params)
        ctx.dims = dims
        ctx.mark_non_differentiable(*nondiff)
        ctx.set_materialize_grads(False)
        return tuple(outs)

    @staticmethod
    def backward(ctx, *g):
        params = ctx.saved_tensors
        dims = ctx.dims
        n = len(dims)
        dev = params[0].device
        with FOLDS.late(dev):   # the incoming gradients are (deferred) fold results of the mixers' backward
            gtab, dflat, keep = [], [], []
            for i, (dm, di, gn, P) in enumerate(dims):
                nh, cx = di // P, di + 2 * gn
                shapes = [(2 * di + 2 * gn + nh, dm), (9, di + cx), (di,), (di,), (dm, 2 * di)]
                gi = [t.contiguous() if t is not None else torch.zeros(shp, dtype=torch.float32, device=dev) for t, shp in zip(g[9 * i:9 * i + 5], shapes)]
                keep += gi
                g_w_in, g_taps, g_ln_w, g_ln_b, g_w_out = gi
                gtab += [g_w_in, g_taps[:, di:], g_taps[:, :di], g_ln_w, g_ln_b, g_w_out]
                dflat += [dm, di, gn, P, 2 * di + 2 * gn]
            dparams = [grad_dst(p.data_ptr(), p.shape, dev, p.dtype) for p in params]
            nb = lib.query("adnm_adnprep_bwd_multi_ws_bytes", n)
            ws = _ws(nb, dev)
            with FOLDS.defer(dev, ws, *keep):
                lib.call("adnm_adnprep_bwd_multi", n, lib.ptr_table(params), lib.ptr_table(gtab), lib.ptr_table(dparams), lib.i64_table(dflat),
                         ws.data_ptr(), nb, _stream())
        return (None, *dparams)


class WtPrepMultiFn(torch.autograd.Function):
    """WtPrepFn for every WTConv2d of a model stage in ONE launch each way.  args: dims = [(C, Cp, K, levels, has_bias), ...], then per
    module: bias (if has_bias), (1 + levels) conv weights, (1 + levels) scales.  Returns per module: bias_t (if has_bias), (1 + levels) taps."""

    @staticmethod
    def forward(ctx, dims, *ts):
        dev = ts[0].device
        _need_gpu(ts[0])
        w, s, bias, taps, bias_t, dflat, outs = [], [], [], [], [], [], []
        it = iter(ts)
        saved = []
        for C, Cp, K, levels, has_bias in dims:
            n1 = 1 + levels
            b = next(it).contiguous() if has_bias else None
            ws_ = [next(it).contiguous() for _ in range(n1)]
            ss = [next(it).contiguous() for _ in range(n1)]
            tp = [torch.empty((K * K, Cp if k == 0 else 4 * Cp), dtype=torch.float32, device=dev) for k in range(n1)]
            bt = torch.empty(Cp, dtype=torch.float32, device=dev) if has_bias else None
            pad = [None] * (5 - n1)
            w += ws_ + pad; s += ss + pad; taps += tp + pad
            bias.append(b); bias_t.append(bt)
            dflat += [C, Cp, K, levels]
            outs += ([bt] if has_bias else []) + tp
            saved += ([b] if has_bias else []) + ws_ + ss
        lib.call("adnm_wtprep_fwd_multi", len(dims), lib.ptr_table(w), lib.ptr_table(s), lib.ptr_table(bias), lib.ptr_table(taps), lib.ptr_table(bias_t),
                 lib.i64_table(dflat), _stream())
        ctx.save_for_backward(*saved)
        ctx.dims = dims
        return tuple(outs)

    @staticmethod
    def backward(ctx, *g):
        dims = ctx.dims
        saved = ctx.saved_tensors
        dev = saved[0].device
        with FOLDS.late(dev):   # the incoming tap gradients are (deferred) fold results of the depthwise weight-gradient kernels
            w, s, bias, gtaps, gbias_t, dw, ds, dbias, dflat, grads = [], [], [], [], [], [], [], [], [], []
            si, gi = iter(saved), iter(g)
            for C, Cp, K, levels, has_bias in dims:
                n1 = 1 + levels
                b = next(si) if has_bias else None
                ws_ = [next(si) for _ in range(n1)]
                ss = [next(si) for _ in range(n1)]
                gb = next(gi).contiguous() if has_bias else None
                gt = [next(gi).contiguous() for _ in range(n1)]
                dws, dss = [torch.empty_like(t) for t in ws_], [torch.empty_like(t) for t in ss]
                db = torch.empty_like(b) if has_bias else None
                pad = [None] * (5 - n1)
                w += list(ws_) + pad; s += list(ss) + pad; gtaps += gt + pad; dw += dws + pad; ds += dss + pad
                bias.append(b); gbias_t.append(gb); dbias.append(db)
                dflat += [C, Cp, K, levels]
                grads += ([db] if has_bias else []) + dws + dss
            lib.call("adnm_wtprep_bwd_multi", len(dims), lib.ptr_table(w), lib.ptr_table(s), lib.ptr_table(bias), lib.ptr_table(gtaps), lib.ptr_table(gbias_t),
                     lib.ptr_table(dw), lib.ptr_table(ds), lib.ptr_table(dbias), lib.i64_table(dflat), _stream())
            SIDE.keep(dev, gtaps, gbias_t)
        return (None, *grads)


def prep_group(*roots):
    """Prepare the kernel-layout parameters of EVERY ADN-SSD mixer and WTConv2d under `roots` in one launch per kind (instead of one per
    module): each module finds its tensors in `_adnm_prepped` at its next forward and uses them once.  The modules define
    adnm_prep_kind ("adn" / "wt") and adnm_prep_args().  Called by VisionMamba.forward_stage1 / forward_stage2 (one group per model
    stage, so a staged backward still finds all of a group's gradients inside one stage)."""
    adn, wt = [], []
    for r in roots:
        for m in r.modules():
            kind = getattr(m, "adnm_prep_kind", None)
            if kind == "adn" and m.adnm_prep_ready():
                adn.append(m)
            elif kind == "wt":
                wt.append(m)
    for m in adn[:1] + wt[:1]:   # the product path has no CPU fallback: a model left on the CPU raises here, not somewhere inside a module
        _need_gpu(m.in_proj.weight if m.adnm_prep_kind == "adn" else m.base_conv.weight)
    if adn:
        dims, params = [], []
        for m in adn:
            d, p = m.adnm_prep_args()
            dims.append(d)
            params += p
        out = AdnPrepMultiFn.apply(dims, *params)
        for i, m in enumerate(adn):
            m.__dict__["_adnm_prepped"] = out[9 * i:9 * i + 9]   # (w_in, taps, ln_w, ln_b, w_out, w_in_n, s_in, w_out_n, s_out)
    if wt:
        dims, ts = [], []
        for m in wt:
            d, t = m.adnm_prep_args()
            dims.append(d)
            ts += t
        out = list(WtPrepMultiFn.apply(dims, *ts))
        for m, (C, Cp, K, levels, has_bias) in zip(wt, dims):
            bias_t = out.pop(0) if has_bias else None
            taps = [out.pop(0) for _ in range(1 + levels)]
            m.__dict__["_adnm_prepped"] = (Cp, taps[0], bias_t, taps[1:])


class MaxPoolFn(torch.autograd.Function):
    """nn.MaxPool2d on (B, H*W, C) tokens: stride == kernel (DownSample) or stride 1 'same' (EncoderToDecoder).  tap: also hand back x as
    an autograd alias for its OTHER consumer (an encoder stage's output is pooled and kept as a skip tensor): that consumer's gradient is
    added inside the pool's backward kernel instead of by a separate autograd add."""

    @staticmethod
    def forward(ctx, x, H, W, kh, kw, stride, tap):
        B, L, C = x.shape
        xc = x.contiguous()
        _need_gpu(xc)
        Ho, Wo = (H, W) if stride == 1 else (H // stride, W // stride)
        y = torch.empty((B, Ho * Wo, C), dtype=x.dtype, device=x.device)
        lib.call("adnm_maxpool_fwd", xc.data_ptr(), y.data_ptr(), B, H, W, C, kh, kw, stride, _dt(xc), _stream())
        ctx.save_for_backward(xc)
        ctx.dims = (B, H, W, C, kh, kw, stride)
        ctx.set_materialize_grads(False)
        return (y, x) if tap else y

    @staticmethod
    def backward(ctx, dy, dalias=None):
        (x,) = ctx.saved_tensors
        B, H, W, C, kh, kw, stride = ctx.dims
        if dy is None:
            return dalias, None, None, None, None, None, None
        dy = dy.contiguous()
        if dalias is not None:
            dalias = dalias.reshape(x.shape).contiguous()
        dx = torch.empty_like(x)
        lib.call("adnm_maxpool_bwd", dy.data_ptr(), x.data_ptr(), _p(dalias), dx.data_ptr(), B, H, W, C, kh, kw, stride, _dt(x), _stream())
        return dx, None, None, None, None, None, None


def maxpool(x, H, W, kh, kw, stride, tap=False):
    """tap=True: -> (pooled, alias of x) — hand the alias to x's other consumer (see MaxPoolFn)."""
    return MaxPoolFn.apply(x, H, W, kh, kw, stride, tap)


# ------------------------------------------------------------------------------------------- Linear / 1x1 GEMMs (K6, K6b)
# Static routing, the same in every process and on every rank: the tall-skinny kernel (csrc/tsgemm.hip) takes the full-resolution
# projections (>= 2048 token rows, weight <= 8192 elements), the short-GEMM kernel (csrc/skgemm.hip) everything else.  There is
# no library GEMM behind these: a shape neither kernel takes raises.
TS_MIN_ROWS = int(os.environ.get("ADNM_TS_MIN_ROWS", "32768"))   # the environment override is a measurement aid (tools/kbench_ts.py)
SK_NT, SK_NN, SK_TN = 0, 1, 2


def ts_ok_nt(M, N, K, x):
    return (x.is_cuda and x.dtype in _DT and M >= TS_MIN_ROWS and x.stride(0) % 4 == 0 and lib.query("adnm_tsgemm_supported", M, N, K) == 1)


def ts_ok_tn(M, N, K, x):
    return (x.is_cuda and x.dtype in _DT and M >= TS_MIN_ROWS and lib.query("adnm_tsgemm_tn_supported", M, N, K) == 1)


def low_storage(M, shapes):
    """torch.bfloat16 when the wide INTERNAL tensors of a fused node (ADNMixerFn, FeedForwardFn) over M token rows may be kept in bf16,
    else torch.float32: the matrix-core precision of these GEMMs is bf16 (so the values are rounded to bf16 on their way into the MFMA
    anyway), every GEMM of the node — shapes = [(N, K), ...] — runs on the tall-skinny kernel (the only GEMM kernel with bf16 token
    I/O: the full-resolution level), and ADNM_BF16_STORAGE=1.  The node's inputs, outputs and parameter gradients stay fp32.
    OFF by default: measured on MI355X at config 2 (profiles/r03_bf16_storage_ab.txt) the step is 0.2 ms SLOWER with it (9.78 vs 9.56 ms) —
    the full-resolution kernels are bound by load issue / latency, not by HBM bytes: halving the bytes leaves the stencils, K1 and the
    forward GEMMs where they were (+-0.02 ms each) and the element-per-lane loaders of the weight-gradient kernels (tsgemm_tn,
    dwconv_wgrad3_roll) get slower on 2-byte elements (+0.18 / +0.13 ms).  The paths stay (tested) for kernels that move 16 B of bf16 per lane."""
    if MFMA_PREC[0] == 0 or M < TS_MIN_ROWS or os.environ.get("ADNM_BF16_STORAGE", "0") != "1":
        return torch.float32
    if MFMA_PREC[0] == 2 and M <= QUANT.max_rows:   # fp8 operands for this many rows: fp32 storage, quantised on load
        return torch.float32
    for N, K in shapes:
        if lib.query("adnm_tsgemm_supported", M, N, K) != 1 or lib.query("adnm_tsgemm_tn_supported", M, N, K) != 1:
            return torch.float32
    return torch.bfloat16


def _sk_operand(t, what):
    """skgemm reads 16-byte vectors along the contiguous axis of a row view."""
    if not (t.is_cuda and t.dtype == torch.float32):
        raise RuntimeError(f"adnm_hip linear: {what} must be an fp32 GPU tensor, got {t.dtype} on {t.device}")
    if t.stride(-1) != 1 or t.stride(0) % 4 or t.data_ptr() % 16:
        t = t.contiguous()
        if t.stride(0) % 4 or t.data_ptr() % 16:
            raise RuntimeError(f"adnm_hip linear: {what} of shape {tuple(t.shape)} cannot be 16-byte aligned per row (last dim must be a multiple of 4)")
    return t


def _skgemm(op, a, b, bias, c, dbias, M, N, K, defer=False, side=False, q=None, role="f", narrow=None):
    """narrow = (pointer, b_dtype, scale tensor | None): read the weight operand from its narrow shadow (ShadowRegistry / the mixer prep's
    narrow copies) instead of the fp32 tensor b, which then only provides the shape and row stride"""
    if lib.query("adnm_skgemm_supported", op, M, N, K) != 1:
        raise RuntimeError(f"adnm_hip linear: no kernel takes op={('NT', 'NN', 'TN')[op]} M={M} N={N} K={K} "
                           "(every op needs N % 4 == 0 and K % 4 == 0)")
    nb = lib.query("adnm_skgemm_ws_bytes", op, M, N, K)
    ucp, ucn = None, 0
    if op != SK_TN and nb > 16:
        # NT / NN split over workgroups: the slabs are combined inside the launch — arrival counters + uncached slab space of this stream
        # (or this capture scope: SplitWorkspaces); an un-scoped capture takes a torch workspace [zeroed counters | slabs], fenced protocol
        reg = SPLITWS.take(a.device, nb, lib.query("adnm_skgemm_counter_bytes", op, M, N, K))
        if reg is None:
            ws = _ws((nb + 255) // 256 * 256 + 256, a.device)
            ws.zero_()
            off = (-ws.data_ptr()) % 256
            wsp, wsn = ws.data_ptr() + off, nb
        else:
            ws = reg
            wsp, wsn = reg.counters.data_ptr(), reg.counters.numel() * 4
            if os.environ.get("ADNM_SK_UC_SLABS", "1") == "0":   # measurement aid / test: the fenced protocol, slabs behind the counters
                ws = (reg, _ws(nb + 256, a.device))
                ws[1].zero_()
                wsp, wsn = ws[1].data_ptr() + (-ws[1].data_ptr()) % 256, nb
            else:
                ucp, ucn = reg.slabs.ptr, reg.slabs.nbytes
    else:
        ws = _ws(nb, a.device)
        wsp, wsn = ws.data_ptr(), nb
    # only the weight-gradient op (TN) may wait for its split-K fold, and only it leaves the critical path for the side stream
    pc, ldc, pdb = c.data_ptr(), c.stride(0), _p(dbias)   # (the outputs are not captured: see SideStreams)
    if op == SK_TN:   # the weight gradient: bf16 operands in the fp8 configuration, no record
        prec, qp = (1 if MFMA_PREC[0] == 2 or QUANT.calibrating else MFMA_PREC[0]), None
    else:
        prec, qp = _gemm_prec(q, role)
    bp, bdt, bsc = b.data_ptr(), 0, None
    if narrow is None and op != SK_TN:
        narrow = SHADOWS.lookup(b, prec)   # a parameter of a FlatTrainer: its bf16 / fp8 shadow, kept current by the optimiser pass
    if narrow is not None and op != SK_TN:
        if narrow[1] == 1 and prec == 1:
            bp, bdt = narrow[0], 1
        elif narrow[1] == 2 and prec in (2, 3):
            bp, bdt, bsc = narrow[0], 2, narrow[2]
    ldb = b.stride(0)
    call = lambda: lib.call("adnm_skgemm", op, a.data_ptr(), a.stride(0), bp, ldb, bdt, _p(bsc), _p(bias), pc, ldc, pdb,
                            wsp, wsn, ucp, ucn, M, N, K, prec, qp, _stream())
    if side:   # (a, b are kept with the workspace: under a bound leaf queue the launch itself waits for the grouped flush)
        SIDE.submit(a.device, (a, b), FOLDS.defer(a.device, ws, a, b) if defer else _NODEFER, call)
    else:
        with FOLDS.defer(a.device, ws) if defer else _NODEFER:
            call()


def _out_view_ok(out):
    return out.stride(-1) == 1 and out.stride(0) % 4 == 0 and out.data_ptr() % 16 == 0


class _NarrowW:
    """stand-in for a weight that exists only as a narrow copy (the mixer prep's bf16 / fp8 matrices): shape, row stride, device"""

    def __init__(self, narrow, N, K, device):
        self.narrow, self.shape, self.device = narrow, (N, K), device

    def data_ptr(self):
        return self.narrow[0]

    def stride(self, i):
        return (self.shape[1], 1)[i]


def k_linear(x2, w, bias, out=None, qkey=None, role="f", out_dtype=None, narrow=None):
    """Y = X W^T (+bias) for row views X (M,K) [stride (ld,1)], W (N,K) contiguous.
    qkey: stable identity of the weight for the fp8 call-site record (default: its data_ptr — right for parameters, wrong for
    per-step temporaries such as the mixer's prepared weights, whose callers pass the parameter's); role "f": X are activations,
    "g": X is an output gradient (this GEMM computes an input gradient)."""
    M, K = x2.shape
    N = w.shape[0]
    _need_gpu(x2)
    q = QUANT.record(x2.device, w.data_ptr() if qkey is None else qkey, role + "nt", M)
    if ts_ok_nt(M, N, K, x2):
        y = out if out is not None else torch.empty((M, N), dtype=out_dtype or x2.dtype, device=x2.device)
        prec, qp = _gemm_prec(q, role)
        lib.call("adnm_tsgemm_nt", x2.data_ptr(), x2.stride(0), w.data_ptr(), K, 1, _p(bias), y.data_ptr(), y.stride(0), M, N, K, prec, qp, _dt(x2),
                 _dt(y), _stream())
        return y
    x2 = _sk_operand(x2, "input")
    w = _sk_operand(w, "weight") if narrow is None else _NarrowW(narrow, N, K, x2.device)
    y = out if out is not None and _out_view_ok(out) else torch.empty((M, N), dtype=x2.dtype, device=x2.device)
    _skgemm(SK_NT, x2, w, bias, y, None, M, N, K, q=q, role=role, narrow=narrow)
    if out is not None and y is not out:
        out.copy_(y)
        return out
    return y


def k_linear_dx(dy2, w, out=None, qkey=None, role="g", out_dtype=None, narrow=None):
    """dX = dY W for dY (M,N) row view, W (N,K) contiguous.  (qkey / role: see k_linear; the transposed conv's FORWARD is this product
    with activations as the first operand, role "f".)"""
    M, N = dy2.shape
    K = w.shape[1]
    q = QUANT.record(dy2.device, w.data_ptr() if qkey is None else qkey, role + "nn", M)
    if ts_ok_nt(M, K, N, dy2):
        dx = out if out is not None else torch.empty((M, K), dtype=out_dtype or dy2.dtype, device=dy2.device)
        prec, qp = _gemm_prec(q, role)
        lib.call("adnm_tsgemm_nt", dy2.data_ptr(), dy2.stride(0), w.data_ptr(), 1, K, None, dx.data_ptr(), dx.stride(0), M, K, N, prec, qp, _dt(dy2),
                 _dt(dx), _stream())
        return dx
    dy2 = _sk_operand(dy2, "output gradient")
    w = _sk_operand(w, "weight") if narrow is None else _NarrowW(narrow, N, K, dy2.device)
    dx = out if out is not None and _out_view_ok(out) else torch.empty((M, K), dtype=dy2.dtype, device=dy2.device)
    _skgemm(SK_NN, dy2, w, None, dx, None, M, N, K, q=q, role=role, narrow=narrow)
    if out is not None and dx is not out:
        out.copy_(dx)
        return out
    return dx


def colsum(t, out=None, defer=False):
    """sum over the rows of a contiguous 2-D fp32 matrix (a bias gradient) with the deterministic fold kernel."""
    _need_gpu(t)
    if not (t.dtype == torch.float32 and t.dim() == 2):
        raise RuntimeError(f"adnm_hip colsum: needs a 2-D fp32 tensor, got {t.dtype} {tuple(t.shape)}")
    t = t if t.is_contiguous() else t.contiguous()
    if out is None:
        out = torch.empty(t.shape[1], dtype=torch.float32, device=t.device)
    nb = lib.query("adnm_colsum_ws_bytes", t.shape[0], t.shape[1])
    ws = _ws(nb, t.device)
    with FOLDS.defer(t.device, ws, t) if defer else _NODEFER:
        lib.call("adnm_colsum", t.data_ptr(), out.data_ptr(), t.shape[0], t.shape[1], ws.data_ptr(), nb, _stream())
    return out


def k_linear_dw(dy2, x2, want_bias, w_ptr=0, b_ptr=0, dw_out=None):
    """dW = dY^T X (N,K), dbias = column sums of dY.  w_ptr / b_ptr: data_ptr of the parameters (gradient-destination lookup);
    dw_out: an explicit contiguous (N,K) destination instead."""
    M, N = dy2.shape
    K = x2.shape[1]
    dev = x2.device
    dw = dw_out if dw_out is not None else grad_dst(w_ptr, (N, K), dev)
    db = grad_dst(b_ptr, (N,), dev) if want_bias else None
    if ts_ok_tn(M, N, K, x2):
        nb = lib.query("adnm_tsgemm_tn_ws_bytes", M, N, K)
        ws = _ws(nb, dev)
        pdw, pdb = dw.data_ptr(), _p(db)
        dty, dtx = _dt(dy2), _dt(x2)
        # (the operands stay referenced until the flush: under the leaf queue the GEMM itself is deferred, not just its fold)
        SIDE.submit(dev, (dy2, x2), FOLDS.defer(dev, ws, dy2, x2), lambda: lib.call(
            "adnm_tsgemm_tn", dy2.data_ptr(), dy2.stride(0), x2.data_ptr(), x2.stride(0), pdw, pdb, ws.data_ptr(), nb, M, N, K, dty, dtx, _stream()))
        return dw, db
    dy2, x2 = _sk_operand(dy2, "output gradient"), _sk_operand(x2, "input")
    _skgemm(SK_TN, dy2, x2, None, dw, db, M, N, K, defer=True, side=True)
    return dw, db


class LinearFn(torch.autograd.Function):
    """nn.Linear / 1x1 conv on tokens."""

    @staticmethod
    def forward(ctx, x, w, bias, qkey=None):
        shp = x.shape
        K = shp[-1]
        x2 = x.reshape(-1, K)
        x2 = x2 if x2.stride(-1) == 1 else x2.contiguous()
        w = w.contiguous()
        ctx.qkey = w.data_ptr() if qkey is None else qkey
        y = k_linear(x2, w, bias, qkey=ctx.qkey)
        ctx.save_for_backward(x2, w)
        ctx.has_bias = bias is not None
        ctx.ptrs = (w.data_ptr(), bias.data_ptr() if bias is not None else 0)
        ctx.shp = shp
        return y.view(*shp[:-1], w.shape[0])

    @staticmethod
    def backward(ctx, dy):
        x2, w = ctx.saved_tensors
        N = w.shape[0]
        dy2 = dy.reshape(-1, N)
        dy2 = dy2 if dy2.stride(-1) == 1 else dy2.contiguous()
        dx = k_linear_dx(dy2, w, qkey=ctx.qkey).view(ctx.shp) if ctx.needs_input_grad[0] else None
        dw, db = k_linear_dw(dy2, x2, ctx.has_bias, *ctx.ptrs)
        return dx, dw, db, None


def linear(x, w, bias=None):
    """nn.Linear on tokens: the MFMA kernels (tall-skinny / short GEMM); raises for anything they do not take."""
    _need_gpu(x)
    if x.dtype != torch.float32 or w.dim() != 2:
        raise RuntimeError(f"adnm_hip linear: needs fp32 tokens and a 2-D weight, got {x.dtype}, weight {tuple(w.shape)}")
    N = w.shape[0]
    if N % 4:   # the GEMM kernels move 16-byte vectors along N: run them on the weight padded with zero rows (a frame count like 6)
        pad = 4 - N % 4
        key = w.data_ptr()   # the padded copy is a per-step temporary: the parameter identifies the call site
        w = torch.nn.functional.pad(w, (0, 0, 0, pad))
        bias = torch.nn.functional.pad(bias, (0, pad)) if bias is not None else None
        return LinearFn.apply(x, w, bias, key)[..., :N]
    return LinearFn.apply(x, w, bias)


# ------------------------------------------------------------------------------------------- stand-alone activations
class ActFn(torch.autograd.Function):
    """GELU (exact) / SiLU over fp32 tokens (csrc/activation.hip)."""

    @staticmethod
    def forward(ctx, x, act):
        _need_gpu(x)
        if x.dtype != torch.float32 or x.numel() % 4:
            _unsupported("act", f"needs fp32 tokens with a multiple of 4 elements, got {x.dtype} x {x.numel()}")
        xc = x.contiguous()
        y = torch.empty_like(xc)
        lib.call("adnm_act_fwd", xc.data_ptr(), y.data_ptr(), xc.numel(), act, _stream())
        ctx.save_for_backward(xc)
        ctx.act = act
        return y

    @staticmethod
    def backward(ctx, dy):
        (xc,) = ctx.saved_tensors
        dy = dy.contiguous()
        dx = torch.empty_like(xc)
        lib.call("adnm_act_bwd", dy.data_ptr(), xc.data_ptr(), dx.data_ptr(), xc.numel(), ctx.act, _stream())
        return dx, None


def act(x, code):
    return ActFn.apply(x, code)


class SwishFn(torch.autograd.Function):
    """x * sigmoid(beta * x) with a learnable 1-element beta (model_untils.py:162-169)."""

    @staticmethod
    def forward(ctx, x, beta):
        _need_gpu(x)
        if x.dtype != torch.float32 or x.numel() % 4 or beta.numel() != 1:
            _unsupported("swish", f"needs fp32 tokens with a multiple of 4 elements and a 1-element beta, got {x.dtype} x {x.numel()}")
        xc = x.contiguous()
        y = torch.empty_like(xc)
        lib.call("adnm_swish_fwd", xc.data_ptr(), beta.data_ptr(), y.data_ptr(), xc.numel(), _stream())
        ctx.save_for_backward(xc, beta)
        return y

    @staticmethod
    def backward(ctx, dy):
        xc, beta = ctx.saved_tensors
        dy = dy.contiguous()
        dx = torch.empty_like(xc)
        dbeta = grad_dst(beta.data_ptr(), beta.shape, xc.device)
        nb = lib.query("adnm_swish_bwd_ws_bytes", xc.numel())
        ws = _ws(nb, xc.device)
        with FOLDS.defer(xc.device, ws):
            lib.call("adnm_swish_bwd", dy.data_ptr(), xc.data_ptr(), beta.data_ptr(), dx.data_ptr(), dbeta.data_ptr(), ws.data_ptr(), nb, xc.numel(), _stream())
        return dx, dbeta


def swish(x, beta):
    return SwishFn.apply(x, beta)


# ------------------------------------------------------------------------------------------- dense 3x3 conv (K5)
def _conv3_wstrides(w):
    """(ws_n, ws_tap, ws_k) of a (Cout, Cin, 3, 3) weight in either memory layout the kernel reads in place."""
    sn, sk, skh, skw = w.stride()
    if skh != 3 * skw:
        return None
    return sn, skw, sk


_CONV3_DPRE_ONCE = [os.environ.get("ADNM_CONV3_DPRE_ONCE", "1") != "0"]   # (measurement aid: 0 = the gradient kernels apply act' themselves)


class Conv3Fn(torch.autograd.Function):
    """nn.Conv2d(k=3, s=1, p=1) [+ bias] [+ GELU] on (B, H*W, Cin) tokens (csrc/conv3.hip): implicit GEMM on MFMA, bias and
    activation in the epilogue; the pre-activation is saved for backward exactly as autograd saves it for a separate GELU."""

    @staticmethod
    def forward(ctx, x, w, bias, H, W, act):
        B, L, K = x.shape
        N = w.shape[0]
        _need_gpu(x)
        if x.dtype != torch.float32 or w.dtype != torch.float32 or tuple(w.shape[1:]) != (K, 3, 3) or L != H * W:
            _unsupported("conv3", f"needs fp32 (B, H*W, Cin) tokens and a (Cout, Cin, 3, 3) fp32 weight, got {x.dtype} {tuple(x.shape)}, {tuple(w.shape)}")
        x2 = x.reshape(B * L, K)
        x2 = x2 if x2.stride(-1) == 1 else x2.contiguous()
        ws_ = _conv3_wstrides(w)
        if ws_ is None:
            w = w.contiguous()
            ws_ = _conv3_wstrides(w)
        dev = x.device
        y = torch.empty((B * L, N), dtype=torch.float32, device=dev)
        pre = torch.empty((B * L, N), dtype=torch.float32, device=dev) if act != lib.ACT_NONE else None
        nb = lib.query("adnm_conv3_ws_bytes", B, H, W, K, N)
        wsb = _ws(nb, dev)
        prec, qp = _gemm_prec(QUANT.record(dev, w.data_ptr(), "fc3", B * L), "f")
        lib.call("adnm_conv3_fwd", x2.data_ptr(), x2.stride(0), w.data_ptr(), ws_[0], ws_[1], ws_[2], _p(bias), y.data_ptr(), N, _p(pre), N,
                 wsb.data_ptr(), nb, B, H, W, K, N, act, prec, qp, _stream())
        ctx.save_for_backward(x2, w, pre)
        ctx.meta = (B, H, W, K, N, act, ws_, bias.data_ptr() if bias is not None else 0, bias is not None)
        return y.view(B, L, N)

    @staticmethod
    def backward(ctx, dy):
        x2, w, pre = ctx.saved_tensors
        B, H, W, K, N, act, ws_, b_ptr, has_bias = ctx.meta
        dev = x2.device
        dy2 = dy.reshape(B * H * W, N)
        dy2 = dy2 if dy2.stride(-1) == 1 else dy2.contiguous()
        if act != lib.ACT_NONE and _CONV3_DPRE_ONCE[0] and dy2.is_contiguous() and (B * H * W * N) % 4 == 0:
            # dpre = dy * act'(pre) ONCE, as one elementwise pass: both gradient kernels stage this product tile by tile — the input gradient
            # once per output-channel group, the weight gradient once per 16-channel input chunk (4 x for a 64 -> 64 conv) — and their
            # loads (two arrays + the erf) are ~half of their time (profiles/r04_conv3_phases.txt).  Same fp32 product, same rounding after.
            dpre = torch.empty_like(pre)
            lib.call("adnm_act_bwd", dy2.data_ptr(), pre.data_ptr(), dpre.data_ptr(), pre.numel(), act, _stream())
            dy2, pre, act = dpre, None, lib.ACT_NONE
        dx = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty((B * H * W, K), dtype=torch.float32, device=dev)
            nb = lib.query("adnm_conv3_ws_bytes", B, H, W, N, K)
            wsb = _ws(nb, dev)
            prec, qp = _gemm_prec(QUANT.record(dev, w.data_ptr(), "gc3", B * H * W), "g")
            lib.call("adnm_conv3_dgrad", dy2.data_ptr(), dy2.stride(0), _p(pre), N, act, w.data_ptr(), ws_[0], ws_[1], ws_[2], dx.data_ptr(), K,
                     wsb.data_ptr(), nb, B, H, W, K, N, prec, qp, _stream())
            dx = dx.view(B, H * W, K)
        # the weight gradient is produced in (Cout, 3, 3, Cin) memory order: the flat trainer's channels-last slice takes it as it lies
        # (a registered slice in any other layout is refused without being claimed: the trainer's gather then copies the gradient)
        g = grad_dst(w.data_ptr(), (N, K, 3, 3), dev, strides=(9 * K, 1, 3 * K, K))
        if not g.permute(0, 2, 3, 1).is_contiguous():
            g = torch.empty((N, 3, 3, K), dtype=torch.float32, device=dev).permute(0, 3, 1, 2)
        db = grad_dst(b_ptr, (N,), dev) if has_bias else None
        nb = lib.query("adnm_conv3_wgrad_ws_bytes", B, H, W, K, N)
        wsb = _ws(nb, dev)
        pg, pdb, prec = g.data_ptr(), _p(db), (1 if MFMA_PREC[0] == 2 or QUANT.calibrating else MFMA_PREC[0])   # fp8 configuration: bf16 operands here
        SIDE.submit(dev, (dy2, pre, x2), FOLDS.defer(dev, wsb), lambda: lib.call(
            "adnm_conv3_wgrad", dy2.data_ptr(), dy2.stride(0), _p(pre), N, act, x2.data_ptr(), x2.stride(0), pg, pdb,
            wsb.data_ptr(), nb, B, H, W, K, N, prec, _stream()))
        return dx, g, db, None, None, None


def conv3(x, w, bias, H, W, act=lib.ACT_NONE):
    return Conv3Fn.apply(x, w, bias, H, W, act)


# ------------------------------------------------------------------------------------------- stride-2 transposed conv (K9)
class ConvT2xFn(torch.autograd.Function):
    """nn.ConvTranspose2d(Cin, Cout, 3, stride=2, padding=1, output_padding=1) on tokens, (B, H*W, Cin) -> (B, 2H*2W, Cout)
    (csrc/convt.hip): cols = X . Wf on the short-GEMM MFMA kernel with the weight read as the (Cin, 9*Cout) matrix it is in memory,
    then the 4-phase gather; backward = the inverse gather + two more GEMMs."""

    @staticmethod
    def forward(ctx, x, w, bias, H, W):
        B, L, Cin = x.shape
        Cout = w.shape[1]
        _need_gpu(x)
        if x.dtype != torch.float32 or tuple(w.shape) != (Cin, Cout, 3, 3) or L != H * W or Cout % 4 or Cin % 4:
            _unsupported("convt2x", f"needs fp32 (B, H*W, Cin) tokens, a (Cin, Cout, 3, 3) weight and 4 | Cin, Cout; got {tuple(x.shape)}, {tuple(w.shape)}")
        if w.stride() == (9 * Cout, 1, 3 * Cout, Cout):      # (Cin, 3, 3, Cout) memory order (the flat trainer's): columns (tap, co)
            ct, cc = Cout, 1
        else:                                                # nn.ConvTranspose2d's own order: columns (co, tap)
            w = w.contiguous()
            ct, cc = 1, 9
        wf = torch.as_strided(w, (Cin, 9 * Cout), (9 * Cout, 1))
        x2 = x.reshape(B * L, Cin)
        x2 = x2 if x2.stride(-1) == 1 else x2.contiguous()
        cols = k_linear_dx(x2, wf, qkey=w.data_ptr(), role="f")   # (M, Cin) . (Cin, 9 Cout): activations x weight
        out = torch.empty((B * 4 * L, Cout), dtype=torch.float32, device=x.device)
        lib.call("adnm_convt_col2im", cols.data_ptr(), 9 * Cout, ct, cc, _p(bias), out.data_ptr(), Cout, B, H, W, Cout, _stream())
        ctx.save_for_backward(x2, wf)
        ctx.meta = (B, H, W, Cin, Cout, ct, cc, w.data_ptr(), bias.data_ptr() if bias is not None else 0, bias is not None)
        return out.view(B, 4 * L, Cout)

    @staticmethod
    def backward(ctx, dy):
        x2, wf = ctx.saved_tensors
        B, H, W, Cin, Cout, ct, cc, w_ptr, b_ptr, has_bias = ctx.meta
        dev = x2.device
        M = B * H * W
        dy2 = dy.reshape(4 * M, Cout)
        dy2 = dy2 if dy2.is_contiguous() else dy2.contiguous()
        dcols = torch.empty((M, 9 * Cout), dtype=torch.float32, device=dev)
        lib.call("adnm_convt_im2col", dy2.data_ptr(), Cout, dcols.data_ptr(), 9 * Cout, ct, cc, B, H, W, Cout, _stream())
        dx = k_linear(dcols, wf, None, qkey=w_ptr, role="g").view(B, H * W, Cin) if ctx.needs_input_grad[0] else None
        want = (9 * Cout, 1, 3 * Cout, Cout) if cc == 1 else (9 * Cout, 9, 3, 1)
        g = grad_dst(w_ptr, (Cin, Cout, 3, 3), dev, strides=want)   # the slice is taken only if it has the weight's own memory order
        if g.stride() != want:
            g = torch.empty((Cin, 3, 3, Cout), dtype=torch.float32, device=dev).permute(0, 3, 1, 2) if cc == 1 else \
                torch.empty((Cin, Cout, 3, 3), dtype=torch.float32, device=dev)
        k_linear_dw(x2, dcols, False, dw_out=torch.as_strided(g, (Cin, 9 * Cout), (9 * Cout, 1)))   # dWf = X^T . dcols
        db = None
        if has_bias:
            db = grad_dst(b_ptr, (Cout,), dev)
            colsum(dy2, out=db, defer=True)
        return dx, g, db, None, None


def convt2x(x, w, bias, H, W):
    return ConvT2xFn.apply(x, w, bias, H, W)


# ------------------------------------------------------------------------------------------- K1b chunked scan
def k_ssd_scan_fwd(x, xhs, Bm, Cm, dt, dths, dt_bias, A_log, D, phs, y, yhs, B, L, H, P, N, G, chunk, reverse):
    """x / y: row views whose element (row, h, p) sits at [row, h*hs + p]; Bm, Cm (M, G*N) row views; dt (M, .) row view."""
    dev = x.device
    NC = (L + chunk - 1) // chunk
    S_in = torch.empty((B, H, NC, P, N), dtype=torch.float32, device=dev)
    nb = lib.query("adnm_ssd_scan_ws_bytes", B, L, H, N, chunk, 0)
    ws = _ws(nb, dev)
    px, ldx = _rows(x); pb, ldb = _rows(Bm); pc, ldc = _rows(Cm); pt, ldt = _rows(dt); py, ldy = _rows(y)
    lib.call("adnm_ssd_scan_fwd", px, ldx, xhs, pb, ldb, pc, ldc, pt, ldt, dths, _p(dt_bias), _p(A_log), _p(D), phs, py, ldy, yhs,
             S_in.data_ptr(), ws.data_ptr(), nb, B, L, H, P, N, G, chunk, int(reverse), _dt(x), _stream())
    return S_in


def k_ssd_scan_bwd(dy, dyhs, x, xhs, Bm, Cm, dt, dths, dt_bias, A_log, D, phs, S_in, dx, dxhs, dBm, dCm, ddt, ddths, B, L, H, P, N, G, chunk,
                   reverse):
    dev = x.device
    dbias, dA, dD = (torch.empty(H, dtype=torch.float32, device=dev) for _ in range(3))
    nb = lib.query("adnm_ssd_scan_ws_bytes", B, L, H, N, chunk, 1)
    ws = _ws(nb, dev)
    pdy, lddy = _rows(dy); px, ldx = _rows(x); pb, ldb = _rows(Bm); pc, ldc = _rows(Cm); pt, ldt = _rows(dt)
    pdx, lddx = _rows(dx); pdb, lddb = _rows(dBm); pdc, lddc = _rows(dCm); pdt, ldddt = _rows(ddt)
    lib.call("adnm_ssd_scan_bwd", pdy, lddy, dyhs, px, ldx, xhs, pb, ldb, pc, ldc, pt, ldt, dths, _p(dt_bias), _p(A_log), _p(D), phs,
             S_in.data_ptr(), pdx, lddx, dxhs, pdb, lddb, pdc, lddc, pdt, ldddt, ddths, dbias.data_ptr(), dA.data_ptr(), dD.data_ptr(),
             ws.data_ptr(), nb, B, L, H, P, N, G, chunk, int(reverse), _dt(x), _stream())
    return dbias, dA, dD


class SSDScanFn(torch.autograd.Function):
    """Stand-alone K1b: y = chunk_scan(x (B,L,H,P), Bm/Cm (B,L,G*N), dt_raw (B,L,H), dt_bias, A_log, D (H))."""

    @staticmethod
    def forward(ctx, x, Bm, Cm, dt_raw, dt_bias, A_log, D, G, chunk, reverse):
        B, L, H, P = x.shape
        N = Bm.shape[-1] // G
        M = B * L
        xs, bs, cs, ts = x.reshape(M, H * P).contiguous(), Bm.reshape(M, -1).contiguous(), Cm.reshape(M, -1).contiguous(), dt_raw.reshape(M, H).contiguous()
        _need_gpu(xs)
        y = torch.empty_like(xs)
        S_in = k_ssd_scan_fwd(xs, P, bs, cs, ts, 1, dt_bias, A_log, D, 1, y, P, B, L, H, P, N, G, chunk, reverse)
        ctx.save_for_backward(xs, bs, cs, ts, dt_bias, A_log, D, S_in)
        ctx.dims = (B, L, H, P, N, G, chunk, reverse)
        return y.view(B, L, H, P)

    @staticmethod
    def backward(ctx, dy):
        xs, bs, cs, ts, dt_bias, A_log, D, S_in = ctx.saved_tensors
        B, L, H, P, N, G, chunk, reverse = ctx.dims
        M = B * L
        dy2 = dy.reshape(M, H * P).contiguous()
        dx, dB, dC, ddt = torch.empty_like(xs), torch.empty_like(bs), torch.empty_like(cs), torch.empty_like(ts)
        dbias, dA, dD = k_ssd_scan_bwd(dy2, P, xs, P, bs, cs, ts, 1, dt_bias, A_log, D, 1, S_in, dx, P, dB, dC, ddt, 1, B, L, H, P, N, G, chunk, reverse)
        return dx.view(B, L, H, P), dB.view(B, L, -1), dC.view(B, L, -1), ddt.view(B, L, H), dbias, dA, dD, None, None, None


def ssd_scan(x, Bm, Cm, dt_raw, dt_bias, A_log, D, groups=1, chunk=256, reverse=False):
    return SSDScanFn.apply(x, Bm, Cm, dt_raw, dt_bias, A_log, D, groups, chunk, reverse)


class IGateFn(torch.autograd.Function):
    """IntensityGate: silu(enhance * (x - threshold)) with learnable scalars, one pass each way."""

    @staticmethod
    def forward(ctx, x, enhance, threshold):
        x = x.contiguous()
        _need_gpu(x)
        y = torch.empty_like(x)
        lib.call("adnm_igate_fwd", x.data_ptr(), enhance.data_ptr(), threshold.data_ptr(), y.data_ptr(), x.numel(), _dt(x), _stream())
        ctx.save_for_backward(x, enhance, threshold)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, enhance, threshold = ctx.saved_tensors
        dy = dy.contiguous()
        dx = torch.empty_like(x)
        de, dth = torch.empty_like(enhance), torch.empty_like(threshold)
        nb = lib.query("adnm_igate_bwd_ws_bytes", x.numel())
        ws = _ws(nb, x.device)
        lib.call("adnm_igate_bwd", dy.data_ptr(), x.data_ptr(), enhance.data_ptr(), threshold.data_ptr(), dx.data_ptr(), de.data_ptr(),
                 dth.data_ptr(), ws.data_ptr(), nb, x.numel(), _dt(x), _stream())
        return dx, de, dth


def igate(x, enhance, threshold):
    _need_gpu(x)
    if x.numel() % 4 or x.dtype not in _DT or enhance.dtype != torch.float32:
        _unsupported("igate", f"needs fp32/bf16 tokens with a multiple of 4 elements and fp32 scalars, got {x.dtype} x {x.numel()}")
    return IGateFn.apply(x, enhance, threshold)


class EMulFn(torch.autograd.Function):
    """a * b on (…, C) fp32 tokens (row views allowed): one HIP pass each way."""

    @staticmethod
    def forward(ctx, a, b):
        _need_gpu(a)
        C = a.shape[-1]
        a2, b2 = a.reshape(-1, C), b.reshape(-1, C)
        a2 = a2 if a2.stride(-1) == 1 and a2.stride(0) % 4 == 0 else a2.contiguous()
        b2 = b2 if b2.stride(-1) == 1 and b2.stride(0) % 4 == 0 else b2.contiguous()
        y = torch.empty((a2.shape[0], C), dtype=torch.float32, device=a.device)
        lib.call("adnm_emul_fwd", a2.data_ptr(), a2.stride(0), b2.data_ptr(), b2.stride(0), y.data_ptr(), a2.shape[0], C, _stream())
        ctx.save_for_backward(a2, b2)
        ctx.shp = a.shape
        return y.view(a.shape)

    @staticmethod
    def backward(ctx, dy):
        a2, b2 = ctx.saved_tensors
        M, C = a2.shape
        g = dy.reshape(M, C)
        g = g if g.is_contiguous() else g.contiguous()
        da, db = torch.empty((M, C), dtype=torch.float32, device=g.device), torch.empty((M, C), dtype=torch.float32, device=g.device)
        lib.call("adnm_emul_bwd", g.data_ptr(), a2.data_ptr(), a2.stride(0), b2.data_ptr(), b2.stride(0), da.data_ptr(), db.data_ptr(), M, C, _stream())
        return da.view(ctx.shp), db.view(ctx.shp)


def emul(a, b):
    if a.dtype != torch.float32 or b.dtype != torch.float32 or a.shape != b.shape or a.shape[-1] % 4:
        _unsupported("emul", f"needs two fp32 token tensors of one shape with 4 | C, got {a.dtype} {tuple(a.shape)}, {b.dtype} {tuple(b.shape)}")
    return EMulFn.apply(a, b)


class ChanPadFn(torch.autograd.Function):
    """(…, Cin) tokens -> (…, Cout): zero-padded (Cout > Cin) or cropped channels in one HIP pass; backward is the same kernel the other way."""

    @staticmethod
    def forward(ctx, x, Cout):
        _need_gpu(x)
        Cin = x.shape[-1]
        x2 = x.reshape(-1, Cin)
        x2 = x2 if x2.stride(-1) == 1 else x2.contiguous()
        y = torch.empty((x2.shape[0], Cout), dtype=torch.float32, device=x.device)
        lib.call("adnm_chancopy", x2.data_ptr(), x2.stride(0), Cin, y.data_ptr(), Cout, x2.shape[0], _stream())
        ctx.meta = (x.shape, Cin, Cout)
        return y.view(*x.shape[:-1], Cout)

    @staticmethod
    def backward(ctx, dy):
        shp, Cin, Cout = ctx.meta
        d2 = dy.reshape(-1, Cout)
        d2 = d2 if d2.stride(-1) == 1 else d2.contiguous()
        dx = torch.empty((d2.shape[0], Cin), dtype=torch.float32, device=dy.device)
        lib.call("adnm_chancopy", d2.data_ptr(), d2.stride(0), Cout, dx.data_ptr(), Cin, d2.shape[0], _stream())
        return dx.view(shp), None


def chanpad(x, Cout):
    if x.dtype != torch.float32:
        _unsupported("chanpad", f"needs fp32 tokens, got {x.dtype}")
    return ChanPadFn.apply(x, Cout)


class IGateResFn(torch.autograd.Function):
    """EncoderToDecoder's entry (model_untils.py:761-763): IntensityGate(x + gama * res), res the (B, 1, C) bridge gate broadcast over
    the tokens — one pass each way (csrc/elementwise.hip)."""

    @staticmethod
    def forward(ctx, x, res, gama, enhance, threshold):
        _need_gpu(x)
        B, L, C = x.shape
        per_token = int(res.numel() == x.numel())   # the reference's own call form hands over the gate already expanded over the tokens
        x, r = x.contiguous(), (res.reshape(B, L, C) if per_token else res.reshape(B, C)).contiguous()
        y = torch.empty_like(x)
        lib.call("adnm_igate_res_fwd", x.data_ptr(), r.data_ptr(), per_token, gama.data_ptr(), enhance.data_ptr(), threshold.data_ptr(), y.data_ptr(),
                 B, L, C, _stream())
        ctx.save_for_backward(x, r, gama, enhance, threshold)
        ctx.rshape, ctx.per_token = res.shape, per_token
        return y

    @staticmethod
    def backward(ctx, dy):
        x, r, gama, enhance, threshold = ctx.saved_tensors
        B, L, C = x.shape
        dev = x.device
        dy = dy.contiguous()
        dx, dres = torch.empty_like(x), torch.empty_like(r)
        dg, de, dt = (grad_dst(p.data_ptr(), p.shape, dev) for p in (gama, enhance, threshold))
        nb = lib.query("adnm_igate_res_bwd_ws_bytes", B, L, C)
        ws = _ws(nb, dev)
        with FOLDS.defer(dev, ws):   # d res is complete when the launch ends; only the three scalar gradients go through the fold
            lib.call("adnm_igate_res_bwd", dy.data_ptr(), x.data_ptr(), r.data_ptr(), ctx.per_token, gama.data_ptr(), enhance.data_ptr(), threshold.data_ptr(),
                     dx.data_ptr(), dres.data_ptr(), dg.data_ptr(), de.data_ptr(), dt.data_ptr(), ws.data_ptr(), nb, B, L, C, _stream())
        return dx, dres.view(ctx.rshape), dg, de, dt


def igate_res(x, res, gama, enhance, threshold):
    _need_gpu(x)
    B, L, C = x.shape
    if x.dtype != torch.float32 or C % 4 or res.numel() not in (B * C, B * L * C) or any(p.numel() != 1 for p in (gama, enhance, threshold)):
        _unsupported("igate_res", f"needs fp32 (B, L, C) tokens with 4 | C, a (B, 1, C) or (B, L, C) gate and 1-element scalars, got {x.dtype} {tuple(x.shape)}, "
                                  f"gate {tuple(res.shape)}")
    return IGateResFn.apply(x, res, gama, enhance, threshold)


class SkipGateFn(torch.autograd.Function):
    """EncoderToDecoder's three pooled, gated branches + their mix in 2 launches forward / 5-6 backward
    (csrc/skipgate.hip).  params: the 18 tensors in the order include/adnm_hip.h documents."""

    @staticmethod
    def forward(ctx, x, h, w, *params):
        x = x.contiguous()
        _need_gpu(x)
        b, l, c = x.shape
        params = tuple(p.contiguous() for p in params)
        pooled = torch.empty((3, b, l, c), dtype=x.dtype, device=x.device)
        conv = torch.empty_like(pooled)
        out = torch.empty_like(x)
        lib.call("adnm_skipgate_fwd", x.data_ptr(), lib.ptr_table(params), pooled.data_ptr(), conv.data_ptr(), out.data_ptr(), b, h, w, c, _stream())
        ctx.save_for_backward(x, pooled, conv, *params)
        ctx.hw = (h, w)
        return out

    @staticmethod
    def backward(ctx, dout):
        x, pooled, conv, *params = ctx.saved_tensors
        h, w = ctx.hw
        b, l, c = x.shape
        dout = dout.contiguous()
        dx = torch.empty_like(x)
        dp = torch.empty(int(lib.query("adnm_skipgate_grad_floats", c)), dtype=torch.float32, device=x.device)
        nb = lib.query("adnm_skipgate_bwd_ws_bytes", b, h, w, c)
        ws = _ws(nb, x.device)
        with FOLDS.defer(x.device, ws):   # every parameter of an EncoderToDecoder has this one node
            lib.call("adnm_skipgate_bwd", dout.data_ptr(), x.data_ptr(), lib.ptr_table(params), pooled.data_ptr(), conv.data_ptr(), dx.data_ptr(),
                     dp.data_ptr(), ws.data_ptr(), nb, b, h, w, c, _stream())
        o = [0]

        def take(n, like):
            v = dp[o[0]:o[0] + n].view_as(like)
            o[0] += n
            return v
        dw0, dw1, dw2 = take(12 * c, params[0]), take(12 * c, params[2]), take(36 * c, params[4])
        dgamma = take(c, params[17])
        dfw13, dfb13, dfw33, dfb33 = take(c, params[6]), take(c, params[7]), take(c, params[8]), take(c, params[9])
        db0, db1, db2 = take(c, params[1]), take(c, params[3]), take(c, params[5])
        da = [take(1, params[14 + i]) for i in range(3)]
        de13, dt13, de33, dt33 = (take(1, params[10 + i]) for i in range(4))
        return (dx, None, None, dw0, db0, dw1, db1, dw2, db2, dfw13, dfb13, dfw33, dfb33, de13, dt13, de33, dt33, da[0], da[1], da[2], dgamma)


def skipgate(x, h, w, params):
    _need_gpu(x)
    if x.dtype != torch.float32 or x.dim() != 3 or x.shape[-1] % 4 or x.shape[1] != h * w:
        _unsupported("skipgate", f"needs fp32 (B, H*W, C) tokens with 4 | C, got {x.dtype} {tuple(x.shape)} for a {h}x{w} map")
    return SkipGateFn.apply(x, h, w, *params)


class RainLossFn(torch.autograd.Function):
    """enRainfallLoss value + d/dpred in one pass; backward is a single scale of the stored gradient."""

    @staticmethod
    def forward(ctx, pred, target, omega_t, alpha, gamma):
        p, t = pred.contiguous(), target.contiguous()
        _need_gpu(p)
        loss = torch.empty((), dtype=torch.float32, device=p.device)
        grad = torch.empty_like(p)
        nb = lib.query("adnm_rainloss_ws_bytes", p.numel())
        ws = _ws(nb, p.device)
        lib.call("adnm_rainloss", p.data_ptr(), t.data_ptr(), loss.data_ptr(), grad.data_ptr(), ws.data_ptr(), nb, p.numel(), float(omega_t),
                 float(alpha), float(gamma), _stream())
        ctx.save_for_backward(grad)
        ctx.shp = pred.shape
        return loss

    @staticmethod
    def backward(ctx, dl):
        (grad,) = ctx.saved_tensors
        return (grad * dl).view(ctx.shp), None, None, None, None


def rainloss(pred, target, omega_t, alpha, gamma):
    return RainLossFn.apply(pred, target, omega_t, alpha, gamma)


class BridgePoolFn(torch.autograd.Function):
    """Channel_Att_Bridge's pooling of all skips in one launch each way (csrc/bridge.hip): (*aliases of the skips, att (B, sum C)).  The
    skips come back as autograd aliases: the gradient of a skip's later consumers and the pool's broadcast gradient are summed in the
    one backward launch."""

    @staticmethod
    def forward(ctx, *xs):
        _need_gpu(xs[0])
        B, dev = xs[0].shape[0], xs[0].device
        xc = [x if x.is_contiguous() else x.contiguous() for x in xs]
        Ls, Cs = [x.shape[1] for x in xs], [x.shape[2] for x in xs]
        S = sum(Cs)
        att = torch.empty((B, S), dtype=torch.float32, device=dev)
        nb = lib.query("adnm_bridge_pool_ws_bytes", B, S)
        ws = _ws(nb, dev)
        lib.call("adnm_bridge_pool_fwd", len(xs), lib.ptr_table(xc), lib.i64_table(Ls), lib.i64_table(Cs), att.data_ptr(), ws.data_ptr(), nb, B, _stream())
        ctx.meta = (B, Ls, Cs)
        ctx.set_materialize_grads(False)   # an unused output arrives as None, not as a full-size zeros tensor
        return (*xs, att)

    @staticmethod
    def backward(ctx, *grads):
        B, Ls, Cs = ctx.meta
        dxa, datt = list(grads[:-1]), grads[-1]
        if datt is None:
            return tuple(dxa)
        dev = datt.device
        datt = datt if datt.is_contiguous() else datt.contiguous()
        dxa = [g if g is None or g.is_contiguous() else g.contiguous() for g in dxa]
        need = ctx.needs_input_grad
        dx = [torch.empty((B, Ls[k], Cs[k]), dtype=torch.float32, device=dev) if need[k] else None for k in range(len(Ls))]
        lib.call("adnm_bridge_pool_bwd", len(Ls), lib.ptr_table(dxa), datt.data_ptr(), lib.i64_table(Ls), lib.i64_table(Cs), lib.ptr_table(dx), B, _stream())
        return tuple(dx)


def bridge_pool(xs):
    """-> ([aliases of the skips], att (B, sum C)): the token mean of every skip, concatenated.  Callers hand the aliases to the skips'
    later consumers."""
    _need_gpu(xs[0])
    if not 1 <= len(xs) <= 8 or any(x.dtype != torch.float32 or x.dim() != 3 or x.shape[-1] % 4 or x.shape[0] != xs[0].shape[0] for x in xs):
        _unsupported("bridge_pool", f"pools 1-8 fp32 (B, L, C) token tensors with 4 | C, got {[(x.dtype, tuple(x.shape)) for x in xs]}")
    out = BridgePoolFn.apply(*xs)
    return list(out[:-1]), out[-1]


_BRIDGE_ROWS = 8   # csrc/bridge.hip: kMaxB sample rows of att per launch


class BridgeHeadsFn(torch.autograd.Function):
    """The live heads of Channel_Att_Bridge in one launch each way + one fold (csrc/bridge.hip): for every head i,
    gate_i = IntensityGate(att . W_i^T + b_i).  args: att (B, 1, S), enhance, threshold, then W_0, b_0, W_1, b_1, ...; returns the gates
    (B, 1, C_i)."""

    @staticmethod
    def forward(ctx, att, enhance, threshold, *wb):
        _need_gpu(att)
        B, _, S = att.shape
        a2 = att.reshape(B, S).contiguous()
        Ws, bs = [w.contiguous() for w in wb[0::2]], list(wb[1::2])
        Cs = [w.shape[0] for w in Ws]
        dev = att.device
        zs = [torch.empty((B, c), dtype=torch.float32, device=dev) for c in Cs]
        ys = [torch.empty((B, 1, c), dtype=torch.float32, device=dev) for c in Cs]
        for b0 in range(0, B, _BRIDGE_ROWS):   # the kernel holds <= 8 sample rows of att in LDS: a larger batch goes through in row chunks
            nb_ = min(_BRIDGE_ROWS, B - b0)
            lib.call("adnm_bridge_heads_fwd", a2[b0:].data_ptr(), lib.ptr_table(Ws), lib.ptr_table(bs), lib.i64_table(Cs), len(Ws), enhance.data_ptr(),
                     threshold.data_ptr(), lib.ptr_table([z[b0:] for z in zs]), lib.ptr_table([y[b0:] for y in ys]), nb_, S, _stream())
        ctx.save_for_backward(a2, enhance, threshold, *Ws, *zs)
        ctx.meta = (B, S, Cs, [b is not None for b in bs], [b.data_ptr() if b is not None else 0 for b in bs], att.shape)
        ctx.set_materialize_grads(False)
        return tuple(ys)

    @staticmethod
    def backward(ctx, *dys):
        B, S, Cs, has_b, b_ptrs, ashape = ctx.meta
        n = len(Cs)
        saved = ctx.saved_tensors
        a2, enhance, threshold, Ws, zs = saved[0], saved[1], saved[2], saved[3:3 + n], saved[3 + n:]
        dev = a2.device
        dys = [(d.reshape(B, c).contiguous() if d is not None else torch.zeros((B, c), dtype=torch.float32, device=dev)) for d, c in zip(dys, Cs)]
        datt = torch.empty((B, S), dtype=torch.float32, device=dev)
        dWs = [grad_dst(w.data_ptr(), w.shape, dev) for w in Ws]
        dbs = [grad_dst(p, (c,), dev) if h else None for p, c, h in zip(b_ptrs, Cs, has_b)]
        de, dt = grad_dst(enhance.data_ptr(), enhance.shape, dev), grad_dst(threshold.data_ptr(), threshold.shape, dev)
        nb = lib.query("adnm_bridge_heads_bwd_ws_bytes", sum(Cs), min(B, _BRIDGE_ROWS), S)
        ws = _ws(nb, dev)
        for b0 in range(0, B, _BRIDGE_ROWS):
            nb_ = min(_BRIDGE_ROWS, B - b0)
            if b0 == 0:
                tW, tb, te, tt = dWs, dbs, de, dt
            else:   # row chunks after the first: their parameter-gradient contributions are added to the first chunk's
                tW = [torch.empty_like(w) for w in dWs]
                tb = [torch.empty_like(b) if b is not None else None for b in dbs]
                te, tt = torch.empty_like(de), torch.empty_like(dt)
            lib.call("adnm_bridge_heads_bwd", a2[b0:].data_ptr(), lib.ptr_table(Ws), lib.i64_table(Cs), n, enhance.data_ptr(), threshold.data_ptr(),
                     lib.ptr_table([z[b0:] for z in zs]), lib.ptr_table([d[b0:] for d in dys]), datt[b0:].data_ptr(), lib.ptr_table(tW), lib.ptr_table(tb),
                     te.data_ptr(), tt.data_ptr(), ws.data_ptr(), nb, nb_, S, _stream())
            if b0:
                dst = dWs + [b for b in dbs if b is not None] + [de, dt]
                torch._foreach_add_(dst, tW + [b for b in tb if b is not None] + [te, tt])
        out = [datt.view(ashape), de, dt]
        for dw, db in zip(dWs, dbs):
            out += [dw, db]
        return tuple(out)


def bridge_heads(att, enhance, threshold, weights, biases):
    """att (B, 1, S) -> [IntensityGate(att . W_i^T + b_i) for i]: every live head of Channel_Att_Bridge in one launch."""
    _need_gpu(att)
    B, one, S = att.shape
    if att.dtype != torch.float32 or one != 1 or S % 4 or S > 3072 or B < 1 or not 1 <= len(weights) <= 8 or \
            any(w.dim() != 2 or w.shape[1] != S for w in weights) or enhance.numel() != 1 or threshold.numel() != 1:
        _unsupported("bridge_heads", f"needs fp32 (B, 1, S) pooled channels with 4 | S <= 3072 and 1..8 (C_i, S) weights, got {tuple(att.shape)}, "
                                     f"{[tuple(w.shape) for w in weights]}")
    wb = []
    for w, b in zip(weights, biases):
        wb += [w, b]
    return list(BridgeHeadsFn.apply(att, enhance, threshold, *wb))


class Conv1d3Fn(torch.autograd.Function):
    """nn.Conv1d(1, 1, 3, padding=1) on (B, 1, n): one single-workgroup launch each way."""

    @staticmethod
    def forward(ctx, x, w, bias):
        _need_gpu(x)
        B, _, n = x.shape
        xc, wc = x.contiguous(), w.contiguous()
        y = torch.empty_like(xc)
        lib.call("adnm_conv1d3_fwd", xc.data_ptr(), wc.data_ptr(), _p(bias), y.data_ptr(), B, n, _stream())
        ctx.save_for_backward(xc, wc)
        ctx.meta = (w.shape, bias is not None)
        return y

    @staticmethod
    def backward(ctx, dy):
        xc, wc = ctx.saved_tensors
        B, _, n = xc.shape
        dy = dy.contiguous()
        dx = torch.empty_like(xc)
        dwb = torch.empty(4, dtype=torch.float32, device=xc.device)
        lib.call("adnm_conv1d3_bwd", dy.data_ptr(), xc.data_ptr(), wc.data_ptr(), dx.data_ptr(), dwb.data_ptr(), B, n, _stream())
        wshape, has_bias = ctx.meta
        return dx, dwb[:3].view(wshape), dwb[3:4] if has_bias else None


def conv1d3(x, w, bias):
    _need_gpu(x)
    if x.dtype != torch.float32 or x.dim() != 3 or x.shape[1] != 1 or w.numel() != 3 or x.shape[0] * x.shape[2] > (1 << 20):
        _unsupported("conv1d3", f"is nn.Conv1d(1, 1, 3, padding=1) on fp32 (B, 1, n) with B*n <= 2^20, got {x.dtype} {tuple(x.shape)}, weight {tuple(w.shape)}")
    return Conv1d3Fn.apply(x, w, bias)


class Attn4Fn(torch.autograd.Function):
    """softmax(q k^T * scale) v for 4-wide heads straight from to_qkv's (B, L, 3*inner) output (csrc/attn4.hip)."""

    @staticmethod
    def forward(ctx, qkv, heads, scale):
        _need_gpu(qkv)
        qkv = qkv.contiguous()
        B, L, three = qkv.shape
        inner = three // 3
        out = torch.empty((B, L, inner), dtype=torch.float32, device=qkv.device)
        lse = torch.empty((B, heads, L), dtype=torch.float32, device=qkv.device)
        lib.call("adnm_attn4_fwd", qkv.data_ptr(), out.data_ptr(), lse.data_ptr(), B, L, heads, float(scale), _stream())
        ctx.save_for_backward(qkv, out, lse)
        ctx.meta = (heads, float(scale))
        return out

    @staticmethod
    def backward(ctx, dout):
        qkv, out, lse = ctx.saved_tensors
        heads, scale = ctx.meta
        B, L, _ = qkv.shape
        dqkv = torch.empty_like(qkv)
        lib.call("adnm_attn4_bwd", dout.contiguous().data_ptr(), qkv.data_ptr(), out.data_ptr(), lse.data_ptr(), dqkv.data_ptr(), B, L, heads, scale,
                 _stream())
        return dqkv, None, None


def attn4(qkv, heads, scale):
    _need_gpu(qkv)
    if qkv.dtype != torch.float32 or qkv.dim() != 3 or qkv.shape[-1] != 12 * heads or qkv.shape[1] > 2048:
        _unsupported("attn4", f"is soft-max attention with 4-wide heads over <= 2048 tokens on fp32 (B, L, 12*heads) qkv, got {qkv.dtype} "
                              f"{tuple(qkv.shape)} for {heads} heads")
    return Attn4Fn.apply(qkv, heads, scale)
