import os
import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_case(name):
    """Returns (params, grads, inputs, input_grads, outs, cots) of a module_case fixture."""
    z = np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)
    t = lambda k: torch.from_numpy(z[k])
    pick = lambda pre: {k[len(pre):]: t(k) for k in z.files if k.startswith(pre)}
    outs = [t(f"out{i}") for i in range(16) if f"out{i}" in z.files]
    cots = [t(f"cot{i}") for i in range(16) if f"cot{i}" in z.files]
    return pick("p."), pick("g."), pick("in."), pick("gin."), outs, cots


def load_npz(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)
    return {k: (torch.from_numpy(z[k]) if z[k].dtype.kind in "fi" else z[k]) for k in z.files}


def rel_l2(a, b):
    a, b = a.detach().double().flatten(), b.detach().double().flatten()
    return float((a - b).norm() / (b.norm() + 1e-30))


def assert_close(a, b, tol, what="", atol=0.0):
    """rel-L2 check; `atol` (per-element RMS) absorbs tensors that are mathematically zero
    (e.g. the gradient of a bias that an InstanceNorm removes) and hold only round-off."""
    assert a.shape == b.shape, f"{what}: shape {tuple(a.shape)} vs {tuple(b.shape)}"
    a, b = a.detach().double().cpu().flatten(), b.detach().double().cpu().flatten()
    err = float((a - b).norm())
    bound = tol * float(b.norm()) + atol * (b.numel() ** 0.5)
    assert err <= bound, f"{what}: |err| {err:.3e} > {bound:.3e} (rel-L2 {err / (float(b.norm()) + 1e-30):.3e}, tol {tol:.1e})"


def check_update_deltas(z, names, deltas, lr=1e-3):
    """The update itself, p_after - p_before, against the reference's.  The first AdamW step moves every element by
    -lr * g / (|g| + eps) - lr * wd * p: by ~lr in the direction of -sign(g), whatever |g| is.  So the MAGNITUDE of the update is
    checked everywhere (it catches a missing / doubled / mis-scaled update), and its SIGN wherever the gradient is well above the
    fp32 round-off of the reference itself (an element whose gradient is noise moves by +-lr at random on both sides):
      * 1-element tensors (the 360 scalar mixes) whose |g| >= 5e-4 of the total norm: the value of the update;
      * every tensor: the absolute sum (5 %), and the sum up to the flips of a few % of near-zero-gradient elements.
    (Gradient directions themselves are pinned by the grad_probe check; the AdamW arithmetic by tests/test_trainer_gpu.py.)"""
    small = iter(torch.split(z["delta_small"].double(), [int(n) for n in z["delta_small_sizes"]]))
    gn, gtot = z["grad_norms"].double(), float(z["grad_total_norm"])
    checked_scalars = 0
    for i, (k, d) in enumerate(zip(names, deltas)):
        n = d.numel()
        d = d.flatten().cpu()
        ref = next(small) if n <= 8 else None
        if float(gn[i]) < 0:
            assert float(d.abs().max()) == 0.0, f"{k}: a parameter without gradient must not move"
            continue
        # 5 %: a gradient that is ~eps = 1e-9 after clipping gives |g| / (|g| + eps) anywhere below 1, and its round-off moves that
        assert abs(float(d.abs().sum()) - float(z["delta_abs"][i])) <= 0.05 * lr * n + 1e-12, f"{k}: |update| sum {float(d.abs().sum())} vs {float(z['delta_abs'][i])}"
        if n == 1 and float(gn[i]) >= 5e-4 * gtot:
            checked_scalars += 1
            assert float((d - ref).abs().max()) <= 0.02 * lr, f"{k}: update {d.tolist()} vs reference {ref.tolist()}"
        elif n > 1:
            assert abs(float(d.sum()) - float(z["delta_sum"][i])) <= 3 * lr * n ** 0.5 + 0.05 * lr * n, f"{k}: update sum"
    assert checked_scalars >= 30, checked_scalars
