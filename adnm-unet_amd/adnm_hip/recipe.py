"""Deterministic, framework-independent parameter / input recipe.

Used on both sides of every parity check: the container-only reference harness
(oracle/ref_harness.py) fills the *reference* model with it when the golden
fixtures are generated, and the tests / bench fill *this* package's model with
it on the GPU box, so no state_dict ever has to travel.

All pseudo-random numbers come from a splitmix64 integer hash evaluated with
numpy uint64 arithmetic (exact, platform independent), seeded by crc32(name).
"""
import zlib
import math
import numpy as np
import torch

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _splitmix64(x):
    x = (x + np.uint64(0x9E3779B97F4A7C15)) & _M64
    z = x
    z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M64
    z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M64
    return z ^ (z >> np.uint64(31))


def uniform01(name, n, salt=0):
    """n float64 values in [0,1), a pure function of (name, salt, index)."""
    seed = np.uint64(zlib.crc32(name.encode()) + (salt << 32))
    with np.errstate(over="ignore"):
        idx = np.arange(n, dtype=np.uint64) * np.uint64(0x2545F4914F6CDD1D) + seed
        z = _splitmix64(idx)
    return (z >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)


def sym(name, n, salt=0):
    """n float64 values in [-1,1)."""
    return uniform01(name, n, salt) * 2.0 - 1.0


def _values_for(name, p, salt):
    n = p.numel()
    leaf = name.split(".")[-1]
    if leaf == "A_log":  # reference init: log U(1,16)  (ADNssd.py:215-216)
        return np.log(1.0 + 15.0 * uniform01(name, n, salt))
    if leaf == "dt_bias":  # reference init: softplus^-1(exp U(log 1e-3, log 0.1)) (ADNssd.py:201-208)
        dt = np.exp(uniform01(name, n, salt) * (math.log(0.1) - math.log(1e-3)) + math.log(1e-3))
        dt = np.maximum(dt, 1e-4)
        return dt + np.log(-np.expm1(-dt))
    v = sym(name, n, salt)
    if leaf == "bias":  # zero-initialised (Linear/LayerNorm) or random (conv) biases alike
        return 0.05 * v
    if p.dim() >= 2 and "_scale" not in name:
        fan_in = max(1, n // p.shape[0])
        return v * math.sqrt(3.0 / fan_in)
    flat = p.detach().reshape(-1).double().cpu().numpy()
    if n > 0 and np.all(flat == flat[0]):
        base = float(flat[0])  # deterministic constant init (1, 0, 0.33, 0.1 ...)
        return base + 0.1 * max(abs(base), 0.5) * v
    return 0.05 * v  # randomly initialised vectors (conv biases)


@torch.no_grad()
def fill_parameters(model, salt=0):
    """Overwrite every trainable parameter of `model` in place. Frozen tensors
    (the Haar wt_filter / iwt_filter, WTConv2d.py:75-76) are left untouched."""
    for name, p in model.named_parameters():
        if not p.requires_grad:
            continue
        if name.startswith("module."):
            name = name[len("module."):]
        vals = _values_for(name, p, salt)
        p.copy_(torch.from_numpy(np.asarray(vals, dtype=np.float64)).to(p.dtype).reshape(p.shape))
    return model


def state_dict_from_manifest(manifest, salt=0):
    """Rebuild the recipe-filled state_dict from tests/golden/state_dict_manifest.json
    (shapes + init constants recorded from the reference model) without any model object.
    Frozen tensors (Haar filters) come back as zeros: nothing on the oracle path reads them."""
    sd = {}
    for name, m in manifest.items():
        shape = tuple(m["shape"])
        if not m["trainable"]:
            sd[name] = torch.zeros(shape)
            continue
        proto = torch.full(shape, m["const"]) if m["const"] is not None else torch.arange(max(1, int(np.prod(shape))), dtype=torch.float32).reshape(shape if shape else ())
        vals = _values_for(name, proto, salt)
        sd[name] = torch.from_numpy(np.asarray(vals, dtype=np.float64)).float().reshape(shape)
    return sd


def radar_batch(batch, frames, size, salt=0, name="radar"):
    """Synthetic look-alike of a Shanghai radar batch: values in [0, 70/255]
    (datasets/Shanghai.py:55-57 scales uint8 0..70 by 1/255), smooth in space so
    that a few pixels exceed nothing exotic. float32 (B, frames, 1, S, S)."""
    n = batch * frames * size * size
    u = uniform01(name, n, salt).reshape(batch, frames, 1, size, size)
    return torch.from_numpy((u * (70.0 / 255.0)).astype(np.float32))


def tensor(name, shape, scale=1.0, salt=0, positive=False):
    n = int(np.prod(shape))
    v = uniform01(name, n, salt) if positive else sym(name, n, salt)
    return torch.from_numpy((v * scale).astype(np.float32).reshape(shape))
