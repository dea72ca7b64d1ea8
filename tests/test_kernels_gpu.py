"""HIP kernels (through the C-ABI, via adnm_hip.ops) against the oracle on identical seeded inputs
and against the golden fixtures produced by the reference.  fp32 tolerance: rel-L2 <= 1e-4 on outputs,
<= 1e-3 on gradients; bf16 storage: <= 2e-2 vs the fp32 oracle."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

import adnm_oracle as O
from adnm_hip import ops, lib, recipe
from util import load_case, load_npz, assert_close

pytestmark = pytest.mark.gpu
DEV = "cuda"
OUT_TOL, GRAD_TOL = 1e-4, 1e-3


def T(name, shape, scale=1.0, positive=False):
    return recipe.tensor(name, shape, scale, positive=positive)


def leaf(t, dev=None):
    t = t.clone().to(dev) if dev else t.clone()
    return t.requires_grad_(True)


# ------------------------------------------------------------------------------------------- row norms
@pytest.mark.parametrize("M,d,mean,bias,affine", [
    (4 * 16384, 32, False, False, True),   # refiner RMSNorm
    (1000, 32, False, False, False),
    (64, 1024, False, False, True),        # deep-level RMSNorm
    (777, 64, True, True, False),          # Mamba2.norm
    (64, 2048, True, True, False),
    (1024, 128, True, False, True),        # BiasFree_LayerNorm + attn_scale/shift
    (37, 48, True, True, True),
    (5, 260, False, False, True),
])
def test_rownorm(M, d, mean, bias, affine):
    eps = 1e-6 if not mean else 1e-5
    x, w = T(f"rn.x{M}{d}", (M, d), 2.0), 1 + 0.2 * T(f"rn.w{d}", (d,))
    b = 0.1 * T(f"rn.b{d}", (d,)) if bias else None
    sc = torch.tensor(1.3) if affine else None
    sh = torch.tensor(-0.2) if affine else None
    cot = T(f"rn.c{M}{d}", (M, d))
    # oracle (fp64 to make it the ground truth)
    xo, wo = leaf(x.double()), leaf(w.double())
    bo = leaf(b.double()) if bias else None
    sco, sho = (leaf(sc.double()), leaf(sh.double())) if affine else (None, None)
    if mean:
        yo = O.layernorm(xo, wo, bo if bias else 0.0, eps)
    else:
        yo = O.rmsnorm(xo, wo, eps)
    if affine:
        yo = sco * yo + sho
    (yo * cot.double()).sum().backward()
    # kernel
    xg, wg = leaf(x, DEV), leaf(w, DEV)
    bg = leaf(b, DEV) if bias else None
    scg, shg = (leaf(sc, DEV), leaf(sh, DEV)) if affine else (None, None)
    yg = ops.rownorm(xg, wg, bg, scg, shg, eps, mean)
    (yg * cot.to(DEV)).sum().backward()
    assert_close(yg, yo, OUT_TOL, "y")
    assert_close(xg.grad, xo.grad, GRAD_TOL, "dx")
    assert_close(wg.grad, wo.grad, GRAD_TOL, "dw")
    if bias:
        assert_close(bg.grad, bo.grad, GRAD_TOL, "db")
    if affine:
        assert_close(scg.grad, sco.grad, GRAD_TOL, "dscale", atol=1e-4)
        assert_close(shg.grad, sho.grad, GRAD_TOL, "dshift", atol=1e-4)


@pytest.mark.parametrize("M,d,mean,gamma,affine,res", [(1000, 32, True, False, True, True), (777, 64, False, True, True, True),
                                                      (4096, 256, True, True, False, True), (50, 512, False, False, True, False),
                                                      (131, 1024, True, True, True, True), (65536, 32, True, False, True, True)])
def test_mixnorm(M, d, mean, gamma, affine, res):
    """x = gamma * (s0*x0 + s1*x1); xn = scale * norm(x) + shift (ADNMUNet.py:149-158 of the reference) as one launch each way; `res`:
    x is also read downstream (the residual path), i.e. a gradient arrives on both outputs.  Forward must equal the two-kernel chain
    bit for bit (the mix is stored by both with the same operation order)."""
    eps = 1e-5
    x0, x1 = T(f"mn.a{M}{d}", (M, d), 2.0), T(f"mn.b{M}{d}", (M, d), 1.5)
    w, g = 1 + 0.2 * T(f"mn.w{d}", (d,)), (1 + 0.3 * T(f"mn.g{d}", (d,))) if gamma else None
    s0, s1 = torch.tensor([0.9]), torch.tensor([1.2])
    sc, sh = (torch.tensor(1.3), torch.tensor(-0.2)) if affine else (None, None)
    c1, c2 = T(f"mn.c{M}{d}", (M, d)), T(f"mn.e{M}{d}", (M, d))
    dbl = lambda t: leaf(t.double()) if t is not None else None
    po = [dbl(t) for t in (x0, x1, w, g, s0, s1, sc, sh)]
    xo = po[4] * po[0] + po[5] * po[1]
    if gamma:
        xo = xo * po[3]
    no = O.layernorm(xo, po[2], 0.0, eps) if mean else O.rmsnorm(xo, po[2], eps)
    if affine:
        no = po[6] * no + po[7]
    ((no * c1.double()).sum() + ((xo * c2.double()).sum() if res else 0)).backward()
    dev = lambda t: leaf(t, DEV) if t is not None else None
    pg = [dev(t) for t in (x0, x1, w, g, s0, s1, sc, sh)]
    ng, xg = ops.mixnorm([pg[0], pg[1]], [pg[4], pg[5]], pg[3], pg[2], None, pg[6], pg[7], eps, mean)
    ((ng * c1.to(DEV)).sum() + ((xg * c2.to(DEV)).sum() if res else 0)).backward()
    with torch.no_grad():   # the unfused chain
        x2 = ops.lincomb([pg[0].detach(), pg[1].detach()], [pg[4].detach(), pg[5].detach()], pg[3].detach() if gamma else None)
        n2 = ops.rownorm(x2, pg[2].detach(), None, pg[6].detach() if affine else None, pg[7].detach() if affine else None, eps, mean)
    assert torch.equal(xg, x2), "mix differs from lincomb_fwd"
    assert_close(ng, n2, 1e-6, "xn vs the two-kernel chain")
    assert_close(xg, xo, OUT_TOL, "x")
    assert_close(ng, no, OUT_TOL, "xn")
    for name, a, o in zip(("dx0", "dx1", "dw", "dgamma", "ds0", "ds1", "dscale", "dshift"), pg, po):
        if a is not None:
            assert_close(a.grad, o.grad, GRAD_TOL, name, atol=1e-4)


def test_rownorm_rejects_bad_shape():
    x = torch.zeros(4, 6, device=DEV)
    with pytest.raises(RuntimeError, match="multiple of 4"):
        ops.rownorm(x, torch.ones(6, device=DEV), None, None, None, 1e-5, True)


# ------------------------------------------------------------------------------------------- SSD (K1)
def test_ssd_golden_single_group():
    z = load_npz("k1_single_group")
    dt_raw = torch.log(torch.expm1(z["dt"]))  # kernel applies softplus(dt_raw + bias); fixture holds dt itself
    y = ops.ssd_reduce(z["x"].to(DEV), z["B"].to(DEV), z["C"].to(DEV), dt_raw.to(DEV), torch.zeros(8, device=DEV),
                       torch.log(-z["A"]).to(DEV), z["D"].to(DEV), 1)
    assert_close(y, z["y"], OUT_TOL, "k1 y vs reference")


def test_ssd_golden_grouped():
    z = load_npz("k1_grouped")
    dt_raw = torch.log(torch.expm1(z["dt"]))
    y = ops.ssd_reduce(z["x"].to(DEV), z["B"].to(DEV), z["C"].to(DEV), dt_raw.to(DEV), torch.zeros(8, device=DEV),
                       torch.log(-z["A"]).to(DEV), z["D"].to(DEV), 2)
    assert_close(y, z["y"], OUT_TOL, "k1 grouped y vs reference")


@pytest.mark.parametrize("B,L,H,P,N,G", [
    (2, 300, 16, 4, 16, 2),     # refiner-like (both halves as 2 groups), ragged L
    (4, 16384, 16, 4, 16, 2),   # config-2 refiner shape
    (2, 256, 64, 4, 16, 1),     # enc4-like
    (2, 16, 512, 4, 16, 2),     # decoder1-like: H spans 8 head blocks
    (1, 70, 12, 4, 16, 2),      # H not a power of two
    (1, 33, 24, 8, 8, 1),       # headdim 8
    (3, 1, 4, 4, 8, 4),         # single token
])
def test_ssd_fwd_bwd(B, L, H, P, N, G):
    x, Bm, Cm = T("s.x", (B, L, H, P)), T("s.B", (B, L, G * N)), T("s.C", (B, L, G * N))
    dt_raw, bias = T("s.dt", (B, L, H), 2.0) - 3.0, T("s.bias", (H,), 0.5)
    A_log, D = T("s.A", (H,), 1.0) + 1.0, 1 + 0.1 * T("s.D", (H,))
    cot = T("s.cot", (B, L, H, P))
    ins = [x, Bm, Cm, dt_raw, bias, A_log, D]
    o = [leaf(t.double()) for t in ins]
    dt = F.softplus(o[3] + o[4])
    yo, kvo = O.ssd_reduce(o[0], dt, torch.exp(o[5]), o[1], o[2], o[6], groups=G)
    (yo * cot.double()).sum().backward()
    g = [leaf(t, DEV) for t in ins]
    yg = ops.ssd_reduce(*g, G)
    (yg * cot.to(DEV)).sum().backward()
    assert_close(yg, yo, OUT_TOL, "y")
    for name, a, b_ in zip(["dx", "dB", "dC", "ddt", "dbias", "dA_log", "dD"], g, o):
        assert_close(a.grad, b_.grad, GRAD_TOL, name, atol=1e-6)


def test_ssd_linearity_full_size():
    """Size-independent property at config-2 size: y is linear in x (for fixed B, C, dt)."""
    B, L, H, P, N, G = 4, 16384, 16, 4, 16, 2
    gen = torch.Generator(device="cpu").manual_seed(1)
    x1, x2 = torch.randn(B, L, H, P, generator=gen).to(DEV), torch.randn(B, L, H, P, generator=gen).to(DEV)
    Bm, Cm = torch.randn(B, L, G * N, generator=gen).to(DEV), torch.randn(B, L, G * N, generator=gen).to(DEV)
    dt = torch.randn(B, L, H, generator=gen).to(DEV)
    bias, A_log, D = torch.zeros(H, device=DEV), torch.zeros(H, device=DEV), torch.ones(H, device=DEV)
    f = lambda x: ops.ssd_reduce(x, Bm, Cm, dt, bias, A_log, D, G)
    assert_close(f(x1 + 2 * x2), f(x1) + 2 * f(x2), 1e-5, "linearity")
    assert torch.equal(f(x1), f(x1)), "bitwise reproducible"


def test_ssd_rejects_unsupported():
    with pytest.raises(RuntimeError, match="not in"):
        ops.ssd_reduce(torch.zeros(1, 4, 2, 16, device=DEV), torch.zeros(1, 4, 16, device=DEV), torch.zeros(1, 4, 16, device=DEV),
                       torch.zeros(1, 4, 2, device=DEV), torch.zeros(2, device=DEV), torch.zeros(2, device=DEV), torch.zeros(2, device=DEV), 1)


# ------------------------------------------------------------------------------------------- depthwise conv
@pytest.mark.parametrize("B,H,W,C,K,act,bias", [
    (2, 12, 12, 64, 3, lib.ACT_SILU, False),
    (4, 128, 128, 128, 3, lib.ACT_SILU, False),   # refiner xBC conv
    (1, 7, 9, 8, 3, lib.ACT_NONE, True),
    (2, 4, 4, 4096, 3, lib.ACT_NONE, True),       # deep FFN dwconv
    (2, 10, 14, 32, 5, lib.ACT_NONE, False),      # wavelet-domain 5x5
    (1, 5, 3, 12, 5, lib.ACT_GELU, True),
    (1, 1, 1, 4, 3, lib.ACT_SILU, True),
    (1, 131, 130, 24, 3, lib.ACT_NONE, True),     # column-walker weight gradient on ragged strips (H, W not multiples of 4 / SEG)
    (3, 128, 129, 8, 3, lib.ACT_GELU, False),
])
def test_dwconv(B, H, W, C, K, act, bias):
    x, w = T("dw.x", (B, H * W, C)), T("dw.w", (C, 1, K, K), 0.5)
    b = T("dw.b", (C,), 0.3) if bias else None
    cot = T("dw.c", (B, H * W, C))
    xo, wo = leaf(x.double()), leaf(w.double())
    bo = leaf(b.double()) if bias else None
    pre = F.conv2d(O.img(xo, H, W), wo, bo, padding=K // 2, groups=C)
    yo = O.seq({lib.ACT_NONE: lambda t: t, lib.ACT_SILU: O.silu, lib.ACT_GELU: O.gelu}[act](pre))
    (yo * cot.double()).sum().backward()
    xg, wg = leaf(x, DEV), leaf(w, DEV)
    bg = leaf(b, DEV) if bias else None
    yg = ops.dwconv(xg, wg, bg, H, W, act)
    (yg * cot.to(DEV)).sum().backward()
    assert_close(yg, yo, OUT_TOL, "y")
    assert_close(xg.grad, xo.grad, GRAD_TOL, "dx")
    assert_close(wg.grad, wo.grad, GRAD_TOL, "dw")
    if bias:
        assert_close(bg.grad, bo.grad, GRAD_TOL, "db")


# ------------------------------------------------------------------------------------------- WTConv2d (K3)
def _wt_args(params, C, Cp, levels, K, dev):
    """Fold base_scale / wavelet_scale into tap-major taps, zero-padding C -> Cp channels."""
    pad = lambda t, n: torch.cat([t, t.new_zeros((n - t.shape[0],) + tuple(t.shape[1:]))], 0) if n > t.shape[0] else t
    bw = params["base_conv.weight"] * params["base_scale.weight"].reshape(C, 1, 1, 1)
    bb = params["base_conv.bias"] * params["base_scale.weight"].reshape(C) if "base_conv.bias" in params else None
    base_wt = ops.tap_major(pad(bw, Cp)).to(dev)
    base_b = pad(bb, Cp).to(dev) if bb is not None else None
    lws = []
    for i in range(levels):
        w = params[f"wavelet_convs.{i}.weight"] * params[f"wavelet_scale.{i}.weight"].reshape(4 * C, 1, 1, 1)
        lws.append(ops.tap_major(pad(w, 4 * Cp)).to(dev))
    return base_wt, base_b, lws


@pytest.mark.parametrize("name,levels", [("wtconv_c5_l3_16x16", 3), ("wtconv_c8_l2_20x28", 2), ("wtconv_c4_l3_11x13_k3", 3)])
def test_wtconv_golden(name, levels):
    params, grads, ins, gins, outs, cots = load_case(name)
    x = ins["x"]
    B, C, H, W = x.shape
    K = params["base_conv.weight"].shape[-1]
    Cp = (C + 3) // 4 * 4
    xt = F.pad(O.seq(x), (0, Cp - C)).to(DEV).requires_grad_(True)
    base_wt, base_b, lws = _wt_args(params, C, Cp, levels, K, DEV)
    base_wt.requires_grad_(True)
    for t in lws:
        t.requires_grad_(True)
    if base_b is not None:
        base_b.requires_grad_(True)
    y = ops.wtconv(xt, H, W, K, base_wt, base_b, lws)
    yo = O.seq(outs[0])
    assert_close(y[..., :C], yo, OUT_TOL, "wtconv out vs reference")
    (y[..., :C] * O.seq(cots[0]).to(DEV)).sum().backward()
    assert_close(xt.grad[..., :C], O.seq(gins["x"]), GRAD_TOL, "wtconv dx vs reference")
    # tap gradients -> reference parameter gradients (chain rule through the folded scale)
    gw = base_wt.grad.t().reshape(Cp, 1, K, K)[:C].cpu()
    sc = params["base_scale.weight"].reshape(C, 1, 1, 1)
    assert_close(gw * sc, grads["base_conv.weight"], GRAD_TOL, "d base_conv.weight")
    dscale = (gw * params["base_conv.weight"]).sum((1, 2, 3))
    if base_b is not None:
        gb = base_b.grad[:C].cpu()
        dscale = dscale + gb * params["base_conv.bias"]
        assert_close(gb * sc.reshape(C), grads["base_conv.bias"], GRAD_TOL, "d base_conv.bias")
    assert_close(dscale, grads["base_scale.weight"].reshape(C), GRAD_TOL, "d base_scale", atol=1e-5)
    for i in range(levels):
        g = lws[i].grad.t().reshape(4 * Cp, 1, K, K)[: 4 * C].cpu()
        assert_close(g * params[f"wavelet_scale.{i}.weight"].reshape(4 * C, 1, 1, 1), grads[f"wavelet_convs.{i}.weight"], GRAD_TOL,
                     f"d wavelet_convs.{i}")


def test_haar_roundtrip_full_size():
    """IDWT(DWT(x)) == x at config-2 size (the Haar pair is orthogonal), incl. an odd size."""
    for B, H, W, C in ((4, 128, 128, 32), (2, 33, 47, 8)):
        x = torch.randn(B * H * W, C, device=DEV)
        s = ops.k_haar_dwt(x, B, H, W, C)
        back = ops.k_haar_idwt(s, None, B, H, W, C)
        assert_close(back, x, 1e-6, "haar round trip")


@pytest.mark.parametrize("B,H,W,C,levels,dtype", [(2, 128, 128, 32, 3, torch.float32), (1, 37, 51, 8, 3, torch.float32), (2, 19, 23, 12, 2, torch.float32),
                                                  (1, 45, 30, 4, 5, torch.float32), (2, 64, 64, 16, 3, torch.bfloat16)])
def test_haar_synthesis_cascade(B, H, W, C, levels, dtype):
    """the synthesis cascade (three levels per launch, LL addends derived from the coarser levels' sub-bands) against the chain of
    single-level launches: bitwise, odd sizes and both storage types included; the two optional output addends ride in the finest launch."""
    shapes, bands, (h, w) = [], [], (H, W)
    for i in range(levels):
        shapes.append((h, w))
        h, w = (h + 1) // 2, (w + 1) // 2
        bands.append(T(f"hs.{i}.{H}", (B * h * w, 4 * C)).to(DEV).to(dtype))
    adds = (T(f"hs.a.{H}", (B * H * W, C)).to(DEV).to(dtype), T(f"hs.b.{H}", (B * H * W, C)).to(DEV).to(dtype))
    nxt = None
    for i in range(levels - 1, -1, -1):
        nxt = ops.k_haar_idwt(bands[i], nxt, B, shapes[i][0], shapes[i][1], C, y_add=adds if i == 0 else ())
    got = ops.k_haar_synthesis(bands, shapes, B, C, y_add=adds)
    assert torch.equal(got, nxt)


# ------------------------------------------------------------------------------------------- InstanceNorm
@pytest.mark.parametrize("B,HW,C,act", [(4, 16384, 32, lib.ACT_NONE), (2, 1024, 64, lib.ACT_GELU), (2, 16, 1024, lib.ACT_NONE), (3, 77, 12, lib.ACT_GELU)])
def test_instnorm(B, HW, C, act):
    x = T("in.x", (B, HW, C), 2.0) + 3.0 * T("in.m", (1, 1, C))
    cot = T("in.c", (B, HW, C))
    sc, sh = torch.tensor(0.9), torch.tensor(0.15)
    xo, sco, sho = leaf(x.double()), leaf(sc.double()), leaf(sh.double())
    mu = xo.mean(1, keepdim=True)
    var = ((xo - mu) ** 2).mean(1, keepdim=True)
    yo = sco * (xo - mu) * torch.rsqrt(var + 1e-5) + sho
    if act == lib.ACT_GELU:
        yo = O.gelu(yo)
    (yo * cot.double()).sum().backward()
    xg, scg, shg = leaf(x, DEV), leaf(sc, DEV), leaf(sh, DEV)
    yg = ops.instnorm(xg, scg, shg, 1e-5, act)
    (yg * cot.to(DEV)).sum().backward()
    assert_close(yg, yo, OUT_TOL, "y")
    assert_close(xg.grad, xo.grad, GRAD_TOL, "dx", atol=1e-7)
    assert_close(scg.grad, sco.grad, GRAD_TOL, "dscale", atol=1e-4)
    assert_close(shg.grad, sho.grad, GRAD_TOL, "dshift", atol=1e-4)


def test_gate():
    h, cot = T("g.h", (1000, 256), 3.0), T("g.c", (1000, 128))
    ho = leaf(h.double())
    yo = O.gelu(ho[:, :128]) * torch.sigmoid(ho[:, 128:])
    (yo * cot.double()).sum().backward()
    hg = leaf(h, DEV)
    yg = ops.gate(hg)
    (yg * cot.to(DEV)).sum().backward()
    assert_close(yg, yo, OUT_TOL, "gate")
    assert_close(hg.grad, ho.grad, GRAD_TOL, "dgate")


@pytest.mark.parametrize("B,H,W,d,bias", [(2, 16, 16, 32, True), (1, 9, 13, 16, False), (2, 32, 32, 64, True)])
def test_feedforward_chain(B, H, W, d, bias):
    """FeedForward (reference model_untils.py:179-198): 1x1 d->4d, depthwise 3x3, gelu(x1)*sigmoid(x2), 1x1 2d->d as ONE autograd node (FeedForwardFn);
    against the float64 statement of the same chain."""
    x = T("ffg.x", (B, H * W, d))
    w_in, w_out, dw = T("ffg.wi", (4 * d, d), 0.3), T("ffg.wo", (d, 2 * d), 0.3), T("ffg.dw", (4 * d, 1, 3, 3), 0.5)
    bs = [T("ffg.bi", (4 * d,), 0.2), T("ffg.bd", (4 * d,), 0.2), T("ffg.bo", (d,), 0.2)] if bias else [None, None, None]
    cot = T("ffg.c", (B, H * W, d))

    def ref(x, w_in, dw, w_out, b_in, b_dw, b_out):
        h = F.linear(x, w_in, b_in).view(B, H, W, 4 * d).permute(0, 3, 1, 2)
        h = F.conv2d(h, dw, b_dw, padding=1, groups=4 * d)
        x1, x2 = h.chunk(2, dim=1)
        g = (O.gelu(x1) * torch.sigmoid(x2)).permute(0, 2, 3, 1).reshape(B, H * W, 2 * d)
        return F.linear(g, w_out, b_out)

    po = [leaf(t.double()) for t in (x, w_in, dw, w_out)] + [leaf(t.double()) if t is not None else None for t in bs]
    yo = ref(*po)
    (yo * cot.double()).sum().backward()
    pg = [leaf(t, DEV) for t in (x, w_in, dw, w_out)] + [leaf(t, DEV) if t is not None else None for t in bs]
    yg = ops.feedforward(pg[0], pg[1], pg[4], pg[2], pg[5], pg[3], pg[6], H, W)
    (yg * cot.to(DEV)).sum().backward()
    assert_close(yg, yo, OUT_TOL, "ffn y")
    for name, g, o in zip(("dx", "dw_in", "ddw", "dw_out", "db_in", "db_dw", "db_out"), pg, po):
        if g is not None:
            assert_close(g.grad, o.grad, GRAD_TOL, name, atol=1e-5)


# ------------------------------------------------------------------------------------------- bf16 storage
def test_bf16_storage_paths():
    B, L, H, P, N, G = 2, 512, 16, 4, 16, 2
    x, Bm, Cm = T("b.x", (B, L, H, P)), T("b.B", (B, L, G * N)), T("b.C", (B, L, G * N))
    dt_raw, bias, A_log, D = T("b.dt", (B, L, H)) - 3, torch.zeros(H), torch.ones(H), torch.ones(H)
    yo, _ = O.ssd_reduce(x, F.softplus(dt_raw), torch.exp(A_log), Bm, Cm, D, groups=G)
    h16 = lambda t: t.to(DEV).bfloat16()
    yg = ops.ssd_reduce(h16(x), h16(Bm), h16(Cm), h16(dt_raw), bias.to(DEV), A_log.to(DEV), D.to(DEV), G)
    assert yg.dtype == torch.bfloat16
    assert_close(yg.float(), yo, 2e-2, "bf16 ssd")
    xr, w = T("b.rx", (333, 64)), torch.ones(64)
    yr = ops.rownorm(h16(xr), w.to(DEV), None, None, None, 1e-6, False)
    assert_close(yr.float(), O.rmsnorm(xr, w, 1e-6), 2e-2, "bf16 rmsnorm")


# ------------------------------------------------------------------------------------------- fused mixes
@pytest.mark.parametrize("M,C,K,use_gamma", [(4 * 16384, 32, 2, False), (1000, 32, 1, True), (64, 1024, 2, True), (333, 64, 3, True), (7, 2048, 2, False)])
def test_lincomb(M, C, K, use_gamma):
    xs = [T(f"lc.x{k}", (M, C)) for k in range(K)]
    ss = [torch.tensor([0.7 + 0.3 * k]) for k in range(K)]
    ss[0] = None if K > 1 else ss[0]
    gamma = 1 + 0.2 * T("lc.g", (C,)) if use_gamma else None
    cot = T("lc.c", (M, C))
    xo = [leaf(x.double()) for x in xs]
    so = [leaf(s.double()) if s is not None else None for s in ss]
    go = leaf(gamma.double()) if use_gamma else None
    yo = sum((x if s is None else s * x) for x, s in zip(xo, so))
    yo = yo * go if use_gamma else yo
    (yo * cot.double()).sum().backward()
    xg = [leaf(x, DEV) for x in xs]
    sg = [leaf(s, DEV) if s is not None else None for s in ss]
    gg = leaf(gamma, DEV) if use_gamma else None
    yg = ops.lincomb(xg, sg, gg)
    (yg * cot.to(DEV)).sum().backward()
    assert_close(yg, yo, OUT_TOL, "y")
    for k in range(K):
        assert_close(xg[k].grad, xo[k].grad, GRAD_TOL, f"dx{k}")
        if ss[k] is not None:
            assert_close(sg[k].grad, so[k].grad, GRAD_TOL, f"ds{k}", atol=1e-4)
    if use_gamma:
        assert_close(gg.grad, go.grad, GRAD_TOL, "dgamma", atol=1e-5)


@pytest.mark.parametrize("B,H,W,C,kh,kw,stride", [(2, 16, 16, 32, 2, 2, 2), (4, 128, 128, 32, 2, 2, 2), (2, 9, 7, 8, 2, 2, 2),
                                                   (2, 8, 8, 16, 3, 1, 1), (2, 8, 8, 16, 1, 3, 1), (1, 5, 6, 12, 3, 3, 1), (2, 4, 4, 1024, 3, 3, 1)])
def test_maxpool(B, H, W, C, kh, kw, stride):
    x, = (T("mp.x", (B, H * W, C)),)
    pad = (0, 0) if stride > 1 else (kh // 2, kw // 2)
    xo = leaf(x.double())
    yo = O.seq(F.max_pool2d(O.img(xo, H, W), (kh, kw), stride, pad))
    cot = T("mp.c", tuple(yo.shape))
    (yo * cot.double()).sum().backward()
    xg = leaf(x, DEV)
    yg = ops.maxpool(xg, H, W, kh, kw, stride)
    (yg * cot.to(DEV)).sum().backward()
    assert torch.equal(yg.cpu().double(), yo.detach()), "max pooling is exact"
    assert_close(xg.grad, xo.grad, 1e-6, "dx")


def test_maxpool_tap():
    """DownSample's pool with the input handed back as an alias for the skip connection: the skip's gradient is added inside the pool's
    backward kernel; a pooled output that needs no gradient leaves the alias gradient untouched."""
    B, H, W, C = 2, 16, 12, 32
    x, c1, c2 = T("mpt.x", (B, H * W, C)), T("mpt.c1", (B, (H // 2) * (W // 2), C)), T("mpt.c2", (B, H * W, C))
    xo = leaf(x.double())
    yo = O.seq(F.max_pool2d(O.img(xo, H, W), 2, 2))
    ((yo * c1.double()).sum() + (xo * c2.double()).sum()).backward()
    xg = leaf(x, DEV)
    yg, xa = ops.maxpool(xg, H, W, 2, 2, 2, tap=True)
    ((yg * c1.to(DEV)).sum() + (xa * c2.to(DEV)).sum()).backward()
    assert torch.equal(yg.cpu().double(), yo.detach())
    assert_close(xg.grad, xo.grad, 1e-6, "dx = pool gradient + skip gradient")
    xh = leaf(x, DEV)
    _, xb = ops.maxpool(xh, H, W, 2, 2, 2, tap=True)
    (xb * c2.to(DEV)).sum().backward()
    assert_close(xh.grad, c2, 1e-7, "alias-only gradient")


# ------------------------------------------------------------------------------------------- tall-skinny MFMA GEMMs (K6)
@pytest.mark.parametrize("M,K,N,bias", [
    (4 * 16384, 32, 208, False),   # refiner in_proj
    (4 * 16384, 128, 32, False),   # refiner out_proj
    (4 * 16384, 32, 128, True),    # FFN project_in
    (4 * 16384, 64, 32, True),     # FFN project_out
    (4 * 16384, 64, 20, False),    # OutProj 1x1 (N not a multiple of 16)
    (16384, 128, 256, True),       # decoder5 Mlp.fc1
    (16384, 256, 128, True),       # decoder5 Mlp.fc2 (dW falls back to the library)
    (4099, 48, 40, True),          # ragged M, odd block counts
])
def test_tsgemm_linear(M, K, N, bias):
    x, w = T("ts.x", (M, K)), T("ts.w", (N, K), 0.3)
    b = T("ts.b", (N,), 0.2) if bias else None
    cot = T("ts.c", (M, N))
    xo, wo = leaf(x.double()), leaf(w.double())
    bo = leaf(b.double()) if bias else None
    yo = F.linear(xo, wo, bo)
    (yo * cot.double()).sum().backward()
    xg, wg = leaf(x, DEV), leaf(w, DEV)
    bg = leaf(b, DEV) if bias else None
    yg = ops.linear(xg, wg, bg)
    (yg * cot.to(DEV)).sum().backward()
    assert_close(yg, yo, 1e-5, "y")
    assert_close(xg.grad, xo.grad, 1e-5, "dx")
    assert_close(wg.grad, wo.grad, 1e-4, "dw")
    if bias:
        assert_close(bg.grad, bo.grad, 1e-4, "db")


def test_tsgemm_is_taken_for_refiner_shapes():
    assert lib.query("adnm_tsgemm_supported", 65536, 208, 32) == 1 and lib.query("adnm_tsgemm_tn_supported", 65536, 208, 32) == 1
    assert lib.query("adnm_tsgemm_supported", 65536, 32, 208) == 1     # dX of in_proj: K' = 208 = 13 x 16
    assert lib.query("adnm_tsgemm_supported", 64, 4672, 1024) == 0     # deep levels stay on the library GEMM


# ------------------------------------------------------------------------------------------- K1b chunked scan (parity unpinned)
@pytest.mark.parametrize("B,L,H,N,G,chunk,reverse", [(2, 70, 8, 8, 2, 16, False), (2, 70, 8, 8, 2, 16, True), (1, 257, 4, 16, 1, 64, False),
                                                     (2, 64, 32, 8, 2, 256, True), (1, 33, 2, 16, 2, 8, False)])
def test_ssd_scan_vs_sequential_oracle(B, L, H, N, G, chunk, reverse):
    """fp32 HIP chunked scan vs the oracle's sequential fp64 recurrence (the reference's arithmetic is un-vendored)."""
    P = 4
    x, Bm, Cm = T("sc.x", (B, L, H, P)), T("sc.B", (B, L, G * N)), T("sc.C", (B, L, G * N))
    dt_raw, bias = T("sc.dt", (B, L, H), 2.0) - 2.0, T("sc.bias", (H,), 0.5)
    A_log, D = T("sc.A", (H,), 1.0) + 1.0, 1 + 0.1 * T("sc.D", (H,))
    cot = T("sc.cot", (B, L, H, P))
    ins = [x, Bm, Cm, dt_raw, bias, A_log, D]
    o = [leaf(t.double()) for t in ins]
    dt = F.softplus(o[3] + o[4])
    flip = (lambda t: t.flip(1)) if reverse else (lambda t: t)
    yo = flip(O.ssd_chunk_scan(flip(o[0]), flip(dt), -torch.exp(o[5]), flip(o[1]), flip(o[2]), o[6], G))
    (yo * cot.double()).sum().backward()
    g = [leaf(t, DEV) for t in ins]
    yg = ops.ssd_scan(*g, G, chunk, reverse)
    (yg * cot.to(DEV)).sum().backward()
    assert_close(yg, yo, OUT_TOL, "y")
    for name, a, b_ in zip(["dx", "dB", "dC", "ddt", "dbias", "dA_log", "dD"], g, o):
        assert_close(a.grad, b_.grad, GRAD_TOL, name, atol=1e-6)


def test_ssd_scan_chunk_invariance_full_size():
    """Size-independent property at config-2 size: the result does not depend on the chunk length."""
    B, L, H, P, N, G = 2, 16384, 8, 4, 8, 2
    gen = torch.Generator().manual_seed(3)
    r = lambda *s: torch.randn(*s, generator=gen).to(DEV)
    x, Bm, Cm, dt = r(B, L, H, P), r(B, L, G * N), r(B, L, G * N), r(B, L, H) - 2
    bias, A_log, D = torch.zeros(H, device=DEV), torch.ones(H, device=DEV), torch.ones(H, device=DEV)
    y64 = ops.ssd_scan(x, Bm, Cm, dt, bias, A_log, D, G, 64, False)
    y256 = ops.ssd_scan(x, Bm, Cm, dt, bias, A_log, D, G, 256, False)
    assert_close(y64, y256, 2e-6, "chunk invariance")


def test_igate():
    x, cot = T("ig.x", (3, 100, 64), 2.0), T("ig.c", (3, 100, 64))
    e, t = torch.tensor(1.2), torch.tensor(0.15)
    xo, eo, to = leaf(x.double()), leaf(e.double()), leaf(t.double())
    yo = O.silu(eo * (xo - to))
    (yo * cot.double()).sum().backward()
    xg, eg, tg = leaf(x, DEV), leaf(e, DEV), leaf(t, DEV)
    yg = ops.igate(xg, eg, tg)
    (yg * cot.to(DEV)).sum().backward()
    assert_close(yg, yo, OUT_TOL, "igate")
    assert_close(xg.grad, xo.grad, GRAD_TOL, "dx")
    assert_close(eg.grad, eo.grad, GRAD_TOL, "denhance", atol=1e-5)
    assert_close(tg.grad, to.grad, GRAD_TOL, "dthreshold", atol=1e-5)


@pytest.mark.parametrize("B,L,C,per_token", [(4, 16, 1024, 0), (2, 64, 512, 0), (3, 256, 256, 0), (2, 77, 36, 0), (1, 4096, 32, 0), (2, 64, 16, 1)])
def test_igate_res(B, L, C, per_token):
    """EncoderToDecoder's entry IntensityGate(x + gama * res) with the (B, 1, C) bridge gate broadcast over the tokens
    (model_untils.py:761-763) vs the same maths in fp64 torch ops."""
    tag = f"igr{B}{L}{C}"
    x, r, cot = T(tag + "x", (B, L, C), 2.0), T(tag + "r", (B, L if per_token else 1, C)), T(tag + "c", (B, L, C))
    g, e, t = torch.tensor(0.8), torch.tensor(1.2), torch.tensor(0.15)
    lo = [leaf(v.double()) for v in (x, r, g, e, t)]
    yo = O.silu(lo[3] * (lo[0] + lo[2] * lo[1] - lo[4]))
    (yo * cot.double()).sum().backward()
    lg = [leaf(v, DEV) for v in (x, r, g, e, t)]
    yg = ops.igate_res(*lg)
    (yg * cot.to(DEV)).sum().backward()
    assert_close(yg, yo, OUT_TOL, "igate_res")
    for n, a, b in zip(("dx", "dres", "dgama", "denhance", "dthreshold"), lg, lo):
        assert_close(a.grad, b.grad, GRAD_TOL, n, atol=1e-5)


# ------------------------------------------------------------------------------------------- EncoderToDecoder core (K12)
@pytest.mark.parametrize("B,H,W,C", [(2, 4, 4, 256), (1, 8, 8, 512), (4, 2, 2, 1024), (2, 16, 16, 256), (1, 40, 40, 256),
                                     (1, 32, 32, 128), (2, 9, 7, 64), (1, 64, 64, 32), (3, 5, 5, 36)])
def test_skipgate(B, H, W, C):
    """pools + grouped convs + gates + mix vs the same maths in fp64 torch ops (model_untils.py:767-787).
    (1, 40, 40, 256) has > 1 pixel slice in the weight-gradient pass (partials + fold); the last four are the narrow maps of
    e2ds[3..6] (C = 128 / 64 / 32: waves that straddle pixel slices, a partly idle last chunk of conv groups) and a width that is
    no multiple of 16 groups."""
    tag = f"sg{B}{H}{C}"
    x, cot = T(tag + "x", (B, H * W, C), 1.5), T(tag + "c", (B, H * W, C))
    shapes = [(C, 4, 1, 3), (C,), (C, 4, 3, 1), (C,), (C, 4, 3, 3), (C,), (C,), (C,), (C,), (C,)]
    params = [T(f"{tag}p{i}", s, 0.5 if len(s) > 1 else 0.3) for i, s in enumerate(shapes)]
    params[6], params[8] = 1 + params[6], 1 + params[8]
    params += [torch.tensor(v) for v in (1.3, 0.1, 0.8, -0.05, 0.33, 0.4, 0.27)] + [1 + 0.2 * T(tag + "g", (C,))]

    def ref(x, p):
        xi = x.reshape(B, H, W, C).permute(0, 3, 1, 2)
        ys = []
        for k, (win, pad, cpad) in enumerate((((3, 1), (1, 0), (0, 1)), ((1, 3), (0, 1), (1, 0)), ((3, 3), (1, 1), (1, 1)))):
            pool = F.max_pool2d(xi, win, 1, pad) + F.avg_pool2d(xi, win, 1, pad)
            c = F.conv2d(pool, p[2 * k], p[2 * k + 1], 1, cpad, 1, C // 4)
            f = k >> 1
            z = xi * F.gelu(c) * p[6 + 2 * f].view(1, -1, 1, 1) + p[7 + 2 * f].view(1, -1, 1, 1)
            ys.append(F.silu(p[10 + 2 * f] * (z - p[11 + 2 * f])))
        out = (p[14] * ys[0] + p[15] * ys[1] + p[16] * ys[2]) * p[17].view(1, -1, 1, 1)
        return out.permute(0, 2, 3, 1).reshape(B, H * W, C)

    xo, po = leaf(x.double()), [leaf(t.double()) for t in params]
    yo = ref(xo, po)
    (yo * cot.double()).sum().backward()
    xg, pg = leaf(x, DEV), [leaf(t, DEV) for t in params]
    yg = ops.skipgate(xg, H, W, pg)
    (yg * cot.to(DEV)).sum().backward()
    assert_close(yg, yo, OUT_TOL, "skipgate out")
    assert_close(xg.grad, xo.grad, GRAD_TOL, "dx")
    names = ["w13", "b13", "w31", "b31", "w33", "b33", "ffd13.w", "ffd13.b", "ffd33.w", "ffd33.b", "enh13", "thr13", "enh33", "thr33",
             "alpha1", "alpha2", "alpha3", "gamma"]
    for n, a, b in zip(names, pg, po):
        assert_close(a.grad, b.grad, GRAD_TOL, "d" + n, atol=1e-5)


@pytest.mark.parametrize("M,d,with_f", [(4 * 1024, 128, True), (300, 64, False), (64, 512, True)])
def test_catmix(M, d, with_f):
    x, r, f, cot = (T(f"cm.{n}{M}", (2, M // 2, d)) for n in "xrfc")
    cot = T(f"cm.cot{M}", (2, M // 2, 2 * d))
    al = [torch.tensor(v) for v in (1.1, 0.9, 0.7, -0.4)]

    def run(x, r, f, al, fn):
        y = fn(x, r, f if with_f else None, al[0], al[1], al[2] if with_f else None, al[3] if with_f else None)
        (y * cot.to(y)).sum().backward()
        return y

    def ref(x, r, f, a1, a2, a3, a4):
        y = torch.cat((a1 * x, a2 * r), -1)
        return y if f is None else y + torch.cat((a3 * f, a4 * f), -1)
    xo, ro, fo, ao = leaf(x.double()), leaf(r.double()), leaf(f.double()), [leaf(a.double()) for a in al]
    yo = run(xo, ro, fo, ao, ref)
    xg, rg, fg, ag = leaf(x, DEV), leaf(r, DEV), leaf(f, DEV), [leaf(a, DEV) for a in al]
    yg = run(xg, rg, fg, ag, ops.catmix)
    assert_close(yg, yo, OUT_TOL, "catmix")
    assert_close(xg.grad, xo.grad, GRAD_TOL, "dx")
    assert_close(rg.grad, ro.grad, GRAD_TOL, "dr")
    if with_f:
        assert_close(fg.grad, fo.grad, GRAD_TOL, "df")
    for i in range(4 if with_f else 2):
        assert_close(ag[i].grad, ao[i].grad, GRAD_TOL, f"da{i + 1}", atol=1e-5)


@pytest.mark.parametrize("mean", [False, True])
def test_rownorm_tap(mean):
    """pre-norm residual head: (norm(x), x) with the residual gradient added inside the norm's backward kernel."""
    M, d = 1000, 64
    x, w, c1, c2 = T("rt.x", (M, d), 2.0), 1 + 0.2 * T("rt.w", (d,)), T("rt.c1", (M, d)), T("rt.c2", (M, d))
    sc, sh = torch.tensor(1.3), torch.tensor(-0.2)
    xo, wo, so, ho = leaf(x.double()), leaf(w.double()), leaf(sc.double()), leaf(sh.double())
    yo = O.biasfree_layernorm(xo, wo) if mean else O.rmsnorm(xo, wo, 1e-5)
    ((so * yo + ho) * c1.double()).sum().backward(retain_graph=True)
    (xo * c2.double()).sum().backward()
    xg, wg, sg, hg = leaf(x, DEV), leaf(w, DEV), leaf(sc, DEV), leaf(sh, DEV)
    yn, xr = ops.rownorm_tap(xg, wg, None, sg, hg, 1e-5, mean)
    ((yn * c1.to(DEV)).sum() + (xr * c2.to(DEV)).sum()).backward()
    assert_close(xg.grad, xo.grad, GRAD_TOL, "dx (norm + residual)")
    assert_close(wg.grad, wo.grad, GRAD_TOL, "dw")
    assert_close(sg.grad, so.grad, GRAD_TOL, "dscale")


# ------------------------------------------------------------------------------------------- short GEMMs (K6b)
@pytest.mark.parametrize("M,K,N,bias", [
    (64, 2048, 1024, True),     # deepest Mamba2.out_proj: slices through partials + fold
    (64, 1024, 4672, False),    # deepest in_proj
    (256, 512, 2368, False),
    (1024, 256, 1216, True),    # no split: direct stores
    (4, 2144, 256, True),       # Channel_Att_Bridge.att*: 4 rows
    (100, 48, 36, True),        # ragged tiles
    (300, 20, 64, False),       # ragged reduction (K = 20: a chunk of 16 and a quad) forward, ragged N in the input gradient
    (128, 40, 24, True),
    (4096, 128, 544, False),
    (64, 4096, 1024, True),     # long reduction, few tiles: the LDS-tiled kernel split over workgroups, slabs combined in the launch
    (64, 1024, 4672, False),    # 73 column tiles forward; 4672-step reduction in the input gradient
    (256, 1024, 2048, True),
    (1024, 512, 2368, True),    # >= 128 tiles: LDS-tiled, no split
    (1024, 128, 256, False),    # short reduction
    (72, 2144, 100, True),      # ragged in every direction (reduction 2144 = 67 step tiles of 32)
])
def test_skgemm_linear(M, K, N, bias):
    x, w, cot = T(f"sk.x{M}{K}", (M, K)), T(f"sk.w{N}{K}", (N, K), 0.05), T(f"sk.c{M}{N}", (M, N))
    b = T(f"sk.b{N}", (N,)) if bias else None
    xo, wo, bo = leaf(x.double()), leaf(w.double()), (leaf(b.double()) if bias else None)
    yo = F.linear(xo, wo, bo)
    (yo * cot.double()).sum().backward()
    assert lib.query("adnm_skgemm_supported", 0, M, N, K) == 1
    xg, wg, bg = leaf(x, DEV), leaf(w, DEV), (leaf(b, DEV) if bias else None)
    yg = ops.linear(xg, wg, bg)
    (yg * cot.to(DEV)).sum().backward()
    assert_close(yg, yo, OUT_TOL, "y")
    assert_close(xg.grad, xo.grad, GRAD_TOL, "dx")
    assert_close(wg.grad, wo.grad, GRAD_TOL, "dw")
    if bias:
        assert_close(bg.grad, bo.grad, GRAD_TOL, "db")


@pytest.mark.parametrize("M,K,N", [(64, 4096, 1024), (256, 4096, 1024), (64, 1024, 4672), (1024, 2048, 512), (4, 2144, 256)])
def test_skgemm_split_slab_protocols_agree(M, K, N, monkeypatch):
    """The in-launch split-K combine publishes its slabs through the CALLER's workspace: this stream's region of UNCACHED device memory
    (ops.SPLITWS; no fences around the arrival ticket) or, with ws_uncached = 0, any memory bracketed by agent-scope release / acquire
    fences: same slabs, same summation order — forward and input gradient must agree bit for bit, run to run and between the protocols."""
    x, w, cot = T(f"sp.x{M}{K}", (M, K)).to(DEV), T(f"sp.w{N}{K}", (N, K), 0.05).to(DEV), T(f"sp.c{M}{N}", (M, N)).to(DEV)
    outs = []
    for mode in ("1", "0", "1"):
        monkeypatch.setenv("ADNM_SK_UC_SLABS", mode)
        ys = [ops.k_linear(x, w, None) for _ in range(3)]
        dxs = [ops.k_linear_dx(cot, w) for _ in range(3)]
        assert all(torch.equal(ys[0], y) for y in ys) and all(torch.equal(dxs[0], d) for d in dxs)
        outs.append((ys[0], dxs[0]))
    for y, dx in outs[1:]:
        assert torch.equal(outs[0][0], y) and torch.equal(outs[0][1], dx)
    assert_close(outs[0][0], x.double() @ w.double().t(), OUT_TOL, "y")


def test_split_gemm_graphs_replay_side_by_side():
    """the split-K workspace is owned per unit of ordering.  Two captured graphs of split GEMMs, each with its own
    capture scope, replayed AT THE SAME TIME on two streams (plus eager split launches on a third) give the single-stream results bit
    for bit; a capture nobody opened a scope for (ordinary memory, zeroed counters, fenced protocol) agrees too."""
    shapes = [(64, 4096, 1024), (256, 4096, 1024), (64, 1024, 4672)]
    data = [(T(f"sg.x{M}{K}", (M, K)).to(DEV), T(f"sg.w{N}{K}", (N, K), 0.05).to(DEV), T(f"sg.c{M}{N}", (M, N)).to(DEV)) for M, K, N in shapes]
    assert all(lib.query("adnm_skgemm_ws_bytes", 0, M, N, K) > 16 for M, K, N in shapes), "these shapes are meant to split"

    def work():
        out = []
        for _ in range(4):
            for x, w, cot in data:
                out += [ops.k_linear(x, w, None), ops.k_linear_dx(cot, w)]
        return out
    ref = [t.clone() for t in work()]
    torch.cuda.synchronize()
    graphs = []
    for scoped in (True, True, False):
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        scope = ops.SPLITWS.open_scope(torch.device(DEV)) if scoped else None
        g = torch.cuda.CUDAGraph()
        with torch.cuda.stream(s):
            work()
            with ops.SPLITWS.capturing(scope), torch.cuda.graph(g, stream=s):
                outs = work()
        torch.cuda.synchronize()
        graphs.append((g, s, outs, scope))
    eager_stream = torch.cuda.Stream()
    for rep in range(5):
        for _, _, outs, _ in graphs:
            for t in outs:
                t.fill_(float("nan"))
        torch.cuda.synchronize()
        for g, s, _, _ in graphs:
            with torch.cuda.stream(s):
                g.replay()
        with torch.cuda.stream(eager_stream):
            eager = work()
        torch.cuda.synchronize()
        for k, (_, _, outs, _) in enumerate(graphs):
            assert all(torch.equal(a, b) for a, b in zip(outs, ref)), f"graph {k}, replay {rep}"
        assert all(torch.equal(a, b) for a, b in zip(eager, ref)), f"eager beside the graphs, replay {rep}"


def test_skgemm_strided_operands():
    """column slices of wider buffers as input and output (how the mixer calls it): no partials for the strided output."""
    M, K, N = 256, 512, 128
    big_in, big_out = T("sks.in", (M, K + 64)).to(DEV), torch.zeros((M, N + 32), device=DEV)
    w = T("sks.w", (N, K), 0.05).to(DEV)
    y = ops.k_linear(big_in[:, 32:32 + K], w, None, out=big_out[:, 16:16 + N])
    ref = big_in[:, 32:32 + K].double() @ w.double().t()
    assert_close(y, ref, OUT_TOL, "strided y")
    assert float(big_out[:, :16].abs().max()) == 0.0 and float(big_out[:, 16 + N:].abs().max()) == 0.0


@pytest.mark.parametrize("B,shapes", [(4, [(16384, 32), (4096, 64), (1024, 128), (256, 256), (64, 512), (16, 512), (4, 1024)]),
                                      (2, [(700, 64)]), (3, [(33, 8), (5, 4), (1, 12)])])
def test_bridge_pool(B, shapes):
    """Channel_Att_Bridge's avgpool of every skip + the concat as one launch each way; every skip comes back as an alias whose consumer
    gradient is summed with the pool's broadcast gradient in the one backward launch (one skip is left without another consumer)."""
    xs = [T(f"bp.x{i}.{L}", (B, L, C)) for i, (L, C) in enumerate(shapes)]
    cs = [T(f"bp.c{i}.{L}", (B, L, C)) for i, (L, C) in enumerate(shapes)]
    S = sum(C for _, C in shapes)
    c2 = T(f"bp.m{S}", (B, S))
    xo = [leaf(x.double()) for x in xs]
    atto = torch.cat([x.mean(1) for x in xo], dim=-1)
    (sum((x * c.double()).sum() for x, c in list(zip(xo, cs))[1:]) + (atto * c2.double()).sum()).backward()
    xg = [leaf(x, DEV) for x in xs]
    al, att = ops.bridge_pool(xg)
    (sum((a * c.to(DEV)).sum() for a, c in list(zip(al, cs))[1:]) + (att * c2.to(DEV)).sum()).backward()
    assert_close(att, atto, OUT_TOL, "pooled means")
    for i, (g, o) in enumerate(zip(xg, xo)):
        assert_close(g.grad, o.grad, GRAD_TOL, f"dx[{i}] = consumer gradient + broadcast pool gradient")


def test_conv1d3():
    B, n = 4, 2144
    x, cot = T("c1.x", (B, 1, n)), T("c1.c", (B, 1, n))
    w, b = torch.tensor([[[0.3, -0.8, 0.5]]]), torch.tensor([0.1])
    xo, wo, bo = leaf(x.double()), leaf(w.double()), leaf(b.double())
    (F.conv1d(xo, wo, bo, padding=1) * cot.double()).sum().backward()
    xg, wg, bg = leaf(x, DEV), leaf(w, DEV), leaf(b, DEV)
    y = ops.conv1d3(xg, wg, bg)
    (y * cot.to(DEV)).sum().backward()
    assert_close(y, F.conv1d(x.double(), w.double(), b.double(), padding=1), OUT_TOL, "y")
    assert_close(xg.grad, xo.grad, GRAD_TOL, "dx")
    assert_close(wg.grad, wo.grad, GRAD_TOL, "dw")
    assert_close(bg.grad, bo.grad, GRAD_TOL, "db")


@pytest.mark.parametrize("B,L,heads", [(4, 256, 32), (2, 64, 64), (1, 100, 3), (1, 1024, 8)])
def test_attn4(B, L, heads):
    """fused 4-wide-head attention vs the reference expression (ADNssd.py:38-46) in fp64."""
    inner, scale = heads * 4, 4 ** -0.5
    qkv, cot = T(f"at.q{L}{heads}", (B, L, 3 * inner), 1.5), T(f"at.c{L}{heads}", (B, L, inner))
    qo = leaf(qkv.double())
    q, k, v = qo.chunk(3, dim=-1)
    sp = lambda t: t.reshape(B, L, heads, 4).transpose(1, 2)
    att = torch.softmax(torch.matmul(sp(q), sp(k).transpose(-1, -2)) * scale, dim=-1)
    yo = torch.matmul(att, sp(v)).transpose(1, 2).reshape(B, L, inner)
    (yo * cot.double()).sum().backward()
    qg = leaf(qkv, DEV)
    yg = ops.attn4(qg, heads, scale)
    (yg * cot.to(DEV)).sum().backward()
    assert_close(yg, yo, OUT_TOL, "attention out")
    assert_close(qg.grad, qo.grad, GRAD_TOL, "dqkv")


def test_unsupported_shapes_raise():
    """The token ops have no PyTorch fallback: a shape a kernel does not take is an error, as a CPU tensor is."""
    one = torch.ones((), device=DEV)
    odd = torch.zeros(2, 8, 6, device=DEV)        # 6 channels: not a multiple of 4
    with pytest.raises(RuntimeError, match="no PyTorch fallback"):
        ops.catmix(odd, odd, None, one, one)
    with pytest.raises(RuntimeError, match="no PyTorch fallback"):
        ops.lincomb([odd, odd], [one, one], gamma=torch.ones(6, device=DEV))     # a per-channel gamma needs 4 | C
    with pytest.raises(RuntimeError, match="no PyTorch fallback"):
        ops.igate(torch.zeros(3, device=DEV), one, one)
    with pytest.raises(RuntimeError, match="no PyTorch fallback"):
        ops.bridge_pool([odd])
    with pytest.raises(RuntimeError, match="no PyTorch fallback"):
        ops.conv1d3(torch.zeros(2, 2, 8, device=DEV), torch.zeros(1, 1, 3, device=DEV), None)
    with pytest.raises(RuntimeError, match="adnm_hip linear"):
        ops.linear(torch.zeros(8, 10, device=DEV), torch.zeros(12, 10, device=DEV))      # K = 10: not a multiple of 4
    with pytest.raises(RuntimeError, match="GPU only"):
        ops.linear(torch.zeros(8, 16), torch.zeros(12, 16))
    from models.model_untils import DownSample
    with pytest.raises(RuntimeError, match="max-pool"):
        DownSample(dim=6)(torch.zeros(1, 16, 6, device=DEV))
    # the branches that used to fall back to torch / library ops now raise like everything else
    import torch.nn as nn
    from models.model_untils import Conv2dLayer, WTConvLayer, DeConv2dLayer, EncoderToDecoder
    from models.ADNssd import StandardAttention
    with pytest.raises(RuntimeError, match="no PyTorch fallback"):       # GroupNorm has no HIP kernel
        Conv2dLayer(8, 8, norm=nn.GroupNorm(4, 8)).to(DEV).forward_tokens(torch.zeros(1, 16, 8, device=DEV), 4, 4)
    with pytest.raises(RuntimeError, match="no PyTorch fallback"):       # shape-changing conv
        Conv2dLayer(8, 8, kernel_size=(3, 3), stride=(2, 2)).to(DEV)(torch.zeros(1, 8, 4, 4, device=DEV))
    with pytest.raises(RuntimeError, match="no PyTorch fallback"):
        WTConvLayer(8, 8, kernel_size=5, wt_levels=1, norm=nn.GroupNorm(4, 8)).to(DEV).forward_tokens(torch.zeros(1, 16, 8, device=DEV), 4, 4)
    with pytest.raises(RuntimeError, match="no PyTorch fallback"):       # ratio-4 transposed conv
        DeConv2dLayer(8, 8, ratio=4).to(DEV)(torch.zeros(1, 8, 4, 4, device=DEV))
    with pytest.raises(RuntimeError, match="no PyTorch fallback"):       # 6 channels: no conv group of 4 per lane
        EncoderToDecoder(embed_dim=8, InstanceNorm=False).to(DEV)(torch.zeros(1, 16, 8, device=DEV), torch.zeros(1, 1, 8, device=DEV))
    with pytest.raises(RuntimeError, match="no PyTorch fallback"):       # 8-wide heads
        StandardAttention(32, heads=4, dim_head=8).to(DEV)(torch.zeros(1, 16, 32, device=DEV), 4, 4)
    with pytest.raises(RuntimeError, match="no PyTorch fallback"):
        ops.skipgate(odd, 2, 4, [one] * 18)


# ------------------------------------------------------------------------------------------- dense 3x3 conv (K5)
@pytest.mark.parametrize("B,H,W,K,N,act,bias,cl", [
    (2, 16, 16, 5, 32, lib.ACT_GELU, False, False),     # PatchEmbed.conv2: ragged Cin, nn.Conv2d weight layout
    (4, 128, 128, 64, 32, lib.ACT_GELU, True, True),    # config-2 decoder6 conv, channels-last weight (the flat trainer's layout)
    (2, 8, 8, 256, 64, lib.ACT_GELU, True, True),       # deep map: split-K partials + join
    (1, 4, 4, 64, 128, lib.ACT_NONE, True, False),      # 4x4 map: a pixel block is 4 rows x 4 columns
    (2, 12, 12, 16, 24, lib.ACT_GELU, True, False),     # width not a multiple of the tile
    (1, 64, 64, 20, 20, lib.ACT_NONE, False, True),     # OutProj.conv2: 20 -> 20
    (2, 7, 9, 8, 12, lib.ACT_GELU, True, False),        # odd sizes
    (1, 32, 32, 32, 200, lib.ACT_NONE, True, True),     # several output-channel groups
])
def test_conv3(B, H, W, K, N, act, bias, cl):
    x, w, cot = T(f"c3.x{H}{K}", (B, H * W, K)), T(f"c3.w{N}{K}", (N, K, 3, 3), 0.2), T(f"c3.c{H}{N}", (B, H * W, N))
    b = T(f"c3.b{N}", (N,)) if bias else None
    xo, wo, bo = leaf(x.double()), leaf(w.double()), (leaf(b.double()) if bias else None)
    yo = F.conv2d(xo.view(B, H, W, K).permute(0, 3, 1, 2), wo, bo, padding=1)
    if act == lib.ACT_GELU:
        yo = F.gelu(yo)
    yo = yo.permute(0, 2, 3, 1).reshape(B, H * W, N)
    (yo * cot.double()).sum().backward()
    xg, bg = leaf(x, DEV), (leaf(b, DEV) if bias else None)
    wg = w.to(DEV)
    if cl:   # (Cout, 3, 3, Cin) memory order behind the logical (Cout, Cin, 3, 3) shape
        wg = wg.permute(0, 2, 3, 1).contiguous().permute(0, 3, 1, 2)
    wg = wg.detach().requires_grad_(True)
    yg = ops.conv3(xg, wg, bg, H, W, act)
    (yg * cot.to(DEV)).sum().backward()
    assert_close(yg, yo, OUT_TOL, "y")
    assert_close(xg.grad, xo.grad, GRAD_TOL, "dx")
    assert_close(wg.grad, wo.grad, GRAD_TOL, "dw")
    if bias:
        assert_close(bg.grad, bo.grad, GRAD_TOL, "db")


# ------------------------------------------------------------------------------------------- stride-2 transposed conv (K9)
@pytest.mark.parametrize("B,H,W,C,cl", [(2, 4, 4, 64, True), (4, 8, 8, 256, True), (1, 16, 16, 32, False), (2, 6, 10, 16, True), (4, 64, 64, 32, True)])
def test_convt2x(B, H, W, C, cl):
    x, w, b, cot = T(f"ct.x{H}{C}", (B, H * W, C)), T(f"ct.w{C}", (C, C, 3, 3), 0.2), T(f"ct.b{C}", (C,)), T(f"ct.c{H}{C}", (B, 4 * H * W, C))
    xo, wo, bo = leaf(x.double()), leaf(w.double()), leaf(b.double())
    yo = F.conv_transpose2d(xo.view(B, H, W, C).permute(0, 3, 1, 2), wo, bo, stride=2, padding=1, output_padding=1)
    yo = yo.permute(0, 2, 3, 1).reshape(B, 4 * H * W, C)
    (yo * cot.double()).sum().backward()
    xg, bg = leaf(x, DEV), leaf(b, DEV)
    wg = w.to(DEV)
    if cl:   # (Cin, 3, 3, Cout) memory order behind the logical (Cin, Cout, 3, 3) shape: the flat trainer's layout
        wg = wg.permute(0, 2, 3, 1).contiguous().permute(0, 3, 1, 2)
    wg = wg.detach().requires_grad_(True)
    yg = ops.convt2x(xg, wg, bg, H, W)
    (yg * cot.to(DEV)).sum().backward()
    assert_close(yg, yo, OUT_TOL, "y")
    assert_close(xg.grad, xo.grad, GRAD_TOL, "dx")
    assert_close(wg.grad, wo.grad, GRAD_TOL, "dw")
    assert_close(bg.grad, bo.grad, GRAD_TOL, "db")


# ------------------------------------------------------------------------------------------- stand-alone activations
@pytest.mark.parametrize("code,fn", [(lib.ACT_GELU, F.gelu), (lib.ACT_SILU, F.silu)])
def test_act(code, fn):
    x, cot = T("act.x", (3, 1000, 8), 3.0), T("act.c", (3, 1000, 8))
    xo = leaf(x.double())
    yo = fn(xo)
    (yo * cot.double()).sum().backward()
    xg = leaf(x, DEV)
    yg = ops.act(xg, code)
    (yg * cot.to(DEV)).sum().backward()
    assert_close(yg, yo, 1e-6, "y")
    assert_close(xg.grad, xo.grad, 1e-5, "dx")


def test_swish_learnable_beta():
    x, cot = T("sw.x", (2, 4096, 20), 3.0), T("sw.c", (2, 4096, 20))
    xo, bo = leaf(x.double()), leaf(torch.tensor(1.3).double())
    yo = xo * torch.sigmoid(bo * xo)
    (yo * cot.double()).sum().backward()
    xg, bg = leaf(x, DEV), leaf(torch.tensor(1.3), DEV)
    yg = ops.swish(xg, bg)
    (yg * cot.to(DEV)).sum().backward()
    assert_close(yg, yo, 1e-6, "y")
    assert_close(xg.grad, xo.grad, 1e-5, "dx")
    assert abs(float(bg.grad) - float(bo.grad)) <= 1e-4 * abs(float(bo.grad))


def test_lincomb_scalar_mix_any_channel_count():
    """PatchEmbed's alpha1 * conv(x) + beta1 * x on 5-channel frames (model_untils.py:306): elementwise, so it runs on the flat arrays."""
    a, b, cot = T("lc5.a", (2, 64, 5)), T("lc5.b", (2, 64, 5)), T("lc5.c", (2, 64, 5))
    s1, s2 = torch.tensor(0.7), torch.tensor(-1.2)
    o = [leaf(t.double()) for t in (a, b, s1, s2)]
    ((o[2] * o[0] + o[3] * o[1]) * cot.double()).sum().backward()
    g = [leaf(t, DEV) for t in (a, b, s1, s2)]
    y = ops.lincomb([g[0], g[1]], [g[2], g[3]])
    (y * cot.to(DEV)).sum().backward()
    assert_close(y, o[2] * o[0] + o[3] * o[1], 1e-6, "y")
    for name, u, v in zip(("da", "db", "ds1", "ds2"), g, o):
        assert_close(u.grad, v.grad, 1e-4, name)


# ------------------------------------------------------------------------------------------- bf16 MFMA precision (BASELINE's bf16 configs)
@pytest.mark.parametrize("rows,n", [(2, 16384), (5, 65536 + 4), (16, 262144), (40, 32768), (300, 20480)])
def test_fold_four_columns_per_lane_is_bitwise_the_scalar_fold(rows, n):
    """The shared second-stage fold takes large, 16-byte aligned partial sets four columns per lane (csrc/core.hip: kFoldVec); row slices,
    unroll and tree are those of the one-column walk, so every column is summed in the same order: the same bits.  The scalar walk is
    reached here through a partial set that starts 4 bytes off a 16-byte boundary."""
    buf = torch.randn(rows * n + 8, device=DEV)
    a = buf[:rows * n].view(rows, n)
    b = torch.empty(rows * n + 8, device=DEV)[1:1 + rows * n].view(rows, n)
    b.copy_(a)
    assert a.data_ptr() % 16 == 0 and b.data_ptr() % 16 == 4
    va, vb = ops.colsum(a), ops.colsum(b)
    assert torch.equal(va, vb), "vector and scalar folds must agree bit for bit"
    assert_close(va, a.double().sum(0), 2e-6, "fold")


@pytest.mark.parametrize("rows,n", [(65536, 32), (5000, 257), (2048, 4), (1500, 4), (1024, 131), (300, 3000), (7, 5), (3, 5000), (12, 3001), (20, 4000), (2, 263168)])
def test_colsum(rows, n):
    """bias-gradient column sums: two-stage above 2048 rows, the shared fold (every workgroup geometry, 4 .. 1024 columns) below"""
    x = T(f"cs.{rows}.{n}", (rows, n)).to(DEV)
    out = ops.colsum(x)
    assert_close(out, x.double().sum(0), 2e-6, "colsum")
    assert torch.equal(out, ops.colsum(x)), "colsum must be bitwise reproducible"


def _bf16_round(t):
    return t.to(torch.bfloat16).to(t.dtype)


@pytest.fixture
def bf16_mfma():
    ops.set_mfma_precision("bf16")
    yield
    ops.set_mfma_precision("f32")


@pytest.mark.parametrize("M,K,N", [(64, 1024, 512), (1024, 256, 1216), (300, 20, 64), (64, 4096, 1024), (256, 1024, 2048), (1024, 512, 2368)])
def test_skgemm_bf16_mfma(M, K, N, bf16_mfma):
    """prec = ADNM_MFMA_BF16: operands rounded to bf16 (RNE), products and sums in fp32 — i.e. EXACTLY the fp32 GEMM of the rounded
    operands (up to summation order), and within bf16 rounding (2^-9 per operand) of the unrounded one."""
    x, w, cot = T(f"bf.x{M}{K}", (M, K)), T(f"bf.w{N}{K}", (N, K), 0.05), T(f"bf.c{M}{N}", (M, N))
    xg, wg = leaf(x, DEV), leaf(w, DEV)
    yg = ops.linear(xg, wg, None)
    (yg * cot.to(DEV)).sum().backward()
    xr, wr, cr = _bf16_round(x).double(), _bf16_round(w).double(), _bf16_round(cot).double()
    assert_close(yg, xr @ wr.t(), 2e-6, "y vs the GEMM of bf16-rounded operands")
    assert_close(yg, x.double() @ w.double().t(), 1e-2, "y vs the unrounded GEMM")
    assert_close(xg.grad, cr @ wr, 2e-6, "dx")
    assert_close(wg.grad, cr.t() @ xr, 2e-6, "dw")


@pytest.mark.parametrize("M,K,N", [(65536, 32, 128), (65536, 32, 208), (65536, 128, 32), (40000, 64, 20)])
def test_tsgemm_bf16_mfma(M, K, N, bf16_mfma):
    """the full-resolution Linears in the bf16 mode: forward and input gradient on bf16 MFMA (exactly the fp32 GEMM of the bf16-rounded
    operands), the weight gradient stays exact fp32"""
    x, w, cot = T(f"tb.x{M}{K}", (M, K)), T(f"tb.w{N}{K}", (N, K), 0.05), T(f"tb.c{M}{N}", (M, N))
    b = T(f"tb.b{N}", (N,))
    xg, wg, bg = leaf(x, DEV), leaf(w, DEV), leaf(b, DEV)
    yg = ops.linear(xg, wg, bg)
    (yg * cot.to(DEV)).sum().backward()
    xr, wr, cr = _bf16_round(x).double().to(DEV), _bf16_round(w).double().to(DEV), _bf16_round(cot).double().to(DEV)
    assert_close(yg, xr @ wr.t() + b.double().to(DEV), 2e-6, "y vs the GEMM of bf16-rounded operands")
    assert_close(xg.grad, cr @ wr, 2e-6, "dx")
    assert_close(wg.grad, cot.double().to(DEV).t() @ x.double().to(DEV), 1e-5, "dw (fp32 MFMA)")
    assert_close(bg.grad, cot.double().sum(0), 1e-5, "db")


@pytest.mark.parametrize("B,H,W,K,N,act", [(2, 16, 16, 32, 64, "none"), (1, 32, 32, 5, 32, "gelu"), (2, 8, 8, 20, 20, "none"), (1, 16, 16, 128, 32, "gelu"),
                                           (4, 4, 4, 256, 64, "none"), (1, 8, 8, 144, 48, "gelu"), (2, 4, 4, 512, 128, "gelu"), (1, 24, 40, 72, 16, "none")])
def test_conv3_bf16_mfma(B, H, W, K, N, act, bf16_mfma):
    """bf16 configuration: forward and input gradient run on bf16 LDS images (csrc/conv3.hip: conv3_bf16_kernel — 32-channel chunks, one
    rounding per staged value): EXACTLY (fp32 summation order) the conv of the bf16-rounded operands.  Shapes: one / several chunks, a
    ragged last chunk (K = 5, 20, 72, 144), split reductions with the join kernel (4 x 4 maps), GELU in the epilogue / on the way in."""
    a = lib.ACT_GELU if act == "gelu" else lib.ACT_NONE
    x, w, b, cot = T(f"bfc.x{K}", (B, H * W, K)), T(f"bfc.w{K}{N}", (N, K, 3, 3), 0.2), T(f"bfc.b{N}", (N,)), T(f"bfc.c{N}{H}", (B, H * W, N))
    xg, wg, bg = leaf(x, DEV), leaf(w, DEV), leaf(b, DEV)
    yg = ops.conv3(xg, wg, bg, H, W, a)
    (yg * cot.to(DEV)).sum().backward()
    xr, wr = _bf16_round(x).double(), _bf16_round(w).double()
    xo, wo = leaf(xr), leaf(wr)
    pre = F.conv2d(xo.view(B, H, W, K).permute(0, 3, 1, 2), wo, b.double(), padding=1).permute(0, 2, 3, 1).reshape(B, H * W, N)
    assert_close(yg, F.gelu(pre) if act == "gelu" else pre, 2e-6, "y vs the conv of bf16-rounded operands")
    # backward: the kernels round dpre = cot * act'(pre) (pre = the fp32 value the forward saved) on the way in
    pre32 = pre.detach().float()
    if act == "gelu":
        pl = pre32.clone().requires_grad_(True)
        F.gelu(pl).backward(cot)
        dpre = pl.grad
    else:
        dpre = cot
    dr = _bf16_round(dpre).double()
    dxo = F.conv_transpose2d(dr.view(B, H, W, N).permute(0, 3, 1, 2), wr, padding=1).permute(0, 2, 3, 1).reshape(B, H * W, K)
    # (GELU: the reference's dpre comes from ITS pre-activation, the kernel's from the fp32 value it saved — the last-bit differences flip a
    # bf16 rounding of dpre once in ~4e4 elements, each flip 2^-9 of its element: rel-L2 up to ~5e-5; a wrong lane map would be >= 1e-2)
    gt = 2e-4 if act == "gelu" else 2e-6
    assert_close(xg.grad, dxo, gt, "dx")
    dwo = torch.nn.grad.conv2d_weight(xr.view(B, H, W, K).permute(0, 3, 1, 2), (N, K, 3, 3), dr.view(B, H, W, N).permute(0, 3, 1, 2), padding=1)
    assert_close(wg.grad, dwo, gt, "dw")
    assert_close(bg.grad, dpre.double().sum((0, 1)), 1e-5, "db (fp32 sums of the unrounded gradient)")


# ------------------------------------------------------------------------------------------- fp8 MFMA precision (BASELINE config 5)
F8MAX = {False: 448.0, True: 57344.0}


def _fp8_round(t, scale, grad=False):
    """what the kernels feed the MFMA: fp8(clamp(t * scale)) — OCP e4m3 (activations, weights) or e5m2 (gradients), RNE — as fp64"""
    q = (t.float() * scale).clamp(-F8MAX[grad], F8MAX[grad]).to(torch.float8_e5m2 if grad else torch.float8_e4m3fn)
    return q.float().double()


def _scale_for(t, grad=False):
    return F8MAX[grad] / (2.0 * float(t.abs().max()))   # one binade of headroom, as the delayed scaling leaves


@pytest.fixture
def fp8_mfma():
    ops.QUANT.reset()
    ops.set_mfma_precision("fp8")
    rows, ops.QUANT.max_rows = ops.QUANT.max_rows, 1 << 30   # kernel tests: every shape on fp8 (the model keeps bf16 above 8192 token rows)
    yield
    ops.QUANT.max_rows = rows
    ops.set_mfma_precision("f32")
    ops.QUANT.reset()


@pytest.mark.parametrize("M,K,N", [(64, 1024, 512), (1024, 256, 1216), (300, 20, 64), (64, 4096, 1024), (256, 1024, 2048), (65536, 32, 128),
                                   (65536, 128, 32), (40000, 208, 32)])
def test_linear_fp8_mfma(M, K, N, fp8_mfma):
    """prec = ADNM_MFMA_FP8: the forward GEMM is EXACTLY (up to fp32 summation order) the product of the e4m3-rounded scaled operands
    divided by the two scales; the input gradient the product of the e5m2-rounded output gradient and the e4m3 weight; the weight
    gradient stays on bf16 operands (tall-skinny shapes: exact fp32).  Covers the streaming, the LDS-tiled and the tall-skinny kernel.
    Tolerance 2e-5, not the bf16 test's 2e-6: the fp8 MFMA sums its 32 products per step in the hardware's own internal format
    (measured 6e-6 at K = 1024 against the fp64 sum of the same rounded operands); a wrong scale, format or lane map would be >= 1e-2."""
    x, w, cot = T(f"f8.x{M}{K}", (M, K)), T(f"f8.w{N}{K}", (N, K), 0.05), T(f"f8.c{M}{N}", (M, N))
    xg, wg = leaf(x, DEV), leaf(w, DEV)
    sx, sw, sc = _scale_for(x), _scale_for(w), _scale_for(cot, True)
    ops.QUANT.set(xg.device, wg.data_ptr(), "linear_fwd", sx, sw)
    ops.QUANT.set(xg.device, wg.data_ptr(), "linear_dgrad", sc, sw)
    yg = ops.linear(xg, wg, None)
    (yg * cot.to(DEV)).sum().backward()
    xq, wq, cq = _fp8_round(x, sx), _fp8_round(w, sw), _fp8_round(cot, sc, True)
    assert_close(yg, (xq @ wq.t()) / (sx * sw), 2e-5, "y vs the GEMM of the fp8-rounded operands")
    assert_close(yg, x.double() @ w.double().t(), 8e-2, "y vs the unrounded GEMM")
    assert_close(xg.grad, (cq @ wq) / (sc * sw), 2e-5, "dx vs the GEMM of the e5m2 gradient and the e4m3 weight")
    big = M >= 32768
    cr, xr = (cot.double(), x.double()) if big else (_bf16_round(cot).double(), _bf16_round(x).double())
    assert_close(wg.grad, cr.t() @ xr, 1e-5 if big else 2e-6, "dw (bf16 operands / exact fp32 for the tall-skinny shapes)")


def test_conv3_fp8_mfma(fp8_mfma):
    B, H, W, K, N = 2, 16, 16, 32, 64
    x, w, b, cot = T("f8c.x", (B, H * W, K)), T("f8c.w", (N, K, 3, 3), 0.2), T("f8c.b", (N,)), T("f8c.c", (B, H * W, N))
    xg, wg, bg = leaf(x, DEV), leaf(w, DEV), leaf(b, DEV)
    sx, sw, sc = _scale_for(x), _scale_for(w), _scale_for(cot, True)
    ops.QUANT.set(xg.device, wg.data_ptr(), "conv3_fwd", sx, sw)
    ops.QUANT.set(xg.device, wg.data_ptr(), "conv3_dgrad", sc, sw)
    yg = ops.conv3(xg, wg, bg, H, W, lib.ACT_NONE)
    (yg * cot.to(DEV)).sum().backward()
    xq, wq, cq = _fp8_round(x, sx), _fp8_round(w, sw), _fp8_round(cot, sc, True)
    conv = lambda a, ww: F.conv2d(a.view(B, H, W, -1).permute(0, 3, 1, 2), ww, None, padding=1).permute(0, 2, 3, 1).reshape(B, H * W, -1)
    assert_close(yg, conv(xq, wq) / (sx * sw) + b.double(), 2e-5, "y vs the conv of the fp8-rounded operands")
    dx_ref = F.conv_transpose2d(cq.view(B, H, W, N).permute(0, 3, 1, 2), wq, None, padding=1).permute(0, 2, 3, 1).reshape(B, H * W, K) / (sc * sw)
    assert_close(xg.grad, dx_ref, 2e-5, "dx vs the transposed conv of the e5m2 gradient and the e4m3 weight")
    xo, wo = leaf(_bf16_round(x).double()), leaf(_bf16_round(w).double())
    (conv(xo, wo) * _bf16_round(cot).double()).sum().backward()
    assert_close(wg.grad, wo.grad, 2e-6, "dw (bf16 operands)")


def test_fp8_amax_collection_and_update(fp8_mfma):
    """a calibration pass (bf16 operands, record flag set) collects exactly max |operand| on every kernel path; adnm_quant_update turns it
    into fmax / (amax * headroom), clears the amax and schedules the next calibration by the period"""
    ops.set_mfma_precision("f32")
    for M, K, N in [(64, 1024, 512), (256, 1024, 2048), (65536, 32, 128)]:
        x, w, cot = T(f"am.x{M}", (M, K), 3.0), T(f"am.w{N}", (N, K), 0.05), T(f"am.c{M}", (M, N), 0.01)
        xg, wg = leaf(x, DEV), leaf(w, DEV)
        y = ops.fp8_calibrate(xg.device, lambda: ops.linear(xg, wg, None).mul(cot.to(DEV)).sum().backward())
        tab = ops.QUANT.dump(xg.device)
        f, g = tab[(wg.data_ptr(), "fnt")], tab[(wg.data_ptr(), "gnn")]
        h = ops.QUANT.headroom
        assert abs(f[0] - 448.0 / (float(x.abs().max()) * h)) <= 1e-5 * f[0], (M, K, N, f)
        assert abs(f[1] - 448.0 / (float(w.abs().max()) * h)) <= 1e-5 * f[1], (M, K, N, f)
        assert abs(g[0] - 57344.0 / (float(cot.abs().max()) * h)) <= 1e-5 * g[0], (M, K, N, g)
        assert f[2] == 0.0 and f[3] == 0.0 and f[6] == 0.0   # amax cleared, not recording until the period comes round
        for _ in range(ops.QUANT.period - 2):
            ops.QUANT.update(xg.device)
        assert ops.QUANT.dump(xg.device)[(wg.data_ptr(), "fnt")][6] == 0.0
        ops.QUANT.update(xg.device)
        assert ops.QUANT.dump(xg.device)[(wg.data_ptr(), "fnt")][6] == 1.0   # step `period`: collect again
        ops.set_mfma_precision("f32")


@pytest.mark.parametrize("B,Cs", [(4, [256, 512, 1024]), (2, [32, 64, 128, 128, 256, 512, 1024]), (7, [36, 20]), (16, [256, 512, 1024]), (19, [36, 20])])
def test_bridge_heads(B, Cs):
    """every head of Channel_Att_Bridge (Linear + IntensityGate, model_untils.py:594-613,744-750) in one launch each way vs fp64 torch ops;
    the enhance / threshold scalars are shared by all heads"""
    S = sum(Cs) if len(Cs) == 7 else 2144
    tag = f"bh{B}{len(Cs)}"
    att, e, t = T(tag + "a", (B, 1, S)), torch.tensor(1.3), torch.tensor(0.1)
    Ws, bs = [T(f"{tag}w{i}", (c, S), 0.05) for i, c in enumerate(Cs)], [T(f"{tag}b{i}", (c,), 0.1) for i, c in enumerate(Cs)]
    cots = [T(f"{tag}c{i}", (B, 1, c)) for i, c in enumerate(Cs)]
    lo = [leaf(v.double()) for v in [att, e, t] + Ws + bs]
    n = len(Cs)
    ys = [O.silu(lo[1] * (lo[0] @ lo[3 + i].t() + lo[3 + n + i] - lo[2])) for i in range(n)]
    sum((y * c.double()).sum() for y, c in zip(ys, cots)).backward()
    lg = [leaf(v, DEV) for v in [att, e, t] + Ws + bs]
    yg = ops.bridge_heads(lg[0], lg[1], lg[2], lg[3:3 + n], lg[3 + n:])
    sum((y * c.to(DEV)).sum() for y, c in zip(yg, cots)).backward()
    for i in range(n):
        assert_close(yg[i], ys[i], OUT_TOL, f"gate {i}")
    names = ["datt", "denhance", "dthreshold"] + [f"dW{i}" for i in range(n)] + [f"db{i}" for i in range(n)]
    for nm, a, b in zip(names, lg, lo):
        assert_close(a.grad, b.grad, GRAD_TOL, nm, atol=1e-6)


def test_chanpad_roundtrip():
    x = T("cp.x", (3, 50, 5))
    xg = leaf(x, DEV)
    y = ops.chanpad(xg, 8)
    assert torch.equal(y[..., :5].cpu(), x) and float(y[..., 5:].abs().max()) == 0.0
    z = ops.chanpad(y, 5)
    assert torch.equal(z.cpu(), x)
    cot = T("cp.c", (3, 50, 5)).to(DEV)
    (z * cot).sum().backward()
    assert torch.equal(xg.grad, cot)


@pytest.mark.parametrize("M,K,N", [(65536, 32, 208), (65536, 128, 32), (40000, 208, 32)])
def test_tsgemm_bf16_token_storage(M, K, N, bf16_mfma):
    """the tall-skinny kernels with bf16 token rows in and / or out (the wide internal tensors of ADNMixerFn / FeedForwardFn at the
    full-resolution level): the result is the fp32-accumulated product of the bf16 inputs, rounded once (RNE) when stored as bf16"""
    x, w, cot = T(f"tbs.x{M}{K}", (M, K)), T(f"tbs.w{N}{K}", (N, K), 0.05), T(f"tbs.c{M}{N}", (M, N))
    xb, cb = x.to(torch.bfloat16).to(DEV), cot.to(torch.bfloat16).to(DEV)
    wd = w.to(DEV)
    ref = (xb.double() @ _bf16_round(w).double().to(DEV).t())
    y32 = ops.k_linear(xb, wd, None, out_dtype=torch.float32)             # bf16 rows in, fp32 out
    assert y32.dtype == torch.float32
    assert_close(y32, ref, 2e-6, "bf16 in / fp32 out")
    y16 = ops.k_linear(x.to(DEV), wd, None, out_dtype=torch.bfloat16)     # fp32 rows in, bf16 out
    assert y16.dtype == torch.bfloat16
    ref2 = _bf16_round(x).double().to(DEV) @ _bf16_round(w).double().to(DEV).t()
    assert_close(y16.double(), ref2, 4e-3, "fp32 in / bf16 out (one rounding to bf16)")
    assert torch.equal(y16, ops.k_linear(x.to(DEV), wd, None, out_dtype=torch.float32).to(torch.bfloat16)), "the bf16 result is the fp32 result rounded once"
    dx = ops.k_linear_dx(cb, wd, out_dtype=torch.float32)                 # bf16 gradient rows in
    assert_close(dx, cb.double() @ _bf16_round(w).double().to(DEV), 2e-6, "dx from bf16 rows")
    dw, _ = ops.k_linear_dw(cb, xb, False)                                # both operands bf16 rows, exact fp32 products
    assert_close(dw, cb.double().t() @ xb.double(), 1e-5, "dw from bf16 rows")
    dw2, _ = ops.k_linear_dw(cot.to(DEV), xb, False)                      # mixed: fp32 gradient rows, bf16 activations
    assert_close(dw2, cot.double().to(DEV).t() @ xb.double(), 1e-5, "dw from fp32 x bf16 rows")


def _e4m3_bytes(t):
    """e4m3 bytes of an fp32 tensor (saturating), as the kernels write them"""
    return t.clamp(-448, 448).to(torch.float8_e4m3fn).view(torch.uint8)


@pytest.mark.parametrize("M,K,N", [(64, 1024, 512), (1024, 256, 1216), (64, 4096, 1024), (256, 1024, 4672), (1024, 512, 2368), (16, 2048, 512)])
def test_skgemm_narrow_weight_operand(M, K, N):
    """the weight operand of the short GEMMs read from its narrow SHADOW (adnm_skgemm b_dtype) — bf16 in the bf16 mode,
    per-tensor scaled e4m3 in the fp8 modes — gives BIT FOR BIT what the fp32 operand gives after the same rounding, forward (NT) and
    input gradient (NN), on both kernels (register-streaming and LDS-tiled, split or not)."""
    x, w, cot = T(f"nw.x{M}{K}", (M, K)).to(DEV), T(f"nw.w{N}{K}", (N, K), 0.05).to(DEV), T(f"nw.c{M}{N}", (M, N)).to(DEV)
    try:
        ops.set_mfma_precision("bf16")
        w16 = w.to(torch.bfloat16)
        y0, dx0 = ops.k_linear(x, w, None), ops.k_linear_dx(cot, w)
        y1 = ops.k_linear(x, w, None, narrow=(w16.data_ptr(), 1, None))
        dx1 = ops.k_linear_dx(cot, w, narrow=(w16.data_ptr(), 1, None))
        assert torch.equal(y0, y1) and torch.equal(dx0, dx1), "bf16 shadow"
        # fp8: explicit scales for both call sites; the shadow is what the optimiser pass would write: e4m3(w * s)
        ops.QUANT.reset()
        ops.set_mfma_precision("fp8")
        rows, ops.QUANT.max_rows = ops.QUANT.max_rows, 1 << 30
        sx, sw, sc = 448.0 / float(x.abs().max()) / 2, 448.0 / float(w.abs().max()) / 2, 57344.0 / float(cot.abs().max()) / 2
        ops.QUANT.set(x.device, w.data_ptr(), "linear_fwd", sx, sw)
        ops.QUANT.set(x.device, w.data_ptr(), "linear_dgrad", sc, sw)
        w8 = _e4m3_bytes(w * sw)
        s_t = torch.tensor([sw], dtype=torch.float32, device=DEV)
        y0, dx0 = ops.k_linear(x, w, None), ops.k_linear_dx(cot, w)
        y1 = ops.k_linear(x, w, None, narrow=(w8.data_ptr(), 2, s_t))
        dx1 = ops.k_linear_dx(cot, w, narrow=(w8.data_ptr(), 2, s_t))
        ops.QUANT.max_rows = rows
        assert torch.equal(y0, y1) and torch.equal(dx0, dx1), "fp8 shadow"
    finally:
        ops.set_mfma_precision("f32")
        ops.QUANT.reset()


def test_adamw_writes_the_shadows():
    """adnm_adamw_step with a shadow: the update itself is unchanged (bitwise the shadow-less pass), the bf16 shadow is bf16(p) of the
    UPDATED values, the fp8 shadow e4m3(p * scale_b of the tensor's record) with max |p| collected per record while its flag is set."""
    n_t = [4096, 12, 70000, 4, 1024 * 33]
    offs, total = [], 0
    for k in n_t:
        offs.append(total)
        total += (k + 3) // 4 * 4
    g = T("aw.g", (total,), 0.01).to(DEV)
    p0 = T("aw.p", (total,), 0.3).to(DEV)

    def run(mode):
        p, m, v = p0.clone(), torch.zeros(total, device=DEV), torch.zeros(total, device=DEV)
        state = torch.zeros(4, device=DEV)
        ws = torch.empty(int(lib.query("adnm_adamw_ws_bytes")), dtype=torch.uint8, device=DEV)
        sh = torch.zeros(total, dtype=torch.bfloat16 if mode == 1 else torch.uint8, device=DEV) if mode else None
        tab = torch.zeros((8, 8), device=DEV)
        tab[:, 1], tab[:, 5], tab[:, 6] = torch.tensor([3.0, 1.0, 50.0, 1.0, 200.0, 1, 1, 1]), 448.0, 1.0
        seg_end = torch.tensor([o // 4 for o in offs[1:]] + [total // 4], dtype=torch.int32, device=DEV)
        seg_rec = torch.tensor([0, -1, 2, -1, 4], dtype=torch.int32, device=DEV)
        for _ in range(2):
            lib.call("adnm_adamw_step", p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), total, state.data_ptr(), 1e-3, 0.9, 0.999, 1e-9, 1e-2, 0.05,
                     ws.data_ptr(), ws.numel(), None if sh is None else sh.data_ptr(), mode, seg_end.data_ptr(), seg_rec.data_ptr(), 5,
                     tab.data_ptr() if mode == 2 else None, None, torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        return p, m, v, sh, tab
    base = run(0)
    b16 = run(1)
    f8 = run(2)
    for k in range(3):
        assert torch.equal(base[k], b16[k]) and torch.equal(base[k], f8[k]), "the shadow must not change the update"
    assert torch.equal(b16[3], base[0].to(torch.bfloat16))
    p, sh, tab = f8[0], f8[3], f8[4]
    for k, (o, nk) in enumerate(zip(offs, n_t)):
        if k in (0, 2, 4):
            s = float(tab[k, 1])
            assert torch.equal(sh[o:o + nk], _e4m3_bytes(p[o:o + nk] * s)), f"fp8 shadow of tensor {k}"
            # the maximum covers the tensor and whatever alignment padding follows it (zeros here: p0 is dense, so compare with the segment)
            end = offs[k + 1] if k + 1 < len(offs) else total
            assert float(tab[k, 3]) == float(p[o:end].abs().max()), f"max |p| of tensor {k}"
        else:
            assert float(tab[1, 3]) == 0.0 and float(tab[3, 3]) == 0.0


@pytest.mark.parametrize("B,H,W,C,K,act,bias", [(2, 16, 16, 32, 3, lib.ACT_SILU, False), (1, 131, 130, 24, 3, lib.ACT_NONE, True), (2, 64, 64, 192, 3, lib.ACT_SILU, False),
                                                (1, 9, 11, 16, 5, lib.ACT_GELU, True), (4, 128, 128, 128, 3, lib.ACT_NONE, True)])
def test_dwconv_bf16_tokens(B, H, W, C, K, act, bias):
    """bf16 token storage (the wide internals of the full-resolution level): a lane moves 8 channels = 16 bytes per pixel; arithmetic in
    fp32 on the bf16 values, the results rounded once when stored; the tap / bias gradients stay fp32."""
    x, w = T("dwb.x", (B, H * W, C)), T("dwb.w", (C, 1, K, K), 0.5)
    b = T("dwb.b", (C,), 0.3) if bias else None
    cot = T("dwb.c", (B, H * W, C))
    xr, cr = _bf16_round(x), _bf16_round(cot)
    xo, wo = leaf(xr.double()), leaf(w.double())
    bo = leaf(b.double()) if bias else None
    pre = F.conv2d(O.img(xo, H, W), wo, bo, padding=K // 2, groups=C)
    yo = O.seq({lib.ACT_NONE: lambda t: t, lib.ACT_SILU: O.silu, lib.ACT_GELU: O.gelu}[act](pre))
    (yo * cr.double()).sum().backward()
    xg, wg = leaf(xr.to(torch.bfloat16), DEV), leaf(w, DEV)
    bg = leaf(b, DEV) if bias else None
    yg = ops.dwconv(xg, wg, bg, H, W, act)
    assert yg.dtype == torch.bfloat16
    (yg.float() * cr.to(DEV)).sum().backward()
    assert_close(yg.float(), yo, 4e-3, "y (one rounding to bf16)")
    assert_close(xg.grad.float(), xo.grad, 8e-3, "dx (bf16 pre-activation gradient, one more rounding)")
    assert_close(wg.grad, wo.grad, 8e-3, "dw")
    if bias:
        assert_close(bg.grad, bo.grad, 8e-3, "db")
