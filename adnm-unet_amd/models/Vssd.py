"""Plain VSSD mixer — drop-in for the reference's models/Vssd.py (`Mamba2`, `StandardAttention`).
Not instantiated by create_ADNMUNet (the import is commented out at ADNMUNet.py:24 of the reference)
but part of the surface named by the north star; it shares the K1 / K4 / row-norm HIP kernels.

Reference algorithm (Vssd.py:211-283): in_proj -> z | xBC | dt; softplus(dt+dt_bias); depthwise 3x3(+bias)
+ SiLU on all of xBC; grouped non-causal linear attention where head h uses K/Q group h % ngroups
(Vssd.py:197-199); LayerNorm(y) * z; out_proj.
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F

from adnm_hip import ops, lib
from models.ADNssd import StandardAttention  # identical class in the reference (Vssd.py:24-45)


class Mamba2(nn.Module):
    def __init__(self, d_model, d_conv=3, conv_init=None, expand=2, headdim=8, ngroups=2, A_init_range=(1, 16),
                 dt_min=0.001, dt_max=0.1, dt_init_floor=1e-4, dt_limit=(0.0, float("inf")), learnable_init_states=False,
                 activation="silu", bias=False, conv_bias=True, chunk_size=256, use_mem_eff_path=False, layer_idx=None,
                 device=None, dtype=None, linear_attn_duality=True, d_state=16, bimamba=True, **kwargs):
        fk = {"device": device, "dtype": dtype}
        super().__init__()
        self.d_model, self.d_conv, self.conv_init, self.expand = d_model, d_conv, conv_init, expand
        self.d_inner = int(expand * d_model)
        self.headdim, self.d_state = headdim, d_state
        if ngroups == -1:
            ngroups = self.d_inner // headdim
        self.ngroups = ngroups
        assert self.d_inner % headdim == 0
        self.nheads = self.d_inner // headdim
        self.dt_limit = dt_limit
        self.learnable_init_states = learnable_init_states
        self.activation = activation
        self.chunk_size = chunk_size
        self.use_mem_eff_path = use_mem_eff_path
        self.layer_idx = layer_idx
        self.ssd_positve_dA = kwargs.get('ssd_positve_dA', True)
        d_in_proj = 2 * self.d_inner + 2 * ngroups * d_state + self.nheads
        self.in_proj = nn.Linear(d_model, int(d_in_proj), bias=bias, **fk)
        conv_dim = self.d_inner + 2 * ngroups * d_state
        self.bimamba = bimamba
        self.conv2d = nn.Conv2d(conv_dim, conv_dim, groups=conv_dim, bias=conv_bias, kernel_size=d_conv,
                                padding=(d_conv - 1) // 2, **fk)
        if learnable_init_states:
            self.init_states = nn.Parameter(torch.zeros(self.nheads, headdim, d_state, **fk))
            self.init_states._no_weight_decay = True
        self.act = nn.SiLU()
        dt = torch.exp(torch.rand(self.nheads, **fk) * (math.log(dt_max) - math.log(dt_min)) + math.log(dt_min))
        dt = torch.clamp(dt, min=dt_init_floor)
        self.dt_bias = nn.Parameter(dt + torch.log(-torch.expm1(-dt)))
        self.dt_bias._no_weight_decay = True
        assert A_init_range[0] > 0 and A_init_range[1] >= A_init_range[0]
        A = torch.empty(self.nheads, dtype=torch.float32, device=device).uniform_(*A_init_range)
        self.A_log = nn.Parameter(torch.log(A).to(dtype=dtype))
        self.A_log._no_weight_decay = True
        self.D = nn.Parameter(torch.ones(self.nheads, device=device))
        self.D._no_weight_decay = True
        self.norm = nn.LayerNorm(self.d_inner)
        self.out_proj = nn.Linear(self.d_inner, d_model, bias=bias, **fk)
        self.linear_attn_duality = linear_attn_duality
        self.kwargs = kwargs

    def forward(self, u, H, W, seq_idx=None):
        if self.d_conv != 3:
            raise NotImplementedError("only d_conv=3")
        b, l, _ = u.shape
        di, gn, nh = self.d_inner, self.ngroups * self.d_state, self.nheads
        proj = ops.linear(u, self.in_proj.weight, self.in_proj.bias)
        z, xBC, dt = proj[..., :di], proj[..., di:2 * di + 2 * gn], proj[..., 2 * di + 2 * gn:]
        xBC = ops.dwconv(xBC, self.conv2d.weight, self.conv2d.bias, H, W, lib.ACT_SILU)       # Vssd.py:232-234
        x = xBC[..., :di].reshape(b, l, nh, self.headdim)
        if self.linear_attn_duality:
            y = ops.ssd_reduce(x, xBC[..., di:di + gn], xBC[..., di + gn:], dt, self.dt_bias, self.A_log, self.D, self.ngroups)
        else:
            # chunked bidirectional scan (K1b; PARITY UNPINNED — un-vendored mamba_ssm in the reference, Vssd.py:245-275):
            # first half of the heads / groups forward in time, second half on the reversed sequence (:248-261)
            if not self.bimamba or self.ngroups % 2 or nh % 2:
                raise NotImplementedError("the scan branch is built for bimamba=True with an even number of groups and heads")
            hh, g2, ds = nh // 2, self.ngroups // 2, self.d_state
            Bm, Cm = xBC[..., di:di + gn], xBC[..., di + gn:]
            halves = []
            for e in (0, 1):
                sl = slice(e * hh, (e + 1) * hh)
                gs = slice(e * g2 * ds, (e + 1) * g2 * ds)
                halves.append(ops.ssd_scan(x[:, :, sl], Bm[..., gs], Cm[..., gs], dt[..., sl], self.dt_bias[sl], self.A_log[sl], self.D[sl],
                                           g2, int(self.chunk_size), e == 1))
            y = torch.cat(halves, dim=2)
        y = ops.emul(ops.rownorm(y.reshape(b, l, di), self.norm.weight, self.norm.bias, None, None, self.norm.eps, True), z)  # :280-281
        return ops.linear(y, self.out_proj.weight, self.out_proj.bias)
