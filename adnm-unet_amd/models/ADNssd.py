"""ADN-SSD mixer — drop-in for the reference's models/ADNssd.py (`Mamba2`, `StandardAttention`):
same constructor arguments, forward(u, H, W) signature and state_dict keys, executed by HIP kernels.

Reference algorithm (ADNssd.py:302-462): in_proj -> split z | xBC | dt; softplus(dt + dt_bias);
even xBC channels -> depthwise 3x3 + SiLU, odd channels split again by parity into four chains of
asymmetric 1x3 / 3x1 depthwise convs + SiLU; z -> depthwise 3x3 + SiLU; non-causal linear-attention
reduction (K1) on the even and on the odd half; channel interleave; LayerNorm; cat(alpha1*y, alpha1*z);
out_proj.

MI355X formulation (adnm_hip.ops.ADNMixerFn): the even/odd gathers are folded into a row permutation of
in_proj.weight, the 3x1 o 1x3 chains into one effective 3x3 tap set, the two half-reductions into ONE
K1 launch with the halves as two K/Q groups, the scalar alpha1 into out_proj — all as tiny differentiable
ops on PARAMETERS, never on activations.
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F

from adnm_hip import ops


class StandardAttention(nn.Module):
    """Soft-max attention with head_dim 4 (ADNssd.py:26-47 of the reference): to_qkv / to_out on the short-GEMM MFMA kernel,
    the attention itself fused per head in LDS (csrc/attn4.hip)."""

    def __init__(self, dim, heads=8, dim_head=64, dropout=0., **kwargs):
        super().__init__()
        inner_dim = dim_head * heads
        self.heads = heads
        self.scale = dim_head ** -0.5
        self.to_qkv = nn.Linear(dim, inner_dim * 3, bias=False)
        self.to_out = nn.Linear(inner_dim, dim)
        self.dropout = nn.Dropout(dropout)
        self.inner_dim = inner_dim

    def forward(self, x, H, W):
        b, n, _ = x.shape
        if self.dropout.p != 0.0 and self.training:
            raise RuntimeError("StandardAttention: the HIP attention kernel has no dropout (the reference configuration uses dropout=0)")
        qkv = ops.linear(x, self.to_qkv.weight, None)
        # 4-wide heads: fused, no (L, L) score tensor; any other head width raises (one code path, no PyTorch fallback)
        return ops.linear(ops.attn4(qkv, self.heads, self.scale), self.to_out.weight, self.to_out.bias)


def _dw(channels, k, pad, bias):
    return nn.Conv2d(channels, channels, kernel_size=k, padding=pad, groups=channels, bias=bias)


class Mamba2(nn.Module):
    def __init__(self, d_model, d_conv=3, conv_init=None, expand=2, headdim=8, ngroups=2, A_init_range=(1, 16),
                 dt_min=0.001, dt_max=0.1, dt_init_floor=1e-4, dt_limit=(0.0, float("inf")), learnable_init_states=False,
                 bias=False, conv_bias=False, chunk_size=256, use_mem_eff_path=False, layer_idx=None, device=None,
                 dtype=None, linear_attn_duality=True, d_state=16, bimamba=True, **kwargs):
        fk = {"device": device, "dtype": dtype}
        super().__init__()
        self.bimamba = bimamba
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
        self.chunk_size = chunk_size
        self.use_mem_eff_path = use_mem_eff_path
        self.layer_idx = layer_idx
        self.ssd_positve_dA = kwargs.get('ssd_positve_dA', True)
        gn = ngroups * d_state
        d_in_proj = 2 * self.d_inner + 2 * gn + self.nheads
        self.in_proj = nn.Linear(d_model, int(d_in_proj), bias=bias, **fk)
        conv_dim = self.d_inner + 2 * gn
        q, qbc = self.d_inner // 4, 2 * gn // 4
        self.conv_13_x1 = _dw(q, (1, 3), (0, 1), conv_bias)
        self.conv_31_x1 = _dw(q, (3, 1), (1, 0), conv_bias)
        self.conv_13_x2 = _dw(q, (1, 3), (0, 1), conv_bias)
        self.conv_31_x2 = _dw(q, (3, 1), (1, 0), conv_bias)
        self.conv_13_bc1 = _dw(qbc, (1, 3), (0, 1), conv_bias)
        self.conv_31_bc1 = _dw(qbc, (3, 1), (1, 0), conv_bias)
        self.conv_13_bc2 = _dw(qbc, (1, 3), (0, 1), conv_bias)
        self.conv_31_bc2 = _dw(qbc, (3, 1), (1, 0), conv_bias)
        self.conv2d = nn.Conv2d(conv_dim // 2, conv_dim // 2, groups=conv_dim // 2, bias=conv_bias, kernel_size=d_conv,
                                padding=(d_conv - 1) // 2, **fk)
        if learnable_init_states:
            self.init_states = nn.Parameter(torch.zeros(self.nheads, headdim, d_state, **fk))
            self.init_states._no_weight_decay = True
        self.act = nn.SiLU()
        self.act2 = nn.SiLU()
        dt = torch.exp(torch.rand(self.nheads, **fk) * (math.log(dt_max) - math.log(dt_min)) + math.log(dt_min))
        dt = torch.clamp(dt, min=dt_init_floor)
        self.dt_bias = nn.Parameter(dt + torch.log(-torch.expm1(-dt)))  # softplus^-1
        self.dt_bias._no_weight_decay = True
        assert A_init_range[0] > 0 and A_init_range[1] >= A_init_range[0]
        A = torch.empty(self.nheads, dtype=torch.float32, device=device).uniform_(*A_init_range)
        self.A_log = nn.Parameter(torch.log(A).to(dtype=dtype))
        self.A_log._no_weight_decay = True
        self.D = nn.Parameter(torch.ones(self.nheads, device=device))
        self.D._no_weight_decay = True
        self.norm = nn.LayerNorm(self.d_inner)
        self.scale = nn.Parameter(torch.tensor(1.))
        self.shift = nn.Parameter(torch.tensor(0.))
        self.linear_attn_duality = linear_attn_duality
        self.act_z = nn.SiLU()
        self.conv2d_z = nn.Conv2d(self.d_inner, self.d_inner, groups=self.d_inner, bias=conv_bias, kernel_size=d_conv,
                                  padding=(d_conv - 1) // 2, **fk)
        self.alpha1 = nn.Parameter(torch.tensor(1, dtype=torch.float))
        self.alpha2 = nn.Parameter(torch.tensor(1, dtype=torch.float))
        self.out_proj = nn.Linear(self.d_inner * 2, d_model, bias=bias, **fk)
        self.kwargs = kwargs
        self._build_index_maps()

    # ------------------------------------------------------------------ static channel bookkeeping
    def _build_index_maps(self):
        di, gn, P = self.d_inner, self.ngroups * self.d_state, self.headdim
        nh = self.nheads
        if di % 8 or gn % 4 or nh % 2:
            raise NotImplementedError("ADN-SSD needs d_inner % 8 == 0, ngroups*d_state % 4 == 0 and an even head count")
        # x': kernel head h = 2j+e is head j of the parity-e half (ADNssd.py:371-372,397-404); its channel p is
        # original x channel 2*(j*P+p)+e.  The interleave of y (:409-411) is the same map.
        perm_x = torch.tensor([2 * ((h // 2) * P + p) + (h % 2) for h in range(nh) for p in range(P)])
        perm_b = torch.tensor([di + 2 * n + e for e in (0, 1) for n in range(gn // 2)])
        perm_c = perm_b + gn
        self.register_buffer("_perm_x", perm_x, persistent=False)
        rows = torch.cat([torch.arange(di), di + torch.cat([perm_x, perm_b, perm_c]), torch.arange(2 * di + 2 * gn, 2 * di + 2 * gn + nh)])
        self.register_buffer("_rows_in", rows, persistent=False)
        self.register_buffer("_perm_xbc", torch.cat([perm_x, perm_b, perm_c]), persistent=False)
        self.register_buffer("_cols_out", torch.cat([perm_x, torch.arange(di, 2 * di)]), persistent=False)

    def _effective_taps(self):
        """(9, di+2gN) tap-major taps of every xBC channel in kernel order (see ADNMixerFn)."""
        outer = lambda c31, c13: c31.weight.reshape(-1, 3, 1) * c13.weight.reshape(-1, 1, 3)
        k_oe = torch.cat([outer(self.conv_31_x1, self.conv_13_x1), outer(self.conv_31_bc1, self.conv_13_bc1)], 0)  # :343,345
        k_oo = torch.cat([outer(self.conv_31_x2, self.conv_13_x2), outer(self.conv_31_bc2, self.conv_13_bc2)], 0)  # :344,346
        k_odd = torch.stack((k_oe, k_oo), dim=1).reshape(-1, 3, 3)            # O'[:, 0::2] / O'[:, 1::2] (:363-364)
        k_all = torch.stack((self.conv2d.weight.reshape(-1, 3, 3), k_odd), dim=1).reshape(-1, 3, 3)  # even / odd xBC channels
        k_all = k_all.index_select(0, self._perm_xbc)
        return k_all.reshape(k_all.shape[0], 9).t().contiguous().float()

    adnm_prep_kind = "adn"   # adnm_hip.ops.prep_group prepares every mixer of a model stage in one launch

    def adnm_prep_ready(self):
        return self.conv2d.bias is None and self.in_proj.bias is None and self.d_conv == 3 and not self.learnable_init_states

    def adnm_prep_args(self):
        return ((self.d_model, self.d_inner, self.ngroups * self.d_state, self.headdim),
                [self.in_proj.weight, self.conv2d.weight, self.conv_31_x1.weight, self.conv_31_bc1.weight, self.conv_31_x2.weight,
                 self.conv_31_bc2.weight, self.conv_13_x1.weight, self.conv_13_bc1.weight, self.conv_13_x2.weight, self.conv_13_bc2.weight,
                 self.conv2d_z.weight, self.norm.weight, self.norm.bias, self.out_proj.weight, self.alpha1])

    def forward(self, u, H, W, seq_idx=None):
        """u: (B, L, d_model) tokens, L = H*W.  Returns the same shape (ADNssd.py:302-462 of the reference)."""
        scan_chunk = 0
        if not self.linear_attn_duality:
            # chunked bidirectional scan (K1b, csrc/ssd_scan.hip).  PARITY UNPINNED: the reference delegates this branch to
            # un-vendored mamba_ssm Triton kernels (ADNssd.py:413-454); create_ADNMUNet never takes it (ADNMUNet.py:277).
            if not self.bimamba:
                raise NotImplementedError("bimamba=False references undefined names in the reference (ADNssd.py:442-454)")
            scan_chunk = int(self.chunk_size)
        if self.conv2d.bias is not None or self.in_proj.bias is not None or self.d_conv != 3:
            raise NotImplementedError("ADN-SSD HIP path covers the reference configuration: conv_bias=False, bias=False, d_conv=3")
        if self.learnable_init_states:
            raise NotImplementedError("learnable_init_states is only meaningful on the chunked-scan branch")
        # the kernel-layout tensors (csrc/paramprep.hip; _effective_taps()/_rows_in state the same map in torch): prepared for the whole model
        # stage in one launch by ops.prep_group (VisionMamba.forward_stage*), or here for this module alone
        pre = self.__dict__.pop("_adnm_prepped", None)
        if pre is None:
            dims, params = self.adnm_prep_args()
            pre = ops.adn_prep(*dims, params)
        w_in, taps, ln_w, ln_b, w_out = pre[:5]
        narrow = tuple(pre[5:9]) if len(pre) == 9 else (None, None, None, None)   # the grouped prep's bf16 / fp8 copies of the two projections
        return ops.adn_mixer(u, w_in, taps, None, self.dt_bias, self.A_log, self.D, ln_w, ln_b, w_out, H, W,
                             self.headdim, self.ngroups * self.d_state // 2, scan_chunk, self.ngroups,
                             qkeys=(self.in_proj.weight.data_ptr(), self.out_proj.weight.data_ptr()), narrow=narrow)
