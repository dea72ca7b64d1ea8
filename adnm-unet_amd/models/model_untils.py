"""Building blocks of the ADNM-UNet conv stack — drop-in for the reference's models/model_untils.py
(same class names, constructor arguments, forward signatures and state_dict keys).

What runs where (DESIGN.md has the table): depthwise convs, the Haar wavelet pyramid (WTConv2d),
InstanceNorm(+scalar affine, +GELU), BiasFree/Layer/RMS norms, the gated-FFN activation, the dense
3x3 / transposed convs and the 1x1 / Linear projections are all hand-written HIP kernels (MFMA for the
GEMM-shaped ones) working directly on the channels-last token layout (B, H*W, C) that the reference
keeps between stages.  There is ONE code path: a configuration no kernel takes raises (no torch / library
fallback).  Modules whose reference forward takes NCHW keep accepting NCHW; parents call the `*_tokens`
methods so no BLD<->BCHW copies (13.5 % of the reference's CPU step) remain on the token path.
"""
import math
import numbers

import torch
import torch.nn as nn
import torch.nn.functional as F

from adnm_hip import ops, lib
from models.WTConv2d import WTConv2d


def to_2tuple(x):
    return tuple(x) if isinstance(x, (tuple, list)) else (x, x)


def to_bchw(x):
    b, l, d = x.shape
    h = w = int(math.isqrt(l))
    return x.reshape(b, h, w, d).permute(0, 3, 1, 2)


def to_bld(x):
    return x.flatten(2).transpose(1, 2)


def _hw(l):
    h = int(math.isqrt(l))
    return h, h


def nchw_view(x, h, w):
    """(B, L, C) tokens -> (B, C, H, W) logical NCHW tensor with channels-last strides (no copy)."""
    b, l, c = x.shape
    return x.reshape(b, h, w, c).permute(0, 3, 1, 2)


def tokens_of(y):
    """(B, C, H, W) -> (B, H*W, C); free when y is channels-last."""
    b, c, h, w = y.shape
    return y.permute(0, 2, 3, 1).reshape(b, h * w, c)


class DropPath(nn.Module):
    def __init__(self, drop_prob=0.0):
        super().__init__()
        self.drop_prob = drop_prob

    def forward(self, x):
        if self.drop_prob == 0.0 or not self.training:
            return x
        keep = 1.0 - self.drop_prob
        mask = x.new_empty((x.shape[0],) + (1,) * (x.dim() - 1)).bernoulli_(keep)
        return x * mask / keep


class BiasFree_LayerNorm(nn.Module):
    """model_untils.py:29-48 of the reference: (x - mu) / sqrt(var + 1e-5) * weight."""

    def __init__(self, normalized_shape):
        super().__init__()
        if isinstance(normalized_shape, numbers.Integral):
            normalized_shape = (normalized_shape,)
        normalized_shape = torch.Size(normalized_shape)
        assert len(normalized_shape) == 1
        self.weight = nn.Parameter(torch.ones(normalized_shape))
        self.normalized_shape = normalized_shape

    def forward(self, x, scale=None, shift=None):
        return ops.rownorm(x, self.weight, None, scale, shift, 1e-5, True)

    def tap(self, x, scale=None, shift=None):
        return ops.rownorm_tap(x, self.weight, None, scale, shift, 1e-5, True)

    def mix_tap(self, xs, scalars, scale=None, shift=None, gamma=None):
        """(scale * norm(x) + shift, x) for x = gamma * (s0*xs[0] + s1*xs[1]): the residual mix that precedes this norm rides in its kernel"""
        return ops.mixnorm(xs, scalars, gamma, self.weight, None, scale, shift, 1e-5, True)


class RMSNorm(nn.Module):
    """Stands in for mamba_ssm.ops.triton.layer_norm.RMSNorm (bound at ADNMUNet.py:278 of the reference):
    x * rsqrt(mean(x^2) + eps) * weight; `scale`/`shift` fuse Block's scalar affine (ADNMUNet.py:149,155)."""

    def __init__(self, hidden_size, eps=1e-5, **kwargs):
        super().__init__()
        self.eps = eps
        self.weight = nn.Parameter(torch.ones(hidden_size))
        self.register_parameter("bias", None)

    def forward(self, x, scale=None, shift=None):
        return ops.rownorm(x, self.weight, None, scale, shift, self.eps, False)

    def tap(self, x, scale=None, shift=None):
        return ops.rownorm_tap(x, self.weight, None, scale, shift, self.eps, False)

    def mix_tap(self, xs, scalars, scale=None, shift=None, gamma=None):
        return ops.mixnorm(xs, scalars, gamma, self.weight, None, scale, shift, self.eps, False)


class Mlp(nn.Module):
    def __init__(self, in_features, out_features=None, hidden_features=None, act_func=nn.GELU, drop=0., bias=True):
        super().__init__()
        out_features = out_features or in_features
        hidden_features = hidden_features or in_features * 2
        self.fc1 = nn.Linear(in_features, hidden_features, bias=bias)
        self.act1 = act_func()
        self.fc2 = nn.Linear(hidden_features, out_features, bias=bias)
        self.drop = nn.Dropout(drop)
        self.act2 = act_func()

    def forward(self, x):
        h = ops.linear(x, self.fc1.weight, self.fc1.bias)
        code = _act_code(self.act1)
        h = self.drop(_apply_act(self.act1, h))
        return self.drop(ops.linear(h, self.fc2.weight, self.fc2.bias))


class Swish(nn.Module):
    def __init__(self, beta_init=1.0):
        super().__init__()
        self.beta = nn.Parameter(torch.tensor(beta_init, dtype=torch.float))

    def forward(self, x):
        return ops.swish(x, self.beta)   # one HIP pass each way (d beta through partials + the fold)


def _apply_act(act, x):
    """an activation that was not fused into the producing kernel: GELU / SiLU / Swish have HIP kernels, anything else raises"""
    code = _act_code(act)
    if code in (lib.ACT_GELU, lib.ACT_SILU):
        return ops.act(x, code)
    if isinstance(act, Swish):
        return act(x)
    raise RuntimeError(f"no HIP kernel for the activation {act} (GELU, SiLU and Swish are implemented; the HIP path has no PyTorch fallback)")


def _act_code(act):
    if act is None:
        return lib.ACT_NONE
    if isinstance(act, nn.GELU):
        return lib.ACT_GELU
    if isinstance(act, nn.SiLU):
        return lib.ACT_SILU
    return None


class Conv2dLayer(nn.Module):
    """conv -> [scale*norm+shift] -> [act] (model_untils.py:73-93 of the reference)."""

    def __init__(self, in_channels, out_channels, kernel_size=(3, 3), stride=(1, 1), padding=(1, 1),
                 dilation=(1, 1), groups=1, bias=True, dropout=0, norm=None, act_func=None):
        super().__init__()
        self.dropout = nn.Dropout2d(dropout) if dropout > 0 else None
        self.conv = nn.Conv2d(in_channels, out_channels, kernel_size, stride, padding, dilation, groups, bias)
        self.norm = norm
        self.act = act_func() if act_func else None
        if norm:
            self.scale = nn.Parameter(torch.tensor(1.))
            self.shift = nn.Parameter(torch.tensor(0.))

    def _is_depthwise_same(self):
        c = self.conv
        k = c.kernel_size
        return (c.groups == c.in_channels == c.out_channels and k[0] == k[1] and k[0] in (3, 5) and c.stride == (1, 1)
                and c.dilation == (1, 1) and c.padding in ((k[0] // 2, k[0] // 2), 'same') and c.in_channels % 4 == 0)

    def adnm_nhwc_parameters(self):
        """adnm_hip.trainer.FlatTrainer keeps these in (out, kh, kw, in) memory order: the NHWC conv kernel's reduction axis contiguous."""
        return [self.conv.weight] if self._is_dense3() else []

    def _is_dense3(self):
        c = self.conv
        return (c.kernel_size == (3, 3) and c.groups == 1 and c.stride == (1, 1) and c.dilation == (1, 1) and c.padding in ((1, 1), 'same')
                and c.padding_mode == 'zeros')

    def _is_pointwise(self):
        c = self.conv
        return c.kernel_size == (1, 1) and c.groups == 1 and c.stride == (1, 1) and c.padding in ((0, 0), 'same', 'valid')

    def forward_tokens(self, x, h, w):
        """(B, H*W, Cin) -> (B, H*W, Cout) (stride-1 'same' convs only)."""
        if self.dropout and self.training:
            raise RuntimeError("Conv2dLayer: Dropout2d has no HIP kernel (the reference configuration uses dropout=0; no PyTorch fallback)")
        code = _act_code(self.act)
        fused_act = False
        if self._is_depthwise_same() and not self.norm and code is not None:
            x = ops.dwconv(x, self.conv.weight, self.conv.bias, h, w, code)  # HIP stencil, act fused
            fused_act = True
        elif self._is_depthwise_same():
            x = ops.dwconv(x, self.conv.weight, self.conv.bias, h, w, lib.ACT_NONE)
        elif self._is_pointwise():
            x = ops.linear(x, self.conv.weight.reshape(self.conv.out_channels, -1), self.conv.bias)
        elif self._is_dense3():   # implicit-GEMM MFMA conv, bias + GELU in its epilogue
            fuse = not self.norm and code in (lib.ACT_NONE, lib.ACT_GELU)
            x = ops.conv3(x, self.conv.weight, self.conv.bias, h, w, code if fuse else lib.ACT_NONE)
            fused_act = fuse
        else:
            raise RuntimeError(f"Conv2dLayer: no HIP kernel for {self.conv} on tokens (depthwise 3x3/5x5, 1x1 and dense 3x3 'same' convs are implemented)")
        if self.norm:
            x, fused_act = self._norm_tokens(x, h, w, code)
        if self.act and not fused_act:
            x = _apply_act(self.act, x)
        return x

    def _norm_tokens(self, x, h, w, code):
        n = self.norm
        if isinstance(n, nn.InstanceNorm2d) and not n.affine and not n.track_running_stats and x.shape[-1] % 4 == 0:
            fuse = code in (lib.ACT_NONE, lib.ACT_GELU)
            return ops.instnorm(x, self.scale, self.shift, n.eps, code if fuse else lib.ACT_NONE), fuse
        raise RuntimeError(f"Conv2dLayer: no HIP kernel for the norm {n} on {x.shape[-1]} channels (affine-less InstanceNorm2d on a multiple "
                           "of 4 channels is implemented; the HIP path has no PyTorch fallback)")

    def _keeps_shape(self):
        c = self.conv
        if c.stride != (1, 1):
            return False
        if isinstance(c.padding, str):
            return c.padding == 'same'
        return all(2 * p == d * (k - 1) for p, d, k in zip(c.padding, c.dilation, c.kernel_size))

    def forward(self, x):
        if not self._keeps_shape():
            raise RuntimeError(f"Conv2dLayer: no HIP kernel for the shape-changing conv {self.conv} (stride-1 'same' convs are implemented; the HIP "
                               "path has no PyTorch fallback)")
        b, c, h, w = x.shape
        return nchw_view(self.forward_tokens(tokens_of(x), h, w), h, w)


class WTConvLayer(nn.Module):
    """WTConv2d -> [scale*norm+shift] -> [act] (model_untils.py:96-116 of the reference)."""

    def __init__(self, in_channels, out_channels, kernel_size=3, stride=1, wt_levels=2, bias=True,
                 dropout=0, norm=None, act_func=None):
        super().__init__()
        self.dropout = nn.Dropout2d(dropout) if dropout > 0 else None
        self.conv = WTConv2d(in_channels, out_channels, kernel_size, stride, bias, wt_levels=wt_levels)
        self.norm = norm
        self.act = act_func() if act_func else None
        if norm:
            self.scale = nn.Parameter(torch.tensor(1.))
            self.shift = nn.Parameter(torch.tensor(0.))

    def forward_tokens(self, x, h, w, tap=False):
        """tap=True: -> (y, alias of x): hand the alias to x's other consumer (the residual mix); its gradient is then added inside the
        wavelet conv's last backward kernel instead of by a separate autograd add (ops.WTConvFn)."""
        if self.dropout:
            if self.training:
                raise RuntimeError("WTConvLayer: Dropout2d has no HIP kernel (the reference configuration uses dropout=0; no PyTorch fallback)")
        alias = None
        if tap:
            x, alias = self.conv.forward_tokens(x, h, w, tap=True)
        else:
            x = self.conv.forward_tokens(x, h, w)
        code = _act_code(self.act)
        fused = False
        if self.norm:
            n = self.norm
            if isinstance(n, nn.InstanceNorm2d) and not n.affine and x.shape[-1] % 4 == 0 and code in (lib.ACT_NONE, lib.ACT_GELU):
                x = ops.instnorm(x, self.scale, self.shift, n.eps, code)  # IN + scalar affine + GELU in one pass
                fused = True
            else:
                raise RuntimeError(f"WTConvLayer: no HIP kernel for the norm {n} with activation {self.act} on {x.shape[-1]} channels "
                                   "(affine-less InstanceNorm2d [+ GELU] on a multiple of 4 channels is implemented; no PyTorch fallback)")
        if self.act and not fused:
            x = _apply_act(self.act, x)
        return (x, alias) if tap else x

    def forward(self, x):
        b, c, h, w = x.shape
        y = nchw_view(self.forward_tokens(tokens_of(x), h, w), h, w)
        if self.conv.stride > 1:
            y = y[:, :, ::self.conv.stride, ::self.conv.stride]
        return y


class DeConv2dLayer(nn.Module):
    def __init__(self, in_channels, out_channels, ratio=4, kernel_size=(3, 3), groups=1,
                 bias=True, dropout=0, norm=None, act_func=None):
        super().__init__()
        padding_w = max(0, (kernel_size[1] - ratio + 1) // 2)
        output_padding_w = ratio - kernel_size[1] + 2 * padding_w
        if not (0 <= output_padding_w < ratio):
            raise ValueError(f"output_padding {output_padding_w} must satisfy 0 <= output_padding < {ratio}")
        self.dropout = nn.Dropout3d(dropout, inplace=False) if dropout > 0 else None
        self.trans_conv = nn.ConvTranspose2d(in_channels, out_channels, kernel_size, stride=(ratio, ratio),
                                             padding=(padding_w, padding_w),
                                             output_padding=(output_padding_w, output_padding_w), groups=groups, bias=bias)
        self.norm = norm if norm else None
        self.act = act_func() if act_func else None
        if norm:
            self.scale = nn.Parameter(torch.tensor(1.))
            self.shift = nn.Parameter(torch.tensor(0.))

    def adnm_nhwc_parameters(self):
        return [self.trans_conv.weight]

    def forward(self, x):
        """x: (B, Cin, H, W) as the reference (model_untils.py:150-158) — runs the token kernel behind UpSample's configuration."""
        c = self.trans_conv
        if (c.stride != (2, 2) or c.kernel_size != (3, 3) or c.padding != (1, 1) or c.output_padding != (1, 1) or c.groups != 1 or self.norm
                or self.act or self.dropout is not None):
            raise RuntimeError(f"DeConv2dLayer: the HIP transposed conv is k=3, s=2, p=1, output_padding=1 without norm/act, got {c} "
                               "(the HIP path has no PyTorch fallback)")
        b, ci, h, w = x.shape
        return nchw_view(ops.convt2x(tokens_of(x), c.weight, c.bias, h, w), 2 * h, 2 * w)


class FeedForward(nn.Module):
    """Gated conv FFN (model_untils.py:172-197 of the reference): 1x1 d->4d, depthwise 3x3,
    gelu(x1)*sigmoid(x2), 1x1 2d->d.  Token path: GEMM, HIP stencil, HIP gate, GEMM."""

    def __init__(self, dim, ffn_expansion_factor=2, bias=True):
        super().__init__()
        hidden = int(dim * ffn_expansion_factor)
        self.project_in = Conv2dLayer(dim, hidden * 2, kernel_size=(1, 1), stride=(1, 1), padding=(0, 0), bias=bias)
        self.dwconv = Conv2dLayer(hidden * 2, hidden * 2, kernel_size=(3, 3), stride=(1, 1), padding=(1, 1),
                                  groups=hidden * 2, bias=bias)
        self.project_out = Conv2dLayer(hidden, dim, kernel_size=(1, 1), stride=(1, 1), padding=(0, 0), bias=bias)

    def forward_tokens(self, x, h, w):
        """the whole chain as one autograd node (ops.FeedForwardFn): GEMM, HIP stencil, HIP gate, GEMM"""
        pi, dc, po = self.project_in.conv, self.dwconv.conv, self.project_out.conv
        return ops.feedforward(x, pi.weight.reshape(pi.out_channels, -1), pi.bias, dc.weight, dc.bias, po.weight.reshape(po.out_channels, -1), po.bias,
                               h, w)

    def forward(self, x):
        b, c, h, w = x.shape
        return nchw_view(self.forward_tokens(tokens_of(x), h, w), h, w)


class ConvFFD(nn.Module):
    def __init__(self, dim, bias=True):
        super().__init__()
        self.in_proj = nn.Linear(dim, dim * 2, bias=bias)
        self.dw_conv = Conv2dLayer(dim * 2, dim * 2, kernel_size=3, stride=1, padding=1, bias=bias, groups=dim * 2)
        self.out_proj = nn.Linear(dim * 2, dim, bias=bias)
        self.act = nn.GELU()

    def forward(self, x):
        b, l, d = x.shape
        h, w = _hw(l)
        x = ops.linear(x, self.in_proj.weight, self.in_proj.bias)
        c = self.dw_conv.conv
        x = ops.dwconv(x, c.weight, c.bias, h, w, lib.ACT_GELU)  # conv + GELU (model_untils.py:219-220)
        return ops.linear(x, self.out_proj.weight, self.out_proj.bias)


class PatchEmbed(nn.Module):
    def __init__(self, img_size=256, patch_size=2, in_channels=3, embed_dim=256, kernel=6, num_frames=5, target_frames=3,
                 wt_levels=2, ls_init_value=1, act=nn.GELU, InstanceNorm=True):
        super().__init__()
        img_size = to_2tuple(img_size)
        patch_size = to_2tuple(patch_size)
        self.num_patches = (img_size[1] // patch_size[1]) * (img_size[0] // patch_size[0])
        self.patches_resolution = [img_size[0] // patch_size[0], img_size[1] // patch_size[1]]
        self.img_size, self.patch_size = img_size, patch_size
        self.embed_dim, self.num_frames, self.target_frames = embed_dim, num_frames, target_frames
        self.gamma = nn.Parameter(ls_init_value * torch.ones(embed_dim)) if ls_init_value is not None else None
        self.conv1 = nn.Sequential(WTConvLayer(in_channels, in_channels, kernel_size=kernel, stride=1, bias=False,
                                               wt_levels=wt_levels, act_func=nn.GELU))
        self.conv2 = nn.Sequential(Conv2dLayer(in_channels, embed_dim, kernel_size=(3, 3), stride=(1, 1), padding=(1, 1),
                                               groups=1, bias=False, act_func=nn.GELU))
        self.conv3 = nn.Sequential(WTConvLayer(embed_dim, embed_dim, kernel_size=kernel, stride=1, bias=False, wt_levels=wt_levels,
                                               norm=nn.InstanceNorm2d(embed_dim) if InstanceNorm else nn.GroupNorm(4, embed_dim)))
        self.alpha1 = nn.Parameter(torch.tensor(1, dtype=torch.float))
        self.beta1 = nn.Parameter(torch.tensor(1, dtype=torch.float))
        self.alpha2 = nn.Parameter(torch.tensor(1, dtype=torch.float))
        self.beta2 = nn.Parameter(torch.tensor(1, dtype=torch.float))

    def forward(self, x):
        """x: (B, L, T_in) tokens -> ((B, L, embed_dim), last input frame (B, H, W)) (model_untils.py:299-314)."""
        b, l, d = x.shape
        h, w = _hw(l)
        res = x[..., -1].reshape(b, h, w)
        x = ops.lincomb([self.conv1[0].forward_tokens(x, h, w).contiguous(), x.contiguous()], [self.alpha1, self.beta1])
        shortcut = self.conv2[0].forward_tokens(x, h, w)
        y, shortcut = self.conv3[0].forward_tokens(shortcut, h, w, tap=True)
        x = ops.lincomb([y, shortcut], [self.alpha2, self.beta2], self.gamma)
        return x, res


class WTLayer(nn.Module):
    def __init__(self, this_dim=128, next_dim=256, kernel=5, bias=True, wt_levels=2, ls_init_value=1, act=nn.GELU,
                 if_res=False, InstanceNorm=True):
        super().__init__()
        self.next_dim = next_dim
        norm_group = 8 if if_res else 4
        self.wtconv = WTConvLayer(this_dim, this_dim, kernel_size=kernel, stride=1, bias=bias, wt_levels=wt_levels,
                                  norm=nn.InstanceNorm2d(this_dim) if InstanceNorm else nn.GroupNorm(norm_group, this_dim))
        self.conv = Conv2dLayer(this_dim, next_dim, kernel_size=3, padding=1, stride=1, bias=True, act_func=nn.GELU)
        self.mlp = Mlp(this_dim)
        self.gamma = nn.Parameter(ls_init_value * torch.ones(this_dim)) if ls_init_value is not None else None
        self.alpha = nn.Parameter(torch.tensor(1, dtype=torch.float))
        self.beta = nn.Parameter(torch.tensor(1, dtype=torch.float))
        self.gama1 = nn.Parameter(torch.tensor(1, dtype=torch.float))
        self.gama2 = nn.Parameter(torch.tensor(1, dtype=torch.float))
        self.gama3 = nn.Parameter(torch.tensor(1, dtype=torch.float))
        self.gama4 = nn.Parameter(torch.tensor(1, dtype=torch.float))

    def forward(self, x, residual=None, features=None):
        """(model_untils.py:402-426 of the reference).  With a residual the reference builds the feature
        concat and discards it (:408), so `features` does not enter the result on that branch."""
        if residual is not None:
            x = ops.catmix(x, residual, None, self.gama1, self.gama2)
        elif features is not None:
            x = ops.lincomb([x, features], [None, self.gama3])
        b, l, d = x.shape
        h, w = _hw(l)
        y, x = self.wtconv.forward_tokens(x, h, w, tap=True)
        x = ops.lincomb([y, x], [self.alpha, self.beta])
        x = self.mlp(x)
        if self.gamma is not None:
            x = ops.lincomb([x], [None], self.gamma)
        return self.conv.forward_tokens(x, h, w)


class DownSample(nn.Module):
    def __init__(self, dim=256, kernel=3, ratio=2):
        super().__init__()
        self.ratio = ratio
        self.dim = dim
        self.max_pool = nn.MaxPool2d(kernel_size=ratio, stride=ratio, padding=0)

    def forward(self, x, tap=False):
        b, l, d = x.shape
        h, w = _hw(l)
        if d % 4 or not 2 <= self.ratio <= 4:
            raise RuntimeError(f"DownSample: the HIP max-pool takes 4 | channels and ratio 2..4, got dim {d}, ratio {self.ratio}")
        return ops.maxpool(x, h, w, self.ratio, self.ratio, self.ratio, tap=tap)

    def tap(self, x):
        """-> (pooled, alias of x): the alias goes to x's other consumer (the skip connection); its gradient is then summed with the
        pool's inside the pool's backward kernel"""
        return self.forward(x, tap=True)


class UpSample(nn.Module):
    def __init__(self, dim=128, kernel=3, ratio=2, bias=True):
        super().__init__()
        self.ratio = ratio
        self.trans_conv = DeConv2dLayer(dim, dim, ratio=ratio, kernel_size=(kernel, kernel), bias=bias, act_func=None)

    def forward(self, x):
        b, l, d = x.shape
        h, w = _hw(l)
        tc = self.trans_conv
        c = tc.trans_conv
        if (self.ratio != 2 or c.kernel_size != (3, 3) or c.padding != (1, 1) or c.output_padding != (1, 1) or c.groups != 1 or tc.norm or tc.act
                or tc.dropout is not None):
            raise RuntimeError(f"UpSample: the HIP transposed conv is k=3, s=2, p=1, output_padding=1 without norm/act, got {c}")
        return ops.convt2x(x, c.weight, c.bias, h, w)


class IntensityGate(nn.Module):
    def __init__(self, threshold=0.):
        super().__init__()
        self.threshold = nn.Parameter(torch.tensor(threshold, dtype=torch.float))
        self.enhance = nn.Parameter(torch.tensor(1, dtype=torch.float))
        self.act = nn.SiLU()

    def forward(self, x):
        return ops.igate(x, self.enhance, self.threshold)  # fused silu(enhance*(x-threshold)) + scalar gradients


class Channel_Att_Bridge(nn.Module):
    """Channel-attention bridge over the 7 encoder skips (model_untils.py:535-616 of the reference):
    GAP -> Conv1d(k=3) across the concatenated channels -> 7 Linear heads -> IntensityGate.
    Returns (B, 1, C_i) gates; broadcasting replaces the reference's expand_as over L.  `live`
    restricts the heads to the ones whose result is consumed (att5..7 feed e2ds[0..2]; att1..4 feed
    only branches that Decoder.forward never reads, ADNMUNet.py:612-613)."""

    def __init__(self, c_list=[8, 16, 32, 64, 128, 256], split_att='fc'):
        super().__init__()
        if split_att != 'fc':
            raise NotImplementedError("only split_att='fc' (the reference default) is implemented")
        c_list_sum = sum(c_list)
        self.split_att = split_att
        self.avgpool = nn.AdaptiveAvgPool2d(1)
        self.get_all_att = nn.Conv1d(1, 1, kernel_size=3, padding=1, bias=True)
        for i in range(7):
            setattr(self, f"att{i + 1}", nn.Linear(c_list_sum, c_list[i]))
        self.sigmoid1 = IntensityGate()

    def forward(self, t, live=None, alias_out=None):
        """alias_out: optional list that receives autograd aliases of the skips — later consumers that read those instead of
        t[i] get their gradient summed with the pool's in one HIP pass (ops.bridge_pool)."""
        aliases, att = ops.bridge_pool([t[i] for i in range(len(t))])   # every skip's avgpool + the concat: one launch (+ one fold)
        if alias_out is not None:
            alias_out[:] = aliases
        att = att.unsqueeze(1)  # (B,1,sum C)
        att = ops.conv1d3(att, self.get_all_att.weight, self.get_all_att.bias)
        idx = [i for i in range(7) if live is None or i in live]
        lins = [getattr(self, f"att{i + 1}") for i in idx]
        # every live head = Linear + IntensityGate, all in ONE launch each way (csrc/bridge.hip)
        gates = ops.bridge_heads(att, self.sigmoid1.enhance, self.sigmoid1.threshold, [l.weight for l in lins], [l.bias for l in lins])
        return dict(zip(idx, gates))


class EncoderToDecoder(nn.Module):
    """Skip-connection gating block (model_untils.py:620-794 of the reference): gate + InstanceNorm, the pooled / grouped-conv gating
    core (csrc/skipgate.hip), FeedForward and ConvFFD — all HIP token kernels.  The nn.MaxPool2d / nn.AvgPool2d members only keep the
    reference's module tree; nothing calls them."""

    def __init__(self, embed_dim=256, InstanceNorm=True):
        super().__init__()
        g = embed_dim // 4
        mk = lambda k, p: Conv2dLayer(embed_dim, embed_dim, kernel_size=k, stride=(1, 1), padding=p, bias=True, groups=g, act_func=nn.GELU)
        pw = lambda: Conv2dLayer(embed_dim, embed_dim, kernel_size=1, stride=1, padding=0, groups=embed_dim, bias=True)
        self.conv13pool = mk((1, 3), (0, 1))
        self.ffd13 = pw()
        self.act_func13 = IntensityGate()
        self.conv31pool = mk((3, 1), (1, 0))
        self.ffd31 = pw()
        self.act_func31 = IntensityGate()
        self.conv33pool = mk((3, 3), (1, 1))
        self.ffd33 = pw()
        self.act_func33 = IntensityGate()
        self.max_pool_13 = nn.MaxPool2d((1, 3), (1, 1), (0, 1))
        self.avg_pool_13 = nn.AvgPool2d((1, 3), (1, 1), (0, 1))
        self.max_pool_31 = nn.MaxPool2d((3, 1), (1, 1), (1, 0))
        self.avg_pool_31 = nn.AvgPool2d((3, 1), (1, 1), (1, 0))
        self.max_pool_33 = nn.MaxPool2d((3, 3), (1, 1), (1, 1))
        self.avg_pool_33 = nn.AvgPool2d((3, 3), (1, 1), (1, 1))
        self.conv33 = mk((3, 3), (1, 1))
        self.ffd = FeedForward(dim=embed_dim, bias=True)
        self.act = IntensityGate()
        self.norm = nn.InstanceNorm2d(embed_dim) if InstanceNorm else nn.GroupNorm(4, embed_dim)
        self.alpha1 = nn.Parameter(torch.tensor(0.33, dtype=torch.float))
        self.alpha2 = nn.Parameter(torch.tensor(0.33, dtype=torch.float))
        self.alpha3 = nn.Parameter(torch.tensor(0.33, dtype=torch.float))
        self.gama = nn.Parameter(torch.tensor(1, dtype=torch.float))
        self.gamma = nn.Parameter(1 * torch.ones(embed_dim))
        self.mlp = ConvFFD(embed_dim, bias=True)
        self.scale = nn.Parameter(torch.tensor(1.))
        self.shift = nn.Parameter(torch.tensor(0.))

    def forward(self, x, res):
        """x: (B, L, d) skip; res: the bridge gate, (B, 1, d) or (B, L, d)."""
        b, l, d = x.shape
        h, w = _hw(l)
        if not (isinstance(self.norm, nn.InstanceNorm2d) and d % 4 == 0):
            raise RuntimeError(f"EncoderToDecoder: the HIP kernels take InstanceNorm2d and a multiple of 4 channels, got {self.norm}, dim {d} "
                               "(the HIP path has no PyTorch fallback)")
        x = ops.igate_res(x, res, self.gama, self.act.enhance, self.act.threshold)   # act(x + gama * res), one pass each way
        x = ops.instnorm(x, self.scale, self.shift, self.norm.eps, lib.ACT_NONE)
        # pools + grouped convs + gates + mix: 2 launches forward, 5-6 backward (csrc/skipgate.hip).  The reference applies
        # ffd13 / act_func13 to both the 1x3 and the 3x1 branch (:770-777); ffd31 / act_func31 / conv33 are never used.
        f13, f33 = self.ffd13.conv, self.ffd33.conv
        xp = ops.skipgate(x, h, w, (
            self.conv13pool.conv.weight, self.conv13pool.conv.bias, self.conv31pool.conv.weight, self.conv31pool.conv.bias,
            self.conv33pool.conv.weight, self.conv33pool.conv.bias, f13.weight.reshape(-1), f13.bias, f33.weight.reshape(-1), f33.bias,
            self.act_func13.enhance, self.act_func13.threshold, self.act_func33.enhance, self.act_func33.threshold,
            self.alpha1, self.alpha2, self.alpha3, self.gamma))
        return self.mlp(self.ffd.forward_tokens(xp, h, w))


class OutProj(nn.Module):
    def __init__(self, num_frames=3, embed_dim=256, img_size=[256, 256], act_func=Swish, wt_levels=2, ls_init_value=1,
                 out_expand=2, InstanceNorm=True):
        super().__init__()
        self.img_size = img_size
        self.embed_dim = embed_dim
        self.activation = act_func
        self.wtconv = WTConvLayer(embed_dim, embed_dim, kernel_size=5, stride=1, bias=False, wt_levels=3, act_func=nn.GELU,
                                  norm=nn.InstanceNorm2d(embed_dim) if InstanceNorm else nn.GroupNorm(4, embed_dim))
        self.conv = nn.Sequential(
            Conv2dLayer(embed_dim, embed_dim * out_expand, kernel_size=(3, 3), stride=(1, 1), padding=(1, 1), bias=False, act_func=nn.GELU),
            Conv2dLayer(embed_dim * out_expand, num_frames, kernel_size=(1, 1), stride=(1, 1), padding=(0, 0), bias=False, act_func=nn.GELU))
        self.conv2 = Conv2dLayer(num_frames, num_frames, kernel_size=3, stride=1, bias=False, act_func=self.activation)
        self.alpha1 = nn.Parameter(torch.tensor(1., dtype=torch.float))
        self.alpha2 = nn.Parameter(torch.tensor(1., dtype=torch.float))
        self.gamma = nn.Parameter(ls_init_value * torch.ones(embed_dim)) if ls_init_value is not None else None
        self.alpha = nn.Parameter(torch.tensor(1., dtype=torch.float))
        self.beta = nn.Parameter(torch.tensor(1., dtype=torch.float))

    def forward(self, x, residual):
        """x: (B, L, d) tokens, residual: last input frame (B, H, W) -> (B, T_out, H, W) (model_untils.py:871-892)."""
        h, w = self.img_size[0], self.img_size[1]
        b, l, d = x.shape
        y, x = self.wtconv.forward_tokens(x, h, w, tap=True)
        x = ops.lincomb([y, x], [self.alpha, self.beta], self.gamma)
        x = self.conv[0].forward_tokens(x, h, w)
        x = self.conv[1].forward_tokens(x, h, w)
        if residual is not None:   # the last input frame, broadcast over the T_out channels (model_untils.py:886-888)
            x = ops.lincomb([x, residual.reshape(b, l, 1).expand(b, l, x.shape[-1]).contiguous()], [self.alpha1, self.alpha2])
        x = self.conv2.forward_tokens(x, h, w)
        return nchw_view(x, h, w)
