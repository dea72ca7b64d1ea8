"""WTConv2d — drop-in for the reference's models/WTConv2d.py (same constructor, forward and
state_dict keys: wt_filter, iwt_filter, base_conv.*, base_scale.weight, wavelet_convs.i.weight,
wavelet_scale.i.weight), executed by hand-written HIP kernels on the channels-last token layout.

Reference behaviour (WTConv2d.py:100-153): a `wt_levels`-deep Haar (db1) pyramid; at each level a
depthwise KxK conv + per-channel scale on the 4C sub-bands of the RAW low-pass of the level above;
reconstruction from the deepest level, adding the deeper reconstruction to the convolved LL band;
plus base_scale * base_conv(x).  Here: adnm_haar_dwt / adnm_dwconv_fwd / adnm_haar_idwt through
adnm_hip.ops.WTConvFn (hand-written backward).  The per-channel scales are folded into the conv taps
(tiny differentiable parameter ops), so no scale pass over the activations exists.
"""
import math
import torch
import torch.nn as nn
import torch.nn.functional as F

from adnm_hip import ops


def create_wavelet_filter(wave, in_size, out_size, type=torch.float):
    """Frozen Haar analysis / synthesis filter banks with the reference's layout (WTConv2d.py:9-29):
    (4*C, 1, 2, 2), channel c*4+k, k = LL, (rows differ), (cols differ), (diagonal)."""
    if wave != "db1":
        raise NotImplementedError("only the Haar wavelet ('db1') is implemented")
    s = torch.tensor(1.0 / math.sqrt(2.0), dtype=type)
    h = float(s * s)  # the reference stores float32(1/sqrt2)^2 = 0.49999997, keep the checkpoints bit-compatible
    bank = torch.tensor([[[h, h], [h, h]], [[h, h], [-h, -h]], [[h, -h], [h, -h]], [[h, -h], [-h, h]]], dtype=type)
    dec = bank[:, None].repeat(in_size, 1, 1, 1)
    rec = bank[:, None].repeat(out_size, 1, 1, 1)
    return dec, rec


class _ScaleModule(nn.Module):
    def __init__(self, dims, init_scale=1.0, init_bias=0):
        super().__init__()
        self.dims = dims
        self.weight = nn.Parameter(torch.ones(*dims) * init_scale)
        self.bias = None

    def forward(self, x):
        return torch.mul(self.weight, x)


def _pad_rows(t, n):
    if t is None or t.shape[0] == n:
        return t
    return torch.cat([t, t.new_zeros((n - t.shape[0],) + tuple(t.shape[1:]))], 0)


class WTConv2d(nn.Module):
    def __init__(self, in_channels, out_channels, kernel_size=5, stride=1, bias=True, wt_levels=2, wt_type='db1'):
        super().__init__()
        assert in_channels == out_channels
        self.in_channels = in_channels
        self.wt_levels = wt_levels
        self.stride = stride
        self.dilation = 1
        self.kernel_size = kernel_size
        wt, iwt = create_wavelet_filter(wt_type, in_channels, in_channels, torch.float)
        self.wt_filter = nn.Parameter(wt, requires_grad=False)
        self.iwt_filter = nn.Parameter(iwt, requires_grad=False)
        self.base_conv = nn.Conv2d(in_channels, in_channels, kernel_size, padding='same', stride=1, dilation=1,
                                   groups=in_channels, bias=bias)
        self.base_scale = _ScaleModule([1, in_channels, 1, 1])
        self.wavelet_convs = nn.ModuleList(
            [nn.Conv2d(in_channels * 4, in_channels * 4, kernel_size, padding='same', stride=1, dilation=1,
                       groups=in_channels * 4, bias=False) for _ in range(wt_levels)])
        self.wavelet_scale = nn.ModuleList(
            [_ScaleModule([1, in_channels * 4, 1, 1], init_scale=0.1) for _ in range(wt_levels)])
        if stride > 1:
            self.stride_filter = nn.Parameter(torch.ones(in_channels, 1, 1, 1), requires_grad=False)

    adnm_prep_kind = "wt"   # adnm_hip.ops.prep_group prepares every WTConv2d of a model stage in one launch

    def adnm_prep_args(self):
        C = self.in_channels
        b = self.base_conv.bias
        return ((C, (C + 3) // 4 * 4, self.kernel_size, self.wt_levels, b is not None),
                ([b] if b is not None else []) + [self.base_conv.weight] + [c.weight for c in self.wavelet_convs] +
                [self.base_scale.weight] + [m.weight for m in self.wavelet_scale])

    def _taps(self):
        """Tap-major fp32 taps with base_scale / wavelet_scale folded in; channels zero-padded to a multiple of 4 (the 5-frame input
        stage) so every kernel works on 16-byte channel quads (csrc/paramprep.hip): prepared for the whole model stage in one launch
        by ops.prep_group, or here for this module alone."""
        pre = self.__dict__.pop("_adnm_prepped", None)
        if pre is not None:
            return pre
        C = self.in_channels
        Cp = (C + 3) // 4 * 4
        bias, base, levels = ops.wt_prep(C, Cp, self.kernel_size, self.wt_levels, self.base_conv.bias,
                                         [self.base_conv.weight] + [c.weight for c in self.wavelet_convs],
                                         [self.base_scale.weight] + [m.weight for m in self.wavelet_scale])
        return Cp, base, bias, levels

    def forward_tokens(self, x, H, W, tap=False):
        """x: (B, H*W, C) channels-last tokens -> same shape.  tap=True: -> (y, alias of x) for x's other consumer (ops.WTConvFn)."""
        C = self.in_channels
        Cp, base, bias, levels = self._taps()
        if Cp != C:
            if tap:
                raise RuntimeError("WTConv2d: the input alias is only offered for channel counts that are multiples of 4")
            x = ops.chanpad(x, Cp)   # the 5-frame input stage runs on 8 channels (HIP pad / crop, csrc/elementwise.hip)
        out = ops.wtconv(x, H, W, self.kernel_size, base, bias, levels, tap=tap)
        if Cp != C:
            out = ops.chanpad(out, C)
        return out

    def forward(self, x):
        """x: (B, C, H, W), as the reference (WTConv2d.py:100)."""
        B, C, H, W = x.shape
        y = self.forward_tokens(x.permute(0, 2, 3, 1).reshape(B, H * W, C), H, W)
        y = y.reshape(B, H, W, C).permute(0, 3, 1, 2)
        if self.stride > 1:  # do_stride: a ones 1x1 depthwise conv with stride s == subsampling (WTConv2d.py:93-96,149-151)
            y = y[:, :, ::self.stride, ::self.stride]
        return y
