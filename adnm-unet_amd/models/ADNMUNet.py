"""ADNM-UNet — drop-in for the reference's models/ADNMUNet.py: `create_ADNMUNet(input_frames,
output_frames, frame_interval)`, `VisionMamba`, `Encoder`, `Decoder`, `Refiner`, `Block`, `Attention`,
`create_block` with the reference's constructor arguments, forward signatures and the same 992
state_dict keys, so the reference's train.py / validate.py run unchanged against this package.

Differences that do not change results:
  * tokens stay channels-last (B, H*W, C) end to end; the hot ops are HIP kernels (adnm_hip.ops);
  * skip tensors / features are locals, not module-level dicts (the reference's dicts are shared by
    nn.DataParallel replicas and race);
  * `Decoder` uses self.img_size where the reference hard-codes 256 (ADNMUNet.py:634), so 128x128 works;
  * e2ds[3..6] / att1..4 — computed and never read by the reference (ADNMUNet.py:612-613) — are skipped
    unless `compute_dead_branches=True`; their parameters still exist and, as in the reference, never
    receive a gradient.
"""
import math
import os
from functools import partial

import torch
import torch.nn as nn
import torch.nn.functional as F

from adnm_hip import ops
from .ADNssd import Mamba2, StandardAttention
from .model_untils import *  # noqa: F401,F403  (same star-import surface as the reference, ADNMUNet.py:34)
from .model_untils import (BiasFree_LayerNorm, RMSNorm, Mlp, Swish, FeedForward, PatchEmbed, WTLayer, DownSample, UpSample,
                           Channel_Att_Bridge, EncoderToDecoder, OutProj, Conv2dLayer, DropPath, _hw)


def _merge(mod, x, residual, features):
    """Skip / feature merge at the head of Block and Attention (ADNMUNet.py:124-131, :214-221)."""
    if residual is not None:
        x = ops.catmix(x, residual, features, mod.alpha1, mod.alpha2, mod.alpha3, mod.alpha4)
    elif features is not None:
        x = ops.lincomb([x, features], [None, mod.alpha3])
    return x


def _mix_normed(norm, xs, scalars, scale, shift, gamma=None):
    """-> (scale*norm(x)+shift, x) for x = gamma*(s0*xs[0] + s1*xs[1]): the residual mix and the pre-norm that follows it as one HIP launch"""
    if isinstance(norm, (RMSNorm, BiasFree_LayerNorm)):
        return norm.mix_tap(xs, scalars, scale, shift, gamma)
    return _normed(norm, ops.lincomb(xs, scalars, gamma), scale, shift)


def _normed(norm, x, scale, shift):
    """-> (scale*norm(x)+shift, x).  With the HIP row-norms the second value is an autograd alias of x whose gradient is
    added inside the norm's backward kernel (the residual path of the pre-norm block)."""
    if isinstance(norm, (RMSNorm, BiasFree_LayerNorm)):
        return norm.tap(x, scale, shift)  # scalar affine fused into the HIP row-norm
    return scale * norm(x) + shift, x


class Block(nn.Module):
    def __init__(self, dim, out_dim, mixer, norm_layer=BiasFree_LayerNorm, fused_add_norm=False, residual_in_fp32=False,
                 drop_path=0., drop=0., patches_resolution=[64, 64], mlp_ratio=4, num_layers=1, act_layer=nn.SiLU, attn=False):
        super().__init__()
        self.residual_in_fp32 = residual_in_fp32
        self.fused_add_norm = fused_add_norm
        self.dim, self.out_dim, self.num_layers = dim, out_dim, num_layers
        self.alpha1 = nn.Parameter(torch.tensor(1, dtype=torch.float))
        self.alpha2 = nn.Parameter(torch.tensor(1, dtype=torch.float))
        self.alpha3 = nn.Parameter(torch.tensor(1, dtype=torch.float))
        self.alpha4 = nn.Parameter(torch.tensor(1, dtype=torch.float))
        self.beta1 = nn.Parameter(torch.ones(num_layers))
        self.beta2 = nn.Parameter(torch.ones(num_layers))
        self.beta3 = nn.Parameter(torch.ones(num_layers))
        self.beta4 = nn.Parameter(torch.ones(num_layers))
        self.mixer_layers = nn.ModuleList([mixer() for _ in range(num_layers)])
        self.drop_path_layers = nn.ModuleList([DropPath(drop_path) if drop_path > 0. else nn.Identity() for _ in range(num_layers)])
        self.norm1_layers = nn.ModuleList([norm_layer(dim) for _ in range(num_layers)])
        self.ffns = nn.ModuleList([FeedForward(dim=dim, ffn_expansion_factor=2, bias=True) for _ in range(num_layers)])
        self.norm2_layers = nn.ModuleList([norm_layer(dim) for _ in range(num_layers)])
        self.scale1 = nn.ParameterList([nn.Parameter(torch.tensor(1.)) for _ in range(num_layers)])
        self.shift1 = nn.ParameterList([nn.Parameter(torch.tensor(0.)) for _ in range(num_layers)])
        self.scale2 = nn.ParameterList([nn.Parameter(torch.tensor(1.)) for _ in range(num_layers)])
        self.shift2 = nn.ParameterList([nn.Parameter(torch.tensor(0.)) for _ in range(num_layers)])
        self.act = Swish()
        if self.dim != self.out_dim:
            self.out_proj = nn.Linear(dim, out_dim)
        self.gamma = nn.Parameter(1 * torch.ones(dim))

    def forward(self, hidden_states, residual=None, features=None, inference_params=None, use_checkpoint=False, then=None):
        """then: the Block that consumes this one's output directly (a chain like Refiner's refiner1 -> .. -> refiner4): this block's closing
        mix (:158, :161) and THAT block's opening norm (:149) then run as one launch, and the pair (normed, x) is handed over — the next
        block takes it as `hidden_states`."""
        if isinstance(hidden_states, tuple):   # (normed by MY norm1[0], x): handed over by the block before (see `then`)
            assert residual is None and features is None
            xn, x = hidden_states
        else:
            x = _merge(self, hidden_states, residual, features)
            xn = None
        b, l, d = x.shape
        h, w = _hw(l)
        n = self.num_layers
        if xn is None:
            xn, x = _normed(self.norm1_layers[0], x, self.scale1[0], self.shift1[0])
        for i in range(n):
            # beta3/beta4 alias beta1/beta2 in the reference (:145-146)
            beta1, beta2 = (self.beta1, self.beta2) if n == 1 else (self.beta1[i:i + 1], self.beta2[i:i + 1])
            # every residual mix (:152, :158) that is followed by a pre-norm (:155, the next layer's :149) shares that norm's launch
            xn, x = _mix_normed(self.norm2_layers[i], [x, self.drop_path_layers[i](self.mixer_layers[i](xn, h, w))], [beta1, beta2],
                                self.scale2[i], self.shift2[i])
            f = self.ffns[i].forward_tokens(xn, h, w)
            if i + 1 < n:
                xn, x = _mix_normed(self.norm1_layers[i + 1], [x, f], [beta1, beta2], self.scale1[i + 1], self.shift1[i + 1])
            elif then is not None and self.dim == self.out_dim:   # ... and the next block's opening norm rides along too
                return _mix_normed(then.norm1_layers[0], [x, f], [beta1, beta2], then.scale1[0], then.shift1[0], self.gamma)
            else:   # the block's closing per-channel gamma (:161) rides on the last residual mix: one launch instead of two
                x = ops.lincomb([x, f], [beta1, beta2], self.gamma)
        if self.dim != self.out_dim:
            x = ops.linear(x, self.out_proj.weight, self.out_proj.bias)
        return x


class Attention(nn.Module):
    def __init__(self, dim, out_dim=None, headdim=4):
        super().__init__()
        self.dim = dim
        self.out_dim = out_dim or dim
        self.attn_norm1 = BiasFree_LayerNorm(dim)
        self.attn_norm2 = BiasFree_LayerNorm(dim)
        self.attn_layer = StandardAttention(dim, heads=dim // headdim, dim_head=headdim, dropout=0.)
        self.attn_mlp = Mlp(dim)
        self.attn_scale1 = nn.Parameter(torch.tensor(1.))
        self.attn_shift1 = nn.Parameter(torch.tensor(0.))
        self.attn_scale2 = nn.Parameter(torch.tensor(1.))
        self.attn_shift2 = nn.Parameter(torch.tensor(0.))
        if self.dim != self.out_dim:
            self.out_proj = nn.Linear(dim, out_dim)
        self.gamma = nn.Parameter(1 * torch.ones(dim))
        for n in ("alpha1", "alpha2", "alpha3", "alpha4", "beta1", "beta2", "beta3", "beta4"):
            setattr(self, n, nn.Parameter(torch.tensor(1, dtype=torch.float)))

    def forward(self, hidden_states, residual=None, features=None, inference_params=None, use_checkpoint=False):
        x = _merge(self, hidden_states, residual, features)
        b, l, d = x.shape
        h, w = _hw(l)
        xn, x = self.attn_norm1.tap(x, self.attn_scale1, self.attn_shift1)
        xn, x = self.attn_norm2.mix_tap([x, self.attn_layer(xn, h, w)], [self.beta1, self.beta2], self.attn_scale2, self.attn_shift2)
        x = ops.lincomb([x, self.attn_mlp(xn)], [self.beta3, self.beta4], self.gamma)
        if self.dim != self.out_dim:
            x = ops.linear(x, self.out_proj.weight, self.out_proj.bias)
        return x


def create_block(d_model, out_dim, headdim=None, ssm_cfg=None, num_layers=1, norm_epsilon=1e-5, drop_path=0., drop=0.,
                 rms_norm=True, residual_in_fp32=True, fused_add_norm=True, layer_idx=None, d_state=16, device=None, dtype=None):
    if ssm_cfg is None:
        ssm_cfg = {}
    if headdim is None:
        headdim = 4 if d_model <= 32 else 8 if d_model <= 256 else 16 if d_model <= 512 else 24 if d_model <= 768 else 32
    mixer = partial(Mamba2, layer_idx=layer_idx, d_model=d_model, headdim=headdim, linear_attn_duality=True, d_state=d_state, **ssm_cfg)
    norm_layer = partial(nn.LayerNorm if not rms_norm else RMSNorm, eps=norm_epsilon)
    block = Block(dim=d_model, out_dim=out_dim, mixer=mixer, num_layers=num_layers, norm_layer=norm_layer, drop_path=drop_path,
                  drop=drop, fused_add_norm=fused_add_norm, residual_in_fp32=residual_in_fp32)
    block.layer_idx = layer_idx
    return block


def _init_weights(module, n_layer, initializer_range=0.02, rescale_prenorm_residual=True, n_residuals_per_layer=1):
    """Second init pass of the reference (ADNMUNet.py:294-313)."""
    if isinstance(module, nn.Linear):
        if module.bias is not None and not getattr(module.bias, "_no_reinit", False):
            nn.init.zeros_(module.bias)
    elif isinstance(module, nn.Embedding):
        nn.init.normal_(module.weight, std=initializer_range)
    if rescale_prenorm_residual:
        for name, p in module.named_parameters():
            if name in ("out_proj.weight", "fc2.weight"):
                nn.init.kaiming_uniform_(p, a=math.sqrt(5))
                with torch.no_grad():
                    p /= math.sqrt(n_residuals_per_layer * n_layer)


def segm_init_weights(m):
    """First init pass of the reference (ADNMUNet.py:316-323)."""
    if isinstance(m, nn.Linear):
        nn.init.trunc_normal_(m.weight, std=0.02, a=-2.0, b=2.0)
        if m.bias is not None:
            nn.init.constant_(m.bias, 0)
    elif isinstance(m, nn.LayerNorm):
        nn.init.constant_(m.bias, 0)
        nn.init.constant_(m.weight, 1.0)


class Encoder(nn.Module):
    def __init__(self, img_size=256, depth=[1, 1, 1], embed_dim=[64, 128, 128], headdim=8, in_channels=5, kernel=[5, 4, 3],
                 ratio=[2, 2, 2], wt_levels=[4, 3, 2], simple_patch=False, norm_epsilon=1e-5, ssm_cfg=None, InstanceNorm=True):
        super().__init__()
        if simple_patch:
            raise NotImplementedError("simple_patch=True is unused by create_ADNMUNet")
        self.img_size, self.depth, self.embed_dim = img_size, depth, embed_dim
        self.in_channels, self.kernel, self.ratio = in_channels, kernel, ratio
        self.encoder1 = PatchEmbed(img_size=img_size, patch_size=ratio[0], in_channels=in_channels, embed_dim=embed_dim[0],
                                   kernel=kernel[0], wt_levels=wt_levels[0], InstanceNorm=InstanceNorm)
        self.down_sample1 = DownSample(dim=embed_dim[0], ratio=ratio[0])
        self.encoder2 = WTLayer(embed_dim[0], embed_dim[1], kernel=kernel[1], wt_levels=wt_levels[1], InstanceNorm=InstanceNorm)
        self.down_sample2 = DownSample(dim=embed_dim[1], ratio=ratio[1])
        self.encoder3 = WTLayer(embed_dim[1], embed_dim[2], kernel=kernel[2], wt_levels=wt_levels[2], InstanceNorm=InstanceNorm)
        self.down_sample3 = DownSample(dim=embed_dim[2], ratio=ratio[2])
        self.attn = Attention(dim=embed_dim[2], headdim=headdim)
        blk = lambda i, n: create_block(d_model=embed_dim[i], out_dim=embed_dim[i + 1], headdim=headdim, num_layers=n,
                                        ssm_cfg=ssm_cfg, norm_epsilon=norm_epsilon)
        self.encoder4 = blk(2, depth[0])
        self.down_sample4 = DownSample(dim=embed_dim[3], ratio=ratio[3])
        self.encoder5 = blk(3, depth[1])
        self.down_sample5 = DownSample(dim=embed_dim[4], ratio=ratio[4])
        self.encoder6 = blk(4, depth[2])
        self.attn2 = Attention(embed_dim[5], headdim=headdim)

    def forward(self, x):
        """x: (B, T_in, H, W) -> (bottleneck tokens, [7 skips], last input frame) (ADNMUNet.py:437-483)."""
        x, lo, res = self.forward_lo(x)
        x, hi = self.forward_hi(x)
        return x, lo + hi, res

    # the same forward in two halves (a stage cut of adnm_hip.trainer.FlatTrainer): the conv stages down to 16x16, then the mixer stages
    def forward_lo(self, x):
        x = x.flatten(2).transpose(1, 2)
        skips = []
        # a stage's output is pooled AND kept as a skip: the skip list gets the pool's alias of it (one gradient sum inside maxpool_bwd)
        x, res = self.encoder1(x)
        d, x = self.down_sample1.tap(x)
        skips.append(x)
        x = self.encoder2(d)
        d, x = self.down_sample2.tap(x)
        skips.append(x)
        x = self.encoder3(d)
        d, x = self.down_sample3.tap(x)
        skips.append(x)
        x = self.attn(d)
        skips.append(x)
        return x, skips, res

    def forward_hi(self, x):
        skips = []
        x = self.encoder4(x)
        d, x = self.down_sample4.tap(x)
        skips.append(x)
        x = self.encoder5(d)
        d, x = self.down_sample5.tap(x)
        skips.append(x)
        x = self.encoder6(d)
        skips.append(x)
        x = self.attn2(x)
        return x, skips

    def lo_modules(self):
        return [self.encoder1, self.down_sample1, self.encoder2, self.down_sample2, self.encoder3, self.down_sample3, self.attn]

    def hi_modules(self):
        return [self.encoder4, self.down_sample4, self.encoder5, self.down_sample5, self.encoder6, self.attn2]


class Decoder(nn.Module):
    def __init__(self, img_size=256, depth=[1, 1, 1], embed_dim=[64, 128, 128], headdim=8, refine_dim=[32, 32, 32], kernel=[5, 4, 3],
                 ratio=[2, 2, 2], wt_levels=[4, 3, 2], simple_patch=False, norm_epsilon=1e-5, ssm_cfg=None, InstanceNorm=True,
                 compute_dead_branches=False):
        super().__init__()
        self.img_size, self.depth, self.embed_dim, self.kernel, self.ratio = img_size, depth, embed_dim, kernel, ratio
        self.compute_dead_branches = compute_dead_branches
        blk = lambda d, o, n: create_block(d_model=d, out_dim=o, headdim=headdim, num_layers=n, ssm_cfg=ssm_cfg, norm_epsilon=norm_epsilon)
        self.decoder1 = blk(embed_dim[5], embed_dim[4], depth[2])
        self.up_sample1 = UpSample(dim=embed_dim[4], ratio=ratio[4])
        self.decoder2 = blk(embed_dim[4] * 2, embed_dim[3], depth[1])
        self.up_sample2 = UpSample(dim=embed_dim[3], ratio=ratio[3])
        self.decoder3 = blk(embed_dim[3] * 2, embed_dim[2], depth[0])
        self.attn = Attention(embed_dim[2], embed_dim[2], headdim=headdim)
        self.up_sample3 = UpSample(dim=embed_dim[2], ratio=ratio[2])
        self.decoder4 = WTLayer(embed_dim[2] * 2, embed_dim[1], kernel=kernel[2], wt_levels=wt_levels[2], if_res=True, InstanceNorm=InstanceNorm)
        self.up_sample4 = UpSample(dim=embed_dim[1], ratio=ratio[1])
        self.decoder5 = WTLayer(embed_dim[1] * 2, embed_dim[0], kernel=kernel[1], wt_levels=wt_levels[1], if_res=True, InstanceNorm=InstanceNorm)
        self.up_sample5 = UpSample(dim=embed_dim[0], ratio=ratio[0])
        self.decoder6 = WTLayer(embed_dim[0] * 2, embed_dim[0], kernel=kernel[0], wt_levels=wt_levels[0], if_res=True, InstanceNorm=InstanceNorm)
        self.decoder6_s = Conv2dLayer(embed_dim[0], refine_dim[0], kernel_size=1, stride=1, padding=0)
        c_list = list(embed_dim)
        c_list.insert(2, c_list[2])  # the attention stage's skip (the reference mutates the shared list, ADNMUNet.py:590)
        embed_dim.insert(2, embed_dim[2])
        self.fusion = Channel_Att_Bridge(c_list=c_list)
        self.e2ds = nn.ModuleList([EncoderToDecoder(embed_dim=c_list[len(c_list) - 1 - i], InstanceNorm=InstanceNorm)
                                   for i in range(len(c_list))])

    def forward(self, x, skips):
        """(ADNMUNet.py:603-636 of the reference).  skips[i] = encoder_layer_residual[i]."""
        skips, feats = self.forward_skips(skips)
        return self.forward_blocks(x, skips, feats)

    # the same forward in two halves (a stage cut of adnm_hip.trainer.FlatTrainer): the skip path (channel-attention bridge + the
    # EncoderToDecoder gates), then the decoder blocks
    def forward_skips(self, skips):
        """-> (skip aliases, feats): feats[i] = e2ds[i](skip 6 - i) for the live branches (i < 3; all 7 with compute_dead_branches)."""
        dead = self.compute_dead_branches
        aliases = []
        gates = self.fusion(skips, live=None if dead else {4, 5, 6}, alias_out=aliases)
        skips = aliases   # same tensors; their gradients now meet the bridge pool's inside one kernel
        feats = [self.e2ds[i](x=skips[6 - i], res=gates[6 - i]) for i in range(7 if dead else 3)]
        return skips, feats

    def forward_blocks(self, x, skips, feats):
        f = lambda i: feats[i] if i < len(feats) else None
        x = self.up_sample1(self.decoder1(x, features=f(0)))
        x = self.up_sample2(self.decoder2(x, residual=skips[5], features=f(1)))
        x = self.decoder3(x, residual=skips[4], features=f(2))
        x = self.up_sample3(self.attn(x))
        x = self.up_sample4(self.decoder4(x, residual=skips[2], features=f(4)))
        x = self.up_sample5(self.decoder5(x, residual=skips[1], features=f(5)))
        x = self.decoder6(x, residual=skips[0], features=f(6))
        return self.decoder6_s.forward_tokens(x, self.img_size, self.img_size)

    def skip_modules(self):
        return [self.fusion, self.e2ds]

    def block_modules(self):
        return [self.decoder1, self.up_sample1, self.decoder2, self.up_sample2, self.decoder3, self.attn, self.up_sample3, self.decoder4,
                self.up_sample4, self.decoder5, self.up_sample5, self.decoder6, self.decoder6_s]


class Refiner(nn.Module):
    def __init__(self, img_size=256, refine_depth=[1, 1, 1, 1], refine_dim=[64, 128, 128], wt_levels=[4, 3, 2], out_channels=3,
                 refine_headdim=[4, 4, 4, 4], norm_epsilon=1e-5, out_expand=2, ssm_cfg=None, InstanceNorm=True):
        super().__init__()
        self.img_size, self.refine_depth = img_size, refine_depth
        blk = lambda i, o: create_block(d_model=refine_dim[i], out_dim=refine_dim[o], num_layers=refine_depth[i], ssm_cfg=ssm_cfg,
                                        norm_epsilon=norm_epsilon, headdim=refine_headdim[i])
        self.refiner1 = blk(0, 1)
        self.refiner2 = blk(1, 2)
        self.refiner3 = blk(2, 3)
        self.refiner4 = blk(3, -1)
        self.out_proj = OutProj(num_frames=out_channels, embed_dim=refine_dim[-1], img_size=[img_size, img_size],
                                wt_levels=wt_levels[0], out_expand=out_expand, InstanceNorm=InstanceNorm)

    def forward(self, x, res):
        # the four blocks are a chain: each block's closing mix shares a launch with the next block's opening norm (Block.forward: `then`)
        x = self.refiner1(x, then=self.refiner2)
        x = self.refiner2(x, then=self.refiner3)
        x = self.refiner3(x, then=self.refiner4)
        return self.out_proj(self.refiner4(x), res)


class VisionMamba(nn.Module):
    def __init__(self, img_size=256, depth=[1, 1, 1], refine_depth=[1, 1], refine_dim=[32, 32, 32], refine_headdim=[8, 4],
                 embed_dim=[64, 128, 128], headdim=8, channels=5, out_channels=3, ssm_cfg=None, norm_epsilon=1e-5,
                 initializer_cfg=None, kernel=[5, 4, 3], ratio=[2, 2, 2], wt_levels=[4, 3, 2], out_expand=2, InstanceNorm=True,
                 simple_patch=False, e2d=True, compute_dead_branches=False, **kwargs):
        super().__init__()
        self.depth = depth
        self.encoder = Encoder(img_size=img_size, depth=depth, embed_dim=embed_dim, headdim=headdim, in_channels=channels, kernel=kernel,
                               ratio=ratio, wt_levels=wt_levels, simple_patch=simple_patch, norm_epsilon=norm_epsilon, InstanceNorm=InstanceNorm)
        self.decoder = Decoder(img_size=img_size, depth=depth, embed_dim=embed_dim, headdim=headdim, refine_dim=refine_dim, kernel=kernel,
                               ratio=ratio, wt_levels=wt_levels, norm_epsilon=norm_epsilon, InstanceNorm=InstanceNorm,
                               compute_dead_branches=compute_dead_branches)
        self.refiner = Refiner(img_size=img_size, refine_depth=refine_depth, refine_dim=refine_dim, refine_headdim=refine_headdim,
                               out_channels=out_channels, wt_levels=wt_levels, out_expand=out_expand, norm_epsilon=norm_epsilon,
                               InstanceNorm=InstanceNorm)
        self.apply(segm_init_weights)
        self.apply(partial(_init_weights, n_layer=len(depth), **(initializer_cfg if initializer_cfg is not None else {})))

    @torch.jit.ignore
    def no_weight_decay(self):
        return {"pos_embed", "cls_token", "temporal_pos_embedding"}

    def forward(self, x, mask=None):
        """x: (B, T_in, 1, H, W) float32 -> (B, T_out, 1, H, W) (ADNMUNet.py:824-829 of the reference)."""
        return self.forward_stage2(*self.forward_stage1(x))

    # The same forward, cut into stages.  adnm_hip.trainer.FlatTrainer uses the cuts for a staged backward on multi-GPU runs: the gradients
    # of a stage are all-reduced over xGMI while the backward of the stages before it runs (refiner -> decoder ->
    # e2ds / fusion -> encoder4-6 -> encoder1-3).  forward_stages() = [(function, modules whose parameters it owns), ...] in forward
    # order; every function takes and returns a flat tuple of tensors.  forward_stage1 / forward_stage2 is the older two-stage form
    # (encoder | decoder + refiner) of the same cut.
    def forward_stages(self):
        enc, dec = self.encoder, self.decoder

        def s0(x):                      # encoder1-3 + attn: 128x128 .. 16x16 conv stages
            ops.prep_group(*enc.lo_modules())
            x, skips, res = enc.forward_lo(x.squeeze(2))
            return (x, res, *skips)

        def s1(x, res, *lo):            # encoder4-6 + attn2: the mixer stages
            ops.prep_group(*enc.hi_modules())
            x, hi = enc.forward_hi(x)
            return (x, res, *lo, *hi)

        def s2(x, res, *skips):         # channel-attention bridge + EncoderToDecoder gates
            skips, feats = dec.forward_skips(list(skips))
            return (x, res, *skips, *feats)

        def s3(x, res, *rest):          # decoder blocks
            ops.prep_group(*dec.block_modules())
            return (dec.forward_blocks(x, list(rest[:7]), list(rest[7:])), res)

        def s4(x, res):                 # refiner + output head
            ops.prep_group(self.refiner)
            return (self.refiner(x, res).unsqueeze(2),)

        return [(s0, enc.lo_modules()), (s1, enc.hi_modules()), (s2, dec.skip_modules()), (s3, dec.block_modules()), (s4, [self.refiner])]

    def forward_stage1(self, x):
        ops.prep_group(self.encoder)   # kernel-layout parameters of all its mixers / WTConv2ds: one launch per kind
        x, skips, res = self.encoder(x.squeeze(2))
        return (x, res, *skips)

    def forward_stage2(self, x, res, *skips):
        ops.prep_group(self.decoder, self.refiner)
        return self.refiner(self.decoder(x, list(skips)), res).unsqueeze(2)

    def stage1_parameters(self):
        return self.encoder.parameters()


def get_scalar_parameters(model):
    return [p for p in model.parameters() if p.requires_grad and p.nelement() == 1]


def create_vm(img_size=256, depth=[1, 1, 1], refine_depth=[1, 1, 1, 1], refine_headdim=[4, 4, 4, 4], refine_dim=[32, 32, 32, 32],
              embed_dim=[32, 64, 128, 256, 512, 1024], headdim=4, channels=3, out_channels=3, ssm_cfg=None, norm_epsilon=1e-6,
              kernel=[5, 3, 3], ratio=[2, 2, 2, 2, 2, 2], wt_levels=[3, 1, 1], out_expand=2, InstanceNorm=True, initializer_cfg=None):
    return VisionMamba(img_size=img_size, depth=depth, refine_depth=refine_depth, refine_headdim=refine_headdim, refine_dim=refine_dim,
                       embed_dim=list(embed_dim), headdim=headdim, channels=channels, out_channels=out_channels, ssm_cfg=ssm_cfg,
                       norm_epsilon=norm_epsilon, initializer_cfg=initializer_cfg, kernel=kernel, ratio=ratio, wt_levels=wt_levels,
                       out_expand=out_expand, InstanceNorm=InstanceNorm)


def videomamba_middle(pretrained=False, **kwargs):
    model = create_vm(img_size=256, channels=5, norm_epsilon=1e-5, **kwargs)
    model.default_cfg = {}
    return model


def create_ADNMUNet(input_frames, output_frames, frame_interval, img_size=256, **overrides):
    """Same recipe as the reference factory (ADNMUNet.py:906-940).  `img_size` (default 256, the
    reference's hard-wired value) is the one extra, optional argument: BASELINE configs use 128."""
    refine_dim = [32, 32, 32, 32] if output_frames > 5 else [32, 32, 16, 16]
    if frame_interval < 120 / input_frames:
        InstanceNorm, kernel = True, [5, 5, 5]
    else:
        InstanceNorm, kernel = False, [5, 3, 3]
    model = VisionMamba(img_size=img_size, depth=[1, 1, 1], refine_depth=[1, 1, 1, 1], refine_headdim=[4, 4, 4, 4], refine_dim=refine_dim,
                        embed_dim=[32, 64, 128, 256, 512, 1024], headdim=4, channels=input_frames, out_channels=output_frames,
                        ssm_cfg=None, norm_epsilon=1e-6, initializer_cfg=None, kernel=kernel, ratio=[2, 2, 2, 2, 2, 2],
                        wt_levels=[3, 2, 1], out_expand=2, InstanceNorm=InstanceNorm, **overrides)
    # one process per GPU (WORLD_SIZE > 1): an unmodified train.py gets bucketed RCCL gradient averaging during backward
    # (adnm_hip.ddp.attach) in place of its nn.DataParallel branch (train.py:99-102)
    if int(os.environ.get("WORLD_SIZE", "1")) > 1 and os.environ.get("ADNM_AUTO_DDP", "1") == "1":
        from adnm_hip import ddp
        ddp.attach(model)
    return model
