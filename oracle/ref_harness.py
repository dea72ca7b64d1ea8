"""CONTAINER-ONLY harness that imports the reference's own Python (read-only, from
/root/reference) so that golden vectors can be generated from it.

TEST INFRASTRUCTURE — never imported by the product path, never shipped as a
dependency, and inert on the GPU box (there is no /root/reference there).

The reference imports four third-party packages that are not installed here
(`timm`, `pywt`, `mamba_ssm`, `torchvision`).  Only the handful
of names it actually touches are provided as in-memory module objects:

  * timm: DropPath / to_2tuple / trunc_normal_ / _cfg / register_model /
    _load_weights (+ names imported but never used by model_untils.py:14)
  * pywt.Wavelet('db1'): the Haar coefficients (WTConv2d.py:10-21)
  * mamba_ssm...layer_norm.RMSNorm: y = x * rsqrt(mean(x^2)+eps) * w in fp32,
    the formula the reference README states (README.md:22-30); the Triton scan
    entry points raise if called (they are unreachable from create_ADNMUNet).
  * literal `.to('cuda')` on index tensors (ADNssd.py:329-330,...) is mapped to
    'cpu'.

Nothing under /root/reference is modified or copied.
"""
import os
import sys
import types
import math
import importlib

import torch
import torch.nn as nn

REF_ROOT = os.environ.get("ADNM_REFERENCE_ROOT", "/root/reference")


def available():
    return os.path.isdir(os.path.join(REF_ROOT, "models"))


def _mod(name):
    m = types.ModuleType(name)
    m.__path__ = []  # behave as a package
    sys.modules[name] = m
    return m


class _DropPath(nn.Module):
    def __init__(self, p=0.0):
        super().__init__()
        self.p = p

    def forward(self, x):
        assert self.p == 0.0 or not self.training
        return x


class _RMSNorm(nn.Module):
    def __init__(self, hidden_size, eps=1e-5, **kw):
        super().__init__()
        self.eps = eps
        self.weight = nn.Parameter(torch.ones(hidden_size))
        self.register_parameter("bias", None)

    def forward(self, x):
        xf = x.float()
        y = xf * torch.rsqrt(xf.pow(2).mean(-1, keepdim=True) + self.eps)
        return (y * self.weight.float()).to(x.dtype)


def _unavailable(*a, **k):
    raise RuntimeError("mamba_ssm Triton kernel is not vendored by the reference (parity unpinned)")


def _install_stubs():
    if "timm" in sys.modules and getattr(sys.modules["timm"], "_adnm_stub", False):
        return
    timm = _mod("timm")
    timm._adnm_stub = True
    layers = _mod("timm.layers")
    models = _mod("timm.models")
    mlayers = _mod("timm.models.layers")
    vit = _mod("timm.models.vision_transformer")
    timm.layers, timm.models = layers, models
    models.layers, models.vision_transformer = mlayers, vit

    def to_2tuple(x):
        return tuple(x) if isinstance(x, (tuple, list)) else (x, x)

    def trunc_normal_(t, mean=0.0, std=1.0, a=-2.0, b=2.0):
        return nn.init.trunc_normal_(t, mean=mean, std=std, a=a, b=b)

    for m in (layers, mlayers):
        m.DropPath = _DropPath
        m.to_2tuple = to_2tuple
        m.trunc_normal_ = trunc_normal_
        # imported by model_untils.py:14 but never used on the ADNM-UNet path
        for unused in ("AvgPool2dSame", "Mlp", "GlobalResponseNormMlp", "LayerNorm2d", "LayerNorm",
                       "create_conv2d", "get_act_layer", "make_divisible", "to_ntuple"):
            setattr(m, unused, None)
    vit._cfg = lambda **kw: dict(kw)
    vit._load_weights = _unavailable
    models.register_model = lambda f: f

    pywt = _mod("pywt")
    _mod("pywt.data")
    s = 1.0 / math.sqrt(2.0)

    class Wavelet:
        def __init__(self, name):
            assert name == "db1"
            self.dec_lo, self.dec_hi = [s, s], [-s, s]
            self.rec_lo, self.rec_hi = [s, s], [s, -s]

    pywt.Wavelet = Wavelet

    ms = _mod("mamba_ssm")
    ops = _mod("mamba_ssm.ops")
    tri = _mod("mamba_ssm.ops.triton")
    ln = _mod("mamba_ssm.ops.triton.layer_norm")
    ssd = _mod("mamba_ssm.ops.triton.ssd_combined")
    lng = _mod("mamba_ssm.ops.triton.layernorm_gated")
    ssu = _mod("mamba_ssm.ops.triton.selective_state_update")
    ms.ops, ops.triton = ops, tri
    ln.RMSNorm, ln.layer_norm_fn, ln.rms_norm_fn = _RMSNorm, _unavailable, _unavailable
    ssd.mamba_chunk_scan_combined = _unavailable
    ssd.mamba_split_conv1d_scan_combined = _unavailable
    lng.RMSNorm = _RMSNorm
    ssu.selective_state_update = _unavailable

    tv = _mod("torchvision")
    tvt = _mod("torchvision.transforms")
    tvf = _mod("torchvision.transforms.functional")
    tv.transforms, tvt.functional = tvt, tvf
    _mod("cv2")

    # literal .to('cuda') on index tensors -> stay on the tensor's device (cpu here)
    if not torch.cuda.is_available():
        _orig_to = torch.Tensor.to

        def _to(self, *args, **kwargs):
            if args and isinstance(args[0], str) and args[0].startswith("cuda"):
                args = ("cpu",) + tuple(args[1:])
            return _orig_to(self, *args, **kwargs)

        torch.Tensor.to = _to


_REF = None


def load_reference():
    """Returns a namespace with the reference modules: ADNMUNet, ADNssd, Vssd,
    WTConv2d, model_untils, loss."""
    global _REF
    if _REF is not None:
        return _REF
    if not available():
        raise RuntimeError("reference tree not present (this harness only runs in the build container)")
    _install_stubs()
    # the reference uses the top-level package name `models`; make sure ours is not shadowing it
    for k in [k for k in sys.modules if k == "models" or k.startswith("models.")]:
        del sys.modules[k]
    saved = list(sys.path)
    sys.path.insert(0, REF_ROOT)
    try:
        ns = types.SimpleNamespace()
        ns.ADNMUNet = importlib.import_module("models.ADNMUNet")
        ns.ADNssd = importlib.import_module("models.ADNssd")
        ns.Vssd = importlib.import_module("models.Vssd")
        ns.WTConv2d = importlib.import_module("models.WTConv2d")
        ns.model_untils = importlib.import_module("models.model_untils")
        ns.loss = importlib.import_module("models.loss")
    finally:
        sys.path[:] = saved
        # leave `models.*` of the reference cached under private names only
        for k in [k for k in sys.modules if k == "models" or k.startswith("models.")]:
            sys.modules["_adnm_ref_" + k] = sys.modules.pop(k)
    _REF = ns
    return ns


def build_visionmamba(img_size, channels=5, out_channels=20, batchless=False):
    """create_ADNMUNet's exact hyper-parameters (ADNMUNet.py:906-940, frame_interval=6 ->
    InstanceNorm=True, kernel=[5,5,5]) at a chosen img_size.  img_size != 256 needs the
    literal 256 of Decoder.forward (ADNMUNet.py:634) generalised to img_size; that is done
    by handing that one .view() call a tensor subclass that rewrites (b,256,256,d) to
    (b,S,S,d) - the reference file is untouched and none of it is restated here."""
    ref = load_reference()
    A = ref.ADNMUNet
    model = A.VisionMamba(
        img_size=img_size, depth=[1, 1, 1], refine_depth=[1, 1, 1, 1], refine_headdim=[4, 4, 4, 4],
        refine_dim=[32, 32, 32, 32] if out_channels > 5 else [32, 32, 16, 16],
        embed_dim=[32, 64, 128, 256, 512, 1024], headdim=4, channels=channels, out_channels=out_channels,
        ssm_cfg=None, norm_epsilon=1e-6, initializer_cfg=None, kernel=[5, 5, 5], ratio=[2, 2, 2, 2, 2, 2],
        wt_levels=[3, 2, 1], out_expand=2, InstanceNorm=True)
    if img_size != 256:
        size = img_size

        class _ViewFix(torch.Tensor):
            def view(self, *shape):
                if len(shape) == 4 and tuple(shape[1:3]) == (256, 256):
                    shape = (shape[0], size, size, shape[3])
                return super().view(*shape).as_subclass(torch.Tensor)

        # decoder6's output is the tensor on which ADNMUNet.py:634 calls .view(b,256,256,d)
        model.decoder.decoder6.register_forward_hook(lambda m, i, o: o.as_subclass(_ViewFix))
    return model
