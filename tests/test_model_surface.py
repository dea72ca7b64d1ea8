"""The drop-in surface (CPU): import paths, factory signature, the 992 state_dict keys / shapes of the
reference, the init constants the parameter recipe relies on, frozen Haar filters, and that the
recipe-filled parameters are bit-identical to the ones the golden fixtures were generated with."""
import json
import os
import torch

from adnm_hip import recipe
from util import GOLDEN


def manifest():
    with open(os.path.join(GOLDEN, "state_dict_manifest.json")) as f:
        return json.load(f)


def test_reference_import_paths():
    from models.ADNMUNet import create_ADNMUNet, VisionMamba, Block, Attention, Encoder, Decoder, Refiner, create_block  # noqa
    from models.ADNssd import Mamba2, StandardAttention  # noqa
    from models.Vssd import Mamba2 as V  # noqa
    from models.WTConv2d import WTConv2d  # noqa
    from models.model_untils import (BiasFree_LayerNorm, Mlp, Conv2dLayer, WTConvLayer, DeConv2dLayer, Swish, FeedForward, ConvFFD,  # noqa
                                     PatchEmbed, WTLayer, DownSample, UpSample, IntensityGate, Channel_Att_Bridge, EncoderToDecoder, OutProj)
    from models.loss import enRainfallLoss  # noqa


def test_state_dict_matches_reference():
    from models.ADNMUNet import create_ADNMUNet
    m = manifest()
    model = create_ADNMUNet(5, 20, 6)
    sd = model.state_dict()
    assert set(sd) == set(m), (sorted(set(sd) - set(m))[:5], sorted(set(m) - set(sd))[:5])
    trainable = {k: p.requires_grad for k, p in model.named_parameters()}
    for k, v in sd.items():
        assert list(v.shape) == m[k]["shape"], k
        assert trainable[k] == m[k]["trainable"], k
        if m[k]["const"] is not None and k.split(".")[-1] != "bias":
            assert bool((v == m[k]["const"]).all()), f"{k}: init constant differs from the reference"
        if not m[k]["trainable"]:  # frozen Haar banks: values must equal the reference's
            assert abs(float(v.double().sum()) - m[k]["sum"]) < 1e-9 * m[k]["abs"] + 1e-9, k
            assert abs(float(v.double().abs().sum()) - m[k]["abs"]) < 1e-9 * m[k]["abs"], k
    assert sum(p.numel() for p in model.parameters()) == 73096309
    recipe.fill_parameters(model)
    for k, v in model.state_dict().items():
        if m[k]["trainable"]:
            assert abs(float(v.double().sum()) - m[k]["sum"]) <= 1e-6 * max(1.0, m[k]["abs"]), k


def test_factory_variants_build():
    from models.ADNMUNet import create_ADNMUNet
    a = create_ADNMUNet(5, 3, 60)       # LAPS recipe: refine_dim [32,32,16,16], GroupNorm, kernel [5,3,3]
    assert a.refiner.refiner3.out_dim == 16
    b = create_ADNMUNet(10, 40, 6, img_size=128)
    assert b.decoder.img_size == 128
