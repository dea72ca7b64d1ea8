"""CPU-side checks of the drop-in boundary: the C-ABI library builds/loads and exports every
symbol include/adnm_hip.h declares; the product path refuses to run without a GPU."""
import ctypes
import os
import pytest
import torch

from adnm_hip import lib, ops


def test_header_parses_and_library_exports_every_symbol():
    protos = lib.parse_header()
    assert len(protos) >= 18
    assert os.path.exists(lib.LIB_PATH), "libadnm_hip.so not built (python adnm-unet_amd/build.py)"
    so = ctypes.CDLL(lib.LIB_PATH)
    for name in protos:
        assert hasattr(so, name), f"{name} declared in include/adnm_hip.h but not exported"
    assert lib.load().adnm_abi_version() == 10


def test_ws_queries_are_pure_host_functions():
    assert lib.query("adnm_rownorm_bwd_ws_bytes", 65536, 32) > 0
    assert lib.query("adnm_ssd_ws_bytes", 4, 16384, 16, 4, 16, 2) > 0
    assert lib.query("adnm_dwconv_bwd_ws_bytes", 4, 128, 128, 128, 3, 3) > 0
    assert lib.query("adnm_instnorm_ws_bytes", 4, 16384, 32) > 0


def test_argument_validation_happens_before_any_launch():
    # invalid shapes are rejected on the host with an error string, no GPU needed
    rc = lib.load().adnm_rownorm_fwd(1, 6, 1, None, None, None, 1, 6, None, 1, 4, 6, 1e-5, 0, 0, None)
    assert rc == -1 and "multiple of 4" in lib.last_error()
    rc = lib.load().adnm_ssd_reduce_fwd(1, 64, 1, 16, 1, 16, 1, 16, 1, 1, 1, 1, 1, 1, 64, 1, None, None, None, 0, None, None, 0.0, 1, 0, 1, 4, 4, 16, 16, 1, 0, None)
    assert rc == -1 and "not in" in lib.last_error()


@pytest.mark.skipif(torch.cuda.is_available(), reason="checks the no-GPU behaviour")
def test_no_cpu_fallback():
    with pytest.raises(RuntimeError, match="GPU only"):
        ops.rownorm(torch.zeros(4, 8), torch.ones(8), None, None, None, 1e-5, True)


@pytest.mark.skipif(torch.cuda.is_available(), reason="checks the no-GPU behaviour")
def test_loss_and_model_refuse_cpu_tensors():
    """the loss module and the model's parameter preparation raise on the CPU like every kernel wrapper (no silent torch path)."""
    from models.loss import enRainfallLoss
    with pytest.raises(RuntimeError, match="GPU only"):
        enRainfallLoss(0.57, 0.25, 0.0)(torch.zeros(1, 2, 1, 8, 8), torch.zeros(1, 2, 1, 8, 8))
    from models.ADNMUNet import create_ADNMUNet
    model = create_ADNMUNet(5, 20, 6, img_size=64)
    with pytest.raises(RuntimeError, match="GPU only"):
        model(torch.zeros(1, 5, 1, 64, 64))
