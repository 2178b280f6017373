"""CPU-side checks: the C-ABI library builds for gfx950, loads, and exports every symbol that
include/saunet_hip.h declares (no compute without a GPU)."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    import saunet_amd  # noqa: F401
    from saunet_amd import _build, lib as L
    _build.build()
    return L


def test_header_symbols_are_exported(lib):
    hdr = open(os.path.join(ROOT, "include", "saunet_hip.h")).read()
    declared = set(re.findall(r"\b(saunet_[a-z0-9_]+)\s*\(", hdr))
    declared -= {"saunet_status", "saunet_dtype", "saunet_pack_mode"}
    handle = lib.load()
    missing = [s for s in sorted(declared) if not hasattr(handle, s)]
    assert not missing, missing
    assert set(lib.EXPORTS) == declared, (set(lib.EXPORTS) ^ declared)
    assert handle.saunet_version() >= 1


def test_errors_do_not_cross_the_abi_as_exceptions(lib):
    import ctypes as C
    handle = lib.load()
    d = lib.ConvDesc()  # all zeros: invalid shape -> status code + message, no launch
    rc = handle.saunet_conv2d_forward(C.byref(d), None, None, None, None, None, None, None, None, None)
    assert rc == -1
    assert b"bad shape" in handle.saunet_last_error()


def test_product_path_refuses_cpu_tensors(lib):
    import torch
    import saunet_amd
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        saunet_amd.functional.conv2d(torch.zeros(1, 8, 4, 4), torch.zeros(8, 8, 1, 1))


def test_state_dict_keys_match_reference_layout():
    """reference key set: every tensor of the oracle's spec is present, plus aliases."""
    import saunet_amd
    from oracle import saunet_ref as R
    net = saunet_amd.SAUNet(num_classes=4)
    keys = set(net.state_dict().keys())
    want = {k for k, _, _ in R.state_dict_spec()}
    assert want <= keys, sorted(want - keys)[:5]
    assert "conv2.denselayer1.norm1.weight" in keys and "conv5.1.running_mean" in keys  # aliases like the reference
    assert "res1.bn1._running_iter" in keys
    assert sum(p.numel() for p in net.parameters()) == 32896505


def test_param_grouping_matches_train_py():
    import saunet_amd
    net = saunet_amd.SAUNet(num_classes=4)
    g = saunet_amd.optim.group_weight(net)
    n_decay = sum(p.numel() for p in g[0]["params"]); n_nodecay = sum(p.numel() for p in g[1]["params"])
    assert n_decay + n_nodecay == 32896505
    assert g[1]["weight_decay"] == 0.0
    assert all(p.dim() in (2, 4) for p in g[0]["params"]) and all(p.dim() == 1 for p in g[1]["params"])
