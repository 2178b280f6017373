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
    # the library answers the ABI version of the header it was built from (the struct layout changed in round 5 under version 1)
    assert handle.saunet_version() == int(re.search(r"#define SAUNET_ABI_VERSION (\d+)", hdr).group(1)) == lib.ABI_VERSION >= 3


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


def test_product_sources_have_no_global_switches():
    """include/saunet_hip.h: 'nothing here allocates, synchronises'; no process-global mutable behaviour.  The kernel sources must not call
    getenv / hipMalloc / hipFree / hipDeviceSynchronize (A/B switches live behind -DSAUNET_AB_SWITCHES in common.h, variant builds only)."""
    csrc = os.path.join(ROOT, "shape-attentive-unet_amd", "csrc")
    bad = []
    for f in sorted(os.listdir(csrc)):
        if not f.endswith(".hip"):
            continue
        for i, line in enumerate(open(os.path.join(csrc, f)), 1):
            code = line.split("//")[0]
            if re.search(r"\b(getenv|hipMalloc|hipFree|hipDeviceSynchronize|hipStreamSynchronize|hipStreamIsCapturing)\b", code):
                bad.append("%s:%d: %s" % (f, i, line.strip()))
    assert not bad, bad
    common = open(os.path.join(csrc, "common.h")).read()
    head, _, rest = common.partition("#ifdef SAUNET_AB_SWITCHES")
    assert "getenv" not in head and "getenv" not in rest.partition("#else")[2], "getenv outside the SAUNET_AB_SWITCHES branch of common.h"
    from saunet_amd import _build
    assert not any("SAUNET_AB_SWITCHES" in f for f in _build.FLAGS)


def test_forward_workspace_query(lib):
    """saunet_conv2d_forward_workspace: only the 8 x 8 `center` geometry (bf16, Cin % 64 == 0) asks for split-K scratch; bytes = items x splits x 64 KB."""
    import ctypes as C
    handle = lib.load()

    def desc(n, h, cin, cout, dtype=1):
        d = lib.ConvDesc()
        d.dtype, d.N, d.H, d.W, d.Cin, d.ldx = dtype, n, h, h, cin, cin
        d.Ho, d.Wo, d.Cout, d.ldy, d.KH, d.KW, d.stride, d.pad = h, h, cout, cout, 3, 3, 1, 1
        return d
    assert handle.saunet_conv2d_forward_workspace(C.byref(desc(32, 8, 1024, 512))) == 8 * 8 * 4 * 32 * 512 * 4      # 64 items x 4 splits
    assert handle.saunet_conv2d_forward_workspace(C.byref(desc(32, 8, 512, 1024))) == 8 * 16 * 2 * 32 * 512 * 4     # its data gradient
    assert handle.saunet_conv2d_forward_workspace(C.byref(desc(32, 16, 1536, 512))) == 0
    assert handle.saunet_conv2d_forward_workspace(C.byref(desc(32, 8, 1024, 512, dtype=0))) == 0
    assert handle.saunet_conv2d_forward_workspace(C.byref(desc(2, 8, 1024, 512))) == 0                                # N % 4 != 0: not cell mode


def test_dense_layer_pair_queries_and_argument_checks(lib):
    """saunet_dense_layer_backward_pair_supported is host logic (no device call): 1 only where the LDS-staged kernels run with at least 192
    tiles of 128 pixels and a lower layer of at least 64 channels (SAUNET_BAD_SHAPE = -1 otherwise); saunet_dense_layer_backward_conv1_pair / _conv1 reject what is not a pair of
    consecutive layers of one block, or a bad channel window, with a status code and a message -- before any launch."""
    import ctypes as C
    handle = lib.load()

    def desc(n, h, w, cin, ctot, base=0x10000):
        d = lib.DenseLayerBwd()
        d.N, d.H, d.W, d.Cin, d.Ctot, d.c_begin = n, h, w, cin, ctot, 0
        for i, f in enumerate(("buf", "dbuf", "xhat", "ab", "z1", "g", "dz1", "dz2", "w2_dgrad", "w1_dgrad", "p1", "p2", "sums2", "sums1", "dgamma2", "dbeta2")):
            setattr(d, f, base + 0x1000 * i)             # never dereferenced: every call below fails its argument checks first
        d.ld_xhat, d.ab_replicas, d.ab_rstride, d.count = ctot, 16, 2 * ctot, float(n * h * w)
        d.sums2_replicas, d.sums2_rstride, d.sums1_replicas, d.sums1_rstride = 16, 256, 16, 2 * cin
        return d

    q = handle.saunet_dense_layer_backward_pair_supported
    assert q(C.byref(desc(32, 32, 32, 640, 1024))) == 1          # block 3 of the bench geometry: 256 tiles
    assert q(C.byref(desc(32, 64, 64, 320, 512))) == 1           # block 2: 1024 tiles, the largest map the LDS-staged kernel takes
    assert q(C.byref(desc(32, 128, 128, 160, 256))) == 0         # block 1: the transposed kernel's range
    assert q(C.byref(desc(32, 16, 16, 768, 1024))) == 0          # block 4: 64 tiles -- the steps are split over workgroups instead
    assert q(C.byref(desc(6, 64, 64, 96, 256))) == 1 and q(C.byref(desc(6, 64, 64, 64, 256))) == 0     # the lower layer needs one 64-channel step
    assert q(None) == 0
    hi, lo = desc(32, 32, 32, 640, 1024), desc(32, 32, 32, 608, 1024)
    bad = desc(32, 32, 32, 576, 1024)                              # not the layer below hi
    assert handle.saunet_dense_layer_backward_conv1_pair(C.byref(hi), C.byref(bad), None) == -1
    assert b"consecutive layers" in handle.saunet_last_error()
    other = desc(32, 32, 32, 608, 1024, base=0x900000)             # right size, another block's buffers
    assert handle.saunet_dense_layer_backward_conv1_pair(C.byref(hi), C.byref(other), None) == -1
    small_hi, small_lo = desc(32, 16, 16, 640, 1024), desc(32, 16, 16, 608, 1024)
    assert handle.saunet_dense_layer_backward_conv1_pair(C.byref(small_hi), C.byref(small_lo), None) == -1      # unsupported geometry: refused, not approximated
    hi.c_begin = 100                                               # windows start on 32-channel chunks
    assert handle.saunet_dense_layer_backward_conv1(C.byref(hi), None) == -1
    assert b"channel window" in handle.saunet_last_error()
    del lo


def test_ctypes_structs_match_the_header_layout(tmp_path):
    """the host mirror's ctypes structures against the C compiler's view of include/saunet_hip.h (sizes and the offsets of the last fields): a field
    added on one side only would shift every later pointer silently"""
    import ctypes as C
    import subprocess
    from saunet_amd import lib as L
    src = tmp_path / "layout.c"
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "saunet_hip.h"\n'
                   'int main(void) {\n'
                   '  printf("%zu %zu %zu\\n", sizeof(saunet_conv_desc), offsetof(saunet_conv_desc, workspace), offsetof(saunet_conv_desc, workspace_bytes));\n'
                   '  printf("%zu %zu %zu\\n", sizeof(saunet_dense_layer_bwd), offsetof(saunet_dense_layer_bwd, count), offsetof(saunet_dense_layer_bwd, dbeta2));\n'
                   '  printf("%zu %zu %zu\\n", sizeof(saunet_dense_bn1_list), offsetof(saunet_dense_bn1_list, cin), offsetof(saunet_dense_bn1_list, dbeta));\n'
                   '  printf("%zu %zu %zu\\n", sizeof(saunet_bn_epilogue), offsetof(saunet_bn_epilogue, sums), offsetof(saunet_bn_epilogue, sums_rstride));\n'
                   '  printf("%zu %zu %zu\\n", sizeof(saunet_bn_prologue), offsetof(saunet_bn_prologue, params), offsetof(saunet_bn_prologue, running_var));\n'
                   '  return 0; }\n')
    exe = tmp_path / "layout"
    subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)], check=True)
    rows = [tuple(int(v) for v in l.split()) for l in subprocess.run([str(exe)], capture_output=True, text=True, check=True).stdout.splitlines()]
    want = [(C.sizeof(L.ConvDesc), L.ConvDesc.workspace.offset, L.ConvDesc.workspace_bytes.offset),
            (C.sizeof(L.DenseLayerBwd), L.DenseLayerBwd.count.offset, L.DenseLayerBwd.dbeta2.offset),
            (C.sizeof(L.DenseBn1List), L.DenseBn1List.cin.offset, L.DenseBn1List.dbeta.offset),
            (C.sizeof(L.BnEpilogue), L.BnEpilogue.sums.offset, L.BnEpilogue.sums_rstride.offset),
            (C.sizeof(L.BnPrologue), L.BnPrologue.params.offset, L.BnPrologue.running_var.offset)]
    assert rows == want, (rows, want)


def test_bench_reads_the_kernel_ranking_of_the_committed_profile():
    """bench.py's roofline takes its kernels from the committed rocprofv3 table of the round: the parser must return the [families] first row and the
    top symbols of the [symbols] section, with names in the form the library's launch log uses"""
    import importlib.util
    import sys
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    argv, sys.argv = sys.argv, ["bench.py"]
    try:
        b = importlib.util.module_from_spec(spec); spec.loader.exec_module(b)
    finally:
        sys.argv = argv
    path = os.path.join(ROOT, b.STATS_FILE)
    assert os.path.exists(path), "profiles/ lacks this round's kernel table: %s" % b.STATS_FILE
    syms, fams = b.read_kernel_ranking(path)
    first_family_row = next(l for l in open(path).read().split("# [families]")[1].splitlines()[2:] if l.strip())
    assert fams[0][0] in first_family_row and fams[0][0].endswith("_kernel")
    assert all("_kernel" in s[0] and "(" not in s[0] and "saunet::" not in s[0] for s in syms[:10])
    assert syms[0][1] >= syms[1][1] >= syms[2][1] > 0
    assert b._main_symbol("bn_bwd_correct_ab_kernel<unsigned short, 8>+dense_dgrad3_kernel<true, false>")[0] == "dense_dgrad3_kernel<true, false>"
    assert b._norm_symbol("void saunet::conv_igemm_fwd_kernel<unsigned short, 64, 64, 32, 32, 8, false>(saunet::IgemmArgs)") == \
        "conv_igemm_fwd_kernel<unsigned short, 64, 64, 32, 32, 8, false>"
    assert b._norm_symbol("wgrad_reduce_multi") == "wgrad_reduce_multi_kernel"
