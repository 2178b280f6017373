"""Parity of the LDS-DMA staged matrix-core 3x3 convolution (csrc/conv_mm.hip: decoder `c3x3rb` forward and data gradient,
/root/reference/models/attention_blocks.py:215-220, models/models.py:316) against float64 torch on the bf16-rounded operands: output, the
BatchNorm statistics of the un-biased accumulator, bias / ReLU epilogue, channel-slice views on both sides, ragged output-channel tiles,
image-border tiles, both output-channel tile widths, and the dgrad route (flipped packing) through conv_dgrad_raw."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def HF():
    import saunet_amd
    return saunet_amd.functional


def _rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * scale


CASES = [
    # (N, Cin, H, W, Cout, ldx_extra, ldy_extra, bias, relu)
    (2, 128, 16, 16, 64, 0, 0, False, False),      # one tile per image: the whole halo ring is padding; BN = 64
    (1, 256, 32, 48, 128, 0, 0, True, False),      # 2 x 3 tiles: every border / interior combination; BN = 128 (or 64 when few workgroups)
    (2, 192, 32, 32, 72, 64, 64, True, True),      # 3 channel blocks, ragged Cout (72), channel-slice views in and out, fused ReLU
    (1, 512, 16, 16, 512, 0, 0, False, False),     # dec5-like: many n tiles share one halo
    (9, 128, 32, 32, 256, 0, 0, False, False),     # 36 pixel tiles: not a multiple of 8 (plain block order), two n tiles
    (8, 128, 64, 64, 128, 0, 0, True, False),      # 128 x 1 workgroups >= 256? no: 128 -> BN 64 path with 2 n tiles; XCD-grouped order
    (16, 128, 64, 64, 128, 0, 0, False, False),    # 256 workgroups at BN = 128: the wide tile
    # 8 x 8 maps (`center`): 2 x 2 image cells, every quadrant with its own zero border
    (4, 128, 8, 8, 64, 0, 0, False, False),        # one cell, one n tile
    (8, 256, 8, 8, 192, 64, 64, True, True),       # two cells, three n tiles, channel-slice views, bias + ReLU
    (12, 1024, 8, 8, 512, 0, 0, True, False),      # the layer itself (16 channel blocks), 3 cells: plain block order
]


@pytest.mark.parametrize("case", CASES)
def test_conv_mm_forward_matches_float64(case):
    n, cin, h, w, cout, lxe, lye, has_bias, relu = case
    H = HF()
    dt = torch.bfloat16
    xw = _rnd(n, cin + lxe, h, w, seed=1).cuda().to(dt).contiguous(memory_format=torch.channels_last)
    x = xw[:, lxe // 2: lxe // 2 + cin] if lxe else xw
    wt = torch.nn.Parameter((_rnd(cout, cin, 3, 3, seed=2) * (1.5 / (cin * 9) ** 0.5)).cuda())
    bias = _rnd(cout, seed=3).cuda() if has_bias else None
    yw = torch.full((n, cout + lye, h, w), 7.0, device="cuda", dtype=dt).contiguous(memory_format=torch.channels_last)
    y = yw[:, lye // 2: lye // 2 + cout] if lye else yw
    st = torch.zeros(H.STAT_R, 2, cout, dtype=torch.float64, device="cuda")
    H.conv_forward_raw(x, wt, bias, 1, 1, out=y, stats=st, act_relu=relu)
    torch.cuda.synchronize()
    ref0 = F.conv2d(x.double().cpu(), wt.detach().to(dt).double().cpu(), None, padding=1)
    ref = ref0 + (bias.double().cpu().view(1, -1, 1, 1) if has_bias else 0.0)
    if relu:
        ref = ref.clamp_min(0)
    got = y.double().cpu()
    scale = float(ref.abs().max())
    assert float((got - ref).abs().max()) <= 6e-3 * scale, "output: %g of %g" % (float((got - ref).abs().max()), scale)   # bf16 output rounding
    # statistics: sums of the un-biased float32 accumulator, float64 across workgroups
    s = st.sum(0).cpu()
    cnt = n * h * w
    r1, r2 = ref0.sum((0, 2, 3)), (ref0 * ref0).sum((0, 2, 3))
    assert float((s[0] - r1).abs().max()) <= 2e-5 * float(ref0.abs().max()) * cnt
    assert float(((s[1] - r2) / r2).abs().max()) <= 1e-4
    if lye:     # the channels around the slice stay untouched
        assert float((yw[:, :lye // 2].float() - 7).abs().max()) == 0 and float((yw[:, lye // 2 + cout:].float() - 7).abs().max()) == 0


@pytest.mark.parametrize("case", [(2, 256, 32, 32, 128), (4, 512, 16, 16, 192), (2, 128, 16, 48, 128), (8, 256, 8, 8, 128)])     # last: 8 x 8 maps, cell mode
def test_conv_mm_dgrad_matches_float64(case):
    """dx of y = conv3x3(x, w): Cin(dy side) = the forward's Cout >= 128 routes through the same kernel with the flipped packing"""
    n, cin, h, w, cout = case
    H = HF()
    dt = torch.bfloat16
    wt = torch.nn.Parameter((_rnd(cout, cin, 3, 3, seed=5) * (1.5 / (cout * 9) ** 0.5)).cuda())
    dy = _rnd(n, cout, h, w, seed=6).cuda().to(dt).contiguous(memory_format=torch.channels_last)
    dx = H.conv_dgrad_raw(dy, wt, (n, cin, h, w), 1, 1)
    torch.cuda.synchronize()
    xr = torch.zeros(n, cin, h, w, dtype=torch.float64, requires_grad=True)
    F.conv2d(xr, wt.detach().to(dt).double().cpu(), None, padding=1).backward(dy.double().cpu())
    scale = float(xr.grad.abs().max())
    assert float((dx.double().cpu() - xr.grad).abs().max()) <= 6e-3 * scale


@pytest.mark.parametrize("case", [(2, 128, 16, 16, 64), (1, 256, 24, 48, 128), (3, 128, 8, 32, 192), (8, 256, 32, 32, 64), (2, 384, 16, 16, 64)])
def test_wgrad_mm_matches_float64(case):
    """conv3x3_wgrad_mm_kernel (LDS-DMA staged weight gradient of the prologue-free 3x3 convolutions, csrc/conv_tile.hip): every tap of every
    (co, ci), image borders, several pixel groups (the partial-gradient reduce), odd tile counts -- against float64 autograd on the bf16 operands."""
    n, cin, h, w, cout = case
    H = HF()
    dt = torch.bfloat16
    x = _rnd(n, cin, h, w, seed=11).cuda().to(dt).contiguous(memory_format=torch.channels_last)
    dy = _rnd(n, cout, h, w, seed=12).cuda().to(dt).contiguous(memory_format=torch.channels_last)
    wt = torch.nn.Parameter(torch.zeros(cout, cin, 3, 3, device="cuda"))
    H.GRADS.reset()
    dw = H.conv_wgrad_raw(x, dy, wt, 1, 1)
    torch.cuda.synchronize()
    wr = torch.zeros(cout, cin, 3, 3, dtype=torch.float64, requires_grad=True)
    F.conv2d(x.double().cpu(), wr, None, padding=1).backward(dy.double().cpu())
    scale = float(wr.grad.abs().max())
    err = float((dw.double().cpu() - wr.grad).abs().max())
    assert err <= 2e-5 * scale + 1e-6 * (n * h * w) ** 0.5, (err, scale)          # float32 accumulation of exact bf16 products


@pytest.mark.parametrize("case", [(2, 128, 128, 16, 16), (1, 256, 128, 8, 32), (4, 128, 256, 16, 16), (16, 128, 128, 32, 32)])
def test_convt_wgrad_mm_matches_float64(case):
    """convt_wgrad_mm_kernel: dW of ConvTranspose2d(k4 s2 p1) (attention_blocks.py:179-186 `mrf.up`) from x and the stride-2 parity images of dy read in
    place; every (ci, co, kh, kw), all four parities, borders, several pixel groups -- against float64 autograd on the bf16 operands."""
    n, ci, co, h, w = case
    H = HF()
    dt = torch.bfloat16
    x = _rnd(n, ci, h, w, seed=21).cuda().to(dt).contiguous(memory_format=torch.channels_last)
    dy = _rnd(n, co, 2 * h, 2 * w, seed=22).cuda().to(dt).contiguous(memory_format=torch.channels_last)
    wt = torch.nn.Parameter(torch.zeros(ci, co, 4, 4, device="cuda"))
    H.GRADS.reset()
    dw = H.conv_wgrad_raw(x, dy, wt, 2, 1, transposed=True)
    torch.cuda.synchronize()
    wr = torch.zeros(ci, co, 4, 4, dtype=torch.float64, requires_grad=True)
    F.conv_transpose2d(x.double().cpu(), wr, None, stride=2, padding=1).backward(dy.double().cpu())
    scale = float(wr.grad.abs().max())
    err = float((dw.double().cpu() - wr.grad).abs().max())
    assert err <= 2e-5 * scale + 1e-6 * (n * h * w) ** 0.5, (err, scale)


CONVT_CASES = [
    # (N, Cin, Cout, H, W, ldx_extra, ldy_extra, bias, relu)
    (2, 128, 64, 16, 16, 0, 0, False, False),      # one tile per image: every halo pixel outside the image; BN = 64
    (1, 256, 128, 32, 48, 0, 0, True, False),      # 2 x 3 tiles: every border / interior combination, bias
    (2, 192, 72, 32, 32, 64, 64, True, True),      # 3 channel blocks, ragged Cout (72), channel-slice views in and out, fused ReLU
    (9, 128, 256, 16, 16, 0, 0, False, False),     # 9 pixel tiles (plain block order), two n tiles per parity at BN = 128
    (8, 128, 128, 64, 64, 0, 0, True, False),      # `mrf.up` of dec2 in small: 128 tiles x 4 parities, XCD-grouped order
    (4, 512, 512, 16, 16, 0, 0, False, False),     # `mrf.up` of dec4 in small: 8 channel blocks x 4 taps = 32 stages
]


@pytest.mark.parametrize("case", CONVT_CASES)
def test_convt_mm_forward_matches_float64(case):
    """ConvTranspose2d(k4 s2 p1) forward on the LDS-DMA stage pipeline (conv3x3_mm_kernel<.., CONVT>, round 6: `mrf.up`,
    /root/reference/models/attention_blocks.py:179-186): all four output parities, image borders, bias / ReLU epilogue, BatchNorm statistics of
    the un-biased accumulator over ALL parities, channel-slice views on both sides, ragged output-channel tiles -- against float64
    conv_transpose2d on the bf16-rounded operands; and the launch log must name the kernel (no silent implicit-GEMM route)."""
    n, cin, cout, h, w, lxe, lye, has_bias, relu = case
    H = HF()
    dt = torch.bfloat16
    xw = _rnd(n, cin + lxe, h, w, seed=31).cuda().to(dt).contiguous(memory_format=torch.channels_last)
    x = xw[:, lxe // 2: lxe // 2 + cin] if lxe else xw
    wt = torch.nn.Parameter((_rnd(cin, cout, 4, 4, seed=32) * (1.5 / (cin * 4) ** 0.5)).cuda())
    bias = _rnd(cout, seed=33).cuda() if has_bias else None
    yw = torch.full((n, cout + lye, 2 * h, 2 * w), 7.0, device="cuda", dtype=dt).contiguous(memory_format=torch.channels_last)
    y = yw[:, lye // 2: lye // 2 + cout] if lye else yw
    st = torch.zeros(H.STAT_R, 2, cout, dtype=torch.float64, device="cuda")
    H.PACKS.get(wt, H.L.PACK_CONVT_FWD, dt)                 # (lazy packing launches stay out of the log)
    H.L.load().saunet_launch_log()
    H.conv_forward_raw(x, wt, bias, 2, 1, transposed=True, out=y, stats=st, act_relu=relu)
    launched = H.L.load().saunet_launch_log().decode()
    assert "conv3x3_mm" in launched and "true>" in launched, launched
    torch.cuda.synchronize()
    ref0 = F.conv_transpose2d(x.double().cpu(), wt.detach().to(dt).double().cpu(), None, stride=2, padding=1)
    ref = ref0 + (bias.double().cpu().view(1, -1, 1, 1) if has_bias else 0.0)
    if relu:
        ref = ref.clamp_min(0)
    got = y.double().cpu()
    scale = float(ref.abs().max())
    assert float((got - ref).abs().max()) <= 6e-3 * scale, "output: %g of %g" % (float((got - ref).abs().max()), scale)   # bf16 output rounding
    s = st.sum(0).cpu()
    cnt = n * 4 * h * w
    r1, r2 = ref0.sum((0, 2, 3)), (ref0 * ref0).sum((0, 2, 3))
    assert float((s[0] - r1).abs().max()) <= 2e-5 * float(ref0.abs().max()) * cnt
    assert float(((s[1] - r2) / r2).abs().max()) <= 1e-4
    if lye:     # the channels around the slice stay untouched
        assert float((yw[:, :lye // 2].float() - 7).abs().max()) == 0 and float((yw[:, lye // 2 + cout:].float() - 7).abs().max()) == 0
