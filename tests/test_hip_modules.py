"""GPU parity of the drop-in nn.Modules against the fixtures the REAL reference produced
(tests/golden/modules_*.npz) -- float32 storage, tolerance 1e-4 of each tensor's max (north_star: 1e-3)."""
import numpy as np
import pytest
import torch

from tests.golden_util import load, rnd, module_state, close

pytestmark = pytest.mark.gpu


def build(name):
    import saunet_amd as S
    return {
        "SEModule": lambda: S.SEModule(32, 16),
        "SpatialAttentionBlock": lambda: S.SpatialAttentionBlock(32, 8, 2),
        "DualAttBlock": lambda: S.DualAttBlock(inchannels=[32, 48], outchannels=32),
        "GatedSpatialConv2d": lambda: S.GatedSpatialConv2d(16, 16),
        "BasicBlock": lambda: S.BasicBlock(16, 16),
        "DecoderBlock": lambda: S.DecoderBlock(32, 24, 16, True),
        "conv3x3_bn_relu": lambda: S.conv3x3_bn_relu(24, 16),
        # models/models.py:215-220, the Upsample branch (round 6): fixture from oracle/make_golden_r6.py
        "DecoderBlockUpsample": lambda: S.DecoderBlock(32, 24, 16, False),
    }[name]()


CALL = {"DualAttBlock": lambda m, xs: m([xs[0], xs[1]])}
NAMES = ["SEModule", "SpatialAttentionBlock", "DualAttBlock", "GatedSpatialConv2d", "BasicBlock", "DecoderBlock", "conv3x3_bn_relu", "DecoderBlockUpsample"]


@pytest.mark.parametrize("name", NAMES)
def test_module_matches_reference_fixture(name):
    gold = load("modules_%s.npz" % name)
    seed = int(gold["meta.seed"])
    sd = module_state(gold, name, seed)
    mod = build(name).cuda()
    res = mod.load_state_dict(sd, strict=False)
    assert not res.unexpected_keys, res.unexpected_keys
    assert all(("num_batches_tracked" in k) or ("_tmp_running" in k) or ("_running_iter" in k) for k in res.missing_keys), res.missing_keys
    nin = sum(1 for k in gold if k.startswith("meta.shape"))
    xs = [rnd(tuple(gold["meta.shape%d" % i]), seed, "%s.in%d" % (name, i)).cuda().requires_grad_(True) for i in range(nin)]
    call = CALL.get(name, lambda m, xs: m(*xs))
    mod.train()
    y = call(mod, xs)
    ys = list(y) if isinstance(y, (tuple, list)) else [y]
    cots = [rnd(tuple(t.shape), seed, "%s.cot%d" % (name, i)).cuda() for i, t in enumerate(ys)]
    torch.autograd.backward(ys, cots)
    tol = 1e-4
    for i, t in enumerate(ys):
        ok, err, sc = close(t.detach().float().cpu().numpy(), gold["train.out%d" % i], tol, 1e-6)
        assert ok, "%s out%d err %.3g scale %.3g" % (name, i, err, sc)
    for i, x in enumerate(xs):
        ok, err, sc = close(x.grad.float().cpu().numpy(), gold["train.dx%d" % i], tol, 1e-6)
        assert ok, "%s dx%d err %.3g scale %.3g" % (name, i, err, sc)
    gscale = max(float(np.abs(gold[k]).max()) for k in gold if k.startswith("train.grad."))
    for k, p in mod.named_parameters():
        g = p.grad.float().cpu().numpy() if p.grad is not None else np.zeros(tuple(p.shape), np.float32)
        ok, err, sc = close(g, gold["train.grad." + k], tol, 1e-5 * gscale)
        assert ok, "%s grad %s err %.3g scale %.3g" % (name, k, err, sc)
    for k, b in mod.named_buffers():
        if ("train.buf." + k) in gold:
            ok, err, sc = close(b.float().cpu().numpy(), gold["train.buf." + k], tol, 1e-6)
            assert ok, "%s buffer %s err %.3g" % (name, k, err)
    # eval mode (running statistics) from the ORIGINAL state
    mod.load_state_dict(sd, strict=False)
    mod.eval()
    with torch.no_grad():
        y = call(mod, [x.detach() for x in xs])
    ys = list(y) if isinstance(y, (tuple, list)) else [y]
    for i, t in enumerate(ys):
        ok, err, sc = close(t.float().cpu().numpy(), gold["eval.out%d" % i], tol, 1e-6)
        assert ok, "%s eval out%d err %.3g scale %.3g" % (name, i, err, sc)


def test_fused_basic_block_matches_the_unfused_path():
    """functional._BasicBlock (bn1 + ReLU in conv2's operand load, opt-in) against the two conv_bn_act calls it replaces: output, input
    gradient, parameter gradients and running statistics, float32 storage."""
    import saunet_amd as S
    HF = S.functional
    S.set_compute_dtype(torch.float32)
    torch.manual_seed(11)
    blk = S.BasicBlock(32, 32).cuda().train()
    with torch.no_grad():
        for m in blk.modules():
            if hasattr(m, "running_mean") and m.weight is not None and m.weight.dim() == 1:
                m.weight.uniform_(0.5, 1.5); m.bias.uniform_(-0.3, 0.3)
    state0 = {k: v.clone() for k, v in blk.state_dict().items()}
    x0 = torch.randn(2, 32, 32, 48, device="cuda").contiguous(memory_format=torch.channels_last)
    cot = torch.randn(2, 32, 32, 48, device="cuda")
    res = {}
    saved = HF.FUSED_BASIC_BLOCK
    try:
        for mode in (True, False):
            HF.FUSED_BASIC_BLOCK = mode
            blk.load_state_dict(state0); HF.notify_params_changed(); blk.zero_grad(set_to_none=True)
            x = x0.clone().requires_grad_(True)
            y = blk(x)
            (y * cot).sum().backward()
            torch.cuda.synchronize()
            res[mode] = (y.detach().clone(), x.grad.clone(), {k: v.grad.clone() for k, v in blk.named_parameters()},
                         {k: v.clone() for k, v in blk.state_dict().items() if "running" in k})
    finally:
        HF.FUSED_BASIC_BLOCK = saved
    (ya, dxa, ga, ra), (yb, dxb, gb, rb) = res[True], res[False]
    def rel(a, b): return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))
    assert rel(ya, yb) < 1e-4 and rel(dxa, dxb) < 2e-4
    for k in ga:
        assert rel(ga[k], gb[k]) < 2e-4, k
    for k in ra:
        assert rel(ra[k], rb[k]) < 1e-5, k


@pytest.mark.gpu
@pytest.mark.parametrize("ch,shape", [(16, (2, 32, 48)), (64, (1, 16, 16)), (32, (3, 16, 32))])
def test_basic_block_relu_bit_mask_is_bit_identical(ch, shape):
    """bf16 BasicBlock with the ReLU decisions of its output kept as bits (saunet_affine_act_mask + saunet_bn_backward_*_masked, round 6)
    against the same block re-reading the skip tensor in bn2's backward (SAUNET_RELU_MASK=0 path): the arithmetic is the same, so output,
    input gradient and every parameter gradient must be IDENTICAL; the mask bytes themselves are checked against out > 0."""
    import saunet_amd as S
    HF = S.functional
    S.set_compute_dtype(torch.bfloat16)
    try:
        torch.manual_seed(ch)
        blk = S.BasicBlock(ch, ch).cuda().train()
        with torch.no_grad():
            for m in blk.modules():
                if hasattr(m, "running_mean") and m.weight is not None and m.weight.dim() == 1:
                    m.weight.uniform_(0.5, 1.5); m.bias.uniform_(-0.3, 0.3)
        state0 = {k: v.clone() for k, v in blk.state_dict().items()}
        n, h, w = shape
        x0 = torch.randn(n, ch, h, w, device="cuda").to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
        cot = torch.randn(n, ch, h, w, device="cuda").to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
        res, saved, calls = {}, HF.RELU_MASK, []
        orig = S.lib.call

        def traced(name, *a):
            calls.append(name); return orig(name, *a)
        try:
            for mode in (True, False):
                HF.RELU_MASK = mode
                blk.load_state_dict(state0); HF.notify_params_changed(); blk.zero_grad(set_to_none=True)
                x = x0.clone().requires_grad_(True)
                del calls[:]
                S.lib.call = traced
                try:
                    y = blk(x)
                    (y * cot).sum().backward()
                finally:
                    S.lib.call = orig
                torch.cuda.synchronize()
                assert ("saunet_bn_backward_apply_masked" in calls) == mode, calls
                assert ("saunet_affine_act_mask" in calls or "saunet_affine_act_bn" in calls) == (mode or HF.BN_FINALIZE_FUSED), calls
                res[mode] = (y.detach().clone(), x.grad.clone(), {k: v.grad.clone() for k, v in blk.named_parameters()})
        finally:
            HF.RELU_MASK = saved
        (ya, dxa, ga), (yb, dxb, gb) = res[True], res[False]
        assert torch.equal(ya, yb) and torch.equal(dxa, dxb)
        for k in ga:
            assert torch.equal(ga[k], gb[k]), k
        # the mask itself: bit (c % 8) of byte [pixel][c / 8] is out > 0
        z = torch.randn(n, ch, h, w, device="cuda").to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
        r = torch.randn(n, ch, h, w, device="cuda").to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
        sc = torch.rand(ch, device="cuda") + 0.5; sh = torch.randn(ch, device="cuda") * 0.2
        mask = torch.zeros(z.numel() // 8, dtype=torch.uint8, device="cuda")
        out = HF.affine_act(z, sc, sh, True, r, mask=mask)
        pre = torch.addcmul(sh.view(1, -1, 1, 1), z.float(), sc.view(1, -1, 1, 1)) + r.float()      # (fmaf vs mul+add: equal sign except within an ulp of zero)
        want = (pre > 0).permute(0, 2, 3, 1).reshape(-1, ch // 8, 8)
        got = ((mask.view(-1, ch // 8, 1).int() >> torch.arange(8, device="cuda").view(1, 1, 8)) & 1).bool()
        differ = (got != want)
        assert int(differ.sum()) <= 2 and float(pre.permute(0, 2, 3, 1).reshape(-1, ch // 8, 8)[differ].abs().max() if differ.any() else 0.0) < 1e-5
        assert torch.equal(out > 0, (got.reshape(n, h, w, ch).permute(0, 3, 1, 2)) & (out > 0))
    finally:
        S.set_compute_dtype(torch.float32)


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
def test_consumer_side_bn_finalize_matches_the_separate_launch(dtype):
    """round 6: saunet_affine_act_bn / _pool_bn derive the BatchNorm coefficients in the prologue of the pass that applies them.  Against the saunet_bn_finalize launch they replace (SAUNET_BN_FINALIZE_FUSED=0 path): outputs and
    gradients identical (same arithmetic), running statistics to float rounding, and no bn_finalize call left on the fused path."""
    import saunet_amd as S
    HF = S.functional
    S.set_compute_dtype(dtype)
    try:
        torch.manual_seed(21)
        blk = S.BasicBlock(32, 32).cuda().train()
        cbr = S.conv3x3_bn_relu(32, 64).cuda().train()
        with torch.no_grad():
            for mod in (blk, cbr):
                for m in mod.modules():
                    if hasattr(m, "running_mean") and m.weight is not None and m.weight.dim() == 1:
                        m.weight.uniform_(0.5, 1.5); m.bias.uniform_(-0.3, 0.3)
        st_b = {k: v.clone() for k, v in blk.state_dict().items()}; st_c = {k: v.clone() for k, v in cbr.state_dict().items()}
        x0 = torch.randn(2, 32, 32, 48, device="cuda").to(dtype).contiguous(memory_format=torch.channels_last)
        bn = [m for m in cbr.modules() if hasattr(m, "running_mean")][0]
        conv = [m for m in cbr.modules() if isinstance(m, torch.nn.Conv2d)][0]
        res, saved, calls, orig = {}, HF.BN_FINALIZE_FUSED, [], S.lib.call

        def traced(name, *a):
            calls.append(name); return orig(name, *a)
        try:
            for mode in (True, False):
                HF.BN_FINALIZE_FUSED = mode
                blk.load_state_dict(st_b); cbr.load_state_dict(st_c); HF.notify_params_changed()
                blk.zero_grad(set_to_none=True); cbr.zero_grad(set_to_none=True)
                x = x0.clone().requires_grad_(True)
                del calls[:]
                S.lib.call = traced
                try:
                    y = cbr(blk(x))
                    pool = torch.empty(2, 64, dtype=torch.float32, device="cuda")
                    y2 = HF.conv_bn_act(y, conv.weight.new_tensor(conv.weight.detach()[:, :, :1, :1].repeat(1, 2, 1, 1) * 0.1), None, bn, relu=True, pool=pool)
                    (y.float().square().sum() + y2.float().sum()).backward()
                finally:
                    S.lib.call = orig
                torch.cuda.synchronize()
                assert calls.count("saunet_bn_finalize") == (1 if mode else 4), calls          # (fused: BasicBlock.bn1 feeds conv2's operand prologue, kept; the ConvBNReLU's conv has a bias)
                assert ("saunet_affine_act_bn" in calls) == mode and ("saunet_affine_act_pool_bn" in calls) == mode
                res[mode] = (y.detach().clone(), y2.detach().clone(), pool.clone(), x.grad.clone(),
                             {k: v.grad.clone() for mod in (blk, cbr) for k, v in mod.named_parameters()},
                             {k: v.clone() for mod in (blk, cbr) for k, v in mod.state_dict().items() if "running" in k})
        finally:
            HF.BN_FINALIZE_FUSED = saved
        a, b = res[True], res[False]
        assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) and torch.equal(a[3], b[3])
        assert float((a[2] - b[2]).abs().max()) <= 1e-6 * float(b[2].abs().max())            # (float atomics into the pool: order)
        for k in a[4]:
            assert float((a[4][k] - b[4][k]).abs().max()) <= 1e-6 * float(b[4][k].abs().max()) + 1e-12, k
        for k in a[5]:
            assert float((a[5][k] - b[5][k]).abs().max()) <= 1e-6 * float(b[5][k].abs().max()) + 1e-9, k
    finally:
        S.set_compute_dtype(torch.float32)


@pytest.mark.gpu
@pytest.mark.parametrize("ch,ck,shape", [(64, 32, (2, 32, 48)), (16, 8, (1, 32, 32)), (32, 16, (3, 16, 32))])
def test_basic_block_and_its_1x1_consumer_as_one_node(ch, ck, shape):
    """functional._BasicBlockConv (round 6): the data gradient of the 1x1 convolution behind a residual block applies the block's ReLU bits and takes
    bn2's two backward sums in its epilogue (saunet_bn_epilogue.relu_mask), and the masked gradient it writes is the skip branch's gradient.  Against
    the two separate nodes (SAUNET_BLOCK_CONV_FUSED=0): same forward (identical), input / parameter gradients to summation order; the fused path
    must not call saunet_bn_backward_reduce* at all."""
    import saunet_amd as S
    HF = S.functional
    S.set_compute_dtype(torch.bfloat16)
    try:
        torch.manual_seed(ch + ck)
        blk = S.BasicBlock(ch, ch).cuda().train()
        conv = torch.nn.Conv2d(ch, ck, 1).cuda()
        with torch.no_grad():
            for m in blk.modules():
                if hasattr(m, "running_mean") and m.weight is not None and m.weight.dim() == 1:
                    m.weight.uniform_(0.5, 1.5); m.bias.uniform_(-0.3, 0.3)
        state0 = {k: v.clone() for k, v in blk.state_dict().items()}
        n, h, w = shape
        x0 = torch.randn(n, ch, h, w, device="cuda").to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
        cot = torch.randn(n, ck, h, w, device="cuda").to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
        res, saved, calls, orig = {}, HF.BLOCK_CONV_FUSED, [], S.lib.call

        def traced(name, *a):
            calls.append(name); return orig(name, *a)
        try:
            for mode in (True, False):
                HF.BLOCK_CONV_FUSED = mode
                blk.load_state_dict(state0); HF.notify_params_changed(); blk.zero_grad(set_to_none=True); conv.zero_grad(set_to_none=True)
                x = x0.clone().requires_grad_(True)
                del calls[:]
                S.lib.call = traced
                try:
                    y = HF.basic_block_conv1x1(x, blk, conv)
                    (y * cot).sum().backward()
                finally:
                    S.lib.call = orig
                torch.cuda.synchronize()
                assert any(c.startswith("saunet_bn_backward_reduce") for c in calls) == (not mode), calls
                res[mode] = (y.detach().clone(), x.grad.clone(), {k: v.grad.clone() for mod in (blk, conv) for k, v in mod.named_parameters()},
                             {k: v.clone() for k, v in blk.state_dict().items() if "running" in k})
        finally:
            HF.BLOCK_CONV_FUSED = saved
        (ya, dxa, ga, ra), (yb, dxb, gb, rb) = res[True], res[False]
        assert torch.equal(ya, yb)
        def rel(a, b): return float((a.float() - b.float()).abs().max() / b.float().abs().max().clamp_min(1e-30))
        assert rel(dxa, dxb) < 2e-2, rel(dxa, dxb)          # (bf16 tensors: one ulp of the largest element is 4e-3)
        for k in ga:
            assert rel(ga[k], gb[k]) < 2e-3, (k, rel(ga[k], gb[k]))
        for k in ra:
            assert torch.equal(ra[k], rb[k]), k
    finally:
        S.set_compute_dtype(torch.float32)
