"""GPU parity of every C-ABI op (through the autograd layer) against plain PyTorch fp32 on the CPU.

float32 storage: tolerance 1e-4 relative to the tensor's max (north_star asks 1e-3);
bf16 storage: 3e-2 (bf16 has 8 mantissa bits; accumulation stays fp32)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

DTYPES = [torch.float32, torch.bfloat16]
TOL = {torch.float32: 1e-4, torch.bfloat16: 3e-2}


def HF():
    import saunet_amd
    return saunet_amd.functional


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed + sum(shape))
    return torch.randn(*shape, generator=g) * scale


def to_dev(t, dtype):
    t = t.detach().cuda()
    if t.dim() == 4:
        t = t.to(dtype).contiguous(memory_format=torch.channels_last)
    return t


def q(t, dtype):
    """what the device sees after storage rounding (so bf16 tests compare like with like)"""
    return t.detach().to(dtype).float().clone()


def close(a, b, tol, what=""):
    a = a.detach().float().cpu(); b = b.detach().float().cpu()
    assert a.shape == b.shape, (what, a.shape, b.shape)
    scale = max(float(b.abs().max()), 1e-6)
    err = float((a - b).abs().max())
    assert err <= tol * scale + 1e-7, "%s: max err %.3g vs scale %.3g (tol %.1g)" % (what, err, scale, tol)


CONV_CASES = [
    # >= 256 pixel tiles: the persistent resident-weight 3x3 kernel (conv3x3_res_fwd_kernel) takes these
    (4, 128, 128, 128, 32, 3, 1, 1),   # DenseNet conv2 forward geometry; its dgrad is the BN=128 narrow-K variant
    (4, 64, 128, 128, 64, 3, 1, 1),    # res1
    (1, 16, 256, 256, 16, 3, 1, 1),    # res3
    # (N, Cin, H, W, Cout, k, stride, pad)
    (2, 128, 16, 16, 32, 3, 1, 1),     # DenseNet conv2
    (2, 64, 16, 16, 128, 1, 1, 0),     # DenseNet conv1
    (2, 96, 8, 8, 128, 1, 1, 0),       # ragged K-step (96 = 64 + 32)
    (1, 64, 32, 32, 64, 3, 1, 1),      # res1
    (2, 16, 16, 16, 16, 3, 1, 1),      # res3 (narrow rows)
    (2, 64, 16, 16, 48, 3, 1, 1),      # dec1.block.0 (Cout 48)
    (2, 8, 32, 32, 64, 7, 2, 3),       # conv0 on the 8-channel padded image
    (3, 256, 8, 8, 256, 3, 1, 1),      # decoder-like, BN=128 tiles, ragged M (192 rows)
    (2, 33, 8, 8, 33, 1, 1, 0),        # gate inner conv (pointwise path)
    (2, 256, 8, 8, 1, 1, 1, 0),        # c3 (pointwise, Cout 1)
    (2, 1, 16, 16, 32, 1, 1, 0),       # expand (pointwise, Cin 1)
    (2, 32, 16, 16, 4, 1, 1, 0),       # final
    (2, 16, 16, 16, 8, 1, 1, 0),       # d3
    # >= 4096 pixels, small channels: the matrix-core pointwise weight gradient (pointwise_wgrad_mma_kernel)
    (2, 32, 48, 48, 16, 1, 1, 0),      # d2
    (1, 64, 64, 72, 32, 1, 1, 0),      # two input tiles
    (2, 32, 64, 40, 4, 1, 1, 0),       # final (scalar dy rows)
    (1, 8, 72, 64, 1, 1, 1, 0),        # fuse
    (3, 1024, 8, 8, 1, 1, 1, 0),       # c5: many inputs, one output, few pixels (pointwise_wgrad_fewout_kernel)
    (4, 64, 8, 8, 32, 3, 1, 1),        # `center`-like 8x8 map: weight gradient through im2col + pointwise rows
    (4, 64, 8, 8, 128, 1, 1, 0),       # pointwise on an 8x8 map: rows regrouped into 16x16 tiles for the tiled wgrad
]


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("case", CONV_CASES)
def test_conv2d_fwd_bwd(case, dtype):
    n, ci, h, w, co, k, s, p = case
    hf = HF()
    x, wt, b = rnd(n, ci, h, w), rnd(co, ci, k, k, scale=(2.0 / (ci * k * k)) ** 0.5), rnd(co, scale=0.1)
    xr = q(x, dtype).requires_grad_(True); wr = wt.clone().requires_grad_(True); br = b.clone().requires_grad_(True)
    wq = q(wt, dtype)
    yr = F.conv2d(xr, wq + (wr - wr.detach()), br, s, p)
    cot = rnd(*yr.shape, seed=5)
    yr.backward(q(cot, dtype))
    xd = to_dev(x, dtype).requires_grad_(s == 1); wd = wt.cuda().requires_grad_(True); bd = b.cuda().requires_grad_(True)
    yd = hf.conv2d(xd, wd, bd, s, p)
    yd.backward(to_dev(cot, dtype))
    tol = TOL[dtype]
    close(yd, yr, tol, "y")
    close(wd.grad, wr.grad, tol, "dw")
    close(bd.grad, br.grad, tol, "db")
    if s == 1:
        close(xd.grad, xr.grad, tol, "dx")


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("case", [(2, 64, 8, 8, 64), (1, 48, 16, 16, 32), (2, 128, 4, 4, 128),
                                  (2, 128, 32, 16, 72), (1, 256, 16, 32, 256)])      # maps that are multiples of 16 (bf16): the direct parity weight gradient
def test_conv_transpose_fwd_bwd(case, dtype):
    n, ci, h, w, co = case
    hf = HF()
    x, wt, b = rnd(n, ci, h, w), rnd(ci, co, 4, 4, scale=(2.0 / (ci * 4)) ** 0.5), rnd(co, scale=0.1)
    xr = q(x, dtype).requires_grad_(True); wr = wt.clone().requires_grad_(True); br = b.clone().requires_grad_(True)
    yr = F.conv_transpose2d(xr, q(wt, dtype) + (wr - wr.detach()), br, stride=2, padding=1)
    cot = rnd(*yr.shape, seed=7)
    yr.backward(q(cot, dtype))
    xd = to_dev(x, dtype).requires_grad_(True); wd = wt.cuda().requires_grad_(True); bd = b.cuda().requires_grad_(True)
    yd = hf.conv_transpose2d(xd, wd, bd)
    yd.backward(to_dev(cot, dtype))
    tol = TOL[dtype]
    close(yd, yr, tol, "y"); close(xd.grad, xr.grad, tol, "dx"); close(wd.grad, wr.grad, tol, "dw"); close(bd.grad, br.grad, tol, "db")


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("geom", [(2, 16, 16, 96), (8, 32, 32, 128), (3, 48, 64, 96), (5, 32, 48, 64)])
def test_conv_prologue_stats_and_channel_slices(dtype, geom):
    """DenseNet pattern: read a channel slice, BN+ReLU in the operand load, write into another slice with stats.
    >= 32 pixel tiles take the persistent resident-weight kernel; in bf16 with Cin in {64, 96, 128} its double-buffered variant
    (odd unit counts, ragged per-workgroup tile shares, 2-4 channel blocks)."""
    hf = HF()
    n, h, w, cin = geom
    ctot, co = 160, 32
    buf = rnd(n, ctot, h, w)
    wt = rnd(co, cin, 3, 3, scale=0.05)
    scale, shift = rnd(cin, seed=3).abs() + 0.5, rnd(cin, seed=4) * 0.3
    a = F.relu(q(buf, dtype)[:, :cin] * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1))
    if dtype == torch.bfloat16:
        a = q(a, dtype)
    yr = F.conv2d(a, q(wt, dtype), None, 1, 1)
    bd = to_dev(buf, dtype)
    stats_r = torch.zeros(hf.STAT_R, 2, ctot, dtype=torch.float64, device="cuda")      # replicated accumulators
    hf.conv_forward_raw(bd[:, :cin], wt.cuda(), None, 1, 1, pro=(scale.cuda(), shift.cuda(), True), out=bd[:, 128:160],
                        stats=stats_r[:, :, 128:160])
    stats = stats_r.sum(0)
    tol = TOL[dtype]
    close(bd[:, 128:160], yr, tol, "slice out")
    close(bd[:, :128], q(buf, dtype)[:, :128], 0, "untouched channels")
    close(stats[0, 128:160], yr.double().sum((0, 2, 3)), max(tol, 1e-4), "sum")
    close(stats[1, 128:160], (yr.double() ** 2).sum((0, 2, 3)), max(tol, 1e-4) * 2, "sumsq")
    assert float(stats[:, :128].abs().max()) == 0
    # wgrad with the same prologue
    dy = rnd(n, co, h, w, seed=9)
    dw = hf.conv_wgrad_raw(bd[:, :cin], to_dev(dy, dtype), wt.cuda(), 1, 1, pro=(scale.cuda(), shift.cuda(), True))
    ar = a.clone().requires_grad_(False)
    wr = wt.clone().requires_grad_(True)
    F.conv2d(ar, wr, None, 1, 1).backward(q(dy, dtype))
    close(dw, wr.grad, tol, "dw with prologue")


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("relu,res,transposed", [(True, False, False), (True, True, False), (False, False, False), (True, False, True)])
def test_conv_bn_act(dtype, relu, res, transposed):
    import torch.nn as nn
    hf = HF()
    n, ci, h, w, co = 4, 32, 8, 8, 32
    x = rnd(n, ci, h, w)
    wt = rnd(ci, co, 4, 4, scale=0.1) if transposed else rnd(co, ci, 3, 3, scale=0.1)
    b = rnd(co, scale=0.1)
    bn_r, bn_d = nn.BatchNorm2d(co), nn.BatchNorm2d(co).cuda()
    with torch.no_grad():
        bn_r.weight.copy_(rnd(co, seed=11).abs() + 0.5); bn_r.bias.copy_(rnd(co, seed=12) * 0.2)
        bn_d.weight.copy_(bn_r.weight); bn_d.bias.copy_(bn_r.bias)
    xr = q(x, dtype).requires_grad_(True); wr = wt.clone().requires_grad_(True); br = b.clone().requires_grad_(True)
    wq = q(wt, dtype) + (wr - wr.detach())
    z = F.conv_transpose2d(xr, wq, br, stride=2, padding=1) if transposed else F.conv2d(xr, wq, br, 1, 1)
    z = z + (q(z, dtype) - z).detach()   # the device stores the conv output in `dtype` (straight-through for the gradient)
    y = bn_r(z)
    if res:
        y = y + xr
    if relu:
        y = F.relu(y)
    cot = rnd(*y.shape, seed=13)
    y.backward(q(cot, dtype))
    xd = to_dev(x, dtype).requires_grad_(True); wd = wt.cuda().requires_grad_(True); bd = b.cuda().requires_grad_(True)
    yd = hf.conv_bn_act(xd, wd, bd, bn_d, relu=relu, residual=xd if res else None, padding=1, transposed=transposed)
    yd.backward(to_dev(cot, dtype))
    tol = TOL[dtype] * (3 if dtype == torch.bfloat16 else 1)
    close(yd, y, tol, "y"); close(xd.grad, xr.grad, tol, "dx"); close(wd.grad, wr.grad, tol, "dw")
    close(bn_d.weight.grad, bn_r.weight.grad, tol, "dgamma"); close(bn_d.bias.grad, bn_r.bias.grad, tol, "dbeta")
    close(bn_d.running_mean, bn_r.running_mean, tol, "running_mean"); close(bn_d.running_var, bn_r.running_var, tol, "running_var")
    assert int(bn_d.num_batches_tracked) == 1
    # eval mode uses the running statistics
    bn_r.eval(); bn_d.eval()
    with torch.no_grad():
        z = F.conv_transpose2d(q(x, dtype), q(wt, dtype), b, stride=2, padding=1) if transposed else F.conv2d(q(x, dtype), q(wt, dtype), b, 1, 1)
        ye = bn_r(z) + (q(x, dtype) if res else 0)
        ye = F.relu(ye) if relu else ye
        yde = hf.conv_bn_act(to_dev(x, dtype), wd, bd, bn_d, relu=relu, residual=to_dev(x, dtype) if res else None, padding=1, transposed=transposed)
    close(yde, ye, tol, "eval y")


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("c", [33, 64, 1])
def test_standalone_bn(dtype, c):
    import torch.nn as nn
    hf = HF()
    x = rnd(3, c, 8, 8) * 2 + 0.5
    bn_r, bn_d = nn.BatchNorm2d(c), nn.BatchNorm2d(c).cuda()
    xr = q(x, dtype).requires_grad_(True)
    y = bn_r(xr); cot = rnd(*y.shape, seed=2); y.backward(q(cot, dtype))
    xd = to_dev(x, dtype).requires_grad_(True)
    yd = hf.batch_norm_act(xd, bn_d); yd.backward(to_dev(cot, dtype))
    tol = TOL[dtype]
    close(yd, y, tol, "y"); close(xd.grad, xr.grad, tol * 2, "dx"); close(bn_d.weight.grad, bn_r.weight.grad, tol * 2, "dgamma")
    close(bn_d.running_var, bn_r.running_var, tol, "rv")


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("shape,size", [((2, 64, 8, 8), (32, 32)), ((2, 1, 4, 4), (64, 64)), ((1, 16, 16, 16), (32, 32)),
                                        ((2, 1, 32, 32), (32, 32)), ((1, 8, 6, 10), (24, 20))])
def test_bilinear(dtype, shape, size):
    hf = HF()
    x = rnd(*shape)
    xr = q(x, dtype).requires_grad_(True)
    y = F.interpolate(xr, size=size, mode="bilinear", align_corners=True)
    cot = rnd(*y.shape, seed=4); y.backward(q(cot, dtype))
    xd = to_dev(x, dtype).requires_grad_(True)
    yd = hf.interpolate_bilinear(xd, size=size); yd.backward(to_dev(cot, dtype))
    close(yd, y, TOL[dtype], "y"); close(xd.grad, xr.grad, TOL[dtype], "dx")


@pytest.mark.parametrize("dtype", DTYPES)
def test_pools_sigmoid_relu_cat_gate(dtype):
    hf = HF()
    tol = TOL[dtype]
    x = rnd(2, 24, 8, 8)
    for is_max in (True, False):
        xr = q(x, dtype).requires_grad_(True)
        y = F.max_pool2d(xr, 2, 2) if is_max else F.avg_pool2d(xr, 2, 2)
        cot = rnd(*y.shape, seed=6); y.backward(q(cot, dtype))
        xd = to_dev(x, dtype).requires_grad_(True)
        yd = hf.max_pool2x2(xd) if is_max else hf.avg_pool2x2(xd)
        yd.backward(to_dev(cot, dtype))
        close(yd, y, tol, "pool"); close(xd.grad, xr.grad, tol, "pool dx")
    for fn_r, fn_d in ((torch.sigmoid, hf.sigmoid), (F.relu, hf.relu)):
        xr = q(x, dtype).requires_grad_(True); y = fn_r(xr); cot = rnd(*y.shape, seed=8); y.backward(q(cot, dtype))
        xd = to_dev(x, dtype).requires_grad_(True); yd = fn_d(xd); yd.backward(to_dev(cot, dtype))
        close(yd, y, tol, "act"); close(xd.grad, xr.grad, tol, "act dx")
    g = rnd(2, 1, 8, 8, seed=3)
    xr = q(x, dtype).requires_grad_(True); gr = q(g, dtype).requires_grad_(True)
    y = torch.cat([xr, gr], 1) ; y2 = xr * (gr + 1)
    cot, cot2 = rnd(*y.shape, seed=9), rnd(*y2.shape, seed=10)
    (y * q(cot, dtype)).sum().backward(retain_graph=True); gx1, gg1 = xr.grad.clone(), gr.grad.clone(); xr.grad = None; gr.grad = None
    y2.backward(q(cot2, dtype)); gx2, gg2 = xr.grad, gr.grad
    xd = to_dev(x, dtype).requires_grad_(True); gd = to_dev(g, dtype).requires_grad_(True)
    yd = hf.cat([xd, gd]); yd.backward(to_dev(cot, dtype))
    close(yd, y, tol, "cat"); close(xd.grad, gx1, tol, "cat dx"); close(gd.grad, gg1, tol, "cat dg")
    xd.grad = None; gd.grad = None
    yd2 = hf.gate_mul(xd, gd); yd2.backward(to_dev(cot2, dtype))
    close(yd2, y2, tol, "gate"); close(xd.grad, gx2, tol, "gate dx"); close(gd.grad, gg2, tol * 2, "gate dalpha")


@pytest.mark.parametrize("shape", [(2, 64, 8, 8), (1, 512, 4, 4), (2, 128, 6, 10), (3, 256, 5, 7)])   # bf16: 8 / 64 / 16 / 32 chunks per pixel, ragged pixel ranges
@pytest.mark.parametrize("dtype", DTYPES)
def test_dual_att_tail(dtype, shape):
    import torch.nn as nn
    hf = HF()
    (n, c, h, w), r = shape, 16
    Fm, S = rnd(n, c, h, w), torch.sigmoid(rnd(n, 1, h, w, seed=2))
    fc1_r, fc2_r = nn.Conv2d(c, c // r, 1), nn.Conv2d(c // r, c, 1)
    fc1_d, fc2_d = nn.Conv2d(c, c // r, 1).cuda(), nn.Conv2d(c // r, c, 1).cuda()
    fc1_d.load_state_dict(fc1_r.state_dict()); fc2_d.load_state_dict(fc2_r.state_dict())
    Fr, Sr = q(Fm, dtype).requires_grad_(True), q(S, dtype).requires_grad_(True)
    se = torch.sigmoid(fc2_r(F.relu(fc1_r(F.adaptive_avg_pool2d(Fr, 1)))))
    out = (Sr + 1) * (Fr * se)
    cot = rnd(*out.shape, seed=5); out.backward(q(cot, dtype))
    Fd, Sd = to_dev(Fm, dtype).requires_grad_(True), to_dev(S, dtype).requires_grad_(True)
    od = hf.dual_att_tail(Fd, Sd, fc1_d, fc2_d); od.backward(to_dev(cot, dtype))
    tol = TOL[dtype]
    close(od, out, tol, "out"); close(Fd.grad, Fr.grad, tol, "dF"); close(Sd.grad, Sr.grad, tol * 2, "dS")
    for a, b in ((fc1_d, fc1_r), (fc2_d, fc2_r)):
        close(a.weight.grad, b.weight.grad, tol * 2, "dfc w"); close(a.bias.grad, b.bias.grad, tol * 2, "dfc b")


@pytest.mark.parametrize("dtype", DTYPES)
def test_dual_loss_against_oracle_fixture(dtype):
    from oracle import saunet_ref as R
    from tests.golden_util import load, rnd as grnd
    hf = HF()
    g = load("loss.npz")
    seed = int(g["meta.seed"])
    logits = grnd((3, 4, 16, 16), seed, "loss.logits", 2.0)
    edge = torch.sigmoid(grnd((3, 1, 16, 16), seed, "loss.edge", 2.0))
    seg, edge_t = torch.from_numpy(g["seg"]), torch.from_numpy(g["edge_t"])
    ld, ed = to_dev(logits, dtype).requires_grad_(True), to_dev(edge, dtype).requires_grad_(True)
    loss, metrics = hf.dual_loss(ld, ed, seg.cuda(), edge_t.cuda())
    loss.backward()
    if dtype == torch.float32:
        close(loss, torch.tensor(g["dual"]), 2e-6, "loss vs reference fixture")
        close(ld.grad, torch.from_numpy(g["dlogits"]), 1e-4, "dlogits"); close(ed.grad, torch.from_numpy(g["dedge"]), 1e-4, "dedge")
        close(metrics[0], torch.tensor(g["acc"]), 1e-6, "acc"); close(metrics[1:], torch.from_numpy(g["jac"]), 1e-6, "jaccard")
    else:
        lr, er = q(logits, dtype).requires_grad_(True), q(edge, dtype).requires_grad_(True)
        L = R.dual_loss(lr, er, seg, edge_t); L.backward()
        close(loss, L, 1e-4, "loss"); close(ld.grad, lr.grad, 3e-2, "dlogits"); close(ed.grad, er.grad, 3e-2, "dedge")


def test_pixel_acc_and_jaccard_with_the_reference_signatures():
    """SegmentationModuleBase.pixel_acc(pred, label, num_class) / .jaccard(pred, label) (/root/reference/models/models.py:51-78) as thin device
    implementations: against the reference's own values -- loss.npz (pred = round(softmax(logits)).long(), the train branch's
    call, :92) and metrics.npz (a one-hot prediction with all-zero pixels, i.e. torch.max ties -> class 0; seeded binary masks for jaccard) --
    in every prediction dtype / memory format the kernel takes."""
    import saunet_amd as S
    from tests.golden_util import load, rnd as grnd
    base = S.modules.SegmentationModuleBase()
    g = load("loss.npz")
    logits = grnd((3, 4, 16, 16), int(g["meta.seed"]), "loss.logits", 2.0)
    seg = torch.from_numpy(g["seg"]).cuda()
    pred = torch.round(torch.softmax(logits, 1)).long()
    for p in (pred.cuda(), pred.cuda().float(), pred.cuda().float().contiguous(memory_format=torch.channels_last), pred.cuda().to(torch.uint8),
              pred.cuda().to(torch.bfloat16)):
        acc, jac = base.pixel_acc(p, seg, 4)
        assert acc.dim() == 0 and len(jac) == 3
        close(acc, torch.tensor(g["acc"]), 1e-6, "acc"); close(torch.stack(jac), torch.from_numpy(g["jac"]), 1e-6, "jaccard")
    # raw scores (no rounding): the class is the first maximum
    sc = torch.softmax(logits, 1).cuda()
    acc, jac = base.pixel_acc(sc, seg, 4)
    am = sc.argmax(1); valid = seg >= 1
    close(acc, ((am == seg) & valid).sum().float() / valid.sum().float(), 1e-6, "acc of raw scores")
    m = load("metrics.npz")
    acc, jac = base.pixel_acc(torch.from_numpy(m["pixel_acc.pred"]).cuda(), torch.from_numpy(m["pixel_acc.label"]).cuda(), 4)
    close(acc, torch.tensor(float(m["pixel_acc.acc"])), 1e-6, "acc (ties)"); close(torch.stack(jac), torch.from_numpy(m["pixel_acc.jac"]).float(), 1e-6, "jaccard (ties)")
    jp, jl = torch.from_numpy(m["jaccard.pred"]).cuda(), torch.from_numpy(m["jaccard.label"]).cuda()
    for p in (jp, jp.long(), jp.bool(), jp.to(torch.bfloat16)):
        close(base.jaccard(p, jl), torch.tensor(float(m["jaccard.value"])), 1e-6, "binary jaccard")
    # host-side intersectionAndUnion (models/models.py:24-49): mean Jaccard of classes 1, 2 of two label maps
    a, b = m["pixel_acc.pred"].argmax(1)[0], m["pixel_acc.label"][0]
    j = []
    for c in (1, 2):
        pa, pb = (a == c) & (b >= 0), b == c
        j.append((pa & pb).sum() / float((pa | pb).sum()))
    assert abs(base.intersectionAndUnion(torch.from_numpy(a), torch.from_numpy(b), 4) - (j[0] + j[1]) / 2) < 1e-12
    with pytest.raises(RuntimeError):
        base.pixel_acc(sc[:, :3], seg, 4)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("n,cin,cout,h,w", [(2, 64, 1, 32, 32), (3, 32, 4, 24, 40), (2, 2, 1, 64, 64), (1, 1024, 1, 16, 16), (2, 24, 3, 8, 8)])
def test_few_output_weight_and_bias_gradient_in_one_pass(dtype, n, cin, cout, h, w):
    """saunet_conv2d_wgrad_bias (round 6): dW and dbias of the few-output 1x1 layers (c3 / c4 / c5 / phi / cw / fuse / final,
    /root/reference/models/models.py:286-301,324) from ONE pass -- per-split partials with the bias as a ones-channel, ordered reduce -- against
    float64 autograd, and bit-identical from run to run (no float atomics on this path)."""
    hf = HF()
    g = torch.Generator().manual_seed(cin * 7 + cout)
    x = to_dev(torch.randn(n, cin, h, w, generator=g), dtype)
    dy = to_dev(torch.randn(n, cout, h, w, generator=g), dtype)
    wt = torch.nn.Parameter(torch.zeros(cout, cin, 1, 1, device="cuda"))
    entries, orig = [], hf.L.call
    hf.L.call = lambda name, *a: (entries.append(name), orig(name, *a))[1]
    try:
        hf.GRADS.reset()
        dw, db = hf.conv_wgrad_bias_raw(x, dy, wt, 1, 0)
        dw1, db1 = dw.clone(), db.clone()
        hf.GRADS.reset()
        dw2, db2 = hf.conv_wgrad_bias_raw(x, dy, wt, 1, 0)
    finally:
        hf.L.call = orig
    assert entries.count("saunet_conv2d_wgrad_bias") == 2
    assert torch.equal(dw1, dw2) and torch.equal(db1, db2)                   # deterministic
    xr, dyr = x.double().cpu(), dy.double().cpu()
    ref_w = torch.einsum("nohw,nihw->oi", dyr, xr).view(cout, cin, 1, 1)
    ref_b = dyr.sum((0, 2, 3))
    close(dw1, ref_w, 1e-5, "dW"); close(db1, ref_b, 1e-5, "dbias")


def test_canny_bit_exact_with_oracle():
    from oracle import canny as oc, weights as Wt
    hf = HF()
    img, _, _ = Wt.synthetic_batch(3, 64, 96, seed=17)
    r = np.random.default_rng(0)
    noisy = torch.from_numpy((r.standard_normal((2, 1, 64, 96)) * 3).astype(np.float32)).repeat(1, 3, 1, 1)
    big = torch.from_numpy((r.standard_normal((1, 1, 400, 400)) * 2).astype(np.float32)).repeat(1, 3, 1, 1)   # > 152 KiB map: global-memory sweeps
    for x in (img, noisy, torch.zeros(1, 3, 32, 32), big):
        want = oc.canny_batch(x.numpy())
        got = hf.canny(x.cuda(), 10, 100, dtype=torch.float32).cpu().numpy()
        assert got.shape == want.shape
        assert (got == want).all(), "canny mismatch: %d pixels differ" % int((got != want).sum())
        got16 = hf.canny(x.cuda(), 10, 100, dtype=torch.bfloat16).float().cpu().numpy()
        assert (got16 == want).all()


def test_mask_to_edges_on_device_matches_loader_and_reference_fixture():
    from oracle import saunet_ref as R, weights as Wt
    from tests.golden_util import load
    hf = HF()
    g = load("loss.npz")
    m = torch.from_numpy(g["m2e_mask"])[None]
    assert (hf.mask_to_edges(m.cuda()).cpu().numpy()[0] == g["m2e_edge"]).all()          # fixture from the REAL reference
    _, seg, edge = Wt.synthetic_batch(3, 96, 80, seed=9)
    assert (hf.mask_to_edges(seg.cuda()).cpu() == edge).all()
    r = np.random.default_rng(3)
    rnd_seg = torch.from_numpy(r.integers(0, 4, (2, 40, 56)))
    want = np.stack([R.mask_to_edges(s.numpy()) for s in rnd_seg])
    assert (hf.mask_to_edges(rnd_seg.cuda()).cpu().numpy() == want).all()


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("pool_type", ["avg", "max", "avgmax", "avgmaxc"])
@pytest.mark.parametrize("shape", [(2, 16, 8, 8), (3, 24, 16, 48), (2, 7, 5, 9)])
def test_adaptive_avgmax_pool_kat(dtype, pool_type, shape):
    """models/adaptive_avgmax_pool.py:19-40 restated with torch CPU ops (the reference file is never imported by its own training path;
    SURVEY 8 A16): forward values and the gradient, including max ties (first occurrence wins, like F.max_pool2d)."""
    import torch.nn.functional as F
    import saunet_amd as S
    torch.manual_seed(sum(shape))
    x = torch.randn(*shape).to(dtype).float()
    x[0, 0, 0, 0] = x[0, 0].max() + 1.0; x[0, 0, -1, -1] = x[0, 0, 0, 0]          # an exact tie for the maximum
    xr = x.clone().requires_grad_(True)
    k = (shape[2], shape[3])
    avg, mx = F.avg_pool2d(xr, k), F.max_pool2d(xr, k)
    want = {"avg": avg, "max": mx, "avgmax": 0.5 * (avg + mx), "avgmaxc": torch.cat([avg, mx], 1)}[pool_type]
    cot = torch.randn_like(want)
    (want * cot).sum().backward()
    xh = x.to(dtype).cuda().contiguous(memory_format=torch.channels_last).requires_grad_(True)
    got = S.adaptive_avgmax_pool2d(xh, pool_type)
    assert got.shape == want.shape
    (got.float() * cot.cuda()).sum().backward()
    tol = 1e-5 if dtype == torch.float32 else 1e-2
    assert float((got.float().cpu() - want).abs().max()) < tol * max(1.0, float(want.abs().max()))
    assert float((xh.grad.float().cpu() - xr.grad).abs().max()) < tol * max(1.0, float(xr.grad.abs().max()))
    m = S.AdaptiveAvgMaxPool2d(1, pool_type)
    assert m.factor() == (2 if pool_type == "avgmaxc" else 1)
    assert torch.equal(m(xh.detach()), got.detach())


@pytest.mark.parametrize("dtype,c,ctot,shape", [(torch.bfloat16, 1, 1, (4, 64, 64)), (torch.bfloat16, 2, 2, (3, 16, 48)), (torch.float32, 4, 4, (2, 64, 64)),
                                                (torch.bfloat16, 64, 96, (2, 32, 32)), (torch.float32, 3, 3, (2, 16, 16)), (torch.bfloat16, 8, 8, (5, 16, 16)),
                                                (torch.float32, 1, 1, (3, 16, 16))])
def test_channel_sum_vector_and_scalar_paths(dtype, c, ctot, shape):
    """bias-gradient reduction (sum over pixels per channel): 16-byte-chunk path, the flat view of dense few-channel tensors, and the scalar
    fallback (odd channel counts) against a float64 sum"""
    import saunet_amd as S
    HF = S.functional
    n, h, w = shape
    torch.manual_seed(c * 7 + h)
    full = torch.randn(n, ctot, h, w, device="cuda").to(dtype).contiguous(memory_format=torch.channels_last)
    t = full[:, :c]
    got = HF.channel_sum(t)
    want = t.double().sum((0, 2, 3))
    assert got.shape == (c,)
    assert float((got.double() - want).abs().max()) < 1e-5 * max(1.0, float(want.abs().max())) * (1 if dtype == torch.float32 else 10)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_deferred_wgrad_reduction_matches_the_immediate_one(dtype):
    """saunet_conv2d_wgrad_deferred + saunet_wgrad_reduce_multi (several pending reductions in one launch) against saunet_conv2d_wgrad."""
    import saunet_amd as S
    HF = S.functional
    torch.manual_seed(3)
    # (the last three: many pixel groups over a tiny weight tensor -- the sliced fold of the multi kernel; the LDS-DMA kernel of the decoder's
    # c3x3rb geometry, whose [tap][co][ci] partials the multi kernel permutes; few groups)
    cases = [(2, 64, 32, 32, 32, 3), (2, 96, 32, 32, 128, 1), (1, 128, 16, 48, 32, 3), (4, 16, 64, 64, 16, 3), (2, 128, 32, 32, 64, 3), (1, 256, 16, 16, 128, 3)]
    pend, deferred, immediate = [], [], []
    for (n, cin, h, w, cout, k) in cases:
        x = torch.randn(n, cin, h, w, device="cuda").to(dtype).contiguous(memory_format=torch.channels_last)
        dy = torch.randn(n, cout, h, w, device="cuda").to(dtype).contiguous(memory_format=torch.channels_last)
        wt = torch.nn.Parameter(torch.randn(cout, cin, k, k, device="cuda") * 0.05)
        immediate.append(HF.conv_wgrad_raw(x, dy, wt, 1, k // 2).clone())
        deferred.append(HF.conv_wgrad_raw(x, dy, wt, 1, k // 2, pending=pend))
    assert len(pend) == len(cases), "the tiled kernels should have left one pending reduction per case"
    HF.flush_wgrad_reductions(pend)
    torch.cuda.synchronize()
    assert not pend
    for a, b in zip(deferred, immediate):
        assert float((a - b).abs().max()) <= 1e-5 * float(b.abs().max())


def test_deferred_conv_transpose_wgrad_reduction():
    """ConvTranspose2d(4, 2, 1) weight gradient on the LDS-DMA kernel with its permuting reduction deferred (taps = 16) against the immediate one"""
    import saunet_amd as S
    HF = S.functional
    torch.manual_seed(5)
    x = torch.randn(2, 128, 16, 16, device="cuda").to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    dy = torch.randn(2, 128, 32, 32, device="cuda").to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    wt = torch.nn.Parameter(torch.randn(128, 128, 4, 4, device="cuda") * 0.05)
    want = HF.conv_wgrad_raw(x, dy, wt, 2, 1, transposed=True).clone()
    pend = []
    got = HF.conv_wgrad_raw(x, dy, wt, 2, 1, transposed=True, pending=pend)
    assert len(pend) == 1 and pend[0][0].taps == 16
    HF.flush_wgrad_reductions(pend)
    assert float((got - want).abs().max()) <= 1e-5 * float(want.abs().max())


def test_backward_pass_defers_every_reduction_to_one_launch():
    """Inside a backward pass the per-layer reductions are collected and run by ONE saunet_wgrad_reduce_multi launch from the autograd engine's
    final callback (functional._DEFERRED); the gradients equal those of the per-layer reductions (SAUNET_WGRAD_DEFER=0) and a parameter that
    already holds a gradient is not deferred (AccumulateGrad would add the incomplete tensor)."""
    import saunet_amd as S
    from saunet_amd import lib as L
    HF = S.functional
    S.set_compute_dtype(torch.bfloat16)
    try:
        torch.manual_seed(11)
        ws = [torch.nn.Parameter(torch.randn(co, ci, k, k, device="cuda") * 0.05) for (co, ci, k) in ((64, 16, 3), (128, 64, 3), (32, 128, 1))]
        x = torch.randn(2, 16, 32, 32, device="cuda").to(torch.bfloat16).contiguous(memory_format=torch.channels_last).requires_grad_(True)

        def run():
            for w in ws:
                w.grad = None
            y = x
            for w in ws:
                y = HF.conv2d(y, w, None, 1, w.shape[2] // 2)
            names, orig, handle = [], L.call, L.load()

            def traced(name, *args):          # (the launch log is per thread: read it on the thread that made the call -- autograd's worker)
                handle.saunet_launch_log()
                orig(name, *args)
                names.extend(n for n in (handle.saunet_launch_log() or b"").decode().split("+") if n)
            L.call = traced
            try:
                y.float().square().sum().backward()
            finally:
                L.call = orig
            torch.cuda.synchronize()
            return [w.grad.clone() for w in ws], names
        was = HF.WGRAD_DEFER
        try:
            HF.WGRAD_DEFER = False
            want, names0 = run()
            HF.WGRAD_DEFER = True
            got, names1 = run()
            assert not HF._DEFERRED
            n0 = sum(1 for n in names0 if "reduce" in n)
            n1 = sum(1 for n in names1 if "reduce" in n)
            assert n0 == 3 and n1 == 1, (names0, names1)
            for a, b in zip(got, want):
                assert float((a - b).abs().max()) <= 1e-5 * float(b.abs().max())
            # second backward on top of existing gradients: accumulation must see complete tensors
            y = x
            for w in ws:
                y = HF.conv2d(y, w, None, 1, w.shape[2] // 2)
            y.float().square().sum().backward()
            torch.cuda.synchronize()
            for w, b in zip(ws, want):
                assert float((w.grad - 2 * b).abs().max()) <= 2e-5 * float(b.abs().max())
        finally:
            HF.WGRAD_DEFER = was
    finally:
        S.set_compute_dtype(torch.float32)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("shape", [(32, 128, 3, 3), (40, 72, 3, 3), (128, 200, 1, 1), (64, 8, 7, 7), (9, 33, 1, 1)])
def test_weight_packings_are_the_documented_permutations(dtype, shape):
    """include/saunet_hip.h saunet_pack_mode: FWD [Co][kh][kw][Ci], DGRAD [Ci][kh'][kw'][Co] with flipped taps (tiled through LDS since round 3;
    ragged channel counts and the element-wise path of the 7x7 stem included)"""
    import ctypes as C
    from saunet_amd import lib as L
    co, ci, kh, kw = shape
    w = rnd(co, ci, kh, kw).cuda()
    for mode, ref in ((L.PACK_FWD, w.permute(0, 2, 3, 1)), (L.PACK_DGRAD, w.flip(2, 3).permute(1, 2, 3, 0))):
        out = torch.full((w.numel(),), float("nan"), dtype=dtype, device="cuda")
        L.call("saunet_pack_weight", mode, L.BF16 if dtype == torch.bfloat16 else L.F32, w.data_ptr(), co, ci, kh, kw, out.data_ptr(), L.stream())
        assert torch.equal(out, ref.contiguous().reshape(-1).to(dtype)), (mode, shape)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("shape", [(64, 32), (48, 40), (130, 8)])
def test_conv_transpose_weight_packings(dtype, shape):
    """CONVT_FWD [ph][pw][Co][th][tw][Ci] with kh = (1 - ph) + 2 th, kw = (1 - pw) + 2 tw;  CONVT_DGRAD [Ci][kh][kw][Co]"""
    from saunet_amd import lib as L
    ci, co = shape
    w = rnd(ci, co, 4, 4).cuda()
    v = w.view(ci, co, 2, 2, 2, 2)                       # kh = 2 th + (1 - ph): dims (th, q = 1 - ph, tw, r = 1 - pw)
    fwd = v.flip(3, 5).permute(3, 5, 1, 2, 4, 0)         # [ph][pw][co][th][tw][ci]
    for mode, ref in ((L.PACK_CONVT_FWD, fwd), (L.PACK_CONVT_DGRAD, w.permute(0, 2, 3, 1))):
        out = torch.full((w.numel(),), float("nan"), dtype=dtype, device="cuda")
        L.call("saunet_pack_weight", mode, L.BF16 if dtype == torch.bfloat16 else L.F32, w.data_ptr(), co, ci, 4, 4, out.data_ptr(), L.stream())
        assert torch.equal(out, ref.contiguous().reshape(-1).to(dtype)), (mode, shape)


@pytest.mark.parametrize("dtype", DTYPES)
def test_multi_tensor_packing_matches_torch_permutes(dtype):
    """saunet_pack_weight_multi (LDS-tiled since round 4): all four packings, 1x1 / 3x3 / 4x4 / 7x7 / 5x5 (generic path) kernels, channel counts
    that are not multiples of the tile, against the permutations written with torch; bit-exact (a packing is a rounding + a permutation)."""
    import ctypes as C
    from saunet_amd import lib as L
    cases = []                                   # (mode, weight, reference)
    for co, ci, k in ((128, 160, 1), (37, 70, 1), (32, 128, 3), (72, 40, 3), (512, 96, 3), (64, 8, 7), (10, 6, 5)):
        w = rnd(co, ci, k, k, seed=k).cuda()
        cases.append((L.PACK_FWD, w, w.permute(0, 2, 3, 1), (co, ci, k, k)))
        cases.append((L.PACK_DGRAD, w, w.flip(2, 3).permute(1, 2, 3, 0), (co, ci, k, k)))
    for ci, co in ((48, 32), (128, 128), (20, 70)):
        w = rnd(ci, co, 4, 4, seed=9).cuda()
        v = w.view(ci, co, 2, 2, 2, 2)
        cases.append((L.PACK_CONVT_FWD, w, v.flip(3, 5).permute(3, 5, 1, 2, 4, 0), (co, ci, 4, 4)))
        cases.append((L.PACK_CONVT_DGRAD, w, w.permute(0, 2, 3, 1), (co, ci, 4, 4)))
    pl = L.PackList()
    pl.count = len(cases)
    outs = []
    for i, (mode, w, ref, dims) in enumerate(cases):
        out = torch.full((w.numel(),), float("nan"), dtype=dtype, device="cuda")
        pl.mode[i] = mode
        for j in range(4):
            pl.dims[i][j] = dims[j]
        pl.src[i] = w.data_ptr(); pl.dst[i] = out.data_ptr()
        outs.append(out)
    L.call("saunet_pack_weight_multi", C.byref(pl), L.BF16 if dtype == torch.bfloat16 else L.F32, L.stream())
    torch.cuda.synchronize()
    for (mode, w, ref, dims), out in zip(cases, outs):
        assert torch.equal(out, ref.contiguous().reshape(-1).to(dtype)), (mode, dims)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("shape", [(4, 64, 128, 128, 64), (2, 32, 64, 128, 32), (1, 16, 256, 256, 16), (2, 64, 16, 16, 64), (2, 256, 32, 32, 128)])
def test_conv_dgrad_accumulates_onto_the_residual_gradient(dtype, shape):
    """out += dgrad(dy, w) (BasicBlock backward: the skip branch's gradient is already in `out`; /root/reference/models/resnet.py:30-59): the first
    three geometries have the fused epilogue (saunet_conv2d_accumulate_supported), the others take the separate add -- same result contract."""
    hf = HF()
    n, cin, h, w, cout = shape
    wt = rnd(cout, cin, 3, 3, scale=0.05, seed=2)
    dy = rnd(n, cout, h, w, seed=3)
    base = rnd(n, cin, h, w, seed=4)
    out = to_dev(base, dtype)
    got = hf.conv_dgrad_raw(to_dev(dy, dtype), wt.cuda(), (n, cin, h, w), 1, 1, out=out, accumulate=True)
    assert got.data_ptr() == out.data_ptr()
    xr = torch.zeros(n, cin, h, w, requires_grad=True)
    F.conv2d(xr, q(wt, dtype), None, 1, 1).backward(q(dy, dtype))
    ref = q(base, dtype) + xr.grad
    close(out, ref, TOL[dtype] * (2 if dtype == torch.bfloat16 else 1), "out += dx")


@pytest.mark.parametrize("hw", [32, 160])      # 2 x 32 x 32 = 2048 pixels: exact f32 MFMA; 2 x 160 x 160 = 51200 >= 16384: 3 x bf16 split MFMA
def test_f32_split_nonfinite_operand_poisons_only_its_outputs(hw):
    """float32 storage, one +Inf in the input of a 3x3 convolution: the outputs inside its 3x3 footprint are non-finite -- +-Inf
    from the exact f32 MFMA of small launches, NaN from the split MFMA of large ones (csrc/common.h f32_split3: h = Inf, x - h = NaN; the guard
    was measured at +6 % of the float32 step and not adopted) -- and EVERY output outside the footprint is finite and equals the clean run."""
    import saunet_amd
    HF = saunet_amd.functional
    torch.manual_seed(3)
    x = torch.randn(2, 32, hw, hw, device="cuda").contiguous(memory_format=torch.channels_last)
    w = torch.nn.Parameter(torch.randn(32, 32, 3, 3, device="cuda") * 0.05)
    clean = HF.conv_forward_raw(x, w, None, 1, 1).clone()
    x2 = x.clone(); x2[1, 5, 10, 12] = float("inf")
    y = HF.conv_forward_raw(x2, w, None, 1, 1)
    torch.cuda.synchronize()
    foot = torch.zeros(2, 1, hw, hw, dtype=torch.bool, device="cuda"); foot[1, :, 9:12, 11:14] = True
    bad = ~torch.isfinite(y)
    assert bool(bad[foot.expand_as(bad)].all())                      # every channel reads channel 5 with a non-zero weight
    assert not bool(bad[~foot.expand_as(bad)].any())
    assert torch.equal(y[~foot.expand_as(y)], clean[~foot.expand_as(clean)])
    if hw * hw * 2 < 16384:
        assert bool(torch.isinf(y[foot.expand_as(y)]).all())         # exact f32 MFMA: Inf propagates as Inf
