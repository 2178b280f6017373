"""Whole-network GPU parity: SAUNet forward + DualLoss + backward on the HIP path vs
 (a) the fixture the REAL reference produced for config #1 (B=2, 128x128) and
 (b) the CPU oracle on other sizes / seeds; plus size-independent properties at the bench size."""
import numpy as np
import pytest
import torch

from oracle import saunet_ref as R, weights as Wt
from tests.golden_util import load, close, assert_grads_per_tensor

pytestmark = pytest.mark.gpu


def make_net(seed, dtype=torch.float32):
    import saunet_amd as S
    S.set_compute_dtype(dtype)
    spec = R.state_dict_spec()
    sd = Wt.make_state_dict(spec, seed)
    net = S.SAUNet(num_classes=4).cuda()
    res = net.load_state_dict(sd, strict=False)
    assert not res.unexpected_keys
    sm = S.SegmentationModule(S.DualLoss(mode="train"), net, 4)
    return S, spec, sd, net, sm


def test_config1_against_reference_fixture():
    g = load("saunet_128.npz")
    seed, B, H = int(g["meta.seed"]), int(g["meta.B"]), int(g["meta.H"])
    S, spec, sd, net, sm = make_net(seed)
    img, seg, edge = Wt.synthetic_batch(B, H, H)
    feed = {"image": img.cuda(), "mask": (seg.cuda(), edge.cuda())}
    sm.train()
    loss, (acc, jac) = sm(feed, 1)
    loss.backward()
    assert abs(float(loss) - float(g["loss0"])) < 1e-4, (float(loss), float(g["loss0"]))
    assert abs(float(acc) - float(g["acc0"])) < 1e-4
    assert np.abs(np.array([float(j) for j in jac]) - g["jac0"]).max() < 1e-4
    keys = list(g["grad_keys"])
    pd = dict(net.named_parameters())
    norms = np.array([float(pd[k].grad.double().norm()) for k in keys])
    gmax = g["grad_norms"].max()
    assert np.abs(norms - g["grad_norms"]).max() <= 1e-3 * gmax, np.abs(norms - g["grad_norms"]).max() / gmax
    for k in g:
        if k.startswith("grad."):
            ok, err, sc = close(pd[k[5:]].grad.float().cpu().numpy(), g[k], 1e-3, 1e-5 * gmax)
            assert ok, "%s err %.3g scale %.3g" % (k, err, sc)
    bd = dict(net.named_buffers())
    for k in g:
        if k.startswith("buf."):
            ok, err, sc = close(bd[k[4:]].float().cpu().numpy(), g[k], 1e-4, 1e-6)
            assert ok, "%s err %.3g" % (k, err)
    # forward tensors with fresh weights
    net.load_state_dict(sd, strict=False)
    with torch.no_grad():
        lg, eo = net(img.cuda())
    ok, err, sc = close(lg[:, :, ::8, ::8].float().cpu().numpy(), g["logits_s8"], 1e-3, 1e-5)
    assert ok, "logits err %.3g scale %.3g" % (err, sc)
    ok, err, sc = close(eo[:, :, ::8, ::8].float().cpu().numpy(), g["edge_s8"], 1e-3, 1e-5)
    assert ok, "edge err %.3g" % err
    # inference branch on the initial weights
    net.load_state_dict(sd, strict=False)
    sm.eval()
    with torch.no_grad():
        pred, l_eval = sm({"image": img[:1].cuda(), "mask": (seg[0].cuda(), edge[0].cuda())}, epoch=0, segSize=(H, H))
    assert abs(float(l_eval) - float(g["eval0_loss"])) < 1e-3 * max(1.0, abs(float(g["eval0_loss"])))
    ok, err, sc = close(pred[:, :, ::8, ::8].float().cpu().numpy(), g["eval0_pred_s8"], 1e-3, 1e-5)
    assert ok, "eval softmax err %.3g" % err


def test_sgd_trajectory_config1():
    """10 fused-SGD steps (lr 5e-4, m 0.9, wd 1e-4 on conv weights) track the reference's loss curve."""
    g = load("saunet_128.npz")
    seed, B, H = int(g["meta.seed"]), int(g["meta.B"]), int(g["meta.H"])
    S, spec, sd, net, sm = make_net(seed)
    img, seg, edge = Wt.synthetic_batch(B, H, H)
    feed = {"image": img.cuda(), "mask": (seg.cuda(), edge.cuda())}
    opt = S.optim.create_optimizers(net, "sgd", lr=5e-4, momentum=0.9, weight_decay=1e-4)[0]
    sm.train()
    traj = []
    for it in range(10):
        sm.zero_grad()
        loss, _ = sm(feed, 1)
        loss.backward(); opt.step(); traj.append(float(loss))
    d = np.abs(np.array(traj) - g["sgd_traj"])
    assert d[:3].max() < 2e-4 and d.max() < 1e-2, (traj, list(g["sgd_traj"]))
    assert traj[-1] < traj[0] - 0.5


@pytest.mark.parametrize("B,H,W,seed", [(1, 64, 96, 7), (3, 64, 64, 9), (4, 256, 256, 13)])   # the last one reaches every large-grid kernel variant
def test_other_shapes_against_oracle(B, H, W, seed):
    S, spec, sd, net, sm = make_net(seed)
    img, seg, edge = Wt.synthetic_batch(B, H, W, seed=100 + seed)
    sdo = {k: v.clone() for k, v in sd.items()}
    keys = Wt.trainable_keys(spec)
    for k in keys:
        sdo[k].requires_grad_(True)
    loss_o, acc_o, lg_o, eo_o = R.segmentation_step(sdo, img, seg, edge, True)
    loss_o.backward()
    sm.train()
    loss, (acc, jac) = sm({"image": img.cuda(), "mask": (seg.cuda(), edge.cuda())}, 1)
    loss.backward()
    assert abs(float(loss) - float(loss_o)) < 1e-4 * max(1.0, float(loss_o))
    pd = dict(net.named_parameters())
    gmax = max(float(sdo[k].grad.abs().max()) for k in keys)
    for k in keys:
        err = float((pd[k].grad.cpu() - sdo[k].grad).abs().max())
        assert err < 1e-3 * gmax, (k, err, gmax)
    assert_grads_per_tensor({k: pd[k].grad for k in keys}, {k: sdo[k].grad for k in keys}, keys)      # every tensor on its OWN scale


def test_fix_bn_training_step_against_oracle():
    """train.py --fix_bn (train.py:85 `segmentation_module.train(not args.fix_bn)`): BatchNorm in eval mode (running statistics, no
    batch terms in its backward) while gradients flow -- loss and every parameter gradient against the oracle."""
    seed = 21
    S, spec, sd, net, sm = make_net(seed)
    img, seg, edge = Wt.synthetic_batch(2, 64, 96, seed=140)
    sdo = {k: v.clone() for k, v in sd.items()}
    keys = Wt.trainable_keys(spec)
    for k in keys:
        sdo[k].requires_grad_(True)
    loss_o, acc_o, lg_o, eo_o = R.segmentation_step(sdo, img, seg, edge, False)
    loss_o.backward()
    sm.eval()
    loss, (acc, jac) = sm({"image": img.cuda(), "mask": (seg.cuda(), edge.cuda())}, 1)
    loss.backward()
    assert abs(float(loss) - float(loss_o)) < 1e-4 * max(1.0, float(loss_o))
    pd = dict(net.named_parameters())
    gmax = max(float(sdo[k].grad.abs().max()) for k in keys)
    for k in keys:
        err = float((pd[k].grad.cpu() - sdo[k].grad).abs().max())
        assert err < 1e-3 * gmax, (k, err, gmax)
    assert_grads_per_tensor({k: pd[k].grad for k in keys}, {k: sdo[k].grad for k in keys}, keys)
    # running statistics untouched
    for k, v in net.state_dict().items():
        if k.endswith("running_mean") and "_tmp" not in k and k in sd:
            assert torch.equal(v.cpu(), sd[k]), k


def test_attention_maps_branch_against_oracle():
    """SegmentationModule test branch (segSize=True, models/models.py:96-103): softmax + the 7 attention / gate maps."""
    S, spec, sd, net, sm = make_net(17)
    img, seg, edge = Wt.synthetic_batch(2, 64, 64, seed=41)
    sm.eval()
    with torch.no_grad():
        pred, maps = sm({"image": img.cuda(), "mask": (seg.cuda(), edge.cuda())}, epoch=0, segSize=True)
        lg_o, eo_o, maps_o = R.saunet_forward({k: v.clone() for k, v in sd.items()}, img, False, return_att=True)
    assert len(maps) == 7 and len(maps_o) == 7
    ref = torch.softmax(lg_o, 1)
    assert float((pred.float().cpu() - ref).abs().max()) < 1e-4
    for i, (m, mo) in enumerate(zip(maps, maps_o)):
        assert tuple(m.shape) == tuple(mo.shape) == (2, 1, 64, 64), (i, m.shape, mo.shape)
        assert float((m.float().cpu() - mo).abs().max()) < 1e-4 * max(1.0, float(mo.abs().max())), i


def test_bf16_storage_tracks_fp32_oracle():
    """bf16 activations/weights, fp32 accumulate/statistics/loss: loss within 2% and Dice-style metrics close."""
    seed = 3
    S, spec, sd, net, sm = make_net(seed, torch.bfloat16)
    try:
        img, seg, edge = Wt.synthetic_batch(2, 128, 128)
        sdo = {k: v.clone() for k, v in sd.items()}
        loss_o, acc_o, _, _ = R.segmentation_step(sdo, img, seg, edge, True)
        sm.train()
        loss, (acc, jac) = sm({"image": img.cuda(), "mask": (seg.cuda(), edge.cuda())}, 1)
        loss.backward()
        assert abs(float(loss) - float(loss_o)) < 0.02 * float(loss_o), (float(loss), float(loss_o))
        assert all(torch.isfinite(p.grad).all() for p in net.parameters() if p.grad is not None)
    finally:
        S.set_compute_dtype(torch.float32)


def test_bf16_training_tracks_fp32_training():
    """40 fused-SGD steps on one fixed batch in bf16 storage (fused gate / expand / dense-block kernels) against the same run
    in float32 storage: both must fit the batch, and the loss curves must stay close (they are not bit-comparable: bf16
    rounding flips ReLU masks, see tests/test_hip_dense.py)."""
    curves = {}
    for dt in (torch.float32, torch.bfloat16):
        S, spec, sd, net, sm = make_net(29, dt)
        try:
            img, seg, edge = Wt.synthetic_batch(4, 128, 128, seed=77)
            feed = {"image": img.cuda(), "mask": (seg.cuda(), edge.cuda())}
            opt = S.optim.create_optimizers(net, "sgd", lr=2e-3, momentum=0.9, weight_decay=1e-4)[0]
            sm.train()
            c = []
            for it in range(40):
                sm.zero_grad(set_to_none=True)
                loss, _ = sm(feed, 1)
                loss.backward(); opt.step(); c.append(float(loss))
            curves[dt] = c
        finally:
            S.set_compute_dtype(torch.float32)
    f, b = np.array(curves[torch.float32]), np.array(curves[torch.bfloat16])
    assert np.isfinite(b).all()
    assert f[-1] < 0.6 * f[0] and b[-1] < 0.6 * b[0], (f[0], f[-1], b[0], b[-1])
    assert abs(b[0] - f[0]) < 0.02 * f[0]
    assert np.abs(b - f).max() < 0.12 * f[0], (list(f[::8]), list(b[::8]))


def test_properties_at_bench_size():
    """256x256 (BASELINE config 2 geometry, small batch): size-independent checks."""
    S, spec, sd, net, sm = make_net(11)
    img, seg, edge = Wt.synthetic_batch(2, 256, 256, seed=5)
    net.train()
    x = img.cuda()
    lg, eo = net(x)
    assert lg.shape == (2, 4, 256, 256) and eo.shape == (2, 1, 256, 256)
    assert float(eo.min()) >= 0 and float(eo.max()) <= 1 and torch.isfinite(lg).all()
    # batch-permutation equivariance of a train-mode step (batch statistics are permutation invariant)
    net.load_state_dict(sd, strict=False)
    lg1, _ = net(x)
    net.load_state_dict(sd, strict=False)
    lg2, _ = net(x.flip(0))
    assert float((lg1 - lg2.flip(0)).abs().max()) < 1e-3 * float(lg1.abs().max())
    # loss(logits) is invariant to adding a per-pixel constant to all class logits (softmax shift invariance)
    crit = S.DualLoss()
    l1 = crit((lg1.detach(), eo.detach()), (seg.cuda(), edge.cuda()))
    l2 = crit((lg1.detach() + 3.0, eo.detach()), (seg.cuda(), edge.cuda()))
    assert abs(float(l1) - float(l2)) < 1e-4


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 1e-3), (torch.bfloat16, 0.12)])
def test_inference_path_folded_and_cached(dtype, tol):
    """Eval-mode forward without grad (the inference path: BatchNorm folded into the conv weights -- incl. DenseNet norm2 -> conv1 --, folded
    weights cached across calls, device softmax/argmax) against the oracle's eval forward; the cache must follow weight changes
    (load_state_dict, an optimiser step) and survive unchanged weights (second call issues no fold / pack work: same result object ids)."""
    S, spec, sd, net, sm = make_net(23, dtype)
    try:
        img, seg, edge = Wt.synthetic_batch(2, 128, 128, seed=55)
        with torch.no_grad():
            lg_o, eo_o = R.saunet_forward({k: v.clone() for k, v in sd.items()}, img, False)
        net.eval()
        HF = S.functional
        with torch.no_grad():
            lg, eo = net(img.cuda())
            n_entries = len(HF.INFER.entries)
            ids = {k: id(v[2]) for k, v in HF.INFER.entries.items()}
            lg2, _ = net(img.cuda())
        assert n_entries >= 58 + 10                                   # every dense layer's folded conv1 + the conv-BN-ReLU units
        assert {k: id(v[2]) for k, v in HF.INFER.entries.items()} == ids      # second call: pure cache hits
        # float atomics in the SE pool: last-bit noise in float32, a few flipped bf16 roundings downstream in bf16
        assert float((lg.float() - lg2.float()).abs().max()) <= (1e-3 if dtype == torch.float32 else 3e-2) * float(lg.float().abs().max())
        scale = float(lg_o.abs().max())
        assert float((lg.float().cpu() - lg_o).abs().max()) < tol * scale
        assert float((eo.float().cpu() - eo_o).abs().max()) < max(tol, 2e-3)
        prob, label = HF.softmax_argmax(lg)
        ref = torch.softmax(lg.float(), 1)
        assert float((prob - ref).abs().max()) < 1e-5 and torch.equal(label, lg.float().argmax(1))
        # weights change -> folded copies must follow
        sd2 = Wt.make_state_dict(spec, 24)
        net.load_state_dict(sd2, strict=False)            # in-place copies bump the version counters the caches are keyed on
        with torch.no_grad():
            lg3, _ = net(img.cuda())
            lg3_o, _ = R.saunet_forward({k: v.clone() for k, v in sd2.items()}, img, False)
        assert float((lg3.float().cpu() - lg3_o).abs().max()) < tol * float(lg3_o.abs().max())
    finally:
        S.set_compute_dtype(torch.float32)


def test_eval_mode_with_gradients_in_bf16_storage():
    """eval mode WITH gradients (--fix_bn, the saliency scripts) in bf16 storage takes the unfused `expand` path, whose float32
    result is cast into the decoder's concat slice -- the forward must not trip cat_alias, and loss + gradients must follow the fp32 oracle
    as closely as bf16 storage allows (the same bounds as test_bf16_storage_tracks_fp32_oracle)."""
    seed = 23
    S, spec, sd, net, sm = make_net(seed, torch.bfloat16)
    try:
        img, seg, edge = Wt.synthetic_batch(2, 64, 96, seed=141)
        sdo = {k: v.clone() for k, v in sd.items()}
        keys = Wt.trainable_keys(spec)
        for k in keys:
            sdo[k].requires_grad_(True)
        loss_o, *_ = R.segmentation_step(sdo, img, seg, edge, False)
        loss_o.backward()
        sm.eval()
        loss, _ = sm({"image": img.cuda(), "mask": (seg.cuda(), edge.cuda())}, 1)
        loss.backward()
        assert abs(float(loss) - float(loss_o)) < 3e-2 * max(1.0, float(loss_o)), (float(loss), float(loss_o))
        pd = dict(net.named_parameters())
        num = sum(float((pd[k].grad.cpu().double() - sdo[k].grad.double()).pow(2).sum()) for k in keys)
        den = sum(float(sdo[k].grad.double().pow(2).sum()) for k in keys)
        assert (num / den) ** 0.5 < 0.25, (num / den) ** 0.5
        assert all(torch.isfinite(pd[k].grad).all() for k in keys)
    finally:
        S.set_compute_dtype(torch.float32)


@pytest.mark.parametrize("C", [3, 5, 19])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_softmax_argmax_any_class_count(C, dtype):
    """the device softmax / argmax head for class counts other than 2 / 4 / 8 (a 3-class dataset validates through it), incl. the
    first-maximum tie rule and NaN-wins of torch.argmax."""
    import saunet_amd as S
    HF = S.functional
    torch.manual_seed(C)
    lg = torch.randn(2, C, 24, 40, device="cuda").to(dtype).contiguous(memory_format=torch.channels_last)
    lg[0, :, 3, 5] = 1.25                      # an all-equal pixel: argmax = 0
    if dtype == torch.float32:
        lg[1, C - 1, 7, 9] = float("nan")
    prob, label = HF.softmax_argmax(lg)
    ref = torch.softmax(lg.float(), 1)
    ok = torch.isfinite(ref).all(1, keepdim=True).expand_as(ref)
    assert float((prob - ref)[ok].abs().max()) < 1e-5
    assert torch.equal(label, lg.float().argmax(1))
