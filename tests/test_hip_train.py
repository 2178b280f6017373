"""GPU: the training loop end to end on synthetic slices, and checkpoint interchange with the reference layout
(a checkpoint written by the HIP path drives the CPU oracle to the same eval-mode predictions)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_short_training_run_and_checkpoint_roundtrip(tmp_path):
    import saunet_amd as S
    from saunet_amd import train
    from oracle import saunet_ref as R
    args = ["--num_epoch", "2", "--batch_size_per_gpu", "4", "--synthetic", "8", "--size", "64", "--val_slices", "2",
            "--dtype", "f32", "--workers", "0", "--ckpt", str(tmp_path), "--disp_iter", "1", "--lr_encoder", "0.002"]
    hist = train.main(args)
    assert len(hist["train"]["loss"]) == 2 and hist["train"]["loss"][1] < hist["train"]["loss"][0]
    ck = os.path.join(str(tmp_path), "unet_epoch_2.pth")
    assert os.path.exists(ck)
    sd = torch.load(ck, map_location="cpu")
    want = {k for k, _, _ in R.state_dict_spec()}
    assert want <= set(sd.keys())
    # the oracle (reference key layout) consumes the checkpoint directly
    img, seg, edge = S.data.synthetic_batch(1, 64, 64, seed=77)
    sdo = {k: sd[k].clone() for k in want}
    with torch.no_grad():
        lo, eo = R.saunet_forward(sdo, img, training=False)
    S.set_compute_dtype(torch.float32)
    net = S.SAUNet(num_classes=4).cuda()
    net.load_state_dict(sd, strict=True)
    net.eval()
    with torch.no_grad():
        lh, eh = net(img.cuda())
    assert float((lh.cpu() - lo).abs().max()) < 1e-3 * max(1.0, float(lo.abs().max()))
    assert float((eh.cpu() - eo).abs().max()) < 1e-3


def test_fused_radam_matches_reference_formula():
    """one parameter tensor, 8 steps: fused RAdam kernel vs a line-by-line evaluation of radam.py:39-76 in torch."""
    import saunet_amd as S
    from saunet_amd.optim import FusedRAdam
    torch.manual_seed(0)
    p = torch.nn.Parameter(torch.randn(1000, device="cuda"))
    ref = p.detach().clone().double(); m = torch.zeros_like(ref); v = torch.zeros_like(ref)
    opt = FusedRAdam([p], lr=1e-2)
    for step in range(1, 9):
        g = torch.randn(1000, device="cuda")
        p.grad = g.clone()
        opt.step()
        gd = g.double()
        v = 0.999 * v + 0.001 * gd * gd; m = 0.9 * m + 0.1 * gd
        n_sma, step_size = FusedRAdam.schedule(step, 1e-2, 0.9, 0.999)
        ref = ref - step_size * m / (v.sqrt() + 1e-8) if n_sma >= 5 else ref - step_size * m
        assert float((p.detach().double() - ref).abs().max()) < 1e-5, step


def _make_training(dtype, seed=3, B=2, H=64, lr=5e-3):
    import saunet_amd as S
    from oracle import saunet_ref as R, weights as Wt
    S.set_compute_dtype(dtype)
    net = S.SAUNet(num_classes=4).cuda()
    net.load_state_dict(Wt.make_state_dict(R.state_dict_spec(), seed), strict=False)
    S.functional.notify_params_changed()
    sm = S.SegmentationModule(S.DualLoss(mode="train"), net, 4).train()
    opt = S.optim.create_optimizers(net, "sgd", lr=lr, momentum=0.9, weight_decay=1e-4)[0]
    img, seg, edge = Wt.synthetic_batch(B, H, H, seed=61)
    feed = {"image": img.cuda(), "mask": (seg.cuda(), edge.cuda())}
    return S, net, sm, opt, feed


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_graph_replays_follow_the_eager_trajectory(dtype):
    """A captured fwd+bwd+SGD step must re-pack the weights inside the graph: N replays reproduce the eager loss curve (a graph that
    trained against frozen packed weights would print the SAME loss on every replay), and an eager eval after the replays sees the
    replayed weights (host-side pack / eval-BN caches are invalidated by GraphedStep.replay)."""
    import saunet_amd as S
    from saunet_amd.graph import GraphedStep
    # the loss of this B=2 toy is sensitive to summation-order noise (float atomics) and the sensitivity grows with the learning rate
    # (scripts/graph_debug3.py: two EAGER fp32 runs differ by 1.6e-2 at lr 5e-3 and by 3e-4 at lr 2e-4; bf16 runs by ~5e-2 at any lr), so the
    # test trains gently: a graph that replayed frozen packed weights would hold the loss constant while the eager curve falls 0.1-0.3 per step
    # Round 3: the convolution epilogues fold their partial sums in a fixed order now (no float atomics on LDS); single-step gradients repeat to
    # 4e-7 of their scale (scripts/smoke_variance.py) and five float32 steps at this learning rate to 2e-4 in the loss
    # (scripts/replay_vs_eager_probe.py) -- not bit-equal: float64 atomics between workgroups and the float atomics of the SE pool / the
    # few-channel weight gradients still depend on arrival order.
    lr, tol0, tol, wtol = (2e-4, 1e-4, 1e-3, 1e-5) if dtype == torch.float32 else (1e-3, 0.1, 0.15, 3e-4)
    try:
        S_, net, sm, opt, feed = _make_training(dtype, lr=lr)
        w_init = net.final.weight.detach().clone()
        eager = []
        for _ in range(6):
            sm.zero_grad(set_to_none=True)
            loss, _ = sm(feed, 1)
            loss.backward(); opt.step(); eager.append(float(loss))
        w_eager = net.final.weight.detach().clone()
        sm.eval()
        with torch.no_grad():
            ref_eval = net(feed["image"])[0].float().clone()

        S_, net, sm, opt, feed = _make_training(dtype, lr=lr)

        def step():
            sm.zero_grad(set_to_none=True)
            loss, _ = sm(feed, 1)
            loss.backward()
            opt.step(upload=False)
            return loss.detach()

        g = GraphedStep(step, warmup=1, optimizers=[opt])          # the warm-up call is training step 0
        replayed = []
        for _ in range(5):
            replayed.append(float(g.replay()))
        torch.cuda.synchronize()
        assert eager[1] - eager[-1] > 0.3, eager                           # the eager curve falls by much more than the tolerance
        assert abs(eager[1] - replayed[0]) < tol0, (eager, replayed)
        assert max(abs(a - b) for a, b in zip(eager[1:], replayed)) < tol, (eager, replayed)
        wscale = float(w_eager.abs().max())
        assert float((w_eager - w_init).abs().max()) > 5 * wtol * wscale    # ... and so do the weights
        assert float((net.final.weight.detach() - w_eager).abs().max()) < wtol * wscale
        sm.eval()
        with torch.no_grad():
            got = net(feed["image"])[0].float()
        scale = float(ref_eval.abs().max())
        assert float((got - ref_eval).abs().max()) < (0.15 if dtype == torch.float32 else 0.5) * scale      # same weights to ~1e-2 -> eval logits to a few percent
    finally:
        S.set_compute_dtype(torch.float32)


def test_fused_adam_matches_torch_adam():
    """fused Adam kernel against torch.optim.Adam on the CPU (float64), 8 steps, the construction of train.py:197-201 (no weight decay)
    plus one run with L2 weight decay."""
    from saunet_amd.optim import FusedAdam
    for wd in (0.0, 1e-2):
        torch.manual_seed(1)
        p = torch.nn.Parameter(torch.randn(1537, device="cuda"))
        q = torch.nn.Parameter(p.detach().cpu().double())
        opt = FusedAdam([p], lr=1e-2, weight_decay=wd)
        ref = torch.optim.Adam([q], lr=1e-2, betas=(0.9, 0.999), weight_decay=wd)
        for step in range(8):
            g = torch.randn(1537, device="cuda")
            p.grad = g.clone(); q.grad = g.cpu().double()
            opt.step(); ref.step()
            assert float((p.detach().cpu().double() - q.detach()).abs().max()) < 2e-6, (wd, step)
    sd = opt.state_dict()
    assert set(sd["param_groups"][0].keys()) >= {"lr", "betas", "eps", "weight_decay", "params"}
    assert not any(k.startswith("_") for k in sd["param_groups"][0])          # no private bookkeeping leaks into checkpoints


def test_fused_sgd_state_dict_roundtrip_keeps_momentum():
    """optimizer.state_dict() holds only torch-style entries; a resumed FusedSGD continues with the loaded momentum buffers."""
    from saunet_amd.optim import FusedSGD
    torch.manual_seed(2)
    p = torch.nn.Parameter(torch.randn(300, device="cuda"))
    q = torch.nn.Parameter(p.detach().clone())
    a = FusedSGD([p], lr=0.1, momentum=0.9)
    ref = torch.optim.SGD([q], lr=0.1, momentum=0.9)
    gs = [torch.randn(300, device="cuda") for _ in range(4)]
    for g in gs[:2]:
        p.grad = g.clone(); q.grad = g.clone(); a.step(); ref.step()
    sd = a.state_dict()
    assert not any(k.startswith("_") for k in sd["param_groups"][0])
    b = FusedSGD([p], lr=0.1, momentum=0.9)
    b.load_state_dict(sd)
    for g in gs[2:]:
        p.grad = g.clone(); q.grad = g.clone(); b.step(); ref.step()
    assert float((p.detach() - q.detach()).abs().max()) < 1e-5


def test_training_with_device_augmentation(tmp_path):
    """--augment: raw ragged slices -> DeviceAugmenter (crop/pad, flips, rotation, gamma, z-score, elastic deformation, edges on the GPU) -> step."""
    from saunet_amd import train
    args = ["--num_epoch", "2", "--batch_size_per_gpu", "4", "--synthetic", "8", "--size", "64", "--val_slices", "2", "--dtype", "f32",
            "--workers", "0", "--ckpt", str(tmp_path), "--disp_iter", "1", "--lr_encoder", "0.002", "--augment"]
    hist = train.main(args)
    assert len(hist["train"]["loss"]) == 2 and all(np.isfinite(hist["train"]["loss"]))
    assert 0.0 <= hist["train"]["acc"][-1] <= 1.0


@pytest.mark.parametrize("name", ["adam", "radam"])
def test_graphed_adam_family_refreshes_bias_correction(name):
    """A captured Adam / RAdam step reads its bias-correction terms from the device hyper-parameter array: GraphedStep refreshes that array before
    every replay and advances the host step counters after it, so N replays equal N eager steps (a graph that baked step 1's terms would drift)."""
    from saunet_amd.graph import GraphedStep
    from saunet_amd.optim import FusedAdam, FusedRAdam
    torch.manual_seed(4)
    cls = FusedAdam if name == "adam" else FusedRAdam
    grad = torch.randn(2000, device="cuda")
    ref_p = torch.nn.Parameter(torch.randn(2000, device="cuda")); ref_p.grad = grad.clone()
    p = torch.nn.Parameter(ref_p.detach().clone()); p.grad = grad.clone()
    ref = cls([ref_p], lr=1e-2)
    for _ in range(7):
        ref.step()
    opt = cls([p], lr=1e-2)
    g = GraphedStep(lambda: opt.step(upload=False), warmup=1, optimizers=[opt])          # the warm-up call is step 1
    for _ in range(6):
        g.replay()
    torch.cuda.synchronize()
    assert opt.state[p]["step"] == 7 == ref.state[ref_p]["step"]
    assert float((p.detach() - ref_p.detach()).abs().max()) < 1e-6
