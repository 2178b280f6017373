"""bf16-storage parity at the BENCHMARKED geometry (BASELINE configs[1]/[3]: 256x256, bf16) and the float32 high-resolution
geometry (configs[4]: 512x512).

What "parity" can mean in bf16.  SAUNet at 256x256 ends in 8x8 / 16x16 maps normalised by BATCH statistics behind ReLUs; rounding
any forward tensor (or just the weights) to bfloat16 flips a fraction of the ReLU masks and the deep DenseNet blocks amplify it: the
REFERENCE ITSELF evaluated with bf16-rounded tensors (oracle.saunet_ref.bf16_storage -- float32 arithmetic, every stored activation /
gradient / weight operand rounded to bf16) lands at a gradient cosine of only ~0.75-0.85 against its own float32 gradients in
denseblock2-4 / center / dec5, ~0.99 in denseblock1 and >0.999 in the full-resolution shape stream and head.  Rounding the GRADIENTS
alone changes nothing (cosine 1.000); the forward perturbation is what matters, so no bf16 implementation can do better.
The tests therefore pin the HIP bf16 path to BOTH references:
  * exact float32 oracle: loss within 1e-2 relative; every parameter group whose bf16 emulation is benign (cosine >= 0.99) must be
    benign in the HIP path too (absolute bound, stated per group below);
  * bf16-storage emulation of the oracle: in every group the HIP path's deviation from the exact gradients must not exceed the
    emulation's deviation by more than a stated margin (median cosine >= emulation - 0.05, worst parameter >= emulation's worst - 0.12).
float32 storage keeps the north-star bound (1e-3 on every gradient) at every geometry, including 512x512 here.
"""
import collections
import os

import numpy as np
import pytest
import torch

from oracle import saunet_ref as R, weights as Wt

pytestmark = pytest.mark.gpu

GROUPS_BENIGN = ("res1", "res2", "res3", "gate1", "gate2", "gate3", "d0", "d1", "d2", "d3", "fuse", "cw", "final", "dec0", "dec1", "dec2")


def group_of(key):
    return key.split(".")[2] if key.startswith("encoder") else key.split(".")[0]


def oracle_grads(sd, keys, batch, emulate, training=True):
    s = {k: v.clone() for k, v in sd.items()}
    for k in keys:
        s[k].requires_grad_(True)
    with R.bf16_storage(emulate):
        loss, acc, logits, _ = R.segmentation_step(s, *batch, training)
    loss.backward()
    return float(loss), {k: s[k].grad for k in keys}, logits.detach()


def cos_rel(a, b):
    a = a.detach().double().reshape(-1).cpu(); b = b.double().reshape(-1)
    nb = float(b.norm())
    return float((a @ b) / (a.norm() * nb + 1e-300)), float((a - b).norm() / (nb + 1e-300))


def make_hip(sd, dtype):
    import saunet_amd as S
    S.set_compute_dtype(dtype)
    net = S.SAUNet(num_classes=4).cuda()
    res = net.load_state_dict(sd, strict=False)
    assert not res.unexpected_keys
    S.functional.notify_params_changed()
    return S, net, S.SegmentationModule(S.DualLoss(mode="train"), net, 4)


def report(lines, name):
    """the measured table also goes to gpurun_out/ (scratch) so it can be copied into profiles/"""
    text = "\n".join(lines)
    print(text)
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    try:
        os.makedirs(out, exist_ok=True)
        with open(os.path.join(out, name), "w") as f:
            f.write(text + "\n")
    except OSError:
        pass


def test_bf16_gradients_at_256_against_exact_and_bf16_emulated_oracle():
    """configs[1] geometry (256x256, bf16 storage), B=8: loss + EVERY parameter gradient (incl. the 24-deep accumulate chain of
    denseblock3) against the float32 oracle and against the oracle's bf16-storage emulation."""
    _bf16_gradient_parity(8, "r2_parity_bf16_256.txt")


@pytest.mark.slow
@pytest.mark.skipif(os.environ.get("SAUNET_SLOW") != "1", reason="B=32 oracle on the CPU (minutes): SAUNET_SLOW=1; the table of the last run is profiles/r06_parity_bf16_256_b32.txt")
def test_bf16_gradients_at_256_b32_the_benchmarked_batch():
    """the same comparison at the HEADLINE batch (configs[1]: B=32, 256x256, bf16) -- the geometry at which block 3 runs
    dense_conv1_dgrad_pair_kernel (256 tiles of 128 pixels) and every small-map kernel its bench tiling."""
    _bf16_gradient_parity(32, "r6_parity_bf16_256_b32.txt")


def _bf16_gradient_parity(B, table_name):
    import saunet_amd as S
    torch.set_num_threads(min(os.cpu_count() or 8, 32))
    spec = R.state_dict_spec()
    sd = Wt.make_state_dict(spec, 13)
    keys = Wt.trainable_keys(spec)
    batch = Wt.synthetic_batch(B, 256, 256, seed=113)
    loss_a, ga, _ = oracle_grads(sd, keys, batch, False)
    loss_b, gb, _ = oracle_grads(sd, keys, batch, True)
    try:
        S_, net, sm = make_hip(sd, torch.bfloat16)
        sm.train()
        loss, _ = sm({"image": batch[0].cuda(), "mask": (batch[1].cuda(), batch[2].cuda())}, 1)
        loss.backward()
        torch.cuda.synchronize()
        pd = dict(net.named_parameters())
        gmax = max(float(ga[k].abs().max()) for k in keys)
        hip, emu = collections.defaultdict(list), collections.defaultdict(list)
        tiny_bad = []
        for k in keys:
            if float(ga[k].abs().max()) < 1e-5 * gmax:      # (near-)zero gradients, e.g. a conv bias in front of a training-mode BN: no direction to compare
                if float(pd[k].grad.abs().max()) >= 1e-3 * gmax:
                    tiny_bad.append(k)
                continue
            hip[group_of(k)].append(cos_rel(pd[k].grad, ga[k]) + (k,))
            emu[group_of(k)].append(cos_rel(gb[k], ga[k]) + (k,))
        lines = ["bf16 gradient parity, B=%d 256x256, seed 13: loss exact %.6f  bf16-emulated oracle %.6f  HIP bf16 %.6f" % (B, loss_a, loss_b, float(loss)),
                 "%-14s %4s | %-26s | %-26s" % ("group", "n", "HIP bf16 vs exact f32", "bf16-emulated oracle vs exact"),
                 "%-14s %4s | %8s %8s %8s | %8s %8s %8s" % ("", "", "med cos", "min cos", "max rel", "med cos", "min cos", "max rel")]
        bad = []
        for g in sorted(hip, key=lambda g: np.median([r[0] for r in hip[g]])):
            h, e = hip[g], emu[g]
            hm, hmin, hrel = np.median([r[0] for r in h]), min(r[0] for r in h), max(r[1] for r in h)
            em, emin, erel = np.median([r[0] for r in e]), min(r[0] for r in e), max(r[1] for r in e)
            lines.append("%-14s %4d | %8.4f %8.4f %8.4f | %8.4f %8.4f %8.4f" % (g, len(h), hm, hmin, hrel, em, emin, erel))
            if g in ("norm0", "expand"):
                continue          # 1-3 parameters whose exact gradient is a near-cancellation (|g| ~ 1e-5 of the scale): direction is noise in any bf16 run
            # margins = the measured worst gaps + 40 %: five runs on three boxes gave 0.025-0.029 (median) and 0.046-0.049 (minimum), both in `center`
            # (round 3: 0.05 / 0.12)
            if hm < em - 0.04 or hmin < emin - 0.07:
                bad.append("%s: HIP median/min cosine %.4f/%.4f below the bf16 emulation's %.4f/%.4f" % (g, hm, hmin, em, emin))
            if g in GROUPS_BENIGN and hmin < 0.99:
                bad.append("%s: min cosine %.4f < 0.99 in a group that is benign under bf16 storage" % (g, hmin))
        report(lines, table_name)
        assert abs(float(loss) - loss_a) < 1e-2 * loss_a, (float(loss), loss_a)
        assert not tiny_bad, "gradients that are ~0 in the oracle are not small in the HIP path: %s" % tiny_bad
        assert not bad, "\n".join(bad)
    finally:
        S.set_compute_dtype(torch.float32)


def _dice(pred, seg):
    from saunet_amd import train as T
    i, u = T.intersection_and_union(pred, seg, 4)
    return T.dice_from_iu(i, u)


def test_eval_dice_parity_fp32_and_bf16_at_256():
    """Hard Dice per class (argmax of the eval-mode logits, train.dice_from_iu) of the HIP path against the oracle on identical
    weights, B=8 256x256.  float32 storage: within 1e-4 (north star).  bf16 storage: within the bf16-storage emulation's own deviation
    (x2) -- stated in the assertion message."""
    import saunet_amd as S
    spec = R.state_dict_spec()
    sd = Wt.make_state_dict(spec, 13)
    batch = Wt.synthetic_batch(8, 256, 256, seed=113)
    seg = batch[1].numpy()
    with torch.no_grad():
        lg_a, _ = R.saunet_forward({k: v.clone() for k, v in sd.items()}, batch[0], False)
        with R.bf16_storage(True):
            lg_b, _ = R.saunet_forward({k: v.clone() for k, v in sd.items()}, batch[0], False)
    d_a, d_b = _dice(lg_a.argmax(1).numpy(), seg), _dice(lg_b.argmax(1).numpy(), seg)
    lines = ["eval-mode hard Dice per class, B=8 256x256, seed 13", "exact f32 oracle      %s" % np.round(d_a, 6), "bf16-emulated oracle  %s  max|d| %.2e" % (np.round(d_b, 6), np.abs(d_b - d_a).max())]
    try:
        res = {}
        for dt in (torch.float32, torch.bfloat16):
            S_, net, sm = make_hip(sd, dt)
            net.eval()
            with torch.no_grad():
                lg, _ = net(batch[0].cuda())
            res[dt] = (_dice(lg.float().argmax(1).cpu().numpy(), seg), float((lg.float().cpu() - lg_a).abs().max()) / float(lg_a.abs().max()))
            lines.append("HIP %-8s          %s  max|d| %.2e  logits err %.2e of scale" % (str(dt).split(".")[1], np.round(res[dt][0], 6), np.abs(res[dt][0] - d_a).max(), res[dt][1]))
        report(lines, "r2_parity_dice_256.txt")
        assert np.abs(res[torch.float32][0] - d_a).max() < 1e-4, lines
        assert res[torch.float32][1] < 1e-3
        tol_bf16 = max(1e-4, 2.0 * float(np.abs(d_b - d_a).max()))
        assert np.abs(res[torch.bfloat16][0] - d_a).max() <= tol_bf16, (lines, tol_bf16)
    finally:
        S.set_compute_dtype(torch.float32)


def test_fp32_512_against_oracle():
    """configs[4] geometry: 512x512 float32 storage, B=1 -- loss and every parameter gradient within 1e-3 of the gradient scale."""
    import saunet_amd as S
    spec = R.state_dict_spec()
    sd = Wt.make_state_dict(spec, 17)
    keys = Wt.trainable_keys(spec)
    batch = Wt.synthetic_batch(1, 512, 512, seed=171)
    loss_a, ga, _ = oracle_grads(sd, keys, batch, False)
    S_, net, sm = make_hip(sd, torch.float32)
    sm.train()
    loss, _ = sm({"image": batch[0].cuda(), "mask": (batch[1].cuda(), batch[2].cuda())}, 1)
    loss.backward()
    assert abs(float(loss) - loss_a) < 1e-4 * max(1.0, loss_a)
    pd = dict(net.named_parameters())
    gmax = max(float(ga[k].abs().max()) for k in keys)
    for k in keys:
        err = float((pd[k].grad.cpu() - ga[k]).abs().max())
        assert err < 1e-3 * gmax, (k, err, gmax)
    from tests.golden_util import assert_grads_per_tensor
    assert_grads_per_tensor({k: pd[k].grad for k in keys}, ga, keys)           # configs[4] geometry: every tensor on its own scale


@pytest.mark.parametrize("dtype,B", [(torch.bfloat16, 32), (torch.float32, 4)])
def test_duplicated_batch_invariance_at_bench_batches(dtype, B):
    """configs[1] (B=32) and configs[3] (B=64) as benchmarked, bf16 at 256x256 (and a small float32 instance of the same property):
    a batch made of two copies of a B-slice batch has the same batch statistics, the same per-batch Dice ratio and the same mean loss,
    so loss(2B) == loss(B) and every parameter gradient is unchanged -- a size-independent check of every large-grid kernel variant,
    the BN statistic reductions and the loss normalisation at B=64.  In bf16 the B=32 forward loss is also checked against the
    float32 oracle evaluated at the full B=32."""
    import saunet_amd as S
    spec = R.state_dict_spec()
    sd = Wt.make_state_dict(spec, 23)
    img, seg, edge = Wt.synthetic_batch(B, 256, 256, seed=321)
    try:
        out = {}
        for rep in (1, 2):
            S_, net, sm = make_hip(sd, dtype)
            sm.train()
            feed = {"image": img.repeat(rep, 1, 1, 1).cuda(), "mask": (seg.repeat(rep, 1, 1).cuda(), edge.repeat(rep, 1, 1, 1).cuda())}
            loss, (acc, jac) = sm(feed, 1)
            loss.backward()
            torch.cuda.synchronize()
            out[rep] = (float(loss), float(acc), {k: p.grad.detach().double().cpu() for k, p in net.named_parameters() if p.grad is not None})
            assert all(torch.isfinite(g).all() for g in out[rep][2].values())
            del net, sm, feed, loss
            torch.cuda.empty_cache()
        rtol = 3e-4 if dtype == torch.bfloat16 else 1e-5            # measured (profiles/r04_duplicated_batch.txt): 5.4e-5 / 0
        assert abs(out[1][0] - out[2][0]) < rtol * out[1][0], (out[1][0], out[2][0])
        assert abs(out[1][1] - out[2][1]) < 1e-3
        gmax = max(float(g.abs().max()) for g in out[1][2].values())
        # error of every gradient tensor relative to its own norm (floored at 1e-3 of the global gradient scale: norm0 / expand.0 sit in
        # front of batch norms and have near-cancelled gradients whose direction is numerical noise)
        worst = max((float((out[2][2][k] - g).norm() / max(float(g.norm()), 1e-3 * gmax * g.numel() ** 0.5)), k) for k, g in out[1][2].items())
        # bf16: the two runs differ in the summation order of the float64 statistic atomics -> a few flipped roundings that the deep blocks
        # amplify (the same sensitivity the bf16-emulated oracle shows); float32: summation-order noise only
        try:      # the measured value next to the gate (gpurun merges gpurun_out/ back)
            os.makedirs(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out"), exist_ok=True)
            with open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "r04_duplicated_batch.txt"), "a") as f:
                f.write("%s B=%d: loss %.6f vs %.6f, worst gradient tensor %s rel-L2 %.4f\n" % (str(dtype), B, out[1][0], out[2][0], worst[1], worst[0]))
        except OSError:
            pass
        # measured over three runs (profiles/r04_duplicated_batch.txt): bf16 0.065 (denseblock2.denselayer1.conv2.weight, every run), float32 2e-4 .. 8e-4
        assert worst[0] < (0.10 if dtype == torch.bfloat16 else 2e-3), worst
        if dtype == torch.bfloat16:
            torch.set_num_threads(min(os.cpu_count() or 8, 32))
            with torch.no_grad():
                loss_o = float(R.segmentation_step({k: v.clone() for k, v in sd.items()}, img, seg, edge, True)[0])
            assert abs(out[1][0] - loss_o) < 1e-2 * loss_o, (out[1][0], loss_o)
    finally:
        S.set_compute_dtype(torch.float32)
