"""Container-only: run the REAL reference side by side with the restatement on fresh random
inputs (not the committed fixtures).  Skipped where /root/reference is absent (GPU box)."""
import contextlib
import io

import numpy as np
import pytest
import torch

from oracle import ref_import, saunet_ref as R, weights as Wt

pytestmark = pytest.mark.skipif(not ref_import.available(), reason="/root/reference not present")


def test_full_step_matches_reference_on_fresh_inputs():
    ns = ref_import.load()
    spec = R.state_dict_spec()
    sd = Wt.make_state_dict(spec, seed=21)
    with contextlib.redirect_stdout(io.StringIO()):
        net = ns.SAUNet(num_classes=4)
    net.load_state_dict(sd, strict=False)
    img, seg, edge = Wt.synthetic_batch(2, 64, 64, seed=99)
    sm = ns.SegmentationModule(ns.DualLoss(mode="train"), net, 4).train()
    loss, (acc, jac) = sm({"image": img, "mask": (seg.double(), edge)}, 1)
    loss.backward()
    sd2 = {k: v.clone() for k, v in sd.items()}
    keys = Wt.trainable_keys(spec)
    for k in keys:
        sd2[k].requires_grad_(True)
    loss2, acc2, _, _ = R.segmentation_step(sd2, img, seg, edge, True)
    loss2.backward()
    assert abs(float(loss) - float(loss2)) < 1e-6
    assert abs(float(acc) - float(acc2[0])) < 1e-7
    pd = dict(net.named_parameters())
    gmax = max(float(pd[k].grad.abs().max()) for k in keys)
    for k in keys:
        assert float((pd[k].grad - sd2[k].grad).abs().max()) < 1e-5 * gmax, k
    bd = dict(net.named_buffers())
    for k, _, kind in spec:
        if kind in ("rmean", "rvar"):
            assert float((bd[k] - sd2[k]).abs().max()) < 1e-6, k


def test_syncbn_parallel_branch_matches_reference_compute_mean_std():
    """The multi-replica SyncBN arithmetic of the oracle's data-parallel emulation (oracle/saunet_ref._sync_bn_train) against the
    reference's own `_compute_mean_std` + normalisation line (lib/nn/modules/batchnorm.py:77-84, 118-139), two consecutive steps."""
    ns = ref_import.load()
    bn = ns.resnet.BasicBlock(16, 16).bn1          # a SynchronizedBatchNorm2d as the path builds it (momentum 0.001)
    C = 16
    g = torch.Generator().manual_seed(5)
    bn.weight.data = torch.rand(C, generator=g) + 0.5
    bn.bias.data = torch.randn(C, generator=g) * 0.1
    sd = {"res3.bn1.weight": bn.weight.data.clone(), "res3.bn1.bias": bn.bias.data.clone(),
          "res3.bn1.running_mean": bn.running_mean.clone(), "res3.bn1.running_var": bn.running_var.clone(),
          "res3.bn1._tmp_running_mean": bn._tmp_running_mean.clone(), "res3.bn1._tmp_running_var": bn._tmp_running_var.clone(),
          "res3.bn1._running_iter": bn._running_iter.clone()}
    for step in range(2):
        x = torch.randn(6, C, 5, 7, generator=g) * 2 + 0.3
        xs = x.view(6, C, -1)
        n = xs.shape[0] * xs.shape[2]
        mean, inv_std = bn._compute_mean_std(xs.sum(0).sum(-1), (xs ** 2).sum(0).sum(-1), n)
        want = (xs - mean[None, :, None]) * (inv_std * bn.weight)[None, :, None] + bn.bias[None, :, None]
        got = R._sync_bn_train(sd, "res3.bn1", x, R.SYNCBN_MOM)
        assert float((got.view(6, C, -1) - want).abs().max()) < 1e-5
        assert float((sd["res3.bn1.running_mean"] - bn.running_mean).abs().max()) < 1e-6
        assert float((sd["res3.bn1.running_var"] - bn.running_var).abs().max()) < 1e-5
        assert float((sd["res3.bn1._running_iter"] - bn._running_iter).abs().max()) < 1e-6
