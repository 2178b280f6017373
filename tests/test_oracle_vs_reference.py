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
