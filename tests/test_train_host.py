"""Host-side logic of the training loop (CPU): schedules, grouping, metrics, checkpoint policy -- mirrors
/root/reference/train.py:150-216,294-329 and utils.py:119-140."""
import math

import numpy as np
import torch


def test_cosine_and_resume_lr():
    import saunet_amd
    from saunet_amd import optim, train
    assert optim.cosine_lr(5e-4, 0, 120) == 5e-4
    assert abs(optim.cosine_lr(1e-3, 60, 120) - 1e-3 * 0.5 * (1 + math.cos(3.14159 * 0.5))) < 1e-12
    assert optim.cosine_lr(1e-3, 120, 120) < 1e-9
    assert abs(train.poly_resume_lr(1e-3, 31, 120) - 1e-3 * (1 - 30 / 120) ** 0.9) < 1e-12


def test_radam_schedule_matches_reference_formula():
    import saunet_amd
    from saunet_amd.optim import FusedRAdam
    # radam.py:52-66 evaluated by hand for a few steps
    for step in (1, 2, 5, 6, 100, 1000):
        beta1, beta2, lr = 0.9, 0.999, 1e-4
        b2t = beta2 ** step
        nmax = 2 / (1 - beta2) - 1
        nsma = nmax - 2 * step * b2t / (1 - b2t)
        if nsma >= 5:
            want = lr * math.sqrt((1 - b2t) * (nsma - 4) / (nmax - 4) * (nsma - 2) / nsma * nmax / (nmax - 2)) / (1 - beta1 ** step)
        else:
            want = lr / (1 - beta1 ** step)
        got_n, got = FusedRAdam.schedule(step, lr, beta1, beta2)
        assert abs(got - want) < 1e-15 and abs(got_n - nsma) < 1e-9
    assert FusedRAdam.schedule(5, 1e-4, 0.9, 0.999)[0] < 5 <= FusedRAdam.schedule(6, 1e-4, 0.9, 0.999)[0]


def test_intersection_and_union_and_dice():
    import saunet_amd
    from saunet_amd import train
    from oracle import saunet_ref as R
    r = np.random.default_rng(0)
    pred, lab = r.integers(0, 4, (32, 32)), r.integers(0, 4, (32, 32))
    a, u = train.intersection_and_union(pred, lab, 4)
    a2, u2 = R.intersection_and_union(pred, lab, 4)
    assert (a == a2).all() and (u == u2).all()
    d = train.dice_from_iu(a, u)
    for c in range(4):
        inter = ((pred == c) & (lab == c)).sum()
        want = 2 * inter / ((pred == c).sum() + (lab == c).sum())
        assert abs(d[c] - want) < 1e-9


def test_checkpoint_policy():
    """/root/reference/train.py:294-329: bests are tracked from epoch 1, a new best saves from epoch 15 on, every 50th epoch and the
    last epoch always save."""
    import saunet_amd
    from saunet_amd import train
    best = {"class": [0.0, 0.0, 0.0], "mean": 0.0}
    assert not train.should_checkpoint(10, [0.3, 0.3, 0.3], best, 120)      # improvement before epoch 15: tracked, not saved
    assert best["class"] == [0.3, 0.3, 0.3] and abs(best["mean"] - 0.3) < 1e-12
    assert not train.should_checkpoint(14, [0.35, 0.3, 0.3], best, 120)
    assert train.should_checkpoint(15, [0.5, 0.4, 0.3], best, 120)           # epoch 15 may save (only `epoch < 15` is suppressed)
    assert not train.should_checkpoint(16, [0.5, 0.4, 0.3], best, 120)      # ties are not improvements
    assert not train.should_checkpoint(17, [0.4, 0.3, 0.2], best, 120)
    assert train.should_checkpoint(18, [0.4, 0.45, 0.2], best, 120)          # one class improved
    assert train.should_checkpoint(50, [0.1, 0.1, 0.1], best, 120)           # every 50 epochs
    assert train.should_checkpoint(120, [0.0, 0.0, 0.0], best, 120)          # last epoch
    best2 = {"class": [0.0, 0.0, 0.0], "mean": 0.0}
    assert train.should_checkpoint(100, [0.2, 0.2, 0.2], best2, 120) and best2["mean"] > 0.19   # bests updated on forced saves too


def test_synthetic_dataset_format():
    import saunet_amd
    from saunet_amd import train
    ds = train.SyntheticSlices(4, 64)
    b = train.collate([ds[0], ds[1]])
    assert b["image"].shape == (2, 3, 64, 64) and b["image"].dtype == torch.float32
    assert b["mask"][0].shape == (2, 64, 64) and b["mask"][0].dtype == torch.float64    # the loader hands float64 labels
    assert b["mask"][1].shape == (2, 1, 64, 64) and set(b["mask"][1].unique().tolist()) <= {0.0, 1.0}
    assert torch.equal(b["image"][:, 0], b["image"][:, 1])                                # one plane replicated x3
    assert abs(float(b["image"][0, 0].mean())) < 1e-4 and abs(float(b["image"][0, 0].std()) - 1) < 1e-2


def test_mask_to_edges_matches_loader_formulation():
    import saunet_amd
    from saunet_amd import data
    from oracle import saunet_ref as R
    from tests.golden_util import load
    g = load("loss.npz")
    assert (data.mask_to_edges(g["m2e_mask"]) == g["m2e_edge"]).all()          # fixture made by the REAL reference
    r = np.random.default_rng(1)
    m = r.integers(0, 4, (24, 40))
    assert (data.mask_to_edges(m) == R.mask_to_edges(m)).all()


def test_dense_block_refuses_batchnorm_layers_with_different_eps():
    """the fused dense block shares one normalisation (invstd per concat channel) between all its norm1 layers: layers whose NORM1 eps disagree
    must be refused before anything is launched (torchvision _DenseLayer, /root/reference/models/models.py:306-313, always agrees).  norm2
    normalises a layer's own 128 channels: its eps / momentum travel per layer and may differ -- such a block passes the guard (and,
    on this CPU-only box, stops at the device check instead)."""
    import pytest
    import saunet_amd
    from saunet_amd import functional as HF

    class Layer(torch.nn.Module):
        def __init__(self, cin, eps1, eps2):
            super().__init__()
            self.norm1 = torch.nn.BatchNorm2d(cin, eps=eps1); self.conv1 = torch.nn.Conv2d(cin, 128, 1, bias=False)
            self.norm2 = torch.nn.BatchNorm2d(128, eps=eps2); self.conv2 = torch.nn.Conv2d(128, 32, 3, padding=1, bias=False)
    with pytest.raises(RuntimeError, match="share eps"):
        HF.dense_block(torch.zeros(1, 64, 16, 16), [Layer(64, 1e-5, 1e-5), Layer(96, 1e-3, 1e-5)], True)
    with pytest.raises(RuntimeError) as e:
        HF.dense_block(torch.zeros(1, 64, 16, 16), [Layer(64, 1e-5, 1e-5), Layer(96, 1e-5, 1e-3)], True)
    assert "share eps" not in str(e.value)
