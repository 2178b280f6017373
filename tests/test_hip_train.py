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
