"""Run-to-run spread of the float32 whole-network gradient parity (tests/test_hip_saunet.py::test_other_shapes_against_oracle and the
512^2 / fixture cases): max |grad - oracle| / grad scale over all parameters, several HIP runs per case."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
from test_hip_saunet import make_net, R, Wt
for (B, H, W, seed) in [(1, 64, 96, 7), (3, 64, 64, 9), (2, 128, 128, 11)]:
    S, spec, sd, net, sm = make_net(seed)
    img, seg, edge = Wt.synthetic_batch(B, H, W, seed=100 + seed)
    sdo = {k: v.clone() for k, v in sd.items()}
    keys = Wt.trainable_keys(spec)
    for k in keys:
        sdo[k].requires_grad_(True)
    loss_o, _, _, _ = R.segmentation_step(sdo, img, seg, edge, True)
    loss_o.backward()
    gmax = max(float(sdo[k].grad.abs().max()) for k in keys)
    out = []
    for run in range(8):
        S, spec, sd, net, sm = make_net(seed)
        sm.train()
        loss, _ = sm({"image": img.cuda(), "mask": (seg.cuda(), edge.cuda())}, 1)
        loss.backward(); torch.cuda.synchronize()
        pd = dict(net.named_parameters())
        e, k = max((float((pd[k].grad.cpu() - sdo[k].grad).abs().max()), k) for k in keys)
        out.append("%.2e" % (e / gmax))
    print((B, H, W), "max err / grad scale per run:", " ".join(out), flush=True)
