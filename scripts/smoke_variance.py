"""Run-to-run spread of the smoke() comparison (float32, 2 x 64 x 64): which parameter carries the largest gradient error, per run."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import saunet_amd
from oracle import saunet_ref as R, weights as Wt
spec = R.state_dict_spec()
sd = Wt.make_state_dict(spec, seed=5)
img, seg, edge = Wt.synthetic_batch(2, 64, 64, seed=41)
sdo = {k: v.clone() for k, v in sd.items()}
keys = Wt.trainable_keys(spec)
for k in keys:
    sdo[k].requires_grad_(True)
loss_o, _, _, _ = R.segmentation_step(sdo, img, seg, edge, True)
loss_o.backward()
gmax = max(float(sdo[k].grad.abs().max()) for k in keys)
saunet_amd.set_compute_dtype(torch.float32)
prev = None
for run in range(8):
    net = saunet_amd.SAUNet(num_classes=4).cuda()
    net.load_state_dict(sd, strict=False)
    saunet_amd.functional.notify_params_changed()
    sm = saunet_amd.SegmentationModule(saunet_amd.DualLoss(mode="train"), net, 4).train()
    loss, _ = sm({"image": img.cuda(), "mask": (seg.cuda(), edge.cuda())}, 1)
    loss.backward(); torch.cuda.synchronize()
    pd = dict(net.named_parameters())
    errs = sorted(((float((pd[k].grad.cpu() - sdo[k].grad).abs().max()), k) for k in keys), reverse=True)
    cur = {k: pd[k].grad.detach().cpu().clone() for k in keys}
    drift = max(float((cur[k] - prev[k]).abs().max()) for k in keys) if prev else 0.0
    prev = cur
    print("run %d  loss %.7f  top errors: %s   run-to-run max diff %.2e" % (run, float(loss), ", ".join("%s %.2e" % (k, e / gmax) for e, k in errs[:3]), drift / gmax), flush=True)
