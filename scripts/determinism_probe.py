"""Where do two identical float32 runs of the HIP path start to differ?  Forward: per-module output difference (first modules above 1e-6 of
the tensor scale); backward: per-parameter gradient difference, largest first."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import saunet_amd
from oracle import saunet_ref as R, weights as Wt
spec = R.state_dict_spec()
sd = Wt.make_state_dict(spec, seed=5)
img, seg, edge = Wt.synthetic_batch(2, 64, 64, seed=41)
saunet_amd.set_compute_dtype(torch.float32)


def run():
    net = saunet_amd.SAUNet(num_classes=4).cuda()
    net.load_state_dict(sd, strict=False)
    saunet_amd.functional.notify_params_changed()
    sm = saunet_amd.SegmentationModule(saunet_amd.DualLoss(mode="train"), net, 4).train()
    outs = {}
    def hook(name):
        def f(m, i, o):
            t = o[0] if isinstance(o, (tuple, list)) else o
            if torch.is_tensor(t):
                outs[name] = t.detach().float().clone()
        return f
    hs = [m.register_forward_hook(hook(n)) for n, m in net.named_modules() if n]
    loss, _ = sm({"image": img.cuda(), "mask": (seg.cuda(), edge.cuda())}, 1)
    loss.backward(); torch.cuda.synchronize()
    for h in hs: h.remove()
    return outs, {k: v.grad.detach().clone() for k, v in net.named_parameters() if v.grad is not None}, float(loss)

o1, g1, l1 = run()
for trial in range(3):
    o2, g2, l2 = run()
    print("trial", trial, "loss", l1, l2)
    shown = 0
    for k in o1:
        if k in o2 and o1[k].shape == o2[k].shape:
            d = float((o1[k] - o2[k]).abs().max()) / max(float(o1[k].abs().max()), 1e-30)
            if d > 1e-6 and shown < 8:
                print("   fwd %-50s rel diff %.2e" % (k, d)); shown += 1
    gs = max(float(v.abs().max()) for v in g1.values())
    diffs = sorted(((float((g1[k] - g2[k]).abs().max()) / gs, k) for k in g1), reverse=True)
    print("   bwd top:", ", ".join("%s %.1e" % (k, d) for d, k in diffs[:6]))
