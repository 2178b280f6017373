"""Diagnostic (GPU box): bf16-storage HIP path vs the fp32 CPU oracle at the benchmarked geometry (256x256).
Prints per-parameter gradient cosine / rel-L2 (worst first), the loss difference and, after N identical SGD steps on one fixed
batch, the hard Dice per class of both models.  Usage: python scripts/bf16_parity_diag.py [B] [steps] [dtype]"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import saunet_amd as S                                   # noqa: E402
from saunet_amd import train as T                        # noqa: E402
from oracle import saunet_ref as R, weights as Wt        # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
STEPS = int(sys.argv[2]) if len(sys.argv) > 2 else 10
DT = {"bf16": torch.bfloat16, "f32": torch.float32}[sys.argv[3] if len(sys.argv) > 3 else "bf16"]
H = 256
torch.set_num_threads(min(os.cpu_count() or 8, 32))
spec = R.state_dict_spec()
sd = Wt.make_state_dict(spec, 13)
keys = Wt.trainable_keys(spec)
img, seg, edge = Wt.synthetic_batch(B, H, H, seed=113)

t0 = time.time()
sdo = {k: v.clone() for k, v in sd.items()}
for k in keys:
    sdo[k].requires_grad_(True)
loss_o, acc_o, lg_o, _ = R.segmentation_step(sdo, img, seg, edge, True)
loss_o.backward()
print("oracle fwd+bwd %.1fs loss %.6f" % (time.time() - t0, float(loss_o)), flush=True)

S.set_compute_dtype(DT)
net = S.SAUNet(num_classes=4).cuda()
net.load_state_dict(sd, strict=False)
sm = S.SegmentationModule(S.DualLoss(mode="train"), net, 4).train()
feed = {"image": img.cuda(), "mask": (seg.cuda(), edge.cuda())}
loss, _ = sm(feed, 1)
loss.backward()
torch.cuda.synchronize()
print("hip %s loss %.6f  |d| %.3e" % (DT, float(loss), abs(float(loss) - float(loss_o))), flush=True)
pd = dict(net.named_parameters())
rows = []
for k in keys:
    a = pd[k].grad.detach().double().cpu().reshape(-1); b = sdo[k].grad.double().reshape(-1)
    nb = float(b.norm())
    cos = float((a @ b) / (a.norm() * b.norm() + 1e-300))
    rel = float((a - b).norm() / (nb + 1e-300))
    rows.append((cos, rel, nb, k))
rows.sort()
print("worst 25 by cosine:")
for cos, rel, nb, k in rows[:25]:
    print("  cos %.5f  relL2 %.4f  |g| %.3e  %s" % (cos, rel, nb, k))
rels = np.array([r[1] for r in rows]); coss = np.array([r[0] for r in rows])
print("summary: min cos %.5f  median cos %.6f  max relL2 %.4f  median relL2 %.4f" % (coss.min(), np.median(coss), rels.max(), np.median(rels)))
for grp in ("encoder.features.denseblock1", "encoder.features.denseblock2", "encoder.features.denseblock3", "encoder.features.denseblock4",
            "encoder.features.conv0", "res", "gate", "dec5", "dec4", "dec3", "dec2", "dec1", "dec0", "center", "final", "d0", "c3", "c4", "c5"):
    sel = [r for r in rows if r[3].startswith(grp)]
    if sel:
        print("  %-32s n=%3d  min cos %.5f  max relL2 %.4f" % (grp, len(sel), min(r[0] for r in sel), max(r[1] for r in sel)))

if STEPS > 0:
    # N identical SGD steps on the fixed batch in both, then hard Dice per class (eval mode, argmax) against the labels
    LR = 2e-3
    sdo = {k: v.clone() for k, v in sd.items()}
    for k in keys:
        sdo[k].requires_grad_(True)
    decay = [sdo[k] for k, _, kind in spec if kind == "conv"]
    rest = [sdo[k] for k, _, kind in spec if kind in ("bias", "gamma", "beta")]
    opt_o = torch.optim.SGD([dict(params=decay), dict(params=rest, weight_decay=0.0)], lr=LR, momentum=0.9, weight_decay=1e-4)
    net.load_state_dict(sd, strict=False)
    S.functional.notify_params_changed()
    opt = S.optim.create_optimizers(net, "sgd", lr=LR, momentum=0.9, weight_decay=1e-4)[0]
    lo, lh = [], []
    t0 = time.time()
    for it in range(STEPS):
        opt_o.zero_grad()
        l, *_ = R.segmentation_step(sdo, img, seg, edge, True)
        l.backward(); opt_o.step(); lo.append(float(l))
        sm.train(); sm.zero_grad(set_to_none=True)
        l2, _ = sm(feed, 1)
        l2.backward(); opt.step(); lh.append(float(l2))
    print("train %d steps %.1fs" % (STEPS, time.time() - t0))
    print("oracle losses", np.round(lo, 4))
    print("hip    losses", np.round(lh, 4))
    with torch.no_grad():
        lg_o, _ = R.saunet_forward({k: v.detach() for k, v in sdo.items()}, img, False)
        sm.eval()
        lg_h, _ = net(img.cuda())
    po, ph = lg_o.argmax(1).numpy(), lg_h.float().argmax(1).cpu().numpy()
    io, uo = T.intersection_and_union(po, seg.numpy(), 4); ih, uh = T.intersection_and_union(ph, seg.numpy(), 4)
    do, dh = T.dice_from_iu(io, uo), T.dice_from_iu(ih, uh)
    print("dice oracle", np.round(do, 5), " hip", np.round(dh, 5), " |d|", np.abs(do - dh).max())
    print("argmax disagreement: %d of %d pixels" % ((po != ph).sum(), po.size))
    # soft dice (the loss term's definition) is less sensitive to ties
    print("eval logits max abs diff %.4f (scale %.3f)" % (float((lg_o - lg_h.float().cpu()).abs().max()), float(lg_o.abs().max())))
