"""tests/golden/dice_ref.npz: the REAL reference trained on the phantom task -- CONTAINER-ONLY (imports /root/reference via ref_import).

    PYTHONDONTWRITEBYTECODE=1 python -m oracle.make_golden_dice [--seeds 304 305 306 307 308] [--steps 300]

The metric's second half is "val Dice vs ref" (BASELINE.json).  For every seed s this script
  * builds the reference SAUNet (models/models.py:264-394) with the deterministic weights oracle.weights.make_state_dict(spec, seed=s),
  * trains it with the reference's own RAdam (radam.py:5-78; train.py:197-201: group_weight, betas (0.9, 0.999), no weight decay) for `steps`
    iterations of SegmentationModule(...)(feed, epoch) -> loss.backward() -> step() (train.py:95-106) on the mini-batch schedule of
    saunet_amd.dice.run: pool = phantoms s .. s+63 at 128 x 128, batch 8, slices (it*8 + j) % 64,
  * evaluates it the way train.py:25-64 does -- eval mode, argmax of the scores, utils.intersectionAndUnion histograms summed over the
    validation set -- on 512 held-out phantoms (seeds s + 100003 ...), hard Dice_c = 2 I_c / (|P_c| + |Y_c|) for RV / MYO / LV,
and stores per seed: the loss curve, per-class Dice and IoU.  tests/test_hip_dice.py trains the HIP path from the SAME weights on the SAME
batches and compares against these numbers (data only travels: no reference source or bytecode)."""
import argparse
import contextlib
import io
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import ref_import, saunet_ref as R, weights as Wt  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden", "dice_ref.npz")


def group_weight_like_reference(ns, net):
    """train.py:166-185 restated with the reference's module classes (train.py itself cannot be imported on this torch: SURVEY 8c)"""
    import torch.nn as nn
    decay, no_decay = [], []
    for m in net.modules():
        if isinstance(m, nn.Linear) or isinstance(m, nn.modules.conv._ConvNd):
            decay.append(m.weight)
            if m.bias is not None:
                no_decay.append(m.bias)
        elif isinstance(m, nn.modules.batchnorm._BatchNorm):
            if m.weight is not None:
                no_decay.append(m.weight)
            if m.bias is not None:
                no_decay.append(m.bias)
    return [dict(params=decay), dict(params=no_decay, weight_decay=.0)]


def run_seed(ns, seed, size, batch, steps, pool, eval_n, lr):
    spec = R.state_dict_spec()
    sd = Wt.make_state_dict(spec, seed=seed)
    with contextlib.redirect_stdout(io.StringIO()):
        net = ns.SAUNet(num_classes=4)
    res = net.load_state_dict(sd, strict=False)
    assert not res.unexpected_keys
    sm = ns.SegmentationModule(ns.DualLoss(mode="train"), net, 4)
    sm.train()
    opt = ns.RAdam(group_weight_like_reference(ns, net), lr=lr, betas=(0.9, 0.999))
    img, seg, edge = Wt.synthetic_batch(pool, size, size, seed=seed)
    curve = []
    t0 = time.time()
    for it in range(steps):
        idx = [(it * batch + j) % pool for j in range(batch)]
        sm.zero_grad()
        loss, _ = sm({"image": img[idx], "mask": (seg[idx].double(), edge[idx])}, 1)
        loss.backward()
        opt.step()
        curve.append(float(loss))
        if it % 25 == 0:
            print("seed %d it %3d loss %.4f  (%.0f s)" % (seed, it, curve[-1], time.time() - t0), flush=True)
    vimg, vseg, _ = Wt.synthetic_batch(eval_n, size, size, seed=seed + 100003)
    net.eval()
    inter, union = np.zeros(4), np.zeros(4)
    with torch.no_grad():
        for i in range(0, eval_n, 16):
            logits, _ = net(vimg[i:i + 16])
            pred = logits.argmax(1).numpy()
            for j in range(pred.shape[0]):
                a, u = ns.intersectionAndUnion(pred[j], vseg[i + j].numpy(), 4)
                inter += a; union += u
    dice = 2 * inter / (union + inter + 1e-10)
    iou = inter / (union + 1e-10)
    return np.array(curve), dice[1:], iou[1:]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seeds", type=int, nargs="+", default=[304, 305, 306, 307, 308])
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--size", type=int, default=128)
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--pool", type=int, default=64)
    ap.add_argument("--eval-n", type=int, default=512)
    ap.add_argument("--lr", type=float, default=2e-3)
    ap.add_argument("--threads", type=int, default=8)
    a = ap.parse_args()
    torch.set_num_threads(a.threads)
    ns = ref_import.load()
    out = {"seeds": [], "dice": [], "iou": [], "loss_curve": []}
    if os.path.exists(GOLD):
        z = np.load(GOLD)
        if int(z["steps"]) == a.steps and int(z["size"]) == a.size:
            out = {k: list(z[k]) for k in out}
    for s in a.seeds:
        if s in [int(v) for v in out["seeds"]]:
            continue
        curve, dice, iou = run_seed(ns, s, a.size, a.batch, a.steps, a.pool, a.eval_n, a.lr)
        out["seeds"].append(s); out["dice"].append(dice); out["iou"].append(iou); out["loss_curve"].append(curve)
        print("seed %d: dice %s  loss %.4f -> %.4f" % (s, np.round(dice, 4), curve[0], np.mean(curve[-10:])), flush=True)
        np.savez_compressed(GOLD, seeds=np.array(out["seeds"], np.int64), dice=np.array(out["dice"]), iou=np.array(out["iou"]),
                            loss_curve=np.array(out["loss_curve"], np.float32), steps=np.int64(a.steps), size=np.int64(a.size), batch=np.int64(a.batch),
                            pool=np.int64(a.pool), eval_n=np.int64(a.eval_n), lr=np.float64(a.lr))


if __name__ == "__main__":
    main()
