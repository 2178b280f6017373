"""Trained-weights Dice experiment: the second half of the headline metric ("val Dice vs ref").

From ONE seeded initialisation the SAUNet is trained for N steps on synthetic ACDC-like phantoms (data.synthetic_batch) once per storage
dtype (float32 = the precision at which the 1e-3 / Dice-1e-4 parity with the reference's CPU path holds, bfloat16 = the benchmarked
precision), then evaluated on HELD-OUT phantoms with the reference's validation protocol: argmax of the softmax scores, per-class
intersection / union histograms over all validation pixels (/root/reference/train.py:25-64, utils.py:119-140 `intersectionAndUnion`),
hard Dice_c = 2 I_c / (|P_c| + |Y_c|) for the three foreground classes (train.dice_from_iu).  Optimiser recipe = the reference's train.sh
(RAdam, no weight decay); the learning rate is raised so that a few hundred steps suffice on the phantoms.

    python -m saunet_amd.dice [--size 128 --batch 8 --steps 300 ...]      -> one JSON line

The float32 run of the HIP path is anchored to the CPU oracle by tests/test_hip_dice.py (first K steps, same data, loss curve within the
float32 parity bound); this module itself never touches the oracle.
"""
import argparse
import json
import time

import numpy as np
import torch

from . import data as sdata
from . import functional as HF
from . import modules as M
from . import optim


def _iu(label, seg, num_class):
    """per-class intersection / union pixel counts of a batch of label maps (device tensors) -> two float64 numpy vectors"""
    inter = torch.zeros(num_class, dtype=torch.float64, device=label.device)
    union = torch.zeros(num_class, dtype=torch.float64, device=label.device)
    for c in range(num_class):
        p, y = label == c, seg == c
        inter[c] = (p & y).sum()
        union[c] = (p | y).sum()
    return inter.cpu().numpy(), union.cpu().numpy()


def evaluate(net, images, segs, chunk=8):
    """hard Dice / IoU per foreground class of `net` (eval mode, BatchNorm folded) on held-out slices"""
    net.eval()
    nc = net.num_classes
    inter, union = np.zeros(nc), np.zeros(nc)
    with torch.no_grad():
        for i in range(0, images.shape[0], chunk):
            logits, _ = net(images[i:i + chunk])
            _, label = HF.softmax_argmax(logits, want_prob=False)
            a, u = _iu(label, segs[i:i + chunk], nc)
            inter += a; union += u
    net.train()
    dice = 2 * inter / (union + inter + 1e-10)
    return dice[1:], (inter / (union + 1e-10))[1:]


def run(size=128, batch=8, steps=300, pool=64, eval_n=32, seed=304, optimizer="radam", lr=2e-3, dtypes=("f32", "bf16"), device="cuda",
        state_dict=None, record_every=1, noise_floor=False):
    """-> {"f32": {...}, "bf16": {...}, "delta": {...}}.  Both runs start from the same weights (``state_dict`` or a seeded initialisation)
    and see the same mini-batches: pool slices seed .. seed+pool-1 cycled in order; held-out slices come from a disjoint seed range.
    noise_floor: a third run "f32p" = float32 from the same weights perturbed by 1e-6 relative noise -- training is chaotic, so this shows how
    far two float32 runs drift apart on their own: the yard-stick for the bf16 - f32 Dice difference."""
    dev = torch.device(device)
    prev = M.get_compute_dtype()
    M.set_compute_dtype(torch.float32)
    torch.manual_seed(seed)
    init = state_dict if state_dict is not None else {k: v.clone() for k, v in M.SAUNet(num_classes=4).state_dict().items()}
    img, seg, edge = sdata.synthetic_batch(pool, size, size, seed=seed)
    vimg, vseg, _ = sdata.synthetic_batch(eval_n, size, size, seed=seed + 100003)
    img, seg, edge, vimg, vseg = img.to(dev), seg.to(dev), edge.to(dev), vimg.to(dev), vseg.to(dev)
    out = {"config": {"size": size, "batch": batch, "steps": steps, "train_slices": pool, "heldout_slices": eval_n, "optimizer": optimizer, "lr": lr,
                      "seed": seed, "data": "synthetic ellipse phantoms (saunet_amd.data.synthetic_batch), held-out seeds disjoint from the training pool"}}
    try:
        for name in tuple(dtypes) + (("f32p",) if noise_floor else ()):
            dtype = torch.bfloat16 if name == "bf16" else torch.float32
            M.set_compute_dtype(dtype)
            net = M.SAUNet(num_classes=4)
            net.load_state_dict(init, strict=False)
            if name == "f32p":
                g = torch.Generator().manual_seed(seed + 1)
                with torch.no_grad():
                    for p_ in net.parameters():
                        p_.mul_(1.0 + 1e-6 * torch.randn(p_.shape, generator=g))
            net = net.to(dev)
            sm = M.SegmentationModule(M.DualLoss(mode="train"), net, 4).train()
            opts = optim.create_optimizers(net, optimizer, lr=lr, momentum=0.9, weight_decay=1e-4)
            losses = []
            t0 = time.time()
            for it in range(steps):
                lo = (it * batch) % pool
                idx = torch.arange(lo, lo + batch, device=dev) % pool
                feed = {"image": img[idx], "mask": (seg[idx], edge[idx])}
                sm.zero_grad(set_to_none=True)
                loss, _ = sm(feed, 1)
                loss.backward()
                for o in opts:
                    o.step()
                if it % record_every == 0 or it == steps - 1:
                    losses.append(loss.detach())
            torch.cuda.synchronize(dev)
            train_s = time.time() - t0
            dice, iou = evaluate(net, vimg, vseg)
            curve = [float(v) for v in torch.stack(losses).float().cpu()]
            out[name] = {"dice": [round(float(d), 5) for d in dice], "mean_dice": round(float(dice.mean()), 5), "iou": [round(float(v), 5) for v in iou],
                         "loss_first": round(curve[0], 5), "loss_last": round(float(np.mean(curve[-10:])), 5), "train_seconds": round(train_s, 2),
                         "loss_curve": [round(v, 5) for v in curve]}
            del net, sm, opts
            HF.notify_params_changed()
            torch.cuda.empty_cache()
        if "f32" in out and "bf16" in out:
            d = np.array(out["bf16"]["dice"]) - np.array(out["f32"]["dice"])
            a, b = np.array(out["f32"]["loss_curve"]), np.array(out["bf16"]["loss_curve"])
            k = max(1, len(a) // 10)          # loss-curve distance on 10 % windows (single steps are noisy under different rounding)
            wa = np.array([a[i:i + k].mean() for i in range(0, len(a) - k + 1, k)]); wb = np.array([b[i:i + k].mean() for i in range(0, len(b) - k + 1, k)])
            out["delta"] = {"dice_bf16_minus_f32": [round(float(v), 5) for v in d], "max_abs_dice_delta": round(float(np.abs(d).max()), 5),
                            "mean_dice_delta": round(float(d.mean()), 5),
                            "loss_curve_rel_distance": round(float(np.abs(wa - wb).max() / max(abs(float(a[0])), 1e-12)), 5)}
            if "f32p" in out:
                dp_ = np.array(out["f32p"]["dice"]) - np.array(out["f32"]["dice"])
                out["delta"]["f32_noise_floor_max_abs_dice_delta"] = round(float(np.abs(dp_).max()), 5)
                out["delta"]["f32_noise_floor_mean_dice_delta"] = round(float(dp_.mean()), 5)
    finally:
        M.set_compute_dtype(prev)
    return out


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__.splitlines()[0])
    ap.add_argument("--size", type=int, default=128)
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--pool", type=int, default=64)
    ap.add_argument("--eval-n", type=int, default=32)
    ap.add_argument("--seed", type=int, default=304)
    ap.add_argument("--optimizer", default="radam")
    ap.add_argument("--lr", type=float, default=2e-3)
    ap.add_argument("--noise-floor", action="store_true", help="add a float32 run from 1e-6-perturbed weights (run-to-run drift yard-stick)")
    ap.add_argument("--curves", action="store_true", help="keep the full loss curves in the output")
    a = ap.parse_args(argv)
    res = run(a.size, a.batch, a.steps, a.pool, a.eval_n, a.seed, a.optimizer, a.lr, noise_floor=a.noise_floor)
    if not a.curves:
        for k in [k for k in ("f32", "bf16", "f32p") if k in res]:
            c = res[k].pop("loss_curve")
            res[k]["loss_curve_every_10pct"] = [round(float(np.mean(c[i:i + max(1, len(c) // 10)])), 5) for i in range(0, len(c), max(1, len(c) // 10))]
    print(json.dumps(res))


if __name__ == "__main__":
    main()
