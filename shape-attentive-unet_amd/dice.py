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


def seeded_state_dict(seed, num_classes=4):
    """Deterministic SAUNet weights from (seed, parameter name): numpy PCG64 seeded with [seed, crc32(name)], He-style fan-in scaling for
    convolutions, non-trivial BatchNorm parameters and running statistics.  The SAME generator produced the initialisation of the reference
    runs behind tests/golden/dice_ref.npz (oracle/make_golden_dice.py uses oracle.weights.make_state_dict; tests/test_hip_dice.py asserts that
    the two agree bit for bit), so a HIP run started from it sees exactly the reference's weights and mini-batches."""
    import zlib
    net = M.SAUNet(num_classes=num_classes)
    kinds = {}
    for mname, m in net.named_modules():
        pre = mname + "." if mname else ""
        if isinstance(m, torch.nn.modules.batchnorm._BatchNorm):
            kinds[pre + "weight"] = "gamma"; kinds[pre + "bias"] = "beta"; kinds[pre + "running_mean"] = "rmean"; kinds[pre + "running_var"] = "rvar"
        elif isinstance(m, (torch.nn.modules.conv._ConvNd, torch.nn.Linear)):
            kinds[pre + "weight"] = "conv"
            if m.bias is not None:
                kinds[pre + "bias"] = "bias"
    out, seen = {}, set()
    for key, v in net.state_dict().items():
        if v.data_ptr() in seen and v.numel() > 0:
            continue                         # an alias of a tensor that already has its (first, torchvision-style) name: conv1.0 = encoder.features.conv0, ...
        seen.add(v.data_ptr())
        kind, shape = kinds.get(key), tuple(v.shape)
        if kind is None:
            out[key] = v.clone()             # counters (num_batches_tracked, _running_iter, ...)
            continue
        r = np.random.default_rng([seed, zlib.crc32(key.encode())])
        if kind == "conv":
            fan_in = int(np.prod(shape[1:])) if len(shape) > 1 else 1
            if key.endswith("mrf.up.0.weight") or key == "dec1.block.1.weight":      # ConvTranspose2d [Cin, Cout, 4, 4]: 2 x 2 taps reach an output pixel
                fan_in = shape[0] * 4
            a = r.standard_normal(shape, dtype=np.float32) * np.float32(np.sqrt(2.0 / max(fan_in, 1)))
        elif kind == "bias":
            a = r.standard_normal(shape, dtype=np.float32) * np.float32(0.05)
        elif kind == "gamma":
            a = r.uniform(0.6, 1.4, shape).astype(np.float32)
        elif kind in ("beta", "rmean"):
            a = r.standard_normal(shape, dtype=np.float32) * np.float32(0.1)
        else:
            a = r.uniform(0.5, 1.5, shape).astype(np.float32)
        out[key] = torch.from_numpy(np.ascontiguousarray(a))
    return out


def paired_study(seeds=(304, 305, 306, 307, 308), ref_npz=None, steps=300, size=128, batch=8, pool=64, eval_n=512, lr=2e-3, device="cuda"):
    """The metric's second half as a STATISTIC: for every seed the HIP path is trained in float32 and in bf16 storage from seeded_state_dict(seed)
    on the mini-batch schedule of `run` and evaluated on `eval_n` held-out phantoms; `ref_npz` (tests/golden/dice_ref.npz: the REAL reference
    trained on the CPU from the same weights, batches and optimiser, /root/reference/train.py:25-64, 95-106, radam.py) supplies the third arm.
    Returns per-seed per-class Dice of each arm and, for every pair of arms, the paired mean difference over seeds with its 95 % confidence
    half-width (Student t) -- training is chaotic (two float32 runs from weights 1e-6 apart drift percent-level in Dice on a small validation
    set), so single-seed differences mean nothing; the paired mean over seeds and a 512-slice validation set does."""
    tcrit = {2: 12.706, 3: 4.303, 4: 3.182, 5: 2.776, 6: 2.571, 7: 2.447, 8: 2.365}
    ref = None
    if ref_npz is not None:
        z = np.load(ref_npz)
        if int(z["steps"]) == steps and int(z["size"]) == size and int(z["batch"]) == batch and int(z["pool"]) == pool and int(z["eval_n"]) == eval_n:
            ref = {int(s_): (z["dice"][i], z["loss_curve"][i]) for i, s_ in enumerate(z["seeds"])}
    rows = []
    for sd_ in seeds:
        res = run(size=size, batch=batch, steps=steps, pool=pool, eval_n=eval_n, seed=sd_, optimizer="radam", lr=lr, device=device,
                  state_dict=seeded_state_dict(sd_))
        row = {"seed": int(sd_), "f32": res["f32"]["dice"], "bf16": res["bf16"]["dice"], "loss_first_f32": res["f32"]["loss_curve"][:5],
               "loss_last_f32": res["f32"]["loss_last"], "loss_last_bf16": res["bf16"]["loss_last"]}
        if ref is not None and int(sd_) in ref:
            row["ref"] = [round(float(v), 5) for v in ref[int(sd_)][0]]
            row["loss_first_ref"] = [round(float(v), 5) for v in ref[int(sd_)][1][:5]]
            row["loss_last_ref"] = round(float(np.mean(ref[int(sd_)][1][-10:])), 5)
        rows.append(row)
    out = {"protocol": "%d RAdam steps (lr %g, no weight decay) from seeded weights on %d synthetic %dx%d phantoms, batch %d; hard Dice of RV / MYO / LV on %d "
                       "held-out phantoms (argmax of the scores, intersection / union histograms: train.py:25-64); arms: HIP float32, HIP bf16 storage, "
                       "REFERENCE (CPU, tests/golden/dice_ref.npz); paired over seeds" % (steps, lr, pool, size, size, batch, eval_n),
           "seeds": [int(s_) for s_ in seeds], "rows": rows, "pairs": {}}
    arms = ["f32", "bf16"] + (["ref"] if all("ref" in r for r in rows) else [])
    for a_ in arms:
        out["mean_dice_" + a_] = round(float(np.mean([np.mean(r[a_]) for r in rows])), 5)
    n = len(rows)
    for a_, b_ in (("f32", "ref"), ("bf16", "ref"), ("bf16", "f32")):
        if a_ not in arms or b_ not in arms:
            continue
        d = np.array([np.array(r[a_]) - np.array(r[b_]) for r in rows])             # [seeds, 3 classes]
        dm = d.mean(1)                                                               # per-seed mean over classes
        hw = (tcrit.get(n, 2.0) * dm.std(ddof=1) / np.sqrt(n)) if n > 1 else float("nan")
        out["pairs"]["%s_minus_%s" % (a_, b_)] = {"mean": round(float(dm.mean()), 5), "ci95_halfwidth": round(float(hw), 5),
                                                  "per_class_mean": [round(float(v), 5) for v in d.mean(0)], "max_abs_single": round(float(np.abs(d).max()), 5)}
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
