"""Training / evaluation loop for the HIP SAUNet path -- the counterpart of the reference's train.py.

Mirrors /root/reference/train.py: per-iteration order ``zero_grad -> forward -> loss.mean() -> backward -> step``
(:93-106), parameter grouping (:166-185), SGD / RAdam construction (:187-207), cosine LR per EPOCH (:210-216,
applied after each epoch :150), resume scaling (:84-88), validation IoU from ``intersectionAndUnion`` over
argmax(softmax) (:25-64, utils.py:119-140) and the checkpoint policy (:153-163, :294-329).  Differences, all
deliberate: one process per GPU with RCCL gradient averaging instead of single-process nn.DataParallel
(saunet_amd/dp.py), logging scalars are read back every ``--disp_iter`` iterations instead of five ``.item()``
syncs per iteration, and a synthetic phantom dataset stands in when no ACDC volumes are mounted.

    python -m saunet_amd.train --num_epoch 2 --batch_size_per_gpu 8 --synthetic 64
    python -m torch.distributed.run --nproc-per-node 8 -m saunet_amd.train ...
"""
import argparse
import math
import os
import time

import numpy as np
import torch

from . import data as sdata
from . import dp, optim
from .modules import SAUNet, SegmentationModule, DualLoss, set_compute_dtype


# ------------------------------------------------------------------------------------------------ metrics
def intersection_and_union(pred, label, num_class):
    """utils.intersectionAndUnion: labels shifted by one, histograms over [1, num_class]."""
    pred = np.asarray(pred).copy() + 1
    label = np.asarray(label).copy() + 1
    pred = pred * (label > 0)
    inter = pred * (pred == label)
    ai, _ = np.histogram(inter, bins=num_class, range=(1, num_class))
    ap, _ = np.histogram(pred, bins=num_class, range=(1, num_class))
    al, _ = np.histogram(label, bins=num_class, range=(1, num_class))
    return ai, ap + al - ai


def dice_from_iu(inter, union):
    """Dice_c = 2 I_c / (U_c + I_c): the hard Dice derived from the same histograms (the reference only prints IoU)."""
    inter = np.asarray(inter, np.float64); union = np.asarray(union, np.float64)
    return 2 * inter / (union + inter + 1e-10)


class AverageMeter:
    def __init__(self):
        self.sum, self.count = 0.0, 0

    def update(self, v, n=1):
        self.sum = self.sum + v * n
        self.count += n

    def average(self):
        return self.sum / max(self.count, 1)


# ------------------------------------------------------------------------------------------------ data
class SyntheticSlices(torch.utils.data.Dataset):
    """Deterministic ellipse-phantom slices in the loader's format (image [3,H,W], mask (seg [H,W], edge [1,H,W]))."""

    def __init__(self, n, size=256, seed=304):
        self.n, self.size, self.seed = n, size, seed

    def __len__(self):
        return self.n

    def __getitem__(self, i):
        img, seg, edge = sdata.synthetic_batch(1, self.size, self.size, seed=self.seed + i)
        return {"image": img[0], "mask": (seg[0].double(), edge[0])}


def collate(batch):
    return {"image": torch.stack([b["image"] for b in batch]),
            "mask": (torch.stack([b["mask"][0] for b in batch]), torch.stack([b["mask"][1] for b in batch]))}


class RawSyntheticSlices(torch.utils.data.Dataset):
    """Un-augmented phantom slices of RAGGED sizes with raw (scanner-like, non-negative) intensities: what AC17Data holds after the 1.25 mm
    re-scaling and before its per-slice augmentation chain (data/ac17_dataloader.py:133-150) -- the input of augment.DeviceAugmenter."""

    def __init__(self, n, size=256, seed=304):
        self.n, self.size, self.seed = n, size, seed

    def __len__(self):
        return self.n

    def __getitem__(self, i):
        r = np.random.default_rng(self.seed + 7919 * i)
        h, w = int(self.size * r.uniform(0.8, 1.25)), int(self.size * r.uniform(0.8, 1.25))
        img, seg, _ = sdata.synthetic_batch(1, h, w, seed=self.seed + i)
        raw = img[0, 0].numpy()
        raw = (raw - raw.min()) * 300.0
        return {"raw": raw.astype(np.float32), "seg": seg[0].numpy().astype(np.float32)}


def collate_raw(batch):
    return {"raw": [b["raw"] for b in batch], "seg": [b["seg"] for b in batch]}


# ------------------------------------------------------------------------------------------------ loops
def poly_resume_lr(lr0, epoch, num_epoch, lr_pow=0.9):
    return lr0 * ((1.0 - float(epoch - 1) / num_epoch) ** lr_pow)   # train.py:84-88


def train_one_epoch(sm, loader, optimizers, epoch, args, history, buckets=None, device="cuda"):
    meters = {k: AverageMeter() for k in ("loss", "acc", "j1", "j2", "j3", "batch_time", "data_time")}
    sm.train(not args.fix_bn)
    if epoch == args.start_epoch and args.start_epoch > 1:
        args.running_lr_encoder = poly_resume_lr(args.lr_encoder, epoch, args.num_epoch, args.lr_pow)
        for g in optimizers[0].param_groups:
            g["lr"] = args.running_lr_encoder
    tic = time.time()
    pending = []
    for it, batch in enumerate(loader):
        meters["data_time"].update(time.time() - tic)
        if "raw" in batch:      # --augment: crop/pad, flips, rotation, gamma, z-score, elastic deformation and edge maps on the device
            feed = args.augmenter(batch["raw"], batch["seg"])
        else:
            feed = {"image": batch["image"].to(device, non_blocking=True),
                    "mask": (batch["mask"][0].to(device, non_blocking=True), batch["mask"][1].to(device, non_blocking=True))}
        sm.zero_grad(set_to_none=True)
        loss, (acc, jac) = sm(feed, epoch)
        loss = loss.mean()
        loss.backward()
        if buckets is not None:
            buckets.finish()
        for opt in optimizers:
            opt.step()
        pending.append(torch.stack([loss.detach(), acc, jac[0], jac[1], jac[2]]))
        meters["batch_time"].update(time.time() - tic)
        tic = time.time()
        if (it + 1) % args.disp_iter == 0 or it + 1 == len(loader):
            vals = dp.all_reduce_scalars(torch.stack(pending).mean(0)).tolist()   # ONE device->host sync per disp_iter
            n = len(pending); pending = []
            for k, v in zip(("loss", "acc", "j1", "j2", "j3"), vals):
                meters[k].update(v, n)
            if args.rank == 0:
                print("Epoch: [{}][{}/{}], Time: {:.3f}, Data: {:.3f}, lr_unet: {:.6f}, Accuracy: {:4.2f}, Loss: {:.6f}, "
                      "Jaccard: [{:4.2f} {:4.2f} {:4.2f}]".format(
                          epoch, it + 1, len(loader), meters["batch_time"].average(), meters["data_time"].average(),
                          args.running_lr_encoder, meters["acc"].average() * 100, meters["loss"].average(),
                          meters["j1"].average() * 100, meters["j2"].average() * 100, meters["j3"].average() * 100), flush=True)
    history["train"]["epoch"].append(epoch)
    history["train"]["loss"].append(meters["loss"].average())
    history["train"]["acc"].append(meters["acc"].average())
    history["train"]["jaccard"].append((meters["j1"].average() + meters["j2"].average() + meters["j3"].average()) / 3)
    args.running_lr_encoder = optim.adjust_learning_rate(optimizers, epoch, args.lr_encoder, args.num_epoch)
    return meters


@torch.no_grad()
def evaluate(sm, dataset, args, device="cuda"):
    """train.py:25-64 -- batch size 1, inference branch, IoU over classes 1..3 (+ derived Dice)."""
    sm.eval()
    inter, union, losses = np.zeros(args.num_class), np.zeros(args.num_class), []
    for i in range(len(dataset)):
        s = dataset[i]
        seg = s["mask"][0]
        feed = {"image": s["image"].unsqueeze(0).to(device), "mask": (seg.to(device), s["mask"][1].to(device))}
        scores, loss = sm(feed, epoch=0, segSize=tuple(seg.shape))
        pred = scores.argmax(1).squeeze(0).cpu().numpy()
        a, u = intersection_and_union(pred, seg.long().numpy(), args.num_class)
        inter += a; union += u; losses.append(float(loss))
    iou = inter / (union + 1e-10)
    return iou[1:], dice_from_iu(inter, union)[1:], float(np.mean(losses))


def checkpoint(unet, history, args, epoch):
    """unet_epoch_N.pth = unet.state_dict() (same key set as the reference incl. the aliased encoder keys)."""
    os.makedirs(args.ckpt, exist_ok=True)
    torch.save(history, os.path.join(args.ckpt, "history_epoch_{}.pth".format(epoch)))
    torch.save({k: v.detach().cpu() for k, v in unet.state_dict().items()}, os.path.join(args.ckpt, "unet_epoch_{}.pth".format(epoch)))


def should_checkpoint(epoch, iou, best, num_epoch):
    """train.py:294-329: the per-class / mean best IoUs are tracked from the FIRST epoch; a new best saves unless
    ``epoch < 15``; every 50th epoch and the last epoch always save."""
    improved = False
    mean = float(np.mean(iou))
    for c in range(len(iou)):
        if iou[c] > best["class"][c]:
            best["class"][c] = float(iou[c]); improved = True
    if mean > best["mean"]:
        best["mean"] = mean; improved = True
    if epoch % 50 == 0 or epoch == num_epoch:
        return True
    return improved and epoch >= 15


def build_parser():
    p = argparse.ArgumentParser(description="SAUNet training on MI355X (reference flags from train.py:342-391)")
    p.add_argument("--id", default="saunet_hip")
    p.add_argument("--arch_unet", default="saunet")
    p.add_argument("--weights_unet", default="")
    p.add_argument("--num_epoch", type=int, default=120)
    p.add_argument("--start_epoch", type=int, default=1)
    p.add_argument("--batch_size_per_gpu", type=int, default=1)
    p.add_argument("--optimizer", default="sgd")
    p.add_argument("--lr_encoder", type=float, default=5e-4)
    p.add_argument("--lr_pow", type=float, default=0.9)
    p.add_argument("--beta1", type=float, default=0.9)
    p.add_argument("--weight_decay", type=float, default=1e-4)
    p.add_argument("--fix_bn", action="store_true")
    p.add_argument("--num_class", type=int, default=4)
    p.add_argument("--workers", type=int, default=2)
    p.add_argument("--seed", type=int, default=304)
    p.add_argument("--ckpt", default="./ckpt")
    p.add_argument("--disp_iter", type=int, default=10)
    p.add_argument("--dtype", default="bf16", choices=["bf16", "f32"])
    p.add_argument("--synthetic", type=int, default=64, help="number of synthetic training slices (no ACDC data mounted)")
    p.add_argument("--size", type=int, default=256)
    p.add_argument("--val_slices", type=int, default=8)
    p.add_argument("--augment", action="store_true", help="train on raw ragged slices augmented on the GPU (augment.DeviceAugmenter: the loader's "
                   "crop/flip/rotate/gamma/z-score/elastic chain, train.py:236 + data/ac17_dataloader.py)")
    return p


def main(argv=None):
    args = build_parser().parse_args(argv)
    rank, local, world = dp.init_from_env()
    args.rank, args.world = rank, world
    device = torch.device("cuda", local)
    torch.manual_seed(args.seed)
    set_compute_dtype(torch.bfloat16 if args.dtype == "bf16" else torch.float32)
    unet = SAUNet(num_classes=args.num_class).to(device)
    if args.weights_unet:
        unet.load_state_dict(torch.load(args.weights_unet, map_location="cpu"), strict=False)
    dp.broadcast_parameters(unet)
    sm = SegmentationModule(DualLoss(mode="train"), unet, args.num_class)
    optimizers = optim.create_optimizers(unet, args.optimizer, args.lr_encoder, args.beta1, args.weight_decay)
    buckets = dp.GradientBuckets(list(unet.parameters())) if world > 1 else None
    train_set = RawSyntheticSlices(args.synthetic, args.size, args.seed) if args.augment else SyntheticSlices(args.synthetic, args.size, args.seed)
    if args.augment:
        from .augment import DeviceAugmenter
        args.augmenter = DeviceAugmenter(size=args.size, seed=args.seed + 1000 * rank)
    val_set = SyntheticSlices(args.val_slices, args.size, args.seed + 100000)
    args.running_lr_encoder = args.lr_encoder
    history = {"train": {"epoch": [], "loss": [], "acc": [], "jaccard": []}}
    best = {"class": [0.0] * (args.num_class - 1), "mean": 0.0}
    for epoch in range(args.start_epoch, args.num_epoch + 1):
        idx = dp.shard_indices(len(train_set), rank, world, epoch=epoch, seed=args.seed)
        loader = torch.utils.data.DataLoader(torch.utils.data.Subset(train_set, idx), batch_size=args.batch_size_per_gpu,
                                             shuffle=False, collate_fn=collate_raw if args.augment else collate, num_workers=args.workers,
                                             drop_last=True, pin_memory=not args.augment)
        train_one_epoch(sm, loader, optimizers, epoch, args, history, buckets, device)
        if rank == 0:
            iou, dice, vloss = evaluate(sm, val_set, args, device)
            print("epoch {}: val IoU {} Dice {} loss {:.4f}".format(epoch, np.round(iou, 4), np.round(dice, 4), vloss), flush=True)
            if should_checkpoint(epoch, iou, best, args.num_epoch):
                checkpoint(unet, history, args, epoch)
        if world > 1:
            torch.distributed.barrier()
    return history


if __name__ == "__main__":
    main()
