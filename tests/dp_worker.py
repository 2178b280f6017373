"""Worker of tests/test_hip_dp.py: a few data-parallel training steps of the HIP SAUNet -- dp.GradientBuckets (overlap hooks, bucketed
all-reduce) + the SyncBN statistic all-reduces -- with a DIFFERENT shard on every rank.  Launched by torch.distributed.run, either one
rank per GPU over RCCL (backend nccl) or, on a one-GPU box, several ranks sharing the GPU over gloo (SAUNET_DIST_BACKEND=gloo
SAUNET_SHARE_GPU=1).  Rank 0 writes its parameters and buffers; the test compares them with the oracle's K-replica emulation."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import saunet_amd as S                                  # noqa: E402
from saunet_amd import dp, optim                        # noqa: E402
from oracle import saunet_ref as R, weights as Wt       # noqa: E402  (test infrastructure: identical initial weights + inputs)

SEED, SIZE, PER_RANK = 3, 64, 2
LR, MOM, WD = 1e-2, 0.9, 1e-4


def shard(rank):
    return Wt.synthetic_batch(PER_RANK, SIZE, SIZE, seed=500 + 17 * rank)


def main(out_path):
    rank, local, world = dp.init_from_env()
    dev = torch.device("cuda", local)
    S.set_compute_dtype(torch.float32)
    net = S.SAUNet(num_classes=4).to(dev)
    net.load_state_dict(Wt.make_state_dict(R.state_dict_spec(), SEED), strict=False)
    dp.broadcast_parameters(net)
    sm = S.SegmentationModule(S.DualLoss(mode="train"), net, 4).train()
    opts = optim.create_optimizers(net, "sgd", LR, MOM, WD)
    buckets = dp.GradientBuckets(list(net.parameters()), bucket_mb=8.0) if world > 1 else None
    img, seg, edge = shard(rank)
    feed = {"image": img.to(dev), "mask": (seg.to(dev), edge.to(dev))}
    losses = []
    for _ in range(int(os.environ.get("DP_WORKER_STEPS", "2"))):
        sm.zero_grad(set_to_none=True)
        loss, _ = sm(feed, 1)
        loss.mean().backward()
        if buckets is not None:
            buckets.finish()
        for o in opts:
            o.step()
        losses.append(float(loss.mean()))
    torch.cuda.synchronize()
    all_losses = [None] * world
    if world > 1:
        torch.distributed.all_gather_object(all_losses, losses)
    else:
        all_losses = [losses]
    if rank == 0:
        torch.save({"params": {k: v.detach().float().cpu() for k, v in net.named_parameters()},
                    "buffers": {k: v.detach().float().cpu() for k, v in net.named_buffers()},
                    "grad_none": [k for k, v in net.named_parameters() if v.grad is None],
                    "losses": all_losses, "world": world, "backend": torch.distributed.get_backend() if world > 1 else "none"}, out_path)
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main(sys.argv[1])
