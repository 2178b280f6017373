"""Worker of tests/test_hip_dp.py: two training steps of the HIP SAUNet with the bucketed gradient all-reduce (dp.GradientBuckets,
overlap hooks) and the SyncBN statistic all-reduces.  Launched by torch.distributed.run (2 ranks sharing one GPU over gloo:
SAUNET_DIST_BACKEND=gloo SAUNET_SHARE_GPU=1) or as a single process.  Every rank trains on the SAME batch, so the averaged
gradients and the synchronised statistics must equal the single-process ones.  Writes the parameter vector of rank 0."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import saunet_amd as S                      # noqa: E402
from saunet_amd import data, dp, optim      # noqa: E402


def main(out_path):
    rank, local, world = dp.init_from_env()
    dev = torch.device("cuda", local)
    S.set_compute_dtype(torch.float32)
    torch.manual_seed(1234)
    net = S.SAUNet(num_classes=4).to(dev)
    dp.broadcast_parameters(net)
    sm = S.SegmentationModule(S.DualLoss(mode="train"), net, 4).train()
    opts = optim.create_optimizers(net, "sgd", 1e-2, 0.9, 1e-4)
    buckets = dp.GradientBuckets(list(net.parameters()), bucket_mb=8.0) if world > 1 else None
    img, seg, edge = data.synthetic_batch(2, 64, 64, seed=5)
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
    if rank == 0:
        flat = torch.cat([p.detach().float().reshape(-1).cpu() for p in net.parameters()])
        stats = torch.cat([b.detach().float().reshape(-1).cpu() for n, b in net.named_buffers() if "running_" in n and "_tmp" not in n])
        torch.save({"params": flat, "running": stats, "losses": losses, "world": world}, out_path)
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main(sys.argv[1])
