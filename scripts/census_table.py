"""Per-kernel-symbol table of one live training step (bench.py's census): ms, launches, algorithmic GB/s and TF/s -- where the step's time is
and how far each kernel is from its bound.   python scripts/census_table.py [batch] [size]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
import saunet_amd as S
from saunet_amd import optim, data
batch = int(sys.argv[1]) if len(sys.argv) > 1 else 32
size = int(sys.argv[2]) if len(sys.argv) > 2 else 256
S.set_compute_dtype(torch.bfloat16)
torch.manual_seed(304)
net = S.SAUNet(num_classes=4).cuda()
sm = S.SegmentationModule(S.DualLoss(mode="train"), net, 4).train()
opt = optim.create_optimizers(net, "sgd", lr=5e-4, momentum=0.9, weight_decay=1e-4)[0]
img, seg, edge = data.synthetic_batch(batch, size, size, seed=304, device="cuda")
feed = {"image": img, "mask": (seg, edge)}
def step():
    sm.zero_grad(set_to_none=True)
    loss, _ = sm(feed, 1)
    loss.backward()
    opt.upload_hyper(); opt.step(upload=False)
for _ in range(2): step()
c = bench.live_kernel_census(S, step, torch.bfloat16)
rows = sorted(c.items(), key=lambda kv: -kv[1]["ms"])
tot = sum(v["ms"] for _, v in rows)
print("# live census of one eager step (B=%d, %dx%d, bf16): %.2f ms in %d launches of %d symbols; event pair overhead %s us" % (batch, size, size, tot, sum(v["launches"] for _, v in rows), len(rows), bench.live_kernel_census.event_pair_overhead_us))
print("%-8s %-7s %-9s %-9s %-8s %s" % ("ms", "calls", "GB/s", "TF/s", "AI", "kernel"))
for k, v in rows:
    if v["ms"] < 0.02: continue
    pm = v["priced_ms"]
    gbs = v["algorithmic_bytes"] / pm / 1e6 if pm > 0 else 0
    tf = v["flops"] / pm / 1e9 if pm > 0 else 0
    ai = v["flops"] / v["algorithmic_bytes"] if v["algorithmic_bytes"] else 0
    print("%-8.3f %-7d %-9.0f %-9.1f %-8.1f %s%s" % (v["ms"], v["launches"], gbs, tf, ai, k, ("  (+" + ", ".join(sorted(v["includes"])) + ")") if v["includes"] else ""))
