"""Every weight-gradient call of one live training step (B = 32, 256 x 256, bf16) with its geometry, HIP-event time and algorithmic rate --
the per-launch view of the step's largest pool.   python scripts/wgrad_census.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
import saunet_amd as S
from saunet_amd import optim, data
batch, size = 32, 256
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
for _ in range(3): step()
torch.cuda.synchronize()
L = S.lib
handle = L.load()
orig = L.call
recs = []
def traced(name, *args):
    if "wgrad" not in name:
        return orig(name, *args)
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    handle.saunet_launch_log()
    e0.record(); orig(name, *args); e1.record()
    log = (handle.saunet_launch_log() or b"").decode()
    geo = ""
    o = getattr(args[0], "_obj", args[0]) if args else None
    if name in ("saunet_conv2d_wgrad", "saunet_conv2d_wgrad_deferred") and hasattr(o, "Cin"):
        geo = "%dx%d%s  %d -> %d  @ %d x %d x %d" % (o.KH, o.KW, " T" if o.transposed else "", o.Cin, o.Cout, o.N, o.H, o.W)
    elif name == "saunet_conv2d_wgrad_grouped":
        geo = "%dx%d  %d problems  Cin %d..%d -> %d  @ %d x %d x %d" % (o.KH, o.KH, o.count, o.item[0].Cin, o.item[o.count - 1].Cin, o.item[0].Cout, o.N, o.H, o.W)
    recs.append((name, geo, log, e0, e1, bench._call_work(name, args, 2)))
c0, c1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
c0.record(); torch.cuda._sleep(1000000); c1.record(); torch.cuda.synchronize()
per_ms = 1000000 / max(c0.elapsed_time(c1), 1e-3)
import time
t0 = time.perf_counter(); step(); host_ms = (time.perf_counter() - t0) * 1e3
torch.cuda.synchronize()
L.call = traced
try:
    torch.cuda._sleep(int(per_ms * min(2.0 * host_ms + 20.0, 600.0)))      # the GPU is held until the host has queued the whole step
    t0 = time.perf_counter(); step(); print("# host time of the traced step %.1f ms (held %.1f ms)" % ((time.perf_counter() - t0) * 1e3, min(2.0 * host_ms + 20.0, 600.0)))
finally:
    L.call = orig
torch.cuda.synchronize()
tot = 0.0
print("%-9s %-9s %-9s %-52s %s" % ("us", "GB/s", "TF/s", "geometry", "kernels"))
for name, geo, log, e0, e1, work in recs:
    us = e0.elapsed_time(e1) * 1e3
    tot += us
    gbs = work[0] / us / 1e3 if work else 0
    tf = work[1] / us / 1e6 if work else 0
    print("%-9.1f %-9.0f %-9.1f %-52s %s" % (us, gbs, tf, geo or name, log[:150]))
print("# %d weight-gradient calls, %.3f ms per step" % (len(recs), tot / 1e3))
