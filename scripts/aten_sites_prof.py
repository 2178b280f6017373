"""Python call sites of the ATen / runtime kernels (fills, copies, adds) left in one eager training step: torch.profiler with stacks.
python scripts/aten_sites_prof.py"""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity
import saunet_amd as S
from saunet_amd import optim, data
S.set_compute_dtype(torch.bfloat16)
torch.manual_seed(0)
net = S.SAUNet(num_classes=4).cuda()
sm = S.SegmentationModule(S.DualLoss(mode="train"), net, 4).train()
opts = optim.create_optimizers(net, "sgd", 5e-4, 0.9, 1e-4)
img, seg, edge = data.synthetic_batch(32, 256, 256, seed=1)
feed = {"image": img.cuda(), "mask": (seg.cuda(), edge.cuda())}
def step():
    sm.zero_grad(set_to_none=True)
    loss, _ = sm(feed, 1)
    loss.mean().backward()
    for o in opts: o.step()
for _ in range(2): step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
    step()
    torch.cuda.synchronize()
WANT = ("aten::fill_", "aten::zero_", "aten::copy_", "aten::add", "aten::add_", "aten::zeros", "aten::clone", "aten::_to_copy", "aten::mul", "aten::div", "aten::sum",
        "hipMemcpyAsync", "hipMemsetAsync", "aten::cat", "aten::contiguous", "aten::mean", "aten::ones_like", "aten::constant_pad_nd")
cnt = collections.Counter(); dur = collections.Counter()
for e in prof.events():
    if e.name in WANT and (e.device_time_total > 0 or e.name.startswith("hip")):
        st = [s for s in (e.stack or []) if "saunet_amd" in s or "shape-attentive" in s or "autograd" in s]
        site = " <- ".join(s.split("/")[-1] for s in st[:3]) or "<no python frame: autograd engine>"
        key = (e.name, str(e.input_shapes)[:60], site)
        cnt[key] += 1; dur[key] += e.device_time_total
for k, n in sorted(cnt.items(), key=lambda kv: -dur[kv[0]]):
    print("%4d %8.1f us  %-22s %-60s %s" % (n, dur[k], k[0], k[1], k[2]))
