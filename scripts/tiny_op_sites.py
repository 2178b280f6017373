"""Which Python call sites issue the small aten copies / fills / adds of one eager training step?  (torch.profiler with stacks)
python scripts/tiny_op_sites.py"""
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
for _ in range(3): step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU], with_stack=True, record_shapes=True) as prof:
    step()
torch.cuda.synchronize()
allops = collections.Counter(ev.name for ev in prof.events() if ev.name.startswith("aten::"))
print("all aten ops of one step:", dict(allops.most_common(40)))
tot = collections.Counter(); sites = collections.Counter()
for ev in prof.events():
    if ev.name in ("aten::copy_", "aten::fill_", "aten::zero_", "aten::zeros", "aten::add_", "aten::add", "aten::clone", "aten::mul", "aten::sum"):
        tot[ev.name] += 1
        st = [f for f in (ev.stack or []) if ("repo" in f or "saunet" in f) and "tiny_op_sites" not in f]
        shp = str([tuple(x) for x in (ev.input_shapes or []) if x])[:70]
        sites[(ev.name, (st[0][-90:] if st else "(autograd thread) shapes " + shp))] += 1
print(dict(tot))
for (name, site), n in sorted(sites.items(), key=lambda kv: -kv[1])[:50]:
    print("%4d  %-12s %s" % (n, name, site))
