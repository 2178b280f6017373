import os, sys, collections
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
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
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    step(); torch.cuda.synchronize()
cnt = collections.Counter()
for e in prof.events():
    n = e.name
    if "emcpy" in n or "emset" in n or "copyBuffer" in n or "fillBuffer" in n:
        cnt[(n, str(e.device_type))] += 1
for k, v in cnt.most_common(): print(v, k)
# parents of memcpy calls
par = collections.Counter()
for e in prof.events():
    if "hipMemcpy" in e.name:
        p = e.cpu_parent
        chain = []
        while p is not None and len(chain) < 4:
            chain.append(p.name); p = p.cpu_parent
        par[" <- ".join(chain)] += 1
for k, v in par.most_common(20): print(v, k)
