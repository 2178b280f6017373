"""cProfile of the eager (no hipGraph) training step: where does the host time go?  python scripts/eager_cpu_profile.py"""
import cProfile, pstats, io, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
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
import time
t=time.time()
for _ in range(5): step()
t1=time.time()-t          # host time to ENQUEUE 5 steps
torch.cuda.synchronize()
print("host enqueue per step: %.1f ms ; incl. GPU drain: %.1f ms" % (t1/5*1e3, (time.time()-t)/5*1e3))
pr = cProfile.Profile(); pr.enable()
for _ in range(3): step()
pr.disable(); torch.cuda.synchronize()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(22); print(s.getvalue()[:5000])
