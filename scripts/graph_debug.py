import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import saunet_amd as S
from saunet_amd import data, optim
what = sys.argv[1]
S.set_compute_dtype(torch.bfloat16)
net = S.SAUNet(num_classes=4).cuda()
sm = S.SegmentationModule(S.DualLoss(), net, 4).train()
opt = optim.create_optimizers(net, "sgd")[0]
B, HW = int(sys.argv[2]), int(sys.argv[3])
img, seg, edge = data.synthetic_batch(B, HW, HW, device="cuda")
feed = {"image": img, "mask": (seg, edge)}
def fb():
    sm.zero_grad(set_to_none=True)
    loss, _ = sm(feed, 1); loss.backward(); return loss
for _ in range(2):
    fb(); opt.step()
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    if what == "fwd":
        with torch.no_grad(): sm(feed, 1)
    elif what == "fb": fb()
    elif what == "fbo": fb(); opt.step(upload=False)
    elif what == "opt": opt.step(upload=False)
    elif what == "pack": S.functional.PACKS.invalidate(); S.functional.PACKS.prepack()
torch.cuda.current_stream().wait_stream(s)
torch.cuda.synchronize()
print("capturing", what, flush=True)
with torch.cuda.graph(g):
    if what == "fwd":
        with torch.no_grad(): out = sm(feed, 1)
    elif what == "fb": out = fb()
    elif what == "fbo": out = fb(); opt.step(upload=False)
    elif what == "opt": opt.step(upload=False)
    elif what == "pack": S.functional.PACKS.invalidate(); S.functional.PACKS.prepack()
print("captured", flush=True)
g.replay(); torch.cuda.synchronize()
print("replayed OK", what, flush=True)
