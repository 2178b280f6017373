import os, sys, collections, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import saunet_amd as S
from saunet_amd import optim, data, lib as L
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
cnt = collections.Counter()
orig = L.call
WATCH = ("saunet_copy_channels", "saunet_affine_act", "saunet_bn_backward_reduce", "saunet_bn_backward_apply", "saunet_channel_sum")
def call(name, *a):
    if name in WATCH:
        st = [f for f in traceback.extract_stack()[:-1] if "saunet" in f.filename or "functional" in f.filename or "modules" in f.filename]
        site = " <- ".join("%s:%d" % (os.path.basename(f.filename), f.lineno) for f in st[-3:])
        if name == "saunet_copy_channels": key = (name, "P=%d C=%d acc=%d" % (a[6], a[7], a[8]), site)
        elif name == "saunet_affine_act": key = (name, "P=%d C=%d" % (a[10], a[11]), site)
        elif name == "saunet_bn_backward_reduce": key = (name, "P=%d C=%d" % (a[15], a[16]), site)
        elif name == "saunet_bn_backward_apply": key = (name, "P=%d C=%d" % (a[24], a[25]), site)
        else: key = (name, "", site)
        cnt[key] += 1
    return orig(name, *a)
L.call = call
import saunet_amd.functional as HF
HF.L.call = call
step()
torch.cuda.synchronize()
for k, v in sorted(cnt.items(), key=lambda kv: (kv[0][0], -kv[1])):
    print(v, *k)
