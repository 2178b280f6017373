"""Which ATen operators still launch kernels inside one training step, and from where: python scripts/aten_sites.py
(TorchDispatchMode over one eager step: operator, shapes, the nearest saunet_amd frames; autograd-engine calls show as <engine>)"""
import os, sys, collections, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.utils._python_dispatch import TorchDispatchMode
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
cnt = collections.Counter()
SKIP = ("aten.view", "aten.detach", "aten.alias", "aten._unsafe_view", "aten.as_strided", "aten.slice", "aten.select", "aten.t.", "aten.permute",
        "aten.reshape", "aten.expand", "aten.unsqueeze", "aten.squeeze", "aten.transpose", "aten.empty", "aten.narrow", "aten.split", "aten.unbind",
        "aten.is_", "aten.sym_", "aten.stride", "aten.size", "aten.numel", "aten.dim", "aten._local_scalar", "aten.lift_fresh", "aten.new_empty")
class Log(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = str(func)
        if not name.startswith(SKIP):
            st = [f for f in traceback.extract_stack()[:-1] if "saunet_amd" in f.filename or "shape-attentive" in f.filename]
            site = " <- ".join("%s:%d" % (os.path.basename(f.filename), f.lineno) for f in st[-3:][::-1]) or "<engine>"
            shp = ",".join(str(tuple(a.shape)) + ("" if a.is_contiguous() or a.dim() != 4 else "cl") + str(a.dtype).replace("torch.", ":") for a in args if isinstance(a, torch.Tensor))
            cnt[(name, shp, site)] += 1
        return func(*args, **(kwargs or {}))
with Log():
    step()
torch.cuda.synchronize()
for (name, shp, site), n in sorted(cnt.items(), key=lambda kv: (kv[0][0], -kv[1])):
    print("%4d  %-28s %-70s %s" % (n, name, shp[:70], site))
