import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
from tests.test_hip_train import _make_training
from saunet_amd.graph import GraphedStep
for dtype in (torch.float32,):
    S, net, sm, opt, feed = _make_training(dtype)
    eager = []
    for _ in range(6):
        sm.zero_grad(set_to_none=True)
        loss, _ = sm(feed, 1)
        loss.backward(); opt.step(); eager.append(float(loss))
    print("eager ", eager)
    S, net, sm, opt, feed = _make_training(dtype)
    def step():
        sm.zero_grad(set_to_none=True)
        loss, _ = sm(feed, 1)
        loss.backward()
        opt.step(upload=False)
        return loss.detach()
    g = GraphedStep(step, warmup=1, optimizers=[opt])
    rep = [float(g.replay()) for _ in range(5)]
    print("replay", rep)
    # second experiment: eager again but with upload=False steps after one upload
    S, net, sm, opt, feed = _make_training(dtype)
    opt.upload_hyper()
    e2 = []
    for _ in range(6):
        l = step(); e2.append(float(l))
    print("eager2", e2)
