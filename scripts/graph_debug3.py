"""Run-to-run spread of short training trajectories (eager vs eager vs graph) at several learning rates: picks the tolerances of
tests/test_hip_train.py::test_graph_replays_follow_the_eager_trajectory."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import saunet_amd as S
from tests.test_hip_train import _make_training
from saunet_amd.graph import GraphedStep


def run(dtype, lr, graph, n=6):
    S_, net, sm, opt, feed = _make_training(dtype)
    for g in opt.param_groups:
        g["lr"] = lr

    def step():
        sm.zero_grad(set_to_none=True)
        loss, _ = sm(feed, 1)
        loss.backward()
        opt.step(upload=False)
        return loss.detach()
    opt.upload_hyper()
    if not graph:
        out = [float(step()) for _ in range(n)]
    else:
        g = GraphedStep(step, warmup=1, optimizers=[opt])
        out = [float("nan")] + [float(g.replay()) for _ in range(n - 1)]
    return out, net.final.weight.detach().clone()


for dtype in (torch.float32, torch.bfloat16):
    for lr in (5e-3, 1e-3, 2e-4):
        base, wb = run(dtype, lr, False)
        print(dtype, lr, "base", ["%.5f" % v for v in base])
        for trial in range(3):
            for graph in (False, True):
                o, w = run(dtype, lr, graph)
                d = max(abs(a - b) for a, b in zip(base[1:], o[1:]))
                print("   graph=%d  max|dloss|=%.2e  first=%.2e  dW=%.2e" % (graph, d, abs(base[1] - o[1]), float((w - wb).abs().max() / wb.abs().max())))
S.set_compute_dtype(torch.float32)
