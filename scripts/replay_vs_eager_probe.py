import sys; sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import torch
import test_hip_train as T
import saunet_amd as S
from saunet_amd.graph import GraphedStep
for dtype, lr in ((torch.float32, 2e-4), (torch.float32, 5e-3), (torch.bfloat16, 1e-3)):
    try:
        S_, net, sm, opt, feed = T._make_training(dtype, lr=lr)
        eager = []
        for _ in range(6):
            sm.zero_grad(set_to_none=True); loss, _ = sm(feed, 1); loss.backward(); opt.step(); eager.append(float(loss))
        w_e = {k: v.detach().clone() for k, v in net.named_parameters()}
        S_, net, sm, opt, feed = T._make_training(dtype, lr=lr)
        def step():
            sm.zero_grad(set_to_none=True); loss, _ = sm(feed, 1); loss.backward(); opt.step(upload=False); return loss.detach()
        g = GraphedStep(step, warmup=1, optimizers=[opt])
        rep = [float(g.replay()) for _ in range(5)]
        torch.cuda.synchronize()
        w_g = dict(net.named_parameters())
        wd = max(float((w_g[k].detach() - w_e[k]).abs().max() / w_e[k].abs().max().clamp_min(1e-12)) for k in w_e)
        print(dtype, lr, "max |eager-replay| loss", max(abs(a - b) for a, b in zip(eager[1:], rep)), "max rel weight diff", wd, "bit-equal losses", eager[1:] == rep)
    finally:
        S.set_compute_dtype(torch.float32)
