"""Full-size integration check of the round-4 kernels: the configs[1] step (B=32, 256x256, bf16, SGD) trained for N eager steps on one fixed synthetic
batch, once with the new kernels and once with their A/B switches off (SAUNET_DGRAD3_HALO, SAUNET_MM_CELL, SAUNET_DGRAD_ACCUMULATE, SAUNET_WGRAD_MM,
SAUNET_CONV_MM = 0 in a child process); prints both loss curves.  python scripts/fullsize_ab_losses.py [steps]"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
steps = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1] != "--child" else 30
if "--child" in sys.argv:
    import torch
    import saunet_amd as S
    from saunet_amd import optim, data
    S.set_compute_dtype(torch.bfloat16)
    torch.manual_seed(304)
    net = S.SAUNet(num_classes=4).cuda()
    sm = S.SegmentationModule(S.DualLoss(mode="train"), net, 4).train()
    opt = optim.create_optimizers(net, "sgd", 5e-4, 0.9, 1e-4)[0]
    img, seg, edge = data.synthetic_batch(32, 256, 256, seed=304, device="cuda")
    feed = {"image": img, "mask": (seg, edge)}
    out = []
    for i in range(int(sys.argv[-1])):
        sm.zero_grad(set_to_none=True)
        loss, _ = sm(feed, 1)
        loss.backward()
        opt.step()
        out.append(float(loss))
    print("LOSSES " + json.dumps(out))
    sys.exit(0)
res = {}
for name, env in (("new", {}), ("old", {"SAUNET_DGRAD3_HALO": "0", "SAUNET_MM_CELL": "0", "SAUNET_DGRAD_ACCUMULATE": "0", "SAUNET_WGRAD_MM": "0", "SAUNET_CONV_MM": "0"})):
    e = dict(os.environ); e.update(env)
    r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", str(steps)], env=e, capture_output=True, text=True)
    line = [l for l in r.stdout.splitlines() if l.startswith("LOSSES ")]
    if not line:
        print(name, "FAILED", r.stderr[-2000:]); sys.exit(1)
    res[name] = json.loads(line[0][7:])
n, o = res["new"], res["old"]
print("step   new        old        rel.diff")
for i in range(len(n)):
    if i < 10 or i % 5 == 4: print("%4d  %9.5f  %9.5f  %+.2e" % (i, n[i], o[i], (n[i] - o[i]) / o[i]))
import math
assert all(math.isfinite(v) for v in n + o)
print("first-loss diff %.2e, max rel diff over first 5 steps %.2e, final %.5f vs %.5f" % (abs(n[0] - o[0]) / o[0], max(abs(a - b) / b for a, b in zip(n[:5], o[:5])), n[-1], o[-1]))
