"""Mixed-storage study on the CPU oracle: would float32 storage on the low-resolution maps (blocks 3 / 4, center, dec5 -- under
10 % of the bytes) bring the per-group gradient cosines of the bf16 step back to >= 0.99?  Exact float32 oracle vs its bf16-storage emulation with
(a) everything in bf16, (b) float32 on maps <= 16 x 16, (c) <= 32 x 32, (d) <= 64 x 64.  B=8, 256 x 256, seed 13 (the geometry of
tests/test_hip_parity_bf16.py).   python scripts/mixed_storage_study.py > profiles/r05_mixed_storage_study.txt"""
import collections, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
from oracle import saunet_ref as R, weights as Wt
from test_hip_parity_bf16 import group_of, cos_rel
torch.set_num_threads(min(os.cpu_count() or 8, 32))
spec = R.state_dict_spec(); sd = Wt.make_state_dict(spec, 13); keys = Wt.trainable_keys(spec)
batch = Wt.synthetic_batch(8, 256, 256, seed=113)


def grads(emulate, maxhw):
    s = {k: v.clone() for k, v in sd.items()}
    for k in keys:
        s[k].requires_grad_(True)
    with R.bf16_storage(emulate, maxhw):
        loss, _, _, _ = R.segmentation_step(s, *batch, True)
    loss.backward()
    return float(loss), {k: s[k].grad for k in keys}


la, ga = grads(False, 0)
gmax = max(float(ga[k].abs().max()) for k in keys)
modes = [("all bf16", 0), ("f32 <= 16x16", 16), ("f32 <= 32x32", 32), ("f32 <= 64x64", 64)]
tabs, losses = [], []
for name, m in modes:
    lb, gb = grads(True, m)
    t = collections.defaultdict(list)
    for k in keys:
        if float(ga[k].abs().max()) < 1e-5 * gmax:
            continue
        t[group_of(k)].append(cos_rel(gb[k], ga[k])[0])
    tabs.append(t); losses.append(lb)
print("# bf16-storage emulation of the CPU oracle against its exact float32 gradients: median / minimum cosine per parameter group")
print("# loss exact %.6f ; " % la + " ; ".join("%s %.6f" % (n, l) for (n, _), l in zip(modes, losses)))
print("%-14s %4s | " % ("group", "n") + " | ".join("%-17s" % n for n, _ in modes))
order = sorted(tabs[0], key=lambda g: np.median(tabs[0][g]))
for g in order:
    print("%-14s %4d | " % (g, len(tabs[0][g])) + " | ".join("%7.4f  %7.4f " % (np.median(t[g]), min(t[g])) for t in tabs))
