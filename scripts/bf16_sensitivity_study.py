"""How much of the bf16 step's run-to-run / batch-duplication spread is inherent?  The CPU oracle's bf16-storage emulation evaluated twice, the
second time with every weight perturbed by a relative 1e-7 (one float32 ulp, the size of a summation-order difference): relative L2 distance
and cosine of the two gradients per parameter group, next to the same experiment in exact float32 arithmetic.  B=8, 256 x 256, seed 13.
python scripts/bf16_sensitivity_study.py > profiles/r05_bf16_sensitivity.txt"""
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
g = torch.Generator().manual_seed(5)
sd2 = {k: (v * (1.0 + 1e-7 * torch.randn(v.shape, generator=g)).to(v.dtype) if (k in keys and v.is_floating_point()) else v.clone()) for k, v in sd.items()}


def grads(state, emulate):
    s = {k: v.clone() for k, v in state.items()}
    for k in keys:
        s[k].requires_grad_(True)
    with R.bf16_storage(emulate):
        loss, _, _, _ = R.segmentation_step(s, *batch, True)
    loss.backward()
    return float(loss), {k: s[k].grad for k in keys}


rows = {}
for emu in (True, False):
    l1, g1 = grads(sd, emu); l2, g2 = grads(sd2, emu)
    t = collections.defaultdict(list)
    gmax = max(float(g1[k].abs().max()) for k in keys)
    for k in keys:
        if float(g1[k].abs().max()) < 1e-5 * gmax:
            continue
        t[group_of(k)].append(cos_rel(g2[k], g1[k]))
    rows[emu] = (t, l1, l2)
print("# gradient change under a relative 1e-7 perturbation of every weight: bf16-storage emulation vs exact float32 arithmetic (CPU oracle)")
print("# losses: bf16 emulation %.6f -> %.6f ; float32 %.6f -> %.6f" % (rows[True][1], rows[True][2], rows[False][1], rows[False][2]))
print("%-14s %4s | %-28s | %-28s" % ("group", "n", "bf16 emulation: med cos, max relL2", "float32: med cos, max relL2"))
for gname in sorted(rows[True][0], key=lambda q: np.median([r[0] for r in rows[True][0][q]])):
    a, b = rows[True][0][gname], rows[False][0].get(gname, [])
    if not a or not b:
        continue
    print("%-14s %4d | %9.5f   %10.3e        | %9.7f   %10.3e" % (gname, len(a), np.median([r[0] for r in a]), max(r[1] for r in a), np.median([r[0] for r in b]), max(r[1] for r in b)))
