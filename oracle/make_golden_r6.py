"""Round-6 additions to tests/golden/ from the REAL reference -- CONTAINER-ONLY (imports /root/reference through oracle/ref_import.py).

    PYTHONDONTWRITEBYTECODE=1 python -m oracle.make_golden_r6

modules_DecoderBlockUpsample.npz   models/models.py:203-237 with is_deconv=False (nn.Upsample x2 bilinear align_corners + two
                                   conv3x3_bn_relu): train forward / dX / parameter gradients / running statistics, eval forward
metrics.npz                        SegmentationModuleBase.jaccard (models/models.py:76-78) on seeded binary masks, and
                                   pixel_acc (:51-74) on a seeded ONE-HOT prediction with ties (all-zero pixels: torch.max -> class 0)
The older fixtures are not touched (oracle/make_golden.py wrote them).
"""
import contextlib
import io
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import ref_import  # noqa: E402
from oracle.make_golden import GOLD, rnd, run_module  # noqa: E402


def gen_decoder_upsample(ns, seed=7):
    B, H = 2, 8
    name = "DecoderBlockUpsample"
    with contextlib.redirect_stdout(io.StringIO()):
        mod = ns.DecoderBlock(32, 24, 16, False)
    shapes = [(B, 32, H, H)]
    inputs = [rnd(s, seed, "%s.in%d" % (name, i)) for i, s in enumerate(shapes)]
    out = run_module(name, mod, inputs, seed, None)
    out["meta.seed"] = np.int64(seed)
    out["meta.shape0"] = np.array(shapes[0], np.int64)
    np.savez_compressed(os.path.join(GOLD, "modules_%s.npz" % name), **out)
    print("wrote modules_%s.npz (%d arrays)" % (name, len(out)))


def gen_metrics(ns, seed=23):
    r = np.random.default_rng(seed)
    base = ns.models.SegmentationModuleBase()
    out = {"meta.seed": np.int64(seed)}
    # jaccard(pred, label): pred any integer/bool-like tensor (.long() inside), label long
    pred = torch.from_numpy((r.random((3, 24, 20)) < 0.4).astype(np.float32))
    label = torch.from_numpy((r.random((3, 24, 20)) < 0.35).astype(np.int64))
    out["jaccard.pred"] = pred.numpy(); out["jaccard.label"] = label.numpy()
    out["jaccard.value"] = np.float64(float(base.jaccard(pred, label)))
    # pixel_acc(pred, label, num_class) the way the train branch calls it (:92): pred = round(softmax).long() -- one-hot or all zeros
    cls = r.integers(-1, 4, (2, 18, 22))                       # -1: no class above 0.5 -> an all-zero pixel (tie -> class 0)
    onehot = np.stack([(cls == c) for c in range(4)], 1).astype(np.int64)
    lab = r.integers(0, 4, (2, 18, 22)).astype(np.int64)
    acc, jac = base.pixel_acc(torch.from_numpy(onehot), torch.from_numpy(lab), 4)
    out["pixel_acc.pred"] = onehot; out["pixel_acc.label"] = lab
    out["pixel_acc.acc"] = np.float64(float(acc)); out["pixel_acc.jac"] = np.array([float(j) for j in jac], np.float64)
    np.savez_compressed(os.path.join(GOLD, "metrics.npz"), **out)
    print("wrote metrics.npz  jaccard=%.6f acc=%.6f jac=%s" % (out["jaccard.value"], out["pixel_acc.acc"], out["pixel_acc.jac"]))


if __name__ == "__main__":
    torch.set_num_threads(8)
    torch.manual_seed(0)
    ns = ref_import.load()
    gen_decoder_upsample(ns)
    gen_metrics(ns)
