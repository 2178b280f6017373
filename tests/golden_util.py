"""Shared helpers for the golden-fixture tests (inputs are regenerated from seeds)."""
import os
import zlib

import numpy as np
import torch

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    return dict(np.load(os.path.join(GOLD, name), allow_pickle=False))


def rnd(shape, seed, key, scale=1.0):
    r = np.random.default_rng([seed, zlib.crc32(key.encode())])
    return torch.from_numpy(r.standard_normal(shape).astype(np.float32) * np.float32(scale))


def module_state(gold, name, seed):
    """Rebuild the module's state dict (reference key names) from the fixture's key list."""
    from oracle import weights as Wt
    sd = {}
    for k in gold:
        if k.startswith("train.grad."):
            key = k[len("train.grad."):]
            arr = gold[k]
            if arr.ndim == 4:
                kind = "conv"
            elif _is_bn(gold, key):
                kind = "gamma" if key.endswith("weight") else "beta"
            else:
                kind = "bias"
            sd[key] = Wt.make_tensor(name + "." + key, tuple(arr.shape), kind, seed)
        elif k.startswith("train.buf."):
            key = k[len("train.buf."):]
            kind = "rmean" if key.endswith("running_mean") else "rvar"
            sd[key] = Wt.make_tensor(name + "." + key, tuple(gold[k].shape), kind, seed)
    return sd


def _is_bn(gold, key):
    pre = key.rsplit(".", 1)[0]
    return ("train.buf." + pre + ".running_mean") in gold


def close(a, b, rtol=1e-4, atol=1e-5):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    scale = max(np.abs(b).max(), 1e-30)
    err = np.abs(a - b).max()
    return err <= atol + rtol * scale, err, scale


def assert_grads_per_tensor(got, ref, keys, tol=3e-2, floor=3e-4):
    """PER-TENSOR gradient check (a bound relative to the GLOBAL gradient maximum lets a small-magnitude tensor be 100 % wrong):
    for every parameter the RMS error must be <= tol x max(its own reference RMS, floor x the largest per-tensor RMS of the network).
    Measured on the float32 HIP path (scripts/gradcheck_per_tensor.py): worst relative L2 over tensors above the floor 4e-3 .. 1.5e-2 at
    64 x 96 .. 256 x 256 (B = 4) -- the BatchNorm betas of dense blocks 2 / 3, whose gradients are 1e-3 of the network's scale and sums over
    10^5 pixels with heavy cancellation, in BOTH float32 implementations (the CPU oracle is float32 too); gradients that are analytically zero (a bias in
    front of a training-mode BatchNorm: the oracle returns ~1e-9 rounding noise, the HIP path exact zeros) sit below the floor."""
    rms = {k: float(ref[k].double().norm()) / max(ref[k].numel(), 1) ** 0.5 for k in keys}
    G = max(rms.values())
    worst = (0.0, None)
    for k in keys:
        g, r = got[k].detach().cpu().double(), ref[k].double()
        if not bool(g.any()) and rms[k] < 1e-4 * G:
            continue          # analytically zero (a conv bias in front of a training-mode BatchNorm): exact zeros here, rounding noise in the oracle
        e = float((g - r).norm()) / max(r.numel(), 1) ** 0.5
        bound = tol * max(rms[k], floor * G)
        assert e <= bound, (k, "rms error %.3e > %.3e (tensor rms %.3e, network scale %.3e)" % (e, bound, rms[k], G))
        worst = max(worst, (e / max(rms[k], floor * G), k))
    return worst
