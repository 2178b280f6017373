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
