"""Deterministic per-key weight generator -- TEST INFRASTRUCTURE (see oracle/__init__.py).

ImageNet DenseNet weights are unobtainable offline and committing a 130 MB state dict
is not an option, so parity tests regenerate identical weights on every machine from
``(seed, key)`` with numpy's PCG64 (bit-stable across numpy versions/platforms).
Scales are chosen so activations stay O(1) through 120+ layers (He-style fan-in
scaling) and BatchNorm affine parameters are non-trivial.
"""
import zlib

import numpy as np
import torch


def _rng(seed, key):
    return np.random.default_rng([seed, zlib.crc32(key.encode())])


def make_tensor(key, shape, kind, seed=0):
    r = _rng(seed, key)
    if kind == "conv":
        fan_in = int(np.prod(shape[1:])) if len(shape) > 1 else 1
        if key.endswith("mrf.up.0.weight") or key == "dec1.block.1.weight":  # ConvTranspose [Cin,Cout,4,4]
            fan_in = shape[0] * 4  # 2x2 taps reach each output pixel
        a = r.standard_normal(shape, dtype=np.float32) * np.float32(np.sqrt(2.0 / max(fan_in, 1)))
    elif kind == "bias":
        a = r.standard_normal(shape, dtype=np.float32) * np.float32(0.05)
    elif kind == "gamma":
        a = r.uniform(0.6, 1.4, shape).astype(np.float32)
    elif kind == "beta":
        a = r.standard_normal(shape, dtype=np.float32) * np.float32(0.1)
    elif kind == "rmean":
        a = r.standard_normal(shape, dtype=np.float32) * np.float32(0.1)
    elif kind == "rvar":
        a = r.uniform(0.5, 1.5, shape).astype(np.float32)
    elif kind == "count":
        return torch.zeros((), dtype=torch.long)
    else:
        raise ValueError(kind)
    return torch.from_numpy(np.ascontiguousarray(a))


def make_state_dict(spec, seed=0):
    return {k: make_tensor(k, shape, kind, seed) for k, shape, kind in spec}


def trainable_keys(spec):
    return [k for k, _, kind in spec if kind in ("conv", "bias", "gamma", "beta")]


def synthetic_batch(B, H, W, seed=304):
    """Ellipse phantom batch in the loader's format:
    image float32 [B,3,H,W] (one z-scored plane x3), seg int64 [B,H,W] in {0..3},
    edge float32 [B,1,H,W] in {0,1} (radius-2 distance-transform edges)."""
    from .saunet_ref import mask_to_edges
    imgs, segs, edges = [], [], []
    yy, xx = np.mgrid[0:H, 0:W].astype(np.float32)
    for b in range(B):
        r = np.random.default_rng(seed + b)
        cy, cx = H * (0.5 + 0.08 * r.standard_normal()), W * (0.5 + 0.08 * r.standard_normal())
        ro = min(H, W) * r.uniform(0.13, 0.2)           # LV+MYO outer radius
        ri = ro * r.uniform(0.55, 0.7)                  # LV cavity radius
        ecc = r.uniform(0.8, 1.25)
        d_lv = np.sqrt(((yy - cy) * ecc) ** 2 + (xx - cx) ** 2)
        rvx = cx - ro * r.uniform(1.3, 1.6)
        d_rv = np.sqrt(((yy - cy) / 1.4) ** 2 + (xx - rvx) ** 2)
        seg = np.zeros((H, W), np.int64)
        seg[d_rv < ro * 0.75] = 1       # RV
        seg[d_lv < ro] = 2              # MYO
        seg[d_lv < ri] = 3              # LV
        inten = np.array([0.15, 0.75, 0.35, 0.9], np.float32)[seg]
        inten = inten + 0.25 * np.exp(-(((yy - H / 2) / (0.45 * H)) ** 2 + ((xx - W / 2) / (0.45 * W)) ** 2)).astype(np.float32)
        img = inten + 0.1 * r.standard_normal((H, W)).astype(np.float32)
        img = (img - img.mean()) / (img.std() + 1e-8)
        imgs.append(np.repeat(img[None].astype(np.float32), 3, 0))
        segs.append(seg)
        edges.append(mask_to_edges(seg))
    return (torch.from_numpy(np.stack(imgs)), torch.from_numpy(np.stack(segs)),
            torch.from_numpy(np.stack(edges)))
