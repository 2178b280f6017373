"""Synthetic ACDC-like batches for benchmarking and smoke runs (no ACDC data / nibabel in the container).

Format is the loader's (/root/reference/data/ac17_dataloader.py:146-148, 208-221, 254-258):
``image`` float32 [B,3,H,W] = one z-scored plane replicated x3, ``seg`` int64 [B,H,W] in {0,1,2,3}
(0 bg, 1 RV, 2 MYO, 3 LV), ``edge`` float32 [B,1,H,W] in {0,1} = radius-2 distance-transform edges.
"""
import numpy as np
import torch

# offsets with dy^2 + dx^2 <= radius^2 (radius 2): a pixel is an edge of class c iff a pixel of the opposite
# membership lies within Euclidean distance 2  <=>  EDT(m) + EDT(1-m) <= 2 (onehot_to_binary_edges, :236-252)
_DISK2 = [(dy, dx) for dy in range(-2, 3) for dx in range(-2, 3) if 0 < dy * dy + dx * dx <= 4]


def mask_to_edges(mask, num_classes=3):
    """mask: int array [H,W] -> float32 [1,H,W]; same result as the loader's EDT formulation (the image border
    counts as 'outside every class', like the loader's one-pixel zero pad)."""
    mask = np.asarray(mask)
    h, w = mask.shape
    edge = np.zeros((h, w), bool)
    for c in range(1, num_classes + 1):
        m = np.zeros((h + 4, w + 4), bool)
        m[2:-2, 2:-2] = mask == c
        inner = m[2:-2, 2:-2]
        diff = np.zeros((h, w), bool)
        for dy, dx in _DISK2:
            diff |= m[2 + dy:2 + dy + h, 2 + dx:2 + dx + w] != inner
        # the zero pad of the loader is ONE pixel wide: positions two pixels outside the image do not exist
        # for class pixels on the border the nearest "outside" pixel is at distance 1 -> already covered by |dy|,|dx|<=1
        edge |= diff & _valid_partner(m, inner, h, w)
    return edge.astype(np.float32)[None]


def _valid_partner(m, inner, h, w):
    """Restrict the radius-2 test to partners inside the one-pixel-padded frame used by the loader."""
    ok = np.zeros((h, w), bool)
    yy, xx = np.mgrid[0:h, 0:w]
    for dy, dx in _DISK2:
        py, px = yy + dy, xx + dx
        inside = (py >= -1) & (py <= h) & (px >= -1) & (px <= w)
        ok |= inside & (m[2 + dy:2 + dy + h, 2 + dx:2 + dx + w] != inner)
    return ok


def synthetic_batch(B, H, W, seed=304, device=None):
    imgs, segs, edges = [], [], []
    yy, xx = np.mgrid[0:H, 0:W].astype(np.float32)
    for b in range(B):
        r = np.random.default_rng(seed + b)
        cy, cx = H * (0.5 + 0.08 * r.standard_normal()), W * (0.5 + 0.08 * r.standard_normal())
        ro = min(H, W) * r.uniform(0.13, 0.2)
        ri = ro * r.uniform(0.55, 0.7)
        ecc = r.uniform(0.8, 1.25)
        d_lv = np.sqrt(((yy - cy) * ecc) ** 2 + (xx - cx) ** 2)
        rvx = cx - ro * r.uniform(1.3, 1.6)
        d_rv = np.sqrt(((yy - cy) / 1.4) ** 2 + (xx - rvx) ** 2)
        seg = np.zeros((H, W), np.int64)
        seg[d_rv < ro * 0.75] = 1
        seg[d_lv < ro] = 2
        seg[d_lv < ri] = 3
        inten = np.array([0.15, 0.75, 0.35, 0.9], np.float32)[seg]
        inten = inten + 0.25 * np.exp(-(((yy - H / 2) / (0.45 * H)) ** 2 + ((xx - W / 2) / (0.45 * W)) ** 2)).astype(np.float32)
        img = inten + 0.1 * r.standard_normal((H, W)).astype(np.float32)
        img = (img - img.mean()) / (img.std() + 1e-8)
        imgs.append(np.repeat(img[None].astype(np.float32), 3, 0)); segs.append(seg); edges.append(mask_to_edges(seg))
    out = (torch.from_numpy(np.stack(imgs)), torch.from_numpy(np.stack(segs)), torch.from_numpy(np.stack(edges)))
    if device is not None:
        out = tuple(t.to(device) for t in out)
    return out
