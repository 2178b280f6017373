"""Test-set post-processing of predicted label volumes: undo the centre crop / zero pad and resample back to the original grid.

Mirrors /root/reference/test_and_pack.py:28-76 (`round_num`, `undo_crop`, `resample_to_orig`):
  * `undo_crop(img, pred)`: the network saw a centre crop / zero pad of the re-scaled slice `img`; put the prediction back on
    `img`'s grid -- zero border where the image was cropped (PIL ImageOps.expand), centre crop where it was padded;
  * `resample_to_orig`: every slice un-cropped, then the whole volume resized to the original shape with order-0 (nearest)
    interpolation, `skimage.transform.resize(..., order=0, preserve_range=True, mode='constant')` = pixel-centre sampling
    src = floor((dst + 0.5) * n_src / n_dst).
Both steps are pure index arithmetic, so the device path (`resample_to_orig_device`) does them as ONE gather kernel over the
predicted labels (libsaunet_hip.so: saunet_labels_uncrop_resize); the numpy functions below are the host restatement the kernel
and the reference-generated fixture (tests/golden/undo_crop.npz) are checked against.
"""
import ctypes as C

import numpy as np
import torch

from . import lib as L


def round_num(x):
    """test_and_pack.py:28-29 (round half up for positive x)"""
    return int(x) + 1 if (x - int(x)) >= 0.5 else int(x)


def undo_crop_geometry(w, h, tw, th):
    """-> (bx0, by0, cw, ch, left, top): the un-cropped map is  p[y][x] = pred[by0 + y - top][bx0 + x - left]  for
    0 <= y - top < ch, 0 <= x - left < cw, else 0  (image size w x h, prediction size tw x th; test_and_pack.py:31-59)."""
    if w >= tw and h >= th:
        return 0, 0, tw, th, int(round_num((w - tw) / 2.0)), int(round_num((h - th) / 2.0))
    pad_h, pad_w = max(th - h, 0), max(tw - w, 0)
    b = [pad_w // 2, pad_h // 2, pad_w // 2 + w, pad_h // 2 + h]
    if pad_w == 0:
        b[2] = tw
    if pad_h == 0:
        b[3] = th
    left = max(int(round_num((w - tw) / 2.0)), 0)
    top = max(int(round_num((h - th) / 2.0)), 0)
    return b[0], b[1], b[2] - b[0], b[3] - b[1], left, top


def undo_crop(img, pred):
    """img: the re-scaled 2-D slice the crop was taken from (only its shape matters); pred: [th, tw] labels -> [h, w] uint8."""
    h, w = img.shape
    th, tw = pred.shape
    bx0, by0, cw, ch, left, top = undo_crop_geometry(w, h, tw, th)
    out = np.zeros((h, w), np.uint8)
    src = pred[by0:by0 + ch, bx0:bx0 + cw].astype(np.uint8)
    out[top:top + ch, left:left + cw] = src[:max(min(ch, h - top), 0), :max(min(cw, w - left), 0)]
    return out


def nearest_index(n_dst, n_src):
    """order-0 resize sampling positions: pixel centre of the destination mapped onto the source grid"""
    return np.minimum(np.floor((np.arange(n_dst) + 0.5) * n_src / n_dst).astype(np.int64), n_src - 1)


def resample_to_orig(post_scale_shape, orig_shape, pred):
    """pred [th, tw, Z] labels -> volume of `orig_shape` [H, W, Z] (test_and_pack.py:61-76); host restatement."""
    h, w, z = post_scale_shape
    assert z == orig_shape[2] == pred.shape[2]
    stack = np.zeros(post_scale_shape, dtype=np.uint8)
    dummy = np.empty((h, w))
    for i in range(z):
        stack[:, :, i] = undo_crop(dummy, pred[:, :, i])
    iy, ix = nearest_index(orig_shape[0], h), nearest_index(orig_shape[1], w)
    return stack[iy][:, ix]


def resample_to_orig_device(pred, post_scale_hw, orig_hw):
    """pred: int64 labels [Z, th, tw] on the GPU (the argmax of HF.softmax_argmax) -> uint8 [Z, H, W] on the GPU."""
    if not pred.is_cuda:
        raise RuntimeError("resample_to_orig_device runs on the GPU (use resample_to_orig for host arrays)")
    pred = pred.to(torch.int64).contiguous()
    z, th, tw = pred.shape
    h, w = post_scale_hw
    H, W = orig_hw
    bx0, by0, cw, ch, left, top = undo_crop_geometry(w, h, tw, th)
    out = torch.empty((z, H, W), dtype=torch.uint8, device=pred.device)
    L.call("saunet_labels_uncrop_resize", pred.data_ptr(), z, th, tw, bx0, by0, cw, ch, left, top, w, h, W, H, out.data_ptr(), L.stream())
    return out
