"""Volume-level ACDC data pipeline: everything between a NIfTI volume and the per-slice tensors train.py feeds.

Restates /root/reference/data/ac17_dataloader.py:59-164 (`AC17Data`: 5-fold split over data_series.txt, in-plane re-scaling to 1.25 mm,
per-slice min-shift / integer cast / crop-pad / flips / rotation / gamma / z-score, ONCE at load time), :175-229 (`AC17_2DLoad`: every slice of
every volume cached in RAM, 50 % elastic deformation re-drawn per access, image replicated x3, mask -> (labels, edges)) and
data/test_loader.py:19-104 (`AC17Test`: 1.5 mm, no labels).  Volumes enter as numpy arrays [H, W, Z] + the in-plane voxel size (what
`nibabel.load(...).get_data()` and `header['pixdim'][1]` deliver; nibabel is not needed here), so the pipeline is testable on synthetic volumes.

The re-scaling is the one step the reference delegates to a third-party library absent from this image: `skimage.transform.rescale(vol,
[r, r, 1], order=1|0, preserve_range=True, mode='constant')` (unpinned version; restated from the published algorithm, see `rescale_volume`).
"""
import numpy as np
import torch

import os

from . import augment as A
from . import data as sdata
from . import nifti


# ------------------------------------------------------------------------------------------------ fold split
def read_series(path):
    """data/data_series.txt: one "<patient> <frame>" pair per line, 200 lines in a fixed shuffled order (ac17_dataloader.py:84-88)"""
    out = []
    with open(path) as f:
        for line in f:
            t = line.split()
            if len(t) >= 2:
                out.append((int(t[0]), int(t[1])))
    return out


def fold_split(series, split="train", k=5, k_split=1):
    """AC17Data.read_files (:80-98): lines [(k_split-1)*len, k_split*len) with len = int(200 / k) are the validation fold, all other lines the
    training set -- in file order.  (The reference hard-codes 200; here len(series) takes its place, equal for the real file.)"""
    n = int(len(series) / k)
    lo, hi = (int(k_split) - 1) * n, int(k_split) * n
    if split == "val":
        return [s for i, s in enumerate(series) if lo <= i < hi]
    if split == "train":
        return [s for i, s in enumerate(series) if not (lo <= i < hi)]
    raise ValueError("split is 'train' or 'val'")


def volume_name(patient, frame):
    return "patient%03d/patient%03d_frame%02d" % (patient, patient, frame)


# ------------------------------------------------------------------------------------------------ NIfTI files
def load_training_volume(img_root, seg_root, patient, frame):
    """ac17_dataloader.py:107-114: `<root>/patientNNN/patientNNN_frameFF.nii.gz` and its `_gt` label volume -> (img [H, W, Z], seg [H, W, Z],
    pixdim[1]).  (The reference joins the patient directory with a BACKSLASH, a Windows path; both spellings resolve to the same file there.)"""
    name = volume_name(patient, frame)
    img, pix = nifti.load_volume(os.path.join(img_root, name + ".nii.gz"))
    seg, _ = nifti.load_volume(os.path.join(seg_root, name + "_gt.nii.gz"))
    if seg.shape != img.shape:
        raise ValueError("%s: label volume %s does not match the image %s" % (name, seg.shape, img.shape))
    return np.array(img), np.array(seg), pix


def load_test_volume(img_root, patient, frame):
    """data/test_loader.py:46-51 -> (img [H, W, Z], pixdim[1])"""
    img, pix = nifti.load_volume(os.path.join(img_root, volume_name(patient, frame) + ".nii.gz"))
    return np.array(img), pix


def load_fold(img_root, seg_root, series, split="train", k=5, k_split=1):
    """The `volumes` mapping build_cache expects, read from disk for one fold: (patient, frame) -> (img, seg, pix_dim)."""
    return {key: load_training_volume(img_root, seg_root, *key) for key in fold_split(series, split, k, k_split)}


# ------------------------------------------------------------------------------------------------ in-plane re-scaling
def _resize_axis_linear(a, n_out, axis, edge_mode="blend"):
    """order-1 resize along one axis, skimage / scipy convention: output pixel centre o maps to source coordinate (o + 0.5) * n_in / n_out - 0.5;
    mode='constant': samples outside the array are 0.  edge_mode 'blend' interpolates between the edge sample and that zero for coordinates
    in (-1, 0) and (n-1, n) (scipy's 'grid-constant', skimage's own 2-D warp); 'cval' returns 0 for every coordinate outside [0, n-1]
    (`ndi.map_coordinates(mode='constant')`, the n-D path of skimage 0.15-0.18)."""
    n_in = a.shape[axis]
    src = (np.arange(n_out) + 0.5) * (n_in / float(n_out)) - 0.5
    i0 = np.floor(src).astype(np.int64); f = src - i0
    outside = (src < 0) | (src > n_in - 1)
    a = np.moveaxis(a, axis, 0)
    pad = np.concatenate([np.zeros((1,) + a.shape[1:], a.dtype), a, np.zeros((1,) + a.shape[1:], a.dtype)], 0)     # index -1 and n_in read 0
    lo = pad[np.clip(i0 + 1, 0, n_in + 1)]; hi = pad[np.clip(i0 + 2, 0, n_in + 1)]
    shape = (-1,) + (1,) * (a.ndim - 1)
    out = lo * (1.0 - f).reshape(shape) + hi * f.reshape(shape)
    if edge_mode == "cval":
        out = np.where(outside.reshape(shape), 0.0, out)
    elif edge_mode != "blend":
        raise ValueError("edge_mode is 'blend' or 'cval'")
    return np.moveaxis(out, 0, axis)


def _gauss1d_zero(a, sigma, axis):
    if sigma <= 0:
        return a
    w = A.gaussian_weights(sigma)
    r = len(w) // 2
    a = np.moveaxis(a, axis, 0)
    p = np.concatenate([np.zeros((r,) + a.shape[1:]), a, np.zeros((r,) + a.shape[1:])], 0)
    out = sum(w[k] * p[k:k + a.shape[0]] for k in range(2 * r + 1))
    return np.moveaxis(out, 0, axis)


def rescale_volume(vol, pix_dim, target_mm=1.25, order=1, anti_aliasing=None, anti_aliasing_labels=False, edge_mode="blend"):
    """`transform.rescale(vol, [r, r, 1], order, preserve_range=True, multichannel=False, mode='constant')` with r = pix_dim / target_mm
    (ac17_dataloader.py:112-131, test_loader.py:55-64): output in-plane shape round(n * r); order 1 = separable linear interpolation at the
    pixel-centre-aligned coordinates, zeros outside the volume; order 0 = nearest (the source index under the output pixel's centre); the slice
    axis is untouched.  anti_aliasing=None follows skimage 0.15-0.18 (the versions that still accept `multichannel=`): a Gaussian pre-filter
    with sigma = (1/r - 1) / 2 when an order-1 image is SHRUNK (r < 1), none otherwise and never for labels.  UNPINNED: skimage is not
    installed here and the reference pins no version; tests/test_acdc.py holds known-answer vectors of this restatement.
    Two DELIBERATE deviations from what skimage 0.15-0.18 literally executes, each with a switch that reproduces the library:
    (1) those versions Gaussian-filter EVERY input that is shrunk, the order-0 label volume included (labels 0..3 blurred, then sampled at the
    nearest voxel and truncated to uint8 -- class boundaries erode towards the smaller label); here labels are sampled unfiltered unless
    `anti_aliasing_labels=True`; (2) their n-D path samples with `ndi.map_coordinates(mode='constant')`, which returns 0 for coordinates
    outside [0, n-1] instead of blending the edge sample towards 0: `edge_mode='cval'` reproduces that, the default 'blend' keeps the
    one-pixel border of an up-sampled volume (spacing > target) from going dark."""
    vol = np.asarray(vol)
    r = float(pix_dim) / float(target_mm)
    h, w = vol.shape[0], vol.shape[1]
    ho, wo = int(np.round(h * r)), int(np.round(w * r))
    if order == 0:
        if anti_aliasing_labels and (ho < h or wo < w):
            f = _gauss1d_zero(vol.astype(np.float64), max(0.0, (h / float(ho) - 1.0) / 2.0), 0)
            vol = _gauss1d_zero(f, max(0.0, (w / float(wo) - 1.0) / 2.0), 1)
        iy = np.minimum(np.floor((np.arange(ho) + 0.5) * h / ho).astype(np.int64), h - 1)
        ix = np.minimum(np.floor((np.arange(wo) + 0.5) * w / wo).astype(np.int64), w - 1)
        return vol[iy][:, ix].astype(np.float64)
    out = vol.astype(np.float64)
    if anti_aliasing is None:
        anti_aliasing = True
    if anti_aliasing:
        out = _gauss1d_zero(out, max(0.0, (h / float(ho) - 1.0) / 2.0), 0)
        out = _gauss1d_zero(out, max(0.0, (w / float(wo) - 1.0) / 2.0), 1)
    out = _resize_axis_linear(out, ho, 0, edge_mode)
    return _resize_axis_linear(out, wo, 1, edge_mode)


# ------------------------------------------------------------------------------------------------ per-volume preparation (load time)
def prepare_training_volume(img, seg, pix_dim, size=256, rng=None, degree=180.0, gamma_range=(0.5, 2.0), target_mm=1.25):
    """AC17Data.__getitem__ (:100-164) with train.py:236's augmentations: re-scale, then per slice  min-shift (only when the minimum is
    positive) -> uint32 / uint8 cast -> PaddingCenterCrop(size) -> horizontal / vertical flip (p = 0.5 each) -> rotation by U(-degree, degree)
    -> gamma (augment_gamma: gamma ~ U(0.5, 1) or U(1, 2) with equal probability, :22-37) -> z-score.
    rng=None (validation: train.py:245 uses PaddingCenterCrop only, but gamma + z-score still run, :143-148 -- the gamma draw needs an rng
    there too, so validation passes rng as well and degree=0, no flips by flips=False) -> see `prepare_volume`.
    -> (img [size, size, Z] float64, seg [size, size, Z] float64)"""
    return prepare_volume(img, seg, pix_dim, size, rng, flips=True, degree=degree, gamma_range=gamma_range, target_mm=target_mm)


def _draw_gamma(rng, gamma_range):
    # augment_gamma (:26-34): below-one and above-one gammas are equally likely
    if rng.random() < 0.5 and gamma_range[0] < 1:
        return rng.uniform(gamma_range[0], 1.0)
    return rng.uniform(max(gamma_range[0], 1.0), gamma_range[1])


def prepare_volume(img, seg, pix_dim, size=256, rng=None, flips=False, degree=0.0, gamma_range=(0.5, 2.0), target_mm=1.25, gamma=True):
    img = rescale_volume(img, pix_dim, target_mm, order=1)
    seg_r = rescale_volume(seg, pix_dim, target_mm, order=0) if seg is not None else None
    z = img.shape[2]
    img_c = np.zeros((size, size, z)); seg_c = np.zeros((size, size, z))
    for k in range(z):
        sl = img[:, :, k].copy()
        if sl.min() > 0:
            sl -= sl.min()
        a = A.center_crop_pad(sl.astype(np.uint32), size).astype(np.float64)          # the cast truncates like numpy's astype(np.uint32)
        m = A.center_crop_pad(seg_r[:, :, k].astype(np.uint8), size).astype(np.float64) if seg_r is not None else np.zeros((size, size))
        if flips and rng is not None:
            hf, vf = rng.random() < 0.5, rng.random() < 0.5
            a, m = A.flip(a, hf, vf), A.flip(m, hf, vf)
        if degree and rng is not None:
            a, m = A.rotate(a, m, rng.uniform(-degree, degree))
        if gamma and rng is not None:
            a = A.gamma_curve(a, _draw_gamma(rng, gamma_range))
        img_c[:, :, k] = A.zscore(a)
        seg_c[:, :, k] = m
    return img_c, seg_c


def prepare_test_volume(img, pix_dim, size=256, target_mm=1.5, img_norm=True):
    """AC17Test.__getitem__ (test_loader.py:43-104): 1.5 mm, min-shift, uint32 cast, PaddingCenterCropTest, z-score; no labels, no randomness.
    -> (img [size, size, Z], post_scale_shape) -- the shape test_and_pack needs to undo the crop and the re-scaling (postprocess.py)."""
    res = rescale_volume(img, pix_dim, target_mm, order=1)
    z = res.shape[2]
    out = np.zeros((size, size, z))
    for k in range(z):
        sl = res[:, :, k].copy()
        if sl.min() > 0:
            sl -= sl.min()
        a = A.center_crop_pad(sl.astype(np.uint32), size).astype(np.float64)
        out[:, :, k] = A.zscore(a) if img_norm else a
    return out, res.shape


# ------------------------------------------------------------------------------------------------ slice cache
class SliceCache(torch.utils.data.Dataset):
    """AC17_2DLoad (:175-229): every slice of every prepared volume kept in RAM; per access, training slices are elastically deformed with
    probability 0.5 (alpha 500, sigma 20, image and mask together, order 1, edge replication), the image is replicated to 3 channels and the
    mask becomes (labels, radius-2 distance-transform edges).  `volumes`: iterable of (name, img [S, S, Z], seg [S, S, Z]) as prepare_volume
    returns them."""

    def __init__(self, volumes, split="train", deform=True, seed=None):
        self.split, self.deform = split, deform
        self.seed = seed
        self._rng, self._rng_key = None, ()
        self.data = []
        for name, img, seg in volumes:
            for x in range(img.shape[-1]):
                self.data.append({"image": torch.from_numpy(np.ascontiguousarray(img[:, :, x])).float(),
                                  "mask": torch.from_numpy(np.ascontiguousarray(seg[:, :, x])).long(), "name": "%s_z%d" % (name, x)})

    @property
    def rng(self):
        """The generator of the CURRENT process.  The reference draws from the `random` / `np.random` globals, which torch re-seeds per DataLoader
        worker and per epoch; a generator stored in the Dataset would be copied into every worker with the same state (and re-copied, never
        advanced, each epoch), so all workers would replay one sequence of deform decisions and displacement fields.  Inside a
        worker the generator is therefore derived from `get_worker_info().seed` (= base seed of this epoch's iterator + worker id), in the
        parent process from `seed`; it is rebuilt whenever the worker identity changes."""
        info = torch.utils.data.get_worker_info()
        key = None if info is None else (info.id, info.seed)
        if self._rng is None or key != self._rng_key:
            if info is None:
                self._rng = np.random.default_rng(self.seed)
            else:
                self._rng = np.random.default_rng([0 if self.seed is None else int(self.seed), int(info.seed) & 0xffffffffffffffff])
            self._rng_key = key
        return self._rng

    def __getstate__(self):
        d = dict(self.__dict__); d["_rng"], d["_rng_key"] = None, ()      # never ship generator state to a worker
        return d

    def __len__(self):
        return len(self.data)

    def __getitem__(self, i):
        e = self.data[i]
        if self.split == "train":
            img, seg = e["image"].double().numpy(), e["mask"].double().numpy()
            if self.deform and self.rng.uniform(0.0, 1.0) <= 0.5:
                h, w = img.shape
                red = A.elastic_deform(np.stack([img, seg], 2), self.rng.random((h, w)), self.rng.random((h, w)))
                img, seg = red[:, :, 0], red[:, :, 1]
            edges = torch.from_numpy(sdata.mask_to_edges(seg))
            return {"image": torch.from_numpy(np.repeat(img[None], 3, 0)).float(), "mask": (torch.from_numpy(seg), edges), "name": e["name"]}
        img = e["image"].unsqueeze(0).repeat(3, 1, 1).float()
        return {"image": img, "mask": (e["mask"], torch.from_numpy(sdata.mask_to_edges(e["mask"].numpy()))), "name": e["name"]}


def build_cache(volumes, series, split="train", k=5, k_split=1, size=256, seed=304, deform=True):
    """volumes: mapping (patient, frame) -> (img [H, W, Z], seg [H, W, Z], pix_dim).  The fold's volumes are prepared once (training: flips,
    rotation, gamma drawn per slice from `seed`; validation: crop / pad + gamma + z-score, train.py:245-249) and cached slice by slice."""
    rng = np.random.default_rng(seed)
    prepared = []
    for key in fold_split(series, split, k, k_split):
        img, seg, pix = volumes[key]
        a, m = prepare_volume(img, seg, pix, size, rng, flips=split == "train", degree=180.0 if split == "train" else 0.0)
        prepared.append((volume_name(*key), a, m))
    return SliceCache(prepared, split, deform, seed)
