"""Training-time augmentation of ACDC slices -- host restatement + device pipeline.

Per-slice chain of the reference (train.py:236 `Compose([PaddingCenterCrop(256), RandomHorizontallyFlip(), RandomVerticallyFlip(),
RandomRotate(180)])`, data/ac17_dataloader.py:139-150 gamma + z-score, :196-216 50 % elastic deformation, :254-258 edge ground truth):

    crop/pad to S x S  ->  flips (p = 0.5 each)  ->  rotation by U(-180, 180) degrees (bilinear image / nearest mask, fill 0)
    ->  gamma curve  ->  z-score  ->  [p = 0.5] elastic deformation(alpha 500, sigma 20) of image AND mask (order 1, mode 'nearest')
    ->  mask_to_edges

The numpy functions below restate each step and are pinned by tests/golden/augment.npz, produced by the reference's OWN functions
(oracle/make_golden_augment.py) -- except the rotation, which the reference delegates to torchvision.transforms.functional.affine
(not installed here; unpinned version): it is restated from torchvision's inverse-affine formula (centre = S/2 + 0.5 as torchvision
<= 0.5 computes it) evaluated the way PIL's AFFINE transform does (pixel centres, BILINEAR with edge-clipped neighbours, NEAREST =
floor, fill 0 outside), and differs from PIL in one documented way: no rounding of the interpolated intensities to integers.
`DeviceAugmenter` runs the same chain on the GPU (csrc/augment.hip) over a zero-padded batch of raw slices; random draws come
from a numpy Generator on the host (a few scalars per slice) and a counter-hash generator on the device (the noise fields).
"""
import ctypes as C
import math

import numpy as np
import torch

from . import lib as L
from . import functional as HF


# ------------------------------------------------------------------------------------------------ host restatement
def crop_offset(n, t):
    """source index = output index + crop_offset along one axis (PaddingCenterCrop, augmentations.py:236-264): centre crop start
    `int(round((n - t) / 2.))` (Python 3 rounds halves to even) minus the zero padding `max(t - n, 0) // 2` put in front"""
    return max(int(round((n - t) / 2.0)), 0) - max(t - n, 0) // 2


def center_crop_pad(a, size):
    """PaddingCenterCrop(size) of one 2-D array (zeros where the slice is smaller than the crop)"""
    h, w = a.shape
    oy, ox = crop_offset(h, size), crop_offset(w, size)
    out = np.zeros((size, size), a.dtype)
    ys, xs = np.arange(size) + oy, np.arange(size) + ox
    vy, vx = (ys >= 0) & (ys < h), (xs >= 0) & (xs < w)
    out[np.ix_(vy, vx)] = a[np.ix_(ys[vy], xs[vx])]
    return out


def flip(a, hflip, vflip):
    if hflip:
        a = a[:, ::-1]
    if vflip:
        a = a[::-1, :]
    return a


def rotate(img, seg, degrees):
    """RandomRotate's tf.affine(angle) for a square S x S pair: image BILINEAR, mask NEAREST, fill 0 (see the module docstring)."""
    s = img.shape[0]
    a = math.radians(degrees)
    ca, sa = math.cos(a), math.sin(a)
    c = 0.5 * s + 0.5
    yy, xx = np.mgrid[0:s, 0:s].astype(np.float64)
    X, Y = xx + 0.5 - c, yy + 0.5 - c
    xin, yin = ca * X + sa * Y + c, -sa * X + ca * Y + c
    inside = (xin >= 0) & (xin < s) & (yin >= 0) & (yin < s)
    xi, yi = np.clip(np.floor(xin).astype(int), 0, s - 1), np.clip(np.floor(yin).astype(int), 0, s - 1)
    out_seg = np.where(inside, seg[yi, xi], 0)
    xf, yf = xin - 0.5, yin - 0.5
    x0, y0 = np.floor(xf).astype(int), np.floor(yf).astype(int)
    dx, dy = xf - x0, yf - y0
    xa, xb, ya, yb = np.clip(x0, 0, s - 1), np.clip(x0 + 1, 0, s - 1), np.clip(y0, 0, s - 1), np.clip(y0 + 1, 0, s - 1)
    im = img.astype(np.float64)
    r0 = im[ya, xa] + (im[ya, xb] - im[ya, xa]) * dx
    r1 = im[yb, xa] + (im[yb, xb] - im[yb, xa]) * dx
    out_img = np.where(inside, r0 + (r1 - r0) * dy, 0.0)
    return out_img, out_seg


def gamma_curve(x, gamma, epsilon=1e-7):
    """augment_gamma's transform for a drawn gamma (ac17_dataloader.py:35-37)"""
    mn = x.min(); rng = x.max() - mn
    return np.power((x - mn) / float(rng + epsilon), gamma) * rng + mn


def zscore(x):
    return (x - x.mean()) / (x.std() + 1e-10)          # ac17_dataloader.py:146-148


def gaussian_weights(sigma, truncate=4.0):
    r = int(truncate * float(sigma) + 0.5)
    k = np.arange(-r, r + 1)
    w = np.exp(-0.5 * (k / float(sigma)) ** 2)
    return w / w.sum()


def gaussian_filter_zero(a, sigma):
    """scipy.ndimage.gaussian_filter(a, sigma, mode='constant', cval=0): separable, zero outside"""
    w = gaussian_weights(sigma)
    r = len(w) // 2
    out = a.astype(np.float64)
    for axis in (0, 1):
        pad = [(0, 0), (0, 0)]; pad[axis] = (r, r)
        p = np.pad(out, pad)
        acc = np.zeros_like(out)
        for k in range(2 * r + 1):
            sl = [slice(None), slice(None)]; sl[axis] = slice(k, k + out.shape[axis])
            acc += w[k] * p[tuple(sl)]
        out = acc
    return out


def elastic_deform(stack, u1, u2, alpha=500.0, sigma=20.0):
    """random_elastic_deformation (ac17_dataloader.py:260-287) for given uniform fields u1, u2 in [0, 1): rows move by the blurred
    first field, columns by the second; order-1 interpolation with edge replication for every channel of `stack` [H, W, C]."""
    h, w, _ = stack.shape
    dr = gaussian_filter_zero(2 * u1 - 1, sigma) * alpha
    dc = gaussian_filter_zero(2 * u2 - 1, sigma) * alpha
    rr, cc = np.mgrid[0:h, 0:w].astype(np.float64)
    r = np.clip(rr + dr, 0, h - 1); c = np.clip(cc + dc, 0, w - 1)
    r0, c0 = np.floor(r).astype(int), np.floor(c).astype(int)
    r1, c1 = np.minimum(r0 + 1, h - 1), np.minimum(c0 + 1, w - 1)
    fr, fc = (r - r0)[..., None], (c - c0)[..., None]
    s = stack.astype(np.float64)
    return (s[r0, c0] * (1 - fc) + s[r0, c1] * fc) * (1 - fr) + (s[r1, c0] * (1 - fc) + s[r1, c1] * fc) * fr


# ------------------------------------------------------------------------------------------------ device pipeline
class GeoParams(C.Structure):
    _fields_ = [("h", C.c_int32), ("w", C.c_int32), ("oy", C.c_int32), ("ox", C.c_int32), ("hflip", C.c_int32), ("vflip", C.c_int32),
                ("rotate", C.c_int32), ("cosa", C.c_float), ("sina", C.c_float)]


class DeviceAugmenter:
    def __init__(self, size=256, degree=180.0, deform=True, alpha=500.0, sigma=20.0, seed=304, gamma_range=(0.5, 2.0)):
        self.size, self.degree, self.deform, self.alpha, self.sigma = int(size), float(degree), bool(deform), float(alpha), float(sigma)
        self.gamma_range = gamma_range
        self.rng = np.random.default_rng(seed)
        self._weights = None

    def draw(self, batch):
        """the reference's per-slice random decisions (augmentations.py:313, 326, 397; ac17_dataloader.py:31-34, 199)"""
        r = self.rng
        lo, hi = self.gamma_range
        out = []
        for _ in range(batch):
            g = r.uniform(lo, 1.0) if (r.random() < 0.5 and lo < 1) else r.uniform(max(lo, 1.0), hi)
            out.append(dict(hflip=bool(r.random() < 0.5), vflip=bool(r.random() < 0.5), angle=float(r.random() * 2 * self.degree - self.degree),
                            gamma=float(g), deform=bool(self.deform and r.uniform(0, 1.0) <= 0.5), noise_seed=int(r.integers(0, 2 ** 62))))
        return out

    def _gauss(self, device):
        if self._weights is None or self._weights[1].device != torch.device(device):
            w = gaussian_weights(self.sigma)
            r = len(w) // 2
            self._weights = (r, torch.tensor(w[r:], dtype=torch.float32, device=device))
        return self._weights

    def __call__(self, images, masks, params=None, noise=None):
        """images / masks: lists of 2-D arrays (one raw re-scaled slice each, any sizes).  Returns the loader's batch format:
        {"image": float32 [B,3,S,S], "mask": (int64 [B,S,S], float32 [B,1,S,S] edges)} on the GPU.  `noise`: optional
        (u1, u2) uniform fields [B,S,S] to use instead of the device generator (tests)."""
        if not torch.cuda.is_available():
            raise RuntimeError("DeviceAugmenter runs on the GPU through libsaunet_hip.so (no CPU fallback)")
        B, S = len(images), self.size
        params = params if params is not None else self.draw(B)
        dev = torch.device("cuda", torch.cuda.current_device())
        hm, wm = max(a.shape[0] for a in images), max(a.shape[1] for a in images)
        stage = np.zeros((2, B, hm, wm), np.float32)
        gp = (GeoParams * B)()
        for b, (im, mk, p) in enumerate(zip(images, masks, params)):
            h, w = im.shape
            stage[0, b, :h, :w] = im; stage[1, b, :h, :w] = mk
            a = math.radians(p["angle"])
            gp[b] = GeoParams(h, w, crop_offset(h, S), crop_offset(w, S), int(p["hflip"]), int(p["vflip"]), int(p["angle"] != 0.0), math.cos(a), math.sin(a))
        raw = torch.from_numpy(stage).to(dev, non_blocking=True)
        gpd = torch.frombuffer(bytearray(bytes(gp)), dtype=torch.uint8).to(dev)
        st = L.stream()
        img = torch.empty((B, S, S), dtype=torch.float32, device=dev)
        seg = torch.empty((B, S, S), dtype=torch.float32, device=dev)
        L.call("saunet_augment_geometric", raw[0].data_ptr(), raw[1].data_ptr(), B, hm, wm, gpd.data_ptr(), S, img.data_ptr(), seg.data_ptr(), st)
        gam = torch.tensor([p["gamma"] for p in params], dtype=torch.float32).to(dev)
        L.call("saunet_augment_gamma_zscore", img.data_ptr(), B, S * S, gam.data_ptr(), st)
        apply = torch.tensor([int(p["deform"]) for p in params], dtype=torch.int32).to(dev)
        out_img = torch.empty_like(img)
        seg_l = torch.empty((B, S, S), dtype=torch.int64, device=dev)
        seg_e = torch.empty((B, S, S), dtype=torch.int64, device=dev)
        if any(p["deform"] for p in params):
            r, wts = self._gauss(dev)
            disp = torch.empty((2, B, S, S), dtype=torch.float32, device=dev)
            tmp = torch.empty((B, S, S), dtype=torch.float32, device=dev)
            for k in range(2):
                if noise is not None:
                    u = noise[k].to(dev, torch.float32).contiguous()
                else:
                    u = torch.empty((B, S, S), dtype=torch.float32, device=dev)
                    L.call("saunet_uniform_noise", C.c_uint64(params[0]["noise_seed"] + k), u.data_ptr(), B * S * S, st)
                L.call("saunet_gauss_blur", u.data_ptr(), tmp.data_ptr(), disp[k].data_ptr(), B, S, S, wts.data_ptr(), r, 1, C.c_float(self.alpha), st)
            dr, dc = disp[0], disp[1]
        else:
            dr = dc = img       # never read: apply == 0 everywhere
        L.call("saunet_elastic_warp", img.data_ptr(), seg.data_ptr(), dr.data_ptr(), dc.data_ptr(), apply.data_ptr(), B, S, S,
               out_img.data_ptr(), None, seg_l.data_ptr(), seg_e.data_ptr(), st)
        edge = HF.mask_to_edges(seg_e)
        return {"image": out_img.unsqueeze(1).expand(-1, 3, -1, -1), "mask": (seg_l, edge)}
