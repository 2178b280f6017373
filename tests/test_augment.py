"""Training-time augmentation: the numpy restatement of every step against vectors produced by the reference's OWN
functions (tests/golden/augment.npz, oracle/make_golden_augment.py), known-answer tests of the (unpinned) rotation, and on the GPU
the device kernels against the restatement."""
import numpy as np
import pytest
import torch

from tests.golden_util import load


@pytest.fixture(scope="module")
def A():
    import saunet_amd
    from saunet_amd import augment
    return augment


def test_crop_pad_flip_match_reference(A):
    g = load("augment.npz")
    n = len([k for k in g if k.endswith(".out_img")])
    assert n == 8
    for i in range(n):
        hf, vf = [bool(v) for v in g["crop%d.flags" % i]]
        img = A.flip(A.center_crop_pad(g["crop%d.img" % i].astype(np.int64), 64), hf, vf)
        seg = A.flip(A.center_crop_pad(g["crop%d.seg" % i], 64), hf, vf)
        assert np.array_equal(img, g["crop%d.out_img" % i].astype(np.int64)), i
        assert np.array_equal(seg, g["crop%d.out_seg" % i]), i


def test_gamma_matches_reference(A):
    g = load("augment.npz")
    for i in range(4):
        y = A.gamma_curve(g["gamma%d.x" % i], float(g["gamma%d.gamma" % i]))
        assert np.abs(y - g["gamma%d.y" % i]).max() < 1e-9 * np.abs(g["gamma%d.y" % i]).max(), i
    z = A.zscore(y)
    assert abs(z.mean()) < 1e-12 and abs(z.std() - 1) < 1e-9


def test_elastic_deformation_matches_reference(A):
    """same uniform fields -> same displacement (scipy gaussian_filter, zero boundary) and same warped stack (map_coordinates order 1,
    mode 'nearest'); the fixture stores float32, hence the tolerance"""
    g = load("augment.npz")
    for i in range(2):
        u1, u2 = g["deform%d.u1" % i].astype(np.float64), g["deform%d.u2" % i].astype(np.float64)
        dr = A.gaussian_filter_zero(2 * u1 - 1, 20) * 500
        assert np.abs(dr - g["deform%d.dx" % i]).max() < 1e-4 * np.abs(g["deform%d.dx" % i]).max(), i
        out = A.elastic_deform(g["deform%d.in" % i].astype(np.float64), u1, u2)
        assert np.abs(out - g["deform%d.out" % i]).max() < 2e-3 * np.abs(g["deform%d.out" % i]).max(), i


def test_rotation_kats(A):
    """RandomRotate restatement (torchvision affine, centre S/2 + 0.5): 0 degrees is the identity; +-90 / 180 degrees move every pixel to
    its rotated position about that centre exactly (no interpolation weights); mask values stay in the label set for any angle."""
    r = np.random.default_rng(3)
    s = 16
    img = r.integers(1, 100, size=(s, s)).astype(np.float64); seg = r.integers(0, 4, size=(s, s))
    ri, rs = A.rotate(img, seg, 0.0)
    assert np.array_equal(ri, img) and np.array_equal(rs, seg)
    ri, rs = A.rotate(img, seg, 180.0)
    # about c = S/2 + 0.5 in pixel-centre coordinates: out[y][x] = in[S - y][S - x] (row / column 0 fall outside -> fill 0)
    want = np.zeros_like(img); want[1:, 1:] = img[::-1, ::-1][:-1, :-1]
    assert np.allclose(ri, want, atol=1e-9)
    wseg = np.zeros_like(seg); wseg[1:, 1:] = seg[::-1, ::-1][:-1, :-1]
    assert np.array_equal(rs, wseg)
    _, rs = A.rotate(img, seg, 37.0)
    assert set(np.unique(rs)) <= {0, 1, 2, 3}


def test_crop_offset_rounds_half_to_even(A):
    assert [A.crop_offset(n, 256) for n in (256, 257, 258, 259, 261, 300)] == [0, 0, 1, 2, 2, 22]       # Python 3 round(): 0.5 -> 0, 1.5 -> 2, 2.5 -> 2
    assert [A.crop_offset(n, 256) for n in (255, 254, 200)] == [0, -1, -28]                               # zero padding in front


@pytest.mark.gpu
def test_device_augmenter_matches_host_restatement(A):
    """the four device kernels against the numpy chain on ragged slices with every branch exercised (crop / pad, flips, rotation, gamma,
    deformation on / off), the noise fields supplied from the host"""
    S = 64
    aug = A.DeviceAugmenter(size=S, seed=11)
    r = np.random.default_rng(5)
    shapes = [(75, 70), (50, 45), (64, 64), (49, 78), (80, 60), (63, 65)]
    imgs = [r.integers(0, 1500, size=s).astype(np.float32) for s in shapes]
    segs = [r.integers(0, 4, size=s).astype(np.float32) for s in shapes]
    for i, s in enumerate(shapes):                        # smooth blobs so that the interpolated mask has exact-integer plateaus
        yy, xx = np.mgrid[0:s[0], 0:s[1]]
        segs[i] = (((yy - s[0] / 2) ** 2 + (xx - s[1] / 2) ** 2 < (s[0] / 4) ** 2) * 2 + (yy > s[0] * 0.75)).astype(np.float32)
    params = [dict(hflip=bool(i & 1), vflip=bool(i & 2), angle=[0.0, 33.0, -120.0, 180.0, 0.0, 77.5][i], gamma=[0.7, 1.6, 1.0, 0.55, 1.9, 1.2][i],
                   deform=[True, False, True, True, False, True][i], noise_seed=1) for i in range(6)]
    u = r.random((2, 6, S, S)).astype(np.float32)
    out = aug(imgs, segs, params=params, noise=(torch.from_numpy(u[0]), torch.from_numpy(u[1])))
    image, (seg_l, edge) = out["image"], out["mask"]
    assert image.shape == (6, 3, S, S) and seg_l.shape == (6, S, S) and edge.shape == (6, 1, S, S)
    from oracle import saunet_ref as R
    for b in range(6):
        p = params[b]
        im = A.flip(A.center_crop_pad(imgs[b].astype(np.float64), S), p["hflip"], p["vflip"])
        sg = A.flip(A.center_crop_pad(segs[b].astype(np.float64), S), p["hflip"], p["vflip"])
        if p["angle"] != 0.0:
            im, sg = A.rotate(im, sg, p["angle"])
        im = A.zscore(A.gamma_curve(np.asarray(im, np.float64), p["gamma"]))
        if p["deform"]:
            st = A.elastic_deform(np.stack([im, np.asarray(sg, np.float64)], 2), u[0, b].astype(np.float64), u[1, b].astype(np.float64))
            im, sg = st[:, :, 0], st[:, :, 1]
        got = image[b, 0].cpu().numpy().astype(np.float64)
        assert np.abs(got - im).max() < 2e-3 * max(1.0, np.abs(im).max()), (b, np.abs(got - im).max())
        assert torch.equal(image[b, 0], image[b, 1]) and torch.equal(image[b, 0], image[b, 2])
        want_l = np.trunc(sg).astype(np.int64)
        gl = seg_l[b].cpu().numpy()
        # float32 vs float64 interpolation can differ where the interpolated mask sits exactly on an integer: allow a handful of pixels
        assert (gl != want_l).mean() < 2e-3, (b, (gl != want_l).mean())
        exact = np.where((sg == np.trunc(sg)) & (sg >= 1) & (sg <= 3), sg, 0).astype(np.int64)
        want_e = R.mask_to_edges(exact)[0]
        assert (edge[b, 0].cpu().numpy() != want_e).mean() < 5e-3, b


@pytest.mark.gpu
def test_device_noise_is_uniform_and_seeded(A):
    import ctypes as C
    from saunet_amd import lib as L
    n = 1 << 20
    a = torch.empty(n, device="cuda"); b = torch.empty(n, device="cuda"); c = torch.empty(n, device="cuda")
    L.call("saunet_uniform_noise", C.c_uint64(7), a.data_ptr(), n, L.stream())
    L.call("saunet_uniform_noise", C.c_uint64(7), b.data_ptr(), n, L.stream())
    L.call("saunet_uniform_noise", C.c_uint64(8), c.data_ptr(), n, L.stream())
    assert torch.equal(a, b) and not torch.equal(a, c)
    assert float(a.min()) >= 0 and float(a.max()) < 1
    assert abs(float(a.mean()) - 0.5) < 2e-3 and abs(float(a.var()) - 1 / 12) < 2e-3
    hist = torch.histc(a, 16, 0, 1) / n
    assert float((hist - 1 / 16).abs().max()) < 2e-3
    assert abs(float(((a[:-1] - 0.5) * (a[1:] - 0.5)).mean())) < 1e-3          # no lag-1 correlation
