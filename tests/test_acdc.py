"""Volume-level ACDC pipeline (saunet_amd.acdc) on the CPU:
  * the 5-fold split against the reference's own AC17Data.read_files run on its own data_series.txt (tests/golden/acdc.npz,
    oracle/make_golden_acdc.py);
  * known-answer tests of the in-plane re-scaling restatement (skimage.transform.rescale is not installed: unpinned, see acdc.rescale_volume);
  * per-volume preparation and the slice cache (AC17Data.__getitem__ / AC17_2DLoad semantics) on synthetic volumes;
  * hand-computed affine vectors pinning RandomRotate's restatement (augment.rotate)."""
import os

import numpy as np
import pytest
import torch

import saunet_amd  # noqa: F401
from saunet_amd import acdc, augment, data as sdata

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "acdc.npz")


def test_fold_split_matches_the_reference_read_files():
    g = np.load(GOLD)
    series = [tuple(int(v) for v in row) for row in g["series"]]
    assert len(series) == 200
    for k_split in (1, 2, 3, 4, 5):
        for split in ("train", "val"):
            got = acdc.fold_split(series, split, 5, k_split)
            want = [tuple(int(v) for v in row) for row in g["fold%d.%s" % (k_split, split)]]
            assert got == want, (k_split, split)
    # folds partition the series
    assert sorted(sum((acdc.fold_split(series, "val", 5, k) for k in range(1, 6)), [])) == sorted(series)
    assert acdc.volume_name(7, 1) == "patient007/patient007_frame01"


def test_read_series_roundtrip(tmp_path):
    p = tmp_path / "series.txt"
    p.write_text("33 1\n35 1\n90 4\n\n23 9\n")
    assert acdc.read_series(str(p)) == [(33, 1), (35, 1), (90, 4), (23, 9)]


def test_rescale_known_answers():
    r = np.random.default_rng(3)
    vol = r.uniform(0, 1000, size=(20, 24, 3))
    # ratio 1: identity for both orders
    assert np.allclose(acdc.rescale_volume(vol, 1.25, 1.25, order=1), vol)
    assert np.array_equal(acdc.rescale_volume(np.floor(vol), 1.25, 1.25, order=0), np.floor(vol))
    # output shape = round(n * ratio), slice axis untouched (ACDC pixdim 1.5625 -> 1.25 mm: ratio 1.25)
    up = acdc.rescale_volume(vol, 1.5625, 1.25, order=1)
    assert up.shape == (25, 30, 3)
    # order 1 reproduces a linear ramp exactly in the interior (pixel-centre aligned coordinates) and bleeds towards 0 at the border (mode='constant')
    yy, xx = np.mgrid[0:20, 0:24].astype(np.float64)
    ramp = (3.0 * yy + 2.0 * xx + 5.0)[:, :, None]
    got = acdc.rescale_volume(ramp, 1.5625, 1.25, order=1)[:, :, 0]
    oy = (np.arange(25) + 0.5) * 20 / 25 - 0.5; ox = (np.arange(30) + 0.5) * 24 / 30 - 0.5
    want = 3.0 * oy[:, None] + 2.0 * ox[None, :] + 5.0
    inner = (oy[:, None] >= 0) & (oy[:, None] <= 19) & (ox[None, :] >= 0) & (ox[None, :] <= 23)
    assert np.allclose(got[inner], want[inner]) and inner.sum() > 500
    assert got[0, 0] < want[0, 0]                      # first output pixel centre lies outside the first input centre: mixed with the zero outside
    # order 0 keeps the label set and picks the source pixel under the output pixel's centre
    lab = r.integers(0, 4, size=(20, 24, 2)).astype(np.float64)
    l2 = acdc.rescale_volume(lab, 2.5, 1.25, order=0)           # exact 2x: every label becomes a 2 x 2 block
    assert l2.shape == (40, 48, 2) and np.array_equal(l2, np.repeat(np.repeat(lab, 2, 0), 2, 1))
    # shrinking an image applies the anti-aliasing Gaussian (sigma = (1/r - 1)/2) first: a single bright pixel spreads but keeps (most of) its mass
    imp = np.zeros((32, 32, 1)); imp[16, 16, 0] = 1000.0
    dn = acdc.rescale_volume(imp, 0.625, 1.25, order=1)          # ratio 0.5
    assert dn.shape == (16, 16, 1) and dn.max() < 250.0 and abs(dn.sum() * 4 - 1000.0) < 60.0
    assert acdc.rescale_volume(imp, 0.625, 1.25, order=1, anti_aliasing=False).max() == pytest.approx(250.0)   # plain bilinear: centre of four pixels


def _phantom_volume(z=5, h=200, w=180, seed=1):
    img, seg = [], []
    for k in range(z):
        i, s, _ = sdata.synthetic_batch(1, h, w, seed=seed + k)
        img.append((i[0, 0].numpy() - i[0, 0].numpy().min() + 0.5) * 300.0); seg.append(s[0].numpy())
    return np.stack(img, 2), np.stack(seg, 2).astype(np.float64)


def test_prepare_volume_and_slice_cache_follow_the_loader():
    img, seg = _phantom_volume()
    rng = np.random.default_rng(5)
    a, m = acdc.prepare_volume(img, seg, 1.5625, size=256, rng=rng, flips=True, degree=180.0)
    assert a.shape == (256, 256, 5) and m.shape == (256, 256, 5)
    for k in range(5):                                          # per-slice z-score (ac17_dataloader.py:146-148), labels stay labels
        assert abs(a[:, :, k].mean()) < 1e-9 and abs(a[:, :, k].std() - 1.0) < 1e-6
        assert set(np.unique(m[:, :, k])) <= {0.0, 1.0, 2.0, 3.0}
    # validation preparation is deterministic given the rng (crop / pad + gamma + z-score only) and keeps the geometry: the re-scaled 250 x 225
    # slice is cropped (rows) and zero-padded (columns) around its centre
    v1, s1 = acdc.prepare_volume(img, seg, 1.5625, 256, np.random.default_rng(9))
    v2, _ = acdc.prepare_volume(img, seg, 1.5625, 256, np.random.default_rng(9))
    assert np.array_equal(v1, v2)
    res = acdc.rescale_volume(seg, 1.5625, 1.25, order=0)
    assert res.shape[:2] == (250, 225)
    assert np.array_equal(s1[:, :, 2], augment.center_crop_pad(res[:, :, 2].astype(np.uint8), 256).astype(np.float64))
    # slice cache: names, 3-channel replication, (labels, edges) masks; validation path has no randomness
    cache = acdc.SliceCache([("patient001/patient001_frame01", v1, s1)], split="val")
    assert len(cache) == 5 and cache[3]["name"] == "patient001/patient001_frame01_z3"
    e = cache[3]
    assert e["image"].shape == (3, 256, 256) and torch.equal(e["image"][0], e["image"][2]) and e["image"].dtype == torch.float32
    assert torch.equal(e["mask"][0], torch.from_numpy(s1[:, :, 3]).long())
    assert torch.equal(e["mask"][1], torch.from_numpy(sdata.mask_to_edges(s1[:, :, 3])))
    # training path: about half of the accesses are elastically deformed (p = 0.5), the others return the cached slice itself
    tr = acdc.SliceCache([("v", v1, s1)], split="train", deform=True, seed=11)
    same = sum(bool(torch.equal(tr[i % 5]["image"][0], torch.from_numpy(v1[:, :, i % 5]).float())) for i in range(40))
    assert 10 <= same <= 30
    assert tr[0]["mask"][0].dtype == torch.float64 and tr[0]["mask"][1].shape == (1, 256, 256)       # the loader hands float64 labels to DualLoss
    nodef = acdc.SliceCache([("v", v1, s1)], split="train", deform=False, seed=11)
    assert all(torch.equal(nodef[i]["image"][0], torch.from_numpy(v1[:, :, i]).float()) for i in range(5))


def test_build_cache_uses_the_fold_and_test_volume_is_label_free():
    series = [(1, 1), (2, 1), (3, 1), (4, 1), (5, 1)]
    vols = {}
    for i, key in enumerate(series):
        img, seg = _phantom_volume(z=2, h=120, w=130, seed=10 * i)
        vols[key] = (img, seg, 1.4)
    val = acdc.build_cache(vols, series, "val", k=5, k_split=2, size=128, seed=3)
    assert len(val) == 2 and val[0]["name"].startswith("patient002/")
    trn = acdc.build_cache(vols, series, "train", k=5, k_split=2, size=128, seed=3)
    assert len(trn) == 8 and {e["name"].split("/")[0] for e in trn.data} == {"patient001", "patient003", "patient004", "patient005"}
    img, _ = _phantom_volume(z=3, h=150, w=150)
    out, post = acdc.prepare_test_volume(img, 1.8, size=128, target_mm=1.5)                  # AC17Test: 1.5 mm
    assert post == (180, 180, 3) and out.shape == (128, 128, 3) and abs(out[:, :, 1].std() - 1.0) < 1e-6


def test_rotation_matches_hand_computed_affine_vectors():
    """augment.rotate restates tf.affine(angle) of torchvision <= 0.5 (centre c = S/2 + 0.5, inverse map evaluated at pixel centres the PIL way).
    For S = 4 the inverse map of a 90-degree rotation is  (x_in - 0.5, y_in - 0.5) = (y, 4 - x):  out[y][x] = in[4 - x][y], zero where 4 - x
    leaves the image (x = 0) -- the historical one-pixel offset of that centre.  180 degrees: out[y][x] = in[4 - y][4 - x]."""
    img = np.arange(16, dtype=np.float64).reshape(4, 4) + 1.0
    seg = (np.arange(16).reshape(4, 4) % 4).astype(np.int64)
    want90 = np.zeros((4, 4)); want90s = np.zeros((4, 4), np.int64)
    want180 = np.zeros((4, 4)); want180s = np.zeros((4, 4), np.int64)
    for y in range(4):
        for x in range(4):
            if 0 <= 4 - x <= 3:
                want90[y, x] = img[4 - x, y]; want90s[y, x] = seg[4 - x, y]
            if 0 <= 4 - x <= 3 and 0 <= 4 - y <= 3:
                want180[y, x] = img[4 - y, 4 - x]; want180s[y, x] = seg[4 - y, 4 - x]
    o, s = augment.rotate(img, seg, 90.0)
    assert np.allclose(o, want90, atol=1e-9) and np.array_equal(s, want90s)
    o, s = augment.rotate(img, seg, 180.0)
    assert np.allclose(o, want180, atol=1e-9) and np.array_equal(s, want180s)
    o, s = augment.rotate(img, seg, 0.0)
    assert np.allclose(o, img) and np.array_equal(s, seg)
    # a non-trivial angle, one output pixel by hand: S = 8, 30 degrees, output (x, y) = (5, 2)
    S = 8; c = 4.5; a = np.radians(30.0)
    big = np.random.default_rng(0).uniform(0, 100, size=(S, S))
    X, Y = 5 + 0.5 - c, 2 + 0.5 - c
    xin, yin = np.cos(a) * X + np.sin(a) * Y + c, -np.sin(a) * X + np.cos(a) * Y + c
    xf, yf = xin - 0.5, yin - 0.5
    x0, y0 = int(np.floor(xf)), int(np.floor(yf)); dx, dy = xf - x0, yf - y0
    hand = (big[y0, x0] * (1 - dx) + big[y0, x0 + 1] * dx) * (1 - dy) + (big[y0 + 1, x0] * (1 - dx) + big[y0 + 1, x0 + 1] * dx) * dy
    o, s = augment.rotate(big, (big > 50).astype(np.int64), 30.0)
    assert abs(o[2, 5] - hand) < 1e-9 and s[2, 5] == int(big[int(np.floor(yin)), int(np.floor(xin))] > 50)


def _first_fields(cache, workers, n=8):
    """images of the first n accesses of each epoch, keyed by (epoch, index)"""
    out = []
    dl = torch.utils.data.DataLoader(cache, batch_size=1, shuffle=False, num_workers=workers)
    for epoch in range(2):
        imgs = []
        for i, b in enumerate(dl):
            imgs.append(b["image"][0, 0].clone())
            if i + 1 == n:
                break
        out.append(imgs)
    return out


def test_slice_cache_draws_differ_per_worker_and_per_epoch():
    """a generator stored in the Dataset is copied into every DataLoader worker with one state and re-copied unchanged each
    epoch, so every worker of every epoch would replay the same deform decisions and displacement fields.  The reference draws from the
    random / np.random globals, re-seeded by torch per worker and per epoch (ac17_dataloader.py:196-216).  Here: with two workers, (a) the
    deformed slices the two workers produce for the SAME cached slice differ, (b) the second epoch differs from the first."""
    torch.manual_seed(7)
    img = np.random.default_rng(0).standard_normal((64, 64, 1))
    seg = (np.random.default_rng(1).random((64, 64, 1)) * 4).astype(np.int64).astype(np.float64)
    same_slice = [("v", np.repeat(img, 16, 2), np.repeat(seg, 16, 2))]                 # 16 copies of one slice: worker 0 serves even, worker 1 odd indices
    cache = acdc.SliceCache(same_slice, split="train", deform=True, seed=11)
    base = torch.from_numpy(img[:, :, 0]).float()
    e0, e1 = _first_fields(cache, workers=2, n=16)
    w0 = [x for x in e0[0::2] if not torch.equal(x, base)]
    w1 = [x for x in e0[1::2] if not torch.equal(x, base)]
    assert len(w0) >= 1 and len(w1) >= 1                                                  # p = 0.5 over 8 draws each
    assert not any(torch.equal(a, b) for a in w0 for b in w1)                            # (a) workers do not share a displacement field
    assert [torch.equal(x, base) for x in e0] != [torch.equal(x, base) for x in e1] or not any(
        torch.equal(a, b) for a in e0 for b in e1 if not torch.equal(a, base))           # (b) epochs are not replays
    d1 = [x for x in e1 if not torch.equal(x, base)]
    assert not any(torch.equal(a, b) for a in w0 + w1 for b in d1)
    # single-process loading keeps advancing ONE generator and stays reproducible from the seed
    a = acdc.SliceCache(same_slice, split="train", deform=True, seed=11); b = acdc.SliceCache(same_slice, split="train", deform=True, seed=11)
    assert all(torch.equal(a[i]["image"], b[i]["image"]) for i in range(6))


def test_rescale_switches_reproduce_the_library_literal_behaviour():
    """the two deliberate deviations of rescale_volume from skimage 0.15-0.18 and the switches that undo them."""
    seg = np.zeros((40, 40, 1)); seg[10:30, 10:30] = 3.0
    plain = acdc.rescale_volume(seg, 1.0, 1.25, order=0)
    blur = acdc.rescale_volume(seg, 1.0, 1.25, order=0, anti_aliasing_labels=True)
    assert plain.shape == blur.shape == (32, 32, 1) and set(np.unique(plain)) == {0.0, 3.0}
    assert np.all(blur <= plain + 1e-12) and blur.max() > 2.9 and np.any((blur > 0) & (blur < 3))     # filtered labels erode at the boundary
    assert np.array_equal(acdc.rescale_volume(seg, 1.5, 1.25, order=0, anti_aliasing_labels=True), acdc.rescale_volume(seg, 1.5, 1.25, order=0))  # never when enlarging
    img = np.ones((10, 10, 1))
    up_b = acdc.rescale_volume(img, 1.5, 1.25)                                  # 12 x 12: the outermost samples lie at -0.083 / 9.083
    up_c = acdc.rescale_volume(img, 1.5, 1.25, edge_mode="cval")
    assert abs(up_b[0, 5, 0] - (1 - 1 / 12.0)) < 1e-12 and up_c[0, 5, 0] == 0.0 and up_c[5, 0, 0] == 0.0
    assert np.array_equal(up_b[1:-1, 1:-1], up_c[1:-1, 1:-1]) and np.all(up_c[1:-1, 1:-1] == 1.0)
