"""Test-set post-processing (/root/reference/test_and_pack.py:28-76): the host restatement of undo_crop against
vectors produced by the REAL reference function (tests/golden/undo_crop.npz, oracle/make_golden_postprocess.py), known-answer tests
of the order-0 resize, and (GPU) the one-kernel device path against the host restatement."""
import numpy as np
import pytest
import torch

from tests.golden_util import load


def test_undo_crop_matches_reference_vectors():
    import saunet_amd
    from saunet_amd import postprocess as PP
    g = load("undo_crop.npz")
    n = len([k for k in g if k.endswith(".pred")])
    assert n >= 10
    for i in range(n):
        h, w = [int(v) for v in g["case%d.img_shape" % i]]
        got = PP.undo_crop(np.empty((h, w)), g["case%d.pred" % i])
        assert got.shape == g["case%d.out" % i].shape == (h, w), i
        assert np.array_equal(got, g["case%d.out" % i]), i
    assert [PP.round_num(float(v)) for v in g["round_num.x"]] == [int(v) for v in g["round_num.y"]]


def test_order0_resize_kats():
    """skimage.transform.resize(order=0) = pixel-centre nearest sampling: identity at equal size, exact 2x decimation picks the odd
    samples' left neighbour (floor((i+.5)*2) = 2i+1), 2x up-sampling repeats every sample twice, and a label volume keeps its label set."""
    import saunet_amd
    from saunet_amd import postprocess as PP
    assert list(PP.nearest_index(5, 5)) == [0, 1, 2, 3, 4]
    assert list(PP.nearest_index(3, 6)) == [1, 3, 5]
    assert list(PP.nearest_index(6, 3)) == [0, 0, 1, 1, 2, 2]
    assert list(PP.nearest_index(4, 10)) == [1, 3, 6, 8]
    r = np.random.default_rng(0)
    pred = r.integers(0, 4, size=(256, 256, 3)).astype(np.uint8)
    vol = PP.resample_to_orig((300, 280, 3), (216, 256, 3), pred)
    assert vol.shape == (216, 256, 3) and set(np.unique(vol)) <= {0, 1, 2, 3}
    same = PP.resample_to_orig((256, 256, 3), (256, 256, 3), pred)
    assert np.array_equal(same, pred)


@pytest.mark.gpu
@pytest.mark.parametrize("post,orig,crop", [((300, 280), (216, 256), (256, 256)), ((200, 180), (154, 139), (256, 256)), ((300, 200), (512, 341), (256, 256)),
                                            ((256, 256), (256, 256), (256, 256)), ((129, 127), (97, 240), (128, 128))])
def test_device_uncrop_resize_matches_host(post, orig, crop):
    import saunet_amd
    from saunet_amd import postprocess as PP
    z = 5
    r = np.random.default_rng(sum(post) + sum(orig))
    pred = r.integers(0, 4, size=crop + (z,)).astype(np.uint8)
    want = PP.resample_to_orig(post + (z,), orig + (z,), pred)
    got = PP.resample_to_orig_device(torch.from_numpy(pred).permute(2, 0, 1).contiguous().long().cuda(), post, orig)
    assert got.dtype == torch.uint8 and tuple(got.shape) == (z,) + orig
    assert np.array_equal(got.permute(1, 2, 0).cpu().numpy(), want)
