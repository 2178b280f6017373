"""NIfTI-1 reader (first item): what the reference obtains from nibabel at data/ac17_dataloader.py:108-113 and
data/test_loader.py:47-51 -- the voxel block indexed [x, y, z] and header pixdim[1].  Known-answer tests on files written BYTE BY BYTE from the
NIfTI-1 layout (nifti1.h: sizeof_hdr @0, dim @40, datatype @70, bitpix @72, pixdim @76, vox_offset @108, scl_slope @112, scl_inter @116,
magic @344), independent of the package's own writer, plus round trips through the writer."""
import gzip
import os
import struct

import numpy as np
import pytest

import saunet_amd  # noqa: F401
from saunet_amd import acdc, nifti


def _hand_file(shape, code, bitpix, payload, pixdim=(1.0, 1.5625, 1.5625, 10.0), endian="<", slope=0.0, inter=0.0, vox_offset=352.0, ext=b"\0\0\0\0"):
    h = bytearray(348)
    struct.pack_into(endian + "i", h, 0, 348)
    struct.pack_into(endian + "8h", h, 40, len(shape), *(list(shape) + [1] * (7 - len(shape))))
    struct.pack_into(endian + "h", h, 70, code)
    struct.pack_into(endian + "h", h, 72, bitpix)
    struct.pack_into(endian + "8f", h, 76, *(list(pixdim) + [0.0] * (8 - len(pixdim))))
    struct.pack_into(endian + "f", h, 108, vox_offset)
    struct.pack_into(endian + "f", h, 112, slope)
    struct.pack_into(endian + "f", h, 116, inter)
    h[344:348] = b"n+1\0"
    return bytes(h) + ext + payload


def test_known_answer_int16_fortran_order_and_pixdim(tmp_path):
    # voxel (x, y, z) holds 100*z + 10*y + x; the file stores x fastest
    nx, ny, nz = 4, 3, 2
    vals = [100 * z + 10 * y + x for z in range(nz) for y in range(ny) for x in range(nx)]
    p = tmp_path / "a.nii"
    p.write_bytes(_hand_file((nx, ny, nz), 4, 16, struct.pack("<%dh" % len(vals), *vals)))
    a, pix = nifti.load_volume(str(p))
    assert a.shape == (4, 3, 2) and a.dtype == np.int16 and pix == 1.5625
    for x, y, z in ((0, 0, 0), (3, 0, 0), (1, 2, 0), (2, 1, 1), (3, 2, 1)):
        assert a[x, y, z] == 100 * z + 10 * y + x


def test_gzip_big_endian_scaling_and_header_extension(tmp_path):
    vals = list(range(24))
    raw = _hand_file((2, 3, 4), 512, 16, struct.pack(">24H", *vals) , endian=">", slope=0.5, inter=-1.0, vox_offset=368.0,
                     ext=b"\1\0\0\0" + b"\0" * 16)            # one 16-byte header extension, voxels at 368
    p = tmp_path / "b.nii.gz"
    with gzip.open(p, "wb") as f:
        f.write(raw)
    a, h = nifti.load(str(p))
    assert h["endian"] == ">" and h["vox_offset"] == 368 and a.dtype == np.float64         # nibabel: scaled data comes back as float64
    assert a[1, 2, 3] == 0.5 * (1 + 2 * 2 + 6 * 3) - 1.0 and a[0, 0, 0] == -1.0
    raw_a, _ = nifti.load(str(p), scaled=False)
    assert raw_a.dtype == np.uint16 and raw_a[1, 0, 0] == 1 and raw_a[0, 1, 0] == 2 and raw_a[0, 0, 1] == 6


def test_uint8_labels_float32_images_and_4d_singleton(tmp_path):
    lab = (np.arange(60) % 4).astype(np.uint8)
    p = tmp_path / "gt.nii.gz"
    with gzip.open(p, "wb") as f:
        f.write(_hand_file((5, 4, 3), 2, 8, lab.tobytes()))
    a, pix = nifti.load_volume(str(p))
    assert a.dtype == np.uint8 and np.array_equal(a, lab.reshape((5, 4, 3), order="F"))
    img = np.linspace(-1, 1, 60, dtype=np.float32)
    q = tmp_path / "im.nii"
    q.write_bytes(_hand_file((5, 4, 3, 1), 16, 32, img.tobytes(), slope=1.0, inter=0.0))       # slope 1 / inter 0 = identity: dtype kept
    b, _ = nifti.load_volume(str(q))
    assert b.dtype == np.float32 and b.shape == (5, 4, 3) and np.array_equal(b, img.reshape((5, 4, 3), order="F"))


@pytest.mark.parametrize("dtype", ["u1", "i2", "i4", "f4", "f8", "u2"])
@pytest.mark.parametrize("ext", [".nii", ".nii.gz"])
def test_writer_reader_round_trip(tmp_path, dtype, ext):
    rng = np.random.default_rng(3)
    a = (rng.random((7, 6, 5)) * 200).astype(dtype)
    p = str(tmp_path / ("v" + ext))
    nifti.save(p, a, pixdim=(1.40625, 1.40625, 5.0))
    b, pix = nifti.load_volume(p)
    assert b.dtype == a.dtype and np.array_equal(a, b) and abs(pix - 1.40625) < 1e-7


def test_errors_are_loud(tmp_path):
    p = tmp_path / "x.nii"
    p.write_bytes(b"\0" * 100)
    with pytest.raises(nifti.NiftiError):
        nifti.load(str(p))
    good = _hand_file((2, 2, 2), 4, 16, b"\0" * 16)
    p.write_bytes(good[:-4])                                   # voxel block cut short
    with pytest.raises(nifti.NiftiError, match="truncated"):
        nifti.load(str(p))
    p.write_bytes(good[:70] + struct.pack("<h", 1792) + good[72:])   # complex128: not a scalar image
    with pytest.raises(nifti.NiftiError, match="datatype"):
        nifti.load(str(p))


def test_acdc_directory_to_slice_cache(tmp_path):
    """The whole of row f4 from files: patientNNN/patientNNN_frameFF(.nii.gz | _gt.nii.gz) -> fold -> re-scale -> prepared slices."""
    series = [(1, 1), (2, 4), (3, 1), (4, 9), (5, 1)]
    rng = np.random.default_rng(0)
    truth = {}
    for pat, fr in series:
        d = tmp_path / ("patient%03d" % pat); d.mkdir()
        img = (rng.random((96, 80, 3)) * 255).astype(np.int16)
        seg = (rng.random((96, 80, 3)) * 4).astype(np.uint8)
        nifti.save(str(d / ("patient%03d_frame%02d.nii.gz" % (pat, fr))), img, pixdim=(1.5625, 1.5625, 10.0))
        nifti.save(str(d / ("patient%03d_frame%02d_gt.nii.gz" % (pat, fr))), seg, pixdim=(1.5625, 1.5625, 10.0))
        truth[(pat, fr)] = (img, seg)
    vols = acdc.load_fold(str(tmp_path), str(tmp_path), series, "val", k=5, k_split=2)
    assert list(vols) == [(2, 4)]
    img, seg, pix = vols[(2, 4)]
    assert np.array_equal(img, truth[(2, 4)][0]) and np.array_equal(seg, truth[(2, 4)][1]) and pix == 1.5625
    cache = acdc.build_cache(vols, series, "val", k=5, k_split=2, size=128, seed=1)
    assert len(cache) == 3 and cache[0]["image"].shape == (3, 128, 128) and cache[0]["name"] == "patient002/patient002_frame04_z0"
    timg, tpix = acdc.load_test_volume(str(tmp_path), 4, 9)
    assert timg.shape == (96, 80, 3) and tpix == 1.5625
    with pytest.raises(FileNotFoundError):
        acdc.load_training_volume(str(tmp_path), str(tmp_path), 9, 1)
