"""NIfTI-1 volume reader (and a minimal writer for tests / label export): the first step of the ACDC pipeline.

Replaces what the reference gets from nibabel at /root/reference/data/ac17_dataloader.py:108-113 and data/test_loader.py:47-51:

    img = nibabel.load(path + ".nii.gz");  pix_dim = img.header.structarr['pixdim'][1];  img = np.array(img.get_data())

i.e. the voxel block as an array indexed [x, y, z] (NIfTI stores x fastest = Fortran order), scaled by scl_slope / scl_inter when the header
asks for it (nibabel's `get_data()` semantics: slope 0 or NaN = "no scaling", raw dtype kept), plus `pixdim[1]`, the in-plane voxel size the
re-sampling ratio is computed from.  Single-file NIfTI-1 (`.nii`, `.nii.gz`; magic "n+1") of either byte order; NIfTI-2, the two-file
`.hdr/.img` form and header extensions beyond skipping them are out of scope (ACDC ships single-file NIfTI-1).  nibabel is not installed in
this image and is not needed.
"""
import gzip
import struct

import numpy as np

# NIfTI-1 datatype codes -> numpy (nifti1.h)
_DTYPES = {2: "u1", 4: "i2", 8: "i4", 16: "f4", 64: "f8", 256: "i1", 512: "u2", 768: "u4", 1024: "i8", 1280: "u8"}
_CODES = {np.dtype(v).str[1:]: k for k, v in _DTYPES.items()}


class NiftiError(ValueError):
    pass


def _open(path):
    with open(path, "rb") as f:
        head = f.read(2)
    return gzip.open(path, "rb") if head == b"\x1f\x8b" else open(path, "rb")


def read_header(buf):
    """The fields of the 348-byte NIfTI-1 header the pipeline needs, as a dict (plus 'endian': '<' or '>')."""
    if len(buf) < 348:
        raise NiftiError("truncated NIfTI-1 header (%d bytes)" % len(buf))
    endian = "<"
    if struct.unpack("<i", buf[0:4])[0] != 348:
        if struct.unpack(">i", buf[0:4])[0] != 348:
            raise NiftiError("sizeof_hdr is not 348: not a NIfTI-1 file")
        endian = ">"
    magic = bytes(buf[344:348])
    if magic not in (b"n+1\x00", b"ni1\x00"):
        raise NiftiError("bad magic %r" % magic)
    if magic == b"ni1\x00":
        raise NiftiError("two-file NIfTI (.hdr/.img) is not supported")
    dim = struct.unpack(endian + "8h", buf[40:56])
    datatype, bitpix = struct.unpack(endian + "2h", buf[70:74])
    pixdim = struct.unpack(endian + "8f", buf[76:108])
    vox_offset, slope, inter = struct.unpack(endian + "3f", buf[108:120])
    if not 1 <= dim[0] <= 7:
        raise NiftiError("dim[0] = %d" % dim[0])
    if datatype not in _DTYPES:
        raise NiftiError("unsupported NIfTI datatype code %d" % datatype)
    return {"endian": endian, "dim": dim, "shape": tuple(int(d) for d in dim[1:1 + dim[0]]), "datatype": datatype, "bitpix": bitpix,
            "pixdim": pixdim, "vox_offset": int(vox_offset) if vox_offset >= 352 else 352, "scl_slope": slope, "scl_inter": inter}


def load(path, scaled=True):
    """-> (array, header dict).  The array is indexed [x, y, z(, t)] exactly like `nibabel.load(path).get_data()`: raw dtype when the header
    carries no scaling, float64 `raw * scl_slope + scl_inter` otherwise."""
    with _open(path) as f:
        raw = f.read()
    h = read_header(raw)
    dt = np.dtype(h["endian"] + _DTYPES[h["datatype"]])
    n = int(np.prod(h["shape"], dtype=np.int64))
    off = h["vox_offset"]
    if len(raw) < off + n * dt.itemsize:
        raise NiftiError("voxel block truncated: need %d bytes at offset %d, file has %d" % (n * dt.itemsize, off, len(raw)))
    a = np.frombuffer(raw, dt, n, off).reshape(h["shape"], order="F")
    a = a.astype(dt.newbyteorder("="), copy=True)
    s, i = h["scl_slope"], h["scl_inter"]
    if scaled and s == s and s != 0.0 and not (s == 1.0 and (i == 0.0 or i != i)):
        a = a.astype(np.float64) * float(s) + (float(i) if i == i else 0.0)
    return a, h


def load_volume(path):
    """What the reference reads per file: (voxels [H, W, Z], pixdim[1])."""
    a, h = load(path)
    if a.ndim == 4 and a.shape[3] == 1:
        a = a[..., 0]
    if a.ndim != 3:
        raise NiftiError("expected a 3-D volume, got shape %s" % (a.shape,))
    return a, float(h["pixdim"][1])


def save(path, array, pixdim=(1.0, 1.0, 1.0), scl_slope=0.0, scl_inter=0.0, endian="<"):
    """Minimal single-file NIfTI-1 writer (`.nii` or `.nii.gz` by extension): enough header for `load` and for nibabel."""
    a = np.asarray(array)
    key = a.dtype.str[1:]
    if key not in _CODES:
        raise NiftiError("dtype %s has no NIfTI-1 code" % a.dtype)
    hdr = bytearray(348)
    struct.pack_into(endian + "i", hdr, 0, 348)
    dim = [a.ndim] + list(a.shape) + [1] * (7 - a.ndim)
    struct.pack_into(endian + "8h", hdr, 40, *dim)
    struct.pack_into(endian + "2h", hdr, 70, _CODES[key], a.dtype.itemsize * 8)
    pd = [1.0] + [float(p) for p in pixdim] + [1.0] * (7 - len(pixdim))
    struct.pack_into(endian + "8f", hdr, 76, *pd)
    struct.pack_into(endian + "3f", hdr, 108, 352.0, float(scl_slope), float(scl_inter))
    hdr[123] = 2                                   # xyzt_units: millimetres
    hdr[344:348] = b"n+1\x00"
    body = bytes(hdr) + b"\x00\x00\x00\x00" + a.astype(a.dtype.newbyteorder(endian)).tobytes(order="F")
    opener = gzip.open if str(path).endswith(".gz") else open
    with opener(path, "wb") as f:
        f.write(body)
