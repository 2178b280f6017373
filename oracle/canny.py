"""ctypes wrapper over oracle/canny.c -- TEST INFRASTRUCTURE (see oracle/__init__.py).

``Canny(img_u8, lo, hi)`` mirrors the call ``cv2.Canny(im_arr[i], 10, 100)`` at
/root/reference/models/models.py:362; ``gray_u8`` mirrors the cast at :359.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_BUILD = os.path.join(_HERE, "_build")
_SO = os.path.join(_BUILD, "liboracle_canny.so")
_lib = None


def build(force=False):
    src = os.path.join(_HERE, "canny.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        os.makedirs(_BUILD, exist_ok=True)
        subprocess.check_call(["gcc", "-O2", "-shared", "-fPIC", "-o", _SO, src])
    return _SO


def _load():
    global _lib
    if _lib is None:
        _lib = ctypes.CDLL(build())
        _lib.saunet_oracle_canny.restype = ctypes.c_int
    return _lib


def Canny(img, lo, hi):
    img = np.ascontiguousarray(img, dtype=np.uint8)
    assert img.ndim == 2
    out = np.empty_like(img)
    rc = _load().saunet_oracle_canny(
        img.ctypes.data_as(ctypes.c_void_p), ctypes.c_int(img.shape[0]), ctypes.c_int(img.shape[1]),
        ctypes.c_int(int(np.floor(lo))), ctypes.c_int(int(np.floor(hi))), out.ctypes.data_as(ctypes.c_void_p))
    if rc != 0:
        raise MemoryError("oracle canny")
    return out


def gray_u8(x):
    """x: float32 [3,H,W] -> uint8 [H,W] (mean over channels, trunc, low 8 bits)."""
    x = np.ascontiguousarray(x, dtype=np.float32)
    assert x.ndim == 3 and x.shape[0] == 3
    out = np.empty(x.shape[1:], dtype=np.uint8)
    _load().saunet_oracle_gray_u8(x.ctypes.data_as(ctypes.c_void_p), ctypes.c_int(x.shape[1]),
                                  ctypes.c_int(x.shape[2]), out.ctypes.data_as(ctypes.c_void_p))
    return out


def canny_batch(x):
    """x: float32 [B,3,H,W] -> float32 [B,1,H,W] with values {0,255} (models.py:359-363)."""
    x = np.asarray(x, dtype=np.float32)
    out = np.zeros((x.shape[0], 1, x.shape[2], x.shape[3]), dtype=np.float32)
    for i in range(x.shape[0]):
        out[i, 0] = Canny(gray_u8(x[i]), 10, 100)
    return out
