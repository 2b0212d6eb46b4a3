"""ctypes wrapper of oracle/resize_ref.c (CPU ORACLE -- test infrastructure only, see the header of the C file)."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, '_build', 'liboracle_resize.so')


def build() -> str:
    src = os.path.join(HERE, 'resize_ref.c')
    if not os.path.exists(LIB) or os.path.getmtime(LIB) < os.path.getmtime(src):
        subprocess.run(['make', '-C', HERE, '-s'], check=True)
    return LIB


_lib = None


def resize_linear_u8(src: np.ndarray, dsize_wh) -> np.ndarray:
    """`cv2.resize(src, (w, h), interpolation=cv2.INTER_LINEAR)` for uint8 HxWxC, by the C restatement."""
    global _lib
    if _lib is None:
        _lib = C.CDLL(build())
        _lib.resize_linear_u8_ref.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int]
    src = np.ascontiguousarray(src, dtype=np.uint8)
    sh, sw, ch = src.shape
    dw, dh = int(dsize_wh[0]), int(dsize_wh[1])
    dst = np.empty((dh, dw, ch), dtype=np.uint8)
    rc = _lib.resize_linear_u8_ref(src.ctypes.data, sh, sw, ch, dst.ctypes.data, dh, dw)
    assert rc == 0, f'resize_linear_u8_ref failed ({rc})'
    return dst
