"""Crop preparation of `VitInference.inference` (easy_ViTPose/inference.py:259-266, :314-316):

    bbox (+10 px, clipped) -> crop -> zero-pad to 3:4 (pad_image) -> cv2.resize(.., (192,256), INTER_LINEAR)

`resize_linear_u8` restates OpenCV's 8-bit INTER_LINEAR (opencv-python 4.8, `resize.cpp`): half-pixel
centres, 11-bit fixed-point coefficients (`INTER_RESIZE_COEF_BITS`), horizontal pass into int32, vertical
pass `(((b0*(S0>>4))>>16) + ((b1*(S1>>4))>>16) + 2) >> 2`, and the exact-2x case handled as the 2x2 box
average OpenCV switches to.  It is integer arithmetic, so the HIP kernel (`crop_resize_kernel`) reproduces
it bit for bit; against the real OpenCV binary it is PARITY UNPINNED (cv2 is not installed here).
"""
from __future__ import annotations

import numpy as np

from .configs import IMG_H, IMG_W

COEF_BITS = 11
COEF_ONE = 1 << COEF_BITS


def _axis_coeffs(dsize: int, ssize: int):
    inv = float(dsize) / float(ssize)
    scale = 1.0 / inv                                     # OpenCV: scale_x = 1./inv_scale_x
    d = np.arange(dsize, dtype=np.float64)
    f = ((d + 0.5) * scale - 0.5).astype(np.float32)      # computed in double, stored as float
    s = np.floor(f).astype(np.int64)
    f = (f - s.astype(np.float32)).astype(np.float32)
    lo = s < 0
    f[lo] = 0; s[lo] = 0
    hi = s >= ssize - 1
    f[hi] = 0; s[hi] = ssize - 1
    a1 = np.rint(f * np.float32(COEF_ONE)).astype(np.int64)            # cvRound = round half to even
    a0 = np.rint((np.float32(1.0) - f) * np.float32(COEF_ONE)).astype(np.int64)
    s1 = np.minimum(s + 1, ssize - 1)                     # tap 1 has weight 0 where it would fall outside
    return s, s1, a0, a1, scale


def resize_linear_u8(src: np.ndarray, dsize_wh) -> np.ndarray:
    """`cv2.resize(src, (w, h), interpolation=cv2.INTER_LINEAR)` for uint8 HxWxC images."""
    dw, dh = int(dsize_wh[0]), int(dsize_wh[1])
    sh, sw = src.shape[:2]
    if (sw, sh) == (dw, dh):
        return src
    sx, sx1, ax0, ax1, scale_x = _axis_coeffs(dw, sw)
    sy, sy1, ay0, ay1, scale_y = _axis_coeffs(dh, sh)
    if scale_x == 2.0 and scale_y == 2.0:                 # INTER_LINEAR == fast INTER_AREA at exactly 2x
        s = src.astype(np.int64)
        return ((s[0::2, 0::2] + s[0::2, 1::2] + s[1::2, 0::2] + s[1::2, 1::2] + 2) >> 2).astype(np.uint8)
    s = src.astype(np.int64)
    rows = s[:, sx] * ax0[None, :, None] + s[:, sx1] * ax1[None, :, None]          # [sh, dw, C], scale 2^11
    r0, r1 = rows[sy], rows[sy1]
    out = (((ay0[:, None, None] * (r0 >> 4)) >> 16) + ((ay1[:, None, None] * (r1 >> 4)) >> 16) + 2) >> 2
    return out.astype(np.uint8)


def crop_params(bboxes: np.ndarray, frame_hw, pad_bbox: int = 10, aspect: float = 3 / 4) -> np.ndarray:
    """Per box int32 [x0, y0, cw, ch, left_pad, top_pad, pw, ph]: the padded+clipped box of
    inference.py:261-262 and the zero-pad geometry of `pad_image` (vit_utils/inference.py:41-70)."""
    H, W = frame_hw
    out = np.zeros((len(bboxes), 8), dtype=np.int32)
    for i, b in enumerate(np.asarray(bboxes)[:, :4].round().astype(int)):
        x0, x1 = np.clip([b[0] - pad_bbox, b[2] + pad_bbox], 0, W)
        y0, y1 = np.clip([b[1] - pad_bbox, b[3] + pad_bbox], 0, H)
        cw, ch = int(x1 - x0), int(y1 - y0)
        assert cw > 0 and ch > 0, 'empty box'
        left = top = 0
        pw, ph = cw, ch
        if cw / ch < aspect:
            pw = int(aspect * ch)
            left = (pw - cw) // 2
        else:
            ph = int(cw / aspect)
            top = (ph - ch) // 2
        out[i] = (x0, y0, cw, ch, left, top, pw, ph)
    return out


def prepare_crops_host(frame: np.ndarray, params: np.ndarray) -> np.ndarray:
    """Host restatement of the crop path -> uint8 [n, 256, 192, 3] (what the HIP kernel must reproduce)."""
    out = np.empty((len(params), IMG_H, IMG_W, 3), dtype=np.uint8)
    for i, (x0, y0, cw, ch, left, top, pw, ph) in enumerate(params):
        canvas = np.zeros((ph, pw, 3), dtype=np.uint8)
        canvas[top:top + ch, left:left + cw] = frame[y0:y0 + ch, x0:x0 + cw]
        out[i] = resize_linear_u8(canvas, (IMG_W, IMG_H))
    return out
