"""ORACLE (test infrastructure, never imported by the product path): the input step in front of SuperPoint, restated on
the CPU with numpy integer arithmetic.

PARITY UNPINNED. The reference performs this step with OpenCV (``cv2`` is absent from this container and from the GPU
box; SURVEY.md F10 / section 8c caveat 4), so neither function below can be checked against the reference here. They
restate OpenCV's 8-bit fixed-point algorithms from its published source (``opencv-python >= 4.5.4.60`` is the version the
reference pins, pyproject.toml:75), scalar code paths:

* ``rgb_to_gray_u8``  -- ``cv.cvtColor(..., COLOR_RGB2GRAY)`` on uint8 as called by ``gtsfm/utils/images.py:15-42``
  (imgproc/src/color_rgb.simd.hpp ``RGB2Gray<uchar>``): 15-bit coefficients ``(R 9798 + G 19235 + B 3735 + 2^14) >> 15``.
* ``resize_inter_cubic_u8`` -- ``cv.resize(..., interpolation=cv.INTER_CUBIC)`` on uint8 as called by
  ``gtsfm/utils/images.py:102-129`` from ``gtsfm/loader/loader_base.py:160-200`` (imgproc/src/resize.cpp): source
  coordinate ``(d + 0.5) scale - 0.5`` in float32, cubic weights with A = -0.75 in float32, weights rounded to 11-bit
  integers, horizontal pass in int32, vertical pass in int32, ``(v + 2^21) >> 22`` saturated to uint8, taps clamped to the
  image (replicate border). OpenCV's SIMD vertical pass sums in float32 and can differ from this scalar form by one grey
  level on rare pixels.
* ``downsampled_size`` -- the loader's target size, ``gtsfm/utils/images.py:150-220``.
"""

from __future__ import annotations

from typing import Tuple

import numpy as np

INTER_RESIZE_COEF_BITS = 11
INTER_RESIZE_COEF_SCALE = 1 << INTER_RESIZE_COEF_BITS


def rgb_to_gray_u8(rgb: np.ndarray) -> np.ndarray:
    """HxWx3 (or x4, alpha ignored) uint8 -> HxW uint8."""
    assert rgb.dtype == np.uint8 and rgb.ndim == 3 and rgb.shape[2] in (3, 4)
    c = rgb[..., :3].astype(np.int64)
    return ((c[..., 0] * 9798 + c[..., 1] * 19235 + c[..., 2] * 3735 + (1 << 14)) >> 15).astype(np.uint8)


def downsampled_size(img_h: int, img_w: int, max_resolution: int) -> Tuple[int, int]:
    """(new_h, new_w): the shorter side becomes ``max_resolution`` when it is larger, the other side is rounded
    (``get_downsampling_factor_per_axis`` / ``get_rescaling_factor_per_axis``)."""
    if min(img_h, img_w) <= max_resolution:
        return img_h, img_w
    if min(img_h, img_w) == img_h:
        new_h = max_resolution
        return new_h, int(np.round(img_w * (new_h / float(img_h))).astype(np.int32))
    new_w = max_resolution
    return int(np.round(img_h * (new_w / float(img_w))).astype(np.int32)), new_w


def _cubic_taps(dst_size: int, src_size: int):
    """Per destination index: first source index (tap 0 = s - 1) and the four 11-bit integer weights."""
    scale = np.float64(src_size) / np.float64(dst_size)
    d = np.arange(dst_size, dtype=np.float64)
    f = ((d + 0.5) * scale - 0.5).astype(np.float32)
    s = np.floor(f).astype(np.int64)
    x = (f - s.astype(np.float32)).astype(np.float32)
    a = np.float32(-0.75)
    one = np.float32(1.0)
    c0 = ((a * (x + one) - np.float32(5) * a) * (x + one) + np.float32(8) * a) * (x + one) - np.float32(4) * a
    c1 = ((a + np.float32(2)) * x - (a + np.float32(3))) * x * x + one
    c2 = ((a + np.float32(2)) * (one - x) - (a + np.float32(3))) * (one - x) * (one - x) + one
    c3 = one - c0 - c1 - c2
    coef = np.stack([c0, c1, c2, c3], 1).astype(np.float32)
    icoef = np.rint(coef * np.float32(INTER_RESIZE_COEF_SCALE)).astype(np.int64)  # cvRound: round half to even
    return s, np.clip(icoef, -32768, 32767)


def resize_inter_cubic_u8(img: np.ndarray, new_h: int, new_w: int) -> np.ndarray:
    """HxW or HxWxC uint8 -> new_h x new_w (x C) uint8."""
    assert img.dtype == np.uint8
    squeeze = img.ndim == 2
    src = img[..., None] if squeeze else img
    h, w, _ = src.shape
    sx, ax = _cubic_taps(new_w, w)
    sy, ay = _cubic_taps(new_h, h)
    s64 = src.astype(np.int64)
    # horizontal pass on every source row: [h][new_w][c]
    hor = np.zeros((h, new_w, src.shape[2]), dtype=np.int64)
    for t in range(4):
        cols = np.clip(sx - 1 + t, 0, w - 1)
        hor += s64[:, cols, :] * ax[:, t][None, :, None]
    out = np.zeros((new_h, new_w, src.shape[2]), dtype=np.int64)
    for t in range(4):
        rows = np.clip(sy - 1 + t, 0, h - 1)
        out += hor[rows] * ay[:, t][:, None, None]
    out = np.clip((out + (1 << (2 * INTER_RESIZE_COEF_BITS - 1))) >> (2 * INTER_RESIZE_COEF_BITS), 0, 255).astype(np.uint8)
    return out[..., 0] if squeeze else out
