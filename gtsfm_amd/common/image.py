"""``Image`` boundary type: re-exports ``gtsfm.common.image.Image`` when GTSfM is importable, otherwise a stand-in
with the fields the deep front-end reads (``gtsfm/common/image.py:20-43``: ``value_array`` HxW[xC] uint8, ``mask``,
``height``/``width``/``shape``). GTSfM cannot be imported in the build container (gtsam missing)."""

from __future__ import annotations

from typing import Any, Dict, NamedTuple, Optional, Tuple

import numpy as np

try:  # pragma: no cover
    from gtsfm.common.image import Image  # type: ignore  # noqa: F401
except Exception:  # noqa: BLE001

    class Image(NamedTuple):  # type: ignore[no-redef]
        value_array: np.ndarray
        exif_data: Optional[Dict[str, Any]] = None
        sensor_width_db: Any = None
        file_name: Optional[str] = None
        mask: Optional[np.ndarray] = None

        @property
        def height(self) -> int:
            return self.value_array.shape[0]

        @property
        def width(self) -> int:
            return self.value_array.shape[1]

        @property
        def shape(self) -> Tuple[int, ...]:
            return self.value_array.shape


def rgb_to_gray_u8(value_array: np.ndarray) -> np.ndarray:
    """Grayscale conversion of ``gtsfm/utils/images.py:15-42`` (``cv.cvtColor(..., COLOR_RGB2GRAY)`` on uint8).

    With OpenCV present the reference function itself is used. Without it (this container), OpenCV's 8-bit fixed-point
    formula is restated: ``(R*9798 + G*19235 + B*3735 + 2^14) >> 15`` -- the 15-bit coefficients of
    ``RGB2Gray<uchar>`` in the OpenCV 4.5 line the reference pins (``opencv-python>=4.5.4.60``, pyproject.toml:75; round 1
    used the older 14-bit set 4899 / 9617 / 1868, which differs by one grey level on 0.26 % of random pixels). The same
    arithmetic runs on the device in ``gtsfm_prep_rgb_to_gray_u8``. UNPINNED without cv2 (SURVEY.md section 8c caveat 4):
    synthetic configs feed gray images directly.
    """
    if value_array.ndim == 2:
        return value_array
    if value_array.shape[2] not in (3, 4):
        raise ValueError("Input image dimensions are wrong")
    try:  # pragma: no cover
        import cv2 as cv

        code = cv.COLOR_RGBA2GRAY if value_array.shape[2] == 4 else cv.COLOR_RGB2GRAY
        return cv.cvtColor(value_array, code)
    except ImportError:
        rgb = value_array[..., :3].astype(np.uint32)
        gray = (rgb[..., 0] * 9798 + rgb[..., 1] * 19235 + rgb[..., 2] * 3735 + (1 << 14)) >> 15
        return gray.astype(np.uint8)
