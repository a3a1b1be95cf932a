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

    With OpenCV present the reference function itself is used. Without it (this container), OpenCV's 8-bit
    fixed-point formula is restated: ``(R*4899 + G*9617 + B*1868 + 2^13) >> 14`` (coefficients 0.299/0.587/0.114 in
    Q14). This conversion sits outside the bit-exact contract (SURVEY.md section 8c caveat 4): synthetic configs feed
    gray images directly.
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
        gray = (rgb[..., 0] * 4899 + rgb[..., 1] * 9617 + rgb[..., 2] * 1868 + (1 << 13)) >> 14
        return gray.astype(np.uint8)
