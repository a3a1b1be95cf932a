"""Host side of the device input step: what happens to an image between the loader and
SuperPoint's first convolution in the reference --

1. ``LoaderBase.get_image`` downsizes the full-resolution RGB image with ``cv.INTER_CUBIC`` so that its shorter side is at
   most ``max_resolution`` (``gtsfm/loader/loader_base.py:160-200``, ``gtsfm/utils/images.py:102-129,150-220``),
2. the SuperPoint wrapper converts RGB -> gray with ``cv.cvtColor`` (``gtsfm/frontend/detector_descriptor/superpoint.py:73``,
   ``gtsfm/utils/images.py:15-42``) and divides by 255 (done inside the first-layer kernel here)

-- as two HIP kernels on uint8 (``imageprep_kernels.hip``). PyTorch provides device memory and streams only. OpenCV's
8-bit fixed-point arithmetic is restated, not called (cv2 is absent): PARITY UNPINNED, see ``oracle/imageprep_oracle.py``.
"""

from __future__ import annotations

from typing import Dict, Optional, Tuple

import numpy as np
import torch

from gtsfm_amd.runtime import lib as _lib
from gtsfm_amd.runtime.superpoint_engine import require_gpu


def downsampled_size(img_h: int, img_w: int, max_resolution: int) -> Tuple[int, int]:
    """Target (height, width) of ``get_downsampling_factor_per_axis`` (``gtsfm/utils/images.py:189-220``): unchanged when
    the shorter side already fits, otherwise shorter side = ``max_resolution`` and the other side rounded."""
    if min(img_h, img_w) <= max_resolution:
        return img_h, img_w
    if min(img_h, img_w) == img_h:
        return max_resolution, int(np.round(img_w * (max_resolution / float(img_h))).astype(np.int32))
    return int(np.round(img_h * (max_resolution / float(img_w))).astype(np.int32)), max_resolution


class ImagePrep:
    """Device-resident resize + gray conversion; tap tables are cached per (source size, target size)."""

    def __init__(self, device: Optional[torch.device] = None):
        self.device = require_gpu(device)
        self._lib = _lib.load()
        self._taps: Dict[Tuple[int, int], Tuple[torch.Tensor, torch.Tensor]] = {}

    def _axis_taps(self, dst: int, src: int) -> Tuple[torch.Tensor, torch.Tensor]:
        key = (dst, src)
        if key not in self._taps:
            first = np.empty(dst, dtype=np.int32)
            weights = np.empty((dst, 4), dtype=np.int16)
            _lib.check(self._lib.gtsfm_prep_cubic_taps(dst, src, first.ctypes.data, weights.ctypes.data), "gtsfm_prep_cubic_taps")
            if len(self._taps) > 32:
                self._taps.clear()
            self._taps[key] = (torch.from_numpy(first).to(self.device), torch.from_numpy(weights).to(self.device))
        return self._taps[key]

    def resize_cubic(self, image: torch.Tensor, new_h: int, new_w: int) -> torch.Tensor:
        """image [H,W] or [H,W,C] uint8 on the device -> [new_h,new_w(,C)] uint8 (``cv.resize(..., INTER_CUBIC)``)."""
        assert image.dtype == torch.uint8 and image.is_cuda and image.is_contiguous() and image.dim() in (2, 3)
        h, w = int(image.shape[0]), int(image.shape[1])
        c = 1 if image.dim() == 2 else int(image.shape[2])
        if (h, w) == (new_h, new_w):
            return image
        xofs, xw = self._axis_taps(new_w, w)
        yofs, yw = self._axis_taps(new_h, h)
        out = torch.empty((new_h, new_w) if image.dim() == 2 else (new_h, new_w, c), dtype=torch.uint8, device=image.device)
        rc = self._lib.gtsfm_prep_resize_cubic_u8(image.data_ptr(), h, w, c, xofs.data_ptr(), xw.data_ptr(), yofs.data_ptr(), yw.data_ptr(),
                                                  out.data_ptr(), new_h, new_w, torch.cuda.current_stream(image.device).cuda_stream)
        _lib.check(rc, "gtsfm_prep_resize_cubic_u8")
        return out

    def rgb_to_gray(self, image: torch.Tensor) -> torch.Tensor:
        """image [..., H, W, 3 or 4] uint8 on the device -> [..., H, W] uint8; gray input is returned as it is."""
        assert image.dtype == torch.uint8 and image.is_cuda and image.is_contiguous()
        if image.shape[-1] not in (3, 4):
            raise ValueError("Input image dimensions are wrong")
        lead = image.shape[:-1]
        out = torch.empty(lead, dtype=torch.uint8, device=image.device)
        rows = int(np.prod(lead[:-1])) if len(lead) > 1 else 1
        rc = self._lib.gtsfm_prep_rgb_to_gray_u8(image.data_ptr(), rows, int(lead[-1]), int(image.shape[-1]), out.data_ptr(),
                                                 torch.cuda.current_stream(image.device).cuda_stream)
        _lib.check(rc, "gtsfm_prep_rgb_to_gray_u8")
        return out

    def prepare(self, value_array: np.ndarray, max_resolution: Optional[int] = None) -> torch.Tensor:
        """Host image (HxW or HxWx3/4 uint8, as ``Image.value_array``) -> gray uint8 [h,w] on the device, downsized like the
        loader when ``max_resolution`` is given."""
        img = torch.from_numpy(np.ascontiguousarray(value_array)).to(self.device)
        if max_resolution is not None:
            img = self.resize_cubic(img, *downsampled_size(int(img.shape[0]), int(img.shape[1]), max_resolution))
        return img if img.dim() == 2 else self.rgb_to_gray(img)
