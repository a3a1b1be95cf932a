"""Disk cache around a detector-descriptor plugin, in the reference's format and key scheme (mirror of
``gtsfm/frontend/cacher/detector_descriptor_cacher.py:28-93``; the reference's own cacher wraps the plugins of this
package unchanged -- this mirror exists for installations without GTSfM and for the interchange tests)."""

from __future__ import annotations

from pathlib import Path
from typing import Optional, Tuple

import numpy as np

from gtsfm_amd.common.image import Image
from gtsfm_amd.common.keypoints import Keypoints
from gtsfm_amd.frontend.cacher import cache_format
from gtsfm_amd.frontend.detector_descriptor.detector_descriptor_base import DetectorDescriptorBase

CACHE_ROOT_PATH = Path(__file__).resolve().parent.parent.parent.parent / "cache"


class DetectorDescriptorCacher(DetectorDescriptorBase):
    """Cacher for detector-descriptor output on disk, keyed on the input."""

    def __init__(self, detector_descriptor_obj: DetectorDescriptorBase, cache_root: Optional[Path] = None) -> None:
        super().__init__(max_keypoints=detector_descriptor_obj.max_keypoints)
        self._detector_descriptor = detector_descriptor_obj
        self._cache_root = Path(cache_root) if cache_root is not None else CACHE_ROOT_PATH

    def _cache_path(self, image: Image) -> Path:
        key = cache_format.detector_descriptor_cache_key(self._detector_descriptor, image)
        return self._cache_root / "detector_descriptor" / "{}.pbz2".format(key)

    def detect_and_describe(self, image: Image) -> Tuple[Keypoints, np.ndarray]:
        path = self._cache_path(image)
        cached = cache_format.read_from_bz2_file(path)
        if cached is not None:
            return cached["keypoints"], cached["descriptors"]
        keypoints, descriptors = self._detector_descriptor.detect_and_describe(image)
        cache_format.write_to_bz2_file({"keypoints": keypoints, "descriptors": descriptors}, path)
        return keypoints, descriptors
