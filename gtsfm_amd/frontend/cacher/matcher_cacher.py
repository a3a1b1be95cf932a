"""Disk cache around a matcher plugin, in the reference's format and key scheme (mirror of
``gtsfm/frontend/cacher/matcher_cacher.py:27-169``)."""

from __future__ import annotations

from pathlib import Path
from typing import Optional, Tuple

import numpy as np

from gtsfm_amd.common.keypoints import Keypoints
from gtsfm_amd.frontend.cacher import cache_format
from gtsfm_amd.frontend.matcher.matcher_base import MatcherBase

CACHE_ROOT_PATH = Path(__file__).resolve().parent.parent.parent.parent / "cache"


class MatcherCacher(MatcherBase):
    """Cacher for matcher output on disk, keyed on the input."""

    def __init__(self, matcher_obj: MatcherBase, cache_root: Optional[Path] = None) -> None:
        super().__init__()
        self._matcher = matcher_obj
        self._cache_root = Path(cache_root) if cache_root is not None else CACHE_ROOT_PATH

    def match(
        self,
        keypoints_i1: Keypoints,
        keypoints_i2: Keypoints,
        descriptors_i1: np.ndarray,
        descriptors_i2: np.ndarray,
        im_shape_i1: Tuple[int, int, int],
        im_shape_i2: Tuple[int, int, int],
    ) -> np.ndarray:
        key = cache_format.matcher_cache_key(self._matcher, keypoints_i1, keypoints_i2, descriptors_i1, descriptors_i2, im_shape_i1, im_shape_i2)
        path = self._cache_root / "matcher" / "{}.pbz2".format(key)
        cached = cache_format.read_from_bz2_file(path)
        if cached is not None:
            return cached
        match_indices = self._matcher.match(
            keypoints_i1=keypoints_i1, keypoints_i2=keypoints_i2, descriptors_i1=descriptors_i1, descriptors_i2=descriptors_i2,
            im_shape_i1=im_shape_i1, im_shape_i2=im_shape_i2,
        )
        cache_format.write_to_bz2_file(match_indices, path)
        return match_indices
