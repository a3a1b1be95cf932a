"""``MatcherBase``: the reference's class when GTSfM is importable, else a stand-in with the same contract
(``gtsfm/frontend/matcher/matcher_base.py:14-65``)."""

from __future__ import annotations

import abc
from typing import Tuple

import numpy as np

from gtsfm_amd.common.keypoints import Keypoints
from gtsfm_amd.frontend.registry import GTSFMProcess, UiMetadata

try:  # pragma: no cover
    from gtsfm.frontend.matcher.matcher_base import MatcherBase  # type: ignore  # noqa: F401
except Exception:  # noqa: BLE001

    class MatcherBase(GTSFMProcess):  # type: ignore[no-redef]
        """Matches the descriptors of two images; returns (K, 2) index pairs."""

        @staticmethod
        def get_ui_metadata() -> UiMetadata:
            return UiMetadata(
                display_name="Matcher",
                input_products=("Keypoints", "Descriptors", "Image Shapes"),
                output_products=("Putative Correspondences",),
                parent_plate="DetDescCorrespondenceGenerator",
            )

        @abc.abstractmethod
        def match(
            self,
            keypoints_i1: Keypoints,
            keypoints_i2: Keypoints,
            descriptors_i1: np.ndarray,
            descriptors_i2: np.ndarray,
            im_shape_i1: Tuple[int, int, int],
            im_shape_i2: Tuple[int, int, int],
        ) -> np.ndarray:
            """Returns match indices (K, 2): column 0 indexes image i1's keypoints, column 1 image i2's."""
