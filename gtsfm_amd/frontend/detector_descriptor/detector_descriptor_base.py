"""``DetectorDescriptorBase``: the reference's class when GTSfM is importable, else a stand-in with the same contract
(``gtsfm/frontend/detector_descriptor/detector_descriptor_base.py:19-57``)."""

from __future__ import annotations

import abc
from typing import Tuple

import numpy as np

from gtsfm_amd.common.image import Image
from gtsfm_amd.common.keypoints import Keypoints
from gtsfm_amd.frontend.registry import GTSFMProcess, UiMetadata

try:  # pragma: no cover
    from gtsfm.frontend.detector_descriptor.detector_descriptor_base import DetectorDescriptorBase  # type: ignore  # noqa: F401
except Exception:  # noqa: BLE001

    class DetectorDescriptorBase(GTSFMProcess):  # type: ignore[no-redef]
        """Joint detector-descriptor working on one image."""

        @staticmethod
        def get_ui_metadata() -> UiMetadata:
            return UiMetadata(
                display_name="DetectorDescriptor",
                input_products=("Images",),
                output_products=("Keypoints", "Descriptors"),
                parent_plate="DetDescCorrespondenceGenerator",
            )

        def __init__(self, max_keypoints: int = 5000):
            self.max_keypoints = max_keypoints

        @abc.abstractmethod
        def detect_and_describe(self, image: Image) -> Tuple[Keypoints, np.ndarray]:
            """Returns keypoints (N <= max_keypoints) and their (N, D) descriptors."""
