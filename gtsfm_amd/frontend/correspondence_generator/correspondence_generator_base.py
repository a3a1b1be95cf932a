"""``CorrespondenceGeneratorBase``: the reference's class when GTSfM is importable, else a stand-in with the same
contract (``gtsfm/frontend/correspondence_generator/correspondence_generator_base.py:16-36``)."""

from __future__ import annotations

from abc import abstractmethod
from typing import Any, Dict, List, Tuple

import numpy as np

from gtsfm_amd.common.keypoints import Keypoints

try:  # pragma: no cover
    from gtsfm.frontend.correspondence_generator.correspondence_generator_base import CorrespondenceGeneratorBase  # type: ignore  # noqa: F401
except Exception:  # noqa: BLE001

    class CorrespondenceGeneratorBase:  # type: ignore[no-redef]
        """Base class for correspondence generators."""

        @abstractmethod
        def generate_correspondences(
            self, client: Any, images: List[Any], visibility_graph: List[Tuple[int, int]]
        ) -> Tuple[List[Keypoints], Dict[Tuple[int, int], np.ndarray]]:
            """Returns the keypoints of every image and, per visibility-graph edge, (K, 2) keypoint index pairs."""
