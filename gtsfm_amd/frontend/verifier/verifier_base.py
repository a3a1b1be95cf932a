"""``VerifierBase``: the reference's class when GTSfM is importable, else a stand-in with the same contract
(``gtsfm/frontend/verifier/verifier_base.py:14-90``)."""

from __future__ import annotations

import abc
from typing import Any, Optional, Tuple

import numpy as np

from gtsfm_amd.common.keypoints import Keypoints
from gtsfm_amd.frontend.registry import GTSFMProcess, UiMetadata

NUM_MATCHES_REQ_E_MATRIX = 5
NUM_MATCHES_REQ_F_MATRIX = 8

try:  # pragma: no cover
    from gtsfm.frontend.verifier.verifier_base import VerifierBase  # type: ignore  # noqa: F401
except Exception:  # noqa: BLE001

    class VerifierBase(GTSFMProcess):  # type: ignore[no-redef]
        """Takes the coordinates of the matches; returns the relative pose and the geometrically verified matches."""

        @staticmethod
        def get_ui_metadata() -> UiMetadata:
            return UiMetadata(
                display_name="Verifier",
                input_products=("Keypoints", "Putative Correspondences", "Camera Intrinsics"),
                output_products=("Relative Rotation", "Relative Translation", "Verified Correspondences"),
                parent_plate="Two-View Estimator",
            )

        def __repr__(self) -> str:
            return f"{type(self).__name__}__use_intrinsics{self._use_intrinsics_in_verification}_{self._estimation_threshold_px}px"

        def __init__(self, use_intrinsics_in_verification: bool, estimation_threshold_px: float) -> None:
            self._use_intrinsics_in_verification = use_intrinsics_in_verification
            self._estimation_threshold_px = estimation_threshold_px
            self._min_matches = NUM_MATCHES_REQ_E_MATRIX if use_intrinsics_in_verification else NUM_MATCHES_REQ_F_MATRIX
            self._failure_result = (None, None, np.array([], dtype=np.uint64), 0.0)

        @abc.abstractmethod
        def verify(
            self, keypoints_i1: Keypoints, keypoints_i2: Keypoints, match_indices: np.ndarray, camera_intrinsics_i1: Any,
            camera_intrinsics_i2: Any,
        ) -> Tuple[Optional[Any], Optional[Any], np.ndarray, float]:
            """Returns (i2Ri1, i2Ui1, verified (N,2) subset of match_indices, inlier ratio w.r.t. the estimated model)."""
