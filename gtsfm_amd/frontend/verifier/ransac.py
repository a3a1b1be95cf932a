"""RANSAC verifier plugin on the MI355X HIP path.

Drop-in for ``gtsfm/frontend/verifier/ransac.py:51-112`` + ``opencv_verifier_base.py:21-111``: same class name, constructor
and ``verify`` signature, same failure tuple. ``use_intrinsics_in_verification=True`` (the deep front-end's configuration,
``gtsfm/configs/deep_front_end.yaml:46-49``): verified correspondences = the rows of ``match_indices`` whose squared Sampson
error under the estimated essential matrix is below ``(threshold_px / fx)^2``; ``False``: a fundamental matrix from seven-point
samples on pixel coordinates, residual = the larger squared point-to-epipolar-line distance against ``threshold_px^2``, then
``E = K2^T F K1``. ``inlier_ratio_est_model`` = the verified share. With gtsam importable the pose comes back as ``Rot3`` / ``Unit3``; without it (this
container) as a 3x3 and a 3-vector.

Differences, stated rather than hidden: the estimator is five-point RANSAC with a counter-based sampler
(``gtsfm_verify_essential_f64``: MSAC scoring, one round of inner sampling, Gauss-Newton polish of the pose; OpenCV's
``USAC_ACCURATE`` uses a graph-cut local optimisation and draws from its own generator -- PARITY UNPINNED, ``oracle/verifier_oracle.py``); at most 1024 + 256 hypotheses per pair (OpenCV: 1000 for the
essential matrix, ``RANSAC_MAX_ITERS`` = 10^6 for the fundamental matrix); lens distortion / skew are removed on the host with
the calibration's own ``calibrate`` before upload (essential mode), pure pinhole models are normalised on the device; the
fundamental mode uses K() only, as ``fundamental_to_essential_matrix`` does.
"""

from __future__ import annotations

from typing import Any, Optional, Tuple

import numpy as np

from gtsfm_amd.common.calibration import pinhole_parameters
from gtsfm_amd.common.keypoints import Keypoints
from gtsfm_amd.frontend.verifier.verifier_base import VerifierBase


def _to_pose_types(rotation: np.ndarray, direction: np.ndarray):
    try:  # pragma: no cover - only where gtsam is installed
        from gtsam import Rot3, Unit3  # type: ignore

        return Rot3(rotation), Unit3(direction)
    except ImportError:
        return rotation, direction


class Ransac(VerifierBase):
    def __init__(self, use_intrinsics_in_verification: bool, estimation_threshold_px: float, seed: int = 0) -> None:
        super().__init__(use_intrinsics_in_verification, estimation_threshold_px)
        self._seed = int(seed)
        self._engine = None  # lazy: the object must pickle before first use (Dask scatter)

    def __getstate__(self):
        state = dict(self.__dict__)
        state["_engine"] = None
        return state

    def _ensure_engine(self):
        if self._engine is None:
            from gtsfm_amd.runtime.verifier_engine import VerifierEngine

            self._engine = VerifierEngine()
        return self._engine

    def verify(
        self,
        keypoints_i1: Keypoints,
        keypoints_i2: Keypoints,
        match_indices: np.ndarray,
        camera_intrinsics_i1: Any,
        camera_intrinsics_i2: Any,
    ) -> Tuple[Optional[Any], Optional[Any], np.ndarray, float]:
        """See ``OpencvVerifierBase.verify`` (``opencv_verifier_base.py:47-70``) for the contract."""
        import torch

        match_indices = np.asarray(match_indices)
        if match_indices.ndim != 2 or match_indices.shape[0] < (6 if self._use_intrinsics_in_verification else self._min_matches):  # :71-80
            return self._failure_result
        fx1, fy1, cx1, cy1, pure1 = pinhole_parameters(camera_intrinsics_i1)
        fx2, fy2, cx2, cy2, pure2 = pinhole_parameters(camera_intrinsics_i2)
        c1 = np.asarray(keypoints_i1.coordinates)
        c2 = np.asarray(keypoints_i2.coordinates)
        if match_indices.min() < 0 or match_indices[:, 0].max() >= c1.shape[0] or match_indices[:, 1].max() >= c2.shape[0]:
            # the reference indexes numpy arrays with these (IndexError); the device would read out of bounds silently
            raise IndexError("match_indices refer to keypoints outside the keypoint tables")
        threshold_px = float(self._estimation_threshold_px)
        if (pure1 and pure2) or not self._use_intrinsics_in_verification:  # ((u - cx) / fx, (v - cy) / fy) in double precision on the device
            intr = [fx1, fy1, cx1, cy1, fx2, fy2, cx2, cy2]
        else:  # lens distortion or skew: the calibration's own ``calibrate`` on the host, as the reference does (features.py:41-51)
            c1 = np.vstack([np.asarray(camera_intrinsics_i1.calibrate(np.asarray(x[:2], dtype=np.float64).reshape(2, 1))).reshape(1, 2) for x in c1])
            c2 = np.vstack([np.asarray(camera_intrinsics_i2.calibrate(np.asarray(x[:2], dtype=np.float64).reshape(2, 1))).reshape(1, 2) for x in c2])
            threshold_px = threshold_px / max(fx1, fx2)
            intr = [1.0, 1.0, 0.0, 0.0, 1.0, 1.0, 0.0, 0.0]
        # the device table is float32 (SuperPoint's keypoints are; float64 input is rounded: < 1e-7 relative)
        engine = self._ensure_engine()
        dev = engine.device
        table = torch.from_numpy(np.ascontiguousarray(np.concatenate([c1[:, :2], c2[:, :2]], 0), dtype=np.float32)).to(dev)
        idx = torch.from_numpy(np.ascontiguousarray(match_indices.astype(np.int32))).to(dev)
        out = engine.verify_batch(table, [0], [c1.shape[0]], idx, [0, match_indices.shape[0]], np.asarray([intr]), threshold_px, [self._seed],
                                  use_intrinsics=bool(self._use_intrinsics_in_verification))
        stats = out["stats"][0].cpu().numpy()
        if stats[0] == 0:
            return self._failure_result
        mask = out["mask"].cpu().numpy().astype(bool)
        v_corr_idxs = match_indices[mask]
        rotation, direction = _to_pose_types(out["R"][0].cpu().numpy(), out["t"][0].cpu().numpy())
        return rotation, direction, v_corr_idxs, float(mask.mean())
