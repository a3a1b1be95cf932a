"""GPU-resident correspondence generator ("next" row 1).

Same contract as ``DetDescCorrespondenceGenerator`` (``gtsfm/frontend/correspondence_generator/
det_desc_correspondence_generator.py:19-87``): ``generate_correspondences(client, images, visibility_graph)`` returns
``(List[Keypoints], Dict[(i1, i2) -> (K, 2) index array])`` and is constructed from the same two plugin objects. The
reference submits one Dask task per image and one per pair, each with its own host<->device round trip and pickle;
here the images of a cluster are detected in batches, the features stay in HBM, and all edges of the visibility graph
are matched from that resident table in ragged multi-pair launches (``gtsfm_amd.runtime.pipeline``).

Differences that are visible to a caller, by design:
* keypoints are returned in detection (row-major) order; the reference's order is ``np.argpartition``'s
  (implementation-defined). Index pairs refer to the returned lists, so downstream code is unaffected.
* the work runs in the calling process, which must own a GPU (the reference fans out over Dask workers). ``client``
  is only used to resolve image futures.
"""

from __future__ import annotations

from typing import Any, Dict, List, Sequence, Tuple

import numpy as np

from gtsfm_amd.common.keypoints import Keypoints
from gtsfm_amd.frontend.correspondence_generator.correspondence_generator_base import CorrespondenceGeneratorBase
from gtsfm_amd.frontend.detector_descriptor.superpoint import SuperPointDetectorDescriptor
from gtsfm_amd.frontend.matcher.lightglue_matcher import LightGlueMatcher
from gtsfm_amd.frontend.matcher.matcher_base import MatcherBase
from gtsfm_amd.frontend.matcher.superglue_matcher import DEFAULT_MATCH_THRESHOLD, SuperGlueMatcher


class BatchedDetDescCorrespondenceGenerator(CorrespondenceGeneratorBase):
    """Batched, GPU-resident SuperPoint + {SuperGlue, LightGlue} correspondence generation."""

    def __init__(
        self, matcher: MatcherBase, detector_descriptor: SuperPointDetectorDescriptor, image_batch: int = 16, pair_batch: int = 32
    ) -> None:
        if not isinstance(detector_descriptor, SuperPointDetectorDescriptor):
            raise TypeError("BatchedDetDescCorrespondenceGenerator needs gtsfm_amd's SuperPointDetectorDescriptor")
        if not isinstance(matcher, (SuperGlueMatcher, LightGlueMatcher)):
            raise TypeError("BatchedDetDescCorrespondenceGenerator needs gtsfm_amd's SuperGlueMatcher or LightGlueMatcher")
        self._detector_descriptor = detector_descriptor
        self._matcher = matcher
        self._image_batch = image_batch
        self._pair_batch = pair_batch

    def __repr__(self) -> str:
        return f"""
        BatchedDetDescCorrespondenceGenerator:
           {self._detector_descriptor}
           {self._matcher}
        """

    @staticmethod
    def _resolve(client: Any, images: Sequence[Any]) -> List[Any]:
        if client is not None and len(images) > 0 and hasattr(images[0], "key"):  # Dask futures
            return list(client.gather(list(images)))
        return [im.result() if hasattr(im, "result") else im for im in images]

    def generate_correspondences(
        self, client: Any, images: List[Any], visibility_graph: List[Tuple[int, int]]
    ) -> Tuple[List[Keypoints], Dict[Tuple[int, int], np.ndarray]]:
        keypoints_list, putative, _ = self._detect_and_match(client, images, visibility_graph)
        return keypoints_list, putative

    def generate_correspondences_and_verify(
        self, client: Any, images: List[Any], visibility_graph: List[Tuple[int, int]], camera_intrinsics: List[Any], verifier: Any
    ) -> Tuple[List[Keypoints], Dict[Tuple[int, int], np.ndarray], Dict[Tuple[int, int], Tuple[Any, Any, np.ndarray, float]]]:
        """``generate_correspondences`` followed by the verifier stage of ``TwoViewEstimator.run_2view``
        (``gtsfm/two_view_estimator.py:391-397``) for every edge, with keypoints and matches staying in HBM between the stages.
        ``verifier``: a ``gtsfm_amd.frontend.verifier.ransac.Ransac`` (threshold and estimation mode are read from it);
        ``camera_intrinsics``: one calibration per image. Returns the keypoints, the putative correspondences and, per edge, the
        verifier's return tuple ``(i2Ri1, i2Ui1, v_corr_idxs, inlier_ratio_est_model)``; edge (i1, i2) draws its samples from the
        seed ``i1 << 32 | i2``. Calibrations with lens distortion or skew fall back to the per-pair plugin call for their edges."""
        from gtsfm_amd.common.calibration import pinhole_parameters
        from gtsfm_amd.frontend.verifier.ransac import Ransac, _to_pose_types

        if not isinstance(verifier, Ransac):
            raise TypeError("generate_correspondences_and_verify needs gtsfm_amd's Ransac verifier")
        keypoints_list, putative, state = self._detect_and_match(client, images, visibility_graph)
        params = [pinhole_parameters(c) for c in camera_intrinsics]
        use_intrinsics = bool(verifier._use_intrinsics_in_verification)
        verified: Dict[Tuple[int, int], Tuple[Any, Any, np.ndarray, float]] = {}
        on_device = [r for r in state["results"]]
        if use_intrinsics and not all(p[4] for p in params):
            # distortion / skew / a non-pinhole model: those edges use the calibration's own calibrate(), per pair on the host,
            # and stay out of the device batch (a chunk holding such an edge goes to the host as a whole: its match lists are
            # interleaved on the device)
            on_device = [r for r in on_device if all(params[i][4] and params[j][4] for i, j in r["pairs"])]
        if on_device:
            intr = np.array([p[:4] for p in params], dtype=np.float64)
            ver = state["pipe"].verify(state["feats"], on_device, intr, float(verifier._estimation_threshold_px), use_intrinsics=use_intrinsics)
            for pair, res in state["pipe"].verified_to_numpy(ver).items():
                dtype = putative[pair].dtype
                if res["R"] is None:
                    verified[pair] = verifier._failure_result
                else:
                    rot, direction = _to_pose_types(res["R"], res["t"])
                    verified[pair] = (rot, direction, res["v_corr_idxs"].astype(dtype), res["inlier_ratio"])
        for pair in putative:
            if pair not in verified:  # empty keypoint sets, or a calibration the device path does not model
                i1, i2 = pair
                per_pair = Ransac(use_intrinsics, verifier._estimation_threshold_px, seed=(i1 << 32) | i2)
                per_pair._engine = verifier._ensure_engine()  # one lib handle / workspace for every fallback edge
                verified[pair] = per_pair.verify(keypoints_list[i1], keypoints_list[i2], putative[pair], camera_intrinsics[i1], camera_intrinsics[i2])
        return keypoints_list, putative, {p: verified[p] for p in putative}

    def _detect_and_match(self, client: Any, images: List[Any], visibility_graph: List[Tuple[int, int]]):
        from gtsfm_amd.runtime.pipeline import FrontEndPipeline

        imgs = self._resolve(client, images)
        det, matcher = self._detector_descriptor, self._matcher
        det._ensure_model_loaded()
        matcher._ensure_model_loaded()
        pipe = FrontEndPipeline(det._model, matcher._model, max_keypoints=det.max_keypoints, pair_chunk=self._pair_batch)
        shapes = [(im.height, im.width) for im in imgs]
        feats = pipe.detect_image_objects(imgs, self._image_batch)  # batched by shape, gray conversion / masks / top-k on the device
        counts = feats["count"].cpu().numpy().astype(np.int64)
        keypoints_list = keypoints_from_table(feats, counts)

        pairs = [(int(i1), int(i2)) for (i1, i2) in visibility_graph]
        todo, empty = split_empty_pairs(pairs, counts)  # superglue.py:233-240 early-out
        dtype, kwargs = match_output_convention(matcher)
        results = pipe.match(feats, todo, shapes, counts=counts, **kwargs) if todo else []
        result = pipe.matches_to_numpy(results, dtype=dtype)
        for p in empty:
            result[p] = np.zeros((0, 2), dtype=dtype)
        return keypoints_list, {p: result[p] for p in pairs}, {"pipe": pipe, "feats": feats, "results": results}


def keypoints_from_table(feats, counts) -> List[Keypoints]:
    """The generator's first return value from a device feature table: per image the first ``count`` rows (detection order)."""
    xy_h, sc_h = feats["xy"].cpu().numpy(), feats["scores"].cpu().numpy()
    return [Keypoints(coordinates=xy_h[i, : int(c)].copy(), scales=None, responses=sc_h[i, : int(c)].copy()) for i, c in enumerate(counts)]


def split_empty_pairs(pairs, counts):
    """(pairs to match, pairs with an empty keypoint set): the latter never reach the matcher (superglue.py:233-240 early-out)."""
    empty = [p for p in pairs if counts[p[0]] == 0 or counts[p[1]] == 0]
    empty_set = set(empty)
    return [p for p in pairs if p not in empty_set], empty


def match_output_convention(matcher):
    """(dtype of the (K, 2) arrays, matcher keyword arguments) per plugin: SuperGlue returns uint32 (superglue_matcher.py:104-113) and runs
    its configured Sinkhorn iterations, LightGlue returns int64 (lightglue_matcher.py:107-110)."""
    if isinstance(matcher, SuperGlueMatcher):
        return np.uint32, {"sinkhorn_iterations": matcher._config["sinkhorn_iterations"], "match_threshold": DEFAULT_MATCH_THRESHOLD}
    return np.int64, {}
