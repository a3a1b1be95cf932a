"""Per-image / per-pair correspondence generator with the reference's contract
(``gtsfm/frontend/correspondence_generator/det_desc_correspondence_generator.py:19-87``): constructed from a matcher and a
detector-descriptor (plugins or their cachers), ``generate_correspondences(client, images, visibility_graph)`` returns
``(List[Keypoints], Dict[(i1, i2) -> (K, 2) index array])``.

Inside a GTSfM installation the REFERENCE'S generator drives this package's plugins unchanged and this
module is not needed; it exists for installations without GTSfM (``gtsfm_amd/configs/deep_front_end_amd.yaml``) and for the tests.
With a Dask ``client`` it submits what the reference submits -- one task per image, one per pair, the plugin objects scattered
once --; with ``client=None`` the same calls run in the calling process, in visibility-graph order. For throughput use
``BatchedDetDescCorrespondenceGenerator`` (features stay in HBM between the stages)."""

from __future__ import annotations

from typing import Any, Dict, List, Tuple

import numpy as np

from gtsfm_amd.common.keypoints import Keypoints
from gtsfm_amd.frontend.correspondence_generator.correspondence_generator_base import CorrespondenceGeneratorBase


def _describe(det_desc, image):
    return det_desc.detect_and_describe(image)


def _match(matcher, features_i1, features_i2, **shapes):
    return matcher.match(features_i1[0], features_i2[0], features_i1[1], features_i2[1], **shapes)


def _shape_of(image):
    return image.shape


def _first(pair):
    return pair[0]


class DetDescCorrespondenceGenerator(CorrespondenceGeneratorBase):
    """Pair-wise matching of descriptors, one plugin call per image and per pair."""

    def __init__(self, matcher, detector_descriptor) -> None:
        self._detector_descriptor = detector_descriptor
        self._matcher = matcher

    def __repr__(self) -> str:
        return f"""
        DetDescCorrespondenceGenerator:
           {self._detector_descriptor}
           {self._matcher}
        """

    def generate_correspondences(
        self, client: Any, images: List[Any], visibility_graph: List[Tuple[int, int]]
    ) -> Tuple[List[Keypoints], Dict[Tuple[int, int], np.ndarray]]:
        if client is None:  # no scheduler: the calling process is the worker
            images = [im.result() if hasattr(im, "result") else im for im in images]
            features = [_describe(self._detector_descriptor, im) for im in images]
            putative = {
                (i1, i2): _match(self._matcher, features[i1], features[i2], im_shape_i1=_shape_of(images[i1]), im_shape_i2=_shape_of(images[i2]))
                for (i1, i2) in visibility_graph
            }
            return [f[0] for f in features], putative
        det = client.scatter(self._detector_descriptor, broadcast=False)
        features = [client.submit(_describe, det, image) for image in images]
        del det
        matcher = client.scatter(self._matcher, broadcast=False)
        shapes = [client.submit(_shape_of, image) for image in images]
        tasks = {
            (i1, i2): client.submit(_match, matcher, features[i1], features[i2], im_shape_i1=shapes[i1], im_shape_i2=shapes[i2])
            for (i1, i2) in visibility_graph
        }
        putative = client.gather(tasks)
        return client.gather(client.map(_first, features)), putative
