"""SuperPoint detector+descriptor plugin on the MI355X HIP path.

Drop-in for ``gtsfm/frontend/detector_descriptor/superpoint.py:32-93``: same class name (so the front-end cachers,
which key on ``type(obj).__name__`` -- ``gtsfm/frontend/cacher/detector_descriptor_cacher.py:40`` -- share entries
with the reference), same constructor and ``detect_and_describe`` signatures, same error behaviour
(``FileNotFoundError`` at construction when the weights file is missing), lazy model creation so the object pickles
before any device state exists (``tests/frontend/detector/test_detector_base.py:51-56``).

The model itself (``thirdparty/SuperGluePretrainedNetwork/models/superpoint.py:145-202``) runs as hand-written HIP
kernels through ``libgtsfm_amd.so``; the host-side post-processing (mask filter, top-k) calls the same ``Keypoints``
methods as the reference so the selection is identical.
"""

from __future__ import annotations

from pathlib import Path
from typing import Optional, Tuple, Union

import numpy as np

from gtsfm_amd.common.image import Image, rgb_to_gray_u8
from gtsfm_amd.common.keypoints import Keypoints
from gtsfm_amd.frontend.detector_descriptor.detector_descriptor_base import DetectorDescriptorBase

ROOT_PATH = Path(__file__).resolve().parent.parent.parent.parent
MODEL_WEIGHTS_PATH = (
    ROOT_PATH / "thirdparty" / "SuperGluePretrainedNetwork" / "models" / "weights" / "superpoint_v1.pth"
)


class SuperPointDetectorDescriptor(DetectorDescriptorBase):
    """SuperPoint on gfx950 behind the reference's detector-descriptor plugin interface."""

    def __init__(
        self, max_keypoints: int = 5000, use_cuda: bool = True, weights_path: Union[Path, str] = MODEL_WEIGHTS_PATH
    ) -> None:
        super().__init__(max_keypoints=max_keypoints)
        checkpoint = Path(weights_path)
        if not checkpoint.exists():  # same failure point as the reference: construction, not first use
            raise FileNotFoundError(
                f"no SuperPoint checkpoint at {checkpoint}; fetch it with scripts/download_model_weights.sh "
                f"or pass weights_path="
            )
        self._use_cuda = bool(use_cuda)
        self._config = {"weights_path": checkpoint}
        self._model = None  # SuperPointEngine, built in the process that first calls detect_and_describe

    def __getstate__(self):
        # the packed weights live in HBM of one process; a pickled copy (Dask scatter) rebuilds its own
        return {**self.__dict__, "_model": None}

    def _ensure_model_loaded(self) -> None:
        if self._model is not None:
            return
        from gtsfm_amd.frontend.registry import MODEL_LOAD_LOCK, warn_if_cpu_requested

        with MODEL_LOAD_LOCK:
            if self._model is not None:
                return
            warn_if_cpu_requested(self._use_cuda, "SuperPointDetectorDescriptor")
            import torch

            from gtsfm_amd.runtime.superpoint_engine import SuperPointEngine

            self._model = SuperPointEngine(torch.load(str(self._config["weights_path"]), map_location="cpu"))

    def detect_and_describe(self, image: Image) -> Tuple[Keypoints, np.ndarray]:
        """Keypoints (with responses, no scales) and their (K, 256) float32 unit descriptors for one image."""
        self._ensure_model_loaded()
        gray = rgb_to_gray_u8(image.value_array)
        if gray.dtype != np.uint8:  # the reference computes astype(float32) / 255.0 whatever the input dtype
            gray = gray.astype(np.float32) / 255.0
        xy, responses, fetch_descriptors = self._model.detect_lazy(np.ascontiguousarray(gray))
        detections = Keypoints(xy, scales=None, responses=responses)

        # selection on the host with the reference's own Keypoints methods -> identical ordering and ties; the descriptor rows
        # of the survivors are gathered on the device and only those cross PCIe
        keep = np.arange(len(detections))
        if image.mask is not None:
            detections, inside = detections.filter_by_mask(image.mask)
            keep = keep[inside]
        detections, strongest = detections.get_top_k(self.max_keypoints)
        return detections, fetch_descriptors(keep[strongest])
