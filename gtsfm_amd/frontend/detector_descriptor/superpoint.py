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
    """Superpoint Detector+Descriptor implementation (HIP / gfx950)."""

    def __init__(
        self, max_keypoints: int = 5000, use_cuda: bool = True, weights_path: Union[Path, str] = MODEL_WEIGHTS_PATH
    ) -> None:
        super().__init__(max_keypoints=max_keypoints)
        self._use_cuda = use_cuda
        self._config = {"weights_path": weights_path}
        self._model = None  # lazy: created on the worker at first use
        if not Path(weights_path).exists():
            raise FileNotFoundError(
                f"SuperPoint weights not found at {weights_path}. "
                f"Please run 'bash scripts/download_model_weights.sh' from the repo root."
            )

    # device state never travels with the pickled object
    def __getstate__(self):
        state = dict(self.__dict__)
        state["_model"] = None
        return state

    def _ensure_model_loaded(self) -> None:
        if self._model is None:
            import torch

            from gtsfm_amd.runtime.superpoint_engine import SuperPointEngine

            if not self._use_cuda:
                raise RuntimeError(
                    "gtsfm_amd's SuperPointDetectorDescriptor runs on the GPU only (use_cuda=False requested); "
                    "use the reference implementation for CPU execution."
                )
            state_dict = torch.load(str(self._config["weights_path"]), map_location="cpu")
            self._model = SuperPointEngine(state_dict)

    def detect_and_describe(self, image: Image) -> Tuple[Keypoints, np.ndarray]:
        """Jointly generate keypoint detections and their associated descriptors from a single image."""
        self._ensure_model_loaded()
        assert self._model is not None
        gray = rgb_to_gray_u8(image.value_array)
        if gray.dtype != np.uint8:  # the reference computes astype(float32) / 255.0 whatever the input dtype
            gray = gray.astype(np.float32) / 255.0
        coordinates, scores, descriptors = self._model.detect(np.ascontiguousarray(gray))
        keypoints = Keypoints(coordinates, scales=None, responses=scores)

        if image.mask is not None:
            keypoints, valid_idxs = keypoints.filter_by_mask(image.mask)
            descriptors = descriptors[valid_idxs]
        keypoints, selection_idxs = keypoints.get_top_k(self.max_keypoints)
        descriptors = descriptors[selection_idxs]
        return keypoints, descriptors
