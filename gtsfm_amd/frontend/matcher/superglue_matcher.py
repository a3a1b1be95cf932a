"""SuperGlue matcher plugin on the MI355X HIP path.

Drop-in for ``gtsfm/frontend/matcher/superglue_matcher.py:30-115``: same class name, constructor and ``match``
signatures, same checks (``ValueError`` without responses, ``Exception`` for non-256-d descriptors), same output
((K, 2) ``uint32`` ordered by image-i1 keypoint index). The reference builds its model eagerly in ``__init__``; here
the checkpoint is read eagerly (so a missing file fails at construction, like the reference's ``torch.load``) but
device state is created on first use, keeping the object picklable for Dask (``tests/frontend/matcher/
test_matcher_base.py:102-107``).

The model (``thirdparty/SuperGluePretrainedNetwork/models/superglue.py:228-283``) runs as hand-written HIP kernels.
"""

from __future__ import annotations

from pathlib import Path
from typing import Optional, Tuple, Union

import numpy as np

from gtsfm_amd.common.keypoints import Keypoints
from gtsfm_amd.frontend.matcher.matcher_base import MatcherBase

SUPERGLUE_DESC_DIM = 256
# hyperparameter below set per the author's default demo recommendations (superglue_matcher.py:27)
DEFAULT_NUM_SINKHORN_ITERATIONS = 20
DEFAULT_MATCH_THRESHOLD = 0.2  # superglue.py:201

ROOT_PATH = Path(__file__).resolve().parent.parent.parent.parent
WEIGHTS_DIR = ROOT_PATH / "thirdparty" / "SuperGluePretrainedNetwork" / "models" / "weights"


class SuperGlueMatcher(MatcherBase):
    """Implements the SuperGlue matcher -- a pretrained graph neural network using attention (HIP / gfx950)."""

    def __init__(self, use_cuda: bool = True, use_outdoor_model: bool = True, weights_path: Optional[Union[Path, str]] = None):
        """``weights_path`` (extension over the reference signature) overrides the default checkpoint location
        ``thirdparty/SuperGluePretrainedNetwork/models/weights/superglue_{outdoor,indoor}.pth``."""
        super().__init__()
        self._config = {
            "descriptor_dim": SUPERGLUE_DESC_DIM,
            "weights": "outdoor" if use_outdoor_model else "indoor",
            "sinkhorn_iterations": DEFAULT_NUM_SINKHORN_ITERATIONS,
        }
        self._use_cuda = use_cuda
        self._weights_path = Path(weights_path) if weights_path is not None else WEIGHTS_DIR / f"superglue_{self._config['weights']}.pth"
        if not self._weights_path.exists():
            raise FileNotFoundError(f"SuperGlue weights not found at {self._weights_path}.")
        self._model = None

    def __getstate__(self):
        state = dict(self.__dict__)
        state["_model"] = None
        return state

    def _ensure_model_loaded(self) -> None:
        if self._model is not None:
            return
        from gtsfm_amd.frontend.registry import MODEL_LOAD_LOCK, warn_if_cpu_requested

        with MODEL_LOAD_LOCK:
            if self._model is None:
                import torch

                from gtsfm_amd.runtime.matcher_engine import SuperGlueEngine

                warn_if_cpu_requested(self._use_cuda, "SuperGlueMatcher")
                self._model = SuperGlueEngine(torch.load(str(self._weights_path), map_location="cpu"))

    def match(
        self,
        keypoints_i1: Keypoints,
        keypoints_i2: Keypoints,
        descriptors_i1: np.ndarray,
        descriptors_i2: np.ndarray,
        im_shape_i1: Tuple[int, int, int],
        im_shape_i2: Tuple[int, int, int],
    ) -> np.ndarray:
        """Match keypoints using their 2d positions and descriptor vectors; returns (K, 2) uint32 indices."""
        if keypoints_i1.responses is None or keypoints_i2.responses is None:
            raise ValueError("Responses for keypoints required for SuperGlue")
        if descriptors_i1.shape[1] != SUPERGLUE_DESC_DIM or descriptors_i2.shape[1] != SUPERGLUE_DESC_DIM:
            raise Exception("Superglue pretrained network only works on 256 dimensional descriptors")
        self._ensure_model_loaded()
        H1, W1, _ = im_shape_i1
        H2, W2, _ = im_shape_i2
        pred = self._model.match_pair(
            keypoints_i1.coordinates, keypoints_i1.responses, descriptors_i1,
            keypoints_i2.coordinates, keypoints_i2.responses, descriptors_i2,
            (H1, W1), (H2, W2), sinkhorn_iterations=self._config["sinkhorn_iterations"], match_threshold=DEFAULT_MATCH_THRESHOLD,
        )
        matches = pred["matches0"]
        num_kps_i1, num_kps_i2 = len(keypoints_i1), len(keypoints_i2)
        valid = matches > -1
        return np.hstack(
            [np.arange(num_kps_i1)[valid].reshape(-1, 1), np.arange(num_kps_i2)[matches[valid]].reshape(-1, 1)]
        ).astype(np.uint32)
