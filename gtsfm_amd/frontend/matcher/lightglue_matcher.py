"""LightGlue matcher plugin on the MI355X HIP path.

Drop-in for ``gtsfm/frontend/matcher/lightglue_matcher.py:24-112``: same class name, constructor
(``features: str, use_cuda: bool = True``) and ``match`` signatures, lazy model creation, ``ValueError`` without
responses, output = ``matches["matches"]`` as a (K, 2) ``int64`` array in image-i1 keypoint order.

The reference obtains the model from the un-vendored ``thirdparty/LightGlue`` submodule, which also downloads the
``superpoint_lightglue`` checkpoint; offline, ``weights_path`` (extension over the reference signature) points at an
upstream-format ``state_dict`` -- either key layout: the published file's ``self_attn.{i}.*`` / ``cross_attn.{i}.*`` or
the in-memory ``transformers.{i}.*`` names (``normalize_lightglue_state_dict``). LightGlue parity is UNPINNED: the HIP path is checked against
``oracle/lightglue_oracle.py``, a restatement of the published algorithm.
"""

from __future__ import annotations

from pathlib import Path
from typing import Optional, Tuple, Union

import numpy as np

from gtsfm_amd.common.keypoints import Keypoints
from gtsfm_amd.frontend.matcher.matcher_base import MatcherBase

ROOT_PATH = Path(__file__).resolve().parent.parent.parent.parent
# Upstream LightGlue fetches "superpoint_lightglue.pth" (release v0.1_arxiv) into the torch.hub checkpoint cache under
# the name below; a copy next to the (un-vendored) submodule is looked up first.
HUB_FILE_NAME = "superpoint_lightglue_v0-1_arxiv.pth"
DEFAULT_WEIGHTS = {"superpoint": ROOT_PATH / "thirdparty" / "LightGlue" / "weights" / "superpoint_lightglue.pth"}


def _default_weight_candidates(features: str):
    import os

    hub = Path(os.environ.get("TORCH_HOME", Path.home() / ".cache" / "torch")) / "hub" / "checkpoints"
    return [DEFAULT_WEIGHTS[features], hub / HUB_FILE_NAME.replace("superpoint", features)]


class LightGlueMatcher(MatcherBase):
    """Implements the LightGlue matcher -- a pretrained graph neural network using attention (HIP / gfx950)."""

    def __init__(self, features: str, use_cuda: bool = True, weights_path: Optional[Union[Path, str]] = None):
        super().__init__()
        self._use_cuda = use_cuda
        self._features = features
        self._weights_path = weights_path
        self._model = None  # lazy

    def __getstate__(self):
        state = dict(self.__dict__)
        state["_model"] = None
        return state

    def _ensure_model_loaded(self):
        if self._model is not None:
            return
        from gtsfm_amd.frontend.registry import MODEL_LOAD_LOCK, warn_if_cpu_requested

        with MODEL_LOAD_LOCK:
            if self._model is not None:
                return
            import torch

            from gtsfm_amd.runtime.matcher_engine import LightGlueEngine

            if self._features != "superpoint":
                raise ValueError(f"gtsfm_amd's LightGlueMatcher supports features='superpoint' only (got {self._features!r}).")
            warn_if_cpu_requested(self._use_cuda, "LightGlueMatcher")
            candidates = [Path(self._weights_path)] if self._weights_path is not None else _default_weight_candidates(self._features)
            path = next((c for c in candidates if c.exists()), None)
            if path is None:
                raise FileNotFoundError(f"LightGlue weights not found at {' or '.join(str(c) for c in candidates)}.")
            self._model = LightGlueEngine(torch.load(str(path), map_location="cpu"))

    def match(
        self,
        keypoints_i1: Keypoints,
        keypoints_i2: Keypoints,
        descriptors_i1: np.ndarray,
        descriptors_i2: np.ndarray,
        im_shape_i1: Tuple[int, int, int],
        im_shape_i2: Tuple[int, int, int],
    ) -> np.ndarray:
        """Match keypoints using their 2D positions and descriptor vectors; returns (K, 2) int64 indices."""
        self._ensure_model_loaded()
        if keypoints_i1.responses is None or keypoints_i2.responses is None:
            raise ValueError("Responses for keypoints required for LightGlue.")
        H1, W1, _ = im_shape_i1
        H2, W2, _ = im_shape_i2
        out = self._model.match_pair(
            keypoints_i1.coordinates, descriptors_i1, keypoints_i2.coordinates, descriptors_i2, (H1, W1), (H2, W2)
        )
        return out["matches"]
