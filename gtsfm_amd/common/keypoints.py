"""``Keypoints`` boundary type.

When GTSfM is importable, ``gtsfm.common.keypoints.Keypoints`` is re-exported unchanged, so the plugins hand the rest
of the pipeline the reference's own class. GTSfM cannot be imported in the build container (cv2 / gtsam are missing,
SURVEY.md F10); the stand-in below restates the subset of the interface the deep front-end touches
(``gtsfm/common/keypoints.py:16-127,218-231``): construction, ``len``, equality, ``get_top_k``,
``filter_by_mask``, ``extract_indices``.
"""

from __future__ import annotations

from typing import Optional, Tuple

import numpy as np

try:  # pragma: no cover - exercised only where GTSfM is installed
    from gtsfm.common.keypoints import Keypoints  # type: ignore  # noqa: F401
except Exception:  # noqa: BLE001 - any import failure (cv2, gtsam, ...) selects the stand-in

    def _opt_equal(a: Optional[np.ndarray], b: Optional[np.ndarray]) -> bool:
        if a is None or b is None:
            return a is None and b is None
        return bool(np.array_equal(a, b))

    class Keypoints:  # type: ignore[no-redef]
        """(x, y) coordinates with optional scales / responses; deliberately not a NamedTuple (Dask sizes those by
        sampling elements)."""

        def __init__(
            self, coordinates: np.ndarray, scales: Optional[np.ndarray] = None, responses: Optional[np.ndarray] = None
        ) -> None:
            self.coordinates = coordinates
            self.scales = scales
            self.responses = responses

        def __len__(self) -> int:
            return self.coordinates.shape[0]

        def __sizeof__(self) -> int:
            return (
                object.__sizeof__(self)
                + self.coordinates.__sizeof__()
                + self.scales.__sizeof__()
                + self.responses.__sizeof__()
            )

        def __eq__(self, other: object) -> bool:
            if not isinstance(other, Keypoints):
                return False
            return (
                bool(np.array_equal(self.coordinates, other.coordinates))
                and _opt_equal(self.scales, other.scales)
                and _opt_equal(self.responses, other.responses)
            )

        def __ne__(self, other: object) -> bool:
            return not self == other

        def extract_indices(self, indices: np.ndarray) -> "Keypoints":
            if indices.size == 0:
                return Keypoints(coordinates=np.zeros(shape=(0, 2)))
            return Keypoints(
                self.coordinates[indices],
                None if self.scales is None else self.scales[indices],
                None if self.responses is None else self.responses[indices],
            )

        def get_top_k(self, k: int) -> Tuple["Keypoints", np.ndarray]:
            """Top-k by response via ``np.argpartition`` (unordered, like the reference)."""
            n = len(self)
            if k >= n:
                return Keypoints(
                    np.array(self.coordinates, copy=True),
                    None if self.scales is None else np.array(self.scales, copy=True),
                    None if self.responses is None else np.array(self.responses, copy=True),
                ), np.arange(n)
            if self.responses is None:
                sel = np.arange(k, dtype=np.uint32)
            else:
                sel = np.argpartition(-self.responses, k)[:k]
            return self.extract_indices(sel), sel

        def filter_by_mask(self, mask: np.ndarray) -> Tuple["Keypoints", np.ndarray]:
            rc = np.round(self.coordinates).astype(int)
            valid = np.flatnonzero(mask[rc[:, 1], rc[:, 0]] == 1)
            return self.extract_indices(valid), valid

        def get_x_coordinates(self) -> np.ndarray:
            return self.coordinates[:, 0]

        def get_y_coordinates(self) -> np.ndarray:
            return self.coordinates[:, 1]
