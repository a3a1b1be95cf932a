"""The reference's front-end cache format and key scheme, so that cache entries are interchangeable with a GTSfM
installation (its ``cache/`` directory, CI's ``gtsfm-cache``):

* files: ``pickle`` inside ``bz2`` (``gtsfm/utils/io.py:437-457``), ``<root>/detector_descriptor/<key>.pbz2`` holding
  ``{"keypoints": Keypoints, "descriptors": ndarray}`` (``gtsfm/frontend/cacher/detector_descriptor_cacher.py:48-69``) and
  ``<root>/matcher/<key>.pbz2`` holding the (K, 2) match array (``gtsfm/frontend/cacher/matcher_cacher.py:46-126``);
* keys: ``<type(obj).__name__>_<input hash>`` with SHA-1 hashes of the image name / size / bytes
  (``gtsfm/utils/cache.py:14-23``) or of the first 10 keypoints / responses / scales / descriptors of both images plus the
  two image shapes (``matcher_cacher.py:24,46-80``). The plugin classes here carry the reference's class names on purpose,
  so both installations address the same entries.

``Keypoints`` objects must unpickle on the other side: when GTSfM is importable its own class is used throughout
(``gtsfm_amd/common/keypoints.py``); where it is not (this container), ``enable_reference_pickle_names`` registers the
stand-in under the reference's module path so that files written here carry ``gtsfm.common.keypoints.Keypoints``.
"""

from __future__ import annotations

import hashlib
import os
import pickle
import sys
import types
from bz2 import BZ2File
from pathlib import Path
from typing import Any, List, Optional, Tuple

import numpy as np

from gtsfm_amd.common.keypoints import Keypoints

NUM_KEYPOINTS_TO_SAMPLE_FOR_HASH = 10  # matcher_cacher.py:24


def generate_hash_for_numpy_array(array: np.ndarray) -> str:
    """gtsfm/utils/cache.py:21-23."""
    return hashlib.sha1(array.tobytes()).hexdigest()


def generate_hash_for_image(image) -> str:
    """gtsfm/utils/cache.py:14-18: image name and size, then the pixel bytes."""
    return hashlib.sha1("{}_{}_{}".format(image.file_name, image.width, image.height).encode()).hexdigest() + generate_hash_for_numpy_array(
        image.value_array
    )


def detector_descriptor_cache_key(detector_descriptor_obj, image) -> str:
    """detector_descriptor_cacher.py:40,52-55."""
    return "{}_{}".format(type(detector_descriptor_obj).__name__, generate_hash_for_image(image))


def matcher_cache_key(matcher_obj, keypoints_i1, keypoints_i2, descriptors_i1: np.ndarray, descriptors_i2: np.ndarray,
                      im_shape_i1: Tuple[int, int, int], im_shape_i2: Tuple[int, int, int]) -> str:
    """matcher_cacher.py:46-80."""
    arrays: List[np.ndarray] = []
    for keypoints_i, descriptors_i in zip([keypoints_i1, keypoints_i2], [descriptors_i1, descriptors_i2]):
        arrays.append(keypoints_i.coordinates[:NUM_KEYPOINTS_TO_SAMPLE_FOR_HASH].flatten())
        if keypoints_i.responses is not None:
            arrays.append(keypoints_i.responses[:NUM_KEYPOINTS_TO_SAMPLE_FOR_HASH].flatten())
        if keypoints_i.scales is not None:
            arrays.append(keypoints_i.scales[:NUM_KEYPOINTS_TO_SAMPLE_FOR_HASH].flatten())
        arrays.append(descriptors_i[:NUM_KEYPOINTS_TO_SAMPLE_FOR_HASH].flatten())
    h1, w1, c1 = im_shape_i1
    h2, w2, c2 = im_shape_i2
    arrays.append(np.array([h1, w1, c1, h2, w2, c2]))
    return "{}_{}".format(type(matcher_obj).__name__, generate_hash_for_numpy_array(np.concatenate(arrays)))


def read_from_bz2_file(file_path: Path) -> Optional[Any]:
    """gtsfm/utils/io.py:437-449: None when the file is missing; a corrupted file is removed."""
    file_path = Path(file_path)
    if not file_path.exists():
        return None
    try:
        enable_reference_pickle_names()
        return pickle.load(BZ2File(file_path, "rb"))
    except Exception:  # noqa: BLE001 - the reference swallows every failure and drops the file
        os.remove(file_path)
        return None


def write_to_bz2_file(data: Any, file_path: Path) -> None:
    """gtsfm/utils/io.py:452-457."""
    file_path = Path(file_path)
    file_path.parent.mkdir(exist_ok=True, parents=True)
    enable_reference_pickle_names()
    pickle.dump(data, BZ2File(file_path, "wb"))


def enable_reference_pickle_names() -> bool:
    """Make ``Keypoints`` pickle as ``gtsfm.common.keypoints.Keypoints`` where GTSfM itself cannot be imported: registers
    placeholder modules ``gtsfm`` / ``gtsfm.common`` / ``gtsfm.common.keypoints`` holding the stand-in class. A no-op (returns
    False) when the real package is present -- its class is what ``gtsfm_amd.common.keypoints`` re-exports then."""
    if Keypoints.__module__ == "gtsfm.common.keypoints":
        return "gtsfm_amd_placeholder" in getattr(sys.modules.get("gtsfm.common.keypoints"), "__dict__", {})
    for name in ("gtsfm", "gtsfm.common", "gtsfm.common.keypoints"):
        if name not in sys.modules:
            mod = types.ModuleType(name)
            mod.__dict__["gtsfm_amd_placeholder"] = True
            mod.__path__ = []  # type: ignore[attr-defined]
            sys.modules[name] = mod
    sys.modules["gtsfm"].common = sys.modules["gtsfm.common"]  # type: ignore[attr-defined]
    sys.modules["gtsfm.common"].keypoints = sys.modules["gtsfm.common.keypoints"]  # type: ignore[attr-defined]
    sys.modules["gtsfm.common.keypoints"].Keypoints = Keypoints  # type: ignore[attr-defined]
    Keypoints.__module__ = "gtsfm.common.keypoints"
    Keypoints.__qualname__ = "Keypoints"
    return True
