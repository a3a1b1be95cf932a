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
(``gtsfm_amd/common/keypoints.py``); where it is not (this container), the cache files are written and read by a pickler /
unpickler pair that names the stand-in ``gtsfm.common.keypoints.Keypoints`` INSIDE THE FILE ONLY -- neither the class nor
``sys.modules`` is touched, so ordinary pickles of the process (Dask scatter, multiprocessing) keep their real module path.
"""

from __future__ import annotations

import hashlib
import logging
import os
import pickle
from bz2 import BZ2File
from pathlib import Path
from typing import Any, List, Optional, Tuple

import numpy as np

from gtsfm_amd.common.keypoints import Keypoints

logger = logging.getLogger(__name__)

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


REFERENCE_KEYPOINTS_PATH = ("gtsfm.common.keypoints", "Keypoints")


class _ReferenceNamesPickler(pickle._Pickler):  # the pure-Python pickler: the C one has no hook for how a class is named
    """Writes the stand-in ``Keypoints`` class under the reference's module path. The hook is ``save`` itself, not
    ``save_global`` / the ``dispatch`` table: libraries such as dill replace ``pickle._Pickler.dispatch[type]`` process-wide."""

    def save(self, obj, save_persistent_id=True):
        if obj is not Keypoints:
            return super().save(obj, save_persistent_id)
        memoized = self.memo.get(id(obj))
        if memoized is not None:
            self.write(self.get(memoized[0]))
            return None
        self.save(REFERENCE_KEYPOINTS_PATH[0])
        self.save(REFERENCE_KEYPOINTS_PATH[1])
        self.write(pickle.STACK_GLOBAL)
        self.memoize(obj)
        return None


class ForeignClassInCacheEntry(Exception):
    """An entry names a class this installation cannot import (raised by ``find_class`` only)."""


class _ForeignAwareUnpickler(pickle.Unpickler):
    """Tells "this entry names a class that does not exist HERE" (the entry is valid for its writer: leave it) apart from every
    other failure (a damaged or incompatible entry: the reference removes it, gtsfm/utils/io.py:442-447). Only the class lookup
    is wrapped, so an ``AttributeError`` / ``ImportError`` raised while an object is being rebuilt stays a corruption."""

    def find_class(self, module, name):
        try:
            return super().find_class(module, name)
        except (ImportError, AttributeError) as exc:  # ModuleNotFoundError is an ImportError
            raise ForeignClassInCacheEntry(f"{module}.{name}") from exc


class _ReferenceNamesUnpickler(_ForeignAwareUnpickler):
    def find_class(self, module, name):
        if (module, name) == REFERENCE_KEYPOINTS_PATH:
            return Keypoints
        return super().find_class(module, name)


def _stand_in_keypoints() -> bool:
    """True where GTSfM is not importable and ``Keypoints`` is this package's own class."""
    return Keypoints.__module__ != REFERENCE_KEYPOINTS_PATH[0]


def read_from_bz2_file(file_path: Path) -> Optional[Any]:
    """gtsfm/utils/io.py:437-449: None when the file is missing; a corrupted file is removed. One deliberate difference: an
    entry that names a CLASS which cannot be imported here (an entry of a shared reference cache holding types this installation
    lacks; detected in ``find_class`` only) is left alone -- it is valid for its writer -- logged and reported as a miss. Every other
    failure, including an ``AttributeError`` / ``ImportError`` raised while an object is being rebuilt, removes the file like the reference."""
    file_path = Path(file_path)
    if not file_path.exists():
        return None
    try:
        with BZ2File(file_path, "rb") as f:
            return (_ReferenceNamesUnpickler if _stand_in_keypoints() else _ForeignAwareUnpickler)(f).load()
    except ForeignClassInCacheEntry as exc:
        logger.warning("Cache entry %s names a class that cannot be imported here (%s): treated as a miss, file kept.", file_path, exc)
        return None
    except Exception:  # noqa: BLE001 - the reference swallows every failure and drops the file
        logger.exception("Cache file %s was corrupted, removing it...", file_path)
        os.remove(file_path)
        return None


def write_to_bz2_file(data: Any, file_path: Path) -> None:
    """gtsfm/utils/io.py:452-457."""
    file_path = Path(file_path)
    file_path.parent.mkdir(exist_ok=True, parents=True)
    with BZ2File(file_path, "wb") as f:
        if _stand_in_keypoints():
            _ReferenceNamesPickler(f, protocol=pickle.DEFAULT_PROTOCOL).dump(data)
        else:
            pickle.dump(data, f)
