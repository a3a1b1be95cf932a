"""ORACLE tooling (build container only): cache files interchanged with the REFERENCE'S OWN cacher code, in both directions.

``gtsfm/frontend/cacher/{detector_descriptor,matcher}_cacher.py`` cannot normally be imported here (they pull gtsam, cv2, h5py, open3d ...
through ``gtsfm.utils.io``). Their caching logic needs none of those packages, so this script imports the reference's modules
from /root/reference with the ABSENT THIRD-PARTY PACKAGES replaced by inert stand-ins (``unittest.mock`` modules; nothing of GTSfM itself is
replaced) and lets the reference's code run:

1. the REFERENCE cachers wrap stand-in plugins (same class names as the real ones: the class name is the cache namespace) and write cache
   entries with the reference's own ``Keypoints`` class, key scheme and ``write_to_bz2_file``;
2. a SECOND PROCESS without /root/reference on its path (as on a GPU box: ``gtsfm_amd``'s stand-in ``Keypoints``) reads them through
   ``gtsfm_amd.frontend.cacher``: every lookup must HIT (the wrapped plugin raises if it is called) and return the same arrays; it then writes
   entries of its own for other inputs;
3. the REFERENCE cachers read those: HIT (plugin raises if called), same arrays.

``--write`` stores the reference-written files of step 1 and their inputs under ``tests/golden/reference_cache/`` so that
``tests/test_cache_format.py`` repeats step 2 wherever the tests run.   Usage: python oracle/validate_cache_against_reference.py [--write]"""

from __future__ import annotations

import argparse
import importlib.abc
import importlib.machinery
import os
import shutil
import subprocess
import sys
import tempfile
from pathlib import Path
from unittest import mock

import numpy as np

REPO = Path(__file__).resolve().parent.parent
REFERENCE = Path(os.environ.get("GTSFM_REFERENCE", "/root/reference"))
GOLDEN = REPO / "tests" / "golden" / "reference_cache"


class _AbsentPackages(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    """Inert modules for third-party packages that are not installed here and that the caching code never touches."""

    ROOTS = {"gtsam", "cv2", "h5py", "open3d", "simplejson", "dask", "distributed", "pycolmap", "trimesh", "hydra", "omegaconf", "pydot", "matplotlib",
             "plotly", "networkx", "seaborn", "kornia", "pydegensac", "colour", "torchvision", "graphviz", "rawpy", "imageio", "shapely", "pyvista"}

    def find_spec(self, name, path, target=None):
        if name.split(".")[0] in self.ROOTS:
            try:
                if importlib.machinery.PathFinder.find_spec(name.split(".")[0]) is not None:
                    return None  # really installed: use it
            except (ImportError, ValueError):
                pass
            return importlib.machinery.ModuleSpec(name, self, is_package=True)
        return None

    def create_module(self, spec):
        m = mock.MagicMock(name=spec.name)
        m.__name__, m.__path__, m.__spec__, m.__loader__ = spec.name, [], spec, self
        return m

    def exec_module(self, module):
        pass


def sample_inputs(seed: int):
    rng = np.random.default_rng(seed)
    img = rng.integers(0, 256, (48, 64, 3), dtype=np.uint8)
    n1, n2 = 37, 29
    kp = lambda n: (rng.random((n, 2)).astype(np.float32) * 40, rng.random(n).astype(np.float32))  # noqa: E731
    (c1, r1), (c2, r2) = kp(n1), kp(n2)
    d1, d2 = rng.random((n1, 256)).astype(np.float32), rng.random((n2, 256)).astype(np.float32)
    matches = np.stack([np.arange(11), np.arange(11)[::-1]], 1).astype(np.uint32)
    return {"image": img, "file_name": f"frame_{seed}.jpg", "c1": c1, "r1": r1, "c2": c2, "r2": r2, "d1": d1, "d2": d2, "matches": matches}


OURS_SCRIPT = r'''
import sys, numpy as np
sys.path.insert(0, {repo!r})
assert not any("reference" in p for p in sys.path)
from gtsfm_amd.common.image import Image
from gtsfm_amd.common.keypoints import Keypoints
from gtsfm_amd.frontend.cacher.detector_descriptor_cacher import DetectorDescriptorCacher
from gtsfm_amd.frontend.cacher.matcher_cacher import MatcherCacher
from gtsfm_amd.frontend.detector_descriptor.detector_descriptor_base import DetectorDescriptorBase
from gtsfm_amd.frontend.matcher.matcher_base import MatcherBase
assert Keypoints.__module__ == "gtsfm_amd.common.keypoints"   # the stand-in class, as on a box without GTSfM
sys.path.insert(0, {oracle_dir!r})
from validate_cache_against_reference import sample_inputs
root = {root!r}
class SuperPointDetectorDescriptor(DetectorDescriptorBase):
    def __init__(self, data=None): super().__init__(max_keypoints=5000); self.data = data
    def detect_and_describe(self, image):
        if self.data is None: raise AssertionError("cache miss on an entry the reference wrote")
        return Keypoints(self.data["c1"], responses=self.data["r1"]), self.data["d1"]
class SuperGlueMatcher(MatcherBase):
    def __init__(self, data=None): super().__init__(); self.data = data
    def match(self, **kw):
        if self.data is None: raise AssertionError("cache miss on an entry the reference wrote")
        return self.data["matches"]
# (2) read what the reference wrote
a = sample_inputs(1)
kps, desc = DetectorDescriptorCacher(SuperPointDetectorDescriptor(), cache_root=root).detect_and_describe(Image(value_array=a["image"], file_name=a["file_name"]))
assert type(kps) is Keypoints and np.array_equal(kps.coordinates, a["c1"]) and np.array_equal(kps.responses, a["r1"]) and kps.scales is None and np.array_equal(desc, a["d1"])
m = MatcherCacher(SuperGlueMatcher(), cache_root=root).match(Keypoints(a["c1"], responses=a["r1"]), Keypoints(a["c2"], responses=a["r2"]), a["d1"], a["d2"], (48, 64, 3), (48, 64, 3))
assert m.dtype == np.uint32 and np.array_equal(m, a["matches"])
# write entries of our own for the reference to read
b = sample_inputs(2)
DetectorDescriptorCacher(SuperPointDetectorDescriptor(b), cache_root=root).detect_and_describe(Image(value_array=b["image"], file_name=b["file_name"]))
MatcherCacher(SuperGlueMatcher(b), cache_root=root).match(Keypoints(b["c1"], responses=b["r1"]), Keypoints(b["c2"], responses=b["r2"]), b["d1"], b["d2"], (48, 64, 3), (48, 64, 3))
print("OURS_OK")
'''


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--write", action="store_true", help="store the reference-written entries and their inputs under tests/golden/reference_cache/")
    args = ap.parse_args()
    if not (REFERENCE / "gtsfm" / "frontend" / "cacher").exists():
        raise SystemExit(f"reference cachers not found under {REFERENCE}")
    sys.meta_path.insert(0, _AbsentPackages())
    sys.path.insert(0, str(REFERENCE))
    import gtsfm.frontend.cacher.detector_descriptor_cacher as ref_ddc
    import gtsfm.frontend.cacher.matcher_cacher as ref_mc
    from gtsfm.common.image import Image as RefImage
    from gtsfm.common.keypoints import Keypoints as RefKeypoints
    from gtsfm.frontend.detector_descriptor.detector_descriptor_base import DetectorDescriptorBase as RefDDBase
    from gtsfm.frontend.matcher.matcher_base import MatcherBase as RefMatcherBase

    class SuperPointDetectorDescriptor(RefDDBase):  # the class NAME is the reference's cache namespace (detector_descriptor_cacher.py:40)
        def __init__(self, data=None):
            super().__init__(max_keypoints=5000)
            self.data = data

        def detect_and_describe(self, image):
            if self.data is None:
                raise AssertionError("cache miss on an entry gtsfm_amd wrote")
            return RefKeypoints(self.data["c1"], responses=self.data["r1"]), self.data["d1"]

    class SuperGlueMatcher(RefMatcherBase):
        def __init__(self, data=None):
            super().__init__()
            self.data = data

        def match(self, **kw):
            if self.data is None:
                raise AssertionError("cache miss on an entry gtsfm_amd wrote")
            return self.data["matches"]

    with tempfile.TemporaryDirectory() as tmp:
        root = Path(tmp) / "cache"
        ref_ddc.CACHE_ROOT_PATH = ref_mc.CACHE_ROOT_PATH = root  # the reference derives it from its own location (read-only here)
        # (1) the reference's code writes
        a = sample_inputs(1)
        ref_ddc.DetectorDescriptorCacher(SuperPointDetectorDescriptor(a)).detect_and_describe(RefImage(value_array=a["image"], file_name=a["file_name"]))
        ref_mc.MatcherCacher(SuperGlueMatcher(a)).match(RefKeypoints(a["c1"], responses=a["r1"]), RefKeypoints(a["c2"], responses=a["r2"]), a["d1"], a["d2"],
                                                        (48, 64, 3), (48, 64, 3))
        written = sorted(p.relative_to(root) for p in root.rglob("*.pbz2"))
        assert len(written) == 2, written
        print("reference wrote:", [str(p) for p in written])
        if args.write:
            if GOLDEN.exists():
                shutil.rmtree(GOLDEN)
            for rel in written:
                (GOLDEN / rel.parent).mkdir(parents=True, exist_ok=True)
                shutil.copy(root / rel, GOLDEN / rel)
        # (2) our cachers, in a process that cannot see the reference, read them and write their own
        env = {k: v for k, v in os.environ.items() if k != "PYTHONPATH"}
        out = subprocess.run([sys.executable, "-c", OURS_SCRIPT.format(repo=str(REPO), oracle_dir=str(REPO / "oracle"), root=str(root))], capture_output=True, text=True, env=env)
        assert out.returncode == 0 and "OURS_OK" in out.stdout, out.stderr[-3000:]
        print("gtsfm_amd (stand-in Keypoints, no reference on its path) read both entries as hits and wrote two of its own")
        # (3) the reference's code reads ours
        b = sample_inputs(2)
        kps, desc = ref_ddc.DetectorDescriptorCacher(SuperPointDetectorDescriptor()).detect_and_describe(RefImage(value_array=b["image"], file_name=b["file_name"]))
        assert type(kps) is RefKeypoints and np.array_equal(kps.coordinates, b["c1"]) and np.array_equal(kps.responses, b["r1"]) and np.array_equal(desc, b["d1"])
        m = ref_mc.MatcherCacher(SuperGlueMatcher()).match(RefKeypoints(b["c1"], responses=b["r1"]), RefKeypoints(b["c2"], responses=b["r2"]), b["d1"], b["d2"],
                                                          (48, 64, 3), (48, 64, 3))
        assert np.array_equal(m, b["matches"])
        print("the reference's cachers read gtsfm_amd's entries as hits, same arrays")
    print("OK")


if __name__ == "__main__":
    main()
