"""CPU: API-contract tests of the plugin classes, modelled on the reference's
``tests/frontend/detector/test_detector_base.py:51-56`` (picklable), ``tests/frontend/detector_descriptor/
test_superpoint.py`` and the boundary table of SURVEY.md section 8(b)."""

import pickle

import numpy as np
import pytest
import torch

from gtsfm_amd.common.image import Image, rgb_to_gray_u8
from gtsfm_amd.common.keypoints import Keypoints
from gtsfm_amd.frontend.detector_descriptor.detector_descriptor_base import DetectorDescriptorBase
from gtsfm_amd.frontend.detector_descriptor.superpoint import SuperPointDetectorDescriptor
from gtsfm_amd.frontend.registry import GTSFMProcess
from gtsfm_amd.utils import synthetic


@pytest.fixture()
def sp_weights(tmp_path):
    path = tmp_path / "superpoint_v1.pth"
    torch.save(synthetic.synthetic_superpoint_state_dict(), str(path))
    return path


def test_missing_weights_raise_at_construction(tmp_path):
    with pytest.raises(FileNotFoundError):
        SuperPointDetectorDescriptor(weights_path=tmp_path / "nope.pth")


def test_superpoint_plugin_is_lazy_picklable_and_registered(sp_weights):
    det = SuperPointDetectorDescriptor(max_keypoints=123, weights_path=sp_weights)
    assert isinstance(det, DetectorDescriptorBase) and isinstance(det, GTSFMProcess)
    assert det.max_keypoints == 123 and det._model is None
    clone = pickle.loads(pickle.dumps(det))
    assert clone.max_keypoints == 123 and clone._model is None
    meta = det.get_ui_metadata()
    assert meta.display_name == "DetectorDescriptor" and meta.output_products == ("Keypoints", "Descriptors")
    assert type(det).__name__ == "SuperPointDetectorDescriptor"  # cache-key convention (detector_descriptor_cacher.py:40)


def test_superpoint_plugin_has_no_cpu_fallback(sp_weights):
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    img = Image(value_array=synthetic.synthetic_gray_image(64, 64, 0))
    for use_cuda in (True, False):  # use_cuda=False (the reference's contract tests) is no CPU switch either: no GPU, no result
        det = SuperPointDetectorDescriptor(use_cuda=use_cuda, weights_path=sp_weights)
        with pytest.raises(RuntimeError):
            det.detect_and_describe(img)


def test_strict_use_cuda_switch(monkeypatch):
    """GTSFM_AMD_STRICT_USE_CUDA=1 turns ``use_cuda=False`` into an error before any device is touched (a caller
    that uses the flag to keep a worker off the GPU must be able to rely on it)."""
    from gtsfm_amd.frontend.registry import warn_if_cpu_requested

    monkeypatch.setenv("GTSFM_AMD_STRICT_USE_CUDA", "1")
    warn_if_cpu_requested(True, "SuperPointDetectorDescriptor")  # use_cuda=True: nothing to say
    with pytest.raises(RuntimeError, match="STRICT_USE_CUDA"):
        warn_if_cpu_requested(False, "SuperPointDetectorDescriptor")


def test_keypoints_top_k_and_mask():
    coords = np.array([[1, 1], [2, 3], [5, 5], [7, 2]], dtype=np.float32)
    resp = np.array([0.1, 0.9, 0.5, 0.7], dtype=np.float32)
    kp = Keypoints(coords, responses=resp)
    top, idx = kp.get_top_k(2)
    assert len(top) == 2 and set(idx.tolist()) == {1, 3}
    allk, idx = kp.get_top_k(10)
    assert allk == kp and idx.tolist() == [0, 1, 2, 3]
    mask = np.zeros((8, 8), dtype=np.uint8)
    mask[3, 2] = mask[2, 7] = 1
    kept, idx = kp.filter_by_mask(mask)
    assert idx.tolist() == [1, 3] and len(kept) == 2
    assert pickle.loads(pickle.dumps(kp)) == kp


def test_rgb_to_gray_matches_fixed_point_formula():
    rng = np.random.default_rng(0)
    rgb = rng.integers(0, 256, size=(5, 7, 3), dtype=np.uint8)
    gray = rgb_to_gray_u8(rgb)
    assert gray.shape == (5, 7) and gray.dtype == np.uint8
    ref = np.rint(0.299 * rgb[..., 0] + 0.587 * rgb[..., 1] + 0.114 * rgb[..., 2])
    assert np.abs(gray.astype(int) - ref.astype(int)).max() <= 1
    assert rgb_to_gray_u8(gray) is gray


# ----------------------------------------------------------------------------------------------------------------------
# matchers
# ----------------------------------------------------------------------------------------------------------------------


def test_matcher_plugins_lazy_picklable_registered(tmp_path):
    """tests/frontend/matcher/test_matcher_base.py:102-107 (pickle) + registry / cache-key naming."""
    from gtsfm_amd.frontend.matcher.lightglue_matcher import LightGlueMatcher
    from gtsfm_amd.frontend.matcher.matcher_base import MatcherBase
    from gtsfm_amd.frontend.matcher.superglue_matcher import SuperGlueMatcher

    sg_path = tmp_path / "superglue_outdoor.pth"
    torch.save(synthetic.synthetic_superglue_state_dict(num_layers=2), str(sg_path))
    with pytest.raises(FileNotFoundError):
        SuperGlueMatcher(weights_path=tmp_path / "nope.pth")
    sg = SuperGlueMatcher(use_cuda=True, use_outdoor_model=True, weights_path=sg_path)
    lg = LightGlueMatcher(features="superpoint")  # lazy like the reference: nothing is touched at construction
    for obj, name in ((sg, "SuperGlueMatcher"), (lg, "LightGlueMatcher")):
        assert isinstance(obj, MatcherBase) and isinstance(obj, GTSFMProcess)
        assert type(obj).__name__ == name and obj._model is None
        clone = pickle.loads(pickle.dumps(obj))
        assert clone._model is None
        assert obj.get_ui_metadata().display_name == "Matcher"
    assert sg._config["sinkhorn_iterations"] == 20  # gtsfm/frontend/matcher/superglue_matcher.py:27


def test_matcher_input_validation_needs_no_gpu(tmp_path):
    from gtsfm_amd.frontend.matcher.superglue_matcher import SuperGlueMatcher

    sg_path = tmp_path / "superglue_outdoor.pth"
    torch.save(synthetic.synthetic_superglue_state_dict(num_layers=1), str(sg_path))
    sg = SuperGlueMatcher(weights_path=sg_path)
    k = np.zeros((3, 2), dtype=np.float32)
    with pytest.raises(ValueError):
        sg.match(Keypoints(k), Keypoints(k), np.zeros((3, 256), np.float32), np.zeros((3, 256), np.float32), (8, 8, 3), (8, 8, 3))
    r = np.ones(3, dtype=np.float32)
    with pytest.raises(Exception, match="256"):
        sg.match(Keypoints(k, responses=r), Keypoints(k, responses=r), np.zeros((3, 128), np.float32), np.zeros((3, 128), np.float32), (8, 8, 3), (8, 8, 3))


def test_same_named_classes_coexist_in_the_reference_registry():
    """``gtsfm/ui/registry.py:15-45`` keys its global registry on ``__name__``; the plugins here carry the reference's class names on
    purpose (shared cache namespace, section 8b). With GTSfM importable both ``gtsfm.frontend...SuperPointDetectorDescriptor`` and
    this package's class are defined under one metaclass: the two classes stay distinct and usable, and the registry (which only the
    UI's process-graph renderer reads, ``gtsfm/ui/process_graph_generator.py``) holds whichever was defined LAST under the shared
    name. A fake ``gtsfm.ui`` restating that metaclass stands in for GTSfM (own interpreter: ``sys.modules`` stays clean here)."""
    import subprocess
    import sys
    import textwrap
    from pathlib import Path

    repo = Path(__file__).resolve().parent.parent
    script = textwrap.dedent("""
        import abc, sys, types
        from dataclasses import dataclass
        from typing import Optional, Tuple
        for name in ("gtsfm", "gtsfm.ui", "gtsfm.ui.registry", "gtsfm.ui.gtsfm_process"):
            mod = types.ModuleType(name); mod.__path__ = []; sys.modules[name] = mod
        class RegistryHolder(type):                                   # gtsfm/ui/registry.py:15-40
            REGISTRY = {}
            def __new__(cls, name, bases, attrs):
                new_cls = type.__new__(cls, name, bases, attrs)
                cls.REGISTRY[new_cls.__name__] = new_cls
                return new_cls
            @classmethod
            def get_registry(cls):
                return dict(cls.REGISTRY)
        class AbstractableRegistryHolder(abc.ABCMeta, RegistryHolder):  # gtsfm/ui/registry.py:43-45
            pass
        @dataclass(frozen=True, order=True)
        class UiMetadata:                                              # gtsfm/ui/gtsfm_process.py:36-52
            display_name: str
            input_products: Tuple[str, ...]
            output_products: Tuple[str, ...]
            parent_plate: Optional[str] = None
        class GTSFMProcess(metaclass=AbstractableRegistryHolder):      # gtsfm/ui/gtsfm_process.py:55-65
            @staticmethod
            @abc.abstractmethod
            def get_ui_metadata():
                ...
        sys.modules["gtsfm.ui.registry"].RegistryHolder = RegistryHolder
        sys.modules["gtsfm.ui.registry"].AbstractableRegistryHolder = AbstractableRegistryHolder
        sys.modules["gtsfm.ui.gtsfm_process"].GTSFMProcess = GTSFMProcess
        sys.modules["gtsfm.ui.gtsfm_process"].UiMetadata = UiMetadata
        class SuperPointDetectorDescriptor(GTSFMProcess):              # stands for the reference's class of that name
            @staticmethod
            def get_ui_metadata():
                return UiMetadata("DetectorDescriptor", ("Images",), ("Keypoints", "Descriptors"))
        reference_cls = SuperPointDetectorDescriptor
        assert RegistryHolder.get_registry()["SuperPointDetectorDescriptor"] is reference_cls
        from gtsfm_amd.frontend.detector_descriptor.superpoint import SuperPointDetectorDescriptor as amd_cls
        from gtsfm_amd.frontend.matcher.lightglue_matcher import LightGlueMatcher
        from gtsfm_amd.frontend.registry import GTSFMProcess as used
        assert used is GTSFMProcess, "the plugins must derive from GTSfM's own process base when it is importable"
        reg = RegistryHolder.get_registry()
        assert amd_cls is not reference_cls and issubclass(amd_cls, GTSFMProcess) and issubclass(reference_cls, GTSFMProcess)
        assert reg["SuperPointDetectorDescriptor"] is amd_cls          # last definition wins the NAME ...
        assert reference_cls.get_ui_metadata().display_name == amd_cls.get_ui_metadata().display_name  # ... the UI node it names is the same either way
        assert reference_cls().get_ui_metadata().display_name == "DetectorDescriptor"  # the shadowed class still works
        assert reg["LightGlueMatcher"] is LightGlueMatcher
        print("registry OK")
    """)
    run = subprocess.run([sys.executable, "-c", script], capture_output=True, text=True, cwd=str(repo), timeout=300)
    assert run.returncode == 0 and "registry OK" in run.stdout, (run.stdout, run.stderr[-3000:])
