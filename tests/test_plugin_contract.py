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
    det = SuperPointDetectorDescriptor(weights_path=sp_weights)
    img = Image(value_array=synthetic.synthetic_gray_image(64, 64, 0))
    with pytest.raises(RuntimeError):
        det.detect_and_describe(img)


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
