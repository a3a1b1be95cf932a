"""-m gpu: the reference's OWN plugin contract tests, restated against the gtsfm_amd plugins (the reference constructs
``SuperPointDetectorDescriptor(use_cuda=False)``; this package has no CPU path: with a GPU present such an object runs on it and
warns, ``test_use_cuda_false_as_the_reference_tests_construct_it`` -- INTEGRATION.md section 1):

* ``tests/frontend/detector/test_detector_base.py:27-56``  (number of detections, coordinate range, scales, pickling)
* ``tests/frontend/detector_descriptor/test_detector_descriptor_base.py:29-42``  (keypoints <-> descriptors)
* ``tests/frontend/matcher/test_matcher_base.py:51-107``  (empty input, valid indices, one-to-one, pickling)
* ``tests/frontend/matcher/test_superglue_matcher.py:24-41``  (random input of the reference's dtypes -> uint32 array)

on the reference's fixture images where they travel: two Lund-door photographs (``tests/data/set1_lund_door/images``), stored
downsized in ``tests/golden/lund_door_pair.npz`` by ``oracle/validate_against_reference.py`` (/root/reference does not exist on the
GPU box)."""

import pickle

import numpy as np
import pytest
import torch

from gtsfm_amd.common.image import Image
from gtsfm_amd.common.keypoints import Keypoints
from gtsfm_amd.utils import synthetic

from conftest import GOLDEN

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def lund_images():
    g = np.load(GOLDEN / "lund_door_pair.npz")
    # the loader hands RGB images over (olsson_loader -> Image(value_array HxWx3 uint8)); the photographs are stored gray
    return [Image(value_array=np.repeat(g[k][:, :, None], 3, axis=2), file_name=f"DSC_000{i + 1}.JPG") for i, k in enumerate(("gray0", "gray1"))]


@pytest.fixture(scope="module")
def plugins(tmp_path_factory, gpu_device):
    from gtsfm_amd.frontend.detector_descriptor.superpoint import SuperPointDetectorDescriptor
    from gtsfm_amd.frontend.matcher.lightglue_matcher import LightGlueMatcher
    from gtsfm_amd.frontend.matcher.superglue_matcher import SuperGlueMatcher

    tmp = tmp_path_factory.mktemp("weights")
    torch.save(synthetic.synthetic_superpoint_state_dict(), str(tmp / "sp.pth"))
    torch.save(synthetic.synthetic_superglue_state_dict(num_layers=4), str(tmp / "sg.pth"))
    torch.save(synthetic.synthetic_lightglue_state_dict(num_layers=3), str(tmp / "lg.pth"))
    return {
        "detector_descriptor": SuperPointDetectorDescriptor(max_keypoints=500, use_cuda=True, weights_path=tmp / "sp.pth"),
        "superglue": SuperGlueMatcher(use_cuda=True, weights_path=tmp / "sg.pth"),
        "lightglue": LightGlueMatcher("superpoint", use_cuda=True, weights_path=tmp / "lg.pth"),
    }


# ---- test_detector_base.py:27-56 + test_detector_descriptor_base.py:29-42 ---------------------------------------------


def test_number_of_detections_coordinates_range_scales_and_shapes(plugins, lund_images):
    det = plugins["detector_descriptor"]
    for image in lund_images:
        keypoints, descriptors = det.detect_and_describe(image)
        assert 0 < len(keypoints) <= det.max_keypoints                       # test_number_of_detections
        c = keypoints.coordinates
        assert np.all(c[:, 0] >= 0) and np.all(c[:, 0] <= image.width)     # test_coordinates_range
        assert np.all(c[:, 1] >= 0) and np.all(c[:, 1] <= image.height)
        assert keypoints.scales is None or np.all(keypoints.scales >= 0)     # test_scale
        assert len(keypoints) == descriptors.shape[0]                        # test_detect_and_describe_shape
        assert c.dtype == np.float32 and descriptors.dtype == np.float32 and descriptors.shape[1] == 256 and keypoints.responses.shape == (len(keypoints),)
    # DetectorFromDetectorDescriptor (detector_from_joint_detector_descriptor.py): detect() is detect_and_describe()[0]
    again, _ = det.detect_and_describe(lund_images[0])
    first, _ = det.detect_and_describe(lund_images[0])
    assert again == first                                                    # repro_tests/.../test_superpoint.py: deterministic


def test_detector_and_matchers_pickle_with_a_loaded_model(plugins, lund_images):
    """test_pickleable -- here AFTER first use, when the engines hold device state (Dask re-scatters live objects)."""
    plugins["detector_descriptor"].detect_and_describe(lund_images[0])
    for obj in plugins.values():
        clone = pickle.loads(pickle.dumps(obj))
        assert type(clone) is type(obj) and clone._model is None


# ---- test_matcher_base.py:51-107, test_superglue_matcher.py:24-41 -----------------------------------------------------


@pytest.mark.parametrize("which", ["superglue", "lightglue"])
def test_empty_input(plugins, which):
    matcher = plugins[which]
    rng = np.random.default_rng(0)
    some = Keypoints(coordinates=rng.uniform(0, 100, (9, 2)).astype(np.float32), responses=rng.uniform(0, 1, 9).astype(np.float32))
    some_desc = rng.standard_normal((9, 256)).astype(np.float32)
    none = Keypoints(coordinates=np.zeros((0, 2), dtype=np.float32), responses=np.zeros(0, dtype=np.float32))
    none_desc = np.zeros((0, 256), dtype=np.float32)
    shape = (300, 200, 3)
    for a, da, b, db in ((none, none_desc, some, some_desc), (some, some_desc, none, none_desc), (none, none_desc, none, none_desc)):
        assert matcher.match(a, b, da, db, shape, shape).size == 0


@pytest.mark.parametrize("which", ["superglue", "lightglue"])
def test_match_validity_on_real_image_features(plugins, lund_images, which):
    det, matcher = plugins["detector_descriptor"], plugins[which]
    (k1, d1), (k2, d2) = (det.detect_and_describe(im) for im in lund_images)
    m = matcher.match(k1, k2, d1, d2, lund_images[0].shape, lund_images[1].shape)
    assert m.ndim == 2 and m.shape[1] == 2 and m.shape[0] > 10
    assert np.all((m[:, 0] >= 0) & (m[:, 0] < len(k1))) and np.all((m[:, 1] >= 0) & (m[:, 1] < len(k2)))     # __assert_valid_indices
    assert len(set(m[:, 0].tolist())) == m.shape[0] == len(set(m[:, 1].tolist()))                              # __assert_one_to_one_constraint
    assert m.shape[0] < min(len(k1), len(k2))                                                                  # matcher_base.py:51-53


def test_on_dummy_data_of_the_reference_dtypes(plugins):
    """test_superglue_matcher.py:24-41: int64 coordinates from np.random.randint, float64 responses and descriptors."""
    rng = np.random.RandomState(0)
    h = w = 20
    k1 = Keypoints(coordinates=rng.randint(0, h, size=(50, 2)), responses=rng.rand(50))
    k2 = Keypoints(coordinates=rng.randint(0, h, size=(100, 2)), responses=rng.rand(100))
    d1, d2 = rng.randn(50, 256), rng.randn(100, 256)
    m = plugins["superglue"].match(k1, k2, d1, d2, (h, w, 3), (h, w, 3))
    assert isinstance(m, np.ndarray) and m.dtype == np.uint32
    m = plugins["lightglue"].match(k1, k2, d1, d2, (h, w, 3), (h, w, 3))
    assert isinstance(m, np.ndarray) and m.dtype == np.int64 and (m.size == 0 or m.shape[1] == 2)


def test_use_cuda_false_as_the_reference_tests_construct_it(plugins, lund_images, tmp_path):
    """tests/frontend/detector_descriptor/test_superpoint.py:18 builds ``SuperPointDetectorDescriptor(use_cuda=False)``. There is no CPU
    path here: on a GPU box the object runs on the GPU, says so (RuntimeWarning), and returns exactly what ``use_cuda=True`` returns."""
    from gtsfm_amd.frontend.detector_descriptor.superpoint import SuperPointDetectorDescriptor
    from gtsfm_amd.frontend.matcher.superglue_matcher import SuperGlueMatcher

    ref = plugins["detector_descriptor"]
    det = SuperPointDetectorDescriptor(max_keypoints=ref.max_keypoints, use_cuda=False, weights_path=ref._config["weights_path"])
    with pytest.warns(RuntimeWarning, match="no CPU path"):
        kp, desc = det.detect_and_describe(lund_images[0])
    kp_ref, desc_ref = ref.detect_and_describe(lund_images[0])
    assert kp == kp_ref and np.array_equal(desc, desc_ref)
    sg = SuperGlueMatcher(use_cuda=False, weights_path=plugins["superglue"]._weights_path)
    kp2, desc2 = ref.detect_and_describe(lund_images[1])
    with pytest.warns(RuntimeWarning, match="no CPU path"):
        m = sg.match(kp, kp2, desc, desc2, lund_images[0].shape, lund_images[1].shape)
    np.testing.assert_array_equal(m, plugins["superglue"].match(kp, kp2, desc, desc2, lund_images[0].shape, lund_images[1].shape))
