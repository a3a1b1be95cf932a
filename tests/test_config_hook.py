"""CPU: the config hook -- GTSfM builds its front end from ``_target_`` entries of a Hydra YAML
(``gtsfm/configs/deep_front_end.yaml:22-35``); the integration is a change of those entries. Hydra / omegaconf are not installed here, so the
few lines of ``hydra.utils.instantiate`` this needs are restated below (recursive ``_target_`` resolution, keyword arguments, overrides) and
``gtsfm_amd/configs/deep_front_end_amd.yaml`` -- the reference's front-end subtree with this package's targets -- is instantiated through
them: the object graph has the reference's shape, the arguments land where the reference's signatures put them, and everything pickles
before any device state exists (the graph is scattered to Dask workers, ``det_desc_correspondence_generator.py:65-68``)."""

import importlib
import pickle
from pathlib import Path

import pytest
import torch
import yaml

from gtsfm_amd.utils import synthetic
from tests.conftest import REPO

CONFIGS = REPO / "gtsfm_amd" / "configs"
CONFIG = CONFIGS / "deep_front_end_amd.yaml"


def _load(name):
    """``yaml.safe_load`` = ONE document per file, as Hydra / OmegaConf require (the two variants used to share a multi-document file)."""
    return yaml.safe_load((CONFIGS / name).read_text())


def instantiate(node, overrides=None, path=""):
    """The subset of hydra.utils.instantiate GTSfM's configs use: a mapping with ``_target_`` is a call of that dotted name with the other
    entries as keyword arguments, nested mappings first; ``overrides`` maps dotted config paths to values (Hydra's command-line overrides)."""
    overrides = overrides or {}
    if isinstance(node, dict):
        built = {k: instantiate(v, overrides, f"{path}.{k}" if path else k) for k, v in node.items() if k != "_target_"}
        for key, value in overrides.items():
            prefix, _, leaf = key.rpartition(".")
            if prefix == path:
                built[leaf] = value
        if "_target_" not in node:
            return built
        module, _, name = node["_target_"].rpartition(".")
        return getattr(importlib.import_module(module), name)(**built)
    if isinstance(node, list):
        return [instantiate(v, overrides, f"{path}.{i}") for i, v in enumerate(node)]
    return node


@pytest.fixture(scope="module")
def weights(tmp_path_factory):
    tmp = tmp_path_factory.mktemp("weights")
    torch.save(synthetic.synthetic_superpoint_state_dict(), str(tmp / "sp.pth"))
    torch.save(synthetic.synthetic_superglue_state_dict(num_layers=2), str(tmp / "sg.pth"))
    torch.save(synthetic.synthetic_lightglue_state_dict(num_layers=2), str(tmp / "lg.pth"))
    return tmp


def test_front_end_subtree_instantiates_like_the_reference_config(weights):
    from gtsfm_amd.frontend.cacher.detector_descriptor_cacher import DetectorDescriptorCacher
    from gtsfm_amd.frontend.cacher.matcher_cacher import MatcherCacher
    from gtsfm_amd.frontend.correspondence_generator.batched_det_desc_correspondence_generator import BatchedDetDescCorrespondenceGenerator
    from gtsfm_amd.frontend.correspondence_generator.det_desc_correspondence_generator import DetDescCorrespondenceGenerator
    from gtsfm_amd.frontend.detector_descriptor.superpoint import SuperPointDetectorDescriptor
    from gtsfm_amd.frontend.matcher.lightglue_matcher import LightGlueMatcher
    from gtsfm_amd.frontend.matcher.superglue_matcher import SuperGlueMatcher

    from gtsfm_amd.frontend.correspondence_generator.sharded_det_desc_correspondence_generator import ShardedDetDescCorrespondenceGenerator

    per_pair, batched, sharded = _load("deep_front_end_amd.yaml"), _load("deep_front_end_amd_batched.yaml"), _load("deep_front_end_amd_sharded.yaml")
    # variant A: the reference's graph shape -- generator(cacher(SuperPoint), cacher(LightGlue))
    gen = instantiate(per_pair, {
        "correspondence_generator.detector_descriptor.detector_descriptor_obj.weights_path": weights / "sp.pth",
        "correspondence_generator.matcher.matcher_obj.weights_path": weights / "lg.pth",
    })["correspondence_generator"]
    assert isinstance(gen, DetDescCorrespondenceGenerator)
    assert isinstance(gen._detector_descriptor, DetectorDescriptorCacher) and isinstance(gen._matcher, MatcherCacher)
    det, lg = gen._detector_descriptor._detector_descriptor, gen._matcher._matcher
    assert isinstance(det, SuperPointDetectorDescriptor) and det.max_keypoints == 5000 == gen._detector_descriptor.max_keypoints
    assert isinstance(lg, LightGlueMatcher) and lg._features == "superpoint" and lg._model is None and det._model is None  # lazy: nothing touched a device
    clone = pickle.loads(pickle.dumps(gen))  # scattered to the workers before first use
    assert type(clone._matcher._matcher) is LightGlueMatcher and "DetDescCorrespondenceGenerator" in repr(clone)
    # variant B: the GPU-resident generator takes the plugins themselves
    gen_b = instantiate(batched, {
        "correspondence_generator.detector_descriptor.weights_path": weights / "sp.pth",
        "correspondence_generator.matcher.weights_path": weights / "sg.pth",
    })["correspondence_generator"]
    assert isinstance(gen_b, BatchedDetDescCorrespondenceGenerator) and isinstance(gen_b._matcher, SuperGlueMatcher)
    assert gen_b._matcher._config["weights"] == "outdoor" and gen_b._detector_descriptor.max_keypoints == 5000
    # variant C: one scene sharded over the node's GPUs -- same two plugin arguments, ranks are started at the first call, not here
    gen_c = instantiate(sharded, {
        "correspondence_generator.detector_descriptor.weights_path": weights / "sp.pth",
        "correspondence_generator.matcher.weights_path": weights / "sg.pth",
    })["correspondence_generator"]
    assert isinstance(gen_c, ShardedDetDescCorrespondenceGenerator) and isinstance(gen_c._matcher, SuperGlueMatcher) and gen_c._num_gpus is None
    assert gen_c._pool is None and gen_c._pipe is None and gen_c._detector_descriptor._model is None
    assert type(pickle.loads(pickle.dumps(gen_c))._detector_descriptor) is SuperPointDetectorDescriptor
    # a missing checkpoint fails at instantiation, like the reference's constructor (gtsfm/frontend/detector_descriptor/superpoint.py:47-53)
    with pytest.raises(FileNotFoundError):
        instantiate(per_pair, {"correspondence_generator.detector_descriptor.detector_descriptor_obj.weights_path": Path("/nonexistent/sp.pth")})


def test_per_pair_generator_without_a_scheduler_calls_the_plugins_in_graph_order():
    """``DetDescCorrespondenceGenerator.generate_correspondences(None, ...)``: one ``detect_and_describe`` per image, one keyword-argument
    ``match`` per edge (``im_shape_i1=`` / ``im_shape_i2=`` as the reference passes them, det_desc_correspondence_generator.py:72-79)."""
    import numpy as np

    from gtsfm_amd.common.image import Image
    from gtsfm_amd.common.keypoints import Keypoints
    from gtsfm_amd.frontend.correspondence_generator.det_desc_correspondence_generator import DetDescCorrespondenceGenerator

    calls = []

    class Det:
        def detect_and_describe(self, image):
            calls.append(("det", image.file_name))
            n = int(image.value_array[0, 0, 0])
            return Keypoints(np.zeros((n, 2), dtype=np.float32), responses=np.ones(n, dtype=np.float32)), np.zeros((n, 256), dtype=np.float32)

    class Mat:
        def match(self, k1, k2, d1, d2, im_shape_i1, im_shape_i2):
            calls.append(("match", len(k1), len(k2), im_shape_i1, im_shape_i2))
            return np.zeros((min(len(k1), len(k2)), 2), dtype=np.uint32)

    images = [Image(value_array=np.full((4 + i, 6, 3), 3 + i, dtype=np.uint8), file_name=f"{i}.jpg") for i in range(3)]
    kps, putative = DetDescCorrespondenceGenerator(Mat(), Det()).generate_correspondences(None, images, [(0, 2), (1, 2)])
    assert [len(k) for k in kps] == [3, 4, 5] and sorted(putative) == [(0, 2), (1, 2)] and putative[(1, 2)].shape == (4, 2)
    assert calls == [("det", "0.jpg"), ("det", "1.jpg"), ("det", "2.jpg"), ("match", 3, 5, (4, 6, 3), (6, 6, 3)), ("match", 4, 5, (5, 6, 3), (6, 6, 3))]
