"""CPU, build container only: the oracle restatements against the REFERENCE'S OWN model files, run live on inputs no committed fixture holds.

``tests/golden/*.npz`` were written by ``oracle/validate_against_reference.py`` from the same model files; this test repeats the comparison on fresh
seeds, odd sizes and ragged keypoint counts each time the CPU suite runs where ``/root/reference`` is mounted, so a change to ``oracle/`` cannot
drift from the reference between two regenerations of the fixtures. It is skipped on the GPU box (no reference there; nothing under ``-m gpu``,
``smoke()`` or ``bench.py`` reads it). Bit-exact: same torch, same operations, same order (thirdparty/SuperGluePretrainedNetwork/models/
superpoint.py:145-202, superglue.py:228-283)."""

import os
from pathlib import Path

import numpy as np
import pytest
import torch

REFERENCE = Path(os.environ.get("GTSFM_REFERENCE", "/root/reference"))
MODELS = REFERENCE / "thirdparty" / "SuperGluePretrainedNetwork" / "models"
pytestmark = pytest.mark.skipif(not (MODELS / "superpoint.py").exists(), reason="the reference tree is only mounted in the build container")


@pytest.fixture(scope="module")
def ref():
    from oracle import validate_against_reference as V

    return V


@pytest.mark.parametrize("h,w,seed", [(97, 131, 101), (64, 200, 102), (136, 120, 103)])
def test_superpoint_restatement_equals_the_reference_model_on_fresh_images(ref, h, w, seed):
    from gtsfm_amd.utils import synthetic
    from oracle import superpoint_oracle

    sd = synthetic.synthetic_superpoint_state_dict()
    model = ref.reference_superpoint(sd)
    img = superpoint_oracle.gray_u8_to_tensor(synthetic.synthetic_gray_image(h, w, seed))
    with torch.no_grad(), ref._force_align_corners():
        out = model({"image": img})
        ora = superpoint_oracle.superpoint_forward(sd, img)
    assert out["keypoints"][0].shape[0] > 20
    assert torch.equal(out["keypoints"][0], ora["keypoints"])
    assert torch.equal(out["scores"][0], ora["scores"])
    assert torch.equal(out["descriptors"][0], ora["descriptors"])


@pytest.mark.parametrize("n0,n1,iters,seed", [(150, 97, 20, 201), (33, 260, 100, 202), (1, 50, 20, 203)])
def test_superglue_restatement_equals_the_reference_model_on_fresh_pairs(ref, n0, n1, iters, seed):
    from gtsfm_amd.utils import synthetic
    from oracle import superglue_oracle

    sd = synthetic.synthetic_superglue_state_dict(num_layers=ref.SUPERGLUE_LAYERS_GOLDEN)
    model = ref.reference_superglue(sd, iters)
    shp0, shp1 = (240, 320), (200, 304)
    k0, s0, d0, k1, s1, d1, _ = synthetic.synthetic_pair_features(n0, n1, shp0, shp1, seed=seed)
    data = {
        "keypoints0": torch.from_numpy(k0)[None], "keypoints1": torch.from_numpy(k1)[None],
        "descriptors0": torch.from_numpy(d0).T[None].contiguous(), "descriptors1": torch.from_numpy(d1).T[None].contiguous(),
        "scores0": torch.from_numpy(s0)[None], "scores1": torch.from_numpy(s1)[None],
        "image0": torch.empty((1, 1) + shp0), "image1": torch.empty((1, 1) + shp1),
    }
    with torch.no_grad():
        out = model(data)
        ora = superglue_oracle.superglue_forward(sd, data["keypoints0"], data["keypoints1"], data["scores0"], data["scores1"],
                                                 data["descriptors0"], data["descriptors1"], shp0, shp1, sinkhorn_iterations=iters)
    for key in ("matches0", "matches1", "matching_scores0", "matching_scores1"):
        assert out[key].dtype == ora[key].dtype and torch.equal(out[key], ora[key]), key


def _reference_keypoints_class():
    """gtsfm/common/keypoints.py imports cv2 for one method (cast_to_opencv_keypoints) that the path never calls: a stub module stands in."""
    import importlib.util
    import sys
    import types

    had = "cv2" in sys.modules
    if not had:
        sys.modules["cv2"] = types.ModuleType("cv2")
        sys.modules["cv2"].KeyPoint = object
    try:
        spec = importlib.util.spec_from_file_location("ref_keypoints", str(REFERENCE / "gtsfm" / "common" / "keypoints.py"))
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
    finally:
        if not had:
            del sys.modules["cv2"]
    return mod.Keypoints


@pytest.mark.parametrize("seed", range(6))
def test_keypoints_stand_in_equals_the_reference_class(seed):
    """Row a19: ``get_top_k`` / ``filter_by_mask`` / ``extract_indices`` of the stand-in class used when GTSfM is not
    importable (gtsfm_amd/common/keypoints.py) against gtsfm/common/keypoints.py:87-127 -- same selections in the same order, same dtypes,
    incl. ties in the responses, k >= n, missing responses / scales, an empty set and keypoints exactly on .5 coordinates (np.round: half to even)."""
    from gtsfm_amd.common import keypoints as K

    if not getattr(K, "USING_STAND_IN", True):
        pytest.skip("GTSfM's own Keypoints is importable: nothing to compare")
    Ref = _reference_keypoints_class()
    rng = np.random.default_rng(seed)
    n = [0, 1, 37, 500, 500, 2000][seed]
    h, w = 120, 160
    coords = np.stack([rng.uniform(0, w - 1, n), rng.uniform(0, h - 1, n)], 1).astype(np.float32)
    if n >= 37:
        coords[::5] = np.floor(coords[::5]) + 0.5  # rounding ties
        coords[:, 0] = np.clip(coords[:, 0], 0, w - 1.5)
        coords[:, 1] = np.clip(coords[:, 1], 0, h - 1.5)
    resp = rng.random(n).astype(np.float32)
    if seed == 4:
        resp = np.round(resp * 8) / 8  # many exact ties: argpartition's choice among them must be the same call on the same data
    scales = rng.random(n).astype(np.float32) if seed % 2 else None
    mask = (rng.random((h, w)) < 0.7).astype(np.uint8)
    for responses in (resp, None):
        ours, theirs = K.Keypoints(coords.copy(), scales, responses), Ref(coords.copy(), scales, responses)
        assert len(ours) == len(theirs) == n
        for k in (0, 1, n // 3, n, n + 5):
            (a, ia), (b, ib) = ours.get_top_k(k), theirs.get_top_k(k)
            assert ia.dtype == ib.dtype and np.array_equal(ia, ib), (k, responses is None)
            assert np.array_equal(a.coordinates, b.coordinates) and len(a) == len(b)
            for fa, fb in ((a.scales, b.scales), (a.responses, b.responses)):
                assert (fa is None) == (fb is None) and (fa is None or (fa.dtype == fb.dtype and np.array_equal(fa, fb)))
        (a, ia), (b, ib) = ours.filter_by_mask(mask), theirs.filter_by_mask(mask)
        assert ia.dtype == ib.dtype and np.array_equal(ia, ib)
        assert np.array_equal(a.coordinates, b.coordinates)
        assert np.array_equal(ours.get_x_coordinates(), theirs.get_x_coordinates()) and np.array_equal(ours.get_y_coordinates(), theirs.get_y_coordinates())
        assert (ours == K.Keypoints(coords.copy(), scales, responses)) and (theirs == Ref(coords.copy(), scales, responses))
        if n:
            assert ours != K.Keypoints(coords + 1, scales, responses) and theirs != Ref(coords + 1, scales, responses)


def test_reference_plugin_classes_return_what_the_restatement_and_the_config1_golden_hold():
    """``oracle/validate_wrappers_against_reference.py`` in a process of its own (it makes ``gtsfm`` importable with cv2 / gtsam stubbed, which
    must not leak into this one): the reference's ``SuperPointDetectorDescriptor.detect_and_describe`` and ``SuperGlueMatcher.match`` -- the two
    plugin classes the HIP plugins replace -- run live on the CPU and equal the oracle's restatement bit for bit; on the
    first two Lund-door frames and their pair they return exactly the arrays ``tests/golden/lund_door_config1.npz`` holds."""
    import subprocess
    import sys

    repo = Path(__file__).resolve().parents[1]
    out = subprocess.run([sys.executable, str(repo / "oracle" / "validate_wrappers_against_reference.py"), "--frames", "2", "--pairs", "1"],
                         capture_output=True, text=True, timeout=900, cwd=str(repo))
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert out.stdout.rstrip().endswith("OK") and "lund door pair (0,1): the reference wrapper returns the golden's 243 matches" in out.stdout
    assert out.stdout.count("wrapper == restatement") >= 9
