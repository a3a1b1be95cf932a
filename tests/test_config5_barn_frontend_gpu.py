"""-m gpu: the FRONT-END SLICE of BASELINE config 5 on the reference's own Barn fixture -- the three Tanks-and-Temples frames of
``tests/data/tanks_and_temples_barn`` at the COLMAP loader's resolution (``max_resolution: 760`` -> 760 x 1351) through the chain
``gtsfm/configs/deep_front_end.yaml:22-49`` wires ahead of gtsam: ``SuperPointDetectorDescriptor(max_keypoints=5000)`` ->
``LightGlueMatcher("superpoint")`` -> ``Ransac(use_intrinsics_in_verification=True, estimation_threshold_px=4)``, called the way
``TwoViewEstimator.run_2view`` calls them (``gtsfm/two_view_estimator.py:350-397``: one ``match`` and one ``verify`` per pair, intrinsics from the loader),
and once more through the device-resident ``BatchedDetDescCorrespondenceGenerator.generate_correspondences_and_verify``.

Golden vectors: ``oracle/make_barn_config5_golden.py`` -- the REFERENCE SuperPoint model file + the wrapper's ``get_top_k`` for the detections, the LightGlue
restatement (upstream's source is absent from the reference: unpinned) and the verifier oracle (unpinned towards OpenCV's USAC) behind them. Config 5 itself
(Barn end to end through two-view BA, averaging and bundle adjustment in gtsam, real checkpoints) cannot run in this environment; this is the part of it that can.
Seeded synthetic weights: the matches are matches of random-weight descriptors, so no pose is compared with the fixture's COLMAP ground truth.
What the run saw is recorded (``gpurun_out/config5_barn_observed.json`` -> ``profiles/``) and asserted exactly."""

import io
import json

import numpy as np
import pytest
import torch

from conftest import GOLDEN, REPO
from gtsfm_amd.common.calibration import PinholeIntrinsics
from gtsfm_amd.common.image import Image
from gtsfm_amd.common.keypoints import Keypoints
from gtsfm_amd.utils import synthetic

pytestmark = pytest.mark.gpu

TOL = 1e-4
PAIRS = [(0, 1), (0, 2), (1, 2)]
OBSERVED = {}


def _record(key, value):
    OBSERVED[key] = value
    out = REPO / "gpurun_out"
    out.mkdir(exist_ok=True)
    (out / "config5_barn_observed.json").write_text(json.dumps(OBSERVED, indent=1, sort_keys=True))


@pytest.fixture(scope="module")
def golden():
    path = GOLDEN / "barn_config5_frontend.npz"
    if not path.exists():
        pytest.fail("tests/golden/barn_config5_frontend.npz is missing: run oracle/make_barn_config5_golden.py in the build container")
    from PIL import Image as PILImage

    g = dict(np.load(path))
    g["gray"] = np.stack([np.asarray(PILImage.open(io.BytesIO(g[f"gray_png_{i}"].tobytes()))) for i in range(3)])
    assert g["gray"].dtype == np.uint8 and g["gray"].shape == (3, 760, 1351)
    return g


@pytest.fixture(scope="module")
def chain(golden, tmp_path_factory, gpu_device):
    from gtsfm_amd.frontend.detector_descriptor.superpoint import SuperPointDetectorDescriptor
    from gtsfm_amd.frontend.matcher.lightglue_matcher import LightGlueMatcher

    tmp = tmp_path_factory.mktemp("barn_weights")
    torch.save(synthetic.synthetic_superpoint_state_dict(), str(tmp / "sp.pth"))
    torch.save(synthetic.synthetic_lightglue_state_dict(), str(tmp / "lg.pth"))
    det = SuperPointDetectorDescriptor(max_keypoints=5000, weights_path=tmp / "sp.pth")
    lg = LightGlueMatcher("superpoint", weights_path=tmp / "lg.pth")
    # the loader hands RGB uint8 over; the frames are stored gray (R = G = B: the fixed-point gray conversion is the identity)
    images = [Image(value_array=np.repeat(golden["gray"][i][:, :, None], 3, axis=2), file_name=str(golden["names"][i])) for i in range(3)]
    intrinsics = [PinholeIntrinsics(fx=row[0], u0=row[2], v0=row[3], fy=row[1]) for row in golden["intrinsics"]]
    return {"det": det, "lg": lg, "images": images, "intrinsics": intrinsics}


def _pixel_key(xy, width):
    xy = np.asarray(xy)
    return xy[:, 1].astype(np.int64) * width + xy[:, 0].astype(np.int64)


def test_run_2view_front_end_on_the_barn_frames(golden, chain):
    """detect_and_describe per frame, then per pair match + verify as run_2view does: the reference's 5000 keypoints per frame (as a set), the
    golden (K, 2) int64 match arrays, the verifier oracle's verified index arrays and poses."""
    from gtsfm_amd.frontend.verifier.ransac import Ransac

    det, lg = chain["det"], chain["lg"]
    width = chain["images"][0].width
    feats, strays = [], []
    for i, im in enumerate(chain["images"]):
        kps, desc = det.detect_and_describe(im)
        ref_xy, ref_sc, ref_head = golden[f"keypoints_{i}"].astype(np.float32), golden[f"scores_{i}"], golden[f"descriptors_head_{i}"]
        assert isinstance(kps, Keypoints) and len(kps) == 5000 == desc.shape[0] and kps.scales is None and desc.dtype == np.float32
        got_key, ref_key = _pixel_key(kps.coordinates, width), _pixel_key(ref_xy, width)
        strays.append(int(len(np.setxor1d(got_key, ref_key))))
        # the plugin's rows in the order the REFERENCE handed to its matcher (np.argpartition's: implementation-defined)
        order = np.argsort(got_key)
        rows = order[np.clip(np.searchsorted(got_key[order], ref_key), 0, len(order) - 1)]
        assert np.array_equal(got_key[rows], ref_key), f"frame {i}: keypoint sets differ"
        np.testing.assert_allclose(kps.responses[rows], ref_sc, rtol=0, atol=TOL)
        np.testing.assert_allclose(desc[rows[: len(ref_head)]], ref_head, rtol=0, atol=TOL)
        feats.append((Keypoints(kps.coordinates[rows], scales=None, responses=kps.responses[rows]), np.ascontiguousarray(desc[rows])))
    _record("plugin_detect.stray_keypoints_per_frame", strays)
    assert strays == [0, 0, 0]

    shape = (chain["images"][0].height, width, 3)
    differing, verified_equal, total = [], [], 0
    for i, j in PAIRS:
        got = lg.match(feats[i][0], feats[j][0], feats[i][1], feats[j][1], shape, shape)
        ref = golden[f"matches_{i}_{j}"].astype(np.int64)
        assert got.dtype == np.int64 and got.ndim == 2 and got.shape[1] == 2
        total += len(ref)
        diff = set(map(tuple, got.tolist())) ^ set(map(tuple, ref.tolist()))
        differing.append(len(diff))
        ver = Ransac(True, float(golden["threshold_px"]), seed=(i << 32) | j)
        rot, direction, v_idx, ratio = ver.verify(feats[i][0], feats[j][0], got, chain["intrinsics"][i], chain["intrinsics"][j])
        if not diff:  # same putative matches -> the device verifier must pick the oracle's samples, winner and inliers bit for bit
            np.testing.assert_array_equal(v_idx, golden[f"v_corr_idxs_{i}_{j}"].astype(np.int64))
            assert ratio == float(golden[f"inlier_ratio_{i}_{j}"])
            assert bool(golden[f"has_pose_{i}_{j}"]) == (rot is not None)
            if rot is not None:
                np.testing.assert_allclose(np.asarray(rot), golden[f"R_{i}_{j}"], rtol=0, atol=1e-9)
                np.testing.assert_allclose(np.asarray(direction).reshape(3), golden[f"t_{i}_{j}"], rtol=0, atol=1e-9)
            verified_equal.append(True)
        else:
            verified_equal.append(False)
    _record("run_2view.reference_matches_over_3_pairs", int(total))
    _record("run_2view.differing_matches_per_pair", differing)
    _record("run_2view.verified_sets_equal_per_pair", verified_equal)
    assert total > 50
    assert differing == [0, 0, 0] and verified_equal == [True, True, True]


def test_device_resident_generator_and_verifier_on_the_barn_frames(golden, chain):
    """The same chain with features, matches and verification staying in HBM (``generate_correspondences_and_verify``): its keypoints are the
    reference's 5000 per frame in detection order; its putative and verified correspondences, compared as COORDINATE pairs (the index order differs by
    design, and with it the fp32 sums inside LightGlue: a match at the 0.1 filter may fall on the other side), are the golden chain's."""
    from gtsfm_amd.frontend.correspondence_generator.batched_det_desc_correspondence_generator import BatchedDetDescCorrespondenceGenerator
    from gtsfm_amd.frontend.verifier.ransac import Ransac

    gen = BatchedDetDescCorrespondenceGenerator(chain["lg"], chain["det"])
    width = chain["images"][0].width
    keypoints, putative, verified = gen.generate_correspondences_and_verify(None, chain["images"], PAIRS, chain["intrinsics"], Ransac(True, float(golden["threshold_px"])))
    assert len(keypoints) == 3 and list(putative) == PAIRS == list(verified)
    strays = [int(len(np.setxor1d(_pixel_key(k.coordinates, width), _pixel_key(golden[f"keypoints_{i}"], width)))) for i, k in enumerate(keypoints)]
    _record("batched_generator.stray_keypoints_per_frame", strays)
    assert strays == [0, 0, 0]
    put_diff, ver_diff = [], []
    for i, j in PAIRS:
        ki, kj = _pixel_key(golden[f"keypoints_{i}"], width), _pixel_key(golden[f"keypoints_{j}"], width)
        gi, gj = _pixel_key(keypoints[i].coordinates, width), _pixel_key(keypoints[j].coordinates, width)
        as_set = lambda idx, a, b: set(zip(a[idx[:, 0].astype(np.int64)].tolist(), b[idx[:, 1].astype(np.int64)].tolist()))  # noqa: E731
        assert putative[(i, j)].dtype == np.int64
        ref_put = as_set(golden[f"matches_{i}_{j}"], ki, kj)
        ref_ver = as_set(golden[f"v_corr_idxs_{i}_{j}"].reshape(-1, 2), ki, kj)
        got_put = as_set(putative[(i, j)], gi, gj)
        rot, direction, v_idx, ratio = verified[(i, j)]
        got_ver = as_set(np.asarray(v_idx).reshape(-1, 2), gi, gj)
        put_diff.append(len(ref_put ^ got_put))
        ver_diff.append(len(ref_ver ^ got_ver))
        assert got_ver <= got_put and (rot is not None) == bool(golden[f"has_pose_{i}_{j}"])
    _record("batched_generator.putative_coordinate_pairs_differing_per_pair", put_diff)
    _record("batched_generator.verified_coordinate_pairs_differing_per_pair", ver_diff)
    # putative matches: the golden chain's (the keypoint ORDER enters LightGlue's fp32 sums; no match of this fixture sits near the filter threshold).
    # verified matches: the verifier draws its samples by match INDEX, and the batched path lists a pair's matches in ITS keypoint order -- another
    # (equally valid) RANSAC run over the same putative set; it may keep a slightly different inlier set
    # -- observed on MI355X (profiles/r05_config5_barn_observed.json): it keeps exactly the same ones on this fixture; asserted as observed
    assert put_diff == [0, 0, 0]
    assert ver_diff == [0, 0, 0], ver_diff
