"""-m gpu: BASELINE config 1 LITERALLY -- all 12 frames of the reference's own fixture ``tests/data/set1_lund_door`` at the Olsson
loader's resolution (``max_resolution: 760``, gtsfm/configs/loader/olsson.yaml:5 -> 760 x 1135), GTSfM's 5000-keypoint cap
(gtsfm/configs/deep_front_end.yaml:29), all 66 exhaustive pairs -- through the plugin classes and their cachers, and through the
batched correspondence generator, against golden vectors WRITTEN BY THE REFERENCE'S OWN MODEL FILES
(``oracle/validate_against_reference.py::check_lund_door_config1``: reference SuperPoint + the wrapper's ``get_top_k`` restated,
reference SuperGlue with GTSfM's 20 Sinkhorn iterations + the wrapper's output marshalling; the reference's own plugin classes, run live by
``oracle/validate_wrappers_against_reference.py``, return exactly these arrays for all 12 frames and 66 pairs). The reduced gray frames travel in the
fixture (/root/reference does not exist on the GPU box); the loader's ``cv.INTER_CUBIC`` reduction in front of them is the one step
restated without a pin (cv2 absent).

Mirrors ``tests/frontend/detector_descriptor/test_superpoint.py:12-21`` (the plugin on the fixture's images) and the per-image /
per-pair calls of ``gtsfm/frontend/correspondence_generator/det_desc_correspondence_generator.py:57-87``."""

import numpy as np
import pytest
import torch

from gtsfm_amd.common.image import Image
from gtsfm_amd.common.keypoints import Keypoints
from gtsfm_amd.utils import synthetic

from conftest import GOLDEN

pytestmark = pytest.mark.gpu

TOL = 1e-4              # scores / descriptors / match scores (BASELINE.json north_star)
MATCH_THRESHOLD = 0.2   # superglue.py default match_threshold (GTSfM does not override it)
NUM_IMAGES = 12

# What the run SAW, not only that it passed: every count a tolerance below could hide is recorded here and written to
# gpurun_out/config1_observed.json (copied to profiles/r05_config1_observed.json), and the asserts hold the run to the values observed on
# MI355X (round 5, profiles/r05_config1_observed.json) -- zero stray keypoints in all 12 frames, 64 of 66 match arrays array_equal and exactly
# two threshold-borderline matches (one each in two pairs, of 8454), zero matches0 disagreements on the fully compared pairs. ``Keypoints.__eq__`` = ``np.array_equal`` is the reference's own bar
# (tests/repro_tests/frontend/detector_descriptor/reproducibility_base.py:25-36).
OBSERVED = {}
EXPECTED_STRAY_KEYPOINTS_PER_FRAME = 0      # plugin path and batched generator, all 12 frames
# plugin path, all 66 pairs: 64 of the (K, 2) arrays are array_equal to the reference's; pairs (3, 9) and (4, 7) differ in ONE match each (of 8454),
# and for both the test proves the cause: the match's score sits within 2e-4 of the 0.2 threshold in the reference's own output or in the HIP
# path's (scores agree to 6.9e-6; a contract of 1e-4 on scores cannot decide a comparison against a constant closer than that)
EXPECTED_BORDERLINE_MATCHES = 2
EXPECTED_PAIRS_WITH_A_BORDERLINE_MATCH = [[3, 9, 1], [4, 7, 1]]
EXPECTED_MATCHES0_DISAGREEMENTS = 0         # plugin path, the 11 pairs whose full matches0 / score vectors are compared
# generators that feed the matcher the keypoints in THEIR OWN order (detection order / this process's get_top_k order): the order enters the
# fp32 sums of attention and Sinkhorn, scores move in the sixth digit, a match AT the 0.2 threshold may fall on the other side. The numbers
# below are the symmetric differences observed on MI355X over all edges; the run must reproduce them exactly.
EXPECTED_BATCHED_GENERATOR_DIFFERING = 1      # of 8454 matches over 66 pairs: pair (4, 7), the same borderline match as above
EXPECTED_PER_PAIR_GENERATOR_DIFFERING = 0     # of 2106 matches over the first 22 pairs


def _exact_fp32() -> bool:
    """The pinned counts are statements about the default arithmetic; under the opt-in bf16x3 switches the old bounds apply (and the
    observations go to a file of their own)."""
    import os

    return not (os.environ.get("GTSFM_ATTENTION_MATH") or os.environ.get("GTSFM_GEMM_MATH"))


def _record(key, value):
    import json
    import os

    from conftest import REPO

    OBSERVED[key] = value
    out = REPO / "gpurun_out"
    out.mkdir(exist_ok=True)
    tag = "" if _exact_fp32() else "_" + (os.environ.get("GTSFM_ATTENTION_MATH") or os.environ.get("GTSFM_GEMM_MATH") or "switched")
    (out / f"config1_observed{tag}.json").write_text(json.dumps(OBSERVED, indent=1, sort_keys=True))


@pytest.fixture(scope="module")
def golden():
    path = GOLDEN / "lund_door_config1.npz"
    if not path.exists():
        pytest.fail("tests/golden/lund_door_config1.npz is missing: run oracle/validate_against_reference.py --only-config1 --write in the build container")
    import io

    from PIL import Image as PILImage

    g = dict(np.load(path))
    g["gray"] = np.stack([np.asarray(PILImage.open(io.BytesIO(g[f"gray_png_{i}"].tobytes()))) for i in range(NUM_IMAGES)])  # lossless PNG streams
    assert g["gray"].dtype == np.uint8 and g["gray"].shape == (NUM_IMAGES, int(g["height"]), int(g["width"]))
    return g


@pytest.fixture(scope="module")
def images(golden):
    # the loader hands RGB uint8 images over; the frames are stored gray (R = G = B: the fixed-point gray conversion is the identity)
    return [Image(value_array=np.repeat(golden["gray"][i][:, :, None], 3, axis=2), file_name=str(golden["names"][i])) for i in range(NUM_IMAGES)]


@pytest.fixture(scope="module")
def plugins(tmp_path_factory, gpu_device):
    from gtsfm_amd.frontend.cacher.detector_descriptor_cacher import DetectorDescriptorCacher
    from gtsfm_amd.frontend.cacher.matcher_cacher import MatcherCacher
    from gtsfm_amd.frontend.detector_descriptor.superpoint import SuperPointDetectorDescriptor
    from gtsfm_amd.frontend.matcher.superglue_matcher import SuperGlueMatcher

    tmp = tmp_path_factory.mktemp("config1")
    torch.save(synthetic.synthetic_superpoint_state_dict(), str(tmp / "sp.pth"))
    torch.save(synthetic.synthetic_superglue_state_dict(), str(tmp / "sg.pth"))
    det = SuperPointDetectorDescriptor(max_keypoints=5000, weights_path=tmp / "sp.pth")
    sg = SuperGlueMatcher(weights_path=tmp / "sg.pth")
    return {"det": det, "sg": sg, "det_cacher": DetectorDescriptorCacher(det, cache_root=tmp / "cache"), "sg_cacher": MatcherCacher(sg, cache_root=tmp / "cache"),
            "calls": {"det": 0, "sg": 0}}


def _pixel_key(xy, width):
    xy = np.asarray(xy)
    return xy[:, 1].astype(np.int64) * width + xy[:, 0].astype(np.int64)


@pytest.fixture(scope="module")
def detections(golden, images, plugins):
    """Every frame once through DetectorDescriptorCacher(SuperPointDetectorDescriptor)."""
    det = plugins["det"]
    real = det.detect_and_describe

    def counted(image):
        plugins["calls"]["det"] += 1
        return real(image)

    det.detect_and_describe = counted
    out = [plugins["det_cacher"].detect_and_describe(im) for im in images]
    assert plugins["calls"]["det"] == NUM_IMAGES
    return out


def test_detections_equal_the_reference_on_all_12_frames(golden, images, detections):
    """Keypoints identical to the reference's 5000 in every frame -- as a set: ``np.argpartition``'s order is implementation-defined, and
    scores that differ in the last bits order differently; ZERO stray keypoints, recorded per frame --, responses and descriptors within 1e-4."""
    width = images[0].width
    strays, max_dscore, max_ddesc = [], 0.0, 0.0
    for i, (kps, desc) in enumerate(detections):
        ref_xy, ref_sc, ref_head = golden[f"keypoints_{i}"].astype(np.float32), golden[f"scores_{i}"], golden[f"descriptors_head_{i}"]
        assert isinstance(kps, Keypoints) and len(kps) == 5000 == desc.shape[0] and desc.shape[1] == 256 and kps.scales is None
        assert kps.coordinates.dtype == np.float32 and desc.dtype == np.float32
        got_key, ref_key = _pixel_key(kps.coordinates, width), _pixel_key(ref_xy, width)
        assert len(np.unique(got_key)) == 5000
        stray = np.setxor1d(got_key, ref_key)
        strays.append(int(len(stray)))
        common, gi, ri = np.intersect1d(got_key, ref_key, return_indices=True)
        head = np.flatnonzero(ri < len(ref_head))  # the stored descriptor rows: the first 64 of the reference's order
        max_dscore = max(max_dscore, float(np.abs(kps.responses[gi] - ref_sc[ri]).max()))
        max_ddesc = max(max_ddesc, float(np.abs(desc[gi[head]] - ref_head[ri[head]]).max()))
    _record("plugin_detect.stray_keypoints_per_frame", strays)
    _record("plugin_detect.max_abs_dresponse", max_dscore)
    _record("plugin_detect.max_abs_ddescriptor_first64", max_ddesc)
    assert strays == [EXPECTED_STRAY_KEYPOINTS_PER_FRAME] * NUM_IMAGES, strays  # the reference's 5000 keypoints of every frame, as a set (SuperPoint has no opt-in arithmetic)
    assert max_dscore < TOL and max_ddesc < TOL, (max_dscore, max_ddesc)


def _in_reference_order(golden, images, plugins, detections, i):
    """Image i's plugin output rearranged into the order the REFERENCE handed to its matcher (``sel_i``: its argpartition order)."""
    width = images[0].width
    kps, desc = detections[i]
    ref_xy = golden[f"keypoints_{i}"].astype(np.float32)
    got_key, ref_key = _pixel_key(kps.coordinates, width), _pixel_key(ref_xy, width)
    order = np.argsort(got_key)
    pos = np.searchsorted(got_key[order], ref_key)
    pos = np.clip(pos, 0, len(order) - 1)
    rows = order[pos]
    if not np.array_equal(got_key[rows], ref_key):  # a top-k boundary flip (see above): take the reference's rows from the full detection
        xy, sc, fetch = plugins["det"]._model.detect_lazy(np.ascontiguousarray(golden["gray"][i]))
        sel = golden[f"sel_{i}"].astype(np.int64)
        return Keypoints(xy[sel], scales=None, responses=sc[sel]), fetch(sel)
    return Keypoints(kps.coordinates[rows], scales=None, responses=kps.responses[rows]), np.ascontiguousarray(desc[rows])


def test_all_66_pairs_through_the_matcher_plugin_equal_the_reference(golden, images, plugins, detections):
    """``MatcherCacher(SuperGlueMatcher).match`` on every exhaustive pair, keypoints in the order the reference used: the (K, 2) uint32
    arrays are the reference's. A match whose score sits within 2e-4 of the 0.2 threshold may fall on either side (scores agree to
    1e-4, not to the bit); the match scores of every sixth pair are compared in full."""
    sg = plugins["sg"]
    real = sg.match

    def counted(*a, **k):
        plugins["calls"]["sg"] += 1
        return real(*a, **k)

    sg.match = counted
    feats = [_in_reference_order(golden, images, plugins, detections, i) for i in range(NUM_IMAGES)]
    shape = (images[0].height, images[0].width, 3)
    pairs = [(i, j) for i in range(NUM_IMAGES) for j in range(i + 1, NUM_IMAGES)]
    assert int(golden["num_pairs"]) == len(pairs) == 66
    total, borderline, results = 0, 0, {}
    differing_pairs, m0_disagreements, off_scores, max_dms = [], 0, 0, 0.0
    for q, (i, j) in enumerate(pairs):
        got = plugins["sg_cacher"].match(feats[i][0], feats[j][0], feats[i][1], feats[j][1], im_shape_i1=shape, im_shape_i2=shape)
        results[(i, j)] = got
        ref = golden[f"match_indices_{i}_{j}"].astype(np.uint32)
        ref_scores = golden[f"matching_scores0_{i}_{j}"]
        assert got.dtype == np.uint32 and got.ndim == 2 and got.shape[1] == 2
        total += len(ref)
        if not np.array_equal(got, ref):
            diff = set(map(tuple, got.tolist())) ^ set(map(tuple, ref.tolist()))
            res = sg._model.match_pair(feats[i][0].coordinates, feats[i][0].responses, feats[i][1], feats[j][0].coordinates, feats[j][0].responses, feats[j][1],
                                       shape[:2], shape[:2], sinkhorn_iterations=20)
            for a, _ in diff:
                assert abs(float(ref_scores[a]) - MATCH_THRESHOLD) < 2e-4 or abs(float(res["matching_scores0"][a]) - MATCH_THRESHOLD) < 2e-4, ((i, j), a)
            borderline += len(diff)
            differing_pairs.append([i, j, len(diff)])
        if q % 6 == 0:
            res = sg._model.match_pair(feats[i][0].coordinates, feats[i][0].responses, feats[i][1], feats[j][0].coordinates, feats[j][0].responses, feats[j][1],
                                       shape[:2], shape[:2], sinkhorn_iterations=20)
            ref_m0 = golden[f"matches0_{i}_{j}"].astype(np.int64)
            same = res["matches0"] == ref_m0
            m0_disagreements += int((~same).sum())
            assert same.mean() > 0.999
            matched = same & (ref_m0 > -1)
            max_dms = max(max_dms, float(np.abs(res["matching_scores0"][matched] - ref_scores[matched]).max()))
            np.testing.assert_allclose(res["matching_scores0"][matched], ref_scores[matched], rtol=0, atol=TOL)
            # unmatched keypoints carry exp(max) where they are mutual nearest neighbours and 0 where not (superglue.py:270-272): a
            # score may differ only where that flag flipped between two negligible candidates, never near the threshold
            off = np.abs(res["matching_scores0"] - ref_scores) > TOL
            off_scores += int(off.sum())
            assert off.sum() <= 5 and np.all(np.maximum(res["matching_scores0"][off], ref_scores[off]) < 0.5 * MATCH_THRESHOLD), ((i, j), int(off.sum()))
    _record("plugin_match.reference_matches_over_66_pairs", int(total))
    _record("plugin_match.borderline_matches_differing", int(borderline))
    _record("plugin_match.pairs_with_a_differing_match_array", differing_pairs)
    _record("plugin_match.matches0_disagreements_over_11_full_pairs", int(m0_disagreements))
    _record("plugin_match.unmatched_score_flag_flips_over_11_full_pairs", int(off_scores))
    _record("plugin_match.max_abs_dmatching_score", max_dms)
    assert plugins["calls"]["sg"] == 66 and total > 3000
    if _exact_fp32():
        assert borderline == EXPECTED_BORDERLINE_MATCHES, f"{borderline} threshold-borderline matches differ over {total}: {differing_pairs}"
        assert differing_pairs == EXPECTED_PAIRS_WITH_A_BORDERLINE_MATCH, differing_pairs
        assert m0_disagreements == EXPECTED_MATCHES0_DISAGREEMENTS, m0_disagreements
    else:
        assert borderline <= 3, f"{borderline} threshold-borderline matches differ over {total}"
    # second pass: cache hits only (matcher_cacher.py:46-126, detector_descriptor_cacher.py:48-69) -- neither plugin runs again
    for (i, j) in pairs[::5]:
        again = plugins["sg_cacher"].match(feats[i][0], feats[j][0], feats[i][1], feats[j][1], im_shape_i1=shape, im_shape_i2=shape)
        np.testing.assert_array_equal(again, results[(i, j)])
    for im, (kps, desc) in zip(images, detections):
        kps2, desc2 = plugins["det_cacher"].detect_and_describe(im)
        assert kps2 == kps and np.array_equal(desc2, desc)
    assert plugins["calls"] == {"det": NUM_IMAGES, "sg": 66}


def test_batched_correspondence_generator_on_config1(golden, images, plugins):
    """``BatchedDetDescCorrespondenceGenerator.generate_correspondences`` over the same 12 frames / 66 pairs (features resident in HBM,
    ragged multi-pair launches): its keypoints are the reference's 5000 per frame in detection order, and its matches -- compared as
    coordinate pairs, the index order differs by design -- are the reference's up to matches at the 0.2 threshold (the keypoint ORDER
    enters the fp32 sums of attention and Sinkhorn: scores move in the sixth digit)."""
    from gtsfm_amd.frontend.correspondence_generator.batched_det_desc_correspondence_generator import BatchedDetDescCorrespondenceGenerator

    gen = BatchedDetDescCorrespondenceGenerator(plugins["sg"], plugins["det"])
    pairs = [(i, j) for i in range(NUM_IMAGES) for j in range(i + 1, NUM_IMAGES)]
    keypoints, putative = gen.generate_correspondences(None, images, pairs)
    width = images[0].width
    assert len(keypoints) == NUM_IMAGES and sorted(putative) == pairs
    strays = []
    for i, kps in enumerate(keypoints):
        stray = np.setxor1d(_pixel_key(kps.coordinates, width), _pixel_key(golden[f"keypoints_{i}"], width))
        strays.append(int(len(stray)))
        assert len(kps) == 5000
    _record("batched_generator.stray_keypoints_per_frame", strays)
    assert strays == [EXPECTED_STRAY_KEYPOINTS_PER_FRAME] * NUM_IMAGES, strays
    total = differing = 0
    per_pair_diff = []
    for (i, j) in pairs:
        got = putative[(i, j)]
        assert got.dtype == np.uint32
        ref = golden[f"match_indices_{i}_{j}"].astype(np.int64)
        ki, kj = _pixel_key(golden[f"keypoints_{i}"], width), _pixel_key(golden[f"keypoints_{j}"], width)
        ref_set = set(zip(ki[ref[:, 0]].tolist(), kj[ref[:, 1]].tolist()))
        gi, gj = _pixel_key(keypoints[i].coordinates, width), _pixel_key(keypoints[j].coordinates, width)
        got_set = set(zip(gi[got[:, 0].astype(np.int64)].tolist(), gj[got[:, 1].astype(np.int64)].tolist()))
        total += len(ref_set)
        differing += len(ref_set ^ got_set)
        if ref_set ^ got_set:
            per_pair_diff.append([i, j, len(ref_set ^ got_set)])
        assert len(ref_set ^ got_set) <= max(2, 0.01 * len(ref_set)), ((i, j), len(ref_set ^ got_set), len(ref_set))
    _record("batched_generator.reference_matches_over_66_pairs", int(total))
    _record("batched_generator.differing_matches_own_keypoint_order", int(differing))
    _record("batched_generator.pairs_with_differences", per_pair_diff)
    if EXPECTED_BATCHED_GENERATOR_DIFFERING is None or not _exact_fp32():
        assert differing <= 0.002 * total + 2, (differing, total)
    else:
        assert differing == EXPECTED_BATCHED_GENERATOR_DIFFERING, (differing, total, per_pair_diff)


def test_per_pair_generator_on_config1(golden, images, plugins, detections, tmp_path_factory):
    """The reference's own flow, ``DetDescCorrespondenceGenerator(matcher=MatcherCacher(SuperGlueMatcher), detector_descriptor=
    DetectorDescriptorCacher(SuperPointDetectorDescriptor)).generate_correspondences(client, images, visibility_graph)``
    (gtsfm/configs/deep_front_end.yaml:22-35, det_desc_correspondence_generator.py:57-87), without a scheduler: keypoints in the plugin's own
    ``get_top_k`` order, matches -- as coordinate pairs -- the reference's, on the first 22 edges."""
    from gtsfm_amd.frontend.cacher.matcher_cacher import MatcherCacher
    from gtsfm_amd.frontend.correspondence_generator.det_desc_correspondence_generator import DetDescCorrespondenceGenerator

    # a matcher cache of its own: the reference's key scheme hashes the FIRST 10 rows of each image's features (matcher_cacher.py:24,46-80), and the
    # test above matched the same frames with their 5000 rows in the reference's order -- equal in the first rows, permuted further down for some
    # frames (np.argpartition on scores that differ in the last bits): a shared directory would serve those entries for these calls
    fresh = MatcherCacher(plugins["sg"], cache_root=tmp_path_factory.mktemp("config1_generator_cache"))
    gen = DetDescCorrespondenceGenerator(matcher=fresh, detector_descriptor=plugins["det_cacher"])
    pairs = [(i, j) for i in range(NUM_IMAGES) for j in range(i + 1, NUM_IMAGES)][:22]
    keypoints, putative = gen.generate_correspondences(None, images, pairs)
    width = images[0].width
    assert len(keypoints) == NUM_IMAGES and all(keypoints[i] == detections[i][0] for i in range(NUM_IMAGES))  # served from the detector cache
    total = differing = 0
    per_pair = []
    for (i, j) in pairs:
        got = putative[(i, j)].astype(np.int64)
        ref = golden[f"match_indices_{i}_{j}"].astype(np.int64)
        ki, kj = _pixel_key(golden[f"keypoints_{i}"], width), _pixel_key(golden[f"keypoints_{j}"], width)
        ref_set = set(zip(ki[ref[:, 0]].tolist(), kj[ref[:, 1]].tolist()))
        gi, gj = _pixel_key(keypoints[i].coordinates, width), _pixel_key(keypoints[j].coordinates, width)
        got_set = set(zip(gi[got[:, 0]].tolist(), gj[got[:, 1]].tolist()))
        total += len(ref_set)
        differing += len(ref_set ^ got_set)
        per_pair.append(((i, j), len(ref_set), len(ref_set ^ got_set)))
    _record("per_pair_generator.reference_matches_over_22_pairs", int(total))
    _record("per_pair_generator.differing_matches_own_keypoint_order", int(differing))
    _record("per_pair_generator.pairs_with_differences", [[p[0][0], p[0][1], p[2]] for p in per_pair if p[2]])
    assert total > 1000
    if EXPECTED_PER_PAIR_GENERATOR_DIFFERING is None or not _exact_fp32():
        assert differing <= 0.002 * total + 2, (differing, total, [p for p in per_pair if p[2]])
    else:
        assert differing == EXPECTED_PER_PAIR_GENERATOR_DIFFERING, (differing, total, [p for p in per_pair if p[2]])
