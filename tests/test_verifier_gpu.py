"""Verifier stage on the device against ``oracle/verifier_oracle.py``.

PARITY UNPINNED towards the reference (OpenCV's USAC is absent and not reproducible, see the oracle's header); what is
pinned here is HIP kernel == oracle: the same counter-based minimal samples, the same sequence of double operations, hence
identical winners, identical inlier masks (bit-exact index work) and E / R / t within 1e-9 (expected: identical)."""

import numpy as np
import pytest
import torch

from gtsfm_amd.utils import synthetic
from oracle import verifier_oracle as vo

pytestmark = pytest.mark.gpu

# (matches, outlier share, noise px, threshold px, extra keypoints)
SCENES = [
    (8, 0.0, 0.0, 0.5, 0),
    (40, 0.0, 0.0, 0.5, 7),
    (223, 0.3, 0.5, 2.0, 100),
    (300, 0.6, 0.5, 4.0, 50),
    (718, 0.5, 1.0, 4.0, 300),
    (2048, 0.7, 0.5, 4.0, 0),
]


def _batch(scenes, device):
    """Pack scenes the way the pipeline does: one keypoint table, offsets per pair, ragged match lists."""
    tables, off1, off2, idx, moff, intr = [], [], [], [], [0], []
    row = 0
    for s in scenes:
        off1.append(row)
        row += s["coordinates_i1"].shape[0]
        off2.append(row)
        row += s["coordinates_i2"].shape[0]
        tables += [s["coordinates_i1"], s["coordinates_i2"]]
        idx.append(s["match_indices"])
        moff.append(moff[-1] + s["match_indices"].shape[0])
        intr.append(list(s["intrinsics"]) * 2)
    kp = torch.from_numpy(np.concatenate(tables, 0)).to(device)
    mi = torch.from_numpy(np.ascontiguousarray(np.concatenate(idx, 0).astype(np.int32))).to(device)
    return kp, off1, off2, mi, moff, np.asarray(intr)


def _compare(out, p, lo, hi, ref):
    stats = out["stats"][p].cpu().numpy()
    mask = out["mask"][lo:hi].cpu().numpy().astype(bool)
    assert stats[1] == ref["hypotheses"]
    assert (int(stats[2]), int(stats[3])) == ref["winner"]
    np.testing.assert_array_equal(mask, ref["mask"])
    assert stats[0] == ref["mask"].sum()
    np.testing.assert_array_equal(stats[4:8], ref["cheirality"])
    scale = np.abs(ref["E"]).max()
    np.testing.assert_allclose(out["E"][p].cpu().numpy(), ref["E"], rtol=0, atol=1e-9 * scale)
    np.testing.assert_allclose(out["R"][p].cpu().numpy(), ref["R"], rtol=0, atol=1e-9)
    np.testing.assert_allclose(out["t"][p].cpu().numpy(), ref["t"], rtol=0, atol=1e-9)


def test_batch_of_scenes_equals_oracle_and_single_calls(gpu_device):
    from gtsfm_amd.runtime.verifier_engine import VerifierEngine

    engine = VerifierEngine(gpu_device)
    scenes = [synthetic.synthetic_two_view_matches(m, o, n, seed=10 + k, num_extra_keypoints=x) for k, (m, o, n, _, x) in enumerate(SCENES)]
    seeds = [3 + 17 * k for k in range(len(scenes))]
    for thr in (2.0,):
        kp, off1, off2, mi, moff, intr = _batch(scenes, gpu_device)
        out = engine.verify_batch(kp, off1, off2, mi, moff, intr, thr, seeds)
        for p, s in enumerate(scenes):
            ref = vo.verify(s["coordinates_i1"], s["coordinates_i2"], s["match_indices"], s["intrinsics"], s["intrinsics"], thr, seed=seeds[p])
            _compare(out, p, moff[p], moff[p + 1], ref)
            single = engine.verify_batch(*_batch([s], gpu_device), thr, [seeds[p]])
            for key in ("E", "R", "t", "stats"):
                assert torch.equal(single[key][0], out[key][p]), key  # batched == single, bit for bit
            assert torch.equal(single["mask"], out["mask"][moff[p] : moff[p + 1]])


def test_fundamental_mode_equals_oracle(gpu_device):
    """use_intrinsics_in_verification=False: seven-point samples on pixel coordinates, OpenCV's point-to-line residual,
    E = K2^T F K1 -- same bit-for-bit criteria as the essential mode."""
    from gtsfm_amd.runtime.verifier_engine import VerifierEngine

    engine = VerifierEngine(gpu_device)
    scenes = [synthetic.synthetic_two_view_matches(m, o, n, seed=40 + k, num_extra_keypoints=x, fx=700.0 + 30 * k) for k, (m, o, n, _, x) in enumerate(SCENES)]
    scenes.append(synthetic.synthetic_two_view_matches(7, seed=3))  # fewer than NUM_MATCHES_REQ_F_MATRIX
    seeds = [5 + 11 * k for k in range(len(scenes))]
    kp, off1, off2, mi, moff, intr = _batch(scenes, gpu_device)
    out = engine.verify_batch(kp, off1, off2, mi, moff, intr, 2.0, seeds, use_intrinsics=False)
    for p, s in enumerate(scenes):
        ref = vo.verify(s["coordinates_i1"], s["coordinates_i2"], s["match_indices"], s["intrinsics"], s["intrinsics"], 2.0, seed=seeds[p],
                        use_intrinsics_in_verification=False)
        if ref["R"] is None:
            assert out["stats"][p, 0].item() == 0 and torch.isnan(out["F"][p]).all()
            continue
        _compare(out, p, moff[p], moff[p + 1], ref)
        np.testing.assert_allclose(out["F"][p].cpu().numpy(), ref["F"], rtol=0, atol=1e-9 * np.abs(ref["F"]).max())


@pytest.mark.parametrize("m,outliers,noise,thr,extra", SCENES[2:5])
def test_recovers_the_planted_geometry(gpu_device, m, outliers, noise, thr, extra):
    from gtsfm_amd.runtime.verifier_engine import VerifierEngine

    s = synthetic.synthetic_two_view_matches(m, outliers, noise, seed=77, num_extra_keypoints=extra)
    out = VerifierEngine(gpu_device).verify_batch(*_batch([s], gpu_device), thr, [1])
    mask = out["mask"].cpu().numpy().astype(bool)
    assert (mask & s["is_inlier"]).sum() >= 0.9 * s["is_inlier"].sum()  # planted matches are kept
    assert (mask & ~s["is_inlier"]).sum() <= 0.05 * m  # random re-pointed matches are rejected (a few fall on epipolar lines)
    rot = out["R"][0].cpu().numpy()
    angle = np.degrees(np.arccos(np.clip((np.trace(rot.T @ s["i2Ri1"]) - 1) / 2, -1, 1)))
    assert angle < 2.0 and abs(np.linalg.det(rot) - 1) < 1e-9


def test_too_few_matches_and_empty_batch(gpu_device):
    from gtsfm_amd.runtime.verifier_engine import VerifierEngine

    engine = VerifierEngine(gpu_device)
    s = synthetic.synthetic_two_view_matches(5, seed=1)
    ok = synthetic.synthetic_two_view_matches(30, seed=2)
    kp, off1, off2, mi, moff, intr = _batch([s, ok], gpu_device)
    out = engine.verify_batch(kp, off1, off2, mi, moff, intr, 1.0)
    assert out["stats"][0, 0].item() == 0 and out["stats"][0, 1].item() == 0 and torch.isnan(out["E"][0]).all()
    assert out["mask"][:5].sum().item() == 0 and out["stats"][1, 0].item() == 30
    empty = engine.verify_batch(kp, [], [], torch.zeros((0, 2), dtype=torch.int32, device=gpu_device), [0], np.zeros((0, 8)), 1.0)
    assert empty["E"].shape == (0, 3, 3)


def _two_planes_scene(m_points, n_points):
    """``simulate_two_planes_scene`` of the reference's verifier tests (tests/frontend/verifier/test_verifier_base.py:229-295)
    without gtsam: points on two planes, cameras wTi1 = (Rx(pi/20), (0.1, 0, -20)), wTi2 = (Ry(pi/6), (1, -2, -20.4)), unit
    intrinsics."""
    rng = np.random.default_rng(15)

    def on_plane(coeffs, count):
        a, b, c, d = coeffs
        x, y = rng.uniform(-5, 7, count), rng.uniform(-10, 10, count)
        return np.stack([x, y, -(a * x + b * y + d) / c], 1)

    pts = np.vstack([on_plane((-10, -1, -20, 150), m_points), on_plane((15, -2, -35, 200), n_points)])
    w_r_1 = synthetic._rotation_about([1, 0, 0], np.pi / 20)
    w_r_2 = synthetic._rotation_about([0, 1, 0], np.pi / 6)
    t1, t2 = np.array([0.1, 0, -20]), np.array([1, -2, -20.4])
    c1 = (pts - t1) @ w_r_1
    c2 = (pts - t2) @ w_r_2
    rot = w_r_2.T @ w_r_1
    trans = w_r_2.T @ (t1 - t2)
    return c1[:, :2] / c1[:, 2:], c2[:, :2] / c2[:, 2:], rot, trans / np.linalg.norm(trans)


def test_plugin_contract_of_the_reference_verifier_suite(gpu_device):
    """The reference's own acceptance tests for every verifier (test_verifier_base.py:80-147), through the plugin."""
    import pickle

    from gtsfm_amd.common.calibration import PinholeIntrinsics
    from gtsfm_amd.common.keypoints import Keypoints
    from gtsfm_amd.frontend.verifier.ransac import Ransac

    uv1, uv2, rot, direction = _two_planes_scene(4, 4)
    matches = np.stack([np.arange(8), np.arange(8)], 1)
    for use_intrinsics in (True, False):  # TestRansacForEssentialMatrix, TestRansacForFundamentalMatrix (test_ransac.py:12-31)
        verifier = Ransac(use_intrinsics_in_verification=use_intrinsics, estimation_threshold_px=0.5)
        pickle.loads(pickle.dumps(verifier))  # test_pickleable
        r, u, verified, ratio = verifier.verify(Keypoints(uv1), Keypoints(uv2), matches, PinholeIntrinsics(), PinholeIntrinsics())  # two_plane_scene
        r, u = np.asarray(getattr(r, "matrix", lambda: r)()), np.asarray(getattr(u, "point3", lambda: u)())
        assert np.degrees(np.arccos(np.clip((np.trace(r.T @ rot) - 1) / 2, -1, 1))) < 2
        assert np.degrees(np.arccos(np.clip(u @ direction, -1, 1))) < 2
        np.testing.assert_array_equal(verified, matches)
        assert ratio == 1.0
    verifier = Ransac(use_intrinsics_in_verification=True, estimation_threshold_px=0.5)
    r, u, verified, ratio = verifier.verify(Keypoints(uv1), Keypoints(uv2), np.array([], dtype=np.int32), PinholeIntrinsics(), PinholeIntrinsics())
    assert r is None and u is None and verified.size == 0 and ratio == 0.0  # test_verify_empty_matches
    rng = np.random.default_rng(15)
    for _ in range(10):  # test_valid_verified_indices
        n1, n2 = int(rng.integers(6, 100)), int(rng.integers(6, 100))
        h1, w1, h2, w2 = (int(v) for v in rng.integers(100, 400, 4))
        k1 = Keypoints(rng.uniform([0, 0], [w1, h1], (n1, 2)).astype(np.float32))
        k2 = Keypoints(rng.uniform([0, 0], [w2, h2], (n2, 2)).astype(np.float32))
        count = int(rng.integers(1, min(n1, n2) + 1))
        idx = np.stack([rng.choice(n1, count, replace=False), rng.choice(n2, count, replace=False)], 1).astype(np.uint32)
        _, _, verified, _ = verifier.verify(k1, k2, idx, PinholeIntrinsics(min(h1, w1), h1 / 2, w1 / 2), PinholeIntrinsics(min(h2, w2), h2 / 2, w2 / 2))
        if verified.size:
            assert (verified[:, 0] < n1).all() and (verified[:, 1] < n2).all()
            assert set(map(tuple, verified.tolist())) <= set(map(tuple, idx.tolist()))


def test_compaction_equals_the_plugins_match_arrays(gpu_device):
    from gtsfm_amd.runtime.verifier_engine import VerifierEngine

    rng = np.random.default_rng(4)
    n0, n1 = [700, 0, 300, 1, 2048], [650, 10, 0, 1, 2048]
    blocks, rows, row = [], [], 0
    for a, b in zip(n0, n1):
        m0 = np.where(rng.random(a) < 0.4, rng.integers(0, max(b, 1), a), -1).astype(np.int32)
        blocks += [m0, rng.integers(-1, max(a, 1), b).astype(np.int32)]
        rows.append(row)
        row += a + b
    matches = torch.from_numpy(np.concatenate(blocks)).to(gpu_device)
    idx, off, count = VerifierEngine(gpu_device).compact_matches(matches, rows, n0)
    idx, count = idx.cpu().numpy(), count.cpu().numpy()
    for p, a in enumerate(n0):
        m0 = blocks[2 * p]
        valid = m0 > -1
        expect = np.stack([np.flatnonzero(valid), m0[valid]], -1)  # superglue_matcher.py:100-102
        assert count[p] == valid.sum() and off[p + 1] - off[p] == a
        np.testing.assert_array_equal(idx[off[p] : off[p] + count[p]], expect)


def test_pipeline_detect_match_verify_stays_on_the_device_and_equals_the_oracle(gpu_device):
    """detect -> match -> verify through the resident pipeline on the benchmark's overlapping views; every pair's verified
    set, pose and hypothesis count against the oracle run on the (K, 2) match arrays of the same pipeline."""
    from gtsfm_amd.runtime import matcher_engine as ME
    from gtsfm_amd.runtime.pipeline import FrontEndPipeline
    from gtsfm_amd.runtime.superpoint_engine import SuperPointEngine

    views = synthetic.synthetic_overlapping_views(4, 512, 512)
    det = SuperPointEngine(synthetic.synthetic_superpoint_state_dict(), gpu_device)
    eng = ME.LightGlueEngine(synthetic.synthetic_lightglue_state_dict(num_layers=3), gpu_device)
    pipe = FrontEndPipeline(det, eng, max_keypoints=1024, pair_chunk=2, num_streams=2)
    feats = pipe.detect(torch.from_numpy(views).to(gpu_device))
    pairs = [(0, 1), (0, 2), (1, 3), (2, 3), (0, 3)]
    res = pipe.match(feats, pairs, [(512, 512)] * 4)
    intr = np.array([[600.0 + 10 * i, 600.0 + 10 * i, 256.0, 256.0] for i in range(4)])
    ver = pipe.verify(feats, res, intr, threshold_px=1.0)
    got = pipe.verified_to_numpy(ver)
    putative = pipe.matches_to_numpy(res)
    xy, cnt = feats["xy"].cpu().numpy(), feats["count"].cpu().numpy()
    some = 0
    for i, j in pairs:
        np.testing.assert_array_equal(got[(i, j)]["putative"], putative[(i, j)])
        ref = vo.verify(xy[i, : cnt[i]], xy[j, : cnt[j]], putative[(i, j)], tuple(intr[i]), tuple(intr[j]), 1.0, seed=(i << 32) | j)
        np.testing.assert_array_equal(got[(i, j)]["v_corr_idxs"], ref["v_corr_idxs"])
        assert got[(i, j)]["hypotheses"] == ref["hypotheses"]
        if ref["R"] is None:
            assert got[(i, j)]["R"] is None and got[(i, j)]["inlier_ratio"] == 0.0
        else:
            some += 1
            np.testing.assert_allclose(got[(i, j)]["R"], ref["R"], atol=1e-9)
            np.testing.assert_allclose(got[(i, j)]["t"], ref["t"], atol=1e-9)
            assert got[(i, j)]["inlier_ratio"] == ref["inlier_ratio"]
    assert some >= 3


def test_generator_detect_match_verify_equals_per_pair_plugin_calls(gpu_device, tmp_path):
    """BatchedDetDescCorrespondenceGenerator.generate_correspondences_and_verify (detect -> match -> verify for all edges,
    device-resident) against what TwoViewEstimator would do edge by edge: the verifier plugin on the generator's keypoints and
    putative matches (same seed per edge). One image has no keypoints (fully masked), one calibration has skew (host fallback)."""
    from gtsfm_amd.common.calibration import PinholeIntrinsics
    from gtsfm_amd.common.image import Image
    from gtsfm_amd.frontend.correspondence_generator.batched_det_desc_correspondence_generator import BatchedDetDescCorrespondenceGenerator
    from gtsfm_amd.frontend.detector_descriptor.superpoint import SuperPointDetectorDescriptor
    from gtsfm_amd.frontend.matcher.lightglue_matcher import LightGlueMatcher
    from gtsfm_amd.frontend.verifier.ransac import Ransac

    torch.save(synthetic.synthetic_superpoint_state_dict(), str(tmp_path / "sp.pth"))
    torch.save(synthetic.synthetic_lightglue_state_dict(num_layers=3), str(tmp_path / "lg.pth"))
    views = synthetic.synthetic_overlapping_views(4, 256, 320, seed=9)
    images = [Image(value_array=v) for v in views] + [Image(value_array=views[0].copy(), mask=np.zeros((256, 320), dtype=np.uint8))]  # fully masked

    class Skewed(PinholeIntrinsics):
        def K(self):  # noqa: N802
            k = super().K()
            k[0, 1] = 0.5
            return k

        def calibrate(self, uv):
            uv = np.asarray(uv, dtype=np.float64).reshape(2)
            y = (uv[1] - self.v0) / self.fy
            return np.array([(uv[0] - self.u0 - 0.5 * y) / self.fx, y])

    cams = [PinholeIntrinsics(400.0 + 5 * i, 160.0, 128.0) for i in range(5)]
    cams[3] = Skewed(410.0, 160.0, 128.0)
    gen = BatchedDetDescCorrespondenceGenerator(
        LightGlueMatcher("superpoint", weights_path=tmp_path / "lg.pth"), SuperPointDetectorDescriptor(max_keypoints=600, weights_path=tmp_path / "sp.pth"),
        pair_batch=2)
    edges = [(0, 1), (0, 2), (1, 2), (2, 3), (1, 3), (0, 4)]
    kps, putative, verified = gen.generate_correspondences_and_verify(None, images, edges, cams, Ransac(True, 1.0))
    assert list(verified) == edges and len(kps[4]) == 0 and verified[(0, 4)][0] is None and verified[(0, 4)][2].size == 0
    models = 0
    for i, j in edges:
        ref = Ransac(True, 1.0, seed=(i << 32) | j).verify(kps[i], kps[j], putative[(i, j)], cams[i], cams[j])
        got = verified[(i, j)]
        np.testing.assert_array_equal(got[2], ref[2])
        assert got[3] == ref[3] and (got[0] is None) == (ref[0] is None)
        if ref[0] is not None:
            models += 1
            np.testing.assert_array_equal(np.asarray(got[0]), np.asarray(ref[0]))
            np.testing.assert_array_equal(np.asarray(got[1]), np.asarray(ref[1]))
            assert set(map(tuple, got[2].tolist())) <= set(map(tuple, putative[(i, j)].tolist()))
    assert models >= 3
