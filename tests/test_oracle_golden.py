"""CPU: the oracle restatements reproduce the golden vectors generated from the reference's own model files
(``oracle/validate_against_reference.py --write``, which also asserts bit-exactness against the reference)."""

import numpy as np
import pytest
import torch

from gtsfm_amd.utils import synthetic
from oracle import superglue_oracle, superpoint_oracle
from tests.conftest import GOLDEN

SP_FILES = sorted(GOLDEN.glob("superpoint_*.npz"))
SG_FILES = sorted(GOLDEN.glob("superglue_*.npz"))


def test_golden_fixtures_present():
    assert len(SP_FILES) >= 3 and len(SG_FILES) >= 3


@pytest.mark.parametrize("path", SP_FILES, ids=lambda p: p.stem)
def test_superpoint_oracle_matches_golden(path):
    g = np.load(path)
    sd = synthetic.synthetic_superpoint_state_dict()
    gray = synthetic.synthetic_gray_image(int(g["height"]), int(g["width"]), int(g["seed"]))
    with torch.no_grad():
        out = superpoint_oracle.superpoint_forward(sd, superpoint_oracle.gray_u8_to_tensor(gray), return_intermediates=True)
    # keypoints / indices: exact. Floating point: the fixtures were produced by the same ATen kernels, but CPU ISA
    # dispatch may differ between machines, so allow fp32 round-off on values.
    np.testing.assert_array_equal(out["keypoints"].numpy().astype(np.int32), g["keypoints"])
    np.testing.assert_allclose(out["scores"].numpy(), g["scores"], rtol=0, atol=1e-6)
    np.testing.assert_allclose(out["descriptors"].numpy().T, g["descriptors"], rtol=0, atol=1e-6)
    np.testing.assert_allclose(out["dense_scores"][0, ::7, ::5].numpy(), g["dense_scores_sample"], rtol=0, atol=1e-6)


def test_superpoint_oracle_fp64_agrees():
    """The float64 variant (tie-breaker for fp32 disagreements) finds the same keypoints on a golden case."""
    g = np.load(SP_FILES[0])
    sd = synthetic.synthetic_superpoint_state_dict()
    gray = synthetic.synthetic_gray_image(int(g["height"]), int(g["width"]), int(g["seed"]))
    with torch.no_grad():
        out = superpoint_oracle.superpoint_forward(sd, superpoint_oracle.gray_u8_to_tensor(gray, torch.float64))
    np.testing.assert_array_equal(out["keypoints"].numpy().astype(np.int32), g["keypoints"])
    np.testing.assert_allclose(out["descriptors"].numpy().T, g["descriptors"], rtol=0, atol=1e-5)


@pytest.mark.parametrize("path", SG_FILES, ids=lambda p: p.stem)
def test_superglue_oracle_matches_golden(path):
    g = np.load(path)
    sd = synthetic.synthetic_superglue_state_dict()
    shp0, shp1 = tuple(int(v) for v in g["shape0"]), tuple(int(v) for v in g["shape1"])
    k0, s0, d0, k1, s1, d1, _ = synthetic.synthetic_pair_features(int(g["n0"]), int(g["n1"]), shp0, shp1, seed=int(g["seed"]))
    T = torch.from_numpy
    with torch.no_grad():
        out = superglue_oracle.superglue_forward(
            sd, T(k0)[None], T(k1)[None], T(s0)[None], T(s1)[None], T(d0).T[None].contiguous(), T(d1).T[None].contiguous(),
            shp0, shp1, sinkhorn_iterations=int(g["iters"]), return_intermediates=True,
        )
    np.testing.assert_array_equal(out["matches0"][0].numpy(), g["matches0"])
    np.testing.assert_array_equal(out["matches1"][0].numpy(), g["matches1"])
    np.testing.assert_allclose(out["matching_scores0"][0].numpy(), g["matching_scores0"], rtol=0, atol=2e-5)
    np.testing.assert_allclose(out["ot"][0, ::3, ::3].numpy(), g["ot_sample"], rtol=0, atol=2e-4)


def test_superglue_wrapper_marshalling_dtype():
    """gtsfm/frontend/matcher/superglue_matcher.py:104-113: (K,2) uint32, ordered by image-1 keypoint index."""
    sd = synthetic.synthetic_superglue_state_dict(num_layers=2)
    k0, s0, d0, k1, s1, d1, _ = synthetic.synthetic_pair_features(40, 30, (100, 120), (100, 120), seed=1)
    m = superglue_oracle.match(sd, k0, k1, s0, s1, d0, d1, (100, 120, 3), (100, 120, 3))
    assert m.dtype == np.uint32 and m.ndim == 2 and m.shape[1] == 2
    assert np.all(np.diff(m[:, 0].astype(np.int64)) > 0)
    assert len(set(m[:, 1].tolist())) == m.shape[0]


def test_superglue_empty_input_early_out():
    """superglue.py:233-240."""
    sd = synthetic.synthetic_superglue_state_dict(num_layers=2)
    out = superglue_oracle.superglue_forward(
        sd, torch.zeros((1, 0, 2)), torch.zeros((1, 3, 2)), torch.zeros((1, 0)), torch.zeros((1, 3)),
        torch.zeros((1, 256, 0)), torch.zeros((1, 256, 3)), (8, 8), (8, 8),
    )
    assert out["matches0"].shape == (1, 0) and out["matches0"].dtype == torch.int
    assert torch.equal(out["matches1"], torch.full((1, 3), -1, dtype=torch.int))


def test_real_image_fixture_lund_door():
    """BASELINE config 1 (plumbing) fixture: two frames of the reference's set1_lund_door, reduced to 568x380 gray, with the
    reference SuperPoint / SuperGlue outputs (synthetic weights)."""
    g = np.load(GOLDEN / "lund_door_pair.npz")
    sd = synthetic.synthetic_superpoint_state_dict()
    assert g["gray0"].dtype == np.uint8 and g["gray0"].shape == (568, 380)
    with torch.no_grad():
        out = superpoint_oracle.superpoint_forward(sd, superpoint_oracle.gray_u8_to_tensor(g["gray0"]))
    np.testing.assert_array_equal(out["keypoints"].numpy().astype(np.int32), g["keypoints0"])
    np.testing.assert_allclose(out["scores"].numpy(), g["scores0"], rtol=0, atol=1e-6)
    np.testing.assert_allclose(out["descriptors"].numpy().T[:256], g["descriptors0_head"], rtol=0, atol=1e-6)


def test_superpoint_oracle_matches_golden_config2_shape():
    """BASELINE config-2 shape (480x640) vs the fixture generated from the reference (full keypoint list)."""
    g = np.load(GOLDEN / "config2_superpoint_480x640_s4.npz")
    sd = synthetic.synthetic_superpoint_state_dict()
    gray = synthetic.synthetic_gray_image(480, 640, 4)
    with torch.no_grad():
        out = superpoint_oracle.superpoint_forward(sd, superpoint_oracle.gray_u8_to_tensor(gray))
    np.testing.assert_array_equal(out["keypoints"].numpy().astype(np.int32), g["keypoints"])
    np.testing.assert_allclose(out["scores"].numpy(), g["scores"], rtol=0, atol=1e-6)
    np.testing.assert_allclose(out["descriptors"].numpy().T[:256], g["descriptors_head"], rtol=0, atol=1e-6)


def test_bench_work_formulas_match_survey():
    """bench.py's algorithmic-work formulas reproduce SURVEY.md section 8(d): 177.85 / 52.10 GFLOP per image,
    254.8 GFLOP per SuperGlue pair and 229.8 GFLOP per full-depth LightGlue pair at N = 2048."""
    import bench

    assert abs(bench.superpoint_flops(1024, 1024) / 1e9 - 177.85) < 0.01
    assert abs(bench.superpoint_flops(480, 640) / 1e9 - 52.10) < 0.01
    assert abs(bench.matcher_flops("superglue", 2048, 18, 100) / 1e9 - 254.8) < 0.1
    assert abs(bench.matcher_flops("lightglue", 2048, 9.0, 0) / 1e9 - 229.8) < 0.1
    assert abs(bench.matcher_flops("superglue", 5000, 18, 100) / 1e9 - 1173.8) < 0.5


def test_config1_fixture_oracle_reproduces_the_reference_on_a_frame():
    """BASELINE config 1 literally (12 Lund-door frames at 1135x760, 5000-keypoint cap, 66 SuperGlue pairs: written by the reference's
    own model files, oracle/validate_against_reference.py::check_lund_door_config1): the SuperPoint oracle + the restated wrapper
    top-k reproduce frame 11's stored keypoints in the reference's order; the fixture is complete."""
    import io

    from PIL import Image as PILImage

    g = np.load(GOLDEN / "lund_door_config1.npz")
    assert int(g["num_pairs"]) == 66 and int(g["max_keypoints"]) == 5000 and (int(g["height"]), int(g["width"])) == (1135, 760)
    assert all(f"match_indices_{i}_{j}" in g.files for i in range(12) for j in range(i + 1, 12))
    gray = np.asarray(PILImage.open(io.BytesIO(g["gray_png_11"].tobytes())))
    assert gray.dtype == np.uint8 and gray.shape == (1135, 760)
    sd = synthetic.synthetic_superpoint_state_dict()
    with torch.no_grad():
        out = superpoint_oracle.superpoint_forward(sd, superpoint_oracle.gray_u8_to_tensor(gray))
    assert out["keypoints"].shape[0] == int(g["k_raw_11"])
    sel = g["sel_11"].astype(np.int64)
    np.testing.assert_array_equal(out["keypoints"].numpy()[sel].astype(np.int16), g["keypoints_11"])
    np.testing.assert_allclose(out["scores"].numpy()[sel], g["scores_11"], rtol=0, atol=1e-6)
    np.testing.assert_allclose(out["descriptors"].numpy().T[sel[:64]], g["descriptors_head_11"], rtol=0, atol=1e-6)
    # the wrapper's selection (gtsfm/common/keypoints.py:89-110): the 5000 strongest responses
    assert set(np.argpartition(-out["scores"].numpy(), 5000)[:5000].tolist()) == set(sel.tolist())


def test_config5_barn_fixture_oracle_chain_on_a_pair():
    """The front-end slice of BASELINE config 5 (oracle/make_barn_config5_golden.py: the reference's three Barn fixture frames at the COLMAP loader's
    760 x 1351): the SuperPoint oracle + the restated wrapper top-k reproduce frame 2's stored keypoints, the loader's intrinsics are the EXIF formula
    rescaled, and the verifier oracle reproduces the stored verified indices of pair (0, 2) from the stored matches (the LightGlue leg, 8 s per pair on
    the CPU, is re-run by the generator script and under -m gpu)."""
    import io

    from PIL import Image as PILImage

    from oracle import verifier_oracle

    g = np.load(GOLDEN / "barn_config5_frontend.npz")
    assert (int(g["height"]), int(g["width"])) == (760, 1351) and int(g["max_resolution"]) == 760 and list(g["names"]) == ["000001.jpg", "000002.jpg", "000003.jpg"]
    gray = np.asarray(PILImage.open(io.BytesIO(g["gray_png_2"].tobytes())))
    assert gray.dtype == np.uint8 and gray.shape == (760, 1351)
    sd = synthetic.synthetic_superpoint_state_dict()
    with torch.no_grad():
        out = superpoint_oracle.superpoint_forward(sd, superpoint_oracle.gray_u8_to_tensor(gray))
    assert out["keypoints"].shape[0] == int(g["k_raw_2"]) > 5000
    sel = g["sel_2"].astype(np.int64)
    np.testing.assert_array_equal(out["keypoints"].numpy()[sel].astype(np.int16), g["keypoints_2"])
    np.testing.assert_allclose(out["scores"].numpy()[sel], g["scores_2"], rtol=0, atol=1e-6)
    # gtsfm/common/image.py:108-111 at 1920 x 1080, rescaled like gtsfm/loader/loader_base.py:224-233 to 1351 x 760
    f = 21.0 / 35.0 * 1920 * (1351 / 1920)
    np.testing.assert_allclose(g["intrinsics"][2], [f, f, 960 * (1351 / 1920), 540 * (760 / 1080)], rtol=1e-12)
    k0, k2 = g["keypoints_0"].astype(np.float32), g["keypoints_2"].astype(np.float32)
    ver = verifier_oracle.verify(k0, k2, g["matches_0_2"].astype(np.int64), tuple(g["intrinsics"][0]), tuple(g["intrinsics"][2]), float(g["threshold_px"]), seed=(0 << 32) | 2)
    np.testing.assert_array_equal(np.asarray(ver["v_corr_idxs"]).astype(np.int64).reshape(-1, 2), g["v_corr_idxs_0_2"].astype(np.int64))
    np.testing.assert_allclose(np.asarray(ver["R"]), g["R_0_2"], rtol=0, atol=1e-12)
    assert len(g["matches_0_1"]) + len(g["matches_0_2"]) + len(g["matches_1_2"]) == 70
