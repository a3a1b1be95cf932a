"""-m gpu: the opt-in attention arithmetic GTSFM_ATTENTION_MATH=bf16x3 (both attention products on v_mfma_f32_32x32x16_bf16 with every
fp32 operand split exactly into three bf16 pieces, fp32 accumulation; attention_kernels.hip) under the SAME parity checks as the
exact-fp32 default: the reference-written SuperGlue goldens at the benchmark's shapes, the HuggingFace-port LightGlue goldens, the
LightGlue oracle at the 5000-keypoint cap -- match indices identical, scores within 1e-4 -- and run-to-run determinism of the two-stream
pipeline. (The kernel-level check against float64 is tests/test_matchers_gpu.py::test_attention_bf16x3_arithmetic; the whole matcher
suite also passes with the variable set in the environment, profiles/r04_gpu_tests_bf16x3.txt.)"""

import numpy as np
import pytest
import torch

from conftest import GOLDEN
from gtsfm_amd.utils import synthetic

pytestmark = pytest.mark.gpu


@pytest.fixture()
def bf16x3(monkeypatch):
    monkeypatch.setenv("GTSFM_ATTENTION_MATH", "bf16x3")  # read per call by gtsfm_{sg,lg}_workspace_bytes / gtsfm_{sg,lg}_forward*


@pytest.mark.parametrize("name", ["plain", "early_stop", "pruning", "early_stop_pruning", "n2048_full_depth"])
def test_lightglue_hf_goldens_under_bf16x3(gpu_device, bf16x3, name):
    from test_lightglue_hf_golden_gpu import test_hip_path_equals_the_hf_fixture

    test_hip_path_equals_the_hf_fixture(gpu_device, name)


@pytest.mark.parametrize("path", sorted(GOLDEN.glob("bench_superglue_*.npz")), ids=lambda p: p.stem)
def test_reference_superglue_goldens_under_bf16x3(gpu_device, bf16x3, path):
    from test_bench_shapes_gpu import test_superglue_full_depth_matches_reference_golden

    test_superglue_full_depth_matches_reference_golden(gpu_device, path)


def test_lightglue_oracle_at_the_cap_under_bf16x3(gpu_device, bf16x3):
    from test_bench_shapes_gpu import LG_BENCH_CASES, test_lightglue_full_depth_vs_oracle

    test_lightglue_full_depth_vs_oracle(gpu_device, *LG_BENCH_CASES[-1])  # 5000 x 4800 with pruning


def test_bf16x3_differs_from_exact_fp32_only_within_tolerance(gpu_device, monkeypatch):
    """The two arithmetics on one pair at N = 2048: the same matches, scores within 1e-4 of each other -- and NOT the same bits (a test
    that silently ran the exact kernel twice would pass everything above)."""
    from gtsfm_amd.runtime import matcher_engine as ME

    k0, s0, d0, k1, s1, d1, _ = synthetic.synthetic_pair_features(2048, 2000, (1024, 1024), (1024, 1024), seed=77)
    eng = ME.LightGlueEngine(synthetic.synthetic_lightglue_state_dict(), gpu_device)  # the weights of test_bench_shapes_gpu's N = 2048 case: > 1000 matches
    monkeypatch.setenv("GTSFM_PLUGIN_IMAGE_CACHE", "0")
    eng.image_cache_capacity = 0  # the per-image cache would hand the second run the first run's first block
    exact = eng.match_pair(k0, d0, k1, d1, (1024, 1024), (1024, 1024))
    monkeypatch.setenv("GTSFM_ATTENTION_MATH", "bf16x3")
    split = eng.match_pair(k0, d0, k1, d1, (1024, 1024), (1024, 1024))
    again = eng.match_pair(k0, d0, k1, d1, (1024, 1024), (1024, 1024))
    np.testing.assert_array_equal(exact["matches0"], split["matches0"])
    np.testing.assert_array_equal(exact["matches1"], split["matches1"])
    assert (exact["matches0"] > -1).sum() > 100
    diff = np.abs(exact["matching_scores0"] - split["matching_scores0"]).max()
    assert 0.0 < diff < 1e-4, diff
    np.testing.assert_array_equal(split["matching_scores0"], again["matching_scores0"])


def test_bf16x3_two_stream_pipeline_is_deterministic(gpu_device, bf16x3):
    """Pair chunks alternate over two HIP streams; a chunk's result must not depend on what runs beside it: 40 repetitions of a
    three-chunk step, bit-identical every time (a two-score-tile variant of the kernel failed exactly this in 1 - 5 % of the runs)."""
    from gtsfm_amd.runtime import matcher_engine as ME
    from gtsfm_amd.runtime.pipeline import FrontEndPipeline
    from gtsfm_amd.runtime.superpoint_engine import SuperPointEngine

    det = SuperPointEngine(synthetic.synthetic_superpoint_state_dict(), gpu_device)
    eng = ME.LightGlueEngine(synthetic.synthetic_lightglue_state_dict(num_layers=3), gpu_device)
    views = synthetic.synthetic_overlapping_views(5, 192, 256, seed=31)
    pairs = [(0, 1), (0, 2), (1, 2), (2, 3), (0, 3), (3, 4)]
    pipe = FrontEndPipeline(det, eng, max_keypoints=256, pair_chunk=2, num_streams=2, use_graphs=False, share_first_layer=False)
    feats = pipe.detect(torch.from_numpy(views).to(gpu_device))
    ref = pipe.match(feats, pairs, [(192, 256)] * 5)
    torch.cuda.synchronize()
    assert sum(int((r["matches"] > -1).sum()) for r in ref) > 0
    for _ in range(40):
        out = pipe.match(feats, pairs, [(192, 256)] * 5)
        torch.cuda.synchronize()
        for x, y in zip(ref, out):
            assert torch.equal(x["matches"], y["matches"]) and torch.equal(x["mscores"], y["mscores"])
