"""-m gpu: the opt-in arithmetics GTSFM_ATTENTION_MATH / GTSFM_GEMM_MATH = bf16x3 (both attention products on v_mfma_f32_32x32x16_bf16 with every
fp32 operand split exactly into three bf16 pieces, fp32 accumulation; attention_kernels.hip, bf16x3.h) and = f16x2 (round 6: two fp16 pieces per
operand, three v_mfma_f32_32x32x16_f16 per block; f16x2.h) -- every test below runs once per mode -- under the SAME parity checks as the
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


MODES = ["bf16x3", "f16x2"]


@pytest.fixture(params=MODES)
def bf16x3(monkeypatch, request):
    """The attention switch set to one of the two split arithmetics (the fixture keeps its round-4 name)."""
    monkeypatch.setenv("GTSFM_ATTENTION_MATH", request.param)  # read per call by gtsfm_{sg,lg}_workspace_bytes / gtsfm_{sg,lg}_forward*
    return request.param


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


@pytest.mark.parametrize("mode", MODES)
def test_bf16x3_differs_from_exact_fp32_only_within_tolerance(gpu_device, monkeypatch, mode):
    """The two arithmetics on one pair at N = 2048: the same matches, scores within 1e-4 of each other -- and NOT the same bits (a test
    that silently ran the exact kernel twice would pass everything above)."""
    from gtsfm_amd.runtime import matcher_engine as ME

    k0, s0, d0, k1, s1, d1, _ = synthetic.synthetic_pair_features(2048, 2000, (1024, 1024), (1024, 1024), seed=77)
    eng = ME.LightGlueEngine(synthetic.synthetic_lightglue_state_dict(), gpu_device)  # the weights of test_bench_shapes_gpu's N = 2048 case: > 1000 matches
    monkeypatch.setenv("GTSFM_PLUGIN_IMAGE_CACHE", "0")
    eng.image_cache_capacity = 0  # the per-image cache would hand the second run the first run's first block
    monkeypatch.delenv("GTSFM_ATTENTION_MATH", raising=False)  # (the file also runs with both switches exported: the first run must still be exact)
    monkeypatch.delenv("GTSFM_GEMM_MATH", raising=False)
    exact = eng.match_pair(k0, d0, k1, d1, (1024, 1024), (1024, 1024))
    monkeypatch.setenv("GTSFM_ATTENTION_MATH", mode)
    split = eng.match_pair(k0, d0, k1, d1, (1024, 1024), (1024, 1024))
    again = eng.match_pair(k0, d0, k1, d1, (1024, 1024), (1024, 1024))
    np.testing.assert_array_equal(exact["matches0"], split["matches0"])
    np.testing.assert_array_equal(exact["matches1"], split["matches1"])
    assert (exact["matches0"] > -1).sum() > 100
    diff = np.abs(exact["matching_scores0"] - split["matching_scores0"]).max()
    assert 0.0 < diff < 1e-4, diff
    np.testing.assert_array_equal(split["matching_scores0"], again["matching_scores0"])


@pytest.mark.parametrize("gemm_too", [False, True], ids=["attention", "attention+gemm"])
def test_bf16x3_two_stream_pipeline_is_deterministic(gpu_device, bf16x3, monkeypatch, gemm_too):
    """Pair chunks alternate over two HIP streams; a chunk's result must not depend on what runs beside it: 80 repetitions of a
    three-chunk step, bit-identical every time (a two-score-tile variant of the kernel failed exactly this in 1 - 5 % of the runs in round 4;
    in round 6 a build whose DMA issue the compiler had interleaved with the softmax tail failed it in 6 - 35 %: profiles/r06_x3_two_stream_bisect.txt).
    Also with the GEMM switch set to the same arithmetic."""
    if gemm_too:
        monkeypatch.setenv("GTSFM_GEMM_MATH", bf16x3)
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
    for _ in range(80):
        out = pipe.match(feats, pairs, [(192, 256)] * 5)
        torch.cuda.synchronize()
        for x, y in zip(ref, out):
            assert torch.equal(x["matches"], y["matches"]) and torch.equal(x["mscores"], y["mscores"])


@pytest.mark.parametrize("m,k,n,relu,res,m_live,n_live", [(1000, 512, 512, 1, 1, 1000, 512), (131, 256, 768, 0, 0, 131, 768), (640, 512, 256, 0, 1, 640, 132),
                                                          (4097, 32, 128, 1, 0, 4097, 128), (260, 256, 300, 0, 1, 200, 296)])
@pytest.mark.parametrize("mode", MODES)
def test_gemm_bf16x3_arithmetic(gpu_device, monkeypatch, mode, m, k, n, relu, res, m_live, n_live):
    """GTSFM_GEMM_MATH=bf16x3: the LDS-DMA GEMM's products on bf16 MFMA with both operands split exactly into three bf16 pieces in registers,
    same stages / epilogues / masking as the exact kernel. Against a FLOAT64 reference its error must be of the exact kernel's class (at most
    twice + 1e-6 of the result's scale), untouched cells stay untouched, and it must NOT be the exact kernel's bits (the switch did something)."""
    import torch.nn.functional as F

    from gtsfm_amd.runtime import lib as L

    lib = L.load()
    gen = torch.Generator().manual_seed(m * 3 + k + n)
    a = torch.randn((m, k), generator=gen)
    w = torch.randn((n, k), generator=gen) / k**0.5
    b = torch.randn((n,), generator=gen)
    r = torch.randn((m, n + 4), generator=gen)
    ref = F.linear(a.double(), w.double(), b.double()) * 0.5
    if relu:
        ref = F.relu(ref)
    if res:
        ref = r[:, :n].double() + ref
    ad, wd, rd = a.to(gpu_device), w.to(gpu_device), r.to(gpu_device)
    bp = torch.zeros((n + 63) // 64 * 64, device=gpu_device)
    bp[:n] = b.to(gpu_device)
    md = torch.tensor([m_live], dtype=torch.int32, device=gpu_device)
    nd = torch.tensor([n_live], dtype=torch.int32, device=gpu_device)
    monkeypatch.setenv("GTSFM_GEMM_SMALL_BELOW", "0")  # the 128 x 128 tiling in both runs

    def run(math):
        monkeypatch.setenv("GTSFM_GEMM_MATH", math)
        out = torch.full((m, n + 8), -5.0, device=gpu_device)
        rc = lib.gtsfm_linear_rowmajor_f32(ad.data_ptr(), k, m, md.data_ptr() if m_live < m else None, k, wd.data_ptr(), k, bp.data_ptr(), n,
                                           nd.data_ptr() if n_live < n else None, out.data_ptr(), n + 8, 4, rd.data_ptr() if res else None, n + 4,
                                           0.5, relu, torch.cuda.current_stream().cuda_stream)
        assert rc == 0, lib.gtsfm_last_error()
        torch.cuda.synchronize()
        return out.cpu()

    exact, x3 = run("f32"), run(mode)
    for got in (exact, x3):
        assert torch.all(got[:, :4] == -5.0) and torch.all(got[:, n_live + 4 :] == -5.0) and torch.all(got[m_live:] == -5.0)
    live = (slice(0, m_live), slice(4, n_live + 4))
    e_exact = float((exact[live].double() - ref[:m_live, :n_live]).abs().max())
    e_x3 = float((x3[live].double() - ref[:m_live, :n_live]).abs().max())
    assert e_x3 <= 2.0 * e_exact + 1e-6 * max(1.0, float(ref.abs().max())), (e_x3, e_exact)
    assert not torch.equal(exact[live], x3[live])


@pytest.mark.parametrize("mode", MODES)
def test_matcher_goldens_under_both_bf16x3_switches(gpu_device, monkeypatch, mode):
    """Attention AND projection / score GEMMs in the bf16x3 arithmetic (GTSFM_ATTENTION_MATH + GTSFM_GEMM_MATH): the reference-written
    SuperGlue golden at 5000 x 4800, the HuggingFace-port LightGlue golden at N = 2048 and the LightGlue oracle at the cap still hold --
    indices identical, scores within 1e-4."""
    from test_bench_shapes_gpu import LG_BENCH_CASES, test_lightglue_full_depth_vs_oracle, test_superglue_full_depth_matches_reference_golden
    from test_lightglue_hf_golden_gpu import test_hip_path_equals_the_hf_fixture

    monkeypatch.setenv("GTSFM_ATTENTION_MATH", mode)
    monkeypatch.setenv("GTSFM_GEMM_MATH", mode)
    monkeypatch.setenv("GTSFM_GEMM_SMALL_BELOW", "0")  # (both tilings honour the switch; the goldens are checked on the batch tiling)
    test_superglue_full_depth_matches_reference_golden(gpu_device, GOLDEN / "bench_superglue_5000x4800_s15_it20.npz")
    test_hip_path_equals_the_hf_fixture(gpu_device, "n2048_full_depth")
    test_lightglue_full_depth_vs_oracle(gpu_device, *LG_BENCH_CASES[-1])


@pytest.mark.parametrize("mode", MODES)
def test_superpoint_is_bit_identical_under_the_gemm_switch(gpu_device, monkeypatch, mode):
    """GTSFM_GEMM_MATH=bf16x3 is the MATCHERS' switch (GemmParams.math, set by gtsfm_{sg,lg}_forward* and the stand-alone linear entry points):
    SuperPoint's convPb / convDb reach the same LDS-DMA launcher (superpoint_api.hip, 1x1 convolutions with K = 256) and must stay exact fp32 --
    score maps, keypoints, scores and descriptors bit for bit the same with the variable set (until round 5 the launcher read
    the environment itself and the logits changed with it)."""
    from gtsfm_amd.runtime.superpoint_engine import SuperPointEngine

    eng = SuperPointEngine(synthetic.synthetic_superpoint_state_dict(), gpu_device)
    imgs = torch.from_numpy(np.stack([synthetic.synthetic_gray_image(240, 320, s) for s in (5, 6)])).to(gpu_device)
    exact = eng.forward(imgs, return_score_maps=True)
    monkeypatch.setenv("GTSFM_GEMM_MATH", mode)
    monkeypatch.setenv("GTSFM_ATTENTION_MATH", mode)
    switched = eng.forward(imgs, return_score_maps=True)
    assert int(exact["count"].min()) > 100
    for key in ("count", "dense_scores", "nms_scores"):
        assert torch.equal(exact[key], switched[key]), key
    for i in range(imgs.shape[0]):  # rows beyond an image's count are uninitialised capacity
        k = int(exact["count"][i])
        for key in ("xy", "scores", "descriptors"):
            assert torch.equal(exact[key][i, :k], switched[key][i, :k]), (key, i)


def test_f16x2_fails_loudly_beyond_fp16_range(gpu_device, monkeypatch):
    """f16x2 has fp16's exponent range: descriptors scaled by 1e7 put the tokens and the projected queries / keys beyond 65504, the leading pieces become inf, every
    score NaN -- and the engine raises instead of returning "no matches". The same input in exact fp32 and under bf16x3 (fp32's range) matches."""
    from gtsfm_amd.runtime import matcher_engine as ME

    k0, s0, d0, k1, s1, d1, _ = synthetic.synthetic_pair_features(600, 500, (480, 640), (480, 640), seed=5)
    eng = ME.LightGlueEngine(synthetic.synthetic_lightglue_state_dict(num_layers=2), gpu_device)
    eng.image_cache_capacity = 0
    big0, big1 = d0 * np.float32(1e7), d1 * np.float32(1e7)
    for mode in ("f32", "bf16x3"):
        monkeypatch.setenv("GTSFM_ATTENTION_MATH", mode)
        monkeypatch.setenv("GTSFM_GEMM_MATH", mode)
        out = eng.match_pair(k0, big0, k1, big1, (480, 640), (480, 640))
        assert not np.isnan(out["matching_scores0"]).any()
    monkeypatch.setenv("GTSFM_ATTENTION_MATH", "f16x2")
    monkeypatch.setenv("GTSFM_GEMM_MATH", "f16x2")
    with pytest.raises(FloatingPointError, match="fp16's range"):
        eng.match_pair(k0, big0, k1, big1, (480, 640), (480, 640))
    ok = eng.match_pair(k0, d0, k1, d1, (480, 640), (480, 640))  # the engine is fine afterwards
    assert not np.isnan(ok["matching_scores0"]).any()


def test_captured_graphs_follow_the_switches(gpu_device, monkeypatch):
    """``FrontEndPipeline(use_graphs=True)`` replays a captured launch sequence per chunk shape; the C side reads the arithmetic switches when the sequence
    is CAPTURED. The graph cache is keyed by the switches (round 6), so one pipeline object serves exact fp32 -> f16x2 -> exact fp32 calls with the
    arithmetic each call asked for: the first and the third result are bit-identical, the second differs from them within the tolerance."""
    from gtsfm_amd.runtime import matcher_engine as ME
    from gtsfm_amd.runtime.pipeline import FrontEndPipeline
    from gtsfm_amd.runtime.superpoint_engine import SuperPointEngine

    for k in ("GTSFM_ATTENTION_MATH", "GTSFM_GEMM_MATH"):
        monkeypatch.delenv(k, raising=False)
    det = SuperPointEngine(synthetic.synthetic_superpoint_state_dict(), gpu_device)
    eng = ME.LightGlueEngine(synthetic.synthetic_lightglue_state_dict(num_layers=3), gpu_device)
    views = synthetic.synthetic_overlapping_views(4, 192, 256, seed=33)
    pairs = [(0, 1), (1, 2), (2, 3), (0, 3)]
    pipe = FrontEndPipeline(det, eng, max_keypoints=256, pair_chunk=2, num_streams=2, use_graphs=True, share_first_layer=False)
    feats = pipe.detect(torch.from_numpy(views).to(gpu_device))
    assert int(feats["count"].min()) == 256  # full chunks: the graph path

    def run():
        out = pipe.match(feats, pairs, [(192, 256)] * 4)
        torch.cuda.synchronize()
        return [(r["matches"].clone(), r["mscores"].clone()) for r in out]

    exact = run()
    graphs_exact = len(pipe._graphs)
    assert graphs_exact > 0
    monkeypatch.setenv("GTSFM_ATTENTION_MATH", "f16x2")
    monkeypatch.setenv("GTSFM_GEMM_MATH", "f16x2")
    split = run()
    assert len(pipe._graphs) == 2 * graphs_exact  # its own captures
    monkeypatch.delenv("GTSFM_ATTENTION_MATH")
    monkeypatch.delenv("GTSFM_GEMM_MATH")
    again = run()
    assert len(pipe._graphs) == 2 * graphs_exact
    assert sum(int((m > -1).sum()) for m, _ in exact) > 0
    for (m0, s0), (m1, s1), (m2, s2) in zip(exact, split, again):
        assert torch.equal(m0, m2) and torch.equal(s0, s2)
        assert torch.equal(m0, m1) and not torch.equal(s0, s1) and float((s0 - s1).abs().max()) < 1e-4


def test_the_image_cache_follows_the_switches(gpu_device, monkeypatch):
    """The per-call path keeps an image's uploaded keypoints and the OUTPUT of the matcher's per-image first block on the device. That output belongs to the
    arithmetic it was computed under: the cache key carries the two switches (round 6), so the same host arrays matched exact -> f16x2 -> exact give the exact
    bits again in the third call and the f16x2 bits (not a mixture) in the second."""
    from gtsfm_amd.runtime import matcher_engine as ME

    for k in ("GTSFM_ATTENTION_MATH", "GTSFM_GEMM_MATH", "GTSFM_PLUGIN_IMAGE_CACHE"):
        monkeypatch.delenv(k, raising=False)
    k0, s0, d0, k1, s1, d1, _ = synthetic.synthetic_pair_features(900, 800, (480, 640), (480, 640), seed=21)
    eng = ME.LightGlueEngine(synthetic.synthetic_lightglue_state_dict(num_layers=3), gpu_device)
    assert eng.image_cache_capacity > 0
    run = lambda: eng.match_pair(k0, d0, k1, d1, (480, 640), (480, 640))  # noqa: E731
    exact = run()
    monkeypatch.setenv("GTSFM_ATTENTION_MATH", "f16x2")
    monkeypatch.setenv("GTSFM_GEMM_MATH", "f16x2")
    split = run()
    fresh = ME.LightGlueEngine(synthetic.synthetic_lightglue_state_dict(num_layers=3), gpu_device)  # nothing cached: the mode's own bits
    fresh.image_cache_capacity = 0
    split_ref = fresh.match_pair(k0, d0, k1, d1, (480, 640), (480, 640))
    monkeypatch.delenv("GTSFM_ATTENTION_MATH")
    monkeypatch.delenv("GTSFM_GEMM_MATH")
    again = run()
    assert (exact["matches0"] > -1).sum() > 50
    np.testing.assert_array_equal(exact["matching_scores0"], again["matching_scores0"])
    np.testing.assert_array_equal(split["matching_scores0"], split_ref["matching_scores0"])
    assert not np.array_equal(exact["matching_scores0"], split["matching_scores0"])
    np.testing.assert_array_equal(exact["matches0"], split["matches0"])
