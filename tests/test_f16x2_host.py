"""CPU: the arithmetic claims behind the opt-in f16x2 mode (gtsfm_amd/csrc/f16x2.h), checked with numpy's IEEE float16 (round to nearest even,
subnormals kept -- what v_cvt_pk_f16_f32 and the fp16 matrix pipe do on gfx950: tools/probe_f16_mfma.hip, profiles/r06_probe_f16_mfma.txt).

(1) ``h2_split_pair``: hi = RN16(x), lo = RN16(x - hi) with x - hi EXACT in fp32; hi + lo misses x by at most 2^-22 |x| (2^-25 |x| on average) while lo
    is a normal fp16 (|x| >= 2^-3) and by at most 2^-25 absolute below that.
(2) the three executed piece products (lo hi, hi lo, hi hi), each exact in fp32, reproduce x y to within 2^-20.9 |x y| in the worst case and about
    2^-24 |x y| on average -- the class of bf16x3's six products (tests/test_bf16x3_host.py: < 2^-21, < 2^-24) -- and to within
    2^-21 |x y| + 2^-24 (|x| + |y|) for ANY magnitudes inside fp16's range; the dropped product lo lo and the two representation errors are all
    that is missing.
(3) beyond +-65504 the leading piece is infinite: the mode fails loudly (NaN), it never clamps.
The kernels' own accuracy against float64 is measured on the GPU (tests/test_attention_bf16x3_gpu.py, tests/test_matchers_gpu.py::test_attention_bf16x3_arithmetic[f16x2])."""

import numpy as np


def split2(x: np.ndarray):
    x = x.astype(np.float32)
    with np.errstate(over="ignore"):
        hi = x.astype(np.float16)
    with np.errstate(invalid="ignore"):
        r = (x - hi.astype(np.float32)).astype(np.float32)  # one v_sub_f32 (or the f16-source form of v_fma_mix_f32) in the kernel
        lo = r.astype(np.float16)
    return hi, lo, r


def _samples(lo_exp=-30, hi_exp=15, n=400000, seed=7):
    rng = np.random.default_rng(seed)
    mant = rng.random(n, dtype=np.float64) + 1.0
    expo = rng.integers(lo_exp, hi_exp, n)
    sign = rng.choice([-1.0, 1.0], n)
    x = (sign * mant * np.exp2(expo)).astype(np.float32)
    special = np.array([0.0, -0.0, 1.0, -1.0, 255.99998, 0.125, 0.12499999, 65504.0, -65504.0, 1.0 + 2.0**-23, 1.0 - 2.0**-24, 2.0**-14, 2.0**-24, 2.0**-26], dtype=np.float32)
    return np.concatenate([x, special])


def test_the_residual_is_exact_and_two_pieces_keep_22_bits():
    x = _samples()
    hi, lo, r = split2(x)
    d = np.float64
    np.testing.assert_array_equal(r.astype(d), x.astype(d) - hi.astype(d))  # x - hi needs no rounding in fp32
    err = np.abs(x.astype(d) - hi.astype(d) - lo.astype(d))
    normal_lo = np.abs(x) >= 2.0**-3
    rel = err[normal_lo] / np.abs(x[normal_lo])
    assert rel.max() <= 2.0**-22 and rel.mean() < 2.0**-25, (rel.max(), rel.mean())
    assert err[~normal_lo].max() <= 2.0**-25
    assert np.all(np.abs(lo.astype(d)) <= np.abs(x.astype(d)) * 2.0**-11 + 2.0**-25)  # the second piece is at most half an ulp of the first


def test_three_piece_products_are_within_the_class_of_bf16x3():
    x, y = _samples(), _samples()[::-1].copy()
    xh, xl, _ = split2(x)
    yh, yl, _ = split2(y)
    d = np.float64
    executed = xl.astype(d) * yh + xh.astype(d) * yl + xh.astype(d) * yh
    exact = x.astype(d) * y.astype(d)
    # nothing but lo lo and the two representation errors is missing
    ex, ey = x.astype(d) - xh.astype(d) - xl.astype(d), y.astype(d) - yh.astype(d) - yl.astype(d)
    rest = xl.astype(d) * yl + ex * y.astype(d) + (xh.astype(d) + xl.astype(d)) * ey
    np.testing.assert_allclose(executed + rest, exact, rtol=1e-13, atol=1e-300)
    err = np.abs(executed - exact)
    assert np.all(err <= 2.0**-21 * np.abs(exact) + 2.0**-24 * (np.abs(x.astype(d)) + np.abs(y.astype(d))))
    both = (np.abs(x) >= 2.0**-3) & (np.abs(y) >= 2.0**-3)
    rel = err[both] / np.abs(exact[both])
    assert rel.max() < 2.0**-20.9 and rel.mean() < 2.0**-24, (np.log2(rel.max()), np.log2(rel.mean()))
    # each executed piece product is exact in fp32: two 11-bit significands give at most 22 bits (and the exponents stay inside fp32's range)
    for a, b in ((xh, yh), (xh, yl), (xl, yh)):
        p32 = a.astype(np.float32) * b.astype(np.float32)
        np.testing.assert_array_equal(p32.astype(d), a.astype(d) * b.astype(d))


def test_softmax_weights_shifted_by_seven_use_fp16s_range_from_the_top():
    """attention_x3_kernel<.., 2> keeps the reference exponent 7 below the running maximum: weights 2^(s - m) lie in (0, 2^15]; a weight 2^-21 of the
    row's largest (2^7 after a rebase) is still a NORMAL fp16, and the largest possible one (2^15) is far from 65504."""
    w = np.exp2(np.array([15.0, 7.0, 0.0, -14.0], dtype=np.float32))
    hi, lo, _ = split2(w)
    assert np.all(np.isfinite(hi.astype(np.float32))) and np.array_equal(hi.astype(np.float32), w) and not lo.any()
    assert np.float16(2.0**-14) == np.finfo(np.float16).smallest_normal and 2.0**-14 / 2.0**7 == 2.0**-21


def test_out_of_range_operands_fail_loudly():
    hi, lo, _ = split2(np.array([70000.0, -1e6, 65520.0], dtype=np.float32))
    assert np.all(np.isinf(hi.astype(np.float32)))
    assert np.all(np.isnan(lo.astype(np.float32)) | np.isinf(lo.astype(np.float32)))


def test_the_engines_raise_on_nan_scores_only_under_the_f16x2_switches(monkeypatch):
    """``check_split_arithmetic_range``: NaN scores are an error under GTSFM_ATTENTION_MATH / GTSFM_GEMM_MATH = f16x2 (an operand left fp16's range) and are
    passed through in exact fp32 ("f32" starts with an f as well) and under bf16x3 (fp32's range: NaN there means NaN inputs)."""
    import pytest

    from gtsfm_amd.runtime.matcher_engine import check_split_arithmetic_range, split_arithmetic_has_fp16_range

    scores = np.array([0.5, np.nan, 0.0], dtype=np.float32)
    for k in ("GTSFM_ATTENTION_MATH", "GTSFM_GEMM_MATH"):
        monkeypatch.delenv(k, raising=False)
    assert not split_arithmetic_has_fp16_range()
    check_split_arithmetic_range(scores)
    for value in ("f32", "bf16x3", ""):
        monkeypatch.setenv("GTSFM_GEMM_MATH", value)
        assert not split_arithmetic_has_fp16_range()
        check_split_arithmetic_range(scores)
    for key in ("GTSFM_ATTENTION_MATH", "GTSFM_GEMM_MATH"):
        monkeypatch.setenv("GTSFM_GEMM_MATH", "f32")
        monkeypatch.setenv("GTSFM_ATTENTION_MATH", "f32")
        monkeypatch.setenv(key, "f16x2")
        assert split_arithmetic_has_fp16_range()
        check_split_arithmetic_range(np.zeros(4, dtype=np.float32))
        with pytest.raises(FloatingPointError, match="fp16's range"):
            check_split_arithmetic_range(scores)
