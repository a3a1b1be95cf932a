"""CPU: the arithmetic claims behind the opt-in bf16x3 mode (gtsfm_amd/csrc/bf16x3.h), checked with numpy bit operations.

(1) ``x3_split``: an fp32 number is the EXACT sum of three bf16 pieces (hi = x truncated to its top 16 bits, mid = (x - hi) truncated,
    lo = (x - hi - mid) truncated): 8 + 8 + 8 significand bits -- for numbers of any sign whose lowest piece is still a normal bf16
    (|x| >= 2^-102; below that the pieces run into bf16's subnormal range, which has fp32's exponent but 16 fewer fraction bits).
(2) the six executed piece products (hi hi, hi mid, mid hi, mid mid, hi lo, lo hi), each exact in fp32, reproduce x * y to within 2^-21 |x y| in
    the worst case and about 2^-24 |x y| on average (the split truncates, so |mid| < 2^-7 and |lo| < 2^-15 of x's power of two): the three dropped
    products (mid lo, lo mid, lo lo) are what is missing, and nothing else.
The kernels' own accuracy against float64 is measured on the GPU (tests/test_attention_bf16x3_gpu.py, tests/test_lightglue_fp64_arbiter_gpu.py)."""

import numpy as np


def _trunc_bf16(x: np.ndarray) -> np.ndarray:
    """fp32 -> the fp32 value of its top 16 bits (what the kernel's `& 0xffff0000` and the bf16 pack keep)."""
    return (x.astype(np.float32).view(np.uint32) & np.uint32(0xFFFF0000)).view(np.float32)


def split3(x: np.ndarray):
    x = x.astype(np.float32)
    hi = _trunc_bf16(x)
    r1 = (x - hi).astype(np.float32)  # exact in fp32 (the kernel computes it with one v_sub_f32)
    mid = _trunc_bf16(r1)
    r2 = (r1 - mid).astype(np.float32)
    lo = _trunc_bf16(r2)              # the pack truncates the last piece too
    return hi, mid, lo, r2


def _samples():
    rng = np.random.default_rng(7)
    mant = rng.random(200000, dtype=np.float64) + 1.0
    expo = rng.integers(-60, 60, 200000)
    sign = rng.choice([-1.0, 1.0], 200000)
    x = (sign * mant * np.exp2(expo)).astype(np.float32)
    special = np.array([0.0, -0.0, 1.0, -1.0, 255.99998, 2.0**-102, 3.4e38, 1.0 + 2.0**-23, 1.0 - 2.0**-24, 1.9999999], dtype=np.float32)
    return np.concatenate([x, special])


def test_three_bf16_pieces_reconstruct_an_fp32_number_exactly():
    x = _samples()
    hi, mid, lo, r2 = split3(x)
    # the residual after two pieces fits the third one without truncation loss: 24 significand bits = 8 + 8 + 8
    np.testing.assert_array_equal(lo, r2)
    total = hi.astype(np.float64) + mid.astype(np.float64) + lo.astype(np.float64)
    np.testing.assert_array_equal(total, x.astype(np.float64))
    # every piece is a bf16 (low 16 bits clear) and the pieces shrink by at least 2^-8 each (normal range)
    for piece in (hi, mid, lo):
        assert not np.any(piece.view(np.uint32) & np.uint32(0xFFFF))
    normal = np.abs(x) > 2.0**-100
    assert np.all(np.abs(mid[normal]) <= np.abs(x[normal]) * 2.0**-7) and np.all(np.abs(lo[normal]) <= np.abs(x[normal]) * 2.0**-15)


def test_six_piece_products_are_within_one_fp32_rounding_class_of_the_product():
    x, y = _samples(), _samples()[::-1].copy()
    keep = (np.abs(x) > 2.0**-40) & (np.abs(y) > 2.0**-40) & (np.abs(x) < 2.0**40) & (np.abs(y) < 2.0**40)
    x, y = x[keep], y[keep]
    xh, xm, xl, _ = split3(x)
    yh, ym, yl, _ = split3(y)
    d = np.float64
    executed = xh.astype(d) * yh + xh.astype(d) * ym + xm.astype(d) * yh + xm.astype(d) * ym + xh.astype(d) * yl + xl.astype(d) * yh
    dropped = xm.astype(d) * yl + xl.astype(d) * ym + xl.astype(d) * yl
    exact = x.astype(d) * y.astype(d)
    np.testing.assert_allclose(executed + dropped, exact, rtol=1e-15)  # nothing else is missing
    rel = np.abs(executed - exact) / np.abs(exact)
    assert rel.max() < 2.0**-21, rel.max()
    assert rel.mean() < 2.0**-24, rel.mean()
    assert np.all(executed * np.sign(exact) <= np.abs(exact))  # truncating pieces share x's sign: the dropped terms pull towards zero
    # each executed piece product is exact in fp32: two 8-bit significands give at most 16 bits
    for a, b in ((xh, yh), (xh, ym), (xm, yh), (xm, ym), (xh, yl), (xl, yh)):
        p32 = (a * b).astype(np.float32)
        np.testing.assert_array_equal(p32.astype(d), a.astype(d) * b.astype(d))
