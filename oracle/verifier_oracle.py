"""CPU oracle of the verifier stage -- TEST INFRASTRUCTURE, never imported by the product.

What the reference does (``gtsfm/frontend/verifier/ransac.py:52-84``, ``opencv_verifier_base.py:47-111``,
``gtsfm/utils/verification.py:54-96,172-220``, called from ``gtsfm/two_view_estimator.py:391-397``):

1. normalise the keypoint coordinates with the intrinsics (``gtsfm/utils/features.py:41-51``),
2. ``cv2.findEssentialMat(..., method=USAC_ACCURATE, threshold=px/fx, prob=0.999999)`` -- a RANSAC over 5-point minimal
   samples with the squared Sampson error as the residual,
3. ``cv.recoverPose`` -- the one of the four (R, t) decompositions of E with most points in front of both cameras,
4. return ``(i2Ri1, i2Ui1, match_indices[inliers], mean(inlier_mask))``.

PARITY UNPINNED. Steps 2 and 3 live in OpenCV (``opencv-python>=4.5.4.60``, pyproject.toml:75), which is absent from this
snapshot and from the GPU box, and whose USAC sampler draws from an internal generator: its inlier sets are not reproducible
even with OpenCV at hand, and the reference's four verifiers (Ransac, LoRansac, Degensac, LMEDS) do not agree with one
another either. What is restated here is the published mathematics --

* the five-point relative-pose solver (Nister, PAMI 2004, section 3: null space of the 5x9 epipolar system, the ten cubic
  constraints det(E) = 0 and 2 E E^T E - tr(E E^T) E = 0, Gauss-Jordan elimination, the 3x3 polynomial matrix B(z) and its
  tenth-degree determinant);
* the squared Sampson error exactly as ``gtsfm/utils/verification.py:172-220`` computes it;
* RANSAC scored as USAC scores by default -- MSAC, the sum of min(error, threshold^2), ties to the earlier hypothesis --
  with the standard stopping rule (1 - w^5)^n <= 1 - p on the winner's inlier share and OpenCV's default cap of 1000
  iterations for ``findEssentialMat`` (here 4 rounds of 256), followed by one round of LO-RANSAC-style inner sampling
  (Chum, Matas, Kittler 2003): 256 minimal samples drawn from the inliers of the winner, scored on all matches;
* ``recoverPose``'s choice among (R1, t), (R2, t), (R1, -t), (R2, -t) by counting points with depth in (0, 50) in both
  cameras (OpenCV triangulates with a DLT; here the two depths come from the 2x2 normal equations of
  ``l1 R x1 + t = l2 x2`` -- same sign decisions away from degenerate geometry)

-- with a counter-based sampler (splitmix64 of seed / hypothesis / attempt) so that the HIP kernel and this file draw the
SAME minimal samples and are compared bit-for-bit on the inlier masks (``tests/test_verifier_gpu.py``). The anchor towards
the reference is its own verifier contract suite, ``tests/frontend/verifier/test_verifier_base.py`` (two-plane scene: pose
within 2 degrees and every match verified; empty input; index validity; pickling), restated in ``tests/test_verifier.py``.
USAC_ACCURATE's graph-cut local optimisation is NOT restated (the inner-sampling round stands in for it); its final polish is
restated in spirit: six Gauss-Newton steps on the inliers' Sampson error over the five pose parameters (``polish_pose``), kept
when the MSAC cost over all matches decreases.

Every arithmetic step below is written as an explicit sequence of IEEE double operations (no ``np.dot`` / ``np.sum`` / BLAS,
no fused multiply-add), vectorised over the hypothesis axis only, so the device code can follow the same sequence."""

from __future__ import annotations

from typing import Dict, List, Optional, Tuple

import numpy as np

ROUND = 256  # hypotheses per round (one workgroup of 256 threads on the device)
MAX_ROUNDS = 4  # 1024 hypotheses >= OpenCV's findEssentialMat default maxIters = 1000
SUCCESS_PROB = 0.999999  # RANSAC_SUCCESS_PROB, gtsfm/frontend/verifier/ransac.py:22
ROOT_RANGE_CAP = 1.0e8
BISECT_ITERS = 128
JACOBI_SWEEPS = 8
DEPTH_LIMIT = 50.0  # cv.recoverPose's distanceThresh

# ---------------------------------------------------------------------------------------------------------------------
# monomial bookkeeping
LIN = [(1, 0, 0), (0, 1, 0), (0, 0, 1), (0, 0, 0)]  # x, y, z, 1
QUAD = [(2, 0, 0), (1, 1, 0), (1, 0, 1), (1, 0, 0), (0, 2, 0), (0, 1, 1), (0, 1, 0), (0, 0, 2), (0, 0, 1), (0, 0, 0)]
# Nister's order: the ten leading monomials, then x*[z^2, z, 1], y*[z^2, z, 1], [z^3, z^2, z, 1]
CUBIC = [(3, 0, 0), (0, 3, 0), (2, 1, 0), (1, 2, 0), (2, 0, 1), (2, 0, 0), (0, 2, 1), (0, 2, 0), (1, 1, 1), (1, 1, 0),
         (1, 0, 2), (1, 0, 1), (1, 0, 0), (0, 1, 2), (0, 1, 1), (0, 1, 0), (0, 0, 3), (0, 0, 2), (0, 0, 1), (0, 0, 0)]


def _add(a, b):
    return (a[0] + b[0], a[1] + b[1], a[2] + b[2])


LIN_LIN = [[QUAD.index(_add(a, b)) for b in LIN] for a in LIN]  # [4][4] -> index into QUAD
QUAD_LIN = [[CUBIC.index(_add(a, b)) for b in LIN] for a in QUAD]  # [10][4] -> index into CUBIC


def splitmix64(x: np.ndarray) -> np.ndarray:
    x = (x + np.uint64(0x9E3779B97F4A7C15)).astype(np.uint64)
    x = (x ^ (x >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
    x = (x ^ (x >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
    return x ^ (x >> np.uint64(31))


def sample_indices(seed: int, hyp: np.ndarray, m: int, size: int = 5) -> np.ndarray:
    """``size`` distinct match indices per hypothesis (5 for the essential, 7 for the fundamental matrix): draw k takes ``splitmix64(seed ^ splitmix64(hyp << 8 | attempt)) % m``
    and repeats with the next attempt while it collides with an earlier draw; after 64 attempts the remaining draws take the
    smallest unused indices (only reachable for tiny m)."""
    with np.errstate(over="ignore"):
        hyp = hyp.astype(np.uint64)
        out = np.zeros((hyp.shape[0], size), dtype=np.int64)
        attempt = np.zeros(hyp.shape[0], dtype=np.uint64)
        for k in range(size):
            done = np.zeros(hyp.shape[0], dtype=bool)
            while not done.all():
                key = (hyp << np.uint64(8)) | attempt
                draw = (splitmix64(np.uint64(seed) ^ splitmix64(key)) % np.uint64(m)).astype(np.int64)
                clash = np.zeros(hyp.shape[0], dtype=bool)
                for j in range(k):
                    clash |= out[:, j] == draw
                exhausted = attempt >= np.uint64(64)
                if exhausted.any():  # smallest unused index
                    for r in np.nonzero(exhausted & ~done)[0]:
                        used = set(out[r, :k].tolist())
                        draw[r] = next(i for i in range(m) if i not in used)
                        clash[r] = False
                take = ~done & ~clash
                out[take, k] = draw[take]
                attempt = np.where(~done & ~exhausted, attempt + np.uint64(1), attempt)
                done |= take
        return out


# ---------------------------------------------------------------------------------------------------------------------
# five-point solver, vectorised over hypotheses (arrays [H, ...])


def _null_space(q: np.ndarray) -> np.ndarray:
    """q [H,R,9] -> basis [H,9-R,9] of the null space by Gauss-Jordan with complete pivoting (first maximum wins); R = 5
    (essential matrix) or 7 (fundamental matrix)."""
    a = q.copy()
    h, nr = a.shape[0], a.shape[1]
    rows = np.arange(h)
    perm = np.tile(np.arange(9), (h, 1))
    for r in range(nr):
        best = np.full(h, -1.0)
        pr = np.full(h, r)
        pc = np.full(h, r)
        for i in range(r, nr):
            for j in range(r, 9):
                v = np.abs(a[:, i, j])
                better = v > best
                best = np.where(better, v, best)
                pr = np.where(better, i, pr)
                pc = np.where(better, j, pc)
        tmp = a[rows, r, :].copy()
        a[rows, r, :] = a[rows, pr, :]
        a[rows, pr, :] = tmp
        tmp = a[rows, :, r].copy()
        a[rows, :, r] = a[rows, :, pc]
        a[rows, :, pc] = tmp
        tmp = perm[rows, r].copy()
        perm[rows, r] = perm[rows, pc]
        perm[rows, pc] = tmp
        piv = a[:, r, r].copy()
        for j in range(r, 9):
            a[:, r, j] = a[:, r, j] / piv
        for i in range(nr):
            if i == r:
                continue
            f = a[:, i, r].copy()
            for j in range(r + 1, 9):
                a[:, i, j] = a[:, i, j] - f * a[:, r, j]
            a[:, i, r] = 0.0
    basis = np.zeros((h, 9 - nr, 9))
    for k in range(9 - nr):
        basis[rows, k, perm[:, nr + k]] = 1.0
        for i in range(nr):
            basis[rows, k, perm[:, i]] = -a[:, i, nr + k]
    return basis


def _mul_lin_lin(p, q, out):
    for a in range(4):
        for b in range(4):
            k = LIN_LIN[a][b]
            out[:, k] = out[:, k] + p[:, a] * q[:, b]


def _mul_quad_lin(p, q, out):
    for a in range(10):
        for b in range(4):
            k = QUAD_LIN[a][b]
            out[:, k] = out[:, k] + p[:, a] * q[:, b]


def _constraints(basis: np.ndarray) -> np.ndarray:
    """basis [H,4,9] (X, Y, Z, W) -> the 10x20 coefficient matrix [H,10,20] in CUBIC order: rows 0..8 the entries of
    (E E^T - tr(E E^T)/2 I) E, row 9 det(E)."""
    h = basis.shape[0]
    e = [[basis[:, :, 3 * i + j] for j in range(3)] for i in range(3)]  # each [H,4] = coefficients of x, y, z, 1
    eet = [[None] * 3 for _ in range(3)]
    for i in range(3):
        for j in range(i, 3):
            acc = np.zeros((h, 10))
            for k in range(3):
                _mul_lin_lin(e[i][k], e[j][k], acc)
            eet[i][j] = acc
            eet[j][i] = acc
    half_trace = ((eet[0][0] + eet[1][1]) + eet[2][2]) * 0.5
    lam = [[eet[i][j] - half_trace if i == j else eet[i][j] for j in range(3)] for i in range(3)]
    m = np.zeros((h, 10, 20))
    for i in range(3):
        for j in range(3):
            acc = np.zeros((h, 20))
            for k in range(3):
                _mul_quad_lin(lam[i][k], e[k][j], acc)
            m[:, 3 * i + j, :] = acc

    def minor(a, b, c, d):  # a*b - c*d
        p = np.zeros((h, 10))
        q = np.zeros((h, 10))
        _mul_lin_lin(a, b, p)
        _mul_lin_lin(c, d, q)
        return p - q

    c0 = minor(e[1][1], e[2][2], e[1][2], e[2][1])
    c1 = minor(e[1][2], e[2][0], e[1][0], e[2][2])
    c2 = minor(e[1][0], e[2][1], e[1][1], e[2][0])
    acc = np.zeros((h, 20))
    _mul_quad_lin(c0, e[0][0], acc)
    _mul_quad_lin(c1, e[0][1], acc)
    _mul_quad_lin(c2, e[0][2], acc)
    m[:, 9, :] = acc
    return m


def _gauss_jordan_10x20(m: np.ndarray) -> np.ndarray:
    """Reduced row echelon form on the ten leading columns (row pivoting, first maximum wins); returns the 10x10 tail."""
    a = m.copy()
    h = a.shape[0]
    rows = np.arange(h)
    for c in range(10):
        best = np.full(h, -1.0)
        pr = np.full(h, c)
        for i in range(c, 10):
            v = np.abs(a[:, i, c])
            better = v > best
            best = np.where(better, v, best)
            pr = np.where(better, i, pr)
        tmp = a[rows, c, :].copy()
        a[rows, c, :] = a[rows, pr, :]
        a[rows, pr, :] = tmp
        piv = a[:, c, c].copy()
        for j in range(c, 20):
            a[:, c, j] = a[:, c, j] / piv
        for i in range(10):
            if i == c:
                continue
            f = a[:, i, c].copy()
            for j in range(c + 1, 20):
                a[:, i, j] = a[:, i, j] - f * a[:, c, j]
            a[:, i, c] = 0.0
    return a[:, :, 10:]


def _poly_mul(a: np.ndarray, b: np.ndarray) -> np.ndarray:
    out = np.zeros((a.shape[0], a.shape[1] + b.shape[1] - 1))
    for i in range(a.shape[1]):
        for j in range(b.shape[1]):
            out[:, i + j] = out[:, i + j] + a[:, i] * b[:, j]
    return out


def _pad(a: np.ndarray, n: int) -> np.ndarray:
    return np.concatenate([a, np.zeros((a.shape[0], n - a.shape[1]))], axis=1)


def _hidden_variable(b: np.ndarray):
    """b [H,10,10] -> (p1 [H,8], p2 [H,8], p3 [H,7], det [H,11]), coefficients from degree 0 up. Rows k = e - z f,
    l = g - z h, m = i - z j of Nister's B(z); (x, y, 1) is proportional to (p1, p2, p3)(z) = row k x row l."""

    def row(p, q):
        rx = np.stack([b[:, p, 2], b[:, p, 1] - b[:, q, 2], b[:, p, 0] - b[:, q, 1], -b[:, q, 0]], axis=1)
        ry = np.stack([b[:, p, 5], b[:, p, 4] - b[:, q, 5], b[:, p, 3] - b[:, q, 4], -b[:, q, 3]], axis=1)
        rc = np.stack([b[:, p, 9], b[:, p, 8] - b[:, q, 9], b[:, p, 7] - b[:, q, 8], b[:, p, 6] - b[:, q, 7], -b[:, q, 6]], axis=1)
        return rx, ry, rc

    kx, ky, kc = row(4, 5)
    lx, ly, lc = row(6, 7)
    mx, my, mc = row(8, 9)
    p1 = _poly_mul(ky, lc) - _poly_mul(kc, ly)  # degree 7
    p2 = _poly_mul(kc, lx) - _poly_mul(kx, lc)  # degree 7
    p3 = _poly_mul(kx, ly) - _poly_mul(ky, lx)  # degree 6
    det = (_poly_mul(p1, mx) + _poly_mul(p2, my)) + _poly_mul(p3, mc)  # degree 10
    return p1, p2, p3, det


def _horner(coeffs: np.ndarray, deg: int, x: np.ndarray) -> np.ndarray:
    v = coeffs[:, deg].copy()
    for k in range(deg - 1, -1, -1):
        v = v * x + coeffs[:, k]
    return v


def real_roots(p: np.ndarray, n: int) -> Tuple[np.ndarray, np.ndarray]:
    """Real roots of [H,n+1] polynomials of degree n in [-R, R], R = min(1 + max|a_k / a_n|, 1e8): the roots of the (n-d)-th
    derivative split the line into intervals on which the (n-1-d)-th derivative is monotonic; every sign change is bisected until
    the midpoint stops moving. Returns (roots [H,n] ascending, count [H]). n = 10 (five-point solver), 3 (seven-point)."""
    h = p.shape[0]
    with np.errstate(all="ignore"):
        big = np.zeros(h)
        for k in range(n):
            v = np.abs(p[:, k] / p[:, n])
            big = np.where(v > big, v, big)
        rng = 1.0 + big
        rng = np.where(rng > ROOT_RANGE_CAP, ROOT_RANGE_CAP, rng)
        prev = np.zeros((h, n))
        nprev = np.zeros(h, dtype=np.int64)
        for deg in range(1, n + 1):
            s = n - deg
            d = np.zeros((h, deg + 1))
            for k in range(deg + 1):
                factor = 1.0
                for i in range(1, s + 1):
                    factor *= float(k + i)
                d[:, k] = p[:, k + s] * factor
            cur = np.zeros((h, n))
            ncur = np.zeros(h, dtype=np.int64)
            for j in range(deg):  # at most deg intervals (nprev <= deg - 1)
                active = j <= nprev
                lo = -rng if j == 0 else np.where(active, prev[:, j - 1], 0.0)
                hi = np.where(j == nprev, rng, prev[:, min(j, n - 1)])
                flo = _horner(d, deg, lo)
                fhi = _horner(d, deg, hi)
                has = active & ((flo < 0) != (fhi < 0))
                neg_lo = flo < 0
                lo = lo.copy()
                hi = hi.copy()
                running = has.copy()
                for _ in range(BISECT_ITERS):
                    mid = 0.5 * (lo + hi)
                    running = running & (mid > lo) & (mid < hi)
                    if not running.any():
                        break
                    fm = _horner(d, deg, mid)
                    same = (fm < 0) == neg_lo
                    lo = np.where(running & same, mid, lo)
                    hi = np.where(running & ~same, mid, hi)
                root = 0.5 * (lo + hi)
                idx = np.nonzero(has)[0]
                cur[idx, ncur[idx]] = root[idx]
                ncur[idx] += 1
            prev, nprev = cur, ncur
        return prev, nprev


def real_roots_deg10(p: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:
    return real_roots(p, 10)


def five_point_models(x1: np.ndarray, x2: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:
    """x1, x2 [H,5,2] normalised coordinates -> (E [H,10,3,3], count [H]): all real solutions with x2^T E x1 = 0."""
    h = x1.shape[0]
    with np.errstate(all="ignore"):
        q = np.empty((h, 5, 9))
        a, b = x1[:, :, 0], x1[:, :, 1]
        c, d = x2[:, :, 0], x2[:, :, 1]
        q[:, :, 0], q[:, :, 1], q[:, :, 2] = c * a, c * b, c
        q[:, :, 3], q[:, :, 4], q[:, :, 5] = d * a, d * b, d
        q[:, :, 6], q[:, :, 7], q[:, :, 8] = a, b, 1.0
        basis = _null_space(q)
        tail = _gauss_jordan_10x20(_constraints(basis))
        p1, p2, p3, det = _hidden_variable(tail)
        roots, count = real_roots(det, 10)
        models = np.full((h, 10, 3, 3), np.nan)
        for r in range(10):
            z = roots[:, r]
            x = _horner(p1, 7, z) / _horner(p3, 6, z)
            y = _horner(p2, 7, z) / _horner(p3, 6, z)
            e = ((x[:, None] * basis[:, 0] + y[:, None] * basis[:, 1]) + z[:, None] * basis[:, 2]) + basis[:, 3]
            models[:, r] = np.where((r < count)[:, None, None], e.reshape(h, 3, 3), np.nan)
        return models, count


def sampson_sq(e: np.ndarray, x1: np.ndarray, x2: np.ndarray) -> np.ndarray:
    """Squared Sampson error, ``gtsfm/utils/verification.py:172-220``. e [...,3,3] broadcast against x1, x2 [M,2] ->
    [..., M]."""
    with np.errstate(all="ignore"):
        a, b = x1[:, 0], x1[:, 1]
        c, d = x2[:, 0], x2[:, 1]
        g = lambda i, j: e[..., i, j][..., None]  # noqa: E731
        l2x = (g(0, 0) * a + g(0, 1) * b) + g(0, 2)
        l2y = (g(1, 0) * a + g(1, 1) * b) + g(1, 2)
        l2z = (g(2, 0) * a + g(2, 1) * b) + g(2, 2)
        l1x = (g(0, 0) * c + g(1, 0) * d) + g(2, 0)
        l1y = (g(0, 1) * c + g(1, 1) * d) + g(2, 1)
        r = (c * l2x + d * l2y) + l2z
        den = ((l2x * l2x + l2y * l2y) + l1x * l1x) + l1y * l1y
        return (r * r) / den


def seven_point_models(x1: np.ndarray, x2: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:
    """x1, x2 [H,7,2] pixel coordinates -> (F [H,3,3,3], count [H]): the real solutions of det(x F1 + F2) = 0 on the
    two-dimensional null space of the 7x9 epipolar system (Hartley & Zisserman, Multiple View Geometry, algorithm 11.x "7-point";
    what ``cv2.findFundamentalMat(FM_RANSAC)`` draws its hypotheses from, ``ransac.py:103-110``)."""
    h = x1.shape[0]
    with np.errstate(all="ignore"):
        q = np.empty((h, 7, 9))
        a, b = x1[:, :, 0], x1[:, :, 1]
        c, d = x2[:, :, 0], x2[:, :, 1]
        q[:, :, 0], q[:, :, 1], q[:, :, 2] = c * a, c * b, c
        q[:, :, 3], q[:, :, 4], q[:, :, 5] = d * a, d * b, d
        q[:, :, 6], q[:, :, 7], q[:, :, 8] = a, b, 1.0
        basis = _null_space(q)  # [H,2,9]
        e = [[np.stack([basis[:, 1, 3 * i + j], basis[:, 0, 3 * i + j]], axis=1) for j in range(3)] for i in range(3)]  # F2 + x F1
        c0 = _poly_mul(e[1][1], e[2][2]) - _poly_mul(e[1][2], e[2][1])
        c1 = _poly_mul(e[1][2], e[2][0]) - _poly_mul(e[1][0], e[2][2])
        c2 = _poly_mul(e[1][0], e[2][1]) - _poly_mul(e[1][1], e[2][0])
        det = (_poly_mul(e[0][0], c0) + _poly_mul(e[0][1], c1)) + _poly_mul(e[0][2], c2)  # degree 3
        roots, count = real_roots(det, 3)
        models = np.full((h, 3, 3, 3), np.nan)
        for r in range(3):
            x = roots[:, r]
            f = x[:, None] * basis[:, 0] + basis[:, 1]
            models[:, r] = np.where((r < count)[:, None, None], f.reshape(h, 3, 3), np.nan)
        return models, count


def epipolar_distance_sq_max(f: np.ndarray, x1: np.ndarray, x2: np.ndarray) -> np.ndarray:
    """The residual of OpenCV's fundamental-matrix RANSAC: the larger of the two squared point-to-epipolar-line distances
    (calib3d/src/fundam.cpp, FMEstimatorCallback::computeError). f [...,3,3], x1, x2 [M,2] pixels -> [..., M]."""
    with np.errstate(all="ignore"):
        a, b = x1[:, 0], x1[:, 1]
        c, d = x2[:, 0], x2[:, 1]
        g = lambda i, j: f[..., i, j][..., None]  # noqa: E731
        l2x = (g(0, 0) * a + g(0, 1) * b) + g(0, 2)
        l2y = (g(1, 0) * a + g(1, 1) * b) + g(1, 2)
        l2z = (g(2, 0) * a + g(2, 1) * b) + g(2, 2)
        l1x = (g(0, 0) * c + g(1, 0) * d) + g(2, 0)
        l1y = (g(0, 1) * c + g(1, 1) * d) + g(2, 1)
        r = (c * l2x + d * l2y) + l2z
        d2 = (r * r) / (l2x * l2x + l2y * l2y)
        d1 = (r * r) / (l1x * l1x + l1y * l1y)
        return np.where(d1 > d2, d1, d2)


def normalize_pinhole(xy: np.ndarray, fx: float, fy: float, cx: float, cy: float) -> np.ndarray:
    """``Cal3Bundler.calibrate`` without distortion (``gtsfm/utils/features.py:41-51``): float64 ((u - cx)/fx, (v - cy)/fy)."""
    xy = np.asarray(xy, dtype=np.float64)
    return np.stack([(xy[:, 0] - cx) / fx, (xy[:, 1] - cy) / fy], axis=1)


def _stop_after(rounds_done: int, inliers: int, m: int, size: int = 5) -> bool:
    w = float(inliers) / float(m)
    ws = w
    for _ in range(size - 1):
        ws = ws * w
    q = 1.0 - ws
    p256 = q
    for _ in range(8):
        p256 = p256 * p256
    acc = p256
    for _ in range(rounds_done - 1):
        acc = acc * p256
    return acc <= 1.0 - SUCCESS_PROB


def _ransac(x1: np.ndarray, x2: np.ndarray, thr2: float, seed: int, solver, size: int, max_models: int, error) -> Dict[str, object]:
    m = x1.shape[0]
    if m < size:
        return {"model": None, "mask": np.zeros(m, dtype=bool), "hypotheses": 0, "winner": None}
    best_cost, best_count, best_e, winner = np.inf, 0, None, None

    def one_round(hyp, idx):
        models, nroots = solver(x1[idx], x2[idx])
        with np.errstate(all="ignore"):
            err = error(models, x1, x2)  # [ROUND,max_models,M]
            inl = err < thr2  # NaN errors are outliers
            # MSAC cost, summed left to right (np.cumsum accumulates sequentially; np.sum would add pairwise)
            cost = np.cumsum(np.where(inl, err, thr2), axis=-1)[..., -1]
        cost = np.where(np.arange(max_models)[None, :] < nroots[:, None], cost, np.inf)  # roots that do not exist
        flat = cost.reshape(-1)
        k = int(np.argmin(flat))  # first minimum = smallest (hypothesis, root)
        return float(flat[k]), inl.reshape(-1, m)[k].copy(), models.reshape(-1, 3, 3)[k].copy(), (int(hyp[k // max_models]), k % max_models)

    done = 0
    for rnd in range(MAX_ROUNDS):
        hyp = np.arange(rnd * ROUND, (rnd + 1) * ROUND)
        cost, inl, model, who = one_round(hyp, sample_indices(seed, hyp, m, size))
        if cost < best_cost:
            best_cost, best_count, best_e, winner = cost, int(inl.sum()), model, who
        done = rnd + 1
        if best_count > 0 and _stop_after(done, best_count, m, size):
            break
    if best_e is None:
        return {"model": None, "mask": np.zeros(m, dtype=bool), "hypotheses": done * ROUND, "winner": None}
    with np.errstate(all="ignore"):
        mask = error(best_e, x1, x2) < thr2
    hypotheses = done * ROUND
    inliers = np.flatnonzero(mask)
    if inliers.shape[0] >= size + 1:
        # local optimisation, LO-RANSAC's inner sampling with minimal samples: one more round whose samples come from the
        # inliers of the winner (hypothesis numbers MAX_ROUNDS * ROUND ...), scored on all matches as before
        hyp = np.arange(MAX_ROUNDS * ROUND, (MAX_ROUNDS + 1) * ROUND)
        cost, inl, model, who = one_round(hyp, inliers[sample_indices(seed, hyp, inliers.shape[0], size)])
        if cost < best_cost:
            best_cost, best_e, winner, mask = cost, model, who, inl
        hypotheses += ROUND
    return {"model": best_e, "mask": mask, "hypotheses": hypotheses, "winner": winner, "cost": best_cost}


def ransac_essential(x1: np.ndarray, x2: np.ndarray, threshold: float, seed: int = 0) -> Dict[str, object]:
    """x1, x2 [M,2] normalised matched coordinates, threshold in normalised units (px / fx) -> best model.

    Returns {"E" [3,3] or None, "mask" [M] bool, "hypotheses" int, "winner" (hypothesis, root)}."""
    res = _ransac(x1, x2, threshold * threshold, seed, five_point_models, 5, 10, sampson_sq)
    res["E"] = res.pop("model")
    return res


def ransac_fundamental(x1: np.ndarray, x2: np.ndarray, threshold_px: float, seed: int = 0) -> Dict[str, object]:
    """x1, x2 [M,2] matched PIXEL coordinates -> best fundamental matrix ("F"), seven-point samples, OpenCV's residual."""
    res = _ransac(x1, x2, threshold_px * threshold_px, seed, seven_point_models, 7, 3, epipolar_distance_sq_max)
    res["F"] = res.pop("model")
    return res


# ---------------------------------------------------------------------------------------------------------------------
# pose recovery


def _jacobi_eigen_sym3(s: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:
    """Cyclic Jacobi on a symmetric 3x3, fixed sweep count -> (eigenvalues [3], eigenvectors as columns [3,3])."""
    a = s.copy()
    v = np.eye(3)
    for _ in range(JACOBI_SWEEPS):
        for p, q in ((0, 1), (0, 2), (1, 2)):
            if a[p, q] == 0.0:
                continue
            with np.errstate(over="ignore"):  # a vanishing off-diagonal makes tau^2 overflow: t becomes 0, as it should
                tau = (a[q, q] - a[p, p]) / (2.0 * a[p, q])
                t = (1.0 if tau >= 0 else -1.0) / (abs(tau) + np.sqrt(1.0 + tau * tau))
            c = 1.0 / np.sqrt(1.0 + t * t)
            sn = t * c
            app, aqq, apq = a[p, p], a[q, q], a[p, q]
            a[p, p] = app - t * apq
            a[q, q] = aqq + t * apq
            a[p, q] = a[q, p] = 0.0
            r = 3 - p - q
            arp, arq = a[r, p], a[r, q]
            a[r, p] = a[p, r] = c * arp - sn * arq
            a[r, q] = a[q, r] = sn * arp + c * arq
            for k in range(3):
                vkp, vkq = v[k, p], v[k, q]
                v[k, p] = c * vkp - sn * vkq
                v[k, q] = sn * vkp + c * vkq
    return np.array([a[0, 0], a[1, 1], a[2, 2]]), v


def _cross(a, b):
    return np.array([a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0]])


def decompose_essential(e: np.ndarray) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
    """E -> (R1, R2, t) as ``cv.decomposeEssentialMat``: E = U diag(s, s, 0) V^T with det U = det V = +1, R1 = U W V^T,
    R2 = U W^T V^T, t = U[:, 2]."""
    lam, v = _jacobi_eigen_sym3(_ata(e))
    order = sorted(range(3), key=lambda i: (-lam[i], i))
    v0, v1 = v[:, order[0]], v[:, order[1]]
    u0 = _matvec(e, v0)
    u0 = u0 / np.sqrt((u0[0] * u0[0] + u0[1] * u0[1]) + u0[2] * u0[2])
    u1 = _matvec(e, v1)
    u1 = u1 - ((u0[0] * u1[0] + u0[1] * u1[1]) + u0[2] * u1[2]) * u0
    u1 = u1 / np.sqrt((u1[0] * u1[0] + u1[1] * u1[1]) + u1[2] * u1[2])
    u2 = _cross(u0, u1)
    v2 = _cross(v0, v1)
    u = np.stack([u0, u1, u2], axis=1)
    vm = np.stack([v0, v1, v2], axis=1)
    w = np.array([[0.0, -1.0, 0.0], [1.0, 0.0, 0.0], [0.0, 0.0, 1.0]])
    return _mat3(_mat3(u, w), vm.T), _mat3(_mat3(u, w.T), vm.T), u2


def _ata(e):
    out = np.zeros((3, 3))
    for i in range(3):
        for j in range(3):
            out[i, j] = (e[0, i] * e[0, j] + e[1, i] * e[1, j]) + e[2, i] * e[2, j]
    return out


def _matvec(a, x):
    return np.array([(a[i, 0] * x[0] + a[i, 1] * x[1]) + a[i, 2] * x[2] for i in range(3)])


def _mat3(a, b):
    out = np.zeros((3, 3))
    for i in range(3):
        for j in range(3):
            out[i, j] = (a[i, 0] * b[0, j] + a[i, 1] * b[1, j]) + a[i, 2] * b[2, j]
    return out


def cheirality_count(r: np.ndarray, t: np.ndarray, x1: np.ndarray, x2: np.ndarray) -> int:
    """Number of correspondences with both depths in (0, DEPTH_LIMIT): least-squares (l1, l2) of l1 R x1 + t = l2 x2."""
    with np.errstate(all="ignore"):
        ax = (r[0, 0] * x1[:, 0] + r[0, 1] * x1[:, 1]) + r[0, 2]
        ay = (r[1, 0] * x1[:, 0] + r[1, 1] * x1[:, 1]) + r[1, 2]
        az = (r[2, 0] * x1[:, 0] + r[2, 1] * x1[:, 1]) + r[2, 2]
        bx, by = x2[:, 0], x2[:, 1]
        aa = (ax * ax + ay * ay) + az * az
        bb = (bx * bx + by * by) + 1.0
        ab = (ax * bx + ay * by) + az
        at = (ax * t[0] + ay * t[1]) + az * t[2]
        bt = (bx * t[0] + by * t[1]) + t[2]
        det = aa * bb - ab * ab
        l1 = (ab * bt - at * bb) / det
        l2 = (aa * bt - ab * at) / det
        good = (l1 > 0) & (l2 > 0) & (l1 < DEPTH_LIMIT) & (l2 < DEPTH_LIMIT)
        return int(good.sum())


def recover_pose(e: np.ndarray, x1: np.ndarray, x2: np.ndarray) -> Tuple[np.ndarray, np.ndarray, List[int]]:
    """``cv.recoverPose``'s selection order: the first of (R1,t), (R2,t), (R1,-t), (R2,-t) whose count is >= the others."""
    r1, r2, t = decompose_essential(e)
    cands = [(r1, t), (r2, t), (r1, -t), (r2, -t)]
    good = [cheirality_count(r, tt, x1, x2) for r, tt in cands]
    for k in range(4):
        if all(good[k] >= good[j] for j in range(4)):
            return cands[k][0], cands[k][1], good
    raise AssertionError("unreachable")


# ---------------------------------------------------------------------------------------------------------------------
# final polish: Gauss-Newton on the squared Sampson error of the inliers, over the pose (what OpenCV's USAC does last)
POLISH_ITERS = 6
POLISH_STEP = 1.0e-6
REDUCE_LANES = 256  # the device sums per thread (i = tid, tid + 256, ...) and then halves the 256 partial sums 8 times


def _reduce_like_the_device(c: np.ndarray) -> np.ndarray:
    """c [M, C] per-match contributions -> [C] totals, added in the order the workgroup adds them."""
    m = c.shape[0]
    rows = -(-m // REDUCE_LANES)
    pad = np.zeros((rows * REDUCE_LANES, c.shape[1]))
    pad[:m] = c
    part = np.zeros((REDUCE_LANES, c.shape[1]))
    for k in range(rows):  # acc = acc + c[tid + 256 k]
        part = part + pad[k * REDUCE_LANES : (k + 1) * REDUCE_LANES]
    step = REDUCE_LANES // 2
    while step > 0:
        part[:step] = part[:step] + part[step : 2 * step]
        step //= 2
    return part[0]


def _essential_from_pose(r: np.ndarray, t: np.ndarray) -> np.ndarray:
    sk = np.array([[0.0, -t[2], t[1]], [t[2], 0.0, -t[0]], [-t[1], t[0], 0.0]])
    return _mat3(sk, r)


def _tangent_basis(t: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:
    k = 0
    for i in (1, 2):  # axis of the smallest |t_k|, first minimum
        if abs(t[i]) < abs(t[k]):
            k = i
    a = np.zeros(3)
    a[k] = 1.0
    b1 = _cross(t, a)
    b1 = b1 / np.sqrt((b1[0] * b1[0] + b1[1] * b1[1]) + b1[2] * b1[2])
    return b1, _cross(t, b1)


def _perturb_pose(r: np.ndarray, t: np.ndarray, b1: np.ndarray, b2: np.ndarray, d: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:
    """R <- R cayley(d[0:3]) (a rotation without trigonometry: (I - W/2)^-1 (I + W/2), W = [w]x); t <- unit(t + d3 b1 + d4 b2)."""
    wx, wy, wz = 0.5 * d[0], 0.5 * d[1], 0.5 * d[2]
    a = np.array([[1.0, wz, -wy], [-wz, 1.0, wx], [wy, -wx, 1.0]])  # I - W/2
    b = np.array([[1.0, -wz, wy], [wz, 1.0, -wx], [-wy, wx, 1.0]])  # I + W/2
    adj = np.array([
        [a[1, 1] * a[2, 2] - a[1, 2] * a[2, 1], a[0, 2] * a[2, 1] - a[0, 1] * a[2, 2], a[0, 1] * a[1, 2] - a[0, 2] * a[1, 1]],
        [a[1, 2] * a[2, 0] - a[1, 0] * a[2, 2], a[0, 0] * a[2, 2] - a[0, 2] * a[2, 0], a[0, 2] * a[1, 0] - a[0, 0] * a[1, 2]],
        [a[1, 0] * a[2, 1] - a[1, 1] * a[2, 0], a[0, 1] * a[2, 0] - a[0, 0] * a[2, 1], a[0, 0] * a[1, 1] - a[0, 1] * a[1, 0]],
    ])
    det = (a[0, 0] * adj[0, 0] + a[0, 1] * adj[1, 0]) + a[0, 2] * adj[2, 0]
    q = _mat3(adj / det, b)
    tn = np.array([(t[i] + d[3] * b1[i]) + d[4] * b2[i] for i in range(3)])
    tn = tn / np.sqrt((tn[0] * tn[0] + tn[1] * tn[1]) + tn[2] * tn[2])
    return _mat3(r, q), tn


def _signed_sampson(e: np.ndarray, x1: np.ndarray, x2: np.ndarray) -> np.ndarray:
    a, b = x1[:, 0], x1[:, 1]
    c, d = x2[:, 0], x2[:, 1]
    l2x = (e[0, 0] * a + e[0, 1] * b) + e[0, 2]
    l2y = (e[1, 0] * a + e[1, 1] * b) + e[1, 2]
    l2z = (e[2, 0] * a + e[2, 1] * b) + e[2, 2]
    l1x = (e[0, 0] * c + e[1, 0] * d) + e[2, 0]
    l1y = (e[0, 1] * c + e[1, 1] * d) + e[2, 1]
    r = (c * l2x + d * l2y) + l2z
    den = ((l2x * l2x + l2y * l2y) + l1x * l1x) + l1y * l1y
    return r / np.sqrt(den)


def _solve5(h: np.ndarray, g: np.ndarray) -> np.ndarray:
    """h d = -g by Gaussian elimination with row pivoting (first maximum)."""
    a = np.concatenate([h, -g[:, None]], axis=1)
    for c in range(5):
        pr = c
        for i in range(c + 1, 5):
            if abs(a[i, c]) > abs(a[pr, c]):
                pr = i
        a[[c, pr]] = a[[pr, c]]
        for j in range(5, c - 1, -1):
            a[c, j] = a[c, j] / a[c, c]
        for i in range(5):
            if i != c:
                f = a[i, c]
                for j in range(c, 6):
                    a[i, j] = a[i, j] - f * a[c, j]
    return a[:, 5].copy()


def msac_cost(e: np.ndarray, x1: np.ndarray, x2: np.ndarray, thr2: float) -> float:
    with np.errstate(all="ignore"):
        err = sampson_sq(e, x1, x2)
        return float(np.cumsum(np.where(err < thr2, err, thr2))[-1])


def polish_pose(r: np.ndarray, t: np.ndarray, x1: np.ndarray, x2: np.ndarray, mask: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:
    """POLISH_ITERS Gauss-Newton steps on sum over the inliers of the (signed) Sampson distance squared, in the five
    parameters (rotation vector through the Cayley map, two tangent directions of the unit translation), forward-difference
    Jacobian."""
    with np.errstate(all="ignore"):
        for _ in range(POLISH_ITERS):
            b1, b2 = _tangent_basis(t)
            r0 = _signed_sampson(_essential_from_pose(r, t), x1, x2)
            cols = []
            for k in range(5):
                d = np.zeros(5)
                d[k] = POLISH_STEP
                rk, tk = _perturb_pose(r, t, b1, b2, d)
                cols.append((_signed_sampson(_essential_from_pose(rk, tk), x1, x2) - r0) / POLISH_STEP)
            contrib = []
            for a in range(5):
                for b in range(a, 5):
                    contrib.append(cols[a] * cols[b])
            for a in range(5):
                contrib.append(cols[a] * r0)
            c = np.where(mask[:, None], np.stack(contrib, axis=1), 0.0)
            tot = _reduce_like_the_device(c)
            h = np.zeros((5, 5))
            n = 0
            for a in range(5):
                for b in range(a, 5):
                    h[a, b] = h[b, a] = tot[n]
                    n += 1
            g = tot[15:20]
            for a in range(5):
                h[a, a] = h[a, a] + 1.0e-12 * (1.0 + h[a, a])
            r, t = _perturb_pose(r, t, b1, b2, _solve5(h, g))
    return r, t


def verify(
    coords_i1: np.ndarray,
    coords_i2: np.ndarray,
    match_indices: np.ndarray,
    intrinsics_i1: Tuple[float, float, float, float],
    intrinsics_i2: Tuple[float, float, float, float],
    estimation_threshold_px: float,
    seed: int = 0,
    use_intrinsics_in_verification: bool = True,
) -> Dict[str, object]:
    """The whole of ``OpencvVerifierBase.verify`` (``opencv_verifier_base.py:47-111``). intrinsics = (fx, fy, cx, cy).
    ``use_intrinsics_in_verification``: essential matrix on normalised coordinates with threshold px / max(fx) (:74-90), else
    fundamental matrix on pixel coordinates and E = K2^T F K1 (:91-97, ``verification.py:99-112``). Returns R, t (None on
    failure), v_corr_idxs, inlier_ratio, E, mask."""
    match_indices = np.asarray(match_indices)
    failure = {"R": None, "t": None, "v_corr_idxs": np.array([], dtype=np.uint64), "inlier_ratio": 0.0, "E": None,
               "mask": np.zeros(match_indices.shape[0] if match_indices.ndim == 2 else 0, dtype=bool), "hypotheses": 0}
    need = 6 if use_intrinsics_in_verification else 8  # NUM_MATCHES_REQ_E_MATRIX = 5 plus the "< 6" guard at :79; NUM_MATCHES_REQ_F_MATRIX = 8
    if match_indices.ndim != 2 or match_indices.shape[0] < need:
        return failure
    n1 = normalize_pinhole(coords_i1, *intrinsics_i1)
    n2 = normalize_pinhole(coords_i2, *intrinsics_i2)
    x1 = n1[match_indices[:, 0].astype(np.int64)]
    x2 = n2[match_indices[:, 1].astype(np.int64)]
    if use_intrinsics_in_verification:
        fx = max(intrinsics_i1[0], intrinsics_i2[0])
        res = ransac_essential(x1, x2, estimation_threshold_px / fx, seed)
        essential = res["E"]
    else:
        p1 = np.asarray(coords_i1, dtype=np.float64)[match_indices[:, 0].astype(np.int64)]
        p2 = np.asarray(coords_i2, dtype=np.float64)[match_indices[:, 1].astype(np.int64)]
        res = ransac_fundamental(p1, p2, estimation_threshold_px, seed)
        essential = None
        if res["F"] is not None:
            k1 = np.array([[intrinsics_i1[0], 0.0, intrinsics_i1[2]], [0.0, intrinsics_i1[1], intrinsics_i1[3]], [0.0, 0.0, 1.0]])
            k2 = np.array([[intrinsics_i2[0], 0.0, intrinsics_i2[2]], [0.0, intrinsics_i2[1], intrinsics_i2[3]], [0.0, 0.0, 1.0]])
            essential = _mat3(_mat3(k2.T, res["F"]), k1)
    if essential is None:
        failure["hypotheses"] = res["hypotheses"]
        return failure
    mask = res["mask"]
    if not mask.any():
        failure["hypotheses"] = res["hypotheses"]
        return failure
    r, t, good = recover_pose(essential, x1[mask], x2[mask])
    polished = False
    if use_intrinsics_in_verification:
        # final polish of the pose on the winner's inliers; kept only if the MSAC cost over ALL matches goes down
        thr2 = (estimation_threshold_px / fx) * (estimation_threshold_px / fx)
        r2, t2 = polish_pose(r, t, x1, x2, mask)
        e2 = _essential_from_pose(r2, t2)
        if msac_cost(e2, x1, x2, thr2) < res["cost"]:
            with np.errstate(all="ignore"):
                r, t, essential, mask, polished = r2, t2, e2, sampson_sq(e2, x1, x2) < thr2, True
    return {"R": r, "t": t, "v_corr_idxs": match_indices[mask], "inlier_ratio": float(mask.mean()), "E": essential,
            "F": res.get("F"), "mask": mask, "hypotheses": res["hypotheses"], "cheirality": good, "winner": res["winner"], "polished": polished}
