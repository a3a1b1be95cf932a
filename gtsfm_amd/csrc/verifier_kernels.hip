// Verifier stage on the device: what OpencvVerifierBase.verify does per image pair with
// use_intrinsics_in_verification=True (gtsfm/frontend/verifier/opencv_verifier_base.py:47-111, ransac.py:52-84, called from
// gtsfm/two_view_estimator.py:391-397) -- normalise the matched keypoints, RANSAC over five-point essential-matrix
// hypotheses with the squared Sampson error (gtsfm/utils/verification.py:172-220), cv.recoverPose's cheirality choice --
// for a whole batch of pairs in two launches, keypoints and matches never leaving HBM.
//
// PARITY UNPINNED: cv2.findEssentialMat / cv.recoverPose are OpenCV (absent; its USAC sampler is not reproducible). The
// published mathematics is restated (Nister's five-point solver, PAMI 2004) with a counter-based sampler; every double
// operation below follows oracle/verifier_oracle.py in the same order (this file is compiled with -ffp-contract=off and
// honours NaNs: a degenerate sample turns into NaN models that count zero inliers), so inlier masks are compared bit for bit.
//
// Mapping: one workgroup of 256 threads per pair; a round = 256 hypotheses, one per thread (sample, solve, score the <= 10
// real solutions against all matches of the pair, MSAC cost); rounds stop by the (1 - w^5)^n <= 1 - p rule, at most 4, and one
// more round samples from the inliers of the winner (local optimisation). The solver's
// 10x20 elimination matrix lives in per-thread scratch: the stage is latency-bound double-precision scalar work, a few
// hundred microseconds per round, against ~2 ms of matcher time per pair -- no MFMA, no LDS tiling worth having.

#include <math.h>

#include "../../include/gtsfm_amd.h"
#include "common.h"

#define VF_ROUND 256
#define VF_MAX_ROUNDS 4
#define VF_ROOT_RANGE_CAP 1.0e8
#define VF_BISECT_ITERS 128
#define VF_JACOBI_SWEEPS 8
#define VF_DEPTH_LIMIT 50.0
#define VF_SUCCESS_PROB 0.999999
#define VF_POLISH_ITERS 6
#define VF_POLISH_STEP 1.0e-6

__constant__ int VF_LIN_LIN[4][4] = {{0, 1, 2, 3}, {1, 4, 5, 6}, {2, 5, 7, 8}, {3, 6, 8, 9}};
__constant__ int VF_QUAD_LIN[10][4] = {{0, 2, 4, 5},     {2, 3, 8, 9},     {4, 8, 10, 11},   {5, 9, 11, 12},   {3, 1, 6, 7},
                                        {8, 6, 13, 14},   {9, 7, 14, 15},   {10, 13, 16, 17}, {11, 14, 17, 18}, {12, 15, 18, 19}};

__device__ __forceinline__ unsigned long long vf_splitmix64(unsigned long long x) {
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

// `size` (5 or 7) distinct match indices of hypothesis `hyp` (oracle: sample_indices).
__device__ void vf_sample(unsigned long long seed, unsigned long long hyp, int m, int* out, int size) {
    unsigned long long attempt = 0;
    for (int k = 0; k < size; ++k) {
        for (;;) {
            const bool exhausted = attempt >= 64;
            int draw = (int)(vf_splitmix64(seed ^ vf_splitmix64((hyp << 8) | attempt)) % (unsigned long long)m);
            bool clash = false;
            for (int j = 0; j < k; ++j) clash |= out[j] == draw;
            if (exhausted) {
                for (draw = 0;; ++draw) {
                    bool used = false;
                    for (int j = 0; j < k; ++j) used |= out[j] == draw;
                    if (!used) break;
                }
                clash = false;
            } else {
                ++attempt;
            }
            if (!clash) {
                out[k] = draw;
                break;
            }
        }
    }
}

// Null space of the NRx9 epipolar system (NR = 5: essential, 7: fundamental): Gauss-Jordan with complete pivoting
// (oracle: _null_space).
template <int NR>
__device__ void vf_null_space(double a[NR][9], double basis[9 - NR][9]) {
    int perm[9];
    for (int j = 0; j < 9; ++j) perm[j] = j;
    for (int r = 0; r < NR; ++r) {
        double best = -1.0;
        int pr = r, pc = r;
        for (int i = r; i < NR; ++i)
            for (int j = r; j < 9; ++j) {
                const double v = fabs(a[i][j]);
                if (v > best) {
                    best = v;
                    pr = i;
                    pc = j;
                }
            }
        for (int j = 0; j < 9; ++j) {
            const double t = a[r][j];
            a[r][j] = a[pr][j];
            a[pr][j] = t;
        }
        for (int i = 0; i < NR; ++i) {
            const double t = a[i][r];
            a[i][r] = a[i][pc];
            a[i][pc] = t;
        }
        {
            const int t = perm[r];
            perm[r] = perm[pc];
            perm[pc] = t;
        }
        const double piv = a[r][r];
        for (int j = r; j < 9; ++j) a[r][j] = a[r][j] / piv;
        for (int i = 0; i < NR; ++i) {
            if (i == r) continue;
            const double f = a[i][r];
            for (int j = r + 1; j < 9; ++j) a[i][j] = a[i][j] - f * a[r][j];
            a[i][r] = 0.0;
        }
    }
    for (int k = 0; k < 9 - NR; ++k) {
        for (int j = 0; j < 9; ++j) basis[k][j] = 0.0;
        basis[k][perm[NR + k]] = 1.0;
        for (int i = 0; i < NR; ++i) basis[k][perm[i]] = -a[i][NR + k];
    }
}

__device__ __forceinline__ void vf_mul_lin_lin(const double* p, const double* q, double* out) {
    for (int a = 0; a < 4; ++a)
        for (int b = 0; b < 4; ++b) {
            const int k = VF_LIN_LIN[a][b];
            out[k] = out[k] + p[a] * q[b];
        }
}

__device__ __forceinline__ void vf_mul_quad_lin(const double* p, const double* q, double* out) {
    for (int a = 0; a < 10; ++a)
        for (int b = 0; b < 4; ++b) {
            const int k = VF_QUAD_LIN[a][b];
            out[k] = out[k] + p[a] * q[b];
        }
}

// The ten cubic constraints on E = x X + y Y + z Z + W (oracle: _constraints): rows 0..8 (E E^T - tr(E E^T)/2 I) E, row 9 det E.
__device__ void vf_constraints(const double basis[4][9], double m[10][20]) {
    double e[3][3][4];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j)
            for (int c = 0; c < 4; ++c) e[i][j][c] = basis[c][3 * i + j];
    double eet[6][10];  // (0,0) (0,1) (0,2) (1,1) (1,2) (2,2)
    const int slot[3][3] = {{0, 1, 2}, {1, 3, 4}, {2, 4, 5}};
    for (int i = 0; i < 3; ++i)
        for (int j = i; j < 3; ++j) {
            double* acc = eet[slot[i][j]];
            for (int k = 0; k < 10; ++k) acc[k] = 0.0;
            for (int k = 0; k < 3; ++k) vf_mul_lin_lin(e[i][k], e[j][k], acc);
        }
    double lam_diag[3][10];
    for (int k = 0; k < 10; ++k) {
        const double half_trace = ((eet[0][k] + eet[3][k]) + eet[5][k]) * 0.5;
        lam_diag[0][k] = eet[0][k] - half_trace;
        lam_diag[1][k] = eet[3][k] - half_trace;
        lam_diag[2][k] = eet[5][k] - half_trace;
    }
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            double* acc = m[3 * i + j];
            for (int k = 0; k < 20; ++k) acc[k] = 0.0;
            for (int k = 0; k < 3; ++k) vf_mul_quad_lin(i == k ? lam_diag[i] : eet[slot[i][k]], e[k][j], acc);
        }
    double c[3][10];
    const int ma[3][4][2] = {{{1, 1}, {2, 2}, {1, 2}, {2, 1}}, {{1, 2}, {2, 0}, {1, 0}, {2, 2}}, {{1, 0}, {2, 1}, {1, 1}, {2, 0}}};
    for (int n = 0; n < 3; ++n) {
        double p[10], q[10];
        for (int k = 0; k < 10; ++k) p[k] = q[k] = 0.0;
        vf_mul_lin_lin(e[ma[n][0][0]][ma[n][0][1]], e[ma[n][1][0]][ma[n][1][1]], p);
        vf_mul_lin_lin(e[ma[n][2][0]][ma[n][2][1]], e[ma[n][3][0]][ma[n][3][1]], q);
        for (int k = 0; k < 10; ++k) c[n][k] = p[k] - q[k];
    }
    double* acc = m[9];
    for (int k = 0; k < 20; ++k) acc[k] = 0.0;
    vf_mul_quad_lin(c[0], e[0][0], acc);
    vf_mul_quad_lin(c[1], e[0][1], acc);
    vf_mul_quad_lin(c[2], e[0][2], acc);
}

// Reduced row echelon form on the ten leading columns, row pivoting (oracle: _gauss_jordan_10x20).
__device__ void vf_gauss_jordan(double a[10][20]) {
    for (int c = 0; c < 10; ++c) {
        double best = -1.0;
        int pr = c;
        for (int i = c; i < 10; ++i) {
            const double v = fabs(a[i][c]);
            if (v > best) {
                best = v;
                pr = i;
            }
        }
        for (int j = 0; j < 20; ++j) {
            const double t = a[c][j];
            a[c][j] = a[pr][j];
            a[pr][j] = t;
        }
        const double piv = a[c][c];
        for (int j = c; j < 20; ++j) a[c][j] = a[c][j] / piv;
        for (int i = 0; i < 10; ++i) {
            if (i == c) continue;
            const double f = a[i][c];
            for (int j = c + 1; j < 20; ++j) a[i][j] = a[i][j] - f * a[c][j];
            a[i][c] = 0.0;
        }
    }
}

__device__ __forceinline__ void vf_poly_mul(const double* a, int na, const double* b, int nb, double* out) {
    for (int k = 0; k < na + nb - 1; ++k) out[k] = 0.0;
    for (int i = 0; i < na; ++i)
        for (int j = 0; j < nb; ++j) out[i + j] = out[i + j] + a[i] * b[j];
}

__device__ __forceinline__ double vf_horner(const double* c, int deg, double x) {
    double v = c[deg];
    for (int k = deg - 1; k >= 0; --k) v = v * x + c[k];
    return v;
}

// Rows k, l, m of Nister's B(z) from the eliminated tail, then (p1, p2, p3) = row k x row l and det (oracle: _hidden_variable).
__device__ void vf_hidden_variable(const double a[10][20], double p1[8], double p2[8], double p3[7], double det[11]) {
    double rx[3][4], ry[3][4], rc[3][5];
    for (int n = 0; n < 3; ++n) {
        const double* bp = &a[4 + 2 * n][10];
        const double* bq = &a[5 + 2 * n][10];
        rx[n][0] = bp[2], rx[n][1] = bp[1] - bq[2], rx[n][2] = bp[0] - bq[1], rx[n][3] = -bq[0];
        ry[n][0] = bp[5], ry[n][1] = bp[4] - bq[5], ry[n][2] = bp[3] - bq[4], ry[n][3] = -bq[3];
        rc[n][0] = bp[9], rc[n][1] = bp[8] - bq[9], rc[n][2] = bp[7] - bq[8], rc[n][3] = bp[6] - bq[7], rc[n][4] = -bq[6];
    }
    double u[11], v[11], w[11];
    vf_poly_mul(ry[0], 4, rc[1], 5, u);
    vf_poly_mul(rc[0], 5, ry[1], 4, v);
    for (int k = 0; k < 8; ++k) p1[k] = u[k] - v[k];
    vf_poly_mul(rc[0], 5, rx[1], 4, u);
    vf_poly_mul(rx[0], 4, rc[1], 5, v);
    for (int k = 0; k < 8; ++k) p2[k] = u[k] - v[k];
    vf_poly_mul(rx[0], 4, ry[1], 4, u);
    vf_poly_mul(ry[0], 4, rx[1], 4, v);
    for (int k = 0; k < 7; ++k) p3[k] = u[k] - v[k];
    vf_poly_mul(p1, 8, rx[2], 4, u);
    vf_poly_mul(p2, 8, ry[2], 4, v);
    vf_poly_mul(p3, 7, rc[2], 5, w);
    for (int k = 0; k < 11; ++k) det[k] = (u[k] + v[k]) + w[k];
}

// Real roots of a degree-n polynomial (n <= 10) in [-R, R] through the chain of its derivatives (oracle: real_roots).
__device__ int vf_real_roots(const double* p, int n, double* roots) {
    double big = 0.0;
    for (int k = 0; k < n; ++k) {
        const double v = fabs(p[k] / p[n]);
        if (v > big) big = v;
    }
    double rng = 1.0 + big;
    if (rng > VF_ROOT_RANGE_CAP) rng = VF_ROOT_RANGE_CAP;
    double prev[10], cur[10], d[11];
    int nprev = 0;
    for (int deg = 1; deg <= n; ++deg) {
        const int s = n - deg;
        for (int k = 0; k <= deg; ++k) {
            double factor = 1.0;
            for (int i = 1; i <= s; ++i) factor *= (double)(k + i);
            d[k] = p[k + s] * factor;
        }
        int ncur = 0;
        for (int j = 0; j <= nprev; ++j) {
            double lo = j == 0 ? -rng : prev[j - 1];
            double hi = j == nprev ? rng : prev[j];
            const double flo = vf_horner(d, deg, lo), fhi = vf_horner(d, deg, hi);
            if ((flo < 0) == (fhi < 0)) continue;
            const bool neg_lo = flo < 0;
            for (int it = 0; it < VF_BISECT_ITERS; ++it) {
                const double mid = 0.5 * (lo + hi);
                if (!(mid > lo && mid < hi)) break;
                const double fm = vf_horner(d, deg, mid);
                if ((fm < 0) == neg_lo)
                    lo = mid;
                else
                    hi = mid;
            }
            cur[ncur++] = 0.5 * (lo + hi);
        }
        for (int j = 0; j < ncur; ++j) prev[j] = cur[j];
        nprev = ncur;
    }
    for (int j = 0; j < nprev; ++j) roots[j] = prev[j];
    return nprev;
}

// The residual of OpenCV's fundamental-matrix RANSAC: the larger squared point-to-epipolar-line distance of the two images
// (oracle: epipolar_distance_sq_max).
__device__ __forceinline__ double vf_epipolar_sq_max(const double* f, double a, double b, double c, double d) {
    const double l2x = (f[0] * a + f[1] * b) + f[2];
    const double l2y = (f[3] * a + f[4] * b) + f[5];
    const double l2z = (f[6] * a + f[7] * b) + f[8];
    const double l1x = (f[0] * c + f[3] * d) + f[6];
    const double l1y = (f[1] * c + f[4] * d) + f[7];
    const double r = (c * l2x + d * l2y) + l2z;
    const double d2 = (r * r) / (l2x * l2x + l2y * l2y);
    const double d1 = (r * r) / (l1x * l1x + l1y * l1y);
    return d1 > d2 ? d1 : d2;
}

// Seven-point solver: the cubic det(x F1 + F2) on the two-dimensional null space (oracle: seven_point_models).
__device__ void vf_seven_point_cubic(const double basis[2][9], double det[4]) {
    double e[3][3][2];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) e[i][j][0] = basis[1][3 * i + j], e[i][j][1] = basis[0][3 * i + j];
    double c[3][3], u[3], v[3], t0[4], t1[4], t2[4];
    vf_poly_mul(e[1][1], 2, e[2][2], 2, u);
    vf_poly_mul(e[1][2], 2, e[2][1], 2, v);
    for (int k = 0; k < 3; ++k) c[0][k] = u[k] - v[k];
    vf_poly_mul(e[1][2], 2, e[2][0], 2, u);
    vf_poly_mul(e[1][0], 2, e[2][2], 2, v);
    for (int k = 0; k < 3; ++k) c[1][k] = u[k] - v[k];
    vf_poly_mul(e[1][0], 2, e[2][1], 2, u);
    vf_poly_mul(e[1][1], 2, e[2][0], 2, v);
    for (int k = 0; k < 3; ++k) c[2][k] = u[k] - v[k];
    vf_poly_mul(e[0][0], 2, c[0], 3, t0);
    vf_poly_mul(e[0][1], 2, c[1], 3, t1);
    vf_poly_mul(e[0][2], 2, c[2], 3, t2);
    for (int k = 0; k < 4; ++k) det[k] = (t0[k] + t1[k]) + t2[k];
}

// Squared Sampson error of one correspondence (gtsfm/utils/verification.py:172-220; oracle: sampson_sq).
__device__ __forceinline__ double vf_sampson_sq(const double* e, double a, double b, double c, double d) {
    const double l2x = (e[0] * a + e[1] * b) + e[2];
    const double l2y = (e[3] * a + e[4] * b) + e[5];
    const double l2z = (e[6] * a + e[7] * b) + e[8];
    const double l1x = (e[0] * c + e[3] * d) + e[6];
    const double l1y = (e[1] * c + e[4] * d) + e[7];
    const double r = (c * l2x + d * l2y) + l2z;
    const double den = ((l2x * l2x + l2y * l2y) + l1x * l1x) + l1y * l1y;
    return (r * r) / den;
}

__global__ __launch_bounds__(256) void verify_gather_kernel(const float* __restrict__ kp_xy, const long long* __restrict__ kp_off1,
                                                            const long long* __restrict__ kp_off2, const int* __restrict__ match_idx,
                                                            const long long* __restrict__ match_off, const int* __restrict__ match_count,
                                                            const double* __restrict__ intrinsics, int normalise, double* __restrict__ pts) {
    const int pair = blockIdx.y;
    const long long begin = match_off[pair], m = match_count ? (long long)match_count[pair] : match_off[pair + 1] - begin;
    const double* K = intrinsics + 8 * (size_t)pair;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < m; i += (long long)gridDim.x * blockDim.x) {
        const int* mi = match_idx + 2 * (begin + i);
        const float* a = kp_xy + 2 * (kp_off1[pair] + mi[0]);
        const float* b = kp_xy + 2 * (kp_off2[pair] + mi[1]);
        double* o = pts + 4 * (begin + i);
        if (normalise) {  // Cal3Bundler.calibrate without distortion, gtsfm/utils/features.py:41-51
            o[0] = ((double)a[0] - K[2]) / K[0];
            o[1] = ((double)a[1] - K[3]) / K[1];
            o[2] = ((double)b[0] - K[6]) / K[4];
            o[3] = ((double)b[1] - K[7]) / K[5];
        } else {  // fundamental-matrix mode works on pixels
            o[0] = (double)a[0], o[1] = (double)a[1], o[2] = (double)b[0], o[3] = (double)b[1];
        }
    }
}

struct VfShared {
    double cost[256];
    int index[256];
    double best_e[9];
    double best_cost;
    int best_count, best_index, stop, inliers;
    double pose[2][9];
    double t[3];
    int good[4];
    double part[256][20];  // polish: per-thread partial sums of J^T J (15) and J^T r (5)
};

// Cyclic Jacobi eigen-decomposition of a symmetric 3x3 (oracle: _jacobi_eigen_sym3).
__device__ void vf_jacobi3(double a[3][3], double v[3][3]) {
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) v[i][j] = i == j ? 1.0 : 0.0;
    const int pq[3][2] = {{0, 1}, {0, 2}, {1, 2}};
    for (int sweep = 0; sweep < VF_JACOBI_SWEEPS; ++sweep)
        for (int n = 0; n < 3; ++n) {
            const int p = pq[n][0], q = pq[n][1];
            if (a[p][q] == 0.0) continue;
            const double tau = (a[q][q] - a[p][p]) / (2.0 * a[p][q]);
            const double t = (tau >= 0 ? 1.0 : -1.0) / (fabs(tau) + sqrt(1.0 + tau * tau));
            const double c = 1.0 / sqrt(1.0 + t * t);
            const double sn = t * c;
            const double app = a[p][p], aqq = a[q][q], apq = a[p][q];
            a[p][p] = app - t * apq;
            a[q][q] = aqq + t * apq;
            a[p][q] = a[q][p] = 0.0;
            const int r = 3 - p - q;
            const double arp = a[r][p], arq = a[r][q];
            a[r][p] = a[p][r] = c * arp - sn * arq;
            a[r][q] = a[q][r] = sn * arp + c * arq;
            for (int k = 0; k < 3; ++k) {
                const double vkp = v[k][p], vkq = v[k][q];
                v[k][p] = c * vkp - sn * vkq;
                v[k][q] = sn * vkp + c * vkq;
            }
        }
}

__device__ __forceinline__ void vf_mat3(const double a[3][3], const double b[3][3], double out[3][3]) {
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) out[i][j] = (a[i][0] * b[0][j] + a[i][1] * b[1][j]) + a[i][2] * b[2][j];
}

// E -> R1, R2 (row-major) and t as cv.decomposeEssentialMat (oracle: decompose_essential).
__device__ void vf_decompose(const double* ev, double r1[9], double r2[9], double t[3]) {
    double e[3][3], s[3][3], v[3][3];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) e[i][j] = ev[3 * i + j];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) s[i][j] = (e[0][i] * e[0][j] + e[1][i] * e[1][j]) + e[2][i] * e[2][j];
    vf_jacobi3(s, v);
    const double lam[3] = {s[0][0], s[1][1], s[2][2]};
    int order[3] = {0, 1, 2};  // descending eigenvalue, ties by index (insertion sort, stable)
    for (int i = 1; i < 3; ++i)
        for (int j = i; j > 0 && lam[order[j]] > lam[order[j - 1]]; --j) {
            const int tmp = order[j];
            order[j] = order[j - 1];
            order[j - 1] = tmp;
        }
    double v0[3], v1[3], u0[3], u1[3], u2[3], v2[3];
    for (int k = 0; k < 3; ++k) v0[k] = v[k][order[0]], v1[k] = v[k][order[1]];
    for (int i = 0; i < 3; ++i) u0[i] = (e[i][0] * v0[0] + e[i][1] * v0[1]) + e[i][2] * v0[2];
    double n = sqrt((u0[0] * u0[0] + u0[1] * u0[1]) + u0[2] * u0[2]);
    for (int i = 0; i < 3; ++i) u0[i] = u0[i] / n;
    for (int i = 0; i < 3; ++i) u1[i] = (e[i][0] * v1[0] + e[i][1] * v1[1]) + e[i][2] * v1[2];
    const double dot = (u0[0] * u1[0] + u0[1] * u1[1]) + u0[2] * u1[2];
    for (int i = 0; i < 3; ++i) u1[i] = u1[i] - dot * u0[i];
    n = sqrt((u1[0] * u1[0] + u1[1] * u1[1]) + u1[2] * u1[2]);
    for (int i = 0; i < 3; ++i) u1[i] = u1[i] / n;
    u2[0] = u0[1] * u1[2] - u0[2] * u1[1], u2[1] = u0[2] * u1[0] - u0[0] * u1[2], u2[2] = u0[0] * u1[1] - u0[1] * u1[0];
    v2[0] = v0[1] * v1[2] - v0[2] * v1[1], v2[1] = v0[2] * v1[0] - v0[0] * v1[2], v2[2] = v0[0] * v1[1] - v0[1] * v1[0];
    double um[3][3], vt[3][3], uw[3][3], out[3][3];
    for (int i = 0; i < 3; ++i) um[i][0] = u0[i], um[i][1] = u1[i], um[i][2] = u2[i];
    for (int j = 0; j < 3; ++j) vt[0][j] = v0[j], vt[1][j] = v1[j], vt[2][j] = v2[j];
    const double w[3][3] = {{0.0, -1.0, 0.0}, {1.0, 0.0, 0.0}, {0.0, 0.0, 1.0}};
    const double wt[3][3] = {{0.0, 1.0, 0.0}, {-1.0, 0.0, 0.0}, {0.0, 0.0, 1.0}};
    vf_mat3(um, w, uw);
    vf_mat3(uw, vt, out);
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) r1[3 * i + j] = out[i][j];
    vf_mat3(um, wt, uw);
    vf_mat3(uw, vt, out);
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) r2[3 * i + j] = out[i][j];
    for (int i = 0; i < 3; ++i) t[i] = u2[i];
}

// Both depths of l1 R x1 + t = l2 x2 in (0, VF_DEPTH_LIMIT)? (oracle: cheirality_count)
__device__ __forceinline__ bool vf_in_front(const double* r, double t0, double t1, double t2, double x, double y, double bx, double by) {
    const double ax = (r[0] * x + r[1] * y) + r[2];
    const double ay = (r[3] * x + r[4] * y) + r[5];
    const double az = (r[6] * x + r[7] * y) + r[8];
    const double aa = (ax * ax + ay * ay) + az * az;
    const double bb = (bx * bx + by * by) + 1.0;
    const double ab = (ax * bx + ay * by) + az;
    const double at = (ax * t0 + ay * t1) + az * t2;
    const double bt = (bx * t0 + by * t1) + t2;
    const double det = aa * bb - ab * ab;
    const double l1 = (ab * bt - at * bb) / det;
    const double l2 = (aa * bt - ab * at) / det;
    return l1 > 0 && l2 > 0 && l1 < VF_DEPTH_LIMIT && l2 < VF_DEPTH_LIMIT;
}

// ---- final polish of the pose (oracle: polish_pose and helpers) ----
__device__ __forceinline__ void vf_essential_from_pose(const double r[3][3], const double t[3], double e[9]) {
    const double sk[3][3] = {{0.0, -t[2], t[1]}, {t[2], 0.0, -t[0]}, {-t[1], t[0], 0.0}};
    double em[3][3];
    vf_mat3(sk, r, em);
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) e[3 * i + j] = em[i][j];
}

__device__ void vf_tangent_basis(const double t[3], double b1[3], double b2[3]) {
    int k = 0;
    for (int i = 1; i < 3; ++i)
        if (fabs(t[i]) < fabs(t[k])) k = i;
    double a[3] = {0.0, 0.0, 0.0};
    a[k] = 1.0;
    b1[0] = t[1] * a[2] - t[2] * a[1], b1[1] = t[2] * a[0] - t[0] * a[2], b1[2] = t[0] * a[1] - t[1] * a[0];
    const double n = sqrt((b1[0] * b1[0] + b1[1] * b1[1]) + b1[2] * b1[2]);
    for (int i = 0; i < 3; ++i) b1[i] = b1[i] / n;
    b2[0] = t[1] * b1[2] - t[2] * b1[1], b2[1] = t[2] * b1[0] - t[0] * b1[2], b2[2] = t[0] * b1[1] - t[1] * b1[0];
}

// R <- R cayley(d[0:3]), t <- unit(t + d3 b1 + d4 b2)
__device__ void vf_perturb_pose(const double r[3][3], const double t[3], const double b1[3], const double b2[3], const double d[5], double rn[3][3],
                                double tn[3]) {
    const double wx = 0.5 * d[0], wy = 0.5 * d[1], wz = 0.5 * d[2];
    const double a[3][3] = {{1.0, wz, -wy}, {-wz, 1.0, wx}, {wy, -wx, 1.0}};
    const double b[3][3] = {{1.0, -wz, wy}, {wz, 1.0, -wx}, {-wy, wx, 1.0}};
    double adj[3][3];
    adj[0][0] = a[1][1] * a[2][2] - a[1][2] * a[2][1], adj[0][1] = a[0][2] * a[2][1] - a[0][1] * a[2][2], adj[0][2] = a[0][1] * a[1][2] - a[0][2] * a[1][1];
    adj[1][0] = a[1][2] * a[2][0] - a[1][0] * a[2][2], adj[1][1] = a[0][0] * a[2][2] - a[0][2] * a[2][0], adj[1][2] = a[0][2] * a[1][0] - a[0][0] * a[1][2];
    adj[2][0] = a[1][0] * a[2][1] - a[1][1] * a[2][0], adj[2][1] = a[0][1] * a[2][0] - a[0][0] * a[2][1], adj[2][2] = a[0][0] * a[1][1] - a[0][1] * a[1][0];
    const double det = (a[0][0] * adj[0][0] + a[0][1] * adj[1][0]) + a[0][2] * adj[2][0];
    double inv[3][3], q[3][3];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) inv[i][j] = adj[i][j] / det;
    vf_mat3(inv, b, q);
    vf_mat3(r, q, rn);
    for (int i = 0; i < 3; ++i) tn[i] = (t[i] + d[3] * b1[i]) + d[4] * b2[i];
    const double n = sqrt((tn[0] * tn[0] + tn[1] * tn[1]) + tn[2] * tn[2]);
    for (int i = 0; i < 3; ++i) tn[i] = tn[i] / n;
}

__device__ __forceinline__ double vf_signed_sampson(const double* e, double a, double b, double c, double d) {
    const double l2x = (e[0] * a + e[1] * b) + e[2];
    const double l2y = (e[3] * a + e[4] * b) + e[5];
    const double l2z = (e[6] * a + e[7] * b) + e[8];
    const double l1x = (e[0] * c + e[3] * d) + e[6];
    const double l1y = (e[1] * c + e[4] * d) + e[7];
    const double r = (c * l2x + d * l2y) + l2z;
    const double den = ((l2x * l2x + l2y * l2y) + l1x * l1x) + l1y * l1y;
    return r / sqrt(den);
}

// h d = -g, Gaussian elimination with row pivoting (oracle: _solve5)
__device__ void vf_solve5(const double h[5][5], const double g[5], double d[5]) {
    double a[5][6];
    for (int i = 0; i < 5; ++i) {
        for (int j = 0; j < 5; ++j) a[i][j] = h[i][j];
        a[i][5] = -g[i];
    }
    for (int c = 0; c < 5; ++c) {
        int pr = c;
        for (int i = c + 1; i < 5; ++i)
            if (fabs(a[i][c]) > fabs(a[pr][c])) pr = i;
        for (int j = 0; j < 6; ++j) {
            const double tmp = a[c][j];
            a[c][j] = a[pr][j];
            a[pr][j] = tmp;
        }
        for (int j = 5; j >= c; --j) a[c][j] = a[c][j] / a[c][c];
        for (int i = 0; i < 5; ++i) {
            if (i == c) continue;
            const double f = a[i][c];
            for (int j = c; j < 6; ++j) a[i][j] = a[i][j] - f * a[c][j];
        }
    }
    for (int i = 0; i < 5; ++i) d[i] = a[i][5];
}

// All real solutions of one minimal sample. MODE 0: five-point essential matrices (<= 10); MODE 1: seven-point fundamental
// matrices (<= 3). P = the pair's points (x1, y1, x2, y2), normalised (MODE 0) or pixels (MODE 1).
template <int MODE>
__device__ int vf_solve(const double* __restrict__ P, const int* idx, double models[][9]) {
    if (MODE == 0) {
        double basis[4][9];
        {
            double q[5][9];
            for (int k = 0; k < 5; ++k) {
                const double a = P[4 * idx[k]], b = P[4 * idx[k] + 1], c = P[4 * idx[k] + 2], d = P[4 * idx[k] + 3];
                q[k][0] = c * a, q[k][1] = c * b, q[k][2] = c, q[k][3] = d * a, q[k][4] = d * b, q[k][5] = d, q[k][6] = a, q[k][7] = b, q[k][8] = 1.0;
            }
            vf_null_space<5>(q, basis);
        }
        double p1[8], p2[8], p3[7], det[11], roots[10];
        {
            double mat[10][20];
            vf_constraints(basis, mat);
            vf_gauss_jordan(mat);
            vf_hidden_variable(mat, p1, p2, p3, det);
        }
        const int nroots = vf_real_roots(det, 10, roots);
        for (int r = 0; r < nroots; ++r) {
            const double z = roots[r];
            const double x = vf_horner(p1, 7, z) / vf_horner(p3, 6, z);
            const double y = vf_horner(p2, 7, z) / vf_horner(p3, 6, z);
            for (int k = 0; k < 9; ++k) models[r][k] = ((x * basis[0][k] + y * basis[1][k]) + z * basis[2][k]) + basis[3][k];
        }
        return nroots;
    } else {
        double basis[2][9];
        {
            double q[7][9];
            for (int k = 0; k < 7; ++k) {
                const double a = P[4 * idx[k]], b = P[4 * idx[k] + 1], c = P[4 * idx[k] + 2], d = P[4 * idx[k] + 3];
                q[k][0] = c * a, q[k][1] = c * b, q[k][2] = c, q[k][3] = d * a, q[k][4] = d * b, q[k][5] = d, q[k][6] = a, q[k][7] = b, q[k][8] = 1.0;
            }
            vf_null_space<7>(q, basis);
        }
        double det[4], roots[3];
        vf_seven_point_cubic(basis, det);
        const int nroots = vf_real_roots(det, 3, roots);
        for (int r = 0; r < nroots; ++r)
            for (int k = 0; k < 9; ++k) models[r][k] = roots[r] * basis[0][k] + basis[1][k];
        return nroots;
    }
}

template <int MODE>
__device__ __forceinline__ double vf_residual(const double* model, double a, double b, double c, double d) {
    return MODE == 0 ? vf_sampson_sq(model, a, b, c, d) : vf_epipolar_sq_max(model, a, b, c, d);
}

template <int MODE>
__global__ __launch_bounds__(256) void verify_ransac_kernel(const double* __restrict__ pts, const long long* __restrict__ match_off,
                                                            const int* __restrict__ match_count, const double* __restrict__ intrinsics, const unsigned long long* __restrict__ seeds,
                                                            double threshold_px, double* __restrict__ out_e, double* __restrict__ out_r,
                                                            double* __restrict__ out_t, unsigned char* __restrict__ out_mask,
                                                            int* __restrict__ out_stats, int* __restrict__ inlier_lists, double* __restrict__ out_f) {
    constexpr int SIZE = MODE == 0 ? 5 : 7;      // minimal sample
    constexpr int MAX_MODELS = MODE == 0 ? 10 : 3;
    constexpr int MIN_MATCHES = MODE == 0 ? 6 : 8;  // opencv_verifier_base.py:71-80 (NUM_MATCHES_REQ_E_MATRIX and the "< 6" guard; NUM_MATCHES_REQ_F_MATRIX)
    __shared__ VfShared sh;
    const int pair = blockIdx.x, tid = threadIdx.x;
    const long long begin = match_off[pair];
    const int m = match_count ? match_count[pair] : (int)(match_off[pair + 1] - begin);
    const double* P = pts + 4 * begin;
    const double* K = intrinsics + 8 * (size_t)pair;
    unsigned char* mask = out_mask + begin;
    int* inl_list = inlier_lists + begin;
    int* stats = out_stats + 8 * (size_t)pair;
    for (long long i = m + tid; i < match_off[pair + 1] - begin; i += 256) mask[i] = 0;  // unused capacity behind the list
    if (m < MIN_MATCHES) {
        for (int i = tid; i < m; i += 256) mask[i] = 0;
        if (tid < 8) stats[tid] = tid >= 2 && tid < 4 ? -1 : 0;
        if (tid < 9) out_e[9 * (size_t)pair + tid] = out_r[9 * (size_t)pair + tid] = NAN;
        if (MODE == 1 && tid < 9) out_f[9 * (size_t)pair + tid] = NAN;
        if (tid < 3) out_t[3 * (size_t)pair + tid] = NAN;
        return;
    }
    // essential mode: px / max(fx1, fx2) in normalised units (opencv_verifier_base.py:86-90); fundamental mode: pixels
    const double thr = MODE == 0 ? threshold_px / (K[0] > K[4] ? K[0] : K[4]) : threshold_px;
    const double thr2 = thr * thr;
    const unsigned long long seed = seeds[pair];
    if (tid == 0) sh.best_cost = INFINITY, sh.best_count = 0, sh.best_index = -1, sh.stop = 0;
    __syncthreads();

    int rounds = 0;
    bool lo_round = false;  // the last round draws its minimal samples from the inliers of the best model so far
    for (;;) {
        const int hyp = (lo_round ? VF_MAX_ROUNDS : rounds) * VF_ROUND + tid;
        int idx[SIZE];
        vf_sample(seed, (unsigned long long)hyp, lo_round ? sh.inliers : m, idx, SIZE);
        if (lo_round)
            for (int k = 0; k < SIZE; ++k) idx[k] = inl_list[idx[k]];
        double models[MAX_MODELS][9];
        const int nroots = vf_solve<MODE>(P, idx, models);
        double my_cost = INFINITY;
        int my_count = 0, my_root = 0;
        for (int r = 0; r < nroots; ++r) {
            // MSAC cost (USAC's default score): sum of min(error, thr^2), added in match order; NaN errors are outliers
            double cost = 0.0;
            int count = 0;
            for (int i = 0; i < m; ++i) {
                const double err = vf_residual<MODE>(models[r], P[4 * i], P[4 * i + 1], P[4 * i + 2], P[4 * i + 3]);
                const bool in = err < thr2;
                cost = cost + (in ? err : thr2);
                count += in ? 1 : 0;
            }
            if (cost < my_cost) my_cost = cost, my_count = count, my_root = r;
        }
        // round winner: lowest cost, then the smallest (hypothesis, root)
        sh.cost[tid] = my_cost;
        sh.index[tid] = hyp * 16 + my_root;
        __syncthreads();
        for (int step = 128; step > 0; step >>= 1) {
            if (tid < step) {
                const double c0 = sh.cost[tid], c1 = sh.cost[tid + step];
                if (c1 < c0 || (c1 == c0 && sh.index[tid + step] < sh.index[tid])) sh.cost[tid] = c1, sh.index[tid] = sh.index[tid + step];
            }
            __syncthreads();
        }
        const double win_cost = sh.cost[0];
        const int win_index = sh.index[0];
        const bool improves = win_cost < sh.best_cost;
        __syncthreads();
        if (improves && hyp * 16 + my_root == win_index) {
            for (int k = 0; k < 9; ++k) sh.best_e[k] = models[my_root][k];
            sh.best_cost = win_cost;
            sh.best_count = my_count;
            sh.best_index = win_index;
        }
        __syncthreads();
        if (lo_round) break;
        ++rounds;
        if (tid == 0 && sh.best_count > 0) {  // (1 - w^SIZE)^(256 rounds) <= 1 - p, exact multiplication chain (oracle: _stop_after)
            const double w = (double)sh.best_count / (double)m;
            double ws = w;
            for (int k = 0; k < SIZE - 1; ++k) ws = ws * w;
            const double q = 1.0 - ws;
            double p256 = q;
            for (int k = 0; k < 8; ++k) p256 = p256 * p256;
            double acc = p256;
            for (int k = 0; k < rounds - 1; ++k) acc = acc * p256;
            sh.stop = acc <= 1.0 - VF_SUCCESS_PROB ? 1 : 0;
        }
        __syncthreads();
        if (!sh.stop && rounds < VF_MAX_ROUNDS) continue;
        // local optimisation (LO-RANSAC's inner sampling, minimal samples): list the inliers of the best model in match order
        if (sh.best_index < 0) break;
        double e[9];
        for (int k = 0; k < 9; ++k) e[k] = sh.best_e[k];
        if (tid == 0) sh.inliers = 0;
        __syncthreads();
        for (int base = 0; base < m; base += 256) {
            const int i = base + tid;
            const bool in = i < m && vf_residual<MODE>(e, P[4 * i], P[4 * i + 1], P[4 * i + 2], P[4 * i + 3]) < thr2;
            const unsigned long long ballot = __ballot(in);
            if ((tid & 63) == 0) sh.index[tid >> 6] = __popcll(ballot);
            __syncthreads();
            int offset = sh.inliers;
            for (int w = 0; w < (tid >> 6); ++w) offset += sh.index[w];
            if (in) inl_list[offset + __popcll(ballot & ((1ull << (tid & 63)) - 1ull))] = i;
            __threadfence_block();
            __syncthreads();
            if (tid == 0) sh.inliers += sh.index[0] + sh.index[1] + sh.index[2] + sh.index[3];
            __syncthreads();
        }
        if (sh.inliers < SIZE + 1) break;
        lo_round = true;
    }
    const int hypotheses = (rounds + (lo_round ? 1 : 0)) * VF_ROUND;

    if (sh.best_index < 0) {  // no sample produced a model
        for (int i = tid; i < m; i += 256) mask[i] = 0;
        if (tid < 8) stats[tid] = tid == 1 ? hypotheses : (tid >= 2 && tid < 4 ? -1 : 0);
        if (tid < 9) out_e[9 * (size_t)pair + tid] = out_r[9 * (size_t)pair + tid] = NAN;
        if (MODE == 1 && tid < 9) out_f[9 * (size_t)pair + tid] = NAN;
        if (tid < 3) out_t[3 * (size_t)pair + tid] = NAN;
        return;
    }
    double model[9], e[9];
    for (int k = 0; k < 9; ++k) model[k] = e[k] = sh.best_e[k];
    if (MODE == 1) {  // E = K2^T F K1 (gtsfm/utils/verification.py:99-112), products in the oracle's order
        const double k2t[3][3] = {{K[4], 0.0, 0.0}, {0.0, K[5], 0.0}, {K[6], K[7], 1.0}};
        const double k1[3][3] = {{K[0], 0.0, K[2]}, {0.0, K[1], K[3]}, {0.0, 0.0, 1.0}};
        double f[3][3], t[3][3], em[3][3];
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) f[i][j] = model[3 * i + j];
        vf_mat3(k2t, f, t);
        vf_mat3(t, k1, em);
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) e[3 * i + j] = em[i][j];
    }
    __syncthreads();
    if (tid == 0) {
        sh.inliers = 0;
        for (int k = 0; k < 4; ++k) sh.good[k] = 0;
        double r1[9], r2[9], t[3];
        vf_decompose(e, r1, r2, t);
        for (int k = 0; k < 9; ++k) sh.pose[0][k] = r1[k], sh.pose[1][k] = r2[k];
        for (int k = 0; k < 3; ++k) sh.t[k] = t[k];
    }
    __syncthreads();
    int inl = 0, good[4] = {0, 0, 0, 0};
    const double t0 = sh.t[0], t1 = sh.t[1], t2 = sh.t[2];
    for (int i = tid; i < m; i += 256) {
        double a = P[4 * i], b = P[4 * i + 1], c = P[4 * i + 2], d = P[4 * i + 3];
        const bool in = vf_residual<MODE>(model, a, b, c, d) < thr2;
        mask[i] = in ? 1 : 0;
        if (in) {
            ++inl;
            if (MODE == 1) a = (a - K[2]) / K[0], b = (b - K[3]) / K[1], c = (c - K[6]) / K[4], d = (d - K[7]) / K[5];  // recoverPose works on normalised points
            good[0] += vf_in_front(sh.pose[0], t0, t1, t2, a, b, c, d) ? 1 : 0;
            good[1] += vf_in_front(sh.pose[1], t0, t1, t2, a, b, c, d) ? 1 : 0;
            good[2] += vf_in_front(sh.pose[0], -t0, -t1, -t2, a, b, c, d) ? 1 : 0;
            good[3] += vf_in_front(sh.pose[1], -t0, -t1, -t2, a, b, c, d) ? 1 : 0;
        }
    }
    atomicAdd(&sh.inliers, inl);
    for (int k = 0; k < 4; ++k) atomicAdd(&sh.good[k], good[k]);
    __syncthreads();
    int pick = 3;
    for (int k = 0; k < 4; ++k) {  // cv.recoverPose's order: the first candidate whose count is >= all others
        bool top = true;
        for (int j = 0; j < 4; ++j) top &= sh.good[k] >= sh.good[j];
        if (top) {
            pick = k;
            break;
        }
    }
    double rot[3][3], tr[3];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) rot[i][j] = sh.pose[pick & 1][3 * i + j];
    for (int k = 0; k < 3; ++k) tr[k] = pick >= 2 ? -sh.t[k] : sh.t[k];
    int inliers = sh.inliers;
    const int good0 = sh.good[0], good1 = sh.good[1], good2 = sh.good[2], good3 = sh.good[3];

    if (MODE == 0) {
        // final polish (oracle: polish_pose): Gauss-Newton on the winner's inliers over (rotation vector, 2 tangent directions of t),
        // forward-difference Jacobian; every thread carries the same pose, the 20 sums are reduced in a fixed order
        for (int it = 0; it < VF_POLISH_ITERS; ++it) {
            double b1[3], b2[3], ev[6][9];
            vf_tangent_basis(tr, b1, b2);
            vf_essential_from_pose(rot, tr, ev[0]);
            for (int k = 0; k < 5; ++k) {
                double d[5] = {0.0, 0.0, 0.0, 0.0, 0.0}, rk[3][3], tk[3];
                d[k] = VF_POLISH_STEP;
                vf_perturb_pose(rot, tr, b1, b2, d, rk, tk);
                vf_essential_from_pose(rk, tk, ev[k + 1]);
            }
            double acc[20];
            for (int k = 0; k < 20; ++k) acc[k] = 0.0;
            for (int i = tid; i < m; i += 256) {
                const double a = P[4 * i], b = P[4 * i + 1], c = P[4 * i + 2], d = P[4 * i + 3];
                const bool in = vf_sampson_sq(model, a, b, c, d) < thr2;  // the winner's inliers, fixed during the polish
                const double r0 = vf_signed_sampson(ev[0], a, b, c, d);
                double col[5];
                for (int k = 0; k < 5; ++k) col[k] = (vf_signed_sampson(ev[k + 1], a, b, c, d) - r0) / VF_POLISH_STEP;
                int n = 0;
                for (int u = 0; u < 5; ++u)
                    for (int v = u; v < 5; ++v, ++n) acc[n] = acc[n] + (in ? col[u] * col[v] : 0.0);
                for (int u = 0; u < 5; ++u) acc[15 + u] = acc[15 + u] + (in ? col[u] * r0 : 0.0);
            }
            for (int k = 0; k < 20; ++k) sh.part[tid][k] = acc[k];
            __syncthreads();
            for (int step = 128; step > 0; step >>= 1) {
                if (tid < step)
                    for (int k = 0; k < 20; ++k) sh.part[tid][k] = sh.part[tid][k] + sh.part[tid + step][k];
                __syncthreads();
            }
            double h[5][5], g[5], delta[5];
            {
                int n = 0;
                for (int u = 0; u < 5; ++u)
                    for (int v = u; v < 5; ++v, ++n) h[u][v] = h[v][u] = sh.part[0][n];
                for (int u = 0; u < 5; ++u) g[u] = sh.part[0][15 + u];
            }
            __syncthreads();  // part[0] is rewritten in the next iteration
            for (int u = 0; u < 5; ++u) h[u][u] = h[u][u] + 1.0e-12 * (1.0 + h[u][u]);
            vf_solve5(h, g, delta);
            double rn[3][3], tn[3];
            vf_perturb_pose(rot, tr, b1, b2, delta, rn, tn);
            for (int i = 0; i < 3; ++i) {
                tr[i] = tn[i];
                for (int j = 0; j < 3; ++j) rot[i][j] = rn[i][j];
            }
        }
        double e2[9];
        vf_essential_from_pose(rot, tr, e2);
        double cost2 = 0.0;  // MSAC cost over ALL matches, in match order (every thread: same value)
        for (int i = 0; i < m; ++i) {
            const double err = vf_sampson_sq(e2, P[4 * i], P[4 * i + 1], P[4 * i + 2], P[4 * i + 3]);
            cost2 = cost2 + (err < thr2 ? err : thr2);
        }
        if (cost2 < sh.best_cost) {  // keep the polished pose; the verified set is the one of the polished model
            for (int k = 0; k < 9; ++k) e[k] = e2[k];
            if (tid == 0) sh.inliers = 0;
            __syncthreads();
            int cnt = 0;
            for (int i = tid; i < m; i += 256) {
                const bool in = vf_sampson_sq(e2, P[4 * i], P[4 * i + 1], P[4 * i + 2], P[4 * i + 3]) < thr2;
                mask[i] = in ? 1 : 0;
                cnt += in ? 1 : 0;
            }
            atomicAdd(&sh.inliers, cnt);
            __syncthreads();
            inliers = sh.inliers;
        } else {
            for (int i = 0; i < 3; ++i) {
                tr[i] = pick >= 2 ? -sh.t[i] : sh.t[i];
                for (int j = 0; j < 3; ++j) rot[i][j] = sh.pose[pick & 1][3 * i + j];
            }
        }
    }
    if (tid == 0) {
        for (int k = 0; k < 9; ++k) {
            out_e[9 * (size_t)pair + k] = e[k];
            out_r[9 * (size_t)pair + k] = rot[k / 3][k % 3];
            if (MODE == 1) out_f[9 * (size_t)pair + k] = model[k];
        }
        for (int k = 0; k < 3; ++k) out_t[3 * (size_t)pair + k] = tr[k];
        stats[0] = inliers, stats[1] = hypotheses, stats[2] = sh.best_index >> 4, stats[3] = sh.best_index & 15;
        stats[4] = good0, stats[5] = good1, stats[6] = good2, stats[7] = good3;
    }
}

// Matcher output -> ragged match lists without a host round trip: pair p's matches0 block (n0 rows of `matches`, -1 =
// unmatched) becomes the (row, matches0[row]) pairs in row order (the plugins' (K, 2) format, superglue_matcher.py:100-102)
// at match_off[p], their number in match_count[p]. One workgroup per pair, ordered by a ballot prefix per 256 rows.
__global__ __launch_bounds__(256) void verify_compact_matches_kernel(const int* __restrict__ matches, const long long* __restrict__ row_off,
                                                                     const int* __restrict__ n0, const long long* __restrict__ match_off,
                                                                     int* __restrict__ match_idx, int* __restrict__ match_count) {
    __shared__ int wave_total[4];
    __shared__ int running;
    const int pair = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int* src = matches + row_off[pair];
    int* dst = match_idx + 2 * match_off[pair];
    const int rows = n0[pair];
    if (tid == 0) running = 0;
    __syncthreads();
    for (int base = 0; base < rows; base += 256) {
        const int row = base + tid;
        const int v = row < rows ? src[row] : -1;
        const unsigned long long ballot = __ballot(v > -1);
        const int before = __popcll(ballot & ((1ull << lane) - 1ull));
        if (lane == 0) wave_total[wave] = __popcll(ballot);
        __syncthreads();
        int offset = running;
        for (int w = 0; w < wave; ++w) offset += wave_total[w];
        if (v > -1) {
            dst[2 * (offset + before)] = row;
            dst[2 * (offset + before) + 1] = v;
        }
        __syncthreads();
        if (tid == 0) running += wave_total[0] + wave_total[1] + wave_total[2] + wave_total[3];
        __syncthreads();
    }
    if (tid == 0) match_count[pair] = running;
}

extern "C" int gtsfm_verify_compact_matches(const int32_t* matches_dev, const long long* row_off_dev, const int32_t* n0_dev,
                                            const long long* match_off_dev, int num_pairs, int32_t* match_idx_dev,
                                            int32_t* match_count_dev, void* stream) {
    GTSFM_CHECK_ARG(num_pairs >= 0, "verify_compact_matches: negative pair count");
    if (num_pairs == 0) return GTSFM_OK;
    GTSFM_CHECK_ARG(matches_dev && row_off_dev && n0_dev && match_off_dev && match_idx_dev && match_count_dev, "verify_compact_matches: null pointer");
    hipLaunchKernelGGL(verify_compact_matches_kernel, dim3(num_pairs), dim3(256), 0, (hipStream_t)stream, matches_dev, row_off_dev, n0_dev,
                       match_off_dev, match_idx_dev, match_count_dev);
    GTSFM_CHECK_LAUNCH("verify_compact_matches_kernel");
    return GTSFM_OK;
}

extern "C" size_t gtsfm_verify_workspace_bytes(long long total_matches) {
    const size_t n = (size_t)(total_matches > 0 ? total_matches : 0);
    return align_up(n * 4 * sizeof(double), 256) + align_up(n * sizeof(int), 256) + 256;  // normalised points, inlier lists
}

static int verify_two_view(int mode, const float* kp_xy_dev, const long long* kp_off1_dev, const long long* kp_off2_dev, const int32_t* match_idx_dev,
                           const long long* match_off_dev, const int32_t* match_count_dev, long long total_matches, const double* intrinsics_dev,
                           const unsigned long long* seeds_dev, double threshold_px, int num_pairs, void* workspace_dev, size_t workspace_bytes,
                           double* fundamental_dev, double* essential_dev, double* rotation_dev, double* translation_dev, uint8_t* inlier_mask_dev,
                           int32_t* stats_dev, void* stream) {
    GTSFM_CHECK_ARG(num_pairs >= 0 && total_matches >= 0 && threshold_px > 0, "verify: bad sizes or threshold");
    if (num_pairs == 0) return GTSFM_OK;
    GTSFM_CHECK_ARG(kp_xy_dev && kp_off1_dev && kp_off2_dev && match_off_dev && intrinsics_dev && seeds_dev, "verify: null input");
    GTSFM_CHECK_ARG(essential_dev && rotation_dev && translation_dev && stats_dev && (mode == 0 || fundamental_dev), "verify: null output");
    GTSFM_CHECK_ARG(total_matches == 0 || (match_idx_dev && inlier_mask_dev), "verify: null match arrays");
    GTSFM_CHECK_ARG(num_pairs <= 65535, "verify: at most 65535 pairs per call");
    if (workspace_bytes < gtsfm_verify_workspace_bytes(total_matches) || !workspace_dev) {
        gtsfm_set_error("verify: workspace too small (%zu < %zu)", workspace_bytes, gtsfm_verify_workspace_bytes(total_matches));
        return GTSFM_ERR_WORKSPACE;
    }
    double* pts = (double*)align_up((size_t)workspace_dev, 256);
    int* inlier_lists = (int*)((char*)pts + align_up((size_t)total_matches * 4 * sizeof(double), 256));
    if (total_matches > 0) {
        hipLaunchKernelGGL(verify_gather_kernel, dim3(4, num_pairs), dim3(256), 0, (hipStream_t)stream, kp_xy_dev, kp_off1_dev, kp_off2_dev,
                           match_idx_dev, match_off_dev, match_count_dev, intrinsics_dev, mode == 0 ? 1 : 0, pts);
        GTSFM_CHECK_LAUNCH("verify_gather_kernel");
    }
    if (mode == 0)
        hipLaunchKernelGGL(verify_ransac_kernel<0>, dim3(num_pairs), dim3(256), 0, (hipStream_t)stream, pts, match_off_dev, match_count_dev, intrinsics_dev,
                           seeds_dev, threshold_px, essential_dev, rotation_dev, translation_dev, inlier_mask_dev, stats_dev, inlier_lists, (double*)nullptr);
    else
        hipLaunchKernelGGL(verify_ransac_kernel<1>, dim3(num_pairs), dim3(256), 0, (hipStream_t)stream, pts, match_off_dev, match_count_dev, intrinsics_dev,
                           seeds_dev, threshold_px, essential_dev, rotation_dev, translation_dev, inlier_mask_dev, stats_dev, inlier_lists, fundamental_dev);
    GTSFM_CHECK_LAUNCH("verify_ransac_kernel");
    return GTSFM_OK;
}

extern "C" int gtsfm_verify_essential_f64(const float* kp_xy_dev, const long long* kp_off1_dev, const long long* kp_off2_dev,
                                          const int32_t* match_idx_dev, const long long* match_off_dev, const int32_t* match_count_dev,
                                          long long total_matches, const double* intrinsics_dev, const unsigned long long* seeds_dev, double threshold_px,
                                          int num_pairs, void* workspace_dev, size_t workspace_bytes, double* essential_dev,
                                          double* rotation_dev, double* translation_dev, uint8_t* inlier_mask_dev, int32_t* stats_dev,
                                          void* stream) {
    return verify_two_view(0, kp_xy_dev, kp_off1_dev, kp_off2_dev, match_idx_dev, match_off_dev, match_count_dev, total_matches, intrinsics_dev, seeds_dev,
                           threshold_px, num_pairs, workspace_dev, workspace_bytes, nullptr, essential_dev, rotation_dev, translation_dev, inlier_mask_dev,
                           stats_dev, stream);
}

extern "C" int gtsfm_verify_fundamental_f64(const float* kp_xy_dev, const long long* kp_off1_dev, const long long* kp_off2_dev,
                                            const int32_t* match_idx_dev, const long long* match_off_dev, const int32_t* match_count_dev,
                                            long long total_matches, const double* intrinsics_dev, const unsigned long long* seeds_dev, double threshold_px,
                                            int num_pairs, void* workspace_dev, size_t workspace_bytes, double* fundamental_dev, double* essential_dev,
                                            double* rotation_dev, double* translation_dev, uint8_t* inlier_mask_dev, int32_t* stats_dev,
                                            void* stream) {
    return verify_two_view(1, kp_xy_dev, kp_off1_dev, kp_off2_dev, match_idx_dev, match_off_dev, match_count_dev, total_matches, intrinsics_dev, seeds_dev,
                           threshold_px, num_pairs, workspace_dev, workspace_bytes, fundamental_dev, essential_dev, rotation_dev, translation_dev,
                           inlier_mask_dev, stats_dev, stream);
}
