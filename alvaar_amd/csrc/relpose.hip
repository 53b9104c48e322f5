// f2b: two-view map initialisation = MultiViewGeometry::compute5ptEssentialMatrix
// (src/slam/src/multi_view_geometry.cpp:225-320), the call VisualFrontend::checkReadyForInit makes until the map can be
// seeded (src/slam/src/visual_frontend.cpp:517-528).  Under it, in the reference:
//   RANSAC loop              opengv/include/opengv/sac/implementation/Ransac.hpp:45-143 (adaptive k, 99 % confidence)
//   sampling                 .../SampleConsensusProblem.hpp:36-120 (std::mt19937 + prefix Fisher-Yates, 8 indices / sample)
//   hypothesis               opengv/src/sac_problems/relative_pose/CentralRelativePoseSacProblem.cpp:38-247: Nister's five-point
//                            solver (opengv/src/relative_pose/methods.cpp:239-268, modules/main.cpp:135-261, Sturm.cpp), one
//                            Levenberg-Marquardt polish per root (modules/fivept_nister/modules.cpp:517-545), SVD
//                            decomposition into 4 poses per essential matrix, disambiguation on the 8 sampled points
//   score                    :250-283: triangulate2 (opengv/src/triangulation/methods.cpp:67-90) + 2 bearing reprojections
//   refinement               opengv/src/relative_pose/methods.cpp:1082-1180: Eigen LevenbergMarquardt over (t, Cayley(R)) with
//                            a forward-difference Jacobian (unsupported/Eigen/src/NonLinearOptimization, NumericalDiff)
//
// MI355X mapping.  The sample stream is drawn on the host (it only depends on the generator).  ONE launch evaluates every
// drawn hypothesis, one wavefront each (k_relpose_hyp): the small dense algebra is spread over the 64 lanes where it is
// wide (the 10 x 20 constraint matrix: 200 coefficients; its elimination; the level-synchronous Sturm bisection: one lane per
// bracket; one lane per real root for Newton / polish / SVD; one lane per (root, decomposition) for the disambiguation) and
// then all lanes score the N correspondences.  The reference's sequential "stop after k iterations, k shrinking with the best
// inlier count" is replayed over the per-hypothesis inlier counts by one thread (k_relpose_select), which also writes the
// inlier mask of the winner.  The refinement runs as ONE workgroup that keeps the whole Levenberg-Marquardt loop on the
// device (k_relpose_lm): residuals and the forward-difference Jacobian in parallel over the inliers, Householder QR with
// column pivoting as workgroup reductions, the 6 x 6 trust-region algebra replicated in every thread.
#include "common.hpp"
#include "wave_utils.hpp"
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <ctime>
#include <limits>
#include <random>
#include <vector>

namespace {

constexpr double RP_EPS = 2.220446049250313e-16;
constexpr double RP_DBL_MIN = 2.2250738585072014e-308;
constexpr int RP_MAXR = 10;  // real roots of the tenth-degree polynomial

// ---------------------------------------------------------------------------------------------------------------------
// scoring: identical operation order to the reference's Eigen expressions, so that a model gives the same inlier set
// (fixed-size 3x3 * 3 products: rows 0,1 accumulate in order, row 2 reduces as x0 + (x1 + x2); the 3x4 * 4 product's row 2
// as (x0 + x1) + (x2 + x3))
__device__ __forceinline__ double rp_dot3(const double *a, const double *b) { return (a[0] * b[0] + a[1] * b[1]) + a[2] * b[2]; }
__device__ __forceinline__ void rp_matvec3(const double *R, const double *v, double *o) {
    o[0] = (R[0] * v[0] + R[1] * v[1]) + R[2] * v[2];
    o[1] = (R[3] * v[0] + R[4] * v[1]) + R[5] * v[2];
    o[2] = R[6] * v[0] + (R[7] * v[1] + R[8] * v[2]);
}
__device__ double rp_score(const double *R, const double *t, const double *f1, const double *f2) {
    double f2u[3], p[3], q[3], d[3];
    rp_matvec3(R, f2, f2u);
    const double b0 = rp_dot3(t, f1), b1 = rp_dot3(t, f2u);
    const double a00 = rp_dot3(f1, f1), a10 = rp_dot3(f1, f2u), a01 = -a10, a11 = -rp_dot3(f2u, f2u);
    const double invdet = 1.0 / (a00 * a11 - a10 * a01);
    const double l0 = (a11 * invdet) * b0 + (-a01 * invdet) * b1, l1 = (-a10 * invdet) * b0 + (a00 * invdet) * b1;
#pragma unroll
    for (int k = 0; k < 3; k++) p[k] = (l0 * f1[k] + (t[k] + l1 * f2u[k])) / 2;
    d[0] = -((R[0] * t[0] + R[3] * t[1]) + R[6] * t[2]);
    d[1] = -((R[1] * t[0] + R[4] * t[1]) + R[7] * t[2]);
    d[2] = -(R[2] * t[0] + (R[5] * t[1] + R[8] * t[2]));
    q[0] = ((R[0] * p[0] + R[3] * p[1]) + R[6] * p[2]) + d[0];
    q[1] = ((R[1] * p[0] + R[4] * p[1]) + R[7] * p[2]) + d[1];
    q[2] = (R[2] * p[0] + R[5] * p[1]) + (R[8] * p[2] + d[2]);
    const double n1 = sqrt(rp_dot3(p, p)), n2 = sqrt(rp_dot3(q, q));
    const double e1 = 1.0 - ((f1[0] * (p[0] / n1) + f1[1] * (p[1] / n1)) + f1[2] * (p[2] / n1));
    const double e2 = 1.0 - ((f2[0] * (q[0] / n2) + f2[1] * (q[1] / n2)) + f2[2] * (q[2] / n2));
    return e1 + e2;
}

// ---------------------------------------------------------------------------------------------------------------------
// Householder pieces in Eigen's conventions (Householder.h:65-92, :110-130); the sum of squares follows Eigen's two-packet
// SSE2 reduction because for unit bearings the FIRST pivot of the column-pivoting QR below is decided by the last bits of five
// column norms that are all 1 up to rounding, and the pivot order fixes the basis the polynomial is written in.
__device__ double rp_eig_sqnorm(const double *v, int n) {
    const int aligned2 = (n / 4) * 4, aligned = (n / 2) * 2;
    double res;
    if (aligned) {
        double p0a = v[0] * v[0], p0b = v[1] * v[1];
        if (aligned > 2) {
            double p1a = v[2] * v[2], p1b = v[3] * v[3];
            for (int i = 4; i < aligned2; i += 4) {
                p0a += v[i] * v[i];
                p0b += v[i + 1] * v[i + 1];
                p1a += v[i + 2] * v[i + 2];
                p1b += v[i + 3] * v[i + 3];
            }
            p0a += p1a;
            p0b += p1b;
            if (aligned > aligned2) {
                p0a += v[aligned2] * v[aligned2];
                p0b += v[aligned2 + 1] * v[aligned2 + 1];
            }
        }
        res = p0a + p0b;
        for (int i = aligned; i < n; i++) res += v[i] * v[i];
    } else {
        res = v[0] * v[0];
        for (int i = 1; i < n; i++) res += v[i] * v[i];
    }
    return res;
}
__device__ void rp_householder_make(double *x, int n, double *tau, double *beta) {
    const double tailSq = n == 1 ? 0.0 : rp_eig_sqnorm(x + 1, n - 1);
    const double c0 = x[0];
    if (tailSq <= RP_DBL_MIN) {
        *tau = 0;
        *beta = c0;
        for (int i = 1; i < n; i++) x[i] = 0;
    } else {
        double b = sqrt(c0 * c0 + tailSq);
        if (c0 >= 0) b = -b;
        for (int i = 1; i < n; i++) x[i] = x[i] / (c0 - b);
        *tau = (b - c0) / b;
        *beta = b;
    }
}
__device__ void rp_householder_apply(double *col, int r, const double *ess, double tau) {  // one column of length r
    if (r == 1) {
        col[0] *= 1 - tau;
        return;
    }
    if (tau == 0) return;
    double tmp = 0;
    for (int i = 1; i < r; i++) tmp += ess[i - 1] * col[i];
    tmp += col[0];
    col[0] -= tau * tmp;
    for (int i = 1; i < r; i++) col[i] -= tau * ess[i - 1] * tmp;
}
// ColPivHouseholderQR of the 9 x 5 adjoint of the scaled constraint matrix (ColPivHouseholderQR.h:478-581), serial
__device__ void rp_cpqr_9x5(double *A /* column-major 9 x 5 */, double *hc) {
    double nu[5], nd[5];
    for (int k = 0; k < 5; k++) nd[k] = nu[k] = sqrt(rp_eig_sqnorm(A + k * 9, 9));
    const double downdate = sqrt(RP_EPS);
    for (int k = 0; k < 5; k++) {
        int big = k;
        for (int j = k + 1; j < 5; j++)
            if (nu[j] > nu[big]) big = j;
        if (k != big) {
            for (int i = 0; i < 9; i++) {
                const double t = A[k * 9 + i];
                A[k * 9 + i] = A[big * 9 + i];
                A[big * 9 + i] = t;
            }
            double t = nu[k]; nu[k] = nu[big]; nu[big] = t;
            t = nd[k]; nd[k] = nd[big]; nd[big] = t;
        }
        double beta;
        rp_householder_make(A + k * 9 + k, 9 - k, &hc[k], &beta);
        A[k * 9 + k] = beta;
        for (int j = k + 1; j < 5; j++) rp_householder_apply(A + j * 9 + k, 9 - k, A + k * 9 + k + 1, hc[k]);
        for (int j = k + 1; j < 5; j++) {
            if (nu[j] != 0) {
                double temp = fabs(A[j * 9 + k]) / nu[j];
                temp = (1 + temp) * (1 - temp);
                temp = temp < 0 ? 0 : temp;
                const double q = nu[j] / nd[j];
                const double temp2 = temp * (q * q);
                if (temp2 <= downdate) {
                    nd[j] = sqrt(8 - k > 0 ? rp_eig_sqnorm(A + j * 9 + k + 1, 8 - k) : 0.0);
                    nu[j] = nd[j];
                } else
                    nu[j] *= sqrt(temp);
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Eigen::LevenbergMarquardt pieces for N unknowns (lmpar.h:160-296, qrsolv.h:17-88, Jacobi.h makeGivens); all N x N
// matrices column-major m[j][i] = M(i, j).
// Jacobi.h makeGivens (real case).  Its two branches (|p| > |q| or not) both reduce to c = p / r, s = -q / r with r = |(p, q)|;
// one reciprocal square root (hardware estimate + two Newton steps) replaces a division, a square root and a second division.
// This algebra feeds a trust-region search that runs at its rounding-noise floor anyway (DESIGN.md, f2b note).
__device__ __forceinline__ double rp_rsqrt(double x) {
    double y = __builtin_amdgcn_rsq(x);
    y = y * (1.5 - 0.5 * x * y * y);
    y = y * (1.5 - 0.5 * x * y * y);
    return y;
}
__device__ __forceinline__ void rp_givens(double p, double q, double &c, double &s) {
    if (q == 0) {
        c = p < 0 ? -1 : 1;
        s = 0;
    } else if (p == 0) {
        c = 0;
        s = q < 0 ? 1 : -1;
    } else {
        const double ir = rp_rsqrt(p * p + q * q);
        c = p * ir;
        s = -q * ir;
    }
}
template <int N> __device__ double rp_norm(const double (&v)[N]) {
    double s = 0;
#pragma unroll
    for (int i = 0; i < N; i++) s += v[i] * v[i];
    return sqrt(s);
}
template <int N>
__device__ void rp_qrsolv(double (&s)[N][N], const int (&ipvt)[N], const double (&diag)[N], const double (&qtb)[N], double (&x)[N],
                          double (&sdiag)[N]) {
    double wa[N];
#pragma unroll
    for (int j = 0; j < N; j++) {
        x[j] = s[j][j];
        wa[j] = qtb[j];
#pragma unroll
        for (int i = j + 1; i < N; i++) s[j][i] = s[i][j];
    }
#pragma unroll
    for (int j = 0; j < N; j++) {
        double dl = 0;
#pragma unroll
        for (int q = 0; q < N; q++) dl = ipvt[j] == q ? diag[q] : dl;
        if (dl == 0) break;
#pragma unroll
        for (int k = j; k < N; k++) sdiag[k] = 0;
        sdiag[j] = dl;
        double qtbpj = 0;
#pragma unroll
        for (int k = j; k < N; k++) {
            double c, sn;
            rp_givens(-s[k][k], sdiag[k], c, sn);
            s[k][k] = c * s[k][k] + sn * sdiag[k];
            double temp = c * wa[k] + sn * qtbpj;
            qtbpj = -sn * wa[k] + c * qtbpj;
            wa[k] = temp;
#pragma unroll
            for (int i = k + 1; i < N; i++) {
                temp = c * s[k][i] + sn * sdiag[i];
                sdiag[i] = -sn * s[k][i] + c * sdiag[i];
                s[k][i] = temp;
            }
        }
    }
    int nsing = 0;
#pragma unroll
    for (int j = 0; j < N; j++)
        if (nsing == j && sdiag[j] != 0) nsing = j + 1;
#pragma unroll
    for (int j = 0; j < N; j++)
        if (j >= nsing) wa[j] = 0;
#pragma unroll
    for (int i = N - 1; i >= 0; i--) {
        if (i < nsing) {
            double sum = wa[i];
#pragma unroll
            for (int j = i + 1; j < N; j++)
                if (j < nsing) sum -= s[i][j] * wa[j];
            wa[i] = sum / s[i][i];
        }
    }
#pragma unroll
    for (int j = 0; j < N; j++) {
        const double d = s[j][j];
        s[j][j] = x[j];
        sdiag[j] = d;
    }
#pragma unroll
    for (int j = 0; j < N; j++)
#pragma unroll
        for (int q = 0; q < N; q++)
            if (ipvt[j] == q) x[q] = wa[j];
}
template <int N> __device__ __forceinline__ double rp_pick(const double (&v)[N], int idx) {
    double r = 0;
#pragma unroll
    for (int q = 0; q < N; q++) r = idx == q ? v[q] : r;
    return r;
}
template <int N>
__device__ void rp_lmpar(const double (&R)[N][N], int rank, const int (&ipvt)[N], const double (&diag)[N], const double (&qtb)[N],
                         double delta, double &par, double (&x)[N]) {
    double wa1[N], wa2[N], sdiag[N], s[N][N];
#pragma unroll
    for (int j = 0; j < N; j++) wa1[j] = j < rank ? qtb[j] : 0;
#pragma unroll
    for (int i = N - 1; i >= 0; i--) {
        if (i < rank) {
            double sum = wa1[i];
#pragma unroll
            for (int j = i + 1; j < N; j++)
                if (j < rank) sum -= R[j][i] * wa1[j];
            wa1[i] = sum / R[i][i];
        }
    }
#pragma unroll
    for (int j = 0; j < N; j++)
#pragma unroll
        for (int q = 0; q < N; q++)
            if (ipvt[j] == q) x[q] = wa1[j];
    int iter = 0;
#pragma unroll
    for (int j = 0; j < N; j++) wa2[j] = diag[j] * x[j];
    double dxnorm = rp_norm<N>(wa2), fp = dxnorm - delta;
    if (fp <= 0.1 * delta) {
        par = 0;
        return;
    }
    double parl = 0;
    if (rank == N) {
#pragma unroll
        for (int j = 0; j < N; j++) wa1[j] = rp_pick<N>(diag, ipvt[j]) * rp_pick<N>(wa2, ipvt[j]) / dxnorm;
#pragma unroll
        for (int j = 0; j < N; j++) {
            double sum = wa1[j];
#pragma unroll
            for (int i = 0; i < j; i++) sum -= R[j][i] * wa1[i];
            wa1[j] = sum / R[j][j];
        }
        const double temp = rp_norm<N>(wa1);
        parl = fp / delta / temp / temp;
    }
#pragma unroll
    for (int j = 0; j < N; j++) {
        double sum = 0;
#pragma unroll
        for (int i = 0; i <= j; i++) sum += R[j][i] * qtb[i];
        wa1[j] = sum / rp_pick<N>(diag, ipvt[j]);
    }
    const double gnorm = rp_norm<N>(wa1);
    double paru = gnorm / delta;
    if (paru == 0) paru = RP_DBL_MIN / fmin(delta, 0.1);
    par = fmax(par, parl);
    par = fmin(par, paru);
    if (par == 0) par = gnorm / dxnorm;
#pragma unroll
    for (int j = 0; j < N; j++)
#pragma unroll
        for (int i = 0; i < N; i++) s[j][i] = R[j][i];
    for (;;) {
        ++iter;
        if (par == 0) par = fmax(RP_DBL_MIN, 0.001 * paru);
        const double sp = sqrt(par);
#pragma unroll
        for (int j = 0; j < N; j++) wa1[j] = sp * diag[j];
        rp_qrsolv<N>(s, ipvt, wa1, qtb, x, sdiag);
#pragma unroll
        for (int j = 0; j < N; j++) wa2[j] = diag[j] * x[j];
        dxnorm = rp_norm<N>(wa2);
        double temp = fp;
        fp = dxnorm - delta;
        if (fabs(fp) <= 0.1 * delta || (parl == 0 && fp <= temp && temp < 0) || iter == 10) break;
#pragma unroll
        for (int j = 0; j < N; j++) wa1[j] = rp_pick<N>(diag, ipvt[j]) * (rp_pick<N>(wa2, ipvt[j]) / dxnorm);
#pragma unroll
        for (int j = 0; j < N; j++) {
            wa1[j] /= sdiag[j];
            temp = wa1[j];
#pragma unroll
            for (int i = j + 1; i < N; i++) wa1[i] -= s[j][i] * temp;
        }
        temp = rp_norm<N>(wa1);
        const double parc = fp / delta / temp / temp;
        if (fp > 0) parl = fmax(parl, par);
        if (fp < 0) paru = fmin(paru, par);
        par = fmax(parl, par + parc);
    }
    if (iter == 0) par = 0;
}
// Trust-region bookkeeping of LevenbergMarquardt::minimizeOneStep once a trial point has been evaluated
// (LevenbergMarquardt.h:283-313) and its termination tests (:326-348).
struct RpLm {
    double par, delta, xnorm, fnorm, temp;
    int iter, nfev;
};
template <int N>
__device__ void rp_lm_ratio(RpLm &S, const double (&R)[N][N], const int (&ipvt)[N], const double (&wa1)[N], double pnorm, double fnorm1,
                            double &actred, double &prered, double &ratio) {
    actred = -1;
    if (0.1 * fnorm1 < S.fnorm) actred = 1 - (fnorm1 / S.fnorm) * (fnorm1 / S.fnorm);
    double wa3[N];
#pragma unroll
    for (int i = 0; i < N; i++) {
        double sum = 0;
#pragma unroll
        for (int j = i; j < N; j++) sum += R[j][i] * rp_pick<N>(wa1, ipvt[j]);
        wa3[i] = sum;
    }
    const double t1 = rp_norm<N>(wa3) / S.fnorm, t2 = sqrt(S.par) * pnorm / S.fnorm;
    const double temp1 = t1 * t1, temp2 = t2 * t2;
    prered = temp1 + temp2 / 0.5;
    const double dirder = -(temp1 + temp2);
    ratio = 0;
    if (prered != 0) ratio = actred / prered;
    if (ratio <= 0.25) {
        if (actred >= 0) S.temp = 0.5;
        if (actred < 0) S.temp = 0.5 * dirder / (dirder + 0.5 * actred);
        if (0.1 * fnorm1 >= S.fnorm || S.temp < 0.1) S.temp = 0.1;
        S.delta = S.temp * fmin(S.delta, pnorm / 0.1);
        S.par /= S.temp;
    } else if (!(S.par != 0 && ratio < 0.75)) {
        S.delta = pnorm / 0.5;
        S.par = 0.5 * S.par;
    }
}
__device__ int rp_lm_tests(const RpLm &S, double actred, double prered, double ratio, double gnorm, double ftol, double xtol, int maxfev) {
    const bool small = fabs(actred) <= ftol && prered <= ftol && 0.5 * ratio <= 1;
    if (small && S.delta <= xtol * S.xnorm) return 1;
    if (small) return 2;
    if (S.delta <= xtol * S.xnorm) return 3;
    if (S.nfev >= maxfev) return 5;
    if (fabs(actred) <= RP_EPS && prered <= RP_EPS && 0.5 * ratio <= 1) return 6;
    if (S.delta <= RP_EPS * S.xnorm) return 7;
    if (gnorm <= RP_EPS) return 8;
    return 0;
}

// ---------------------------------------------------------------------------------------------------------------------
// polish of one root (modules.cpp:517-545): LevenbergMarquardt<NumericalDiff<>> on the ten cubic constraints, three
// unknowns, maxfev = 5 => exactly one trust-region trial.  Runs in ONE lane; A is read from LDS.
__device__ void rp_constraints(const double *A, const double (&x)[3], double (&f)[10]) {
    const double X = x[0], Y = x[1], Z = x[2];
    const double mono[20] = {X * X * X, Y * Y * Y, X * X * Y, X * Y * Y, X * X * Z, X * X, Y * Y * Z, Y * Y, X * Y * Z, X * Y,
                             X * Z * Z, X * Z,     X,         Y * Z * Z, Y * Z,     Y,     Z * Z * Z, Z * Z, Z,         1.0};
#pragma unroll
    for (int i = 0; i < 10; i++) {
        double s = 0;
#pragma unroll
        for (int m = 0; m < 20; m++) s += A[20 * i + m] * mono[m];
        f[i] = s;
    }
}
__device__ double rp_norm10(const double (&v)[10]) {
    double s = 0;
#pragma unroll
    for (int i = 0; i < 10; i++) s += v[i] * v[i];
    return sqrt(s);
}
__device__ void rp_polish(const double *A, double (&x)[3]) {
    constexpr int N = 3, M = 10;
    const double ftol = 1e10 * RP_EPS, xtol = 1e10 * RP_EPS;
    const int maxfev = 5;
    double fvec[M], J[N][M], wa4[M], val1[M], val2[M];
    RpLm S{};
    S.nfev = 1;
    S.iter = 1;
    rp_constraints(A, x, fvec);
    S.fnorm = rp_norm10(fvec);
    double diag[N] = {0, 0, 0};
    for (int guard = 0; guard < 8; guard++) {
        // forward differences (NumericalDiff.h:63-121)
        rp_constraints(A, x, val1);
        S.nfev++;
        const double fdeps = sqrt(RP_EPS);
#pragma unroll
        for (int j = 0; j < N; j++) {
            double h = fdeps * fabs(x[j]);
            if (h == 0) h = fdeps;
            double xt[3] = {x[0], x[1], x[2]};
            xt[j] += h;
            rp_constraints(A, xt, val2);
            S.nfev++;
#pragma unroll
            for (int i = 0; i < M; i++) J[j][i] = (val2[i] - val1[i]) / h;
        }
        double wa2[N];
#pragma unroll
        for (int j = 0; j < N; j++) wa2[j] = rp_norm10(J[j]);
        // column-pivoting Householder QR of the 10 x 3 Jacobian, applied to fvec at the same time
        int perm[N] = {0, 1, 2};
        double nu[N], nd[N], Rm[N][N] = {}, qtf[N];
#pragma unroll
        for (int i = 0; i < M; i++) wa4[i] = fvec[i];
#pragma unroll
        for (int k = 0; k < N; k++) nd[k] = nu[k] = sqrt(rp_eig_sqnorm(J[k], M));
        const double maxn = fmax(nu[0], fmax(nu[1], nu[2]));
        const double th = (maxn * RP_EPS) * (maxn * RP_EPS) / (double) M, downdate = sqrt(RP_EPS);
        int nonzero = N;
        double maxpivot = 0;
#pragma unroll
        for (int k = 0; k < N; k++) {
            int big = k;
#pragma unroll
            for (int j = k + 1; j < N; j++)
                if (nu[j] > nu[big]) big = j;
            if (nonzero == N && nu[big] * nu[big] < th * (double) (M - k)) nonzero = k;
#pragma unroll
            for (int j = k + 1; j < N; j++)
                if (big == j) {
#pragma unroll
                    for (int i = 0; i < M; i++) {
                        const double t = J[k][i];
                        J[k][i] = J[j][i];
                        J[j][i] = t;
                    }
#pragma unroll
                    for (int i = 0; i < N; i++) {  // rows above k of R travel with the column
                        const double t = Rm[k][i];
                        Rm[k][i] = Rm[j][i];
                        Rm[j][i] = t;
                    }
                    double t = nu[k]; nu[k] = nu[j]; nu[j] = t;
                    t = nd[k]; nd[k] = nd[j]; nd[j] = t;
                    const int ti = perm[k]; perm[k] = perm[j]; perm[j] = ti;
                }
            const double tailSq = rp_eig_sqnorm(&J[k][k + 1], M - k - 1);
            const double c0 = J[k][k];
            double tau, beta;
            if (tailSq <= RP_DBL_MIN) {
                tau = 0;
                beta = c0;
#pragma unroll
                for (int i = k + 1; i < M; i++) J[k][i] = 0;
            } else {
                beta = sqrt(c0 * c0 + tailSq);
                if (c0 >= 0) beta = -beta;
#pragma unroll
                for (int i = k + 1; i < M; i++) J[k][i] = J[k][i] / (c0 - beta);
                tau = (beta - c0) / beta;
            }
            Rm[k][k] = beta;
            if (fabs(beta) > maxpivot) maxpivot = fabs(beta);
            if (tau != 0) {
#pragma unroll
                for (int j = k + 1; j <= N; j++) {  // j == N: the residual vector
                    double *col = j < N ? J[j < N ? j : 0] : wa4;
                    double tmp = 0;
#pragma unroll
                    for (int i = k + 1; i < M; i++) tmp += J[k][i] * col[i];
                    tmp += col[k];
                    col[k] -= tau * tmp;
#pragma unroll
                    for (int i = k + 1; i < M; i++) col[i] -= tau * J[k][i] * tmp;
                }
            }
#pragma unroll
            for (int j = k + 1; j < N; j++) {
                Rm[j][k] = J[j][k];
                if (nu[j] != 0) {
                    double temp = fabs(J[j][k]) / nu[j];
                    temp = (1 + temp) * (1 - temp);
                    temp = temp < 0 ? 0 : temp;
                    const double q = nu[j] / nd[j];
                    if (temp * (q * q) <= downdate) {
                        nd[j] = nu[j] = sqrt(rp_eig_sqnorm(&J[j][k + 1], M - k - 1));
                    } else
                        nu[j] *= sqrt(temp);
                }
            }
            qtf[k] = wa4[k];
        }
        int rank = 0;
        {
            const double pm = maxpivot * (RP_EPS * (double) N);
#pragma unroll
            for (int i = 0; i < N; i++) rank += (i < nonzero && fabs(Rm[i][i]) > pm) ? 1 : 0;
        }
        if (S.iter == 1) {
#pragma unroll
            for (int j = 0; j < N; j++) diag[j] = wa2[j] == 0 ? 1 : wa2[j];
            double t[N];
#pragma unroll
            for (int j = 0; j < N; j++) t[j] = diag[j] * x[j];
            S.xnorm = rp_norm<N>(t);
            S.delta = 100.0 * S.xnorm;
            if (S.delta == 0) S.delta = 100.0;
        }
        double gnorm = 0;
        if (S.fnorm != 0) {
#pragma unroll
            for (int j = 0; j < N; j++) {
                const double w = rp_pick<N>(wa2, perm[j]);
                if (w != 0) {
                    double sum = 0;
#pragma unroll
                    for (int i = 0; i <= j; i++) sum += Rm[j][i] * (qtf[i] / S.fnorm);
                    gnorm = fmax(gnorm, fabs(sum / w));
                }
            }
        }
        if (gnorm <= 0.0) return;
#pragma unroll
        for (int j = 0; j < N; j++) diag[j] = fmax(diag[j], wa2[j]);
        double ratio;
        int status;
        do {
            double wa1[N], xn[N], t[N];
            rp_lmpar<N>(Rm, rank, perm, diag, qtf, S.delta, S.par, wa1);
#pragma unroll
            for (int j = 0; j < N; j++) {
                wa1[j] = -wa1[j];
                xn[j] = x[j] + wa1[j];
                t[j] = diag[j] * wa1[j];
            }
            const double pnorm = rp_norm<N>(t);
            if (S.iter == 1) S.delta = fmin(S.delta, pnorm);
            rp_constraints(A, xn, wa4);
            S.nfev++;
            const double fnorm1 = rp_norm10(wa4);
            double actred, prered;
            rp_lm_ratio<N>(S, Rm, perm, wa1, pnorm, fnorm1, actred, prered, ratio);
            if (ratio >= 1e-4) {
#pragma unroll
                for (int j = 0; j < N; j++) {
                    x[j] = xn[j];
                    t[j] = diag[j] * x[j];
                }
#pragma unroll
                for (int i = 0; i < M; i++) fvec[i] = wa4[i];
                S.xnorm = rp_norm<N>(t);
                S.fnorm = fnorm1;
                S.iter++;
            }
            status = rp_lm_tests(S, actred, prered, ratio, gnorm, ftol, xtol, maxfev);
        } while (!status && ratio < 1e-4);
        if (status) return;
    }
}

// 3 x 3 SVD by one-sided Jacobi; U's third column completed as +-(u0 x u1) (the smallest singular value of an essential
// matrix is ~0).  Any SVD yields the same set of four decompositions.
__device__ void rp_svd3(const double *E, double (&U)[9], double (&s)[3], double (&V)[9]) {
    double a[3][3], v[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
#pragma unroll
    for (int r = 0; r < 3; r++)
#pragma unroll
        for (int c = 0; c < 3; c++) a[r][c] = E[3 * r + c];
    for (int sweep = 0; sweep < 60; sweep++) {
        double off = 0;
#pragma unroll
        for (int pq = 0; pq < 3; pq++) {
            const int p = pq == 2 ? 1 : 0, q = pq == 0 ? 1 : 2;
            double alpha = 0, beta = 0, gamma = 0;
#pragma unroll
            for (int r = 0; r < 3; r++) {
                alpha += a[r][p] * a[r][p];
                beta += a[r][q] * a[r][q];
                gamma += a[r][p] * a[r][q];
            }
            if (fabs(gamma) <= 1e-300 || fabs(gamma) <= 1e-17 * sqrt(alpha * beta)) continue;
            off = fmax(off, fabs(gamma) / sqrt(alpha * beta));
            const double zeta = (beta - alpha) / (2 * gamma);
            const double t = (zeta >= 0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1 + zeta * zeta));
            const double c = 1 / sqrt(1 + t * t), sn = c * t;
#pragma unroll
            for (int r = 0; r < 3; r++) {
                const double x = a[r][p], y = a[r][q];
                a[r][p] = c * x - sn * y;
                a[r][q] = sn * x + c * y;
                const double vx = v[r][p], vy = v[r][q];
                v[r][p] = c * vx - sn * vy;
                v[r][q] = sn * vx + c * vy;
            }
        }
        if (off < 1e-16) break;
    }
    double nrm[3];
#pragma unroll
    for (int c = 0; c < 3; c++) nrm[c] = sqrt(a[0][c] * a[0][c] + a[1][c] * a[1][c] + a[2][c] * a[2][c]);
    int o0 = 0, o1 = 1, o2 = 2;
    if (nrm[o1] > nrm[o0]) { const int t = o0; o0 = o1; o1 = t; }
    if (nrm[o2] > nrm[o0]) { const int t = o0; o0 = o2; o2 = t; }
    if (nrm[o2] > nrm[o1]) { const int t = o1; o1 = o2; o2 = t; }
    const int ord[3] = {o0, o1, o2};
#pragma unroll
    for (int k = 0; k < 3; k++) {
        const int c = ord[k];
        double nc = 0, ac[3] = {0, 0, 0}, vc[3] = {0, 0, 0};
#pragma unroll
        for (int q = 0; q < 3; q++)
            if (c == q) {
                nc = nrm[q];
#pragma unroll
                for (int r = 0; r < 3; r++) {
                    ac[r] = a[r][q];
                    vc[r] = v[r][q];
                }
            }
        s[k] = nc;
#pragma unroll
        for (int r = 0; r < 3; r++) {
            V[3 * r + k] = vc[r];
            U[3 * r + k] = nc > 1e-300 ? ac[r] / nc : 0;
        }
    }
    const double u0[3] = {U[0], U[3], U[6]}, u1[3] = {U[1], U[4], U[7]};
    const double c[3] = {u0[1] * u1[2] - u0[2] * u1[1], u0[2] * u1[0] - u0[0] * u1[2], u0[0] * u1[1] - u0[1] * u1[0]};
    const double sg = (c[0] * U[2] + c[1] * U[5] + c[2] * U[8]) < 0 ? -1.0 : 1.0;
    U[2] = sg * c[0];
    U[5] = sg * c[1];
    U[8] = sg * c[2];
}
__device__ __forceinline__ double rp_det3(const double *M) {
    return M[0] * (M[4] * M[8] - M[5] * M[7]) - M[1] * (M[3] * M[8] - M[5] * M[6]) + M[2] * (M[3] * M[7] - M[4] * M[6]);
}
__device__ __forceinline__ void rp_mat3mul(const double *A, const double *B, double *C) {
#pragma unroll
    for (int r = 0; r < 3; r++)
#pragma unroll
        for (int c = 0; c < 3; c++) C[3 * r + c] = (A[3 * r] * B[c] + A[3 * r + 1] * B[3 + c]) + A[3 * r + 2] * B[6 + c];
}


// ---------------------------------------------------------------------------------------------------------------------
// k_relpose_hyp: one wavefront per drawn sample
struct RelposeArgs {
    const double *bv1, *bv2;  // [n][3] unit bearings in the previous keyframe / the current frame
    const int *samples;       // [H][8]
    int n, H;
    double threshold;
    double *models;  // [H][12]
    int *counts;     // [H]: inliers of the hypothesis, -1 when the sample gave no model
};

// quadratic monomials of (x, y, z, w): index of the unordered pair (i <= j)
__device__ __forceinline__ int rp_pair(int i, int j) { return i * 4 - (i * (i - 1)) / 2 + (j - i); }
// the twenty cubic monomials in the column order of the constraint matrix (modules.cpp:484-503), as sorted variable triples
__constant__ signed char RP_TRI[20][3] = {{0, 0, 0}, {1, 1, 1}, {0, 0, 1}, {0, 1, 1}, {0, 0, 2}, {0, 0, 3}, {1, 1, 2}, {1, 1, 3}, {0, 1, 2}, {0, 1, 3},
                                          {0, 2, 2}, {0, 2, 3}, {0, 3, 3}, {1, 2, 2}, {1, 2, 3}, {1, 3, 3}, {2, 2, 2}, {2, 2, 3}, {2, 3, 3}, {3, 3, 3}};
// product of the linear forms a, b (coefficients of x, y, z, 1): coefficient of the pair monomial q
__device__ __forceinline__ double rp_lin2(const double *a, const double *b, int i, int j) {
    return i == j ? a[i] * b[i] : a[i] * b[j] + a[j] * b[i];
}
// coefficient of the cubic monomial (i <= j <= k) in quad * lin
__device__ __forceinline__ double rp_quadlin(const double *quad, const double *lin, int i, int j, int k) {
    double s = quad[rp_pair(j, k)] * lin[i];
    if (j != i) s += quad[rp_pair(i, k)] * lin[j];
    if (k != j) s += quad[rp_pair(i, j)] * lin[k];
    return s;
}

struct RpShared {
    double sb1[8][3], sb2[8][3];
    double qr[45], hc[5], EE[9][4];
    double G[9][10], minor[3][10], tr[10];
    double A[200], M[200], f[10], A3[10][10];
    double b[3][3][5], p1[8], p2[8], p3[7], p10[11];
    double C[11][11];
    double blo[2][64], bhi[2][64];
    int bloCh[2][64], bhiCh[2][64];
    double roots[RP_MAXR];
    double cand[RP_MAXR * 4][12];
    double best[12];
    int colp[10], idx[8], nroots, have;
};

__device__ int rp_chain(const double (*C)[11], double bound) {  // Sturm::evaluateChain2 (Sturm.cpp:394-442)
    double mono[11];
    mono[10] = 1.0;
#pragma unroll
    for (int i = 2; i <= 11; i++) mono[11 - i] = mono[11 - i + 1] * bound;
    int positive = 0, changes = 0;
#pragma unroll
    for (int i = 0; i < 11; i++) {
        double sign = 0.0;
#pragma unroll
        for (int j = i; j < 11; j++) sign += C[i][j] * mono[j];
        if (i == 0) positive = sign > 0.0;
        else if (positive) {
            if (sign < 0.0) { changes++; positive = 0; }
        } else if (sign > 0.0) { changes++; positive = 1; }
    }
    return changes;
}
__device__ __forceinline__ double rp_polyval(const double *p, int ncoef, double x) {  // polyVal (modules.cpp:371-382)
    double v = 0;
    for (int power = ncoef; power > 0; power--) {
        double pw = 1.0;  // pow(x, power - 1) by repeated multiplication
        for (int e = 0; e < power - 1; e++) pw *= x;
        v += p[ncoef - power] * pw;
    }
    return v;
}
__device__ __forceinline__ void rp_conv(const double *a, int na, const double *b, int nb, int k, double &o) {
    double s = 0;
    bool first = true;
    for (int i = 0; i < na; i++) {
        const int j = k - i;
        if (j < 0 || j >= nb) continue;
        if (first) { s = a[i] * b[j]; first = false; }
        else s += a[i] * b[j];
    }
    o = s;
}

__global__ __launch_bounds__(64) void k_relpose_hyp(const RelposeArgs P) {
    __shared__ RpShared sh;
    const int h = blockIdx.x, lane = threadIdx.x;
    // ---- the sample and its 8 correspondences
    if (lane < 8) sh.idx[lane] = P.samples[8 * h + lane];
    __syncthreads();
    if (lane < 48) {
        const int which = lane / 24, r = (lane % 24) / 3, c = lane % 3;
        const double v = (which ? P.bv2 : P.bv1)[3 * (size_t) sh.idx[r] + c];
        if (which) sh.sb2[r][c] = v;
        else sh.sb1[r][c] = v;
    }
    __syncthreads();
    // ---- adjoint of the 5 x 9 epipolar constraint matrix (methods.cpp:246-260; the solver works on the inverse transformation),
    //      scaled by its largest |entry| as JacobiSVD does
    {
        double v = 0;
        if (lane < 45) {
            const int i = lane / 9, j = lane % 9, a = j / 3, c = j % 3;
            v = sh.sb2[i][c] * sh.sb1[i][a];
        }
        double mx = fabs(v);
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) mx = fmax(mx, __shfl_xor(mx, d));
        if (mx == 0) mx = 1;
        if (lane < 45) sh.qr[lane] = v / mx;
    }
    __syncthreads();
    if (lane == 0) rp_cpqr_9x5(sh.qr, sh.hc);
    __syncthreads();
    if (lane < 4) {  // column 5 + lane of the full Householder Q (HouseholderSequence::evalTo)
        double v[9];
#pragma unroll
        for (int i = 0; i < 9; i++) v[i] = i == 5 + lane ? 1.0 : 0.0;
        for (int k = 4; k >= 0; k--) rp_householder_apply(v + k, 9 - k, sh.qr + k * 9 + k + 1, sh.hc[k]);
#pragma unroll
        for (int i = 0; i < 9; i++) sh.EE[i][lane] = v[i];
    }
    __syncthreads();
    // ---- the ten cubic constraints on E = x E0 + y E1 + z E2 + E3: quadratic stage (E E^T and the 2 x 2 minors of row 0) ...
    for (int o = lane; o < 120; o += 64) {
        if (o < 90) {
            const int rc = o / 10, q = o % 10, r = rc / 3, c = rc % 3;
            int pi = 0, pj = 0;
            for (int i = 0, cnt = 0; i < 4; i++)
                for (int j = i; j < 4; j++, cnt++)
                    if (cnt == q) { pi = i; pj = j; }
            double s = 0;
#pragma unroll
            for (int k = 0; k < 3; k++) s += rp_lin2(sh.EE[3 * r + k], sh.EE[3 * c + k], pi, pj);
            sh.G[rc][q] = s;
        } else {
            const int j = (o - 90) / 10, q = (o - 90) % 10;
            const int cof[3][4] = {{4, 8, 5, 7}, {5, 6, 3, 8}, {3, 7, 4, 6}};
            int pi = 0, pj = 0;
            for (int i = 0, cnt = 0; i < 4; i++)
                for (int jj = i; jj < 4; jj++, cnt++)
                    if (cnt == q) { pi = i; pj = jj; }
            sh.minor[j][q] = rp_lin2(sh.EE[cof[j][0]], sh.EE[cof[j][1]], pi, pj) - rp_lin2(sh.EE[cof[j][2]], sh.EE[cof[j][3]], pi, pj);
        }
    }
    __syncthreads();
    if (lane < 10) sh.tr[lane] = (sh.G[0][lane] + sh.G[4][lane]) + sh.G[8][lane];
    __syncthreads();
    // ... cubic stage: row 0 = det E, row 1 + 3c + r = (2 E E^T E - tr(E E^T) E)_rc
    for (int o = lane; o < 200; o += 64) {
        const int row = o / 20, m = o % 20;
        const int i = RP_TRI[m][0], j = RP_TRI[m][1], k = RP_TRI[m][2];
        double s = 0;
        if (row == 0) {
#pragma unroll
            for (int q = 0; q < 3; q++) s += rp_quadlin(sh.minor[q], sh.EE[q], i, j, k);
        } else {
            const int c = (row - 1) / 3, r = (row - 1) % 3;
#pragma unroll
            for (int q = 0; q < 3; q++) s += 2.0 * rp_quadlin(sh.G[3 * r + q], sh.EE[3 * q + c], i, j, k);
            s -= rp_quadlin(sh.tr, sh.EE[3 * r + c], i, j, k);
        }
        sh.A[o] = s;
        sh.M[o] = s;
    }
    if (lane < 10) sh.colp[lane] = lane;
    __syncthreads();
    // ---- A3 = A1^-1 A2 (main.cpp:143-147) by elimination with full pivoting
    bool singular = false;
    for (int k = 0; k < 10; k++) {
        double best = -1.0;
        int key = 1 << 20;
        for (int o = lane; o < 100; o += 64) {
            const int c = o / 10, r = o % 10;  // column-major scan order: the first maximum wins
            if (c >= k && r >= k) {
                const double a = fabs(sh.M[20 * r + c]);
                if (a > best) { best = a; key = o; }
            }
        }
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) {
            const double ob = __shfl_xor(best, d);
            const int ok = __shfl_xor(key, d);
            if (ob > best || (ob == best && ok < key)) { best = ob; key = ok; }
        }
        if (!(best > 0)) { singular = true; break; }
        const int pc = key / 10, pr = key % 10;
        if (pr != k && lane < 20) {
            const double t = sh.M[20 * k + lane];
            sh.M[20 * k + lane] = sh.M[20 * pr + lane];
            sh.M[20 * pr + lane] = t;
        }
        __syncthreads();
        if (pc != k) {
            if (lane < 10) {
                const double t = sh.M[20 * lane + k];
                sh.M[20 * lane + k] = sh.M[20 * lane + pc];
                sh.M[20 * lane + pc] = t;
            }
            if (lane == 10) { const int t = sh.colp[k]; sh.colp[k] = sh.colp[pc]; sh.colp[pc] = t; }
        }
        __syncthreads();
        if (lane > k && lane < 10) sh.f[lane] = sh.M[20 * lane + k] / sh.M[20 * k + k];
        __syncthreads();
        for (int o = lane; o < 200; o += 64) {
            const int r = o / 20, c = o % 20;
            if (r > k && c > k) sh.M[o] -= sh.f[r] * sh.M[20 * k + c];
        }
        __syncthreads();
    }
    if (!singular && lane < 10) {  // back substitution, one right-hand side per lane
        double y[10];
        for (int r = 9; r >= 0; r--) {
            double s = sh.M[20 * r + 10 + lane];
            for (int j = r + 1; j < 10; j++) s -= sh.M[20 * r + j] * y[j];
            y[r] = s / sh.M[20 * r + r];
        }
        for (int r = 0; r < 10; r++) sh.A3[sh.colp[r]][lane] = y[r];
    }
    __syncthreads();
    int nroots = 0;
    if (!singular) {
        // ---- the univariate polynomial (main.cpp:149-223)
        if (lane < 45) {
            const int rp = lane / 15, g = (lane % 15) / 5, k = lane % 5;
            const int c0 = g == 0 ? 0 : (g == 1 ? 3 : 6), w = g == 2 ? 4 : 3;
            double v = 0;
            if (k <= w) {
                const double part1 = k >= 1 ? sh.A3[4 + 2 * rp][c0 + k - 1] : 0.0;
                const double part2 = k < w ? sh.A3[5 + 2 * rp][c0 + k] : 0.0;
                v = part1 - part2;
            }
            sh.b[rp][g][k] = v;
        }
        __syncthreads();
        if (lane < 8) {
            double t1, t2;
            rp_conv(sh.b[1][2], 5, sh.b[0][1], 4, lane, t1);
            rp_conv(sh.b[0][2], 5, sh.b[1][1], 4, lane, t2);
            sh.p1[lane] = t1 - t2;
            rp_conv(sh.b[0][2], 5, sh.b[1][0], 4, lane, t1);
            rp_conv(sh.b[1][2], 5, sh.b[0][0], 4, lane, t2);
            sh.p2[lane] = t1 - t2;
            if (lane < 7) {
                rp_conv(sh.b[0][0], 4, sh.b[1][1], 4, lane, t1);
                rp_conv(sh.b[0][1], 4, sh.b[1][0], 4, lane, t2);
                sh.p3[lane] = t1 - t2;
            }
        }
        __syncthreads();
        if (lane < 11) {
            double q1, q2, q3;
            rp_conv(sh.p1, 8, sh.b[2][0], 4, lane, q1);
            rp_conv(sh.p2, 8, sh.b[2][1], 4, lane, q2);
            rp_conv(sh.p3, 7, sh.b[2][2], 5, lane, q3);
            sh.p10[lane] = (q1 + q2) + q3;
        }
        __syncthreads();
        // ---- Sturm sequence (Sturm.cpp:150-172, :445-464)
        for (int o = lane; o < 121; o += 64) sh.C[o / 11][o % 11] = 0.0;
        __syncthreads();
        if (lane < 11) {
            sh.C[0][lane] = sh.p10[lane];
            if (lane >= 1) sh.C[1][lane] = sh.p10[lane - 1] * (double) (11 - lane);
        }
        __syncthreads();
        for (int i = 2; i < 11; i++) {
            const double *p1 = &sh.C[i - 2][i - 2], *p2 = &sh.C[i - 1][i - 1];
            const int n1 = 11 - (i - 2), n2 = n1 - 1;
            double r = 0;
            if (lane >= 2 && lane < n1) {
                const double f1 = p1[0] / p2[0], f2 = p1[1] / p2[0], f3 = (-p2[1] * p1[0]) / (p2[0] * p2[0]);
                const int k = lane;
                const double a = k < n2 ? f1 * p2[k] : 0.0, b = f2 * p2[k - 1], c = f3 * p2[k - 1];
                r = ((-p1[k] + a) + b) + c;
            }
            __syncthreads();
            if (lane >= 2 && lane < n1) sh.C[i][i + lane - 2] = r;
            __syncthreads();
        }
        // ---- root bracketing: the reference's FIFO bisection (Sturm.cpp:303-350) processed level by level, one lane per bracket
        if (lane == 0) {
            // computeLagrangianBound (:466-492)
            double c[10];
            for (int i = 0; i < 10; i++) c[i] = pow(fabs(sh.C[0][i + 1] / sh.C[0][0]), 1.0 / (double) (i + 1));
            int j = 0;
            double max1 = -1.0, max2 = -1.0;
            for (int i = 0; i < 10; i++)
                if (c[i] > max1) { j = i; max1 = c[i]; }
            for (int i = 0; i < 10; i++)
                if (i != j && c[i] > max2) max2 = c[i];
            const double bound = max1 + max2;
            sh.blo[0][0] = -bound;
            sh.bhi[0][0] = bound;
            sh.bloCh[0][0] = rp_chain(sh.C, -bound);
            sh.bhiCh[0][0] = rp_chain(sh.C, bound);
        }
        __syncthreads();
        const double bound = sh.bhi[0][0];
        // numberRoots() is a size_t difference in the reference: a negative difference counts as "many"
        const long long nr0 = (long long) sh.bloCh[0][0] - (long long) sh.bhiCh[0][0];
        const double eps = bound / (10.0 * (nr0 < 0 ? 1.8446744073709552e19 : (double) nr0));
        int nActive = 1, cur = 0;
        for (int level = 0; level < 2200 && nActive > 0; level++) {
            bool divide = false, emit = false;
            double lo = 0, hi = 0, center = 0;
            int loCh = 0, hiCh = 0, ch = 0;
            if (lane < nActive) {
                lo = sh.blo[cur][lane];
                hi = sh.bhi[cur][lane];
                loCh = sh.bloCh[cur][lane];
                hiCh = sh.bhiCh[cur][lane];
                const int nr = loCh - hiCh;
                center = (hi + lo) / 2.0;
                divide = true;
                if (nr == 1 && (hi - lo) < eps) divide = false;
                else if (nr == 0) divide = false;
                else if (center == hi || center == lo) divide = false;
                emit = !divide && nr != 0;
                if (divide) ch = rp_chain(sh.C, center);
            }
            const unsigned long long dm = __ballot(divide), em = __ballot(emit), below = (1ull << lane) - 1ull;
            const int child = 2 * __popcll(dm & below), slot = nroots + __popcll(em & below);
            if (divide && child + 1 < 64) {
                sh.blo[cur ^ 1][child] = lo;
                sh.bhi[cur ^ 1][child] = center;
                sh.bloCh[cur ^ 1][child] = loCh;
                sh.bhiCh[cur ^ 1][child] = ch;
                sh.blo[cur ^ 1][child + 1] = center;
                sh.bhi[cur ^ 1][child + 1] = hi;
                sh.bloCh[cur ^ 1][child + 1] = ch;
                sh.bhiCh[cur ^ 1][child + 1] = hiCh;
            }
            if (emit && slot < RP_MAXR) sh.roots[slot] = 0.5 * (lo + hi);
            nroots = min(RP_MAXR, nroots + __popcll(em));
            nActive = min(64, 2 * __popcll(dm));
            cur ^= 1;
            __syncthreads();
        }
    }
    // ---- per real root: five Newton steps (Sturm.cpp:286-300), (x, y) from z (main.cpp:228-230), polish, essential matrix,
    //      SVD and the four (R, t) decompositions (CentralRelativePoseSacProblem.cpp:100-140)
    if (lane < nroots) {
        double z = sh.roots[lane];
        for (int it = 0; it < 5; it++) {
            double mono[11], v = 0, d = 0;
            mono[10] = 1.0;
#pragma unroll
            for (int i = 2; i <= 11; i++) mono[11 - i] = mono[11 - i + 1] * z;
#pragma unroll
            for (int j = 0; j < 11; j++) v += sh.C[0][j] * mono[j];
#pragma unroll
            for (int j = 0; j < 11; j++) d += sh.C[1][j] * mono[j];
            z = z - (v / d);
        }
        const double den = rp_polyval(sh.p3, 7, z);
        double xyz[3] = {rp_polyval(sh.p1, 8, z) / den, rp_polyval(sh.p2, 8, z) / den, z};
        rp_polish(sh.A, xyz);
        double E[9], nrm = 0;
#pragma unroll
        for (int k = 0; k < 9; k++) {
            E[k] = ((xyz[0] * sh.EE[k][0] + xyz[1] * sh.EE[k][1]) + xyz[2] * sh.EE[k][2]) + sh.EE[k][3];
            nrm += E[k] * E[k];
        }
        nrm = sqrt(nrm);
#pragma unroll
        for (int k = 0; k < 9; k++) E[k] /= nrm;
        double U[9], sv[3], V[9], Vt[9], T[9], Ra[9], Rb[9];
        rp_svd3(E, U, sv, V);
#pragma unroll
        for (int r = 0; r < 3; r++)
#pragma unroll
            for (int c = 0; c < 3; c++) Vt[3 * r + c] = V[3 * c + r];
        const double W[9] = {0, -1, 0, 1, 0, 0, 0, 0, 1}, Wt[9] = {0, 1, 0, -1, 0, 0, 0, 0, 1};
        rp_mat3mul(U, W, T);
        rp_mat3mul(T, Vt, Ra);
        rp_mat3mul(U, Wt, T);
        rp_mat3mul(T, Vt, Rb);
        const double sa = rp_det3(Ra) < 0 ? -1.0 : 1.0, sb = rp_det3(Rb) < 0 ? -1.0 : 1.0;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            double *m = sh.cand[4 * lane + j];
#pragma unroll
            for (int k = 0; k < 9; k++) m[k] = (j & 1) ? sb * Rb[k] : sa * Ra[k];
#pragma unroll
            for (int k = 0; k < 3; k++) m[9 + k] = (j < 2 ? 1.0 : -1.0) * (sv[0] * U[3 * k + 2]);
        }
    }
    __syncthreads();
    // ---- disambiguation on the 8 sampled correspondences (:142-198): lowest summed reprojection error, first wins on ties
    {
        double quality = 1000000.0;
        int key = 1 << 20;
        if (lane < 4 * nroots) {
            const double *m = sh.cand[lane];
            double q = 0;
            for (int k = 0; k < 8; k++) q += rp_score(m, m + 9, sh.sb1[k], sh.sb2[k]);
            if (q < 1000000.0) { quality = q; key = lane; }
        }
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) {
            const double oq = __shfl_xor(quality, d);
            const int ok = __shfl_xor(key, d);
            if (oq < quality || (oq == quality && ok < key)) { quality = oq; key = ok; }
        }
        if (lane == 0) sh.have = key < (1 << 20);
        if (key < (1 << 20) && lane < 12) sh.best[lane] = sh.cand[key][lane];
    }
    __syncthreads();
    // ---- countWithinDistance over all correspondences (SampleConsensusProblem.hpp:185-200)
    int cnt = 0;
    if (sh.have) {
        double m[12];
#pragma unroll
        for (int k = 0; k < 12; k++) m[k] = sh.best[k];
        for (int i = lane; i < P.n; i += 64) {
            const double f1[3] = {P.bv1[3 * (size_t) i], P.bv1[3 * (size_t) i + 1], P.bv1[3 * (size_t) i + 2]};
            const double f2[3] = {P.bv2[3 * (size_t) i], P.bv2[3 * (size_t) i + 1], P.bv2[3 * (size_t) i + 2]};
            cnt += rp_score(m, m + 9, f1, f2) < P.threshold ? 1 : 0;
        }
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) cnt += __shfl_xor(cnt, d);
    }
    if (lane == 0) P.counts[h] = sh.have ? cnt : -1;
    if (lane < 12) P.models[12 * (size_t) h + lane] = sh.have ? sh.best[lane] : 0.0;
}


// ---------------------------------------------------------------------------------------------------------------------
// k_relpose_select: replay of Ransac<>::computeModel's loop (Ransac.hpp:66-123) over the hypotheses in draw order, then
// selectWithinDistance for the winner (:139) with an ordered list of the inlier indices.
struct RelposeSelectOut {
    double model[12];
    int ok, have_model, best, iterations, n_inliers, consumed, need_more;
};
struct RelposeSelectArgs {
    const double *bv1, *bv2, *models;
    const int *counts;
    int n, H, max_iters;
    double threshold;
    RelposeSelectOut *out, *out_host;  // device copy for the refinement kernel, pinned host copy for the caller
    uint8_t *mask;                     // [n] pinned host
    int *inl;                          // [n] ordered inlier indices (device)
};
constexpr int RP_NT = 256;     // k_relpose_select
constexpr int RP_LM_NT = 256;  // k_relpose_lm (512 threads measured slower: every barrier and the replicated 6 x 6 algebra cost more than the residual passes gain)

__global__ __launch_bounds__(RP_NT) void k_relpose_select(const RelposeSelectArgs P) {
    __shared__ RelposeSelectOut so;
    __shared__ int s_cnt[RP_NT];
    const int tid = threadIdx.x;
    if (tid == 0) {
        int iterations = 0, best = -2147483647, bestIdx = -1, j = 0, needMore = 0;
        unsigned skipped = 0;
        const unsigned maxSkip = (unsigned) P.max_iters * 10u;
        double k = 1.0;
        while ((double) iterations < k && skipped < maxSkip) {
            if (j >= P.H) { needMore = 1; break; }
            const int c = P.counts[j];
            if (c < 0) {  // computeModelCoefficients failed: not an iteration (:79-85)
                ++skipped;
                ++j;
                continue;
            }
            if (c > best) {
                best = c;
                bestIdx = j;
                const double w = (double) best / (double) P.n;
                double pNo = 1.0 - pow(w, 8.0);
                pNo = fmax(RP_EPS, pNo);
                pNo = fmin(1.0 - RP_EPS, pNo);
                k = log(1.0 - 0.99) / log(pNo);
            }
            ++iterations;
            ++j;
            if (iterations > P.max_iters) break;
        }
        so.have_model = bestIdx >= 0;
        so.best = bestIdx;
        so.iterations = iterations;
        so.consumed = j;
        so.need_more = needMore;
        for (int q = 0; q < 12; q++) so.model[q] = bestIdx >= 0 ? P.models[12 * (size_t) bestIdx + q] : 0.0;
    }
    __syncthreads();
    // inlier mask + ordered index list: each thread owns a contiguous chunk
    const int chunk = (P.n + RP_NT - 1) / RP_NT, i0 = tid * chunk, i1 = min(P.n, i0 + chunk);
    int cnt = 0;
    unsigned long long mine = 0;  // this thread's inlier flags (chunk <= 64); the pinned host mask is write-only for the device
    if (so.have_model && !so.need_more) {
        double m[12];
#pragma unroll
        for (int q = 0; q < 12; q++) m[q] = so.model[q];
        for (int i = i0; i < i1; i++) {
            const double f1[3] = {P.bv1[3 * (size_t) i], P.bv1[3 * (size_t) i + 1], P.bv1[3 * (size_t) i + 2]};
            const double f2[3] = {P.bv2[3 * (size_t) i], P.bv2[3 * (size_t) i + 1], P.bv2[3 * (size_t) i + 2]};
            const bool in = rp_score(m, m + 9, f1, f2) < P.threshold;
            P.mask[i] = in ? 1 : 0;
            if (in && i - i0 < 64) mine |= 1ull << (i - i0);
            cnt += in ? 1 : 0;
        }
    } else
        for (int i = i0; i < i1; i++) P.mask[i] = 0;
    s_cnt[tid] = cnt;
    __syncthreads();
    int base = 0;
    for (int t = 0; t < tid; t++) base += s_cnt[t];
    if (so.have_model && !so.need_more)
        for (int i = i0; i < i1; i++)
            if (i - i0 < 64 ? (mine >> (i - i0)) & 1ull : P.mask[i] != 0) P.inl[base++] = i;
    if (tid == RP_NT - 1) {
        so.n_inliers = base;  // the last thread's running total
        so.ok = so.have_model && !so.need_more && base >= 10;  // multi_view_geometry.cpp:281-285
        *P.out = so;
        *P.out_host = so;
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// k_relpose_lm: optimize_nonlinear (methods.cpp:1152-1180) = Eigen::LevenbergMarquardt (LevenbergMarquardt.h:168-356) over
// x = (t, Cayley(R)) with residual_i = score_i (methods.cpp:1085-1150) and a forward-difference Jacobian; one workgroup.
struct RelposeLmOut {
    double model[12];
    int iterations, status, nfev, ran;
    long long cyc[6];  // wall_clock64 ticks (100 MHz): Jacobian passes | QR | trust-region algebra | trial passes | total | lmpar calls
};
struct RelposeLmArgs {
    const double *bv1, *bv2;
    const int *inl;
    const RelposeSelectOut *sel;
    double *fvec, *wa4, *fjac;  // [n], [n], [6][n]
    int n;
    RelposeLmOut *out;  // pinned host
};
__device__ __forceinline__ void rp_cayley2rot(const double *c, double *R) {  // cayley.cpp:34-53
    const double scale = 1 + c[0] * c[0] + c[1] * c[1] + c[2] * c[2];
    R[0] = 1 + c[0] * c[0] - c[1] * c[1] - c[2] * c[2];
    R[1] = 2 * (c[0] * c[1] - c[2]);
    R[2] = 2 * (c[0] * c[2] + c[1]);
    R[3] = 2 * (c[0] * c[1] + c[2]);
    R[4] = 1 - c[0] * c[0] + c[1] * c[1] - c[2] * c[2];
    R[5] = 2 * (c[1] * c[2] - c[0]);
    R[6] = 2 * (c[0] * c[2] - c[1]);
    R[7] = 2 * (c[1] * c[2] + c[0]);
    R[8] = 1 - c[0] * c[0] - c[1] * c[1] + c[2] * c[2];
    const double f = 1 / scale;
#pragma unroll
    for (int k = 0; k < 9; k++) R[k] = f * R[k];
}
__device__ void rp_rot2cayley(const double *R, double *c) {  // cayley.cpp:74-88: C = (R - I)(R + I)^-1
    double C1[9], C2[9], inv[9], Cm[9];
#pragma unroll
    for (int k = 0; k < 9; k++) {
        C1[k] = R[k] - (k % 4 == 0 ? 1.0 : 0.0);
        C2[k] = R[k] + (k % 4 == 0 ? 1.0 : 0.0);
    }
    const double id = 1.0 / rp_det3(C2);
    inv[0] = (C2[4] * C2[8] - C2[5] * C2[7]) * id;
    inv[1] = (C2[2] * C2[7] - C2[1] * C2[8]) * id;
    inv[2] = (C2[1] * C2[5] - C2[2] * C2[4]) * id;
    inv[3] = (C2[5] * C2[6] - C2[3] * C2[8]) * id;
    inv[4] = (C2[0] * C2[8] - C2[2] * C2[6]) * id;
    inv[5] = (C2[2] * C2[3] - C2[0] * C2[5]) * id;
    inv[6] = (C2[3] * C2[7] - C2[4] * C2[6]) * id;
    inv[7] = (C2[1] * C2[6] - C2[0] * C2[7]) * id;
    inv[8] = (C2[0] * C2[4] - C2[1] * C2[3]) * id;
    rp_mat3mul(C1, inv, Cm);
    c[0] = -Cm[5];
    c[1] = Cm[2];
    c[2] = -Cm[1];
}
// sum of K per-thread values over the workgroup, delivered to every thread in a fixed order; ONE barrier per call (the
// staging buffer alternates, so the next call's writes cannot overtake this call's reads)
template <int K> __device__ void rp_block_sum(double (&v)[K], double (*red)[RP_LM_NT / 64][8], int &parity) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < K; k++) {
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) v[k] += __shfl_xor(v[k], d);
        if (lane == 0) red[parity][wave][k] = v[k];
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < K; k++) {
        double t = red[parity][0][k];
#pragma unroll
        for (int w = 1; w < RP_LM_NT / 64; w++) t += red[parity][w][k];
        v[k] = t;
    }
    parity ^= 1;
}

// LDS = true: residuals, their working copy and the Jacobian (8 doubles per inlier) live in dynamic LDS (n <= RP_LDS_ROWS);
// otherwise in the context's scratch.  Row i of every array belongs to thread i % RP_LM_NT.
constexpr int RP_LDS_ROWS = 2368;  // 8 * 8 B * 2368 = 148 KB of the CU's 160 KB
template <bool LDS> __global__ __launch_bounds__(RP_LM_NT) void k_relpose_lm(const RelposeLmArgs P) {
    extern __shared__ double rp_dyn[];
    __shared__ double red[2][RP_LM_NT / 64][8];
    constexpr int N = 6;
    const int tid = threadIdx.x;
    const RelposeSelectOut *sel = P.sel;
    if (!sel->ok) {
        if (tid == 0) P.out->ran = 0;
        return;
    }
    const int m = sel->n_inliers, ld = P.n;
    double *fvec, *wa4, *fjac;
    if constexpr (LDS) {
        fvec = rp_dyn;
        wa4 = rp_dyn + ld;
        fjac = rp_dyn + 2 * (size_t) ld;
    } else {
        fvec = P.fvec;
        wa4 = P.wa4;
        fjac = P.fjac;
    }
    const double ftol = 10 * RP_EPS, xtol = 10 * RP_EPS;
    const int maxfev = 1000;
    int parity = 0;
    double x[N];
    {
        double mdl[12];
#pragma unroll
        for (int q = 0; q < 12; q++) mdl[q] = sel->model[q];
        x[0] = mdl[9];
        x[1] = mdl[10];
        x[2] = mdl[11];
        rp_rot2cayley(mdl, x + 3);
    }
    auto residual = [&](const double *R, const double *t, int i) -> double {
        const int id = P.inl[i];
        const double f1[3] = {P.bv1[3 * (size_t) id], P.bv1[3 * (size_t) id + 1], P.bv1[3 * (size_t) id + 2]};
        const double f2[3] = {P.bv2[3 * (size_t) id], P.bv2[3 * (size_t) id + 1], P.bv2[3 * (size_t) id + 2]};
        return rp_score(R, t, f1, f2);
    };
    RpLm S{};
    S.nfev = 1;
    S.iter = 1;
    {
        double R[9], sq[1] = {0};
        rp_cayley2rot(x + 3, R);
        for (int i = tid; i < m; i += RP_LM_NT) {
            const double r = residual(R, x, i);
            fvec[i] = r;
            sq[0] += r * r;
        }
        rp_block_sum<1>(sq, red, parity);
        S.fnorm = sqrt(sq[0]);
    }
    double diag[N] = {0, 0, 0, 0, 0, 0};
    int status = 0;
    const double fdeps = sqrt(RP_EPS);
    long long cyc[6] = {0, 0, 0, 0, 0, 0};
    const long long tAll = wall_clock64(), cAll = clock64();
    while (!status) {
        long long tc = wall_clock64();
        // ---- forward-difference Jacobian (NumericalDiff.h:63-121): f(x) again (bit-identical to fvec) + one evaluation per unknown
        double colsq[N] = {0, 0, 0, 0, 0, 0};
#pragma unroll
        for (int j = 0; j < N; j++) {
            double h = fdeps * fabs(x[j]);
            if (h == 0) h = fdeps;
            double xt[N], R[9];
#pragma unroll
            for (int q = 0; q < N; q++) xt[q] = x[q];
            xt[j] += h;
            rp_cayley2rot(xt + 3, R);
            for (int i = tid; i < m; i += RP_LM_NT) {
                const double d = (residual(R, xt, i) - fvec[i]) / h;
                fjac[(size_t) j * ld + i] = d;
                colsq[j] += d * d;
            }
        }
        for (int i = tid; i < m; i += RP_LM_NT) wa4[i] = fvec[i];
        S.nfev += N + 1;
        rp_block_sum<N>(colsq, red, parity);
        double wa2[N];
#pragma unroll
        for (int j = 0; j < N; j++) wa2[j] = sqrt(colsq[j]);
        // ---- ColPivHouseholderQR (ColPivHouseholderQR.h:478-581) of the m x 6 Jacobian with Q^T applied to the residual copy in
        //      the same passes.  perm[j] = physical column at pivoted position j; R and Q^T f are kept in registers by every
        //      thread; the reflectors themselves are not stored (nothing downstream needs Q).  Per step: one pass that
        //      accumulates the tail's sum of squares and its dot products with the other columns, ONE workgroup reduction,
        //      one update pass.
        int perm[N] = {0, 1, 2, 3, 4, 5};
        double nu[N], nd[N], Rm[N][N] = {}, qtf[N];
        double maxn = 0;
        cyc[0] += wall_clock64() - tc;
        tc = wall_clock64();
#pragma unroll
        for (int j = 0; j < N; j++) {
            nu[j] = nd[j] = wa2[j];
            maxn = fmax(maxn, nu[j]);
        }
        const double th = (maxn * RP_EPS) * (maxn * RP_EPS) / (double) m, downdate = sqrt(RP_EPS);
        int nonzero = N;
        double maxpivot = 0;
#pragma unroll
        for (int k = 0; k < N; k++) {
            __syncthreads();  // the previous pass's writes (row k of every column has a single owner)
            int big = k;
#pragma unroll
            for (int j = k + 1; j < N; j++)
                if (nu[j] > nu[big]) big = j;
            if (nonzero == N && nu[big] * nu[big] < th * (double) (m - k)) nonzero = k;
#pragma unroll
            for (int j = k + 1; j < N; j++)
                if (big == j) {
#pragma unroll
                    for (int i = 0; i < N; i++) {
                        const double t = Rm[k][i];
                        Rm[k][i] = Rm[j][i];
                        Rm[j][i] = t;
                    }
                    double t = nu[k]; nu[k] = nu[j]; nu[j] = t;
                    t = nd[k]; nd[k] = nd[j]; nd[j] = t;
                    const int ti = perm[k]; perm[k] = perm[j]; perm[j] = ti;
                }
            double *colk = fjac + (size_t) perm[k] * ld;
            double acc[N + 1] = {0, 0, 0, 0, 0, 0, 0};  // [0] tail sum of squares, [j] . column j (j > k), [N] . residual copy
            for (int i = tid; i < m; i += RP_LM_NT)
                if (i > k) {
                    const double e = colk[i];
                    acc[0] += e * e;
#pragma unroll
                    for (int j = k + 1; j < N; j++) acc[j] += e * fjac[(size_t) perm[j] * ld + i];
                    acc[N] += e * wa4[i];
                }
            const double c0 = colk[k];
            double rowk[N + 1];  // row k of the other columns / of the residual copy before this reflection
#pragma unroll
            for (int j = k + 1; j < N; j++) rowk[j] = fjac[(size_t) perm[j] * ld + k];
            rowk[N] = wa4[k];
            rp_block_sum<N + 1>(acc, red, parity);
            double tau, beta, denom = 1.0;
            if (acc[0] <= RP_DBL_MIN) {
                tau = 0;
                beta = c0;
            } else {
                beta = sqrt(c0 * c0 + acc[0]);
                if (c0 >= 0) beta = -beta;
                denom = c0 - beta;
                tau = (beta - c0) / beta;
            }
            Rm[k][k] = beta;
            if (fabs(beta) > maxpivot) maxpivot = fabs(beta);
            double tmp[N + 1];  // essential^T column + its row-k entry
#pragma unroll
            for (int j = k + 1; j <= N; j++) tmp[j] = acc[j] / denom + rowk[j];
#pragma unroll
            for (int j = k + 1; j < N; j++) Rm[j][k] = tau != 0 ? rowk[j] - tau * tmp[j] : rowk[j];
            qtf[k] = tau != 0 ? rowk[N] - tau * tmp[N] : rowk[N];
            if (tau != 0)
                for (int i = tid; i < m; i += RP_LM_NT)
                    if (i > k) {
                        const double e = tau * (colk[i] / denom);
#pragma unroll
                        for (int j = k + 1; j < N; j++) fjac[(size_t) perm[j] * ld + i] -= e * tmp[j];
                        wa4[i] -= e * tmp[N];
                    }
            // norm downdate (the LAPACK xGEQP3 rule); a direct recomputation costs one more reduction
            bool redo[N] = {false, false, false, false, false, false};
            bool any = false;
#pragma unroll
            for (int j = k + 1; j < N; j++) {
                if (nu[j] != 0) {
                    double temp = fabs(Rm[j][k]) / nu[j];
                    temp = (1 + temp) * (1 - temp);
                    temp = temp < 0 ? 0 : temp;
                    const double q = nu[j] / nd[j];
                    if (temp * (q * q) <= downdate) {
                        redo[j] = true;
                        any = true;
                    } else
                        nu[j] *= sqrt(temp);
                }
            }
            if (any) {
                double sq[N] = {0, 0, 0, 0, 0, 0};
                for (int i = tid; i < m; i += RP_LM_NT)  // own rows only: written by this thread just above
                    if (i > k) {
#pragma unroll
                        for (int j = k + 1; j < N; j++)
                            if (redo[j]) {
                                const double v = fjac[(size_t) perm[j] * ld + i];
                                sq[j] += v * v;
                            }
                    }
                rp_block_sum<N>(sq, red, parity);
#pragma unroll
                for (int j = k + 1; j < N; j++)
                    if (redo[j]) nd[j] = nu[j] = sqrt(sq[j]);
            }
        }
        int rank = 0;
        cyc[1] += wall_clock64() - tc;
        {
            const double pm = maxpivot * (RP_EPS * (double) N);
#pragma unroll
            for (int i = 0; i < N; i++) rank += (i < nonzero && fabs(Rm[i][i]) > pm) ? 1 : 0;
        }
        if (S.iter == 1) {
            double t[N];
#pragma unroll
            for (int j = 0; j < N; j++) {
                diag[j] = wa2[j] == 0 ? 1 : wa2[j];
                t[j] = diag[j] * x[j];
            }
            S.xnorm = rp_norm<N>(t);
            S.delta = 100.0 * S.xnorm;
            if (S.delta == 0) S.delta = 100.0;
        }
        double gnorm = 0;
        if (S.fnorm != 0) {
#pragma unroll
            for (int j = 0; j < N; j++) {
                const double w = rp_pick<N>(wa2, perm[j]);
                if (w != 0) {
                    double sum = 0;
#pragma unroll
                    for (int i = 0; i <= j; i++) sum += Rm[j][i] * (qtf[i] / S.fnorm);
                    gnorm = fmax(gnorm, fabs(sum / w));
                }
            }
        }
        if (gnorm <= 0.0) {
            status = 4;
            break;
        }
#pragma unroll
        for (int j = 0; j < N; j++) diag[j] = fmax(diag[j], wa2[j]);
        double ratio;
        do {
            double wa1[N], xn[N], t[N], R[9];
            tc = wall_clock64();
            rp_lmpar<N>(Rm, rank, perm, diag, qtf, S.delta, S.par, wa1);
            cyc[2] += wall_clock64() - tc;
            cyc[5]++;
            tc = wall_clock64();
#pragma unroll
            for (int j = 0; j < N; j++) {
                wa1[j] = -wa1[j];
                xn[j] = x[j] + wa1[j];
                t[j] = diag[j] * wa1[j];
            }
            const double pnorm = rp_norm<N>(t);
            if (S.iter == 1) S.delta = fmin(S.delta, pnorm);
            double sq[1] = {0};
            rp_cayley2rot(xn + 3, R);
            for (int i = tid; i < m; i += RP_LM_NT) {  // wa4: each thread touches its own rows only from here on
                const double r = residual(R, xn, i);
                wa4[i] = r;
                sq[0] += r * r;
            }
            S.nfev++;
            rp_block_sum<1>(sq, red, parity);
            cyc[3] += wall_clock64() - tc;
            const double fnorm1 = sqrt(sq[0]);
            double actred, prered;
            rp_lm_ratio<N>(S, Rm, perm, wa1, pnorm, fnorm1, actred, prered, ratio);
            if (ratio >= 1e-4) {
#pragma unroll
                for (int j = 0; j < N; j++) {
                    x[j] = xn[j];
                    t[j] = diag[j] * x[j];
                }
                for (int i = tid; i < m; i += RP_LM_NT) fvec[i] = wa4[i];
                S.xnorm = rp_norm<N>(t);
                S.fnorm = fnorm1;
                S.iter++;
            }
            status = rp_lm_tests(S, actred, prered, ratio, gnorm, ftol, xtol, maxfev);
        } while (!status && ratio < 1e-4);
    }
    if (tid == 0) {
        double R[9];
        rp_cayley2rot(x + 3, R);
        for (int q = 0; q < 9; q++) P.out->model[q] = R[q];
        for (int q = 0; q < 3; q++) P.out->model[9 + q] = x[q];
        P.out->iterations = S.iter;
        P.out->status = status;
        P.out->nfev = S.nfev;
        P.out->ran = 1;
        cyc[4] = wall_clock64() - tAll;
        for (int q = 0; q < 6; q++) P.out->cyc[q] = cyc[q];
        P.out->cyc[5] = clock64() - cAll;  // shader clock ticks over the same span
    }
}

// SampleConsensusProblem<M>: rng_dist_ = uniform_int_distribution<>(0, INT_MAX) over std::mt19937 seeded 12345u (or time +
// clock), shuffled_indices_ persists across draws (SampleConsensusProblem.hpp:36-84); sample size 5 + 3
// (CentralRelativePoseSacProblem.cpp:304-312).
struct Sampler8 {
    std::mt19937 alg;
    std::uniform_int_distribution<> dist{0, std::numeric_limits<int>::max()};
    std::vector<int> shuffled;
    Sampler8(int n, bool random_seed, uint32_t seed) : shuffled((size_t) n) {
        if (random_seed) alg.seed(static_cast<unsigned>(time(0)) + static_cast<unsigned>(clock()));
        else alg.seed(seed);
        for (int i = 0; i < n; i++) shuffled[(size_t) i] = i;
    }
    void draw(int *out8) {
        const size_t index_size = shuffled.size();
        for (unsigned i = 0; i < 8; ++i) std::swap(shuffled[i], shuffled[i + ((size_t) dist(alg) % (index_size - i))]);
        for (int i = 0; i < 8; i++) out8[i] = shuffled[(size_t) i];
    }
};

double relpose_threshold(float error_threshold, float fx, float fy) {  // multi_view_geometry.cpp:272-276
    float focal = fx + fy;
    focal /= 2.f;
    return 2.0 * (1.0 - std::cos(std::atan((double) (error_threshold / focal))));
}

}  // namespace

extern "C" int alva_relpose_draw_samples(int n_points, int count, int do_random, uint32_t seed, int *h_samples8) {
    ALVA_ARG(n_points >= 8 && count >= 0 && h_samples8);
    Sampler8 s(n_points, do_random != 0, seed);
    for (int k = 0; k < count; k++) s.draw(h_samples8 + 8 * k);
    return ALVA_OK;
}

extern "C" int alva_relpose_hypotheses(alva_ctx *ctx, const double *d_bv1, const double *d_bv2, int n, const int *h_samples8, int n_samples,
                                       float error_threshold, float fx, float fy, double *h_models12, int *h_counts) {
    ALVA_ARG(ctx && d_bv1 && d_bv2 && n >= 8 && h_samples8 && n_samples > 0 && h_models12 && h_counts);
    const size_t off_m = ((size_t) n_samples * 32 + 255) / 256 * 256, off_c = off_m + (size_t) n_samples * 96;
    uint8_t *pin = nullptr;
    int rc = alva_ctx_pinned(ctx, off_c + (size_t) n_samples * 4, (void **) &pin);
    if (rc) return rc;
    memcpy(pin, h_samples8, (size_t) n_samples * 32);
    RelposeArgs A{d_bv1, d_bv2, (const int *) pin, n, n_samples, relpose_threshold(error_threshold, fx, fy), (double *) (pin + off_m),
                  (int *) (pin + off_c)};
    hipLaunchKernelGGL(k_relpose_hyp, dim3(n_samples), dim3(64), 0, ctx->stream, A);
    ALVA_LAUNCH_CHECK();
    ALVA_HIP(alva_stream_sync(ctx->stream));
    memcpy(h_models12, pin + off_m, (size_t) n_samples * 96);
    memcpy(h_counts, pin + off_c, (size_t) n_samples * 4);
    return ALVA_OK;
}

extern "C" int alva_compute_5pt_essential(alva_ctx *ctx, const double *d_bv1, const double *d_bv2, int n, int max_iters,
                                          float error_threshold, int optimize, int do_random, uint32_t seed, float fx, float fy,
                                          double *h_R, double *h_t, uint8_t *h_inlier, alva_relpose_info *h_info, int *h_ok) {
    ALVA_ARG(ctx && h_R && h_t && h_ok && max_iters > 0);
    *h_ok = 0;
    if (h_info) memset(h_info, 0, sizeof(*h_info));
    if (n < 8) return ALVA_OK;  // multi_view_geometry.cpp:242-245
    ALVA_ARG(d_bv1 && d_bv2);
    const double threshold = relpose_threshold(error_threshold, fx, fy);
    // the loop runs at most max_iters + 1 iterations (:124) plus up to 10 x max_iters skipped samples (:64)
    const int max_draws = max_iters + 1 + 10 * max_iters;
    int H = std::min(max_draws, max_iters + 1 + 11);
    RelposeSelectOut sel{};
    RelposeLmOut lm{};
    const uint8_t *mask = nullptr;
    for (;;) {
        // pinned host: samples | select out | refinement out | inlier mask; device scratch: models | counts | select out | inlier list |
        // residuals, residual copy, Jacobian
        const size_t off_sel = ((size_t) H * 32 + 255) / 256 * 256, off_lm = off_sel + 256, off_mask = off_lm + 256;
        uint8_t *pin = nullptr;
        int rc = alva_ctx_pinned(ctx, off_mask + (size_t) n, (void **) &pin);
        if (rc) return rc;
        const size_t s_counts = (size_t) H * 96, s_sel = (s_counts + (size_t) H * 4 + 255) / 256 * 256, s_inl = s_sel + 256,
                     s_f = (s_inl + (size_t) n * 4 + 255) / 256 * 256, total = s_f + (size_t) n * 8 * 8;
        uint8_t *scr = nullptr;
        rc = alva_ctx_scratch(ctx, 8, total, (void **) &scr);
        if (rc) return rc;
        {
            Sampler8 smp(n, do_random != 0, seed);
            for (int k = 0; k < H; k++) smp.draw((int *) pin + 8 * k);
        }
        RelposeArgs A{d_bv1, d_bv2, (const int *) pin, n, H, threshold, (double *) scr, (int *) (scr + s_counts)};
        hipLaunchKernelGGL(k_relpose_hyp, dim3(H), dim3(64), 0, ctx->stream, A);
        ALVA_LAUNCH_CHECK();
        RelposeSelectArgs B{d_bv1, d_bv2, A.models, A.counts, n, H, max_iters, threshold, (RelposeSelectOut *) (scr + s_sel),
                            (RelposeSelectOut *) (pin + off_sel), pin + off_mask, (int *) (scr + s_inl)};
        hipLaunchKernelGGL(k_relpose_select, dim3(1), dim3(RP_NT), 0, ctx->stream, B);
        ALVA_LAUNCH_CHECK();
        if (optimize) {
            double *f = (double *) (scr + s_f);
            RelposeLmArgs C{d_bv1, d_bv2, B.inl, B.out, f, f + n, f + 2 * (size_t) n, n, (RelposeLmOut *) (pin + off_lm)};
            if (n <= RP_LDS_ROWS) {
                const size_t lds = (size_t) n * 8 * sizeof(double);
                static bool attr_set[64] = {};  // dynamic LDS beyond 64 KB has to be allowed once per device
                if (ctx->device >= 64 || !attr_set[ctx->device]) {
                    ALVA_HIP(hipFuncSetAttribute((const void *) k_relpose_lm<true>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                                 RP_LDS_ROWS * 8 * (int) sizeof(double)));
                    if (ctx->device < 64) attr_set[ctx->device] = true;
                }
                hipLaunchKernelGGL(k_relpose_lm<true>, dim3(1), dim3(RP_LM_NT), lds, ctx->stream, C);
            } else
                hipLaunchKernelGGL(k_relpose_lm<false>, dim3(1), dim3(RP_LM_NT), 0, ctx->stream, C);
            ALVA_LAUNCH_CHECK();
        }
        ALVA_HIP(alva_stream_sync(ctx->stream));
        memcpy(&sel, pin + off_sel, sizeof(sel));
        memcpy(&lm, pin + off_lm, sizeof(lm));
        mask = pin + off_mask;
        if (!sel.need_more || H >= max_draws) break;
        H = std::min(max_draws, H * 2);  // rare: many samples without a real root; redo with a longer prefix of the same stream
    }
    if (optimize && sel.ok && getenv("ALVA_RP_PROFILE"))
        fprintf(stderr, "[k_relpose_lm] us: jacobian %.1f | qr %.1f | lmpar %.1f (%lld calls) | trial passes %.1f | total %.1f | %d iterations\n",
                lm.cyc[0] * 0.01, lm.cyc[1] * 0.01, lm.cyc[2] * 0.01, (long long) lm.nfev, lm.cyc[3] * 0.01, lm.cyc[4] * 0.01, lm.iterations);
    if (optimize && sel.ok && getenv("ALVA_RP_PROFILE"))
        fprintf(stderr, "[k_relpose_lm] shader clock over the kernel: %.0f MHz\n", (double) lm.cyc[5] / (lm.cyc[4] * 0.01));
    if (h_info) {
        h_info->iterations = sel.iterations;
        h_info->n_inliers = sel.n_inliers;
        h_info->draws = sel.consumed;
        memcpy(h_info->ransac_model, sel.model, sizeof(sel.model));
        if (optimize && sel.ok) {
            h_info->lm_iterations = lm.iterations;
            h_info->lm_status = lm.status;
            h_info->lm_nfev = lm.nfev;
        }
    }
    if (h_inlier) memcpy(h_inlier, mask, (size_t) n);
    if (!sel.ok) return ALVA_OK;  // no model or fewer than 10 inliers (:281-285)
    const double *m = (optimize ? lm.model : sel.model);
    for (int k = 0; k < 9; k++) h_R[k] = m[k];
    for (int k = 0; k < 3; k++) h_t[k] = m[9 + k];
    *h_ok = 1;
    return ALVA_OK;
}
