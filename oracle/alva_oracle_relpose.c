/* TEST INFRASTRUCTURE ONLY -- CPU restatement of the reference's two-view map initialisation.
 * Never linked into, imported by or executed from the product path (alvaar_amd/); only tests/, smoke() and bench.py's
 * cpu_baseline leg may use it.
 *
 * Path restated (all under /root/reference/):
 *   MultiViewGeometry::compute5ptEssentialMatrix            src/slam/src/multi_view_geometry.cpp:225-320
 *   opengv::sac::Ransac<>::computeModel                      src/libs/opengv/include/opengv/sac/implementation/Ransac.hpp:45-143
 *   SampleConsensusProblem (sampling, count/select)          .../sac/implementation/SampleConsensusProblem.hpp:36-200
 *   CentralRelativePoseSacProblem (NISTER)                   src/libs/opengv/src/sac_problems/relative_pose/CentralRelativePoseSacProblem.cpp:38-334
 *   relative_pose::fivept_nister                             src/libs/opengv/src/relative_pose/methods.cpp:239-268
 *   modules::fivept_nister_main                              src/libs/opengv/src/relative_pose/modules/main.cpp:135-261
 *   fivept_nister::composeA / polynomial products / polish   src/libs/opengv/src/relative_pose/modules/fivept_nister/modules.cpp
 *   math::Sturm                                              src/libs/opengv/src/math/Sturm.cpp:150-492
 *   triangulation::triangulate2                              src/libs/opengv/src/triangulation/methods.cpp:67-90
 *   relative_pose::optimize_nonlinear                        src/libs/opengv/src/relative_pose/methods.cpp:1082-1180
 *   math::cayley2rot / rot2cayley                            src/libs/opengv/src/math/cayley.cpp:34-88
 *   Eigen::LevenbergMarquardt / lmpar2 / qrsolv              src/libs/eigen/unsupported/Eigen/src/NonLinearOptimization/{LevenbergMarquardt.h:168-356,lmpar.h:160-296,qrsolv.h:17-88}
 *   Eigen::NumericalDiff (Forward)                           src/libs/eigen/unsupported/Eigen/src/NumericalDiff/NumericalDiff.h:63-121
 *   Eigen::ColPivHouseholderQR / JacobiSVD preconditioner    src/libs/eigen/Eigen/src/QR/ColPivHouseholderQR.h:478-581, Eigen/src/SVD/JacobiSVD.h
 *
 * Parity bar for this stage is a float tolerance (BASELINE.json: pose RMSE <= 1e-5): the reference refines the RANSAC model
 * with a forward-difference Levenberg-Marquardt whose Jacobian carries ~1e-8 of rounding noise, so the last digits of its
 * own result depend on summation order inside Eigen.  What IS discrete is restated exactly: the sample stream, the null-space
 * basis the 10th-degree polynomial is written in (pivot order of Eigen's column-pivoting QR, which for unit bearings is
 * decided by the rounding of five column norms -- reproduced with Eigen's packet summation order), the Sturm bracketing
 * with its FIFO bisection and five Newton steps, the hypothesis order, the adaptive iteration count and the inlier test. */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <float.h>
#include "alva_oracle.h"

#define DBL_EPS 2.220446049250313e-16

/* ------------------------------------------------------------------------------------------------------------------
 * Eigen's vectorised (SSE2, 2 doubles / packet) sum of squares of a contiguous column: redux with two packet
 * accumulators, then the scalar tail (Eigen/src/Core/Redux.h, LinearVectorizedTraversal, start 0). */
static double eig_sqnorm(const double *v, int n) {
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

/* Column-pivoting Householder QR of a column-major rows x cols matrix, in place (ColPivHouseholderQR.h:478-581).
 * perm[j] = original column now at position j.  Returns nothing; hcoef[size], maxpivot and nonzero pivots for rank(). */
typedef struct { int rows, cols, size, nonzero; double maxpivot; } cpqr_info;
static void householder_make(double *x, int n, double *tau, double *beta) { /* Householder.h:65-92: x = [c0; tail] -> [.; essential] */
    double tailSq = n == 1 ? 0.0 : eig_sqnorm(x + 1, n - 1);
    const double c0 = x[0];
    if (tailSq <= DBL_MIN) {
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
/* apply H = I - tau [1;ess][1;ess]^T to the column-major block M (r x c, leading dimension ld) from the left */
static void householder_apply_left(double *M, int ld, int r, int c, const double *ess, double tau) {
    if (r == 1) {
        for (int j = 0; j < c; j++) M[(size_t) j * ld] *= 1 - tau;
        return;
    }
    if (tau == 0) return;
    for (int j = 0; j < c; j++) {
        double *col = M + (size_t) j * ld;
        double tmp = 0;
        for (int i = 1; i < r; i++) tmp += ess[i - 1] * col[i];
        tmp += col[0];
        col[0] -= tau * tmp;
        for (int i = 1; i < r; i++) col[i] -= tau * ess[i - 1] * tmp;
    }
}
static void cpqr(double *A, int rows, int cols, double *hcoef, int *perm, cpqr_info *info) {
    const int size = rows < cols ? rows : cols;
    double *nu = (double *) malloc(sizeof(double) * 2 * (size_t) cols), *nd = nu + cols;
    int *transp = (int *) malloc(sizeof(int) * (size_t) cols);
    double maxn = 0;
    for (int k = 0; k < cols; k++) {
        nd[k] = nu[k] = sqrt(eig_sqnorm(A + (size_t) k * rows, rows));
        if (k == 0 || nu[k] > maxn) maxn = nu[k];
    }
    const double th = (maxn * DBL_EPS) * (maxn * DBL_EPS) / (double) rows;
    const double downdate = sqrt(DBL_EPS);
    info->rows = rows; info->cols = cols; info->size = size; info->nonzero = size; info->maxpivot = 0;
    for (int k = 0; k < size; k++) {
        int big = k;
        for (int j = k + 1; j < cols; j++)
            if (nu[j] > nu[big]) big = j; /* first maximum */
        const double bigSq = nu[big] * nu[big];
        if (info->nonzero == size && bigSq < th * (double) (rows - k)) info->nonzero = k;
        transp[k] = big;
        if (k != big) {
            for (int i = 0; i < rows; i++) {
                double t = A[(size_t) k * rows + i];
                A[(size_t) k * rows + i] = A[(size_t) big * rows + i];
                A[(size_t) big * rows + i] = t;
            }
            double t = nu[k]; nu[k] = nu[big]; nu[big] = t;
            t = nd[k]; nd[k] = nd[big]; nd[big] = t;
        }
        double beta;
        householder_make(A + (size_t) k * rows + k, rows - k, &hcoef[k], &beta);
        A[(size_t) k * rows + k] = beta;
        if (fabs(beta) > info->maxpivot) info->maxpivot = fabs(beta);
        householder_apply_left(A + (size_t) (k + 1) * rows + k, rows, rows - k, cols - k - 1, A + (size_t) k * rows + k + 1, hcoef[k]);
        for (int j = k + 1; j < cols; j++) {
            if (nu[j] != 0) {
                double temp = fabs(A[(size_t) j * rows + k]) / nu[j];
                temp = (1 + temp) * (1 - temp);
                temp = temp < 0 ? 0 : temp;
                const double q = nu[j] / nd[j];
                const double temp2 = temp * (q * q);
                if (temp2 <= downdate) {
                    nd[j] = sqrt(rows - k - 1 > 0 ? eig_sqnorm(A + (size_t) j * rows + k + 1, rows - k - 1) : 0.0);
                    nu[j] = nd[j];
                } else
                    nu[j] *= sqrt(temp);
            }
        }
    }
    for (int j = 0; j < cols; j++) perm[j] = j;
    for (int k = 0; k < size; k++) { /* applyTranspositionOnTheRight(k, transp[k]) */
        int t = perm[k]; perm[k] = perm[transp[k]]; perm[transp[k]] = t;
    }
    free(nu);
    free(transp);
}
static int cpqr_rank(const double *A, const cpqr_info *info) { /* ColPivHouseholderQR::rank(): threshold eps * diagonalSize */
    const double pm = fabs(info->maxpivot) * (DBL_EPS * (double) info->size);
    int r = 0;
    for (int i = 0; i < info->nonzero; i++) r += fabs(A[(size_t) i * info->rows + i]) > pm;
    return r;
}

/* ------------------------------------------------------------------------------------------------------------------
 * Null space of the 5 x 9 epipolar constraint matrix as JacobiSVD<MatrixXd>(Q, ComputeFullV).matrixV().block(0,5,9,4)
 * delivers it (methods.cpp:262-263): for more columns than rows JacobiSVD scales the matrix by its largest |entry|, runs
 * ColPivHouseholderQR on the adjoint (9 x 5) and takes V = full Householder Q; the Jacobi sweeps and the final sort only
 * touch V's first five columns, so columns 5..8 are exactly the last four columns of that Q.  EE is 9 x 4 row-major. */
static void nister_nullspace(const double Q[5][9], double EE[9][4]) {
    double scale = 0;
    for (int i = 0; i < 5; i++)
        for (int j = 0; j < 9; j++)
            if (fabs(Q[i][j]) > scale) scale = fabs(Q[i][j]);
    if (scale == 0) scale = 1;
    double A[45], h[5];
    int perm[5];
    for (int i = 0; i < 5; i++)
        for (int j = 0; j < 9; j++) A[i * 9 + j] = Q[i][j] / scale; /* adjoint, column-major 9 x 5 */
    cpqr_info info;
    cpqr(A, 9, 5, h, perm, &info);
    /* householderQ().evalTo: dst = I; for k = 4..0: dst.bottomRightCorner(9-k, 9-k).applyHouseholderOnTheLeft(v_k, h_k) */
    double Qf[81];
    memset(Qf, 0, sizeof(Qf));
    for (int i = 0; i < 9; i++) Qf[i * 9 + i] = 1;
    for (int k = 4; k >= 0; k--) householder_apply_left(Qf + (size_t) k * 9 + k, 9, 9 - k, 9 - k, A + (size_t) k * 9 + k + 1, h[k]);
    for (int r = 0; r < 9; r++)
        for (int c = 0; c < 4; c++) EE[r][c] = Qf[(size_t) (5 + c) * 9 + r];
}

/* ------------------------------------------------------------------------------------------------------------------
 * The ten cubic constraints on E(x,y,z) = x E0 + y E1 + z E2 + E3 (modules.cpp:38-368 holds them as expanded expressions;
 * here they are derived by polynomial arithmetic): row 0 = det E, row 1 + 3c + r = (2 E E^T E - tr(E E^T) E)_{rc}.
 * Column (monomial) order of A, modules.cpp:484-503:
 *   x^3 y^3 x^2y xy^2 x^2z x^2 y^2z y^2 xyz xy xz^2 xz x yz^2 yz y z^3 z^2 z 1 */
static const int MONO[20][3] = {{3, 0, 0}, {0, 3, 0}, {2, 1, 0}, {1, 2, 0}, {2, 0, 1}, {2, 0, 0}, {0, 2, 1}, {0, 2, 0}, {1, 1, 1}, {1, 1, 0},
                                {1, 0, 2}, {1, 0, 1}, {1, 0, 0}, {0, 1, 2}, {0, 1, 1}, {0, 1, 0}, {0, 0, 3}, {0, 0, 2}, {0, 0, 1}, {0, 0, 0}};
typedef struct { double c[4][4][4]; } poly3; /* coefficient of x^a y^b z^c, total degree <= 3 */
static void poly_mul(const poly3 *p, const poly3 *q, poly3 *o) {
    poly3 r;
    memset(&r, 0, sizeof(r));
    for (int a = 0; a < 4; a++)
        for (int b = 0; a + b < 4; b++)
            for (int c = 0; a + b + c < 4; c++) {
                const double pv = p->c[a][b][c];
                if (pv == 0) continue;
                for (int d = 0; a + d < 4; d++)
                    for (int e = 0; a + b + d + e < 4; e++)
                        for (int f = 0; a + b + c + d + e + f < 4; f++) r.c[a + d][b + e][c + f] += pv * q->c[d][e][f];
            }
    *o = r;
}
static void poly_axpy(poly3 *o, double s, const poly3 *p) {
    for (int a = 0; a < 4; a++)
        for (int b = 0; b < 4; b++)
            for (int c = 0; c < 4; c++) o->c[a][b][c] += s * p->c[a][b][c];
}
void orc_nister_compose_a(const double *EEflat /* [9][4] */, double *Aflat /* [10][20] */) {
    poly3 e[9];
    memset(e, 0, sizeof(e));
    for (int k = 0; k < 9; k++) {
        e[k].c[1][0][0] = EEflat[4 * k];
        e[k].c[0][1][0] = EEflat[4 * k + 1];
        e[k].c[0][0][1] = EEflat[4 * k + 2];
        e[k].c[0][0][0] = EEflat[4 * k + 3];
    }
    poly3 row[10], t, u;
    memset(row, 0, sizeof(row));
    /* det E by cofactors of the first row */
    const int cof[3][4] = {{4, 8, 5, 7}, {5, 6, 3, 8}, {3, 7, 4, 6}};
    for (int j = 0; j < 3; j++) {
        poly3 m;
        memset(&m, 0, sizeof(m));
        poly_mul(&e[cof[j][0]], &e[cof[j][1]], &t);
        poly_axpy(&m, 1.0, &t);
        poly_mul(&e[cof[j][2]], &e[cof[j][3]], &t);
        poly_axpy(&m, -1.0, &t);
        poly_mul(&e[j], &m, &t);
        poly_axpy(&row[0], 1.0, &t);
    }
    /* G = E E^T (quadratic), trace, then 2 G E - tr E */
    poly3 G[9], tr;
    memset(G, 0, sizeof(G));
    memset(&tr, 0, sizeof(tr));
    for (int r = 0; r < 3; r++)
        for (int c = 0; c < 3; c++)
            for (int k = 0; k < 3; k++) {
                poly_mul(&e[3 * r + k], &e[3 * c + k], &t);
                poly_axpy(&G[3 * r + c], 1.0, &t);
            }
    for (int r = 0; r < 3; r++) poly_axpy(&tr, 1.0, &G[4 * r]);
    for (int r = 0; r < 3; r++)
        for (int c = 0; c < 3; c++) {
            poly3 *o = &row[1 + 3 * c + r];
            for (int k = 0; k < 3; k++) {
                poly_mul(&G[3 * r + k], &e[3 * k + c], &t);
                poly_axpy(o, 2.0, &t);
            }
            poly_mul(&tr, &e[3 * r + c], &u);
            poly_axpy(o, -1.0, &u);
        }
    for (int i = 0; i < 10; i++)
        for (int m = 0; m < 20; m++) Aflat[20 * i + m] = row[i].c[MONO[m][0]][MONO[m][1]][MONO[m][2]];
}

/* ------------------------------------------------------------------------------------------------------------------
 * opengv::math::Sturm (Sturm.cpp:150-492), dimension = number of coefficients (highest power first). */
#define STURM_MAXDIM 12
typedef struct { int dim; double C[STURM_MAXDIM][STURM_MAXDIM]; } sturm_t;
static void sturm_init(sturm_t *S, const double *p, int dim) {
    S->dim = dim;
    memset(S->C, 0, sizeof(S->C));
    for (int i = 0; i < dim; i++) S->C[0][i] = p[i];
    for (int i = 1; i < dim; i++) S->C[1][i] = S->C[0][i - 1] * (double) (dim - i);
    for (int i = 2; i < dim; i++) { /* computeNegatedRemainder, :445-464 */
        const double *p1 = &S->C[i - 2][i - 2], *p2 = &S->C[i - 1][i - 1];
        const int n1 = dim - (i - 2), n2 = n1 - 1;
        const double f1 = p1[0] / p2[0], f2 = p1[1] / p2[0], f3 = (-p2[1] * p1[0]) / (p2[0] * p2[0]);
        double r[STURM_MAXDIM];
        for (int k = 0; k < n1; k++) {
            const double a = k < n2 ? f1 * p2[k] : 0.0;
            const double b = k >= 1 ? f2 * p2[k - 1] : 0.0;
            const double c = k >= 1 ? f3 * p2[k - 1] : 0.0;
            r[k] = ((-p1[k] + a) + b) + c;
        }
        for (int k = 0; k < dim - i; k++) S->C[i][i + k] = r[2 + k];
    }
}
static size_t sturm_chain(const sturm_t *S, double bound) { /* evaluateChain2, :394-442 */
    const int dim = S->dim;
    double mono[STURM_MAXDIM], sign;
    mono[dim - 1] = 1.0;
    for (int i = 2; i <= dim; i++) mono[dim - i] = mono[dim - i + 1] * bound;
    int positive = 0, changes = 0;
    for (int i = 0; i < dim; i++) {
        sign = 0.0;
        for (int j = i; j < dim; j++) sign += S->C[i][j] * mono[j];
        if (i == 0) {
            positive = sign > 0.0;
            continue;
        }
        if (positive) {
            if (sign < 0.0) { changes++; positive = 0; }
        } else if (sign > 0.0) { changes++; positive = 1; }
    }
    return (size_t) changes;
}
static double sturm_bound(const sturm_t *S) { /* computeLagrangianBound, :466-492 */
    double c[STURM_MAXDIM];
    const int n = S->dim - 1;
    for (int i = 0; i < n; i++) c[i] = pow(fabs(S->C[0][i + 1] / S->C[0][0]), 1.0 / (double) (i + 1));
    int j = 0;
    double max1 = -1.0, max2 = -1.0;
    for (int i = 0; i < n; i++)
        if (c[i] > max1) { j = i; max1 = c[i]; }
    c[j] = -1.0;
    for (int i = 0; i < n; i++)
        if (c[i] > max2) max2 = c[i];
    return max1 + max2;
}
typedef struct { double lo, hi; size_t loCh, hiCh; } bracket_t;
static int sturm_find_roots(const sturm_t *S, double *roots, int cap) { /* findRoots + bracketRoots(eps = -1), :275-350 */
    const double bound = sturm_bound(S);
    int qcap = 64, head = 0, tail = 0, nroots = 0;
    bracket_t *q = (bracket_t *) malloc(sizeof(bracket_t) * (size_t) qcap);
    bracket_t b0 = {-bound, bound, sturm_chain(S, -bound), sturm_chain(S, bound)};
    q[tail++] = b0;
    const double eps = bound / (10.0 * (double) (b0.loCh - b0.hiCh)); /* size_t difference, as in the reference */
    while (head < tail) {
        const bracket_t b = q[head++];
        const size_t nr = b.loCh - b.hiCh;
        const double center = (b.hi + b.lo) / 2.0;
        int dividable = 1;
        if (nr == 1 && (b.hi - b.lo) < eps) dividable = 0;
        else if (nr == 0) dividable = 0;
        else if (center == b.hi || center == b.lo) dividable = 0;
        if (dividable) {
            if (tail + 2 > qcap) {
                qcap *= 2;
                q = (bracket_t *) realloc(q, sizeof(bracket_t) * (size_t) qcap);
            }
            const size_t ch = sturm_chain(S, center);
            bracket_t lo = {b.lo, center, b.loCh, ch}, hi = {center, b.hi, ch, b.hiCh};
            q[tail++] = lo;
            q[tail++] = hi;
        } else if (nr > 0 && nroots < cap)
            roots[nroots++] = 0.5 * (b.lo + b.hi);
    }
    free(q);
    const int dim = S->dim;
    for (int r = 0; r < nroots; r++)
        for (int k = 0; k < 5; k++) { /* five Newton steps, :286-300 */
            double mono[STURM_MAXDIM], v = 0, d = 0;
            mono[dim - 1] = 1.0;
            for (int i = 2; i <= dim; i++) mono[dim - i] = mono[dim - i + 1] * roots[r];
            for (int j = 0; j < dim; j++) v += S->C[0][j] * mono[j];
            for (int j = 0; j < dim; j++) d += S->C[1][j] * mono[j];
            roots[r] = roots[r] - (v / d);
        }
    return nroots;
}
int orc_sturm_roots(const double *coeffs, int ncoef, double *roots) {
    sturm_t S;
    if (ncoef > STURM_MAXDIM) return -1;
    sturm_init(&S, coeffs, ncoef);
    return sturm_find_roots(&S, roots, ncoef - 1);
}

/* ------------------------------------------------------------------------------------------------------------------
 * Eigen::LevenbergMarquardt<NumericalDiff<F>> (MINPACK lmdif with Eigen's column-pivoting QR). */
typedef void (*lm_fn)(const double *x, double *fvec, void *ctx);
static double vnorm(const double *v, int n) { /* stableNorm / blueNorm: a plain two-norm is the same to rounding here */
    double s = 0;
    for (int i = 0; i < n; i++) s += v[i] * v[i];
    return sqrt(s);
}
static void givens_make(double p, double q, double *c, double *s) { /* Jacobi.h makeGivens, real case */
    if (q == 0) {
        *c = p < 0 ? -1 : 1;
        *s = 0;
    } else if (p == 0) {
        *c = 0;
        *s = q < 0 ? 1 : -1;
    } else if (fabs(p) > fabs(q)) {
        double t = q / p, u = sqrt(1 + t * t);
        if (p < 0) u = -u;
        *c = 1 / u;
        *s = -t * *c;
    } else {
        double t = p / q, u = sqrt(1 + t * t);
        if (q < 0) u = -u;
        *s = -1 / u;
        *c = -t * *s;
    }
}
/* s: n x n column-major copy of R (upper) whose strict lower part is scratch; qrsolv.h:17-88 */
static void lm_qrsolv(double *s, int n, const int *ipvt, const double *diag, const double *qtb, double *x, double *sdiag) {
    double wa[8];
    for (int j = 0; j < n; j++) {
        x[j] = s[j * n + j];
        wa[j] = qtb[j];
        for (int i = j + 1; i < n; i++) s[j * n + i] = s[i * n + j]; /* lower = R^T */
    }
    for (int j = 0; j < n; j++) {
        const int l = ipvt[j];
        if (diag[l] == 0) break;
        for (int k = j; k < n; k++) sdiag[k] = 0;
        sdiag[j] = diag[l];
        double qtbpj = 0;
        for (int k = j; k < n; k++) {
            double c, sn;
            givens_make(-s[k * n + k], sdiag[k], &c, &sn);
            s[k * n + k] = c * s[k * n + k] + sn * sdiag[k];
            double temp = c * wa[k] + sn * qtbpj;
            qtbpj = -sn * wa[k] + c * qtbpj;
            wa[k] = temp;
            for (int i = k + 1; i < n; i++) {
                temp = c * s[k * n + i] + sn * sdiag[i];
                sdiag[i] = -sn * s[k * n + i] + c * sdiag[i];
                s[k * n + i] = temp;
            }
        }
    }
    int nsing = 0;
    while (nsing < n && sdiag[nsing] != 0) nsing++;
    for (int j = nsing; j < n; j++) wa[j] = 0;
    /* s.topLeftCorner(nsing,nsing).transpose().triangularView<Upper>().solveInPlace(wa): upper system U(i,j) = s(j,i) (lower part) */
    for (int i = nsing - 1; i >= 0; i--) {
        double sum = wa[i];
        for (int j = i + 1; j < nsing; j++) sum -= s[i * n + j] * wa[j];
        wa[i] = sum / s[i * n + i];
    }
    for (int j = 0; j < n; j++) {
        double d = s[j * n + j];
        s[j * n + j] = x[j];
        sdiag[j] = d;
    }
    for (int j = 0; j < n; j++) x[ipvt[j]] = wa[j];
}
/* R: m x n column-major factor (upper part = R); lmpar.h:160-296 */
static void lm_lmpar(const double *R, int ldr, int n, int rank, const int *ipvt, const double *diag, const double *qtb, double delta,
                     double *par, double *x) {
    double wa1[8], wa2[8], sdiag[8], s[64];
    for (int j = 0; j < n; j++) wa1[j] = j < rank ? qtb[j] : 0;
    for (int i = rank - 1; i >= 0; i--) {
        double sum = wa1[i];
        for (int j = i + 1; j < rank; j++) sum -= R[(size_t) j * ldr + i] * wa1[j];
        wa1[i] = sum / R[(size_t) i * ldr + i];
    }
    for (int j = 0; j < n; j++) x[ipvt[j]] = wa1[j];
    int iter = 0;
    for (int j = 0; j < n; j++) wa2[j] = diag[j] * x[j];
    double dxnorm = vnorm(wa2, n), fp = dxnorm - delta;
    if (fp <= 0.1 * delta) {
        *par = 0;
        return;
    }
    double parl = 0;
    if (rank == n) {
        for (int j = 0; j < n; j++) wa1[j] = diag[ipvt[j]] * wa2[ipvt[j]] / dxnorm;
        for (int j = 0; j < n; j++) { /* R^T (lower) forward substitution */
            double sum = wa1[j];
            for (int i = 0; i < j; i++) sum -= R[(size_t) j * ldr + i] * wa1[i];
            wa1[j] = sum / R[(size_t) j * ldr + j];
        }
        const double temp = vnorm(wa1, n);
        parl = fp / delta / temp / temp;
    }
    for (int j = 0; j < n; j++) {
        double sum = 0;
        for (int i = 0; i <= j; i++) sum += R[(size_t) j * ldr + i] * qtb[i];
        wa1[j] = sum / diag[ipvt[j]];
    }
    const double gnorm = vnorm(wa1, n);
    double paru = gnorm / delta;
    if (paru == 0) paru = DBL_MIN / fmin(delta, 0.1);
    *par = fmax(*par, parl);
    *par = fmin(*par, paru);
    if (*par == 0) *par = gnorm / dxnorm;
    for (int j = 0; j < n; j++)
        for (int i = 0; i < n; i++) s[j * n + i] = R[(size_t) j * ldr + i];
    for (;;) {
        ++iter;
        if (*par == 0) *par = fmax(DBL_MIN, 0.001 * paru);
        const double sp = sqrt(*par);
        for (int j = 0; j < n; j++) wa1[j] = sp * diag[j];
        lm_qrsolv(s, n, ipvt, wa1, qtb, x, sdiag);
        for (int j = 0; j < n; j++) wa2[j] = diag[j] * x[j];
        dxnorm = vnorm(wa2, n);
        double temp = fp;
        fp = dxnorm - delta;
        if (fabs(fp) <= 0.1 * delta || (parl == 0 && fp <= temp && temp < 0) || iter == 10) break;
        for (int j = 0; j < n; j++) wa1[j] = diag[ipvt[j]] * (wa2[ipvt[j]] / dxnorm);
        for (int j = 0; j < n; j++) {
            wa1[j] /= sdiag[j];
            temp = wa1[j];
            for (int i = j + 1; i < n; i++) wa1[i] -= s[j * n + i] * temp;
        }
        temp = vnorm(wa1, n);
        const double parc = fp / delta / temp / temp;
        if (fp > 0) parl = fmax(parl, *par);
        if (fp < 0) paru = fmin(paru, *par);
        *par = fmax(parl, *par + parc);
    }
    if (iter == 0) *par = 0;
}
/* returns the number of function evaluations; x[n] in/out, n <= 8.  info[0] = outer iterations, info[1] = status code
 * (1 both tolerances, 2 ftol, 3 xtol, 5 maxfev, 6 ftol too small, 7 xtol too small, 8 gtol too small, 4 gradient). */
static int lm_minimize(lm_fn fn, void *ctx, double *x, int n, int m, double ftol, double xtol, int maxfev, int *info) {
    double *fvec = (double *) malloc(sizeof(double) * (size_t) m * (size_t) (n + 4));
    double *wa4 = fvec + m, *val1 = wa4 + m, *val2 = val1 + m, *fjac = val2 + m; /* fjac m x n column-major */
    double diag[8], qtf[8], wa1[8], wa2[8], wa3[8], hc[8], xt[8];
    int perm[8], status = 0, nfev = 1, iter = 1;
    const double factor = 100.0, gtol = 0.0, fdeps = sqrt(DBL_EPS);
    double par = 0, delta = 0, xnorm = 0, temp = 0;
    fn(x, fvec, ctx);
    double fnorm = vnorm(fvec, m);
    while (!status) {
        /* NumericalDiff::df, Forward */
        memcpy(xt, x, sizeof(double) * (size_t) n);
        fn(xt, val1, ctx);
        nfev++;
        for (int j = 0; j < n; j++) {
            double h = fdeps * fabs(xt[j]);
            if (h == 0) h = fdeps;
            xt[j] += h;
            fn(xt, val2, ctx);
            nfev++;
            xt[j] = x[j];
            for (int i = 0; i < m; i++) fjac[(size_t) j * m + i] = (val2[i] - val1[i]) / h;
        }
        for (int j = 0; j < n; j++) wa2[j] = vnorm(fjac + (size_t) j * m, m);
        cpqr_info qi;
        cpqr(fjac, m, n, hc, perm, &qi);
        const int rank = cpqr_rank(fjac, &qi);
        if (iter == 1) {
            for (int j = 0; j < n; j++) diag[j] = wa2[j] == 0 ? 1 : wa2[j];
            for (int j = 0; j < n; j++) wa3[j] = diag[j] * x[j];
            xnorm = vnorm(wa3, n);
            delta = factor * xnorm;
            if (delta == 0) delta = factor;
        }
        memcpy(wa4, fvec, sizeof(double) * (size_t) m);
        for (int k = 0; k < n; k++) householder_apply_left(wa4 + k, m, m - k, 1, fjac + (size_t) k * m + k + 1, hc[k]);
        for (int j = 0; j < n; j++) qtf[j] = wa4[j];
        double gnorm = 0;
        if (fnorm != 0)
            for (int j = 0; j < n; j++)
                if (wa2[perm[j]] != 0) {
                    double sum = 0;
                    for (int i = 0; i <= j; i++) sum += fjac[(size_t) j * m + i] * (qtf[i] / fnorm);
                    gnorm = fmax(gnorm, fabs(sum / wa2[perm[j]]));
                }
        if (gnorm <= gtol) {
            status = 4;
            break;
        }
        for (int j = 0; j < n; j++) diag[j] = fmax(diag[j], wa2[j]);
        double ratio;
        do {
            lm_lmpar(fjac, m, n, rank, perm, diag, qtf, delta, &par, wa1);
            for (int j = 0; j < n; j++) {
                wa1[j] = -wa1[j];
                wa2[j] = x[j] + wa1[j];
                wa3[j] = diag[j] * wa1[j];
            }
            const double pnorm = vnorm(wa3, n);
            if (iter == 1) delta = fmin(delta, pnorm);
            fn(wa2, wa4, ctx);
            ++nfev;
            const double fnorm1 = vnorm(wa4, m);
            double actred = -1;
            if (0.1 * fnorm1 < fnorm) actred = 1 - (fnorm1 / fnorm) * (fnorm1 / fnorm);
            for (int i = 0; i < n; i++) { /* wa3 = R * (P^-1 wa1) */
                double sum = 0;
                for (int j = i; j < n; j++) sum += fjac[(size_t) j * m + i] * wa1[perm[j]];
                wa3[i] = sum;
            }
            const double t1 = vnorm(wa3, n) / fnorm, t2 = sqrt(par) * pnorm / fnorm;
            const double temp1 = t1 * t1, temp2 = t2 * t2;
            const double prered = temp1 + temp2 / 0.5, dirder = -(temp1 + temp2);
            ratio = 0;
            if (prered != 0) ratio = actred / prered;
            if (ratio <= 0.25) {
                if (actred >= 0) temp = 0.5;
                if (actred < 0) temp = 0.5 * dirder / (dirder + 0.5 * actred);
                if (0.1 * fnorm1 >= fnorm || temp < 0.1) temp = 0.1;
                delta = temp * fmin(delta, pnorm / 0.1);
                par /= temp;
            } else if (!(par != 0 && ratio < 0.75)) {
                delta = pnorm / 0.5;
                par = 0.5 * par;
            }
            if (ratio >= 1e-4) {
                for (int j = 0; j < n; j++) {
                    x[j] = wa2[j];
                    wa2[j] = diag[j] * x[j];
                }
                memcpy(fvec, wa4, sizeof(double) * (size_t) m);
                xnorm = vnorm(wa2, n);
                fnorm = fnorm1;
                ++iter;
            }
            const int small = fabs(actred) <= ftol && prered <= ftol && 0.5 * ratio <= 1;
            if (small && delta <= xtol * xnorm) status = 1;
            else if (small) status = 2;
            else if (delta <= xtol * xnorm) status = 3;
            else if (nfev >= maxfev) status = 5;
            else if (fabs(actred) <= DBL_EPS && prered <= DBL_EPS && 0.5 * ratio <= 1) status = 6;
            else if (delta <= DBL_EPS * xnorm) status = 7;
            else if (gnorm <= DBL_EPS) status = 8;
        } while (!status && ratio < 1e-4);
    }
    if (info) {
        info[0] = iter;
        info[1] = status;
    }
    free(fvec);
    return nfev;
}

/* ------------------------------------------------------------------------------------------------------------------
 * fivept_nister_main (main.cpp:135-261) */
static void polish_fn(const double *x, double *fvec, void *ctx) { /* PollishCoefficientsFunctor, modules.cpp:472-510 */
    const double *A = (const double *) ctx;
    double mono[20];
    for (int m = 0; m < 20; m++) mono[m] = pow(x[0], MONO[m][0]) * pow(x[1], MONO[m][1]) * pow(x[2], MONO[m][2]);
    for (int i = 0; i < 10; i++) {
        double s = 0;
        for (int m = 0; m < 20; m++) s += A[20 * i + m] * mono[m];
        fvec[i] = s;
    }
}
static double poly_val(const double *p, int ncoef, double x) { /* polyVal, modules.cpp:371-382 */
    double v = 0;
    for (int power = ncoef; power > 0; power--) v += p[ncoef - power] * pow(x, power - 1);
    return v;
}
static void conv(const double *a, int na, const double *b, int nb, double *o) { /* products in the order of modules.cpp:384-464 */
    for (int k = 0; k < na + nb - 1; k++) {
        double s = 0;
        int first = 1;
        for (int i = 0; i < na; i++) {
            const int j = k - i;
            if (j < 0 || j >= nb) continue;
            if (first) { s = a[i] * b[j]; first = 0; }
            else s += a[i] * b[j];
        }
        o[k] = s;
    }
}
/* solve A1 X = A2 (10 x 10 each) by Gaussian elimination with full pivoting; the reference forms FullPivLU(A1).inverse() * A2 */
static int solve10(const double *A /* [10][20] */, double X[10][10]) {
    double M[10][20];
    int colp[10];
    memcpy(M, A, sizeof(M));
    for (int j = 0; j < 10; j++) colp[j] = j;
    for (int k = 0; k < 10; k++) {
        int pr = k, pc = k;
        double best = -1;
        for (int c = k; c < 10; c++)
            for (int r = k; r < 10; r++)
                if (fabs(M[r][c]) > best) { best = fabs(M[r][c]); pr = r; pc = c; }
        if (best == 0) return 0;
        if (pr != k)
            for (int c = 0; c < 20; c++) { double t = M[k][c]; M[k][c] = M[pr][c]; M[pr][c] = t; }
        if (pc != k) {
            for (int r = 0; r < 10; r++) { double t = M[r][k]; M[r][k] = M[r][pc]; M[r][pc] = t; }
            int t = colp[k]; colp[k] = colp[pc]; colp[pc] = t;
        }
        for (int r = k + 1; r < 10; r++) {
            const double f = M[r][k] / M[k][k];
            if (f == 0) continue;
            for (int c = k + 1; c < 20; c++) M[r][c] -= f * M[k][c];
            M[r][k] = 0;
        }
    }
    for (int c = 0; c < 10; c++) {
        double y[10];
        for (int r = 9; r >= 0; r--) {
            double s = M[r][10 + c];
            for (int j = r + 1; j < 10; j++) s -= M[r][j] * y[j];
            y[r] = s / M[r][r];
        }
        for (int r = 0; r < 10; r++) X[colp[r]][c] = y[r];
    }
    return 1;
}
static int fivept_nister_main(double EE[9][4], double E[10][9]) {
    double A[200], A3[10][10];
    orc_nister_compose_a(&EE[0][0], A);
    if (!solve10(A, A3)) return 0;
    double b[3][3][5]; /* b[row pair][column group]: 4 coefficients for groups 0,1, 5 for group 2 */
    for (int rp = 0; rp < 3; rp++)
        for (int g = 0; g < 3; g++) {
            const int c0 = g == 0 ? 0 : (g == 1 ? 3 : 6), w = g == 2 ? 4 : 3;
            for (int k = 0; k <= w; k++) {
                const double part1 = k >= 1 ? A3[4 + 2 * rp][c0 + k - 1] : 0.0;
                const double part2 = k < w ? A3[5 + 2 * rp][c0 + k] : 0.0;
                b[rp][g][k] = part1 - part2;
            }
        }
    double t1[11], t2[11], p1[8], p2[8], p3[7], q1[11], q2[11], q3[11], p10[11];
    conv(b[1][2], 5, b[0][1], 4, t1);
    conv(b[0][2], 5, b[1][1], 4, t2);
    for (int k = 0; k < 8; k++) p1[k] = t1[k] - t2[k];
    conv(b[0][2], 5, b[1][0], 4, t1);
    conv(b[1][2], 5, b[0][0], 4, t2);
    for (int k = 0; k < 8; k++) p2[k] = t1[k] - t2[k];
    conv(b[0][0], 4, b[1][1], 4, t1);
    conv(b[0][1], 4, b[1][0], 4, t2);
    for (int k = 0; k < 7; k++) p3[k] = t1[k] - t2[k];
    conv(p1, 8, b[2][0], 4, q1);
    conv(p2, 8, b[2][1], 4, q2);
    conv(p3, 7, b[2][2], 5, q3);
    for (int k = 0; k < 11; k++) p10[k] = (q1[k] + q2[k]) + q3[k];
    sturm_t S;
    sturm_init(&S, p10, 11);
    double roots[10];
    const int nr = sturm_find_roots(&S, roots, 10);
    for (int i = 0; i < nr; i++) {
        const double z = roots[i];
        double xyz[3] = {poly_val(p1, 8, z) / poly_val(p3, 7, z), poly_val(p2, 8, z) / poly_val(p3, 7, z), z};
        lm_minimize(polish_fn, A, xyz, 3, 10, 1e10 * DBL_EPS, 1e10 * DBL_EPS, 5, NULL); /* pollishCoefficients, modules.cpp:517-545 */
        double nrm = 0;
        for (int k = 0; k < 9; k++) {
            E[i][k] = ((xyz[0] * EE[k][0] + xyz[1] * EE[k][1]) + xyz[2] * EE[k][2]) + EE[k][3];
            nrm += E[i][k] * E[i][k];
        }
        nrm = sqrt(nrm);
        for (int k = 0; k < 9; k++) E[i][k] /= nrm;
    }
    return nr;
}
static int fivept_nister(const double *bv1, const double *bv2, const int *idx5, double E[10][9]) { /* methods.cpp:239-268 */
    double Q[5][9], EE[9][4];
    for (int i = 0; i < 5; i++) {
        const double *f = bv2 + 3 * idx5[i], *fp = bv1 + 3 * idx5[i]; /* the solver works on the inverse transformation */
        for (int a = 0; a < 3; a++)
            for (int c = 0; c < 3; c++) Q[i][3 * a + c] = f[c] * fp[a];
    }
    nister_nullspace(Q, EE);
    return fivept_nister_main(EE, E);
}
int orc_nister_nullspace(const double *bv1, const double *bv2, double *EEflat) {
    double Q[5][9], EE[9][4];
    for (int i = 0; i < 5; i++)
        for (int a = 0; a < 3; a++)
            for (int c = 0; c < 3; c++) Q[i][3 * a + c] = bv2[3 * i + c] * bv1[3 * i + a];
    nister_nullspace(Q, EE);
    memcpy(EEflat, EE, sizeof(EE));
    return 0;
}
int orc_fivept_nister(const double *bv1, const double *bv2, double *Eflat) {
    const int idx[5] = {0, 1, 2, 3, 4};
    double E[10][9];
    const int n = fivept_nister(bv1, bv2, idx, E);
    memcpy(Eflat, E, sizeof(double) * 9 * (size_t) n);
    return n;
}

/* ------------------------------------------------------------------------------------------------------------------
 * 3 x 3 singular value decomposition E = U diag(s) V^T, s descending (one-sided Jacobi on the columns; any SVD gives the same
 * set of four decompositions below). */
static void svd3(const double *E /* row-major */, double U[9], double s[3], double V[9]) {
    double a[3][3], v[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
    for (int r = 0; r < 3; r++)
        for (int c = 0; c < 3; c++) a[r][c] = E[3 * r + c];
    for (int sweep = 0; sweep < 60; sweep++) {
        double off = 0;
        for (int p = 0; p < 2; p++)
            for (int q = p + 1; q < 3; q++) {
                double alpha = 0, beta = 0, gamma = 0;
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
    int ord[3] = {0, 1, 2};
    for (int c = 0; c < 3; c++) nrm[c] = sqrt(a[0][c] * a[0][c] + a[1][c] * a[1][c] + a[2][c] * a[2][c]);
    for (int i = 0; i < 2; i++)
        for (int j = i + 1; j < 3; j++)
            if (nrm[ord[j]] > nrm[ord[i]]) { int t = ord[i]; ord[i] = ord[j]; ord[j] = t; }
    for (int k = 0; k < 3; k++) {
        const int c = ord[k];
        s[k] = nrm[c];
        for (int r = 0; r < 3; r++) {
            V[3 * r + k] = v[r][c];
            U[3 * r + k] = nrm[c] > 1e-300 ? a[r][c] / nrm[c] : 0;
        }
    }
    /* the smallest singular value of an essential matrix is ~0 and leaves its left vector badly determined: the third
     * column of an orthogonal U is +-(u0 x u1) (the sign does not change the set of four decompositions) */
    {
        const double u0[3] = {U[0], U[3], U[6]}, u1[3] = {U[1], U[4], U[7]};
        const double c[3] = {u0[1] * u1[2] - u0[2] * u1[1], u0[2] * u1[0] - u0[0] * u1[2], u0[0] * u1[1] - u0[1] * u1[0]};
        const double sg = (c[0] * U[2] + c[1] * U[5] + c[2] * U[8]) < 0 ? -1.0 : 1.0;
        U[2] = sg * c[0];
        U[5] = sg * c[1];
        U[8] = sg * c[2];
    }
}
static double det3(const double *M) {
    return M[0] * (M[4] * M[8] - M[5] * M[7]) - M[1] * (M[3] * M[8] - M[5] * M[6]) + M[2] * (M[3] * M[7] - M[4] * M[6]);
}
static void mat3mul(const double *A, const double *B, double *C) {
    for (int r = 0; r < 3; r++)
        for (int c = 0; c < 3; c++) C[3 * r + c] = (A[3 * r] * B[c] + A[3 * r + 1] * B[3 + c]) + A[3 * r + 2] * B[6 + c];
}
static double dot3(const double *a, const double *b) { return (a[0] * b[0] + a[1] * b[1]) + a[2] * b[2]; }
/* Eigen's fixed-size 3x3 * 3x1: rows 0,1 form one SSE2 packet (terms added in order), row 2 is a scalar coefficient whose
 * three products are reduced as x0 + (x1 + x2) (Redux.h, redux_novec_unroller) */
static void matvec3(const double *R, const double *v, double *o) {
    for (int i = 0; i < 2; i++) o[i] = (R[3 * i] * v[0] + R[3 * i + 1] * v[1]) + R[3 * i + 2] * v[2];
    o[2] = R[6] * v[0] + (R[7] * v[1] + R[8] * v[2]);
}
/* reprojection score of one correspondence under X1 = R X2 + t (CentralRelativePoseSacProblem.cpp:176-197, :250-283) */
static double relpose_score(const double *R, const double *t, const double *f1, const double *f2) {
    double f2u[3], p[3], q[3], d[3];
    matvec3(R, f2, f2u);
    const double b0 = dot3(t, f1), b1 = dot3(t, f2u);
    const double a00 = dot3(f1, f1), a10 = dot3(f1, f2u), a01 = -a10, a11 = -dot3(f2u, f2u);
    const double invdet = 1.0 / (a00 * a11 - a10 * a01);
    const double l0 = (a11 * invdet) * b0 + (-a01 * invdet) * b1, l1 = (-a10 * invdet) * b0 + (a00 * invdet) * b1;
    for (int k = 0; k < 3; k++) p[k] = (l0 * f1[k] + (t[k] + l1 * f2u[k])) / 2;
    /* inverse transformation [R^T | -(R^T t)] applied to (p, 1): the 3x4 * 4x1 product again is one packet (rows 0,1, terms in
     * order) plus one scalar coefficient (row 2: (x0 + x1) + (x2 + x3)) */
    for (int k = 0; k < 2; k++) d[k] = -((R[k] * t[0] + R[3 + k] * t[1]) + R[6 + k] * t[2]);
    d[2] = -(R[2] * t[0] + (R[5] * t[1] + R[8] * t[2]));
    for (int k = 0; k < 2; k++) q[k] = ((R[k] * p[0] + R[3 + k] * p[1]) + R[6 + k] * p[2]) + d[k];
    q[2] = (R[2] * p[0] + R[5] * p[1]) + (R[8] * p[2] + d[2]);
    const double n1 = sqrt(dot3(p, p)), n2 = sqrt(dot3(q, q));
    const double e1 = 1.0 - ((f1[0] * (p[0] / n1) + f1[1] * (p[1] / n1)) + f1[2] * (p[2] / n1));
    const double e2 = 1.0 - ((f2[0] * (q[0] / n2) + f2[1] * (q[1] / n2)) + f2[2] * (q[2] / n2));
    return e1 + e2;
}
/* computeModelCoefficients for NISTER (CentralRelativePoseSacProblem.cpp:38-247); model = R (row-major 9) + t (3) */
static int relpose_model(const double *bv1, const double *bv2, const int *idx8, double *model) {
    double E[10][9];
    const int ne = fivept_nister(bv1, bv2, idx8, E);
    static const double W[9] = {0, -1, 0, 1, 0, 0, 0, 0, 1}, Wt[9] = {0, 1, 0, -1, 0, 0, 0, 0, 1};
    double bestQ = 1000000.0;
    int have = 0;
    for (int i = 0; i < ne; i++) {
        double U[9], s[3], V[9], Vt[9], T[9], Ra[9], Rb[9], ta[3];
        svd3(E[i], U, s, V);
        for (int r = 0; r < 3; r++)
            for (int c = 0; c < 3; c++) Vt[3 * r + c] = V[3 * c + r];
        mat3mul(U, W, T);
        mat3mul(T, Vt, Ra);
        mat3mul(U, Wt, T);
        mat3mul(T, Vt, Rb);
        for (int k = 0; k < 3; k++) ta[k] = s[0] * U[3 * k + 2];
        if (det3(Ra) < 0)
            for (int k = 0; k < 9; k++) Ra[k] = -Ra[k];
        if (det3(Rb) < 0)
            for (int k = 0; k < 9; k++) Rb[k] = -Rb[k];
        for (int j = 0; j < 4; j++) {
            const double *R = (j & 1) ? Rb : Ra;
            double t[3];
            for (int k = 0; k < 3; k++) t[k] = j < 2 ? ta[k] : -ta[k];
            double quality = 0;
            for (int k = 0; k < 8; k++) quality += relpose_score(R, t, bv1 + 3 * idx8[k], bv2 + 3 * idx8[k]);
            if (quality < bestQ) {
                bestQ = quality;
                memcpy(model, R, 72);
                memcpy(model + 9, t, 24);
                have = 1;
            }
        }
    }
    return have;
}
int orc_relpose_model(const double *bv1, const double *bv2, int n, const int *idx8, double *model12) {
    (void) n;
    return relpose_model(bv1, bv2, idx8, model12);
}
void orc_relpose_scores(const double *bv1, const double *bv2, int n, const double *model12, double *scores) {
    for (int i = 0; i < n; i++) scores[i] = relpose_score(model12, model12 + 9, bv1 + 3 * i, bv2 + 3 * i);
}

/* ------------------------------------------------------------------------------------------------------------------
 * optimize_nonlinear (methods.cpp:1082-1180): x = (t, cayley(R)), residual_i = score_i, LM with ftol = xtol = 10 eps */
typedef struct { const double *bv1, *bv2; const int *idx; int m; } opt_ctx;
static void cayley2rot(const double *c, double *R) { /* cayley.cpp:34-53 */
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
    for (int k = 0; k < 9; k++) R[k] = f * R[k];
}
static void rot2cayley(const double *R, double *c) { /* cayley.cpp:74-88: C = (R - I)(R + I)^-1 */
    double C1[9], C2[9], inv[9], C[9];
    for (int k = 0; k < 9; k++) {
        C1[k] = R[k] - (k % 4 == 0);
        C2[k] = R[k] + (k % 4 == 0);
    }
    const double d = det3(C2), id = 1.0 / d;
    inv[0] = (C2[4] * C2[8] - C2[5] * C2[7]) * id;
    inv[1] = (C2[2] * C2[7] - C2[1] * C2[8]) * id;
    inv[2] = (C2[1] * C2[5] - C2[2] * C2[4]) * id;
    inv[3] = (C2[5] * C2[6] - C2[3] * C2[8]) * id;
    inv[4] = (C2[0] * C2[8] - C2[2] * C2[6]) * id;
    inv[5] = (C2[2] * C2[3] - C2[0] * C2[5]) * id;
    inv[6] = (C2[3] * C2[7] - C2[4] * C2[6]) * id;
    inv[7] = (C2[1] * C2[6] - C2[0] * C2[7]) * id;
    inv[8] = (C2[0] * C2[4] - C2[1] * C2[3]) * id;
    mat3mul(C1, inv, C);
    c[0] = -C[5];
    c[1] = C[2];
    c[2] = -C[1];
}
static void opt_fn(const double *x, double *fvec, void *vctx) {
    const opt_ctx *c = (const opt_ctx *) vctx;
    double R[9];
    cayley2rot(x + 3, R);
    for (int i = 0; i < c->m; i++) fvec[i] = relpose_score(R, x, c->bv1 + 3 * c->idx[i], c->bv2 + 3 * c->idx[i]);
}
static void relpose_optimize(const double *bv1, const double *bv2, const int *inliers, int nIn, const double *model, double *out, int *info) {
    double x[6];
    memcpy(x, model + 9, 24);
    rot2cayley(model, x + 3);
    opt_ctx c = {bv1, bv2, inliers, nIn};
    const int nfev = lm_minimize(opt_fn, &c, x, 6, nIn, 10 * DBL_EPS, 10 * DBL_EPS, 1000, info);
    if (info) info[2] = nfev;
    cayley2rot(x + 3, out);
    memcpy(out + 9, x, 24);
}
void orc_relpose_optimize(const double *bv1, const double *bv2, int n, const int *inliers, int nIn, const double *model12, double *out12,
                          int *info3) {
    (void) n;
    relpose_optimize(bv1, bv2, inliers, nIn, model12, out12, info3);
}

/* ------------------------------------------------------------------------------------------------------------------
 * RANSAC (Ransac.hpp:45-143) with the std::mt19937 prefix Fisher-Yates sampler (SampleConsensusProblem.hpp:65-84). */
typedef struct { uint32_t mt[624]; int idx; } rp_mt;
static void rp_seed(rp_mt *g, uint32_t s) {
    g->mt[0] = s;
    for (int i = 1; i < 624; i++) g->mt[i] = 1812433253u * (g->mt[i - 1] ^ (g->mt[i - 1] >> 30)) + (uint32_t) i;
    g->idx = 624;
}
static uint32_t rp_next(rp_mt *g) {
    if (g->idx >= 624) {
        for (int i = 0; i < 624; i++) {
            uint32_t y = (g->mt[i] & 0x80000000u) | (g->mt[(i + 1) % 624] & 0x7fffffffu);
            g->mt[i] = g->mt[(i + 397) % 624] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
        }
        g->idx = 0;
    }
    uint32_t y = g->mt[g->idx++];
    y ^= y >> 11;
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= y >> 18;
    return y;
}
int orc_relpose_draw_samples(int n, int count, uint32_t seed, int *samples8) {
    rp_mt g;
    rp_seed(&g, seed);
    int *shuf = (int *) malloc(sizeof(int) * (size_t) n);
    for (int i = 0; i < n; i++) shuf[i] = i;
    for (int k = 0; k < count; k++) {
        for (int i = 0; i < 8; i++) {
            const int r = (int) (rp_next(&g) >> 1); /* uniform_int_distribution<int>(0, INT_MAX) on a 32-bit engine */
            const int j = i + (int) ((unsigned) r % (unsigned) (n - i));
            const int t = shuf[i]; shuf[i] = shuf[j]; shuf[j] = t;
        }
        memcpy(samples8 + 8 * k, shuf, 8 * sizeof(int));
    }
    free(shuf);
    return 0;
}
int orc_relpose_ransac(const double *bv1, const double *bv2, int n, int maxIterations, float errorThreshold, uint32_t seed, float fx,
                       float fy, double *ransacModel12, uint8_t *inlierMask, int *info) {
    float focal = fx + fy; /* multi_view_geometry.cpp:272-276 */
    focal /= 2.f;
    const double threshold = 2.0 * (1.0 - cos(atan((double) (errorThreshold / focal))));
    memset(inlierMask, 0, (size_t) n);
    info[0] = info[1] = 0;
    if (n < 8) return 0;
    rp_mt g;
    rp_seed(&g, seed);
    int *shuf = (int *) malloc(sizeof(int) * (size_t) n);
    for (int i = 0; i < n; i++) shuf[i] = i;
    int iterations = 0, best = -2147483647, have = 0;
    unsigned skipped = 0;
    const unsigned maxSkip = (unsigned) maxIterations * 10u;
    double k = 1.0, model[12];
    while ((double) iterations < k && skipped < maxSkip) {
        int s[8];
        for (int i = 0; i < 8; i++) {
            const int r = (int) (rp_next(&g) >> 1);
            const int j = i + (int) ((unsigned) r % (unsigned) (n - i));
            const int t = shuf[i]; shuf[i] = shuf[j]; shuf[j] = t;
        }
        memcpy(s, shuf, sizeof(s));
        if (!relpose_model(bv1, bv2, s, model)) {
            ++skipped;
            continue;
        }
        int count = 0;
        for (int i = 0; i < n; i++) count += relpose_score(model, model + 9, bv1 + 3 * i, bv2 + 3 * i) < threshold;
        if (count > best) {
            best = count;
            memcpy(ransacModel12, model, sizeof(model));
            have = 1;
            const double w = (double) best / (double) n;
            double pNo = 1.0 - pow(w, 8.0);
            pNo = fmax(DBL_EPS, pNo);
            pNo = fmin(1.0 - DBL_EPS, pNo);
            k = log(1.0 - 0.99) / log(pNo);
        }
        ++iterations;
        if (iterations > maxIterations) break;
    }
    free(shuf);
    info[0] = iterations;
    if (!have) return 0;
    int cnt = 0;
    for (int i = 0; i < n; i++) {
        inlierMask[i] = relpose_score(ransacModel12, ransacModel12 + 9, bv1 + 3 * i, bv2 + 3 * i) < threshold;
        cnt += inlierMask[i];
    }
    info[1] = cnt;
    return cnt < 10 ? 0 : 1;
}
int orc_compute_5pt(const double *bv1, const double *bv2, int n, int maxIterations, float errorThreshold, int optimize, uint32_t seed,
                    float fx, float fy, double *R_out, double *t_out, int *outliers, int *nOutliers) {
    *nOutliers = 0;
    if (n < 8) return 0; /* :242-245 */
    uint8_t *mask = (uint8_t *) malloc((size_t) n);
    double model[12], opt[12];
    int info[2];
    const int ok = orc_relpose_ransac(bv1, bv2, n, maxIterations, errorThreshold, seed, fx, fy, model, mask, info);
    if (ok) {
        if (optimize) {
            int *in = (int *) malloc(sizeof(int) * (size_t) n), nIn = 0;
            for (int i = 0; i < n; i++)
                if (mask[i]) in[nIn++] = i;
            relpose_optimize(bv1, bv2, in, nIn, model, opt, NULL);
            memcpy(model, opt, sizeof(opt));
            free(in);
        }
        memcpy(R_out, model, 72);
        memcpy(t_out, model + 9, 24);
        for (int i = 0; i < n; i++)
            if (!mask[i]) outliers[(*nOutliers)++] = i;
    }
    free(mask);
    return ok;
}
