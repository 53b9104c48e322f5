/* TEST INFRASTRUCTURE ONLY -- CPU restatement of the INTENDED algorithm of System::processPlane
 * (/root/reference/src/slam/src/system.cpp:177-342), SURVEY.md §8f-3.
 *
 * PINNED AGAINST THE REPAIRED REFERENCE (round 6).  As shipped the reference function has no defined behaviour:
 * cv::eigen2cv turns its 1x3 CV_32F point matrices into 3x1 CV_64F ones, every at<float>() after that reinterprets halves of doubles,
 * the RANSAC design matrix is never filled, the three sample indices come from a std::random_device-seeded generator that is
 * re-created in every iteration, and std::nth_element permutes the distances it then pairs with the points.  oracle/_ref holds the
 * function compiled from the reference's own source with exactly these four defects repaired (oracle/ref_shim_plane.cpp,
 * oracle/ref_plane_patch.sed: ref_find_plane_patched); tests/test_plane.py compares this file with it (poses to 1e-5 wherever no
 * decision hangs on float rounding, see orc_find_plane_margins), and tests/test_plane.py's GPU tests compare alva_find_plane with both.
 * What is restated here is what the code says it wants to do, statement by statement, on float copies of the points and with the
 * sample indices supplied by the caller:
 *   :205-215   plane through 3 sampled points = null vector of the 3x4 matrix [x y z 1]      (unit 4-vector (a, b, c, d))
 *   :222-229   skip unless |(a,b,c) x (0,0,1)| <= sin(5 deg)
 *   :231-236   dist_i = |a x + b y + c z + d| / |(a,b,c,d)|
 *   :238-246   score = k-th smallest distance, k = max((int)(0.2 N), 20); keep the hypothesis with the smallest score
 *   :249-259   inliers: dist < 1.4 * best score; fewer than 32 => no plane
 *   :272-291   refit: null vector of the N_in x 4 matrix [x y z 1]; origin = mean of the inliers
 *   :293-308   flip the normal away from Oc - origin, Oc = -R t of Utils::toPoseMat(Twc) (utils.cpp:54-79: it stores R^T | t)
 *   :310-331   pose = [Rodrigues(v ang / |v|) Rodrigues((1,0,0)) | origin], v = (1,0,0) x n, ang = atan2(|v|, n_x)
 *   output     Utils::toPoseArray(cv::Mat) (utils.cpp:29-52): columns of the rotation, then the translation
 * The two null vectors are taken from a double-precision Jacobi eigen-decomposition of A^T A (OpenCV would run a float SVD). */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include "alva_oracle.h"

static int cmp_float(const void *a, const void *b) {
    const float x = *(const float *) a, y = *(const float *) b;
    return (x > y) - (x < y);
}
/* eigenvector of the smallest eigenvalue of the symmetric 4x4 matrix M (cyclic Jacobi) */
static void smallest_eigvec4(const double M[16], double v[4]) {
    double A[4][4], V[4][4];
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++) {
            A[i][j] = M[4 * i + j];
            V[i][j] = i == j;
        }
    for (int sweep = 0; sweep < 60; sweep++) {
        double off = 0;
        for (int p = 0; p < 3; p++)
            for (int q = p + 1; q < 4; q++) off += A[p][q] * A[p][q];
        if (off < 1e-300) break;
        for (int p = 0; p < 3; p++)
            for (int q = p + 1; q < 4; q++) {
                if (fabs(A[p][q]) < 1e-300) continue;
                const double th = (A[q][q] - A[p][p]) / (2 * A[p][q]);
                const double t = (th >= 0 ? 1.0 : -1.0) / (fabs(th) + sqrt(th * th + 1)), c = 1 / sqrt(t * t + 1), s = t * c;
                for (int k = 0; k < 4; k++) {
                    const double akp = A[k][p], akq = A[k][q];
                    A[k][p] = c * akp - s * akq;
                    A[k][q] = s * akp + c * akq;
                }
                for (int k = 0; k < 4; k++) {
                    const double apk = A[p][k], aqk = A[q][k];
                    A[p][k] = c * apk - s * aqk;
                    A[q][k] = s * apk + c * aqk;
                }
                for (int k = 0; k < 4; k++) {
                    const double vkp = V[k][p], vkq = V[k][q];
                    V[k][p] = c * vkp - s * vkq;
                    V[k][q] = s * vkp + c * vkq;
                }
            }
    }
    int m = 0;
    for (int i = 1; i < 4; i++)
        if (A[i][i] < A[m][m]) m = i;
    for (int k = 0; k < 4; k++) v[k] = V[k][m];
}
static void rodrigues(const double r[3], double R[9]) {
    const double th = sqrt(r[0] * r[0] + r[1] * r[1] + r[2] * r[2]);
    if (th < 1e-300) {
        for (int i = 0; i < 9; i++) R[i] = i % 4 == 0;
        return;
    }
    const double k[3] = {r[0] / th, r[1] / th, r[2] / th}, c = cos(th), s = sin(th), c1 = 1 - c;
    const double K[9] = {0, -k[2], k[1], k[2], 0, -k[0], -k[1], k[0], 0};
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) R[3 * i + j] = c * (i == j) + c1 * k[i] * k[j] + s * K[3 * i + j];
}

/* pts: n x 3 world points; pose7: Twc (t, q xyzw); samples3: numIterations x 3 indices; out16: plane pose.  Returns 1 if found. */
int orc_find_plane(const double *pts, int n, const double *pose7, const int *samples3, int numIterations, float *out16) {
    if (n < 32) return 0; /* :181 */
    float *P = (float *) malloc(sizeof(float) * 3 * (size_t) n), *dists = (float *) malloc(sizeof(float) * (size_t) n * 3);
    float *best_d = dists + n, *sorted = dists + 2 * (size_t) n;
    for (int i = 0; i < 3 * n; i++) P[i] = (float) pts[i];
    for (int i = 0; i < n; i++) best_d[i] = 0.f;
    float bestDist = 1e10f;
    const int kth = (int) (0.2 * n) > 20 ? (int) (0.2 * n) : 20;
    const float sinTh = sinf(5.0f * 3.14159265358979323846f / 180.0f);
    for (int it = 0; it < numIterations; it++) {
        const float *p0 = P + 3 * samples3[3 * it], *p1 = P + 3 * samples3[3 * it + 1], *p2 = P + 3 * samples3[3 * it + 2];
        const double u[3] = {(double) p1[0] - p0[0], (double) p1[1] - p0[1], (double) p1[2] - p0[2]};
        const double w[3] = {(double) p2[0] - p0[0], (double) p2[1] - p0[1], (double) p2[2] - p0[2]};
        double pl[4] = {u[1] * w[2] - u[2] * w[1], u[2] * w[0] - u[0] * w[2], u[0] * w[1] - u[1] * w[0], 0};
        pl[3] = -(pl[0] * p0[0] + pl[1] * p0[1] + pl[2] * p0[2]);
        const double nn = sqrt(pl[0] * pl[0] + pl[1] * pl[1] + pl[2] * pl[2] + pl[3] * pl[3]);
        if (!(nn > 0)) continue; /* collinear sample: no null vector of rank 3 */
        const float a = (float) (pl[0] / nn), b = (float) (pl[1] / nn), c = (float) (pl[2] / nn), d = (float) (pl[3] / nn);
        if (sqrt((double) b * b + (double) a * a) > sinTh) continue;
        const float f = 1.0f / sqrtf(a * a + b * b + c * c + d * d);
        for (int i = 0; i < n; i++) dists[i] = fabsf(P[3 * i] * a + P[3 * i + 1] * b + P[3 * i + 2] * c + d) * f;
        memcpy(sorted, dists, sizeof(float) * (size_t) n);
        qsort(sorted, (size_t) n, sizeof(float), cmp_float);
        const float med = sorted[kth];
        if (med < bestDist) {
            bestDist = med;
            memcpy(best_d, dists, sizeof(float) * (size_t) n);
        }
    }
    const float threshold = 1.4f * bestDist;
    double M[16] = {0}, sum[3] = {0, 0, 0};
    float origin[3] = {0, 0, 0};
    int nin = 0;
    for (int i = 0; i < n; i++)
        if (best_d[i] < threshold) {
            const double r[4] = {P[3 * i], P[3 * i + 1], P[3 * i + 2], 1.0};
            for (int x = 0; x < 4; x++)
                for (int y = 0; y < 4; y++) M[4 * x + y] += r[x] * r[y];
            for (int k = 0; k < 3; k++) origin[k] += P[3 * i + k];
            nin++;
        }
    (void) sum;
    int found = 0;
    if (nin >= 32) {
        double v[4];
        smallest_eigvec4(M, v);
        float a = (float) v[0], b = (float) v[1], c = (float) v[2];
        for (int k = 0; k < 3; k++) origin[k] = origin[k] * (1.0f / (float) nin);
        const float f = 1.0f / sqrtf(a * a + b * b + c * c);
        /* camPose = [R^T | t] of Twc; Oc = -(R^T)^T t = -R t */
        const double qx = pose7[3], qy = pose7[4], qz = pose7[5], qw = pose7[6];
        const double R[9] = {1 - 2 * (qy * qy + qz * qz), 2 * (qx * qy - qz * qw), 2 * (qx * qz + qy * qw),
                             2 * (qx * qy + qz * qw), 1 - 2 * (qx * qx + qz * qz), 2 * (qy * qz - qx * qw),
                             2 * (qx * qz - qy * qw), 2 * (qy * qz + qx * qw), 1 - 2 * (qx * qx + qy * qy)};
        float Oc[3];
        for (int i = 0; i < 3; i++)
            Oc[i] = -((float) R[3 * i] * (float) pose7[0] + (float) R[3 * i + 1] * (float) pose7[1] + (float) R[3 * i + 2] * (float) pose7[2]);
        const float mx[3] = {Oc[0] - origin[0], Oc[1] - origin[1], Oc[2] - origin[2]};
        if (mx[0] * a + mx[1] * b + mx[2] * c > 0) {
            a = -a; b = -b; c = -c;
        }
        const float nx = a * f, ny = b * f, nz = c * f;
        const float v3[3] = {0.f, -nz, ny}; /* (1,0,0) x n */
        const float sa = (float) sqrt((double) v3[0] * v3[0] + (double) v3[1] * v3[1] + (double) v3[2] * v3[2]), ca = nx;
        const float ang = atan2f(sa, ca);
        const double r1[3] = {v3[0] * ang / sa, v3[1] * ang / sa, v3[2] * ang / sa}, r2[3] = {1, 0, 0};
        double R1[9], R2[9], RR[9];
        rodrigues(r1, R1);
        rodrigues(r2, R2);
        for (int i = 0; i < 3; i++)
            for (int j = 0; j < 3; j++) {
                float s = 0.f;
                for (int k = 0; k < 3; k++) s += (float) R1[3 * i + k] * (float) R2[3 * k + j];
                RR[3 * i + j] = s;
            }
        for (int col = 0; col < 3; col++) {
            for (int row = 0; row < 3; row++) out16[4 * col + row] = (float) RR[3 * row + col];
            out16[4 * col + 3] = 0.f;
        }
        for (int k = 0; k < 3; k++) out16[12 + k] = origin[k];
        out16[15] = 1.f;
        found = 1;
    }
    free(P);
    free(dists);
    return found;
}

/* How far a scene is from a float-sensitive decision of orc_find_plane (the patched reference runs a float SVD where this file and the
 * HIP path take cross products / a double eigen-decomposition: plane parameters agree to ~1e-6, so a comparison is meaningful where no
 * decision hangs on less than that).  out3[0] = best score, out3[1] = the smallest score of any OTHER hypothesis, out3[2] = the
 * smallest |dist_i - 1.4 best| / (1.4 best) over the points.  Returns the number of hypotheses that passed the orientation test. */
int orc_find_plane_margins(const double *pts, int n, const int *samples3, int numIterations, float *out3) {
    out3[0] = out3[1] = 1e10f;
    out3[2] = 0.f;
    if (n < 32) return 0;
    float *P = (float *) malloc(sizeof(float) * 3 * (size_t) n), *dists = (float *) malloc(sizeof(float) * (size_t) n * 3);
    float *best_d = dists + n, *sorted = dists + 2 * (size_t) n;
    for (int i = 0; i < 3 * n; i++) P[i] = (float) pts[i];
    const int kth = (int) (0.2 * n) > 20 ? (int) (0.2 * n) : 20;
    const float sinTh = sinf(5.0f * 3.14159265358979323846f / 180.0f);
    int accepted = 0;
    for (int it = 0; it < numIterations; it++) {
        const float *p0 = P + 3 * samples3[3 * it], *p1 = P + 3 * samples3[3 * it + 1], *p2 = P + 3 * samples3[3 * it + 2];
        const double u[3] = {(double) p1[0] - p0[0], (double) p1[1] - p0[1], (double) p1[2] - p0[2]};
        const double w[3] = {(double) p2[0] - p0[0], (double) p2[1] - p0[1], (double) p2[2] - p0[2]};
        double pl[4] = {u[1] * w[2] - u[2] * w[1], u[2] * w[0] - u[0] * w[2], u[0] * w[1] - u[1] * w[0], 0};
        pl[3] = -(pl[0] * p0[0] + pl[1] * p0[1] + pl[2] * p0[2]);
        const double nn = sqrt(pl[0] * pl[0] + pl[1] * pl[1] + pl[2] * pl[2] + pl[3] * pl[3]);
        if (!(nn > 0)) continue;
        const float a = (float) (pl[0] / nn), b = (float) (pl[1] / nn), c = (float) (pl[2] / nn), d = (float) (pl[3] / nn);
        if (sqrt((double) b * b + (double) a * a) > sinTh) continue;
        accepted++;
        const float f = 1.0f / sqrtf(a * a + b * b + c * c + d * d);
        for (int i = 0; i < n; i++) dists[i] = fabsf(P[3 * i] * a + P[3 * i + 1] * b + P[3 * i + 2] * c + d) * f;
        memcpy(sorted, dists, sizeof(float) * (size_t) n);
        qsort(sorted, (size_t) n, sizeof(float), cmp_float);
        const float med = sorted[kth];
        if (med < out3[0]) {
            out3[1] = out3[0];
            out3[0] = med;
            memcpy(best_d, dists, sizeof(float) * (size_t) n);
        } else if (med < out3[1]) out3[1] = med;
    }
    if (accepted) {
        const float thr = 1.4f * out3[0];
        float m = 1e10f;
        for (int i = 0; i < n; i++) {
            const float g = fabsf(best_d[i] - thr) / thr;
            if (g < m) m = g;
        }
        out3[2] = m;
    }
    free(P);
    free(dists);
    return accepted;
}
