// f3: the plane under the map points of the current frame -- the INTENDED algorithm of System::processPlane
// (src/slam/src/system.cpp:177-342; caller System::findPlane :123-137).
//
// As shipped the reference function has no defined behaviour (reinterpreted point matrices, a design matrix that is
// never filled, distances permuted by nth_element before they are paired with the points, a random_device-seeded generator per
// iteration).  Parity is pinned against the reference's OWN function compiled with exactly those four defects repaired
// (oracle/ref_shim_plane.cpp, oracle/ref_plane_patch.sed; tests/test_plane.py::test_gpu_equals_the_repaired_reference).  This is what
// its statements compute, on float copies of the points:
//   RANSAC (:205-246)   per iteration a plane through 3 sampled points (unit 4-vector a, b, c, d), skipped unless
//                       |(a,b,c) x (0,0,1)| <= sin 5 deg; score = k-th smallest |a x + b y + c z + d| / |(a,b,c,d)| with
//                       k = max((int)(0.2 N), 20); the smallest score wins (the first one on ties)
//   inliers (:249-270)  distance < 1.4 x that score (all points when no iteration survived the orientation test); < 32 => none
//   refit (:272-291)    null vector of the N_in x 4 matrix [x y z 1], origin = mean of the inliers
//   pose (:293-331)     normal flipped away from Oc - origin (Oc = -R t, utils.cpp:54-79), rotation Rodrigues(v ang / |v|) *
//                       Rodrigues((1,0,0)), v = (1,0,0) x n; output layout of Utils::toPoseArray(cv::Mat) (utils.cpp:29-52)
// One launch scores every iteration (one workgroup each: distances in LDS, radix select of the k-th smallest), a second one
// picks the winner, classifies the points and reduces the 4 x 4 normal matrix; the 4 x 4 eigenvector and the pose are host work.
#include "common.hpp"
#include <algorithm>
#include <cmath>
#include <ctime>
#include <random>
#include <vector>

namespace {

constexpr int PL_NT = 256;
struct PlaneArgs {
    const double *pts;   // [n][3]
    const int *samples;  // [iters][3]
    int n, iters, kth;
    float sinTh;
    float *scores;  // [iters]: k-th smallest distance, +inf when the iteration is skipped
    float *planes;  // [iters][4]
};

__device__ __forceinline__ bool plane_of(const PlaneArgs &A, int it, float &a, float &b, float &c, float &d) {
    const int i0 = A.samples[3 * it], i1 = A.samples[3 * it + 1], i2 = A.samples[3 * it + 2];
    const float p0[3] = {(float) A.pts[3 * i0], (float) A.pts[3 * i0 + 1], (float) A.pts[3 * i0 + 2]};
    const float p1[3] = {(float) A.pts[3 * i1], (float) A.pts[3 * i1 + 1], (float) A.pts[3 * i1 + 2]};
    const float p2[3] = {(float) A.pts[3 * i2], (float) A.pts[3 * i2 + 1], (float) A.pts[3 * i2 + 2]};
    const double u[3] = {(double) p1[0] - p0[0], (double) p1[1] - p0[1], (double) p1[2] - p0[2]};
    const double w[3] = {(double) p2[0] - p0[0], (double) p2[1] - p0[1], (double) p2[2] - p0[2]};
    double pl[4] = {u[1] * w[2] - u[2] * w[1], u[2] * w[0] - u[0] * w[2], u[0] * w[1] - u[1] * w[0], 0};
    pl[3] = -(pl[0] * p0[0] + pl[1] * p0[1] + pl[2] * p0[2]);
    const double nn = sqrt(pl[0] * pl[0] + pl[1] * pl[1] + pl[2] * pl[2] + pl[3] * pl[3]);
    if (!(nn > 0)) return false;
    a = (float) (pl[0] / nn);
    b = (float) (pl[1] / nn);
    c = (float) (pl[2] / nn);
    d = (float) (pl[3] / nn);
    return true;
}
__device__ __forceinline__ float plane_dist(const double *p, float a, float b, float c, float d, float f) {
    return fabsf((float) p[0] * a + (float) p[1] * b + (float) p[2] * c + d) * f;
}

__global__ void __launch_bounds__(PL_NT) k_plane_hyp(const PlaneArgs A) {
    extern __shared__ float pl_d[];  // n distances
    __shared__ int hist[256];
    __shared__ unsigned s_prefix;
    __shared__ int s_k;
    const int it = blockIdx.x, tid = threadIdx.x;
    float a = 0, b = 0, c = 0, d = 0;
    bool ok = plane_of(A, it, a, b, c, d);
    ok = ok && !(sqrt((double) b * b + (double) a * a) > A.sinTh);  // :222-229
    if (!ok) {
        if (tid == 0) A.scores[it] = INFINITY;
        return;
    }
    const float f = 1.0f / sqrtf(a * a + b * b + c * c + d * d);
    for (int i = tid; i < A.n; i += PL_NT) pl_d[i] = plane_dist(A.pts + 3 * (size_t) i, a, b, c, d, f);
    if (tid == 0) {
        s_prefix = 0;
        s_k = A.kth;
    }
    // k-th smallest (std::nth_element, :238-240): radix select over the bit patterns of the non-negative floats, one byte per pass
    unsigned mask = 0;
    for (int pass = 3; pass >= 0; pass--) {
        hist[tid] = 0;
        __syncthreads();
        const unsigned prefix = s_prefix;
        for (int i = tid; i < A.n; i += PL_NT) {
            const unsigned key = __float_as_uint(pl_d[i]);
            if ((key & mask) == prefix) atomicAdd(&hist[(key >> (8 * pass)) & 255u], 1);
        }
        __syncthreads();
        if (tid < 64) {  // first wavefront: 4 bins per lane, inclusive scan, locate the bin that holds rank k
            const int h0 = hist[4 * tid], h1 = hist[4 * tid + 1], h2 = hist[4 * tid + 2], h3 = hist[4 * tid + 3];
            int incl = h0 + h1 + h2 + h3;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const int t = __shfl_up(incl, o);
                if (tid >= o) incl += t;
            }
            const int excl = incl - (h0 + h1 + h2 + h3), k = s_k;
            if (k >= excl && k < incl) {
                int r = k - excl, bin = 4 * tid;
                if (r >= h0) { r -= h0; bin++;
                    if (r >= h1) { r -= h1; bin++;
                        if (r >= h2) { r -= h2; bin++; } } }
                s_prefix = prefix | ((unsigned) bin << (8 * pass));
                s_k = r;
            }
        }
        mask |= 0xffu << (8 * pass);
        __syncthreads();
    }
    if (tid == 0) {
        A.scores[it] = __uint_as_float(s_prefix);
        A.planes[4 * it] = a;
        A.planes[4 * it + 1] = b;
        A.planes[4 * it + 2] = c;
        A.planes[4 * it + 3] = d;
    }
}

struct PlaneFinish {
    double M[10];  // upper triangle of sum [x y z 1]^T [x y z 1] over the inliers
    double sum[3];
    int n_inliers, best;
    float best_score;
};
__global__ void __launch_bounds__(PL_NT) k_plane_finish(const PlaneArgs A, PlaneFinish *out) {
    __shared__ double red[PL_NT / 64][14];
    __shared__ int s_best;
    __shared__ float s_score;
    const int tid = threadIdx.x;
    if (tid == 0) {
        float bestDist = 1e10f;  // :193
        int best = -1;
        for (int it = 0; it < A.iters; it++)
            if (A.scores[it] < bestDist) {
                bestDist = A.scores[it];
                best = it;
            }
        s_best = best;
        s_score = bestDist;
    }
    __syncthreads();
    const int best = s_best;
    const float threshold = 1.4f * s_score;
    float a = 0, b = 0, c = 0, d = 0, f = 0;
    if (best >= 0) {
        a = A.planes[4 * best]; b = A.planes[4 * best + 1]; c = A.planes[4 * best + 2]; d = A.planes[4 * best + 3];
        f = 1.0f / sqrtf(a * a + b * b + c * c + d * d);
    }
    double acc[14];
#pragma unroll
    for (int k = 0; k < 14; k++) acc[k] = 0;
    for (int i = tid; i < A.n; i += PL_NT) {
        const double *p = A.pts + 3 * (size_t) i;
        const float dist = best >= 0 ? plane_dist(p, a, b, c, d, f) : 0.0f;  // no surviving iteration: the stored distances are all 0 (:192)
        if (dist < threshold) {
            const double r[4] = {(double) (float) p[0], (double) (float) p[1], (double) (float) p[2], 1.0};
            int t = 0;
#pragma unroll
            for (int x = 0; x < 4; x++)
#pragma unroll
                for (int y = x; y < 4; y++) acc[t++] += r[x] * r[y];
            acc[10] += r[0];
            acc[11] += r[1];
            acc[12] += r[2];
            acc[13] += 1.0;
        }
    }
    const int lane = tid & 63, wave = tid >> 6;
#pragma unroll
    for (int k = 0; k < 14; k++) {
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) acc[k] += __shfl_xor(acc[k], o);
        if (lane == 0) red[wave][k] = acc[k];
    }
    __syncthreads();
    if (tid < 14) {
        double v = 0;
        for (int w = 0; w < PL_NT / 64; w++) v += red[w][tid];
        if (tid < 10) out->M[tid] = v;
        else if (tid < 13) out->sum[tid - 10] = v;
        else out->n_inliers = (int) v;
    }
    if (tid == 0) {
        out->best = best;
        out->best_score = s_score;
    }
}

void smallest_eigvec4(const double M[16], double v[4]) {  // cyclic Jacobi on a symmetric 4 x 4
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
                if (std::fabs(A[p][q]) < 1e-300) continue;
                const double th = (A[q][q] - A[p][p]) / (2 * A[p][q]);
                const double t = (th >= 0 ? 1.0 : -1.0) / (std::fabs(th) + std::sqrt(th * th + 1)), c = 1 / std::sqrt(t * t + 1), s = t * c;
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
void rodrigues(const double r[3], double R[9]) {
    const double th = std::sqrt(r[0] * r[0] + r[1] * r[1] + r[2] * r[2]);
    if (th < 1e-300) {
        for (int i = 0; i < 9; i++) R[i] = i % 4 == 0;
        return;
    }
    const double k[3] = {r[0] / th, r[1] / th, r[2] / th}, c = std::cos(th), s = std::sin(th), c1 = 1 - c;
    const double K[9] = {0, -k[2], k[1], k[2], 0, -k[0], -k[1], k[0], 0};
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) R[3 * i + j] = c * (i == j) + c1 * k[i] * k[j] + s * K[3 * i + j];
}

}  // namespace

extern "C" int alva_find_plane(alva_ctx *ctx, const double *d_points, int n, const double *h_pose7_twc, int num_iterations, int do_random,
                               uint32_t seed, const int *h_samples3, float *h_plane_pose16, int *h_found) {
    ALVA_ARG(ctx && h_pose7_twc && h_plane_pose16 && h_found && num_iterations > 0 && n >= 0);
    *h_found = 0;
    if (n < 32) return ALVA_OK;  // system.cpp:181
    ALVA_ARG(d_points && n <= 12288);  // distances of one hypothesis live in LDS
    // pinned: samples | finish record ; scratch: scores | planes
    const size_t off_fin = ((size_t) num_iterations * 12 + 255) / 256 * 256;
    uint8_t *pin = nullptr;
    int rc = alva_ctx_pinned(ctx, off_fin + sizeof(PlaneFinish), (void **) &pin);
    if (rc) return rc;
    int *smp = (int *) pin;
    if (h_samples3) memcpy(smp, h_samples3, (size_t) num_iterations * 12);
    else {
        // the reference draws with std::sample from a generator seeded by std::random_device in every iteration (:203); any
        // three distinct indices are as faithful as any other
        std::mt19937 gen(do_random ? (uint32_t) (time(nullptr) ^ clock()) : seed);
        std::uniform_int_distribution<int> pick(0, n - 1);
        for (int it = 0; it < num_iterations; it++) {
            int s[3];
            s[0] = pick(gen);
            do s[1] = pick(gen); while (s[1] == s[0]);
            do s[2] = pick(gen); while (s[2] == s[0] || s[2] == s[1]);
            std::sort(s, s + 3);
            memcpy(smp + 3 * it, s, sizeof(s));
        }
    }
    uint8_t *scr = nullptr;
    rc = alva_ctx_scratch(ctx, 9, (size_t) num_iterations * 20 + 256, (void **) &scr);
    if (rc) return rc;
    PlaneArgs A{};
    A.pts = d_points;
    A.samples = smp;
    A.n = n;
    A.iters = num_iterations;
    A.kth = std::max((int) (0.2 * n), 20);
    A.sinTh = sinf(5.0f * 3.14159265358979323846f / 180.0f);
    A.scores = (float *) scr;
    A.planes = (float *) (scr + ((size_t) num_iterations * 4 + 15) / 16 * 16);
    hipLaunchKernelGGL(k_plane_hyp, dim3(num_iterations), dim3(PL_NT), (size_t) n * sizeof(float), ctx->stream, A);
    hipLaunchKernelGGL(k_plane_finish, dim3(1), dim3(PL_NT), 0, ctx->stream, A, (PlaneFinish *) (pin + off_fin));
    ALVA_LAUNCH_CHECK();
    ALVA_HIP(alva_stream_sync(ctx->stream));
    PlaneFinish fin;
    memcpy(&fin, pin + off_fin, sizeof(fin));
    if (fin.n_inliers < 32) return ALVA_OK;  // :261-269
    double M[16], v[4];
    {
        int t = 0;
        for (int x = 0; x < 4; x++)
            for (int y = x; y < 4; y++) M[4 * x + y] = M[4 * y + x] = fin.M[t++];
    }
    smallest_eigvec4(M, v);
    float a = (float) v[0], b = (float) v[1], c = (float) v[2];
    float origin[3];
    for (int k = 0; k < 3; k++) origin[k] = (float) fin.sum[k] * (1.0f / (float) fin.n_inliers);
    const float f = 1.0f / std::sqrt(a * a + b * b + c * c);
    const double *p7 = h_pose7_twc;
    const double qx = p7[3], qy = p7[4], qz = p7[5], qw = p7[6];
    const double R[9] = {1 - 2 * (qy * qy + qz * qz), 2 * (qx * qy - qz * qw), 2 * (qx * qz + qy * qw),
                         2 * (qx * qy + qz * qw), 1 - 2 * (qx * qx + qz * qz), 2 * (qy * qz - qx * qw),
                         2 * (qx * qz - qy * qw), 2 * (qy * qz + qx * qw), 1 - 2 * (qx * qx + qy * qy)};
    float Oc[3];  // -camPose.R^T * camPose.t with camPose = [R^T | t] (utils.cpp:54-79)
    for (int i = 0; i < 3; i++) Oc[i] = -((float) R[3 * i] * (float) p7[0] + (float) R[3 * i + 1] * (float) p7[1] + (float) R[3 * i + 2] * (float) p7[2]);
    const float mx[3] = {Oc[0] - origin[0], Oc[1] - origin[1], Oc[2] - origin[2]};
    if (mx[0] * a + mx[1] * b + mx[2] * c > 0) {
        a = -a;
        b = -b;
        c = -c;
    }
    const float nx = a * f, ny = b * f, nz = c * f;
    const float v3[3] = {0.f, -nz, ny};  // (1,0,0) x n
    const float sa = (float) std::sqrt((double) v3[0] * v3[0] + (double) v3[1] * v3[1] + (double) v3[2] * v3[2]), ca = nx;
    const float ang = std::atan2(sa, ca);
    const double r1[3] = {v3[0] * ang / sa, v3[1] * ang / sa, v3[2] * ang / sa}, r2[3] = {1, 0, 0};
    double R1[9], R2[9];
    rodrigues(r1, R1);
    rodrigues(r2, R2);
    for (int col = 0; col < 3; col++) {
        for (int row = 0; row < 3; row++) {
            float s = 0.f;
            for (int k = 0; k < 3; k++) s += (float) R1[3 * row + k] * (float) R2[3 * k + col];
            h_plane_pose16[4 * col + row] = s;  // Utils::toPoseArray(cv::Mat): columns of the rotation (utils.cpp:29-52)
        }
        h_plane_pose16[4 * col + 3] = 0.f;
    }
    for (int k = 0; k < 3; k++) h_plane_pose16[12 + k] = origin[k];
    h_plane_pose16[15] = 1.f;
    *h_found = 1;
    return ALVA_OK;
}
