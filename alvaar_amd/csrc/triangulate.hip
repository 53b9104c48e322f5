// f2a: triangulation of a new keyframe's 2-D keypoints against their first observation.
//
// Replaces the per-keypoint arithmetic of Mapper::triangulateTemporal (src/slam/src/mapper.cpp:222-287):
//   rotation-compensated parallax   cv::norm(unpx_l - projCamToImage(R_lr bv_r))                          (:246-248)
//   lPoint                          MultiViewGeometry::triangulate = opengv triangulate2 (mid-point, closed 2x2 inverse;
//                                   src/libs/opengv/src/triangulation/methods.cpp:67-90)                     (:253)
//   gates                           z < 0.1 in either camera (:256); reprojection error > mapMaxReprojectionError_ in
//                                   either image (:266-272), projections rounded to float as cv::Point2f
//                                   (camera_calibration.cpp:25-32)
//   world point, inverse depth      keyframe->projCamToWorld(lPoint), 1 / lPoint.z                           (:283-284)
// The map bookkeeping around it (which keyframe observed the point first, removeMapPointObs, updateMapPoint) stays on
// the host: the caller groups the points by first-observing keyframe and passes one transform block per group.
// One thread per keypoint, FP64; a few hundred points per keyframe, so this is one short launch.
#include "common.hpp"

namespace {

__device__ __forceinline__ void matvec(const double *R, const double *v, double *o) {
#pragma unroll
    for (int i = 0; i < 3; i++) o[i] = (R[3 * i] * v[0] + R[3 * i + 1] * v[1]) + R[3 * i + 2] * v[2];
}
__device__ __forceinline__ double dot3(const double *a, const double *b) { return (a[0] * b[0] + a[1] * b[1]) + a[2] * b[2]; }
__device__ __forceinline__ void project(const double *p, double fx, double fy, double cx, double cy, float &u, float &v) {
    const double iz = 1. / p[2], x = p[0] * iz, y = p[1] * iz;
    u = (float) (fx * x + cx);
    v = (float) (fy * y + cy);
}
__device__ __forceinline__ double norm2f(float dx, float dy) { return sqrt((double) dx * (double) dx + (double) dy * (double) dy); }

__global__ void __launch_bounds__(256) k_triangulate(int n, const double *__restrict__ T, const int *__restrict__ group,
                                                     const double *__restrict__ bvl, const double *__restrict__ bvr,
                                                     const float *__restrict__ unpxl, const float *__restrict__ unpxr, double fx, double fy,
                                                     double cx, double cy, float maxErr, double *__restrict__ lpt, double *__restrict__ wpt,
                                                     double *__restrict__ invDepth, uint8_t *__restrict__ status,
                                                     double *__restrict__ parallax) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const double *G = T + 36 * (size_t) group[i];
    double Rlr[9], tlr[3], f1[3], f2[3];
#pragma unroll
    for (int k = 0; k < 9; k++) Rlr[k] = G[k];
#pragma unroll
    for (int k = 0; k < 3; k++) {
        tlr[k] = G[9 + k];
        f1[k] = bvl[3 * (size_t) i + k];
        f2[k] = bvr[3 * (size_t) i + k];
    }
    const float ulx = unpxl[2 * (size_t) i], uly = unpxl[2 * (size_t) i + 1], urx = unpxr[2 * (size_t) i], ury = unpxr[2 * (size_t) i + 1];
    double f2u[3];
    matvec(Rlr, f2, f2u);
    float ru, rv;
    project(f2u, fx, fy, cx, cy, ru, rv);
    parallax[i] = norm2f(ulx - ru, uly - rv);
    const double b0 = dot3(tlr, f1), b1 = dot3(tlr, f2u);
    const double a00 = dot3(f1, f1), a10 = dot3(f1, f2u), a01 = -a10, a11 = -dot3(f2u, f2u);
    const double invdet = 1.0 / (a00 * a11 - a10 * a01);
    const double i00 = a11 * invdet, i10 = -a10 * invdet, i01 = -a01 * invdet, i11 = a00 * invdet;
    const double l0 = i00 * b0 + i01 * b1, l1 = i10 * b0 + i11 * b1;
    double lp[3], rp[3], wp[3], t[3];
#pragma unroll
    for (int k = 0; k < 3; k++) lp[k] = (l0 * f1[k] + (tlr[k] + l1 * f2u[k])) / 2;
    matvec(G + 12, lp, t);
#pragma unroll
    for (int k = 0; k < 3; k++) rp[k] = t[k] + G[21 + k];
    matvec(G + 24, lp, t);
#pragma unroll
    for (int k = 0; k < 3; k++) wp[k] = t[k] + G[33 + k];
#pragma unroll
    for (int k = 0; k < 3; k++) {
        lpt[3 * (size_t) i + k] = lp[k];
        wpt[3 * (size_t) i + k] = wp[k];
    }
    invDepth[i] = 1. / lp[2];
    uint8_t st = 0;
    if (lp[2] < 0.1 || rp[2] < 0.1) st = 1;
    else {
        float lu, lv, pu, pv;
        project(lp, fx, fy, cx, cy, lu, lv);
        project(rp, fx, fy, cx, cy, pu, pv);
        const float lDist = (float) norm2f(lu - ulx, lv - uly), rDist = (float) norm2f(pu - urx, pv - ury);
        if (lDist > maxErr || rDist > maxErr) st = 2;
    }
    status[i] = st;
}

}  // namespace

extern "C" int alva_triangulate(alva_ctx *ctx, int n, const double *d_T, int n_groups, const int *d_group, const double *d_bv_l,
                                const double *d_bv_r, const float *d_unpx_l, const float *d_unpx_r, double fx, double fy, double cx,
                                double cy, float max_reproj_err, double *d_lpt, double *d_wpt, double *d_inv_depth, uint8_t *d_status,
                                double *d_parallax) {
    ALVA_ARG(ctx && n >= 0 && n_groups >= 0);
    if (n == 0) return ALVA_OK;
    ALVA_ARG(n_groups > 0 && d_T && d_group && d_bv_l && d_bv_r && d_unpx_l && d_unpx_r && d_lpt && d_wpt && d_inv_depth && d_status && d_parallax);
    hipLaunchKernelGGL(k_triangulate, dim3(alva_divup(n, 256)), dim3(256), 0, ctx->stream, n, d_T, d_group, d_bv_l, d_bv_r, d_unpx_l, d_unpx_r,
                       fx, fy, cx, cy, max_reproj_err, d_lpt, d_wpt, d_inv_depth, d_status, d_parallax);
    ALVA_LAUNCH_CHECK();
    return ALVA_OK;
}
