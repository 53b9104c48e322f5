// f4b: the lens-distortion paths of CameraCalibration, bit-exact.
//
//   alva_undistort_points   CameraCalibration::undistortImagePoint (src/slam/src/camera_calibration.cpp:56-72) =
//                           cv::undistortPoints(pts, out, K, D, R = K) -- the reference passes the camera matrix in the
//                           rectification slot -- with D = (k1, k2, p1, p2) and the default 5 fixed iterations
//                           (calib3d/src/undistort.dispatch.cpp:384-556).  Every keypoint goes through it
//                           (frame / map-manager keypoint construction), also when the coefficients are zero.
//   alva_project_dist       CameraCalibration::projectCamToImageDist (:34-54) = cv::projectPoints of (x/z, y/z, 1) rounded
//                           to FLOAT (cv::Point3f), zero rvec / tvec (calib3d/src/calibration.cpp:522-)
// One thread per point, IEEE double in the reference's operation order (compile with -ffp-contract=off), float results.
#include "common.hpp"

namespace {

struct Cam {
    double fx, fy, cx, cy, k1, k2, p1, p2;
};

__global__ void __launch_bounds__(256) k_undistort(Cam C, const float *__restrict__ px, int n, float *__restrict__ out) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const double ifx = 1. / C.fx, ify = 1. / C.fy;
    const double u = px[2 * (size_t) i], v = px[2 * (size_t) i + 1];
    double x = (u - C.cx) * ifx, y = (v - C.cy) * ify;
    const double x0 = x, y0 = y;
    for (int j = 0; j < 5; j++) {
        const double r2 = x * x + y * y;
        const double icdist = (1 + ((0 * r2 + 0) * r2 + 0) * r2) / (1 + ((0 * r2 + C.k2) * r2 + C.k1) * r2);
        if (icdist < 0) {
            x = (u - C.cx) * ifx;
            y = (v - C.cy) * ify;
            break;
        }
        const double deltaX = 2 * C.p1 * x * y + C.p2 * (r2 + 2 * x * x) + 0 * r2 + 0 * r2 * r2;
        const double deltaY = C.p1 * (r2 + 2 * y * y) + 2 * C.p2 * x * y + 0 * r2 + 0 * r2 * r2;
        x = (x0 - deltaX) * icdist;
        y = (y0 - deltaY) * icdist;
    }
    const double xx = C.fx * x + 0 * y + C.cx, yy = 0 * x + C.fy * y + C.cy, ww = 1. / (0 * x + 0 * y + 1);
    out[2 * (size_t) i] = (float) (xx * ww);
    out[2 * (size_t) i + 1] = (float) (yy * ww);
}

__global__ void __launch_bounds__(256) k_project_dist(Cam C, const double *__restrict__ P, int n, float *__restrict__ out) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const double iz = 1. / P[3 * (size_t) i + 2];
    const float Xf = (float) (P[3 * (size_t) i] * iz), Yf = (float) (P[3 * (size_t) i + 1] * iz);
    const double x = (double) Xf, y = (double) Yf;
    const double r2 = x * x + y * y, r4 = r2 * r2, r6 = r4 * r2;
    const double a1 = 2 * x * y, a2 = r2 + 2 * x * x, a3 = r2 + 2 * y * y;
    const double cdist = 1 + C.k1 * r2 + C.k2 * r4 + 0 * r6;
    const double icdist2 = 1. / (1 + 0 * r2 + 0 * r4 + 0 * r6);
    const double xd0 = x * cdist * icdist2 + C.p1 * a1 + C.p2 * a2 + 0 * r2 + 0 * r4;
    const double yd0 = y * cdist * icdist2 + C.p1 * a3 + C.p2 * a1 + 0 * r2 + 0 * r4;
    out[2 * (size_t) i] = (float) (xd0 * C.fx + C.cx);
    out[2 * (size_t) i + 1] = (float) (yd0 * C.fy + C.cy);
}

}  // namespace

extern "C" int alva_undistort_points(alva_ctx *ctx, const float *d_px, int n, double fx, double fy, double cx, double cy, double k1, double k2,
                                     double p1, double p2, float *d_out) {
    ALVA_ARG(ctx && n >= 0);
    if (n == 0) return ALVA_OK;
    ALVA_ARG(d_px && d_out);
    const Cam C{fx, fy, cx, cy, k1, k2, p1, p2};
    hipLaunchKernelGGL(k_undistort, dim3(alva_divup(n, 256)), dim3(256), 0, ctx->stream, C, d_px, n, d_out);
    ALVA_LAUNCH_CHECK();
    return ALVA_OK;
}

extern "C" int alva_project_dist(alva_ctx *ctx, const double *d_cam_pts, int n, double fx, double fy, double cx, double cy, double k1,
                                 double k2, double p1, double p2, float *d_out) {
    ALVA_ARG(ctx && n >= 0);
    if (n == 0) return ALVA_OK;
    ALVA_ARG(d_cam_pts && d_out);
    const Cam C{fx, fy, cx, cy, k1, k2, p1, p2};
    hipLaunchKernelGGL(k_project_dist, dim3(alva_divup(n, 256)), dim3(256), 0, ctx->stream, C, d_cam_pts, n, d_out);
    ALVA_LAUNCH_CHECK();
    return ALVA_OK;
}
