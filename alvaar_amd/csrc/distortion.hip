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
#include "camera_device.hpp"

namespace {

typedef AlvaCam Cam;

__global__ void __launch_bounds__(256) k_undistort(Cam C, const float *__restrict__ px, int n, float *__restrict__ out) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float u, v;
    alva_undistort_dev(C, px[2 * (size_t) i], px[2 * (size_t) i + 1], u, v);
    out[2 * (size_t) i] = u;
    out[2 * (size_t) i + 1] = v;
}

__global__ void __launch_bounds__(256) k_project_dist(Cam C, const double *__restrict__ P, int n, float *__restrict__ out) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float u, v;
    alva_project_dist_dev(C, P[3 * (size_t) i], P[3 * (size_t) i + 1], P[3 * (size_t) i + 2], u, v);
    out[2 * (size_t) i] = u;
    out[2 * (size_t) i + 1] = v;
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
