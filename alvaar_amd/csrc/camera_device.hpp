// Device-side camera model shared by distortion.hip (stage entry points) and the fused tracking step (stages_hip.hip):
// CameraCalibration::undistortImagePoint / projectCamToImageDist (src/slam/src/camera_calibration.cpp:34-72) and the bearing of
// Frame::computeKeypoint (frame.cpp:105-113).  IEEE double in the reference's operation order (compile with -ffp-contract=off).
#pragma once
#include <hip/hip_runtime.h>

struct AlvaCam {
    double fx, fy, cx, cy, k1, k2, p1, p2;
};

// cv::undistortPoints(pts, out, K, D, R = K), 5 fixed iterations (calib3d/src/undistort.dispatch.cpp:384-556)
__device__ __forceinline__ void alva_undistort_dev(const AlvaCam &C, float pu, float pv, float &ou, float &ov) {
    const double ifx = 1. / C.fx, ify = 1. / C.fy;
    const double u = pu, v = pv;
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
    ou = (float) (xx * ww);
    ov = (float) (yy * ww);
}

// cv::projectPoints of (x/z, y/z, 1) rounded to float, zero rvec / tvec (calib3d/src/calibration.cpp:522-)
__device__ __forceinline__ void alva_project_dist_dev(const AlvaCam &C, double X, double Y, double Z, float &ou, float &ov) {
    const double iz = 1. / Z;
    const float Xf = (float) (X * iz), Yf = (float) (Y * iz);
    const double x = (double) Xf, y = (double) Yf;
    const double r2 = x * x + y * y, r4 = r2 * r2, r6 = r4 * r2;
    const double a1 = 2 * x * y, a2 = r2 + 2 * x * x, a3 = r2 + 2 * y * y;
    const double cdist = 1 + C.k1 * r2 + C.k2 * r4 + 0 * r6;
    const double icdist2 = 1. / (1 + 0 * r2 + 0 * r4 + 0 * r6);
    const double xd0 = x * cdist * icdist2 + C.p1 * a1 + C.p2 * a2 + 0 * r2 + 0 * r4;
    const double yd0 = y * cdist * icdist2 + C.p1 * a3 + C.p2 * a1 + 0 * r2 + 0 * r4;
    ou = (float) (xd0 * C.fx + C.cx);
    ov = (float) (yd0 * C.fy + C.cy);
}

// bv = normalised K^-1 (unpx, 1), Eigen's operation order (frame.cpp:109-112); invK row-major
__device__ __forceinline__ void alva_bearing_dev(const double *invK, float uu, float vv, double *bv) {
    const double u = (double) uu, v = (double) vv;
    const double b0 = (invK[0] * u + invK[1] * v) + invK[2] * 1.;
    const double b1 = (invK[3] * u + invK[4] * v) + invK[5] * 1.;
    const double b2 = (invK[6] * u + invK[7] * v) + invK[8] * 1.;
    const double z = (b0 * b0 + b1 * b1) + b2 * b2;
    if (z > 0.) {
        const double s = sqrt(z);
        bv[0] = b0 / s; bv[1] = b1 / s; bv[2] = b2 / s;
    } else {
        bv[0] = b0; bv[1] = b1; bv[2] = b2;
    }
}

// Sophus SE3 * point = Eigen quaternion rotation + translation (q = x y z w)
__device__ __forceinline__ void alva_se3_apply_dev(const double *q, const double *t, const double *v, double *o) {
    const double uv0 = q[1] * v[2] - q[2] * v[1], uv1 = q[2] * v[0] - q[0] * v[2], uv2 = q[0] * v[1] - q[1] * v[0];
    const double u0 = uv0 + uv0, u1 = uv1 + uv1, u2 = uv2 + uv2;
    const double c0 = q[1] * u2 - q[2] * u1, c1 = q[2] * u0 - q[0] * u2, c2 = q[0] * u1 - q[1] * u0;
    o[0] = ((v[0] + q[3] * u0) + c0) + t[0];
    o[1] = ((v[1] + q[3] * u1) + c1) + t[1];
    o[2] = ((v[2] + q[3] * u2) + c2) + t[2];
}
