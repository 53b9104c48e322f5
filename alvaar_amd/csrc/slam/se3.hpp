// Host-side rigid-body algebra for the map layer (poses live as unit quaternion + translation, like the reference's
// Sophus::SE3d members Frame::Twc_ / Tcw_, src/slam/src/frame.hpp:163-164).  Plain C++, no HIP, no third-party headers.
// Conventions: q = (x, y, z, w); pose7 = [tx ty tz qx qy qz qw] (src/slam/src/ceres_parametrization.hpp:64-71);
// tangent = (upsilon, omega) as Sophus (se3.hpp:763-784).
#pragma once
#include <cmath>

namespace alva_slam {

struct SE3 {
    double q[4] = {0, 0, 0, 1};
    double t[3] = {0, 0, 0};
};

inline void quat_mul(const double *a, const double *b, double *o) {
    o[0] = a[3] * b[0] + a[0] * b[3] + a[1] * b[2] - a[2] * b[1];
    o[1] = a[3] * b[1] + a[1] * b[3] + a[2] * b[0] - a[0] * b[2];
    o[2] = a[3] * b[2] + a[2] * b[3] + a[0] * b[1] - a[1] * b[0];
    o[3] = a[3] * b[3] - a[0] * b[0] - a[1] * b[1] - a[2] * b[2];
}

inline void quat_normalize(double *q) {
    const double n = std::sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    for (int i = 0; i < 4; i++) q[i] /= n;
}

// R row-major
inline void quat_to_rot(const double *q, double *R) {
    const double x = q[0], y = q[1], z = q[2], w = q[3];
    const double tx = 2 * x, ty = 2 * y, tz = 2 * z;
    const double twx = tx * w, twy = ty * w, twz = tz * w, txx = tx * x, txy = ty * x, txz = tz * x, tyy = ty * y, tyz = tz * y, tzz = tz * z;
    R[0] = 1 - (tyy + tzz); R[1] = txy - twz; R[2] = txz + twy;
    R[3] = txy + twz; R[4] = 1 - (txx + tzz); R[5] = tyz - twx;
    R[6] = txz - twy; R[7] = tyz + twx; R[8] = 1 - (txx + tyy);
}

// rotation matrix -> unit quaternion (largest-pivot form), then normalised as SO3::setQuaternion does (so3.hpp:409-412)
inline void rot_to_quat(const double *R, double *q) {
    double tr = R[0] + R[4] + R[8];
    if (tr > 0) {
        double s = std::sqrt(tr + 1.0);
        q[3] = 0.5 * s;
        s = 0.5 / s;
        q[0] = (R[7] - R[5]) * s;
        q[1] = (R[2] - R[6]) * s;
        q[2] = (R[3] - R[1]) * s;
    } else {
        int i = 0;
        if (R[4] > R[0]) i = 1;
        if (R[8] > R[4 * i]) i = 2;
        const int j = (i + 1) % 3, k = (j + 1) % 3;
        double s = std::sqrt(R[4 * i] - R[4 * j] - R[4 * k] + 1.0);
        q[i] = 0.5 * s;
        s = 0.5 / s;
        q[3] = (R[3 * k + j] - R[3 * j + k]) * s;
        q[j] = (R[3 * j + i] + R[3 * i + j]) * s;
        q[k] = (R[3 * k + i] + R[3 * i + k]) * s;
    }
    quat_normalize(q);
}

// v' = q v q^-1 through the doubled cross product (no matrix)
inline void quat_rotate(const double *q, const double *v, double *o) {
    const double ux = 2 * (q[1] * v[2] - q[2] * v[1]), uy = 2 * (q[2] * v[0] - q[0] * v[2]), uz = 2 * (q[0] * v[1] - q[1] * v[0]);
    const double r0 = v[0] + q[3] * ux + (q[1] * uz - q[2] * uy);
    const double r1 = v[1] + q[3] * uy + (q[2] * ux - q[0] * uz);
    const double r2 = v[2] + q[3] * uz + (q[0] * uy - q[1] * ux);
    o[0] = r0; o[1] = r1; o[2] = r2;
}

inline SE3 se3_inverse(const SE3 &T) {
    SE3 o;
    o.q[0] = -T.q[0]; o.q[1] = -T.q[1]; o.q[2] = -T.q[2]; o.q[3] = T.q[3];
    double r[3];
    quat_rotate(o.q, T.t, r);
    o.t[0] = -r[0]; o.t[1] = -r[1]; o.t[2] = -r[2];
    return o;
}

// group product; the quaternion is re-scaled when its squared norm drifted from 1 (first-order, as so3.hpp:329-343)
inline SE3 se3_mul(const SE3 &A, const SE3 &B) {
    SE3 o;
    quat_mul(A.q, B.q, o.q);
    const double n2 = o.q[0] * o.q[0] + o.q[1] * o.q[1] + o.q[2] * o.q[2] + o.q[3] * o.q[3];
    if (n2 != 1.0) {
        const double s = 2.0 / (1.0 + n2);
        for (double &c: o.q) c *= s;
    }
    double r[3];
    quat_rotate(A.q, B.t, r);
    for (int i = 0; i < 3; i++) o.t[i] = A.t[i] + r[i];
    return o;
}

inline void se3_apply(const SE3 &T, const double *p, double *o) {
    double r[3];
    quat_rotate(T.q, p, r);
    for (int i = 0; i < 3; i++) o[i] = r[i] + T.t[i];
}

inline void mat3_mul(const double *A, const double *B, double *C) {
    for (int r = 0; r < 3; r++)
        for (int c = 0; c < 3; c++) C[3 * r + c] = A[3 * r] * B[c] + A[3 * r + 1] * B[3 + c] + A[3 * r + 2] * B[6 + c];
}
inline void mat3_vec(const double *A, const double *v, double *o) {
    const double a = A[0] * v[0] + A[1] * v[1] + A[2] * v[2], b = A[3] * v[0] + A[4] * v[1] + A[5] * v[2], c = A[6] * v[0] + A[7] * v[1] + A[8] * v[2];
    o[0] = a; o[1] = b; o[2] = c;
}

// SO(3) logarithm with the atan form (so3.hpp:247-288); returns theta
inline double so3_log(const double *q, double *w) {
    const double n2 = q[0] * q[0] + q[1] * q[1] + q[2] * q[2], qw = q[3];
    double k, theta;
    if (n2 < 1e-10 * 1e-10) {
        k = 2.0 / qw - (2.0 / 3.0) * n2 / (qw * qw * qw);
        theta = 2.0 * n2 / qw;
    } else {
        const double n = std::sqrt(n2);
        if (std::fabs(qw) < 1e-10) k = (qw > 0 ? M_PI : -M_PI) / n;
        else k = 2.0 * std::atan(n / qw) / n;
        theta = k * n;
    }
    for (int i = 0; i < 3; i++) w[i] = k * q[i];
    return theta;
}

// SE(3) logarithm (se3.hpp:223-256): xi = (V^-1 t, omega)
inline void se3_log(const SE3 &T, double *xi) {
    double w[3];
    const double theta = so3_log(T.q, w);
    const double O[9] = {0, -w[2], w[1], w[2], 0, -w[0], -w[1], w[0], 0};
    double O2[9], Vi[9];
    mat3_mul(O, O, O2);
    double c;
    if (std::fabs(theta) < 1e-10) c = 1.0 / 12.0;
    else {
        const double h = 0.5 * theta;
        c = (1.0 - theta * std::cos(h) / (2.0 * std::sin(h))) / (theta * theta);
    }
    for (int i = 0; i < 9; i++) Vi[i] = ((i % 4 == 0) ? 1.0 : 0.0) - 0.5 * O[i] + c * O2[i];
    mat3_vec(Vi, T.t, xi);
    xi[3] = w[0]; xi[4] = w[1]; xi[5] = w[2];
}

// SE(3) exponential (se3.hpp:763-784, so3.hpp:585-621)
inline SE3 se3_exp(const double *xi) {
    const double *u = xi, *w = xi + 3;
    const double th2 = w[0] * w[0] + w[1] * w[1] + w[2] * w[2];
    double theta, imag, real;
    if (th2 < 1e-10 * 1e-10) {
        theta = 0;
        const double th4 = th2 * th2;
        imag = 0.5 - (1.0 / 48.0) * th2 + (1.0 / 3840.0) * th4;
        real = 1.0 - (1.0 / 8.0) * th2 + (1.0 / 384.0) * th4;
    } else {
        theta = std::sqrt(th2);
        const double h = 0.5 * theta;
        imag = std::sin(h) / theta;
        real = std::cos(h);
    }
    SE3 o;
    o.q[0] = imag * w[0]; o.q[1] = imag * w[1]; o.q[2] = imag * w[2]; o.q[3] = real;
    double V[9];
    if (theta < 1e-10) {
        quat_to_rot(o.q, V);
    } else {
        const double O[9] = {0, -w[2], w[1], w[2], 0, -w[0], -w[1], w[0], 0};
        double O2[9];
        mat3_mul(O, O, O2);
        const double a = (1.0 - std::cos(theta)) / th2, b = (theta - std::sin(theta)) / (th2 * theta);
        for (int i = 0; i < 9; i++) V[i] = ((i % 4 == 0) ? 1.0 : 0.0) + a * O[i] + b * O2[i];
    }
    mat3_vec(V, u, o.t);
    return o;
}

inline void se3_to_pose7(const SE3 &T, double *p) {
    p[0] = T.t[0]; p[1] = T.t[1]; p[2] = T.t[2];
    p[3] = T.q[0]; p[4] = T.q[1]; p[5] = T.q[2]; p[6] = T.q[3];
}
inline SE3 se3_from_pose7(const double *p) {
    SE3 T;
    T.t[0] = p[0]; T.t[1] = p[1]; T.t[2] = p[2];
    T.q[0] = p[3]; T.q[1] = p[4]; T.q[2] = p[5]; T.q[3] = p[6];
    quat_normalize(T.q);
    return T;
}

}  // namespace alva_slam
