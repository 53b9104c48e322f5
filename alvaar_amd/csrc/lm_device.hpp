// Device/host helpers shared by the PnP (a9) and bundle-adjustment (a10-a13) kernels: SE(3) in the
// reference's conventions, the pinhole reprojection residual with its J_pi * R_cw block, Huber.
//
// Conventions restated from the reference:
//   pose storage  [tx,ty,tz, qx,qy,qz,qw] = Twc (cam -> world)          src/slam/src/ceres_parametrization.hpp:64-71
//   update        T <- Exp(delta) * T, delta = (upsilon, omega)          ceres_parametrization.hpp:224-240
//   Exp           Sophus se3.hpp:763-784, so3.hpp:585-621; product normalises the quaternion (so3.hpp:329-343)
//   residual      r = K * (R_wc^T (X - t_wc)) / z - uv                   ceres_parametrization.cpp:96-155
//   Huber         rho0 = 2 a sqrt(s) - a^2, rho' = a / sqrt(s) for s > a^2; r and J scaled by sqrt(rho')
//                 (ceres loss_function.cc:48-62, corrector.cc:41-110)
#pragma once
#include <hip/hip_runtime.h>

#define ALVA_HD __host__ __device__ __forceinline__

struct Se3 {
    double q[4];  // x,y,z,w (unit)
    double t[3];
    double R[9];  // row-major R_wc
};

ALVA_HD void quat_to_R(const double q[4], double R[9]) {
    const double x = q[0], y = q[1], z = q[2], w = q[3];
    const double tx = 2 * x, ty = 2 * y, tz = 2 * z, twx = tx * w, twy = ty * w, twz = tz * w, txx = tx * x, txy = ty * x, txz = tz * x,
                 tyy = ty * y, tyz = tz * y, tzz = tz * z;
    R[0] = 1 - (tyy + tzz); R[1] = txy - twz; R[2] = txz + twy;
    R[3] = txy + twz; R[4] = 1 - (txx + tzz); R[5] = tyz - twx;
    R[6] = txz - twy; R[7] = tyz + twx; R[8] = 1 - (txx + tyy);
}

ALVA_HD void se3_from_pose7(const double *p, Se3 &T) {
    const double n = sqrt(p[3] * p[3] + p[4] * p[4] + p[5] * p[5] + p[6] * p[6]);
    for (int i = 0; i < 4; i++) T.q[i] = p[3 + i] / n;
    for (int i = 0; i < 3; i++) T.t[i] = p[i];
    quat_to_R(T.q, T.R);
}

ALVA_HD void se3_plus(const double *x7, const double *d6, double *out7) {
    Se3 T;
    se3_from_pose7(x7, T);
    const double *u = d6, *w = d6 + 3;
    const double th2 = w[0] * w[0] + w[1] * w[1] + w[2] * w[2];
    double theta, imag, real;
    if (th2 < 1e-10 * 1e-10) {
        theta = 0;
        const double th4 = th2 * th2;
        imag = 0.5 - (1.0 / 48.0) * th2 + (1.0 / 3840.0) * th4;
        real = 1 - (1.0 / 8.0) * th2 + (1.0 / 384.0) * th4;
    } else {
        theta = sqrt(th2);
        const double h = 0.5 * theta;
        imag = sin(h) / theta;
        real = cos(h);
    }
    const double qd[4] = {imag * w[0], imag * w[1], imag * w[2], real};
    double Rd[9], V[9];
    quat_to_R(qd, Rd);
    if (theta < 1e-10) {
        for (int i = 0; i < 9; i++) V[i] = Rd[i];
    } else {
        const double O[9] = {0, -w[2], w[1], w[2], 0, -w[0], -w[1], w[0], 0};
        const double a = (1 - cos(theta)) / th2, b = (theta - sin(theta)) / (th2 * theta);
        for (int r = 0; r < 3; r++)
            for (int c = 0; c < 3; c++) {
                const double o2 = O[3 * r] * O[c] + O[3 * r + 1] * O[3 + c] + O[3 * r + 2] * O[6 + c];
                V[3 * r + c] = (r == c ? 1.0 : 0.0) + a * O[3 * r + c] + b * o2;
            }
    }
    const double *a4 = qd, *b4 = T.q;
    const double qn[4] = {a4[3] * b4[0] + a4[0] * b4[3] + a4[1] * b4[2] - a4[2] * b4[1],
                          a4[3] * b4[1] + a4[1] * b4[3] + a4[2] * b4[0] - a4[0] * b4[2],
                          a4[3] * b4[2] + a4[2] * b4[3] + a4[0] * b4[1] - a4[1] * b4[0],
                          a4[3] * b4[3] - a4[0] * b4[0] - a4[1] * b4[1] - a4[2] * b4[2]};
    const double n = sqrt(qn[0] * qn[0] + qn[1] * qn[1] + qn[2] * qn[2] + qn[3] * qn[3]);
    for (int i = 0; i < 3; i++)
        out7[i] = (V[3 * i] * u[0] + V[3 * i + 1] * u[1] + V[3 * i + 2] * u[2]) + (Rd[3 * i] * T.t[0] + Rd[3 * i + 1] * T.t[1] + Rd[3 * i + 2] * T.t[2]);
    for (int i = 0; i < 4; i++) out7[3 + i] = qn[i] / n;
}

ALVA_HD void huber_rho(double s, double a, int robust, double &rho0, double &rho1) {
    const double b = a * a;
    if (robust && s > b) {
        const double r = sqrt(s);
        rho0 = 2.0 * a * r - b;
        const double v = a / r;
        rho1 = v > 2.2250738585072014e-308 ? v : 2.2250738585072014e-308;
    } else {
        rho0 = s;
        rho1 = 1.0;
    }
}

// r (2), JR = J_pi * R_cw (2x3 row-major, only if WANT_J), chi2 = |r|^2, depth flag
template<bool WANT_J>
ALVA_HD void reproj(const Se3 &Twc, const double K[4], const double X[3], double u, double v, double r[2], double JR[6], double &chi2,
                    int &depth_pos) {
    const double d0 = X[0] - Twc.t[0], d1 = X[1] - Twc.t[1], d2 = X[2] - Twc.t[2];
    const double c0 = Twc.R[0] * d0 + Twc.R[3] * d1 + Twc.R[6] * d2;
    const double c1 = Twc.R[1] * d0 + Twc.R[4] * d1 + Twc.R[7] * d2;
    const double c2 = Twc.R[2] * d0 + Twc.R[5] * d1 + Twc.R[8] * d2;
    const double iz = 1.0 / c2;
    r[0] = K[0] * c0 * iz + K[2] - u;
    r[1] = K[1] * c1 * iz + K[3] - v;
    chi2 = r[0] * r[0] + r[1] * r[1];
    depth_pos = c2 > 0;
    if (WANT_J) {
        const double iz2 = iz * iz;
        const double Jp[6] = {iz * K[0], 0, -c0 * iz2 * K[0], 0, iz * K[1], -c1 * iz2 * K[1]};
        for (int rr = 0; rr < 2; rr++)
            for (int cc = 0; cc < 3; cc++)  // R_cw[k][cc] = R_wc[cc][k]
                JR[3 * rr + cc] = Jp[3 * rr] * Twc.R[3 * cc] + Jp[3 * rr + 1] * Twc.R[3 * cc + 1] + Jp[3 * rr + 2] * Twc.R[3 * cc + 2];
    }
}

// out (2x3) = JR (2x3) * hat(X)
ALVA_HD void times_hat(const double JR[6], const double X[3], double out[6]) {
    for (int r = 0; r < 2; r++) {
        const double *j = JR + 3 * r;
        out[3 * r + 0] = j[1] * X[2] - j[2] * X[1];
        out[3 * r + 1] = j[2] * X[0] - j[0] * X[2];
        out[3 * r + 2] = j[0] * X[1] - j[1] * X[0];
    }
}

// In-place Cholesky solve of a dense SPD system, row-major n x n.  Returns false if not SPD.
ALVA_HD bool chol_solve_dense(double *A, double *b, int n) {
    for (int j = 0; j < n; j++) {
        double d = A[j * n + j];
        for (int k = 0; k < j; k++) d -= A[j * n + k] * A[j * n + k];
        if (!(d > 0)) return false;
        d = sqrt(d);
        A[j * n + j] = d;
        for (int i = j + 1; i < n; i++) {
            double s = A[i * n + j];
            for (int k = 0; k < j; k++) s -= A[i * n + k] * A[j * n + k];
            A[i * n + j] = s / d;
        }
    }
    for (int i = 0; i < n; i++) {
        double s = b[i];
        for (int k = 0; k < i; k++) s -= A[i * n + k] * b[k];
        b[i] = s / A[i * n + i];
    }
    for (int i = n - 1; i >= 0; i--) {
        double s = b[i];
        for (int k = i + 1; k < n; k++) s -= A[k * n + i] * b[k];
        b[i] = s / A[i * n + i];
    }
    return true;
}

// Fixed-size variant with every loop unrolled so that A and b stay in registers (no scratch / LDS round trips).
template<int N>
ALVA_HD bool chol_solve_fixed(double (&A)[N * N], double (&b)[N]) {
#pragma unroll
    for (int j = 0; j < N; j++) {
        double d = A[j * N + j];
#pragma unroll
        for (int k = 0; k < j; k++) d -= A[j * N + k] * A[j * N + k];
        if (!(d > 0)) return false;
        d = sqrt(d);
        A[j * N + j] = d;
#pragma unroll
        for (int i = j + 1; i < N; i++) {
            double s = A[i * N + j];
#pragma unroll
            for (int k = 0; k < j; k++) s -= A[i * N + k] * A[j * N + k];
            A[i * N + j] = s / d;
        }
    }
#pragma unroll
    for (int i = 0; i < N; i++) {
        double s = b[i];
#pragma unroll
        for (int k = 0; k < i; k++) s -= A[i * N + k] * b[k];
        b[i] = s / A[i * N + i];
    }
#pragma unroll
    for (int i = N - 1; i >= 0; i--) {
        double s = b[i];
#pragma unroll
        for (int k = i + 1; k < N; k++) s -= A[k * N + i] * b[k];
        b[i] = s / A[i * N + i];
    }
    return true;
}

// Ceres LevenbergMarquardtStrategy state (levenberg_marquardt_strategy.cc:48-160)
struct LmState {
    double radius = 1e4, decrease_factor = 2.0;
    int reuse_diagonal = 0;
    ALVA_HD void accepted(double q) {
        const double c = 2.0 * q - 1.0;
        double f = 1.0 - c * c * c;
        if (f < 1.0 / 3.0) f = 1.0 / 3.0;
        radius = radius / f;
        if (radius > 1e16) radius = 1e16;
        decrease_factor = 2.0;
        reuse_diagonal = 0;
    }
    ALVA_HD void rejected() {
        radius = radius / decrease_factor;
        decrease_factor *= 2.0;
        reuse_diagonal = 1;
    }
};

#ifdef __HIPCC__
// Reciprocal / reciprocal square root for places where ONE lane's (or one wave's) dependent chain is kernel time -- the pivots of the
// small Cholesky factorisations, a minimiser step on lane 0: the hardware estimates refined by two Newton steps (<= 1-2 ulp) cost ~10
// dependent instructions where the IEEE-rounded divide / square root expand to 25-30.  x > 0 and normal.
__device__ __forceinline__ double alva_fast_rcp(double x) {
    double r = __builtin_amdgcn_rcp(x);
    r = fma(fma(-x, r, 1.0), r, r);
    return fma(fma(-x, r, 1.0), r, r);
}
// returns 1 / sqrt(x); root ~ sqrt(x)
__device__ __forceinline__ double alva_fast_rsqrt(double x, double &root) {
    const double r = __builtin_amdgcn_rsq(x);
    double g = x * r, h = 0.5 * r;
    double e = fma(-h, g, 0.5);
    g = fma(g, e, g);
    h = fma(h, e, h);
    e = fma(-h, g, 0.5);
    root = fma(g, e, g);
    h = fma(h, e, h);
    return 2.0 * h;
}
#endif

