// a10-a13: local bundle adjustment -- Levenberg-Marquardt + Huber with a Schur complement on the point
// blocks, FP64, deterministic (no floating-point atomics).
//
// Replaces the ceres::Solve inside Optimizer::localBA (src/slam/src/optimizer.cpp:251-262; problem built at
// :20-247) with cost functions ReprojectionErrorKSE3AnchInvDepth / ...KSE3XYZ
// (src/slam/src/ceres_parametrization.cpp:157-268 / :6-94) and Ceres' SPARSE_SCHUR + LM semantics
// (trust_region_minimizer.cc, levenberg_marquardt_strategy.cc, schur_eliminator_impl.h:177-375), the
// reference's 10 ms wall-clock cap removed.
//
// Per Jacobian evaluation (observations are sorted by point once on the host):
//   k_point   one wave per map point, one lane per observation: residual, Huber, J_obs (2x6), J_e (2xd),
//             chi2 / depth side outputs; wave-reduces E'E, E'r and writes the point's row of W' = (F'E)'
//             into a dense [points x 6*cams(+pad)] matrix.  For anchored inverse depth J_anchor = -J_obs
//             (ceres_parametrization.cpp:239-256), so only J_obs and the scaled residual are stored
//             (112 B / residual block; inputs 60 B) for the camera-pair pass.
//   k_pairs   one workgroup per (observing kf, anchor kf) pair: sum of J_obs' J_obs and J_obs' r over the pair's
//             observations (a permutation built once on the host) -- from these 27 numbers per pair every
//             block of F'F and F'r follows by signs.
//   k_rowcol, k_hcc, k_gmax   F'F, F'r, Jacobi column scaling (iteration 0), gradient max-norm, total cost.
// Per LM step:
//   k_prep    per point: (E'E + D^2)^-1, its Cholesky factor L, Z_p = S_c W_p S_p L  and v_p = L'(E'r)
//   k_gemm    G = Z' Z with v appended as one more column: ONE dense [6*cams+1 x points*d]^2 FP64 GEMM on
//             v_mfma_f64_16x16x4_f64 (the only GEMM-shaped piece of the whole path) --
//             S = F'F + D^2 - G,  rhs = F'r - G[:, last]
//   k_reduced_system   S and the right-hand side, padded to a multiple of 16
//   k_solve   blocked dense Cholesky of the reduced camera system in one workgroup + blocked triangular solves
//   k_backsub per point: y_p, candidate point; model-cost-change and step-norm partials
//   k_update  candidate poses, model cost change, step norm, candidate norm
//   then the candidate is evaluated WITH its Jacobian (the first five kernels above): an accepted step -- the normal case --
//   needs exactly that evaluation next (trust_region_minimizer.cc:809-829), so the host reads its scalars ONCE per LM
//   iteration and applies Ceres' accept / reject logic; a rejected step re-evaluates at the previous point.
#include "common.hpp"
#include "lm_device.hpp"
#include "wave_utils.hpp"
#include <chrono>
#include <cstdlib>
#include <algorithm>
#include <cmath>
#include <numeric>

namespace {

typedef double double4_t __attribute__((ext_vector_type(4)));

struct BaDev {
    int nKf, nPt, nObs, inv, dp, nc, n6, NP;  // NP = padded reduced size (multiple of 16, >= n6 + 1)
    int npd;                                   // nPt * dp
    int kpad;                                  // K of the GEMM padded to a multiple of 4*KSPLIT chunks
    double K[4], huber_a;
    // static problem data
    const int *obsKf;       // [nObs] sorted by point
    const double *obsUv;    // [nObs][2]
    const int *ptPtr;       // [nPt+1]
    const int *ancKf;       // [nPt]
    const double *ancUv;    // [nPt][2]
    const int *cidx;        // [nKf] free index or -1
    double *rowcol;         // [2][nKf][27] row / column sums of M
    const int *kfOf;        // [nc] free camera index -> keyframe id
    const int *pairPerm;    // [nObs] observation ids grouped by pair key
    const int *pairPtr;     // [nKf*nKf+1]
    // work
    double *Jobs;   // [nObs][12]
    double *rs;     // [nObs][2]
    double *chi2;   // [nObs]
    uint8_t *depth; // [nObs]
    double *ptCost; // [nPt]
    double *Hpp;    // [nPt][dp*dp]
    double *gp;     // [nPt][dp]
    double *Wt;     // [npd][NP]   W' rows (unscaled)
    double *M;      // [nKf*nKf][27]
    double *Hcc;    // [n6][n6]
    double *gc;     // [n6]
    double *sc, *sp, *dc, *dpd;  // scalings and LM diagonals
    double *hinv;   // [nPt][dp*dp]
    double *Zt;     // [kpad][NP]
    double *Gpart;  // [KSPLIT][NP][NP]
    double *S;      // [n6][n6]
    double *yc;     // [NP]
    double *yp;     // [npd]
    double *scal;   // scalars: 0 cost, 1 mcc, 2 step_norm^2, 3 gmax, 4 x_norm^2, 5 chol_ok
    double *partial;  // [nPt][3] per-point partials for mcc / step norm / x norm
};

constexpr int KSPLIT = 8;

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off);
    return __shfl(v, 0);
}

// ------------------------------------------------------------------------------------------------------
template<bool INV, bool WANT_J>
__global__ void __launch_bounds__(256) k_point(BaDev B, const double *__restrict__ poses, const double *__restrict__ pts) {
    const int lane = threadIdx.x & 63;
    const int p = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (p >= B.nPt) return;
    constexpr int DP = INV ? 1 : 3;
    const int o0 = B.ptPtr[p], o1 = B.ptPtr[p + 1];
    double X[3], dJl[3] = {0, 0, 0};
    int a = -1;
    if (INV) {
        a = B.ancKf[p];
        Se3 Ta;
        se3_from_pose7(poses + 7 * a, Ta);
        const double zanch = 1.0 / pts[p];
        // anchpt = zanch * K^-1 * (u_a, v_a, 1)   (ceres_parametrization.cpp:176-184)
        const double ap[3] = {zanch * ((B.ancUv[2 * p] - B.K[2]) / B.K[0]), zanch * ((B.ancUv[2 * p + 1] - B.K[3]) / B.K[1]), zanch};
        double Ra[3];
        for (int i = 0; i < 3; i++) Ra[i] = Ta.R[3 * i] * ap[0] + Ta.R[3 * i + 1] * ap[1] + Ta.R[3 * i + 2] * ap[2];
        for (int i = 0; i < 3; i++) {
            X[i] = Ra[i] + Ta.t[i];
            dJl[i] = -zanch * Ra[i];  // J_lambda = -zanch * R_w,anch * anchpt (:262)
        }
    } else {
        for (int i = 0; i < 3; i++) X[i] = pts[3 * p + i];
    }
    double cost = 0, hpp[DP * DP], gpv[DP], wanc[6 * DP];
#pragma unroll
    for (int i = 0; i < DP * DP; i++) hpp[i] = 0;
#pragma unroll
    for (int i = 0; i < DP; i++) gpv[i] = 0;
#pragma unroll
    for (int i = 0; i < 6 * DP; i++) wanc[i] = 0;
    for (int o = o0 + lane; o < o1; o += 64) {
        const int k = B.obsKf[o];
        Se3 T;
        se3_from_pose7(poses + 7 * k, T);
        double r[2], JR[6], chi2;
        int dpz;
        reproj<WANT_J>(T, B.K, X, B.obsUv[2 * o], B.obsUv[2 * o + 1], r, JR, chi2, dpz);
        B.chi2[o] = chi2;
        B.depth[o] = (uint8_t) dpz;
        double rho0, rho1;
        huber_rho(chi2, B.huber_a, 1, rho0, rho1);
        cost += 0.5 * rho0;
        if (WANT_J) {
            const double s = sqrt(rho1);
            double JH[6], Jo[12], Je[2 * DP];
            times_hat(JR, X, JH);
#pragma unroll
            for (int rr = 0; rr < 2; rr++)
#pragma unroll
                for (int c = 0; c < 3; c++) {
                    Jo[6 * rr + c] = -JR[3 * rr + c] * s;
                    Jo[6 * rr + 3 + c] = JH[3 * rr + c] * s;
                }
            if (INV) {
                Je[0] = (JR[0] * dJl[0] + JR[1] * dJl[1] + JR[2] * dJl[2]) * s;
                Je[1] = (JR[3] * dJl[0] + JR[4] * dJl[1] + JR[5] * dJl[2]) * s;
            } else {
#pragma unroll
                for (int rr = 0; rr < 2; rr++)
#pragma unroll
                    for (int c = 0; c < 3; c++) Je[DP * rr + c] = JR[3 * rr + c] * s;
            }
            const double r0 = r[0] * s, r1 = r[1] * s;
#pragma unroll
            for (int i = 0; i < 12; i++) B.Jobs[(size_t) o * 12 + i] = Jo[i];
            B.rs[2 * (size_t) o] = r0;
            B.rs[2 * (size_t) o + 1] = r1;
#pragma unroll
            for (int x = 0; x < DP; x++) {
                gpv[x] += Je[x] * r0 + Je[DP + x] * r1;
#pragma unroll
                for (int y = 0; y < DP; y++) hpp[x * DP + y] += Je[x] * Je[y] + Je[DP + x] * Je[DP + y];
            }
            const int c = B.cidx[k];
#pragma unroll
            for (int i = 0; i < 6; i++)
#pragma unroll
                for (int x = 0; x < DP; x++) {
                    const double w = Jo[i] * Je[x] + Jo[6 + i] * Je[DP + x];
                    if (c >= 0) B.Wt[((size_t) p * DP + x) * B.NP + 6 * c + i] = w;  // W_c = J_obs' J_e
                    wanc[i * DP + x] -= w;                                            // W_a = -sum J_obs' J_e
                }
        }
    }
    cost = wave_sum(cost);
    if (lane == 0) B.ptCost[p] = cost;
    if (WANT_J) {
#pragma unroll
        for (int i = 0; i < DP * DP; i++) hpp[i] = wave_sum(hpp[i]);
#pragma unroll
        for (int i = 0; i < DP; i++) gpv[i] = wave_sum(gpv[i]);
        if (lane == 0) {
#pragma unroll
            for (int i = 0; i < DP * DP; i++) B.Hpp[(size_t) p * DP * DP + i] = hpp[i];
#pragma unroll
            for (int i = 0; i < DP; i++) B.gp[(size_t) p * DP + i] = gpv[i];
        }
        if (INV) {
            const int ca = B.cidx[a];
            if (ca >= 0) {
#pragma unroll
                for (int i = 0; i < 6; i++) {
                    const double w = wave_sum(wanc[i]);
                    if (lane == 0) B.Wt[(size_t) p * B.NP + 6 * ca + i] = w;
                }
            }
        }
    }
}

// One wave per (observing kf, anchor kf) pair: sums of J_obs'J_obs (21) and J_obs' r (6).
__global__ void __launch_bounds__(256) k_pairs(BaDev B) {
    // one WORKGROUP per (observing kf, anchor kf) pair: the big pairs (thousands of observations) set the kernel time, so
    // their observations are spread over 256 lanes; lane partials -> butterfly reduce-scatter per wave -> 4 waves in LDS
    // (fixed order: bit-reproducible)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int key = blockIdx.x;
    __shared__ double s_part[4][28];
    double v[32];
#pragma unroll
    for (int i = 0; i < 32; i++) v[i] = 0;
    for (int q = B.pairPtr[key] + threadIdx.x; q < B.pairPtr[key + 1]; q += 256) {
        const int o = B.pairPerm[q];
        double J[12];
#pragma unroll
        for (int i = 0; i < 12; i++) J[i] = B.Jobs[(size_t) o * 12 + i];
        const double r0 = B.rs[2 * (size_t) o], r1 = B.rs[2 * (size_t) o + 1];
        int t = 0;
#pragma unroll
        for (int x = 0; x < 6; x++) {
            v[21 + x] += J[x] * r0 + J[6 + x] * r1;
#pragma unroll
            for (int y = x; y < 6; y++) v[t++] += J[x] * J[y] + J[6 + x] * J[6 + y];
        }
    }
    wave_reduce_scatter32(v);  // lane l ends with the wave total of value l >> 1
    if (!(lane & 1) && (lane >> 1) < 27) s_part[wave][lane >> 1] = v[0];
    __syncthreads();
    if (threadIdx.x < 27) B.M[(size_t) key * 27 + threadIdx.x] = ((s_part[0][threadIdx.x] + s_part[1][threadIdx.x]) + s_part[2][threadIdx.x]) + s_part[3][threadIdx.x];
}

__device__ __forceinline__ int tri6(int x, int y) {
    if (x > y) {
        const int t = x;
        x = y;
        y = t;
    }
    return x * 6 - x * (x - 1) / 2 + (y - x);
}

// F'F, F'r from the pair sums; Jacobi scaling at iteration 0; gradient max-norm; cost.
// With M[c][a] = sum over observations (cam c, anchor a) of [J'J (21) | J'r (6)]:
//   F'F(c,c) = sum_a M[c][a] + sum_c' M[c'][c]      F'F(c,a) = -(M[c][a] + M[a][c])  (c != a)
//   F'r(c)   = sum_a m[c][a] - sum_c' m[c'][c]       (J_anchor = -J_obs).  XYZ mode: only M[c][c].
// Assembly of the camera block H_cc / g_c from the per-pair sums, in three small launches instead of one single-workgroup
// kernel (which spent 34 us walking 11.6 k dependent index computations + loads with 256 threads):
//   k_rowcol   per keyframe: row and column sums of the pair sums
//   k_hcc      every element of H_cc, g_c, and (first evaluation) the Jacobi scaling of the cameras
//   k_gmax     (first evaluation) the Jacobi scaling of the points; max |gradient| -> scal[3]
__global__ void __launch_bounds__(64) k_rowcol(BaDev B) {
    const int k = blockIdx.x, t = threadIdx.x, nKf = B.nKf;
    if (t >= 27) return;
    double rs = 0, cs = 0;
    for (int j = 0; j < nKf; j++) {
        rs += B.M[(size_t) (k * nKf + j) * 27 + t];
        cs += B.M[(size_t) (j * nKf + k) * 27 + t];
    }
    B.rowcol[(size_t) k * 27 + t] = rs;
    B.rowcol[(size_t) (nKf + k) * 27 + t] = cs;
}

__global__ void __launch_bounds__(256) k_hcc(BaDev B, int first) {
    const int n6 = B.n6, nKf = B.nKf;
    const double *rowsum = B.rowcol, *colsum = B.rowcol + (size_t) nKf * 27;
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e < n6 * n6) {
        const int r = e / n6, c = e - r * n6, cr = r / 6, cc = c / 6, x = r - 6 * cr, y = c - 6 * cc;
        const int kr = B.kfOf[cr], kc = B.kfOf[cc];
        const int t = tri6(x, y);
        double v = 0;
        if (B.inv) {
            if (kr == kc) v = rowsum[kr * 27 + t] + colsum[kr * 27 + t];
            else v = -(B.M[(size_t) (kr * nKf + kc) * 27 + t] + B.M[(size_t) (kc * nKf + kr) * 27 + t]);
        } else {
            if (kr == kc) v = B.M[(size_t) (kr * nKf + kr) * 27 + t];
        }
        B.Hcc[e] = v;
        if (first && r == c) B.sc[r] = 1.0 / (1.0 + sqrt(v));
    } else if (e < n6 * n6 + n6) {
        const int r = e - n6 * n6, kr = B.kfOf[r / 6], x = r % 6;
        B.gc[r] = B.inv ? rowsum[kr * 27 + 21 + x] - colsum[kr * 27 + 21 + x] : B.M[(size_t) (kr * nKf + kr) * 27 + 21 + x];
    }
}

// max |gradient| (scal[3]) and, same single workgroup, the total cost (scal[0] = sum of the per-point costs of k_point)
__global__ void __launch_bounds__(256) k_gmax(BaDev B, int first) {
    __shared__ double s_red[256], s_cost[256];
    if (first)
        for (int i = threadIdx.x; i < B.npd; i += 256) {
            const int p = i / B.dp, x = i % B.dp;
            B.sp[i] = 1.0 / (1.0 + sqrt(B.Hpp[(size_t) p * B.dp * B.dp + x * B.dp + x]));
        }
    double gm = 0, v = 0;
    for (int r = threadIdx.x; r < B.n6; r += 256) gm = fmax(gm, fabs(B.gc[r]));
    for (int i = threadIdx.x; i < B.npd; i += 256) gm = fmax(gm, fabs(B.gp[i]));
    for (int p = threadIdx.x; p < B.nPt; p += 256) v += B.ptCost[p];
    s_red[threadIdx.x] = gm;
    s_cost[threadIdx.x] = v;
    __syncthreads();
    for (int s2 = 128; s2 > 0; s2 >>= 1) {
        if (threadIdx.x < s2) {
            s_red[threadIdx.x] = fmax(s_red[threadIdx.x], s_red[threadIdx.x + s2]);
            s_cost[threadIdx.x] += s_cost[threadIdx.x + s2];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        B.scal[3] = s_red[0];
        B.scal[0] = s_cost[0];
    }
}

// LM diagonal (levenberg_marquardt_strategy.cc:79-90), refreshed only after an accepted step.
__global__ void __launch_bounds__(256) k_diag(BaDev B) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < B.n6) B.dc[i] = fmin(fmax(B.Hcc[(size_t) i * B.n6 + i] * B.sc[i] * B.sc[i], 1e-6), 1e32);
    if (i < B.npd) {
        const int p = i / B.dp, x = i % B.dp;
        B.dpd[i] = fmin(fmax(B.Hpp[(size_t) p * B.dp * B.dp + x * B.dp + x] * B.sp[i] * B.sp[i], 1e-6), 1e32);
    }
}

// per point: hinv = (S_p E'E S_p + D_p^2/radius)^-1 , L = chol(hinv), Zt rows = S_c W S_p L, last column v = L' (S_p E'r)
template<int DP>
__global__ void __launch_bounds__(256) k_prep(BaDev B, double radius) {
    const int lane = threadIdx.x & 63;
    const int p = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (p >= B.nPt) return;
    double Mx[DP * DP], Hi[DP * DP], L[DP * DP], gs[DP];
#pragma unroll
    for (int x = 0; x < DP; x++) {
        gs[x] = B.gp[(size_t) p * DP + x] * B.sp[p * DP + x];
#pragma unroll
        for (int y = 0; y < DP; y++)
            Mx[x * DP + y] = B.Hpp[(size_t) p * DP * DP + x * DP + y] * B.sp[p * DP + x] * B.sp[p * DP + y] +
                             (x == y ? B.dpd[p * DP + x] / radius : 0.0);
    }
    if (DP == 1) {
        Hi[0] = 1.0 / Mx[0];
        L[0] = sqrt(Hi[0]);
    } else {
        // inverse of a 3x3 SPD matrix by solving against the identity, then its Cholesky factor
        for (int c = 0; c < DP; c++) {
            double A3[DP * DP], e[DP];
            for (int i = 0; i < DP * DP; i++) A3[i] = Mx[i];
            for (int i = 0; i < DP; i++) e[i] = (i == c);
            chol_solve_dense(A3, e, DP);
            for (int i = 0; i < DP; i++) Hi[i * DP + c] = e[i];
        }
        for (int i = 0; i < DP * DP; i++) L[i] = 0;
        for (int j = 0; j < DP; j++) {
            double d = Hi[j * DP + j];
            for (int k = 0; k < j; k++) d -= L[j * DP + k] * L[j * DP + k];
            d = sqrt(d);
            L[j * DP + j] = d;
            for (int i = j + 1; i < DP; i++) {
                double s = Hi[i * DP + j];
                for (int k = 0; k < j; k++) s -= L[i * DP + k] * L[j * DP + k];
                L[i * DP + j] = s / d;
            }
        }
    }
    if (lane == 0)
        for (int i = 0; i < DP * DP; i++) B.hinv[(size_t) p * DP * DP + i] = Hi[i];
    // Z_p[r][y] = sum_x sc[r] W[r][x] sp[x] L[x][y]; stored transposed: Zt[p*DP+y][r]
    for (int r = lane; r < B.NP; r += 64) {
#pragma unroll
        for (int y = 0; y < DP; y++) {
            double v = 0;
            if (r < B.n6) {
#pragma unroll
                for (int x = 0; x < DP; x++) v += B.sc[r] * B.Wt[((size_t) p * DP + x) * B.NP + r] * B.sp[p * DP + x] * L[x * DP + y];
            } else if (r == B.n6) {
#pragma unroll
                for (int x = 0; x < DP; x++) v += L[x * DP + y] * gs[x];  // v = L' g_s
            }
            B.Zt[((size_t) p * DP + y) * B.NP + r] = v;
        }
    }
}

// G_part[ks] (16x16 tile) = Zt[kchunk]' Zt[kchunk] on the FP64 matrix core.
// v_mfma_f64_16x16x4_f64: A[l&15][k=l>>4], B[k=l>>4][l&15], C/D col = l&15, row = (l>>4) + 4*reg.
__global__ void __launch_bounds__(64) k_gemm(BaDev B) {
    const int tiles = B.NP / 16;
    const int ti = blockIdx.x / tiles, tj = blockIdx.x % tiles, ks = blockIdx.y;
    const int lane = threadIdx.x;
    const int chunk = B.kpad / KSPLIT;
    const int k0 = ks * chunk;
    double4_t acc = {0, 0, 0, 0};
    const double *Za = B.Zt + (size_t) (k0 + (lane >> 4)) * B.NP + ti * 16 + (lane & 15);
    const double *Zb = B.Zt + (size_t) (k0 + (lane >> 4)) * B.NP + tj * 16 + (lane & 15);
    // eight k-steps of operands in flight per trip: with one load pair per MFMA the loop ran at the latency of one L2 round trip
    // per step (30 us for 94 steps)
    int k = 0;
    for (; k + 32 <= chunk; k += 32) {
        double a[8], b[8];
#pragma unroll
        for (int u = 0; u < 8; u++) {
            a[u] = Za[(size_t) (k + 4 * u) * B.NP];
            b[u] = Zb[(size_t) (k + 4 * u) * B.NP];
        }
#pragma unroll
        for (int u = 0; u < 8; u++) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[u], b[u], acc, 0, 0, 0);
    }
    for (; k < chunk; k += 4) {
        const double a = Za[(size_t) k * B.NP], b = Zb[(size_t) k * B.NP];
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
    }
    double *G = B.Gpart + ((size_t) ks * B.NP + ti * 16) * B.NP + tj * 16;
#pragma unroll
    for (int r = 0; r < 4; r++) G[(size_t) ((lane >> 4) + 4 * r) * B.NP + (lane & 15)] = acc[r];
}

// S = S_c F'F S_c + D_c^2/radius - G ; rhs = S_c F'r - G[:, n6] ; dense Cholesky; y_c.  One workgroup.
//
// Blocked right-looking Cholesky, block 16, the matrix padded with an identity to a multiple of 16 (no edge cases):
//   diagonal block   wave 0, one row per lane IN REGISTERS, pivots / multipliers broadcast with v_readlane (no LDS, no barrier)
//   panel            one thread per row below: x L11' = a, 136 FMAs against the (broadcast-read) diagonal block
//   trailing update  A22 -= L21 L21' as 16 x 16 tiles on v_mfma_f64_16x16x4_f64, one wavefront per tile
// = 3 barriers per 16 columns instead of 3 per column; the triangular solves are blocked the same way.  Row stride is
// odd (padded size + 1) so that threads reading different rows hit different LDS banks.
constexpr int SOLVE_NT = 256, NB = 16;

// reduced camera system of this LM step, padded: S [np][np + 1] and the right-hand side [np] right behind it (all CUs)
__global__ void __launch_bounds__(256) k_reduced_system(BaDev B, double radius) {
    const int n = B.n6, np = (n + NB - 1) / NB * NB, ld = np + 1;
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e < np * np) {
        const int r = e / np, c = e - r * np;
        double v = r == c ? 1.0 : 0.0;  // identity padding
        if (r < n && c < n) {
            double g = 0;
#pragma unroll
            for (int ks = 0; ks < KSPLIT; ks++) g += B.Gpart[((size_t) ks * B.NP + r) * B.NP + c];
            v = B.Hcc[(size_t) r * n + c] * B.sc[r] * B.sc[c] - g;
            if (r == c) v += B.dc[r] / radius;
        }
        B.S[(size_t) r * ld + c] = v;
    } else if (e < np * np + np) {
        const int r = e - np * np;
        double v = 0;
        if (r < n) {
            double g = 0;
#pragma unroll
            for (int ks = 0; ks < KSPLIT; ks++) g += B.Gpart[((size_t) ks * B.NP + r) * B.NP + n];
            v = B.gc[r] * B.sc[r] - g;
        }
        B.S[(size_t) np * ld + r] = v;
    }
}

template<bool IN_LDS>
__global__ void __launch_bounds__(SOLVE_NT) k_solve(BaDev B, double radius) {
    extern __shared__ double s_S[];
    const int n = B.n6, np = (n + NB - 1) / NB * NB, ld = np + 1, nb = np / NB;
    __shared__ double s_inv[NB], s_z[NB];
    __shared__ int s_ok;
    double *S = IN_LDS ? s_S : B.S;            // [np][ld]
    double *y = IN_LDS ? s_S + (size_t) np * ld : B.S + (size_t) np * ld;  // [np] right-hand side / solution
    if (IN_LDS) {  // matrix + right-hand side from k_reduced_system: coalesced, eight loads in flight per thread (one load -> one
                   // LDS store per trip took 14 us for the 100 KB of a 108-unknown system)
        const int total = np * ld + np;
        for (int e0 = threadIdx.x; e0 < total; e0 += 8 * SOLVE_NT) {
            double v[8];
#pragma unroll
            for (int u = 0; u < 8; u++) {
                const int e = e0 + u * SOLVE_NT;
                v[u] = e < total ? B.S[e] : 0.0;
            }
#pragma unroll
            for (int u = 0; u < 8; u++) {
                const int e = e0 + u * SOLVE_NT;
                if (e < total) s_S[e] = v[u];
            }
        }
    }
    // micro-tile enumeration of a lower triangle, row by row: t -> (ta, tb), the same for every trailing size
    constexpr int TILE_LUT = 1024;   // 16 x 16 tiles: covers np <= 736; larger systems decode arithmetically
    __shared__ unsigned short s_tile[TILE_LUT];
    for (int t = threadIdx.x; t < TILE_LUT; t += SOLVE_NT) {
        int ta = (int) ((sqrtf(8.f * (float) t + 1.f) - 1.f) * 0.5f);
        while (ta * (ta + 1) / 2 > t) ta--;
        while ((ta + 1) * (ta + 2) / 2 <= t) ta++;
        s_tile[t] = (unsigned short) ((ta << 8) | (t - ta * (ta + 1) / 2));
    }
    if (threadIdx.x == 0) s_ok = 1;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    for (int kb = 0; kb < nb; kb++) {
        const int base = kb * NB;

        // ---- diagonal block: lane i (< 16) owns row i --------------------------------------------------------------
        if (threadIdx.x < 64) {
            double a[NB];
            const int row = base + (lane & 15);
#pragma unroll
            for (int k = 0; k < NB; k++) a[k] = S[(size_t) row * ld + base + k];
            int ok = 1;
#pragma unroll
            for (int j = 0; j < NB; j++) {
                const double d = lane_bcast(a[j], j);
                if (!(d > 0)) ok = 0;
                const double inv = rsqrt(d), piv = d * inv;
                a[j] = (lane & 15) == j ? piv : a[j] * inv;  // rows above j hold garbage in column j: never read
                if (lane == j) s_inv[j] = inv;
#pragma unroll
                for (int k = j + 1; k < NB; k++) {
                    const double lkj = lane_bcast(a[j], k);
                    a[k] -= a[j] * lkj;  // used for rows >= k only
                }
            }
            if (lane < NB) {
#pragma unroll
                for (int k = 0; k < NB; k++)
                    if (k <= lane) S[(size_t) row * ld + base + k] = a[k];
            }
            if (lane == 0 && !ok) s_ok = 0;
        }
        __syncthreads();
        if (!s_ok) break;
        // ---- panel: x L11' = a for every row below --------------------------------------------------------------------
        const int m = np - base - NB;  // rows below the block
        for (int r = threadIdx.x; r < m; r += SOLVE_NT) {
            double *rowp = S + (size_t) (base + NB + r) * ld + base;
            double x[NB];
#pragma unroll
            for (int j = 0; j < NB; j++) {
                double v = rowp[j];
                const double *lj = S + (size_t) (base + j) * ld + base;
#pragma unroll
                for (int k = 0; k < j; k++) v -= x[k] * lj[k];
                x[j] = v * s_inv[j];
            }
#pragma unroll
            for (int j = 0; j < NB; j++) rowp[j] = x[j];
        }
        __syncthreads();
        // ---- trailing update A22 -= L21 L21': a rank-16 update, i.e. one 16 x 16 x 16 product per 16 x 16 tile of the lower triangle
        //      = four v_mfma_f64_16x16x4_f64 per tile, one wavefront per tile (operand a: L[rowa + (lane & 15)][4c + (lane >> 4)],
        //      operand b the same for rowb; result row (lane >> 4) + 4r, column lane & 15).  Diagonal tiles are updated in full: the
        //      strict upper triangle is never read.
        const int mt = m / NB, ntile = mt * (mt + 1) / 2;
        for (int t = threadIdx.x >> 6; t < ntile; t += SOLVE_NT / 64) {
            int ta, tb;
            if (t < TILE_LUT) {
                ta = s_tile[t] >> 8;
                tb = s_tile[t] & 255;
            } else {
                ta = (int) ((sqrt(8.0 * t + 1.0) - 1.0) * 0.5);
                while (ta * (ta + 1) / 2 > t) ta--;
                while ((ta + 1) * (ta + 2) / 2 <= t) ta++;
                tb = t - ta * (ta + 1) / 2;
            }
            const double *La = S + (size_t) (base + NB + NB * ta + (lane & 15)) * ld + base + (lane >> 4);
            const double *Lb = S + (size_t) (base + NB + NB * tb + (lane & 15)) * ld + base + (lane >> 4);
            double4_t acc = {0, 0, 0, 0};
#pragma unroll
            for (int c = 0; c < NB / 4; c++) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(La[4 * c], Lb[4 * c], acc, 0, 0, 0);
            double *C = S + (size_t) (base + NB + NB * ta + (lane >> 4)) * ld + base + NB + NB * tb + (lane & 15);
#pragma unroll
            for (int r = 0; r < 4; r++) C[(size_t) (4 * r) * ld] -= acc[r];
        }
        __syncthreads();
    }
    if (s_ok) {
        // ---- L z = rhs, blocked -----------------------------------------------------------------------------------------
        for (int kb = 0; kb < nb; kb++) {
            const int base = kb * NB;
            if (threadIdx.x < 64) {
                const int i = lane & 15;
                double l[NB];
#pragma unroll
                for (int k = 0; k < NB; k++) l[k] = S[(size_t) (base + i) * ld + base + k];
                double yi = y[base + i];
#pragma unroll
                for (int j = 0; j < NB; j++) {
                    const double zj = lane_bcast(yi, j) / lane_bcast(l[j], j);
                    if (i == j) yi = zj;
                    else if (i > j) yi -= l[j] * zj;
                }
                if (lane < NB) {
                    y[base + i] = yi;
                    s_z[i] = yi;
                }
            }
            __syncthreads();
            for (int r = base + NB + threadIdx.x; r < np; r += SOLVE_NT) {
                const double *lr = S + (size_t) r * ld + base;
                double v = y[r];
#pragma unroll
                for (int k = 0; k < NB; k++) v -= lr[k] * s_z[k];
                y[r] = v;
            }
            __syncthreads();
        }
        // ---- L' y = z, blocked, last block first ------------------------------------------------------------------------------
        for (int kb = nb - 1; kb >= 0; kb--) {
            const int base = kb * NB;
            if (threadIdx.x < 64) {
                const int i = lane & 15;
                double lc[NB];  // column i of the diagonal block: lc[j] = L[base + j][base + i]
#pragma unroll
                for (int j = 0; j < NB; j++) lc[j] = S[(size_t) (base + j) * ld + base + i];
                double zi = y[base + i];
#pragma unroll
                for (int j = NB - 1; j >= 0; j--) {
                    const double yj = lane_bcast(zi, j) / lane_bcast(lc[j], j);
                    if (i == j) zi = yj;
                    else if (i < j) zi -= lc[j] * yj;
                }
                if (lane < NB) {
                    y[base + i] = zi;
                    s_z[i] = zi;
                }
            }
            __syncthreads();
            for (int r = threadIdx.x; r < base; r += SOLVE_NT) {
                double v = y[r];
#pragma unroll
                for (int k = 0; k < NB; k++) v -= S[(size_t) (base + k) * ld + r] * s_z[k];
                y[r] = v;
            }
            __syncthreads();
        }
        for (int r = threadIdx.x; r < n; r += SOLVE_NT) B.yc[r] = y[r];
    }
    if (threadIdx.x == 0) B.scal[5] = s_ok ? 1.0 : 0.0;
}

// per point: y_p = hinv (g_s - (S_c W S_p)' y_c); candidate point; partials for the model cost change
// (1/2 y'(g_s + D y), exact for the exact solution of (H_s + D) y = g_s) and the step norm.
template<int DP>
__global__ void __launch_bounds__(256) k_backsub(BaDev B, double radius, const double *__restrict__ x_t, double *__restrict__ c_t) {
    const int lane = threadIdx.x & 63;
    const int p = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (p >= B.nPt) return;
    double t[DP];
#pragma unroll
    for (int x = 0; x < DP; x++) {
        double v = 0;
        for (int r = lane; r < B.n6; r += 64) v += B.Wt[((size_t) p * DP + x) * B.NP + r] * B.sc[r] * B.yc[r];
        v = wave_sum(v);
        t[x] = B.gp[(size_t) p * DP + x] * B.sp[p * DP + x] - v * B.sp[p * DP + x];
    }
    if (lane == 0) {
        double mcc = 0, sn = 0;
#pragma unroll
        for (int x = 0; x < DP; x++) {
            double y = 0;
#pragma unroll
            for (int z = 0; z < DP; z++) y += B.hinv[(size_t) p * DP * DP + x * DP + z] * t[z];
            B.yp[p * DP + x] = y;
            const double gs = B.gp[(size_t) p * DP + x] * B.sp[p * DP + x];
            mcc += 0.5 * y * (gs + B.dpd[p * DP + x] / radius * y);
            const double d = -y * B.sp[p * DP + x];
            c_t[p * DP + x] = x_t[p * DP + x] + d;
            sn += d * d;
        }
        B.partial[3 * (size_t) p] = mcc;
        B.partial[3 * (size_t) p + 1] = sn;
    }
}

// candidate poses (SE3 Plus), camera part of mcc / step norm, and the final deterministic reductions.
// candidate poses, model cost change (scal[1]), squared step norm (scal[2]) and the squared norm of the CANDIDATE (scal[4]: free
// poses + point parameters c_t written by k_backsub), which becomes x_norm when the step is accepted
__global__ void __launch_bounds__(256) k_update(BaDev B, double radius, const double *__restrict__ x_p, double *__restrict__ c_p,
                                                const double *__restrict__ c_t) {
    __shared__ double s_a[256], s_b[256], s_c[256];
    double mcc = 0, sn = 0, xn = 0;
    for (int k = threadIdx.x; k < B.nKf; k += 256) {
        const int c = B.cidx[k];
        if (c < 0) {
            for (int i = 0; i < 7; i++) c_p[7 * k + i] = x_p[7 * k + i];
            continue;
        }
        double d6[6];
        for (int i = 0; i < 6; i++) {
            const double y = B.yc[6 * c + i];
            d6[i] = -y * B.sc[6 * c + i];
            mcc += 0.5 * y * (B.gc[6 * c + i] * B.sc[6 * c + i] + B.dc[6 * c + i] / radius * y);
        }
        se3_plus(x_p + 7 * k, d6, c_p + 7 * k);
        for (int i = 0; i < 7; i++) {
            const double cv = c_p[7 * k + i], d = x_p[7 * k + i] - cv;
            sn += d * d;
            xn += cv * cv;
        }
    }
    for (int p = threadIdx.x; p < B.nPt; p += 256) {
        mcc += B.partial[3 * (size_t) p];
        sn += B.partial[3 * (size_t) p + 1];
    }
    for (int i = threadIdx.x; i < B.npd; i += 256) xn += c_t[i] * c_t[i];
    s_a[threadIdx.x] = mcc;
    s_b[threadIdx.x] = sn;
    s_c[threadIdx.x] = xn;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (threadIdx.x < s) {
            s_a[threadIdx.x] += s_a[threadIdx.x + s];
            s_b[threadIdx.x] += s_b[threadIdx.x + s];
            s_c[threadIdx.x] += s_c[threadIdx.x + s];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        B.scal[1] = s_a[0];
        B.scal[2] = s_b[0];
        B.scal[4] = s_c[0];
    }
}

// |x|^2 over the variable blocks (free poses in their 7-vector form + all point parameters) -> scal[4]

template<typename T>
T *carve(uint8_t *&cur, size_t count) {
    T *p = reinterpret_cast<T *>(cur);
    cur += (count * sizeof(T) + 255) / 256 * 256;
    return p;
}

}  // namespace

extern "C" int alva_local_ba(alva_ctx *ctx, int n_kf, double *h_poses, const uint8_t *h_kf_const, const double *h_calib, int inv_depth,
                             int n_pt, const int *h_pt_anchor_kf, const double *h_pt_anchor_uv, double *h_pt_param, int n_obs,
                             const int *h_obs_kf, const int *h_obs_pt, const double *h_obs_uv, int max_iters,
                             double function_tolerance, double huber_chi2, double *h_chi2, uint8_t *h_depth_pos, double *h_info,
                             int *h_ok) {
    ALVA_ARG(ctx && h_poses && h_kf_const && h_calib && h_pt_param && h_ok && n_kf > 0 && n_pt >= 0 && n_obs >= 0 && max_iters >= 0);
    ALVA_ARG(n_obs == 0 || (h_obs_kf && h_obs_pt && h_obs_uv));
    ALVA_ARG(!inv_depth || n_pt == 0 || (h_pt_anchor_kf && h_pt_anchor_uv));
    *h_ok = 1;
    if (h_info) memset(h_info, 0, 4 * sizeof(double));
    const int dp = inv_depth ? 1 : 3;
    // ---- sizes -----------------------------------------------------------------------------------------------
    const auto t_begin = std::chrono::steady_clock::now();
    std::vector<int> cidx((size_t) n_kf);
    int nc = 0;
    for (int k = 0; k < n_kf; k++) cidx[(size_t) k] = h_kf_const[k] ? -1 : nc++;
    for (int o = 0; o < n_obs; o++) ALVA_ARG(h_obs_kf[o] >= 0 && h_obs_kf[o] < n_kf && h_obs_pt[o] >= 0 && h_obs_pt[o] < n_pt);
    BaDev B{};
    B.nKf = n_kf; B.nPt = n_pt; B.nObs = n_obs; B.inv = inv_depth; B.dp = dp; B.nc = nc; B.n6 = 6 * nc;
    B.NP = (B.n6 + 1 + 15) / 16 * 16;
    B.npd = n_pt * dp;
    const int kq = 4 * KSPLIT;
    B.kpad = std::max(kq, (B.npd + kq - 1) / kq * kq);
    for (int i = 0; i < 4; i++) B.K[i] = h_calib[i];
    B.huber_a = (double) sqrtf((float) huber_chi2);  // optimizer.cpp:22: std::sqrt of a float

    // ---- one scratch block, carved; the INPUT arrays come first and contiguous so that one copy uploads them ----------
    const size_t nObs = (size_t) n_obs, nPt = (size_t) n_pt, npd = (size_t) B.npd, n6 = (size_t) B.n6, NP = (size_t) B.NP;
    int *d_obsKf, *d_ptPtr, *d_ancKf, *d_cidx, *d_pairPerm, *d_pairPtr, *d_kfOf;
    double *d_obsUv, *d_ancUv, *d_xp, *d_cp, *d_xt, *d_ct;
    size_t in_bytes = 0, off_res = 0;
    auto layout = [&](uint8_t *base) -> size_t {
        uint8_t *cur = base;
        d_obsKf = carve<int>(cur, nObs);
        d_obsUv = carve<double>(cur, nObs * 2);
        d_ptPtr = carve<int>(cur, nPt + 1);
        d_ancKf = carve<int>(cur, nPt);
        d_ancUv = carve<double>(cur, nPt * 2);
        d_cidx = carve<int>(cur, (size_t) n_kf);
        d_kfOf = carve<int>(cur, (size_t) n_kf);
        d_pairPerm = carve<int>(cur, nObs);
        d_pairPtr = carve<int>(cur, (size_t) n_kf * n_kf + 1);
        d_xp = carve<double>(cur, (size_t) n_kf * 7);
        d_xt = carve<double>(cur, npd);
        in_bytes = (size_t) (cur - base);
        off_res = in_bytes;
        B.chi2 = carve<double>(cur, nObs);   // results the host reads back: chi2 | depth flags, contiguous
        B.depth = carve<uint8_t>(cur, nObs);
        B.Jobs = carve<double>(cur, nObs * 12);
        B.rs = carve<double>(cur, nObs * 2);
        B.ptCost = carve<double>(cur, nPt);
        B.Hpp = carve<double>(cur, npd * dp);
        B.gp = carve<double>(cur, npd);
        B.Wt = carve<double>(cur, npd * NP);
        B.M = carve<double>(cur, (size_t) n_kf * n_kf * 27);
        B.rowcol = carve<double>(cur, (size_t) n_kf * 27 * 2);
        B.Hcc = carve<double>(cur, n6 * n6);
        B.gc = carve<double>(cur, n6);
        B.sc = carve<double>(cur, n6);
        B.sp = carve<double>(cur, npd);
        B.dc = carve<double>(cur, n6);
        B.dpd = carve<double>(cur, npd);
        B.hinv = carve<double>(cur, npd * dp);
        B.Zt = carve<double>(cur, (size_t) B.kpad * NP);
        B.Gpart = carve<double>(cur, (size_t) KSPLIT * NP * NP);
        B.S = carve<double>(cur, (n6 + 16) * (n6 + 17) + n6 + 16);  // padded to a multiple of 16, odd row stride, + rhs
        B.yc = carve<double>(cur, NP);
        B.yp = carve<double>(cur, npd);
        B.scal = carve<double>(cur, 64);
        B.partial = carve<double>(cur, nPt * 3);
        d_cp = carve<double>(cur, (size_t) n_kf * 7);
        d_ct = carve<double>(cur, npd);
        return (size_t) (cur - base);
    };
    const size_t bytes = layout(nullptr);
    uint8_t *base = nullptr;
    int rc = alva_ctx_scratch(ctx, 4, bytes, (void **) &base);
    if (rc) return rc;
    // pinned staging: [0, 256) the per-iteration scalars | the input block (mirror of the device layout) | results
    const size_t res_bytes = (nObs * 9 + 511) / 256 * 256 + (size_t) n_kf * 56 + 256 + npd * 8 + 256;
    uint8_t *pin = nullptr;
    rc = alva_ctx_pinned(ctx, 256 + std::max(in_bytes, res_bytes), (void **) &pin);
    if (rc) return rc;
    uint8_t *stage = pin + 256;
    layout(stage);  // host mirrors of the input arrays, written in place
    int *h_obsKf = d_obsKf, *h_ptPtr = d_ptPtr, *h_ancKf = d_ancKf, *h_cidx = d_cidx, *h_kfOf = d_kfOf, *h_pairPerm = d_pairPerm,
        *h_pairPtr = d_pairPtr;
    double *h_obsUv = d_obsUv, *h_ancUv = d_ancUv, *h_xp = d_xp, *h_xt = d_xt;
    layout(base);
    B.obsKf = d_obsKf; B.obsUv = d_obsUv; B.ptPtr = d_ptPtr; B.ancKf = d_ancKf; B.ancUv = d_ancUv; B.cidx = d_cidx; B.kfOf = d_kfOf;
    B.pairPerm = d_pairPerm; B.pairPtr = d_pairPtr;
    hipStream_t st = ctx->stream;
    ALVA_HIP(hipStreamSynchronize(st));  // nothing enqueued earlier may still be reading the staging area

    // ---- host-side structure (the analogue of Ceres' program / block-structure build), built in the staging area ---------
    // observations grouped by point, original order kept inside a point: a stable counting sort (O(n))
    std::vector<int> order((size_t) n_obs), pairKey((size_t) n_obs);
    for (size_t p2 = 0; p2 <= nPt; p2++) h_ptPtr[p2] = 0;
    {
        std::vector<int> cursor((size_t) n_pt + 1, 0);
        for (int o = 0; o < n_obs; o++) cursor[(size_t) h_obs_pt[o] + 1]++;
        for (int p2 = 0; p2 < n_pt; p2++) cursor[(size_t) p2 + 1] += cursor[(size_t) p2];
        for (int o = 0; o < n_obs; o++) order[(size_t) cursor[(size_t) h_obs_pt[o]]++] = o;
    }
    for (int q = 0; q < n_obs; q++) {
        const int o = order[(size_t) q];
        h_obsKf[q] = h_obs_kf[o];
        h_obsUv[2 * (size_t) q] = h_obs_uv[2 * o];
        h_obsUv[2 * (size_t) q + 1] = h_obs_uv[2 * o + 1];
        h_ptPtr[(size_t) h_obs_pt[o] + 1]++;
        const int anc = inv_depth ? h_pt_anchor_kf[h_obs_pt[o]] : h_obs_kf[o];
        ALVA_ARG(anc >= 0 && anc < n_kf);
        pairKey[(size_t) q] = h_obs_kf[o] * n_kf + anc;
    }
    for (int p2 = 0; p2 < n_pt; p2++) h_ptPtr[(size_t) p2 + 1] += h_ptPtr[(size_t) p2];
    // the same observations grouped by (observing kf, anchor kf) pair, stable: counting sort again
    const size_t nPairs = (size_t) n_kf * n_kf;
    for (size_t i = 0; i <= nPairs; i++) h_pairPtr[i] = 0;
    for (int q = 0; q < n_obs; q++) h_pairPtr[(size_t) pairKey[(size_t) q] + 1]++;
    for (size_t i = 0; i < nPairs; i++) h_pairPtr[i + 1] += h_pairPtr[i];
    {
        std::vector<int> cursor(h_pairPtr, h_pairPtr + nPairs);
        for (int q = 0; q < n_obs; q++) h_pairPerm[(size_t) cursor[(size_t) pairKey[(size_t) q]]++] = q;
    }
    if (inv_depth && n_pt > 0) {
        memcpy(h_ancKf, h_pt_anchor_kf, nPt * 4);
        memcpy(h_ancUv, h_pt_anchor_uv, nPt * 16);
    }
    for (int k = 0; k < n_kf; k++) {
        h_cidx[k] = cidx[(size_t) k];
        h_kfOf[k] = 0;
    }
    for (int k = 0; k < n_kf; k++)
        if (cidx[(size_t) k] >= 0) h_kfOf[cidx[(size_t) k]] = k;
    // poses are stored the way PoseParametersBlock(id, SE3d) stores them: unit quaternion
    for (int k = 0; k < n_kf; k++) {
        Se3 T;
        se3_from_pose7(h_poses + 7 * k, T);
        for (int i = 0; i < 3; i++) h_xp[7 * (size_t) k + i] = T.t[i];
        for (int i = 0; i < 4; i++) h_xp[7 * (size_t) k + 3 + i] = T.q[i];
    }
    if (npd) memcpy(h_xt, h_pt_param, npd * 8);
    const auto t_built = std::chrono::steady_clock::now();
    ALVA_HIP(hipMemcpyAsync(base, stage, in_bytes, hipMemcpyHostToDevice, st));   // ONE upload from pinned memory
    ALVA_HIP(hipMemsetAsync(B.Wt, 0, npd * NP * 8, st));                     // sparsity pattern is fixed: zero once
    ALVA_HIP(hipMemsetAsync(B.Zt, 0, (size_t) B.kpad * NP * 8, st));         // K padding rows stay zero
    const auto t_up = std::chrono::steady_clock::now();

    const dim3 gPt((unsigned) alva_divup(std::max(n_pt, 1), 4)), blk(256);
    const size_t np16 = (n6 + 15) / 16 * 16, solve_lds = (np16 * (np16 + 1) + np16) * sizeof(double);
    const bool solve_in_lds = solve_lds <= 152 * 1024;   // + ~5 KB of static LDS in k_solve stays under the CU's 160 KB
    if (solve_in_lds && solve_lds > 48 * 1024)
        ALVA_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_solve<true>), hipFuncAttributeMaxDynamicSharedMemorySize, 152 * 1024));
    // cost, Jacobian blocks, Schur products and gradient at (xp, xt); every evaluation carries its Jacobian (see the loop below)
    auto eval = [&](const double *xp, const double *xt, bool first) -> int {
        if (n_pt > 0) {
            if (inv_depth) hipLaunchKernelGGL((k_point<true, true>), gPt, blk, 0, st, B, xp, xt);
            else hipLaunchKernelGGL((k_point<false, true>), gPt, blk, 0, st, B, xp, xt);
        }
        hipLaunchKernelGGL(k_pairs, dim3((unsigned) (n_kf * n_kf)), blk, 0, st, B);
        hipLaunchKernelGGL(k_rowcol, dim3((unsigned) n_kf), dim3(64), 0, st, B);
        if (B.n6 > 0) hipLaunchKernelGGL(k_hcc, dim3((unsigned) alva_divup(B.n6 * B.n6 + B.n6, 256)), blk, 0, st, B, first ? 1 : 0);
        hipLaunchKernelGGL(k_gmax, dim3(1), blk, 0, st, B, first ? 1 : 0);  // also sums the cost
        ALVA_LAUNCH_CHECK();
        return ALVA_OK;
    };
    double scal[8];
    double *pin_scal = reinterpret_cast<double *>(pin);
    auto read_scal = [&]() -> int {
        ALVA_HIP(hipMemcpyAsync(pin_scal, B.scal, sizeof(scal), hipMemcpyDeviceToHost, st));  // pinned: a plain DMA, no staging
        ALVA_HIP(hipStreamSynchronize(st));
        memcpy(scal, pin_scal, sizeof(scal));
        return ALVA_OK;
    };

    // ---- Ceres TrustRegionMinimizer::Minimize, restated (trust_region_minimizer.cc:67-136) -------------------
    rc = eval(d_xp, d_xt, true);
    if (rc) return rc;
    rc = read_scal();
    if (rc) return rc;
    double x_cost = scal[0], gmax = scal[3], x_norm = -1, initial = x_cost;
    LmState lm;
    int iteration = 0, nsucc = 1, invalid = 0, nsummaries = 1;
    double *xp = d_xp, *xt = d_xt, *cp = d_cp, *ct = d_ct;
    bool need_restore = false;  // the Jacobian-derived state belongs to a rejected candidate (restored lazily: when the loop ends right
                                // after a rejection, the last evaluation stays the candidate's, as in the reference)
    while (true) {
        if (iteration >= max_iters || gmax <= 1e-10 || lm.radius <= 1e-32) break;
        iteration++;
        if (need_restore) {
            rc = eval(xp, xt, false);
            if (rc) return rc;
            need_restore = false;
        }
        const int ndiag = std::max(B.n6, B.npd);
        if (!lm.reuse_diagonal && ndiag > 0) hipLaunchKernelGGL(k_diag, dim3((unsigned) alva_divup(ndiag, 256)), blk, 0, st, B);
        lm.reuse_diagonal = 1;
        if (n_pt > 0) {
            if (dp == 1) hipLaunchKernelGGL(k_prep<1>, gPt, blk, 0, st, B, lm.radius);
            else hipLaunchKernelGGL(k_prep<3>, gPt, blk, 0, st, B, lm.radius);
        }
        const int tiles = B.NP / 16;
        hipLaunchKernelGGL(k_gemm, dim3((unsigned) (tiles * tiles), KSPLIT), dim3(64), 0, st, B);
        if (np16 > 0) {
            const int np16i = (int) np16;
            hipLaunchKernelGGL(k_reduced_system, dim3((unsigned) alva_divup(np16i * np16i + np16i, 256)), blk, 0, st, B, lm.radius);
        }
        if (solve_in_lds) hipLaunchKernelGGL(k_solve<true>, dim3(1), dim3(SOLVE_NT), solve_lds, st, B, lm.radius);
        else hipLaunchKernelGGL(k_solve<false>, dim3(1), dim3(SOLVE_NT), 0, st, B, lm.radius);
        if (n_pt > 0) {
            if (dp == 1) hipLaunchKernelGGL(k_backsub<1>, gPt, blk, 0, st, B, lm.radius, (const double *) xt, ct);
            else hipLaunchKernelGGL(k_backsub<3>, gPt, blk, 0, st, B, lm.radius, (const double *) xt, ct);
        }
        hipLaunchKernelGGL(k_update, dim3(1), blk, 0, st, B, lm.radius, (const double *) xp, cp, (const double *) ct);
        ALVA_LAUNCH_CHECK();
        // The candidate is evaluated WITH its Jacobian and its norm straight away: when the step is accepted (the normal case) Ceres
        // re-evaluates at the same point (HandleSuccessfulStep, trust_region_minimizer.cc:809-829) and would produce exactly these
        // numbers again, so one host round trip per iteration disappears.  A rejected step costs one re-evaluation at x instead.
        rc = eval(cp, ct, false);
        if (rc) return rc;
        rc = read_scal();
        if (rc) return rc;
        const double cand_cost = scal[0], mcc = scal[1], step_norm = std::sqrt(scal[2]);
        const bool okstep = scal[5] != 0.0 && std::isfinite(mcc);
        if (!okstep || !(mcc > 0)) {  // HandleInvalidStep (:461-490)
            if (++invalid >= 5) {
                *h_ok = 0;
                break;
            }
            lm.rejected();
            nsummaries++;
            need_restore = true;
            continue;
        }
        invalid = 0;
        if (step_norm <= 1e-8 * (x_norm + 1e-8)) break;                                // ParameterToleranceReached
        if (std::fabs(x_cost - cand_cost) <= function_tolerance * x_cost) break;      // FunctionToleranceReached
        const double rel = (x_cost - cand_cost) / mcc;
        if (rel > 1e-3) {
            std::swap(xp, cp);
            std::swap(xt, ct);
            x_cost = cand_cost;
            gmax = scal[3];
            x_norm = std::sqrt(scal[4]);
            lm.accepted(rel);
            nsucc++;
        } else {
            lm.rejected();
            need_restore = true;
        }
        nsummaries++;
    }
    const auto t_lm = std::chrono::steady_clock::now();
    // results: poses / points at the last accepted x; chi2 / depth flags of the LAST evaluation (what the
    // reference's outlier sweep reads from its cost-function objects, optimizer.cpp:266-309)
    // the input staging area is free again (its upload finished long ago): results land there, three DMA copies
    uint8_t *r_chi = stage;
    const size_t chi_bytes = (size_t) ((uint8_t *) B.depth - (uint8_t *) B.chi2) + nObs;   // chi2 | pad | depth, as carved
    double *r_poses = reinterpret_cast<double *>(stage + (chi_bytes + 255) / 256 * 256);
    double *r_pts = r_poses + (size_t) n_kf * 7 + 32;
    if (n_obs) ALVA_HIP(hipMemcpyAsync(r_chi, B.chi2, chi_bytes, hipMemcpyDeviceToHost, st));
    ALVA_HIP(hipMemcpyAsync(r_poses, xp, (size_t) n_kf * 56, hipMemcpyDeviceToHost, st));
    if (npd) ALVA_HIP(hipMemcpyAsync(r_pts, xt, npd * 8, hipMemcpyDeviceToHost, st));
    ALVA_HIP(hipStreamSynchronize(st));
    if (npd) memcpy(h_pt_param, r_pts, npd * 8);
    const double *chi2s = reinterpret_cast<const double *>(r_chi);
    const uint8_t *deps = r_chi + ((uint8_t *) B.depth - (uint8_t *) B.chi2);
    for (int k = 0; k < n_kf; k++)
        if (cidx[(size_t) k] >= 0) memcpy(h_poses + 7 * k, r_poses + 7 * (size_t) k, 56);
    for (int q = 0; q < n_obs; q++) {  // back to the caller's observation order
        if (h_chi2) h_chi2[order[(size_t) q]] = chi2s[(size_t) q];
        if (h_depth_pos) h_depth_pos[order[(size_t) q]] = deps[(size_t) q];
    }
    if (h_info) {
        h_info[0] = nsummaries;
        h_info[1] = initial;
        h_info[2] = x_cost;
        h_info[3] = nsucc;
    }
    if (getenv("ALVA_BA_TIMING")) {
        auto us = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) {
            return (double) std::chrono::duration_cast<std::chrono::nanoseconds>(b - a).count() * 1e-3;
        };
        fprintf(stderr, "[alva_local_ba] host structure %.0f us | scratch + upload %.0f us | LM loop (%d summaries) %.0f us | download %.0f us\n",
                us(t_begin, t_built), us(t_built, t_up), nsummaries, us(t_up, t_lm), us(t_lm, std::chrono::steady_clock::now()));
    }
    return ALVA_OK;
}
