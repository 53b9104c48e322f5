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
//   k_assemble   F'F, F'r, Jacobi column scaling (iteration 0), gradient max-norm, total cost, the step's scalars (one workgroup).
// Per LM step:
//   k_prep    per point: (E'E + D^2)^-1, its Cholesky factor L, Z_p = S_c W_p S_p L  and v_p = L'(E'r)
//   k_gemm    G = Z' Z with v appended as one more column: ONE dense [6*cams+1 x points*d]^2 FP64 GEMM on
//             v_mfma_f64_16x16x4_f64 (the only GEMM-shaped piece of the whole path) --
//             S = F'F + D^2 - G,  rhs = F'r - G[:, last]
//   k_reduced_system   S and the right-hand side, padded to a multiple of 16
//   k_solve   blocked dense Cholesky of the reduced camera system in one workgroup + blocked triangular solves
//   k_backsub per point: y_p, candidate point; model-cost-change and step-norm partials
//             (+ one workgroup for the candidate poses and the camera part of model cost change / step norm / candidate norm)
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
    double *h_scal; // single problem: pinned host mirror of scal[0..7] + the evaluation's sequence number at [8] (k_assemble publishes it last,
                    // system-scope release; the host polls it instead of a copy command + stream wait per LM iteration); null in a batch
    long long seq;
    double *partial;  // [nPt][3] per-point partials for mcc / step norm / x norm
    unsigned long long *dbg;   // phase stamps of k_solve (alva_kstamp_buffer, entries 3072..) or null
};
#define SOLVE_STAMP(k) do { if (B.dbg && threadIdx.x == 0) B.dbg[3072 + (k)] = wall_clock64(); } while (0)
#define ASM_STAMP(k) do { if (B.dbg && threadIdx.x == 0) B.dbg[3072 + 48 + (k)] = wall_clock64(); } while (0)   // k_assemble's phases (tools/solve_stamps.py)

constexpr int KSPLIT = 8;

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off);
    return __shfl(v, 0);
}

// ------------------------------------------------------------------------------------------------------
template<bool INV, bool WANT_J>
__device__ __forceinline__ void point_body(const BaDev &B, const double *__restrict__ poses, const double *__restrict__ pts, const int BX_, const int BY_) {
    const int lane = threadIdx.x & 63;
    const int p = BX_ * 4 + (threadIdx.x >> 6);
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
template<bool INV, bool WANT_J>
__global__ void __launch_bounds__(256) k_point(BaDev B, const double *__restrict__ poses, const double *__restrict__ pts) {
    point_body<INV, WANT_J>(B, poses, pts, blockIdx.x, blockIdx.y);
}

// One wave per (observing kf, anchor kf) pair: sums of J_obs'J_obs (21) and J_obs' r (6).
__device__ __forceinline__ void pairs_body(const BaDev &B, const int BX_, const int BY_) {
    // one WORKGROUP per (observing kf, anchor kf) pair: the big pairs (thousands of observations) set the kernel time, so
    // their observations are spread over 256 lanes; lane partials -> butterfly reduce-scatter per wave -> 4 waves in LDS
    // (fixed order: bit-reproducible)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int key = BX_;
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
__global__ void __launch_bounds__(256) k_pairs(BaDev B) {
    pairs_body(B, blockIdx.x, blockIdx.y);
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
// Assembly of the camera block H_cc / g_c from the per-pair sums:
// ONE workgroup (1024 threads) per problem does all three steps -- they used to be three launches (k_rowcol, k_hcc, k_gmax: 8 + 6.5 + 12 us
// of kernels and two launch gaps per evaluation for ~20 k loads and 12 k stores):
//   1. per keyframe the row and column sums of the pair sums, into LDS (fixed order over the keyframes)
//   2. every element of H_cc and g_c (+ the Jacobi scaling of the cameras at the first evaluation); g_c also into LDS
//   3. the evaluation's scalars: total cost, max |gradient| (+ the points' Jacobi scaling at the first evaluation), and the step's model
//      cost change / squared step norm / squared candidate norm from the per-point partials of k_backsub and the camera part its pose
//      workgroup left behind the partials -- then the scalars are published to the host (single problem)
constexpr int ASM_NT = 1024;
static inline size_t assemble_lds(int n_kf) { return (size_t) 2 * n_kf * 27 * sizeof(double); }   // row | column sums (dynamic LDS)
__device__ __forceinline__ void assemble_body(const BaDev &B, int first) {
    extern __shared__ double s_rc[];
    __shared__ double s_red[5][ASM_NT / 64];
    __shared__ int s_kfof[64];   // free camera -> keyframe (a dependent global load in front of every pair-sum load otherwise)
    const int nKf = B.nKf, n6 = B.n6, tid = threadIdx.x;
    ASM_STAMP(0);
    if (tid < B.nc && tid < 64) s_kfof[tid] = B.kfOf[tid];   // (visible after the barrier behind the row / column sums)
    double *rowsum = s_rc, *colsum = s_rc + (size_t) nKf * 27;
    for (int e = tid; e < 2 * nKf * 27; e += ASM_NT) {
        const bool col = e >= nKf * 27;
        const int q = col ? e - nKf * 27 : e, k = q / 27, t = q - 27 * k;
        // sum over the keyframes IN ORDER, eight loads in flight at a time (one dependent load per add ran at one L2 round trip per term)
        const size_t stride = col ? (size_t) nKf * 27 : 27, first_e = col ? (size_t) k * 27 + t : (size_t) k * nKf * 27 + t;
        double acc = 0;
        for (int j0 = 0; j0 < nKf; j0 += 8) {
            double v[8];
#pragma unroll
            for (int u = 0; u < 8; u++) v[u] = j0 + u < nKf ? B.M[first_e + (size_t) (j0 + u) * stride] : 0.0;
#pragma unroll
            for (int u = 0; u < 8; u++)
                if (j0 + u < nKf) acc += v[u];
        }
        s_rc[e] = acc;
    }
    __syncthreads();
    ASM_STAMP(1);
    double gm = 0;
    {
        // H_cc row by row: wave w takes rows w, w + 16, ..., lane l the columns l, l + 64, ... -- no division by the runtime size (a
        // flat element index cost ~150 instructions of index arithmetic per element, and 16 waves share the compute unit's four SIMDs:
        // the kernel was bound by that).  Two rows x two column chunks per trip: their (up to eight) pair-sum loads are issued
        // together, then the stores (which may alias the loads as far as the compiler knows).
        const double *__restrict__ M = B.M;
        const int lane = tid & 63, wave = tid >> 6, NW = ASM_NT / 64;
        const bool lds_map = B.nc <= 64;
        for (int r0 = wave; r0 < n6; r0 += 2 * NW) {
            for (int c0 = lane; c0 < n6; c0 += 128) {
                double v[4];
                int er[4], ec[4];
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    const int r = r0 + (u >> 1) * NW, c = c0 + (u & 1) * 64;
                    er[u] = r; ec[u] = c;
                    v[u] = 0;
                    if (r < n6 && c < n6) {
                        const int cr = r / 6, cc = c / 6, x = r - 6 * cr, y = c - 6 * cc;
                        const int kr = lds_map ? s_kfof[cr] : B.kfOf[cr], kc = lds_map ? s_kfof[cc] : B.kfOf[cc];
                        const int t = tri6(x, y);
                        if (B.inv) {
                            if (kr == kc) v[u] = rowsum[kr * 27 + t] + colsum[kr * 27 + t];
                            else v[u] = -(M[(size_t) (kr * nKf + kc) * 27 + t] + M[(size_t) (kc * nKf + kr) * 27 + t]);
                        } else {
                            if (kr == kc) v[u] = M[(size_t) (kr * nKf + kr) * 27 + t];
                        }
                    }
                }
#pragma unroll
                for (int u = 0; u < 4; u++)
                    if (er[u] < n6 && ec[u] < n6) {
                        B.Hcc[(size_t) er[u] * n6 + ec[u]] = v[u];
                        if (first && er[u] == ec[u]) B.sc[er[u]] = 1.0 / (1.0 + sqrt(v[u]));
                    }
            }
        }
        for (int r = tid; r < n6; r += ASM_NT) {
            const int cr = r / 6, x = r - 6 * cr, kr = lds_map ? s_kfof[cr] : B.kfOf[cr];
            const double g = B.inv ? rowsum[kr * 27 + 21 + x] - colsum[kr * 27 + 21 + x] : M[(size_t) (kr * nKf + kr) * 27 + 21 + x];
            B.gc[r] = g;
            gm = fmax(gm, fabs(g));
        }
    }
    ASM_STAMP(2);
    if (first)
        for (int i = tid; i < B.npd; i += ASM_NT) {
            const int p = i / B.dp, x = i % B.dp;
            B.sp[i] = 1.0 / (1.0 + sqrt(B.Hpp[(size_t) p * B.dp * B.dp + x * B.dp + x]));
        }
    double cost = 0, mcc = 0, sn = 0, xn = 0;
    for (int i = tid; i < B.npd; i += ASM_NT) gm = fmax(gm, fabs(B.gp[i]));
    for (int p = tid; p < B.nPt; p += ASM_NT) {
        cost += B.ptCost[p];
        mcc += B.partial[3 * (size_t) p];
        sn += B.partial[3 * (size_t) p + 1];
        xn += B.partial[3 * (size_t) p + 2];
    }
    // wave totals (xor butterfly: the same value on every lane), then the 16 wave totals in wave order: a fixed summation tree
    const int lane = tid & 63, wave = tid >> 6;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        gm = fmax(gm, __shfl_xor(gm, off));
        cost += __shfl_xor(cost, off);
        mcc += __shfl_xor(mcc, off);
        sn += __shfl_xor(sn, off);
        xn += __shfl_xor(xn, off);
    }
    if (lane == 0) {
        s_red[0][wave] = gm; s_red[1][wave] = cost; s_red[2][wave] = mcc; s_red[3][wave] = sn; s_red[4][wave] = xn;
    }
    __syncthreads();
    ASM_STAMP(3);
    if (tid == 0) {
        double g2 = 0, c2 = 0, m2 = 0, s2 = 0, x2 = 0;
        for (int w = 0; w < ASM_NT / 64; w++) {
            g2 = fmax(g2, s_red[0][w]);
            c2 += s_red[1][w]; m2 += s_red[2][w]; s2 += s_red[3][w]; x2 += s_red[4][w];
        }
        const double *cam = B.partial + 3 * (size_t) B.nPt;   // camera part of the step's scalars (k_backsub's pose workgroup)
        B.scal[0] = c2;
        B.scal[3] = g2;
        B.scal[1] = m2 + cam[0];
        B.scal[2] = s2 + cam[1];
        B.scal[4] = x2 + cam[2];
        if (B.h_scal) {   // scal[5] (Cholesky ok) was written by k_solve of this stream: re-read here, published by THIS kernel
            B.h_scal[0] = c2; B.h_scal[1] = m2 + cam[0]; B.h_scal[2] = s2 + cam[1]; B.h_scal[3] = g2;
            B.h_scal[4] = x2 + cam[2]; B.h_scal[5] = B.scal[5]; B.h_scal[6] = B.scal[6]; B.h_scal[7] = B.scal[7];
            __threadfence_system();
            __hip_atomic_store(reinterpret_cast<long long *>(B.h_scal + 8), B.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}
__global__ void __launch_bounds__(ASM_NT) k_assemble(BaDev B, int first) { assemble_body(B, first); }

// The LM diagonal (levenberg_marquardt_strategy.cc:79-90) is refreshed only after an accepted step (`refresh`): the points' part by k_prep
// (each point its own entries), the cameras' part by k_reduced_system (the thread of each diagonal element) -- no launch of its own.
// per point: hinv = (S_p E'E S_p + D_p^2/radius)^-1 , L = chol(hinv), Zt rows = S_c W S_p L, last column v = L' (S_p E'r)
template<int DP>
__device__ __forceinline__ void prep_body(const BaDev &B, double radius, int refresh, const int BX_, const int BY_) {
    const int lane = threadIdx.x & 63;
    const int p = BX_ * 4 + (threadIdx.x >> 6);
    if (p >= B.nPt) return;
    double Mx[DP * DP], Hi[DP * DP], L[DP * DP], gs[DP], dpd[DP];
#pragma unroll
    for (int x = 0; x < DP; x++) {
        const int i = p * DP + x;
        if (refresh) {
            dpd[x] = fmin(fmax(B.Hpp[(size_t) p * DP * DP + x * DP + x] * B.sp[i] * B.sp[i], 1e-6), 1e32);
            if (lane == 0) B.dpd[i] = dpd[x];
        } else {
            dpd[x] = B.dpd[i];
        }
    }
#pragma unroll
    for (int x = 0; x < DP; x++) {
        gs[x] = B.gp[(size_t) p * DP + x] * B.sp[p * DP + x];
#pragma unroll
        for (int y = 0; y < DP; y++)
            Mx[x * DP + y] = B.Hpp[(size_t) p * DP * DP + x * DP + y] * B.sp[p * DP + x] * B.sp[p * DP + y] + (x == y ? dpd[x] / radius : 0.0);
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
template<int DP>
__global__ void __launch_bounds__(256) k_prep(BaDev B, double radius, int refresh) {
    prep_body<DP>(B, radius, refresh, blockIdx.x, blockIdx.y);
}

// G_part[ks] (16x16 tile) = Zt[kchunk]' Zt[kchunk] on the FP64 matrix core.
// v_mfma_f64_16x16x4_f64: A[l&15][k=l>>4], B[k=l>>4][l&15], C/D col = l&15, row = (l>>4) + 4*reg.
__device__ __forceinline__ void gemm_body(const BaDev &B, const int BX_, const int BY_) {
    const int tiles = B.NP / 16;
    const int ti = BX_ / tiles, tj = BX_ % tiles, ks = BY_;
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
__global__ void __launch_bounds__(64) k_gemm(BaDev B) {
    gemm_body(B, blockIdx.x, blockIdx.y);
}

// S = S_c F'F S_c + D_c^2/radius - G ; rhs = S_c F'r - G[:, n6] ; dense Cholesky; y_c.  One workgroup.
//
// Blocked right-looking Cholesky, block 16, the matrix padded with an identity to a multiple of 16 (no edge cases):
//   diagonal block   wave 0, one row per lane IN REGISTERS, pivots / multipliers broadcast with v_readlane (no LDS, no barrier)
//   panel            one thread per row below: x L11' = a, 136 FMAs against the (broadcast-read) diagonal block
//   trailing update  A22 -= L21 L21' as 16 x 16 tiles on v_mfma_f64_16x16x4_f64, one wavefront per tile
// = 3 barriers per 16 columns instead of 3 per column; the triangular solves are blocked the same way.  Row stride is
// odd (padded size + 1) so that threads reading different rows hit different LDS banks.
#ifndef ALVA_SOLVE_LOOKAHEAD
#define ALVA_SOLVE_LOOKAHEAD 1
#endif
constexpr int SOLVE_NT = 256, NB = 16;   // measured and dropped: 1024 threads (16 waves for the trailing update's tiles): 80 vs 67 us, the
                                        // barriers cost more than the tiles save; triangular solves through explicit inverses of the diagonal
                                        // blocks' factors (computed by a spare wave beside the panel): 91 us, the inverse is a longer chain
                                        // than the panel it hides behind

// reduced camera system of this LM step, padded: S [np][np + 1] and the right-hand side [np] right behind it (all CUs)
__device__ __forceinline__ void reduced_system_body(const BaDev &B, double radius, int refresh, const int BX_, const int BY_) {
    const int n = B.n6, np = (n + NB - 1) / NB * NB, ld = np + 1;
    const int e = BX_ * 256 + threadIdx.x;
    if (e < np * np) {
        const int r = e / np, c = e - r * np;
        double v = r == c ? 1.0 : 0.0;  // identity padding
        if (r < n && c < n) {
            double g = 0;
#pragma unroll
            for (int ks = 0; ks < KSPLIT; ks++) g += B.Gpart[((size_t) ks * B.NP + r) * B.NP + c];
            const double h = B.Hcc[(size_t) r * n + c] * B.sc[r] * B.sc[c];
            v = h - g;
            if (r == c) {
                double dcr = B.dc[r];
                if (refresh) {
                    dcr = fmin(fmax(h, 1e-6), 1e32);
                    B.dc[r] = dcr;
                }
                v += dcr / radius;
            }
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
__global__ void __launch_bounds__(256) k_reduced_system(BaDev B, double radius, int refresh) {
    reduced_system_body(B, radius, refresh, blockIdx.x, blockIdx.y);
}

template<bool IN_LDS>
__device__ __forceinline__ void solve_body(const BaDev &B, double radius, const int BX_, const int BY_) {
    extern __shared__ double s_S[];
    const int n = B.n6, np = (n + NB - 1) / NB * NB, ld = np + 1, nb = np / NB;
    __shared__ double s_z[NB];
    __shared__ int s_ok;
    SOLVE_STAMP(0);
    double *S = IN_LDS ? s_S : B.S;            // [np][ld]
    double *y = IN_LDS ? s_S + (size_t) np * ld : B.S + (size_t) np * ld;  // [np] right-hand side / solution
    // micro-tile enumeration of a lower triangle, row by row: t -> (ta, tb), the same for every trailing size
    constexpr int TILE_LUT = 1024;   // 16 x 16 tiles: covers np <= 736; larger systems decode arithmetically
    __shared__ unsigned short s_tile[TILE_LUT];
    auto fill_lut = [&]() {
        for (int t = threadIdx.x; t < TILE_LUT; t += SOLVE_NT) {
            int ta = (int) ((sqrtf(8.f * (float) t + 1.f) - 1.f) * 0.5f);
            while (ta * (ta + 1) / 2 > t) ta--;
            while ((ta + 1) * (ta + 2) / 2 <= t) ta++;
            s_tile[t] = (unsigned short) ((ta << 8) | (t - ta * (ta + 1) / 2));
        }
    };
    if (IN_LDS) {
        // matrix + right-hand side from k_reduced_system into LDS: a chain of round trips to memory another kernel has just written, so
        // as many loads in flight as the register file allows (32 per thread: two trips for the 100 KB of a 108-unknown system; eight in
        // flight took six trips, 6.1 us, a tenth of the kernel) and the tile table computed while the first trip flies.  Tried and
        // dropped: fetching only the lower triangle (the one half the factorisation reads) -- its row / column arithmetic per word
        // costs what the halved traffic saves, with a packed source as well (6.8 - 7.5 us).
        const int total = np * ld + np;
        bool lut_done = false;
        for (int e0 = threadIdx.x; e0 < total || !lut_done; e0 += 32 * SOLVE_NT) {
            double v[32];
#pragma unroll
            for (int u = 0; u < 32; u++) {
                const int e = e0 + u * SOLVE_NT;
                v[u] = e < total ? B.S[e] : 0.0;
            }
            if (!lut_done) {
                fill_lut();
                lut_done = true;
            }
#pragma unroll
            for (int u = 0; u < 32; u++) {
                const int e = e0 + u * SOLVE_NT;
                if (e < total) s_S[e] = v[u];
            }
        }
    } else {
        fill_lut();
    }
    if (threadIdx.x == 0) s_ok = 1;
    __syncthreads();
    SOLVE_STAMP(1);
    const int lane = threadIdx.x & 63;
    // ---- diagonal block at `base`, by ONE wave: lane i (< 16) owns row i ---------------------------------------------------------
    auto diag_factor = [&](const int base) {
        double a[NB];
        const int row = base + (lane & 15);
#pragma unroll
        for (int k = 0; k < NB; k++) a[k] = S[(size_t) row * ld + base + k];
        int ok = 1;
#pragma unroll
        for (int j = 0; j < NB; j++) {
            const double d = lane_bcast(a[j], j);
            if (!(d > 0)) ok = 0;
            // the pivot's reciprocal square root is the head of every column's dependent chain (and 1 / L_jj of both triangular
            // solves): refined hardware estimate instead of the IEEE-rounded rsqrt / divide (~10 dependent instructions instead of
            // ~30 each, x ~330 pivots per call).  The DIAGONAL of the factor stores 1 / L_jj: the solves multiply.
            double piv;
            const double inv = alva_fast_rsqrt(d > 0 ? d : 1.0, piv);
            a[j] = (lane & 15) == j ? inv : a[j] * inv;  // rows above j hold garbage in column j: never read
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
    };
    // one 16 x 16 tile of the trailing update A22 -= L21 L21' (a rank-16 update = four v_mfma_f64_16x16x4_f64 per tile, one wavefront per
    // tile: operand a: L[rowa + (lane & 15)][4c + (lane >> 4)], operand b the same for rowb; result row (lane >> 4) + 4r, column
    // lane & 15).  Diagonal tiles are updated in full: the strict upper triangle is never read.
    auto trailing_tile = [&](const int base, const int t) {
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
    };
    // The factorisation carries the right-hand side along as ROW np of the matrix (it is stored there): the panel solve of block kb turns
    // its 16 entries into z = L11^-1 y and the trailing update subtracts L21 z from the rest -- the forward substitution L z = rhs costs no
    // pass (and no barriers) of its own.  Look-ahead: while three waves run the trailing update of block kb, the first wave updates the ONE
    // tile the next diagonal block lives in and factors it (stamps before: 2.2 us per diagonal block with three waves waiting).
    const int wave = threadIdx.x >> 6;
    constexpr bool LOOKAHEAD = ALVA_SOLVE_LOOKAHEAD;
    if (LOOKAHEAD && nb > 0) {   // (nb == 0: every camera constant, nothing to factor)
        if (wave == 0) diag_factor(0);
        __syncthreads();
    }
    for (int kb = 0; kb < nb; kb++) {
        const int base = kb * NB;
        SOLVE_STAMP(8 + 4 * kb);
        if (!LOOKAHEAD) {
            if (wave == 0) diag_factor(base);
            __syncthreads();
        }
        if (!s_ok) break;
        // ---- panel: x L11' = a for every row below (and the right-hand side's row) ------------------------------------------------
        // Every row needs all 120 off-diagonal entries and the 16 pivots of L11: the same values for every thread.  Lane j of each wave
        // holds row j of the block in registers and the entries are BROADCAST from there (v_readlane, a scalar operand of the FMA) instead
        // of 136 uniform-address LDS reads per thread (stamps: 2.2 us per panel, LDS-latency bound).
        const int m = np - base - NB;  // rows below the block
        {
            double l[NB];
            const double *lrow = S + (size_t) (base + (lane & 15)) * ld + base;
#pragma unroll
            for (int k = 0; k < NB; k++) {
                l[k] = lrow[k];   // (k > lane & 15: strict upper triangle, garbage, never broadcast)
                // pinned HERE, in every lane: the loop below runs with few lanes (one, for the right-hand side's row of the last
                // blocks) and reads lanes 0..15 whatever their state; a load sunk into the loop would never execute for them
                asm volatile("" : "+v"(l[k]));
            }
            for (int r = threadIdx.x; r <= m; r += SOLVE_NT) {   // r == m: the right-hand side (y = S + np * ld)
                double *rowp = S + (size_t) (base + NB + r) * ld + base;
                double x[NB];
#pragma unroll
                for (int j = 0; j < NB; j++) x[j] = rowp[j];
#pragma unroll
                for (int j = 0; j < NB; j++) {
                    double v = x[j];
#pragma unroll
                    for (int k = 0; k < j; k++) v -= x[k] * lane_bcast(l[k], j);   // L11[j][k]
                    x[j] = v * lane_bcast(l[j], j);                                 // the diagonal holds 1 / L_jj
                }
#pragma unroll
                for (int j = 0; j < NB; j++) rowp[j] = x[j];
            }
        }
        __syncthreads();
        SOLVE_STAMP(9 + 4 * kb);
        // ---- trailing update + the next diagonal block -------------------------------------------------------------------------------
        const int mt = m / NB, ntile = mt * (mt + 1) / 2;
        if (!LOOKAHEAD) {
            for (int t = wave; t < ntile; t += SOLVE_NT / 64) trailing_tile(base, t);
            for (int c = threadIdx.x; c < m; c += SOLVE_NT) {
                const double *lr = S + (size_t) (base + NB + c) * ld + base;
                double v = y[base + NB + c];
#pragma unroll
                for (int k = 0; k < NB; k++) v -= lr[k] * y[base + k];
                y[base + NB + c] = v;
            }
        } else if (wave == 0) {
            if (ntile > 0) {
                trailing_tile(base, 0);            // tile (0, 0): the next diagonal block
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");   // other lanes' stores of the tile, before this wave reads it back
                __builtin_amdgcn_wave_barrier();
                diag_factor(base + NB);
            }
        } else {
            for (int t = wave; t < ntile; t += SOLVE_NT / 64 - 1) trailing_tile(base, t);
            // the right-hand side's part of the update: y[c] -= L21[c][:] . z
            for (int c = threadIdx.x - 64; c < m; c += SOLVE_NT - 64) {
                const double *lr = S + (size_t) (base + NB + c) * ld + base;
                double v = y[base + NB + c];
#pragma unroll
                for (int k = 0; k < NB; k++) v -= lr[k] * y[base + k];
                y[base + NB + c] = v;
            }
        }
        __syncthreads();
    }
    SOLVE_STAMP(2);
    if (s_ok) {
        SOLVE_STAMP(3);
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
                    const double yj = lane_bcast(zi, j) * lane_bcast(lc[j], j);
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
        SOLVE_STAMP(4);
        for (int r = threadIdx.x; r < n; r += SOLVE_NT) B.yc[r] = y[r];
    }
    if (threadIdx.x == 0) B.scal[5] = s_ok ? 1.0 : 0.0;
    SOLVE_STAMP(5);
}
template<bool IN_LDS>
__global__ void __launch_bounds__(SOLVE_NT) k_solve(BaDev B, double radius) {
    solve_body<IN_LDS>(B, radius, blockIdx.x, blockIdx.y);
}

// per point: y_p = hinv (g_s - (S_c W S_p)' y_c); candidate point; partials for the model cost change
// (1/2 y'(g_s + D y), exact for the exact solution of (H_s + D) y = g_s) and the step norm.
__device__ __forceinline__ void pose_update_body(const BaDev &B, double radius, const double *__restrict__ x_p, double *__restrict__ c_p);
template<int DP>
__device__ __forceinline__ void backsub_body(const BaDev &B, double radius, const double *__restrict__ x_t, double *__restrict__ c_t,
                                             const double *__restrict__ x_p, double *__restrict__ c_p, const int BX_, const int BY_) {
    const int lane = threadIdx.x & 63;
    if (BX_ == (B.nPt + 3) / 4) {   // one workgroup behind the points': the candidate poses and the camera part of the step's scalars
        pose_update_body(B, radius, x_p, c_p);
        return;
    }
    const int p = BX_ * 4 + (threadIdx.x >> 6);
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
        double mcc = 0, sn = 0, xn = 0;
#pragma unroll
        for (int x = 0; x < DP; x++) {
            double y = 0;
#pragma unroll
            for (int z = 0; z < DP; z++) y += B.hinv[(size_t) p * DP * DP + x * DP + z] * t[z];
            B.yp[p * DP + x] = y;
            const double gs = B.gp[(size_t) p * DP + x] * B.sp[p * DP + x];
            mcc += 0.5 * y * (gs + B.dpd[p * DP + x] / radius * y);
            const double d = -y * B.sp[p * DP + x];
            const double cv = x_t[p * DP + x] + d;
            c_t[p * DP + x] = cv;
            sn += d * d;
            xn += cv * cv;
        }
        B.partial[3 * (size_t) p] = mcc;
        B.partial[3 * (size_t) p + 1] = sn;
        B.partial[3 * (size_t) p + 2] = xn;
    }
}
template<int DP>
__global__ void __launch_bounds__(256) k_backsub(BaDev B, double radius, const double *__restrict__ x_t, double *__restrict__ c_t,
                                                 const double *__restrict__ x_p, double *__restrict__ c_p) {
    backsub_body<DP>(B, radius, x_t, c_t, x_p, c_p, blockIdx.x, blockIdx.y);
}

// candidate poses (SE3 Plus) and the CAMERA part of the step's scalars -- model cost change, squared step norm, squared norm of the
// candidate -- left behind the per-point partials (B.partial[3 nPt ..]); k_assemble of the candidate's evaluation adds the points' part
// and publishes them.  One workgroup (the extra one of k_backsub); a problem has a few tens of keyframes.
__device__ __forceinline__ void pose_update_body(const BaDev &B, double radius, const double *__restrict__ x_p, double *__restrict__ c_p) {
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
        double *cam = B.partial + 3 * (size_t) B.nPt;
        cam[0] = s_a[0];
        cam[1] = s_b[0];
        cam[2] = s_c[0];
    }
}

// |x|^2 over the variable blocks (free poses in their 7-vector form + all point parameters) -> scal[4]

// The results of a single solve straight into pinned host memory -- chi2 | depth flags of the last evaluation, the poses and the
// point parameters at the last accepted x -- instead of three copy commands (~20 us each): every workgroup copies its share with 8-byte
// stores, fences at system scope and arrives on a device counter; the last one publishes `seq` in the completion word the host polls.
struct BaResultsArgs {
    const unsigned long long *src[3];
    unsigned long long *dst[3];
    size_t words[3];
    int *counter;          // zero between launches
    long long *word;       // pinned
    long long seq;
};
__global__ void __launch_bounds__(256) k_results(BaResultsArgs R) {
    const size_t stride = (size_t) gridDim.x * 256, t0 = (size_t) blockIdx.x * 256 + threadIdx.x;
#pragma unroll
    for (int a = 0; a < 3; a++)
        for (size_t i = t0; i < R.words[a]; i += stride) R.dst[a][i] = R.src[a][i];
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) {
        const int arrived = __hip_atomic_fetch_add(R.counter, 1, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        if (arrived == (int) gridDim.x - 1) {
            *R.counter = 0;
            // every workgroup's words are behind its own system-scope fence + arrival: the completion word is one store, no second fence
            __hip_atomic_store(R.word, R.seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

// ---- the (observing keyframe, anchor keyframe) grouping of the observations ON THE DEVICE (alva_local_ba_csr): a stable counting sort
// over chunks of 256 observations -- per-chunk histograms, one scan, per-chunk scatter with the rank inside the chunk -- so that
// pairPerm / pairPtr are exactly what BaHost::build's two host passes produce (ascending observation index inside a pair), and the host
// never walks the observations for it
constexpr int PAIR_CHUNK = 1024;   // (256 at first: 110 chunks for 28 k observations, and the scan below -- one thread per pair walking the chunks'
                                   // histograms through dependent global loads -- took 43 us per solve; 28 chunks, read eight at a time: a few us)
struct PairArgs {
    const int *obsKf, *ptPtr, *ancKf;
    int nObs, nPt, nKf, nChunks;
    int *key;        // [nObs]
    int *hist;       // [nChunks][nKf * nKf] -> exclusive bases
    int *pairPtr;    // [nKf * nKf + 1]
    int *pairPerm;   // [nObs]
};
__global__ void __launch_bounds__(PAIR_CHUNK) k_pair_hist(PairArgs A) {
    extern __shared__ int s_hist[];
    const int nPairs = A.nKf * A.nKf;
    for (int i = threadIdx.x; i < nPairs; i += PAIR_CHUNK) s_hist[i] = 0;
    __syncthreads();
    const int q = blockIdx.x * PAIR_CHUNK + threadIdx.x;
    if (q < A.nObs) {
        int lo = 0, hi = A.nPt;   // the point of observation q: the last p with ptPtr[p] <= q
        while (hi - lo > 1) {
            const int mid = (lo + hi) >> 1;
            if (A.ptPtr[mid] <= q) lo = mid;
            else hi = mid;
        }
        const int key = A.obsKf[q] * A.nKf + A.ancKf[lo];
        A.key[q] = key;
        atomicAdd(&s_hist[key], 1);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < nPairs; i += PAIR_CHUNK) A.hist[(size_t) blockIdx.x * nPairs + i] = s_hist[i];
}
__global__ void __launch_bounds__(1024) k_pair_scan(PairArgs A) {
    __shared__ int s_tot[1024];
    const int nPairs = A.nKf * A.nKf, k = threadIdx.x;
    int tot = 0;
    if (k < nPairs)
        for (int c0 = 0; c0 < A.nChunks; c0 += 8) {   // eight independent loads in flight, then their running sum
            int v[8];
#pragma unroll
            for (int j = 0; j < 8; j++) v[j] = c0 + j < A.nChunks ? A.hist[(size_t) (c0 + j) * nPairs + k] : 0;
#pragma unroll
            for (int j = 0; j < 8; j++)
                if (c0 + j < A.nChunks) {
                    A.hist[(size_t) (c0 + j) * nPairs + k] = tot;
                    tot += v[j];
                }
        }
    s_tot[k] = k < nPairs ? tot : 0;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {   // inclusive scan of the per-key totals
        const int v = k >= off ? s_tot[k - off] : 0;
        __syncthreads();
        s_tot[k] += v;
        __syncthreads();
    }
    if (k < nPairs) {
        const int base = s_tot[k] - tot;
        A.pairPtr[k] = base;
        if (k == nPairs - 1) A.pairPtr[nPairs] = s_tot[k];
        for (int c = 0; c < A.nChunks; c++) A.hist[(size_t) c * nPairs + k] += base;   // (independent read-modify-writes: they pipeline)
    }
}
__global__ void __launch_bounds__(PAIR_CHUNK) k_pair_fill(PairArgs A) {
    __shared__ __attribute__((aligned(16))) int s_key[PAIR_CHUNK];
    const int q = blockIdx.x * PAIR_CHUNK + threadIdx.x, nPairs = A.nKf * A.nKf;
    const int key = q < A.nObs ? A.key[q] : -1;
    s_key[threadIdx.x] = key;
    __syncthreads();
    if (q >= A.nObs) return;
    int rank = 0;
    const int t4 = (int) threadIdx.x & ~3;
    for (int t = 0; t < t4; t += 4) {   // the chunk's earlier observations with the same pair, four LDS words at a time
        const int4 k4 = *reinterpret_cast<const int4 *>(&s_key[t]);
        rank += (k4.x == key) + (k4.y == key) + (k4.z == key) + (k4.w == key);
    }
    for (int t = t4; t < (int) threadIdx.x; t++) rank += s_key[t] == key;
    A.pairPerm[A.hist[(size_t) blockIdx.x * nPairs + key] + rank] = q;
}
// the outlier sweep's per-observation test (optimizer.cpp:266-309: chi2 above the threshold, or the point behind the camera) as a bit
// per observation, so that the host reads 1 / 72 of the chi2 | depth block
__global__ void __launch_bounds__(256) k_bad_bits(const double *__restrict__ chi2, const uint8_t *__restrict__ depth, int n, double threshold,
                                                 unsigned long long *__restrict__ bits) {
    const int q = blockIdx.x * 256 + threadIdx.x;
    const bool bad = q < n && (chi2[q] > threshold || !depth[q]);
    const unsigned long long m = __ballot(bad);
    if ((threadIdx.x & 63) == 0 && q < n) bits[q >> 6] = m;
}

template<typename T>
T *carve(uint8_t *&cur, size_t count) {
    T *p = reinterpret_cast<T *>(cur);
    cur += (count * sizeof(T) + 255) / 256 * 256;
    return p;
}

// ---- a batch of problems: every kernel with the problem in blockIdx.z (blockIdx.y for the ones that use one grid dimension), the
// per-problem description (BaDev) and this iteration's run parameters (BaRun) in device memory.  The bodies are the single-problem
// kernels' bodies, so a problem's result is bit-identical to its own alva_local_ba.
struct BaRun {
    double radius;
    const double *xp, *xt;  // the accepted point
    double *cp, *ct;        // the candidate
    // evaluation modes: 0 = first evaluation at x (every problem) | 1 = restore evaluation at x (the previous step was rejected) |
    // 2 = evaluation at the candidate (problems that take part in this LM iteration).  Where to evaluate and whether, per mode, as
    // plain tables: a three-way branch that selected among xp / cp here was miscompiled by hipcc 7.2 (the mode-2 arm left both
    // pointers undefined), an indexed load cannot be.
    const double *ev_p[3], *ev_t[3];
    int ev_on[3];
    int step;               // take part in this LM iteration
    int diag;               // refresh the LM diagonal first
    int pad;
};
static_assert(sizeof(BaRun) % 16 == 0, "BaRun array stride");
#define RUN_SEL(R, mode, p, t)            \
    if (!(R).ev_on[mode]) return;         \
    const double *p = (R).ev_p[mode], *t = (R).ev_t[mode];
__global__ void __launch_bounds__(256) k_point_b(const BaDev *Bs, const BaRun *Rs, int mode) {
    RUN_SEL(Rs[blockIdx.y], mode, p, t)
    point_body<true, true>(Bs[blockIdx.y], p, t, blockIdx.x, 0);
}
__global__ void __launch_bounds__(256) k_pairs_b(const BaDev *Bs, const BaRun *Rs, int mode) {
    if (!Rs[blockIdx.y].ev_on[mode]) return;
    const BaDev &B = Bs[blockIdx.y];
    if ((int) blockIdx.x >= B.nKf * B.nKf) return;
    pairs_body(B, blockIdx.x, 0);
}
__global__ void __launch_bounds__(ASM_NT) k_assemble_b(const BaDev *Bs, const BaRun *Rs, int mode) {
    if (!Rs[blockIdx.x].ev_on[mode]) return;
    assemble_body(Bs[blockIdx.x], mode == 0 ? 1 : 0);
}
__global__ void __launch_bounds__(256) k_prep_b(const BaDev *Bs, const BaRun *Rs) {
    const BaRun &R = Rs[blockIdx.y];
    if (!R.step) return;
    prep_body<1>(Bs[blockIdx.y], R.radius, R.diag, blockIdx.x, 0);
}
__global__ void __launch_bounds__(64) k_gemm_b(const BaDev *Bs, const BaRun *Rs) {
    const BaRun &R = Rs[blockIdx.z];
    if (!R.step) return;
    const BaDev &B = Bs[blockIdx.z];
    const int tiles = B.NP / 16;
    if ((int) blockIdx.x >= tiles * tiles) return;
    gemm_body(B, blockIdx.x, blockIdx.y);
}
__global__ void __launch_bounds__(256) k_reduced_system_b(const BaDev *Bs, const BaRun *Rs) {
    const BaRun &R = Rs[blockIdx.y];
    if (!R.step) return;
    reduced_system_body(Bs[blockIdx.y], R.radius, R.diag, blockIdx.x, 0);
}
__global__ void __launch_bounds__(SOLVE_NT) k_solve_b(const BaDev *Bs, const BaRun *Rs) {
    const BaRun &R = Rs[blockIdx.x];
    if (!R.step) return;
    solve_body<true>(Bs[blockIdx.x], R.radius, 0, 0);
}
__global__ void __launch_bounds__(256) k_backsub_b(const BaDev *Bs, const BaRun *Rs) {
    const BaRun &R = Rs[blockIdx.y];
    if (!R.step) return;
    const BaDev &B = Bs[blockIdx.y];
    if ((int) blockIdx.x > (B.nPt + 3) / 4) return;   // the grid covers the largest problem (+ its pose workgroup)
    backsub_body<1>(B, R.radius, R.xt, R.ct, R.xp, R.cp, blockIdx.x, 0);
}

// ---- the LM loop's DECISIONS on the device (round 6) -----------------------------------------------------------------------------------
// Ceres' TrustRegionMinimizer::Minimize was restated on the host (BaHost::advance): after every candidate's evaluation the host read six
// scalars, decided, and only then enqueued the next iteration's eight kernels -- the GPU sat idle for a bus round trip + the enqueue once
// per LM iteration, and a rejected step cost three more launches (the Jacobian-derived state had to be restored at x).  Here
//   * the decision is taken by the thread that has just summed the evaluation's scalars (k_assemble_lm's thread 0: lm_after_eval below --
//     trust_region_minimizer.cc:377-451, 461-490, 781-829; levenberg_marquardt_strategy.cc:66-160 through LmState) and written to a
//     state block in device memory that the next iteration's kernels read: radius, "refresh the diagonal", "this iteration runs";
//   * the evaluation's outputs are DOUBLE-BUFFERED (two BaDev descriptors that differ in chi2 | depth, Jobs, rs, ptCost, Hpp, gp, Wt, M,
//     Hcc, gc; two parameter pairs): the candidate is evaluated into the set x does not use, an accepted step flips the roles, a rejected
//     one leaves x's set untouched -- no restore launches;
//   * the host stays ONE ITERATION AHEAD: it enqueues iteration i + 1 when it sees that iteration i runs, so the queue never drains; when
//     the minimiser stops, at most one enqueued iteration finds "step == 0" and returns at once.
// Same kernels' bodies, same arithmetic, same decisions as the host loop (tests/test_gpu_ba.py: iteration and accepted-step counts and
// chi2 classes against Ceres; ALVA_BA_HOST_LM=1 keeps the host loop for A/B).
struct BaLmDev {
    double radius, decrease_factor;
    double x_cost, initial, gmax, x_norm, function_tolerance;
    double *P[2], *T[2];   // pose / point parameter vectors: [cur] = x, [1 - cur] = the candidate
    double *h_pub;         // pinned host, two slots ([0..7] even, [16..23] odd evaluation numbers): step, iteration, nsummaries, initial, x_cost, nsucc, ok,
                           // cur | last << 8; [8] = evaluations published (written last, system-scope release)
    long long evals;       // evaluations made so far (the publication's sequence number)
    int reuse_diagonal, iteration, nsucc, invalid, nsummaries, ok, max_iters;
    int cur;               // which descriptor / parameter pair holds x
    int last;              // the descriptor the latest evaluation wrote (the outlier sweep reads chi2 | depth there)
    int step;              // 1: the next iteration runs, 0: the minimiser has stopped
    int diag;              // ... and refreshes the LM diagonal first
    int pad;
};
static_assert(sizeof(BaLmDev) % 8 == 0, "BaLmDev is copied as 8-byte words");

__device__ __forceinline__ void lm_after_eval(BaLmDev &L, const double *scal, const int first) {
    LmState lm;
    lm.radius = L.radius; lm.decrease_factor = L.decrease_factor; lm.reuse_diagonal = L.reuse_diagonal;
    bool stop = false;
    if (first) {
        L.x_cost = L.initial = scal[0];
        L.gmax = scal[3];
        L.last = L.cur;
    } else {
        const int cand = 1 - L.cur;
        L.last = cand;
        const double cand_cost = scal[0], mcc = scal[1], step_norm = sqrt(scal[2]);
        const bool okstep = scal[5] != 0.0 && isfinite(mcc);
        if (!okstep || !(mcc > 0)) {  // HandleInvalidStep (:461-490)
            if (++L.invalid >= 5) {
                L.ok = 0;
                stop = true;
            } else {
                lm.rejected();
                L.nsummaries++;
            }
        } else {
            L.invalid = 0;
            if (step_norm <= 1e-8 * (L.x_norm + 1e-8)) stop = true;                                   // ParameterToleranceReached
            else if (fabs(L.x_cost - cand_cost) <= L.function_tolerance * L.x_cost) stop = true;    // FunctionToleranceReached
            else {
                const double rel = (L.x_cost - cand_cost) / mcc;
                if (rel > 1e-3) {
                    L.cur = cand;
                    L.x_cost = cand_cost;
                    L.gmax = scal[3];
                    L.x_norm = sqrt(scal[4]);
                    lm.accepted(rel);
                    L.nsucc++;
                } else {
                    lm.rejected();
                }
                L.nsummaries++;
            }
        }
    }
    L.radius = lm.radius; L.decrease_factor = lm.decrease_factor; L.reuse_diagonal = lm.reuse_diagonal;
    // the loop's head (ba_drive): stop tests, then the next iteration's prologue
    if (stop || L.iteration >= L.max_iters || L.gmax <= 1e-10 || L.radius <= 1e-32) {
        L.step = 0;
        return;
    }
    L.iteration++;
    L.diag = L.reuse_diagonal ? 0 : 1;
    L.reuse_diagonal = 1;
    L.step = 1;
}

// the evaluation kernels: first = the evaluation at x (descriptor / parameters `cur`), otherwise the candidate's (the other set)
#define LM_EVAL_SEL(first)                                  \
    if (!(first) && !L->step) return;                       \
    const int set = (first) ? L->cur : 1 - L->cur;          \
    const BaDev &B = dB[set];
template<bool INV>
__global__ void __launch_bounds__(256) k_point_lm(const BaDev *__restrict__ dB, const BaLmDev *__restrict__ L, int first) {
    LM_EVAL_SEL(first)
    point_body<INV, true>(B, L->P[set], L->T[set], blockIdx.x, 0);
}
__global__ void __launch_bounds__(256) k_pairs_lm(const BaDev *__restrict__ dB, const BaLmDev *__restrict__ L, int first) {
    LM_EVAL_SEL(first)
    pairs_body(B, blockIdx.x, 0);
}
__global__ void __launch_bounds__(ASM_NT) k_assemble_lm(const BaDev *__restrict__ dB, BaLmDev *L, int first) {
    LM_EVAL_SEL(first)
    assemble_body(B, first);   // (h_scal is null in these descriptors: the publication below replaces it)
    if (threadIdx.x == 0) {
        BaLmDev S = *L;
        lm_after_eval(S, B.scal, first);
        S.evals++;
        *L = S;
        // two slots, evaluation number odd / even: the host reads the slot of the number it saw while the next evaluation -- the only one that
        // can be under way, the host being one iteration ahead -- writes the other
        double *h = S.h_pub + ((S.evals & 1) ? 16 : 0);
        h[0] = S.step; h[1] = S.iteration; h[2] = S.nsummaries; h[3] = S.initial; h[4] = S.x_cost; h[5] = S.nsucc; h[6] = S.ok;
        h[7] = (double) (S.cur | (S.last << 8));
        __threadfence_system();
        __hip_atomic_store(reinterpret_cast<long long *>(S.h_pub + 8), S.evals, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}
// the step kernels: on x's descriptor, only while the minimiser runs
template<int DP>
__global__ void __launch_bounds__(256) k_prep_lm(const BaDev *__restrict__ dB, const BaLmDev *__restrict__ L) {
    if (!L->step) return;
    prep_body<DP>(dB[L->cur], L->radius, L->diag, blockIdx.x, 0);
}
__global__ void __launch_bounds__(64) k_gemm_lm(const BaDev *__restrict__ dB, const BaLmDev *__restrict__ L) {
    if (!L->step) return;
    gemm_body(dB[L->cur], blockIdx.x, blockIdx.y);
}
__global__ void __launch_bounds__(256) k_reduced_system_lm(const BaDev *__restrict__ dB, const BaLmDev *__restrict__ L) {
    if (!L->step) return;
    reduced_system_body(dB[L->cur], L->radius, L->diag, blockIdx.x, 0);
}
template<bool IN_LDS>
__global__ void __launch_bounds__(SOLVE_NT) k_solve_lm(const BaDev *__restrict__ dB, const BaLmDev *__restrict__ L) {
    if (!L->step) return;
    solve_body<IN_LDS>(dB[L->cur], L->radius, 0, 0);
}
template<int DP>
__global__ void __launch_bounds__(256) k_backsub_lm(const BaDev *__restrict__ dB, const BaLmDev *__restrict__ L) {
    if (!L->step) return;
    const int c = L->cur;
    backsub_body<DP>(dB[c], L->radius, L->T[c], L->T[1 - c], L->P[c], L->P[1 - c], blockIdx.x, 0);
}
// ---- one problem on the host: sizes, the carved device block with its pinned mirror, the structure build -------------------------
struct BaIn {
    int n_kf;
    double *h_poses;
    const uint8_t *h_kf_const;
    const double *h_calib;
    int inv_depth, n_pt;
    const int *h_pt_anchor_kf;
    const double *h_pt_anchor_uv;
    double *h_pt_param;
    int n_obs;
    const int *h_obs_kf, *h_obs_pt;
    const double *h_obs_uv;
    double huber_chi2;
};

// the same problem with the observations already grouped by point (alva_local_ba_csr): h_pt_ptr[n_pt + 1], observation arrays in that order
struct BaCsr {
    const int *h_pt_ptr;
    double chi2_threshold;
    unsigned long long *h_bad_bits;
    int *h_n_bad;
};

struct BaHost {
    BaDev B{};
    std::vector<int> cidx, order;
    bool grouped_ = false;   // the caller's observations arrived grouped by point: `order` is the identity
    bool csr_ = false;       // the pair grouping is built on the device (k_pair_*): its arrays lie behind the uploaded block
    int *d_pairKey = nullptr, *d_pairHist = nullptr;
    unsigned long long *d_badBits = nullptr;
    int n_chunks = 0;
    int *d_obsKf = nullptr, *d_ptPtr = nullptr, *d_ancKf = nullptr, *d_cidx = nullptr, *d_pairPerm = nullptr, *d_pairPtr = nullptr, *d_kfOf = nullptr;
    double *d_obsUv = nullptr, *d_ancUv = nullptr, *d_xp = nullptr, *d_cp = nullptr, *d_xt = nullptr, *d_ct = nullptr;
    size_t in_bytes = 0, bytes = 0;
    size_t np16 = 0, solve_lds = 0;
    // LM state (Ceres TrustRegionMinimizer, trust_region_minimizer.cc:67-136)
    LmState lm;
    double x_cost = 0, gmax = 0, x_norm = -1, initial = 0;
    int iteration = 0, nsucc = 1, invalid = 0, nsummaries = 1, ok = 1;
    bool need_restore = false, done = false;
    double *xp = nullptr, *xt = nullptr, *cp = nullptr, *ct = nullptr;
    // the device-side LM loop (BaLmDev): its state block and the two descriptors live in the uploaded input block; B2 = the second set of
    // the evaluation's outputs (everything else of the descriptor is shared)
    BaLmDev *d_lm = nullptr;
    BaDev *d_desc = nullptr;
    BaDev B2{};

    int sizes(const BaIn &in, const BaCsr *csr = nullptr) {
        const int n_kf = in.n_kf, n_pt = in.n_pt, n_obs = in.n_obs, dp = in.inv_depth ? 1 : 3;
        cidx.assign((size_t) n_kf, -1);
        int nc = 0;
        for (int k = 0; k < n_kf; k++) cidx[(size_t) k] = in.h_kf_const[k] ? -1 : nc++;
        csr_ = csr != nullptr;
        if (csr) {
            unsigned bad = 0;   // (branch-free passes: the arrays are the map layer's own, the check is the C ABI's)
            for (int o = 0; o < n_obs; o++) bad |= (unsigned) in.h_obs_kf[o] >= (unsigned) n_kf;
            for (int p2 = 0; p2 < n_pt; p2++) bad |= (unsigned) in.h_pt_anchor_kf[p2] >= (unsigned) n_kf || csr->h_pt_ptr[p2] > csr->h_pt_ptr[p2 + 1];
            ALVA_ARG(!bad && csr->h_pt_ptr[0] == 0 && csr->h_pt_ptr[n_pt] == n_obs && n_kf <= 32);
            n_chunks = (n_obs + PAIR_CHUNK - 1) / PAIR_CHUNK;
        } else
        for (int o = 0; o < n_obs; o++) ALVA_ARG(in.h_obs_kf[o] >= 0 && in.h_obs_kf[o] < n_kf && in.h_obs_pt[o] >= 0 && in.h_obs_pt[o] < n_pt);
        B.nKf = n_kf; B.nPt = n_pt; B.nObs = n_obs; B.inv = in.inv_depth; B.dp = dp; B.nc = nc; B.n6 = 6 * nc;
        B.NP = (B.n6 + 1 + 15) / 16 * 16;
        B.npd = n_pt * dp;
        const int kq = 4 * KSPLIT;
        B.kpad = std::max(kq, (B.npd + kq - 1) / kq * kq);
        for (int i = 0; i < 4; i++) B.K[i] = in.h_calib[i];
        B.huber_a = (double) sqrtf((float) in.huber_chi2);  // optimizer.cpp:22: std::sqrt of a float
        np16 = ((size_t) B.n6 + 15) / 16 * 16;
        solve_lds = (np16 * (np16 + 1) + np16) * sizeof(double);
        bytes = layout(nullptr);
        return ALVA_OK;
    }
    // one block, carved; the INPUT arrays come first and contiguous so that one copy uploads them
    size_t layout(uint8_t *base) {
        const size_t nObs = (size_t) B.nObs, nPt = (size_t) B.nPt, npd = (size_t) B.npd, n6 = (size_t) B.n6, NP = (size_t) B.NP, dp = (size_t) B.dp;
        const size_t n_kf = (size_t) B.nKf;
        uint8_t *cur = base;
        d_obsKf = carve<int>(cur, nObs);
        d_obsUv = carve<double>(cur, nObs * 2);
        d_ptPtr = carve<int>(cur, nPt + 1);
        d_ancKf = carve<int>(cur, nPt);
        d_ancUv = carve<double>(cur, nPt * 2);
        d_cidx = carve<int>(cur, n_kf);
        d_kfOf = carve<int>(cur, n_kf);
        if (!csr_) {
            d_pairPerm = carve<int>(cur, nObs);
            d_pairPtr = carve<int>(cur, n_kf * n_kf + 1);
        }
        d_xp = carve<double>(cur, n_kf * 7);
        d_xt = carve<double>(cur, npd);
        d_lm = carve<BaLmDev>(cur, 1);
        d_desc = carve<BaDev>(cur, 2);
        in_bytes = (size_t) (cur - base);
        if (csr_) {
            d_pairPerm = carve<int>(cur, nObs);
            d_pairPtr = carve<int>(cur, n_kf * n_kf + 1);
            d_pairKey = carve<int>(cur, nObs);
            d_pairHist = carve<int>(cur, (size_t) n_chunks * n_kf * n_kf);
            d_badBits = carve<unsigned long long>(cur, nObs / 64 + 2);
        }
        B.chi2 = carve<double>(cur, nObs);   // results the host reads back: chi2 | depth flags, contiguous
        B.depth = carve<uint8_t>(cur, nObs);
        B.Jobs = carve<double>(cur, nObs * 12);
        B.rs = carve<double>(cur, nObs * 2);
        B.ptCost = carve<double>(cur, nPt);
        B.Hpp = carve<double>(cur, npd * dp);
        B.gp = carve<double>(cur, npd);
        B.Wt = carve<double>(cur, npd * NP);
        B.Zt = carve<double>(cur, (size_t) B.kpad * NP);   // right behind Wt: the two are zeroed by ONE fill
        B.M = carve<double>(cur, n_kf * n_kf * 27);
        B.rowcol = carve<double>(cur, n_kf * 27 * 2);
        B.Hcc = carve<double>(cur, n6 * n6);
        B.gc = carve<double>(cur, n6);
        B.sc = carve<double>(cur, n6);
        B.sp = carve<double>(cur, npd);
        B.dc = carve<double>(cur, n6);
        B.dpd = carve<double>(cur, npd);
        B.hinv = carve<double>(cur, npd * dp);
        B.Gpart = carve<double>(cur, (size_t) KSPLIT * NP * NP);
        B.S = carve<double>(cur, (n6 + 16) * (n6 + 17) + n6 + 16);  // padded to a multiple of 16, odd row stride, + rhs
        B.yc = carve<double>(cur, NP);
        B.yp = carve<double>(cur, npd);
        B.scal = carve<double>(cur, 64);
        B.dbg = alva_kstamp_buffer();
        B.partial = carve<double>(cur, nPt * 3 + 8);   // per point: mcc, step^2, candidate^2 partials; then the camera part (3)
        d_cp = carve<double>(cur, n_kf * 7);
        d_ct = carve<double>(cur, npd);
        B2.chi2 = carve<double>(cur, nObs);   // (chi2 | depth in the first set's order: chi_bytes() holds for both)
        B2.depth = carve<uint8_t>(cur, nObs);
        B2.Jobs = carve<double>(cur, nObs * 12);
        B2.rs = carve<double>(cur, nObs * 2);
        B2.ptCost = carve<double>(cur, nPt);
        B2.Hpp = carve<double>(cur, npd * dp);
        B2.gp = carve<double>(cur, npd);
        B2.Wt = carve<double>(cur, npd * NP);
        B2.M = carve<double>(cur, n_kf * n_kf * 27);
        B2.Hcc = carve<double>(cur, n6 * n6);
        B2.gc = carve<double>(cur, n6);
        return (size_t) (cur - base);
    }
    // the analogue of Ceres' program / block-structure build, written into the pinned mirror `stage` of the input arrays; then the
    // device pointers are bound to `base`
    int build(const BaIn &in, uint8_t *base, uint8_t *stage) {
        const int n_kf = in.n_kf, n_pt = in.n_pt, n_obs = in.n_obs;
        const size_t nPt = (size_t) n_pt, npd = (size_t) B.npd;
        layout(stage);
        int *h_obsKf = d_obsKf, *h_ptPtr = d_ptPtr, *h_ancKf = d_ancKf, *h_cidx = d_cidx, *h_kfOf = d_kfOf, *h_pairPerm = d_pairPerm, *h_pairPtr = d_pairPtr;
        double *h_obsUv = d_obsUv, *h_ancUv = d_ancUv, *h_xp = d_xp, *h_xt = d_xt;
        layout(base);
        B.obsKf = d_obsKf; B.obsUv = d_obsUv; B.ptPtr = d_ptPtr; B.ancKf = d_ancKf; B.ancUv = d_ancUv; B.cidx = d_cidx; B.kfOf = d_kfOf;
        B.pairPerm = d_pairPerm; B.pairPtr = d_pairPtr;
        // observations grouped by point, original order kept inside a point: a stable counting sort (O(n)).  The map layer hands them
        // over point by point already (then the sort is the identity): ONE pass then copies them, counts per point and per (observing
        // keyframe, anchor keyframe) pair -- the host structure of an in-system solve is a fixed cost of every keyframe (it was ~80 us
        // of six passes and three vector allocations for 14 k observations)
        order.resize((size_t) n_obs);
        static thread_local std::vector<int> pairKey, cursor;
        pairKey.resize((size_t) n_obs);
        for (size_t p2 = 0; p2 <= nPt; p2++) h_ptPtr[p2] = 0;
        const size_t nPairs = (size_t) n_kf * n_kf;
        for (size_t i = 0; i <= nPairs; i++) h_pairPtr[i] = 0;
        bool grouped = true;
        for (int o = 1; o < n_obs && grouped; o++) grouped = in.h_obs_pt[o] >= in.h_obs_pt[o - 1];
        grouped_ = grouped;
        if (grouped) {
            for (int o = 0; o < n_obs; o++) order[(size_t) o] = o;
        } else {
            cursor.assign((size_t) n_pt + 1, 0);
            for (int o = 0; o < n_obs; o++) cursor[(size_t) in.h_obs_pt[o] + 1]++;
            for (int p2 = 0; p2 < n_pt; p2++) cursor[(size_t) p2 + 1] += cursor[(size_t) p2];
            for (int o = 0; o < n_obs; o++) order[(size_t) cursor[(size_t) in.h_obs_pt[o]]++] = o;
        }
        for (int q = 0; q < n_obs; q++) {
            const int o = order[(size_t) q], pt = in.h_obs_pt[o], kf = in.h_obs_kf[o];
            h_obsKf[q] = kf;
            h_obsUv[2 * (size_t) q] = in.h_obs_uv[2 * o];
            h_obsUv[2 * (size_t) q + 1] = in.h_obs_uv[2 * o + 1];
            h_ptPtr[(size_t) pt + 1]++;
            const int anc = in.inv_depth ? in.h_pt_anchor_kf[pt] : kf;
            ALVA_ARG(anc >= 0 && anc < n_kf);
            const int key = kf * n_kf + anc;
            pairKey[(size_t) q] = key;
            h_pairPtr[(size_t) key + 1]++;
        }
        for (int p2 = 0; p2 < n_pt; p2++) h_ptPtr[(size_t) p2 + 1] += h_ptPtr[(size_t) p2];
        // the same observations grouped by (observing kf, anchor kf) pair, stable: counting sort again
        for (size_t i = 0; i < nPairs; i++) h_pairPtr[i + 1] += h_pairPtr[i];
        cursor.assign(h_pairPtr, h_pairPtr + nPairs);
        for (int q = 0; q < n_obs; q++) h_pairPerm[(size_t) cursor[(size_t) pairKey[(size_t) q]]++] = q;
        if (in.inv_depth && n_pt > 0) {
            memcpy(h_ancKf, in.h_pt_anchor_kf, nPt * 4);
            memcpy(h_ancUv, in.h_pt_anchor_uv, nPt * 16);
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
            se3_from_pose7(in.h_poses + 7 * k, T);
            for (int i = 0; i < 3; i++) h_xp[7 * (size_t) k + i] = T.t[i];
            for (int i = 0; i < 4; i++) h_xp[7 * (size_t) k + 3 + i] = T.q[i];
        }
        if (npd) memcpy(h_xt, in.h_pt_param, npd * 8);
        xp = d_xp; xt = d_xt; cp = d_cp; ct = d_ct;
        return ALVA_OK;
    }
    // the same for observations that arrive grouped by point with their ptPtr (alva_local_ba_csr): the input arrays are copied as they
    // are; the pair grouping is left to the device (enqueue_pairs, behind the upload)
    int build_csr(const BaIn &in, const BaCsr &csr, uint8_t *base, uint8_t *stage) {
        const int n_kf = in.n_kf, n_pt = in.n_pt, n_obs = in.n_obs;
        const size_t nPt = (size_t) n_pt, npd = (size_t) B.npd;
        layout(stage);
        int *h_obsKf = d_obsKf, *h_ptPtr = d_ptPtr, *h_ancKf = d_ancKf, *h_cidx = d_cidx, *h_kfOf = d_kfOf;
        double *h_obsUv = d_obsUv, *h_ancUv = d_ancUv, *h_xp = d_xp, *h_xt = d_xt;
        layout(base);
        B.obsKf = d_obsKf; B.obsUv = d_obsUv; B.ptPtr = d_ptPtr; B.ancKf = d_ancKf; B.ancUv = d_ancUv; B.cidx = d_cidx; B.kfOf = d_kfOf;
        B.pairPerm = d_pairPerm; B.pairPtr = d_pairPtr;
        grouped_ = true;
        if (n_obs) {
            memcpy(h_obsKf, in.h_obs_kf, (size_t) n_obs * 4);
            memcpy(h_obsUv, in.h_obs_uv, (size_t) n_obs * 16);
        }
        memcpy(h_ptPtr, csr.h_pt_ptr, (nPt + 1) * 4);
        if (n_pt > 0) {
            memcpy(h_ancKf, in.h_pt_anchor_kf, nPt * 4);
            memcpy(h_ancUv, in.h_pt_anchor_uv, nPt * 16);
        }
        for (int k = 0; k < n_kf; k++) {
            h_cidx[k] = cidx[(size_t) k];
            h_kfOf[k] = 0;
        }
        for (int k = 0; k < n_kf; k++)
            if (cidx[(size_t) k] >= 0) h_kfOf[cidx[(size_t) k]] = k;
        for (int k = 0; k < n_kf; k++) {
            Se3 T;
            se3_from_pose7(in.h_poses + 7 * k, T);
            for (int i = 0; i < 3; i++) h_xp[7 * (size_t) k + i] = T.t[i];
            for (int i = 0; i < 4; i++) h_xp[7 * (size_t) k + 3 + i] = T.q[i];
        }
        if (npd) memcpy(h_xt, in.h_pt_param, npd * 8);
        xp = d_xp; xt = d_xt; cp = d_cp; ct = d_ct;
        return ALVA_OK;
    }
    // the device-side LM loop's state and its two descriptors, written into the pinned mirror of the input block (uploaded with it)
    void stage_lm(uint8_t *base, uint8_t *stage, int max_iters, double function_tolerance, double *h_pub) {
        BaDev D0 = B, D1 = B;
        D0.h_scal = D1.h_scal = nullptr;
        D1.chi2 = B2.chi2; D1.depth = B2.depth; D1.Jobs = B2.Jobs; D1.rs = B2.rs; D1.ptCost = B2.ptCost; D1.Hpp = B2.Hpp; D1.gp = B2.gp;
        D1.Wt = B2.Wt; D1.M = B2.M; D1.Hcc = B2.Hcc; D1.gc = B2.gc;
        BaDev *h_desc = reinterpret_cast<BaDev *>(stage + ((uint8_t *) d_desc - base));
        h_desc[0] = D0;
        h_desc[1] = D1;
        BaLmDev L{};
        L.radius = lm.radius; L.decrease_factor = lm.decrease_factor; L.reuse_diagonal = lm.reuse_diagonal;
        L.x_norm = x_norm;
        L.function_tolerance = function_tolerance;
        L.P[0] = d_xp; L.P[1] = d_cp; L.T[0] = d_xt; L.T[1] = d_ct;
        L.h_pub = h_pub;
        L.nsucc = nsucc; L.nsummaries = nsummaries; L.ok = 1; L.max_iters = max_iters;
        L.step = 0;
        *reinterpret_cast<BaLmDev *>(stage + ((uint8_t *) d_lm - base)) = L;
    }
    void enqueue_pairs(hipStream_t st) {
        if (B.nObs <= 0) {
            (void) hipMemsetAsync(d_pairPtr, 0, ((size_t) B.nKf * B.nKf + 1) * 4, st);
            return;
        }
        PairArgs A{d_obsKf, d_ptPtr, d_ancKf, B.nObs, B.nPt, B.nKf, n_chunks, d_pairKey, d_pairHist, d_pairPtr, d_pairPerm};
        hipLaunchKernelGGL(k_pair_hist, dim3((unsigned) n_chunks), dim3(PAIR_CHUNK), (size_t) B.nKf * B.nKf * 4, st, A);
        hipLaunchKernelGGL(k_pair_scan, dim3(1), dim3(1024), 0, st, A);
        hipLaunchKernelGGL(k_pair_fill, dim3((unsigned) n_chunks), dim3(PAIR_CHUNK), 0, st, A);
    }
    // Ceres' accept / reject logic on the scalars of the candidate's evaluation (trust_region_minimizer.cc:461-490, :781-829);
    // returns true when the minimiser stops
    bool advance(const double *scal, double function_tolerance) {
        const double cand_cost = scal[0], mcc = scal[1], step_norm = std::sqrt(scal[2]);
        const bool okstep = scal[5] != 0.0 && std::isfinite(mcc);
        if (!okstep || !(mcc > 0)) {  // HandleInvalidStep (:461-490)
            if (++invalid >= 5) {
                ok = 0;
                return true;
            }
            lm.rejected();
            nsummaries++;
            need_restore = true;
            return false;
        }
        invalid = 0;
        if (step_norm <= 1e-8 * (x_norm + 1e-8)) return true;                                // ParameterToleranceReached
        if (std::fabs(x_cost - cand_cost) <= function_tolerance * x_cost) return true;      // FunctionToleranceReached
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
        return false;
    }
    bool loop_ends(int max_iters) const { return iteration >= max_iters || gmax <= 1e-10 || lm.radius <= 1e-32; }
    // results back in the caller's arrays; r_* point at the downloaded chi2 | depth block, poses and point parameters
    void finish(const BaIn &in, const uint8_t *r_chi, const double *r_poses, const double *r_pts, double *h_chi2, uint8_t *h_depth_pos, double *h_info) {
        const size_t npd = (size_t) B.npd;
        if (npd) memcpy(in.h_pt_param, r_pts, npd * 8);
        const double *chi2s = reinterpret_cast<const double *>(r_chi);
        const uint8_t *deps = r_chi + ((uint8_t *) B.depth - (uint8_t *) B.chi2);
        for (int k = 0; k < in.n_kf; k++)
            if (cidx[(size_t) k] >= 0) memcpy(in.h_poses + 7 * k, r_poses + 7 * (size_t) k, 56);
        if (grouped_) {   // the caller's order IS the device order
            if (h_chi2) memcpy(h_chi2, chi2s, (size_t) in.n_obs * 8);
            if (h_depth_pos) memcpy(h_depth_pos, deps, (size_t) in.n_obs);
        } else {
            for (int q = 0; q < in.n_obs; q++) {  // back to the caller's observation order
                if (h_chi2) h_chi2[order[(size_t) q]] = chi2s[(size_t) q];
                if (h_depth_pos) h_depth_pos[order[(size_t) q]] = deps[(size_t) q];
            }
        }
        if (h_info) {
            h_info[0] = nsummaries;
            h_info[1] = initial;
            h_info[2] = x_cost;
            h_info[3] = nsucc;
        }
    }
    size_t chi_bytes() const { return (size_t) ((uint8_t *) B.depth - (uint8_t *) B.chi2) + (size_t) B.nObs; }   // chi2 | pad | depth, as carved
};

}  // namespace

// one solve: `csr` null = alva_local_ba (observations in any order, the structure built on the host, chi2 / depth flags returned);
// non-null = alva_local_ba_csr (observations grouped by point, the pair grouping built on the device, the sweep's flags returned as bits)
static int ba_drive(alva_ctx *ctx, const BaIn &in, const BaCsr *csr, int max_iters, double function_tolerance, double *h_chi2,
                    uint8_t *h_depth_pos, double *h_info, int *h_ok) {
    const int n_kf = in.n_kf, n_pt = in.n_pt, n_obs = in.n_obs, inv_depth = in.inv_depth;
    *h_ok = 1;
    if (h_info) memset(h_info, 0, 4 * sizeof(double));
    const auto t_begin = std::chrono::steady_clock::now();
    BaHost H;
    int rc = H.sizes(in, csr);
    if (rc) return rc;
    BaDev &B = H.B;
    const int dp = B.dp;
    const size_t nObs = (size_t) n_obs, npd = (size_t) B.npd, NP = (size_t) B.NP;
    uint8_t *base = nullptr;
    rc = alva_ctx_scratch(ctx, 4, H.bytes, (void **) &base);
    if (rc) return rc;
    // pinned staging: [0, 256) the per-iteration scalars | the input block (mirror of the device layout) | results
    const size_t res_bytes = (nObs * 9 + 511) / 256 * 256 + (size_t) n_kf * 56 + 256 + npd * 8 + 256;
    uint8_t *pin = nullptr;
    rc = alva_ctx_pinned(ctx, 256 + std::max(H.in_bytes, res_bytes), (void **) &pin);
    if (rc) return rc;
    uint8_t *stage = pin + 256;
    hipStream_t st = ctx->stream;
    const auto t_sized = std::chrono::steady_clock::now();
    ALVA_HIP(alva_stream_sync(st));  // nothing enqueued earlier may still be reading the staging area
    const auto t_synced = std::chrono::steady_clock::now();
    rc = csr ? H.build_csr(in, *csr, base, stage) : H.build(in, base, stage);
    if (rc) return rc;
    const auto t_built = std::chrono::steady_clock::now();
    H.stage_lm(base, stage, max_iters, function_tolerance, reinterpret_cast<double *>(pin));
    ALVA_HIP(hipMemcpyAsync(base, stage, H.in_bytes, hipMemcpyHostToDevice, st));   // ONE upload from pinned memory
    if (csr) H.enqueue_pairs(st);
    // Wt: the sparsity pattern is fixed, zero once; Zt: the K padding rows stay zero -- neighbours in the layout, one fill
    ALVA_HIP(hipMemsetAsync(B.Wt, 0, (size_t) ((uint8_t *) B.Zt - (uint8_t *) B.Wt) + (size_t) B.kpad * NP * 8, st));
    const auto t_up = std::chrono::steady_clock::now();

    const dim3 gPt((unsigned) alva_divup(std::max(n_pt, 1), 4)), blk(256);
    // per-iteration scalars: published by k_assemble into the first 128 bytes of the pinned staging and polled (ALVA_NO_POLL=1: copy + wait)
    static const bool poll = getenv("ALVA_NO_POLL") == nullptr;
    double *pin_scal = reinterpret_cast<double *>(pin);
    long long eval_seq = 0;
    reinterpret_cast<volatile long long *>(pin_scal + 8)[0] = 0;
    B.h_scal = poll ? pin_scal : nullptr;
    const size_t np16 = H.np16, solve_lds = H.solve_lds;
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
        B.seq = ++eval_seq;
        hipLaunchKernelGGL(k_assemble, dim3(1), dim3(ASM_NT), assemble_lds(n_kf), st, B, first ? 1 : 0);  // H_cc, g_c, the scalars; publishes them
        ALVA_LAUNCH_CHECK();
        return ALVA_OK;
    };
    double scal[8];
    auto read_scal = [&]() -> int {
        if (poll) {
            const volatile long long *flag = reinterpret_cast<const volatile long long *>(pin_scal + 8);
            unsigned spins = 0;
            while (*flag != eval_seq) {
                if (++spins > (1u << 26)) {
                    ALVA_HIP(alva_stream_sync(st));
                    break;
                }
                alva_poll_relax(spins);
            }
            __atomic_thread_fence(__ATOMIC_ACQUIRE);
            memcpy(scal, pin_scal, sizeof(scal));
            return ALVA_OK;
        }
        ALVA_HIP(hipMemcpyAsync(pin_scal, B.scal, sizeof(scal), hipMemcpyDeviceToHost, st));  // pinned: a plain DMA, no staging
        ALVA_HIP(alva_stream_sync(st));
        memcpy(scal, pin_scal, sizeof(scal));
        return ALVA_OK;
    };

    static const bool host_lm = getenv("ALVA_BA_HOST_LM") != nullptr;
    const bool dev_lm = poll && !host_lm;
    int last_set = 0, cur_set = 0;
    if (dev_lm) {
        // ---- the same minimiser with its decisions on the device (BaLmDev): the host enqueues, one iteration ahead, and watches ---------
        ALVA_HIP(hipMemsetAsync(H.B2.Wt, 0, (size_t) B.npd * NP * 8, st));
        const BaDev *dB = H.d_desc;
        BaLmDev *dL = H.d_lm;
        const int tiles = B.NP / 16, np16i = (int) np16;
        const dim3 gBs(n_pt > 0 ? gPt.x + 1 : 1);
        auto eval_lm = [&](int first) {
            if (n_pt > 0) {
                if (inv_depth) hipLaunchKernelGGL(k_point_lm<true>, gPt, blk, 0, st, dB, (const BaLmDev *) dL, first);
                else hipLaunchKernelGGL(k_point_lm<false>, gPt, blk, 0, st, dB, (const BaLmDev *) dL, first);
            }
            hipLaunchKernelGGL(k_pairs_lm, dim3((unsigned) (n_kf * n_kf)), blk, 0, st, dB, (const BaLmDev *) dL, first);
            hipLaunchKernelGGL(k_assemble_lm, dim3(1), dim3(ASM_NT), assemble_lds(n_kf), st, dB, dL, first);
        };
        auto step_lm = [&]() {
            if (n_pt > 0) {
                if (dp == 1) hipLaunchKernelGGL(k_prep_lm<1>, gPt, blk, 0, st, dB, (const BaLmDev *) dL);
                else hipLaunchKernelGGL(k_prep_lm<3>, gPt, blk, 0, st, dB, (const BaLmDev *) dL);
            }
            hipLaunchKernelGGL(k_gemm_lm, dim3((unsigned) (tiles * tiles), KSPLIT), dim3(64), 0, st, dB, (const BaLmDev *) dL);
            if (np16 > 0)
                hipLaunchKernelGGL(k_reduced_system_lm, dim3((unsigned) alva_divup(np16i * np16i + np16i, 256)), blk, 0, st, dB, (const BaLmDev *) dL);
            if (solve_in_lds) hipLaunchKernelGGL(k_solve_lm<true>, dim3(1), dim3(SOLVE_NT), solve_lds, st, dB, (const BaLmDev *) dL);
            else hipLaunchKernelGGL(k_solve_lm<false>, dim3(1), dim3(SOLVE_NT), 0, st, dB, (const BaLmDev *) dL);
            if (dp == 1) hipLaunchKernelGGL(k_backsub_lm<1>, gBs, blk, 0, st, dB, (const BaLmDev *) dL);
            else hipLaunchKernelGGL(k_backsub_lm<3>, gBs, blk, 0, st, dB, (const BaLmDev *) dL);
            eval_lm(0);
        };
        if (solve_in_lds && solve_lds > 48 * 1024)
            ALVA_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_solve_lm<true>), hipFuncAttributeMaxDynamicSharedMemorySize, 152 * 1024));
        eval_lm(1);
        int issued = 0;
        if (max_iters > 0) {
            step_lm();   // iteration 0: it runs if the first evaluation says so
            issued = 1;
        }
        ALVA_LAUNCH_CHECK();
        long long seen = 0;
        double pub[8];
        for (;;) {
            // publication `seen + 1`: the decision behind evaluation number `seen` (0 = the first one)
            const volatile long long *flag = reinterpret_cast<const volatile long long *>(pin_scal + 8);
            unsigned spins = 0;
            while (*flag < seen + 1) {
                if (++spins > (1u << 26)) {
                    ALVA_HIP(alva_stream_sync(st));
                    break;
                }
                alva_poll_relax(spins);
            }
            const long long f = *flag;
            __atomic_thread_fence(__ATOMIC_ACQUIRE);
            memcpy(pub, pin_scal + ((f & 1) ? 16 : 0), sizeof(pub));   // (the newest publication: `seen` may jump by two)
            seen = f;
            if (pub[0] == 0.0) break;                 // the minimiser has stopped; what is still queued returns at once
            if (issued < max_iters) {                 // the iteration behind the one that is running now
                step_lm();
                issued++;
                ALVA_LAUNCH_CHECK();
            } else if (seen >= (long long) max_iters + 1) {
                break;                                // (cannot happen: the last allowed iteration's evaluation publishes step = 0)
            }
        }
        H.nsummaries = (int) pub[2]; H.initial = pub[3]; H.x_cost = pub[4]; H.nsucc = (int) pub[5]; H.ok = (int) pub[6];
        cur_set = ((int) pub[7]) & 0xff;
        last_set = ((int) pub[7]) >> 8;
    } else {
    // ---- Ceres TrustRegionMinimizer::Minimize, restated (trust_region_minimizer.cc:67-136) -------------------
    rc = eval(H.d_xp, H.d_xt, true);
    if (rc) return rc;
    rc = read_scal();
    if (rc) return rc;
    H.x_cost = H.initial = scal[0];
    H.gmax = scal[3];
    LmState &lm = H.lm;
    // need_restore: the Jacobian-derived state belongs to a rejected candidate (restored lazily: when the loop ends right after a
    // rejection, the last evaluation stays the candidate's, as in the reference)
    while (true) {
        if (H.loop_ends(max_iters)) break;
        H.iteration++;
        if (H.need_restore) {
            rc = eval(H.xp, H.xt, false);
            if (rc) return rc;
            H.need_restore = false;
        }
        const int refresh = lm.reuse_diagonal ? 0 : 1;   // the LM diagonal is refreshed inside k_prep / k_reduced_system
        lm.reuse_diagonal = 1;
        if (n_pt > 0) {
            if (dp == 1) hipLaunchKernelGGL(k_prep<1>, gPt, blk, 0, st, B, lm.radius, refresh);
            else hipLaunchKernelGGL(k_prep<3>, gPt, blk, 0, st, B, lm.radius, refresh);
        }
        const int tiles = B.NP / 16;
        hipLaunchKernelGGL(k_gemm, dim3((unsigned) (tiles * tiles), KSPLIT), dim3(64), 0, st, B);
        if (np16 > 0) {
            const int np16i = (int) np16;
            hipLaunchKernelGGL(k_reduced_system, dim3((unsigned) alva_divup(np16i * np16i + np16i, 256)), blk, 0, st, B, lm.radius, refresh);
        }
        if (solve_in_lds) hipLaunchKernelGGL(k_solve<true>, dim3(1), dim3(SOLVE_NT), solve_lds, st, B, lm.radius);
        else hipLaunchKernelGGL(k_solve<false>, dim3(1), dim3(SOLVE_NT), 0, st, B, lm.radius);
        {   // the points' back-substitution + ONE more workgroup for the candidate poses
            const dim3 gBs(gPt.x + 1);
            if (dp == 1) hipLaunchKernelGGL(k_backsub<1>, n_pt > 0 ? gBs : dim3(1), blk, 0, st, B, lm.radius, (const double *) H.xt, H.ct, (const double *) H.xp, H.cp);
            else hipLaunchKernelGGL(k_backsub<3>, n_pt > 0 ? gBs : dim3(1), blk, 0, st, B, lm.radius, (const double *) H.xt, H.ct, (const double *) H.xp, H.cp);
        }
        ALVA_LAUNCH_CHECK();
        // The candidate is evaluated WITH its Jacobian and its norm straight away: when the step is accepted (the normal case) Ceres
        // re-evaluates at the same point (HandleSuccessfulStep, trust_region_minimizer.cc:809-829) and would produce exactly these
        // numbers again, so one host round trip per iteration disappears.  A rejected step costs one re-evaluation at x instead.
        rc = eval(H.cp, H.ct, false);
        if (rc) return rc;
        rc = read_scal();
        if (rc) return rc;
        if (H.advance(scal, function_tolerance)) break;
    }
    }
    *h_ok = H.ok;
    const auto t_lm = std::chrono::steady_clock::now();
    // results: poses / points at the last accepted x; chi2 / depth flags of the LAST evaluation (what the
    // reference's outlier sweep reads from its cost-function objects, optimizer.cpp:266-309)
    // the input staging area is free again (its upload finished long ago): results land there, three DMA copies
    uint8_t *r_chi = stage;
    const size_t bad_words = csr ? (nObs + 63) / 64 : 0;
    // (device-side loop: the last evaluation's outputs and x live in the set its final publication named)
    const double *res_chi2 = last_set ? H.B2.chi2 : B.chi2;
    const uint8_t *res_depth = last_set ? H.B2.depth : B.depth;
    if (dev_lm) {
        H.xp = cur_set ? H.d_cp : H.d_xp;
        H.xt = cur_set ? H.d_ct : H.d_xt;
    }
    if (csr && n_obs) {
        hipLaunchKernelGGL(k_bad_bits, dim3((unsigned) alva_divup(n_obs, 256)), dim3(256), 0, st, res_chi2, res_depth, n_obs,
                           csr->chi2_threshold, H.d_badBits);
    }
    const size_t chi_bytes = csr ? bad_words * 8 : H.chi_bytes();
    double *r_poses = reinterpret_cast<double *>(stage + (chi_bytes + 255) / 256 * 256);
    double *r_pts = r_poses + (size_t) n_kf * 7 + 32;
    if (poll) {
        // (chi_bytes, the pose block and the point block are multiples of 8 bytes or are rounded up inside their 256-byte carved slots)
        BaResultsArgs R{};
        R.src[0] = csr ? H.d_badBits : reinterpret_cast<const unsigned long long *>(res_chi2); R.dst[0] = reinterpret_cast<unsigned long long *>(r_chi);
        R.words[0] = n_obs ? (chi_bytes + 7) / 8 : 0;
        R.src[1] = reinterpret_cast<const unsigned long long *>(H.xp); R.dst[1] = reinterpret_cast<unsigned long long *>(r_poses);
        R.words[1] = (size_t) n_kf * 7;
        R.src[2] = reinterpret_cast<const unsigned long long *>(H.xt); R.dst[2] = reinterpret_cast<unsigned long long *>(r_pts);
        R.words[2] = npd;
        R.counter = ctx->d_counters + 1;   // slot 1 (slot 0 belongs to the P3P selection)
        R.word = reinterpret_cast<long long *>(pin_scal + 9);
        R.seq = ++eval_seq;
        reinterpret_cast<volatile long long *>(pin_scal + 9)[0] = 0;
        const size_t words = R.words[0] + R.words[1] + R.words[2];
        const unsigned g = (unsigned) std::min<size_t>(64, std::max<size_t>(1, (words + 2047) / 2048));
        hipLaunchKernelGGL(k_results, dim3(g), dim3(256), 0, st, R);
        ALVA_LAUNCH_CHECK();
        const volatile long long *flag = reinterpret_cast<const volatile long long *>(pin_scal + 9);
        unsigned spins = 0;
        while (*flag != eval_seq) {
            if (++spins > (1u << 26)) {
                ALVA_HIP(alva_stream_sync(st));
                break;
            }
            alva_poll_relax(spins);
        }
        __atomic_thread_fence(__ATOMIC_ACQUIRE);
    } else {
        if (n_obs) ALVA_HIP(hipMemcpyAsync(r_chi, csr ? (const void *) H.d_badBits : (const void *) res_chi2, chi_bytes, hipMemcpyDeviceToHost, st));
        ALVA_HIP(hipMemcpyAsync(r_poses, H.xp, (size_t) n_kf * 56, hipMemcpyDeviceToHost, st));
        if (npd) ALVA_HIP(hipMemcpyAsync(r_pts, H.xt, npd * 8, hipMemcpyDeviceToHost, st));
        ALVA_HIP(alva_stream_sync(st));
    }
    if (csr) {
        // poses / point parameters as finish() returns them; the sweep's flags as bits
        if (npd) memcpy(in.h_pt_param, r_pts, npd * 8);
        for (int k = 0; k < n_kf; k++)
            if (H.cidx[(size_t) k] >= 0) memcpy(in.h_poses + 7 * k, r_poses + 7 * (size_t) k, 56);
        int n_bad = 0;
        const unsigned long long *bits = reinterpret_cast<const unsigned long long *>(r_chi);
        for (size_t w = 0; w < bad_words; w++) {
            unsigned long long v = bits[w];
            if (w == bad_words - 1 && (nObs & 63)) v &= (1ull << (nObs & 63)) - 1ull;
            csr->h_bad_bits[w] = v;
            n_bad += __builtin_popcountll(v);
        }
        if (csr->h_n_bad) *csr->h_n_bad = n_bad;
        if (h_info) {
            h_info[0] = H.nsummaries;
            h_info[1] = H.initial;
            h_info[2] = H.x_cost;
            h_info[3] = H.nsucc;
        }
    } else
    H.finish(in, r_chi, r_poses, r_pts, h_chi2, h_depth_pos, h_info);
    if (getenv("ALVA_BA_TIMING")) {
        auto us = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) {
            return (double) std::chrono::duration_cast<std::chrono::nanoseconds>(b - a).count() * 1e-3;
        };
        fprintf(stderr, "[alva_local_ba] host structure %.0f us (sizes + arenas %.0f, stream sync %.0f, build %.0f) | scratch + upload %.0f us | LM loop (%d summaries) %.0f us | download %.0f us\n",
                us(t_begin, t_built), us(t_begin, t_sized), us(t_sized, t_synced), us(t_synced, t_built), us(t_built, t_up), H.nsummaries, us(t_up, t_lm),
                us(t_lm, std::chrono::steady_clock::now()));
    }
    return ALVA_OK;
}

extern "C" int alva_local_ba(alva_ctx *ctx, int n_kf, double *h_poses, const uint8_t *h_kf_const, const double *h_calib, int inv_depth,
                             int n_pt, const int *h_pt_anchor_kf, const double *h_pt_anchor_uv, double *h_pt_param, int n_obs,
                             const int *h_obs_kf, const int *h_obs_pt, const double *h_obs_uv, int max_iters,
                             double function_tolerance, double huber_chi2, double *h_chi2, uint8_t *h_depth_pos, double *h_info,
                             int *h_ok) {
    ALVA_ARG(ctx && h_poses && h_kf_const && h_calib && h_pt_param && h_ok && n_kf > 0 && n_pt >= 0 && n_obs >= 0 && max_iters >= 0);
    ALVA_ARG(n_obs == 0 || (h_obs_kf && h_obs_pt && h_obs_uv));
    ALVA_ARG(!inv_depth || n_pt == 0 || (h_pt_anchor_kf && h_pt_anchor_uv));
    const BaIn in{n_kf, h_poses, h_kf_const, h_calib, inv_depth, n_pt, h_pt_anchor_kf, h_pt_anchor_uv, h_pt_param, n_obs, h_obs_kf, h_obs_pt, h_obs_uv,
                  huber_chi2};
    return ba_drive(ctx, in, nullptr, max_iters, function_tolerance, h_chi2, h_depth_pos, h_info, h_ok);
}

extern "C" int alva_local_ba_csr(alva_ctx *ctx, int n_kf, double *h_poses, const uint8_t *h_kf_const, const double *h_calib, int n_pt,
                                 const int *h_pt_ptr, const int *h_pt_anchor_kf, const double *h_pt_anchor_uv, double *h_pt_inv_depth, int n_obs,
                                 const int *h_obs_kf, const double *h_obs_uv, int max_iters, double function_tolerance, double huber_chi2,
                                 double chi2_threshold, unsigned long long *h_bad_bits, int *h_n_bad, double *h_info, int *h_ok) {
    ALVA_ARG(ctx && h_poses && h_kf_const && h_calib && h_pt_inv_depth && h_ok && h_pt_ptr && h_bad_bits && n_kf > 0 && n_pt >= 0 && n_obs >= 0 &&
             max_iters >= 0);
    ALVA_ARG(n_obs == 0 || (h_obs_kf && h_obs_uv));
    ALVA_ARG(n_pt == 0 || (h_pt_anchor_kf && h_pt_anchor_uv));
    const BaIn in{n_kf, h_poses, h_kf_const, h_calib, 1, n_pt, h_pt_anchor_kf, h_pt_anchor_uv, h_pt_inv_depth, n_obs, h_obs_kf, nullptr, h_obs_uv, huber_chi2};
    const BaCsr csr{h_pt_ptr, chi2_threshold, h_bad_bits, h_n_bad};
    return ba_drive(ctx, in, &csr, max_iters, function_tolerance, nullptr, nullptr, h_info, h_ok);
}

// `count` independent local-BA problems (anchored inverse depth) in ONE set of launches per LM iteration: a rig's cameras, or the
// sessions of a server, each with its own keyframes / points / observations (ragged).  Every kernel carries the problem in a grid
// dimension -- the reduced camera systems factor on `count` compute units at once, the Schur-complement GEMMs form one grouped MFMA
// launch -- and the host reads ONE block of scalars per iteration for all problems and steps each problem's trust region separately
// (problems stop at different iterations; a stopped problem's workgroups return at once).  Results are bit-identical to `count`
// calls of alva_local_ba.  Arrays of `count` pointers / sizes; h_info [count][4]; h_ok [count].
extern "C" int alva_local_ba_batch(alva_ctx *ctx, int count, const int *n_kf, double *const *h_poses, const uint8_t *const *h_kf_const,
                                   const double *h_calib, const int *n_pt, const int *const *h_pt_anchor_kf,
                                   const double *const *h_pt_anchor_uv, double *const *h_pt_param, const int *n_obs, const int *const *h_obs_kf,
                                   const int *const *h_obs_pt, const double *const *h_obs_uv, int max_iters, double function_tolerance,
                                   double huber_chi2, double *const *h_chi2, uint8_t *const *h_depth_pos, double *h_info, int *h_ok) {
    ALVA_ARG(ctx && count > 0 && count <= 4096 && n_kf && h_poses && h_kf_const && h_calib && n_pt && h_pt_anchor_kf && h_pt_anchor_uv &&
             h_pt_param && n_obs && h_obs_kf && h_obs_pt && h_obs_uv && h_ok && max_iters >= 0);
    std::vector<BaIn> ins((size_t) count);
    std::vector<BaHost> Hs((size_t) count);
    size_t dev_bytes = 0, in_total = 0, res_total = 0;
    int max_pt = 1, max_kf = 1, max_n6 = 0, max_ndiag = 1, max_tiles = 1;
    size_t max_np16 = 0, max_lds = 0;
    for (int b = 0; b < count; b++) {
        ALVA_ARG(n_kf[b] > 0 && n_pt[b] > 0 && n_obs[b] > 0 && h_poses[b] && h_kf_const[b] && h_pt_anchor_kf[b] && h_pt_anchor_uv[b] && h_pt_param[b] &&
                 h_obs_kf[b] && h_obs_pt[b] && h_obs_uv[b]);
        ins[(size_t) b] = BaIn{n_kf[b], h_poses[b], h_kf_const[b], h_calib, 1, n_pt[b], h_pt_anchor_kf[b], h_pt_anchor_uv[b], h_pt_param[b], n_obs[b],
                               h_obs_kf[b], h_obs_pt[b], h_obs_uv[b], huber_chi2};
        BaHost &H = Hs[(size_t) b];
        int rc = H.sizes(ins[(size_t) b]);
        if (rc) return rc;
        ALVA_ARG(H.solve_lds <= 152 * 1024);   // the batched factorisation keeps every reduced system in LDS (<= 23 free keyframes)
        dev_bytes += H.bytes;
        in_total += H.in_bytes;
        res_total += (H.chi_bytes() + 255) / 256 * 256 + ((size_t) n_kf[b] * 56 + 255) / 256 * 256 + ((size_t) H.B.npd * 8 + 255) / 256 * 256;
        max_pt = std::max(max_pt, n_pt[b]);
        max_kf = std::max(max_kf, n_kf[b]);
        max_n6 = std::max(max_n6, H.B.n6);
        max_ndiag = std::max(max_ndiag, std::max(H.B.n6, H.B.npd));
        max_tiles = std::max(max_tiles, H.B.NP / 16);
        max_np16 = std::max(max_np16, H.np16);
        max_lds = std::max(max_lds, H.solve_lds);
        h_ok[b] = 1;
    }
    const size_t cnt = (size_t) count;
    const size_t off_desc = dev_bytes, off_run = off_desc + (cnt * sizeof(BaDev) + 255) / 256 * 256, off_scal = off_run + (cnt * sizeof(BaRun) + 255) / 256 * 256;
    uint8_t *base = nullptr;
    int rc = alva_ctx_scratch(ctx, 4, off_scal + cnt * 64, (void **) &base);
    if (rc) return rc;
    // pinned: scalars of all problems | run blocks | descriptors | input mirrors (later: results)
    const size_t p_run = (cnt * 64 + 255) / 256 * 256, p_desc = p_run + (cnt * sizeof(BaRun) + 255) / 256 * 256,
                 p_stage = p_desc + (cnt * sizeof(BaDev) + 255) / 256 * 256;
    uint8_t *pin = nullptr;
    rc = alva_ctx_pinned(ctx, p_stage + std::max(in_total, res_total) + 256, (void **) &pin);
    if (rc) return rc;
    hipStream_t st = ctx->stream;
    ALVA_HIP(alva_stream_sync(st));
    BaDev *d_desc = (BaDev *) (base + off_desc), *h_desc = (BaDev *) (pin + p_desc);
    BaRun *d_run = (BaRun *) (base + off_run), *h_run = (BaRun *) (pin + p_run);
    double *d_scal = (double *) (base + off_scal), *h_scal = (double *) pin;
    {
        size_t doff = 0, soff = 0;
        for (int b = 0; b < count; b++) {
            BaHost &H = Hs[(size_t) b];
            rc = H.build(ins[(size_t) b], base + doff, pin + p_stage + soff);
            if (rc) return rc;
            H.B.scal = d_scal + 8 * (size_t) b;   // every problem's scalars in one block: one read-back per iteration
            h_desc[b] = H.B;
            ALVA_HIP(hipMemcpyAsync(base + doff, pin + p_stage + soff, H.in_bytes, hipMemcpyHostToDevice, st));
            ALVA_HIP(hipMemsetAsync(H.B.Wt, 0, (size_t) H.B.npd * H.B.NP * 8, st));
            ALVA_HIP(hipMemsetAsync(H.B.Zt, 0, (size_t) H.B.kpad * H.B.NP * 8, st));
            doff += H.bytes;
            soff += H.in_bytes;
        }
    }
    ALVA_HIP(hipMemcpyAsync(d_desc, h_desc, cnt * sizeof(BaDev), hipMemcpyHostToDevice, st));
    if (max_lds > 48 * 1024)
        ALVA_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_solve_b), hipFuncAttributeMaxDynamicSharedMemorySize, 152 * 1024));
    const dim3 blk(256);
    const unsigned ub = (unsigned) count;
    auto push_run = [&]() -> int {
        for (int b = 0; b < count; b++) {
            BaHost &H = Hs[(size_t) b];
            BaRun &R = h_run[b];
            R.radius = H.lm.radius;
            R.xp = H.xp; R.xt = H.xt; R.cp = H.cp; R.ct = H.ct;
            R.ev_p[0] = R.ev_p[1] = H.xp; R.ev_t[0] = R.ev_t[1] = H.xt;
            R.ev_p[2] = H.cp; R.ev_t[2] = H.ct;
        }
        ALVA_HIP(hipMemcpyAsync(d_run, h_run, cnt * sizeof(BaRun), hipMemcpyHostToDevice, st));
        return ALVA_OK;
    };
    auto eval = [&](int mode) -> int {
        hipLaunchKernelGGL(k_point_b, dim3((unsigned) alva_divup(max_pt, 4), ub), blk, 0, st, d_desc, d_run, mode);
        hipLaunchKernelGGL(k_pairs_b, dim3((unsigned) (max_kf * max_kf), ub), blk, 0, st, d_desc, d_run, mode);
        hipLaunchKernelGGL(k_assemble_b, dim3(ub), dim3(ASM_NT), assemble_lds(max_kf), st, d_desc, d_run, mode);
        ALVA_LAUNCH_CHECK();
        return ALVA_OK;
    };
    auto read_scal = [&]() -> int {
        ALVA_HIP(hipMemcpyAsync(h_scal, d_scal, cnt * 64, hipMemcpyDeviceToHost, st));
        ALVA_HIP(alva_stream_sync(st));
        return ALVA_OK;
    };
    for (int b = 0; b < count; b++) {
        h_run[b] = BaRun{};
        h_run[b].ev_on[0] = 1;
    }
    rc = push_run();
    if (rc) return rc;
    rc = eval(0);
    if (rc) return rc;
    rc = read_scal();
    if (rc) return rc;
    for (int b = 0; b < count; b++) {
        BaHost &H = Hs[(size_t) b];
        H.x_cost = H.initial = h_scal[8 * (size_t) b];
        H.gmax = h_scal[8 * (size_t) b + 3];
    }
    for (;;) {
        int active = 0, restores = 0, diags = 0;
        for (int b = 0; b < count; b++) {
            BaHost &H = Hs[(size_t) b];
            BaRun &R = h_run[b];
            if (!H.done && H.loop_ends(max_iters)) H.done = true;
            R.step = R.diag = 0;
            R.ev_on[0] = R.ev_on[1] = R.ev_on[2] = 0;
            if (H.done) continue;
            H.iteration++;
            R.step = R.ev_on[2] = 1;
            R.ev_on[1] = H.need_restore ? 1 : 0;
            H.need_restore = false;
            R.diag = H.lm.reuse_diagonal ? 0 : 1;
            H.lm.reuse_diagonal = 1;
            active++;
            restores += R.ev_on[1];
            diags += R.diag;
        }
        if (!active) break;
        rc = push_run();
        if (rc) return rc;
        if (restores) {
            rc = eval(1);
            if (rc) return rc;
        }
        (void) diags;   // the diagonal refresh rides in k_prep_b / k_reduced_system_b (BaRun::diag)
        hipLaunchKernelGGL(k_prep_b, dim3((unsigned) alva_divup(max_pt, 4), ub), blk, 0, st, d_desc, d_run);
        hipLaunchKernelGGL(k_gemm_b, dim3((unsigned) (max_tiles * max_tiles), KSPLIT, ub), dim3(64), 0, st, d_desc, d_run);
        if (max_np16 > 0) {
            const int n16 = (int) max_np16;
            hipLaunchKernelGGL(k_reduced_system_b, dim3((unsigned) alva_divup(n16 * n16 + n16, 256), ub), blk, 0, st, d_desc, d_run);
        }
        hipLaunchKernelGGL(k_solve_b, dim3(ub), dim3(SOLVE_NT), max_lds, st, d_desc, d_run);
        hipLaunchKernelGGL(k_backsub_b, dim3((unsigned) alva_divup(max_pt, 4) + 1, ub), blk, 0, st, d_desc, d_run);
        ALVA_LAUNCH_CHECK();
        rc = eval(2);
        if (rc) return rc;
        rc = read_scal();
        if (rc) return rc;
        for (int b = 0; b < count; b++) {
            BaHost &H = Hs[(size_t) b];
            if (!h_run[b].step) continue;
            if (H.advance(h_scal + 8 * (size_t) b, function_tolerance)) H.done = true;
        }
    }
    // results: per problem chi2 | depth, poses, point parameters -> the (free again) input staging area
    {
        size_t roff = 0;
        std::vector<size_t> o_chi((size_t) count), o_pose((size_t) count), o_pts((size_t) count);
        for (int b = 0; b < count; b++) {
            BaHost &H = Hs[(size_t) b];
            o_chi[(size_t) b] = roff;
            ALVA_HIP(hipMemcpyAsync(pin + p_stage + roff, H.B.chi2, H.chi_bytes(), hipMemcpyDeviceToHost, st));
            roff += (H.chi_bytes() + 255) / 256 * 256;
            o_pose[(size_t) b] = roff;
            ALVA_HIP(hipMemcpyAsync(pin + p_stage + roff, H.xp, (size_t) n_kf[b] * 56, hipMemcpyDeviceToHost, st));
            roff += ((size_t) n_kf[b] * 56 + 255) / 256 * 256;
            o_pts[(size_t) b] = roff;
            ALVA_HIP(hipMemcpyAsync(pin + p_stage + roff, H.xt, (size_t) H.B.npd * 8, hipMemcpyDeviceToHost, st));
            roff += ((size_t) H.B.npd * 8 + 255) / 256 * 256;
        }
        ALVA_HIP(alva_stream_sync(st));
        for (int b = 0; b < count; b++) {
            BaHost &H = Hs[(size_t) b];
            H.finish(ins[(size_t) b], pin + p_stage + o_chi[(size_t) b], (const double *) (pin + p_stage + o_pose[(size_t) b]),
                     (const double *) (pin + p_stage + o_pts[(size_t) b]), h_chi2 ? h_chi2[b] : nullptr, h_depth_pos ? h_depth_pos[b] : nullptr,
                     h_info ? h_info + 4 * (size_t) b : nullptr);
            h_ok[b] = H.ok;
        }
    }
    return ALVA_OK;
}
