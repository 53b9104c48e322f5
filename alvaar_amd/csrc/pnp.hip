// a9: robust PnP refinement (motion-only bundle adjustment) -- the whole Levenberg-Marquardt loop in
// ONE persistent workgroup.
//
// Replaces MultiViewGeometry::ceresPnP (src/slam/src/multi_view_geometry.cpp:129-223): Ceres LM +
// Huber on the 6-DoF pose with cost ReprojectionErrorSE3 (ceres_parametrization.cpp:96-155), chi2 /
// depth outlier sweep, optional L2 re-solve.  The minimiser control flow is Ceres 2.0's
// (trust_region_minimizer.cc:67-136, 244-311, 377-451, 744-829; levenberg_marquardt_strategy.cc:66-160)
// with the reference's 5 ms wall-clock cap removed (it makes the reference non-deterministic).
//
// Why one workgroup: the problem is N <= a few thousand 2x6 Jacobian rows and a 6x6 solve per iteration;
// a multi-launch version would pay ~10 dependent kernel boundaries + host round trips (~1.5-5 us each,
// MI355X_MICROARCH "boundary" row) for ~1 us of FP64 work per evaluation.  Here 512 threads evaluate
// residuals/Jacobians and reduce the 28 normal-equation scalars (21 of J^T J, 6 of J^T r, cost) through
// a butterfly reduce-scatter + LDS; lane 0 does the 6x6 Cholesky and the trust-region bookkeeping; nothing
// leaves the CU until the pose is final.  FP64 throughout (the reference is all double).
//
// An evaluation is FP64-issue bound on the one CU (~13 k wave instructions for 2 k points), so this translation unit
// allows fused multiply-add contraction: the normal-equation accumulation is half the instructions as FMAs.  Results
// are compared with the reference under a tolerance (FP64 sums in a different order anyway), not bitwise.
#include "common.hpp"
#include "multi_kernel.hpp"
#include "p3p_device.hpp"   // (the fused P3P -> PnP launch below; in front of the pragma: P3P is compiled without contraction)
#include "track_compact_device.hpp"
#pragma clang fp contract(fast)
#include "lm_device.hpp"
#include "wave_utils.hpp"
#include "pose_internal.hpp"
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdlib>

namespace {

constexpr int NT = 512;   // 8 waves: 256 VGPRs per lane keep the unrolled 6x6 minimiser step and the 28 accumulators out of scratch
constexpr int NW = NT / 64;
constexpr int NACC = 28;  // H upper triangle (21) | g (6) | cost (1)

struct PnpArgs {
    const double *uv, *wpt;
    int n;
    double K[4];
    double huber_a;
    double chi2_th;
    int use_robust, apply_l2, max_iters;
    double ftol;
    double pose0[7];  // initial pose (ignored in chained mode)
    int seq;          // k_pnp publishes it in PnpOut::seq after every result (system-scope release): a host may poll that word
    unsigned long long *dbg;   // phase stamps (alva_kstamp_buffer) or null
};
// sequential stamps from entry 2048 (entry 2047 = how many)
#define PNP_STAMP() do { if (A.dbg && threadIdx.x == 0 && sh.nstamp < 2000) A.dbg[2048 + sh.nstamp++] = wall_clock64(); } while (0)

struct PnpOut {
    double pose[7];
    double info[8];
    int ok, n_bad;
    int p3p_ok, n_active;  // chained mode: P3P verdict and the number of points handed to the refinement
    int p3p_n_valid_used, seq;
    double pose_p3p[7];  // chained mode: the accepted P3P pose as the refinement starts from it (normalised quaternion)
};

// LmState of lm_device.hpp without member initialisers (it lives in LDS)
struct LmPod {
    double radius, decrease_factor;
    int reuse_diagonal;
    __device__ __forceinline__ void reset() {
        radius = 1e4;
        decrease_factor = 2.0;
        reuse_diagonal = 0;
    }
    __device__ __forceinline__ void accepted(double q) {
        LmState t;
        t.radius = radius; t.decrease_factor = decrease_factor; t.reuse_diagonal = reuse_diagonal;
        t.accepted(q);
        radius = t.radius; decrease_factor = t.decrease_factor; reuse_diagonal = t.reuse_diagonal;
    }
    __device__ __forceinline__ void rejected() {
        LmState t;
        t.radius = radius; t.decrease_factor = decrease_factor; t.reuse_diagonal = reuse_diagonal;
        t.rejected();
        radius = t.radius; decrease_factor = t.decrease_factor; reuse_diagonal = t.reuse_diagonal;
    }
};

struct PnpShared {
    double part[NW][NACC];
    double acc[NACC];
    double x[7], cand[7];
    double T[12];   // R_wc row-major | t_wc of the pose the NEXT evaluation uses (written by lane 0 beside the step)
    int flag, nstamp;
    // minimiser state (touched by lane 0 only; kept in LDS so that dynamic indexing never goes to scratch memory)
    double H[36], g[6], scale[6], diag[6];
    // ... and its scalars: loop-carried in every thread's registers they cost ~20 VGPRs of the 256 and pushed the evaluation's
    // accumulators into scratch around the point loop
    double x_cost, gmax, x_norm, initial, mcc;
    LmPod lm;
    int iteration, nsucc, invalid, nsummaries;
};

enum { F_DONE = 0, F_EVAL_CAND = 1, F_FAIL = 4 };

__device__ __forceinline__ int tri(int a, int b) {  // index of (a,b), a <= b, in the packed upper triangle of a 6x6
    return a * 6 - a * (a - 1) / 2 + (b - a);
}

// ---- lane 0's arithmetic between two evaluations ---------------------------------------------------------------------------------
// One lane runs the trust-region step while 511 wait, so its LATENCY is kernel time (stamps: 5.4 us per step with IEEE divides and
// square roots -- 39 of them in a 6x6 Cholesky solve, ~100 ns each as a dependent chain).  Reciprocals and reciprocal square roots are
// therefore the hardware estimates refined by two Newton steps (<= 1-2 ulp; the results of this kernel are compared under a
// tolerance, the iteration itself is Ceres', not bit-for-bit Eigen), each used once per pivot and multiplied in.
__device__ __forceinline__ double fast_rcp(double x) { return alva_fast_rcp(x); }
__device__ __forceinline__ double fast_rsqrt(double x, double &g) { return alva_fast_rsqrt(x, g); }   // g ~ sqrt(x), returns 1 / sqrt(x)
__device__ __forceinline__ double fast_sqrt(double x) {
    double g;
    (void) fast_rsqrt(x, g);
    return x > 0 ? g : 0.0;
}

// Cholesky solve of the damped 6x6 system, all loops unrolled on registers; one reciprocal square root per pivot
__device__ __forceinline__ bool chol6_solve(double (&A)[36], double (&b)[6]) {
    double inv[6];
#pragma unroll
    for (int j = 0; j < 6; j++) {
        double d = A[7 * j];
#pragma unroll
        for (int k = 0; k < j; k++) d -= A[6 * j + k] * A[6 * j + k];
        if (!(d > 0)) return false;
        double root;
        inv[j] = fast_rsqrt(d, root);
#pragma unroll
        for (int i = j + 1; i < 6; i++) {
            double v = A[6 * i + j];
#pragma unroll
            for (int k = 0; k < j; k++) v -= A[6 * i + k] * A[6 * j + k];
            A[6 * i + j] = v * inv[j];
        }
    }
#pragma unroll
    for (int i = 0; i < 6; i++) {
        double v = b[i];
#pragma unroll
        for (int k = 0; k < i; k++) v -= A[6 * i + k] * b[k];
        b[i] = v * inv[i];
    }
#pragma unroll
    for (int i = 5; i >= 0; i--) {
        double v = b[i];
#pragma unroll
        for (int k = i + 1; k < 6; k++) v -= A[6 * k + i] * b[k];
        b[i] = v * inv[i];
    }
    return true;
}

// R_wc | t_wc of pose7 (quaternion normalised as Sophus does on construction)
__device__ __forceinline__ void pose_to_T(const double *p, double *T12) {
    double root;
    const double inv = fast_rsqrt(p[3] * p[3] + p[4] * p[4] + p[5] * p[5] + p[6] * p[6], root);
    const double q[4] = {p[3] * inv, p[4] * inv, p[5] * inv, p[6] * inv};
    quat_to_R(q, T12);
    T12[9] = p[0]; T12[10] = p[1]; T12[11] = p[2];
}

// out7 = Exp(d6) * x7 (se3_plus of lm_device.hpp with the fast reciprocals; sin / cos by one sincos per angle)
__device__ __forceinline__ void se3_plus_fast(const double *x7, const double *d6, double *out7) {
    double root;
    const double invn = fast_rsqrt(x7[3] * x7[3] + x7[4] * x7[4] + x7[5] * x7[5] + x7[6] * x7[6], root);
    const double b4[4] = {x7[3] * invn, x7[4] * invn, x7[5] * invn, x7[6] * invn};
    const double *u = d6, *w = d6 + 3;
    const double th2 = w[0] * w[0] + w[1] * w[1] + w[2] * w[2];
    double imag, real, Rd[9], V[9];
    const bool small = th2 < 1e-10 * 1e-10;
    double theta = 0, inv_theta = 0;
    if (small) {
        const double th4 = th2 * th2;
        imag = 0.5 - (1.0 / 48.0) * th2 + (1.0 / 3840.0) * th4;
        real = 1 - (1.0 / 8.0) * th2 + (1.0 / 384.0) * th4;
    } else {
        inv_theta = fast_rsqrt(th2, theta);
        double sh_, ch_;
        sincos(0.5 * theta, &sh_, &ch_);
        imag = sh_ * inv_theta;
        real = ch_;
    }
    const double qd[4] = {imag * w[0], imag * w[1], imag * w[2], real};
    quat_to_R(qd, Rd);
    if (small) {   // theta < 1e-10
#pragma unroll
        for (int i = 0; i < 9; i++) V[i] = Rd[i];
    } else {
        const double O[9] = {0, -w[2], w[1], w[2], 0, -w[0], -w[1], w[0], 0};
        double st, ct;
        sincos(theta, &st, &ct);
        const double inv_th2 = inv_theta * inv_theta;
        const double a = (1 - ct) * inv_th2, b = (theta - st) * (inv_th2 * inv_theta);
#pragma unroll
        for (int r = 0; r < 3; r++)
#pragma unroll
            for (int c = 0; c < 3; c++) {
                const double o2 = O[3 * r] * O[c] + O[3 * r + 1] * O[3 + c] + O[3 * r + 2] * O[6 + c];
                V[3 * r + c] = (r == c ? 1.0 : 0.0) + a * O[3 * r + c] + b * o2;
            }
    }
    const double *a4 = qd;
    const double qn[4] = {a4[3] * b4[0] + a4[0] * b4[3] + a4[1] * b4[2] - a4[2] * b4[1],
                          a4[3] * b4[1] + a4[1] * b4[3] + a4[2] * b4[0] - a4[0] * b4[2],
                          a4[3] * b4[2] + a4[2] * b4[3] + a4[0] * b4[1] - a4[1] * b4[0],
                          a4[3] * b4[3] - a4[0] * b4[0] - a4[1] * b4[1] - a4[2] * b4[2]};
    const double inv2 = fast_rsqrt(qn[0] * qn[0] + qn[1] * qn[1] + qn[2] * qn[2] + qn[3] * qn[3], root);
#pragma unroll
    for (int i = 0; i < 3; i++)
        out7[i] = (V[3 * i] * u[0] + V[3 * i + 1] * u[1] + V[3 * i + 2] * u[2]) + (Rd[3 * i] * x7[0] + Rd[3 * i + 1] * x7[1] + Rd[3 * i + 2] * x7[2]);
#pragma unroll
    for (int i = 0; i < 4; i++) out7[3 + i] = qn[i] * inv2;
}

// Block-wide evaluation at the pose in sh.T: cost (+ packed J^T J and J^T r when WANT_J) into sh.acc.  Inlined at its call sites: the
// problem's scalars stay in SGPRs and the pointers keep their address space (as an out-of-line function every field of the argument
// struct was re-read through flat loads inside the point loop: four dependent memory round trips per point).  The next point's five
// doubles are requested before the current one is processed.
// The loop is FP64-issue bound on the one CU (a wave64 FP64 instruction takes 4 cycles; 2300 points x ~230 instructions / 4 SIMDs), and a
// third of those instructions were three IEEE divides / square roots per point: 1 / z, and Huber's sqrt(s), a / sqrt(s), sqrt(rho').
// They are the refined hardware estimates here (<= 2 ulp).
// Returns this thread's verdicts of multi_view_geometry.cpp:194-207 as a mask: bit k = point threadIdx.x + k * NT is active and has
// chi2 > chi2_th or a non-positive depth AT THIS evaluation (the reference reads the cost functors' members after the solve, i.e. the
// values of the minimiser's last evaluation) -- the outlier sweep needs no chi2 / depth arrays in memory.
template<bool WANT_J>
__device__ __forceinline__ unsigned long long eval(PnpShared &sh, const PnpArgs &A, int robust, const uint8_t *__restrict__ active) {
    PNP_STAMP();
    double R[9], t[3];
#pragma unroll
    for (int k = 0; k < 9; k++) R[k] = sh.T[k];
#pragma unroll
    for (int k = 0; k < 3; k++) t[k] = sh.T[9 + k];
    const double *__restrict__ wpt = A.wpt, *__restrict__ uvp = A.uv;
    const int n = A.n;
    const double K0 = A.K[0], K1 = A.K[1], K2 = A.K[2], K3 = A.K[3], ha = A.huber_a, hb = ha * ha, chi2_th = A.chi2_th;
    double acc[32];  // NACC sums + padding: reduced in place by wave_reduce_scatter32 (a second 32-entry copy does not fit the register file)
#pragma unroll
    for (int k = 0; k < 32; k++) acc[k] = 0.0;
    unsigned long long badmask = 0, bit = 1;
    int i = threadIdx.x;
    uint8_t a = 0;
    double X[3] = {0, 0, 1}, u = 0, v = 0;
    if (i < n) {
        a = active[i];
        X[0] = wpt[3 * i]; X[1] = wpt[3 * i + 1]; X[2] = wpt[3 * i + 2];
        u = uvp[2 * i]; v = uvp[2 * i + 1];
    }
    while (i < n) {
        const int in = i + NT;
        uint8_t an = 0;
        double Xn[3] = {0, 0, 1}, un = 0, vn = 0;
        if (in < n) {
            an = active[in];
            Xn[0] = wpt[3 * in]; Xn[1] = wpt[3 * in + 1]; Xn[2] = wpt[3 * in + 2];
            un = uvp[2 * in]; vn = uvp[2 * in + 1];
        }
        if (a) {
            // reproj of lm_device.hpp (ceres_parametrization.cpp:96-155)
            const double d0 = X[0] - t[0], d1 = X[1] - t[1], d2 = X[2] - t[2];
            const double c0 = R[0] * d0 + R[3] * d1 + R[6] * d2;
            const double c1 = R[1] * d0 + R[4] * d1 + R[7] * d2;
            const double c2 = R[2] * d0 + R[5] * d1 + R[8] * d2;
            const double iz = fast_rcp(c2);
            const double r0 = K0 * c0 * iz + K2 - u, r1 = K1 * c1 * iz + K3 - v;
            const double chi2 = r0 * r0 + r1 * r1;
            if (chi2 > chi2_th || !(c2 > 0)) badmask |= bit;
            double rho0 = chi2, s = 1.0;
            if (robust && chi2 > hb) {   // Huber (loss_function.cc:48-62) and the corrector's sqrt(rho') (corrector.cc:41-110)
                double root;
                const double ir = fast_rsqrt(chi2, root);
                rho0 = 2.0 * ha * root - hb;
                double rho1 = ha * ir;
                rho1 = rho1 > 2.2250738585072014e-308 ? rho1 : 2.2250738585072014e-308;
                s = fast_sqrt(rho1);
            }
            acc[27] += 0.5 * rho0;
            if (WANT_J) {
                const double iz2 = iz * iz;
                const double Jp[6] = {iz * K0, 0, -c0 * iz2 * K0, 0, iz * K1, -c1 * iz2 * K1};
                double JR[6], JH[6];
#pragma unroll
                for (int rr = 0; rr < 2; rr++)
#pragma unroll
                    for (int cc = 0; cc < 3; cc++)  // R_cw[k][cc] = R_wc[cc][k]
                        JR[3 * rr + cc] = Jp[3 * rr] * R[3 * cc] + Jp[3 * rr + 1] * R[3 * cc + 1] + Jp[3 * rr + 2] * R[3 * cc + 2];
                times_hat(JR, X, JH);
                double J[12];
#pragma unroll
                for (int rr = 0; rr < 2; rr++)
#pragma unroll
                    for (int c = 0; c < 3; c++) {
                        J[6 * rr + c] = -JR[3 * rr + c] * s;
                        J[6 * rr + 3 + c] = JH[3 * rr + c] * s;
                    }
                const double rs0 = r0 * s, rs1 = r1 * s;
                int tt = 0;
#pragma unroll
                for (int aa = 0; aa < 6; aa++) {
                    acc[21 + aa] += J[aa] * rs0 + J[6 + aa] * rs1;
#pragma unroll
                    for (int bb = aa; bb < 6; bb++) acc[tt++] += J[aa] * J[bb] + J[6 + aa] * J[6 + bb];
                }
            }
        }
        bit <<= 1;
        i = in;
        a = an;
        X[0] = Xn[0]; X[1] = Xn[1]; X[2] = Xn[2];
        u = un; v = vn;
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (WANT_J) {
        wave_reduce_scatter32(acc);  // lane l ends with the wave total of value l >> 1
        if (!(lane & 1) && (lane >> 1) < NACC) sh.part[wave][lane >> 1] = acc[0];
    } else {
        double vv = acc[27];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) vv += __shfl_down(vv, off);
        if (lane == 0) sh.part[wave][27] = vv;
    }
    __syncthreads();
    if (threadIdx.x < NACC && (WANT_J || threadIdx.x == 27)) {
        double vv = 0;
#pragma unroll
        for (int w = 0; w < NW; w++) vv += sh.part[w][threadIdx.x];
        sh.acc[threadIdx.x] = vv;
    }
    __syncthreads();
    PNP_STAMP();
    return badmask;
}

// ---- lane 0 between two evaluations: out-of-line on purpose ---------------------------------------------------------------------------
// These run on ONE lane; inlined into the kernel their ~200 live registers (two 6x6 matrices, the sincos expansions) are allocated on top
// of the evaluation loop's and the excess goes to scratch in both.  As functions they get their own allocation, the kernel keeps only
// pointers across the call, and all minimiser state lives in LDS (`sh` arrives as a generic pointer into it).

// steps until one is valid (trust_region_minimizer.cc:377-451): F_EVAL_CAND (candidate in sh.cand, its R | t in sh.T), F_DONE or F_FAIL
__device__ __forceinline__ int lm_next_step(PnpShared &sh, int max_iters) {
    double *H = sh.H, *g = sh.g, *scale = sh.scale, *diag = sh.diag;
    for (;;) {
        if (sh.iteration >= max_iters || sh.gmax <= 1e-10 || sh.lm.radius <= 1e-32) return F_DONE;
        sh.iteration++;
        // everything below is straight-line code on registers (all loops unrolled, constant indices)
        double Hr[36], gr[6], sc[6], Mr[36], yr[6];
#pragma unroll
        for (int a = 0; a < 6; a++) sc[a] = scale[a];
#pragma unroll
        for (int a = 0; a < 6; a++) {
            gr[a] = g[a] * sc[a];
#pragma unroll
            for (int b = 0; b < 6; b++) Hr[6 * a + b] = H[6 * a + b] * sc[a] * sc[b];
        }
        if (!sh.lm.reuse_diagonal) {
#pragma unroll
            for (int a = 0; a < 6; a++) diag[a] = fmin(fmax(Hr[7 * a], 1e-6), 1e32);
        }
#pragma unroll
        for (int k = 0; k < 36; k++) Mr[k] = Hr[k];
        const double inv_radius = fast_rcp(sh.lm.radius);
#pragma unroll
        for (int a = 0; a < 6; a++) {
            Mr[7 * a] += diag[a] * inv_radius;
            yr[a] = gr[a];
        }
        const bool okstep = chol6_solve(Mr, yr);
        sh.lm.reuse_diagonal = 1;
        double mcc = 0;
        if (okstep) {
            double sg = 0, sHs = 0;
#pragma unroll
            for (int a = 0; a < 6; a++) sg += -yr[a] * gr[a];
#pragma unroll
            for (int a = 0; a < 6; a++)
#pragma unroll
                for (int b = 0; b < 6; b++) sHs += yr[a] * Hr[6 * a + b] * yr[b];
            mcc = -sg - 0.5 * sHs;  // = -(J s)'(f + J s / 2), trust_region_minimizer.cc:419-431
        }
        sh.mcc = mcc;
        if (!okstep || !(mcc > 0)) {
            if (++sh.invalid >= 5) return F_FAIL;
            sh.lm.rejected();
            sh.nsummaries++;
            continue;
        }
        sh.invalid = 0;
        double delta[6], x7[7], cand[7];
#pragma unroll
        for (int a = 0; a < 6; a++) delta[a] = -yr[a] * sc[a];
#pragma unroll
        for (int a = 0; a < 7; a++) x7[a] = sh.x[a];
        se3_plus_fast(x7, delta, cand);
#pragma unroll
        for (int a = 0; a < 7; a++) sh.cand[a] = cand[a];
        double T12[12];
        pose_to_T(cand, T12);
#pragma unroll
        for (int a = 0; a < 12; a++) sh.T[a] = T12[a];
        return F_EVAL_CAND;
    }
}

// after the first evaluation of a solve: gradient / Hessian / cost in, Jacobi scaling (trust_region_minimizer.cc:266-275), first step
__device__ __noinline__ int lm_first_step(PnpShared *shp, int max_iters) {
    PnpShared &sh = *shp;
    double *H = sh.H, *g = sh.g;
#pragma unroll
    for (int a = 0; a < 6; a++)
#pragma unroll
        for (int b = a; b < 6; b++) H[6 * a + b] = H[6 * b + a] = sh.acc[tri(a, b)];
    double gmax = 0;
#pragma unroll
    for (int a = 0; a < 6; a++) {
        g[a] = sh.acc[21 + a];
        gmax = fmax(gmax, fabs(g[a]));
        sh.scale[a] = fast_rcp(1.0 + fast_sqrt(H[7 * a]));
    }
    sh.gmax = gmax;
    sh.x_cost = sh.initial = sh.acc[27];
    return lm_next_step(sh, max_iters);
}

// after a candidate's evaluation: the verdict (trust_region_minimizer.cc:744-829), then the next step
__device__ __noinline__ int lm_after_candidate(PnpShared *shp, int max_iters, double ftol) {
    PnpShared &sh = *shp;
    double *H = sh.H, *g = sh.g;
    const double cand_cost = sh.acc[27], x_cost = sh.x_cost;
    double sn = 0;
#pragma unroll
    for (int i = 0; i < 7; i++) sn += (sh.x[i] - sh.cand[i]) * (sh.x[i] - sh.cand[i]);
    if (sqrt(sn) <= 1e-8 * (sh.x_norm + 1e-8)) return F_DONE;            // ParameterToleranceReached
    if (fabs(x_cost - cand_cost) <= ftol * x_cost) return F_DONE;        // FunctionToleranceReached
    const double rel = (x_cost - cand_cost) / sh.mcc;
    if (rel > 1e-3) {
        double nn = 0;
#pragma unroll
        for (int i = 0; i < 7; i++) {
            const double c = sh.cand[i];
            sh.x[i] = c;
            nn += c * c;
        }
        sh.x_norm = sqrt(nn);
        sh.lm.accepted(rel);
        sh.nsucc++;
#pragma unroll
        for (int a = 0; a < 6; a++)
#pragma unroll
            for (int b = a; b < 6; b++) H[6 * a + b] = H[6 * b + a] = sh.acc[tri(a, b)];
        double gmax = 0;
#pragma unroll
        for (int a = 0; a < 6; a++) {
            g[a] = sh.acc[21 + a];
            gmax = fmax(gmax, fabs(g[a]));
        }
        sh.gmax = gmax;
        sh.x_cost = cand_cost;
    } else {
        sh.lm.rejected();
    }
    sh.nsummaries++;
    return lm_next_step(sh, max_iters);
}

// lane 0, before a solve (a barrier must follow): minimiser state, and R | t of the start pose for the first evaluation
__device__ __forceinline__ void solve_begin(PnpShared &sh) {
    sh.x_cost = 0; sh.gmax = 0; sh.x_norm = -1; sh.initial = 0; sh.mcc = 0;
    sh.lm.reset();
    sh.iteration = 0; sh.nsucc = 1; sh.invalid = 0; sh.nsummaries = 1;
    double T12[12];
    pose_to_T(sh.x, T12);
#pragma unroll
    for (int a = 0; a < 12; a++) sh.T[a] = T12[a];
}

// One ceres::Solve on the pose in sh.x (after solve_begin).  Returns (block-uniformly) 1 = usable, 0 = failure.
// Between two evaluations lane 0 runs ONE block -- the verdict on the candidate just evaluated, then steps until one is valid -- so
// there is one flag exchange per evaluation.
// `bad` = the verdict mask of the LAST evaluation (see eval).
__device__ __forceinline__ int solve(PnpShared &sh, const PnpArgs &A, int robust, const uint8_t *active, double *info, unsigned long long &bad) {
    bad = eval<true>(sh, A, robust, active);
    if (threadIdx.x == 0) sh.flag = lm_first_step(&sh, A.max_iters);
    __syncthreads();
    int flag = sh.flag;   // (lane 0 writes it again only behind the two barriers of the next evaluation)
    while (flag == F_EVAL_CAND) {
        // The candidate is evaluated WITH its Jacobian: when the step is accepted Ceres re-evaluates at the same point
        // (HandleSuccessfulStep, trust_region_minimizer.cc:809-829), which would produce exactly these sums again.
        bad = eval<true>(sh, A, robust, active);
        if (threadIdx.x == 0) sh.flag = lm_after_candidate(&sh, A.max_iters, A.ftol);
        __syncthreads();
        flag = sh.flag;
    }
    if (threadIdx.x == 0 && info) {
        info[0] = sh.nsummaries;
        info[1] = sh.initial;
        info[2] = sh.x_cost;
        info[3] = sh.nsucc;
    }
    __syncthreads();
    return flag == F_DONE;
}

// p3p / inlier0 non-null = chained mode (VisualFrontend::computePose, visual_frontend.cpp:300-375): the initial pose is the
// P3P-LMedS model and only its inliers are refined; the P3P acceptance tests of multi_view_geometry.cpp:82-91 run here.
// `out`, `bad` and `p3p_outlier` may live in pinned host memory (written once, at the end).
// (`chi2` / `depth`: per-point scratch of earlier versions; the verdicts now stay in registers -- see eval -- and the two arrays are unused,
// kept in the signatures so that the callers' scratch layouts did not have to move)
__device__ __forceinline__ void pnp_block(const PnpArgs &A, uint8_t *__restrict__ active, double *__restrict__ chi2,
                                          uint8_t *__restrict__ depth, uint8_t *__restrict__ bad, PnpOut *__restrict__ out,
                                          const P3pSelectOut *__restrict__ p3p, const uint8_t *__restrict__ inlier0,
                                          uint8_t *__restrict__ p3p_outlier) {
    __shared__ PnpShared sh;
    __shared__ int s_nbad, s_p3p_ok, s_nact;
    __shared__ double s_info[8];
    __shared__ double s_model[12];
    __shared__ int s_sel[4];
    if (threadIdx.x == 0) {
        sh.nstamp = 0;
        s_nbad = 0;
        s_nact = 0;
    }
    if (threadIdx.x < 8) s_info[threadIdx.x] = 0;
    __syncthreads();
    PNP_STAMP();
    // One memory phase: the selection record (lanes 12.. of wave 0; lane 0 alone would walk it in dependent round trips) and the pass over
    // the P3P inlier mask.  Chained mode: the robust solve reads that mask as its activity mask directly, and its complement goes out to
    // the host's (pinned) buffer NOW, under the solve, instead of in the kernel's tail behind the final system-scope fence (should the P3P
    // model be refused below, the rare path overwrites it with zeros).
    if (p3p) {
        if (threadIdx.x < 12) s_model[threadIdx.x] = p3p->model[threadIdx.x];
        if (threadIdx.x == 12) s_sel[0] = p3p->have_model;
        if (threadIdx.x == 13) s_sel[1] = p3p->n_inliers;
        if (threadIdx.x == 14) s_sel[2] = p3p->n_valid_used;
    }
    const uint8_t *act1 = inlier0 ? inlier0 : active;
    {
        int na = 0;
        for (int i = threadIdx.x; i < A.n; i += NT) {
            uint8_t a = 1;
            if (inlier0) a = inlier0[i];
            else active[i] = 1;
            if (p3p_outlier) p3p_outlier[i] = !a;
            na += a;
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) na += __shfl_down(na, off);
        if ((threadIdx.x & 63) == 0) atomicAdd(&s_nact, na);
    }
    __syncthreads();   // (also publishes `active` in the unchained mode: every thread wrote only the entries it reads)
    if (threadIdx.x == 0) {
        s_p3p_ok = 1;
        if (p3p) {
            const double *R = s_model;
            double e = 0;
            for (int r = 0; r < 3; r++)
                for (int c = 0; c < 3; c++) {
                    const double v = R[3 * r] * R[3 * c] + R[3 * r + 1] * R[3 * c + 1] + R[3 * r + 2] * R[3 * c + 2] - (r == c ? 1.0 : 0.0);
                    e += v * v;
                }
            // + the translation test of visual_frontend.cpp:323 (isInf / isNaN)
            s_p3p_ok = s_sel[0] && s_sel[1] >= 5 && sqrt(e) < 1e-10 && isfinite(R[9]) && isfinite(R[10]) && isfinite(R[11]);
            if (s_p3p_ok) {
                // rotation matrix -> unit quaternion (Sophus::SE3d::setRotationMatrix -> Eigen::Quaternion(R))
                double q[4];
                const double tr = R[0] + R[4] + R[8];
                if (tr > 0) {
                    double t = sqrt(tr + 1.0);
                    q[3] = 0.5 * t;
                    t = 0.5 / t;
                    q[0] = (R[7] - R[5]) * t; q[1] = (R[2] - R[6]) * t; q[2] = (R[3] - R[1]) * t;
                } else {
                    int i = R[4] > R[0] ? 1 : 0;
                    if (R[8] > R[4 * i]) i = 2;
                    const int j = (i + 1) % 3, k = (i + 2) % 3;
                    double t = sqrt(R[4 * i] - R[4 * j] - R[4 * k] + 1.0);
                    q[i] = 0.5 * t;
                    t = 0.5 / t;
                    q[3] = (R[3 * k + j] - R[3 * j + k]) * t;
                    q[j] = (R[3 * j + i] + R[3 * i + j]) * t;
                    q[k] = (R[3 * k + i] + R[3 * i + k]) * t;
                }
                // ... normalised like PoseParametersBlock(0, Sophus::SE3d(q, t)) does before the solve
                const double nq = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
                for (int c = 0; c < 3; c++) sh.x[c] = R[9 + c];
                for (int c = 0; c < 4; c++) sh.x[3 + c] = q[c] / nq;
            }
        } else {
            const double nq = sqrt(A.pose0[3] * A.pose0[3] + A.pose0[4] * A.pose0[4] + A.pose0[5] * A.pose0[5] + A.pose0[6] * A.pose0[6]);
            for (int c = 0; c < 3; c++) sh.x[c] = A.pose0[c];
            for (int c = 0; c < 4; c++) sh.x[3 + c] = A.pose0[3 + c] / nq;
        }
        if (s_p3p_ok) solve_begin(sh);
    }
    __syncthreads();
    if (!s_p3p_ok) {
        if (threadIdx.x == 0) {
            out->ok = 0;
            out->p3p_ok = 0;
            out->n_bad = 0;
            out->n_active = 0;
            out->p3p_n_valid_used = s_sel[2];
        }
        if (threadIdx.x < 8) out->info[threadIdx.x] = 0;
        for (int i = threadIdx.x; i < A.n; i += NT) {
            bad[i] = 0;
            if (p3p_outlier) p3p_outlier[i] = 0;
        }
        return;
    }
    if (p3p && threadIdx.x < 7) out->pose_p3p[threadIdx.x] = sh.x[threadIdx.x];
    unsigned long long badmask;
    int ok = solve(sh, A, A.use_robust, act1, s_info, badmask);
    const int nact = s_nact;
    {
        // multi_view_geometry.cpp:194-207 from the verdicts of the last evaluation, which this thread still holds for its own points
        int nb = 0, k = 0;
        for (int i = threadIdx.x; i < A.n; i += NT, k++) {
            const bool b = (badmask >> k) & 1ull;
            bad[i] = b;
            if (A.apply_l2) active[i] = act1[i] && !b;
            nb += b;
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) nb += __shfl_down(nb, off);
        if ((threadIdx.x & 63) == 0 && nb) atomicAdd(&s_nbad, nb);
    }
    __syncthreads();
    const int nbad = s_nbad;
    if (nbad == nact) ok = 0;
    else if (A.apply_l2 && nbad > 0) {   // :214-218
        if (threadIdx.x == 0) solve_begin(sh);
        __syncthreads();
        ok = solve(sh, A, 0, active, s_info + 4, badmask);
    }
    if (threadIdx.x < 7) out->pose[threadIdx.x] = sh.x[threadIdx.x];
    if (threadIdx.x < 8) out->info[threadIdx.x] = s_info[threadIdx.x];
    PNP_STAMP();
    if (A.dbg && threadIdx.x == 0) A.dbg[2047] = (unsigned long long) sh.nstamp;
    if (threadIdx.x == 0) {
        out->ok = ok;
        out->n_bad = nbad;
        out->p3p_ok = 1;
        out->n_active = nact;
        out->p3p_n_valid_used = p3p ? s_sel[2] : 0;
    }
}

__device__ __forceinline__ void pnp_body(const PnpArgs &A, uint8_t *__restrict__ active, double *__restrict__ chi2,
                                         uint8_t *__restrict__ depth, uint8_t *__restrict__ bad, PnpOut *__restrict__ out,
                                         const P3pSelectOut *__restrict__ p3p, const uint8_t *__restrict__ inlier0,
                                         uint8_t *__restrict__ p3p_outlier) {
    pnp_block(A, active, chi2, depth, bad, out, p3p, inlier0, p3p_outlier);
    // (Measured alternatives to this fence, 2.5 us of kernel tail: plain stores + s_waitcnt vmcnt(0) -> the host reads stale masks (host
    // memory is cached in L2); every host-visible word as a system-scope write-through store + vmcnt(0) -> correct, the fence's time
    // reappears in front of every later vmcnt wait of the storing waves (the counter is in-order and a write-through store is acknowledged
    // by the bus), +4 us over the kernel.)
    __threadfence_system();   // this thread's writes to `out` / `bad` / `p3p_outlier` (possibly pinned host memory) ...
    __syncthreads();          // ... of every thread ...
    if (threadIdx.x == 0) __hip_atomic_store(&out->seq, A.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);   // ... before the word a host may poll
}
__global__ void __launch_bounds__(NT) k_pnp(PnpArgs A, uint8_t *__restrict__ active, double *__restrict__ chi2,
                                            uint8_t *__restrict__ depth, uint8_t *__restrict__ bad, PnpOut *__restrict__ out,
                                            const P3pSelectOut *__restrict__ p3p, const uint8_t *__restrict__ inlier0,
                                            uint8_t *__restrict__ p3p_outlier) {
    pnp_body(A, active, chi2, depth, bad, out, p3p, inlier0, p3p_outlier);
}

// B chained P3P -> PnP problems in one launch, one workgroup each (blockIdx.x = camera).
struct PnpBatchItem {
    PnpArgs A;
    uint8_t *active;
    double *chi2;
    uint8_t *depth, *bad;
    PnpOut *out;
    const P3pSelectOut *p3p;
    const uint8_t *inlier0;
    uint8_t *p3p_outlier;
};
// several sessions' refinements in one launch (lane.hpp): one workgroup each, with k_pnp's completion word
ALVA_MULTI_KERNEL(MK_PNP, k_pnp_multi, PnpBatchItem, dim3(NT), NT, pnp_body(A.A, A.active, A.chi2, A.depth, A.bad, A.out, A.p3p, A.inlier0, A.p3p_outlier));

// ---- the single session's pose solve as ONE launch (round 6) ---------------------------------------------------------------------
// VisualFrontend::computePose chains P3P-LMedS and the refinement (visual_frontend.cpp:245-417).  The P3P launch already ends in ONE
// workgroup -- the last to arrive selects the winner and classifies its inliers -- and the refinement is one workgroup of the same 512
// threads: that workgroup simply goes on.  Its selection record and inlier mask are its own stores (workgroup scope: a barrier orders
// them), so the kernel boundary between k_p3p_s and k_pnp -- an agent-scope release, a dispatch, ~4 us of the frame's dependent chain --
// disappears; every other workgroup has left long before.  Same arithmetic as the two launches (p3p_block, pnp_body).
static_assert(NT == 512, "the fused launch runs P3P's workgroup shape");
__global__ void __launch_bounds__(NT) k_p3p_pnp_s(P3pArgs P, P3pInlineSamples S, PnpBatchItem it) {
    if (!p3p_block<0, NT>(P, blockIdx.x, S.v + 4 * blockIdx.x)) return;
    __syncthreads();
    pnp_body(it.A, it.active, it.chi2, it.depth, it.bad, it.out, it.p3p, it.inlier0, it.p3p_outlier);
}

// ---- ... and the whole tail of the tracking frame as ONE launch: compaction -> P3P-LMedS -> refinement ---------------------------------
// The tracking frame's chain was tracker -> compaction -> P3P -> PnP, and the compaction (13 us + a kernel boundary) sat on it for two
// reasons: the pose solve reads the correspondences it gathers, and the host needs the tracker's counts before it can draw P3P's samples
// -- the host did that WHILE the compaction ran.  Here the launch is queued right behind the tracker, before the host knows anything:
//   phase A   the first G workgroups ARE the compaction (track_compact_body: per-slot results to the host, correspondences gathered,
//             completion word published by the last one) -- meanwhile the host sees the tracker's early word, draws the samples for the
//             now known n into the pinned PoseGo block and publishes its go word (or "abort": p3pReq_, too few correspondences);
//   wait      every workgroup: all slices gathered (a device word the last compaction workgroup writes) and the host's word -- a few
//             polls, bounded; an aborted launch ends here;
//   phase B   P3P-LMedS, one workgroup per hypothesis, n / H / its four sample indices read from the PoseGo block;
//   phase C   the selecting workgroup goes on with the refinement (as k_p3p_pnp_s).
// No workgroup ever waits for a workgroup with a higher index (phase A's are the first ones dispatched), so the wait cannot deadlock on
// residency; the host always writes one of the two words (HipStages::track_begin), and the poll gives up after ~1 s anyway.
struct PoseGo {
    long long word;   // seq = go, -seq = abort (the launch's sequence number: strictly increasing, so a stale word never matches)
    int n, H;
    int samples[4 * P3P_INLINE_H];
    long long nack;   // written by the KERNEL: seq = "gave up waiting for the word" (a tool that makes launches synchronous -- counter
                      // collection does -- keeps the host from answering while the kernel runs); the host then solves the pose the old way
};
__device__ __forceinline__ int sys_load(const int *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
#define POSE_ALL_STAMP(k)                                                                                                                  \
    do {                                                                                                                                   \
        if (it.A.dbg && threadIdx.x == 0 && (blockIdx.x == 0 || (int) blockIdx.x == G || blockIdx.x == gridDim.x - 1))                  \
            it.A.dbg[4048 + 8 * (blockIdx.x == 0 ? 0 : ((int) blockIdx.x == G ? 1 : 2)) + (k)] = wall_clock64();                        \
    } while (0)
// relay: 8-byte words in DEVICE memory (ctx->d_counters + 64): [0] the word, [1] n | H << 32, [2 + h] hypothesis h's samples 0,1 | [2 +
// P3P_INLINE_H + h] its samples 2,3.  Workgroup 0 alone talks to the host: 128 workgroups polling pinned host memory over the bus is
// 128 reads in flight against the compaction's own traffic to the host; the others poll the relay's word in device memory.
__global__ void __launch_bounds__(NT) k_pose_all(TrackSlots D, int G, P3pArgs P, PnpBatchItem it, const PoseGo *go, unsigned long long *relay,
                                                 int seq) {
    __shared__ int s_go[8];   // verdict, n, H, - | this workgroup's four sample indices
    POSE_ALL_STAMP(0);
    // workgroups [0, G): the compaction, nothing else (their slices' host copies and system-scope fences run beside the pose solve);
    // workgroups [G, G + H): the hypotheses; workgroup G also relays the host's word
    if ((int) blockIdx.x < G) {
        (void) track_compact_body<NT, true>(D, (int) blockIdx.x, G);
        return;
    }
    const int hb = (int) blockIdx.x - G;   // this workgroup's hypothesis
    POSE_ALL_STAMP(1);
    if (hb == 0) {
        if (threadIdx.x == 0) {
            unsigned spins = 0;
            long long w = 0;
            for (;;) {
                w = __hip_atomic_load(&go->word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                if (w == (long long) seq || w == -(long long) seq || ++spins > 400u) break;   // ~0.6 ms of polls over the bus: the answer takes ~10 us
                __builtin_amdgcn_s_sleep(4);
            }
            s_go[0] = w == (long long) seq;
            if (w != (long long) seq && w != -(long long) seq)
                __hip_atomic_store(const_cast<long long *>(&go->nack), (long long) seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        __syncthreads();
        if (s_go[0]) {
            const int t = threadIdx.x;
            if (t == 0) agent_store(relay + 1, (unsigned long long) (unsigned) sys_load(&go->n) | ((unsigned long long) (unsigned) sys_load(&go->H) << 32));
            else if (t <= 2 * P3P_INLINE_H) {
                const int h = (t - 1) >> 1, half = (t - 1) & 1;
                agent_store(relay + 2 + half * P3P_INLINE_H + h, (unsigned long long) (unsigned) sys_load(go->samples + 4 * h + 2 * half) |
                                                                     ((unsigned long long) (unsigned) sys_load(go->samples + 4 * h + 2 * half + 1) << 32));
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __syncthreads();
        if (threadIdx.x == 0) agent_store(relay, (unsigned long long) (long long) (s_go[0] ? seq : -seq));
    }
    if (threadIdx.x == 0) {
        unsigned spins = 0;
        bool ok = true;
        while (__hip_atomic_load(D.cnt + 10, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != seq && ok) {
            __builtin_amdgcn_s_sleep(8);
            ok = ++spins < (1u << 16);
        }
        POSE_ALL_STAMP(2);
        long long w = 0;
        while (ok) {
            w = (long long) agent_load(relay);
            if (w == (long long) seq || w == -(long long) seq) break;
            __builtin_amdgcn_s_sleep(8);
            ok = ++spins < (1u << 16);   // (~15 ms: far beyond the relaying workgroup's own bound)
        }
        s_go[0] = ok && w == (long long) seq;
    }
    __syncthreads();
    POSE_ALL_STAMP(3);
    if (!s_go[0]) return;
    if (threadIdx.x < 3) {
        const int k = threadIdx.x, h = min(hb, P3P_INLINE_H - 1);
        const unsigned long long v = agent_load(relay + (k == 0 ? 1 : 2 + (k - 1) * P3P_INLINE_H + h));
        s_go[k == 0 ? 1 : 2 + 2 * k] = (int) (unsigned) v;
        s_go[k == 0 ? 2 : 3 + 2 * k] = (int) (unsigned) (v >> 32);
    }
    __syncthreads();
    if (hb >= s_go[2]) return;
    // No agent-scope acquire here, on purpose: an L2 invalidate per workgroup, staggered over the scoring phase, has every XCD refetch the
    // correspondences again and again (measured: the launch took 147 us instead of ~70).  It is not needed either: the kernel boundary in
    // front of this launch left no line of the gathered arrays in any L2, nothing reads them before this point, and their writers wrote
    // them back (system-scope fence) before they arrived -- so the first read after the wait misses and fetches what they wrote.
    P.n = s_go[1];
    P.H = s_go[2];
    it.A.n = s_go[1];
    if (!p3p_block<0, NT>(P, hb, s_go + 4)) return;
    __syncthreads();
    pnp_body(it.A, it.active, it.chi2, it.depth, it.bad, it.out, it.p3p, it.inlier0, it.p3p_outlier);
}

__global__ void __launch_bounds__(NT) k_pnp_batch(const PnpBatchItem *__restrict__ items) {
    const PnpBatchItem &it = items[blockIdx.x];
    pnp_block(it.A, it.active, it.chi2, it.depth, it.bad, it.out, it.p3p, it.inlier0, it.p3p_outlier);
}

}  // namespace

extern "C" int alva_pnp_refine(alva_ctx *ctx, const double *d_uv, const double *d_wpts, int n, double *h_pose7, int max_iters,
                               float chi2_th, int use_robust, int apply_l2_after_robust, float fx, float fy, float cx, float cy,
                               int *h_outliers, int *h_n_outliers, double *h_info, int *h_ok) {
    ALVA_ARG(ctx && h_pose7 && h_outliers && h_n_outliers && h_ok && n >= 0 && n <= 64 * NT && max_iters >= 0);   // (64 verdict bits per thread)
    *h_ok = 0;
    *h_n_outliers = 0;
    if (h_info) memset(h_info, 0, 8 * sizeof(double));
    if (n == 0) return ALVA_OK;  // nbad == numKeyPoints -> false (:209-212)
    ALVA_ARG(d_uv && d_wpts);
    PnpArgs A{};
    A.uv = d_uv;
    A.wpt = d_wpts;
    A.n = n;
    A.K[0] = fx; A.K[1] = fy; A.K[2] = cx; A.K[3] = cy;
    A.huber_a = (double) sqrtf(chi2_th);  // :135 std::sqrt(float)
    A.chi2_th = (double) chi2_th;
    A.use_robust = use_robust;
    A.apply_l2 = apply_l2_after_robust;
    A.max_iters = max_iters;
    A.ftol = 1.e-3;  // :186
    memcpy(A.pose0, h_pose7, sizeof(A.pose0));
    A.dbg = alva_kstamp_buffer();
    // device scratch: chi2(n) | active(n) | depth(n);  pinned (written by the kernel, read after the sync): out | bad(n)
    const size_t off_act = (size_t) n * 8, off_dep = off_act + (size_t) n;
    uint8_t *base = nullptr, *pin = nullptr;
    int rc = alva_ctx_scratch(ctx, 3, off_dep + (size_t) n, (void **) &base);
    if (rc) return rc;
    rc = alva_ctx_pinned(ctx, 256 + (size_t) n, (void **) &pin);
    if (rc) return rc;
    static_assert(sizeof(PnpOut) <= 256, "pinned layout");
    hipLaunchKernelGGL(k_pnp, dim3(1), dim3(NT), 0, ctx->stream, A, base + off_act, (double *) base, base + off_dep, pin + 256,
                       (PnpOut *) pin, (const P3pSelectOut *) nullptr, (const uint8_t *) nullptr, (uint8_t *) nullptr);
    ALVA_LAUNCH_CHECK();
    ALVA_HIP(alva_stream_sync(ctx->stream));
    PnpOut res;
    memcpy(&res, pin, sizeof(res));
    const uint8_t *bad = pin + 256;
    int no = 0;
    for (int i = 0; i < n; i++)
        if (bad[i]) h_outliers[no++] = i;
    *h_n_outliers = no;
    if (h_info) memcpy(h_info, res.info, sizeof(res.info));
    if (no == n) return ALVA_OK;
    memcpy(h_pose7, res.pose, sizeof(res.pose));
    *h_ok = res.ok;
    return ALVA_OK;
}

// VisualFrontend::computePose (src/slam/src/visual_frontend.cpp:245-417) as ONE device-side chain and ONE host
// synchronisation: P3P-LMedS -> acceptance tests -> drop its outliers -> robust PnP on the inliers -> acceptance tests.
// Split in enqueue / collect so that a caller can put other work (the detector, on a second stream) between the two.
struct alva_pose_pending {
    const double *bearings, *wpts;
    PnpArgs A;
    int n, p3p_iters, do_random, H, max_draws;
    float p3p_err, fx, fy;
    uint32_t seed;
    size_t poff_out, poff_bad, poff_po;
    uint8_t *pin;
    bool active;
    int seq = 0;
    int go_seq = 0;   // > 0: a k_pose_all is queued and waits for alva_pose_all_go / _abort (the tracker launch's sequence number)
    int went_seq = 0; // > 0: the sequence number alva_pose_all_go answered (until the result is collected): PoseGo::nack is compared with it
    // the fused launch's sampler, split: the generator's outputs do not depend on n (uniform_int_distribution<>(0, INT_MAX) over
    // std::mt19937: SampleConsensusProblem.hpp:40-46), only "% (n - i)" and the swaps do -- so the 4 H raw draws are made when the
    // launch is queued (the tracker is still running) and alva_pose_all_go is left with 4 H swaps on a persistent iota array
    std::vector<int> raw, iota;
};

static void pose_pending_free(void *p) { delete (alva_pose_pending *) p; }

static int pose_launch(alva_ctx *ctx, alva_pose_pending &P) {
    const int n = P.n;
    const size_t off_act = (size_t) n * 8, off_dep = off_act + (size_t) n, off_sel = (off_dep + (size_t) n + 63) / 64 * 64;
    const size_t off_inl = off_sel + 256;
    uint8_t *base = nullptr;
    int rc = alva_ctx_scratch(ctx, 3, off_inl + (size_t) n, (void **) &base);
    if (rc) return rc;
    // pinned: samples | PnpOut | bad(n) | p3p outlier(n): read / written by the kernels directly, no copy commands
    P.poff_out = ((size_t) P.H * 16 + 255) / 256 * 256;
    P.poff_bad = P.poff_out + 256;
    P.poff_po = P.poff_bad + (size_t) n;
    rc = alva_ctx_pinned(ctx, P.poff_po + (size_t) n, (void **) &P.pin);
    if (rc) return rc;
    P3pSelectOut *d_sel = (P3pSelectOut *) (base + off_sel);
    uint8_t *d_inl = base + off_inl;
    P3pArgs PA{};
    rc = alva_p3p_prepare(ctx, P.bearings, P.wpts, n, P.p3p_iters, P.p3p_err, P.do_random, P.seed, P.fx, P.fy, P.H, (int *) P.pin, d_sel, d_inl, &PA);
    if (rc) return rc;
    P.A.seq = ++P.seq;
    ((PnpOut *) (P.pin + P.poff_out))->seq = 0;   // the staging may be fresh memory; every earlier user of it has completed (polled or synchronised)
    const PnpBatchItem item{P.A, base + off_act, (double *) base, base + off_dep, P.pin + P.poff_bad, (PnpOut *) (P.pin + P.poff_out),
                            (const P3pSelectOut *) d_sel, (const uint8_t *) d_inl, P.pin + P.poff_po};
    // ONE launch for both stages: a session of its own (no lane), samples that fit the kernel arguments, keys that fit the default
    // dynamic-LDS limit (ALVA_POSE_UNFUSED=1: the two launches, for A/B)
    const bool fuse = getenv("ALVA_POSE_UNFUSED") == nullptr;
    if (fuse && !g_alva_lane && P.H <= P3P_INLINE_H && alva_p3p_inline_samples_ok() && n <= 7168) {
        P3pInlineSamples S;
        memcpy(S.v, PA.samples, (size_t) P.H * 16);
        ctx->p3p_deferred = false;
        hipLaunchKernelGGL(k_p3p_pnp_s, dim3((unsigned) P.H), dim3(NT), (size_t) n * sizeof(double), ctx->stream, PA, S, item);
        ALVA_LAUNCH_CHECK();
        return ALVA_OK;
    }
    rc = alva_p3p_launch(ctx, PA);
    if (rc) return rc;
    // deposited only behind a DEPOSITED P3P (same lane stream, chain order); a P3P that was launched on this context's own stream
    // (n > 7168) is followed on that stream: the lane's stream and the session's are not ordered against each other
    if (ctx->p3p_deferred && alva_lane_defer(MK_PNP, ctx, 1, 0, &item, sizeof(item))) return ALVA_OK;
    hipLaunchKernelGGL(k_pnp, dim3(1), dim3(NT), 0, ctx->stream, P.A, base + off_act, (double *) base, base + off_dep, P.pin + P.poff_bad,
                       (PnpOut *) (P.pin + P.poff_out), (const P3pSelectOut *) d_sel, (const uint8_t *) d_inl, P.pin + P.poff_po);
    ALVA_LAUNCH_CHECK();
    return ALVA_OK;
}

// ---- the fused tail of the tracking frame (k_pose_all): enqueue right behind the tracker | go, once the host knows n | abort ----------
static void pose_params(alva_pose_pending &P, const double *d_bearings, const double *d_uv, const double *d_wpts, int n, int p3p_iters, float p3p_err,
                        int do_random, uint32_t seed, int pnp_iters, float chi2_th, float fx, float fy, float cx, float cy) {
    P.bearings = d_bearings;
    P.wpts = d_wpts;
    P.p3p_iters = p3p_iters;
    P.p3p_err = p3p_err;
    P.do_random = do_random;
    P.seed = seed;
    P.fx = fx;
    P.fy = fy;
    P.max_draws = p3p_iters + p3p_iters * 10;
    P.H = std::min(P.max_draws, p3p_iters + 28);
    PnpArgs &A = P.A;
    A = PnpArgs{};
    A.uv = d_uv;
    A.wpt = d_wpts;
    A.n = n;
    A.K[0] = fx; A.K[1] = fy; A.K[2] = cx; A.K[3] = cy;
    A.huber_a = (double) sqrtf(chi2_th);
    A.chi2_th = (double) chi2_th;
    A.use_robust = 1;   // visual_frontend.cpp:361
    A.apply_l2 = 1;     // state.hpp:76 robustCostRefineWithL2_
    A.max_iters = pnp_iters;
    A.ftol = 1.e-3;
    A.dbg = alva_kstamp_buffer();
}

// how often the fused launch was queued / answered with go / gave up waiting and was replaced by the separate launches (process-wide;
// tools/soak.py reports them: a fallback costs ~0.7 ms once, a rate above ~1e-4 would mean the host's answer is not as prompt as assumed)
static std::atomic<long> g_pose_all_queued{0}, g_pose_all_go{0}, g_pose_all_fallback{0};
extern "C" void alva_debug_pose_all_stats(long *out3) {
    out3[0] = g_pose_all_queued.load();
    out3[1] = g_pose_all_go.load();
    out3[2] = g_pose_all_fallback.load();
}

bool alva_pose_all_possible(int n_cap, int p3p_iters) {
    const bool on = getenv("ALVA_POSE_UNFUSED") == nullptr && getenv("ALVA_NO_POSE_ALL") == nullptr;   // (per call: tests flip them in one process)
    // Not inside a session group: there the calling thread runs OTHER sessions' frames while this one's kernels fly (its polls yield to
    // the fiber scheduler, alva_fiber_yield), so the answer the queued launch waits for could be milliseconds away -- with ~170 workgroups
    // spinning meanwhile.  A session that owns its thread answers within ~10 us.
    return on && !g_alva_lane && !alva_fiber_yield && n_cap >= 4 && n_cap <= 7168 && p3p_iters + 28 <= P3P_INLINE_H && alva_p3p_inline_samples_ok();
}

int alva_pose_all_enqueue(alva_ctx *ctx, const TrackSlots &D, int G, int p3p_iters, float p3p_err, int do_random, uint32_t seed, int pnp_iters,
                          float chi2_th, float fx, float fy, float cx, float cy) {
    ALVA_ARG(ctx && alva_pose_all_possible(D.n, p3p_iters) && G >= 1);
    if (!ctx->pose_pending) {
        ctx->pose_pending = new alva_pose_pending();
        ctx->pose_pending_free = pose_pending_free;
    }
    alva_pose_pending &P = *(alva_pose_pending *) ctx->pose_pending;
    const int n_cap = D.n;
    pose_params(P, D.Pbv, D.Puv, D.Pwpt, n_cap, p3p_iters, p3p_err, do_random, seed, pnp_iters, chi2_th, fx, fy, cx, cy);
    P.active = false;   // until alva_pose_all_go
    P.n = 0;
    const size_t off_act = (size_t) n_cap * 8, off_dep = off_act + (size_t) n_cap, off_sel = (off_dep + (size_t) n_cap + 63) / 64 * 64;
    const size_t off_inl = off_sel + 256;
    uint8_t *base = nullptr;
    int rc = alva_ctx_scratch(ctx, 3, off_inl + (size_t) n_cap, (void **) &base);
    if (rc) return rc;
    // pinned: PoseGo (samples inside) | PnpOut | bad(n) | p3p outlier(n)
    P.poff_out = (sizeof(PoseGo) + 255) / 256 * 256;
    P.poff_bad = P.poff_out + 256;
    P.poff_po = P.poff_bad + (size_t) n_cap;
    rc = alva_ctx_pinned(ctx, P.poff_po + (size_t) n_cap, (void **) &P.pin);
    if (rc) return rc;
    P3pSelectOut *d_sel = (P3pSelectOut *) (base + off_sel);
    uint8_t *d_inl = base + off_inl;
    P3pArgs PA{};
    rc = alva_p3p_prepare(ctx, P.bearings, P.wpts, n_cap, P.p3p_iters, P.p3p_err, P.do_random, P.seed, P.fx, P.fy, P.H, nullptr, d_sel, d_inl, &PA);
    if (rc) return rc;
    P.A.seq = ++P.seq;
    ((PnpOut *) (P.pin + P.poff_out))->seq = 0;
    P.go_seq = D.seq;
    {
        P.raw.resize((size_t) P.H * 4);
        rc = alva_p3p_raw_draws(P.H * 4, P.do_random, P.seed, P.raw.data());
        if (rc) return rc;
        const size_t have = P.iota.size();
        if ((size_t) n_cap > have) {
            P.iota.resize((size_t) n_cap);
            for (size_t i = have; i < (size_t) n_cap; i++) P.iota[i] = (int) i;
        }
    }
    const PnpBatchItem item{P.A, base + off_act, (double *) base, base + off_dep, P.pin + P.poff_bad, (PnpOut *) (P.pin + P.poff_out),
                            (const P3pSelectOut *) d_sel, (const uint8_t *) d_inl, P.pin + P.poff_po};
    ctx->p3p_deferred = false;
    g_pose_all_queued++;
    hipLaunchKernelGGL(k_pose_all, dim3((unsigned) (P.H + G)), dim3(NT), (size_t) n_cap * sizeof(double), ctx->stream, D, G, PA, item,
                       (const PoseGo *) P.pin, reinterpret_cast<unsigned long long *>(ctx->d_counters + 64), D.seq);
    ALVA_LAUNCH_CHECK();
    return ALVA_OK;
}

// the host's word to a queued k_pose_all: n correspondences (4 <= n <= the launch's slot count) -> samples drawn, go
int alva_pose_all_go(alva_ctx *ctx, int n) {
    alva_pose_pending *pp = (alva_pose_pending *) ctx->pose_pending;
    ALVA_ARG(pp && pp->go_seq > 0 && n >= 4);
    alva_pose_pending &P = *pp;
    PoseGo *go = (PoseGo *) P.pin;
    if ((size_t) n > P.iota.size() || P.raw.size() < (size_t) P.H * 4) {
        (void) alva_pose_all_abort(ctx);
        alva_set_error("alva_pose_all_go: %d correspondences, the launch was sized for %zu", n, P.iota.size());
        return ALVA_ERR_ARG;
    }
    {
        // Sampler::draw of p3p.hip (SampleConsensusProblem.hpp:65-84: a prefix Fisher-Yates on an index array that persists across the
        // draws of one call) on the raw draws made at enqueue time; the touched entries are put back afterwards: the array stays iota
        int *sh = P.iota.data();
        const size_t index_size = (size_t) n;
        static thread_local std::vector<int> touched;
        touched.clear();
        // (raw % (n - i): four divisors for the whole call -- the remainders by multiplication (Lemire's fastmod: exact for 32-bit operands)
        // instead of 4 H hardware divisions, 2 of the answer's 2.5 us on the frame's critical path)
        uint64_t magic[4];
        for (unsigned i = 0; i < 4; ++i) magic[i] = UINT64_C(0xFFFFFFFFFFFFFFFF) / (uint64_t) (index_size - i) + 1;
        for (int k = 0; k < P.H; k++) {
            for (unsigned i = 0; i < 4; ++i) {
                const uint64_t low = magic[i] * (uint64_t) (uint32_t) P.raw[(size_t) 4 * k + i];
                const size_t j = i + (size_t) (((__uint128_t) low * (uint64_t) (index_size - i)) >> 64);
                std::swap(sh[i], sh[j]);
                touched.push_back((int) j);
            }
            for (int i = 0; i < 4; i++) go->samples[4 * k + i] = sh[i];
        }
        for (int i = 0; i < 4; i++) sh[i] = i;
        for (int j: touched) sh[j] = j;
    }
    int rc = ALVA_OK;
    go->n = n;
    go->H = P.H;
    P.n = n;
    P.A.n = n;
    P.active = true;
    P.went_seq = P.go_seq;
    g_pose_all_go++;
    {   // test hook (tests/test_gpu_system.py): answer too late on purpose, so that the launch gives up and the fallback is exercised
        const char *late = getenv("ALVA_POSE_ALL_LATE_US");   // (read per call: a test flips it inside one process)
        const int late_us = late ? atoi(late) : 0;
        if (late_us > 0) {
            const auto t0 = std::chrono::steady_clock::now();
            while (std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - t0).count() < late_us) {}
        }
    }
    __atomic_store_n(&go->word, (long long) P.go_seq, __ATOMIC_RELEASE);
    P.go_seq = 0;
    return ALVA_OK;
}

int alva_pose_all_abort(alva_ctx *ctx) {
    alva_pose_pending *pp = (alva_pose_pending *) ctx->pose_pending;
    if (!pp || pp->go_seq <= 0) return ALVA_OK;
    __atomic_store_n(&((PoseGo *) pp->pin)->word, -(long long) pp->go_seq, __ATOMIC_RELEASE);
    pp->go_seq = 0;
    pp->active = false;
    return ALVA_OK;
}

extern "C" int alva_compute_pose_enqueue(alva_ctx *ctx, const double *d_bearings, const double *d_uv, const double *d_wpts, int n,
                                         int p3p_iters, float p3p_err, int do_random, uint32_t seed, int pnp_iters, float chi2_th, float fx,
                                         float fy, float cx, float cy) {
    ALVA_ARG(ctx && n >= 0 && n <= 64 * NT && p3p_iters > 0 && pnp_iters >= 0);
    if (!ctx->pose_pending) {
        ctx->pose_pending = new alva_pose_pending();
        ctx->pose_pending_free = pose_pending_free;
    }
    alva_pose_pending &P = *(alva_pose_pending *) ctx->pose_pending;
    (void) alva_pose_all_abort(ctx);   // (a queued fused launch nobody answered: cannot happen through HipStages, cheap to be sure)
    P.active = true;
    P.n = n;
    if (n < 4) return ALVA_OK;  // visual_frontend.cpp:249-257
    ALVA_ARG(d_bearings && d_uv && d_wpts);
    pose_params(P, d_bearings, d_uv, d_wpts, n, p3p_iters, p3p_err, do_random, seed, pnp_iters, chi2_th, fx, fy, cx, cy);
    return pose_launch(ctx, P);
}

extern "C" int alva_compute_pose_collect(alva_ctx *ctx, double *h_pose7, uint8_t *h_p3p_outlier, uint8_t *h_pnp_outlier, int *h_status) {
    return alva_compute_pose_collect_p3p(ctx, h_pose7, nullptr, h_p3p_outlier, h_pnp_outlier, h_status);
}

// the same, also returning the accepted P3P pose (status >= 1): the frame keeps it when the refinement is rejected (visual_frontend.cpp:335)
int alva_compute_pose_collect_p3p(alva_ctx *ctx, double *h_pose7, double *h_pose7_p3p, uint8_t *h_p3p_outlier, uint8_t *h_pnp_outlier,
                                  int *h_status) {
    ALVA_ARG(ctx && h_pose7 && h_status);
    alva_pose_pending *pp = (alva_pose_pending *) ctx->pose_pending;
    if (!pp || !pp->active) {
        alva_set_error("alva_compute_pose_collect: nothing enqueued on this context");
        return ALVA_ERR_STATE;
    }
    alva_pose_pending &P = *pp;
    P.active = false;
    *h_status = 0;
    const int n = P.n;
    if (n < 4) return ALVA_OK;
    PnpOut res{};
    static const bool poll = getenv("ALVA_NO_POLL") == nullptr;
    for (;;) {
        if (poll) {
            // k_pnp publishes its sequence number after all results; spinning on that word in pinned memory returns a few microseconds
            // before hipStreamSynchronize would (the stream itself is waited for by whoever synchronises next)
            const volatile int *flag = &((const PnpOut *) (P.pin + P.poff_out))->seq;
            unsigned spins = 0;
            while (*flag != P.seq) {
                if (P.went_seq > 0 && *reinterpret_cast<const volatile long long *>(&((const PoseGo *) P.pin)->nack) == (long long) P.went_seq) {
                    // the fused launch gave up before the go word reached it (see PoseGo::nack): its compaction phase is done, the
                    // correspondences are gathered -- solve the pose with the separate launches
                    P.went_seq = 0;
                    g_pose_all_fallback++;
                    const int rc = pose_launch(ctx, P);
                    if (rc) return rc;
                    flag = &((const PnpOut *) (P.pin + P.poff_out))->seq;
                    spins = 0;
                    continue;
                }
                if (++spins > (1u << 26)) {
                    ALVA_HIP(alva_stream_sync(ctx->stream));
                    break;
                }
                alva_poll_relax(spins);
            }
            P.went_seq = 0;
            __atomic_thread_fence(__ATOMIC_ACQUIRE);
        } else {
            ALVA_HIP(alva_stream_sync(ctx->stream));
        }
        memcpy(&res, P.pin + P.poff_out, sizeof(res));
        if (res.p3p_n_valid_used >= P.p3p_iters || P.H >= P.max_draws) break;
        P.H = std::min(P.max_draws, P.H * 2);   // rare: too many degenerate samples, redo with a longer prefix of the stream
        int rc = pose_launch(ctx, P);
        if (rc) return rc;
    }
    const uint8_t *bad = P.pin + P.poff_bad, *p3p_out = P.pin + P.poff_po;
    for (int i = 0; i < n; i++) {
        if (h_p3p_outlier) h_p3p_outlier[i] = p3p_out[i];
        if (h_pnp_outlier) h_pnp_outlier[i] = bad[i];
    }
    if (!res.p3p_ok) return ALVA_OK;                                   // :318-330 -> resetFrame, false
    *h_status = 1;                                                      // P3P pose accepted
    if (h_pose7_p3p) memcpy(h_pose7_p3p, res.pose_p3p, sizeof(res.pose_p3p));
    if (res.n_bad == res.n_active) return ALVA_OK;                      // ceresPnP returns false before writing the pose
    memcpy(h_pose7, res.pose, sizeof(res.pose));
    const int inliers = res.n_active - res.n_bad;
    bool finite = true;
    for (int c = 0; c < 3; c++) finite = finite && std::isfinite(res.pose[c]);
    if (res.ok && inliers >= 5 && res.n_bad <= 0.5 * res.n_active && finite) *h_status = 2;   // :383-399
    return ALVA_OK;
}

extern "C" int alva_compute_pose(alva_ctx *ctx, const double *d_bearings, const double *d_uv, const double *d_wpts, int n, int p3p_iters,
                                 float p3p_err, int do_random, uint32_t seed, int pnp_iters, float chi2_th, float fx, float fy, float cx,
                                 float cy, double *h_pose7, uint8_t *h_p3p_outlier, uint8_t *h_pnp_outlier, int *h_status) {
    ALVA_ARG(ctx && h_pose7 && h_status);
    int rc = alva_compute_pose_enqueue(ctx, d_bearings, d_uv, d_wpts, n, p3p_iters, p3p_err, do_random, seed, pnp_iters, chi2_th, fx, fy,
                                       cx, cy);
    if (rc) return rc;
    return alva_compute_pose_collect(ctx, h_pose7, h_p3p_outlier, h_pnp_outlier, h_status);
}

// ---- batch of cameras (internal: track_batch.hip) -------------------------------------------------------------------------
size_t alva_pnp_batch_item_size() { return sizeof(PnpBatchItem); }
size_t alva_pnp_out_size() { return sizeof(PnpOut); }

// device scratch one problem of n correspondences needs: chi2(n) | active(n) | depth(n) | bad(n) | p3p outlier(n)
size_t alva_pnp_batch_scratch_bytes(int n) { return ((size_t) n * 12 + 63) / 64 * 64; }

// computePose's PnP stage chained behind the P3P selection `d_sel` / inlier mask `d_inlier0` of the same camera
int alva_pnp_batch_item_fill(void *dst, const double *d_uv, const double *d_wpts, int n, int pnp_iters, float chi2_th, float fx, float fy,
                             float cx, float cy, uint8_t *d_scratch, void *out, const P3pSelectOut *d_sel, const uint8_t *d_inlier0) {
    ALVA_ARG(dst && d_uv && d_wpts && n >= 4 && d_scratch && out && d_sel && d_inlier0);
    PnpBatchItem it{};
    PnpArgs &A = it.A;
    A.uv = d_uv;
    A.wpt = d_wpts;
    A.n = n;
    A.K[0] = fx; A.K[1] = fy; A.K[2] = cx; A.K[3] = cy;
    A.huber_a = (double) sqrtf(chi2_th);
    A.chi2_th = (double) chi2_th;
    A.use_robust = 1;
    A.apply_l2 = 1;
    A.max_iters = pnp_iters;
    A.ftol = 1.e-3;
    it.chi2 = (double *) d_scratch;
    it.active = d_scratch + (size_t) n * 8;
    it.depth = it.active + n;
    it.bad = it.depth + n;
    it.p3p_outlier = it.bad + n;
    it.out = (PnpOut *) out;
    it.p3p = d_sel;
    it.inlier0 = d_inlier0;
    memcpy(dst, &it, sizeof(it));
    return ALVA_OK;
}

int alva_pnp_batch_enqueue(alva_ctx *ctx, const void *d_items, int count) {
    ALVA_ARG(ctx && d_items && count > 0);
    hipLaunchKernelGGL(k_pnp_batch, dim3(count), dim3(NT), 0, ctx->stream, (const PnpBatchItem *) d_items);
    ALVA_LAUNCH_CHECK();
    return ALVA_OK;
}

// the acceptance logic of alva_compute_pose_collect on one camera's PnpOut (status 0 / 1 / 2; pose written when the solver wrote one);
// *needs_more_draws = the P3P stage ran short of valid hypotheses with H draws (the caller redoes that camera with a longer prefix)
int alva_pnp_out_decode(const void *out, int p3p_iters, int H, int max_draws, double *h_pose7, int *h_status, int *needs_more_draws) {
    PnpOut res;
    memcpy(&res, out, sizeof(res));
    *h_status = 0;
    *needs_more_draws = !(res.p3p_n_valid_used >= p3p_iters || H >= max_draws);
    if (*needs_more_draws) return ALVA_OK;
    if (!res.p3p_ok) return ALVA_OK;
    *h_status = 1;
    if (res.n_bad == res.n_active) return ALVA_OK;
    memcpy(h_pose7, res.pose, sizeof(res.pose));
    const int inliers = res.n_active - res.n_bad;
    bool finite = true;
    for (int c = 0; c < 3; c++) finite = finite && std::isfinite(res.pose[c]);
    if (res.ok && inliers >= 5 && res.n_bad <= 0.5 * res.n_active && finite) *h_status = 2;
    return ALVA_OK;
}
