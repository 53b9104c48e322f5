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
#pragma clang fp contract(fast)
#include "lm_device.hpp"
#include "wave_utils.hpp"
#include "pose_internal.hpp"
#include <algorithm>
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
};

struct PnpOut {
    double pose[7];
    double info[8];
    int ok, n_bad;
    int p3p_ok, n_active;  // chained mode: P3P verdict and the number of points handed to the refinement
    int p3p_n_valid_used, seq;
    double pose_p3p[7];  // chained mode: the accepted P3P pose as the refinement starts from it (normalised quaternion)
};

struct PnpShared {
    double part[NW][NACC];
    double acc[NACC];
    double x[7], cand[7];
    int flag;
    // minimiser state (touched by lane 0 only; kept in LDS so that dynamic indexing never goes to scratch memory)
    double H[36], g[6], scale[6], diag[6], Hs[36], gs[6], M[36], y[6], step[6], delta[6];
};

enum { F_DONE = 0, F_EVAL_CAND = 1, F_RETRY = 2, F_ACCEPT = 3, F_FAIL = 4 };

__device__ __forceinline__ int tri(int a, int b) {  // index of (a,b), a <= b, in the packed upper triangle of a 6x6
    return a * 6 - a * (a - 1) / 2 + (b - a);
}

// Block-wide evaluation at pose p7: cost (+ packed J^T J and J^T r when WANT_J) into sh.acc.
template<bool WANT_J>
__device__ void eval(PnpShared &sh, const PnpArgs &A, const double *p7, int robust, const uint8_t *active, double *chi2_out,
                     uint8_t *depth_out) {
    Se3 T;
    se3_from_pose7(p7, T);
    double acc[32];  // NACC sums + padding: reduced in place by wave_reduce_scatter32 (a second 32-entry copy does not fit the register file)
#pragma unroll
    for (int k = 0; k < 32; k++) acc[k] = 0.0;
    for (int i = threadIdx.x; i < A.n; i += NT) {
        if (!active[i]) continue;
        const double X[3] = {A.wpt[3 * i], A.wpt[3 * i + 1], A.wpt[3 * i + 2]};
        double r[2], JR[6], chi2;
        int dp;
        reproj<WANT_J>(T, A.K, X, A.uv[2 * i], A.uv[2 * i + 1], r, JR, chi2, dp);
        chi2_out[i] = chi2;
        depth_out[i] = (uint8_t) dp;
        double rho0, rho1;
        huber_rho(chi2, A.huber_a, robust, rho0, rho1);
        acc[27] += 0.5 * rho0;
        if (WANT_J) {
            double JH[6];
            times_hat(JR, X, JH);
            const double s = sqrt(rho1);
            double J[12];
#pragma unroll
            for (int rr = 0; rr < 2; rr++)
#pragma unroll
                for (int c = 0; c < 3; c++) {
                    J[6 * rr + c] = -JR[3 * rr + c] * s;
                    J[6 * rr + 3 + c] = JH[3 * rr + c] * s;
                }
            const double r0 = r[0] * s, r1 = r[1] * s;
            int t = 0;
#pragma unroll
            for (int a = 0; a < 6; a++) {
                acc[21 + a] += J[a] * r0 + J[6 + a] * r1;
#pragma unroll
                for (int b = a; b < 6; b++) acc[t++] += J[a] * J[b] + J[6 + a] * J[6 + b];
            }
        }
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (WANT_J) {
        wave_reduce_scatter32(acc);  // lane l ends with the wave total of value l >> 1
        if (!(lane & 1) && (lane >> 1) < NACC) sh.part[wave][lane >> 1] = acc[0];
    } else {
        double v = acc[27];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off);
        if (lane == 0) sh.part[wave][27] = v;
    }
    __syncthreads();
    if (threadIdx.x < NACC && (WANT_J || threadIdx.x == 27)) {
        double v = 0;
        for (int w = 0; w < NW; w++) v += sh.part[w][threadIdx.x];
        sh.acc[threadIdx.x] = v;
    }
    __syncthreads();
}

// One ceres::Solve on the pose in sh.x.  Returns (block-uniformly) 1 = usable, 0 = failure.
__device__ __forceinline__ int solve(PnpShared &sh, const PnpArgs &A, int robust, const uint8_t *active, double *chi2, uint8_t *depth, double *info) {
    double *H = sh.H, *g = sh.g, *scale = sh.scale, *diag = sh.diag;
    double x_cost = 0, gmax = 0, x_norm = -1, initial = 0, mcc = 0;
    LmState lm;
    int iteration = 0, nsucc = 1, invalid = 0, nsummaries = 1;

    eval<true>(sh, A, sh.x, robust, active, chi2, depth);
    if (threadIdx.x == 0) {
        for (int a = 0; a < 6; a++)
            for (int b = a; b < 6; b++) H[6 * a + b] = H[6 * b + a] = sh.acc[tri(a, b)];
        for (int a = 0; a < 6; a++) g[a] = sh.acc[21 + a];
        x_cost = initial = sh.acc[27];
        for (int a = 0; a < 6; a++) {
            scale[a] = 1.0 / (1.0 + sqrt(H[7 * a]));  // trust_region_minimizer.cc:266-275, iteration 0 only
            gmax = fmax(gmax, fabs(g[a]));
        }
    }
    int result = 1;
    for (;;) {
        if (threadIdx.x == 0) {
            int flag;
            if (iteration >= A.max_iters || gmax <= 1e-10 || lm.radius <= 1e-32) {
                flag = F_DONE;
            } else {
                iteration++;
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
                if (!lm.reuse_diagonal) {
#pragma unroll
                    for (int a = 0; a < 6; a++) diag[a] = fmin(fmax(Hr[7 * a], 1e-6), 1e32);
                }
#pragma unroll
                for (int k = 0; k < 36; k++) Mr[k] = Hr[k];
#pragma unroll
                for (int a = 0; a < 6; a++) {
                    Mr[7 * a] += diag[a] / lm.radius;
                    yr[a] = gr[a];
                }
                const bool okstep = chol_solve_fixed<6>(Mr, yr);
                lm.reuse_diagonal = 1;
                double *step = sh.step;
                mcc = 0;
                if (okstep) {
                    double sg = 0, sHs = 0;
#pragma unroll
                    for (int a = 0; a < 6; a++) {
                        step[a] = -yr[a];
                        sg += -yr[a] * gr[a];
                    }
#pragma unroll
                    for (int a = 0; a < 6; a++)
#pragma unroll
                        for (int b = 0; b < 6; b++) sHs += yr[a] * Hr[6 * a + b] * yr[b];
                    mcc = -sg - 0.5 * sHs;  // = -(J s)'(f + J s / 2), trust_region_minimizer.cc:419-431
                }
                if (!okstep || !(mcc > 0)) {
                    if (++invalid >= 5) flag = F_FAIL;
                    else {
                        lm.rejected();
                        nsummaries++;
                        flag = F_RETRY;
                    }
                } else {
                    invalid = 0;
                    double *delta = sh.delta;
                    for (int a = 0; a < 6; a++) delta[a] = step[a] * scale[a];
                    se3_plus(sh.x, delta, sh.cand);
                    flag = F_EVAL_CAND;
                }
            }
            sh.flag = flag;
        }
        __syncthreads();
        int flag = sh.flag;
        __syncthreads();  // everyone has read the flag before thread 0 may overwrite it
        if (flag == F_DONE) break;
        if (flag == F_FAIL) {
            result = 0;
            break;
        }
        if (flag == F_RETRY) continue;
        // The candidate is evaluated WITH its Jacobian: when the step is accepted Ceres re-evaluates at the same point
        // (HandleSuccessfulStep, trust_region_minimizer.cc:809-829), which would produce exactly these sums again.
        eval<true>(sh, A, sh.cand, robust, active, chi2, depth);
        if (threadIdx.x == 0) {
            const double cand_cost = sh.acc[27];
            double sn = 0;
            for (int i = 0; i < 7; i++) sn += (sh.x[i] - sh.cand[i]) * (sh.x[i] - sh.cand[i]);
            if (sqrt(sn) <= 1e-8 * (x_norm + 1e-8)) flag = F_DONE;                       // ParameterToleranceReached
            else if (fabs(x_cost - cand_cost) <= A.ftol * x_cost) flag = F_DONE;        // FunctionToleranceReached
            else {
                const double rel = (x_cost - cand_cost) / mcc;
                if (rel > 1e-3) {
                    double nn = 0;
                    for (int i = 0; i < 7; i++) {
                        sh.x[i] = sh.cand[i];
                        nn += sh.x[i] * sh.x[i];
                    }
                    x_norm = sqrt(nn);
                    lm.accepted(rel);
                    nsucc++;
                    flag = F_ACCEPT;
                    for (int a = 0; a < 6; a++)
                        for (int b = a; b < 6; b++) H[6 * a + b] = H[6 * b + a] = sh.acc[tri(a, b)];
                    gmax = 0;
                    for (int a = 0; a < 6; a++) {
                        g[a] = sh.acc[21 + a];
                        gmax = fmax(gmax, fabs(g[a]));
                    }
                    x_cost = cand_cost;
                    nsummaries++;
                } else {
                    lm.rejected();
                    nsummaries++;
                    flag = F_RETRY;
                }
            }
            sh.flag = flag;
        }
        __syncthreads();
        flag = sh.flag;
        __syncthreads();
        if (flag == F_DONE) break;
    }
    if (threadIdx.x == 0 && info) {
        info[0] = nsummaries;
        info[1] = initial;
        info[2] = x_cost;
        info[3] = nsucc;
    }
    __syncthreads();
    return result;
}

// p3p / inlier0 non-null = chained mode (VisualFrontend::computePose, visual_frontend.cpp:300-375): the initial pose is the
// P3P-LMedS model and only its inliers are refined; the P3P acceptance tests of multi_view_geometry.cpp:82-91 run here.
// `out`, `bad` and `p3p_outlier` may live in pinned host memory (written once, at the end).
__device__ __forceinline__ void pnp_block(const PnpArgs &A, uint8_t *__restrict__ active, double *__restrict__ chi2,
                                          uint8_t *__restrict__ depth, uint8_t *__restrict__ bad, PnpOut *__restrict__ out,
                                          const P3pSelectOut *__restrict__ p3p, const uint8_t *__restrict__ inlier0,
                                          uint8_t *__restrict__ p3p_outlier) {
    __shared__ PnpShared sh;
    __shared__ int s_nbad, s_p3p_ok, s_nact;
    __shared__ double s_info[8];
    if (threadIdx.x < 8) s_info[threadIdx.x] = 0;
    if (threadIdx.x == 0) {
        s_nbad = 0;
        s_nact = 0;
        s_p3p_ok = 1;
        if (p3p) {
            const double *R = p3p->model;
            double e = 0;
            for (int r = 0; r < 3; r++)
                for (int c = 0; c < 3; c++) {
                    const double v = R[3 * r] * R[3 * c] + R[3 * r + 1] * R[3 * c + 1] + R[3 * r + 2] * R[3 * c + 2] - (r == c ? 1.0 : 0.0);
                    e += v * v;
                }
            // + the translation test of visual_frontend.cpp:323 (isInf / isNaN)
            s_p3p_ok = p3p->have_model && p3p->n_inliers >= 5 && sqrt(e) < 1e-10 && isfinite(p3p->model[9]) && isfinite(p3p->model[10]) &&
                       isfinite(p3p->model[11]);
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
                for (int c = 0; c < 3; c++) sh.x[c] = p3p->model[9 + c];
                for (int c = 0; c < 4; c++) sh.x[3 + c] = q[c];
            }
        } else {
            for (int c = 0; c < 7; c++) sh.x[c] = A.pose0[c];
        }
    }
    __syncthreads();
    if (!s_p3p_ok) {
        if (threadIdx.x == 0) {
            out->ok = 0;
            out->p3p_ok = 0;
            out->n_bad = 0;
            out->n_active = 0;
            out->p3p_n_valid_used = p3p->n_valid_used;
        }
        if (threadIdx.x < 8) out->info[threadIdx.x] = 0;
        for (int i = threadIdx.x; i < A.n; i += NT) {
            bad[i] = 0;
            if (p3p_outlier) p3p_outlier[i] = 0;
        }
        return;
    }
    {
        int na = 0;
        for (int i = threadIdx.x; i < A.n; i += NT) {
            const uint8_t a = inlier0 ? inlier0[i] : (uint8_t) 1;
            active[i] = a;
            chi2[i] = 0.0;
            depth[i] = 1;
            na += a;
        }
        atomicAdd(&s_nact, na);
    }
    __syncthreads();
    {
        // normalise like PoseParametersBlock(0, Sophus::SE3d(q, t)) does before the solve
        if (threadIdx.x == 0) {
            Se3 T;
            se3_from_pose7(sh.x, T);
            for (int i = 0; i < 4; i++) sh.x[3 + i] = T.q[i];
        }
        __syncthreads();
        if (p3p && threadIdx.x < 7) out->pose_p3p[threadIdx.x] = sh.x[threadIdx.x];
    }
    int ok = solve(sh, A, A.use_robust, active, chi2, depth, s_info);
    const int nact = s_nact;
    int nb = 0;
    for (int i = threadIdx.x; i < A.n; i += NT) {
        const bool b = active[i] && (chi2[i] > A.chi2_th || !depth[i]);  // multi_view_geometry.cpp:194-207
        bad[i] = b;
        if (p3p_outlier) p3p_outlier[i] = !inlier0[i];
        if (b) {
            nb++;
            if (A.apply_l2) active[i] = 0;
        }
    }
    atomicAdd(&s_nbad, nb);
    __syncthreads();
    const int nbad = s_nbad;
    if (nbad == nact) ok = 0;
    else if (A.apply_l2 && nbad > 0) ok = solve(sh, A, 0, active, chi2, depth, s_info + 4);  // :214-218
    if (threadIdx.x < 7) out->pose[threadIdx.x] = sh.x[threadIdx.x];
    if (threadIdx.x < 8) out->info[threadIdx.x] = s_info[threadIdx.x];
    if (threadIdx.x == 0) {
        out->ok = ok;
        out->n_bad = nbad;
        out->p3p_ok = 1;
        out->n_active = nact;
        out->p3p_n_valid_used = p3p ? p3p->n_valid_used : 0;
    }
}

__global__ void __launch_bounds__(NT) k_pnp(PnpArgs A, uint8_t *__restrict__ active, double *__restrict__ chi2,
                                            uint8_t *__restrict__ depth, uint8_t *__restrict__ bad, PnpOut *__restrict__ out,
                                            const P3pSelectOut *__restrict__ p3p, const uint8_t *__restrict__ inlier0,
                                            uint8_t *__restrict__ p3p_outlier) {
    pnp_block(A, active, chi2, depth, bad, out, p3p, inlier0, p3p_outlier);
    __threadfence_system();   // this thread's writes to `out` / `bad` / `p3p_outlier` (possibly pinned host memory) ...
    __syncthreads();          // ... of every thread ...
    if (threadIdx.x == 0) __hip_atomic_store(&out->seq, A.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);   // ... before the word a host may poll
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

__global__ void __launch_bounds__(NT) k_pnp_batch(const PnpBatchItem *__restrict__ items) {
    const PnpBatchItem &it = items[blockIdx.x];
    pnp_block(it.A, it.active, it.chi2, it.depth, it.bad, it.out, it.p3p, it.inlier0, it.p3p_outlier);
}

}  // namespace

extern "C" int alva_pnp_refine(alva_ctx *ctx, const double *d_uv, const double *d_wpts, int n, double *h_pose7, int max_iters,
                               float chi2_th, int use_robust, int apply_l2_after_robust, float fx, float fy, float cx, float cy,
                               int *h_outliers, int *h_n_outliers, double *h_info, int *h_ok) {
    ALVA_ARG(ctx && h_pose7 && h_outliers && h_n_outliers && h_ok && n >= 0 && max_iters >= 0);
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
    rc = alva_p3p_enqueue(ctx, P.bearings, P.wpts, n, P.p3p_iters, P.p3p_err, P.do_random, P.seed, P.fx, P.fy, P.H, (int *) P.pin, d_sel,
                          d_inl);
    if (rc) return rc;
    P.A.seq = ++P.seq;
    ((PnpOut *) (P.pin + P.poff_out))->seq = 0;   // the staging may be fresh memory; every earlier user of it has completed (polled or synchronised)
    hipLaunchKernelGGL(k_pnp, dim3(1), dim3(NT), 0, ctx->stream, P.A, base + off_act, (double *) base, base + off_dep, P.pin + P.poff_bad,
                       (PnpOut *) (P.pin + P.poff_out), (const P3pSelectOut *) d_sel, (const uint8_t *) d_inl, P.pin + P.poff_po);
    ALVA_LAUNCH_CHECK();
    return ALVA_OK;
}

extern "C" int alva_compute_pose_enqueue(alva_ctx *ctx, const double *d_bearings, const double *d_uv, const double *d_wpts, int n,
                                         int p3p_iters, float p3p_err, int do_random, uint32_t seed, int pnp_iters, float chi2_th, float fx,
                                         float fy, float cx, float cy) {
    ALVA_ARG(ctx && n >= 0 && p3p_iters > 0 && pnp_iters >= 0);
    if (!ctx->pose_pending) {
        ctx->pose_pending = new alva_pose_pending();
        ctx->pose_pending_free = pose_pending_free;
    }
    alva_pose_pending &P = *(alva_pose_pending *) ctx->pose_pending;
    P.active = true;
    P.n = n;
    if (n < 4) return ALVA_OK;  // visual_frontend.cpp:249-257
    ALVA_ARG(d_bearings && d_uv && d_wpts);
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
                if (++spins > (1u << 26)) {
                    ALVA_HIP(alva_stream_sync(ctx->stream));
                    break;
                }
                alva_poll_relax(spins);
            }
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
