// a4: pyramidal Lucas-Kanade and the reference's forward-backward KLT, one wavefront per keypoint.
//
// Restates cv::calcOpticalFlowPyrLK (video/src/lkpyramid.cpp:1239-1404; per point per level
// LKTrackerInvoker::operator(), :183-724) with flags USE_INITIAL_FLOW | LK_GET_MIN_EIGENVALS, and
// FeatureTracker::fbKltTracking (src/slam/src/feature_tracker.cpp:5-111).
//
// Bit-exact by construction, including the float accumulations: the reference build sums the 9x9
// window in the CV_SIMD128 order -- per row 8 "vector" pixels (lane l takes pixels l and l+4) plus
// one scalar pixel, lanes folded as (a0+a2)+(a1+a3) (core/hal/intrin_sse.hpp:1690-1711).  Here the 81
// bilinear taps of a window are evaluated lane-parallel (pixel p on lane p, pixels 64..80 on lanes
// 0..16), written to LDS as integers, and 15 (A matrix) / 10 (b vector) lanes then replay exactly
// those sequential float chains.  All scalars of the iteration (weights, A, D, delta, next) are
// wave-uniform and computed redundantly by every lane, so control flow never diverges.
//
// Memory: the gathers hit the padded pyramid levels (gray u8 + interleaved int16 Ix,Iy) produced by
// image.hip; one frame's pyramid is ~2 MB at 640x480, i.e. L2-resident -- this stage is
// latency/gather-bound, not HBM-bound.  One 64-thread workgroup per keypoint gives
// N >= 2000 independent waves (8/CU), which is what hides the L2 latency.
//
// This translation unit must be compiled with -ffp-contract=off.
#include "common.hpp"

namespace {

constexpr int WIN = 9;
constexpr int NPX = WIN * WIN;  // 81
constexpr int MAXL = 8;

struct LkLevel {
    const uint8_t *gray;
    const int16_t *deriv;
    int gpitch, dpitch;  // bytes
    int w, h;
};

struct LkPyr {
    LkLevel lv[MAXL];
    int nlevels;
};

struct LkShared {
    short I[NPX + 3];
    short dIx[NPX + 3];
    short dIy[NPX + 3];
    int px[NPX + 3];  // diff * Ix
    int py[NPX + 3];  // diff * Iy
};

__device__ __forceinline__ int descale(int x, int n) { return (x + (1 << (n - 1))) >> n; }

struct Weights {
    int w00, w01, w10, w11;
};

__device__ __forceinline__ Weights bilinear_weights(float a, float b) {
    // lkpyramid.cpp:232-239
    const float W14 = 16384.f;
    float oma = 1.f - a, omb = 1.f - b;
    Weights w;
    w.w00 = __float2int_rn((oma * omb) * W14);
    w.w01 = __float2int_rn((a * omb) * W14);
    w.w10 = __float2int_rn((oma * b) * W14);
    w.w11 = (1 << 14) - w.w00 - w.w01 - w.w10;
    return w;
}

// One point, one pyramid level (lkpyramid.cpp:199-680).  All arguments and results are wave-uniform.
__device__ void lk_level(LkShared &sh, const LkLevel &I, const LkLevel &J, int level, int maxLevel, int maxCount, double epsilon,
                         float minEigThreshold, float ptx, float pty, float &nx, float &ny, int &status, float &err) {
    const int lane = threadIdx.x;
    const float halfWin = (WIN - 1) * 0.5f;
    const float lscale = 1.0f / (float) (1 << level);  // exact power of two
    float prevx = ptx * lscale, prevy = pty * lscale;
    float nextx, nexty;
    if (level == maxLevel) {
        nextx = nx * lscale;
        nexty = ny * lscale;
    } else {
        nextx = nx * 2.f;
        nexty = ny * 2.f;
    }
    nx = nextx;
    ny = nexty;
    prevx -= halfWin;
    prevy -= halfWin;
    const int ipx = (int) floorf(prevx), ipy = (int) floorf(prevy);
    if (ipx < -WIN || ipx >= I.w || ipy < -WIN || ipy >= I.h) {
        if (level == 0) {
            status = 0;
            err = 0.f;
        }
        return;
    }
    Weights wt = bilinear_weights(prevx - (float) ipx, prevy - (float) ipy);

    // ---- patch extraction: pixel p -> lane p (and p+64 for lanes < 17) ------------------------------
    short rI[2], rIx[2], rIy[2];
#pragma unroll
    for (int r = 0; r < 2; r++) {
        const int p = lane + 64 * r;
        rI[r] = rIx[r] = rIy[r] = 0;
        if (p < NPX) {
            const int y = p / WIN, x = p - y * WIN;
            const uint8_t *src = I.gray + (ptrdiff_t) (y + ipy) * I.gpitch + (x + ipx);
            const int ival = descale(src[0] * wt.w00 + src[1] * wt.w01 + src[I.gpitch] * wt.w10 + src[I.gpitch + 1] * wt.w11, 9);
            const uint8_t *drow = reinterpret_cast<const uint8_t *>(I.deriv) + (ptrdiff_t) (y + ipy) * I.dpitch + (ptrdiff_t) (x + ipx) * 4;
            const short2 d00 = *reinterpret_cast<const short2 *>(drow);
            const short2 d01 = *reinterpret_cast<const short2 *>(drow + 4);
            const short2 d10 = *reinterpret_cast<const short2 *>(drow + I.dpitch);
            const short2 d11 = *reinterpret_cast<const short2 *>(drow + I.dpitch + 4);
            const int ixval = descale(d00.x * wt.w00 + d01.x * wt.w01 + d10.x * wt.w10 + d11.x * wt.w11, 14);
            const int iyval = descale(d00.y * wt.w00 + d01.y * wt.w01 + d10.y * wt.w10 + d11.y * wt.w11, 14);
            rI[r] = (short) ival;
            rIx[r] = (short) ixval;
            rIy[r] = (short) iyval;
            sh.dIx[p] = (short) ixval;
            sh.dIy[p] = (short) iyval;
        }
    }
    __syncthreads();
    // ---- A = sum of dI dI^T in the reference's SIMD128 order (15 chains on lanes 0..14) ---------------
    float acc = 0.f;
    if (lane < 15) {
        const int comp = lane / 5, ch = lane - comp * 5;
        for (int y = 0; y < WIN; y++) {
            if (ch < 4) {
#pragma unroll
                for (int half = 0; half < 2; half++) {
                    const int p = y * WIN + ch + 4 * half;
                    const float fx = (float) sh.dIx[p], fy = (float) sh.dIy[p];
                    const float prod = comp == 0 ? fx * fx : (comp == 1 ? fx * fy : fy * fy);
                    acc = prod + acc;
                }
            } else {
                const int p = y * WIN + 8;
                const int ix = sh.dIx[p], iy = sh.dIy[p];
                const int prod = comp == 0 ? ix * ix : (comp == 1 ? ix * iy : iy * iy);
                acc += (float) prod;
            }
        }
    }
    float A[3];
#pragma unroll
    for (int k = 0; k < 3; k++) {
        const float q0 = __shfl(acc, 5 * k + 0), q1 = __shfl(acc, 5 * k + 1), q2 = __shfl(acc, 5 * k + 2), q3 = __shfl(acc, 5 * k + 3);
        float s = __shfl(acc, 5 * k + 4);
        s += (q0 + q2) + (q1 + q3);
        A[k] = s * (1.f / (1 << 20));
    }
    const float A11 = A[0], A12 = A[1], A22 = A[2];
    float D = A11 * A22 - A12 * A12;
    const float dA = A11 - A22;
    // sqrtf and '/' are IEEE correctly rounded here (-fhip-fp32-correctly-rounded-divide-sqrt); the __fsqrt_rn
    // intrinsic is NOT (it lowers to the native approximate v_sqrt_f32)
    const float minEig = ((A22 + A11) - sqrtf(dA * dA + (4.f * A12) * A12)) / (float) (2 * WIN * WIN);
    err = minEig;
    if (minEig < minEigThreshold || D < 1.1920928955078125e-07f) {
        if (level == 0) status = 0;
        return;
    }
    D = 1.f / D;
    nextx -= halfWin;
    nexty -= halfWin;
    float pdx = 0.f, pdy = 0.f;
    for (int j = 0; j < maxCount; j++) {
        const int inx = (int) floorf(nextx), iny = (int) floorf(nexty);
        if (inx < -WIN || inx >= J.w || iny < -WIN || iny >= J.h) {
            if (level == 0) status = 0;
            break;
        }
        wt = bilinear_weights(nextx - (float) inx, nexty - (float) iny);
        __syncthreads();  // previous iteration's chain reads are done before px/py are overwritten
#pragma unroll
        for (int r = 0; r < 2; r++) {
            const int p = lane + 64 * r;
            if (p < NPX) {
                const int y = p / WIN, x = p - y * WIN;
                const uint8_t *src = J.gray + (ptrdiff_t) (y + iny) * J.gpitch + (x + inx);
                const int jval = descale(src[0] * wt.w00 + src[1] * wt.w01 + src[J.gpitch] * wt.w10 + src[J.gpitch + 1] * wt.w11, 9);
                const int diff = (int) (short) (jval - rI[r]);
                sh.px[p] = diff * rIx[r];
                sh.py[p] = diff * rIy[r];
            }
        }
        __syncthreads();
        // b chains (lkpyramid.cpp:553-562, 628-646): lanes 0..7 = (pixel pair (q, q+4), component),
        // lanes 8,9 = scalar pixel 8
        float bacc = 0.f;
        if (lane < 10) {
            const int comp = lane & 1;
            const int *src = comp ? sh.py : sh.px;
            if (lane < 8) {
                const int q = lane >> 1;
                for (int y = 0; y < WIN; y++) bacc += (float) (src[y * WIN + q] + src[y * WIN + q + 4]);
            } else {
                for (int y = 0; y < WIN; y++) bacc += (float) src[y * WIN + 8];
            }
        }
        float ib[2];
#pragma unroll
        for (int c = 0; c < 2; c++) {
            const float s0 = __shfl(bacc, 0 + c) + __shfl(bacc, 4 + c);  // qb0[0|1] + qb1[0|1]
            const float s2 = __shfl(bacc, 2 + c) + __shfl(bacc, 6 + c);  // qb0[2|3] + qb1[2|3]
            float s = __shfl(bacc, 8 + c);
            s += (s0 + 0.f) + (s2 + 0.f);
            ib[c] = s;
        }
        const float b1 = ib[0] * (1.f / (1 << 20)), b2 = ib[1] * (1.f / (1 << 20));
        const float dx = (A12 * b2 - A22 * b1) * D;
        const float dy = (A12 * b1 - A11 * b2) * D;
        nextx += dx;
        nexty += dy;
        nx = nextx + halfWin;
        ny = nexty + halfWin;
        if ((double) dx * (double) dx + (double) dy * (double) dy <= epsilon) break;
        if (j > 0 && fabs((double) (dx + pdx)) < 0.01 && fabs((double) (dy + pdy)) < 0.01) {
            nx -= dx * 0.5f;
            ny -= dy * 0.5f;
            break;
        }
        pdx = dx;
        pdy = dy;
    }
    __syncthreads();  // LDS reuse by the next level
}

// mode 0: plain calcOpticalFlowPyrLK (next in/out, status, err).
// mode 1: FeatureTracker::fbKltTracking (prior in/out, status).
__global__ void __launch_bounds__(64) k_klt(LkPyr P, LkPyr C, int mode, int maxLevel, int maxCount, double epsilon, float errThresh,
                                            float fbDist, const float *__restrict__ pts, float *__restrict__ nextio,
                                            uint8_t *__restrict__ status_out, float *__restrict__ err_out, int n) {
    __shared__ LkShared sh;
    const int kp = blockIdx.x;
    if (kp >= n) return;
    const float ptx = pts[2 * kp], pty = pts[2 * kp + 1];
    float nx = nextio[2 * kp], ny = nextio[2 * kp + 1];
    int status = 1;
    float err = 0.f;
    for (int level = maxLevel; level >= 0; level--)
        lk_level(sh, P.lv[level], C.lv[level], level, maxLevel, maxCount, epsilon, 1e-4f, ptx, pty, nx, ny, status, err);
    if (mode == 0) {
        if (threadIdx.x == 0) {
            nextio[2 * kp] = nx;
            nextio[2 * kp + 1] = ny;
            status_out[kp] = (uint8_t) status;
            err_out[kp] = err;
        }
        return;
    }
    // feature_tracker.cpp:48-73: gate on status, err (min eigenvalue) <= errThresh, inBorder(level-0 size)
    int ok = status && !(err > errThresh);
    const float fw = (float) C.lv[0].w, fh = (float) C.lv[0].h;
    ok = ok && (1.0f <= nx && nx < fw - 1.0f && 1.0f <= ny && ny < fh - 1.0f);
    if (ok) {
        // backward LK on level 0 only, initial flow = the original point (:84-87)
        float bx = ptx, by = pty;
        int st2 = 1;
        float err2 = 0.f;
        lk_level(sh, C.lv[0], P.lv[0], 0, 0, maxCount, epsilon, 1e-4f, nx, ny, bx, by, st2, err2);
        if (!st2) ok = 0;
        else {
            const float ddx = ptx - bx, ddy = pty - by;
            const double nrm = sqrt((double) ddx * (double) ddx + (double) ddy * (double) ddy);  // cv::norm(Point2f), :103
            if (nrm > (double) fbDist) ok = 0;
        }
    }
    if (threadIdx.x == 0) {
        nextio[2 * kp] = nx;
        nextio[2 * kp + 1] = ny;
        status_out[kp] = (uint8_t) ok;
    }
}

int fill_pyr(const alva_pyramid *p, LkPyr &out) {
    out.nlevels = p->nlevels;
    for (int l = 0; l < p->nlevels && l < MAXL; l++) {
        const alva_level &L = p->lv[l];
        out.lv[l].gray = L.gray;
        out.lv[l].deriv = L.deriv;
        out.lv[l].gpitch = (int) L.gray_pitch;
        out.lv[l].dpitch = (int) L.deriv_pitch;
        out.lv[l].w = L.w;
        out.lv[l].h = L.h;
    }
    return 0;
}

int launch(alva_ctx *ctx, const alva_pyramid *prev, const alva_pyramid *curr, int mode, int num_levels, int max_iters, float eps,
           float err_thresh, float fb_dist, const float *d_pts, float *d_nextio, uint8_t *d_status, float *d_err, int n) {
    ALVA_ARG(ctx && prev && curr && n >= 0 && num_levels >= 0);
    if (n == 0) return ALVA_OK;
    ALVA_ARG(d_pts && d_nextio && d_status);
    ALVA_ARG(prev->win == WIN && curr->win == WIN);  // kltWinSizeWH_ = 9 (state.hpp:53); the lane layout is specific to it
    ALVA_ARG(prev->nlevels == curr->nlevels && prev->lv[0].w == curr->lv[0].w && prev->lv[0].h == curr->lv[0].h);
    LkPyr P, C;
    fill_pyr(prev, P);
    fill_pyr(curr, C);
    int maxLevel = num_levels;
    if (prev->nlevels - 1 < maxLevel) maxLevel = prev->nlevels - 1;  // lkpyramid.cpp:1318-1319; feature_tracker.cpp:19-22
    int maxCount = max_iters < 0 ? 0 : (max_iters > 100 ? 100 : max_iters);  // :1358-1361
    double epsilon = (double) eps;
    epsilon = epsilon < 0 ? 0 : (epsilon > 10 ? 10 : epsilon);
    epsilon *= epsilon;  // :1365
    hipLaunchKernelGGL(k_klt, dim3(n), dim3(64), 0, ctx->stream, P, C, mode, maxLevel, maxCount, epsilon, err_thresh, fb_dist, d_pts,
                       d_nextio, d_status, d_err, n);
    ALVA_LAUNCH_CHECK();
    return ALVA_OK;
}

}  // namespace

extern "C" int alva_lk_track(alva_ctx *ctx, const alva_pyramid *prev, const alva_pyramid *next, int num_levels, int max_iters,
                             float eps, const float *d_pts, float *d_next, uint8_t *d_status, float *d_err, int n) {
    ALVA_ARG(n == 0 || d_err);
    return launch(ctx, prev, next, 0, num_levels, max_iters, eps, 0.f, 0.f, d_pts, d_next, d_status, d_err, n);
}

extern "C" int alva_fbklt_track(alva_ctx *ctx, const alva_pyramid *prev, const alva_pyramid *curr, int num_levels, float err_thresh,
                                float fb_dist, int max_iters, float eps, const float *d_pts, float *d_prior, uint8_t *d_status,
                                int n) {
    return launch(ctx, prev, curr, 1, num_levels, max_iters, eps, err_thresh, fb_dist, d_pts, d_prior, d_status, nullptr, n);
}
