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
#include "wave_utils.hpp"

namespace {

constexpr int WIN = 9;
constexpr int NPX = WIN * WIN;  // 81
constexpr int MAXL = 8;
// The iteration loop is a chain of dependent steps, and the slowest keypoint (30 iterations on every level) sets the
// kernel time, so its latency is what matters: the searched image is read through a 16x16 LDS tile (the 10x10 bilinear
// footprint + 3 px of travel each way) that is re-staged from L2 only when the window leaves it.
constexpr int TR = 3, TW = WIN + 1 + 2 * TR;

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
    short2 dxy[NPX + 3];  // (Ix, Iy) of the template window
    int2 pxy[NPX + 3];    // (diff * Ix, diff * Iy) of the current iteration
    uint8_t jt[TW * TW];  // tile of the searched image around the current window (see lk_level)
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

    // ---- patch extraction: pixel p -> lane p, pixels 64..80 -> lanes 0..16 (the other lanes redo pixel 80: no branch) ----
    short rI[2], rIx[2], rIy[2];
    int toff[2];
#pragma unroll
    for (int r = 0; r < 2; r++) {
        const int p = min(lane + 64 * r, NPX - 1);
        const int y = p / WIN, x = p - y * WIN;
        toff[r] = y * TW + x;
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
        sh.dxy[p] = make_short2((short) ixval, (short) iyval);
    }
    __syncthreads();
    // ---- A = sum of dI dI^T in the reference's SIMD128 order: 15 chains on lanes 0..14 = (component, vector lane 0..3 |
    // scalar pixel 8).  Branch-free: every lane runs the same 18 adds (the scalar chain adds +0.f for its missing half;
    // lanes >= 15 repeat lane 14).  (float) ix * (float) iy == (float) (ix * iy): both round the same exact product.
    float acc = 0.f;
    {
        const int l15 = min(lane, 14), comp = l15 / 5, ch = l15 - comp * 5;
        const bool two = ch < 4;
        const int qa = two ? ch : 8, qb = two ? ch + 4 : 8;
#pragma unroll
        for (int y = 0; y < WIN; y++) {
            const short2 da = sh.dxy[y * WIN + qa], db = sh.dxy[y * WIN + qb];
            const float fxa = (float) da.x, fya = (float) da.y, fxb = (float) db.x, fyb = (float) db.y;
            const float p1 = (comp == 2 ? fya : fxa) * (comp == 0 ? fxa : fya);
            float p2 = (comp == 2 ? fyb : fxb) * (comp == 0 ? fxb : fyb);
            p2 = two ? p2 : 0.f;
            acc = p1 + acc;
            acc = p2 + acc;
        }
    }
    float A[3];
#pragma unroll
    for (int k = 0; k < 3; k++) {
        const float q0 = lane_bcast(acc, 5 * k + 0), q1 = lane_bcast(acc, 5 * k + 1), q2 = lane_bcast(acc, 5 * k + 2),
                    q3 = lane_bcast(acc, 5 * k + 3);
        float sres = lane_bcast(acc, 5 * k + 4);
        sres += (q0 + q2) + (q1 + q3);
        A[k] = sres * (1.f / (1 << 20));
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
    int tx0 = 0x40000000, ty0 = 0x40000000;  // tile origin in image coordinates (invalid: staged on first use)
    for (int j = 0; j < maxCount; j++) {
        const int inx = (int) floorf(nextx), iny = (int) floorf(nexty);
        if (inx < -WIN || inx >= J.w || iny < -WIN || iny >= J.h) {
            if (level == 0) status = 0;
            break;
        }
        wt = bilinear_weights(nextx - (float) inx, nexty - (float) iny);
        if (inx < tx0 || inx > tx0 + 2 * TR || iny < ty0 || iny > ty0 + 2 * TR) {  // wave-uniform
            tx0 = inx - TR;
            ty0 = iny - TR;
            __syncthreads();
            // lane -> row lane/4, 4 consecutive columns; coordinates clamped to the padded level (the window itself
            // never leaves it, lkpyramid.cpp:518-523, so clamped bytes are never used)
            const int row = lane >> 2, c0 = (lane & 3) * 4;
            const int gy = min(max(ty0 + row, -WIN), J.h + WIN - 1);
            const uint8_t *srow = J.gray + (ptrdiff_t) gy * J.gpitch;
#pragma unroll
            for (int k = 0; k < 4; k++) sh.jt[row * TW + c0 + k] = srow[min(max(tx0 + c0 + k, -WIN), J.w + WIN - 1)];
        }
        __syncthreads();  // tile staged; previous iteration's chain reads are done before px/py are overwritten
        const int tbase = (iny - ty0) * TW + (inx - tx0);
#pragma unroll
        for (int r = 0; r < 2; r++) {
            const int p = min(lane + 64 * r, NPX - 1);
            const uint8_t *src = sh.jt + toff[r] + tbase;
            const int jval = descale(src[0] * wt.w00 + src[1] * wt.w01 + src[TW] * wt.w10 + src[TW + 1] * wt.w11, 9);
            const int diff = (int) (short) (jval - rI[r]);
            sh.pxy[p] = make_int2(diff * rIx[r], diff * rIy[r]);
        }
        __syncthreads();
        // b chains (lkpyramid.cpp:553-562, 628-646): lanes 0..7 = (vector lane q: pixels q and q+4, component),
        // lanes 8,9 = scalar pixel 8; branch-free like the A chains
        float bacc = 0.f;
        {
            const int l10 = min(lane, 9), comp = l10 & 1, q = l10 >> 1;
            const bool two = q < 4;
            const int qa = two ? q : 8, qb = two ? q + 4 : 8;
            const int *src = reinterpret_cast<const int *>(sh.pxy) + comp;
#pragma unroll
            for (int y = 0; y < WIN; y++) {
                const int va = src[2 * (y * WIN + qa)];
                int vb = src[2 * (y * WIN + qb)];
                vb = two ? vb : 0;
                bacc += (float) (va + vb);
            }
        }
        float ib[2];
#pragma unroll
        for (int c = 0; c < 2; c++) {
            const float s0 = lane_bcast(bacc, 0 + c) + lane_bcast(bacc, 4 + c);  // qb0[0|1] + qb1[0|1]
            const float s2 = lane_bcast(bacc, 2 + c) + lane_bcast(bacc, 6 + c);  // qb0[2|3] + qb1[2|3]
            float sres = lane_bcast(bacc, 8 + c);
            sres += (s0 + 0.f) + (s2 + 0.f);
            ib[c] = sres;
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
                                            float fbDist, const float *__restrict__ pts, const float *init,
                                            float *nextio, uint8_t *__restrict__ status_out, float *__restrict__ err_out,
                                            int n) {
    __shared__ LkShared sh;
    // XCD-aware order: workgroup b runs on XCD b % 8, and each XCD has its own L2.  Giving every XCD one CONTIGUOUS
    // eighth of the keypoint list (callers keep keypoints in spatial / grid order) keeps an image region in one L2
    // instead of pulling the whole pyramid through all eight.
    const int per = gridDim.x >> 3;
    const int kp = (blockIdx.x & 7) * per + (blockIdx.x >> 3);
    if (kp >= n) return;
    const float ptx = pts[2 * kp], pty = pts[2 * kp + 1];
    float nx = init[2 * kp], ny = init[2 * kp + 1];  // initial flow; may be the same buffer as the output
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
           float err_thresh, float fb_dist, const float *d_pts, const float *d_init, float *d_nextio, uint8_t *d_status, float *d_err, int n) {
    ALVA_ARG(ctx && prev && curr && n >= 0 && num_levels >= 0);
    if (n == 0) return ALVA_OK;
    ALVA_ARG(d_pts && d_init && d_nextio && d_status);
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
    hipLaunchKernelGGL(k_klt, dim3(8 * alva_divup(n, 8)), dim3(64), 0, ctx->stream, P, C, mode, maxLevel, maxCount, epsilon, err_thresh, fb_dist, d_pts,
                       d_init, d_nextio, d_status, d_err, n);
    ALVA_LAUNCH_CHECK();
    return ALVA_OK;
}

}  // namespace

extern "C" int alva_lk_track(alva_ctx *ctx, const alva_pyramid *prev, const alva_pyramid *next, int num_levels, int max_iters,
                             float eps, const float *d_pts, float *d_next, uint8_t *d_status, float *d_err, int n) {
    ALVA_ARG(n == 0 || d_err);
    return launch(ctx, prev, next, 0, num_levels, max_iters, eps, 0.f, 0.f, d_pts, d_next, d_next, d_status, d_err, n);
}

extern "C" int alva_fbklt_track(alva_ctx *ctx, const alva_pyramid *prev, const alva_pyramid *curr, int num_levels, float err_thresh,
                                float fb_dist, int max_iters, float eps, const float *d_pts, float *d_prior, uint8_t *d_status,
                                int n) {
    return launch(ctx, prev, curr, 1, num_levels, max_iters, eps, err_thresh, fb_dist, d_pts, d_prior, d_prior, d_status, nullptr, n);
}

// alva_fbklt_track with the prior read from one buffer and the result written to another (internal: the per-frame driver
// tracks from the caller's keypoint buffer without first copying it)
int alva_fbklt_track_to(alva_ctx *ctx, const alva_pyramid *prev, const alva_pyramid *curr, int num_levels, float err_thresh, float fb_dist,
                        int max_iters, float eps, const float *d_pts, const float *d_prior_in, float *d_out, uint8_t *d_status, int n) {
    return launch(ctx, prev, curr, 1, num_levels, max_iters, eps, err_thresh, fb_dist, d_pts, d_prior_in, d_out, d_status, nullptr, n);
}
