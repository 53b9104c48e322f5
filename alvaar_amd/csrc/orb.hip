// a5': FAST-9/16 + NMS, and the full ORB detector (8-level pyramid, Harris ranking, IC angle, rBRIEF).
//
// Replaces cv::FAST(img, kps, thr, true, TYPE_9_16) (features2d/src/fast.cpp:56-292; score fast_score.cpp:120-) and
// cv::ORB::create(n, s, L, 31, 0, 2, HARRIS_SCORE, 31, thr)->detectAndCompute (features2d/src/orb.cpp:784-1218):
//   pyramid   resize(prev, cur, INTER_LINEAR_EXACT): 8-bit fixed-point taps from fval = scale (d + .5) - .5 in IEEE double
//             (imgproc/src/resize.cpp:733-900); the tap tables are built on the host, the pixels on the device
//   FAST      9 contiguous of 16 brighter/darker by > t; score = largest t that keeps it a corner; strict 3x3 NMS
//   cull      border 31, keep response >= (2 n_l)-th largest FAST score (keypoint.cpp:69-90: ties at the cut all kept),
//             Harris 7x7 (orb.cpp:130-177), keep response >= n_l-th largest Harris
//   angle     intensity centroid over the radius-15 disc, OpenCV's polynomial fastAtan2 (mathfuncs_core.simd.hpp:34-71)
//   describe  7x7 sigma-2 Gaussian of each level + 256 steered BRIEF tests (describe.hip)
// Everything up to the angle is integer (or float with a fixed operation order) => bit-exact.  cv::ORB's keypoint ORDER
// inside a level comes from std::nth_element; this implementation emits the same SET in row-major order per level.
//
// HBM traffic: the 8 levels total 3.27 P pixels; each is written once (resize) and read once by the fused
// FAST + NMS kernel through an LDS tile (+4 px halo): the ORB path keeps no score map (plain alva_fast still uses the
// score-map kernels, whose row-major emission order is part of cv::FAST's contract).  All levels are processed by ONE launch
// per stage (blockIdx.y = level) so that a 640x480 frame (3.3 k tiles) fills the 256 CUs; candidates are appended per tile
// with one atomic, every cull is a threshold (order-free), and the survivors are put in row-major order at the very end.
#include "common.hpp"
#include "exact_sincos.hpp"
#include <cmath>

int alva_blur7_batch_launch(alva_ctx *ctx, int n, const uint8_t *const *src, uint8_t *const *dst, const int *w, const int *h,
                            const int *pitch);

size_t alva_blur7_batch_size();
int alva_blur7_batch_fill(void *out, int n, const uint8_t *const *src, uint8_t *const *dst, const int *w, const int *h, const int *pitch);
int alva_blur7_multi_launch(alva_ctx *ctx, const void *d_batches, int count, int n_levels, int total_tiles, hipStream_t on = nullptr);

namespace {

constexpr int MAXLV = 12;

struct Level {
    int w, h, pitch;       // pitch in bytes (multiple of 64)
    size_t img, score, blur;  // byte offsets into the image pool
    int rowOff;            // offset of this level's rows in the row-count arrays
    int candOff, candCap;  // region of this level in the candidate arrays
    int nKeep;             // n_l
    int border;            // 31 for ORB, 0 for plain FAST
    float scale;           // layerScale
    int tabOff;            // offset of this level's resize taps (x then y) in the tap arrays
};

struct OrbDev {
    int nlevels, threshold;
    Level lv[MAXLV];
    uint8_t *pool;          // images | scores | blurred
    int *rowCnt, *rowStart; // per row survivors / exclusive offsets (per level)
    int *n1, *n2, *n3;      // per level counts after NMS, after FAST cull, after Harris cull
    int *hist;              // [MAXLV][256] FAST score histogram, then [MAXLV][FAST_REGIONS] append counters of the fused FAST + NMS kernel
    int *c1x, *c1y, *c1s;   // NMS survivors (row-major per level)
    int *c2x, *c2y;         // after the FAST-score cull
    float *c2r;             // Harris responses
    int *c3x, *c3y;         // after the Harris cull
    float *c3r;
    const int *tapOfs;      // resize tap source offsets
    const int *tapCoef;     // resize tap coefficient (second tap, 0..256), -1 = clamp to first, -2 = clamp to last
    int *h_n3;              // pinned host mirror of n3, written by k_angle_emit (the host reads it after the stream sync)
    int umax[17];
    // the fused pyramid launch (k_pyramid): levels 0 .. pyrFused in ONE launch, a workgroup per column of the pyramid (pyr_column_body)
    int pyrFused;               // 0 = none (then k_copy_level0 + k_resize per level)
    const int *pyrSpan;         // [tile column of level pyrFused][k = 0 .. pyrFused - 1] (x0, nx, owned x0, owned x1) in level k, then the rows' (y0, ny, ...)
};

// One camera of a batched launch (k_*_b: camera = alva_xcd_item().cam): the detector's own state plus the call's arguments.
struct OrbItem {
    OrbDev D;
    const uint8_t *gray;
    size_t gray_pitch;
    float *kp;
    uint8_t *desc;
    int *total;
    int cap, pad;
};

__device__ __forceinline__ void fast_ring(const uint8_t *p, int stride, int d[16]) {
    // ring in the order of makeOffsets (fast_score.cpp:50-80)
    const int v = p[0];
    d[0] = v - p[3 * stride];
    d[1] = v - p[3 * stride + 1];
    d[2] = v - p[2 * stride + 2];
    d[3] = v - p[stride + 3];
    d[4] = v - p[3];
    d[5] = v - p[-stride + 3];
    d[6] = v - p[-2 * stride + 2];
    d[7] = v - p[-3 * stride + 1];
    d[8] = v - p[-3 * stride];
    d[9] = v - p[-3 * stride - 1];
    d[10] = v - p[-2 * stride - 2];
    d[11] = v - p[-stride - 3];
    d[12] = v - p[-3];
    d[13] = v - p[stride - 3];
    d[14] = v - p[2 * stride - 2];
    d[15] = v - p[3 * stride - 1];
}

// 9 contiguous ring pixels darker than v - t or brighter than v + t (fast.cpp:56-292)
__device__ __forceinline__ bool fast_is_corner(const int d[16], int threshold) {
    unsigned dark = 0, bright = 0;
#pragma unroll
    for (int k = 0; k < 16; k++) {
        dark |= (unsigned) (d[k] > threshold) << k;
        bright |= (unsigned) (d[k] < -threshold) << k;
    }
    auto has9 = [](unsigned m) -> bool {  // 9 contiguous set bits on the 16-ring
        m |= m << 16;
        unsigned r = m & (m >> 1);
        r &= r >> 2;
        r &= r >> 4;       // runs of 8
        r &= m >> 8;       // runs of 9
        return (r & 0xffffu) != 0;
    };
    return has9(dark) || has9(bright);
}

// The same test straight from the pixels, for the fused kernel's packed second step: ring pixel q is "darker" iff v - q > t iff q + (t - v) < 0
// and "brighter" iff v - q < -t iff (v + t) - q < 0, so each of the 32 comparisons is one add and one v_alignbit that shifts the sign bit
// into the mask (a compare + select + shift-or is three).  The masks come out in reversed ring order; "9 contiguous" does not care.
__device__ __forceinline__ bool fast_is_corner_at(const uint8_t *p, int stride, int threshold) {
    const int v = p[0], A = threshold - v, B = v + threshold;
    unsigned dark = 0, bright = 0;
#define ALVA_RING(off)                                                                  \
    {                                                                                   \
        const int q = p[(off)];                                                         \
        dark = __builtin_amdgcn_alignbit(dark, (unsigned) (q + A), 31);                 \
        bright = __builtin_amdgcn_alignbit(bright, (unsigned) (B - q), 31);             \
    }
    ALVA_RING(3 * stride) ALVA_RING(3 * stride + 1) ALVA_RING(2 * stride + 2) ALVA_RING(stride + 3)
    ALVA_RING(3) ALVA_RING(-stride + 3) ALVA_RING(-2 * stride + 2) ALVA_RING(-3 * stride + 1)
    ALVA_RING(-3 * stride) ALVA_RING(-3 * stride - 1) ALVA_RING(-2 * stride - 2) ALVA_RING(-stride - 3)
    ALVA_RING(-3) ALVA_RING(stride - 3) ALVA_RING(2 * stride - 2) ALVA_RING(3 * stride - 1)
#undef ALVA_RING
    auto has9 = [](unsigned m) -> bool {
        m |= m << 16;
        unsigned r = m & (m >> 1);
        r &= r >> 2;
        r &= r >> 4;
        r &= m >> 8;
        return (r & 0xffffu) != 0;
    };
    return has9(dark) || has9(bright);
}

// cornerScore<16> (fast_score.cpp:120-): max over the 16 arcs of 9 of min(d) and min(-d), floored at threshold, minus 1.
// The arc minima / maxima come from doubling windows (2, 4, 8, then +1): min and max are exact in any association.
__device__ __forceinline__ int fast_corner_score(const int d[16], int threshold) {
    int n2[16], x2[16], n4[16], x4[16];
#pragma unroll
    for (int k = 0; k < 16; k++) {
        n2[k] = min(d[k], d[(k + 1) & 15]);
        x2[k] = max(d[k], d[(k + 1) & 15]);
    }
#pragma unroll
    for (int k = 0; k < 16; k++) {
        n4[k] = min(n2[k], n2[(k + 2) & 15]);
        x4[k] = max(x2[k], x2[(k + 2) & 15]);
    }
    int a0 = threshold;
#pragma unroll
    for (int k = 0; k < 16; k++) {
        const int mn = min(min(n4[k], n4[(k + 4) & 15]), d[(k + 8) & 15]);
        const int mx = max(max(x4[k], x4[(k + 4) & 15]), d[(k + 8) & 15]);
        a0 = max(a0, max(mn, -mx));
    }
    return a0 - 1;
}

__device__ __forceinline__ int fast_score_at(const uint8_t *p, int stride, int threshold) {
    int d[16];
    fast_ring(p, stride, d);
    if (!fast_is_corner(d, threshold)) return 0;
    return fast_corner_score(d, threshold);
}

// resize level l from level l-1 (INTER_LINEAR_EXACT)
__device__ __forceinline__ void resize_body(const OrbDev &D, int l, const int bx, const int by) {
    const Level &S = D.lv[l - 1], &T = D.lv[l];
    const int x = bx * 64 + (threadIdx.x & 63), y = by * 4 + (threadIdx.x >> 6);
    if (x >= T.w || y >= T.h) return;
    const int *xo = D.tapOfs + T.tabOff, *xc = D.tapCoef + T.tabOff, *yo = xo + T.w, *yc = xc + T.w;
    const uint8_t *src = D.pool + S.img;
    int rows[2], rc[2], nr;
    const int cy = yc[y];
    if (cy == -1) { rows[0] = 0; rc[0] = 256; nr = 1; }
    else if (cy == -2) { rows[0] = S.h - 1; rc[0] = 256; nr = 1; }
    else { rows[0] = yo[y]; rows[1] = yo[y] + 1; rc[0] = 256 - cy; rc[1] = cy; nr = 2; }
    const int cx = xc[x], ox = xo[x];
    unsigned acc = 0;
    for (int k = 0; k < nr; k++) {
        const uint8_t *r = src + (size_t) rows[k] * S.pitch;
        unsigned hv;
        if (cx == -1) hv = (unsigned) r[0] << 8;
        else if (cx == -2) hv = (unsigned) r[S.w - 1] << 8;
        else hv = (unsigned) (256 - cx) * r[ox] + (unsigned) cx * r[ox + 1];
        acc += (unsigned) rc[k] * hv;
    }
    const unsigned v = (acc + 32768u) >> 16;
    D.pool[T.img + (size_t) y * T.pitch + x] = (uint8_t) min(v, 255u);
}

// Four horizontally adjacent outputs per thread (the batched launch): their source bytes lie within 8 consecutive bytes of a row
// (scale 1.2: 4 outputs span < 5 source pixels), fetched as one unaligned 8-byte load per row instead of 8 byte loads, and stored as
// one dword.  Same integer arithmetic per pixel as resize_body; quads touching a clamped tap or the right edge take that path.
__device__ __forceinline__ void resize4_body(const OrbDev &D, int l, const int bx, const int by) {
    const Level &S = D.lv[l - 1], &T = D.lv[l];
    const int x = (bx * 64 + (threadIdx.x & 63)) * 4, y = by * 4 + (threadIdx.x >> 6);
    if (x >= T.w || y >= T.h) return;
    const int *xo = D.tapOfs + T.tabOff, *xc = D.tapCoef + T.tabOff, *yo = xo + T.w, *yc = xc + T.w;
    const uint8_t *src = D.pool + S.img;
    int rows[2], rc[2], nr;
    const int cy = yc[y];
    if (cy == -1) { rows[0] = 0; rc[0] = 256; nr = 1; }
    else if (cy == -2) { rows[0] = S.h - 1; rc[0] = 256; nr = 1; }
    else { rows[0] = yo[y]; rows[1] = yo[y] + 1; rc[0] = 256 - cy; rc[1] = cy; nr = 2; }
    uint8_t *out = D.pool + T.img + (size_t) y * T.pitch + x;
    bool quad = x + 3 < T.w;
    int cx[4] = {0, 0, 0, 0}, ox[4] = {0, 0, 0, 0};
    if (quad) {
#pragma unroll
        for (int j = 0; j < 4; j++) {
            cx[j] = xc[x + j];
            ox[j] = xo[x + j];
        }
        quad = cx[0] >= 0 && cx[1] >= 0 && cx[2] >= 0 && cx[3] >= 0 && ox[3] - ox[0] <= 6 && ox[1] >= ox[0] && ox[2] >= ox[0];
    }
    if (quad) {
        unsigned acc[4] = {0, 0, 0, 0};
        for (int k = 0; k < nr; k++) {
            unsigned long long w8;
            __builtin_memcpy(&w8, src + (size_t) rows[k] * S.pitch + ox[0], 8);   // rows are padded: the 8 bytes exist
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const unsigned pair = (unsigned) (w8 >> (8 * (ox[j] - ox[0])));
                const unsigned hv = (unsigned) (256 - cx[j]) * (pair & 0xffu) + (unsigned) cx[j] * ((pair >> 8) & 0xffu);
                acc[j] += (unsigned) rc[k] * hv;
            }
        }
        unsigned packed = 0;
#pragma unroll
        for (int j = 0; j < 4; j++) packed |= min((acc[j] + 32768u) >> 16, 255u) << (8 * j);
        *reinterpret_cast<unsigned *>(out) = packed;
        return;
    }
    for (int j = 0; j < 4 && x + j < T.w; j++) {
        const int cxx = xc[x + j], oxx = xo[x + j];
        unsigned acc = 0;
        for (int k = 0; k < nr; k++) {
            const uint8_t *r = src + (size_t) rows[k] * S.pitch;
            unsigned hv;
            if (cxx == -1) hv = (unsigned) r[0] << 8;
            else if (cxx == -2) hv = (unsigned) r[S.w - 1] << 8;
            else hv = (unsigned) (256 - cxx) * r[oxx] + (unsigned) cxx * r[oxx + 1];
            acc += (unsigned) rc[k] * hv;
        }
        out[j] = (uint8_t) min((acc + 32768u) >> 16, 255u);
    }
}

__global__ void __launch_bounds__(256) k_resize(OrbDev D, int l) { resize_body(D, l, blockIdx.x, blockIdx.y); }
__global__ void __launch_bounds__(256) k_resize_b(const OrbItem *__restrict__ items, int l, int count, int gx, int gy) {
    const AlvaXcdItem w = alva_xcd_item(count, gx * gy);
    if (w.cam < count) resize4_body(items[w.cam].D, l, w.item % gx, w.item / gx);
}

// FAST score map of every level: 64x16 tile + 3 px halo staged in LDS
constexpr int FT_W = 64, FT_H = 16;
// The fused kernel's tiles append their survivors to the level's candidate list with one atomic per tile.  ONE counter per level meant
// ~900 atomics with a return value on one address from eight XCDs at level 0 of a 1280x720 frame -- serialised at ~50 ns each they WERE the
// kernel's 50 us (its tiles' own work is ~7 us of the chip).  FAST_REGIONS counters per level (behind the score histograms in D.hist,
// zeroed per frame with them), each owning a fixed slice of the list; k_cull_fast walks the slices (the list's order was never defined).
constexpr int FAST_REGIONS = 32;   // (8: 37 + 26 us for FAST + cull at 1280x720; 32: 31 + 23)
constexpr int FAST_HIST_INTS = 12 * 256 + 12 * FAST_REGIONS + 12 * FAST_REGIONS * 256;   // D.hist: flat | region counters | per-region (MAXLV = 12)
__host__ __device__ inline int fast_region_cap(int w, int h) {
    const int nt = ((w + FT_W - 1) / FT_W) * ((h + FT_H - 1) / FT_H);
    return (nt + FAST_REGIONS - 1) / FAST_REGIONS * (FT_W * FT_H / 4);
}
// ---- levels 0 .. F of the pyramid in ONE launch ---------------------------------------------------------------------------------------------
// cv::ORB resizes level l from level l - 1 (orb.cpp:1086-1099, INTER_LINEAR_EXACT): a chain, seven dependent launches of 7 - 8 us each at
// 1280x720 (plus the copy of level 0) although the chip needs < 1 us for any of them.  Every level IS a pure integer function of level 0,
// so a workgroup takes a COLUMN of the pyramid: a PT_W x PT_H tile of the deepest fused level F and everything under it.  It loads the
// tile's footprint in the caller's image into LDS, computes the region of level 1 that the column needs from it, level 2 from that, ...
// (the same taps at the same global coordinates, the same (acc + 2^15) >> 16 per pixel -> the same bytes as the chain), and of every
// level it WRITES the rectangle it owns: the levels are partitioned among the columns by the first source index of each tile (per axis:
// column t owns [first(t), first(t + 1)) of level k; the regions are widened to contain what they own, which matters where a level's
// last pixels are not read by the next level).  No inter-level wait, no second pass: ~1.9 x the pyramid's pixels are evaluated (the
// halos), level 0 is read ~2 x (from L2).  The per-axis tables -- region and owned range of every tile column / row in every level
// -- are built at create time (orb_build), so a workgroup starts with one table read instead of a chain of F dependent tap look-ups.
// (A first form gave every level its own tiles, each recomputing ALL its ancestors: 5.8 x the pixels, 37.7 us; the chain: 72 us.)
#ifndef ALVA_PT_W
#define ALVA_PT_W 16
#define ALVA_PT_H 8
#endif
constexpr int PT_W = ALVA_PT_W, PT_H = ALVA_PT_H;   // (tools/build_variant.sh builds other sizes for A/B runs)
constexpr int PYR_BUF_A = 12288, PYR_BUF_B = 8192;   // even / odd levels' regions (level 0's footprint is the largest)
constexpr int PYR_TAPS = 2048;                       // all stages' taps of one column

#define PYR_STAMP(k) do { if (dbg && threadIdx.x == 0 && tile < 256) dbg[16 * tile + (k)] = wall_clock64(); } while (0)
__device__ __forceinline__ void pyr_column_body(const OrbDev &D, const uint8_t *__restrict__ src, const size_t pitch, const int tile, unsigned long long *dbg) {
    PYR_STAMP(0);
    __shared__ __attribute__((aligned(16))) uint8_t s_a[PYR_BUF_A];
    __shared__ __attribute__((aligned(16))) uint8_t s_b[PYR_BUF_B];
    __shared__ int s_taps[PYR_TAPS];
    // per level: region (x0, nx, y0, ny) and owned range [ox0, ox1) x [oy0, oy1)
    __shared__ int s_x0[MAXLV], s_nx[MAXLV], s_y0[MAXLV], s_ny[MAXLV], s_ox0[MAXLV], s_ox1[MAXLV], s_oy0[MAXLV], s_oy1[MAXLV], s_toff[MAXLV + 1];
    const int F = D.pyrFused, tid = threadIdx.x;
    const Level &T = D.lv[F];
    const int gxt = (T.w + PT_W - 1) / PT_W;
    const int tx = tile % gxt, ty = tile / gxt;
    if (tid < F) {
        const int *e = D.pyrSpan + (tx * F + tid) * 4;
        s_x0[tid] = e[0]; s_nx[tid] = e[1]; s_ox0[tid] = e[2]; s_ox1[tid] = e[3];
    } else if (tid >= 64 && tid < 64 + F) {
        const int k = tid - 64;
        const int *e = D.pyrSpan + (gxt * F + ty * F + k) * 4;
        s_y0[k] = e[0]; s_ny[k] = e[1]; s_oy0[k] = e[2]; s_oy1[k] = e[3];
    } else if (tid == 128) {
        s_x0[F] = tx * PT_W;
        s_nx[F] = min(PT_W, T.w - tx * PT_W);
        s_y0[F] = ty * PT_H;
        s_ny[F] = min(PT_H, T.h - ty * PT_H);
    }
    __syncthreads();
    if (tid == 0) {
        int t = 0;
        for (int k = 1; k <= F; k++) {
            s_toff[k] = t;
            t += s_nx[k] + s_ny[k];
        }
        s_toff[0] = t;                  // total
    }
    PYR_STAMP(1);
    const int lx = tid & 31, ly = tid >> 5;
    // level 0: the column's footprint -> s_a (unaligned dword loads; the last dword of a row by bytes when it would pass the image's edge)
    {
        const int X0 = s_x0[0], NX = s_nx[0], Y0 = s_y0[0], NY = s_ny[0], W0 = D.lv[0].w;
        const int stride = (NX + 3) & ~3, dwr = stride >> 2;
        for (int r = ly; r < NY; r += 8) {
            const uint8_t *row = src + (size_t) (Y0 + r) * pitch + X0;
            for (int c = lx; c < dwr; c += 32) {
                uint32_t v;
                if (X0 + 4 * c + 4 <= W0) __builtin_memcpy(&v, row + 4 * c, 4);
                else {
                    v = 0;
                    for (int j = 0; j < 4; j++)
                        if (X0 + 4 * c + j < W0) v |= (uint32_t) row[4 * c + j] << (8 * j);
                }
                *reinterpret_cast<uint32_t *>(s_a + r * stride + 4 * c) = v;
            }
        }
    }
    __syncthreads();
    PYR_STAMP(2);
    // every stage's taps, relative to the region they read: (first source index) | (weight of the second tap << 16); a clamped tap
    // (-1 / -2: copies the first / last source element) is weight 0 on that element -- 256 * v + 0 * v, the same ufixedpoint16 value
    {
        const int total = s_toff[0];
        for (int i = tid; i < total; i += 256) {
            int k = 1;
            while (k < F && i >= s_toff[k + 1]) k++;
            const Level &K = D.lv[k], &S = D.lv[k - 1];
            int j = i - s_toff[k];
            const bool yax = j >= s_nx[k];
            if (yax) j -= s_nx[k];
            const int g = (yax ? s_y0[k] : s_x0[k]) + j;
            const int tp = K.tabOff + (yax ? K.w : 0) + g;
            const int c = D.tapCoef[tp], of = D.tapOfs[tp];
            const int base = yax ? s_y0[k - 1] : s_x0[k - 1], sn = yax ? S.h : S.w;
            int a, wb;
            if (c == -1) { a = 0; wb = 0; }
            else if (c == -2) { a = sn - 1; wb = 0; }
            else { a = of; wb = c; }
            s_taps[i] = (a - base) | (wb << 16);
        }
    }
    // level 0's owned rectangle: the copy into the pool
    {
        const Level &L0 = D.lv[0];
        const int stride = (s_nx[0] + 3) & ~3, X0 = s_x0[0], Y0 = s_y0[0];
        const int ox0 = s_ox0[0], onx = s_ox1[0] - ox0, oy0 = s_oy0[0], ony = s_oy1[0] - oy0;
        for (int r = ly; r < ony; r += 8)
            for (int c = lx; c < onx; c += 32) D.pool[L0.img + (size_t) (oy0 + r) * L0.pitch + ox0 + c] = s_a[(oy0 - Y0 + r) * stride + (ox0 - X0 + c)];
    }
    __syncthreads();
    PYR_STAMP(3);
    for (int k = 1; k <= F; k++) {
        const uint8_t *in = (k - 1) & 1 ? s_b : s_a;
        uint8_t *outl = k & 1 ? s_b : s_a;
        const int sstride = (s_nx[k - 1] + 3) & ~3;
        const int rx0 = s_x0[k], ry0 = s_y0[k], onx = s_nx[k], ony = s_ny[k], ostride = (onx + 3) & ~3;
        const int *tapx = s_taps + s_toff[k], *tapy = tapx + onx;
        const bool last = k == F;
        // owned rectangle, relative to the region (the deepest level owns its whole tile)
        const int wx0 = last ? 0 : s_ox0[k] - rx0, wx1 = last ? onx : s_ox1[k] - rx0, wy0 = last ? 0 : s_oy0[k] - ry0, wy1 = last ? ony : s_oy1[k] - ry0;
        uint8_t *gout = D.pool + D.lv[k].img + (size_t) ry0 * D.lv[k].pitch + rx0;
        const int gpitch = D.lv[k].pitch;
        for (int xx = lx; xx < onx; xx += 32) {
            const int t = tapx[xx];
            const int xa = t & 0xffff, wb = t >> 16, wa = 256 - wb, xb = xa + (wb != 0);
            const bool ownx = xx >= wx0 && xx < wx1;
#pragma unroll 2
            for (int yy = ly; yy < ony; yy += 8) {
                const int u = tapy[yy];
                const int r0 = u & 0xffff, c1 = u >> 16, c0 = 256 - c1, r1 = r0 + (c1 != 0);
                const uint8_t *p0 = in + r0 * sstride, *p1 = in + r1 * sstride;
                const unsigned h0 = (unsigned) wa * p0[xa] + (unsigned) wb * p0[xb];
                const unsigned h1 = (unsigned) wa * p1[xa] + (unsigned) wb * p1[xb];
                const unsigned acc = (unsigned) c0 * h0 + (unsigned) c1 * h1;
                const uint8_t v = (uint8_t) min((acc + 32768u) >> 16, 255u);
                if (!last) outl[yy * ostride + xx] = v;
                if (ownx && yy >= wy0 && yy < wy1) gout[(size_t) yy * gpitch + xx] = v;
            }
        }
        __syncthreads();
        PYR_STAMP(3 + k);
    }
}

// grid: 8 * ceil(columns / 8) workgroups; workgroup b runs on XCD b % 8 (each with its own L2), so every XCD gets one CONTIGUOUS eighth of
// the columns (row-major) and neighbouring columns' overlapping footprints meet in one L2.  Every workgroup also clears a slice of the
// frame's counters (what k_copy_level0 does for the chain: the score histograms, the append counters, n1 | n2).
__global__ void __launch_bounds__(256) k_pyramid(OrbDev D, const uint8_t *__restrict__ src, size_t pitch, unsigned long long *dbg) {
    const Level &T = D.lv[D.pyrFused];
    const int nt = ((T.w + PT_W - 1) / PT_W) * ((T.h + PT_H - 1) / PT_H), per = (nt + 7) / 8;
    const int b = (int) blockIdx.x, tile = (b & 7) * per + (b >> 3);
    {
        const int total = MAXLV * 256 + MAXLV * FAST_REGIONS + D.nlevels * FAST_REGIONS * 256, nb = (int) gridDim.x;
        for (int k = b * 256 + (int) threadIdx.x; k < total; k += nb * 256) D.hist[k] = 0;
        if (b == 0 && threadIdx.x < 2 * MAXLV) D.n1[threadIdx.x] = 0;   // n1 | n2 (k_cull_fast appends through n2)
    }
    if (tile < nt) pyr_column_body(D, src, pitch, tile, dbg);
}
// batched: camera = alva_xcd_item().cam (a camera's columns on one XCD), item = column
__global__ void __launch_bounds__(256) k_pyramid_b(const OrbItem *__restrict__ items, int count, int per_cam) {
    const AlvaXcdItem w = alva_xcd_item(count, per_cam);
    if (w.cam >= count) return;
    const OrbItem &it = items[w.cam];
    const OrbDev &D = it.D;
    {
        const int total = MAXLV * 256 + MAXLV * FAST_REGIONS + D.nlevels * FAST_REGIONS * 256;
        for (int k = w.item * 256 + (int) threadIdx.x; k < total; k += per_cam * 256) D.hist[k] = 0;
        if (w.item == 0 && threadIdx.x < 2 * MAXLV) D.n1[threadIdx.x] = 0;
    }
    pyr_column_body(D, it.gray, it.gray_pitch, w.item, nullptr);
}

__global__ void __launch_bounds__(256) k_fast_score(OrbDev D) {
    const Level &L = D.lv[blockIdx.y];
    const int tilesX = (L.w + FT_W - 1) / FT_W, tilesY = (L.h + FT_H - 1) / FT_H;
    if ((int) blockIdx.x >= tilesX * tilesY) return;
    const int x0 = (blockIdx.x % tilesX) * FT_W, y0 = (blockIdx.x / tilesX) * FT_H;
    __shared__ uint8_t s[(FT_H + 6) * (FT_W + 8)];
    constexpr int SW = FT_W + 8;
    const uint8_t *img = D.pool + L.img;
    for (int i = threadIdx.x; i < (FT_H + 6) * (FT_W + 6); i += 256) {
        const int ly = i / (FT_W + 6), lx = i % (FT_W + 6);
        const int gx = min(max(x0 + lx - 3, 0), L.w - 1), gy = min(max(y0 + ly - 3, 0), L.h - 1);
        s[ly * SW + lx] = img[(size_t) gy * L.pitch + gx];
    }
    __syncthreads();
    uint8_t *sc = D.pool + L.score;
    for (int i = threadIdx.x; i < FT_W * FT_H; i += 256) {
        const int ly = i / FT_W, lx = i % FT_W, gx = x0 + lx, gy = y0 + ly;
        if (gx >= L.w || gy >= L.h) continue;
        int v = 0;
        if (gx >= 3 && gx < L.w - 3 && gy >= 3 && gy < L.h - 3) v = fast_score_at(s + (ly + 3) * SW + lx + 3, SW, D.threshold);
        sc[(size_t) gy * L.pitch + gx] = (uint8_t) v;
    }
}

// FAST + strict 3x3 non-maximum suppression fused over an LDS tile (the ORB path): gray tile with a 4-px halo in LDS,
// corner test for every pixel of the tile + 1, the few corners are then packed (wave ballot) so that the score is
// computed with dense lanes, scores stay in LDS, NMS survivors are packed per row by ballot and appended to the level's
// candidate list with ONE atomic per tile.  No score map in HBM, no count/scan/emit passes.  The list order depends on
// tile completion order; every later stage is order-free (threshold culls) and k_cull_harris finally sorts by position.
__device__ __forceinline__ void fast_nms_body(const OrbDev &D, const int lvl, const int tile) {
    const Level &L = D.lv[lvl];
    const int tilesX = (L.w + FT_W - 1) / FT_W, tilesY = (L.h + FT_H - 1) / FT_H;
    if ((int) tile >= tilesX * tilesY) return;
    const int x0 = (tile % tilesX) * FT_W, y0 = (tile / tilesX) * FT_H;
    // candidates live in [border, dim - border): tiles wholly outside (+1 px for the NMS neighbours) have nothing to do
    const int lo = max(L.border, 3), hx = L.w - lo, hy = L.h - lo;
    if (x0 >= hx || x0 + FT_W <= lo || y0 >= hy || y0 + FT_H <= lo) return;
    constexpr int GW = FT_W + 8, GH = FT_H + 8, SW = FT_W + 4, SH = FT_H + 2;   // gray tile (halo 4), score tile (halo 1)
    __shared__ __attribute__((aligned(4))) uint8_t g[GH * GW];
    __shared__ uint8_t sc[SH * SW];
    __shared__ unsigned short list[SH * (FT_W + 2)], list2[SH * (FT_W + 2)];
    __shared__ int s_n, s_n2, s_ns, s_base;
    __shared__ unsigned s_surv[FT_W * FT_H / 4 + 64];   // NMS survivors: score | x << 8 | y << 16 (tile coordinates)
    const uint8_t *img = D.pool + L.img;
    if (threadIdx.x == 0) s_n = s_n2 = s_ns = 0;
    // gray tile, a dword per thread and step (x0 - 4 and the level's rows are 4-byte aligned); bytes are clamped one by one only in
    // dwords that cross the image border
    static_assert(GW % 4 == 0, "dword tile rows");
    for (int i = threadIdx.x; i < GH * (GW / 4); i += 256) {
        const int ly = i / (GW / 4), dq = i - ly * (GW / 4);
        const int gy = min(max(y0 + ly - 4, 0), L.h - 1), gx0 = x0 - 4 + 4 * dq;
        const uint8_t *row = img + (size_t) gy * L.pitch;
        uint32_t v;
        if (gx0 >= 0 && gx0 + 3 < L.w) {
            v = *reinterpret_cast<const uint32_t *>(row + gx0);
        } else {
            v = 0;
#pragma unroll
            for (int k = 0; k < 4; k++) v |= (uint32_t) row[min(max(gx0 + k, 0), L.w - 1)] << (8 * k);
        }
        *reinterpret_cast<uint32_t *>(g + ly * GW + 4 * dq) = v;
    }
    __syncthreads();
    const int lane = threadIdx.x & 63;
    // corner test on the (FT_H + 2) x (FT_W + 2) region, in two packed steps.  Step 1, every pixel: the high-speed test -- an arc of
    // 9 on the ring of 16 contains k or k + 8 for every k, so a corner needs (ring 0 or 8) AND (ring 4 or 12) beyond the threshold
    // on the same side (fast.cpp's quick rejection; a necessary condition, so the corner set is unchanged).  Step 2, the survivors
    // only (a quarter of the pixels on textured frames), densely packed again: the full ring.
    for (int i0 = 0; i0 < SH * (FT_W + 2); i0 += 256) {
        const int i = i0 + threadIdx.x;
        bool maybe = false;
        int ly = 0, lx = 0;
        if (i < SH * (FT_W + 2)) {
            ly = i / (FT_W + 2);
            lx = i % (FT_W + 2);
            const int gx = x0 + lx - 1, gy = y0 + ly - 1;
            if (gx >= 3 && gx < L.w - 3 && gy >= 3 && gy < L.h - 3) {
                const uint8_t *p = g + (ly + 3) * GW + lx + 3;
                const int v = p[0], t = D.threshold;
                const int d0 = v - p[3 * GW], d8 = v - p[-3 * GW], d4 = v - p[3], d12 = v - p[-3];
                maybe = ((d0 > t || d8 > t) && (d4 > t || d12 > t)) || ((d0 < -t || d8 < -t) && (d4 < -t || d12 < -t));
            }
            sc[ly * SW + lx] = 0;
        }
        const unsigned long long m = __ballot(maybe);
        int base = 0;
        if (lane == 0 && m) base = atomicAdd(&s_n2, __popcll(m));
        base = __shfl(base, 0);
        if (maybe) list2[base + __popcll(m & ((1ull << lane) - 1ull))] = (unsigned short) (ly * 256 + lx);
    }
    __syncthreads();
    const int n2 = s_n2;
    for (int j0 = 0; j0 < n2; j0 += 256) {
        const int j = j0 + threadIdx.x;
        bool corner = false;
        unsigned short code = 0;
        if (j < n2) {
            code = list2[j];
            const int ly = code >> 8, lx = code & 255;
            corner = fast_is_corner_at(g + (ly + 3) * GW + lx + 3, GW, D.threshold);
        }
        const unsigned long long m = __ballot(corner);
        int base = 0;
        if (lane == 0 && m) base = atomicAdd(&s_n, __popcll(m));
        base = __shfl(base, 0);
        if (corner) list[base + __popcll(m & ((1ull << lane) - 1ull))] = code;
    }
    __syncthreads();
    const int nc = s_n;
    for (int j = threadIdx.x; j < nc; j += 256) {
        const int ly = list[j] >> 8, lx = list[j] & 255;
        int d[16];
        fast_ring(g + (ly + 3) * GW + lx + 3, GW, d);
        sc[ly * SW + lx] = (uint8_t) fast_corner_score(d, D.threshold);
    }
    __syncthreads();
    // NMS over the corner LIST (a few per cent of the tile's pixels; walking all 1024 pixels with nine score reads each was 15 % of the
    // kernel's instructions): the strict local maxima inside the tile proper are packed into LDS (at most a quarter of the pixels can
    // be one), the tile takes its place in its region's slice of the candidate list with ONE atomic, then they are written out
    for (int j0 = 0; j0 < nc; j0 += 256) {
        const int j = j0 + (int) threadIdx.x;
        bool keep = false;
        unsigned packed = 0;
        if (j < nc) {
            const int ly = list[j] >> 8, lx = list[j] & 255;   // score-tile coordinates: pixel (x0 + lx - 1, y0 + ly - 1)
            if (ly >= 1 && ly <= FT_H && lx >= 1 && lx <= FT_W) {
                const int gx = x0 + lx - 1, gy = y0 + ly - 1;
                const uint8_t *q = sc + ly * SW + lx;
                const int v = q[0];
                keep = v > 0 && gx >= lo && gx < hx && gy >= lo && gy < hy;
                keep = keep && v > q[-1] && v > q[1] && v > q[-SW - 1] && v > q[-SW] && v > q[-SW + 1] && v > q[SW - 1] && v > q[SW] && v > q[SW + 1];
                if (L.border > 0) keep = keep && L.w > 2 * L.border && L.h > 2 * L.border;   // KeyPointsFilter::runByImageBorder
                packed = (unsigned) v | ((unsigned) (lx - 1) << 8) | ((unsigned) (ly - 1) << 16);
            }
        }
        const unsigned long long m = __ballot(keep);
        int base = 0;
        if (lane == 0 && m) base = atomicAdd(&s_ns, __popcll(m));
        base = __shfl(base, 0);
        if (keep) s_surv[base + __popcll(m & ((1ull << lane) - 1ull))] = packed;
    }
    __syncthreads();
    const int ns = s_ns;
    if (threadIdx.x == 0) s_base = ns ? atomicAdd(&D.hist[MAXLV * 256 + lvl * FAST_REGIONS + (int) (tile % FAST_REGIONS)], ns) : 0;   // the tile's REGION of the list
    __syncthreads();
    const int rcap = fast_region_cap(L.w, L.h);
    for (int j = threadIdx.x; j < ns; j += 256) {
        const int inr = s_base + j;
        if (inr >= rcap) continue;
        const unsigned pk = s_surv[j];
        const int score = (int) (pk & 255u);
        const int pos = (int) (tile % FAST_REGIONS) * rcap + inr;
        D.c1x[L.candOff + pos] = x0 + (int) ((pk >> 8) & 255u);
        D.c1y[L.candOff + pos] = y0 + (int) (pk >> 16);
        D.c1s[L.candOff + pos] = score;
        // the level's score histogram, one per REGION too: with the append counter's contention gone, ~20 k atomics on a few hundred
        // hot counters from every XCD were the next 12 us (k_cull_fast adds the regions' histograms up)
        atomicAdd(&D.hist[MAXLV * 256 + MAXLV * FAST_REGIONS + (lvl * FAST_REGIONS + (int) (tile % FAST_REGIONS)) * 256 + score], 1);
    }
}

// One frame: a 1-D grid over the tiles of ALL levels (a grid sized for level 0 on every level was 60 % empty workgroups: 7 232 dispatched
// for 2 950 tiles at 1280x720 -- the launch was bound by workgroup dispatch, SQ counters: 1 100 waves resident on average of 8 192).  Each
// level's share is padded to a multiple of 8 and, inside it, workgroup b runs on XCD b % 8 (each with its own L2), so every XCD gets one
// CONTIGUOUS eighth of the level's tiles (row-major): the 4-px halo rows and the 128-byte lines that neighbouring tiles share meet in ONE
// L2 instead of being fetched through up to three (PMC, plain order: 2.06 MB fetched per launch at 640x480 against ~1.0 MB of pyramid;
// the same cure as k_pyr_rest's).  The candidate order was never defined (tile completion order), later stages sort.
// (SQ counters at 1280x720 before the list NMS, profiles/r4h_sq_orb720.txt: 7.97 M VALU wave-instructions for 3.0 M pixels = 170 lane-instructions per pixel, the SIMDs'
// VALU 54 % busy over the launch, 8 waves per SIMD resident (the maximum: 60 VGPRs) -- an instruction-count kernel now, not a latency one)
__global__ void __launch_bounds__(256) k_fast_nms(OrbDev D) {
    int b = (int) blockIdx.x, l = 0, n = 0, per = 0;
    for (; l < D.nlevels; l++) {
        n = ((D.lv[l].w + FT_W - 1) / FT_W) * ((D.lv[l].h + FT_H - 1) / FT_H);
        per = (n + 7) / 8;
        if (b < 8 * per) break;
        b -= 8 * per;
    }
    if (l >= D.nlevels) return;
    const int tile = (b & 7) * per + (b >> 3);
    if (tile >= n) return;
    fast_nms_body(D, l, tile);
}
// batched: blockIdx.x runs over the tiles of ALL levels (a grid sized for level 0 on every level would be 60 % empty workgroups)
__global__ void __launch_bounds__(256) k_fast_nms_b(const OrbItem *__restrict__ items, int count, int per_cam) {
    const AlvaXcdItem w = alva_xcd_item(count, per_cam);
    if (w.cam >= count) return;
    const OrbDev &D = items[w.cam].D;
    int t = w.item, l = 0;
    for (; l < D.nlevels; l++) {
        const int nt = ((D.lv[l].w + FT_W - 1) / FT_W) * ((D.lv[l].h + FT_H - 1) / FT_H);
        if (t < nt) break;
        t -= nt;
    }
    if (l < D.nlevels) fast_nms_body(D, l, t);
}

__device__ __forceinline__ bool nms_keep(const uint8_t *sc, int pitch, int x, int y, const Level &L) {
    if (x < 3 || x >= L.w - 3 || y < 3 || y >= L.h - 3) return false;
    const uint8_t *q = sc + (size_t) y * pitch + x;
    const int s = q[0];
    if (!s) return false;
    if (!(s > q[-1] && s > q[1] && s > q[-pitch - 1] && s > q[-pitch] && s > q[-pitch + 1] && s > q[pitch - 1] && s > q[pitch] && s > q[pitch + 1]))
        return false;
    if (L.border > 0)  // KeyPointsFilter::runByImageBorder (keypoint.cpp:105-117)
        if (!(L.w > 2 * L.border && L.h > 2 * L.border && x >= L.border && x < L.w - L.border && y >= L.border && y < L.h - L.border)) return false;
    return true;
}

// one wave per image row: EMIT = false counts the NMS survivors, EMIT = true writes them in x order
template<bool EMIT>
__global__ void __launch_bounds__(256) k_fast_rows(OrbDev D) {
    const Level &L = D.lv[blockIdx.y];
    const int y = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (y >= L.h) return;
    const uint8_t *sc = D.pool + L.score;
    int base = EMIT ? D.rowStart[L.rowOff + y] : 0;
    int count = 0;
    for (int x0 = 0; x0 < L.w; x0 += 64) {
        const int x = x0 + lane;
        const bool keep = x < L.w && nms_keep(sc, L.pitch, x, y, L);
        const unsigned long long m = __ballot(keep);
        if (EMIT && keep) {
            const int pos = base + count + __popcll(m & ((1ull << lane) - 1ull));
            if (pos < L.candCap) {
                const int s = sc[(size_t) y * L.pitch + x];
                D.c1x[L.candOff + pos] = x;
                D.c1y[L.candOff + pos] = y;
                D.c1s[L.candOff + pos] = s;
                atomicAdd(&D.hist[blockIdx.y * 256 + s], 1);
            }
        }
        count += __popcll(m);
    }
    if (!EMIT && lane == 0) D.rowCnt[L.rowOff + y] = count;
}

// exclusive scan of the per-row counts of one level (one workgroup per level)
__global__ void __launch_bounds__(1024) k_scan_rows(OrbDev D) {
    const Level &L = D.lv[blockIdx.x];
    __shared__ int s[1024];
    int carry = 0;
    for (int b = 0; b < L.h; b += 1024) {
        const int y = b + threadIdx.x;
        const int v = y < L.h ? D.rowCnt[L.rowOff + y] : 0;
        s[threadIdx.x] = v;
        __syncthreads();
        for (int off = 1; off < 1024; off <<= 1) {
            const int t = threadIdx.x >= off ? s[threadIdx.x - off] : 0;
            __syncthreads();
            s[threadIdx.x] += t;
            __syncthreads();
        }
        if (y < L.h) D.rowStart[L.rowOff + y] = carry + s[threadIdx.x] - v;
        carry += s[1023];
        __syncthreads();
    }
    if (threadIdx.x == 0) D.n1[blockIdx.x] = min(carry, L.candCap);
}

// Wave-0 helper for the culls: first bin b (in the order given by `load`) whose inclusive prefix sum exceeds k.
// 256 bins, 4 per lane; returns through LDS words (bin, k - exclusive prefix).  Call with threadIdx.x < 64.
template<typename Load>
__device__ __forceinline__ void wave_find_bin(Load load, unsigned k, unsigned *s_bin, unsigned *s_rem) {
    const int l = threadIdx.x;
    const unsigned c0 = load(4 * l), c1 = load(4 * l + 1), c2 = load(4 * l + 2), c3 = load(4 * l + 3);
    const unsigned tot = c0 + c1 + c2 + c3;
    unsigned incl = tot;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const unsigned v = __shfl_up(incl, off);
        if (l >= off) incl += v;
    }
    const unsigned excl = incl - tot;
    if (k >= excl && k < incl) {
        unsigned r = k - excl;
        int b;
        if (r < c0) b = 0;
        else if (r < c0 + c1) { b = 1; r -= c0; }
        else if (r < c0 + c1 + c2) { b = 2; r -= c0 + c1; }
        else { b = 3; r -= c0 + c1 + c2; }
        *s_bin = (unsigned) (4 * l + b);
        *s_rem = r;
    }
}

// ordered compaction helper: keep[i] decided by the caller's predicate; one workgroup, chunks of 1024
template<typename Pred, typename Emit>
__device__ int compact_ordered(int n, Pred pred, Emit emit) {
    __shared__ int s_w[16];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int carry = 0;
    for (int b = 0; b < n; b += 1024) {
        const int i = b + threadIdx.x;
        const bool k = i < n && pred(i);
        const unsigned long long m = __ballot(k);   // position inside the wave from the ballot, wave offsets from 16 LDS words
        if (lane == 0) s_w[wave] = __popcll(m);
        __syncthreads();
        int before = 0, tot = 0;
#pragma unroll
        for (int w = 0; w < 16; w++) {
            const int c = s_w[w];
            before += w < wave ? c : 0;
            tot += c;
        }
        if (k) emit(i, carry + before + __popcll(m & ((1ull << lane) - 1ull)));
        carry += tot;
        __syncthreads();
    }
    return carry;
}

// cull by FAST score: keep score >= the (2 n_l)-th largest (all ties kept).  One workgroup per REGION of the level's candidate list
// (fast_nms_body appends per region): every workgroup finds the level's threshold from the regions' histograms (8 K ints, L2), then
// filters its own slice and appends the survivors to the level's second list with ONE atomic on n2[l] (zeroed by the frame's first
// launch).  The list's order was never defined (tile completion order) and nothing downstream depends on it: cull_harris_body ranks by
// response and emits in position order.  (One 1024-thread workgroup per level walking the whole list in order was 25.7 us at 1280x720,
// with eight workgroups on the chip.)
__device__ __forceinline__ void cull_fast_body(const OrbDev &D, const int l, const int r) {
    const Level &L = D.lv[l];
    __shared__ int s_n, s_mine, s_base, s_wc[4];
    __shared__ unsigned s_hist[256], s_bin, s_rem;
    const int rcap = fast_region_cap(L.w, L.h), tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid < 64) {
        const int c = lane < FAST_REGIONS ? min(D.hist[MAXLV * 256 + l * FAST_REGIONS + lane], rcap) : 0;
        int tot = c;
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) tot += __shfl_xor(tot, d);
        if (lane == r) s_mine = c;
        if (lane == 0) s_n = tot;
    }
    __syncthreads();
    const int n = s_n, mine = s_mine, keepN = 2 * L.nKeep;
    int thr = 0;
    if (keepN == 0) thr = 1 << 30;
    else if (n > keepN) {
        // largest score v with #{score >= v} >= keepN: prefix sums over the bins in DESCENDING score order
        const int *rh = D.hist + MAXLV * 256 + MAXLV * FAST_REGIONS + l * FAST_REGIONS * 256;
        unsigned acc = 0;
#pragma unroll 8
        for (int rr = 0; rr < FAST_REGIONS; rr++) acc += (unsigned) rh[rr * 256 + tid];
        s_hist[tid] = acc;
        __syncthreads();
        if (tid < 64) wave_find_bin([&](int b) { return s_hist[255 - b]; }, (unsigned) (keepN - 1), &s_bin, &s_rem);
        __syncthreads();
        thr = 255 - (int) s_bin;
    }
    const int o = L.candOff, src0 = o + r * rcap;
    // pass 1: how many of the slice survive; pass 2 (after the one atomic) writes them
    int cnt = 0;
    for (int i = tid; i < mine; i += 256) cnt += D.c1s[src0 + i] >= thr;
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) cnt += __shfl_xor(cnt, d);
    if (lane == 0) s_wc[wave] = cnt;
    __syncthreads();
    if (tid == 0) {
        const int tot = s_wc[0] + s_wc[1] + s_wc[2] + s_wc[3];
        s_base = tot ? atomicAdd(&D.n2[l], tot) : 0;
        if (r == 0) D.n1[l] = n;
    }
    __syncthreads();
    int carry = s_base;
    for (int b = 0; b < mine; b += 256) {
        const int i = b + tid;
        const bool k = i < mine && D.c1s[src0 + i] >= thr;
        const unsigned long long m = __ballot(k);
        __syncthreads();   // (s_wc of the previous round / of pass 1 has been read)
        if (lane == 0) s_wc[wave] = __popcll(m);
        __syncthreads();
        int before = 0, tot = 0;
#pragma unroll
        for (int w = 0; w < 4; w++) {
            const int c = s_wc[w];
            before += w < wave ? c : 0;
            tot += c;
        }
        if (k) {
            const int pos = o + carry + before + __popcll(m & ((1ull << lane) - 1ull));
            D.c2x[pos] = D.c1x[src0 + i];
            D.c2y[pos] = D.c1y[src0 + i];
        }
        carry += tot;
    }
}

__global__ void __launch_bounds__(256) k_cull_fast(OrbDev D) { cull_fast_body(D, blockIdx.y, blockIdx.x); }
__global__ void __launch_bounds__(256) k_cull_fast_b(const OrbItem *__restrict__ items, int count, int nlevels) {
    const AlvaXcdItem w = alva_xcd_item(count, nlevels * FAST_REGIONS);
    if (w.cam < count) cull_fast_body(items[w.cam].D, w.item / FAST_REGIONS, w.item % FAST_REGIONS);
}

// Harris response of every surviving candidate (orb.cpp:130-177): one wave each, the 49 block positions spread over the
// lanes.  a, b, c are INTEGER sums in the reference, so the cross-lane reduction order cannot change them.
__device__ __forceinline__ void harris_body(const OrbDev &D, const int l, const int bx, const int gx) {
    const Level &L = D.lv[l];
    const int lane = threadIdx.x & 63, n2 = D.n2[l], W = L.pitch;
    const uint8_t *img = D.pool + L.img;
    for (int i = bx * 4 + (threadIdx.x >> 6); i < n2; i += gx * 4) {
    const int x = D.c2x[L.candOff + i], y = D.c2y[L.candOff + i];
    int a = 0, b = 0, c = 0;
    if (lane < 49) {
        const uint8_t *p = img + (size_t) (y - 3 + lane / 7) * W + (x - 3 + lane % 7);
        const int Ix = (p[1] - p[-1]) * 2 + (p[-W + 1] - p[-W - 1]) + (p[W + 1] - p[W - 1]);
        const int Iy = (p[W] - p[-W]) * 2 + (p[W - 1] - p[-W - 1]) + (p[W + 1] - p[-W + 1]);
        a = Ix * Ix;
        b = Iy * Iy;
        c = Ix * Iy;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        a += __shfl_down(a, off);
        b += __shfl_down(b, off);
        c += __shfl_down(c, off);
    }
    if (lane == 0) {
        const float scale = 1.f / ((1 << 2) * 7 * 255.f);
        const float scale_sq_sq = scale * scale * scale * scale;
        const float fa = (float) a, fb = (float) b, fc = (float) c;
        D.c2r[L.candOff + i] = ((fa * fb - fc * fc) - (0.04f * (fa + fb)) * (fa + fb)) * scale_sq_sq;
    }
    }
}

__global__ void __launch_bounds__(256) k_harris(OrbDev D) { harris_body(D, blockIdx.y, blockIdx.x, gridDim.x); }
__global__ void __launch_bounds__(256) k_harris_b(const OrbItem *__restrict__ items, int count, int gx, int nlevels) {
    const AlvaXcdItem w = alva_xcd_item(count, gx * nlevels);
    if (w.cam < count) harris_body(items[w.cam].D, w.item / gx, w.item % gx, gx);
}

__device__ __forceinline__ unsigned f2key(float f) {  // order-preserving map float -> uint
    const unsigned u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

// cull by Harris for a level with more than HARRIS_RANK_CAP candidates: keep response >= the n_l-th largest, preserving the list's order
// (radix select on the float keys, passes over global memory)
__device__ __forceinline__ void cull_harris_big(const OrbDev &D, const int l) {
    const Level &L = D.lv[l];
    const int n = D.n2[l], keepN = L.nKeep, o = L.candOff;
    __shared__ unsigned s_hist[256];
    __shared__ unsigned s_prefix, s_k;
    unsigned thrKey = 0;
    if (keepN == 0) thrKey = 0xffffffffu;
    else if (n > keepN) {
        // the keepN-th largest = element of rank (n - keepN) in ascending order
        unsigned prefix = 0, mask = 0;
        unsigned k = (unsigned) (n - keepN);
        for (int shift = 24; shift >= 0; shift -= 8) {
            if (threadIdx.x < 256) s_hist[threadIdx.x] = 0;
            __syncthreads();
            for (int i = threadIdx.x; i < n; i += 1024) {
                const unsigned key = f2key(D.c2r[o + i]);
                if ((key & mask) == prefix) atomicAdd(&s_hist[(key >> shift) & 255u], 1u);
            }
            __syncthreads();
            if (threadIdx.x < 64) wave_find_bin([&](int b) { return s_hist[b]; }, k, &s_prefix, &s_k);
            __syncthreads();
            prefix |= s_prefix << shift;
            k = s_k;
            mask |= 0xffu << shift;
            __syncthreads();
        }
        thrKey = prefix;
    }
    const int m = compact_ordered(
        n, [&](int i) { return f2key(D.c2r[o + i]) >= thrKey; },
        [&](int i, int pos) {
            D.c3x[o + pos] = D.c2x[o + i];
            D.c3y[o + pos] = D.c2y[o + i];
            D.c3r[o + pos] = D.c2r[o + i];
        });
    if (threadIdx.x == 0) D.n3[l] = m;
    // Deterministic output order: row-major by position inside the level (positions are unique).  Rank sort: every element
    // counts the keys below its own from an LDS copy (broadcast reads, no barriers in the loop); m is ~n_l (a few hundred).
    constexpr int SORT_CAP = 4096;
    __shared__ unsigned s_key[SORT_CAP];
    if (m > SORT_CAP) return;  // pathological tie at the Harris cut: the set is still right, the order is as compacted
    __syncthreads();
    int ex[4], ey[4];
    float er[4];
#pragma unroll
    for (int q = 0; q < 4; q++) {
        const int i = threadIdx.x + q * 1024;
        if (i < m) {
            ex[q] = D.c3x[o + i];
            ey[q] = D.c3y[o + i];
            er[q] = D.c3r[o + i];
            s_key[i] = ((unsigned) ey[q] << 16) | (unsigned) ex[q];
        }
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 4; q++) {
        const int i = threadIdx.x + q * 1024;
        if (i < m) {
            const unsigned mine = ((unsigned) ey[q] << 16) | (unsigned) ex[q];
            int rank = 0;
            for (int j = 0; j < m; j++) rank += s_key[j] < mine;
            D.c3x[o + rank] = ex[q];
            D.c3y[o + rank] = ey[q];
            D.c3r[o + rank] = er[q];
        }
    }
}


// cull by Harris: keep response >= the n_l-th largest (ties kept), emit in row-major position order.  The level's candidates (<= 4096: four
// per thread) stay in registers: the radix select makes its four passes over them with an LDS histogram (no passes over global memory),
// and the output position of a survivor is (survivors in earlier rows) + (survivors of its row to its left) -- a per-row count, a scan
// over the rows, a slot per survivor, and a look at the handful of survivors that share its row: O(n) work for one workgroup, and no
// dependence on the order the candidates arrive in.  (Radix select over global memory + ordered compaction + an m^2 rank sort: 25.6 us at
// 1280x720; ranking all pairs in LDS instead: 205 us -- 3 M pairs are too many for ONE compute unit.)
constexpr int HARRIS_RANK_CAP = 4096, HARRIS_ROW_CAP = 2048;
__device__ __forceinline__ void cull_harris_body(const OrbDev &D, const int l) {
    const Level &L = D.lv[l];
    const int n = D.n2[l], keepN = L.nKeep, o = L.candOff;
    if (n > HARRIS_RANK_CAP || L.h > HARRIS_ROW_CAP) {
        cull_harris_big(D, l);
        return;
    }
    __shared__ unsigned s_hist[256], s_prefix, s_k;
    __shared__ int s_row[HARRIS_ROW_CAP], s_start[HARRIS_ROW_CAP], s_list[HARRIS_RANK_CAP], s_wsum[16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int ex[4] = {0, 0, 0, 0}, ey[4] = {0, 0, 0, 0};
    float er[4] = {0.f, 0.f, 0.f, 0.f};
    unsigned rk[4] = {0u, 0u, 0u, 0u};
    bool valid[4];
#pragma unroll
    for (int q = 0; q < 4; q++) {
        const int i = tid + q * 1024;
        valid[q] = i < n;
        if (valid[q]) {
            ex[q] = D.c2x[o + i];
            ey[q] = D.c2y[o + i];
            er[q] = D.c2r[o + i];
            rk[q] = f2key(er[q]);
        }
    }
    for (int y = tid; y < L.h; y += 1024) s_row[y] = 0;
    unsigned thrKey = 0;
    if (keepN == 0) thrKey = 0xffffffffu;
    else if (n > keepN) {
        // the keepN-th largest = element of rank (n - keepN) in ascending order
        unsigned prefix = 0, mask = 0;
        unsigned k = (unsigned) (n - keepN);
        for (int shift = 24; shift >= 0; shift -= 8) {
            if (tid < 256) s_hist[tid] = 0;
            __syncthreads();
#pragma unroll
            for (int q = 0; q < 4; q++)
                if (valid[q] && (rk[q] & mask) == prefix) atomicAdd(&s_hist[(rk[q] >> shift) & 255u], 1u);
            __syncthreads();
            if (tid < 64) wave_find_bin([&](int b) { return s_hist[b]; }, k, &s_prefix, &s_k);
            __syncthreads();
            prefix |= s_prefix << shift;
            k = s_k;
            mask |= 0xffu << shift;
        }
        thrKey = prefix;
    }
    __syncthreads();
    bool keep[4];
    int slot[4] = {0, 0, 0, 0};
#pragma unroll
    for (int q = 0; q < 4; q++) {
        keep[q] = valid[q] && rk[q] >= thrKey && (keepN != 0);
        if (keep[q]) slot[q] = atomicAdd(&s_row[ey[q]], 1);
    }
    __syncthreads();
    // exclusive scan of the per-row counts: two rows per thread, a wave scan, the 16 wave totals
    {
        const int r0 = 2 * tid, a = r0 < L.h ? s_row[r0] : 0, b = r0 + 1 < L.h ? s_row[r0 + 1] : 0;
        const int sum = a + b;
        int incl = sum;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int v = __shfl_up(incl, d);
            if (lane >= d) incl += v;
        }
        if (lane == 63) s_wsum[wave] = incl;
        __syncthreads();
        int before = 0;
#pragma unroll
        for (int w = 0; w < 16; w++) before += w < wave ? s_wsum[w] : 0;
        const int excl = before + incl - sum;
        if (r0 < L.h) s_start[r0] = excl;
        if (r0 + 1 < L.h) s_start[r0 + 1] = excl + a;
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 4; q++)
        if (keep[q]) s_list[s_start[ey[q]] + slot[q]] = ex[q];
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 4; q++) {
        if (keep[q]) {
            const int st = s_start[ey[q]], c = s_row[ey[q]];
            int pos = st;
            for (int j = 0; j < c; j++) pos += s_list[st + j] < ex[q];
            D.c3x[o + pos] = ex[q];
            D.c3y[o + pos] = ey[q];
            D.c3r[o + pos] = er[q];
        }
    }
    if (tid == 0) {
        int m = 0;
        for (int w = 0; w < 16; w++) m += s_wsum[w];
        D.n3[l] = m;
    }
}

__global__ void __launch_bounds__(1024) k_cull_harris(OrbDev D) { cull_harris_body(D, blockIdx.x); }
__global__ void __launch_bounds__(1024) k_cull_harris_b(const OrbItem *__restrict__ items, int count, int nlevels) {
    const AlvaXcdItem w = alva_xcd_item(count, nlevels);
    if (w.cam < count) cull_harris_body(items[w.cam].D, w.item);
}

__device__ __forceinline__ float fast_atan2f(float y, float x) {  // mathfuncs_core.simd.hpp:34-71
    const float s = (float) (180 / 3.1415926535897932384626433832795);
    const float p1 = 0.9997878412794807f * s, p3 = -0.3258083974640975f * s, p5 = 0.1555786518463281f * s, p7 = -0.04432655554792128f * s;
    const float ax = fabsf(x), ay = fabsf(y);
    const float eps = (float) 2.2204460492503131e-16;
    float a;
    if (ax >= ay) {
        const float c = ay / (ax + eps), c2 = c * c;
        a = (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
    } else {
        const float c = ax / (ay + eps), c2 = c * c;
        a = 90.f - (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
    }
    if (x < 0) a = 180.f - a;
    if (y < 0) a = 360.f - a;
    return a;
}

// IC angle + output record, one wave per kept keypoint (orb.cpp:181-215, :952-958)
__device__ __forceinline__ void angle_emit_body(const OrbDev &D, float *__restrict__ kp, int cap, int *__restrict__ total, const int bx, const int l) {
    const Level &L = D.lv[l];
    const int i = bx * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    int base = 0;
    for (int q = 0; q < l; q++) base += D.n3[q];
    if (bx == 0 && threadIdx.x == 0) {
        D.h_n3[l] = D.n3[l];
        if (l == D.nlevels - 1) *total = base + D.n3[l];
    }
    if (i >= D.n3[l]) return;
    const int x = D.c3x[L.candOff + i], y = D.c3y[L.candOff + i], W = L.pitch;
    const uint8_t *ctr = D.pool + L.img + (size_t) y * W + x;
    // rows v = -15..15 spread over lanes 0..30; integer sums => any order is exact
    int m01 = 0, m10 = 0;
    if (lane < 31) {
        const int v = lane - 15, av = v < 0 ? -v : v, d = av == 0 ? 15 : D.umax[av];
        // the row's 31 bytes u = -15..15 (+1) as eight independent unaligned dword loads -- one round trip to L2 instead of a
        // dependent load per pixel; the circular patch keeps |u| <= d (keypoints lie >= 31 px inside the level, so the full row exists)
        const uint8_t *rowp = ctr + v * W - 15;
        uint32_t dw[8];
#pragma unroll
        for (int k = 0; k < 8; k++) __builtin_memcpy(&dw[k], rowp + 4 * k, 4);
        int rs = 0, ms = 0;
#pragma unroll
        for (int k = 0; k < 31; k++) {
            const int u = k - 15;
            const int val = (u >= -d && u <= d) ? (int) ((dw[k >> 2] >> (8 * (k & 3))) & 0xffu) : 0;
            rs += val;
            ms += u * val;
        }
        m10 = ms;
        m01 = v * rs;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        m01 += __shfl_down(m01, off);
        m10 += __shfl_down(m10, off);
    }
    if (lane == 0 && base + i < cap) {
        float *o = kp + 6 * (size_t) (base + i);
        o[0] = (float) x * L.scale;
        o[1] = (float) y * L.scale;
        o[2] = 31 * L.scale;
        o[3] = fast_atan2f((float) m01, (float) m10);
        o[4] = D.c3r[L.candOff + i];
        o[5] = (float) l;
    }
}

__global__ void __launch_bounds__(256) k_angle_emit(OrbDev D, float *__restrict__ kp, int cap, int *__restrict__ total) { angle_emit_body(D, kp, cap, total, blockIdx.x, blockIdx.y); }
__global__ void __launch_bounds__(256) k_angle_emit_b(const OrbItem *__restrict__ items, int count, int gx, int nlevels) {
    // a level keeps a few hundred keypoints while the launch bound is >= 1024: a short grid that loops instead of ~130 k workgroups
    // (for 64 cameras) of which three quarters find nothing to do
    const AlvaXcdItem w = alva_xcd_item(count, gx * nlevels);
    if (w.cam >= count) return;
    const OrbItem &it = items[w.cam];
    const int l = w.item / gx, n3 = it.D.n3[l];
    for (int bx = w.item % gx; bx == 0 || bx * 4 < n3; bx += gx) angle_emit_body(it.D, it.kp, it.cap, it.total, bx, l);
}

// keypoints whose rotation could not be PROVEN to round like the host library's (see exact_sincos.hpp); expected to stay 0
__device__ int g_orb_ambiguous = 0;

__constant__ int8_t c_pattern_orb[1024] = {
#include "orb_pattern.inc"
};

// steered BRIEF of the ORB keypoints on their (blurred) pyramid level; 32 lanes per keypoint (orb.cpp:219-284)
__device__ __forceinline__ void brief_orb_body(const OrbDev &D, const float *__restrict__ kp, const int *__restrict__ total, int cap,
                                               uint8_t *__restrict__ desc, const int bx) {
    const int n = min(*total, cap);
    const int k = bx * 8 + threadIdx.x / 32, byte = threadIdx.x % 32;
    if (k >= n) return;
    const float *rec = kp + 6 * (size_t) k;
    const int l = (int) rec[5];
    const Level &L = D.lv[l];
    const float inv = 1.f / L.scale;
    const int cx = __float2int_rn(rec[0] * inv), cy = __float2int_rn(rec[1] * inv);
    float angle = rec[3];
    angle *= (float) (3.1415926535897932384626433832795 / 180.f);
    // (float) cos((double) angle), (float) sin((double) angle) of the host's C library, decided from a double-double evaluation (exact_sincos.hpp)
    float a, b;
    int amb = 0;
    alva_dd::sincos_float(angle, &a, &b, &amb);
    if (amb && byte == 0) atomicAdd(&g_orb_ambiguous, 1);
    const uint8_t *center = D.pool + L.blur + (size_t) cy * L.pitch + cx;
    const int8_t *pat = c_pattern_orb + byte * 32;
    unsigned val = 0;
#pragma unroll
    for (int t = 0; t < 8; t++) {
        const float x0 = (float) pat[4 * t], y0 = (float) pat[4 * t + 1], x1 = (float) pat[4 * t + 2], y1 = (float) pat[4 * t + 3];
        const int ix0 = __float2int_rn(x0 * a - y0 * b), iy0 = __float2int_rn(x0 * b + y0 * a);
        const int ix1 = __float2int_rn(x1 * a - y1 * b), iy1 = __float2int_rn(x1 * b + y1 * a);
        const int t0 = center[(ptrdiff_t) iy0 * L.pitch + ix0], t1 = center[(ptrdiff_t) iy1 * L.pitch + ix1];
        val |= (unsigned) (t0 < t1) << t;
    }
    desc[(size_t) k * 32 + byte] = (uint8_t) val;
}

__global__ void __launch_bounds__(256) k_brief_orb(OrbDev D, const float *__restrict__ kp, const int *__restrict__ total, int cap,
                                                   uint8_t *__restrict__ desc) {
    brief_orb_body(D, kp, total, cap, desc, blockIdx.x);
}
__global__ void __launch_bounds__(256) k_brief_orb_b(const OrbItem *__restrict__ items, int count, int per_cam) {
    const AlvaXcdItem w = alva_xcd_item(count, per_cam);
    if (w.cam >= count) return;
    const OrbItem &it = items[w.cam];
    brief_orb_body(it.D, it.kp, it.total, it.cap, it.desc, w.item);
}

__device__ __forceinline__ void copy_level0_body(const OrbDev &D, const uint8_t *__restrict__ src, size_t pitch, const int bx, const int by) {
    const Level &L = D.lv[0];
    const int x = bx * 64 + (threadIdx.x & 63), y = by * 4 + (threadIdx.x >> 6);
    if (x < L.w && y < L.h) D.pool[L.img + (size_t) y * L.pitch + x] = src[(size_t) y * pitch + x];
    // first launch of the chain: clear the FAST-score histograms and append counters here instead of a separate fill command, every
    // workgroup a slice (flat histograms | region counters | per-region histograms: 100 k ints)
    {
        const int gx = (L.w + 63) / 64, gy = (L.h + 3) / 4, nb = gx * gy, lb = by * gx + bx;
        const int total = MAXLV * 256 + MAXLV * FAST_REGIONS + D.nlevels * FAST_REGIONS * 256;
        for (int k = lb * 256 + (int) threadIdx.x; k < total; k += nb * 256) D.hist[k] = 0;
        if (lb == 0 && threadIdx.x < 2 * MAXLV) D.n1[threadIdx.x] = 0;   // n1 (the unfused FAST path's counts) | n2 (k_cull_fast appends through it)
    }
}

__global__ void k_copy_level0(OrbDev D, const uint8_t *__restrict__ src, size_t pitch) { copy_level0_body(D, src, pitch, blockIdx.x, blockIdx.y); }
__global__ void k_copy_level0_b(const OrbItem *__restrict__ items, int count, int gx, int gy) {
    const AlvaXcdItem w = alva_xcd_item(count, gx * gy);
    if (w.cam >= count) return;
    const OrbItem &it = items[w.cam];
    copy_level0_body(it.D, it.gray, it.gray_pitch, w.item % gx, w.item / gx);
}

int cv_round_f(float v) { return (int) lrintf(v); }

}  // namespace

struct alva_orb {
    int device = 0;
    OrbDev D{};
    void *d_block = nullptr;
    int maxRows = 0, maxTiles = 0, candTotal = 0;
    int *d_total = nullptr;
    bool fast_only = false;
    void *d_blur_batch = nullptr;   // the levels' BlurBatch in device memory (the pool's addresses never change)
    int blurTiles = 0;
    bool chain_resize = false;      // ALVA_ORB_PYRAMID=chain: one k_resize per level (the fused launch's check)
    // the levels' 7x7 blur needs the pyramid only, the descriptors need it last: it runs on a stream of its own beside FAST -> Harris -> angles
    hipStream_t side = nullptr;
    hipEvent_t pyr_done = nullptr, blur_done = nullptr;
};

static int orb_build(alva_ctx *ctx, int width, int height, int nfeatures, float scale_factor, int nlevels, int fast_threshold, int border,
                     alva_orb **out) {
    ALVA_ARG(ctx && out && width >= 16 && height >= 16 && nlevels >= 1 && nlevels <= MAXLV && nfeatures >= 0);
    ALVA_HIP(hipSetDevice(ctx->device));
    alva_orb *o = new alva_orb();
    o->device = ctx->device;
    OrbDev &D = o->D;
    D.nlevels = nlevels;
    D.threshold = std::min(std::max(fast_threshold, 0), 255);
    {
        const char *e_ = getenv("ALVA_ORB_PYRAMID");
        o->chain_resize = e_ && !strcmp(e_, "chain");
    }
    const double scaleFactor = (double) scale_factor;
    // features per level (orb.cpp:799-813)
    {
        const float factor = (float) (1.0 / scaleFactor);
        float nd = (float) nfeatures * (1 - factor) / (1 - (float) std::pow((double) factor, (double) nlevels));
        int sum = 0;
        for (int l = 0; l < nlevels - 1; l++) {
            D.lv[l].nKeep = cv_round_f(nd);
            sum += D.lv[l].nKeep;
            nd *= factor;
        }
        D.lv[nlevels - 1].nKeep = std::max(nfeatures - sum, 0);
    }
    // umax (orb.cpp:819-834)
    {
        const int hp = 15;
        const int vmax = (int) std::floor(hp * std::sqrt(2.f) / 2 + 1), vmin = (int) std::ceil(hp * std::sqrt(2.f) / 2);
        for (int v = 0; v <= vmax; v++) D.umax[v] = (int) std::lrint(std::sqrt((double) hp * hp - v * v));
        for (int v = hp, v0 = 0; v >= vmin; --v) {
            while (D.umax[v0] == D.umax[v0 + 1]) ++v0;
            D.umax[v] = v0;
            ++v0;
        }
    }
    // level geometry + resize taps (orb.cpp:1041-1058; resize.cpp:733-770)
    std::vector<int> tapOfs, tapCoef;
    size_t pool = 0;
    int rows = 0, cands = 0;
    for (int l = 0; l < nlevels; l++) {
        Level &L = D.lv[l];
        L.scale = (float) std::pow(scaleFactor, (double) l);
        const float inv = 1.0f / L.scale;
        L.w = cv_round_f((float) width * inv);
        L.h = cv_round_f((float) height * inv);
        if (L.w < 8 || L.h < 8) {
            delete o;
            alva_set_error("alva_orb_create: level %d would be %dx%d", l, L.w, L.h);
            return ALVA_ERR_ARG;
        }
        L.pitch = (L.w + 63) / 64 * 64;
        L.border = border;
        L.rowOff = rows;
        rows += L.h;
        L.candOff = cands;
        // NMS survivors: at most a quarter of the pixels.  The fused FAST + NMS kernel appends per REGION (FAST_REGIONS counters per level, a
        // tile's region = tile % FAST_REGIONS): room for every tile of a region to deliver its maximum of FT_W * FT_H / 4
        L.candCap = std::max(L.w * L.h / 4 + 64, FAST_REGIONS * fast_region_cap(L.w, L.h));
        cands += L.candCap;
        o->maxRows = std::max(o->maxRows, L.h);
        o->maxTiles = std::max(o->maxTiles, ((L.w + FT_W - 1) / FT_W) * ((L.h + FT_H - 1) / FT_H));
        L.tabOff = (int) tapOfs.size();
        if (l > 0) {
            const Level &S = D.lv[l - 1];
            for (int pass = 0; pass < 2; pass++) {
                const int dn = pass ? L.h : L.w, sn = pass ? S.h : S.w;
                const double inv_scale = (double) dn / sn;
                const double scale = 1.0 / inv_scale;
                for (int d = 0; d < dn; d++) {
                    const double fval = scale * ((double) d + 0.5) - 0.5;
                    const int ival = (int) std::floor(fval);
                    if (ival >= 0 && sn > 1) {
                        if (ival < sn - 1) {
                            tapOfs.push_back(ival);
                            tapCoef.push_back((int) std::lrint((fval - (double) ival) * 256.0));
                        } else {
                            tapOfs.push_back(sn - 1);
                            tapCoef.push_back(-2);
                        }
                    } else {
                        tapOfs.push_back(0);
                        tapCoef.push_back(-1);
                    }
                }
            }
        } else {
            tapOfs.resize(tapOfs.size() + L.w + L.h, 0);
            tapCoef.resize(tapCoef.size() + L.w + L.h, 0);
        }
    }
    // the fused pyramid launch's plan (pyr_column_body): the deepest level F whose columns fit the kernel's LDS buffers (steeper pyramids
    // than 1.2 leave their deep levels to k_resize), and per tile column / row of level F its region and owned range in every level below
    std::vector<int> spans;
    D.pyrFused = 0;
    for (int F = nlevels - 1; F >= 1 && !o->chain_resize; F--) {
        const Level &T = D.lv[F];
        const int gxt = alva_divup(T.w, PT_W), gyt = alva_divup(T.h, PT_H);
        std::vector<int> sp((size_t) (gxt + gyt) * F * 4);
        int maxw[MAXLV] = {0}, maxh[MAXLV] = {0};
        for (int axis = 0; axis < 2; axis++) {
            const int nt = axis ? gyt : gxt, ts = axis ? PT_H : PT_W;
            auto dim = [&](int k) { return axis ? D.lv[k].h : D.lv[k].w; };
            auto taps = [&](int k) { return tapOfs.data() + D.lv[k].tabOff + (axis ? D.lv[k].w : 0); };   // level k's taps into level k - 1
            // first source index of every tile in every level below (the plain tap chain): the ownership boundaries
            std::vector<int> first((size_t) (nt + 1) * F);
            for (int t = 0; t < nt; t++) {
                int a0 = t * ts;
                for (int k = F; k >= 1; k--) {
                    a0 = taps(k)[a0];
                    first[(size_t) t * F + (k - 1)] = t == 0 ? 0 : a0;
                }
            }
            for (int k = 0; k < F; k++) first[(size_t) nt * F + k] = dim(k);
            for (int t = 0; t < nt; t++) {
                int a0 = t * ts, an = std::min(ts, dim(F) - a0);
                for (int k = F; k >= 1; k--) {
                    const int sn = dim(k - 1);
                    int lo = taps(k)[a0], hi = std::min(taps(k)[a0 + an - 1] + 1, sn - 1);
                    const int o0 = first[(size_t) t * F + (k - 1)], o1 = std::max(first[(size_t) (t + 1) * F + (k - 1)], o0);
                    if (o1 > o0) {   // the region contains what the column owns
                        lo = std::min(lo, o0);
                        hi = std::max(hi, o1 - 1);
                    }
                    a0 = lo;
                    an = hi - lo + 1;
                    int *e = sp.data() + ((axis ? (size_t) gxt * F : 0) + (size_t) t * F + (size_t) (k - 1)) * 4;
                    e[0] = a0;
                    e[1] = an;
                    e[2] = o0;
                    e[3] = o1;
                    int &m = (axis ? maxh : maxw)[k - 1];
                    m = std::max(m, an);
                }
            }
        }
        bool fits = true;
        int ntaps = PT_W + PT_H;
        for (int k = 0; k < F; k++) {
            fits = fits && ((maxw[k] + 3) & ~3) * maxh[k] <= (k & 1 ? PYR_BUF_B : PYR_BUF_A) && maxw[k] < 65536 && maxh[k] < 65536;
            if (k >= 1) ntaps += maxw[k] + maxh[k];
        }
        if (!fits || ntaps > PYR_TAPS) continue;
        D.pyrFused = F;
        spans = sp;
        break;
    }
    if (spans.empty()) spans.push_back(0);
    for (int l = 0; l < nlevels; l++) {
        Level &L = D.lv[l];
        const size_t bytes = (size_t) L.pitch * L.h;
        L.img = pool; pool += bytes + 256;
        L.score = pool; pool += bytes + 256;
        L.blur = pool; pool += bytes + 256;
    }
    o->candTotal = cands;
    // one allocation, carved
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t r = off; off += (bytes + 255) / 256 * 256; return r; };
    const size_t o_pool = take(pool), o_rowCnt = take((size_t) rows * 4), o_rowStart = take((size_t) rows * 4), o_n = take(3 * MAXLV * 4),
                 o_hist = take((size_t) FAST_HIST_INTS * 4), o_c1 = take((size_t) cands * 12), o_c2 = take((size_t) cands * 12),
                 o_c3 = take((size_t) cands * 12), o_tap = take(tapOfs.size() * 8), o_total = take(64), o_span = take(spans.size() * 4),
                 o_blur = take(alva_blur7_batch_size());
    hipError_t e = hipMalloc(&o->d_block, off);
    if (e != hipSuccess) {
        delete o;
        alva_set_error("alva_orb_create: hipMalloc(%zu): %s", off, hipGetErrorString(e));
        return ALVA_ERR_NOMEM;
    }
    uint8_t *b = (uint8_t *) o->d_block;
    D.pool = b + o_pool;
    D.rowCnt = (int *) (b + o_rowCnt);
    D.rowStart = (int *) (b + o_rowStart);
    D.n1 = (int *) (b + o_n);
    D.n2 = D.n1 + MAXLV;
    D.n3 = D.n2 + MAXLV;
    D.hist = (int *) (b + o_hist);
    D.c1x = (int *) (b + o_c1); D.c1y = D.c1x + cands; D.c1s = D.c1y + cands;
    D.c2x = (int *) (b + o_c2); D.c2y = D.c2x + cands; D.c2r = (float *) (D.c2y + cands);
    D.c3x = (int *) (b + o_c3); D.c3y = D.c3x + cands; D.c3r = (float *) (D.c3y + cands);
    int *d_tap = (int *) (b + o_tap);
    D.tapOfs = d_tap;
    D.tapCoef = d_tap + tapOfs.size();
    o->d_total = (int *) (b + o_total);
    D.pyrSpan = (const int *) (b + o_span);
    o->d_blur_batch = b + o_blur;
    e = hipHostMalloc((void **) &D.h_n3, MAXLV * sizeof(int), hipHostMallocDefault);
    if (e != hipSuccess) {
        (void) hipFree(o->d_block);
        delete o;
        alva_set_error("alva_orb_create: hipHostMalloc: %s", hipGetErrorString(e));
        return ALVA_ERR_NOMEM;
    }
    memset(D.h_n3, 0, MAXLV * sizeof(int));
    ALVA_HIP(hipMemcpyAsync(d_tap, tapOfs.data(), tapOfs.size() * 4, hipMemcpyHostToDevice, ctx->stream));
    ALVA_HIP(hipMemcpyAsync(d_tap + tapOfs.size(), tapCoef.data(), tapCoef.size() * 4, hipMemcpyHostToDevice, ctx->stream));
    ALVA_HIP(hipMemcpyAsync(b + o_span, spans.data(), spans.size() * 4, hipMemcpyHostToDevice, ctx->stream));
    std::vector<uint8_t> blurBatch(alva_blur7_batch_size());
    {
        const uint8_t *bs[MAXLV];
        uint8_t *bd[MAXLV];
        int bw[MAXLV], bh[MAXLV], bp[MAXLV];
        for (int l = 0; l < nlevels; l++) {
            const Level &L = D.lv[l];
            bs[l] = D.pool + L.img;
            bd[l] = D.pool + L.blur;
            bw[l] = L.w;
            bh[l] = L.h;
            bp[l] = L.pitch;
        }
        o->blurTiles = alva_blur7_batch_fill(blurBatch.data(), nlevels, bs, bd, bw, bh, bp);
        if (o->blurTiles < 0) {
            (void) hipStreamSynchronize(ctx->stream);
            alva_orb_destroy(o);
            alva_set_error("alva_orb_create: image pool rows are not 4-byte aligned");
            return ALVA_ERR_STATE;
        }
        ALVA_HIP(hipMemcpyAsync(o->d_blur_batch, blurBatch.data(), blurBatch.size(), hipMemcpyHostToDevice, ctx->stream));
    }
    ALVA_HIP(hipStreamSynchronize(ctx->stream));
    *out = o;
    return ALVA_OK;
}

extern "C" int alva_orb_create(alva_ctx *ctx, int width, int height, int nfeatures, float scale_factor, int nlevels, int fast_threshold,
                               alva_orb **out) {
    return orb_build(ctx, width, height, nfeatures, scale_factor, nlevels, fast_threshold, 31, out);
}

extern "C" void alva_orb_destroy(alva_orb *orb) {
    if (!orb) return;
    (void) hipSetDevice(orb->device);
    if (orb->d_block) (void) hipFree(orb->d_block);
    if (orb->D.h_n3) (void) hipHostFree(orb->D.h_n3);
    if (orb->side) (void) hipStreamDestroy(orb->side);
    if (orb->pyr_done) (void) hipEventDestroy(orb->pyr_done);
    if (orb->blur_done) (void) hipEventDestroy(orb->blur_done);
    delete orb;
}

// levels -> FAST -> NMS survivors (row-major per level) with counts in n1
static int run_fast_stages(alva_ctx *ctx, alva_orb *o, const uint8_t *d_gray, size_t gray_pitch, bool fused, hipEvent_t pyramid_done = nullptr) {
    OrbDev &D = o->D;
    hipStream_t st = ctx->stream;
    const Level &L0 = D.lv[0];
    int chained = 1;   // first level that still needs its own k_resize
    if (D.pyrFused == 0) {
        hipLaunchKernelGGL(k_copy_level0, dim3(alva_divup(L0.w, 64), alva_divup(L0.h, 4)), dim3(256), 0, st, D, d_gray, gray_pitch);
    } else {
        const Level &T = D.lv[D.pyrFused];
        const int nt = alva_divup(T.w, PT_W) * alva_divup(T.h, PT_H);
        hipLaunchKernelGGL(k_pyramid, dim3((unsigned) (8 * alva_divup(nt, 8))), dim3(256), 0, st, D, d_gray, gray_pitch, alva_kstamp_buffer());   // (stamps: ALVA_KSTAMPS=1, tools/pyr_stamps.py)
        chained = D.pyrFused + 1;
    }
    for (int l = chained; l < D.nlevels; l++)
        hipLaunchKernelGGL(k_resize, dim3(alva_divup(D.lv[l].w, 64), alva_divup(D.lv[l].h, 4)), dim3(256), 0, st, D, l);
    if (pyramid_done) ALVA_HIP(hipEventRecord(pyramid_done, st));
    if (fused) {
        // ORB: candidate order is irrelevant downstream (k_cull_harris re-sorts by position), so FAST + NMS is one launch
        int total = 0;
        for (int l = 0; l < D.nlevels; l++)
            total += 8 * alva_divup(alva_divup(D.lv[l].w, FT_W) * alva_divup(D.lv[l].h, FT_H), 8);
        hipLaunchKernelGGL(k_fast_nms, dim3((unsigned) total), dim3(256), 0, st, D);
        ALVA_LAUNCH_CHECK();
        return ALVA_OK;
    }
    hipLaunchKernelGGL(k_fast_score, dim3(o->maxTiles, D.nlevels), dim3(256), 0, st, D);
    hipLaunchKernelGGL(k_fast_rows<false>, dim3(alva_divup(o->maxRows, 4), D.nlevels), dim3(256), 0, st, D);
    hipLaunchKernelGGL(k_scan_rows, dim3(D.nlevels), dim3(1024), 0, st, D);
    hipLaunchKernelGGL(k_fast_rows<true>, dim3(alva_divup(o->maxRows, 4), D.nlevels), dim3(256), 0, st, D);
    ALVA_LAUNCH_CHECK();
    return ALVA_OK;
}

extern "C" int alva_orb_collect(alva_ctx *ctx, alva_orb *orb, int *h_count);

// device-resident total of the last detect_and_compute (internal: lets the driver chain the matcher without a host round trip)
extern "C" const int *alva_orb_device_count(const alva_orb *orb) { return orb ? orb->d_total : nullptr; }

// one level of the detector's pyramid as the last alva_orb_detect_and_compute left it (which = 0) or its 7x7 blur (which = 1): w x h bytes
extern "C" int alva_orb_debug_level(alva_ctx *ctx, alva_orb *orb, int level, int which, uint8_t *d_out, size_t out_pitch, int *w, int *h) {
    ALVA_ARG(ctx && orb && level >= 0 && level < orb->D.nlevels && (which == 0 || which == 1));
    const Level &L = orb->D.lv[level];
    if (w) *w = L.w;
    if (h) *h = L.h;
    if (!d_out) return ALVA_OK;
    ALVA_ARG(out_pitch >= (size_t) L.w);
    ALVA_HIP(hipMemcpy2DAsync(d_out, out_pitch, orb->D.pool + (which ? L.blur : L.img), (size_t) L.pitch, (size_t) L.w, (size_t) L.h, hipMemcpyDeviceToDevice,
                              ctx->stream));
    ALVA_HIP(hipStreamSynchronize(ctx->stream));
    return ALVA_OK;
}

extern "C" int alva_orb_ambiguous_rotations(int *h_count, int reset) {
    ALVA_ARG(h_count);
    ALVA_HIP(hipMemcpyFromSymbol(h_count, HIP_SYMBOL(g_orb_ambiguous), sizeof(int)));
    if (reset) {
        const int zero = 0;
        ALVA_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_orb_ambiguous), &zero, sizeof(int)));
    }
    return ALVA_OK;
}

extern "C" int alva_orb_detect_and_compute(alva_ctx *ctx, alva_orb *orb, const uint8_t *d_gray, size_t gray_pitch, float *d_kp,
                                           uint8_t *d_desc, int cap, int *h_count) {
    ALVA_ARG(ctx && orb && d_gray && d_kp && cap >= 0 && gray_pitch >= (size_t) orb->D.lv[0].w);
    OrbDev &D = orb->D;
    hipStream_t st = ctx->stream;
    // The blur beside the detector (round 6, OFF by default): k_blur7_multi (12 us at 1280x720) depends on the pyramid alone and only the LAST
    // kernel reads it, so it can go on a side stream between two events.  Measured: -3.5 us per call in a process of its own
    // (tools/orb_kernels.py: 117.2 vs 120.7 us), but +80 us inside bench.py's process, where dozens of streams from the earlier sections
    // share the hardware queues and every cross-stream dependency becomes a barrier packet + a signal round trip.  ALVA_ORB_SIDE_BLUR=1
    // turns it on.
    static const bool side_blur = getenv("ALVA_ORB_SIDE_BLUR") != nullptr;
    const bool overlap = side_blur && d_desc != nullptr && !g_alva_prof_on;   // (the in-library profiler brackets launches on ctx->stream)
    if (overlap && !orb->side) {
        ALVA_HIP(hipStreamCreateWithFlags(&orb->side, hipStreamNonBlocking));
        ALVA_HIP(hipEventCreateWithFlags(&orb->pyr_done, hipEventDisableTiming));
        ALVA_HIP(hipEventCreateWithFlags(&orb->blur_done, hipEventDisableTiming));
    }
    int rc = run_fast_stages(ctx, orb, d_gray, gray_pitch, true, overlap ? orb->pyr_done : nullptr);
    if (rc) return rc;
    if (overlap) {
        ALVA_HIP(hipStreamWaitEvent(orb->side, orb->pyr_done, 0));
        rc = alva_blur7_multi_launch(ctx, orb->d_blur_batch, 1, D.nlevels, orb->blurTiles, orb->side);
        if (rc) return rc;
        ALVA_HIP(hipEventRecord(orb->blur_done, orb->side));
    }
    hipLaunchKernelGGL(k_cull_fast, dim3(FAST_REGIONS, D.nlevels), dim3(256), 0, st, D);
    hipLaunchKernelGGL(k_harris, dim3(256, D.nlevels), dim3(256), 0, st, D);   // wave-strided over the level's candidates
    hipLaunchKernelGGL(k_cull_harris, dim3(D.nlevels), dim3(1024), 0, st, D);
    int maxKeep = 0;
    for (int l = 0; l < D.nlevels; l++) maxKeep = std::max(maxKeep, std::min(D.lv[l].candCap, std::max(4 * D.lv[l].nKeep + 64, 1024)));
    // ties at the Harris cut are rare (float responses): 4 n_l + 64 waves per level is a generous bound; if a level ever had
    // more survivors than that the extra ones would be dropped silently, so the bound is checked below.
    hipLaunchKernelGGL(k_angle_emit, dim3(alva_divup(maxKeep, 4), D.nlevels), dim3(256), 0, st, D, d_kp, cap, orb->d_total);
    ALVA_LAUNCH_CHECK();
    if (d_desc) {
        // the levels' 7x7 blur: a 1-D grid over the real tiles of all levels, four pixels per thread (describe.hip; the (maxTiles, level)
        // grid of k_blur7_batch was 60 % empty workgroups at 1280x720 and byte-granular: 31 us)
        if (overlap) ALVA_HIP(hipStreamWaitEvent(st, orb->blur_done, 0));
        else {
            rc = alva_blur7_multi_launch(ctx, orb->d_blur_batch, 1, D.nlevels, orb->blurTiles);
            if (rc) return rc;
        }
        int nmax = 0;
        for (int l = 0; l < D.nlevels; l++) nmax += std::min(D.lv[l].candCap, std::max(4 * D.lv[l].nKeep + 64, 1024));
        nmax = std::min(nmax, cap);
        if (nmax > 0) hipLaunchKernelGGL(k_brief_orb, dim3(alva_divup(nmax, 8)), dim3(256), 0, st, D, (const float *) d_kp, (const int *) orb->d_total, cap, d_desc);
        ALVA_LAUNCH_CHECK();
    }
    if (!h_count) return ALVA_OK;  // enqueue only; alva_orb_collect() fetches the count later
    return alva_orb_collect(ctx, orb, h_count);
}

extern "C" int alva_orb_collect(alva_ctx *ctx, alva_orb *orb, int *h_count) {
    ALVA_ARG(ctx && orb && h_count);
    OrbDev &D = orb->D;
    hipStream_t st = ctx->stream;
    int n3[MAXLV], total = 0;
    ALVA_HIP(hipStreamSynchronize(st));
    memcpy(n3, D.h_n3, sizeof(n3));
    for (int l = 0; l < D.nlevels; l++) {
        if (n3[l] > std::min(D.lv[l].candCap, std::max(4 * D.lv[l].nKeep + 64, 1024))) {
            alva_set_error("alva_orb_detect_and_compute: level %d kept %d keypoints, above the launch bound", l, n3[l]);
            return ALVA_ERR_STATE;
        }
        total += n3[l];
    }
    *h_count = total;
    return ALVA_OK;
}

// ---- cv::ORB::detectAndCompute of `count` cameras, one set of launches (camera from alva_xcd_item) -----------------------------------
// Every camera has its own detector object (image pool, candidate lists, counters); the objects must share one geometry.  The
// kernels are the single-camera kernels' bodies, so every camera's keypoints and descriptors are those of its own
// alva_orb_detect_and_compute.  Enqueue-only; alva_orb_collect_batch() waits and returns the counts.
extern "C" int alva_orb_detect_and_compute_batch(alva_ctx *ctx, alva_orb *const *orbs, int count, const uint8_t *const *d_gray,
                                                 size_t gray_pitch, float *const *d_kp, uint8_t *const *d_desc, int cap) {
    ALVA_ARG(ctx && orbs && count > 0 && count <= 65535 && d_gray && d_kp && d_desc && cap >= 0 && orbs[0]);
    const OrbDev &D0 = orbs[0]->D;
    ALVA_ARG(gray_pitch >= (size_t) D0.lv[0].w);
    hipStream_t st = ctx->stream;
    const size_t blur_sz = alva_blur7_batch_size();
    const size_t off_blur = ((size_t) count * sizeof(OrbItem) + 255) / 256 * 256, bytes = off_blur + (size_t) count * blur_sz;
    std::vector<uint8_t> host(bytes);
    int blurTiles = 0;
    for (int c = 0; c < count; c++) {
        const alva_orb *o = orbs[c];
        ALVA_ARG(o && !o->fast_only && d_gray[c] && d_kp[c] && d_desc[c] && o->D.nlevels == D0.nlevels && o->maxTiles == orbs[0]->maxTiles && o->D.pyrFused == D0.pyrFused);
        for (int l = 0; l < D0.nlevels; l++)
            ALVA_ARG(o->D.lv[l].w == D0.lv[l].w && o->D.lv[l].h == D0.lv[l].h && o->D.lv[l].nKeep == D0.lv[l].nKeep && o->D.lv[l].candCap == D0.lv[l].candCap);
        OrbItem it{};
        it.D = o->D;
        it.gray = d_gray[c];
        it.gray_pitch = gray_pitch;
        it.kp = d_kp[c];
        it.desc = d_desc[c];
        it.total = o->d_total;
        it.cap = cap;
        memcpy(host.data() + (size_t) c * sizeof(OrbItem), &it, sizeof(it));
        const uint8_t *bs[MAXLV];
        uint8_t *bd[MAXLV];
        int bw[MAXLV], bh[MAXLV], bp[MAXLV];
        for (int l = 0; l < D0.nlevels; l++) {
            const Level &L = o->D.lv[l];
            bs[l] = o->D.pool + L.img;
            bd[l] = o->D.pool + L.blur;
            bw[l] = L.w;
            bh[l] = L.h;
            bp[l] = L.pitch;
        }
        blurTiles = alva_blur7_batch_fill(host.data() + off_blur + (size_t) c * blur_sz, D0.nlevels, bs, bd, bw, bh, bp);
        if (blurTiles < 0) {
            alva_set_error("alva_orb_detect_and_compute_batch: image pool rows are not 4-byte aligned");
            return ALVA_ERR_STATE;
        }
    }
    uint8_t *dev = nullptr;
    int rc = alva_ctx_scratch(ctx, 1, bytes, (void **) &dev);
    if (rc) return rc;
    ALVA_HIP(hipMemcpyAsync(dev, host.data(), bytes, hipMemcpyHostToDevice, st));
    const OrbItem *items = (const OrbItem *) dev;
    const Level &L0 = D0.lv[0];
    // every launch below: 1-D grid, camera -> XCD affinity (alva_xcd_item, common.hpp)
    int chained = 1;   // first level that still needs its own k_resize_b
    if (D0.pyrFused == 0) {
        const int gx = alva_divup(L0.w, 64), gy = alva_divup(L0.h, 4);
        hipLaunchKernelGGL(k_copy_level0_b, dim3(alva_xcd_grid(count, gx * gy)), dim3(256), 0, st, items, count, gx, gy);
    } else {
        // levels 0 .. pyrFused of every camera in one launch (pyr_column_body; the objects share one geometry, hence one plan shape)
        const Level &T = D0.lv[D0.pyrFused];
        const int nt = alva_divup(T.w, PT_W) * alva_divup(T.h, PT_H);
        hipLaunchKernelGGL(k_pyramid_b, dim3(alva_xcd_grid(count, nt)), dim3(256), 0, st, items, count, nt);
        chained = D0.pyrFused + 1;
    }
    for (int l = chained; l < D0.nlevels; l++) {
        const int gx = alva_divup(D0.lv[l].w, 256), gy = alva_divup(D0.lv[l].h, 4);
        hipLaunchKernelGGL(k_resize_b, dim3(alva_xcd_grid(count, gx * gy)), dim3(256), 0, st, items, l, count, gx, gy);
    }
    int fastTiles = 0;
    for (int l = 0; l < D0.nlevels; l++) fastTiles += alva_divup(D0.lv[l].w, FT_W) * alva_divup(D0.lv[l].h, FT_H);
    hipLaunchKernelGGL(k_fast_nms_b, dim3(alva_xcd_grid(count, fastTiles)), dim3(256), 0, st, items, count, fastTiles);
    hipLaunchKernelGGL(k_cull_fast_b, dim3(alva_xcd_grid(count, D0.nlevels * FAST_REGIONS)), dim3(256), 0, st, items, count, D0.nlevels);
    hipLaunchKernelGGL(k_harris_b, dim3(alva_xcd_grid(count, 64 * D0.nlevels)), dim3(256), 0, st, items, count, 64, D0.nlevels);   // wave-strided loop, as k_harris
    hipLaunchKernelGGL(k_cull_harris_b, dim3(alva_xcd_grid(count, D0.nlevels)), dim3(1024), 0, st, items, count, D0.nlevels);
    int maxKeep = 0, nmax = 0;
    for (int l = 0; l < D0.nlevels; l++) {
        const int bound = std::min(D0.lv[l].candCap, std::max(4 * D0.lv[l].nKeep + 64, 1024));
        maxKeep = std::max(maxKeep, bound);
        nmax += bound;
    }
    {
        const int gx = std::min(64, alva_divup(maxKeep, 4));
        hipLaunchKernelGGL(k_angle_emit_b, dim3(alva_xcd_grid(count, gx * D0.nlevels)), dim3(256), 0, st, items, count, gx, D0.nlevels);
    }
    ALVA_LAUNCH_CHECK();
    rc = alva_blur7_multi_launch(ctx, dev + off_blur, count, D0.nlevels, blurTiles);
    if (rc) return rc;
    nmax = std::min(nmax, cap);
    if (nmax > 0) hipLaunchKernelGGL(k_brief_orb_b, dim3(alva_xcd_grid(count, alva_divup(nmax, 8))), dim3(256), 0, st, items, count, alva_divup(nmax, 8));
    ALVA_LAUNCH_CHECK();
    return ALVA_OK;
}

extern "C" int alva_orb_collect_batch(alva_ctx *ctx, alva_orb *const *orbs, int count, int *h_counts) {
    ALVA_ARG(ctx && orbs && count > 0 && h_counts);
    ALVA_HIP(hipStreamSynchronize(ctx->stream));
    for (int c = 0; c < count; c++) {
        const OrbDev &D = orbs[c]->D;
        int total = 0;
        for (int l = 0; l < D.nlevels; l++) {
            const int n3 = D.h_n3[l];
            if (n3 > std::min(D.lv[l].candCap, std::max(4 * D.lv[l].nKeep + 64, 1024))) {
                alva_set_error("alva_orb_detect_and_compute_batch: camera %d level %d kept %d keypoints, above the launch bound", c, l, n3);
                return ALVA_ERR_STATE;
            }
            total += n3;
        }
        h_counts[c] = total;
    }
    return ALVA_OK;
}

// plain FAST on one image (no border cull): a one-level "orb" object cached in the context would avoid the allocation; the
// function is a parity/utility entry point, so it simply builds and tears down its buffers.
extern "C" int alva_fast(alva_ctx *ctx, const uint8_t *d_gray, size_t gray_pitch, int width, int height, int threshold, int *d_xy,
                         int *d_score, int cap, int *h_count) {
    ALVA_ARG(ctx && d_gray && d_xy && d_score && h_count && cap >= 0);
    alva_orb *o = nullptr;
    int rc = orb_build(ctx, width, height, 0, 1.0f, 1, threshold, 0, &o);
    if (rc) return rc;
    rc = run_fast_stages(ctx, o, d_gray, gray_pitch, false);
    int n = 0;
    if (!rc) {
        hipError_t e = hipMemcpyAsync(&n, o->D.n1, 4, hipMemcpyDeviceToHost, ctx->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
        if (e == hipSuccess && n > 0) {
            const int m = std::min(n, cap);
            // interleave x,y
            std::vector<int> xs((size_t) m), ys((size_t) m), xy((size_t) 2 * m);
            e = hipMemcpy(xs.data(), o->D.c1x, (size_t) m * 4, hipMemcpyDeviceToHost);
            if (e == hipSuccess) e = hipMemcpy(ys.data(), o->D.c1y, (size_t) m * 4, hipMemcpyDeviceToHost);
            for (int i = 0; i < m; i++) {
                xy[2 * (size_t) i] = xs[(size_t) i];
                xy[2 * (size_t) i + 1] = ys[(size_t) i];
            }
            if (e == hipSuccess) e = hipMemcpy(d_xy, xy.data(), (size_t) 2 * m * 4, hipMemcpyHostToDevice);
            if (e == hipSuccess) e = hipMemcpy(d_score, o->D.c1s, (size_t) m * 4, hipMemcpyDeviceToDevice);
        }
        if (e != hipSuccess) {
            alva_set_error("alva_fast: %s", hipGetErrorString(e));
            rc = ALVA_ERR_HIP;
        }
    }
    alva_orb_destroy(o);
    *h_count = n;
    return rc;
}
