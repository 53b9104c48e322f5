// a2 + a3: RGBA -> gray and the LK pyramid (padded gray levels + Scharr derivative levels).
//
// Arithmetic restated from the reference's vendored OpenCV (all integer, bit-exact):
//   gray     Y = (9798 R + 19235 G + 3735 B + 16384) >> 15       imgproc/src/color_rgb.simd.hpp:646-664
//   pyrDown  separable [1 4 6 4 1], (sum + 128) >> 8, REFLECT_101 imgproc/src/pyramids.cpp:746-
//   Scharr   Ix = d/dx of (3,10,3)-smoothed rows, Iy likewise     video/src/lkpyramid.cpp:70-151
//   borders  gray REFLECT_101 by `win`, deriv constant 0          video/src/lkpyramid.cpp:726-822
//
// HBM layout: every level lives in its own padded 2-D buffer whose interior column 0 is 16-byte
// aligned (so the interior is written with 4/16-byte stores); the REFLECT_101 border of a gray
// level is written by the same threads that produce the mirrored interior pixels, so a level is
// complete after ONE launch.  Launch plan for an L-level pyramid: 1 + L launches
//   [gray L0 (+border)] , [scharr(l) | pyrDown(l -> l+1) (+border)] for l = 0..L-2 , [scharr(L-1)]
#include "common.hpp"
#include "multi_kernel.hpp"

namespace {

__device__ __forceinline__ uint32_t gray_of(uint32_t rgba) {
    uint32_t r = rgba & 0xff, g = (rgba >> 8) & 0xff, b = (rgba >> 16) & 0xff;
    return (9798u * r + 19235u * g + 3735u * b + 16384u) >> 15;
}

// Writes v at (x,y) of a padded gray level and at every REFLECT_101 mirror image of (x,y) inside
// the `win` border.  Interior store is done by the caller (vectorised); this handles mirrors only.
__device__ __forceinline__ void store_mirrors(uint8_t *g, size_t pitch, int w, int h, int win, int x, int y, uint8_t v) {
    int xs[3], ys[3];
    int nx = 0, ny = 0;
    xs[nx++] = x;
    if (x >= 1 && x <= win) xs[nx++] = -x;
    if (x <= w - 2 && x >= w - 1 - win) xs[nx++] = 2 * (w - 1) - x;
    ys[ny++] = y;
    if (y >= 1 && y <= win) ys[ny++] = -y;
    if (y <= h - 2 && y >= h - 1 - win) ys[ny++] = 2 * (h - 1) - y;
    if (nx == 1 && ny == 1) return;
    for (int j = 0; j < ny; j++)
        for (int i = 0; i < nx; i++)
            if (i | j) g[(ptrdiff_t) ys[j] * (ptrdiff_t) pitch + xs[i]] = v;
}

// Level 0 from RGBA (SRC_RGBA) or from a gray image: 4 pixels per thread.
template<bool SRC_RGBA>
__device__ __forceinline__ void level0_body(const uint8_t *__restrict__ src, size_t src_pitch, int w, int h, int win,
                                            uint8_t *__restrict__ dst, size_t dst_pitch, uint8_t *__restrict__ gray_out,
                                            size_t gray_out_pitch, const int bx, const int by) {
    int x4 = (bx * 64 + threadIdx.x) * 4;
    int y = by * 4 + threadIdx.y;
    if (x4 >= w || y >= h) return;
    uint32_t packed;
    if (SRC_RGBA) {
        const uint4 p = *reinterpret_cast<const uint4 *>(src + (size_t) y * src_pitch + (size_t) x4 * 4);
        packed = gray_of(p.x) | (gray_of(p.y) << 8) | (gray_of(p.z) << 16) | (gray_of(p.w) << 24);
    } else {
        packed = *reinterpret_cast<const uint32_t *>(src + (size_t) y * src_pitch + x4);
    }
    *reinterpret_cast<uint32_t *>(dst + (size_t) y * dst_pitch + x4) = packed;
    if (gray_out) *reinterpret_cast<uint32_t *>(gray_out + (size_t) y * gray_out_pitch + x4) = packed;
    bool edge_y = (y <= win) || (y >= h - 1 - win);
    bool edge_x = (x4 <= win) || (x4 + 3 >= w - 1 - win);
    if (edge_x || edge_y) {
#pragma unroll
        for (int k = 0; k < 4; k++) store_mirrors(dst, dst_pitch, w, h, win, x4 + k, y, (uint8_t) (packed >> (8 * k)));
    }
}
template<bool SRC_RGBA>
__global__ void __launch_bounds__(256) k_level0(const uint8_t *__restrict__ src, size_t src_pitch, int w, int h, int win,
                                                uint8_t *__restrict__ dst, size_t dst_pitch,
                                                uint8_t *__restrict__ gray_out, size_t gray_out_pitch) {
    level0_body<SRC_RGBA>(src, src_pitch, w, h, win, dst, dst_pitch, gray_out, gray_out_pitch, blockIdx.x, blockIdx.y);
}
// several sessions' frames in one launch (lane.hpp): the 2-D grid of the single launch, row-major in bx
struct Level0Args {
    const uint8_t *src;
    size_t src_pitch;
    int w, h, win, gx0;
    uint8_t *dst;
    size_t dst_pitch;
    uint8_t *gray_out;
    size_t gray_out_pitch;
};
ALVA_MULTI_KERNEL(MK_LEVEL0, k_level0_multi, Level0Args, dim3(64, 4), 256,
                  level0_body<true>(A.src, A.src_pitch, A.w, A.h, A.win, A.dst, A.dst_pitch, A.gray_out, A.gray_out_pitch, bx % A.gx0, bx / A.gx0));
// The same for B cameras in one launch (camera -> XCD affinity, alva_xcd_item): one frame is 1.2 MB and every launch of this file is bound by
// launch latency, not by HBM; B frames per launch is what lets the same code run at memory speed.
struct Level0Item {
    const uint8_t *src;
    uint8_t *dst, *gray_out;
};
__global__ void __launch_bounds__(256) k_level0_batch(const Level0Item *__restrict__ items, size_t src_pitch, int w, int h, int win,
                                                      size_t dst_pitch, size_t gray_out_pitch, int count, int gx, int gy) {
    const AlvaXcdItem wi = alva_xcd_item(count, gx * gy);
    if (wi.cam >= count) return;
    const Level0Item it = items[wi.cam];
    level0_body<true>(it.src, src_pitch, w, h, win, it.dst, dst_pitch, it.gray_out, gray_out_pitch, wi.item % gx, wi.item / gx);
}

struct StageArgs {
    // Scharr part: level l
    const uint8_t *g;  // interior (0,0) of padded gray level l
    size_t g_pitch;
    int w, h;
    int16_t *d;  // interior (0,0) of deriv level l
    size_t d_pitch;
    int scharr_bx, scharr_blocks;  // blocks per row of tiles, total scharr blocks
    // pyrDown part: level l -> l+1 (dw == 0: none)
    uint8_t *ng;
    size_t ng_pitch;
    int dw, dh, win;
    int down_bx;
};

// Scharr pair of the four pixels x4 .. x4+3 of row y from rows y-1 .. y+1, bytes x4-4 .. x4+7, as three dwords per row
__device__ __forceinline__ void scharr_quad(const uint32_t (&r)[3][3], int16_t *d, size_t d_pitch, int x4, int y, int w) {
    int t0[6], t1[6];
#pragma unroll
    for (int c = 0; c < 6; c++) {
        // column x4-1+c
        int p0, p1, p2;
        if (c == 0) {
            p0 = r[0][0] >> 24; p1 = r[1][0] >> 24; p2 = r[2][0] >> 24;
        } else if (c == 5) {
            p0 = r[0][2] & 0xff; p1 = r[1][2] & 0xff; p2 = r[2][2] & 0xff;
        } else {
            int sh = 8 * (c - 1);
            p0 = (r[0][1] >> sh) & 0xff; p1 = (r[1][1] >> sh) & 0xff; p2 = (r[2][1] >> sh) & 0xff;
        }
        t0[c] = (p0 + p2) * 3 + p1 * 10;
        t1[c] = p2 - p0;
    }
    short out[8];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        out[2 * k] = (short) (t0[k + 2] - t0[k]);
        out[2 * k + 1] = (short) ((t1[k + 2] + t1[k]) * 3 + t1[k + 1] * 10);
    }
    int16_t *drow = reinterpret_cast<int16_t *>(reinterpret_cast<uint8_t *>(d) + (size_t) y * d_pitch) + 2 * x4;
    if (x4 + 3 < w) {
        uint4 v;
        v.x = (uint16_t) out[0] | ((uint32_t) (uint16_t) out[1] << 16);
        v.y = (uint16_t) out[2] | ((uint32_t) (uint16_t) out[3] << 16);
        v.z = (uint16_t) out[4] | ((uint32_t) (uint16_t) out[5] << 16);
        v.w = (uint16_t) out[6] | ((uint32_t) (uint16_t) out[7] << 16);
        *reinterpret_cast<uint4 *>(drow) = v;
    } else {
        for (int k = 0; k < 4 && x4 + k < w; k++) {
            drow[2 * k] = out[2 * k];
            drow[2 * k + 1] = out[2 * k + 1];
        }
    }
}

__device__ __forceinline__ void scharr_tile(const StageArgs &a, int bid) {
    int bx = bid % a.scharr_bx, by = bid / a.scharr_bx;
    int x4 = (bx * 64 + threadIdx.x) * 4;
    int y = by * 4 + threadIdx.y;
    if (x4 >= a.w || y >= a.h) return;
    // rows y-1..y+1, bytes x4-4 .. x4+7 via three aligned u32 loads per row (the padding is
    // REFLECT_101, which is exactly the border rule of ScharrDerivInvoker, lkpyramid.cpp:83-121)
    uint32_t r[3][3];
#pragma unroll
    for (int j = 0; j < 3; j++) {
        const uint8_t *row = a.g + (ptrdiff_t) (y - 1 + j) * (ptrdiff_t) a.g_pitch + x4;
        r[j][0] = *reinterpret_cast<const uint32_t *>(row - 4);
        r[j][1] = *reinterpret_cast<const uint32_t *>(row);
        r[j][2] = *reinterpret_cast<const uint32_t *>(row + 4);
    }
    scharr_quad(r, a.d, a.d_pitch, x4, y, a.w);
}

// pyrDown of level l into level l+1: FOUR output pixels per thread.  Their 5 x 5 binomial windows span input columns
// 2x-2 .. 2x+8 (x a multiple of 4), i.e. the 16 bytes from 2x-4 on: four aligned dword loads per input row instead of 25 byte
// loads per output pixel.  Out-of-image taps read the REFLECT_101 padding (identical to borderInterpolate on the level size,
// pyramids.cpp:760-775, because win >= 2); the four bytes beyond 2x+8 are padding or interior too (pitch is a multiple of 64
// and the window reaches at most 2 (dw - 1) + 11 <= w + 12 <= w + win + 3).
__device__ __forceinline__ void pyrdown_tile(const StageArgs &a, int bid) {
    const int bx = bid % a.down_bx, by = bid / a.down_bx;
    const int x = (bx * 64 + threadIdx.x) * 4;
    const int y = by * 4 + threadIdx.y;
    if (x >= a.dw || y >= a.dh) return;
    int acc[4] = {0, 0, 0, 0};
#pragma unroll
    for (int j = 0; j < 5; j++) {
        const uint8_t *row = a.g + (ptrdiff_t) (2 * y - 2 + j) * (ptrdiff_t) a.g_pitch + (2 * x - 4);
        const uint32_t *rp = reinterpret_cast<const uint32_t *>(row);  // columns 2x-4 .. 2x+11; 2x - 4 is a multiple of 4 (not of 16)
        const uint32_t w[4] = {rp[0], rp[1], rp[2], rp[3]};
        auto px = [&](int c) -> int { return (int) ((w[c >> 2] >> (8 * (c & 3))) & 0xffu); };  // byte c of the 16, column 2x - 4 + c
        const int wj = (j == 0 || j == 4) ? 1 : (j == 2 ? 6 : 4);
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int c0 = 2 + 2 * k;  // window of output x + k starts at column 2 (x + k) - 2 = (2x - 4) + 2 + 2k
            const int s = px(c0) + px(c0 + 4) + 4 * (px(c0 + 1) + px(c0 + 3)) + 6 * px(c0 + 2);
            acc[k] += wj * s;
        }
    }
    uint8_t out[4];
#pragma unroll
    for (int k = 0; k < 4; k++) out[k] = (uint8_t) ((acc[k] + 128) >> 8);
    uint8_t *dst = a.ng + (size_t) y * a.ng_pitch + x;
    if (x + 3 < a.dw) {
        *reinterpret_cast<uint32_t *>(dst) = (uint32_t) out[0] | ((uint32_t) out[1] << 8) | ((uint32_t) out[2] << 16) | ((uint32_t) out[3] << 24);
    } else {
        for (int k = 0; k < 4 && x + k < a.dw; k++) dst[k] = out[k];
    }
    for (int k = 0; k < 4 && x + k < a.dw; k++) store_mirrors(a.ng, a.ng_pitch, a.dw, a.dh, a.win, x + k, y, out[k]);
}

// ---- everything behind level 0 in ONE launch ---------------------------------------------------------------------------------------
// The stage kernels above are a chain of four dependent launches (level l+1 is read by stage l+1 only after stage l wrote it): at one
// 640x480 frame each is ~5 us of kernel for ~0.5 MB, i.e. four launch latencies in front of the tracker.  Here every workgroup that needs
// a level computes it ITSELF from level 0, in LDS: the pyramid is a 5x5 filter per level, so a 16 x 16 tile of level L (+ 1 px for Scharr)
// has a (2^L 17 + ...) ~ 39 / 81 / 165 pixel footprint in level 0 -- 28 KB of LDS at most, a few thousand taps per workgroup.  Roles by
// workgroup index: Scharr of level 0 (scharr_tile, as before) | for L = 1 .. 3: tile of level L from level 0 through the levels between
// (none of them read from memory), its interior + REFLECT_101 border written once, Scharr of level L from the same LDS tile.
// Integer arithmetic and border rule are the stage kernels' (pyrDown: borderInterpolate REFLECT_101 on the INPUT level's size,
// pyramids.cpp:760-775; Scharr: the REFLECT_101 padding, lkpyramid.cpp:83-121): an out-of-image tile entry holds the value of its
// mirror image, which is what the padding of the stored level holds.  Results are bit-identical (tests/test_gpu_image.py).
// Output tile side per target level: 16 x 16 for levels 1 and 2, 8 x 8 for level 3 -- its footprint in level 0 would be 165 x 165
// (28 KB of LDS, 650 LDS reads per thread for the level-1 tile alone: the launch's critical path, and the LDS bill of every workgroup
// of the launch); with 8 x 8 it is 101 x 101.  Buffer sides: footprints (t + 2) -> 2 (n - 1) + 5 per level down.
__host__ __device__ constexpr int rest_tile(int L) { return L == 3 ? 8 : 16; }
constexpr int R3 = 18;                   // target tile + halo: 18 (levels 1, 2) or 10 (level 3)
constexpr int R2 = 23;                   // level 2 inside a level-3 workgroup: 2 * 9 + 5
constexpr int R1 = 49;                   // level 1: 2 * 22 + 5 (level-3 workgroup) > 39 = 2 * 17 + 5 (level-2 workgroup)
constexpr int R0 = 101;                  // level 0: 2 * 48 + 5 (level-3 workgroup) > 81 (level 2) > 39 (level 1)
constexpr int R0P = (R0 + 3 + 3) / 4 * 4;   // level-0 LDS pitch (dword staging: up to 3 bytes of alignment slack)
struct RestLevel {
    uint8_t *g;
    size_t g_pitch;
    int16_t *d;
    size_t d_pitch;
    int w, h;
};
struct RestArgs {
    StageArgs s0;          // Scharr of level 0
    RestLevel lv[4];
    int nlevels, win;
    int first[4], bx[4], ntiles[4];   // deep tiles of level L: first workgroup index, tiles per row, number of tiles
    int ts[4];                        // ... and their side (<= rest_tile(L): the LDS buffers are sized for that)
};
struct Span {
    int lo, hi;            // inclusive, unmirrored coordinates
};
__device__ __forceinline__ int mirror101(int c, int n) { return c < 0 ? -c : (c >= n ? 2 * (n - 1) - c : c); }
// the same as an index that is safe to dereference whatever the size (a level narrower than the reach of the mirror: clamped)
__device__ __forceinline__ int mirror_in(int c, int n) { return min(max(mirror101(c, n), 0), n - 1); }
// the range of the level below that the entries of `s` (mirrored into [0, n)) read: 2 m - 2 .. 2 m + 2
__device__ __forceinline__ Span below(Span s, int n) { return Span{2 * max(s.lo, 0) - 2, 2 * min(s.hi, n - 1) + 2}; }

// dst tile (span dx, dy at a level of size n_x x n_y) from the src tile (span sx, sy) of the level below: pyrDown at the mirrored coordinate
__device__ __forceinline__ void down_tile(const uint8_t *src, int spitch, Span sx, Span sy, uint8_t *dst, int dpitch, Span dx, Span dy, int nx,
                                          int ny, int tid) {
    const int dw = dx.hi - dx.lo + 1, dh = dy.hi - dy.lo + 1;
    for (int e = tid; e < dw * dh; e += 256) {
        const int j = e / dw, i = e - j * dw;
        const int mx = mirror101(dx.lo + i, nx), my = mirror101(dy.lo + j, ny);
        const uint8_t *p = src + (2 * my - 2 - sy.lo) * spitch + (2 * mx - 2 - sx.lo);
        int acc = 0;
#pragma unroll
        for (int r = 0; r < 5; r++) {
            const uint8_t *q = p + r * spitch;
            const int srow = q[0] + q[4] + 4 * (q[1] + q[3]) + 6 * q[2];
            acc += ((r == 0 || r == 4) ? 1 : (r == 2 ? 6 : 4)) * srow;
        }
        dst[j * dpitch + i] = (uint8_t) ((acc + 128) >> 8);
    }
}

// FROM_RGBA (k_pyr_all): the level-0 footprint is converted from the RGBA frame on the way into LDS -- level 0 need not exist in memory, so
// the tile does not wait for the launch that writes it.  Out-of-image entries: the REFLECT_101 mirror image, which is what the padding of
// the stored level holds (store_mirrors).
template<bool FROM_RGBA>
__device__ __forceinline__ void deep_tile(const RestArgs &A, int L, int bid, const uint8_t *__restrict__ rgba = nullptr, size_t rgba_pitch = 0) {
    __shared__ __attribute__((aligned(16))) uint8_t b0[R0 * R0P];
    __shared__ uint8_t b1[R1 * R1], b2[R2 * R2], b3[R3 * R3];
    const int tid = threadIdx.y * 64 + threadIdx.x;
    const RestLevel &T = A.lv[L];
    const int ts = A.ts[L];
    const int ox = (bid % A.bx[L]) * ts, oy = (bid / A.bx[L]) * ts;
    // spans per level, from the target down to level 0.  In LDS, written by one thread: as per-thread arrays indexed with the run-time
    // level they lived in scratch memory -- 80 B per thread, ~6 MB of write traffic per launch in the PMC counters against 1.75 MB of
    // stored levels + derivatives.
    __shared__ Span sx[4], sy[4];
    if (tid == 0) {
        sx[L] = Span{ox - 1, min(ox + ts, T.w)};   // one pixel of halo for Scharr; beyond the level's last column + 1 nothing is needed
        sy[L] = Span{oy - 1, min(oy + ts, T.h)};
        for (int l = L; l >= 1; l--) {
            sx[l - 1] = below(sx[l], A.lv[l].w);
            sy[l - 1] = below(sy[l], A.lv[l].h);
        }
    }
    __syncthreads();
    // ---- level 0 footprint from memory, aligned dwords (columns >= -4: inside the REFLECT_101 padding of `win` >= 3 ... 9 pixels)
    const RestLevel &Z = A.lv[0];
    const int xa = (sx[0].lo & ~3);   // floor to a multiple of 4 (also for negative lo: two's complement)
    const int ndw = (sx[0].hi - xa) / 4 + 1, nrow = sy[0].hi - sy[0].lo + 1;
    for (int e = tid; e < ndw * nrow; e += 256) {
        const int r = e / ndw, c = e - r * ndw;
        uint32_t v;
        if (FROM_RGBA) {
            const int yy = mirror_in(sy[0].lo + r, Z.h), x0 = xa + 4 * c;
            const uint8_t *rowp = rgba + (size_t) yy * rgba_pitch;
            if (x0 >= 0 && x0 + 3 < Z.w) {
                const uint4 p = *reinterpret_cast<const uint4 *>(rowp + 4 * (size_t) x0);
                v = gray_of(p.x) | (gray_of(p.y) << 8) | (gray_of(p.z) << 16) | (gray_of(p.w) << 24);
            } else {
                v = 0;
#pragma unroll
                for (int k = 0; k < 4; k++)
                    v |= gray_of(*reinterpret_cast<const uint32_t *>(rowp + 4 * (size_t) mirror_in(x0 + k, Z.w))) << (8 * k);
            }
        } else {
            v = *reinterpret_cast<const uint32_t *>(Z.g + (ptrdiff_t) (sy[0].lo + r) * (ptrdiff_t) Z.g_pitch + xa + 4 * c);
        }
        *reinterpret_cast<uint32_t *>(b0 + r * R0P + 4 * c) = v;
    }
    __syncthreads();
    const Span s0x{xa, sx[0].hi};   // the staged columns start at xa
    // levels 1 .. L in LDS: intermediate levels in their own buffers, the TARGET level always in the small buffer b3
    Span px = s0x, py = sy[0];
    for (int l = 1; l <= L; l++) {
        const uint8_t *src = l == 1 ? b0 : (l == 2 ? b1 : b2);
        const int sp = l == 1 ? R0P : (l == 2 ? R1 : R2);
        uint8_t *dst = l == L ? b3 : (l == 1 ? b1 : b2);
        const int dp = l == L ? R3 : (l == 1 ? R1 : R2);
        down_tile(src, sp, px, py, dst, dp, sx[l], sy[l], A.lv[l].w, A.lv[l].h, tid);
        __syncthreads();
        px = sx[l];
        py = sy[l];
    }
    // ---- this workgroup's 16 x 16 pixels of level L: the stored level (+ border mirrors) and its Scharr derivatives
    const int tx = tid % ts, ty = tid / ts;
    const int x = ox + tx, y = oy + ty;
    if (ty >= ts || x >= T.w || y >= T.h) return;
    const uint8_t *c = b3 + (ty + 1) * R3 + (tx + 1);   // entry of (x, y): spans start at ox - 1, oy - 1
    const uint8_t v = c[0];
    T.g[(size_t) y * T.g_pitch + x] = v;
    store_mirrors(T.g, T.g_pitch, T.w, T.h, A.win, x, y, v);
    int t0[3], t1[3];
#pragma unroll
    for (int k = 0; k < 3; k++) {
        const int p0 = c[-R3 + k - 1], p1 = c[k - 1], p2 = c[R3 + k - 1];
        t0[k] = (p0 + p2) * 3 + p1 * 10;
        t1[k] = p2 - p0;
    }
    const short ix = (short) (t0[2] - t0[0]), iy = (short) ((t1[2] + t1[0]) * 3 + t1[1] * 10);
    int16_t *drow = reinterpret_cast<int16_t *>(reinterpret_cast<uint8_t *>(T.d) + (size_t) y * T.d_pitch) + 2 * x;
    *reinterpret_cast<uint32_t *>(drow) = (uint32_t) (uint16_t) ix | ((uint32_t) (uint16_t) iy << 16);
}

__device__ __forceinline__ void pyr_rest_body(const RestArgs &A, const int bid) {
    if (bid < A.s0.scharr_blocks) {
        scharr_tile(A.s0, bid);
        return;
    }
    int L = 1;
    while (L + 1 < A.nlevels && bid >= A.first[L + 1]) L++;
    // XCD-aware order inside a level (workgroup b runs on XCD b % 8, each with its own L2): every XCD gets one CONTIGUOUS eighth of the
    // level's tiles (row-major), so neighbouring tiles -- whose level-0 footprints share cache lines -- meet in one L2 instead of pulling
    // the same lines through eight (PMC: 9.3 MB fetched per launch with the plain order against ~2.1 MB algorithmic)
    const int n = A.ntiles[L];
    const int lb = bid - A.first[L], per = (n + 7) / 8;
    const int tile = (lb & 7) * per + (lb >> 3);
    if (tile >= n) return;   // (whole workgroup: no barrier is skipped by a part of it)
    deep_tile<false>(A, L, tile);
}
__global__ void __launch_bounds__(256) k_pyr_rest(RestArgs A) { pyr_rest_body(A, (int) blockIdx.x); }
ALVA_MULTI_KERNEL(MK_PYR_REST, k_pyr_rest_multi, RestArgs, dim3(64, 4), 256, pyr_rest_body(A, bx));

// ---- the whole pyramid in ONE launch (round 6) -------------------------------------------------------------------------------------------
// k_level0 -> k_pyr_rest is a dependent pair only because the deep tiles read level 0 from memory.  Here they convert their footprint from
// the RGBA frame themselves (deep_tile<true>: one 16-B load per four pixels instead of one dword of gray -- the frame is read ~7 times, out
// of L2), and the level-0 role does gray + padded level 0 + the un-padded copy + Scharr of level 0 from its own pixels: a 64 x 16 tile with
// one pixel of halo in LDS (the halo from the frame as well, REFLECT_101 at the image's edge = the padding ScharrDerivInvoker reads).  No
// role reads anything another role writes => one launch, one launch latency in front of the tracker instead of two (k_level0 6.6 us +
// k_pyr_rest 11.9 us + the gap between them, per frame).  Same integer arithmetic, same bytes out (tests/test_gpu_image.py).
// Deep tiles come FIRST in the grid (deepest level first: they are the launch's critical path), the level-0 tiles behind them.
// Deep tiles are SMALLER here than in k_pyr_rest (16 / 12 / 4 instead of 16 / 16 / 8): this launch's duration is the slowest workgroup's, and
// a level-3 tile of 8 x 8 walks a 101 x 101 footprint (10 conversions + 235 LDS taps per thread for its level 1 alone); 4 x 4 needs 69 x 69.
// Four times the tiles, twice the redundant arithmetic, half the critical path -- while the grid is a few workgroups per compute unit.
// Measured (tools/pyr_times.py, event-timed launch, us): 640 x 480: 16/16/8 15.7, 16/8/4 12.0, 16/12/4 11.7, 16/8/3 11.5, 8/4/4 13.3;
// 1280 x 720 (3.4 k workgroups: throughput counts too): 16/16/8 16.9, 16/12/6 16.5, 16/12/4 18.2, 8/8/4 21.2.
static int env_tile(const char *name, int dflt, int most) {   // (A/B: ALVA_PYR_T1 / _T2 / _T3)
    const char *v = getenv(name);
    const int x = v ? atoi(v) : dflt;
    return x >= 2 && x <= most ? x : dflt;
}
static int all_tile(int L, size_t pixels) {
    static const int t[4] = {0, env_tile("ALVA_PYR_T1", 16, rest_tile(1)), env_tile("ALVA_PYR_T2", 12, rest_tile(2)), env_tile("ALVA_PYR_T3", 0, rest_tile(3))};
    if (L == 3 && !t[3]) return pixels <= 500000 ? 4 : 6;
    return t[L];
}
constexpr int L0_TW = 64, L0_TH = 16, L0_P = L0_TW + 8;   // LDS row: 4 bytes in front (the left halo is byte 3), 64 pixels, 4 behind
struct AllArgs {
    RestArgs R;            // (s0 unused; first[] counts from this kernel's workgroup 0, deepest level first)
    const uint8_t *rgba;
    size_t rgba_pitch;
    uint8_t *gray_out;
    size_t gray_out_pitch;
    int l0_first, l0_bx;   // level-0 tiles: first workgroup, tiles per row
};
__device__ __forceinline__ uint32_t gray4(const uint8_t *p) {
    const uint4 q = *reinterpret_cast<const uint4 *>(p);
    return gray_of(q.x) | (gray_of(q.y) << 8) | (gray_of(q.z) << 16) | (gray_of(q.w) << 24);
}
__device__ __forceinline__ void level0_tile(const AllArgs &A, const int tile) {
    __shared__ __attribute__((aligned(16))) uint8_t t[(L0_TH + 2) * L0_P];
    const int tid = threadIdx.y * 64 + threadIdx.x;
    const RestLevel &Z = A.R.lv[0];
    const int w = Z.w, h = Z.h, win = A.R.win;
    const int ox = (tile % A.l0_bx) * L0_TW, oy = (tile / A.l0_bx) * L0_TH;
    const int tx = tid & 15, ty = tid >> 4;
    const int x4 = ox + 4 * tx, y = oy + ty;
    const bool in = x4 < w && y < h;
    if (in) {
        const uint32_t packed = gray4(A.rgba + (size_t) y * A.rgba_pitch + (size_t) x4 * 4);
        *reinterpret_cast<uint32_t *>(t + (ty + 1) * L0_P + 4 + 4 * tx) = packed;
        *reinterpret_cast<uint32_t *>(Z.g + (size_t) y * Z.g_pitch + x4) = packed;
        if (A.gray_out) *reinterpret_cast<uint32_t *>(A.gray_out + (size_t) y * A.gray_out_pitch + x4) = packed;
        const bool edge_y = (y <= win) || (y >= h - 1 - win);
        const bool edge_x = (x4 <= win) || (x4 + 3 >= w - 1 - win);
        if (edge_x || edge_y) {
#pragma unroll
            for (int k = 0; k < 4; k++) store_mirrors(Z.g, Z.g_pitch, w, h, win, x4 + k, y, (uint8_t) (packed >> (8 * k)));
        }
    }
    // halo: row -1 and the row behind the tile's last (rb), column -1 and the column behind its last (cr) -- at the image's edge the mirror image
    const int cr = min(L0_TW, w - ox), rb = min(L0_TH, h - oy);
    if (tid < 32) {
        const int q = tid & 15, r = tid < 16 ? -1 : rb, xx = ox + 4 * q;
        if (xx < w)
            *reinterpret_cast<uint32_t *>(t + (r + 1) * L0_P + 4 + 4 * q) = gray4(A.rgba + (size_t) mirror_in(oy + r, h) * A.rgba_pitch + (size_t) xx * 4);
    } else if (tid < 32 + 2 * (L0_TH + 2)) {
        const int k = tid - 32, side = k >= L0_TH + 2, r = (side ? k - (L0_TH + 2) : k) - 1;
        if (r <= rb) {
            const int c = side ? cr : -1;
            const uint32_t px = *reinterpret_cast<const uint32_t *>(A.rgba + (size_t) mirror_in(oy + r, h) * A.rgba_pitch + (size_t) mirror_in(ox + c, w) * 4);
            t[(r + 1) * L0_P + 4 + c] = (uint8_t) gray_of(px);
        }
    }
    __syncthreads();
    if (!in) return;
    uint32_t r[3][3];
#pragma unroll
    for (int j = 0; j < 3; j++) {
        const uint32_t *row = reinterpret_cast<const uint32_t *>(t + (ty + j) * L0_P + 4 * tx);   // bytes x4-4 .. x4+7 of row y-1+j
        r[j][0] = row[0];
        r[j][1] = row[1];
        r[j][2] = row[2];
    }
    scharr_quad(r, Z.d, Z.d_pitch, x4, y, w);
}
__global__ void __launch_bounds__(256) k_pyr_all(AllArgs A) {
    const int bid = (int) blockIdx.x;
    if (bid >= A.l0_first) {
        level0_tile(A, bid - A.l0_first);
        return;
    }
    int L = 1;
    while (bid < A.R.first[L]) L++;   // first[] falls with the level: the deepest level owns workgroup 0
    const int n = A.R.ntiles[L];      // XCD-aware order inside a level, as in pyr_rest_body
    const int lb = bid - A.R.first[L], per = (n + 7) / 8;
    const int tile = (lb & 7) * per + (lb >> 3);
    if (tile >= n) return;
    deep_tile<true>(A.R, L, tile, A.rgba, A.rgba_pitch);
}

__global__ void __launch_bounds__(256) k_pyr_stage(StageArgs a) {
    int bid = blockIdx.x;
    if (bid < a.scharr_blocks) scharr_tile(a, bid);
    else pyrdown_tile(a, bid - a.scharr_blocks);
}
__global__ void __launch_bounds__(256) k_pyr_stage_batch(const StageArgs *__restrict__ args, int count, int per_cam) {
    const AlvaXcdItem wi = alva_xcd_item(count, per_cam);
    if (wi.cam >= count) return;
    const StageArgs a = args[wi.cam];
    int bid = wi.item;
    if (bid < a.scharr_blocks) scharr_tile(a, bid);
    else pyrdown_tile(a, bid - a.scharr_blocks);
}

__global__ void __launch_bounds__(256) k_rgba2gray(const uint8_t *__restrict__ src, size_t src_pitch, int w, int h,
                                                   uint8_t *__restrict__ dst, size_t dst_pitch) {
    int x4 = (blockIdx.x * 64 + threadIdx.x) * 4;
    int y = blockIdx.y * 4 + threadIdx.y;
    if (x4 >= w || y >= h) return;
    const uint4 p = *reinterpret_cast<const uint4 *>(src + (size_t) y * src_pitch + (size_t) x4 * 4);
    uint32_t packed = gray_of(p.x) | (gray_of(p.y) << 8) | (gray_of(p.z) << 16) | (gray_of(p.w) << 24);
    *reinterpret_cast<uint32_t *>(dst + (size_t) y * dst_pitch + x4) = packed;
}

static size_t round_up(size_t a, size_t b) { return (a + b - 1) / b * b; }

}  // namespace

extern "C" int alva_rgba2gray(alva_ctx *ctx, const uint8_t *d_rgba, size_t rgba_pitch, int width, int height,
                              uint8_t *d_gray, size_t gray_pitch) {
    ALVA_ARG(ctx && d_rgba && d_gray);
    ALVA_ARG(width > 0 && height > 0 && width % 4 == 0);
    ALVA_ARG(rgba_pitch % 16 == 0 && rgba_pitch >= (size_t) width * 4 && ((uintptr_t) d_rgba % 16) == 0);
    ALVA_ARG(gray_pitch % 4 == 0 && gray_pitch >= (size_t) width && ((uintptr_t) d_gray % 4) == 0);
    dim3 block(64, 4), grid(alva_divup(width, 256), alva_divup(height, 4));
    hipLaunchKernelGGL(k_rgba2gray, grid, block, 0, ctx->stream, d_rgba, rgba_pitch, width, height, d_gray, gray_pitch);
    ALVA_LAUNCH_CHECK();
    return ALVA_OK;
}

extern "C" int alva_pyramid_create(alva_ctx *ctx, int width, int height, int win, int max_level, alva_pyramid **out) {
    ALVA_ARG(ctx && out);
    ALVA_ARG(width > 0 && height > 0 && width % 4 == 0 && win > 2 && win <= 15 && max_level >= 0 && max_level < 8);
    ALVA_HIP(hipSetDevice(ctx->device));
    alva_pyramid *p = new alva_pyramid();
    p->device = ctx->device;
    p->win = win;
    int w = width, h = height;
    for (int l = 0; l <= max_level; l++) {
        alva_level &L = p->lv[l];
        L.w = w;
        L.h = h;
        size_t xoff = (16 - win % 16) % 16;
        L.gray_pitch = round_up(xoff + (size_t) w + 2 * win + 8, 64);
        size_t gbytes = L.gray_pitch * (size_t) (h + 2 * win) + 64;
        size_t xoffd = (4 - win % 4) % 4;
        L.deriv_pitch = round_up((xoffd + (size_t) w + 2 * win) * 4 + 16, 64);
        size_t dbytes = L.deriv_pitch * (size_t) (h + 2 * win) + 64;
        if (hipMalloc((void **) &L.gray_base, gbytes) != hipSuccess || hipMalloc((void **) &L.deriv_base, dbytes) != hipSuccess) {
            alva_pyramid_destroy(p);
            alva_set_error("alva_pyramid_create: hipMalloc failed at level %d", l);
            return ALVA_ERR_NOMEM;
        }
        // zero once: the derivative border is CONSTANT 0 and is never written afterwards
        (void) hipMemsetAsync(L.gray_base, 0, gbytes, ctx->stream);
        (void) hipMemsetAsync(L.deriv_base, 0, dbytes, ctx->stream);
        L.gray = L.gray_base + (size_t) win * L.gray_pitch + xoff + win;
        L.deriv = reinterpret_cast<int16_t *>(reinterpret_cast<uint8_t *>(L.deriv_base) + (size_t) win * L.deriv_pitch + (xoffd + win) * 4);
        p->nlevels = l + 1;
        // lkpyramid.cpp:811-816: stop when the next level would be <= win in either dimension
        w = (w + 1) / 2;
        h = (h + 1) / 2;
        if (w <= win || h <= win) break;
    }
    *out = p;
    return ALVA_OK;
}

extern "C" void alva_pyramid_destroy(alva_pyramid *pyr) {
    if (!pyr) return;
    (void) hipSetDevice(pyr->device);
    for (int l = 0; l < 8; l++) {
        if (pyr->lv[l].gray_base) (void) hipFree(pyr->lv[l].gray_base);
        if (pyr->lv[l].deriv_base) (void) hipFree(pyr->lv[l].deriv_base);
    }
    delete pyr;
}

extern "C" int alva_pyramid_num_levels(const alva_pyramid *pyr) { return pyr ? pyr->nlevels : 0; }

extern "C" int alva_pyramid_level(const alva_pyramid *pyr, int level, alva_pyr_level *out) {
    ALVA_ARG(pyr && out && level >= 0 && level < pyr->nlevels);
    const alva_level &L = pyr->lv[level];
    out->width = L.w;
    out->height = L.h;
    out->d_gray = L.gray;
    out->gray_pitch = L.gray_pitch;
    out->d_deriv = L.deriv;
    out->deriv_pitch = L.deriv_pitch;
    return ALVA_OK;
}

extern "C" int alva_pyramid_download_level(alva_ctx *ctx, const alva_pyramid *pyr, int level, uint8_t *h_gray, int16_t *h_deriv) {
    ALVA_ARG(ctx && pyr && level >= 0 && level < pyr->nlevels);
    const alva_level &L = pyr->lv[level];
    const int win = pyr->win;
    const size_t W = (size_t) L.w + 2 * win, H = (size_t) L.h + 2 * win;
    if (h_gray)
        ALVA_HIP(hipMemcpy2DAsync(h_gray, W, L.gray - (size_t) win * L.gray_pitch - win, L.gray_pitch, W, H, hipMemcpyDeviceToHost, ctx->stream));
    if (h_deriv)
        ALVA_HIP(hipMemcpy2DAsync(h_deriv, W * 4, reinterpret_cast<const uint8_t *>(L.deriv) - (size_t) win * L.deriv_pitch - (size_t) win * 4,
                                  L.deriv_pitch, W * 4, H, hipMemcpyDeviceToHost, ctx->stream));
    ALVA_HIP(hipStreamSynchronize(ctx->stream));
    return ALVA_OK;
}

static int stage_args(const alva_pyramid *p, int l, StageArgs &a) {  // returns the number of workgroups of stage l
    const alva_level &L = p->lv[l];
    a = StageArgs{};
    a.g = L.gray;
    a.g_pitch = L.gray_pitch;
    a.w = L.w;
    a.h = L.h;
    a.d = L.deriv;
    a.d_pitch = L.deriv_pitch;
    a.scharr_bx = alva_divup(L.w, 256);
    a.scharr_blocks = a.scharr_bx * alva_divup(L.h, 4);
    a.win = p->win;
    int down_blocks = 0;
    if (l + 1 < p->nlevels) {
        const alva_level &N = p->lv[l + 1];
        a.ng = N.gray;
        a.ng_pitch = N.gray_pitch;
        a.dw = N.w;
        a.dh = N.h;
        a.down_bx = alva_divup(N.w, 256);
        down_blocks = a.down_bx * alva_divup(N.h, 4);
    }
    return a.scharr_blocks + down_blocks;
}

// Is the frame in device memory?  Asked per ALLOCATION, not per call: the answers are kept for the few allocations a process hands in
// (a ring of frames inside one tensor, the wrapper's frame buffer).  A stale answer (an address range freed and handed out again as host
// memory) costs speed, not correctness: both paths read the frame through the same pointer.
static bool frame_in_device_memory(const void *p) {
    struct Range {
        uintptr_t lo, hi;
        bool device;
    };
    static thread_local Range known[8];
    static thread_local int n_known = 0, next = 0;
    const uintptr_t a = (uintptr_t) p;
    for (int i = 0; i < n_known; i++)
        if (a >= known[i].lo && a < known[i].hi) return known[i].device;
    hipPointerAttribute_t at{};
    bool device = false;
    uintptr_t lo = a, hi = a + 1;
    if (hipPointerGetAttributes(&at, p) == hipSuccess) {
        device = at.type == hipMemoryTypeDevice;
        hipDeviceptr_t base = nullptr;
        size_t size = 0;
        if (device && hipMemGetAddressRange(&base, &size, (hipDeviceptr_t) p) == hipSuccess && size) {
            lo = (uintptr_t) base;
            hi = lo + size;
        }
    } else {
        (void) hipGetLastError();   // (an ordinary host pointer: not ours to launch on either way; the kernel's own error will say so)
    }
    known[next] = Range{lo, hi, device};
    next = (next + 1) % 8;
    if (n_known < 8) n_known++;
    return device;
}

// lane_ok: level 0 was deposited on the lane too (a level 0 launched directly runs on the context's own stream: the rest must follow it there)
static int build_rest(alva_ctx *ctx, alva_pyramid *p, bool lane_ok = false) {
    static const bool staged = getenv("ALVA_PYRAMID_STAGES") != nullptr;   // A/B: the chain of stage launches instead of the fused one
    if (!staged && p->nlevels >= 2 && p->nlevels <= 4 && p->win >= 3) {
        RestArgs A{};
        (void) stage_args(p, 0, A.s0);
        A.s0.dw = 0;
        A.nlevels = p->nlevels;
        A.win = p->win;
        int blocks = A.s0.scharr_blocks;
        for (int l = 0; l < p->nlevels; l++) {
            const alva_level &L = p->lv[l];
            A.lv[l] = RestLevel{L.gray, L.gray_pitch, L.deriv, L.deriv_pitch, L.w, L.h};
            if (l >= 1) {
                A.first[l] = blocks;
                A.ts[l] = rest_tile(l);
                A.bx[l] = alva_divup(L.w, rest_tile(l));
                const int tiles = A.bx[l] * alva_divup(L.h, rest_tile(l));
                A.ntiles[l] = tiles;
                blocks += 8 * alva_divup(tiles, 8);   // eight XCD shares of equal length (the kernel drops the padding workgroups)
            }
        }
        if (lane_ok && alva_lane_defer(MK_PYR_REST, ctx, (unsigned) blocks, 0, &A, sizeof(A))) return ALVA_OK;
        hipLaunchKernelGGL(k_pyr_rest, dim3(blocks), dim3(64, 4), 0, ctx->stream, A);
        ALVA_LAUNCH_CHECK();
        return ALVA_OK;
    }
    for (int l = 0; l < p->nlevels; l++) {
        StageArgs a;
        const int blocks = stage_args(p, l, a);
        hipLaunchKernelGGL(k_pyr_stage, dim3(blocks), dim3(64, 4), 0, ctx->stream, a);
        ALVA_LAUNCH_CHECK();
    }
    return ALVA_OK;
}

extern "C" int alva_pyramid_build_from_gray(alva_ctx *ctx, alva_pyramid *pyr, const uint8_t *d_gray, size_t gray_pitch) {
    ALVA_ARG(ctx && pyr && d_gray && gray_pitch % 4 == 0 && ((uintptr_t) d_gray % 4) == 0);
    const alva_level &L = pyr->lv[0];
    ALVA_ARG(gray_pitch >= (size_t) L.w);
    dim3 block(64, 4), grid(alva_divup(L.w, 256), alva_divup(L.h, 4));
    hipLaunchKernelGGL(k_level0<false>, grid, block, 0, ctx->stream, d_gray, gray_pitch, L.w, L.h, pyr->win, L.gray,
                       L.gray_pitch, (uint8_t *) nullptr, (size_t) 0);
    ALVA_LAUNCH_CHECK();
    return build_rest(ctx, pyr);
}

extern "C" int alva_pyramid_build_from_rgba(alva_ctx *ctx, alva_pyramid *pyr, const uint8_t *d_rgba, size_t rgba_pitch,
                                            uint8_t *d_gray_out, size_t gray_out_pitch) {
    ALVA_ARG(ctx && pyr && d_rgba && rgba_pitch % 16 == 0 && ((uintptr_t) d_rgba % 16) == 0);
    const alva_level &L = pyr->lv[0];
    ALVA_ARG(rgba_pitch >= (size_t) L.w * 4);
    if (d_gray_out) ALVA_ARG(gray_out_pitch % 4 == 0 && gray_out_pitch >= (size_t) L.w && ((uintptr_t) d_gray_out % 4) == 0);
    dim3 block(64, 4), grid(alva_divup(L.w, 256), alva_divup(L.h, 4));
    // one launch for the whole pyramid (k_pyr_all) when the frame is in DEVICE memory -- its workgroups read the frame ~7 times, which is
    // free out of L2 and ruinous over the bus (a registered host frame buffer) -- and the launch is this session's own (not a lane's)
    static const bool two_launches = getenv("ALVA_PYRAMID_TWO_LAUNCHES") != nullptr || getenv("ALVA_PYRAMID_STAGES") != nullptr;   // A/B
    if (!two_launches && !g_alva_lane && pyr->nlevels >= 2 && pyr->nlevels <= 4 && pyr->win >= 3 && L.w >= 8 && L.h >= 2 &&
        frame_in_device_memory(d_rgba)) {
        AllArgs A{};
        A.R.nlevels = pyr->nlevels;
        A.R.win = pyr->win;
        A.rgba = d_rgba;
        A.rgba_pitch = rgba_pitch;
        A.gray_out = d_gray_out;
        A.gray_out_pitch = gray_out_pitch;
        int blocks = 0;
        for (int l = 0; l < pyr->nlevels; l++) {
            const alva_level &V = pyr->lv[l];
            A.R.lv[l] = RestLevel{V.gray, V.gray_pitch, V.deriv, V.deriv_pitch, V.w, V.h};
        }
        for (int l = pyr->nlevels - 1; l >= 1; l--) {
            const alva_level &V = pyr->lv[l];
            A.R.first[l] = blocks;
            A.R.ts[l] = all_tile(l, (size_t) L.w * (size_t) L.h);
            A.R.bx[l] = alva_divup(V.w, A.R.ts[l]);
            A.R.ntiles[l] = A.R.bx[l] * alva_divup(V.h, A.R.ts[l]);
            blocks += 8 * alva_divup(A.R.ntiles[l], 8);
        }
        A.l0_first = blocks;
        A.l0_bx = alva_divup(L.w, L0_TW);
        blocks += A.l0_bx * alva_divup(L.h, L0_TH);
        hipLaunchKernelGGL(k_pyr_all, dim3(blocks), block, 0, ctx->stream, A);
        ALVA_LAUNCH_CHECK();
        return ALVA_OK;
    }
    const Level0Args LA{d_rgba, rgba_pitch, L.w, L.h, pyr->win, (int) grid.x, L.gray, L.gray_pitch, d_gray_out, gray_out_pitch};
    if (alva_lane_defer(MK_LEVEL0, ctx, grid.x * grid.y, 0, &LA, sizeof(LA))) return build_rest(ctx, pyr, true);
    hipLaunchKernelGGL(k_level0<true>, grid, block, 0, ctx->stream, d_rgba, rgba_pitch, L.w, L.h, pyr->win, L.gray, L.gray_pitch,
                       d_gray_out, gray_out_pitch);
    ALVA_LAUNCH_CHECK();
    return build_rest(ctx, pyr);
}

// B cameras of the same geometry in five launches (blockIdx.z = camera) instead of 5 B.  The per-camera arguments are staged in
// pinned host memory and read by the workgroups directly.
extern "C" int alva_pyramid_build_from_rgba_batch(alva_ctx *ctx, alva_pyramid *const *pyrs, const uint8_t *const *d_rgba, size_t rgba_pitch,
                                                  uint8_t *const *d_gray_out, size_t gray_out_pitch, int count) {
    ALVA_ARG(ctx && pyrs && d_rgba && count > 0 && count <= 65535 && rgba_pitch % 16 == 0);
    const alva_pyramid *p0 = pyrs[0];
    ALVA_ARG(p0);
    const alva_level &L0 = p0->lv[0];
    ALVA_ARG(rgba_pitch >= (size_t) L0.w * 4);
    if (d_gray_out) ALVA_ARG(gray_out_pitch % 4 == 0 && gray_out_pitch >= (size_t) L0.w);
    // per-camera argument blocks: built in ordinary host memory and copied with hipMemcpyAsync, which stages pageable memory before
    // it returns -- the call only enqueues work, so the staging area must not be one the next call on this context overwrites
    const size_t off_stage = ((size_t) count * sizeof(Level0Item) + 255) / 256 * 256;
    const size_t arg_bytes = off_stage + (size_t) count * p0->nlevels * sizeof(StageArgs);
    std::vector<uint8_t> host(arg_bytes);
    uint8_t *pin = host.data();
    int rc = ALVA_OK;
    Level0Item *items = (Level0Item *) pin;
    StageArgs *st = (StageArgs *) (pin + off_stage);
    int blocks[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int c = 0; c < count; c++) {
        const alva_pyramid *p = pyrs[c];
        ALVA_ARG(p && d_rgba[c] && ((uintptr_t) d_rgba[c] % 16) == 0 && p->nlevels == p0->nlevels && p->win == p0->win && p->lv[0].w == L0.w &&
                 p->lv[0].h == L0.h && p->lv[0].gray_pitch == L0.gray_pitch);
        items[c].src = d_rgba[c];
        items[c].dst = p->lv[0].gray;
        items[c].gray_out = d_gray_out ? d_gray_out[c] : nullptr;
        if (d_gray_out) ALVA_ARG(d_gray_out[c] && ((uintptr_t) d_gray_out[c] % 4) == 0);
        for (int l = 0; l < p->nlevels; l++) blocks[l] = stage_args(p, l, st[(size_t) l * count + c]);
    }
    // the argument blocks go to device memory in one copy: tens of thousands of workgroups fetching them from host memory would
    // each start with a trip over the bus (measured with a pinned staging area: 1.52 instead of 1.85 TB/s)
    uint8_t *dev = nullptr;
    rc = alva_ctx_scratch(ctx, 10, arg_bytes, (void **) &dev);
    if (rc) return rc;
    ALVA_HIP(hipMemcpyAsync(dev, pin, arg_bytes, hipMemcpyHostToDevice, ctx->stream));
    const Level0Item *d_items = (const Level0Item *) dev;
    const StageArgs *d_st = (const StageArgs *) (dev + off_stage);
    const int gx0 = alva_divup(L0.w, 256), gy0 = alva_divup(L0.h, 4);
    hipLaunchKernelGGL(k_level0_batch, dim3(alva_xcd_grid(count, gx0 * gy0)), dim3(64, 4), 0, ctx->stream, d_items, rgba_pitch, L0.w, L0.h, p0->win,
                       L0.gray_pitch, gray_out_pitch, count, gx0, gy0);
    for (int l = 0; l < p0->nlevels; l++)
        hipLaunchKernelGGL(k_pyr_stage_batch, dim3(alva_xcd_grid(count, blocks[l])), dim3(64, 4), 0, ctx->stream, d_st + (size_t) l * count, count,
                           blocks[l]);
    ALVA_LAUNCH_CHECK();
    return ALVA_OK;
}
