// a5: the reference's per-grid-cell Shi-Tomasi detector, bit-exact.
//
// Replaces FeatureExtractor::detectFeaturePoints (src/slam/src/feature_extractor.cpp:11-158).  Arithmetic restated
// from the vendored OpenCV exactly as the reference's 128-bit-SIMD build executes it (see oracle/alva_oracle.c,
// orc_cell_mineig / orc_detect_grid / orc_corner_subpix, which is pinned bitwise to the compiled reference):
//   3x3 Gaussian of the cell, reading true neighbours (smooth.dispatch.cpp:654,754; integer, the vector columns round
//   half-to-even, the scalar tail half-up, filter.simd.hpp:1010-1099), Sobel with the 1/3060 scale folded into the
//   smoothing taps (deriv.cpp:427-439), dx^2 / dxdy / dy^2, 3x3 box as SLIDING double sums (box_filter.simd.hpp),
//   lambda_min (corner.cpp:52-102), masked first-maximum x2 with filled circles zeroed in a shared mask
//   (drawing.cpp:1477-1617), cornerSubPix (cornersubpix.cpp:44-156, samplers.cpp:129-268).
//
// Structure on the GPU:
//   k_cell_eig   one workgroup per grid cell, everything in LDS: blur -> Sobel -> products -> sliding box sums
//                (one lane per row, then one lane per column: the rounding history of the running sums is part of
//                the result) -> lambda_min written once to HBM (4 B/px).  Algorithmic HBM traffic: read P, write 4P.
//   selection    the cells share one mask and are visited in row-major order in the reference; a circle (radius cell/4)
//                reaches only the 4 already-visited neighbours, so the result is the fixed point of a DAG recurrence.
//                k_cell_eig also makes every cell's speculative pick (no neighbour circles), k_round repairs the cells
//                whose pick lies under a neighbour's circle (one wave per cell on all CUs, Jacobi rounds, one launch
//                each), k_select finishes in one workgroup in the rare case that 4 rounds were not enough.  See the
//                comment above SelView.
//   k_compact    ordered compaction of primaries + secondaries, the reference's top-up rule (:117-134).
//   k_subpix     one wave per detected corner: 9x9 bilinear patch lane-parallel, the five double accumulators
//                replayed sequentially (one lane each) so the float result is bit-identical.
// This translation unit must be compiled with -ffp-contract=off.
#include "common.hpp"
#include <cstdlib>
#include "wave_utils.hpp"

namespace {

constexpr int MAX_CELL = 40;
// per-cell stride of the sorted candidate lists (uint16 entries): all n2 pixels, padded to 8, + one terminator chunk
__host__ __device__ inline int cand_stride(int n2) { return ((n2 + 7) & ~7) + 8; }

__device__ __forceinline__ int refl(int p, int len) {
    if (len == 1) return 0;
    while (p < 0 || p >= len) p = p < 0 ? -p : 2 * (len - 1) - p;
    return p;
}

struct GridArgs {
    const uint8_t *gray;
    size_t pitch;
    int w, h, cell, nCW, nCH, radius;
    int roiX, roiY, roiW, roiH;
    double maxQuality;
    const float *occupied;
    int nOcc;
    float *eig;       // [nCells][cell*cell]
    uint16_t *cand;   // [nCells][cand_stride] in-cell index of the cell's pixels, sorted (lambda_min desc, index asc); 0xffff
                      // ends the list (no more pixels with a positive value)
    uint8_t *cellOcc; // [nCells] 1 = occupied (skipped and counted)
    uint32_t *maskBits;  // static part of the reference's mask as a bit-plane [h][(w+31)/32]: 0 under the circles of the
                         // tracked keypoints (:32-36), 1 elsewhere
    uint8_t *needFull;   // [nCells] 1 = the speculative pick could not be made (only when maxQuality <= 0)
    int maskInLds;       // k_select keeps a copy of maskBits in LDS (when it fits next to the per-cell arrays)
    int *nGood;          // [nCells] length of the list prefix whose lambda_min >= maxQuality (the list is sorted)
    int *prim;        // [2][nCells] packed (y << 16 | x) or -1: double buffer of the fixed-point rounds
    int *sec;         // [2][nCells]
    uint8_t *dirty;   // [2][nCells] cell has to be looked at in the round that reads this buffer
    uint8_t *infl;    // [nCells] last evaluation met a neighbour's circle (or must be exact: needFull)
    int *roundCnt;    // [8] number of cells changed by round k
    int hw[MAX_CELL / 4 + 1];  // filled-circle half widths
};

__device__ __forceinline__ int hw_of(const GridArgs &A, int dy) { return A.hw[dy < 0 ? -dy : dy]; }

// mask = ones, no cell occupied (one launch instead of two fill commands)
__global__ void __launch_bounds__(256) k_prepare(GridArgs A) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    const int words = ((A.w + 31) / 32) * A.h;
    if (i < words) A.maskBits[i] = 0xffffffffu;
    if (i < A.nCW * A.nCH) A.cellOcc[i] = 0;
    if (i < 8) A.roundCnt[i] = 0;
}

// one wave per tracked keypoint: mark its cell, zero its filled circle (centre = Point(cvRound(px)), :32-36)
__global__ void __launch_bounds__(256) k_mark_occupied(GridArgs A) {
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (i >= A.nOcc) return;
    const float px = A.occupied[2 * i], py = A.occupied[2 * i + 1];
    const int cy = (int) (py / (float) A.cell), cx = (int) (px / (float) A.cell);  // occupiedCells[px.y / cellSize][px.x / cellSize] (:32)
    if (lane == 0 && cy >= 0 && cy < A.nCH && cx >= 0 && cx < A.nCW) A.cellOcc[cy * A.nCW + cx] = 1;
    const int mx = __float2int_rn(px), my = __float2int_rn(py), R = A.radius, side = 2 * R + 1, wpr = (A.w + 31) / 32;
    for (int k = lane; k < side * side; k += 64) {
        const int dy = k / side - R, dx = k % side - R;
        if ((dx < 0 ? -dx : dx) > hw_of(A, dy)) continue;
        const int x = mx + dx, y = my + dy;
        if (x >= 0 && x < A.w && y >= 0 && y < A.h) atomicAnd(&A.maskBits[y * wpr + (x >> 5)], ~(1u << (x & 31)));
    }
}

// ---------------------------------------------------------------------------------------------------------------
// NT = 64 for cells up to 16x16 (one wave per cell: the many phase barriers cost nothing and 4x more cells are in flight
// per CU), 256 for larger cells.
template<int NT>
__global__ void __launch_bounds__(NT) k_cell_eig(GridArgs A) {
    extern __shared__ __align__(16) unsigned char smem[];
    const int cell = A.cell, n2 = cell * cell;
    const int ci = blockIdx.x;
    const int r = ci / A.nCW, c = ci % A.nCW;
    const int x0 = c * cell, y0 = r * cell;
    if (A.cellOcc[ci] || !(x0 + cell < A.w - 1 && y0 + cell < A.h - 1)) {  // feature_extractor.cpp:50-54, :62
        if (threadIdx.x == 0) {
            A.prim[ci] = -1;
            A.sec[ci] = -1;
            A.needFull[ci] = 0;
            A.nGood[ci] = 0;
            A.infl[ci] = 0;
            A.dirty[ci] = 0;
            A.dirty[A.nCW * A.nCH + ci] = 0;
        }
        return;
    }
    // LDS carve
    float *sdx = reinterpret_cast<float *>(smem);
    float *sdy = sdx + n2;
    double *sR = reinterpret_cast<double *>(sdy + n2);     // one channel of row sums at a time
    float *sbox = reinterpret_cast<float *>(sR + n2);      // 3 channels
    uint8_t *sB = reinterpret_cast<uint8_t *>(sbox + 3 * n2);
    uint8_t *sG = sB + n2;                                   // (cell+2)^2 gray with 1-px halo
    const int gw = cell + 2;
    for (int i = threadIdx.x; i < gw * gw; i += NT) {
        int ly = i / gw, lx = i % gw;
        sG[i] = A.gray[(size_t) refl(y0 + ly - 1, A.h) * A.pitch + refl(x0 + lx - 1, A.w)];
    }
    __syncthreads();
    for (int i = threadIdx.x; i < n2; i += NT) {
        int y = i / cell, x = i % cell;
        const uint8_t *g = sG + (y + 1) * gw + (x + 1);
        int acc = g[-gw - 1] + 2 * g[-gw] + g[-gw + 1] + 2 * g[-1] + 4 * g[0] + 2 * g[1] + g[gw - 1] + 2 * g[gw] + g[gw + 1];
        int q = acc >> 4, rem = acc & 15, v;
        if (x < (cell & ~3)) v = rem > 8 ? q + 1 : (rem < 8 ? q : (q + (q & 1)));  // vector columns: round half to even
        else v = (acc + 8) >> 4;                                                   // scalar tail: half up
        sB[i] = (uint8_t) min(v, 255);
    }
    __syncthreads();
    const float s = (float) (1.0 / (4.0 * 3.0 * 255.0));
    const float s2 = 2.0f * s;
    for (int i = threadIdx.x; i < n2; i += NT) {
        int y = i / cell, x = i % cell;
        int ym = refl(y - 1, cell), yp = refl(y + 1, cell), xm = refl(x - 1, cell), xp = refl(x + 1, cell);
        float a0 = sB[ym * cell + xm], a1 = sB[ym * cell + x], a2 = sB[ym * cell + xp];
        float b0 = sB[y * cell + xm], b2 = sB[y * cell + xp];
        float c0 = sB[yp * cell + xm], c1 = sB[yp * cell + x], c2 = sB[yp * cell + xp];
        float r0 = a2 - a0, r1 = b2 - b0, r2 = c2 - c0;
        sdx[i] = (r0 + r2) * s + r1 * s2;
        float up = (s * a0 + s2 * a1) + s * a2;
        float dn = (s * c0 + s2 * c1) + s * c2;
        sdy[i] = dn - up;
    }
    __syncthreads();
    for (int ch = 0; ch < 3; ch++) {
        // RowSum<float,double>: s = c[-1] + c[0] + c[1]; then s += c[x+1] - c[x-2]   (one lane per row)
        if (threadIdx.x < cell) {
            const int y = threadIdx.x;
            auto cov = [&](int xx) -> double {
                const int k = y * cell + refl(xx, cell);
                const float fx = sdx[k], fy = sdy[k];
                const float v = ch == 0 ? fx * fx : (ch == 1 ? fx * fy : fy * fy);
                return (double) v;
            };
            double acc = 0;
            acc += cov(-1);
            acc += cov(0);
            acc += cov(1);
            sR[y * cell] = acc;
            for (int x = 1; x < cell; x++) {
                acc += cov(x + 1) - cov(x - 2);
                sR[y * cell + x] = acc;
            }
        }
        __syncthreads();
        // ColumnSum<double,float>: SUM = R[-1] + R[0]; out = (float)(SUM + R[y+1]); SUM = that - R[y-1]   (one lane per column)
        if (threadIdx.x < cell) {
            const int x = threadIdx.x;
            double SUM = 0;
            SUM += sR[refl(-1, cell) * cell + x];
            SUM += sR[x];
            for (int y = 0; y < cell; y++) {
                const double s0 = SUM + sR[refl(y + 1, cell) * cell + x];
                sbox[ch * n2 + y * cell + x] = (float) s0;
                SUM = s0 - sR[refl(y - 1, cell) * cell + x];
            }
        }
        __syncthreads();
    }
    float *out = A.eig + (size_t) ci * n2;
    // sort keys (value desc, index asc) -> 64-bit key sorted descending: [orderable float bits | ~index]
    unsigned long long *skey = reinterpret_cast<unsigned long long *>(smem + ((29 * n2 + gw * gw + 15) & ~15));
    int np2 = 256;
    while (np2 < n2) np2 <<= 1;
    for (int i = threadIdx.x; i < np2; i += NT) {
        unsigned long long key = 0;
        if (i < n2) {
            const float a = sbox[i] * 0.5f, b = sbox[n2 + i], cc = sbox[2 * n2 + i] * 0.5f;
            const float t = a - cc;
            const float e = (a + cc) - sqrtf(b * b + t * t);
            out[i] = e;
            const unsigned u = __float_as_uint(e);
            const unsigned ord = (u & 0x80000000u) ? ~u : (u | 0x80000000u);
            key = ((unsigned long long) ord << 32) | (unsigned long long) (0xffffffffu - (unsigned) i);
        }
        skey[i] = key;
    }
    __syncthreads();
    for (int k = 2; k <= np2; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = threadIdx.x; i < np2; i += NT) {
                const int l = i ^ j;
                if (l > i) {
                    const unsigned long long x = skey[i], y = skey[l];
                    const bool desc = (i & k) == 0;
                    if ((x < y) == desc) {
                        skey[i] = y;
                        skey[l] = x;
                    }
                }
            }
            __syncthreads();
        }
    // the whole sorted list (index only; 0xffff from the first non-positive value on, and as terminator chunk), and the
    // length of its prefix that passes the quality threshold (the only thing the selection needs the values for)
    const int stride = cand_stride(n2);
    __shared__ int s_good;
    if (threadIdx.x == 0) s_good = 0;
    __syncthreads();
    int good = 0;
    for (int i = threadIdx.x; i < stride; i += NT) {
        uint16_t o = 0xffff;
        if (i < n2) {
            const unsigned long long key = skey[i];
            const unsigned ord = (unsigned) (key >> 32);
            const unsigned u = (ord & 0x80000000u) ? (ord & 0x7fffffffu) : ~ord;
            const float v = __uint_as_float(u);
            if (v > 0.f) o = (uint16_t) (0xffffffffu - (unsigned) (key & 0xffffffffu));
            good += (double) v >= A.maxQuality;
        }
        A.cand[(size_t) ci * stride + i] = o;
    }
    if (good) atomicAdd(&s_good, good);
    __syncthreads();
    if (threadIdx.x == 0) A.nGood[ci] = s_good;
    // Speculative pick of this cell, as if no other cell had drawn a circle yet (k_select then repairs the few cells whose
    // pick lies under a neighbour's circle): wave 0 walks the sorted list 64 entries at a time; "first free entry" is one
    // ballot.  Pass 0 = primary; pass 1 continues behind it with the primary's own circle masked as well.
    if (threadIdx.x < 64) {
        const int lane = threadIdx.x, wpr = (A.w + 31) / 32, R = A.radius;
        const int RX1 = A.roiX + A.roiW, RY1 = A.roiY + A.roiH;
        int prim = -1, sec = -1, need = 0, pos = 0, own = -1;
        for (int pass = 0; pass < 2; pass++) {
            int fxy = -1;
            float fv = 0.f;
            while (pos < n2) {
                const int i = pos + lane;
                bool ok = false, nonpos = false;
                float v = 0.f;
                int xy = 0;
                if (i < n2) {
                    const unsigned long long key = skey[i];
                    const unsigned ord = (unsigned) (key >> 32);
                    v = __uint_as_float((ord & 0x80000000u) ? (ord & 0x7fffffffu) : ~ord);
                    const int idx = (int) (0xffffffffu - (unsigned) (key & 0xffffffffu));
                    const int x = x0 + idx % cell, y = y0 + idx / cell;
                    xy = (y << 16) | x;
                    nonpos = !(v > 0.f);
                    if (!nonpos) {
                        ok = (A.maskBits[y * wpr + (x >> 5)] >> (x & 31)) & 1u;
                        if (ok && own >= 0) {
                            int dy = y - (own >> 16), dx = x - (own & 0xffff);
                            dy = dy < 0 ? -dy : dy;
                            dx = dx < 0 ? -dx : dx;
                            if (dy <= R && dx <= A.hw[dy]) ok = false;
                        }
                    }
                }
                const unsigned long long m = __ballot(ok), mz = __ballot(nonpos);
                // the list is sorted: an entry counts only if it comes before the first non-positive one
                const int firstOk = m ? __ffsll((long long) m) - 1 : 64, firstZ = mz ? __ffsll((long long) mz) - 1 : 64;
                if (firstOk < firstZ) {
                    fv = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(v), firstOk));
                    fxy = __builtin_amdgcn_readlane(xy, firstOk);
                    pos += firstOk + 1;
                    break;
                }
                if (mz) {
                    pos = n2;  // nothing positive left
                    break;
                }
                pos += 64;
            }
            if (fxy < 0) {
                if (!(A.maxQuality > 0.0)) need = 1;  // the exact arg-max over non-positive values is left to k_select
                break;
            }
            const int mx = fxy & 0xffff, my = fxy >> 16;
            if (mx < A.roiX || my < A.roiY || mx >= RX1 || my >= RY1) break;
            if (!((double) fv >= A.maxQuality)) break;
            if (pass == 0) {
                prim = fxy;
                own = fxy;
            } else {
                sec = fxy;
            }
        }
        if (lane == 0) {
            A.prim[ci] = need ? -1 : prim;
            A.sec[ci] = need ? -1 : sec;
            A.needFull[ci] = (uint8_t) need;
            A.infl[ci] = (uint8_t) need;
            A.dirty[ci] = 1;   // round 0 looks at every cell
            A.dirty[A.nCW * A.nCH + ci] = 0;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Selection.  The reference visits the cells in row-major order and every accepted point zeroes a filled circle of
// radius cell/4 in a mask shared by all cells, so cell i sees the circles of cells j < i -- of its four already-visited
// neighbours only, because the radius is smaller than a cell.  That is a recurrence v_i = f_i(v_j, j < i) on a DAG, and
// its solution is the unique fixed point of "every cell re-evaluates f_i on the current values of its neighbours".
// So instead of walking the 2r + c wavefronts one after another (133 dependent steps for 640x480 / cell 12):
//   * k_cell_eig already made every cell's SPECULATIVE pick (no neighbour circles at all), in parallel on all CUs;
//   * each round, one thread per cell checks whether the cell has to be re-evaluated: neighbours' circles can only
//     REMOVE candidates, so a pick made without any candidate being rejected by a neighbour stands unless a present
//     circle covers the primary or the secondary itself; a pick that did meet a neighbour's circle (`influenced`) is
//     re-evaluated whenever a predecessor changed;
//   * the (few) cells that need it are re-evaluated by one WAVE each: 64 entries of the cell's sorted candidate list
//     (value desc, index asc = the reference's "first maximum") per step, "first free entry" = one ballot; free = not
//     under a static circle (tracked keypoints, bit-plane), one of <= 8 neighbour circles or the own primary's;
//   * a changed cell marks its four successors for the next round; a round without change is the fixed point, i.e.
//     exactly the sequential result (at most longest-dependency-path + 1 rounds; 3 on the bench frames).
struct SelView {
    const uint32_t *smask;    // static bit-plane (LDS copy), only read when hasStatic
    unsigned long long hwp;   // filled-circle half widths, 4 bits per |dy| (radius <= 10)
    int wordsPerRow, cell, inv, n2, R, hasStatic;
    int RX0, RY0, RX1, RY1;
    double MAXQ;
};

__device__ __forceinline__ bool sel_covered(int x, int y, int cxy, const SelView &V) {
    int dy = y - (cxy >> 16);  // cxy = -1 (no circle) gives dx far outside any radius
    dy = dy < 0 ? -dy : dy;
    int dx = x - (cxy & 0xffff);
    dx = dx < 0 ? -dx : dx;
    const int hw = (int) ((V.hwp >> (4 * min(dy, 15))) & 15ull);
    return cxy >= 0 && dy <= V.R && dx <= hw;
}

// the <= 8 circles of the four predecessors of cell (r, c); those that cannot reach the cell are dropped (-1)
__device__ __forceinline__ void sel_circles(const int *s_prim, const int *s_sec, int ci, int r, int c, int NCW, int cell, int R, int circ[8]) {
    const bool up = r > 0, lf = c > 0, rt = c + 1 < NCW;
    const int n0 = up && lf ? ci - NCW - 1 : -1, n1 = up ? ci - NCW : -1, n2i = up && rt ? ci - NCW + 1 : -1, n3 = lf ? ci - 1 : -1;
    circ[0] = n0 >= 0 ? s_prim[n0] : -1;
    circ[1] = n0 >= 0 ? s_sec[n0] : -1;
    circ[2] = n1 >= 0 ? s_prim[n1] : -1;
    circ[3] = n1 >= 0 ? s_sec[n1] : -1;
    circ[4] = n2i >= 0 ? s_prim[n2i] : -1;
    circ[5] = n2i >= 0 ? s_sec[n2i] : -1;
    circ[6] = n3 >= 0 ? s_prim[n3] : -1;
    circ[7] = n3 >= 0 ? s_sec[n3] : -1;
    const int x0 = c * cell, y0 = r * cell;
#pragma unroll
    for (int q = 0; q < 8; q++) {
        const int cx = circ[q] & 0xffff, cy = circ[q] >> 16;
        if (circ[q] >= 0 && (cx + R < x0 || cx - R >= x0 + cell || cy + R < y0 || cy - R >= y0 + cell)) circ[q] = -1;
    }
}

// f_i by one wave: (primary, secondary) of one cell given its predecessors' circles.  `pre` = entries 0..63 of the cell's
// sorted list (prefetched), influenced = some candidate was rejected by a neighbour's circle.  All results wave-uniform.
__device__ void sel_eval_wave(const SelView &V, int x0, int y0, unsigned pre0, unsigned pre1, const uint16_t *cand, int stride, int nGood,
                              const float *eig, const int circ[8], int &prim, int &sec, int &influenced) {
    const int lane = threadIdx.x & 63;
    prim = -1;
    sec = -1;
    influenced = 0;
    int own = -1, pos = 0;
    for (int pass = 0; pass < 2; pass++) {
        int fxy = -1, fpos = 0;
        while (pos < stride) {
            // the 64-entry chunk that contains pos (chunks 0 and 1 were prefetched); entries before pos are already consumed
            const int base = pos & ~63, i = base + lane;
            unsigned idx = 0xffffu;
            if (i < stride) idx = base == 0 ? pre0 : (base == 64 ? pre1 : (unsigned) cand[i]);
            const bool live = i >= pos;
            const bool end = live && idx == 0xffffu;
            bool ok = false, byNb = false;
            int xy = 0;
            if (live && !end) {
                const int dyc = (int) ((idx * (unsigned) V.inv) >> 16), dxc = (int) idx - dyc * V.cell;
                const int x = x0 + dxc, y = y0 + dyc;
                xy = (y << 16) | x;
                ok = !sel_covered(x, y, own, V);
                if (ok) {
#pragma unroll
                    for (int q = 0; q < 8; q++)
                        if (circ[q] >= 0) byNb = byNb || sel_covered(x, y, circ[q], V);
                    ok = !byNb;
                    if (ok && V.hasStatic) ok = (V.smask[y * V.wordsPerRow + (x >> 5)] >> (x & 31)) & 1u;
                }
            }
            const unsigned long long m = __ballot(ok), me = __ballot(end), mn = __ballot(byNb);
            const int firstOk = m ? __ffsll((long long) m) - 1 : 64, firstEnd = me ? __ffsll((long long) me) - 1 : 64;
            const int upto = min(firstOk, firstEnd);  // entries actually examined by the sequential scan
            if (mn & (upto >= 64 ? ~0ull : ((1ull << upto) - 1ull))) influenced = 1;
            if (firstOk < firstEnd) {
                fxy = __builtin_amdgcn_readlane(xy, firstOk);
                fpos = base + firstOk;
                pos = fpos + 1;  // entries before it stay masked in the second pass, this one by its own circle
                break;
            }
            if (me) {
                pos = stride;  // nothing positive left
                break;
            }
            pos = base + 64;
        }
        bool accept;
        if (fxy >= 0) {
            accept = fpos < nGood;  // lambda_min >= maxQuality: the sorted list's first nGood entries
        } else {
            // No free pixel with a positive value: the masked arg-max is <= 0.  With a positive quality threshold (always, in
            // the reference: maxQuality_ starts at 0.001 and only halves) nothing can be accepted, whatever the ROI test says.
            if (V.MAXQ > 0.0) break;
            // general case: exact scan of eig * mask, first maximum (minMaxLoc)
            float best = -3.402823466e+38f;
            int bi = 0x7fffffff;
            for (int q = lane; q < V.n2; q += 64) {
                const int dyc = (int) (((unsigned) q * (unsigned) V.inv) >> 16), dxc = q - dyc * V.cell;
                const int x = x0 + dxc, y = y0 + dyc;
                bool fr = !sel_covered(x, y, own, V);
#pragma unroll
                for (int t = 0; t < 8; t++)
                    if (circ[t] >= 0) fr = fr && !sel_covered(x, y, circ[t], V);
                if (fr && V.hasStatic) fr = (V.smask[y * V.wordsPerRow + (x >> 5)] >> (x & 31)) & 1u;
                const float v = eig[q] * (fr ? 1.f : 0.f);
                if (v > best) {
                    best = v;
                    bi = q;
                }
            }
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) {
                const float ob = __shfl_xor(best, off);
                const int oi = __shfl_xor(bi, off);
                if (ob > best || (ob == best && oi < bi)) {
                    best = ob;
                    bi = oi;
                }
            }
            if (bi == 0x7fffffff) bi = 0;  // nothing exceeded -FLT_MAX: minMaxLoc reports index 0
            const int dyc = (int) (((unsigned) bi * (unsigned) V.inv) >> 16), dxc = bi - dyc * V.cell;
            fxy = ((y0 + dyc) << 16) | (x0 + dxc);
            influenced = 1;  // exact path: always re-evaluate when a predecessor changes
            accept = (double) best >= V.MAXQ;
        }
        const int mx = fxy & 0xffff, my = fxy >> 16;
        if (mx < V.RX0 || my < V.RY0 || mx >= V.RX1 || my >= V.RY1) break;  // `continue` of the cell loop (:76-79, :90-93)
        if (!accept) break;  // no circle drawn: the second arg-max finds the same pixel and rejects it too
        if (pass == 0) {
            prim = fxy;
            own = fxy;
        } else {
            sec = fxy;
        }
    }
}

// One Jacobi round on all CUs: one wave per cell, reads buffer `src`, writes buffer `src ^ 1`.
__global__ void __launch_bounds__(256) k_round(GridArgs A, int src, int round) {
    const int NCW = A.nCW, NCH = A.nCH, nCells = NCW * NCH, cell = A.cell, RAD = A.radius;
    const int lane = threadIdx.x & 63, ci = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (ci >= nCells) return;
    const int dst = src ^ 1;
    const int *Ps = A.prim + (size_t) src * nCells, *Ss = A.sec + (size_t) src * nCells;
    uint8_t *dS = A.dirty + (size_t) src * nCells, *dD = A.dirty + (size_t) dst * nCells;
    const int p0 = Ps[ci], q0 = Ss[ci];
    int p = p0, q = q0;
    if (dS[ci]) {  // wave-uniform
        const int r = ci / NCW, c = ci - r * NCW;
        SelView V;
        V.smask = A.maskBits;
        unsigned long long hwp = 0;
        for (int d = 0; d <= RAD; d++) hwp |= (unsigned long long) (A.hw[d] & 15) << (4 * d);
        V.hwp = hwp;
        V.wordsPerRow = (A.w + 31) / 32;
        V.cell = cell;
        V.inv = (65536 + cell - 1) / cell;
        V.n2 = cell * cell;
        V.R = RAD;
        V.hasStatic = A.nOcc > 0;
        V.RX0 = A.roiX;
        V.RY0 = A.roiY;
        V.RX1 = A.roiX + A.roiW;
        V.RY1 = A.roiY + A.roiH;
        V.MAXQ = A.maxQuality;
        int circ[8];
        sel_circles(Ps, Ss, ci, r, c, NCW, cell, RAD, circ);
        bool need = A.infl[ci];
        if (!need) {
#pragma unroll
            for (int t = 0; t < 8; t++) {
                if (p0 >= 0) need = need || sel_covered(p0 & 0xffff, p0 >> 16, circ[t], V);
                if (q0 >= 0) need = need || sel_covered(q0 & 0xffff, q0 >> 16, circ[t], V);
            }
        }
        if (need) {
            const int cstride = cand_stride(V.n2);
            const uint16_t *l = A.cand + (size_t) ci * cstride;
            const unsigned pre0 = lane < cstride ? (unsigned) l[lane] : 0xffffu;
            const unsigned pre1 = 64 + lane < cstride ? (unsigned) l[64 + lane] : 0xffffu;
            int infl;
            sel_eval_wave(V, c * cell, r * cell, pre0, pre1, l, cstride, A.nGood[ci], A.eig + (size_t) ci * V.n2, circ, p, q, infl);
            if (lane == 0) A.infl[ci] = (uint8_t) (infl | A.needFull[ci]);
        }
        if (lane == 0) {
            dS[ci] = 0;  // this buffer is the next round's destination: leave it clean
            if (p != p0 || q != q0) {
                atomicAdd(&A.roundCnt[round], 1);
                // successors in visiting order: right, below-left, below, below-right (cells that do not take part ignore it)
                if (c + 1 < NCW) dD[ci + 1] = 1;
                if (r + 1 < NCH) {
                    if (c > 0) dD[ci + NCW - 1] = 1;
                    dD[ci + NCW] = 1;
                    if (c + 1 < NCW) dD[ci + NCW + 1] = 1;
                }
            }
        }
    }
    if (lane == 0) {
        A.prim[(size_t) dst * nCells + ci] = p;
        A.sec[(size_t) dst * nCells + ci] = q;
    }
}

// Finishes the iteration in ONE workgroup (Gauss-Seidel order, no launch per round) when the Jacobi rounds above did not
// reach the fixed point yet; returns at once when the last of them changed nothing.
__global__ void __launch_bounds__(1024) k_select(GridArgs A, int buf, int lastRound) {
    extern __shared__ int s_prim[];
    if (A.roundCnt[lastRound] == 0) return;  // the last Jacobi round changed nothing: buffer `buf` is the fixed point
    const int IW = A.w, IH = A.h, NCW = A.nCW, NCH = A.nCH, nCells = NCW * NCH;
    const int cell = A.cell, n2 = cell * cell, RAD = A.radius;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwaves = 16;
    const int pad = (nCells + 15) & ~15;
    int *s_sec = s_prim + nCells;
    int *s_list = s_sec + nCells;                                    // cells to re-evaluate this round
    uint8_t *s_use = reinterpret_cast<uint8_t *>(s_list + nCells);   // cell takes part (not occupied, not at the border)
    uint8_t *s_dirty0 = s_use + pad;
    uint8_t *s_dirty1 = s_dirty0 + pad;
    uint8_t *s_infl = s_dirty1 + pad;   // last evaluation met a neighbour's circle (or must be exact: needFull)
    uint8_t *s_need = s_infl + pad;
    int *s_good = reinterpret_cast<int *>(s_need + pad);
    uint32_t *smask = reinterpret_cast<uint32_t *>(s_good + nCells);  // static bit-plane, only when there are tracked keypoints
    __shared__ int s_flag[2], s_n;
    const int hasStatic = A.nOcc > 0;
    const int wordsPerRow = (IW + 31) / 32;
    for (int i = threadIdx.x; i < nCells; i += 1024) {
        s_need[i] = A.needFull[i];
        s_infl[i] = A.infl[i];
        s_good[i] = A.nGood[i];
        s_prim[i] = A.prim[(size_t) buf * nCells + i];
        s_sec[i] = A.sec[(size_t) buf * nCells + i];
        const int r = i / NCW, c = i - r * NCW;
        const bool use = !A.cellOcc[i] && (c * cell + cell < IW - 1 && r * cell + cell < IH - 1);  // :50-54, :62
        s_use[i] = use;
        s_dirty0[i] = use && A.dirty[(size_t) buf * nCells + i];
        s_dirty1[i] = 0;
    }
    if (hasStatic && A.maskInLds)
        for (int i = threadIdx.x; i < wordsPerRow * IH; i += 1024) smask[i] = A.maskBits[i];
    unsigned long long hwp = 0;
    for (int d = 0; d <= RAD; d++) hwp |= (unsigned long long) (A.hw[d] & 15) << (4 * d);
    if (threadIdx.x < 2) s_flag[threadIdx.x] = 0;
    if (threadIdx.x == 0) s_n = 0;
    __syncthreads();
    SelView V;
    V.smask = A.maskInLds ? smask : A.maskBits;
    V.hwp = hwp;
    V.wordsPerRow = wordsPerRow;
    V.cell = cell;
    V.inv = (65536 + cell - 1) / cell;   // idx / cell == (idx * inv) >> 16 for idx < cell^2 <= 1600 (checked exhaustively)
    V.n2 = n2;
    V.R = RAD;
    V.hasStatic = hasStatic;
    V.RX0 = A.roiX;
    V.RY0 = A.roiY;
    V.RX1 = A.roiX + A.roiW;
    V.RY1 = A.roiY + A.roiH;
    V.MAXQ = A.maxQuality;
    const int cstride = cand_stride(n2);
    const int maxRounds = NCW + 2 * NCH + 4;  // longest dependency path (the 2r + c wavefront count) + slack
    uint8_t *dcur = s_dirty0, *dnext = s_dirty1;
    for (int round = 0; round < maxRounds; round++) {
        // ---- A: one thread per marked cell decides whether it has to be re-evaluated --------------------------------------
        for (int ci = threadIdx.x; ci < nCells; ci += 1024) {
            if (!dcur[ci]) continue;
            dcur[ci] = 0;
            bool need = s_infl[ci];
            if (!need) {
                const int r = ci / NCW, c = ci - r * NCW;
                int circ[8];
                sel_circles(s_prim, s_sec, ci, r, c, NCW, cell, RAD, circ);
                const int p0 = s_prim[ci], q0 = s_sec[ci];
#pragma unroll
                for (int q = 0; q < 8; q++) {
                    if (p0 >= 0) need = need || sel_covered(p0 & 0xffff, p0 >> 16, circ[q], V);
                    if (q0 >= 0) need = need || sel_covered(q0 & 0xffff, q0 >> 16, circ[q], V);
                }
            }
            if (need) s_list[atomicAdd(&s_n, 1)] = ci;
        }
        __syncthreads();
        // ---- B: one wave per listed cell ----------------------------------------------------------------------------
        const int nList = s_n;
        int changed = 0;
        unsigned pre0 = 0xffffu, pre1 = 0xffffu;
        auto fetch = [&](int cell_index) {
            const uint16_t *l = A.cand + (size_t) cell_index * cstride;
            pre0 = lane < cstride ? (unsigned) l[lane] : 0xffffu;
            pre1 = 64 + lane < cstride ? (unsigned) l[64 + lane] : 0xffffu;
        };
        if (wave < nList) fetch(s_list[wave]);
        for (int e = wave; e < nList; e += nwaves) {
            const int ci = s_list[e];
            const unsigned cur0 = pre0, cur1 = pre1;
            if (e + nwaves < nList) fetch(s_list[e + nwaves]);  // next cell's first 128 entries, in flight during this cell
            const int r = ci / NCW, c = ci - r * NCW;
            int circ[8];
            sel_circles(s_prim, s_sec, ci, r, c, NCW, cell, RAD, circ);
            int prim, sec, infl;
            sel_eval_wave(V, c * cell, r * cell, cur0, cur1, A.cand + (size_t) ci * cstride, cstride, s_good[ci], A.eig + (size_t) ci * n2, circ, prim,
                          sec, infl);
            if (lane == 0) {
                s_infl[ci] = (uint8_t) (infl | s_need[ci]);
                if (prim != s_prim[ci] || sec != s_sec[ci]) {
                    s_prim[ci] = prim;
                    s_sec[ci] = sec;
                    changed = 1;
                    // successors in visiting order: right, below-left, below, below-right
                    if (c + 1 < NCW && s_use[ci + 1]) dnext[ci + 1] = 1;
                    if (r + 1 < NCH) {
                        if (c > 0 && s_use[ci + NCW - 1]) dnext[ci + NCW - 1] = 1;
                        if (s_use[ci + NCW]) dnext[ci + NCW] = 1;
                        if (c + 1 < NCW && s_use[ci + NCW + 1]) dnext[ci + NCW + 1] = 1;
                    }
                }
            }
        }
        if (changed) s_flag[round & 1] = 1;
        __syncthreads();
        const int any = s_flag[round & 1];
        if (threadIdx.x == 0) {
            s_flag[(round + 1) & 1] = 0;
            s_n = 0;
        }
        if (!any) break;
        uint8_t *t = dcur;
        dcur = dnext;
        dnext = t;
        __syncthreads();
    }
    for (int i = threadIdx.x; i < nCells; i += 1024) {
        A.prim[(size_t) buf * nCells + i] = s_prim[i];
        A.sec[(size_t) buf * nCells + i] = s_sec[i];
    }
}

struct CompactOut {
    int n_total, n_occupied;
};

// ordered compaction: primaries in cell order, then up to numSec secondaries in cell order (:107-134)
__global__ void __launch_bounds__(1024) k_compact(GridArgs A, float *__restrict__ pts, int cap, CompactOut *__restrict__ out,
                                                  CompactOut *__restrict__ host_out) {
    __shared__ int s_cnt[1024];
    __shared__ int s_base;
    const int nCells = A.nCW * A.nCH;
    const int per = (nCells + 1023) / 1024;
    const int b = threadIdx.x * per, e = min(nCells, b + per);
    int total = 0, nocc = 0;
    for (int phase = 0; phase < 3; phase++) {
        // phase 0: count occupied; 1: primaries; 2: secondaries
        int cnt = 0;
        for (int i = b; i < e; i++) {
            if (phase == 0) cnt += A.cellOcc[i];
            else cnt += (phase == 1 ? A.prim[i] : A.sec[i]) >= 0;
        }
        s_cnt[threadIdx.x] = cnt;
        __syncthreads();
        // inclusive scan (Hillis-Steele)
        for (int off = 1; off < 1024; off <<= 1) {
            int v = threadIdx.x >= off ? s_cnt[threadIdx.x - off] : 0;
            __syncthreads();
            s_cnt[threadIdx.x] += v;
            __syncthreads();
        }
        const int mine_end = s_cnt[threadIdx.x], all = s_cnt[1023];
        const int start = mine_end - cnt;
        __syncthreads();
        if (phase == 0) {
            nocc = all;
        } else {
            int limit = 0x7fffffff;
            if (phase == 2) {
                // numSec = numCells - (numKeypoints + numOccupied), only if numKeypoints + numOccupied < numCells
                limit = (total + nocc < nCells) ? nCells - (total + nocc) : 0;
            }
            int k = start;
            for (int i = b; i < e; i++) {
                const int v = phase == 1 ? A.prim[i] : A.sec[i];
                if (v < 0) continue;
                if (k < limit) {
                    const int dst = total + k;
                    if (dst < cap) {
                        pts[2 * dst] = (float) (v & 0xffff);
                        pts[2 * dst + 1] = (float) (v >> 16);
                    }
                }
                k++;
            }
            total += min(all, limit);
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        out->n_total = total;
        out->n_occupied = nocc;
        host_out->n_total = total;  // pinned host mirror: the host reads it after the stream sync, no copy command
        host_out->n_occupied = nocc;
    }
    (void) s_base;
}

// ---------------------------------------------------------------------------------------------------------------
// cv::getRectSubPix(8U -> 32F) for one output element (samplers.cpp:219-268 fast path, :129-216 border path)
struct SubpixGeom {
    int inside;
    int ipx, ipy;
    float a, b, a11, a12, a21, a22, b1, b2, oma;
    double s;
    int rx, ry, rw, rh, offx, offy;  // border path
};

__device__ SubpixGeom subpix_geom(float cx, float cy, int ww, int wh, int w, int h) {
    SubpixGeom g;
    cx -= (ww - 1) * 0.5f;
    cy -= (wh - 1) * 0.5f;
    g.ipx = (int) floorf(cx);
    g.ipy = (int) floorf(cy);
    g.inside = 0 <= g.ipx && g.ipx + ww < w && 0 <= g.ipy && g.ipy + wh < h;
    float a = cx - (float) g.ipx, b = cy - (float) g.ipy;
    if (g.inside) {
        a = fmaxf(a, 0.0001f);
        g.a12 = a * (1.f - b);
        g.a22 = a * b;
        g.b1 = 1.f - b;
        g.b2 = b;
        g.oma = 1 - a;
        g.s = (1. - (double) a) / (double) a;
    } else {
        g.a11 = (1.f - a) * (1.f - b);
        g.a12 = a * (1.f - b);
        g.a21 = (1.f - a) * b;
        g.a22 = a * b;
        g.b1 = 1.f - b;
        g.b2 = b;
        // adjustRect (samplers.cpp:43-110): offsets of the source pointer, and the valid rectangle
        int offx = 0, offy = 0;
        if (g.ipx >= 0) { offx += g.ipx; g.rx = 0; } else { g.rx = -g.ipx; if (g.rx > ww) g.rx = ww; }
        if (g.ipx < w - ww) g.rw = ww; else { g.rw = w - g.ipx - 1; if (g.rw < 0) { offx += g.rw; g.rw = 0; } }
        if (g.ipy >= 0) { offy += g.ipy; g.ry = 0; } else g.ry = -g.ipy;
        if (g.ipy < h - wh) g.rh = wh; else { g.rh = h - g.ipy - 1; if (g.rh < 0) { offy += g.rh; g.rh = 0; } }
        g.offx = offx - g.rx;
        g.offy = offy;
    }
    g.a = a;
    g.b = b;
    return g;
}

__device__ float subpix_at(const SubpixGeom &g, const uint8_t *src, size_t pitch, int i, int j, int ww) {
    if (g.inside) {
        const uint8_t *p = src + (size_t) (g.ipy + i) * pitch + g.ipx;
        // dst[j] = prev + t_j ; prev = (j == 0) ? (1-a)(b1 p[0] + b2 p[step]) : (float)(t_{j-1} * s)
        const float t = g.a12 * (float) p[j + 1] + g.a22 * (float) p[j + 1 + pitch];
        float prev;
        if (j == 0) prev = g.oma * (g.b1 * (float) p[0] + g.b2 * (float) p[pitch]);
        else {
            const float tp = g.a12 * (float) p[j] + g.a22 * (float) p[j + pitch];
            prev = (float) ((double) tp * g.s);
        }
        return prev + t;
    }
    // border path: the row pointer advances only while i < rh; rows outside [ry, rh) read the same row twice
    int row = g.offy;
    for (int k = 0; k < i; k++)
        if (k < g.rh) {
            // src = src2, where src2 = src + step unless (k < ry || k >= rh)
            if (!(k < g.ry || k >= g.rh)) row++;
        }
    const int row2 = (i < g.ry || i >= g.rh) ? row : row + 1;
    const uint8_t *p = src + (ptrdiff_t) row * (ptrdiff_t) pitch + g.offx;
    const uint8_t *p2 = src + (ptrdiff_t) row2 * (ptrdiff_t) pitch + g.offx;
    if (j < g.rx) return (float) p[g.rx] * g.b1 + (float) p2[g.rx] * g.b2;
    if (j >= g.rw) return (float) p[g.rw] * g.b1 + (float) p2[g.rw] * g.b2;
    return (((float) p[j] * g.a11 + (float) p[j + 1] * g.a12) + (float) p2[j] * g.a21) + (float) p2[j + 1] * g.a22;
}

// inside path of subpix_at with the source pixels in an LDS tile whose (0,0) is image pixel (tx0, ty0)
__device__ __forceinline__ float subpix_at_tile(const SubpixGeom &g, const uint8_t *tile, int tpitch, int tx0, int ty0, int i, int j) {
    const uint8_t *p = tile + (g.ipy + i - ty0) * tpitch + (g.ipx - tx0);
    const float t = g.a12 * (float) p[j + 1] + g.a22 * (float) p[j + 1 + tpitch];
    float prev;
    if (j == 0) prev = g.oma * (g.b1 * (float) p[0] + g.b2 * (float) p[tpitch]);
    else {
        const float tp = g.a12 * (float) p[j] + g.a22 * (float) p[j + tpitch];
        prev = (float) ((double) tp * g.s);
    }
    return prev + t;
}

// The slowest corner (30 iterations) sets the kernel time, so the iteration is kept short: the 10x10 source footprint of
// the 9x9 patch is read from a 24x24 LDS tile (re-staged from L2 only when the window leaves it), broadcasts are v_readlane.
__global__ void __launch_bounds__(64) k_subpix(const uint8_t *__restrict__ gray, size_t pitch, int w, int h, float *__restrict__ pts,
                                              const CompactOut *__restrict__ cnt, int cap) {
    const int n = min(cnt->n_total, cap);
    const int pi = blockIdx.x;
    if (pi >= n) return;
    constexpr int WINH = 3, WW = 7, BW = WW + 2;
    constexpr int TS = 24, TM = (TS - BW - 1) / 2;  // tile side, margin around the footprint when staged
    __shared__ float s_buf[BW * BW];
    __shared__ float s_mask[WW * WW];
    __shared__ double s_term[5][WW * WW + 1];
    __shared__ uint8_t s_tile[TS * TS];
    int tx0 = 0x40000000, ty0 = 0x40000000;  // tile origin (invalid: staged on first use)
    const int lane = threadIdx.x;
    // exp(-(k/3)^2), k = 0..3, as glibc's expf returns them (the reference computes the mask with std::exp(float))
    const uint32_t ebits[4] = {0x3f800000u, 0x3f651430u, 0x3f242466u, 0x3ebc5ab2u};
    if (lane < WW * WW) {
        const int i = lane / WW, j = lane % WW;
        const float vy = __uint_as_float(ebits[abs(i - WINH)]), vx = __uint_as_float(ebits[abs(j - WINH)]);
        s_mask[lane] = vy * vx;
    }
    const float cTx = pts[2 * pi], cTy = pts[2 * pi + 1];
    float cIx = cTx, cIy = cTy;
    int iter = 0;
    const double eps = 0.01 * 0.01;
    double err = 0;
    do {
        const SubpixGeom g = subpix_geom(cIx, cIy, BW, BW, w, h);
        __syncthreads();
        if (g.inside) {  // wave-uniform
            if (g.ipx < tx0 || g.ipx + BW + 1 > tx0 + TS || g.ipy < ty0 || g.ipy + BW + 1 > ty0 + TS) {
                tx0 = g.ipx - TM;
                ty0 = g.ipy - TM;
                for (int e = lane; e < TS * TS; e += 64) {
                    const int yy = min(max(ty0 + e / TS, 0), h - 1), xx = min(max(tx0 + e % TS, 0), w - 1);  // clamped bytes are never used
                    s_tile[e] = gray[(size_t) yy * pitch + xx];
                }
                __syncthreads();
            }
            for (int e = lane; e < BW * BW; e += 64) s_buf[e] = subpix_at_tile(g, s_tile, TS, tx0, ty0, e / BW, e % BW);
        } else {
            for (int e = lane; e < BW * BW; e += 64) s_buf[e] = subpix_at(g, gray, pitch, e / BW, e % BW, BW);
        }
        __syncthreads();
        // the 49 per-pixel terms are evaluated lane-parallel (one term per lane) into LDS; lanes 0..4 then replay the
        // reference's five sequential double accumulations a, b, c, bb1, bb2 in row-major order
        if (lane < WW * WW) {
            const int i = lane / WW, j = lane % WW;
            const float *sp = s_buf + (i + 1) * BW + 1;
            const double m = s_mask[lane];
            const double tgx = (double) (sp[j + 1] - sp[j - 1]);
            const double tgy = (double) (sp[j + BW] - sp[j - BW]);
            const double gxx = tgx * tgx * m, gxy = tgx * tgy * m, gyy = tgy * tgy * m, px = j - WINH, py = i - WINH;
            s_term[0][lane] = gxx;
            s_term[1][lane] = gxy;
            s_term[2][lane] = gyy;
            s_term[3][lane] = gxx * px + gxy * py;
            s_term[4][lane] = gxy * px + gyy * py;
        }
        __syncthreads();
        double acc = 0;
        if (lane < 5) {
#pragma unroll 7
            for (int k = 0; k < WW * WW; k++) acc += s_term[lane][k];
        }
        const double a = lane_bcast(acc, 0), b = lane_bcast(acc, 1), c = lane_bcast(acc, 2), bb1 = lane_bcast(acc, 3),
                     bb2 = lane_bcast(acc, 4);
        const double det = a * c - b * b;
        if (fabs(det) <= 2.220446049250313e-16 * 2.220446049250313e-16) break;
        const double scale = 1.0 / det;
        const float nx = (float) ((double) cIx + c * scale * bb1 - b * scale * bb2);
        const float ny = (float) ((double) cIy - b * scale * bb1 + a * scale * bb2);
        const float ex = nx - cIx, ey = ny - cIy;
        err = (double) (ex * ex + ey * ey);
        cIx = nx;
        cIy = ny;
        if (cIx < 0 || cIx >= (float) w || cIy < 0 || cIy >= (float) h) break;
    } while (++iter < 30 && err > eps);
    if (fabs((double) (cIx - cTx)) > WINH || fabs((double) (cIy - cTy)) > WINH) {
        cIx = cTx;
        cIy = cTy;
    }
    if (lane == 0) {
        pts[2 * pi] = cIx;
        pts[2 * pi + 1] = cIy;
    }
}

void circle_halfwidths(int radius, int *hw) {  // drawing.cpp:1477-1617 (Circle, fill)
    for (int i = 0; i <= radius; i++) hw[i] = -1;
    int err = 0, dx = radius, dy = 0, plus = 1, minus = (radius << 1) - 1;
    while (dx >= dy) {
        if (dx > hw[dy]) hw[dy] = dx;
        if (dy > hw[dx]) hw[dx] = dy;
        dy++;
        err += plus;
        plus += 2;
        int mask = (err <= 0) - 1;
        err -= minus & mask;
        dx += mask;
        minus -= mask & 2;
    }
}

}  // namespace

extern "C" int alva_detect_grid_enqueue(alva_ctx *ctx, const uint8_t *d_gray, size_t gray_pitch, int width, int height, int cell_size,
                                        const float *d_occupied, int n_occ, int roi_x, int roi_y, int roi_w, int roi_h, double max_quality,
                                        float *d_out_pts, int cap, alva_detect_pending *pending) {
    ALVA_ARG(ctx && d_gray && pending && d_out_pts && width > 8 && height > 8 && cap >= 0 && n_occ >= 0);
    pending->h_cnt = nullptr;
    pending->n_cells = 0;
    ALVA_ARG(cell_size >= 4 && cell_size <= MAX_CELL);
    ALVA_ARG(n_occ == 0 || d_occupied);
    ALVA_ARG(width < 65536 && height < 32768);
    GridArgs A{};
    A.gray = d_gray;
    A.pitch = gray_pitch;
    A.w = width;
    A.h = height;
    A.cell = cell_size;
    A.nCW = width / cell_size;
    A.nCH = height / cell_size;
    A.radius = cell_size / 4;
    A.roiX = roi_x; A.roiY = roi_y; A.roiW = roi_w; A.roiH = roi_h;
    A.maxQuality = max_quality;
    A.occupied = d_occupied;
    A.nOcc = n_occ;
    circle_halfwidths(A.radius, A.hw);
    const int nCells = A.nCW * A.nCH, n2 = cell_size * cell_size;
    if (nCells == 0) return ALVA_OK;
    size_t off_occ = (size_t) nCells * n2 * 4, off_prim = (off_occ + nCells + 63) / 64 * 64, off_sec = off_prim + (size_t) nCells * 8,
           off_cv = (off_sec + (size_t) nCells * 8 + 63) / 64 * 64, off_out = (off_cv + (size_t) nCells * cand_stride(n2) * 2 + 63) / 64 * 64;
    uint8_t *base = nullptr;
    const size_t off_mask = off_out + 64, off_need = off_mask + (size_t) ((width + 31) / 32) * height * 4;
    const size_t off_good = (off_need + (size_t) nCells + 63) / 64 * 64;
    const size_t off_dirty = off_good + (size_t) nCells * 4, off_infl = off_dirty + 2 * (size_t) nCells;
    const size_t off_rc = (off_infl + (size_t) nCells + 63) / 64 * 64;
    int rc = alva_ctx_scratch(ctx, 5, off_rc + 64, (void **) &base);
    if (rc) return rc;
    A.eig = (float *) base;
    A.cellOcc = base + off_occ;
    A.prim = (int *) (base + off_prim);
    A.sec = (int *) (base + off_sec);
    A.cand = (uint16_t *) (base + off_cv);
    A.maskBits = (uint32_t *) (base + off_mask);
    A.needFull = base + off_need;
    A.nGood = (int *) (base + off_good);
    A.dirty = base + off_dirty;
    A.infl = base + off_infl;
    A.roundCnt = (int *) (base + off_rc);
    CompactOut *d_cnt = (CompactOut *) (base + off_out);
    hipStream_t st = ctx->stream;
    const int maskWords = ((width + 31) / 32) * height;
    hipLaunchKernelGGL(k_prepare, dim3(alva_divup(std::max(maskWords, nCells), 256)), dim3(256), 0, st, A);
    if (n_occ > 0) hipLaunchKernelGGL(k_mark_occupied, dim3(alva_divup(n_occ, 4)), dim3(256), 0, st, A);
    int np2 = 256;
    while (np2 < n2) np2 <<= 1;
    const size_t lds_eig = (size_t) n2 * (4 + 4 + 8 + 12 + 1) + (size_t) (cell_size + 2) * (cell_size + 2) + 32 + (size_t) np2 * 8;
    ALVA_ARG(lds_eig <= 64 * 1024);
    if (n2 <= 256) hipLaunchKernelGGL(k_cell_eig<64>, dim3(nCells), dim3(64), lds_eig, st, A);
    else hipLaunchKernelGGL(k_cell_eig<256>, dim3(nCells), dim3(256), lds_eig, st, A);
    size_t lds_sel = (size_t) nCells * 16 + 5 * (size_t) ((nCells + 15) & ~15) + 64;
    ALVA_ARG(lds_sel <= 160 * 1024 - 1024);
    const size_t mask_bytes = (size_t) ((width + 31) / 32) * height * 4;
    A.maskInLds = n_occ > 0 && lds_sel + mask_bytes <= 160 * 1024 - 1024;
    if (A.maskInLds) lds_sel += mask_bytes;
    if (lds_sel > 48 * 1024)
        ALVA_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_select), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 1024));
    // fixed-point rounds: K on all CUs (3 reach the fixed point on typical frames, the 4th confirms it), the rest -- if any --
    // in one workgroup without further launches
    int K_ROUNDS = 4;
    if (const char *e = getenv("ALVA_GRID_ROUNDS")) K_ROUNDS = std::min(4, std::max(1, atoi(e)));   // test hook: leave work to the finisher
    for (int k = 0; k < K_ROUNDS; k++) hipLaunchKernelGGL(k_round, dim3(alva_divup(nCells, 4)), dim3(256), 0, st, A, k & 1, k);
    const int fin = K_ROUNDS & 1;
    hipLaunchKernelGGL(k_select, dim3(1), dim3(1024), lds_sel, st, A, fin, K_ROUNDS - 1);
    CompactOut *h_cnt = nullptr;
    rc = alva_ctx_pinned(ctx, sizeof(CompactOut), (void **) &h_cnt);
    if (rc) return rc;
    GridArgs AF = A;  // the final buffer
    AF.prim = A.prim + (size_t) fin * nCells;
    AF.sec = A.sec + (size_t) fin * nCells;
    hipLaunchKernelGGL(k_compact, dim3(1), dim3(1024), 0, st, AF, d_out_pts, cap, d_cnt, h_cnt);
    ALVA_LAUNCH_CHECK();
    // one wave per candidate slot; the kernel reads the actual count from device memory (no host round trip before it)
    const int maxPts = std::min(cap, 2 * nCells);
    if (maxPts > 0) hipLaunchKernelGGL(k_subpix, dim3(maxPts), dim3(64), 0, st, d_gray, gray_pitch, width, height, d_out_pts, d_cnt, cap);
    ALVA_LAUNCH_CHECK();
    pending->h_cnt = h_cnt;
    pending->n_cells = nCells;
    return ALVA_OK;
}

// waits for the enqueued detection and applies the adaptive threshold rule (feature_extractor.cpp:138-145)
extern "C" int alva_detect_grid_collect(alva_ctx *ctx, const alva_detect_pending *pending, double *h_max_quality, int *h_count) {
    ALVA_ARG(ctx && pending && h_max_quality && h_count);
    *h_count = 0;
    if (!pending->h_cnt) return ALVA_OK;   // no cells: nothing was enqueued
    ALVA_HIP(alva_stream_sync(ctx->stream));
    const CompactOut res = *static_cast<const CompactOut *>(pending->h_cnt);
    *h_count = res.n_total;
    const double freeCells = (double) ((size_t) pending->n_cells - (size_t) res.n_occupied);
    if ((double) res.n_total < 0.33 * freeCells) *h_max_quality *= 0.5;
    else if ((double) res.n_total > 0.9 * freeCells) *h_max_quality *= 1.5;
    return ALVA_OK;
}

extern "C" int alva_detect_grid(alva_ctx *ctx, const uint8_t *d_gray, size_t gray_pitch, int width, int height, int cell_size,
                                const float *d_occupied, int n_occ, int roi_x, int roi_y, int roi_w, int roi_h, double *h_max_quality,
                                float *d_out_pts, int cap, int *h_count) {
    ALVA_ARG(h_max_quality && h_count);
    alva_detect_pending pending;
    const int rc = alva_detect_grid_enqueue(ctx, d_gray, gray_pitch, width, height, cell_size, d_occupied, n_occ, roi_x, roi_y, roi_w, roi_h,
                                            *h_max_quality, d_out_pts, cap, &pending);
    if (rc) return rc;
    return alva_detect_grid_collect(ctx, &pending, h_max_quality, h_count);
}
