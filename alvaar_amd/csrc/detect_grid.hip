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
//   k_select     the cells share one mask and are visited in row-major order in the reference; a circle reaches only
//                the 4 already-visited neighbours, so cells on the anti-diagonal wavefront t = c + 2r are
//                independent.  ONE workgroup walks the wavefronts with the whole mask as a bit-plane in LDS
//                (640x480: 38 KB, 1280x720: 113 KB of the CU's 160 KB), one wave per cell: masked arg-max (first
//                maximum), accept, clear the circle with LDS atomics, second arg-max.
//   k_compact    ordered compaction of primaries + secondaries, the reference's top-up rule (:117-134).
//   k_subpix     one wave per detected corner: 9x9 bilinear patch lane-parallel, the five double accumulators
//                replayed sequentially (one lane each) so the float result is bit-identical.
// This translation unit must be compiled with -ffp-contract=off.
#include "common.hpp"

namespace {

constexpr int MAX_CELL = 40;
constexpr int NCAND = 256;  // sorted candidates kept per cell (4 per lane of the selecting wave)

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
    float *candVal;   // [nCells][NCAND] lambda_min of the cell's best pixels, sorted (value desc, index asc)
    int *candIdx;     // [nCells][NCAND] their in-cell index, -1 = none
    uint8_t *cellOcc; // [nCells] 1 = occupied (skipped and counted)
    int *prim;        // [nCells] packed (y << 16 | x) or -1
    int *sec;         // [nCells]
    int hw[MAX_CELL / 4 + 1];  // filled-circle half widths
    int dbg;
    long long *dbgbuf;  // optional cycle counters (ALVA_DBG_SELECT=5)
};

__global__ void __launch_bounds__(256) k_mark_occupied(GridArgs A) {
    int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= A.nOcc) return;
    float px = A.occupied[2 * i], py = A.occupied[2 * i + 1];
    int cy = (int) (py / (float) A.cell), cx = (int) (px / (float) A.cell);  // occupiedCells[px.y / cellSize][px.x / cellSize] (:32)
    if (cy >= 0 && cy < A.nCH && cx >= 0 && cx < A.nCW) A.cellOcc[cy * A.nCW + cx] = 1;
}

// ---------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_cell_eig(GridArgs A) {
    extern __shared__ __align__(16) unsigned char smem[];
    const int cell = A.cell, n2 = cell * cell;
    const int ci = blockIdx.x;
    const int r = ci / A.nCW, c = ci % A.nCW;
    const int x0 = c * cell, y0 = r * cell;
    if (A.cellOcc[ci]) return;
    if (!(x0 + cell < A.w - 1 && y0 + cell < A.h - 1)) return;  // feature_extractor.cpp:62
    // LDS carve
    float *sdx = reinterpret_cast<float *>(smem);
    float *sdy = sdx + n2;
    double *sR = reinterpret_cast<double *>(sdy + n2);     // one channel of row sums at a time
    float *sbox = reinterpret_cast<float *>(sR + n2);      // 3 channels
    uint8_t *sB = reinterpret_cast<uint8_t *>(sbox + 3 * n2);
    uint8_t *sG = sB + n2;                                   // (cell+2)^2 gray with 1-px halo
    const int gw = cell + 2;
    for (int i = threadIdx.x; i < gw * gw; i += 256) {
        int ly = i / gw, lx = i % gw;
        sG[i] = A.gray[(size_t) refl(y0 + ly - 1, A.h) * A.pitch + refl(x0 + lx - 1, A.w)];
    }
    __syncthreads();
    for (int i = threadIdx.x; i < n2; i += 256) {
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
    for (int i = threadIdx.x; i < n2; i += 256) {
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
    for (int i = threadIdx.x; i < np2; i += 256) {
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
            for (int i = threadIdx.x; i < np2; i += 256) {
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
    if (threadIdx.x < NCAND) {
        const int i = threadIdx.x;
        float v = 0.f;
        int idx = -1;
        if (i < n2) {
            const unsigned long long key = skey[i];
            const unsigned ord = (unsigned) (key >> 32);
            const unsigned u = (ord & 0x80000000u) ? (ord & 0x7fffffffu) : ~ord;
            v = __uint_as_float(u);
            idx = (int) (0xffffffffu - (unsigned) (key & 0xffffffffu));
        }
        A.candVal[(size_t) ci * NCAND + i] = v;
        A.candIdx[(size_t) ci * NCAND + i] = idx;
    }
}

// ---------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(1024) k_select(GridArgs A) {
    extern __shared__ uint32_t smask[];
    // kernel arguments live in the kernarg segment: copy what the dependent loop needs into registers once, so that no
    // scalar load (and its wait) sits on the per-wavefront critical path
    const int IW = A.w, IH = A.h, NCW = A.nCW, NCH = A.nCH, RX0 = A.roiX, RY0 = A.roiY, RX1 = A.roiX + A.roiW, RY1 = A.roiY + A.roiH;
    const int DBG = A.dbg, RAD = A.radius;
    const double MAXQ = A.maxQuality;
    const float *const EIG = A.eig;
    const float *const CANDV = A.candVal;
    const int *const CANDI = A.candIdx;
    const int wordsPerRow = (IW + 31) / 32;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwaves = 16;
    const int cell = A.cell, n2 = cell * cell, side = 2 * RAD + 1;
    uint8_t *s_occ = reinterpret_cast<uint8_t *>(smask + wordsPerRow * IH);   // per-cell occupied flags
    uint16_t *s_xy = reinterpret_cast<uint16_t *>(s_occ + ((NCW * NCH + 15) & ~15));  // in-cell index -> (dy << 8) | dx
    uint16_t *s_circ = s_xy + ((n2 + 7) & ~7);  // filled-circle pixels: ((dy + 128) << 8) | (dx + 128), 0xffff = outside
    // results stay in LDS until the end: on CDNA4 vmcnt also counts stores, so a global store per step would sit on the
    // critical path of the next step's candidate-load wait
    int *s_prim = reinterpret_cast<int *>(s_circ + ((side * side + 7) & ~7));
    int *s_sec = s_prim + NCW * NCH;
    for (int i = threadIdx.x; i < NCW * NCH; i += 1024) {
        s_prim[i] = -1;
        s_sec[i] = -1;
    }
    for (int i = threadIdx.x; i < wordsPerRow * IH; i += 1024) smask[i] = 0xffffffffu;
    for (int i = threadIdx.x; i < NCW * NCH; i += 1024) s_occ[i] = A.cellOcc[i];
    for (int i = threadIdx.x; i < n2; i += 1024) s_xy[i] = (uint16_t) (((i / cell) << 8) | (i % cell));
    for (int i = threadIdx.x; i < side * side; i += 1024) {
        const int dy = i / side - RAD, dx = i % side - RAD;
        const int half = A.hw[dy < 0 ? -dy : dy];
        s_circ[i] = (dx < 0 ? -dx : dx) <= half ? (uint16_t) (((dy + 128) << 8) | (dx + 128)) : (uint16_t) 0xffff;
    }
    __syncthreads();
    const unsigned circ0 = lane < side * side ? s_circ[lane] : 0xffffu;  // this lane's circle pixel (radius <= 3: all of them)
    auto clear = [&](int cx, int cy) {
        for (int i = lane; i < side * side; i += 64) {
            const unsigned e = i == lane ? circ0 : s_circ[i];
            if (e == 0xffffu) continue;
            const int x = cx + (int) (e & 255u) - 128, y = cy + (int) (e >> 8) - 128;
            if (x >= 0 && x < IW && y >= 0 && y < IH) atomicAnd(&smask[y * wordsPerRow + (x >> 5)], ~(1u << (x & 31)));
        }
    };
    // pre-zero circles of the occupied keypoints, centre = Point(cvRound(px)) (:32-36)
    for (int k = wave; k < A.nOcc; k += nwaves) clear(__float2int_rn(A.occupied[2 * k]), __float2int_rn(A.occupied[2 * k + 1]));
    __syncthreads();
    const int T = (NCW - 1) + 2 * (NCH - 1);
    // Each wave's cells are known in advance (cell (r, c) on wavefront t = c + 2r, r = rmin + wave + 16 k).  The sorted
    // candidate list of the NEXT wavefront's cell (4 per lane) is requested from HBM/L2 at the START of a step and first
    // touched after the barrier that ends it, so the load latency hides behind the current cell + the barrier; inside
    // the dependent section a pass is: one LDS mask bit per candidate, one ballot, one circle of LDS atomics.
    constexpr int SLOTS = 3;  // cells per wave per wavefront whose candidates are prefetched (16 waves x 3 = 48 cells)
    float cv[SLOTS][4], nv[SLOTS][4];
    int ck[SLOTS][4], nk[SLOTS][4];
    auto cell_of = [&](int t, int slot, int &r, int &c) -> bool {
        const int rmin = max(0, (t - (NCW - 1) + 1) / 2), rmax = min(NCH - 1, t / 2);
        r = rmin + wave + slot * nwaves;
        c = t - 2 * r;
        return r <= rmax && c >= 0 && c < NCW;
    };
    auto usable = [&](int r, int c) -> bool {
        const int ci = r * NCW + c;
        return !s_occ[ci] && (c * cell + cell < IW - 1 && r * cell + cell < IH - 1);
    };
#define ALVA_LOAD_CANDS(V, K, r, c)                                                   \
    do {                                                                              \
        const size_t base_ = (size_t) ((r) * NCW + (c)) * NCAND;                     \
        _Pragma("unroll") for (int q = 0; q < 4; q++) {                               \
            K[q] = CANDI[base_ + lane + 64 * q];                                  \
            V[q] = CANDV[base_ + lane + 64 * q];                                  \
        }                                                                             \
    } while (0)
    {
        int r, c;
#pragma unroll
        for (int sl = 0; sl < SLOTS; sl++)
            if (cell_of(0, sl, r, c) && usable(r, c)) ALVA_LOAD_CANDS(cv[sl], ck[sl], r, c);
    }
    long long c_pre = 0, c_work = 0, c_bar = 0, c_rot = 0, c0 = 0, c1 = 0;
    for (int t = 0; t <= T; t++) {
        if (DBG == 5) c0 = wall_clock64();
        {
            int r, c;
#pragma unroll
            for (int sl = 0; sl < SLOTS; sl++)
                if (t < T && cell_of(t + 1, sl, r, c) && usable(r, c)) ALVA_LOAD_CANDS(nv[sl], nk[sl], r, c);
        }
        if (DBG == 5) { c1 = wall_clock64(); c_pre += c1 - c0; c0 = c1; }
#pragma unroll
        for (int slot = 0; slot < SLOTS + 1; slot++) {
          // slots 0..SLOTS-1 use prefetched registers; slot SLOTS loops over any remaining cells with late fetches
          for (int sl2 = slot;; sl2 += 1) {
            int r, c;
            if (!cell_of(t, sl2, r, c)) break;
            float (&CV)[4] = cv[slot < SLOTS ? slot : 0];
            int (&CK)[4] = ck[slot < SLOTS ? slot : 0];
            const int ci = r * NCW + c;
            int prim = -1, sec = -1;
            const int x0 = c * cell, y0 = r * cell;
            if (usable(r, c) && DBG != 1) {
                if (slot >= SLOTS) ALVA_LOAD_CANDS(CV, CK, r, c);  // more than 48 cells on this wavefront: fetch late
                // absolute pixel of each candidate (one batch of independent LDS reads per cell, not per pass)
                int cxy[4];
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    const unsigned e = CK[q] >= 0 ? s_xy[CK[q]] : 0u;
                    cxy[q] = ((y0 + (int) (e >> 8)) << 16) | (x0 + (int) (e & 255u));
                }
                for (int pass = 0; pass < (DBG == 2 ? 1 : 2); pass++) {
                    // first unmasked candidate with a positive value, in sorted order = the reference's masked arg-max
                    float best = 0.f;
                    int bi = -1, bxy = 0;
#pragma unroll
                    for (int q = 0; q < 4; q++) {
                        if (bi >= 0) break;
                        bool hit = false;
                        if (CK[q] >= 0 && CV[q] > 0.f) {
                            const int x = cxy[q] & 0xffff, y = cxy[q] >> 16;
                            hit = DBG == 4 ? true : (bool) ((smask[y * wordsPerRow + (x >> 5)] >> (x & 31)) & 1u);
                        }
                        const unsigned long long m = __ballot(hit);
                        if (m) {
                            const int src = __ffsll((long long) m) - 1;  // wave-uniform -> v_readlane, no LDS crossbar
                            best = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(CV[q]), src));
                            bi = __builtin_amdgcn_readlane(CK[q], src);
                            bxy = __builtin_amdgcn_readlane(cxy[q], src);
                        }
                    }
                    if (bi < 0) {
                        // rare: every listed candidate is masked (or <= 0): exact full scan of eig * mask, first maximum
                        const float *eig = EIG + (size_t) ci * n2;
                        best = -3.402823466e+38f;
                        bi = 0x7fffffff;
                        for (int k = lane; k < n2; k += 64) {
                            const unsigned e = s_xy[k];
                            const int x = x0 + (int) (e & 255u), y = y0 + (int) (e >> 8);
                            const float m = (float) ((smask[y * wordsPerRow + (x >> 5)] >> (x & 31)) & 1u);
                            const float v = eig[k] * m;
                            if (v > best) {
                                best = v;
                                bi = k;
                            }
                        }
#pragma unroll
                        for (int off = 32; off > 0; off >>= 1) {
                            const float ob = __shfl_down(best, off);
                            const int oi = __shfl_down(bi, off);
                            if (ob > best || (ob == best && oi < bi)) {
                                best = ob;
                                bi = oi;
                            }
                        }
                        best = __shfl(best, 0);
                        bi = __shfl(bi, 0);
                        if (bi == 0x7fffffff) bi = 0;  // nothing exceeded -FLT_MAX: minMaxLoc reports index 0
                        const unsigned eb = s_xy[bi];
                        bxy = ((y0 + (int) (eb >> 8)) << 16) | (x0 + (int) (eb & 255u));
                    }
                    const int mx = bxy & 0xffff, my = bxy >> 16;
                    if (mx < RX0 || my < RY0 || mx >= RX1 || my >= RY1) break;  // `continue` of the cell loop
                    if ((double) best >= MAXQ) {
                        if (pass == 0) prim = (my << 16) | mx;
                        else sec = (my << 16) | mx;
                        if (DBG != 3) clear(mx, my);
                    }
                }
            }
            if (lane == 0) {
                s_prim[ci] = prim;
                s_sec[ci] = sec;
            }
            if (slot < SLOTS) break;  // prefetched slots handle exactly one cell each
          }
        }
        if (DBG == 5) { c1 = wall_clock64(); c_work += c1 - c0; c0 = c1; }
        __syncthreads();
        if (DBG == 5) { c1 = wall_clock64(); c_bar += c1 - c0; c0 = c1; }
#pragma unroll
        for (int sl = 0; sl < SLOTS; sl++)
#pragma unroll
            for (int q = 0; q < 4; q++) {
                cv[sl][q] = nv[sl][q];
                ck[sl][q] = nk[sl][q];
            }
        if (DBG == 5) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); c1 = wall_clock64(); c_rot += c1 - c0; }
    }
    if (DBG == 5 && A.dbgbuf && lane == 0) {
        A.dbgbuf[4 * wave + 0] = c_pre;
        A.dbgbuf[4 * wave + 1] = c_work;
        A.dbgbuf[4 * wave + 2] = c_bar;
        A.dbgbuf[4 * wave + 3] = c_rot;
    }
    for (int i = threadIdx.x; i < NCW * NCH; i += 1024) {
        A.prim[i] = s_prim[i];
        A.sec[i] = s_sec[i];
    }
#undef ALVA_LOAD_CANDS
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

__global__ void __launch_bounds__(64) k_subpix(const uint8_t *__restrict__ gray, size_t pitch, int w, int h, float *__restrict__ pts,
                                              const CompactOut *__restrict__ cnt, int cap) {
    const int n = min(cnt->n_total, cap);
    const int pi = blockIdx.x;
    if (pi >= n) return;
    constexpr int WINH = 3, WW = 7, BW = WW + 2;
    __shared__ float s_buf[BW * BW];
    __shared__ float s_mask[WW * WW];
    __shared__ double s_term[5][WW * WW + 1];
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
        for (int e = lane; e < BW * BW; e += 64) s_buf[e] = subpix_at(g, gray, pitch, e / BW, e % BW, BW);
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
        const double a = __shfl(acc, 0), b = __shfl(acc, 1), c = __shfl(acc, 2), bb1 = __shfl(acc, 3), bb2 = __shfl(acc, 4);
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

extern "C" int alva_detect_grid(alva_ctx *ctx, const uint8_t *d_gray, size_t gray_pitch, int width, int height, int cell_size,
                                const float *d_occupied, int n_occ, int roi_x, int roi_y, int roi_w, int roi_h, double *h_max_quality,
                                float *d_out_pts, int cap, int *h_count) {
    ALVA_ARG(ctx && d_gray && h_max_quality && d_out_pts && h_count && width > 8 && height > 8 && cap >= 0 && n_occ >= 0);
    ALVA_ARG(cell_size >= 4 && cell_size <= MAX_CELL);
    ALVA_ARG(n_occ == 0 || d_occupied);
    ALVA_ARG(width < 65536 && height < 32768);
    *h_count = 0;
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
    A.maxQuality = *h_max_quality;
    A.occupied = d_occupied;
    A.nOcc = n_occ;
    circle_halfwidths(A.radius, A.hw);
    A.dbg = getenv("ALVA_DBG_SELECT") ? atoi(getenv("ALVA_DBG_SELECT")) : 0;
    A.dbgbuf = nullptr;
    if (A.dbg == 5) {
        static long long *dbg_dev = nullptr;
        if (!dbg_dev) (void) hipMalloc((void **) &dbg_dev, 64 * 8);
        A.dbgbuf = dbg_dev;
    }
    const int nCells = A.nCW * A.nCH, n2 = cell_size * cell_size;
    if (nCells == 0) return ALVA_OK;
    size_t off_occ = (size_t) nCells * n2 * 4, off_prim = (off_occ + nCells + 63) / 64 * 64, off_sec = off_prim + (size_t) nCells * 4,
           off_cv = (off_sec + (size_t) nCells * 4 + 63) / 64 * 64, off_ci = off_cv + (size_t) nCells * NCAND * 4,
           off_out = off_ci + (size_t) nCells * NCAND * 4;
    uint8_t *base = nullptr;
    int rc = alva_ctx_scratch(ctx, 5, off_out + 64, (void **) &base);
    if (rc) return rc;
    A.eig = (float *) base;
    A.cellOcc = base + off_occ;
    A.prim = (int *) (base + off_prim);
    A.sec = (int *) (base + off_sec);
    A.candVal = (float *) (base + off_cv);
    A.candIdx = (int *) (base + off_ci);
    CompactOut *d_cnt = (CompactOut *) (base + off_out);
    hipStream_t st = ctx->stream;
    ALVA_HIP(hipMemsetAsync(A.cellOcc, 0, (size_t) nCells, st));
    if (n_occ > 0) hipLaunchKernelGGL(k_mark_occupied, dim3(alva_divup(n_occ, 256)), dim3(256), 0, st, A);
    int np2 = 256;
    while (np2 < n2) np2 <<= 1;
    const size_t lds_eig = (size_t) n2 * (4 + 4 + 8 + 12 + 1) + (size_t) (cell_size + 2) * (cell_size + 2) + 32 + (size_t) np2 * 8;
    ALVA_ARG(lds_eig <= 64 * 1024);
    hipLaunchKernelGGL(k_cell_eig, dim3(nCells), dim3(256), lds_eig, st, A);
    const int side = 2 * A.radius + 1;
    const size_t lds_mask = (size_t) ((width + 31) / 32) * height * 4 + (size_t) nCells + 16 + (size_t) (n2 + 8) * 2 + (size_t) (side * side + 8) * 2 + (size_t) nCells * 8 + 64;
    ALVA_ARG(lds_mask <= 160 * 1024 - 1024);
    if (lds_mask > 48 * 1024)
        ALVA_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_select), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 1024));
    hipLaunchKernelGGL(k_select, dim3(1), dim3(1024), lds_mask, st, A);
    CompactOut *h_cnt = nullptr;
    rc = alva_ctx_pinned(ctx, sizeof(CompactOut), (void **) &h_cnt);
    if (rc) return rc;
    hipLaunchKernelGGL(k_compact, dim3(1), dim3(1024), 0, st, A, d_out_pts, cap, d_cnt, h_cnt);
    ALVA_LAUNCH_CHECK();
    // one wave per candidate slot; the kernel reads the actual count from device memory (no host round trip before it)
    const int maxPts = std::min(cap, 2 * nCells);
    if (maxPts > 0) hipLaunchKernelGGL(k_subpix, dim3(maxPts), dim3(64), 0, st, d_gray, gray_pitch, width, height, d_out_pts, d_cnt, cap);
    ALVA_LAUNCH_CHECK();
    ALVA_HIP(hipStreamSynchronize(st));
    const CompactOut res = *h_cnt;
    if (A.dbg == 5 && A.dbgbuf) {
        long long hb[64];
        (void) hipMemcpy(hb, A.dbgbuf, sizeof(hb), hipMemcpyDeviceToHost);
        static int once = 0;
        if (once++ == 3)
            for (int wv = 0; wv < 16; wv++)
                fprintf(stderr, "[select dbg] wave %2d: prefetch %lld work %lld barrier %lld rotate %lld (100 MHz ticks)\n", wv, hb[4 * wv], hb[4 * wv + 1], hb[4 * wv + 2], hb[4 * wv + 3]);
    }
    *h_count = res.n_total;
    // adaptive threshold (:138-145)
    const double freeCells = (double) ((size_t) nCells - (size_t) res.n_occupied);
    if ((double) res.n_total < 0.33 * freeCells) *h_max_quality *= 0.5;
    else if ((double) res.n_total > 0.9 * freeCells) *h_max_quality *= 1.5;
    return ALVA_OK;
}
