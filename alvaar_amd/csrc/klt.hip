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
#include "multi_kernel.hpp"
#include "wave_utils.hpp"
#include "track_slots.hpp"
#include <cstdlib>

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
#ifdef ALVA_KLT_COUNT
    unsigned cnt[8];   // iterations, origin moves, restages, levels | level-0 failures: template out of range, min-eig / det, window out of bounds
#endif
    alignas(16) uint32_t jt[TW * TW];  // tile of the searched image around the current window (see lk_level), one dword per pixel: a tap's
                                       // two horizontal neighbours come back from ONE ds_read2_b32, ready to multiply
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

// The bilinear sums multiply 8-bit pixels / 13-bit derivatives by 15-bit weights: v_mul_i32_i24 / v_mad_i32_i24 are full-rate, the
// 32-bit v_mul_lo_u32 the compiler must otherwise use is quarter-rate.  SIGNED 24-bit: w11 = 2^14 - w00 - w01 - w10 is -1 when the
// three rounded weights add up to 2^14 + 1 (lkpyramid.cpp:236-239 has the same -1).
__device__ __forceinline__ int bl_u8(int s00, int s01, int s10, int s11, const Weights &w) {
    return __mul24(s00, w.w00) + __mul24(s01, w.w01) + __mul24(s10, w.w10) + __mul24(s11, w.w11);
}
// The weights of the ITERATION (bilinear_weights above, to the bit), in eleven instructions instead of twenty-one.  2^14 moves onto
// the first factor -- a power of two commutes with every rounding here: (1 - a) 2^14 == 2^14 - a 2^14, ((1 - a)(1 - b)) 2^14 ==
// ((1 - a) 2^14)(1 - b) -- and the round-to-nearest-even conversion is the float addition of 2^23: for 0 <= x < 2^23 the sum lies in
// [2^23, 2^24), where floats are the integers, so its mantissa IS rint(x), and its low 24 bits -- all v_mul/mad_i32_i24 read -- are
// that integer (bit 23 of 0x4b000000 is clear).  w11 needs the true integers once: the three biases leave through one constant.
struct WeightsB {
    int w00, w01, w10, w11;   // only the low 24 bits are meaningful (w11: the whole word)
};
__device__ __forceinline__ WeightsB bilinear_weights_i24(float a, float b) {
    const float as = a * 16384.f, omas = 16384.f - as, omb = 1.f - b;
    const float bias = 8388608.f;
    WeightsB w;
    w.w00 = __float_as_int(omas * omb + bias);
    w.w01 = __float_as_int(as * omb + bias);
    w.w10 = __float_as_int(omas * b + bias);
    w.w11 = (int) ((unsigned) ((1 << 14) + 3 * 0x4b000000ll) - (unsigned) w.w00 - (unsigned) w.w01 - (unsigned) w.w10);
    return w;
}

// The same instructions by name, for the iteration of lk_level: left to itself the compiler widens some of these products to the
// quarter-rate v_mul_lo_u32 (it cannot always prove the 24-bit range through the byte unpacking), and the iteration is the kernel.
__device__ __forceinline__ int mul_i24(int a, int b) {
    int r;
    asm("v_mul_i32_i24 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ int mad_i24(int a, int b, int c) {
    int r;
    asm("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
// descale(bl_u8(...), 9): the rounding constant rides on the first multiply-add
__device__ __forceinline__ int tap_u8(const uint32_t (&s)[4], const WeightsB &w, int half) {
    return mad_i24((int) s[3], w.w11, mad_i24((int) s[2], w.w10, mad_i24((int) s[1], w.w01, mad_i24((int) s[0], w.w00, half)))) >> 9;
}
__device__ __forceinline__ int bl_i16(int s00, int s01, int s10, int s11, const Weights &w) {
    return __mul24(s00, w.w00) + __mul24(s01, w.w01) + __mul24(s10, w.w10) + __mul24(s11, w.w11);
}

// ALVA_KLT_COUNT (a build-time switch of tools/klt_slot_stamps.py's one-off counting build; never defined in the shipped library):
// LK iterations / window moves / tile restages / levels of a slot, reported through the stamp buffer's upper half
#ifdef ALVA_KLT_COUNT
#define KLT_COUNT(k) do { if (threadIdx.x == 0) sh.cnt[k]++; } while (0)
#else
#define KLT_COUNT(k) ((void) 0)
#endif

// cross-lane moves of the row layout (all without LDS, all without scalar registers): the value of the lane below within the 16-lane
// DPP row (0 into the row's first lane), and the two "last lane of a row into the next rows" broadcasts of gfx9 (lanes without a source
// read 0; only lanes 31 / 63 of the results are used)
__device__ __forceinline__ float dpp_shr1(float v) { return dpp_row_shr<1>(v); }
__device__ __forceinline__ float dpp_bcast15(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x142, 0xf, 0xf, true));
}
__device__ __forceinline__ float dpp_bcast31(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x143, 0xf, 0xf, true));
}
// (q0 + q2) + (q1 + q3) of the four vector-lane chains, whose ends sit on lanes 15 (q0), 31 (q2), 47 (q1), 63 (q3): valid on lane 63
__device__ __forceinline__ float rows_fold(float v) {
    const float t = v + dpp_bcast15(v);   // lane 31: q2 + q0, lane 63: q3 + q1 (IEEE addition is commutative: the reference's pairs)
    return t + dpp_bcast31(t);            // lane 63: (q3 + q1) + (q2 + q0)
}

// The 16 x 16 tile of the searched image with origin (tx0, ty0): lane -> row lane / 4, 4 consecutive columns; coordinates clamped to
// the padded level (the window itself never leaves it, lkpyramid.cpp:518-523, so clamped bytes are never used).  Load and LDS store
// are separate so that the level's first tile is in flight together with the template's gathers.
__device__ __forceinline__ uint4 tile_load(const LkLevel &J, int lane, int tx0, int ty0) {
    const int trow = lane >> 2, c0 = (lane & 3) * 4;
    uint4 px;
    if (tx0 >= -WIN && ty0 >= -WIN && tx0 + TW <= J.w + WIN && ty0 + TW <= J.h + WIN) {   // wave-uniform: the tile lies inside the padded level
        // one (unaligned) dword per lane from a uniform base + a small unsigned offset: no clamps, no 64-bit address arithmetic per byte
        const uint8_t *base = J.gray + (ptrdiff_t) ty0 * J.gpitch + tx0;
        uint32_t v;
        __builtin_memcpy(&v, base + (unsigned) (trow * J.gpitch + c0), 4);
        px.x = v & 0xffu; px.y = (v >> 8) & 0xffu; px.z = (v >> 16) & 0xffu; px.w = v >> 24;
        return px;
    }
    const int gy = min(max(ty0 + trow, -WIN), J.h + WIN - 1);
    const uint8_t *srow = J.gray + (ptrdiff_t) gy * J.gpitch;
    px.x = srow[min(max(tx0 + c0 + 0, -WIN), J.w + WIN - 1)];
    px.y = srow[min(max(tx0 + c0 + 1, -WIN), J.w + WIN - 1)];
    px.z = srow[min(max(tx0 + c0 + 2, -WIN), J.w + WIN - 1)];
    px.w = srow[min(max(tx0 + c0 + 3, -WIN), J.w + WIN - 1)];
    return px;
}
__device__ __forceinline__ void tile_store(LkShared &sh, int lane, const uint4 &px) {
    const int trow = lane >> 2, c0 = (lane & 3) * 4;
    __syncthreads();   // (one wave: no barrier instruction, only the order of the LDS accesses)
    *reinterpret_cast<uint4 *>(sh.jt + trow * TW + c0) = px;
    __syncthreads();
}

// One point, one pyramid level (lkpyramid.cpp:199-680).  All arguments and results are wave-uniform.
//
// ROW LAYOUT.  The reference's SIMD128 loop keeps, per window row, running float sums for vector lane q = 0..3 (window columns q and
// q + 4) and one scalar sum for column 8, and walks the rows top to bottom -- ten (b) / fifteen (A) SEQUENTIAL float chains over the
// nine rows.  Here window row y lives on lane 7 + y of every 16-lane DPP row, DPP row r owns vector lane q = {0, 2, 1, 3}[r] (both of
// its columns) and a share of column 8, and a chain is a scan along the lanes: acc = f + row_shr:1(acc), eight dependent v_add_f32_dpp
// -- lane 15 of the row ends up with exactly the reference's sum, the lanes below 7 hold zeros and feed the scan its initial 0.  The
// four chain ends then meet through row_bcast:15 / row_bcast:31 in the reference's (q0 + q2) + (q1 + q3) order.  No tap goes through
// LDS and nothing waits on an LDS round trip inside the iteration (round 6; before, the 81 taps sat one per lane and reached the
// chain lanes through LDS: two dependent LDS round trips per iteration, which for the launch's last, lonely slots -- the ones that
// run all 30 iterations on six levels and decide the kernel time -- were a third of the iteration).  A lone wave issues one
// instruction every four cycles whatever its type, so the iteration is also written for instruction COUNT: the template's part of
// (J - I) * dI is folded into one constant per lane, tile bytes and bounds are revisited only when the window's integer origin moved.
__device__ void lk_level(LkShared &sh, const LkLevel &I_, const LkLevel &J_, int level, int maxLevel, int maxCount, double epsilon,
                         float minEigThreshold, float ptx, float pty, float &nx, float &ny, int &status, float &err) {
    // Both levels' descriptors are fetched HERE, in one go: they sit in the kernel arguments behind a run-time level index, and left to
    // itself the compiler loads each field where it is first used -- five scalar loads in a row, each waited for, on the critical path of
    // every level of the launch's last slots.
    const LkLevel I = I_, J = J_;
    asm volatile("" ::"s"(I.gray), "s"(I.deriv), "s"(I.gpitch), "s"(I.dpitch), "s"(I.w), "s"(I.h), "s"(J.gray), "s"(J.gpitch), "s"(J.w), "s"(J.h));
    const int lane = threadIdx.x;
    const int row = lane >> 4, jl = lane & 15;
    const bool act = jl >= 7;                          // lanes 7..15 of a row carry window rows 0..8
    const int y = act ? jl - 7 : 0;
    const int q = ((row & 1) << 1) | (row >> 1);       // rows 0, 1, 2, 3 -> vector lanes 0, 2, 1, 3
    const float halfWin = (WIN - 1) * 0.5f;
    const float lscale = __int_as_float((127 - level) << 23);  // 2^-level, exact (1.0f / (float) (1 << level) without the division)
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
            KLT_COUNT(4);
        }
        return;
    }
    Weights wt = bilinear_weights(prevx - (float) ipx, prevy - (float) ipy);
    KLT_COUNT(3);
    // the tile of the searched image around the initial window: its loads go out now, beside the template's, and land in LDS after the
    // template arithmetic -- one memory latency per level instead of two
    nextx -= halfWin;
    nexty -= halfWin;
    int tx0 = 0x40000000, ty0 = 0x40000000;  // tile origin in image coordinates (invalid: staged on first use)
    uint4 tile0 = make_uint4(0, 0, 0, 0);
    {
        const int inx = __builtin_amdgcn_readfirstlane((int) floorf(nextx)), iny = __builtin_amdgcn_readfirstlane((int) floorf(nexty));
        if ((unsigned) (inx + WIN) < (unsigned) (J.w + WIN) && (unsigned) (iny + WIN) < (unsigned) (J.h + WIN)) {
            tx0 = inx - TR;
            ty0 = iny - TR;
            tile0 = tile_load(J, lane, tx0, ty0);
        }
    }

    // ---- template: this lane's three pixels of window row y -- columns q, q + 4 and 8 (lkpyramid.cpp:440-471) ----
    int tI[3], tIx[3], tIy[3];
    // (uniform 64-bit bases + small unsigned per-lane offsets: the loads take the scalar-base addressing form)
    const uint8_t *gbase = I.gray + (ptrdiff_t) ipy * I.gpitch + ipx;
    const uint8_t *dbase = reinterpret_cast<const uint8_t *>(I.deriv) + (ptrdiff_t) ipy * I.dpitch + (ptrdiff_t) ipx * 4;
    const uint8_t *gbase1 = gbase + I.gpitch, *dbase1 = dbase + I.dpitch;   // the row below, as bases of their own: ONE per-lane offset serves both rows
#pragma unroll
    for (int p = 0; p < 3; p++) {
        const int x = p == 0 ? q : (p == 1 ? q + 4 : 8);
        const unsigned go = (unsigned) (y * I.gpitch + x), dof = (unsigned) (y * I.dpitch + x * 4);
        const uint8_t *g0 = gbase + go, *g1 = gbase1 + go, *d0 = dbase + dof, *d1 = dbase1 + dof;
        const int ival = descale(bl_u8(g0[0], g0[1], g1[0], g1[1], wt), 9);
        const short2 d00 = *reinterpret_cast<const short2 *>(d0);
        const short2 d01 = *reinterpret_cast<const short2 *>(d0 + 4);
        const short2 d10 = *reinterpret_cast<const short2 *>(d1);
        const short2 d11 = *reinterpret_cast<const short2 *>(d1 + 4);
        const int ixval = descale(bl_i16(d00.x, d01.x, d10.x, d11.x, wt), 14);
        const int iyval = descale(bl_i16(d00.y, d01.y, d10.y, d11.y, wt), 14);
        tI[p] = ival;
        tIx[p] = act ? ixval : 0;   // the lanes below a row's chain contribute exact zeros
        tIy[p] = act ? iyval : 0;
    }
    // ---- A = sum of dI dI^T.  Per component (xx, xy, yy) and vector lane the reference runs acc = p(col q) + acc; acc = p(col q + 4) + acc
    // down the rows, from acc = 0; the scalar column's chain is acc = p(col 8) + acc.  Nine scan steps from zero reproduce that,
    // "+ 0" of the first row included.  The three scalar chains (one per component) share one scan: DPP rows 0, 1, 2.
    // (float) ix * (float) iy == (float) (ix * iy): both round the same exact product.
    float A[3];
    {
        float fx[3], fy[3];
#pragma unroll
        for (int p = 0; p < 3; p++) {
            fx[p] = (float) tIx[p];
            fy[p] = (float) tIy[p];
        }
        const float pa[3] = {fx[0] * fx[0], fx[0] * fy[0], fy[0] * fy[0]};
        const float pb[3] = {fx[1] * fx[1], fx[1] * fy[1], fy[1] * fy[1]};
        const float sxx = fx[2] * fx[2], sxy = fx[2] * fy[2], syy = fy[2] * fy[2];
        const float ps = row == 0 ? sxx : (row == 1 ? sxy : syy);   // (three products + two selects: a branch per row would cost more)
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, as = 0.f;
#pragma unroll
        for (int s = 0; s < WIN; s++) {
            a0 = pb[0] + (pa[0] + dpp_shr1(a0));
            a1 = pb[1] + (pa[1] + dpp_shr1(a1));
            a2 = pb[2] + (pa[2] + dpp_shr1(a2));
            as = ps + dpp_shr1(as);
        }
        const float v[3] = {rows_fold(a0), rows_fold(a1), rows_fold(a2)};
#pragma unroll
        for (int k = 0; k < 3; k++) {
            float sres = lane_bcast(as, 15 + 16 * k);
            sres += lane_bcast(v[k], 63);
            A[k] = sres * (1.f / (1 << 20));
        }
    }
    const float A11 = A[0], A12 = A[1], A22 = A[2];
    float D = A11 * A22 - A12 * A12;
    const float dA = A11 - A22;
    // sqrtf and '/' are IEEE correctly rounded here (-fhip-fp32-correctly-rounded-divide-sqrt); the __fsqrt_rn
    // intrinsic is NOT (it lowers to the native approximate v_sqrt_f32)
    const float minEig = ((A22 + A11) - sqrtf(dA * dA + (4.f * A12) * A12)) / (float) (2 * WIN * WIN);
    err = minEig;
    if (minEig < minEigThreshold || D < 1.1920928955078125e-07f) {
        if (level == 0) {
            status = 0;
            KLT_COUNT(5);
        }
        return;
    }
    D = 1.f / D;
    const float Ds = D * (1.f / (1 << 20));
    // ---- the iteration's per-lane constants.  (J - I) * Ix summed over the lane's two columns, an exact integer, is
    // Ja * Ixa + Jb * Ixb - (Ia * Ixa + Ib * Ixb): two v_mad_i32_i24 on top of a constant.  (J - I fits 14 bits -- the reference's
    // cast to short never truncates -- and the derivatives 13, so every product fits 26 bits and the sums 28.)  The scalar column's
    // chains run on DPP rows 3 (x) and 2 (y), where their ends meet the folded vector sums.
    const int ixa = tIx[0], iya = tIy[0], ixb = tIx[1], iyb = tIy[1];
    const int cx = -(__mul24(tI[0], ixa) + __mul24(tI[1], ixb)), cy = -(__mul24(tI[0], iya) + __mul24(tI[1], iyb));
    const int is = row == 3 ? tIx[2] : (row == 2 ? tIy[2] : 0);
    const int cs = -__mul24(tI[2], is);
    const int toff_a = y * TW + q, toff_s = y * TW + 8;
    if (tx0 != 0x40000000) tile_store(sh, lane, tile0);
    float pdx = 0.f, pdy = 0.f;
    float pfx = __int_as_float(0x7fc00000), pfy = pfx;   // floor of the previous iteration's window origin (NaN: differs from everything)
    uint32_t sa[4] = {0, 0, 0, 0}, sb[4] = {0, 0, 0, 0}, ss[4] = {0, 0, 0, 0};   // this lane's 3 x 4 tile pixels at that origin
    const float ef = (float) epsilon, ef_hi = ef * 1.00001f, ef_lo = ef * 0.99999f;
    for (int j = 0; j < maxCount; j++) {
        const float flx = floorf(nextx), fly = floorf(nexty);
        // bounds, tile and the twelve tile pixels only when the window's INTEGER origin moved: once the iteration is down to sub-pixel
        // steps -- most iterations -- the pixels are last iteration's and only the weights change
        KLT_COUNT(0);
        if (flx != pfx || fly != pfy) {
            KLT_COUNT(1);
            const int inx = __builtin_amdgcn_readfirstlane((int) flx), iny = __builtin_amdgcn_readfirstlane((int) fly);
            if ((unsigned) (inx + WIN) >= (unsigned) (J.w + WIN) || (unsigned) (iny + WIN) >= (unsigned) (J.h + WIN)) {   // inx < -WIN || inx >= J.w || ...
                if (level == 0) {
                    status = 0;
                    KLT_COUNT(6);
                }
                break;
            }
            if ((unsigned) (inx - tx0) > 2u * TR || (unsigned) (iny - ty0) > 2u * TR) {
                KLT_COUNT(2);
                tx0 = inx - TR;
                ty0 = iny - TR;
                tile_store(sh, lane, tile_load(J, lane, tx0, ty0));
            }
            const int tbase = (iny - ty0) * TW + (inx - tx0);
            const uint32_t *pa = sh.jt + toff_a + tbase, *ps = sh.jt + toff_s + tbase;
            sa[0] = pa[0]; sa[1] = pa[1]; sa[2] = pa[TW]; sa[3] = pa[TW + 1];
            sb[0] = pa[4]; sb[1] = pa[5]; sb[2] = pa[TW + 4]; sb[3] = pa[TW + 5];
            ss[0] = ps[0]; ss[1] = ps[1]; ss[2] = ps[TW]; ss[3] = ps[TW + 1];
            pfx = flx;
            pfy = fly;
        }
        // flx == (float) (int) flx here (a saturated conversion left the loop above): the reference's a = next - cvFloor(next)
        const WeightsB wb = bilinear_weights_i24(nextx - flx, nexty - fly);
        const int ja = tap_u8(sa, wb, 256), jb = tap_u8(sb, wb, 256), js = tap_u8(ss, wb, 256);
        const float fx = (float) mad_i24(jb, ixb, mad_i24(ja, ixa, cx)), fy = (float) mad_i24(jb, iyb, mad_i24(ja, iya, cy)),
                    fs = (float) mad_i24(js, is, cs);
        // b chains (lkpyramid.cpp:553-562, 628-646): bacc += (float) (column q + column q + 4) down the rows, from 0 (0 + f == f)
        float bx = fx, by = fy, bs = fs;
#pragma unroll
        for (int s = 1; s < WIN; s++) {
            bx = fx + dpp_shr1(bx);
            by = fy + dpp_shr1(by);
            bs = fs + dpp_shr1(bs);
        }
        // sres = scalar + ((q0 + q2) + (q1 + q3)) on lane 63: the x scalar chain ended there (row 3), the y one on lane 47 (row 2)
        const float wx = rows_fold(bx) + bs, wy = rows_fold(by) + dpp_bcast15(bs);
        // b = sres * 2^-20 and delta = (A12 b2 - A22 b1, A12 b1 - A11 b2) * D (lkpyramid.cpp:660-668); the power of two commutes with
        // every rounding on the way (nothing here is near the denormal range: a difference of two floats is zero or at least an ulp of
        // them), so it is applied once, to D
        const float b1 = lane_bcast(wx, 63), b2 = lane_bcast(wy, 63);
        const float dx = (A12 * b2 - A22 * b1) * Ds;
        const float dy = (A12 * b1 - A11 * b2) * Ds;
        nextx += dx;
        nexty += dy;
        nx = nextx + halfWin;
        ny = nexty + halfWin;
        // The two stopping rules of lkpyramid.cpp:671-679 are written in double there.  (1) |delta|^2 <= epsilon: the squares are exact
        // in double, their sum is rounded once; the same sum in float is within 3 float ulps of it, so the double arithmetic (half-rate
        // instructions on the critical path of every iteration) is only needed inside a band of +-1e-5 (relative) around epsilon.
        // (2) fabs((double) (dx + pdx)) < 0.01: the argument IS a float; no float lies between 0.01f (= 0.00999999977...) and 0.01, so
        // "< 0.01 in double" is "<= 0.01f in float", exactly.  Both are rare: one test in front of them keeps them off the common path.
        const float d2 = dx * dx + dy * dy;
        const bool osc = (int) (j > 0) & (int) (fabsf(dx + pdx) <= 0.01f) & (int) (fabsf(dy + pdy) <= 0.01f);
        if ((int) !(d2 > ef_hi) | (int) osc) {
            bool stop;
            if (d2 > ef_hi) stop = false;
            else if (d2 < ef_lo) stop = true;
            else stop = (double) dx * (double) dx + (double) dy * (double) dy <= epsilon;
            if (stop) break;
            if (osc) {
                nx -= dx * 0.5f;
                ny -= dy * 0.5f;
                break;
            }
        }
        pdx = dx;
        pdy = dy;
    }
}

// The verdict of FeatureTracker::fbKltTracking on one keypoint after its forward pass (feature_tracker.cpp:48-103): status,
// err (min eigenvalue) <= errThresh, inBorder(level-0 size), then the backward pass on level 0 and the forward-backward distance.
__device__ __forceinline__ int fbklt_gate(LkShared &sh, const LkPyr &P, const LkPyr &C, int maxCount, double epsilon, float errThresh,
                                          float fbDist, float ptx, float pty, float nx, float ny, int status, float err, int *why = nullptr) {
    int ok = status && !(err > errThresh);
    const float fw = (float) C.lv[0].w, fh = (float) C.lv[0].h;
    if (why) *why = !status ? 1 : (err > errThresh ? 2 : 0);
    if (ok && !(1.0f <= nx && nx < fw - 1.0f && 1.0f <= ny && ny < fh - 1.0f)) {
        ok = 0;
        if (why) *why = 3;
    }
    if (ok) {
        // backward LK on level 0 only, initial flow = the original point (:84-87)
        float bx = ptx, by = pty;
        int st2 = 1;
        float err2 = 0.f;
        lk_level(sh, C.lv[0], P.lv[0], 0, 0, maxCount, epsilon, 1e-4f, nx, ny, bx, by, st2, err2);
        if (!st2) {
            ok = 0;
            if (why) *why = 4;
        } else {
            const float ddx = ptx - bx, ddy = pty - by;
            const double nrm = sqrt((double) ddx * (double) ddx + (double) ddy * (double) ddy);  // cv::norm(Point2f), :103
            if (nrm > (double) fbDist) {
                ok = 0;
                if (why) *why = 5;
            }
        }
    }
    return ok;
}

// fbKltTracking of one keypoint given by value: (nx, ny) is the prior on entry and the tracked position on return
__device__ __forceinline__ int fbklt_value(LkShared &sh, const LkPyr &P, const LkPyr &C, int maxLevel, int maxCount, double epsilon,
                                           float errThresh, float fbDist, float ptx, float pty, float &nx, float &ny, int *why = nullptr) {
    int status = 1;
    float err = 0.f;
    for (int level = maxLevel; level >= 0; level--)
        lk_level(sh, P.lv[level], C.lv[level], level, maxLevel, maxCount, epsilon, 1e-4f, ptx, pty, nx, ny, status, err);
    return fbklt_gate(sh, P, C, maxCount, epsilon, errThresh, fbDist, ptx, pty, nx, ny, status, err, why);
}

// One keypoint through the whole pyramid (all arguments wave-uniform).
// mode 0: plain calcOpticalFlowPyrLK (next in/out, status, err).
// mode 1: FeatureTracker::fbKltTracking (prior in/out, status).
__device__ __forceinline__ void klt_point(LkShared &sh, const LkPyr &P, const LkPyr &C, int mode, int maxLevel, int maxCount, double epsilon,
                                          float errThresh, float fbDist, const float *__restrict__ pts, const float *init, float *nextio,
                                          uint8_t *__restrict__ status_out, float *__restrict__ err_out, int kp) {
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
    const int ok = fbklt_gate(sh, P, C, maxCount, epsilon, errThresh, fbDist, ptx, pty, nx, ny, status, err);
    if (threadIdx.x == 0) {
        nextio[2 * kp] = nx;
        nextio[2 * kp + 1] = ny;
        status_out[kp] = (uint8_t) ok;
    }
}

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
    klt_point(sh, P, C, mode, maxLevel, maxCount, epsilon, errThresh, fbDist, pts, init, nextio, status_out, err_out, kp);
}

// the same with the keypoint count in DEVICE memory (written by the kernel that built the list; the grid covers an upper bound)
__global__ void __launch_bounds__(64) k_klt_dn(LkPyr P, LkPyr C, int mode, int maxLevel, int maxCount, double epsilon, float errThresh,
                                               float fbDist, const float *__restrict__ pts, const float *init, float *nextio,
                                               uint8_t *__restrict__ status_out, const int *__restrict__ d_n) {
    __shared__ LkShared sh;
    const int per = gridDim.x >> 3;
    const int kp = (blockIdx.x & 7) * per + (blockIdx.x >> 3);
    if (kp >= *d_n) return;
    klt_point(sh, P, C, mode, maxLevel, maxCount, epsilon, errThresh, fbDist, pts, init, nextio, status_out, nullptr, kp);
}

// ---- the tracking step of one frame, one workgroup per slot of the frame container (track_slots.hpp) ----------------------------
// per-slot results of a finished slot: Frame::computeKeypoint (frame.cpp:105-113) for a tracked one, zeros for a lost one
__device__ __forceinline__ void track_slot_store(const TrackSlots &D, int i, int code, float nx, float ny) {
    float ux = 0.f, uy = 0.f;
    double bv[3] = {0., 0., 0.};
    if (code) {
        alva_undistort_dev(D.cam, nx, ny, ux, uy);
        alva_bearing_dev(D.invK, ux, uy, bv);
    } else {
        nx = 0.f;
        ny = 0.f;
    }
    if (threadIdx.x == 0) {   // device memory only: the compaction kernel publishes them to the host (track_slots.hpp)
        D.d_code[i] = (uint8_t) code;
        D.d_px[2 * i] = nx; D.d_px[2 * i + 1] = ny;
        D.d_unpx[2 * i] = ux; D.d_unpx[2 * i + 1] = uy;
        D.d_bv[3 * (size_t) i] = bv[0]; D.d_bv[3 * (size_t) i + 1] = bv[1]; D.d_bv[3 * (size_t) i + 2] = bv[2];
    }
}

// the frame's slot table (positions, 3-D flags, world points) from pinned host memory into device memory, 16 bytes per lane
__device__ __forceinline__ void track_stage_in_body(const TrackSlots &D, const int bx) {
    const size_t t = (size_t) bx * 256 + threadIdx.x;
    const size_t n = (size_t) D.n, q_px = (n * 8 + 15) / 16, q_3d = (n + 15) / 16, q_w = (n * 24 + 15) / 16;
    const uint4 *src;
    uint4 *dst;
    size_t k;
    if (t < q_px) {
        src = (const uint4 *) D.in_px; dst = (uint4 *) D.d_pts; k = t;
    } else if (t < q_px + q_3d) {
        src = (const uint4 *) D.in_is3d; dst = (uint4 *) D.d_is3d; k = t - q_px;
    } else if (t < q_px + q_3d + q_w) {
        src = (const uint4 *) D.in_wpt; dst = (uint4 *) D.d_wpt; k = t - q_px - q_3d;
    } else return;
    dst[k] = src[k];
}
__global__ void __launch_bounds__(256) k_track_stage_in(TrackSlots D) { track_stage_in_body(D, (int) blockIdx.x); }
ALVA_MULTI_KERNEL(MK_STAGE_IN, k_track_stage_in_multi, TrackSlots, dim3(256), 256, track_stage_in_body(A, bx));

// the slot's row of the frame's table: written by the host (or staged by k_track_stage_in), or CARRIED from the previous frame's buffers
// through the host's index (track_slots.hpp) -- then `writer` (one lane of the slot) also writes the row into this frame's table
__device__ __forceinline__ void track_slot_load(const TrackSlots &D, const int i, const bool writer, float &px, float &py, int &is3, double (&X)[3]) {
    if (D.carry) {
        const size_t s = (size_t) D.carry[i];
        px = D.p_px[2 * s]; py = D.p_px[2 * s + 1];
        is3 = D.p_is3d[s];
        X[0] = X[1] = X[2] = 0.;
        if (is3) {
            X[0] = D.p_wpt[3 * s]; X[1] = D.p_wpt[3 * s + 1]; X[2] = D.p_wpt[3 * s + 2];
        }
        if (writer) {
            D.d_pts[2 * (size_t) i] = px; D.d_pts[2 * (size_t) i + 1] = py;
            D.d_is3d[i] = (uint8_t) is3;
            D.d_wpt[3 * (size_t) i] = X[0]; D.d_wpt[3 * (size_t) i + 1] = X[1]; D.d_wpt[3 * (size_t) i + 2] = X[2];
        }
        return;
    }
    px = D.d_pts[2 * i]; py = D.d_pts[2 * i + 1];
    is3 = D.d_is3d[i];
    X[0] = X[1] = X[2] = 0.;
    if (is3) {
        X[0] = D.d_wpt[3 * (size_t) i]; X[1] = D.d_wpt[3 * (size_t) i + 1]; X[2] = D.d_wpt[3 * (size_t) i + 2];
    }
}

// gx = the launch's workgroups (a multiple of 8), bx = this one: slot i = the XCD-contiguous order of k_klt
__device__ __forceinline__ void track_klt_body(const LkPyr &P, const LkPyr &C, const TrackSlots &D, const int maxLevelPrior, const int maxLevelFull,
                                               const int maxCount, const double epsilon, const float errThresh, const float fbDist, const int bx,
                                               const int gx) {
    __shared__ LkShared sh;
    const unsigned long long t_begin = D.dbg ? wall_clock64() : 0ull;
#ifdef ALVA_KLT_COUNT
    if (threadIdx.x < 8) sh.cnt[threadIdx.x] = 0;
    __syncthreads();
#endif
    const int per = gx >> 3;
    const int i = (bx & 7) * per + (bx >> 3);
    if (i >= D.n) return;
    float px, py;
    int is3;
    double X[3];
    track_slot_load(D, i, threadIdx.x == 0, px, py, is3, X);
    bool from_prior = false;
    float nx = px, ny = py;
    if (D.use_prior && is3) {  // visual_frontend.cpp:125-152: project under the predicted pose, keep it if it is inside the image
        double pc[3];
        float qu, qv;
        alva_se3_apply_dev(D.q, D.t, X, pc);
        alva_project_dist_dev(D.cam, pc[0], pc[1], pc[2], qu, qv);
        from_prior = qu >= 0 && qv >= 0 && (double) qu < (double) D.width && (double) qv < (double) D.height;  // Frame::isInImage
        if (from_prior) {
            nx = qu;
            ny = qv;
        }
    }
    int why = 0, why2 = 0;
    const int ok = fbklt_value(sh, P, C, from_prior ? maxLevelPrior : maxLevelFull, maxCount, epsilon, errThresh, fbDist, px, py, nx, ny, D.dbg ? &why : nullptr);
    int code = ok ? (from_prior ? 1 : 2) : 0;
    if (from_prior && !ok) {  // full-pyramid retry from where the forward tracker left the keypoint (:185-190)
        // (s_setprio(3) here -- and s_setprio(2) for every full-pyramid slot -- measured in round 5: 65.6 vs 65.8 / 66.0 us, nothing: by the
        // time the slow slots are alone on their SIMDs the crowd has left anyway)
        const int ok2 = fbklt_value(sh, P, C, maxLevelFull, maxCount, epsilon, errThresh, fbDist, px, py, nx, ny, D.dbg ? &why2 : nullptr);
        code = ok2 ? 3 : 0;
    }
    if (threadIdx.x == 0) D.d_retried[i] = (uint8_t) (from_prior && !ok);
    track_slot_store(D, i, code, nx, ny);
    if (D.dbg && threadIdx.x == 0 && i < 8192)
        D.dbg[i] = ((wall_clock64() - t_begin) & 0xffffffffull) | ((unsigned long long) code << 32) | ((unsigned long long) (from_prior ? 1 : 0) << 36) |
                   ((unsigned long long) (from_prior && !ok ? 1 : 0) << 37) | ((unsigned long long) why << 40) | ((unsigned long long) why2 << 44);
#ifdef ALVA_KLT_COUNT
    if (D.dbg && threadIdx.x == 0 && i < 8192)
        D.dbg[8192 + i] = (unsigned long long) sh.cnt[0] | ((unsigned long long) sh.cnt[1] << 16) | ((unsigned long long) sh.cnt[2] << 32) | ((unsigned long long) (sh.cnt[3] & 0xff) << 48) |
                          ((unsigned long long) (sh.cnt[4] & 3) << 56) | ((unsigned long long) (sh.cnt[5] & 3) << 58) | ((unsigned long long) (sh.cnt[6] & 3) << 60);
#endif
    if (threadIdx.x == 0) {
        // ONE atomic per slot on a packed counter (track_slots.hpp) -- STRIPED: 2 600 atomics-with-return on one address from eight XCDs
        // queue at the memory side (k_fast_nms: 900 of them were that kernel's 50 us), so slot i arrives on stripe i % TRK_STRIPES, the
        // last arrival of a stripe adds the stripe's totals to the top counter, and the last of those learns that the launch is complete
        // and publishes the counts to the host right away: nothing but the atomics' own results is read, so no fence is needed.
        const unsigned long long add = (1ull << 48) | ((unsigned long long) (code != 0 && is3 != 0) << 32) |
                                       ((unsigned long long) (from_prior ? 1 : 0) << 16) | (unsigned long long) (from_prior && ok ? 1 : 0);
        const int st = i % TRK_STRIPES, in_stripe = (D.n - st + TRK_STRIPES - 1) / TRK_STRIPES;
        unsigned long long *stripes = reinterpret_cast<unsigned long long *>(D.cnt) + 16;
        const unsigned long long s_now = atomicAdd(stripes + st, add) + add;
        if ((int) (s_now >> 48) == in_stripe) {
            const unsigned long long now = atomicAdd(reinterpret_cast<unsigned long long *>(D.cnt) + 2, s_now) + s_now;
            if ((int) (now >> 48) == D.n) {
                const int n_pose = (int) ((now >> 32) & 0xffff), nA = (int) ((now >> 16) & 0xffff), good = (int) (now & 0xffff);
                // everything the host needs now -- the launch's sequence number, the size of the pose problem, p3pReq_ -- in ONE 8-byte
                // system-scope store: a single word is consistent by itself, so no system-scope fence (an L2 write-back, ~2.5 us at the very
                // end of the longest kernel of the frame) stands in front of it.  [seq : 32 | p3pReq_ : 1 | n_pose : 31] at o_hdr[10..11]
                const int req = nA > 0 && (double) good < 0.33 * (double) nA ? 1 : 0;
                const unsigned long long word = ((unsigned long long) (unsigned) D.seq << 32) | ((unsigned long long) req << 31) | (unsigned) n_pose;
                __hip_atomic_store(reinterpret_cast<unsigned long long *>(D.o_hdr + 10), word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            }
        }
    }
}
__global__ void __launch_bounds__(64) k_track_klt(LkPyr P, LkPyr C, TrackSlots D, int maxLevelPrior, int maxLevelFull, int maxCount,
                                                  double epsilon, float errThresh, float fbDist) {
    track_klt_body(P, C, D, maxLevelPrior, maxLevelFull, maxCount, epsilon, errThresh, fbDist, (int) blockIdx.x, (int) gridDim.x);
}
struct TrackKltArgs {
    LkPyr P, C;
    TrackSlots D;
    int maxLevelPrior, maxLevelFull, maxCount;
    float errThresh, fbDist;
    double epsilon;
};

__global__ void __launch_bounds__(64) k_track_klt_retry(LkPyr P, LkPyr C, TrackSlots D, int maxLevelFull, int maxCount, double epsilon,
                                                        float errThresh, float fbDist) {
    __shared__ LkShared sh;
    const int per = gridDim.x >> 3;
    const int i = (blockIdx.x & 7) * per + (blockIdx.x >> 3);
    if (i >= D.n) return;
    if (!D.d_retried[i]) return;   // launched only when the first launch's counts said p3pReq_ (the host read them from the header)
    const float px = D.d_pts[2 * i], py = D.d_pts[2 * i + 1];
    float nx = px, ny = py;                                          // :193-203: every prior falls back to the keypoint's own position
    const int ok = fbklt_value(sh, P, C, maxLevelFull, maxCount, epsilon, errThresh, fbDist, px, py, nx, ny);
    track_slot_store(D, i, ok ? 3 : 0, nx, ny);
}

// B cameras in one launch (fbKltTracking only): blockIdx.y = camera, each with its own pyramids, keypoint list and count.  The
// per-camera block sits in device memory; it is wave-uniform, so the compiler reads it with scalar loads.  A camera's keypoints
// keep the XCD-contiguous order of k_klt, so its pyramid still lives in the L2s of the XCDs that track into it.
struct KltBatchItem {
    LkPyr P, C;
    const float *pts, *init;
    float *out;
    uint8_t *status;
    int n, pad;
};

__global__ void __launch_bounds__(64) k_klt_batch(const KltBatchItem *__restrict__ items, int maxLevel, int maxCount, double epsilon,
                                                  float errThresh, float fbDist) {
    __shared__ LkShared sh;
    const KltBatchItem &it = items[blockIdx.y];
    const int per = gridDim.x >> 3;
    const int kp = (blockIdx.x & 7) * per + (blockIdx.x >> 3);
    if (kp >= it.n) return;
    klt_point(sh, it.P, it.C, 1, min(maxLevel, it.P.nlevels - 1), maxCount, epsilon, errThresh, fbDist, it.pts, it.init, it.out, it.status, nullptr,
              kp);
}

// ---- throughput variant: 64 / L keypoints per wavefront ------------------------------------------------------------------------
// One camera's launch of k_klt is a latency problem (the slowest keypoint sets the time), and a whole wave per keypoint is right
// for it.  A batch of cameras is a THROUGHPUT problem: every SIMD has a queue of waves, and what counts is instructions issued per
// keypoint.  With a wave per keypoint the window sums run on 15 / 10 of 64 lanes and every lane repeats the scalar update.  Here a
// keypoint owns a group of L lanes (its 81 taps take ceil(81 / L) rounds), so one issued instruction advances 64 / L keypoints;
// the arithmetic of a keypoint -- operand order of every float sum included -- is that of lk_level, so results are bit-identical.
// Groups of a wave are neighbouring keypoints of one camera; a group that has left the loop (converged, out of the image) idles
// until the last group of its wave is done.
template <int L>
struct LkSharedG {
    short2 dxy[64 / L][NPX + 3];
    int2 pxy[64 / L][NPX + 3];
    uint8_t jt[64 / L][TW * TW];
};

template <int L>
__device__ __forceinline__ float group_bcast(float v, int gbase, int k) {
    return __shfl(v, gbase + k, 64);
}

template <int L>
__device__ void lk_level_g(LkSharedG<L> &sh, const LkLevel &I, const LkLevel &J, int level, int maxLevel, int maxCount, double epsilon,
                           float minEigThreshold, bool live, float ptx, float pty, float &nx, float &ny, int &status, float &err) {
    constexpr int R = (NPX + L - 1) / L;
    const int lane = threadIdx.x, sub = lane & (L - 1), gbase = lane & ~(L - 1), g = lane / L;
    const float halfWin = (WIN - 1) * 0.5f;
    const float lscale = 1.0f / (float) (1 << level);
    float prevx = ptx * lscale, prevy = pty * lscale;
    float nextx, nexty;
    if (level == maxLevel) {
        nextx = nx * lscale;
        nexty = ny * lscale;
    } else {
        nextx = nx * 2.f;
        nexty = ny * 2.f;
    }
    if (live) {
        nx = nextx;
        ny = nexty;
    }
    prevx -= halfWin;
    prevy -= halfWin;
    const int ipx = (int) floorf(prevx), ipy = (int) floorf(prevy);
    if (live && (ipx < -WIN || ipx >= I.w || ipy < -WIN || ipy >= I.h)) {
        if (level == 0) {
            status = 0;
            err = 0.f;
        }
        live = false;
    }
    Weights wt = bilinear_weights(prevx - (float) ipx, prevy - (float) ipy);
    short rI[R], rIx[R], rIy[R];
    int toff[R];
#pragma unroll
    for (int r = 0; r < R; r++) {
        const int p = min(sub + L * r, NPX - 1);
        const int y = p / WIN, x = p - y * WIN;
        toff[r] = y * TW + x;
        rI[r] = rIx[r] = rIy[r] = 0;
        if (live) {
            const uint8_t *src = I.gray + (ptrdiff_t) (y + ipy) * I.gpitch + (x + ipx);
            const int ival = descale(bl_u8(src[0], src[1], src[I.gpitch], src[I.gpitch + 1], wt), 9);
            const uint8_t *drow = reinterpret_cast<const uint8_t *>(I.deriv) + (ptrdiff_t) (y + ipy) * I.dpitch + (ptrdiff_t) (x + ipx) * 4;
            const short2 d00 = *reinterpret_cast<const short2 *>(drow);
            const short2 d01 = *reinterpret_cast<const short2 *>(drow + 4);
            const short2 d10 = *reinterpret_cast<const short2 *>(drow + I.dpitch);
            const short2 d11 = *reinterpret_cast<const short2 *>(drow + I.dpitch + 4);
            const int ixval = descale(bl_i16(d00.x, d01.x, d10.x, d11.x, wt), 14);
            const int iyval = descale(bl_i16(d00.y, d01.y, d10.y, d11.y, wt), 14);
            rI[r] = (short) ival;
            rIx[r] = (short) ixval;
            rIy[r] = (short) iyval;
            sh.dxy[g][p] = make_short2((short) ixval, (short) iyval);
        }
    }
    __syncthreads();
    float acc = 0.f;
    {
        const int l15 = min(sub, 14), comp = l15 / 5, ch = l15 - comp * 5;
        const bool two = ch < 4;
        const int qa = two ? ch : 8, qb = two ? ch + 4 : 8;
#pragma unroll
        for (int y = 0; y < WIN; y++) {
            const short2 da = sh.dxy[g][y * WIN + qa], db = sh.dxy[g][y * WIN + qb];
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
        const float q0 = group_bcast<L>(acc, gbase, 5 * k + 0), q1 = group_bcast<L>(acc, gbase, 5 * k + 1),
                    q2 = group_bcast<L>(acc, gbase, 5 * k + 2), q3 = group_bcast<L>(acc, gbase, 5 * k + 3);
        float sres = group_bcast<L>(acc, gbase, 5 * k + 4);
        sres += (q0 + q2) + (q1 + q3);
        A[k] = sres * (1.f / (1 << 20));
    }
    const float A11 = A[0], A12 = A[1], A22 = A[2];
    float D = A11 * A22 - A12 * A12;
    const float dA = A11 - A22;
    const float minEig = ((A22 + A11) - sqrtf(dA * dA + (4.f * A12) * A12)) / (float) (2 * WIN * WIN);
    if (live) {
        err = minEig;
        if (minEig < minEigThreshold || D < 1.1920928955078125e-07f) {
            if (level == 0) status = 0;
            live = false;
        }
    }
    D = 1.f / D;
    nextx -= halfWin;
    nexty -= halfWin;
    float pdx = 0.f, pdy = 0.f;
    int tx0 = 0x40000000, ty0 = 0x40000000;
    bool run = live;
    for (int j = 0; j < maxCount; j++) {
        if (!__any(run)) break;
        const int inx = (int) floorf(nextx), iny = (int) floorf(nexty);
        if (run && (inx < -WIN || inx >= J.w || iny < -WIN || iny >= J.h)) {
            if (level == 0) status = 0;
            run = false;
        }
        wt = bilinear_weights(nextx - (float) inx, nexty - (float) iny);
        const bool restage = run && (inx < tx0 || inx > tx0 + 2 * TR || iny < ty0 || iny > ty0 + 2 * TR);
        if (__any(restage)) {
            __syncthreads();
            if (restage) {
                tx0 = inx - TR;
                ty0 = iny - TR;
                constexpr int PER = TW * TW / L;  // bytes of the 16 x 16 tile each lane of the group stages
#pragma unroll
                for (int k = 0; k < PER; k++) {
                    const int e = sub * PER + k, row = e / TW, col = e - row * TW;
                    const int gy = min(max(ty0 + row, -WIN), J.h + WIN - 1);
                    sh.jt[g][e] = J.gray[(ptrdiff_t) gy * J.gpitch + min(max(tx0 + col, -WIN), J.w + WIN - 1)];
                }
            }
        }
        __syncthreads();
        const int tbase = run ? (iny - ty0) * TW + (inx - tx0) : 0;
#pragma unroll
        for (int r = 0; r < R; r++) {
            const int p = min(sub + L * r, NPX - 1);
            const uint8_t *src = sh.jt[g] + toff[r] + tbase;
            const int jval = descale(bl_u8(src[0], src[1], src[TW], src[TW + 1], wt), 9);
            const int diff = (int) (short) (jval - rI[r]);
            sh.pxy[g][p] = make_int2(diff * rIx[r], diff * rIy[r]);
        }
        __syncthreads();
        float bacc = 0.f;
        {
            const int l10 = min(sub, 9), comp = l10 & 1, q = l10 >> 1;
            const bool two = q < 4;
            const int qa = two ? q : 8, qb = two ? q + 4 : 8;
            const int *src = reinterpret_cast<const int *>(sh.pxy[g]) + comp;
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
            const float s0 = group_bcast<L>(bacc, gbase, 0 + c) + group_bcast<L>(bacc, gbase, 4 + c);
            const float s2 = group_bcast<L>(bacc, gbase, 2 + c) + group_bcast<L>(bacc, gbase, 6 + c);
            float sres = group_bcast<L>(bacc, gbase, 8 + c);
            sres += (s0 + 0.f) + (s2 + 0.f);
            ib[c] = sres;
        }
        const float b1 = ib[0] * (1.f / (1 << 20)), b2 = ib[1] * (1.f / (1 << 20));
        const float dx = (A12 * b2 - A22 * b1) * D;
        const float dy = (A12 * b1 - A11 * b2) * D;
        if (run) {
            nextx += dx;
            nexty += dy;
            nx = nextx + halfWin;
            ny = nexty + halfWin;
            if ((double) dx * (double) dx + (double) dy * (double) dy <= epsilon) run = false;
            else if (j > 0 && fabs((double) (dx + pdx)) < 0.01 && fabs((double) (dy + pdy)) < 0.01) {
                nx -= dx * 0.5f;
                ny -= dy * 0.5f;
                run = false;
            }
            pdx = dx;
            pdy = dy;
        }
    }
    __syncthreads();
}

template <int L>
__global__ void __launch_bounds__(64) k_klt_batch_g(const KltBatchItem *__restrict__ items, int maxLevelArg, int maxCount, double epsilon,
                                                    float errThresh, float fbDist) {
    __shared__ LkSharedG<L> sh;
    constexpr int G = 64 / L;
    const KltBatchItem &it = items[blockIdx.y];
    const int per = gridDim.x >> 3;
    const int w = (blockIdx.x & 7) * per + (blockIdx.x >> 3);  // XCD-contiguous, as k_klt
    if (w * G >= it.n) return;
    const int kp = w * G + (int) threadIdx.x / L;
    const bool has = kp < it.n;
    const int kpc = has ? kp : it.n - 1;
    const int maxLevel = min(maxLevelArg, it.P.nlevels - 1);
    const float ptx = it.pts[2 * kpc], pty = it.pts[2 * kpc + 1];
    float nx = it.init[2 * kpc], ny = it.init[2 * kpc + 1];
    int status = 1;
    float err = 0.f;
    for (int level = maxLevel; level >= 0; level--)
        lk_level_g<L>(sh, it.P.lv[level], it.C.lv[level], level, maxLevel, maxCount, epsilon, 1e-4f, has, ptx, pty, nx, ny, status, err);
    // feature_tracker.cpp:48-73, as klt_point
    int ok = status && !(err > errThresh);
    const float fw = (float) it.C.lv[0].w, fh = (float) it.C.lv[0].h;
    ok = ok && (1.0f <= nx && nx < fw - 1.0f && 1.0f <= ny && ny < fh - 1.0f);
    float bx = ptx, by = pty;
    int st2 = 1;
    float err2 = 0.f;
    lk_level_g<L>(sh, it.C.lv[0], it.P.lv[0], 0, 0, maxCount, epsilon, 1e-4f, has && ok, nx, ny, bx, by, st2, err2);
    if (ok) {
        if (!st2) ok = 0;
        else {
            const float ddx = ptx - bx, ddy = pty - by;
            const double nrm = sqrt((double) ddx * (double) ddx + (double) ddy * (double) ddy);
            if (nrm > (double) fbDist) ok = 0;
        }
    }
    if (has && (threadIdx.x & (L - 1)) == 0) {
        it.out[2 * kp] = nx;
        it.out[2 * kp + 1] = ny;
        it.status[kp] = (uint8_t) ok;
    }
}

// ---- throughput variant 2: a lane per column pair -----------------------------------------------------------------------------
// PMC counters of the variants above on a 64-camera batch (profiles/): with a wave per keypoint the VALU is ~75 % busy
// and the LDS array ~90 %; sharing the wave between four keypoints moves the bound to the LDS (bpermute broadcasts, four tiles on
// the same banks).  Both costs come from the layout "pixel -> lane": every tap goes through LDS to reach the lane that owns its
// reference sum.  Here the layout follows the reference's SUMMATION instead: its SIMD128 loop keeps, for vector lane q = 0..3, the
// running sums of window columns q and q + 4, plus one scalar sum for column 8 (lkpyramid.cpp:553-562, 628-646).  A keypoint gets
// GL lanes (5 used): lane q walks down its two columns (18 taps, 9 for the scalar lane), holds its part of the template in
// registers, and adds its taps straight into its own float chains -- no tap ever goes through LDS.  Vertical neighbours share a row
// of the searched image, so a lane reads each of its 10 rows once.  Only the 16 x 16 tile of the searched image is in LDS.  The
// five partial sums of a keypoint meet in the reference's order through 8 bpermutes per iteration.  64 / GL keypoints per wave.
constexpr int TPITCH = TW * TW + 4;  // LDS bytes per tile; +4 rotates the banks between the keypoints of a wave

template <int GL>
struct LkSharedQ {
    uint8_t jt[64 / GL][TPITCH];
};

template <int GL>
__device__ void lk_level_q(LkSharedQ<GL> &sh, const LkLevel &I, const LkLevel &J, int level, int maxLevel, int maxCount, double epsilon,
                           float minEigThreshold, bool live, float ptx, float pty, float &nx, float &ny, int &status, float &err) {
    constexpr int NG = 64 / GL;
    const int lane = threadIdx.x, g = min(lane / GL, NG - 1), sub = lane - (lane / GL) * GL, gbase = g * GL;
    const int qq = min(sub, 4);
    const bool two = qq < 4;
    const int ca = two ? qq : 8, cb = two ? qq + 4 : 8;  // the scalar lane repeats column 8 and drops the copy
    const float halfWin = (WIN - 1) * 0.5f;
    const float lscale = 1.0f / (float) (1 << level);
    float prevx = ptx * lscale, prevy = pty * lscale;
    float nextx, nexty;
    if (level == maxLevel) {
        nextx = nx * lscale;
        nexty = ny * lscale;
    } else {
        nextx = nx * 2.f;
        nexty = ny * 2.f;
    }
    if (live) {
        nx = nextx;
        ny = nexty;
    }
    prevx -= halfWin;
    prevy -= halfWin;
    const int ipx = (int) floorf(prevx), ipy = (int) floorf(prevy);
    if (live && (ipx < -WIN || ipx >= I.w || ipy < -WIN || ipy >= I.h)) {
        if (level == 0) {
            status = 0;
            err = 0.f;
        }
        live = false;
    }
    Weights wt = bilinear_weights(prevx - (float) ipx, prevy - (float) ipy);
    // ---- template: this lane's two columns, all nine rows, in registers; A chains as it goes ----
    short tI[2 * WIN], tIx[2 * WIN], tIy[2 * WIN];
    float accA[3] = {0.f, 0.f, 0.f};
#pragma unroll
    for (int y = 0; y < WIN; y++) {
        float fx[2], fy[2];
#pragma unroll
        for (int h = 0; h < 2; h++) {
            const int x = h ? cb : ca;
            int ival = 0, ixval = 0, iyval = 0;
            if (live) {
                const uint8_t *src = I.gray + (ptrdiff_t) (y + ipy) * I.gpitch + (x + ipx);
                ival = descale(bl_u8(src[0], src[1], src[I.gpitch], src[I.gpitch + 1], wt), 9);
                const uint8_t *drow = reinterpret_cast<const uint8_t *>(I.deriv) + (ptrdiff_t) (y + ipy) * I.dpitch + (ptrdiff_t) (x + ipx) * 4;
                const short2 d00 = *reinterpret_cast<const short2 *>(drow);
                const short2 d01 = *reinterpret_cast<const short2 *>(drow + 4);
                const short2 d10 = *reinterpret_cast<const short2 *>(drow + I.dpitch);
                const short2 d11 = *reinterpret_cast<const short2 *>(drow + I.dpitch + 4);
                ixval = descale(bl_i16(d00.x, d01.x, d10.x, d11.x, wt), 14);
                iyval = descale(bl_i16(d00.y, d01.y, d10.y, d11.y, wt), 14);
            }
            tI[2 * y + h] = (short) ival;
            tIx[2 * y + h] = (short) ixval;
            tIy[2 * y + h] = (short) iyval;
            fx[h] = (float) (short) ixval;
            fy[h] = (float) (short) iyval;
        }
        // comp 0: Ix Ix, 1: Ix Iy, 2: Iy Iy -- the operand order of lk_level's chains
        const float pa[3] = {fx[0] * fx[0], fx[0] * fy[0], fy[0] * fy[0]};
        const float pb[3] = {fx[1] * fx[1], fx[1] * fy[1], fy[1] * fy[1]};
#pragma unroll
        for (int c = 0; c < 3; c++) {
            accA[c] = pa[c] + accA[c];
            accA[c] = (two ? pb[c] : 0.f) + accA[c];
        }
    }
    float A[3];
#pragma unroll
    for (int k = 0; k < 3; k++) {
        const float q0 = __shfl(accA[k], gbase + 0, 64), q1 = __shfl(accA[k], gbase + 1, 64), q2 = __shfl(accA[k], gbase + 2, 64),
                    q3 = __shfl(accA[k], gbase + 3, 64);
        float sres = __shfl(accA[k], gbase + 4, 64);
        sres += (q0 + q2) + (q1 + q3);
        A[k] = sres * (1.f / (1 << 20));
    }
    const float A11 = A[0], A12 = A[1], A22 = A[2];
    float D = A11 * A22 - A12 * A12;
    const float dA = A11 - A22;
    const float minEig = ((A22 + A11) - sqrtf(dA * dA + (4.f * A12) * A12)) / (float) (2 * WIN * WIN);
    if (live) {
        err = minEig;
        if (minEig < minEigThreshold || D < 1.1920928955078125e-07f) {
            if (level == 0) status = 0;
            live = false;
        }
    }
    D = 1.f / D;
    nextx -= halfWin;
    nexty -= halfWin;
    float pdx = 0.f, pdy = 0.f;
    int tx0 = 0x40000000, ty0 = 0x40000000;
    bool run = live && lane < NG * GL;
    uint8_t *tile = sh.jt[g];
    for (int j = 0; j < maxCount; j++) {
        if (!__any(run)) break;
        const int inx = (int) floorf(nextx), iny = (int) floorf(nexty);
        if (run && (inx < -WIN || inx >= J.w || iny < -WIN || iny >= J.h)) {
            if (level == 0) status = 0;
            run = false;
        }
        wt = bilinear_weights(nextx - (float) inx, nexty - (float) iny);
        const bool restage = run && (inx < tx0 || inx > tx0 + 2 * TR || iny < ty0 || iny > ty0 + 2 * TR);
        if (__any(restage)) {
            __syncthreads();
            if (restage) {
                tx0 = inx - TR;
                ty0 = iny - TR;
                if (tx0 >= -WIN && ty0 >= -WIN && tx0 + TW <= J.w + WIN && ty0 + TW <= J.h + WIN) {
                    // the whole tile lies inside the padded level: 64 dwords (the level's rows are not 4-byte aligned at tx0)
                    for (int e = sub; e < TW * TW / 4; e += GL) {
                        const int row = e >> 2, c4 = (e & 3) * 4;
                        uint32_t v;
                        __builtin_memcpy(&v, J.gray + (ptrdiff_t) (ty0 + row) * J.gpitch + (tx0 + c4), 4);
                        *reinterpret_cast<uint32_t *>(tile + row * TW + c4) = v;
                    }
                } else {
                    // coordinates clamped to the padded level, byte by byte (clamped bytes are never used, see lk_level)
                    for (int e = sub; e < TW * TW; e += GL) {
                        const int row = e / TW, col = e - row * TW;
                        const int gy = min(max(ty0 + row, -WIN), J.h + WIN - 1);
                        tile[e] = J.gray[(ptrdiff_t) gy * J.gpitch + min(max(tx0 + col, -WIN), J.w + WIN - 1)];
                    }
                }
            }
        }
        __syncthreads();
        const int tbase = run ? (iny - ty0) * TW + (inx - tx0) : 0;
        float bx = 0.f, by = 0.f;
        {
            const uint8_t *sa = tile + tbase + ca, *sb = tile + tbase + cb;
            // row r of the tile feeds the lower half of tap (r - 1) and the upper half of tap r
            int topa = __mul24(sa[0], wt.w00) + __mul24(sa[1], wt.w01), topb = __mul24(sb[0], wt.w00) + __mul24(sb[1], wt.w01);
#pragma unroll
            for (int y = 0; y < WIN; y++) {
                const int a0 = sa[(y + 1) * TW], a1 = sa[(y + 1) * TW + 1], b0 = sb[(y + 1) * TW], b1 = sb[(y + 1) * TW + 1];
                const int ja = descale(topa + (__mul24(a0, wt.w10) + __mul24(a1, wt.w11)), 9), jb = descale(topb + (__mul24(b0, wt.w10) + __mul24(b1, wt.w11)), 9);
                topa = __mul24(a0, wt.w00) + __mul24(a1, wt.w01);
                topb = __mul24(b0, wt.w00) + __mul24(b1, wt.w01);
                const int da = (int) (short) (ja - tI[2 * y]), db = (int) (short) (jb - tI[2 * y + 1]);
                const int vxa = da * tIx[2 * y], vya = da * tIy[2 * y];
                const int vxb = two ? db * tIx[2 * y + 1] : 0, vyb = two ? db * tIy[2 * y + 1] : 0;
                bx += (float) (vxa + vxb);
                by += (float) (vya + vyb);
            }
        }
        float ib[2];
#pragma unroll
        for (int c = 0; c < 2; c++) {
            const float v = c ? by : bx;
            const float s0 = __shfl(v, gbase + 0, 64) + __shfl(v, gbase + 2, 64);
            const float s2 = __shfl(v, gbase + 1, 64) + __shfl(v, gbase + 3, 64);
            float sres = __shfl(v, gbase + 4, 64);
            sres += (s0 + 0.f) + (s2 + 0.f);
            ib[c] = sres;
        }
        const float b1 = ib[0] * (1.f / (1 << 20)), b2 = ib[1] * (1.f / (1 << 20));
        const float dx = (A12 * b2 - A22 * b1) * D;
        const float dy = (A12 * b1 - A11 * b2) * D;
        if (run) {
            nextx += dx;
            nexty += dy;
            nx = nextx + halfWin;
            ny = nexty + halfWin;
            if ((double) dx * (double) dx + (double) dy * (double) dy <= epsilon) run = false;
            else if (j > 0 && fabs((double) (dx + pdx)) < 0.01 && fabs((double) (dy + pdy)) < 0.01) {
                nx -= dx * 0.5f;
                ny -= dy * 0.5f;
                run = false;
            }
            pdx = dx;
            pdy = dy;
        }
    }
    __syncthreads();
}

template <int GL>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(4, 4))) k_klt_batch_q(const KltBatchItem *__restrict__ items, int maxLevelArg, int maxCount, double epsilon,
                                                    float errThresh, float fbDist, int count, int per_cam) {
    __shared__ LkSharedQ<GL> sh;
    constexpr int NG = 64 / GL;
    // 8 cameras or more: a camera's keypoints on ONE XCD (both pyramids go through one L2; alva_xcd_item); fewer: the camera's keypoint
    // list in eight XCD-contiguous pieces, as k_klt
    int cam, w;
    if (count >= 8) {
        const AlvaXcdItem wi = alva_xcd_item(count, per_cam);
        cam = wi.cam;
        w = wi.item;
        if (cam >= count) return;
    } else {
        const int per = gridDim.x >> 3;
        cam = blockIdx.y;
        w = (blockIdx.x & 7) * per + (blockIdx.x >> 3);
    }
    const KltBatchItem &it = items[cam];
    if (w * NG >= it.n) return;
    const int grp = (int) threadIdx.x / GL;
    const int kp = w * NG + grp;
    const bool has = grp < NG && kp < it.n;
    const int kpc = has ? kp : it.n - 1;
    const int maxLevel = min(maxLevelArg, it.P.nlevels - 1);
    const float ptx = it.pts[2 * kpc], pty = it.pts[2 * kpc + 1];
    float nx = it.init[2 * kpc], ny = it.init[2 * kpc + 1];
    int status = 1;
    float err = 0.f;
    for (int level = maxLevel; level >= 0; level--)
        lk_level_q<GL>(sh, it.P.lv[level], it.C.lv[level], level, maxLevel, maxCount, epsilon, 1e-4f, has, ptx, pty, nx, ny, status, err);
    int ok = status && !(err > errThresh);
    const float fw = (float) it.C.lv[0].w, fh = (float) it.C.lv[0].h;
    ok = ok && (1.0f <= nx && nx < fw - 1.0f && 1.0f <= ny && ny < fh - 1.0f);
    float bx = ptx, by = pty;
    int st2 = 1;
    float err2 = 0.f;
    lk_level_q<GL>(sh, it.C.lv[0], it.P.lv[0], 0, 0, maxCount, epsilon, 1e-4f, has && ok, nx, ny, bx, by, st2, err2);
    if (ok) {
        if (!st2) ok = 0;
        else {
            const float ddx = ptx - bx, ddy = pty - by;
            const double nrm = sqrt((double) ddx * (double) ddx + (double) ddy * (double) ddy);
            if (nrm > (double) fbDist) ok = 0;
        }
    }
    if (has && (int) threadIdx.x == grp * GL) {
        it.out[2 * kp] = nx;
        it.out[2 * kp + 1] = ny;
        it.status[kp] = (uint8_t) ok;
    }
}

// ---- the tracking step of one frame in the throughput layout: GL lanes per slot, 64 / GL slots per wave (lk_level_q) -------------------
// What k_track_klt computes for a slot, for the sessions of a group (lane.hpp): S sessions' trackers in one launch fill the chip several
// times over, so the layout that finishes a SLOT soonest (a wave per slot: 2 300 short-lived waves, issue rate 0.18) loses to the one
// that does the most slots per wave-cycle (measured with 8 concurrent sessions: k_track_klt 74 -> 106 us each and every small kernel
// beside it 3 - 5x slower, the chip's wave slots held by trackers; k_klt_batch_q tracks a 2 120-keypoint camera in 11 us of a 64-camera
// launch).  Slots of a wave run the pyramid levels together; a slot tracked from its projection joins at level maxLevelPrior.  Bitwise
// equal to k_track_klt: the same arithmetic per slot (lk_level_q is lk_level's summation order, see above), the same gates.
// the two throughput layouts behind one face: slots per wave, lanes per slot, the LDS block, one pyramid level of every slot of the wave
template <int GL>
struct LayoutQ {   // lk_level_q: GL lanes per slot (5 used)
    static constexpr int LANES = GL, SLOTS = 64 / GL;
    typedef LkSharedQ<GL> Shared;
    static __device__ __forceinline__ void level(Shared &sh, const LkLevel &I, const LkLevel &J, int level, int maxLevel, int maxCount, double epsilon,
                                                 bool live, float ptx, float pty, float &nx, float &ny, int &status, float &err) {
        lk_level_q<GL>(sh, I, J, level, maxLevel, maxCount, epsilon, 1e-4f, live, ptx, pty, nx, ny, status, err);
    }
};
template <int L>
struct LayoutG {   // lk_level_g: L = 32 or 16 lanes per slot, pixel -> lane like lk_level
    static constexpr int LANES = L, SLOTS = 64 / L;
    typedef LkSharedG<L> Shared;
    static __device__ __forceinline__ void level(Shared &sh, const LkLevel &I, const LkLevel &J, int level, int maxLevel, int maxCount, double epsilon,
                                                 bool live, float ptx, float pty, float &nx, float &ny, int &status, float &err) {
        lk_level_g<L>(sh, I, J, level, maxLevel, maxCount, epsilon, 1e-4f, live, ptx, pty, nx, ny, status, err);
    }
};

template <class LAY>
__device__ __forceinline__ int fbklt_value_w(typename LAY::Shared &sh, const LkPyr &P, const LkPyr &C, const int myMax, const int top, const int maxCount,
                                             const double epsilon, const float errThresh, const float fbDist, const bool has, const float ptx,
                                             const float pty, float &nx, float &ny) {
    int status = 1;
    float err = 0.f;
    for (int level = top; level >= 0; level--)   // top = the wave's highest starting level (wave-uniform)
        LAY::level(sh, P.lv[level], C.lv[level], level, myMax, maxCount, epsilon, has && level <= myMax, ptx, pty, nx, ny, status, err);
    int ok = status && !(err > errThresh);
    const float fw = (float) C.lv[0].w, fh = (float) C.lv[0].h;
    ok = ok && (1.0f <= nx && nx < fw - 1.0f && 1.0f <= ny && ny < fh - 1.0f);
    float bx = ptx, by = pty;
    int st2 = 1;
    float err2 = 0.f;
    LAY::level(sh, C.lv[0], P.lv[0], 0, 0, maxCount, epsilon, has && ok, nx, ny, bx, by, st2, err2);
    if (ok) {
        if (!st2) ok = 0;
        else {
            const float ddx = ptx - bx, ddy = pty - by;
            const double nrm = sqrt((double) ddx * (double) ddx + (double) ddy * (double) ddy);
            if (nrm > (double) fbDist) ok = 0;
        }
    }
    return has ? ok : 0;
}

template <class LAY>
__device__ __forceinline__ void track_klt_w_body(const LkPyr &P, const LkPyr &C, const TrackSlots &D, const int maxLevelPrior, const int maxLevelFull,
                                                 const int maxCount, const double epsilon, const float errThresh, const float fbDist, const int bx,
                                                 const int gx) {
    __shared__ typename LAY::Shared sh;
    constexpr int NG = LAY::SLOTS, GL = LAY::LANES;
    const unsigned long long t_begin = D.dbg ? wall_clock64() : 0ull;
    const int per = gx >> 3;
    const int w = (bx & 7) * per + (bx >> 3);   // the XCD-contiguous order of k_klt, in units of NG slots
    if (w * NG >= D.n) return;
    const int grp = (int) threadIdx.x / GL;
    const int i = w * NG + grp;
    const bool has = grp < NG && i < D.n;
    const bool leader = has && (int) threadIdx.x == grp * GL;
    const int ic = has ? i : D.n - 1;
    float px, py;
    int is3;
    double X[3];
    track_slot_load(D, ic, leader, px, py, is3, X);
    bool from_prior = false;
    float nx = px, ny = py;
    if (D.use_prior && is3) {  // visual_frontend.cpp:125-152: project under the predicted pose, keep it if it is inside the image
        double pc[3];
        float qu, qv;
        alva_se3_apply_dev(D.q, D.t, X, pc);
        alva_project_dist_dev(D.cam, pc[0], pc[1], pc[2], qu, qv);
        from_prior = qu >= 0 && qv >= 0 && (double) qu < (double) D.width && (double) qv < (double) D.height;  // Frame::isInImage
        if (from_prior) {
            nx = qu;
            ny = qv;
        }
    }
    const int myMax = from_prior ? maxLevelPrior : maxLevelFull;
    const int top = __any(has && !from_prior) ? maxLevelFull : maxLevelPrior;
    const int ok = fbklt_value_w<LAY>(sh, P, C, myMax, top, maxCount, epsilon, errThresh, fbDist, has, px, py, nx, ny);
    int code = ok ? (from_prior ? 1 : 2) : 0;
    const bool retry = has && from_prior && !ok;
    if (__any(retry)) {  // full-pyramid retry from where the forward tracker left the keypoint (:185-190)
        float rx = nx, ry = ny;
        const int ok2 = fbklt_value_w<LAY>(sh, P, C, maxLevelFull, maxLevelFull, maxCount, epsilon, errThresh, fbDist, retry, px, py, rx, ry);
        if (retry) {
            nx = rx;
            ny = ry;
            code = ok2 ? 3 : 0;
        }
    }
    // per-slot results: Frame::computeKeypoint for a tracked slot, zeros for a lost one (track_slot_store, by the slot's first lane)
    float ux = 0.f, uy = 0.f;
    double bv[3] = {0., 0., 0.};
    if (code) {
        alva_undistort_dev(D.cam, nx, ny, ux, uy);
        alva_bearing_dev(D.invK, ux, uy, bv);
    } else {
        nx = 0.f;
        ny = 0.f;
    }
    if (leader) {
        D.d_retried[i] = (uint8_t) retry;
        D.d_code[i] = (uint8_t) code;
        D.d_px[2 * i] = nx; D.d_px[2 * i + 1] = ny;
        D.d_unpx[2 * i] = ux; D.d_unpx[2 * i + 1] = uy;
        D.d_bv[3 * (size_t) i] = bv[0]; D.d_bv[3 * (size_t) i + 1] = bv[1]; D.d_bv[3 * (size_t) i + 2] = bv[2];
        if (D.dbg && i < 16384)
            D.dbg[i] = ((wall_clock64() - t_begin) & 0xffffffffull) | ((unsigned long long) code << 32) | ((unsigned long long) (from_prior ? 1 : 0) << 36) |
                       ((unsigned long long) (retry ? 1 : 0) << 37);
    }
    // ONE atomic per wave on the packed counter (track_slots.hpp): the wave's slots, tracked 3-D slots, slots from the projection, successes of those
    const unsigned long long b_has = __ballot(leader), b_pose = __ballot(leader && code != 0 && is3 != 0), b_prior = __ballot(leader && from_prior),
                             b_good = __ballot(leader && from_prior && ok);
    if (threadIdx.x == 0) {
        const unsigned long long add = ((unsigned long long) __popcll(b_has) << 48) | ((unsigned long long) __popcll(b_pose) << 32) |
                                       ((unsigned long long) __popcll(b_prior) << 16) | (unsigned long long) __popcll(b_good);
        const unsigned long long before = atomicAdd(reinterpret_cast<unsigned long long *>(D.cnt) + 2, add);
        const unsigned long long now = before + add;
        if ((int) (now >> 48) == D.n) {   // the launch's last wave of this session publishes the counts (see k_track_klt)
            const int n_pose = (int) ((now >> 32) & 0xffff), nA = (int) ((now >> 16) & 0xffff), good = (int) (now & 0xffff);
            const int req = nA > 0 && (double) good < 0.33 * (double) nA ? 1 : 0;
            const unsigned long long word = ((unsigned long long) (unsigned) D.seq << 32) | ((unsigned long long) req << 31) | (unsigned) n_pose;
            __hip_atomic_store(reinterpret_cast<unsigned long long *>(D.o_hdr + 10), word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}
// the single session's launch in a throughput layout (ALVA_TRACK_KLT_LANES = 32 | 16 | 5: A/B against the wave-per-slot k_track_klt)
template <class LAY>
__global__ void __launch_bounds__(64) k_track_klt_w(LkPyr P, LkPyr C, TrackSlots D, int maxLevelPrior, int maxLevelFull, int maxCount, double epsilon,
                                                    float errThresh, float fbDist) {
    track_klt_w_body<LAY>(P, C, D, maxLevelPrior, maxLevelFull, maxCount, epsilon, errThresh, fbDist, (int) blockIdx.x, (int) gridDim.x);
}
// the lane's tracker (lane.hpp): 12 slots per wave by default; ALVA_LANE_KLT_LANES = 32 | 16 | 64 registers another layout instead (A/B:
// what finishes a slot soonest -- a wave per slot, 81 us for one session -- against what does the most slots per wave-cycle)
static int lane_klt_lanes() {
    static const int v = [] {
        const char *e = getenv("ALVA_LANE_KLT_LANES");
        const int l = e ? atoi(e) : 5;
        return (l == 32 || l == 16 || l == 64) ? l : 5;
    }();
    return v;
}
static int lane_klt_slots_per_wave() { return lane_klt_lanes() == 64 ? 1 : (lane_klt_lanes() == 32 ? 2 : (lane_klt_lanes() == 16 ? 4 : 12)); }
ALVA_MULTI_KERNEL_ATTR_IF(lane_klt_lanes() == 5, MK_TRACK_KLT, k_track_klt_q_multi, TrackKltArgs, dim3(64), __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(4, 4))),
                          track_klt_w_body<LayoutQ<5>>(A.P, A.C, A.D, A.maxLevelPrior, A.maxLevelFull, A.maxCount, A.epsilon, A.errThresh, A.fbDist, bx, (int) gx));
ALVA_MULTI_KERNEL_ATTR_IF(lane_klt_lanes() == 32, MK_TRACK_KLT, k_track_klt_g32_multi, TrackKltArgs, dim3(64), __launch_bounds__(64),
                          track_klt_w_body<LayoutG<32>>(A.P, A.C, A.D, A.maxLevelPrior, A.maxLevelFull, A.maxCount, A.epsilon, A.errThresh, A.fbDist, bx, (int) gx));
ALVA_MULTI_KERNEL_ATTR_IF(lane_klt_lanes() == 16, MK_TRACK_KLT, k_track_klt_g16_multi, TrackKltArgs, dim3(64), __launch_bounds__(64),
                          track_klt_w_body<LayoutG<16>>(A.P, A.C, A.D, A.maxLevelPrior, A.maxLevelFull, A.maxCount, A.epsilon, A.errThresh, A.fbDist, bx, (int) gx));
ALVA_MULTI_KERNEL_ATTR_IF(lane_klt_lanes() == 64, MK_TRACK_KLT, k_track_klt_multi, TrackKltArgs, dim3(64), __launch_bounds__(64),
                          track_klt_body(A.P, A.C, A.D, A.maxLevelPrior, A.maxLevelFull, A.maxCount, A.epsilon, A.errThresh, A.fbDist, bx, (int) gx));

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

// fbKltTracking of a keypoint list whose length lives in device memory (internal: the fused tracking step of stages_hip.hip);
// n_max bounds the grid.  Enqueue only.
int alva_fbklt_track_dn(alva_ctx *ctx, const alva_pyramid *prev, const alva_pyramid *curr, int num_levels, float err_thresh, float fb_dist,
                        int max_iters, float eps, const float *d_pts, const float *d_prior_in, float *d_out, uint8_t *d_status, const int *d_n,
                        int n_max) {
    ALVA_ARG(ctx && prev && curr && n_max >= 0 && num_levels >= 0);
    if (n_max == 0) return ALVA_OK;
    ALVA_ARG(d_pts && d_prior_in && d_out && d_status && d_n);
    ALVA_ARG(prev->win == WIN && curr->win == WIN);
    ALVA_ARG(prev->nlevels == curr->nlevels && prev->lv[0].w == curr->lv[0].w && prev->lv[0].h == curr->lv[0].h);
    LkPyr P, C;
    fill_pyr(prev, P);
    fill_pyr(curr, C);
    int maxLevel = num_levels;
    if (prev->nlevels - 1 < maxLevel) maxLevel = prev->nlevels - 1;
    const int maxCount = max_iters < 0 ? 0 : (max_iters > 100 ? 100 : max_iters);
    double epsilon = (double) eps;
    epsilon = epsilon < 0 ? 0 : (epsilon > 10 ? 10 : epsilon);
    epsilon *= epsilon;
    hipLaunchKernelGGL(k_klt_dn, dim3(8 * alva_divup(n_max, 8)), dim3(64), 0, ctx->stream, P, C, 1, maxLevel, maxCount, epsilon, err_thresh, fb_dist,
                       d_pts, d_prior_in, d_out, d_status, d_n);
    ALVA_LAUNCH_CHECK();
    return ALVA_OK;
}

int alva_track_slots_klt(alva_ctx *ctx, const alva_pyramid *prev, const alva_pyramid *curr, const TrackSlots &D, int levels_prior, int levels_full,
                         float err_thresh, float fb_dist, int max_iters, float eps, int retry) {
    ALVA_ARG(ctx && prev && curr && D.n >= 0 && levels_prior >= 0 && levels_full >= 0);
    ALVA_ARG(D.n < 65536);   // the packed counter of the tracker launch gives 16 bits to each count (track_slots.hpp)
    if (D.n == 0) return ALVA_OK;
    ALVA_ARG(prev->win == WIN && curr->win == WIN);
    ALVA_ARG(prev->nlevels == curr->nlevels && prev->lv[0].w == curr->lv[0].w && prev->lv[0].h == curr->lv[0].h);
    LkPyr P, C;
    fill_pyr(prev, P);
    fill_pyr(curr, C);
    const int top = prev->nlevels - 1;
    const int lp = levels_prior < top ? levels_prior : top, lf = levels_full < top ? levels_full : top;
    const int maxCount = max_iters < 0 ? 0 : (max_iters > 100 ? 100 : max_iters);
    double epsilon = (double) eps;
    epsilon = epsilon < 0 ? 0 : (epsilon > 10 ? 10 : epsilon);
    epsilon *= epsilon;
    const dim3 grid(8 * alva_divup(D.n, 8));
    if (!retry) {
        if (D.in_px) {   // a slot table in pinned host memory: one coalesced pass copies it (null: the host wrote it into d_pts / d_is3d / d_wpt)
            const size_t n = (size_t) D.n, quads = (n * 8 + 15) / 16 + (n + 15) / 16 + (n * 24 + 15) / 16;
            const unsigned g_in = (unsigned) ((quads + 255) / 256);
            const bool in_lane = alva_lane_defer(MK_STAGE_IN, ctx, g_in, 0, &D, sizeof(D));
            if (!in_lane) hipLaunchKernelGGL(k_track_stage_in, dim3(g_in), dim3(256), 0, ctx->stream, D);
        }
        const TrackKltArgs KA{P, C, D, lp, lf, maxCount, err_thresh, fb_dist, epsilon};
        if (alva_lane_defer(MK_TRACK_KLT, ctx, (unsigned) (8 * alva_divup(alva_divup(D.n, lane_klt_slots_per_wave()), 8)), 0, &KA, sizeof(KA))) return ALVA_OK;
    }
    static const int lanes = getenv("ALVA_TRACK_KLT_LANES") ? atoi(getenv("ALVA_TRACK_KLT_LANES")) : 64;
    if (!retry && lanes != 64) {
        const int per_wave = lanes == 32 ? 2 : (lanes == 16 ? 4 : 12);
        const dim3 gw((unsigned) (8 * alva_divup(alva_divup(D.n, per_wave), 8)));
        if (lanes == 32) hipLaunchKernelGGL((k_track_klt_w<LayoutG<32>>), gw, dim3(64), 0, ctx->stream, P, C, D, lp, lf, maxCount, epsilon, err_thresh, fb_dist);
        else if (lanes == 16) hipLaunchKernelGGL((k_track_klt_w<LayoutG<16>>), gw, dim3(64), 0, ctx->stream, P, C, D, lp, lf, maxCount, epsilon, err_thresh, fb_dist);
        else hipLaunchKernelGGL((k_track_klt_w<LayoutQ<5>>), gw, dim3(64), 0, ctx->stream, P, C, D, lp, lf, maxCount, epsilon, err_thresh, fb_dist);
        ALVA_LAUNCH_CHECK();
        return ALVA_OK;
    }
    if (!retry) hipLaunchKernelGGL(k_track_klt, grid, dim3(64), 0, ctx->stream, P, C, D, lp, lf, maxCount, epsilon, err_thresh, fb_dist);
    else hipLaunchKernelGGL(k_track_klt_retry, grid, dim3(64), 0, ctx->stream, P, C, D, lf, maxCount, epsilon, err_thresh, fb_dist);
    ALVA_LAUNCH_CHECK();
    return ALVA_OK;
}

// ---- batch of cameras (internal: track_batch.hip) -------------------------------------------------------------------------
size_t alva_klt_batch_item_size() { return sizeof(KltBatchItem); }

int alva_klt_batch_item_fill(void *dst, const alva_pyramid *prev, const alva_pyramid *curr, const float *d_pts, const float *d_init,
                             float *d_out, uint8_t *d_status, int n) {
    ALVA_ARG(dst && prev && curr && n >= 0 && (n == 0 || (d_pts && d_init && d_out && d_status)));
    ALVA_ARG(prev->win == WIN && curr->win == WIN);
    ALVA_ARG(prev->nlevels == curr->nlevels && prev->lv[0].w == curr->lv[0].w && prev->lv[0].h == curr->lv[0].h);
    KltBatchItem it{};
    fill_pyr(prev, it.P);
    fill_pyr(curr, it.C);
    it.pts = d_pts;
    it.init = d_init;
    it.out = d_out;
    it.status = d_status;
    it.n = n;
    memcpy(dst, &it, sizeof(it));
    return ALVA_OK;
}

// fbKltTracking of `count` cameras, items already in device memory; n_max = the largest keypoint count among them
// lanes = lanes of a wavefront per keypoint: 64 (k_klt's layout), 32 or 16 (2 / 4 keypoints per wave, see lk_level_g)
int alva_fbklt_track_batch_enqueue(alva_ctx *ctx, const void *d_items, int count, int n_max, int num_levels, float err_thresh, float fb_dist,
                                   int max_iters, float eps, int lanes) {
    ALVA_ARG(ctx && d_items && count > 0 && count <= 65535 && n_max >= 0 && num_levels >= 0 && (lanes == 64 || lanes == 32 || lanes == 16 || lanes == 8 || lanes == 5));
    if (n_max == 0) return ALVA_OK;
    const int maxCount = max_iters < 0 ? 0 : (max_iters > 100 ? 100 : max_iters);
    double epsilon = (double) eps;
    epsilon = epsilon < 0 ? 0 : (epsilon > 10 ? 10 : epsilon);
    epsilon *= epsilon;
    const KltBatchItem *items = (const KltBatchItem *) d_items;
    if (lanes == 64)
        hipLaunchKernelGGL(k_klt_batch, dim3(8 * alva_divup(n_max, 8), count), dim3(64), 0, ctx->stream, items, num_levels, maxCount, epsilon, err_thresh,
                           fb_dist);
    else if (lanes == 5 || lanes == 8) {
        const int per_cam = 8 * alva_divup(alva_divup(n_max, lanes == 5 ? 12 : 8), 8);
        const dim3 grid = count >= 8 ? dim3(alva_xcd_grid(count, per_cam)) : dim3(per_cam, count);
        if (lanes == 5)
            hipLaunchKernelGGL(k_klt_batch_q<5>, grid, dim3(64), 0, ctx->stream, items, num_levels, maxCount, epsilon, err_thresh, fb_dist, count, per_cam);
        else
            hipLaunchKernelGGL(k_klt_batch_q<8>, grid, dim3(64), 0, ctx->stream, items, num_levels, maxCount, epsilon, err_thresh, fb_dist, count, per_cam);
    }
    else if (lanes == 32)
        hipLaunchKernelGGL(k_klt_batch_g<32>, dim3(8 * alva_divup(alva_divup(n_max, 2), 8), count), dim3(64), 0, ctx->stream, items, num_levels, maxCount,
                           epsilon, err_thresh, fb_dist);
    else
        hipLaunchKernelGGL(k_klt_batch_g<16>, dim3(8 * alva_divup(alva_divup(n_max, 4), 8), count), dim3(64), 0, ctx->stream, items, num_levels, maxCount,
                           epsilon, err_thresh, fb_dist);
    ALVA_LAUNCH_CHECK();
    return ALVA_OK;
}

// fbKltTracking of `count` independent (previous, current) pyramid pairs in ONE launch -- the batched form of alva_fbklt_track
// (pyramids may differ in size; d_prior[c] in/out as there).  Enqueue-only; the argument blocks are staged from pageable memory.
extern "C" int alva_fbklt_track_batch(alva_ctx *ctx, const alva_pyramid *const *prev, const alva_pyramid *const *curr, int count, int num_levels,
                                      float err_thresh, float fb_dist, int max_iters, float eps, const float *const *d_pts,
                                      float *const *d_prior, uint8_t *const *d_status, const int *n, int lanes_per_keypoint) {
    ALVA_ARG(ctx && prev && curr && count > 0 && count <= 65535 && d_pts && d_prior && d_status && n);
    std::vector<KltBatchItem> items((size_t) count);
    int n_max = 0;
    for (int c = 0; c < count; c++) {
        int rc = alva_klt_batch_item_fill(&items[(size_t) c], prev[c], curr[c], d_pts[c], d_prior[c], d_prior[c], d_status[c], n[c]);
        if (rc) return rc;
        n_max = n[c] > n_max ? n[c] : n_max;
    }
    void *dev = nullptr;
    int rc = alva_ctx_scratch(ctx, 11, items.size() * sizeof(KltBatchItem), &dev);
    if (rc) return rc;
    ALVA_HIP(hipMemcpyAsync(dev, items.data(), items.size() * sizeof(KltBatchItem), hipMemcpyHostToDevice, ctx->stream));
    return alva_fbklt_track_batch_enqueue(ctx, dev, count, n_max, num_levels, err_thresh, fb_dist, max_iters, eps, lanes_per_keypoint);
}
