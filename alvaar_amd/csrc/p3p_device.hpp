// Device side of a8 (P3P-LMedS): the Kneip solver, the radix select and the hypothesis workgroup, shared by p3p.hip (its kernels) and
// pnp.hip (the fused P3P -> PnP launch of the single-session tracking chain).  See p3p.hip for the references.
// This header must be included BEFORE any "#pragma clang fp contract(fast)": P3P's results are compared with the reference's to 1e-8 and
// its outlier sets exactly, under the build's -ffp-contract=off.
#pragma once
#include "common.hpp"
#include "pose_internal.hpp"
#include <cmath>

struct P3pArgs {
    const double *bv, *wpt;
    const int *samples;     // H x 4 (pinned host memory, read once)
    int n, H, max_iters;
    double threshold;
    double *models;         // H x 12
    int *valid;             // H
    double *penalty;        // H
    int *counter;           // device-scope arrival counter (zero between launches)
    P3pSelectOut *out;
    uint8_t *inlier;
    unsigned long long *dbg;   // phase stamps (alva_kstamp_buffer) or null
    unsigned long long *masks; // one launch (MODE 0): H x ceil(n / 64) words, hypothesis h's inlier mask (bit i = score(i) <= threshold)
};
// The single-problem launch carries its samples IN the kernel arguments when they fit (the hypothesis then starts with one scalar load from
// the argument segment instead of a pointer chase into pinned host memory over the bus)
constexpr int P3P_INLINE_H = 192;
struct P3pInlineSamples {
    int v[4 * P3P_INLINE_H];
};

namespace {

using SelectOut = P3pSelectOut;

struct cplx {
    double re, im;
};
__device__ __forceinline__ cplx C_(double r, double i) { return cplx{r, i}; }
__device__ __forceinline__ cplx cadd(cplx a, cplx b) { return C_(a.re + b.re, a.im + b.im); }
__device__ __forceinline__ cplx csub(cplx a, cplx b) { return C_(a.re - b.re, a.im - b.im); }
__device__ __forceinline__ cplx cscale(cplx a, double s) { return C_(a.re * s, a.im * s); }
__device__ __forceinline__ cplx cdivc(cplx a, cplx b) {
    double den = b.re * b.re + b.im * b.im;
    return C_((a.re * b.re + a.im * b.im) / den, (a.im * b.re - a.re * b.im) / den);
}
__device__ cplx csqrt_(cplx z) {
    double m = hypot(z.re, z.im);
    if (m == 0) return C_(0, z.im);
    if (z.re >= 0) {
        double t = sqrt(0.5 * (m + z.re));
        return C_(t, z.im / (2 * t));
    }
    double t = sqrt(0.5 * (m - z.re));
    return C_(fabs(z.im) / (2 * t), z.im < 0 ? -t : t);
}
// std::pow(std::complex<double>, double) as libstdc++ evaluates it (principal branch via log/polar)
__device__ cplx cpow_(cplx x, double y) {
    if (x.im == 0 && x.re > 0) return C_(pow(x.re, y), 0);
    double lr = log(hypot(x.re, x.im)), th = atan2(x.im, x.re);
    double r = exp(y * lr), a = y * th;
    return C_(r * cos(a), r * sin(a));
}

__device__ void o4_roots(const double f[5], double roots[4]) {
    const double A = f[0], B = f[1], C = f[2], D = f[3], E = f[4];
    const double A2 = A * A, B2 = B * B, A3 = A2 * A, B3 = B2 * B, A4 = A3 * A, B4 = B3 * B;
    const double alpha = -3 * B2 / (8 * A2) + C / A;
    const double beta = B3 / (8 * A3) - B * C / (2 * A2) + D / A;
    const double gamma = -3 * B4 / (256 * A4) + B2 * C / (16 * A3) - B * D / (4 * A2) + E / A;
    const double alpha2 = alpha * alpha, alpha3 = alpha2 * alpha;
    const cplx P = C_(-alpha2 / 12 - gamma, 0);
    const cplx Q = C_(-alpha3 / 108 + alpha * gamma / 3 - beta * beta / 8, 0);
    const cplx R = cadd(cscale(Q, -0.5), csqrt_(cadd(cscale(cpow_(Q, 2.0), 0.25), cscale(cpow_(P, 3.0), 1.0 / 27.0))));
    const cplx U = cpow_(R, 1.0 / 3.0);
    cplx y;
    if (U.re == 0) y = csub(C_(-5.0 * alpha / 6.0, 0), cpow_(Q, 1.0 / 3.0));
    else y = cadd(csub(C_(-5.0 * alpha / 6.0, 0), cdivc(P, cscale(U, 3.0))), U);
    const cplx w = csqrt_(cadd(C_(alpha, 0), cscale(y, 2.0)));
    const cplx base = cadd(C_(3.0 * alpha, 0), cscale(y, 2.0));
    const cplx bw = cdivc(C_(2.0 * beta, 0), w);
    const cplx s1 = csqrt_(cscale(cadd(base, bw), -1.0)), s2 = csqrt_(cscale(csub(base, bw), -1.0));
    const double off = -B / (4.0 * A);
    roots[0] = off + 0.5 * (w.re + s1.re);
    roots[1] = off + 0.5 * (w.re - s1.re);
    roots[2] = off + 0.5 * (-w.re + s2.re);
    roots[3] = off + 0.5 * (-w.re - s2.re);
}

struct V3 {
    double x, y, z;
};
__device__ __forceinline__ V3 ld3(const double *p) { return V3{p[0], p[1], p[2]}; }
__device__ __forceinline__ V3 sub(V3 a, V3 b) { return V3{a.x - b.x, a.y - b.y, a.z - b.z}; }
__device__ __forceinline__ V3 add(V3 a, V3 b) { return V3{a.x + b.x, a.y + b.y, a.z + b.z}; }
__device__ __forceinline__ V3 cross(V3 a, V3 b) { return V3{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
__device__ __forceinline__ double dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ double norm(V3 a) { return sqrt(dot(a, a)); }
__device__ __forceinline__ V3 divs(V3 a, double s) { return V3{a.x / s, a.y / s, a.z / s}; }
struct M3 {
    V3 r0, r1, r2;  // rows
};
__device__ __forceinline__ V3 mul(const M3 &M, V3 v) { return V3{dot(M.r0, v), dot(M.r1, v), dot(M.r2, v)}; }
__device__ __forceinline__ V3 mulT(const M3 &M, V3 v) {
    return V3{M.r0.x * v.x + M.r1.x * v.y + M.r2.x * v.z, M.r0.y * v.x + M.r1.y * v.y + M.r2.y * v.z,
              M.r0.z * v.x + M.r1.z * v.y + M.r2.z * v.z};
}

// model = {R row-major (cam->world) [9], t [3]}
__device__ __forceinline__ double p3p_score(const double *m, V3 wp, V3 bv) {
    // inverse = [R^T | -R^T t] applied to the homogeneous point (AbsolutePoseSacProblem.cpp:171-190)
    const V3 t = ld3(m + 9);
    const V3 c0{m[0], m[3], m[6]}, c1{m[1], m[4], m[7]}, c2{m[2], m[5], m[8]};  // columns of R = rows of R^T
    const V3 nRt{-dot(c0, t), -dot(c1, t), -dot(c2, t)};
    V3 r{dot(c0, wp) + nRt.x, dot(c1, wp) + nRt.y, dot(c2, wp) + nRt.z};
    r = divs(r, norm(r));
    return 1.0 - dot(r, bv);
}

// Kneip P3P (opengv p3p_kneip): the quartic is shared, solution `which` (0..3) is back-substituted into sol[12].
// Four lanes of a wave each take one root; returns 0 for a degenerate (collinear) world triple.
__device__ int p3p_kneip(const V3 f[3], const V3 p[3], int which, double sol[12]) {
    V3 P1 = p[0], P2 = p[1], P3 = p[2];
    const V3 t1 = sub(P2, P1), t2 = sub(P3, P1);
    if (norm(cross(t1, t2)) == 0) return 0;
    V3 f1 = f[0], f2 = f[1], f3;
    M3 T;
    for (int pass = 0; pass < 2; pass++) {
        const V3 e1 = f1;
        V3 e3 = cross(f1, f2);
        e3 = divs(e3, norm(e3));
        const V3 e2 = cross(e3, e1);
        T = M3{e1, e2, e3};
        f3 = mul(T, f[2]);
        if (pass == 0 && f3.z > 0) {
            f1 = f[1];
            f2 = f[0];
            P1 = p[1];
            P2 = p[0];
            P3 = p[2];
            continue;
        }
        break;
    }
    V3 n1 = sub(P2, P1);
    n1 = divs(n1, norm(n1));
    const V3 d = sub(P3, P1);
    V3 n3 = cross(n1, d);
    n3 = divs(n3, norm(n3));
    const V3 n2 = cross(n3, n1);
    const M3 N{n1, n2, n3};
    const V3 P3n = mul(N, d);
    const double d_12 = norm(t1);
    const double f_1 = f3.x / f3.z, f_2 = f3.y / f3.z, p_1 = P3n.x, p_2 = P3n.y;
    const double cos_beta = dot(f1, f2);
    double b = 1 / (1 - cos_beta * cos_beta) - 1;
    b = cos_beta < 0 ? -sqrt(b) : sqrt(b);
    const double f_1_pw2 = f_1 * f_1, f_2_pw2 = f_2 * f_2, p_1_pw2 = p_1 * p_1, p_1_pw3 = p_1_pw2 * p_1, p_1_pw4 = p_1_pw3 * p_1;
    const double p_2_pw2 = p_2 * p_2, p_2_pw3 = p_2_pw2 * p_2, p_2_pw4 = p_2_pw3 * p_2, d_12_pw2 = d_12 * d_12, b_pw2 = b * b;
    double fac[5];
    fac[0] = -f_2_pw2 * p_2_pw4 - p_2_pw4 * f_1_pw2 - p_2_pw4;
    fac[1] = 2 * p_2_pw3 * d_12 * b + 2 * f_2_pw2 * p_2_pw3 * d_12 * b - 2 * f_2 * p_2_pw3 * f_1 * d_12;
    fac[2] = -f_2_pw2 * p_2_pw2 * p_1_pw2 - f_2_pw2 * p_2_pw2 * d_12_pw2 * b_pw2 - f_2_pw2 * p_2_pw2 * d_12_pw2 + f_2_pw2 * p_2_pw4 +
             p_2_pw4 * f_1_pw2 + 2 * p_1 * p_2_pw2 * d_12 + 2 * f_1 * f_2 * p_1 * p_2_pw2 * d_12 * b - p_2_pw2 * p_1_pw2 * f_1_pw2 +
             2 * p_1 * p_2_pw2 * f_2_pw2 * d_12 - p_2_pw2 * d_12_pw2 * b_pw2 - 2 * p_1_pw2 * p_2_pw2;
    fac[3] = 2 * p_1_pw2 * p_2 * d_12 * b + 2 * f_2 * p_2_pw3 * f_1 * d_12 - 2 * f_2_pw2 * p_2_pw3 * d_12 * b - 2 * p_1 * p_2 * d_12_pw2 * b;
    fac[4] = -2 * f_2 * p_2_pw2 * f_1 * p_1 * d_12 * b + f_2_pw2 * p_2_pw2 * d_12_pw2 + 2 * p_1_pw3 * d_12 - p_1_pw2 * d_12_pw2 +
             f_2_pw2 * p_2_pw2 * p_1_pw2 - p_1_pw4 - 2 * f_2_pw2 * p_2_pw2 * p_1 * d_12 + p_2_pw2 * f_1_pw2 * p_1_pw2 +
             f_2_pw2 * p_2_pw2 * d_12_pw2 * b_pw2;
    double roots[4];
    o4_roots(fac, roots);
    {
        const double root = which == 0 ? roots[0] : (which == 1 ? roots[1] : (which == 2 ? roots[2] : roots[3]));
        const double cot_alpha = (-f_1 * p_1 / f_2 - root * p_2 + d_12 * b) / (-f_1 * root * p_2 / f_2 + p_1 - d_12);
        const double cos_theta = root, sin_theta = sqrt(1 - root * root);
        const double sin_alpha = sqrt(1 / (cot_alpha * cot_alpha + 1));
        double cos_alpha = sqrt(1 - sin_alpha * sin_alpha);
        if (cot_alpha < 0) cos_alpha = -cos_alpha;
        const double k = d_12 * (sin_alpha * b + cos_alpha);
        const V3 Cv{cos_alpha * k, cos_theta * sin_alpha * k, sin_theta * sin_alpha * k};
        const V3 Cw = add(P1, mulT(N, Cv));
        const M3 R{V3{-cos_alpha, -sin_alpha * cos_theta, -sin_alpha * sin_theta}, V3{sin_alpha, -cos_alpha * cos_theta, -cos_alpha * sin_theta},
                   V3{0.0, -sin_theta, cos_theta}};
        // Rout = N^T R^T T : column j of (R^T T) = R^T * (column j of T)
        const V3 Tc[3] = {V3{T.r0.x, T.r1.x, T.r2.x}, V3{T.r0.y, T.r1.y, T.r2.y}, V3{T.r0.z, T.r1.z, T.r2.z}};
        for (int j = 0; j < 3; j++) {
            const V3 col = mulT(N, mulT(R, Tc[j]));
            sol[j] = col.x;
            sol[3 + j] = col.y;
            sol[6 + j] = col.z;
        }
        sol[9] = Cw.x;
        sol[10] = Cw.y;
        sol[11] = Cw.z;
    }
    return 4;
}

// ---------------------------------------------------------------------------------------------------------------
// ONE launch for the whole LMedS loop (Lmeds.hpp:60-190): one workgroup per drawn sample
//   1. hypothesis: 4 lanes back-substitute the 4 quartic roots, the 4th point picks the solution
//      (AbsolutePoseSacProblem.cpp:41-110)
//   2. penalty: squared clipped scores of all n points -> LDS, median by MSB-first radix SELECT on the IEEE bit patterns
//      (non-negative doubles order like their uint64 bits) instead of the reference's full std::sort (Lmeds.hpp:96-130
//      only ever reads distances[mid-1] and distances[mid]): 8 passes of a 256-bin LDS histogram
//   3. the workgroup that finishes LAST (device-scope counter) picks the best of the first max_iters valid hypotheses
//      and classifies the inliers of the winner (Lmeds.hpp:150-190)
// (Tried and dropped: starting the first digit at the highest bit in which the keys differ -- min / max reduced across the workgroup first;
// the squared distances of a model share sign and upper exponent bits, so from bit 63 the first pass decides little.  The two extra
// barriers cost more (score phase +1.2 us) than the shorter select saved (-0.3 us): the select is barrier-bound, not atomic-bound.)
// k-th smallest of n 64-bit keys in LDS, NT threads (256 bins).  Most-significant-digit radix select, 8 bits per pass, with two histogram
// buffers (the next pass's buffer is cleared while this pass counts: two barriers per pass instead of four) and an early exit: as
// soon as the selected bin holds ONE key, that key is the answer and one scan fetches it (squared distances of a model differ
// within their first 3-4 digits, so 3-4 passes instead of 8).  The barriers of this routine were the fixed cost of a workgroup:
// with thousands of workgroups in flight (a batch of cameras) the kernel time did not depend on n.
template <int NT>
__device__ unsigned long long radix_select(const unsigned long long *keys, int n, int k, unsigned int *hist /* [512] */, int *s_bin, int *s_k,
                                           int *s_binc, unsigned long long *s_key) {
    unsigned long long prefix = 0, mask = 0;
    if (threadIdx.x < 256) hist[threadIdx.x] = 0;
    __syncthreads();
    int pass = 0;
    for (int shift = 56; shift >= 0; shift -= 8, pass++) {
        unsigned int *h = hist + 256 * (pass & 1);
        if (threadIdx.x < 256) hist[256 * ((pass + 1) & 1) + threadIdx.x] = 0;  // last read two barriers ago
        for (int i = threadIdx.x; i < n; i += NT) {
            const unsigned long long key = keys[i];
            if ((key & mask) == prefix) atomicAdd(&h[(unsigned) (key >> shift) & 255u], 1u);
        }
        __syncthreads();
        if (threadIdx.x < 64) {
            const int l = threadIdx.x;
            const unsigned c0 = h[4 * l], c1 = h[4 * l + 1], c2 = h[4 * l + 2], c3 = h[4 * l + 3];
            const unsigned tot = c0 + c1 + c2 + c3;
            unsigned incl = tot;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const unsigned v = __shfl_up(incl, off);
                if (l >= off) incl += v;
            }
            const unsigned excl = incl - tot;
            if ((unsigned) k >= excl && (unsigned) k < incl) {
                unsigned r = (unsigned) k - excl, c = c0;
                int b = 0;
                if (r >= c0) {
                    r -= c0; b = 1; c = c1;
                    if (r >= c1) {
                        r -= c1; b = 2; c = c2;
                        if (r >= c2) { r -= c2; b = 3; c = c3; }
                    }
                }
                *s_bin = 4 * l + b;
                *s_k = (int) r;
                *s_binc = (int) c;
            }
        }
        __syncthreads();
        prefix |= (unsigned long long) (unsigned) *s_bin << shift;
        mask |= 0xffull << shift;
        k = *s_k;
        if (*s_binc == 1 && shift > 0) {  // workgroup-uniform
            for (int i = threadIdx.x; i < n; i += NT) {
                const unsigned long long key = keys[i];
                if ((key & mask) == prefix) *s_key = key;  // exactly one key matches
            }
            __syncthreads();
            return *s_key;
        }
    }
    return prefix;
}


#define P3P_STAMP(k) do { if (MODE == 0 && A.dbg && threadIdx.x == 0 && h < 256) A.dbg[8 * h + (k)] = wall_clock64(); } while (0)

// Inter-workgroup hand-off inside ONE launch (MODE 0): every published word is an 8-byte agent-scope atomic on both sides -- one of the
// valid forms of MI355X_MICROARCH.md "inter-workgroup visibility" -- so neither the arriving workgroups need a release fence (an L2
// write-back, ~1.7 us) nor the selecting one an acquire (~3.5 us as __threadfence()); the arrival counter is ordered behind the stores by
// an explicit s_waitcnt vmcnt(0) (the compiler may not drop inline asm).
__device__ __forceinline__ void agent_store(double *p, double v) {
    __hip_atomic_store(reinterpret_cast<unsigned long long *>(p), (unsigned long long) __double_as_longlong(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void agent_store(unsigned long long *p, unsigned long long v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ unsigned long long agent_load(const unsigned long long *p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ double agent_load(const double *p) {
    return __longlong_as_double((long long) __hip_atomic_load(reinterpret_cast<const unsigned long long *>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}

// ---- 1. hypothesis: Kneip P3P on one sample, on 4 lanes (one per candidate solution); the other lanes of the wave mirror them.
// Returns whether a model was found; lanes 0..3 of the group have written it to A.models / A.valid.
__device__ __forceinline__ int p3p_hypothesis(const P3pArgs &A, const int h, const int which, const bool writer, double *s_m,
                                              const int *smp_inline = nullptr, const bool agent = false) {
    const double *bv = A.bv, *wpt = A.wpt;
    const int *smp = smp_inline ? smp_inline : A.samples + 4 * h;
    const int i0 = smp[0], i1 = smp[1], i2 = smp[2], i3 = smp[3];
    V3 f[3] = {ld3(bv + 3 * (size_t) i0), ld3(bv + 3 * (size_t) i1), ld3(bv + 3 * (size_t) i2)};
    V3 p[3] = {ld3(wpt + 3 * (size_t) i0), ld3(wpt + 3 * (size_t) i1), ld3(wpt + 3 * (size_t) i2)};
    double sol[12];
    const int ns = p3p_kneip(f, p, which, sol);
    // the solution closest to the 4th correspondence, first one on ties (strict <, initial 1e6)
    double sc = ns ? p3p_score(sol, ld3(wpt + 3 * (size_t) i3), ld3(bv + 3 * (size_t) i3)) : 2000000.0;
    if (!(sc < 1000000.0)) sc = 2000000.0;  // NaN or too large: never selected
    double best = sc;
    int bi = which;
#pragma unroll
    for (int off = 1; off < 4; off <<= 1) {
        const double o = __shfl_xor(best, off);
        const int oi = __shfl_xor(bi, off);
        if (o < best || (o == best && oi < bi)) {
            best = o;
            bi = oi;
        }
    }
    const int ok = best < 1000000.0;
    if (writer && ok && bi == which) {
#pragma unroll
        for (int k = 0; k < 12; k++) {
            if (s_m) s_m[k] = sol[k];
            if (agent) agent_store(A.models + 12 * (size_t) h + k, sol[k]);
            else A.models[12 * (size_t) h + k] = sol[k];
        }
    }
    if (writer && which == 0 && !agent) A.valid[h] = ok;   // (the one-launch form publishes validity inside the penalty word)
    return ok;
}

// MODE 0: the whole LMedS in one launch (hypothesis, penalty, and the workgroup that finishes last selects).
// MODE 1 / 2: the batch's middle and last launch -- penalty of hypothesis h from the model k_p3p_hyp_batch left in memory | selection.
// Between launches the kernel boundary orders the memory; inside one launch every workgroup pays a device-scope fence (an L2
// write-back on a multi-XCD part) before it signals -- 8 192 of them per step for 64 cameras.
template <int MODE, int NT>
// Returns 1 in the workgroup that made the selection (MODE 0: the last one to arrive; MODE 2: the only one), 0 in the others: the fused
// P3P -> PnP launch continues with the refinement in exactly that workgroup (pnp.hip).
__device__ __forceinline__ int p3p_block(const P3pArgs &A, const int h, const int *smp_inline = nullptr) {
    extern __shared__ unsigned long long s_keys[];
    constexpr int NWV = NT / 64;
    __shared__ unsigned int s_hist[512];
    __shared__ int s_bin, s_k, s_binc, s_valid, s_last;
    __shared__ unsigned long long s_key;
    __shared__ unsigned long long s_min[NWV];
    __shared__ unsigned int s_cnt[NWV];
    __shared__ double s_m[12];
    const int n = A.n;
    const double *bv = A.bv, *wpt = A.wpt;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    P3P_STAMP(0);
    if (MODE == 1) {
        if (threadIdx.x < 12) s_m[threadIdx.x] = A.models[12 * (size_t) h + threadIdx.x];
        if (threadIdx.x == 0) s_valid = A.valid[h];
    } else if (MODE == 0 && threadIdx.x < 64) {
        const int ok = p3p_hypothesis(A, h, threadIdx.x & 3, threadIdx.x < 4, s_m, smp_inline, true);
        if (threadIdx.x == 0) s_valid = ok;
    }
    __syncthreads();
    P3P_STAMP(1);
    // ---- 2. LMedS penalty ----------------------------------------------------------------------------------------
    double pen = INFINITY;
    if (MODE != 2 && s_valid) {
        if (MODE == 0) {
            // ... and, while the score is at hand, this hypothesis' inlier mask (Lmeds.hpp:180-183: raw, unsquared distance <= threshold),
            // one ballot per 64 points: should this hypothesis win, the selecting workgroup copies the mask instead of scoring all
            // points once more (stamps: 2.9 us of the kernel's serial tail)
            const int words = (n + 63) >> 6;
            for (int base = wave * 64; base < n; base += NT) {   // wave-uniform bounds: the mask is a ballot
                const int i = base + lane;
                bool in = false;
                if (i < n) {
                    double d = p3p_score(s_m, ld3(wpt + 3 * (size_t) i), ld3(bv + 3 * (size_t) i));
                    in = d <= A.threshold;
                    if (d < 0) d = 0;
                    double v = d * d;
                    if (v != v) v = INFINITY;  // NaN scores order last (std::sort's behaviour with NaN is unspecified)
                    s_keys[i] = (unsigned long long) __double_as_longlong(v);
                }
                const unsigned long long mk = __ballot(in);
                if (lane == 0) agent_store(A.masks + (size_t) h * words + (base >> 6), mk);
            }
        } else
        for (int i = threadIdx.x; i < n; i += NT) {
            double d = p3p_score(s_m, ld3(wpt + 3 * (size_t) i), ld3(bv + 3 * (size_t) i));
            if (d < 0) d = 0;
            double v = d * d;
            if (v != v) v = INFINITY;  // NaN scores order last (std::sort's behaviour with NaN is unspecified)
            s_keys[i] = (unsigned long long) __double_as_longlong(v);
        }
        __syncthreads();
        P3P_STAMP(2);
        const int mid = n / 2;
        const unsigned long long kmid = radix_select<NT>(s_keys, n, mid, s_hist, &s_bin, &s_k, &s_binc, &s_key);
        if (n % 2 != 0) {
            pen = __longlong_as_double((long long) kmid);
        } else {
            // even n: also need the (mid-1)-th smallest = max{x : x < kmid} unless kmid is duplicated below rank mid
            unsigned long long below = 0;
            unsigned int cnt_lt = 0;
            for (int i = threadIdx.x; i < n; i += NT) {
                const unsigned long long key = s_keys[i];
                if (key < kmid) {
                    cnt_lt++;
                    below = key > below ? key : below;
                }
            }
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) {
                const unsigned long long ob = __shfl_xor(below, off);
                below = ob > below ? ob : below;
                cnt_lt += __shfl_xor(cnt_lt, off);
            }
            if (lane == 0) {
                s_min[wave] = below;
                s_cnt[wave] = cnt_lt;
            }
            __syncthreads();
            unsigned long long bmax = 0;
            unsigned int ctot = 0;
#pragma unroll
            for (int w = 0; w < NWV; w++) {
                bmax = s_min[w] > bmax ? s_min[w] : bmax;
                ctot += s_cnt[w];
            }
            // elements < kmid occupy ranks [0, cnt_lt); rank mid-1 is below kmid only if cnt_lt == mid
            const unsigned long long klo = (ctot == (unsigned) mid) ? bmax : kmid;
            pen = (__longlong_as_double((long long) klo) + __longlong_as_double((long long) kmid)) / 2;
        }
    }
    P3P_STAMP(3);
    // ---- 3. last workgroup selects -------------------------------------------------------------------------------
    if (MODE == 1) {
        if (threadIdx.x == 0) A.penalty[h] = pen;
        return 0;
    }
    if (MODE == 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // every wave's mask words have landed ...
        __syncthreads();                                    // ... before thread 0 arrives
        if (threadIdx.x == 0) {
            agent_store(A.penalty + h, s_valid ? pen : -1.0);   // one word: penalties are >= 0 (or +inf), a failed model publishes -1
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // this wave's model stores (lanes 0..3) and the word above have landed
            s_last = __hip_atomic_fetch_add(A.counter, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == A.H - 1;
        }
        __syncthreads();
        P3P_STAMP(4);
        if (!s_last) return 0;
        if (threadIdx.x == 0) *A.counter = 0;  // ready for the next launch (stream order)
    }
    const int *vvalid = A.valid;            // MODE 2 reads what earlier LAUNCHES wrote: plain loads
    const double *vpen = A.penalty, *vmodels = A.models;
    // first max_iters VALID hypotheses in draw order (failed models do not count as iterations, Lmeds.hpp:88-92);
    // smallest penalty, earliest on ties (strict <).  One wave, 64 hypotheses per round, no barriers.
    __shared__ int s_best, s_used;
    if (wave == 0) {
        int used = 0, bestI = -1;
        double bestP = 1.7976931348623157e308;
        for (int base0 = 0; base0 < A.H && used < A.max_iters; base0 += 256) {
            // four rounds' flags and penalties requested at once (they come from other CUs' writes: every dependent read is a trip to
            // L2 / memory, and "penalty only if valid" made two trips per round)
            int vv[4];
            double pp[4];
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int hh = base0 + 64 * r + lane;
                const bool in = hh < A.H;
                if (MODE == 0) {
                    pp[r] = in ? agent_load(vpen + hh) : -1.0;
                    vv[r] = !(pp[r] < 0);
                } else {
                    vv[r] = in ? vvalid[hh] : 0;
                    pp[r] = in ? vpen[hh] : INFINITY;
                }
            }
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int hh = base0 + 64 * r + lane;
                if (base0 + 64 * r >= A.H || used >= A.max_iters) break;
                const bool v = vv[r] != 0;
                const unsigned long long m = __ballot(v);
                const int before = used + __popcll(m & ((1ull << lane) - 1ull));
                const bool ok = v && before < A.max_iters;
                double bp = ok ? pp[r] : INFINITY;
                int bi = ok ? hh : 0x7fffffff;
#pragma unroll
                for (int off = 32; off > 0; off >>= 1) {
                    const double o = __shfl_xor(bp, off);
                    const int oi = __shfl_xor(bi, off);
                    if (o < bp || (o == bp && oi < bi)) {
                        bp = o;
                        bi = oi;
                    }
                }
                if (bi != 0x7fffffff && bp < bestP) {
                    bestP = bp;
                    bestI = bi;
                }
                used = min(used + __popcll(m), A.max_iters);
            }
        }
        if (lane == 0) {
            s_best = bestI;
            s_used = used;
        }
    }
    __syncthreads();
    const int bestI = s_best;
    P3P_STAMP(5);
    SelectOut *out = A.out;
    if (threadIdx.x == 0) {
        out->best = bestI;
        out->n_valid_used = s_used;
        out->have_model = bestI >= 0;
        s_k = 0;
    }
    if (bestI < 0) {
        if (threadIdx.x == 0) out->n_inliers = 0;
        return 1;
    }
    if (threadIdx.x < 12) {
        s_m[threadIdx.x] = MODE == 0 ? agent_load(vmodels + 12 * (size_t) bestI + threadIdx.x) : vmodels[12 * (size_t) bestI + threadIdx.x];
        out->model[threadIdx.x] = s_m[threadIdx.x];
    }
    __syncthreads();
    int cnt = 0;
    if (MODE == 0) {   // the winner's mask, as its own workgroup classified the points
        const int words = (n + 63) >> 6;
        for (int w = threadIdx.x; w < words; w += NT) {
            const unsigned long long mk = agent_load(A.masks + (size_t) bestI * words + w);
            s_keys[w] = mk;
            cnt += __popcll(mk);
        }
        __syncthreads();
        for (int i = threadIdx.x; i < n; i += NT) A.inlier[i] = (uint8_t) ((s_keys[i >> 6] >> (i & 63)) & 1ull);
    } else
    for (int i = threadIdx.x; i < n; i += NT) {
        const double d = p3p_score(s_m, ld3(wpt + 3 * (size_t) i), ld3(bv + 3 * (size_t) i));
        const bool in = d <= A.threshold;  // Lmeds.hpp:180-183 (raw, unsquared distance)
        A.inlier[i] = in;
        cnt += in;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) cnt += __shfl_xor(cnt, off);
    if (lane == 0) atomicAdd(&s_k, cnt);
    __syncthreads();
    if (threadIdx.x == 0) out->n_inliers = s_k;
    P3P_STAMP(6);
    return 1;
}

}  // namespace
