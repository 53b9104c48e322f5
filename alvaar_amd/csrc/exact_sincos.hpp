// (float) cos((double) a) and (float) sin((double) a) for a float angle |a| <= 8, decided from a double-double (~100-bit) evaluation.
//
// Why: rBRIEF rotates its pattern by the keypoint angle with `float a = (float) cos(angle), b = (float) sin(angle)` (features2d/src/
// orb.cpp:230-232: the float angle is promoted, the C library's DOUBLE cos / sin rounded to float).  The descriptor bits are integer
// compares at coordinates rounded from these two floats, so bit-exact descriptors need exactly these floats.  A device cos() is not
// the host's, but both are within 1 ulp (double) of the true value, and a float rounding of such a value can differ from the rounding
// of the TRUE value only when the true value lies within that error of a float rounding boundary (a midpoint between adjacent floats).
// This routine computes the true value to ~2^-95, rounds THAT to float, and reports `ambiguous` when the true value is closer than
// 2^-50 (relative) to a midpoint -- four double ulps, more than the 0.55-ulp bound of glibc's cos / sin (sysdeps/ieee754/dbl-64/s_sin.c)
// plus this routine's own error.  Outside that band (probability ~3e-8 per evaluation) the result is PROVABLY the reference's.
#pragma once
#include <hip/hip_runtime.h>

namespace alva_dd {

struct dd {
    double h, l;
};
__device__ __forceinline__ dd two_sum(double a, double b) {
    const double s = a + b, bb = s - a;
    return dd{s, (a - (s - bb)) + (b - bb)};
}
__device__ __forceinline__ dd quick_two_sum(double a, double b) {
    const double s = a + b;
    return dd{s, b - (s - a)};
}
__device__ __forceinline__ dd two_prod(double a, double b) {
    const double p = a * b;
    return dd{p, fma(a, b, -p)};
}
__device__ __forceinline__ dd add(dd a, dd b) {
    dd s = two_sum(a.h, b.h);
    dd t = two_sum(a.l, b.l);
    s.l += t.h;
    s = quick_two_sum(s.h, s.l);
    s.l += t.l;
    return quick_two_sum(s.h, s.l);
}
__device__ __forceinline__ dd mul(dd a, dd b) {
    dd p = two_prod(a.h, b.h);
    p.l += a.h * b.l + a.l * b.h;
    return quick_two_sum(p.h, p.l);
}
__device__ __forceinline__ dd mul_d(dd a, double b) {
    dd p = two_prod(a.h, b);
    p.l += a.l * b;
    return quick_two_sum(p.h, p.l);
}
__device__ __forceinline__ dd div_d(dd a, double b) {  // b an exactly representable small integer
    const double q1 = a.h / b;
    dd p = two_prod(q1, b);
    const double r = ((a.h - p.h) - p.l) + a.l;
    return quick_two_sum(q1, r / b);
}
__device__ __forceinline__ dd neg(dd a) { return dd{-a.h, -a.l}; }

// round a double-double to float (round to nearest even on the exact value); *ambiguous |= the value is within rel 2^-50 of a midpoint
__device__ __forceinline__ float to_float(dd v, int *ambiguous) {
    float f = (float) v.h;                       // candidate; may be off by one float ulp only when v.h sits at a boundary
    for (int it = 0; it < 2; it++) {
        const double err = (v.h - (double) f) + v.l;   // v - f (v.h - f is exact)
        const float up = __uint_as_float(__float_as_uint(fabsf(f)) + 1u), dn = __uint_as_float(__float_as_uint(fabsf(f)) - 1u);
        const double half_up = 0.5 * ((double) up - (double) fabsf(f)), half_dn = 0.5 * ((double) fabsf(f) - (double) dn);
        const double e = f < 0 ? -err : err;     // error towards larger magnitude
        if (e > half_up) f = f < 0 ? -up : up;
        else if (-e > half_dn) f = f < 0 ? -dn : dn;
        else {
            const double tol = ldexp(fabs(v.h), -50);
            if (fabs(e - half_up) < tol || fabs(-e - half_dn) < tol) *ambiguous = 1;
            break;
        }
    }
    return f;
}

// pi/2 in three doubles (exact sum to ~2^-160)
#define ALVA_PIO2_1 1.5707963267948966
#define ALVA_PIO2_2 6.123233995736766e-17
#define ALVA_PIO2_3 -1.4973849048591698e-33

__device__ __forceinline__ void sincos_float(float angle, float *c_out, float *s_out, int *ambiguous) {
    const double x = (double) angle;
    const double kd = rint(x * 0.63661977236758134308);   // x / (pi/2)
    const int k = (int) kd;
    // r = x - k * pi/2 as a double-double (|k| <= 6: every product below is exact or has a tiny, accounted rounding)
    dd r = add(dd{x, 0.0}, neg(two_prod(kd, ALVA_PIO2_1)));
    r = add(r, neg(two_prod(kd, ALVA_PIO2_2)));
    r = add(r, dd{-kd * ALVA_PIO2_3, 0.0});
    // Taylor series of sin r and cos r, |r| <= pi/4 + eps: terms to r^29 (< 1e-33)
    const dd r2 = mul(r, r);
    dd term = r, s = r;
    for (int n = 1; n <= 14; n++) {
        term = div_d(mul(term, r2), (double) ((2 * n) * (2 * n + 1)));
        s = add(s, (n & 1) ? neg(term) : term);
    }
    dd c = dd{1.0, 0.0};
    term = dd{1.0, 0.0};
    for (int n = 1; n <= 14; n++) {
        term = div_d(mul(term, r2), (double) ((2 * n - 1) * (2 * n)));
        c = add(c, (n & 1) ? neg(term) : term);
    }
    dd cs, sn;
    switch (k & 3) {
        case 0: cs = c; sn = s; break;
        case 1: cs = neg(s); sn = c; break;
        case 2: cs = neg(c); sn = neg(s); break;
        default: cs = s; sn = neg(c); break;
    }
    *c_out = to_float(cs, ambiguous);
    *s_out = to_float(sn, ambiguous);
}

}  // namespace alva_dd
