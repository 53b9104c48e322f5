// TEST: alvaar_amd/csrc/exact_sincos.hpp compiled as HOST code against the C library's (float) cos((double) a) / (float) sin((double) a)
// -- the two floats cv::ORB's rBRIEF rotates its pattern with (features2d/src/orb.cpp:230-232).  Prints "N mismatches ambiguous".
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#define __device__
#define __forceinline__ inline
static inline float __uint_as_float(unsigned u) { float f; std::memcpy(&f, &u, 4); return f; }
static inline unsigned __float_as_uint(float f) { unsigned u; std::memcpy(&u, &f, 4); return u; }
#include "exact_sincos_nohip.hpp"
int main(int argc, char **argv) {
    const long N = argc > 1 ? std::atol(argv[1]) : 3000000;
    std::mt19937 g(1);
    long mism = 0, amb = 0;
    for (long i = 0; i < N; i++) {
        float a;
        if (i % 3 == 0) a = (float) ((g() >> 8) * (360.0 / 16777216.0)) * (float) (3.1415926535897932384626433832795 / 180.f);  // degrees -> radians as ORB does
        else if (i % 3 == 1) a = __uint_as_float(0x30000000u + (g() % 0x10c90fdbu));                                            // every binade up to 2 pi
        else a = (float) (g() * (6.2831853 / 4294967296.0));
        float c, s;
        int am = 0;
        alva_dd::sincos_float(a, &c, &s, &am);
        if (c != (float) std::cos((double) a) || s != (float) std::sin((double) a)) mism++;
        amb += am;
    }
    std::printf("%ld %ld %ld\n", N, mism, amb);
    return 0;
}
