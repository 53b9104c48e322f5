// FrameRec::floor_to_int (slam.hpp) against (int) std::floor(float) -- what Frame::getKeypointCellIdx (frame.cpp:313-318) computes -- on
// every float the product can feed it and then some: all cell boundaries of the supported cell sizes +- a few ulps, the quotient x / cell
// for x on a dense grid over [-64, 4200), negative values, exact integers, tiny values, and a few million random bit patterns in int range.
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <random>
#include "../../alvaar_amd/csrc/slam/slam.hpp"

static long bad = 0, n = 0;
static void check(float v) {
    if (!(std::fabs(v) < 2.0e9f)) return;   // outside int: undefined for both
    n++;
    const int want = (int) std::floor(v), got = alva_slam::FrameRec::floor_to_int(v);
    if (want != got) {
        if (bad < 10) std::printf("floor(%.9g): %d != %d\n", (double) v, got, want);
        bad++;
    }
}
int main() {
    for (int cell = 8; cell <= 64; cell++) {
        const float cf = (float) cell;
        for (int k = -2; k <= 4200 / cell + 2; k++) {
            float x = (float) (k * cell);
            for (int u = -3; u <= 3; u++) {
                float y = x;
                for (int s = 0; s < (u < 0 ? -u : u); s++) y = std::nextafterf(y, u < 0 ? -1e30f : 1e30f);
                check(y / cf);
                check(y);
            }
        }
        for (int i = -64 * 16; i < 4200 * 16; i++) check(((float) i * 0.0625f + 0.013f) / cf);
    }
    for (int i = -100000; i <= 100000; i++) {
        check((float) i);
        check((float) i + 0.5f);
        check((float) i * 1e-3f);
    }
    const float tiny[] = {0.f, -0.f, 1e-30f, -1e-30f, 1e-45f, -1e-45f, 0.99999994f, -0.99999994f, 1.0000001f, -1.0000001f};
    for (float v: tiny) check(v);
    std::mt19937 rng(12345);
    for (int i = 0; i < 4000000; i++) {
        uint32_t b = rng();
        float v;
        std::memcpy(&v, &b, 4);
        if (v == v) check(v);
    }
    std::printf("%ld values, %ld mismatches\n", n, bad);
    return bad != 0;
}
