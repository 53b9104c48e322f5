// TEST (GPU): the drop-in class `alva::System` (include/alvaar_system.h) LINKED against libalvaar_hip.so and driven exactly as the
// reference's embind surface is (src/slam/src/system.hpp:30-36): every buffer argument is a 32-bit heap offset passed as `int`.
// The buffers are allocated below 4 GiB with mmap(MAP_32BIT), the native stand-in for the wasm heap.
// Usage: system_int_offsets <width> <height> <frames.bin (N x H x W x 4 RGBA)> <N>; prints one line per frame: status + pose.
#include <sys/mman.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include "alvaar_system.h"

static void *heap32(size_t bytes) {
    void *p = mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_32BIT, -1, 0);
    if (p == MAP_FAILED || (uintptr_t) p + bytes > 0xffffffffull) {
        std::fprintf(stderr, "no memory below 4 GiB\n");
        std::exit(3);
    }
    return p;
}

int main(int argc, char **argv) {
    if (argc < 5) return 2;
    const int w = std::atoi(argv[1]), h = std::atoi(argv[2]), n = std::atoi(argv[4]);
    const size_t fb = (size_t) w * h * 4;
    uint8_t *memImg = (uint8_t *) heap32(fb);                 // src/system.js:63-67: the five shared buffers
    float *memCam = (float *) heap32(16 * sizeof(float));
    float *memObj = (float *) heap32(16 * sizeof(float));
    int *memPts = (int *) heap32(4096 * sizeof(int));
    double *memIMU = (double *) heap32(256 * sizeof(double));
    FILE *f = std::fopen(argv[3], "rb");
    if (!f) return 2;
    alva::System sys;
    const double fov = 45.0 * 0.01745329251994329576, fl = (w * 0.5) / std::tan(fov * ((double) w / h) * 0.5) < (h * 0.5) / std::tan(fov * 0.5)
                                                                ? (w * 0.5) / std::tan(fov * ((double) w / h) * 0.5) : (h * 0.5) / std::tan(fov * 0.5);
    sys.configure(w, h, fl, fl, w * 0.5, h * 0.5, 0, 0, 0, 0);
    if (sys.status() != 0) {
        std::fprintf(stderr, "configure failed: %s\n", alva_system_last_error());
        return 4;
    }
    for (int k = 0; k < n; k++) {
        if (std::fread(memImg, 1, fb, f) != fb) return 5;
        const int status = sys.findCameraPose((int) (uintptr_t) memImg, (int) (uintptr_t) memCam);   // the int-typed twins
        const int n2d = sys.getFramePoints((int) (uintptr_t) memPts);
        std::printf("%d %d %d", k, status, n2d);
        for (int i = 0; i < 16; i++) std::printf(" %.9g", memCam[i]);
        std::printf("\n");
    }
    const int plane = sys.findPlane((int) (uintptr_t) memObj, 250);
    std::memset(memIMU, 0, 256 * sizeof(double));
    memIMU[0] = 1.0;
    std::rewind(f);
    if (std::fread(memImg, 1, fb, f) != fb) return 5;
    const int imu = sys.findCameraPoseWithIMU((int) (uintptr_t) memImg, (int) (uintptr_t) memIMU, (int) (uintptr_t) memCam);
    std::printf("plane %d imu %d %.9g\n", plane, imu, memCam[15]);
    sys.reset();
    std::fclose(f);
    return 0;
}
