// How fast does the HOST reach pinned memory?  Random read-modify-write over 1 KB records (the map layer's access pattern) in
// malloc'd memory, hipHostMalloc default / non-coherent / NUMA-user memory, and malloc'd memory registered with hipHostRegister.
// build + run on the GPU box: hipcc -O2 tools/probes/pinned_probe.cpp -o /tmp/pinned_probe && /tmp/pinned_probe
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <sys/mman.h>
static double run(uint8_t *base, size_t nrec, const std::vector<int> &order) {
    volatile long sink = 0;
    double best = 1e9;
    for (int rep = 0; rep < 5; rep++) {
        auto t0 = std::chrono::steady_clock::now();
        for (int idx: order) {
            long *r = (long *) (base + (size_t) idx * 1024);
            r[0] += 1;          // header line
            r[9] += r[0];       // second line
            r[17] ^= r[9];      // third line
            sink += r[17];
        }
        double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        best = dt < best ? dt : best;
    }
    (void) nrec;
    return best / order.size() * 1e9;
}
int main() {
    const size_t nrec = 16384, bytes = nrec * 1024;
    std::vector<int> order(5000);
    srand(1);
    for (int &v: order) v = rand() % nrec;
    uint8_t *p = nullptr;
    p = (uint8_t *) malloc(bytes); memset(p, 0, bytes);
    printf("malloc                      %6.1f ns / record\n", run(p, nrec, order));
    if (hipHostRegister(p, bytes, hipHostRegisterDefault) == hipSuccess) {
        printf("malloc + hipHostRegister    %6.1f ns / record\n", run(p, nrec, order));
        (void) hipHostUnregister(p);
    }
    free(p);
    p = (uint8_t *) aligned_alloc(2 << 20, bytes); madvise(p, bytes, MADV_HUGEPAGE); memset(p, 0, bytes);
    printf("aligned + MADV_HUGEPAGE     %6.1f ns / record\n", run(p, nrec, order));
    if (hipHostRegister(p, bytes, hipHostRegisterDefault) == hipSuccess) {
        printf("  ... + hipHostRegister     %6.1f ns / record\n", run(p, nrec, order));
        (void) hipHostUnregister(p);
    }
    free(p);
    const unsigned flags[] = {hipHostMallocDefault, hipHostMallocNonCoherent, hipHostMallocCoherent, hipHostMallocNumaUser, hipHostMallocPortable | hipHostMallocMapped};
    const char *names[] = {"hipHostMallocDefault", "hipHostMallocNonCoherent", "hipHostMallocCoherent", "hipHostMallocNumaUser", "Portable|Mapped"};
    for (int i = 0; i < 5; i++) {
        if (hipHostMalloc((void **) &p, bytes, flags[i]) != hipSuccess) { printf("%s: failed\n", names[i]); continue; }
        memset(p, 0, bytes);
        printf("%-27s %6.1f ns / record\n", names[i], run(p, nrec, order));
        (void) hipHostFree(p);
    }
    return 0;
}
