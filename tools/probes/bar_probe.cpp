// Can the HOST write device memory directly (large BAR), and how fast?  hipExtMallocWithFlags(hipDeviceMallocFinegrained / Uncached) and
// plain hipMalloc, written by the CPU with ordinary stores, checked by a kernel.
// hipcc -O2 --offload-arch=gfx950 tools/probes/bar_probe.cpp -o /tmp/bar_probe && /tmp/bar_probe
#include <hip/hip_runtime.h>
#include <chrono>
#include <csetjmp>
#include <csignal>
#include <cstdio>
#include <cstring>
static sigjmp_buf jb;
static void on_segv(int) { siglongjmp(jb, 1); }
__global__ void k_sum(const unsigned *p, int n, unsigned *out) {
    unsigned s = 0;
    for (int i = threadIdx.x; i < n; i += 256) s += p[i];
    atomicAdd(out, s);
}
int main() {
    signal(SIGSEGV, on_segv);
    signal(SIGBUS, on_segv);
    const size_t bytes = 128 << 10;
    unsigned *out;
    hipMalloc(&out, 4);
    const unsigned flags[] = {hipDeviceMallocFinegrained, hipDeviceMallocUncached, hipDeviceMallocDefault};
    const char *names[] = {"hipDeviceMallocFinegrained", "hipDeviceMallocUncached", "hipDeviceMallocDefault"};
    for (int f = 0; f < 3; f++) {
        unsigned *p = nullptr;
        if (hipExtMallocWithFlags((void **) &p, bytes, flags[f]) != hipSuccess) { printf("%s: alloc failed\n", names[f]); continue; }
        hipMemset(p, 0, bytes);
        hipDeviceSynchronize();
        if (sigsetjmp(jb, 1)) { printf("%s: host store faulted\n", names[f]); continue; }
        double best = 1e9;
        for (int rep = 0; rep < 5; rep++) {
            auto t0 = std::chrono::steady_clock::now();
            for (size_t i = 0; i < bytes / 4; i++) p[i] = (unsigned) (i + rep);
            __builtin_ia32_sfence();
            double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            best = dt < best ? dt : best;
        }
        hipMemset(out, 0, 4);
        k_sum<<<1, 256>>>(p, (int) (bytes / 4), out);
        unsigned got = 0, want = 0;
        hipMemcpy(&got, out, 4, hipMemcpyDeviceToHost);
        for (size_t i = 0; i < bytes / 4; i++) want += (unsigned) (i + 4);
        printf("%-28s host write of 128 KB: %.1f us; kernel sees %s\n", names[f], best * 1e6, got == want ? "the data" : "STALE/other data");
        hipFree(p);
    }
    return 0;
}
