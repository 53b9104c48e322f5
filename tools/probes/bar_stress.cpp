// Stress of the host-writes-device-memory path the slot table uses: every iteration the host stores a new pattern into an uncached device
// block (ordinary stores + sfence), launches a kernel that checks it (wave-uniform scalar-style reads AND per-lane reads) and counts
// mismatches; optionally alternating with a second stream / thread.   hipcc -O2 --offload-arch=gfx950 ... && /tmp/bar_stress
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>
__global__ void k_check(const unsigned *p, int n, unsigned want, unsigned *bad) {
    // block b checks element b with a wave-uniform index (scalar load) and lane l element (b + l) % n
    const int b = blockIdx.x;
    const unsigned u = p[b];
    const unsigned v = p[(b + threadIdx.x) % n];
    if (u != want + (unsigned) b) atomicAdd(bad, 1u);
    if (v != want + (unsigned) ((b + threadIdx.x) % n)) atomicAdd(bad + 1, 1u);
}
__global__ void k_busy(unsigned *sink, int spins) {
    unsigned v = threadIdx.x;
    for (int i = 0; i < spins; i++) v = v * 1664525u + 1013904223u;
    if (v == 12345u) *sink = v;
}
// the real pattern: the stream is BUSY (earlier kernels of the frame) while the host stores the table; the reading kernel is enqueued behind
// them without a host synchronisation in between; the same addresses were read by the previous iteration's kernel a moment ago
static void run_busy(int n, int iters, unsigned flags, const char *name) {
    unsigned *p = nullptr, *bad = nullptr, *sink = nullptr;
    hipExtMallocWithFlags((void **) &p, (size_t) n * 4, flags);
    hipMalloc(&bad, 8);
    hipMalloc(&sink, 4);
    hipMemset(bad, 0, 8);
    hipStream_t st = nullptr;
    hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
    hipEvent_t ev;
    hipEventCreateWithFlags(&ev, hipEventDisableTiming);
    for (int it = 0; it < iters; it++) {
        const unsigned want = (unsigned) it * 7919u;
        hipLaunchKernelGGL(k_busy, dim3(256), dim3(256), 0, st, sink, 4000);   // ~20 us of other work in front
        for (int i = 0; i < n; i++) p[i] = want + (unsigned) i;
        __builtin_ia32_sfence();
        hipLaunchKernelGGL(k_check, dim3(n), dim3(64), 0, st, (const unsigned *) p, n, want, bad);
        hipEventRecord(ev, st);
        while (hipEventQuery(ev) != hipSuccess) {}
    }
    unsigned h[2];
    hipMemcpy(h, bad, 8, hipMemcpyDeviceToHost);
    printf("%-12s n=%6d iters=%d busy stream, no sync before the reader: uniform-read mismatches %u, per-lane mismatches %u\n", name, n, iters, h[0], h[1]);
    hipFree(p); hipFree(bad); hipFree(sink);
}
static void run(int n, int iters, unsigned flags, const char *name, bool own_stream, bool churn) {
    unsigned *p = nullptr, *bad = nullptr;
    hipExtMallocWithFlags((void **) &p, (size_t) n * 4, flags);
    hipMalloc(&bad, 8);
    hipMemset(bad, 0, 8);
    hipStream_t st = nullptr;
    if (own_stream) hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
    std::vector<void *> junk;
    for (int it = 0; it < iters; it++) {
        const unsigned want = (unsigned) it * 7919u;
        for (int i = 0; i < n; i++) p[i] = want + (unsigned) i;
        __builtin_ia32_sfence();
        hipLaunchKernelGGL(k_check, dim3(n), dim3(64), 0, st, (const unsigned *) p, n, want, bad);
        hipStreamSynchronize(st);
        if (churn && (it % 64) == 0) {   // allocator churn beside it
            void *q = nullptr;
            hipMalloc(&q, 1 << 20);
            hipMemsetAsync(q, 1, 1 << 20, st);
            junk.push_back(q);
            if (junk.size() > 4) { hipStreamSynchronize(st); hipFree(junk.front()); junk.erase(junk.begin()); }
        }
    }
    unsigned h[2];
    hipMemcpy(h, bad, 8, hipMemcpyDeviceToHost);
    printf("%-12s n=%6d iters=%d stream=%d churn=%d: uniform-read mismatches %u, per-lane mismatches %u\n", name, n, iters, own_stream, churn, h[0], h[1]);
    for (void *q: junk) hipFree(q);
    hipFree(p); hipFree(bad);
}
int main() {
    run_busy(380, 20000, hipDeviceMallocUncached, "uncached");
    run_busy(5200, 5000, hipDeviceMallocUncached, "uncached");
    run_busy(380, 20000, hipDeviceMallocDefault, "default");
    run(380, 20000, hipDeviceMallocUncached, "uncached", true, false);
    run(380, 20000, hipDeviceMallocUncached, "uncached", true, true);
    run(5200, 5000, hipDeviceMallocUncached, "uncached", true, true);
    run(380, 20000, hipDeviceMallocDefault, "default", true, true);
    run(380, 20000, hipDeviceMallocFinegrained, "finegrained", true, true);
    return 0;
}
