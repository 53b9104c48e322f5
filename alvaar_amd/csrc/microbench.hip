// Measured ceilings that MI355X_MICROARCH.md does not list and the roofline lines of bench.py need: the FP64 matrix rate
// (v_mfma_f64_16x16x4_f64 -- the instruction the Schur-complement GEMM and the reduced-camera Cholesky of ba.hip run on) and the plain
// integer VALU rate (v_xor / v_bcnt / v_add -- the Hamming matchers).  Independent accumulator chains, every SIMD of the chip busy.
#include "common.hpp"
#include <chrono>

namespace {

typedef double double4_t __attribute__((ext_vector_type(4)));

__global__ void __launch_bounds__(256) k_mfma_f64_peak(int iters, double *sink) {
    double4_t acc[8];
#pragma unroll
    for (int u = 0; u < 8; u++) acc[u] = double4_t{0, 0, 0, 0};
    double a = 1.0 + threadIdx.x * 1e-9, b = 1.0 - threadIdx.x * 1e-9;
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int u = 0; u < 8; u++) acc[u] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[u], 0, 0, 0);
    }
    double s = 0;
#pragma unroll
    for (int u = 0; u < 8; u++) s += acc[u][0] + acc[u][1] + acc[u][2] + acc[u][3];
    if (s == 123.456) sink[0] = s;
}

__global__ void __launch_bounds__(256) k_valu_int_peak(int iters, unsigned *sink) {
    unsigned x0 = threadIdx.x, x1 = threadIdx.x * 3u + 1u, x2 = blockIdx.x, x3 = 7u, a0 = 0, a1 = 0, a2 = 0, a3 = 0;
    const unsigned k0 = 0x9e3779b9u + blockIdx.x, k1 = 0x85ebca6bu, k2 = 0xc2b2ae35u, k3 = 0x27d4eb2fu;
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int u = 0; u < 8; u++) {  // the matcher's inner triple: xor, popcount-accumulate (v_bcnt_u32_b32 adds), 4 independent chains
            a0 = __builtin_popcount(x0 ^ k0) + a0;
            a1 = __builtin_popcount(x1 ^ k1) + a1;
            a2 = __builtin_popcount(x2 ^ k2) + a2;
            a3 = __builtin_popcount(x3 ^ k3) + a3;
            x0 += a1; x1 += a2; x2 += a3; x3 += a0;
        }
    }
    if ((a0 ^ a1 ^ a2 ^ a3) == 0x12345u) sink[0] = a0;
}

__global__ void k_null(int *sink) {
    if (sink && threadIdx.x == 12345) sink[0] = 1;
}

}  // namespace

// Launch-latency figures for the end-to-end bound of SURVEY.md 8(d) ("~10-15 kernel launches ~ 50-100 us"): the time per kernel of a
// chain of `chain` DEPENDENT empty launches on one stream (each waits for its predecessor: what a stage chain pays per link), and
// the round trip of one empty launch + stream synchronisation (what a host decision between two stages pays).  Microseconds.
extern "C" int alva_microbench_launch(alva_ctx *ctx, int chain, double *h_us_per_dependent_launch, double *h_us_launch_sync_roundtrip) {
    ALVA_ARG(ctx && chain > 0);
    void *sink = nullptr;
    int rc = alva_ctx_scratch(ctx, 9, 256, &sink);
    if (rc) return rc;
    for (int i = 0; i < 8; i++) hipLaunchKernelGGL(k_null, dim3(1), dim3(64), 0, ctx->stream, (int *) sink);
    ALVA_HIP(hipStreamSynchronize(ctx->stream));
    if (h_us_per_dependent_launch) {
        const auto t0 = std::chrono::steady_clock::now();
        for (int i = 0; i < chain; i++) hipLaunchKernelGGL(k_null, dim3(1), dim3(64), 0, ctx->stream, (int *) sink);
        ALVA_HIP(hipStreamSynchronize(ctx->stream));
        *h_us_per_dependent_launch = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / chain;
    }
    if (h_us_launch_sync_roundtrip) {
        const int reps = 50;
        const auto t0 = std::chrono::steady_clock::now();
        for (int i = 0; i < reps; i++) {
            hipLaunchKernelGGL(k_null, dim3(1), dim3(64), 0, ctx->stream, (int *) sink);
            ALVA_HIP(hipStreamSynchronize(ctx->stream));
        }
        *h_us_launch_sync_roundtrip = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / reps;
    }
    return ALVA_OK;
}

// *h_tflops = FP64 MFMA rate (2 * 16 * 16 * 4 flop per instruction per wave); *h_tops = integer VALU lane-operations per second / 1e12
// (3 counted instructions per inner triple: xor, bcnt-accumulate, add).  Synchronous; ~10 ms.
extern "C" int alva_microbench_peaks(alva_ctx *ctx, double *h_tflops_mfma_f64, double *h_tops_valu_int) {
    ALVA_ARG(ctx);
    void *sink = nullptr;
    int rc = alva_ctx_scratch(ctx, 9, 256, &sink);
    if (rc) return rc;
    hipEvent_t e0, e1;
    ALVA_HIP(hipEventCreate(&e0));
    ALVA_HIP(hipEventCreate(&e1));
    const int blocks = 256 * 8, iters = 2000;   // 8 workgroups of 4 waves per CU
    float ms = 0.f;
    if (h_tflops_mfma_f64) {
        hipLaunchKernelGGL(k_mfma_f64_peak, dim3(blocks), dim3(256), 0, ctx->stream, 10, (double *) sink);
        ALVA_HIP(hipEventRecord(e0, ctx->stream));
        hipLaunchKernelGGL(k_mfma_f64_peak, dim3(blocks), dim3(256), 0, ctx->stream, iters, (double *) sink);
        ALVA_HIP(hipEventRecord(e1, ctx->stream));
        ALVA_HIP(hipEventSynchronize(e1));
        ALVA_HIP(hipEventElapsedTime(&ms, e0, e1));
        *h_tflops_mfma_f64 = (double) blocks * 4 * iters * 8 * 2048.0 / (ms * 1e-3) / 1e12;
    }
    if (h_tops_valu_int) {
        hipLaunchKernelGGL(k_valu_int_peak, dim3(blocks), dim3(256), 0, ctx->stream, 10, (unsigned *) sink);
        ALVA_HIP(hipEventRecord(e0, ctx->stream));
        hipLaunchKernelGGL(k_valu_int_peak, dim3(blocks), dim3(256), 0, ctx->stream, iters, (unsigned *) sink);
        ALVA_HIP(hipEventRecord(e1, ctx->stream));
        ALVA_HIP(hipEventSynchronize(e1));
        ALVA_HIP(hipEventElapsedTime(&ms, e0, e1));
        *h_tops_valu_int = (double) blocks * 256 * iters * 8 * 4 * 3.0 / (ms * 1e-3) / 1e12;
    }
    (void) hipEventDestroy(e0);
    (void) hipEventDestroy(e1);
    return ALVA_OK;
}

unsigned long long *alva_kstamp_buffer() {
    static unsigned long long *buf = [] {
        unsigned long long *b = nullptr;
        if (getenv("ALVA_KSTAMPS") && hipMalloc((void **) &b, 4096 * 8) == hipSuccess) (void) hipMemset(b, 0, 4096 * 8);
        return b;
    }();
    return buf;
}

// Debug: per-SLOT stamps of the tracker launch (ALVA_KLT_STAMPS=1): entry i = slot i's wall time in the kernel (100 MHz ticks, low 32 bits)
// | result code << 32 | tracked-from-projection << 36 | retried << 37, overwritten by every launch; 16384 entries
unsigned long long *alva_klt_stamp_buffer() {
    static unsigned long long *buf = [] {
        unsigned long long *b = nullptr;
        if (getenv("ALVA_KLT_STAMPS") && hipMalloc((void **) &b, 16384 * 8) == hipSuccess) (void) hipMemset(b, 0, 16384 * 8);
        return b;
    }();
    return buf;
}
extern "C" int alva_debug_klt_stamps(unsigned long long *h_out16384) {
    ALVA_ARG(h_out16384);
    unsigned long long *b = alva_klt_stamp_buffer();
    if (!b) {
        alva_set_error("alva_debug_klt_stamps: the process was not started with ALVA_KLT_STAMPS=1");
        return ALVA_ERR_STATE;
    }
    ALVA_HIP(hipDeviceSynchronize());
    ALVA_HIP(hipMemcpy(h_out16384, b, 16384 * 8, hipMemcpyDeviceToHost));
    return ALVA_OK;
}

// copies the stamp buffer (4096 x u64; see common.hpp) to the host and clears it; ALVA_ERR_STATE without ALVA_KSTAMPS=1
extern "C" int alva_debug_kstamps(unsigned long long *h_out) {
    ALVA_ARG(h_out);
    unsigned long long *b = alva_kstamp_buffer();
    if (!b) {
        alva_set_error("alva_debug_kstamps: the process was not started with ALVA_KSTAMPS=1");
        return ALVA_ERR_STATE;
    }
    ALVA_HIP(hipDeviceSynchronize());
    ALVA_HIP(hipMemcpy(h_out, b, 4096 * 8, hipMemcpyDeviceToHost));
    ALVA_HIP(hipMemset(b, 0, 4096 * 8));
    return ALVA_OK;
}

