// §8(e) optional shared-map merge (north_star extension; the reference has ONE map, so this has no reference behaviour to match --
// PARITY UNPINNED): after the all_gather of every stream's map-point records, a point is absorbed by the EARLIEST surviving record
// of ANOTHER stream that lies within max_dist and whose 256-bit descriptor is within max_hamming bits -- the smallest Hamming distance
// wins, the earliest record on ties.  The rule follows the intent of MapManager::mergeMapPoints (src/slam/src/map_manager.cpp:428-513:
// the older point absorbs the newer) with mapMaxDescriptorDistance_ (state.hpp:60: 0.2 x 256 bits).
//
// "Surviving" makes the rule sequential in record order; it is evaluated as the fixed point of that recurrence instead:
//   k_fuse_candidates   one wavefront per record: all earlier records of other streams within the radius and the Hamming bound
//                       (ballot-compacted, ascending record index), up to 32 per record
//   k_fuse_round        one thread per record: absorbed = best SURVIVING candidate; repeated until nothing changes -- every record's
//                       verdict depends on earlier records only, so the iteration reaches the sequential result after at most
//                       (longest dependency chain) rounds: the number of streams, in practice
// Records must be sorted by (stream, point id).  Integer / IEEE double arithmetic only: the result is exact and order-independent.
#include "common.hpp"

namespace {

constexpr int FUSE_K = 32;

__device__ __forceinline__ int ham256(const uint4 *a, const uint4 *b) {
    const uint4 a0 = a[0], a1 = a[1], b0 = b[0], b1 = b[1];
    return __popc(a0.x ^ b0.x) + __popc(a0.y ^ b0.y) + __popc(a0.z ^ b0.z) + __popc(a0.w ^ b0.w) + __popc(a1.x ^ b1.x) + __popc(a1.y ^ b1.y) +
           __popc(a1.z ^ b1.z) + __popc(a1.w ^ b1.w);
}

__global__ void __launch_bounds__(256) k_fuse_candidates(int n, const int *__restrict__ stream, const double *__restrict__ xyz,
                                                         const uint8_t *__restrict__ desc, double max_dist2, int max_hamming,
                                                         int *__restrict__ cand, int *__restrict__ cand_n, int *__restrict__ overflow) {
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (i >= n) return;
    const int si = stream[i];
    const double x = xyz[3 * (size_t) i], y = xyz[3 * (size_t) i + 1], z = xyz[3 * (size_t) i + 2];
    const uint4 *di = reinterpret_cast<const uint4 *>(desc + 32 * (size_t) i);
    int count = 0;
    for (int j0 = 0; j0 < i; j0 += 64) {
        const int j = j0 + lane;
        int h = 0;
        bool ok = false;
        if (j < i && stream[j] != si) {
            const double dx = xyz[3 * (size_t) j] - x, dy = xyz[3 * (size_t) j + 1] - y, dz = xyz[3 * (size_t) j + 2] - z;
            if ((dx * dx + dy * dy) + dz * dz <= max_dist2) {
                h = ham256(di, reinterpret_cast<const uint4 *>(desc + 32 * (size_t) j));
                ok = h <= max_hamming;
            }
        }
        const unsigned long long m = __ballot(ok);
        if (ok) {
            const int pos = count + __popcll(m & ((1ull << lane) - 1ull));
            if (pos < FUSE_K) cand[(size_t) i * FUSE_K + pos] = (h << 24) | j;   // n < 2^24 records
        }
        count += __popcll(m);
    }
    if (lane == 0) {
        cand_n[i] = count < FUSE_K ? count : FUSE_K;
        if (count > FUSE_K) atomicAdd(overflow, 1);
    }
}

__global__ void __launch_bounds__(256) k_fuse_round(int n, const int *__restrict__ cand, const int *__restrict__ cand_n,
                                                    const uint8_t *__restrict__ keep_in, uint8_t *__restrict__ keep_out,
                                                    int *__restrict__ absorbed, int *__restrict__ changed) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    int best = -1, best_h = 1 << 20;
    const int c = cand_n[i];
    for (int k = 0; k < c; k++) {
        const int v = cand[(size_t) i * FUSE_K + k], j = v & 0xffffff, h = v >> 24;
        if (keep_in[j] && h < best_h) {  // ascending j: the earliest record wins ties
            best_h = h;
            best = j;
        }
    }
    const uint8_t k = best < 0;
    keep_out[i] = k;
    absorbed[i] = best;
    if (k != keep_in[i]) *changed = 1;
}

}  // namespace

extern "C" int alva_fuse_map_points(alva_ctx *ctx, int n, const int *d_stream, const double *d_xyz, const uint8_t *d_desc, double max_dist,
                                    int max_hamming, uint8_t *d_keep, int *d_absorbed_by, int *h_rounds) {
    ALVA_ARG(ctx && n >= 0 && n < (1 << 24) && max_dist >= 0 && max_hamming >= 0 && max_hamming <= 256);
    if (h_rounds) *h_rounds = 0;
    if (n == 0) return ALVA_OK;
    ALVA_ARG(d_stream && d_xyz && d_desc && d_keep && d_absorbed_by && ((uintptr_t) d_desc % 16) == 0);
    uint8_t *base = nullptr;
    const size_t off_n = (size_t) n * FUSE_K * 4, off_keep = off_n + (size_t) n * 4, off_flag = (off_keep + (size_t) n + 255) / 256 * 256;
    int rc = alva_ctx_scratch(ctx, 8, off_flag + 256, (void **) &base);
    if (rc) return rc;
    int *cand = (int *) base, *cand_n = (int *) (base + off_n), *flags = (int *) (base + off_flag);
    uint8_t *keep_b = base + off_keep;
    ALVA_HIP(hipMemsetAsync(flags, 0, 8, ctx->stream));
    ALVA_HIP(hipMemsetAsync(d_keep, 1, (size_t) n, ctx->stream));
    hipLaunchKernelGGL(k_fuse_candidates, dim3(alva_divup(n, 4)), dim3(256), 0, ctx->stream, n, d_stream, d_xyz, d_desc, max_dist * max_dist,
                       max_hamming, cand, cand_n, flags + 1);
    int *pin = nullptr;
    rc = alva_ctx_pinned(ctx, 64, (void **) &pin);
    if (rc) return rc;
    uint8_t *a = d_keep, *b = keep_b;
    int rounds = 0;
    for (;;) {
        ALVA_HIP(hipMemsetAsync(flags, 0, 4, ctx->stream));
        hipLaunchKernelGGL(k_fuse_round, dim3(alva_divup(n, 256)), dim3(256), 0, ctx->stream, n, cand, cand_n, a, b, d_absorbed_by, flags);
        ALVA_LAUNCH_CHECK();
        ALVA_HIP(hipMemcpyAsync(pin, flags, 8, hipMemcpyDeviceToHost, ctx->stream));
        ALVA_HIP(hipStreamSynchronize(ctx->stream));
        rounds++;
        uint8_t *t = a;
        a = b;
        b = t;
        if (pin[1]) {
            alva_set_error("alva_fuse_map_points: more than %d candidates for one record", FUSE_K);
            return ALVA_ERR_STATE;
        }
        if (!pin[0] || rounds > n) break;
    }
    if (a != d_keep) {
        // "synchronous" in the header: the caller may read d_keep on ANY stream when this returns
        ALVA_HIP(hipMemcpyAsync(d_keep, a, (size_t) n, hipMemcpyDeviceToDevice, ctx->stream));
        ALVA_HIP(hipStreamSynchronize(ctx->stream));
    }
    if (h_rounds) *h_rounds = rounds;
    return ALVA_OK;
}
