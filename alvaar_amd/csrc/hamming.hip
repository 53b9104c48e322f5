// a7: brute-force Hamming matcher for 256-bit descriptors.
//
// Replaces cv::BFMatcher(NORM_HAMMING).match (core/src/batch_distance.cpp:199-251; norm.cpp:99-):
// per query the minimum popcount(q ^ t) over all train rows, LOWEST train index on ties (strict '<'
// at batch_distance.cpp:238).  Integer, bit-exact.
//
// This stage is VALU-bound, not HBM-bound (32(N+M)+8N bytes vs 24 N M ops), so the
// layout is about issue slots: one wave = 64 queries held in registers (8 dwords/lane); a chunk of
// 64 train descriptors is staged once in LDS and read back as wave-uniform (broadcast) b128 reads;
// per pair 8 v_xor + 8 v_bcnt (popcount-accumulate).  The N x M rectangle is cut into
// ceil(N/64) x ceil(M/64) single-wave workgroups (>= 1024 at 2000 x 2000) so all 256 CUs get work;
// each writes one packed key (dist << 20 | train_idx) per query, and a second tiny kernel takes the
// min over chunks -- min of the packed key is exactly "smallest distance, then lowest index".
#include "common.hpp"
#include <algorithm>

namespace {

constexpr int QT = 64;  // queries per workgroup (one wave)
constexpr int TC = 64;  // train descriptors per chunk

// dnq / dnt non-null: the counts are read from device memory (clamped to the launch's nq / nt capacities), so that the match can
// be enqueued before the host knows how many descriptors the detector produced
__global__ void __launch_bounds__(QT) k_bf_partial(const uint4 *__restrict__ q, int nq, const uint4 *__restrict__ t, int nt,
                                                   uint32_t *__restrict__ partial /* [chunks][nq_pad] */, int nq_pad,
                                                   const int *__restrict__ dnq, const int *__restrict__ dnt) {
    __shared__ uint4 s_t[TC * 2];
    if (dnq) {
        nq = min(nq, *dnq);
        nt = min(nt, *dnt);
        if ((int) blockIdx.x * QT >= nq || (int) blockIdx.y * TC >= nt) return;
    }
    const int lane = threadIdx.x;
    const int qi = blockIdx.x * QT + lane;
    const int t0 = blockIdx.y * TC;
    const int tcount = min(TC, nt - t0);
    if (lane < tcount) {
        s_t[2 * lane] = t[2 * (size_t) (t0 + lane)];
        s_t[2 * lane + 1] = t[2 * (size_t) (t0 + lane) + 1];
    }
    uint4 qa = make_uint4(0, 0, 0, 0), qb = qa;
    if (qi < nq) {
        qa = q[2 * (size_t) qi];
        qb = q[2 * (size_t) qi + 1];
    }
    __syncthreads();
    uint32_t best = 0xffffffffu;
    for (int j = 0; j < tcount; j++) {
        const uint4 ta = s_t[2 * j], tb = s_t[2 * j + 1];
        uint32_t d = __popc(qa.x ^ ta.x);
        d += __popc(qa.y ^ ta.y);
        d += __popc(qa.z ^ ta.z);
        d += __popc(qa.w ^ ta.w);
        d += __popc(qb.x ^ tb.x);
        d += __popc(qb.y ^ tb.y);
        d += __popc(qb.z ^ tb.z);
        d += __popc(qb.w ^ tb.w);
        uint32_t key = (d << 20) | (uint32_t) (t0 + j);
        best = min(best, key);
    }
    if (qi < nq) partial[(size_t) blockIdx.y * nq_pad + qi] = best;
}

__global__ void __launch_bounds__(256) k_bf_final(const uint32_t *__restrict__ partial, int chunks, int nq, int nq_pad,
                                                  int *__restrict__ idx, int *__restrict__ dist, const int *__restrict__ dnq,
                                                  const int *__restrict__ dnt) {
    if (dnq) {
        nq = min(nq, *dnq);
        chunks = min(chunks, (min(chunks * TC, *dnt) + TC - 1) / TC);
    }
    int qi = blockIdx.x * 256 + threadIdx.x;
    if (qi >= nq) return;
    uint32_t best = 0xffffffffu;
    for (int c = 0; c < chunks; c++) best = min(best, partial[(size_t) c * nq_pad + qi]);
    if (best == 0xffffffffu) {
        idx[qi] = -1;
        dist[qi] = -1;
    } else {
        idx[qi] = (int) (best & 0xfffffu);
        dist[qi] = (int) (best >> 20);
    }
}

// ---- several independent matches in one pair of launches (blockIdx.z = match) ------------------------------------------------
// The query count of a match is only known on the device (the detector has just produced it), so the query blocks are covered by a
// grid-stride loop instead of a grid sized for the capacity (tens of thousands of empty workgroups per camera otherwise).
struct BfItem {
    const uint4 *q, *t;
    const int *dnq;      // device: number of queries
    uint32_t *partial;   // [chunks][nq_pad]
    int *idx, *dist;
    int nq_cap, nt, nq_pad, chunks;
};

constexpr int BTPB = 4;
__global__ void __launch_bounds__(QT) k_bf_partial_b(const BfItem *__restrict__ items, int count, int gx, int gy) {
    __shared__ uint4 s_t[TC * BTPB * 2];
    const AlvaXcdItem w = alva_xcd_item(count, gx * gy);   // a camera's descriptor sets go through one XCD's L2
    if (w.cam >= count) return;
    const BfItem it = items[w.cam];
    const int bx = w.item % gx, by = w.item / gx;
    const int nt = it.nt, t0 = by * (TC * BTPB);
    if (t0 >= nt) return;  // also: a match without a train set (its count pointer may be null)
    const int nq = min(it.nq_cap, *it.dnq);
    const int lane = threadIdx.x;
    // BTPB train tiles per workgroup: the partial minima written for (and read back by) k_bf_final_b shrink by that factor; the
    // keys carry the train index, so the minimum over a larger set is the same minimum
    const int tcount = min(TC * BTPB, nt - t0);
    for (int j = lane; j < tcount; j += QT) {
        s_t[2 * j] = it.t[2 * (size_t) (t0 + j)];
        s_t[2 * j + 1] = it.t[2 * (size_t) (t0 + j) + 1];
    }
    __syncthreads();
    for (int qb = bx; qb * QT < nq; qb += gx) {
        const int qi = qb * QT + lane;
        uint4 qa = make_uint4(0, 0, 0, 0), qb4 = qa;
        if (qi < nq) {
            qa = it.q[2 * (size_t) qi];
            qb4 = it.q[2 * (size_t) qi + 1];
        }
        uint32_t best = 0xffffffffu;
        for (int j = 0; j < tcount; j++) {
            const uint4 ta = s_t[2 * j], tb = s_t[2 * j + 1];
            uint32_t d = __popc(qa.x ^ ta.x);
            d += __popc(qa.y ^ ta.y);
            d += __popc(qa.z ^ ta.z);
            d += __popc(qa.w ^ ta.w);
            d += __popc(qb4.x ^ tb.x);
            d += __popc(qb4.y ^ tb.y);
            d += __popc(qb4.z ^ tb.z);
            d += __popc(qb4.w ^ tb.w);
            const uint32_t key = (d << 20) | (uint32_t) (t0 + j);
            best = min(best, key);
        }
        if (qi < nq) it.partial[(size_t) by * it.nq_pad + qi] = best;
    }
}

__global__ void __launch_bounds__(256) k_bf_final_b(const BfItem *__restrict__ items, int count, int gx) {
    const AlvaXcdItem w = alva_xcd_item(count, gx);
    if (w.cam >= count) return;
    const BfItem it = items[w.cam];
    if (it.nt == 0) return;
    const int nq = min(it.nq_cap, *it.dnq);
    for (int qi = w.item * 256 + threadIdx.x; qi < nq; qi += gx * 256) {
        uint32_t best = 0xffffffffu;
        for (int c = 0; c < it.chunks; c++) best = min(best, it.partial[(size_t) c * it.nq_pad + qi]);
        if (best == 0xffffffffu) {
            it.idx[qi] = -1;
            it.dist[qi] = -1;
        } else {
            it.idx[qi] = (int) (best & 0xfffffu);
            it.dist[qi] = (int) (best >> 20);
        }
    }
}

}  // namespace

// `count` independent brute-force matches (cv::BFMatcher(NORM_HAMMING), batch_distance.cpp:199-251) in one pair of launches.
// Match c: queries d_query[c] (their number is read from device memory, *d_n_query[c], clamped to cap_query), train set d_train[c]
// with n_train[c] rows known to the host (0 = skip the match); rows of d_idx[c] / d_dist[c] beyond the query count stay untouched.
// expected_queries sizes the grid (more queries are covered by a loop).  Enqueue-only.
extern "C" int alva_bf_match_hamming_batch(alva_ctx *ctx, int count, const uint8_t *const *d_query, const int *const *d_n_query, int cap_query,
                                           const uint8_t *const *d_train, const int *n_train, int *const *d_idx, int *const *d_dist,
                                           int expected_queries) {
    ALVA_ARG(ctx && count > 0 && count <= 65535 && d_query && d_n_query && d_train && n_train && d_idx && d_dist && cap_query > 0);
    const int nq_pad = alva_divup(cap_query, QT) * QT;
    int max_chunks = 0;
    size_t partial_words = 0;
    std::vector<BfItem> items((size_t) count);
    for (int c = 0; c < count; c++) {
        ALVA_ARG(n_train[c] >= 0 && n_train[c] < (1 << 20));
        const int chunks = alva_divup(n_train[c], TC * BTPB);
        max_chunks = chunks > max_chunks ? chunks : max_chunks;
        partial_words += (size_t) chunks * nq_pad;
    }
    if (max_chunks == 0) return ALVA_OK;
    const size_t off_partial = (items.size() * sizeof(BfItem) + 255) / 256 * 256;
    uint8_t *dev = nullptr;
    int rc = alva_ctx_scratch(ctx, 0, off_partial + partial_words * sizeof(uint32_t), (void **) &dev);
    if (rc) return rc;
    uint32_t *partial = (uint32_t *) (dev + off_partial);
    for (int c = 0; c < count; c++) {
        BfItem &it = items[(size_t) c];
        it.nt = n_train[c];
        it.chunks = alva_divup(n_train[c], TC * BTPB);
        if (it.nt > 0) ALVA_ARG(d_query[c] && d_train[c] && d_n_query[c] && d_idx[c] && d_dist[c] && ((uintptr_t) d_query[c] % 16) == 0 && ((uintptr_t) d_train[c] % 16) == 0);
        it.q = (const uint4 *) d_query[c];
        it.t = (const uint4 *) d_train[c];
        it.dnq = d_n_query[c];
        it.partial = partial;
        it.idx = d_idx[c];
        it.dist = d_dist[c];
        it.nq_cap = it.nt > 0 ? cap_query : 0;
        it.nq_pad = nq_pad;
        partial += (size_t) it.chunks * nq_pad;
    }
    ALVA_HIP(hipMemcpyAsync(dev, items.data(), items.size() * sizeof(BfItem), hipMemcpyHostToDevice, ctx->stream));
    const int qblocks = std::max(1, std::min(alva_divup(cap_query, QT), alva_divup(std::max(expected_queries, 1), QT)));
    hipLaunchKernelGGL(k_bf_partial_b, dim3(alva_xcd_grid(count, qblocks * max_chunks)), dim3(QT), 0, ctx->stream, (const BfItem *) dev, count, qblocks,
                       max_chunks);
    const int fblocks = std::max(1, alva_divup(qblocks * QT, 256));
    hipLaunchKernelGGL(k_bf_final_b, dim3(alva_xcd_grid(count, fblocks)), dim3(256), 0, ctx->stream, (const BfItem *) dev, count, fblocks);
    ALVA_LAUNCH_CHECK();
    return ALVA_OK;
}

extern "C" int alva_bf_match_hamming(alva_ctx *ctx, const uint8_t *d_query, int n_query, const uint8_t *d_train,
                                     int n_train, int *d_idx, int *d_dist) {
    ALVA_ARG(ctx && n_query >= 0 && n_train >= 0 && n_train < (1 << 20));
    if (n_query == 0) return ALVA_OK;  // nothing to match: output pointers may be NULL
    ALVA_ARG(d_idx && d_dist);
    ALVA_ARG(d_query && ((uintptr_t) d_query % 16) == 0);
    ALVA_ARG(n_train == 0 || (d_train && ((uintptr_t) d_train % 16) == 0));
    int chunks = alva_divup(n_train, TC);
    int nq_pad = alva_divup(n_query, QT) * QT;
    uint32_t *partial = nullptr;
    if (chunks > 0) {
        int rc = alva_ctx_scratch(ctx, 0, (size_t) chunks * nq_pad * sizeof(uint32_t), (void **) &partial);
        if (rc) return rc;
        hipLaunchKernelGGL(k_bf_partial, dim3(nq_pad / QT, chunks), dim3(QT), 0, ctx->stream, (const uint4 *) d_query, n_query,
                           (const uint4 *) d_train, n_train, partial, nq_pad, (const int *) nullptr, (const int *) nullptr);
        ALVA_LAUNCH_CHECK();
    }
    hipLaunchKernelGGL(k_bf_final, dim3(alva_divup(n_query, 256)), dim3(256), 0, ctx->stream, partial, chunks, n_query, nq_pad,
                       d_idx, d_dist, (const int *) nullptr, (const int *) nullptr);
    ALVA_LAUNCH_CHECK();
    return ALVA_OK;
}

// The same match with the two counts still on the device (internal: the per-frame driver enqueues it right behind the detector).
// cap_query / cap_train bound the counts; rows of d_idx / d_dist beyond the actual query count are left untouched.
int alva_bf_match_hamming_devcount(alva_ctx *ctx, const uint8_t *d_query, const int *d_n_query, int cap_query, const uint8_t *d_train,
                                   const int *d_n_train, int cap_train, int *d_idx, int *d_dist) {
    ALVA_ARG(ctx && d_query && d_train && d_n_query && d_n_train && d_idx && d_dist && cap_query > 0 && cap_train > 0 && cap_train < (1 << 20));
    ALVA_ARG(((uintptr_t) d_query % 16) == 0 && ((uintptr_t) d_train % 16) == 0);
    const int chunks = alva_divup(cap_train, TC), nq_pad = alva_divup(cap_query, QT) * QT;
    uint32_t *partial = nullptr;
    int rc = alva_ctx_scratch(ctx, 0, (size_t) chunks * nq_pad * sizeof(uint32_t), (void **) &partial);
    if (rc) return rc;
    hipLaunchKernelGGL(k_bf_partial, dim3(nq_pad / QT, chunks), dim3(QT), 0, ctx->stream, (const uint4 *) d_query, cap_query,
                       (const uint4 *) d_train, cap_train, partial, nq_pad, d_n_query, d_n_train);
    hipLaunchKernelGGL(k_bf_final, dim3(alva_divup(cap_query, 256)), dim3(256), 0, ctx->stream, partial, chunks, cap_query, nq_pad, d_idx,
                       d_dist, d_n_query, d_n_train);
    ALVA_LAUNCH_CHECK();
    return ALVA_OK;
}
