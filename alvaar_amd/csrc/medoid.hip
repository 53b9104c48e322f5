// f1: MapPoint's descriptor tables and medoids on the device -- MapPoint::addDesc (map_point.cpp:131-181) and the
// descriptor half of MapPoint::removeObservedKeyframeId (:93-128) replayed from the map layer's operation log (slam/medoid_table.hpp holds
// the record layout, the list surgery of libstdc++'s unordered_map that decides ties, and the one-thread statement of both routines).
//
// One WAVEFRONT per touched map point: its record (2.6 KB) is pulled into LDS, the point's operations run in program order, the record goes
// back.  Inside an operation lane 0 does the list surgery (find / insert / rehash / erase: a few dependent LDS accesses) and lays the
// table's ITERATION ORDER out as an index array; then lane i owns the i-th entry in that order: one 256-bit Hamming distance against the
// operation's descriptor (4 x v_bcnt on LDS-resident bytes), its distance sum updated in place, and the reference's "first strict minimum
// while iterating" becomes a wave-wide minimum over (value, position) pairs -- equal values resolve to the earlier position, which is what
// the sequential scan does.  Sums of popcounts are exact in float, so the order of the additions does not matter.
#include "common.hpp"
#include "slam/medoid_table.hpp"
#include "slam/mp_rec.hpp"
#include <algorithm>
#include <vector>

using namespace alva_medoid;

namespace {

__device__ __forceinline__ unsigned long long wave_min_u64(unsigned long long v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const unsigned long long w = __shfl_xor(v, o, 64);
        v = w < v ? w : v;
    }
    return v;
}
__device__ __forceinline__ int wave_sum_i32(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
// order-preserving map float -> unsigned (for the (value, position) minimum of the removal, whose values are float sums)
__device__ __forceinline__ unsigned int float_key(float f) {
    const unsigned int u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

__device__ __forceinline__ void op_add(Table &t, int *order, const MedoidOp &o, const int lane) {
    __shared__ int s_new, s_count, s_had;
    if (lane == 0) {
        s_new = END;
        s_had = t.medoid_valid;
        if (find_slot(t, o.kf) == END) s_new = insert(t, o.kf, o.desc, o.rehash_to);
        s_count = t.count;
        if (s_new != END) {
            int i = 0;
            for (int s = t.head; s != END; s = t.slot[s].next) order[i++] = s;
        }
    }
    __syncthreads();
    const int sn = s_new, n = s_count;
    if (sn == END) return;
    if (n == 1) {
        if (lane < 32) t.medoid[lane] = o.desc[lane];
        if (lane == 0) {
            t.medoid_valid = 1;
            t.medoid_kf = o.kf;
        }
        __syncthreads();
        return;
    }
    int dist = 0;
    const bool mine = lane < n;
    const int s = mine ? order[lane] : 0;
    if (mine) {
        dist = popcount256(o.desc, t.slot[s].desc);
        if (s != sn) t.slot[s].dist += (float) dist;
    }
    const int nd = wave_sum_i32(mine ? dist : 0);
    // first strict minimum of `dist` in iteration order, against the initial bound desc_.cols * 8 (256 with a desc_, 0 without)
    const unsigned long long best = wave_min_u64(mine ? ((unsigned long long) (unsigned) dist << 32) | (unsigned) lane : ~0ull);
    const float bound = s_had ? 256.f : 0.f;
    float min_dist = bound;
    int min_slot = END;
    if ((float) (int) (best >> 32) < bound) {
        min_dist = (float) (int) (best >> 32);
        min_slot = order[(int) (best & 0xffffffffu)];
    }
    if ((float) nd < min_dist) min_slot = sn;   // :175-178
    __syncthreads();
    if (lane == 0) t.slot[sn].dist = (float) nd;
    if (min_slot != END) {
        if (lane < 32) t.medoid[lane] = t.slot[min_slot].desc[lane];
        if (lane == 0) {
            t.medoid_valid = 1;
            t.medoid_kf = t.slot[min_slot].key;
        }
    }
    __syncthreads();
}

__device__ __forceinline__ void op_remove(Table &t, int *order, const MedoidOp &o, const int lane) {
    __shared__ int s_sd, s_n, s_had;
    if (lane == 0) {
        s_sd = find_slot(t, o.kf);
        s_had = t.medoid_valid;
        int i = 0;
        if (s_sd != END)
            for (int s = t.head; s != END; s = t.slot[s].next)
                if (s != s_sd) order[i++] = s;
        s_n = i;
    }
    __syncthreads();
    const int sd = s_sd, n = s_n;
    if (sd == END) return;
    const bool mine = lane < n;
    const int s = mine ? order[lane] : 0;
    float left = 0.f;
    if (mine) {
        left = t.slot[s].dist - (float) popcount256(t.slot[sd].desc, t.slot[s].desc);
        t.slot[s].dist = left;
    }
    const unsigned long long best = wave_min_u64(mine ? ((unsigned long long) float_key(left) << 32) | (unsigned) lane : ~0ull);
    const float bound = s_had ? 256.f : 0.f;
    int min_slot = END;
    if (n > 0) {
        const int bl = (int) (best & 0xffffffffu);
        const float bv = __shfl(left, bl, 64);
        if (bv < bound) min_slot = order[bl];
    }
    __syncthreads();
    const int min_id = min_slot != END ? t.slot[min_slot].key : -1;
    if (min_id > 0) {   // sic: keyframe 0 is never chosen (:123)
        if (lane < 32) t.medoid[lane] = t.slot[min_slot].desc[lane];
        if (lane == 0) {
            t.medoid_valid = 1;
            t.medoid_kf = min_id;
        }
    }
    __syncthreads();
    if (lane == 0) erase_slot(t, sd);
    __syncthreads();
}

__global__ void __launch_bounds__(64) k_medoid_replay(Table *__restrict__ tables, const MedoidOp *__restrict__ ops, const int *__restrict__ mp_slot,
                                                      const int *__restrict__ first_op, int n_ops) {
    __shared__ Table t;
    __shared__ MedoidOp op;
    __shared__ int order[CAP];
    const int lane = (int) threadIdx.x;
    Table *g = tables + mp_slot[blockIdx.x];
    constexpr int W = (int) (sizeof(Table) / 4);
    for (int i = lane; i < W; i += 64) reinterpret_cast<int *>(&t)[i] = reinterpret_cast<const int *>(g)[i];
    __syncthreads();
    for (int oi = first_op[blockIdx.x]; oi >= 0 && oi < n_ops;) {
        if (lane < 16) reinterpret_cast<int *>(&op)[lane] = reinterpret_cast<const int *>(ops + oi)[lane];
        __syncthreads();
        const int next = op.next;
        if (op.op == OP_ADD) op_add(t, order, op, lane);
        else if (op.op == OP_REMOVE) op_remove(t, order, op, lane);
        else {
            if (lane == 0) {
                if (op.op == OP_CLEAR) clear_keep(t);
                else reset(t);
            }
            __syncthreads();
        }
        oi = next;
    }
    for (int i = lane; i < W; i += 64) reinterpret_cast<int *>(g)[i] = reinterpret_cast<const int *>(&t)[i];
}

__global__ void __launch_bounds__(256) k_medoid_reset(Table *tables, int first, int n) {
    const int i = first + (int) (blockIdx.x * 256 + threadIdx.x);
    if (i < first + n) reset(tables[i]);
}

// per requested slot: desc_ (32 B) | valid, overflow (2 B) + pad | #descriptors, keyframe of desc_  -> 48-byte records
__global__ void __launch_bounds__(256) k_medoid_export(const Table *__restrict__ tables, const int *__restrict__ slots, int n, uint8_t *__restrict__ out) {
    const int i = (int) (blockIdx.x * 256 + threadIdx.x);
    if (i >= n) return;
    const Table &t = tables[slots[i]];
    uint8_t *o = out + 48 * (size_t) i;
    for (int k = 0; k < 32; k++) o[k] = t.medoid[k];
    o[32] = (uint8_t) t.medoid_valid;
    o[33] = (uint8_t) t.overflow;
    reinterpret_cast<int *>(o + 36)[0] = t.count;
    reinterpret_cast<int *>(o + 36)[1] = t.medoid_kf;
}

// The shared-map exchange's record block (alvaar_amd/multi.py: stream id | point id | xyz 3 x f64 | descriptor medoid 32 B = 64 B) written
// straight from the resident data: one thread per record SLOT reads the map-point record's header out of pinned host memory (slam/mp_rec.hpp:
// id, is3d, !desc_.empty(), worldPoint_) and the medoid out of the descriptor table of the same slot; 3-D points with a descriptor claim
// an output row with one atomic (the consumers order the block by (stream, id) themselves).  Rows beyond `capacity` are counted, not
// written: the caller falls back to the host-side export, which keeps the OLDEST `capacity` points.
__global__ void __launch_bounds__(256) k_pack_map_records(const alva_slam::MpRec *const *__restrict__ chunks, int n_slots,
                                                          const Table *__restrict__ tables, int stream_id, int capacity,
                                                          uint8_t *__restrict__ out, int *__restrict__ count) {
    const int s = (int) (blockIdx.x * 256 + threadIdx.x);
    if (s >= n_slots) return;
    const alva_slam::MpRec *r = chunks[s >> alva_slam::MP_CHUNK_SHIFT] + (s & (alva_slam::MP_CHUNK - 1));
    const unsigned long long *h8 = reinterpret_cast<const unsigned long long *>(r);
    const unsigned long long ida = __builtin_nontemporal_load(h8 + 4), fl = __builtin_nontemporal_load(h8 + 5);
    const int id = (int) (unsigned) (ida & 0xffffffffull);
    if (id < 0 || !(fl & 0xffull) || !((fl >> 8) & 0xffull)) return;   // free record | not 3-D | no descriptor
    const Table &t = tables[s];
    if (!t.medoid_valid || t.count <= 0) return;   // (at least one keyframe descriptor: what the host-side export requires too)
    const int row = atomicAdd(count, 1);
    if (row >= capacity) return;
    uint8_t *o = out + 64 * (size_t) row;
    reinterpret_cast<int *>(o)[0] = stream_id;
    reinterpret_cast<int *>(o)[1] = id;
    for (int k = 0; k < 3; k++) reinterpret_cast<unsigned long long *>(o + 8)[k] = __builtin_nontemporal_load(h8 + k);
    for (int k = 0; k < 4; k++) reinterpret_cast<unsigned long long *>(o + 32)[k] = reinterpret_cast<const unsigned long long *>(t.medoid)[k];
}
__global__ void __launch_bounds__(256) k_pack_fill_unused(uint8_t *__restrict__ out, const int *__restrict__ count, int capacity) {
    const int row = (int) (blockIdx.x * 256 + threadIdx.x);
    if (row >= capacity || row < *count) return;
    unsigned long long *o = reinterpret_cast<unsigned long long *>(out + 64 * (size_t) row);
    for (int k = 0; k < 8; k++) o[k] = 0ull;
    reinterpret_cast<int *>(o)[1] = -1;   // unused row: point id -1
}

}  // namespace

struct alva_medoid_store {
    alva_ctx *ctx = nullptr;
    Table *tables = nullptr;
    int cap = 0;
    // two pinned staging buffers for the logs (the kernel reads the log where it lies): a buffer is reused once the event behind the
    // replay that read it has passed
    struct Stage {
        uint8_t *host = nullptr;
        size_t bytes = 0;
        hipEvent_t done = nullptr;
        bool busy = false;
    } stage[2];
    int next_stage = 0;
};

extern "C" size_t alva_medoid_table_bytes(void) { return sizeof(Table); }
extern "C" size_t alva_medoid_op_bytes(void) { return sizeof(MedoidOp); }

extern "C" int alva_medoid_store_create(alva_ctx *ctx, alva_medoid_store **out) {
    ALVA_ARG(ctx && out);
    *out = new alva_medoid_store();
    (*out)->ctx = ctx;
    return ALVA_OK;
}

extern "C" void alva_medoid_store_destroy(alva_medoid_store *s) {
    if (!s) return;
    (void) hipSetDevice(s->ctx->device);
    (void) hipStreamSynchronize(s->ctx->stream);
    for (auto &g: s->stage) {
        if (g.done) (void) hipEventDestroy(g.done);
        if (g.host) (void) hipHostFree(g.host);
    }
    if (s->tables) (void) hipFree(s->tables);
    delete s;
}

static int medoid_reserve(alva_medoid_store *s, int slots) {
    if (slots <= s->cap) return ALVA_OK;
    int cap = std::max(s->cap, 8192);
    while (cap < slots) cap *= 2;
    Table *nt = nullptr;
    ALVA_HIP(hipMalloc((void **) &nt, (size_t) cap * sizeof(Table)));
    hipStream_t st = s->ctx->stream;
    if (s->cap > 0) ALVA_HIP(hipMemcpyAsync(nt, s->tables, (size_t) s->cap * sizeof(Table), hipMemcpyDeviceToDevice, st));
    hipLaunchKernelGGL(k_medoid_reset, dim3((unsigned) alva_divup(cap - s->cap, 256)), dim3(256), 0, st, nt, s->cap, cap - s->cap);
    ALVA_LAUNCH_CHECK();
    if (s->tables) {
        ALVA_HIP(alva_stream_sync(st));   // (growth: rare) the old block is not freed under a kernel that may still read it
        ALVA_HIP(hipFree(s->tables));
    }
    s->tables = nt;
    s->cap = cap;
    return ALVA_OK;
}

// enqueue only: the log is copied into pinned staging, the replay kernel reads it from there
extern "C" int alva_medoid_replay(alva_medoid_store *s, int n_ops, const void *ops, int n_mp, const int *mp_slot, const int *first_op, int slots) {
    ALVA_ARG(s && n_ops >= 0 && n_mp >= 0 && slots >= 0 && (n_ops == 0 || ops) && (n_mp == 0 || (mp_slot && first_op)));
    ALVA_HIP(hipSetDevice(s->ctx->device));
    int rc = medoid_reserve(s, slots);
    if (rc) return rc;
    if (n_mp == 0 || n_ops == 0) return ALVA_OK;
    for (int i = 0; i < n_mp; i++) ALVA_ARG(mp_slot[i] >= 0 && mp_slot[i] < s->cap && first_op[i] >= 0 && first_op[i] < n_ops);
    const size_t op_bytes = (size_t) n_ops * sizeof(MedoidOp), idx_bytes = ((size_t) n_mp * 4 + 63) / 64 * 64;
    const size_t need = op_bytes + 2 * idx_bytes;
    alva_medoid_store::Stage &g = s->stage[s->next_stage];
    s->next_stage ^= 1;
    if (g.busy) {
        ALVA_HIP(alva_event_sync(g.done));
        g.busy = false;
    }
    if (g.bytes < need) {
        if (g.host) ALVA_HIP(hipHostFree(g.host));
        g.host = nullptr;
        g.bytes = 0;
        ALVA_HIP(hipHostMalloc((void **) &g.host, need + need / 2, hipHostMallocDefault));
        g.bytes = need + need / 2;
    }
    if (!g.done) ALVA_HIP(hipEventCreateWithFlags(&g.done, hipEventDisableTiming));
    memcpy(g.host, ops, op_bytes);
    memcpy(g.host + op_bytes, mp_slot, (size_t) n_mp * 4);
    memcpy(g.host + op_bytes + idx_bytes, first_op, (size_t) n_mp * 4);
    hipStream_t st = s->ctx->stream;
    hipLaunchKernelGGL(k_medoid_replay, dim3((unsigned) n_mp), dim3(64), 0, st, s->tables, (const MedoidOp *) g.host, (const int *) (g.host + op_bytes),
                       (const int *) (g.host + op_bytes + idx_bytes), n_ops);
    ALVA_LAUNCH_CHECK();
    ALVA_HIP(hipEventRecord(g.done, st));
    g.busy = true;
    return ALVA_OK;
}

// waits for the replays enqueued so far
extern "C" int alva_medoid_export(alva_medoid_store *s, int n, const int *mp_slot, uint8_t *desc32, uint8_t *valid, int *info3) {
    ALVA_ARG(s && n >= 0 && (n == 0 || mp_slot));
    if (n == 0) return ALVA_OK;
    ALVA_HIP(hipSetDevice(s->ctx->device));
    for (int i = 0; i < n; i++) ALVA_ARG(mp_slot[i] >= 0 && mp_slot[i] < s->cap);
    uint8_t *d = nullptr, *pin = nullptr;
    const size_t idx_bytes = ((size_t) n * 4 + 255) / 256 * 256;
    int rc = alva_ctx_scratch(s->ctx, 11, idx_bytes + (size_t) n * 48, (void **) &d);
    if (rc) return rc;
    rc = alva_ctx_pinned(s->ctx, idx_bytes + (size_t) n * 48, (void **) &pin);
    if (rc) return rc;
    hipStream_t st = s->ctx->stream;
    ALVA_HIP(alva_stream_sync(st));   // nothing enqueued earlier may still use the context's staging
    memcpy(pin, mp_slot, (size_t) n * 4);
    ALVA_HIP(hipMemcpyAsync(d, pin, (size_t) n * 4, hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(k_medoid_export, dim3((unsigned) alva_divup(n, 256)), dim3(256), 0, st, (const Table *) s->tables, (const int *) d, n, d + idx_bytes);
    ALVA_LAUNCH_CHECK();
    ALVA_HIP(hipMemcpyAsync(pin + idx_bytes, d + idx_bytes, (size_t) n * 48, hipMemcpyDeviceToHost, st));
    ALVA_HIP(alva_stream_sync(st));
    for (int i = 0; i < n; i++) {
        const uint8_t *o = pin + idx_bytes + 48 * (size_t) i;
        if (desc32) memcpy(desc32 + 32 * (size_t) i, o, 32);
        if (valid) valid[i] = o[32];
        if (info3) {
            info3[3 * i] = reinterpret_cast<const int *>(o + 36)[0];
            info3[3 * i + 1] = reinterpret_cast<const int *>(o + 36)[1];
            info3[3 * i + 2] = o[33];
        }
    }
    return ALVA_OK;
}

extern "C" const void *alva_medoid_tables(alva_medoid_store *s) { return s ? s->tables : nullptr; }

extern "C" int alva_pack_map_records(alva_medoid_store *s, const void *const *d_record_chunks, int n_slots, int stream_id, int capacity, uint8_t *d_out,
                                     int *h_count) {
    ALVA_ARG(s && d_record_chunks && n_slots >= 0 && capacity >= 0 && d_out && h_count && n_slots <= s->cap);
    ALVA_HIP(hipSetDevice(s->ctx->device));
    hipStream_t st = s->ctx->stream;
    int *d_count = nullptr, *pin = nullptr;
    int rc = alva_ctx_scratch(s->ctx, 11, 256, (void **) &d_count);
    if (rc) return rc;
    rc = alva_ctx_pinned(s->ctx, 256, (void **) &pin);
    if (rc) return rc;
    ALVA_HIP(alva_stream_sync(st));   // nothing enqueued earlier may still use the context's staging
    ALVA_HIP(hipMemsetAsync(d_count, 0, 4, st));
    if (n_slots > 0)
        hipLaunchKernelGGL(k_pack_map_records, dim3((unsigned) alva_divup(n_slots, 256)), dim3(256), 0, st,
                           reinterpret_cast<const alva_slam::MpRec *const *>(d_record_chunks), n_slots, (const Table *) s->tables, stream_id, capacity, d_out, d_count);
    if (capacity > 0) hipLaunchKernelGGL(k_pack_fill_unused, dim3((unsigned) alva_divup(capacity, 256)), dim3(256), 0, st, d_out, (const int *) d_count, capacity);
    ALVA_LAUNCH_CHECK();
    ALVA_HIP(hipMemcpyAsync(pin, d_count, 4, hipMemcpyDeviceToHost, st));
    ALVA_HIP(alva_stream_sync(st));
    *h_count = *pin;
    return ALVA_OK;
}

extern "C" int alva_medoid_dump(alva_medoid_store *s, int mp_slot, void *table_out, size_t bytes) {
    ALVA_ARG(s && table_out && bytes == sizeof(Table) && mp_slot >= 0 && mp_slot < s->cap);
    ALVA_HIP(hipSetDevice(s->ctx->device));
    ALVA_HIP(alva_stream_sync(s->ctx->stream));
    ALVA_HIP(hipMemcpy(table_out, s->tables + mp_slot, sizeof(Table), hipMemcpyDeviceToHost));
    return ALVA_OK;
}
