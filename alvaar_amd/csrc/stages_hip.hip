// The numeric stages of the map layer (slam/stages.hpp) on the GPU: every method is one or more calls through
// include/alvaar_hip.h on this object's HIP stream.  This is the only `Stages` implementation in libalvaar_hip.so; there is no
// CPU path -- creation fails when no HIP device is present.
//
// Host arrays cross into device memory through two bump arenas that are reset per call: a pinned host arena (staging both
// ways, so every copy is asynchronous on the stream) and a device arena.  One stream synchronisation per method.
#include "common.hpp"
#include "multi_kernel.hpp"
#include <algorithm>
#include "camera_device.hpp"
#include "pose_internal.hpp"
#include "track_slots.hpp"
#include "stages_hip.hpp"
#include <cmath>
#include <cstdlib>

namespace alva_slam {

namespace {

// Frame::computeKeypoint's second half (frame.cpp:109-112)
__global__ void __launch_bounds__(256) k_bearing(const float *unpx, int n, const double *invK, double *bv) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    alva_bearing_dev(invK, unpx[2 * i], unpx[2 * i + 1], bv + 3 * (size_t) i);
}

// ---- the fused tracking step (VisualFrontend::kltTrackingFromMotionPrior + the set-up of computePose) -------------------------
// Three single-workgroup glue kernels around the two tracker launches.  Lists are built with STABLE block-wide compaction
// (ballot + popcount prefix), so every list keeps the slot order = the frame container's iteration order the reference works in.
struct TrackDev {
    int n, use_prior, width, height;
    const float *in_px;        // pinned host, [n][2]
    const uint8_t *in_is3d;    // pinned host, [n]
    const double *in_wpt;      // pinned host, [n][3]
    double q[4], t[3];         // T_cw (predicted)
    AlvaCam cam;
    const double *invK;
    int *cnt;                  // device: nA, nB0, nB, good1, p3p_req, n_pose
    int *slotA, *slotB;
    float *ptsA, *priorA, *outA, *ptsB, *priorB, *outB;
    uint8_t *stA, *stB;
    uint8_t *d_is3d, *d_code;
    double *d_wpt;
    float *d_px;
    uint8_t *o_code;           // pinned host outputs
    float *o_px, *o_unpx;
    double *o_bv;
    int *o_hdr;
    double *Pbv, *Puv, *Pwpt;  // device: correspondences of the pose solve
};

#define TRK_NT 1024
// exclusive prefix of `flag` over the block in thread order; *total = number of set flags.  s_w: 17 ints of LDS.
__device__ __forceinline__ int block_prefix(bool flag, int *s_w, int *total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned long long b = __ballot(flag);
    const int within = __popcll(b & ((1ull << lane) - 1ull));
    __syncthreads();  // s_w reuse
    if (lane == 0) s_w[wave] = __popcll(b);
    __syncthreads();
    if (threadIdx.x == 0) {
        int acc = 0;
        for (int w = 0; w < TRK_NT / 64; w++) {
            const int c = s_w[w];
            s_w[w] = acc;
            acc += c;
        }
        s_w[16] = acc;
    }
    __syncthreads();
    *total = s_w[16];
    return s_w[wave] + within;
}

// priors of the 3-D slots from the predicted pose (visual_frontend.cpp:125-152) and the two lists: A = 3-D slots whose projection
// falls into the image (tracked from it on one level), B = everything else (tracked from its own position on the full pyramid)
__global__ void __launch_bounds__(TRK_NT) k_track_prepare(TrackDev D) {
    __shared__ int s_w[17];
    int baseA = 0, baseB = 0;
    for (int c0 = 0; c0 < D.n; c0 += TRK_NT) {
        const int i = c0 + threadIdx.x;
        bool inA = false, valid = i < D.n;
        float px = 0.f, py = 0.f, qu = 0.f, qv = 0.f;
        if (valid) {
            px = D.in_px[2 * i];
            py = D.in_px[2 * i + 1];
            const uint8_t is3 = D.in_is3d[i];
            D.d_is3d[i] = is3;
            double X[3] = {0, 0, 0};
            if (is3) {
                X[0] = D.in_wpt[3 * (size_t) i]; X[1] = D.in_wpt[3 * (size_t) i + 1]; X[2] = D.in_wpt[3 * (size_t) i + 2];
            }
            D.d_wpt[3 * (size_t) i] = X[0]; D.d_wpt[3 * (size_t) i + 1] = X[1]; D.d_wpt[3 * (size_t) i + 2] = X[2];
            if (D.use_prior && is3) {
                double pc[3];
                alva_se3_apply_dev(D.q, D.t, X, pc);
                alva_project_dist_dev(D.cam, pc[0], pc[1], pc[2], qu, qv);
                inA = qu >= 0 && qv >= 0 && (double) qu < (double) D.width && (double) qv < (double) D.height;  // Frame::isInImage
            }
        }
        int totA, totB;
        const int pa = block_prefix(inA, s_w, &totA);
        const int pb = block_prefix(valid && !inA, s_w, &totB);
        if (inA) {
            const int k = baseA + pa;
            D.slotA[k] = i;
            D.ptsA[2 * k] = px; D.ptsA[2 * k + 1] = py;
            D.priorA[2 * k] = qu; D.priorA[2 * k + 1] = qv;
        } else if (valid) {
            const int k = baseB + pb;
            D.slotB[k] = i;
            D.ptsB[2 * k] = px; D.ptsB[2 * k + 1] = py;
            D.priorB[2 * k] = px; D.priorB[2 * k + 1] = py;
        }
        baseA += totA;
        baseB += totB;
    }
    if (threadIdx.x == 0) {
        D.cnt[0] = baseA;
        D.cnt[1] = baseB;
        D.cnt[2] = baseB;
        D.cnt[3] = 0;
        D.cnt[4] = 0;
        D.cnt[5] = 0;
    }
}

// after the one-level pass (:173-203): failures join list B behind its original entries, keeping their order; fewer than 33 % successes
// => p3pReq_ and every prior of list B falls back to the keypoint's own position
__global__ void __launch_bounds__(TRK_NT) k_track_pass2(TrackDev D) {
    __shared__ int s_w[17];
    const int nA = D.cnt[0], nB0 = D.cnt[1];
    int failed = 0;
    for (int c0 = 0; c0 < nA; c0 += TRK_NT) {
        const int j = c0 + threadIdx.x;
        const bool bad = j < nA && !D.stA[j];
        int tot;
        const int p = block_prefix(bad, s_w, &tot);
        if (bad) {
            const int k = nB0 + failed + p;
            D.slotB[k] = D.slotA[j];
            D.ptsB[2 * k] = D.ptsA[2 * j]; D.ptsB[2 * k + 1] = D.ptsA[2 * j + 1];
            D.priorB[2 * k] = D.outA[2 * j]; D.priorB[2 * k + 1] = D.outA[2 * j + 1];  // the forward tracker's result (in/out prior)
        }
        failed += tot;
    }
    const int good = nA - failed, nB = nB0 + failed;
    const bool req = nA > 0 && (double) good < 0.33 * (double) nA;
    __syncthreads();
    if (req)
        for (int k = threadIdx.x; k < nB; k += TRK_NT) {
            D.priorB[2 * k] = D.ptsB[2 * k];
            D.priorB[2 * k + 1] = D.ptsB[2 * k + 1];
        }
    if (threadIdx.x == 0) {
        D.cnt[2] = nB;
        D.cnt[3] = good;
        D.cnt[4] = req ? 1 : 0;
    }
}

// after the full-pyramid pass: per-slot verdicts and positions, Frame::computeKeypoint for every tracked slot, and the
// correspondences of the pose solve (3-D survivors in slot order: visual_frontend.cpp:275-298)
__global__ void __launch_bounds__(TRK_NT) k_track_finish(TrackDev D) {
    __shared__ int s_w[17];
    const int nA = D.cnt[0], nB0 = D.cnt[1], nB = D.cnt[2];
    for (int i = threadIdx.x; i < D.n; i += TRK_NT) D.d_code[i] = 0;
    __syncthreads();
    for (int j = threadIdx.x; j < nA; j += TRK_NT)
        if (D.stA[j]) {
            const int s = D.slotA[j];
            D.d_code[s] = 1;
            D.d_px[2 * s] = D.outA[2 * j]; D.d_px[2 * s + 1] = D.outA[2 * j + 1];
        }
    for (int k = threadIdx.x; k < nB; k += TRK_NT)
        if (D.stB[k]) {
            const int s = D.slotB[k];
            D.d_code[s] = k < nB0 ? 2 : 3;
            D.d_px[2 * s] = D.outB[2 * k]; D.d_px[2 * s + 1] = D.outB[2 * k + 1];
        }
    __syncthreads();
    int base = 0;
    for (int c0 = 0; c0 < D.n; c0 += TRK_NT) {
        const int i = c0 + threadIdx.x;
        uint8_t code = 0;
        float px = 0.f, py = 0.f, ux = 0.f, uy = 0.f;
        double bv[3] = {0, 0, 0};
        bool pose = false;
        if (i < D.n) {
            code = D.d_code[i];
            if (code) {
                px = D.d_px[2 * i]; py = D.d_px[2 * i + 1];
                alva_undistort_dev(D.cam, px, py, ux, uy);
                alva_bearing_dev(D.invK, ux, uy, bv);
                pose = D.d_is3d[i] != 0;
            }
            D.o_code[i] = code;
            D.o_px[2 * i] = px; D.o_px[2 * i + 1] = py;
            D.o_unpx[2 * i] = ux; D.o_unpx[2 * i + 1] = uy;
            D.o_bv[3 * (size_t) i] = bv[0]; D.o_bv[3 * (size_t) i + 1] = bv[1]; D.o_bv[3 * (size_t) i + 2] = bv[2];
        }
        int tot;
        const int p = block_prefix(pose, s_w, &tot);
        if (pose) {
            const size_t k = (size_t) (base + p);
            D.Pbv[3 * k] = bv[0]; D.Pbv[3 * k + 1] = bv[1]; D.Pbv[3 * k + 2] = bv[2];
            D.Puv[2 * k] = (double) ux; D.Puv[2 * k + 1] = (double) uy;
            D.Pwpt[3 * k] = D.d_wpt[3 * (size_t) i]; D.Pwpt[3 * k + 1] = D.d_wpt[3 * (size_t) i + 1]; D.Pwpt[3 * k + 2] = D.d_wpt[3 * (size_t) i + 2];
        }
        base += tot;
    }
    if (threadIdx.x == 0) {
        D.cnt[5] = base;
        D.o_hdr[0] = nA; D.o_hdr[1] = nB0; D.o_hdr[2] = nB; D.o_hdr[3] = D.cnt[3]; D.o_hdr[4] = D.cnt[4]; D.o_hdr[5] = base;
    }
}

#include "track_compact_device.hpp"
__global__ void __launch_bounds__(CMP_NT) k_track_compact(TrackSlots D) { (void) track_compact_body<CMP_NT>(D, (int) blockIdx.x, (int) gridDim.x); }
ALVA_MULTI_KERNEL(MK_TRACK_COMPACT, k_track_compact_multi, TrackSlots, dim3(CMP_NT), CMP_NT, (void) track_compact_body<CMP_NT>(A, bx, (int) gx));
static inline int compact_grid(int n) {   // ~256 slots per workgroup
    const int g = (n + 255) / 256;
    return g < 1 ? 1 : g > CMP_MAX_WG ? CMP_MAX_WG : g;
}

struct Arena {
    uint8_t *base = nullptr;
    size_t cap = 0, used = 0;
    bool pinned = false;
    int grow(size_t need, hipStream_t st) {
        if (need <= cap) return ALVA_OK;
        if (base) {
            ALVA_HIP(alva_stream_sync(st));
            if (pinned) ALVA_HIP(hipHostFree(base));
            else ALVA_HIP(hipFree(base));
            base = nullptr;
        }
        size_t c = cap ? cap : (size_t) 1 << 20;
        cap = 0;   // nothing is held while the allocation below can still fail (a failed grow must not leave cap > 0 with base == nullptr)
        while (c < need) c *= 2;
        if (pinned) ALVA_HIP(hipHostMalloc((void **) &base, c, hipHostMallocDefault));
        else ALVA_HIP(hipMalloc((void **) &base, c));
        cap = c;
        return ALVA_OK;
    }
    void release() {
        if (base) {
            if (pinned) (void) hipHostFree(base);
            else (void) hipFree(base);
        }
        base = nullptr;
        cap = used = 0;
    }
};

}  // namespace

struct HipStages::Impl {
    int device = 0;
    alva_medoid_store *med = nullptr;   // the map points' descriptor tables (medoid.hip), created with the first replay
    alva_ctx *ctx = nullptr;
    hipStream_t st = nullptr;
    Camera cam;
    bool clahe = false;
    // Three pyramids / two gray images in rotation: the current frame's, the previous frame's (the tracker reads both) and the one a
    // look-ahead build may be filling for the next frame (hint_next_frame_device).  d_gray always names the CURRENT frame's image.
    alva_pyramid *pyr[3] = {nullptr, nullptr, nullptr};
    int cur = 0, prev = 1, nxt = 2;
    uint8_t *d_rgba = nullptr, *d_gray = nullptr, *d_gray_next = nullptr, *d_eq = nullptr, *h_rgba = nullptr;
    // look-ahead: the hinted source, whether its build has been enqueued (same stream, behind the pose kernels)
    const uint8_t *ahead_src = nullptr;
    bool ahead_enqueued = false;
    double *d_invK = nullptr;
    const uint8_t *registered = nullptr, *registered_dev = nullptr;  // caller buffer locked + mapped by register_frame_buffer
    size_t registered_bytes = 0;
    uint8_t *bar_frame = nullptr;   // alloc_frame_buffer: the caller's frame buffer in device memory, written by the host over the BAR
    size_t bar_frame_bytes = 0;
    bool upload_in_flight = false;
    hipEvent_t upload_done = nullptr;
    double max_quality = 0.001;  // state.hpp:57; lives as long as the reference's FeatureExtractor object (system.cpp:31)
    Arena dev, pin;
    // the map layer's record arena (slam/mp_rec.hpp): chunks of pinned host memory + the device-resident table of their addresses
    std::vector<MpRec *> rec_chunks;
    const MpRec **d_rec_tab = nullptr;
    static constexpr int REC_TAB_CAP = 4096;   // 16.7 M map points
    // the fused tracking step: persistent device / pinned blocks (grown when the keypoint count outgrows them)
    Arena trk_dev, trk_pin;
    // Device memory that the HOST is going to write over the BAR must not have a previous life left in an L2: the allocator recycles
    // pages, and a line that an earlier owner's kernel wrote may still sit DIRTY in an L2 (device-local memory is only written back when
    // the line is evicted) -- evicted later, it would overwrite what the host stored in the meantime (seen as a tracker reading last
    // frame's slot table, once in a few hundred session starts).  One device-wide synchronisation behind a fill: its system-scope
    // release writes every dirty line back; nothing on the device writes the block afterwards.
    // FINE-GRAINED device memory (coherent with the host by definition; the device does not keep its lines in an L2 across kernels).
    // hipDeviceMallocUncached looked equivalent and is not: with BOTH the slot table and the caller's frame buffer allocated that way,
    // test_group_sessions_equal_their_solo_runs[one_lane] failed in 8 of 13 runs of tests/test_gpu_system.py (a session's second tracking
    // frame tracked from its first frame's table); either one alone, or the table fine-grained, never did (3 / 3, 3 / 3, and every run
    // since).  ALVA_BAR_FLAG=uncached restores the failing combination for whoever wants to find out why.  What the HIP memory model
    // promises: fine-grained allocations are coherent between host and device at system scope WHILE kernels run; "uncached" only selects
    // a cache policy for the device's own accesses and promises nothing about host stores that arrive over the BAR -- so the shipped flag
    // is the documented one, and the failing one was never covered by a rule.  tests/test_gpu_bar_buffers.py starts 32 sessions (and 30
    // one-lane groups) back to back under the shipped flag and compares every one with the first, bit for bit.
    static unsigned bar_alloc_flag() {
        static const unsigned f = getenv("ALVA_BAR_FLAG") && strcmp(getenv("ALVA_BAR_FLAG"), "uncached") == 0 ? hipDeviceMallocUncached : hipDeviceMallocFinegrained;
        return f;
    }
    // Host stores into device memory need the whole of it mapped into the CPU's address space (a "large" / resizable BAR).  On a small-BAR
    // host -- many passthrough VMs -- hipExtMallocWithFlags still succeeds, the pointer is simply not CPU-mapped and the first host store
    // faults: ask the driver instead of finding out (both paths then fall back: pinned slot table + copy kernel, registered frame buffer).
    static bool host_can_store_to_device_memory(int device) {
        int large = 0;
        if (hipDeviceGetAttribute(&large, hipDeviceAttributeIsLargeBar, device) != hipSuccess) {
            (void) hipGetLastError();
            return false;
        }
        return large != 0;
    }
    static hipError_t scrub_host_written(void *p, size_t bytes) {
        hipError_t e = hipMemset(p, 0, bytes);
        if (e != hipSuccess) return e;
        return hipDeviceSynchronize();
    }
    uint8_t *trk_in = nullptr;   // see track_reserve
    bool bar_table = true;       // the slot table in host-written device memory; ALVA_NO_BAR_TABLE=1: pinned table + k_track_stage_in
    struct TrackIn {
        float *px;
        uint8_t *is3d;
        double *wpt;
    };
    // TWO tables (round 6): a frame's table is either written by the host or built by the tracker from the previous frame's (the carried
    // table, track_slots.hpp), so consecutive frames alternate between them; behind the tables the carry index the host writes instead.
    static size_t track_in_bytes(size_t c) { return c * 8 + c + 256 + c * 24; }
    TrackIn track_in(int par) const {
        const size_t c = (size_t) trk_cap;
        TrackIn T;
        uint8_t *b = trk_in + (size_t) (par & 1) * track_in_bytes(c);
        T.px = (float *) b; b += c * 8;
        T.is3d = b; b += c + 256 - (c & 255);
        T.wpt = (double *) b;
        return T;
    }
    uint16_t *track_carry() const { return (uint16_t *) (trk_in + 2 * track_in_bytes((size_t) trk_cap)); }
    int trk_par = 0;         // the table (and d_px buffer) of the LAST tracker launch; the next one takes the other
    int trk_valid_n = -1;    // slots of that launch if its table + tracked positions are complete on the device (a carried table may follow), else -1
    bool carry_ok = true;    // ALVA_NO_CARRY=1: every frame's table assembled by the host (A/B)
    int trk_cap = 0;
    bool fused = true;       // ALVA_TRACK_UNFUSED=1: compose the tracking step from the fine-grained stages instead (A/B testing)
    bool poll = true;        // wait for the tracking step by polling its completion word in pinned memory (ALVA_NO_POLL=1: stream synchronisation)
    int trk_seq = 0;
    bool lists = false;      // ALVA_TRACK_LISTS=1: the fused step with explicit keypoint lists (five launches) instead of slot-wise (three)
    // pinned staging of the tracking step: slot table in (the map layer writes it there directly, track_slot_buffers), results out
    struct TrackPin {
        float *in_px;
        uint8_t *in_is3d;
        double *in_wpt;
        int *o_hdr;
        uint8_t *o_code;
        float *o_px, *o_unpx;
        double *o_bv;
    };
    TrackPin track_pin() const {
        const size_t c = (size_t) trk_cap;
        TrackPin P;
        uint8_t *h = trk_pin.base;
        P.in_px = (float *) h; h += c * 8;
        P.in_is3d = h; h += c + 64 - (c & 63);
        P.in_wpt = (double *) h; h += c * 24;
        P.o_hdr = (int *) h; h += 256;
        P.o_code = h; h += c + 64 - (c & 63);
        P.o_px = (float *) h; h += c * 8;
        P.o_unpx = (float *) h; h += c * 8;
        P.o_bv = (double *) h; h += c * 24;
        return P;
    }
    int track_reserve(int n) {
        if (n <= trk_cap) return ALVA_OK;
        const int cap = ((n + 1023) / 1024 + 1) * 1024;
        const size_t c = (size_t) cap;
        // device: cnt | slotA slotB | ptsA priorA outA ptsB priorB outB | stA stB is3d code | wpt | px | Pbv Puv Pwpt  (the list form; the
        // slot-wise form needs less)
        const size_t dev_bytes = 1024 + c * 8 + c * 48 + c * 4 + 256 + c * 24 + c * 8 + c * 64 + c * 8 + 256;   // (+ the second d_px of the slot-wise form)
        const size_t pin_bytes = c * 8 + c + 64 + c * 24 + 256 + c + 64 + c * 16 + c * 24 + 256;
        int rc = trk_dev.grow(dev_bytes, st);
        if (rc) return rc;
        rc = trk_pin.grow(pin_bytes, st);
        if (rc) return rc;
        // the slot table (positions | 3-D flags | world points) in DEVICE memory that the host writes directly: the whole of the device's
        // memory is visible to the CPU (large BAR; tools/probes/bar_probe.cpp: 128 KB of ordinary stores in 2.6 us, write-combined and
        // posted), so the map layer assembles the table where the tracker reads it and the copy kernel of rounds 2 - 4 (k_track_stage_in:
        // 7 us + a launch gap in front of every frame's tracker) is gone.  Fine-grained memory (bar_alloc_flag): an L2 must not answer with
        // last frame's line.  The host never READS this memory (a load over the bus costs ~1 us).
        trk_valid_n = -1;   // (the buffers move)
        if (trk_in) {
            ALVA_HIP(alva_stream_sync(st));
            ALVA_HIP(hipFree(trk_in));
            trk_in = nullptr;
        }
        if (bar_table) {
            const size_t in_bytes = 2 * track_in_bytes(c) + c * 2 + 256;
            if (hipExtMallocWithFlags((void **) &trk_in, in_bytes, bar_alloc_flag()) != hipSuccess) {
                (void) hipGetLastError();
                trk_in = nullptr;
                bar_table = false;   // no such memory here: the pinned table + the copy kernel
            } else {
                ALVA_HIP(scrub_host_written(trk_in, in_bytes));
            }
        }
        trk_cap = cap;
        ALVA_HIP(hipMemsetAsync(trk_dev.base, 0, 1024, st));  // the slot-wise step's counters start at zero
        // fresh (or recycled) pinned memory: the completion word must not equal a sequence number the host is about to wait for.  The
        // stream is idle here (both grows synchronised it), so a plain host store cannot race a kernel's publication.
        ALVA_HIP(alva_stream_sync(st));
        track_pin().o_hdr[8] = 0;
        track_pin().o_hdr[9] = 0;
        track_pin().o_hdr[10] = 0;
        track_pin().o_hdr[11] = 0;
        track_pin().o_hdr[12] = 0;
        track_pin().o_hdr[13] = 0;
        return ALVA_OK;
    }
    bool pose_pending = false;
    int pose_n = 0;
    alva_detect_pending det_pending{};   // detect_begin -> detect_end
    int dsc_n = 0;                       // describe_begin -> describe_end: count, where the results will be (pinned staging)
    const uint8_t *dsc_desc = nullptr, *dsc_valid = nullptr;
    const uint8_t *det_h = nullptr;
    int det_cap = 0;
    // a call plans its buffers first (sizes), then the arenas are grown once and carved
    struct Plan {
        std::vector<size_t> sizes;
        size_t add(size_t bytes) {
            sizes.push_back((bytes + 255) / 256 * 256);
            return sizes.size() - 1;
        }
    };
    int carve(Plan &p, std::vector<uint8_t *> &d, std::vector<uint8_t *> &h) {
        size_t total = 0;
        for (size_t s: p.sizes) total += s;
        int rc = dev.grow(total, st);
        if (rc) return rc;
        rc = pin.grow(total, st);
        if (rc) return rc;
        d.resize(p.sizes.size());
        h.resize(p.sizes.size());
        size_t off = 0;
        for (size_t i = 0; i < p.sizes.size(); i++) {
            d[i] = dev.base + off;
            h[i] = pin.base + off;
            off += p.sizes[i];
        }
        return ALVA_OK;
    }
    // planned buffers i0 .. i1 (consecutive in both arenas, same offsets) with ONE copy command instead of one per buffer: a copy
    // command costs ~5 us of host time and ~5-10 us of queue latency, and the small keyframe stages issued 4 - 10 of them each
    int up_span(const Plan &p, const std::vector<uint8_t *> &d, const std::vector<uint8_t *> &h, size_t i0, size_t i1) {
        ALVA_HIP(hipMemcpyAsync(d[i0], h[i0], (size_t) (h[i1] - h[i0]) + p.sizes[i1], hipMemcpyHostToDevice, st));
        return ALVA_OK;
    }
    int down_span(const Plan &p, const std::vector<uint8_t *> &d, const std::vector<uint8_t *> &h, size_t i0, size_t i1) {
        ALVA_HIP(hipMemcpyAsync(h[i0], d[i0], (size_t) (h[i1] - h[i0]) + p.sizes[i1], hipMemcpyDeviceToHost, st));
        return ALVA_OK;
    }
};

#define UP(i, src, bytes)                                                                         \
    do {                                                                                          \
        if ((bytes) > 0) {                                                                        \
            memcpy(h[i], (src), (bytes));                                                         \
            ALVA_HIP(hipMemcpyAsync(d[i], h[i], (bytes), hipMemcpyHostToDevice, m->st));          \
        }                                                                                         \
    } while (0)
#define DOWN(i, bytes)                                                                            \
    do {                                                                                          \
        if ((bytes) > 0) ALVA_HIP(hipMemcpyAsync(h[i], d[i], (bytes), hipMemcpyDeviceToHost, m->st)); \
    } while (0)

HipStages::HipStages() : m(new Impl()) {}

HipStages::~HipStages() {
    if (!m) return;
    (void) hipSetDevice(m->device);
    if (m->ctx) (void) alva_ctx_sync(m->ctx);
    for (auto &p: m->pyr) alva_pyramid_destroy(p);
    void *bufs[] = {m->d_rgba, m->d_gray, m->d_gray_next, m->d_eq, m->d_invK};
    for (void *b: bufs)
        if (b) (void) hipFree(b);
    if (m->h_rgba) (void) hipHostFree(m->h_rgba);
    if (m->registered) (void) hipHostUnregister((void *) m->registered);
    if (m->upload_done) (void) hipEventDestroy(m->upload_done);
    m->dev.release();
    m->pin.release();
    m->trk_dev.release();
    m->trk_pin.release();
    if (m->trk_in) (void) hipFree(m->trk_in);
    if (m->bar_frame) (void) hipFree(m->bar_frame);
    alva_medoid_store_destroy(m->med);
    for (MpRec *c: m->rec_chunks) (void) hipHostFree(c);
    if (m->d_rec_tab) (void) hipFree(m->d_rec_tab);
    alva_ctx_destroy(m->ctx);
    delete m;
}

int HipStages::init(int device, const Camera &cam, bool clahe, const double *invK, void *hip_stream) {
    m->device = device;
    m->cam = cam;
    m->clahe = clahe;
    m->pin.pinned = true;
    m->trk_pin.pinned = true;
    m->fused = getenv("ALVA_TRACK_UNFUSED") == nullptr;
    m->lists = getenv("ALVA_TRACK_LISTS") != nullptr;
    m->poll = getenv("ALVA_NO_POLL") == nullptr;
    // the slot table in host-written device memory (track_reserve): ALVA_NO_BAR_TABLE=1 keeps the pinned table + k_track_stage_in (A/B)
    m->bar_table = getenv("ALVA_NO_BAR_TABLE") == nullptr && Impl::host_can_store_to_device_memory(m->device);
    m->carry_ok = getenv("ALVA_NO_CARRY") == nullptr;
    int rc = hip_stream ? alva_ctx_create(device, hip_stream, 0, &m->ctx) : alva_ctx_create(device, nullptr, 1, &m->ctx);
    if (rc) return rc;
    m->st = (hipStream_t) alva_ctx_stream(m->ctx);
    const size_t P = (size_t) cam.width * cam.height;
    ALVA_HIP(hipMalloc((void **) &m->d_rgba, P * 4));
    ALVA_HIP(hipMalloc((void **) &m->d_gray, P));
    ALVA_HIP(hipMalloc((void **) &m->d_gray_next, P));
    if (clahe) ALVA_HIP(hipMalloc((void **) &m->d_eq, P));
    ALVA_HIP(hipMalloc((void **) &m->d_invK, 9 * sizeof(double)));
    ALVA_HIP(hipMemcpy(m->d_invK, invK, 9 * sizeof(double), hipMemcpyHostToDevice));
    ALVA_HIP(hipHostMalloc((void **) &m->h_rgba, P * 4, hipHostMallocDefault));
    ALVA_HIP(hipEventCreateWithFlags(&m->upload_done, hipEventDisableTiming));
    for (auto &p: m->pyr) {
        rc = alva_pyramid_create(m->ctx, cam.width, cam.height, 9, 3, &p);  // state.hpp:51-53: 9 x 9 window, 3 levels
        if (rc) return rc;
    }
    return ALVA_OK;
}

int HipStages::warm_up(int cell) {
    const Camera &k = m->cam;
    const int W = k.width, H = k.height;
    uint32_t lcg = 12345u;
    auto rnd = [&]() {
        lcg = lcg * 1664525u + 1013904223u;
        return (float) (lcg >> 8) * (1.f / 16777216.f);
    };
    int rc;
    // two frames: gray + pyramid (+ CLAHE)
    {
        std::vector<uint8_t> img((size_t) W * H * 4);
        for (size_t i = 0; i < img.size(); i += 4) {  // a smooth pattern with some texture: trackable, detectable
            const size_t p = i / 4, x = p % (size_t) W, y = p / (size_t) W;
            const uint8_t v = (uint8_t) (128 + 60 * std::sin(0.13 * (double) x) * std::cos(0.11 * (double) y) + 30 * ((x / 7 + y / 5) & 1));
            img[i] = img[i + 1] = img[i + 2] = v;
            img[i + 3] = 255;
        }
        for (int f = 0; f < 2; f++) {
            rc = new_frame(img.data());
            if (rc) return rc;
            rc = frame_done();
            if (rc) return rc;
        }
        ALVA_HIP(alva_stream_sync(m->st));
    }
    const int cw = (W + cell - 1) / cell, chh = (H + cell - 1) / cell, n = cw * chh;
    // the tracking step (stage-in, tracker, retry, compaction) and the pose solve behind it
    {
        float *px;
        uint8_t *is3d;
        double *wpt;
        std::vector<float> vpx;
        std::vector<uint8_t> v3;
        std::vector<double> vw;
        if (!track_slot_buffers(n + n / 4, &px, &is3d, &wpt)) {
            vpx.resize((size_t) (n + n / 4) * 2); v3.resize((size_t) (n + n / 4)); vw.resize((size_t) (n + n / 4) * 3);
            px = vpx.data(); is3d = v3.data(); wpt = vw.data();
        }
        for (int i = 0; i < n; i++) {
            const float x = 12.f + rnd() * (float) (W - 24), y = 12.f + rnd() * (float) (H - 24);
            px[2 * i] = x; px[2 * i + 1] = y;
            is3d[i] = (i % 8) != 0;
            const double z = 3.0 + 2.0 * rnd();
            wpt[3 * i] = ((double) x - k.cx) / k.fx * z; wpt[3 * i + 1] = ((double) y - k.cy) / k.fy * z; wpt[3 * i + 2] = z;
        }
        TrackJob job;
        job.n = n;
        job.px = px; job.is3d = is3d; job.wpt = wpt;
        job.want_pose = 1;
        job.do_random = 0;
        TrackKlt kl;
        TrackPose po;
        rc = track_begin(job, kl);
        if (!rc) rc = track_pose_collect(po);
        if (rc) return rc;
    }
    // keyframe stages
    {
        std::vector<float> pts((size_t) n * 2), un((size_t) n * 2), np((size_t) (n + 8) * 2);
        std::vector<double> bv((size_t) n * 3);
        std::vector<uint8_t> desc((size_t) n * 32), valid((size_t) n);
        for (int i = 0; i < n; i++) {
            pts[2 * (size_t) i] = 40.f + rnd() * (float) (W - 80);
            pts[2 * (size_t) i + 1] = 40.f + rnd() * (float) (H - 80);
        }
        rc = describe(n, pts.data(), desc.data(), valid.data());
        if (!rc) rc = compute_keypoints(n, pts.data(), un.data(), bv.data());
        if (rc) return rc;
        const double q = m->max_quality;
        int count = 0;
        rc = detect(cell, n / 2, pts.data(), n + 8, np.data(), &count);
        m->max_quality = q;
        if (rc) return rc;
    }
    {   // triangulation
        const int nt = n / 4 + 8;
        std::vector<double> T(36, 0.), bl((size_t) nt * 3), br((size_t) nt * 3), w((size_t) nt * 3), inv((size_t) nt), par((size_t) nt);
        std::vector<float> ul((size_t) nt * 2), ur((size_t) nt * 2);
        std::vector<int> grp((size_t) nt, 0);
        std::vector<uint8_t> st((size_t) nt);
        for (int b = 0; b < 3; b++) {   // R = I, t = (0.3, 0, 0) / its inverse / the keyframe's pose
            T[12 * b] = T[12 * b + 4] = T[12 * b + 8] = 1.;
        }
        T[9] = 0.3; T[21] = -0.3;
        for (int i = 0; i < nt; i++) {
            const double x = -0.3 + 0.6 * rnd(), y = -0.2 + 0.4 * rnd(), z = 4.;
            const double l[3] = {x, y, z}, r[3] = {x - 0.3, y, z};
            const double nl = std::sqrt(l[0] * l[0] + l[1] * l[1] + l[2] * l[2]), nr = std::sqrt(r[0] * r[0] + r[1] * r[1] + r[2] * r[2]);
            for (int c = 0; c < 3; c++) {
                bl[3 * (size_t) i + c] = l[c] / nl;
                br[3 * (size_t) i + c] = r[c] / nr;
            }
            ul[2 * (size_t) i] = (float) (k.fx * l[0] / z + k.cx); ul[2 * (size_t) i + 1] = (float) (k.fy * l[1] / z + k.cy);
            ur[2 * (size_t) i] = (float) (k.fx * r[0] / z + k.cx); ur[2 * (size_t) i + 1] = (float) (k.fy * r[1] / z + k.cy);
        }
        rc = triangulate(nt, 1, T.data(), grp.data(), bl.data(), br.data(), ul.data(), ur.data(), w.data(), inv.data(), st.data(), par.data());
        if (rc) return rc;
    }
    const int n_kf = 14;
    {   // matchToMap: the frame's keypoints (one per cell) + a local map of 3 n points with 6 observations each
        const int n_local = 3 * n, n_mp = n + n_local, per = 6;
        std::vector<int> cell_ptr((size_t) n + 1), cell_mp((size_t) n), obs_ptr((size_t) n_mp + 1), obs_kf, local((size_t) n_local), match((size_t) n_mp);
        std::vector<double> kq((size_t) n_kf * 4, 0.), kt((size_t) n_kf * 3, 0.), X((size_t) n_mp * 3);
        std::vector<uint8_t> m3((size_t) n_mp, 1), mhd((size_t) n_mp, 1), od, ohd;
        std::vector<float> opx;
        for (int f = 0; f < n_kf; f++) {
            kq[4 * (size_t) f + 3] = 1.;
            kt[3 * (size_t) f] = -0.05 * f;
        }
        for (int i = 0; i <= n; i++) cell_ptr[(size_t) i] = i;
        for (int i = 0; i < n; i++) cell_mp[(size_t) i] = i;
        for (int i = 0; i < n_local; i++) local[(size_t) i] = n + i;
        for (int p = 0; p < n_mp; p++) {
            const int c = p % n;
            const double x = ((c % cw) + 0.5 + 0.3 * (p / n)) * cell, y = ((c / cw) + 0.5) * cell, z = 4.;
            X[3 * (size_t) p] = (x - k.cx) / k.fx * z; X[3 * (size_t) p + 1] = (y - k.cy) / k.fy * z; X[3 * (size_t) p + 2] = z;
            obs_ptr[(size_t) p] = (int) obs_kf.size();
            for (int o = 0; o < per; o++) {
                obs_kf.push_back((p + o) % n_kf);
                opx.push_back((float) x);
                opx.push_back((float) y);
                for (int b = 0; b < 32; b++) od.push_back((uint8_t) (rnd() * 255.f));
                ohd.push_back(1);
            }
        }
        obs_ptr[(size_t) n_mp] = (int) obs_kf.size();
        rc = match_to_map(cell, cw, n, cell_ptr.data(), cell_mp.data(), n_kf, kq.data(), kt.data(), n_mp, X.data(), m3.data(), mhd.data(), obs_ptr.data(),
                          obs_kf.data(), opx.data(), od.data(), ohd.data(), n_kf - 1, n, n_local, local.data(), 2.0f, 0.2f, match.data());
        if (rc) return rc;
    }
    {   // local BA: 14 keyframes, 3 n points anchored round-robin, 6 further observations each
        const int n_pt = 3 * n, per = 6, n_obs = n_pt * per;
        std::vector<double> poses((size_t) n_kf * 7, 0.), auv((size_t) n_pt * 2), inv((size_t) n_pt, 0.25), ouv((size_t) n_obs * 2), chi2((size_t) n_obs);
        std::vector<uint8_t> kc((size_t) n_kf, 0), dp((size_t) n_obs);
        std::vector<int> pa((size_t) n_pt), okf((size_t) n_obs), opt((size_t) n_obs);
        kc[0] = kc[1] = 1;
        for (int f = 0; f < n_kf; f++) {
            poses[7 * (size_t) f] = 0.05 * f;   // Twc: camera f sits at x = 0.05 f
            poses[7 * (size_t) f + 6] = 1.;
        }
        for (int p = 0; p < n_pt; p++) {
            const int a = p % n_kf;
            const double u = 40. + rnd() * (W - 80), v = 40. + rnd() * (H - 80), z = 4.;
            pa[(size_t) p] = a;
            auv[2 * (size_t) p] = u; auv[2 * (size_t) p + 1] = v;
            const double xw = (u - k.cx) / k.fx * z + 0.05 * a, yw = (v - k.cy) / k.fy * z;
            for (int o = 0; o < per; o++) {
                const int f = (a + 1 + o) % n_kf;
                const size_t e = (size_t) p * per + (size_t) o;
                okf[e] = f;
                opt[e] = p;
                ouv[2 * e] = k.fx * (xw - 0.05 * f) / z + k.cx + 0.4 * (rnd() - 0.5);
                ouv[2 * e + 1] = k.fy * yw / z + k.cy + 0.4 * (rnd() - 0.5);
            }
        }
        rc = local_ba(n_kf, poses.data(), kc.data(), n_pt, pa.data(), auv.data(), inv.data(), n_obs, okf.data(), opt.data(), ouv.data(), 5, chi2.data(),
                      dp.data());
        if (rc) return rc;
    }
    pending_.active = false;
    fused_active_ = false;
    m->pose_pending = false;
    return ALVA_OK;
}

int HipStages::build_from(const uint8_t *d_src) {
    Impl &M = *m;
    const bool hit = M.ahead_enqueued && M.ahead_src == d_src;
    M.ahead_enqueued = false;
    M.ahead_src = nullptr;
    // rotate: previous <- current <- next; the old previous becomes the free slot
    const int freed = M.prev;
    M.prev = M.cur;
    M.cur = M.nxt;
    M.nxt = freed;
    std::swap(M.d_gray, M.d_gray_next);
    if (hit) return ALVA_OK;
    if (!m->clahe) return alva_pyramid_build_from_rgba(m->ctx, m->pyr[m->cur], d_src, (size_t) m->cam.width * 4, m->d_gray, (size_t) m->cam.width);
    int rc = alva_rgba2gray(m->ctx, d_src, (size_t) m->cam.width * 4, m->cam.width, m->cam.height, m->d_gray, (size_t) m->cam.width);
    if (rc) return rc;
    // visual_frontend.cpp:16-18: clip limit 3, grid = image size / 50 (state.hpp:45-46)
    rc = alva_clahe(m->ctx, m->d_gray, (size_t) m->cam.width, m->cam.width, m->cam.height, 3.0, m->cam.width / 50, m->cam.height / 50, m->d_eq,
                    (size_t) m->cam.width);
    if (rc) return rc;
    rc = alva_pyramid_build_from_gray(m->ctx, m->pyr[m->cur], m->d_eq, (size_t) m->cam.width);
    if (rc) return rc;
    if (g_alva_lane) ALVA_HIP(alva_stream_sync(m->st));   // in a group the tracker runs on the lane's stream: these images are on the session's own
    return ALVA_OK;
}

// Look-ahead (no reference counterpart; the reference receives one frame per call): the caller names the frame of its NEXT call, and its
// gray image + LK pyramid are enqueued on the SAME stream right behind this frame's pose kernels, into the rotation's free slot: they run
// while the host does its pose bookkeeping, the keyframe decision and the next frame's slot table -- a window in which the GPU is otherwise
// idle -- instead of at the start of the next call.  (First version: a second stream beside the pose solve, ordered by two events.  The
// cross-queue waits cost what the overlap gained: 2 721 vs 2 742 frames/s sustained, `profiles/r3j_bench_n1.json`.)  The next
// new_frame_device takes the slot over when its pointer is the hinted one, and rebuilds as usual when it is not: results never depend
// on hints.  A frame that enqueues no pose solve builds ahead at the end of the call.
void HipStages::hint_next_frame_device(const uint8_t *d_rgba) {
    if (m->clahe) return;   // (the CLAHE chain shares d_eq / context scratch with the current frame: no look-ahead there)
    m->ahead_src = d_rgba;
    m->ahead_enqueued = false;
}

int HipStages::build_ahead() {
    Impl &M = *m;
    if (!M.ahead_src || M.ahead_enqueued) return ALVA_OK;
    const int rc = alva_pyramid_build_from_rgba(M.ctx, M.pyr[M.nxt], M.ahead_src, (size_t) M.cam.width * 4, M.d_gray_next, (size_t) M.cam.width);
    if (rc) return rc;
    M.ahead_enqueued = true;
    return ALVA_OK;
}

// The caller's frame buffer is pageable memory (the reference's wasm heap): by default a frame goes through a pinned staging copy
// (one memcpy + one DMA).  A caller that reuses ONE buffer -- src/system.js allocates memImg once and writes every frame into it
// (:63-67, :175) -- registers it explicitly (register_frame_buffer = alva_system_register_frame_buffer): the pages are locked and
// mapped, and k_level0 (gray + pyramid level 0) then reads the frame straight out of host memory over PCIe -- no staging copy, no
// copy command, no separate device RGBA buffer.  Registration is the caller's statement about the buffer's lifetime; nothing is
// inferred from pointer values (an address seen twice says nothing about the pages behind it).
int HipStages::register_frame_buffer(const uint8_t *buf, size_t bytes) {
    ALVA_HIP(hipSetDevice(m->device));
    const int urc = unregister_frame_buffer();
    if (urc) return urc;
    if (!buf) return ALVA_OK;
    ALVA_ARG(bytes >= (size_t) m->cam.width * m->cam.height * 4 && ((uintptr_t) buf & 15) == 0);
    ALVA_HIP(hipHostRegister((void *) buf, bytes, hipHostRegisterMapped));
    void *dp = nullptr;
    if (hipHostGetDevicePointer(&dp, (void *) buf, 0) != hipSuccess || !dp) {
        (void) hipGetLastError();
        (void) hipHostUnregister((void *) buf);
        alva_set_error("alva_system_register_frame_buffer: the registered buffer has no device mapping");
        return ALVA_ERR_HIP;
    }
    m->registered = buf;
    m->registered_dev = (const uint8_t *) dp;
    m->registered_bytes = bytes;
    return ALVA_OK;
}

int HipStages::alloc_frame_buffer(size_t bytes, uint8_t **h_writable) {
    ALVA_HIP(hipSetDevice(m->device));
    ALVA_ARG(h_writable && bytes >= (size_t) m->cam.width * m->cam.height * 4);
    if (m->bar_frame) {
        ALVA_HIP(alva_stream_sync(m->st));
        ALVA_HIP(hipFree(m->bar_frame));
        m->bar_frame = nullptr;
        m->bar_frame_bytes = 0;
    }
    if (!Impl::host_can_store_to_device_memory(m->device) || getenv("ALVA_NO_BAR_FRAME")) {
        alva_set_error("alva_system_alloc_frame_buffer: device memory is not host-writable on this system (no large BAR)");
        return ALVA_ERR_STATE;   // the caller falls back to its registered host buffer (alvaar_amd/system.py)
    }
    // fine-grained (bar_alloc_flag): every frame rewrites the buffer from the host, an L2 must not answer with the previous frame's line
    if (hipExtMallocWithFlags((void **) &m->bar_frame, bytes, Impl::bar_alloc_flag()) != hipSuccess) {
        (void) hipGetLastError();
        m->bar_frame = nullptr;
        alva_set_error("alva_system_alloc_frame_buffer: no host-writable device memory on this system");
        return ALVA_ERR_STATE;
    }
    ALVA_HIP(Impl::scrub_host_written(m->bar_frame, bytes));
    m->bar_frame_bytes = bytes;
    *h_writable = m->bar_frame;
    return ALVA_OK;
}

int HipStages::unregister_frame_buffer() {
    if (!m->registered) return ALVA_OK;
    ALVA_HIP(hipSetDevice(m->device));
    ALVA_HIP(alva_stream_sync(m->st));   // no kernel may still be reading the pages
    m->upload_in_flight = false;
    const uint8_t *b = m->registered;
    m->registered = m->registered_dev = nullptr;
    m->registered_bytes = 0;
    ALVA_HIP(hipHostUnregister((void *) b));
    return ALVA_OK;
}

int HipStages::new_frame(const uint8_t *rgba) {
    ALVA_HIP(hipSetDevice(m->device));
    // a HOST frame inside a group (no group entry point takes one): the upload, its completion event and the kernels that read it belong on
    // the session's own stream -- the images are built there and waited for, so that the lane's tracker may read them
    struct NoLane {
        alva_lane *saved = g_alva_lane;
        NoLane() { g_alva_lane = nullptr; }
        ~NoLane() { g_alva_lane = saved; }
    } no_lane;
    const bool in_group = no_lane.saved != nullptr;
    const size_t bytes = (size_t) m->cam.width * m->cam.height * 4;
    if (m->bar_frame && rgba >= m->bar_frame && rgba + bytes <= m->bar_frame + m->bar_frame_bytes && (((uintptr_t) rgba) & 15) == 0) {
        // the caller stored the frame into device memory itself (alloc_frame_buffer): it IS a device frame; its write-combined stores
        // leave the core before the launch's doorbell does
        __builtin_ia32_sfence();
        const int rc = build_from(rgba);
        if (rc) return rc;
        if (in_group) ALVA_HIP(alva_stream_sync(m->st));
        else {
            // the buffer is the caller's again when the call returns (frame_done waits): a frame that makes no later stream wait -- too
            // few tracks, a reset -- must not leave the image kernels reading what the caller's next memImg.write overwrites
            m->upload_in_flight = true;
            ALVA_HIP(hipEventRecord(m->upload_done, m->st));
        }
        return ALVA_OK;
    }
    if (m->registered && rgba >= m->registered && rgba + bytes <= m->registered + m->registered_bytes && (((uintptr_t) rgba) & 15) == 0) {
        // zero-copy: the image kernels read the caller's pages; frame_done() waits for them before the call returns the buffer
        const int rc = build_from(m->registered_dev + (rgba - m->registered));
        if (rc) return rc;
        m->upload_in_flight = true;
        ALVA_HIP(hipEventRecord(m->upload_done, m->st));
        if (in_group) ALVA_HIP(alva_stream_sync(m->st));
        return ALVA_OK;
    }
    memcpy(m->h_rgba, rgba, bytes);
    ALVA_HIP(hipMemcpyAsync(m->d_rgba, m->h_rgba, bytes, hipMemcpyHostToDevice, m->st));
    const int rc = build_from(m->d_rgba);
    if (rc) return rc;
    if (in_group) ALVA_HIP(alva_stream_sync(m->st));
    return ALVA_OK;
}

int HipStages::new_frame_device(const uint8_t *d_rgba) {
    ALVA_HIP(hipSetDevice(m->device));
    return build_from(d_rgba);
}

int HipStages::frame_done() {
    if (m->ahead_src && !m->ahead_enqueued) {   // a frame that tracked nothing: the look-ahead goes out now
        const int rc = build_ahead();
        if (rc) return rc;
    }
    if (m->upload_in_flight) {
        m->upload_in_flight = false;
        ALVA_HIP(alva_event_sync(m->upload_done));
    }
    return ALVA_OK;
}

void HipStages::reset_images() {}  // the pyramids are rebuilt before they are read again (frame 0 tracks nothing)

// One tracking step as ONE device-side chain: glue -> tracker (one level, from the projected priors) -> glue -> tracker (full pyramid)
// -> glue, one host wait; then the pose solve (P3P-LMedS -> PnP, alva_compute_pose) is enqueued and collected by track_pose_collect
// after the map layer has done its tracker bookkeeping.  Inputs are read from and per-slot results written to pinned host memory
// by the kernels themselves: no copy commands.
int HipStages::track_begin(const TrackJob &job, TrackKlt &out) {
    if (!m->fused || (job.want_pose && !job.do_p3p)) {
        m->trk_valid_n = -1;
        if (job.carry) {
            alva_set_error("tracking step: a carried slot table on the composed path");
            return ALVA_ERR_STATE;
        }
        if (m->trk_in && job.n > 0 && job.px == m->track_in(m->trk_par ^ 1).px) {
            // the composed step READS the slot table on the host, and this one was written into device memory (track_slot_buffers): one
            // copy back instead of a load over the bus per element (p3pEnabled_ off: not the shipped configuration)
            static thread_local std::vector<uint8_t> back;
            const size_t n = (size_t) job.n;
            back.resize(n * 40);
            ALVA_HIP(hipSetDevice(m->device));
            __builtin_ia32_sfence();
            ALVA_HIP(hipMemcpy(back.data(), job.px, n * 8, hipMemcpyDeviceToHost));
            ALVA_HIP(hipMemcpy(back.data() + n * 8, job.wpt, n * 24, hipMemcpyDeviceToHost));
            ALVA_HIP(hipMemcpy(back.data() + n * 32, job.is3d, n, hipMemcpyDeviceToHost));
            TrackJob j2 = job;
            j2.px = (const float *) back.data();
            j2.wpt = (const double *) (back.data() + n * 8);
            j2.is3d = back.data() + n * 32;
            return Stages::track_begin(j2, out);
        }
        return Stages::track_begin(job, out);
    }
    m->pose_pending = false;
    pending_.active = false;
    const int n = job.n;
    out.code_v = nullptr;
    out.px_v = out.unpx_v = nullptr;
    out.bv_v = nullptr;
    out.p3p_req = 0;
    out.n_pose = 0;
    const int n_prev = m->trk_valid_n;
    m->trk_valid_n = -1;   // (set again at the end of a launch that leaves a complete table behind)
    if (n == 0) return ALVA_OK;
    ALVA_HIP(hipSetDevice(m->device));
    int rc = m->track_reserve(n);
    if (rc) return rc;
    const size_t c = (size_t) m->trk_cap;
    const alva_pyramid *prev = m->pyr[m->prev], *cur = m->pyr[m->cur];
    const Camera &k = m->cam;
    const int par = m->trk_par ^ 1;   // this frame's table and d_px
    // the table carried from the previous frame (track_carry_buffer said yes and the map layer filled the index): nothing to read on the host
    const bool carried = job.carry && m->bar_table && m->trk_in && !m->lists && job.carry == m->track_carry() && n_prev >= n && !g_alva_lane;
    if (job.carry && !carried) {
        alva_set_error("tracking step: a carried slot table without a previous frame's table to carry it from");
        return ALVA_ERR_STATE;
    }
    const bool in_device = carried || (m->bar_table && m->trk_in && !m->lists && job.px == m->track_in(par).px && job.is3d == m->track_in(par).is3d &&
                                       job.wpt == m->track_in(par).wpt);   // track_slot_buffers handed out the device table: it is written, never read here
    int n3d = 0;
    if (!in_device)
        for (int i = 0; i < n; i++) n3d += job.is3d[i] ? 1 : 0;
    const uint8_t *o_code = nullptr;
    const float *o_px = nullptr, *o_unpx = nullptr;
    const double *o_bv = nullptr, *Pbv = nullptr, *Puv = nullptr, *Pwpt = nullptr;
    const int *o_hdr = nullptr;
    int poll_seq = 0;
    TrackSlots slots_D{};
    bool slots_path = false, pose_all = false;
    const Impl::TrackPin pin = m->track_pin();
    const bool staged = job.px == pin.in_px && job.is3d == pin.in_is3d && job.wpt == pin.in_wpt;   // track_slot_buffers was used
    if (in_device) {
        __builtin_ia32_sfence();   // the table's write-combined stores leave the core before the launch's doorbell does
    } else if (!staged) {
        memcpy(pin.in_px, job.px, (size_t) n * 8);
        memcpy(pin.in_is3d, job.is3d, (size_t) n);
        memcpy(pin.in_wpt, job.wpt, (size_t) n * 24);
    }
    if (!m->lists) {
        TrackSlots D{};
        uint8_t *b = m->trk_dev.base;
        D.cnt = (int *) b; b += 1024;
        D.d_code = b; b += c;
        D.d_is3d = b; b += c;    // c is a multiple of 1024: every block below starts 16-byte aligned (k_track_stage_in copies in 16-byte units)
        b += 256 - ((uintptr_t) b & 255);
        D.d_pts = (float *) b; b += c * 8;
        D.d_retried = b; b += c;
        D.d_px = (float *) b; b += c * 8;
        D.d_unpx = (float *) b; b += c * 8;
        D.d_bv = (double *) b; b += c * 24;
        D.d_wpt = (double *) b; b += c * 24;
        D.Pbv = (double *) b; b += c * 24;
        D.Puv = (double *) b; b += c * 16;
        D.Pwpt = (double *) b; b += c * 24;
        b += 256 - ((uintptr_t) b & 255);
        float *d_px_alt = (float *) b; b += c * 8;
        float *const d_px2[2] = {D.d_px, d_px_alt};   // the tracked positions of consecutive frames alternate (a carried table reads the previous frame's)
        D.d_px = d_px2[par];
        D.in_px = pin.in_px; D.in_is3d = pin.in_is3d; D.in_wpt = pin.in_wpt;
        if (in_device) {   // the table is where the tracker reads it: no copy kernel (in_px == nullptr tells alva_track_slots_klt)
            const Impl::TrackIn T = m->track_in(par);
            D.d_pts = T.px; D.d_is3d = T.is3d; D.d_wpt = T.wpt;
            D.in_px = nullptr; D.in_is3d = nullptr; D.in_wpt = nullptr;
        }
        if (carried) {
            const Impl::TrackIn Tp = m->track_in(par ^ 1);
            D.carry = m->track_carry();
            D.p_px = d_px2[par ^ 1];
            D.p_is3d = Tp.is3d;
            D.p_wpt = Tp.wpt;
        }
        D.o_hdr = pin.o_hdr; D.o_code = pin.o_code; D.o_px = pin.o_px; D.o_unpx = pin.o_unpx; D.o_bv = pin.o_bv;
        D.n = n;
        D.use_prior = job.use_prior;
        D.width = m->cam.width;
        D.height = m->cam.height;
        memcpy(D.q, job.Tcw_q, 32);
        memcpy(D.t, job.Tcw_t, 24);
        D.cam = AlvaCam{k.fx, k.fy, k.cx, k.cy, k.k1, k.k2, k.p1, k.p2};
        D.invK = m->d_invK;
        D.dbg = alva_klt_stamp_buffer();
        // state.hpp:50-56 constants; the prior pass works on one pyramid level (visual_frontend.cpp:166)
        D.seq = ++m->trk_seq;   // the tracker launch publishes its counts under this number too (the word at o_hdr[10])
        rc = alva_track_slots_klt(m->ctx, prev, cur, D, 1, job.klt_levels, 30.f, 0.5f, 30, 0.01f, 0);
        if (rc) return rc;
        // The frame's tail -- compaction -> P3P-LMedS -> refinement -- as ONE launch queued right here, behind the tracker (pnp.hip
        // k_pose_all): its first workgroups are the compaction, the others wait for the host's word, which goes out below as soon as
        // the tracker's early word has told the host how many correspondences there are.  A session of its own, polling, that wants
        // a pose; everything else keeps the compaction kernel.
        pose_all = m->poll && job.want_pose && job.do_p3p && alva_pose_all_possible(D.n, 100);
        if (pose_all) {
            // (the launch has ~128 workgroups anyway: 64-slot slices -- the slice length is a multiple of the wave -- instead of 256-slot ones)
            rc = alva_pose_all_enqueue(m->ctx, D, std::min(96, std::max(1, (D.n + 63) / 64)), 100, 3.0f, job.do_random, 12345u, 5, 5.9915f, (float) k.fx, (float) k.fy,
                                       (float) k.cx, (float) k.cy);  // state.hpp:68-69, visual_frontend.cpp:363-375
            if (rc) return rc;
        } else if (!alva_lane_defer(MK_TRACK_COMPACT, m->ctx, (unsigned) compact_grid(D.n), 0, &D, sizeof(D))) {
            hipLaunchKernelGGL(k_track_compact, dim3(compact_grid(D.n)), dim3(CMP_NT), 0, m->st, D);
            ALVA_LAUNCH_CHECK();
        }
        if (in_device) {   // this frame's table and (once the launch is through) its tracked positions are complete on the device
            m->trk_par = par;
            m->trk_valid_n = n;
        }
        poll_seq = m->poll ? D.seq : 0;
        slots_D = D;
        slots_path = true;
        o_code = D.o_code; o_px = D.o_px; o_unpx = D.o_unpx; o_bv = D.o_bv; o_hdr = D.o_hdr;
        Pbv = D.Pbv; Puv = D.Puv; Pwpt = D.Pwpt;
    } else {
    TrackDev D{};
    {
        uint8_t *b = m->trk_dev.base;
        D.cnt = (int *) b; b += 1024;
        D.slotA = (int *) b; b += c * 4;
        D.slotB = (int *) b; b += c * 4;
        D.ptsA = (float *) b; b += c * 8;
        D.priorA = (float *) b; b += c * 8;
        D.outA = (float *) b; b += c * 8;
        D.ptsB = (float *) b; b += c * 8;
        D.priorB = (float *) b; b += c * 8;
        D.outB = (float *) b; b += c * 8;
        D.stA = b; b += c;
        D.stB = b; b += c;
        D.d_is3d = b; b += c;
        D.d_code = b; b += c;
        b += 256 - ((uintptr_t) b & 255);
        D.d_wpt = (double *) b; b += c * 24;
        D.d_px = (float *) b; b += c * 8;
        D.Pbv = (double *) b; b += c * 24;
        D.Puv = (double *) b; b += c * 16;
        D.Pwpt = (double *) b; b += c * 24;
        D.in_px = pin.in_px; D.in_is3d = pin.in_is3d; D.in_wpt = pin.in_wpt;
        D.o_hdr = pin.o_hdr; D.o_code = pin.o_code; D.o_px = pin.o_px; D.o_unpx = pin.o_unpx; D.o_bv = pin.o_bv;
    }
    D.n = n;
    D.use_prior = job.use_prior;
    D.width = m->cam.width;
    D.height = m->cam.height;
    memcpy(D.q, job.Tcw_q, 32);
    memcpy(D.t, job.Tcw_t, 24);
    D.cam = AlvaCam{k.fx, k.fy, k.cx, k.cy, k.k1, k.k2, k.p1, k.p2};
    D.invK = m->d_invK;
    hipLaunchKernelGGL(k_track_prepare, dim3(1), dim3(TRK_NT), 0, m->st, D);
    if (job.use_prior && n3d > 0)  // state.hpp:50-56 constants; one pyramid level (visual_frontend.cpp:166)
        rc = alva_fbklt_track_dn(m->ctx, prev, cur, 1, 30.f, 0.5f, 30, 0.01f, D.ptsA, D.priorA, D.outA, D.stA, D.cnt + 0, n3d);
    if (rc) return rc;
    hipLaunchKernelGGL(k_track_pass2, dim3(1), dim3(TRK_NT), 0, m->st, D);
    rc = alva_fbklt_track_dn(m->ctx, prev, cur, job.klt_levels, 30.f, 0.5f, 30, 0.01f, D.ptsB, D.priorB, D.outB, D.stB, D.cnt + 2, n);
    if (rc) return rc;
    hipLaunchKernelGGL(k_track_finish, dim3(1), dim3(TRK_NT), 0, m->st, D);
    ALVA_LAUNCH_CHECK();
    o_code = D.o_code; o_px = D.o_px; o_unpx = D.o_unpx; o_bv = D.o_bv; o_hdr = D.o_hdr;
    Pbv = D.Pbv; Puv = D.Puv; Pwpt = D.Pwpt;
    }
    int step_req = 0, step_n_pose = 0;   // the slot-wise step's header, out of its completion word
    auto wait_step = [&](int seq) -> int {
        if (seq && slots_path) {
            // the compaction kernel publishes [seq | p3pReq_ | n_pose] as one word after all results; spinning on it in pinned memory
            // returns a few microseconds before hipStreamSynchronize would
            const volatile unsigned long long *flag = reinterpret_cast<const volatile unsigned long long *>(o_hdr + 12);
            unsigned spins = 0;
            unsigned long long word = *flag;
            while ((int) (word >> 32) != seq) {
                if (++spins > (1u << 26)) {   // ~ seconds: something is wrong with the stream; let the runtime report it
                    ALVA_HIP(alva_stream_sync(m->st));
                    word = *flag;
                    break;
                }
                alva_poll_relax(spins);
                word = *flag;
            }
            __atomic_thread_fence(__ATOMIC_ACQUIRE);
            step_req = (int) ((word >> 31) & 1);
            step_n_pose = (int) (word & 0x7fffffffu);
        } else if (seq) {
            // the compaction kernel publishes its sequence number after all results (system-scope release); spinning on that word in
            // pinned memory returns a few microseconds before hipStreamSynchronize would
            const volatile int *flag = o_hdr + 8;
            unsigned spins = 0;
            while (*flag != seq) {
                if (++spins > (1u << 26)) {   // ~ seconds: something is wrong with the stream; let the runtime report it
                    ALVA_HIP(alva_stream_sync(m->st));
                    break;
                }
                alva_poll_relax(spins);
            }
            __atomic_thread_fence(__ATOMIC_ACQUIRE);
        } else {
            ALVA_HIP(alva_stream_sync(m->st));
            if (slots_path) {
                const unsigned long long word = *reinterpret_cast<const volatile unsigned long long *>(o_hdr + 12);
                step_req = (int) ((word >> 31) & 1);
                step_n_pose = (int) (word & 0x7fffffffu);
            }
        }
        return ALVA_OK;
    };
    // The tracker launch's last workgroup has published the step's counts one kernel EARLIER (the 64-bit word at o_hdr[10], track_slots.hpp): in the
    // normal case (no p3pReq_) the pose solve is enqueued NOW -- host-side sample draw + two launches, queued behind the compaction
    // kernel in stream order -- instead of after the compaction's completion word: the GPU goes from the compaction straight into P3P.
    bool pose_early = false;
    int early_n_pose = -1;
    const int n_pose_cap = 19000;   // P3P-LMedS keeps its median in LDS: at most 19000 correspondences (the first ones, in slot order)
    if (slots_path && poll_seq && job.want_pose) {
        const volatile unsigned long long *early = reinterpret_cast<const volatile unsigned long long *>(o_hdr + 10);
        unsigned spins = 0;
        unsigned long long word = *early;
        while ((int) (word >> 32) != poll_seq && ++spins < (1u << 26)) {
            alva_poll_relax(spins);
            word = *early;
        }
        early_n_pose = (int) (word & 0x7fffffffu);
        const bool early_ok = (int) (word >> 32) == poll_seq && !((word >> 31) & 1) && early_n_pose >= 4;
        if (pose_all && !early_ok) {
            (void) alva_pose_all_abort(m->ctx);   // p3pReq_, fewer than four correspondences, or no word at all: the queued launch ends after its compaction
            pose_all = false;
        }
        if (early_ok) {
            m->pose_n = early_n_pose > n_pose_cap ? n_pose_cap : early_n_pose;
            if (pose_all) rc = alva_pose_all_go(m->ctx, m->pose_n);   // the queued launch's second phase: samples for this n, go
            else
            rc = alva_compute_pose_enqueue(m->ctx, Pbv, Puv, Pwpt, m->pose_n, 100, 3.0f, job.do_random, 12345u, 5, 5.9915f, (float) k.fx,
                                           (float) k.fy, (float) k.cx, (float) k.cy);  // state.hpp:68-69, visual_frontend.cpp:363-375
            pose_all = false;
            if (rc) return rc;
            pose_early = true;
            rc = build_ahead();   // a hinted next frame: its images are queued behind the pose kernels
            if (rc) return rc;
            alva_lane_yield();    // in a group: the thread's other sessions deposit their pose solves before this one starts its bookkeeping
        }
    }
    rc = wait_step(poll_seq);
    if (rc) return rc;
    int p3p_req = slots_path ? step_req : o_hdr[4];
    if (slots_path && p3p_req) {
        // fewer than 33 % of the one-level passes held (visual_frontend.cpp:193-203): the retries must start from the keypoints' own
        // positions instead -- redo them and compact again (rare: tracking is about to be lost)
        rc = alva_track_slots_klt(m->ctx, prev, cur, slots_D, 1, job.klt_levels, 30.f, 0.5f, 30, 0.01f, 1);
        if (rc) return rc;
        slots_D.seq = ++m->trk_seq;
        hipLaunchKernelGGL(k_track_compact, dim3(compact_grid(slots_D.n)), dim3(CMP_NT), 0, m->st, slots_D);
        ALVA_LAUNCH_CHECK();
        rc = wait_step(m->poll ? slots_D.seq : 0);
        if (rc) return rc;
    }
    out.code_v = o_code;   // read in place (pinned host memory, written by the kernels; stays until the next track_begin)
    out.px_v = o_px;
    out.unpx_v = o_unpx;
    out.bv_v = o_bv;
    out.p3p_req = p3p_req;
    out.n_pose = slots_path ? step_n_pose : o_hdr[5];
    if (pose_early) {
        if (p3p_req || out.n_pose != early_n_pose) {   // cannot happen: both kernels count the same flags
            alva_set_error("tracking step: the tracker's early counts (%d) disagree with the compaction (%d)", early_n_pose, out.n_pose);
            return ALVA_ERR_STATE;
        }
        m->pose_pending = true;
    } else if (job.want_pose && out.n_pose >= 4) {
        m->pose_n = out.n_pose > n_pose_cap ? n_pose_cap : out.n_pose;
        rc = alva_compute_pose_enqueue(m->ctx, Pbv, Puv, Pwpt, m->pose_n, 100, 3.0f, job.do_random, 12345u, 5, 5.9915f, (float) k.fx,
                                       (float) k.fy, (float) k.cx, (float) k.cy);  // state.hpp:68-69, visual_frontend.cpp:363-375
        if (rc) return rc;
        m->pose_pending = true;
    }
    m->pose_n = out.n_pose >= 4 ? m->pose_n : 0;
    pose_total_ = out.n_pose;
    fused_active_ = true;
    if (!m->pose_pending) alva_lane_clean();   // in a group: the compaction's word was the chain's last (lane.hpp); with a pose solve, its own is
    return ALVA_OK;
}

bool HipStages::track_slot_buffers(int n, float **px, uint8_t **is3d, double **wpt) {
    if (!m->fused || n <= 0) return false;
    if (hipSetDevice(m->device) != hipSuccess || m->track_reserve(n) != ALVA_OK) return false;
    if (m->bar_table && m->trk_in && !m->lists) {   // device memory, written in place (track_reserve): the table the NEXT launch takes
        const Impl::TrackIn T = m->track_in(m->trk_par ^ 1);
        *px = T.px;
        *is3d = T.is3d;
        *wpt = T.wpt;
        return true;
    }
    const Impl::TrackPin pin = m->track_pin();
    *px = pin.in_px;
    *is3d = pin.in_is3d;
    *wpt = pin.in_wpt;
    return true;
}

uint16_t *HipStages::track_carry_buffer(int n_prev, int n) {
    if (!m->fused || !m->carry_ok || m->lists || !m->bar_table || !m->trk_in || g_alva_lane) return nullptr;
    if (n <= 0 || n > n_prev || n_prev != m->trk_valid_n || n_prev >= 65536 || n + 1 > m->trk_cap) return nullptr;   // (the caller may write entry n: room for n + 1)
    return m->track_carry();
}

int HipStages::track_pose_collect(TrackPose &out) {
    if (!fused_active_) return Stages::track_pose_collect(out);
    fused_active_ = false;
    out.status = -1;
    out.p3p_outlier.assign((size_t) pose_total_, 0);
    out.pnp_outlier.assign((size_t) pose_total_, 0);
    if (!m->pose_pending) return ALVA_OK;
    m->pose_pending = false;
    int status = 0;
    int rc = alva_compute_pose_collect_p3p(m->ctx, out.pose7, out.pose7_p3p, out.p3p_outlier.data(), out.pnp_outlier.data(), &status);
    if (rc) return rc;
    alva_lane_clean();   // the refinement's completion word has been seen: nothing of this session is left on the lane
    out.status = status;
    return ALVA_OK;
}

int HipStages::fbklt(int levels, int n, const float *pts, float *prior, uint8_t *status) {
    if (n <= 0) return ALVA_OK;
    Impl::Plan p;
    const size_t a = p.add((size_t) n * 8), b = p.add((size_t) n * 8), c = p.add((size_t) n);
    std::vector<uint8_t *> d, h;
    int rc = m->carve(p, d, h);
    if (rc) return rc;
    UP(a, pts, (size_t) n * 8);
    UP(b, prior, (size_t) n * 8);
    // state.hpp:50-56: kltError_ 30, kltMaxFbDistance_ 0.5, 30 iterations, 0.01 px
    rc = alva_fbklt_track(m->ctx, m->pyr[m->prev], m->pyr[m->cur], levels, 30.f, 0.5f, 30, 0.01f, (const float *) d[a], (float *) d[b], d[c], n);
    if (rc) return rc;
    DOWN(b, (size_t) n * 8);
    DOWN(c, (size_t) n);
    ALVA_HIP(alva_stream_sync(m->st));
    memcpy(prior, h[b], (size_t) n * 8);
    memcpy(status, h[c], (size_t) n);
    return ALVA_OK;
}

int HipStages::compute_keypoints(int n, const float *px, float *unpx, double *bv) {
    if (n <= 0) return ALVA_OK;
    Impl::Plan p;
    const size_t a = p.add((size_t) n * 8), b = p.add((size_t) n * 8), c = p.add((size_t) n * 24);
    std::vector<uint8_t *> d, h;
    int rc = m->carve(p, d, h);
    if (rc) return rc;
    UP(a, px, (size_t) n * 8);
    const Camera &k = m->cam;
    rc = alva_undistort_points(m->ctx, (const float *) d[a], n, k.fx, k.fy, k.cx, k.cy, k.k1, k.k2, k.p1, k.p2, (float *) d[b]);
    if (rc) return rc;
    hipLaunchKernelGGL(k_bearing, dim3(alva_divup(n, 256)), dim3(256), 0, m->st, (const float *) d[b], n, m->d_invK, (double *) d[c]);
    ALVA_LAUNCH_CHECK();
    rc = m->down_span(p, d, h, b, c);
    if (rc) return rc;
    ALVA_HIP(alva_stream_sync(m->st));
    memcpy(unpx, h[b], (size_t) n * 8);
    memcpy(bv, h[c], (size_t) n * 24);
    return ALVA_OK;
}

int HipStages::project_dist(int n, const double *cam_pts, float *px) {
    if (n <= 0) return ALVA_OK;
    Impl::Plan p;
    const size_t a = p.add((size_t) n * 24), b = p.add((size_t) n * 8);
    std::vector<uint8_t *> d, h;
    int rc = m->carve(p, d, h);
    if (rc) return rc;
    UP(a, cam_pts, (size_t) n * 24);
    const Camera &k = m->cam;
    rc = alva_project_dist(m->ctx, (const double *) d[a], n, k.fx, k.fy, k.cx, k.cy, k.k1, k.k2, k.p1, k.p2, (float *) d[b]);
    if (rc) return rc;
    DOWN(b, (size_t) n * 8);
    ALVA_HIP(alva_stream_sync(m->st));
    memcpy(px, h[b], (size_t) n * 8);
    return ALVA_OK;
}

int HipStages::p3p(int n, const double *bv, const double *wpt, int do_random, double *pose7, int *outliers, int *n_outliers, int *ok) {
    *ok = 0;
    *n_outliers = 0;
    if (n < 4) return ALVA_OK;  // multi_view_geometry.cpp:40-43
    // the LMedS median lives in LDS: at most 19000 correspondences per call; a larger frame is solved on its first 19000 keypoints
    // (container order) and the rest is left to the PnP's chi2 sweep
    const int nn = n > 19000 ? 19000 : n;
    Impl::Plan p;
    const size_t a = p.add((size_t) nn * 24), b = p.add((size_t) nn * 24);
    std::vector<uint8_t *> d, h;
    int rc = m->carve(p, d, h);
    if (rc) return rc;
    UP(a, bv, (size_t) nn * 24);
    UP(b, wpt, (size_t) nn * 24);
    double R[9], t[3];
    rc = alva_p3p_lmeds(m->ctx, (const double *) d[a], (const double *) d[b], nn, 100, 3.0f, do_random, 12345u, (float) m->cam.fx, (float) m->cam.fy,
                        R, t, outliers, n_outliers, ok);  // state.hpp:68-69
    if (rc) return rc;
    if (*ok) {
        SE3 T;
        rot_to_quat(R, T.q);  // Twc.setRotationMatrix (multi_view_geometry.cpp:104-105)
        for (int i = 0; i < 3; i++) T.t[i] = t[i];
        se3_to_pose7(T, pose7);
    } else {
        *n_outliers = 0;
    }
    return ALVA_OK;
}

int HipStages::pnp(int n, const double *unpx_d, const double *wpt, double *pose7, int *outliers, int *n_outliers, int *ok) {
    *ok = 0;
    *n_outliers = 0;
    if (n <= 0) return ALVA_OK;
    Impl::Plan p;
    const size_t a = p.add((size_t) n * 16), b = p.add((size_t) n * 24);
    std::vector<uint8_t *> d, h;
    int rc = m->carve(p, d, h);
    if (rc) return rc;
    UP(a, unpx_d, (size_t) n * 16);
    UP(b, wpt, (size_t) n * 24);
    double info[8];
    const Camera &k = m->cam;
    // visual_frontend.cpp:363-375: 5 iterations, robustCostThreshold_ 5.9915, robust + L2 refinement
    return alva_pnp_refine(m->ctx, (const double *) d[a], (const double *) d[b], n, pose7, 5, 5.9915f, 1, 1, (float) k.fx, (float) k.fy, (float) k.cx,
                           (float) k.cy, outliers, n_outliers, info, ok);
}

int HipStages::five_point(int n, const double *bv_kf, const double *bv_cur, int do_random, double *R, double *t, int *outliers, int *n_outliers,
                          int *ok) {
    *ok = 0;
    *n_outliers = 0;
    if (n < 8) return ALVA_OK;
    Impl::Plan p;
    const size_t a = p.add((size_t) n * 24), b = p.add((size_t) n * 24);
    std::vector<uint8_t *> d, h;
    int rc = m->carve(p, d, h);
    if (rc) return rc;
    UP(a, bv_kf, (size_t) n * 24);
    UP(b, bv_cur, (size_t) n * 24);
    std::vector<uint8_t> inl((size_t) n);
    rc = alva_compute_5pt_essential(m->ctx, (const double *) d[a], (const double *) d[b], n, 100, 3.0f, 1, do_random, 12345u, (float) m->cam.fx,
                                    (float) m->cam.fy, R, t, inl.data(), nullptr, ok);
    if (rc) return rc;
    if (*ok)
        for (int i = 0; i < n; i++)
            if (!inl[(size_t) i]) outliers[(*n_outliers)++] = i;
    return ALVA_OK;
}

int HipStages::detect(int cell, int n_occ, const float *occupied, int cap, float *pts, int *count) {
    const int rc = detect_begin(cell, n_occ, occupied, cap);
    return rc ? rc : detect_end(pts, count);
}

// the detector enqueued on the stream (all its launches + the copy of the whole output buffer back: the count is not known yet); the
// map layer updates its descriptor medoids meanwhile
int HipStages::detect_begin(int cell, int n_occ, const float *occupied, int cap) {
    Impl::Plan p;
    const size_t a = p.add((size_t) (n_occ > 0 ? n_occ : 1) * 8), b = p.add((size_t) cap * 8);
    std::vector<uint8_t *> d, h;
    int rc = m->carve(p, d, h);
    if (rc) return rc;
    // Small keyframe stages read their inputs from and write their results to the PINNED mirror of the plan directly (h[], device-
    // accessible, uncached by the GPU): a copy command is a blit kernel of its own (~9 us on the GPU + a launch + two queue gaps), and these
    // stages move 3 - 80 KB.  The wait that ends each stage is a stream synchronisation, which makes the kernels' writes visible.
    if (n_occ > 0) memcpy(h[a], occupied, (size_t) n_occ * 8);
    const Camera &k = m->cam;
    const uint8_t *img = m->clahe ? m->d_eq : m->d_gray;  // detection runs on currImage_ (map_manager.cpp:213)
    // roi = CameraCalibration::roi_rect_ (camera_calibration.cpp:20): the image minus a border of 20 px
    rc = alva_detect_grid_enqueue(m->ctx, img, (size_t) k.width, k.width, k.height, cell, (const float *) h[a], n_occ, k.border, k.border,
                                  k.width - 2 * k.border, k.height - 2 * k.border, m->max_quality, (float *) h[b], cap, &m->det_pending);
    if (rc) return rc;
    m->det_h = h[b];
    m->det_cap = cap;
    return ALVA_OK;
}

int HipStages::detect_end(float *pts, int *count) {
    const int rc = alva_detect_grid_collect(m->ctx, &m->det_pending, &m->max_quality, count);   // waits for the stream: the copy above is done too
    if (rc) return rc;
    if (*count > m->det_cap) *count = m->det_cap;
    if (*count > 0) memcpy(pts, m->det_h, (size_t) *count * 8);
    return ALVA_OK;
}

int HipStages::describe(int n, const float *pts, uint8_t *desc, uint8_t *valid) {
    const int rc = describe_begin(n, pts);
    return rc ? rc : describe_end(desc, valid);
}

// the description enqueued on the stream (results land in the pinned mirror of the plan); the map layer enters the new keyframe into its
// map points' records meanwhile (no stage call in between: the next one carves the same staging)
int HipStages::describe_begin(int n, const float *pts) {
    m->dsc_n = n > 0 ? n : 0;
    if (n <= 0) return ALVA_OK;
    Impl::Plan p;
    const size_t a = p.add((size_t) n * 8), b = p.add((size_t) n * 32), c = p.add((size_t) n);
    std::vector<uint8_t *> d, h;
    int rc = m->carve(p, d, h);
    if (rc) return rc;
    memcpy(h[a], pts, (size_t) n * 8);
    rc = alva_describe(m->ctx, m->d_gray, (size_t) m->cam.width, m->cam.width, m->cam.height, (const float *) h[a], n, h[b], h[c]);
    if (rc) return rc;
    m->dsc_desc = h[b];
    m->dsc_valid = h[c];
    return ALVA_OK;
}

int HipStages::describe_end(uint8_t *desc, uint8_t *valid) {
    const int n = m->dsc_n;
    m->dsc_n = 0;
    if (n <= 0) return ALVA_OK;
    ALVA_HIP(alva_stream_sync(m->st));
    memcpy(desc, m->dsc_desc, (size_t) n * 32);
    memcpy(valid, m->dsc_valid, (size_t) n);
    return ALVA_OK;
}

int HipStages::describe_and_compute(int n, const float *pts, uint8_t *desc, uint8_t *valid, float *unpx, double *bv) {
    if (n <= 0) return ALVA_OK;
    Impl::Plan p;
    const size_t a = p.add((size_t) n * 8), b = p.add((size_t) n * 32), c = p.add((size_t) n), u = p.add((size_t) n * 8), v = p.add((size_t) n * 24);
    std::vector<uint8_t *> d, h;
    int rc = m->carve(p, d, h);
    if (rc) return rc;
    memcpy(h[a], pts, (size_t) n * 8);
    rc = alva_describe(m->ctx, m->d_gray, (size_t) m->cam.width, m->cam.width, m->cam.height, (const float *) h[a], n, h[b], h[c]);
    if (rc) return rc;
    const Camera &k = m->cam;
    rc = alva_undistort_points(m->ctx, (const float *) h[a], n, k.fx, k.fy, k.cx, k.cy, k.k1, k.k2, k.p1, k.p2, (float *) h[u]);
    if (rc) return rc;
    hipLaunchKernelGGL(k_bearing, dim3(alva_divup(n, 256)), dim3(256), 0, m->st, (const float *) h[u], n, m->d_invK, (double *) h[v]);
    ALVA_LAUNCH_CHECK();
    ALVA_HIP(alva_stream_sync(m->st));   // descriptors | validity | undistorted positions | bearings: one wait
    memcpy(desc, h[b], (size_t) n * 32);
    memcpy(valid, h[c], (size_t) n);
    memcpy(unpx, h[u], (size_t) n * 8);
    memcpy(bv, h[v], (size_t) n * 24);
    return ALVA_OK;
}

int HipStages::triangulate(int n, int n_groups, const double *T36, const int *group, const double *bv_l, const double *bv_r, const float *unpx_l,
                           const float *unpx_r, double *wpt, double *inv_depth, uint8_t *status, double *parallax) {
    if (n <= 0) return ALVA_OK;
    Impl::Plan p;
    const size_t iT = p.add((size_t) n_groups * 288), ig = p.add((size_t) n * 4), il = p.add((size_t) n * 24), ir = p.add((size_t) n * 24),
                 iul = p.add((size_t) n * 8), iur = p.add((size_t) n * 8), ilp = p.add((size_t) n * 24), iw = p.add((size_t) n * 24),
                 iid = p.add((size_t) n * 8), ist = p.add((size_t) n), ipar = p.add((size_t) n * 8);
    std::vector<uint8_t *> d, h;
    int rc = m->carve(p, d, h);
    if (rc) return rc;
    memcpy(h[iT], T36, (size_t) n_groups * 288);
    memcpy(h[ig], group, (size_t) n * 4);
    memcpy(h[il], bv_l, (size_t) n * 24);
    memcpy(h[ir], bv_r, (size_t) n * 24);
    memcpy(h[iul], unpx_l, (size_t) n * 8);
    memcpy(h[iur], unpx_r, (size_t) n * 8);
    const Camera &k = m->cam;
    rc = alva_triangulate(m->ctx, n, (const double *) h[iT], n_groups, (const int *) h[ig], (const double *) h[il], (const double *) h[ir],
                          (const float *) h[iul], (const float *) h[iur], k.fx, k.fy, k.cx, k.cy, 3.0f /* mapMaxReprojectionError_ */,
                          (double *) d[ilp], (double *) h[iw], (double *) h[iid], h[ist], (double *) h[ipar]);
    if (rc) return rc;
    ALVA_HIP(alva_stream_sync(m->st));
    memcpy(wpt, h[iw], (size_t) n * 24);
    memcpy(inv_depth, h[iid], (size_t) n * 8);
    memcpy(status, h[ist], (size_t) n);
    memcpy(parallax, h[ipar], (size_t) n * 8);
    return ALVA_OK;
}

uint8_t *HipStages::stage_scratch(size_t bytes) {
    // the call that follows carves [0, bytes) of both arenas: its arrays are already in place in the pinned one
    if (m->dev.grow(bytes, m->st) != ALVA_OK || m->pin.grow(bytes, m->st) != ALVA_OK) return nullptr;
    return m->pin.base;
}

int HipStages::match_to_map(int cell_size, int num_cells_w, int grid_cells, const int *cell_ptr, const int *cell_mp, int n_kf, const double *kf_q,
                            const double *kf_t, int n_mp, const double *mp_wpt, const uint8_t *mp_is3d, const uint8_t *mp_has_desc,
                            const int *obs_ptr, const int *obs_kf, const float *obs_px, const uint8_t *obs_desc, const uint8_t *obs_has_desc,
                            int frame_kf, int num_keypoints_3d, int n_local, const int *local, float max_proj_err, float dist_ratio,
                            int *match_of_mp) {
    if (n_mp <= 0) return ALVA_OK;
    const int n_obs = obs_ptr[n_mp], n_cell = cell_ptr[grid_cells];
    const Camera &k = m->cam;
    const double calib[10] = {k.fx, k.fy, k.cx, k.cy, k.k1, k.k2, k.p1, k.p2, (double) k.width, (double) k.height};
    // The map layer assembled the arrays in stage_scratch() (= the pinned arena): they go up with ONE copy of the span they cover and the
    // kernels read the same offsets of the device arena; the matches come back into the caller's array, which lies in the span too.
    const uint8_t *pb = m->pin.base, *pe = pb ? pb + m->pin.cap : nullptr;
    const uint8_t *ptrs[] = {(const uint8_t *) cell_ptr, (const uint8_t *) cell_mp, (const uint8_t *) kf_q, (const uint8_t *) kf_t, (const uint8_t *) mp_wpt,
                             mp_is3d, mp_has_desc, (const uint8_t *) obs_ptr, (const uint8_t *) obs_kf, (const uint8_t *) obs_px, obs_desc, obs_has_desc,
                             (const uint8_t *) local, (const uint8_t *) match_of_mp};
    const size_t lens[] = {(size_t) (grid_cells + 1) * 4, (size_t) n_cell * 4, (size_t) n_kf * 32, (size_t) n_kf * 24, (size_t) n_mp * 24, (size_t) n_mp,
                           (size_t) n_mp, (size_t) (n_mp + 1) * 4, (size_t) n_obs * 4, (size_t) n_obs * 8, (size_t) n_obs * 32, (size_t) n_obs,
                           (size_t) n_local * 4, (size_t) n_mp * 4};
    bool in_place = pb != nullptr;
    size_t lo = (size_t) -1, hi = 0;
    for (int i = 0; i < 14 && in_place; i++) {
        if (lens[i] == 0) continue;
        if (ptrs[i] < pb || ptrs[i] + lens[i] > pe) in_place = false;
        else {
            if (i < 13) {   // inputs: the span to upload
                lo = std::min(lo, (size_t) (ptrs[i] - pb));
                hi = std::max(hi, (size_t) (ptrs[i] - pb) + lens[i]);
            }
        }
    }
    if (in_place && m->dev.cap >= m->pin.cap) {
        auto dv = [&](const void *hp) { return m->dev.base + ((const uint8_t *) hp - pb); };
        ALVA_HIP(hipMemcpyAsync(m->dev.base + lo, pb + lo, hi - lo, hipMemcpyHostToDevice, m->st));
        int rc = alva_match_to_map_flags(m->ctx, calib, cell_size, num_cells_w, grid_cells, (const int *) dv(cell_ptr), (const int *) dv(cell_mp), n_kf,
                                         (const double *) dv(kf_q), (const double *) dv(kf_t), n_mp, (const double *) dv(mp_wpt), dv(mp_is3d),
                                         dv(mp_has_desc), (const int *) dv(obs_ptr), (const int *) dv(obs_kf), (const float *) dv(obs_px), dv(obs_desc),
                                         dv(obs_has_desc), frame_kf, num_keypoints_3d, n_local, (const int *) dv(local), max_proj_err, dist_ratio,
                                         match_of_mp);   // (pinned, written once per entry by the last kernel: no copy back)
        if (rc) return rc;
        ALVA_HIP(alva_stream_sync(m->st));
        return ALVA_OK;
    }
    Impl::Plan p;
    const size_t icp = p.add((size_t) (grid_cells + 1) * 4), icm = p.add((size_t) (n_cell > 0 ? n_cell : 1) * 4), iq = p.add((size_t) n_kf * 32),
                 it = p.add((size_t) n_kf * 24), iw = p.add((size_t) n_mp * 24), i3 = p.add((size_t) n_mp), ihd = p.add((size_t) n_mp),
                 iop = p.add((size_t) (n_mp + 1) * 4), iok = p.add((size_t) (n_obs > 0 ? n_obs : 1) * 4),
                 iox = p.add((size_t) (n_obs > 0 ? n_obs : 1) * 8), iod = p.add((size_t) (n_obs > 0 ? n_obs : 1) * 32),
                 ioh = p.add((size_t) (n_obs > 0 ? n_obs : 1)), il = p.add((size_t) (n_local > 0 ? n_local : 1) * 4), im = p.add((size_t) n_mp * 4);
    std::vector<uint8_t *> d, h;
    int rc = m->carve(p, d, h);
    if (rc) return rc;
    UP(icp, cell_ptr, (size_t) (grid_cells + 1) * 4);
    UP(icm, cell_mp, (size_t) n_cell * 4);
    UP(iq, kf_q, (size_t) n_kf * 32);
    UP(it, kf_t, (size_t) n_kf * 24);
    UP(iw, mp_wpt, (size_t) n_mp * 24);
    UP(i3, mp_is3d, (size_t) n_mp);
    UP(ihd, mp_has_desc, (size_t) n_mp);
    UP(iop, obs_ptr, (size_t) (n_mp + 1) * 4);
    UP(iok, obs_kf, (size_t) n_obs * 4);
    UP(iox, obs_px, (size_t) n_obs * 8);
    UP(iod, obs_desc, (size_t) n_obs * 32);
    UP(ioh, obs_has_desc, (size_t) n_obs);
    UP(il, local, (size_t) n_local * 4);
    rc = alva_match_to_map_flags(m->ctx, calib, cell_size, num_cells_w, grid_cells, (const int *) d[icp], (const int *) d[icm], n_kf,
                                 (const double *) d[iq], (const double *) d[it], n_mp, (const double *) d[iw], d[i3], d[ihd], (const int *) d[iop],
                                 (const int *) d[iok], (const float *) d[iox], d[iod], d[ioh], frame_kf, num_keypoints_3d, n_local,
                                 (const int *) d[il], max_proj_err, dist_ratio, (int *) d[im]);
    if (rc) return rc;
    DOWN(im, (size_t) n_mp * 4);
    ALVA_HIP(alva_stream_sync(m->st));
    memcpy(match_of_mp, h[im], (size_t) n_mp * 4);
    return ALVA_OK;
}

// The record arena: pinned, zero-filled, never moved.  The kernels find a record through the device-resident table of chunk addresses.
MpRec *HipStages::mp_arena_chunk(int chunk) {
    if (chunk < 0 || chunk >= Impl::REC_TAB_CAP) return nullptr;
    if (hipSetDevice(m->device) != hipSuccess) return nullptr;
    if (!m->d_rec_tab) {
        if (hipMalloc((void **) &m->d_rec_tab, (size_t) Impl::REC_TAB_CAP * sizeof(void *)) != hipSuccess) return nullptr;
        if (hipMemset(m->d_rec_tab, 0, (size_t) Impl::REC_TAB_CAP * sizeof(void *)) != hipSuccess) return nullptr;
    }
    if (m->rec_chunks.capacity() < (size_t) Impl::REC_TAB_CAP) m->rec_chunks.reserve((size_t) Impl::REC_TAB_CAP);
    while ((int) m->rec_chunks.size() <= chunk) {
        MpRec *c = nullptr;
        if (hipHostMalloc((void **) &c, (size_t) MP_CHUNK * sizeof(MpRec), hipHostMallocDefault) != hipSuccess) return nullptr;
        memset(c, 0, (size_t) MP_CHUNK * sizeof(MpRec));
        const MpRec *cc = c;
        // (rare: once per 4096 map points) a blocking 8-byte copy: the kernels of later calls read the entry
        if (hipMemcpy(m->d_rec_tab + m->rec_chunks.size(), &cc, sizeof(cc), hipMemcpyHostToDevice) != hipSuccess) {
            (void) hipHostFree(c);
            return nullptr;
        }
        m->rec_chunks.push_back(c);
    }
    return m->rec_chunks[(size_t) chunk];
}

// Mapper::matchToMap on the records: the job's own arrays (cells, keyframe table, row slots, local list) lie in stage_scratch() and go
// up with ONE copy; the map itself is gathered by the kernels (alva_match_to_map_records)
int HipStages::match_to_map_rec(const MatchJob &J, int *match_of_mp) {
    if (J.n_mp <= 0) return ALVA_OK;
    if (!m->med || !alva_medoid_tables(m->med) || !m->d_rec_tab) return ALVA_ERR_STATE;   // (map points exist => their tables were replayed)
    int frame_kf = -1;
    for (int i = 0; i < J.n_kf; i++)
        if (J.kf_ids[i] == J.frame_kfid) frame_kf = i;
    if (frame_kf < 0) return ALVA_ERR_ARG;
    const int n_cell = J.cell_ptr[J.grid_cells];
    const Camera &k = m->cam;
    const double calib[10] = {k.fx, k.fy, k.cx, k.cy, k.k1, k.k2, k.p1, k.p2, (double) k.width, (double) k.height};
    const uint8_t *pb = m->pin.base, *pe = pb ? pb + m->pin.cap : nullptr;
    const uint8_t *ptrs[] = {(const uint8_t *) J.cell_ptr, (const uint8_t *) J.cell_mp, (const uint8_t *) J.kf_ids, (const uint8_t *) J.kf_q,
                             (const uint8_t *) J.kf_t, (const uint8_t *) J.mp_slot, (const uint8_t *) J.local, (const uint8_t *) match_of_mp};
    const size_t lens[] = {(size_t) (J.grid_cells + 1) * 4, (size_t) n_cell * 4, (size_t) J.n_kf * 4, (size_t) J.n_kf * 32, (size_t) J.n_kf * 24,
                           (size_t) J.n_mp * 4, (size_t) J.n_local * 4, (size_t) J.n_mp * 4};
    size_t lo = (size_t) -1, hi = 0;
    for (int i = 0; i < 8; i++) {
        if (lens[i] == 0) continue;
        if (!pb || ptrs[i] < pb || ptrs[i] + lens[i] > pe) return ALVA_ERR_ARG;   // the map layer assembles the job in stage_scratch()
        if (i < 7) {
            lo = std::min(lo, (size_t) (ptrs[i] - pb));
            hi = std::max(hi, (size_t) (ptrs[i] - pb) + lens[i]);
        }
    }
    if (m->dev.cap < m->pin.cap) return ALVA_ERR_STATE;
    auto dv = [&](const void *hp) { return m->dev.base + ((const uint8_t *) hp - pb); };
    ALVA_HIP(hipMemcpyAsync(m->dev.base + lo, pb + lo, hi - lo, hipMemcpyHostToDevice, m->st));
    const int rc = alva_match_to_map_records(m->ctx, calib, J.cell_size, J.num_cells_w, J.grid_cells, (const int *) dv(J.cell_ptr), (const int *) dv(J.cell_mp),
                                             J.n_kf, (const int *) dv(J.kf_ids), (const double *) dv(J.kf_q), (const double *) dv(J.kf_t), frame_kf,
                                             J.frame_kfid, J.n_mp, (const int *) dv(J.mp_slot), (const void *const *) m->d_rec_tab,
                                             alva_medoid_tables(m->med), J.num_keypoints_3d, J.n_local, (const int *) dv(J.local), J.max_proj_err,
                                             J.dist_ratio, match_of_mp);   // (pinned, written once per entry by the last kernel: no copy back)
    if (rc) return rc;
    ALVA_HIP(alva_stream_sync(m->st));
    return ALVA_OK;
}

int HipStages::local_ba(int n_kf, double *poses7, const uint8_t *kf_const, int n_pt, const int *pt_anchor_kf, const double *pt_anchor_uv,
                        double *pt_inv_depth, int n_obs, const int *obs_kf, const int *obs_pt, const double *obs_uv, int max_iters, double *chi2,
                        uint8_t *depth_pos) {
    const Camera &k = m->cam;
    const double calib[4] = {k.fx, k.fy, k.cx, k.cy};
    double info[4];
    int ok = 0;
    // optimizer.cpp:251-262: function tolerance 1e-3, Huber on sqrt(robustCostThreshold_) -- a float in the reference (:8, :22)
    return alva_local_ba(m->ctx, n_kf, poses7, kf_const, calib, 1, n_pt, pt_anchor_kf, pt_anchor_uv, pt_inv_depth, n_obs, obs_kf, obs_pt, obs_uv,
                         max_iters, 0.001, (double) 5.9915f, chi2, depth_pos, info, &ok);
}

int HipStages::local_ba_csr(int n_kf, double *poses7, const uint8_t *kf_const, int n_pt, const int *pt_ptr, const int *pt_anchor_kf,
                            const double *pt_anchor_uv, double *pt_inv_depth, int n_obs, const int *obs_kf, const double *obs_uv, int max_iters,
                            double chi2_threshold, uint64_t *bad_bits, int *n_bad) {
    const Camera &k = m->cam;
    const double calib[4] = {k.fx, k.fy, k.cx, k.cy};
    double info[4];
    int ok = 0;
    if (n_kf > 32) {   // (the device-side pair grouping is sized for the mapper's window; larger problems take the host-structured solve)
        return Stages::local_ba_csr(n_kf, poses7, kf_const, n_pt, pt_ptr, pt_anchor_kf, pt_anchor_uv, pt_inv_depth, n_obs, obs_kf, obs_uv, max_iters,
                                    chi2_threshold, bad_bits, n_bad);
    }
    // optimizer.cpp:251-262: function tolerance 1e-3, Huber on sqrt(robustCostThreshold_) -- a float in the reference (:8, :22)
    return alva_local_ba_csr(m->ctx, n_kf, poses7, kf_const, calib, n_pt, pt_ptr, pt_anchor_kf, pt_anchor_uv, pt_inv_depth, n_obs, obs_kf, obs_uv,
                             max_iters, 0.001, (double) 5.9915f, chi2_threshold, (unsigned long long *) bad_bits, n_bad, info, &ok);
}

// f1: the map points' descriptor tables live on the device (medoid.hip); the replay is enqueued behind the keyframe's other work on the
// session's stream and nothing waits for it -- only an export does
int HipStages::medoid_replay(int n_ops, const alva_medoid::MedoidOp *ops, int n_mp, const int *mp_slot, const int *first_op, int slots) {
    ALVA_HIP(hipSetDevice(m->device));
    if (!m->med) {
        const int rc = alva_medoid_store_create(m->ctx, &m->med);
        if (rc) return rc;
    }
    return alva_medoid_replay(m->med, n_ops, ops, n_mp, mp_slot, first_op, slots);
}
int HipStages::medoid_export(int n, const int *mp_slot, uint8_t *desc32, uint8_t *valid, int *info3) {
    if (!m->med) {   // nothing was ever replayed: every table is a fresh one
        if (desc32) memset(desc32, 0, 32 * (size_t) n);
        if (valid) memset(valid, 0, (size_t) n);
        if (info3)
            for (int i = 0; i < n; i++) { info3[3 * i] = 0; info3[3 * i + 1] = -1; info3[3 * i + 2] = 0; }
        return ALVA_OK;
    }
    return alva_medoid_export(m->med, n, mp_slot, desc32, valid, info3);
}
int HipStages::pack_map_records(int n_slots, int stream_id, int capacity, uint8_t *d_out, int *count) {
    if (!m->med || !m->d_rec_tab) return ALVA_ERR_STATE;
    return alva_pack_map_records(m->med, (const void *const *) m->d_rec_tab, n_slots, stream_id, capacity, d_out, count);
}
int HipStages::medoid_dump(int mp_slot, alva_medoid::Table *out) {
    if (!m->med) return ALVA_ERR_STATE;
    return alva_medoid_dump(m->med, mp_slot, out, sizeof(*out));
}

int HipStages::find_plane(int n, const double *pts, const double *pose7_twc, int iterations, float *pose16, int *found) {
    *found = 0;
    if (n < 32) return ALVA_OK;  // system.cpp:181
    Impl::Plan p;
    const size_t a = p.add((size_t) n * 24);
    std::vector<uint8_t *> d, h;
    int rc = m->carve(p, d, h);
    if (rc) return rc;
    UP(a, pts, (size_t) n * 24);
    // the reference seeds a fresh generator from std::random_device in every iteration (system.cpp:210)
    return alva_find_plane(m->ctx, (const double *) d[a], n, pose7_twc, iterations, 1, 0u, nullptr, pose16, found);
}

}  // namespace alva_slam
