// Shared launches for the sessions of an alva_system_group: see lane.hpp.
#include "common.hpp"
#include "lane.hpp"
#include <atomic>
#include <cstdlib>
#include <ctime>
#include <mutex>
#include <vector>

thread_local alva_lane *g_alva_lane = nullptr;
thread_local bool *g_alva_lane_dirty = nullptr;

namespace {
AlvaMultiKindInfo g_kinds[MK_COUNT];

inline long long now_ns() {
    timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (long long) ts.tv_sec * 1000000000ll + ts.tv_nsec;
}
// the launches a lane issues run on the issuing thread, through the same launch macro as everything else: its alva_lane_flush must not
// come back into a lane (the issuing thread holds that lane's mutex, and taking a second lane's would order the mutexes both ways)
thread_local bool t_issuing = false;
}  // namespace

int alva_multi_register(int kind, const char *name, size_t arg_bytes, void (*launch)(hipStream_t, const uint8_t *, const unsigned *, int, unsigned, unsigned)) {
    if (kind >= 0 && kind < MK_COUNT) {
        g_kinds[kind].name = name;
        g_kinds[kind].arg_bytes = arg_bytes;
        g_kinds[kind].launch = launch;
    }
    return kind;
}

// One lane = one HIP stream of a group + the deposits of the sessions that run on it, from whichever worker threads those sessions are
// fibers of.  Everything but n_pending is touched under `mu`; a launch is issued by the thread whose deposit (or tick) completes the set.
struct alva_lane {
    struct Pending {
        std::vector<uint8_t> args;
        std::vector<unsigned> gx, shmem;
        std::vector<const void *> owner;
        int count = 0;
    } kind[MK_COUNT];
    std::mutex mu;
    hipStream_t stream = nullptr;
    int device = 0;
    std::atomic<int> n_pending{0};
    long long last_ns = 0;          // the most recent deposit
    long long max_idle_ns = 250000;
    int n_unfinished = 0;
    long launches = 0, entries = 0;
    // argument tables: a ring of (pinned staging, device) buffer pairs; a flush fills the next pair, copies it up in ONE command on the
    // lane's stream and launches against the device copy; the pair is reused once the event behind its last launch has passed
    static constexpr int NSLOT = 8;
    static constexpr size_t SLOT_BYTES = 512 << 10;
    struct Slot {
        uint8_t *host = nullptr, *dev = nullptr;
        hipEvent_t done = nullptr;
        bool busy = false;
    } slot[NSLOT];
    int next_slot = 0;

    bool slot_ready(Slot &s) {
        if (!s.host) {   // all or nothing: a partial allocation is released, so that a later flush starts from scratch
            uint8_t *h = nullptr, *d = nullptr;
            hipEvent_t e = nullptr;
            if (hipHostMalloc((void **) &h, SLOT_BYTES, hipHostMallocDefault) != hipSuccess || hipMalloc((void **) &d, SLOT_BYTES) != hipSuccess ||
                hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) {
                if (h) (void) hipHostFree(h);
                if (d) (void) hipFree(d);
                return false;
            }
            s.host = h; s.dev = d; s.done = e;
        }
        if (s.busy) {
            (void) hipEventSynchronize(s.done);   // (eight flushes back: long over)
            s.busy = false;
        }
        return true;
    }
    // every kind <= upto, in chain order (mu held): one table upload, one launch per kind
    void flush_upto(int upto) {
        if (n_pending.load(std::memory_order_relaxed) == 0) return;
        t_issuing = true;
        int cur = device;
        (void) hipGetDevice(&cur);   // a group may hold sessions of several devices: the lane's stream belongs to `device`
        if (cur != device) (void) hipSetDevice(device);
        Slot &s = slot[next_slot];
        next_slot = (next_slot + 1) % NSLOT;
        size_t off[MK_COUNT], off_gx[MK_COUNT], total = 0;
        for (int k = 0; k <= upto; k++) {
            const Pending &p = kind[k];
            if (!p.count) continue;
            off_gx[k] = total;
            total += ((size_t) p.count * 4 + 15) / 16 * 16;
            off[k] = total;
            total += (p.args.size() + 15) / 16 * 16;
        }
        if (total > 0 && total <= SLOT_BYTES && slot_ready(s)) {
            for (int k = 0; k <= upto; k++) {
                const Pending &p = kind[k];
                if (!p.count) continue;
                memcpy(s.host + off_gx[k], p.gx.data(), (size_t) p.count * 4);
                memcpy(s.host + off[k], p.args.data(), p.args.size());
            }
            (void) (hipMemcpyAsync)(s.dev, s.host, total, hipMemcpyHostToDevice, stream);
            for (int k = 0; k <= upto; k++) {
                Pending &p = kind[k];
                if (!p.count) continue;
                unsigned gmax = 0, smax = 0;
                for (int i = 0; i < p.count; i++) {
                    gmax = p.gx[(size_t) i] > gmax ? p.gx[(size_t) i] : gmax;
                    smax = p.shmem[(size_t) i] > smax ? p.shmem[(size_t) i] : smax;
                }
                g_kinds[k].launch(stream, s.dev + off[k], (const unsigned *) (s.dev + off_gx[k]), p.count, gmax, smax);
                launches++;
                entries += p.count;
            }
            (void) (hipEventRecord)(s.done, stream);
            s.busy = true;
        } else if (total > 0) {
            // the staging slots cannot carry this flush (tables larger than a slot, or the slot's allocation failed): a one-off device
            // table filled by a synchronous copy -- slow, but every deposit is launched and its owner's completion words arrive
            uint8_t *tmp = nullptr;
            std::vector<uint8_t> img(total);
            for (int k = 0; k <= upto; k++) {
                const Pending &p = kind[k];
                if (!p.count) continue;
                memcpy(img.data() + off_gx[k], p.gx.data(), (size_t) p.count * 4);
                memcpy(img.data() + off[k], p.args.data(), p.args.size());
            }
            if (hipMalloc((void **) &tmp, total) == hipSuccess && (hipMemcpy)(tmp, img.data(), total, hipMemcpyHostToDevice) == hipSuccess) {
                for (int k = 0; k <= upto; k++) {
                    Pending &p = kind[k];
                    if (!p.count) continue;
                    unsigned gmax = 0, smax = 0;
                    for (int i = 0; i < p.count; i++) {
                        gmax = p.gx[(size_t) i] > gmax ? p.gx[(size_t) i] : gmax;
                        smax = p.shmem[(size_t) i] > smax ? p.shmem[(size_t) i] : smax;
                    }
                    g_kinds[k].launch(stream, tmp + off[k], (const unsigned *) (tmp + off_gx[k]), p.count, gmax, smax);
                    launches++;
                    entries += p.count;
                }
                (void) (hipStreamSynchronize)(stream);
            } else {
                alva_set_error("lane: argument tables of %zu bytes could not be staged", total);   // (the sessions' completion polls time out and report)
            }
            if (tmp) (void) hipFree(tmp);
        }
        for (int k = 0; k <= upto; k++) {
            Pending &p = kind[k];
            n_pending.fetch_sub(p.count, std::memory_order_relaxed);
            p.count = 0;
            p.args.clear();
            p.gx.clear();
            p.shmem.clear();
            p.owner.clear();
        }
        if (cur != device) (void) hipSetDevice(cur);
        t_issuing = false;
    }
    // kinds that every unfinished session has deposited (mu held)
    void flush_complete() {
        if (n_pending.load(std::memory_order_relaxed) == 0) return;
        for (int k = MK_COUNT - 1; k >= 0; k--)
            if ((k == MK_TRACK_COMPACT || k == MK_PNP) && kind[k].count > 0 && kind[k].count >= n_unfinished) {
                flush_upto(k);
                return;
            }
        if (n_unfinished == 0) flush_upto(MK_COUNT - 1);
    }
};

alva_lane *alva_lane_create(int device, hipStream_t stream) {
    alva_lane *l = new alva_lane();
    l->device = device;
    l->stream = stream;
    if (const char *e = getenv("ALVA_LANE_FLUSH_US")) l->max_idle_ns = (long long) (atof(e) * 1000.0);
    return l;
}
void alva_lane_destroy(alva_lane *l) {
    if (!l) return;
    (void) hipSetDevice(l->device);
    for (alva_lane::Slot &s: l->slot) {
        if (s.busy) (void) hipEventSynchronize(s.done);
        if (s.done) (void) hipEventDestroy(s.done);
        if (s.dev) (void) hipFree(s.dev);
        if (s.host) (void) hipHostFree(s.host);
    }
    delete l;
}
hipStream_t alva_lane_stream(const alva_lane *l) { return l->stream; }
int alva_lane_device(const alva_lane *l) { return l->device; }

void alva_lane_begin_step(alva_lane *l, int n_sessions) {
    std::lock_guard<std::mutex> lk(l->mu);
    l->n_unfinished = n_sessions;
}
void alva_lane_session_done(alva_lane *l) {
    std::lock_guard<std::mutex> lk(l->mu);
    if (l->n_unfinished > 0) l->n_unfinished--;
    l->flush_complete();
}
void alva_lane_tick(alva_lane *l) {
    // the IDLE rule: sessions that are going to deposit do so within a few microseconds of each other (each runs until its next wait);
    // once nothing has arrived for max_idle_ns, whoever is missing is not coming (it tracks nothing this frame, or it is elsewhere in its frame)
    if (l->n_pending.load(std::memory_order_relaxed) == 0) return;
    std::unique_lock<std::mutex> lk(l->mu, std::try_to_lock);
    if (!lk.owns_lock()) return;
    if (l->n_pending.load(std::memory_order_relaxed) > 0 && now_ns() - l->last_ns > l->max_idle_ns) l->flush_upto(MK_COUNT - 1);
}
void alva_lane_flush_now(alva_lane *l) {
    if (l->n_pending.load(std::memory_order_relaxed) == 0) return;
    std::lock_guard<std::mutex> lk(l->mu);
    l->flush_upto(MK_COUNT - 1);
}
void alva_lane_stats(alva_lane *l, long *launches, long *entries) {
    std::lock_guard<std::mutex> lk(l->mu);
    *launches = l->launches;
    *entries = l->entries;
}

bool alva_lane_defer_slow(int kind, const void *owner, unsigned gx, unsigned shmem, const void *args, size_t bytes) {
    alva_lane *l = g_alva_lane;
    if (!l || t_issuing || kind < 0 || kind >= MK_COUNT || !g_kinds[kind].launch || g_kinds[kind].arg_bytes != bytes) return false;
    if (g_alva_lane_dirty) *g_alva_lane_dirty = true;
    std::lock_guard<std::mutex> lk(l->mu);
    // chain order per session: a kind <= one this session still has pending means a new chain of it -- the old one goes out first
    for (int k = kind; k < MK_COUNT && l->n_pending.load(std::memory_order_relaxed) > 0; k++) {
        bool hit = false;
        for (const void *o: l->kind[k].owner) hit |= o == owner;
        if (hit) {
            l->flush_upto(MK_COUNT - 1);
            break;
        }
    }
    alva_lane::Pending &p = l->kind[kind];
    const uint8_t *a = (const uint8_t *) args;
    p.args.insert(p.args.end(), a, a + bytes);
    p.gx.push_back(gx);
    p.shmem.push_back(shmem);
    p.owner.push_back(owner);
    p.count++;
    l->n_pending.fetch_add(1, std::memory_order_relaxed);
    l->last_ns = now_ns();
    // a chain SEGMENT goes out as one: the kinds of a segment are deposited back to back (images; slot table -> tracker -> compaction;
    // P3P -> PnP), so only a segment's LAST kind triggers -- one table upload and the launches behind each other, instead of an upload
    // and a launch per kind with the depositing threads' scheduling in between (measured: 70 us bubbles between 13 - 80 us kernels)
    static const bool by_segment = !(getenv("ALVA_LANE_SEGMENTS") && atoi(getenv("ALVA_LANE_SEGMENTS")) == 0);   // (A/B: 0 = every kind triggers)
    const bool segment_end = !by_segment || kind == MK_TRACK_COMPACT || kind == MK_PNP;
    if (segment_end && p.count >= l->n_unfinished) l->flush_upto(kind);
    return true;
}

// A DIRECT operation of a session that has chain work on the lane which its host side has not seen complete: the lane's stream and the
// session's own stream are not ordered against each other, so everything the session deposited goes out and is waited for first.  Rare by
// construction -- the chain ends in completion words the host polls (compaction, PnP), and alva_lane_clean() marks those points; what is
// left are frames that build their images and then track nothing (initialisation), and the fall-back paths of the tracking step.
void alva_lane_flush_slow() {
    alva_lane *l = g_alva_lane;
    if (!l || t_issuing || !g_alva_lane_dirty || !*g_alva_lane_dirty) return;
    *g_alva_lane_dirty = false;
    alva_lane_flush_now(l);
    hipEvent_t ev = nullptr;
    if (hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) {
        (void) (hipStreamSynchronize)(l->stream);
        return;
    }
    (void) (hipEventRecord)(ev, l->stream);
    while (hipEventQuery(ev) == hipErrorNotReady) {
        if (alva_fiber_yield) alva_fiber_yield();
        else __builtin_ia32_pause();
    }
    (void) hipEventDestroy(ev);
}

void alva_lane_clean() {
    if (g_alva_lane_dirty) *g_alva_lane_dirty = false;
}

void alva_lane_yield() {
    if (g_alva_lane && g_alva_lane->n_pending.load(std::memory_order_relaxed) > 0 && alva_fiber_yield) alva_fiber_yield();
}
