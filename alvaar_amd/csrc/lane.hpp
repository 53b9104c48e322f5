// Shared launches for the sessions of an alva_system_group (a "lane" = a HIP stream of the group + the sessions assigned to it, whichever
// worker threads run them as fibers).  A session keeps its OWN stream for everything else (keyframe stages, local BA, ...).
//
// A session's tracking frame is a chain of seven launches (gray + level 0, pyramid, slot table, tracker, compaction, P3P, PnP); S sessions
// side by side issue 7 S launches per group step and the GPU's command path -- not its compute units -- is what saturates (round 3: 8 / 32
// sessions 9.9 / 10.9 k frames/s, empty launches of 8 concurrent streams 26 us each).  Inside a group's worker thread a chain launch is
// therefore DEPOSITED (kernel kind + its argument block) instead of issued, and the lane issues ONE launch per kind for all of its
// sessions: a `*_multi` kernel whose blockIdx.y selects the session's argument block out of a device table (one upload per flush for
// all kinds) and whose body is the single-session kernel's body, unchanged -- results are those of the solo run bit for bit.
//
// Rules (lane.hip):
//  * the chain has two segments whose kinds a session deposits back to back -- images + slot table + tracker + compaction, and
//    P3P + PnP; the deposit of a segment's LAST kind that makes its count == the lane's unfinished sessions of this group step flushes
//    every kind up to it at once (the normal case: all sessions of the lane reach the same point of the same frame), on the depositing
//    thread: one table upload, then the launches behind each other;
//  * anything else is flushed once no deposit has arrived for ALVA_LANE_FLUSH_US (default 250 us; checked at every fiber switch): a
//    partial set costs a launch of its own, serialised in the lane's stream in front of the rest -- and every kernel of the chain takes
//    its latency whatever it carries -- so stragglers are worth waiting for;
//  * kinds are issued in chain order; a session that deposits a kind <= one it still has pending flushes first;
//  * the chain reads nothing that the session's own stream produces (frame in HBM, pinned slot table, the chain's own outputs), and the
//    host has seen the chain's completion words before it enqueues anything that reads the chain's results -- so the two streams need no
//    events in the steady state; the exceptions go through the dirty flag below.
// A worker runs sessions of SEVERAL lanes (session i: worker i % W; its lane is its stream): while one lane's launches run on the GPU the
// worker does the host half of its sessions on the other lanes -- software pipelining across lanes, with no cross-stream dependency.
#pragma once
#include <hip/hip_runtime.h>
#include <cstddef>
#include <cstdint>

enum AlvaMultiKind {
    MK_LEVEL0 = 0,      // k_level0<true>      image.hip
    MK_PYR_REST,        // k_pyr_rest          image.hip
    MK_STAGE_IN,        // k_track_stage_in    klt.hip
    MK_TRACK_KLT,       // k_track_klt         klt.hip
    MK_TRACK_COMPACT,   // k_track_compact     stages_hip.hip
    MK_P3P,             // k_p3p               p3p.hip
    MK_PNP,             // k_pnp               pnp.hip
    MK_COUNT
};

struct AlvaMultiKindInfo {
    const char *name = nullptr;
    size_t arg_bytes = 0;
    // issue ONE launch over `count` argument blocks (device table d_args, arg_bytes each; d_gx: their grid widths)
    void (*launch)(hipStream_t st, const uint8_t *d_args, const unsigned *d_gx, int count, unsigned gmax, unsigned smax) = nullptr;
};
int alva_multi_register(int kind, const char *name, size_t arg_bytes, void (*launch)(hipStream_t, const uint8_t *, const unsigned *, int, unsigned, unsigned));

struct alva_lane;
// the lane of the session that is running on this thread right now (set by the group's scheduler at every fiber switch); null outside
// a group's worker thread and for a session that is not on one of the group's streams
extern thread_local alva_lane *g_alva_lane;

alva_lane *alva_lane_create(int device, hipStream_t stream);
void alva_lane_destroy(alva_lane *l);
hipStream_t alva_lane_stream(const alva_lane *l);
int alva_lane_device(const alva_lane *l);
// the group, around one group step: n sessions of this step run on the lane's stream (from any of the worker threads)
void alva_lane_begin_step(alva_lane *l, int n_sessions);
void alva_lane_session_done(alva_lane *l);
void alva_lane_tick(alva_lane *l);        // a worker, at every fiber switch: the idle rule
void alva_lane_flush_now(alva_lane *l);   // everything pending, now
void alva_lane_stats(alva_lane *l, long *launches, long *entries);

// true: deposited (the lane issues it ON THE LANE'S STREAM); false: the caller launches directly, on its own stream
// (ctx = the session's context: the unit whose launches must stay in program order)
bool alva_lane_defer_slow(int kind, const void *owner, unsigned gx, unsigned shmem, const void *args, size_t bytes);
#define alva_lane_defer(kind, ctx, gx, shmem, args, bytes) (g_alva_lane ? alva_lane_defer_slow((kind), (ctx), (gx), (shmem), (args), (bytes)) : false)
// Ordering between the lane's stream and the session's own: the scheduler points g_alva_lane_dirty at the running session's flag; a
// deposit sets it, alva_lane_clean() clears it where the host has SEEN the chain's last kernel complete (a polled completion word), and
// any direct stream operation or wait of a session whose flag is set first issues + waits for the lane (alva_lane_flush).
extern thread_local bool *g_alva_lane_dirty;
void alva_lane_flush_slow();
static inline void alva_lane_flush() {
    if (g_alva_lane) alva_lane_flush_slow();
}
void alva_lane_clean();
// after the last deposit of a chain whose results the session will poll for: let the thread's other sessions make theirs
void alva_lane_yield();
