// Internal definitions shared by the HIP translation units of libalvaar_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>
#include "../../include/alvaar_hip.h"
#include "lane.hpp"

void alva_set_error(const char *fmt, ...);

// Optional per-kernel timing (alva_prof_enable / alva_prof_report): every launch of this library goes through the
// macro below; with profiling on it is bracketed by two HIP events recorded on the launch stream.  Off = one load.
extern int g_alva_prof_on;
void alva_prof_mark(hipStream_t stream, const char *kernel, int end);
#undef hipLaunchKernelGGL
#define hipLaunchKernelGGL(kernelName, numBlocks, numThreads, memPerBlock, streamId, ...)            \
    do {                                                                                             \
        alva_lane_flush(); /* deposits of this thread's lane go out before any direct launch (lane.hpp) */ \
        if (g_alva_prof_on) alva_prof_mark((streamId), #kernelName, 0);                              \
        (kernelName)<<<(numBlocks), (numThreads), (memPerBlock), (streamId)>>>(__VA_ARGS__);         \
        if (g_alva_prof_on) alva_prof_mark((streamId), #kernelName, 1);                              \
    } while (0)

// every other stream operation of this library: the thread's lane (lane.hpp) issues its deposits first, so that stream order == program
// order for a session whether or not it runs inside a group (a function-like macro is not re-expanded inside its own replacement)
#define hipMemcpyAsync(...) (alva_lane_flush(), hipMemcpyAsync(__VA_ARGS__))
#define hipMemsetAsync(...) (alva_lane_flush(), hipMemsetAsync(__VA_ARGS__))
#define hipEventRecord(...) (alva_lane_flush(), hipEventRecord(__VA_ARGS__))
#define hipStreamWaitEvent(...) (alva_lane_flush(), hipStreamWaitEvent(__VA_ARGS__))
#define hipStreamSynchronize(...) (alva_lane_flush(), hipStreamSynchronize(__VA_ARGS__))
#define hipDeviceSynchronize() (alva_lane_flush(), hipDeviceSynchronize())

#define ALVA_HIP(expr)                                                                         \
    do {                                                                                       \
        hipError_t e__ = (expr);                                                               \
        if (e__ != hipSuccess) {                                                               \
            alva_set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #expr, hipGetErrorString(e__)); \
            return ALVA_ERR_HIP;                                                               \
        }                                                                                      \
    } while (0)

#define ALVA_ARG(cond)                                                            \
    do {                                                                          \
        if (!(cond)) {                                                            \
            alva_set_error("%s:%d: bad argument: %s", __FILE__, __LINE__, #cond); \
            return ALVA_ERR_ARG;                                                  \
        }                                                                         \
    } while (0)

#define ALVA_LAUNCH_CHECK() ALVA_HIP(hipGetLastError())

// Growable device scratch owned by the context (no hipMalloc on the per-frame path once warm).
struct alva_scratch {
    void *ptr = nullptr;
    size_t bytes = 0;
};

struct alva_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    bool owns_stream = false;
    alva_scratch scratch[12];
    void *pinned = nullptr;  // small pinned host staging (counters, results)
    size_t pinned_bytes = 0;
    int *d_counters = nullptr;  // 64 device ints, zero between launches (inter-workgroup arrival counters)
    hipEvent_t fence = nullptr;  // lazily created; alva_ctx_wait records it on this context's stream
    void *pose_pending = nullptr;  // alva_compute_pose_enqueue -> _collect hand-over (pnp.hip)
    void (*pose_pending_free)(void *) = nullptr;
    // alva_p3p_enqueue's last launch went to the group's lane (deposited) rather than to this context's stream: the refinement that
    // consumes its output must travel the same way, or the two would run on unordered streams (pnp.hip pose_launch)
    bool p3p_deferred = false;
};

int alva_ctx_scratch(alva_ctx *ctx, int slot, size_t bytes, void **out);
// Pinned, device-visible host staging of at least `bytes` (grown on demand; growing waits for the stream).  Kernels read
// small inputs from it and write small results into it directly, so a call needs no copy commands, only the final
// stream synchronisation.
int alva_ctx_pinned(alva_ctx *ctx, size_t bytes, void **out);

static inline int alva_divup(int a, int b) { return (a + b - 1) / b; }

// One step of a host-side wait on a completion word in pinned memory: a pause.  ALVA_POLL_YIELD_AFTER=<n> makes a waiter give its core
// away (sched_yield) after n pauses -- for a process that runs more sessions than it has cores (every session is a host thread with its own
// waits; measured on 8 cores: 16 sessions 11.6 k frames/s with n = 512 against 9.3 k at 8 sessions).  Off by default: with sessions <=
// cores a yielded thread comes back late (4 sessions: 8.3 k frames/s spinning, 6.0 k yielding).
#include <sched.h>
#include <cstdlib>
// Several sessions on ONE host thread (alva_system_group, system.hip): every session runs in a fiber, and a wait hands the thread to the
// next session instead of spinning -- the hook below is that switch (null outside a group's worker thread).
extern thread_local void (*alva_fiber_yield)(void);
static inline void alva_poll_relax(unsigned spins) {
    if (alva_fiber_yield) {
        alva_fiber_yield();
        return;
    }
    static const unsigned yield_after = [] {
        const char *e = getenv("ALVA_POLL_YIELD_AFTER");
        return e ? (unsigned) strtoul(e, nullptr, 10) : 0xffffffffu;
    }();
    if (spins < yield_after) __builtin_ia32_pause();
    else sched_yield();
}
// hipStreamSynchronize for the session paths: inside a group's fiber the wait is a query loop that lets the thread's other sessions run
static inline hipError_t alva_stream_sync(hipStream_t st) {
    alva_lane_flush();   // a deposit not yet issued would make an idle stream look finished
    if (!alva_fiber_yield) return hipStreamSynchronize(st);
    for (;;) {
        const hipError_t e = hipStreamQuery(st);
        if (e != hipErrorNotReady) return e;
        alva_fiber_yield();
    }
}
static inline hipError_t alva_event_sync(hipEvent_t ev) {
    alva_lane_flush();
    if (!alva_fiber_yield) return hipEventSynchronize(ev);
    for (;;) {
        const hipError_t e = hipEventQuery(ev);
        if (e != hipErrorNotReady) return e;
        alva_fiber_yield();
    }
}

// Batched launches (B cameras, `per_cam` workgroups each).  Workgroups are dealt to the 8 XCDs round-robin by linear workgroup id and
// every XCD has its own L2: with the camera in blockIdx.z a camera's consecutive tiles land on eight different L2s, each of which
// fetches the shared halo rows / overlapping patches again (profiles/r02c_pmc_traffic_frame_step64.json: 2.7x the algorithmic bytes).
// A 1-D grid of 8 * ceil(B / 8) * per_cam workgroups in which workgroup L serves camera 8 * (L / 8 / per_cam) + L % 8 keeps a camera
// on ONE XCD, its work items consecutive in time.  With fewer than 8 cameras that would leave XCDs idle: the plain order is used.
struct AlvaXcdItem {
    int cam, item;
};
static inline unsigned alva_xcd_grid(int count, int per_cam) { return count >= 8 ? 8u * (unsigned) ((count + 7) / 8) * (unsigned) per_cam : (unsigned) count * (unsigned) per_cam; }
#if defined(__HIPCC__)
__device__ __forceinline__ AlvaXcdItem alva_xcd_item(int count, int per_cam) {
    const int L = (int) blockIdx.x;
    if (count < 8) return AlvaXcdItem{L / per_cam, L % per_cam};
    const int j = L >> 3;
    return AlvaXcdItem{(j / per_cam) * 8 + (L & 7), j % per_cam};
}
#endif

struct alva_level {
    int w = 0, h = 0;
    uint8_t *gray_base = nullptr;   // allocation base
    uint8_t *gray = nullptr;        // interior (0,0)
    size_t gray_pitch = 0;
    int16_t *deriv_base = nullptr;
    int16_t *deriv = nullptr;       // interior (0,0)
    size_t deriv_pitch = 0;         // bytes
};

struct alva_pyramid {
    int device = 0;
    int win = 0;
    int nlevels = 0;
    alva_level lv[8];
};

// Debug: in-kernel phase stamps (wall_clock64, 100 MHz) of the pose kernels.  Null unless the process runs with ALVA_KSTAMPS=1; then
// a 4096-entry device buffer that k_p3p (entries 0..2047: 8 per workgroup) and k_pnp (2048..) write and alva_debug_kstamps reads.
unsigned long long *alva_kstamp_buffer();
unsigned long long *alva_klt_stamp_buffer();   // null unless ALVA_KLT_STAMPS=1: per-slot wall time + verdict of the tracker launch (microbench.hip)

