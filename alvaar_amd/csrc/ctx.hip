// Context, error reporting, scratch management.
#include "common.hpp"

thread_local void (*alva_fiber_yield)(void) = nullptr;

static thread_local char g_err[512] = "";

void alva_set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char *alva_last_error(void) { return g_err; }
extern "C" const char *alva_version(void) { return "alvaar_hip 0.1 (gfx950)"; }

// ---- per-kernel event timing ---------------------------------------------------------------------------------
#include <map>
#include <mutex>
#include <string>
int g_alva_prof_on = 0;
namespace {
struct ProfRec {
    const char *name;
    hipEvent_t e0, e1;
};
std::mutex g_prof_mu;
std::vector<ProfRec> g_prof_recs;
std::vector<hipEvent_t> g_prof_pool;
std::map<std::string, std::pair<long, double>> g_prof_acc;  // kernel -> (launches, total ms)

hipEvent_t prof_event() {
    if (!g_prof_pool.empty()) {
        hipEvent_t e = g_prof_pool.back();
        g_prof_pool.pop_back();
        return e;
    }
    hipEvent_t e = nullptr;
    (void) hipEventCreate(&e);
    return e;
}

// waits for every recorded pair and folds it into the per-kernel totals (caller holds the mutex)
void prof_drain() {
    for (ProfRec &r: g_prof_recs) {
        if (!r.e0 || !r.e1) continue;
        float ms = 0.f;
        if (alva_event_sync(r.e1) == hipSuccess && hipEventElapsedTime(&ms, r.e0, r.e1) == hipSuccess) {
            std::string key(r.name);
            if (!key.empty() && key.front() == '(' && key.back() == ')') key = key.substr(1, key.size() - 2);  // (k<a, b>) -> k<a, b>
            auto &a = g_prof_acc[key];
            a.first++;
            a.second += ms;
        }
        g_prof_pool.push_back(r.e0);
        g_prof_pool.push_back(r.e1);
    }
    g_prof_recs.clear();
}
}  // namespace

void alva_prof_mark(hipStream_t stream, const char *kernel, int end) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    if (!end) {
        if (g_prof_recs.size() >= 8192) prof_drain();
        ProfRec r{kernel, prof_event(), nullptr};
        if (r.e0) (void) hipEventRecord(r.e0, stream);
        g_prof_recs.push_back(r);
    } else {
        // the matching begin is the last record of this kernel name without an end (launches of one host thread nest trivially)
        for (size_t i = g_prof_recs.size(); i-- > 0;)
            if (g_prof_recs[i].name == kernel && !g_prof_recs[i].e1) {
                g_prof_recs[i].e1 = prof_event();
                if (g_prof_recs[i].e1) (void) hipEventRecord(g_prof_recs[i].e1, stream);
                break;
            }
    }
}

extern "C" int alva_prof_enable(int on) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    if (on && !g_alva_prof_on) {
        prof_drain();
        g_prof_acc.clear();
    }
    g_alva_prof_on = on ? 1 : 0;
    return ALVA_OK;
}

extern "C" int alva_prof_report(char *buf, size_t cap) {
    ALVA_ARG(buf && cap > 0);
    std::lock_guard<std::mutex> lk(g_prof_mu);
    prof_drain();
    std::string out;
    char line[256];
    for (auto &kv: g_prof_acc) {
        snprintf(line, sizeof(line), "%s\t%ld\t%.6f\n", kv.first.c_str(), kv.second.first, kv.second.second * 1e3 / (double) kv.second.first);
        out += line;
    }
    if (out.size() + 1 > cap) {
        alva_set_error("alva_prof_report: buffer too small (%zu needed)", out.size() + 1);
        return ALVA_ERR_ARG;
    }
    memcpy(buf, out.c_str(), out.size() + 1);
    return ALVA_OK;
}

static int ctx_create(int device, void *hip_stream, int own_stream, int priority_class, alva_ctx **out) {
    ALVA_ARG(out != nullptr);
    int ndev = 0;
    ALVA_HIP(hipGetDeviceCount(&ndev));
    ALVA_ARG(device >= 0 && device < ndev);
    ALVA_HIP(hipSetDevice(device));
    alva_ctx *c = new alva_ctx();
    c->device = device;
    if (!own_stream) {
        c->stream = (hipStream_t) hip_stream;  // NULL = legacy default stream
    } else {
        int least = 0, greatest = 0;  // numerically: greatest priority <= least priority
        (void) hipDeviceGetStreamPriorityRange(&least, &greatest);
        const int prio = priority_class < 0 ? greatest : (priority_class > 0 ? least : (least + greatest) / 2);
        hipError_t e = hipStreamCreateWithPriority(&c->stream, hipStreamNonBlocking, prio);
        if (e != hipSuccess) {
            delete c;
            alva_set_error("hipStreamCreate: %s", hipGetErrorString(e));
            return ALVA_ERR_HIP;
        }
        c->owns_stream = true;
    }
    c->pinned_bytes = 1 << 16;
    hipError_t e = hipHostMalloc(&c->pinned, c->pinned_bytes, hipHostMallocDefault);
    if (e != hipSuccess) {
        if (c->owns_stream) (void) hipStreamDestroy(c->stream);
        delete c;
        alva_set_error("hipHostMalloc: %s", hipGetErrorString(e));
        return ALVA_ERR_NOMEM;
    }
    // 64 arrival counters + 8 KB behind them: the relay block of the fused pose launch (pnp.hip k_pose_all: d_counters + 64)
    e = hipMalloc((void **) &c->d_counters, 64 * sizeof(int) + 8192);
    if (e == hipSuccess) e = hipMemset(c->d_counters, 0, 64 * sizeof(int) + 8192);
    if (e != hipSuccess) {
        (void) hipHostFree(c->pinned);
        if (c->owns_stream) (void) hipStreamDestroy(c->stream);
        delete c;
        alva_set_error("hipMalloc(counters): %s", hipGetErrorString(e));
        return ALVA_ERR_NOMEM;
    }
    *out = c;
    return ALVA_OK;
}

extern "C" int alva_ctx_create(int device, void *hip_stream, int own_stream, alva_ctx **out) {
    return ctx_create(device, hip_stream, own_stream, 0, out);
}

// Own non-blocking stream in one of the device's priority classes (-1 high, 0 normal, +1 low).  Besides the scheduling hint this
// decides the hardware queue: the HIP runtime multiplexes the streams of one priority class onto a small pool of hardware
// queues (GPU_MAX_HW_QUEUES, default 4), and two streams that share a queue execute strictly one after the other.
extern "C" int alva_ctx_create_with_priority(int device, int priority_class, alva_ctx **out) {
    return ctx_create(device, nullptr, 1, priority_class, out);
}

extern "C" void alva_ctx_destroy(alva_ctx *ctx) {
    if (!ctx) return;
    (void) hipSetDevice(ctx->device);
    (void) alva_stream_sync(ctx->stream);
    for (auto &s: ctx->scratch)
        if (s.ptr) (void) hipFree(s.ptr);
    if (ctx->pinned) (void) hipHostFree(ctx->pinned);
    if (ctx->d_counters) (void) hipFree(ctx->d_counters);
    if (ctx->fence) (void) hipEventDestroy(ctx->fence);
    if (ctx->pose_pending && ctx->pose_pending_free) ctx->pose_pending_free(ctx->pose_pending);
    if (ctx->owns_stream) (void) hipStreamDestroy(ctx->stream);
    delete ctx;
}

extern "C" int alva_ctx_sync(alva_ctx *ctx) {
    ALVA_ARG(ctx != nullptr);
    ALVA_HIP(alva_stream_sync(ctx->stream));
    return ALVA_OK;
}

// Work enqueued on `ctx` after this call starts only after everything enqueued so far on `producer` has finished.
// No host synchronisation.  Lets one frame run independent stages (detection vs tracking + pose) on two HIP streams.
extern "C" int alva_ctx_wait(alva_ctx *ctx, alva_ctx *producer) {
    ALVA_ARG(ctx && producer && ctx->device == producer->device);
    if (ctx->stream == producer->stream) return ALVA_OK;
    if (!producer->fence) ALVA_HIP(hipEventCreateWithFlags(&producer->fence, hipEventDisableTiming));
    ALVA_HIP(hipEventRecord(producer->fence, producer->stream));
    ALVA_HIP(hipStreamWaitEvent(ctx->stream, producer->fence, 0));
    return ALVA_OK;
}

extern "C" void *alva_ctx_stream(alva_ctx *ctx) { return ctx ? (void *) ctx->stream : nullptr; }

int alva_ctx_pinned(alva_ctx *ctx, size_t bytes, void **out) {
    if (ctx->pinned_bytes < bytes) {
        ALVA_HIP(alva_stream_sync(ctx->stream));
        if (ctx->pinned) ALVA_HIP(hipHostFree(ctx->pinned));
        ctx->pinned = nullptr;
        ctx->pinned_bytes = 0;
        const size_t want = bytes + bytes / 2;
        hipError_t e = hipHostMalloc(&ctx->pinned, want, hipHostMallocDefault);
        if (e != hipSuccess) {
            alva_set_error("hipHostMalloc(%zu): %s", want, hipGetErrorString(e));
            return ALVA_ERR_NOMEM;
        }
        ctx->pinned_bytes = want;
    }
    *out = ctx->pinned;
    return ALVA_OK;
}

int alva_ctx_scratch(alva_ctx *ctx, int slot, size_t bytes, void **out) {
    ALVA_ARG(slot >= 0 && slot < 12);
    alva_scratch &s = ctx->scratch[slot];
    if (s.bytes < bytes) {
        // growing frees the old block: wait for work that may still read it
        ALVA_HIP(alva_stream_sync(ctx->stream));
        if (s.ptr) ALVA_HIP(hipFree(s.ptr));
        s.ptr = nullptr;
        s.bytes = 0;
        size_t want = bytes + bytes / 2 + 4096;
        hipError_t e = hipMalloc(&s.ptr, want);
        if (e != hipSuccess) {
            alva_set_error("hipMalloc(%zu): %s", want, hipGetErrorString(e));
            return ALVA_ERR_NOMEM;
        }
        s.bytes = want;
    }
    *out = s.ptr;
    return ALVA_OK;
}
