// a1 / §8(b): the reference's `System` surface (src/slam/src/system.hpp:24-38) as a C ABI.  The per-frame state machine, the map
// and its bookkeeping live in the host-side map layer (slam/: VisualFrontend / MapManager / Mapper / Optimizer behaviour,
// SURVEY.md §1 layer L2); every numeric stage it calls runs on the GPU (stages_hip.hip -> include/alvaar_hip.h).  There is no CPU
// path: without a HIP device alva_system_create fails.
#include "common.hpp"
#include "stages_hip.hpp"
#include "slam/slam.hpp"
#include "slam/inspect.hpp"
#include "slam/stage_trace.hpp"
#include "../../include/alvaar_system.h"
#include "../../include/alvaar_system_testing.h"
#include <algorithm>
#include <chrono>
#include <cmath>
#include <condition_variable>
#include <memory>
#include <mutex>
#include <thread>
#include <sys/mman.h>
#include <ucontext.h>
#include <unistd.h>
#include <vector>

using namespace alva_slam;

static thread_local char g_sys_err[256] = "";
extern "C" const char *alva_system_last_error(void) { return g_sys_err[0] ? g_sys_err : alva_last_error(); }
static int sys_fail(int rc, const char *what) {
    snprintf(g_sys_err, sizeof(g_sys_err), "%s: %s", what, alva_last_error());
    return rc;
}

// No C++ exception may cross the C ABI (the reference's contract: "no exceptions"): the map layer keeps the reference's
// .at() look-ups (std::out_of_range on an inconsistent map) and its containers can throw std::bad_alloc.  Every entry point that runs
// the map layer goes through guarded(): the error text is kept, the tracker is asked to reset (what the reference's own failure paths
// do, visual_frontend.cpp:73-92) and ALVA_ERR_STATE is returned.
template <class F>
static int guarded(alva_system *s, const char *what, F &&body);

struct alva_system {
    int device = 0;
    std::unique_ptr<HipStages> stages;
    std::unique_ptr<TraceStages> trace;  // ALVA_STAGE_TRACE=<file>: log every stage call (debugging aid, see slam/stage_trace.hpp)
    std::unique_ptr<Slam> slam;
    void *hip_stream = nullptr;   // alva_system_set_stream: the stream the next configure builds the stages on (null: a stream of its own)
    // findCameraPoseWithIMU (system.cpp:57-104)
    double imu_translation[3] = {0, 0, 0}, prev_translation[3] = {0, 0, 0};
};

template <class F>
static int guarded(alva_system *s, const char *what, F &&body) {
    try {
        return body();
    } catch (const std::exception &e) {
        snprintf(g_sys_err, sizeof(g_sys_err), "%s: %s", what, e.what());
    } catch (...) {
        snprintf(g_sys_err, sizeof(g_sys_err), "%s: unknown C++ exception", what);
    }
    if (s && s->stages) (void) s->stages->frame_done();   // no kernel keeps reading a caller-owned frame buffer behind an error return
    if (s && s->slam) s->slam->reset_requested = true;
    return ALVA_ERR_STATE;
}

extern "C" int alva_system_create(int device, alva_system **out) {
    g_sys_err[0] = 0;
    if (!out) return ALVA_ERR_ARG;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || device < 0 || device >= count) {
        snprintf(g_sys_err, sizeof(g_sys_err), "alva_system_create: no HIP device %d (this library has no CPU path)", device);
        return ALVA_ERR_HIP;
    }
    alva_system *s = new alva_system();
    s->device = device;
    *out = s;
    return ALVA_OK;
}

extern "C" void alva_system_destroy(alva_system *s) {
    if (!s) return;
    s->slam.reset();
    s->trace.reset();
    s->stages.reset();
    delete s;
}

static int configure_impl(alva_system *s, int width, int height, double fx, double fy, double cx, double cy, double k1, double k2,
                          double p1, double p2, int cell_size, int clahe_enabled, int random_sampling) {
    g_sys_err[0] = 0;
    if (!s || width < 64 || height < 64 || width % 4 || cell_size < 8 || !(fx > 0) || !(fy > 0)) {
        snprintf(g_sys_err, sizeof(g_sys_err), "alva_system_configure: bad argument");
        return ALVA_ERR_ARG;
    }
    // a failed (re-)configuration leaves the object unconfigured, never half-built
    s->slam.reset();
    s->trace.reset();
    s->stages.reset();
    Camera cam;
    cam.width = width; cam.height = height; cam.border = 20;  // system.cpp:29
    cam.fx = fx; cam.fy = fy; cam.cx = cx; cam.cy = cy; cam.k1 = k1; cam.k2 = k2; cam.p1 = p1; cam.p2 = p2;
    Settings cfg;  // system.cpp:15-19 over state.hpp:29-78
    cfg.cell_size = cell_size;
    cfg.clahe = clahe_enabled != 0;
    cfg.keyframe_filtering_ratio = 0.95f;
    cfg.p3p_enabled = true;
    cfg.random_sampling = random_sampling != 0;
    std::unique_ptr<HipStages> st(new HipStages());
    std::unique_ptr<Slam> slam(new Slam(st.get(), cam, cfg));
    const int rc = st->init(s->device, cam, cfg.clahe, slam->invK, s->hip_stream);
    if (rc) return sys_fail(rc, "alva_system_configure");
    if (!getenv("ALVA_NO_WARMUP")) {
        // code-object loads, arena growth and launch attributes belong to configure, not to the first frames and keyframes
        const int wrc = st->warm_up(cell_size);
        if (wrc) return sys_fail(wrc, "alva_system_configure (warm-up)");
    }
    s->stages = std::move(st);
    s->slam = std::move(slam);
    if (const char *path = getenv("ALVA_STAGE_TRACE")) {
        s->trace.reset(new TraceStages(s->stages.get(), path));
        s->trace->image_width_ = width;
        s->trace->image_height_ = height;
        s->slam->st = s->trace.get();
    }
    for (int i = 0; i < 3; i++) s->imu_translation[i] = s->prev_translation[i] = 0;
    return ALVA_OK;
}

extern "C" int alva_system_configure_ex(alva_system *s, int width, int height, double fx, double fy, double cx, double cy, double k1, double k2,
                                        double p1, double p2, int cell_size, int clahe_enabled, int random_sampling) {
    return guarded(s, "alva_system_configure", [&]() -> int {
        return configure_impl(s, width, height, fx, fy, cx, cy, k1, k2, p1, p2, cell_size, clahe_enabled, random_sampling);
    });
}

extern "C" int alva_system_configure(alva_system *s, int width, int height, double fx, double fy, double cx, double cy, double k1, double k2,
                                     double p1, double p2) {
    return alva_system_configure_ex(s, width, height, fx, fy, cx, cy, k1, k2, p1, p2, 40 /* system.cpp:15 */, 0 /* :17 */, 1 /* state.hpp:67 */);
}

extern "C" void alva_system_reset(alva_system *s) {  // system.cpp:42-55
    if (!s || !s->slam) return;
    guarded(s, "alva_system_reset", [&]() -> int { s->slam->reset(); return ALVA_OK; });
    for (double &v: s->prev_translation) v = 0;
}

extern "C" int alva_system_register_frame_buffer(alva_system *s, const uint8_t *h_rgba, size_t bytes) {
    g_sys_err[0] = 0;
    if (!s || !s->stages || !h_rgba) {
        snprintf(g_sys_err, sizeof(g_sys_err), "alva_system_register_frame_buffer: not configured or NULL argument");
        return ALVA_ERR_ARG;
    }
    const int rc = s->stages->register_frame_buffer(h_rgba, bytes);
    return rc ? sys_fail(rc, "alva_system_register_frame_buffer") : ALVA_OK;
}

extern "C" int alva_system_alloc_frame_buffer(alva_system *s, size_t bytes, uint8_t **h_writable) {
    g_sys_err[0] = 0;
    if (!s || !s->stages || !h_writable) {
        snprintf(g_sys_err, sizeof(g_sys_err), "alva_system_alloc_frame_buffer: not configured or NULL argument");
        return ALVA_ERR_ARG;
    }
    const int rc = s->stages->alloc_frame_buffer(bytes, h_writable);
    return rc ? sys_fail(rc, "alva_system_alloc_frame_buffer") : ALVA_OK;
}

extern "C" int alva_system_unregister_frame_buffer(alva_system *s) {
    g_sys_err[0] = 0;
    if (!s || !s->stages) return ALVA_OK;
    const int rc = s->stages->unregister_frame_buffer();
    return rc ? sys_fail(rc, "alva_system_unregister_frame_buffer") : ALVA_OK;
}

extern "C" int alva_system_find_camera_pose_ts(alva_system *s, const uint8_t *h_rgba, double timestamp, float *h_pose) {
    g_sys_err[0] = 0;
    if (!s || !s->slam || !h_rgba || !h_pose) {
        snprintf(g_sys_err, sizeof(g_sys_err), "alva_system_find_camera_pose: not configured or NULL argument");
        return ALVA_ERR_ARG;
    }
    return guarded(s, "alva_system_find_camera_pose", [&]() -> int {
        const int status = s->slam->process_frame(h_rgba, timestamp);  // system.cpp:156-175
        if (status < 0) return sys_fail(status, "alva_system_find_camera_pose");
        pose_to_array(s->slam->cur->Twc, h_pose);  // written whatever the status (system.cpp:118)
        return status;
    });
}

extern "C" int alva_system_find_camera_pose_device(alva_system *s, const uint8_t *d_rgba, double timestamp, float *h_pose) {
    g_sys_err[0] = 0;
    if (!s || !s->slam || !d_rgba || !h_pose) {
        snprintf(g_sys_err, sizeof(g_sys_err), "alva_system_find_camera_pose_device: not configured or NULL argument");
        return ALVA_ERR_ARG;
    }
    return guarded(s, "alva_system_find_camera_pose_device", [&]() -> int {
        const int status = s->slam->process_frame(d_rgba, timestamp, true);
        if (status < 0) return sys_fail(status, "alva_system_find_camera_pose_device");
        pose_to_array(s->slam->cur->Twc, h_pose);
        return status;
    });
}

extern "C" int alva_system_hint_next_frame_device(alva_system *s, const uint8_t *d_rgba_next) {
    g_sys_err[0] = 0;
    if (!s || !s->slam) {
        snprintf(g_sys_err, sizeof(g_sys_err), "alva_system_hint_next_frame_device: not configured");
        return ALVA_ERR_ARG;
    }
    s->slam->next_frame_hint = d_rgba_next;
    return ALVA_OK;
}

extern "C" int alva_system_find_camera_pose(alva_system *s, const uint8_t *h_rgba, float *h_pose) {
    // system.cpp:114: milliseconds of the system clock
    const double ts = (double) std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::system_clock::now().time_since_epoch()).count();
    return alva_system_find_camera_pose_ts(s, h_rgba, ts, h_pose);
}

extern "C" int alva_system_find_camera_pose_with_imu(alva_system *s, const uint8_t *h_rgba, const double *h_imu, float *h_pose) {
    const double ts = (double) std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::system_clock::now().time_since_epoch()).count();
    return alva_system_find_camera_pose_with_imu_ts(s, h_rgba, h_imu, ts, h_pose);   // system.cpp:87: milliseconds of the system clock
}

extern "C" int alva_system_find_camera_pose_with_imu_ts(alva_system *s, const uint8_t *h_rgba, const double *h_imu, double timestamp, float *h_pose) {
    if (!s || !s->slam || !h_imu || !h_pose) return ALVA_ERR_ARG;
    float tmp[16];
    const int status = alva_system_find_camera_pose_ts(s, h_rgba, timestamp, tmp);
    if (status < 0) return status;
    // system.cpp:66-103: orientation = inverse of the IMU quaternion (w, -x, y, z); the translation integrates the visual one
    double q[4] = {-h_imu[1], h_imu[2], h_imu[3], h_imu[0]};
    quat_normalize(q);
    double R[9];
    quat_to_rot(q, R);
    const SE3 &T = s->slam->cur->Twc;
    if (status == 1) {
        for (int c = 0; c < 3; c++) {
            s->imu_translation[c] += T.t[c] - s->prev_translation[c];
            s->prev_translation[c] = T.t[c];
        }
    } else {
        for (double &v: s->prev_translation) v = 0;
    }
    for (int r = 0; r < 3; r++) {
        for (int c = 0; c < 3; c++) h_pose[4 * r + c] = (float) R[3 * c + r];  // inverse rotation = transpose
        h_pose[4 * r + 3] = 0.f;
    }
    for (int c = 0; c < 3; c++) h_pose[12 + c] = (float) s->imu_translation[c];
    h_pose[15] = 1.f;
    return 1;
}

extern "C" int alva_system_find_plane(alva_system *s, float *h_pose, int num_iterations) {
    if (!s || !s->slam || !h_pose || num_iterations <= 0) return 0;
    const int rc = guarded(s, "alva_system_find_plane", [&]() -> int {
        // MapManager::getCurrentFrameMapPoints (map_manager.cpp:340-357): observed 3-D map points, in the map's container order
        std::vector<double> pts;
        for (const auto &e: s->slam->map_points)
            if (e.second->r->observed && e.second->r->is3d) pts.insert(pts.end(), e.second->r->X, e.second->r->X + 3);
        double pose7[7];
        se3_to_pose7(s->slam->cur->Twc, pose7);
        int found = 0;
        if (s->stages->find_plane((int) (pts.size() / 3), pts.data(), pose7, num_iterations, h_pose, &found) != ALVA_OK) return 0;
        return found ? 1 : 0;
    });
    return rc == 1 ? 1 : 0;
}

extern "C" int alva_system_get_frame_points(alva_system *s, int *h_points) {
    if (!s || !s->slam || !h_points) return 0;
    // system.cpp:139-154 as intended: the 2-D (not yet triangulated) keypoints' undistorted positions as ints, at most 2048
    int n2d = 0, written = 0;
    for (const auto &e: s->slam->cur->kps)
        if (!e.second.is3d) {
            if (written < 2048) {
                h_points[2 * written] = (int) e.second.unpx[0];
                h_points[2 * written + 1] = (int) e.second.unpx[1];
                written++;
            }
            n2d++;
        }
    return n2d;
}

extern "C" int alva_system_get_keypoints(alva_system *s, int *h_ids, float *h_px, uint8_t *h_is3d, int cap) {
    if (!s || !s->slam) return 0;
    return inspect_frame(*s->slam->cur, cap, h_ids, h_px, nullptr, h_is3d, nullptr);
}

// ---- inspection / test hooks ------------------------------------------------------------------------------------------
extern "C" int alva_system_debug_state(alva_system *s, int *out16) {
    if (!s || !s->slam || !out16) return ALVA_ERR_ARG;
    inspect_state(*s->slam, out16);
    return ALVA_OK;
}
extern "C" int alva_system_debug_pose7(alva_system *s, double *pose7, double *init_pose7) {
    if (!s || !s->slam) return ALVA_ERR_ARG;
    if (pose7) se3_to_pose7(s->slam->cur->Twc, pose7);
    if (init_pose7) se3_to_pose7(s->slam->init_computed, init_pose7);
    return ALVA_OK;
}
extern "C" int alva_system_debug_frame_keypoints(alva_system *s, int cap, int *ids, float *px, float *unpx, uint8_t *is3d, uint8_t *has_desc) {
    if (!s || !s->slam) return ALVA_ERR_ARG;
    return inspect_frame(*s->slam->cur, cap, ids, px, unpx, is3d, has_desc);
}
extern "C" int alva_system_debug_keyframe_ids(alva_system *s, int cap, int *ids) {
    if (!s || !s->slam) return ALVA_ERR_ARG;
    return inspect_keyframe_ids(*s->slam, cap, ids);
}
extern "C" int alva_system_debug_keyframe(alva_system *s, int kfid, double *pose7, int *info6, int cap, int *ids, float *px, uint8_t *is3d) {
    if (!s || !s->slam) return ALVA_ERR_ARG;
    return inspect_keyframe(*s->slam, kfid, pose7, info6, cap, ids, px, is3d);
}
extern "C" int alva_system_debug_covisibility(alva_system *s, int kfid, int cap, int *pairs) {
    if (!s || !s->slam) return ALVA_ERR_ARG;
    return inspect_covisibility(*s->slam, kfid, cap, pairs);
}
extern "C" int alva_system_debug_map_points(alva_system *s, int cap, int *ids, double *xyz, int *flags5, double *inv_depth, uint8_t *desc) {
    if (!s || !s->slam) return ALVA_ERR_ARG;
    return inspect_map_points(*s->slam, cap, ids, xyz, flags5, inv_depth, desc);
}
// ---- the shared-map merge across sessions, applied (north_star's optional extra; semantics: MapManager::mergeMapPoints, map_manager.cpp:428-513)
extern "C" int alva_system_merge_map_points(alva_system *s, int prev_id, int new_id) {
    if (!s || !s->slam) return ALVA_ERR_ARG;
    return guarded(s, "alva_system_merge_map_points", [&]() -> int {
        // MapManager::mergeMapPoints is written for the mapper's own merges -- a keypoint of the new keyframe against a local-map point
        // the frame does NOT see (mapper.cpp:354-588): a frame or keyframe that observes BOTH points cannot change one id into the other
        // (Frame::updateKeypointId fails) and would be left with a keypoint of a map point that no longer exists.  Two points of a session
        // that a shared-map round declares the same are merged only where the reference's routine is safe: never co-observed.
        Slam &S = *s->slam;
        const auto ia = S.map_points.find(prev_id), ib = S.map_points.find(new_id);
        if (ia == S.map_points.end() || ib == S.map_points.end() || prev_id == new_id) return 0;
        const MapPt *a = ia->second, *b = ib->second;
        if (S.cur->observes(prev_id) && S.cur->observes(new_id)) return 0;
        for (int kf: a->observers())
            if (b->obs_has(kf)) return 0;
        const long before = s->slam->n_merges;
        s->slam->merge_map_points(prev_id, new_id);
        return s->slam->n_merges > before ? 1 : 0;   // 0: the reference's early return (a point is gone, or the survivor is not 3-D)
    });
}
extern "C" int alva_system_pack_map_records(alva_system *s, int stream_id, int capacity, uint8_t *d_out, int *h_count) {
    if (!s || !s->slam || !d_out || !h_count || capacity < 0) return ALVA_ERR_ARG;
    return guarded(s, "alva_system_pack_map_records", [&]() -> int {
        Slam &S = *s->slam;
        S.flush_medoids();   // the medoids of this keyframe's edits
        if (S.last_error()) return sys_fail(S.last_error(), "alva_system_pack_map_records");
        const int rc = s->stages->pack_map_records(S.med_log.next_slot, stream_id, capacity, d_out, h_count);
        return rc ? sys_fail(rc, "alva_system_pack_map_records") : ALVA_OK;
    });
}
extern "C" int alva_system_set_shared_ids(alva_system *s, int n, const int *local_id, const int *shared_stream, const int *shared_id) {
    if (!s || !s->slam || n < 0 || (n > 0 && (!local_id || !shared_stream || !shared_id))) return ALVA_ERR_ARG;
    return guarded(s, "alva_system_set_shared_ids", [&]() -> int {
        int applied = 0;
        for (int i = 0; i < n; i++)
            if (s->slam->map_points.count(local_id[i])) {
                s->slam->shared_ids[local_id[i]] = {shared_stream[i], shared_id[i]};
                applied++;
            }
        return applied;
    });
}
extern "C" int alva_system_get_shared_ids(alva_system *s, int cap, int *local_id, int *shared_stream, int *shared_id) {
    if (!s || !s->slam || cap < 0) return ALVA_ERR_ARG;
    return guarded(s, "alva_system_get_shared_ids", [&]() -> int {
        std::vector<std::pair<int, std::pair<int, int>>> v;
        for (auto it = s->slam->shared_ids.begin(); it != s->slam->shared_ids.end();) {
            if (!s->slam->map_points.count(it->first)) it = s->slam->shared_ids.erase(it);   // the map point has been culled since
            else {
                v.push_back(*it);
                ++it;
            }
        }
        std::sort(v.begin(), v.end());
        for (size_t i = 0; i < v.size() && (int) i < cap; i++) {
            if (local_id) local_id[i] = v[i].first;
            if (shared_stream) shared_stream[i] = v[i].second.first;
            if (shared_id) shared_id[i] = v[i].second.second;
        }
        return (int) v.size();
    });
}

extern "C" int alva_system_debug_counters(alva_system *s, long *out3) {
    if (!s || !s->slam || !out3) return ALVA_ERR_ARG;
    out3[0] = s->slam->n_ba_runs; out3[1] = s->slam->n_merges; out3[2] = s->slam->n_kf_culled;
    return ALVA_OK;
}
extern "C" int alva_system_debug_klt_work(alva_system *s, long *out2, int reset) {
    if (!s || !s->slam || !out2) return ALVA_ERR_ARG;
    out2[0] = s->slam->n_klt_kp_levels; out2[1] = s->slam->n_klt_slots;
    if (reset) s->slam->n_klt_kp_levels = s->slam->n_klt_slots = 0;
    return ALVA_OK;
}
extern "C" int alva_system_debug_timing(alva_system *s, double *out8, int reset) {
    if (!s || !s->slam) return ALVA_ERR_ARG;
    if (out8) memcpy(out8, s->slam->t_section, sizeof(s->slam->t_section));
    if (reset) memset(s->slam->t_section, 0, sizeof(s->slam->t_section));
    return ALVA_OK;
}
extern "C" int alva_system_debug_timing_keyframe(alva_system *s, double *out16, int reset) {
    if (!s || !s->slam) return ALVA_ERR_ARG;
    if (out16) memcpy(out16, s->slam->t_kf, sizeof(s->slam->t_kf));
    if (reset) memset(s->slam->t_kf, 0, sizeof(s->slam->t_kf));
    return ALVA_OK;
}
extern "C" int alva_system_debug_timing_fine(alva_system *s, double *out32, int reset) {
    if (!s || !s->slam) return ALVA_ERR_ARG;
    if (out32) memcpy(out32, s->slam->t_fine, sizeof(s->slam->t_fine));
    if (reset) memset(s->slam->t_fine, 0, sizeof(s->slam->t_fine));
    return ALVA_OK;
}
extern "C" int alva_system_debug_set_init_pose(alva_system *s, const double *pose7) {
    if (!s || !s->slam) return ALVA_ERR_ARG;
    s->slam->init_override.armed = pose7 != nullptr;
    if (pose7) memcpy(s->slam->init_override.pose7, pose7, 7 * sizeof(double));
    return ALVA_OK;
}


// ---- a GROUP of sessions on a few host threads ------------------------------------------------------------------------------------
// One session is a chain of latency-bound kernels with the host waiting in between: a thread per session burns a core on waits (8
// sessions on 8 cores: 9.3 k frames/s, every core spinning).  A group runs its sessions as FIBERS: W worker threads, each with a share
// of the sessions; a session executes the unmodified synchronous path (alva_system_find_camera_pose_device) on its own stack, and
// every wait of that path -- the completion-word polls and the stream waits, all through alva_poll_relax / alva_stream_sync
// (common.hpp) -- switches to the thread's next session instead of spinning.  The GPU sees the sessions' streams side by side; the host
// threads only ever execute map-layer work.  Sessions stay independent: every session's results are those of its solo run, bit for bit.
namespace {
const size_t kFiberStack = [] {   // ALVA_FIBER_STACK_KB (default 4 MiB: map layer + HIP runtime + exception unwinding run on it)
    const char *e = getenv("ALVA_FIBER_STACK_KB");
    const size_t kb = e ? (size_t) strtoul(e, nullptr, 10) : 4096;
    return (kb < 256 ? 256 : kb) << 10;
}();
struct Fiber {
    ucontext_t ctx;
    char *stack = nullptr;
    void *stack_map = nullptr;
    Fiber() = default;
    Fiber(const Fiber &) = delete;
    Fiber(Fiber &&o) noexcept { *this = std::move(o); }
    Fiber &operator=(Fiber &&o) noexcept {
        memcpy(&ctx, &o.ctx, sizeof(ctx));
        stack = o.stack; stack_map = o.stack_map; sys = o.sys; d_rgba = o.d_rgba; ts = o.ts; pose = o.pose; status = o.status; done = o.done;
        lane = o.lane; lane_dirty = o.lane_dirty;
        o.stack = nullptr; o.stack_map = nullptr;
        return *this;
    }
    ~Fiber() {
        if (stack_map) munmap(stack_map, kFiberStack + (size_t) sysconf(_SC_PAGESIZE));
    }
    alva_system *sys = nullptr;
    const uint8_t *d_rgba = nullptr;
    double ts = 0;
    float *pose = nullptr;
    int *status = nullptr;
    bool done = true;
    alva_lane *lane = nullptr;   // the shared launches this session's tracking chain goes through, if any (lane.hpp)
    bool lane_dirty = false;     // chain work deposited whose completion the session's host side has not seen yet
};
struct Worker;
thread_local Worker *g_worker = nullptr;
struct Worker {
    ucontext_t sched;
    std::vector<Fiber> fibers;
    int current = -1;
    static void yield_hook() {
        Worker *w = g_worker;
        swapcontext(&w->fibers[(size_t) w->current].ctx, &w->sched);
    }
    static void entry() {
        Worker *w = g_worker;
        Fiber &f = w->fibers[(size_t) w->current];
        *f.status = alva_system_find_camera_pose_device(f.sys, f.d_rgba, f.ts, f.pose);
        f.done = true;
        swapcontext(&f.ctx, &w->sched);
    }
    double t_run = 0, t_work = 0;   // seconds inside run() | inside fiber slices that did more than poll
    long n_slices = 0, n_work_slices = 0;
    // run the first n fibers to completion, round robin over the unfinished ones
    void run(int n) {
        const auto t_run0 = std::chrono::steady_clock::now();
        g_worker = this;
        alva_fiber_yield = &Worker::yield_hook;
        for (int i = 0; i < n; i++) {
            Fiber &f = fibers[(size_t) i];
            if (!f.stack) {   // own mapping with a PROT_NONE guard page below it: an overflow faults instead of corrupting the heap
                const size_t page = (size_t) sysconf(_SC_PAGESIZE);
                void *p = mmap(nullptr, kFiberStack + page, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_STACK, -1, 0);
                if (p == MAP_FAILED) {
                    *f.status = ALVA_ERR_STATE;
                    f.done = true;
                    continue;
                }
                (void) mprotect(p, page, PROT_NONE);
                f.stack = (char *) p + page;
                f.stack_map = p;
            }
            getcontext(&f.ctx);
            f.ctx.uc_stack.ss_sp = f.stack;
            f.ctx.uc_stack.ss_size = kFiberStack;
            f.ctx.uc_link = &sched;
            makecontext(&f.ctx, (void (*)()) & Worker::entry, 0);
            f.done = false;
        }
        int left = 0;
        for (int i = 0; i < n; i++) {
            Fiber &f = fibers[(size_t) i];
            if (f.done) {   // never started (no stack): it still counts as one of its lane's sessions of this step
                if (f.lane) alva_lane_session_done(f.lane);
            } else left++;
        }
        while (left > 0)
            for (int i = 0; i < n; i++) {
                Fiber &f = fibers[(size_t) i];
                if (f.done) continue;
                current = i;
                // HIP's current device is per THREAD and the fibers share this one: whatever the previous fiber set must not leak into
                // this one's allocations and launches after a yield
                if (f.sys) (void) hipSetDevice(f.sys->device);   // (a NULL session entry is answered with ALVA_ERR_ARG by the fiber's body)
                if (f.lane) alva_lane_tick(f.lane);
                g_alva_lane = f.lane;
                g_alva_lane_dirty = &f.lane_dirty;
                const auto t_in = std::chrono::steady_clock::now();
                swapcontext(&sched, &f.ctx);
                const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_in).count();
                n_slices++;
                if (dt > 1.5e-6) {   // (a slice that only looked at a completion word and yielded again is a poll, not work)
                    t_work += dt;
                    n_work_slices++;
                }
                g_alva_lane = nullptr;
                g_alva_lane_dirty = nullptr;
                if (f.done) {
                    left--;
                    if (f.lane) alva_lane_session_done(f.lane);
                }
            }
        alva_fiber_yield = nullptr;
        g_worker = nullptr;
        current = -1;
        t_run += std::chrono::duration<double>(std::chrono::steady_clock::now() - t_run0).count();
    }
};
}  // namespace

struct alva_system_group {
    std::vector<std::thread> threads;
    std::vector<Worker> workers;
    std::vector<int> share;          // sessions of worker w in this call
    std::vector<hipStream_t> streams;   // alva_system_group_stream: streams that several sessions share (destroyed with the group)
    std::vector<alva_lane *> lanes;     // shared launches (lane.hpp): lane k = a stream of its own + the sessions i with (i / workers) % lanes == k
    std::vector<hipStream_t> lane_streams;
    int n_lanes = 4;                    // ALVA_GROUP_LANES / alva_system_group_set_lanes; 0: every session launches for itself
    bool lockstep = true;
    int stream_device = 0;
    std::mutex mu;
    std::condition_variable cv_go, cv_done;
    long generation = 0;
    int pending = 0;
    bool quit = false;
    void loop(int w) {
        long seen = 0;
        for (;;) {
            {
                std::unique_lock<std::mutex> lk(mu);
                cv_go.wait(lk, [&] { return quit || generation != seen; });
                if (quit) return;
                seen = generation;
            }
            if (share[(size_t) w] > 0) workers[(size_t) w].run(share[(size_t) w]);
            {
                std::lock_guard<std::mutex> lk(mu);
                if (--pending == 0) cv_done.notify_all();
            }
        }
    }
};

extern "C" int alva_system_set_stream(alva_system *s, void *hip_stream) {
    if (!s) return ALVA_ERR_ARG;
    s->hip_stream = hip_stream;
    return ALVA_OK;
}

extern "C" int alva_system_group_stream(alva_system_group *g, int device, int index, void **out_stream) {
    if (!g || !out_stream || index < 0) return ALVA_ERR_ARG;
    std::lock_guard<std::mutex> lk(g->mu);
    if ((int) g->streams.size() <= index) g->streams.resize((size_t) index + 1, nullptr);
    if (!g->streams[(size_t) index]) {
        if (hipSetDevice(device) != hipSuccess) return ALVA_ERR_HIP;
        hipStream_t st = nullptr;
        if (hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess) return ALVA_ERR_HIP;
        g->streams[(size_t) index] = st;
        g->stream_device = device;
    }
    *out_stream = g->streams[(size_t) index];
    return ALVA_OK;
}

extern "C" int alva_system_group_set_lockstep(alva_system_group *g, int on) {
    if (!g) return ALVA_ERR_ARG;
    std::lock_guard<std::mutex> lk(g->mu);   // (the workers are parked between two group calls)
    g->lockstep = on != 0;
    return ALVA_OK;
}

extern "C" int alva_system_group_time_stats(alva_system_group *g, double *out4, int reset) {
    if (!g || !out4) return ALVA_ERR_ARG;
    std::lock_guard<std::mutex> lk(g->mu);
    out4[0] = out4[1] = out4[2] = out4[3] = 0;
    for (Worker &w: g->workers) {
        out4[0] += w.t_run;
        out4[1] += w.t_work;
        out4[2] += (double) w.n_slices;
        out4[3] += (double) w.n_work_slices;
        if (reset) {
            w.t_run = w.t_work = 0;
            w.n_slices = w.n_work_slices = 0;
        }
    }
    return ALVA_OK;
}

extern "C" int alva_system_group_set_lanes(alva_system_group *g, int n_lanes) {
    if (!g || n_lanes < 0 || n_lanes > 64) return ALVA_ERR_ARG;
    std::lock_guard<std::mutex> lk(g->mu);
    g->n_lanes = n_lanes;
    return ALVA_OK;
}

extern "C" int alva_system_group_launch_stats(alva_system_group *g, long *out2) {
    if (!g || !out2) return ALVA_ERR_ARG;
    std::lock_guard<std::mutex> lk(g->mu);
    out2[0] = out2[1] = 0;
    for (alva_lane *l: g->lanes)
        if (l) {
            long a = 0, b = 0;
            alva_lane_stats(l, &a, &b);
            out2[0] += a;
            out2[1] += b;
        }
    return ALVA_OK;
}

extern "C" int alva_system_group_create(int n_threads, alva_system_group **out) {
    if (!out || n_threads < 1 || n_threads > 256) return ALVA_ERR_ARG;
    alva_system_group *g = new alva_system_group();
    g->workers.resize((size_t) n_threads);
    {
        const char *e = getenv("ALVA_GROUP_LOCKSTEP");   // default on; 0 = every session launches for itself (round 3's behaviour)
        g->lockstep = !e || atoi(e) != 0;
        if (const char *l = getenv("ALVA_GROUP_LANES")) g->n_lanes = atoi(l) < 0 ? 0 : (atoi(l) > 64 ? 64 : atoi(l));
    }
    g->share.assign((size_t) n_threads, 0);
    for (int w = 0; w < n_threads; w++) g->threads.emplace_back([g, w] { g->loop(w); });
    *out = g;
    return ALVA_OK;
}

extern "C" void alva_system_group_destroy(alva_system_group *g) {
    if (!g) return;
    {
        std::lock_guard<std::mutex> lk(g->mu);
        g->quit = true;
    }
    g->cv_go.notify_all();
    for (std::thread &t: g->threads) t.join();
    for (alva_lane *l: g->lanes)
        if (l) alva_lane_destroy(l);
    for (hipStream_t st: g->lane_streams)
        if (st) (void) hipStreamDestroy(st);
    if (!g->streams.empty()) (void) hipSetDevice(g->stream_device);
    for (hipStream_t st: g->streams)
        if (st) (void) hipStreamDestroy(st);
    delete g;
}

extern "C" int alva_system_group_find_camera_pose_device(alva_system_group *g, int count, alva_system *const *systems, const uint8_t *const *d_rgba,
                                                         double timestamp_ms, float *h_poses, int *h_status) {
    if (!g || count < 0 || (count > 0 && (!systems || !d_rgba || !h_poses || !h_status))) return ALVA_ERR_ARG;
    if (count == 0) return ALVA_OK;
    const int W = (int) g->workers.size();
    {
        std::lock_guard<std::mutex> lk(g->mu);
        for (int w = 0; w < W; w++) g->share[(size_t) w] = 0;
        for (int i = 0; i < count; i++) {   // session i -> worker i % W, always the same worker for the same i (its stacks stay warm)
            Worker &wk = g->workers[(size_t) (i % W)];
            const int k = g->share[(size_t) (i % W)]++;
            if ((int) wk.fibers.size() <= k) wk.fibers.resize((size_t) k + 1);
            Fiber &f = wk.fibers[(size_t) k];
            f.sys = systems[i];
            f.lane = nullptr;
            f.lane_dirty = false;
            if (g->lockstep && g->n_lanes > 0 && systems[i]) {
                // lane (i + i / W) % lanes: a worker's consecutive sessions (i, i + W, ...) sit on DIFFERENT lanes, so that it does the
                // host half of one lane's sessions while another lane's launches run, and every lane draws its sessions from all workers.  A lane's stream is created on the device of its first session; sessions of another
                // device launch for themselves.
                const size_t k = (size_t) ((i + i / W) % g->n_lanes);
                if (g->lanes.size() <= k) {
                    g->lanes.resize(k + 1, nullptr);
                    g->lane_streams.resize(k + 1, nullptr);
                }
                if (!g->lanes[k]) {
                    hipStream_t st = nullptr;
                    if (hipSetDevice(systems[i]->device) == hipSuccess && hipStreamCreateWithFlags(&st, hipStreamNonBlocking) == hipSuccess) {
                        g->lane_streams[k] = st;
                        g->lanes[k] = alva_lane_create(systems[i]->device, st);
                    }
                }
                if (g->lanes[k] && alva_lane_device(g->lanes[k]) == systems[i]->device) f.lane = g->lanes[k];
            }
            f.d_rgba = d_rgba[i];
            f.ts = timestamp_ms;
            f.pose = h_poses + 16 * (size_t) i;
            f.status = h_status + i;
        }
        // every lane learns how many of this step's sessions run on its stream: a kind that all of them have deposited goes out at once
        for (size_t k = 0; k < g->lanes.size(); k++) {
            if (!g->lanes[k]) continue;
            int on_lane = 0;
            for (int w = 0; w < W; w++)
                for (int j = 0; j < g->share[(size_t) w]; j++) on_lane += g->workers[(size_t) w].fibers[(size_t) j].lane == g->lanes[k] ? 1 : 0;
            alva_lane_begin_step(g->lanes[k], on_lane);
        }
        g->pending = W;
        g->generation++;
    }
    g->cv_go.notify_all();
    std::unique_lock<std::mutex> lk(g->mu);
    g->cv_done.wait(lk, [&] { return g->pending == 0; });
    for (alva_lane *l: g->lanes)
        if (l) alva_lane_flush_now(l);   // (nothing is left when every session waited for its frame; a frame that tracks nothing may leave its images)
    return ALVA_OK;
}
