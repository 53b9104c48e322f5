// Per-frame driver of the tracking front-end, host side in C++ over the stage seam.
//
// Mirrors the per-frame order of VisualFrontend::trackMono (src/slam/src/visual_frontend.cpp:83-150):
//   preprocessImage   gray + LK pyramid of the new frame              (:672-698)  -> alva_pyramid_build_from_rgba
//   kltTracking       forward-backward KLT of the tracked keypoints   (:152-243)  -> alva_fbklt_track
//   computePose       P3P-LMedS -> robust PnP                         (:245-417)  -> alva_compute_pose_{enqueue,collect}
// and the keyframe branch's feature work (MapManager::extractKeypoints, map_manager.cpp:196-231, with the
// north_star-named detector; descriptor matching as in MapPoint/Mapper, map_point.cpp:106-212):
//   detect + describe cv::ORB::detectAndCompute                                   -> alva_orb_detect_and_compute
//   match             BFMatcher(NORM_HAMMING) against the previous frame's set    -> alva_bf_match_hamming
//
// The reference runs these one after another on one CPU thread.  Here the detector needs nothing but the gray
// image, so it runs on a second HIP stream (lane B) while lane A tracks and solves the pose; the only host waits are
// the two results the caller needs on the host (pose, keypoint count).  All buffers are owned by the object and
// stay resident; the frame and the correspondences are device pointers supplied by the caller.
// When the caller already holds the NEXT frame (a ring of decoded frames, a file), alva_frontend_track_ahead builds that
// frame's gray image + pyramid on a third stream (lane C) while this frame is tracked: preprocessImage depends on nothing but
// the pixels, so it leaves the tracker's dependent chain (pyramid -> KLT -> P3P -> PnP) one frame early.
#include "common.hpp"
#include <cstdlib>

int alva_fbklt_track_to(alva_ctx *ctx, const alva_pyramid *prev, const alva_pyramid *curr, int num_levels, float err_thresh, float fb_dist,
                        int max_iters, float eps, const float *d_pts, const float *d_prior_in, float *d_out, uint8_t *d_status, int n);

int alva_bf_match_hamming_devcount(alva_ctx *ctx, const uint8_t *d_query, const int *d_n_query, int cap_query, const uint8_t *d_train,
                                   const int *d_n_train, int cap_train, int *d_idx, int *d_dist);

struct alva_frontend {
    int device = 0, width = 0, height = 0, n_track = 0, cap = 0;
    alva_ctx *A = nullptr, *B = nullptr, *C = nullptr;
    alva_pyramid *pyr[3] = {nullptr, nullptr, nullptr};  // ring: previous, current, next (look-ahead)
    alva_orb *orb = nullptr;
    uint8_t *d_gray[2] = {nullptr, nullptr};
    const uint8_t *prebuilt = nullptr;  // frame whose gray + pyramid lane C has already been asked to build for the next call
    size_t prebuilt_pitch = 0;
    hipEvent_t prebuilt_done = nullptr;  // recorded on lane C behind that build
    float *d_kp[2] = {nullptr, nullptr};
    uint8_t *d_desc[2] = {nullptr, nullptr};
    float *d_prior = nullptr;
    uint8_t *d_status = nullptr;
    int *d_match = nullptr;  // idx | dist
    int *d_counts = nullptr; // [2] descriptor count of each buffer, device copy (the matcher reads them before the host does)
    int n_desc[2] = {0, 0};
    long frame = 0;
    int klt_levels = 3;  // state.hpp:54 kltPyramidLevels_
};

extern "C" void alva_frontend_destroy(alva_frontend *fe) {
    if (!fe) return;
    (void) hipSetDevice(fe->device);
    if (fe->A) (void) alva_ctx_sync(fe->A);
    if (fe->B) (void) alva_ctx_sync(fe->B);
    if (fe->C) (void) alva_ctx_sync(fe->C);
    if (fe->prebuilt_done) (void) hipEventDestroy(fe->prebuilt_done);
    if (fe->orb) alva_orb_destroy(fe->orb);
    for (auto p: fe->pyr)
        if (p) alva_pyramid_destroy(p);
    void *bufs[] = {fe->d_gray[0], fe->d_gray[1], fe->d_kp[0], fe->d_kp[1], fe->d_desc[0], fe->d_desc[1], fe->d_prior, fe->d_status, fe->d_match, fe->d_counts};
    for (void *b: bufs)
        if (b) (void) hipFree(b);
    if (fe->C) alva_ctx_destroy(fe->C);
    if (fe->B) alva_ctx_destroy(fe->B);
    if (fe->A) alva_ctx_destroy(fe->A);
    delete fe;
}

extern "C" int alva_frontend_create(int device, int width, int height, int max_tracked, int orb_features, alva_frontend **out) {
    ALVA_ARG(out && width >= 64 && height >= 64 && width % 4 == 0 && max_tracked > 0 && orb_features > 0);
    alva_frontend *fe = new alva_frontend();
    fe->device = device;
    fe->width = width;
    fe->height = height;
    fe->n_track = max_tracked;
    fe->cap = 4 * orb_features + 1024;
        // One priority class for the three lanes: measured on MI355X, putting the tracker lane in the high class and the look-ahead
    // lane in the low class (alva_ctx_create_with_priority) gained nothing for one camera and cost up to half of the aggregate
    // rate with 16 cameras.  What matters is that the lanes do not SHARE a hardware queue: GPU_MAX_HW_QUEUES >= the number of
    // streams in the process.
    int rc = alva_ctx_create(device, nullptr, 1, &fe->A);
    if (!rc) rc = alva_ctx_create(device, nullptr, 1, &fe->B);
    if (!rc) rc = alva_ctx_create(device, nullptr, 1, &fe->C);
    for (int k = 0; k < 3 && !rc; k++) rc = alva_pyramid_create(fe->A, width, height, 9, 3, &fe->pyr[k]);  // state.hpp:53-54
    if (!rc) rc = alva_orb_create(fe->B, width, height, orb_features, 1.2f, 8, 20, &fe->orb);
    auto dev_alloc = [&](void **p, size_t bytes) {
        if (rc) return;
        if (hipMalloc(p, bytes) != hipSuccess) {
            alva_set_error("alva_frontend_create: hipMalloc(%zu) failed", bytes);
            rc = ALVA_ERR_NOMEM;
        }
    };
    for (int k = 0; k < 2; k++) {
        dev_alloc((void **) &fe->d_gray[k], (size_t) width * height);
        dev_alloc((void **) &fe->d_kp[k], (size_t) fe->cap * 6 * sizeof(float));
        dev_alloc((void **) &fe->d_desc[k], (size_t) fe->cap * 32);
    }
    dev_alloc((void **) &fe->d_prior, (size_t) max_tracked * 2 * sizeof(float));
    dev_alloc((void **) &fe->d_status, (size_t) max_tracked);
    dev_alloc((void **) &fe->d_match, (size_t) fe->cap * 2 * sizeof(int));
    dev_alloc((void **) &fe->d_counts, 2 * sizeof(int));
    if (rc) {
        alva_frontend_destroy(fe);
        return rc;
    }
    *out = fe;
    return ALVA_OK;
}

#include <chrono>
#include <cstdlib>
namespace {
struct FeTiming {  // ALVA_FE_TIMING=1: host-side timeline of the driver call, averaged, printed every 200 calls
    bool on = std::getenv("ALVA_FE_TIMING") != nullptr;
    double acc[6] = {0, 0, 0, 0, 0, 0};
    long n = 0;
    std::chrono::steady_clock::time_point t0;
    void start() { if (on) t0 = std::chrono::steady_clock::now(); }
    void mark(int k) {
        if (on) acc[k] += (double) std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count() * 1e-3;
    }
    void end() {
        if (!on || ++n % 200) return;
        fprintf(stderr, "[alva_frontend] us since call start: preprocess/waits issued %.1f | A enqueued %.1f | B enqueued %.1f | C enqueued %.1f | pose on host %.1f | "
                        "keypoint count on host %.1f\n", acc[5] / 200, acc[0] / 200, acc[1] / 200, acc[2] / 200, acc[3] / 200, acc[4] / 200);
        for (double &a: acc) a = 0;
    }
};
thread_local FeTiming g_fe_timing;
}  // namespace

extern "C" int alva_frontend_track_ahead(alva_frontend *fe, const uint8_t *d_rgba, size_t rgba_pitch, const uint8_t *d_rgba_next,
                                         const float *d_pts, int n_pts, const double *d_bearings, const double *d_uv,
                                         const double *d_wpts, int n_corr, float fx, float fy, float cx, float cy, double *h_pose7,
                                         int *h_pose_status, int *h_n_keypoints) {
    ALVA_ARG(fe && d_rgba && h_pose7 && h_pose_status && h_n_keypoints && n_pts >= 0 && n_pts <= fe->n_track && n_corr >= 0);
    ALVA_HIP(hipSetDevice(fe->device));
    const int cur = (int) (fe->frame % 3), prv = (int) ((fe->frame + 2) % 3), nxt = (int) ((fe->frame + 1) % 3);
    const int g = (int) (fe->frame & 1), dc = g, dp = g ^ 1;
    int rc;
    g_fe_timing.start();
    if (fe->prebuilt == d_rgba && fe->prebuilt_pitch == rgba_pitch) {
        // preprocessImage of this frame was enqueued on lane C during the previous call; normally it finished long ago, and then
        // the lanes need no barrier packets in front of their first kernels
        if (hipEventQuery(fe->prebuilt_done) != hipSuccess) {
            ALVA_HIP(hipStreamWaitEvent(fe->A->stream, fe->prebuilt_done, 0));
            ALVA_HIP(hipStreamWaitEvent(fe->B->stream, fe->prebuilt_done, 0));
        }
    } else {
        // a look-ahead build for ANOTHER frame may still be writing this pyramid / gray image on lane C: order lane A behind it
        if (fe->prebuilt && fe->prebuilt_done && hipEventQuery(fe->prebuilt_done) != hipSuccess)
            ALVA_HIP(hipStreamWaitEvent(fe->A->stream, fe->prebuilt_done, 0));
        // lane A: preprocessImage
        rc = alva_pyramid_build_from_rgba(fe->A, fe->pyr[cur], d_rgba, rgba_pitch, fe->d_gray[g], (size_t) fe->width);
        if (rc) return rc;
        rc = alva_ctx_wait(fe->B, fe->A);  // lane B may start as soon as the gray image exists
        if (rc) return rc;
    }
    fe->prebuilt = nullptr;
    g_fe_timing.mark(5);
    // lane A: kltTracking (prior = previous positions, feature_tracker.cpp:5-111) -- from the second frame on
    if (fe->frame > 0 && n_pts > 0) {
        ALVA_ARG(d_pts);
        rc = alva_fbklt_track_to(fe->A, fe->pyr[prv], fe->pyr[cur], fe->klt_levels, 30.f, 0.5f, 30, 0.01f, d_pts, d_pts, fe->d_prior,
                                 fe->d_status, n_pts);  // state.hpp:55-59; prior = the previous positions, result in d_prior
        if (rc) return rc;
    }
    // lane A: computePose (state.hpp:64-76: 100 LMedS iterations, 3 px, chi2 5.9915, 5 LM iterations)
    rc = alva_compute_pose_enqueue(fe->A, d_bearings, d_uv, d_wpts, n_corr, 100, 3.0f, 0, 12345u, 5, 5.9915f, fx, fy, cx, cy);
    if (rc) return rc;
    g_fe_timing.mark(0);
    // lane B: detector + descriptors of this frame
    rc = alva_orb_detect_and_compute(fe->B, fe->orb, fe->d_gray[g], (size_t) fe->width, fe->d_kp[dc], fe->d_desc[dc], fe->cap, nullptr);
    if (rc) return rc;
    // lane B: match against the previous frame's descriptors, enqueued right behind the detector with both counts still on
    // the device (nothing of it is left on the host's path between two frames)
    ALVA_HIP(hipMemcpyAsync(fe->d_counts + dc, alva_orb_device_count(fe->orb), sizeof(int), hipMemcpyDeviceToDevice, fe->B->stream));
    if (fe->frame > 0 && fe->n_desc[dp] > 0) {
        rc = alva_bf_match_hamming_devcount(fe->B, fe->d_desc[dc], fe->d_counts + dc, fe->cap, fe->d_desc[dp], fe->d_counts + dp,
                                            fe->n_desc[dp], fe->d_match, fe->d_match + fe->cap);
        if (rc) return rc;
    }
    g_fe_timing.mark(1);
    // lane C: preprocessImage of the NEXT frame.  Its targets are free: pyr[nxt] was the previous call's "previous" pyramid
    // (that KLT finished before the previous call returned the pose) and d_gray[g ^ 1] was the previous call's detector input
    // (lane B was drained when the previous call fetched the keypoint count).
    if (d_rgba_next) {
        rc = alva_pyramid_build_from_rgba(fe->C, fe->pyr[nxt], d_rgba_next, rgba_pitch, fe->d_gray[g ^ 1], (size_t) fe->width);
        if (rc) return rc;
        if (!fe->prebuilt_done) ALVA_HIP(hipEventCreateWithFlags(&fe->prebuilt_done, hipEventDisableTiming));
        ALVA_HIP(hipEventRecord(fe->prebuilt_done, fe->C->stream));
        fe->prebuilt = d_rgba_next;
        fe->prebuilt_pitch = rgba_pitch;
    }
    g_fe_timing.mark(2);
    // host results.  Lane B (about twenty short commands) finishes before lane A's chain does: fetching its count first lets the
    // runtime retire those commands while lane A is still computing, instead of after the pose has arrived.
    int nkp = 0;
    rc = alva_orb_collect(fe->B, fe->orb, &nkp);
    if (rc) return rc;
    g_fe_timing.mark(4);
    rc = alva_compute_pose_collect(fe->A, h_pose7, nullptr, nullptr, h_pose_status);
    if (rc) return rc;
    g_fe_timing.mark(3);
    g_fe_timing.end();
    nkp = nkp < fe->cap ? nkp : fe->cap;
    fe->n_desc[dc] = nkp;
    *h_n_keypoints = nkp;
    fe->frame++;
    return ALVA_OK;
}

extern "C" int alva_frontend_track(alva_frontend *fe, const uint8_t *d_rgba, size_t rgba_pitch, const float *d_pts, int n_pts,
                                   const double *d_bearings, const double *d_uv, const double *d_wpts, int n_corr, float fx, float fy,
                                   float cx, float cy, double *h_pose7, int *h_pose_status, int *h_n_keypoints) {
    return alva_frontend_track_ahead(fe, d_rgba, rgba_pitch, nullptr, d_pts, n_pts, d_bearings, d_uv, d_wpts, n_corr, fx, fy, cx, cy, h_pose7,
                                     h_pose_status, h_n_keypoints);
}

// Device-resident results of the last alva_frontend_track (valid until the next call; lane B may still be
// writing the matches: alva_frontend_sync() first if the host or another stream is going to read them).
extern "C" int alva_frontend_results(alva_frontend *fe, const float **d_tracked, const uint8_t **d_track_status, const float **d_keypoints,
                                     const uint8_t **d_descriptors, const int **d_match_idx, const int **d_match_dist) {
    ALVA_ARG(fe && fe->frame > 0);
    const int last = (int) ((fe->frame - 1) & 1);
    if (d_tracked) *d_tracked = fe->d_prior;
    if (d_track_status) *d_track_status = fe->d_status;
    if (d_keypoints) *d_keypoints = fe->d_kp[last];
    if (d_descriptors) *d_descriptors = fe->d_desc[last];
    if (d_match_idx) *d_match_idx = fe->d_match;
    if (d_match_dist) *d_match_dist = fe->d_match + fe->cap;
    return ALVA_OK;
}

extern "C" int alva_frontend_sync(alva_frontend *fe) {
    ALVA_ARG(fe);
    int rc = alva_ctx_sync(fe->A);
    if (!rc) rc = alva_ctx_sync(fe->B);
    if (!rc) rc = alva_ctx_sync(fe->C);
    return rc;
}

// ---- many independent camera streams on one GPU (measurement helper) -------------------------------------------
// One host thread per stream, each driving its own alva_frontend (two HIP streams each).  A single 640x480 stream keeps
// the GPU a few percent busy (every stage is a short dependent chain); independent streams fill the remaining CUs.
// d_frames[s * ring + k] is frame k of stream s; the other inputs are per stream.  *h_seconds = wall time of
// `steps` frames per stream, started together.
#include <atomic>
#include <chrono>
#include <thread>

extern "C" int alva_frontend_run_many(alva_frontend **fes, int n_streams, int steps, int warmup, const uint8_t *const *d_frames, int ring,
                                      size_t rgba_pitch, const float *const *d_pts, int n_pts, const double *const *d_bearings,
                                      const double *const *d_uv, const double *const *d_wpts, int n_corr, float fx, float fy, float cx,
                                      float cy, double *h_seconds, int *h_accepted) {
    ALVA_ARG(fes && n_streams > 0 && steps > 0 && warmup >= 0 && d_frames && ring > 0 && d_pts && d_bearings && d_uv && d_wpts && h_seconds);
    std::atomic<int> ready{0}, go{0}, failed{0}, accepted{0};
    // look-ahead preprocessing pays for ONE stream (it shortens that stream's dependent chain); with several independent streams the
    // other streams already fill the idle CUs and a third HIP stream per camera only adds queue pressure
    const char *la = std::getenv("ALVA_FE_LOOKAHEAD");
    const bool lookahead = la ? atoi(la) != 0 : n_streams == 1;
    std::vector<std::thread> th;
    std::vector<std::chrono::steady_clock::time_point> t_end((size_t) n_streams);
    auto body = [&](int s) {
        double pose[7];
        int st = 0, nkp = 0, ok = 0;
        auto one = [&](int k) {
            const int rc = alva_frontend_track_ahead(fes[s], d_frames[(size_t) s * ring + (size_t) (k % ring)], rgba_pitch,
                                                     lookahead && ring > 1 ? d_frames[(size_t) s * ring + (size_t) ((k + 1) % ring)] : nullptr, d_pts[s], n_pts,
                                                     d_bearings[s], d_uv[s], d_wpts[s], n_corr, fx, fy, cx, cy, pose, &st, &nkp);
            if (rc) failed.store(1);
            ok += st == 2;
        };
        for (int k = 0; k < warmup; k++) one(k);
        (void) alva_frontend_sync(fes[s]);
        ok = 0;
        ready.fetch_add(1);
        while (!go.load(std::memory_order_acquire)) std::this_thread::yield();
        for (int k = 0; k < steps; k++) one(warmup + k);
        (void) alva_frontend_sync(fes[s]);
        t_end[(size_t) s] = std::chrono::steady_clock::now();
        accepted.fetch_add(ok);
    };
    for (int s = 0; s < n_streams; s++) th.emplace_back(body, s);
    while (ready.load() < n_streams) std::this_thread::yield();
    const auto t0 = std::chrono::steady_clock::now();
    go.store(1, std::memory_order_release);
    for (auto &t: th) t.join();
    auto t1 = t0;
    for (auto &e: t_end) t1 = e > t1 ? e : t1;
    *h_seconds = (double) std::chrono::duration_cast<std::chrono::nanoseconds>(t1 - t0).count() * 1e-9;
    if (h_accepted) *h_accepted = accepted.load();
    if (failed.load()) {
        alva_set_error("alva_frontend_run_many: a stream reported an error");
        return ALVA_ERR_STATE;
    }
    return ALVA_OK;
}
