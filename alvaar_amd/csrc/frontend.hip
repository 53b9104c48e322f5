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
#include "common.hpp"

int alva_fbklt_track_to(alva_ctx *ctx, const alva_pyramid *prev, const alva_pyramid *curr, int num_levels, float err_thresh, float fb_dist,
                        int max_iters, float eps, const float *d_pts, const float *d_prior_in, float *d_out, uint8_t *d_status, int n);

int alva_bf_match_hamming_devcount(alva_ctx *ctx, const uint8_t *d_query, const int *d_n_query, int cap_query, const uint8_t *d_train,
                                   const int *d_n_train, int cap_train, int *d_idx, int *d_dist);
const int *alva_orb_device_count(const alva_orb *orb);

struct alva_frontend {
    int device = 0, width = 0, height = 0, n_track = 0, cap = 0;
    alva_ctx *A = nullptr, *B = nullptr;
    alva_pyramid *pyr[2] = {nullptr, nullptr};
    alva_orb *orb = nullptr;
    uint8_t *d_gray = nullptr;
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
    if (fe->orb) alva_orb_destroy(fe->orb);
    for (auto p: fe->pyr)
        if (p) alva_pyramid_destroy(p);
    void *bufs[] = {fe->d_gray, fe->d_kp[0], fe->d_kp[1], fe->d_desc[0], fe->d_desc[1], fe->d_prior, fe->d_status, fe->d_match, fe->d_counts};
    for (void *b: bufs)
        if (b) (void) hipFree(b);
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
    int rc = alva_ctx_create(device, nullptr, 1, &fe->A);
    if (!rc) rc = alva_ctx_create(device, nullptr, 1, &fe->B);
    for (int k = 0; k < 2 && !rc; k++) rc = alva_pyramid_create(fe->A, width, height, 9, 3, &fe->pyr[k]);  // state.hpp:53-54
    if (!rc) rc = alva_orb_create(fe->B, width, height, orb_features, 1.2f, 8, 20, &fe->orb);
    auto dev_alloc = [&](void **p, size_t bytes) {
        if (rc) return;
        if (hipMalloc(p, bytes) != hipSuccess) {
            alva_set_error("alva_frontend_create: hipMalloc(%zu) failed", bytes);
            rc = ALVA_ERR_NOMEM;
        }
    };
    dev_alloc((void **) &fe->d_gray, (size_t) width * height);
    for (int k = 0; k < 2; k++) {
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

extern "C" int alva_frontend_track(alva_frontend *fe, const uint8_t *d_rgba, size_t rgba_pitch, const float *d_pts, int n_pts,
                                   const double *d_bearings, const double *d_uv, const double *d_wpts, int n_corr, float fx, float fy,
                                   float cx, float cy, double *h_pose7, int *h_pose_status, int *h_n_keypoints) {
    ALVA_ARG(fe && d_rgba && h_pose7 && h_pose_status && h_n_keypoints && n_pts >= 0 && n_pts <= fe->n_track && n_corr >= 0);
    ALVA_HIP(hipSetDevice(fe->device));
    const int cur = (int) (fe->frame & 1), prv = cur ^ 1;
    // lane A: preprocessImage
    int rc = alva_pyramid_build_from_rgba(fe->A, fe->pyr[cur], d_rgba, rgba_pitch, fe->d_gray, (size_t) fe->width);
    if (rc) return rc;
    rc = alva_ctx_wait(fe->B, fe->A);  // lane B may start as soon as the gray image exists
    if (rc) return rc;
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
    // lane B: detector + descriptors of this frame
    rc = alva_orb_detect_and_compute(fe->B, fe->orb, fe->d_gray, (size_t) fe->width, fe->d_kp[cur], fe->d_desc[cur], fe->cap, nullptr);
    if (rc) return rc;
    // lane B: match against the previous frame's descriptors, enqueued right behind the detector with both counts still on
    // the device (nothing of it is left on the host's path between two frames)
    ALVA_HIP(hipMemcpyAsync(fe->d_counts + cur, alva_orb_device_count(fe->orb), sizeof(int), hipMemcpyDeviceToDevice, fe->B->stream));
    if (fe->frame > 0 && fe->n_desc[prv] > 0) {
        rc = alva_bf_match_hamming_devcount(fe->B, fe->d_desc[cur], fe->d_counts + cur, fe->cap, fe->d_desc[prv], fe->d_counts + prv,
                                            fe->n_desc[prv], fe->d_match, fe->d_match + fe->cap);
        if (rc) return rc;
    }
    // host results
    rc = alva_compute_pose_collect(fe->A, h_pose7, nullptr, nullptr, h_pose_status);
    if (rc) return rc;
    int nkp = 0;
    rc = alva_orb_collect(fe->B, fe->orb, &nkp);
    if (rc) return rc;
    nkp = nkp < fe->cap ? nkp : fe->cap;
    fe->n_desc[cur] = nkp;
    *h_n_keypoints = nkp;
    fe->frame++;
    return ALVA_OK;
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
    if (rc) return rc;
    return alva_ctx_sync(fe->B);
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
    std::vector<std::thread> th;
    std::vector<std::chrono::steady_clock::time_point> t_end((size_t) n_streams);
    auto body = [&](int s) {
        double pose[7];
        int st = 0, nkp = 0, ok = 0;
        auto one = [&](int k) {
            const int rc = alva_frontend_track(fes[s], d_frames[(size_t) s * ring + (size_t) (k % ring)], rgba_pitch, d_pts[s], n_pts,
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
