// VisualFrontend::trackMono for B lock-step cameras: every stage is ONE launch that carries all cameras.
//
// The per-frame path of the reference (src/slam/src/visual_frontend.cpp:83-150) is preprocessImage -> kltTracking -> computePose;
// the detector only runs when the mapper decides on a keyframe.  One 640x480 camera gives each of those stages a few microseconds
// to a few tens of microseconds of work in a dependent chain, so a single stream cannot occupy 256 CUs, and threads driving
// independent streams run into the launch path of the runtime.  A rig of cameras (or a server tracking many
// sessions) has a data-parallel dimension the reference cannot use: here it becomes the grid's y / z dimension.
//
//   preprocessImage  x B   alva_pyramid_build_from_rgba_batch                      5 launches   (image.hip)
//   kltTracking      x B   k_klt_batch_q                                           1 launch     (klt.hip)
//   computePose      x B   k_p3p_hyp_batch -> k_p3p_batch -> k_p3p_select_batch -> k_pnp_batch   4 launches   (p3p.hip, pnp.hip)
//
// in stream order on one HIP stream (the pose is solved from what the tracker has just moved), one host wait for the B poses; the
// optional detector lane (cv::ORB + Hamming match per camera, 14 launches) runs on a second stream beside the tracker + pose chain.  Every camera keeps its own pyramids, keypoints, correspondences and counts; the
// device code of a camera is the single-camera code (klt_point / p3p_block / pnp_block), so results are bit-identical to B calls of
// alva_frontend_track (tests/test_gpu_track_batch.py).
#include "common.hpp"
#include "pose_internal.hpp"
#include <algorithm>
#include <cstdlib>
#include <unordered_map>

size_t alva_klt_batch_item_size();
int alva_klt_batch_item_fill(void *dst, const alva_pyramid *prev, const alva_pyramid *curr, const float *d_pts, const float *d_init,
                             float *d_out, uint8_t *d_status, int n);
int alva_fbklt_track_batch_enqueue(alva_ctx *ctx, const void *d_items, int count, int n_max, int num_levels, float err_thresh, float fb_dist,
                                   int max_iters, float eps, int lanes);
size_t alva_p3p_batch_item_size();
size_t alva_p3p_batch_scratch_bytes(int H);
int alva_p3p_batch_item_fill(void *dst, const double *d_bearings, const double *d_wpts, int n, int max_iters, float err_threshold, float fx,
                             float fy, int H, const int *d_samples, uint8_t *d_scratch, int *d_counter, P3pSelectOut *d_out,
                             uint8_t *d_inlier);
int alva_p3p_batch_enqueue(alva_ctx *ctx, const void *d_items, int count, int H_max, int n_max);
size_t alva_pnp_batch_item_size();
size_t alva_pnp_out_size();
size_t alva_pnp_batch_scratch_bytes(int n);
int alva_pnp_batch_item_fill(void *dst, const double *d_uv, const double *d_wpts, int n, int pnp_iters, float chi2_th, float fx, float fy,
                             float cx, float cy, uint8_t *d_scratch, void *out, const P3pSelectOut *d_sel, const uint8_t *d_inlier0);
int alva_pnp_batch_enqueue(alva_ctx *ctx, const void *d_items, int count);
int alva_pnp_out_decode(const void *out, int p3p_iters, int H, int max_draws, double *h_pose7, int *h_status, int *needs_more_draws);

namespace {
constexpr int P3P_ITERS = 100;                 // state.hpp:64-76, as alva_frontend_track
constexpr int P3P_DRAWS = P3P_ITERS + 28;      // first prefix of the sample stream (alva_compute_pose_enqueue)
constexpr int P3P_MAX_DRAWS = P3P_ITERS * 11;  // Lmeds.hpp:67
constexpr size_t OUT_STRIDE = 256;
size_t up(size_t x, size_t a) { return (x + a - 1) / a * a; }
}  // namespace

struct alva_track_batch {
    int device = 0, width = 0, height = 0, B = 0, n_track = 0, n_corr = 0, klt_levels = 3, klt_lanes = 5;
    alva_ctx *ctx = nullptr;  // tracker lane: pyramids -> KLT -> P3P -> PnP
    std::vector<alva_pyramid *> pyr[2];
    uint8_t *slab = nullptr;      // per camera: tracked | status | p3p scratch | selection | inlier mask | pnp scratch; then counters
    size_t cam_stride = 0, off_status = 0, off_p3p = 0, off_sel = 0, off_inl = 0, off_pnp = 0;
    int *d_counters = nullptr;
    uint8_t *d_items = nullptr, *h_items = nullptr;  // klt | p3p | pnp argument blocks (device copy, pinned staging)
    size_t off_items_p3p = 0, off_items_pnp = 0, items_bytes = 0;
    uint8_t *h_out = nullptr;     // pinned: PnpOut per camera, written by k_pnp_batch
    std::unordered_map<int, int *> samples;  // n -> device copy of the first P3P_DRAWS samples of the fixed-seed stream
    std::vector<int> slot;        // camera -> index among the cameras that solve a pose this frame (-1: fewer than 4 correspondences)
    // keyframe branch's feature work (optional, alva_track_batch_enable_detector): cv::ORB + BFMatcher per camera on a third lane
    alva_ctx *det = nullptr;
    int orb_features = 0, cap = 0;
    std::vector<alva_orb *> orbs;
    uint8_t *det_slab = nullptr;  // per camera: gray | kp[2] | desc[2] | match idx + dist
    size_t det_stride = 0, off_kp[2] = {0, 0}, off_desc[2] = {0, 0}, off_match = 0;
    std::vector<uint8_t *> gray_ptrs;
    std::vector<int> n_desc[2];
    long frame = 0, fallbacks = 0;  // fallbacks: cameras re-solved by the single-camera call (P3P prefix too short)
};

extern "C" void alva_track_batch_destroy(alva_track_batch *tb) {
    if (!tb) return;
    (void) hipSetDevice(tb->device);
    if (tb->ctx) (void) alva_ctx_sync(tb->ctx);
    for (auto &v: tb->pyr)
        for (auto p: v)
            if (p) alva_pyramid_destroy(p);
    if (tb->det && tb->det != tb->ctx) (void) alva_ctx_sync(tb->det);
    for (auto o: tb->orbs)
        if (o) alva_orb_destroy(o);
    if (tb->det_slab) (void) hipFree(tb->det_slab);
    if (tb->det && tb->det != tb->ctx) alva_ctx_destroy(tb->det);
    for (auto &kv: tb->samples) (void) hipFree(kv.second);
    if (tb->slab) (void) hipFree(tb->slab);
    if (tb->d_items) (void) hipFree(tb->d_items);
    if (tb->h_items) (void) hipHostFree(tb->h_items);
    if (tb->h_out) (void) hipHostFree(tb->h_out);
    if (tb->ctx) alva_ctx_destroy(tb->ctx);
    delete tb;
}

extern "C" int alva_track_batch_create(int device, int width, int height, int cameras, int max_tracked, int max_corr, alva_track_batch **out) {
    ALVA_ARG(out && width >= 64 && height >= 64 && width % 4 == 0 && cameras > 0 && cameras <= 4096 && max_tracked > 0 && max_corr >= 4 &&
             max_corr <= 7168);
    static_assert(sizeof(P3pSelectOut) <= 256, "selection slot");
    if (alva_pnp_out_size() > OUT_STRIDE) {
        alva_set_error("alva_track_batch_create: result slot too small");
        return ALVA_ERR_STATE;
    }
    alva_track_batch *tb = new alva_track_batch();
    tb->device = device;
    tb->width = width;
    tb->height = height;
    tb->B = cameras;
    tb->n_track = max_tracked;
    tb->n_corr = max_corr;
    tb->slot.assign((size_t) cameras, -1);
    if (const char *e = std::getenv("ALVA_KLT_BATCH_LANES")) {
        const int v = atoi(e);
        if (v == 5 || v == 8 || v == 16 || v == 32 || v == 64) tb->klt_lanes = v;
    }
    int rc = alva_ctx_create(device, nullptr, 1, &tb->ctx);
    for (int k = 0; k < 2 && !rc; k++) {
        tb->pyr[k].assign((size_t) cameras, nullptr);
        for (int c = 0; c < cameras && !rc; c++) rc = alva_pyramid_create(tb->ctx, width, height, 9, 3, &tb->pyr[k][(size_t) c]);  // state.hpp:53-54
    }
    tb->off_status = up((size_t) max_tracked * 2 * sizeof(float), 64);
    tb->off_p3p = tb->off_status + up((size_t) max_tracked, 64);
    tb->off_sel = tb->off_p3p + alva_p3p_batch_scratch_bytes(P3P_DRAWS);
    tb->off_inl = tb->off_sel + 256;
    tb->off_pnp = tb->off_inl + up((size_t) max_corr, 64);
    tb->cam_stride = up(tb->off_pnp + alva_pnp_batch_scratch_bytes(max_corr), 256);
    const size_t slab_bytes = tb->cam_stride * (size_t) cameras + up((size_t) cameras * sizeof(int), 256);
    tb->off_items_p3p = up(alva_klt_batch_item_size() * (size_t) cameras, 256);
    tb->off_items_pnp = tb->off_items_p3p + up(alva_p3p_batch_item_size() * (size_t) cameras, 256);
    tb->items_bytes = tb->off_items_pnp + up(alva_pnp_batch_item_size() * (size_t) cameras, 256);
    auto check = [&](hipError_t e, const char *what) {
        if (rc || e == hipSuccess) return;
        alva_set_error("alva_track_batch_create: %s failed: %s", what, hipGetErrorString(e));
        rc = ALVA_ERR_NOMEM;
    };
    if (!rc) check(hipMalloc((void **) &tb->slab, slab_bytes), "hipMalloc(slab)");
    if (!rc) check(hipMemset(tb->slab, 0, slab_bytes), "hipMemset(slab)");
    if (!rc) check(hipMalloc((void **) &tb->d_items, tb->items_bytes), "hipMalloc(items)");
    if (!rc) check(hipHostMalloc((void **) &tb->h_items, tb->items_bytes, hipHostMallocDefault), "hipHostMalloc(items)");
    if (!rc) check(hipHostMalloc((void **) &tb->h_out, OUT_STRIDE * (size_t) cameras, hipHostMallocDefault), "hipHostMalloc(results)");
    if (rc) {
        alva_track_batch_destroy(tb);
        return rc;
    }
    tb->d_counters = (int *) (tb->slab + tb->cam_stride * (size_t) cameras);
    *out = tb;
    return ALVA_OK;
}

// the first P3P_DRAWS samples of SampleConsensusProblem's fixed-seed stream for n points (p3p.hip Sampler), resident on the device
static int samples_for(alva_track_batch *tb, int n, const int **out) {
    auto it = tb->samples.find(n);
    if (it != tb->samples.end()) {
        *out = it->second;
        return ALVA_OK;
    }
    std::vector<int> h((size_t) P3P_DRAWS * 4);
    int rc = alva_p3p_draw_samples(n, P3P_DRAWS, 0, 12345u, h.data());
    if (rc) return rc;
    int *d = nullptr;
    ALVA_HIP(hipMalloc((void **) &d, h.size() * sizeof(int)));
    ALVA_HIP(hipMemcpy(d, h.data(), h.size() * sizeof(int), hipMemcpyHostToDevice));
    tb->samples.emplace(n, d);
    *out = d;
    return ALVA_OK;
}

// The detector of every camera: cv::ORB::detectAndCompute(orb_features, 1.2, 8 levels, FAST 20) on the frame's gray image and a
// brute-force Hamming match against the camera's previous descriptors, as alva_frontend_track does for one camera.  Call before
// the first step.
extern "C" int alva_track_batch_enable_detector(alva_track_batch *tb, int orb_features) {
    ALVA_ARG(tb && orb_features > 0 && tb->frame == 0 && !tb->det);
    ALVA_HIP(hipSetDevice(tb->device));
    const int B = tb->B;
    int rc = ALVA_OK;
    if (std::getenv("ALVA_TRACK_BATCH_ONE_LANE")) tb->det = tb->ctx;
    else rc = alva_ctx_create(tb->device, nullptr, 1, &tb->det);
    if (rc) return rc;
    tb->orb_features = orb_features;
    tb->cap = 4 * orb_features + 1024;
    tb->orbs.assign((size_t) B, nullptr);
    for (int c = 0; c < B && !rc; c++) rc = alva_orb_create(tb->det, tb->width, tb->height, orb_features, 1.2f, 8, 20, &tb->orbs[(size_t) c]);
    if (rc) return rc;
    size_t off = up((size_t) tb->width * tb->height, 256);
    for (int k = 0; k < 2; k++) {
        tb->off_kp[k] = off;
        off += up((size_t) tb->cap * 6 * sizeof(float), 256);
    }
    for (int k = 0; k < 2; k++) {
        tb->off_desc[k] = off;
        off += up((size_t) tb->cap * 32, 256);
    }
    tb->off_match = off;
    off += up((size_t) tb->cap * 2 * sizeof(int), 256);
    tb->det_stride = off;
    if (hipMalloc((void **) &tb->det_slab, tb->det_stride * (size_t) B) != hipSuccess) {
        alva_set_error("alva_track_batch_enable_detector: hipMalloc(%zu) failed", tb->det_stride * (size_t) B);
        return ALVA_ERR_NOMEM;
    }
    tb->gray_ptrs.resize((size_t) B);
    for (int c = 0; c < B; c++) tb->gray_ptrs[(size_t) c] = tb->det_slab + tb->det_stride * (size_t) c;
    tb->n_desc[0].assign((size_t) B, 0);
    tb->n_desc[1].assign((size_t) B, 0);
    return ALVA_OK;
}

extern "C" int alva_track_batch_step_detect(alva_track_batch *tb, const uint8_t *const *d_rgba, size_t rgba_pitch, const float *const *d_pts,
                                            const int *n_pts, const double *const *d_bearings, const double *const *d_uv,
                                            const double *const *d_wpts, const int *n_corr, float fx, float fy, float cx, float cy,
                                            double *h_pose7, int *h_pose_status, int *h_n_keypoints) {
    ALVA_ARG(tb && d_rgba && n_pts && n_corr && h_pose7 && h_pose_status && (!tb->det || h_n_keypoints));
    ALVA_HIP(hipSetDevice(tb->device));
    const int B = tb->B, cur = (int) (tb->frame & 1), prv = cur ^ 1;
    alva_ctx *ctx = tb->ctx;
    int n_pts_max = 0, n_corr_max = 0, n_pose = 0;
    for (int c = 0; c < B; c++) {
        ALVA_ARG(n_pts[c] >= 0 && n_pts[c] <= tb->n_track && n_corr[c] >= 0 && n_corr[c] <= tb->n_corr);
        ALVA_ARG(n_pts[c] == 0 || tb->frame == 0 || (d_pts && d_pts[c]));
        ALVA_ARG(n_corr[c] < 4 || (d_bearings && d_uv && d_wpts && d_bearings[c] && d_uv[c] && d_wpts[c]));
    }
    // argument blocks of the three launches behind the pyramids, one copy
    int rc = ALVA_OK;
    const size_t klt_sz = alva_klt_batch_item_size(), p3p_sz = alva_p3p_batch_item_size(), pnp_sz = alva_pnp_batch_item_size();
    for (int c = 0; c < B; c++) {
        uint8_t *cam = tb->slab + tb->cam_stride * (size_t) c;
        const int np = tb->frame > 0 ? n_pts[c] : 0;
        // kltTracking: prior = the previous positions (feature_tracker.cpp:5-111), result in the camera's own buffer
        rc = alva_klt_batch_item_fill(tb->h_items + klt_sz * (size_t) c, tb->pyr[prv][(size_t) c], tb->pyr[cur][(size_t) c], np ? d_pts[c] : nullptr,
                                      np ? d_pts[c] : nullptr, (float *) cam, cam + tb->off_status, np);
        if (rc) return rc;
        n_pts_max = std::max(n_pts_max, np);
        h_pose_status[c] = 0;
        tb->slot[(size_t) c] = -1;
        if (n_corr[c] < 4) continue;  // visual_frontend.cpp:249-257
        const int *d_samples = nullptr;
        rc = samples_for(tb, n_corr[c], &d_samples);
        if (rc) return rc;
        P3pSelectOut *sel = (P3pSelectOut *) (cam + tb->off_sel);
        rc = alva_p3p_batch_item_fill(tb->h_items + tb->off_items_p3p + p3p_sz * (size_t) n_pose, d_bearings[c], d_wpts[c], n_corr[c], P3P_ITERS, 3.0f,
                                      fx, fy, P3P_DRAWS, d_samples, cam + tb->off_p3p, tb->d_counters + c, sel, cam + tb->off_inl);
        if (!rc)
            rc = alva_pnp_batch_item_fill(tb->h_items + tb->off_items_pnp + pnp_sz * (size_t) n_pose, d_uv[c], d_wpts[c], n_corr[c], 5, 5.9915f, fx, fy,
                                          cx, cy, cam + tb->off_pnp, tb->h_out + OUT_STRIDE * (size_t) c, sel, cam + tb->off_inl);
        if (rc) return rc;
        tb->slot[(size_t) c] = n_pose++;
        n_corr_max = std::max(n_corr_max, n_corr[c]);
    }
    ALVA_HIP(hipMemcpyAsync(tb->d_items, tb->h_items, tb->items_bytes, hipMemcpyHostToDevice, ctx->stream));
    // tracker lane: preprocessImage, kltTracking, then computePose of every camera
    rc = alva_pyramid_build_from_rgba_batch(ctx, tb->pyr[cur].data(), d_rgba, rgba_pitch, tb->det ? tb->gray_ptrs.data() : nullptr,
                                            tb->det ? (size_t) tb->width : 0, B);
    if (rc) return rc;
    if (tb->det) {
        rc = alva_ctx_wait(tb->det, ctx);  // the gray images; recorded in front of the tracker's launch
        if (rc) return rc;
    }
    rc = alva_fbklt_track_batch_enqueue(ctx, tb->d_items, B, n_pts_max, tb->klt_levels, 30.f, 0.5f, 30, 0.01f, tb->klt_lanes);  // state.hpp:55-59
    if (rc) return rc;
    // computePose BEHIND the tracker in stream order, as in alva_frontend_track: in the reference the pose is solved from the keypoints
    // kltTracking has just moved (visual_frontend.cpp:104-110), so the timed chain is the dependent one although the correspondence
    // buffers are inputs of this call
    if (n_pose > 0) {
        rc = alva_p3p_batch_enqueue(ctx, tb->d_items + tb->off_items_p3p, n_pose, P3P_DRAWS, n_corr_max);
        if (!rc) rc = alva_pnp_batch_enqueue(ctx, tb->d_items + tb->off_items_pnp, n_pose);
        if (rc) return rc;
    }
    if (tb->det) {
        // detector lane: detect + describe every camera's frame, then match against the camera's previous descriptors with the new
        // counts still on the device
        std::vector<float *> kp((size_t) B);
        std::vector<uint8_t *> desc((size_t) B);
        std::vector<const uint8_t *> gray((size_t) B), query((size_t) B), train((size_t) B);
        std::vector<const int *> dnq((size_t) B);
        std::vector<int *> idx((size_t) B), dist((size_t) B);
        int any_train = 0;
        for (int c = 0; c < B; c++) {
            uint8_t *base = tb->det_slab + tb->det_stride * (size_t) c;
            gray[(size_t) c] = base;
            kp[(size_t) c] = (float *) (base + tb->off_kp[cur]);
            desc[(size_t) c] = base + tb->off_desc[cur];
            query[(size_t) c] = desc[(size_t) c];
            train[(size_t) c] = base + tb->off_desc[prv];
            dnq[(size_t) c] = alva_orb_device_count(tb->orbs[(size_t) c]);
            idx[(size_t) c] = (int *) (base + tb->off_match);
            dist[(size_t) c] = idx[(size_t) c] + tb->cap;
            if (tb->frame == 0) tb->n_desc[prv][(size_t) c] = 0;
            any_train |= tb->n_desc[prv][(size_t) c] > 0;
        }
        rc = alva_orb_detect_and_compute_batch(tb->det, tb->orbs.data(), B, gray.data(), (size_t) tb->width, kp.data(), desc.data(), tb->cap);
        if (!rc && any_train)
            rc = alva_bf_match_hamming_batch(tb->det, B, query.data(), dnq.data(), tb->cap, train.data(), tb->n_desc[prv].data(), idx.data(), dist.data(),
                                             tb->orb_features + 64);
        if (rc) return rc;
        // the detector lane's count first, as alva_frontend_track does (its short commands retire while the tracker still computes)
        rc = alva_orb_collect_batch(tb->det, tb->orbs.data(), B, h_n_keypoints);
        if (rc) return rc;
        for (int c = 0; c < B; c++) {
            h_n_keypoints[c] = std::min(h_n_keypoints[c], tb->cap);
            tb->n_desc[cur][(size_t) c] = h_n_keypoints[c];
        }
    }
    ALVA_HIP(hipStreamSynchronize(ctx->stream));
    for (int c = 0; c < B; c++) {
        if (tb->slot[(size_t) c] < 0) continue;
        int more = 0;
        rc = alva_pnp_out_decode(tb->h_out + OUT_STRIDE * (size_t) c, P3P_ITERS, P3P_DRAWS, P3P_MAX_DRAWS, h_pose7 + 7 * (size_t) c, h_pose_status + c, &more);
        if (rc) return rc;
        if (more) {
            tb->fallbacks++;
            // rare: so many degenerate samples that the first prefix of the stream did not yield 100 models; that camera goes through the
            // single-camera call, which extends the prefix (alva_compute_pose_collect)
            rc = alva_compute_pose(ctx, d_bearings[c], d_uv[c], d_wpts[c], n_corr[c], P3P_ITERS, 3.0f, 0, 12345u, 5, 5.9915f, fx, fy, cx, cy,
                                   h_pose7 + 7 * (size_t) c, nullptr, nullptr, h_pose_status + c);
            if (rc) return rc;
        }
    }
    tb->frame++;
    return ALVA_OK;
}

extern "C" int alva_track_batch_step(alva_track_batch *tb, const uint8_t *const *d_rgba, size_t rgba_pitch, const float *const *d_pts,
                                     const int *n_pts, const double *const *d_bearings, const double *const *d_uv,
                                     const double *const *d_wpts, const int *n_corr, float fx, float fy, float cx, float cy, double *h_pose7,
                                     int *h_pose_status) {
    ALVA_ARG(tb);
    std::vector<int> nkp;
    if (tb->det) nkp.resize((size_t) tb->B);
    return alva_track_batch_step_detect(tb, d_rgba, rgba_pitch, d_pts, n_pts, d_bearings, d_uv, d_wpts, n_corr, fx, fy, cx, cy, h_pose7, h_pose_status,
                                        tb->det ? nkp.data() : nullptr);
}

// Device-resident detector results of camera `cam` from the last step: keypoints [n][6] (x, y, size, angle, response, octave),
// descriptors [n][32], and per keypoint the index / distance of its best match among the camera's PREVIOUS descriptors.
extern "C" int alva_track_batch_detections(alva_track_batch *tb, int cam, const float **d_keypoints, const uint8_t **d_descriptors,
                                           const int **d_match_idx, const int **d_match_dist) {
    ALVA_ARG(tb && tb->det && cam >= 0 && cam < tb->B && tb->frame > 0);
    const int last = (int) ((tb->frame - 1) & 1);
    const uint8_t *base = tb->det_slab + tb->det_stride * (size_t) cam;
    if (d_keypoints) *d_keypoints = (const float *) (base + tb->off_kp[last]);
    if (d_descriptors) *d_descriptors = base + tb->off_desc[last];
    if (d_match_idx) *d_match_idx = (const int *) (base + tb->off_match);
    if (d_match_dist) *d_match_dist = (const int *) (base + tb->off_match) + tb->cap;
    return ALVA_OK;
}

// Device-resident kltTracking results of camera `cam` from the last step (valid until the next one).
extern "C" int alva_track_batch_results(alva_track_batch *tb, int cam, const float **d_tracked, const uint8_t **d_track_status) {
    ALVA_ARG(tb && cam >= 0 && cam < tb->B && tb->frame > 0);
    const uint8_t *base = tb->slab + tb->cam_stride * (size_t) cam;
    if (d_tracked) *d_tracked = (const float *) base;
    if (d_track_status) *d_track_status = base + tb->off_status;
    return ALVA_OK;
}

// lanes of a wavefront that share one keypoint in the tracking launch: 32 (default; two keypoints per wave), 16, or 64 (the
// single-camera kernel's layout).  Results do not depend on it.
extern "C" int alva_track_batch_set_klt_lanes(alva_track_batch *tb, int lanes) {
    ALVA_ARG(tb && (lanes == 5 || lanes == 8 || lanes == 16 || lanes == 32 || lanes == 64));
    tb->klt_lanes = lanes;
    return ALVA_OK;
}

extern "C" int alva_track_batch_stats(alva_track_batch *tb, long *frames, long *single_camera_fallbacks) {
    ALVA_ARG(tb);
    if (frames) *frames = tb->frame;
    if (single_camera_fallbacks) *single_camera_fallbacks = tb->fallbacks;
    return ALVA_OK;
}

extern "C" alva_ctx *alva_track_batch_ctx(alva_track_batch *tb) { return tb ? tb->ctx : nullptr; }
