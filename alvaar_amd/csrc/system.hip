// a1 / §8(b): the reference's `System` surface over the HIP hot path (host orchestration only -- every numeric stage is
// a call through include/alvaar_hip.h).
//
// Mirrors, in reduced form, System::findCameraPose -> VisualFrontend::track/process (src/slam/src/system.cpp:106-175,
// visual_frontend.cpp:21-101): per frame  upload RGBA -> gray + LK pyramid (fused) -> forward-backward KLT of the frame's
// keypoints (visual_frontend.cpp:103-243; every keypoint uses its previous position as prior, 3 levels) -> P3P-LMedS +
// robust PnP on the keypoints that carry a 3-D map point (:245-417) -> constant-velocity motion model update; and on a
// keyframe (:554-594, here: fewer than half of the cells still tracked)  MapManager::extractKeypoints
// (map_manager.cpp:193-222): grid detection in the unoccupied cells + ORB description.
// Cold start (visual_frontend.cpp:33-71, :419-551; mapper.cpp:9-51, :144-291): the first frame becomes keyframe 0; while the map
// is not initialised every frame checks checkReadyForInit (median / rotation-compensated average parallax against the keyframe
// > 40 px, then the 5-point RANSAC of alva_compute_5pt_essential, translation normalised to 1); the frame that passes becomes
// keyframe 1 and its 2-D keypoints are triangulated against the keyframe that first observed them (alva_triangulate).  Every
// later keyframe (checkNewKeyframeRequired, :554-594) extracts new keypoints and triangulates the same way.
// findPlane runs the intended plane fit (alva_find_plane, parity unpinned) on the current frame's 3-D keypoints.
// NOT mirrored (the reference's L2 map layer): matching to the local map, local-BA scheduling, keyframe / map-point culling.  alva_local_ba and alva_match_to_map exist behind the C ABI; the graph bookkeeping that feeds them does not.
#include "common.hpp"
#include "lm_device.hpp"
#include "../../include/alvaar_system.h"
#include <algorithm>
#include <array>
#include <cmath>
#include <set>
#include <unordered_map>

static thread_local char g_sys_err[256] = "";
extern "C" const char *alva_system_last_error(void) { return g_sys_err[0] ? g_sys_err : alva_last_error(); }

namespace {
struct Keypoint {
    int id;
    float px, py;
    bool is3d;
    double X[3];
    int kf_first = -1, kf_last = -1;  // keyframe that first observed it / latest keyframe that holds it
    float fpx = 0, fpy = 0, lpx = 0, lpy = 0;  // its position in those keyframes
};
struct KeyframeRec {
    double pose[7];  // Twc
    int frame_id, n3d;
};
}  // namespace

struct alva_system {
    int device = 0;
    alva_ctx *ctx = nullptr;
    int w = 0, h = 0, cell = 40, border = 20;          // system.cpp:15,29
    double fx = 0, fy = 0, cx = 0, cy = 0;
    alva_pyramid *pyr[2] = {nullptr, nullptr};
    int cur = 0;
    bool have_prev = false, configured = false;
    uint8_t *d_rgba = nullptr, *d_gray = nullptr, *d_desc = nullptr, *d_status = nullptr, *d_valid = nullptr;
    double *d_tri = nullptr;  // triangulation staging: T blocks | bv_l | bv_r | lpt | wpt | inv depth | parallax | unpx_l | unpx_r | group | status
    std::vector<KeyframeRec> kfs;
    bool ready = false;       // state_->slamReadyForInit_
    bool external_map = false;
    uint8_t *h_rgba_pinned = nullptr;
    float *d_pts = nullptr, *d_prior = nullptr, *d_new = nullptr;
    double *d_bv = nullptr, *d_wpt = nullptr, *d_uv = nullptr;
    int cap = 0;
    std::vector<Keypoint> kps;
    int next_id = 0, frame_id = 0, pose_failures = 0;
    double max_quality = 0.001;                          // state.hpp:57 extractorMaxQuality_
    double pose[7] = {0, 0, 0, 0, 0, 0, 1};              // Twc
    double imu_translation[3] = {0, 0, 0}, prev_translation[3] = {0, 0, 0};
};

static void pose_to_array(const double *p7, float *out) {  // Utils::toPoseArray, utils.cpp:3-27
    Se3 T;
    se3_from_pose7(p7, T);
    for (int r = 0; r < 3; r++) {
        for (int c = 0; c < 3; c++) out[4 * r + c] = (float) T.R[3 * r + c];
        out[4 * r + 3] = 0.f;
    }
    out[12] = (float) p7[0];
    out[13] = (float) p7[1];
    out[14] = (float) p7[2];
    out[15] = 1.f;
}

static void sys_free(alva_system *s) {
    if (!s) return;
    (void) hipSetDevice(s->device);
    if (s->ctx) (void) alva_ctx_sync(s->ctx);
    for (auto &p: s->pyr) {
        alva_pyramid_destroy(p);
        p = nullptr;
    }
    void *bufs[] = {s->d_rgba, s->d_gray, s->d_desc, s->d_status, s->d_valid, s->d_pts, s->d_prior, s->d_new, s->d_bv, s->d_wpt, s->d_uv, s->d_tri};
    s->d_tri = nullptr;
    for (void *b: bufs)
        if (b) (void) hipFree(b);
    s->d_rgba = s->d_gray = s->d_desc = s->d_status = s->d_valid = nullptr;
    s->d_pts = s->d_prior = s->d_new = nullptr;
    s->d_bv = s->d_wpt = s->d_uv = nullptr;
    if (s->h_rgba_pinned) (void) hipHostFree(s->h_rgba_pinned);
    s->h_rgba_pinned = nullptr;
}

extern "C" int alva_system_create(int device, alva_system **out) {
    if (!out) return ALVA_ERR_ARG;
    alva_system *s = new alva_system();
    s->device = device;
    int rc = alva_ctx_create(device, nullptr, 1, &s->ctx);
    if (rc) {
        snprintf(g_sys_err, sizeof(g_sys_err), "%s", alva_last_error());
        delete s;
        return rc;
    }
    *out = s;
    return ALVA_OK;
}

extern "C" void alva_system_destroy(alva_system *s) {
    if (!s) return;
    sys_free(s);
    alva_ctx_destroy(s->ctx);
    delete s;
}

extern "C" int alva_system_configure(alva_system *s, int width, int height, double fx, double fy, double cx, double cy, double k1,
                                     double k2, double p1, double p2) {
    if (!s || width < 64 || height < 64 || width % 4) return ALVA_ERR_ARG;
    if (k1 != 0 || k2 != 0 || p1 != 0 || p2 != 0) {
        snprintf(g_sys_err, sizeof(g_sys_err), "distortion coefficients are not supported yet (SURVEY.md 8f row 4)");
        return ALVA_ERR_ARG;
    }
    sys_free(s);
    s->w = width; s->h = height; s->fx = fx; s->fy = fy; s->cx = cx; s->cy = cy;
    s->cap = 2 * (width / s->cell) * (height / s->cell) + 64;   // state.cpp:8-11: one keypoint per cell (+ secondaries)
    ALVA_HIP(hipSetDevice(s->device));
    const size_t P = (size_t) width * height;
    ALVA_HIP(hipMalloc((void **) &s->d_rgba, P * 4));
    ALVA_HIP(hipMalloc((void **) &s->d_gray, P));
    ALVA_HIP(hipHostMalloc((void **) &s->h_rgba_pinned, P * 4, hipHostMallocDefault));
    const size_t c = (size_t) s->cap;
    ALVA_HIP(hipMalloc((void **) &s->d_desc, c * 32));
    ALVA_HIP(hipMalloc((void **) &s->d_status, c));
    ALVA_HIP(hipMalloc((void **) &s->d_valid, c));
    ALVA_HIP(hipMalloc((void **) &s->d_pts, c * 8));
    ALVA_HIP(hipMalloc((void **) &s->d_prior, c * 8));
    ALVA_HIP(hipMalloc((void **) &s->d_new, c * 8));
    ALVA_HIP(hipMalloc((void **) &s->d_bv, c * 24));
    ALVA_HIP(hipMalloc((void **) &s->d_wpt, c * 24));
    ALVA_HIP(hipMalloc((void **) &s->d_uv, c * 16));
    ALVA_HIP(hipMalloc((void **) &s->d_tri, 32 * 36 * 8 + c * (24 * 4 + 8 * 2 + 8 * 2 + 4 + 8)));
    for (auto &p: s->pyr) {
        int rc = alva_pyramid_create(s->ctx, width, height, 9, 3, &p);   // state.hpp:51-53: 3 levels, 9x9 window
        if (rc) return rc;
    }
    s->configured = true;
    alva_system_reset(s);
    return ALVA_OK;
}

extern "C" void alva_system_reset(alva_system *s) {  // system.cpp:42-55
    if (!s) return;
    s->kps.clear();
    s->kfs.clear();
    s->ready = false;
    s->external_map = false;
    s->have_prev = false;
    s->pose_failures = 0;
    s->max_quality = 0.001;
    const double id[7] = {0, 0, 0, 0, 0, 0, 1};
    memcpy(s->pose, id, sizeof(id));
    memset(s->prev_translation, 0, sizeof(s->prev_translation));
}

// MapManager::extractKeypoints (map_manager.cpp:193-222): detect in the cells not occupied by a tracked keypoint, describe
static int extract_keypoints(alva_system *s) {
    const int nocc = (int) s->kps.size();
    std::vector<float> occ((size_t) nocc * 2);
    for (int i = 0; i < nocc; i++) {
        occ[2 * (size_t) i] = s->kps[(size_t) i].px;
        occ[2 * (size_t) i + 1] = s->kps[(size_t) i].py;
    }
    hipStream_t st = (hipStream_t) alva_ctx_stream(s->ctx);
    if (nocc) ALVA_HIP(hipMemcpyAsync(s->d_pts, occ.data(), occ.size() * 4, hipMemcpyHostToDevice, st));
    int count = 0;
    int rc = alva_detect_grid(s->ctx, s->d_gray, (size_t) s->w, s->w, s->h, s->cell, s->d_pts, nocc, s->border, s->border,
                              s->w - 2 * s->border, s->h - 2 * s->border, &s->max_quality, s->d_new, s->cap, &count);
    if (rc) return rc;
    count = std::min(count, s->cap - nocc);
    if (count <= 0) return ALVA_OK;
    rc = alva_describe(s->ctx, s->d_gray, (size_t) s->w, s->w, s->h, s->d_new, count, s->d_desc, s->d_valid);
    if (rc) return rc;
    std::vector<float> np((size_t) count * 2);
    ALVA_HIP(hipMemcpyAsync(np.data(), s->d_new, np.size() * 4, hipMemcpyDeviceToHost, st));
    ALVA_HIP(hipStreamSynchronize(st));
    for (int i = 0; i < count; i++) {
        Keypoint k{};
        k.id = s->next_id++;
        k.px = np[2 * (size_t) i];
        k.py = np[2 * (size_t) i + 1];
        k.is3d = false;
        s->kps.push_back(k);
    }
    return ALVA_OK;
}

static void bearing_of(const alva_system *s, float px, float py, double *bv) {  // Frame::computeKeypoint: K^-1 px, normalised
    const double x = (px - s->cx) / s->fx, y = (py - s->cy) / s->fy, nn = std::sqrt(x * x + y * y + 1.0);
    bv[0] = x / nn;
    bv[1] = y / nn;
    bv[2] = 1.0 / nn;
}
static void se3_inverse(const Se3 &T, Se3 &Ti) {
    for (int r = 0; r < 3; r++)
        for (int c = 0; c < 3; c++) Ti.R[3 * r + c] = T.R[3 * c + r];
    for (int r = 0; r < 3; r++) Ti.t[r] = -(Ti.R[3 * r] * T.t[0] + Ti.R[3 * r + 1] * T.t[1] + Ti.R[3 * r + 2] * T.t[2]);
}
static void se3_mul(const Se3 &A, const Se3 &B, Se3 &C) {
    for (int r = 0; r < 3; r++)
        for (int c = 0; c < 3; c++) C.R[3 * r + c] = A.R[3 * r] * B.R[c] + A.R[3 * r + 1] * B.R[3 + c] + A.R[3 * r + 2] * B.R[6 + c];
    for (int r = 0; r < 3; r++) C.t[r] = A.R[3 * r] * B.t[0] + A.R[3 * r + 1] * B.t[1] + A.R[3 * r + 2] * B.t[2] + A.t[r];
}
static void rot_to_quat(const double *R, double *q) {  // x y z w
    const double tr = R[0] + R[4] + R[8];
    if (tr > 0) {
        const double S = std::sqrt(tr + 1.0) * 2;
        q[3] = 0.25 * S; q[0] = (R[7] - R[5]) / S; q[1] = (R[2] - R[6]) / S; q[2] = (R[3] - R[1]) / S;
    } else if (R[0] > R[4] && R[0] > R[8]) {
        const double S = std::sqrt(1.0 + R[0] - R[4] - R[8]) * 2;
        q[3] = (R[7] - R[5]) / S; q[0] = 0.25 * S; q[1] = (R[1] + R[3]) / S; q[2] = (R[2] + R[6]) / S;
    } else if (R[4] > R[8]) {
        const double S = std::sqrt(1.0 + R[4] - R[0] - R[8]) * 2;
        q[3] = (R[2] - R[6]) / S; q[0] = (R[1] + R[3]) / S; q[1] = 0.25 * S; q[2] = (R[5] + R[7]) / S;
    } else {
        const double S = std::sqrt(1.0 + R[8] - R[0] - R[4]) * 2;
        q[3] = (R[3] - R[1]) / S; q[0] = (R[2] + R[6]) / S; q[1] = (R[5] + R[7]) / S; q[2] = 0.25 * S;
    }
}
static int count3d(const alva_system *s) {
    int n = 0;
    for (const Keypoint &k: s->kps) n += k.is3d ? 1 : 0;
    return n;
}

// VisualFrontend::computeParallax (visual_frontend.cpp:596-670) against the latest keyframe
static float compute_parallax(const alva_system *s, bool unrotate, bool median) {
    if (s->kfs.empty()) return 0.f;
    const int kf = (int) s->kfs.size() - 1;
    double Rkc[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    if (unrotate) {
        Se3 Tk, Tc;
        se3_from_pose7(s->kfs[(size_t) kf].pose, Tk);
        se3_from_pose7(s->pose, Tc);
        for (int r = 0; r < 3; r++)
            for (int c = 0; c < 3; c++) Rkc[3 * r + c] = Tk.R[r] * Tc.R[c] + Tk.R[3 + r] * Tc.R[3 + c] + Tk.R[6 + r] * Tc.R[6 + c];  // Rkw * Rwc
    }
    float avg = 0.f;
    int n = 0;
    std::set<float> all;
    for (const Keypoint &k: s->kps) {
        if (k.kf_last != kf) continue;
        float ux = k.px, uy = k.py;
        if (unrotate) {
            double bv[3], r[3];
            bearing_of(s, k.px, k.py, bv);
            for (int i = 0; i < 3; i++) r[i] = Rkc[3 * i] * bv[0] + Rkc[3 * i + 1] * bv[1] + Rkc[3 * i + 2] * bv[2];
            ux = (float) (s->fx * r[0] / r[2] + s->cx);
            uy = (float) (s->fy * r[1] / r[2] + s->cy);
        }
        const float p = (float) std::sqrt((double) (ux - k.lpx) * (ux - k.lpx) + (double) (uy - k.lpy) * (uy - k.lpy));
        avg += p;
        n++;
        if (median) all.insert(p);
    }
    if (!n) return 0.f;
    avg /= (float) n;
    if (median) {
        auto it = all.begin();
        std::advance(it, all.size() / 2);
        avg = *it;
    }
    return avg;
}

// Mapper::triangulateTemporal (mapper.cpp:144-291) for the keyframe just created: every 2-D keypoint that an earlier keyframe
// observed first is triangulated against that keyframe
static int triangulate_new_keyframe(alva_system *s) {
    const int newKf = (int) s->kfs.size() - 1;
    std::vector<int> sel, groupOf, kfOfGroup;
    for (size_t i = 0; i < s->kps.size(); i++) {
        const Keypoint &k = s->kps[i];
        if (k.is3d || k.kf_first < 0 || k.kf_first == newKf) continue;
        int g = -1;
        for (size_t q = 0; q < kfOfGroup.size(); q++)
            if (kfOfGroup[q] == k.kf_first) g = (int) q;
        if (g < 0) {
            if (kfOfGroup.size() >= 32) continue;  // staging holds 32 distinct first keyframes; the map keeps 30 (mapper.cpp:15-18)
            g = (int) kfOfGroup.size();
            kfOfGroup.push_back(k.kf_first);
        }
        sel.push_back((int) i);
        groupOf.push_back(g);
    }
    const int n = (int) sel.size(), G = (int) kfOfGroup.size();
    if (!n) return ALVA_OK;
    std::vector<double> T((size_t) G * 36), bvl((size_t) n * 3), bvr((size_t) n * 3);
    std::vector<float> ul((size_t) n * 2), ur((size_t) n * 2);
    Se3 Twr;
    se3_from_pose7(s->kfs[(size_t) newKf].pose, Twr);
    for (int g = 0; g < G; g++) {
        Se3 Twl, Tlw, Tlr, Trl;
        se3_from_pose7(s->kfs[(size_t) kfOfGroup[(size_t) g]].pose, Twl);
        se3_inverse(Twl, Tlw);
        se3_mul(Tlw, Twr, Tlr);  // Tcicj = Tciw * Twcj (:226-228)
        se3_inverse(Tlr, Trl);
        double *o = &T[(size_t) g * 36];
        memcpy(o, Tlr.R, 72); memcpy(o + 9, Tlr.t, 24);
        memcpy(o + 12, Trl.R, 72); memcpy(o + 21, Trl.t, 24);
        memcpy(o + 24, Twl.R, 72); memcpy(o + 33, Twl.t, 24);
    }
    for (int k = 0; k < n; k++) {
        const Keypoint &kp = s->kps[(size_t) sel[(size_t) k]];
        bearing_of(s, kp.fpx, kp.fpy, &bvl[3 * (size_t) k]);
        bearing_of(s, kp.px, kp.py, &bvr[3 * (size_t) k]);
        ul[2 * (size_t) k] = kp.fpx; ul[2 * (size_t) k + 1] = kp.fpy;
        ur[2 * (size_t) k] = kp.px; ur[2 * (size_t) k + 1] = kp.py;
    }
    hipStream_t st = (hipStream_t) alva_ctx_stream(s->ctx);
    const size_t c = (size_t) s->cap;
    double *dT = s->d_tri, *dbl = dT + 32 * 36, *dbr = dbl + 3 * c, *dlp = dbr + 3 * c, *dwp = dlp + 3 * c, *dinv = dwp + 3 * c, *dpar = dinv + c;
    float *dul = (float *) (dpar + c), *dur = dul + 2 * c;
    int *dgrp = (int *) (dur + 2 * c);
    uint8_t *dst = (uint8_t *) (dgrp + c);
    ALVA_HIP(hipMemcpyAsync(dT, T.data(), T.size() * 8, hipMemcpyHostToDevice, st));
    ALVA_HIP(hipMemcpyAsync(dbl, bvl.data(), bvl.size() * 8, hipMemcpyHostToDevice, st));
    ALVA_HIP(hipMemcpyAsync(dbr, bvr.data(), bvr.size() * 8, hipMemcpyHostToDevice, st));
    ALVA_HIP(hipMemcpyAsync(dul, ul.data(), ul.size() * 4, hipMemcpyHostToDevice, st));
    ALVA_HIP(hipMemcpyAsync(dur, ur.data(), ur.size() * 4, hipMemcpyHostToDevice, st));
    ALVA_HIP(hipMemcpyAsync(dgrp, groupOf.data(), groupOf.size() * 4, hipMemcpyHostToDevice, st));
    int rc = alva_triangulate(s->ctx, n, dT, G, dgrp, dbl, dbr, dul, dur, s->fx, s->fy, s->cx, s->cy, 3.0f /* state.hpp:64 */, dlp, dwp, dinv,
                              dst, dpar);
    if (rc) return rc;
    std::vector<double> wp((size_t) n * 3), par((size_t) n);
    std::vector<uint8_t> stt((size_t) n);
    ALVA_HIP(hipMemcpyAsync(wp.data(), dwp, wp.size() * 8, hipMemcpyDeviceToHost, st));
    ALVA_HIP(hipMemcpyAsync(par.data(), dpar, par.size() * 8, hipMemcpyDeviceToHost, st));
    ALVA_HIP(hipMemcpyAsync(stt.data(), dst, stt.size(), hipMemcpyDeviceToHost, st));
    ALVA_HIP(hipStreamSynchronize(st));
    std::vector<uint8_t> drop(s->kps.size(), 0);
    for (int k = 0; k < n; k++) {
        Keypoint &kp = s->kps[(size_t) sel[(size_t) k]];
        if (stt[(size_t) k] == 0) {
            kp.is3d = true;  // MapManager::updateMapPoint (:286)
            for (int q = 0; q < 3; q++) kp.X[q] = wp[3 * (size_t) k + q];
        } else if (par[(size_t) k] > 20.) {
            drop[(size_t) sel[(size_t) k]] = 1;  // removeMapPointObs (:258-262, :274-278)
        }
    }
    std::vector<Keypoint> kept;
    for (size_t i = 0; i < s->kps.size(); i++)
        if (!drop[i]) kept.push_back(s->kps[i]);
    s->kps.swap(kept);
    return ALVA_OK;
}

// MapManager::createKeyframe (map_manager.cpp:45-89) + Mapper::processNewKeyframe's triangulation (mapper.cpp:9-25)
static int create_keyframe(alva_system *s) {
    const size_t before = s->kps.size();
    int rc = extract_keypoints(s);
    if (rc) return rc;
    KeyframeRec kf{};
    memcpy(kf.pose, s->pose, sizeof(kf.pose));
    kf.frame_id = s->frame_id;
    s->kfs.push_back(kf);
    const int id = (int) s->kfs.size() - 1;
    for (size_t i = 0; i < s->kps.size(); i++) {
        Keypoint &k = s->kps[i];
        if (i >= before || k.kf_first < 0) {
            k.kf_first = id;
            k.fpx = k.px;
            k.fpy = k.py;
        }
        k.kf_last = id;
        k.lpx = k.px;
        k.lpy = k.py;
    }
    if (id > 0) {
        rc = triangulate_new_keyframe(s);
        if (rc) return rc;
    }
    s->kfs.back().n3d = count3d(s);
    return ALVA_OK;
}

// VisualFrontend::checkReadyForInit (visual_frontend.cpp:419-551)
static int check_ready_for_init(alva_system *s, bool *ready) {
    *ready = false;
    if (compute_parallax(s, false, true) <= 40.f) return ALVA_OK;  // state.hpp:37 minAvgRotationParallax_
    if (s->kps.size() < 8) return ALVA_OK;
    const int kf = (int) s->kfs.size() - 1;
    std::vector<int> sel;
    std::vector<double> b1, b2;
    float avg = 0.f;
    for (size_t i = 0; i < s->kps.size(); i++) {
        const Keypoint &k = s->kps[i];
        if (k.kf_last != kf) continue;
        double a[3], b[3];
        bearing_of(s, k.lpx, k.lpy, a);
        bearing_of(s, k.px, k.py, b);
        b1.insert(b1.end(), a, a + 3);
        b2.insert(b2.end(), b, b + 3);
        sel.push_back((int) i);
        avg += (float) std::sqrt((double) (k.px - k.lpx) * (k.px - k.lpx) + (double) (k.py - k.lpy) * (k.py - k.lpy));  // both poses are identity
    }
    const int n = (int) sel.size();
    if (n < 8) return ALVA_OK;
    if (avg / (float) n < 40.f) return ALVA_OK;
    hipStream_t st = (hipStream_t) alva_ctx_stream(s->ctx);
    ALVA_HIP(hipMemcpyAsync(s->d_bv, b1.data(), b1.size() * 8, hipMemcpyHostToDevice, st));
    ALVA_HIP(hipMemcpyAsync(s->d_wpt, b2.data(), b2.size() * 8, hipMemcpyHostToDevice, st));
    double R[9], t[3];
    std::vector<uint8_t> inl((size_t) n);
    int ok = 0;
    // state.hpp:67-69: 100 iterations, 3 px, random sampling (a fixed seed here, as for the P3P stage)
    int rc = alva_compute_5pt_essential(s->ctx, s->d_bv, s->d_wpt, n, 100, 3.0f, 1, 0, 12345u, (float) s->fx, (float) s->fy, R, t, inl.data(),
                                        nullptr, &ok);
    if (rc) return rc;
    if (!ok) return ALVA_OK;
    std::vector<uint8_t> drop(s->kps.size(), 0);
    for (int k = 0; k < n; k++)
        if (!inl[(size_t) k]) drop[(size_t) sel[(size_t) k]] = 1;  // :541-544
    std::vector<Keypoint> kept;
    for (size_t i = 0; i < s->kps.size(); i++)
        if (!drop[i]) kept.push_back(s->kps[i]);
    s->kps.swap(kept);
    const double tn = std::sqrt(t[0] * t[0] + t[1] * t[1] + t[2] * t[2]);  // :547 twc.normalize()
    for (int q = 0; q < 3; q++) s->pose[q] = t[q] / tn;
    rot_to_quat(R, s->pose + 3);
    *ready = true;
    return ALVA_OK;
}

// VisualFrontend::checkNewKeyframeRequired (visual_frontend.cpp:554-594)
static bool new_keyframe_required(const alva_system *s) {
    if (s->kfs.empty()) return false;
    const KeyframeRec &kf = s->kfs.back();
    const float med = compute_parallax(s, true, true);
    const int idDiff = s->frame_id - kf.frame_id, n3d = count3d(s), cellsW = s->w / s->cell;
    const int maxKp = cellsW * (s->h / s->cell);  // state.cpp:8-11 frameMaxNumKeypoints_
    std::set<int> occ;
    for (const Keypoint &k: s->kps) occ.insert((int) (k.py / (float) s->cell) * cellsW + (int) (k.px / (float) s->cell));
    const int occupied = (int) occ.size();
    if (idDiff >= 5 && occupied < 0.33 * maxKp) return true;
    if (idDiff >= 2 && n3d < 20) return true;
    if (idDiff < 2 && n3d > 0.5 * maxKp) return false;
    const bool cx = med >= 40.f / 2., c0 = med >= 40.f, c1 = n3d < 0.75 * kf.n3d, c2 = occupied < 0.5 * maxKp && n3d < 0.85 * kf.n3d;
    return (c0 || c1 || c2) && cx;
}

extern "C" int alva_system_find_camera_pose(alva_system *s, const uint8_t *h_rgba, float *h_pose) {
    if (!s || !s->configured || !h_rgba || !h_pose) return ALVA_ERR_ARG;
    hipStream_t st = (hipStream_t) alva_ctx_stream(s->ctx);
    ALVA_HIP(hipSetDevice(s->device));
    s->frame_id++;
    const size_t P = (size_t) s->w * s->h;
    memcpy(s->h_rgba_pinned, h_rgba, P * 4);  // the caller's buffer is pageable wasm-heap style memory
    ALVA_HIP(hipMemcpyAsync(s->d_rgba, s->h_rgba_pinned, P * 4, hipMemcpyHostToDevice, st));
    s->cur ^= 1;
    alva_pyramid *cur = s->pyr[s->cur], *prev = s->pyr[s->cur ^ 1];
    int rc = alva_pyramid_build_from_rgba(s->ctx, cur, s->d_rgba, (size_t) s->w * 4, s->d_gray, (size_t) s->w);  // system.cpp:111-112 + :696
    if (rc) return rc;
    auto reset_and_report = [&]() {  // slamResetRequested_ -> System::reset, status 2 (system.cpp:163-167)
        alva_system_reset(s);
        pose_to_array(s->pose, h_pose);
        return 2;
    };
    // ---- first frame: keyframe 0 (visual_frontend.cpp:37-41) ---------------------------------------------------------
    if (!s->have_prev) {
        rc = create_keyframe(s);
        if (rc) return rc;
        s->have_prev = true;
        pose_to_array(s->pose, h_pose);
        return 3;
    }
    // ---- KLT tracking of the frame's keypoints (visual_frontend.cpp:103-243) -------------------------------------
    if (!s->kps.empty()) {
        const int n = (int) s->kps.size();
        std::vector<float> pts((size_t) n * 2);
        for (int i = 0; i < n; i++) {
            pts[2 * (size_t) i] = s->kps[(size_t) i].px;
            pts[2 * (size_t) i + 1] = s->kps[(size_t) i].py;
        }
        ALVA_HIP(hipMemcpyAsync(s->d_pts, pts.data(), pts.size() * 4, hipMemcpyHostToDevice, st));
        ALVA_HIP(hipMemcpyAsync(s->d_prior, pts.data(), pts.size() * 4, hipMemcpyHostToDevice, st));
        rc = alva_fbklt_track(s->ctx, prev, cur, 3, 30.f, 0.5f, 30, 0.01f, s->d_pts, s->d_prior, s->d_status, n);  // state.hpp:50-56
        if (rc) return rc;
        std::vector<uint8_t> ok((size_t) n);
        ALVA_HIP(hipMemcpyAsync(pts.data(), s->d_prior, pts.size() * 4, hipMemcpyDeviceToHost, st));
        ALVA_HIP(hipMemcpyAsync(ok.data(), s->d_status, (size_t) n, hipMemcpyDeviceToHost, st));
        ALVA_HIP(hipStreamSynchronize(st));
        std::vector<Keypoint> kept;
        kept.reserve((size_t) n);
        for (int i = 0; i < n; i++)
            if (ok[(size_t) i]) {
                Keypoint k = s->kps[(size_t) i];
                k.px = pts[2 * (size_t) i];
                k.py = pts[2 * (size_t) i + 1];
                kept.push_back(k);   // failed tracks are removed from the frame (visual_frontend.cpp:229-232)
            }
        s->kps.swap(kept);
    }
    // ---- not initialised yet (visual_frontend.cpp:52-71) --------------------------------------------------------------
    if (!s->ready) {
        if ((int) s->kps.size() - count3d(s) < 50) return reset_and_report();
        bool ready = false;
        rc = check_ready_for_init(s, &ready);
        if (rc) return rc;
        if (!ready) {
            pose_to_array(s->pose, h_pose);
            return 3;
        }
        s->ready = true;
        rc = create_keyframe(s);  // keyframe 1: new keypoints + triangulation of the tracked ones against keyframe 0
        if (rc) return rc;
        if (s->kfs.size() == 2 && s->kfs.back().n3d < 30) return reset_and_report();  // mapper.cpp:29-39 bad initialisation
        pose_to_array(s->pose, h_pose);
        return 1;
    }
    // ---- pose from the 3-D keypoints (visual_frontend.cpp:245-417) ------------------------------------------------
    std::vector<int> idx3d;
    for (size_t i = 0; i < s->kps.size(); i++)
        if (s->kps[i].is3d) idx3d.push_back((int) i);
    bool good = false;
    if (idx3d.size() >= 4) {
        const int n = (int) idx3d.size();
        std::vector<double> bv((size_t) n * 3), wp((size_t) n * 3), uv((size_t) n * 2);
        for (int k = 0; k < n; k++) {
            const Keypoint &kp = s->kps[(size_t) idx3d[(size_t) k]];
            bearing_of(s, kp.px, kp.py, &bv[3 * (size_t) k]);
            for (int c = 0; c < 3; c++) wp[3 * (size_t) k + c] = kp.X[c];
            uv[2 * (size_t) k] = kp.px;
            uv[2 * (size_t) k + 1] = kp.py;
        }
        ALVA_HIP(hipMemcpyAsync(s->d_bv, bv.data(), bv.size() * 8, hipMemcpyHostToDevice, st));
        ALVA_HIP(hipMemcpyAsync(s->d_wpt, wp.data(), wp.size() * 8, hipMemcpyHostToDevice, st));
        ALVA_HIP(hipMemcpyAsync(s->d_uv, uv.data(), uv.size() * 8, hipMemcpyHostToDevice, st));
        // p3pEnabled_ = true (system.cpp:19): P3P-LMedS -> drop its outliers -> robust PnP on the inliers, chained on the
        // device (visual_frontend.cpp:300-399).  multiViewRandomEnabled_ seeds from the clock in the reference -- fixed seed here.
        double pose7[7];
        std::vector<uint8_t> outP3p((size_t) n), outPnp((size_t) n);
        int pstat = 0;
        rc = alva_compute_pose(s->ctx, s->d_bv, s->d_uv, s->d_wpt, n, 100, 3.0f, 0, 12345u, 5, 5.9915f, (float) s->fx, (float) s->fy,
                               (float) s->cx, (float) s->cy, pose7, outP3p.data(), outPnp.data(), &pstat);
        if (rc) return rc;
        good = pstat == 2;
        if (good) {
            // remove the observations P3P and ceresPnP flagged (visual_frontend.cpp:344-352, :411-414)
            std::vector<uint8_t> drop(s->kps.size(), 0);
            for (int k = 0; k < n; k++)
                if (outP3p[(size_t) k] || outPnp[(size_t) k]) drop[(size_t) idx3d[(size_t) k]] = 1;
            std::vector<Keypoint> kept;
            for (size_t i = 0; i < s->kps.size(); i++)
                if (!drop[i]) kept.push_back(s->kps[i]);
            s->kps.swap(kept);
            memcpy(s->pose, pose7, sizeof(pose7));
            s->pose_failures = 0;
        }
    }
    if (!good && ++s->pose_failures > 3) return reset_and_report();   // visual_frontend.cpp:73-87
    // ---- keyframe decision + creation (visual_frontend.cpp:554-594, map_manager.cpp:45-89, mapper.cpp:9-25) --------------
    bool keyframe;
    if (s->external_map) {
        const int cells = (s->w / s->cell) * (s->h / s->cell);
        keyframe = (int) s->kps.size() < cells / 2;  // host-fed map: only refill the grid
    } else
        keyframe = new_keyframe_required(s);
    if (keyframe) {
        rc = create_keyframe(s);
        if (rc) return rc;
        if (!s->external_map && s->kfs.size() < 11 && s->kfs.back().n3d < 3) return reset_and_report();  // mapper.cpp:41-50
    }
    pose_to_array(s->pose, h_pose);
    return 1;  // system.cpp:169-174: 1 whenever the map is initialised
}

extern "C" int alva_system_find_camera_pose_with_imu(alva_system *s, const uint8_t *h_rgba, const double *h_imu, float *h_pose) {
    if (!s || !h_imu || !h_pose) return ALVA_ERR_ARG;
    float tmp[16];
    const int status = alva_system_find_camera_pose(s, h_rgba, tmp);
    if (status < 0) return status;
    // system.cpp:66-103: orientation = inverse of the IMU quaternion (w, -x, y, z); translation integrates the visual one
    double q[4] = {-h_imu[1], h_imu[2], h_imu[3], h_imu[0]};  // x,y,z,w with x mirrored
    const double nn = std::sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    for (double &v: q) v /= nn;
    double R[9];
    quat_to_R(q, R);
    if (status == 1) {
        for (int c = 0; c < 3; c++) {
            s->imu_translation[c] += s->pose[c] - s->prev_translation[c];
            s->prev_translation[c] = s->pose[c];
        }
    } else {
        memset(s->prev_translation, 0, sizeof(s->prev_translation));
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
    if (!s || !s->configured || !h_pose || num_iterations <= 0) return 0;
    // System::findPlane (system.cpp:123-137) on MapManager::getCurrentFrameMapPoints (map_manager.cpp:340-357): the observed 3-D map
    // points of the current frame.  The plane fit itself is the INTENDED algorithm of processPlane (parity unpinned).
    std::vector<double> pts;
    for (const Keypoint &k: s->kps)
        if (k.is3d) pts.insert(pts.end(), k.X, k.X + 3);
    const int n = (int) (pts.size() / 3);
    if (n < 32) return 0;  // :181
    if (hipSetDevice(s->device) != hipSuccess) return 0;
    hipStream_t st = (hipStream_t) alva_ctx_stream(s->ctx);
    if (hipMemcpyAsync(s->d_wpt, pts.data(), pts.size() * 8, hipMemcpyHostToDevice, st) != hipSuccess) return 0;
    int found = 0;
    // the reference seeds a fresh generator from std::random_device in every iteration (:203)
    if (alva_find_plane(s->ctx, s->d_wpt, n, s->pose, num_iterations, 1, 0u, nullptr, h_pose, &found) != ALVA_OK) return 0;
    return found ? 1 : 0;
}

extern "C" int alva_system_get_frame_points(alva_system *s, int *h_points) {
    if (!s || !h_points) return 0;
    int n2d = 0, written = 0;
    for (const Keypoint &k: s->kps)
        if (!k.is3d) {
            if (written < 2048) {
                h_points[2 * written] = (int) k.px;      // truncation like `(int) p.x` (system.cpp:150-151)
                h_points[2 * written + 1] = (int) k.py;
                written++;
            }
            n2d++;
        }
    return n2d;
}

extern "C" int alva_system_get_keypoints(alva_system *s, int *h_ids, float *h_px, uint8_t *h_is3d, int cap) {
    if (!s) return 0;
    const int n = (int) std::min<size_t>(s->kps.size(), (size_t) std::max(cap, 0));
    for (int i = 0; i < n; i++) {
        if (h_ids) h_ids[i] = s->kps[(size_t) i].id;
        if (h_px) {
            h_px[2 * i] = s->kps[(size_t) i].px;
            h_px[2 * i + 1] = s->kps[(size_t) i].py;
        }
        if (h_is3d) h_is3d[i] = s->kps[(size_t) i].is3d;
    }
    return (int) s->kps.size();
}

extern "C" int alva_system_set_map_points(alva_system *s, const int *h_ids, const double *h_xyz, int n) {
    if (!s || !h_ids || !h_xyz) return ALVA_ERR_ARG;
    std::unordered_map<int, size_t> byid;
    for (size_t i = 0; i < s->kps.size(); i++) byid[s->kps[i].id] = i;
    int m = 0;
    for (int k = 0; k < n; k++) {
        auto it = byid.find(h_ids[k]);
        if (it == byid.end()) continue;
        Keypoint &kp = s->kps[it->second];
        kp.is3d = true;
        for (int c = 0; c < 3; c++) kp.X[c] = h_xyz[3 * k + c];
        m++;
    }
    if (m) s->ready = s->external_map = true;  // a host-fed map replaces the two-view initialisation
    return m;
}

extern "C" int alva_system_set_pose(alva_system *s, const double *h_pose7) {
    if (!s || !h_pose7) return ALVA_ERR_ARG;
    memcpy(s->pose, h_pose7, 7 * sizeof(double));
    return ALVA_OK;
}
