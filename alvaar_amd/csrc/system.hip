// a1 / §8(b): the reference's `System` surface over the HIP hot path (host orchestration only -- every numeric stage is
// a call through include/alvaar_hip.h).
//
// Mirrors, in reduced form, System::findCameraPose -> VisualFrontend::track/process (src/slam/src/system.cpp:106-175,
// visual_frontend.cpp:21-101): per frame  upload RGBA -> gray + LK pyramid (fused) -> forward-backward KLT of the frame's
// keypoints (visual_frontend.cpp:103-243; every keypoint uses its previous position as prior, 3 levels) -> P3P-LMedS +
// robust PnP on the keypoints that carry a 3-D map point (:245-417) -> constant-velocity motion model update; and on a
// keyframe (:554-594, here: fewer than half of the cells still tracked)  MapManager::extractKeypoints
// (map_manager.cpp:193-222): grid detection in the unoccupied cells + ORB description.
// NOT mirrored yet: initialisation (5-pt essential matrix), triangulation, map matching,
// local-BA scheduling, plane fitting -- the host graph logic around them is the reference's L2 layer.
#include "common.hpp"
#include "lm_device.hpp"
#include "../../include/alvaar_system.h"
#include <algorithm>
#include <cmath>
#include <unordered_map>

static thread_local char g_sys_err[256] = "";
extern "C" const char *alva_system_last_error(void) { return g_sys_err[0] ? g_sys_err : alva_last_error(); }

namespace {
struct Keypoint {
    int id;
    float px, py;
    bool is3d;
    double X[3];
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
    void *bufs[] = {s->d_rgba, s->d_gray, s->d_desc, s->d_status, s->d_valid, s->d_pts, s->d_prior, s->d_new, s->d_bv, s->d_wpt, s->d_uv};
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
    int status = 3;
    // ---- KLT tracking of the frame's keypoints (visual_frontend.cpp:103-243) -------------------------------------
    if (s->have_prev && !s->kps.empty()) {
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
    // ---- pose from the 3-D keypoints (visual_frontend.cpp:245-417) ------------------------------------------------
    std::vector<int> idx3d;
    for (size_t i = 0; i < s->kps.size(); i++)
        if (s->kps[i].is3d) idx3d.push_back((int) i);
    if (s->have_prev && idx3d.size() >= 4) {
        const int n = (int) idx3d.size();
        std::vector<double> bv((size_t) n * 3), wp((size_t) n * 3), uv((size_t) n * 2);
        for (int k = 0; k < n; k++) {
            const Keypoint &kp = s->kps[(size_t) idx3d[(size_t) k]];
            const double x = (kp.px - s->cx) / s->fx, y = (kp.py - s->cy) / s->fy, nn = std::sqrt(x * x + y * y + 1.0);
            bv[3 * (size_t) k] = x / nn; bv[3 * (size_t) k + 1] = y / nn; bv[3 * (size_t) k + 2] = 1.0 / nn;
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
        const bool good = pstat == 2;
        if (good) {
            // remove the observations P3P and ceresPnP flagged (visual_frontend.cpp:344-352, :411-414)
            std::vector<uint8_t> drop(s->kps.size(), 0);
            for (int k = 0; k < n; k++)
                if (outP3p[(size_t) k] || outPnp[(size_t) k]) drop[(size_t) idx3d[(size_t) k]] = 1;
            std::vector<Keypoint> kept;
            for (size_t i = 0; i < s->kps.size(); i++)
                if (!drop[i]) kept.push_back(s->kps[i]);
            s->kps.swap(kept);
        }
        if (good) {
            memcpy(s->pose, pose7, sizeof(pose7));
            s->pose_failures = 0;
            status = 1;
        } else if (++s->pose_failures > 3) {   // visual_frontend.cpp:86-92 -> System::reset, status 2
            alva_system_reset(s);
            pose_to_array(s->pose, h_pose);
            return 2;
        }
    }
    // ---- keyframe: extract new keypoints when too few cells are still tracked ---------------------------------------
    const int cells = (s->w / s->cell) * (s->h / s->cell);
    if (!s->have_prev || (int) s->kps.size() < cells / 2) {
        rc = extract_keypoints(s);
        if (rc) return rc;
    }
    s->have_prev = true;
    pose_to_array(s->pose, h_pose);
    return status;
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
    (void) h_pose;
    (void) num_iterations;
    if (!s) return 0;
    // System::processPlane (system.cpp:177-342) is SURVEY.md §8f row 3 ("next"); until it lands the call reports
    // "no plane" exactly like the reference does with fewer than 32 observed 3-D points (:181,269).
    return 0;
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
    return m;
}

extern "C" int alva_system_set_pose(alva_system *s, const double *h_pose7) {
    if (!s || !h_pose7) return ALVA_ERR_ARG;
    memcpy(s->pose, h_pose7, 7 * sizeof(double));
    return ALVA_OK;
}
