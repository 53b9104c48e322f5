// The numeric stages of the map layer (slam/stages.hpp) on the GPU: every method is one or more calls through
// include/alvaar_hip.h on this object's HIP stream.  This is the only `Stages` implementation in libalvaar_hip.so; there is no
// CPU path -- creation fails when no HIP device is present.
//
// Host arrays cross into device memory through two bump arenas that are reset per call: a pinned host arena (staging both
// ways, so every copy is asynchronous on the stream) and a device arena.  One stream synchronisation per method.
#include "common.hpp"
#include "stages_hip.hpp"
#include <cmath>

namespace alva_slam {

namespace {

// Frame::computeKeypoint's second half (frame.cpp:109-112): bv = normalised K^-1 (unpx, 1), Eigen's operation order
__global__ void __launch_bounds__(256) k_bearing(const float *unpx, int n, const double *invK, double *bv) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const double u = (double) unpx[2 * i], v = (double) unpx[2 * i + 1];
    const double b0 = (invK[0] * u + invK[1] * v) + invK[2] * 1.;
    const double b1 = (invK[3] * u + invK[4] * v) + invK[5] * 1.;
    const double b2 = (invK[6] * u + invK[7] * v) + invK[8] * 1.;
    const double z = (b0 * b0 + b1 * b1) + b2 * b2;
    if (z > 0.) {
        const double s = sqrt(z);
        bv[3 * i] = b0 / s; bv[3 * i + 1] = b1 / s; bv[3 * i + 2] = b2 / s;
    } else {
        bv[3 * i] = b0; bv[3 * i + 1] = b1; bv[3 * i + 2] = b2;
    }
}

struct Arena {
    uint8_t *base = nullptr;
    size_t cap = 0, used = 0;
    bool pinned = false;
    int grow(size_t need, hipStream_t st) {
        if (need <= cap) return ALVA_OK;
        if (base) {
            ALVA_HIP(hipStreamSynchronize(st));
            if (pinned) ALVA_HIP(hipHostFree(base));
            else ALVA_HIP(hipFree(base));
            base = nullptr;
        }
        size_t c = cap ? cap : (size_t) 1 << 20;
        while (c < need) c *= 2;
        if (pinned) ALVA_HIP(hipHostMalloc((void **) &base, c, hipHostMallocDefault));
        else ALVA_HIP(hipMalloc((void **) &base, c));
        cap = c;
        return ALVA_OK;
    }
    void release() {
        if (base) {
            if (pinned) (void) hipHostFree(base);
            else (void) hipFree(base);
        }
        base = nullptr;
        cap = used = 0;
    }
};

}  // namespace

struct HipStages::Impl {
    int device = 0;
    alva_ctx *ctx = nullptr;
    hipStream_t st = nullptr;
    Camera cam;
    bool clahe = false;
    alva_pyramid *pyr[2] = {nullptr, nullptr};
    int cur = 0;
    uint8_t *d_rgba = nullptr, *d_gray = nullptr, *d_eq = nullptr, *h_rgba = nullptr;
    double *d_invK = nullptr;
    double max_quality = 0.001;  // state.hpp:57; lives as long as the reference's FeatureExtractor object (system.cpp:31)
    Arena dev, pin;
    // a call plans its buffers first (sizes), then the arenas are grown once and carved
    struct Plan {
        std::vector<size_t> sizes;
        size_t add(size_t bytes) {
            sizes.push_back((bytes + 255) / 256 * 256);
            return sizes.size() - 1;
        }
    };
    int carve(Plan &p, std::vector<uint8_t *> &d, std::vector<uint8_t *> &h) {
        size_t total = 0;
        for (size_t s: p.sizes) total += s;
        int rc = dev.grow(total, st);
        if (rc) return rc;
        rc = pin.grow(total, st);
        if (rc) return rc;
        d.resize(p.sizes.size());
        h.resize(p.sizes.size());
        size_t off = 0;
        for (size_t i = 0; i < p.sizes.size(); i++) {
            d[i] = dev.base + off;
            h[i] = pin.base + off;
            off += p.sizes[i];
        }
        return ALVA_OK;
    }
};

#define UP(i, src, bytes)                                                                         \
    do {                                                                                          \
        if ((bytes) > 0) {                                                                        \
            memcpy(h[i], (src), (bytes));                                                         \
            ALVA_HIP(hipMemcpyAsync(d[i], h[i], (bytes), hipMemcpyHostToDevice, m->st));          \
        }                                                                                         \
    } while (0)
#define DOWN(i, bytes)                                                                            \
    do {                                                                                          \
        if ((bytes) > 0) ALVA_HIP(hipMemcpyAsync(h[i], d[i], (bytes), hipMemcpyDeviceToHost, m->st)); \
    } while (0)

HipStages::HipStages() : m(new Impl()) {}

HipStages::~HipStages() {
    if (!m) return;
    (void) hipSetDevice(m->device);
    if (m->ctx) (void) alva_ctx_sync(m->ctx);
    for (auto &p: m->pyr) alva_pyramid_destroy(p);
    void *bufs[] = {m->d_rgba, m->d_gray, m->d_eq, m->d_invK};
    for (void *b: bufs)
        if (b) (void) hipFree(b);
    if (m->h_rgba) (void) hipHostFree(m->h_rgba);
    m->dev.release();
    m->pin.release();
    alva_ctx_destroy(m->ctx);
    delete m;
}

int HipStages::init(int device, const Camera &cam, bool clahe, const double *invK) {
    m->device = device;
    m->cam = cam;
    m->clahe = clahe;
    m->pin.pinned = true;
    int rc = alva_ctx_create(device, nullptr, 1, &m->ctx);
    if (rc) return rc;
    m->st = (hipStream_t) alva_ctx_stream(m->ctx);
    const size_t P = (size_t) cam.width * cam.height;
    ALVA_HIP(hipMalloc((void **) &m->d_rgba, P * 4));
    ALVA_HIP(hipMalloc((void **) &m->d_gray, P));
    if (clahe) ALVA_HIP(hipMalloc((void **) &m->d_eq, P));
    ALVA_HIP(hipMalloc((void **) &m->d_invK, 9 * sizeof(double)));
    ALVA_HIP(hipMemcpy(m->d_invK, invK, 9 * sizeof(double), hipMemcpyHostToDevice));
    ALVA_HIP(hipHostMalloc((void **) &m->h_rgba, P * 4, hipHostMallocDefault));
    for (auto &p: m->pyr) {
        rc = alva_pyramid_create(m->ctx, cam.width, cam.height, 9, 3, &p);  // state.hpp:51-53: 9 x 9 window, 3 levels
        if (rc) return rc;
    }
    return ALVA_OK;
}

int HipStages::new_frame(const uint8_t *rgba) {
    ALVA_HIP(hipSetDevice(m->device));
    const size_t P = (size_t) m->cam.width * m->cam.height;
    memcpy(m->h_rgba, rgba, P * 4);  // the caller's buffer is pageable (wasm-heap style) memory
    ALVA_HIP(hipMemcpyAsync(m->d_rgba, m->h_rgba, P * 4, hipMemcpyHostToDevice, m->st));
    m->cur ^= 1;
    if (!m->clahe) return alva_pyramid_build_from_rgba(m->ctx, m->pyr[m->cur], m->d_rgba, (size_t) m->cam.width * 4, m->d_gray, (size_t) m->cam.width);
    int rc = alva_rgba2gray(m->ctx, m->d_rgba, (size_t) m->cam.width * 4, m->cam.width, m->cam.height, m->d_gray, (size_t) m->cam.width);
    if (rc) return rc;
    // visual_frontend.cpp:16-18: clip limit 3, grid = image size / 50 (state.hpp:45-46)
    rc = alva_clahe(m->ctx, m->d_gray, (size_t) m->cam.width, m->cam.width, m->cam.height, 3.0, m->cam.width / 50, m->cam.height / 50, m->d_eq,
                    (size_t) m->cam.width);
    if (rc) return rc;
    return alva_pyramid_build_from_gray(m->ctx, m->pyr[m->cur], m->d_eq, (size_t) m->cam.width);
}

void HipStages::reset_images() {}  // the pyramids are rebuilt before they are read again (frame 0 tracks nothing)

int HipStages::fbklt(int levels, int n, const float *pts, float *prior, uint8_t *status) {
    if (n <= 0) return ALVA_OK;
    Impl::Plan p;
    const size_t a = p.add((size_t) n * 8), b = p.add((size_t) n * 8), c = p.add((size_t) n);
    std::vector<uint8_t *> d, h;
    int rc = m->carve(p, d, h);
    if (rc) return rc;
    UP(a, pts, (size_t) n * 8);
    UP(b, prior, (size_t) n * 8);
    // state.hpp:50-56: kltError_ 30, kltMaxFbDistance_ 0.5, 30 iterations, 0.01 px
    rc = alva_fbklt_track(m->ctx, m->pyr[m->cur ^ 1], m->pyr[m->cur], levels, 30.f, 0.5f, 30, 0.01f, (const float *) d[a], (float *) d[b], d[c], n);
    if (rc) return rc;
    DOWN(b, (size_t) n * 8);
    DOWN(c, (size_t) n);
    ALVA_HIP(hipStreamSynchronize(m->st));
    memcpy(prior, h[b], (size_t) n * 8);
    memcpy(status, h[c], (size_t) n);
    return ALVA_OK;
}

int HipStages::compute_keypoints(int n, const float *px, float *unpx, double *bv) {
    if (n <= 0) return ALVA_OK;
    Impl::Plan p;
    const size_t a = p.add((size_t) n * 8), b = p.add((size_t) n * 8), c = p.add((size_t) n * 24);
    std::vector<uint8_t *> d, h;
    int rc = m->carve(p, d, h);
    if (rc) return rc;
    UP(a, px, (size_t) n * 8);
    const Camera &k = m->cam;
    rc = alva_undistort_points(m->ctx, (const float *) d[a], n, k.fx, k.fy, k.cx, k.cy, k.k1, k.k2, k.p1, k.p2, (float *) d[b]);
    if (rc) return rc;
    hipLaunchKernelGGL(k_bearing, dim3(alva_divup(n, 256)), dim3(256), 0, m->st, (const float *) d[b], n, m->d_invK, (double *) d[c]);
    ALVA_LAUNCH_CHECK();
    DOWN(b, (size_t) n * 8);
    DOWN(c, (size_t) n * 24);
    ALVA_HIP(hipStreamSynchronize(m->st));
    memcpy(unpx, h[b], (size_t) n * 8);
    memcpy(bv, h[c], (size_t) n * 24);
    return ALVA_OK;
}

int HipStages::project_dist(int n, const double *cam_pts, float *px) {
    if (n <= 0) return ALVA_OK;
    Impl::Plan p;
    const size_t a = p.add((size_t) n * 24), b = p.add((size_t) n * 8);
    std::vector<uint8_t *> d, h;
    int rc = m->carve(p, d, h);
    if (rc) return rc;
    UP(a, cam_pts, (size_t) n * 24);
    const Camera &k = m->cam;
    rc = alva_project_dist(m->ctx, (const double *) d[a], n, k.fx, k.fy, k.cx, k.cy, k.k1, k.k2, k.p1, k.p2, (float *) d[b]);
    if (rc) return rc;
    DOWN(b, (size_t) n * 8);
    ALVA_HIP(hipStreamSynchronize(m->st));
    memcpy(px, h[b], (size_t) n * 8);
    return ALVA_OK;
}

int HipStages::p3p(int n, const double *bv, const double *wpt, int do_random, double *pose7, int *outliers, int *n_outliers, int *ok) {
    *ok = 0;
    *n_outliers = 0;
    if (n < 4) return ALVA_OK;  // multi_view_geometry.cpp:40-43
    // the LMedS median lives in LDS: at most 7168 correspondences per call; a larger frame is solved on its first 7168 keypoints
    // (container order) and the rest is left to the PnP's chi2 sweep
    const int nn = n > 7168 ? 7168 : n;
    Impl::Plan p;
    const size_t a = p.add((size_t) nn * 24), b = p.add((size_t) nn * 24);
    std::vector<uint8_t *> d, h;
    int rc = m->carve(p, d, h);
    if (rc) return rc;
    UP(a, bv, (size_t) nn * 24);
    UP(b, wpt, (size_t) nn * 24);
    double R[9], t[3];
    rc = alva_p3p_lmeds(m->ctx, (const double *) d[a], (const double *) d[b], nn, 100, 3.0f, do_random, 12345u, (float) m->cam.fx, (float) m->cam.fy,
                        R, t, outliers, n_outliers, ok);  // state.hpp:68-69
    if (rc) return rc;
    if (*ok) {
        SE3 T;
        rot_to_quat(R, T.q);  // Twc.setRotationMatrix (multi_view_geometry.cpp:104-105)
        for (int i = 0; i < 3; i++) T.t[i] = t[i];
        se3_to_pose7(T, pose7);
    } else {
        *n_outliers = 0;
    }
    return ALVA_OK;
}

int HipStages::pnp(int n, const double *unpx_d, const double *wpt, double *pose7, int *outliers, int *n_outliers, int *ok) {
    *ok = 0;
    *n_outliers = 0;
    if (n <= 0) return ALVA_OK;
    Impl::Plan p;
    const size_t a = p.add((size_t) n * 16), b = p.add((size_t) n * 24);
    std::vector<uint8_t *> d, h;
    int rc = m->carve(p, d, h);
    if (rc) return rc;
    UP(a, unpx_d, (size_t) n * 16);
    UP(b, wpt, (size_t) n * 24);
    double info[8];
    const Camera &k = m->cam;
    // visual_frontend.cpp:363-375: 5 iterations, robustCostThreshold_ 5.9915, robust + L2 refinement
    return alva_pnp_refine(m->ctx, (const double *) d[a], (const double *) d[b], n, pose7, 5, 5.9915f, 1, 1, (float) k.fx, (float) k.fy, (float) k.cx,
                           (float) k.cy, outliers, n_outliers, info, ok);
}

int HipStages::five_point(int n, const double *bv_kf, const double *bv_cur, int do_random, double *R, double *t, int *outliers, int *n_outliers,
                          int *ok) {
    *ok = 0;
    *n_outliers = 0;
    if (n < 8) return ALVA_OK;
    Impl::Plan p;
    const size_t a = p.add((size_t) n * 24), b = p.add((size_t) n * 24);
    std::vector<uint8_t *> d, h;
    int rc = m->carve(p, d, h);
    if (rc) return rc;
    UP(a, bv_kf, (size_t) n * 24);
    UP(b, bv_cur, (size_t) n * 24);
    std::vector<uint8_t> inl((size_t) n);
    rc = alva_compute_5pt_essential(m->ctx, (const double *) d[a], (const double *) d[b], n, 100, 3.0f, 1, do_random, 12345u, (float) m->cam.fx,
                                    (float) m->cam.fy, R, t, inl.data(), nullptr, ok);
    if (rc) return rc;
    if (*ok)
        for (int i = 0; i < n; i++)
            if (!inl[(size_t) i]) outliers[(*n_outliers)++] = i;
    return ALVA_OK;
}

int HipStages::detect(int cell, int n_occ, const float *occupied, int cap, float *pts, int *count) {
    Impl::Plan p;
    const size_t a = p.add((size_t) (n_occ > 0 ? n_occ : 1) * 8), b = p.add((size_t) cap * 8);
    std::vector<uint8_t *> d, h;
    int rc = m->carve(p, d, h);
    if (rc) return rc;
    UP(a, occupied, (size_t) n_occ * 8);
    const Camera &k = m->cam;
    const uint8_t *img = m->clahe ? m->d_eq : m->d_gray;  // detection runs on currImage_ (map_manager.cpp:213)
    // roi = CameraCalibration::roi_rect_ (camera_calibration.cpp:20): the image minus a border of 20 px
    rc = alva_detect_grid(m->ctx, img, (size_t) k.width, k.width, k.height, cell, (const float *) d[a], n_occ, k.border, k.border,
                          k.width - 2 * k.border, k.height - 2 * k.border, &m->max_quality, (float *) d[b], cap, count);
    if (rc) return rc;
    if (*count > cap) *count = cap;
    if (*count > 0) {
        DOWN(b, (size_t) *count * 8);
        ALVA_HIP(hipStreamSynchronize(m->st));
        memcpy(pts, h[b], (size_t) *count * 8);
    }
    return ALVA_OK;
}

int HipStages::describe(int n, const float *pts, uint8_t *desc, uint8_t *valid) {
    if (n <= 0) return ALVA_OK;
    Impl::Plan p;
    const size_t a = p.add((size_t) n * 8), b = p.add((size_t) n * 32), c = p.add((size_t) n);
    std::vector<uint8_t *> d, h;
    int rc = m->carve(p, d, h);
    if (rc) return rc;
    UP(a, pts, (size_t) n * 8);
    rc = alva_describe(m->ctx, m->d_gray, (size_t) m->cam.width, m->cam.width, m->cam.height, (const float *) d[a], n, d[b], d[c]);
    if (rc) return rc;
    DOWN(b, (size_t) n * 32);
    DOWN(c, (size_t) n);
    ALVA_HIP(hipStreamSynchronize(m->st));
    memcpy(desc, h[b], (size_t) n * 32);
    memcpy(valid, h[c], (size_t) n);
    return ALVA_OK;
}

int HipStages::triangulate(int n, int n_groups, const double *T36, const int *group, const double *bv_l, const double *bv_r, const float *unpx_l,
                           const float *unpx_r, double *wpt, double *inv_depth, uint8_t *status, double *parallax) {
    if (n <= 0) return ALVA_OK;
    Impl::Plan p;
    const size_t iT = p.add((size_t) n_groups * 288), ig = p.add((size_t) n * 4), il = p.add((size_t) n * 24), ir = p.add((size_t) n * 24),
                 iul = p.add((size_t) n * 8), iur = p.add((size_t) n * 8), ilp = p.add((size_t) n * 24), iw = p.add((size_t) n * 24),
                 iid = p.add((size_t) n * 8), ist = p.add((size_t) n), ipar = p.add((size_t) n * 8);
    std::vector<uint8_t *> d, h;
    int rc = m->carve(p, d, h);
    if (rc) return rc;
    UP(iT, T36, (size_t) n_groups * 288);
    UP(ig, group, (size_t) n * 4);
    UP(il, bv_l, (size_t) n * 24);
    UP(ir, bv_r, (size_t) n * 24);
    UP(iul, unpx_l, (size_t) n * 8);
    UP(iur, unpx_r, (size_t) n * 8);
    const Camera &k = m->cam;
    rc = alva_triangulate(m->ctx, n, (const double *) d[iT], n_groups, (const int *) d[ig], (const double *) d[il], (const double *) d[ir],
                          (const float *) d[iul], (const float *) d[iur], k.fx, k.fy, k.cx, k.cy, 3.0f /* mapMaxReprojectionError_ */,
                          (double *) d[ilp], (double *) d[iw], (double *) d[iid], d[ist], (double *) d[ipar]);
    if (rc) return rc;
    DOWN(iw, (size_t) n * 24);
    DOWN(iid, (size_t) n * 8);
    DOWN(ist, (size_t) n);
    DOWN(ipar, (size_t) n * 8);
    ALVA_HIP(hipStreamSynchronize(m->st));
    memcpy(wpt, h[iw], (size_t) n * 24);
    memcpy(inv_depth, h[iid], (size_t) n * 8);
    memcpy(status, h[ist], (size_t) n);
    memcpy(parallax, h[ipar], (size_t) n * 8);
    return ALVA_OK;
}

int HipStages::match_to_map(int cell_size, int num_cells_w, int grid_cells, const int *cell_ptr, const int *cell_mp, int n_kf, const double *kf_q,
                            const double *kf_t, int n_mp, const double *mp_wpt, const uint8_t *mp_is3d, const uint8_t *mp_has_desc,
                            const int *obs_ptr, const int *obs_kf, const float *obs_px, const uint8_t *obs_desc, const uint8_t *obs_has_desc,
                            int frame_kf, int num_keypoints_3d, int n_local, const int *local, float max_proj_err, float dist_ratio,
                            int *match_of_mp) {
    if (n_mp <= 0) return ALVA_OK;
    const int n_obs = obs_ptr[n_mp], n_cell = cell_ptr[grid_cells];
    Impl::Plan p;
    const size_t icp = p.add((size_t) (grid_cells + 1) * 4), icm = p.add((size_t) (n_cell > 0 ? n_cell : 1) * 4), iq = p.add((size_t) n_kf * 32),
                 it = p.add((size_t) n_kf * 24), iw = p.add((size_t) n_mp * 24), i3 = p.add((size_t) n_mp), ihd = p.add((size_t) n_mp),
                 iop = p.add((size_t) (n_mp + 1) * 4), iok = p.add((size_t) (n_obs > 0 ? n_obs : 1) * 4),
                 iox = p.add((size_t) (n_obs > 0 ? n_obs : 1) * 8), iod = p.add((size_t) (n_obs > 0 ? n_obs : 1) * 32),
                 ioh = p.add((size_t) (n_obs > 0 ? n_obs : 1)), il = p.add((size_t) (n_local > 0 ? n_local : 1) * 4), im = p.add((size_t) n_mp * 4);
    std::vector<uint8_t *> d, h;
    int rc = m->carve(p, d, h);
    if (rc) return rc;
    UP(icp, cell_ptr, (size_t) (grid_cells + 1) * 4);
    UP(icm, cell_mp, (size_t) n_cell * 4);
    UP(iq, kf_q, (size_t) n_kf * 32);
    UP(it, kf_t, (size_t) n_kf * 24);
    UP(iw, mp_wpt, (size_t) n_mp * 24);
    UP(i3, mp_is3d, (size_t) n_mp);
    UP(ihd, mp_has_desc, (size_t) n_mp);
    UP(iop, obs_ptr, (size_t) (n_mp + 1) * 4);
    UP(iok, obs_kf, (size_t) n_obs * 4);
    UP(iox, obs_px, (size_t) n_obs * 8);
    UP(iod, obs_desc, (size_t) n_obs * 32);
    UP(ioh, obs_has_desc, (size_t) n_obs);
    UP(il, local, (size_t) n_local * 4);
    const Camera &k = m->cam;
    const double calib[10] = {k.fx, k.fy, k.cx, k.cy, k.k1, k.k2, k.p1, k.p2, (double) k.width, (double) k.height};
    rc = alva_match_to_map_flags(m->ctx, calib, cell_size, num_cells_w, grid_cells, (const int *) d[icp], (const int *) d[icm], n_kf,
                                 (const double *) d[iq], (const double *) d[it], n_mp, (const double *) d[iw], d[i3], d[ihd], (const int *) d[iop],
                                 (const int *) d[iok], (const float *) d[iox], d[iod], d[ioh], frame_kf, num_keypoints_3d, n_local,
                                 (const int *) d[il], max_proj_err, dist_ratio, (int *) d[im]);
    if (rc) return rc;
    DOWN(im, (size_t) n_mp * 4);
    ALVA_HIP(hipStreamSynchronize(m->st));
    memcpy(match_of_mp, h[im], (size_t) n_mp * 4);
    return ALVA_OK;
}

int HipStages::local_ba(int n_kf, double *poses7, const uint8_t *kf_const, int n_pt, const int *pt_anchor_kf, const double *pt_anchor_uv,
                        double *pt_inv_depth, int n_obs, const int *obs_kf, const int *obs_pt, const double *obs_uv, int max_iters, double *chi2,
                        uint8_t *depth_pos) {
    const Camera &k = m->cam;
    const double calib[4] = {k.fx, k.fy, k.cx, k.cy};
    double info[4];
    int ok = 0;
    // optimizer.cpp:251-262: function tolerance 1e-3, Huber on sqrt(robustCostThreshold_) -- a float in the reference (:8, :22)
    return alva_local_ba(m->ctx, n_kf, poses7, kf_const, calib, 1, n_pt, pt_anchor_kf, pt_anchor_uv, pt_inv_depth, n_obs, obs_kf, obs_pt, obs_uv,
                         max_iters, 0.001, (double) 5.9915f, chi2, depth_pos, info, &ok);
}

int HipStages::find_plane(int n, const double *pts, const double *pose7_twc, int iterations, float *pose16, int *found) {
    *found = 0;
    if (n < 32) return ALVA_OK;  // system.cpp:181
    Impl::Plan p;
    const size_t a = p.add((size_t) n * 24);
    std::vector<uint8_t *> d, h;
    int rc = m->carve(p, d, h);
    if (rc) return rc;
    UP(a, pts, (size_t) n * 24);
    // the reference seeds a fresh generator from std::random_device in every iteration (system.cpp:210)
    return alva_find_plane(m->ctx, (const double *) d[a], n, pose7_twc, iterations, 1, 0u, nullptr, pose16, found);
}

}  // namespace alva_slam
