// TEST INFRASTRUCTURE ONLY (part of oracle/_ref/libalva_ref.so) -- never linked into, shipped with or loaded by alvaar_amd/.
//
// syscpu_*: the product's host-side map layer (alvaar_amd/csrc/slam/*.cpp -- the very sources libalvaar_hip.so compiles) with
// its numeric stages provided by the REFERENCE's own L1 functions (FeatureTracker, FeatureExtractor, MultiViewGeometry,
// CameraCalibration and the vendored OpenCV / OpenGV / Ceres under them) instead of the HIP kernels.  Purpose: on a machine
// without a GPU, check the host logic -- keyframe policy, map bookkeeping, container orders, BA graph construction, write-back,
// culling -- against the reference's System (ref_system_*, ref_shim_system.cpp) frame by frame.  With identical stage
// arithmetic underneath, any difference is a bookkeeping difference.  The GPU tests then run the same map layer over the HIP
// stages (alva_system_*) against the same reference.
// Two stages have no stand-alone reference entry point and use the plain-C restatements that tests/test_oracle_vs_ref.py pins
// to the reference: orc_match_to_map_flags (== Mapper::matchToMap on flattened maps) and orc_find_plane.
#include <sys/time.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <sstream>
#include <string>
#include <vector>
#include <Eigen/Core>
#include <Eigen/Geometry>
#include <sophus/se3.hpp>
#include <opencv2/core.hpp>
#include <opencv2/imgproc.hpp>
#include <opencv2/video/tracking.hpp>
#include "frame.hpp"
#include "feature_tracker.hpp"
#include "feature_extractor.hpp"
#include "multi_view_geometry.hpp"
#include "camera_calibration.hpp"
#include "alva_oracle.h"
#include "slam/slam.hpp"
#include "slam/inspect.hpp"
#include "slam/stage_trace.hpp"

extern "C" int ref_local_ba(int nKf, double *poses, const uint8_t *kfConst, const double *calib, int invDepth, int nPt, const int *ptAnchorKf,
                            const double *ptAnchorUv, double *ptParam, int nObs, const int *obsKf, const int *obsPt, const double *obsUv,
                            int maxIterations, double functionTolerance, double huberChi2, double *chi2, uint8_t *depthPos, double *info);

namespace {
using namespace alva_slam;
typedef std::vector<Eigen::Vector3d, Eigen::aligned_allocator<Eigen::Vector3d>> V3;
typedef std::vector<Eigen::Vector2d, Eigen::aligned_allocator<Eigen::Vector2d>> V2;

Sophus::SE3d se3_of(const double *p) { return Sophus::SE3d(Eigen::Quaterniond(p[6], p[3], p[4], p[5]), Eigen::Vector3d(p[0], p[1], p[2])); }
void pose7_of(const Sophus::SE3d &T, double *p) {
    const Eigen::Quaterniond q = T.unit_quaternion();
    for (int i = 0; i < 3; i++) p[i] = T.translation()(i);
    p[3] = q.x(); p[4] = q.y(); p[5] = q.z(); p[6] = q.w();
}

struct RefStages : Stages {
    Camera cam;
    bool clahe = false;
    std::shared_ptr<CameraCalibration> cal;
    FeatureExtractor extractor;
    FeatureTracker tracker;
    cv::Ptr<cv::CLAHE> clahe_op;
    cv::Mat raw, cur, prev;
    std::vector<cv::Mat> cur_pyr, prev_pyr;
    double perturb = 0.;  // syscpu_set_perturbation
    unsigned long long lcg = 88172645463325252ULL;

    RefStages(const Camera &c, bool clahe_on) : cam(c), clahe(clahe_on), extractor(0.001), tracker(30, 0.01f) {  // state.hpp:55-57
        cal = std::make_shared<CameraCalibration>(c.fx, c.fy, c.cx, c.cy, c.k1, c.k2, c.p1, c.p2, c.width, c.height, c.border);
        clahe_op = cv::createCLAHE(3, cv::Size(c.width / 50, c.height / 50));  // visual_frontend.cpp:16-18, state.hpp:45-46
    }
    int new_frame(const uint8_t *rgba) override {
        cv::Mat image(cam.height, cam.width, CV_8UC4, const_cast<uint8_t *>(rgba));
        cv::cvtColor(image, raw, cv::COLOR_RGBA2GRAY);
        cv::swap(cur, prev);
        if (clahe) clahe_op->apply(raw, cur);
        else cur = raw;
        if (!cur_pyr.empty()) prev_pyr.swap(cur_pyr);
        cv::buildOpticalFlowPyramid(cur, cur_pyr, cv::Size(9, 9), 3);
        return 0;
    }
    void reset_images() override {
        cur.release();
        prev.release();
        cur_pyr.clear();
        prev_pyr.clear();
    }
    int fbklt(int levels, int n, const float *pts, float *prior, uint8_t *status) override {
        std::vector<cv::Point2f> vp((size_t) n), vq((size_t) n);
        for (int i = 0; i < n; i++) {
            vp[(size_t) i] = cv::Point2f(pts[2 * i], pts[2 * i + 1]);
            vq[(size_t) i] = cv::Point2f(prior[2 * i], prior[2 * i + 1]);
        }
        std::vector<bool> st;
        tracker.fbKltTracking(prev_pyr, cur_pyr, 9, levels, 30, 0.5f, vp, vq, st);  // state.hpp:50-54
        for (int i = 0; i < n; i++) {
            prior[2 * i] = vq[(size_t) i].x;
            prior[2 * i + 1] = vq[(size_t) i].y;
            status[i] = st[(size_t) i] ? 1 : 0;
        }
        return 0;
    }
    int compute_keypoints(int n, const float *px, float *unpx, double *bv) override {
        for (int i = 0; i < n; i++) {
            const cv::Point2f u = cal->undistortImagePoint(cv::Point2f(px[2 * i], px[2 * i + 1]));
            Eigen::Vector3d h(u.x, u.y, 1.);  // Frame::computeKeypoint (frame.cpp:105-113)
            Eigen::Vector3d b = cal->inverseK_ * h;
            b.normalize();
            unpx[2 * i] = u.x; unpx[2 * i + 1] = u.y;
            bv[3 * i] = b(0); bv[3 * i + 1] = b(1); bv[3 * i + 2] = b(2);
            if (perturb > 0) {  // sensitivity probe (tests): move every bearing by ~perturb, deterministically
                for (int c = 0; c < 3; c++) {
                    lcg = lcg * 6364136223846793005ULL + 1442695040888963407ULL;
                    bv[3 * i + c] *= 1.0 + perturb * ((double) (lcg >> 40) / 8388608.0 - 1.0);
                }
            }
        }
        return 0;
    }
    int project_dist(int n, const double *P, float *px) override {
        for (int i = 0; i < n; i++) {
            const cv::Point2f r = cal->projectCamToImageDist(Eigen::Vector3d(P[3 * i], P[3 * i + 1], P[3 * i + 2]));
            px[2 * i] = r.x; px[2 * i + 1] = r.y;
        }
        return 0;
    }
    int p3p(int n, const double *bv, const double *wpt, int do_random, double *pose7, int *outliers, int *n_out, int *ok) override {
        V3 vb((size_t) n), vw((size_t) n);
        for (int i = 0; i < n; i++) {
            vb[(size_t) i] = Eigen::Vector3d(bv[3 * i], bv[3 * i + 1], bv[3 * i + 2]);
            vw[(size_t) i] = Eigen::Vector3d(wpt[3 * i], wpt[3 * i + 1], wpt[3 * i + 2]);
        }
        Sophus::SE3d Twc = se3_of(pose7);
        std::vector<int> out;
        *ok = MultiViewGeometry::p3pRansac(vb, vw, 100, 3.0f, false, do_random != 0, cam.fx, cam.fy, Twc, out) ? 1 : 0;  // state.hpp:68-69
        if (*ok) pose7_of(Twc, pose7);
        *n_out = (int) out.size();
        for (size_t i = 0; i < out.size(); i++) outliers[i] = out[i];
        return 0;
    }
    int pnp(int n, const double *uv, const double *wpt, double *pose7, int *outliers, int *n_out, int *ok) override {
        V2 vk((size_t) n);
        V3 vw((size_t) n);
        for (int i = 0; i < n; i++) {
            vk[(size_t) i] = Eigen::Vector2d(uv[2 * i], uv[2 * i + 1]);
            vw[(size_t) i] = Eigen::Vector3d(wpt[3 * i], wpt[3 * i + 1], wpt[3 * i + 2]);
        }
        Sophus::SE3d Twc = se3_of(pose7);
        std::vector<int> out;
        *ok = MultiViewGeometry::ceresPnP(vk, vw, Twc, 5, 5.9915f, true, true, cam.fx, cam.fy, cam.cx, cam.cy, out) ? 1 : 0;
        pose7_of(Twc, pose7);
        *n_out = (int) out.size();
        for (size_t i = 0; i < out.size(); i++) outliers[i] = out[i];
        return 0;
    }
    int five_point(int n, const double *b1, const double *b2, int do_random, double *R, double *t, int *outliers, int *n_out, int *ok) override {
        V3 v1((size_t) n), v2((size_t) n);
        for (int i = 0; i < n; i++) {
            v1[(size_t) i] = Eigen::Vector3d(b1[3 * i], b1[3 * i + 1], b1[3 * i + 2]);
            v2[(size_t) i] = Eigen::Vector3d(b2[3 * i], b2[3 * i + 1], b2[3 * i + 2]);
        }
        Eigen::Matrix3d Rwc = Eigen::Matrix3d::Identity();
        Eigen::Vector3d twc = Eigen::Vector3d::Zero();
        std::vector<int> out;
        *ok = MultiViewGeometry::compute5ptEssentialMatrix(v1, v2, 100, 3.0f, true, do_random != 0, cam.fx, cam.fy, Rwc, twc, out) ? 1 : 0;
        for (int r = 0; r < 3; r++) {
            for (int c = 0; c < 3; c++) R[3 * r + c] = Rwc(r, c);
            t[r] = twc(r);
        }
        *n_out = (int) out.size();
        for (size_t i = 0; i < out.size(); i++) outliers[i] = out[i];
        return 0;
    }
    int detect(int cell, int n_occ, const float *occ, int cap, float *pts, int *count) override {
        std::vector<cv::Point2f> o((size_t) n_occ);
        for (int i = 0; i < n_occ; i++) o[(size_t) i] = cv::Point2f(occ[2 * i], occ[2 * i + 1]);
        const std::vector<cv::Point2f> r = extractor.detectFeaturePoints(cur, cell, o, cal->roi_rect_);
        *count = (int) std::min<size_t>(r.size(), (size_t) cap);
        for (int i = 0; i < *count; i++) {
            pts[2 * i] = r[(size_t) i].x;
            pts[2 * i + 1] = r[(size_t) i].y;
        }
        return 0;
    }
    int describe(int n, const float *pts, uint8_t *desc, uint8_t *valid) override {
        std::vector<cv::Point2f> v((size_t) n);
        for (int i = 0; i < n; i++) v[(size_t) i] = cv::Point2f(pts[2 * i], pts[2 * i + 1]);
        const std::vector<cv::Mat> d = extractor.describeFeaturePoints(raw, v);
        for (int i = 0; i < n; i++) {
            const bool ok = i < (int) d.size() && !d[(size_t) i].empty();
            valid[i] = ok;
            if (ok) std::memcpy(desc + 32 * (size_t) i, d[(size_t) i].ptr<uint8_t>(0), 32);
            else std::memset(desc + 32 * (size_t) i, 0, 32);
        }
        return 0;
    }
    int triangulate(int n, int, const double *T, const int *group, const double *bvl, const double *bvr, const float *ul, const float *ur,
                    double *wpt, double *inv_depth, uint8_t *status, double *parallax) override {
        std::vector<double> lpt((size_t) n * 3);
        // the restatement pinned to MultiViewGeometry::triangulate + the gates of mapper.cpp:246-280 (tests/test_triangulate.py)
        orc_triangulate(n, T, group, bvl, bvr, ul, ur, cam.fx, cam.fy, cam.cx, cam.cy, 3.0f, lpt.data(), wpt, inv_depth, status, parallax);
        return 0;
    }
    int match_to_map(int cell_size, int ncw, int grid_cells, const int *cell_ptr, const int *cell_mp, int n_kf, const double *kf_q,
                     const double *kf_t, int n_mp, const double *mp_wpt, const uint8_t *mp_is3d, const uint8_t *mp_has_desc, const int *obs_ptr,
                     const int *obs_kf, const float *obs_px, const uint8_t *obs_desc, const uint8_t *obs_has_desc, int frame_kf, int n3d,
                     int n_local, const int *local, float max_proj_err, float dist_ratio, int *match_of_mp) override {
        const double calib[10] = {cam.fx, cam.fy, cam.cx, cam.cy, cam.k1, cam.k2, cam.p1, cam.p2, (double) cam.width, (double) cam.height};
        orc_match_to_map_flags(calib, cell_size, ncw, grid_cells, cell_ptr, cell_mp, n_kf, kf_q, kf_t, n_mp, mp_wpt, mp_is3d, mp_has_desc, obs_ptr,
                               obs_kf, obs_px, obs_desc, obs_has_desc, frame_kf, n3d, n_local, local, max_proj_err, dist_ratio, match_of_mp);
        return 0;
    }
    int local_ba(int n_kf, double *poses7, const uint8_t *kf_const, int n_pt, const int *pt_anchor_kf, const double *pt_anchor_uv,
                 double *pt_inv_depth, int n_obs, const int *obs_kf, const int *obs_pt, const double *obs_uv, int max_iters, double *chi2,
                 uint8_t *depth_pos) override {
        const double calib[4] = {cam.fx, cam.fy, cam.cx, cam.cy};
        ref_local_ba(n_kf, poses7, kf_const, calib, 1, n_pt, pt_anchor_kf, pt_anchor_uv, pt_inv_depth, n_obs, obs_kf, obs_pt, obs_uv, max_iters,
                     0.001, 5.9915f, chi2, depth_pos, nullptr);  // optimizer.cpp:251-262
        return 0;
    }
    int find_plane(int, const double *, const double *, int, float *, int *found) override {
        *found = 0;
        return 0;
    }
};


// A tape of stage RESULTS (test tooling, tools/host_replay_cpu.py): the first run of a stream records what every stage call returned, a
// second Slam over the same frames gets the recorded results back at memcpy cost -- the map layer then runs as it does over a device that
// answers at once, with its own working set in the caches instead of OpenCV's / Ceres's, and its per-section timers (Slam::t_kf / t_fine)
// show the HOST's share of a keyframe on a machine without a GPU.  The host logic is deterministic, so the call sequence of the second
// run equals the first's; every record carries a tag and the sizes it was made with, and a mismatch aborts.
struct Tape {
    std::vector<uint8_t> bytes;
    size_t pos = 0;
};
struct MemoStages : Stages {
    Stages *in_;
    Tape *tape_;
    bool replay_;
    int nest_ = 0;   // inside a composed default (track_begin, match_to_map_rec, local_ba_csr): the fine-grained calls under it just forward
    MemoStages(Stages *inner, Tape *t, bool replay) : in_(inner), tape_(t), replay_(replay) {}
    bool live() const { return nest_ == 0; }
    void tag(uint32_t what, long long a = 0, long long b = 0) {
        if (!live()) return;
        long long rec[3] = {(long long) what, a, b};
        if (!replay_) {
            put(rec, sizeof(rec));
            return;
        }
        long long got[3];
        get(got, sizeof(got));
        if (std::memcmp(rec, got, sizeof(rec))) {
            std::fprintf(stderr, "syscpu tape: call %u (%lld, %lld) where the recording has %lld (%lld, %lld)\n", what, a, b, got[0], got[1], got[2]);
            std::abort();
        }
    }
    void put(const void *p, size_t n) {
        const uint8_t *b = static_cast<const uint8_t *>(p);
        tape_->bytes.insert(tape_->bytes.end(), b, b + n);
    }
    void get(void *p, size_t n) {
        if (tape_->pos + n > tape_->bytes.size()) {
            std::fprintf(stderr, "syscpu tape: read past the end\n");
            std::abort();
        }
        std::memcpy(p, tape_->bytes.data() + tape_->pos, n);
        tape_->pos += n;
    }
    void io(void *p, size_t n) {   // an output array: recorded after the inner call, delivered on replay
        if (!live()) return;
        if (replay_) get(p, n);
        else put(p, n);
    }
    struct Nest {
        MemoStages *m;
        explicit Nest(MemoStages *s) : m(s) { m->nest_++; }
        ~Nest() { m->nest_--; }
    };
    bool skip() const { return replay_ && live(); }   // the inner call is not made

    int track_begin(const TrackJob &job, TrackKlt &out) override {
        tag(1, job.n, job.want_pose);
        int rc = 0;
        if (!skip()) {
            Nest n(this);
            rc = Stages::track_begin(job, out);
        } else {
            out.code.resize((size_t) job.n); out.px.resize((size_t) job.n * 2); out.unpx.resize((size_t) job.n * 2); out.bv.resize((size_t) job.n * 3);
            out.code_v = out.code.data(); out.px_v = out.px.data(); out.unpx_v = out.unpx.data(); out.bv_v = out.bv.data();
        }
        io(&rc, 4);
        io(const_cast<uint8_t *>(out.code_v), (size_t) job.n);
        io(const_cast<float *>(out.px_v), (size_t) job.n * 8);
        io(const_cast<float *>(out.unpx_v), (size_t) job.n * 8);
        io(const_cast<double *>(out.bv_v), (size_t) job.n * 24);
        io(&out.p3p_req, 4);
        io(&out.n_pose, 4);
        n_pose_ = out.n_pose;
        want_pose_ = job.want_pose;
        return rc;
    }
    int track_pose_collect(TrackPose &out) override {
        tag(2, n_pose_, want_pose_);
        int rc = 0;
        if (!skip()) {
            Nest n(this);
            rc = Stages::track_pose_collect(out);
        }
        io(&rc, 4);
        io(&out.status, 4);
        io(out.pose7_p3p, 56);
        io(out.pose7, 56);
        long long sz[2] = {(long long) out.p3p_outlier.size(), (long long) out.pnp_outlier.size()};
        io(sz, 16);
        out.p3p_outlier.resize((size_t) sz[0]);
        out.pnp_outlier.resize((size_t) sz[1]);
        io(out.p3p_outlier.data(), (size_t) sz[0]);
        io(out.pnp_outlier.data(), (size_t) sz[1]);
        return rc;
    }
    int new_frame(const uint8_t *rgba) override { return skip() ? 0 : in_->new_frame(rgba); }
    void reset_images() override {
        if (!skip()) in_->reset_images();
    }
    int fbklt(int levels, int n, const float *pts, float *prior, uint8_t *status) override {
        tag(3, n, levels);
        int rc = skip() ? 0 : in_->fbklt(levels, n, pts, prior, status);
        io(&rc, 4); io(prior, (size_t) n * 8); io(status, (size_t) n);
        return rc;
    }
    int compute_keypoints(int n, const float *px, float *unpx, double *bv) override {
        tag(4, n);
        int rc = skip() ? 0 : in_->compute_keypoints(n, px, unpx, bv);
        io(&rc, 4); io(unpx, (size_t) n * 8); io(bv, (size_t) n * 24);
        return rc;
    }
    int project_dist(int n, const double *cam_pts, float *px) override {
        tag(5, n);
        int rc = skip() ? 0 : in_->project_dist(n, cam_pts, px);
        io(&rc, 4); io(px, (size_t) n * 8);
        return rc;
    }
    int p3p(int n, const double *bv, const double *wpt, int do_random, double *pose7, int *outliers, int *n_outliers, int *ok) override {
        tag(6, n);
        int rc = skip() ? 0 : in_->p3p(n, bv, wpt, do_random, pose7, outliers, n_outliers, ok);
        io(&rc, 4); io(ok, 4); io(pose7, 56); io(n_outliers, 4); io(outliers, (size_t) *n_outliers * 4);
        return rc;
    }
    int pnp(int n, const double *uv, const double *wpt, double *pose7, int *outliers, int *n_outliers, int *ok) override {
        tag(7, n);
        int rc = skip() ? 0 : in_->pnp(n, uv, wpt, pose7, outliers, n_outliers, ok);
        io(&rc, 4); io(ok, 4); io(pose7, 56); io(n_outliers, 4); io(outliers, (size_t) *n_outliers * 4);
        return rc;
    }
    int five_point(int n, const double *b1, const double *b2, int do_random, double *R, double *t, int *outliers, int *n_outliers, int *ok) override {
        tag(8, n);
        int rc = skip() ? 0 : in_->five_point(n, b1, b2, do_random, R, t, outliers, n_outliers, ok);
        io(&rc, 4); io(ok, 4); io(R, 72); io(t, 24); io(n_outliers, 4); io(outliers, (size_t) *n_outliers * 4);
        return rc;
    }
    int detect(int cell, int n_occ, const float *occupied, int cap, float *pts, int *count) override {
        tag(9, n_occ, cap);
        int rc = skip() ? 0 : in_->detect(cell, n_occ, occupied, cap, pts, count);
        io(&rc, 4); io(count, 4); io(pts, (size_t) (*count > 0 ? *count : 0) * 8);
        return rc;
    }
    int describe(int n, const float *pts, uint8_t *desc, uint8_t *valid) override {
        tag(10, n);
        int rc = skip() ? 0 : in_->describe(n, pts, desc, valid);
        io(&rc, 4); io(desc, (size_t) n * 32); io(valid, (size_t) n);
        return rc;
    }
    int triangulate(int n, int n_groups, const double *T36, const int *group, const double *bv_l, const double *bv_r, const float *unpx_l,
                    const float *unpx_r, double *wpt, double *inv_depth, uint8_t *status, double *parallax) override {
        tag(11, n, n_groups);
        int rc = skip() ? 0 : in_->triangulate(n, n_groups, T36, group, bv_l, bv_r, unpx_l, unpx_r, wpt, inv_depth, status, parallax);
        io(&rc, 4); io(wpt, (size_t) n * 24); io(inv_depth, (size_t) n * 8); io(status, (size_t) n); io(parallax, (size_t) n * 8);
        return rc;
    }
    int match_to_map(int cell_size, int ncw, int grid_cells, const int *cell_ptr, const int *cell_mp, int n_kf, const double *kf_q,
                     const double *kf_t, int n_mp, const double *mp_wpt, const uint8_t *mp_is3d, const uint8_t *mp_has_desc, const int *obs_ptr,
                     const int *obs_kf, const float *obs_px, const uint8_t *obs_desc, const uint8_t *obs_has_desc, int frame_kf, int n3d,
                     int n_local, const int *local, float max_proj_err, float dist_ratio, int *match_of_mp) override {
        tag(12, n_mp, n_local);
        int rc = skip() ? 0 : in_->match_to_map(cell_size, ncw, grid_cells, cell_ptr, cell_mp, n_kf, kf_q, kf_t, n_mp, mp_wpt, mp_is3d, mp_has_desc,
                                                obs_ptr, obs_kf, obs_px, obs_desc, obs_has_desc, frame_kf, n3d, n_local, local, max_proj_err,
                                                dist_ratio, match_of_mp);
        io(&rc, 4); io(match_of_mp, (size_t) n_mp * 4);
        return rc;
    }
    int match_to_map_rec(const MatchJob &job, int *match_of_mp) override {
        tag(13, job.n_mp, job.n_local);
        int rc = 0;
        if (!skip()) {
            Nest n(this);
            rc = Stages::match_to_map_rec(job, match_of_mp);   // the default's flatten over THIS object's descriptor tables
        }
        io(&rc, 4); io(match_of_mp, (size_t) job.n_mp * 4);
        return rc;
    }
    int local_ba(int n_kf, double *poses7, const uint8_t *kf_const, int n_pt, const int *pt_anchor_kf, const double *pt_anchor_uv,
                 double *pt_inv_depth, int n_obs, const int *obs_kf, const int *obs_pt, const double *obs_uv, int max_iters, double *chi2,
                 uint8_t *depth_pos) override {
        tag(14, n_pt, n_obs);
        int rc = skip() ? 0 : in_->local_ba(n_kf, poses7, kf_const, n_pt, pt_anchor_kf, pt_anchor_uv, pt_inv_depth, n_obs, obs_kf, obs_pt, obs_uv,
                                            max_iters, chi2, depth_pos);
        io(&rc, 4); io(poses7, (size_t) n_kf * 56); io(pt_inv_depth, (size_t) n_pt * 8); io(chi2, (size_t) n_obs * 8); io(depth_pos, (size_t) n_obs);
        return rc;
    }
    int local_ba_csr(int n_kf, double *poses7, const uint8_t *kf_const, int n_pt, const int *pt_ptr, const int *pt_anchor_kf,
                     const double *pt_anchor_uv, double *pt_inv_depth, int n_obs, const int *obs_kf, const double *obs_uv, int max_iters,
                     double chi2_threshold, uint64_t *bad_bits, int *n_bad) override {
        tag(15, n_pt, n_obs);
        int rc = 0;
        if (!skip()) {
            Nest n(this);
            rc = Stages::local_ba_csr(n_kf, poses7, kf_const, n_pt, pt_ptr, pt_anchor_kf, pt_anchor_uv, pt_inv_depth, n_obs, obs_kf, obs_uv, max_iters,
                                      chi2_threshold, bad_bits, n_bad);
        }
        io(&rc, 4); io(poses7, (size_t) n_kf * 56); io(pt_inv_depth, (size_t) n_pt * 8); io(bad_bits, (size_t) (n_obs / 64 + 1) * 8); io(n_bad, 4);
        return rc;
    }
    // the descriptor tables: kept (host build) while recording -- the flatten above reads them; on replay the log goes nowhere, as with
    // the HIP stages where the replay is a kernel the host only enqueues
    int medoid_replay(int n_ops, const alva_medoid::MedoidOp *ops, int n_mp, const int *mp_slot, const int *first_op, int slots) override {
        return replay_ ? 0 : Stages::medoid_replay(n_ops, ops, n_mp, mp_slot, first_op, slots);
    }
    int medoid_export(int n, const int *mp_slot, uint8_t *desc32, uint8_t *valid, int *info3) override {
        tag(16, n, (desc32 ? 1 : 0) | (valid ? 2 : 0) | (info3 ? 4 : 0));
        int rc = skip() ? 0 : Stages::medoid_export(n, mp_slot, desc32, valid, info3);
        io(&rc, 4);
        if (desc32) io(desc32, (size_t) n * 32);
        if (valid) io(valid, (size_t) n);
        if (info3) io(info3, (size_t) n * 12);
        return rc;
    }
    int find_plane(int n, const double *pts, const double *pose7_twc, int iterations, float *pose16, int *found) override {
        return in_->find_plane(n, pts, pose7_twc, iterations, pose16, found);
    }
    int n_pose_ = 0, want_pose_ = 0;
};

struct CpuSys {
    std::unique_ptr<RefStages> stages;
    std::unique_ptr<TraceStages> trace;
    std::unique_ptr<MemoStages> memo;
    std::unique_ptr<Slam> slam;
};
}  // namespace

extern "C" {

void *syscpu_create(int w, int h, double fx, double fy, double cx, double cy, double k1, double k2, double p1, double p2, int cellSize,
                    int claheEnabled, int doRandom) {
    auto *s = new CpuSys();
    Camera cam;
    cam.width = w; cam.height = h; cam.fx = fx; cam.fy = fy; cam.cx = cx; cam.cy = cy; cam.k1 = k1; cam.k2 = k2; cam.p1 = p1; cam.p2 = p2;
    Settings cfg;
    cfg.cell_size = cellSize > 0 ? cellSize : 40;
    cfg.clahe = claheEnabled != 0;
    cfg.random_sampling = doRandom != 0;
    s->stages.reset(new RefStages(cam, cfg.clahe));
    s->slam.reset(new Slam(s->stages.get(), cam, cfg));
    if (const char *path = getenv("ALVA_STAGE_TRACE_CPU")) {
        s->trace.reset(new TraceStages(s->stages.get(), path));
        s->trace->image_width_ = w;
        s->trace->image_height_ = h;
        s->slam->st = s->trace.get();
    }
    return s;
}
void syscpu_destroy(void *p) { delete static_cast<CpuSys *>(p); }

// the stage-result tape (MemoStages above): attach to a FRESH system before its first frame; replay != 0 delivers the recording
void *syscpu_tape_new() { return new Tape(); }
void syscpu_tape_free(void *t) { delete static_cast<Tape *>(t); }
long long syscpu_tape_bytes(void *t) { return (long long) static_cast<Tape *>(t)->bytes.size(); }
void syscpu_attach_tape(void *p, void *t, int replay) {
    CpuSys &s = *static_cast<CpuSys *>(p);
    Tape *tape = static_cast<Tape *>(t);
    if (replay) tape->pos = 0;
    s.memo.reset(new MemoStages(s.stages.get(), tape, replay != 0));
    s.memo->image_width_ = s.slam->st->image_width_;
    s.memo->image_height_ = s.slam->st->image_height_;
    s.slam->st = s.memo.get();
}
void syscpu_reset(void *p) { static_cast<CpuSys *>(p)->slam->reset(); }
int syscpu_find_camera_pose(void *p, const uint8_t *rgba, double timestamp, float *pose16, double *pose7) {
    Slam &s = *static_cast<CpuSys *>(p)->slam;
    const int status = s.process_frame(rgba, timestamp);
    if (pose16) pose_to_array(s.cur->Twc, pose16);
    if (pose7) se3_to_pose7(s.cur->Twc, pose7);
    return status;
}
void syscpu_set_init_pose(void *p, const double *pose7) {
    Slam &s = *static_cast<CpuSys *>(p)->slam;
    s.init_override.armed = pose7 != nullptr;
    if (pose7) std::memcpy(s.init_override.pose7, pose7, 56);
}
// sensitivity probe: relative perturbation applied to every bearing the map layer receives (0 = off)
void syscpu_set_perturbation(void *p, double rel) { static_cast<CpuSys *>(p)->stages->perturb = rel; }
void syscpu_state(void *p, int *out) { inspect_state(*static_cast<CpuSys *>(p)->slam, out); }
int syscpu_frame_keypoints(void *p, int cap, int *ids, float *px, float *unpx, uint8_t *is3d, uint8_t *hasDesc) {
    return inspect_frame(*static_cast<CpuSys *>(p)->slam->cur, cap, ids, px, unpx, is3d, hasDesc);
}
int syscpu_keyframe_ids(void *p, int cap, int *ids) { return inspect_keyframe_ids(*static_cast<CpuSys *>(p)->slam, cap, ids); }
int syscpu_keyframe(void *p, int kfid, double *pose7, int *info, int cap, int *ids, float *px, uint8_t *is3d) {
    return inspect_keyframe(*static_cast<CpuSys *>(p)->slam, kfid, pose7, info, cap, ids, px, is3d);
}
int syscpu_covisibility(void *p, int kfid, int cap, int *pairs) { return inspect_covisibility(*static_cast<CpuSys *>(p)->slam, kfid, cap, pairs); }
int syscpu_map_points(void *p, int cap, int *ids, double *xyz, int *flags, double *invDepth, uint8_t *desc) {
    return inspect_map_points(*static_cast<CpuSys *>(p)->slam, cap, ids, xyz, flags, invDepth, desc);
}
void syscpu_timing(void *p, double *sections8, double *keyframe16, int reset) {
    Slam &s = *static_cast<CpuSys *>(p)->slam;
    if (sections8) std::memcpy(sections8, s.t_section, sizeof(s.t_section));
    if (keyframe16) std::memcpy(keyframe16, s.t_kf, sizeof(s.t_kf));
    if (reset) {
        std::memset(s.t_section, 0, sizeof(s.t_section));
        std::memset(s.t_kf, 0, sizeof(s.t_kf));
    }
}
void syscpu_timing_fine(void *p, double *out32, int reset) {
    Slam &s = *static_cast<CpuSys *>(p)->slam;
    if (out32) std::memcpy(out32, s.t_fine, sizeof(s.t_fine));
    if (reset) std::memset(s.t_fine, 0, sizeof(s.t_fine));
}
// The product's descriptor-table record (alvaar_amd/csrc/slam/medoid_table.hpp), HOST build, one map point: applies `n` logged operations
// (64-byte MedoidOp records, chained or not: applied in array order) to `table` (alva_medoid::Table bytes, in/out; reset first when
// `fresh`).  tests/test_medoid_table.py drives it operation by operation against the reference's MapPoint (ref_mappoint_desc_ops).
int syscpu_medoid_apply(void *table, int fresh, int n, const void *ops) {
    alva_medoid::Table &t = *static_cast<alva_medoid::Table *>(table);
    if (fresh) alva_medoid::reset(t);
    const alva_medoid::MedoidOp *o = static_cast<const alva_medoid::MedoidOp *>(ops);
    for (int i = 0; i < n; i++) alva_medoid::apply(t, o[i]);
    return (int) sizeof(alva_medoid::Table);
}
void syscpu_counters(void *p, long *out) {
    Slam &s = *static_cast<CpuSys *>(p)->slam;
    out[0] = s.n_ba_runs; out[1] = s.n_merges; out[2] = s.n_kf_culled;
}

}  // extern "C"
