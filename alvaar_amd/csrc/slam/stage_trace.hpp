// Debug decorator around any `Stages` implementation: appends every stage call's inputs and outputs to a binary log
// (ALVA_STAGE_TRACE=<path> turns it on in alva_system_configure; the test-only CPU library honours the same variable).  Two logs
// of the same frame sequence -- HIP stages vs the reference's L1 stages -- are compared call by call with tools/stage_trace_diff.py
// to find the FIRST stage call whose outputs differ for equal inputs.  Record: char name[16], int32 count, then per array
// { char tag[8], char dtype, int64 bytes, data }.
#pragma once
#include <cstdio>
#include <cstring>
#include <string>
#include "stages.hpp"

namespace alva_slam {

class TraceStages : public Stages {
public:
    // Tracing composes the tracking step from the fine-grained stages (the default track_begin / track_pose_collect of this object),
    // so that every stage call is visible -- a fused override of the inner implementation is bypassed while tracing.  The same holds, ON
    // PURPOSE, for the record-level entry points added since (match_to_map_rec, local_ba_csr, describe_begin / _end, the medoid log, the
    // record arena): they are NOT forwarded, so a traced run takes their DEFAULT forms -- the records flattened on the host and handed to
    // the traced match_to_map / local_ba, descriptor tables kept by the host-side default -- and the log stays comparable call by call
    // with the reference-backed stages' log.  A traced run therefore exercises the fine-grained kernels, not the fused production path.
    TraceStages(Stages *inner, const char *path) : in_(inner) { f_ = std::fopen(path, "wb"); }
    ~TraceStages() override {
        if (f_) std::fclose(f_);
    }
    int new_frame(const uint8_t *rgba) override {
        begin("new_frame", 0);
        return in_->new_frame(rgba);
    }
    void reset_images() override { in_->reset_images(); }
    int fbklt(int levels, int n, const float *pts, float *prior, uint8_t *status) override {
        begin("fbklt", 5);
        arr("levels", 'i', &levels, 4);
        arr("pts", 'f', pts, (size_t) n * 8);
        arr("prior_in", 'f', prior, (size_t) n * 8);
        const int rc = in_->fbklt(levels, n, pts, prior, status);
        arr("prior_out", 'f', prior, (size_t) n * 8);
        arr("status", 'b', status, (size_t) n);
        return rc;
    }
    int compute_keypoints(int n, const float *px, float *unpx, double *bv) override {
        begin("compute_kp", 3);
        arr("px", 'f', px, (size_t) n * 8);
        const int rc = in_->compute_keypoints(n, px, unpx, bv);
        arr("unpx", 'f', unpx, (size_t) n * 8);
        arr("bv", 'd', bv, (size_t) n * 24);
        return rc;
    }
    int project_dist(int n, const double *cam_pts, float *px) override {
        begin("project_dist", 2);
        arr("cam", 'd', cam_pts, (size_t) n * 24);
        const int rc = in_->project_dist(n, cam_pts, px);
        arr("px", 'f', px, (size_t) n * 8);
        return rc;
    }
    int p3p(int n, const double *bv, const double *wpt, int do_random, double *pose7, int *outliers, int *n_outliers, int *ok) override {
        begin("p3p", 6);
        arr("bv", 'd', bv, (size_t) n * 24);
        arr("wpt", 'd', wpt, (size_t) n * 24);
        arr("pose_in", 'd', pose7, 56);
        const int rc = in_->p3p(n, bv, wpt, do_random, pose7, outliers, n_outliers, ok);
        arr("ok", 'i', ok, 4);
        arr("pose_out", 'd', pose7, 56);
        arr("outliers", 'i', outliers, (size_t) *n_outliers * 4);
        return rc;
    }
    int pnp(int n, const double *uv, const double *wpt, double *pose7, int *outliers, int *n_outliers, int *ok) override {
        begin("pnp", 6);
        arr("uv", 'd', uv, (size_t) n * 16);
        arr("wpt", 'd', wpt, (size_t) n * 24);
        arr("pose_in", 'd', pose7, 56);
        const int rc = in_->pnp(n, uv, wpt, pose7, outliers, n_outliers, ok);
        arr("ok", 'i', ok, 4);
        arr("pose_out", 'd', pose7, 56);
        arr("outliers", 'i', outliers, (size_t) *n_outliers * 4);
        return rc;
    }
    int five_point(int n, const double *b1, const double *b2, int do_random, double *R, double *t, int *outliers, int *n_outliers, int *ok) override {
        begin("five_point", 6);
        arr("bv1", 'd', b1, (size_t) n * 24);
        arr("bv2", 'd', b2, (size_t) n * 24);
        const int rc = in_->five_point(n, b1, b2, do_random, R, t, outliers, n_outliers, ok);
        arr("ok", 'i', ok, 4);
        arr("R", 'd', R, 72);
        arr("t", 'd', t, 24);
        arr("outliers", 'i', outliers, (size_t) *n_outliers * 4);
        return rc;
    }
    int detect(int cell, int n_occ, const float *occupied, int cap, float *pts, int *count) override {
        begin("detect", 2);
        arr("occupied", 'f', occupied, (size_t) n_occ * 8);
        const int rc = in_->detect(cell, n_occ, occupied, cap, pts, count);
        arr("pts", 'f', pts, (size_t) (*count > 0 ? *count : 0) * 8);
        return rc;
    }
    int describe(int n, const float *pts, uint8_t *desc, uint8_t *valid) override {
        begin("describe", 3);
        arr("pts", 'f', pts, (size_t) n * 8);
        const int rc = in_->describe(n, pts, desc, valid);
        arr("desc", 'b', desc, (size_t) n * 32);
        arr("valid", 'b', valid, (size_t) n);
        return rc;
    }
    int triangulate(int n, int n_groups, const double *T36, const int *group, const double *bv_l, const double *bv_r, const float *unpx_l,
                    const float *unpx_r, double *wpt, double *inv_depth, uint8_t *status, double *parallax) override {
        begin("triangulate", 8);
        arr("T", 'd', T36, (size_t) n_groups * 288);
        arr("bvl", 'd', bv_l, (size_t) n * 24);
        arr("bvr", 'd', bv_r, (size_t) n * 24);
        arr("group", 'i', group, (size_t) n * 4);
        const int rc = in_->triangulate(n, n_groups, T36, group, bv_l, bv_r, unpx_l, unpx_r, wpt, inv_depth, status, parallax);
        arr("wpt", 'd', wpt, (size_t) n * 24);
        arr("inv_depth", 'd', inv_depth, (size_t) n * 8);
        arr("status", 'b', status, (size_t) n);
        arr("parallax", 'd', parallax, (size_t) n * 8);
        return rc;
    }
    int match_to_map(int cell_size, int ncw, int grid_cells, const int *cell_ptr, const int *cell_mp, int n_kf, const double *kf_q,
                     const double *kf_t, int n_mp, const double *mp_wpt, const uint8_t *mp_is3d, const uint8_t *mp_has_desc, const int *obs_ptr,
                     const int *obs_kf, const float *obs_px, const uint8_t *obs_desc, const uint8_t *obs_has_desc, int frame_kf, int n3d,
                     int n_local, const int *local, float max_proj_err, float dist_ratio, int *match_of_mp) override {
        begin("match_to_map", 4);
        arr("mp_wpt", 'd', mp_wpt, (size_t) n_mp * 24);
        arr("kf_q", 'd', kf_q, (size_t) n_kf * 32);
        arr("local", 'i', local, (size_t) n_local * 4);
        const int rc = in_->match_to_map(cell_size, ncw, grid_cells, cell_ptr, cell_mp, n_kf, kf_q, kf_t, n_mp, mp_wpt, mp_is3d, mp_has_desc, obs_ptr,
                                         obs_kf, obs_px, obs_desc, obs_has_desc, frame_kf, n3d, n_local, local, max_proj_err, dist_ratio, match_of_mp);
        arr("match", 'i', match_of_mp, (size_t) n_mp * 4);
        return rc;
    }
    int local_ba(int n_kf, double *poses7, const uint8_t *kf_const, int n_pt, const int *pt_anchor_kf, const double *pt_anchor_uv,
                 double *pt_inv_depth, int n_obs, const int *obs_kf, const int *obs_pt, const double *obs_uv, int max_iters, double *chi2,
                 uint8_t *depth_pos) override {
        begin("local_ba", 11);
        arr("poses_in", 'd', poses7, (size_t) n_kf * 56);
        arr("kf_const", 'b', kf_const, (size_t) n_kf);
        arr("anc_kf", 'i', pt_anchor_kf, (size_t) n_pt * 4);
        arr("anc_uv", 'd', pt_anchor_uv, (size_t) n_pt * 16);
        arr("inv_in", 'd', pt_inv_depth, (size_t) n_pt * 8);
        arr("obs_kf", 'i', obs_kf, (size_t) n_obs * 4);
        arr("obs_pt", 'i', obs_pt, (size_t) n_obs * 4);
        arr("obs_uv", 'd', obs_uv, (size_t) n_obs * 16);
        const int rc = in_->local_ba(n_kf, poses7, kf_const, n_pt, pt_anchor_kf, pt_anchor_uv, pt_inv_depth, n_obs, obs_kf, obs_pt, obs_uv,
                                     max_iters, chi2, depth_pos);
        arr("poses_out", 'd', poses7, (size_t) n_kf * 56);
        arr("inv_out", 'd', pt_inv_depth, (size_t) n_pt * 8);
        arr("chi2", 'd', chi2, (size_t) n_obs * 8);
        return rc;
    }
    int find_plane(int n, const double *pts, const double *pose7_twc, int iterations, float *pose16, int *found) override {
        return in_->find_plane(n, pts, pose7_twc, iterations, pose16, found);
    }

private:
    void begin(const char *name, int count) {
        if (!f_) return;
        char nm[16] = {0};
        std::strncpy(nm, name, 15);
        std::fwrite(nm, 1, 16, f_);
        std::fwrite(&count, 4, 1, f_);
    }
    void arr(const char *tag, char dtype, const void *data, size_t bytes) {
        if (!f_) return;
        char tg[8] = {0};
        std::strncpy(tg, tag, 7);
        const long long b = (long long) bytes;
        std::fwrite(tg, 1, 8, f_);
        std::fwrite(&dtype, 1, 1, f_);
        std::fwrite(&b, 8, 1, f_);
        if (bytes) std::fwrite(data, 1, bytes, f_);
        std::fflush(f_);
    }
    Stages *in_;
    FILE *f_ = nullptr;
};

}  // namespace alva_slam
