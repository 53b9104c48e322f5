// HIP implementation of the map layer's stage interface (see slam/stages.hpp and stages_hip.hip).
#pragma once
#include "slam/se3.hpp"
#include "slam/stages.hpp"

namespace alva_slam {

class HipStages : public Stages {
public:
    HipStages();
    ~HipStages() override;
    // hip_stream != nullptr: run on that (caller-owned) stream instead of a stream of its own -- sessions of a group may share streams
    int init(int device, const Camera &cam, bool clahe, const double *invK, void *hip_stream = nullptr);
    // One pass of synthetic data through every stage at the sizes a `cell`-pixel grid produces: loads the kernels' code objects, grows
    // the staging arenas and the context scratch to their working sizes and raises the launch attributes, so that none of this lands in
    // the first frames / the first keyframes.  No state survives it (the detector's adaptive threshold is restored).
    int warm_up(int cell);

    int track_begin(const TrackJob &job, TrackKlt &out) override;
    int track_pose_collect(TrackPose &out) override;
    bool track_slot_buffers(int n, float **px, uint8_t **is3d, double **wpt) override;
    uint16_t *track_carry_buffer(int n_prev, int n) override;
    int new_frame(const uint8_t *rgba) override;
    int new_frame_device(const uint8_t *d_rgba) override;
    void hint_next_frame_device(const uint8_t *d_rgba) override;
    int frame_done() override;
    // explicit page-lock + device mapping of the caller's frame buffer (see new_frame); buf == nullptr releases it
    int register_frame_buffer(const uint8_t *buf, size_t bytes);
    int unregister_frame_buffer();
    // the caller's one frame buffer in host-writable DEVICE memory (include/alvaar_system.h alva_system_alloc_frame_buffer)
    int alloc_frame_buffer(size_t bytes, uint8_t **h_writable);
    void reset_images() override;
    int fbklt(int levels, int n, const float *pts, float *prior, uint8_t *status) override;
    int compute_keypoints(int n, const float *px, float *unpx, double *bv) override;
    int project_dist(int n, const double *cam_pts, float *px) override;
    int p3p(int n, const double *bv, const double *wpt, int do_random, double *pose7, int *outliers, int *n_outliers, int *ok) override;
    int pnp(int n, const double *unpx_d, const double *wpt, double *pose7, int *outliers, int *n_outliers, int *ok) override;
    int five_point(int n, const double *bv_kf, const double *bv_cur, int do_random, double *R, double *t, int *outliers, int *n_outliers,
                   int *ok) override;
    int detect(int cell, int n_occ, const float *occupied, int cap, float *pts, int *count) override;
    int detect_begin(int cell, int n_occ, const float *occupied, int cap) override;
    int detect_end(float *pts, int *count) override;
    int describe(int n, const float *pts, uint8_t *desc, uint8_t *valid) override;
    int describe_begin(int n, const float *pts) override;
    int describe_end(uint8_t *desc, uint8_t *valid) override;
    int describe_and_compute(int n, const float *pts, uint8_t *desc, uint8_t *valid, float *unpx, double *bv) override;
    int triangulate(int n, int n_groups, const double *T36, const int *group, const double *bv_l, const double *bv_r, const float *unpx_l,
                    const float *unpx_r, double *wpt, double *inv_depth, uint8_t *status, double *parallax) override;
    int match_to_map(int cell_size, int num_cells_w, int grid_cells, const int *cell_ptr, const int *cell_mp, int n_kf, const double *kf_q,
                     const double *kf_t, int n_mp, const double *mp_wpt, const uint8_t *mp_is3d, const uint8_t *mp_has_desc, const int *obs_ptr,
                     const int *obs_kf, const float *obs_px, const uint8_t *obs_desc, const uint8_t *obs_has_desc, int frame_kf,
                     int num_keypoints_3d, int n_local, const int *local, float max_proj_err, float dist_ratio, int *match_of_mp) override;
    int match_to_map_rec(const MatchJob &job, int *match_of_mp) override;
    MpRec *mp_arena_chunk(int chunk) override;
    int local_ba(int n_kf, double *poses7, const uint8_t *kf_const, int n_pt, const int *pt_anchor_kf, const double *pt_anchor_uv,
                 double *pt_inv_depth, int n_obs, const int *obs_kf, const int *obs_pt, const double *obs_uv, int max_iters, double *chi2,
                 uint8_t *depth_pos) override;
    int local_ba_csr(int n_kf, double *poses7, const uint8_t *kf_const, int n_pt, const int *pt_ptr, const int *pt_anchor_kf,
                     const double *pt_anchor_uv, double *pt_inv_depth, int n_obs, const int *obs_kf, const double *obs_uv, int max_iters,
                     double chi2_threshold, uint64_t *bad_bits, int *n_bad) override;
    int find_plane(int n, const double *pts, const double *pose7_twc, int iterations, float *pose16, int *found) override;
    uint8_t *stage_scratch(size_t bytes) override;
    int medoid_replay(int n_ops, const alva_medoid::MedoidOp *ops, int n_mp, const int *mp_slot, const int *first_op, int slots) override;
    int medoid_export(int n, const int *mp_slot, uint8_t *desc32, uint8_t *valid, int *info3) override;
    int medoid_dump(int mp_slot, alva_medoid::Table *out) override;
    int pack_map_records(int n_slots, int stream_id, int capacity, uint8_t *d_out, int *count) override;

private:
    int build_from(const uint8_t *d_src);
    int build_ahead();
    struct Impl;
    Impl *m;
    bool fused_active_ = false;
    int pose_total_ = 0;
};

}  // namespace alva_slam
