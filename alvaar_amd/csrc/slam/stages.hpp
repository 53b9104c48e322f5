// The seam between the host-side map layer (slam.cpp: the reference's L2 bookkeeping) and the numeric stages
// (the reference's L1 -> L0 calls).  In libalvaar_hip.so the ONLY implementation is the HIP one
// (stages_hip.hip): every method is one or more calls through include/alvaar_hip.h on the GPU, and it fails loudly without a
// device.  The interface exists so that the host logic can be unit-tested on a machine without a GPU: the test-only library
// under oracle/ (never shipped, never loaded by alvaar_amd/) compiles the same slam.cpp against the reference's own L1
// functions and compares it with the reference's System frame by frame.
//
// All pointers are HOST pointers; arrays are flat.  Every method returns 0 on success, a negative alva error code otherwise.
#pragma once
#include <cstddef>
#include <cstdint>
#include <cstring>
#include <vector>
#include "medoid_table.hpp"
#include "mp_rec.hpp"
#include <memory>

namespace alva_slam {

struct Camera {
    int width = 0, height = 0, border = 20;  // system.cpp:29
    double fx = 0, fy = 0, cx = 0, cy = 0, k1 = 0, k2 = 0, p1 = 0, p2 = 0;
};

// One tracking step (VisualFrontend::kltTrackingFromMotionPrior + computePose, visual_frontend.cpp:103-417) as data: the frame's
// keypoints as SLOTS in the frame container's iteration order.
struct TrackJob {
    int n = 0;
    const float *px = nullptr;        // [n][2] keypoint positions in the previous image
    const uint8_t *is3d = nullptr;    // [n]
    const double *wpt = nullptr;      // [n][3] world point of the slot's map point (3-D slots; ignored otherwise)
    double Tcw_q[4] = {0, 0, 0, 1}, Tcw_t[3] = {0, 0, 0};  // predicted pose, world -> camera (motion model applied)
    double pose7_pred[7] = {0, 0, 0, 0, 0, 0, 1};           // the same as Twc (start of the refinement when P3P is off)
    int use_prior = 1;                // kltUsePrior_
    int klt_levels = 3;               // kltPyramidLevels_
    int want_pose = 0;                // map initialised: solve the pose behind the tracker
    int do_p3p = 1;                   // p3pReq_ || p3pEnabled_
    int do_random = 1;                // multiViewRandomEnabled_
    // null, or the buffer track_carry_buffer handed out, filled: slot i of this frame was slot carry[i] of the PREVIOUS track_begin -- the
    // implementation still holds that frame's table and tracked positions and builds this one from them; px / is3d / wpt are not read
    const uint16_t *carry = nullptr;
};
// after the tracker: per slot  code 0 = lost, 1 = tracked from its projected prior on one level, 2 = tracked on the full pyramid,
// 3 = tracked on the full pyramid after failing with the prior; px / unpx / bv valid where code != 0
struct TrackKlt {
    // read through the views; they point at the vectors below (default implementation) or at the implementation's own staging
    // (the HIP stages: pinned host memory the kernels wrote, valid until the next track_begin)
    const uint8_t *code_v = nullptr;
    const float *px_v = nullptr, *unpx_v = nullptr;
    const double *bv_v = nullptr;
    std::vector<uint8_t> code;
    std::vector<float> px, unpx;
    std::vector<double> bv;
    int p3p_req = 0;   // fewer than 33 % of the priors held (visual_frontend.cpp:197-202)
    int n_pose = 0;    // surviving 3-D slots = correspondences of the pose solve, in slot order
};
// after the pose solve: status -1 = not attempted (fewer than 4 correspondences), 0 = P3P rejected, 1 = P3P pose accepted but the
// refinement rejected, 2 = refined pose accepted; masks indexed by correspondence (k-th surviving 3-D slot)
struct TrackPose {
    int status = -1;
    double pose7_p3p[7] = {0, 0, 0, 0, 0, 0, 1}, pose7[7] = {0, 0, 0, 0, 0, 0, 1};
    std::vector<uint8_t> p3p_outlier, pnp_outlier;
};

// Mapper::matchToMap (mapper.cpp:354-588) as a job over the map layer's RECORDS (mp_rec.hpp): the frame's grid cells and the local
// list name rows of a map-point table, a row names the record / descriptor-table slot of its map point; the observations are read
// from the records, the descriptors from the tables of the same slots (flush the medoid log first).
struct MatchJob {
    int cell_size = 0, num_cells_w = 0, grid_cells = 0;
    const int *cell_ptr = nullptr, *cell_mp = nullptr;   // [grid_cells + 1], rows of the cells' keypoints
    int n_kf = 0;
    const int *kf_ids = nullptr;                         // the keyframes of the map, ascending id
    const double *kf_q = nullptr, *kf_t = nullptr;       // their poses world -> camera
    int n_mp = 0;
    const int *mp_slot = nullptr;                        // row -> record slot
    int frame_kfid = 0, num_keypoints_3d = 0;
    int n_local = 0;
    const int *local = nullptr;                          // rows of the local map points, in the local map's order
    float max_proj_err = 0.f, dist_ratio = 0.f;
};

struct Stages {
    virtual ~Stages() {}

    // The tracking step.  The default implementation composes it from the fine-grained stages below in the reference's order
    // (track_default.cpp); the HIP implementation overrides it with one device-side chain (one host wait after the tracker, one after
    // the pose).  track_begin returns with the tracker's results; the pose solve (when job.want_pose) may still be running and is
    // collected by track_pose_collect -- the map layer does its tracker bookkeeping in between.
    virtual int track_begin(const TrackJob &job, TrackKlt &out);
    virtual int track_pose_collect(TrackPose &out);
    // Where the map layer may assemble the slot table of the next track_begin (positions, 3-D flags, world points; capacity n):
    // an implementation that stages its inputs anyway hands out that staging, so the table is written once.  false = none.
    virtual bool track_slot_buffers(int n, float **px, uint8_t **is3d, double **wpt) { (void) n; (void) px; (void) is3d; (void) wpt; return false; }
    // The slot table of a frame that only LOST slots since the previous track_begin (no keyframe, no reset in between: positions are that
    // frame's tracked positions, flags and world points unchanged) need not be assembled at all: the caller names, per slot, the slot it
    // was (n indices into the previous frame's n_prev slots).  Non-null = the implementation can do that for the next track_begin (it
    // still has the previous frame's table); null = assemble the table.
    virtual uint16_t *track_carry_buffer(int n_prev, int n) { (void) n_prev; (void) n; return nullptr; }

    // System::findCameraPose's cvtColor(RGBA2GRAY) (system.cpp:111-112) + VisualFrontend::preprocessImage
    // (visual_frontend.cpp:672-698): the current image / pyramid become the previous ones, the new frame's gray image
    // (CLAHE-equalised copy when enabled) and LK pyramid become current.
    virtual int new_frame(const uint8_t *rgba) = 0;
    // the same with the frame already in the implementation's device memory (frames resident in HBM: a capture pipeline that delivers
    // into device memory, bench.py's timed loop); unsupported (-4) where there is no device
    virtual int new_frame_device(const uint8_t *d_rgba) { (void) d_rgba; return -4; }
    // Optional: the frame the NEXT new_frame_device call will pass (device memory, valid until then).  An implementation may build that
    // frame's gray image / pyramid ahead, behind the current frame's pose solve; results never depend on the hint (a wrong one is dropped).
    virtual void hint_next_frame_device(const uint8_t *d_rgba) { (void) d_rgba; }
    // end of System::processCameraPose: the caller may reuse its frame buffer once this returns
    virtual int frame_done() { return 0; }
    // VisualFrontend::reset (visual_frontend.cpp:716-727): forget both images / pyramids
    virtual void reset_images() = 0;

    // FeatureTracker::fbKltTracking(prevPyramid_, currPyramid_, 9, levels, kltError_, kltMaxFbDistance_, pts, prior, status)
    // (feature_tracker.cpp:5-111; call sites visual_frontend.cpp:162-171, :211-220).  prior in/out, status 1 = tracked.
    virtual int fbklt(int levels, int n, const float *pts, float *prior, uint8_t *status) = 0;

    // Frame::computeKeypoint (frame.cpp:105-122): px -> unpx (CameraCalibration::undistortImagePoint) -> bv = normalised K^-1 (unpx, 1)
    virtual int compute_keypoints(int n, const float *px, float *unpx, double *bv) = 0;
    // Frame::projCamToImageDist (camera_calibration.cpp:34-54) of camera-frame points
    virtual int project_dist(int n, const double *cam_pts, float *px) = 0;

    // VisualFrontend::computePose's two solver calls (visual_frontend.cpp:300-317, :363-375) with the constants of state.hpp:67-77.
    // p3p: *ok = return value of MultiViewGeometry::p3pRansac; pose7 written when ok; outliers = ascending index list.
    virtual int p3p(int n, const double *bv, const double *wpt, int do_random, double *pose7, int *outliers, int *n_outliers, int *ok) = 0;
    // pnp: pose7 in/out; *ok = return value of MultiViewGeometry::ceresPnP
    virtual int pnp(int n, const double *unpx_d, const double *wpt, double *pose7, int *outliers, int *n_outliers, int *ok) = 0;

    // MultiViewGeometry::compute5ptEssentialMatrix (visual_frontend.cpp:517-528): R (row-major), t (not normalised), outlier list
    virtual int five_point(int n, const double *bv_kf, const double *bv_cur, int do_random, double *R, double *t, int *outliers,
                           int *n_outliers, int *ok) = 0;

    // FeatureExtractor::detectFeaturePoints(currImage_, cell, occupied, roi) (map_manager.cpp:213) on the current (equalised) image;
    // the detector's adaptive quality threshold is state of the implementation and survives System::reset like the
    // reference's FeatureExtractor object does (system.cpp:31, :42-55).
    virtual int detect(int cell, int n_occ, const float *occupied, int cap, float *pts, int *count) = 0;
    // The same call split so that the caller can work while the detector runs: detect_begin starts it, detect_end delivers what
    // detect() would have.  Default: nothing happens until detect_end, which calls detect().  No other stage call in between.
    virtual int detect_begin(int cell, int n_occ, const float *occupied, int cap) {
        det_cell_ = cell; det_n_occ_ = n_occ; det_cap_ = cap;
        det_occ_.assign(occupied, occupied + 2 * (size_t) (n_occ > 0 ? n_occ : 0));
        return 0;
    }
    virtual int detect_end(float *pts, int *count) { return detect(det_cell_, det_n_occ_, det_occ_.data(), det_cap_, pts, count); }
    // FeatureExtractor::describeFeaturePoints(imageRaw, pts) (map_manager.cpp:204, :218) on the current RAW gray image
    virtual int describe(int n, const float *pts, uint8_t *desc, uint8_t *valid) = 0;

    // The same call split like detect(): describe_begin starts it, describe_end delivers what describe() would have; the caller works in
    // between (the keyframe's observer bookkeeping, Slam::create_keyframe) and makes no other stage call.  Default: nothing happens until
    // describe_end, which calls describe().
    virtual int describe_begin(int n, const float *pts) {
        dsc_n_ = n;
        dsc_pts_.assign(pts, pts + 2 * (size_t) (n > 0 ? n : 0));
        return 0;
    }
    virtual int describe_end(uint8_t *desc, uint8_t *valid) { return describe(dsc_n_, dsc_pts_.data(), desc, valid); }

    // describe() + compute_keypoints() of the SAME points in one call (MapManager::addKeypointsToFrame needs both for the detector's new
    // points, map_manager.cpp:166-191 / :218): the default composes the two; an implementation with a device round trip per call saves one
    virtual int describe_and_compute(int n, const float *pts, uint8_t *desc, uint8_t *valid, float *unpx, double *bv) {
        const int rc = describe(n, pts, desc, valid);
        return rc ? rc : compute_keypoints(n, pts, unpx, bv);
    }

    // per-keypoint arithmetic of Mapper::triangulateTemporal (mapper.cpp:222-287); see alva_triangulate
    virtual int triangulate(int n, int n_groups, const double *T36, const int *group, const double *bv_l, const double *bv_r,
                            const float *unpx_l, const float *unpx_r, double *wpt, double *inv_depth, uint8_t *status,
                            double *parallax) = 0;

    // Chunk `chunk` of the map layer's record arena (mp_rec.hpp): MP_CHUNK zero-filled MpRec records that stay where they are for the
    // life of the stages object.  The map layer edits them in place; an implementation whose kernels read the records hands out memory
    // they can reach (the HIP stages: pinned host memory, gathered by zero-copy reads).  nullptr = allocation failed.
    virtual MpRec *mp_arena_chunk(int chunk) {   // (may be called from the map layer's helper thread while the caller's thread READS other chunks)
        if (arena_.capacity() < 4096) arena_.reserve(4096);   // growth inside the capacity never moves the elements a reader holds
        if ((size_t) chunk >= arena_.size()) arena_.resize((size_t) chunk + 1);
        if (!arena_[(size_t) chunk]) arena_[(size_t) chunk].reset(new MpRec[MP_CHUNK]());
        return arena_[(size_t) chunk].get();
    }

    // Host scratch of at least `bytes` in which the map layer may assemble the arrays of the NEXT match_to_map / local_ba call (valid
    // until that call returns).  An implementation that stages its inputs anyway hands out that staging (the HIP stages: pinned memory,
    // uploaded with one copy when the arrays of the call lie inside it); the default is a plain vector.  nullptr = allocation failed.
    virtual uint8_t *stage_scratch(size_t bytes) {
        if (scratch_.size() < bytes) scratch_.resize(bytes + bytes / 2);
        return scratch_.data();
    }
    // Mapper::matchToMap on a flattened map (mapper.cpp:354-588); see alva_match_to_map_flags for the layout
    virtual int match_to_map(int cell_size, int num_cells_w, int grid_cells, const int *cell_ptr, const int *cell_mp, int n_kf,
                             const double *kf_q, const double *kf_t, int n_mp, const double *mp_wpt, const uint8_t *mp_is3d,
                             const uint8_t *mp_has_desc, const int *obs_ptr, const int *obs_kf, const float *obs_px,
                             const uint8_t *obs_desc, const uint8_t *obs_has_desc, int frame_kf,
                             int num_keypoints_3d, int n_local, const int *local, float max_proj_err, float dist_ratio,
                             int *match_of_mp) = 0;

    // The same call on the records.  Default: flatten the rows' records (observing keyframes that exist and hold the keypoint; the
    // descriptor an observation's keyframe contributed, from the default medoid tables) and call match_to_map -- the GPU-less harness;
    // the HIP stages gather the records on the device instead and never build the flat map.
    virtual int match_to_map_rec(const MatchJob &job, int *match_of_mp);

    // the solve inside Optimizer::localBA (optimizer.cpp:251-262, anchored inverse depth); see alva_local_ba
    virtual int local_ba(int n_kf, double *poses7, const uint8_t *kf_const, int n_pt, const int *pt_anchor_kf, const double *pt_anchor_uv,
                         double *pt_inv_depth, int n_obs, const int *obs_kf, const int *obs_pt, const double *obs_uv, int max_iters,
                         double *chi2, uint8_t *depth_pos) = 0;

    // The same solve for residual blocks GROUPED BY POINT (the map layer emits them that way): pt_ptr[n_pt + 1] delimits each point's
    // blocks in obs_kf / obs_uv; every point has at least one.  The outlier sweep's test (optimizer.cpp:266-309: chi2 > chi2_threshold or
    // the point behind the camera) comes back as one bit per residual block (bad_bits: n_obs / 64 + 1 words) + their count.  Default:
    // local_ba() on the expanded arrays (the GPU-less harness); the HIP stages run alva_local_ba_csr.
    virtual int local_ba_csr(int n_kf, double *poses7, const uint8_t *kf_const, int n_pt, const int *pt_ptr, const int *pt_anchor_kf,
                             const double *pt_anchor_uv, double *pt_inv_depth, int n_obs, const int *obs_kf, const double *obs_uv, int max_iters,
                             double chi2_threshold, uint64_t *bad_bits, int *n_bad) {
        std::vector<int> obs_pt((size_t) n_obs);
        for (int p = 0; p < n_pt; p++)
            for (int q = pt_ptr[p]; q < pt_ptr[p + 1]; q++) obs_pt[(size_t) q] = p;
        std::vector<double> chi2((size_t) n_obs);
        std::vector<uint8_t> dpos((size_t) n_obs);
        const int rc = local_ba(n_kf, poses7, kf_const, n_pt, pt_anchor_kf, pt_anchor_uv, pt_inv_depth, n_obs, obs_kf, obs_pt.data(), obs_uv, max_iters,
                                chi2.data(), dpos.data());
        if (rc) return rc;
        int nb = 0;
        for (int w = 0; w < n_obs / 64 + 1; w++) bad_bits[w] = 0;
        for (int q = 0; q < n_obs; q++)
            if (chi2[(size_t) q] > chi2_threshold || !dpos[(size_t) q]) {
                bad_bits[q >> 6] |= 1ull << (q & 63);
                nb++;
            }
        *n_bad = nb;
        return 0;
    }

    // MapPoint's descriptor tables and medoids (map_point.cpp:73-181; medoid_table.hpp) as a replayed operation log: the map layer keeps
    // only the KEY sets of the tables (its control flow needs nothing else) and logs every edit -- addDesc, the descriptor half of
    // removeObservedKeyframeId, the release when the last observation goes, a new map point -- per map point SLOT (slots are recycled;
    // `slots` = highest slot in use + 1).  mp_slot[i] names a map point with operations in this log, first_op[i] the head of its chain
    // (MedoidOp::next).  medoid_replay may only ENQUEUE the work (the HIP stages: one wavefront per touched map point, behind the
    // keyframe's other kernels); medoid_export waits for everything replayed so far and returns desc_ / !desc_.empty() / {#descriptors,
    // keyframe desc_ was taken from, overflow} per requested slot; medoid_dump copies one table as it is (tests).
    // The defaults keep the tables on the host (the GPU-less harness under oracle/); the HIP stages override all three.
    virtual int medoid_replay(int n_ops, const alva_medoid::MedoidOp *ops, int n_mp, const int *mp_slot, const int *first_op, int slots) {
        if ((int) med_tables_.size() < slots) {
            const size_t old = med_tables_.size();
            med_tables_.resize((size_t) slots + 1024);
            for (size_t i = old; i < med_tables_.size(); i++) alva_medoid::reset(med_tables_[i]);
        }
        for (int i = 0; i < n_mp; i++)
            for (int o = first_op[i]; o >= 0 && o < n_ops; o = ops[o].next) alva_medoid::apply(med_tables_[(size_t) mp_slot[i]], ops[o]);
        return 0;
    }
    virtual int medoid_export(int n, const int *mp_slot, uint8_t *desc32, uint8_t *valid, int *info3) {
        for (int i = 0; i < n; i++) {
            static const alva_medoid::Table fresh = [] { alva_medoid::Table t{}; alva_medoid::reset(t); return t; }();
            const alva_medoid::Table &t = mp_slot[i] >= 0 && (size_t) mp_slot[i] < med_tables_.size() ? med_tables_[(size_t) mp_slot[i]] : fresh;
            if (desc32) memcpy(desc32 + 32 * (size_t) i, t.medoid, 32);
            if (valid) valid[i] = (uint8_t) t.medoid_valid;
            if (info3) { info3[3 * i] = t.count; info3[3 * i + 1] = t.medoid_kf; info3[3 * i + 2] = t.overflow; }
        }
        return 0;
    }
    virtual int medoid_dump(int mp_slot, alva_medoid::Table *out) {
        if (mp_slot < 0 || (size_t) mp_slot >= med_tables_.size()) return -1;
        *out = med_tables_[(size_t) mp_slot];
        return 0;
    }

    // the shared-map exchange's record block written on the device from the records + descriptor tables of slots 0 .. n_slots - 1
    // (alva_pack_map_records); -4 where the map is not device-resident (the default stages)
    virtual int pack_map_records(int n_slots, int stream_id, int capacity, uint8_t *d_out, int *count) {
        (void) n_slots; (void) stream_id; (void) capacity; (void) d_out; (void) count;
        return -4;
    }

    // System::processPlane's fit (system.cpp:177-342, intended algorithm, parity unpinned)
    virtual int find_plane(int n, const double *pts, const double *pose7_twc, int iterations, float *pose16, int *found) = 0;

    // image size for Frame::isInImage in the default tracking step (set by the map layer)
    int image_width_ = 0, image_height_ = 0;

protected:
    std::vector<alva_medoid::Table> med_tables_;   // (default medoid_* only)
    std::vector<std::unique_ptr<MpRec[]>> arena_;  // (default mp_arena_chunk only)
    std::vector<uint8_t> scratch_;
    int det_cell_ = 0, det_n_occ_ = 0, det_cap_ = 0, dsc_n_ = 0;
    std::vector<float> det_occ_, dsc_pts_;
    // hand-over from the default track_begin to the default track_pose_collect
    struct PendingPose {
        bool active = false;
        int n = 0, do_p3p = 1, do_random = 1;
        std::vector<double> bv, uv, wpt;
        double pose7[7];
    } pending_;
};

}  // namespace alva_slam
