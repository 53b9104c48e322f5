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

namespace alva_slam {

struct Camera {
    int width = 0, height = 0, border = 20;  // system.cpp:29
    double fx = 0, fy = 0, cx = 0, cy = 0, k1 = 0, k2 = 0, p1 = 0, p2 = 0;
};

struct Stages {
    virtual ~Stages() {}

    // System::findCameraPose's cvtColor(RGBA2GRAY) (system.cpp:111-112) + VisualFrontend::preprocessImage
    // (visual_frontend.cpp:672-698): the current image / pyramid become the previous ones, the new frame's gray image
    // (CLAHE-equalised copy when enabled) and LK pyramid become current.
    virtual int new_frame(const uint8_t *rgba) = 0;
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
    // FeatureExtractor::describeFeaturePoints(imageRaw, pts) (map_manager.cpp:204, :218) on the current RAW gray image
    virtual int describe(int n, const float *pts, uint8_t *desc, uint8_t *valid) = 0;

    // per-keypoint arithmetic of Mapper::triangulateTemporal (mapper.cpp:222-287); see alva_triangulate
    virtual int triangulate(int n, int n_groups, const double *T36, const int *group, const double *bv_l, const double *bv_r,
                            const float *unpx_l, const float *unpx_r, double *wpt, double *inv_depth, uint8_t *status,
                            double *parallax) = 0;

    // Mapper::matchToMap on a flattened map (mapper.cpp:354-588); see alva_match_to_map_flags for the layout
    virtual int match_to_map(int cell_size, int num_cells_w, int grid_cells, const int *cell_ptr, const int *cell_mp, int n_kf,
                             const double *kf_q, const double *kf_t, int n_mp, const double *mp_wpt, const uint8_t *mp_is3d,
                             const uint8_t *mp_has_desc, const int *obs_ptr, const int *obs_kf, const float *obs_px,
                             const uint8_t *obs_desc, const uint8_t *obs_has_desc, int frame_kf,
                             int num_keypoints_3d, int n_local, const int *local, float max_proj_err, float dist_ratio,
                             int *match_of_mp) = 0;

    // the solve inside Optimizer::localBA (optimizer.cpp:251-262, anchored inverse depth); see alva_local_ba
    virtual int local_ba(int n_kf, double *poses7, const uint8_t *kf_const, int n_pt, const int *pt_anchor_kf, const double *pt_anchor_uv,
                         double *pt_inv_depth, int n_obs, const int *obs_kf, const int *obs_pt, const double *obs_uv, int max_iters,
                         double *chi2, uint8_t *depth_pos) = 0;

    // System::processPlane's fit (system.cpp:177-342, intended algorithm, parity unpinned)
    virtual int find_plane(int n, const double *pts, const double *pose7_twc, int iterations, float *pose16, int *found) = 0;
};

}  // namespace alva_slam
