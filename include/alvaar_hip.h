/*
 * alvaar_hip.h -- C ABI of the MI355X (gfx950) hot path of AlvaAR's visual-SLAM
 * front-end and local bundle adjustment.
 *
 * This is the drop-in seam: each entry point replaces one L1 -> L0 call of the
 * reference (array in / array out, SURVEY.md §8(a), §8(b) "internal seam for
 * HIP").  Plain pointers and sizes only; no C++/torch types.  Pointers named
 * d_* are DEVICE pointers (HBM, on the context's device); h_* are host
 * pointers.  All work is enqueued on the context's HIP stream; functions that
 * return results through h_* pointers synchronise that stream before returning,
 * all others are asynchronous.
 *
 * Return value: ALVA_OK (0) or a negative ALVA_ERR_* code; alva_last_error()
 * gives a thread-local message.  There is NO CPU fallback anywhere behind this
 * header: if no gfx950 device is usable the calls fail.
 */
#ifndef ALVAAR_HIP_H
#define ALVAAR_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum {
    ALVA_OK = 0,
    ALVA_ERR_ARG = -1,   /* bad argument */
    ALVA_ERR_HIP = -2,   /* HIP runtime error (see alva_last_error) */
    ALVA_ERR_NOMEM = -3, /* allocation failed */
    ALVA_ERR_STATE = -4  /* object not in a usable state */
};

typedef struct alva_ctx alva_ctx;
typedef struct alva_pyramid alva_pyramid;

/* ---- context -------------------------------------------------------------------------------- */
/* own_stream != 0: the context creates and owns a non-blocking stream (hip_stream ignored).
 * own_stream == 0: enqueue on the caller's hipStream_t `hip_stream` as given -- NULL is the
 * legacy default stream (e.g. torch's current stream handle, which is 0 for the default stream). */
int alva_ctx_create(int device, void *hip_stream, int own_stream, alva_ctx **out);
/* Context with its own non-blocking stream in one of the device's priority classes: -1 high, 0 normal, +1 low.  The HIP runtime
 * multiplexes the streams of a class onto a few hardware queues (GPU_MAX_HW_QUEUES, default 4); streams sharing a queue run
 * strictly one after the other, so lanes that must overlap (alva_frontend's tracker / detector / look-ahead lanes) are
 * created in different classes. */
int alva_ctx_create_with_priority(int device, int priority_class, alva_ctx **out);
void alva_ctx_destroy(alva_ctx *ctx);
int alva_ctx_sync(alva_ctx *ctx);
/* Per-kernel timing for bench.py's roofline line: while enabled every kernel launch of the library is bracketed by two
 * HIP events on its own stream.  alva_prof_report waits for them and writes one line per kernel,
 * "name<TAB>launches<TAB>average microseconds", into buf.  (Measurement plumbing; no reference counterpart.) */
int alva_prof_enable(int on);
int alva_prof_report(char *buf, size_t cap);
/* Stream-order dependency without a host wait: work enqueued on `ctx` after this call starts only after everything
 * enqueued so far on `producer` has completed.  (The reference is single-threaded; this is what lets one frame run
 * detection and tracking+pose on two HIP streams.) */
int alva_ctx_wait(alva_ctx *ctx, alva_ctx *producer);
void *alva_ctx_stream(alva_ctx *ctx);
const char *alva_last_error(void);
const char *alva_version(void);

/* ---- a2: RGBA -> gray ------------------------------------------------------------------------
 * Replaces cv::cvtColor(image, image, COLOR_RGBA2GRAY) at src/slam/src/system.cpp:112
 * (Y = (9798 R + 19235 G + 3735 B + 16384) >> 15; imgproc/src/color_rgb.simd.hpp:646-664).
 * Pitches in bytes; rgba_pitch % 16 == 0, gray_pitch % 4 == 0, width % 4 == 0. */
int alva_rgba2gray(alva_ctx *ctx, const uint8_t *d_rgba, size_t rgba_pitch, int width, int height,
                   uint8_t *d_gray, size_t gray_pitch);

/* ---- a3: LK pyramid with Scharr derivatives --------------------------------------------------
 * Replaces cv::buildOpticalFlowPyramid(img, pyr, Size(win,win), max_level) at
 * src/slam/src/visual_frontend.cpp:696 (video/src/lkpyramid.cpp:726-822; pyrDown
 * imgproc/src/pyramids.cpp:746-; ScharrDerivInvoker lkpyramid.cpp:70-151).
 * Every level is stored padded by `win` pixels on each side: gray REFLECT_101, derivatives
 * (interleaved int16 Ix,Iy) constant 0 -- the layout the LK tracker reads. */
typedef struct alva_pyr_level {
    int width, height;   /* interior size of this level */
    uint8_t *d_gray;     /* device pointer to interior pixel (0,0); valid for x,y in [-win, size+win) */
    size_t gray_pitch;   /* bytes */
    int16_t *d_deriv;    /* device pointer to interior element (0,0), 2 x int16 per pixel */
    size_t deriv_pitch;  /* bytes */
} alva_pyr_level;

int alva_pyramid_create(alva_ctx *ctx, int width, int height, int win, int max_level, alva_pyramid **out);
void alva_pyramid_destroy(alva_pyramid *pyr);
/* number of levels actually built (OpenCV stops when the next level would be <= win in either
 * dimension, lkpyramid.cpp:811-816) */
int alva_pyramid_num_levels(const alva_pyramid *pyr);
int alva_pyramid_level(const alva_pyramid *pyr, int level, alva_pyr_level *out);
int alva_pyramid_build_from_gray(alva_ctx *ctx, alva_pyramid *pyr, const uint8_t *d_gray, size_t gray_pitch);
/* Copies one padded level to host (synchronises): h_gray (h+2win) x (w+2win) u8 contiguous,
 * h_deriv same x 2 int16.  Either may be NULL. */
int alva_pyramid_download_level(alva_ctx *ctx, const alva_pyramid *pyr, int level, uint8_t *h_gray, int16_t *h_deriv);
/* fused a2+a3: level 0 is converted straight from the RGBA frame (also the entry of
 * System::findCameraPose, system.cpp:106-112).  d_gray_out may be NULL. */
int alva_pyramid_build_from_rgba(alva_ctx *ctx, alva_pyramid *pyr, const uint8_t *d_rgba, size_t rgba_pitch,
                                 uint8_t *d_gray_out, size_t gray_out_pitch);
/* The same for `count` cameras of one geometry in FIVE launches (one grid layer per camera) instead of 5 x count: one 640x480
 * frame is 1.2 MB and every launch above is bound by launch latency; many frames per launch is what lets the same kernels run at
 * HBM speed (bench.py "batched_preprocess").  Results are identical to `count` calls of alva_pyramid_build_from_rgba.
 * pyrs / d_rgba / d_gray_out (may be NULL) are HOST arrays of `count` handles / device pointers.  Enqueue only. */
int alva_pyramid_build_from_rgba_batch(alva_ctx *ctx, alva_pyramid *const *pyrs, const uint8_t *const *d_rgba, size_t rgba_pitch,
                                       uint8_t *const *d_gray_out, size_t gray_out_pitch, int count);

/* ---- a4: forward-backward pyramidal KLT ------------------------------------------------------
 * alva_lk_track replaces one cv::calcOpticalFlowPyrLK(prevPyr, nextPyr, pts, next, status, err,
 * Size(win,win), num_levels, {COUNT+EPS,max_iters,eps}, USE_INITIAL_FLOW|LK_GET_MIN_EIGENVALS)
 * (video/src/lkpyramid.cpp:183-724,1239-1404).  d_next is in/out (initial flow), d_status u8,
 * d_err = min eigenvalue (float).
 * alva_fbklt_track replaces FeatureTracker::fbKltTracking (src/slam/src/feature_tracker.cpp:5-111):
 * forward LK on num_levels, gate (status && err <= err_thresh && inBorder), backward LK on level 0
 * only, keep if |p - back| <= fb_dist.  d_prior in/out, d_status out (1 = tracked). */
int alva_lk_track(alva_ctx *ctx, const alva_pyramid *prev, const alva_pyramid *next, int num_levels,
                  int max_iters, float eps, const float *d_pts, float *d_next, uint8_t *d_status,
                  float *d_err, int n);
int alva_fbklt_track(alva_ctx *ctx, const alva_pyramid *prev, const alva_pyramid *curr, int num_levels,
                     float err_thresh, float fb_dist, int max_iters, float eps, const float *d_pts,
                     float *d_prior, uint8_t *d_status, int n);

/* ---- a6: 256-bit steered-BRIEF (ORB) description of given points ------------------------------
 * Replaces FeatureExtractor::describeFeaturePoints (src/slam/src/feature_extractor.cpp:160-214) =
 * cv::ORB::create(500,1.,0)->compute(): border-32 REFLECT_101 copy, 7x7 sigma=2 Gaussian
 * (features2d/src/orb.cpp:1188), 256 tests at round(R(angle) * pattern) with angle = -1 deg
 * (orb.cpp:219-284; core/src/types.cpp:93-101).  Points within 31 px of the image edge get
 * d_valid[i] = 0 and a zero descriptor (the reference returns an empty Mat for them).
 * d_desc: n x 32 bytes. */
int alva_describe(alva_ctx *ctx, const uint8_t *d_gray, size_t gray_pitch, int width, int height,
                  const float *d_pts, int n, uint8_t *d_desc, uint8_t *d_valid);
/* The blurred border-32 image ORB samples from (for stage-level parity tests):
 * d_out is (height+64) x (width+64), pitch out_pitch. */
int alva_orb_blur(alva_ctx *ctx, const uint8_t *d_gray, size_t gray_pitch, int width, int height,
                  uint8_t *d_out, size_t out_pitch);

/* ---- a5': FAST-9/16 + NMS, and the full ORB detector -------------------------------------------
 * alva_fast replaces cv::FAST(img, kps, threshold, true, TYPE_9_16) (features2d/src/fast.cpp:56-292,
 * score fast_score.cpp:120-): keypoints are emitted in row-major order; d_xy int32 x,y pairs,
 * d_score int32.  *h_count receives the total found (may exceed cap; only cap are written). */
int alva_fast(alva_ctx *ctx, const uint8_t *d_gray, size_t gray_pitch, int width, int height, int threshold,
              int *d_xy, int *d_score, int cap, int *h_count);

typedef struct alva_orb alva_orb;
/* Replaces cv::ORB::create(nfeatures, scale, nlevels, 31, 0, 2, HARRIS_SCORE, 31, fast_threshold)
 * ->detectAndCompute (features2d/src/orb.cpp:784-1218). */
int alva_orb_create(alva_ctx *ctx, int width, int height, int nfeatures, float scale_factor, int nlevels,
                    int fast_threshold, alva_orb **out);
void alva_orb_destroy(alva_orb *orb);
/* d_kp: cap x 6 floats {x, y, size, angle, response, octave}; d_desc: cap x 32 bytes (may be NULL for
 * detect-only).  Keypoint ORDER within a level is canonical (row-major by y then x) rather than
 * nth_element's unspecified order; the SET equals the reference's (SURVEY.md §8a a5'). */
int alva_orb_detect_and_compute(alva_ctx *ctx, alva_orb *orb, const uint8_t *d_gray, size_t gray_pitch,
                                float *d_kp, uint8_t *d_desc, int cap, int *h_count);
/* h_count == NULL above only enqueues the work (no host wait); this call then waits for it and returns the count. */
int alva_orb_collect(alva_ctx *ctx, alva_orb *orb, int *h_count);
/* cv::ORB::detectAndCompute of `count` cameras in one set of launches (grid z = camera): every camera has its own detector object
 * (all of one geometry), gray image, keypoint and descriptor buffer (each of capacity cap); each camera's output equals its own
 * alva_orb_detect_and_compute.  Enqueue-only; alva_orb_collect_batch waits and returns the per-camera keypoint counts. */
int alva_orb_detect_and_compute_batch(alva_ctx *ctx, alva_orb *const *orbs, int count, const uint8_t *const *d_gray,
                                      size_t gray_pitch, float *const *d_kp, uint8_t *const *d_desc, int cap);
int alva_orb_collect_batch(alva_ctx *ctx, alva_orb *const *orbs, int count, int *h_counts);
/* Number of keypoints (since the last reset) whose rBRIEF rotation (float) cos / sin could not be PROVEN to round like the host C
 * library's (features2d/src/orb.cpp:230-232): the kernel decides the two floats from a ~100-bit evaluation and only a true value
 * within 2^-50 of a float rounding boundary is left unproven (probability ~3e-8 per keypoint).  Synchronous. */
int alva_orb_ambiguous_rotations(int *h_count, int reset);
/* One level of the detector's pyramid as the last run left it (which = 0: orb.cpp:1086-1099's resize chain; which = 1: its 7x7 sigma-2
 * blur, orb.cpp:1188) copied to d_out (w x h bytes, rows out_pitch apart; d_out NULL only returns the size).  For stage-level parity tests.
 * ALVA_ORB_PYRAMID=chain (read at alva_orb_create) builds the pyramid with one resize launch per level instead of the fused launch. */
int alva_orb_debug_level(alva_ctx *ctx, alva_orb *orb, int level, int which, uint8_t *d_out, size_t out_pitch, int *w, int *h);
/* device-resident keypoint count of the detector's last run (what alva_orb_collect copies to the host) */
const int *alva_orb_device_count(const alva_orb *orb);

/* ---- a5: the reference's grid Shi-Tomasi detector ---------------------------------------------
 * Replaces FeatureExtractor::detectFeaturePoints (src/slam/src/feature_extractor.cpp:11-158).
 * h_max_quality is the detector's adaptive threshold, in/out (stateful across calls, :138-145).
 * d_occupied: n_occ x 2 floats (already-tracked keypoints).  d_out_pts: cap x 2 floats,
 * sub-pixel refined (cornerSubPix win 3, 30 it, eps 0.01).  *h_count = number of points. */
int alva_detect_grid(alva_ctx *ctx, const uint8_t *d_gray, size_t gray_pitch, int width, int height,
                     int cell_size, const float *d_occupied, int n_occ, int roi_x, int roi_y, int roi_w,
                     int roi_h, double *h_max_quality, float *d_out_pts, int cap, int *h_count);
/* The same call split at its only host wait: _enqueue launches the detection on the context's stream with the threshold `max_quality`
 * and returns at once; _collect waits, reports the count and applies the adaptive-threshold rule (:138-145) to *h_max_quality.  The
 * caller does host work in between (the map layer updates its descriptor medoids while the detector runs).  No other call on the
 * context between the two. */
typedef struct alva_detect_pending {
    void *h_cnt;
    int n_cells;
    int reserved;
} alva_detect_pending;
int alva_detect_grid_enqueue(alva_ctx *ctx, const uint8_t *d_gray, size_t gray_pitch, int width, int height, int cell_size,
                             const float *d_occupied, int n_occ, int roi_x, int roi_y, int roi_w, int roi_h, double max_quality,
                             float *d_out_pts, int cap, alva_detect_pending *pending);
int alva_detect_grid_collect(alva_ctx *ctx, const alva_detect_pending *pending, double *h_max_quality, int *h_count);

/* ---- a7: Hamming brute-force matcher ----------------------------------------------------------
 * Replaces cv::BFMatcher(NORM_HAMMING).match(query, train) (core/src/batch_distance.cpp:199-251,
 * norm.cpp:99-): per query the smallest distance, LOWEST train index on ties.
 * Descriptors are 32 bytes each, rows 32-byte aligned. */
int alva_bf_match_hamming(alva_ctx *ctx, const uint8_t *d_query, int n_query, const uint8_t *d_train,
                          int n_train, int *d_idx, int *d_dist);
/* `count` independent matches in one pair of launches.  Match c: queries d_query[c], whose number is read from device memory
 * (*d_n_query[c], clamped to cap_query) so that the call can be enqueued behind the detector that produces them; train set
 * d_train[c] with n_train[c] rows (0 = skip that match); rows of d_idx[c] / d_dist[c] beyond the query count stay untouched.
 * expected_queries sizes the grid (larger counts are covered by a loop).  Enqueue-only. */
int alva_bf_match_hamming_batch(alva_ctx *ctx, int count, const uint8_t *const *d_query, const int *const *d_n_query,
                                int cap_query, const uint8_t *const *d_train, const int *n_train, int *const *d_idx,
                                int *const *d_dist, int expected_queries);

/* ---- a8: P3P + LMedS absolute pose ------------------------------------------------------------
 * Replaces MultiViewGeometry::p3pRansac(obs, wpts, max_iters, err_thr, optimize=false, doRandom,
 * fx, fy, Twc, outliers) (src/slam/src/multi_view_geometry.cpp:24-127) = opengv::sac::Lmeds<
 * AbsolutePoseSacProblem(KNEIP)> (opengv/sac/implementation/Lmeds.hpp:43-195).
 * Sampling runs on the host with the reference's own sampler classes (std::mt19937 +
 * std::uniform_int_distribution, prefix Fisher-Yates; SampleConsensusProblem.hpp:40-120): do_random = 0
 * seeds with `seed` (the reference's fixed seed is 12345u), do_random != 0 seeds from the clock like
 * the reference's default.  alva_p3p_draw_samples exposes that index stream (count x 4 int32).
 * Outputs (host): R row-major 3x3 + t (Twc), outlier index list (capacity n); *h_ok = 1 on success
 * (>= 5 inliers and an orthogonal R, multi_view_geometry.cpp:82-91).  n <= 19000 (the LMedS median is LDS-resident). */
int alva_p3p_draw_samples(int n_points, int count, int do_random, uint32_t seed, int *h_samples);
int alva_p3p_lmeds(alva_ctx *ctx, const double *d_bearings, const double *d_wpts, int n, int max_iters,
                   float err_threshold, int do_random, uint32_t seed, float fx, float fy, double *h_R,
                   double *h_t, int *h_outliers, int *h_n_outliers, int *h_ok);

/* ---- a9: robust PnP refinement (motion-only BA) -----------------------------------------------
 * Replaces MultiViewGeometry::ceresPnP (src/slam/src/multi_view_geometry.cpp:129-223): Huber LM on
 * the 6-DoF pose (Ceres trust_region_minimizer.cc / levenberg_marquardt_strategy.cc semantics,
 * wall-clock cap removed), chi2 outlier sweep, optional L2 re-solve.
 * h_pose7 = [tx,ty,tz,qx,qy,qz,qw] (Twc) in/out.  h_info[8]: iterations/cost of both solves. */
int alva_pnp_refine(alva_ctx *ctx, const double *d_uv, const double *d_wpts, int n, double *h_pose7,
                    int max_iters, float chi2_th, int use_robust, int apply_l2_after_robust, float fx,
                    float fy, float cx, float cy, int *h_outliers, int *h_n_outliers, double *h_info,
                    int *h_ok);

/* ---- a8 + a9 chained: VisualFrontend::computePose (src/slam/src/visual_frontend.cpp:245-417) -----------------
 * P3P-LMedS, its acceptance tests, removal of its outliers and the robust PnP refinement of the inliers run as one
 * device-side chain with ONE host synchronisation.  d_uv are the undistorted pixel observations of the same n points.
 * Outputs: h_pose7 (written when *h_status >= 1 and PnP produced a pose), per-point masks in the caller's
 * indexing (either may be NULL), *h_status: 0 = P3P rejected (the reference resets the frame), 1 = P3P pose accepted
 * but the refinement failed its checks (:383-399), 2 = refined pose accepted. */
int alva_compute_pose(alva_ctx *ctx, const double *d_bearings, const double *d_uv, const double *d_wpts, int n,
                      int p3p_iters, float p3p_err, int do_random, uint32_t seed, int pnp_iters, float chi2_th,
                      float fx, float fy, float cx, float cy, double *h_pose7, uint8_t *h_p3p_outlier,
                      uint8_t *h_pnp_outlier, int *h_status);
/* The same in two halves: _enqueue launches the chain without waiting, _collect waits and returns the results.  No other
 * call may be made on `ctx` in between (other contexts / streams are free to run: that is the point).  _collect waits on a completion
 * word the last kernel publishes in pinned host memory after all results (ALVA_NO_POLL=1: on the stream instead); later calls on `ctx`
 * are stream-ordered behind the chain either way. */
int alva_compute_pose_enqueue(alva_ctx *ctx, const double *d_bearings, const double *d_uv, const double *d_wpts, int n,
                              int p3p_iters, float p3p_err, int do_random, uint32_t seed, int pnp_iters, float chi2_th,
                              float fx, float fy, float cx, float cy);
int alva_compute_pose_collect(alva_ctx *ctx, double *h_pose7, uint8_t *h_p3p_outlier, uint8_t *h_pnp_outlier,
                              int *h_status);

/* ---- f2b: two-view map initialisation ----------------------------------------------------------------
 * Replaces MultiViewGeometry::compute5ptEssentialMatrix(bvs1, bvs2, maxIterations, errorThreshold, optimize, doRandom, fx, fy,
 * Rwc, twc, outliers) (src/slam/src/multi_view_geometry.cpp:225-320; caller VisualFrontend::checkReadyForInit,
 * visual_frontend.cpp:517-528 with state.hpp:67-69: 100 iterations, 3 px, optimize = true): OpenGV RANSAC (99 % confidence,
 * adaptive iteration count) over Nister's five-point solver with 8-index samples, then Levenberg-Marquardt refinement of
 * (t, Cayley(R)) on the inliers.  d_bv1 / d_bv2: n x 3 unit bearings in the previous keyframe / the current frame (device);
 * the model is X1 = R X2 + t.  Outputs (host): h_R (3 x 3 row-major), h_t (as the reference, NOT normalised), h_inlier[n]
 * (1 = inlier; the reference's outlier list is its complement; may be NULL), *h_ok = the reference's return value
 * (0: n < 8, no model, or fewer than 10 inliers).  do_random = 0 draws the reference's deterministic sample stream
 * (std::mt19937 seeded 12345u when seed = 12345).  Synchronous.
 * The refinement is a forward-difference optimiser working at its rounding-noise floor: the reference's own result moves by
 * 1e-6 .. 1e-4 when one input changes by one ulp, and agreement with it is of that size. */
typedef struct alva_relpose_info {
    int iterations;      /* RANSAC iterations the reference would have run (Ransac::iterations_) */
    int n_inliers;
    int draws;           /* samples consumed (iterations + samples without a real root) */
    int lm_iterations, lm_status, lm_nfev; /* Eigen::LevenbergMarquardt iter / status code / function evaluations */
    double ransac_model[12]; /* R (row-major) | t before the refinement */
} alva_relpose_info;
int alva_compute_5pt_essential(alva_ctx *ctx, const double *d_bv1, const double *d_bv2, int n, int max_iters,
                               float error_threshold, int optimize, int do_random, uint32_t seed, float fx, float fy,
                               double *h_R, double *h_t, uint8_t *h_inlier, alva_relpose_info *h_info, int *h_ok);
/* The sample stream (count x 8 int32, SampleConsensusProblem.hpp:65-84) and the hypothesis stage alone: one
 * CentralRelativePoseSacProblem::computeModelCoefficients (CentralRelativePoseSacProblem.cpp:38-247) + countWithinDistance per
 * sample; h_counts[k] = -1 when sample k has no model. */
int alva_relpose_draw_samples(int n_points, int count, int do_random, uint32_t seed, int *h_samples8);
int alva_relpose_hypotheses(alva_ctx *ctx, const double *d_bv1, const double *d_bv2, int n, const int *h_samples8, int n_samples,
                            float error_threshold, float fx, float fy, double *h_models12, int *h_counts);

/* ---- f3: plane under the map points -------------------------------------------------------------------
 * The INTENDED algorithm of System::processPlane(mapPoints, Twc, numIterations) (src/slam/src/system.cpp:177-342, caller
 * findPlane :123-137): RANSAC over planes through 3 sampled points (orientation test, k-th smallest distance as the score),
 * inliers within 1.4 x the best score, least-squares refit, pose = [Rodrigues(...) Rodrigues((1,0,0)) | mean of the inliers]
 * in the layout of Utils::toPoseArray(cv::Mat).  As shipped the reference function has no defined behaviour;
 * parity is pinned against the reference's own function compiled with its four defects repaired (oracle/ref_shim_plane.cpp) and
 * against the CPU restatement oracle/alva_oracle_plane.c, which is pinned to the same (tests/test_plane.py).
 * d_points: n x 3 world points (device, f64); h_pose7_twc: current pose; h_samples3 (num_iterations x 3 int32, may be NULL):
 * the sample indices, otherwise drawn from std::mt19937(seed) (clock-seeded when do_random).  *h_found = 0 when n < 32 or
 * fewer than 32 inliers.  Synchronous. */
int alva_find_plane(alva_ctx *ctx, const double *d_points, int n, const double *h_pose7_twc, int num_iterations, int do_random,
                    uint32_t seed, const int *h_samples3, float *h_plane_pose16, int *h_found);

/* ---- f4a: CLAHE ------------------------------------------------------------------------------
 * Replaces cv::createCLAHE(clip_limit, Size(tiles_x, tiles_y))->apply(src, dst) for 8-bit images
 * (imgproc/src/clahe.cpp:120-420), which VisualFrontend::preprocessImage runs when claheEnabled_
 * (src/slam/src/visual_frontend.cpp:16-18 with clip 3 and tiles = size / 50, :678-681; off in the shipped
 * configuration, system.cpp:17).  Bit-exact, including the REFLECT_101 extension for sizes that the grid does not divide.
 * d_dst may not alias d_src.  Enqueue only. */
int alva_clahe(alva_ctx *ctx, const uint8_t *d_src, size_t src_pitch, int width, int height, double clip_limit,
               int tiles_x, int tiles_y, uint8_t *d_dst, size_t dst_pitch);

/* ---- f1: Mapper::matchToMap on a flattened map ----------------------------------------------------
 * Replaces the loops of Mapper::matchToMap(frame, maxProjectionError, distRatio, localMapPointIds)
 * (src/slam/src/mapper.cpp:354-588; caller matchingToLocalMap :334 with state.hpp:62-63) for a consistent map.  The host
 * flattens its containers once per keyframe:
 *   h_calib10        fx fy cx cy k1 k2 p1 p2 imgWidth imgHeight (the shared CameraCalibration)
 *   grid             the frame's keypoint grid as Frame stores it (frame.cpp:250-260, :313-341): cell_size, num_cells_w,
 *                    grid_cells, d_cell_ptr[grid_cells + 1], d_cell_mp = map point INDEX of each stored keypoint, stored order
 *   keyframes        d_kf_q[n_kf][4] (x y z w) and d_kf_t[n_kf][3]: T_cw of every keyframe; frame_kf = the one being matched
 *   map points       d_mp_wpt[n_mp][3], d_mp_is3d[n_mp], observations d_obs_ptr[n_mp + 1] -> d_obs_kf (ascending keyframe
 *                    index = MapPoint::observedKeyframeIds_ order), d_obs_px (the keypoint's px_ in that keyframe),
 *                    d_obs_desc (32 B each, 16-B aligned: MapPoint::mapKeyframeDescriptors_)
 *   d_local          indices of the local map points in the iteration order of frame.localMapPointIds_
 * Output d_match_of_mp[n_mp]: for the map point m of a frame keypoint, the index of the local map point that
 * matchToMap pairs with it (the reference's mapPrevIdNewId[keypointId] = mapPointId), else -1.  Enqueue only. */
int alva_match_to_map(alva_ctx *ctx, const double *h_calib10, int cell_size, int num_cells_w, int grid_cells,
                      const int *d_cell_ptr, const int *d_cell_mp, int n_kf, const double *d_kf_q, const double *d_kf_t,
                      int n_mp, const double *d_mp_wpt, const uint8_t *d_mp_is3d, const int *d_obs_ptr,
                      const int *d_obs_kf, const float *d_obs_px, const uint8_t *d_obs_desc, int frame_kf,
                      int num_keypoints_3d, int n_local, const int *d_local, float max_proj_err, float dist_ratio,
                      int *d_match_of_mp);
/* The same for a LIVE map, in which a keyframe may hold a keypoint it could not describe (within 31 px of the border,
 * src/slam/src/feature_extractor.cpp:191-209): d_mp_has_desc[m] = !MapPoint::desc_.empty() (the gates at mapper.cpp:404, :465),
 * d_obs_has_desc[o] = MapPoint::mapKeyframeDescriptors_ has an entry for that observation's keyframe (its 32-byte slot is ignored
 * otherwise).  NULL flags = every observation carries a descriptor (alva_match_to_map). */
int alva_match_to_map_flags(alva_ctx *ctx, const double *h_calib10, int cell_size, int num_cells_w, int grid_cells,
                            const int *d_cell_ptr, const int *d_cell_mp, int n_kf, const double *d_kf_q, const double *d_kf_t, int n_mp,
                            const double *d_mp_wpt, const uint8_t *d_mp_is3d, const uint8_t *d_mp_has_desc, const int *d_obs_ptr,
                            const int *d_obs_kf, const float *d_obs_px, const uint8_t *d_obs_desc, const uint8_t *d_obs_has_desc,
                            int frame_kf, int num_keypoints_3d, int n_local, const int *d_local, float max_proj_err, float dist_ratio,
                            int *d_match_of_mp);

/* The same call on the map layer's RECORDS instead of a flattened map (round 5; csrc/slam/mp_rec.hpp states the layout: one
 * 1024-byte record per map point -- worldPoint_, is3d_, !desc_.empty(), and per keyframe an entry {keyframe id, flags
 * observed-by / holds-the-keypoint / has-a-descriptor, the keypoint's px_ and unpx_ in that keyframe}, i.e.
 * MapPoint::observedKeyframeIds_ + what Frame::getKeypointById returns for it (src/slam/src/mapper.cpp:487-515) -- in chunks of
 * 4096 records of PINNED host memory that the host edits in place).  The caller names one record slot per table row
 * (d_mp_slot); a gather kernel reads the rows' records out of host memory (d_record_chunks: device-readable table of the chunks'
 * addresses), keeps the observations whose keyframe is listed in d_kf_ids (ascending ids; d_kf_q / d_kf_t their T_cw;
 * frame_kf_index / frame_kf_id name the keyframe being matched) and holds the keypoint, and takes the descriptors from the
 * device-resident descriptor tables of the same slots (d_desc_tables = alva_medoid_tables(): mapKeyframeDescriptors_, what
 * MapPoint::computeMinDescDist iterates, map_point.cpp:206-222).  Grid, local list, thresholds and output as above.  Enqueue only. */
int alva_match_to_map_records(alva_ctx *ctx, const double *h_calib10, int cell_size, int num_cells_w, int grid_cells,
                              const int *d_cell_ptr, const int *d_cell_mp, int n_kf, const int *d_kf_ids, const double *d_kf_q,
                              const double *d_kf_t, int frame_kf_index, int frame_kf_id, int n_mp, const int *d_mp_slot,
                              const void *const *d_record_chunks, const void *d_desc_tables, int num_keypoints_3d, int n_local,
                              const int *d_local, float max_proj_err, float dist_ratio, int *d_match_of_mp);

/* ---- f4b: lens distortion paths of CameraCalibration --------------------------------------
 * alva_undistort_points replaces CameraCalibration::undistortImagePoint (src/slam/src/camera_calibration.cpp:56-72) =
 * cv::undistortPoints(pts, out, K, D, R = K) with D = (k1, k2, p1, p2), 5 fixed iterations
 * (calib3d/src/undistort.dispatch.cpp:384-556): pixel in, undistorted pixel out (n x 2 f32 each).
 * alva_project_dist replaces CameraCalibration::projectCamToImageDist (:34-54) = cv::projectPoints of (x/z, y/z, 1)
 * rounded to float, zero rvec / tvec (calib3d/src/calibration.cpp:522-): camera-frame points (n x 3 f64) in, distorted
 * pixels (n x 2 f32) out.  Both bit-exact; both are used by the reference even when the coefficients are zero.
 * Enqueue only. */
int alva_undistort_points(alva_ctx *ctx, const float *d_px, int n, double fx, double fy, double cx, double cy, double k1,
                          double k2, double p1, double p2, float *d_out);
int alva_project_dist(alva_ctx *ctx, const double *d_cam_pts, int n, double fx, double fy, double cx, double cy,
                      double k1, double k2, double p1, double p2, float *d_out);

/* ---- f2a: triangulation of a new keyframe's 2-D keypoints -------------------------------------
 * Replaces the per-keypoint arithmetic of Mapper::triangulateTemporal (src/slam/src/mapper.cpp:222-287):
 * MultiViewGeometry::triangulate (= opengv::triangulation::triangulate2, opengv/src/triangulation/methods.cpp:67-90),
 * the rotation-compensated parallax (:246-248), the cheirality gate z < 0.1 (:256) and the reprojection gate (:266-272).
 * The caller groups the points by the keyframe that first observed them: d_T holds one block of 36 doubles per group,
 * { R_lr[9], t_lr[3], R_rl[9], t_rl[3], R_wl[9], t_wl[3] } row-major (l = that keyframe, r = the new keyframe,
 * w = world; T_lr = T_lw * T_wr, :226-228), d_group[i] selects the block of point i.  d_bv_*: unit bearings (n x 3 f64),
 * d_unpx_*: undistorted pixels (n x 2 f32).  Outputs per point: point in the l camera, world point, inverse depth
 * 1 / z_l, status (0 = accepted, 1 = behind a camera, 2 = reprojection error), parallax in pixels (the reference drops
 * the observation of rejected points whose parallax exceeds 20, :258-262/:274-278 -- host bookkeeping).  Enqueue only. */
int alva_triangulate(alva_ctx *ctx, int n, const double *d_T, int n_groups, const int *d_group, const double *d_bv_l,
                     const double *d_bv_r, const float *d_unpx_l, const float *d_unpx_r, double fx, double fy, double cx,
                     double cy, float max_reproj_err, double *d_lpt, double *d_wpt, double *d_inv_depth,
                     uint8_t *d_status, double *d_parallax);

/* ---- per-frame driver: the caller of a2-a9 -------------------------------------------------------------------
 * Mirrors the order of VisualFrontend::trackMono (src/slam/src/visual_frontend.cpp:83-150): preprocessImage (:672-698)
 * -> kltTracking (:152-243) -> computePose (:245-417), plus the keyframe branch's feature work
 * (MapManager::extractKeypoints, map_manager.cpp:196-231, with the detector the north_star names, and BFMatcher
 * matching against the previous frame, map_point.cpp:106-212).  The reference runs them back to back on one CPU
 * thread; here the detector + matcher run on a second HIP stream while the first tracks and solves the pose.  One host
 * wait for the pose, one for the keypoint count.  All inputs are device pointers: d_rgba the frame; d_pts n_pts x 2
 * float keypoints to track from the previous frame; d_bearings / d_uv / d_wpts n_corr 2-D/3-D correspondences for the
 * pose (doubles, as alva_compute_pose).  *h_pose_status as alva_compute_pose. */
typedef struct alva_frontend alva_frontend;
int alva_frontend_create(int device, int width, int height, int max_tracked, int orb_features, alva_frontend **out);
void alva_frontend_destroy(alva_frontend *fe);
int alva_frontend_track(alva_frontend *fe, const uint8_t *d_rgba, size_t rgba_pitch, const float *d_pts, int n_pts,
                        const double *d_bearings, const double *d_uv, const double *d_wpts, int n_corr, float fx,
                        float fy, float cx, float cy, double *h_pose7, int *h_pose_status, int *h_n_keypoints);
/* The same with one frame of look-ahead: d_rgba_next (may be NULL) is the frame the NEXT call will pass as d_rgba; its gray
 * image and pyramid are built on a third HIP stream while this frame is tracked, so preprocessImage leaves the dependent chain
 * pyramid -> KLT -> P3P -> PnP.  Contract: d_rgba_next is read asynchronously from now until the NEXT alva_frontend_track* call on this
 * object has returned, so its contents must not change in between, and "the same pointer" on that next call means "the same, unchanged
 * contents" (the look-ahead result is used as is).  A next call with any other d_rgba waits for the pending look-ahead build and
 * rebuilds.  Results are identical to alva_frontend_track. */
int alva_frontend_track_ahead(alva_frontend *fe, const uint8_t *d_rgba, size_t rgba_pitch, const uint8_t *d_rgba_next,
                              const float *d_pts, int n_pts, const double *d_bearings, const double *d_uv, const double *d_wpts,
                              int n_corr, float fx, float fy, float cx, float cy, double *h_pose7, int *h_pose_status,
                              int *h_n_keypoints);
/* Device-resident results of the last alva_frontend_track: tracked positions (n_pts x 2) + status, ORB keypoints
 * (n x 6) + descriptors (n x 32), matches of the n descriptors against the previous frame's.  Any may be NULL.
 * Call alva_frontend_sync first if anything but later alva_frontend_* calls is going to read them. */
int alva_frontend_results(alva_frontend *fe, const float **d_tracked, const uint8_t **d_track_status,
                          const float **d_keypoints, const uint8_t **d_descriptors, const int **d_match_idx,
                          const int **d_match_dist);
int alva_frontend_sync(alva_frontend *fe);
/* Measurement helper: n_streams independent camera streams on one GPU, one host thread per stream, each running `steps`
 * alva_frontend_track calls on its own alva_frontend (d_frames[s * ring + k] = frame k of stream s; the other inputs are
 * per-stream arrays of device pointers).  *h_seconds = wall time from the common start to the last stream's finish. */
int alva_frontend_run_many(alva_frontend **fes, int n_streams, int steps, int warmup, const uint8_t *const *d_frames,
                           int ring, size_t rgba_pitch, const float *const *d_pts, int n_pts,
                           const double *const *d_bearings, const double *const *d_uv, const double *const *d_wpts,
                           int n_corr, float fx, float fy, float cx, float cy, double *h_seconds, int *h_accepted);

/* a4 for many cameras: FeatureTracker::fbKltTracking (src/slam/src/feature_tracker.cpp:5-111) of `count` independent
 * (previous, current) pyramid pairs in ONE launch; arguments per pair as alva_fbklt_track (d_prior[c] in/out), pyramids may differ
 * in size.  lanes_per_keypoint selects the wave layout: 5 (a lane per column pair of the 9x9 window, 12 keypoints per wave: the
 * fastest for large batches), 8, 16, 32 or 64 (alva_fbklt_track's layout); results are bit-identical for all of them.
 * Enqueue-only, like alva_fbklt_track. */
int alva_fbklt_track_batch(alva_ctx *ctx, const alva_pyramid *const *prev, const alva_pyramid *const *curr, int count,
                           int num_levels, float err_thresh, float fb_dist, int max_iters, float eps,
                           const float *const *d_pts, float *const *d_prior, uint8_t *const *d_status, const int *n,
                           int lanes_per_keypoint);

/* ---- a1 for a rig: VisualFrontend::trackMono of B lock-step cameras, one launch per stage for all of them ----------------
 * The per-frame path of src/slam/src/visual_frontend.cpp:83-150 -- preprocessImage (:672-698), kltTracking (:152-243),
 * computePose (:245-417); the detector is the keyframe branch's, not this path's -- for `cameras` independent cameras that deliver
 * their frames together: 5 launches build all gray images + LK pyramids, 1 launch tracks every camera's keypoints, 4 launches solve
 * every camera's P3P-LMedS -> PnP behind it, and one host synchronisation returns the poses.  Each camera has its own frame, keypoints
 * (n_pts[c] <= max_tracked), correspondences (n_corr[c] <= max_corr <= 7168) and state; results are identical to `cameras`
 * alva_frontend_track calls.  d_rgba / d_pts / d_bearings / d_uv / d_wpts are host arrays of `cameras` device pointers;
 * h_pose7 is [cameras][7] (written where status >= 1 and the solver produced a pose), h_pose_status [cameras] as alva_compute_pose. */
typedef struct alva_track_batch alva_track_batch;
int alva_track_batch_create(int device, int width, int height, int cameras, int max_tracked, int max_corr, alva_track_batch **out);
void alva_track_batch_destroy(alva_track_batch *tb);
int alva_track_batch_step(alva_track_batch *tb, const uint8_t *const *d_rgba, size_t rgba_pitch, const float *const *d_pts,
                          const int *n_pts, const double *const *d_bearings, const double *const *d_uv,
                          const double *const *d_wpts, const int *n_corr, float fx, float fy, float cx, float cy, double *h_pose7,
                          int *h_pose_status);
/* Optional: the keyframe branch's feature work for every camera, as alva_frontend_track does for one -- cv::ORB::detectAndCompute
 * (orb_features, 1.2, 8 levels, FAST 20) on the frame's gray image and a brute-force Hamming match against the camera's previous
 * descriptors, batched over the cameras on a third HIP stream (12 + 2 launches for all cameras).  Enable before the first step;
 * alva_track_batch_step_detect then also returns the per-camera keypoint counts, alva_track_batch_detections the device-resident
 * keypoints [n][6], descriptors [n][32] and matches (index / distance into the camera's previous descriptor set). */
int alva_track_batch_enable_detector(alva_track_batch *tb, int orb_features);
int alva_track_batch_step_detect(alva_track_batch *tb, const uint8_t *const *d_rgba, size_t rgba_pitch, const float *const *d_pts,
                                 const int *n_pts, const double *const *d_bearings, const double *const *d_uv,
                                 const double *const *d_wpts, const int *n_corr, float fx, float fy, float cx, float cy,
                                 double *h_pose7, int *h_pose_status, int *h_n_keypoints);
int alva_track_batch_detections(alva_track_batch *tb, int cam, const float **d_keypoints, const uint8_t **d_descriptors,
                                const int **d_match_idx, const int **d_match_dist);
/* device-resident kltTracking result of one camera from the last step: [n_pts][2] positions, [n_pts] status */
int alva_track_batch_results(alva_track_batch *tb, int cam, const float **d_tracked, const uint8_t **d_track_status);
/* Tuning knob of the tracking launch: lanes of a wavefront per keypoint, 5 (default, see alva_fbklt_track_batch), 8, 16, 32 or 64 (the single-camera kernel's layout).  Results are identical for all three.  The environment variable
 * ALVA_KLT_BATCH_LANES sets the default at creation. */
int alva_track_batch_set_klt_lanes(alva_track_batch *tb, int lanes);
/* steps done, and how often a camera had to be re-solved by the single-camera call because the first 128 samples of its P3P
 * stream held fewer than 100 non-degenerate ones (the reference keeps drawing, Lmeds.hpp:67-92) */
int alva_track_batch_stats(alva_track_batch *tb, long *frames, long *single_camera_fallbacks);
/* the context (stream) the batch runs on, e.g. for alva_prof_* or alva_ctx_sync */
alva_ctx *alva_track_batch_ctx(alva_track_batch *tb);

/* ---- a10-a13: local bundle adjustment ---------------------------------------------------------
 * Replaces the solve inside Optimizer::localBA (src/slam/src/optimizer.cpp:251-262 on the problem
 * built at :20-247): Levenberg-Marquardt + Huber, Schur complement on the point blocks,
 * anchored-inverse-depth (inv_depth=1, state.hpp:74 default) or XYZ points; cost functions
 * src/slam/src/ceres_parametrization.cpp:6-94,157-268.  Flat problem description (host arrays):
 *   h_poses[n_kf][7] in/out, h_kf_const[n_kf], h_calib[4],
 *   inv_depth=1: h_pt_anchor_kf[n_pt], h_pt_anchor_uv[n_pt][2], h_pt_param[n_pt]    in/out
 *   inv_depth=0: h_pt_param[n_pt][3] in/out
 *   h_obs_kf[n_obs], h_obs_pt[n_obs], h_obs_uv[n_obs][2]
 * Outputs: h_chi2[n_obs], h_depth_pos[n_obs] at the last evaluated point (what the reference's
 * outlier sweep reads, optimizer.cpp:266-309), h_info[0..3] = {#iterations, initial cost, final cost,
 * #successful steps}. */
int alva_local_ba(alva_ctx *ctx, int n_kf, double *h_poses, const uint8_t *h_kf_const, const double *h_calib,
                  int inv_depth, int n_pt, const int *h_pt_anchor_kf, const double *h_pt_anchor_uv,
                  double *h_pt_param, int n_obs, const int *h_obs_kf, const int *h_obs_pt,
                  const double *h_obs_uv, int max_iters, double function_tolerance, double huber_chi2,
                  double *h_chi2, uint8_t *h_depth_pos, double *h_info, int *h_ok);

/* The same solve for a caller that holds its observations GROUPED BY POINT (round 5: the map layer's localBA build emits them that
 * way): h_pt_ptr[n_pt + 1] delimits each point's residual blocks in h_obs_kf / h_obs_uv (anchored inverse depth only).  The
 * (observing keyframe, anchor keyframe) grouping that the camera-block assembly needs is built on the DEVICE (a stable counting
 * sort, identical to the host-built one: results are bit-identical to alva_local_ba on the same problem), and the outlier sweep's
 * test of Optimizer::localBA (src/slam/src/optimizer.cpp:266-309: chi2 > chi2_threshold, or the point behind the camera) comes back as
 * one BIT per residual block (h_bad_bits: (n_obs + 63) / 64 words; *h_n_bad their count) instead of the chi2 / depth arrays.
 * n_kf <= 32.  Synchronous. */
int alva_local_ba_csr(alva_ctx *ctx, int n_kf, double *h_poses, const uint8_t *h_kf_const, const double *h_calib, int n_pt,
                      const int *h_pt_ptr, const int *h_pt_anchor_kf, const double *h_pt_anchor_uv, double *h_pt_inv_depth, int n_obs,
                      const int *h_obs_kf, const double *h_obs_uv, int max_iters, double function_tolerance, double huber_chi2,
                      double chi2_threshold, unsigned long long *h_bad_bits, int *h_n_bad, double *h_info, int *h_ok);

/* `count` independent local-BA problems (anchored inverse depth) with ONE set of launches per LM iteration: a rig's cameras or a
 * server's sessions, each with its own keyframes / points / observations (ragged sizes).  Every kernel carries the problem in a grid
 * dimension -- the reduced camera systems are factored on `count` compute units at once, the Schur-complement GEMMs form one grouped
 * FP64-MFMA launch -- and the host reads ONE block of scalars per iteration and steps each problem's trust region separately (problems
 * stop at different iterations).  Every problem's result is BIT-IDENTICAL to its own alva_local_ba call.  Arguments: arrays of
 * `count` sizes / host pointers with alva_local_ba's meaning; h_calib[4] shared; h_info [count][4]; h_ok [count].  At most 23 free
 * keyframes per problem (the reduced system is factored in LDS). */
int alva_local_ba_batch(alva_ctx *ctx, int count, const int *n_kf, double *const *h_poses, const uint8_t *const *h_kf_const,
                        const double *h_calib, const int *n_pt, const int *const *h_pt_anchor_kf, const double *const *h_pt_anchor_uv,
                        double *const *h_pt_param, const int *n_obs, const int *const *h_obs_kf, const int *const *h_obs_pt,
                        const double *const *h_obs_uv, int max_iters, double function_tolerance, double huber_chi2, double *const *h_chi2,
                        uint8_t *const *h_depth_pos, double *h_info, int *h_ok);

/* ---- measured ceilings for the roofline lines of bench.py (MI355X_MICROARCH.md lists neither): the FP64 matrix rate of
 * v_mfma_f64_16x16x4_f64 in TFLOP/s and the plain integer VALU rate (xor / popcount-accumulate / add) in 1e12 lane-operations/s. */
int alva_microbench_peaks(alva_ctx *ctx, double *h_tflops_mfma_f64, double *h_tops_valu_int);
/* Launch latency for the end-to-end bound of a frame (bench.py "bounds"): microseconds per kernel of `chain` dependent empty
 * launches on the context's stream, and the round trip of one empty launch + stream synchronisation.  Synchronous. */
int alva_microbench_launch(alva_ctx *ctx, int chain, double *h_us_per_dependent_launch, double *h_us_launch_sync_roundtrip);
/* Debug (no reference counterpart): phase stamps of the pose kernels (100 MHz wall clock), recorded only when the process runs with
 * ALVA_KSTAMPS=1: 4096 x u64 -- k_p3p 8 per workgroup from entry 0, k_pnp sequentially from entry 2048; cleared by the call. */
int alva_debug_kstamps(unsigned long long *h_out);
/* Debug: with ALVA_KLT_STAMPS=1 in the environment the tracker launch of the System path (k_track_klt) records, per slot of the frame
 * container, its wall time in the kernel (100 MHz ticks, bits 0-31) | result code << 32 | tracked-from-projection << 36 | retried << 37;
 * copies the 16384 entries of the last launch to the host.  ALVA_ERR_STATE without the variable. */
int alva_debug_klt_stamps(unsigned long long *h_out16384);

/* f1: MapPoint's descriptor tables -- mapKeyframeDescriptors_, mapDescriptorsDist_, desc_ -- on the device, edited by
 * REPLAYING a log of MapPoint::addDesc (map_point.cpp:131-181), the descriptor half of MapPoint::removeObservedKeyframeId (:93-128),
 * the release when the last observation goes (:80-91) and "new map point" operations.  A store holds one fixed-size table per map point
 * SLOT (alva_medoid_table_bytes()); operations are 64-byte records (alva_medoid_op_bytes(): int op {0 add, 1 remove, 2 clear, 3 reset},
 * int keyframe, int rehash_to (bucket count of the rehash this insert triggers in the reference's unordered_map, 0 = none), int next
 * (index of the same map point's next operation, -1 = last), uint8 descriptor[32], 16 bytes padding).  alva_medoid_replay ENQUEUES
 * the log on the context's stream (one wavefront per listed map point: mp_slot[i] with its chain head first_op[i]; `slots` = highest
 * slot in use + 1); alva_medoid_export waits and returns per requested slot desc_ (32 B), !desc_.empty(), {#descriptors, keyframe
 * desc_ was taken from, overflow flag}; alva_medoid_dump copies one raw table (tests; layout in csrc/slam/medoid_table.hpp). */
typedef struct alva_medoid_store alva_medoid_store;
size_t alva_medoid_table_bytes(void);
size_t alva_medoid_op_bytes(void);
int alva_medoid_store_create(alva_ctx *ctx, alva_medoid_store **out);
void alva_medoid_store_destroy(alva_medoid_store *store);
int alva_medoid_replay(alva_medoid_store *store, int n_ops, const void *ops, int n_mp, const int *mp_slot, const int *first_op, int slots);
int alva_medoid_export(alva_medoid_store *store, int n, const int *mp_slot, uint8_t *h_desc32, uint8_t *h_valid, int *h_info3);
int alva_medoid_dump(alva_medoid_store *store, int mp_slot, void *h_table, size_t bytes);
/* the device array of tables (valid until the next alva_medoid_replay grows it; NULL before the first replay): kernels that read the
 * descriptors in place (alva_match_to_map_records) */
const void *alva_medoid_tables(alva_medoid_store *store);
/* The shared-map exchange's record block from the resident data (round 5): one thread per record slot 0 .. n_slots - 1 reads the
 * map-point record's header (csrc/slam/mp_rec.hpp; d_record_chunks as in alva_match_to_map_records) and the descriptor medoid of the
 * same slot's table; every 3-D point with a descriptor becomes one 64-byte row {int32 stream, int32 point id, f64 xyz[3], u8 desc[32]}
 * of d_out [capacity][64] (device), in no particular order; unused rows get point id -1.  *h_count = points found (> capacity: only
 * `capacity` of them were written -- an arbitrary subset).  Semantics of the exchange: MapManager::mergeMapPoints,
 * src/slam/src/map_manager.cpp:428-513 (parity unpinned: the reference has one map).  Synchronous. */
int alva_pack_map_records(alva_medoid_store *store, const void *const *d_record_chunks, int n_slots, int stream_id, int capacity,
                          uint8_t *d_out, int *h_count);

/* ---- §8(e) optional shared-map merge (north_star extension, PARITY UNPINNED: the reference has one map) -----------------------
 * n records sorted by (stream, point id): a record is absorbed by the earliest SURVIVING record of another stream within max_dist
 * (metres) whose descriptor is within max_hamming bits (smallest distance wins, earliest record on ties) -- the intent of
 * MapManager::mergeMapPoints (src/slam/src/map_manager.cpp:428-513) with mapMaxDescriptorDistance_ (state.hpp:60).  d_keep[i] = 1 for
 * survivors, d_absorbed_by[i] = index of the absorbing record or -1.  *h_rounds = fixed-point rounds taken.  Synchronous. */
int alva_fuse_map_points(alva_ctx *ctx, int n, const int *d_stream, const double *d_xyz, const uint8_t *d_desc, double max_dist,
                         int max_hamming, uint8_t *d_keep, int *d_absorbed_by, int *h_rounds);

#ifdef __cplusplus
}
#endif
#endif /* ALVAAR_HIP_H */
