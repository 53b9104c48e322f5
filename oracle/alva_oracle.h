/* TEST INFRASTRUCTURE ONLY -- CPU restatement of the reference hot path (see oracle/README.md).
 * Parity status: PINNED against oracle/_ref/libalva_ref.so (the compiled reference) by
 * tests/test_oracle_vs_ref.py and against tests/golden/ fixtures generated from it. */
#ifndef ALVA_ORACLE_H
#define ALVA_ORACLE_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* a2 */
void orc_rgba2gray(const uint8_t *rgba, int w, int h, uint8_t *gray);

/* a3: level sizes; returns number of levels built (OpenCV stops early, lkpyramid.cpp:811-816) */
int orc_pyramid_dims(int w, int h, int win, int max_level, int *dims /* [2*(max_level+1)] */);
/* gray_out[l]: (h_l+2win) x (w_l+2win) u8 contiguous; deriv_out[l]: same x 2 int16 */
int orc_build_pyramid(const uint8_t *gray, int w, int h, int win, int max_level, uint8_t **gray_out, int16_t **deriv_out);

/* a4 (win must be 9, the reference's constant kltWinSizeWH_) */
int orc_lk(const uint8_t *prevGray, const uint8_t *nextGray, int w, int h, int win, int pyrLevelsBuilt, int numLevels,
           int maxIters, float eps, const float *pts, float *next, uint8_t *status, float *err, int n);
int orc_fbklt(const uint8_t *prevGray, const uint8_t *currGray, int w, int h, int win, int pyrLevelsBuilt, int numLevels,
              float errThresh, float fbDist, int maxIters, float eps, const float *pts, float *prior, uint8_t *status, int n);

/* a5 */
void orc_cell_mineig(const uint8_t *gray, int w, int h, int x0, int y0, int cell, uint8_t *blurOut, float *eig);
void orc_corner_subpix(const uint8_t *gray, int w, int h, float *pts, int n);
int orc_detect_grid(const uint8_t *gray, int w, int h, int cell, const float *occupied, int nOcc, int roiX, int roiY, int roiW,
                    int roiH, double *maxQuality, float *outPts, int cap);

/* a5' */
int orc_fast(const uint8_t *gray, int w, int h, int threshold, int *xy, int *score, int cap);
int orc_orb_detect_and_compute(const uint8_t *gray, int w, int h, int nfeatures, float scaleFactor, int nlevels, int fastThreshold,
                               int doDescribe, float *kp /* [cap][6] */, uint8_t *desc, int cap);

/* a6 */
void orc_orb_blur(const uint8_t *gray, int w, int h, uint8_t *out /* w*h */);
/* the detector's image pyramid (cv::ORB's INTER_LINEAR_EXACT resize chain), levels concatenated; returns the byte count */
long orc_orb_pyramid(const uint8_t *gray, int w, int h, float scaleFactor, int nlevels, uint8_t *out, int *dims /* [nlevels][2] */);
void orc_describe(const uint8_t *gray, int w, int h, const float *pts, int n, uint8_t *desc /* n*32 */, uint8_t *valid);

/* a7 */
int orc_hamming256(const uint8_t *a, const uint8_t *b);
void orc_bf_match_hamming(const uint8_t *q, int nq, const uint8_t *t, int nt, int *idx, int *dist);

/* a8 */
int orc_p3p_lmeds(const double *bv, const double *wpt, int n, int maxIterations, float errorThreshold, uint32_t seed, float fx,
                  float fy, double *R_out /*9, row-major*/, double *t_out, int *outliers, int *nOutliers);
int orc_p3p_draw_samples(int n, int count, uint32_t seed, int *samples /* count*4 */);

/* a9 */
int orc_pnp_refine(const double *uv, const double *wpt, int n, double *pose7, int maxIterations, float chi2th, int useRobust,
                   int applyL2AfterRobust, float fx, float fy, float cx, float cy, int *outliers, int *nOutliers, double *info /*[8]*/);

/* a10-a13 (same flat description as ref_local_ba in ref_shim.cpp / alva_local_ba in include/alvaar_hip.h) */
int orc_local_ba(int nKf, double *poses, const uint8_t *kfConst, const double *calib, int invDepth, int nPt, const int *ptAnchorKf,
                 const double *ptAnchorUv, double *ptParam, int nObs, const int *obsKf, const int *obsPt, const double *obsUv,
                 int maxIterations, double functionTolerance, double huberChi2, double *chi2, uint8_t *depthPos, double *info /*[9]*/);

/* f2b: two-view map initialisation, MultiViewGeometry::compute5ptEssentialMatrix (multi_view_geometry.cpp:225-320) = OpenGV
 * RANSAC over Nister's five-point solver + forward-difference Levenberg-Marquardt refinement; restated in
 * alva_oracle_relpose.c (file:line citations there).  Models are R (row-major 3x3) followed by t: X1 = R X2 + t. */
int orc_sturm_roots(const double *coeffs, int ncoef, double *roots);
void orc_nister_compose_a(const double *EE /* [9][4] */, double *A /* [10][20] */);
int orc_nister_nullspace(const double *bv1, const double *bv2, double *EE /* [9][4] */);
int orc_fivept_nister(const double *bv1, const double *bv2, double *E /* [<=10][9] */);
int orc_relpose_model(const double *bv1, const double *bv2, int n, const int *idx8, double *model12);
void orc_relpose_scores(const double *bv1, const double *bv2, int n, const double *model12, double *scores);
void orc_relpose_optimize(const double *bv1, const double *bv2, int n, const int *inliers, int nIn, const double *model12, double *out12,
                          int *info3 /* LM outer iterations, status, function evaluations */);
int orc_relpose_draw_samples(int n, int count, uint32_t seed, int *samples8);
int orc_relpose_ransac(const double *bv1, const double *bv2, int n, int maxIterations, float errorThreshold, uint32_t seed, float fx,
                       float fy, double *ransacModel12, uint8_t *inlierMask, int *info2 /* iterations, inliers */);
int orc_compute_5pt(const double *bv1, const double *bv2, int n, int maxIterations, float errorThreshold, int optimize, uint32_t seed,
                    float fx, float fy, double *R_out, double *t_out, int *outliers, int *nOutliers);

/* f3: the INTENDED algorithm of System::processPlane (system.cpp:177-342) -- PARITY UNPINNED, see alva_oracle_plane.c. */
int orc_find_plane(const double *pts, int n, const double *pose7_twc, const int *samples3, int numIterations, float *out16);
int orc_find_plane_margins(const double *pts, int n, const int *samples3, int numIterations, float *out3);


/* f2a: the numeric part of Mapper::triangulateTemporal per keypoint (mapper.cpp:246-287): OpenGV triangulate2
 * (opengv/src/triangulation/methods.cpp:67-90), cheirality and reprojection gates.
 * T: nGroups x 36 doubles = { R_lr[9], t_lr[3], R_rl[9], t_rl[3], R_wl[9], t_wl[3] } (row-major; l = the keyframe that first
 * observed the point, r = the new keyframe, w = world).  status: 0 accepted, 1 behind a camera (z < 0.1), 2 reprojection error. */
void orc_triangulate(int n, const double *T, const int *group, const double *bvl, const double *bvr, const float *unpxl,
                     const float *unpxr, double fx, double fy, double cx, double cy, float maxReprojErr, double *lpt, double *wpt,
                     double *invDepth, uint8_t *status, double *parallax);

/* f4a: cv::createCLAHE(clipLimit, Size(tilesX, tilesY))->apply on an 8-bit image (imgproc/src/clahe.cpp:120-420), as
 * VisualFrontend::preprocessImage calls it when claheEnabled_ (visual_frontend.cpp:16-18, :678-681). */
void orc_clahe(const uint8_t *src, int w, int h, double clipLimit, int tilesX, int tilesY, uint8_t *dst);

/* f4b: CameraCalibration::undistortImagePoint (camera_calibration.cpp:56-72) = cv::undistortPoints(pts, out, K, D, R = K)
 * with D = (k1, k2, p1, p2), 5 fixed iterations (calib3d/src/undistort.dispatch.cpp:384-556), and
 * CameraCalibration::projectCamToImageDist (:34-54) = cv::projectPoints of (x/z, y/z, 1) ROUNDED TO FLOAT with zero
 * rvec / tvec (calib3d/src/calibration.cpp:522-). */
void orc_undistort_points(const float *px, int n, double fx, double fy, double cx, double cy, const double *dist4, float *out);
void orc_project_dist(const double *camPts, int n, double fx, double fy, double cx, double cy, const double *dist4, float *out);

/* f1: Mapper::matchToMap (src/slam/src/mapper.cpp:354-588) on a flattened, CONSISTENT map (every keypoint id has its map point,
 * every observing keyframe exists and holds the keypoint -- the reference's repair branches :459-463, :500-509 never fire).
 *   calib[10] = fx fy cx cy k1 k2 p1 p2 imgW imgH ; the frame's keypoint grid as the reference stores it: cellPtr/cellMp (map point
 *   INDEX of each stored keypoint, in stored order), numCellsW, gridCells ; keyframes' T_cw as unit quaternion (x y z w) + translation ;
 *   map points: world point, is3d, observations obsPtr -> (obsKf ascending keyframe index, obsPx, obsDesc) ; frameKf = index of the
 *   keyframe being matched ; local[] = map point indices in the iteration order of the reference's unordered_set.
 * out: matchOfMp[m] = index of the local map point matched to the frame keypoint of map point m, or -1.  Returns #matches. */
int orc_match_to_map(const double *calib, int cellSize, int numCellsW, int gridCells, const int *cellPtr, const int *cellMp, int nKf,
                     const double *kfQ, const double *kfT, int nMp, const double *mpWpt, const uint8_t *mpIs3d, const int *obsPtr,
                     const int *obsKf, const float *obsPx, const uint8_t *obsDesc, int frameKf, int numKeypoints3d, int nLocal,
                     const int *local, float maxProjErr, float distRatio, int *matchOfMp);
/* live-map variant: mpHasDesc[m] = !desc_.empty(), obsHasDesc[o] = the observation's keyframe has a descriptor (NULL = all do) */
int orc_match_to_map_flags(const double *calib, int cellSize, int numCellsW, int gridCells, const int *cellPtr, const int *cellMp, int nKf,
                           const double *kfQ, const double *kfT, int nMp, const double *mpWpt, const uint8_t *mpIs3d,
                           const uint8_t *mpHasDesc, const int *obsPtr, const int *obsKf, const float *obsPx, const uint8_t *obsDesc,
                           const uint8_t *obsHasDesc, int frameKf, int numKeypoints3d, int nLocal, const int *local, float maxProjErr,
                           float distRatio, int *matchOfMp);

#ifdef __cplusplus
}
#endif
#endif
