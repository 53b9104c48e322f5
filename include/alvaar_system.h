/*
 * alvaar_system.h -- the reference's public surface (src/slam/src/system.hpp:24-38, bound to JS at
 * src/slam/src/embind.cpp:9-19) as a C ABI with POINTER-typed arguments, plus `alva::System`, a header-only C++
 * class with the reference's method names whose `int`-typed twins keep the wasm32 calling convention
 * (src/slam/src/system.cpp:59-61,108-109: heap byte offsets passed as int).
 *
 * Status codes of find_camera_pose (system.cpp:163-174): 1 = pose valid, 2 = tracker reset this frame,
 * 3 = still initialising.  The pose is written even when the status is not 1 (system.cpp:118).
 * Pose layout (src/slam/src/utils.cpp:3-27): p[0..2] = R row 0, p[4..6] = R row 1, p[8..10] = R row 2,
 * p[12..14] = t, p[3] = p[7] = p[11] = 0, p[15] = 1 (Twc).
 *
 * Scope (DESIGN.md "System surface"): the per-frame loop of SURVEY.md §3.2 runs on the GPU (gray -> pyramid -> fb-KLT ->
 * P3P-LMedS -> PnP), and so does the cold start: keyframe 0 on the first frame, the parallax gate and the five-point
 * initialisation of VisualFrontend::checkReadyForInit (unit baseline), triangulation of every new keyframe's 2-D keypoints
 * against the keyframe that first saw them (Mapper::triangulateTemporal), new keyframes by checkNewKeyframeRequired with grid
 * detection + ORB description in the free cells.  The reference's map layer above that (matching to the local map, local-BA
 * scheduling, keyframe / map-point culling) is not mirrored: alva_local_ba / alva_match_to_map exist in
 * alvaar_hip.h for a host that keeps that graph.  alva_system_set_map_points lets a host attach its own 3-D points instead of
 * the two-view initialisation.  find_plane runs the plane fit the reference intends (alva_find_plane; the reference function itself
 * computes on reinterpreted memory, so its parity is unpinned) on the current frame's 3-D keypoints.
 */
#ifndef ALVAAR_SYSTEM_H
#define ALVAAR_SYSTEM_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct alva_system alva_system;

int alva_system_create(int device, alva_system **out);
void alva_system_destroy(alva_system *sys);
/* System::configure (system.cpp:13-40).  Non-zero distortion coefficients are rejected for now. */
int alva_system_configure(alva_system *sys, int width, int height, double fx, double fy, double cx, double cy, double k1,
                          double k2, double p1, double p2);
void alva_system_reset(alva_system *sys);
/* System::findCameraPose (system.cpp:106-121).  h_rgba: width*height*4 bytes, caller-owned; h_pose: float[16]. */
int alva_system_find_camera_pose(alva_system *sys, const uint8_t *h_rgba, float *h_pose);
/* System::findCameraPoseWithIMU (system.cpp:57-104).  h_imu: [qw,qx,qy,qz,n, n x {ts,gx,gy,gz,ax,ay,az}]. Always returns 1. */
int alva_system_find_camera_pose_with_imu(alva_system *sys, const uint8_t *h_rgba, const double *h_imu, float *h_pose);
/* System::findPlane (system.cpp:123-137): 1 on success, 0 otherwise (needs >= 32 observed 3-D points). */
int alva_system_find_plane(alva_system *sys, float *h_pose, int num_iterations);
/* System::getFramePoints (system.cpp:139-154): writes x,y int pairs of the current 2-D (not yet triangulated)
 * keypoints, at most 2048 points (the caller's buffer is uint32[4096], src/system.js:64); returns their count. */
int alva_system_get_frame_points(alva_system *sys, int *h_points);

/* Bootstrap until the mapper rows are built: attach world points to current keypoints by keypoint id.
 * alva_system_get_keypoints returns ids + pixel positions of the current frame's keypoints (capacity cap). */
int alva_system_get_keypoints(alva_system *sys, int *h_ids, float *h_px, uint8_t *h_is3d, int cap);
int alva_system_set_map_points(alva_system *sys, const int *h_ids, const double *h_xyz, int n);
int alva_system_set_pose(alva_system *sys, const double *h_pose7);
const char *alva_system_last_error(void);

#ifdef __cplusplus
}

namespace alva {
/* Drop-in for the reference's `class System` (same method names and argument order). */
class System {
public:
    System() { alva_system_create(0, &s_); }
    ~System() { alva_system_destroy(s_); }
    System(const System &) = delete;
    System &operator=(const System &) = delete;
    void configure(int imageWidth, int imageHeight, double fx, double fy, double cx, double cy, double k1, double k2, double p1,
                   double p2) {
        alva_system_configure(s_, imageWidth, imageHeight, fx, fy, cx, cy, k1, k2, p1, p2);
    }
    void reset() { alva_system_reset(s_); }
    /* native, pointer-typed */
    int findCameraPose(const uint8_t *imageRGBA, float *pose) { return alva_system_find_camera_pose(s_, imageRGBA, pose); }
    int findCameraPoseWithIMU(const uint8_t *imageRGBA, const double *imu, float *pose) {
        return alva_system_find_camera_pose_with_imu(s_, imageRGBA, imu, pose);
    }
    int findPlane(float *location, int numIterations) { return alva_system_find_plane(s_, location, numIterations); }
    int getFramePoints(int *points) { return alva_system_get_frame_points(s_, points); }
    /* wasm32 calling convention of the reference (pointers as int heap offsets) */
    int findCameraPose(int imageRGBADataPtr, int posePtr) {
        return findCameraPose(reinterpret_cast<const uint8_t *>((uintptr_t) (uint32_t) imageRGBADataPtr),
                              reinterpret_cast<float *>((uintptr_t) (uint32_t) posePtr));
    }
    int findCameraPoseWithIMU(int imageRGBADataPtr, int imuDataPtr, int posePtr) {
        return findCameraPoseWithIMU(reinterpret_cast<const uint8_t *>((uintptr_t) (uint32_t) imageRGBADataPtr),
                                     reinterpret_cast<const double *>((uintptr_t) (uint32_t) imuDataPtr),
                                     reinterpret_cast<float *>((uintptr_t) (uint32_t) posePtr));
    }
    int findPlane(int locationPtr, int numIterations) {
        return findPlane(reinterpret_cast<float *>((uintptr_t) (uint32_t) locationPtr), numIterations);
    }
    int getFramePoints(int pointsPtr) { return getFramePoints(reinterpret_cast<int *>((uintptr_t) (uint32_t) pointsPtr)); }
    alva_system *handle() { return s_; }

private:
    alva_system *s_ = nullptr;
};
}  // namespace alva
#endif
#endif /* ALVAAR_SYSTEM_H */
