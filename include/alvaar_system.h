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
 * Scope (DESIGN.md "System surface"): the WHOLE per-frame path of System::processCameraPose (system.cpp:156-175) --
 * VisualFrontend::track/process (motion-model priors, two-pass forward-backward KLT, P3P-LMedS + robust PnP, pose-failure
 * handling, five-point initialisation, keyframe policy), MapManager::createKeyframe (grid detection + ORB description, descriptor
 * medoids), Mapper::processNewKeyframe (triangulation, covisibility, guided matching to the local map + map-point merging,
 * local bundle adjustment with outlier sweep and write-back, map-point and keyframe culling).  The bookkeeping is host code
 * (alvaar_amd/csrc/slam/), every numeric stage runs on the GPU through include/alvaar_hip.h.  Lens distortion (k1 k2 p1 p2) and CLAHE
 * are wired (camera_calibration.cpp:34-72, visual_frontend.cpp:678-681).  find_plane runs the plane fit the reference intends
 * (alva_find_plane; the reference function itself computes on reinterpreted memory, so its parity is unpinned).
 * Known deviations: a timestamp older than the previous one resets the tracker (status 2) instead of exit(-1)
 * (visual_frontend.hpp:46-50); P3P-LMedS is solved on the first 19000 3-D keypoints of a frame when there are more (a 3840x2160 frame has ~10 k).
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
/* System::configure (system.cpp:13-40): cell size 40, CLAHE off, clock-seeded sampling.  A failed call leaves the object
 * unconfigured (every later call returns ALVA_ERR_ARG) and alva_system_last_error() says why. */
int alva_system_configure(alva_system *sys, int width, int height, double fx, double fy, double cx, double cy, double k1,
                          double k2, double p1, double p2);
/* The same with the three settings System::configure hard-codes (system.cpp:15-19; state.hpp:67): the keypoint cell size
 * (12 at 640x480 = the 2000-keypoint workload of BASELINE configs[1]), CLAHE, and random_sampling = 0 for the fixed sample
 * streams of OpenGV (seed 12345) that the differential tests use. */
int alva_system_configure_ex(alva_system *sys, int width, int height, double fx, double fy, double cx, double cy, double k1,
                             double k2, double p1, double p2, int cell_size, int clahe_enabled, int random_sampling);
void alva_system_reset(alva_system *sys);
/* System::findCameraPose (system.cpp:106-121).  h_rgba: width*height*4 bytes, caller-owned, read-only to the callee and not retained
 * beyond the call; h_pose: float[16].  By default the frame is copied through a pinned staging buffer. */
int alva_system_find_camera_pose(alva_system *sys, const uint8_t *h_rgba, float *h_pose);
/* ---- many sessions on a few host threads (no reference counterpart: the reference is one System per process / worker).  A group
 * owns n_threads worker threads; alva_system_group_find_camera_pose_device runs ONE frame of each of `count` configured systems
 * (session i: frame d_rgba[i] in device memory, pose -> h_poses[16 i ..], status / error code -> h_status[i]) and returns when all are
 * done.  Sessions are fibers on the workers: a session's waits for the GPU hand the thread to the worker's next session, so the
 * threads only execute map-layer work and the GPU sees the sessions' streams side by side.  Every session's results equal its solo run
 * bit for bit.  Systems must not be used from other threads during the call. */
typedef struct alva_system_group alva_system_group;
int alva_system_group_create(int n_threads, alva_system_group **out);
void alva_system_group_destroy(alva_system_group *group);
/* Streams for sessions to SHARE: the GPU's command processor slows down sharply beyond a handful of concurrently active hardware queues
 * (measured: empty launches 2.5 us each on 1 - 4 streams, 26 us each on 8), so a group's sessions run on a few streams rather than one
 * each.  alva_system_group_stream returns (creating it on first use) the group's stream number `index`; alva_system_set_stream makes a
 * system build its stages on that stream at its next alva_system_configure*.  The systems must be destroyed before the group. */
int alva_system_group_stream(alva_system_group *group, int device, int index, void **out_stream);
int alva_system_set_stream(alva_system *sys, void *hip_stream);
int alva_system_group_find_camera_pose_device(alva_system_group *group, int count, alva_system *const *systems, const uint8_t *const *d_rgba,
                                              double timestamp_ms, float *h_poses, int *h_status);
/* LOCK-STEP LAUNCHES (default on; ALVA_GROUP_LOCKSTEP=0 or alva_system_group_set_lockstep(group, 0) turns them off).  The group owns
 * `lanes` streams of its own (default 4; ALVA_GROUP_LANES / alva_system_group_set_lanes; 0 = none); session i of a call runs on worker
 * i % n_threads and belongs to lane (i + i / n_threads) % lanes.  The seven launches of a tracking frame -- gray + pyramid level 0, pyramid,
 * slot table, fb-KLT, compaction, P3P-LMedS, PnP -- are issued ONCE PER KIND for all sessions of a lane (blockIdx.y = session; the
 * argument blocks ride in the kernel arguments) on the lane's stream, by the thread whose session completes the set, instead of seven per
 * session: S side-by-side chains of small kernels saturate the GPU's command path and wave slots long before its arithmetic.  A worker
 * holds sessions of different lanes and does the host half of one lane's sessions while another lane's launches run.  Everything else a
 * session launches (keyframe stages, local BA) stays on the session's own stream.  Results are unchanged bit for bit (same arithmetic
 * per slot / hypothesis / correspondence).  alva_system_group_launch_stats: out2 = {combined launches issued, session launches they
 * carried}. */
int alva_system_group_set_lanes(alva_system_group *group, int lanes);
/* summed over the workers since the last reset: out4 = {seconds inside group steps, seconds inside session slices that did more than
 * look at a completion word, slices, slices that did work} -- how much of the workers' time is map-layer work and how much is waiting */
int alva_system_group_time_stats(alva_system_group *group, double *out4, int reset);
int alva_system_group_set_lockstep(alva_system_group *group, int on);
int alva_system_group_launch_stats(alva_system_group *group, long *out2);
/* Optional, for a caller that reuses ONE frame buffer the way src/system.js reuses its memImg (:63-67, :175): page-lock and map
 * `bytes` (>= width*height*4, 16-byte aligned) at h_rgba.  Frames passed from inside the registered range are then read in place
 * over PCIe by the gray / pyramid kernel (no staging copy, no copy command); every find_camera_pose* call still returns only after the
 * GPU has finished reading the buffer.  The registration is the caller's promise that the memory stays allocated until
 * alva_system_unregister_frame_buffer / alva_system_configure / alva_system_destroy; one buffer per system (a second call replaces
 * the first).  Must be called after alva_system_configure. */
int alva_system_register_frame_buffer(alva_system *sys, const uint8_t *h_rgba, size_t bytes);
int alva_system_unregister_frame_buffer(alva_system *sys);
/* The ONE frame buffer of src/system.js (memImg, :63-67) placed in DEVICE memory that the host can write (round 5): the whole of an
 * MI355X's memory is visible to the CPU over the PCIe BAR, write-combined, so memImg.write(frame) (:175) -- one copy of the frame, which
 * the caller performs anyway -- IS the upload: *h_writable receives a pointer the caller stores the frame through (ordinary stores /
 * memcpy; never read through it: a load crosses the bus), and a frame passed to alva_system_find_camera_pose* from inside this buffer
 * is read by the gray / pyramid kernel straight out of HBM, exactly like alva_system_find_camera_pose_device's.  Freed by
 * alva_system_configure / alva_system_destroy.  ALVA_ERR_STATE where such memory cannot be had (use alva_system_register_frame_buffer). */
int alva_system_alloc_frame_buffer(alva_system *sys, size_t bytes, uint8_t **h_writable);
/* The same with the frame's timestamp (milliseconds) as an argument instead of the system clock (system.cpp:114): the
 * constant-velocity motion model (visual_frontend.hpp:11-68) is the only consumer. */
int alva_system_find_camera_pose_ts(alva_system *sys, const uint8_t *h_rgba, double timestamp_ms, float *h_pose);
/* The same for a frame that already lives in DEVICE memory of the system's GPU (width*height*4 bytes, 16-byte aligned): no PCIe
 * upload.  For capture pipelines that deliver into HBM, and for bench.py's timed loop (frames resident in HBM). */
int alva_system_find_camera_pose_device(alva_system *sys, const uint8_t *d_rgba, double timestamp_ms, float *h_pose);
/* Look-ahead for frames in device memory (no reference counterpart: the reference is handed one frame per call).  Called BEFORE
 * alva_system_find_camera_pose_device(frame k), it names frame k+1 (device memory that stays valid and unchanged until that call): the
 * gray image and LK pyramid of frame k+1 are then enqueued right behind frame k's pose kernels (they run while the host does call k's
 * bookkeeping and builds call k+1's slot table, a window in which the GPU is otherwise idle), and call k+1 finds them
 * ready when it passes the same pointer (any other pointer: the look-ahead is dropped and the frame is built as usual).  Results are
 * identical with and without hints.  One hint per call; NULL cancels. */
int alva_system_hint_next_frame_device(alva_system *sys, const uint8_t *d_rgba_next);
/* System::findCameraPoseWithIMU (system.cpp:57-104).  h_imu: [qw,qx,qy,qz,n, n x {ts,gx,gy,gz,ax,ay,az}]. Always returns 1. */
int alva_system_find_camera_pose_with_imu(alva_system *sys, const uint8_t *h_rgba, const double *h_imu, float *h_pose);
/* the same with the caller's timestamp in milliseconds instead of the wall clock (system.cpp:87), as alva_system_find_camera_pose_ts */
int alva_system_find_camera_pose_with_imu_ts(alva_system *sys, const uint8_t *h_rgba, const double *h_imu, double timestamp_ms, float *h_pose);
/* System::findPlane (system.cpp:123-137): 1 on success, 0 otherwise (needs >= 32 observed 3-D points). */
int alva_system_find_plane(alva_system *sys, float *h_pose, int num_iterations);
/* System::getFramePoints (system.cpp:139-154): writes x,y int pairs of the current 2-D (not yet triangulated)
 * keypoints, at most 2048 points (the caller's buffer is uint32[4096], src/system.js:64); returns their count. */
int alva_system_get_frame_points(alva_system *sys, int *h_points);

/* ids + pixel positions + 3-D flag of the current frame's keypoints, in the frame container's order; returns their number */
int alva_system_get_keypoints(alva_system *sys, int *h_ids, float *h_px, uint8_t *h_is3d, int cap);

/* ---- inspection (tests, tools): flat views of the map layer's state, same layouts as oracle/ref_shim_system.cpp produces for
 * the reference's System.  out16: frame id, keyframe id, #keypoints, #2-D, #3-D, occupied cells, #keyframes, #map points, map
 * initialised, p3pReq_, poseFailedCounter_, next keyframe id, next map point id, |local map|, |covisible|, frameMaxNumKeypoints_. */
int alva_system_debug_state(alva_system *sys, int *out16);
int alva_system_debug_pose7(alva_system *sys, double *pose7_twc, double *init_pose7);
int alva_system_debug_frame_keypoints(alva_system *sys, int cap, int *ids, float *px, float *unpx, uint8_t *is3d, uint8_t *has_desc);
int alva_system_debug_keyframe_ids(alva_system *sys, int cap, int *ids);
int alva_system_debug_keyframe(alva_system *sys, int kfid, double *pose7, int *info6, int cap, int *ids, float *px, uint8_t *is3d);
int alva_system_debug_covisibility(alva_system *sys, int kfid, int cap, int *pairs);
int alva_system_debug_map_points(alva_system *sys, int cap, int *ids, double *xyz, int *flags5, double *inv_depth, uint8_t *desc);
int alva_system_debug_counters(alva_system *sys, long *out3 /* local-BA solves, map-point merges, culled keyframes */);
/* ---- the optional SHARED-MAP MERGE across sessions, applied to a session (north_star's extra; the reference has one map: parity unpinned).
 * A merge round (alvaar_amd/multi.py: pack -> ONE all_gather over the process group -> alva_fuse_map_points) decides, for map points of
 * DIFFERENT streams that coincide -- same world position within 5 cm, descriptors within 51 bits; this presumes that the streams' maps live
 * in ONE world frame (a rig with known extrinsics, a shared initialisation, or maps registered beforehand), which the round verifies with
 * a similarity fit over the fused pairs before it applies anything -- which point of the shared map each absorbed point IS.  Applying it:
 *   alva_system_set_shared_ids      map point `local_id[i]` of this session is point (shared_stream[i], shared_id[i]) of the shared map;
 *                                   returns how many of the ids exist.  The table follows MapManager::mergeMapPoints (the survivor
 *                                   inherits it) and forgets culled points; alva_system_get_shared_ids reads it (ascending local id).
 *   alva_system_merge_map_points    MapManager::mergeMapPoints(prev_id, new_id) (map_manager.cpp:428-513) on this session's own map: two
 *                                   of its points that the round found to be the SAME shared point become one; 1 = merged, 0 = not
 *                                   merged: the reference's early return (a point is gone or the survivor is not 3-D), or the two
 *                                   points are observed together by the current frame or by a keyframe -- the reference's routine
 *                                   only ever meets points that are not (a new keyframe's keypoint against a local-map point the
 *                                   frame does not see) and would leave that frame with a keypoint of a vanished map point; such a
 *                                   pair keeps its two ids and the same shared id. */
int alva_system_merge_map_points(alva_system *sys, int prev_id, int new_id);
int alva_system_set_shared_ids(alva_system *sys, int n, const int *local_id, const int *shared_stream, const int *shared_id);
int alva_system_get_shared_ids(alva_system *sys, int cap, int *local_id, int *shared_stream, int *shared_id);
/* This session's 3-D map points as the exchange's record block, written ON THE DEVICE from the resident map (alva_pack_map_records):
 * d_out [capacity][64] in the system's GPU memory; *h_count = 3-D points with a descriptor (when it exceeds `capacity` the block holds an
 * arbitrary subset: export through alva_system_debug_map_points instead, which keeps the oldest).  Pending descriptor edits are replayed
 * first.  ALVA_ERR_STATE without a device-resident map (nothing was ever mapped). */
int alva_system_pack_map_records(alva_system *sys, int stream_id, int capacity, uint8_t *d_out, int *h_count);
/* fb-KLT work since the last reset: out2[0] = keypoint-levels (LK passes over one pyramid level, forwards + the backward pass, from
 * the per-slot result codes), out2[1] = slots handed to the tracking steps */
int alva_system_debug_klt_work(alva_system *sys, long *out2, int reset);
/* wall-clock seconds per section of the frame loop since the last reset: upload + pyramid enqueue, slot gathering, tracking step,
 * tracker bookkeeping, wait for the pose, pose bookkeeping + keyframe decision, keyframe creation, mapping (incl. local BA) */
int alva_system_debug_timing(alva_system *sys, double *out8, int reset);
/* finer laps of the keyframe path + element counts (profiling aid; indices documented in alvaar_amd/system.py: timing_fine) */
int alva_system_debug_timing_fine(alva_system *sys, double *out32, int reset);
/* finer split of the keyframe sections: [0] prepareFrame [1] describe tracked keypoints [2] grid detection [3] describe + undistort new
 * keypoints [4] map insertion + keyframe copy | [5] triangulation [6] covisibility [7] local-map matching incl. flattening and merges
 * [8] optimize (local BA + culling) ; inside them: [9] the matchToMap stage call [10] the local-BA stage calls */
int alva_system_debug_timing_keyframe(alva_system *sys, double *out16, int reset);
/* (the differential tests' one behaviour-changing hook, alva_system_debug_set_init_pose, is declared in alvaar_system_testing.h, not here) */
const char *alva_system_last_error(void);

#ifdef __cplusplus
}

namespace alva {
/* Drop-in for the reference's `class System` (same method names and argument order). */
class System {
public:
    System() { status_ = alva_system_create(0, &s_); }
    ~System() { alva_system_destroy(s_); }
    System(const System &) = delete;
    System &operator=(const System &) = delete;
    void configure(int imageWidth, int imageHeight, double fx, double fy, double cx, double cy, double k1, double k2, double p1,
                   double p2) {
        status_ = s_ ? alva_system_configure(s_, imageWidth, imageHeight, fx, fy, cx, cy, k1, k2, p1, p2) : status_;
    }
    /* 0 when construction and the last configure succeeded (the reference's methods are void; errors surface here) */
    int status() const { return status_; }
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
    int status_ = 0;
};
}  // namespace alva
#endif
#endif /* ALVAAR_SYSTEM_H */
