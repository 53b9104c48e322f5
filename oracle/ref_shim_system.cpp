// TEST INFRASTRUCTURE ONLY (part of oracle/_ref/libalva_ref.so) -- never linked into or called from alvaar_amd/.
//
// ref_system_*: the reference's OWN `System` (src/slam/src/system.cpp) -- VisualFrontend, MapManager, Mapper, Optimizer and
// every vendored library under it -- driven frame by frame with the determinism switches SURVEY.md §8(c) lists:
//   * explicit timestamps: System::processCameraPose(image, timestamp) is called directly (system.cpp:156-175); the public
//     findCameraPose reads system_clock (system.cpp:114) and does nothing else besides cvtColor + toPoseArray, both done here;
//   * state_->multiViewRandomEnabled_ = false: OpenGV samples with its fixed seed (SampleConsensusProblem.hpp:43-46);
//   * cv::setNumThreads(1) (ref_shim.cpp);
//   * Ceres' wall-clock caps (optimizer.cpp:258,322; multi_view_geometry.cpp:185) never fire: the library is linked with
//     -Wl,--wrap=gettimeofday and ref_freeze_clock(1) makes Ceres' WallTimeInSeconds() (wall_time.cc:62-64) constant.
// No reference source is modified or copied; private members are reached with `#define private public`.
// The cell size (system.cpp:15 hard-codes 40) can be overridden to run BASELINE configs[1] (cell 12 => 2120 keypoints): the
// members are then constructed exactly as System::configure does (system.cpp:13-40) with the other cell size.
#include <sys/time.h>
#include <cstdint>
#include <cstring>
#include <algorithm>
#include <chrono>
#include <deque>
#include <fstream>
#include <iomanip>
#include <iostream>
#include <map>
#include <memory>
#include <queue>
#include <set>
#include <sstream>
#include <string>
#include <unordered_map>
#include <unordered_set>
#include <vector>
#include <Eigen/Core>
#include <Eigen/Geometry>
#include <Eigen/LU>
#include <sophus/se3.hpp>
#include <opencv2/core.hpp>
#include <opencv2/core/eigen.hpp>
#include <opencv2/calib3d.hpp>
#include <opencv2/highgui.hpp>
#include <opencv2/imgproc.hpp>
#include <ceres/ceres.h>
#define private public  // only the reference's own headers are included under this
#include "system.hpp"
#undef private
#include "utils.hpp"
#include <opencv2/imgproc.hpp>

static int g_freeze = 0;
extern "C" int __real_gettimeofday(struct timeval *tv, void *tz);
extern "C" int __wrap_gettimeofday(struct timeval *tv, void *tz) {
    if (g_freeze) {
        if (tv) {
            tv->tv_sec = 1000000;
            tv->tv_usec = 0;
        }
        return 0;
    }
    return __real_gettimeofday(tv, tz);
}

namespace {
struct RefSys {
    System sys;
    int w = 0, h = 0;
};
void pose7_of(const Sophus::SE3d &T, double *p) {
    const Eigen::Quaterniond q = T.unit_quaternion();
    p[0] = T.translation()(0); p[1] = T.translation()(1); p[2] = T.translation()(2);
    p[3] = q.x(); p[4] = q.y(); p[5] = q.z(); p[6] = q.w();
}
}  // namespace

extern "C" {

void ref_freeze_clock(int on) { g_freeze = on; }

void *ref_system_create(int w, int h, double fx, double fy, double cx, double cy, double k1, double k2, double p1, double p2, int cellSize,
                        int claheEnabled, int doRandom) {
    auto *r = new RefSys();
    r->w = w;
    r->h = h;
    System &s = r->sys;
    if (cellSize <= 0 || cellSize == 40) {
        s.configure(w, h, fx, fy, cx, cy, k1, k2, p1, p2);  // the shipped configuration, through the reference's own method
    } else {
        // system.cpp:13-40 with another frameMaxCellSize_
        s.state_ = std::make_shared<State>(w, h, cellSize);
        s.state_->debug_ = false;
        s.state_->claheEnabled_ = false;
        s.state_->mapKeyframeFilteringRatio_ = 0.95;
        s.state_->p3pEnabled_ = true;
        s.cameraCalibration_ = std::make_shared<CameraCalibration>(fx, fy, cx, cy, k1, k2, p1, p2, w, h, 20);
        s.currFrame_ = std::make_shared<Frame>(s.cameraCalibration_, s.state_->frameMaxCellSize_);
        s.featureExtractor_ = std::make_shared<FeatureExtractor>(s.state_->extractorMaxQuality_);
        s.featureTracker_ = std::make_shared<FeatureTracker>(s.state_->trackerMaxIterations_, s.state_->trackerMaxPxPrecision_);
        s.mapManager_ = std::make_shared<MapManager>(s.state_, s.currFrame_, s.featureExtractor_);
        s.mapper_ = std::make_shared<Mapper>(s.state_, s.mapManager_, s.currFrame_);
        s.visualFrontend_ = std::make_unique<VisualFrontend>(s.state_, s.currFrame_, s.mapManager_, s.mapper_, s.featureTracker_);
    }
    s.state_->multiViewRandomEnabled_ = doRandom != 0;
    s.state_->claheEnabled_ = claheEnabled != 0;
    return r;
}

void ref_system_destroy(void *p) { delete static_cast<RefSys *>(p); }

void ref_system_reset(void *p) { static_cast<RefSys *>(p)->sys.reset(); }

// System::findCameraPose (system.cpp:106-121) with the timestamp as an argument.  pose16 as the reference writes it; pose7 =
// currFrame_->getTwc() in double (t, qx qy qz qw).
int ref_system_find_camera_pose(void *p, const uint8_t *rgba, double timestamp, float *pose16, double *pose7) {
    auto *r = static_cast<RefSys *>(p);
    System &s = r->sys;
    cv::Mat image = cv::Mat(r->h, r->w, CV_8UC4, const_cast<uint8_t *>(rgba));
    cv::Mat gray;
    cv::cvtColor(image, gray, cv::COLOR_RGBA2GRAY);
    const int status = s.processCameraPose(gray, timestamp);
    if (pose16) Utils::toPoseArray(s.currFrame_->getTwc(), pose16);
    if (pose7) pose7_of(s.currFrame_->getTwc(), pose7);
    return status;
}

// The pose array System::findCameraPoseWithIMU hands back (system.cpp:66-103) for an IMU quaternion (w, x, y, z) and a translation, with the
// reference's own classes: Eigen quaternion (w, -x, y, z) -> rotation matrix -> inverse, Sophus::SE3d, Utils::toPoseArray.  (The call
// itself reads the wall clock for its timestamp, system.cpp:87, so the tests drive processCameraPose with explicit timestamps and build the
// expected array from this.)
void ref_imu_pose(const double *imu_wxyz, const double *translation3, float *pose16) {
    Eigen::Quaterniond orientation(imu_wxyz[0], -imu_wxyz[1], imu_wxyz[2], imu_wxyz[3]);
    Eigen::Matrix3d qwc = orientation.toRotationMatrix().inverse();
    Sophus::SE3d Twc(qwc, Eigen::Vector3d(0.0, 0.0, 0.0));
    Twc.translation() = Eigen::Vector3d(translation3[0], translation3[1], translation3[2]);
    Utils::toPoseArray(Twc, pose16);
}

int ref_system_find_plane(void *p, float *pose16, int numIterations) {
    auto *r = static_cast<RefSys *>(p);
    cv::Mat m = r->sys.processPlane(r->sys.mapManager_->getCurrentFrameMapPoints(), r->sys.currFrame_->getTwc(), numIterations);
    if (m.empty()) return 0;
    Utils::toPoseArray(m, pose16);
    return 1;
}

// out[0..15]: frame id, keyframe id, numKeypoints, 2d, 3d, occupied cells, #keyframes in the map, #map points in the map,
// slamReadyForInit, p3pReq, poseFailedCounter, next keyframe id, next map point id, |localMapPointIds_| of the current frame,
// |covisibleKeyframeIds_| of the current frame, frameMaxNumKeypoints_
void ref_system_state(void *p, int *out) {
    System &s = static_cast<RefSys *>(p)->sys;
    const Frame &f = *s.currFrame_;
    out[0] = f.id_;
    out[1] = f.keyframeId_;
    out[2] = (int) f.numKeypoints_;
    out[3] = (int) f.numKeypoints2d_;
    out[4] = (int) f.numKeypoints3d_;
    out[5] = (int) f.numOccupiedCells_;
    out[6] = (int) s.mapManager_->mapKeyframes_.size();
    out[7] = (int) s.mapManager_->mapMapPoints_.size();
    out[8] = s.state_->slamReadyForInit_;
    out[9] = s.visualFrontend_->p3pReq_;
    out[10] = s.visualFrontend_->poseFailedCounter_;
    out[11] = s.mapManager_->numKeyframeIds_;
    out[12] = s.mapManager_->numMapPointIds_;
    out[13] = (int) f.localMapPointIds_.size();
    out[14] = (int) f.covisibleKeyframeIds_.size();
    out[15] = s.state_->frameMaxNumKeypoints_;
}

static int dump_frame(const Frame &f, int cap, int *ids, float *px, float *unpx, uint8_t *is3d, uint8_t *hasDesc) {
    int n = 0;
    for (const auto &it: f.mapKeypoints_) {  // the container's own iteration order
        if (n < cap) {
            const Keypoint &k = it.second;
            if (ids) ids[n] = k.keypointId_;
            if (px) { px[2 * n] = k.px_.x; px[2 * n + 1] = k.px_.y; }
            if (unpx) { unpx[2 * n] = k.unpx_.x; unpx[2 * n + 1] = k.unpx_.y; }
            if (is3d) is3d[n] = k.is3d_;
            if (hasDesc) hasDesc[n] = !k.desc_.empty();
        }
        n++;
    }
    return n;
}

int ref_system_frame_keypoints(void *p, int cap, int *ids, float *px, float *unpx, uint8_t *is3d, uint8_t *hasDesc) {
    return dump_frame(*static_cast<RefSys *>(p)->sys.currFrame_, cap, ids, px, unpx, is3d, hasDesc);
}

// ids of the keyframes in the map, ascending
int ref_system_keyframe_ids(void *p, int cap, int *ids) {
    System &s = static_cast<RefSys *>(p)->sys;
    std::vector<int> v;
    for (const auto &kv: s.mapManager_->mapKeyframes_) v.push_back(kv.first);
    std::sort(v.begin(), v.end());
    for (size_t i = 0; i < v.size() && (int) i < cap; i++) ids[i] = v[i];
    return (int) v.size();
}

// one keyframe: pose7 (Twc), info[0..5] = frame id, numKeypoints, 2d, 3d, |covisible|, |localMapPointIds_|; keypoints in container order
int ref_system_keyframe(void *p, int kfid, double *pose7, int *info, int cap, int *ids, float *px, uint8_t *is3d) {
    System &s = static_cast<RefSys *>(p)->sys;
    auto it = s.mapManager_->mapKeyframes_.find(kfid);
    if (it == s.mapManager_->mapKeyframes_.end()) return -1;
    const Frame &f = *it->second;
    pose7_of(f.getTwc(), pose7);
    if (info) {
        info[0] = f.id_;
        info[1] = (int) f.numKeypoints_;
        info[2] = (int) f.numKeypoints2d_;
        info[3] = (int) f.numKeypoints3d_;
        info[4] = (int) f.covisibleKeyframeIds_.size();
        info[5] = (int) f.localMapPointIds_.size();
    }
    return dump_frame(f, cap, ids, px, nullptr, is3d, nullptr);
}

// covisibility map of a keyframe (kfid >= 0) or of the current frame (kfid < 0): pairs (keyframe id, score), ascending
int ref_system_covisibility(void *p, int kfid, int cap, int *pairs) {
    System &s = static_cast<RefSys *>(p)->sys;
    const Frame *f = s.currFrame_.get();
    if (kfid >= 0) {
        auto it = s.mapManager_->mapKeyframes_.find(kfid);
        if (it == s.mapManager_->mapKeyframes_.end()) return -1;
        f = it->second.get();
    }
    int n = 0;
    for (const auto &kv: f->covisibleKeyframeIds_) {
        if (n < cap) { pairs[2 * n] = kv.first; pairs[2 * n + 1] = kv.second; }
        n++;
    }
    return n;
}

// map points, ascending id: xyz, flags[0] = is3d, [1] = isObserved, [2] = #observing keyframes, [3] = anchor keyframe id,
// [4] = #descriptors; invDepth
int ref_system_map_points(void *p, int cap, int *ids, double *xyz, int *flags, double *invDepth, uint8_t *desc) {
    System &s = static_cast<RefSys *>(p)->sys;
    std::vector<int> v;
    for (const auto &kv: s.mapManager_->mapMapPoints_) v.push_back(kv.first);
    std::sort(v.begin(), v.end());
    for (size_t i = 0; i < v.size() && (int) i < cap; i++) {
        const MapPoint &m = *s.mapManager_->mapMapPoints_.at(v[i]);
        ids[i] = v[i];
        if (xyz) for (int c = 0; c < 3; c++) xyz[3 * i + c] = m.point3d_(c);
        if (flags) {
            flags[5 * i] = m.is3d_;
            flags[5 * i + 1] = m.isObserved_;
            flags[5 * i + 2] = (int) m.observedKeyframeIds_.size();
            flags[5 * i + 3] = m.keyframeId_;
            flags[5 * i + 4] = (int) m.mapKeyframeDescriptors_.size();
        }
        if (invDepth) invDepth[i] = m.invDepth_;
        if (desc) {
            if (!m.desc_.empty()) std::memcpy(desc + 32 * i, m.desc_.ptr<uint8_t>(0), 32);
            else std::memset(desc + 32 * i, 0, 32);
        }
    }
    return (int) v.size();
}

double ref_system_max_quality(void *p) { return static_cast<RefSys *>(p)->sys.featureExtractor_->maxQuality_; }

}  // extern "C"
