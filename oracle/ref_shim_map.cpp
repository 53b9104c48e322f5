// TEST INFRASTRUCTURE ONLY (part of oracle/_ref/libalva_ref.so).  Builds a small map with the reference's OWN classes
// (Frame, MapPoint, MapManager, Mapper) from flat arrays and runs the real Mapper::matchToMap (mapper.cpp:354-588) on it,
// so that the restatement in alva_oracle.c is pinned against the reference itself.
#include <sstream>
#include <string>
#include <cstdint>
#include <cstring>
#include <vector>
#include <map>
#include <memory>
#include <unordered_map>
#include <unordered_set>
#include "frame.hpp"
#include "map_point.hpp"
#include "state.hpp"
#include "feature_extractor.hpp"
#define private public
#include "map_manager.hpp"
#include "mapper.hpp"
#undef private

namespace {
Sophus::SE3d se3_of(const double *p) {
    return Sophus::SE3d(Eigen::Quaterniond(p[6], p[3], p[4], p[5]), Eigen::Vector3d(p[0], p[1], p[2]));
}
void export_tcw(const Sophus::SE3d &Tcw, double *q4, double *t3) {
    const Eigen::Quaterniond q = Tcw.unit_quaternion();
    q4[0] = q.x(); q4[1] = q.y(); q4[2] = q.z(); q4[3] = q.w();
    for (int i = 0; i < 3; i++) t3[i] = Tcw.translation()(i);
}
}  // namespace

// Inputs (all flat):
//   calib[10] = fx fy cx cy k1 k2 p1 p2 imgW imgH ; cellSize
//   keyframes: nKf poses Twc (pose7), ids kfId[nKf]; the LAST keyframe is the frame that is matched
//   map points: nMp ids, wpt, is3d; per map point its observations obsPtr[nMp+1] -> (obsKf = index into keyframes, obsPx, obsDesc[32])
//               in ascending keyframe-id order; the first observation seeds the MapPoint
//   frame keypoints are the observations whose keyframe is the last one, added to the Frame in the order given by frameKpOrder
//   local[nLocal] = map point ids inserted into the unordered_set in this order
// Outputs: kfQ/kfT = Tcw quaternions (x y z w) / translations of every keyframe as the reference holds them; the frame's grid
//   (cellPtr[gridCells+1], cellKp = keypoint IDS in stored order) ; localOrder = iteration order of the unordered_set ;
//   matches (keypointId, mapPointId) pairs in std::map order.  Returns the number of matches.
extern "C" int ref_match_to_map(const double *calib, int cellSize, int nKf, const int *kfId, const double *kfPose, int nMp,
                                const int *mpId, const double *mpWpt, const uint8_t *mpIs3d, const int *obsPtr, const int *obsKf,
                                const float *obsPx, const uint8_t *obsDesc, int nFrameKp, const int *frameKpOrder, int numKeypoints3d,
                                int nLocal, const int *local, float maxProjErr, float distRatio, double *kfQ, double *kfT, int *gridCells,
                                int *numCellsW, int *cellPtr, int *cellKp, int *localOrder, int *matchKp, int *matchMp) {
    auto state = std::make_shared<State>(calib[8], calib[9], cellSize);
    auto cal = std::make_shared<CameraCalibration>(calib[0], calib[1], calib[2], calib[3], calib[4], calib[5], calib[6], calib[7], calib[8],
                                                   calib[9], 1.);
    auto currFrame = std::make_shared<Frame>(cal, (size_t) cellSize);
    std::shared_ptr<FeatureExtractor> fe;
    auto mm = std::make_shared<MapManager>(state, currFrame, fe);
    std::vector<std::shared_ptr<Frame>> kfs((size_t) nKf);
    for (int k = 0; k < nKf; k++) {
        kfs[(size_t) k] = std::make_shared<Frame>(cal, (size_t) cellSize);
        kfs[(size_t) k]->keyframeId_ = kfId[k];
        kfs[(size_t) k]->id_ = kfId[k];
        kfs[(size_t) k]->setTwc(se3_of(kfPose + 7 * k));
        export_tcw(kfs[(size_t) k]->getTcw(), kfQ + 4 * k, kfT + 3 * k);
        mm->mapKeyframes_.emplace(kfId[k], kfs[(size_t) k]);
    }
    const int last = nKf - 1;
    // keypoints of the older keyframes (any order), of the frame in the requested order
    for (int m = 0; m < nMp; m++)
        for (int o = obsPtr[m]; o < obsPtr[m + 1]; o++)
            if (obsKf[o] != last) kfs[(size_t) obsKf[o]]->addKeypoint(cv::Point2f(obsPx[2 * o], obsPx[2 * o + 1]), mpId[m]);
    for (int i = 0; i < nFrameKp; i++) {
        const int o = frameKpOrder[i];   // observation index (its keyframe is the last one)
        int m = 0;
        while (!(obsPtr[m] <= o && o < obsPtr[m + 1])) m++;
        kfs[(size_t) last]->addKeypoint(cv::Point2f(obsPx[2 * o], obsPx[2 * o + 1]), mpId[m]);
    }
    for (int m = 0; m < nMp; m++) {
        std::shared_ptr<MapPoint> mp;
        for (int o = obsPtr[m]; o < obsPtr[m + 1]; o++) {
            cv::Mat d(1, 32, CV_8U);
            std::memcpy(d.data, obsDesc + 32 * (size_t) o, 32);
            const int kid = kfId[obsKf[o]];
            if (!mp) mp = std::make_shared<MapPoint>(mpId[m], kid, d);
            else {
                mp->addObservedKeyframeId(kid);
                mp->addDesc(kid, d);
            }
        }
        if (!mp) mp = std::make_shared<MapPoint>(mpId[m], 0);
        if (mpIs3d[m]) {
            mp->setPoint(Eigen::Vector3d(mpWpt[3 * m], mpWpt[3 * m + 1], mpWpt[3 * m + 2]));
            mp->is3d_ = true;
        }
        mm->mapMapPoints_.emplace(mpId[m], mp);
    }
    Frame &frame = *kfs[(size_t) last];
    frame.numKeypoints3d_ = (size_t) numKeypoints3d;
    *gridCells = (int) frame.gridKeypointsIds_.size();
    *numCellsW = (int) frame.numCellsW_;
    int w = 0;
    for (size_t c = 0; c < frame.gridKeypointsIds_.size(); c++) {
        cellPtr[c] = w;
        for (int id: frame.gridKeypointsIds_[c]) cellKp[w++] = id;
    }
    cellPtr[frame.gridKeypointsIds_.size()] = w;
    std::unordered_set<int> localSet;
    for (int i = 0; i < nLocal; i++) localSet.insert(local[i]);
    int li = 0;
    for (int id: localSet) localOrder[li++] = id;
    Mapper mapper(state, mm, currFrame);
    const std::map<int, int> res = mapper.matchToMap(frame, maxProjErr, distRatio, localSet);
    int n = 0;
    for (const auto &kv: res) {
        matchKp[n] = kv.first;
        matchMp[n] = kv.second;
        n++;
    }
    return n;
}

// MapPoint's descriptor bookkeeping, driven operation by operation on ONE reference MapPoint (map_point.cpp:73-181):
//   op 0: addObservedKeyframeId(kf) + addDesc(kf, desc)      op 1: removeObservedKeyframeId(kf)
// After every operation: desc_ (32 bytes, zeros when empty), !desc_.empty(), the bucket count of mapKeyframeDescriptors_ (the caller
// derives the rehash points from it), its size, and -- in ITERATION order -- the keys and their mapDescriptorsDist_ sums (cap entries
// per operation).  Pins medoid.hip / medoid_table.hpp against the reference's own class.
extern "C" int ref_mappoint_desc_ops(int first_kf, const uint8_t *first_desc, int nOps, const int *op, const int *kf, const uint8_t *desc, int cap,
                                     uint8_t *outMedoid, uint8_t *outHas, int *outBuckets, int *outCount, int *outKeys, float *outDist) {
    auto mat_of = [](const uint8_t *d) {
        cv::Mat m(1, 32, CV_8U);
        std::memcpy(m.data, d, 32);
        return m;
    };
    MapPoint mp = first_desc ? MapPoint(1, first_kf, mat_of(first_desc)) : MapPoint(1, first_kf);
    for (int i = 0; i < nOps; i++) {
        if (op[i] == 0) {
            mp.addObservedKeyframeId(kf[i]);
            mp.addDesc(kf[i], mat_of(desc + 32 * (size_t) i));
        } else {
            mp.removeObservedKeyframeId(kf[i]);
        }
        const bool has = !mp.desc_.empty();
        outHas[i] = has;
        if (has) std::memcpy(outMedoid + 32 * (size_t) i, mp.desc_.data, 32);
        else std::memset(outMedoid + 32 * (size_t) i, 0, 32);
        outBuckets[i] = (int) mp.mapKeyframeDescriptors_.bucket_count();
        outCount[i] = (int) mp.mapKeyframeDescriptors_.size();
        int j = 0;
        for (const auto &e: mp.mapKeyframeDescriptors_) {
            if (j >= cap) return -1;
            outKeys[(size_t) i * cap + j] = e.first;
            outDist[(size_t) i * cap + j] = mp.mapDescriptorsDist_.at(e.first);
            j++;
        }
    }
    return 0;
}
