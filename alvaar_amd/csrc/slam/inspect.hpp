// Read-only views of the map layer's state for the C ABI (alva_system_debug_*) and for the tests: flat arrays out, the same
// layout the reference-side shim (oracle/ref_shim_system.cpp) produces for the reference's System, so that the two can be
// compared field by field.
#pragma once
#include <algorithm>
#include <cstring>
#include "slam.hpp"

namespace alva_slam {

// out[0..15]: frame id, keyframe id, numKeypoints, 2d, 3d, occupied cells, #keyframes, #map points, ready, p3pReq, poseFailed,
// next keyframe id, next map point id, |local map| of the current frame, |covisible| of the current frame, max keypoints
inline void inspect_state(const Slam &s, int *out) {
    const FrameRec &f = *s.cur;
    out[0] = f.id; out[1] = f.kfid; out[2] = (int) f.n_kps; out[3] = (int) f.n_2d; out[4] = (int) f.n_3d; out[5] = (int) f.n_occupied;
    out[6] = (int) s.keyframes.size(); out[7] = (int) s.map_points.size(); out[8] = s.ready_for_init; out[9] = s.p3p_req;
    out[10] = s.pose_failed; out[11] = s.next_kf_id; out[12] = s.next_mp_id; out[13] = (int) f.local_map.size();
    out[14] = (int) f.covisible.size(); out[15] = s.cfg.max_keypoints;
}

inline int inspect_frame(const FrameRec &f, int cap, int *ids, float *px, float *unpx, uint8_t *is3d, uint8_t *has_desc) {
    int n = 0;
    for (const auto &e: f.kps) {  // container order
        if (n < cap) {
            const KeyPt &k = e.second;
            if (ids) ids[n] = k.id;
            if (px) { px[2 * n] = k.px[0]; px[2 * n + 1] = k.px[1]; }
            if (unpx) { unpx[2 * n] = k.unpx[0]; unpx[2 * n + 1] = k.unpx[1]; }
            if (is3d) is3d[n] = k.is3d;
            if (has_desc) has_desc[n] = k.has_desc;
        }
        n++;
    }
    return n;
}

inline int inspect_keyframe_ids(const Slam &s, int cap, int *ids) {
    std::vector<int> v;
    for (const auto &e: s.keyframes) v.push_back(e.first);
    std::sort(v.begin(), v.end());
    for (size_t i = 0; i < v.size() && (int) i < cap; i++) ids[i] = v[i];
    return (int) v.size();
}

// info[0..5] = frame id, numKeypoints, 2d, 3d, |covisible|, |local map|
inline int inspect_keyframe(const Slam &s, int kfid, double *pose7, int *info, int cap, int *ids, float *px, uint8_t *is3d) {
    auto it = s.keyframes.find(kfid);
    if (it == s.keyframes.end()) return -1;
    const FrameRec &f = *it->second;
    if (pose7) se3_to_pose7(f.Twc, pose7);
    if (info) {
        info[0] = f.id; info[1] = (int) f.n_kps; info[2] = (int) f.n_2d; info[3] = (int) f.n_3d; info[4] = (int) f.covisible.size();
        info[5] = (int) f.local_map.size();
    }
    return inspect_frame(f, cap, ids, px, nullptr, is3d, nullptr);
}

inline int inspect_covisibility(const Slam &s, int kfid, int cap, int *pairs) {
    const FrameRec *f = s.cur.get();
    if (kfid >= 0) {
        auto it = s.keyframes.find(kfid);
        if (it == s.keyframes.end()) return -1;
        f = it->second.get();
    }
    int n = 0;
    for (const auto &e: f->covisible) {
        if (n < cap) { pairs[2 * n] = e.first; pairs[2 * n + 1] = e.second; }
        n++;
    }
    return n;
}

// ascending id; flags[5i..] = is3d, observed, #observing keyframes, anchor keyframe, #descriptors; desc = desc_ (the descriptor medoid,
// fetched from the stages' tables: pending edits are replayed first)
inline int inspect_map_points(Slam &s, int cap, int *ids, double *xyz, int *flags, double *inv_depth, uint8_t *desc) {
    std::vector<int> v;
    for (const auto &e: s.map_points) v.push_back(e.first);
    std::sort(v.begin(), v.end());
    std::vector<int> slots;
    for (size_t i = 0; i < v.size() && (int) i < cap; i++) {
        const MapPt &m = *s.map_points.at(v[i]);
        ids[i] = v[i];
        if (xyz) std::memcpy(xyz + 3 * i, m.r->X, 24);
        if (flags) {
            flags[5 * i] = m.r->is3d; flags[5 * i + 1] = m.r->observed; flags[5 * i + 2] = (int) m.n_obs(); flags[5 * i + 3] = m.r->anchor_kf;
            flags[5 * i + 4] = (int) m.kf_desc.size();
        }
        if (inv_depth) inv_depth[i] = m.r->inv_depth;
        slots.push_back(m.dev_slot);
    }
    if (desc && !slots.empty()) {
        s.flush_medoids();
        std::vector<uint8_t> valid(slots.size());
        std::vector<int> info(3 * slots.size());
        const int rc = s.st->medoid_export((int) slots.size(), slots.data(), desc, valid.data(), info.data());
        if (rc) return rc;
        for (size_t i = 0; i < slots.size(); i++) {
            const MapPt &m = *s.map_points.at(v[i]);
            // the map layer's own view of the same table must agree with the stages' (key count; desc_ present) and nothing may have overflowed
            if (info[3 * i + 2] || info[3 * i] != (int) m.kf_desc.size() || (valid[i] != 0) != (m.r->has_desc != 0)) return -5;
            if (!valid[i]) std::memset(desc + 32 * i, 0, 32);
        }
    }
    return (int) v.size();
}

// Utils::toPoseArray (utils.cpp:3-27)
inline void pose_to_array(const SE3 &T, float *p) {
    double R[9];
    quat_to_rot(T.q, R);
    for (int r = 0; r < 3; r++) {
        for (int c = 0; c < 3; c++) p[4 * r + c] = (float) R[3 * r + c];
        p[4 * r + 3] = 0.f;
    }
    p[12] = (float) T.t[0]; p[13] = (float) T.t[1]; p[14] = (float) T.t[2]; p[15] = 1.f;
}

}  // namespace alva_slam
