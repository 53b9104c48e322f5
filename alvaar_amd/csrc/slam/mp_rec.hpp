// The map-point RECORD: what the map layer keeps about one map point outside its descriptor table, as ONE fixed-size plain struct that
// both sides can read -- the host's bookkeeping loops (slam/*.cpp) and the device's kernels (match_to_map.hip, ba_build.hip).
//
// Why a record and not an object with vectors: every keyframe step walks thousands of map points in an order the hardware cannot
// predict (the frame's keypoints, the local map, the local BA's point set: MapManager::updateFrameCovisibility map_manager.cpp:83-164,
// Mapper::matchToMap mapper.cpp:354-588, Optimizer::localBA optimizer.cpp:20-247) and reads, per point, the observing keyframes
// (MapPoint::observedKeyframeIds_, a std::set<int>) and what each of those keyframes holds about the point (the keypoint's pixel in that
// keyframe, Frame::mapKeypoints_).  Round 4 kept that in two heap vectors per map point and FLATTENED it into arrays for every stage call
// (33 000 observations x 45 bytes per keyframe, 1.5 MB assembled and uploaded).  Here the same facts live in records of a fixed stride in
// PINNED host memory (Stages::mp_arena_chunk): the host edits them in place, address = f(slot) so a loop can prefetch them, and a kernel
// GATHERS the records a stage call needs straight out of host memory (one wavefront per record, zero-copy reads of header + live
// entries) -- the call ships one slot number per map point instead of its observations.
//
// Semantics.  An entry exists for keyframe kf while any of its flags is set:
//   MPF_OBS   kf is in observedKeyframeIds_                                            (map_point.hpp:60, addObservedKeyframeId / remove...)
//   MPF_INKF  keyframe kf holds a keypoint with this map point's id; px / unpx = that keypoint's positions (a keyframe's keypoints never
//             move after MapManager::addKeyframe's copy, map_manager.cpp:243-252)
//   MPF_DESC  (free for a user of these helpers; the map layer stopped setting it at the end of round 5: the keys of
//             mapKeyframeDescriptors_ and their bytes live in MapPt::kf_desc / the side arena indexed by the key's slot, slam.hpp)
// Entries are sorted by keyframe id (a std::set<int> walks ascending keys whatever its history).  MP_ENT_CAP bounds them: the mapper's
// window is 30 keyframes + keyframe 0 + the one being created; an insert beyond the capacity sets `overflow` and the frame fails.
#pragma once
#include <cstdint>
#include <cstring>

namespace alva_slam {

constexpr int MP_ENT_CAP = 40;
constexpr int MP_CHUNK_SHIFT = 12, MP_CHUNK = 1 << MP_CHUNK_SHIFT;   // records per arena chunk (4 MB of pinned memory each)
constexpr uint8_t MPF_OBS = 1, MPF_INKF = 2, MPF_DESC = 4;

struct ObsEnt {
    int kf;
    uint8_t flags, pad[3];
    float px[2], unpx[2];
};
static_assert(sizeof(ObsEnt) == 24, "ObsEnt layout");

struct MpRec {
    double X[3];              // worldPoint_
    double inv_depth;         // anchor inverse depth (-1: none yet)
    int id;                   // map point id, -1 = free record
    int anchor_kf;            // keyframeId_
    uint8_t is3d, has_desc, observed, n_ent;   // is3d_; !desc_.empty(); isObserved_; live entries
    uint8_t n_obs, overflow, pad8[2];          // entries with MPF_OBS
    int dev_slot;             // = the record's own slot (the descriptor table of the same map point has the same index)
    int pad32[3];
    ObsEnt ent[MP_ENT_CAP];
};
static_assert(sizeof(MpRec) == 1024, "MpRec stride");

// ---- host-side editing (the device only reads)
inline void rec_init(MpRec &r, int id, int kf, int slot) {
    r.X[0] = r.X[1] = r.X[2] = 0.;
    r.inv_depth = -1.;
    r.id = id;
    r.anchor_kf = kf;
    r.is3d = 0; r.has_desc = 0; r.observed = 1; r.n_ent = 0;
    r.n_obs = 0; r.overflow = 0; r.pad8[0] = r.pad8[1] = 0;
    r.dev_slot = slot;
}
inline int rec_find(const MpRec &r, int kf) {   // index of the entry of keyframe kf, or -1
    for (int i = 0; i < r.n_ent; i++) {
        if (r.ent[i].kf == kf) return i;
        if (r.ent[i].kf > kf) break;
    }
    return -1;
}
// 32 descriptor bytes, HOST ONLY (the device reads descriptors from its own tables).  rec_slot / rec_clear_flag can keep an array of them
// parallel to MpRec::ent (same index, shifted with the entries) for a caller that wants per-entry payload; the map layer indexes its
// descriptor bytes by the key's slot in its key table instead (nothing to shift) and passes no side array.
typedef uint8_t DescBytes[32];

// the entry of keyframe kf, created (no flags yet, positions zero) if absent; nullptr when the record is full
inline ObsEnt *rec_slot(MpRec &r, int kf, DescBytes *side = nullptr) {
    int i = r.n_ent;
    while (i > 0 && r.ent[i - 1].kf > kf) i--;
    if (i > 0 && r.ent[i - 1].kf == kf) return &r.ent[i - 1];
    if (r.n_ent >= MP_ENT_CAP) {
        r.overflow = 1;
        return nullptr;
    }
    std::memmove(&r.ent[i + 1], &r.ent[i], (size_t) (r.n_ent - i) * sizeof(ObsEnt));
    if (side) std::memmove(&side[i + 1], &side[i], (size_t) (r.n_ent - i) * sizeof(DescBytes));
    r.n_ent++;
    ObsEnt &e = r.ent[i];
    e.kf = kf;
    e.flags = 0;
    e.pad[0] = e.pad[1] = e.pad[2] = 0;
    e.px[0] = e.px[1] = e.unpx[0] = e.unpx[1] = 0.f;
    return &e;
}
inline void rec_clear_flag(MpRec &r, int i, uint8_t flag, DescBytes *side = nullptr) {   // clears `flag` of entry i; the entry goes with its last flag
    ObsEnt &e = r.ent[i];
    if ((e.flags & flag) && flag == MPF_OBS) r.n_obs--;
    e.flags = (uint8_t) (e.flags & ~flag);
    if (!e.flags) {
        std::memmove(&r.ent[i], &r.ent[i + 1], (size_t) (r.n_ent - i - 1) * sizeof(ObsEnt));
        if (side) std::memmove(&side[i], &side[i + 1], (size_t) (r.n_ent - i - 1) * sizeof(DescBytes));
        r.n_ent--;
    }
}

// MapPoint::isBad (map_point.cpp:183-202) on the record alone (the hot loops do not touch the map point's object)
inline bool rec_is_bad(MpRec &r) {
    if (r.n_obs < 2) {
        if (!r.observed && r.is3d) {
            r.is3d = 0;
            return true;
        }
    }
    if (r.n_obs == 0 && !r.observed) {
        r.is3d = 0;
        return true;
    }
    return false;
}
// the keypoint the map point has in keyframe kf (null: that keyframe holds none)
inline const ObsEnt *rec_in_kf(const MpRec &r, int kf) {
    const int i = rec_find(r, kf);
    return i >= 0 && (r.ent[i].flags & MPF_INKF) ? &r.ent[i] : nullptr;
}

// a snapshot of the observing keyframes (MapPoint::getObservedKeyframeIds returns a COPY of the set: loops that edit while they walk)
struct ObsList {
    int n = 0;
    int kf[MP_ENT_CAP];
    const int *begin() const { return kf; }
    const int *end() const { return kf + n; }
    size_t size() const { return (size_t) n; }
    bool count(int k) const {
        for (int i = 0; i < n; i++)
            if (kf[i] == k) return true;
        return false;
    }
};
inline ObsList rec_observers(const MpRec &r) {
    ObsList l;
    for (int i = 0; i < r.n_ent; i++)
        if (r.ent[i].flags & MPF_OBS) l.kf[l.n++] = r.ent[i].kf;
    return l;
}

}  // namespace alva_slam
