// Host-side map layer behind alva_system_* (include/alvaar_system.h): the bookkeeping the reference keeps in
// Frame / MapPoint / MapManager / Mapper / Optimizer / VisualFrontend (src/slam/src/*.cpp, SURVEY.md §1 layer L2) --
// keyframes, map points, covisibility, the per-frame state machine -- with every numeric stage behind `Stages`.
//
// Container choice is part of the contract, not a style matter: the reference iterates std::unordered_map<int, ...> /
// std::unordered_set<int> and that order reaches the solvers (the P3P sample stream indexes the keypoints in container order,
// visual_frontend.cpp:275-298; matchToMap breaks ties by the local map's order, mapper.cpp:395, :571-577; the gauge of the
// local BA is fixed on the first keyframes of an unordered_map, optimizer.cpp:234-247).  The same libstdc++ containers
// receiving the same sequence of insert / erase / clear / copy operations iterate in the same order, so the state below uses
// exactly those containers and every mutation follows the reference's sequence.
#pragma once
#if defined(__SANITIZE_ADDRESS__)
#include <sanitizer/asan_interface.h>
#endif
#include <cstdlib>
#include <new>
#include <map>
#include <memory>
#include <memory_resource>
#include <set>
#include <thread>
#include <unordered_map>
#include <unordered_set>
#include <vector>
#include "flat_hash.hpp"
#include "mp_rec.hpp"
#include "se3.hpp"
#include "stages.hpp"

namespace alva_slam {

struct Desc {
    uint8_t b[32];
};

// struct Keypoint, frame.hpp:17-38
struct KeyPt {
    int id = -1;
    float px[2] = {0, 0};
    float unpx[2] = {0, 0};
    double bv[3] = {0, 0, 0};
    Desc desc{};
    bool has_desc = false;
    bool is3d = false;
};

// mapKeypoints_ (std::unordered_map<int, Keypoint>, frame.hpp:170): the ids in a FlatHash -- libstdc++'s iteration order on flat arrays,
// flat_hash.hpp -- whose per-element tag is the keypoint's 3-D flag, and the keypoints themselves in a parallel array indexed by slot.
// Three loops of every keyframe step walk ALL keypoints of every covisible keyframe only to look at (id, is3d)
// (updateFrameCovisibility map_manager.cpp:116-140, the free-keyframe sweep of localBA optimizer.cpp:60-100, the keyframe filter
// mapper.cpp:94-128): they read 12-byte slots; a keyframe (a COPY of the frame, map_manager.cpp:243-252) is three vector copies.
// References to keypoints are invalidated by an insert (the reference's nodes are stable): no caller keeps one across an insert.
class KpTable {
public:
    FlatHash<FlatNoValue> ids;   // key = keypoint id, tag = is3d
    std::vector<KeyPt> kp;       // by slot

    size_t size() const { return ids.size(); }
    bool empty() const { return ids.empty(); }
    size_t count(int id) const { return ids.count(id); }
    size_t bucket_count() const { return ids.bucket_count(); }
    void clear() {
        ids.clear();
        kp.clear();
    }
    KeyPt *find_ptr(int id) {
        const int s = ids.find_slot(id);
        return s == FlatHash<FlatNoValue>::END ? nullptr : &kp[(size_t) s];
    }
    const KeyPt *find_ptr(int id) const { return const_cast<KpTable *>(this)->find_ptr(id); }
    bool emplace(const KeyPt &k) {
        const std::pair<int, bool> r = ids.insert_slot(k.id, FlatNoValue(), k.is3d ? 1 : 0);
        if (!r.second) return false;
        if ((size_t) r.first >= kp.size()) kp.resize((size_t) r.first + 1);
        kp[(size_t) r.first] = k;
        return true;
    }
    void erase(int id) { ids.erase(id); }
    void set_3d(int id) {
        const int s = ids.find_slot(id);
        if (s != FlatHash<FlatNoValue>::END) {
            ids.set_tag(s, 1);
            kp[(size_t) s].is3d = true;
        }
    }
    // for (auto &e: table) e.first / e.second, in mapKeypoints_ order
    template <class T, class K>
    struct Iter {
        T *t;
        int s;
        struct Ref {
            int first;
            K &second;
        };
        Ref operator*() const { return Ref{t->ids.key(s), t->kp[(size_t) s]}; }
        Iter &operator++() {
            s = t->ids.next(s);
            return *this;
        }
        bool operator!=(const Iter &o) const { return s != o.s; }
    };
    Iter<KpTable, KeyPt> begin() { return {this, ids.first()}; }
    Iter<KpTable, KeyPt> end() { return {this, FlatHash<FlatNoValue>::END}; }
    Iter<const KpTable, const KeyPt> begin() const { return {this, ids.first()}; }
    Iter<const KpTable, const KeyPt> end() const { return {this, FlatHash<FlatNoValue>::END}; }
};

// tunables, state.hpp:29-78 with the overrides of System::configure (system.cpp:15-19)
struct Settings {
    int cell_size = 40;
    bool clahe = false;
    float keyframe_filtering_ratio = 0.95f;
    bool p3p_enabled = true;
    bool random_sampling = true;       // multiViewRandomEnabled_
    float min_avg_rot_parallax = 40.f;
    bool klt_use_prior = true;
    int klt_levels = 3;
    float map_max_desc_dist = 0.2f, map_max_proj_px = 2.0f, map_max_reproj_err = 3.0f;
    int ba_min_common_obs = 25;
    bool refine_with_l2 = true;
    float robust_threshold = 5.9915f;
    int max_keypoints = 0;             // frameMaxNumKeypoints_ (state.cpp:8-11)
};

// The id list of one grid cell (gridKeypointsIds_[cell], a std::vector<size_t> in the reference): append, erase preserving order, walk
// in order.  A cell holds one or two ids almost always, so the first four live in the object and a keyframe copy of the grid is a flat
// copy instead of a heap allocation per occupied cell.
struct CellIds {
    int n = 0;
    int inl[4] = {0, 0, 0, 0};
    std::vector<int> more;   // the fifth id onwards
    size_t size() const { return (size_t) n; }
    bool empty() const { return n == 0; }
    int operator[](size_t i) const { return i < 4 ? inl[i] : more[i - 4]; }
    void push_back(int id) {
        if (n < 4) inl[n] = id;
        else more.push_back(id);
        n++;
    }
    void erase_at(size_t i) {
        for (size_t k = i; k + 1 < (size_t) n; k++) {
            const int v = (*this)[k + 1];
            if (k < 4) inl[k] = v;
            else more[k - 4] = v;
        }
        n--;
        if (n >= 4) more.pop_back();
    }
    std::vector<int> to_vector() const {
        std::vector<int> v((size_t) n);
        for (size_t i = 0; i < (size_t) n; i++) v[i] = (*this)[i];
        return v;
    }
};

// class Frame, frame.hpp:40-178
struct FrameRec {
    int id = -1, kfid = 0;
    double timestamp = 0;
    KpTable kps;                               // mapKeypoints_
    std::vector<CellIds> grid;                 // gridKeypointsIds_
    size_t grid_cells = 0, n_occupied = 0, cell = 0, cells_w = 0, cells_h = 0, n_kps = 0, n_2d = 0, n_3d = 0;
    float cell_f = 0.f;   // (float) cell
    SE3 Twc, Tcw;
    std::map<int, int> covisible;              // covisibleKeyframeIds_
    FlatSet local_map;                         // localMapPointIds_ (std::unordered_set<int>)
    const Camera *cam = nullptr;

    void init(const Camera *c, size_t cell_size);
    std::vector<KeyPt> keypoints() const;      // container order
    std::vector<KeyPt> keypoints2d() const;
    std::vector<KeyPt> keypoints3d() const;
    const KeyPt *find(int id) const;
    void add(const KeyPt &k);
    void update(int id, const float *px, const float *unpx, const double *bv);
    void update_slot(int slot, const float *px, const float *unpx, const double *bv);   // the same for a keypoint given by its table slot
    void update_kp(KeyPt &kp, const float *px, const float *unpx, const double *bv);
    void set_desc(int id, const Desc &d);
    bool change_id(int prev_id, int new_id, bool is3d);
    void remove(int id);
    void turn3d(int id);
    // f(id, is3d) for every keypoint in mapKeypoints_ order (reads the 12-byte order slots only)
    template <class F>
    void for_each_id(F &&f) const {
        const FlatHash<FlatNoValue> &h = kps.ids;
        for (int sl = h.first(); sl != FlatHash<FlatNoValue>::END; sl = h.next(sl)) f(h.key(sl), h.tag(sl) != 0);
    }
    // the ids of the 3-D keypoints in mapKeypoints_ order (getKeypoints3d, frame.cpp:55-67) as ONE contiguous array, cached: three loops
    // of every keyframe step walk this list for every covisible keyframe (covisibility's local ids, the local BA's point set, the
    // keyframe filter).  Two levels, because most edits of a keyframe do not change the ORDER of its table:
    //   order_  the table's slots in container order -- a walk of the linked list, ~2 600 dependent loads.  Still right after a keypoint
    //           turned 3-D (triangulation: every observing keyframe, every keyframe step) and after an ERASE (the slot's position becomes
    //           -1: culled observations, bad map points); only an INSERT (the id change of a merge: the newest keyframes) ends it;
    //   ids3d_  the 3-D ids, one sequential pass over order_ (the slots' keys / flags are independent loads); any edit ends it.
    const std::vector<int> &ids3d() const {
        if (!ids3d_valid_) {
            if (!order_valid_) walk_order();
            filter_ids3d();
        }
        return ids3d_;
    }
    void walk_order() const {
        const FlatHash<FlatNoValue> &h = kps.ids;
        order_.clear();
        pos_of_slot_.resize(h.slots());
        for (int sl = h.first(); sl != FlatHash<FlatNoValue>::END; sl = h.next(sl)) {
            pos_of_slot_[(size_t) sl] = (int) order_.size();
            order_.push_back(sl);
        }
        order_valid_ = true;
    }
    void filter_ids3d() const {
        const FlatHash<FlatNoValue> &h = kps.ids;
        ids3d_.resize(order_.size());
        size_t n = 0;
        for (int sl: order_) {
            if (sl < 0) continue;
            ids3d_[n] = h.key(sl);
            n += h.tag(sl) != 0;
        }
        ids3d_.resize(n);
        ids3d_valid_ = true;
    }
    void note_erased_slot(int sl) {   // (before the table's erase)
        ids3d_valid_ = false;
        if (order_valid_ && (size_t) sl < pos_of_slot_.size()) order_[(size_t) pos_of_slot_[(size_t) sl]] = -1;
    }
    void note_inserted() { ids3d_valid_ = order_valid_ = false; }
    // edits of the keypoint table other than removals and the tracker's own position updates (add, change_id, turn3d, update by id, clear):
    // a frame whose count did not move since the previous tracking step can have its slot table carried (Slam::klt_from_motion_prior)
    uint32_t table_edits = 0;
    // The same lists for SEVERAL keyframes, the stale ones rebuilt together: one order walk is a chain of dependent loads (≈ an L2 latency
    // per keypoint), and the loops above walk a dozen keyframes.  Eight chains advance in turn here, so eight loads are in flight instead
    // of one; keyframes whose order still stands only run the sequential pass.
    static void refresh_ids3d(FrameRec *const *kfs, size_t n) {
        constexpr int LANES = 8;
        typedef FlatHash<FlatNoValue> H;
        const FrameRec *f[LANES];
        int cur[LANES], live = 0;
        size_t next = 0;
        auto feed = [&](int l) {
            while (next < n && (!kfs[next] || kfs[next]->order_valid_)) next++;
            if (next >= n) return false;
            f[l] = kfs[next++];
            f[l]->order_.clear();
            f[l]->pos_of_slot_.resize(f[l]->kps.ids.slots());
            f[l]->order_valid_ = true;   // (also keeps a keyframe named twice in kfs out of a second lane)
            f[l]->ids3d_valid_ = false;
            cur[l] = f[l]->kps.ids.first();
            return true;
        };
        for (int l = 0; l < LANES; l++) {
            if (!feed(live)) break;
            live++;
        }
        while (live > 0) {
            for (int l = 0; l < live;) {
                const int sl = cur[l];
                if (sl == H::END) {   // this chain is done: the next stale keyframe takes the lane, or the last lane moves in
                    if (!feed(l)) {
                        live--;
                        f[l] = f[live];
                        cur[l] = cur[live];
                    }
                    continue;
                }
                f[l]->pos_of_slot_[(size_t) sl] = (int) f[l]->order_.size();
                f[l]->order_.push_back(sl);
                cur[l] = f[l]->kps.ids.next(sl);
                l++;
            }
        }
        for (size_t i = 0; i < n; i++)
            if (kfs[i] && !kfs[i]->ids3d_valid_) kfs[i]->filter_ids3d();
    }
    mutable std::vector<int> order_, pos_of_slot_;
    mutable bool order_valid_ = false;
    mutable std::vector<int> ids3d_;
    mutable bool ids3d_valid_ = false;
    bool observes(int id) const { return kps.count(id) != 0; }
    // Frame::getKeypointCellIdx (frame.cpp:313-318): floor(y / cell) * cellsW + floor(x / cell), the divisions in float.  Inline, and the
    // floor by truncation + correction (exact for every float in int range): 2 x per tracked keypoint and frame, and without SSE4.1
    // std::floor(float) is a libm call -- 13 % of the map layer's per-frame time in a sampled profile (tools/host_profile_cpu.py)
    static int floor_to_int(float v) {
        const int i = (int) v;
        return i - (v < (float) i);
    }
    int cell_index(const float *px) const {
        const int r = floor_to_int(px[1] / cell_f), c = floor_to_int(px[0] / cell_f);   // cell_f = (float) cell (an unsigned 64-bit -> float conversion per call otherwise)
        return (int) ((size_t) r * cells_w + (size_t) c);
    }
    void grid_add(const KeyPt &k);
    void grid_remove(const KeyPt &k);
    void set_Twc(const SE3 &T) { Twc = T; Tcw = se3_inverse(T); }
    void add_covisible(int kf);
    void remove_covisible(int kf);
    void decrease_covisible(int kf);
    bool in_image(const float *p) const { return p[0] >= 0 && p[1] >= 0 && p[0] < cam->width && p[1] < cam->height; }
    void project_cam_to_image(const double *p, float *out) const;  // projCamToImage (no distortion)
    void reset();
};

// The operation log of the map points' descriptor tables (Stages::medoid_replay, medoid_table.hpp) + the allocator of their slots.
// The map layer edits the KEY sets (MapPt::kf_desc) at once -- its control flow reads nothing else -- and appends what happened here;
// Slam::flush_medoids hands the log to the stages once per keyframe.
typedef SmallFlatSet<alva_medoid::CAP, alva_medoid::NBKT> DescKeys;
// An arena chunk of n objects on 2 MB pages where the kernel grants them (MADV_HUGEPAGE; a plain allocation otherwise): the host-side
// arenas are walked by map point, one or two objects per 4 KB page -- with small pages every touch is a TLB miss on top of the cache miss
// (three arenas + the object: ~7 000 distinct pages per keyframe against ~2 000 second-level TLB entries).
void *alva_huge_alloc(size_t bytes);   // (map.cpp) zero-filled, 2 MB aligned; free()
template <class T>
struct HugeArray {
    T *p = nullptr;
    HugeArray() {}
    explicit HugeArray(size_t n) : p(static_cast<T *>(alva_huge_alloc(n * sizeof(T)))) {
        if (p)
            for (size_t i = 0; i < n; i++) new (p + i) T;   // (trivial for bytes; the key tables set their one-bucket state)
    }
    HugeArray(HugeArray &&o) : p(o.p) { o.p = nullptr; }
    HugeArray &operator=(HugeArray &&o) {
        if (this != &o) {
            reset();
            p = o.p;
            o.p = nullptr;
        }
        return *this;
    }
    HugeArray(const HugeArray &) = delete;
    HugeArray &operator=(const HugeArray &) = delete;
    ~HugeArray() { reset(); }
    void reset() {
        std::free(p);   // (the arenas' element types have trivial destructors)
        p = nullptr;
    }
    T *get() const { return p; }
    explicit operator bool() const { return p != nullptr; }
};
struct DescBlock {   // the descriptor bytes of one record slot
    DescBytes d[alva_medoid::CAP];
};
struct MedoidLog {
    std::vector<alva_medoid::MedoidOp> ops;
    std::vector<int> touched;                 // slots with operations in `ops`, in first-touch order
    std::vector<int> first_op, last_op;       // per slot: chain head / tail in `ops`, -1 = none
    std::vector<int> free_slots;
    int next_slot = 0;
    // the record arena (mp_rec.hpp): chunks of MP_CHUNK records handed out by the stages (pinned memory behind the HIP stages); record
    // `s` belongs to the map point that holds descriptor-table slot `s`
    std::vector<MpRec *> chunks;
    MpRec *rec(int s) const { return chunks[(size_t) s >> MP_CHUNK_SHIFT] + (s & (MP_CHUNK - 1)); }
    // host-only side arena: the descriptor BYTES of mapKeyframeDescriptors_, alva_medoid::CAP x 32 per record slot, indexed by the key's
    // slot in the key table below (a key keeps its slot while it stays: nothing moves when other keys come and go)
    std::vector<HugeArray<DescBlock>> desc_chunks;
    DescBytes *descs(int s) const { return desc_chunks[(size_t) s >> MP_CHUNK_SHIFT].get()[(size_t) (s & (MP_CHUNK - 1))].d; }
    // ... and the KEYS of mapKeyframeDescriptors_ in libstdc++'s order, one inline table per slot (flat_hash.hpp SmallFlatSet)
    std::vector<HugeArray<DescKeys>> key_chunks;
    DescKeys *keys(int s) const { return key_chunks[(size_t) s >> MP_CHUNK_SHIFT].get() + (size_t) (s & (MP_CHUNK - 1)); }
    // a map point's key set outgrew what its table in the stages holds (medoid_table.hpp: CAP descriptors, NBKT buckets): the table would
    // drop the descriptor and diverge from the key set, so the frame fails instead (Slam::flush_medoids, ALVA_ERR_STATE)
    bool overflow = false;
    int alloc() {
        int s;
        if (!free_slots.empty()) {
            s = free_slots.back();
            free_slots.pop_back();
        } else {
            s = next_slot++;
            first_op.push_back(-1);
            last_op.push_back(-1);
        }
        push(s, alva_medoid::OP_RESET, -1, nullptr, 0);   // whoever had the slot before: a fresh table
        return s;
    }
    void release(int s) { free_slots.push_back(s); }
    void push(int slot, int op, int kf, const uint8_t *desc, int rehash_to) {
        alva_medoid::MedoidOp o{};
        o.op = op; o.kf = kf; o.rehash_to = rehash_to; o.next = -1;
        if (desc) std::memcpy(o.desc, desc, 32);
        const int idx = (int) ops.size();
        ops.push_back(o);
        if (last_op[(size_t) slot] >= 0) ops[(size_t) last_op[(size_t) slot]].next = idx;
        else {
            first_op[(size_t) slot] = idx;
            touched.push_back(slot);
        }
        last_op[(size_t) slot] = idx;
    }
};

// class MapPoint, map_point.hpp:27-86.  The scalars, the observing keyframes (observedKeyframeIds_) and what each keyframe holds about
// the point live in the RECORD `r` (mp_rec.hpp: fixed stride, pinned, read by the kernels); the object keeps the descriptor table's
// host copy and the record's owner-ship.  Not behaviour: the keyframes' own tables stay authoritative for "which keypoint does keyframe
// kf hold", every edit of a KEYFRAME's mapKeypoints_ (the copy, removeKeypointById, the id change of mergeMapPoints, removeKeyframe)
// updates the entry's MPF_INKF half, and ALVA_CHECK_OBS_MIRROR=1 compares every read with the authoritative containers.
struct MapPt {
    MpRec *r = nullptr;
    DescBytes *dsc = nullptr;   // the descriptor bytes (host-only side arena), indexed by the key's slot in kf_desc
    // the KEYS of mapKeyframeDescriptors_ (the reference edits it and mapDescriptorsDist_ together: same keys, same sequence => same
    // iteration order) in libstdc++'s order (flat_hash.hpp); the bytes sit beside the record's entries (merges copy them to the survivor).
    // The distance sums and desc_ itself live in the stages' table `dev_slot` (medoid_table.hpp): every edit below is logged in `mlog`,
    // nothing is read back
    DescKeys &kf_desc;   // (in the host-side arena beside `dsc`: found from the slot number, no allocation per map point or per merge)
    MedoidLog *mlog = nullptr;
    int dev_slot = -1;

    MapPt(MedoidLog *log, int slot, int id_, int kf) : r(log->rec(slot)), dsc(log->descs(slot)), kf_desc(*log->keys(slot)), mlog(log), dev_slot(slot) {
        rec_init(*r, id_, kf, slot);
        kf_desc.reset();   // whoever had the slot before
        obs_insert(kf);
    }
    MapPt(MedoidLog *log, int slot, int id_, int kf, const Desc &d) : MapPt(log, slot, id_, kf) { add_desc(kf, d); }
    MapPt(const MapPt &) = delete;
    MapPt &operator=(const MapPt &) = delete;
    ~MapPt() {
        r->id = -1;
        r->n_ent = r->n_obs = 0;
        mlog->release(dev_slot);
    }
    int id() const { return r->id; }
    // observedKeyframeIds_ (std::set<int>)
    size_t n_obs() const { return r->n_obs; }
    bool obs_has(int kf) const {
        const int i = rec_find(*r, kf);
        return i >= 0 && (r->ent[i].flags & MPF_OBS);
    }
    int obs_first() const {   // *begin()
        for (int i = 0; i < r->n_ent; i++)
            if (r->ent[i].flags & MPF_OBS) return r->ent[i].kf;
        return -1;
    }
    void obs_insert(int kf) {
        ObsEnt *e = rec_slot(*r, kf);
        if (!e) {
            mlog->overflow = true;
            return;
        }
        if (!(e->flags & MPF_OBS)) {
            e->flags |= MPF_OBS;
            r->n_obs++;
        }
    }
    void obs_erase(int kf) {
        const int i = rec_find(*r, kf);
        if (i >= 0 && (r->ent[i].flags & MPF_OBS)) rec_clear_flag(*r, i, MPF_OBS);
    }
    ObsList observers() const { return rec_observers(*r); }   // getObservedKeyframeIds(): a copy
    void remove_obs(int kf);
    void add_desc(int kf, const Desc &d);
    bool is_bad();

    // the keypoint this point has in keyframe kf (null: that keyframe holds none)
    const ObsEnt *in_kf(int kf) const {
        const int i = rec_find(*r, kf);
        return i >= 0 && (r->ent[i].flags & MPF_INKF) ? &r->ent[i] : nullptr;
    }
    void note_px(int kf, const KeyPt &k) {
        ObsEnt *e = rec_slot(*r, kf);
        if (!e) {
            mlog->overflow = true;
            return;
        }
        e->flags |= MPF_INKF;
        e->px[0] = k.px[0]; e->px[1] = k.px[1];
        e->unpx[0] = k.unpx[0]; e->unpx[1] = k.unpx[1];
    }
    void drop_px(int kf) {
        const int i = rec_find(*r, kf);
        if (i >= 0 && (r->ent[i].flags & MPF_INKF)) rec_clear_flag(*r, i, MPF_INKF);
    }
    // mapKeyframeDescriptors_[kf]'s bytes: at the key's slot of kf_desc (add_desc stores them; an erased key's bytes are simply dead)
    const uint8_t *desc_of(int kf) const {
        const int sl = kf_desc.find_slot(kf);
        return sl != DescKeys::END ? dsc[sl] : nullptr;
    }
};

struct InitOverride {  // test hook, see alva_system_debug_set_init_pose
    bool armed = false;
    double pose7[7];
};

class Slam {
public:
    Slam(Stages *stages, const Camera &cam, const Settings &settings);
    ~Slam();

    // System::processCameraPose (system.cpp:156-175): returns 1 / 2 / 3
    int process_frame(const uint8_t *rgba, double timestamp, bool frame_on_device = false);
    void reset();  // System::reset (system.cpp:42-55)
    int last_error() const { return err_; }

    // state (public: the C ABI and the tests read it)
    Stages *st;
    Camera cam;
    Settings cfg;
    double invK[9];
    MedoidLog med_log;                                               // (declared before the map points: they release their slots into it)
    void flush_medoids();                                            // hand the logged descriptor-table edits to the stages (once per keyframe)
    std::vector<int> med_firsts_;
    std::shared_ptr<FrameRec> cur;                                   // currFrame_
    std::unordered_map<int, std::shared_ptr<FrameRec>> keyframes;    // MapManager::mapKeyframes_
    // MapManager::mapMapPoints_ (std::unordered_map<int, shared_ptr<MapPoint>>): the same iteration order on flat arrays (flat_hash.hpp;
    // getCurrentFrameMapPoints walks it), the objects constructed in place in a slot-indexed arena beside the records (mp_obj_chunks_) --
    // no allocation per map point.  An object dies when the reference's last shared_ptr would: at the end of the call that removed it
    // from the map, or at the end of local_ba for the ones removed inside it (mp_graveyard_).
    FlatHash<MapPt *> map_points;
    int next_mp_id = 0, next_kf_id = 0, n_map_points = 0, n_keyframes = 0;
    bool ready_for_init = false, reset_requested = false;           // State::slamReadyForInit_ / slamResetRequested_
    bool p3p_req = false;
    int pose_failed = 0;
    // MotionModel (visual_frontend.hpp:11-68)
    double mm_prev_time = -1.;
    SE3 mm_prev_Twc;
    double mm_log_rel[6] = {0, 0, 0, 0, 0, 0};
    InitOverride init_override;
    SE3 init_computed;  // what checkReadyForInit computed itself on the initialisation frame (before any override)
    // counters for tests / bench
    long n_ba_runs = 0, n_merges = 0, n_kf_culled = 0;
    // The shared-map merge across sessions (north_star's optional extra; no reference counterpart): map point id of THIS session ->
    // (stream, id) of the point that absorbed it in the shared map.  Set by alva_system_set_shared_ids after a merge round, inherited by
    // MapManager::mergeMapPoints' survivor, dropped with the map point.
    std::unordered_map<int, std::pair<int, int>> shared_ids;
    // fb-KLT work done by the tracking steps (bench.py: keypoint-levels per second): LK passes over one pyramid level,
    // forwards + the one-level backward pass, counted from the per-slot result codes (a lost slot counts its first pass only)
    long n_klt_kp_levels = 0, n_klt_slots = 0;
    // wall-clock seconds spent per section since the last reset of the array (tools/system_probe.py): image upload + pyramid enqueue,
    // slot gathering, tracking step until its results are back, tracker bookkeeping, waiting for the pose, pose bookkeeping + keyframe
    // decision, keyframe creation (describe / detect), mapping (triangulation, matching to the local map, local BA)
    double t_section[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    // finer split of the two keyframe sections: prepare, describe tracked, detect, describe + add new | triangulate, covisibility,
    // local-map matching (flatten, stage, merges), BA build, BA solves, BA write-back, keyframe culling
    double t_kf[16] = {0};
    // finer laps for profiling (tools/system_sustained.py FINE=1): see the FINE_* indices in mapper.cpp / map.cpp
    double t_fine[32] = {0};
    const uint8_t *next_frame_hint = nullptr;   // device pointer of the frame after the next processed one (optional, see Stages)

private:
    int err_ = 0;
    // the tracking step's slot tables (reused from frame to frame)
    std::vector<int> job_ids_, pose_ids_, job_slots_;
    std::vector<float> job_px_;
    std::vector<uint8_t> job_is3d_, job_stage3d_;
    std::vector<double> job_wpt_;
    // The CARRIED slot table (Stages::track_carry_buffer): between two keyframes the frame's table only loses rows, and what is left is the
    // previous frame's -- tracked positions, unchanged flags and world points -- so the walk over the keypoints and their map points is
    // replaced by one over the table's order that names, per slot, the slot it was.  Valid while nothing but removals touched the frame
    // (FrameRec::table_edits), no map point moved or went (mp_edits_) and the frame object is the same.
    std::vector<int> job_slots_prev_, job_ids_prev_;   // the previous frame's table slots (in its slot order) | ids
    std::vector<uint8_t> job_is3d_prev_;
    const FrameRec *carry_frame_ = nullptr;
    uint32_t carry_table_edits_ = 0, carry_mp_edits_ = 0, mp_edits_ = 0;
    int carry_n_prev_ = -1;
    bool slots_dirty_ = true;
    bool check_carry_ = false;    // ALVA_CHECK_CARRY=1: every carried table against the assembled one (host side), abort on a difference
    std::vector<float> chk_px_;
    std::vector<uint8_t> chk_is3d_;
    std::vector<uint16_t> chk_carry_;
    std::vector<double> chk_wpt_;
    TrackKlt klt_out_;
    TrackPose pose_out_;
    bool pose_do_p3p_ = true;
    std::vector<uint32_t> parallax_bits_, parallax_tmp_;
    std::vector<uint64_t> par_seen_;   // parallax_of_pairs: the bit patterns seen in this call, [generation : 32 | bits : 32]
    uint32_t par_gen_ = 0;
    uint64_t par_bitmap_[1024];        // ... and the 64 K-bit table of the first pass (hashed bit patterns)
    // compute_parallax's pairing for the keyframe check, put together WHILE the GPU solves the pose (prepare_parallax): slot in the
    // frame's table, the frame keypoint's bearing and the reference keyframe's undistorted pixel of the same id, contiguous
    struct ParPair {
        int slot, id;
        float kf_unpx[2];
        double bv[3];
    };
    std::vector<ParPair> par_pairs_;
    // the same pairs as separate arrays (padded to a multiple of 4): the keyframe check's rotate / project / distance loop runs four pairs
    // per instruction (parallax_of_pairs; the check sits behind the pose with the GPU idle, every frame)
    struct ParSoA {
        std::vector<double> bx, by, bz;
        std::vector<float> ku, kv;
        std::vector<uint32_t> bits;
    } par_soa_;
    int par_frame_ = -1, par_kfid_ = -1;   // the frame / keyframe the pairs were collected for
    void prepare_parallax();
    // NOT the median parallax: a stand-in that compares like it against the two thresholds of the keyframe check (>= min_avg_rot_parallax / 2
    // and >= min_avg_rot_parallax) and nowhere else -- the median's rank is COUNTED, not sorted for (ALVA_CHECK_OBS_MIRROR=1 verifies the two
    // comparisons against the sorted median every frame).  Anything that wants the value itself calls compute_parallax().
    float parallax_of_pairs(const FrameRec &kf);
    float median_of_distinct(std::vector<uint32_t> &all);
    std::vector<int> ids_scratch_, obs_scratch_, index_scratch_, mp_index_, kf_ids_scratch_, local_scratch_, rm_ids_;
    // flat "seen" marks over map point ids (ids are dense, handed out consecutively): inserting a key that is already in a hash set does
    // not change the set, so duplicate inserts are filtered with a byte look-up instead of a hash look-up
    std::vector<uint8_t> mark_a_, mark_b_;
    std::vector<int> touched_a_, touched_b_;   // snapshots of id lists for loops whose bodies may edit the container they walk
    bool fail(int rc) { if (rc && !err_) err_ = rc; return rc != 0; }

    // VisualFrontend
    bool track(const uint8_t *rgba, double timestamp, bool frame_on_device);
    bool process(double timestamp);
    void klt_from_motion_prior();
    bool compute_pose();
    float compute_parallax(int kfid, bool unrotate, bool median);
    bool check_ready_for_init();
    bool check_new_keyframe_required();
    void reset_frame();
    void apply_motion_model(SE3 &Twc, double time);
    void update_motion_model(const SE3 &Twc, double time);

    // MapManager
    void create_keyframe();
    void prepare_frame_removals();   // prepareFrame, the part that edits the frame (thinning, keypoints whose map point is gone)
    void describe_tracked_begin();   // the frame's keypoints in container order; their description enqueued
    void prepare_frame_observers();  // prepareFrame, the observer bookkeeping (under the description)
    std::vector<int> kp_ids_;
    std::vector<float> kp_pts_;
    std::vector<KeyPt *> kp_nodes_;
    void extract_keypoints();
    void add_keyframe();
    void add_map_point(const Desc *d);
    void update_map_point(int id, const double *wpt, double anchor_inv_depth);
public:
    void merge_map_points(int prev_id, int new_id);   // (public: alva_system_merge_map_points applies a shared-map merge through it)
private:
    void remove_keyframe(int kfid);
    void remove_map_point(int id);
    void remove_map_point_obs(int mp_id, int kfid);
    void remove_obs_from_cur(int mp_id);
    bool set_map_point_obs(int mp_id);
    void update_frame_covisibility(FrameRec &frame);
    std::shared_ptr<FrameRec> keyframe(int id) const;
    // The same look-ups through flat mirrors of the two hash maps: ids are handed out consecutively, so id -> object is an array
    // access.  The hash maps stay authoritative (their iteration order is behaviour); every insert / erase / clear updates the mirror.
    // software prefetch for loops that visit map points in an order the hardware cannot predict: the object `far` items ahead, its two
    // small vectors' storage `near` items ahead (their addresses are only known once the object is in cache)
    // Three stages, because the record's address is itself a load from a table indexed by the id (8 bytes x every id ever handed out:
    // the table misses too): 2 x far ahead the TABLE entries, far ahead the record (and the object, the key table, ...), whose addresses
    // the tables -- by then in cache -- give without touching the map point's object.
    void prefetch_tables(int id) const {
        if (id >= 0 && (size_t) id < mp_rec_.size()) {
            __builtin_prefetch(&mp_rec_[(size_t) id]);
            __builtin_prefetch(&mp_slot_[(size_t) id]);
        }
    }
    void prefetch_mp(const int *ids, size_t i, size_t n, size_t far_d = 10) const {
        if (i + 2 * far_d < n) prefetch_tables(ids[i + 2 * far_d]);
        if (i + far_d < n) {
            const MpRec *f = rec_raw(ids[i + far_d]);   // header + the first entries
            if (f) {
                const char *c = (const char *) f;
                __builtin_prefetch(c);
                __builtin_prefetch(c + 64);
                __builtin_prefetch(c + 128);
                __builtin_prefetch(c + 192);
            }
        }
    }
    // the same for loops that run the descriptor-medoid update of every visited map point (addDesc / removeObservedKeyframeId edit the
    // key table and the descriptor bytes beside the record: both found from the slot number)
    void prefetch_mp_desc(const int *ids, size_t i, size_t n, size_t near_d = 4, size_t far_d = 10) const {
        if (i + 2 * far_d < n) {
            const int id = ids[i + 2 * far_d];
            prefetch_tables(id);
            if (id >= 0 && (size_t) id < mp_flat_.size()) __builtin_prefetch(&mp_flat_[(size_t) id]);
        }
        if (i + far_d < n) {
            const int id = ids[i + far_d];
            const MpRec *rc = rec_raw(id);
            if (rc) {
                __builtin_prefetch(mp_flat_[(size_t) id]);   // the object (40 bytes of pointers)
                const char *c = (const char *) rc;
                __builtin_prefetch(c);
                __builtin_prefetch(c + 64);
                __builtin_prefetch(c + 128);
                const int sl = mp_slot_[(size_t) id];
                if (sl >= 0) {
                    const char *t = (const char *) med_log.keys(sl);
                    for (size_t o = 0; o < sizeof(DescKeys); o += 64) __builtin_prefetch(t + o, 1);
                }
            }
        }
        if (i + near_d < n) {   // where a new keyframe's descriptor bytes will go: the slot its key will take (the key table is in cache by now)
            const int id = ids[i + near_d];
            const int sl = id >= 0 && (size_t) id < mp_slot_.size() ? mp_slot_[(size_t) id] : -1;
            if (sl >= 0) __builtin_prefetch((const char *) med_log.descs(sl) + (size_t) med_log.keys(sl)->next_slot() * 32, 1);
        }
    }
    std::vector<int> fresh_ids_;        // scratch: a keyframe's ids that are new to the set being built
    std::vector<FrameRec *> kf_ptrs_;   // scratch: the keyframes a loop is about to walk (FrameRec::refresh_ids3d)
    FrameRec *kf_raw(int id) const { return id >= 0 && (size_t) id < kf_flat_.size() ? kf_flat_[(size_t) id] : nullptr; }
    MapPt *mp_raw(int id) const { return id >= 0 && (size_t) id < mp_flat_.size() ? mp_flat_[(size_t) id] : nullptr; }
    MpRec *rec_raw(int id) const { return id >= 0 && (size_t) id < mp_rec_.size() ? mp_rec_[(size_t) id] : nullptr; }   // the record alone (hot loops)
    std::vector<FrameRec *> kf_flat_;
    std::vector<MapPt *> mp_flat_;
    std::vector<MpRec *> mp_rec_;
    std::vector<int> mp_slot_;   // id -> record / descriptor-table slot (-1: no such point): a stage job names slots without touching the records
    // observer count per map point id (0 = no such point), saturated at 255: the keyframe filter of Mapper::optimize asks "more than
    // four observers?" of every 3-D keypoint of every covisible keyframe; a byte table answers without touching the map point
    std::vector<uint8_t> mp_nobs_;
    void sync_nobs(const MapPt &mp) { mp_nobs_[(size_t) mp.r->id] = mp.r->n_obs; }
    bool check_obs_mirror_ = false;
    std::vector<uint8_t> ba_arena_;                           // backing store of local_ba's function-local containers
    struct BaScratch {   // local_ba's problem arrays (see there)
        std::vector<int> wb_ids;      // the write-back's walk of map_local_plms: ids and records in the container's order
        std::vector<MpRec *> wb_recs;
        std::vector<int> pt_ids, pt_anchor_slot, obs_kf, pt_ptr, lone_ids, slot_ids, slot_kfid, ptr2, as2, okf2, ids2, from2;
        std::vector<double> pt_anchor_uv, pt_inv, obs_uv, lone_inv, auv2, inv2, ouv2;
        std::vector<uint64_t> bad_bits;
        std::vector<uint8_t> kf_used, kc;
        FlatHash<MpRec *> local_mps;   // (the RECORDS: the write-back's loops do not touch the map points' objects)
        FlatSet mps_to_opt;
    } ba_scratch_;
    bool defer_mp_free_ = false;                              // remove_map_point parks the object until local_ba returns
    std::vector<MapPt *> mp_graveyard_;
    struct MapPtBox {
        alignas(8) unsigned char b[64];
    };
    static_assert(sizeof(MapPt) <= sizeof(MapPtBox), "MapPt outgrew its arena box");
    std::vector<HugeArray<MapPtBox>> mp_obj_chunks_;   // one box per record slot
    MapPt *mp_box(int slot) const { return reinterpret_cast<MapPt *>(mp_obj_chunks_[(size_t) slot >> MP_CHUNK_SHIFT].get()[(size_t) (slot & (MP_CHUNK - 1))].b); }
    void destroy_map_point(MapPt *mp) {   // (releases the slot; the box stays where it is)
        mp->~MapPt();
#if defined(__SANITIZE_ADDRESS__)
        __asan_poison_memory_region(mp, sizeof(MapPtBox));   // a sanitizer build of the GPU-less harness flags any later use of the object
#endif
    }
    MapPt *mp_box_fresh(int slot) const {   // where a new map point of this slot is constructed
        MapPt *p = mp_box(slot);
#if defined(__SANITIZE_ADDRESS__)
        __asan_unpoison_memory_region(p, sizeof(MapPtBox));
#endif
        return p;
    }
    const ObsEnt *obs_of(const MapPt &mp, int kfid) const;   // the keypoint of `mp` in keyframe `kfid` (null: that keyframe holds none)
    bool ensure_rec_chunk(int slot);                          // the arena chunk of record `slot` exists (asks the stages for it)
    // The NEXT arena chunk, prepared on a helper thread while the session goes on: a chunk is 4 MB of page-locked memory from the stages
    // (~0.5 ms to allocate, lock and clear) + 5 MB of descriptor bytes whose pages would otherwise be faulted in one by one by the
    // keyframes that hand the slots out (a map grows by ~430 points per keyframe and never shrinks: ~125 us of every keyframe)
    struct ChunkAhead {
        std::thread th;
        int index = -1;
        MpRec *rec = nullptr;
        HugeArray<DescBlock> dsc;
        HugeArray<DescKeys> keys;
        HugeArray<MapPtBox> objs;
    } chunk_ahead_;
    void start_chunk_ahead(int index);

    // Mapper
    void process_new_keyframe(int kfid);
    void triangulate_temporal(FrameRec &frame);
    bool matching_to_local_map(FrameRec &frame);
    std::map<int, int> match_to_map(FrameRec &frame, float max_proj_err, float dist_ratio, FlatSet &local);
    void optimize(const std::shared_ptr<FrameRec> &kf);
    // Optimizer
    void local_ba(FrameRec &new_frame);
};

}  // namespace alva_slam
