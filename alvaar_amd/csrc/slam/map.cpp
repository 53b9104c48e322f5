// Frame / map-point / map bookkeeping of the host-side map layer (see slam.hpp).  Mirrors the observable behaviour of
// src/slam/src/frame.cpp, map_point.cpp and map_manager.cpp of the reference -- including the order in which the hash
// containers are mutated, which is what fixes their iteration order.
#include "slam.hpp"
#if defined(__linux__)
#include <sys/mman.h>
#endif
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <stdexcept>

namespace alva_slam {

namespace {
struct Lap {
    std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
    void operator()(double &acc) {
        const auto t1 = std::chrono::steady_clock::now();
        acc += std::chrono::duration<double>(t1 - t0).count();
        t0 = t1;
    }
};
}  // namespace

static inline int popcount256(const Desc &a, const Desc &b) {  // cv::norm(a, b, NORM_HAMMING) on 32 bytes (map_point.cpp:106,158,212)
    int s = 0;
    for (int i = 0; i < 32; i += 8) {
        uint64_t x, y;
        __builtin_memcpy(&x, a.b + i, 8);
        __builtin_memcpy(&y, b.b + i, 8);
        s += __builtin_popcountll(x ^ y);
    }
    return s;
}

// ---------------------------------------------------------------------------------------------------- FrameRec (frame.cpp)
void FrameRec::init(const Camera *c, size_t cell_size) {  // frame.cpp:9-21
    cam = c;
    cell = cell_size;
    cell_f = (float) cell;
    cells_w = (size_t) std::ceil((float) c->width / (float) cell);
    cells_h = (size_t) std::ceil((float) c->height / (float) cell);
    grid_cells = cells_w * cells_h;
    n_occupied = 0;
    grid.assign(grid_cells, {});
}

std::vector<KeyPt> FrameRec::keypoints() const {  // frame.cpp:30-39
    std::vector<KeyPt> v;
    v.reserve(n_kps);
    for (const auto &e: kps) v.push_back(e.second);
    return v;
}
std::vector<KeyPt> FrameRec::keypoints2d() const {  // :41-53
    std::vector<KeyPt> v;
    v.reserve(n_2d);
    for (const auto &e: kps)
        if (!e.second.is3d) v.push_back(e.second);
    return v;
}
std::vector<KeyPt> FrameRec::keypoints3d() const {  // :55-67
    std::vector<KeyPt> v;
    v.reserve(n_3d);
    for (const auto &e: kps)
        if (e.second.is3d) v.push_back(e.second);
    return v;
}
const KeyPt *FrameRec::find(int id_) const { return kps.find_ptr(id_); }

void FrameRec::add(const KeyPt &k) {  // frame.cpp:124-143
    if (!kps.emplace(k)) return;
    table_edits++;
    note_inserted();
    grid_add(k);
    n_kps++;
    if (k.is3d) n_3d++;
    else n_2d++;
}

void FrameRec::update(int id_, const float *px, const float *unpx, const double *bv) {  // frame.cpp:160-174
    KeyPt *cur_kp = kps.find_ptr(id_);
    if (!cur_kp) return;
    table_edits++;
    update_kp(*cur_kp, px, unpx, bv);
}

void FrameRec::update_slot(int slot, const float *px, const float *unpx, const double *bv) { update_kp(kps.kp[(size_t) slot], px, unpx, bv); }

void FrameRec::update_kp(KeyPt &kp, const float *px, const float *unpx, const double *bv) {
    KeyPt *cur_kp = &kp;
    const int a = cell_index(cur_kp->px), b = cell_index(px);  // updateKeypointInGrid (:296-311)
    if (a != b) grid_remove(*cur_kp);
    cur_kp->px[0] = px[0]; cur_kp->px[1] = px[1];
    cur_kp->unpx[0] = unpx[0]; cur_kp->unpx[1] = unpx[1];
    cur_kp->bv[0] = bv[0]; cur_kp->bv[1] = bv[1]; cur_kp->bv[2] = bv[2];
    if (a != b) grid_add(*cur_kp);
}

void FrameRec::set_desc(int id_, const Desc &d) {  // :176-185
    KeyPt *k = kps.find_ptr(id_);
    if (!k) return;
    k->desc = d;
    k->has_desc = true;
}

bool FrameRec::change_id(int prev_id, int new_id, bool is3d) {  // :187-207
    if (kps.count(new_id)) return false;
    const KeyPt *old = kps.find_ptr(prev_id);
    if (!old) return false;
    KeyPt k = *old;
    k.id = new_id;
    k.is3d = is3d;
    remove(prev_id);
    add(k);
    return true;
}

void FrameRec::remove(int id_) {  // :209-232
    const KeyPt *k = kps.find_ptr(id_);
    if (!k) return;
    grid_remove(*k);
    if (k->is3d) n_3d--;
    else n_2d--;
    n_kps--;
    note_erased_slot((int) (k - kps.kp.data()));   // (a keypoint lives at its slot's index)
    kps.erase(id_);
}

void FrameRec::turn3d(int id_) {  // :234-248
    const KeyPt *k = kps.find_ptr(id_);
    if (!k) return;
    if (!k->is3d) {
        kps.set_3d(id_);
        table_edits++;
        ids3d_valid_ = false;
        n_3d++;
        n_2d--;
    }
}

void FrameRec::grid_add(const KeyPt &k) {  // :255-265
    const int idx = cell_index(k.px);
    CellIds &c = grid.at((size_t) idx);
    if (c.empty()) n_occupied++;
    c.push_back(k.id);
}

void FrameRec::grid_remove(const KeyPt &k) {  // :267-294
    const int idx = cell_index(k.px);
    if (idx < 0 || idx >= (int) grid.size()) return;
    CellIds &c = grid[(size_t) idx];
    for (size_t i = 0; i < c.size(); i++)
        if (c[i] == k.id) {
            c.erase_at(i);
            if (c.empty()) n_occupied--;
            break;
        }
}

void FrameRec::add_covisible(int kf) {  // :348-364
    if (kf == kfid) return;
    auto it = covisible.find(kf);
    if (it != covisible.end()) it->second += 1;
    else covisible.emplace(kf, 1);
}
void FrameRec::remove_covisible(int kf) {  // :366-374
    if (kf == kfid) return;
    covisible.erase(kf);
}
void FrameRec::decrease_covisible(int kf) {  // :376-396
    if (kf == kfid) return;
    auto it = covisible.find(kf);
    if (it != covisible.end() && it->second != 0) {
        it->second -= 1;
        if (it->second == 0) covisible.erase(it);
    }
}

void FrameRec::project_cam_to_image(const double *p, float *out) const {  // camera_calibration.cpp:25-32
    const double iz = 1. / p[2], x = p[0] * iz, y = p[1] * iz;
    out[0] = (float) (cam->fx * x + cam->cx);
    out[1] = (float) (cam->fy * y + cam->cy);
}

void FrameRec::reset() {  // :467-489
    id = -1;
    kfid = 0;
    timestamp = 0.;
    kps.clear();
    table_edits++;
    note_inserted();
    grid.clear();
    grid.resize(grid_cells);
    n_kps = n_2d = n_3d = 0;
    n_occupied = 0;
    Twc = SE3();
    Tcw = SE3();
    covisible.clear();
    local_map.clear();
}

// ---------------------------------------------------------------------------------------------------- MapPt (map_point.cpp)
void MapPt::remove_obs(int kf) {  // map_point.cpp:73-129
    if (!obs_has(kf)) return;
    obs_erase(kf);
    if (r->n_obs == 0) {
        r->has_desc = 0;
        kf_desc.clear();
        mlog->push(dev_slot, alva_medoid::OP_CLEAR, -1, nullptr, 0);
        return;
    }
    if (kf == r->anchor_kf) r->anchor_kf = obs_first();
    // :93-128: the distances of the remaining descriptors, the new desc_ -- in the stages' table (medoid_table.hpp remove_desc)
    if (kf_desc.erase(kf)) {
        mlog->push(dev_slot, alva_medoid::OP_REMOVE, kf, nullptr, 0);
    }
}

void MapPt::add_desc(int kf, const Desc &d) {  // map_point.cpp:131-181 (the descriptor medoid: medoid_table.hpp add_desc)
    const size_t buckets = kf_desc.bucket_count();
    int sl = DescKeys::END;
    const int ins = kf_desc.insert(kf, &sl);
    if (ins == 0) return;
    if (ins < 0) {   // more descriptors than a table holds (CAP keys / NBKT buckets, here and in the stages): the frame fails (flush_medoids)
        mlog->overflow = true;
        return;
    }
    std::memcpy(dsc[sl], d.b, 32);   // mapKeyframeDescriptors_[kf] (the bytes a merge copies to the survivor)
    r->has_desc = 1;   // desc_ is never empty again until the last observation goes
    // the bucket count of the rehash this insert caused, if any: the table in the stages replays the list surgery, not the growth policy
    mlog->push(dev_slot, alva_medoid::OP_ADD, kf, d.b, kf_desc.bucket_count() != buckets ? (int) kf_desc.bucket_count() : 0);
}

bool MapPt::is_bad() { return rec_is_bad(*r); }  // map_point.cpp:183-202

// ---------------------------------------------------------------------------------------------------- map (map_manager.cpp)
std::shared_ptr<FrameRec> Slam::keyframe(int id) const {
    auto it = keyframes.find(id);
    return it == keyframes.end() ? nullptr : it->second;
}

void Slam::flush_medoids() {
    MedoidLog &L = med_log;
    if (L.overflow) {
        L.overflow = false;
        std::fprintf(stderr, "alva_slam: a map point holds more descriptors than its table (medoid_table.hpp CAP / NBKT)\n");
        fail(-4);
    }
    if (L.ops.empty()) return;
    std::vector<int> &firsts = med_firsts_;
    firsts.resize(L.touched.size());
    for (size_t i = 0; i < L.touched.size(); i++) {
        const size_t s = (size_t) L.touched[i];
        firsts[i] = L.first_op[s];
        L.first_op[s] = L.last_op[s] = -1;
    }
    fail(st->medoid_replay((int) L.ops.size(), L.ops.data(), (int) L.touched.size(), L.touched.data(), firsts.data(), L.next_slot));
    L.ops.clear();
    L.touched.clear();
}

void Slam::create_keyframe() {  // map_manager.cpp:12-22
    Lap lap;
    // prepareFrame (:24-81) is two things: keypoints leave the frame (thinning, map points that are gone), then the new keyframe is
    // entered into the observer set of every remaining keypoint's map point -- 2 400 records touched, nothing the description of those
    // keypoints (extractKeypoints' first step, :224-241) depends on.  So the description is STARTED in between and runs on the device
    // under the second half.
    prepare_frame_removals();
    describe_tracked_begin();
    prepare_frame_observers();
    lap(t_kf[0]);
    extract_keypoints();
    add_keyframe();
    // this keyframe's descriptor edits so far (one per tracked keypoint + the new points': the bulk of the log) go to the stages NOW: the
    // replay runs on the device under triangulation / covisibility, and the local-map matching -- which reads the tables -- finds only the
    // few edits made since in front of it (round 5, first form: the whole log was replayed in front of the matcher, +45 us of its wait)
    flush_medoids();
    lap(t_kf[4]);
}

void Slam::prepare_frame_removals() {  // map_manager.cpp:24-81, the part that edits the frame
    cur->kfid = next_kf_id;
    if ((int) cur->n_kps > cfg.max_keypoints) {
        for (size_t ci = 0; ci < cur->grid.size(); ci++) {
            // the reference iterates the cell's id vector while removals edit it (range-for over a reference, :32-68); a copy of the
            // ids taken up front visits the same elements because at most one removal happens per cell and it ends the scan
            if (cur->grid[ci].size() <= 2) continue;
            const std::vector<int> ids = cur->grid[ci].to_vector();
            if (ids.size() > 2) {
                int to_remove = -1;
                size_t min_obs = std::numeric_limits<size_t>::max();
                bool broke = false;
                for (int lmid: ids) {
                    const MpRec *it = rec_raw(lmid);
                    if (it) {
                        const size_t nobs = it->n_obs;
                        if (nobs < min_obs) {
                            to_remove = lmid;
                            min_obs = nobs;
                        }
                    } else {
                        remove_obs_from_cur(lmid);
                        broke = true;
                        break;
                    }
                }
                (void) broke;
                if (to_remove >= 0) remove_obs_from_cur(to_remove);
            }
        }
    }
    // the reference walks a COPY of the keypoints (getKeypoints, :70) because the body may drop keypoints: a snapshot of the ids does.
    // Keypoints whose map point is gone leave the frame here (:72-76); the others get their observer in prepare_frame_observers -- the two
    // act on different objects, so taking the removals first changes nothing
    ids_scratch_.clear();
    for (const auto &e: cur->kps)
        if (!rec_raw(e.first)) ids_scratch_.push_back(e.first);
    for (int id: ids_scratch_) remove_obs_from_cur(id);
}

void Slam::prepare_frame_observers() {  // map_manager.cpp:70-80: addObservedKeyframeId for every keypoint's map point
    const std::vector<int> &ids = kp_ids_;   // the frame's keypoints in container order (describe_tracked_begin's snapshot)
    for (size_t i = 0; i < ids.size(); i++) {
        const int id = ids[i];
        prefetch_mp(ids.data(), i, ids.size());
        MpRec *r = rec_raw(id);   // (the record and its side arena alone: the map point's object is not touched)
        if (!r) continue;
        ObsEnt *e = rec_slot(*r, next_kf_id);
        if (!e) {
            med_log.overflow = true;
            continue;
        }
        if (!(e->flags & MPF_OBS)) {
            e->flags |= MPF_OBS;
            r->n_obs++;
        }
        mp_nobs_[(size_t) id] = r->n_obs;
    }
}

// describeKeypoints (:224-241), first half: the frame's keypoints in container order (getKeypoints()) and the description of their
// positions in the raw image, enqueued (Stages::describe_begin); extract_keypoints collects it
void Slam::describe_tracked_begin() {
    const int n = (int) cur->kps.size();
    kp_ids_.clear();
    kp_pts_.resize((size_t) n * 2);
    kp_nodes_.resize((size_t) n);   // the keypoints themselves (nothing is inserted or erased before they are used)
    size_t i = 0;
    for (auto e: cur->kps) {
        kp_ids_.push_back(e.first);
        kp_nodes_[i] = &e.second;
        kp_pts_[2 * i] = e.second.px[0];
        kp_pts_[2 * i + 1] = e.second.px[1];
        i++;
    }
    if (n) fail(st->describe_begin(n, kp_pts_.data()));
}

void Slam::extract_keypoints() {  // map_manager.cpp:193-241
    if (err_) return;
    const int n = (int) kp_ids_.size();
    const std::vector<int> &kp_ids = kp_ids_;
    const std::vector<float> &pts = kp_pts_;
    const std::vector<KeyPt *> &nodes = kp_nodes_;
    // describeKeypoints (:224-241): refresh the descriptors of the tracked keypoints in the raw image.  The detector of :213 needs only
    // the image and the tracked positions, so it is STARTED before the descriptor medoids are updated on the host (nothing it reads
    // or writes is touched by them) and collected afterwards: the two overlap.
    const int to_detect = cfg.max_keypoints - (int) cur->n_occupied;
    const int cap = (int) cur->grid_cells + 8;
    std::vector<uint8_t> desc((size_t) n * 32), valid((size_t) n);
    if (n) {
        Lap lap;
        if (fail(st->describe_end(desc.data(), valid.data()))) return;
        lap(t_kf[1]);
    }
    Lap lap_det;
    if (to_detect > 0 && fail(st->detect_begin(cfg.cell_size, n, pts.data(), cap))) return;
    lap_det(t_kf[2]);
    if (n) {
        Lap lap;
        for (int i = 0; i < n; i++) {
            prefetch_mp_desc(kp_ids.data(), (size_t) i, (size_t) n);
            if (valid[(size_t) i]) {
                Desc d;
                __builtin_memcpy(d.b, &desc[(size_t) i * 32], 32);
                nodes[(size_t) i]->desc = d;        // Frame::updateKeypointDesc (frame.cpp:176-185)
                nodes[(size_t) i]->has_desc = true;
                MapPt *mp = mp_raw(kp_ids[(size_t) i]);
                if (!mp) throw std::out_of_range("map point");   // mapMapPoints_.at() in the reference (:233)
                mp->add_desc(cur->kfid, d);
            }
        }
        lap(t_kf[15]);
    }
    if (to_detect > 0) {
        std::vector<float> np((size_t) cap * 2);
        int count = 0;
        Lap lap;
        if (fail(st->detect_end(np.data(), &count))) return;
        lap(t_kf[2]);
        if (count > 0) {
            std::vector<uint8_t> desc((size_t) count * 32), valid((size_t) count);
            std::vector<float> unpx((size_t) count * 2);
            std::vector<double> bv((size_t) count * 3);
            if (fail(st->describe_and_compute(count, np.data(), desc.data(), valid.data(), unpx.data(), bv.data()))) return;
            lap(t_kf[3]);
            Lap fine;
            t_fine[26] += (double) count;
            for (int i = 0; i < count; i++) {  // addKeypointsToFrame (:166-191)
                KeyPt k;
                k.id = next_mp_id;
                k.px[0] = np[2 * (size_t) i]; k.px[1] = np[2 * (size_t) i + 1];
                k.unpx[0] = unpx[2 * (size_t) i]; k.unpx[1] = unpx[2 * (size_t) i + 1];
                for (int c = 0; c < 3; c++) k.bv[c] = bv[3 * (size_t) i + c];
                if (valid[(size_t) i]) {
                    __builtin_memcpy(k.desc.b, &desc[(size_t) i * 32], 32);
                    k.has_desc = true;
                    cur->add(k);
                    add_map_point(&k.desc);
                } else {
                    cur->add(k);
                    add_map_point(nullptr);
                }
            }
            fine(t_fine[13]);   // new keypoints + map points
        }
    }
}

void *alva_huge_alloc(size_t bytes) {
    const size_t huge = (size_t) 2 << 20, sz = (bytes + huge - 1) / huge * huge;
    void *p = std::aligned_alloc(huge, sz);
    if (!p) return nullptr;
#if defined(__linux__)
    static const bool off = std::getenv("ALVA_NO_HUGE_ARENA") != nullptr;
    if (!off) madvise(p, sz, MADV_HUGEPAGE);   // advice only: where transparent huge pages are off this is a plain allocation
#endif
    std::memset(p, 0, sz);   // the first touch (on the map layer's helper thread for every chunk but the first)
    return p;
}

Slam::~Slam() {
    if (chunk_ahead_.th.joinable()) chunk_ahead_.th.join();
}

void Slam::start_chunk_ahead(int index) {
    static const bool off = std::getenv("ALVA_NO_CHUNK_AHEAD") != nullptr;
    if (off || chunk_ahead_.th.joinable()) return;
    chunk_ahead_.index = index;
    chunk_ahead_.rec = nullptr;
    chunk_ahead_.dsc.reset();
    chunk_ahead_.keys.reset();
    chunk_ahead_.objs.reset();
    chunk_ahead_.th = std::thread([this, index] {
        chunk_ahead_.rec = st->mp_arena_chunk(index);
        chunk_ahead_.dsc = HugeArray<DescBlock>(MP_CHUNK);   // (zero-filled / constructed here: the first touch of the pages)
        chunk_ahead_.keys = HugeArray<DescKeys>(MP_CHUNK);
        chunk_ahead_.objs = HugeArray<MapPtBox>(MP_CHUNK);
    });
}

bool Slam::ensure_rec_chunk(int slot) {
    const size_t c = (size_t) slot >> MP_CHUNK_SHIFT;
    while (med_log.chunks.size() <= c) {
        const auto t0 = std::chrono::steady_clock::now();
        const int index = (int) med_log.chunks.size();
        MpRec *chunk = nullptr;
        HugeArray<DescBlock> dsc;
        HugeArray<DescKeys> keys;
        HugeArray<MapPtBox> objs;
        if (chunk_ahead_.th.joinable()) {
            chunk_ahead_.th.join();
            if (chunk_ahead_.index == index) {
                chunk = chunk_ahead_.rec;
                dsc = std::move(chunk_ahead_.dsc);
                keys = std::move(chunk_ahead_.keys);
                objs = std::move(chunk_ahead_.objs);
            }
        }
        if (!chunk) chunk = st->mp_arena_chunk(index);
        if (!chunk) return false;
        if (!dsc) dsc = HugeArray<DescBlock>(MP_CHUNK);
        if (!keys) keys = HugeArray<DescKeys>(MP_CHUNK);
        if (!objs) objs = HugeArray<MapPtBox>(MP_CHUNK);
        if (!dsc || !keys || !objs) return false;
        med_log.chunks.push_back(chunk);
        med_log.desc_chunks.push_back(std::move(dsc));
        med_log.key_chunks.push_back(std::move(keys));
        mp_obj_chunks_.push_back(std::move(objs));
        static const bool timing = std::getenv("ALVA_ARENA_TIMING") != nullptr;
        if (timing)
            std::fprintf(stderr, "[arena] chunk %d: %.0f us\n", index, 1e6 * std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count());
        start_chunk_ahead(index + 1);   // the one after it, while this one fills
    }
    return true;
}

const ObsEnt *Slam::obs_of(const MapPt &mp, int kfid) const {
    const ObsEnt *o = mp.in_kf(kfid);
    if (check_obs_mirror_) {
        const FrameRec *kf = kf_raw(kfid);
        const KeyPt *kp = kf ? kf->find(mp.id()) : nullptr;
        if ((kp != nullptr) != (o != nullptr) || (kp && (std::memcmp(kp->px, o->px, 8) || std::memcmp(kp->unpx, o->unpx, 8)))) {
            std::fprintf(stderr, "alva_slam: observation mirror out of sync (map point %d, keyframe %d)\n", mp.id(), kfid);
            std::abort();
        }
    }
    return o;
}

void Slam::add_keyframe() {  // map_manager.cpp:243-252: an independent copy of the current frame
    Lap fine;
    std::shared_ptr<FrameRec> kf = std::make_shared<FrameRec>(*cur);
    fine(t_fine[10]);   // frame copy
    for (const auto &e: kf->kps) {
        MapPt *m = mp_raw(e.first);
        if (m) m->note_px(next_kf_id, e.second);
    }
    fine(t_fine[12]);   // observation mirror
    keyframes.emplace(next_kf_id, kf);
    if (kf_flat_.size() <= (size_t) next_kf_id) kf_flat_.resize((size_t) next_kf_id + 32, nullptr);
    kf_flat_[(size_t) next_kf_id] = kf.get();
    n_keyframes++;
    next_kf_id++;
}

void Slam::add_map_point(const Desc *d) {  // map_manager.cpp:254-327
    const int slot = med_log.alloc();   // descriptor-table slot = record slot
    if (!ensure_rec_chunk(slot)) {
        med_log.release(slot);
        fail(-3);
        return;
    }
    MapPt *mp = d ? new (mp_box_fresh(slot)) MapPt(&med_log, slot, next_mp_id, next_kf_id, *d) : new (mp_box_fresh(slot)) MapPt(&med_log, slot, next_mp_id, next_kf_id);
    map_points.insert_slot(next_mp_id, mp);
    if (mp_flat_.size() <= (size_t) next_mp_id) {
        mp_flat_.resize((size_t) next_mp_id + 4096, nullptr);
        mp_rec_.resize(mp_flat_.size(), nullptr);
        mp_slot_.resize(mp_flat_.size(), -1);
        mp_nobs_.resize(mp_flat_.size(), 0);
    }
    mp_flat_[(size_t) next_mp_id] = mp;
    mp_rec_[(size_t) next_mp_id] = mp->r;
    mp_slot_[(size_t) next_mp_id] = slot;
    sync_nobs(*mp);
    next_mp_id++;
    n_map_points++;
}

void Slam::update_map_point(int id, const double *wpt, double anchor_inv_depth) {  // map_manager.cpp:366-426
    MpRec *rec = rec_raw(id);   // the flat mirror of mapMapPoints_ (same membership); the record alone in the common case
    if (!rec) return;
    mp_edits_++;   // (a world point moves: no carried slot table across this, frontend.cpp)
    if (rec->is3d) {
        rec->X[0] = wpt[0]; rec->X[1] = wpt[1]; rec->X[2] = wpt[2];
        if (anchor_inv_depth >= 0.) rec->inv_depth = anchor_inv_depth;
        return;
    }
    MapPt &mp = *mp_raw(id);
    if (!mp.r->is3d) {
        const ObsList obs = mp.observers();  // getObservedKeyframeIds returns a copy; removals below edit the member
        for (int kf: obs) {
            FrameRec *k = kf_raw(kf);
            if (k) k->turn3d(id);
            else mp.remove_obs(kf);
        }
        sync_nobs(mp);
        if (mp.r->observed) cur->turn3d(id);
    }
    mp.r->X[0] = wpt[0]; mp.r->X[1] = wpt[1]; mp.r->X[2] = wpt[2];
    mp.r->is3d = 1;
    if (anchor_inv_depth >= 0.) mp.r->inv_depth = anchor_inv_depth;
}

void Slam::merge_map_points(int prev_id, int new_id) {  // map_manager.cpp:428-513
    MapPt *prev = mp_raw(prev_id), *nw = mp_raw(new_id);   // (the id -> object table: same membership as mapMapPoints_)
    if (!prev || !nw || !nw->r->is3d) return;
    const ObsList next_kfs = nw->observers(), prev_kfs = prev->observers();
    const DescKeys prev_desc = prev->kf_desc;   // a copy of the keys, in the original's order (300 bytes on the stack)
    for (int pk: prev_kfs) {
        auto kf = keyframes.find(pk);
        if (kf == keyframes.end()) continue;
        if (kf->second->change_id(prev_id, new_id, nw->r->is3d != 0)) {
            prev->drop_px(pk);
            nw->note_px(pk, *kf->second->find(new_id));
            nw->obs_insert(pk);
            sync_nobs(*nw);
            for (int nk: next_kfs) {
                auto co = keyframes.find(nk);
                if (co != keyframes.end()) {
                    kf->second->add_covisible(nk);
                    co->second->add_covisible(pk);
                }
            }
        }
    }
    for (int se = prev_desc.first(); se != DescKeys::END; se = prev_desc.next(se)) {
        const uint8_t *b = prev->dsc[se];   // (the copy's slots are the original's: the bytes of key `se`)
        Desc d;
        std::memcpy(d.b, b, 32);
        nw->add_desc(prev_desc.key(se), d);
    }
    if (cur->observes(prev_id)) {
        if (cur->change_id(prev_id, new_id, nw->r->is3d != 0)) set_map_point_obs(new_id);
    }
    if (prev->r->is3d) n_map_points--;
    {   // the survivor keeps the absorbed point's place in the shared map, unless it has one of its own
        auto sh = shared_ids.find(prev_id);
        if (sh != shared_ids.end()) {
            shared_ids.emplace(new_id, sh->second);
            shared_ids.erase(sh);
        }
    }
    mp_flat_[(size_t) prev_id] = nullptr;
    mp_rec_[(size_t) prev_id] = nullptr;
    mp_slot_[(size_t) prev_id] = -1;
    mp_nobs_[(size_t) prev_id] = 0;
    map_points.erase(prev_id);
    destroy_map_point(prev);   // (the absorbed point lived to the end of the call, like the reference's local shared_ptr)
    n_merges++;
}

void Slam::remove_keyframe(int kfid) {  // map_manager.cpp:515-557
    auto it = keyframes.find(kfid);
    if (it == keyframes.end()) return;
    std::vector<int> &ids = rm_ids_;   // the keyframe's ids in container order (the body edits map points only)
    ids.clear();
    it->second->for_each_id([&](int kid, bool) { ids.push_back(kid); });
    for (size_t i = 0; i < ids.size(); i++) {
        prefetch_mp_desc(ids.data(), i, ids.size());
        MapPt *m = mp_raw(ids[i]);
        if (m) {
            m->remove_obs(kfid);
            m->drop_px(kfid);
            sync_nobs(*m);
        }
    }
    for (const auto &c: it->second->covisible) {
        auto co = keyframes.find(c.first);
        if (co != keyframes.end()) co->second->remove_covisible(kfid);
    }
    kf_flat_[(size_t) kfid] = nullptr;
    keyframes.erase(it);
    n_keyframes--;
}

void Slam::remove_map_point(int id) {  // map_manager.cpp:559-613
    MapPt *mp = mp_raw(id);
    if (!mp) return;
    const ObsList obs = mp->observers();
    for (int kf: obs) {
        auto k = keyframes.find(kf);
        if (k == keyframes.end()) continue;
        k->second->remove(id);
        for (int co: obs)
            if (co != kf) k->second->decrease_covisible(co);
    }
    if (mp->r->observed) cur->remove(id);
    if (mp->r->is3d) n_map_points--;
    mp_flat_[(size_t) id] = nullptr;
    mp_rec_[(size_t) id] = nullptr;
    mp_slot_[(size_t) id] = -1;
    mp_nobs_[(size_t) id] = 0;
    map_points.erase(id);
    if (defer_mp_free_) mp_graveyard_.push_back(mp);
    else destroy_map_point(mp);
}

void Slam::remove_map_point_obs(int mp_id, int kfid) {  // map_manager.cpp:615-647
    auto kf = keyframes.find(kfid);
    if (kf != keyframes.end()) kf->second->remove(mp_id);
    MapPt *m = mp_raw(mp_id);   // (the id -> object table beside mapPoints_: one load instead of a bucket walk)
    if (!m) return;
    m->drop_px(kfid);
    m->remove_obs(kfid);
    sync_nobs(*m);
    if (kf != keyframes.end()) {
        const ObsList obs = m->observers();
        for (int co: obs) {
            auto c = keyframes.find(co);
            if (c != keyframes.end()) {
                kf->second->decrease_covisible(co);
                c->second->decrease_covisible(kfid);
            }
        }
    }
}

void Slam::remove_obs_from_cur(int mp_id) {  // map_manager.cpp:649-679
    cur->remove(mp_id);
    MpRec *m = rec_raw(mp_id);
    if (!m) return;
    m->observed = 0;
}

bool Slam::set_map_point_obs(int mp_id) {  // map_manager.cpp:681-708
    MpRec *m = rec_raw(mp_id);
    if (!m) return false;
    m->observed = 1;
    return true;
}

void Slam::update_frame_covisibility(FrameRec &frame) {  // map_manager.cpp:83-164
    Lap fine;
    std::map<int, int> cov;
    FlatSet local_ids;
    ids_scratch_.clear();  // snapshot: the repair branch below edits frame.kps
    for (const auto &e: frame.kps) ids_scratch_.push_back(e.first);
    // the counts of the reference's std::map<int, int> (:92-103) accumulated in a flat table (keyframe ids are small consecutive
    // integers) and poured into the ordered map afterwards: a std::map's content and order do not depend on how it was filled
    std::vector<int> &count = index_scratch_;
    count.assign((size_t) next_kf_id + 2, 0);
    for (size_t oi = 0; oi < ids_scratch_.size(); oi++) {
        const int id = ids_scratch_[oi];
        prefetch_mp(ids_scratch_.data(), oi, ids_scratch_.size());
        const MpRec *m = rec_raw(id);
        if (!m) {
            remove_map_point_obs(id, frame.kfid);
            remove_obs_from_cur(id);
            continue;
        }
        for (int e = 0; e < m->n_ent; e++) {
            if (!(m->ent[e].flags & MPF_OBS)) continue;
            const int kf = m->ent[e].kf;
            if (kf != frame.kfid) {
                if (kf >= 0 && kf <= next_kf_id) count[(size_t) kf]++;
                else cov[kf] += 1;
            }
        }
    }
    for (int kf = 0; kf <= next_kf_id; kf++)
        if (count[(size_t) kf]) cov[kf] += count[(size_t) kf];
    fine(t_fine[14]);   // covisibility counts
    std::set<int> bad;
    // marks: a = observed by `frame`, b = already in local_ids (see slam.hpp)
    mark_a_.resize((size_t) next_mp_id + 1, 0);
    mark_b_.resize((size_t) next_mp_id + 1, 0);
    touched_a_.clear();
    touched_b_.clear();
    for (const auto &e: frame.kps) {
        mark_a_[(size_t) e.first] = 1;
        touched_a_.push_back(e.first);
    }
    kf_ptrs_.clear();
    for (const auto &c: cov) kf_ptrs_.push_back(kf_raw(c.first));
    FrameRec::refresh_ids3d(kf_ptrs_.data(), kf_ptrs_.size());   // the stale id lists of the keyframes walked below, together
    for (const auto &c: cov) {
        FrameRec *kf = kf_raw(c.first);
        if (kf) {
            kf->covisible[frame.kfid] = c.second;
            // getKeypoints3d(): container order, 3-D only.  Which of a keyframe's points are new to the set is data the branch predictor
            // cannot learn (tracked / already listed / new, interleaved): the list is compacted without a branch first -- a keyframe holds
            // an id once, so the marks only have to be current between keyframes -- and the new ids are inserted in a second pass
            const std::vector<int> &ids = kf->ids3d();
            if (fresh_ids_.size() < ids.size() + 1) fresh_ids_.resize(ids.size() + 1);
            int *fresh = fresh_ids_.data();
            size_t nf = 0;
            const uint8_t *ma = mark_a_.data(), *mb = mark_b_.data();
            for (int kid: ids) {
                fresh[nf] = kid;
                nf += (size_t) !(ma[(size_t) kid] | mb[(size_t) kid]);
            }
            for (size_t i = 0; i < nf; i++) {
                const int kid = fresh[i];
                mark_b_[(size_t) kid] = 1;
                touched_b_.push_back(kid);
                local_ids.insert_new(kid);   // (the marks filter repeats: the key is new)
            }
        } else {
            bad.insert(c.first);
        }
    }
    for (int id: touched_a_) mark_a_[(size_t) id] = 0;
    for (int id: touched_b_) mark_b_[(size_t) id] = 0;
    for (int kf: bad) cov.erase(kf);
    frame.covisible.swap(cov);
    fine(t_fine[15]);   // local ids of the covisible keyframes
    t_fine[27] += (double) local_ids.size(); t_fine[28] += (double) frame.covisible.size();
    if (local_ids.size() > 0.5 * frame.local_map.size()) frame.local_map.swap(local_ids);
    else frame.local_map.insert(local_ids.begin(), local_ids.end());
    fine(t_fine[16]);   // swap / union into the frame's local map
}

}  // namespace alva_slam
