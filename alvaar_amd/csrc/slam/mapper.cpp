// Keyframe processing of the host-side map layer: the behaviour of Mapper (src/slam/src/mapper.cpp) and Optimizer::localBA
// (src/slam/src/optimizer.cpp) of the reference.  The graph work stays here (which keyframes / map points / observations take
// part, what is written back, what is culled); triangulation, the guided Hamming matching and the bundle-adjustment solve are
// stage calls on flattened arrays.
#include "slam.hpp"
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory_resource>

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

void Slam::process_new_keyframe(int kfid) {  // mapper.cpp:9-64
    std::shared_ptr<FrameRec> kf = keyframe(kfid);
    if (!kf) return;
    Lap lap;
    if (kfid > 30) remove_keyframe(kfid - 30);  // "just keep the last 30 keyframes"
    lap(t_fine[9]);
    if (kf->kfid > 0 && kf->n_2d > 0) triangulate_temporal(*kf);
    if (err_) return;
    lap(t_kf[5]);
    if (ready_for_init) {
        if (kfid == 1 && kf->n_3d < 30) {
            reset_requested = true;
            return;
        }
        if (kfid < 10 && kf->n_3d < 3) {
            reset_requested = true;
            return;
        }
    }
    update_frame_covisibility(*kf);
    cur->covisible = kf->covisible;
    lap(t_kf[6]);
    if (kfid > 0) matching_to_local_map(*kf);
    if (err_) return;
    lap(t_kf[7]);
    optimize(kf);
    lap(t_kf[8]);
    flush_medoids();   // this keyframe's descriptor-table edits (new descriptors, merges, culled observations) go to the stages in one piece
}

void Slam::triangulate_temporal(FrameRec &frame) {  // mapper.cpp:144-291
    const std::vector<KeyPt> kps = frame.keypoints2d();
    if (kps.empty()) return;
    // pass 1 (host): the gates that decide which keypoints reach the arithmetic, in the reference's order.  The arithmetic of
    // one keypoint does not depend on the outcome of another (removeMapPointObs / updateMapPoint touch only that keypoint's
    // map point), so the candidates are triangulated in one batch and the outcomes applied in order afterwards.
    struct Cand {
        int id, group;
        KeyPt kfkp;
        std::shared_ptr<FrameRec> kf;
    };
    std::vector<Cand> cands;
    std::vector<int> group_kf;
    std::vector<size_t> cand_of_kp(kps.size(), (size_t) -1);
    for (size_t i = 0; i < kps.size(); i++) {
        MapPt *mp = mp_raw(kps[i].id);
        if (!mp) {
            remove_map_point_obs(kps[i].id, frame.kfid);
            continue;
        }
        if (mp->r->is3d) continue;
        if (mp->n_obs() < 2) continue;
        const int kfid = mp->obs_first();
        if (frame.kfid == kfid) continue;
        std::shared_ptr<FrameRec> kf = keyframe(kfid);
        if (!kf) continue;
        const KeyPt *kk = kf->find(kps[i].id);
        if (!kk) continue;
        int g = -1;
        for (size_t q = 0; q < group_kf.size(); q++)
            if (group_kf[q] == kfid) g = (int) q;
        if (g < 0) {
            g = (int) group_kf.size();
            group_kf.push_back(kfid);
        }
        cand_of_kp[i] = cands.size();
        cands.push_back(Cand{kps[i].id, g, *kk, kf});
    }
    const int n = (int) cands.size(), G = (int) group_kf.size();
    if (!n) return;
    std::vector<double> T((size_t) G * 36), bvl((size_t) n * 3), bvr((size_t) n * 3), wpt((size_t) n * 3), invd((size_t) n), par((size_t) n);
    std::vector<float> ul((size_t) n * 2), ur((size_t) n * 2);
    std::vector<int> grp((size_t) n);
    std::vector<uint8_t> status((size_t) n);
    for (int g = 0; g < G; g++) {
        const FrameRec &kf = *keyframe(group_kf[(size_t) g]);
        const SE3 Tlr = se3_mul(kf.Tcw, frame.Twc);  // Tcicj = Tciw * Twcj (:226-228)
        const SE3 Trl = se3_inverse(Tlr);
        double *o = &T[(size_t) g * 36];
        quat_to_rot(Tlr.q, o); std::memcpy(o + 9, Tlr.t, 24);
        quat_to_rot(Trl.q, o + 12); std::memcpy(o + 21, Trl.t, 24);
        quat_to_rot(kf.Twc.q, o + 24); std::memcpy(o + 33, kf.Twc.t, 24);
    }
    {
        size_t c = 0;
        for (size_t i = 0; i < kps.size(); i++) {
            if (cand_of_kp[i] == (size_t) -1) continue;
            const Cand &cd = cands[c];
            std::memcpy(&bvl[3 * c], cd.kfkp.bv, 24);
            std::memcpy(&bvr[3 * c], kps[i].bv, 24);
            ul[2 * c] = cd.kfkp.unpx[0]; ul[2 * c + 1] = cd.kfkp.unpx[1];
            ur[2 * c] = kps[i].unpx[0]; ur[2 * c + 1] = kps[i].unpx[1];
            grp[c] = cd.group;
            c++;
        }
    }
    if (fail(st->triangulate(n, G, T.data(), grp.data(), bvl.data(), bvr.data(), ul.data(), ur.data(), wpt.data(), invd.data(), status.data(),
                             par.data())))
        return;
    for (int c = 0; c < n; c++) {
        if (status[(size_t) c] == 0) {
            update_map_point(cands[(size_t) c].id, &wpt[3 * (size_t) c], invd[(size_t) c]);  // :283-286
        } else if (par[(size_t) c] > 20.) {
            remove_map_point_obs(cands[(size_t) c].id, frame.kfid);  // :258-262, :274-278
        }
    }
}

bool Slam::matching_to_local_map(FrameRec &frame) {  // mapper.cpp:293-352
    Lap fine;
    const size_t max_local = (size_t) cfg.max_keypoints * 10;
    if (!frame.covisible.empty() && frame.local_map.size() < max_local) {
        int kfid = frame.covisible.begin()->first;
        std::shared_ptr<FrameRec> kf = keyframe(kfid);
        while (!kf && kfid > 0) {
            kfid--;
            kf = keyframe(kfid);
        }
        if (kf) frame.local_map.insert(kf->local_map.begin(), kf->local_map.end());
        // "go for another round" (:316-330): the reference dereferences `keyframe` here unconditionally and looks the SAME keyframe up
        // again; with a null pointer it would crash, so the guard is the only liberty taken
        if (kf && kf->kfid > 0 && frame.local_map.size() < 0.5 * max_local) {
            kf = keyframe(kf->kfid);
            while (!kf && kfid > 0) {
                kfid--;
                kf = keyframe(kfid);
            }
            if (kf) frame.local_map.insert(kf->local_map.begin(), kf->local_map.end());
        }
    }
    fine(t_fine[0]);   // local-map union
    const std::map<int, int> matches = match_to_map(frame, cfg.map_max_proj_px, cfg.map_max_desc_dist, frame.local_map);
    fine(t_fine[1]);   // match_to_map (flatten + stage)
    if (err_ || matches.empty()) return false;
    {   // (the pairs as an array: what the merges a few pairs ahead will touch -- both points' records, key tables, objects -- is prefetched)
        std::vector<int> &pairs = ids_scratch_;
        pairs.clear();
        for (const auto &m: matches) {
            pairs.push_back(m.first);
            pairs.push_back(m.second);
        }
        for (size_t i = 0; i < pairs.size(); i += 2) {
            prefetch_mp_desc(pairs.data(), i, pairs.size(), 2, 6);
            prefetch_mp_desc(pairs.data(), i + 1, pairs.size(), 2, 6);
            merge_map_points(pairs[i], pairs[i + 1]);
        }
    }
    fine(t_fine[2]);   // merges
    return true;
}

// Mapper::matchToMap (mapper.cpp:354-588): flatten the part of the map the call can reach, run the stage, translate indices back.
// The arrays are assembled in the stage's scratch (pinned memory behind the HIP stages: one upload of one contiguous block).
std::map<int, int> Slam::match_to_map(FrameRec &frame, float max_proj_err, float dist_ratio, FlatSet &local) {
    std::map<int, int> result;
    if (local.empty()) return result;
    Lap fine;
    // keyframe table
    std::vector<int> &kf_ids = kf_ids_scratch_;
    kf_ids.clear();
    for (const auto &e: keyframes) kf_ids.push_back(e.first);
    std::sort(kf_ids.begin(), kf_ids.end());
    std::vector<int> &kf_index = index_scratch_;
    kf_index.assign((size_t) next_kf_id + 1, -1);
    for (size_t i = 0; i < kf_ids.size(); i++) kf_index[(size_t) kf_ids[i]] = (int) i;
    if (frame.kfid < 0 || frame.kfid > next_kf_id || kf_index[(size_t) frame.kfid] < 0) return result;
    // map point table: the frame's keypoints first (grid order), then the local map in ITS iteration order
    std::vector<int> &mp_ids = touched_b_;
    mp_ids.clear();
    // id -> row of the table (ids are dense).  The table persists (all -1 between calls) and only the rows used here are reset at the
    // end: ids are never reused and a long run hands out millions, so clearing it per keyframe would cost O(ids ever handed out)
    std::vector<int> &mp_index = mp_index_;
    if (mp_index.size() < (size_t) next_mp_id + 1) mp_index.resize((size_t) next_mp_id + 1 + (size_t) next_mp_id / 2, -1);
    struct ResetIndex {
        std::vector<int> &index;
        const std::vector<int> &ids;
        ~ResetIndex() {
            for (int id: ids) index[(size_t) id] = -1;
        }
    } reset_index{mp_index, mp_ids};
    auto intern = [&](int id) {
        int &slot = mp_index[(size_t) id];
        if (slot < 0) {
            slot = (int) mp_ids.size();
            mp_ids.push_back(id);
        }
        return slot;
    };
    const size_t n_grid = frame.grid.size();
    std::vector<int> &cell_mp_v = ids_scratch_;   // cell lists and local list are small: built in vectors, copied into the block below
    std::vector<int> &cell_ptr_v = obs_scratch_;
    cell_mp_v.clear();
    cell_ptr_v.assign(n_grid + 1, 0);
    for (size_t c = 0; c < n_grid; c++) {
        cell_ptr_v[c] = (int) cell_mp_v.size();
        const CellIds &cell_ids = frame.grid[c];
        for (size_t ci = 0; ci < cell_ids.size(); ci++) {
            const int id = cell_ids[ci];
            // getSurroundingKeypoints keeps ids found in mapKeypoints_ (frame.cpp:333-337); a keypoint whose map point is gone is
            // repaired by the reference on contact (:459-463) -- repaired here up front
            // (the map point exists <=> its slot of the pointer table is set; that the keyframe holds the keypoint follows from the id
            // standing in the keyframe's own grid -- the record is not touched for it; the mirror check still asks the containers)
            if (!rec_raw(id)) continue;
            if (check_obs_mirror_ && !obs_of(*mp_raw(id), frame.kfid)) continue;
            cell_mp_v.push_back(intern(id));
        }
    }
    cell_ptr_v[n_grid] = (int) cell_mp_v.size();
    fine(t_fine[3]);   // keyframe table + grid cells
    std::vector<int> &local_idx = local_scratch_;
    local_idx.clear();
    mark_a_.resize((size_t) next_mp_id + 1, 0);
    touched_a_.clear();
    frame.for_each_id([&](int kid, bool) {
        mark_a_[(size_t) kid] = 1;
        touched_a_.push_back(kid);
    });
    for (int id: local) {
        if (id >= 0 && id <= next_mp_id ? mark_a_[(size_t) id] != 0 : frame.observes(id)) continue;   // frame.isObservingKeypoint (:397-400)
        const MpRec *mp = rec_raw(id);
        if (!mp || !mp->is3d || !mp->has_desc) continue;         // :404-411
        local_idx.push_back(intern(id));
    }
    for (int id: touched_a_) mark_a_[(size_t) id] = 0;
    fine(t_fine[4]);   // local list
    if (local_idx.empty()) return result;
    const int n_mp = (int) mp_ids.size(), n_kf = (int) kf_ids.size(), n_cell = (int) cell_mp_v.size(), n_local = (int) local_idx.size();
    // ---- one block for the call's own arrays: the map itself is NOT flattened -- the stage reads the records (mp_rec.hpp) of the rows'
    //      slots and the descriptor tables of the same slots, so this keyframe's descriptor edits go to the stages first
    flush_medoids();
    if (err_) return result;
    size_t off = 0;
    auto take = [&](size_t bytes) {
        const size_t o = off;
        off += (bytes + 255) / 256 * 256;
        return o;
    };
    const size_t o_cp = take((n_grid + 1) * 4), o_cm = take((size_t) n_cell * 4 + 4), o_ki = take((size_t) n_kf * 4), o_q = take((size_t) n_kf * 32),
                 o_t = take((size_t) n_kf * 24), o_s = take((size_t) n_mp * 4), o_l = take((size_t) n_local * 4), o_m = take((size_t) n_mp * 4);
    uint8_t *blk = st->stage_scratch(off);
    if (!blk) {
        fail(-3);
        return result;
    }
    int *cell_ptr = (int *) (blk + o_cp), *cell_mp = (int *) (blk + o_cm), *kf_id_p = (int *) (blk + o_ki), *mp_slot = (int *) (blk + o_s),
        *local_p = (int *) (blk + o_l), *match_of_mp = (int *) (blk + o_m);
    double *kf_q = (double *) (blk + o_q), *kf_t = (double *) (blk + o_t);
    std::memcpy(cell_ptr, cell_ptr_v.data(), (n_grid + 1) * 4);
    std::memcpy(cell_mp, cell_mp_v.data(), (size_t) n_cell * 4);
    std::memcpy(local_p, local_idx.data(), (size_t) n_local * 4);
    for (int i = 0; i < n_kf; i++) {
        const FrameRec &k = *kf_raw(kf_ids[(size_t) i]);
        kf_id_p[i] = kf_ids[(size_t) i];
        std::memcpy(kf_q + 4 * (size_t) i, k.Tcw.q, 32);
        std::memcpy(kf_t + 3 * (size_t) i, k.Tcw.t, 24);
    }
    size_t no = 0;
    for (int m = 0; m < n_mp; m++) {
        mp_slot[m] = mp_slot_[(size_t) mp_ids[(size_t) m]];   // (the records themselves are not touched here: the stage gathers them)
        no += mp_nobs_[(size_t) mp_ids[(size_t) m]];
        match_of_mp[m] = -1;
        if (check_obs_mirror_) {   // the records against the authoritative containers: every observer's keypoint, every descriptor key
            const MpRec &mp = *rec_raw(mp_ids[(size_t) m]);
            const MapPt &o = *mp_raw(mp_ids[(size_t) m]);
            for (int e = 0; e < mp.n_ent; e++) {
                const ObsEnt &en = mp.ent[e];
                if (!(en.flags & (MPF_OBS | MPF_INKF))) {
                    std::fprintf(stderr, "alva_slam: record entry without a reason to exist (map point %d, keyframe %d)\n", mp.id, en.kf);
                    std::abort();
                }
                if ((en.flags & MPF_OBS) && kf_raw(en.kf)) (void) obs_of(o, en.kf);
            }
        }
    }
    fine(t_fine[5]);   // per-map-point slots
    t_fine[20] += (double) n_mp; t_fine[21] += (double) no; t_fine[22] += (double) n_local;
    Lap lap;
    MatchJob job;
    job.cell_size = (int) frame.cell; job.num_cells_w = (int) frame.cells_w; job.grid_cells = (int) n_grid;
    job.cell_ptr = cell_ptr; job.cell_mp = cell_mp;
    job.n_kf = n_kf; job.kf_ids = kf_id_p; job.kf_q = kf_q; job.kf_t = kf_t;
    job.n_mp = n_mp; job.mp_slot = mp_slot;
    job.frame_kfid = frame.kfid; job.num_keypoints_3d = (int) frame.n_3d;
    job.n_local = n_local; job.local = local_p;
    job.max_proj_err = max_proj_err; job.dist_ratio = dist_ratio;
    const int rc = st->match_to_map_rec(job, match_of_mp);
    lap(t_kf[9]);
    if (fail(rc)) return result;
    for (int m = 0; m < n_mp; m++)
        if (match_of_mp[(size_t) m] >= 0) result.emplace(mp_ids[(size_t) m], mp_ids[(size_t) match_of_mp[(size_t) m]]);
    return result;
}

void Slam::optimize(const std::shared_ptr<FrameRec> &kf) {  // mapper.cpp:66-142
    if (kf->kfid >= 2 && kf->n_3d != 0) local_ba(*kf);
    if (err_) return;
    if (cfg.keyframe_filtering_ratio < 1.0 && kf->kfid >= 20) {
        const std::map<int, int> cov = kf->covisible;
        kf_ptrs_.clear();
        for (const auto &c: cov)
            if (c.first != 0 && c.first < kf->kfid) kf_ptrs_.push_back(kf_raw(c.first));
        FrameRec::refresh_ids3d(kf_ptrs_.data(), kf_ptrs_.size());   // (a keyframe removed below is not walked after its removal)
        for (auto it = cov.rbegin(); it != cov.rend(); ++it) {
            const int kfid = it->first;
            if (kfid == 0) break;
            if (kfid >= kf->kfid) continue;
            FrameRec *co = kf_raw(kfid);
            if (!co) continue;  // the reference dereferences the null pointer here (:88-92); nothing to remove from
            if ((int) co->n_3d < cfg.ba_min_common_obs / 2) {
                Lap rm;
                remove_keyframe(kfid);
                rm(t_fine[8]);
                n_kf_culled++;
                continue;
            }
            size_t good = 0, total = 0;
            ids_scratch_.clear();  // keypoints whose map point is gone: the reference repairs them while walking a COPY (:101-111); the
                                   // repair only drops that keypoint from this keyframe, so doing it after the walk is the same thing
            bool counted = false;
            if (!check_obs_mirror_) {
                // Nothing in this loop depends on the ORDER of the walk: good / total are counts, isBad() acts on one map point, and the
                // repairs drop different keypoints of this keyframe (erasing from a hash table or a cell list commutes).  So the table's
                // slots are taken in memory order, with the observer-count byte table in front of the map point.
                for (int kid: co->ids3d()) {
                    const unsigned nobs = mp_nobs_[(size_t) kid];
                    if (nobs >= 2) {  // two observers or more: isBad() is false and has no side effect (map_point.cpp:183-202)
                        good += nobs > 4;
                        total++;
                        continue;
                    }
                    MapPt *mp = mp_raw(kid);
                    if (!mp) {
                        ids_scratch_.push_back(kid);
                        continue;
                    }
                    if (mp->is_bad()) continue;
                    if (mp->n_obs() > 4) good++;
                    total++;
                }
                counted = true;
            }
            if (!counted)
            co->for_each_id([&](int kid, bool is3d) {
                if (!is3d) return;
                const unsigned nobs = mp_nobs_[(size_t) kid];
                if (nobs >= 2 && !check_obs_mirror_) {  // two observers or more: isBad() is false and has no side effect (map_point.cpp:183-202)
                    if (nobs > 4) good++;
                    total++;
                    return;
                }
                MapPt *mp = mp_raw(kid);
                if (mp && check_obs_mirror_ && (unsigned) mp->n_obs() != nobs) {
                    std::fprintf(stderr, "alva_slam: observer count mirror out of sync (map point %d)\n", mp->id());
                    std::abort();
                }
                if (!mp) {
                    ids_scratch_.push_back(kid);
                    return;
                } else if (mp->is_bad()) {
                    return;
                } else if (mp->n_obs() > 4) {
                    good++;
                }
                total++;
            });
            for (int id: ids_scratch_) remove_map_point_obs(id, kfid);
            const float ratio = (float) good / (float) total;
            if (ratio > cfg.keyframe_filtering_ratio) {
                Lap rm;
                remove_keyframe(kfid);
                rm(t_fine[8]);   // keyframe filter: removals
                n_kf_culled++;
            }
        }
    }
}

// Optimizer::localBA (optimizer.cpp:4-531) with anchored inverse depth (state.hpp:74)
void Slam::local_ba(FrameRec &new_frame) {
    const int min_cov = cfg.ba_min_common_obs;
    if ((int) new_frame.n_3d < min_cov) return;
    Lap lap_ba;
    // ---- 1. problem (optimizer.cpp:20-247)
    // The function-local hash containers of the reference live in a bump arena: same container code, same hash, same growth policy
    // => same iteration order; only where the nodes come from differs.  Map points removed during the write-back stay allocated
    // until the end of the call (the reference's local shared_ptr copies do the same), so the table can hold plain pointers.
    if (ba_arena_.size() < ((size_t) 4 << 20)) ba_arena_.resize((size_t) 4 << 20);
    std::pmr::monotonic_buffer_resource arena(ba_arena_.data(), ba_arena_.size());
    struct Undefer {
        Slam *s;
        ~Undefer() {
            s->defer_mp_free_ = false;
            for (MapPt *mp: s->mp_graveyard_) s->destroy_map_point(mp);
            s->mp_graveyard_.clear();
        }
    } undefer{this};
    defer_mp_free_ = true;
    // (the two big ones -- thousands of map points -- on flat arrays with libstdc++'s order, flat_hash.hpp; capacity kept between calls)
    FlatHash<MpRec *> &local_mps = ba_scratch_.local_mps;                         // map_local_plms
    local_mps.reset();
    std::pmr::unordered_map<int, std::shared_ptr<FrameRec>> local_kfs(&arena);    // map_local_pkfs
    // keyframe id -> row of the flat pose table / keyframe object: ids are small consecutive integers, so plain arrays beside the
    // reference's hash maps (which stay, because their iteration ORDER is behaviour, :234-247)
    std::vector<int> pose_slot((size_t) next_kf_id + 1, -1);
    std::vector<FrameRec *> kf_flat((size_t) next_kf_id + 1, nullptr);
    std::vector<double> poses;
    std::vector<uint8_t> kf_const;
    std::pmr::unordered_set<int> bad_mps(&arena), kfs_to_opt(&arena), const_kfs(&arena);
    FlatSet &mps_to_opt = ba_scratch_.mps_to_opt;
    mps_to_opt.reset();
    auto add_pose = [&](int kfid, const FrameRec &kf, bool constant) {
        pose_slot[(size_t) kfid] = (int) kf_const.size();
        double p[7];
        se3_to_pose7(kf.Twc, p);
        poses.insert(poses.end(), p, p + 7);
        kf_const.push_back(constant ? 1 : 0);
    };
    std::map<int, int> cov = new_frame.covisible;
    cov.emplace(new_frame.kfid, (int) new_frame.n_3d);
    mark_a_.resize((size_t) next_mp_id + 1, 0);
    touched_a_.clear();
    bool all_cst = false;
    const int max_kfid = cov.rbegin()->first;
    {   // the keyframes whose points the loop below collects (same rule): their stale id lists are rebuilt together (FrameRec::refresh_ids3d)
        kf_ptrs_.clear();
        for (auto it = cov.rbegin(); it != cov.rend(); ++it) {
            const int kfid = it->first;
            FrameRec *kf = kf_raw(kfid);
            if (!kf) continue;
            if ((kfid > new_frame.kfid ? (int) new_frame.n_kps : it->second) >= min_cov && kfid > 0) kf_ptrs_.push_back(kf);
            else break;
        }
        FrameRec::refresh_ids3d(kf_ptrs_.data(), kf_ptrs_.size());
    }
    for (auto it = cov.rbegin(); it != cov.rend(); ++it) {
        const int kfid = it->first;
        int score = it->second;
        if (kfid > new_frame.kfid) score = (int) new_frame.n_kps;
        std::shared_ptr<FrameRec> kf = keyframe(kfid);
        if (!kf) {
            new_frame.remove_covisible(kfid);
            continue;
        }
        if (score >= min_cov && !all_cst && kfid > 0) {
            add_pose(kfid, *kf, false);
            kfs_to_opt.insert(kfid);
            {   // (compacted without a branch first, like update_frame_covisibility's list: a repeated insert would not change the set)
                const std::vector<int> &ids = kf->ids3d();
                if (fresh_ids_.size() < ids.size() + 1) fresh_ids_.resize(ids.size() + 1);
                int *fresh = fresh_ids_.data();
                size_t nf = 0;
                const uint8_t *ma = mark_a_.data();
                for (int kid: ids) {
                    fresh[nf] = kid;
                    nf += (size_t) !ma[(size_t) kid];
                }
                for (size_t i = 0; i < nf; i++) {
                    const int kid = fresh[i];
                    mark_a_[(size_t) kid] = 1;
                    touched_a_.push_back(kid);
                    mps_to_opt.insert_new(kid);   // (the marks filter repeats: the key is new)
                }
            }
        } else {
            add_pose(kfid, *kf, true);
            const_kfs.insert(kfid);
            all_cst = true;
        }
        local_kfs.emplace(kfid, kf);
        kf_flat[(size_t) kfid] = kf.get();
    }
    for (int id: touched_a_) mark_a_[(size_t) id] = 0;
    Lap fine;
    t_fine[6] += std::chrono::duration<double>(fine.t0 - lap_ba.t0).count();   // BA build phase 1: keyframes + points to optimise
    // The problem's arrays live in a member (capacity persists from keyframe to keyframe: no allocator traffic, no first-touch faults) and
    // are written ONCE, in the form the solve consumes (Stages::local_ba_csr): residual blocks grouped by point behind a pt_ptr table,
    // only points that have a residual block (Ceres drops parameter blocks that no residual block uses, program.cc RemoveFixedBlocks:
    // an anchored point that nothing else observes keeps its inverse depth and is remembered in `lone`).  Round 4 pushed every observation
    // through five vectors here, copied the live ones into a second set per round and the stage sorted / copied them a third time.
    BaScratch &bs = ba_scratch_;
    std::vector<int> &pt_ids = bs.pt_ids, &anc_slot = bs.pt_anchor_slot, &obs_kf = bs.obs_kf, &pt_ptr = bs.pt_ptr, &lone_ids = bs.lone_ids,
                     &slot_ids = bs.slot_ids;
    std::vector<double> &anc_uv = bs.pt_anchor_uv, &pinv = bs.pt_inv, &obs_uv = bs.obs_uv, &lone_inv = bs.lone_inv;
    const size_t max_pts = mps_to_opt.size();
    pt_ids.resize(max_pts); anc_slot.resize(max_pts); anc_uv.resize(2 * max_pts); pinv.resize(max_pts); pt_ptr.resize(max_pts + 1);
    lone_ids.clear(); lone_inv.clear(); slot_ids.clear();
    size_t obs_cap = std::max<size_t>(obs_kf.size(), 8 * max_pts + 64);
    obs_kf.resize(obs_cap); obs_uv.resize(2 * obs_cap);
    int n_used = 0, n_obs = 0;
    // map_id_invptspar_ is only looked up by id: a persistent id -> slot table (mp_index_, all -1 between calls, see match_to_map):
    // j >= 0 = the point's row in the solve, -2 - i = lone point i
    std::vector<int> &pt_slot = mp_index_;
    if (pt_slot.size() < (size_t) next_mp_id + 1) pt_slot.resize((size_t) next_mp_id + 1 + (size_t) next_mp_id / 2, -1);
    struct ResetSlots {
        std::vector<int> &index;
        const std::vector<int> &ids;
        ~ResetSlots() {
            for (int id: ids) index[(size_t) id] = -1;
        }
    } reset_slots{pt_slot, slot_ids};
    std::vector<uint8_t> &kf_used = bs.kf_used;
    kf_used.assign(64, 0);
    ids_scratch_.clear();
    for (int id: mps_to_opt) ids_scratch_.push_back(id);   // the set's order, as an array (for the prefetcher; the loop below does not edit the set)
    for (size_t oi = 0; oi < ids_scratch_.size(); oi++) {
        const int lmid = ids_scratch_[oi];
        prefetch_mp(ids_scratch_.data(), oi, ids_scratch_.size());
        MpRec *recp = rec_raw(lmid);   // (the record alone: the map point's object is not touched in this loop)
        if (!recp) continue;
        if (rec_is_bad(*recp)) {
            bad_mps.insert(lmid);
            continue;
        }
        local_mps.insert_new_slot(lmid, recp);   // (ids of a set: the key is new)
        if ((size_t) n_obs + MP_ENT_CAP > obs_cap) {
            obs_cap *= 2;
            obs_kf.resize(obs_cap);
            obs_uv.resize(2 * obs_cap);
        }
        int anchor = -1;
        const int q0 = n_obs;
        const MpRec &rec = *recp;
        // The common case -- every observing keyframe exists and holds the keypoint -- straight over the record's entries, no snapshot.
        // The first entry that would need a repair ends it: this point's emission is undone and the reference's own loop (below: a copy
        // of the observer set, repairs as they come) runs from the start; keyframes that joined meanwhile stay joined, in the same order.
        bool regular = !check_obs_mirror_;
        if (regular) {
            for (int e = 0; e < rec.n_ent; e++) {
                const ObsEnt &en = rec.ent[e];
                if (!(en.flags & MPF_OBS)) continue;
                const int kfid = en.kf;
                if (kfid > max_kfid) continue;
                FrameRec *kf = kf_flat[(size_t) kfid];
                if (!kf) {
                    std::shared_ptr<FrameRec> sp = keyframe(kfid);
                    if (!sp) {
                        regular = false;
                        break;
                    }
                    local_kfs.emplace(kfid, sp);
                    kf = kf_flat[(size_t) kfid] = sp.get();
                    add_pose(kfid, *kf, true);
                    const_kfs.insert(kfid);
                }
                if (!(en.flags & MPF_INKF)) {
                    regular = false;
                    break;
                }
                const int ps = pose_slot[(size_t) kfid];
                if (anchor < 0) {
                    anchor = kfid;
                    double pc[3];
                    se3_apply(kf->Tcw, rec.X, pc);
                    anc_slot[(size_t) n_used] = ps;
                    anc_uv[2 * (size_t) n_used] = (double) en.unpx[0];
                    anc_uv[2 * (size_t) n_used + 1] = (double) en.unpx[1];
                    pinv[(size_t) n_used] = 1. / pc[2];
                    continue;
                }
                obs_kf[(size_t) n_obs] = ps;
                obs_uv[2 * (size_t) n_obs] = (double) en.unpx[0];
                obs_uv[2 * (size_t) n_obs + 1] = (double) en.unpx[1];
                n_obs++;
            }
            if (!regular) {
                n_obs = q0;
                anchor = -1;
            }
        }
        const ObsList obs = regular ? ObsList() : rec_observers(rec);  // snapshot (getObservedKeyframeIds returns a copy): the repair branches edit the set
        int si = 0;   // walks the record's entries (sorted by keyframe like obs) beside the observers; a repair below edits them: start over
        for (int kfid: obs) {
            if (kfid > max_kfid) continue;
            FrameRec *kf = kf_flat[(size_t) kfid];
            if (!kf) {  // an observing keyframe outside the covisibility set joins as a constant one (:153-172)
                std::shared_ptr<FrameRec> sp = keyframe(kfid);
                if (!sp) {
                    remove_map_point_obs(kfid, lmid);  // sic: arguments swapped in the reference (optimizer.cpp:162)
                    si = 0;
                    continue;
                }
                local_kfs.emplace(kfid, sp);
                kf = kf_flat[(size_t) kfid] = sp.get();
                add_pose(kfid, *kf, true);
                const_kfs.insert(kfid);
            }
            while (si < rec.n_ent && rec.ent[si].kf < kfid) si++;
            const ObsEnt *kp = si < rec.n_ent && rec.ent[si].kf == kfid && (rec.ent[si].flags & MPF_INKF) ? &rec.ent[si] : nullptr;
            if (check_obs_mirror_ && kp != obs_of(*mp_raw(lmid), kfid)) {
                std::fprintf(stderr, "alva_slam: sorted observation walk out of sync in localBA (map point %d, keyframe %d)\n", lmid, kfid);
                std::abort();
            }
            if (!kp) {
                remove_map_point_obs(lmid, kfid);
                si = 0;
                continue;
            }
            const int ps = pose_slot[(size_t) kfid];
            if (anchor < 0) {  // the first observing keyframe anchors the inverse depth; it gets no residual (:186-201)
                anchor = kfid;
                double pc[3];
                se3_apply(kf->Tcw, rec.X, pc);
                anc_slot[(size_t) n_used] = ps;
                anc_uv[2 * (size_t) n_used] = (double) kp->unpx[0];
                anc_uv[2 * (size_t) n_used + 1] = (double) kp->unpx[1];
                pinv[(size_t) n_used] = 1. / pc[2];  // InvDepthParametersBlock(id, anchor, zanch) stores 1 / zanch
                continue;
            }
            obs_kf[(size_t) n_obs] = ps;
            obs_uv[2 * (size_t) n_obs] = (double) kp->unpx[0];
            obs_uv[2 * (size_t) n_obs + 1] = (double) kp->unpx[1];
            n_obs++;
        }
        if (anchor < 0) continue;
        slot_ids.push_back(lmid);
        if (n_obs == q0) {   // anchored, no residual block: not a parameter of the solve
            pt_slot[(size_t) lmid] = -2 - (int) lone_ids.size();
            lone_ids.push_back(lmid);
            lone_inv.push_back(pinv[(size_t) n_used]);
            continue;
        }
        pt_slot[(size_t) lmid] = n_used;
        pt_ids[(size_t) n_used] = lmid;
        pt_ptr[(size_t) n_used] = q0;
        n_used++;
    }
    pt_ptr[(size_t) n_used] = n_obs;
    fine(t_fine[7]);   // BA build phase 2: observations
    t_fine[23] += (double) (n_used + (int) lone_ids.size()); t_fine[24] += (double) n_obs; t_fine[25] += (double) kf_const.size();
    // gauge: at least two constant keyframes (:234-247), taken in the container's order
    size_t n_const = const_kfs.size();
    if (n_const < 2) {
        for (auto it = local_kfs.begin(); n_const < 2 && it != local_kfs.end(); ++it) {
            kf_const[(size_t) pose_slot[(size_t) it->first]] = 1;
            const_kfs.insert(it->first);
            n_const++;  // sic: counted even when the keyframe was constant already
        }
    }
    lap_ba(t_kf[11]);
    // ---- 2. solve (:251-262) and 3./4. outlier sweep + second solve without the flagged residuals (:266-359; the loss is never
    //         reset to L2 there because vright_reprojerr_kfid_lmid stays empty, :315-318)
    const int n_kf = (int) kf_const.size();
    if (n_kf > 64) {
        fail(-1);
        return;
    }
    std::vector<int> &slot_kfid = bs.slot_kfid;   // pose slot -> keyframe id
    slot_kfid.assign((size_t) n_kf, -1);
    for (const auto &e: local_kfs) slot_kfid[(size_t) pose_slot[(size_t) e.first]] = e.first;
    std::vector<std::pair<int, int>> bad_obs;  // (keyframe, map point)
    std::vector<uint64_t> &bits = bs.bad_bits;
    std::vector<uint8_t> &kc = bs.kc;
    // a free keyframe without residual blocks is passed as constant (Ceres would drop its parameter block)
    auto run_round = [&](int np, const int *ptr, const int *aslot, const double *auv, double *inv, int no, const int *okf, const double *ouv,
                         const int *ids) -> int {
        kf_used.assign((size_t) n_kf, 0);
        for (int q = 0; q < no; q++) kf_used[(size_t) okf[q]] = 1;
        for (int j = 0; j < np; j++) kf_used[(size_t) aslot[j]] = 1;
        kc = kf_const;
        for (int k = 0; k < n_kf; k++)
            if (!kf_used[(size_t) k]) kc[(size_t) k] = 1;
        bits.assign((size_t) no / 64 + 2, 0);
        int n_bad = 0;
        if (no > 0) {
            Lap lap;
            if (fail(st->local_ba_csr(n_kf, poses.data(), kc.data(), np, ptr, aslot, auv, inv, no, okf, ouv, 5, (double) cfg.robust_threshold, bits.data(),
                                      &n_bad)))
                return -1;
            lap(t_kf[10]);
        }
        n_ba_runs++;
        // the flagged residual blocks, in residual order: (keyframe id, map point id)
        int j = 0;
        for (size_t w = 0; w * 64 < (size_t) no; w++) {
            uint64_t v = bits[w];
            while (v) {
                const int q = (int) (w * 64) + __builtin_ctzll(v);
                v &= v - 1;
                while (ptr[j + 1] <= q) j++;
                bad_obs.emplace_back(slot_kfid[(size_t) okf[q]], ids[j]);
                bad_mps.insert(ids[j]);
            }
        }
        return n_bad;
    };
    const int n_bad0 = run_round(n_used, pt_ptr.data(), anc_slot.data(), anc_uv.data(), pinv.data(), n_obs, obs_kf.data(), obs_uv.data(), pt_ids.data());
    if (n_bad0 < 0) return;
    if (cfg.refine_with_l2 && n_bad0 > 0) {
        // the second round's problem: the first one without the flagged residual blocks (and without the points that lose their last one)
        std::vector<int> &ptr2 = bs.ptr2, &as2 = bs.as2, &okf2 = bs.okf2, &ids2 = bs.ids2, &from2 = bs.from2;
        std::vector<double> &auv2 = bs.auv2, &inv2 = bs.inv2, &ouv2 = bs.ouv2;
        ptr2.resize((size_t) n_used + 1); as2.resize((size_t) n_used); ids2.resize((size_t) n_used); from2.resize((size_t) n_used);
        auv2.resize(2 * (size_t) n_used); inv2.resize((size_t) n_used); okf2.resize((size_t) n_obs); ouv2.resize(2 * (size_t) n_obs);
        const std::vector<uint64_t> bits0 = bits;
        int np2 = 0, no2 = 0;
        for (int j = 0; j < n_used; j++) {
            const int q0 = no2;
            for (int q = pt_ptr[(size_t) j]; q < pt_ptr[(size_t) j + 1]; q++) {
                if ((bits0[(size_t) q >> 6] >> (q & 63)) & 1) continue;
                okf2[(size_t) no2] = obs_kf[(size_t) q];
                ouv2[2 * (size_t) no2] = obs_uv[2 * (size_t) q];
                ouv2[2 * (size_t) no2 + 1] = obs_uv[2 * (size_t) q + 1];
                no2++;
            }
            if (no2 == q0) continue;
            ptr2[(size_t) np2] = q0;
            as2[(size_t) np2] = anc_slot[(size_t) j];
            auv2[2 * (size_t) np2] = anc_uv[2 * (size_t) j];
            auv2[2 * (size_t) np2 + 1] = anc_uv[2 * (size_t) j + 1];
            inv2[(size_t) np2] = pinv[(size_t) j];
            ids2[(size_t) np2] = pt_ids[(size_t) j];
            from2[(size_t) np2] = j;
            np2++;
        }
        ptr2[(size_t) np2] = no2;
        if (run_round(np2, ptr2.data(), as2.data(), auv2.data(), inv2.data(), no2, okf2.data(), ouv2.data(), ids2.data()) < 0) return;
        if (no2 > 0)
            for (int j2 = 0; j2 < np2; j2++) pinv[(size_t) from2[(size_t) j2]] = inv2[(size_t) j2];
    }
    lap_ba(t_kf[12]);
    // ---- 5. write-back (:363-530)
    for (const auto &b: bad_obs) {
        if (local_kfs.find(b.first) != local_kfs.end()) remove_map_point_obs(b.second, b.first);
        if (b.first == cur->kfid) remove_obs_from_cur(b.second);
        bad_mps.insert(b.second);
    }
    for (const auto &e: local_kfs) {
        if (const_kfs.count(e.first)) continue;
        if (!e.second) continue;
        const int ps = pose_slot[(size_t) e.first];
        if (ps >= 0) e.second->set_Twc(se3_from_pose7(&poses[7 * (size_t) ps]));
    }
    // (map_local_plms's order as two arrays first: the body does not edit the container, and the records of the points ahead can be
    // prefetched -- 4 500 records of 1 KB in an order only the container knows)
    std::vector<int> &wb_ids = bs.wb_ids;
    std::vector<MpRec *> &wb_recs = bs.wb_recs;
    wb_ids.clear();
    wb_recs.clear();
    for (int ls = local_mps.first(); ls != FlatHash<MpRec *>::END; ls = local_mps.next(ls)) {
        wb_ids.push_back(local_mps.key(ls));
        wb_recs.push_back(local_mps.val(ls));
    }
    for (size_t wi = 0; wi < wb_ids.size(); wi++) {
        if (wi + 8 < wb_ids.size() && wb_recs[wi + 8]) {
            const char *c = (const char *) wb_recs[wi + 8];
            __builtin_prefetch(c, 1);
            __builtin_prefetch(c + 64);
            __builtin_prefetch(c + 128);
        }
        const int lmid = wb_ids[wi];
        MpRec *wrp = wb_recs[wi];
        if (!wrp) {
            bad_mps.erase(lmid);
            continue;
        }
        MpRec &wr = *wrp;
        if (rec_is_bad(wr)) {
            remove_map_point(lmid);
            bad_mps.erase(lmid);
            continue;
        }
        if (wr.n_obs < 3) {
            if (wr.anchor_kf < new_frame.kfid - 3 && !wr.observed) {
                remove_map_point(lmid);
                bad_mps.erase(lmid);
                continue;
            }
        }
        const int ps = pt_slot[(size_t) lmid];
        if (ps == -1) {
            bad_mps.insert(lmid);
            continue;
        }
        const double inv = ps >= 0 ? pinv[(size_t) ps] : lone_inv[(size_t) (-2 - ps)], zanch = 1. / inv;
        if (zanch <= 0.) {
            remove_map_point(lmid);
            bad_mps.erase(lmid);
            continue;
        }
        const FrameRec *akp = wr.anchor_kf >= 0 && (size_t) wr.anchor_kf < kf_flat.size() ? kf_flat[(size_t) wr.anchor_kf] : nullptr;
        if (!akp) {  // the anchor keyframe is not part of the problem (:459-463)
            bad_mps.insert(lmid);
            continue;
        }
        {
            const FrameRec &akf = *akp;
            const ObsEnt *kp = check_obs_mirror_ && mp_raw(lmid) ? obs_of(*mp_raw(lmid), akf.kfid) : rec_in_kf(wr, akf.kfid);
            const float ux = kp ? kp->unpx[0] : 0.f, uy = kp ? kp->unpx[1] : 0.f;  // a default Keypoint has unpx_ = (0, 0)
            const double uv[3] = {(double) ux, (double) uy, 1.};
            double ray[3], pc[3], wpt[3];
            double sK[9];
            for (int i = 0; i < 9; i++) sK[i] = zanch * invK[i];  // (zanch * inverseK_) * uvpt, left to right (:468-470)
            mat3_vec(sK, uv, ray);
            (void) pc;
            se3_apply(akf.Twc, ray, wpt);
            update_map_point(lmid, wpt, inv);
        }
    }
    lap_ba(t_kf[13]);
    for (int lmid: bad_mps) {  // :492-530
        const int lm = local_mps.find_slot(lmid);
        MpRec *rp = lm == FlatHash<MpRec *>::END ? rec_raw(lmid) : local_mps.val(lm);
        if (!rp) continue;
        if (rec_is_bad(*rp)) {
            remove_map_point(lmid);
        } else if (rp->n_obs < 3) {
            if (rp->anchor_kf < new_frame.kfid - 3 && !rp->observed) remove_map_point(lmid);
        }
    }
    lap_ba(t_kf[14]);
}

}  // namespace alva_slam
