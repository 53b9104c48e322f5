// Per-frame state machine of the host-side map layer: the behaviour of System::processCameraPose (src/slam/src/system.cpp:156-175)
// and VisualFrontend (src/slam/src/visual_frontend.cpp) of the reference, with every numeric stage behind `Stages`.
#include "slam.hpp"
#if defined(__x86_64__)
#include <immintrin.h>
#endif
#include <chrono>
#include <cstdio>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <stdexcept>

namespace alva_slam {

namespace {
struct Section {  // adds the scope's wall time to one slot of Slam::t_section
    double &acc;
    std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
    explicit Section(double &a) : acc(a) {}
    ~Section() { acc += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(); }
};
}  // namespace

Slam::Slam(Stages *stages, const Camera &c, const Settings &s) : st(stages), cam(c), cfg(s) {
    // State::State (state.cpp:3-12)
    const float cw = std::ceil((float) cam.width / (float) cfg.cell_size), ch = std::ceil((float) cam.height / (float) cfg.cell_size);
    cfg.max_keypoints = (int) (cw * ch);
    // CameraCalibration: inverseK_ = K_.inverse() (camera_calibration.cpp:15) -- cofactor form of a 3x3 inverse
    const double K[9] = {cam.fx, 0, cam.cx, 0, cam.fy, cam.cy, 0, 0, 1};
    const double c00 = K[4] * K[8] - K[5] * K[7], c10 = K[5] * K[6] - K[3] * K[8], c20 = K[3] * K[7] - K[4] * K[6];
    const double det = c00 * K[0] + c10 * K[1] + c20 * K[2], id = 1.0 / det;
    invK[0] = c00 * id;
    invK[1] = (K[2] * K[7] - K[1] * K[8]) * id;
    invK[2] = (K[1] * K[5] - K[2] * K[4]) * id;
    invK[3] = c10 * id;
    invK[4] = (K[0] * K[8] - K[2] * K[6]) * id;
    invK[5] = (K[2] * K[3] - K[0] * K[5]) * id;
    invK[6] = c20 * id;
    invK[7] = (K[1] * K[6] - K[0] * K[7]) * id;
    invK[8] = (K[0] * K[4] - K[1] * K[3]) * id;
    cur = std::make_shared<FrameRec>();
    cur->init(&cam, (size_t) cfg.cell_size);
    st->image_width_ = cam.width;
    st->image_height_ = cam.height;
    ba_arena_.assign((size_t) 4 << 20, 0);   // local_ba's arena, touched now rather than inside the first keyframe that optimises
    const char *chk = std::getenv("ALVA_CHECK_OBS_MIRROR");
    check_obs_mirror_ = chk && chk[0] == '1';
    const char *chk_carry = std::getenv("ALVA_CHECK_CARRY");
    check_carry_ = chk_carry && chk_carry[0] == '1';
}

void Slam::reset() {  // System::reset (system.cpp:42-55)
    slots_dirty_ = true;
    cur->reset();
    // VisualFrontend::reset (visual_frontend.cpp:716-727): images, pyramids, the failure counter -- p3pReq_ and the motion model stay
    st->reset_images();
    pose_failed = 0;
    // MapManager::reset (map_manager.cpp:710-722)
    next_mp_id = next_kf_id = n_map_points = n_keyframes = 0;
    keyframes.clear();
    for (auto e: map_points) destroy_map_point(e.second);   // (mapMapPoints_.clear(): every map point goes, in the container's order)
    map_points.clear();
    kf_flat_.clear();
    mp_flat_.clear();
    mp_rec_.clear();
    mp_slot_.clear();
    mp_nobs_.clear();
    mp_index_.clear();
    shared_ids.clear();
    // State::reset (state.cpp:14-18)
    ready_for_init = false;
    reset_requested = false;
}

int Slam::process_frame(const uint8_t *rgba, double timestamp, bool frame_on_device) {  // system.cpp:156-175
    err_ = 0;
    cur->id++;
    cur->timestamp = timestamp;
    track(rgba, timestamp, frame_on_device);
    fail(st->frame_done());
    if (err_) return err_;
    if (reset_requested) {
        reset();
        return 2;
    }
    if (!ready_for_init) return 3;
    return 1;
}

bool Slam::track(const uint8_t *rgba, double timestamp, bool frame_on_device) {  // visual_frontend.cpp:21-35
    {
        Section sec(t_section[0]);
        // cvtColor (system.cpp:112) + preprocessImage (:672-698)
        if (fail(frame_on_device ? st->new_frame_device(rgba) : st->new_frame(rgba))) return false;
        if (next_frame_hint) {   // alva_system_hint_next_frame_device: the stages may build that frame's images beside this frame's pose solve
            if (frame_on_device) st->hint_next_frame_device(next_frame_hint);
            next_frame_hint = nullptr;
        }
    }
    const bool kf_required = process(timestamp);
    if (kf_required || reset_requested || !ready_for_init) slots_dirty_ = true;   // (the keyframe steps below add keypoints, move world points, ...)
    if (err_) return false;
    if (kf_required) {
        {
            Section sec(t_section[6]);
            create_keyframe();
        }
        if (err_) return false;
        if (!reset_requested && ready_for_init) {
            Section sec(t_section[7]);
            process_new_keyframe(cur->kfid);
        }
    }
    return true;
}

bool Slam::process(double timestamp) {  // visual_frontend.cpp:37-101
    if (cur->id == 0) return true;
    SE3 Twc = cur->Twc;
    apply_motion_model(Twc, timestamp);
    cur->set_Twc(Twc);
    klt_from_motion_prior();
    if (err_) return false;
    if (!ready_for_init) {
        if (cur->n_2d < 50) {
            reset_requested = true;
            return false;
        }
        if (check_ready_for_init()) {
            ready_for_init = true;
            return true;
        }
        return false;
    }
    {
        const auto t_pp0 = std::chrono::steady_clock::now();
        prepare_parallax();   // host work that only needs the tracker's results, placed under the pose solve the GPU is running
        t_fine[17] += std::chrono::duration<double>(std::chrono::steady_clock::now() - t_pp0).count();
    }
    const bool ok = compute_pose();
    if (err_) return false;
    Section sec(t_section[5]);
    if (!ok) {
        pose_failed++;
        if (pose_failed > 3) {
            reset_requested = true;
            return false;
        }
    }
    update_motion_model(cur->Twc, timestamp);
    return check_new_keyframe_required();
}

void Slam::apply_motion_model(SE3 &Twc, double time) {  // visual_frontend.hpp:17-31
    if (mm_prev_time > 0) {
        double xi[6];
        se3_log(se3_mul(Twc, se3_inverse(mm_prev_Twc)), xi);
        bool zero = true;  // Eigen's isZero(1e-5) on a 6-vector: every |x_i| <= 1e-5 (reference value 1)
        for (double v: xi) zero = zero && std::fabs(v) <= 1e-5;
        if (!zero) mm_prev_Twc = Twc;
        const double dt = time - mm_prev_time;
        double d[6];
        for (int i = 0; i < 6; i++) d[i] = mm_log_rel[i] * dt;
        Twc = se3_mul(Twc, se3_exp(d));
    }
}

void Slam::update_motion_model(const SE3 &Twc, double time) {  // visual_frontend.hpp:33-58
    if (mm_prev_time < 0.) {
        mm_prev_time = time;
        mm_prev_Twc = Twc;
    } else {
        const double dt = time - mm_prev_time;
        mm_prev_time = time;
        // the reference exits the process on dt < 0 (visual_frontend.hpp:46-50); a library must not: report a reset instead
        if (dt < 0.) {
            reset_requested = true;
            return;
        }
        double xi[6];
        se3_log(se3_mul(se3_inverse(mm_prev_Twc), Twc), xi);
        for (int i = 0; i < 6; i++) mm_log_rel[i] = xi[i] / dt;
        mm_prev_Twc = Twc;
    }
}

// VisualFrontend::kltTrackingFromMotionPrior (visual_frontend.cpp:103-243): the frame's keypoints go to the tracking step as slots in
// the container's iteration order; its results are applied in the reference's order (updates of the one-level pass first, then the
// full-pyramid list: the keypoints that never had a prior, then the ones that failed with theirs)
void Slam::klt_from_motion_prior() {
    std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
    auto lap = [&](int slot) {
        const auto t1 = std::chrono::steady_clock::now();
        t_section[slot] += std::chrono::duration<double>(t1 - t0).count();
        t0 = t1;
    };
    const int n = (int) cur->kps.size();
    const FlatHash<FlatNoValue> &order = cur->kps.ids;
    // ---- the carried table: nothing but removals since the previous tracking step => per slot, the slot it was (slam.hpp) ----------------
    uint16_t *carry = nullptr;
    // (only on the path that solves the pose behind the tracker with P3P: the composed step of the other configurations reads the table)
    if (!slots_dirty_ && carry_frame_ == cur.get() && carry_table_edits_ == cur->table_edits && carry_mp_edits_ == mp_edits_ && n > 0 &&
        n <= carry_n_prev_ && ready_for_init && (p3p_req || cfg.p3p_enabled))
        carry = st->track_carry_buffer(carry_n_prev_, n);
    float *jpx = nullptr;
    uint8_t *j3d = nullptr;
    double *jw = nullptr;
    const bool assemble = !carry || check_carry_;
    if (carry) {
        job_ids_prev_.swap(job_ids_);
        job_is3d_prev_.swap(job_is3d_);
        job_slots_prev_.swap(job_slots_);
    }
    job_ids_.resize((size_t) n + 1);     // (+ 1: the compaction below stores before it knows whether the row stays)
    job_is3d_.resize((size_t) n + 1);
    job_slots_.resize((size_t) n + 1);
    if (carry) {
        // The table's iteration order is the previous frame's with the erased slots taken out (an erase unlinks one node, nothing moves), so
        // the previous frame's slot LIST is walked -- an array, no pointer chain through the table -- and the rows whose slot is still
        // live are kept, without a branch.
        const int np = carry_n_prev_;
        if (check_carry_) chk_carry_.resize((size_t) n + 1);
        const int *pid = job_ids_prev_.data(), *psl = job_slots_prev_.data();
        const uint8_t *p3 = job_is3d_prev_.data();
        int *oid = job_ids_.data(), *osl = job_slots_.data();
        uint8_t *o3 = job_is3d_.data();
        int i = 0;
        for (int ip = 0; ip < np; ip++) {
            const int sl = psl[(size_t) ip];
            if (i > n) break;   // (more live slots than keypoints: cannot happen; caught below)
            carry[(size_t) i] = (uint16_t) ip;   // (i <= n < the buffer's capacity, track_carry_buffer; device memory behind the bus: written, never read)
            if (check_carry_) chk_carry_[(size_t) i] = (uint16_t) ip;
            osl[(size_t) i] = sl;
            oid[(size_t) i] = pid[(size_t) ip];
            o3[(size_t) i] = p3[(size_t) ip];
            i += order.slot_live((size_t) sl) ? 1 : 0;
        }
        if (i != n) {
            std::fprintf(stderr, "alva_slam: carried slot table: %d live slots of the previous frame's %d, %d keypoints\n", i, np, n);
            std::abort();
        }
        t_fine[29] += 1.;   // #frames with a carried slot table
    }
    if (assemble) {
        // (ALVA_CHECK_CARRY=1: the table in host memory -- compared and kept below, copied to the implementation's buffers when it is the one
        // that counts; the implementation's buffers are device memory behind the bus: never read)
        if (check_carry_ || !st->track_slot_buffers(n, &jpx, &j3d, &jw)) {
            job_px_.resize((size_t) n * 2);
            job_stage3d_.resize((size_t) n);
            job_wpt_.resize((size_t) n * 3);
            jpx = job_px_.data();
            j3d = job_stage3d_.data();
            jw = job_wpt_.data();
        }
        int i = 0;
        for (int sl = order.first(); sl != FlatHash<FlatNoValue>::END; sl = order.next(sl)) {
            const KeyPt &k = cur->kps.kp[(size_t) sl];
            const uint8_t is3d = order.tag(sl) != 0;   // the table's tag IS the 3-D flag (KpTable): the flag inside the 80-byte record sits on its second cache line
            if (carry) {
                if (job_slots_[(size_t) i] != sl || job_ids_[(size_t) i] != k.id || job_is3d_[(size_t) i] != is3d) {
                    std::fprintf(stderr, "alva_slam: carried slot table out of sync (slot %d: id %d / %d, 3-D %d / %d)\n", i, job_ids_[(size_t) i], k.id,
                                 (int) job_is3d_[(size_t) i], (int) is3d);
                    std::abort();
                }
            } else {
                job_slots_[(size_t) i] = sl;   // the table slot of tracking slot i: the results are written back without a look-up by id
                job_ids_[(size_t) i] = k.id;
                job_is3d_[(size_t) i] = is3d;
            }
            jpx[2 * (size_t) i] = k.px[0];
            jpx[2 * (size_t) i + 1] = k.px[1];
            j3d[(size_t) i] = is3d;
            i++;
        }
        // the 3-D keypoints' world points: one map point per keypoint, each behind a pointer table -- a second pass so that the objects of
        // the next iterations can be requested ahead (this loop runs with the GPU idle, at the head of the frame's critical path)
        for (int s = 0; s < n; s++) {
            if (s + 16 < n && job_is3d_[(size_t) s + 16]) {
                const MpRec *f = rec_raw(job_ids_[(size_t) s + 16]);
                if (f) __builtin_prefetch(f);
            }
            double *w = jw + 3 * (size_t) s;
            if (job_is3d_[(size_t) s]) {
                const MpRec *mp = rec_raw(job_ids_[(size_t) s]);
                if (!mp) throw std::out_of_range("map point");   // mapMapPoints_.at() throws in the reference (:131) if the map lost it
                std::memcpy(w, mp->X, 24);
            } else {
                w[0] = w[1] = w[2] = 0.;
            }
        }
        if (!carry) t_fine[30] += 1.;   // #frames with an assembled one
    }
    if (check_carry_) {
        // the carried table is: positions = the previous step's tracked positions (still in its result buffers), flags and world points =
        // the previous table's; compared here against the table assembled from the map, then kept as the next frame's "previous"
        if (carry) {
            const TrackKlt &pr = klt_out_;
            for (int i = 0; i < n; i++) {
                const size_t ip = (size_t) chk_carry_[(size_t) i];
                const bool same = std::memcmp(&jpx[2 * (size_t) i], &pr.px_v[2 * ip], 8) == 0 && j3d[(size_t) i] == chk_is3d_[ip] &&
                                  std::memcmp(&jw[3 * (size_t) i], &chk_wpt_[3 * ip], 24) == 0;
                if (!same) {
                    std::fprintf(stderr, "alva_slam: carried slot table differs from the assembled one (slot %d <- %d)\n", i, (int) ip);
                    std::abort();
                }
            }
        }
        chk_px_.assign(jpx, jpx + 2 * (size_t) n);
        chk_is3d_.assign(j3d, j3d + (size_t) n);
        chk_wpt_.assign(jw, jw + 3 * (size_t) n);
        if (carry) {   // what the stage is handed in carried mode: its own table pointers are not needed
            jpx = nullptr; j3d = nullptr; jw = nullptr;
        } else {
            float *bpx = nullptr;
            uint8_t *b3d = nullptr;
            double *bw = nullptr;
            if (st->track_slot_buffers(n, &bpx, &b3d, &bw)) {
                std::memcpy(bpx, jpx, 8 * (size_t) n);
                std::memcpy(b3d, j3d, (size_t) n);
                std::memcpy(bw, jw, 24 * (size_t) n);
                jpx = bpx; j3d = b3d; jw = bw;
            }
        }
    }
    carry_frame_ = cur.get();
    carry_table_edits_ = cur->table_edits;
    carry_mp_edits_ = mp_edits_;
    carry_n_prev_ = n;
    slots_dirty_ = false;
    TrackJob job;
    job.n = n;
    job.px = jpx;
    job.is3d = j3d;
    job.wpt = jw;
    job.carry = carry;
    std::memcpy(job.Tcw_q, cur->Tcw.q, 32);
    std::memcpy(job.Tcw_t, cur->Tcw.t, 24);
    se3_to_pose7(cur->Twc, job.pose7_pred);
    job.use_prior = cfg.klt_use_prior;
    job.klt_levels = cfg.klt_levels;
    job.want_pose = ready_for_init;
    job.do_p3p = p3p_req || cfg.p3p_enabled;
    job.do_random = cfg.random_sampling;
    TrackKlt &r = klt_out_;
    lap(1);
    if (fail(st->track_begin(job, r))) return;
    lap(2);
    // (nothing is inserted or erased between the gather above and here: the slots are still the keypoints')
    for (int pass = 1; pass <= 3; pass++)
        for (int s = 0; s < n; s++)
            if (r.code_v[(size_t) s] == pass) cur->update_slot(job_slots_[(size_t) s], &r.px_v[2 * (size_t) s], &r.unpx_v[2 * (size_t) s], &r.bv_v[3 * (size_t) s]);
    {
        const long full = cfg.klt_levels + 1;
        long work = 0;
        for (int s = 0; s < n; s++) {
            const int c = r.code_v[(size_t) s];
            const bool prior = job_is3d_[(size_t) s] && cfg.klt_use_prior;
            work += c == 1 ? 2 : c == 2 ? full + 1 : c == 3 ? 1 + full + 1 : (prior ? 1 : full);
        }
        n_klt_kp_levels += work;
        n_klt_slots += n;
    }
    pose_ids_.clear();
    for (int s = 0; s < n; s++) {
        if (!r.code_v[(size_t) s]) remove_obs_from_cur(job_ids_[(size_t) s]);
        else if (job_is3d_[(size_t) s]) pose_ids_.push_back(job_ids_[(size_t) s]);
    }
    lap(3);
    if (r.p3p_req) p3p_req = true;
    pose_do_p3p_ = job.do_p3p != 0 || r.p3p_req != 0;  // p3pReq_ set by this very pass counts (visual_frontend.cpp:272)
}

// VisualFrontend::computePose (visual_frontend.cpp:245-417) on the correspondences the tracking step put together
bool Slam::compute_pose() {
    TrackPose &r = pose_out_;
    {
        Section sec(t_section[4]);
        if (fail(st->track_pose_collect(r))) return false;
    }
    Section sec(t_section[5]);
    if (cur->n_3d < 4) return false;
    if (r.status < 0) return false;
    if (r.status == 0) {  // P3P rejected (:318-330)
        reset_frame();
        return false;
    }
    const size_t n = pose_ids_.size();
    if (pose_do_p3p_) {
        cur->set_Twc(se3_from_pose7(r.pose7_p3p));
        for (size_t k = 0; k < n; k++)
            if (r.p3p_outlier[k]) remove_obs_from_cur(pose_ids_[k]);
    }
    if (r.status == 1) {  // refinement rejected (:383-399)
        if (!pose_do_p3p_) p3p_req = true;
        reset_frame();
        return false;
    }
    cur->set_Twc(se3_from_pose7(r.pose7));
    p3p_req = false;
    for (size_t k = 0; k < n; k++)
        if (r.pnp_outlier[k]) remove_obs_from_cur(pose_ids_[k]);
    return true;
}

void Slam::reset_frame() {  // visual_frontend.cpp:700-714
    const KpTable copy = cur->kps;
    for (const auto &e: copy) remove_obs_from_cur(e.first);
    slots_dirty_ = true;
    cur->table_edits++;
    cur->kps.clear();
    cur->note_inserted();
    cur->grid.clear();
    cur->grid.resize(cur->grid_cells);
    cur->n_kps = cur->n_2d = cur->n_3d = 0;
    cur->n_occupied = 0;
}

// The pairing half of compute_parallax(cur->kfid, true, true) -- which keypoint of the reference keyframe carries the same id, two
// tables of ~80-byte records walked side by side -- needs nothing of the pose: collected here, the keyframe check after the pose solve
// only rotates / projects / sorts contiguous data (the check sits on the frame's critical path: the GPU is idle while it runs).
void Slam::prepare_parallax() {
    par_frame_ = -1;
    auto kit = keyframes.find(cur->kfid);
    if (kit == keyframes.end()) return;
    const FrameRec &kf = *kit->second;
    const FlatHash<FlatNoValue> &cids = cur->kps.ids, &kids = kf.kps.ids;
    par_pairs_.clear();
    ParSoA &S = par_soa_;
    const size_t cap4 = (cur->kps.size() + 3) / 4 * 4 + 4;
    if (S.bx.size() < cap4) {
        S.bx.resize(cap4); S.by.resize(cap4); S.bz.resize(cap4); S.ku.resize(cap4); S.kv.resize(cap4);
    }
    for (int sl = cids.first(); sl != FlatHash<FlatNoValue>::END; sl = cids.next(sl)) {
        const KeyPt &k = cur->kps.kp[(size_t) sl];
        const KeyPt *kk = (size_t) sl < kids.slots() && kids.slot_live((size_t) sl) && kids.key(sl) == k.id ? &kf.kps.kp[(size_t) sl] : kf.find(k.id);
        if (!kk) continue;
        const size_t i = par_pairs_.size();
        S.bx[i] = k.bv[0]; S.by[i] = k.bv[1]; S.bz[i] = k.bv[2];
        S.ku[i] = kk->unpx[0]; S.kv[i] = kk->unpx[1];
        par_pairs_.push_back(ParPair{sl, k.id, {kk->unpx[0], kk->unpx[1]}, {k.bv[0], k.bv[1], k.bv[2]}});
    }
    for (size_t i = par_pairs_.size(); i < (par_pairs_.size() + 3) / 4 * 4; i++) {   // padding lanes: a harmless pair
        S.bx[i] = 0.; S.by[i] = 0.; S.bz[i] = 1.; S.ku[i] = 0.f; S.kv[i] = 0.f;
    }
    par_frame_ = cur->id;
    par_kfid_ = cur->kfid;
}

float Slam::compute_parallax(int kfid, bool unrotate, bool median) {  // visual_frontend.cpp:596-670
    auto kit = keyframes.find(kfid);
    if (kit == keyframes.end()) return 0.f;
    const FrameRec &kf = *kit->second;
    double Rkc[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    if (unrotate) {
        double Rkw[9], Rwc[9];
        quat_to_rot(kf.Tcw.q, Rkw);
        quat_to_rot(cur->Twc.q, Rwc);
        mat3_mul(Rkw, Rwc, Rkc);
    }
    float avg = 0.f;
    int cnt = 0;
    std::vector<uint32_t> &all = parallax_bits_;  // the reference's std::set<float>: distinct values, ascending
    all.clear();
    const FlatHash<FlatNoValue> &cids = cur->kps.ids, &kids = kf.kps.ids;
    for (int sl = cids.first(); sl != FlatHash<FlatNoValue>::END; sl = cids.next(sl)) {
        const KeyPt &k = cur->kps.kp[(size_t) sl];
        // the keyframe's keypoint with the same id.  The frame IS a copy of that keyframe's table that has only lost elements since
        // (and gained none: new keypoints come with a new keyframe), so the id usually sits in the SAME slot of the keyframe's table:
        // checked first, the hash look-up only when an edit of the keyframe (a merge) moved it
        const KeyPt *kk = (size_t) sl < kids.slots() && kids.slot_live((size_t) sl) && kids.key(sl) == k.id ? &kf.kps.kp[(size_t) sl] : kf.find(k.id);
        if (!kk) continue;
        float un[2] = {k.unpx[0], k.unpx[1]};
        if (unrotate) {
            double r[3];
            mat3_vec(Rkc, k.bv, r);
            kf.project_cam_to_image(r, un);
        }
        const float dx = un[0] - kk->unpx[0], dy = un[1] - kk->unpx[1];
        const float par = (float) std::sqrt((double) dx * dx + (double) dy * dy);  // cv::norm(Point2f) -> double, stored in a float
        avg += par;
        cnt++;
        if (median) {
            uint32_t b;
            std::memcpy(&b, &par, 4);
            all.push_back(b);  // non-negative floats order like their bit patterns
        }
    }
    if (!cnt) return 0.f;
    avg /= (float) cnt;
    if (median) avg = median_of_distinct(all);
    return avg;
}

// element size/2 of the SET of values (visual_frontend.cpp:661-666): radix sort, drop duplicates, index.  `all` holds the bit patterns of
// non-negative floats (they order like the floats); it is reordered.
float Slam::median_of_distinct(std::vector<uint32_t> &all) {
    std::vector<uint32_t> &tmp = parallax_tmp_;
    tmp.resize(all.size());
    uint32_t *src = all.data(), *dst = tmp.data();
    const size_t m = all.size();
    // three 11-bit digits; the three histograms in ONE counting pass
    static thread_local uint32_t hist[3][2049];
    std::memset(hist, 0, sizeof(hist));
    for (size_t i = 0; i < m; i++) {
        const uint32_t v = src[i];
        hist[0][(v & 2047u) + 1]++;
        hist[1][((v >> 11) & 2047u) + 1]++;
        hist[2][(v >> 22) + 1]++;
    }
    for (int d = 0; d < 3; d++) {
        uint32_t *h = hist[d];
        if (h[((src[0] >> (11 * d)) & 2047u) + 1] == m) continue;   // every key has the same digit here: the pass would not move anything
        for (int b = 0; b < 2048; b++) h[b + 1] += h[b];
        const int shift = 11 * d;
        for (size_t i = 0; i < m; i++) dst[h[(src[i] >> shift) & 2047u]++] = src[i];
        std::swap(src, dst);
    }
    size_t u = 0;
    for (size_t i = 0; i < m; i++)
        if (i == 0 || src[i] != src[u - 1]) src[u++] = src[i];
    const uint32_t b = src[u / 2];
    float out;
    std::memcpy(&out, &b, 4);
    return out;
}

// compute_parallax(kfid, true, true) on the pairs prepare_parallax collected, minus the keypoints the pose solve has removed since (the
// median is over the SET of values: the order of the walk does not matter).  A function of its own: inside compute_parallax the
// second loop changed the code generated for the first (measured: the unchanged general loop ran 2.2x slower).
// four pairs per instruction where the CPU has AVX2: the same IEEE operations in the same order as the scalar loop below (no contraction:
// the file is built with -ffp-contract=off and the target has no FMA), so the bits are the scalar loop's
#if defined(__x86_64__)
__attribute__((target("avx2"))) static void parallax_block_avx2(const double *Rkc, double fx, double fy, double cx, double cy, const double *bx,
                                                                const double *by, const double *bz, const float *ku, const float *kv, size_t n4,
                                                                uint32_t *bits) {
    const __m256d r0 = _mm256_set1_pd(Rkc[0]), r1 = _mm256_set1_pd(Rkc[1]), r2 = _mm256_set1_pd(Rkc[2]), r3 = _mm256_set1_pd(Rkc[3]),
                  r4 = _mm256_set1_pd(Rkc[4]), r5 = _mm256_set1_pd(Rkc[5]), r6 = _mm256_set1_pd(Rkc[6]), r7 = _mm256_set1_pd(Rkc[7]),
                  r8 = _mm256_set1_pd(Rkc[8]);
    const __m256d vfx = _mm256_set1_pd(fx), vfy = _mm256_set1_pd(fy), vcx = _mm256_set1_pd(cx), vcy = _mm256_set1_pd(cy), one = _mm256_set1_pd(1.0);
    for (size_t i = 0; i < n4; i += 4) {
        const __m256d x = _mm256_loadu_pd(bx + i), y = _mm256_loadu_pd(by + i), z = _mm256_loadu_pd(bz + i);
        const __m256d a = _mm256_add_pd(_mm256_add_pd(_mm256_mul_pd(r0, x), _mm256_mul_pd(r1, y)), _mm256_mul_pd(r2, z));   // mat3_vec (se3.hpp)
        const __m256d b = _mm256_add_pd(_mm256_add_pd(_mm256_mul_pd(r3, x), _mm256_mul_pd(r4, y)), _mm256_mul_pd(r5, z));
        const __m256d c = _mm256_add_pd(_mm256_add_pd(_mm256_mul_pd(r6, x), _mm256_mul_pd(r7, y)), _mm256_mul_pd(r8, z));
        const __m256d iz = _mm256_div_pd(one, c), px = _mm256_mul_pd(a, iz), py = _mm256_mul_pd(b, iz);
        const __m128 ux = _mm256_cvtpd_ps(_mm256_add_pd(_mm256_mul_pd(vfx, px), vcx)), uy = _mm256_cvtpd_ps(_mm256_add_pd(_mm256_mul_pd(vfy, py), vcy));
        const __m128 dx = _mm_sub_ps(ux, _mm_loadu_ps(ku + i)), dy = _mm_sub_ps(uy, _mm_loadu_ps(kv + i));
        const __m256d dxd = _mm256_cvtps_pd(dx), dyd = _mm256_cvtps_pd(dy);
        const __m128 par = _mm256_cvtpd_ps(_mm256_sqrt_pd(_mm256_add_pd(_mm256_mul_pd(dxd, dxd), _mm256_mul_pd(dyd, dyd))));
        _mm_storeu_si128(reinterpret_cast<__m128i *>(bits + i), _mm_castps_si128(par));
    }
}
#endif

float Slam::parallax_of_pairs(const FrameRec &kf) {
    double Rkw[9], Rwc[9], Rkc[9];
    quat_to_rot(kf.Tcw.q, Rkw);
    quat_to_rot(cur->Twc.q, Rwc);
    mat3_mul(Rkw, Rwc, Rkc);
    const FlatHash<FlatNoValue> &cids = cur->kps.ids;
    const double fx = cam.fx, fy = cam.fy, cx = cam.cx, cy = cam.cy;
    const size_t n = par_pairs_.size();
    ParSoA &S = par_soa_;
    if (n == 0) return 0.f;
    S.bits.resize((n + 3) / 4 * 4);
    bool vec = false;
#if defined(__x86_64__)
    static const bool have_avx2 = __builtin_cpu_supports("avx2") && !std::getenv("ALVA_NO_AVX2");
    if (have_avx2 && S.bx.size() >= (n + 3) / 4 * 4) {
        parallax_block_avx2(Rkc, fx, fy, cx, cy, S.bx.data(), S.by.data(), S.bz.data(), S.ku.data(), S.kv.data(), (n + 3) / 4 * 4, S.bits.data());
        vec = true;
    }
#endif
    if (!vec || check_obs_mirror_) {
        for (size_t i = 0; i < n; i++) {
            const ParPair &pp = par_pairs_[i];
            double r[3];
            mat3_vec(Rkc, pp.bv, r);
            // FrameRec::project_cam_to_image (camera_calibration.cpp:25-32)
            const double iz = 1. / r[2], x = r[0] * iz, y = r[1] * iz;
            const float ux = (float) (fx * x + cx), uy = (float) (fy * y + cy);
            const float dx = ux - pp.kf_unpx[0], dy = uy - pp.kf_unpx[1];
            const float par = (float) std::sqrt((double) dx * dx + (double) dy * dy);  // cv::norm(Point2f) -> double, stored in a float
            uint32_t b;
            std::memcpy(&b, &par, 4);
            if (vec && b != S.bits[i]) {   // (ALVA_CHECK_OBS_MIRROR=1: the four-wide loop against the scalar one, every pair of every frame)
                std::fprintf(stderr, "alva_slam: vectorised parallax differs from the scalar loop (pair %zu: %08x vs %08x)\n", i, S.bits[i], b);
                std::abort();
            }
            S.bits[i] = b;
        }
    }
    // The caller only COMPARES the median with minAvgRotationParallax and with half of it (visual_frontend.cpp:586-593), and the median is
    // element size / 2 of the SET of values (:661-666).  Both tests are monotone in the value, so with L = the number of DISTINCT values
    // that fail a test, the median passes it exactly when size / 2 >= L: counting replaces the sort (round 4: a three-pass radix sort of
    // ~2 400 values on every frame, behind the pose, the GPU idle).  Duplicates are rare but the reference's std::set drops them, so:
    //   pass 1 counts values and failures WITH duplicates and, in a 64 K-bit table of hashed bit patterns, how many values found their bit
    //          set (c >= the number of duplicates).  With d duplicates, dL of them among the failing values (0 <= dL <= d <= c), the test is
    //          (n - d) / 2 - (Ln - dL) >= 0; it lies between (n - c) / 2 - Ln and n / 2 + c - Ln -- same sign at both ends: decided;
    //   pass 2 (the median's rank within ~2 % of a threshold) counts distinct values exactly in an open-addressed table of the bit
    //          patterns, stamped with a generation instead of being cleared.
    // Pairs whose keypoint the pose solve has removed since prepare_parallax are skipped.  Returned: a value of the set that passes the same
    // tests as the median -- it stands for the median in those two comparisons only.
    const double t_half = (double) cfg.min_avg_rot_parallax / 2., t_full = (double) cfg.min_avg_rot_parallax;   // (as the caller writes them)
    std::vector<uint32_t> &all = parallax_bits_;   // the live pairs' values
    all.resize(n);
    size_t cnt = 0, coll = 0, fail_half = 0, fail_full = 0;
    uint32_t lo = 0xffffffffu, hi = 0, lo_pass_half = 0xffffffffu;   // (non-negative floats order like their bit patterns)
    {
        uint64_t *bm = par_bitmap_;
        std::memset(bm, 0, sizeof(par_bitmap_));
        uint32_t *out = all.data();
        for (size_t i = 0; i < n; i++) {
            const ParPair &pp = par_pairs_[i];
            if ((size_t) pp.slot >= cids.slots() || !cids.slot_live((size_t) pp.slot) || cids.key(pp.slot) != pp.id) continue;
            const uint32_t b = S.bits[i];
            out[cnt++] = b;
            const uint32_t h = (b * 0x9E3779B1u) >> 16;
            const uint64_t bit = 1ull << (h & 63);
            uint64_t &w = bm[h >> 6];
            coll += (w & bit) != 0;
            w |= bit;
            float v;
            std::memcpy(&v, &b, 4);
            const bool p_half = (double) v >= t_half, p_full = (double) v >= t_full;
            fail_half += !p_half;
            fail_full += !p_full;
            lo = b < lo ? b : lo;
            hi = b > hi ? b : hi;
            const uint32_t cand = p_half ? b : 0xffffffffu;
            lo_pass_half = cand < lo_pass_half ? cand : lo_pass_half;
        }
    }
    all.resize(cnt);
    if (!cnt) return 0.f;
    auto decide = [&](size_t failing) -> int {   // 1 = the median passes, 0 = it fails, -1 = depends on the duplicates
        const long f_min = (long) ((cnt - coll) / 2) - (long) failing, f_max = (long) (cnt / 2) + (long) coll - (long) failing;
        return f_min >= 0 ? 1 : f_max < 0 ? 0 : -1;
    };
    int d_half = decide(fail_half), d_full = decide(fail_full);
    if (d_half < 0 || d_full < 0 || check_obs_mirror_) {
        size_t cap = 4096;
        while (cap < 2 * cnt) cap *= 2;   // at most half full
        if (par_seen_.size() != cap) {
            par_seen_.assign(cap, 0);
            par_gen_ = 0;
        }
        if (++par_gen_ == 0) {   // the stamp wrapped: old entries could look current
            std::fill(par_seen_.begin(), par_seen_.end(), 0);
            par_gen_ = 1;
        }
        const uint64_t gen = (uint64_t) par_gen_ << 32;
        const size_t mask = cap - 1;
        int shift = 32;
        for (size_t c = cap; c > 1; c >>= 1) shift--;
        uint64_t *seen = par_seen_.data();
        size_t distinct = 0, dfail_half = 0, dfail_full = 0;
        for (size_t i = 0; i < cnt; i++) {
            const uint32_t b = all[i];
            const uint64_t want = gen | b;
            size_t h = (size_t) ((b * 0x9E3779B1u) >> shift);
            bool dup = false;
            while ((seen[h] >> 32) == (uint64_t) par_gen_) {
                if (seen[h] == want) {
                    dup = true;
                    break;
                }
                h = (h + 1) & mask;
            }
            if (dup) continue;
            seen[h] = want;
            float v;
            std::memcpy(&v, &b, 4);
            distinct++;
            dfail_half += !((double) v >= t_half);
            dfail_full += !((double) v >= t_full);
        }
        const int e_half = distinct / 2 >= dfail_half ? 1 : 0, e_full = distinct / 2 >= dfail_full ? 1 : 0;
        if ((d_half >= 0 && d_half != e_half) || (d_full >= 0 && d_full != e_full) || cnt - distinct > coll) {   // (only reachable in check mode)
            std::fprintf(stderr, "alva_slam: the bounded median test differs from the exact count (%d %d vs %d %d; %zu duplicates, bound %zu)\n", d_half,
                         d_full, e_half, e_full, cnt - distinct, coll);
            std::abort();
        }
        d_half = e_half;
        d_full = e_full;
    }
    const bool med_half = d_half != 0, med_full = d_full != 0;
    // a member of the set with the median's two answers: the largest value passes whatever the median passes; the smallest fails whatever
    // it fails; "passes half, fails full": the smallest value that passes half is <= the median, so it fails full as well
    uint32_t pick = med_half && med_full ? hi : !med_half && !med_full ? lo : med_half ? lo_pass_half : hi;
    float out;
    std::memcpy(&out, &pick, 4);
    if (med_full && !med_half) return median_of_distinct(all);   // (thresholds the other way round: not a configuration of the reference)
    if (check_obs_mirror_) {   // the counting against the sort, every frame of the long CPU differentials
        const float med = median_of_distinct(all);
        if (((double) med >= t_half) != ((double) out >= t_half) || ((double) med >= t_full) != ((double) out >= t_full)) {
            std::fprintf(stderr, "alva_slam: the counted median test differs from the sorted one (median %.9g, stand-in %.9g)\n", (double) med, (double) out);
            std::abort();
        }
    }
    return out;
}

bool Slam::check_ready_for_init() {  // visual_frontend.cpp:419-551
    const double med = compute_parallax(cur->kfid, false, true);
    if (med <= cfg.min_avg_rot_parallax) return false;
    std::shared_ptr<FrameRec> kf = keyframes.at(cur->kfid);
    if (!kf) return false;
    if (cur->n_kps < 8) return false;
    std::vector<int> ids;
    std::vector<double> bv_kf, bv_cur;
    double Rck[9], Rkw[9], Rwc[9];
    quat_to_rot(kf->Tcw.q, Rkw);
    quat_to_rot(cur->Twc.q, Rwc);
    mat3_mul(Rkw, Rwc, Rck);
    int cnt = 0;
    float avg = 0.f;
    for (const auto &e: cur->kps) {
        const KeyPt &k = e.second;
        const KeyPt *kk = kf->find(k.id);
        if (!kk) continue;
        bv_kf.insert(bv_kf.end(), kk->bv, kk->bv + 3);
        bv_cur.insert(bv_cur.end(), k.bv, k.bv + 3);
        ids.push_back(k.id);
        double r[3], u[3];
        mat3_vec(Rck, k.bv, r);
        const double K[9] = {cam.fx, 0, cam.cx, 0, cam.fy, cam.cy, 0, 0, 1};
        mat3_vec(K, r, u);
        const float rx = (float) (u[0] / u[2]), ry = (float) (u[1] / u[2]);
        const float dx = rx - kk->unpx[0], dy = ry - kk->unpx[1];
        avg = (float) ((double) avg + std::sqrt((double) dx * dx + (double) dy * dy));  // float += double (visual_frontend.cpp:487)
        cnt++;
    }
    if (cnt < 8) return false;
    avg /= (float) cnt;
    if (avg < cfg.min_avg_rot_parallax) return false;
    const int n = (int) ids.size();
    double R[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, t[3] = {0, 0, 0};
    std::vector<int> outliers((size_t) n + 1);
    int n_out = 0, ok = 0;
    if (fail(st->five_point(n, bv_kf.data(), bv_cur.data(), cfg.random_sampling ? 1 : 0, R, t, outliers.data(), &n_out, &ok))) return false;
    if (!ok) return false;
    for (int i = 0; i < n_out; i++) remove_obs_from_cur(ids[(size_t) outliers[(size_t) i]]);
    const double tn = std::sqrt(t[0] * t[0] + t[1] * t[1] + t[2] * t[2]);
    SE3 T;
    rot_to_quat(R, T.q);
    for (int i = 0; i < 3; i++) T.t[i] = t[i] / tn;
    init_computed = T;
    // test hook: continue from a given two-view pose instead (the five-point refinement sits at a noise floor)
    if (init_override.armed) T = se3_from_pose7(init_override.pose7);
    cur->set_Twc(T);
    return true;
}

bool Slam::check_new_keyframe_required() {  // visual_frontend.cpp:554-594
    auto kit = keyframes.find(cur->kfid);
    if (kit == keyframes.end()) return false;
    const FrameRec &kf = *kit->second;
    const auto t_par0 = std::chrono::steady_clock::now();
    const double med = par_frame_ == cur->id && par_kfid_ == kf.kfid ? parallax_of_pairs(kf) : compute_parallax(kf.kfid, true, true);
    t_fine[18] += std::chrono::duration<double>(std::chrono::steady_clock::now() - t_par0).count();   // (per FRAME, not per keyframe)
    const int id_diff = cur->id - kf.id;
    if (id_diff >= 5 && cur->n_occupied < 0.33 * cfg.max_keypoints) return true;
    if (id_diff >= 2 && cur->n_3d < 20) return true;
    if (id_diff < 2 && cur->n_3d > 0.5 * cfg.max_keypoints) return false;
    const bool cx = med >= cfg.min_avg_rot_parallax / 2.;
    const bool c0 = med >= cfg.min_avg_rot_parallax;
    const bool c1 = cur->n_3d < 0.75 * kf.n_3d;
    const bool c2 = cur->n_occupied < 0.5 * cfg.max_keypoints && cur->n_3d < 0.85 * kf.n_3d;
    return (c0 || c1 || c2) && cx;
}

}  // namespace alva_slam
