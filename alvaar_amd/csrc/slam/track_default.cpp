// The tracking step composed from the fine-grained stages, in the reference's order (VisualFrontend::kltTrackingFromMotionPrior,
// visual_frontend.cpp:103-243, and computePose, :245-417).  This is what a `Stages` implementation gets when it does not override
// track_begin / track_pose_collect; the HIP implementation overrides both with one device-side chain and is tested against this
// composition's semantics through the reference itself.
#include "se3.hpp"
#include "stages.hpp"
#include <cmath>
#include <cstring>

namespace alva_slam {

int Stages::track_begin(const TrackJob &job, TrackKlt &out) {
    const int n = job.n;
    out.code.assign((size_t) n, 0);
    out.px.assign((size_t) n * 2, 0.f);
    out.unpx.assign((size_t) n * 2, 0.f);
    out.bv.assign((size_t) n * 3, 0.);
    out.p3p_req = 0;
    out.n_pose = 0;
    pending_.active = false;
    // projections of the 3-D slots' map points with the predicted pose (:125-140)
    std::vector<int> cand;
    std::vector<double> cam_pts;
    SE3 Tcw;
    std::memcpy(Tcw.q, job.Tcw_q, 32);
    std::memcpy(Tcw.t, job.Tcw_t, 24);
    if (job.use_prior)
        for (int i = 0; i < n; i++)
            if (job.is3d[i]) {
                double pc[3];
                se3_apply(Tcw, job.wpt + 3 * (size_t) i, pc);
                cand.push_back(i);
                cam_pts.insert(cam_pts.end(), pc, pc + 3);
            }
    std::vector<float> proj(cand.size() * 2);
    int rc = cand.empty() ? 0 : project_dist((int) cand.size(), cam_pts.data(), proj.data());
    if (rc) return rc;
    std::vector<int> slotA, slotB;
    std::vector<float> ptsA, priorA, ptsB, priorB;
    size_t ci = 0;
    for (int i = 0; i < n; i++) {
        const float *p = job.px + 2 * (size_t) i;
        if (job.use_prior && job.is3d[i]) {
            const float *q = &proj[2 * ci++];
            // Frame::isInImage (frame.cpp:462-465); width / height are read through the projection's own camera: the caller's frame
            if (q[0] >= 0 && q[1] >= 0 && q[0] < (float) image_width_ && q[1] < (float) image_height_) {
                slotA.push_back(i);
                ptsA.insert(ptsA.end(), p, p + 2);
                priorA.insert(priorA.end(), q, q + 2);
                continue;
            }
        }
        slotB.push_back(i);
        ptsB.insert(ptsB.end(), p, p + 2);
        priorB.insert(priorB.end(), p, p + 2);
    }
    const size_t nB0 = slotB.size();
    if (job.use_prior && !slotA.empty()) {  // 1st pass: 3-D keypoints from their priors on ONE level (:155-203)
        const int na = (int) slotA.size();
        std::vector<uint8_t> ok((size_t) na);
        rc = fbklt(1, na, ptsA.data(), priorA.data(), ok.data());
        if (rc) return rc;
        size_t good = 0;
        for (int k = 0; k < na; k++) {
            if (ok[(size_t) k]) {
                out.code[(size_t) slotA[(size_t) k]] = 1;
                out.px[2 * (size_t) slotA[(size_t) k]] = priorA[2 * (size_t) k];
                out.px[2 * (size_t) slotA[(size_t) k] + 1] = priorA[2 * (size_t) k + 1];
                good++;
            } else {
                slotB.push_back(slotA[(size_t) k]);
                ptsB.insert(ptsB.end(), &ptsA[2 * (size_t) k], &ptsA[2 * (size_t) k] + 2);
                priorB.insert(priorB.end(), &priorA[2 * (size_t) k], &priorA[2 * (size_t) k] + 2);
            }
        }
        if (good < 0.33 * na) {
            out.p3p_req = 1;
            priorB = ptsB;
        }
    }
    if (!slotB.empty()) {  // 2nd pass: everything else on the full pyramid (:205-242)
        const int nb = (int) slotB.size();
        std::vector<uint8_t> ok((size_t) nb);
        rc = fbklt(job.klt_levels, nb, ptsB.data(), priorB.data(), ok.data());
        if (rc) return rc;
        for (int k = 0; k < nb; k++)
            if (ok[(size_t) k]) {
                out.code[(size_t) slotB[(size_t) k]] = (size_t) k < nB0 ? 2 : 3;
                out.px[2 * (size_t) slotB[(size_t) k]] = priorB[2 * (size_t) k];
                out.px[2 * (size_t) slotB[(size_t) k] + 1] = priorB[2 * (size_t) k + 1];
            }
    }
    // Frame::updateKeypoint -> computeKeypoint for every tracked slot
    std::vector<int> upd;
    std::vector<float> upx;
    for (int i = 0; i < n; i++)
        if (out.code[(size_t) i]) {
            upd.push_back(i);
            upx.push_back(out.px[2 * (size_t) i]);
            upx.push_back(out.px[2 * (size_t) i + 1]);
        }
    if (!upd.empty()) {
        std::vector<float> un(upd.size() * 2);
        std::vector<double> bv(upd.size() * 3);
        rc = compute_keypoints((int) upd.size(), upx.data(), un.data(), bv.data());
        if (rc) return rc;
        for (size_t k = 0; k < upd.size(); k++) {
            std::memcpy(&out.unpx[2 * (size_t) upd[k]], &un[2 * k], 8);
            std::memcpy(&out.bv[3 * (size_t) upd[k]], &bv[3 * k], 24);
        }
    }
    PendingPose &P = pending_;
    P.bv.clear();
    P.uv.clear();
    P.wpt.clear();
    for (int i = 0; i < n; i++)
        if (job.is3d[i] && out.code[(size_t) i]) {
            P.bv.insert(P.bv.end(), &out.bv[3 * (size_t) i], &out.bv[3 * (size_t) i] + 3);
            P.uv.push_back((double) out.unpx[2 * (size_t) i]);
            P.uv.push_back((double) out.unpx[2 * (size_t) i + 1]);
            P.wpt.insert(P.wpt.end(), job.wpt + 3 * (size_t) i, job.wpt + 3 * (size_t) i + 3);
            out.n_pose++;
        }
    P.n = out.n_pose;
    P.active = job.want_pose != 0;
    P.do_p3p = job.do_p3p || out.p3p_req;
    P.do_random = job.do_random;
    std::memcpy(P.pose7, job.pose7_pred, sizeof(P.pose7));
    out.code_v = out.code.data();
    out.px_v = out.px.data();
    out.unpx_v = out.unpx.data();
    out.bv_v = out.bv.data();
    return 0;
}

int Stages::track_pose_collect(TrackPose &out) {  // visual_frontend.cpp:245-417
    PendingPose &P = pending_;
    out.status = -1;
    if (!P.active) return 0;
    P.active = false;
    int n = P.n;
    out.p3p_outlier.assign((size_t) n, 0);
    out.pnp_outlier.assign((size_t) n, 0);
    if (n < 4) return 0;
    double pose7[7];
    std::memcpy(pose7, P.pose7, sizeof(pose7));
    std::vector<int> outliers((size_t) n + 1), index((size_t) n);
    for (int i = 0; i < n; i++) index[(size_t) i] = i;
    int n_out = 0, ok = 0;
    std::vector<double> uv = P.uv, wpt = P.wpt;
    if (P.do_p3p) {
        int rc = p3p(n, P.bv.data(), P.wpt.data(), P.do_random, pose7, outliers.data(), &n_out, &ok);
        if (rc) return rc;
        const size_t inliers = (size_t) n - (size_t) (ok ? n_out : 0);
        bool bad_t = false;
        for (int i = 0; i < 3; i++) bad_t = bad_t || std::isinf(pose7[i]) || std::isnan(pose7[i]);
        if (!ok || inliers < 5 || bad_t) {
            out.status = 0;
            return 0;
        }
        std::memcpy(out.pose7_p3p, pose7, sizeof(pose7));
        for (int i = 0; i < n_out; i++) out.p3p_outlier[(size_t) outliers[(size_t) i]] = 1;
        int w = 0;
        for (int i = 0; i < n; i++)
            if (!out.p3p_outlier[(size_t) i]) {
                index[(size_t) w] = i;
                uv[2 * (size_t) w] = uv[2 * (size_t) i]; uv[2 * (size_t) w + 1] = uv[2 * (size_t) i + 1];
                for (int c = 0; c < 3; c++) wpt[3 * (size_t) w + c] = wpt[3 * (size_t) i + c];
                w++;
            }
        n = w;
    } else {
        std::memcpy(out.pose7_p3p, pose7, sizeof(pose7));
    }
    n_out = 0;
    ok = 0;
    int rc = pnp(n, uv.data(), wpt.data(), pose7, outliers.data(), &n_out, &ok);
    if (rc) return rc;
    const size_t inliers = (size_t) n - (size_t) n_out;
    bool bad_t = false;
    for (int i = 0; i < 3; i++) bad_t = bad_t || std::isinf(pose7[i]) || std::isnan(pose7[i]);
    if (!ok || inliers < 5 || n_out > 0.5 * n || bad_t) {
        out.status = 1;
        return 0;
    }
    std::memcpy(out.pose7, pose7, sizeof(pose7));
    for (int i = 0; i < n_out; i++) out.pnp_outlier[(size_t) index[(size_t) outliers[(size_t) i]]] = 1;
    out.status = 2;
    return 0;
}

}  // namespace alva_slam

namespace alva_slam {

// Mapper::matchToMap on the records, default: the flat map of match_to_map built from the rows' records (the HIP stages do not run this)
int Stages::match_to_map_rec(const MatchJob &J, int *match_of_mp) {
    std::vector<int> kf_index;
    int max_kf = -1;
    for (int i = 0; i < J.n_kf; i++) max_kf = J.kf_ids[i] > max_kf ? J.kf_ids[i] : max_kf;
    kf_index.assign((size_t) max_kf + 2, -1);
    for (int i = 0; i < J.n_kf; i++) kf_index[(size_t) J.kf_ids[i]] = i;
    if (J.frame_kfid < 0 || J.frame_kfid > max_kf || kf_index[(size_t) J.frame_kfid] < 0) return -1;
    std::vector<double> wpt((size_t) J.n_mp * 3);
    std::vector<uint8_t> is3d((size_t) J.n_mp), has_desc((size_t) J.n_mp), obs_hd, obs_desc;
    std::vector<int> obs_ptr((size_t) J.n_mp + 1), obs_kf;
    std::vector<float> obs_px;
    static const alva_medoid::Table fresh = [] { alva_medoid::Table t{}; alva_medoid::reset(t); return t; }();
    for (int m = 0; m < J.n_mp; m++) {
        const int slot = J.mp_slot[m];
        // (a subclass that keeps its own arena behind mp_arena_chunk must bring its own match_to_map_rec: this one reads arena_)
        if (slot < 0 || ((size_t) slot >> MP_CHUNK_SHIFT) >= arena_.size() || !arena_[(size_t) slot >> MP_CHUNK_SHIFT]) return -1;
        const MpRec &r = arena_[(size_t) slot >> MP_CHUNK_SHIFT][(size_t) (slot & (MP_CHUNK - 1))];
        const alva_medoid::Table &t = (size_t) slot < med_tables_.size() ? med_tables_[(size_t) slot] : fresh;
        std::memcpy(&wpt[3 * (size_t) m], r.X, 24);
        is3d[(size_t) m] = r.is3d;
        has_desc[(size_t) m] = r.has_desc;
        obs_ptr[(size_t) m] = (int) obs_kf.size();
        for (int e = 0; e < r.n_ent; e++) {
            const ObsEnt &en = r.ent[e];
            if (!(en.flags & MPF_OBS) || !(en.flags & MPF_INKF)) continue;
            if (en.kf < 0 || en.kf > max_kf || kf_index[(size_t) en.kf] < 0) continue;
            obs_kf.push_back(kf_index[(size_t) en.kf]);
            obs_px.push_back(en.px[0]);
            obs_px.push_back(en.px[1]);
            // keyframes in which the keypoint could not be described (within 31 px of the border, feature_extractor.cpp:191-209) have no
            // entry in mapKeyframeDescriptors_: slot zeroed and flagged
            const int ds = alva_medoid::find_slot(t, en.kf);
            const size_t at = obs_desc.size();
            obs_desc.resize(at + 32, 0);
            if (ds != alva_medoid::END) std::memcpy(&obs_desc[at], t.slot[ds].desc, 32);
            obs_hd.push_back(ds != alva_medoid::END ? 1 : 0);
        }
    }
    obs_ptr[(size_t) J.n_mp] = (int) obs_kf.size();
    if (obs_kf.empty()) {   // (the arrays must not be null)
        obs_kf.push_back(0); obs_px.resize(2, 0.f); obs_desc.resize(32, 0); obs_hd.push_back(0);
    }
    return match_to_map(J.cell_size, J.num_cells_w, J.grid_cells, J.cell_ptr, J.cell_mp, J.n_kf, J.kf_q, J.kf_t, J.n_mp, wpt.data(), is3d.data(),
                        has_desc.data(), obs_ptr.data(), obs_kf.data(), obs_px.data(), obs_desc.data(), obs_hd.data(), kf_index[(size_t) J.frame_kfid],
                        J.num_keypoints_3d, J.n_local, J.local, J.max_proj_err, J.dist_ratio, match_of_mp);
}

}  // namespace alva_slam
