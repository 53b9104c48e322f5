// f1: Mapper::matchToMap -- re-association of the local map with a new keyframe -- on a flattened map.
//
// Replaces the loops of Mapper::matchToMap (src/slam/src/mapper.cpp:354-588) for a CONSISTENT map (every keypoint id has
// its map point, every observing keyframe exists and holds the keypoint; the reference's repair branches :459-463, :500-509
// stay on the host).  Per local map point (:389-553): project into the frame (Sophus SE3 * point = Eigen quaternion rotation,
// CameraCalibration::projectCamToImageDist), visibility gates, the 2x2-cell keypoint neighbourhood of
// Frame::getSurroundingKeypoints (frame.cpp:313-341), and for each neighbouring keypoint: pixel distance, "never observed in
// the same keyframe", mean re-projection error of the map point in the keypoint's keyframes, minimum Hamming distance
// between the two descriptor sets (MapPoint::computeMinDescDist, map_point.cpp:206-222); best / second best with the 0.9
// ratio test.  Then per keypoint the closest map point wins, the last one on ties (:555-586).
//
//   k_frame_obs     map point -> its observation in the frame (or -1)
//   k_match_local   ONE WAVE per local map point; lane i evaluates the i-th neighbouring keypoint; the reference's sequential
//                   two-minimum scan (<=) is reproduced with ballots: best = LAST lane holding the minimum, second = minimum
//                   of the rest
//   k_arbitrate     per keypoint the closest map point, the LAST one in list order on ties: k_match_local folds
//                   (distance, list position) into one 64-bit key per keypoint with atomicMax -- an order-independent
//                   maximum, so the result does not depend on which wave gets there first -- and this kernel decodes it
// Everything is integer / IEEE double / float in the reference's operation order (compile with -ffp-contract=off), so the
// result is the reference's, not an approximation of it.
//
// Round 5: the same call on the map layer's RECORDS (slam/mp_rec.hpp) -- alva_match_to_map_records.  The host no longer flattens its
// map per keyframe (33 000 observations x 45 B assembled + uploaded); it names one record slot per table row and
//   k_mtm_gather    ONE WAVE per row reads the record's header + live entries straight out of pinned host memory (zero-copy; the
//                   only bytes that cross PCIe are the ones the call needs: ~64 + 24 n_entries per row), keeps the observations whose
//                   keyframe exists and holds the keypoint (ballot compaction, ascending keyframe = observedKeyframeIds_ order), finds
//                   the row's observation in the frame, and lists the live slots of the map point's descriptor table (medoid_table.hpp,
//                   resident on the device: mapKeyframeDescriptors_ itself, which is what MapPoint::computeMinDescDist iterates)
//   k_match_rows    k_match_local's body on the gathered rows (descriptors read in place from the tables)
#include "common.hpp"
#include "slam/medoid_table.hpp"
#include "slam/mp_rec.hpp"
#include <cmath>

namespace {

struct MtmArgs {
    double calib[10];  // fx fy cx cy k1 k2 p1 p2 imgW imgH
    int cellSize, numCellsW, gridCells;
    const int *cellPtr, *cellMp;
    const double *kfQ, *kfT;
    int nMp;
    const double *mpWpt;
    const uint8_t *mpIs3d, *mpHasDesc;  // mpHasDesc / obsHasDesc may be NULL: every observation carries a descriptor
    const int *obsPtr, *obsKf;
    const float *obsPx;
    const uint8_t *obsDesc, *obsHasDesc;
    int frameKf, nLocal;
    const int *local;
    float maxPxDist, minDist, viewTh;
    int *frameObs;   // [nMp]
    unsigned long long *arb;  // [nMp] per keypoint: ((~distance bits) << 32) | list position of the best claim, 0 = none
    int *matchOfMp;  // [nMp]
};

__device__ __forceinline__ void se3_apply(const double *q, const double *t, const double *v, double *o) {
    const double uv0 = q[1] * v[2] - q[2] * v[1], uv1 = q[2] * v[0] - q[0] * v[2], uv2 = q[0] * v[1] - q[1] * v[0];
    const double u0 = uv0 + uv0, u1 = uv1 + uv1, u2 = uv2 + uv2;
    const double c0 = q[1] * u2 - q[2] * u1, c1 = q[2] * u0 - q[0] * u2, c2 = q[0] * u1 - q[1] * u0;
    o[0] = ((v[0] + q[3] * u0) + c0) + t[0];
    o[1] = ((v[1] + q[3] * u1) + c1) + t[1];
    o[2] = ((v[2] + q[3] * u2) + c2) + t[2];
}

// CameraCalibration::projectCamToImageDist (camera_calibration.cpp:34-54; see distortion.hip)
__device__ __forceinline__ void project_dist(const double *C, const double *P, float &u, float &v) {
    const double iz = 1. / P[2];
    const float Xf = (float) (P[0] * iz), Yf = (float) (P[1] * iz);
    const double x = (double) Xf, y = (double) Yf;
    const double r2 = x * x + y * y, r4 = r2 * r2, r6 = r4 * r2;
    const double a1 = 2 * x * y, a2 = r2 + 2 * x * x, a3 = r2 + 2 * y * y;
    const double cdist = 1 + C[4] * r2 + C[5] * r4 + 0 * r6;
    const double icdist2 = 1. / (1 + 0 * r2 + 0 * r4 + 0 * r6);
    const double xd0 = x * cdist * icdist2 + C[6] * a1 + C[7] * a2 + 0 * r2 + 0 * r4;
    const double yd0 = y * cdist * icdist2 + C[6] * a3 + C[7] * a1 + 0 * r2 + 0 * r4;
    u = (float) (xd0 * C[0] + C[2]);
    v = (float) (yd0 * C[1] + C[3]);
}

__device__ __forceinline__ float norm2f(float dx, float dy) { return (float) sqrt((double) dx * (double) dx + (double) dy * (double) dy); }

__device__ __forceinline__ int hamming256(const uint4 *a, const uint4 *b) {
    const uint4 a0 = a[0], a1 = a[1], b0 = b[0], b1 = b[1];
    return __popc(a0.x ^ b0.x) + __popc(a0.y ^ b0.y) + __popc(a0.z ^ b0.z) + __popc(a0.w ^ b0.w) + __popc(a1.x ^ b1.x) + __popc(a1.y ^ b1.y) +
           __popc(a1.z ^ b1.z) + __popc(a1.w ^ b1.w);
}

__global__ void __launch_bounds__(256) k_frame_obs(MtmArgs A) {
    const int m = blockIdx.x * 256 + threadIdx.x;
    if (m >= A.nMp) return;
    int fo = -1;
    for (int o = A.obsPtr[m]; o < A.obsPtr[m + 1]; o++)
        if (A.obsKf[o] == A.frameKf) fo = o;
    A.frameObs[m] = fo;
    A.arb[m] = 0ull;
}

__global__ void __launch_bounds__(256) k_match_local(MtmArgs A) {
    const int li = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (li >= A.nLocal) return;
    const int M = A.local[li];
    int outKp = -1;
    float outDist = 0.f;
    // wave-uniform gates (:393-433)
    bool go = A.frameObs[M] < 0 && A.mpIs3d[M] && (A.mpHasDesc ? A.mpHasDesc[M] != 0 : A.obsPtr[M] != A.obsPtr[M + 1]);
    double wpt[3] = {0, 0, 0}, campt[3];
    float pu = 0.f, pv = 0.f;
    if (go) {
#pragma unroll
        for (int k = 0; k < 3; k++) wpt[k] = A.mpWpt[3 * (size_t) M + k];
        se3_apply(A.kfQ + 4 * (size_t) A.frameKf, A.kfT + 3 * (size_t) A.frameKf, wpt, campt);
        go = !(campt[2] < 0.1);
        if (go) {
            const float view_angle = (float) (campt[2] / sqrt((campt[0] * campt[0] + campt[1] * campt[1]) + campt[2] * campt[2]));
            go = !(fabsf(view_angle) < A.viewTh);
        }
        if (go) {
            project_dist(A.calib, campt, pu, pv);
            go = pu >= 0 && pv >= 0 && (double) pu < A.calib[8] && (double) pv < A.calib[9];
        }
    }
    if (go) {
        // the 2 x 2 cells of Frame::getSurroundingKeypoints, in its loop order; their stored keypoints, flattened
        const int rkp = (int) floorf(pv / (float) A.cellSize), ckp = (int) floorf(pu / (float) A.cellSize);
        int cb[4], cn[4], total = 0;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int r = rkp - 1 + (j >> 1), c = ckp - 1 + (j & 1), idx = r * A.numCellsW + c;
            const bool ok = !(r < 0 || c < 0 || idx > A.gridCells) && idx < A.gridCells;
            cb[j] = ok ? A.cellPtr[idx] : 0;
            cn[j] = ok ? A.cellPtr[idx + 1] - cb[j] : 0;
            total += cn[j];
        }
        // running state of the reference's scan: (bestDist, bestId), (secDist, secId valid?)
        float bestD = A.minDist, secD = A.minDist;
        int bestK = -1, nValid = 0;
        for (int base = 0; base < total; base += 64) {
            const int i = base + lane;
            bool valid = false;
            float dist = 0.f;
            int K = -1;
            if (i < total) {
                int j = 0, off = i;
                while (off >= cn[j]) {
                    off -= cn[j];
                    j++;
                }
                K = A.cellMp[cb[j] + off];
                const int ko = A.frameObs[K];
                const float pxDist = norm2f(pu - A.obsPx[2 * (size_t) ko], pv - A.obsPx[2 * (size_t) ko + 1]);
                bool cand = !(pxDist > A.maxPxDist);
                if (A.mpHasDesc && !A.mpHasDesc[K]) cand = false;  // kpMapPoint->desc_.empty() (:465-468)
                const int ka = A.obsPtr[K], kb = A.obsPtr[K + 1], ma = A.obsPtr[M], mb = A.obsPtr[M + 1];
                if (cand)  // never both observed in one keyframe (:474-485)
                    for (int a = ka; a < kb && cand; a++)
                        for (int b = ma; b < mb; b++)
                            if (A.obsKf[a] == A.obsKf[b]) {
                                cand = false;
                                break;
                            }
                if (cand) {  // mean re-projection error of the map point in the keypoint's keyframes (:487-515)
                    float coProj = 0.f;
                    int nCo = 0;
                    for (int a = ka; a < kb; a++) {
                        double cp[3];
                        float qu, qv;
                        se3_apply(A.kfQ + 4 * (size_t) A.obsKf[a], A.kfT + 3 * (size_t) A.obsKf[a], wpt, cp);
                        project_dist(A.calib, cp, qu, qv);
                        const float dx = A.obsPx[2 * (size_t) a] - qu, dy = A.obsPx[2 * (size_t) a + 1] - qv;
                        coProj = (float) ((double) coProj + sqrt((double) dx * (double) dx + (double) dy * (double) dy));
                        nCo++;
                    }
                    cand = !(coProj / (float) nCo > A.maxPxDist);
                }
                if (cand) {  // MapPoint::computeMinDescDist
                    int dmin = 1000;
                    for (int a = ma; a < mb; a++)
                        for (int b = ka; b < kb; b++)
                            if (!A.obsHasDesc || (A.obsHasDesc[a] && A.obsHasDesc[b]))
                                dmin = min(dmin, hamming256(reinterpret_cast<const uint4 *>(A.obsDesc + 32 * (size_t) a),
                                                        reinterpret_cast<const uint4 *>(A.obsDesc + 32 * (size_t) b)));
                    dist = (float) dmin;
                    valid = dist <= A.minDist;  // larger distances fail both `<=` tests of the scan (:519-531)
                }
            }
            // fold this chunk into the scan state: best = LAST minimum, second = minimum of everything else seen
            const unsigned long long vm = __ballot(valid);
            if (vm) {
                float cmin = valid ? dist : 3.0e38f;
#pragma unroll
                for (int off = 32; off > 0; off >>= 1) cmin = fminf(cmin, __shfl_xor(cmin, off));
                const unsigned long long mm = __ballot(valid && dist == cmin);
                const int lastLane = 63 - __clzll((long long) mm);
                float rest = (valid && lane != lastLane) ? dist : 3.0e38f;
#pragma unroll
                for (int off = 32; off > 0; off >>= 1) rest = fminf(rest, __shfl_xor(rest, off));
                const int cK = __shfl(K, lastLane);
                // merge (bestD, secD) [earlier] with (cmin @ cK, rest) [later]
                if (cmin <= bestD) {
                    const float oldBest = nValid > 0 ? bestD : 3.0e38f, oldSec = nValid > 1 ? secD : 3.0e38f;
                    secD = fminf(fminf(oldBest, oldSec), rest);
                    bestD = cmin;
                    bestK = cK;
                } else {
                    const float oldSec = nValid > 1 ? secD : 3.0e38f;
                    secD = fminf(oldSec, cmin);
                }
                nValid += __popcll(vm);
            }
        }
        if (bestK != -1 && nValid > 1)
            if (0.9 * (double) secD < (double) bestD) bestK = -1;  // :534-540
        outKp = bestK;
        outDist = bestD;
    }
    if (lane == 0 && outKp >= 0) {
        // smaller distance wins, then the later list position (the reference scans the claims in push order with <=, :563-578)
        const unsigned long long key = ((unsigned long long) (0xffffffffu - __float_as_uint(outDist)) << 32) | (unsigned) (li + 1);
        atomicMax(&A.arb[outKp], key);
    }
}

__global__ void __launch_bounds__(256) k_arbitrate(MtmArgs A) {
    const int K = blockIdx.x * 256 + threadIdx.x;
    if (K >= A.nMp) return;
    const unsigned li1 = (unsigned) (A.arb[K] & 0xffffffffull);
    A.matchOfMp[K] = li1 ? A.local[li1 - 1] : -1;
}


// ---------------------------------------------------------------------------------------------- the record form (round 5)
using alva_slam::MpRec;
using alva_slam::ObsEnt;
using alva_slam::MP_ENT_CAP;

struct MtmRow {
    double X[3];
    int slot, n_obs, frame_obs, n_desc;
    uint8_t is3d, has_desc, pad[6];
    struct Obs {
        int kf;            // index into the keyframe table
        float px[2];
    } obs[MP_ENT_CAP];
    uint8_t desc_slot[alva_medoid::CAP];   // live slots of the descriptor table
};
static_assert(sizeof(MtmRow) == 576, "MtmRow layout");

struct MtmRecArgs {
    double calib[10];
    int cellSize, numCellsW, gridCells;
    const int *cellPtr, *cellMp;
    int nKf;
    const int *kfIds;
    const double *kfQ, *kfT;
    int nMp;
    const int *mpSlot;
    const MpRec *const *chunks;          // arena chunk table (device-readable; the chunks are pinned host memory)
    const alva_medoid::Table *tables;
    int frameKfId, frameKf, nLocal;
    const int *local;
    float maxPxDist, minDist, viewTh;
    MtmRow *rows;
    unsigned long long *arb;
    int *matchOfMp;
};

__global__ void __launch_bounds__(256) k_mtm_gather(MtmRecArgs A) {
    __shared__ int s_kf[64];
    if ((int) threadIdx.x < A.nKf && threadIdx.x < 64) s_kf[threadIdx.x] = A.kfIds[threadIdx.x];
    __syncthreads();
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= A.nMp) return;
    const int slot = A.mpSlot[row];
    const MpRec *r = A.chunks[slot >> alva_slam::MP_CHUNK_SHIFT] + (slot & (alva_slam::MP_CHUNK - 1));
    // header: 6 x 8 bytes (X, inverse depth, id | anchor, flags)
    const unsigned long long *h8 = reinterpret_cast<const unsigned long long *>(r);
    const unsigned long long hv = lane < 6 ? __builtin_nontemporal_load(h8 + lane) : 0ull;
    const unsigned long long fl = __shfl(hv, 5);
    const int n_ent = (int) ((fl >> 24) & 0xffu);
    MtmRow *g = A.rows + row;
    // entries: kf | flags, px -- the first 16 of an entry's 24 bytes
    int kf = -1, kfi = -1;
    unsigned flags = 0;
    float px0 = 0.f, px1 = 0.f;
    if (lane < n_ent && lane < MP_ENT_CAP) {
        const unsigned long long *e8 = reinterpret_cast<const unsigned long long *>(r->ent + lane);
        const unsigned long long a = __builtin_nontemporal_load(e8), b = __builtin_nontemporal_load(e8 + 1);
        kf = (int) (unsigned) (a & 0xffffffffull);
        flags = (unsigned) ((a >> 32) & 0xffu);
        px0 = __uint_as_float((unsigned) (b & 0xffffffffull));
        px1 = __uint_as_float((unsigned) (b >> 32));
        for (int i = 0; i < A.nKf && i < 64; i++)
            if (s_kf[i] == kf) kfi = i;
    }
    const bool keep = (flags & alva_slam::MPF_OBS) && (flags & alva_slam::MPF_INKF) && kfi >= 0;
    const unsigned long long km = __ballot(keep);
    const int pos = __popcll(km & ((1ull << lane) - 1ull));
    if (keep) {
        g->obs[pos].kf = kfi;
        g->obs[pos].px[0] = px0;
        g->obs[pos].px[1] = px1;
    }
    const unsigned long long fm = __ballot(keep && kf == A.frameKfId);
    // the descriptor table's live slots
    const alva_medoid::Table *t = A.tables + slot;
    const int used = t->used;
    const bool live = lane < used && lane < alva_medoid::CAP && t->slot[lane].key != alva_medoid::FREE_KEY;
    const unsigned long long lm = __ballot(live);
    if (live) g->desc_slot[__popcll(lm & ((1ull << lane) - 1ull))] = (uint8_t) lane;
    if (lane < 3) g->X[lane] = __longlong_as_double((long long) hv);
    if (lane == 0) {
        g->slot = slot;
        g->n_obs = __popcll(km);
        g->frame_obs = fm ? __popcll(km & ((1ull << (63 - __clzll((long long) fm))) - 1ull)) : -1;
        g->n_desc = __popcll(lm);
        g->is3d = (uint8_t) (fl & 0xffu);
        g->has_desc = (uint8_t) ((fl >> 8) & 0xffu);
        A.arb[row] = 0ull;
    }
}

__global__ void __launch_bounds__(256) k_match_rows(MtmRecArgs A) {
    const int li = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (li >= A.nLocal) return;
    const int M = A.local[li];
    const MtmRow *gm = A.rows + M;
    int outKp = -1;
    float outDist = 0.f;
    // wave-uniform gates (:393-433)
    bool go = gm->frame_obs < 0 && gm->is3d && gm->has_desc;
    double wpt[3] = {0, 0, 0}, campt[3];
    float pu = 0.f, pv = 0.f;
    if (go) {
#pragma unroll
        for (int k = 0; k < 3; k++) wpt[k] = gm->X[k];
        se3_apply(A.kfQ + 4 * (size_t) A.frameKf, A.kfT + 3 * (size_t) A.frameKf, wpt, campt);
        go = !(campt[2] < 0.1);
        if (go) {
            const float view_angle = (float) (campt[2] / sqrt((campt[0] * campt[0] + campt[1] * campt[1]) + campt[2] * campt[2]));
            go = !(fabsf(view_angle) < A.viewTh);
        }
        if (go) {
            project_dist(A.calib, campt, pu, pv);
            go = pu >= 0 && pv >= 0 && (double) pu < A.calib[8] && (double) pv < A.calib[9];
        }
    }
    if (go) {
        const int rkp = (int) floorf(pv / (float) A.cellSize), ckp = (int) floorf(pu / (float) A.cellSize);
        int cb[4], cn[4], total = 0;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int r = rkp - 1 + (j >> 1), c = ckp - 1 + (j & 1), idx = r * A.numCellsW + c;
            const bool ok = !(r < 0 || c < 0 || idx > A.gridCells) && idx < A.gridCells;
            cb[j] = ok ? A.cellPtr[idx] : 0;
            cn[j] = ok ? A.cellPtr[idx + 1] - cb[j] : 0;
            total += cn[j];
        }
        const alva_medoid::Table *tm = A.tables + gm->slot;
        const int mn = gm->n_obs, mnd = gm->n_desc;
        float bestD = A.minDist, secD = A.minDist;
        int bestK = -1, nValid = 0;
        for (int base = 0; base < total; base += 64) {
            const int i = base + lane;
            bool valid = false;
            float dist = 0.f;
            int K = -1;
            if (i < total) {
                int j = 0, off = i;
                while (off >= cn[j]) {
                    off -= cn[j];
                    j++;
                }
                K = A.cellMp[cb[j] + off];
                const MtmRow *gk = A.rows + K;
                const int ko = gk->frame_obs, kn = gk->n_obs;
                // (a record without the frame keyframe among its observations -- the host's guard against it runs only under
                // ALVA_CHECK_OBS_MIRROR -- is no candidate; obs[-1] would be the row's own header bytes)
                const float pxDist = ko >= 0 ? norm2f(pu - gk->obs[ko].px[0], pv - gk->obs[ko].px[1]) : 0.f;
                bool cand = ko >= 0 && !(pxDist > A.maxPxDist);
                if (!gk->has_desc) cand = false;  // kpMapPoint->desc_.empty() (:465-468)
                if (cand)  // never both observed in one keyframe (:474-485)
                    for (int a = 0; a < kn && cand; a++)
                        for (int b = 0; b < mn; b++)
                            if (gk->obs[a].kf == gm->obs[b].kf) {
                                cand = false;
                                break;
                            }
                if (cand) {  // mean re-projection error of the map point in the keypoint's keyframes (:487-515)
                    float coProj = 0.f;
                    int nCo = 0;
                    for (int a = 0; a < kn; a++) {
                        double cp[3];
                        float qu, qv;
                        const int kf = gk->obs[a].kf;
                        se3_apply(A.kfQ + 4 * (size_t) kf, A.kfT + 3 * (size_t) kf, wpt, cp);
                        project_dist(A.calib, cp, qu, qv);
                        const float dx = gk->obs[a].px[0] - qu, dy = gk->obs[a].px[1] - qv;
                        coProj = (float) ((double) coProj + sqrt((double) dx * (double) dx + (double) dy * (double) dy));
                        nCo++;
                    }
                    cand = !(coProj / (float) nCo > A.maxPxDist);
                }
                if (cand) {  // MapPoint::computeMinDescDist (map_point.cpp:206-222): every pair of the two descriptor tables
                    const alva_medoid::Table *tk = A.tables + gk->slot;
                    const int knd = gk->n_desc;
                    int dmin = 1000;
                    for (int a = 0; a < mnd; a++) {
                        const uint4 *da = reinterpret_cast<const uint4 *>(tm->slot[gm->desc_slot[a]].desc);
                        for (int b = 0; b < knd; b++)
                            dmin = min(dmin, hamming256(da, reinterpret_cast<const uint4 *>(tk->slot[gk->desc_slot[b]].desc)));
                    }
                    dist = (float) dmin;
                    valid = dist <= A.minDist;  // larger distances fail both `<=` tests of the scan (:519-531)
                }
            }
            const unsigned long long vm = __ballot(valid);
            if (vm) {
                float cmin = valid ? dist : 3.0e38f;
#pragma unroll
                for (int off = 32; off > 0; off >>= 1) cmin = fminf(cmin, __shfl_xor(cmin, off));
                const unsigned long long mm = __ballot(valid && dist == cmin);
                const int lastLane = 63 - __clzll((long long) mm);
                float rest = (valid && lane != lastLane) ? dist : 3.0e38f;
#pragma unroll
                for (int off = 32; off > 0; off >>= 1) rest = fminf(rest, __shfl_xor(rest, off));
                const int cK = __shfl(K, lastLane);
                if (cmin <= bestD) {
                    const float oldBest = nValid > 0 ? bestD : 3.0e38f, oldSec = nValid > 1 ? secD : 3.0e38f;
                    secD = fminf(fminf(oldBest, oldSec), rest);
                    bestD = cmin;
                    bestK = cK;
                } else {
                    const float oldSec = nValid > 1 ? secD : 3.0e38f;
                    secD = fminf(oldSec, cmin);
                }
                nValid += __popcll(vm);
            }
        }
        if (bestK != -1 && nValid > 1)
            if (0.9 * (double) secD < (double) bestD) bestK = -1;  // :534-540
        outKp = bestK;
        outDist = bestD;
    }
    if (lane == 0 && outKp >= 0) {
        const unsigned long long key = ((unsigned long long) (0xffffffffu - __float_as_uint(outDist)) << 32) | (unsigned) (li + 1);
        atomicMax(&A.arb[outKp], key);
    }
}

__global__ void __launch_bounds__(256) k_arbitrate_rows(MtmRecArgs A) {
    const int K = blockIdx.x * 256 + threadIdx.x;
    if (K >= A.nMp) return;
    const unsigned li1 = (unsigned) (A.arb[K] & 0xffffffffull);
    A.matchOfMp[K] = li1 ? A.local[li1 - 1] : -1;
}

}  // namespace

extern "C" int alva_match_to_map(alva_ctx *ctx, const double *h_calib10, int cell_size, int num_cells_w, int grid_cells, const int *d_cell_ptr,
                                 const int *d_cell_mp, int n_kf, const double *d_kf_q, const double *d_kf_t, int n_mp, const double *d_mp_wpt,
                                 const uint8_t *d_mp_is3d, const int *d_obs_ptr, const int *d_obs_kf, const float *d_obs_px,
                                 const uint8_t *d_obs_desc, int frame_kf, int num_keypoints_3d, int n_local, const int *d_local,
                                 float max_proj_err, float dist_ratio, int *d_match_of_mp) {
    return alva_match_to_map_flags(ctx, h_calib10, cell_size, num_cells_w, grid_cells, d_cell_ptr, d_cell_mp, n_kf, d_kf_q, d_kf_t, n_mp, d_mp_wpt,
                                   d_mp_is3d, nullptr, d_obs_ptr, d_obs_kf, d_obs_px, d_obs_desc, nullptr, frame_kf, num_keypoints_3d, n_local,
                                   d_local, max_proj_err, dist_ratio, d_match_of_mp);
}

extern "C" int alva_match_to_map_flags(alva_ctx *ctx, const double *h_calib10, int cell_size, int num_cells_w, int grid_cells,
                                       const int *d_cell_ptr, const int *d_cell_mp, int n_kf, const double *d_kf_q, const double *d_kf_t, int n_mp,
                                       const double *d_mp_wpt, const uint8_t *d_mp_is3d, const uint8_t *d_mp_has_desc, const int *d_obs_ptr,
                                       const int *d_obs_kf, const float *d_obs_px, const uint8_t *d_obs_desc, const uint8_t *d_obs_has_desc,
                                       int frame_kf, int num_keypoints_3d, int n_local, const int *d_local, float max_proj_err,
                                       float dist_ratio, int *d_match_of_mp) {
    ALVA_ARG(ctx && h_calib10 && cell_size > 0 && num_cells_w > 0 && grid_cells > 0 && n_kf > 0 && n_mp >= 0 && n_local >= 0);
    ALVA_ARG(frame_kf >= 0 && frame_kf < n_kf);
    if (n_mp == 0) return ALVA_OK;
    ALVA_ARG(d_cell_ptr && d_cell_mp && d_kf_q && d_kf_t && d_mp_wpt && d_mp_is3d && d_obs_ptr && d_obs_kf && d_obs_px && d_obs_desc &&
             d_match_of_mp && (n_local == 0 || d_local) && ((uintptr_t) d_obs_desc % 16) == 0);
    MtmArgs A{};
    for (int i = 0; i < 10; i++) A.calib[i] = h_calib10[i];
    A.cellSize = cell_size; A.numCellsW = num_cells_w; A.gridCells = grid_cells;
    A.cellPtr = d_cell_ptr; A.cellMp = d_cell_mp; A.kfQ = d_kf_q; A.kfT = d_kf_t;
    A.mpHasDesc = d_mp_has_desc; A.obsHasDesc = d_obs_has_desc;
    A.nMp = n_mp; A.mpWpt = d_mp_wpt; A.mpIs3d = d_mp_is3d; A.obsPtr = d_obs_ptr; A.obsKf = d_obs_kf; A.obsPx = d_obs_px; A.obsDesc = d_obs_desc;
    A.frameKf = frame_kf; A.nLocal = n_local; A.local = d_local;
    // thresholds exactly as the reference forms them (:364-387, :439): floats, atanf / cosf of the host libm
    const float fovV = 0.5 * h_calib10[9] / h_calib10[1], fovH = 0.5 * h_calib10[8] / h_calib10[0];
    const float maxRadFov = fovH > fovV ? std::atan(fovH) : std::atan(fovV);
    A.viewTh = std::cos(maxRadFov);
    float maxPxDist = max_proj_err;
    if (num_keypoints_3d < 30) maxPxDist *= 2.;
    A.maxPxDist = maxPxDist;
    A.minDist = 32 * dist_ratio * 8.;
    uint8_t *base = nullptr;
    const size_t off_arb = ((size_t) n_mp * 4 + 255) / 256 * 256;
    int rc = alva_ctx_scratch(ctx, 7, off_arb + (size_t) n_mp * 8, (void **) &base);
    if (rc) return rc;
    A.frameObs = (int *) base;
    A.arb = (unsigned long long *) (base + off_arb);
    A.matchOfMp = d_match_of_mp;
    hipLaunchKernelGGL(k_frame_obs, dim3(alva_divup(n_mp, 256)), dim3(256), 0, ctx->stream, A);
    if (n_local > 0) hipLaunchKernelGGL(k_match_local, dim3(alva_divup(n_local, 4)), dim3(256), 0, ctx->stream, A);
    hipLaunchKernelGGL(k_arbitrate, dim3(alva_divup(n_mp, 256)), dim3(256), 0, ctx->stream, A);
    ALVA_LAUNCH_CHECK();
    return ALVA_OK;
}

extern "C" int alva_match_to_map_records(alva_ctx *ctx, const double *h_calib10, int cell_size, int num_cells_w, int grid_cells,
                                         const int *d_cell_ptr, const int *d_cell_mp, int n_kf, const int *d_kf_ids, const double *d_kf_q,
                                         const double *d_kf_t, int frame_kf_index, int frame_kf_id, int n_mp, const int *d_mp_slot,
                                         const void *const *d_record_chunks, const void *d_desc_tables, int num_keypoints_3d, int n_local,
                                         const int *d_local, float max_proj_err, float dist_ratio, int *d_match_of_mp) {
    ALVA_ARG(ctx && h_calib10 && cell_size > 0 && num_cells_w > 0 && grid_cells > 0 && n_kf > 0 && n_kf <= 64 && n_mp >= 0 && n_local >= 0);
    ALVA_ARG(frame_kf_index >= 0 && frame_kf_index < n_kf);
    if (n_mp == 0) return ALVA_OK;
    ALVA_ARG(d_cell_ptr && d_cell_mp && d_kf_ids && d_kf_q && d_kf_t && d_mp_slot && d_record_chunks && d_desc_tables && d_match_of_mp &&
             (n_local == 0 || d_local));
    MtmRecArgs A{};
    for (int i = 0; i < 10; i++) A.calib[i] = h_calib10[i];
    A.cellSize = cell_size; A.numCellsW = num_cells_w; A.gridCells = grid_cells;
    A.cellPtr = d_cell_ptr; A.cellMp = d_cell_mp; A.nKf = n_kf; A.kfIds = d_kf_ids; A.kfQ = d_kf_q; A.kfT = d_kf_t;
    A.nMp = n_mp; A.mpSlot = d_mp_slot;
    A.chunks = reinterpret_cast<const MpRec *const *>(d_record_chunks);
    A.tables = static_cast<const alva_medoid::Table *>(d_desc_tables);
    A.frameKfId = frame_kf_id; A.frameKf = frame_kf_index; A.nLocal = n_local; A.local = d_local;
    // thresholds exactly as the reference forms them (:364-387, :439): floats, atanf / cosf of the host libm
    const float fovV = 0.5 * h_calib10[9] / h_calib10[1], fovH = 0.5 * h_calib10[8] / h_calib10[0];
    const float maxRadFov = fovH > fovV ? std::atan(fovH) : std::atan(fovV);
    A.viewTh = std::cos(maxRadFov);
    float maxPxDist = max_proj_err;
    if (num_keypoints_3d < 30) maxPxDist *= 2.;
    A.maxPxDist = maxPxDist;
    A.minDist = 32 * dist_ratio * 8.;
    uint8_t *base = nullptr;
    const size_t off_arb = (size_t) n_mp * sizeof(MtmRow);
    int rc = alva_ctx_scratch(ctx, 7, off_arb + (size_t) n_mp * 8, (void **) &base);
    if (rc) return rc;
    A.rows = (MtmRow *) base;
    A.arb = (unsigned long long *) (base + off_arb);
    A.matchOfMp = d_match_of_mp;
    hipLaunchKernelGGL(k_mtm_gather, dim3(alva_divup(n_mp, 4)), dim3(256), 0, ctx->stream, A);
    if (n_local > 0) hipLaunchKernelGGL(k_match_rows, dim3(alva_divup(n_local, 4)), dim3(256), 0, ctx->stream, A);
    hipLaunchKernelGGL(k_arbitrate_rows, dim3(alva_divup(n_mp, 256)), dim3(256), 0, ctx->stream, A);
    ALVA_LAUNCH_CHECK();
    return ALVA_OK;
}
