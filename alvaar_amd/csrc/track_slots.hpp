// The tracking step of one frame, slot-wise (shared by klt.hip, which tracks, and stages_hip.hip, which drives it).
//
// VisualFrontend::kltTrackingFromMotionPrior (src/slam/src/visual_frontend.cpp:152-243) builds two keypoint lists -- 3-D keypoints
// whose projection under the predicted pose falls into the image (one pyramid level, from that projection) and everything else
// (full pyramid, from the keypoint's own position), the failures of the first list joining the second.  What fbKltTracking
// computes for a keypoint does not depend on the keypoint's position in a list, so no list is built here: one workgroup per SLOT
// of the frame container decides by itself which of the two it is, and re-tracks its keypoint on the full pyramid at once when the
// one-level pass fails -- from where that pass left it, which is what the reference does unless fewer than 33 % of the one-level
// passes succeeded (p3pReq_, :193-203: then the retry starts from the keypoint's own position).  That count is known only after the
// launch: the compaction kernel puts it into the header, and in that (rare) case the host launches the retry kernel, which redoes the
// retried slots from their own positions, and the compaction once more.  The compaction kernel gathers the correspondences of the
// pose solve in slot order (:275-298) and copies the per-slot results to pinned host memory ITSELF, behind its own system-scope
// fences and in front of the completion word the host polls.  (Results written to host memory by the tracker's workgroups -- an
// earlier kernel, on other XCDs -- are not ordered against that word: kernels of one stream are separated by agent-scope releases
// only.  With the tracker writing them the host read a stale value about once in 10^5 slots.)
#pragma once
#include "camera_device.hpp"
#include <cstdint>

constexpr int TRK_STRIPES = 64;   // arrival stripes of the wave-per-slot tracker launch: (unsigned long long *) cnt + 16 .. + 16 + TRK_STRIPES - 1 (see cnt below)
static_assert((16 + TRK_STRIPES) * 8 <= 1024, "the arrival stripes must fit the 1024-byte counter block");
struct TrackSlots {
    int n, use_prior, width, height;
    const float *in_px;        // pinned host, [n][2]
    const uint8_t *in_is3d;    // pinned host, [n]
    const double *in_wpt;      // pinned host, [n][3]
    double q[4], t[3];         // T_cw (predicted)
    AlvaCam cam;
    const double *invK;        // device, 9
    int *cnt;                  // device, 1024 bytes, zeroed by the compaction kernel.  (unsigned long long *) cnt + 2 = the tracker launch's ONE
                               // packed counter: arrivals << 48 | tracked 3-D slots << 32 | slots tracked from their projection << 16 |
                               // successes of those (one atomic per workgroup; counts fit 16 bits: a frame holds < 65536 slots);
                               // cnt[8] = the compaction kernel's arrival counter
    float *d_pts;              // [n][2] device copies of the three inputs (one coalesced pass over the bus; a workgroup per slot reading
                               // its 33 bytes from host memory by itself is bound by the number of outstanding PCIe reads)
    uint8_t *d_code;           // per slot: 0 lost | 1 tracked from the projection | 2 tracked on the full pyramid | 3 re-tracked
    uint8_t *d_retried;        // per slot: 1 = the one-level pass failed and the slot was re-tracked from where that pass left it
    uint8_t *d_is3d;           // copy of in_is3d
    float *d_px, *d_unpx;      // [n][2] tracked position | undistorted
    double *d_bv, *d_wpt;      // [n][3]; d_wpt = copy of in_wpt
    uint8_t *o_code;           // pinned host outputs
    float *o_px, *o_unpx;
    double *o_bv;
    int *o_hdr;
    int seq;                   // the compaction's completion word [seq : 32 | p3pReq_ : 1 | n_pose : 31] at o_hdr[12..13] (system scope) after
                               // everything else: the host may poll it instead of waiting on the stream
                               // o_hdr[10..11] as ONE 64-bit word [seq : 32 | p3pReq_ : 1 | n_pose : 31]: the tracker's counts, published by
                               // the LAST workgroup of the tracker launch itself -- the host learns the size of the pose problem one kernel
                               // earlier and enqueues the pose solve (sample draw + two launches) while the compaction kernel runs
    double *Pbv, *Puv, *Pwpt;  // device: correspondences of the pose solve
    unsigned long long *dbg;   // null, or the per-slot stamp buffer of ALVA_KLT_STAMPS=1 (microbench.hip)
    // The table CARRIED from the previous frame (round 6): a frame that only lost slots since the previous tracker launch -- every frame
    // between two keyframes -- is the previous frame's table without the lost rows, with the tracked positions as the new positions.  All
    // of that is on the device already: the host names, per slot, the slot it was (carry, 2 bytes per slot in host-written device memory
    // instead of 33 + the walk over the map points that fetched them), the tracker's workgroup reads its row through that index from the
    // previous frame's buffers (p_*) and writes it to this frame's (d_pts / d_is3d / d_wpt: what the retry launch, the compaction and the
    // next frame read).  The two frames' buffers alternate.  null: d_pts / d_is3d / d_wpt were written by the host (or k_track_stage_in).
    const uint16_t *carry;     // [n]
    const float *p_px;         // previous frame: d_px
    const uint8_t *p_is3d;     //                 d_is3d
    const double *p_wpt;       //                 d_wpt
};

struct alva_ctx;
struct alva_pyramid;
// enqueue only (klt.hip): the first launch over all slots, the retry launch
int alva_track_slots_klt(alva_ctx *ctx, const alva_pyramid *prev, const alva_pyramid *curr, const TrackSlots &D, int levels_prior, int levels_full,
                         float err_thresh, float fb_dist, int max_iters, float eps, int retry);
