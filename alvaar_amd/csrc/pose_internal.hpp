// Shared between p3p.hip and pnp.hip: device-side hand-off of the P3P-LMedS result to the PnP refinement
// (VisualFrontend::computePose chains them, src/slam/src/visual_frontend.cpp:245-417).
#pragma once
#include "common.hpp"

struct P3pSelectOut {
    double model[12];  // R row-major (cam -> world) | t
    int best, n_valid_used, n_inliers, have_model;
};

// Enqueues the P3P-LMedS kernels on ctx->stream WITHOUT synchronising and without copy commands.
//   pin_samples : H*4 ints of PINNED host memory (alva_ctx_pinned); filled here, read by the hypothesis kernel over the bus
//   out/inlier  : where the selection result and the n-byte inlier mask are written: context scratch when another kernel
//                 consumes them (alva_compute_pose), pinned host memory when the host does (alva_p3p_lmeds)
int alva_p3p_enqueue(alva_ctx *ctx, const double *d_bearings, const double *d_wpts, int n, int max_iters, float err_threshold,
                     int do_random, uint32_t seed, float fx, float fy, int n_draws, int *pin_samples, P3pSelectOut *out,
                     uint8_t *inlier);

// the same in two steps (pnp.hip's fused P3P -> PnP launch takes the prepared arguments and launches them itself): samples drawn into
// pin_samples, scratch carved, *args filled | the launch alva_p3p_enqueue makes (lane deposit, or k_p3p_s / k_p3p on ctx->stream)
struct P3pArgs;
int alva_p3p_prepare(alva_ctx *ctx, const double *d_bearings, const double *d_wpts, int n, int max_iters, float err_threshold, int do_random,
                     uint32_t seed, float fx, float fy, int n_draws, int *pin_samples, P3pSelectOut *out, uint8_t *inlier, P3pArgs *args);
int alva_p3p_launch(alva_ctx *ctx, const P3pArgs &args);
bool alva_p3p_inline_samples_ok();
int alva_p3p_raw_draws(int count, int do_random, uint32_t seed, int *h_raw);

// The fused tail of the single session's tracking frame (pnp.hip k_pose_all: compaction -> P3P -> PnP in one launch, queued right behind
// the tracker): enqueue | the host's answer once the tracker's early word gave it n (go: draws the samples; abort: the launch ends after
// its compaction phase).  Every enqueue MUST be answered by exactly one of the two; alva_compute_pose_collect_p3p collects a "go".
struct TrackSlots;
bool alva_pose_all_possible(int n_slots, int p3p_iters);
int alva_pose_all_enqueue(alva_ctx *ctx, const TrackSlots &D, int compact_workgroups, int p3p_iters, float p3p_err, int do_random, uint32_t seed,
                          int pnp_iters, float chi2_th, float fx, float fy, float cx, float cy);
int alva_pose_all_go(alva_ctx *ctx, int n);
int alva_pose_all_abort(alva_ctx *ctx);

// alva_compute_pose_collect that also returns the accepted P3P pose (pnp.hip)
int alva_compute_pose_collect_p3p(alva_ctx *ctx, double *h_pose7, double *h_pose7_p3p, uint8_t *h_p3p_outlier, uint8_t *h_pnp_outlier,
                                  int *h_status);
// fbKltTracking with the keypoint count in device memory (klt.hip)
int alva_fbklt_track_dn(alva_ctx *ctx, const alva_pyramid *prev, const alva_pyramid *curr, int num_levels, float err_thresh, float fb_dist,
                        int max_iters, float eps, const float *d_pts, const float *d_prior_in, float *d_out, uint8_t *d_status, const int *d_n,
                        int n_max);
