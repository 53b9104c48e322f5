// Shared between p3p.hip and pnp.hip: device-side hand-off of the P3P-LMedS result to the PnP refinement
// (VisualFrontend::computePose chains them, src/slam/src/visual_frontend.cpp:245-417).
#pragma once
#include "common.hpp"

struct P3pSelectOut {
    double model[12];  // R row-major (cam -> world) | t
    int best, n_valid_used, n_inliers, have_model;
};

// Enqueues the P3P-LMedS kernels on ctx->stream WITHOUT synchronising.  On return *d_out / *d_inlier point into context
// scratch (slot 2): the selection result and the per-point inlier mask (n bytes).
int alva_p3p_enqueue(alva_ctx *ctx, const double *d_bearings, const double *d_wpts, int n, int max_iters, float err_threshold,
                     int do_random, uint32_t seed, float fx, float fy, int n_draws, P3pSelectOut **d_out, uint8_t **d_inlier);
