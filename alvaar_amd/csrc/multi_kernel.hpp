// The `*_multi` form of a chain kernel (lane.hpp): the sessions' argument blocks sit in a device table (uploaded by the lane in ONE copy
// per flush, all kinds together), blockIdx.y selects one -- wave-uniform, so the compiler reads it with scalar loads -- and the body is the
// single-session kernel's body with (A, bx, gx) in place of (its arguments, blockIdx.x, gridDim.x).  The launch grid is (max gx rounded up
// to a multiple of 8, sessions): workgroups go to XCD (linear id mod 8), so with a row length that is a multiple of 8 a body's own
// XCD-aware mapping of bx sees the same XCD as in its single-session launch.
#pragma once
#include "common.hpp"
#include <cstdlib>

static inline bool alva_multi_xcd_affinity() {
    static const bool on = !(getenv("ALVA_LANE_XCD") && atoi(getenv("ALVA_LANE_XCD")) == 0);
    return on;
}

#define ALVA_MULTI_KERNEL(KIND, KNAME, ARGS, BLOCK, BOUNDS, ...) ALVA_MULTI_KERNEL_ATTR(KIND, KNAME, ARGS, BLOCK, __launch_bounds__(BOUNDS), __VA_ARGS__)
#define ALVA_MULTI_KERNEL_ATTR(KIND, KNAME, ARGS, BLOCK, ATTRS, ...) ALVA_MULTI_KERNEL_ATTR_IF(true, KIND, KNAME, ARGS, BLOCK, ATTRS, __VA_ARGS__)
// (COND: register this kernel for its kind only when it holds -- alternative bodies of one kind, chosen once per process)
#define ALVA_MULTI_KERNEL_ATTR_IF(COND, KIND, KNAME, ARGS, BLOCK, ATTRS, ...)                                                     \
    static_assert(sizeof(ARGS) % 8 == 0, "argument blocks are 8-byte granular in the table");                                     \
    __global__ void ATTRS KNAME(const ARGS *__restrict__ table, const unsigned *__restrict__ gxs, const int count, const unsigned gmax) { \
        unsigned y;                                                                                                               \
        int bx;                                                                                                                   \
        if (gridDim.y == 1 && count > 1) {   /* session -> XCD affinity (alva_xcd_item's scheme): see the launcher below */       \
            const unsigned L = blockIdx.x, j = L >> 3;                                                                            \
            y = (L & 7u) + 8u * (j / gmax);                                                                                       \
            bx = (int) (j % gmax);                                                                                                \
            if (y >= (unsigned) count) return;                                                                                    \
        } else {                                                                                                                  \
            y = blockIdx.y;                                                                                                       \
            bx = (int) blockIdx.x;                                                                                                \
        }                                                                                                                         \
        const unsigned gx = gxs[y];                                                                                               \
        if ((unsigned) bx >= gx) return;                                                                                          \
        const ARGS &A = table[y];                                                                                                 \
        __VA_ARGS__;                                                                                                              \
    }                                                                                                                              \
    static void KNAME##_launch(hipStream_t st, const uint8_t *d_args, const unsigned *d_gx, int count, unsigned gmax, unsigned smax) { \
        const unsigned g8 = (gmax + 7u) & ~7u;                                                                                    \
        /* 8 sessions or more: workgroup L serves session 8 * (L / 8 / g8) + L % 8, so a session's workgroups all run on XCD L % 8 and \
           its images go through ONE L2 (each XCD has its own; with the session in blockIdx.y its tiles are dealt over all eight).  \
           Fewer: the plain 2-D grid, or XCDs would idle.  ALVA_LANE_XCD=0 keeps the 2-D grid (A/B). */                          \
        if (count >= 8 && alva_multi_xcd_affinity())                                                                              \
            hipLaunchKernelGGL(KNAME, dim3(8u * (unsigned) ((count + 7) / 8) * g8), BLOCK, smax, st, (const ARGS *) d_args, d_gx, count, g8); \
        else                                                                                                                      \
            hipLaunchKernelGGL(KNAME, dim3(g8, (unsigned) count), BLOCK, smax, st, (const ARGS *) d_args, d_gx, count, g8);       \
    }                                                                                                                              \
    static const int KNAME##_registered = (COND) ? alva_multi_register(KIND, #KNAME, sizeof(ARGS), &KNAME##_launch) : -1
