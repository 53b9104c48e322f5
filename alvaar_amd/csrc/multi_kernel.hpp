// The `*_multi` form of a chain kernel (lane.hpp): the sessions' argument blocks sit in a device table (uploaded by the lane in ONE copy
// per flush, all kinds together), blockIdx.y selects one -- wave-uniform, so the compiler reads it with scalar loads -- and the body is the
// single-session kernel's body with (A, bx, gx) in place of (its arguments, blockIdx.x, gridDim.x).  The launch grid is (max gx rounded up
// to a multiple of 8, sessions): workgroups go to XCD (linear id mod 8), so with a row length that is a multiple of 8 a body's own
// XCD-aware mapping of bx sees the same XCD as in its single-session launch.
#pragma once
#include "common.hpp"

#define ALVA_MULTI_KERNEL(KIND, KNAME, ARGS, BLOCK, BOUNDS, ...) ALVA_MULTI_KERNEL_ATTR(KIND, KNAME, ARGS, BLOCK, __launch_bounds__(BOUNDS), __VA_ARGS__)
#define ALVA_MULTI_KERNEL_ATTR(KIND, KNAME, ARGS, BLOCK, ATTRS, ...) ALVA_MULTI_KERNEL_ATTR_IF(true, KIND, KNAME, ARGS, BLOCK, ATTRS, __VA_ARGS__)
// (COND: register this kernel for its kind only when it holds -- alternative bodies of one kind, chosen once per process)
#define ALVA_MULTI_KERNEL_ATTR_IF(COND, KIND, KNAME, ARGS, BLOCK, ATTRS, ...)                                                     \
    static_assert(sizeof(ARGS) % 8 == 0, "argument blocks are 8-byte granular in the table");                                     \
    __global__ void ATTRS KNAME(const ARGS *__restrict__ table, const unsigned *__restrict__ gxs) {                               \
        const unsigned y = blockIdx.y, gx = gxs[y];                                                                               \
        const int bx = (int) blockIdx.x;                                                                                          \
        if ((unsigned) bx >= gx) return;                                                                                          \
        const ARGS &A = table[y];                                                                                                 \
        __VA_ARGS__;                                                                                                              \
    }                                                                                                                              \
    static void KNAME##_launch(hipStream_t st, const uint8_t *d_args, const unsigned *d_gx, int count, unsigned gmax, unsigned smax) { \
        hipLaunchKernelGGL(KNAME, dim3((gmax + 7u) & ~7u, (unsigned) count), BLOCK, smax, st, (const ARGS *) d_args, d_gx);       \
    }                                                                                                                              \
    static const int KNAME##_registered = (COND) ? alva_multi_register(KIND, #KNAME, sizeof(ARGS), &KNAME##_launch) : -1
