// Wave64 helpers shared by the kernels: broadcasts through v_readlane (no LDS crossbar round trip) and the butterfly
// reduce-scatter used for the 27/28-value normal-equation sums.
#pragma once
// This library's device code is written for gfx950 (MI355X) only: wave64 DPP row shifts, v_mfma_f64_16x16x4_f64, and -- in the tracking
// step, P3P and the local BA -- completion words that are RELAXED system-scope stores behind an agent-scope arrival counter
// (stages_hip.hip k_track_compact, klt.hip k_track_klt, ba.hip k_results): correct because on this part a store that has been
// acknowledged at system scope is visible to the host, posted PCIe writes stay in order, and s_waitcnt vmcnt(0) covers stores.  Another
// target must re-derive that (or go back to release stores + fences), so it is refused at compile time instead of inheriting it silently.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__)
#error "alvaar_amd device code targets gfx950 only (see wave_utils.hpp)"
#endif
#include <hip/hip_runtime.h>

// value held by `lane` (a wave-uniform index) delivered to every lane
__device__ __forceinline__ float lane_bcast(float v, int lane) {
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane));
}
__device__ __forceinline__ double lane_bcast(double v, int lane) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane), hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
    return __hiloint2double(hi, lo);
}

// v of the lane `n` positions below within the 16-lane DPP row (row_shr:n; lanes without a source read 0): a cross-lane move that costs
// no LDS round trip and no scalar register
template <int N>
__device__ __forceinline__ float dpp_row_shr(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x110 + N, 0xf, 0xf, true));
}

// Sum of 32 per-lane values over the 64 lanes of a wave: at distance d each lane keeps half of its values and adds the
// partner's copy of that half (16 + 8 + 4 + 2 + 1 + 1 = 32 exchanges instead of 32 x 6).  On return v[0] of lane l holds the
// wave total of value (l >> 1).  The summation order is fixed (bit-reproducible).
__device__ __forceinline__ void wave_reduce_scatter32(double (&v)[32]) {
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int half = 16; half >= 1; half >>= 1) {
        const bool hi = (lane & (2 * half)) != 0;
#pragma unroll
        for (int k = 0; k < half; k++) {
            // the two candidates pinned to registers first: otherwise the compiler rewrites "hi ? v[k] : v[k + half]" as
            // v[hi ? k : k + half], i.e. a lane-dependent index, and the whole array moves to scratch memory
            double lo_v = v[k], hi_v = v[k + half];
            asm volatile("" : "+v"(lo_v), "+v"(hi_v));
            const double send = hi ? lo_v : hi_v;
            const double keep = hi ? hi_v : lo_v;
            v[k] = keep + __shfl_xor(send, 2 * half);
        }
    }
    v[0] += __shfl_xor(v[0], 1);
}
