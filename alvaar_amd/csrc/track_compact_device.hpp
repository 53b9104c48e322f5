// The compaction step of the slot-wise tracking frame (track_slots.hpp), shared by stages_hip.hip (k_track_compact) and pnp.hip (the
// fused pose launch, whose first phase it is).
#pragma once
#include "track_slots.hpp"
#include <hip/hip_runtime.h>

// The slot-wise step (track_slots.hpp): after the tracker launches every slot holds its verdict, position, undistorted position and
// bearing; what is left is the list of the pose solve -- the tracked 3-D slots in slot order (visual_frontend.cpp:275-298) -- the
// header, and the counters' reset for the next frame.
constexpr int CMP_NT = 256, CMP_MAX_WG = 32;
// CMP_NT_ = the workgroup's size (k_track_compact: CMP_NT; the fused pose launch of pnp.hip: its own 512).  Returns true in the thread that
// published the step's completion word (thread 0 of the workgroup that arrived last).
// GATHER_FIRST (the fused pose launch): the correspondences go out first, as 8-byte agent-scope atomic stores, and the workgroup arrives on
// a counter of its own (cnt[11]; the last arrival publishes cnt[10] = seq) BEFORE it turns to the host copies -- the pose solve's
// workgroups wait for the gathered arrays only, not for the system-scope fence and the 107 KB that cross the bus behind it.  One slice
// per workgroup (the caller sizes G so that a slice fits the workgroup).
template <int CMP_NT_, bool GATHER_FIRST = false>
__device__ __forceinline__ bool track_compact_body(const TrackSlots &D, const int g, const int G) {
    // SEVERAL workgroups (one used to do all of it: 107 KB to the host + 145 KB of gathers through one compute unit took 23 us).
    // Every workgroup owns a contiguous slice of the slots.  It counts the pose flags of the slots in front of its slice by itself
    // (2 bytes per slot: cheaper than a cross-workgroup scan), copies its slice's results to pinned host memory and gathers its slice's
    // correspondences.  Ordering against the completion word: every workgroup ends with a system-scope fence and an arrival on a device
    // counter; the workgroup that arrives LAST writes the header and then the word (system-scope release) -- the host reads the word
    // with acquire semantics, so it sees every slice.
    __shared__ int s_cnt[CMP_NT_ / 64 + 1];
    const int per = ((D.n + G - 1) / G + 63) / 64 * 64;   // slice length, a multiple of the wave size
    const int lo = g * per, hi = min(D.n, lo + per);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // GATHER_FIRST: the slice's rows requested NOW, beside the flag bytes of the scan below (one trip to memory the tracker wrote instead
    // of two behind each other: the scan's barriers stand between them otherwise).  One slice per workgroup: thread t owns slot lo + t.
    uint8_t h_code = 0, h_is3d = 0;
    float h_px[2] = {0, 0}, h_ux[2] = {0, 0};
    double h_bv[3] = {0, 0, 0}, h_w[3] = {0, 0, 0};
    if (GATHER_FIRST && lo + (int) threadIdx.x < hi) {
        const size_t j = (size_t) (lo + (int) threadIdx.x);
        h_code = D.d_code[j];
        h_is3d = D.d_is3d[j];
        h_px[0] = D.d_px[2 * j]; h_px[1] = D.d_px[2 * j + 1];
        h_ux[0] = D.d_unpx[2 * j]; h_ux[1] = D.d_unpx[2 * j + 1];
        h_bv[0] = D.d_bv[3 * j]; h_bv[1] = D.d_bv[3 * j + 1]; h_bv[2] = D.d_bv[3 * j + 2];
        h_w[0] = D.d_wpt[3 * j]; h_w[1] = D.d_wpt[3 * j + 1]; h_w[2] = D.d_wpt[3 * j + 2];
    }
    // pose flags in front of the slice (and, for the header, behind it): ballot counts
    int before = 0, total = 0;
    // (eight rounds' flag bytes requested before the first ballot: the rounds are dependent trips to memory another kernel wrote, and a
    // workgroup of the fused pose launch spent 5 of its 9 us here, one trip at a time)
    for (int i00 = 0; i00 < D.n; i00 += 8 * CMP_NT_) {
        uint8_t fc[8], f3[8];
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const int i = i00 + u * CMP_NT_ + (int) threadIdx.x;
            fc[u] = i < D.n ? D.d_code[i] : (uint8_t) 0;
            f3[u] = i < D.n ? D.d_is3d[i] : (uint8_t) 0;
        }
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const int i0 = i00 + u * CMP_NT_;
            if (i0 >= D.n) break;
            const bool f = fc[u] != 0 && f3[u] != 0;
            const unsigned long long bal = __ballot(f);
            const int wbase = i0 + wave * 64;
            if (lane == 0) {
                const int c = __popcll(bal);
                total += c;
                if (wbase + 64 <= lo) before += c;
                else if (wbase < lo) before += __popcll(bal & ((1ull << (lo - wbase)) - 1ull));   // (lo is a multiple of 64: never taken)
            }
        }
    }
    if (lane == 0) s_cnt[wave] = before;
    __syncthreads();
    int base = 0;
    for (int w = 0; w < CMP_NT_ / 64; w++) base += s_cnt[w];
    __syncthreads();
    if (lane == 0) s_cnt[wave] = total;
    __syncthreads();
    int n_pose = 0;
    for (int w = 0; w < CMP_NT_ / 64; w++) n_pose += s_cnt[w];
    __syncthreads();
    for (int i0 = lo; i0 < hi; i0 += CMP_NT_) {
        const int i = i0 + (int) threadIdx.x;
        uint8_t code = 0;
        bool pose = false;
        float px[2] = {0, 0}, ux[2] = {0, 0};
        double bv[3] = {0, 0, 0};
        if (GATHER_FIRST && i0 == lo) {   // (the rows requested at the top)
            if (i < hi) {
                code = h_code;
                px[0] = h_px[0]; px[1] = h_px[1];
                ux[0] = h_ux[0]; ux[1] = h_ux[1];
                bv[0] = h_bv[0]; bv[1] = h_bv[1]; bv[2] = h_bv[2];
                pose = code != 0 && h_is3d != 0;
            }
        } else if (i < hi) {   // the slot's results to the host, from THIS kernel (see track_slots.hpp)
            const size_t j = (size_t) i;
            code = D.d_code[i];
            px[0] = D.d_px[2 * j]; px[1] = D.d_px[2 * j + 1];
            ux[0] = D.d_unpx[2 * j]; ux[1] = D.d_unpx[2 * j + 1];
            bv[0] = D.d_bv[3 * j]; bv[1] = D.d_bv[3 * j + 1]; bv[2] = D.d_bv[3 * j + 2];
            if (!GATHER_FIRST) {
                D.o_code[i] = code;
                D.o_px[2 * j] = px[0]; D.o_px[2 * j + 1] = px[1];
                D.o_unpx[2 * j] = ux[0]; D.o_unpx[2 * j + 1] = ux[1];
                D.o_bv[3 * j] = bv[0]; D.o_bv[3 * j + 1] = bv[1]; D.o_bv[3 * j + 2] = bv[2];
            }
            pose = code != 0 && D.d_is3d[i] != 0;
        }
        const unsigned long long bal = __ballot(pose);
        if (lane == 0) s_cnt[wave] = __popcll(bal);
        __syncthreads();
        int wofs = 0, all = 0;
        for (int w = 0; w < CMP_NT_ / 64; w++) {
            if (w < wave) wofs += s_cnt[w];
            all += s_cnt[w];
        }
        if (pose) {
            const size_t k = (size_t) (base + wofs + __popcll(bal & ((1ull << lane) - 1ull))), j = (size_t) i;
            if (GATHER_FIRST) {
                auto put = [](double *p, double v) {
                    __hip_atomic_store(reinterpret_cast<unsigned long long *>(p), (unsigned long long) __double_as_longlong(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                };
                put(D.Pbv + 3 * k, bv[0]); put(D.Pbv + 3 * k + 1, bv[1]); put(D.Pbv + 3 * k + 2, bv[2]);
                put(D.Puv + 2 * k, (double) ux[0]); put(D.Puv + 2 * k + 1, (double) ux[1]);
                const bool first = i0 == lo;
                put(D.Pwpt + 3 * k, first ? h_w[0] : D.d_wpt[3 * j]); put(D.Pwpt + 3 * k + 1, first ? h_w[1] : D.d_wpt[3 * j + 1]);
                put(D.Pwpt + 3 * k + 2, first ? h_w[2] : D.d_wpt[3 * j + 2]);
            } else {
                D.Pbv[3 * k] = bv[0]; D.Pbv[3 * k + 1] = bv[1]; D.Pbv[3 * k + 2] = bv[2];
                D.Puv[2 * k] = (double) ux[0]; D.Puv[2 * k + 1] = (double) ux[1];
                D.Pwpt[3 * k] = D.d_wpt[3 * j]; D.Pwpt[3 * k + 1] = D.d_wpt[3 * j + 1]; D.Pwpt[3 * k + 2] = D.d_wpt[3 * j + 2];
            }
        }
        base += all;
        if (GATHER_FIRST) {
            // the slice's correspondences have landed (8-byte agent-scope stores + vmcnt(0): p3p_device.hpp's hand-off) -> arrive; then
            // the host copies of the same slot, still in this thread's registers
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (threadIdx.x == 0) {
                const int arrived = __hip_atomic_fetch_add(D.cnt + 11, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (arrived == G - 1) {
                    D.cnt[11] = 0;
                    __hip_atomic_store(D.cnt + 10, D.seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
            if (i < hi) {
                const size_t j = (size_t) i;
                D.o_code[i] = code;
                D.o_px[2 * j] = px[0]; D.o_px[2 * j + 1] = px[1];
                D.o_unpx[2 * j] = ux[0]; D.o_unpx[2 * j + 1] = ux[1];
                D.o_bv[3 * j] = bv[0]; D.o_bv[3 * j + 1] = bv[1]; D.o_bv[3 * j + 2] = bv[2];
            }
        }
        __syncthreads();
    }
    __threadfence_system();   // this thread's writes to the host (and the device) ...
    __syncthreads();          // ... of every thread of the workgroup, before its arrival
    if (threadIdx.x == 0) {
        const int arrived = __hip_atomic_fetch_add(D.cnt + 8, 1, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        if (arrived == G - 1) {   // last: every slice is out
            const unsigned long long packed = reinterpret_cast<unsigned long long *>(D.cnt)[2];   // the tracker launch's counts (track_slots.hpp)
            const int nA = (int) ((packed >> 16) & 0xffff), good = (int) (packed & 0xffff);
            const bool req = nA > 0 && (double) good < 0.33 * (double) nA;
            reinterpret_cast<unsigned long long *>(D.cnt)[2] = 0ull;
            for (int s = 0; s < TRK_STRIPES; s++) reinterpret_cast<unsigned long long *>(D.cnt)[16 + s] = 0ull;   // the tracker's arrival stripes
            D.cnt[8] = 0;
            // the step's completion word carries what the host reads of the header -- [seq : 32 | p3pReq_ : 1 | n_pose : 31] at
            // o_hdr[12..13], ONE 8-byte system-scope store: every slice's results are already behind its workgroup's fence + arrival,
            // so no second system-scope fence (an L2 write-back, ~2.5 us of the tracking step) stands in front of it
            const unsigned long long word = ((unsigned long long) (unsigned) D.seq << 32) | ((unsigned long long) (req ? 1 : 0) << 31) | (unsigned) n_pose;
            __hip_atomic_store(reinterpret_cast<unsigned long long *>(D.o_hdr + 12), word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            return true;
        }
    }
    return false;
}
