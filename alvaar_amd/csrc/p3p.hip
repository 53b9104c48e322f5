// a8: P3P (Kneip) + least-median-of-squares absolute pose, hypothesis-parallel, FP64.
//
// Replaces MultiViewGeometry::p3pRansac (src/slam/src/multi_view_geometry.cpp:24-127) =
// opengv::sac::Lmeds<AbsolutePoseSacProblem(KNEIP)>::computeModel
// (src/libs/opengv/include/opengv/sac/implementation/Lmeds.hpp:43-195), with
//   sampling   SampleConsensusProblem.hpp:40-120 (std::mt19937 + prefix Fisher-Yates; done on the HOST with
//              the same libstdc++ classes, so the index stream is the reference's by construction)
//   model      AbsolutePoseSacProblem.cpp:35-163 (Kneip P3P on samples 0..2, disambiguate on the 4th)
//   p3p        src/absolute_pose/modules/main.cpp:50-205, quartic src/math/roots.cpp:88-135
//   score      AbsolutePoseSacProblem.cpp:165-199  (1 - f . normalize(R^T (X - t)))
//
// The reference evaluates hypotheses one after another; they are independent, so here
//   k_p3p      ONE launch, one workgroup per drawn sample: hypothesis (4 lanes = 4 quartic roots) -> N squared scores
//              into LDS, radix-select of the median -> the last workgroup to finish replays the sequential
//              "first max_iters valid, strict <" selection and classifies the winner's inliers
// FP64 VALU work (100 x N x ~40 flop); bytes are negligible (48 N read once per hypothesis,
// L2-resident).  No MFMA: nothing here is GEMM-shaped.
#include "common.hpp"
#include "multi_kernel.hpp"
#include <atomic>
#include "pose_internal.hpp"
#include <cmath>
#include <ctime>
#include <random>

#include "p3p_device.hpp"

namespace {

// The single-problem kernel runs P3P_NT threads per workgroup: ~130 workgroups never fill the chip (one per CU), so the scoring pass, the
// select and the winner's inlier pass -- all strided by the workgroup size -- shorten with it (stamps, 2300 points, 256 -> 512 threads:
// score 3.6 us, select 5.3 us, inliers 5.1 us per workgroup before).
constexpr int P3P_NT = 512;
__global__ void __launch_bounds__(P3P_NT) k_p3p(P3pArgs A) { (void) p3p_block<0, P3P_NT>(A, blockIdx.x); }
// several sessions' problems in one launch (lane.hpp): the samples come through A.samples (pinned host memory), as in k_p3p
ALVA_MULTI_KERNEL(MK_P3P, k_p3p_multi, P3pArgs, dim3(P3P_NT), P3P_NT, (void) p3p_block<0, P3P_NT>(A, bx));
__global__ void __launch_bounds__(P3P_NT) k_p3p_s(P3pArgs A, P3pInlineSamples S) { (void) p3p_block<0, P3P_NT>(A, blockIdx.x, S.v + 4 * blockIdx.x); }

// B independent problems in one launch: blockIdx.y = problem (camera), each with its own correspondences, sample list, scratch
// and arrival counter.  Dynamic LDS is sized for the largest problem.
// In a batch the Kneip solves get their own launch, 16 hypotheses per wave: inside k_p3p they occupy one wave of the four while the
// other three wait, and their register need (120 VGPRs) would cap the scoring kernel at four workgroups per CU.
__global__ void __launch_bounds__(256) k_p3p_hyp_batch(const P3pArgs *__restrict__ args) {
    const P3pArgs A = args[blockIdx.y];
    const int h = (int) (blockIdx.x * 64 + (threadIdx.x >> 2));
    if (h >= A.H) return;  // whole groups of 4 lanes leave together (the xor shuffles stay inside a group)
    (void) p3p_hypothesis(A, h, threadIdx.x & 3, true, nullptr);
}

__global__ void __launch_bounds__(256) k_p3p_batch(const P3pArgs *__restrict__ args) {
    const P3pArgs A = args[blockIdx.y];
    if ((int) blockIdx.x >= A.H) return;
    (void) p3p_block<1, 256>(A, blockIdx.x);
}

__global__ void __launch_bounds__(256) k_p3p_select_batch(const P3pArgs *__restrict__ args) {
    const P3pArgs A = args[blockIdx.x];
    (void) p3p_block<2, 256>(A, 0);
}

// SampleConsensusProblem<M>: rng_dist_ = uniform_int_distribution<>(0, INT_MAX), rng_alg_ = std::mt19937
// seeded 12345u (or time+clock), shuffled_indices_ persists across draws (SampleConsensusProblem.hpp:40-84).
struct Sampler {
    std::mt19937 alg;
    std::uniform_int_distribution<> dist{0, std::numeric_limits<int>::max()};
    std::vector<int> shuffled;
    Sampler(int n, bool random_seed, uint32_t seed) : shuffled((size_t) n) {
        if (random_seed) alg.seed(static_cast<unsigned>(time(0)) + static_cast<unsigned>(clock()));
        else alg.seed(seed);
        for (int i = 0; i < n; i++) shuffled[(size_t) i] = i;
    }
    void draw(int *out4) {
        const size_t index_size = shuffled.size();
        for (unsigned i = 0; i < 4; ++i) std::swap(shuffled[i], shuffled[i + ((size_t) dist(alg) % (index_size - i))]);
        for (int i = 0; i < 4; i++) out4[i] = shuffled[(size_t) i];
    }
};

}  // namespace

// the first `count` outputs of the sampler's generator -- dist(alg) of Sampler::draw, which does not depend on the number of points
int alva_p3p_raw_draws(int count, int do_random, uint32_t seed, int *h_raw) {
    ALVA_ARG(count >= 0 && h_raw);
    Sampler s(0, do_random != 0, seed);
    for (int k = 0; k < count; k++) h_raw[k] = s.dist(s.alg);
    return ALVA_OK;
}

extern "C" int alva_p3p_draw_samples(int n_points, int count, int do_random, uint32_t seed, int *h_samples) {
    ALVA_ARG(n_points >= 4 && count >= 0 && h_samples);
    Sampler s(n_points, do_random != 0, seed);
    for (int k = 0; k < count; k++) s.draw(h_samples + 4 * k);
    return ALVA_OK;
}

int alva_p3p_prepare(alva_ctx *ctx, const double *d_bearings, const double *d_wpts, int n, int max_iters, float err_threshold, int do_random,
                     uint32_t seed, float fx, float fy, int H, int *pin_samples, P3pSelectOut *out, uint8_t *inlier, P3pArgs *args) {
    // LDS-resident median select: n * 8 B of keys beside ~8 KB of static LDS.  Up to 7168 keys fit the default 64 KB per workgroup; gfx950
    // has 160 KB per CU and a workgroup may take it all once the kernel's limit is raised (a 4K frame has ~10 k 3-D keypoints)
    ALVA_ARG(n >= 4 && n <= 19000 && args);
    float focal = fx + fy;          // multi_view_geometry.cpp:72-76
    focal /= 2.f;
    const double threshold = 1.0 - std::cos(std::atan((double) (err_threshold / focal)));
    if (pin_samples) {   // (null: the caller draws later, when it knows n -- the fused pose launch sizes everything for an upper bound)
        Sampler smp(n, do_random != 0, seed);
        for (int k = 0; k < H; k++) smp.draw(pin_samples + 4 * k);
    }
    // scratch layout: models | valid | penalty
    size_t off_valid = (size_t) H * 12 * sizeof(double);
    size_t off_pen = (off_valid + (size_t) H * sizeof(int) + 63) / 64 * 64;
    const size_t off_masks = (off_pen + (size_t) H * sizeof(double) + 63) / 64 * 64;
    size_t total = off_masks + (size_t) H * (((size_t) n + 63) / 64) * 8;
    uint8_t *base = nullptr;
    int rc = alva_ctx_scratch(ctx, 2, total, (void **) &base);
    if (rc) return rc;
    P3pArgs A{};
    A.bv = d_bearings;
    A.wpt = d_wpts;
    A.samples = pin_samples;
    A.n = n;
    A.H = H;
    A.max_iters = max_iters;
    A.threshold = threshold;
    A.models = (double *) base;
    A.valid = (int *) (base + off_valid);
    A.penalty = (double *) (base + off_pen);
    A.masks = (unsigned long long *) (base + off_masks);
    A.counter = ctx->d_counters;  // slot 0: zero between launches (the last workgroup resets it)
    A.out = out;
    A.inlier = inlier;
    A.dbg = alva_kstamp_buffer();
    *args = A;
    return ALVA_OK;
}

int alva_p3p_launch(alva_ctx *ctx, const P3pArgs &A) {
    const int n = A.n, H = A.H;
    if (n > 7168) {
        // the attribute is per DEVICE (one code object per device); sessions on several host threads reach this concurrently
        static std::atomic<bool> raised[64];
        const bool tracked = ctx->device >= 0 && ctx->device < 64;
        if (!tracked || !raised[ctx->device].load(std::memory_order_acquire)) {
            ALVA_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_p3p), hipFuncAttributeMaxDynamicSharedMemorySize, 19000 * 8));
            ALVA_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_p3p_s), hipFuncAttributeMaxDynamicSharedMemorySize, 19000 * 8));
            if (tracked) raised[ctx->device].store(true, std::memory_order_release);
        }
    }
    // (n <= 7168: the multi kernel keeps the default dynamic-LDS limit)
    ctx->p3p_deferred = n <= 7168 && alva_lane_defer(MK_P3P, ctx, (unsigned) H, (unsigned) ((size_t) n * sizeof(double)), &A, sizeof(A));
    if (ctx->p3p_deferred) return ALVA_OK;
    if (H <= P3P_INLINE_H && alva_p3p_inline_samples_ok()) {
        P3pInlineSamples S;
        memcpy(S.v, A.samples, (size_t) H * 16);
        hipLaunchKernelGGL(k_p3p_s, dim3(H), dim3(P3P_NT), (size_t) n * sizeof(double), ctx->stream, A, S);
    } else {
        hipLaunchKernelGGL(k_p3p, dim3(H), dim3(P3P_NT), (size_t) n * sizeof(double), ctx->stream, A);
    }
    ALVA_LAUNCH_CHECK();
    return ALVA_OK;
}

bool alva_p3p_inline_samples_ok() {
    static const bool ok = getenv("ALVA_P3P_NO_INLINE_SAMPLES") == nullptr;
    return ok;
}

int alva_p3p_enqueue(alva_ctx *ctx, const double *d_bearings, const double *d_wpts, int n, int max_iters, float err_threshold,
                     int do_random, uint32_t seed, float fx, float fy, int H, int *pin_samples, P3pSelectOut *out, uint8_t *inlier) {
    P3pArgs A{};
    const int rc = alva_p3p_prepare(ctx, d_bearings, d_wpts, n, max_iters, err_threshold, do_random, seed, fx, fy, H, pin_samples, out, inlier, &A);
    if (rc) return rc;
    return alva_p3p_launch(ctx, A);
}

extern "C" int alva_p3p_lmeds(alva_ctx *ctx, const double *d_bearings, const double *d_wpts, int n, int max_iters, float err_threshold,
                              int do_random, uint32_t seed, float fx, float fy, double *h_R, double *h_t, int *h_outliers,
                              int *h_n_outliers, int *h_ok) {
    ALVA_ARG(ctx && h_R && h_t && h_outliers && h_n_outliers && h_ok && max_iters > 0);
    *h_ok = 0;
    *h_n_outliers = 0;
    if (n < 4) return ALVA_OK;  // multi_view_geometry.cpp:41-44
    ALVA_ARG(d_bearings && d_wpts);
    // draw max_iters + slack samples up front; hypotheses whose model fails do not count as an iteration in
    // the reference (Lmeds.hpp:88-92), so on the rare shortfall re-draw a longer prefix of the same stream.
    const int max_draws = max_iters + max_iters * 10;  // max_skip = 10 x max_iterations (Lmeds.hpp:67)
    int H = std::min(max_draws, max_iters + 28);
    SelectOut res{};
    const uint8_t *inl = nullptr;
    for (;;) {
        // pinned: samples | SelectOut | inlier mask -- the kernels read / write it directly, no copy commands
        const size_t off_out = ((size_t) H * 16 + 255) / 256 * 256, off_inl = off_out + 256;
        uint8_t *pin = nullptr;
        int rc = alva_ctx_pinned(ctx, off_inl + (size_t) n, (void **) &pin);
        if (rc) return rc;
        rc = alva_p3p_enqueue(ctx, d_bearings, d_wpts, n, max_iters, err_threshold, do_random, seed, fx, fy, H, (int *) pin,
                              (SelectOut *) (pin + off_out), pin + off_inl);
        if (rc) return rc;
        ALVA_HIP(alva_stream_sync(ctx->stream));
        memcpy(&res, pin + off_out, sizeof(res));
        inl = pin + off_inl;
        if (res.n_valid_used >= max_iters || H >= max_draws) break;
        H = std::min(max_draws, H * 2);
    }
    if (!res.have_model) return ALVA_OK;
    if (res.n_inliers < 5) return ALVA_OK;  // multi_view_geometry.cpp:82-85
    // Sophus::isOrthogonal(R): ||R R^T - I||_F < 1e-10 (:88-91; sophus/rotation_matrix.hpp:17-27)
    const double *R = res.model;
    double e = 0;
    for (int r = 0; r < 3; r++)
        for (int c = 0; c < 3; c++) {
            double v = R[3 * r] * R[3 * c] + R[3 * r + 1] * R[3 * c + 1] + R[3 * r + 2] * R[3 * c + 2] - (r == c ? 1.0 : 0.0);
            e += v * v;
        }
    if (!(std::sqrt(e) < 1e-10)) return ALVA_OK;
    for (int k = 0; k < 9; k++) h_R[k] = R[k];
    for (int k = 0; k < 3; k++) h_t[k] = res.model[9 + k];
    int no = 0;
    for (int i = 0; i < n; i++)
        if (!inl[i]) h_outliers[no++] = i;  // :109-124: outliers = complement of the inlier list
    *h_n_outliers = no;
    *h_ok = 1;
    return ALVA_OK;
}

// ---- batch of problems (internal: track_batch.hip) --------------------------------------------------------------------------
size_t alva_p3p_batch_item_size() { return sizeof(P3pArgs); }

// scratch bytes one problem with H hypotheses needs (models | valid | penalty), 64-byte granular
size_t alva_p3p_batch_scratch_bytes(int H) {
    const size_t off_valid = (size_t) H * 12 * sizeof(double);
    const size_t off_pen = (off_valid + (size_t) H * sizeof(int) + 63) / 64 * 64;
    return (off_pen + (size_t) H * sizeof(double) + 63) / 64 * 64;
}

int alva_p3p_batch_item_fill(void *dst, const double *d_bearings, const double *d_wpts, int n, int max_iters, float err_threshold, float fx,
                             float fy, int H, const int *d_samples, uint8_t *d_scratch, int *d_counter, P3pSelectOut *d_out,
                             uint8_t *d_inlier) {
    ALVA_ARG(dst && d_bearings && d_wpts && n >= 4 && n <= 7168 && H > 0 && d_samples && d_scratch && d_counter && d_out && d_inlier);
    float focal = fx + fy;
    focal /= 2.f;
    P3pArgs A{};
    A.bv = d_bearings;
    A.wpt = d_wpts;
    A.samples = d_samples;
    A.n = n;
    A.H = H;
    A.max_iters = max_iters;
    A.threshold = 1.0 - std::cos(std::atan((double) (err_threshold / focal)));
    const size_t off_valid = (size_t) H * 12 * sizeof(double);
    const size_t off_pen = (off_valid + (size_t) H * sizeof(int) + 63) / 64 * 64;
    A.models = (double *) d_scratch;
    A.valid = (int *) (d_scratch + off_valid);
    A.penalty = (double *) (d_scratch + off_pen);
    A.counter = d_counter;
    A.out = reinterpret_cast<SelectOut *>(d_out);
    A.inlier = d_inlier;
    memcpy(dst, &A, sizeof(A));
    return ALVA_OK;
}

int alva_p3p_batch_enqueue(alva_ctx *ctx, const void *d_items, int count, int H_max, int n_max) {
    ALVA_ARG(ctx && d_items && count > 0 && count <= 65535 && H_max > 0 && n_max >= 4 && n_max <= 7168);
    hipLaunchKernelGGL(k_p3p_hyp_batch, dim3(alva_divup(H_max, 64), count), dim3(256), 0, ctx->stream, (const P3pArgs *) d_items);
    hipLaunchKernelGGL(k_p3p_batch, dim3(H_max, count), dim3(256), (size_t) n_max * sizeof(double), ctx->stream, (const P3pArgs *) d_items);
    hipLaunchKernelGGL(k_p3p_select_batch, dim3(count), dim3(256), 0, ctx->stream, (const P3pArgs *) d_items);
    ALVA_LAUNCH_CHECK();
    return ALVA_OK;
}
