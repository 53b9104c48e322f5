// a6: ORB description of supplied points = 7x7 sigma=2 Gaussian + 256-bit steered BRIEF.
//
// Restates (reference paths relative to /root/reference/src/libs/opencv/modules):
//   * GaussianBlur(level, level, Size(7,7), 2, 2, BORDER_REFLECT_101)   features2d/src/orb.cpp:1188
//     -> sepFilter2D float path: row  RowFilter<uchar,float>  s = k0*p0; s += k_i*p_i (i ascending)
//                                col  SymmColumnFilter<float,uchar>  s = c0*r0; s += c_j*(r[+j] + r[-j])
//        (imgproc/src/filter.simd.hpp:468-507,1163-1209,2446-2488), result cvRound + saturate.
//        One IEEE rounding per written operation, no FMA (this TU is built with -ffp-contract=off).
//        Taps = cv::getGaussianKernel(7, 2, CV_32F) (bit patterns below, taken from the reference build).
//   * computeOrbDescriptors                                               features2d/src/orb.cpp:219-284
//        sample = center + (cvRound(x*a - y*b), cvRound(x*b + y*a)),  bit = I(p) < I(q)
//   * describeFeaturePoints keeps a point iff Rect(31,31,w-62,h-62).contains(Point(cvRound(pt)))
//        (features2d/src/keypoint.cpp:105-117; src/slam/src/feature_extractor.cpp:191-209); angle is the
//        constant -1 degree because KeyPoint::convert sets angle = -1 (core/src/types.cpp:93-101).
#include "common.hpp"
#include <algorithm>
#include <cmath>

namespace {

__constant__ int8_t c_pattern[1024] = {
#include "orb_pattern.inc"
};

// cv::getGaussianKernel(7, 2.0, CV_32F): 0.07015932 0.13107488 0.19071282 0.21610594 (symmetric)
__device__ __forceinline__ float gk(int i) {
    const uint32_t bits[4] = {0x3d8fafb1u, 0x3e06387eu, 0x3e434a39u, 0x3e5d4ae0u};
    return __uint_as_float(bits[i > 3 ? 6 - i : i]);
}

__device__ __forceinline__ int reflect101(int p, int len) {
    // single reflection is enough for |overshoot| <= 3 on images wider than 3 px
    if (p < 0) p = -p;
    if (p >= len) p = 2 * (len - 1) - p;
    return p;
}

constexpr int BT_W = 64, BT_H = 16;

// One block: BT_W x BT_H outputs.  LDS: u8 tile with 3 px halo, then row-filtered floats.
__device__ __forceinline__ void blur7_tile(const uint8_t *__restrict__ src, size_t src_pitch, int w, int h, uint8_t *__restrict__ dst,
                                           size_t dst_pitch, int bx, int by) {
    __shared__ uint8_t s_px[BT_H + 6][BT_W + 8];
    __shared__ float s_row[BT_H + 6][BT_W + 1];
    const int tx = threadIdx.x % BT_W, ty = threadIdx.x / BT_W;  // 64 x 4
    const int x0 = bx * BT_W, y0 = by * BT_H;
    for (int i = threadIdx.x; i < (BT_H + 6) * (BT_W + 6); i += 256) {
        int ly = i / (BT_W + 6), lx = i % (BT_W + 6);
        int gx = reflect101(x0 + lx - 3, w), gy = reflect101(y0 + ly - 3, h);
        gx = min(max(gx, 0), w - 1);
        gy = min(max(gy, 0), h - 1);
        s_px[ly][lx] = src[(size_t) gy * src_pitch + gx];
    }
    __syncthreads();
    for (int ly = ty; ly < BT_H + 6; ly += 4) {
        float s = gk(0) * (float) s_px[ly][tx];
#pragma unroll
        for (int k = 1; k < 7; k++) s += gk(k) * (float) s_px[ly][tx + k];
        s_row[ly][tx] = s;
    }
    __syncthreads();
    const int gx = x0 + tx;
    for (int ly = ty; ly < BT_H; ly += 4) {
        int gy = y0 + ly;
        float s = gk(3) * s_row[ly + 3][tx];
#pragma unroll
        for (int k = 1; k <= 3; k++) s += gk(3 + k) * (s_row[ly + 3 + k][tx] + s_row[ly + 3 - k][tx]);
        int v = __float2int_rn(s);  // cvRound: round half to even
        v = min(max(v, 0), 255);
        if (gx < w && gy < h) dst[(size_t) gy * dst_pitch + gx] = (uint8_t) v;
    }
}

__global__ void __launch_bounds__(256) k_blur7(const uint8_t *__restrict__ src, size_t src_pitch, int w, int h,
                                               uint8_t *__restrict__ dst, size_t dst_pitch) {
    blur7_tile(src, src_pitch, w, h, dst, dst_pitch, blockIdx.x, blockIdx.y);
}

struct BlurBatch {
    const uint8_t *src[12];
    uint8_t *dst[12];
    int w[12], h[12], pitch[12];
};

// all pyramid levels in ONE launch (blockIdx.y = level, blockIdx.x = tile)
__global__ void __launch_bounds__(256) k_blur7_batch(BlurBatch B) {
    const int l = blockIdx.y;
    const int tilesX = (B.w[l] + BT_W - 1) / BT_W, tilesY = (B.h[l] + BT_H - 1) / BT_H;
    if ((int) blockIdx.x >= tilesX * tilesY) return;
    blur7_tile(B.src[l], (size_t) B.pitch[l], B.w[l], B.h[l], B.dst[l], (size_t) B.pitch[l], blockIdx.x % tilesX, blockIdx.x / tilesX);
}

// 32 lanes per keypoint: lane = descriptor byte = 8 tests = 16 samples.
__global__ void __launch_bounds__(256) k_brief(const uint8_t *__restrict__ img, size_t pitch, int w, int h,
                                               const float *__restrict__ pts, int n, float a, float b,
                                               uint8_t *__restrict__ desc, uint8_t *__restrict__ valid) {
    const int kp = blockIdx.x * 8 + threadIdx.x / 32;
    const int byte = threadIdx.x % 32;
    if (kp >= n) return;
    const float px = pts[2 * kp], py = pts[2 * kp + 1];
    const int cx = __float2int_rn(px), cy = __float2int_rn(py);
    const bool ok = cx >= 31 && cx < w - 31 && cy >= 31 && cy < h - 31;
    if (byte == 0 && valid) valid[kp] = ok ? 1 : 0;
    uint32_t val = 0;
    if (ok) {
        const uint8_t *center = img + (size_t) cy * pitch + cx;
        const int8_t *pat = c_pattern + byte * 32;
#pragma unroll
        for (int t = 0; t < 8; t++) {
            float x0 = (float) pat[4 * t], y0 = (float) pat[4 * t + 1], x1 = (float) pat[4 * t + 2], y1 = (float) pat[4 * t + 3];
            int ix0 = __float2int_rn(x0 * a - y0 * b), iy0 = __float2int_rn(x0 * b + y0 * a);
            int ix1 = __float2int_rn(x1 * a - y1 * b), iy1 = __float2int_rn(x1 * b + y1 * a);
            int t0 = center[(ptrdiff_t) iy0 * (ptrdiff_t) pitch + ix0];
            int t1 = center[(ptrdiff_t) iy1 * (ptrdiff_t) pitch + ix1];
            val |= (uint32_t) (t0 < t1) << t;
        }
    }
    desc[(size_t) kp * 32 + byte] = (uint8_t) val;
}

}  // namespace

int alva_blur7_launch(alva_ctx *ctx, const uint8_t *d_src, size_t src_pitch, int w, int h, uint8_t *d_dst, size_t dst_pitch) {
    hipLaunchKernelGGL(k_blur7, dim3(alva_divup(w, BT_W), alva_divup(h, BT_H)), dim3(256), 0, ctx->stream, d_src, src_pitch, w, h,
                       d_dst, dst_pitch);
    ALVA_LAUNCH_CHECK();
    return ALVA_OK;
}

// ---- the same for several cameras: camera from alva_xcd_item, one BlurBatch per camera in device memory ------------------------------
// Throughput form of blur7_tile for the batched launch: the same 64 x 16 tile and the same float operations per pixel in the same
// order, but four pixels per thread -- dword loads into the byte tile (per-byte reflection only in dwords that cross the image
// border), v_cvt_f32_ubyteN unpacking, float4 LDS traffic, dword stores.  blur7_tile spends its time on byte-granular addressing
// (0.24 TB/s on 64 cameras); one camera's launch is latency-bound and keeps it.  Needs 4-byte aligned rows (the ORB pool's are).
__device__ __forceinline__ void blur7_tile4(const uint8_t *__restrict__ src, size_t src_pitch, int w, int h, uint8_t *__restrict__ dst,
                                            size_t dst_pitch, int bx, int by) {
    constexpr int PW = BT_W + 8, RW = BT_W + 4;                  // byte tile: columns x0 - 4 .. x0 + 67; float rows padded to 68
    __shared__ __attribute__((aligned(16))) uint8_t s_px[BT_H + 6][PW];
    __shared__ __attribute__((aligned(16))) float s_row[BT_H + 6][RW];
    const int x0 = bx * BT_W, y0 = by * BT_H;
    for (int i = threadIdx.x; i < (BT_H + 6) * (PW / 4); i += 256) {
        const int ly = i / (PW / 4), d = i - ly * (PW / 4);
        int gy = reflect101(y0 + ly - 3, h);
        gy = min(max(gy, 0), h - 1);
        const int gx0 = x0 - 4 + 4 * d;
        const uint8_t *row = src + (size_t) gy * src_pitch;
        uint32_t v;
        if (gx0 >= 0 && gx0 + 3 < w) {
            v = *reinterpret_cast<const uint32_t *>(row + gx0);
        } else {
            v = 0;
#pragma unroll
            for (int k = 0; k < 4; k++) {
                int gx = reflect101(gx0 + k, w);
                gx = min(max(gx, 0), w - 1);
                v |= (uint32_t) row[gx] << (8 * k);
            }
        }
        *reinterpret_cast<uint32_t *>(&s_px[ly][4 * d]) = v;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < (BT_H + 6) * (BT_W / 4); i += 256) {
        const int ly = i / (BT_W / 4), c = i - ly * (BT_W / 4);
        const uint32_t *rp = reinterpret_cast<const uint32_t *>(&s_px[ly][4 * c]);
        const uint32_t d0 = rp[0], d1 = rp[1], d2 = rp[2];
        float p[12];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            p[k] = (float) ((d0 >> (8 * k)) & 0xffu);
            p[4 + k] = (float) ((d1 >> (8 * k)) & 0xffu);
            p[8 + k] = (float) ((d2 >> (8 * k)) & 0xffu);
        }
        float4 o;
        float *op = &o.x;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            float s_ = gk(0) * p[j + 1];
#pragma unroll
            for (int k = 1; k < 7; k++) s_ += gk(k) * p[j + 1 + k];
            op[j] = s_;
        }
        *reinterpret_cast<float4 *>(&s_row[ly][4 * c]) = o;
    }
    __syncthreads();
    {
        const int ly = threadIdx.x / (BT_W / 4), c = threadIdx.x - ly * (BT_W / 4);  // 16 rows x 16 quads = 256 threads
        float4 r[7];
#pragma unroll
        for (int k = 0; k < 7; k++) r[k] = *reinterpret_cast<const float4 *>(&s_row[ly + k][4 * c]);
        uint32_t packed = 0;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            float s_ = gk(3) * (&r[3].x)[j];
#pragma unroll
            for (int k = 1; k <= 3; k++) s_ += gk(3 + k) * ((&r[3 + k].x)[j] + (&r[3 - k].x)[j]);
            int v = __float2int_rn(s_);
            v = min(max(v, 0), 255);
            packed |= (uint32_t) v << (8 * j);
        }
        const int gx = x0 + 4 * c, gy = y0 + ly;
        if (gy < h) {
            uint8_t *o = dst + (size_t) gy * dst_pitch + gx;
            if (gx + 3 < w) *reinterpret_cast<uint32_t *>(o) = packed;
            else
                for (int j = 0; j < 4; j++)
                    if (gx + j < w) o[j] = (uint8_t) (packed >> (8 * j));
        }
    }
}

__global__ void __launch_bounds__(256) k_blur7_multi(const BlurBatch *__restrict__ Bs, int n_levels, int count, int per_cam) {
    const AlvaXcdItem w = alva_xcd_item(count, per_cam);   // a camera's tiles on one XCD: the 3-row / 4-column halos hit its L2
    if (w.cam >= count) return;
    const BlurBatch &B = Bs[w.cam];
    int t = w.item, l = 0, tilesX = 1;  // the item index runs over the tiles of all levels
    for (; l < n_levels; l++) {
        tilesX = (B.w[l] + BT_W - 1) / BT_W;
        const int nt = tilesX * ((B.h[l] + BT_H - 1) / BT_H);
        if (t < nt) break;
        t -= nt;
    }
    if (l == n_levels) return;
    blur7_tile4(B.src[l], (size_t) B.pitch[l], B.w[l], B.h[l], B.dst[l], (size_t) B.pitch[l], t % tilesX, t / tilesX);
}

size_t alva_blur7_batch_size() { return sizeof(BlurBatch); }

int alva_blur7_batch_fill(void *out, int n, const uint8_t *const *src, uint8_t *const *dst, const int *w, const int *h, const int *pitch) {
    BlurBatch B{};
    int maxTiles = 0;
    for (int l = 0; l < n && l < 12; l++) {
        B.src[l] = src[l];
        B.dst[l] = dst[l];
        B.w[l] = w[l];
        B.h[l] = h[l];
        B.pitch[l] = pitch[l];
        if (pitch[l] % 4 || ((uintptr_t) src[l] % 4) || ((uintptr_t) dst[l] % 4)) return -1;  // blur7_tile4 moves dwords
        maxTiles += alva_divup(w[l], BT_W) * alva_divup(h[l], BT_H);   // total over the levels
    }
    memcpy(out, &B, sizeof(B));
    return maxTiles;
}

int alva_blur7_multi_launch(alva_ctx *ctx, const void *d_batches, int count, int n_levels, int total_tiles, hipStream_t on) {
    hipLaunchKernelGGL(k_blur7_multi, dim3(alva_xcd_grid(count, total_tiles)), dim3(256), 0, on ? on : ctx->stream, (const BlurBatch *) d_batches, n_levels, count,
                       total_tiles);
    ALVA_LAUNCH_CHECK();
    return ALVA_OK;
}

int alva_blur7_batch_launch(alva_ctx *ctx, int n, const uint8_t *const *src, uint8_t *const *dst, const int *w, const int *h,
                            const int *pitch) {
    BlurBatch B{};
    int maxTiles = 0;
    for (int l = 0; l < n && l < 12; l++) {
        B.src[l] = src[l];
        B.dst[l] = dst[l];
        B.w[l] = w[l];
        B.h[l] = h[l];
        B.pitch[l] = pitch[l];
        maxTiles = std::max(maxTiles, alva_divup(w[l], BT_W) * alva_divup(h[l], BT_H));
    }
    hipLaunchKernelGGL(k_blur7_batch, dim3(maxTiles, n), dim3(256), 0, ctx->stream, B);
    ALVA_LAUNCH_CHECK();
    return ALVA_OK;
}

int alva_brief_launch(alva_ctx *ctx, const uint8_t *d_blur, size_t pitch, int w, int h, const float *d_pts, int n, float a, float b,
                      uint8_t *d_desc, uint8_t *d_valid) {
    if (n <= 0) return ALVA_OK;
    hipLaunchKernelGGL(k_brief, dim3(alva_divup(n, 8)), dim3(256), 0, ctx->stream, d_blur, pitch, w, h, d_pts, n, a, b, d_desc, d_valid);
    ALVA_LAUNCH_CHECK();
    return ALVA_OK;
}

extern "C" int alva_orb_blur(alva_ctx *ctx, const uint8_t *d_gray, size_t gray_pitch, int width, int height, uint8_t *d_out,
                             size_t out_pitch) {
    ALVA_ARG(ctx && d_gray && d_out && width > 6 && height > 6 && gray_pitch >= (size_t) width && out_pitch >= (size_t) width);
    return alva_blur7_launch(ctx, d_gray, gray_pitch, width, height, d_out, out_pitch);
}

extern "C" int alva_describe(alva_ctx *ctx, const uint8_t *d_gray, size_t gray_pitch, int width, int height, const float *d_pts,
                             int n, uint8_t *d_desc, uint8_t *d_valid) {
    ALVA_ARG(ctx && d_gray && n >= 0 && width > 62 && height > 62 && gray_pitch >= (size_t) width);
    if (n == 0) return ALVA_OK;
    ALVA_ARG(d_pts && d_desc);
    uint8_t *blur = nullptr;
    size_t pitch = ((size_t) width + 63) / 64 * 64;
    int rc = alva_ctx_scratch(ctx, 1, pitch * height, (void **) &blur);
    if (rc) return rc;
    rc = alva_blur7_launch(ctx, d_gray, gray_pitch, width, height, blur, pitch);
    if (rc) return rc;
    // orb.cpp:232-235: angle (float, degrees) *= (float)(CV_PI/180.f); a = (float)cos(angle), b = (float)sin(angle)
    float angle = -1.0f;
    angle *= (float) (3.1415926535897932384626433832795 / 180.f);
    float a = (float) std::cos((double) angle), b = (float) std::sin((double) angle);
    return alva_brief_launch(ctx, blur, pitch, width, height, d_pts, n, a, b, d_desc, d_valid);
}
