// f4a: CLAHE on the gray frame, bit-exact.
//
// Replaces cv::createCLAHE(claheContrastLimit_, Size(w / claheTileSize_, h / claheTileSize_))->apply(image, currImage_)
// (src/slam/src/visual_frontend.cpp:16-18, :678-681; off in the shipped configuration, system.cpp:17) =
// imgproc/src/clahe.cpp:120-420 for CV_8UC1:
//   k_clahe_lut     one workgroup per tile: histogram of the tile (of the REFLECT_101-extended image when the size is not
//                   divisible by the grid, :364-384) with LDS atomics, clip + redistribution (:185-212, integer), prefix sum,
//                   lut[i] = saturate_cast<uchar>(sum * lutScale) (float product, round half to even)
//   k_clahe_apply   per pixel: blend of the four neighbouring tiles' LUT entries in float, in the reference's operation
//                   order (:305-311), saturate_cast<uchar>
// HBM: the source is read twice (histogram pass, apply pass), the destination written once: 3 bytes per pixel; the LUTs
// (tiles x 256 B, 27 KB at 640x480 / tile 50) stay in L2.  This translation unit must be compiled with -ffp-contract=off.
#include "common.hpp"

namespace {

__device__ __forceinline__ int reflect101(int p, int len) {
    if (len == 1) return 0;
    while (p < 0 || p >= len) p = p < 0 ? -p : 2 * (len - 1) - p;
    return p;
}

__device__ __forceinline__ uint8_t sat_u8(float v) {
    const int r = __float2int_rn(v);  // cvRound: round half to even
    return (uint8_t) min(max(r, 0), 255);
}

__global__ void __launch_bounds__(256) k_clahe_lut(const uint8_t *__restrict__ src, size_t pitch, int w, int h, int tilesX, int tw, int th,
                                                   int clipLimit, float lutScale, uint8_t *__restrict__ lut) {
    __shared__ int s_hist[256], s_scan[256], s_red[256];
    const int k = blockIdx.x, ty = k / tilesX, tx = k - ty * tilesX, t = threadIdx.x;
    s_hist[t] = 0;
    __syncthreads();
    for (int e = t; e < tw * th; e += 256) {
        const int y = e / tw, x = e - y * tw;
        const int sx = reflect101(tx * tw + x, w), sy = reflect101(ty * th + y, h);
        atomicAdd(&s_hist[src[(size_t) sy * pitch + sx]], 1);
    }
    __syncthreads();
    int hv = s_hist[t];
    if (clipLimit > 0) {
        const int excess = hv > clipLimit ? hv - clipLimit : 0;
        hv = min(hv, clipLimit);
        s_red[t] = excess;
        __syncthreads();
        for (int s2 = 128; s2 > 0; s2 >>= 1) {
            if (t < s2) s_red[t] += s_red[t + s2];
            __syncthreads();
        }
        const int clipped = s_red[0];
        const int batch = clipped / 256, residual = clipped - batch * 256;
        hv += batch;
        if (residual != 0) {
            const int step = max(256 / residual, 1);
            // bins 0, step, 2 step, ... get one more each, `residual` of them at most, while the bin index stays < 256 (:205-211)
            if (t % step == 0 && t / step < residual) hv++;
        }
    }
    // inclusive prefix sum over the 256 bins
    s_scan[t] = hv;
    __syncthreads();
    for (int off = 1; off < 256; off <<= 1) {
        const int v = t >= off ? s_scan[t - off] : 0;
        __syncthreads();
        s_scan[t] += v;
        __syncthreads();
    }
    lut[(size_t) k * 256 + t] = sat_u8((float) s_scan[t] * lutScale);
}

__global__ void __launch_bounds__(256) k_clahe_apply(const uint8_t *__restrict__ src, size_t spitch, int w, int h, int tilesX, int tilesY,
                                                     float inv_tw, float inv_th, const uint8_t *__restrict__ lut, uint8_t *__restrict__ dst,
                                                     size_t dpitch) {
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= w || y >= h) return;
    const float tyf = (float) y * inv_th - 0.5f, txf = (float) x * inv_tw - 0.5f;
    int ty1 = (int) floorf(tyf), tx1 = (int) floorf(txf);
    int ty2 = ty1 + 1, tx2 = tx1 + 1;
    const float ya = tyf - (float) ty1, ya1 = 1.0f - ya, xa = txf - (float) tx1, xa1 = 1.0f - xa;
    ty1 = max(ty1, 0);
    tx1 = max(tx1, 0);
    ty2 = min(ty2, tilesY - 1);
    tx2 = min(tx2, tilesX - 1);
    const int v = src[(size_t) y * spitch + x];
    const uint8_t *p1 = lut + (size_t) (ty1 * tilesX) * 256 + v, *p2 = lut + (size_t) (ty2 * tilesX) * 256 + v;
    const float res = ((float) p1[tx1 * 256] * xa1 + (float) p1[tx2 * 256] * xa) * ya1 + ((float) p2[tx1 * 256] * xa1 + (float) p2[tx2 * 256] * xa) * ya;
    dst[(size_t) y * dpitch + x] = sat_u8(res);
}

}  // namespace

extern "C" int alva_clahe(alva_ctx *ctx, const uint8_t *d_src, size_t src_pitch, int width, int height, double clip_limit, int tiles_x,
                          int tiles_y, uint8_t *d_dst, size_t dst_pitch) {
    ALVA_ARG(ctx && d_src && d_dst && width > 0 && height > 0 && tiles_x > 0 && tiles_y > 0 && src_pitch >= (size_t) width &&
             dst_pitch >= (size_t) width);
    int ew = width, eh = height;
    if (!(width % tiles_x == 0 && height % tiles_y == 0)) {  // clahe.cpp:364-384
        ew = width + (tiles_x - (width % tiles_x));
        eh = height + (tiles_y - (height % tiles_y));
    }
    const int tw = ew / tiles_x, th = eh / tiles_y, total = tw * th;
    ALVA_ARG(tw > 0 && th > 0);
    const float lutScale = (float) (256 - 1) / (float) total;
    int clipLimit = 0;
    if (clip_limit > 0.0) {
        clipLimit = (int) (clip_limit * total / 256);
        clipLimit = clipLimit < 1 ? 1 : clipLimit;
    }
    uint8_t *d_lut = nullptr;
    int rc = alva_ctx_scratch(ctx, 6, (size_t) tiles_x * tiles_y * 256, (void **) &d_lut);
    if (rc) return rc;
    hipLaunchKernelGGL(k_clahe_lut, dim3(tiles_x * tiles_y), dim3(256), 0, ctx->stream, d_src, src_pitch, width, height, tiles_x, tw, th,
                       clipLimit, lutScale, d_lut);
    hipLaunchKernelGGL(k_clahe_apply, dim3(alva_divup(width, 64), alva_divup(height, 4)), dim3(256), 0, ctx->stream, d_src, src_pitch, width,
                       height, tiles_x, tiles_y, 1.0f / (float) tw, 1.0f / (float) th, (const uint8_t *) d_lut, d_dst, dst_pitch);
    ALVA_LAUNCH_CHECK();
    return ALVA_OK;
}
