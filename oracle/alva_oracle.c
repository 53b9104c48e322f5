/* TEST INFRASTRUCTURE ONLY -- CPU restatement of the reference hot path in plain C.
 * See oracle/README.md.  Each function cites the reference code it restates
 * (paths relative to /root/reference). */
#include "alva_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

/* cv::borderInterpolate(BORDER_REFLECT_101): src/libs/opencv/modules/core/src/copy.cpp */
static int reflect101(int p, int len) {
    if (len == 1) return 0;
    while (p < 0 || p >= len) {
        if (p < 0) p = -p;
        else p = 2 * (len - 1) - p;
    }
    return p;
}

/* a2 -- src/slam/src/system.cpp:112 cv::cvtColor(RGBA2GRAY);
 * src/libs/opencv/modules/imgproc/src/color_rgb.simd.hpp:646-664 (RGB2Gray<uchar>),
 * coefficients color.simd_helpers.hpp:16-24: R2Y=9798 (4899<<1), G2Y=19235, B2Y=3735 (1868<<1 -1?),
 * i.e. Y = (R*9798 + G*19235 + B*3735 + (1<<14)) >> 15; alpha ignored. */
void orc_rgba2gray(const uint8_t *rgba, int w, int h, uint8_t *gray) {
    for (size_t i = 0; i < (size_t) w * h; i++) {
        unsigned r = rgba[4 * i], g = rgba[4 * i + 1], b = rgba[4 * i + 2];
        gray[i] = (uint8_t) ((r * 9798u + g * 19235u + b * 3735u + 16384u) >> 15);
    }
}

/* a3 -- src/libs/opencv/modules/video/src/lkpyramid.cpp:726-822 */
int orc_pyramid_dims(int w, int h, int win, int max_level, int *dims) {
    int n = 0;
    for (int l = 0; l <= max_level; l++) {
        dims[2 * l] = w;
        dims[2 * l + 1] = h;
        n = l + 1;
        w = (w + 1) / 2;
        h = (h + 1) / 2;
        if (w <= win || h <= win) break; /* :811-816 */
    }
    return n;
}

/* pyrDown: src/libs/opencv/modules/imgproc/src/pyramids.cpp:746-900 -- 5-tap [1 4 6 4 1] in x (decimating)
 * then in y, integer, result (sum + 128) >> 8 (FixPtCast<uchar,8>, pyramids.cpp:53-58), BORDER_REFLECT_101
 * interpolated on the source size (:760-775, :820-821). */
static void pyr_down(const uint8_t *src, int sw, int sh, uint8_t *dst, int dw, int dh) {
    static const int k[5] = {1, 4, 6, 4, 1};
    for (int y = 0; y < dh; y++)
        for (int x = 0; x < dw; x++) {
            int acc = 0;
            for (int j = 0; j < 5; j++) {
                int sy = reflect101(2 * y - 2 + j, sh);
                int row = 0;
                for (int i = 0; i < 5; i++) row += k[i] * src[(size_t) sy * sw + reflect101(2 * x - 2 + i, sw)];
                acc += k[j] * row;
            }
            dst[(size_t) y * dw + x] = (uint8_t) ((acc + 128) >> 8);
        }
}

/* ScharrDerivInvoker: src/libs/opencv/modules/video/src/lkpyramid.cpp:70-151.  Vertical pass
 * t0 = 3(p[y-1]+p[y+1]) + 10 p[y], t1 = p[y+1]-p[y-1] with rows replicated as index 1 / rows-2 at
 * the edges (:83-85) -- i.e. REFLECT_101; horizontal pass Ix = t0[x+1]-t0[x-1],
 * Iy = 3(t1[x-1]+t1[x+1]) + 10 t1[x], columns likewise (:113-118); int16 interleaved (Ix,Iy). */
static void scharr(const uint8_t *g, int w, int h, int16_t *d /* w*h*2 */) {
    for (int y = 0; y < h; y++) {
        const uint8_t *r0 = g + (size_t) reflect101(y - 1, h) * w;
        const uint8_t *r1 = g + (size_t) y * w;
        const uint8_t *r2 = g + (size_t) reflect101(y + 1, h) * w;
        for (int x = 0; x < w; x++) {
            int xm = reflect101(x - 1, w), xp = reflect101(x + 1, w);
            int t0m = (r0[xm] + r2[xm]) * 3 + r1[xm] * 10, t0p = (r0[xp] + r2[xp]) * 3 + r1[xp] * 10;
            int t1m = r2[xm] - r0[xm], t1c = r2[x] - r0[x], t1p = r2[xp] - r0[xp];
            d[((size_t) y * w + x) * 2] = (int16_t) (t0p - t0m);
            d[((size_t) y * w + x) * 2 + 1] = (int16_t) ((t1m + t1p) * 3 + t1c * 10);
        }
    }
}

int orc_build_pyramid(const uint8_t *gray, int w, int h, int win, int max_level, uint8_t **gray_out, int16_t **deriv_out) {
    int dims[2 * 16];
    if (max_level > 15) return -1;
    int n = orc_pyramid_dims(w, h, win, max_level, dims);
    uint8_t *prev = (uint8_t *) malloc((size_t) w * h);
    memcpy(prev, gray, (size_t) w * h);
    int pw = w, ph = h;
    for (int l = 0; l < n; l++) {
        int lw = dims[2 * l], lh = dims[2 * l + 1];
        uint8_t *cur = prev;
        if (l > 0) {
            cur = (uint8_t *) malloc((size_t) lw * lh);
            pyr_down(prev, pw, ph, cur, lw, lh);
            free(prev);
        }
        int W = lw + 2 * win, H = lh + 2 * win;
        /* copyMakeBorder(REFLECT_101) for gray (:770, :793), CONSTANT 0 for derivatives (:806) */
        if (gray_out && gray_out[l])
            for (int y = 0; y < H; y++)
                for (int x = 0; x < W; x++)
                    gray_out[l][(size_t) y * W + x] = cur[(size_t) reflect101(y - win, lh) * lw + reflect101(x - win, lw)];
        if (deriv_out && deriv_out[l]) {
            int16_t *d = (int16_t *) malloc((size_t) lw * lh * 2 * sizeof(int16_t));
            scharr(cur, lw, lh, d);
            memset(deriv_out[l], 0, (size_t) W * H * 2 * sizeof(int16_t));
            for (int y = 0; y < lh; y++)
                memcpy(deriv_out[l] + ((size_t) (y + win) * W + win) * 2, d + (size_t) y * lw * 2, (size_t) lw * 2 * sizeof(int16_t));
            free(d);
        }
        prev = cur;
        pw = lw;
        ph = lh;
    }
    free(prev);
    return n;
}

/* a7 -- cv::norm(a,b,NORM_HAMMING): src/libs/opencv/modules/core/src/norm.cpp:99- (popcount of xor over
 * the 32 bytes); call sites src/slam/src/map_point.cpp:106,158,212. */
int orc_hamming256(const uint8_t *a, const uint8_t *b) {
    int d = 0;
    for (int i = 0; i < 32; i++) {
        unsigned v = (unsigned) (a[i] ^ b[i]);
        while (v) {
            d += (int) (v & 1u);
            v >>= 1;
        }
    }
    return d;
}

/* a7 -- cv::BFMatcher(NORM_HAMMING).match -> batchDistance k=1:
 * src/libs/opencv/modules/core/src/batch_distance.cpp:199-251; strict '<' at :238 => the lowest
 * train index wins ties. */
void orc_bf_match_hamming(const uint8_t *q, int nq, const uint8_t *t, int nt, int *idx, int *dist) {
    for (int i = 0; i < nq; i++) {
        int best = 0x7fffffff, bi = -1;
        for (int j = 0; j < nt; j++) {
            int d = orc_hamming256(q + 32 * (size_t) i, t + 32 * (size_t) j);
            if (d < best) {
                best = d;
                bi = j;
            }
        }
        idx[i] = bi;
        dist[i] = bi < 0 ? -1 : best;
    }
}

/* ------------------------------------------------------------------------------------------------
 * a6 -- FeatureExtractor::describeFeaturePoints (src/slam/src/feature_extractor.cpp:160-214) =
 * cv::ORB::create(500,1.,0)->compute() (src/libs/opencv/modules/features2d/src/orb.cpp:970-1218). */
static const int8_t orc_pattern[1024] = {
#include "orb_pattern.inc"
};

static float bits2f(uint32_t u) {
    float f;
    memcpy(&f, &u, 4);
    return f;
}

/* cvRound(float): round half to even (core/include/opencv2/core/fast_math.hpp:311) */
static int cv_round_f(float v) { return (int) lrintf(v); }

/* GaussianBlur(level, level, Size(7,7), 2, 2, BORDER_REFLECT_101) at orb.cpp:1188 on a non-isolated
 * sub-matrix whose 32-px surround is a REFLECT_101 copy => plain REFLECT_101 blur of the level.
 * 8u -> 32f row pass (RowFilter<uchar,float,RowVec_8u32f>, imgproc/src/filter.simd.hpp:468-507,2446-2488):
 *   s = k0*p0; s += k1*p1; ... ; s += k6*p6
 * 32f -> 8u column pass (SymmColumnFilter<Cast<float,uchar>,SymmColumnVec_32f8u>, :1163-1209):
 *   s = c0*r0; s += c1*(r[+1]+r[-1]); s += c2*(r[+2]+r[-2]); s += c3*(r[+3]+r[-3]); cvRound; saturate.
 * Taps: cv::getGaussianKernel(7, 2, CV_32F) (smooth.dispatch.cpp), bit patterns from the reference build. */
void orc_orb_blur(const uint8_t *gray, int w, int h, uint8_t *out) {
    const float k[7] = {bits2f(0x3d8fafb1u), bits2f(0x3e06387eu), bits2f(0x3e434a39u), bits2f(0x3e5d4ae0u),
                        bits2f(0x3e434a39u), bits2f(0x3e06387eu), bits2f(0x3d8fafb1u)};
    float *rows = (float *) malloc(sizeof(float) * (size_t) w * h);
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            volatile float s = k[0] * (float) gray[(size_t) y * w + reflect101(x - 3, w)];
            for (int i = 1; i < 7; i++) {
                volatile float p = k[i] * (float) gray[(size_t) y * w + reflect101(x - 3 + i, w)];
                s = s + p;
            }
            rows[(size_t) y * w + x] = s;
        }
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            volatile float s = k[3] * rows[(size_t) y * w + x];
            for (int j = 1; j <= 3; j++) {
                volatile float pair = rows[(size_t) reflect101(y + j, h) * w + x] + rows[(size_t) reflect101(y - j, h) * w + x];
                volatile float p = k[3 + j] * pair;
                s = s + p;
            }
            int v = cv_round_f(s);
            out[(size_t) y * w + x] = (uint8_t) (v < 0 ? 0 : v > 255 ? 255 : v);
        }
    free(rows);
}

/* computeOrbDescriptors, orb.cpp:219-284 (wta_k == 2), for one point on the blurred level. */
static void brief256(const uint8_t *blur, int w, int cx, int cy, float a, float b, uint8_t *desc) {
    for (int i = 0; i < 32; i++) {
        int val = 0;
        for (int t = 0; t < 8; t++) {
            const int8_t *p = orc_pattern + 32 * i + 4 * t;
            volatile float x0a = (float) p[0] * a, y0b = (float) p[1] * b, x0b = (float) p[0] * b, y0a = (float) p[1] * a;
            volatile float x1a = (float) p[2] * a, y1b = (float) p[3] * b, x1b = (float) p[2] * b, y1a = (float) p[3] * a;
            int ix0 = cv_round_f(x0a - y0b), iy0 = cv_round_f(x0b + y0a);
            int ix1 = cv_round_f(x1a - y1b), iy1 = cv_round_f(x1b + y1a);
            int t0 = blur[(size_t) (cy + iy0) * w + cx + ix0], t1 = blur[(size_t) (cy + iy1) * w + cx + ix1];
            val |= (t0 < t1) << t;
        }
        desc[i] = (uint8_t) val;
    }
}

void orc_describe(const uint8_t *gray, int w, int h, const float *pts, int n, uint8_t *desc, uint8_t *valid) {
    uint8_t *blur = (uint8_t *) malloc((size_t) w * h);
    orc_orb_blur(gray, w, h, blur);
    /* KeyPoint::convert => angle = -1 (core/src/types.cpp:93-101); orb.cpp:232-235 */
    float angle = -1.0f;
    angle *= (float) (3.1415926535897932384626433832795 / 180.f);
    float a = (float) cos((double) angle), b = (float) sin((double) angle);
    for (int i = 0; i < n; i++) {
        /* KeyPointsFilter::runByImageBorder(kps, size, 31): int Rect(31,31,w-62,h-62).contains(Point(pt)),
         * Point2f -> Point rounds with cvRound (features2d/src/keypoint.cpp:92-117) */
        int cx = cv_round_f(pts[2 * i]), cy = cv_round_f(pts[2 * i + 1]);
        int ok = w > 62 && h > 62 && cx >= 31 && cx < w - 31 && cy >= 31 && cy < h - 31;
        valid[i] = (uint8_t) ok;
        if (ok) brief256(blur, w, cx, cy, a, b, desc + 32 * (size_t) i);
        else memset(desc + 32 * (size_t) i, 0, 32);
    }
    free(blur);
}
