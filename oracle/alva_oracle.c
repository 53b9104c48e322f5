/* TEST INFRASTRUCTURE ONLY -- CPU restatement of the reference hot path in plain C.
 * See oracle/README.md.  Each function cites the reference code it restates
 * (paths relative to /root/reference). */
#include "alva_oracle.h"
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
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

/* ------------------------------------------------------------------------------------------------
 * a4 -- pyramidal Lucas-Kanade exactly as the reference build executes it.
 * cv::calcOpticalFlowPyrLK wrapper: src/libs/opencv/modules/video/src/lkpyramid.cpp:1239-1404;
 * per point per level: LKTrackerInvoker::operator(), lkpyramid.cpp:183-724.
 *
 * The float accumulations follow the CV_SIMD128 code path (the one compiled into oracle/_ref and into
 * the shipped wasm-simd128 artefact): for a 9-wide window each row is 8 "vector" pixels + 1 scalar
 * pixel; vector lane l accumulates pixels l and l+4 of every row in row order (lkpyramid.cpp:286-350),
 * the scalar accumulator takes pixel 8 (:440-459), and the lanes are folded with the SSE v_reduce_sum
 * tree (a0+a2)+(a1+a3) (core/hal/intrin_sse.hpp:1690-1697, 1711) at the end (:462-466).  The mismatch
 * vector b uses the dot-product pairing of :535-562 / :640-646.  Everything else is scalar float/int. */
typedef struct {
    int w, h;              /* interior size */
    const uint8_t *gray;   /* padded buffer */
    const int16_t *deriv;  /* padded buffer, 2 per pixel */
    int pw;                /* padded width = w + 2 win */
} orc_lk_level;

static float v_reduce_sum4(const float q[4]) {
    volatile float a = q[0] + q[2], b = q[1] + q[3];
    volatile float s = a + b;
    return s;
}

#define ORC_WIN 9
#define ORC_DESCALE(x, n) (((x) + (1 << ((n) -1))) >> (n))

/* one point, one level; returns nothing, mutates next/status/err like the reference */
static void lk_point_level(const orc_lk_level *I, const orc_lk_level *J, int win, int level, int maxLevel, int maxCount,
                           double epsilon, float minEigThreshold, const float *prevPts, float *nextPts, uint8_t *status,
                           float *err, int ptidx) {
    const float halfWin = (win - 1) * 0.5f;
    const float lscale = (float) (1. / (1 << level));
    float prevx = prevPts[2 * ptidx] * lscale, prevy = prevPts[2 * ptidx + 1] * lscale;
    float nextx, nexty;
    if (level == maxLevel) { /* OPTFLOW_USE_INITIAL_FLOW */
        nextx = nextPts[2 * ptidx] * lscale;
        nexty = nextPts[2 * ptidx + 1] * lscale;
    } else {
        nextx = nextPts[2 * ptidx] * 2.f;
        nexty = nextPts[2 * ptidx + 1] * 2.f;
    }
    nextPts[2 * ptidx] = nextx;
    nextPts[2 * ptidx + 1] = nexty;
    prevx -= halfWin;
    prevy -= halfWin;
    int ipx = (int) floorf(prevx), ipy = (int) floorf(prevy);
    if (ipx < -win || ipx >= I->w || ipy < -win || ipy >= I->h) {
        if (level == 0) {
            status[ptidx] = 0;
            err[ptidx] = 0;
        }
        return;
    }
    volatile float a = prevx - ipx, b = prevy - ipy;
    const float W14 = (float) (1 << 14);
    volatile float oma = 1.f - a, omb = 1.f - b;
    volatile float w00f = oma * omb, w01f = a * omb, w10f = oma * b;
    int iw00 = cv_round_f(w00f * W14), iw01 = cv_round_f(w01f * W14), iw10 = cv_round_f(w10f * W14);
    int iw11 = (1 << 14) - iw00 - iw01 - iw10;
    int16_t Iwin[ORC_WIN * ORC_WIN], dIwin[ORC_WIN * ORC_WIN * 2];
    float qA11[4] = {0, 0, 0, 0}, qA12[4] = {0, 0, 0, 0}, qA22[4] = {0, 0, 0, 0};
    volatile float iA11 = 0, iA12 = 0, iA22 = 0;
    const int stepI = I->pw, dstep = I->pw * 2;
    for (int y = 0; y < win; y++) {
        const uint8_t *src = I->gray + (size_t) (y + ipy + win) * stepI + (ipx + win);
        const int16_t *dsrc = I->deriv + (size_t) (y + ipy + win) * dstep + (size_t) (ipx + win) * 2;
        for (int x = 0; x < win; x++) {
            int ival = ORC_DESCALE(src[x] * iw00 + src[x + 1] * iw01 + src[x + stepI] * iw10 + src[x + stepI + 1] * iw11, 9);
            int ixval = ORC_DESCALE(dsrc[2 * x] * iw00 + dsrc[2 * x + 2] * iw01 + dsrc[2 * x + dstep] * iw10 + dsrc[2 * x + dstep + 2] * iw11, 14);
            int iyval = ORC_DESCALE(dsrc[2 * x + 1] * iw00 + dsrc[2 * x + 3] * iw01 + dsrc[2 * x + dstep + 1] * iw10 + dsrc[2 * x + dstep + 3] * iw11, 14);
            Iwin[y * win + x] = (int16_t) ival;
            dIwin[(y * win + x) * 2] = (int16_t) ixval;
            dIwin[(y * win + x) * 2 + 1] = (int16_t) iyval;
            if (x < 8) {
                int l = x & 3;
                float fx = (float) (int16_t) ixval, fy = (float) (int16_t) iyval;
                volatile float p22 = fy * fy, p12 = fx * fy, p11 = fx * fx;
                volatile float s22 = p22 + qA22[l], s12 = p12 + qA12[l], s11 = p11 + qA11[l];
                qA22[l] = s22;
                qA12[l] = s12;
                qA11[l] = s11;
            } else {
                iA11 += (float) (ixval * ixval);
                iA12 += (float) (ixval * iyval);
                iA22 += (float) (iyval * iyval);
            }
        }
    }
    iA11 += v_reduce_sum4(qA11);
    iA12 += v_reduce_sum4(qA12);
    iA22 += v_reduce_sum4(qA22);
    const float FLT_SCALE = 1.f / (1 << 20);
    volatile float A11 = iA11 * FLT_SCALE, A12 = iA12 * FLT_SCALE, A22 = iA22 * FLT_SCALE;
    volatile float p1 = A11 * A22, p2 = A12 * A12;
    volatile float D = p1 - p2;
    volatile float dA = A11 - A22;
    volatile float dA2 = dA * dA, fA = 4.f * A12;
    volatile float fA2 = fA * A12;
    volatile float disc = dA2 + fA2;
    volatile float sq = sqrtf(disc);
    volatile float tr = A22 + A11;
    volatile float num = tr - sq;
    float minEig = num / (float) (2 * win * win);
    err[ptidx] = minEig; /* OPTFLOW_LK_GET_MIN_EIGENVALS */
    if (minEig < minEigThreshold || D < 1.1920928955078125e-07f) {
        if (level == 0) status[ptidx] = 0;
        return;
    }
    D = 1.f / D;
    nextx -= halfWin;
    nexty -= halfWin;
    float pdx = 0, pdy = 0;
    const int stepJ = J->pw;
    for (int j = 0; j < maxCount; j++) {
        int inx = (int) floorf(nextx), iny = (int) floorf(nexty);
        if (inx < -win || inx >= J->w || iny < -win || iny >= J->h) {
            if (level == 0) status[ptidx] = 0;
            break;
        }
        a = nextx - inx;
        b = nexty - iny;
        oma = 1.f - a;
        omb = 1.f - b;
        w00f = oma * omb;
        w01f = a * omb;
        w10f = oma * b;
        iw00 = cv_round_f(w00f * W14);
        iw01 = cv_round_f(w01f * W14);
        iw10 = cv_round_f(w10f * W14);
        iw11 = (1 << 14) - iw00 - iw01 - iw10;
        float qb0[4] = {0, 0, 0, 0}, qb1[4] = {0, 0, 0, 0};
        volatile float ib1 = 0, ib2 = 0;
        for (int y = 0; y < win; y++) {
            const uint8_t *Jp = J->gray + (size_t) (y + iny + win) * stepJ + (inx + win);
            int diff[ORC_WIN];
            for (int x = 0; x < win; x++)
                diff[x] = (int16_t) (ORC_DESCALE(Jp[x] * iw00 + Jp[x + 1] * iw01 + Jp[x + stepJ] * iw10 + Jp[x + stepJ + 1] * iw11, 9) - Iwin[y * win + x]);
            const int16_t *dI = dIwin + y * win * 2;
#define IX(i) ((int) dI[2 * (i)])
#define IY(i) ((int) dI[2 * (i) + 1])
            /* v_dotprod pairing, lkpyramid.cpp:553-562 */
            volatile float t;
            t = (float) (diff[0] * IX(0) + diff[4] * IX(4)); qb0[0] = qb0[0] + t;
            t = (float) (diff[0] * IY(0) + diff[4] * IY(4)); qb0[1] = qb0[1] + t;
            t = (float) (diff[1] * IX(1) + diff[5] * IX(5)); qb0[2] = qb0[2] + t;
            t = (float) (diff[1] * IY(1) + diff[5] * IY(5)); qb0[3] = qb0[3] + t;
            t = (float) (diff[2] * IX(2) + diff[6] * IX(6)); qb1[0] = qb1[0] + t;
            t = (float) (diff[2] * IY(2) + diff[6] * IY(6)); qb1[1] = qb1[1] + t;
            t = (float) (diff[3] * IX(3) + diff[7] * IX(7)); qb1[2] = qb1[2] + t;
            t = (float) (diff[3] * IY(3) + diff[7] * IY(7)); qb1[3] = qb1[3] + t;
            ib1 += (float) (diff[8] * IX(8));
            ib2 += (float) (diff[8] * IY(8));
#undef IX
#undef IY
        }
        /* :640-646: s = qb0 + qb1; qf0 = [s0, s2, 0, 0], qf1 = [s1, s3, 0, 0]; reduce = (q0+q2)+(q1+q3) */
        volatile float s0 = qb0[0] + qb1[0], s1 = qb0[1] + qb1[1], s2 = qb0[2] + qb1[2], s3 = qb0[3] + qb1[3];
        volatile float z = 0.f;
        volatile float r1a = s0 + z, r1b = s2 + z, r2a = s1 + z, r2b = s3 + z;
        volatile float r1 = r1a + r1b, r2 = r2a + r2b;
        ib1 += r1;
        ib2 += r2;
        volatile float b1 = ib1 * FLT_SCALE, b2 = ib2 * FLT_SCALE;
        volatile float m1 = A12 * b2, m2 = A22 * b1, m3 = A12 * b1, m4 = A11 * b2;
        volatile float n1 = m1 - m2, n2 = m3 - m4;
        float dx = n1 * D, dy = n2 * D;
        nextx += dx;
        nexty += dy;
        nextPts[2 * ptidx] = nextx + halfWin;
        nextPts[2 * ptidx + 1] = nexty + halfWin;
        if ((double) dx * dx + (double) dy * dy <= epsilon) break;
        if (j > 0) {
            volatile float sx = dx + pdx, sy = dy + pdy;
            if (fabs((double) sx) < 0.01 && fabs((double) sy) < 0.01) {
                volatile float hx = dx * 0.5f, hy = dy * 0.5f;
                nextPts[2 * ptidx] -= hx;
                nextPts[2 * ptidx + 1] -= hy;
                break;
            }
        }
        pdx = dx;
        pdy = dy;
    }
}

/* levels: arrays of length >= numLevels+1 describing both pyramids (padded buffers as produced by
 * orc_build_pyramid).  flags = USE_INITIAL_FLOW | LK_GET_MIN_EIGENVALS; minEigThreshold 1e-4 default. */
static void lk_pyr(const orc_lk_level *P, const orc_lk_level *N, int nbuilt, int win, int numLevels, int maxIters, float eps,
                   const float *pts, float *next, uint8_t *status, float *err, int n) {
    int maxLevel = numLevels;
    if (nbuilt - 1 < maxLevel) maxLevel = nbuilt - 1; /* :1318-1319 */
    int maxCount = maxIters < 0 ? 0 : (maxIters > 100 ? 100 : maxIters); /* :1358-1365 */
    double epsilon = (double) eps;
    epsilon = epsilon < 0 ? 0 : (epsilon > 10 ? 10 : epsilon);
    epsilon *= epsilon;
    for (int i = 0; i < n; i++) {
        status[i] = 1;
        err[i] = 0; /* the reference leaves it uninitialised when never written; 0 is what a fresh Mat holds in practice */
    }
    for (int level = maxLevel; level >= 0; level--)
        for (int i = 0; i < n; i++)
            lk_point_level(&P[level], &N[level], win, level, maxLevel, maxCount, epsilon, 1e-4f, pts, next, status, err, i);
}

static int make_levels(const uint8_t *gray, int w, int h, int win, int maxLevel, orc_lk_level *L, uint8_t **gb, int16_t **db) {
    int dims[32];
    int n = orc_pyramid_dims(w, h, win, maxLevel, dims);
    for (int l = 0; l < 16; l++) {
        gb[l] = NULL;
        db[l] = NULL;
    }
    for (int l = 0; l < n; l++) {
        size_t W = dims[2 * l] + 2 * win, H = dims[2 * l + 1] + 2 * win;
        gb[l] = (uint8_t *) malloc(W * H);
        db[l] = (int16_t *) malloc(W * H * 2 * sizeof(int16_t));
    }
    orc_build_pyramid(gray, w, h, win, maxLevel, gb, db);
    for (int l = 0; l < n; l++) {
        L[l].w = dims[2 * l];
        L[l].h = dims[2 * l + 1];
        L[l].gray = gb[l];
        L[l].deriv = db[l];
        L[l].pw = dims[2 * l] + 2 * win;
    }
    return n;
}

int orc_lk(const uint8_t *prevGray, const uint8_t *nextGray, int w, int h, int win, int pyrLevelsBuilt, int numLevels,
           int maxIters, float eps, const float *pts, float *next, uint8_t *status, float *err, int n) {
    if (win != ORC_WIN) return -1;
    orc_lk_level P[16], N[16];
    uint8_t *pg[16], *ng[16];
    int16_t *pd[16], *nd[16];
    int nb = make_levels(prevGray, w, h, win, pyrLevelsBuilt, P, pg, pd);
    make_levels(nextGray, w, h, win, pyrLevelsBuilt, N, ng, nd);
    lk_pyr(P, N, nb, win, numLevels, maxIters, eps, pts, next, status, err, n);
    for (int l = 0; l < 16; l++) {
        free(pg[l]); free(ng[l]); free(pd[l]); free(nd[l]);
    }
    return 0;
}

/* FeatureTracker::fbKltTracking -- src/slam/src/feature_tracker.cpp:5-111 */
int orc_fbklt(const uint8_t *prevGray, const uint8_t *currGray, int w, int h, int win, int pyrLevelsBuilt, int numLevels,
              float errThresh, float fbDist, int maxIters, float eps, const float *pts, float *prior, uint8_t *status, int n) {
    if (win != ORC_WIN) return -1;
    if (n == 0) return 0;
    orc_lk_level P[16], C[16];
    uint8_t *pg[16], *cg[16];
    int16_t *pd[16], *cd[16];
    int nb = make_levels(prevGray, w, h, win, pyrLevelsBuilt, P, pg, pd);
    make_levels(currGray, w, h, win, pyrLevelsBuilt, C, cg, cd);
    if (nb < numLevels + 1) numLevels = nb - 1; /* :19-22 */
    uint8_t *st = (uint8_t *) malloc((size_t) n);
    float *er = (float *) malloc(sizeof(float) * (size_t) n);
    lk_pyr(P, C, nb, win, numLevels, maxIters, eps, pts, prior, st, er, n); /* forward, :36-39 */
    float *newk = (float *) malloc(sizeof(float) * 2 * (size_t) n), *back = (float *) malloc(sizeof(float) * 2 * (size_t) n);
    int *index = (int *) malloc(sizeof(int) * (size_t) n);
    int m = 0;
    for (int i = 0; i < n; i++) {
        status[i] = 0;
        if (!st[i]) continue;
        if (er[i] > errThresh) continue; /* :56 */
        float x = prior[2 * i], y = prior[2 * i + 1]; /* inBorder, :113-119, BORDER_SIZE 1 */
        if (!(1.0f <= x && x < (float) w - 1.0f && 1.0f <= y && y < (float) h - 1.0f)) continue;
        newk[2 * m] = x;
        newk[2 * m + 1] = y;
        back[2 * m] = pts[2 * i];
        back[2 * m + 1] = pts[2 * i + 1];
        status[i] = 1;
        index[m++] = i;
    }
    if (m > 0) {
        uint8_t *st2 = (uint8_t *) malloc((size_t) m);
        float *er2 = (float *) malloc(sizeof(float) * (size_t) m);
        lk_pyr(C, P, nb, win, 0, maxIters, eps, newk, back, st2, er2, m); /* backward, level 0 only, :84-87 */
        for (int k = 0; k < m; k++) {
            int i = index[k];
            if (!st2[k]) {
                status[i] = 0;
                continue;
            }
            volatile float dx = pts[2 * i] - back[2 * k], dy = pts[2 * i + 1] - back[2 * k + 1];
            double nrm = sqrt((double) dx * dx + (double) dy * dy); /* cv::norm(Point2f) */
            if (nrm > (double) fbDist) status[i] = 0;
        }
        free(st2);
        free(er2);
    }
    free(st); free(er); free(newk); free(back); free(index);
    for (int l = 0; l < 16; l++) {
        free(pg[l]); free(cg[l]); free(pd[l]); free(cd[l]);
    }
    return 0;
}

/* ------------------------------------------------------------------------------------------------
 * a8 -- MultiViewGeometry::p3pRansac (src/slam/src/multi_view_geometry.cpp:24-127):
 * opengv::sac::Lmeds<AbsolutePoseSacProblem(KNEIP)> (src/libs/opengv/include/opengv/sac/implementation/Lmeds.hpp:43-195).
 * FP64 throughout.  Complex arithmetic follows libstdc++'s std::complex<double> semantics where they
 * matter (pow via log/polar, principal branches). */
typedef struct { double re, im; } cplx;
static cplx c_(double r, double i) { cplx z = {r, i}; return z; }
static cplx cadd(cplx a, cplx b) { return c_(a.re + b.re, a.im + b.im); }
static cplx csub(cplx a, cplx b) { return c_(a.re - b.re, a.im - b.im); }
static cplx cscale(cplx a, double s) { return c_(a.re * s, a.im * s); }
static cplx cmul(cplx a, cplx b) { return c_(a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re); }
static cplx cdivc(cplx a, cplx b) {
    double den = b.re * b.re + b.im * b.im;
    return c_((a.re * b.re + a.im * b.im) / den, (a.im * b.re - a.re * b.im) / den);
}
static cplx csqrt_(cplx z) { /* principal square root */
    double m = hypot(z.re, z.im);
    if (m == 0) return c_(0, z.im);
    double t;
    if (z.re >= 0) {
        t = sqrt(0.5 * (m + z.re));
        return c_(t, z.im / (2 * t));
    }
    t = sqrt(0.5 * (m - z.re));
    return c_(fabs(z.im) / (2 * t), z.im < 0 ? -t : t);
}
/* std::pow(complex<double>, double), libstdc++ <complex>: real fast path for positive reals, else
 * polar(exp(y*log|x|), y*arg(x)) */
static cplx cpow_(cplx x, double y) {
    if (x.im == 0 && x.re > 0) return c_(pow(x.re, y), 0);
    double lr = log(hypot(x.re, x.im)), th = atan2(x.im, x.re);
    double r = exp(y * lr), a = y * th;
    return c_(r * cos(a), r * sin(a));
}

/* opengv::math::o4_roots -- src/libs/opengv/src/math/roots.cpp:88-135 (Ferrari; real parts returned) */
static void o4_roots(const double f[5], double roots[4]) {
    double A = f[0], B = f[1], C = f[2], D = f[3], E = f[4];
    double A2 = A * A, B2 = B * B, A3 = A2 * A, B3 = B2 * B, A4 = A3 * A, B4 = B3 * B;
    double alpha = -3 * B2 / (8 * A2) + C / A;
    double beta = B3 / (8 * A3) - B * C / (2 * A2) + D / A;
    double gamma = -3 * B4 / (256 * A4) + B2 * C / (16 * A3) - B * D / (4 * A2) + E / A;
    double alpha2 = alpha * alpha, alpha3 = alpha2 * alpha;
    cplx P = c_(-alpha2 / 12 - gamma, 0);
    cplx Q = c_(-alpha3 / 108 + alpha * gamma / 3 - pow(beta, 2) / 8, 0);
    cplx R = cadd(cscale(Q, -0.5), csqrt_(cadd(cscale(cpow_(Q, 2.0), 0.25), cscale(cpow_(P, 3.0), 1.0 / 27.0))));
    cplx U = cpow_(R, 1.0 / 3.0);
    cplx y;
    if (U.re == 0) y = csub(c_(-5.0 * alpha / 6.0, 0), cpow_(Q, 1.0 / 3.0));
    else y = cadd(csub(c_(-5.0 * alpha / 6.0, 0), cdivc(P, cscale(U, 3.0))), U);
    cplx w = csqrt_(cadd(c_(alpha, 0), cscale(y, 2.0)));
    cplx base = cadd(c_(3.0 * alpha, 0), cscale(y, 2.0));
    cplx bw = cdivc(c_(2.0 * beta, 0), w);
    cplx s1 = csqrt_(cscale(cadd(base, bw), -1.0)), s2 = csqrt_(cscale(csub(base, bw), -1.0));
    double off = -B / (4.0 * A);
    roots[0] = off + 0.5 * (w.re + s1.re);
    roots[1] = off + 0.5 * (w.re - s1.re);
    roots[2] = off + 0.5 * (-w.re + s2.re);
    roots[3] = off + 0.5 * (-w.re - s2.re);
}

static void v3sub(const double *a, const double *b, double *o) { o[0] = a[0] - b[0]; o[1] = a[1] - b[1]; o[2] = a[2] - b[2]; }
static void v3cross(const double *a, const double *b, double *o) {
    o[0] = a[1] * b[2] - a[2] * b[1]; o[1] = a[2] * b[0] - a[0] * b[2]; o[2] = a[0] * b[1] - a[1] * b[0];
}
static double v3dot(const double *a, const double *b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
static double v3norm(const double *a) { return sqrt(v3dot(a, a)); }
static void m3v(const double M[9], const double *v, double *o) { /* row-major */
    for (int r = 0; r < 3; r++) o[r] = M[3 * r] * v[0] + M[3 * r + 1] * v[1] + M[3 * r + 2] * v[2];
}
static void m3tv(const double M[9], const double *v, double *o) {
    for (int c = 0; c < 3; c++) o[c] = M[c] * v[0] + M[3 + c] * v[1] + M[6 + c] * v[2];
}

/* opengv::absolute_pose::modules::p3p_kneip_main -- src/libs/opengv/src/absolute_pose/modules/main.cpp:50-205.
 * f: 3 bearings, p: 3 world points.  Writes up to 4 solutions {R row-major (cam->world), C}. Returns count (0 or 4). */
static int p3p_kneip(const double f[3][3], const double p[3][3], double sol[4][12]) {
    double P1[3], P2[3], P3[3], t1[3], t2[3], cr[3];
    memcpy(P1, p[0], 24); memcpy(P2, p[1], 24); memcpy(P3, p[2], 24);
    v3sub(P2, P1, t1); v3sub(P3, P1, t2); v3cross(t1, t2, cr);
    if (v3norm(cr) == 0) return 0;
    double f1[3], f2[3], f3[3], e1[3], e2[3], e3[3], T[9];
    memcpy(f1, f[0], 24); memcpy(f2, f[1], 24); memcpy(f3, f[2], 24);
    for (int pass = 0; pass < 2; pass++) {
        memcpy(e1, f1, 24);
        v3cross(f1, f2, e3);
        double n = v3norm(e3);
        e3[0] /= n; e3[1] /= n; e3[2] /= n;
        v3cross(e3, e1, e2);
        memcpy(T, e1, 24); memcpy(T + 3, e2, 24); memcpy(T + 6, e3, 24);
        double tf[3];
        m3v(T, f[2], tf);
        memcpy(f3, tf, 24);
        if (pass == 0 && f3[2] > 0) {
            memcpy(f1, f[1], 24); memcpy(f2, f[0], 24);
            memcpy(P1, p[1], 24); memcpy(P2, p[0], 24); memcpy(P3, p[2], 24);
            continue;
        }
        break;
    }
    double n1[3], n2[3], n3[3], d[3], N[9];
    v3sub(P2, P1, n1);
    double nn = v3norm(n1);
    n1[0] /= nn; n1[1] /= nn; n1[2] /= nn;
    v3sub(P3, P1, d);
    v3cross(n1, d, n3);
    nn = v3norm(n3);
    n3[0] /= nn; n3[1] /= nn; n3[2] /= nn;
    v3cross(n3, n1, n2);
    memcpy(N, n1, 24); memcpy(N + 3, n2, 24); memcpy(N + 6, n3, 24);
    double P3n[3];
    m3v(N, d, P3n);
    double d_12 = v3norm(t1);
    double f_1 = f3[0] / f3[2], f_2 = f3[1] / f3[2], p_1 = P3n[0], p_2 = P3n[1];
    double cos_beta = v3dot(f1, f2);
    double b = 1 / (1 - pow(cos_beta, 2)) - 1;
    b = cos_beta < 0 ? -sqrt(b) : sqrt(b);
    double f_1_pw2 = pow(f_1, 2), f_2_pw2 = pow(f_2, 2), p_1_pw2 = pow(p_1, 2), p_1_pw3 = p_1_pw2 * p_1, p_1_pw4 = p_1_pw3 * p_1;
    double p_2_pw2 = pow(p_2, 2), p_2_pw3 = p_2_pw2 * p_2, p_2_pw4 = p_2_pw3 * p_2, d_12_pw2 = pow(d_12, 2), b_pw2 = pow(b, 2);
    double fac[5];
    fac[0] = -f_2_pw2 * p_2_pw4 - p_2_pw4 * f_1_pw2 - p_2_pw4;
    fac[1] = 2 * p_2_pw3 * d_12 * b + 2 * f_2_pw2 * p_2_pw3 * d_12 * b - 2 * f_2 * p_2_pw3 * f_1 * d_12;
    fac[2] = -f_2_pw2 * p_2_pw2 * p_1_pw2 - f_2_pw2 * p_2_pw2 * d_12_pw2 * b_pw2 - f_2_pw2 * p_2_pw2 * d_12_pw2 + f_2_pw2 * p_2_pw4
             + p_2_pw4 * f_1_pw2 + 2 * p_1 * p_2_pw2 * d_12 + 2 * f_1 * f_2 * p_1 * p_2_pw2 * d_12 * b - p_2_pw2 * p_1_pw2 * f_1_pw2
             + 2 * p_1 * p_2_pw2 * f_2_pw2 * d_12 - p_2_pw2 * d_12_pw2 * b_pw2 - 2 * p_1_pw2 * p_2_pw2;
    fac[3] = 2 * p_1_pw2 * p_2 * d_12 * b + 2 * f_2 * p_2_pw3 * f_1 * d_12 - 2 * f_2_pw2 * p_2_pw3 * d_12 * b - 2 * p_1 * p_2 * d_12_pw2 * b;
    fac[4] = -2 * f_2 * p_2_pw2 * f_1 * p_1 * d_12 * b + f_2_pw2 * p_2_pw2 * d_12_pw2 + 2 * p_1_pw3 * d_12 - p_1_pw2 * d_12_pw2
             + f_2_pw2 * p_2_pw2 * p_1_pw2 - p_1_pw4 - 2 * f_2_pw2 * p_2_pw2 * p_1 * d_12 + p_2_pw2 * f_1_pw2 * p_1_pw2
             + f_2_pw2 * p_2_pw2 * d_12_pw2 * b_pw2;
    double roots[4];
    o4_roots(fac, roots);
    for (int i = 0; i < 4; i++) {
        double cot_alpha = (-f_1 * p_1 / f_2 - roots[i] * p_2 + d_12 * b) / (-f_1 * roots[i] * p_2 / f_2 + p_1 - d_12);
        double cos_theta = roots[i], sin_theta = sqrt(1 - pow(roots[i], 2));
        double sin_alpha = sqrt(1 / (pow(cot_alpha, 2) + 1)), cos_alpha = sqrt(1 - pow(sin_alpha, 2));
        if (cot_alpha < 0) cos_alpha = -cos_alpha;
        double Cv[3] = {d_12 * cos_alpha * (sin_alpha * b + cos_alpha), cos_theta * d_12 * sin_alpha * (sin_alpha * b + cos_alpha),
                        sin_theta * d_12 * sin_alpha * (sin_alpha * b + cos_alpha)};
        double NtC[3];
        m3tv(N, Cv, NtC);
        double R[9] = {-cos_alpha, -sin_alpha * cos_theta, -sin_alpha * sin_theta, sin_alpha, -cos_alpha * cos_theta,
                       -cos_alpha * sin_theta, 0.0, -sin_theta, cos_theta};
        /* R = N^T * R^T * T */
        double RtT[9], out[9];
        for (int r = 0; r < 3; r++)
            for (int c = 0; c < 3; c++) RtT[3 * r + c] = R[r] * T[c] + R[3 + r] * T[3 + c] + R[6 + r] * T[6 + c];
        for (int r = 0; r < 3; r++)
            for (int c = 0; c < 3; c++) out[3 * r + c] = N[r] * RtT[c] + N[3 + r] * RtT[3 + c] + N[6 + r] * RtT[6 + c];
        memcpy(sol[i], out, 72);
        sol[i][9] = P1[0] + NtC[0]; sol[i][10] = P1[1] + NtC[1]; sol[i][11] = P1[2] + NtC[2];
    }
    return 4;
}

/* score of one point under model {R cam->world, t}: 1 - f . normalize(R^T (p - t))
 * (AbsolutePoseSacProblem::getSelectedDistancesToModel, AbsolutePoseSacProblem.cpp:165-199) */
static double p3p_score(const double m[12], const double *wp, const double *bv) {
    double d[3] = {wp[0] - m[9], wp[1] - m[10], wp[2] - m[11]}, r[3];
    /* the reference forms inverse = [R^T | -R^T t] and applies it to the homogeneous point */
    double Rt_t[3];
    m3tv(m, m + 9, Rt_t);
    (void) d;
    for (int c = 0; c < 3; c++) r[c] = (m[c] * wp[0] + m[3 + c] * wp[1] + m[6 + c] * wp[2]) + (-Rt_t[c]) * 1.0;
    double n = v3norm(r);
    r[0] /= n; r[1] /= n; r[2] /= n;
    return 1.0 - v3dot(r, bv);
}

/* computeModelCoefficients, AbsolutePoseSacProblem.cpp:35-163 (KNEIP): P3P on samples 0..2, pick by 4th */
static int p3p_model(const double *bv, const double *wpt, const int s[4], double model[12]) {
    double f[3][3], p[3][3], sol[4][12];
    for (int k = 0; k < 3; k++) {
        memcpy(f[k], bv + 3 * s[k], 24);
        memcpy(p[k], wpt + 3 * s[k], 24);
    }
    int ns = p3p_kneip(f, p, sol);
    double minScore = 1000000.0;
    int minIndex = -1;
    for (int i = 0; i < ns; i++) {
        double sc = p3p_score(sol[i], wpt + 3 * s[3], bv + 3 * s[3]);
        if (sc < minScore) {
            minScore = sc;
            minIndex = i;
        }
    }
    if (minIndex < 0) return 0;
    memcpy(model, sol[minIndex], 96);
    return 1;
}

/* std::mt19937 */
typedef struct { uint32_t mt[624]; int idx; } orc_mt;
static void mt_seed(orc_mt *g, uint32_t s) {
    g->mt[0] = s;
    for (int i = 1; i < 624; i++) g->mt[i] = 1812433253u * (g->mt[i - 1] ^ (g->mt[i - 1] >> 30)) + (uint32_t) i;
    g->idx = 624;
}
static uint32_t mt_next(orc_mt *g) {
    if (g->idx >= 624) {
        for (int i = 0; i < 624; i++) {
            uint32_t y = (g->mt[i] & 0x80000000u) | (g->mt[(i + 1) % 624] & 0x7fffffffu);
            g->mt[i] = g->mt[(i + 397) % 624] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
        }
        g->idx = 0;
    }
    uint32_t y = g->mt[g->idx++];
    y ^= y >> 11;
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= y >> 18;
    return y;
}
/* rnd() = std::uniform_int_distribution<int>(0, INT_MAX)(mt19937): with libstdc++ >= 10 (Lemire's
 * nearly-divisionless path for a 32-bit generator and range 2^31) this is exactly mt() >> 1.
 * SampleConsensusProblem.hpp:40-49 (seed 12345u when !randomSeed), :65-84 (prefix Fisher-Yates). */
static int cmp_double(const void *a, const void *b) {
    double x = *(const double *) a, y = *(const double *) b;
    return (x > y) - (x < y);
}

int orc_p3p_lmeds(const double *bv, const double *wpt, int n, int maxIterations, float errorThreshold, uint32_t seed, float fx,
                  float fy, double *R_out, double *t_out, int *outliers, int *nOutliers) {
    *nOutliers = 0;
    if (n < 4) return 0; /* multi_view_geometry.cpp:41-44 */
    float focal = fx + fy; /* :72-76 */
    focal /= 2.f;
    double threshold = 1.0 - cos(atan((double) (errorThreshold / focal)));
    orc_mt g;
    mt_seed(&g, seed);
    int *shuf = (int *) malloc(sizeof(int) * (size_t) n);
    for (int i = 0; i < n; i++) shuf[i] = i;
    double *dist = (double *) malloc(sizeof(double) * (size_t) n);
    double best = 1.7976931348623157e308, bestModel[12];
    int haveModel = 0, iterations = 0;
    unsigned skipped = 0, maxSkip = (unsigned) maxIterations * 10u;
    while (iterations < maxIterations && skipped < maxSkip) {
        int s[4];
        for (int i = 0; i < 4; i++) { /* drawIndexSample */
            int r = (int) (mt_next(&g) >> 1);
            int j = i + (int) ((unsigned) r % (unsigned) (n - i));
            int tmp = shuf[i]; shuf[i] = shuf[j]; shuf[j] = tmp;
        }
        memcpy(s, shuf, sizeof(s));
        double model[12];
        if (!p3p_model(bv, wpt, s, model)) {
            skipped++;
            continue;
        }
        for (int i = 0; i < n; i++) {
            double d = p3p_score(model, wpt + 3 * i, bv + 3 * i);
            if (d < 0) d = 0;
            dist[i] = d * d;
        }
        qsort(dist, (size_t) n, sizeof(double), cmp_double);
        int mid = n / 2;
        double pen = (n % 2 == 0) ? (dist[mid - 1] + dist[mid]) / 2 : dist[mid];
        if (pen < best) {
            best = pen;
            memcpy(bestModel, model, sizeof(model));
            haveModel = 1;
        }
        iterations++;
    }
    int ok = 0;
    if (haveModel) {
        uint8_t *inl = (uint8_t *) calloc((size_t) n, 1);
        int ninl = 0;
        for (int i = 0; i < n; i++)
            if (p3p_score(bestModel, wpt + 3 * i, bv + 3 * i) <= threshold) {
                inl[i] = 1;
                ninl++;
            }
        /* :82-91: >= 5 inliers and Sophus::isOrthogonal(R): ||R R^T - I||_F < 1e-10 */
        double e = 0;
        for (int r = 0; r < 3; r++)
            for (int c = 0; c < 3; c++) {
                double v = bestModel[3 * r] * bestModel[3 * c] + bestModel[3 * r + 1] * bestModel[3 * c + 1] + bestModel[3 * r + 2] * bestModel[3 * c + 2] - (r == c);
                e += v * v;
            }
        if (ninl >= 5 && sqrt(e) < 1e-10) {
            ok = 1;
            memcpy(R_out, bestModel, 72);
            memcpy(t_out, bestModel + 9, 24);
            for (int i = 0; i < n; i++)
                if (!inl[i]) outliers[(*nOutliers)++] = i;
        }
        free(inl);
    }
    free(shuf);
    free(dist);
    return ok;
}

int orc_p3p_draw_samples(int n, int count, uint32_t seed, int *samples) {
    orc_mt g;
    mt_seed(&g, seed);
    int *shuf = (int *) malloc(sizeof(int) * (size_t) n);
    for (int i = 0; i < n; i++) shuf[i] = i;
    for (int k = 0; k < count; k++) {
        for (int i = 0; i < 4; i++) {
            int r = (int) (mt_next(&g) >> 1);
            int j = i + (int) ((unsigned) r % (unsigned) (n - i));
            int tmp = shuf[i]; shuf[i] = shuf[j]; shuf[j] = tmp;
        }
        memcpy(samples + 4 * k, shuf, 4 * sizeof(int));
    }
    free(shuf);
    return 0;
}

/* ================================================================================================
 * a9 / a10-a13 -- Ceres-style Levenberg-Marquardt, restated.
 *
 * Control flow: src/libs/ceres-solver/internal/ceres/trust_region_minimizer.cc:67-136 (loop),
 * :244-311 (evaluate, Jacobi scaling 1/(1+sqrt(||col||^2)) computed ONCE at iteration 0), :377-451
 * (step, model cost change), :744-829 (tolerances, step quality, accept);
 * levenberg_marquardt_strategy.cc:66-160 (D = sqrt(clamp(diag JtJ,1e-6,1e32)/radius), radius update);
 * robust loss residual_block.cc + corrector.cc:41-110 + loss_function.cc:48-62 (Huber => scale r and J
 * by sqrt(rho')); defaults from include/ceres/solver.h (radius0 1e4, max 1e16, min 1e-32,
 * min_relative_decrease 1e-3, gradient_tolerance 1e-10, parameter_tolerance 1e-8).
 * The linear solve is done on the normal equations (the reference uses DENSE_QR for PnP and
 * SPARSE_SCHUR for BA; same minimiser of ||J y - r||^2 + ||D y||^2 up to FP64 rounding). */
typedef struct { double q[4]; double t[3]; double R[9]; } orc_se3; /* q = x,y,z,w ; R row-major world<-cam */

static void quat_to_R(const double q[4], double R[9]) { /* Eigen::Quaternion::toRotationMatrix */
    double x = q[0], y = q[1], z = q[2], w = q[3];
    double tx = 2 * x, ty = 2 * y, tz = 2 * z, twx = tx * w, twy = ty * w, twz = tz * w, txx = tx * x, txy = ty * x, txz = tz * x,
           tyy = ty * y, tyz = tz * y, tzz = tz * z;
    R[0] = 1 - (tyy + tzz); R[1] = txy - twz; R[2] = txz + twy;
    R[3] = txy + twz; R[4] = 1 - (txx + tzz); R[5] = tyz - twx;
    R[6] = txz - twy; R[7] = tyz + twx; R[8] = 1 - (txx + tyy);
}
static void se3_from_pose7(const double *p, orc_se3 *T) { /* Sophus::SE3d(q, t): normalises q */
    double n = sqrt(p[3] * p[3] + p[4] * p[4] + p[5] * p[5] + p[6] * p[6]);
    for (int i = 0; i < 4; i++) T->q[i] = p[3 + i] / n;
    memcpy(T->t, p, 24);
    quat_to_R(T->q, T->R);
}
/* SE3Parameterization::Plus: T <- Exp(delta) * T, delta = (upsilon, omega)
 * (src/slam/src/ceres_parametrization.hpp:224-240; Sophus se3.hpp:763-784, so3.hpp:585-621, :329-343) */
static void se3_plus(const double *x7, const double *d6, double *out7) {
    orc_se3 T;
    se3_from_pose7(x7, &T);
    const double *u = d6, *w = d6 + 3;
    double th2 = w[0] * w[0] + w[1] * w[1] + w[2] * w[2], theta, imag, real;
    if (th2 < 1e-10 * 1e-10) {
        theta = 0;
        double th4 = th2 * th2;
        imag = 0.5 - (1.0 / 48.0) * th2 + (1.0 / 3840.0) * th4;
        real = 1 - (1.0 / 8.0) * th2 + (1.0 / 384.0) * th4;
    } else {
        theta = sqrt(th2);
        double h = 0.5 * theta;
        imag = sin(h) / theta;
        real = cos(h);
    }
    double qd[4] = {imag * w[0], imag * w[1], imag * w[2], real}, Rd[9];
    quat_to_R(qd, Rd);
    double V[9];
    if (theta < 1e-10) memcpy(V, Rd, sizeof(V));
    else {
        double O[9] = {0, -w[2], w[1], w[2], 0, -w[0], -w[1], w[0], 0}, O2[9];
        for (int r = 0; r < 3; r++)
            for (int c = 0; c < 3; c++) O2[3 * r + c] = O[3 * r] * O[c] + O[3 * r + 1] * O[3 + c] + O[3 * r + 2] * O[6 + c];
        double a = (1 - cos(theta)) / th2, b = (theta - sin(theta)) / (th2 * theta);
        for (int i = 0; i < 9; i++) V[i] = (i % 4 == 0 ? 1.0 : 0.0) + a * O[i] + b * O2[i];
    }
    double td[3];
    m3v(V, u, td);
    /* product: q = qd * qT (normalised), t = td + Rd * tT */
    const double *a4 = qd, *b4 = T.q;
    double qn[4] = {a4[3] * b4[0] + a4[0] * b4[3] + a4[1] * b4[2] - a4[2] * b4[1],
                    a4[3] * b4[1] + a4[1] * b4[3] + a4[2] * b4[0] - a4[0] * b4[2],
                    a4[3] * b4[2] + a4[2] * b4[3] + a4[0] * b4[1] - a4[1] * b4[0],
                    a4[3] * b4[3] - a4[0] * b4[0] - a4[1] * b4[1] - a4[2] * b4[2]};
    double n = sqrt(qn[0] * qn[0] + qn[1] * qn[1] + qn[2] * qn[2] + qn[3] * qn[3]);
    double Rt[3];
    m3v(Rd, T.t, Rt);
    for (int i = 0; i < 3; i++) out7[i] = td[i] + Rt[i];
    for (int i = 0; i < 4; i++) out7[3 + i] = qn[i] / n;
}

/* Cholesky solve of a dense SPD system (n x n, row-major), in place; returns 0 on failure */
static int chol_solve(double *A, double *b, int n) {
    for (int j = 0; j < n; j++) {
        double d = A[j * n + j];
        for (int k = 0; k < j; k++) d -= A[j * n + k] * A[j * n + k];
        if (!(d > 0)) return 0;
        d = sqrt(d);
        A[j * n + j] = d;
        for (int i = j + 1; i < n; i++) {
            double s = A[i * n + j];
            for (int k = 0; k < j; k++) s -= A[i * n + k] * A[j * n + k];
            A[i * n + j] = s / d;
        }
    }
    for (int i = 0; i < n; i++) {
        double s = b[i];
        for (int k = 0; k < i; k++) s -= A[i * n + k] * b[k];
        b[i] = s / A[i * n + i];
    }
    for (int i = n - 1; i >= 0; i--) {
        double s = b[i];
        for (int k = i + 1; k < n; k++) s -= A[k * n + i] * b[k];
        b[i] = s / A[i * n + i];
    }
    return 1;
}

static void huber(double s, double a, int robust, double *rho0, double *rho1) {
    double b = a * a;
    if (robust && s > b) {
        double r = sqrt(s);
        *rho0 = 2.0 * a * r - b;
        double v = a / r;
        *rho1 = v > 2.2250738585072014e-308 ? v : 2.2250738585072014e-308;
    } else {
        *rho0 = s;
        *rho1 = 1.0;
    }
}

/* projection residual + the 2x3 block J_pi * R_cw shared by all the reference's cost functions
 * (src/slam/src/ceres_parametrization.cpp:6-94, 96-155, 157-268) */
static void reproj(const orc_se3 *Twc, const double K[4], const double X[3], const double uv[2], double r[2], double JR[6],
                   double *chi2, int *depthPos) {
    double d[3] = {X[0] - Twc->t[0], X[1] - Twc->t[1], X[2] - Twc->t[2]}, c[3];
    m3tv(Twc->R, d, c); /* camera point = R_wc^T (X - t_wc) */
    double iz = 1.0 / c[2];
    r[0] = K[0] * c[0] * iz + K[2] - uv[0];
    r[1] = K[1] * c[1] * iz + K[3] - uv[1];
    *chi2 = r[0] * r[0] + r[1] * r[1];
    *depthPos = c[2] > 0;
    if (JR) {
        double iz2 = iz * iz;
        double Jp[6] = {iz * K[0], 0, -c[0] * iz2 * K[0], 0, iz * K[1], -c[1] * iz2 * K[1]};
        for (int rr = 0; rr < 2; rr++)
            for (int cc = 0; cc < 3; cc++) /* R_cw = R_wc^T */
                JR[3 * rr + cc] = Jp[3 * rr] * Twc->R[3 * cc] + Jp[3 * rr + 1] * Twc->R[3 * cc + 1] + Jp[3 * rr + 2] * Twc->R[3 * cc + 2];
    }
}
/* 2x3 times hat(X) */
static void times_hat(const double JR[6], const double X[3], double out[6]) {
    for (int r = 0; r < 2; r++) {
        const double *j = JR + 3 * r;
        out[3 * r + 0] = j[1] * X[2] - j[2] * X[1];
        out[3 * r + 1] = j[2] * X[0] - j[0] * X[2];
        out[3 * r + 2] = j[0] * X[1] - j[1] * X[0];
    }
}

typedef struct {
    double radius, decrease_factor;
    int reuse_diagonal;
} orc_lm;

/* ---- a9: MultiViewGeometry::ceresPnP (src/slam/src/multi_view_geometry.cpp:129-223), wall-clock cap removed ---- */
typedef struct {
    const double *uv, *wpt;
    const uint8_t *active;
    int n;
    double K[4], huber_a;
    int robust;
    double *chi2;
    uint8_t *depth;
} pnp_prob;

/* cost (+ H = J^T J (36), g = J^T r (6) when H != NULL) at pose x7 */
static double pnp_eval(const pnp_prob *P, const double *x7, double *H, double *g) {
    orc_se3 T;
    se3_from_pose7(x7, &T);
    double cost = 0;
    if (H) {
        memset(H, 0, 36 * sizeof(double));
        memset(g, 0, 6 * sizeof(double));
    }
    for (int i = 0; i < P->n; i++) {
        if (!P->active[i]) continue;
        double r[2], JR[6], JH[6], chi2;
        int dp;
        reproj(&T, P->K, P->wpt + 3 * i, P->uv + 2 * i, r, H ? JR : NULL, &chi2, &dp);
        P->chi2[i] = chi2;
        P->depth[i] = (uint8_t) dp;
        double rho0, rho1;
        huber(chi2, P->huber_a, P->robust, &rho0, &rho1);
        cost += 0.5 * rho0;
        if (H) {
            times_hat(JR, P->wpt + 3 * i, JH);
            double s = sqrt(rho1), J[12];
            for (int rr = 0; rr < 2; rr++)
                for (int c = 0; c < 3; c++) {
                    J[6 * rr + c] = -JR[3 * rr + c] * s;
                    J[6 * rr + 3 + c] = JH[3 * rr + c] * s;
                }
            double rs[2] = {r[0] * s, r[1] * s};
            for (int a = 0; a < 6; a++) {
                g[a] += J[a] * rs[0] + J[6 + a] * rs[1];
                for (int b = 0; b < 6; b++) H[6 * a + b] += J[a] * J[b] + J[6 + a] * J[6 + b];
            }
        }
    }
    return cost;
}

static int pnp_solve(const pnp_prob *P, double *pose7, int maxIterations, double functionTolerance, double *info) {
    double x[7], cand[7], H[36], g[6], scale[6], diag[6];
    memcpy(x, pose7, sizeof(x));
    orc_lm lm = {1e4, 2.0, 0};
    double x_cost = pnp_eval(P, x, H, g);
    for (int i = 0; i < 6; i++) scale[i] = 1.0 / (1.0 + sqrt(H[7 * i]));
    double gmax = 0;
    for (int i = 0; i < 6; i++) gmax = fmax(gmax, fabs(g[i]));
    double x_norm = -1; /* trust_region_minimizer.cc:187 */
    int iteration = 0, nsucc = 1, invalid = 0, nsummaries = 1;
    double initial = x_cost;
    while (1) {
        if (iteration >= maxIterations) break;
        if (gmax <= 1e-10) break;
        if (lm.radius <= 1e-32) break;
        iteration++;
        /* ComputeStep */
        double Hs[36], gs[6], A[36], y[6];
        for (int a = 0; a < 6; a++) {
            gs[a] = g[a] * scale[a];
            for (int b = 0; b < 6; b++) Hs[6 * a + b] = H[6 * a + b] * scale[a] * scale[b];
        }
        if (!lm.reuse_diagonal)
            for (int a = 0; a < 6; a++) diag[a] = fmin(fmax(Hs[7 * a], 1e-6), 1e32);
        memcpy(A, Hs, sizeof(A));
        for (int a = 0; a < 6; a++) A[7 * a] += diag[a] / lm.radius;
        memcpy(y, gs, sizeof(y));
        int okstep = chol_solve(A, y, 6);
        lm.reuse_diagonal = 1;
        double step[6], mcc = 0;
        if (okstep) {
            for (int a = 0; a < 6; a++) step[a] = -y[a];
            double sg = 0, sHs = 0;
            for (int a = 0; a < 6; a++) {
                sg += step[a] * gs[a];
                for (int b = 0; b < 6; b++) sHs += step[a] * Hs[6 * a + b] * step[b];
            }
            mcc = -sg - 0.5 * sHs;
        }
        if (!okstep || !(mcc > 0)) { /* HandleInvalidStep */
            if (++invalid >= 5) return 0;
            lm.radius /= lm.decrease_factor;
            lm.decrease_factor *= 2;
            lm.reuse_diagonal = 1;
            nsummaries++;
            continue;
        }
        invalid = 0;
        double delta[6];
        for (int a = 0; a < 6; a++) delta[a] = step[a] * scale[a];
        se3_plus(x, delta, cand);
        double cand_cost = pnp_eval(P, cand, NULL, NULL);
        double sn = 0;
        for (int i = 0; i < 7; i++) sn += (x[i] - cand[i]) * (x[i] - cand[i]);
        if (sqrt(sn) <= 1e-8 * (x_norm + 1e-8)) break; /* ParameterToleranceReached */
        if (fabs(x_cost - cand_cost) <= functionTolerance * x_cost) break; /* FunctionToleranceReached */
        double rel = (x_cost - cand_cost) / mcc;
        if (getenv("ALVA_ORC_VERBOSE")) fprintf(stderr, "it %d x_cost %.6e cand %.6e mcc %.6e rel %.3f radius %.3e\n", iteration, x_cost, cand_cost, mcc, rel, lm.radius);
        if (rel > 1e-3) {
            memcpy(x, cand, sizeof(x));
            x_norm = 0;
            for (int i = 0; i < 7; i++) x_norm += x[i] * x[i];
            x_norm = sqrt(x_norm);
            x_cost = pnp_eval(P, x, H, g);
            gmax = 0;
            for (int i = 0; i < 6; i++) gmax = fmax(gmax, fabs(g[i]));
            lm.radius = lm.radius / fmax(1.0 / 3.0, 1.0 - pow(2.0 * rel - 1.0, 3));
            lm.radius = fmin(1e16, lm.radius);
            lm.decrease_factor = 2.0;
            lm.reuse_diagonal = 0;
            nsucc++;
        } else {
            lm.radius /= lm.decrease_factor;
            lm.decrease_factor *= 2;
            lm.reuse_diagonal = 1;
        }
        nsummaries++;
    }
    memcpy(pose7, x, sizeof(x));
    if (info) {
        info[0] = nsummaries;
        info[1] = initial;
        info[2] = x_cost;
        info[3] = nsucc;
    }
    return 1;
}

int orc_pnp_refine(const double *uv, const double *wpt, int n, double *pose7, int maxIterations, float chi2th, int useRobust,
                   int applyL2AfterRobust, float fx, float fy, float cx, float cy, int *outliers, int *nOutliers, double *info) {
    pnp_prob P;
    P.uv = uv;
    P.wpt = wpt;
    P.n = n;
    P.K[0] = fx; P.K[1] = fy; P.K[2] = cx; P.K[3] = cy;
    P.huber_a = (double) sqrtf(chi2th); /* :135 std::sqrt(float) -> float */
    P.robust = useRobust;
    uint8_t *active = (uint8_t *) malloc((size_t) n + 1);
    memset(active, 1, (size_t) n);
    P.active = active;
    P.chi2 = (double *) calloc((size_t) n + 1, sizeof(double));
    P.depth = (uint8_t *) calloc((size_t) n + 1, 1);
    if (info) memset(info, 0, 8 * sizeof(double));
    *nOutliers = 0;
    int ok = pnp_solve(&P, pose7, maxIterations, 1e-3, info);
    int nbad = 0;
    for (int i = 0; i < n; i++)
        if (P.chi2[i] > (double) chi2th || !P.depth[i]) { /* multi_view_geometry.cpp:194-207 */
            if (applyL2AfterRobust) active[i] = 0;
            outliers[(*nOutliers)++] = i;
            nbad++;
        }
    int ret;
    if (nbad == n) ret = 0;
    else {
        if (applyL2AfterRobust && nbad > 0) { /* :214-218 */
            P.robust = 0;
            ok = pnp_solve(&P, pose7, maxIterations, 1e-3, info ? info + 4 : NULL);
        }
        ret = ok;
    }
    free(active);
    free(P.chi2);
    free(P.depth);
    return ret;
}

/* ------------------------------------------------------------------------------------------------
 * a10-a13 -- the solve inside Optimizer::localBA (src/slam/src/optimizer.cpp:251-262) on a flat problem:
 * LM + Huber, SPARSE_SCHUR == Schur complement on the point blocks
 * (ceres schur_eliminator_impl.h:177-375: S = F'F + D_f^2 - sum_p (F'E)(E'E + D_e^2)^-1 (E'F); back-substitution
 * :330-375), cost functions ReprojectionErrorKSE3AnchInvDepth / ...KSE3XYZ
 * (src/slam/src/ceres_parametrization.cpp:157-268 / :6-94).  Constant blocks (calibration, constant
 * keyframes) are not part of the reduced program (ceres program.cc RemoveFixedBlocks). */
typedef struct {
    int nKf, nPt, nObs, invDepth, pdim;
    const uint8_t *kfConst;
    const double *calib;
    const int *ancKf;
    const double *ancUv;
    const int *obsKf, *obsPt;
    const double *obsUv;
    double huber_a;
    int nc;      /* free keyframes */
    int *cidx;   /* kf -> free index or -1 */
    double *chi2;
    uint8_t *depth;
    /* normal equations (unscaled) */
    double *Hcc, *gc;     /* (6nc)^2, 6nc */
    double *Hpp, *gp;     /* nPt*pdim*pdim, nPt*pdim */
    double *W;            /* nPt * nc * 6 * pdim : W[p][c] = sum F^T E */
} ba_prob;

static void ba_accum_block(ba_prob *B, int p, const double *Jc[2], const int cids[2], const double *Je, const double r[2]) {
    int dp = B->pdim, n6 = 6 * B->nc;
    for (int a = 0; a < dp; a++) {
        B->gp[p * dp + a] += Je[a] * r[0] + Je[dp + a] * r[1];
        for (int b = 0; b < dp; b++) B->Hpp[(p * dp + a) * dp + b] += Je[a] * Je[b] + Je[dp + a] * Je[dp + b];
    }
    for (int s = 0; s < 2; s++) {
        if (cids[s] < 0) continue;
        const double *J = Jc[s];
        int c = cids[s];
        for (int a = 0; a < 6; a++) {
            B->gc[6 * c + a] += J[a] * r[0] + J[6 + a] * r[1];
            for (int b = 0; b < dp; b++) B->W[(((size_t) p * B->nc + c) * 6 + a) * dp + b] += J[a] * Je[b] + J[6 + a] * Je[dp + b];
        }
        for (int s2 = 0; s2 < 2; s2++) {
            if (cids[s2] < 0) continue;
            const double *J2 = Jc[s2];
            int c2 = cids[s2];
            for (int a = 0; a < 6; a++)
                for (int b = 0; b < 6; b++) B->Hcc[(size_t) (6 * c + a) * n6 + 6 * c2 + b] += J[a] * J2[b] + J[6 + a] * J2[6 + b];
        }
    }
}

static double ba_eval(ba_prob *B, const double *poses, const double *pts, int wantJ) {
    int dp = B->pdim, n6 = 6 * B->nc;
    if (wantJ) {
        memset(B->Hcc, 0, sizeof(double) * (size_t) n6 * n6);
        memset(B->gc, 0, sizeof(double) * (size_t) n6);
        memset(B->Hpp, 0, sizeof(double) * (size_t) B->nPt * dp * dp);
        memset(B->gp, 0, sizeof(double) * (size_t) B->nPt * dp);
        memset(B->W, 0, sizeof(double) * (size_t) B->nPt * B->nc * 6 * dp);
    }
    orc_se3 *T = (orc_se3 *) malloc(sizeof(orc_se3) * (size_t) B->nKf);
    for (int k = 0; k < B->nKf; k++) se3_from_pose7(poses + 7 * k, &T[k]);
    const double *K = B->calib;
    double cost = 0;
    for (int o = 0; o < B->nObs; o++) {
        int k = B->obsKf[o], p = B->obsPt[o];
        double X[3], r[2], JR[6], chi2, dJl[3] = {0, 0, 0};
        int dpz, a = -1;
        if (B->invDepth) {
            a = B->ancKf[p];
            double zanch = 1.0 / pts[p];
            /* anchpt = zanch * K^-1 * (ua, va, 1) */
            double ap[3] = {zanch * ((B->ancUv[2 * p] - K[2]) / K[0]), zanch * ((B->ancUv[2 * p + 1] - K[3]) / K[1]), zanch};
            double Ra[3];
            m3v(T[a].R, ap, Ra);
            X[0] = Ra[0] + T[a].t[0]; X[1] = Ra[1] + T[a].t[1]; X[2] = Ra[2] + T[a].t[2];
            dJl[0] = -zanch * Ra[0]; dJl[1] = -zanch * Ra[1]; dJl[2] = -zanch * Ra[2]; /* J_lambda, :262 */
        } else {
            memcpy(X, pts + 3 * p, 24);
        }
        reproj(&T[k], K, X, B->obsUv + 2 * o, r, wantJ ? JR : NULL, &chi2, &dpz);
        B->chi2[o] = chi2;
        B->depth[o] = (uint8_t) dpz;
        double rho0, rho1;
        huber(chi2, B->huber_a, 1, &rho0, &rho1);
        cost += 0.5 * rho0;
        if (!wantJ) continue;
        double s = sqrt(rho1), JH[6], Jobs[12], Janc[12], Je[6], rs[2] = {r[0] * s, r[1] * s};
        times_hat(JR, X, JH);
        for (int rr = 0; rr < 2; rr++)
            for (int c = 0; c < 3; c++) {
                Jobs[6 * rr + c] = -JR[3 * rr + c] * s;
                Jobs[6 * rr + 3 + c] = JH[3 * rr + c] * s;
                Janc[6 * rr + c] = JR[3 * rr + c] * s;
                Janc[6 * rr + 3 + c] = -JH[3 * rr + c] * s;
            }
        if (B->invDepth) {
            Je[0] = (JR[0] * dJl[0] + JR[1] * dJl[1] + JR[2] * dJl[2]) * s;
            Je[1] = (JR[3] * dJl[0] + JR[4] * dJl[1] + JR[5] * dJl[2]) * s;
        } else {
            for (int rr = 0; rr < 2; rr++)
                for (int c = 0; c < 3; c++) Je[3 * rr + c] = JR[3 * rr + c] * s;
        }
        const double *Jc[2] = {Jobs, Janc};
        int cids[2] = {B->cidx[k], (B->invDepth ? B->cidx[a] : -1)};
        ba_accum_block(B, p, Jc, cids, Je, rs);
    }
    free(T);
    return cost;
}

int orc_local_ba(int nKf, double *poses, const uint8_t *kfConst, const double *calib, int invDepth, int nPt, const int *ptAnchorKf,
                 const double *ptAnchorUv, double *ptParam, int nObs, const int *obsKf, const int *obsPt, const double *obsUv,
                 int maxIterations, double functionTolerance, double huberChi2, double *chi2, uint8_t *depthPos, double *info) {
    ba_prob B;
    memset(&B, 0, sizeof(B));
    B.nKf = nKf; B.nPt = nPt; B.nObs = nObs; B.invDepth = invDepth; B.pdim = invDepth ? 1 : 3;
    B.kfConst = kfConst; B.calib = calib; B.ancKf = ptAnchorKf; B.ancUv = ptAnchorUv;
    B.obsKf = obsKf; B.obsPt = obsPt; B.obsUv = obsUv;
    B.huber_a = (double) sqrtf((float) huberChi2); /* optimizer.cpp:22: std::sqrt(float) */
    B.chi2 = chi2; B.depth = depthPos;
    B.cidx = (int *) malloc(sizeof(int) * (size_t) nKf);
    for (int k = 0; k < nKf; k++) B.cidx[k] = kfConst[k] ? -1 : B.nc++;
    int dp = B.pdim, nc = B.nc, n6 = 6 * nc, np = nPt * dp;
    B.Hcc = (double *) malloc(sizeof(double) * (size_t) (n6 * n6 + 1));
    B.gc = (double *) malloc(sizeof(double) * (size_t) (n6 + 1));
    B.Hpp = (double *) malloc(sizeof(double) * (size_t) (np * dp + 1));
    B.gp = (double *) malloc(sizeof(double) * (size_t) (np + 1));
    B.W = (double *) malloc(sizeof(double) * ((size_t) nPt * nc * 6 * dp + 1));
    double *x_p = (double *) malloc(sizeof(double) * 7 * (size_t) nKf), *c_p = (double *) malloc(sizeof(double) * 7 * (size_t) nKf);
    double *x_t = (double *) malloc(sizeof(double) * (size_t) (np + 1)), *c_t = (double *) malloc(sizeof(double) * (size_t) (np + 1));
    for (int k = 0; k < nKf; k++) { /* PoseParametersBlock(id, SE3d): unit quaternion */
        orc_se3 T;
        se3_from_pose7(poses + 7 * k, &T);
        memcpy(x_p + 7 * k, T.t, 24);
        memcpy(x_p + 7 * k + 3, T.q, 32);
    }
    memcpy(x_t, ptParam, sizeof(double) * (size_t) np);
    double *sc = (double *) malloc(sizeof(double) * (size_t) (n6 + 1)), *sp = (double *) malloc(sizeof(double) * (size_t) (np + 1));
    double *dc = (double *) malloc(sizeof(double) * (size_t) (n6 + 1)), *dpd = (double *) malloc(sizeof(double) * (size_t) (np + 1));
    double *S = (double *) malloc(sizeof(double) * (size_t) (n6 * n6 + 1)), *rhs = (double *) malloc(sizeof(double) * (size_t) (n6 + 1));
    double *yp = (double *) malloc(sizeof(double) * (size_t) (np + 1)), *Hinv = (double *) malloc(sizeof(double) * (size_t) (np * dp + 1));
    orc_lm lm = {1e4, 2.0, 0};
    double x_cost = ba_eval(&B, x_p, x_t, 1), initial = x_cost, x_norm = -1;
    for (int i = 0; i < n6; i++) sc[i] = 1.0 / (1.0 + sqrt(B.Hcc[(size_t) i * n6 + i]));
    for (int p = 0; p < nPt; p++)
        for (int a = 0; a < dp; a++) sp[p * dp + a] = 1.0 / (1.0 + sqrt(B.Hpp[(p * dp + a) * dp + a]));
    double gmax = 0;
    for (int i = 0; i < n6; i++) gmax = fmax(gmax, fabs(B.gc[i]));
    for (int i = 0; i < np; i++) gmax = fmax(gmax, fabs(B.gp[i]));
    int iteration = 0, nsucc = 1, invalid = 0, nsummaries = 1, ok = 1;
    while (1) {
        if (iteration >= maxIterations || gmax <= 1e-10 || lm.radius <= 1e-32) break;
        iteration++;
        if (!lm.reuse_diagonal) {
            for (int i = 0; i < n6; i++) dc[i] = fmin(fmax(B.Hcc[(size_t) i * n6 + i] * sc[i] * sc[i], 1e-6), 1e32);
            for (int p = 0; p < nPt; p++)
                for (int a = 0; a < dp; a++) dpd[p * dp + a] = fmin(fmax(B.Hpp[(p * dp + a) * dp + a] * sp[p * dp + a] * sp[p * dp + a], 1e-6), 1e32);
        }
        lm.reuse_diagonal = 1;
        /* scaled Schur complement */
        for (int i = 0; i < n6; i++) {
            rhs[i] = B.gc[i] * sc[i];
            for (int j = 0; j < n6; j++) S[(size_t) i * n6 + j] = B.Hcc[(size_t) i * n6 + j] * sc[i] * sc[j];
            S[(size_t) i * n6 + i] += dc[i] / lm.radius;
        }
        int okstep = 1;
        for (int p = 0; p < nPt && okstep; p++) {
            double M[9], Mi[9];
            for (int a = 0; a < dp; a++)
                for (int b = 0; b < dp; b++)
                    M[a * dp + b] = B.Hpp[(p * dp + a) * dp + b] * sp[p * dp + a] * sp[p * dp + b] + (a == b ? dpd[p * dp + a] / lm.radius : 0.0);
            if (dp == 1) Mi[0] = 1.0 / M[0];
            else { /* 3x3 SPD inverse via Cholesky solves of the identity */
                for (int c = 0; c < 3; c++) {
                    double A3[9], e[3] = {c == 0, c == 1, c == 2};
                    memcpy(A3, M, sizeof(A3));
                    if (!chol_solve(A3, e, 3)) { okstep = 0; break; }
                    for (int rr = 0; rr < 3; rr++) Mi[rr * 3 + c] = e[rr];
                }
            }
            memcpy(Hinv + (size_t) p * dp * dp, Mi, sizeof(double) * (size_t) dp * dp);
            /* t = Hinv * gp_s ; for each cam pair accumulate */
            double tg[3] = {0, 0, 0};
            for (int a = 0; a < dp; a++)
                for (int b = 0; b < dp; b++) tg[a] += Mi[a * dp + b] * B.gp[p * dp + b] * sp[p * dp + b];
            for (int c = 0; c < nc; c++) {
                const double *Wc = B.W + (((size_t) p * nc + c) * 6) * dp;
                int nz = 0;
                for (int i = 0; i < 6 * dp; i++) nz |= (Wc[i] != 0);
                if (!nz) continue;
                double WH[18]; /* (W_s Hinv) 6 x dp */
                for (int a = 0; a < 6; a++)
                    for (int b = 0; b < dp; b++) {
                        double v = 0;
                        for (int m = 0; m < dp; m++) v += Wc[a * dp + m] * sc[6 * c + a] * sp[p * dp + m] * Mi[m * dp + b];
                        WH[a * dp + b] = v;
                    }
                for (int a = 0; a < 6; a++) {
                    double v = 0;
                    for (int b = 0; b < dp; b++) v += Wc[a * dp + b] * sc[6 * c + a] * sp[p * dp + b] * tg[b];
                    rhs[6 * c + a] -= v;
                }
                for (int c2 = 0; c2 < nc; c2++) {
                    const double *W2 = B.W + (((size_t) p * nc + c2) * 6) * dp;
                    int nz2 = 0;
                    for (int i = 0; i < 6 * dp; i++) nz2 |= (W2[i] != 0);
                    if (!nz2) continue;
                    for (int a = 0; a < 6; a++)
                        for (int b = 0; b < 6; b++) {
                            double v = 0;
                            for (int m = 0; m < dp; m++) v += WH[a * dp + m] * W2[b * dp + m] * sc[6 * c2 + b] * sp[p * dp + m];
                            S[(size_t) (6 * c + a) * n6 + 6 * c2 + b] -= v;
                        }
                }
            }
        }
        if (okstep && n6 > 0) okstep = chol_solve(S, rhs, n6); /* rhs <- y_c */
        double mcc = 0;
        if (okstep) {
            /* back-substitute y_p = Hinv (gp_s - W_s^T y_c); then step = -y, model cost change */
            double sg = 0, sHs = 0;
            for (int p = 0; p < nPt; p++) {
                double t[3];
                for (int a = 0; a < dp; a++) {
                    double v = B.gp[p * dp + a] * sp[p * dp + a];
                    for (int c = 0; c < nc; c++) {
                        const double *Wc = B.W + (((size_t) p * nc + c) * 6) * dp;
                        for (int i = 0; i < 6; i++) v -= Wc[i * dp + a] * sc[6 * c + i] * sp[p * dp + a] * rhs[6 * c + i];
                    }
                    t[a] = v;
                }
                for (int a = 0; a < dp; a++) {
                    double v = 0;
                    for (int b = 0; b < dp; b++) v += Hinv[(size_t) p * dp * dp + a * dp + b] * t[b];
                    yp[p * dp + a] = v;
                }
            }
            for (int i = 0; i < n6; i++) sg += -rhs[i] * B.gc[i] * sc[i];
            for (int i = 0; i < np; i++) sg += -yp[i] * B.gp[i] * sp[i];
            for (int i = 0; i < n6; i++)
                for (int j = 0; j < n6; j++) sHs += rhs[i] * B.Hcc[(size_t) i * n6 + j] * sc[i] * sc[j] * rhs[j];
            for (int p = 0; p < nPt; p++) {
                for (int a = 0; a < dp; a++)
                    for (int b = 0; b < dp; b++) sHs += yp[p * dp + a] * B.Hpp[(p * dp + a) * dp + b] * sp[p * dp + a] * sp[p * dp + b] * yp[p * dp + b];
                for (int c = 0; c < nc; c++) {
                    const double *Wc = B.W + (((size_t) p * nc + c) * 6) * dp;
                    for (int i = 0; i < 6; i++)
                        for (int a = 0; a < dp; a++) sHs += 2 * rhs[6 * c + i] * Wc[i * dp + a] * sc[6 * c + i] * sp[p * dp + a] * yp[p * dp + a];
                }
            }
            mcc = -sg - 0.5 * sHs;
        }
        if (!okstep || !(mcc > 0)) {
            if (++invalid >= 5) { ok = 0; break; }
            lm.radius /= lm.decrease_factor;
            lm.decrease_factor *= 2;
            nsummaries++;
            continue;
        }
        invalid = 0;
        double sn = 0;
        memcpy(c_p, x_p, sizeof(double) * 7 * (size_t) nKf);
        for (int k = 0; k < nKf; k++) {
            int c = B.cidx[k];
            if (c < 0) continue;
            double d6[6];
            for (int i = 0; i < 6; i++) d6[i] = -rhs[6 * c + i] * sc[6 * c + i];
            se3_plus(x_p + 7 * k, d6, c_p + 7 * k);
            for (int i = 0; i < 7; i++) sn += (x_p[7 * k + i] - c_p[7 * k + i]) * (x_p[7 * k + i] - c_p[7 * k + i]);
        }
        for (int i = 0; i < np; i++) {
            c_t[i] = x_t[i] + (-yp[i] * sp[i]);
            sn += (x_t[i] - c_t[i]) * (x_t[i] - c_t[i]);
        }
        double cand_cost = ba_eval(&B, c_p, c_t, 0);
        if (sqrt(sn) <= 1e-8 * (x_norm + 1e-8)) break;
        if (fabs(x_cost - cand_cost) <= functionTolerance * x_cost) break;
        double rel = (x_cost - cand_cost) / mcc;
        if (getenv("ALVA_ORC_VERBOSE")) fprintf(stderr, "ba it %d x_cost %.9e cand %.9e mcc %.6e rel %.4f radius %.3e\n", iteration, x_cost, cand_cost, mcc, rel, lm.radius);
        if (rel > 1e-3) {
            memcpy(x_p, c_p, sizeof(double) * 7 * (size_t) nKf);
            memcpy(x_t, c_t, sizeof(double) * (size_t) np);
            x_norm = 0;
            for (int k = 0; k < nKf; k++)
                if (B.cidx[k] >= 0)
                    for (int i = 0; i < 7; i++) x_norm += x_p[7 * k + i] * x_p[7 * k + i];
            for (int i = 0; i < np; i++) x_norm += x_t[i] * x_t[i];
            x_norm = sqrt(x_norm);
            x_cost = ba_eval(&B, x_p, x_t, 1);
            gmax = 0;
            for (int i = 0; i < n6; i++) gmax = fmax(gmax, fabs(B.gc[i]));
            for (int i = 0; i < np; i++) gmax = fmax(gmax, fabs(B.gp[i]));
            lm.radius = lm.radius / fmax(1.0 / 3.0, 1.0 - pow(2.0 * rel - 1.0, 3));
            lm.radius = fmin(1e16, lm.radius);
            lm.decrease_factor = 2.0;
            lm.reuse_diagonal = 0;
            nsucc++;
        } else {
            lm.radius /= lm.decrease_factor;
            lm.decrease_factor *= 2;
            lm.reuse_diagonal = 1;
        }
        nsummaries++;
    }
    for (int k = 0; k < nKf; k++)
        if (B.cidx[k] >= 0) memcpy(poses + 7 * k, x_p + 7 * k, 56);
    memcpy(ptParam, x_t, sizeof(double) * (size_t) np);
    if (info) {
        info[0] = nsummaries; info[1] = initial; info[2] = x_cost; info[3] = nsucc;
    }
    free(B.cidx); free(B.Hcc); free(B.gc); free(B.Hpp); free(B.gp); free(B.W); free(x_p); free(c_p); free(x_t); free(c_t);
    free(sc); free(sp); free(dc); free(dpd); free(S); free(rhs); free(yp); free(Hinv);
    return ok;
}

/* ------------------------------------------------------------------------------------------------
 * a5 -- FeatureExtractor::detectFeaturePoints (src/slam/src/feature_extractor.cpp:11-158): the reference's
 * per-grid-cell Shi-Tomasi detector.  Cells are visited in row-major order and share one float mask, so the
 * visiting order is part of the semantics. */

/* cv::circle(mask, c, r, 0, FILLED): imgproc/src/drawing.cpp:1477-1617 (Circle(), fill) -- half-width of the
 * filled midpoint circle per |dy| */
static void circle_halfwidths(int radius, int *hw /* [radius+1] */) {
    for (int i = 0; i <= radius; i++) hw[i] = -1;
    int err = 0, dx = radius, dy = 0, plus = 1, minus = (radius << 1) - 1;
    while (dx >= dy) {
        if (dx > hw[dy]) hw[dy] = dx;
        if (dy > hw[dx]) hw[dx] = dy;
        dy++;
        err += plus;
        plus += 2;
        int mask = (err <= 0) - 1;
        err -= minus & mask;
        dx += mask;
        minus -= mask & 2;
    }
}
static void draw_zero_circle(uint8_t *mask, int w, int h, int cx, int cy, int radius, const int *hw) {
    for (int dy = -radius; dy <= radius; dy++) {
        int y = cy + dy, half = hw[dy < 0 ? -dy : dy];
        if (y < 0 || y >= h || half < 0) continue;
        int x0 = cx - half, x1 = cx + half;
        if (x0 < 0) x0 = 0;
        if (x1 > w - 1) x1 = w - 1;
        for (int x = x0; x <= x1; x++) mask[(size_t) y * w + x] = 0;
    }
}

static float cov_at(const float *dx, const float *dy, int cell, int y, int x, int ch) {
    volatile float fx = dx[y * cell + x], fy = dy[y * cell + x];
    volatile float v = ch == 0 ? fx * fx : (ch == 1 ? fx * fy : fy * fy);
    return v;
}

/* One cell: GaussianBlur(image(roi), 3x3, sigma 0) -- integer sum [1 2 1;2 4 2;1 2 1] p / 16 (rounding: see the column pass note)
 * reading real neighbours, REFLECT_101 only at the image border (imgproc/src/smooth.dispatch.cpp:654,754;
 * filter.dispatch.cpp:336-363) -- then cornerMinEigenVal(block 3, Sobel 3) on the STAND-ALONE blurred cell
 * (imgproc/src/corner.cpp:237-316): scale s = (float)(1/3060);
 *   Dx: row (c - a), column (r0 + r2)*s + r1*(2s)        (deriv.cpp:427-439; filter.simd.hpp:2806-2911)
 *   Dy: row s*a + 2s*b + s*c (left to right), column r2 - r0   (filter.simd.hpp:2446-2488, :2915-2937)
 *   cov = (dx*dx, dx*dy, dy*dy); 3x3 box sum accumulated in double, cast to float once (box_filter.simd.hpp);
 *   lambda = (a/2 + c/2) - sqrt((a/2 - c/2)^2 + b*b)       (corner.cpp:52-102)          all REFLECT_101 at the cell edge. */
void orc_cell_mineig(const uint8_t *gray, int w, int h, int x0, int y0, int cell, uint8_t *blurOut, float *eig) {
    uint8_t *B = (uint8_t *) malloc((size_t) cell * cell);
    float *dx = (float *) malloc(sizeof(float) * (size_t) cell * cell), *dy = (float *) malloc(sizeof(float) * (size_t) cell * cell);
    for (int y = 0; y < cell; y++)
        for (int x = 0; x < cell; x++) {
            static const int k3[3] = {1, 2, 1};
            int acc = 0;
            for (int j = -1; j <= 1; j++)
                for (int i = -1; i <= 1; i++)
                    acc += k3[j + 1] * k3[i + 1] * gray[(size_t) reflect101(y0 + y + j, h) * w + reflect101(x0 + x + i, w)];
            /* column pass: the vector part (SymmColumnVec_32s8u, filter.simd.hpp:1010-1099) evaluates acc/16 exactly in
             * float and rounds HALF-TO-EVEN (v_round); the scalar tail x >= (cell & ~3) uses FixedPtCastEx = (v + 2^15) >> 16,
             * i.e. half-up.  (128-bit universal intrinsics: chunks of 16, then 8, then 4 columns.) */
            int q = acc >> 4, rem = acc & 15, v;
            if (x < (cell & ~3)) v = rem > 8 ? q + 1 : (rem < 8 ? q : (q + (q & 1)));
            else v = (acc + 8) >> 4;
            B[y * cell + x] = (uint8_t) (v > 255 ? 255 : v);
        }
    const float s = (float) (1.0 / (4.0 * 3.0 * 255.0));
    volatile float s2 = 2.0f * s;
#define BP(yy, xx) ((float) B[reflect101(yy, cell) * cell + reflect101(xx, cell)])
    for (int y = 0; y < cell; y++)
        for (int x = 0; x < cell; x++) {
            /* Dx */
            volatile float r0 = BP(y - 1, x + 1) - BP(y - 1, x - 1), r1 = BP(y, x + 1) - BP(y, x - 1), r2 = BP(y + 1, x + 1) - BP(y + 1, x - 1);
            volatile float t0 = r0 + r2;
            volatile float t1 = t0 * s, t2 = r1 * s2;
            dx[y * cell + x] = t1 + t2;
            /* Dy: row-filtered rows y-1 and y+1 */
            volatile float ua = s * BP(y - 1, x - 1), ub = s2 * BP(y - 1, x), uc = s * BP(y - 1, x + 1);
            volatile float u01 = ua + ub;
            volatile float up = u01 + uc;
            volatile float da = s * BP(y + 1, x - 1), db = s2 * BP(y + 1, x), dc = s * BP(y + 1, x + 1);
            volatile float d01 = da + db;
            volatile float dn = d01 + dc;
            dy[y * cell + x] = dn - up;
        }
#undef BP
    /* boxFilter(cov, cov, CV_32F, 3x3, normalize=false): RowSum<float,double> then ColumnSum<double,float>
     * (imgproc/src/box_filter.simd.hpp:65-84,176-270): SLIDING sums in double -- s += new - old along x, and
     * SUM = (SUM + R[y+1]) -> out, SUM -= R[y-1] down y -- so the rounding history is part of the result. */
    double *R = (double *) malloc(sizeof(double) * 3 * (size_t) cell * cell);
    for (int y = 0; y < cell; y++)
        for (int ch = 0; ch < 3; ch++) {
#define COV(xx) cov_at(dx, dy, cell, y, reflect101(xx, cell), ch)
            double sacc = 0;
            sacc += (double) COV(-1);
            sacc += (double) COV(0);
            sacc += (double) COV(1);
            R[((size_t) y * cell + 0) * 3 + ch] = sacc;
            for (int x = 1; x < cell; x++) {
                sacc += (double) COV(x + 1) - (double) COV(x - 2);
                R[((size_t) y * cell + x) * 3 + ch] = sacc;
            }
#undef COV
        }
    for (int x = 0; x < cell; x++) {
        double SUM[3] = {0, 0, 0};
        for (int ch = 0; ch < 3; ch++) {
            SUM[ch] += R[((size_t) reflect101(-1, cell) * cell + x) * 3 + ch];
            SUM[ch] += R[((size_t) 0 * cell + x) * 3 + ch];
        }
        for (int y = 0; y < cell; y++) {
            float box[3];
            for (int ch = 0; ch < 3; ch++) {
                double s0 = SUM[ch] + R[((size_t) reflect101(y + 1, cell) * cell + x) * 3 + ch];
                box[ch] = (float) s0;
                SUM[ch] = s0 - R[((size_t) reflect101(y - 1, cell) * cell + x) * 3 + ch];
            }
            volatile float a = box[0] * 0.5f, b = box[1], c = box[2] * 0.5f;
            volatile float t = a - c;
            volatile float tt = t * t, bb = b * b;
            volatile float sum = bb + tt;
            volatile float sq = sqrtf(sum);
            volatile float ac = a + c;
            eig[y * cell + x] = ac - sq;
        }
    }
    free(R);
    if (blurOut) memcpy(blurOut, B, (size_t) cell * cell);
    free(B); free(dx); free(dy);
}

/* cv::getRectSubPix(8U -> 32F), imgproc/src/samplers.cpp:219-268 (+ generic border branch :129-216) */
static void rect_subpix_8u32f(const uint8_t *src, int w, int h, int ww, int wh, float cx, float cy, float *dst) {
    cx -= (ww - 1) * 0.5f;
    cy -= (wh - 1) * 0.5f;
    int ipx = (int) floorf(cx), ipy = (int) floorf(cy);
    if (0 <= ipx && ipx + ww < w && 0 <= ipy && ipy + wh < h) {
        volatile float a = cx - ipx, b = cy - ipy;
        if (a < 0.0001f) a = 0.0001f;
        volatile float omb = 1.f - b;
        volatile float a12 = a * omb, a22 = a * b, b1 = omb, b2 = b, oma = 1 - a;
        double s = (1. - a) / a;
        const uint8_t *p = src + (size_t) ipy * w + ipx;
        for (int i = 0; i < wh; i++, p += w, dst += ww) {
            volatile float e0 = b1 * p[0], e1 = b2 * p[w];
            volatile float e = e0 + e1;
            volatile float prev = oma * e;
            for (int j = 0; j < ww; j++) {
                volatile float t0 = a12 * p[j + 1], t1 = a22 * p[j + 1 + w];
                volatile float t = t0 + t1;
                dst[j] = prev + t;
                prev = (float) (t * s);
            }
        }
        return;
    }
    /* generic branch with adjustRect (samplers.cpp:43-110) */
    volatile float a = cx - ipx, b = cy - ipy;
    volatile float oma = 1.f - a, omb = 1.f - b;
    volatile float a11 = oma * omb, a12 = a * omb, a21 = oma * b, a22 = a * b, b1 = omb, b2 = b;
    int rx, ry, rw, rh;
    const uint8_t *p = src;
    if (ipx >= 0) { p += ipx; rx = 0; } else { rx = -ipx; if (rx > ww) rx = ww; }
    if (ipx < w - ww) rw = ww; else { rw = w - ipx - 1; if (rw < 0) { p += rw; rw = 0; } }
    if (ipy >= 0) { p += (size_t) ipy * w; ry = 0; } else ry = -ipy;
    if (ipy < h - wh) rh = wh; else { rh = h - ipy - 1; if (rh < 0) { p += (ptrdiff_t) rh * w; rh = 0; } }
    p -= rx;
    for (int i = 0; i < wh; i++, dst += ww) {
        const uint8_t *p2 = p + w;
        if (i < ry || i >= rh) p2 -= w;
        volatile float q0 = p[rx] * b1, q1 = p2[rx] * b2;
        float s0 = q0 + q1;
        for (int j = 0; j < rx; j++) dst[j] = s0;
        q0 = p[rw] * b1; q1 = p2[rw] * b2;
        s0 = q0 + q1;
        for (int j = rw; j < ww; j++) dst[j] = s0;
        for (int j = rx; j < rw; j++) {
            volatile float m0 = p[j] * a11, m1 = p[j + 1] * a12, m2 = p2[j] * a21, m3 = p2[j + 1] * a22;
            volatile float m01 = m0 + m1;
            volatile float m012 = m01 + m2;
            dst[j] = m012 + m3;
        }
        if (i < rh) p = p2;
    }
}

/* cv::cornerSubPix(image, pts, Size(3,3), Size(-1,-1), {EPS+MAX_ITER, 30, 0.01}): imgproc/src/cornersubpix.cpp:44-156 */
void orc_corner_subpix(const uint8_t *gray, int w, int h, float *pts, int n) {
    enum { WINH = 3, WW = 7 };
    float mask[WW * WW], buf[(WW + 2) * (WW + 2)];
    for (int i = 0; i < WW; i++) {
        float y = (float) (i - WINH) / WINH;
        float vy = expf(-y * y);
        for (int j = 0; j < WW; j++) {
            float x = (float) (j - WINH) / WINH;
            mask[i * WW + j] = (float) (vy * expf(-x * x));
        }
    }
    const int max_iters = 30;
    double eps = 0.01;
    eps *= eps;
    for (int pi = 0; pi < n; pi++) {
        float cTx = pts[2 * pi], cTy = pts[2 * pi + 1], cIx = cTx, cIy = cTy;
        int iter = 0;
        double err = 0;
        do {
            double a = 0, b = 0, c = 0, bb1 = 0, bb2 = 0;
            rect_subpix_8u32f(gray, w, h, WW + 2, WW + 2, cIx, cIy, buf);
            const float *sp = buf + (WW + 2) + 1;
            for (int i = 0, k = 0; i < WW; i++, sp += WW + 2) {
                double py = i - WINH;
                for (int j = 0; j < WW; j++, k++) {
                    double m = mask[k];
                    volatile float fgx = sp[j + 1] - sp[j - 1], fgy = sp[j + WW + 2] - sp[j - WW - 2];
                    double tgx = fgx, tgy = fgy;
                    double gxx = tgx * tgx * m, gxy = tgx * tgy * m, gyy = tgy * tgy * m, px = j - WINH;
                    a += gxx;
                    b += gxy;
                    c += gyy;
                    bb1 += gxx * px + gxy * py;
                    bb2 += gxy * px + gyy * py;
                }
            }
            double det = a * c - b * b;
            if (fabs(det) <= 2.220446049250313e-16 * 2.220446049250313e-16) break;
            double scale = 1.0 / det;
            float nx = (float) (cIx + c * scale * bb1 - b * scale * bb2);
            float ny = (float) (cIy - b * scale * bb1 + a * scale * bb2);
            volatile float ex = nx - cIx, ey = ny - cIy;
            volatile float ex2 = ex * ex, ey2 = ey * ey;
            volatile float e = ex2 + ey2;
            err = e;
            cIx = nx;
            cIy = ny;
            if (cIx < 0 || cIx >= w || cIy < 0 || cIy >= h) break;
        } while (++iter < max_iters && err > eps);
        if (fabs((double) (cIx - cTx)) > WINH || fabs((double) (cIy - cTy)) > WINH) {
            cIx = cTx;
            cIy = cTy;
        }
        pts[2 * pi] = cIx;
        pts[2 * pi + 1] = cIy;
    }
}

/* returns the number of points written (<= cap); *maxQuality is updated like FeatureExtractor::maxQuality_ */
int orc_detect_grid(const uint8_t *gray, int w, int h, int cell, const float *occupied, int nOcc, int roiX, int roiY, int roiW,
                    int roiH, double *maxQuality, float *outPts, int cap) {
    int radius = cell / 4, nCH = h / cell, nCW = w / cell, nCells = nCH * nCW;
    int *hw = (int *) malloc(sizeof(int) * (size_t) (radius + 1));
    circle_halfwidths(radius, hw);
    uint8_t *mask = (uint8_t *) malloc((size_t) w * h), *occ = (uint8_t *) calloc((size_t) (nCH + 1) * (nCW + 1), 1);
    memset(mask, 1, (size_t) w * h);
    for (int i = 0; i < nOcc; i++) { /* :30-36 */
        float px = occupied[2 * i], py = occupied[2 * i + 1];
        occ[(size_t) (py / cell) * (nCW + 1) + (size_t) (px / cell)] = 1;
        draw_zero_circle(mask, w, h, cv_round_f(px), cv_round_f(py), radius, hw);
    }
    float *eig = (float *) malloc(sizeof(float) * (size_t) cell * cell);
    float *prim = (float *) malloc(sizeof(float) * 2 * (size_t) nCells), *sec = (float *) malloc(sizeof(float) * 2 * (size_t) nCells);
    uint8_t *hasP = (uint8_t *) calloc((size_t) nCells, 1), *hasS = (uint8_t *) calloc((size_t) nCells, 1);
    size_t numOccupied = 0;
    for (int i = 0; i < nCells; i++) {
        int r = i / nCW, c = i % nCW;
        if (occ[(size_t) r * (nCW + 1) + c]) {
            numOccupied++;
            continue;
        }
        int x = c * cell, y = r * cell;
        if (!(x + cell < w - 1 && y + cell < h - 1)) continue; /* :62 */
        orc_cell_mineig(gray, w, h, x, y, cell, NULL, eig);
        for (int pass = 0; pass < 2; pass++) {
            float best = -3.402823466e+38f;
            int bi = 0;
            for (int k = 0; k < cell * cell; k++) { /* hMap.mul(mask(roi)) + minMaxLoc: first maximum */
                volatile float v = eig[k] * (float) mask[(size_t) (y + k / cell) * w + x + k % cell];
                if (v > best) {
                    best = v;
                    bi = k;
                }
            }
            int mx = x + bi % cell, my = y + bi / cell;
            if (mx < roiX || my < roiY || mx >= roiX + roiW || my >= roiY + roiH) break; /* `continue` of the cell loop, :78-81, :93-96 */
            if ((double) best >= *maxQuality) {
                float *dst = pass == 0 ? prim : sec;
                dst[2 * i] = (float) mx;
                dst[2 * i + 1] = (float) my;
                (pass == 0 ? hasP : hasS)[i] = 1;
                draw_zero_circle(mask, w, h, mx, my, radius, hw);
            }
        }
    }
    int n = 0;
    float *all = (float *) malloc(sizeof(float) * 4 * (size_t) nCells + 8);
    for (int i = 0; i < nCells; i++)
        if (hasP[i]) {
            all[2 * n] = prim[2 * i];
            all[2 * n + 1] = prim[2 * i + 1];
            n++;
        }
    size_t numKeypoints = (size_t) n;
    if (numKeypoints + numOccupied < (size_t) nCells) { /* :117-134 */
        size_t numSec = (size_t) nCells - (numKeypoints + numOccupied), k = 0;
        for (int i = 0; i < nCells; i++)
            if (hasS[i]) {
                all[2 * n] = sec[2 * i];
                all[2 * n + 1] = sec[2 * i + 1];
                n++;
                if (++k == numSec) break;
            }
    }
    numKeypoints = (size_t) n;
    if ((double) numKeypoints < 0.33 * (double) ((size_t) nCells - numOccupied)) *maxQuality *= 0.5; /* :138-145 */
    else if ((double) numKeypoints > 0.9 * (double) ((size_t) nCells - numOccupied)) *maxQuality *= 1.5;
    if (n > 0) orc_corner_subpix(gray, w, h, all, n);
    int m = n < cap ? n : cap;
    memcpy(outPts, all, sizeof(float) * 2 * (size_t) m);
    free(hw); free(mask); free(occ); free(eig); free(prim); free(sec); free(hasP); free(hasS); free(all);
    return n;
}

/* ------------------------------------------------------------------------------------------------
 * a5' -- cv::FAST (TYPE_9_16, NMS) and cv::ORB::detectAndCompute, the north_star-named detector.
 * FAST: src/libs/opencv/modules/features2d/src/fast.cpp:56-292; score: fast_score.cpp:120-;
 * ORB: orb.cpp:784-959 (computeKeyPoints), :130-177 (HarrisResponses), :181-215 (ICAngles),
 * :970-1218 (detectAndCompute), keypoint.cpp:55-117 (retainBest, runByImageBorder),
 * imgproc/src/resize.cpp:733-900 (INTER_LINEAR_EXACT), core/src/mathfuncs_core.simd.hpp:34-71 (fastAtan2,
 * polynomial form -- the wasm build substitutes libm atan2, mathfuncs_core.simd.hpp:39-48). */
static const int fast_off16[16][2] = {{0, 3}, {1, 3}, {2, 2}, {3, 1}, {3, 0}, {3, -1}, {2, -2}, {1, -3},
                                      {0, -3}, {-1, -3}, {-2, -2}, {-3, -1}, {-3, 0}, {-3, 1}, {-2, 2}, {-1, 3}};

/* returns 0 if not a corner at `threshold`, else the FAST score (max threshold keeping it a corner), >= 1 */
static int fast_corner_score(const uint8_t *p, int stride, int threshold) {
    int d[25], v = p[0];
    for (int k = 0; k < 25; k++) d[k] = v - p[fast_off16[k % 16][0] + fast_off16[k % 16][1] * stride];
    int is = 0;
    for (int dir = 0; dir < 2 && !is; dir++) {
        int count = 0;
        for (int k = 0; k < 25; k++) {
            int c = dir == 0 ? (d[k] > threshold) : (d[k] < -threshold); /* darker ring (x < v - t) or brighter ring */
            if (c) {
                if (++count > 8) { is = 1; break; }
            } else count = 0;
        }
    }
    if (!is) return 0;
    /* cornerScore<16>: max over the 16 arcs of 9 of min(d) and of min(-d), floor at threshold, minus 1 */
    int a0 = threshold;
    for (int k = 0; k < 16; k++) {
        int mn = d[k], mx = d[k];
        for (int j = 1; j < 9; j++) {
            int e = d[(k + j) % 16];
            if (e < mn) mn = e;
            if (e > mx) mx = e;
        }
        if (mn > a0) a0 = mn;
        if (-mx > a0) a0 = -mx;
    }
    return a0 - 1;
}

/* score map (0 = no corner) for rows/cols 3..size-4, then 3x3 strict NMS; keypoints in row-major order */
static int fast_detect(const uint8_t *img, int stride, int w, int h, int threshold, int *xy, int *score, int cap, uint8_t *scratch) {
    uint8_t *sc = scratch;
    memset(sc, 0, (size_t) w * h);
    if (threshold < 0) threshold = 0;
    if (threshold > 255) threshold = 255;
    for (int y = 3; y < h - 3; y++)
        for (int x = 3; x < w - 3; x++) sc[(size_t) y * w + x] = (uint8_t) fast_corner_score(img + (size_t) y * stride + x, stride, threshold);
    int n = 0;
    for (int y = 3; y < h - 3; y++)
        for (int x = 3; x < w - 3; x++) {
            int s = sc[(size_t) y * w + x];
            if (!s) continue;
            const uint8_t *q = sc + (size_t) y * w + x;
            if (s > q[-1] && s > q[1] && s > q[-w - 1] && s > q[-w] && s > q[-w + 1] && s > q[w - 1] && s > q[w] && s > q[w + 1]) {
                if (n < cap) {
                    xy[2 * n] = x;
                    xy[2 * n + 1] = y;
                    score[n] = s;
                }
                n++;
            }
        }
    return n;
}

int orc_fast(const uint8_t *gray, int w, int h, int threshold, int *xy, int *score, int cap) {
    uint8_t *scratch = (uint8_t *) malloc((size_t) w * h);
    int n = fast_detect(gray, w, w, h, threshold, xy, score, cap, scratch);
    free(scratch);
    return n;
}

/* resize(prev, cur, sz, 0, 0, INTER_LINEAR_EXACT) for 8UC1: ufixedpoint16 (8 fractional bits) taps from
 * fval = scale*(d + 0.5) - 0.5 in IEEE double (softdouble), rows then columns, (v + 2^15) >> 16 */
static void resize_linear_exact(const uint8_t *src, int sw, int sh, uint8_t *dst, int dw, int dh) {
    int *xo = (int *) malloc(sizeof(int) * (size_t) dw), *yo = (int *) malloc(sizeof(int) * (size_t) dh);
    int *xc = (int *) malloc(sizeof(int) * (size_t) dw), *yc = (int *) malloc(sizeof(int) * (size_t) dh);
    for (int pass = 0; pass < 2; pass++) {
        int dn = pass ? dh : dw, sn = pass ? sh : sw, *o = pass ? yo : xo, *c = pass ? yc : xc;
        double inv_scale = (double) dn / sn;
        volatile double scale = 1.0 / inv_scale;
        for (int d = 0; d < dn; d++) {
            volatile double t = (double) d + 0.5;
            volatile double m = scale * t;
            volatile double fval = m - 0.5;
            int ival = (int) floor(fval);
            if (ival >= 0 && sn > 1) {
                if (ival < sn - 1) {
                    o[d] = ival;
                    volatile double fr = fval - (double) ival;
                    c[d] = (int) lrint(fr * 256.0); /* coeffs[1]; coeffs[0] = 256 - coeffs[1] */
                } else {
                    o[d] = sn - 1;
                    c[d] = -2; /* >= max: copies the last source element */
                }
            } else {
                o[d] = 0;
                c[d] = -1; /* < min: copies the first source element */
            }
        }
    }
    for (int y = 0; y < dh; y++)
        for (int x = 0; x < dw; x++) {
            int rows[2], nr = 0, rc[2];
            if (yc[y] == -1) { rows[0] = 0; rc[0] = 256; nr = 1; }
            else if (yc[y] == -2) { rows[0] = sh - 1; rc[0] = 256; nr = 1; }
            else { rows[0] = yo[y]; rows[1] = yo[y] + 1; rc[0] = 256 - yc[y]; rc[1] = yc[y]; nr = 2; }
            unsigned acc = 0;
            for (int k = 0; k < nr; k++) {
                const uint8_t *r = src + (size_t) rows[k] * sw;
                unsigned hv; /* ufixedpoint16 raw */
                if (xc[x] == -1) hv = (unsigned) r[0] << 8;
                else if (xc[x] == -2) hv = (unsigned) r[sw - 1] << 8;
                else hv = (unsigned) (256 - xc[x]) * r[xo[x]] + (unsigned) xc[x] * r[xo[x] + 1];
                acc += (unsigned) rc[k] * hv;
            }
            unsigned v = (acc + 32768u) >> 16;
            dst[(size_t) y * dw + x] = (uint8_t) (v > 255 ? 255 : v);
        }
    free(xo); free(yo); free(xc); free(yc);
}

static float fast_atan2f(float y, float x) { /* core/src/mathfuncs_core.simd.hpp:34-71 */
    const float s = (float) (180 / 3.1415926535897932384626433832795);
    const float p1 = 0.9997878412794807f * s, p3 = -0.3258083974640975f * s, p5 = 0.1555786518463281f * s, p7 = -0.04432655554792128f * s;
    volatile float ax = fabsf(x), ay = fabsf(y), a, c, c2;
    const float eps = (float) 2.2204460492503131e-16;
    if (ax >= ay) {
        volatile float den = ax + eps;
        c = ay / den;
        c2 = c * c;
        volatile float t = p7 * c2; t = t + p5; t = t * c2; t = t + p3; t = t * c2; t = t + p1;
        a = t * c;
    } else {
        volatile float den = ay + eps;
        c = ax / den;
        c2 = c * c;
        volatile float t = p7 * c2; t = t + p5; t = t * c2; t = t + p3; t = t * c2; t = t + p1;
        volatile float u = t * c;
        a = 90.f - u;
    }
    if (x < 0) a = 180.f - a;
    if (y < 0) a = 360.f - a;
    return a;
}

typedef struct { int x, y; float resp; } orc_cand;

/* retainBest: keep everything with response >= the n-th largest (keypoint.cpp:69-90); stable order */
static int retain_best(orc_cand *c, int n, int keep) {
    if (keep < 0 || n <= keep) return n;
    if (keep == 0) return 0;
    float *tmp = (float *) malloc(sizeof(float) * (size_t) n);
    for (int i = 0; i < n; i++) tmp[i] = c[i].resp;
    /* n-th largest by simple selection (test infrastructure: clarity over speed) */
    for (int i = 0; i < keep; i++) {
        int b = i;
        for (int j = i + 1; j < n; j++)
            if (tmp[j] > tmp[b]) b = j;
        float t = tmp[i]; tmp[i] = tmp[b]; tmp[b] = t;
    }
    float thr = tmp[keep - 1];
    free(tmp);
    int m = 0;
    for (int i = 0; i < n; i++)
        if (c[i].resp >= thr) c[m++] = c[i];
    return m;
}

/* The detector's image pyramid alone (orb.cpp:1041-1058 sizes, :1086-1099 the resize chain): level l is written to out + the sum of the
 * earlier levels' sizes, dims[2 l] = width, dims[2 l + 1] = height.  out == NULL only fills dims.  Returns the total byte count. */
long orc_orb_pyramid(const uint8_t *gray, int w, int h, float scaleFactorF, int nlevels, uint8_t *out, int *dims) {
    double scaleFactor = (double) scaleFactorF;
    long total = 0, prev = 0;
    for (int l = 0; l < nlevels; l++) {
        float sc = (float) pow(scaleFactor, (double) l);
        float inv = 1.0f / sc;
        int lw = cv_round_f((float) w * inv), lh = cv_round_f((float) h * inv);
        dims[2 * l] = lw;
        dims[2 * l + 1] = lh;
        if (out) {
            if (l == 0) memcpy(out, gray, (size_t) w * h);
            else resize_linear_exact(out + prev, dims[2 * l - 2], dims[2 * l - 1], out + total, lw, lh);
        }
        prev = total;
        total += (long) lw * lh;
    }
    return total;
}

int orc_orb_detect_and_compute(const uint8_t *gray, int w, int h, int nfeatures, float scaleFactorF, int nlevels, int fastThreshold,
                               int doDescribe, float *kp /* [cap][6] */, uint8_t *desc, int cap) {
    double scaleFactor = (double) scaleFactorF;
    /* orb.cpp:799-813 */
    int *nPer = (int *) malloc(sizeof(int) * (size_t) nlevels);
    {
        float factor = (float) (1.0 / scaleFactor);
        volatile float num = (float) nfeatures * (1 - factor);
        volatile float den = 1 - (float) pow((double) factor, (double) nlevels);
        float nd = num / den;
        int sum = 0;
        for (int l = 0; l < nlevels - 1; l++) {
            nPer[l] = cv_round_f(nd);
            sum += nPer[l];
            volatile float t = nd * factor;
            nd = t;
        }
        nPer[nlevels - 1] = nfeatures - sum > 0 ? nfeatures - sum : 0;
    }
    /* umax, :819-834 */
    int umax[17];
    {
        const int hp = 15;
        int vmax = (int) floorf(hp * sqrtf(2.f) / 2 + 1), vmin = (int) ceilf(hp * sqrtf(2.f) / 2);
        for (int v = 0; v <= vmax; v++) umax[v] = (int) lrint(sqrt((double) hp * hp - v * v));
        for (int v = hp, v0 = 0; v >= vmin; --v) {
            while (umax[v0] == umax[v0 + 1]) ++v0;
            umax[v] = v0;
            ++v0;
        }
    }
    uint8_t **lv = (uint8_t **) calloc((size_t) nlevels, sizeof(uint8_t *));
    int *lw = (int *) malloc(sizeof(int) * (size_t) nlevels), *lh = (int *) malloc(sizeof(int) * (size_t) nlevels);
    float *lscale = (float *) malloc(sizeof(float) * (size_t) nlevels);
    for (int l = 0; l < nlevels; l++) { /* :1041-1058, :1070-1112 */
        lscale[l] = (float) pow(scaleFactor, (double) l);
        float inv = 1.0f / lscale[l];
        lw[l] = cv_round_f((float) w * inv);
        lh[l] = cv_round_f((float) h * inv);
        lv[l] = (uint8_t *) malloc((size_t) lw[l] * lh[l]);
        if (l == 0) memcpy(lv[0], gray, (size_t) w * h);
        else resize_linear_exact(lv[l - 1], lw[l - 1], lh[l - 1], lv[l], lw[l], lh[l]);
    }
    int total = 0;
    const float hk = 0.04f;
    volatile float hscale = 1.f / ((1 << 2) * 7 * 255.f);
    volatile float hs2 = hscale * hscale;
    volatile float hs3 = hs2 * hscale;
    const float scale_sq_sq = hs3 * hscale;
    for (int l = 0; l < nlevels; l++) {
        int W = lw[l], H = lh[l], capL = W * H / 4 + 16;
        int *xy = (int *) malloc(sizeof(int) * 2 * (size_t) capL), *sc = (int *) malloc(sizeof(int) * (size_t) capL);
        uint8_t *scratch = (uint8_t *) malloc((size_t) W * H);
        int n = fast_detect(lv[l], W, W, H, fastThreshold, xy, sc, capL, scratch);
        if (n > capL) n = capL;
        orc_cand *c = (orc_cand *) malloc(sizeof(orc_cand) * (size_t) (n + 1));
        int m = 0;
        if (H > 62 && W > 62)
            for (int i = 0; i < n; i++) /* runByImageBorder(31) */
                if (xy[2 * i] >= 31 && xy[2 * i] < W - 31 && xy[2 * i + 1] >= 31 && xy[2 * i + 1] < H - 31) {
                    c[m].x = xy[2 * i];
                    c[m].y = xy[2 * i + 1];
                    c[m].resp = (float) sc[i];
                    m++;
                }
        m = retain_best(c, m, 2 * nPer[l]);
        for (int i = 0; i < m; i++) { /* HarrisResponses, block 7 */
            int a = 0, b = 0, cc = 0;
            for (int k = 0; k < 49; k++) {
                const uint8_t *p = lv[l] + (size_t) (c[i].y - 3 + k / 7) * W + (c[i].x - 3 + k % 7);
                int Ix = (p[1] - p[-1]) * 2 + (p[-W + 1] - p[-W - 1]) + (p[W + 1] - p[W - 1]);
                int Iy = (p[W] - p[-W]) * 2 + (p[W - 1] - p[-W - 1]) + (p[W + 1] - p[-W + 1]);
                a += Ix * Ix;
                b += Iy * Iy;
                cc += Ix * Iy;
            }
            volatile float fa = (float) a, fb = (float) b, fc = (float) cc;
            volatile float t1 = fa * fb, t2 = fc * fc, s = fa + fb;
            volatile float t3 = hk * s;
            volatile float t4 = t3 * s;
            volatile float d1 = t1 - t2;
            volatile float d2 = d1 - t4;
            c[i].resp = d2 * scale_sq_sq;
        }
        m = retain_best(c, m, nPer[l]);
        uint8_t *blur = NULL;
        if (doDescribe && m > 0) {
            blur = (uint8_t *) malloc((size_t) W * H);
            orc_orb_blur(lv[l], W, H, blur);
        }
        for (int i = 0; i < m; i++) {
            const uint8_t *ctr = lv[l] + (size_t) c[i].y * W + c[i].x;
            int m01 = 0, m10 = 0;
            for (int u = -15; u <= 15; u++) m10 += u * ctr[u];
            for (int v = 1; v <= 15; v++) {
                int vs = 0, d = umax[v];
                for (int u = -d; u <= d; u++) {
                    int vp = ctr[u + v * W], vm = ctr[u - v * W];
                    vs += vp - vm;
                    m10 += u * (vp + vm);
                }
                m01 += v * vs;
            }
            float angle = fast_atan2f((float) m01, (float) m10);
            if (total < cap) {
                float *o = kp + 6 * (size_t) total;
                volatile float sx = (float) c[i].x * lscale[l], sy = (float) c[i].y * lscale[l];
                o[0] = sx;
                o[1] = sy;
                o[2] = 31 * lscale[l];
                o[3] = angle;
                o[4] = c[i].resp;
                o[5] = (float) l;
                if (doDescribe) {
                    float inv = 1.f / lscale[l];
                    volatile float bx = o[0] * inv, by = o[1] * inv;
                    int cx = cv_round_f(bx), cy = cv_round_f(by);
                    float ang = angle;
                    ang *= (float) (3.1415926535897932384626433832795 / 180.f);
                    float ca = (float) cos((double) ang), sa = (float) sin((double) ang);
                    brief256(blur, W, cx, cy, ca, sa, desc + 32 * (size_t) total);
                }
            }
            total++;
        }
        free(xy); free(sc); free(scratch); free(c); free(blur);
    }
    for (int l = 0; l < nlevels; l++) free(lv[l]);
    free(lv); free(lw); free(lh); free(lscale); free(nPer);
    return total;
}

/* ---------------------------------------------------------------------------------------------------------------
 * f2a: triangulation of the 2-D keypoints of a new keyframe against their first observation
 * (Mapper::triangulateTemporal, src/slam/src/mapper.cpp:246-287).  Pure per-point arithmetic in IEEE double, with the
 * reference's float conversions where it returns cv::Point2f (CameraCalibration::projectCamToImage,
 * camera_calibration.cpp:25-32) and cv::norm(Point2f) = sqrt of the double-promoted squares (core/types.hpp). */
static void tri_matvec(const double *R, const double *v, double *o) {
    for (int i = 0; i < 3; i++) o[i] = (R[3 * i] * v[0] + R[3 * i + 1] * v[1]) + R[3 * i + 2] * v[2];  /* Eigen 3-term redux order */
}
static double tri_dot(const double *a, const double *b) { return (a[0] * b[0] + a[1] * b[1]) + a[2] * b[2]; }
static void tri_project(const double *p, double fx, double fy, double cx, double cy, float *u, float *v) {
    const double iz = 1. / p[2], x = p[0] * iz, y = p[1] * iz;
    *u = (float) (fx * x + cx);
    *v = (float) (fy * y + cy);
}
static double tri_norm2f(float dx, float dy) { return sqrt((double) dx * (double) dx + (double) dy * (double) dy); }

void orc_triangulate(int n, const double *T, const int *group, const double *bvl, const double *bvr, const float *unpxl,
                     const float *unpxr, double fx, double fy, double cx, double cy, float maxReprojErr, double *lpt, double *wpt,
                     double *invDepth, uint8_t *status, double *parallax) {
    for (int i = 0; i < n; i++) {
        const double *G = T + 36 * (size_t) group[i];
        const double *Rlr = G, *tlr = G + 9, *Rrl = G + 12, *trl = G + 21, *Rwl = G + 24, *twl = G + 33;
        const double *f1 = bvl + 3 * i, *f2 = bvr + 3 * i;
        /* rotation-compensated parallax (:246-248) */
        double f2u[3];
        tri_matvec(Rlr, f2, f2u);
        float ru, rv;
        tri_project(f2u, fx, fy, cx, cy, &ru, &rv);
        parallax[i] = tri_norm2f(unpxl[2 * i] - ru, unpxl[2 * i + 1] - rv);
        /* opengv::triangulation::triangulate2 (methods.cpp:67-90), A.inverse() = Eigen's closed 2x2 form */
        const double b0 = tri_dot(tlr, f1), b1 = tri_dot(tlr, f2u);
        const double a00 = tri_dot(f1, f1), a10 = tri_dot(f1, f2u), a01 = -a10, a11 = -tri_dot(f2u, f2u);
        const double invdet = 1.0 / (a00 * a11 - a10 * a01);
        const double i00 = a11 * invdet, i10 = -a10 * invdet, i01 = -a01 * invdet, i11 = a00 * invdet;
        const double l0 = i00 * b0 + i01 * b1, l1 = i10 * b0 + i11 * b1;
        double lp[3], rp[3], wp[3], t[3];
        for (int k = 0; k < 3; k++) lp[k] = (l0 * f1[k] + (tlr[k] + l1 * f2u[k])) / 2;
        tri_matvec(Rrl, lp, t);
        for (int k = 0; k < 3; k++) rp[k] = t[k] + trl[k];
        tri_matvec(Rwl, lp, t);
        for (int k = 0; k < 3; k++) wp[k] = t[k] + twl[k];
        for (int k = 0; k < 3; k++) {
            lpt[3 * i + k] = lp[k];
            wpt[3 * i + k] = wp[k];
        }
        invDepth[i] = 1. / lp[2];
        uint8_t st = 0;
        if (lp[2] < 0.1 || rp[2] < 0.1) st = 1; /* :256 */
        else {
            float lu, lv, pu, pv;
            tri_project(lp, fx, fy, cx, cy, &lu, &lv);
            tri_project(rp, fx, fy, cx, cy, &pu, &pv);
            const float lDist = (float) tri_norm2f(lu - unpxl[2 * i], lv - unpxl[2 * i + 1]);
            const float rDist = (float) tri_norm2f(pu - unpxr[2 * i], pv - unpxr[2 * i + 1]);
            if (lDist > maxReprojErr || rDist > maxReprojErr) st = 2; /* :272 */
        }
        status[i] = st;
    }
}

/* ---------------------------------------------------------------------------------------------------------------
 * f4a: CLAHE, 8-bit (imgproc/src/clahe.cpp).  Integer histogram / clip / redistribution, LUT = saturate_cast<uchar>(sum *
 * lutScale) (float product, round half to even), bilinear blend of the four neighbouring tile LUTs in float with the
 * reference's operation order. */
static int clahe_reflect101(int p, int len) {
    if (len == 1) return 0;
    while (p < 0 || p >= len) p = p < 0 ? -p : 2 * (len - 1) - p;
    return p;
}
static uint8_t clahe_sat_u8(float v) {
    int r = (int) lrintf(v); /* cvRound: round half to even (default rounding mode) */
    return (uint8_t) (r < 0 ? 0 : (r > 255 ? 255 : r));
}
void orc_clahe(const uint8_t *src, int w, int h, double clipLimitD, int tilesX, int tilesY, uint8_t *dst) {
    int ew = w, eh = h; /* extended size (copyMakeBorder right / bottom, REFLECT_101) when not divisible (:364-384) */
    if (!(w % tilesX == 0 && h % tilesY == 0)) {
        ew = w + (tilesX - (w % tilesX));
        eh = h + (tilesY - (h % tilesY));
    }
    const int tw = ew / tilesX, th = eh / tilesY, total = tw * th;
    const float lutScale = (float) (256 - 1) / total;
    int clipLimit = 0;
    if (clipLimitD > 0.0) {
        clipLimit = (int) (clipLimitD * total / 256);
        if (clipLimit < 1) clipLimit = 1;
    }
    uint8_t *lut = (uint8_t *) malloc((size_t) tilesX * tilesY * 256);
    for (int k = 0; k < tilesX * tilesY; k++) {
        const int ty = k / tilesX, tx = k % tilesX;
        int hist[256] = {0};
        for (int y = 0; y < th; y++)
            for (int x = 0; x < tw; x++) {
                const int sx = clahe_reflect101(tx * tw + x, w), sy = clahe_reflect101(ty * th + y, h);
                hist[src[(size_t) sy * w + sx]]++;
            }
        if (clipLimit > 0) {
            int clipped = 0;
            for (int i = 0; i < 256; i++)
                if (hist[i] > clipLimit) {
                    clipped += hist[i] - clipLimit;
                    hist[i] = clipLimit;
                }
            const int batch = clipped / 256;
            int residual = clipped - batch * 256;
            for (int i = 0; i < 256; i++) hist[i] += batch;
            if (residual != 0) {
                int step = 256 / residual;
                if (step < 1) step = 1;
                for (int i = 0; i < 256 && residual > 0; i += step, residual--) hist[i]++;
            }
        }
        int sum = 0;
        for (int i = 0; i < 256; i++) {
            sum += hist[i];
            lut[(size_t) k * 256 + i] = clahe_sat_u8((float) sum * lutScale);
        }
    }
    const float inv_tw = 1.0f / tw, inv_th = 1.0f / th;
    for (int y = 0; y < h; y++) {
        const float tyf = y * inv_th - 0.5f;
        int ty1 = (int) floorf(tyf), ty2 = ty1 + 1;
        const float ya = tyf - ty1, ya1 = 1.0f - ya;
        if (ty1 < 0) ty1 = 0;
        if (ty2 > tilesY - 1) ty2 = tilesY - 1;
        for (int x = 0; x < w; x++) {
            const float txf = x * inv_tw - 0.5f;
            int tx1 = (int) floorf(txf), tx2 = tx1 + 1;
            const float xa = txf - tx1, xa1 = 1.0f - xa;
            if (tx1 < 0) tx1 = 0;
            if (tx2 > tilesX - 1) tx2 = tilesX - 1;
            const int v = src[(size_t) y * w + x];
            const uint8_t *p1 = lut + (size_t) (ty1 * tilesX) * 256, *p2 = lut + (size_t) (ty2 * tilesX) * 256;
            const float res = (p1[tx1 * 256 + v] * xa1 + p1[tx2 * 256 + v] * xa) * ya1 + (p2[tx1 * 256 + v] * xa1 + p2[tx2 * 256 + v] * xa) * ya;
            dst[(size_t) y * w + x] = clahe_sat_u8(res);
        }
    }
    free(lut);
}

/* ---------------------------------------------------------------------------------------------------------------
 * f4b: lens distortion paths of CameraCalibration (camera_calibration.cpp:34-72), k = (k1, k2, p1, p2). */
void orc_undistort_points(const float *px, int n, double fx, double fy, double cx, double cy, const double *k, float *out) {
    const double ifx = 1. / fx, ify = 1. / fy;
    for (int i = 0; i < n; i++) {
        const double u = px[2 * i], v = px[2 * i + 1];
        double x = (u - cx) * ifx, y = (v - cy) * ify;
        const double x0 = x, y0 = y; /* the tilt matrices are identities: invProj * vec = the same values */
        for (int j = 0; j < 5; j++) { /* TermCriteria(MAX_ITER, 5, 0.01): five iterations, no error test */
            const double r2 = x * x + y * y;
            const double icdist = (1 + ((0 * r2 + 0) * r2 + 0) * r2) / (1 + ((0 * r2 + k[1]) * r2 + k[0]) * r2);
            if (icdist < 0) {
                x = (u - cx) * ifx;
                y = (v - cy) * ify;
                break;
            }
            const double deltaX = 2 * k[2] * x * y + k[3] * (r2 + 2 * x * x) + 0 * r2 + 0 * r2 * r2;
            const double deltaY = k[2] * (r2 + 2 * y * y) + 2 * k[3] * x * y + 0 * r2 + 0 * r2 * r2;
            x = (x0 - deltaX) * icdist;
            y = (y0 - deltaY) * icdist;
        }
        /* R = K (the reference passes Kcv_ in the R slot): xx = fx x + 0 y + cx, ww = 1 / (0 x + 0 y + 1) */
        const double xx = fx * x + 0 * y + cx, yy = 0 * x + fy * y + cy, ww = 1. / (0 * x + 0 * y + 1);
        out[2 * i] = (float) (xx * ww);
        out[2 * i + 1] = (float) (yy * ww);
    }
}

void orc_project_dist(const double *P, int n, double fx, double fy, double cx, double cy, const double *k, float *out) {
    for (int i = 0; i < n; i++) {
        const double iz = 1. / P[3 * i + 2];
        const float Xf = (float) (P[3 * i] * iz), Yf = (float) (P[3 * i + 1] * iz); /* cv::Point3f cvPoint(x, y, 1.0) */
        double x = (double) Xf, y = (double) Yf; /* R = I, t = 0, z = 1: exact */
        const double r2 = x * x + y * y, r4 = r2 * r2, r6 = r4 * r2;
        const double a1 = 2 * x * y, a2 = r2 + 2 * x * x, a3 = r2 + 2 * y * y;
        const double cdist = 1 + k[0] * r2 + k[1] * r4 + 0 * r6;
        const double icdist2 = 1. / (1 + 0 * r2 + 0 * r4 + 0 * r6);
        const double xd0 = x * cdist * icdist2 + k[2] * a1 + k[3] * a2 + 0 * r2 + 0 * r4;
        const double yd0 = y * cdist * icdist2 + k[2] * a3 + k[3] * a1 + 0 * r2 + 0 * r4;
        out[2 * i] = (float) (xd0 * fx + cx);
        out[2 * i + 1] = (float) (yd0 * fy + cy);
    }
}

/* ---------------------------------------------------------------------------------------------------------------
 * f1: Mapper::matchToMap (mapper.cpp:354-588).  Sophus::SE3d * point = Eigen::Quaternion::_transformVector + translation
 * (uv = q.vec x v; uv += uv; v + w uv + q.vec x uv), CameraCalibration::projectCamToImageDist (orc_project_dist), the 2x2-cell
 * neighbourhood of Frame::getSurroundingKeypoints (frame.cpp:313-341), the two-minimum scan with <= and the 0.9 ratio test,
 * and the per-keypoint arbitration in push order. */
static void mtm_transform(const double *q, const double *t, const double *v, double *o) {
    const double uv0 = q[1] * v[2] - q[2] * v[1], uv1 = q[2] * v[0] - q[0] * v[2], uv2 = q[0] * v[1] - q[1] * v[0];
    const double u0 = uv0 + uv0, u1 = uv1 + uv1, u2 = uv2 + uv2;
    const double c0 = q[1] * u2 - q[2] * u1, c1 = q[2] * u0 - q[0] * u2, c2 = q[0] * u1 - q[1] * u0;
    o[0] = ((v[0] + q[3] * u0) + c0) + t[0];
    o[1] = ((v[1] + q[3] * u1) + c1) + t[1];
    o[2] = ((v[2] + q[3] * u2) + c2) + t[2];
}

int orc_match_to_map(const double *calib, int cellSize, int numCellsW, int gridCells, const int *cellPtr, const int *cellMp, int nKf,
                     const double *kfQ, const double *kfT, int nMp, const double *mpWpt, const uint8_t *mpIs3d, const int *obsPtr,
                     const int *obsKf, const float *obsPx, const uint8_t *obsDesc, int frameKf, int numKeypoints3d, int nLocal,
                     const int *local, float maxProjErr, float distRatio, int *matchOfMp) {
    return orc_match_to_map_flags(calib, cellSize, numCellsW, gridCells, cellPtr, cellMp, nKf, kfQ, kfT, nMp, mpWpt, mpIs3d, NULL, obsPtr, obsKf,
                                  obsPx, obsDesc, NULL, frameKf, numKeypoints3d, nLocal, local, maxProjErr, distRatio, matchOfMp);
}

/* The same for a LIVE map, where a keyframe may hold a keypoint it could not describe (within 31 px of the border,
 * feature_extractor.cpp:191-209): mpHasDesc[m] = !MapPoint::desc_.empty(), obsHasDesc[o] = mapKeyframeDescriptors_ has an entry for
 * that observation's keyframe.  NULL flags = every observation carries a descriptor. */
int orc_match_to_map_flags(const double *calib, int cellSize, int numCellsW, int gridCells, const int *cellPtr, const int *cellMp, int nKf,
                           const double *kfQ, const double *kfT, int nMp, const double *mpWpt, const uint8_t *mpIs3d,
                           const uint8_t *mpHasDesc, const int *obsPtr, const int *obsKf, const float *obsPx, const uint8_t *obsDesc,
                           const uint8_t *obsHasDesc, int frameKf, int numKeypoints3d, int nLocal, const int *local, float maxProjErr,
                           float distRatio, int *matchOfMp) {
    (void) nKf;
    const double fx = calib[0], fy = calib[1], cx = calib[2], cy = calib[3], imgW = calib[8], imgH = calib[9];
    const float fovV = 0.5 * imgH / fy, fovH = 0.5 * imgW / fx;
    const float maxRadFov = fovH > fovV ? atanf(fovH) : atanf(fovV);
    const float view_th = cosf(maxRadFov);
    float maxPxDist = maxProjErr;
    if (numKeypoints3d < 30) maxPxDist *= 2.;
    int *frameObs = (int *) malloc(sizeof(int) * (size_t) nMp); /* observation of map point m in the frame, or -1 */
    float *bestDistOfKp = (float *) malloc(sizeof(float) * (size_t) nMp);
    for (int m = 0; m < nMp; m++) {
        frameObs[m] = -1;
        for (int o = obsPtr[m]; o < obsPtr[m + 1]; o++)
            if (obsKf[o] == frameKf) frameObs[m] = o;
        matchOfMp[m] = -1;
        bestDistOfKp[m] = 1024;
    }
    int nMatches = 0;
    for (int li = 0; li < nLocal; li++) {
        const int M = local[li];
        if (frameObs[M] >= 0) continue;                                /* frame.isObservingKeypoint (:393) */
        if (!mpIs3d[M] || !(mpHasDesc ? mpHasDesc[M] : obsPtr[M] != obsPtr[M + 1])) continue;       /* !is3d_ || desc_.empty() (:404) */
        const double *wpt = mpWpt + 3 * (size_t) M;
        double campt[3];
        mtm_transform(kfQ + 4 * (size_t) frameKf, kfT + 3 * (size_t) frameKf, wpt, campt);
        if (campt[2] < 0.1) continue;
        const float view_angle = (float) (campt[2] / sqrt((campt[0] * campt[0] + campt[1] * campt[1]) + campt[2] * campt[2]));
        if (fabsf(view_angle) < view_th) continue;
        float proj[2];
        orc_project_dist(campt, 1, fx, fy, cx, cy, calib + 4, proj);
        if (!(proj[0] >= 0 && proj[1] >= 0 && proj[0] < imgW && proj[1] < imgH)) continue;
        const float minDist = 32 * distRatio * 8.;
        int bestId = -1, secId = -1;
        float bestDist = minDist, secDist = minDist;
        const int rkp = (int) floorf(proj[1] / (float) cellSize), ckp = (int) floorf(proj[0] / (float) cellSize);
        for (int r = rkp - 1; r < rkp + 1; r++)
            for (int c = ckp - 1; c < ckp + 1; c++) {
                const int idx = r * numCellsW + c;
                if (r < 0 || c < 0 || idx > gridCells) continue;
                for (int e = cellPtr[idx]; e < cellPtr[idx + 1]; e++) {
                    const int K = cellMp[e], ko = frameObs[K];
                    const float pxDist = (float) sqrt((double) (proj[0] - obsPx[2 * ko]) * (double) (proj[0] - obsPx[2 * ko]) +
                                                      (double) (proj[1] - obsPx[2 * ko + 1]) * (double) (proj[1] - obsPx[2 * ko + 1]));
                    if (pxDist > maxPxDist) continue;
                    if (mpHasDesc && !mpHasDesc[K]) continue; /* kpMapPoint->desc_.empty() (:465-468) */
                    int cand = 1; /* never both observed in one keyframe (:474-485); both lists ascend */
                    for (int a = obsPtr[K]; a < obsPtr[K + 1] && cand; a++)
                        for (int b = obsPtr[M]; b < obsPtr[M + 1]; b++)
                            if (obsKf[a] == obsKf[b]) {
                                cand = 0;
                                break;
                            }
                    if (!cand) continue;
                    float coProj = 0.;
                    size_t nCo = 0;
                    for (int a = obsPtr[K]; a < obsPtr[K + 1]; a++) {
                        double cp[3];
                        float pp[2];
                        mtm_transform(kfQ + 4 * (size_t) obsKf[a], kfT + 3 * (size_t) obsKf[a], wpt, cp);
                        orc_project_dist(cp, 1, fx, fy, cx, cy, calib + 4, pp);
                        const float dx = obsPx[2 * a] - pp[0], dy = obsPx[2 * a + 1] - pp[1];
                        coProj += sqrt((double) dx * (double) dx + (double) dy * (double) dy);
                        nCo++;
                    }
                    if (coProj / nCo > maxPxDist) continue;
                    float dist = 1000.0;
                    for (int a = obsPtr[M]; a < obsPtr[M + 1]; a++)
                        for (int b = obsPtr[K]; b < obsPtr[K + 1]; b++) {
                            if (obsHasDesc && !(obsHasDesc[a] && obsHasDesc[b])) continue;
                            const float d = (float) orc_hamming256(obsDesc + 32 * (size_t) a, obsDesc + 32 * (size_t) b);
                            if (d < dist) dist = d;
                        }
                    if (dist <= bestDist) {
                        secDist = bestDist;
                        secId = bestId;
                        bestDist = dist;
                        bestId = K;
                    } else if (dist <= secDist) {
                        secDist = dist;
                        secId = K;
                    }
                }
            }
        if (bestId != -1 && secId != -1)
            if (0.9 * secDist < bestDist) bestId = -1;
        if (bestId < 0) continue;
        if (bestDist <= bestDistOfKp[bestId]) { /* arbitration per keypoint, in push order, <= (:563-578) */
            if (matchOfMp[bestId] < 0) nMatches++;
            bestDistOfKp[bestId] = bestDist;
            matchOfMp[bestId] = M;
        }
    }
    free(frameObs);
    free(bestDistOfKp);
    return nMatches;
}
