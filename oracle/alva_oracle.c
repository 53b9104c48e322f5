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
