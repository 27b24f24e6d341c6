// orb_ref — CPU restatement of the reference's ORB front end.  TEST INFRASTRUCTURE ONLY:
// see orb_ref.h for what it follows and for the "parity unpinned" statement.
//
// Build: g++ -O2 -ffp-contract=off (no -march=native): the reference's CMake sets no
// optimisation or architecture flags (CMakeLists.txt:6), so its float arithmetic is plain
// IEEE single/double without fused multiply-add.
#include "orb_ref.h"

#include <algorithm>
#include <climits>
#include <cmath>
#include <cfloat>
#include <cstdlib>
#include <cstring>
#include <vector>

namespace {

const int PATCH_SIZE = 31;       // src/ORBextractor.cc:75
const int HALF_PATCH_SIZE = 15;  // :76
const int EDGE_THRESHOLD = 16;   // :77
const float HARRIS_K = 0.04f;    // :73

const signed char kPattern[1024] = {
#include "../include/orb_pattern.inc"
};

struct KeyPoint {
    float x, y, size, angle, response;
    int octave, class_id;
};

// ---------------------------------------------------------------------------------------------
// OpenCV 2.4 rounding helpers.
inline int cvRound(double v) { return (int)lrint(v); }  // round half to even (cvtsd2si)
inline int cvFloor(double v) { int i = (int)v; return i - (i > v); }
inline int cvCeil(double v) { int i = (int)v; return i + (i < v); }
inline unsigned char sat_u8(int v) { return (unsigned char)(v < 0 ? 0 : v > 255 ? 255 : v); }
inline short sat_s16_from_float(float v) {
    int i = cvRound(v);
    return (short)(i < SHRT_MIN ? SHRT_MIN : i > SHRT_MAX ? SHRT_MAX : i);
}

// cv::borderInterpolate for BORDER_REFLECT_101 (single reflection is enough: margin 16 < n).
inline int reflect101(int p, int n) {
    if (n == 1) return 0;
    while (p < 0 || p >= n) {
        if (p < 0) p = -p;
        else p = 2 * n - 2 - p;
    }
    return p;
}

// cv::resize(..., INTER_LINEAR) for 8UC1, scalar fixed-point path of OpenCV 2.4 imgproc/imgwarp.cpp:
// 11-bit coefficients, horizontal pass in int32, vertical pass with the >>4, >>16, +2, >>2 sequence.
void resize_linear_u8(const unsigned char* src, int sw, int sh, ptrdiff_t sstride,
                      unsigned char* dst, int dw, int dh, ptrdiff_t dstride) {
    const int ONE = 2048;
    double inv_scale_x = (double)dw / sw, inv_scale_y = (double)dh / sh;
    double scale_x = 1. / inv_scale_x, scale_y = 1. / inv_scale_y;
    std::vector<int> xofs(dw), yofs(dh);
    std::vector<short> alpha(2 * dw), beta(2 * dh);
    int xmax = dw;
    for (int dx = 0; dx < dw; dx++) {
        float fx = (float)((dx + 0.5) * scale_x - 0.5);
        int sx = cvFloor(fx);
        fx -= sx;
        if (sx < 0) { fx = 0; sx = 0; }
        if (sx + 1 >= sw) {
            xmax = std::min(xmax, dx);
            if (sx >= sw - 1) { fx = 0; sx = sw - 1; }
        }
        xofs[dx] = sx;
        alpha[2 * dx] = sat_s16_from_float((1.f - fx) * ONE);
        alpha[2 * dx + 1] = sat_s16_from_float(fx * ONE);
    }
    for (int dy = 0; dy < dh; dy++) {
        float fy = (float)((dy + 0.5) * scale_y - 0.5);
        int sy = cvFloor(fy);
        fy -= sy;
        yofs[dy] = sy;
        beta[2 * dy] = sat_s16_from_float((1.f - fy) * ONE);
        beta[2 * dy + 1] = sat_s16_from_float(fy * ONE);
    }
    std::vector<int> row0(dw), row1(dw);
    for (int dy = 0; dy < dh; dy++) {
        int sy0 = std::min(std::max(yofs[dy], 0), sh - 1);
        int sy1 = std::min(std::max(yofs[dy] + 1, 0), sh - 1);
        const unsigned char* S0 = src + (ptrdiff_t)sy0 * sstride;
        const unsigned char* S1 = src + (ptrdiff_t)sy1 * sstride;
        for (int dx = 0; dx < dw; dx++) {
            int sx = xofs[dx];
            if (dx < xmax) {
                row0[dx] = S0[sx] * alpha[2 * dx] + S0[sx + 1] * alpha[2 * dx + 1];
                row1[dx] = S1[sx] * alpha[2 * dx] + S1[sx + 1] * alpha[2 * dx + 1];
            } else {
                row0[dx] = S0[sx] * ONE;
                row1[dx] = S1[sx] * ONE;
            }
        }
        short b0 = beta[2 * dy], b1 = beta[2 * dy + 1];
        unsigned char* D = dst + (ptrdiff_t)dy * dstride;
        for (int dx = 0; dx < dw; dx++)
            D[dx] = (unsigned char)((((b0 * (row0[dx] >> 4)) >> 16) + ((b1 * (row1[dx] >> 4)) >> 16) + 2) >> 2);
    }
}

// cv::copyMakeBorder(BORDER_REFLECT_101 [+ISOLATED]) in place on a padded plane whose interior
// (w x h at offset 16,16) is already filled.
void fill_border_reflect101(unsigned char* plane, int w, int h, ptrdiff_t stride) {
    const int E = EDGE_THRESHOLD;
    for (int y = 0; y < h + 2 * E; y++) {
        int sy = reflect101(y - E, h);
        const unsigned char* srow = plane + (ptrdiff_t)(sy + E) * stride + E;
        unsigned char* drow = plane + (ptrdiff_t)y * stride;
        for (int x = 0; x < w + 2 * E; x++) {
            if (y >= E && y < h + E && x >= E && x < w + E) continue;
            drow[x] = srow[reflect101(x - E, w)];
        }
    }
}

// ---------------------------------------------------------------------------------------------
// cv::FAST(img, kps, threshold, true): FAST-9/16 of OpenCV 2.4 features2d/fast.cpp + fast_score.cpp.
const int kCircle[16][2] = {{0, 3},  {1, 3},   {2, 2},   {3, 1},   {3, 0},  {3, -1}, {2, -2}, {1, -3},
                            {0, -3}, {-1, -3}, {-2, -2}, {-3, -1}, {-3, 0}, {-3, 1}, {-2, 2}, {-1, 3}};

int corner_score16(const unsigned char* ptr, const int pixel[25], int threshold) {
    const int K = 8, N = K * 3 + 1;
    int v = ptr[0];
    short d[N];
    for (int k = 0; k < N; k++) d[k] = (short)(v - ptr[pixel[k]]);
    int a0 = threshold;
    for (int k = 0; k < 16; k += 2) {
        int a = std::min((int)d[k + 1], (int)d[k + 2]);
        a = std::min(a, (int)d[k + 3]);
        if (a <= a0) continue;
        a = std::min(a, (int)d[k + 4]);
        a = std::min(a, (int)d[k + 5]);
        a = std::min(a, (int)d[k + 6]);
        a = std::min(a, (int)d[k + 7]);
        a = std::min(a, (int)d[k + 8]);
        a0 = std::max(a0, std::min(a, (int)d[k]));
        a0 = std::max(a0, std::min(a, (int)d[k + 9]));
    }
    int b0 = -a0;
    for (int k = 0; k < 16; k += 2) {
        int b = std::max((int)d[k + 1], (int)d[k + 2]);
        b = std::max(b, (int)d[k + 3]);
        b = std::max(b, (int)d[k + 4]);
        b = std::max(b, (int)d[k + 5]);
        if (b >= b0) continue;
        b = std::max(b, (int)d[k + 6]);
        b = std::max(b, (int)d[k + 7]);
        b = std::max(b, (int)d[k + 8]);
        b0 = std::min(b0, std::max(b, (int)d[k]));
        b0 = std::min(b0, std::max(b, (int)d[k + 9]));
    }
    return -b0 - 1;
}

void fast9_16(const unsigned char* img, int rows, int cols, ptrdiff_t step, int threshold, bool nms,
              std::vector<KeyPoint>& keypoints) {
    const int K = 8, N = 25;
    int pixel[25];
    for (int k = 0; k < 16; k++) pixel[k] = kCircle[k][0] + kCircle[k][1] * (int)step;
    for (int k = 16; k < 25; k++) pixel[k] = pixel[k - 16];
    keypoints.clear();
    threshold = std::min(std::max(threshold, 0), 255);
    unsigned char threshold_tab[512];
    for (int i = -255; i <= 255; i++)
        threshold_tab[i + 255] = (unsigned char)(i < -threshold ? 1 : i > threshold ? 2 : 0);

    std::vector<unsigned char> scorebuf((size_t)std::max(cols, 1) * 3, 0);
    std::vector<int> posbuf((size_t)(std::max(cols, 1) + 1) * 3, 0);
    unsigned char* buf[3] = {&scorebuf[0], &scorebuf[0] + cols, &scorebuf[0] + 2 * cols};
    int* cpbuf[3] = {&posbuf[0] + 1, &posbuf[0] + (cols + 1) + 1, &posbuf[0] + 2 * (cols + 1) + 1};

    for (int i = 3; i < rows - 2; i++) {
        const unsigned char* ptr = img + (ptrdiff_t)i * step + 3;
        unsigned char* curr = buf[(i - 3) % 3];
        int* cornerpos = cpbuf[(i - 3) % 3];
        memset(curr, 0, cols);
        int ncorners = 0;
        if (i < rows - 3) {
            for (int j = 3; j < cols - 3; j++, ptr++) {
                int v = ptr[0];
                const unsigned char* tab = &threshold_tab[0] - v + 255;
                int d = tab[ptr[pixel[0]]] | tab[ptr[pixel[8]]];
                if (d == 0) continue;
                d &= tab[ptr[pixel[2]]] | tab[ptr[pixel[10]]];
                d &= tab[ptr[pixel[4]]] | tab[ptr[pixel[12]]];
                d &= tab[ptr[pixel[6]]] | tab[ptr[pixel[14]]];
                if (d == 0) continue;
                d &= tab[ptr[pixel[1]]] | tab[ptr[pixel[9]]];
                d &= tab[ptr[pixel[3]]] | tab[ptr[pixel[11]]];
                d &= tab[ptr[pixel[5]]] | tab[ptr[pixel[13]]];
                d &= tab[ptr[pixel[7]]] | tab[ptr[pixel[15]]];
                if (d & 1) {
                    int vt = v - threshold, count = 0;
                    for (int k = 0; k < N; k++) {
                        int x = ptr[pixel[k]];
                        if (x < vt) {
                            if (++count > K) {
                                cornerpos[ncorners++] = j;
                                if (nms) curr[j] = (unsigned char)corner_score16(ptr, pixel, threshold);
                                break;
                            }
                        } else
                            count = 0;
                    }
                }
                if (d & 2) {
                    int vt = v + threshold, count = 0;
                    for (int k = 0; k < N; k++) {
                        int x = ptr[pixel[k]];
                        if (x > vt) {
                            if (++count > K) {
                                cornerpos[ncorners++] = j;
                                if (nms) curr[j] = (unsigned char)corner_score16(ptr, pixel, threshold);
                                break;
                            }
                        } else
                            count = 0;
                    }
                }
            }
        }
        cornerpos[-1] = ncorners;
        if (i == 3) continue;
        const unsigned char* prev = buf[(i - 4 + 3) % 3];
        const unsigned char* pprev = buf[(i - 5 + 3) % 3];
        cornerpos = cpbuf[(i - 4 + 3) % 3];
        ncorners = cornerpos[-1];
        for (int k = 0; k < ncorners; k++) {
            int j = cornerpos[k];
            int score = prev[j];
            if (!nms || (score > prev[j + 1] && score > prev[j - 1] && score > pprev[j - 1] && score > pprev[j] &&
                         score > pprev[j + 1] && score > curr[j - 1] && score > curr[j] && score > curr[j + 1])) {
                KeyPoint kp = {(float)j, (float)(i - 1), 7.f, -1.f, (float)score, 0, -1};
                keypoints.push_back(kp);
            }
        }
    }
}

// Closed form used by the device path: M = max(A, B) with A = max over the 16 arcs of nine
// contiguous circle pixels of min(v - I_k) and B the same for (I_k - v).  A pixel is a corner at
// threshold t iff M > t and its OpenCV score is then M - 1.
int fast_measure(const unsigned char* c, ptrdiff_t step) {
    int v = c[0], d[16];
    for (int k = 0; k < 16; k++) d[k] = v - c[kCircle[k][0] + kCircle[k][1] * step];
    int A = INT_MIN, B = INT_MIN;
    for (int s = 0; s < 16; s++) {
        int mn = INT_MAX, mx = INT_MIN;
        for (int k = 0; k < 9; k++) {
            mn = std::min(mn, d[(s + k) & 15]);
            mx = std::max(mx, d[(s + k) & 15]);
        }
        A = std::max(A, mn);
        B = std::max(B, -mx);
    }
    return std::max(A, B);
}

// cv::KeyPointsFilter::retainBest of OpenCV 2.4 (features2d/keypoint.cpp).
struct ResponseGreater {
    bool operator()(const KeyPoint& a, const KeyPoint& b) const { return a.response > b.response; }
};
struct ResponseAtLeast {
    float th;
    bool operator()(const KeyPoint& k) const { return k.response >= th; }
};
void retain_best(std::vector<KeyPoint>& kps, int n_points) {
    if (n_points >= 0 && kps.size() > (size_t)n_points) {
        if (n_points == 0) { kps.clear(); return; }
        std::nth_element(kps.begin(), kps.begin() + n_points, kps.end(), ResponseGreater());
        float ambiguous = kps[n_points - 1].response;
        ResponseAtLeast pred = {ambiguous};
        std::vector<KeyPoint>::iterator new_end = std::partition(kps.begin() + n_points, kps.end(), pred);
        kps.resize(new_end - kps.begin());
    }
}

// cv::fastAtan2 of OpenCV 2.4 (core/mathfuncs.cpp), degrees in [0, 360).
float fast_atan2(float y, float x) {
    static const float p1 = 0.9997878412794807f * (float)(180 / M_PI);
    static const float p3 = -0.3258083974640975f * (float)(180 / M_PI);
    static const float p5 = 0.1555786518463281f * (float)(180 / M_PI);
    static const float p7 = -0.04432655554792128f * (float)(180 / M_PI);
    float ax = std::abs(x), ay = std::abs(y);
    float a, c, c2;
    if (ax >= ay) {
        c = ay / (ax + (float)DBL_EPSILON);
        c2 = c * c;
        a = (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
    } else {
        c = ax / (ay + (float)DBL_EPSILON);
        c2 = c * c;
        a = 90.f - (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
    }
    if (x < 0) a = 180.f - a;
    if (y < 0) a = 360.f - a;
    return a;
}

// 0 = scalar column pass (the profile every parity test uses), 1 = SSE2 column pass; see gauss7_roi.
int g_gauss_profile = 0;

// cv::getGaussianKernel(7, 2, CV_32F) converted to 8-bit fixed point as
// createSeparableLinearFilter does for CV_8U smoothing kernels (x256, cvRound).
void gauss_taps(int taps[7]) {
    const int n = 7;
    const double sigma = 2.0;
    float cf[7];
    double scale2X = -0.5 / (sigma * sigma), sum = 0;
    for (int i = 0; i < n; i++) {
        double x = i - (n - 1) * 0.5;
        double t = std::exp(scale2X * x * x);
        cf[i] = (float)t;
        sum += cf[i];
    }
    sum = 1. / sum;
    for (int i = 0; i < n; i++) cf[i] = (float)(cf[i] * sum);
    for (int i = 0; i < n; i++) taps[i] = cvRound((double)cf[i] * 256);
}

// GaussianBlur(roi, roi, Size(7,7), 2, 2, BORDER_REFLECT_101) on the ROI view of a padded plane:
// the filter's border is the real 3-px ring of the parent buffer.  Row pass int32, column pass
// (sum + 2^15) >> 16 saturated to u8.  in/out are padded planes (w+32)x(h+32); the margin of out is
// copied from in (the in-place reference leaves it untouched).
void gauss7_roi(const unsigned char* in, int w, int h, unsigned char* out, const int taps[7]) {
    const int E = EDGE_THRESHOLD;
    const ptrdiff_t stride = w + 2 * E;
    memcpy(out, in, (size_t)stride * (h + 2 * E));
    std::vector<int> rowsum((size_t)w * (h + 6));
    for (int y = -3; y < h + 3; y++) {
        const unsigned char* s = in + (ptrdiff_t)(y + E) * stride + E;
        int* r = &rowsum[(size_t)(y + 3) * w];
        for (int x = 0; x < w; x++) {
            int acc = 0;
            for (int k = 0; k < 7; k++) acc += taps[k] * s[x + k - 3];
            r[x] = acc;
        }
    }
    // Column pass.  Profile 0 (the oracle's definition): the scalar SymmColumnFilter with FixedPtCastEx,
    // (sum + 2^15) >> 16.  Profile 1: what an SSE2 build of OpenCV 2.4 runs instead for the first w & ~3
    // columns (SymmColumnVec_32s8u, modules/imgproc/src/filter.cpp): the taps as float k / 65536, the centre row
    // times its tap, then for k = 1..3 the integer sum of the two mirrored rows converted to float times its tap,
    // accumulated in fp32, converted with cvtps2dq (round to nearest even) and saturated; the remaining
    // w & 3 columns take the scalar path.  Every product and every partial sum below 2^24 / 65536 is exact in
    // fp32, so the two profiles can only differ where the sum is an exact tie (sum mod 65536 == 32768).
    const int vec_cols = g_gauss_profile == 1 ? (w & ~3) : 0;
    float kf[4];
    for (int k = 0; k < 4; k++) kf[k] = (float)taps[3 + k] * (1.f / 65536.f);
    for (int y = 0; y < h; y++) {
        unsigned char* d = out + (ptrdiff_t)(y + E) * stride + E;
        for (int x = 0; x < w; x++) {
            if (x < vec_cols) {
                float s0 = (float)rowsum[(size_t)(y + 3) * w + x] * kf[0] + 0.f;
                for (int k = 1; k <= 3; k++) {
                    const int pair = rowsum[(size_t)(y + 3 + k) * w + x] + rowsum[(size_t)(y + 3 - k) * w + x];
                    s0 = s0 + (float)pair * kf[k];
                }
                d[x] = sat_u8((int)std::nearbyint(s0));  // default rounding mode: to nearest even, as cvtps2dq
                continue;
            }
            int acc = 0;
            for (int k = 0; k < 7; k++) acc += taps[k] * rowsum[(size_t)(y + k) * w + x];
            d[x] = sat_u8((acc + 32768) >> 16);
        }
    }
}

// IC_Angle (src/ORBextractor.cc:124-151).
float ic_angle(const unsigned char* center, ptrdiff_t step, const int* u_max) {
    int m_01 = 0, m_10 = 0;
    for (int u = -HALF_PATCH_SIZE; u <= HALF_PATCH_SIZE; ++u) m_10 += u * center[u];
    for (int v = 1; v <= HALF_PATCH_SIZE; ++v) {
        int v_sum = 0;
        int d = u_max[v];
        for (int u = -d; u <= d; ++u) {
            int val_plus = center[u + v * step], val_minus = center[u - v * step];
            v_sum += (val_plus - val_minus);
            m_10 += u * (val_plus + val_minus);
        }
        m_01 += v * v_sum;
    }
    return fast_atan2((float)m_01, (float)m_10);
}

// computeOrbDescriptor (src/ORBextractor.cc:154-194).
const float factorPI = (float)(M_PI / 180.f);
void orb_descriptor(const unsigned char* center, ptrdiff_t step, float kp_angle, unsigned char* desc) {
    float angle = (float)kp_angle * factorPI;
    float a = (float)cosf(angle), b = (float)sinf(angle);
    const signed char* pat = kPattern;
    for (int i = 0; i < 32; ++i, pat += 32) {
        int val = 0;
        for (int j = 0; j < 8; j++) {
            int x0 = pat[4 * j], y0 = pat[4 * j + 1], x1 = pat[4 * j + 2], y1 = pat[4 * j + 3];
            int t0 = center[cvRound(x0 * b + y0 * a) * step + cvRound(x0 * a - y0 * b)];
            int t1 = center[cvRound(x1 * b + y1 * a) * step + cvRound(x1 * a - y1 * b)];
            val |= (t0 < t1) << j;
        }
        desc[i] = (unsigned char)val;
    }
}

// HarrisResponses (src/ORBextractor.cc:79-120) on a cell image, block size 7.
void harris_responses(const unsigned char* img, ptrdiff_t step, std::vector<KeyPoint>& pts, int blockSize, float harris_k) {
    int r = blockSize / 2;
    float scale = (1 << 2) * blockSize * 255.0f;
    scale = 1.0f / scale;
    float scale_sq_sq = scale * scale * scale * scale;
    for (size_t p = 0; p < pts.size(); p++) {
        int x0 = cvRound(pts[p].x - r), y0 = cvRound(pts[p].y - r);
        const unsigned char* ptr0 = img + (ptrdiff_t)y0 * step + x0;
        int a = 0, b = 0, c = 0;
        for (int i = 0; i < blockSize; i++)
            for (int j = 0; j < blockSize; j++) {
                const unsigned char* ptr = ptr0 + i * step + j;
                int Ix = (ptr[1] - ptr[-1]) * 2 + (ptr[-step + 1] - ptr[-step - 1]) + (ptr[step + 1] - ptr[step - 1]);
                int Iy = (ptr[step] - ptr[-step]) * 2 + (ptr[step - 1] - ptr[-step - 1]) + (ptr[step + 1] - ptr[-step + 1]);
                a += Ix * Ix;
                b += Iy * Iy;
                c += Ix * Iy;
            }
        pts[p].response = ((float)a * b - (float)c * c - harris_k * ((float)a + b) * ((float)a + b)) * scale_sq_sq;
    }
}

// Third value of the stage taps: the FAST score as an integer, or the IEEE bits of the float Harris response.
inline int tap_response(int scoreType, float r) {
    if (scoreType != 0) return (int)r;
    int bits;
    memcpy(&bits, &r, sizeof bits);
    return bits;
}

struct LevelTap {
    int w, h;
    int cols, rows, cellW, cellH, nfeaturesCell;
    std::vector<unsigned char> plane, blurred;
    std::vector<int> cand_counts, cand_xyr, retain;
    std::vector<int> kp_xyr;
};

}  // namespace

struct orbref_extractor {
    int nfeatures;
    double scaleFactor;  // include/ORBextractor.h:62 — a double member initialised from a float
    int nlevels, scoreType, fastTh;
    std::vector<int> mnFeaturesPerLevel, umax;
    std::vector<float> mvScaleFactor, mvInvScaleFactor;
    int gtaps[7];
    std::vector<LevelTap> lv;
};

extern "C" {

orbref_extractor* orbref_create(int _nfeatures, float _scaleFactor, int _nlevels, int _scoreType, int _fastTh) {
    if (_nlevels < 1 || _nfeatures < 1) return NULL;
    orbref_extractor* e = new orbref_extractor;
    e->nfeatures = _nfeatures;
    e->scaleFactor = _scaleFactor;
    e->nlevels = _nlevels;
    e->scoreType = _scoreType;
    e->fastTh = _fastTh;
    const int nlevels = _nlevels;
    // src/ORBextractor.cc:462-471
    e->mvScaleFactor.resize(nlevels);
    e->mvScaleFactor[0] = 1;
    for (int i = 1; i < nlevels; i++) e->mvScaleFactor[i] = e->mvScaleFactor[i - 1] * e->scaleFactor;
    float invScaleFactor = 1.0f / e->scaleFactor;
    e->mvInvScaleFactor.resize(nlevels);
    e->mvInvScaleFactor[0] = 1;
    for (int i = 1; i < nlevels; i++) e->mvInvScaleFactor[i] = e->mvInvScaleFactor[i - 1] * invScaleFactor;
    // :476-487
    e->mnFeaturesPerLevel.resize(nlevels);
    float factor = (float)(1.0 / e->scaleFactor);
    float nDesiredFeaturesPerScale = e->nfeatures * (1 - factor) / (1 - (float)pow((double)factor, (double)nlevels));
    int sumFeatures = 0;
    for (int level = 0; level < nlevels - 1; level++) {
        e->mnFeaturesPerLevel[level] = cvRound(nDesiredFeaturesPerScale);
        sumFeatures += e->mnFeaturesPerLevel[level];
        nDesiredFeaturesPerScale *= factor;
    }
    e->mnFeaturesPerLevel[nlevels - 1] = std::max(e->nfeatures - sumFeatures, 0);
    // :495-510
    e->umax.resize(HALF_PATCH_SIZE + 1);
    int v, v0, vmax = cvFloor(HALF_PATCH_SIZE * sqrt(2.f) / 2 + 1);
    int vmin = cvCeil(HALF_PATCH_SIZE * sqrt(2.f) / 2);
    const double hp2 = HALF_PATCH_SIZE * HALF_PATCH_SIZE;
    for (v = 0; v <= vmax; ++v) e->umax[v] = cvRound(sqrt(hp2 - v * v));
    for (v = HALF_PATCH_SIZE, v0 = 0; v >= vmin; --v) {
        while (e->umax[v0] == e->umax[v0 + 1]) ++v0;
        e->umax[v] = v0;
        ++v0;
    }
    gauss_taps(e->gtaps);
    e->lv.resize(nlevels);
    return e;
}

void orbref_destroy(orbref_extractor* e) { delete e; }
int orbref_levels(const orbref_extractor* e) { return e->nlevels; }
float orbref_scale_factor(const orbref_extractor* e) { return (float)e->scaleFactor; }

int orbref_table(const orbref_extractor* e, int which, void* out) {
    switch (which) {
        case 0: memcpy(out, &e->mnFeaturesPerLevel[0], sizeof(int) * e->nlevels); return e->nlevels;
        case 1: memcpy(out, &e->mvScaleFactor[0], sizeof(float) * e->nlevels); return e->nlevels;
        case 2: memcpy(out, &e->mvInvScaleFactor[0], sizeof(float) * e->nlevels); return e->nlevels;
        case 3: memcpy(out, &e->umax[0], sizeof(int) * 16); return 16;
        case 4: memcpy(out, e->gtaps, sizeof(int) * 7); return 7;
    }
    return -1;
}

int orbref_extract(orbref_extractor* e, const uint8_t* img, int rows, int cols, ptrdiff_t stride,
                   OrbRefKeypoint* kp_out, uint8_t* desc_out, int cap, int* n_out) {
    if (img == NULL || rows <= 0 || cols <= 0) return 0;  // :721-722
    const int nlevels = e->nlevels;
    const int E = EDGE_THRESHOLD;

    // ---- ComputePyramid (:781-822) -----------------------------------------------------------
    for (int level = 0; level < nlevels; ++level) {
        float scale = e->mvInvScaleFactor[level];
        int w = cvRound((float)cols * scale), h = cvRound((float)rows * scale);
        LevelTap& L = e->lv[level];
        L.w = w;
        L.h = h;
        if (w < 1 || h < 1) return -1;
        const ptrdiff_t ps = w + 2 * E;
        L.plane.assign((size_t)ps * (h + 2 * E), 0);
        unsigned char* roi = &L.plane[0] + E * ps + E;
        if (level != 0) {
            const LevelTap& P = e->lv[level - 1];
            const ptrdiff_t pps = P.w + 2 * E;
            resize_linear_u8(&P.plane[0] + E * pps + E, P.w, P.h, pps, roi, w, h, ps);
        } else {
            for (int y = 0; y < h; y++) memcpy(roi + y * ps, img + (ptrdiff_t)y * stride, w);
        }
        fill_border_reflect101(&L.plane[0], w, h, ps);
    }

    // ---- ComputeKeyPoints (:522-707) ---------------------------------------------------------
    std::vector<std::vector<KeyPoint> > allKeypoints(nlevels);
    float imageRatio = (float)e->lv[0].w / e->lv[0].h;
    for (int level = 0; level < nlevels; ++level) {
        LevelTap& L = e->lv[level];
        const ptrdiff_t ps = L.w + 2 * E;
        const unsigned char* roi = &L.plane[0] + E * ps + E;
        const int nDesiredFeatures = e->mnFeaturesPerLevel[level];
        const int levelCols = std::sqrt((float)nDesiredFeatures / (5 * imageRatio));
        const int levelRows = imageRatio * levelCols;
        if (levelCols < 1 || levelRows < 1) return -1;
        const int minBorderX = EDGE_THRESHOLD, minBorderY = minBorderX;
        const int maxBorderX = L.w - EDGE_THRESHOLD, maxBorderY = L.h - EDGE_THRESHOLD;
        const int W = maxBorderX - minBorderX, H = maxBorderY - minBorderY;
        if (W < 1 || H < 1) return -1;
        const int cellW = std::ceil((float)W / levelCols);
        const int cellH = std::ceil((float)H / levelRows);
        const int nCells = levelRows * levelCols;
        const int nfeaturesCell = std::ceil((float)nDesiredFeatures / nCells);
        L.cols = levelCols; L.rows = levelRows; L.cellW = cellW; L.cellH = cellH; L.nfeaturesCell = nfeaturesCell;

        std::vector<std::vector<std::vector<KeyPoint> > > cellKeyPoints(levelRows, std::vector<std::vector<KeyPoint> >(levelCols));
        std::vector<std::vector<int> > nToRetain(levelRows, std::vector<int>(levelCols));
        std::vector<std::vector<int> > nTotal(levelRows, std::vector<int>(levelCols));
        std::vector<std::vector<bool> > bNoMore(levelRows, std::vector<bool>(levelCols, false));
        std::vector<int> iniXCol(levelCols), iniYRow(levelRows);
        int nNoMore = 0, nToDistribute = 0;

        float hY = cellH + 6;
        for (int i = 0; i < levelRows; i++) {
            const float iniY = minBorderY + i * cellH - 3;
            iniYRow[i] = iniY;
            if (i == levelRows - 1) {
                hY = maxBorderY + 3 - iniY;
                if (hY <= 0) continue;
            }
            float hX = cellW + 6;
            for (int j = 0; j < levelCols; j++) {
                float iniX;
                if (i == 0) {
                    iniX = minBorderX + j * cellW - 3;
                    iniXCol[j] = iniX;
                } else
                    iniX = iniXCol[j];
                if (j == levelCols - 1) {
                    hX = maxBorderX + 3 - iniX;
                    if (hX <= 0) continue;
                }
                // rowRange/colRange of the ROI view: cv::Mat asserts the range lies inside it.
                if ((int)iniX < 0 || (int)iniY < 0 || (int)(iniX + hX) > L.w || (int)(iniY + hY) > L.h) return -1;
                const unsigned char* cell = roi + (ptrdiff_t)(int)iniY * ps + (int)iniX;
                std::vector<KeyPoint>& kc = cellKeyPoints[i][j];
                kc.reserve(nfeaturesCell * 5);
                fast9_16(cell, (int)hY, (int)hX, ps, e->fastTh, true, kc);
                if (kc.size() <= 3) {
                    kc.clear();
                    fast9_16(cell, (int)hY, (int)hX, ps, 7, true, kc);
                }
                // A cell that is not in the last row/column examines cellH x cellW pixels even when that
                // carries it past maxBorder (ceil() of the cell size times the cell count can exceed the
                // bordered extent).  A corner found there makes IC_Angle and the descriptor read rows or
                // columns outside the image: undefined in the reference, reported here as an error.
                for (size_t k = 0; k < kc.size(); k++)
                    if ((int)iniX + kc[k].x >= maxBorderX || (int)iniY + kc[k].y >= maxBorderY) return -2;
                if (e->scoreType == 0) harris_responses(cell, ps, kc, 7, HARRIS_K);
                const int nKeys = kc.size();
                nTotal[i][j] = nKeys;
                if (nKeys > nfeaturesCell) {
                    nToRetain[i][j] = nfeaturesCell;
                    bNoMore[i][j] = false;
                } else {
                    nToRetain[i][j] = nKeys;
                    nToDistribute += nfeaturesCell - nKeys;
                    bNoMore[i][j] = true;
                    nNoMore++;
                }
            }
        }
        // taps: raw candidates
        L.cand_counts.assign(nCells, 0);
        L.cand_xyr.clear();
        for (int i = 0; i < levelRows; i++)
            for (int j = 0; j < levelCols; j++) {
                L.cand_counts[i * levelCols + j] = (int)cellKeyPoints[i][j].size();
                for (size_t k = 0; k < cellKeyPoints[i][j].size(); k++) {
                    L.cand_xyr.push_back((int)cellKeyPoints[i][j][k].x);
                    L.cand_xyr.push_back((int)cellKeyPoints[i][j][k].y);
                    L.cand_xyr.push_back(tap_response(e->scoreType, cellKeyPoints[i][j][k].response));
                }
            }

        while (nToDistribute > 0 && nNoMore < nCells) {
            int nNewFeaturesCell = nfeaturesCell + std::ceil((float)nToDistribute / (nCells - nNoMore));
            nToDistribute = 0;
            for (int i = 0; i < levelRows; i++)
                for (int j = 0; j < levelCols; j++)
                    if (!bNoMore[i][j]) {
                        if (nTotal[i][j] > nNewFeaturesCell) {
                            nToRetain[i][j] = nNewFeaturesCell;
                            bNoMore[i][j] = false;
                        } else {
                            nToRetain[i][j] = nTotal[i][j];
                            nToDistribute += nNewFeaturesCell - nTotal[i][j];
                            bNoMore[i][j] = true;
                            nNoMore++;
                        }
                    }
        }
        L.retain.assign(nCells, 0);
        for (int i = 0; i < levelRows; i++)
            for (int j = 0; j < levelCols; j++) L.retain[i * levelCols + j] = nToRetain[i][j];

        std::vector<KeyPoint>& keypoints = allKeypoints[level];
        keypoints.reserve(nDesiredFeatures * 2);
        const int scaledPatchSize = PATCH_SIZE * e->mvScaleFactor[level];
        for (int i = 0; i < levelRows; i++)
            for (int j = 0; j < levelCols; j++) {
                std::vector<KeyPoint>& keysCell = cellKeyPoints[i][j];
                retain_best(keysCell, nToRetain[i][j]);
                if ((int)keysCell.size() > nToRetain[i][j]) keysCell.resize(nToRetain[i][j]);
                for (size_t k = 0, kend = keysCell.size(); k < kend; k++) {
                    keysCell[k].x += iniXCol[j];
                    keysCell[k].y += iniYRow[i];
                    keysCell[k].octave = level;
                    keysCell[k].size = scaledPatchSize;
                    keypoints.push_back(keysCell[k]);
                }
            }
        if ((int)keypoints.size() > nDesiredFeatures) {
            retain_best(keypoints, nDesiredFeatures);
            keypoints.resize(nDesiredFeatures);
        }
        L.kp_xyr.clear();
        for (size_t k = 0; k < keypoints.size(); k++) {
            L.kp_xyr.push_back((int)keypoints[k].x);
            L.kp_xyr.push_back((int)keypoints[k].y);
            L.kp_xyr.push_back(tap_response(e->scoreType, keypoints[k].response));
        }
    }
    // orientations (:705-706) on the unblurred pyramid
    for (int level = 0; level < nlevels; ++level) {
        LevelTap& L = e->lv[level];
        const ptrdiff_t ps = L.w + 2 * E;
        const unsigned char* roi = &L.plane[0] + E * ps + E;
        for (size_t k = 0; k < allKeypoints[level].size(); k++) {
            KeyPoint& kp = allKeypoints[level][k];
            kp.angle = ic_angle(roi + (ptrdiff_t)cvRound(kp.y) * ps + cvRound(kp.x), ps, &e->umax[0]);
        }
    }

    // ---- operator() tail (:733-778) ----------------------------------------------------------
    int nkeypoints = 0;
    for (int level = 0; level < nlevels; ++level) nkeypoints += (int)allKeypoints[level].size();
    if (nkeypoints > cap) return -1;
    int offset = 0;
    for (int level = 0; level < nlevels; ++level) {
        LevelTap& L = e->lv[level];
        std::vector<KeyPoint>& keypoints = allKeypoints[level];
        int n = (int)keypoints.size();
        L.blurred.clear();
        if (n == 0) continue;
        const ptrdiff_t ps = L.w + 2 * E;
        L.blurred.resize(L.plane.size());
        gauss7_roi(&L.plane[0], L.w, L.h, &L.blurred[0], e->gtaps);
        const unsigned char* roi = &L.blurred[0] + E * ps + E;
        for (int k = 0; k < n; k++) {
            const KeyPoint& kp = keypoints[k];
            orb_descriptor(roi + (ptrdiff_t)cvRound(kp.y) * ps + cvRound(kp.x), ps, kp.angle, desc_out + (size_t)(offset + k) * 32);
        }
        if (level != 0) {
            float scale = e->mvScaleFactor[level];
            for (int k = 0; k < n; k++) {
                keypoints[k].x *= scale;
                keypoints[k].y *= scale;
            }
        }
        for (int k = 0; k < n; k++) {
            OrbRefKeypoint o = {keypoints[k].x, keypoints[k].y, keypoints[k].size, keypoints[k].angle,
                                keypoints[k].response, keypoints[k].octave, keypoints[k].class_id};
            kp_out[offset + k] = o;
        }
        offset += n;
    }
    *n_out = nkeypoints;
    return 0;
}

int orbref_level_size(const orbref_extractor* e, int level, int* w, int* h) {
    if (level < 0 || level >= e->nlevels) return -1;
    *w = e->lv[level].w;
    *h = e->lv[level].h;
    return 0;
}

int orbref_plane(const orbref_extractor* e, int level, int blurred, uint8_t* out) {
    if (level < 0 || level >= e->nlevels) return -1;
    const std::vector<unsigned char>& p = blurred ? e->lv[level].blurred : e->lv[level].plane;
    if (p.empty()) return 0;
    memcpy(out, &p[0], p.size());
    return (int)p.size();
}

int orbref_level_keypoints(const orbref_extractor* e, int level, int32_t* xyr, int cap) {
    const std::vector<int>& v = e->lv[level].kp_xyr;
    int n = (int)v.size() / 3;
    if (n > cap) return -1;
    if (n) memcpy(xyr, &v[0], v.size() * sizeof(int));
    return n;
}

int orbref_level_grid(const orbref_extractor* e, int level, int32_t* out) {
    const LevelTap& L = e->lv[level];
    out[0] = L.cols; out[1] = L.rows; out[2] = L.cellW; out[3] = L.cellH; out[4] = L.nfeaturesCell;
    return 0;
}

int orbref_level_candidates(const orbref_extractor* e, int level, int32_t* counts, int32_t* xyr, int cap) {
    const LevelTap& L = e->lv[level];
    int n = (int)L.cand_xyr.size() / 3;
    if (n > cap) return -1;
    if (!L.cand_counts.empty()) memcpy(counts, &L.cand_counts[0], L.cand_counts.size() * sizeof(int));
    if (n) memcpy(xyr, &L.cand_xyr[0], L.cand_xyr.size() * sizeof(int));
    return n;
}

int orbref_level_retain(const orbref_extractor* e, int level, int32_t* n_to_retain) {
    const LevelTap& L = e->lv[level];
    if (!L.retain.empty()) memcpy(n_to_retain, &L.retain[0], L.retain.size() * sizeof(int));
    return (int)L.retain.size();
}

int orbref_cvround(double v) { return cvRound(v); }
float orbref_fast_atan2(float y, float x) { return fast_atan2(y, x); }
void orbref_resize_linear(const uint8_t* src, int sw, int sh, ptrdiff_t sstride, uint8_t* dst, int dw, int dh, ptrdiff_t dstride) {
    resize_linear_u8(src, sw, sh, sstride, dst, dw, dh, dstride);
}
int orbref_reflect101(int p, int n) { return reflect101(p, n); }

int orbref_fast(const uint8_t* img, int rows, int cols, ptrdiff_t stride, int threshold, int32_t* xyr, int cap) {
    std::vector<KeyPoint> k;
    fast9_16(img, rows, cols, stride, threshold, true, k);
    if ((int)k.size() > cap) return -1;
    for (size_t i = 0; i < k.size(); i++) {
        xyr[3 * i] = (int)k[i].x;
        xyr[3 * i + 1] = (int)k[i].y;
        xyr[3 * i + 2] = (int)k[i].response;
    }
    return (int)k.size();
}

int orbref_fast_measure(const uint8_t* center, ptrdiff_t stride) { return fast_measure(center, stride); }

int orbref_retain_best(const float* responses, int n, int n_points, int32_t* perm_out) {
    std::vector<KeyPoint> v(n);
    for (int i = 0; i < n; i++) {
        KeyPoint k = {0, 0, 0, 0, responses[i], 0, i};
        v[i] = k;
    }
    retain_best(v, n_points);
    for (size_t i = 0; i < v.size(); i++) perm_out[i] = v[i].class_id;
    return (int)v.size();
}

int orbref_nth_prefix(const float* responses, int n, int n_points, int32_t* perm_out) {
    std::vector<KeyPoint> v(n);
    for (int i = 0; i < n; i++) {
        KeyPoint k = {0, 0, 0, 0, responses[i], 0, i};
        v[i] = k;
    }
    retain_best(v, n_points);
    if ((int)v.size() > n_points) v.resize(n_points);
    for (size_t i = 0; i < v.size(); i++) perm_out[i] = v[i].class_id;
    return (int)v.size();
}

// Study switch only (tools/sse2_gaussian_divergence.py): never set by the tests or the bench.
void orbref_set_gauss_profile(int profile) { g_gauss_profile = profile == 1 ? 1 : 0; }

void orbref_gauss7(const uint8_t* padded_in, int w, int h, uint8_t* padded_out) {
    int taps[7];
    gauss_taps(taps);
    gauss7_roi(padded_in, w, h, padded_out, taps);
}

float orbref_ic_angle(const uint8_t* center, ptrdiff_t stride) {
    static const int um[16] = {15, 15, 15, 15, 14, 14, 14, 13, 13, 12, 11, 10, 9, 8, 6, 3};
    return ic_angle(center, stride, um);
}

void orbref_descriptor(const uint8_t* center, ptrdiff_t stride, float angle_deg, uint8_t* desc32) {
    orb_descriptor(center, stride, angle_deg, desc32);
}

float orbref_cosf(float x) { return cosf(x); }
float orbref_sinf(float x) { return sinf(x); }

int orbref_hamming(const void* a, const void* b) {
    const int32_t* pa = (const int32_t*)a;
    const int32_t* pb = (const int32_t*)b;
    int dist = 0;
    for (int i = 0; i < 8; i++, pa++, pb++) {
        unsigned int v = *pa ^ *pb;
        v = v - ((v >> 1) & 0x55555555);
        v = (v & 0x33333333) + ((v >> 2) & 0x33333333);
        dist += (((v + (v >> 4)) & 0xF0F0F0F) * 0x1010101) >> 24;
    }
    return dist;
}

void orbref_knn2(const uint8_t* q, int nq, const uint8_t* t, int nt, int32_t* idx, int32_t* d1, int32_t* d2) {
    for (int i = 0; i < nq; i++) {
        int bestDist = INT_MAX, bestDist2 = INT_MAX, bestIdx = -1;
        const uint8_t* qi = q + (size_t)i * 32;
        for (int j = 0; j < nt; j++) {
            int dist = orbref_hamming(qi, t + (size_t)j * 32);
            if (dist < bestDist) {
                bestDist2 = bestDist;
                bestDist = dist;
                bestIdx = j;
            } else if (dist < bestDist2)
                bestDist2 = dist;
        }
        idx[i] = bestIdx;
        d1[i] = bestDist;
        d2[i] = bestDist2;
    }
}

void orbref_knn2_indexed(const uint8_t* q, const uint8_t* t, const int32_t* cand, int ncand, int32_t* idx, int32_t* d1, int32_t* d2) {
    int bestDist = INT_MAX, bestDist2 = INT_MAX, bestIdx = -1;
    for (int c = 0; c < ncand; c++) {
        int j = cand[c];
        int dist = orbref_hamming(q, t + (size_t)j * 32);
        if (dist < bestDist) {
            bestDist2 = bestDist;
            bestDist = dist;
            bestIdx = j;
        } else if (dist < bestDist2)
            bestDist2 = dist;
    }
    *idx = bestIdx;
    *d1 = bestDist;
    *d2 = bestDist2;
}

int orbref_ratio_accept(int bestDist, int bestDist2, float mfNNratio, int form) {
    switch (form) {
        case 0: return (float)bestDist < mfNNratio * (float)bestDist2;  // src/ORBmatcher.cc:237
        case 1: return bestDist <= bestDist2 * mfNNratio;               // :487
        case 2: return !(bestDist > mfNNratio * bestDist2);             // :119 (levels equal)
    }
    return 0;
}

void orbref_three_maxima(const int32_t* sizes, int L, int32_t* ind) {
    int max1 = 0, max2 = 0, max3 = 0;
    int ind1 = -1, ind2 = -1, ind3 = -1;
    for (int i = 0; i < L; i++) {
        const int s = sizes[i];
        if (s > max1) {
            max3 = max2; max2 = max1; max1 = s;
            ind3 = ind2; ind2 = ind1; ind1 = i;
        } else if (s > max2) {
            max3 = max2; max2 = s;
            ind3 = ind2; ind2 = i;
        } else if (s > max3) {
            max3 = s;
            ind3 = i;
        }
    }
    if (max2 < 0.1f * (float)max1) {
        ind2 = -1;
        ind3 = -1;
    } else if (max3 < 0.1f * (float)max1)
        ind3 = -1;
    ind[0] = ind1; ind[1] = ind2; ind[2] = ind3;
}

int orbref_rot_bin(float a1, float a2) {
    const int HISTO_LENGTH = 30;
    const float factor = 1.0f / HISTO_LENGTH;
    float rot = a1 - a2;
    if (rot < 0.0) rot += 360.0f;
    int bin = round(rot * factor);
    if (bin == HISTO_LENGTH) bin = 0;
    return bin;
}

}  // extern "C"
