/* hs_oracle.cpp — CPU ORACLE (test infrastructure; see hs_oracle.h for status: parity unpinned).
 *
 * Structure mirrors the reference so that it doubles as the "port" CPU baseline:
 *   ORBExtractor  : /root/reference/src/features/ORBExtractor.cpp
 *   ORBFinder     : /root/reference/src/features/low_level/ORBFinder.cpp
 *   ORBDistance   : /root/reference/src/features/low_level/DescriptorDistance.cpp
 *   Stereomatcher : /root/reference/src/features/Stereomatcher.cpp
 *   harness       : /root/reference/src/main/ImageProcessing.cpp:69-116
 * OpenCV 3.4 primitives are restated from OpenCV's published algorithms.
 * Compile with -ffp-contract=off (x86-64 baseline has no FMA; results depend on it).
 */
#include "hs_oracle.h"
#include "../include/hyslam_orb_pattern.h"

#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <list>
#include <thread>
#include <utility>
#include <vector>

namespace {

/* ------------------------------------------------------------------ A.0 cvRound */
// x86-64: cvtss2si / cvtsd2si under the default MXCSR = round half to even.
inline int cvRoundF(float v) { return (int)std::nearbyintf(v); }
inline int cvRoundD(double v) { return (int)std::nearbyint(v); }
inline int cvFloorF(float v) { int i = (int)v; return i - (i > v); }
inline short saturate_short(float v) { int iv = cvRoundF(v); return (short)std::min(std::max(iv, -32768), 32767); }

const int PATCH_SIZE = 31;       // ORBExtractor.cpp:73
const int EDGE_THRESHOLD = 19;   // ORBExtractor.cpp:74
const int HALF_PATCH_SIZE = 15;  // ORBFinder.cpp:14

struct KeyPoint { float x, y, size, angle, response; int octave; int src; };

/* ------------------------------------------------------------------ A.4 cv::fastAtan2 */
float fastAtan2(float y, float x)
{
    // OpenCV 3.4 modules/core/src/mathfuncs_core: atan_f32, degrees; products folded in fp32
    static const float p1 = 0.9997878412794807f * (float)(180 / 3.1415926535897932384626433832795);
    static const float p3 = -0.3258083974640975f * (float)(180 / 3.1415926535897932384626433832795);
    static const float p5 = 0.1555786518463281f * (float)(180 / 3.1415926535897932384626433832795);
    static const float p7 = -0.04432655554792128f * (float)(180 / 3.1415926535897932384626433832795);
    float ax = std::fabs(x), ay = std::fabs(y);
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

/* ------------------------------------------------------------------ A.2 cv::resize INTER_LINEAR 8UC1 */
void resizeLinearU8(const uint8_t* src, int sw, int sh, int sstride, uint8_t* dst, int dw, int dh, int dstride)
{
    const int COEF_BITS = 11, COEF_SCALE = 1 << COEF_BITS;
    double inv_scale_x = (double)dw / sw, inv_scale_y = (double)dh / sh;
    double scale_x = 1. / inv_scale_x, scale_y = 1. / inv_scale_y;

    std::vector<int> xofs(dw), yofs(dh);
    std::vector<short> ialpha(dw * 2), ibeta(dh * 2);
    int xmax = dw;
    for (int dx = 0; dx < dw; dx++) {
        float fx = (float)((dx + 0.5) * scale_x - 0.5);
        int sx = cvFloorF(fx);
        fx -= sx;
        if (sx < 0) { fx = 0; sx = 0; }
        if (sx + 1 >= sw) {
            xmax = std::min(xmax, dx);
            if (sx >= sw - 1) { fx = 0; sx = sw - 1; }
        }
        xofs[dx] = sx;
        ialpha[dx * 2] = saturate_short((1.f - fx) * COEF_SCALE);
        ialpha[dx * 2 + 1] = saturate_short(fx * COEF_SCALE);
    }
    for (int dy = 0; dy < dh; dy++) {
        float fy = (float)((dy + 0.5) * scale_y - 0.5);
        int sy = cvFloorF(fy);
        fy -= sy;
        yofs[dy] = sy;
        ibeta[dy * 2] = saturate_short((1.f - fy) * COEF_SCALE);
        ibeta[dy * 2 + 1] = saturate_short(fy * COEF_SCALE);
    }
    std::vector<int> row0(dw), row1(dw);
    auto hresize = [&](int sy, std::vector<int>& D) {
        const uint8_t* S = src + (size_t)sy * sstride;
        int dx = 0;
        for (; dx < xmax; dx++) {
            int sx = xofs[dx];
            D[dx] = S[sx] * ialpha[dx * 2] + S[sx + 1] * ialpha[dx * 2 + 1];
        }
        for (; dx < dw; dx++) D[dx] = S[xofs[dx]] * COEF_SCALE;
    };
    auto clip = [](int x, int a, int b) { return x >= a ? (x < b ? x : b - 1) : a; };
    for (int dy = 0; dy < dh; dy++) {
        int sy0 = clip(yofs[dy], 0, sh), sy1 = clip(yofs[dy] + 1, 0, sh);
        hresize(sy0, row0);
        hresize(sy1, row1);
        int b0 = ibeta[dy * 2], b1 = ibeta[dy * 2 + 1];
        uint8_t* D = dst + (size_t)dy * dstride;
        for (int x = 0; x < dw; x++)   // VResizeLinear<uchar,int,short,...> 8-bit specialisation
            D[x] = (uint8_t)((((b0 * (row0[x] >> 4)) >> 16) + ((b1 * (row1[x] >> 4)) >> 16) + 2) >> 2);
    }
}

/* ------------------------------------------------------------------ ImageProcessing::PreProcessImg (src/main/ImageProcessing.cpp:118-138)
 * cv::resize(img, img, cv::Size(), fscale, fscale) on the frame AS IT COMES (1, 3 or 4 interleaved channels), then cvtColor to grey.  OpenCV 3.4
 * (modules/imgproc/src/resize.cpp, color_rgb.cpp), restated from its published algorithms — same status as every other primitive here:
 *   dsize    = (saturate_cast<int>(w * inv_scale_x), ...) = cvRound of the DOUBLE product, inv_scale = (double)fscale; the scale the tables use is
 *              1. / inv_scale (it is NOT re-derived from dsize when dsize was given empty)
 *   dsize == ssize                      -> a copy
 *   INTER_LINEAR with scale_x == scale_y == 2 exactly (the reference's "Imaging" camera: scale 0.5, config/sample_primary_config_file.yaml:66)
 *                                       -> silently INTER_AREA's fast path: D = (S00 + S01 + S10 + S11 + 2) >> 2 per channel for the w / 2 full blocks,
 *                                          saturate_cast<uchar>((float)sum / count) over the samples that exist for a trailing partial block
 *   anything else                       -> the 11-bit fixed-point bilinear of A.2 per channel (xofs[dx * cn + k] = sx * cn + k)
 *   RGB2GRAY / BGR2GRAY / RGBA / BGRA   -> (R * 4899 + G * 9617 + B * 1868 + (1 << 13)) >> 14   (R2Y, G2Y, B2Y at yuv_shift = 14); alpha ignored */
void preprocessSize(int w, int h, float fscale, int* ow, int* oh)
{
    const double inv = (double)fscale;
    *ow = cvRoundD((double)w * inv); *oh = cvRoundD((double)h * inv);
}
static void resizeColor(const uint8_t* src, int sw, int sh, int sstride, int cn, float fscale, std::vector<uint8_t>& out, int dw, int dh)
{
    out.assign((size_t)dw * dh * cn, 0);
    const double inv_scale = (double)fscale, scale = 1. / inv_scale;
    if (dw == sw && dh == sh) {
        for (int y = 0; y < sh; y++) memcpy(&out[(size_t)y * dw * cn], src + (size_t)y * sstride, (size_t)sw * cn);
        return;
    }
    const int iscale = cvRoundD(scale);                                     // saturate_cast<int>(scale_x)
    const bool is_area_fast = std::fabs(scale - iscale) < DBL_EPSILON;
    if (is_area_fast && iscale == 2) {
        const int wfull = sw / 2;                                           // blocks with all four samples
        for (int dy = 0; dy < dh; dy++) {
            const int sy0 = dy * 2;
            uint8_t* D = &out[(size_t)dy * dw * cn];
            if (sy0 >= sh) continue;                                        // (row of zeros)
            const uint8_t* S = src + (size_t)sy0 * sstride;
            const uint8_t* nS = sy0 + 1 < sh ? S + sstride : nullptr;
            for (int dx = 0; dx < dw; dx++)
                for (int k = 0; k < cn; k++) {
                    const int sx0 = dx * 2;
                    if (dx < wfull && nS) { D[dx * cn + k] = (uint8_t)((S[sx0 * cn + k] + S[(sx0 + 1) * cn + k] + nS[sx0 * cn + k] + nS[(sx0 + 1) * cn + k] + 2) >> 2); continue; }
                    if (sx0 >= sw) { D[dx * cn + k] = 0; continue; }
                    int sum = 0, count = 0;
                    for (int yy = 0; yy < 2 && sy0 + yy < sh; yy++)
                        for (int xx = 0; xx < 2 && sx0 + xx < sw; xx++) { sum += src[(size_t)(sy0 + yy) * sstride + (sx0 + xx) * cn + k]; count++; }
                    D[dx * cn + k] = (uint8_t)std::min(std::max(cvRoundF((float)sum / count), 0), 255);
                }
        }
        return;
    }
    const int COEF_BITS = 11, COEF_SCALE = 1 << COEF_BITS;
    std::vector<int> xofs(dw), yofs(dh);
    std::vector<short> ialpha(dw * 2), ibeta(dh * 2);
    int xmax = dw;
    for (int dx = 0; dx < dw; dx++) {
        float fx = (float)((dx + 0.5) * scale - 0.5);
        int sx = cvFloorF(fx);
        fx -= sx;
        if (sx < 0) { fx = 0; sx = 0; }
        if (sx + 1 >= sw) { xmax = std::min(xmax, dx); if (sx >= sw - 1) { fx = 0; sx = sw - 1; } }
        xofs[dx] = sx;
        ialpha[dx * 2] = saturate_short((1.f - fx) * COEF_SCALE);
        ialpha[dx * 2 + 1] = saturate_short(fx * COEF_SCALE);
    }
    for (int dy = 0; dy < dh; dy++) {
        float fy = (float)((dy + 0.5) * scale - 0.5);
        int sy = cvFloorF(fy);
        fy -= sy;
        yofs[dy] = sy;
        ibeta[dy * 2] = saturate_short((1.f - fy) * COEF_SCALE);
        ibeta[dy * 2 + 1] = saturate_short(fy * COEF_SCALE);
    }
    auto clip = [](int x, int a, int b) { return x >= a ? (x < b ? x : b - 1) : a; };
    for (int dy = 0; dy < dh; dy++) {
        const uint8_t* S0 = src + (size_t)clip(yofs[dy], 0, sh) * sstride;
        const uint8_t* S1 = src + (size_t)clip(yofs[dy] + 1, 0, sh) * sstride;
        const int b0 = ibeta[dy * 2], b1 = ibeta[dy * 2 + 1];
        uint8_t* D = &out[(size_t)dy * dw * cn];
        for (int dx = 0; dx < dw; dx++)
            for (int k = 0; k < cn; k++) {
                const int sx = xofs[dx] * cn + k;
                int h0, h1;
                if (dx < xmax) { h0 = S0[sx] * ialpha[dx * 2] + S0[sx + cn] * ialpha[dx * 2 + 1]; h1 = S1[sx] * ialpha[dx * 2] + S1[sx + cn] * ialpha[dx * 2 + 1]; }
                else { h0 = S0[sx] * COEF_SCALE; h1 = S1[sx] * COEF_SCALE; }
                D[dx * cn + k] = (uint8_t)((((b0 * (h0 >> 4)) >> 16) + ((b1 * (h1 >> 4)) >> 16) + 2) >> 2);
            }
    }
}
int preprocessImg(const uint8_t* src, int w, int h, int sstride, int cn, int rgb, float fscale, uint8_t* dst, int dstride)
{
    if (!(cn == 1 || cn == 3 || cn == 4) || w < 1 || h < 1) return -1;
    int dw, dh;
    preprocessSize(w, h, fscale, &dw, &dh);
    if (dw < 1 || dh < 1) return -1;                                       // (OpenCV asserts !dsize.empty())
    std::vector<uint8_t> col;
    resizeColor(src, w, h, sstride, cn, fscale, col, dw, dh);
    for (int y = 0; y < dh; y++) {
        const uint8_t* S = &col[(size_t)y * dw * cn];
        uint8_t* D = dst + (size_t)y * dstride;
        if (cn == 1) { memcpy(D, S, (size_t)dw); continue; }
        for (int x = 0; x < dw; x++) {
            const int c0 = S[x * cn], c1 = S[x * cn + 1], c2 = S[x * cn + 2];
            const int r = rgb ? c0 : c2, b = rgb ? c2 : c0;
            D[x] = (uint8_t)((r * 4899 + c1 * 9617 + b * 1868 + (1 << 13)) >> 14);
        }
    }
    return 0;
}

/* ------------------------------------------------------------------ A.1 cv::FAST 9_16 */
static const int kRing[16][2] = { {0,3},{1,3},{2,2},{3,1},{3,0},{3,-1},{2,-2},{1,-3},
                                  {0,-3},{-1,-3},{-2,-2},{-3,-1},{-3,0},{-3,1},{-2,2},{-1,3} };

int cornerScore16(const uint8_t* ptr, const int* pixel, int threshold)
{
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

// cv::FAST(img, keypoints, threshold, nonmax, TYPE_9_16) on a (sub-)image view.
void FAST(const uint8_t* img, int cols, int rows, int step, int threshold, bool nonmax, std::vector<KeyPoint>& keypoints)
{
    keypoints.clear();
    const int K = 8, N = 16 + K + 1;
    int pixel[25];
    for (int k = 0; k < 16; k++) pixel[k] = kRing[k][0] + kRing[k][1] * step;
    for (int k = 16; k < 25; k++) pixel[k] = pixel[k - 16];
    threshold = std::min(std::max(threshold, 0), 255);
    if (rows < 7 || cols < 7) return;
    uint8_t threshold_tab[512];
    for (int i = -255; i <= 255; i++) threshold_tab[i + 255] = (uint8_t)(i < -threshold ? 1 : i > threshold ? 2 : 0);

    // three rolling score rows + corner positions, as in OpenCV
    std::vector<uint8_t> bufmem((size_t)cols * 3, 0);
    uint8_t* buf[3] = { bufmem.data(), bufmem.data() + cols, bufmem.data() + 2 * cols };
    std::vector<int> cpmem((size_t)(cols + 1) * 3, 0);
    int* cpbuf[3] = { cpmem.data() + 1, cpmem.data() + 1 + (cols + 1), cpmem.data() + 1 + 2 * (cols + 1) };

    for (int i = 3; i < rows - 2; i++) {
        const uint8_t* ptr = img + (size_t)i * step + 3;
        uint8_t* curr = buf[(i - 3) % 3];
        int* cornerpos = cpbuf[(i - 3) % 3];
        std::memset(curr, 0, cols);
        int ncorners = 0;
        if (i < rows - 3) {
            for (int j = 3; j < cols - 3; j++, ptr++) {
                int v = ptr[0];
                const uint8_t* tab = &threshold_tab[0] - v + 255;
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
                bool is_corner = false;
                if (d & 1) {   // darker arc
                    int vt = v - threshold, count = 0;
                    for (int k = 0; k < N; k++) {
                        if (ptr[pixel[k]] < vt) { if (++count > K) { is_corner = true; break; } }
                        else count = 0;
                    }
                }
                if (!is_corner && (d & 2)) { // brighter arc
                    int vt = v + threshold, count = 0;
                    for (int k = 0; k < N; k++) {
                        if (ptr[pixel[k]] > vt) { if (++count > K) { is_corner = true; break; } }
                        else count = 0;
                    }
                }
                if (is_corner) {
                    cornerpos[ncorners++] = j;
                    if (nonmax) curr[j] = (uint8_t)cornerScore16(ptr, pixel, threshold);
                }
            }
        }
        cornerpos[-1] = ncorners;
        if (i == 3) continue;
        const uint8_t* prev = buf[(i - 4 + 3) % 3];
        const uint8_t* pprev = buf[(i - 5 + 3) % 3];
        cornerpos = cpbuf[(i - 4 + 3) % 3];
        ncorners = cornerpos[-1];
        for (int k = 0; k < ncorners; k++) {
            int j = cornerpos[k];
            int score = prev[j];
            if (!nonmax ||
                (score > prev[j + 1] && score > prev[j - 1] &&
                 score > pprev[j - 1] && score > pprev[j] && score > pprev[j + 1] &&
                 score > curr[j - 1] && score > curr[j] && score > curr[j + 1])) {
                KeyPoint kp; kp.x = (float)j; kp.y = (float)(i - 1); kp.size = 7.f; kp.angle = -1.f;
                kp.response = (float)score; kp.octave = 0; kp.src = -1;
                keypoints.push_back(kp);
            }
        }
    }
}

/* ------------------------------------------------------------------ A.3 cv::GaussianBlur 7x7 sigma 2, 8U fixed point */
static const uint16_t kDefaultTaps[7] = { 18, 34, 49, 55, 49, 34, 18 };
inline int reflect101(int p, int len) {
    if (len == 1) return 0;
    while (p < 0 || p >= len) { if (p < 0) p = -p; else p = 2 * (len - 1) - p; }
    return p;
}
void gaussianBlur7(const uint8_t* src, int w, int h, int sstride, uint8_t* dst, int dstride, const uint16_t* taps)
{
    std::vector<uint16_t> hbuf((size_t)w * h);
    uint32_t tapsum = 0; for (int k = 0; k < 7; k++) tapsum += taps[k];
    const bool nosat = tapsum * 255u <= 0xFFFFu;      // ufixedpoint16 adds cannot saturate
    for (int y = 0; y < h; y++) {
        const uint8_t* S = src + (size_t)y * sstride;
        uint16_t* H = hbuf.data() + (size_t)y * w;
        for (int x = 0; x < w; x++) {
            if (nosat && x >= 3 && x < w - 3) {
                const uint8_t* s = S + x - 3;
                H[x] = (uint16_t)(taps[0] * s[0] + taps[1] * s[1] + taps[2] * s[2] + taps[3] * s[3] + taps[4] * s[4] + taps[5] * s[5] + taps[6] * s[6]);
                continue;
            }
            uint32_t acc = 0;   // ufixedpoint16 saturating adds
            for (int k = 0; k < 7; k++) {
                uint32_t t = (uint32_t)taps[k] * S[reflect101(x + k - 3, w)];
                if (t > 0xFFFF) t = 0xFFFF;
                acc += t;
                if (acc > 0xFFFF) acc = 0xFFFF;
            }
            H[x] = (uint16_t)acc;
        }
    }
    for (int y = 0; y < h; y++) {
        uint8_t* D = dst + (size_t)y * dstride;
        const uint16_t* R[7];
        for (int k = 0; k < 7; k++) R[k] = hbuf.data() + (size_t)reflect101(y + k - 3, h) * w;
        for (int x = 0; x < w; x++) {
            uint64_t acc = 0;   // ufixedpoint32 saturating adds
            for (int k = 0; k < 7; k++) {
                acc += (uint64_t)taps[k] * R[k][x];
                if (acc > 0xFFFFFFFFull) acc = 0xFFFFFFFFull;
            }
            uint64_t r = (acc + 0x8000) >> 16;
            D[x] = (uint8_t)(r > 255 ? 255 : r);
        }
    }
}

/* ------------------------------------------------------------------ ORBFinder (ORBFinder.cpp) */
struct ORBFinder {
    int threshold = 20;                 // ORBFinder.h:92
    bool non_max_suppression = true;
    std::vector<int> umax;
    const int* pattern;                 // 512 points (x,y)

    explicit ORBFinder(int effective_threshold) {
        threshold = effective_threshold;
        static const int pat[HS_ORB_PATTERN_INTS] = HS_ORB_PATTERN_INIT;
        pattern = pat;
        orientationSetup();
    }
    // ORBFinder.cpp:58-60: `threshold = round(threshold)` self-assigns; the argument is dropped.
    void setThreshold(double /*threshold_*/) { threshold = (int)std::round((double)threshold); }

    void detect(const uint8_t* img, int cols, int rows, int step, std::vector<KeyPoint>& kps) {
        FAST(img, cols, rows, step, threshold, non_max_suppression, kps);   // ORBFinder.cpp:66-68
    }
    // ORBFinder.cpp:131-149
    void orientationSetup() {
        umax.assign(HALF_PATCH_SIZE + 1, 0);
        int v, v0, vmax = (int)std::floor(HALF_PATCH_SIZE * std::sqrt(2.f) / 2 + 1);
        int vmin = (int)std::ceil(HALF_PATCH_SIZE * std::sqrt(2.f) / 2);
        const double hp2 = HALF_PATCH_SIZE * HALF_PATCH_SIZE;
        for (v = 0; v <= vmax; ++v) umax[v] = cvRoundD(std::sqrt(hp2 - v * v));
        for (v = HALF_PATCH_SIZE, v0 = 0; v >= vmin; --v) {
            while (umax[v0] == umax[v0 + 1]) ++v0;
            umax[v] = v0;
            ++v0;
        }
    }
    // ORBFinder.cpp:16-43
    float intensityCentroidAngle(const uint8_t* image, int step, float ptx, float pty) const {
        int m_01 = 0, m_10 = 0;
        const uint8_t* center = image + (ptrdiff_t)cvRoundF(pty) * step + cvRoundF(ptx);
        for (int u = -HALF_PATCH_SIZE; u <= HALF_PATCH_SIZE; ++u) m_10 += u * center[u];
        for (int v = 1; v <= HALF_PATCH_SIZE; ++v) {
            int v_sum = 0;
            int d = umax[v];
            for (int u = -d; u <= d; ++u) {
                int val_plus = center[u + v * step], val_minus = center[u - v * step];
                v_sum += (val_plus - val_minus);
                m_10 += u * (val_plus + val_minus);
            }
            m_01 += v * v_sum;
        }
        return fastAtan2((float)m_01, (float)m_10);
    }
    // ORBFinder.cpp:89-129
    void computeOrbDescriptor(const uint8_t* img, int step, float ptx, float pty, float kpangle, uint8_t* desc) const {
        const float factorPI = (float)(3.1415926535897932384626433832795 / 180.f);
        float angle = (float)kpangle * factorPI;
        // `cos(angle)` in namespace HYSLAM with only <cmath>: binds ::cos(double)
        float a = (float)std::cos((double)angle), b = (float)std::sin((double)angle);
        const uint8_t* center = img + (ptrdiff_t)cvRoundF(pty) * step + cvRoundF(ptx);
        const int* pat = pattern;
        auto value = [&](int idx) -> int {
            float px = (float)pat[idx * 2], py = (float)pat[idx * 2 + 1];
            int dy = cvRoundF(px * b + py * a);
            int dx = cvRoundF(px * a - py * b);
            return center[dy * step + dx];
        };
        for (int i = 0; i < 32; ++i, pat += 32) {
            int val = 0;
            for (int t = 0; t < 8; t++) {
                int t0 = value(2 * t), t1 = value(2 * t + 1);
                val |= (t0 < t1) << t;
            }
            desc[i] = (uint8_t)val;
        }
    }
    // ORBFinder.cpp:70-87
    void compute(const uint8_t* image, int step, std::vector<KeyPoint>& kps, uint8_t* descriptors) const {
        for (auto& kp : kps) kp.angle = intensityCentroidAngle(image, step, kp.x, kp.y);
        for (size_t i = 0; i < kps.size(); i++)
            computeOrbDescriptor(image, step, kps[i].x, kps[i].y, kps[i].angle, descriptors + 32 * i);
    }
};

/* ------------------------------------------------------------------ ORBExtractor (ORBExtractor.cpp) */
struct P2i { int x, y; };

struct ExtractorNode {
    std::vector<KeyPoint> vKeys;
    P2i UL, UR, BL, BR;
    std::list<ExtractorNode>::iterator lit;
    bool bNoMore = false;
    long seq = 0;   // D1: stands in for the node's heap address in the (size, pointer) sort
    void DivideNode(ExtractorNode& n1, ExtractorNode& n2, ExtractorNode& n3, ExtractorNode& n4);
};

// ORBExtractor.cpp:121-177
void ExtractorNode::DivideNode(ExtractorNode& n1, ExtractorNode& n2, ExtractorNode& n3, ExtractorNode& n4)
{
    const int halfX = (int)std::ceil(static_cast<float>(UR.x - UL.x) / 2);
    const int halfY = (int)std::ceil(static_cast<float>(BR.y - UL.y) / 2);
    n1.UL = UL;                          n1.UR = { UL.x + halfX, UL.y };
    n1.BL = { UL.x, UL.y + halfY };      n1.BR = { UL.x + halfX, UL.y + halfY };
    n2.UL = n1.UR;                       n2.UR = UR;
    n2.BL = n1.BR;                       n2.BR = { UR.x, UL.y + halfY };
    n3.UL = n1.BL;                       n3.UR = n1.BR;
    n3.BL = BL;                          n3.BR = { n1.BR.x, BL.y };
    n4.UL = n3.UR;                       n4.UR = n2.BR;
    n4.BL = n3.BR;                       n4.BR = BR;
    for (size_t i = 0; i < vKeys.size(); i++) {
        const KeyPoint& kp = vKeys[i];
        if (kp.x < n1.UR.x) {
            if (kp.y < n1.BR.y) n1.vKeys.push_back(kp); else n3.vKeys.push_back(kp);
        } else if (kp.y < n1.BR.y) n2.vKeys.push_back(kp);
        else n4.vKeys.push_back(kp);
    }
    if (n1.vKeys.size() == 1) n1.bNoMore = true;
    if (n2.vKeys.size() == 1) n2.bNoMore = true;
    if (n3.vKeys.size() == 1) n3.bNoMore = true;
    if (n4.vKeys.size() == 1) n4.bNoMore = true;
}

// ORBExtractor.cpp:179-403
std::vector<KeyPoint> DistributeOctTree(const std::vector<KeyPoint>& vToDistributeKeys, int minX, int maxX, int minY, int maxY, int N)
{
    typedef std::list<ExtractorNode>::iterator Lit;
    long seq_counter = 0;
    const int nIni = (int)std::round(static_cast<float>(maxX - minX) / (maxY - minY));
    std::vector<KeyPoint> vResultKeys;
    if (nIni < 1) return vResultKeys;   // reference: UB (empty vpIniNodes indexed); callers reject W/H < 0.5
    const float hX = static_cast<float>(maxX - minX) / nIni;

    std::list<ExtractorNode> lNodes;
    std::vector<ExtractorNode*> vpIniNodes(nIni);
    for (int i = 0; i < nIni; i++) {
        ExtractorNode ni;
        ni.UL = { (int)(hX * static_cast<float>(i)), 0 };
        ni.UR = { (int)(hX * static_cast<float>(i + 1)), 0 };
        ni.BL = { ni.UL.x, maxY - minY };
        ni.BR = { ni.UR.x, maxY - minY };
        ni.seq = seq_counter++;
        lNodes.push_back(ni);
        vpIniNodes[i] = &lNodes.back();
    }
    for (size_t i = 0; i < vToDistributeKeys.size(); i++) {
        const KeyPoint& kp = vToDistributeKeys[i];
        vpIniNodes[(size_t)(kp.x / hX)]->vKeys.push_back(kp);
    }
    Lit lit = lNodes.begin();
    while (lit != lNodes.end()) {
        if (lit->vKeys.size() == 1) { lit->bNoMore = true; lit++; }
        else if (lit->vKeys.empty()) lit = lNodes.erase(lit);
        else lit++;
    }

    bool bFinish = false;
    typedef std::pair<int, ExtractorNode*> SizePtr;
    std::vector<SizePtr> vSizeAndPointerToNode;
    auto by_size_then_seq = [](const SizePtr& a, const SizePtr& b) {   // D1
        if (a.first != b.first) return a.first < b.first;
        return a.second->seq < b.second->seq;
    };
    auto add_children = [&](ExtractorNode* n[4], int* nToExpand) {
        for (int c = 0; c < 4; c++) {
            if (n[c]->vKeys.size() > 0) {
                n[c]->seq = seq_counter++;
                lNodes.push_front(*n[c]);
                if (n[c]->vKeys.size() > 1) {
                    if (nToExpand) (*nToExpand)++;
                    vSizeAndPointerToNode.push_back(std::make_pair((int)n[c]->vKeys.size(), &lNodes.front()));
                    lNodes.front().lit = lNodes.begin();
                }
            }
        }
    };

    while (!bFinish) {
        int prevSize = (int)lNodes.size();
        lit = lNodes.begin();
        int nToExpand = 0;
        vSizeAndPointerToNode.clear();
        while (lit != lNodes.end()) {
            if (lit->bNoMore) { lit++; continue; }
            ExtractorNode n1, n2, n3, n4;
            lit->DivideNode(n1, n2, n3, n4);
            ExtractorNode* ch[4] = { &n1, &n2, &n3, &n4 };
            add_children(ch, &nToExpand);
            lit = lNodes.erase(lit);
        }
        if ((int)lNodes.size() >= N || (int)lNodes.size() == prevSize) {
            bFinish = true;
        } else if (((int)lNodes.size() + nToExpand * 3) > N) {
            while (!bFinish) {
                prevSize = (int)lNodes.size();
                std::vector<SizePtr> vPrevSizeAndPointerToNode = vSizeAndPointerToNode;
                vSizeAndPointerToNode.clear();
                std::sort(vPrevSizeAndPointerToNode.begin(), vPrevSizeAndPointerToNode.end(), by_size_then_seq);
                for (int j = (int)vPrevSizeAndPointerToNode.size() - 1; j >= 0; j--) {
                    ExtractorNode n1, n2, n3, n4;
                    vPrevSizeAndPointerToNode[j].second->DivideNode(n1, n2, n3, n4);
                    ExtractorNode* ch[4] = { &n1, &n2, &n3, &n4 };
                    add_children(ch, nullptr);
                    lNodes.erase(vPrevSizeAndPointerToNode[j].second->lit);
                    if ((int)lNodes.size() >= N) break;
                }
                if ((int)lNodes.size() >= N || (int)lNodes.size() == prevSize) bFinish = true;
            }
        }
    }
    vResultKeys.reserve(lNodes.size());
    for (Lit l = lNodes.begin(); l != lNodes.end(); l++) {
        std::vector<KeyPoint>& vNodeKeys = l->vKeys;
        KeyPoint* pKP = &vNodeKeys[0];
        float maxResponse = pKP->response;
        for (size_t k = 1; k < vNodeKeys.size(); k++) {
            if (vNodeKeys[k].response > maxResponse) { pKP = &vNodeKeys[k]; maxResponse = vNodeKeys[k].response; }
        }
        vResultKeys.push_back(*pKP);
    }
    return vResultKeys;
}

struct View { const uint8_t* data; int cols, rows, step; };

struct ORBExtractor {
    ORBFinder finder;
    int nfeatures; double scaleFactor; int nlevels; int iniThFAST, minThFAST; int N_CELLS;
    uint16_t taps[7];
    std::vector<int> mnFeaturesPerLevel;
    std::vector<float> mvScaleFactor, mvInvScaleFactor, mvLevelSigma2, mvInvLevelSigma2;
    std::vector<std::vector<uint8_t>> pyrmem;
    std::vector<View> mvImagePyramid;

    // ORBExtractor.cpp:76-119
    explicit ORBExtractor(const hso_orb_params& s) : finder(s.fast_threshold) {
        nfeatures = s.nfeatures; scaleFactor = s.scale_factor; nlevels = s.nlevels;
        N_CELLS = s.cell_px; iniThFAST = s.ini_th_fast; minThFAST = s.min_th_fast;
        bool zero = true; for (int k = 0; k < 7; k++) zero = zero && s.blur_taps[k] == 0;
        for (int k = 0; k < 7; k++) taps[k] = zero ? kDefaultTaps[k] : s.blur_taps[k];
        mvScaleFactor.resize(nlevels); mvLevelSigma2.resize(nlevels);
        mvScaleFactor[0] = 1.0f; mvLevelSigma2[0] = 1.0f;
        for (int i = 1; i < nlevels; i++) {
            mvScaleFactor[i] = (float)(mvScaleFactor[i - 1] * scaleFactor);
            mvLevelSigma2[i] = mvScaleFactor[i] * mvScaleFactor[i];
        }
        mvInvScaleFactor.resize(nlevels); mvInvLevelSigma2.resize(nlevels);
        for (int i = 0; i < nlevels; i++) {
            mvInvScaleFactor[i] = 1.0f / mvScaleFactor[i];
            mvInvLevelSigma2[i] = 1.0f / mvLevelSigma2[i];
        }
        mvImagePyramid.resize(nlevels); pyrmem.resize(nlevels);
        mnFeaturesPerLevel.resize(nlevels);
        float factor = (float)(1.0f / scaleFactor);
        float nDesiredFeaturesPerScale = nfeatures * (1 - factor) / (1 - (float)std::pow((double)factor, (double)nlevels));
        int sumFeatures = 0;
        for (int level = 0; level < nlevels - 1; level++) {
            mnFeaturesPerLevel[level] = cvRoundF(nDesiredFeaturesPerScale);
            sumFeatures += mnFeaturesPerLevel[level];
            nDesiredFeaturesPerScale *= factor;
        }
        mnFeaturesPerLevel[nlevels - 1] = std::max(nfeatures - sumFeatures, 0);
    }
    void levelSize(int cols, int rows, int level, int& lw, int& lh) const {   // ORBExtractor.cpp:568-569
        float scale = mvInvScaleFactor[level];
        lw = cvRoundF((float)cols * scale); lh = cvRoundF((float)rows * scale);
    }
    // ORBExtractor.cpp:564-589.  The EDGE_THRESHOLD border made by copyMakeBorder is never read
    // downstream, so levels are stored border-less.
    void ComputePyramid(const View& image) {
        for (int level = 0; level < nlevels; ++level) {
            int lw, lh; levelSize(image.cols, image.rows, level, lw, lh);
            pyrmem[level].assign((size_t)lw * lh, 0);
            if (level != 0) {
                const View& prev = mvImagePyramid[level - 1];
                resizeLinearU8(prev.data, prev.cols, prev.rows, prev.step, pyrmem[level].data(), lw, lh, lw);
            } else {
                for (int y = 0; y < lh; y++) std::memcpy(pyrmem[0].data() + (size_t)y * lw, image.data + (size_t)y * image.step, lw);
            }
            mvImagePyramid[level] = View{ pyrmem[level].data(), lw, lh, lw };
        }
    }
    void cellGrid(int lw, int lh, int& nCols, int& nRows, int& wCell, int& hCell) const {   // :413-428
        const float W = (float)N_CELLS;
        const int minBorderX = EDGE_THRESHOLD - 3, minBorderY = minBorderX;
        const int maxBorderX = lw - EDGE_THRESHOLD + 3, maxBorderY = lh - EDGE_THRESHOLD + 3;
        const float width = (float)(maxBorderX - minBorderX), height = (float)(maxBorderY - minBorderY);
        nCols = (int)(width / W); nRows = (int)(height / W);
        wCell = nCols > 0 ? (int)std::ceil(width / nCols) : 0;
        hCell = nRows > 0 ? (int)std::ceil(height / nRows) : 0;
    }
    // the cell loop of ComputeKeyPointsOctTree, ORBExtractor.cpp:413-470
    void levelCandidates(const View& img, std::vector<KeyPoint>& vToDistributeKeys) {
        const int minBorderX = EDGE_THRESHOLD - 3, minBorderY = minBorderX;
        const int maxBorderX = img.cols - EDGE_THRESHOLD + 3, maxBorderY = img.rows - EDGE_THRESHOLD + 3;
        vToDistributeKeys.clear();
        int nCols, nRows, wCell, hCell; cellGrid(img.cols, img.rows, nCols, nRows, wCell, hCell);
        for (int i = 0; i < nRows; i++) {
            const float iniY = (float)(minBorderY + i * hCell);
            float maxY = iniY + hCell + 6;
            if (iniY >= maxBorderY - 3) continue;
            if (maxY > maxBorderY) maxY = (float)maxBorderY;
            for (int j = 0; j < nCols; j++) {
                const float iniX = (float)(minBorderX + j * wCell);
                float maxX = iniX + wCell + 6;
                if (iniX >= maxBorderX - 6) continue;
                if (maxX > maxBorderX) maxX = (float)maxBorderX;
                std::vector<KeyPoint> vKeysCell;
                const uint8_t* sub = img.data + (size_t)(int)iniY * img.step + (int)iniX;
                int subrows = (int)maxY - (int)iniY, subcols = (int)maxX - (int)iniX;
                finder.setThreshold(iniThFAST);
                finder.detect(sub, subcols, subrows, img.step, vKeysCell);
                if (vKeysCell.empty()) {
                    finder.setThreshold(minThFAST);
                    finder.detect(sub, subcols, subrows, img.step, vKeysCell);
                }
                for (auto& kp : vKeysCell) {
                    kp.x += j * wCell; kp.y += i * hCell;
                    kp.src = (int)vToDistributeKeys.size();
                    vToDistributeKeys.push_back(kp);
                }
            }
        }
    }
    // ORBExtractor.cpp:405-494
    void ComputeKeyPointsOctTree(std::vector<std::vector<KeyPoint>>& allKeypoints, hso_extract_debug* dbg) {
        allKeypoints.resize(nlevels);
        for (int level = 0; level < nlevels; ++level) {
            const View& img = mvImagePyramid[level];
            const int minBorderX = EDGE_THRESHOLD - 3, minBorderY = minBorderX;
            const int maxBorderX = img.cols - EDGE_THRESHOLD + 3, maxBorderY = img.rows - EDGE_THRESHOLD + 3;
            std::vector<KeyPoint> vToDistributeKeys;
            levelCandidates(img, vToDistributeKeys);
            if (dbg) {
                if (dbg->n_candidates) dbg->n_candidates[level] = (int)vToDistributeKeys.size();
                if (dbg->candidates && dbg->candidates[level]) {
                    int n = std::min((int)vToDistributeKeys.size(), dbg->cand_cap);
                    for (int i = 0; i < n; i++) {
                        dbg->candidates[level][3 * i] = vToDistributeKeys[i].x;
                        dbg->candidates[level][3 * i + 1] = vToDistributeKeys[i].y;
                        dbg->candidates[level][3 * i + 2] = vToDistributeKeys[i].response;
                    }
                }
            }
            std::vector<KeyPoint>& keypoints = allKeypoints[level];
            keypoints = DistributeOctTree(vToDistributeKeys, minBorderX, maxBorderX, minBorderY, maxBorderY, mnFeaturesPerLevel[level]);
            const int scaledPatchSize = (int)(PATCH_SIZE * mvScaleFactor[level]);
            for (auto& kp : keypoints) {
                kp.x += minBorderX; kp.y += minBorderY; kp.octave = level; kp.size = (float)scaledPatchSize;
            }
            if (dbg && dbg->n_selected) dbg->n_selected[level] = (int)keypoints.size();
        }
    }
    // ORBExtractor.cpp:496-562
    int operator()(const View& image, hso_keypoint* out_kps, uint8_t* out_desc, int cap, hso_extract_debug* dbg) {
        if (image.cols == 0 || image.rows == 0 || !image.data) return 0;
        ComputePyramid(image);
        std::vector<std::vector<KeyPoint>> allKeypoints;
        ComputeKeyPointsOctTree(allKeypoints, dbg);
        int nkeypoints = 0;
        for (int level = 0; level < nlevels; ++level) nkeypoints += (int)allKeypoints[level].size();
        std::vector<uint8_t> descriptors_raw((size_t)nkeypoints * 32, 0);
        std::vector<KeyPoint> keypoints_out; keypoints_out.reserve(nkeypoints);
        int offset = 0;
        for (int level = 0; level < nlevels; ++level) {
            std::vector<KeyPoint>& keypoints = allKeypoints[level];
            int nkeypointsLevel = (int)keypoints.size();
            const View& lv = mvImagePyramid[level];
            if (dbg && dbg->pyramid && dbg->pyramid[level]) std::memcpy(dbg->pyramid[level], lv.data, (size_t)lv.cols * lv.rows);
            if (nkeypointsLevel == 0 && !(dbg && dbg->blurred && dbg->blurred[level])) continue;
            std::vector<uint8_t> workingMat((size_t)lv.cols * lv.rows);
            gaussianBlur7(lv.data, lv.cols, lv.rows, lv.step, workingMat.data(), lv.cols, taps);
            if (dbg && dbg->blurred && dbg->blurred[level]) std::memcpy(dbg->blurred[level], workingMat.data(), workingMat.size());
            if (nkeypointsLevel == 0) continue;
            finder.compute(workingMat.data(), lv.cols, keypoints, descriptors_raw.data() + (size_t)offset * 32);
            offset += nkeypointsLevel;
            if (level != 0) {
                float scale = mvScaleFactor[level];
                for (auto& kp : keypoints) { kp.x *= scale; kp.y *= scale; }
            }
            keypoints_out.insert(keypoints_out.end(), keypoints.begin(), keypoints.end());
        }
        int n = std::min(nkeypoints, cap);
        for (int i = 0; i < n; i++) {
            const KeyPoint& k = keypoints_out[i];
            out_kps[i] = hso_keypoint{ k.x, k.y, k.size, k.angle, k.response, k.octave };
        }
        if (n > 0) std::memcpy(out_desc, descriptors_raw.data(), (size_t)n * 32);
        return nkeypoints;
    }
};

/* ------------------------------------------------------------------ ORBDistance (DescriptorDistance.cpp:9-25) */
float ORBDistance(const uint8_t* D1, const uint8_t* D2)
{
    int32_t pa[8], pb[8];
    std::memcpy(pa, D1, 32); std::memcpy(pb, D2, 32);
    int dist = 0;
    for (int i = 0; i < 8; i++) {
        unsigned int v = pa[i] ^ pb[i];
        v = v - ((v >> 1) & 0x55555555);
        v = (v & 0x33333333) + ((v >> 2) & 0x33333333);
        dist += (((v + (v >> 4)) & 0xF0F0F0F) * 0x1010101) >> 24;
    }
    return static_cast<float>(dist);
}

/* ------------------------------------------------------------------ Stereomatcher (Stereomatcher.cpp:36-156) */
void computeStereoMatches(const hso_keypoint* mvKeys, const uint8_t* mDescriptors, int N,
                          const hso_keypoint* mvKeysRight, const uint8_t* mDescriptorsRight, int Nr,
                          const hso_stereo_params& sp, float* mvuRight, float* mvDepth, int32_t* best_idx, int32_t* best_dist)
{
    for (int i = 0; i < N; i++) { mvuRight[i] = -1.0f; mvDepth[i] = -1.0f; if (best_idx) best_idx[i] = -1; if (best_dist) best_dist[i] = -1; }
    const float TH_HIGH = sp.th_high, TH_LOW = sp.th_low;
    const float mbf = sp.mbf, mb = sp.mbf / sp.fx;
    const float dist_threshold = (TH_HIGH + TH_LOW) / 2;
    const int nRows = sp.n_rows;
    std::vector<std::vector<size_t>> vRowIndices(std::max(nRows, 0));
    for (int iR = 0; iR < Nr; iR++) {
        const hso_keypoint& kp = mvKeysRight[iR];
        const float kpY = kp.y;
        const float r = 2.0f * kp.size / sp.size_ref;
        const int maxr = (int)std::ceil(kpY + r);
        const int minr = (int)std::floor(kpY - r);
        for (int yi = minr; yi <= maxr; yi++)
            if (yi >= 0 && yi < nRows) vRowIndices[yi].push_back(iR);     // D2: bounds check
    }
    const float minZ = mb, minD = 0, maxD = mbf / minZ;
    std::vector<std::pair<float, int>> vDistIdx;
    for (int iL = 0; iL < N; iL++) {
        const hso_keypoint& kpL = mvKeys[iL];
        const int levelL = kpL.octave;
        const float vL = kpL.y, uL = kpL.x;
        if (!(vL >= 0) || (size_t)vL >= vRowIndices.size()) continue;      // D2
        const std::vector<size_t>& vCandidates = vRowIndices[(size_t)vL];
        if (vCandidates.empty()) continue;
        const float minU = uL - maxD, maxU = uL - minD;
        if (maxU < 0) continue;
        float bestDist = TH_HIGH;
        size_t bestIdxR = 0;
        const uint8_t* dL = mDescriptors + (size_t)iL * 32;
        for (size_t iC = 0; iC < vCandidates.size(); iC++) {
            const size_t iR = vCandidates[iC];
            const hso_keypoint& kpR = mvKeysRight[iR];
            if (kpR.octave < levelL - 1 || kpR.octave > levelL + 1) continue;
            const float uR = kpR.x;
            if (uR >= minU && uR <= maxU) {
                const float dist = ORBDistance(dL, mDescriptorsRight + iR * 32);
                if (dist < bestDist) { bestDist = dist; bestIdxR = iR; }
            }
        }
        if (bestDist < dist_threshold) {
            float uR0 = mvKeysRight[bestIdxR].x;
            float disparity = (uL - uR0);
            if (disparity >= minD && disparity < maxD) {
                if (disparity <= 0) { disparity = 0.01; uR0 = uL - 0.01; }
                mvDepth[iL] = mbf / disparity;
                mvuRight[iL] = uR0;
                if (best_idx) best_idx[iL] = (int32_t)bestIdxR;
                if (best_dist) best_dist[iL] = (int32_t)bestDist;
                vDistIdx.push_back(std::pair<float, int>(bestDist, iL));
            }
        }
    }
    if (vDistIdx.empty()) return;   // D2
    std::sort(vDistIdx.begin(), vDistIdx.end());
    const float median = vDistIdx[vDistIdx.size() / 2].first;
    const float thDist = 1.5f * 1.4f * median;
    for (int i = (int)vDistIdx.size() - 1; i >= 0; i--) {
        if (vDistIdx[i].first < thDist) break;
        mvuRight[vDistIdx[i].second] = -1;
        mvDepth[vDistIdx[i].second] = -1;
    }
}

} // namespace

/* ================================================================== C exports */
extern "C" {

int hso_cv_round_f(float v) { return cvRoundF(v); }
int hso_cv_round_d(double v) { return cvRoundD(v); }
float hso_fast_atan2(float y, float x) { return fastAtan2(y, x); }

void hso_default_params(hso_orb_params* p)   // ORBFactory.cpp:13-25
{
    std::memset(p, 0, sizeof(*p));
    p->nfeatures = 1000; p->scale_factor = 1.2f; p->nlevels = 8; p->cell_px = 30;
    p->ini_th_fast = 20; p->min_th_fast = 4; p->fast_threshold = 20;
    for (int k = 0; k < 7; k++) p->blur_taps[k] = kDefaultTaps[k];
}

int hso_scale_tables(const hso_orb_params* p, float* scale, float* inv_scale, float* sigma2, float* inv_sigma2, int32_t* quotas)
{
    ORBExtractor ex(*p);
    for (int i = 0; i < p->nlevels; i++) {
        if (scale) scale[i] = ex.mvScaleFactor[i];
        if (inv_scale) inv_scale[i] = ex.mvInvScaleFactor[i];
        if (sigma2) sigma2[i] = ex.mvLevelSigma2[i];
        if (inv_sigma2) inv_sigma2[i] = ex.mvInvLevelSigma2[i];
        if (quotas) quotas[i] = ex.mnFeaturesPerLevel[i];
    }
    return p->nlevels;
}

void hso_pyramid_size(const hso_orb_params* p, int w, int h, int level, int32_t* lw, int32_t* lh)
{
    ORBExtractor ex(*p); int a, b; ex.levelSize(w, h, level, a, b); *lw = a; *lh = b;
}

void hso_cell_grid(const hso_orb_params* p, int lw, int lh, int32_t* ncols, int32_t* nrows, int32_t* wcell, int32_t* hcell)
{
    ORBExtractor ex(*p); int a, b, c, d; ex.cellGrid(lw, lh, a, b, c, d); *ncols = a; *nrows = b; *wcell = c; *hcell = d;
}

void hso_umax(int32_t* umax16) { ORBFinder f(20); for (int i = 0; i < 16; i++) umax16[i] = f.umax[i]; }
const int32_t* hso_pattern(void) { static ORBFinder f(20); return f.pattern; }

void hso_resize_linear_u8(const uint8_t* src, int sw, int sh, int sstride, uint8_t* dst, int dw, int dh, int dstride)
{ resizeLinearU8(src, sw, sh, sstride, dst, dw, dh, dstride); }
void hso_preprocess_size(int w, int h, float scale, int32_t* ow, int32_t* oh) { int a, b; preprocessSize(w, h, scale, &a, &b); *ow = a; *oh = b; }
int hso_preprocess(const uint8_t* src, int w, int h, int sstride, int channels, int rgb, float scale, uint8_t* dst, int dstride)
{ return preprocessImg(src, w, h, sstride, channels, rgb, scale, dst, dstride); }

int hso_fast9_16(const uint8_t* img, int w, int h, int stride, int threshold, int nonmax, int32_t* out_xys, int cap)
{
    std::vector<KeyPoint> kps;
    FAST(img, w, h, stride, threshold, nonmax != 0, kps);
    int n = std::min((int)kps.size(), cap);
    for (int i = 0; i < n; i++) { out_xys[3 * i] = (int)kps[i].x; out_xys[3 * i + 1] = (int)kps[i].y; out_xys[3 * i + 2] = (int)kps[i].response; }
    return (int)kps.size();
}

void hso_gaussian_blur7(const uint8_t* src, int w, int h, int sstride, uint8_t* dst, int dstride, const uint16_t* taps7)
{ gaussianBlur7(src, w, h, sstride, dst, dstride, taps7 ? taps7 : kDefaultTaps); }

float hso_ic_angle(const uint8_t* img, int stride, float x, float y)
{ static ORBFinder f(20); return f.intensityCentroidAngle(img, stride, x, y); }

void hso_orb_descriptor(const uint8_t* img, int stride, float x, float y, float angle, uint8_t* desc32)
{ static ORBFinder f(20); f.computeOrbDescriptor(img, stride, x, y, angle, desc32); }

int hso_hamming256(const uint8_t* a, const uint8_t* b) { return (int)ORBDistance(a, b); }

int hso_distribute_octtree(const float* cand_xyr, int n, int minX, int maxX, int minY, int maxY, int N, int32_t* out_idx, int cap)
{
    std::vector<KeyPoint> v(n);
    for (int i = 0; i < n; i++) { v[i] = KeyPoint{ cand_xyr[3 * i], cand_xyr[3 * i + 1], 7.f, -1.f, cand_xyr[3 * i + 2], 0, i }; }
    std::vector<KeyPoint> r = DistributeOctTree(v, minX, maxX, minY, maxY, N);
    int m = std::min((int)r.size(), cap);
    for (int i = 0; i < m; i++) out_idx[i] = r[i].src;
    return (int)r.size();
}

int hso_level_candidates(const hso_orb_params* p, const uint8_t* level_img, int lw, int lh, int stride, float* out_xyr, int cap)
{
    ORBExtractor ex(*p);
    std::vector<KeyPoint> v;
    ex.levelCandidates(View{ level_img, lw, lh, stride }, v);
    int m = std::min((int)v.size(), cap);
    for (int i = 0; i < m; i++) { out_xyr[3 * i] = v[i].x; out_xyr[3 * i + 1] = v[i].y; out_xyr[3 * i + 2] = v[i].response; }
    return (int)v.size();
}

int hso_orb_extract(const hso_orb_params* p, const uint8_t* img, int w, int h, int stride,
                    hso_keypoint* kps, uint8_t* desc, int cap, hso_extract_debug* dbg)
{
    if (!p || p->nlevels < 1 || p->nlevels > 32 || p->nfeatures < 1) return -1;
    ORBExtractor ex(*p);
    return ex(View{ img, w, h, stride }, kps, desc, cap, dbg);
}

int hso_stereo_match(const hso_keypoint* kpsL, const uint8_t* descL, int nL,
                     const hso_keypoint* kpsR, const uint8_t* descR, int nR,
                     const hso_stereo_params* sp, float* uRight, float* depth, int32_t* best_idx, int32_t* best_dist)
{
    computeStereoMatches(kpsL, descL, nL, kpsR, descR, nR, *sp, uRight, depth, best_idx, best_dist);
    int n = 0; for (int i = 0; i < nL; i++) n += depth[i] > 0;
    return n;
}

int hso_stereo_frontend(const hso_orb_params* p, const hso_stereo_params* sp,
                        const uint8_t* imgL, const uint8_t* imgR, int w, int h, int stride,
                        hso_keypoint* kpsL, uint8_t* descL, int32_t* nL,
                        hso_keypoint* kpsR, uint8_t* descR, int32_t* nR, int cap,
                        float* uRight, float* depth)
{
    // ImageProcessing.cpp:82-84: left on a fresh std::thread, right on the calling thread, two instances
    int nl = 0, nr = 0;
    std::thread orb_thread([&] { ORBExtractor ex(*p); nl = ex(View{ imgL, w, h, stride }, kpsL, descL, cap, nullptr); });
    { ORBExtractor ex(*p); nr = ex(View{ imgR, w, h, stride }, kpsR, descR, cap, nullptr); }
    orb_thread.join();
    nl = std::min(nl, cap); nr = std::min(nr, cap);
    *nL = nl; *nR = nr;
    computeStereoMatches(kpsL, descL, nl, kpsR, descR, nr, *sp, uRight, depth, nullptr, nullptr);   // :100-103
    return nl;
}

} // extern "C"
