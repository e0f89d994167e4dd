// kernels_preprocess.hip — the camera frame as it arrives -> the grey level 0 the extractor works on.
//
// Replaces ImageProcessing::PreProcessImg (src/main/ImageProcessing.cpp:118-138), which sits INSIDE the reference's timing bracket (:70 / :112) in front
// of both extractor calls (:76-77):   cv::resize(img, img, cv::Size(), fscale, fscale);   cvtColor(img, img, CV_RGB2GRAY | CV_BGR2GRAY | CV_RGBA2GRAY | CV_BGRA2GRAY)
// on a 1-, 3- or 4-channel 8-bit frame.  For the reference's own cameras (config/sample_primary_config_file.yaml:35-42,63-70) that is a copy for the
// 1280x720 stereo pair (scale 1.0) and a 2704x2028x3 -> 1352x1014 reduction for the "Imaging" camera (scale 0.5) — milliseconds of CPU in front of a
// 0.1 ms extraction.  Here the frame crosses PCIe as the camera delivered it and no host core touches its pixels.
//
// OpenCV 3.4 semantics (the same status as every other primitive here: restated from its published algorithm, "parity unpinned" — DESIGN.md §1):
//   size     (cvRound(w * (double)fscale), cvRound(h * (double)fscale)); the tables use scale = 1. / (double)fscale
//   MODE 0   the size does not change: a copy (then grey)
//   MODE 1   scale_x == scale_y == 2 exactly: INTER_LINEAR silently becomes INTER_AREA's fast path, D = (S00 + S01 + S10 + S11 + 2) >> 2 per channel for
//            the blocks that have all four samples; a trailing partial block (odd source size) is saturate_cast<uchar>((float)sum / count)
//   MODE 2   anything else: the 11-bit fixed-point bilinear of the pyramid (kernels_pyramid.hip) per channel, its tables evaluated on the fly (the same
//            double / float expressions as the host tables of the pyramid: IEEE arithmetic, no contraction)
//   grey     (R * 4899 + G * 9617 + B * 1868 + (1 << 13)) >> 14, alpha ignored
//
// Mapping: byte work, HBM bound.  A lane makes 4 consecutive grey pixels (one dword store); its source bytes are one contiguous run per source row
// (4 CN bytes, 8 CN bytes for MODE 1) that starts on a dword when the frame's rows do, so a wave reads 256 CN / 512 CN contiguous bytes per row with dword
// loads.  Frames with odd bases / strides and the last, partial quad of a row take byte loads.  Algorithmic bytes: w h CN read + ow oh written.
#include "hs_internal.h"

struct HsPreArgs {
    const uint8_t* src; uint64_t src_row_stride, src_img_stride;
    uint8_t* dst; uint64_t dst_row_stride, dst_img_stride;
    int32_t sw, sh, dw, dh;
    int32_t rgb;                 // 1: channel 0 is red
    int32_t dst_may_pad;         // the grey rows may be written up to the next multiple of 4 columns (our own level-0 buffer: pitch % 64 == 0)
    double scale;                // 1. / (double)fscale (MODE 2)
};

__device__ __forceinline__ uint32_t pre_grey(uint32_t c0, uint32_t c1, uint32_t c2, int rgb)
{
    const uint32_t r = rgb ? c0 : c2, b = rgb ? c2 : c0;
    return (r * 4899u + c1 * 9617u + b * 1868u + (1u << 13)) >> 14;
}

template <int CN, int MODE, bool ALIGNED>
__global__ __launch_bounds__(256) void k_preprocess(HsPreArgs A)
{
    const int img = blockIdx.z;
    const int dy = blockIdx.y * 4 + (threadIdx.x >> 6);
    const int dx0 = (blockIdx.x * 64 + (threadIdx.x & 63)) * 4;
    if (dy >= A.dh || dx0 >= A.dw) return;
    const uint8_t* const simg = A.src + (size_t)img * A.src_img_stride;
    uint8_t* const drow = A.dst + (size_t)img * A.dst_img_stride + (size_t)dy * A.dst_row_stride;
    const bool full_quad = dx0 + 4 <= A.dw;
    uint32_t g[4] = { 0, 0, 0, 0 };

    if (MODE == 0) {
        const uint8_t* S = simg + (size_t)dy * A.src_row_stride + (size_t)dx0 * CN;
        uint8_t b[4 * CN];
        if (ALIGNED && full_quad) {
            uint32_t w[CN];
#pragma unroll
            for (int i = 0; i < CN; i++) w[i] = reinterpret_cast<const uint32_t*>(S)[i];
#pragma unroll
            for (int i = 0; i < 4 * CN; i++) b[i] = (uint8_t)(w[i >> 2] >> (8 * (i & 3)));
        } else {
#pragma unroll
            for (int i = 0; i < 4 * CN; i++) b[i] = (dx0 + i / CN < A.dw) ? S[i] : (uint8_t)0;
        }
#pragma unroll
        for (int p = 0; p < 4; p++) g[p] = CN == 1 ? b[p] : pre_grey(b[p * CN], b[p * CN + (CN > 1 ? 1 : 0)], b[p * CN + (CN > 2 ? 2 : 0)], A.rgb);
    } else if (MODE == 1) {
        const int sy0 = 2 * dy, wfull = A.sw >> 1;
        const bool rows_full = sy0 + 1 < A.sh;
        if (ALIGNED && full_quad && rows_full && dx0 + 4 <= wfull) {
            const uint8_t* S0 = simg + (size_t)sy0 * A.src_row_stride + (size_t)dx0 * 2 * CN;
            const uint8_t* S1 = S0 + A.src_row_stride;
            uint32_t w0[2 * CN], w1[2 * CN];
#pragma unroll
            for (int i = 0; i < 2 * CN; i++) { w0[i] = reinterpret_cast<const uint32_t*>(S0)[i]; w1[i] = reinterpret_cast<const uint32_t*>(S1)[i]; }
            auto by = [&](const uint32_t (&w)[2 * CN], int i) -> uint32_t { return (w[i >> 2] >> (8 * (i & 3))) & 0xFFu; };
#pragma unroll
            for (int p = 0; p < 4; p++) {
                uint32_t c[3] = { 0, 0, 0 };
#pragma unroll
                for (int k = 0; k < (CN < 3 ? CN : 3); k++)
                    c[k] = (by(w0, 2 * p * CN + k) + by(w0, (2 * p + 1) * CN + k) + by(w1, 2 * p * CN + k) + by(w1, (2 * p + 1) * CN + k) + 2u) >> 2;
                g[p] = CN == 1 ? c[0] : pre_grey(c[0], c[1], c[2], A.rgb);
            }
        } else {
            for (int p = 0; p < 4; p++) {
                const int dx = dx0 + p, sx0 = 2 * dx;
                if (dx >= A.dw) break;
                uint32_t c[3] = { 0, 0, 0 };
                for (int k = 0; k < (CN < 3 ? CN : 3); k++) {
                    if (sy0 >= A.sh || sx0 >= A.sw) { c[k] = 0; continue; }
                    uint32_t sum = 0; int count = 0;
                    for (int yy = 0; yy < 2 && sy0 + yy < A.sh; yy++)
                        for (int xx = 0; xx < 2 && sx0 + xx < A.sw; xx++) { sum += simg[(size_t)(sy0 + yy) * A.src_row_stride + (size_t)(sx0 + xx) * CN + k]; count++; }
                    if (dx < wfull && rows_full) c[k] = (sum + 2u) >> 2;
                    else { const int v = (int)rintf((float)sum / (float)count); c[k] = (uint32_t)min(max(v, 0), 255); }
                }
                g[p] = CN == 1 ? c[0] : pre_grey(c[0], c[1], c[2], A.rgb);
            }
        }
    } else {
        // resize.cpp: fy = (float)((dy + 0.5) * scale_y - 0.5); sy = cvFloor(fy); fy -= sy; the rows are clipped, the weights kept
        float fy = (float)(((double)dy + 0.5) * A.scale - 0.5);
        int sy = (int)fy; sy -= (float)sy > fy;
        fy -= (float)sy;
        auto sat_short = [](float v) { const int i = (int)rintf(v); return min(max(i, -32768), 32767); };
        const int b0 = sat_short((1.f - fy) * 2048.f), b1 = sat_short(fy * 2048.f);
        const uint8_t* S0 = simg + (size_t)min(max(sy, 0), A.sh - 1) * A.src_row_stride;
        const uint8_t* S1 = simg + (size_t)min(max(sy + 1, 0), A.sh - 1) * A.src_row_stride;
        for (int p = 0; p < 4; p++) {
            const int dx = dx0 + p;
            if (dx >= A.dw) break;
            float fx = (float)(((double)dx + 0.5) * A.scale - 0.5);
            int sx = (int)fx; sx -= (float)sx > fx;
            fx -= (float)sx;
            if (sx < 0) { fx = 0.f; sx = 0; }
            const bool past = sx + 1 >= A.sw;                  // dx >= xmax: the right tap is not read (monotone in dx: xmax = the first such dx)
            if (sx >= A.sw - 1) { fx = 0.f; sx = A.sw - 1; }
            const int a0 = sat_short((1.f - fx) * 2048.f), a1 = sat_short(fx * 2048.f);
            uint32_t c[3] = { 0, 0, 0 };
            for (int k = 0; k < (CN < 3 ? CN : 3); k++) {
                const size_t o = (size_t)sx * CN + k;
                int h0, h1;
                if (!past) { h0 = S0[o] * a0 + S0[o + CN] * a1; h1 = S1[o] * a0 + S1[o + CN] * a1; }
                else { h0 = S0[o] * 2048; h1 = S1[o] * 2048; }
                c[k] = (uint32_t)((((b0 * (h0 >> 4)) >> 16) + ((b1 * (h1 >> 4)) >> 16) + 2) >> 2) & 0xFFu;
            }
            g[p] = CN == 1 ? c[0] : pre_grey(c[0], c[1], c[2], A.rgb);
        }
    }
    if ((full_quad || A.dst_may_pad) && (((uintptr_t)drow | (uintptr_t)dx0) & 3) == 0) {
        *reinterpret_cast<uint32_t*>(drow + dx0) = g[0] | (g[1] << 8) | (g[2] << 16) | (g[3] << 24);
    } else {
        for (int p = 0; p < 4 && dx0 + p < A.dw; p++) drow[dx0 + p] = (uint8_t)g[p];
    }
}

// cvRound of the double product (saturate_cast<int>(ssize.width * inv_scale_x)); host side, also the C ABI's hs_preprocess_size
void hs_preprocess_out_size(int w, int h, float fscale, int* ow, int* oh)
{
    const double inv = (double)fscale;
    *ow = (int)nearbyint((double)w * inv); *oh = (int)nearbyint((double)h * inv);
}

// 0 copy, 1 the 2x2 area path, 2 bilinear (resize.cpp: is_area_fast && iscale_x == 2 && iscale_y == 2)
int hs_preprocess_mode(int w, int h, int ow, int oh, float fscale)
{
    if (ow == w && oh == h) return 0;
    const double scale = 1. / (double)fscale;
    const int iscale = (int)nearbyint(scale);
    return (fabs(scale - iscale) < 2.220446049250313e-16 && iscale == 2) ? 1 : 2;
}

void hs_launch_preprocess(const uint8_t* d_src, int sw, int sh, size_t src_row_stride, size_t src_img_stride, int channels, int rgb, float fscale,
                          uint8_t* d_dst, int dw, int dh, size_t dst_row_stride, size_t dst_img_stride, int dst_may_pad, int batch, hipStream_t s)
{
    HsPreArgs A{ d_src, (uint64_t)src_row_stride, (uint64_t)src_img_stride, d_dst, (uint64_t)dst_row_stride, (uint64_t)dst_img_stride, sw, sh, dw, dh, rgb ? 1 : 0, dst_may_pad, 1. / (double)fscale };
    const int mode = hs_preprocess_mode(sw, sh, dw, dh, fscale);
    const bool aligned = (((uintptr_t)d_src | src_row_stride | src_img_stride) & 3) == 0;
    dim3 grid((dw + 255) / 256, (dh + 3) / 4, batch), block(256);
#define PRE_K(CN_, M_) do { if (aligned) hipLaunchKernelGGL((k_preprocess<CN_, M_, true>), grid, block, 0, s, A); else hipLaunchKernelGGL((k_preprocess<CN_, M_, false>), grid, block, 0, s, A); } while (0)
#define PRE_M(CN_) do { if (mode == 0) PRE_K(CN_, 0); else if (mode == 1) PRE_K(CN_, 1); else PRE_K(CN_, 2); } while (0)
    if (channels == 1) PRE_M(1); else if (channels == 3) PRE_M(3); else PRE_M(4);
#undef PRE_M
#undef PRE_K
}
