// kernels_pyramid.hip — K1: 8-bit bilinear image pyramid.
// Replaces ORBExtractor::ComputePyramid (src/features/ORBExtractor.cpp:564-589), i.e. a chain of
// cv::resize(level-1 -> level, INTER_LINEAR) calls.  The EDGE_THRESHOLD border the reference adds with
// copyMakeBorder is never read downstream, so levels are stored border-less.
//
// Arithmetic = OpenCV 3.4 fixed-point path: 11-bit horizontal coefficients, 8-bit vertical combine
//   dst = ((b0*(H0>>4))>>16) + ((b1*(H1>>4))>>16) + 2) >> 2,  H = S[sx]*a0 + S[sx+1]*a1.
// The coefficient tables are built on the host with the same double/float expressions OpenCV uses.
//
// Bound: HBM/L2 bandwidth.  Algorithmic bytes per level = src px read once + dst px written once.
// Each lane produces 4 consecutive destination pixels and stores one dword (coalesced 256 B per wave row).
#include "hs_internal.h"

__global__ __launch_bounds__(256) void k_resize_level(const HsLevel* __restrict__ lv, int level, HsImg0 img0)
{
    const HsLevel& D = lv[level];
    const int img = blockIdx.z;
    const int dy = blockIdx.y * 4 + threadIdx.y;
    const int dx0 = (blockIdx.x * 64 + threadIdx.x) * 4;
    if (dy >= D.h || dx0 >= D.w) return;

    const uint8_t* sbase; size_t spitch; int sw, sh;
    if (level == 1) { sbase = hs_img0_ptr(img0, img); spitch = img0.row_stride; }
    else { const HsLevel& S = lv[level - 1]; sbase = S.base + (size_t)img * S.img_stride; spitch = S.pitch; }
    sw = lv[level - 1].w; sh = lv[level - 1].h;

    int sy = D.yofs[dy];
    const int b0 = D.ibeta[2 * dy], b1 = D.ibeta[2 * dy + 1];
    int sy0 = sy < 0 ? 0 : (sy >= sh ? sh - 1 : sy);
    int sy1 = sy + 1 < 0 ? 0 : (sy + 1 >= sh ? sh - 1 : sy + 1);
    const uint8_t* S0 = sbase + (size_t)sy0 * spitch;
    const uint8_t* S1 = sbase + (size_t)sy1 * spitch;

    uint32_t packed = 0;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        int dx = dx0 + i;
        if (dx < D.w) {
            int sx = D.xofs[dx];
            int sx1 = sx + 1 < sw ? sx + 1 : sw - 1;
            int h0, h1;
            if (dx < D.xmax) {
                int a0 = D.ialpha[2 * dx], a1 = D.ialpha[2 * dx + 1];
                h0 = S0[sx] * a0 + S0[sx1] * a1;
                h1 = S1[sx] * a0 + S1[sx1] * a1;
            } else {
                h0 = S0[sx] * 2048;
                h1 = S1[sx] * 2048;
            }
            int v = (((b0 * (h0 >> 4)) >> 16) + ((b1 * (h1 >> 4)) >> 16) + 2) >> 2;
            packed |= (uint32_t)(v & 0xFF) << (8 * i);
        }
    }
    uint8_t* drow = D.base + (size_t)img * D.img_stride + (size_t)dy * D.pitch;
    *reinterpret_cast<uint32_t*>(drow + dx0) = packed;   // pitch is a multiple of 64: padding bytes may be written
}

void hs_launch_pyramid(const HsLevel* d_lv, const HsLevel* h_lv, int nlevels, HsImg0 img0, int batch, hipStream_t s)
{
    for (int l = 1; l < nlevels; l++) {
        dim3 block(64, 4, 1);
        dim3 grid((h_lv[l].w + 255) / 256, (h_lv[l].h + 3) / 4, batch);
        hipLaunchKernelGGL(k_resize_level, grid, block, 0, s, d_lv, l, img0);
    }
}
