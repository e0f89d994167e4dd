// kernels_fast.hip — K2+K3: FAST-9/16 corner detection, corner score and 3x3 non-max suppression, per cell.
//
// Replaces the per-cell loop of ORBExtractor::ComputeKeyPointsOctTree (src/features/ORBExtractor.cpp:430-470),
// which calls ORBFinder::detect = cv::FAST(cell, kps, 20, true) (src/features/low_level/ORBFinder.cpp:66-68)
// once per ~31x31 cell (6342 calls per 1080p frame).  Semantics that must hold for bit-exact keypoints:
//   * a cell's sub-image is its interior plus a 3 px apron; cell interiors tile [19,w-19) x [19,h-19);
//   * NMS is 3x3, strict, and *per cell*: a neighbour outside the cell interior counts as score 0
//     (cv::FAST zero-fills its score rows and never scores the 3 px frame of the Mat it is given);
//   * score = max(t, max_arc min(v-ring), max_arc min(ring-v)) - 1 over the 16 arcs of length 9.
// The reference's cell == one workgroup here: the tile (interior + apron) is staged in LDS with coalesced
// dword row loads, the score map lives only in LDS, so the only HBM traffic is one read of the level.
// One launch covers every level of every image of the batch (blockIdx.x enumerates cells of all levels).
//
// Output: unordered candidate records per (image, level), appended with one global atomic per cell:
//   cand_xy = y<<16 | x          (coordinates relative to (16,16), as in vToDistributeKeys)
//   cand_sk = score<<24 | cell   (cell = row-major cell index; with (y,x) it restores the reference's
//                                 vToDistributeKeys order, which only matters for response ties)
// Bound: integer VALU + LDS byte reads; HBM bytes = P per frame.
#include "hs_internal.h"

#define TILE_PITCH 80                        // >= 3 (dword misalignment) + HS_MAX_CELL + 6, multiple of 4
#define TILE_ROWS (HS_MAX_CELL + 6)
#define SCORE_PITCH (HS_MAX_CELL + 4)        // interior + 1 px zero frame, padded
#define SCORE_ROWS (HS_MAX_CELL + 2)
#define MAX_OUT (HS_MAX_CELL * HS_MAX_CELL / 4)

__device__ __forceinline__ int fast_corner_score(const int (&d)[16], int t)
{
    int lo2[16], lo4[16], lo8[16], hi2[16], hi4[16], hi8[16];
#pragma unroll
    for (int k = 0; k < 16; k++) { lo2[k] = min(d[k], d[(k + 1) & 15]); hi2[k] = max(d[k], d[(k + 1) & 15]); }
#pragma unroll
    for (int k = 0; k < 16; k++) { lo4[k] = min(lo2[k], lo2[(k + 2) & 15]); hi4[k] = max(hi2[k], hi2[(k + 2) & 15]); }
#pragma unroll
    for (int k = 0; k < 16; k++) { lo8[k] = min(lo4[k], lo4[(k + 4) & 15]); hi8[k] = max(hi4[k], hi4[(k + 4) & 15]); }
    int a0 = t;
#pragma unroll
    for (int k = 0; k < 16; k++) a0 = max(a0, min(lo8[k], d[(k + 8) & 15]));      // arcs of 9: d[k..k+8]
    int b0 = -a0;
#pragma unroll
    for (int k = 0; k < 16; k++) b0 = min(b0, max(hi8[k], d[(k + 8) & 15]));
    return -b0 - 1;
}

__global__ __launch_bounds__(256) void k_fast_cells(const HsLevel* __restrict__ lv, int nlevels, HsImg0 img0, int fast_th,
                                                    uint32_t* __restrict__ cand_xy, uint32_t* __restrict__ cand_sk,
                                                    int32_t* __restrict__ cand_count, uint64_t cand_img_stride)
{
    __shared__ __attribute__((aligned(16))) uint8_t tile[TILE_ROWS * TILE_PITCH];
    __shared__ __attribute__((aligned(16))) uint8_t score[SCORE_ROWS * SCORE_PITCH];
    __shared__ uint16_t clist[HS_MAX_CELL * HS_MAX_CELL];
    __shared__ uint32_t out_xy[MAX_OUT];
    __shared__ uint8_t out_s[MAX_OUT];
    __shared__ int n_corner, n_out, out_base;

    const int tid = threadIdx.x;
    const int img = blockIdx.y;
    int level = 0;
    while (level + 1 < nlevels && (int)blockIdx.x >= lv[level + 1].cell_begin) level++;
    const HsLevel& L = lv[level];
    const int c = blockIdx.x - L.cell_begin;
    const int ci = c / L.ncols, cj = c - ci * L.ncols;
    const int iniX = HS_BORDER + cj * L.wcell, iniY = HS_BORDER + ci * L.hcell;
    const int maxX = min(iniX + L.wcell + 6, L.w - HS_BORDER), maxY = min(iniY + L.hcell + 6, L.h - HS_BORDER);
    const int tw = maxX - iniX, th = maxY - iniY;      // sub-image handed to cv::FAST
    if (tw < 7 || th < 7) return;                       // reference skip rules (:435,444) / FAST on < 7 rows
    const int iw = tw - 6, ih = th - 6;                 // interior = pixels FAST can report

    const uint8_t* base; size_t pitch;
    if (level == 0) { base = hs_img0_ptr(img0, img); pitch = img0.row_stride; }
    else { base = L.base + (size_t)img * L.img_stride; pitch = L.pitch; }

    if (tid == 0) { n_corner = 0; n_out = 0; }
    // zero the score frame
    for (int i = tid; i < SCORE_ROWS * SCORE_PITCH / 4; i += 256) reinterpret_cast<uint32_t*>(score)[i] = 0;

    // ---- stage the tile: coalesced dword loads of each row when alignment allows
    const int a0 = iniX & ~3;
    const int off = iniX - a0;                          // tile x = off + (x - iniX)
    if ((((uintptr_t)base | pitch) & 3) == 0) {
        const int ndw = (off + tw + 3) >> 2;            // <= 19
        for (int i = tid; i < th * ndw; i += 256) {
            int r = i / ndw, q = i - r * ndw;
            uint32_t v = *reinterpret_cast<const uint32_t*>(base + (size_t)(iniY + r) * pitch + a0 + 4 * q);
            *reinterpret_cast<uint32_t*>(&tile[r * TILE_PITCH + 4 * q]) = v;
        }
    } else {
        for (int i = tid; i < th * tw; i += 256) {
            int r = i / tw, q = i - r * tw;
            tile[r * TILE_PITCH + off + q] = base[(size_t)(iniY + r) * pitch + iniX + q];
        }
    }
    __syncthreads();

    // ---- segment test: 16-bit darker / brighter ring masks, 9 contiguous (cyclic) set bits
    const int npix = iw * ih;
    const int t = fast_th;
    constexpr int RO[16] = { 3 * TILE_PITCH + 0, 3 * TILE_PITCH + 1, 2 * TILE_PITCH + 2, 1 * TILE_PITCH + 3,
                             0 * TILE_PITCH + 3, -1 * TILE_PITCH + 3, -2 * TILE_PITCH + 2, -3 * TILE_PITCH + 1,
                             -3 * TILE_PITCH + 0, -3 * TILE_PITCH - 1, -2 * TILE_PITCH - 2, -1 * TILE_PITCH - 3,
                             0 * TILE_PITCH - 3, 1 * TILE_PITCH - 3, 2 * TILE_PITCH - 2, 3 * TILE_PITCH - 1 };
    for (int p0 = 0; p0 < npix; p0 += 256) {
        int p = p0 + tid;
        bool valid = p < npix;
        int pp = valid ? p : 0;
        int py = pp / iw, px = pp - py * iw;
        const uint8_t* ctr = &tile[(py + 3) * TILE_PITCH + off + px + 3];
        int v = ctr[0];
        int lo = v - t, hi = v + t;
        // quick reject on the four axis pairs (any 9-arc holds one pixel of every antipodal pair)
        int r0 = ctr[RO[0]], r8 = ctr[RO[8]], r4 = ctr[RO[4]], r12 = ctr[RO[12]];
        bool dk = (r0 < lo || r8 < lo) && (r4 < lo || r12 < lo);
        bool br = (r0 > hi || r8 > hi) && (r4 > hi || r12 > hi);
        bool maybe = valid && (dk || br);
        if (__ballot(maybe) == 0ull) continue;
        uint32_t mdark = 0, mbright = 0;
        if (maybe) {
#pragma unroll
            for (int k = 0; k < 16; k++) {
                int r = ctr[RO[k]];
                mdark |= (uint32_t)(r < lo) << k;
                mbright |= (uint32_t)(r > hi) << k;
            }
            uint32_t m = mdark | (mdark << 16);
            uint32_t x = m & (m >> 1); x &= x >> 2; x &= x >> 4; x &= m >> 8;
            uint32_t m2 = mbright | (mbright << 16);
            uint32_t y = m2 & (m2 >> 1); y &= y >> 2; y &= y >> 4; y &= m2 >> 8;
            if (((x | y) & 0xFFFFu) != 0) {
                int slot = atomicAdd(&n_corner, 1);
                clist[slot] = (uint16_t)p;
            }
        }
    }
    __syncthreads();

    // ---- corner score for the (few) corners, dense lanes
    const int nc = n_corner;
    for (int i = tid; i < nc; i += 256) {
        int p = clist[i];
        int py = p / iw, px = p - py * iw;
        const uint8_t* ctr = &tile[(py + 3) * TILE_PITCH + off + px + 3];
        int v = ctr[0];
        int d[16];
#pragma unroll
        for (int k = 0; k < 16; k++) d[k] = v - (int)ctr[RO[k]];
        score[(py + 1) * SCORE_PITCH + px + 1] = (uint8_t)fast_corner_score(d, t);
    }
    __syncthreads();

    // ---- 3x3 strict NMS inside the cell
    for (int i = tid; i < nc; i += 256) {
        int p = clist[i];
        int py = p / iw, px = p - py * iw;
        const uint8_t* sc = &score[(py + 1) * SCORE_PITCH + px + 1];
        int s = sc[0];
        bool keep = s > sc[1] && s > sc[-1] &&
                    s > sc[-SCORE_PITCH - 1] && s > sc[-SCORE_PITCH] && s > sc[-SCORE_PITCH + 1] &&
                    s > sc[SCORE_PITCH - 1] && s > sc[SCORE_PITCH] && s > sc[SCORE_PITCH + 1];
        if (keep) {
            int slot = atomicAdd(&n_out, 1);
            // coordinates relative to (minBorderX, minBorderY): x_local + j*wCell (ORBExtractor.cpp:463-464)
            int xr = px + 3 + cj * L.wcell, yr = py + 3 + ci * L.hcell;
            out_xy[slot] = ((uint32_t)yr << 16) | (uint32_t)xr;
            out_s[slot] = (uint8_t)s;
        }
    }
    __syncthreads();
    const int no = n_out;
    if (no == 0) return;
    int32_t* cnt = &cand_count[img * nlevels + level];
    if (tid == 0) out_base = atomicAdd(cnt, no);
    __syncthreads();
    const int ob = out_base;
    uint32_t* gxy = cand_xy + (size_t)img * cand_img_stride + L.cand_off;
    uint32_t* gsk = cand_sk + (size_t)img * cand_img_stride + L.cand_off;
    for (int i = tid; i < no; i += 256) {
        if (ob + i < L.cand_cap) {
            gxy[ob + i] = out_xy[i];
            gsk[ob + i] = ((uint32_t)out_s[i] << 24) | (uint32_t)c;
        }
    }
}

void hs_launch_fast(const HsLevel* d_lv, int nlevels, HsImg0 img0, int batch, int total_cells, int fast_th,
                    uint32_t* cand_xy, uint32_t* cand_sk, int32_t* cand_count, uint64_t cand_img_stride, hipStream_t s)
{
    if (total_cells <= 0) return;
    dim3 grid(total_cells, batch, 1);
    hipLaunchKernelGGL(k_fast_cells, grid, dim3(256), 0, s, d_lv, nlevels, img0, fast_th, cand_xy, cand_sk, cand_count, cand_img_stride);
}
