// kernels_fast.hip — K2+K3: FAST-9/16 corner detection, corner score and 3x3 non-max suppression, per cell.
//
// Replaces the per-cell loop of ORBExtractor::ComputeKeyPointsOctTree (src/features/ORBExtractor.cpp:430-470),
// which calls ORBFinder::detect = cv::FAST(cell, kps, 20, true) (src/features/low_level/ORBFinder.cpp:66-68)
// once per ~31x31 cell (6342 calls per 1080p frame).  Semantics that must hold for bit-exact keypoints:
//   * a cell's sub-image is its interior plus a 3 px apron; cell interiors tile [19,w-19) x [19,h-19);
//   * NMS is 3x3, strict, and *per cell*: a neighbour outside the cell interior counts as score 0
//     (cv::FAST zero-fills its score rows and never scores the 3 px frame of the Mat it is given);
//   * score = max(t, max_arc min(v-ring), max_arc min(ring-v)) - 1 over the 16 arcs of length 9.
//
// MI355X mapping.  The reference's cell is the unit of work: its tile (interior + apron) is staged in LDS with
// coalesced dword row loads and the score map lives only in LDS, so the only HBM traffic is one read of each level.
// The launch is PERSISTENT and every workgroup is ONE wavefront: ~28 of them per CU each walk a contiguous range of the
// (image, level, cell) list.  One wave per cell means no workgroup barriers at all (LDS hand-offs are ordered by the wave's own
// in-order LDS queue), list appends are ballot + mbcnt prefix counts in registers instead of LDS atomics, and a wave never idles
// at a barrier while its partner scores a handful of corners (measured: 1.20 ms one-cell-per-256-thread-workgroup ->
// 0.60 ms persistent 128-thread -> 0.5 ms single-wave, 32 frames of 1080p).  The dwords of the NEXT cell's tile are fetched into
// registers before the current cell is processed, which takes the ~2 us global-load latency off the per-cell critical path.
// Ranges are dealt so that workgroups that share an XCD (blockIdx % 8) own neighbouring cells and reuse each other's apron
// lines in that XCD's L2 (measured HBM over-fetch 1.15x).  Per cell:
//   pass 1  every pixel: compass-point quick reject -> "maybe" list in LDS (dense lanes for what follows)
//   pass 2  maybe pixels: 16-bit darker/brighter ring masks, 9 contiguous cyclic bits -> corner list
//   pass 3  corners: score into the LDS score tile;  pass 4: strict 3x3 NMS inside the cell
// Output: each cell owns a fixed slot range (no global atomics, deterministic placement):
//   cand_xy[cell slot] = y<<16 | x         (coordinates relative to (16,16), as in vToDistributeKeys)
//   cand_sk[cell slot] = score<<24 | cell  (cell = row-major cell index; with (y,x) it restores the reference's
//                                           vToDistributeKeys order, which only matters for response ties)
//   cell_count[image][global cell] = number of slots used.
// Bound: integer VALU + LDS byte reads; HBM bytes = P per frame.
#include "hs_internal.h"
#include <algorithm>
#include <cstdlib>

// LDS layout, sized on the host from the largest cell of the configured geometry (a 1080p frame needs ~9 KB per workgroup,
// so the wave limit, not LDS, decides how many cells a CU has in flight)
// TILE_PITCH / SCORE_PITCH are template parameters: compile-time pitches keep the 16 ring offsets immediate operands.
// Two variants: cells up to 37 px wide (every standard configuration) and the general one (up to HS_MAX_CELL).
struct FastLds {
    int32_t tile_pitch;      // >= 3 (dword misalignment) + max cell width + 6, multiple of 4
    int32_t score_pitch;     // max cell width + 2 (1 px zero frame), multiple of 4
    int32_t score_bytes;     // (max cell height + 2) * score_pitch
    int32_t off_score, off_list, total;
};
#define FAST_NT 64                           // one wavefront per workgroup (see above)
#ifndef FAST_WAVES_PER_SIMD
#define FAST_WAVES_PER_SIMD 6                // register budget: 80 VGPRs (the kernel is latency bound: resident waves are what it needs)
#endif

// The workgroup is one wave: its LDS operations execute in program order, so a hand-off through LDS only needs the LDS queue
// drained (no s_barrier, and no vmcnt wait that would expose the latency of the next cell's prefetch).
#define WAVE_LDS_FENCE() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")

// append: lanes with `flag` get consecutive slots after `base` (wave-uniform); returns the lane's slot, advances base
__device__ __forceinline__ int wave_append(bool flag, int& base)
{
    const unsigned long long m = __ballot(flag);
    const int slot = base + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
    base += (int)__popcll(m);
    return slot;
}

// Corner score of a pixel that passed the segment test with polarity `dark` (d[k] = v - ring[k] for dark, ring[k] - v for bright).
// cv::FAST's cornerScore<16> is max(t, max_arc min(v-ring), max_arc min(ring-v)) - 1; for a corner of one polarity the other
// polarity's term cannot exceed t (no 9-arc passes it) while its own term does, so score = max_arc min(d) - 1 with d of its own polarity.
__device__ __forceinline__ int fast_corner_score(const int (&d)[16])
{
    int lo2[16], lo4[16], lo8[16];
#pragma unroll
    for (int k = 0; k < 16; k++) lo2[k] = min(d[k], d[(k + 1) & 15]);
#pragma unroll
    for (int k = 0; k < 16; k++) lo4[k] = min(lo2[k], lo2[(k + 2) & 15]);
#pragma unroll
    for (int k = 0; k < 16; k++) lo8[k] = min(lo4[k], lo4[(k + 4) & 15]);
    int a0 = min(lo8[0], d[8]);
#pragma unroll
    for (int k = 1; k < 16; k++) a0 = max(a0, min(lo8[k], d[(k + 8) & 15]));      // arcs of 9: d[k..k+8]
    return a0 - 1;
}

struct CellGeom {           // wave-uniform description of one work item
    int img, level, ci, cj, c, gcell;
    int xoff, yoff;         // j*wCell, i*hCell: what the reference adds to cv::FAST's local coordinates (:463-464)
    int iniX, iniY, tw, th; // sub-image handed to cv::FAST
    int off, ndw;           // dword staging: tile x = off + (x - iniX); ndw dwords per row
    int ccap;               // slots this cell owns
    const uint8_t* rows;    // address of (a0, iniY) in the level
    size_t pitch;
    bool valid, aligned;
};

// geometry of cell (ci, cj) of `level` in image `img` — no divisions (the walk below steps cells incrementally)
__device__ __forceinline__ void cell_fill(CellGeom& g, const HsLevel* __restrict__ lv, const HsImg0& img0)
{
    const HsLevel& L = lv[g.level];
    g.c = g.ci * L.ncols + g.cj;
    g.gcell = L.cell_begin + g.c;
    g.xoff = g.cj * L.wcell; g.yoff = g.ci * L.hcell;
    g.iniX = HS_BORDER + g.xoff; g.iniY = HS_BORDER + g.yoff;
    const int maxX = min(g.iniX + L.wcell + 6, L.w - HS_BORDER), maxY = min(g.iniY + L.hcell + 6, L.h - HS_BORDER);
    g.tw = maxX - g.iniX; g.th = maxY - g.iniY;
    g.valid = g.tw >= 7 && g.th >= 7;                 // reference skip rules (:435,444) / cv::FAST on < 7 rows or columns
    g.ccap = ((L.wcell + 1) >> 1) * ((L.hcell + 1) >> 1);
    const uint8_t* base;
    if (g.level == 0) { base = hs_img0_ptr(img0, g.img); g.pitch = img0.row_stride; }
    else { base = L.base + (size_t)g.img * L.img_stride; g.pitch = L.pitch; }
    const int a0 = g.iniX & ~3;
    g.off = g.iniX - a0;
    g.ndw = (g.off + g.tw + 3) >> 2;
    g.aligned = (((uintptr_t)base | g.pitch) & 3) == 0;
    g.rows = base + (size_t)g.iniY * g.pitch + a0;
}

// work item w -> geometry (used once per workgroup)
__device__ __forceinline__ CellGeom cell_geom(const HsLevel* __restrict__ lv, int nlevels, const HsImg0& img0, int total_cells, int w)
{
    CellGeom g;
    g.img = w / total_cells;
    const int gcell = w - g.img * total_cells;
    int level = 0;
    while (level + 1 < nlevels && gcell >= lv[level + 1].cell_begin) level++;
    g.level = level;
    const int c = gcell - lv[level].cell_begin;
    g.ci = c / lv[level].ncols; g.cj = c - g.ci * lv[level].ncols;
    cell_fill(g, lv, img0);
    return g;
}

// the next work item: next column, row, level (skipping levels without cells), image
__device__ __forceinline__ void cell_next(CellGeom& g, const HsLevel* __restrict__ lv, int nlevels, const HsImg0& img0)
{
    if (++g.cj == lv[g.level].ncols) {
        g.cj = 0;
        if (++g.ci == lv[g.level].nrows) {
            g.ci = 0;
            do { if (++g.level == nlevels) { g.level = 0; g.img++; } } while (lv[g.level].ncols * lv[g.level].nrows == 0);
        }
    }
    cell_fill(g, lv, img0);
}

// NPRE = prefetched dwords per lane (64 * NPRE >= rows * dwords of the largest tile)
template <int TILE_PITCH, int SCORE_PITCH, int NPRE>
__global__ __launch_bounds__(FAST_NT, FAST_WAVES_PER_SIMD) void k_fast_cells(const HsLevel* __restrict__ lv, int nlevels, HsImg0 img0, int fast_th,
                                                    uint32_t* __restrict__ cand_xy, uint32_t* __restrict__ cand_sk,
                                                    int32_t* __restrict__ cell_count, uint64_t cand_img_stride,
                                                    int total_cells, int total_work, FastLds lds)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    uint8_t* const tile = smem;
    uint8_t* const score = smem + lds.off_score;
    uint16_t* const list = reinterpret_cast<uint16_t*>(smem + lds.off_list);   // "maybe" pixels, then (compacted in place) corners

    const int tid = threadIdx.x;
    const int t = fast_th;
    constexpr int RO[16] = { 3 * TILE_PITCH + 0, 3 * TILE_PITCH + 1, 2 * TILE_PITCH + 2, 1 * TILE_PITCH + 3,
                             0 * TILE_PITCH + 3, -1 * TILE_PITCH + 3, -2 * TILE_PITCH + 2, -3 * TILE_PITCH + 1,
                             -3 * TILE_PITCH + 0, -3 * TILE_PITCH - 1, -2 * TILE_PITCH - 2, -1 * TILE_PITCH - 3,
                             0 * TILE_PITCH - 3, 1 * TILE_PITCH - 3, 2 * TILE_PITCH - 2, 3 * TILE_PITCH - 1 };

    // contiguous range of work items; workgroups with equal blockIdx % 8 (same XCD) get neighbouring ranges
    const int nblk = gridDim.x, per_xcd = nblk >> 3;
    const int chunk = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
    const int per = (total_work + nblk - 1) / nblk;
    const int w_begin = chunk * per, w_end = min(total_work, w_begin + per);
    if (w_begin >= w_end) return;

    for (int i = tid; i < lds.score_bytes / 4; i += FAST_NT) reinterpret_cast<uint32_t*>(score)[i] = 0;

    uint32_t pre[NPRE];
    CellGeom g = cell_geom(lv, nlevels, img0, total_cells, w_begin);
    auto prefetch = [&](const CellGeom& q) {
        if (!(q.valid && q.aligned)) return;
        const float rcp = __builtin_amdgcn_rcpf((float)q.ndw);    // 1 ulp is plenty: (i+0.5)/ndw is >= 1/(2*19) away from an integer
        const int n = q.th * q.ndw;
#pragma unroll
        for (int j = 0; j < NPRE; j++) {
            if (FAST_NT * j >= n) break;                            // uniform: typical cells need 7 of the 24 slots
            int i = tid + FAST_NT * j;
            if (i < n) {
                int r = (int)(((float)i + 0.5f) * rcp), c = i - r * q.ndw;
                pre[j] = *reinterpret_cast<const uint32_t*>(q.rows + (size_t)r * q.pitch + 4 * c);
            }
        }
    };
    prefetch(g);

    for (int w = w_begin; w < w_end; w++) {
        const HsLevel& L = lv[g.level];
        int32_t* cnt = &cell_count[(size_t)g.img * total_cells + g.gcell];
        if (!g.valid) {
            if (tid == 0) *cnt = 0;
            if (w + 1 < w_end) { cell_next(g, lv, nlevels, img0); prefetch(g); }
            continue;
        }
        const int iw = g.tw - 6, ih = g.th - 6;                 // interior = pixels FAST can report
        const int off = g.off;
        // ---- stage the tile
        if (g.aligned) {
            const float rcp = __builtin_amdgcn_rcpf((float)g.ndw);
            const int n = g.th * g.ndw;
#pragma unroll
            for (int j = 0; j < NPRE; j++) {
                if (FAST_NT * j >= n) break;
                int i = tid + FAST_NT * j;
                if (i < n) {
                    int r = (int)(((float)i + 0.5f) * rcp), c = i - r * g.ndw;
                    *reinterpret_cast<uint32_t*>(&tile[r * TILE_PITCH + 4 * c]) = pre[j];
                }
            }
        } else {
            const uint8_t* src = g.rows + off;                  // (iniX, iniY)
            for (int i = tid; i < g.th * g.tw; i += FAST_NT) {
                int r = i / g.tw, c = i - r * g.tw;
                tile[r * TILE_PITCH + off + c] = src[(size_t)r * g.pitch + c];
            }
        }
        WAVE_LDS_FENCE();                                        // tile ready
        const CellGeom cur = g;
        if (w + 1 < w_end) { cell_next(g, lv, nlevels, img0); prefetch(g); }   // in flight during the passes

        // ---- pass 1 (every pixel): quick reject on the four compass points.  A 9-arc of the 16-ring always holds two
        //      ADJACENT compass points, i.e. (p0 or p8) and (p4 or p12).
        int n_maybe = 0;                                         // wave-uniform
        {
            // Four horizontally adjacent pixels per lane: the tile row is read as aligned dwords (centre dword, its two neighbours,
            // the dwords 3 rows above and below) and every operand is a byte lane of those registers (SDWA), so a pixel costs
            // ~12 VALU ops and 1.25 LDS loads instead of 5 byte loads + address math.
            const int c_first = off + 3, c_end = off + 3 + iw;   // tile columns of the interior
            const int g0 = c_first >> 2, ng = ((c_end + 3) >> 2) - g0;
            const int ntask = ih * ng;
            const float rcp_ng = __builtin_amdgcn_rcpf((float)ng);
            for (int q0 = 0; q0 < ntask; q0 += FAST_NT) {
                const int q = q0 + tid;
                const int qc = min(q, ntask - 1);
                const int py = (int)(((float)qc + 0.5f) * rcp_ng), gi = qc - py * ng;
                const int tc = (g0 + gi) << 2;
                const uint32_t* rowc = reinterpret_cast<const uint32_t*>(&tile[(py + 3) * TILE_PITCH + tc]);
                const uint32_t Cm = rowc[-1], Cc = rowc[0], Cp = rowc[1];
                const uint32_t Tt = *reinterpret_cast<const uint32_t*>(&tile[(py + 0) * TILE_PITCH + tc]);     // ring 8: (0,-3)
                const uint32_t Bb = *reinterpret_cast<const uint32_t*>(&tile[(py + 6) * TILE_PITCH + tc]);     // ring 0: (0,+3)
                const uint32_t Lw = __builtin_amdgcn_alignbyte(Cc, Cm, 1);    // bytes tc-3 .. tc   -> ring 12 of pixels 0..3
                const uint32_t Rw = __builtin_amdgcn_alignbyte(Cp, Cc, 3);    // bytes tc+3 .. tc+6 -> ring 4 of pixels 0..3
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const int v = (Cc >> (8 * j)) & 0xFF;
                    const int r0 = (Bb >> (8 * j)) & 0xFF, r8 = (Tt >> (8 * j)) & 0xFF;
                    const int r4 = (Rw >> (8 * j)) & 0xFF, r12 = (Lw >> (8 * j)) & 0xFF;
                    const int lo = v - t, hi = v + t;
                    const bool dk = (r0 < lo || r8 < lo) && (r4 < lo || r12 < lo);
                    const bool br = (r0 > hi || r8 > hi) && (r4 > hi || r12 > hi);
                    const int px = tc + j - c_first;
                    const bool hit = q < ntask && px >= 0 && px < iw && (dk || br);
                    const int slot = wave_append(hit, n_maybe);
                    if (hit) list[slot] = (uint16_t)((py << 8) | px);
                }
            }
        }
        WAVE_LDS_FENCE();

        // ---- pass 2 (maybe pixels): 16-bit darker / brighter ring masks, 9 contiguous (cyclic) set bits.  Corners are compacted
        //      in place: the slots written in an iteration lie below the entries read in it (reads precede writes in program order)
        int n_corner = 0;
        for (int i0 = 0; i0 < n_maybe; i0 += FAST_NT) {
            const int i = i0 + tid;
            const bool act = i < n_maybe;
            int pos = list[act ? i : 0];
            int py = pos >> 8, px = pos & 255;
            const uint8_t* ctr = &tile[(py + 3) * TILE_PITCH + off + px + 3];
            int v = ctr[0];
            int lo = v - t, hi = v + t;
            // one subtract + one v_alignbit per ring pixel and polarity: the sign bit of (r - lo) / (hi - r) is shifted into the mask
            // (bit order comes out reversed, which a cyclic run test does not care about)
            uint32_t mdark = 0, mbright = 0;
#pragma unroll
            for (int k = 0; k < 16; k++) {
                const int r = ctr[RO[k]];
                mdark = __builtin_amdgcn_alignbit(mdark, (uint32_t)(r - lo), 31);
                mbright = __builtin_amdgcn_alignbit(mbright, (uint32_t)(hi - r), 31);
            }
            uint32_t m = mdark | (mdark << 16);
            uint32_t x = m & (m >> 1); x &= x >> 2; x &= x >> 4; x &= m >> 8;
            uint32_t m2 = mbright | (mbright << 16);
            uint32_t y = m2 & (m2 >> 1); y &= y >> 2; y &= y >> 4; y &= m2 >> 8;
            const bool corner = act && ((x | y) & 0xFFFFu) != 0;
            WAVE_LDS_FENCE();                                    // this iteration's list reads have returned before its slots are overwritten
            const int slot = wave_append(corner, n_corner);
            if (corner) list[slot] = (uint16_t)(pos | ((y & 0xFFFFu) ? 0x8000 : 0));       // bit 15: bright corner
        }
        WAVE_LDS_FENCE();

        // ---- pass 3 (corners): corner score
        const int nc = n_corner;
        for (int i = tid; i < nc; i += FAST_NT) {
            int pos = list[i];
            const bool bright = pos & 0x8000;
            int py = (pos >> 8) & 127, px = pos & 255;
            const uint8_t* ctr = &tile[(py + 3) * TILE_PITCH + off + px + 3];
            int v = ctr[0];
            int d[16];
#pragma unroll
            for (int k = 0; k < 16; k++) { int e = v - (int)ctr[RO[k]]; d[k] = bright ? -e : e; }
            score[(py + 1) * SCORE_PITCH + px + 1] = (uint8_t)fast_corner_score(d);
        }
        WAVE_LDS_FENCE();

        // ---- pass 4: 3x3 strict NMS inside the cell; survivors go straight to this cell's slots
        const size_t slot0 = (size_t)cur.img * cand_img_stride + L.cand_off + (size_t)cur.c * cur.ccap;
        int n_out = 0;
        for (int i0 = 0; i0 < nc; i0 += FAST_NT) {
            const int i = i0 + tid;
            const bool act = i < nc;
            int pos = list[act ? i : 0];
            int py = (pos >> 8) & 127, px = pos & 255;
            const uint8_t* sc = &score[(py + 1) * SCORE_PITCH + px + 1];
            int s = sc[0];
            const bool keep = act && s > sc[1] && s > sc[-1] &&
                              s > sc[-SCORE_PITCH - 1] && s > sc[-SCORE_PITCH] && s > sc[-SCORE_PITCH + 1] &&
                              s > sc[SCORE_PITCH - 1] && s > sc[SCORE_PITCH] && s > sc[SCORE_PITCH + 1];
            const int slot = wave_append(keep, n_out);
            if (keep) {
                // coordinates relative to (minBorderX, minBorderY): x_local + j*wCell (ORBExtractor.cpp:463-464)
                cand_xy[slot0 + slot] = ((uint32_t)(py + 3 + cur.yoff) << 16) | (uint32_t)(px + 3 + cur.xoff);
                cand_sk[slot0 + slot] = ((uint32_t)s << 24) | (uint32_t)cur.c;
            }
        }
        if (tid == 0) *cnt = n_out;
        WAVE_LDS_FENCE();                                        // NMS reads of the score tile are done
        // ---- restore the all-zero score tile
        for (int i = tid; i < nc; i += FAST_NT) {
            int pos = list[i];
            score[(((pos >> 8) & 127) + 1) * SCORE_PITCH + (pos & 255) + 1] = 0;
        }
    }
}

void hs_launch_fast(const HsLevel* d_lv, int nlevels, HsImg0 img0, int batch, int total_cells, int fast_th,
                    uint32_t* cand_xy, uint32_t* cand_sk, int32_t* cell_count, uint64_t cand_img_stride,
                    int max_wcell, int max_hcell, hipStream_t s)
{
    if (total_cells <= 0) return;
    auto up = [](int v, int a) { return (v + a - 1) / a * a; };
    FastLds L;
    const bool small = max_wcell <= 37;
    L.tile_pitch = small ? 48 : 80;                        // >= 3 + max_wcell + 6, multiple of 4
    L.score_pitch = small ? 40 : HS_MAX_CELL + 4;          // >= max_wcell + 2, multiple of 4
    L.score_bytes = up((max_hcell + 2) * L.score_pitch, 4);
    int o = up((max_hcell + 6) * L.tile_pitch + 16, 16);
    L.off_score = o; o = up(o + L.score_bytes, 16);
    L.off_list = o; o = up(o + 2 * max_wcell * max_hcell, 16);
    L.total = o;
    const int total_work = total_cells * batch;
    int per_cu = std::min(4 * FAST_WAVES_PER_SIMD, std::max(1, (160 * 1024) / L.total));   // resident waves per CU (register budget) or LDS
    if (const char* e = getenv("HS_FAST_WG_PER_CU")) per_cu = std::max(1, std::min(per_cu, atoi(e)));   // tuning knob
    int nblk = 256 * per_cu;                               // persistent single-wave workgroups
    while (nblk >= 16 && (nblk / 2) % 8 == 0 && nblk / 2 >= total_work) nblk /= 2;   // tiny jobs: fewer idle workgroups; stays a multiple of 8 (XCD dealing)
    const int tile_dwords = (max_hcell + 6) * ((3 + max_wcell + 6 + 3) / 4);
    if (small && tile_dwords <= 64 * 8)
        hipLaunchKernelGGL((k_fast_cells<48, 40, 8>), dim3(nblk), dim3(FAST_NT), L.total, s, d_lv, nlevels, img0, fast_th, cand_xy, cand_sk, cell_count,
                           cand_img_stride, total_cells, total_work, L);
    else if (small)
        hipLaunchKernelGGL((k_fast_cells<48, 40, 16>), dim3(nblk), dim3(FAST_NT), L.total, s, d_lv, nlevels, img0, fast_th, cand_xy, cand_sk, cell_count,
                           cand_img_stride, total_cells, total_work, L);
    else
        hipLaunchKernelGGL((k_fast_cells<80, HS_MAX_CELL + 4, 24>), dim3(nblk), dim3(FAST_NT), L.total, s, d_lv, nlevels, img0, fast_th, cand_xy, cand_sk, cell_count,
                           cand_img_stride, total_cells, total_work, L);
}
