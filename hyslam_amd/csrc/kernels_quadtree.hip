// kernels_quadtree.hip — H4 moved onto the GPU: ORBExtractor::DistributeOctTree (src/features/ORBExtractor.cpp:179-403)
// with ExtractorNode::DivideNode (:121-177).
//
// The reference is a sequential std::list algorithm.  Its observable behaviour is restated here as a
// level-synchronous, order-preserving parallel algorithm; one 1024-thread workgroup owns one (image, level):
//
//   * The std::list is an array in list order.  Every new node is push_front'ed, so after a pass that
//     creates T children (creation order = processing order of the parent, then n1..n4) the list is
//     [children in reverse creation order] ++ [untouched nodes in their old order].
//   * Phase 1 (:246-305) splits every multi-point node, in list order.  Phase 2 (:316-377) sorts the
//     multi-point nodes created by the previous pass by (size, pointer) and splits from the back until the
//     list holds >= N nodes.  Documented deviation D1: the pointer is replaced by the creation sequence
//     number (monotone allocator); because the candidates of a phase-2 pass were all created by the previous
//     pass, "larger sequence number" == "smaller list index", so the processing order is
//     (size descending, list index ascending).  The cut "stop once size >= N" is a prefix sum of
//     (children-1) over that order.
//   * A point only ever moves from a node to one of its children, decided by the reference's comparisons
//     `x < n1.UR.x`, `y < n1.BR.y` with halfX = ceil((UR.x-UL.x)/2).  Per pass: one sweep over the points to
//     count the four children of every splittable node (LDS atomics), a block scan to lay out the new list,
//     one sweep to relabel.
//   * The kept keypoint of a node is its maximum response, first in vToDistributeKeys order on ties (:381-400):
//     a 64-bit LDS atomicMax on score<<56 | ~order, where order = (cell, y, x) reproduces the reference's
//     candidate order (cells row-major, cv::FAST's row-major scan inside a cell).
//
//   * COUNT DOMAIN (round 3).  Which node a point ends up in depends on geometry alone — its path through DivideNode's midpoints below its
//     root — and everything the list bookkeeping needs from the points is COUNTS.  The gather sweep therefore also computes every point's
//     geometric key (root, path to depth DH = 5 or 6) and a histogram of those keys; the per-depth counts are sums of it (a pyramid).  Then
//       - phase 1 in CLOSED FORM: while every multi-point node is split, the list after pass p is
//             R_p ++ finals(R_{p-1}) ++ ... ++ finals(R_0)
//         (R_d = the depth-d nodes in reverse creation order, finals = its single-point nodes), R_d orders the depth-d cells by their path
//         with every second digit complemented (push_front reverses the order once per pass), and the sizes S_p / nToExpand that decide when
//         the reference stops or switches phase are sums over the pyramid: no replay, one scan over the concatenated sequences;
//       - the size-ordered passes of phase 2 run on the list with the children's counts READ from the pyramid instead of counted by a sweep
//         over the points, and without relabelling the points;
//       - one sweep at the end maps every point's geometric key to its node.
//     A node deeper than the pyramid (clustered points), more than 8 root nodes or more than 65535 points fall back to the point-domain
//     passes (sweeps with LDS atomics), which remain the general algorithm.
//
// Inputs are the per-item candidate slots written by k_fast_rows (gathered into a dense list first; its order is irrelevant: every
// order-dependent decision uses the (cell, y, x) key of the reference's candidate order).  Outputs per (image, level):
// selected (x,y,score) in final list order + count.  Bound: LDS latency / barriers; HBM traffic negligible.
#include "hs_internal.h"
#include <cstdlib>

#define QT_T HS_QT_THREADS
// The kernel is a template over <QT_M, QT_PTS> (round 5):
//   QT_M    list capacity in LDS (>= the largest per-level quota + 8)
//   QT_PTS  points kept in LDS (a 1080p level has ~5000 candidates); more fall back to the global arrays
// <HS_QT_MAX_NODES = 2048, 6144> is the general instance: 152 KB of LDS, ONE workgroup per CU — every sweep over the points runs from LDS, which is what a
// launch of few workgroups (one per (image, level): 16 for a stereo pair) wants.  <1024, 0> keeps no points in LDS (the sweeps read them through L2)
// and fits 77 KB: TWO workgroups per CU, for launches of more than 256 workgroups (more than 16 stereo pairs per call), which used to run in
// rounds of 256.  The arrays that double as scratch (histogram pyramid, marks, key tables, gather offsets) are sized by what they must hold, not by QT_M.
#define QT_PTS_BIG 6144
#define QT_M_SMALL 1024

#define QT_HPYR 10928             // histogram pyramid entries (u16): n_ini * (4^(DH+1) - 1) / 3 <= 10922 for (n_ini <= 2, DH = 6) and (n_ini <= 8, DH = 5)

// Node list, double buffered.  Rectangles are only kept in the point domain; in the count domain their LDS holds the histogram pyramid.
template <int QT_M> struct alignas(16) QtRects { int16_t x0[QT_M], x1[QT_M], y0[QT_M], y1[QT_M]; };
constexpr int qt_cmax(int a, int b) { return a > b ? a : b; }
#define QT_MAXT 4096              // 64-px tiles of a level that the spatial order of the kept keypoints is built for (more: list order)
struct QtNodes {                 // a view of buffer `c`
    int16_t *x0, *x1, *y0, *y1; uint32_t* cnt; uint16_t* ekey;
};

// inclusive scan over the wavefront with DPP row shifts / row broadcasts (six VALU adds, no LDS crossbar round trips)
__device__ __forceinline__ int wave_scan_incl(int x)
{
    x += __builtin_amdgcn_update_dpp(0, x, 0x111, 0xF, 0xF, true);      // row_shr:1
    x += __builtin_amdgcn_update_dpp(0, x, 0x112, 0xF, 0xF, true);      // row_shr:2
    x += __builtin_amdgcn_update_dpp(0, x, 0x114, 0xF, 0xF, true);      // row_shr:4
    x += __builtin_amdgcn_update_dpp(0, x, 0x118, 0xF, 0xF, true);      // row_shr:8   -> inclusive within each row of 16
    x += __builtin_amdgcn_update_dpp(0, x, 0x142, 0xA, 0xF, false);     // row_bcast:15 into rows 1 and 3
    x += __builtin_amdgcn_update_dpp(0, x, 0x143, 0xC, 0xF, false);     // row_bcast:31 into rows 2 and 3
    return x;
}
__device__ __forceinline__ int wave_sum(int x) { return __builtin_amdgcn_readlane(wave_scan_incl(x), 63); }

// exclusive scan of one int per thread over the 1024-thread block; returns prefix, writes total.  s_wave holds TWO sets of per-wave sums
// used alternately (`flip`): a set is only rewritten two scans later, and the barrier of the scan in between orders that write after the
// last read — one barrier per scan.
__device__ __forceinline__ int block_scan_excl(int v, int* s_wave /*[2][16]*/, int& flip, int& total)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int incl = wave_scan_incl(v);
    int* const sw = s_wave + 16 * (flip & 1);
    flip ^= 1;
    if (lane == 63) sw[wave] = incl;
    __syncthreads();
    int base = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < QT_T / 64; w++) { int x = sw[w]; if (w < wave) base += x; tot += x; }
    total = tot;
    return base + incl - v;
}

// atomicAdd(&arr[key], 1) for the lanes with `valid`, one atomic per DISTINCT key of the wave: with a handful of counters (the roots,
// the first passes) thousands of points hit the same LDS address and plain atomics serialise
__device__ __forceinline__ void wave_agg_inc(uint32_t* arr, int key, bool valid)
{
    const int lane = threadIdx.x & 63;
    unsigned long long todo = __ballot(valid);
    while (todo) {
        const int leader = __ffsll((long long)todo) - 1;
        const int k0 = __shfl(key, leader, 64);
        const unsigned long long m = __ballot(valid && key == k0);
        if (lane == leader) atomicAdd(&arr[k0], (uint32_t)__popcll(m));
        todo &= ~m;
    }
}

__device__ __forceinline__ int child_of(const QtNodes N, int nd, int x, int y)
{
    // DivideNode: halfX = ceil((UR.x-UL.x)/2); n1.UR.x = UL.x+halfX; n1.BR.y = UL.y+halfY
    int mx = N.x0[nd] + ((N.x1[nd] - N.x0[nd] + 1) >> 1);
    int my = N.y0[nd] + ((N.y1[nd] - N.y0[nd] + 1) >> 1);
    return (x < mx ? 0 : 1) + (y < my ? 0 : 2);      // n1,n2,n3,n4
}

#ifdef HS_QT_PROFILE
__device__ unsigned long long g_qt_prof[128];
extern "C" void hs_debug_qt_profile(unsigned long long* out128) { (void)hipDeviceSynchronize(); (void)hipMemcpyFromSymbol(out128, HIP_SYMBOL(g_qt_prof), sizeof(unsigned long long) * 128); }
#define QT_MARK(tag) do { if (tid == 0 && level_first + blockIdx.x == 0 && blockIdx.y == 0 && qt_k < 126) { g_qt_prof[qt_k++] = __builtin_amdgcn_s_memtime(); g_qt_prof[qt_k++] = (tag); } } while (0)
#else
#define QT_MARK(tag)
#endif
template <int QT_M, int QT_PTS, bool RECT_GLOBAL>
__global__ __launch_bounds__(QT_T) void k_quadtree(const HsLevel* __restrict__ lv, int nlevels, int total_cells,
                                                   const uint2* __restrict__ cand,
                                                   const int32_t* __restrict__ cell_count, uint64_t cand_img_stride,
                                                   uint32_t* __restrict__ pts_xy_all, uint32_t* __restrict__ pts_sk_all,
                                                   uint16_t* __restrict__ pt_node_all, int32_t* __restrict__ cand_count,
                                                   uint32_t* __restrict__ sel_xys, int32_t* __restrict__ sel_count, int sel_img_stride,
                                                   uint16_t* __restrict__ sel_perm, int force_point_domain, int level_first,
                                                   uint32_t* __restrict__ qhist, unsigned long long* __restrict__ qbest, uint32_t qhist_img_stride, uint32_t qbest_img_stride,
                                                   int keep_points, uint8_t* __restrict__ rect_scratch)
{
    // point domain: node rectangles; count domain: the histogram pyramid (u16) — sized for the larger of the two.  RECT_GLOBAL (the large-list
    // instance): the rectangles live in a per-workgroup piece of global scratch instead and the LDS only holds the pyramid
    __shared__ alignas(16) uint8_t s_r1[RECT_GLOBAL ? QT_HPYR * 2 : qt_cmax(2 * (int)sizeof(QtRects<QT_M>), QT_HPYR * 2)];
    QtRects<QT_M>* const s_rect = reinterpret_cast<QtRects<QT_M>*>(s_r1);      // (as the pyramid's base address only when RECT_GLOBAL)
    QtRects<QT_M>* const rect_base = RECT_GLOBAL ? reinterpret_cast<QtRects<QT_M>*>(rect_scratch + ((size_t)blockIdx.y * nlevels + level_first + blockIdx.x) * 2 * sizeof(QtRects<QT_M>)) : s_rect;
    __shared__ uint32_t s_cnt[2][QT_M];            // points per node
    __shared__ uint16_t s_ekey[2][QT_M];           // count domain: depth << 13 | cell index at that depth (root * 4^depth + path)
    // child counts, indexed 4*rank + child; also: the marks (u16 per pyramid entry), the sort keys of phase 2 (<= 8 bytes per node), and at the
    // end the best-point slots (8 QT_M bytes) followed by the tile counters of the spatial order
    constexpr int CCOUNT_BYTES = qt_cmax(qt_cmax(16 * QT_M, QT_HPYR * 2), qt_cmax(8 * QT_M + 4 * QT_MAXT, (QT_M + 3) * 8));
    __shared__ alignas(16) uint32_t ccount[CCOUNT_BYTES / 4];
    // three per-node index arrays (idle during the gather: the geometric-key tables, 8192 + 4096 bytes, live here then)
    __shared__ alignas(16) int16_t s_idx3[qt_cmax(3 * QT_M, (8192 + 4096) / 2)];
    int16_t* const proc_rank = s_idx3;                                                  // processing rank of a node in this pass, -1 = not split
    int16_t* const order_node = s_idx3 + QT_M;                                          // rank -> node
    uint16_t* const new_index = reinterpret_cast<uint16_t*>(s_idx3 + 2 * QT_M);         // surviving node -> index in the next list
    // 4*rank+child -> index in the next list; also the gather's per-item offsets ((2 QT_T + 4) dwords), the partial ranks of phase 2 and the best-point slots
    __shared__ alignas(16) uint16_t child_index[qt_cmax(4 * QT_M, (2 * QT_T + 4) * 2)];
    __shared__ uint32_t s_pxy[QT_PTS > 0 ? QT_PTS : 1];      // the level's points (y<<16|x) and their node (count domain: their geometric key), when there
    __shared__ uint16_t s_pnode[QT_PTS > 0 ? QT_PTS : 1];    // are <= QT_PTS of them: every sweep walks the points from LDS instead of through L2
    __shared__ int s_wave[2 * (QT_T / 64)];
    __shared__ int s_misc[8];
    __shared__ uint32_t s_dcnt[8];                 // per depth: existing nodes | single-point nodes << 16

    const int tid = threadIdx.x;
    int sflip = 0;           // which set of per-wave sums the next block scan uses
#ifdef HS_QT_PROFILE
    int qt_k = 0;
#endif
    QT_MARK(0);
    const int level = level_first + blockIdx.x, img = blockIdx.y;
    const HsLevel& L = lv[level];
    const int N = L.quota;
    uint32_t* pxy = pts_xy_all + (size_t)img * cand_img_stride + L.cand_off;
    uint32_t* psk = pts_sk_all + (size_t)img * cand_img_stride + L.cand_off;
    uint16_t* pnode = pt_node_all + (size_t)img * cand_img_stride + L.cand_off;
    uint32_t* out = sel_xys + ((size_t)img * sel_img_stride + L.sel_off) * 3;
    int32_t* out_n = &sel_count[img * nlevels + level];
    auto view = [&](int c) { QtNodes v; v.x0 = rect_base[c].x0; v.x1 = rect_base[c].x1; v.y0 = rect_base[c].y0; v.y1 = rect_base[c].y1; v.cnt = s_cnt[c]; v.ekey = s_ekey[c]; return v; };

    const int nIni = L.n_ini;
    const float hX = L.hx;
    // ---- count domain set-up: pyramid depth, offsets (deepest level first so that it is 4-byte aligned for the packed atomics)
    const bool cf_geom = nIni >= 1 && nIni <= 8 && !force_point_domain;       // uniform
    const int DH = nIni <= 2 ? 6 : 5;
    uint16_t* const hist = reinterpret_cast<uint16_t*>(s_rect);
    auto hoff = [&](int d) { return nIni * (((1 << (2 * DH + 2)) - (1 << (2 * d + 2))) / 3); };     // entries of the levels deeper than d
    // geometric key of a point: root << 2 DH | its DivideNode decisions down to depth DH (:121-177, :209).  DivideNode halves x and y
    // independently, so the key is the bit-interleave of two one-dimensional cell indices: TABLES (a byte per pixel column of every root and
    // per pixel row, built by the host, copied into LDS that is idle during the gather) replace the six-step descent per point; the root comes
    // from comparisons with the first column that the reference's float division assigns to each root.
    uint8_t* const xtab = reinterpret_cast<uint8_t*>(s_idx3);             // [qt_w + 1] cell index along x within the column's root
    uint8_t* const ytab = xtab + 8192;                                   // [qt_h + 1]
    __shared__ int s_rbound[8];                                          // first x of root i (i >= 1)
    static_assert(sizeof(s_idx3) >= 8192 + 4096, "geometric-key tables");
    const bool use_tab = cf_geom && L.qt_xtab != nullptr;              // (qt_w < 8192 - 16 and qt_h < 4096 - 16: the host's condition)
    auto root_of = [&](int x) { return min((int)((float)x / hX), nIni - 1); };      // vpIniNodes[kp.pt.x/hX]
    // the tables depend on the level geometry alone: the host builds them once per configuration (hs_quadtree_build_tables: the same
    // expressions, IEEE float division and multiplication), the workgroup copies them with 16-byte loads when it needs them — i.e. when it
    // gathers the candidates (round 4: normally it does not, see `have_keys` below).  They live in LDS that is idle then (s_idx3).
    auto load_key_tables = [&]() {
        if (!use_tab) return;
        if (tid >= 1 && tid < nIni) s_rbound[tid] = L.qt_rbound[tid];
        for (int i = tid * 16; i <= L.qt_w; i += QT_T * 16) *reinterpret_cast<hs_u32x4*>(xtab + i) = hs_gload<hs_u32x4>(L.qt_xtab + i);
        for (int i = tid * 16; i <= L.qt_h; i += QT_T * 16) *reinterpret_cast<hs_u32x4*>(ytab + i) = hs_gload<hs_u32x4>(L.qt_ytab + i);
        __syncthreads();
    };
    auto spread = [](uint32_t v) { v = (v | (v << 4)) & 0x0F0Fu; v = (v | (v << 2)) & 0x3333u; v = (v | (v << 1)) & 0x5555u; return v; };   // bit i -> bit 2i
    auto geo_key = [&](int x, int y) {
        if (use_tab) {
            int r = 0;
            for (int i = 1; i < nIni; i++) r += x >= s_rbound[i];
            return (int)(((uint32_t)r << (2 * DH)) | spread(xtab[min(x, L.qt_w)]) | (spread(ytab[min(y, L.qt_h)]) << 1));
        }
        const int r = root_of(x);
        int x0 = (int16_t)(int)(hX * (float)r), x1 = (int16_t)(int)(hX * (float)(r + 1)), y0 = 0, y1 = (int16_t)L.qt_h;
        int path = 0;
        for (int d = 0; d < DH; d++) {
            const int mx = x0 + ((x1 - x0 + 1) >> 1), my = y0 + ((y1 - y0 + 1) >> 1);
            const int c = (x < mx ? 0 : 1) + (y < my ? 0 : 2);
            if (c & 1) x0 = mx; else x1 = mx;
            if (c & 2) y0 = my; else y1 = my;
            path = path * 4 + c;
        }
        return (r << (2 * DH)) | path;
    };
    // ---- round 4: the FAST kernel has left this level's key HISTOGRAM (deepest pyramid level) and the best candidate of every key in global
    //      memory (HsLevel::qt_hist_off; kernels_fast.hip): the count domain starts from them and never touches the candidates — the gather
    //      below (a third of the level-0 workgroup's time: item scan, run search, record fetch, key computation, LDS atomics) only runs when
    //      the points themselves are needed: point-domain passes (clustered corners, > 65535 points, HS_QT_POINT_DOMAIN) or the debug taps.
    //      Whatever happens, the workgroup leaves both global arrays ZERO for the next call.
    const bool have_keys = qhist != nullptr && L.qt_hist_off != 0xFFFFFFFFu;    // uniform; the level has key tables (nIni <= 8)
    const int ncell = nIni << (2 * DH);                                         // deepest cells (have_keys: nIni <= 8, so <= 8192)
    uint32_t* const ghist = have_keys ? qhist + (size_t)img * qhist_img_stride + L.qt_hist_off : nullptr;
    unsigned long long* const gbest = have_keys ? qbest + (size_t)img * qbest_img_stride + L.qt_best_off : nullptr;
    bool best_pending = have_keys;                                                     // gbest still holds this call's keys
    auto zero_gbest = [&]() {
        if (!best_pending) return;
        for (int i = tid; i < ncell / 2; i += QT_T) *reinterpret_cast<uint4*>(gbest + 2 * i) = make_uint4(0, 0, 0, 0);
        best_pending = false;
    };
    int n_pre = 0;
    if (cf_geom || have_keys) {
        uint32_t* const h32 = reinterpret_cast<uint32_t*>(s_rect);
        if (have_keys) {
            // the deepest level of the pyramid comes from global memory (16 bytes per thread and round: <= 8192 cells are ONE round) and goes back
            // to zero there; the levels above it are written in full by the pyramid build below, so nothing else needs clearing
            if (tid < 16) s_wave[tid] = 0;
            __syncthreads();
            int local = 0;
            for (int i = tid * 4; i < ncell / 2; i += QT_T * 4) {
                const uint4 v = *reinterpret_cast<const uint4*>(ghist + i);
                *reinterpret_cast<uint4*>(ghist + i) = make_uint4(0, 0, 0, 0);
                *reinterpret_cast<uint4*>(h32 + i) = v;
                local += (int)(v.x & 0xFFFFu) + (int)(v.x >> 16) + (int)(v.y & 0xFFFFu) + (int)(v.y >> 16) + (int)(v.z & 0xFFFFu) + (int)(v.z >> 16) + (int)(v.w & 0xFFFFu) + (int)(v.w >> 16);
            }
            const int wsum = wave_sum(local);
            if ((tid & 63) == 0 && wsum) atomicAdd(&s_wave[0], wsum);
            __syncthreads();
            n_pre = s_wave[0];
            __syncthreads();                                     // (s_wave is the block scans' scratch from here on)
        } else {
            for (int i = tid; i < QT_HPYR / 2; i += QT_T) h32[i] = 0;
        }
        if (tid < 8) s_dcnt[tid] = 0;
    }
    __syncthreads();
    QT_MARK(20);

    // ---- gather this level's candidates from the slot ranges the FAST kernel filled into one dense list (its order is irrelevant).
    // The FAST kernel packs the survivors of one work ITEM (a group of <= 8 cells of one cell row) from the item's first slot upwards:
    // ~20 records = two cache lines per item.  Per round of <= 1024 items: a thread per item for the count and the block-wide exclusive
    // scan, then a thread per RECORD, which finds its item by binary search over the scan and fetches the record with one 8-byte load.
    // In the count domain the same thread computes the point's geometric key and adds it to the histogram.
    int n = 0;
    bool gathered = false;
    auto gather = [&](const bool add_hist) {
        load_key_tables();
        n = 0;
        uint32_t* const s_pre = reinterpret_cast<uint32_t*>(child_index);        // [round items + 1] exclusive offsets of the round's items
        uint32_t* const s_src0 = s_pre + QT_T + 4;                               // [round items] first slot of the item minus its offset: record e sits at s_src0[item] + e
        static_assert(sizeof(child_index) >= (2 * QT_T + 4) * 4, "s_pre / s_src0 scratch");
        const int nitem = L.nrows * L.ngroups;
        const int ccap = hs_cell_cap(L.wcell, L.hcell);
        const int32_t* ccnt = cell_count + (size_t)img * total_cells + L.cell_begin;
        const uint2* src = cand + (size_t)img * cand_img_stride + L.cand_off;
        bool keys_to_global = true;                                // false once it is certain that all points fit the LDS copy (one round, <= QT_PTS records)
        auto put = [&](int pos, uint32_t key, uint32_t sk) {
            pxy[pos] = key; psk[pos] = sk;
            if (pos < QT_PTS) s_pxy[pos] = key;
            if (cf_geom) {
                const int gk = geo_key(key & 0xFFFF, key >> 16);
                if (pos < QT_PTS) s_pnode[pos] = (uint16_t)gk;
                if (keys_to_global) pnode[pos] = (uint16_t)gk;
                if (add_hist) atomicAdd(reinterpret_cast<uint32_t*>(s_rect) + (gk >> 1), 1u << ((gk & 1) * 16));
            }
        };
        for (int i0 = 0; i0 < nitem; i0 += QT_T) {
            const int ni = min(QT_T, nitem - i0);
            int k = 0, c0 = 0;
            if (tid < ni) {                                        // item -> its first cell, its capacity
                const int it = i0 + tid, ci = it / L.ngroups, gj = it - ci * L.ngroups;
                c0 = ci * L.ncols + gj * L.grp_cells;
                k = min(max(ccnt[c0], 0), min(L.grp_cells, L.ncols - gj * L.grp_cells) * ccap);
            }
            int tot; const int pre = block_scan_excl(k, s_wave, sflip, tot);
            if (tid < ni) { s_pre[tid] = (uint32_t)pre; s_src0[tid] = (uint32_t)(c0 * ccap - pre); }      // (no division per record later)
            if (tid == 0) s_pre[ni] = (uint32_t)tot;
            if (nitem <= QT_T && tot <= QT_PTS) keys_to_global = false;
            __syncthreads();
            QT_MARK(22);
            // Every wave takes a CONTIGUOUS run of the round's records, 64 at a time: the item of a run's first record comes from one
            // wave-uniform binary search, after that the item index only moves forward — a block of the next eight offsets is read at
            // wave-uniform addresses (LDS broadcast, no bank conflicts) and every lane counts how many of them its record has passed.
            // (A per-record binary search cost 10 dependent, scattered LDS reads per record: 6 600 cycles per 4 records.)
            {
                const int lane = tid & 63, wave = tid >> 6;
                const int per_wave = ((tot + QT_T - 1) / QT_T) * 64;                  // a multiple of 64
                const int e_begin = wave * per_wave, e_end = min(tot, e_begin + per_wave);
                if (e_begin < e_end) {
                    int cur = 0;
                    { int lo = 0, hi = ni; for (int step = 0; step < 10; step++) { const int mid = (lo + hi) >> 1; if (hi - lo > 1) { if ((int)s_pre[mid] <= e_begin) lo = mid; else hi = mid; } } cur = lo; }
                    for (int e0 = e_begin; e0 < e_end; e0 += 4 * 64) {               // four records per lane in flight
                        int item[4];
#pragma unroll
                        for (int u = 0; u < 4; u++) {
                            const int e = e0 + 64 * u + lane;
                            const bool on = e0 + 64 * u < e_end;                     // wave-uniform
                            const int e_last = min(e0 + 64 * u + 63, e_end - 1);
                            int it = cur;
                            if (on) {
                                for (int base = cur;; base += 8) {
                                    int c = 0, c_last = 0;
#pragma unroll
                                    for (int j = 1; j <= 8; j++) { const int v = (int)s_pre[min(base + j, ni)]; c += v <= e; c_last += v <= e_last; }
                                    it += c;
                                    if (c_last < 8) { cur = base + c_last; break; }
                                }
                            }
                            item[u] = it;
                        }
                        QT_MARK(23);
                        uint2 rec[4];
#pragma unroll
                        for (int u = 0; u < 4; u++) {
                            const int e = e0 + 64 * u + lane;
                            if (e < e_end) rec[u] = src[s_src0[item[u]] + (uint32_t)e];
                        }
#pragma unroll
                        for (int u = 0; u < 4; u++) { const int e = e0 + 64 * u + lane; if (e < e_end) put(n + e, rec[u].x, rec[u].y); }
                        QT_MARK(25);
                    }
                }
            }
            n += tot;
            __syncthreads();                                       // the scratch is reused by the next round
        }
        gathered = true;
    };
    if (!have_keys) gather(cf_geom);
    else {
        n = n_pre;
        // the points themselves are needed from the start: no count domain at all (HS_QT_POINT_DOMAIN, > 65535 points) or the debug taps
        if (!cf_geom || n > 65535 || keep_points) { if (n > 0) gather(false); zero_gbest(); }
    }
    if (tid == 0) cand_count[img * nlevels + level] = n;
    QT_MARK(1);
    const bool in_lds = QT_PTS > 0 && n <= QT_PTS; // uniform
    auto ld_xy = [&](int p) -> uint32_t { return in_lds ? s_pxy[p] : pxy[p]; };
    auto st_node = [&](int p, int v) { if (in_lds) s_pnode[p] = (uint16_t)v; else pnode[p] = (uint16_t)v; };

    if (n == 0 || nIni < 1 || nIni > QT_M / 4) { if (tid == 0) *out_n = 0; return; }

    int cur = 0, S = 0;
    bool phase2 = false;     // uniform across the block
    int T_prev = 0;          // number of children created by the previous pass (they sit at list indices [0,T_prev))
    bool finished = false;
    bool cm = cf_geom && n <= 65535;               // count domain: the pyramid's 16-bit counters cannot overflow

    // every point's node from its geometric key: the one list node on the point's root-to-leaf chain.  marks = (depth, cell) -> list index.
    uint16_t* const mark = reinterpret_cast<uint16_t*>(ccount);
    static_assert(sizeof(ccount) >= QT_HPYR * 2 && QT_HPYR % 8 == 0, "marks");
    int mark_off[7];
#pragma unroll
    for (int d = 0; d < 7; d++) mark_off[d] = hoff(min(d, DH));
    auto build_marks = [&](const QtNodes C) {
        const int pyr = hoff(-1);
        for (int i = tid; i < (pyr + 7) / 8; i += QT_T) reinterpret_cast<uint4*>(mark)[i] = make_uint4(~0u, ~0u, ~0u, ~0u);
        __syncthreads();
        for (int i = tid; i < S; i += QT_T) { const int ek = C.ekey[i]; mark[hoff(ek >> 13) + (ek & 0x1FFF)] = (uint16_t)i; }
        __syncthreads();
    };
    auto node_of_key = [&](int gk) {
        int m[7];
#pragma unroll
        for (int d = 0; d < 7; d++) m[d] = mark[mark_off[d] + (gk >> (2 * (DH - min(d, DH))))];      // independent reads; exactly one of them is a node
        int e = m[0];
#pragma unroll
        for (int d = 1; d < 7; d++) if (m[d] != 0xFFFF) e = m[d];
        return e;
    };
    auto relabel_from_keys = [&](const QtNodes C) {
        build_marks(C);
        if (in_lds) {
            if constexpr (QT_PTS > 0) {
            int gk[QT_PTS / QT_T];
#pragma unroll
            for (int k = 0; k < QT_PTS / QT_T; k++) { const int p = tid + k * QT_T; gk[k] = p < n ? (int)s_pnode[p] : 0; }
#pragma unroll
            for (int k = 0; k < QT_PTS / QT_T; k++) { const int p = tid + k * QT_T; if (p < n) s_pnode[p] = (uint16_t)node_of_key(gk[k]); }
            }
        } else {
            for (int p0 = tid; p0 < n; p0 += 8 * QT_T) {              // points in global memory: eight loads in flight per thread
                int gk[8];
#pragma unroll
                for (int k = 0; k < 8; k++) { const int p = p0 + k * QT_T; gk[k] = p < n ? (int)pnode[p] : 0; }
#pragma unroll
                for (int k = 0; k < 8; k++) { const int p = p0 + k * QT_T; if (p < n) pnode[p] = (uint16_t)node_of_key(gk[k]); }
            }
        }
        __syncthreads();
    };

    if (cm) {
        // ---- the pyramid: counts per cell of every depth; on the way, per depth, how many cells hold a point (nz) and how many hold exactly
        //      one: a cell of depth d is a NODE of pass d's list iff it holds a point and its parent was split (held more than one), so
        //          nodes_d = nz_d - one_{d-1},   single-point nodes_d = one_d - one_{d-1}
        //      (a parent with one point has exactly one occupied child, with one point)
        auto stat = [](uint32_t c) { return (c > 0 ? 1u : 0u) + (c == 1 ? 0x10000u : 0u); };
        // Two depths per block-wide step (round 4: every step costs a barrier + two wave sums whatever its work, and there were DH of them): a thread takes
        // eight consecutive cells of depth d + 1 — two parents at depth d, half a grandparent at depth d - 1 —, writes both parents' sums with one store and
        // completes the grandparent with its neighbour lane's half.  Statistics of the child and of the parent depth; the grandparents' follow in the next step.
        int d = DH - 1;
        for (; d >= 1; d -= 2) {
            const int nhalf = nIni << (2 * d - 1), oc = hoff(d + 1), od = hoff(d), og = hoff(d - 1);      // half grandparents = pairs of parents (<= 1024)
            uint32_t accC = 0, accP = 0;
            if ((tid & ~63) < nhalf) {                                                        // waves with work (uniform per wave)
                const bool on = tid < nhalf;
                uint32_t half = 0;
                if (on) {
                    const uint4 q = *reinterpret_cast<const uint4*>(&hist[oc + 8 * tid]);     // eight u16 children, 16-byte aligned
                    const uint32_t c0 = q.x & 0xFFFF, c1 = q.x >> 16, c2 = q.y & 0xFFFF, c3 = q.y >> 16, c4 = q.z & 0xFFFF, c5 = q.z >> 16, c6 = q.w & 0xFFFF, c7 = q.w >> 16;
                    const uint32_t p0 = c0 + c1 + c2 + c3, p1 = c4 + c5 + c6 + c7;
                    *reinterpret_cast<uint32_t*>(&hist[od + 2 * tid]) = p0 | (p1 << 16);
                    accC = stat(c0) + stat(c1) + stat(c2) + stat(c3) + stat(c4) + stat(c5) + stat(c6) + stat(c7);
                    accP = stat(p0) + stat(p1);
                    half = p0 + p1;
                }
                const uint32_t other = (uint32_t)__shfl_xor((int)half, 1, 64);                // (nhalf is even: a lane's partner is on exactly when the lane is)
                if (on && !(tid & 1)) hist[og + (tid >> 1)] = (uint16_t)(half + other);
                const uint32_t totC = (uint32_t)wave_sum((int)accC), totP = (uint32_t)wave_sum((int)accP);
                if ((tid & 63) == 0) { if (totC) atomicAdd(&s_dcnt[d + 1], totC); if (totP) atomicAdd(&s_dcnt[d], totP); }
            }
            __syncthreads();
        }
        if (d == 0) {                                                                         // an odd number of depths: the roots from depth 1
            uint32_t acc = 0;
            if (tid < nIni) {
                const uint2 q = *reinterpret_cast<const uint2*>(&hist[hoff(1) + 4 * tid]);
                const uint32_t c0 = q.x & 0xFFFF, c1 = q.x >> 16, c2 = q.y & 0xFFFF, c3 = q.y >> 16;
                hist[hoff(0) + tid] = (uint16_t)(c0 + c1 + c2 + c3);
                acc = stat(c0) + stat(c1) + stat(c2) + stat(c3);
            }
            if (tid < 64) { const uint32_t tot = (uint32_t)wave_sum((int)acc); if (tid == 0 && tot) atomicAdd(&s_dcnt[1], tot); }
            __syncthreads();
        }
        if (tid < 64) {
            const uint32_t acc = (uint32_t)wave_sum((int)(tid < nIni ? stat(hist[hoff(0) + tid]) : 0u));
            if (tid == 0) s_dcnt[0] = acc;
        }
        __syncthreads();
        QT_MARK(30);
        // ---- phase 1 in closed form (:246-313): sizes after every pass from the per-depth sums
        int P = 0;
        {
            int A[8], F[8];
#pragma unroll
            for (int d = 0; d < 8; d++) {
                const uint32_t v = d <= DH ? s_dcnt[d] : 0u, u = (d >= 1 && d <= DH) ? s_dcnt[d - 1] : 0u;
                A[d] = (int)(v & 0xFFFF) - (int)(u >> 16); F[d] = (int)(v >> 16) - (int)(u >> 16);
            }
            S = A[0];
            int fsum = 0, prevS = A[0];
#pragma unroll
            for (int p = 1; p <= 6; p++) {
                if (p <= DH && !finished && !phase2 && P == p - 1) {
                    if (A[p - 1] - F[p - 1] == 0) finished = true;                          // no multi-point node: the pass changes nothing (:309)
                    else {
                        fsum += F[p - 1];
                        S = A[p] + fsum; P = p; T_prev = A[p];
                        if (S >= N || S == prevS) finished = true;
                        else if (S + 3 * (A[p] - F[p]) > N) phase2 = true;
                        prevS = S;
                    }
                }
            }
        }
        QT_MARK(31);
        if (S > QT_M) { zero_gbest(); if (tid == 0) *out_n = 0; return; }                      // cannot happen for quota + 8 <= QT_M (host checks)
        // ---- the list after pass P: R_P ++ finals(R_{P-1}) ++ ... ++ finals(R_0) by ONE scan over the concatenated cell sequences; R_d walks
        //      the depth-d cells with the root order reversed when d is odd and the digits at even distance from the last one complemented.
        //      Depths >= 2: a thread takes 16 consecutive elements = one aligned block of 16 cells (element k <-> cell k ^ 3 of the block),
        //      read with two 16-byte loads + the four parents; depths 1 and 0: a thread per element.
        {
            const QtNodes C = view(cur);
            // which piece of the concatenation is this thread's?
            int d = -1, j = 0, single = 0;                                                    // depth, block (or element) index inside the segment
            {
                int t = tid;
                for (int dd = P; dd >= 0 && d < 0; dd--) {
                    const int nb = nIni << (2 * dd), units = dd >= 2 ? nb >> 4 : nb;
                    if (t < units) { d = dd; j = t; single = dd < 2; }
                    else t -= units;
                }
            }
            uint32_t w[8] = { 0, 0, 0, 0, 0, 0, 0, 0 };                                       // the block's 16 counts (u16 pairs)
            uint32_t flags = 0;                                                               // bit k: element k of this thread's piece is in the list
            int bblock = 0;
            if (d >= 2) {
                const int sh = 2 * d, i0 = j << 4;
                const int rr = i0 >> sh, r = (d & 1) ? nIni - 1 - rr : rr;
                bblock = ((r << sh) | ((i0 & ((1 << sh) - 1)) ^ (0x330 & ((1 << sh) - 1)))) >> 4;
                const uint4 qa = *reinterpret_cast<const uint4*>(&hist[hoff(d) + 16 * bblock]), qb = *reinterpret_cast<const uint4*>(&hist[hoff(d) + 16 * bblock + 8]);
                const uint2 qp = *reinterpret_cast<const uint2*>(&hist[hoff(d - 1) + 4 * bblock]);
                w[0] = qa.x; w[1] = qa.y; w[2] = qa.z; w[3] = qa.w; w[4] = qb.x; w[5] = qb.y; w[6] = qb.z; w[7] = qb.w;
                const uint32_t par[4] = { qp.x & 0xFFFF, qp.x >> 16, qp.y & 0xFFFF, qp.y >> 16 };
#pragma unroll
                for (int k = 0; k < 16; k++) {
                    const int kk = k ^ 3;
                    const uint32_t c = (w[kk >> 1] >> (16 * (kk & 1))) & 0xFFFF;
                    if (c > 0 && par[kk >> 2] > 1 && (d == P || c == 1)) flags |= 1u << k;
                }
            } else if (d >= 0) {
                const int sh = 2 * d;
                const int rr = j >> sh, r = (d & 1) ? nIni - 1 - rr : rr;
                bblock = (r << sh) | ((j & ((1 << sh) - 1)) ^ (0x3 & ((1 << sh) - 1)));      // the cell itself
                const uint32_t c = hist[hoff(d) + bblock];
                const bool ex = c > 0 && (d == 0 || hist[hoff(0) + (bblock >> 2)] > 1);
                w[0] = c;
                if (ex && (d == P || c == 1)) flags = 1u;
            }
            int tot; int pre = block_scan_excl(__popc(flags), s_wave, sflip, tot);
            if (single) { if (flags) { C.ekey[pre] = (uint16_t)((d << 13) | bblock); C.cnt[pre] = w[0]; } }
            else if (flags) {
#pragma unroll
                for (int k = 0; k < 16; k++) {
                    if (flags & (1u << k)) {
                        const int kk = k ^ 3;
                        C.ekey[pre] = (uint16_t)((d << 13) | (16 * bblock + kk)); C.cnt[pre] = (w[kk >> 1] >> (16 * (kk & 1))) & 0xFFFF; pre++;
                    }
                }
            }
            __syncthreads();
        }
    } else {
        // ---- point domain from the start: roots (:183-225): count, drop empty ones, keep list order = root order
        const QtNodes C = view(0);
        for (int i = tid; i < nIni; i += QT_T) ccount[i] = 0;
        __syncthreads();
        for (int p = tid; p < n; p += QT_T) {
            int x = ld_xy(p) & 0xFFFF;
            int r = (int)((float)x / hX);               // vpIniNodes[kp.pt.x/hX]
            r = min(r, nIni - 1);
            wave_agg_inc(ccount, r, true);
        }
        __syncthreads();
        if (tid == 0) {
            int Sr = 0;
            for (int i = 0; i < nIni; i++) {
                if (ccount[i] > 0) {
                    C.x0[Sr] = (int16_t)(int)(hX * (float)i);
                    C.x1[Sr] = (int16_t)(int)(hX * (float)(i + 1));
                    C.y0[Sr] = 0; C.y1[Sr] = (int16_t)L.qt_h;
                    C.cnt[Sr] = ccount[i];
                    new_index[i] = (uint16_t)Sr;
                    Sr++;
                }
            }
            s_misc[0] = Sr;
        }
        __syncthreads();
        for (int p = tid; p < n; p += QT_T) {
            int x = ld_xy(p) & 0xFFFF;
            int r = min((int)((float)x / hX), nIni - 1);
            st_node(p, new_index[r]);
        }
        S = s_misc[0];
        __syncthreads();
    }
    QT_MARK(5);
    // ---- main loop
    for (int iter = 0; iter < 64 && !finished; iter++) {
        // leave the count domain when a node that may be split has no children in the pyramid: label the points, build the rectangles
        if (cm && (!phase2 || (T_prev > 0 && (int)(s_ekey[cur][0] >> 13) >= DH))) {
            const QtNodes C = view(cur);
            if (have_keys && !gathered) { gather(false); zero_gbest(); }      // the point-domain passes need the points after all: fetch them now (keys into s_pnode / pnode)
            relabel_from_keys(C);                       // the pyramid is dead from here on: its LDS becomes the rectangles
            for (int i = tid; i < S; i += QT_T) {
                const int ek = C.ekey[i], d = ek >> 13, g = ek & 0x1FFF, r = g >> (2 * d);
                int x0 = (int16_t)(int)(hX * (float)r), x1 = (int16_t)(int)(hX * (float)(r + 1)), y0 = 0, y1 = (int16_t)L.qt_h;
                for (int j = d - 1; j >= 0; j--) {
                    const int c = (g >> (2 * j)) & 3;
                    const int mx = x0 + ((x1 - x0 + 1) >> 1), my = y0 + ((y1 - y0 + 1) >> 1);
                    if (c & 1) x0 = mx; else x1 = mx;
                    if (c & 2) y0 = my; else y1 = my;
                }
                C.x0[i] = (int16_t)x0; C.x1[i] = (int16_t)x1; C.y0[i] = (int16_t)y0; C.y1[i] = (int16_t)y1;
            }
            __syncthreads();
            cm = false;
        }
        const QtNodes C = view(cur);
        const QtNodes X = view(cur ^ 1);
        const int prevSize = S;
        int E;               // number of nodes considered for splitting this pass

        // -- choose processing order
        if (!phase2) {
            // every multi-point node, in list order (:246-305)
            int flags[(QT_M + QT_T - 1) / QT_T];
            int local = 0;
#pragma unroll
            for (int k = 0; k < (QT_M + QT_T - 1) / QT_T; k++) {
                int i = tid * ((QT_M + QT_T - 1) / QT_T) + k;
                flags[k] = (i < S && C.cnt[i] > 1) ? 1 : 0;
                local += flags[k];
            }
            int tot; int pre = block_scan_excl(local, s_wave, sflip, tot);
#pragma unroll
            for (int k = 0; k < (QT_M + QT_T - 1) / QT_T; k++) {
                int i = tid * ((QT_M + QT_T - 1) / QT_T) + k;
                if (i < S) {
                    if (flags[k]) { proc_rank[i] = (int16_t)pre; order_node[pre] = (int16_t)i; pre++; }
                    else proc_rank[i] = -1;
                }
            }
            E = tot;
        } else {
            // multi-point nodes created by the previous pass, sorted by (size desc, list index asc) (:321-325, D1): rank = number of candidates
            // that go first.  All 1024 threads: candidate i = t % W is compared against the G-th part of the list by thread t, the partial
            // ranks meet in LDS (the child-index array is free here).
            uint32_t* const rank_acc = reinterpret_cast<uint32_t*>(child_index);      // [T_prev]
            // Sort key: count << 16 | ~index — j goes before i <=> key[j] > key[i].  32-bit keys (four per LDS read) while no node can hold more than
            // 65 535 points (n <= 65 535: always in the count domain); a level with more candidates than that (saturated frames: a 4 Mpx checkerboard has
            // 260 000 on one level) ranks with 64-bit keys — the counts used to be CLAMPED to 16 bits there, so two nodes above 65 535 points tied and
            // went by list index instead of by size (found by the long fuzz campaign of round 4: 5 of 59 keypoints differed)
            for (int i = tid; i < S; i += QT_T) proc_rank[i] = -1;
            auto rank_candidates = [&](auto key_zero) {
                using K = decltype(key_zero);
                K* const okey = reinterpret_cast<K*>(ccount);                          // [T_prev + 3]
                static_assert(sizeof(ccount) >= (QT_M + 3) * sizeof(unsigned long long), "sort keys");
                for (int i = tid; i < T_prev + 3; i += QT_T) {
                    if (i < T_prev) { rank_acc[i] = 0; okey[i] = ((K)C.cnt[i] << 16) | (K)(0xFFFFu - (uint32_t)i); }
                    else okey[i] = 0;                                                  // padding of the last group of four
                }
                if (tid == 0) s_misc[2] = 0;
                __syncthreads();
                const int W = T_prev, G = (W > 0 && W <= QT_T / 2) ? QT_T / W : 1;    // G threads per candidate
                const int part = (((W + G - 1) / G) + 3) & ~3;                         // list entries per thread, a multiple of 4
                int mine = 0;
                for (int i0 = 0; i0 < W; i0 += QT_T) {
                    const int i = G > 1 ? tid % W : i0 + tid, g = G > 1 ? tid / W : 0;
                    if (i < W && g < G) {
                        const K ki = okey[i];
                        if (ki >= ((K)2 << 16)) {                                      // a multi-point node
                            const int j0 = g * part, j1 = min((W + 3) & ~3, j0 + part);
                            int r = 0;
                            if constexpr (sizeof(K) == 4) {
                                const uint4* k4 = reinterpret_cast<const uint4*>(okey);   // four keys per LDS read
                                for (int j = j0; j < j1; j += 4) {
                                    const uint4 q = k4[j >> 2];
                                    r += (q.x > ki) + (q.y > ki) + (q.z > ki) + (q.w > ki);
                                }
                            } else {
                                for (int j = j0; j < j1; j++) r += okey[j] > ki;
                            }
                            if (G > 1) { if (r) atomicAdd(&rank_acc[i], (uint32_t)r); } else rank_acc[i] = (uint32_t)r;
                            mine += g == 0;
                        }
                    }
                    if (G > 1) break;
                }
                {   // number of candidates
                    const unsigned long long bal = __ballot(mine > 0);
                    int cntw = mine;
                    if (W > QT_T) {
#pragma unroll
                        for (int o = 32; o > 0; o >>= 1) cntw += __shfl_xor(cntw, o, 64);
                    } else cntw = (int)__popcll(bal);
                    if ((tid & 63) == 0 && cntw) atomicAdd(&s_misc[2], cntw);
                }
                __syncthreads();
            };
            if (n <= 65535) rank_candidates((uint32_t)0); else rank_candidates((unsigned long long)0);
            for (int i = tid; i < T_prev; i += QT_T)
                if (C.cnt[i] > 1) { const int r = (int)rank_acc[i]; proc_rank[i] = (int16_t)r; order_node[r] = (int16_t)i; }
            E = s_misc[2];
        }
        __syncthreads();
        QT_MARK(10 + (phase2 ? 100 : 0));
        if (E == 0) break;                          // nothing can be split: size == prevSize (:309,374)

        // -- count the four children of every candidate node
        if (cm) {                                   // count domain: read them from the pyramid
            for (int i = tid; i < 4 * E; i += QT_T) {
                const int ek = C.ekey[order_node[i >> 2]], d = ek >> 13, g = ek & 0x1FFF;
                ccount[i] = hist[hoff(d + 1) + 4 * g + (i & 3)];
            }
        } else {
        for (int i = tid; i < 4 * E; i += QT_T) ccount[i] = 0;
        __syncthreads();
        if (in_lds) {
            if constexpr (QT_PTS > 0) {
            // all of a thread's points at once: independent LDS loads instead of one dependent chain per point
            int nd[QT_PTS / QT_T], rk[QT_PTS / QT_T];
#pragma unroll
            for (int k = 0; k < QT_PTS / QT_T; k++) { const int p = tid + k * QT_T; nd[k] = p < n ? (int)s_pnode[p] : 0; }
#pragma unroll
            for (int k = 0; k < QT_PTS / QT_T; k++) { const int p = tid + k * QT_T; rk[k] = p < n ? (int)proc_rank[nd[k]] : -1; }
#pragma unroll
            for (int k = 0; k < QT_PTS / QT_T; k++) {
                const int p = tid + k * QT_T;
                const bool on = rk[k] >= 0;
                const uint32_t xy = on ? s_pxy[p] : 0u;
                const int key = on ? 4 * rk[k] + child_of(C, nd[k], xy & 0xFFFF, xy >> 16) : 0;
                if (E <= 4) wave_agg_inc(ccount, key, on);           // <= 16 counters: aggregate per wave
                else if (on) atomicAdd(&ccount[key], 1u);
            }
            }
        } else {
            for (int p = tid; p < n; p += QT_T) {
                int nd = pnode[p];
                int r = proc_rank[nd];
                if (r >= 0) {
                    uint32_t xy = pxy[p];
                    atomicAdd(&ccount[4 * r + child_of(C, nd, xy & 0xFFFF, xy >> 16)], 1u);
                }
            }
        }
        }
        __syncthreads();

        QT_MARK(11);
        // -- how many of them are split this pass, and where their children go: ONE scan over the candidates in processing order of
        //    (occupied children | children with more than one point << 16).  Phase 2 stops at the first k whose split takes the list to >= N
        //    nodes: size after the first k splits = S + sum_{r<k} (children_r - 1).  Children sit at the front of the next list in reverse
        //    creation order.
        int P = E, T, nToExpand;
        {
            constexpr int RPT = (QT_M + QT_T - 1) / QT_T;
            uint32_t nc[RPT], local = 0;
            uint4 cc4[RPT];
#pragma unroll
            for (int k = 0; k < RPT; k++) {
                const int r = tid * RPT + k;
                cc4[k] = r < E ? *reinterpret_cast<const uint4*>(&ccount[4 * r]) : make_uint4(0, 0, 0, 0);
                nc[k] = (cc4[k].x > 0) + (cc4[k].y > 0) + (cc4[k].z > 0) + (cc4[k].w > 0) + (((cc4[k].x > 1) + (cc4[k].y > 1) + (cc4[k].z > 1) + (cc4[k].w > 1)) << 16);
                local += nc[k];
            }
            int tot; const int pre0 = block_scan_excl((int)local, s_wave, sflip, tot);
            if (tid == 0) { s_misc[1] = E; s_misc[3] = tot; }
            __syncthreads();
            if (phase2) {
                uint32_t pre = (uint32_t)pre0;
#pragma unroll
                for (int k = 0; k < RPT; k++) {
                    const int r = tid * RPT + k;
                    if (r < E) {
                        const int before = S + (int)(pre & 0xFFFF) - r, after = before + (int)(nc[k] & 0xFFFF) - 1;
                        pre += nc[k];
                        if (before < N && after >= N) { s_misc[1] = r + 1; s_misc[3] = (int)pre; }      // unique r: the size never decreases
                    }
                }
                __syncthreads();
            }
            P = s_misc[1];
            T = s_misc[3] & 0xFFFF; nToExpand = s_misc[3] >> 16;
            uint32_t pre = (uint32_t)pre0;
#pragma unroll
            for (int k = 0; k < RPT; k++) {
                const int r = tid * RPT + k;
                if (r < P) {
                    const int nd = order_node[r];
                    int ci = (int)(pre & 0xFFFF);
                    const uint32_t cv[4] = { cc4[k].x, cc4[k].y, cc4[k].z, cc4[k].w };
                    int ek = 0, x0 = 0, x1 = 0, y0 = 0, y1 = 0, mx = 0, my = 0;
                    if (cm) ek = C.ekey[nd];
                    else { x0 = C.x0[nd]; x1 = C.x1[nd]; y0 = C.y0[nd]; y1 = C.y1[nd]; mx = x0 + ((x1 - x0 + 1) >> 1); my = y0 + ((y1 - y0 + 1) >> 1); }
#pragma unroll
                    for (int c = 0; c < 4; c++) {
                        if (cv[c] > 0) {
                            const int pos = T - 1 - ci; ci++;
                            child_index[4 * r + c] = (uint16_t)pos;
                            if (pos < QT_M) {
                                X.cnt[pos] = cv[c];
                                if (cm) X.ekey[pos] = (uint16_t)((((ek >> 13) + 1) << 13) | (4 * (ek & 0x1FFF) + c));
                                else {
                                    X.x0[pos] = (int16_t)((c & 1) ? mx : x0); X.x1[pos] = (int16_t)((c & 1) ? x1 : mx);
                                    X.y0[pos] = (int16_t)((c & 2) ? my : y0); X.y1[pos] = (int16_t)((c & 2) ? y1 : my);
                                }
                            }
                        }
                    }
                }
                pre += nc[k];
            }
        }
        QT_MARK(12);
        int nsurv;
        {
            int local = 0; int fl[(QT_M + QT_T - 1) / QT_T];
#pragma unroll
            for (int k = 0; k < (QT_M + QT_T - 1) / QT_T; k++) {
                int i = tid * ((QT_M + QT_T - 1) / QT_T) + k;
                int r = (i < S) ? proc_rank[i] : 0;
                fl[k] = (i < S) && !(r >= 0 && r < P);
                local += fl[k];
            }
            int pre = block_scan_excl(local, s_wave, sflip, nsurv);
#pragma unroll
            for (int k = 0; k < (QT_M + QT_T - 1) / QT_T; k++) {
                int i = tid * ((QT_M + QT_T - 1) / QT_T) + k;
                if (fl[k]) {
                    int pos = T + pre; pre++;
                    new_index[i] = (uint16_t)pos;
                    if (pos < QT_M) {
                        X.cnt[pos] = C.cnt[i];
                        if (cm) X.ekey[pos] = C.ekey[i];
                        else { X.x0[pos] = C.x0[i]; X.x1[pos] = C.x1[i]; X.y0[pos] = C.y0[i]; X.y1[pos] = C.y1[i]; }
                    }
                }
            }
        }
        __syncthreads();
        if (T + nsurv > QT_M) { zero_gbest(); if (tid == 0) *out_n = 0; return; }     // cannot happen for quota+8 <= QT_M (host checks)

        QT_MARK(13);
        // -- relabel the points (point domain; in the count domain the points keep their geometric keys until the end)
        if (cm) {
        } else if (in_lds) {
            if constexpr (QT_PTS > 0) {
            int nd[QT_PTS / QT_T], rk[QT_PTS / QT_T];
#pragma unroll
            for (int k = 0; k < QT_PTS / QT_T; k++) { const int p = tid + k * QT_T; nd[k] = p < n ? (int)s_pnode[p] : 0; }
#pragma unroll
            for (int k = 0; k < QT_PTS / QT_T; k++) rk[k] = proc_rank[nd[k]];
#pragma unroll
            for (int k = 0; k < QT_PTS / QT_T; k++) {
                const int p = tid + k * QT_T;
                if (p < n) {
                    if (rk[k] >= 0 && rk[k] < P) {
                        const uint32_t xy = s_pxy[p];
                        s_pnode[p] = child_index[4 * rk[k] + child_of(C, nd[k], xy & 0xFFFF, xy >> 16)];
                    } else s_pnode[p] = new_index[nd[k]];
                }
            }
            }
        } else {
            for (int p = tid; p < n; p += QT_T) {
                int nd = pnode[p];
                int r = proc_rank[nd];
                if (r >= 0 && r < P) {
                    uint32_t xy = pxy[p];
                    pnode[p] = child_index[4 * r + child_of(C, nd, xy & 0xFFFF, xy >> 16)];
                } else pnode[p] = new_index[nd];
            }
        }
        __syncthreads();
        QT_MARK(14);
        S = T + nsurv;
        cur ^= 1;
        T_prev = T;

        // -- termination (:307-313, :370-375)
        if (S >= N || S == prevSize) break;
        if (!phase2 && (S + nToExpand * 3) > N) phase2 = true;
    }
    // ---- keep the best point of every node (:381-400), emit in list order.  When the distribution ended in the count domain the points still
    //      carry their geometric keys: the node is looked up on the fly (marks in the child-count array, the best-point slots in the
    //      child-index array), one sweep instead of two.
    const bool keyed = cm;                                           // uniform
    if (keyed) build_marks(view(cur));
    QT_MARK(3);
    unsigned long long* best = reinterpret_cast<unsigned long long*>(keyed ? reinterpret_cast<uint32_t*>(child_index) : ccount);      // QT_M * 8 bytes either way
    static_assert(sizeof(child_index) >= QT_M * 8, "best-point slots");
    for (int i = tid; i < S; i += QT_T) best[i] = 0ull;
    __syncthreads();
    auto offer = [&](uint32_t xy, uint32_t sk, int node) {
        unsigned long long order = ((unsigned long long)(sk & 0xFFFFFFu) << 32) | xy;          // (cell, y, x)
        unsigned long long key = ((unsigned long long)(sk >> 24) << 56) | (0x00FFFFFFFFFFFFFFull - order);
        atomicMax(&best[node], key);
    };
    if (keyed && have_keys && !gathered) {
        // the candidates were never fetched: every occupied deepest cell offers the best candidate the FAST kernel recorded for it (the same
        // 64-bit key a sweep over the cell's points would end with) to the list node on the cell's root-to-leaf chain; the global slots go back to zero
        // The eight consecutive cells of a thread share their ancestors down to depth DH - 2 (one mark lookup per depth for all of them) and, in
        // fours, the one at depth DH - 1: 15 LDS reads per thread instead of 7 per cell.  Exactly one cell of a root-to-leaf chain is a list node.
        for (int c0 = tid * 8; c0 < ncell; c0 += QT_T * 8) {
            const uint4 q = *reinterpret_cast<const uint4*>(&hist[c0]);                   // eight u16 counts of the deepest level (hoff(DH) == 0)
            const uint32_t w[4] = { q.x, q.y, q.z, q.w };
            unsigned long long key[8];
#pragma unroll
            for (int k = 0; k < 8; k++) key[k] = ((w[k >> 1] >> (16 * (k & 1))) & 0xFFFFu) ? gbest[c0 + k] : 0ull;                        // loads, all in flight
            if ((q.x | q.y | q.z | q.w) == 0u) continue;
            int mc[5], e_common = 0xFFFF;
#pragma unroll
            for (int d = 0; d < 5; d++) mc[d] = d <= DH - 2 ? (int)mark[hoff(d) + (c0 >> (2 * (DH - d)))] : 0xFFFF;
#pragma unroll
            for (int d = 0; d < 5; d++) if (mc[d] != 0xFFFF) e_common = mc[d];
            const int m1a = mark[hoff(DH - 1) + (c0 >> 2)], m1b = mark[hoff(DH - 1) + (c0 >> 2) + 1];
            const uint4 m2 = *reinterpret_cast<const uint4*>(&mark[c0]);                  // marks of the eight cells themselves (hoff(DH) == 0)
            const uint32_t m2w[4] = { m2.x, m2.y, m2.z, m2.w };
#pragma unroll
            for (int k = 0; k < 8; k++) {
                if (!key[k]) continue;
                const int mk = (int)((m2w[k >> 1] >> (16 * (k & 1))) & 0xFFFFu), m1 = k < 4 ? m1a : m1b;
                const int node = mk != 0xFFFF ? mk : (m1 != 0xFFFF ? m1 : e_common);
                gbest[c0 + k] = 0ull;
                atomicMax(&best[node], key[k]);
            }
        }
        best_pending = false;
    } else if (in_lds) {
        if constexpr (QT_PTS > 0) {
        uint32_t sk[QT_PTS / QT_T]; int nd[QT_PTS / QT_T];
#pragma unroll
        for (int k = 0; k < QT_PTS / QT_T; k++) { const int p = tid + k * QT_T; sk[k] = p < n ? psk[p] : 0u; nd[k] = p < n ? (int)s_pnode[p] : 0; }      // global loads, all in flight
        if (keyed) {
#pragma unroll
            for (int k = 0; k < QT_PTS / QT_T; k++) nd[k] = node_of_key(nd[k]);
        }
#pragma unroll
        for (int k = 0; k < QT_PTS / QT_T; k++) { const int p = tid + k * QT_T; if (p < n) offer(s_pxy[p], sk[k], nd[k]); }
        }
    } else {
        for (int p0 = tid; p0 < n; p0 += 8 * QT_T) {                  // points in global memory: eight records in flight per thread
            uint32_t xy[8], sk[8]; int nd[8];
#pragma unroll
            for (int k = 0; k < 8; k++) { const int p = p0 + k * QT_T; const bool on = p < n; xy[k] = on ? pxy[p] : 0u; sk[k] = on ? psk[p] : 0u; nd[k] = on ? (int)pnode[p] : 0; }
#pragma unroll
            for (int k = 0; k < 8; k++) { const int p = p0 + k * QT_T; if (p < n) offer(xy[k], sk[k], keyed ? node_of_key(nd[k]) : nd[k]); }
        }
    }
    zero_gbest();                                                        // (a no-op unless a path above left the keys in place)
    __syncthreads();
    QT_MARK(40);
    for (int i = tid; i < S; i += QT_T) {
        if (i < L.sel_cap) {
            unsigned long long key = best[i];
            unsigned long long order = 0x00FFFFFFFFFFFFFFull - (key & 0x00FFFFFFFFFFFFFFull);
            uint32_t xy = (uint32_t)order;
            out[3 * i + 0] = (xy & 0xFFFF) + HS_BORDER;       // keypoints[i].pt.x += minBorderX (:484-485)
            out[3 * i + 1] = (xy >> 16) + HS_BORDER;
            out[3 * i + 2] = (uint32_t)(key >> 56);
        }
    }
    if (tid == 0) *out_n = min(S, L.sel_cap);
    QT_MARK(41);
    // ---- spatial order of the kept keypoints for the describe stage: perm[rank] = list index, ranked by 64-px tile (row-major) and list
    //      index.  The describe kernel walks the keypoints of an image in this order (neighbouring patches share their 128-byte lines in one
    //      XCD's L2) but writes every result to its list-order slot, so the output order stays the reference's.
    //      A counting sort over the tiles (the order inside a tile is irrelevant): histogram, block scan, scatter.
    {
        const int Sc = min(S, L.sel_cap);
        uint16_t* const perm = sel_perm + (size_t)img * sel_img_stride + L.sel_off;
        constexpr int MAXT = QT_MAXT;                                  // tile counters live in ccount behind the best-point slots (8 QT_M bytes, whichever array holds them)
        uint32_t* const tcnt = ccount + 2 * QT_M;
        static_assert(sizeof(ccount) >= (2 * QT_M + MAXT) * 4, "tile counters");
        const int ntx = (L.w + 63) >> 6, ntiles = ntx * ((L.h + 63) >> 6);
        if (ntiles <= MAXT) {
            constexpr int TPT = MAXT / QT_T;                           // tiles per thread in the scan
            for (int t = tid; t < ntiles; t += QT_T) tcnt[t] = 0;
            __syncthreads();
            int tile[(QT_M + QT_T - 1) / QT_T];
#pragma unroll
            for (int k = 0; k < (QT_M + QT_T - 1) / QT_T; k++) {
                const int i = tid + k * QT_T;
                tile[k] = -1;
                if (i < Sc) {
                    const uint32_t xy = (uint32_t)(0x00FFFFFFFFFFFFFFull - (best[i] & 0x00FFFFFFFFFFFFFFull));
                    tile[k] = (int)((((xy >> 16) + HS_BORDER) >> 6) * ntx + (((xy & 0xFFFF) + HS_BORDER) >> 6));
                    atomicAdd(&tcnt[tile[k]], 1u);
                }
            }
            __syncthreads();
            int c[TPT], local = 0;
#pragma unroll
            for (int k = 0; k < TPT; k++) { const int t = tid * TPT + k; c[k] = t < ntiles ? (int)tcnt[t] : 0; local += c[k]; }
            int tot; int pre = block_scan_excl(local, s_wave, sflip, tot);
#pragma unroll
            for (int k = 0; k < TPT; k++) { const int t = tid * TPT + k; if (t < ntiles) tcnt[t] = (uint32_t)pre; pre += c[k]; }
            __syncthreads();
#pragma unroll
            for (int k = 0; k < (QT_M + QT_T - 1) / QT_T; k++)
                if (tile[k] >= 0) perm[atomicAdd(&tcnt[tile[k]], 1u)] = (uint16_t)(tid + k * QT_T);
        } else {
            for (int i = tid; i < Sc; i += QT_T) perm[i] = (uint16_t)i;       // levels beyond QT_MAXT tiles: list order
        }
    }
    QT_MARK(4);
#ifdef HS_QT_PROFILE
    if (tid == 0 && level_first + blockIdx.x == 0 && blockIdx.y == 0) g_qt_prof[127] = qt_k;
#endif
}

void hs_launch_quadtree(const HsLevel* d_lv, int nlevels, int batch, int total_cells,
                        const uint2* cand, const int32_t* cell_count, uint64_t cand_img_stride,
                        uint32_t* pts_xy, uint32_t* pts_sk, uint16_t* pt_node, int32_t* cand_count,
                        uint32_t* sel_xys, int32_t* sel_count, int sel_img_stride, uint16_t* sel_perm, int force_point_domain, int level_first, int level_count,
                        uint32_t* qhist, unsigned long long* qbest, uint32_t qhist_img_stride, uint32_t qbest_img_stride, int keep_points, int list_mode, uint8_t* rect_scratch, hipStream_t s)
{
    if (level_count <= 0) return;
    dim3 grid(level_count, batch, 1);
    // more workgroups than CUs and every list fits the small instance: two workgroups per CU (77 KB of LDS each) instead of rounds of 256
    const bool small = list_mode == 1 && (long long)level_count * batch > 256;
    if (list_mode == 2)      // a quota above HS_QT_MAX_NODES - 8 (the reference's init extractor of its "Imaging" camera: 9000 features @1.4 -> 2758 on level 0)
        hipLaunchKernelGGL((k_quadtree<HS_QT_LARGE_NODES, 0, true>), grid, dim3(QT_T), 0, s, d_lv, nlevels, total_cells, cand, cell_count, cand_img_stride,
                           pts_xy, pts_sk, pt_node, cand_count, sel_xys, sel_count, sel_img_stride, sel_perm, force_point_domain, level_first,
                           qhist, qbest, qhist_img_stride, qbest_img_stride, keep_points, rect_scratch);
    else if (small)
        hipLaunchKernelGGL((k_quadtree<QT_M_SMALL, 0, false>), grid, dim3(QT_T), 0, s, d_lv, nlevels, total_cells, cand, cell_count, cand_img_stride,
                           pts_xy, pts_sk, pt_node, cand_count, sel_xys, sel_count, sel_img_stride, sel_perm, force_point_domain, level_first,
                           qhist, qbest, qhist_img_stride, qbest_img_stride, keep_points, nullptr);
    else
        hipLaunchKernelGGL((k_quadtree<HS_QT_MAX_NODES, QT_PTS_BIG, false>), grid, dim3(QT_T), 0, s, d_lv, nlevels, total_cells, cand, cell_count, cand_img_stride,
                           pts_xy, pts_sk, pt_node, cand_count, sel_xys, sel_count, sel_img_stride, sel_perm, force_point_domain, level_first,
                           qhist, qbest, qhist_img_stride, qbest_img_stride, keep_points, nullptr);
}

// bytes of global scratch ONE workgroup (one (image, level)) of the large-list instance needs for its two rectangle buffers
size_t hs_quadtree_large_scratch_bytes() { return 2 * sizeof(QtRects<HS_QT_LARGE_NODES>); }

// largest list the small instance holds (hs_api.hip: every level's quota + 8 must fit)
int hs_quadtree_small_nodes() { return QT_M_SMALL; }

// Host side of the geometric-key tables (see k_quadtree): the expressions of the kernel's former in-kernel build, evaluated once per
// configuration.  hs_api.hip is compiled with -ffp-contract=off and IEEE division, like the device code, so the floats agree.
bool hs_quadtree_build_tables(HsLevel& V, std::vector<uint8_t>& blob, size_t& xoff, size_t& yoff)
{
    V.qt_xtab = V.qt_ytab = nullptr;
    for (int i = 0; i < 8; i++) V.qt_rbound[i] = 0;
    const int nIni = V.n_ini;
    if (nIni < 1 || nIni > 8 || V.qt_w <= 0 || V.qt_h <= 0 || V.qt_w >= 8192 - 16 || V.qt_h >= 4096 - 16) return false;
    const float hX = V.hx;
    const int DH = nIni <= 2 ? 6 : 5;
    auto root_of = [&](int x) { return std::min((int)((float)x / hX), nIni - 1); };
    auto descend = [&](int x, int x0, int x1) {
        int idx = 0;
        for (int d = 0; d < DH; d++) {
            const int mx = x0 + ((x1 - x0 + 1) >> 1);
            if (x < mx) { x1 = mx; idx = 2 * idx; } else { x0 = mx; idx = 2 * idx + 1; }
        }
        return idx;
    };
    for (int r = 1; r < nIni; r++) {                                 // smallest x that lands in root r: around r * hX
        int bnd = (int)(hX * (float)r);
        while (bnd > 0 && root_of(bnd - 1) >= r) bnd--;
        while (root_of(bnd) < r) bnd++;
        V.qt_rbound[r] = bnd;
    }
    auto grow = [&](size_t n) { const size_t o = (blob.size() + 15) & ~(size_t)15; blob.resize(o + ((n + 15) & ~(size_t)15), 0); return o; };
    xoff = grow((size_t)V.qt_w + 1);
    for (int x = 0; x <= V.qt_w; x++) {
        const int r = root_of(x);
        blob[xoff + x] = (uint8_t)descend(x, (int16_t)(int)(hX * (float)r), (int16_t)(int)(hX * (float)(r + 1)));
    }
    yoff = grow((size_t)V.qt_h + 1);
    for (int y = 0; y <= V.qt_h; y++) blob[yoff + y] = (uint8_t)descend(y, 0, (int16_t)V.qt_h);
    return true;
}
