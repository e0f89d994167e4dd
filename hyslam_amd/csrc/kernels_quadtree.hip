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
// Inputs are the per-cell candidate slots written by k_fast_rows (gathered into a dense list first).  Outputs per (image, level):
// selected (x,y,score) in final list order + count.  Bound: LDS atomics / VALU; HBM traffic negligible.
#include "hs_internal.h"
#include <cstdlib>

#define QT_T HS_QT_THREADS
#define QT_M HS_QT_MAX_NODES
#define QT_PTS 6144              // points kept in LDS (a 1080p level has ~5000 candidates); more fall back to the global arrays

struct alignas(16) QtNodes {
    int16_t x0[QT_M], x1[QT_M], y0[QT_M], y1[QT_M];
    uint32_t cnt[QT_M];
};

// exclusive scan of one int per thread over the 1024-thread block; returns prefix, writes total
__device__ __forceinline__ int block_scan_excl(int v, int* s_wave /*[16]*/, int& total)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int incl = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { int n = __shfl_up(incl, o, 64); if (lane >= o) incl += n; }
    __syncthreads();                       // protect s_wave reuse
    if (lane == 63) s_wave[wave] = incl;
    __syncthreads();
    int base = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < QT_T / 64; w++) { int x = s_wave[w]; if (w < wave) base += x; tot += x; }
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

__device__ __forceinline__ int child_of(const QtNodes& N, int nd, int x, int y)
{
    // DivideNode: halfX = ceil((UR.x-UL.x)/2); n1.UR.x = UL.x+halfX; n1.BR.y = UL.y+halfY
    int mx = N.x0[nd] + ((N.x1[nd] - N.x0[nd] + 1) >> 1);
    int my = N.y0[nd] + ((N.y1[nd] - N.y0[nd] + 1) >> 1);
    return (x < mx ? 0 : 1) + (y < my ? 0 : 2);      // n1,n2,n3,n4
}

#ifdef HS_QT_PROFILE
__device__ unsigned long long g_qt_prof[128];
extern "C" void hs_debug_qt_profile(unsigned long long* out128) { (void)hipDeviceSynchronize(); (void)hipMemcpyFromSymbol(out128, HIP_SYMBOL(g_qt_prof), sizeof(unsigned long long) * 128); }
#define QT_MARK(tag) do { if (tid == 0 && blockIdx.x == 0 && blockIdx.y == 0 && qt_k < 126) { g_qt_prof[qt_k++] = __builtin_amdgcn_s_memtime(); g_qt_prof[qt_k++] = (tag); } } while (0)
#else
#define QT_MARK(tag)
#endif
__global__ __launch_bounds__(QT_T) void k_quadtree(const HsLevel* __restrict__ lv, int nlevels, int total_cells,
                                                   const uint32_t* __restrict__ cand_xy, const uint32_t* __restrict__ cand_sk,
                                                   const int32_t* __restrict__ cell_count, uint64_t cand_img_stride,
                                                   uint32_t* __restrict__ pts_xy_all, uint32_t* __restrict__ pts_sk_all,
                                                   uint16_t* __restrict__ pt_node_all, int32_t* __restrict__ cand_count,
                                                   uint32_t* __restrict__ sel_xys, int32_t* __restrict__ sel_count, int sel_img_stride,
                                                   uint16_t* __restrict__ sel_perm)
{
    __shared__ QtNodes nodes[2];
    __shared__ uint32_t ccount[4 * QT_M];          // child counts, indexed 4*rank + child
    __shared__ int16_t proc_rank[QT_M];            // processing rank of a node in this pass, -1 = not split
    __shared__ int16_t order_node[QT_M];           // rank -> node
    __shared__ uint16_t new_index[QT_M];           // surviving node -> index in the next list
    __shared__ uint16_t child_index[4 * QT_M];     // 4*rank+child -> index in the next list
    __shared__ uint32_t s_pxy[QT_PTS];             // the level's points (y<<16|x) and their node, when there are <= QT_PTS of them:
    __shared__ uint16_t s_pnode[QT_PTS];           // every pass walks the points twice, from LDS instead of through L2
    __shared__ int s_wave[QT_T / 64];
    __shared__ int s_misc[8];

    const int tid = threadIdx.x;
#ifdef HS_QT_PROFILE
    int qt_k = 0;
#endif
    QT_MARK(0);
    const int level = blockIdx.x, img = blockIdx.y;
    const HsLevel& L = lv[level];
    const int N = L.quota;
    uint32_t* pxy = pts_xy_all + (size_t)img * cand_img_stride + L.cand_off;
    uint32_t* psk = pts_sk_all + (size_t)img * cand_img_stride + L.cand_off;
    uint16_t* pnode = pt_node_all + (size_t)img * cand_img_stride + L.cand_off;
    uint32_t* out = sel_xys + ((size_t)img * sel_img_stride + L.sel_off) * 3;
    int32_t* out_n = &sel_count[img * nlevels + level];

    // ---- gather this level's candidates from the per-cell slots the FAST kernel filled into one dense list.
    // The FAST kernel fills a cell's slots in no particular order; the reference's list order (vToDistributeKeys) is cell by cell and,
    // inside a cell, cv::FAST's row-major scan = ascending (y<<16 | x) = ascending cand_xy.  Up to 4 * QT_T cells per round:
    //   A  one thread per 4 cells: counts, block-wide exclusive scan, and the cell id of each of its records into LDS
    //   B  one thread per RECORD: key -> LDS;  then rank inside its cell by counting smaller keys (LDS reads), scatter to the dense list.
    // Record-parallel because the high pyramid levels have few cells with many records each (mean 20, up to 34 at level 7 of a 1080p
    // frame): a per-cell thread loop was k dependent global round trips long.  The node arrays are not live yet: their LDS is the scratch.
    int n = 0;
    {
        uint32_t* const s_key = reinterpret_cast<uint32_t*>(&nodes[0]);          // [QT_GKEYS]
        uint16_t* const s_cell = reinterpret_cast<uint16_t*>(ccount);            // [QT_GKEYS]
        uint32_t* const s_pre = reinterpret_cast<uint32_t*>(child_index);        // [QT_T * CPT] exclusive offset of the round's cell
        constexpr int QT_GKEYS = (int)(sizeof(nodes) / 4) < (int)(sizeof(ccount) / 2) ? (int)(sizeof(nodes) / 4) : (int)(sizeof(ccount) / 2);
        const int ncell = L.ncols * L.nrows;
        const int ccap = hs_cell_cap(L.wcell, L.hcell);
        const int32_t* ccnt = cell_count + (size_t)img * total_cells + L.cell_begin;
        const uint32_t* sxy = cand_xy + (size_t)img * cand_img_stride + L.cand_off;
        const uint32_t* ssk = cand_sk + (size_t)img * cand_img_stride + L.cand_off;
        constexpr int CPT = 4;                                     // cells per thread and round: the counts of a round are independent loads
        static_assert(sizeof(child_index) >= QT_T * CPT * 4, "s_pre scratch");
        int cpt = CPT;
        for (int c0 = 0; c0 < ncell;) {
            // a round takes cpt cells per thread: as many as leave the round's records inside the scratch (dense frames: a 4000x3000 level
            // has 3.3 records per cell and 4096 cells overflowed it — the per-cell fallback below then cost a quarter of the kernel);
            // the next round starts from the density this one found
            int k[CPT], ksum, tot, pre;
            for (;;) {
                ksum = 0;
#pragma unroll
                for (int q = 0; q < CPT; q++) { const int c = c0 + tid * cpt + q; k[q] = (q < cpt && c < ncell) ? min(ccnt[c], ccap) : 0; ksum += k[q]; }
                pre = block_scan_excl(ksum, s_wave, tot);
                if (tot <= QT_GKEYS || cpt == 1) break;
                cpt >>= 1;
            }
            const int round_cells = QT_T * cpt;
            if (tot <= QT_GKEYS) {
#pragma unroll
                for (int q = 0; q < CPT; q++) {
                    if (q < cpt) s_pre[tid * cpt + q] = (uint32_t)pre;
                    for (int i = 0; i < k[q]; i++) s_cell[pre + i] = (uint16_t)(tid * cpt + q);
                    pre += k[q];
                }
                __syncthreads();
                for (int e = tid; e < tot; e += QT_T) {
                    const int lc = s_cell[e];
                    s_key[e] = sxy[(size_t)(c0 + lc) * ccap + (e - (int)s_pre[lc])];
                }
                __syncthreads();
                for (int e = tid; e < tot; e += QT_T) {
                    const int lc = s_cell[e];
                    const int first = (int)s_pre[lc], i = e - first;
                    const uint32_t sk = ssk[(size_t)(c0 + lc) * ccap + i];           // in flight during the rank loop
                    const int last = (lc + 1 < round_cells) ? (int)s_pre[lc + 1] : tot;   // cells past the last one have k = 0: s_pre = tot
                    const uint32_t key = s_key[e];
                    int rank = 0;
                    for (int j = first; j < last; j++) rank += s_key[j] < key;
                    pxy[n + first + rank] = key; psk[n + first + rank] = sk;
                    if (n + first + rank < QT_PTS) s_pxy[n + first + rank] = key;
                }
                __syncthreads();                                   // the scratch is reused by the next round
            } else {                                               // saturated image: more records than the scratch holds; one thread per cell
#pragma unroll
                for (int q = 0; q < CPT; q++) {                    // cpt == 1 here: k[1..] = 0
                    const size_t src = (size_t)min(c0 + tid * cpt + q, ncell - 1) * ccap;
                    for (int i = 0; i < k[q]; i++) {
                        const uint32_t key = sxy[src + i];
                        int rank = 0;
                        for (int j = 0; j < k[q]; j++) rank += sxy[src + j] < key;
                        pxy[n + pre + rank] = key; psk[n + pre + rank] = ssk[src + i];
                        if (n + pre + rank < QT_PTS) s_pxy[n + pre + rank] = key;
                    }
                    pre += k[q];
                }
            }
            n += tot;
            c0 += round_cells;
        }
        __syncthreads();      // the dense list is complete (written and read by this workgroup only)
    }
    if (tid == 0) cand_count[img * nlevels + level] = n;
    QT_MARK(1);
    const bool in_lds = n <= QT_PTS;               // uniform
    auto ld_xy = [&](int p) -> uint32_t { return in_lds ? s_pxy[p] : pxy[p]; };
    auto ld_node = [&](int p) -> int { return in_lds ? (int)s_pnode[p] : (int)pnode[p]; };
    auto st_node = [&](int p, int v) { if (in_lds) s_pnode[p] = (uint16_t)v; else pnode[p] = (uint16_t)v; };

    const int nIni = L.n_ini;
    const float hX = L.hx;
    if (n == 0 || nIni < 1 || nIni > QT_M / 4) { if (tid == 0) *out_n = 0; return; }

    // ---- roots (:183-225): count, drop empty ones, keep list order = root order
    for (int i = tid; i < nIni; i += QT_T) ccount[i] = 0;
    __syncthreads();
    for (int p = tid; p < n; p += QT_T) {
        int x = ld_xy(p) & 0xFFFF;
        int r = (int)((float)x / hX);               // vpIniNodes[kp.pt.x/hX]
        r = min(r, nIni - 1);
        wave_agg_inc(ccount, r, true);
    }
    __syncthreads();
    int cur = 0;
    if (tid == 0) {
        int S = 0;
        for (int i = 0; i < nIni; i++) {
            if (ccount[i] > 0) {
                nodes[0].x0[S] = (int16_t)(int)(hX * (float)i);
                nodes[0].x1[S] = (int16_t)(int)(hX * (float)(i + 1));
                nodes[0].y0[S] = 0; nodes[0].y1[S] = (int16_t)L.qt_h;
                nodes[0].cnt[S] = ccount[i];
                new_index[i] = (uint16_t)S;
                S++;
            }
        }
        s_misc[0] = S;
    }
    __syncthreads();
    for (int p = tid; p < n; p += QT_T) {
        int x = ld_xy(p) & 0xFFFF;
        int r = min((int)((float)x / hX), nIni - 1);
        st_node(p, new_index[r]);
    }
    int S = s_misc[0];
    __syncthreads();

    QT_MARK(2);
    bool phase2 = false;     // uniform across the block
    int T_prev = 0;          // number of children created by the previous pass (they sit at list indices [0,T_prev))
    bool finished = false;   // the distribution ended inside the fast-forward below

    // ---- fast-forward of the first (up to three) breadth-first passes.  While the reference is in its first phase EVERY multi-point node
    //      is split, so where a point ends up depends on geometry alone: its path through DivideNode's midpoints.  One sweep computes every
    //      point's depth-3 cell and a histogram of those cells; the per-depth counts are sums of it; a single wavefront then replays the list
    //      bookkeeping of the passes on those counts (a lane per node: which nodes split, which children exist, where they land in the
    //      list, when the reference would stop or switch to its second phase), and a second sweep labels the points with their node.  The
    //      generic pass below (a dozen block-wide steps with LDS atomics per pass) then only runs for what is left — normally the single
    //      size-ordered pass of the second phase.  Applies when the list still fits a wavefront (<= 4 root nodes).
    if (S <= 4) {
        uint32_t* const hist3 = ccount;                    // [S*64] points per depth-3 cell
        uint32_t* const cnt2 = ccount + 256;               // [S*16]
        uint32_t* const cnt1 = ccount + 320;               // [S*4]
        uint16_t* const idx_tab = child_index;             // [340] (depth, key) -> index in the final list, 0xFFFF = not a node of it
        constexpr int TAB0 = 0, TAB1 = 4, TAB2 = 20, TAB3 = 84;
        int16_t* const nkey = order_node;                  // per list entry: depth<<12 | key
        __shared__ int16_t s_rootb[16];                    // the roots' rectangles (the node arrays are overwritten by the replay)
        for (int i = tid; i < 340; i += QT_T) { ccount[i] = 0; idx_tab[i] = 0xFFFF; }
        if (tid < S) { s_rootb[4 * tid] = nodes[0].x0[tid]; s_rootb[4 * tid + 1] = nodes[0].x1[tid]; s_rootb[4 * tid + 2] = nodes[0].y0[tid]; s_rootb[4 * tid + 3] = nodes[0].y1[tid]; }
        __syncthreads();
        // path of a point below root s: three DivideNode decisions
        auto path_of = [&](int s, int x, int y) {
            int x0 = s_rootb[4 * s], x1 = s_rootb[4 * s + 1], y0 = s_rootb[4 * s + 2], y1 = s_rootb[4 * s + 3];
            int path = 0;
#pragma unroll
            for (int d = 0; d < 3; d++) {
                const int mx = x0 + ((x1 - x0 + 1) >> 1), my = y0 + ((y1 - y0 + 1) >> 1);
                const int c = (x < mx ? 0 : 1) + (y < my ? 0 : 2);
                if (c & 1) x0 = mx; else x1 = mx;
                if (c & 2) y0 = my; else y1 = my;
                path = path * 4 + c;
            }
            return path;
        };
        for (int p = tid; p < n; p += QT_T) {
            const uint32_t xy = ld_xy(p);
            const int s = ld_node(p);
            atomicAdd(&hist3[s * 64 + path_of(s, xy & 0xFFFF, xy >> 16)], 1u);
        }
        __syncthreads();
        if (tid < 64) {
            const int lane = tid;
            if (lane < S * 16) cnt2[lane] = hist3[4 * lane] + hist3[4 * lane + 1] + hist3[4 * lane + 2] + hist3[4 * lane + 3];
            __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_s_waitcnt(0xc07f);
            if (lane < S * 4) cnt1[lane] = cnt2[4 * lane] + cnt2[4 * lane + 1] + cnt2[4 * lane + 2] + cnt2[4 * lane + 3];
            if (lane < S) nkey[lane] = (int16_t)lane;                                     // depth 0, key = root list index
            __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_s_waitcnt(0xc07f);
            int Sw = S, Tw = 0, ph2 = 0, fin = 0, curw = 0, passes = 0;
            for (int pass = 1; pass <= 3; pass++) {
                if (Sw > 64) break;                                                         // the list no longer fits a lane per node
                QtNodes& Cw = nodes[curw]; QtNodes& Xw = nodes[curw ^ 1];
                const bool have = lane < Sw;
                const int cntv = have ? (int)Cw.cnt[lane] : 0;
                const bool split = have && cntv > 1;
                const int kd = have ? (int)(uint16_t)nkey[lane] : 0, key = kd & 0xFFF;
                const uint32_t* ctab = pass == 1 ? cnt1 : (pass == 2 ? cnt2 : hist3);
                int cc[4] = { 0, 0, 0, 0 };
                if (split) { cc[0] = (int)ctab[4 * key]; cc[1] = (int)ctab[4 * key + 1]; cc[2] = (int)ctab[4 * key + 2]; cc[3] = (int)ctab[4 * key + 3]; }
                const int nchild = (cc[0] > 0) + (cc[1] > 0) + (cc[2] > 0) + (cc[3] > 0);
                const int nexp = (cc[0] > 1) + (cc[1] > 1) + (cc[2] > 1) + (cc[3] > 1);
                int incl_c = nchild, incl_s = (have && !split) ? 1 : 0, sum_e = nexp;
#pragma unroll
                for (int o = 1; o < 64; o <<= 1) {
                    const int a = __shfl_up(incl_c, o, 64), b = __shfl_up(incl_s, o, 64);
                    if (lane >= o) { incl_c += a; incl_s += b; }
                }
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) sum_e += __shfl_xor(sum_e, o, 64);
                const int T = __builtin_amdgcn_readlane(incl_c, 63), nsurv = __builtin_amdgcn_readlane(incl_s, 63);
                if (T == 0) { fin = 1; break; }                                             // nothing can be split: size == prevSize (:309)
                if (T + nsurv > QT_M) { fin = 2; break; }
                if (split) {
                    const int x0 = Cw.x0[lane], x1 = Cw.x1[lane], y0 = Cw.y0[lane], y1 = Cw.y1[lane];
                    const int mx = x0 + ((x1 - x0 + 1) >> 1), my = y0 + ((y1 - y0 + 1) >> 1);
                    int ci = incl_c - nchild;
#pragma unroll
                    for (int c = 0; c < 4; c++) {
                        if (cc[c] > 0) {
                            const int pos = T - 1 - ci; ci++;
                            Xw.x0[pos] = (int16_t)((c & 1) ? mx : x0); Xw.x1[pos] = (int16_t)((c & 1) ? x1 : mx);
                            Xw.y0[pos] = (int16_t)((c & 2) ? my : y0); Xw.y1[pos] = (int16_t)((c & 2) ? y1 : my);
                            Xw.cnt[pos] = (uint32_t)cc[c];
                            new_index[pos] = (uint16_t)((pass << 12) | (4 * key + c));      // the next list's keys are staged here (nkey is still being read)
                        }
                    }
                } else if (have) {
                    const int pos = T + incl_s - 1;
                    Xw.x0[pos] = Cw.x0[lane]; Xw.x1[pos] = Cw.x1[lane]; Xw.y0[pos] = Cw.y0[lane]; Xw.y1[pos] = Cw.y1[lane]; Xw.cnt[pos] = Cw.cnt[lane];
                    new_index[pos] = (uint16_t)kd;
                }
                __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_s_waitcnt(0xc07f);
                const int prevS = Sw;
                Sw = T + nsurv; Tw = T; curw ^= 1; passes = pass;
                for (int i = lane; i < Sw; i += 64) nkey[i] = (int16_t)new_index[i];
                __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_s_waitcnt(0xc07f);
                if (Sw >= N || Sw == prevS) { fin = 1; break; }                            // (:307-313)
                if (Sw + sum_e * 3 > N) { ph2 = 1; break; }
            }
            // (depth, key) -> list index of every node of the resulting list
            if (passes > 0 && fin != 2)
                for (int i = lane; i < Sw; i += 64) {
                    const int kd = (int)(uint16_t)nkey[i], d = kd >> 12, key = kd & 0xFFF;
                    idx_tab[(d == 0 ? TAB0 : d == 1 ? TAB1 : d == 2 ? TAB2 : TAB3) + key] = (uint16_t)i;
                }
            if (lane == 0) { s_misc[2] = Sw; s_misc[3] = Tw; s_misc[4] = ph2; s_misc[5] = fin; s_misc[6] = curw; s_misc[7] = passes; }
        }
        __syncthreads();
        const int passes = s_misc[7];
        if (s_misc[5] == 2) { if (tid == 0) *out_n = 0; return; }                          // cannot happen for quota + 8 <= QT_M (host checks)
        if (passes > 0) {
            for (int p = tid; p < n; p += QT_T) {
                const uint32_t xy = ld_xy(p);
                const int s = ld_node(p);
                const int path = path_of(s, xy & 0xFFFF, xy >> 16);
                int e = idx_tab[TAB0 + s];
                if (e == 0xFFFF) e = idx_tab[TAB1 + s * 4 + (path >> 4)];
                if (e == 0xFFFF) e = idx_tab[TAB2 + s * 16 + (path >> 2)];
                if (e == 0xFFFF) e = idx_tab[TAB3 + s * 64 + path];
                st_node(p, e);
            }
            S = s_misc[2]; T_prev = s_misc[3]; phase2 = s_misc[4] != 0; finished = s_misc[5] == 1; cur = s_misc[6];
        } else {
            finished = s_misc[5] == 1;                                                      // no root can be split
        }
        __syncthreads();
    }
    QT_MARK(5);
    // ---- main loop
    for (int iter = 0; iter < 64 && !finished; iter++) {
        QtNodes& C = nodes[cur];
        QtNodes& X = nodes[cur ^ 1];
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
            int tot; int pre = block_scan_excl(local, s_wave, tot);
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
            // multi-point nodes created by the previous pass, sorted by (size desc, list index asc) (:321-325, D1)
            for (int i = tid; i < S; i += QT_T) proc_rank[i] = -1;
            __syncthreads();
            int tot_local = 0;
            for (int i = tid; i < T_prev; i += QT_T) {
                uint32_t ci = C.cnt[i];
                if (ci > 1) {
                    int r = 0;
                    const uint4* c4 = reinterpret_cast<const uint4*>(C.cnt);          // four counts per LDS read
                    int j = 0;
                    for (; j + 4 <= T_prev; j += 4) {
                        const uint4 q = c4[j >> 2];
                        r += (q.x > 1) && (q.x > ci || (q.x == ci && j < i));
                        r += (q.y > 1) && (q.y > ci || (q.y == ci && j + 1 < i));
                        r += (q.z > 1) && (q.z > ci || (q.z == ci && j + 2 < i));
                        r += (q.w > 1) && (q.w > ci || (q.w == ci && j + 3 < i));
                    }
                    for (; j < T_prev; j++) {
                        uint32_t cj = C.cnt[j];
                        r += (cj > 1) && (cj > ci || (cj == ci && j < i));
                    }
                    proc_rank[i] = (int16_t)r; order_node[r] = (int16_t)i;
                    tot_local++;
                }
            }
            int tot; block_scan_excl(tot_local, s_wave, tot);
            E = tot;
        }
        __syncthreads();
        QT_MARK(10 + (phase2 ? 100 : 0));
        if (E == 0) break;                          // nothing can be split: size == prevSize (:309,374)

        // -- count the four children of every candidate node
        for (int i = tid; i < 4 * E; i += QT_T) ccount[i] = 0;
        __syncthreads();
        if (in_lds) {
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
        __syncthreads();

        QT_MARK(11);
        // -- how many of them are actually split this pass
        int P = E;
        if (phase2) {
            // size after splitting the first k nodes = S + sum_{i<k}(children_i - 1); stop at the first k with size >= N
            int local = 0; int add[(QT_M + QT_T - 1) / QT_T];
#pragma unroll
            for (int k = 0; k < (QT_M + QT_T - 1) / QT_T; k++) {
                int r = tid * ((QT_M + QT_T - 1) / QT_T) + k;
                int a = 0;
                if (r < E) a = (ccount[4 * r] > 0) + (ccount[4 * r + 1] > 0) + (ccount[4 * r + 2] > 0) + (ccount[4 * r + 3] > 0) - 1;
                add[k] = a; local += a;
            }
            int tot; int pre = block_scan_excl(local, s_wave, tot);
            if (tid == 0) s_misc[1] = E;
            __syncthreads();
#pragma unroll
            for (int k = 0; k < (QT_M + QT_T - 1) / QT_T; k++) {
                int r = tid * ((QT_M + QT_T - 1) / QT_T) + k;
                if (r < E) {
                    int before = S + pre, after = before + add[k];
                    if (before < N && after >= N) s_misc[1] = r + 1;      // unique r: size is non-decreasing
                    pre = after - S;
                }
            }
            __syncthreads();
            P = s_misc[1];
        }

        QT_MARK(12);
        // -- lay out the next list: children of processed nodes in reverse creation order, then survivors
        int T, nToExpand;
        {
            int local = 0, lexp = 0;
            int fl[4 * ((QT_M + QT_T - 1) / QT_T)];
#pragma unroll
            for (int k = 0; k < 4 * ((QT_M + QT_T - 1) / QT_T); k++) {
                int i = tid * (4 * ((QT_M + QT_T - 1) / QT_T)) + k;
                uint32_t cc = (i < 4 * P) ? ccount[i] : 0;
                fl[k] = cc > 0; local += fl[k]; lexp += cc > 1;
            }
            int pre = block_scan_excl(local, s_wave, T);
            block_scan_excl(lexp, s_wave, nToExpand);
#pragma unroll
            for (int k = 0; k < 4 * ((QT_M + QT_T - 1) / QT_T); k++) {
                int i = tid * (4 * ((QT_M + QT_T - 1) / QT_T)) + k;
                if (i < 4 * P && fl[k]) {
                    int pos = T - 1 - pre; pre++;
                    child_index[i] = (uint16_t)pos;
                    int nd = order_node[i >> 2], c = i & 3;
                    int x0 = C.x0[nd], x1 = C.x1[nd], y0 = C.y0[nd], y1 = C.y1[nd];
                    int mx = x0 + ((x1 - x0 + 1) >> 1), my = y0 + ((y1 - y0 + 1) >> 1);
                    if (pos < QT_M) {
                        X.x0[pos] = (int16_t)((c & 1) ? mx : x0); X.x1[pos] = (int16_t)((c & 1) ? x1 : mx);
                        X.y0[pos] = (int16_t)((c & 2) ? my : y0); X.y1[pos] = (int16_t)((c & 2) ? y1 : my);
                        X.cnt[pos] = ccount[i];
                    }
                }
            }
        }
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
            int pre = block_scan_excl(local, s_wave, nsurv);
#pragma unroll
            for (int k = 0; k < (QT_M + QT_T - 1) / QT_T; k++) {
                int i = tid * ((QT_M + QT_T - 1) / QT_T) + k;
                if (fl[k]) {
                    int pos = T + pre; pre++;
                    new_index[i] = (uint16_t)pos;
                    if (pos < QT_M) {
                        X.x0[pos] = C.x0[i]; X.x1[pos] = C.x1[i]; X.y0[pos] = C.y0[i]; X.y1[pos] = C.y1[i];
                        X.cnt[pos] = C.cnt[i];
                    }
                }
            }
        }
        __syncthreads();
        if (T + nsurv > QT_M) { if (tid == 0) *out_n = 0; return; }     // cannot happen for quota+8 <= QT_M (host checks)

        QT_MARK(13);
        // -- relabel the points
        if (in_lds) {
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

    QT_MARK(3);
    // ---- keep the best point of every node (:381-400), emit in list order
    unsigned long long* best = reinterpret_cast<unsigned long long*>(ccount);      // QT_M * 8 bytes <= sizeof(ccount)
    for (int i = tid; i < S; i += QT_T) best[i] = 0ull;
    __syncthreads();
    for (int p = tid; p < n; p += QT_T) {
        uint32_t xy = ld_xy(p), sk = psk[p];
        unsigned long long order = ((unsigned long long)(sk & 0xFFFFFFu) << 32) | xy;          // (cell, y, x)
        unsigned long long key = ((unsigned long long)(sk >> 24) << 56) | (0x00FFFFFFFFFFFFFFull - order);
        atomicMax(&best[ld_node(p)], key);
    }
    __syncthreads();
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
    // ---- spatial order of the kept keypoints for the describe stage: perm[rank] = list index, ranked by 64-px tile (row-major) and list
    //      index.  The describe kernel walks the keypoints of an image in this order (neighbouring patches share their 128-byte lines in one
    //      XCD's L2) but writes every result to its list-order slot, so the output order stays the reference's.
    //      A counting sort over the tiles (the order inside a tile is irrelevant): histogram, block scan, scatter.
    {
        const int Sc = min(S, L.sel_cap);
        uint16_t* const perm = sel_perm + (size_t)img * sel_img_stride + L.sel_off;
        constexpr int MAXT = 2 * QT_M;                                 // tile counters live in the upper half of ccount (best[] occupies the lower half)
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
            int tot; int pre = block_scan_excl(local, s_wave, tot);
#pragma unroll
            for (int k = 0; k < TPT; k++) { const int t = tid * TPT + k; if (t < ntiles) tcnt[t] = (uint32_t)pre; pre += c[k]; }
            __syncthreads();
#pragma unroll
            for (int k = 0; k < (QT_M + QT_T - 1) / QT_T; k++)
                if (tile[k] >= 0) perm[atomicAdd(&tcnt[tile[k]], 1u)] = (uint16_t)(tid + k * QT_T);
        } else {
            for (int i = tid; i < Sc; i += QT_T) perm[i] = (uint16_t)i;       // levels beyond 4096 tiles: list order
        }
    }
    QT_MARK(4);
#ifdef HS_QT_PROFILE
    if (tid == 0 && blockIdx.x == 0 && blockIdx.y == 0) g_qt_prof[127] = qt_k;
#endif
}

void hs_launch_quadtree(const HsLevel* d_lv, int nlevels, int batch, int total_cells,
                        const uint32_t* cand_xy, const uint32_t* cand_sk, const int32_t* cell_count, uint64_t cand_img_stride,
                        uint32_t* pts_xy, uint32_t* pts_sk, uint16_t* pt_node, int32_t* cand_count,
                        uint32_t* sel_xys, int32_t* sel_count, int sel_img_stride, uint16_t* sel_perm, hipStream_t s)
{
    dim3 grid(nlevels, batch, 1);
    hipLaunchKernelGGL(k_quadtree, grid, dim3(QT_T), 0, s, d_lv, nlevels, total_cells, cand_xy, cand_sk, cell_count, cand_img_stride,
                       pts_xy, pts_sk, pt_node, cand_count, sel_xys, sel_count, sel_img_stride, sel_perm);
}
