// hs_api.hip — host side of the C ABI declared in include/hyslam_amd.h.
// Owns the per-handle device workspace (sized for 288 GB HBM: worst-case candidate storage, no overflow
// paths), the host-computed tables (scale factors, per-level quotas, resize coefficients) and the launch
// sequence.  The whole extraction of a batch is GPU-resident: pyramid -> FAST/NMS cells -> quadtree
// distribution -> blur+orientation+rBRIEF, 10 kernel launches for an 8-level pyramid regardless of batch size.
#include "hs_internal.h"
#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#define HS_VERSION "hyslam_amd 0.1 (gfx950)"

struct hs_orb {
    hs_orb_params p;
    int device = 0;
    // communicators created on this handle (hs_comm_create) BORROW it — its device and its stream: while one is alive hs_orb_destroy only drops
    // the owner's reference and the last hs_comm_destroy frees the handle, so the two destroy calls are safe in either order and on two threads
    std::atomic<bool> owned{true};            // hs_orb_destroy has not been called yet (hs_orb_borrowers = refs minus the owner's reference)
    std::atomic<int> refs{1};                 // the owner's reference (dropped by hs_orb_destroy) + one per communicator created on the handle; whoever drops the last one frees it
    hipStream_t stream = nullptr;
    std::string err;
    uint16_t taps[7];
    HsFastKnobs fast_knobs{};          // HS_FAST_* environment knobs, read once in hs_orb_create
    uint32_t fast_epoch = 0;           // FAST launches on this workspace so far (selects the work-queue counter set)
    int split_mode = -1;               // HS_EXTRACT_SPLIT (read once): 1 = always run level 0's FAST + quadtree beside the pyramid, 0 = never, -1 = for small batches
    hipStream_t s_aux = nullptr; hipEvent_t ev_sfork = nullptr, ev_sjoin = nullptr;      // the second launch sequence of the split and its fences
    bool qt_point_domain = false;      // HS_QT_POINT_DOMAIN=1 (read once): the quadtree's general point-domain passes only (parity tests of the fallback)
    int fast_order = 1;                // HS_FAST_ORDER (read once): order of the FAST work items of an image: 1 = reduced levels deepest first, level 0 last; 0 = level 0 first (the order until round 3); 2 = reduced levels interleaved, level 0 last
    bool stereo_fuse = true;           // HS_STEREO_FUSE=0 (read once): the stereo front end with a separate k_stereo_strips launch instead of the strips binned by an extra workgroup of the describe launch
    bool no_fuse = false;              // HS_PYRAMID_NO_FUSE=1 (read once): one pyramid level per launch (parity tests of the unfused kernel)
    bool fast_taps = false;            // every tap fits a byte and the 16-bit row sums cannot saturate
    // ORBExtractor ctor tables (ORBExtractor.cpp:86-118)
    std::vector<float> scale, inv_scale, sigma2, inv_sigma2;
    std::vector<int> quota;
    // geometry currently configured
    int w = 0, h = 0, batch_cap = 0;
    std::vector<HsLevel> lv;
    std::vector<HsLevel> lv_n;         // the same levels with the NARROW FAST work items (grp_cells / ngroups / item_begin differ); device copy at d_lv + nlevels
    int total_cells = 0, max_wcell = 1, max_hcell = 1;
    int fast_items = 0;                // FAST work items per image (HsLevel::item_begin)
    int fast_items_n = 0;              // narrow items per image; 0 = no narrow list (a cell wider than the narrow tile)
    uint64_t cand_img_stride = 0;      // candidate entries per image
    int sel_img_stride = 0;            // selection entries per image
    int max_kp = 0;
    // device memory
    HsLevel* d_lv = nullptr;
    HsFastItem* d_fast_items = nullptr;
    HsFastItem* d_fast_items_n = nullptr;
    uint32_t* d_fast_ovf = nullptr;
    uint8_t* d_pyr = nullptr; size_t pyr_bytes = 0;
    int16_t* d_tables = nullptr;
    uint8_t* d_qt_tabs = nullptr;      // geometric-key tables of the count-domain quadtree (hs_quadtree_build_tables)
    // round 4: the FAST kernel computes the candidates' geometric keys and leaves their histogram + the best candidate per deepest cell in global
    // memory (HsFastQt, HsLevel::qt_hist_off): u16 key tables, the per-level records, the two arrays ([batch][stride]; all zero between calls:
    // the quadtree kernel zeroes what it consumes)
    uint16_t* d_qkeys = nullptr; HsFastQt* d_fast_qt = nullptr;
    uint32_t* d_qhist = nullptr; unsigned long long* d_qbest = nullptr; uint32_t qhist_stride = 0, qbest_stride = 0;
    int fast_keys_levels = HS_MAX_LEVELS;   // HS_FAST_KEYS_LEVELS (read once; tuning): only the levels 0 .. n-1 get keys
    int fast_keys_max_batch = 16;      // HS_FAST_KEYS_MAX_BATCH (read once): calls of more frames than this run without the keys (see run_extract)
    bool qt_large = false;             // a level's quota + 8 exceeds HS_QT_MAX_NODES (up to HS_QT_LARGE_NODES): the quadtree kernel's large-list instance, rectangles in d_qt_rects
    uint8_t* d_qt_rects = nullptr;     // qt_large: batch_cap * nlevels * hs_quadtree_large_scratch_bytes()
    bool qt_small_ok = false;          // every level's list (quota + 8 nodes) fits the quadtree kernel's small instance (two workgroups per CU; HS_QT_SMALL=0 switches it off, read once)
    bool keys_dirty = false;           // a keyed call was enqueued and did not reach its end (any error return of run_extract): d_qhist / d_qbest may hold stale keys -> zeroed before the next call
    bool fast_keys = true;             // HS_FAST_KEYS=0 (read once): the quadtree kernel gathers the candidates and computes the keys itself (the scheme until round 3)
    bool keep_points = false;          // hs_orb_set_debug(h, 1): the quadtree kernel also gathers the candidates into the dense point arrays (hs_orb_debug_candidates reads them)
    uint8_t* d_pyr_tabs = nullptr;     // tile / row records of the two-level pyramid kernel (hs_pyramid_build_tables)
    std::vector<HsPyrFuse> pyr_fuse;   // [level]: kernel argument of the pair (level, level + 1) when it is fused
    std::vector<HsPyrChain> pyr_deep;  // [level]: the small-batch plan — chains as long as the LDS allows (8 levels: all seven in ONE launch); valid = 0 where none starts
    int deep_rows = 8;                 // HS_PYRAMID_DEEP_ROWS (read once): rows of the LAST level per tile in the small-batch plan (a workgroup's stages are a dependent
                                       // sequence whose length goes with the rows per wave: more, flatter tiles shorten the launch although their halo rows cost more work)
    int deep_max_batch = 2;            // HS_PYRAMID_DEEP_MAX (read once): calls of at most this many frames use the small-batch plan (0 = never)
    std::vector<HsPyrChain> pyr_chain; // [level]: kernel argument of the chain launch that starts at this level (HsLevel::chain_n levels)
    int pyr_tbx_max = 0;               // HS_PYRAMID_TBX_MAX (read once): cap on the level-B tile width of the two-level kernel (experiment: lane utilisation against time)
    int chain_mode = -1;               // HS_PYRAMID_CHAIN (read once): -1 = a three-level chain for the tail of an odd number of levels, 0 = never, 2 = chains for every fused pair too (parity tests)
    uint2* d_cand = nullptr; uint32_t *d_pts_xy = nullptr, *d_pts_sk = nullptr; uint16_t* d_pt_node = nullptr;
    int32_t *d_cand_count = nullptr, *d_sel_count = nullptr, *d_cell_count = nullptr;
    uint32_t* d_sel = nullptr;
    uint16_t* d_sel_perm = nullptr;    // spatial order of every level's selection (describe stage)
    uint16_t* d_taps = nullptr;
    // staging for the host-pointer entry points
    uint8_t* d_in = nullptr; size_t in_bytes = 0; size_t in_pitch = 0;
    uint8_t* d_raw = nullptr; size_t raw_bytes = 0;      // hs_orb_extract_camera_batch: the camera's frames as uploaded (before PreProcessImg on the device)
    hs_keypoint* d_kps = nullptr; uint8_t* d_desc = nullptr; int32_t* d_n = nullptr; int out_cap = 0, out_batch = 0;
    float *d_ur = nullptr, *d_depth = nullptr; int32_t* d_bd = nullptr; size_t st_entries = 0;
    int32_t* d_strip_count = nullptr; void* d_strip_list = nullptr; size_t strip_count_entries = 0, strip_list_entries = 0;
    // persistent staging of hs_stereo_match (host-pointer call): device keypoints / descriptors / counts and one pinned host block
    hs_keypoint* d_sm_kps = nullptr; uint8_t* d_sm_desc = nullptr; int32_t* d_sm_n = nullptr; int sm_cap = 0;
    uint8_t* h_pin = nullptr; size_t pin_bytes = 0;
    // hs_orb_extract_batch (host-pointer call): one pinned block the three outputs come back into
    uint8_t* h_pin_out = nullptr; size_t pin_out_bytes = 0;
    int last_batch = 0; HsImg0 last_img0{};
    int last_pyr_launches = 0;         // kernel launches the pyramid stage of the last call really enqueued (hs_launch_pyramid's return value)
    int last_stereo_launches = 2;      // launches of stage 4 in the last stereo call: strips + match (run_stereo) or match only (the front end with the strips inside the describe launch)
    // where the last host-pointer extraction (hs_orb_extract[_batch], hs_orb_wait) left its results on the DEVICE: what hs_frame_publish keeps
    const hs_keypoint* pub_kps = nullptr; const uint8_t* pub_desc = nullptr; int pub_cap = 0, pub_batch = 0;
    // bump-allocated scratch for the host-pointer matcher entry points
    uint8_t* d_scratch = nullptr; size_t scratch_bytes = 0, scratch_used = 0;
    // pipelined host ingest (hs_orb_submit_batch / hs_orb_wait): two staging slots, a copy-in and a copy-out stream next to the compute stream
    struct IngestSlot {
        uint8_t* d_in = nullptr; size_t in_bytes = 0;          // frames of the batch in HBM
        uint8_t* d_raw = nullptr; size_t raw_bytes = 0;        // hs_orb_submit_camera_batch: the camera's frames as uploaded (before PreProcessImg)
        uint8_t* d_out = nullptr; size_t out_bytes = 0;        // [counts | keypoints | descriptors | uRight | depth] in HBM
        uint8_t* h_out = nullptr; size_t h_out_bytes = 0;      // the same block in page-locked host memory
        hipEvent_t ev_in = nullptr, ev_done = nullptr, ev_out = nullptr;
        int32_t ticket = 0; bool busy = false;
        int batch = 0, pairs = 0, cap = 0;
        size_t off_k = 0, off_d = 0, off_u = 0, off_z = 0, used = 0;
    } slot[2];
    hipStream_t s_in = nullptr, s_out = nullptr;
    int32_t next_ticket = 1;
    // second lane (hs_orb_set_lanes): a child handle with its own workspace and stream, fenced against the caller's stream by two events
    hs_orb* lane2 = nullptr; hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    // stage profiling: events[i] marks the start of stage prof_stage[i]; the event after the last stage has stage -1
    bool prof = false;
    std::vector<hipEvent_t> ev_pool; size_t ev_used = 0;
    std::vector<int> prof_stage;
};

namespace {

inline int cv_round_f(float v) { return (int)nearbyintf(v); }           // cvRound: round half to even
inline int cv_floor_f(float v) { int i = (int)v; return i - (i > v); }
inline short sat_short(float v) { int i = cv_round_f(v); return (short)(i < -32768 ? -32768 : (i > 32767 ? 32767 : i)); }

int fail(hs_orb* h, int code, const std::string& msg) { if (h) h->err = msg; return code; }

// record "stage `stage` starts here" (stage -1 closes the previous one) when profiling is on
void mark(hs_orb* h, int stage, hipStream_t s)
{
    if (!h->prof) return;
    if (h->ev_used == h->ev_pool.size()) { hipEvent_t e; if (hipEventCreate(&e) != hipSuccess) return; h->ev_pool.push_back(e); }
    (void)hipEventRecord(h->ev_pool[h->ev_used++], s);
    h->prof_stage.push_back(stage);
}

// a failed HIP call leaves its code in the runtime's sticky "last error": it is cleared here so that the next successful call sequence on this
// thread does not report it again through hipGetLastError()
#define HIP_TRY(h, expr) do { hipError_t e__ = (expr); if (e__ != hipSuccess) { (void)hipGetLastError(); \
    return fail(h, HS_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e__)); } } while (0)

void free_geometry(hs_orb* h)
{
    hipFree(h->d_pyr); h->d_pyr = nullptr;
    hipFree(h->d_tables); h->d_tables = nullptr;
    hipFree(h->d_pyr_tabs); h->d_pyr_tabs = nullptr; h->pyr_fuse.clear(); h->pyr_chain.clear(); h->pyr_deep.clear();
    hipFree(h->d_qt_tabs); h->d_qt_tabs = nullptr;
    hipFree(h->d_qkeys); h->d_qkeys = nullptr; hipFree(h->d_fast_qt); h->d_fast_qt = nullptr;
    hipFree(h->d_qhist); h->d_qhist = nullptr; hipFree(h->d_qbest); h->d_qbest = nullptr; h->qhist_stride = h->qbest_stride = 0; h->keys_dirty = false;
    hipFree(h->d_fast_items); h->d_fast_items = nullptr;
    hipFree(h->d_fast_items_n); h->d_fast_items_n = nullptr; h->fast_items_n = 0;
    hipFree(h->d_fast_ovf); h->d_fast_ovf = nullptr;
    hipFree(h->d_qt_rects); h->d_qt_rects = nullptr;
    hipFree(h->d_cand); hipFree(h->d_pts_xy); hipFree(h->d_pts_sk); hipFree(h->d_pt_node); hipFree(h->d_cell_count);
    h->d_cand = nullptr; h->d_pts_xy = h->d_pts_sk = nullptr; h->d_pt_node = nullptr; h->d_cell_count = nullptr;
    hipFree(h->d_cand_count); hipFree(h->d_sel_count); h->d_cand_count = h->d_sel_count = nullptr;
    hipFree(h->d_sel); h->d_sel = nullptr;
    hipFree(h->d_sel_perm); h->d_sel_perm = nullptr;
    // nothing is configured any more: a failed configure() must not leave a geometry that the early exit would accept
    h->w = h->h = h->batch_cap = 0; h->max_kp = 0; h->total_cells = 0; h->fast_items = 0;
}

// (Re)build per-level geometry, tables and workspace for batches of `batch` frames of w x h.
int configure_impl(hs_orb* h, int w, int hh, int batch);
int configure(hs_orb* h, int w, int hh, int batch)
{
    if (w == h->w && hh == h->h && batch <= h->batch_cap) return HS_OK;
    if (w < 1 || hh < 1 || w > 16384 || hh > 16384 || batch < 1 || batch > 65535)
        return fail(h, HS_ERR_INVALID, "image size / batch out of range");
    const int rc = configure_impl(h, w, hh, batch);
    if (rc != HS_OK) free_geometry(h);             // partial allocations of a failed attempt go away; the handle stays usable
    return rc;
}
int configure_impl(hs_orb* h, int w, int hh, int batch)
{
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    free_geometry(h);                  // also un-configures: every failure return below leaves w = h = batch_cap = 0
    const int L = h->p.nlevels;
    h->lv.assign(L, HsLevel{});
    std::vector<int16_t> tables;
    std::vector<size_t> tab_off(L * 4, 0);
    size_t pyr_per_img = 0; uint64_t cand = 0; int sel = 0, cells = 0, items = 0;
    std::vector<size_t> pyr_off(L, 0);
    for (int l = 0; l < L; l++) {
        HsLevel& V = h->lv[l];
        V.w = cv_round_f((float)w * h->inv_scale[l]);          // ORBExtractor.cpp:568-569
        V.h = cv_round_f((float)hh * h->inv_scale[l]);
        if (V.w < 1 || V.h < 1) return fail(h, HS_ERR_INVALID, "pyramid level collapses to zero size");
        V.pitch = (V.w + 63) & ~63;
        if (l > 0) { pyr_off[l] = pyr_per_img; pyr_per_img += (size_t)V.pitch * V.h; }
        // cell grid, ORBExtractor.cpp:413-428
        const int minB = HS_BORDER, maxBX = V.w - HS_BORDER, maxBY = V.h - HS_BORDER;
        const float width = (float)(maxBX - minB), height = (float)(maxBY - minB);
        const float W = (float)h->p.cell_px;
        V.ncols = width > 0 ? (int)(width / W) : 0;
        V.nrows = height > 0 ? (int)(height / W) : 0;
        if (V.ncols < 1 || V.nrows < 1) { V.ncols = V.nrows = 0; V.wcell = V.hcell = 0; }   // reference: division by zero (UB); no keypoints here
        else { V.wcell = (int)ceilf(width / V.ncols); V.hcell = (int)ceilf(height / V.nrows); }
        if (V.wcell > hs_fast_max_cell_w(6) || V.hcell > HS_MAX_CELL_H)
            return fail(h, HS_ERR_INVALID, "FAST cell wider than 247 px or taller than 125 px is not supported");
        V.cell_begin = cells; cells += V.ncols * V.nrows;
        V.grp_cells = hs_fast_group_cells(V.wcell, V.ncols, 6);
        V.ngroups = V.grp_cells > 0 ? (V.ncols + V.grp_cells - 1) / V.grp_cells : 0;
        V.item_begin = items; items += V.ngroups * V.nrows;
        V.inv_wcell = V.wcell > 0 ? (65536 + V.wcell - 1) / V.wcell : 0;
        V.inv_wcell1 = V.wcell > 0 ? (65536 + V.wcell) / (V.wcell + 1) : 0;
        // quadtree, ORBExtractor.cpp:183-185
        V.qt_w = maxBX - minB; V.qt_h = maxBY - minB;
        V.n_ini = (V.qt_w > 0 && V.qt_h > 0) ? (int)roundf((float)V.qt_w / (float)V.qt_h) : 0;
        if (V.ncols > 0 && V.n_ini < 1) return fail(h, HS_ERR_INVALID, "aspect ratio w/h < 0.5 is undefined behaviour in the reference (nIni == 0)");
        if (V.ncols < 1) V.n_ini = 0;      // a level without a FAST cell has no keypoints (D4) and no tree: its aspect ratio refuses nothing (until round 6 a 560 x 33 level 7 did: 528 roots)
        if (V.n_ini > (h->qt_large ? HS_QT_LARGE_NODES : HS_QT_MAX_NODES) / 4) return fail(h, HS_ERR_INVALID, "aspect ratio too wide");
        V.hx = V.n_ini > 0 ? (float)V.qt_w / V.n_ini : 1.f;
        V.quota = h->quota[l];
        V.cand_cap = V.ncols * V.nrows * hs_cell_cap(V.wcell, V.hcell);
        V.cand_off = cand; cand += (uint64_t)((V.cand_cap + 3) & ~3);
        V.sel_cap = std::max(V.quota + 4, 4 * V.n_ini + 4);
        V.sel_off = sel; sel += V.sel_cap;
        V.scale = h->scale[l];
        V.kp_size = (float)(int)(31 * h->scale[l]);              // ORBExtractor.cpp:478
        // cv::resize tables (OpenCV 3.4 resize.cpp, INTER_LINEAR, 8U fixed point) from level l-1
        if (l > 0) {
            const int sw = h->lv[l - 1].w, sh = h->lv[l - 1].h;
            const double scale_x = 1. / ((double)V.w / sw), scale_y = 1. / ((double)V.h / sh);
            int xmax = V.w;
            auto grow = [&](size_t n) { size_t o = (tables.size() + 3) & ~(size_t)3; tables.resize(o + n); return o; };   // 8-byte aligned
            size_t o0 = grow(4 * (size_t)V.w);   // x table: {sx, a0, a1, 0} per output column
            size_t o1 = o0;
            size_t o2 = grow(V.h);               // yofs
            size_t o3 = grow(2 * (size_t)V.h);   // ibeta
            for (int dx = 0; dx < V.w; dx++) {
                float fx = (float)((dx + 0.5) * scale_x - 0.5);
                int sx = cv_floor_f(fx); fx -= sx;
                if (sx < 0) { fx = 0; sx = 0; }
                if (sx + 1 >= sw) { xmax = std::min(xmax, dx); if (sx >= sw - 1) { fx = 0; sx = sw - 1; } }
                tables[o0 + 4 * dx] = (int16_t)sx;
                tables[o0 + 4 * dx + 1] = sat_short((1.f - fx) * 2048);
                tables[o0 + 4 * dx + 2] = sat_short(fx * 2048);
                tables[o0 + 4 * dx + 3] = 0;
            }
            for (int dy = 0; dy < V.h; dy++) {
                float fy = (float)((dy + 0.5) * scale_y - 0.5);
                int sy = cv_floor_f(fy); fy -= sy;
                tables[o2 + dy] = (int16_t)std::max(-32768, std::min(32767, sy));
                tables[o3 + 2 * dy] = sat_short((1.f - fy) * 2048);
                tables[o3 + 2 * dy + 1] = sat_short(fy * 2048);
            }
            V.xmax = xmax;
            tab_off[4 * l] = o0; tab_off[4 * l + 1] = o1; tab_off[4 * l + 2] = o2; tab_off[4 * l + 3] = o3;
        }
    }
    pyr_per_img = (pyr_per_img + 255) & ~(size_t)255;
    h->max_wcell = h->max_hcell = 1;
    for (int l = 0; l < L; l++) { h->max_wcell = std::max(h->max_wcell, h->lv[l].wcell); h->max_hcell = std::max(h->max_hcell, h->lv[l].hcell); }
    h->total_cells = cells; h->fast_items = items; h->cand_img_stride = cand; h->sel_img_stride = sel; h->max_kp = sel;
    // Order of the FAST work items of an image: the REDUCED levels first, deepest level first, level 0 last.  An item of a reduced level
    // costs 2-3 times an item of level 0 (the same number of pixels, denser corners), and the persistent FAST kernel walks the items in this
    // order: with the cheap, uniform level-0 items at the end of every queue the tail of the launch — waves finishing their last item while
    // the queues are empty — is short (a 32-frame launch spent ~17 % more per frame than a 128-frame launch with the expensive items last).
    auto order_items = [&](std::vector<HsLevel>& lv) {
        int pos = 0;
        if (h->fast_order == 0) { for (int l = 0; l < L; l++) { lv[l].item_begin = pos; pos += lv[l].ngroups * lv[l].nrows; } }
        else {
            for (int l = L - 1; l >= 1; l--) { lv[l].item_begin = pos; pos += lv[l].ngroups * lv[l].nrows; }
            lv[0].item_begin = pos; pos += lv[0].ngroups * lv[0].nrows;
        }
        return pos;
    };
    order_items(h->lv);

    HIP_TRY(h, hipMalloc(&h->d_pyr, std::max<size_t>(pyr_per_img * batch, 256)));
    HIP_TRY(h, hipMalloc(&h->d_tables, std::max<size_t>(tables.size() * sizeof(int16_t), 256)));
    if (!tables.empty()) HIP_TRY(h, hipMemcpy(h->d_tables, tables.data(), tables.size() * sizeof(int16_t), hipMemcpyHostToDevice));
    const size_t ce = std::max<uint64_t>(cand * batch, 64);
    HIP_TRY(h, hipMalloc(&h->d_cand, ce * 8));
    HIP_TRY(h, hipMalloc(&h->d_pts_xy, ce * 4));
    HIP_TRY(h, hipMalloc(&h->d_pts_sk, ce * 4));
    HIP_TRY(h, hipMalloc(&h->d_pt_node, ce * 2));
    if (h->qt_large) HIP_TRY(h, hipMalloc(&h->d_qt_rects, (size_t)batch * L * hs_quadtree_large_scratch_bytes()));
    HIP_TRY(h, hipMalloc(&h->d_cell_count, std::max<size_t>((size_t)cells * batch * 4, 64)));
    HIP_TRY(h, hipMalloc(&h->d_cand_count, (size_t)batch * L * 4));
    HIP_TRY(h, hipMalloc(&h->d_sel_count, (size_t)batch * L * 4));
    HIP_TRY(h, hipMalloc(&h->d_sel, std::max<size_t>((size_t)sel * batch * 12, 64)));
    HIP_TRY(h, hipMalloc(&h->d_sel_perm, std::max<size_t>((size_t)sel * batch * 2, 64)));
    for (int l = 0; l < L; l++) {
        HsLevel& V = h->lv[l];
        V.img_stride = pyr_per_img;
        V.base = l > 0 ? h->d_pyr + pyr_off[l] : nullptr;
        if (l > 0) {
            V.xofs = h->d_tables + tab_off[4 * l]; V.ialpha = h->d_tables + tab_off[4 * l + 1];
            V.yofs = h->d_tables + tab_off[4 * l + 2]; V.ibeta = h->d_tables + tab_off[4 * l + 3];
        }
    }
    {   // geometric-key tables of the quadtree kernel
        std::vector<uint8_t> qblob;
        std::vector<size_t> xo(L, 0), yo2(L, 0); std::vector<char> has(L, 0);
        for (int l = 0; l < L; l++) has[l] = hs_quadtree_build_tables(h->lv[l], qblob, xo[l], yo2[l]) ? 1 : 0;
        HIP_TRY(h, hipMalloc(&h->d_qt_tabs, std::max<size_t>(qblob.size() + 16, 256)));
        if (!qblob.empty()) HIP_TRY(h, hipMemcpy(h->d_qt_tabs, qblob.data(), qblob.size(), hipMemcpyHostToDevice));
        for (int l = 0; l < L; l++) if (has[l]) { h->lv[l].qt_xtab = h->d_qt_tabs + xo[l]; h->lv[l].qt_ytab = h->d_qt_tabs + yo2[l]; }
        // the FAST kernel's side of the same keys (HsFastQt): u16 tables  xkey[x] = root(x) << 2 DH | spread(xtab[x]),  ykey[y] = spread(ytab[y]) << 1,
        // padded by 512 entries, and this level's place in the per-image histogram / best-candidate arrays
        auto spread = [](uint32_t v) { v = (v | (v << 4)) & 0x0F0Fu; v = (v | (v << 2)) & 0x3333u; v = (v | (v << 1)) & 0x5555u; return v; };
        std::vector<uint16_t> keys; std::vector<HsFastQt> fq(L, HsFastQt{});
        std::vector<size_t> kx(L, 0), ky(L, 0);
        uint32_t hoff = 0, boff = 0;
        for (int l = 0; l < L; l++) {
            HsLevel& V = h->lv[l];
            V.qt_hist_off = V.qt_best_off = 0xFFFFFFFFu;
            // (HS_FAST_KEYS_LEVELS: only levels 0 .. n-1.  Measured at one 1080p pair per call, quadtree us for n = 0 / 1 / 2 / 3 / 8: 35.1 / 32.6 /
            // 31.4 / 30.8 / 24.8 — every level's workgroup is about as long as level 0's, the fixed block-wide steps dominate — so it is all or nothing.)
            if (!has[l] || !h->fast_keys || l >= h->fast_keys_levels) continue;
            const int DH = V.n_ini <= 2 ? 6 : 5, ncell = V.n_ini << (2 * DH);
            kx[l] = keys.size(); keys.resize(keys.size() + (size_t)V.qt_w + 1 + 512, 0);
            for (int x = 0; x <= V.qt_w; x++) {
                int r = 0;
                for (int i = 1; i < V.n_ini; i++) r += x >= V.qt_rbound[i];
                keys[kx[l] + x] = (uint16_t)(((uint32_t)r << (2 * DH)) | spread(qblob[xo[l] + x]));
            }
            ky[l] = keys.size(); keys.resize(keys.size() + (size_t)V.qt_h + 1 + 512, 0);
            for (int y = 0; y <= V.qt_h; y++) keys[ky[l] + y] = (uint16_t)(spread(qblob[yo2[l] + y]) << 1);
            V.qt_hist_off = hoff; V.qt_best_off = boff;
            hoff += (uint32_t)(ncell / 2); boff += (uint32_t)ncell;
            fq[l].hist_off = V.qt_hist_off; fq[l].best_off = V.qt_best_off; fq[l].enabled = 1;
        }
        h->qhist_stride = hoff; h->qbest_stride = boff;
        HIP_TRY(h, hipMalloc(&h->d_qkeys, std::max<size_t>(keys.size() * 2 + 16, 256)));
        if (!keys.empty()) HIP_TRY(h, hipMemcpy(h->d_qkeys, keys.data(), keys.size() * 2, hipMemcpyHostToDevice));
        for (int l = 0; l < L; l++) if (fq[l].enabled) { fq[l].xkey = h->d_qkeys + kx[l]; fq[l].ykey = h->d_qkeys + ky[l]; }
        HIP_TRY(h, hipMalloc(&h->d_fast_qt, sizeof(HsFastQt) * HS_MAX_LEVELS));
        HIP_TRY(h, hipMemcpy(h->d_fast_qt, fq.data(), sizeof(HsFastQt) * L, hipMemcpyHostToDevice));
        HIP_TRY(h, hipMalloc(&h->d_qhist, std::max<size_t>((size_t)hoff * batch * 4, 256)));
        HIP_TRY(h, hipMalloc(&h->d_qbest, std::max<size_t>((size_t)boff * batch * 8, 256)));
        // (on the handle's stream: hipMemset on device memory is asynchronous to the host and runs on the NULL stream, which the handle's non-blocking
        //  streams are not ordered with — a memset that lands after the first kernels would wipe what they wrote)
        HIP_TRY(h, hipMemsetAsync(h->d_qhist, 0, std::max<size_t>((size_t)hoff * batch * 4, 256), h->stream));
        HIP_TRY(h, hipMemsetAsync(h->d_qbest, 0, std::max<size_t>((size_t)boff * batch * 8, 256), h->stream));
    }
    {   // which level pairs the fused pyramid kernel can produce (decided on the host copies of the tables)
        std::vector<const int16_t*> xt(L, nullptr), yo(L, nullptr);
        for (int l = 1; l < L; l++) { xt[l] = tables.data() + tab_off[4 * l]; yo[l] = tables.data() + tab_off[4 * l + 2]; }
        hs_pyramid_plan_fusion(h->lv.data(), L, xt.data(), yo.data(), h->pyr_tbx_max);
        if (h->no_fuse) for (int l = 0; l < L; l++) h->lv[l].fuse_tbx = 0;
        std::vector<const int16_t*> ib(L, nullptr);
        for (int l = 1; l < L; l++) ib[l] = tables.data() + tab_off[4 * l + 3];
        std::vector<uint64_t> blob;
        hs_pyramid_build_tables(h->lv.data(), L, xt.data(), yo.data(), ib.data(), blob, h->pyr_fuse);
        // chains: the last three levels in one launch when the number of levels to make is odd (8 levels: (1,2) (3,4) (5,6,7))
        h->pyr_chain.assign(L, HsPyrChain{});
        for (int l = 0; l < L; l++) h->lv[l].chain_n = 0;
        if (!h->no_fuse && h->chain_mode != 0) {
            if (h->chain_mode == 2) {
                for (int l = 1; l + 1 < L; l += 2) {
                    int n = (l + 3 == L) ? 3 : 2;
                    hs_pyramid_plan_chain(h->lv.data(), l, n, xt.data(), yo.data(), ib.data(), blob, h->pyr_chain[l]);
                    if (!h->pyr_chain[l].valid && n == 3) hs_pyramid_plan_chain(h->lv.data(), l, 2, xt.data(), yo.data(), ib.data(), blob, h->pyr_chain[l]);
                    if (h->pyr_chain[l].valid) { h->lv[l].chain_n = h->pyr_chain[l].nstage; if (h->pyr_chain[l].nstage == 3) l++; }
                }
            } else if (const char* plan = getenv("HS_PYRAMID_PLAN")) {      // tuning knob: explicit chain lengths from level 1, e.g. "2,3,2" (1 = a single level, 2 = the two-level kernel unless HS_PYRAMID_CHAIN2=1)
                const bool chain2 = getenv("HS_PYRAMID_CHAIN2") && atoi(getenv("HS_PYRAMID_CHAIN2")) != 0;
                int l = 1;
                for (const char* q = plan; *q && l < L; ) {
                    const int n = std::min(atoi(q), L - l);
                    if (n >= 3 || (n == 2 && (chain2 || !(l & 1)))) {       // (the two-level kernel is planned for pairs that start on an odd level)
                        hs_pyramid_plan_chain(h->lv.data(), l, n, xt.data(), yo.data(), ib.data(), blob, h->pyr_chain[l], HS_PYR_DEEP_LDS, 0);
                        if (h->pyr_chain[l].valid) h->lv[l].chain_n = n;
                    }
                    if (n == 1) h->lv[l].fuse_tbx = 0;
                    l += std::max(n, 1);
                    while (*q && *q != ',') q++;
                    if (*q == ',') q++;
                }
            } else if (L >= 4 && ((L - 1) & 1)) {
                const int l = L - 3;
                hs_pyramid_plan_chain(h->lv.data(), l, 3, xt.data(), yo.data(), ib.data(), blob, h->pyr_chain[l]);
                if (h->pyr_chain[l].valid) h->lv[l].chain_n = 3;
            }
        }
        // the small-batch plan: a launch of few frames lasts as long as one workgroup lives and costs ~5 us whatever it does, so the dependent
        // launches are what counts — greedy: from level 1, the longest chain that fits HS_PYR_DEEP_LDS, then the next (1080p: ONE launch for levels 1-7)
        h->pyr_deep.assign(L, HsPyrChain{});
        if (!h->no_fuse && h->deep_max_batch > 0) {
            for (int l = 1; l + 1 < L;) {
                int took = 0;
                for (int n = std::min(HS_PYR_CHAIN_MAX, L - l); n >= 2 && !took; n--) {
                    hs_pyramid_plan_chain(h->lv.data(), l, n, xt.data(), yo.data(), ib.data(), blob, h->pyr_deep[l], HS_PYR_DEEP_LDS, h->deep_rows);
                    if (h->pyr_deep[l].valid) took = n;
                }
                l += took ? took : 1;
            }
        }
        HIP_TRY(h, hipMalloc(&h->d_pyr_tabs, std::max<size_t>(blob.size() * 8, 256)));
        if (!blob.empty()) HIP_TRY(h, hipMemcpy(h->d_pyr_tabs, blob.data(), blob.size() * 8, hipMemcpyHostToDevice));
        for (int which = 0; which < 2; which++)
        for (HsPyrChain& C : (which ? h->pyr_deep : h->pyr_chain)) {                  // blob offsets -> device pointers
            if (!C.valid) continue;
            for (int i = 0; i < C.nstage; i++) {
                HsPyrStage& S = C.st[i];
                S.rows = reinterpret_cast<const HsPyrRow*>(h->d_pyr_tabs + (uintptr_t)S.rows);
                S.tx = reinterpret_cast<const HsPyrStageX*>(h->d_pyr_tabs + (uintptr_t)S.tx); S.ty = reinterpret_cast<const HsPyrStageY*>(h->d_pyr_tabs + (uintptr_t)S.ty);
            }
        }
        for (HsPyrFuse& F : h->pyr_fuse) {                    // blob offsets -> device pointers
            if (!F.valid) continue;
            F.rowA = reinterpret_cast<const HsPyrRow*>(h->d_pyr_tabs + (uintptr_t)F.rowA); F.rowB = reinterpret_cast<const HsPyrRow*>(h->d_pyr_tabs + (uintptr_t)F.rowB);
            F.xt = reinterpret_cast<const HsPyrXTile*>(h->d_pyr_tabs + (uintptr_t)F.xt); F.yt = reinterpret_cast<const HsPyrYTile*>(h->d_pyr_tabs + (uintptr_t)F.yt);
        }
    }
    // the same levels with NARROW work items (tiles of 32 dwords: <= 119 px of interior per item), for the launches of small batches: the list
    // differs in the grouping of the cells only, so everything downstream of the FAST kernel (the quadtree's gather) reads the grouping it was
    // launched with from ITS copy of the level array (d_lv + L)
    h->lv_n = h->lv;
    bool narrow_ok = h->fast_knobs.cols != 64;
    for (int l = 0; l < L; l++) if (h->lv[l].wcell > hs_fast_max_cell_w(5)) narrow_ok = false;
    if (narrow_ok) {
        for (int l = 0; l < L; l++) {
            HsLevel& V = h->lv_n[l];
            V.grp_cells = hs_fast_group_cells(V.wcell, V.ncols, 5);
            V.ngroups = V.grp_cells > 0 ? (V.ncols + V.grp_cells - 1) / V.grp_cells : 0;
        }
        h->fast_items_n = order_items(h->lv_n);
    }
    HIP_TRY(h, hipMemcpy(h->d_lv, h->lv.data(), sizeof(HsLevel) * L, hipMemcpyHostToDevice));
    HIP_TRY(h, hipMemcpy(h->d_lv + L, h->lv_n.data(), sizeof(HsLevel) * L, hipMemcpyHostToDevice));
    {
        bool items_fit = true;                                 // every item against the tile the kernel stages it into (hs_fast_item_fits: columns, score-tile column, cells)
        auto upload_items = [&](const std::vector<HsLevel>& lv, int n_items, HsFastItem** d_out, int lc) -> hipError_t {
            std::vector<HsFastItem> fi(std::max(n_items, 1));
            hs_fast_build_items(lv.data(), L, fi.data());
            for (int i = 0; i < n_items; i++) items_fit = items_fit && hs_fast_item_fits(fi[i], lc);
            if (h->fast_order == 2 && L > 2) {                 // experiment: the reduced levels interleaved in proportion (every stretch of the list has the same mix of levels), level 0 last
                const int n_red = lv[0].item_begin;
                std::vector<std::pair<double, int>> key(n_red);
                for (int l = 1; l < L; l++) {
                    const int n = lv[l].ngroups * lv[l].nrows;
                    for (int k = 0; k < n; k++) key[lv[l].item_begin + k] = { (k + 0.5) / n, lv[l].item_begin + k };
                }
                std::stable_sort(key.begin(), key.end(), [](const std::pair<double, int>& a, const std::pair<double, int>& b) { return a.first < b.first; });
                std::vector<HsFastItem> t(fi.begin(), fi.begin() + n_red);
                for (int i = 0; i < n_red; i++) fi[i] = t[key[i].second];
            }
            hipError_t e = hipMalloc(d_out, fi.size() * sizeof(HsFastItem));
            if (e != hipSuccess) return e;
            return hipMemcpy(*d_out, fi.data(), fi.size() * sizeof(HsFastItem), hipMemcpyHostToDevice);
        };
        HIP_TRY(h, upload_items(h->lv, items, &h->d_fast_items, 6));
        if (h->fast_items_n > 0) HIP_TRY(h, upload_items(h->lv_n, h->fast_items_n, &h->d_fast_items_n, 5));
        if (!items_fit) return fail(h, HS_ERR_INVALID, "internal: a FAST work item does not fit its tile (hs_fast_group_cells); geometry refused");
        HIP_TRY(h, hipMalloc(&h->d_fast_ovf, std::max<size_t>(hs_fast_overflow_bytes(h->max_hcell, std::max(items, h->fast_items_n) * batch, h->fast_knobs), 256)));
        HIP_TRY(h, hipMemsetAsync(h->d_fast_ovf, 0, 4 * HS_FAST_QUEUE_DWORDS * 4, h->stream));       // all four work-queue counter sets start at zero (stream-ordered before the first launch)
    }
    HIP_TRY(h, hipStreamSynchronize(h->stream));     // the memsets above have landed whatever stream the caller's launches will use (configuration is rare: it allocates)
    h->w = w; h->h = hh; h->batch_cap = batch;      // configured only now
    return HS_OK;
}

inline size_t pad256(size_t b) { return (b + 255) & ~(size_t)255; }

int ensure_pinned(hs_orb* h, uint8_t** p, size_t* have, size_t need)
{
    if (need <= *have) return HS_OK;
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    if (*p) hipHostFree(*p);
    *p = nullptr; *have = 0;
    HIP_TRY(h, hipHostMalloc((void**)p, need, hipHostMallocDefault));
    *have = need;
    return HS_OK;
}

int ensure_outputs(hs_orb* h, int batch, int cap)
{
    if (batch <= h->out_batch && cap == h->out_cap) return HS_OK;
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    hipFree(h->d_n);                     // one block: [counts | keypoints | descriptors], so that the results come back in ONE device-to-host copy
    batch = std::max(batch, h->out_batch);
    h->d_kps = nullptr; h->d_desc = nullptr; h->d_n = nullptr; h->out_batch = 0; h->out_cap = 0;
    const size_t nb = pad256((size_t)batch * 4), kb = pad256((size_t)batch * cap * sizeof(hs_keypoint)), db = (size_t)batch * cap * HS_DESC_BYTES;
    uint8_t* blk = nullptr;
    HIP_TRY(h, hipMalloc(&blk, nb + kb + db));
    h->d_n = reinterpret_cast<int32_t*>(blk); h->d_kps = reinterpret_cast<hs_keypoint*>(blk + nb); h->d_desc = blk + nb + kb;
    h->out_batch = batch; h->out_cap = cap;
    return HS_OK;
}

int ensure_stereo_strips(hs_orb* h, int pairs, int cap, int n_rows)
{
    // two capacities: the counters (pairs * strips) and the lists (pairs * strips * cap) grow independently
    if (n_rows > 65536) return fail(h, HS_ERR_INVALID, "stereo: more than 65536 image rows");      // k_stereo_strips keeps one counter per 32 rows in LDS
    const size_t need_count = (size_t)pairs * hs_stereo_strips(n_rows), need_list = need_count * (size_t)cap;
    if (need_count <= h->strip_count_entries && need_list <= h->strip_list_entries) return HS_OK;
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    if (need_count > h->strip_count_entries) {
        hipFree(h->d_strip_count); h->d_strip_count = nullptr; h->strip_count_entries = 0;
        HIP_TRY(h, hipMalloc(&h->d_strip_count, need_count * 4));
        h->strip_count_entries = need_count;
    }
    if (need_list > h->strip_list_entries) {
        hipFree(h->d_strip_list); h->d_strip_list = nullptr; h->strip_list_entries = 0;
        HIP_TRY(h, hipMalloc(&h->d_strip_list, need_list * HS_STRIP_ENTRY_BYTES));
        h->strip_list_entries = need_list;
    }
    return HS_OK;
}

int ensure_stereo_scratch(hs_orb* h, size_t entries)
{
    if (entries <= h->st_entries) return HS_OK;
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    hipFree(h->d_ur); hipFree(h->d_depth); hipFree(h->d_bd);
    h->d_ur = h->d_depth = nullptr; h->d_bd = nullptr;
    HIP_TRY(h, hipMalloc(&h->d_ur, entries * 4));
    HIP_TRY(h, hipMalloc(&h->d_depth, entries * 4));
    HIP_TRY(h, hipMalloc(&h->d_bd, entries * 4));
    h->st_entries = entries;
    return HS_OK;
}

// `sf`: the stereo front end — the describe launch also bins the right images' keypoints into the matcher's strips (HsStripFuse, kernels_describe.hip)
int run_extract(hs_orb* h, HsImg0 img0, int batch, HsOut out, hipStream_t s, const HsStripFuse* sf = nullptr)
{
    const int L = h->p.nlevels;
    // Level 0 needs no pyramid.  For one or two LARGE frames the chain of launches is latency-bound (dependent pyramid launches, a FAST launch
    // whose duration is its slowest work item, the level-0 quadtree workgroup — 0.11 ms for a 4000 x 3000 frame): level 0's FAST + quadtree run
    // on a second stream BESIDE the pyramid and the other levels' FAST + quadtree, joined before the describe stage.  Same kernels, same
    // results.  Measured (profiles/README.md row "C4", profiles/r03_bench_lines.json -> c4; the same figures as include/hyslam_amd.h quotes for
    // hs_orb_set_split): the 4000 x 3000 "Imaging" extraction 0.413 ms unsplit -> 0.207 ms split (config C4: 2 592 -> 4 068 steps/s with both cameras
    // split); a 1080p pair gets SLOWER (0.132 -> 0.160 ms: the fork / join between the streams costs more than the overlap saves), 16 pairs too
    // (-7 %), hence the size rule.  Not with stage events (they would serialise the two sequences).  The two concurrent FAST launches rely on every
    // earlier launch of the handle having completed: a handle is driven from ONE stream at a time (include/hyslam_amd.h).
    // Item width by batch (round 4): a launch of few frames lasts as long as its slowest wave (one stereo pair: 1 904 wide items for 2 816
    // resident single-wave workgroups), so it gets the NARROW items — twice as many, half as long; measured at 1080p (pairs per call: narrow / wide pairs/s): 1: 9 949 / 8 403,
    // 2: 16 029 / 15 642, 4: 23 004 / 22 660, 8: 32 657 / 33 240, 16: 40 917 / 41 507 — from ~18 k items on the wide ones win (fewer, fuller tiles).  HS_FAST_COLS = 32 / 64 forces one list, HS_FAST_NARROW_MAX moves the threshold.
    const int narrow_max = h->fast_knobs.narrow_max > 0 ? h->fast_knobs.narrow_max : 18000;
    const bool narrow = h->fast_items_n > 0 && (h->fast_knobs.cols == 32 || (h->fast_knobs.cols != 64 && (long long)h->fast_items_n * batch <= narrow_max));
    // Keys by batch as well: the FAST kernel's two global atomics per candidate cost it 5 % at 16 pairs per call (0.162 -> 0.170 ms) and buy the
    // quadtree launch 4 us there (its 256 workgroups fill the chip either way); at one pair per call they cost 1 us and buy 10 (35.1 -> 24.8 us: the
    // level-0 workgroup no longer gathers 5 000 records on one CU).  Both arrays are zero between calls whatever the mode, so the mode may change per call.
    const HsPyrChain* const deep = (batch <= h->deep_max_batch && !h->pyr_deep.empty()) ? h->pyr_deep.data() : nullptr;      // the pyramid's small-batch plan
    const bool use_keys = h->fast_keys && h->d_fast_qt != nullptr && batch <= h->fast_keys_max_batch;
    // d_qhist / d_qbest are all zero between calls (the quadtree kernel zeroes what it consumes).  A call that fails anywhere between the keyed FAST
    // launch and its end — an event / stream call of the split path, a later launch — breaks that: the flag makes the NEXT call start from zeroed arrays.
    if (h->keys_dirty) {
        HIP_TRY(h, hipDeviceSynchronize());
        if (h->d_qhist) HIP_TRY(h, hipMemsetAsync(h->d_qhist, 0, (size_t)h->qhist_stride * h->batch_cap * 4, s));
        if (h->d_qbest) HIP_TRY(h, hipMemsetAsync(h->d_qbest, 0, (size_t)h->qbest_stride * h->batch_cap * 8, s));
        HIP_TRY(h, hipStreamSynchronize(s));
        h->keys_dirty = false;
    }
    if (use_keys) h->keys_dirty = true;                          // cleared at the end of a call that enqueued everything without error
    const std::vector<HsLevel>& lvh = narrow ? h->lv_n : h->lv;
    const HsLevel* const d_lv = h->d_lv + (narrow ? L : 0);
    const HsFastItem* const d_items = narrow ? h->d_fast_items_n : h->d_fast_items;
    const int n_items = narrow ? h->fast_items_n : h->fast_items, lc = narrow ? 5 : 6;
    const int items0 = lvh[0].ngroups * lvh[0].nrows, first0 = lvh[0].item_begin;      // work items of level 0: the LAST items0 of the item list
    const bool split = !h->prof && L > 1 && items0 > 0 && items0 < n_items && (h->split_mode == 1 || (h->split_mode < 0 && batch <= 2 && (size_t)h->w * (size_t)h->h * (size_t)batch >= 6000000));
    auto fast = [&](int item_first, int item_count, int spill_slot, hipStream_t st) -> int {
        // launch N uses work-queue counter set N & 3 and relies on launch N - 2 having zeroed it: the epoch advances only when a launch was
        // enqueued without error; after a failed launch all sets are zeroed again so that the next one starts from a known state
        const bool launched = hs_launch_fast(d_lv, d_items, L, img0, batch, h->total_cells, n_items, h->p.fast_threshold,
                                             h->d_cand, h->d_cell_count, h->cand_img_stride, h->max_wcell, h->max_hcell, h->d_fast_ovf, h->fast_epoch, h->fast_knobs,
                                             item_first, item_count, spill_slot, lc, use_keys ? h->d_fast_qt : nullptr, h->d_qhist, h->d_qbest, h->qhist_stride, h->qbest_stride, st);
        const hipError_t e = hipGetLastError();
        if (e != hipSuccess) {
            (void)hipDeviceSynchronize();
            (void)hipMemsetAsync(h->d_fast_ovf, 0, 4 * HS_FAST_QUEUE_DWORDS * 4, h->stream);
            if (h->d_qhist) (void)hipMemsetAsync(h->d_qhist, 0, (size_t)h->qhist_stride * h->batch_cap * 4, h->stream);      // a launch that died half-way may have left keys behind
            if (h->d_qbest) (void)hipMemsetAsync(h->d_qbest, 0, (size_t)h->qbest_stride * h->batch_cap * 8, h->stream);
            (void)hipStreamSynchronize(h->stream);
            return fail(h, HS_ERR_HIP, std::string("FAST launch: ") + hipGetErrorString(e));
        }
        if (launched) h->fast_epoch++;
        return HS_OK;
    };
    auto quadtree = [&](int level_first, int level_count, hipStream_t st) {
        hs_launch_quadtree(d_lv, L, batch, h->total_cells, h->d_cand, h->d_cell_count, h->cand_img_stride,
                           h->d_pts_xy, h->d_pts_sk, h->d_pt_node, h->d_cand_count, h->d_sel, h->d_sel_count, h->sel_img_stride, h->d_sel_perm, h->qt_point_domain ? 1 : 0,
                           level_first, level_count, use_keys ? h->d_qhist : nullptr, h->d_qbest, h->qhist_stride, h->qbest_stride, h->keep_points ? 1 : 0,
                           h->qt_large ? 2 : (h->qt_small_ok ? 1 : 0), h->d_qt_rects, st);
    };
    if (split) {
        if (!h->s_aux) {
            HIP_TRY(h, hipStreamCreateWithFlags(&h->s_aux, hipStreamNonBlocking));
            HIP_TRY(h, hipEventCreateWithFlags(&h->ev_sfork, hipEventDisableTiming));
            HIP_TRY(h, hipEventCreateWithFlags(&h->ev_sjoin, hipEventDisableTiming));
        }
        HIP_TRY(h, hipEventRecord(h->ev_sfork, s));                           // everything enqueued on s so far (the frames' upload, the previous call) comes first
        HIP_TRY(h, hipStreamWaitEvent(h->s_aux, h->ev_sfork, 0));
        int rc = fast(first0, items0, 1, h->s_aux);
        if (rc != HS_OK) return rc;
        quadtree(0, 1, h->s_aux);
        HIP_TRY(h, hipEventRecord(h->ev_sjoin, h->s_aux));
        h->last_pyr_launches = hs_launch_pyramid(h->d_lv, h->lv.data(), h->pyr_fuse.data(), h->pyr_chain.data(), L, img0, batch, s, deep);
        rc = fast(0, n_items - items0, 0, s);
        if (rc != HS_OK) return rc;
        quadtree(1, L - 1, s);
        HIP_TRY(h, hipStreamWaitEvent(s, h->ev_sjoin, 0));
    } else {
        mark(h, 0, s);
        h->last_pyr_launches = hs_launch_pyramid(h->d_lv, h->lv.data(), h->pyr_fuse.data(), h->pyr_chain.data(), L, img0, batch, s, deep);
        mark(h, 1, s);
        const int rc = fast(0, n_items, 0, s);
        if (rc != HS_OK) return rc;
        mark(h, 2, s);
        quadtree(0, L, s);
    }
    mark(h, 3, s);
    hs_launch_describe(h->d_lv, L, img0, batch, h->d_sel, h->d_sel_count, h->d_sel_perm, h->sel_img_stride, h->max_kp,
                       h->d_taps, out, s, h->fast_taps, sf ? *sf : HsStripFuse{});
    mark(h, -1, s);
    HIP_TRY(h, hipGetLastError());
    h->keys_dirty = false;
    h->last_batch = batch; h->last_img0 = img0;
    return HS_OK;
}

// grow-only device scratch, carved in 256-byte aligned pieces; scratch_begin() invalidates earlier pieces
int scratch_begin(hs_orb* h, size_t total)
{
    total += 4096;
    if (total > h->scratch_bytes) {
        HIP_TRY(h, hipStreamSynchronize(h->stream));
        hipFree(h->d_scratch); h->d_scratch = nullptr; h->scratch_bytes = 0;
        HIP_TRY(h, hipMalloc(&h->d_scratch, total));
        h->scratch_bytes = total;
    }
    h->scratch_used = 0;
    return HS_OK;
}
template <class T> T* carve(hs_orb* h, size_t count)
{
    T* p = reinterpret_cast<T*>(h->d_scratch + h->scratch_used);
    h->scratch_used += (count * sizeof(T) + 255) & ~(size_t)255;
    return p;
}

void run_stereo(hs_orb* h, const hs_keypoint* kL, const uint8_t* dL, const int32_t* nL, const hs_keypoint* kR, const uint8_t* dR,
                const int32_t* nR, int pairs, int cap, const hs_stereo_params& sp, float* ur, float* depth, hipStream_t s)
{
    mark(h, 4, s);
    h->last_stereo_launches = 2;
    hs_launch_stereo(kL, dL, nL, kR, dR, nR, pairs, cap, sp, ur, depth, h->d_bd, h->d_strip_count, h->d_strip_list, s);
    mark(h, 5, s);
    hs_launch_stereo_median(nL, pairs, cap, ur, depth, h->d_bd, h->d_strip_count, sp.n_rows, s);
    mark(h, -1, s);
}

// the stereo front end's matcher: the strips were binned by an extra workgroup of the describe launch (HsStripFuse): two launches where
// run_stereo needs three.  (Round 4 also built the median rejection into the matcher — each pair's last workgroup, found with a ticket counter,
// the three result arrays written through with agent-scope atomic stores so that no L2 write-back is needed: bit-exact, but the write-through
// stores cost more than the launch they save: 0.056 against 0.022 + 0.006 ms per 16 pairs, 10.0 against 4.8 + 4.8 us for one pair.  Dropped.)
void run_stereo_fused(hs_orb* h, const hs_keypoint* kL, const uint8_t* dL, const int32_t* nL, const hs_keypoint* kR, const uint8_t* dR,
                      const int32_t* nR, int pairs, int cap, const hs_stereo_params& sp, float* ur, float* depth, hipStream_t s)
{
    mark(h, 4, s);
    h->last_stereo_launches = 1;
    hs_launch_stereo_match_only(kL, dL, nL, kR, dR, nR, pairs, cap, sp, ur, depth, h->d_bd, h->d_strip_count, h->d_strip_list, s);
    mark(h, 5, s);
    hs_launch_stereo_median(nL, pairs, cap, ur, depth, h->d_bd, h->d_strip_count, sp.n_rows, s);
    mark(h, -1, s);
}

} // namespace

// host-side facts of the current configuration (tests: tests/cpp/host_sanitize, tools): launches of the pyramid's standard / small-batch plan,
// FAST work items per image (wide / narrow), levels with quadtree keys, longest deep chain, its LDS bytes
void hs_debug_plan_summary(const hs_orb* h, int32_t* out /*[8]*/)
{
    for (int i = 0; i < 8; i++) out[i] = 0;
    if (!h || h->lv.empty()) return;
    const int L = h->p.nlevels;
    out[0] = hs_pyramid_launch_count(h->lv.data(), L);
    int deep_launches = 0, longest = 0, lds = 0;
    for (int l = 1; l < L; l++) {
        deep_launches++;
        if (l < (int)h->pyr_deep.size() && h->pyr_deep[l].valid) {
            const HsPyrChain& C = h->pyr_deep[l];
            if (C.nstage > longest) { longest = C.nstage; lds = C.x_bytes + C.h_rows * 512; }
            l += C.nstage - 1;
        } else if (h->lv[l].chain_n > 0 && l + h->lv[l].chain_n <= L) l += h->lv[l].chain_n - 1;
        else if (h->lv[l].fuse_tbx > 0 && l + 1 < L) l++;
    }
    out[1] = L > 1 ? deep_launches : 0; out[2] = h->fast_items; out[3] = h->fast_items_n;
    for (int l = 0; l < L; l++) out[4] += h->lv[l].qt_hist_off != 0xFFFFFFFFu;
    out[5] = longest; out[6] = lds;
    if (L > 1 && !h->pyr_deep.empty() && h->pyr_deep[1].valid) out[7] = h->pyr_deep[1].grid_x * h->pyr_deep[1].grid_y;
}

// accessors for the other translation units of the library (kernels_bow.hip)
void hs_set_error(hs_orb* h, const char* msg) { if (h) h->err = msg ? msg : ""; }
int hs_orb_device_of(const hs_orb* h) { return h ? h->device : 0; }
hipStream_t hs_orb_stream_of(const hs_orb* h) { return h ? h->stream : nullptr; }

extern "C" {

const char* hs_version(void) { return HS_VERSION; }

const char* hs_status_string(int s)
{
    switch (s) {
    case HS_OK: return "ok";
    case HS_ERR_INVALID: return "invalid argument";
    case HS_ERR_HIP: return "HIP runtime error";
    case HS_ERR_CAPACITY: return "output capacity too small";
    case HS_ERR_NO_DEVICE: return "no usable device";
    default: return "unknown status";
    }
}

int hs_device_count(int* count)
{
    if (!count) return HS_ERR_INVALID;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) { *count = 0; return HS_ERR_NO_DEVICE; }
    *count = n;
    return HS_OK;
}

void hs_orb_default_params(hs_orb_params* p)
{
    if (!p) return;
    memset(p, 0, sizeof(*p));
    p->nfeatures = 1000; p->scale_factor = 1.2f; p->nlevels = 8; p->cell_px = 30;
    p->ini_th_fast = 20; p->min_th_fast = 4;
    p->fast_threshold = 20;      // ORBFinder's in-class default; setThreshold() never changes it (ORBFinder.cpp:58-60)
    static const uint16_t t[7] = { 18, 34, 49, 55, 49, 34, 18 };
    memcpy(p->blur_taps, t, sizeof(t));
}

int hs_orb_create(const hs_orb_params* p, int device, hs_orb** out)
{
    if (!p || !out) return HS_ERR_INVALID;
    *out = nullptr;
    if (p->nlevels < 1 || p->nlevels > HS_MAX_LEVELS || p->nfeatures < 1 || !(p->scale_factor > 1.0f) || p->cell_px < 8 ||
        p->fast_threshold < 0 || p->fast_threshold > 255)
        return HS_ERR_INVALID;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1 || device < 0 || device >= ndev) return HS_ERR_NO_DEVICE;
    hs_orb* h = new hs_orb();
    h->p = *p; h->device = device;
    h->fast_knobs = hs_fast_read_knobs();
    { const char* e = getenv("HS_PYRAMID_NO_FUSE"); h->no_fuse = e && atoi(e) != 0; }
    { const char* e = getenv("HS_STEREO_FUSE"); h->stereo_fuse = !(e && atoi(e) == 0); }
    { const char* e = getenv("HS_FAST_KEYS"); h->fast_keys = !(e && atoi(e) == 0); }
    { const char* e = getenv("HS_FAST_KEYS_LEVELS"); if (e && atoi(e) > 0) h->fast_keys_levels = atoi(e); }
    { const char* e = getenv("HS_FAST_KEYS_MAX_BATCH"); if (e && atoi(e) >= 0) h->fast_keys_max_batch = atoi(e); }
    { const char* e = getenv("HS_FAST_ORDER"); h->fast_order = e ? atoi(e) : 1; }
    { const char* e = getenv("HS_PYRAMID_CHAIN"); h->chain_mode = e ? atoi(e) : -1; }
    { const char* e = getenv("HS_PYRAMID_TBX_MAX"); if (e) h->pyr_tbx_max = atoi(e); }
    { const char* e = getenv("HS_PYRAMID_DEEP_MAX"); if (e) h->deep_max_batch = atoi(e); }
    { const char* e = getenv("HS_PYRAMID_DEEP_ROWS"); if (e && atoi(e) >= 2) h->deep_rows = atoi(e); }
    { const char* e = getenv("HS_QT_POINT_DOMAIN"); h->qt_point_domain = e && atoi(e) != 0; }
    { const char* e = getenv("HS_EXTRACT_SPLIT"); h->split_mode = e ? (atoi(e) != 0 ? 1 : 0) : -1; }
    bool zero = true; for (int k = 0; k < 7; k++) zero = zero && p->blur_taps[k] == 0;
    static const uint16_t def[7] = { 18, 34, 49, 55, 49, 34, 18 };
    for (int k = 0; k < 7; k++) h->taps[k] = zero ? def[k] : p->blur_taps[k];
    { uint32_t sum = 0; bool bytes = true; for (int k = 0; k < 7; k++) { sum += h->taps[k]; bytes = bytes && h->taps[k] <= 255; } h->fast_taps = bytes && sum * 255u <= 0xFFFFu; }

    // ORBExtractor::ORBExtractor (ORBExtractor.cpp:86-118); scaleFactor is a double member fed from a float setting
    const int L = p->nlevels;
    const double scaleFactor = p->scale_factor;
    h->scale.resize(L); h->inv_scale.resize(L); h->sigma2.resize(L); h->inv_sigma2.resize(L); h->quota.resize(L);
    h->scale[0] = 1.0f; h->sigma2[0] = 1.0f;
    for (int i = 1; i < L; i++) { h->scale[i] = (float)(h->scale[i - 1] * scaleFactor); h->sigma2[i] = h->scale[i] * h->scale[i]; }
    for (int i = 0; i < L; i++) { h->inv_scale[i] = 1.0f / h->scale[i]; h->inv_sigma2[i] = 1.0f / h->sigma2[i]; }
    float factor = (float)(1.0f / scaleFactor);
    float nDesired = p->nfeatures * (1 - factor) / (1 - (float)pow((double)factor, (double)L));
    int sum = 0;
    for (int l = 0; l < L - 1; l++) { h->quota[l] = cv_round_f(nDesired); sum += h->quota[l]; nDesired *= factor; }
    h->quota[L - 1] = std::max(p->nfeatures - sum, 0);
    for (int l = 0; l < L; l++)
        if (h->quota[l] + 8 > HS_QT_LARGE_NODES) { delete h; return HS_ERR_INVALID; }      // (a level's quota above 3320: nFeatures beyond ~11 600 @1.4 / ~15 300 @1.2 with 8 levels)
    for (int l = 0; l < L; l++) if (h->quota[l] + 8 > HS_QT_MAX_NODES) h->qt_large = true;
    h->qt_small_ok = true;
    for (int l = 0; l < L; l++) if (h->quota[l] + 8 > hs_quadtree_small_nodes()) h->qt_small_ok = false;
    // ... and only on request (HS_QT_SMALL=1, read once): measured at 32 / 64 pairs per call the two-per-CU instance is SLOWER (quadtree 0.0631 against 0.0606 ms,
    // 0.1198 against 0.1117): without the points in LDS every sweep goes through L2, which costs a workgroup more than sharing the CU buys.  Kept as a parity /
    // tuning variant (tests/test_gpu_parity.py runs it).
    { const char* e = getenv("HS_QT_SMALL"); if (!e || atoi(e) == 0) h->qt_small_ok = false; }

    if (hipSetDevice(device) != hipSuccess) { delete h; return HS_ERR_NO_DEVICE; }
    if (hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking) != hipSuccess ||
        hipMalloc(&h->d_lv, sizeof(HsLevel) * 2 * HS_MAX_LEVELS) != hipSuccess ||
        hipMalloc(&h->d_taps, 16) != hipSuccess ||
        hipMemcpy(h->d_taps, h->taps, 14, hipMemcpyHostToDevice) != hipSuccess) {
        hs_orb_destroy(h);
        return HS_ERR_HIP;
    }
    *out = h;
    return HS_OK;
}

static void orb_destroy_now(hs_orb* h);
void hs_orb_destroy(hs_orb* h)
{
    if (!h) return;
    h->owned.store(false);
    if (h->refs.fetch_sub(1) == 1) orb_destroy_now(h);                   // else a communicator still uses the handle: the last hs_comm_destroy frees it
}
int hs_orb_borrowers(const hs_orb* h) { return h ? std::max(0, h->refs.load() - (h->owned.load() ? 1 : 0)) : 0; }
} // extern "C"
// hs_comm.hip: +1 when a communicator is created on the handle, -1 when it is destroyed (which also completes a deferred hs_orb_destroy)
void hs_orb_borrow(hs_orb* h, int delta)
{
    if (!h) return;
    if (h->refs.fetch_add(delta) + delta == 0) orb_destroy_now(h);
}
static void orb_destroy_now(hs_orb* h)
{
    hipSetDevice(h->device);
    if (h->lane2) { hs_orb_destroy(h->lane2); h->lane2 = nullptr; }
    if (h->ev_fork) hipEventDestroy(h->ev_fork);
    if (h->ev_join) hipEventDestroy(h->ev_join);
    if (h->s_aux) { hipStreamSynchronize(h->s_aux); hipStreamDestroy(h->s_aux); }
    if (h->ev_sfork) hipEventDestroy(h->ev_sfork);
    if (h->ev_sjoin) hipEventDestroy(h->ev_sjoin);
    if (h->stream) hipStreamSynchronize(h->stream);
    free_geometry(h);
    hipFree(h->d_lv); hipFree(h->d_taps); hipFree(h->d_in); hipFree(h->d_raw);
    hipFree(h->d_n);                     // the output block (counts, keypoints, descriptors)
    hipFree(h->d_ur); hipFree(h->d_depth); hipFree(h->d_bd); hipFree(h->d_scratch); hipFree(h->d_strip_count); hipFree(h->d_strip_list);
    hipFree(h->d_sm_kps); hipFree(h->d_sm_desc); hipFree(h->d_sm_n);
    if (h->h_pin) hipHostFree(h->h_pin);
    if (h->h_pin_out) hipHostFree(h->h_pin_out);
    if (h->s_in) hipStreamSynchronize(h->s_in);
    if (h->s_out) hipStreamSynchronize(h->s_out);
    for (auto& sl : h->slot) {
        hipFree(sl.d_in); hipFree(sl.d_raw); hipFree(sl.d_out);
        if (sl.h_out) hipHostFree(sl.h_out);
        if (sl.ev_in) hipEventDestroy(sl.ev_in);
        if (sl.ev_done) hipEventDestroy(sl.ev_done);
        if (sl.ev_out) hipEventDestroy(sl.ev_out);
    }
    if (h->s_in) hipStreamDestroy(h->s_in);
    if (h->s_out) hipStreamDestroy(h->s_out);
    for (hipEvent_t e : h->ev_pool) hipEventDestroy(e);
    if (h->stream) hipStreamDestroy(h->stream);
    delete h;
}
extern "C" {

const char* hs_orb_last_error(const hs_orb* h) { return h ? h->err.c_str() : "null handle"; }
int hs_orb_get_levels(const hs_orb* h) { return h ? h->p.nlevels : 0; }
int hs_orb_get_device(const hs_orb* h) { return h ? h->device : -1; }
float hs_orb_get_scale_factor(const hs_orb* h) { return h ? (float)(double)h->p.scale_factor : 0.f; }

int hs_orb_get_scale_tables(const hs_orb* h, float* scale, float* inv_scale, float* sigma2, float* inv_sigma2, int32_t* fpl)
{
    if (!h) return HS_ERR_INVALID;
    for (int i = 0; i < h->p.nlevels; i++) {
        if (scale) scale[i] = h->scale[i];
        if (inv_scale) inv_scale[i] = h->inv_scale[i];
        if (sigma2) sigma2[i] = h->sigma2[i];
        if (inv_sigma2) inv_sigma2[i] = h->inv_sigma2[i];
        if (fpl) fpl[i] = h->quota[i];
    }
    return HS_OK;
}

int hs_orb_max_keypoints(const hs_orb* h)
{
    if (!h) return 0;
    // DistributeOctTree stops at the first split that reaches N nodes: at most N+2 per level, or the
    // 4*nIni nodes of the unconditional first pass.  nIni depends on the frame; assume the widest supported.
    int n = 0;
    for (int l = 0; l < h->p.nlevels; l++) n += std::max(h->quota[l] + 4, 4 * 8 + 4);
    return std::max(n, h->max_kp);          // a configured geometry (hs_orb_reserve / a previous extract) with more than 8 root nodes per level needs more
}

int hs_orb_reserve(hs_orb* h, int w, int h_px, int batch)
{
    if (!h) return HS_ERR_INVALID;
    HIP_TRY(h, hipSetDevice(h->device));
    return configure(h, w, h_px, batch);
}

int hs_orb_extract_batch_device(hs_orb* h, const uint8_t* d_imgs, int batch, int w, int h_px,
                                size_t row_stride, size_t image_stride,
                                hs_keypoint* d_kps, uint8_t* d_desc, int32_t* d_n, int cap, void* stream)
{
    if (!h) return HS_ERR_INVALID;
    if (!d_imgs || !d_kps || !d_desc || !d_n || batch < 1 || row_stride < (size_t)w || cap < 1 || cap > 65535)
        return fail(h, HS_ERR_INVALID, "bad argument");
    if (((uintptr_t)d_desc & 15) != 0 || (((uintptr_t)d_kps | (uintptr_t)d_n) & 3) != 0)
        return fail(h, HS_ERR_INVALID, "output alignment: descriptors 16 bytes (they are written with 16-byte vector stores; hs_record_offsets pads for it), keypoints and counts 4");
    HIP_TRY(h, hipSetDevice(h->device));
    if (h->lane2 && batch >= 2) {      // two lanes: the second half runs on the child handle's stream, fenced by fork / join events
        hipStream_t s = stream ? (hipStream_t)stream : h->stream;
        const int b0 = batch / 2, b1 = batch - b0;
        h->lane2->prof = h->prof;
        HIP_TRY(h, hipEventRecord(h->ev_fork, s));
        HIP_TRY(h, hipStreamWaitEvent(h->lane2->stream, h->ev_fork, 0));
        hs_orb* child = h->lane2; h->lane2 = nullptr;          // the recursive calls below must not split again
        int rc = hs_orb_extract_batch_device(h, d_imgs, b0, w, h_px, row_stride, image_stride, d_kps, d_desc, d_n, cap, s);
        if (rc == HS_OK) {
            rc = hs_orb_extract_batch_device(child, d_imgs + (size_t)b0 * image_stride, b1, w, h_px, row_stride, image_stride,
                                             d_kps + (size_t)b0 * cap, d_desc + (size_t)b0 * cap * HS_DESC_BYTES, d_n + b0, cap, child->stream);
            if (rc != HS_OK) h->err = child->err;
        }
        h->lane2 = child;
        HIP_TRY(h, hipEventRecord(h->ev_join, child->stream));
        HIP_TRY(h, hipStreamWaitEvent(s, h->ev_join, 0));
        return rc;
    }
    int rc = configure(h, w, h_px, batch);
    if (rc != HS_OK) return rc;
    if (cap < h->max_kp) return fail(h, HS_ERR_CAPACITY, "cap < keypoints this frame size can produce; see hs_orb_max_keypoints");
    hipStream_t s = stream ? (hipStream_t)stream : h->stream;
    HsImg0 img0{ d_imgs, d_imgs, batch, (uint64_t)row_stride, (uint64_t)image_stride };
    HsOut out{ d_kps, d_desc, d_n, d_kps, d_desc, d_n, batch, cap };
    return run_extract(h, img0, batch, out, s);
}

// the host-pointer extraction, shared by hs_orb_extract_batch (grey frames) and hs_orb_extract_camera_batch (pp != nullptr: the frames are what the camera
// delivers — 1 / 3 / 4 channels at its own size — and ImageProcessing::PreProcessImg runs on the device between the upload and the pyramid)
namespace {
int extract_host_frames(hs_orb* h, const uint8_t* const* imgs, int batch, int w, int h_px, size_t stride, const hs_preprocess_params* pp,
                        hs_keypoint* kps, uint8_t* desc, int cap, int32_t* n, uint8_t* grey_out)
{
    HIP_TRY(h, hipSetDevice(h->device));
    h->pub_kps = nullptr; h->pub_desc = nullptr; h->pub_batch = 0;
    int gw = w, gh = h_px;                                   // size of the grey level 0
    if (pp) hs_preprocess_out_size(w, h_px, pp->scale, &gw, &gh);
    if (gw < 1 || gh < 1) return fail(h, HS_ERR_INVALID, "the camera scale reduces the frame to nothing (cv::resize asserts on an empty size)");
    int rc = configure(h, gw, gh, batch);
    if (rc != HS_OK) return rc;
    if (cap < h->max_kp) return fail(h, HS_ERR_CAPACITY, "cap < keypoints this frame size can produce; see hs_orb_max_keypoints");
    const size_t pitch = ((size_t)gw + 63) & ~(size_t)63, per_img = pitch * gh;
    if (per_img * batch > h->in_bytes) {
        HIP_TRY(h, hipStreamSynchronize(h->stream));
        hipFree(h->d_in); h->d_in = nullptr;
        HIP_TRY(h, hipMalloc(&h->d_in, per_img * batch));
        h->in_bytes = per_img * batch;
    }
    rc = ensure_outputs(h, batch, cap);
    if (rc != HS_OK) return rc;
    hipStream_t s = h->stream;
    for (int i = 0; i < batch; i++) if (!imgs[i]) return fail(h, HS_ERR_INVALID, "null image in batch");
    if (!pp) {
        // Frames: the runtime's own pageable-memory path (measured: packing the rows into a pinned buffer on the calling thread first is SLOWER —
        // one core copies 2 MB frames at ~10 GB/s, the runtime's staged copy moves them at more than twice that)
        for (int i = 0; i < batch; i++) {
            // a frame whose rows are as far apart as the staging copy's (width a multiple of 64, no padding: 1920 x 1080) is ONE linear copy
            if (stride == pitch && (size_t)w == pitch) HIP_TRY(h, hipMemcpyAsync(h->d_in + per_img * i, imgs[i], per_img, hipMemcpyHostToDevice, s));      // (rows with padding: the last row's padding need not exist in the caller's buffer)
            else HIP_TRY(h, hipMemcpy2DAsync(h->d_in + per_img * i, pitch, imgs[i], stride, w, h_px, hipMemcpyHostToDevice, s));
        }
    } else {
        // the camera's frames cross PCIe as they are (once), rows packed to a multiple of 4 bytes; PreProcessImg runs between the copy and the pyramid
        const size_t row_bytes = (size_t)w * pp->channels, rpitch = (row_bytes + 3) & ~(size_t)3, raw_img = rpitch * h_px;
        if (raw_img * batch > h->raw_bytes) {
            HIP_TRY(h, hipStreamSynchronize(h->stream));
            hipFree(h->d_raw); h->d_raw = nullptr; h->raw_bytes = 0;
            HIP_TRY(h, hipMalloc(&h->d_raw, raw_img * batch));
            h->raw_bytes = raw_img * batch;
        }
        for (int i = 0; i < batch; i++) {
            if (stride == rpitch && row_bytes == rpitch) HIP_TRY(h, hipMemcpyAsync(h->d_raw + raw_img * i, imgs[i], raw_img, hipMemcpyHostToDevice, s));
            else HIP_TRY(h, hipMemcpy2DAsync(h->d_raw + raw_img * i, rpitch, imgs[i], stride, row_bytes, h_px, hipMemcpyHostToDevice, s));
        }
        hs_launch_preprocess(h->d_raw, w, h_px, rpitch, raw_img, pp->channels, pp->rgb, pp->scale, h->d_in, gw, gh, pitch, per_img, 1, batch, s);
        HIP_TRY(h, hipGetLastError());
    }
    HsImg0 img0{ h->d_in, h->d_in, batch, (uint64_t)pitch, (uint64_t)per_img };
    HsOut out{ h->d_kps, h->d_desc, h->d_n, h->d_kps, h->d_desc, h->d_n, batch, cap };
    rc = run_extract(h, img0, batch, out, s);
    if (rc != HS_OK) return rc;
    // counts, keypoints and descriptors live in one device block (ensure_outputs): one copy into pinned memory, then the used part goes to the caller
    const size_t nb = pad256((size_t)h->out_batch * 4), kb = pad256((size_t)h->out_batch * cap * sizeof(hs_keypoint));
    const size_t out_bytes = nb + kb + (size_t)batch * cap * HS_DESC_BYTES;
    rc = ensure_pinned(h, &h->h_pin_out, &h->pin_out_bytes, out_bytes);
    if (rc != HS_OK) return rc;
    HIP_TRY(h, hipMemcpyAsync(h->h_pin_out, h->d_n, out_bytes, hipMemcpyDeviceToHost, s));
    // the grey frame the reference keeps beside the features (track_data.image = mImGray, ImageProcessing.cpp:60,108): tight rows, on request
    if (grey_out) HIP_TRY(h, hipMemcpy2DAsync(grey_out, (size_t)gw, h->d_in, pitch, (size_t)gw, (size_t)gh * batch, hipMemcpyDeviceToHost, s));
    HIP_TRY(h, hipStreamSynchronize(s));
    memcpy(n, h->h_pin_out, (size_t)batch * 4);
    for (int i = 0; i < batch; i++) {            // only the keypoints that exist are copied; the rest of the caller's arrays is left untouched
        const size_t cnt = (size_t)std::min(std::max(n[i], 0), cap);
        memcpy(kps + (size_t)i * cap, h->h_pin_out + nb + (size_t)i * cap * sizeof(hs_keypoint), cnt * sizeof(hs_keypoint));
        memcpy(desc + (size_t)i * cap * HS_DESC_BYTES, h->h_pin_out + nb + kb + (size_t)i * cap * HS_DESC_BYTES, cnt * HS_DESC_BYTES);
    }
    h->pub_kps = h->d_kps; h->pub_desc = h->d_desc; h->pub_cap = cap; h->pub_batch = batch;
    return HS_OK;
}
}  // namespace

int hs_orb_extract_batch(hs_orb* h, const uint8_t* const* imgs, int batch, int w, int h_px, int stride,
                         hs_keypoint* kps, uint8_t* desc, int cap, int32_t* n)
{
    if (!h) return HS_ERR_INVALID;
    if (!n || batch < 1) return fail(h, HS_ERR_INVALID, "bad argument");
    if (w == 0 || h_px == 0 || !imgs) { for (int i = 0; i < batch; i++) n[i] = 0; return HS_OK; }   // ORBExtractor.cpp:499-500
    if (!kps || !desc || stride < w || cap < 1 || cap > 65535) return fail(h, HS_ERR_INVALID, "bad argument");
    return extract_host_frames(h, imgs, batch, w, h_px, (size_t)stride, nullptr, kps, desc, cap, n, nullptr);
}

void hs_preprocess_size(int w, int h_px, float scale, int32_t* ow, int32_t* oh)
{
    int a = 0, b = 0;
    hs_preprocess_out_size(w, h_px, scale, &a, &b);
    if (ow) *ow = a;
    if (oh) *oh = b;
}

static bool preprocess_params_ok(const hs_preprocess_params* pp)
{
    return pp && (pp->channels == 1 || pp->channels == 3 || pp->channels == 4) && pp->scale > 0.f && pp->scale <= 16.f;
}

int hs_preprocess_device(hs_orb* h, const uint8_t* d_src, int w, int h_px, size_t row_stride, size_t image_stride, int batch, const hs_preprocess_params* pp,
                         uint8_t* d_grey, size_t grey_row_stride, size_t grey_image_stride, void* stream)
{
    if (!h) return HS_ERR_INVALID;
    if (!preprocess_params_ok(pp) || !d_src || !d_grey || w < 1 || h_px < 1 || w > 32768 || h_px > 32768 || batch < 1 || row_stride < (size_t)w * pp->channels)
        return fail(h, HS_ERR_INVALID, "bad argument");
    int gw, gh;
    hs_preprocess_out_size(w, h_px, pp->scale, &gw, &gh);
    if (gw < 1 || gh < 1 || gw > 16384 || gh > 16384) return fail(h, HS_ERR_INVALID, "the scaled frame is empty or larger than 16384 px");
    if (grey_row_stride < (size_t)gw) return fail(h, HS_ERR_INVALID, "grey_row_stride < scaled width");
    HIP_TRY(h, hipSetDevice(h->device));
    hs_launch_preprocess(d_src, w, h_px, row_stride, image_stride, pp->channels, pp->rgb, pp->scale, d_grey, gw, gh, grey_row_stride, grey_image_stride, 0, batch,
                         stream ? (hipStream_t)stream : h->stream);
    HIP_TRY(h, hipGetLastError());
    return HS_OK;
}

int hs_orb_extract_camera_batch(hs_orb* h, const uint8_t* const* imgs, int batch, int w, int h_px, size_t row_stride, const hs_preprocess_params* pp,
                                hs_keypoint* kps, uint8_t* desc, int cap, int32_t* n, uint8_t* grey_out)
{
    if (!h) return HS_ERR_INVALID;
    if (!n || batch < 1) return fail(h, HS_ERR_INVALID, "bad argument");
    if (w == 0 || h_px == 0 || !imgs) { for (int i = 0; i < batch; i++) n[i] = 0; return HS_OK; }   // ORBExtractor.cpp:499-500
    if (!preprocess_params_ok(pp) || !kps || !desc || w < 0 || h_px < 0 || w > 32768 || h_px > 32768 || row_stride < (size_t)w * pp->channels || cap < 1 || cap > 65535)
        return fail(h, HS_ERR_INVALID, "bad argument");
    return extract_host_frames(h, imgs, batch, w, h_px, row_stride, pp, kps, desc, cap, n, grey_out);
}

int hs_orb_extract(hs_orb* h, const uint8_t* img, int w, int h_px, int stride,
                   hs_keypoint* kps, uint8_t* desc, int cap, int32_t* n)
{
    if (!h) return HS_ERR_INVALID;
    if (!n) return fail(h, HS_ERR_INVALID, "bad argument");
    if (!img || w == 0 || h_px == 0) { *n = 0; return HS_OK; }
    const uint8_t* one[1] = { img };
    return hs_orb_extract_batch(h, one, 1, w, h_px, stride, kps, desc, cap, n);
}

int hs_stereo_match_batch_device(hs_orb* h, const hs_keypoint* d_kpsL, const uint8_t* d_descL, const int32_t* d_nL,
                                 const hs_keypoint* d_kpsR, const uint8_t* d_descR, const int32_t* d_nR,
                                 int pairs, int cap, const hs_stereo_params* sp,
                                 float* d_uRight, float* d_depth, void* stream)
{
    if (!h) return HS_ERR_INVALID;
    if (!d_kpsL || !d_descL || !d_nL || !d_kpsR || !d_descR || !d_nR || !sp || !d_uRight || !d_depth ||
        pairs < 1 || pairs > 65535 || cap < 1 || cap > 65535)
        return fail(h, HS_ERR_INVALID, "bad argument");
    HIP_TRY(h, hipSetDevice(h->device));
    int rc = ensure_stereo_scratch(h, (size_t)pairs * cap);
    if (rc == HS_OK) rc = ensure_stereo_strips(h, pairs, cap, sp->n_rows);
    if (rc != HS_OK) return rc;
    hipStream_t s = stream ? (hipStream_t)stream : h->stream;
    run_stereo(h, d_kpsL, d_descL, d_nL, d_kpsR, d_descR, d_nR, pairs, cap, *sp, d_uRight, d_depth, s);
    HIP_TRY(h, hipGetLastError());
    return HS_OK;
}

int hs_stereo_match(hs_orb* h, const hs_keypoint* kpsL, const uint8_t* descL, int nL,
                    const hs_keypoint* kpsR, const uint8_t* descR, int nR,
                    const hs_stereo_params* sp, float* uRight, float* depth)
{
    if (!h) return HS_ERR_INVALID;
    if (!sp || nL < 0 || nR < 0 || nL > 65535 || nR > 65535) return fail(h, HS_ERR_INVALID, "bad argument");
    if (nL == 0) return HS_OK;
    if (!kpsL || !descL || !uRight || !depth || (nR > 0 && (!kpsR || !descR))) return fail(h, HS_ERR_INVALID, "bad argument");
    HIP_TRY(h, hipSetDevice(h->device));
    const int cap = std::max(std::max(nL, nR), 1);
    hipStream_t s = h->stream;
    // persistent staging (grow-only): no allocation on the steady-state path.  Inputs go through one pinned block so that the five
    // H2D copies are real asynchronous DMAs; outputs come back into the same block.
    if (cap > h->sm_cap) {
        HIP_TRY(h, hipStreamSynchronize(s));
        hipFree(h->d_sm_kps); hipFree(h->d_sm_desc); hipFree(h->d_sm_n); h->d_sm_kps = nullptr; h->d_sm_desc = nullptr; h->d_sm_n = nullptr; h->sm_cap = 0;
        const int grow = std::max(cap, 2048);
        HIP_TRY(h, hipMalloc(&h->d_sm_kps, (size_t)2 * grow * sizeof(hs_keypoint)));
        HIP_TRY(h, hipMalloc(&h->d_sm_desc, (size_t)2 * grow * HS_DESC_BYTES));
        HIP_TRY(h, hipMalloc(&h->d_sm_n, 8));
        h->sm_cap = grow;
    }
    const size_t kb = (size_t)cap * sizeof(hs_keypoint), db = (size_t)cap * HS_DESC_BYTES;
    const size_t pin_need = 2 * kb + 2 * db + 16 + 2 * (size_t)cap * 4;
    if (pin_need > h->pin_bytes) {
        HIP_TRY(h, hipStreamSynchronize(s));
        if (h->h_pin) hipHostFree(h->h_pin);
        h->h_pin = nullptr; h->pin_bytes = 0;
        const size_t grow = std::max<size_t>(pin_need, 1 << 18);
        HIP_TRY(h, hipHostMalloc(&h->h_pin, grow, hipHostMallocDefault));
        h->pin_bytes = grow;
    }
    int rc = ensure_stereo_scratch(h, (size_t)cap);
    if (rc == HS_OK) rc = ensure_stereo_strips(h, 1, cap, sp->n_rows);
    if (rc != HS_OK) return rc;
    hs_keypoint* dk = h->d_sm_kps; uint8_t* dd = h->d_sm_desc; int32_t* dn = h->d_sm_n;
    uint8_t* pk = h->h_pin; uint8_t* pd = pk + 2 * kb; int32_t* pn = reinterpret_cast<int32_t*>(pd + 2 * db); float* pout = reinterpret_cast<float*>(pd + 2 * db + 16);
    memcpy(pk, kpsL, (size_t)nL * sizeof(hs_keypoint));
    if (nR) memcpy(pk + kb, kpsR, (size_t)nR * sizeof(hs_keypoint));
    memcpy(pd, descL, (size_t)nL * 32);
    if (nR) memcpy(pd + db, descR, (size_t)nR * 32);
    pn[0] = nL; pn[1] = nR;
    HIP_TRY(h, hipMemcpyAsync(dk, pk, 2 * kb, hipMemcpyHostToDevice, s));                 // left at [0, cap), right at [cap, 2 cap)
    HIP_TRY(h, hipMemcpyAsync(dd, pd, 2 * db, hipMemcpyHostToDevice, s));
    HIP_TRY(h, hipMemcpyAsync(dn, pn, 8, hipMemcpyHostToDevice, s));
    run_stereo(h, dk, dd, dn, dk + cap, dd + (size_t)cap * 32, dn + 1, 1, cap, *sp, h->d_ur, h->d_depth, s);
    HIP_TRY(h, hipGetLastError());
    HIP_TRY(h, hipMemcpyAsync(pout, h->d_ur, (size_t)nL * 4, hipMemcpyDeviceToHost, s));
    HIP_TRY(h, hipMemcpyAsync(pout + cap, h->d_depth, (size_t)nL * 4, hipMemcpyDeviceToHost, s));
    HIP_TRY(h, hipStreamSynchronize(s));
    memcpy(uRight, pout, (size_t)nL * 4);
    memcpy(depth, pout + cap, (size_t)nL * 4);
    return HS_OK;
}

int hs_stereo_frontend_batch_device(hs_orb* h, const uint8_t* d_left, const uint8_t* d_right, int pairs,
                                    int w, int h_px, size_t row_stride, size_t image_stride,
                                    hs_keypoint* d_kpsL, uint8_t* d_descL, int32_t* d_nL,
                                    hs_keypoint* d_kpsR, uint8_t* d_descR, int32_t* d_nR, int cap,
                                    const hs_stereo_params* sp, float* d_uRight, float* d_depth, void* stream)
{
    // The reference runs two ORBExtractor instances side by side (ImageProcessing.cpp:82-83); here the left and
    // right frames of all pairs go through ONE launch sequence (images [0,pairs) = left, [pairs,2*pairs) = right).
    if (!h) return HS_ERR_INVALID;
    if (!d_left || !d_right || !d_kpsL || !d_descL || !d_nL || !d_kpsR || !d_descR || !d_nR || !sp || !d_uRight || !d_depth ||
        pairs < 1 || 2 * pairs > 65535 || row_stride < (size_t)w || cap < 1 || cap > 65535)
        return fail(h, HS_ERR_INVALID, "bad argument");
    if ((((uintptr_t)d_descL | (uintptr_t)d_descR) & 15) != 0 || (((uintptr_t)d_kpsL | (uintptr_t)d_kpsR | (uintptr_t)d_nL | (uintptr_t)d_nR | (uintptr_t)d_uRight | (uintptr_t)d_depth) & 3) != 0)
        return fail(h, HS_ERR_INVALID, "output alignment: descriptors 16 bytes (they are written with 16-byte vector stores), everything else 4");
    HIP_TRY(h, hipSetDevice(h->device));
    if (h->lane2 && pairs >= 2) {      // two lanes: each handles half of the pairs end to end (extract L+R, match) on its own stream
        hipStream_t s = stream ? (hipStream_t)stream : h->stream;
        const int p0 = pairs / 2, p1 = pairs - p0;
        h->lane2->prof = h->prof;
        HIP_TRY(h, hipEventRecord(h->ev_fork, s));
        HIP_TRY(h, hipStreamWaitEvent(h->lane2->stream, h->ev_fork, 0));
        hs_orb* child = h->lane2; h->lane2 = nullptr;
        int rc = hs_stereo_frontend_batch_device(h, d_left, d_right, p0, w, h_px, row_stride, image_stride, d_kpsL, d_descL, d_nL,
                                                 d_kpsR, d_descR, d_nR, cap, sp, d_uRight, d_depth, s);
        if (rc == HS_OK) {
            const size_t io = (size_t)p0 * image_stride, ko = (size_t)p0 * cap;
            rc = hs_stereo_frontend_batch_device(child, d_left + io, d_right + io, p1, w, h_px, row_stride, image_stride,
                                                 d_kpsL + ko, d_descL + ko * HS_DESC_BYTES, d_nL + p0, d_kpsR + ko, d_descR + ko * HS_DESC_BYTES, d_nR + p0,
                                                 cap, sp, d_uRight + ko, d_depth + ko, child->stream);
            if (rc != HS_OK) h->err = child->err;
        }
        h->lane2 = child;
        HIP_TRY(h, hipEventRecord(h->ev_join, child->stream));
        HIP_TRY(h, hipStreamWaitEvent(s, h->ev_join, 0));
        return rc;
    }
    int rc = configure(h, w, h_px, 2 * pairs);
    if (rc != HS_OK) return rc;
    if (cap < h->max_kp) return fail(h, HS_ERR_CAPACITY, "cap < keypoints this frame size can produce; see hs_orb_max_keypoints");
    rc = ensure_stereo_scratch(h, (size_t)pairs * cap);
    if (rc == HS_OK) rc = ensure_stereo_strips(h, pairs, cap, sp->n_rows);
    if (rc != HS_OK) return rc;
    hipStream_t s = stream ? (hipStream_t)stream : h->stream;
    HsImg0 img0{ d_left, d_right, pairs, (uint64_t)row_stride, (uint64_t)image_stride };
    HsOut out{ d_kpsL, d_descL, d_nL, d_kpsR, d_descR, d_nR, pairs, cap };
    if (h->stereo_fuse) {
        const HsStripFuse sf{ 1, sp->n_rows, hs_stereo_strips(sp->n_rows), 0, sp->size_ref, h->d_strip_count, reinterpret_cast<HsStripEntry*>(h->d_strip_list) };
        rc = run_extract(h, img0, 2 * pairs, out, s, &sf);
        if (rc != HS_OK) return rc;
        run_stereo_fused(h, d_kpsL, d_descL, d_nL, d_kpsR, d_descR, d_nR, pairs, cap, *sp, d_uRight, d_depth, s);
    } else {
        rc = run_extract(h, img0, 2 * pairs, out, s);
        if (rc != HS_OK) return rc;
        run_stereo(h, d_kpsL, d_descL, d_nL, d_kpsR, d_descR, d_nR, pairs, cap, *sp, d_uRight, d_depth, s);
    }
    HIP_TRY(h, hipGetLastError());
    return HS_OK;
}

/* ---- pipelined host ingest ----
 * The reference feeds its extractor from host memory through a bounded queue: System::TrackStereo pushes frames and throttles when more than two
 * are waiting (src/main/System.cc:194-196), ImageProcessing pops, extracts, matches and pushes the features on (src/main/ImageProcessing.cpp:69-116).
 * Here: submit(i+1) copies its frames in on the copy-in stream WHILE the kernels of batch i run on the compute stream and the results of batch i
 * leave on the copy-out stream; two staging slots, so at most two tickets are in flight — the same bound as the reference's queue. */
int hs_host_alloc(size_t bytes, void** out)
{
    if (!out || bytes == 0) return HS_ERR_INVALID;
    *out = nullptr;
    if (hipHostMalloc(out, bytes, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); *out = nullptr; return HS_ERR_HIP; }
    return HS_OK;
}
void hs_host_free(void* p) { if (p) (void)hipHostFree(p); }

// shared by hs_orb_submit_batch (grey frames: pp == nullptr, src_w x src_h IS the level-0 size) and hs_orb_submit_camera_batch (pp: the camera's own frames;
// ImageProcessing::PreProcessImg runs on the compute stream between the copy-in and the pyramid)
static int submit_frames(hs_orb* h, const uint8_t* const* imgs, int batch, int src_w, int src_h, size_t stride, const hs_preprocess_params* pp, const hs_stereo_params* sp, int32_t* ticket)
{
    for (int i = 0; i < batch; i++) if (!imgs[i]) return fail(h, HS_ERR_INVALID, "null image in batch");
    int w = src_w, h_px = src_h;
    if (pp) hs_preprocess_out_size(src_w, src_h, pp->scale, &w, &h_px);
    if (w < 1 || h_px < 1) return fail(h, HS_ERR_INVALID, "the camera scale reduces the frame to nothing (cv::resize asserts on an empty size)");
    *ticket = 0;
    h->pub_kps = nullptr; h->pub_desc = nullptr; h->pub_batch = 0;      // a slot's device block may be rewritten from here on
    HIP_TRY(h, hipSetDevice(h->device));
    hs_orb::IngestSlot* sl = nullptr;
    for (auto& c : h->slot) if (!c.busy) { sl = &c; break; }
    if (!sl) return fail(h, HS_ERR_INVALID, "both staging slots are in flight: hs_orb_wait for the oldest ticket first");
    if (!h->s_in) {
        HIP_TRY(h, hipStreamCreateWithFlags(&h->s_in, hipStreamNonBlocking));
        HIP_TRY(h, hipStreamCreateWithFlags(&h->s_out, hipStreamNonBlocking));
    }
    if (!sl->ev_in) {
        HIP_TRY(h, hipEventCreateWithFlags(&sl->ev_in, hipEventDisableTiming));
        HIP_TRY(h, hipEventCreateWithFlags(&sl->ev_done, hipEventDisableTiming));
        HIP_TRY(h, hipEventCreateWithFlags(&sl->ev_out, hipEventDisableTiming));
    }
    if (!(w == h->w && h_px == h->h && batch <= h->batch_cap)) {      // a new geometry rebuilds the shared workspace: nothing may be in flight on it
        HIP_TRY(h, hipStreamSynchronize(h->s_in)); HIP_TRY(h, hipStreamSynchronize(h->s_out));
    }
    int rc = configure(h, w, h_px, batch);
    if (rc != HS_OK) return rc;
    const int cap = h->max_kp, pairs = sp ? batch / 2 : 0;
    if (cap < 1) return fail(h, HS_ERR_INVALID, "this frame size yields no keypoints");
    const size_t pitch = ((size_t)w + 63) & ~(size_t)63, per_img = pitch * h_px;
    if (per_img * batch > sl->in_bytes) {
        hipFree(sl->d_in); sl->d_in = nullptr; sl->in_bytes = 0;           // the slot is idle: its last batch was waited for
        HIP_TRY(h, hipMalloc(&sl->d_in, per_img * batch));
        sl->in_bytes = per_img * batch;
    }
    const size_t raw_row = pp ? (size_t)src_w * pp->channels : 0, rpitch = (raw_row + 3) & ~(size_t)3, raw_img = rpitch * src_h;      // the camera's frames as uploaded
    if (pp && raw_img * batch > sl->raw_bytes) {
        hipFree(sl->d_raw); sl->d_raw = nullptr; sl->raw_bytes = 0;
        HIP_TRY(h, hipMalloc(&sl->d_raw, raw_img * batch));
        sl->raw_bytes = raw_img * batch;
    }
    sl->off_k = pad256((size_t)batch * 4);
    sl->off_d = sl->off_k + pad256((size_t)batch * cap * sizeof(hs_keypoint));
    sl->off_u = sl->off_d + pad256((size_t)batch * cap * HS_DESC_BYTES);
    sl->off_z = sl->off_u + pad256((size_t)std::max(pairs, 1) * cap * 4);
    sl->used = sl->off_z + pad256((size_t)std::max(pairs, 1) * cap * 4);
    if (sl->used > sl->out_bytes) {
        hipFree(sl->d_out); sl->d_out = nullptr; sl->out_bytes = 0;
        HIP_TRY(h, hipMalloc(&sl->d_out, sl->used));
        sl->out_bytes = sl->used;
    }
    if (sl->used > sl->h_out_bytes) {
        if (sl->h_out) hipHostFree(sl->h_out);
        sl->h_out = nullptr; sl->h_out_bytes = 0;
        HIP_TRY(h, hipHostMalloc((void**)&sl->h_out, sl->used, hipHostMallocDefault));
        sl->h_out_bytes = sl->used;
    }
    if (sp) {
        rc = ensure_stereo_scratch(h, (size_t)pairs * cap);
        if (rc == HS_OK) rc = ensure_stereo_strips(h, pairs, cap, sp->n_rows);
        if (rc != HS_OK) return rc;
    }
    // From here on work is ENQUEUED that targets the slot's buffers: whatever fails below, the three streams are drained before the call returns,
    // so that a slot handed out again (it stays !busy) is never written by a copy or a kernel of the failed attempt.
    auto enqueue = [&]() -> int {
    // copy-in stream: page-locked frames (hs_host_alloc) go by DMA at link speed and the call returns at once; pageable frames go through the
    // runtime's staging path (the call returns when they are staged) — either way the compute stream keeps running the previous batch
    for (int i = 0; i < batch; i++) {
        if (pp) HIP_TRY(h, hipMemcpy2DAsync(sl->d_raw + raw_img * i, rpitch, imgs[i], stride, raw_row, src_h, hipMemcpyHostToDevice, h->s_in));
        else HIP_TRY(h, hipMemcpy2DAsync(sl->d_in + per_img * i, pitch, imgs[i], stride, w, h_px, hipMemcpyHostToDevice, h->s_in));
    }
    HIP_TRY(h, hipEventRecord(sl->ev_in, h->s_in));
    hipStream_t s = h->stream;
    HIP_TRY(h, hipStreamWaitEvent(s, sl->ev_in, 0));
    if (pp) {      // camera scale + grey on the compute stream, into the slot's level-0 frames
        hs_launch_preprocess(sl->d_raw, src_w, src_h, rpitch, raw_img, pp->channels, pp->rgb, pp->scale, sl->d_in, w, h_px, pitch, per_img, 1, batch, s);
        HIP_TRY(h, hipGetLastError());
    }
    int32_t* d_n = reinterpret_cast<int32_t*>(sl->d_out);
    hs_keypoint* d_k = reinterpret_cast<hs_keypoint*>(sl->d_out + sl->off_k);
    uint8_t* d_d = sl->d_out + sl->off_d;
    if (sp) {      // images [0, pairs) are the left frames, [pairs, 2 pairs) the right ones (hs_stereo_frontend_batch_device's layout)
        HsImg0 img0{ sl->d_in, sl->d_in + per_img * pairs, pairs, (uint64_t)pitch, (uint64_t)per_img };
        HsOut out{ d_k, d_d, d_n, d_k + (size_t)pairs * cap, d_d + (size_t)pairs * cap * HS_DESC_BYTES, d_n + pairs, pairs, cap };
        if (h->stereo_fuse) {
            const HsStripFuse sf{ 1, sp->n_rows, hs_stereo_strips(sp->n_rows), 0, sp->size_ref, h->d_strip_count, reinterpret_cast<HsStripEntry*>(h->d_strip_list) };
            rc = run_extract(h, img0, batch, out, s, &sf);
            if (rc != HS_OK) return rc;
            run_stereo_fused(h, out.kps, out.desc, out.n, out.kps2, out.desc2, out.n2, pairs, cap, *sp,
                             reinterpret_cast<float*>(sl->d_out + sl->off_u), reinterpret_cast<float*>(sl->d_out + sl->off_z), s);
        } else {
            rc = run_extract(h, img0, batch, out, s);
            if (rc != HS_OK) return rc;
            run_stereo(h, out.kps, out.desc, out.n, out.kps2, out.desc2, out.n2, pairs, cap, *sp,
                       reinterpret_cast<float*>(sl->d_out + sl->off_u), reinterpret_cast<float*>(sl->d_out + sl->off_z), s);
        }
        HIP_TRY(h, hipGetLastError());
    } else {
        HsImg0 img0{ sl->d_in, sl->d_in, batch, (uint64_t)pitch, (uint64_t)per_img };
        HsOut out{ d_k, d_d, d_n, d_k, d_d, d_n, batch, cap };
        rc = run_extract(h, img0, batch, out, s);
        if (rc != HS_OK) return rc;
    }
    HIP_TRY(h, hipEventRecord(sl->ev_done, s));
    HIP_TRY(h, hipStreamWaitEvent(h->s_out, sl->ev_done, 0));
    HIP_TRY(h, hipMemcpyAsync(sl->h_out, sl->d_out, sl->used, hipMemcpyDeviceToHost, h->s_out));
    HIP_TRY(h, hipEventRecord(sl->ev_out, h->s_out));
    return HS_OK;
    };
    rc = enqueue();
    if (rc != HS_OK) {
        const std::string why = h->err;
        (void)hipStreamSynchronize(h->s_in); (void)hipStreamSynchronize(h->stream); (void)hipStreamSynchronize(h->s_out); (void)hipGetLastError();
        h->err = why;
        return rc;
    }
    sl->busy = true; sl->batch = batch; sl->pairs = pairs; sl->cap = cap;
    sl->ticket = h->next_ticket++;
    if (h->next_ticket <= 0) h->next_ticket = 1;
    *ticket = sl->ticket;
    return HS_OK;
}

int hs_orb_submit_batch(hs_orb* h, const uint8_t* const* imgs, int batch, int w, int h_px, int stride, const hs_stereo_params* sp, int32_t* ticket)
{
    if (!h) return HS_ERR_INVALID;
    if (!imgs || !ticket || batch < 1 || batch > 65535 || w < 1 || h_px < 1 || stride < w || (sp && (batch & 1))) return fail(h, HS_ERR_INVALID, "bad argument");
    return submit_frames(h, imgs, batch, w, h_px, (size_t)stride, nullptr, sp, ticket);
}

int hs_orb_submit_camera_batch(hs_orb* h, const uint8_t* const* imgs, int batch, int w, int h_px, size_t row_stride, const hs_preprocess_params* pp,
                               const hs_stereo_params* sp, int32_t* ticket)
{
    if (!h) return HS_ERR_INVALID;
    if (!preprocess_params_ok(pp) || !imgs || !ticket || batch < 1 || batch > 65535 || w < 1 || h_px < 1 || w > 32768 || h_px > 32768 ||
        row_stride < (size_t)w * pp->channels || (sp && (batch & 1)))
        return fail(h, HS_ERR_INVALID, "bad argument");
    return submit_frames(h, imgs, batch, w, h_px, row_stride, pp, sp, ticket);
}

int hs_orb_wait(hs_orb* h, int32_t ticket, hs_keypoint* kps, uint8_t* desc, int32_t* n, int cap, float* uRight, float* depth)
{
    if (!h) return HS_ERR_INVALID;
    hs_orb::IngestSlot* sl = nullptr;
    for (auto& c : h->slot) if (c.busy && c.ticket == ticket) sl = &c;
    if (!sl || ticket <= 0) return fail(h, HS_ERR_INVALID, "unknown ticket");
    if (!kps || !desc || !n || (sl->pairs && (!uRight || !depth))) return fail(h, HS_ERR_INVALID, "bad argument");
    if (cap < sl->cap) return fail(h, HS_ERR_CAPACITY, "cap < keypoints this frame size can produce; see hs_orb_max_keypoints");
    HIP_TRY(h, hipSetDevice(h->device));
    {
        const hipError_t e = hipEventSynchronize(sl->ev_out);
        if (e != hipSuccess) {      // the results will never arrive: drain what can be drained and give the slot back (a ticket that fails must not block its slot for ever)
            (void)hipGetLastError();
            (void)hipStreamSynchronize(h->s_in); (void)hipStreamSynchronize(h->stream); (void)hipStreamSynchronize(h->s_out); (void)hipGetLastError();
            sl->busy = false;
            return fail(h, HS_ERR_HIP, std::string("hipEventSynchronize(ticket): ") + hipGetErrorString(e));
        }
    }
    const int B = sl->batch, c0 = sl->cap;
    memcpy(n, sl->h_out, (size_t)B * 4);
    for (int i = 0; i < B; i++) {            // only the keypoints that exist are copied, into the caller's [batch][cap] layout
        const size_t cnt = (size_t)std::min(std::max(n[i], 0), c0);
        memcpy(kps + (size_t)i * cap, sl->h_out + sl->off_k + (size_t)i * c0 * sizeof(hs_keypoint), cnt * sizeof(hs_keypoint));
        memcpy(desc + (size_t)i * cap * HS_DESC_BYTES, sl->h_out + sl->off_d + (size_t)i * c0 * HS_DESC_BYTES, cnt * HS_DESC_BYTES);
    }
    for (int i = 0; i < sl->pairs; i++) {
        const size_t cnt = (size_t)std::min(std::max(n[i], 0), c0);
        memcpy(uRight + (size_t)i * cap, sl->h_out + sl->off_u + (size_t)i * c0 * 4, cnt * 4);
        memcpy(depth + (size_t)i * cap, sl->h_out + sl->off_z + (size_t)i * c0 * 4, cnt * 4);
    }
    sl->busy = false;
    // (the slot's device block stays as it is until the slot is handed to another hs_orb_submit_batch)
    h->pub_kps = reinterpret_cast<const hs_keypoint*>(sl->d_out + sl->off_k); h->pub_desc = sl->d_out + sl->off_d; h->pub_cap = c0; h->pub_batch = B;
    return HS_OK;
}

int hs_orb_cancel(hs_orb* h, int32_t ticket)
{
    if (!h) return HS_ERR_INVALID;
    hs_orb::IngestSlot* sl = nullptr;
    for (auto& c : h->slot) if (c.busy && c.ticket == ticket) sl = &c;
    if (!sl || ticket <= 0) return fail(h, HS_ERR_INVALID, "unknown ticket");
    (void)hipSetDevice(h->device);
    if (hipEventSynchronize(sl->ev_out) != hipSuccess) {      // let the batch finish (its frames may be read until then), then drop the results
        (void)hipGetLastError();
        (void)hipStreamSynchronize(h->s_in); (void)hipStreamSynchronize(h->stream); (void)hipStreamSynchronize(h->s_out); (void)hipGetLastError();
    }
    sl->busy = false;
    return HS_OK;
}

int hs_ticket_frames_copied(hs_orb* h, int32_t ticket)
{
    if (!h) return HS_ERR_INVALID;
    for (auto& c : h->slot) if (c.busy && c.ticket == ticket) { const hipError_t e = hipEventQuery(c.ev_in); (void)hipGetLastError(); return e == hipSuccess ? 1 : 0; }
    return -1;
}

int hs_search_by_projection(hs_orb* h, const hs_frame_view* F, const hs_landmark* lms, int L, const hs_proj_params* pp,
                            int32_t* match_idx, float* match_dist, int32_t* n_matches)
{
    if (!h) return HS_ERR_INVALID;
    if (!F || !pp || L < 0 || !n_matches || (L > 0 && (!lms || !match_idx || !match_dist)) || F->n < 0 || F->n > 65535 ||
        (F->n > 0 && (!F->kps || !F->desc)) || (pp->use_stereo && F->sensor != 0 && F->n > 0 && !F->uR))
        return fail(h, HS_ERR_INVALID, "bad argument");
    *n_matches = 0;
    if (L == 0) return HS_OK;
    HIP_TRY(h, hipSetDevice(h->device));
    const int n = F->n;
    const size_t nn = (size_t)std::max(n, 1);
    int rc = scratch_begin(h, pad256(nn * sizeof(hs_keypoint)) + pad256(nn * 32) + 2 * pad256(nn * 4) + pad256(hs_frame_grid_bytes((int)nn)) + pad256(nn * 4) +
                              pad256((size_t)L * sizeof(hs_landmark)) + 3 * pad256((size_t)L * 4) + 256);
    if (rc != HS_OK) return rc;
    hipStream_t s = h->stream;
    hs_keypoint* d_kps = carve<hs_keypoint>(h, nn); uint8_t* d_desc = carve<uint8_t>(h, nn * 32);
    float* d_uR = carve<float>(h, nn); int32_t* d_obs = carve<int32_t>(h, nn); int8_t* d_cell = carve<int8_t>(h, hs_frame_grid_bytes((int)nn));
    int32_t* d_winner = carve<int32_t>(h, nn);
    hs_landmark* d_lms = carve<hs_landmark>(h, L);
    int32_t* d_midx = carve<int32_t>(h, L); float* d_mdist = carve<float>(h, L); float* d_pangle = carve<float>(h, L);
    int32_t* d_nm = carve<int32_t>(h, 1);
    if (n > 0) {
        HIP_TRY(h, hipMemcpyAsync(d_kps, F->kps, (size_t)n * sizeof(hs_keypoint), hipMemcpyHostToDevice, s));
        HIP_TRY(h, hipMemcpyAsync(d_desc, F->desc, (size_t)n * 32, hipMemcpyHostToDevice, s));
        if (F->uR) HIP_TRY(h, hipMemcpyAsync(d_uR, F->uR, (size_t)n * 4, hipMemcpyHostToDevice, s));
        if (F->kp_lm_obs) HIP_TRY(h, hipMemcpyAsync(d_obs, F->kp_lm_obs, (size_t)n * 4, hipMemcpyHostToDevice, s));
    }
    HIP_TRY(h, hipMemcpyAsync(d_lms, lms, (size_t)L * sizeof(hs_landmark), hipMemcpyHostToDevice, s));
    hs_launch_frame_grid(*F, d_kps, d_cell, true, s);
    hs_launch_search_projection(*F, d_kps, d_desc, F->uR ? d_uR : nullptr, F->kp_lm_obs ? d_obs : nullptr, d_cell, d_lms, L, *pp,
                                d_midx, d_mdist, d_winner, d_pangle, d_nm, s);
    HIP_TRY(h, hipGetLastError());
    HIP_TRY(h, hipMemcpyAsync(match_idx, d_midx, (size_t)L * 4, hipMemcpyDeviceToHost, s));
    HIP_TRY(h, hipMemcpyAsync(match_dist, d_mdist, (size_t)L * 4, hipMemcpyDeviceToHost, s));
    HIP_TRY(h, hipMemcpyAsync(n_matches, d_nm, 4, hipMemcpyDeviceToHost, s));
    HIP_TRY(h, hipStreamSynchronize(s));
    for (int i = 0; i < L; i++) if (match_idx[i] < 0) match_dist[i] = -1.f;      // entries dropped by the rotation check
    return HS_OK;
}

// ================= device-resident frames (FeatureViews stay in HBM between ImageProcessing and Tracking) =================
// hySLAM copies a frame's keypoints and descriptors out of the extractor into FeatureViews (host objects), and every matcher call gathers them
// again and uploads them (Frame.cc:45-72, FeatureViews.h:20-81).  The features were produced on this device a moment earlier: hs_frame_publish
// keeps a copy of one extracted frame in a small per-device cache (device-to-device, on the extractor's stream), hs_frame_find recognises a frame by
// its keypoint array (exact comparison with the host copy kept beside the slot — hySLAM has no field that could carry a token through FeatureViews),
// and the *_frame(s) entry points take the keypoints and descriptors from the cache instead of the host.  A slot is reused oldest first; a token
// whose slot was reused is simply unknown again (HS_ERR_INVALID) and the caller falls back to the host-pointer call.
namespace {
constexpr int HS_FRAME_SLOTS = 16;
struct FrameSlot {
    uint64_t token = 0, stamp = 0;     // token 0 = empty
    int n = 0, cap = 0, readers = 0;
    hs_keypoint* d_kps = nullptr; uint8_t* d_desc = nullptr;
    std::vector<hs_keypoint> h_kps;
    hipEvent_t ready = nullptr;        // recorded behind the copy that filled the slot
};
struct FrameCache { int device = 0; FrameSlot slot[HS_FRAME_SLOTS]; };
std::mutex g_frames_mu;
std::vector<FrameCache*> g_frames;     // one per device that ever published; never freed (process lifetime: static destructors must not call HIP)
uint64_t g_frame_serial = 0;
FrameCache* frame_cache_of(int device, bool create)
{
    for (FrameCache* c : g_frames) if (c->device == device) return c;
    if (!create) return nullptr;
    FrameCache* c = new FrameCache(); c->device = device; g_frames.push_back(c);
    return c;
}
struct FrameRef { FrameSlot* slot = nullptr; const hs_keypoint* d_kps = nullptr; const uint8_t* d_desc = nullptr; int n = 0; hipEvent_t ready = nullptr; };
bool frame_acquire(hs_frame_token tok, int device, FrameRef* r)
{
    std::lock_guard<std::mutex> g(g_frames_mu);
    FrameCache* c = frame_cache_of(device, false);
    if (!c || !tok) return false;
    for (FrameSlot& sl : c->slot) if (sl.token == tok) { sl.readers++; r->slot = &sl; r->d_kps = sl.d_kps; r->d_desc = sl.d_desc; r->n = sl.n; r->ready = sl.ready; return true; }
    return false;
}
void frame_release(FrameRef* r) { if (r->slot) { std::lock_guard<std::mutex> g(g_frames_mu); r->slot->readers--; r->slot = nullptr; } }
struct FrameGuard { FrameRef* a; FrameRef* b; ~FrameGuard() { if (a) frame_release(a); if (b) frame_release(b); } };
}

int hs_frame_publish(hs_orb* h, int image, const hs_keypoint* kps, int n, hs_frame_token* token)
{
    if (!h) return HS_ERR_INVALID;
    if (token) *token = 0;
    if (!token || !kps || n < 1 || n > 65535 || image < 0) return fail(h, HS_ERR_INVALID, "bad argument");
    if (!h->pub_kps || image >= h->pub_batch || n > h->pub_cap) return fail(h, HS_ERR_INVALID, "hs_frame_publish: no host-pointer extraction result of this handle to publish (call right after hs_orb_extract / hs_orb_extract_batch / hs_orb_wait)");
    HIP_TRY(h, hipSetDevice(h->device));
    // The slot is picked and RESERVED under the process-wide lock (readers = -1: neither a publisher nor a reader nor hs_frame_cache_clear touches it),
    // the HIP calls (event wait, a possible free + allocation, two copy enqueues) run without it — hs_frame_find / frame_acquire / hs_frame_release of other
    // threads (the tracking thread's SearchByProjection) never wait behind an extractor thread's allocation — and the slot is published under the lock again.
    FrameSlot* sl = nullptr;
    {
        std::lock_guard<std::mutex> g(g_frames_mu);
        FrameCache* c = frame_cache_of(h->device, true);
        for (FrameSlot& q : c->slot) if (q.readers == 0 && (!sl || (q.token == 0 && sl->token != 0) || ((q.token == 0) == (sl->token == 0) && q.stamp < sl->stamp))) sl = &q;
        if (!sl) return fail(h, HS_ERR_CAPACITY, "hs_frame_publish: every cache slot is being read");
        sl->token = 0;
        sl->readers = -1;
    }
    struct Unreserve { FrameSlot* s; ~Unreserve() { if (s) { std::lock_guard<std::mutex> g(g_frames_mu); s->readers = 0; } } } unreserve{sl};      // failure paths: the slot is empty (token 0) and free again
    if (!sl->ready) HIP_TRY(h, hipEventCreateWithFlags(&sl->ready, hipEventDisableTiming));
    if (n > sl->cap) {
        HIP_TRY(h, hipEventSynchronize(sl->ready));      // (a never-recorded event is complete)
        hipFree(sl->d_kps); hipFree(sl->d_desc); sl->d_kps = nullptr; sl->d_desc = nullptr; sl->cap = 0;
        const int grow = std::max(n, 2048);
        HIP_TRY(h, hipMalloc(&sl->d_kps, (size_t)grow * sizeof(hs_keypoint)));
        HIP_TRY(h, hipMalloc(&sl->d_desc, (size_t)grow * HS_DESC_BYTES));
        sl->cap = grow;
    }
    hipStream_t s = h->stream;
    HIP_TRY(h, hipStreamWaitEvent(s, sl->ready, 0));     // the copy that filled the slot last time (another handle's stream) comes first
    HIP_TRY(h, hipMemcpyAsync(sl->d_kps, h->pub_kps + (size_t)image * h->pub_cap, (size_t)n * sizeof(hs_keypoint), hipMemcpyDeviceToDevice, s));
    HIP_TRY(h, hipMemcpyAsync(sl->d_desc, h->pub_desc + (size_t)image * h->pub_cap * HS_DESC_BYTES, (size_t)n * HS_DESC_BYTES, hipMemcpyDeviceToDevice, s));
    HIP_TRY(h, hipEventRecord(sl->ready, s));
    sl->h_kps.assign(kps, kps + n);                      // (the slot is reserved: nobody compares against h_kps while token == 0)
    {
        std::lock_guard<std::mutex> g(g_frames_mu);
        unreserve.s = nullptr;
        sl->readers = 0;
        sl->n = n; sl->stamp = ++g_frame_serial; sl->token = sl->stamp;
        *token = sl->token;
    }
    return HS_OK;
}

int hs_frame_find(int device, const hs_keypoint* kps, int n, hs_frame_token* token)
{
    if (!token) return HS_ERR_INVALID;
    *token = 0;
    if (!kps || n < 1) return HS_ERR_INVALID;
    std::lock_guard<std::mutex> g(g_frames_mu);
    FrameCache* c = frame_cache_of(device, false);
    if (!c) return HS_ERR_INVALID;
    const FrameSlot* best = nullptr;
    for (const FrameSlot& sl : c->slot)
        if (sl.token && sl.n == n && (!best || sl.stamp > best->stamp) && memcmp(sl.h_kps.data(), kps, (size_t)n * sizeof(hs_keypoint)) == 0) best = &sl;
    if (!best) return HS_ERR_INVALID;
    *token = best->token;
    return HS_OK;
}

int hs_frame_release(int device, hs_frame_token token)
{
    std::lock_guard<std::mutex> g(g_frames_mu);
    FrameCache* c = frame_cache_of(device, false);
    if (!c || !token) return HS_ERR_INVALID;
    for (FrameSlot& sl : c->slot) if (sl.token == token) { sl.token = 0; return HS_OK; }      // (a slot that is being read keeps its buffers until the reader is done: readers > 0 keeps it from being refilled)
    return HS_ERR_INVALID;
}

int hs_frame_cache_clear(int device)
{
    std::lock_guard<std::mutex> g(g_frames_mu);
    for (size_t i = 0; i < g_frames.size(); i++) {
        FrameCache* c = g_frames[i];
        if (c->device != device) continue;
        for (const FrameSlot& sl : c->slot) if (sl.readers != 0) return HS_ERR_INVALID;      // a call is reading a slot (> 0) or a publisher is filling one (-1): not now
        int cur = -1;
        (void)hipGetDevice(&cur);
        (void)hipSetDevice(device);
        for (FrameSlot& sl : c->slot) {
            if (sl.ready) { (void)hipEventSynchronize(sl.ready); (void)hipEventDestroy(sl.ready); }
            (void)hipFree(sl.d_kps); (void)hipFree(sl.d_desc);
        }
        if (cur >= 0) (void)hipSetDevice(cur);
        delete c;
        g_frames.erase(g_frames.begin() + (long)i);
        return HS_OK;
    }
    return HS_OK;      // nothing was ever published on that device
}

int hs_frame_info(int device, hs_frame_token token, int32_t* n)
{
    std::lock_guard<std::mutex> g(g_frames_mu);
    FrameCache* c = frame_cache_of(device, false);
    if (!c || !token) return HS_ERR_INVALID;
    for (const FrameSlot& sl : c->slot) if (sl.token == token) { if (n) *n = sl.n; return HS_OK; }
    return HS_ERR_INVALID;
}

int hs_search_by_projection_frame(hs_orb* h, hs_frame_token frame, const hs_frame_view* F, const hs_landmark* lms, int L, const hs_proj_params* pp,
                                  int32_t* match_idx, float* match_dist, int32_t* n_matches)
{
    if (!h) return HS_ERR_INVALID;
    if (!F || !pp || L < 0 || !n_matches || (L > 0 && (!lms || !match_idx || !match_dist)) || F->n < 1 || F->n > 65535 ||
        (pp->use_stereo && F->sensor != 0 && !F->uR))
        return fail(h, HS_ERR_INVALID, "bad argument");
    *n_matches = 0;
    FrameRef ref;
    if (!frame_acquire(frame, h->device, &ref)) return fail(h, HS_ERR_INVALID, "hs_search_by_projection_frame: unknown frame token (its cache slot was reused, or it lives on another device)");
    FrameGuard guard{ &ref, nullptr };
    if (ref.n != F->n) return fail(h, HS_ERR_INVALID, "hs_search_by_projection_frame: F->n differs from the published frame");
    if (L == 0) return HS_OK;
    HIP_TRY(h, hipSetDevice(h->device));
    const int n = F->n;
    const size_t nn = (size_t)n;
    int rc = scratch_begin(h, 2 * pad256(nn * 4) + pad256(hs_frame_grid_bytes((int)nn)) + pad256(nn * 4) + pad256((size_t)L * sizeof(hs_landmark)) + 3 * pad256((size_t)L * 4) + 256);
    if (rc != HS_OK) return rc;
    hipStream_t s = h->stream;
    float* d_uR = carve<float>(h, nn); int32_t* d_obs = carve<int32_t>(h, nn); int8_t* d_cell = carve<int8_t>(h, hs_frame_grid_bytes((int)nn));
    int32_t* d_winner = carve<int32_t>(h, nn);
    hs_landmark* d_lms = carve<hs_landmark>(h, L);
    int32_t* d_midx = carve<int32_t>(h, L); float* d_mdist = carve<float>(h, L); float* d_pangle = carve<float>(h, L);
    int32_t* d_nm = carve<int32_t>(h, 1);
    if (F->uR) HIP_TRY(h, hipMemcpyAsync(d_uR, F->uR, nn * 4, hipMemcpyHostToDevice, s));
    if (F->kp_lm_obs) HIP_TRY(h, hipMemcpyAsync(d_obs, F->kp_lm_obs, nn * 4, hipMemcpyHostToDevice, s));
    HIP_TRY(h, hipMemcpyAsync(d_lms, lms, (size_t)L * sizeof(hs_landmark), hipMemcpyHostToDevice, s));
    HIP_TRY(h, hipStreamWaitEvent(s, ref.ready, 0));
    hs_launch_frame_grid(*F, ref.d_kps, d_cell, true, s);
    hs_launch_search_projection(*F, ref.d_kps, ref.d_desc, F->uR ? d_uR : nullptr, F->kp_lm_obs ? d_obs : nullptr, d_cell, d_lms, L, *pp,
                                d_midx, d_mdist, d_winner, d_pangle, d_nm, s);
    HIP_TRY(h, hipGetLastError());
    HIP_TRY(h, hipMemcpyAsync(match_idx, d_midx, (size_t)L * 4, hipMemcpyDeviceToHost, s));
    HIP_TRY(h, hipMemcpyAsync(match_dist, d_mdist, (size_t)L * 4, hipMemcpyDeviceToHost, s));
    HIP_TRY(h, hipMemcpyAsync(n_matches, d_nm, 4, hipMemcpyDeviceToHost, s));
    HIP_TRY(h, hipStreamSynchronize(s));
    for (int i = 0; i < L; i++) if (match_idx[i] < 0) match_dist[i] = -1.f;      // entries dropped by the rotation check
    return HS_OK;
}

int hs_stereo_match_frames(hs_orb* h, hs_frame_token left, hs_frame_token right, const hs_stereo_params* sp, float* uRight, float* depth)
{
    if (!h) return HS_ERR_INVALID;
    if (!sp || !uRight || !depth) return fail(h, HS_ERR_INVALID, "bad argument");
    FrameRef L, R;
    if (!frame_acquire(left, h->device, &L)) return fail(h, HS_ERR_INVALID, "hs_stereo_match_frames: unknown left frame token");
    FrameGuard guard{ &L, nullptr };
    if (!frame_acquire(right, h->device, &R)) return fail(h, HS_ERR_INVALID, "hs_stereo_match_frames: unknown right frame token");
    guard.b = &R;
    HIP_TRY(h, hipSetDevice(h->device));
    const int nL = L.n, nR = R.n, cap = std::max(nL, nR);
    hipStream_t s = h->stream;
    if (!h->d_sm_n) HIP_TRY(h, hipMalloc(&h->d_sm_n, 8));
    const size_t pin_need = 16 + 2 * (size_t)cap * 4;
    if (pin_need > h->pin_bytes) {
        HIP_TRY(h, hipStreamSynchronize(s));
        if (h->h_pin) hipHostFree(h->h_pin);
        h->h_pin = nullptr; h->pin_bytes = 0;
        const size_t grow = std::max<size_t>(pin_need, 1 << 18);
        HIP_TRY(h, hipHostMalloc(&h->h_pin, grow, hipHostMallocDefault));
        h->pin_bytes = grow;
    }
    int rc = ensure_stereo_scratch(h, (size_t)cap);
    if (rc == HS_OK) rc = ensure_stereo_strips(h, 1, cap, sp->n_rows);
    if (rc != HS_OK) return rc;
    int32_t* pn = reinterpret_cast<int32_t*>(h->h_pin); float* pout = reinterpret_cast<float*>(h->h_pin + 16);
    pn[0] = nL; pn[1] = nR;
    HIP_TRY(h, hipMemcpyAsync(h->d_sm_n, pn, 8, hipMemcpyHostToDevice, s));
    HIP_TRY(h, hipStreamWaitEvent(s, L.ready, 0));
    HIP_TRY(h, hipStreamWaitEvent(s, R.ready, 0));
    run_stereo(h, L.d_kps, L.d_desc, h->d_sm_n, R.d_kps, R.d_desc, h->d_sm_n + 1, 1, cap, *sp, h->d_ur, h->d_depth, s);
    HIP_TRY(h, hipGetLastError());
    HIP_TRY(h, hipMemcpyAsync(pout, h->d_ur, (size_t)nL * 4, hipMemcpyDeviceToHost, s));
    HIP_TRY(h, hipMemcpyAsync(pout + cap, h->d_depth, (size_t)nL * 4, hipMemcpyDeviceToHost, s));
    HIP_TRY(h, hipStreamSynchronize(s));
    memcpy(uRight, pout, (size_t)nL * 4);
    memcpy(depth, pout + cap, (size_t)nL * 4);
    return HS_OK;
}

int hs_frame_grid(hs_orb* h, const hs_frame_view* F, int8_t* cell_xy)
{
    if (!h) return HS_ERR_INVALID;
    if (!F || F->n < 0 || F->n > 65535 || (F->n > 0 && (!F->kps || !cell_xy))) return fail(h, HS_ERR_INVALID, "bad argument");
    if (F->n == 0) return HS_OK;
    HIP_TRY(h, hipSetDevice(h->device));
    const size_t nn = F->n;
    int rc = scratch_begin(h, pad256(nn * sizeof(hs_keypoint)) + pad256(hs_frame_grid_bytes((int)nn)) + 256);
    if (rc != HS_OK) return rc;
    hipStream_t s = h->stream;
    hs_keypoint* d_kps = carve<hs_keypoint>(h, nn); int8_t* d_cell = carve<int8_t>(h, hs_frame_grid_bytes((int)nn));
    HIP_TRY(h, hipMemcpyAsync(d_kps, F->kps, nn * sizeof(hs_keypoint), hipMemcpyHostToDevice, s));
    hs_launch_frame_grid(*F, d_kps, d_cell, false, s);
    HIP_TRY(h, hipGetLastError());
    HIP_TRY(h, hipMemcpyAsync(cell_xy, d_cell, nn * 2, hipMemcpyDeviceToHost, s));
    HIP_TRY(h, hipStreamSynchronize(s));
    return HS_OK;
}

int hs_search_by_projection_device(hs_orb* h, const hs_frame_view* F, const hs_landmark* d_lms, int L, const hs_proj_params* pp,
                                   int32_t* d_match_idx, float* d_match_dist, int32_t* d_n_matches, void* stream)
{
    if (!h) return HS_ERR_INVALID;
    if (!F || !pp || L < 1 || !d_lms || !d_match_idx || !d_match_dist || !d_n_matches || F->n < 1 || F->n > 65535 || !F->kps || !F->desc ||
        (pp->use_stereo && F->sensor != 0 && !F->uR))
        return fail(h, HS_ERR_INVALID, "bad argument");
    HIP_TRY(h, hipSetDevice(h->device));
    const size_t nn = F->n;
    // scratch is per handle: a second call may only start after the first finished (same stream ordering is enough)
    const size_t need = pad256(hs_frame_grid_bytes((int)nn)) + pad256(nn * 4) + pad256((size_t)L * 4) + 4096;
    if (need > h->scratch_bytes) { int rc = scratch_begin(h, need); if (rc != HS_OK) return rc; }
    h->scratch_used = 0;
    int8_t* d_cell = carve<int8_t>(h, hs_frame_grid_bytes((int)nn)); int32_t* d_winner = carve<int32_t>(h, nn); float* d_pangle = carve<float>(h, L);
    hipStream_t s = stream ? (hipStream_t)stream : h->stream;
    hs_launch_frame_grid(*F, F->kps, d_cell, true, s);
    hs_launch_search_projection(*F, F->kps, F->desc, F->uR, F->kp_lm_obs, d_cell, d_lms, L, *pp, d_match_idx, d_match_dist, d_winner, d_pangle, d_n_matches, s);
    HIP_TRY(h, hipGetLastError());
    return HS_OK;
}

namespace {
// uploads what the Sim3 matchers read of a keyframe and builds its grid lists; returns the carved device pointers
struct DevFrame { hs_keypoint* kps; uint8_t* desc; int8_t* cell; };
int upload_frame(hs_orb* h, const hs_frame_view* F, hipStream_t s, DevFrame* o)
{
    const size_t nn = (size_t)std::max(F->n, 1);
    o->kps = carve<hs_keypoint>(h, nn); o->desc = carve<uint8_t>(h, nn * 32); o->cell = carve<int8_t>(h, hs_frame_grid_bytes((int)nn));
    if (F->n > 0) {
        HIP_TRY(h, hipMemcpyAsync(o->kps, F->kps, (size_t)F->n * sizeof(hs_keypoint), hipMemcpyHostToDevice, s));
        HIP_TRY(h, hipMemcpyAsync(o->desc, F->desc, (size_t)F->n * 32, hipMemcpyHostToDevice, s));
        hs_launch_frame_grid(*F, o->kps, o->cell, true, s);
    }
    return HS_OK;
}
size_t frame_bytes(int n) { const size_t nn = (size_t)std::max(n, 1); return pad256(nn * sizeof(hs_keypoint)) + pad256(nn * 32) + pad256(hs_frame_grid_bytes((int)nn)); }
// one row of A*B (+c): double accumulation, alpha in double, one rounding (cv::gemm on float matrices)
float gemm3h(const float* A, const float* B, float c, double alpha = 1.0)
{
    double s = 0;
    for (int k = 0; k < 3; k++) s += (double)A[k] * (double)B[k];
    return (float)(alpha * s + (double)c);
}
}

int hs_search_by_projection_sim3(hs_orb* h, const hs_frame_view* KF, const float* Scw, const hs_landmark* lms, int L, int th, float th_low,
                                 uint8_t* kp_matched, int32_t* match_idx, int32_t* n_matches)
{
    if (!h) return HS_ERR_INVALID;
    if (!KF || !Scw || L < 0 || !n_matches || (L > 0 && (!lms || !match_idx)) || KF->n < 0 || KF->n > 65535 || (KF->n > 0 && (!KF->kps || !KF->desc || !kp_matched)))
        return fail(h, HS_ERR_INVALID, "bad argument");
    *n_matches = 0;
    for (int i = 0; i < L; i++) match_idx[i] = -1;
    if (L == 0 || KF->n == 0) return HS_OK;
    // Decompose Scw like the reference (FeatureMatcher.cc:641-646): scw = |row 0| (double accumulation), Rcw = sRcw/scw and tcw = t/scw are
    // cv::Mat scalings (every element times (float)(1/scw) in float), Ow = -Rcw.t()*tcw one gemm
    float R[9], t[3], Ow[3];
    const float scw = (float)std::sqrt((double)Scw[0] * Scw[0] + (double)Scw[1] * Scw[1] + (double)Scw[2] * Scw[2]);
    const float inv = (float)(1.0 / (double)scw);
    for (int r = 0; r < 3; r++) { for (int c = 0; c < 3; c++) R[3 * r + c] = Scw[4 * r + c] * inv + 0.0f; t[r] = Scw[4 * r + 3] * inv + 0.0f; }
    for (int i = 0; i < 3; i++) { const float col[3] = { R[i], R[3 + i], R[6 + i] }; Ow[i] = gemm3h(col, t, 0.f, -1.0); }
    HIP_TRY(h, hipSetDevice(h->device));
    int rc = scratch_begin(h, frame_bytes(KF->n) + pad256((size_t)L * sizeof(hs_landmark)) + pad256((size_t)L * 12) + pad256((size_t)KF->n) + pad256((size_t)L * 4) + 512);
    if (rc != HS_OK) return rc;
    hipStream_t s = h->stream;
    DevFrame D; rc = upload_frame(h, KF, s, &D);
    if (rc != HS_OK) return rc;
    hs_landmark* d_lms = carve<hs_landmark>(h, L); float* d_geo = carve<float>(h, (size_t)L * 3);
    uint8_t* d_taken = carve<uint8_t>(h, KF->n); int32_t* d_midx = carve<int32_t>(h, L); int32_t* d_n = carve<int32_t>(h, 1);
    HIP_TRY(h, hipMemcpyAsync(d_lms, lms, (size_t)L * sizeof(hs_landmark), hipMemcpyHostToDevice, s));
    HIP_TRY(h, hipMemcpyAsync(d_taken, kp_matched, KF->n, hipMemcpyHostToDevice, s));
    hs_launch_sim3_projection(*KF, D.kps, D.desc, D.cell, R, t, Ow, d_lms, L, (float)th, th_low, d_geo, d_taken, d_midx, d_n, s);
    HIP_TRY(h, hipGetLastError());
    HIP_TRY(h, hipMemcpyAsync(match_idx, d_midx, (size_t)L * 4, hipMemcpyDeviceToHost, s));
    HIP_TRY(h, hipMemcpyAsync(kp_matched, d_taken, KF->n, hipMemcpyDeviceToHost, s));
    HIP_TRY(h, hipMemcpyAsync(n_matches, d_n, 4, hipMemcpyDeviceToHost, s));
    HIP_TRY(h, hipStreamSynchronize(s));
    return HS_OK;
}

int hs_search_by_sim3(hs_orb* h, const hs_frame_view* KF1, const hs_landmark* lms1, const hs_frame_view* KF2, const hs_landmark* lms2,
                      float s12, const float* R12, const float* t12, float th, float th_high, int32_t* match12, int32_t* n_found)
{
    if (!h) return HS_ERR_INVALID;
    if (!KF1 || !KF2 || !R12 || !t12 || !n_found || KF1->n < 0 || KF2->n < 0 || KF1->n > 65535 || KF2->n > 65535 ||
        (KF1->n > 0 && (!KF1->kps || !KF1->desc || !lms1 || !match12)) || (KF2->n > 0 && (!KF2->kps || !KF2->desc || !lms2)))
        return fail(h, HS_ERR_INVALID, "bad argument");
    *n_found = 0;
    for (int i = 0; i < KF1->n; i++) match12[i] = -1;
    if (KF1->n == 0 || KF2->n == 0) return HS_OK;
    // Transformation between cameras (FeatureMatcher.cc:757-760): sR12 = s12*R12, sR21 = (1/s12)*R12.t() (cv::Mat scalings), t21 = -sR21*t12 (gemm)
    float sR12[9], sR21[9], t21[3];
    const float a12 = (float)(double)s12, a21 = (float)(1.0 / (double)s12);
    for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) { sR12[3 * r + c] = R12[3 * r + c] * a12 + 0.0f; sR21[3 * r + c] = R12[3 * c + r] * a21 + 0.0f; }
    for (int i = 0; i < 3; i++) t21[i] = gemm3h(&sR21[3 * i], t12, 0.f, -1.0);
    HIP_TRY(h, hipSetDevice(h->device));
    const int n1 = KF1->n, n2 = KF2->n;
    int rc = scratch_begin(h, frame_bytes(n1) + frame_bytes(n2) + pad256((size_t)n1 * sizeof(hs_landmark)) + pad256((size_t)n2 * sizeof(hs_landmark)) +
                              2 * pad256((size_t)n1 * 4) + pad256((size_t)n2 * 4) + 512);
    if (rc != HS_OK) return rc;
    hipStream_t s = h->stream;
    DevFrame D1, D2;
    rc = upload_frame(h, KF1, s, &D1); if (rc != HS_OK) return rc;
    rc = upload_frame(h, KF2, s, &D2); if (rc != HS_OK) return rc;
    hs_landmark* d_l1 = carve<hs_landmark>(h, n1); hs_landmark* d_l2 = carve<hs_landmark>(h, n2);
    int32_t* d_m1 = carve<int32_t>(h, n1); int32_t* d_m12 = carve<int32_t>(h, n1); int32_t* d_m2 = carve<int32_t>(h, n2); int32_t* d_n = carve<int32_t>(h, 1);
    HIP_TRY(h, hipMemcpyAsync(d_l1, lms1, (size_t)n1 * sizeof(hs_landmark), hipMemcpyHostToDevice, s));
    HIP_TRY(h, hipMemcpyAsync(d_l2, lms2, (size_t)n2 * sizeof(hs_landmark), hipMemcpyHostToDevice, s));
    hs_launch_sim3_search(*KF1, D1.kps, D1.desc, D1.cell, *KF2, D2.kps, D2.desc, D2.cell, d_l1, d_l2, sR21, t21, sR12, t12, th, th_high, d_m1, d_m2, d_m12, d_n, s);
    HIP_TRY(h, hipGetLastError());
    HIP_TRY(h, hipMemcpyAsync(match12, d_m12, (size_t)n1 * 4, hipMemcpyDeviceToHost, s));
    HIP_TRY(h, hipMemcpyAsync(n_found, d_n, 4, hipMemcpyDeviceToHost, s));
    HIP_TRY(h, hipStreamSynchronize(s));
    return HS_OK;
}

int hs_search_by_bow(hs_orb* h, const hs_keypoint* kps1, const uint8_t* desc1, int n1,
                     const int32_t* node_id1, const int32_t* node_ptr1, const int32_t* idx1, int nn1,
                     const hs_keypoint* kps2, const uint8_t* desc2, int n2,
                     const int32_t* node_id2, const int32_t* node_ptr2, const int32_t* idx2, int nn2,
                     const uint8_t* keep1, float score_threshold, float second_best_ratio, int check_rotation,
                     int32_t* match12, int32_t* n_matches)
{
    return hs_search_by_bow_ex(h, kps1, desc1, n1, node_id1, node_ptr1, idx1, nn1, kps2, desc2, n2, node_id2, node_ptr2, idx2, nn2,
                               keep1, nullptr, nullptr, 31.f, 1.f, score_threshold, second_best_ratio, check_rotation, match12, n_matches);
}

static int bow_host(hs_orb* h, int legacy, const hs_keypoint* kps1, const uint8_t* desc1, int n1,
                        const int32_t* node_id1, const int32_t* node_ptr1, const int32_t* idx1, int nn1,
                        const hs_keypoint* kps2, const uint8_t* desc2, int n2,
                        const int32_t* node_id2, const int32_t* node_ptr2, const int32_t* idx2, int nn2,
                        const uint8_t* keep1, const uint8_t* keep2, const float* F12, float size_ref, float sigma_ref,
                        float score_threshold, float second_best_ratio, int check_rotation,
                        int32_t* match12, int32_t* n_matches)
{
    if (!h) return HS_ERR_INVALID;
    if (n1 < 0 || n2 < 0 || nn1 < 0 || nn2 < 0 || !n_matches || (n1 > 0 && (!kps1 || !desc1 || !match12)) || (n2 > 0 && (!kps2 || !desc2)) ||
        (nn1 > 0 && (!node_id1 || !node_ptr1 || !idx1)) || (nn2 > 0 && (!node_id2 || !node_ptr2 || !idx2)))
        return fail(h, HS_ERR_INVALID, "bad argument");
    *n_matches = 0;
    for (int i = 0; i < n1; i++) match12[i] = -1;
    if (n1 == 0 || n2 == 0 || nn1 == 0 || nn2 == 0) return HS_OK;
    // merge-walk of the two DBoW2::FeatureVector maps (FeatureMatcher.cc:230-265): nodes present on both sides
    std::vector<int32_t> pa, pb;
    for (int a = 0, b = 0; a < nn1 && b < nn2;) {
        if (node_id1[a] == node_id2[b]) { pa.push_back(a++); pb.push_back(b++); }
        else if (node_id1[a] < node_id2[b]) a++; else b++;
    }
    const int np = (int)pa.size();
    const int m1 = node_ptr1[nn1], m2 = node_ptr2[nn2];
    for (int i = 0; i < m1; i++) if (idx1[i] < 0 || idx1[i] >= n1) return fail(h, HS_ERR_INVALID, "feature vector index out of range");
    for (int i = 0; i < m2; i++) if (idx2[i] < 0 || idx2[i] >= n2) return fail(h, HS_ERR_INVALID, "feature vector index out of range");
    HIP_TRY(h, hipSetDevice(h->device));
    int rc = scratch_begin(h, pad256((size_t)n1 * sizeof(hs_keypoint)) + pad256((size_t)n2 * sizeof(hs_keypoint)) + pad256((size_t)n1 * 32) + pad256((size_t)n2 * 32) +
                              pad256((size_t)(nn1 + 1) * 4) + pad256((size_t)(nn2 + 1) * 4) + pad256((size_t)std::max(m1, 1) * 4) + pad256((size_t)std::max(m2, 1) * 4) +
                              2 * pad256((size_t)std::max(np, 1) * 4) + pad256(n1) + pad256(n2) + 3 * pad256((size_t)n1 * 4) + pad256((size_t)n2 * 4) + 256);
    if (rc != HS_OK) return rc;
    hipStream_t s = h->stream;
    hs_keypoint* d_k1 = carve<hs_keypoint>(h, n1); hs_keypoint* d_k2 = carve<hs_keypoint>(h, n2);
    uint8_t* d_d1 = carve<uint8_t>(h, (size_t)n1 * 32); uint8_t* d_d2 = carve<uint8_t>(h, (size_t)n2 * 32);
    int32_t* d_p1 = carve<int32_t>(h, nn1 + 1); int32_t* d_p2 = carve<int32_t>(h, nn2 + 1);
    int32_t* d_i1 = carve<int32_t>(h, std::max(m1, 1)); int32_t* d_i2 = carve<int32_t>(h, std::max(m2, 1));
    int32_t* d_pa = carve<int32_t>(h, std::max(np, 1)); int32_t* d_pb = carve<int32_t>(h, std::max(np, 1));
    uint8_t* d_keep = carve<uint8_t>(h, n1); uint8_t* d_keep2 = carve<uint8_t>(h, n2);
    int32_t* d_m = carve<int32_t>(h, n1); float* d_ang = carve<float>(h, n1); int32_t* d_self = carve<int32_t>(h, n1);
    int32_t* d_nm = carve<int32_t>(h, 1);
    uint32_t* d_taken2 = carve<uint32_t>(h, n2);
    HIP_TRY(h, hipMemcpyAsync(d_k1, kps1, (size_t)n1 * sizeof(hs_keypoint), hipMemcpyHostToDevice, s));
    HIP_TRY(h, hipMemcpyAsync(d_k2, kps2, (size_t)n2 * sizeof(hs_keypoint), hipMemcpyHostToDevice, s));
    HIP_TRY(h, hipMemcpyAsync(d_d1, desc1, (size_t)n1 * 32, hipMemcpyHostToDevice, s));
    HIP_TRY(h, hipMemcpyAsync(d_d2, desc2, (size_t)n2 * 32, hipMemcpyHostToDevice, s));
    HIP_TRY(h, hipMemcpyAsync(d_p1, node_ptr1, (size_t)(nn1 + 1) * 4, hipMemcpyHostToDevice, s));
    HIP_TRY(h, hipMemcpyAsync(d_p2, node_ptr2, (size_t)(nn2 + 1) * 4, hipMemcpyHostToDevice, s));
    if (m1) HIP_TRY(h, hipMemcpyAsync(d_i1, idx1, (size_t)m1 * 4, hipMemcpyHostToDevice, s));
    if (m2) HIP_TRY(h, hipMemcpyAsync(d_i2, idx2, (size_t)m2 * 4, hipMemcpyHostToDevice, s));
    if (np) { HIP_TRY(h, hipMemcpyAsync(d_pa, pa.data(), (size_t)np * 4, hipMemcpyHostToDevice, s)); HIP_TRY(h, hipMemcpyAsync(d_pb, pb.data(), (size_t)np * 4, hipMemcpyHostToDevice, s)); }
    if (keep1) HIP_TRY(h, hipMemcpyAsync(d_keep, keep1, n1, hipMemcpyHostToDevice, s));
    if (keep2) HIP_TRY(h, hipMemcpyAsync(d_keep2, keep2, n2, hipMemcpyHostToDevice, s));
    if (legacy)
        hs_launch_bow_legacy(d_pa, d_pb, np, d_p1, d_i1, d_p2, d_i2, d_d1, d_d2, keep1 ? d_keep : nullptr, keep2 ? d_keep2 : nullptr,
                             score_threshold, second_best_ratio, d_m, n1, n2, d_k1, d_k2, d_ang, check_rotation, d_self, d_taken2, d_nm, s);
    else
        hs_launch_bow(d_pa, d_pb, np, d_p1, d_i1, d_p2, d_i2, d_d1, d_d2, keep1 ? d_keep : nullptr, keep2 ? d_keep2 : nullptr,
                      F12, size_ref, sigma_ref, score_threshold, second_best_ratio,
                      d_m, n1, d_k1, d_k2, d_ang, check_rotation, d_self, d_nm, s);
    HIP_TRY(h, hipGetLastError());
    HIP_TRY(h, hipMemcpyAsync(match12, d_m, (size_t)n1 * 4, hipMemcpyDeviceToHost, s));
    HIP_TRY(h, hipMemcpyAsync(n_matches, d_nm, 4, hipMemcpyDeviceToHost, s));
    HIP_TRY(h, hipStreamSynchronize(s));
    return HS_OK;
}

int hs_search_by_bow_ex(hs_orb* h, const hs_keypoint* kps1, const uint8_t* desc1, int n1,
                        const int32_t* node_id1, const int32_t* node_ptr1, const int32_t* idx1, int nn1,
                        const hs_keypoint* kps2, const uint8_t* desc2, int n2,
                        const int32_t* node_id2, const int32_t* node_ptr2, const int32_t* idx2, int nn2,
                        const uint8_t* keep1, const uint8_t* keep2, const float* F12, float size_ref, float sigma_ref,
                        float score_threshold, float second_best_ratio, int check_rotation,
                        int32_t* match12, int32_t* n_matches)
{
    return bow_host(h, 0, kps1, desc1, n1, node_id1, node_ptr1, idx1, nn1, kps2, desc2, n2, node_id2, node_ptr2, idx2, nn2, keep1, keep2, F12, size_ref, sigma_ref,
                    score_threshold, second_best_ratio, check_rotation, match12, n_matches);
}

int hs_search_by_bow_legacy(hs_orb* h, const hs_keypoint* kps1, const uint8_t* desc1, int n1,
                            const int32_t* node_id1, const int32_t* node_ptr1, const int32_t* idx1, int nn1,
                            const hs_keypoint* kps2, const uint8_t* desc2, int n2,
                            const int32_t* node_id2, const int32_t* node_ptr2, const int32_t* idx2, int nn2,
                            const uint8_t* keep1, const uint8_t* keep2, float th_low, float nnratio, int check_orientation,
                            int32_t* match12, int32_t* n_matches)
{
    return bow_host(h, 1, kps1, desc1, n1, node_id1, node_ptr1, idx1, nn1, kps2, desc2, n2, node_id2, node_ptr2, idx2, nn2, keep1, keep2, nullptr, 31.f, 1.f,
                    th_low, nnratio, check_orientation, match12, n_matches);
}

int hs_search_for_initialization(hs_orb* h, const hs_keypoint* kps1, const uint8_t* desc1, int n1, const hs_frame_view* F2,
                                 float* prev_matched_xy, int window, float th_low, float nnratio, int32_t* matches12, int32_t* n_matches)
{
    if (!h) return HS_ERR_INVALID;
    if (!F2 || n1 < 0 || !n_matches || F2->n < 0 || F2->n > 65535 || (n1 > 0 && (!kps1 || !desc1 || !prev_matched_xy || !matches12)) ||
        (F2->n > 0 && (!F2->kps || !F2->desc)))
        return fail(h, HS_ERR_INVALID, "bad argument");
    *n_matches = 0;
    for (int i = 0; i < n1; i++) matches12[i] = -1;
    if (n1 == 0 || F2->n == 0) return HS_OK;
    HIP_TRY(h, hipSetDevice(h->device));
    const int n2 = F2->n;
    int rc = scratch_begin(h, pad256((size_t)n1 * sizeof(hs_keypoint)) + pad256((size_t)n2 * sizeof(hs_keypoint)) + pad256((size_t)n1 * 32) + pad256((size_t)n2 * 32) +
                              pad256((size_t)n2 * 2) + pad256((size_t)n1 * 8) + 4 * pad256((size_t)n2 * 4) + 256);
    if (rc != HS_OK) return rc;
    hipStream_t s = h->stream;
    hs_keypoint* d_k1 = carve<hs_keypoint>(h, n1); hs_keypoint* d_k2 = carve<hs_keypoint>(h, n2);
    uint8_t* d_d1 = carve<uint8_t>(h, (size_t)n1 * 32); uint8_t* d_d2 = carve<uint8_t>(h, (size_t)n2 * 32);
    int8_t* d_cell = carve<int8_t>(h, (size_t)n2 * 2); float* d_prev = carve<float>(h, (size_t)n1 * 2);
    int32_t* d_owner = carve<int32_t>(h, n2); int32_t* d_odist = carve<int32_t>(h, n2); float* d_ang = carve<float>(h, n2); int32_t* d_self = carve<int32_t>(h, n2);
    int32_t* d_nm = carve<int32_t>(h, 1);
    HIP_TRY(h, hipMemcpyAsync(d_k1, kps1, (size_t)n1 * sizeof(hs_keypoint), hipMemcpyHostToDevice, s));
    HIP_TRY(h, hipMemcpyAsync(d_k2, F2->kps, (size_t)n2 * sizeof(hs_keypoint), hipMemcpyHostToDevice, s));
    HIP_TRY(h, hipMemcpyAsync(d_d1, desc1, (size_t)n1 * 32, hipMemcpyHostToDevice, s));
    HIP_TRY(h, hipMemcpyAsync(d_d2, F2->desc, (size_t)n2 * 32, hipMemcpyHostToDevice, s));
    HIP_TRY(h, hipMemcpyAsync(d_prev, prev_matched_xy, (size_t)n1 * 8, hipMemcpyHostToDevice, s));
    hs_launch_frame_grid(*F2, d_k2, d_cell, false, s);
    hs_launch_search_init(*F2, d_k2, d_d2, d_cell, d_k1, d_d1, n1, d_prev, (float)window, th_low, nnratio, d_owner, d_odist, d_ang, d_self, d_nm, s);
    HIP_TRY(h, hipGetLastError());
    std::vector<int32_t> owner(n2);
    HIP_TRY(h, hipMemcpyAsync(owner.data(), d_owner, (size_t)n2 * 4, hipMemcpyDeviceToHost, s));
    HIP_TRY(h, hipMemcpyAsync(n_matches, d_nm, 4, hipMemcpyDeviceToHost, s));
    HIP_TRY(h, hipStreamSynchronize(s));
    for (int i2 = 0; i2 < n2; i2++) {                          // matches_inverse + vbPrevMatched update (:446-458)
        const int i1 = owner[i2];
        if (i1 < 0) continue;
        matches12[i1] = i2;
        prev_matched_xy[2 * i1] = F2->kps[i2].x; prev_matched_xy[2 * i1 + 1] = F2->kps[i2].y;
    }
    return HS_OK;
}

int hs_bow_transform(hs_orb* h, const hs_vocab_tree* T, const uint8_t* desc, int n, int levelsup, int32_t* word_id, float* weight, int32_t* node_id)
{
    if (!h) return HS_ERR_INVALID;
    if (!T || n < 0 || T->n_nodes < 2 || T->levels < 1 || !T->child_begin || !T->child_count || !T->desc || !T->word_id || !T->weight ||
        (n > 0 && (!desc || !word_id || !weight || !node_id)))
        return fail(h, HS_ERR_INVALID, "bad argument");
    if (n == 0) return HS_OK;
    // the walk must terminate inside the tree: children in range, the root has children
    if (T->child_count[0] < 1) return fail(h, HS_ERR_INVALID, "vocabulary root has no children");
    for (int i = 0; i < T->n_nodes; i++) {
        const long cb = T->child_begin[i], cc = T->child_count[i];
        if (cc < 0 || (cc > 0 && (cb <= i || cb + cc > T->n_nodes))) return fail(h, HS_ERR_INVALID, "vocabulary tree is not a forward-linked flat tree");
    }
    HIP_TRY(h, hipSetDevice(h->device));
    const size_t nn = T->n_nodes;
    int rc = scratch_begin(h, pad256(nn * 4) * 3 + pad256(nn * 4) + pad256(nn * 32) + pad256((size_t)n * 32) + 3 * pad256((size_t)n * 4));
    if (rc != HS_OK) return rc;
    hipStream_t s = h->stream;
    int32_t* d_cb = carve<int32_t>(h, nn); int32_t* d_cc = carve<int32_t>(h, nn); int32_t* d_w = carve<int32_t>(h, nn);
    float* d_wt = carve<float>(h, nn); uint8_t* d_nd = carve<uint8_t>(h, nn * 32); uint8_t* d_d = carve<uint8_t>(h, (size_t)n * 32);
    int32_t* o_w = carve<int32_t>(h, n); float* o_wt = carve<float>(h, n); int32_t* o_n = carve<int32_t>(h, n);
    HIP_TRY(h, hipMemcpyAsync(d_cb, T->child_begin, nn * 4, hipMemcpyHostToDevice, s));
    HIP_TRY(h, hipMemcpyAsync(d_cc, T->child_count, nn * 4, hipMemcpyHostToDevice, s));
    HIP_TRY(h, hipMemcpyAsync(d_w, T->word_id, nn * 4, hipMemcpyHostToDevice, s));
    HIP_TRY(h, hipMemcpyAsync(d_wt, T->weight, nn * 4, hipMemcpyHostToDevice, s));
    HIP_TRY(h, hipMemcpyAsync(d_nd, T->desc, nn * 32, hipMemcpyHostToDevice, s));
    HIP_TRY(h, hipMemcpyAsync(d_d, desc, (size_t)n * 32, hipMemcpyHostToDevice, s));
    hs_launch_bow_transform(n, d_d, d_cb, d_cc, d_nd, d_w, d_wt, T->levels, levelsup, o_w, o_wt, o_n, s);
    HIP_TRY(h, hipGetLastError());
    HIP_TRY(h, hipMemcpyAsync(word_id, o_w, (size_t)n * 4, hipMemcpyDeviceToHost, s));
    HIP_TRY(h, hipMemcpyAsync(weight, o_wt, (size_t)n * 4, hipMemcpyDeviceToHost, s));
    HIP_TRY(h, hipMemcpyAsync(node_id, o_n, (size_t)n * 4, hipMemcpyDeviceToHost, s));
    HIP_TRY(h, hipStreamSynchronize(s));
    if (T->orig_id) for (int i = 0; i < n; i++) node_id[i] = T->orig_id[node_id[i]];      // renumbered vocabulary: report DBoW2's NodeId
    return HS_OK;
}

int hs_hamming_knn2_device(hs_orb* h, const uint8_t* d_q, int nq, const uint8_t* d_t, int nt,
                           int32_t* d_best_idx, int32_t* d_best_dist, int32_t* d_second_dist, void* stream)
{
    if (!h) return HS_ERR_INVALID;
    if (nq < 0 || nt < 0 || (nq > 0 && (!d_q || !d_best_idx || !d_best_dist || !d_second_dist)) || (nt > 0 && !d_t)) return fail(h, HS_ERR_INVALID, "bad argument");
    HIP_TRY(h, hipSetDevice(h->device));
    hs_launch_knn2(d_q, nq, d_t, nt, d_best_idx, d_best_dist, d_second_dist, stream ? (hipStream_t)stream : h->stream);
    HIP_TRY(h, hipGetLastError());
    return HS_OK;
}

int hs_hamming_knn2(hs_orb* h, const uint8_t* q, int nq, const uint8_t* t, int nt, int32_t* best_idx, int32_t* best_dist, int32_t* second_dist)
{
    if (!h) return HS_ERR_INVALID;
    if (nq < 0 || nt < 0 || (nq > 0 && (!q || !best_idx || !best_dist || !second_dist)) || (nt > 0 && !t)) return fail(h, HS_ERR_INVALID, "bad argument");
    if (nq == 0) return HS_OK;
    HIP_TRY(h, hipSetDevice(h->device));
    int rc = scratch_begin(h, pad256((size_t)nq * 32) + pad256((size_t)std::max(nt, 1) * 32) + 3 * pad256((size_t)nq * 4));
    if (rc != HS_OK) return rc;
    hipStream_t s = h->stream;
    uint8_t* dq = carve<uint8_t>(h, (size_t)nq * 32); uint8_t* dt = carve<uint8_t>(h, (size_t)std::max(nt, 1) * 32);
    int32_t* bi = carve<int32_t>(h, nq); int32_t* bd = carve<int32_t>(h, nq); int32_t* sd = carve<int32_t>(h, nq);
    HIP_TRY(h, hipMemcpyAsync(dq, q, (size_t)nq * 32, hipMemcpyHostToDevice, s));
    if (nt) HIP_TRY(h, hipMemcpyAsync(dt, t, (size_t)nt * 32, hipMemcpyHostToDevice, s));
    hs_launch_knn2(dq, nq, dt, nt, bi, bd, sd, s);
    HIP_TRY(h, hipGetLastError());
    HIP_TRY(h, hipMemcpyAsync(best_idx, bi, (size_t)nq * 4, hipMemcpyDeviceToHost, s));
    HIP_TRY(h, hipMemcpyAsync(best_dist, bd, (size_t)nq * 4, hipMemcpyDeviceToHost, s));
    HIP_TRY(h, hipMemcpyAsync(second_dist, sd, (size_t)nq * 4, hipMemcpyDeviceToHost, s));
    HIP_TRY(h, hipStreamSynchronize(s));
    return HS_OK;
}

// [ count | pad to 16 | keypoints[cap] | pad to 16 | descriptors[cap][32] ]: the descriptor block starts on a 16-byte boundary whatever the parity of
// cap (24-byte keypoints), because the describe kernel writes a descriptor as two 16-byte vector stores
static size_t record_off_desc(int cap) { return ((size_t)HS_RECORD_HEADER + (size_t)std::max(cap, 0) * sizeof(hs_keypoint) + 15) & ~(size_t)15; }
size_t hs_record_bytes(int cap) { return cap < 0 ? 0 : record_off_desc(cap) + (size_t)cap * HS_DESC_BYTES; }

void hs_record_offsets(int cap, size_t* off_count, size_t* off_kps, size_t* off_desc)
{
    if (off_count) *off_count = 0;
    if (off_kps) *off_kps = HS_RECORD_HEADER;
    if (off_desc) *off_desc = record_off_desc(cap);
}

int hs_records_knn2_device(hs_orb* h, const uint8_t* d_records, size_t record_stride, int world, int rank, int cap,
                           int32_t* d_best_idx, int32_t* d_best_dist, int32_t* d_second_dist, void* stream)
{
    if (!h) return HS_ERR_INVALID;
    if (!d_records || world < 1 || world > 65535 || rank < 0 || rank >= world || cap < 1 || cap > 65535 || record_stride < hs_record_bytes(cap) ||
        (record_stride & 3) || ((uintptr_t)d_records & 15) || !d_best_idx || !d_best_dist || !d_second_dist)
        return fail(h, HS_ERR_INVALID, "bad argument");
    HIP_TRY(h, hipSetDevice(h->device));
    size_t od; hs_record_offsets(cap, nullptr, nullptr, &od);
    hs_launch_knn2_records(d_records, record_stride, world, rank, cap, od, d_best_idx, d_best_dist, d_second_dist, stream ? (hipStream_t)stream : h->stream);
    HIP_TRY(h, hipGetLastError());
    return HS_OK;
}

int hs_debug_stream_copy(hs_orb* h, void* d_dst, const void* d_src, size_t bytes, int width, void* stream)
{
    if (!h) return HS_ERR_INVALID;
    if (!d_dst || !d_src || (width != 4 && width != 16 && width != 64) || bytes % 16) return fail(h, HS_ERR_INVALID, "bad argument");
    HIP_TRY(h, hipSetDevice(h->device));
    hs_launch_stream_copy(d_dst, d_src, bytes, width, stream ? (hipStream_t)stream : h->stream);
    HIP_TRY(h, hipGetLastError());
    return HS_OK;
}

int hs_orb_stage_launches(const hs_orb* h, int stage)
{
    // launches per stage OF THE LAST CALL on the handle (what hs_orb_profile_end's per-stage times are divided by): the pyramid's plan depends on the
    // batch (calls of <= deep_max_batch frames run the small-batch plan: one launch at 1080p instead of three), the stereo stage on the entry point
    if (!h || stage < 0 || stage >= HS_NUM_STAGES) return 0;
    if (stage == 0) {
        if (h->last_pyr_launches > 0) return h->last_pyr_launches;      // counted by the launcher itself: a per-level fallback (caller-frame alignment, no big LDS) is included
        if (h->lv.empty()) return std::max(h->p.nlevels - 1, 0);
        if (h->last_batch > 0 && h->last_batch <= h->deep_max_batch && !h->pyr_deep.empty()) { int32_t o[8]; hs_debug_plan_summary(h, o); return o[1]; }
        return hs_pyramid_launch_count(h->lv.data(), h->p.nlevels);
    }
    return stage == 4 ? h->last_stereo_launches : 1;
}

int hs_orb_profile_begin(hs_orb* h)
{
    if (!h) return HS_ERR_INVALID;
    h->prof = true; h->ev_used = 0; h->prof_stage.clear();
    if (h->lane2) { h->lane2->prof = true; h->lane2->ev_used = 0; h->lane2->prof_stage.clear(); }
    return HS_OK;
}

int hs_orb_profile_pause(hs_orb* h)
{
    if (!h) return HS_ERR_INVALID;
    h->prof = false;
    if (h->lane2) h->lane2->prof = false;
    return HS_OK;
}

int hs_orb_profile_end(hs_orb* h, double* ms, int32_t* launches)
{
    if (!h) return HS_ERR_INVALID;
    if (!ms || !launches) return fail(h, HS_ERR_INVALID, "bad argument");
    HIP_TRY(h, hipSetDevice(h->device));
    for (int i = 0; i < HS_NUM_STAGES; i++) { ms[i] = 0; launches[i] = 0; }
    // the second lane (hs_orb_set_lanes) records its own event sequence: both are summed
    for (hs_orb* q : { h, h->lane2 }) {
        if (!q) continue;
        q->prof = false;
        if (q->ev_used) HIP_TRY(h, hipEventSynchronize(q->ev_pool[q->ev_used - 1]));
        for (size_t i = 0; i + 1 < q->ev_used; i++) {
            int st = q->prof_stage[i];
            if (st < 0 || st >= HS_NUM_STAGES) continue;
            float t = 0.f;
            HIP_TRY(h, hipEventElapsedTime(&t, q->ev_pool[i], q->ev_pool[i + 1]));
            ms[st] += t; launches[st]++;
        }
        q->ev_used = 0; q->prof_stage.clear();
    }
    return HS_OK;
}

int hs_orb_set_lanes(hs_orb* h, int lanes)
{
    if (!h) return HS_ERR_INVALID;
    if (lanes < 1 || lanes > 2) return fail(h, HS_ERR_INVALID, "lanes must be 1 or 2");
    HIP_TRY(h, hipSetDevice(h->device));
    if (lanes == 1) { if (h->lane2) { hs_orb_destroy(h->lane2); h->lane2 = nullptr; } return HS_OK; }
    if (!h->lane2) {
        int rc = hs_orb_create(&h->p, h->device, &h->lane2);
        if (rc != HS_OK) return fail(h, rc, "could not create the second lane");
        if (!h->ev_fork) HIP_TRY(h, hipEventCreateWithFlags(&h->ev_fork, hipEventDisableTiming));
        if (!h->ev_join) HIP_TRY(h, hipEventCreateWithFlags(&h->ev_join, hipEventDisableTiming));
    }
    return HS_OK;
}

int hs_orb_set_split(hs_orb* h, int mode)
{
    if (!h) return HS_ERR_INVALID;
    if (mode < -1 || mode > 1) return fail(h, HS_ERR_INVALID, "split mode must be -1 (auto), 0 or 1");
    h->split_mode = mode;
    if (h->lane2) h->lane2->split_mode = mode;
    return HS_OK;
}

int hs_orb_synchronize(hs_orb* h, void* stream)
{
    if (!h) return HS_ERR_INVALID;
    HIP_TRY(h, hipSetDevice(h->device));
    HIP_TRY(h, hipStreamSynchronize(stream ? (hipStream_t)stream : h->stream));
    return HS_OK;
}

int hs_orb_debug_level(hs_orb* h, int image, int level, uint8_t* out, size_t cap_bytes, int32_t* lw, int32_t* lh)
{
    if (!h) return HS_ERR_INVALID;
    if (!out || image < 0 || image >= h->last_batch || level < 0 || level >= h->p.nlevels) return fail(h, HS_ERR_INVALID, "bad argument");
    HIP_TRY(h, hipSetDevice(h->device));
    HIP_TRY(h, hipDeviceSynchronize());
    const HsLevel& V = h->lv[level];
    if ((size_t)V.w * V.h > cap_bytes) return fail(h, HS_ERR_CAPACITY, "level larger than buffer");
    if (lw) *lw = V.w;
    if (lh) *lh = V.h;
    const uint8_t* src; size_t pitch;
    if (level == 0) { src = hs_img0_ptr(h->last_img0, image); pitch = h->last_img0.row_stride; }
    else { src = V.base + (size_t)image * V.img_stride; pitch = V.pitch; }
    HIP_TRY(h, hipMemcpy2D(out, V.w, src, pitch, V.w, V.h, hipMemcpyDeviceToHost));
    return HS_OK;
}

int hs_orb_set_debug(hs_orb* h, int on)
{
    if (!h) return HS_ERR_INVALID;
    h->keep_points = on != 0;
    if (h->lane2) h->lane2->keep_points = h->keep_points;
    return HS_OK;
}

int hs_orb_debug_candidates(hs_orb* h, int image, int level, int32_t* xys, int cap, int32_t* n)
{
    if (!h) return HS_ERR_INVALID;
    if (h->fast_keys && h->last_batch <= h->fast_keys_max_batch && !h->keep_points)
        return fail(h, HS_ERR_INVALID, "hs_orb_debug_candidates: call hs_orb_set_debug(h, 1) before the extraction (the candidates are only gathered into a dense list in debug mode)");
    if (!xys || !n || image < 0 || image >= h->last_batch || level < 0 || level >= h->p.nlevels) return fail(h, HS_ERR_INVALID, "bad argument");
    HIP_TRY(h, hipSetDevice(h->device));
    HIP_TRY(h, hipDeviceSynchronize());
    const HsLevel& V = h->lv[level];
    int32_t cnt = 0;
    HIP_TRY(h, hipMemcpy(&cnt, h->d_cand_count + image * h->p.nlevels + level, 4, hipMemcpyDeviceToHost));
    cnt = std::min(cnt, V.cand_cap);
    *n = cnt;
    if (cnt > cap) return fail(h, HS_ERR_CAPACITY, "more candidates than buffer");
    std::vector<uint32_t> xy(cnt), sk(cnt);
    if (cnt) {
        HIP_TRY(h, hipMemcpy(xy.data(), h->d_pts_xy + (size_t)image * h->cand_img_stride + V.cand_off, (size_t)cnt * 4, hipMemcpyDeviceToHost));
        HIP_TRY(h, hipMemcpy(sk.data(), h->d_pts_sk + (size_t)image * h->cand_img_stride + V.cand_off, (size_t)cnt * 4, hipMemcpyDeviceToHost));
    }
    for (int i = 0; i < cnt; i++) { xys[3 * i] = xy[i] & 0xFFFF; xys[3 * i + 1] = xy[i] >> 16; xys[3 * i + 2] = sk[i] >> 24; }
    return HS_OK;
}

int hs_orb_debug_selected(hs_orb* h, int image, int level, int32_t* xys, int cap, int32_t* n)
{
    if (!h) return HS_ERR_INVALID;
    if (!xys || !n || image < 0 || image >= h->last_batch || level < 0 || level >= h->p.nlevels) return fail(h, HS_ERR_INVALID, "bad argument");
    HIP_TRY(h, hipSetDevice(h->device));
    HIP_TRY(h, hipDeviceSynchronize());
    const HsLevel& V = h->lv[level];
    int32_t cnt = 0;
    HIP_TRY(h, hipMemcpy(&cnt, h->d_sel_count + image * h->p.nlevels + level, 4, hipMemcpyDeviceToHost));
    *n = cnt;
    if (cnt > cap) return fail(h, HS_ERR_CAPACITY, "more keypoints than buffer");
    if (cnt) HIP_TRY(h, hipMemcpy(xys, h->d_sel + ((size_t)image * h->sel_img_stride + V.sel_off) * 3, (size_t)cnt * 12, hipMemcpyDeviceToHost));
    return HS_OK;
}

} // extern "C"
