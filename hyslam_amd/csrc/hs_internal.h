// hs_internal.h — shared declarations of the HIP implementation behind include/hyslam_amd.h.
// gfx950 (MI355X) only: wave64, 160 KiB LDS/CU.  Compiled with -ffp-contract=off: several results
// (resize tables, fastAtan2, rBRIEF rotation) are defined by separately rounded fp32 operations.
#pragma once
#include <vector>
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include "../../include/hyslam_amd.h"

#define HS_MAX_LEVELS 16
#define HS_EDGE 19                 // EDGE_THRESHOLD, ORBExtractor.cpp:74
#define HS_BORDER 16               // minBorderX = EDGE_THRESHOLD-3, ORBExtractor.cpp:413
#define HS_MAX_CELL_H 125           // tallest FAST cell (the FAST kernel's LDS tile holds hcell + 6 rows; list entries keep the row in 7 bits)
#define HS_QT_MAX_NODES 2048       // quadtree list capacity in LDS of the general instance (>= largest per-level quota + 8)
#define HS_QT_LARGE_NODES 3328     // ... of the large-list instance (rectangles in global scratch, no points in LDS: 161.9 KB of the CU's 160 KiB); quotas up to 3320 per level
#define HS_QT_THREADS 1024

// Per-level geometry and buffers; lives in device memory, read through scalar loads.
struct HsLevel {
    int32_t w, h;                  // level size (ORBExtractor.cpp:568-569)
    int32_t pitch;                 // row pitch of the level buffer (levels >= 1)
    int32_t _r0;
    uint64_t img_stride;           // bytes between consecutive images of this level's buffer
    uint8_t* base;                 // level buffer (levels >= 1; level 0 is the caller's image)
    // FAST cell grid (ORBExtractor.cpp:413-428)
    int32_t ncols, nrows, wcell, hcell;
    int32_t cell_begin;            // first block index of this level in the all-levels cell launch
    // FAST work items (kernels_fast.hip): one item = `grp_cells` horizontally adjacent cells of one cell row
    int32_t grp_cells, ngroups;    // cells per item (the last item of a row may hold fewer), items per cell row
    int32_t item_begin;            // first item of this level within one image's item list
    int32_t inv_wcell, inv_wcell1; // ceil(65536 / wcell), ceil(65536 / (wcell + 1)): x / wcell == (x * inv) >> 16 for x < 256
    // DistributeOctTree inputs (ORBExtractor.cpp:179-203,475-476)
    int32_t qt_w, qt_h;            // maxBorderX-minBorderX, maxBorderY-minBorderY
    int32_t n_ini;                 // round(qt_w / qt_h)
    float   hx;                    // qt_w / n_ini
    int32_t quota;                 // mnFeaturesPerLevel[level]
    // geometric-key tables of the count-domain quadtree (kernels_quadtree.hip), built on the host from the level geometry: cell index of every
    // pixel column within its root / of every pixel row at the depth of the histogram pyramid, first column of roots 1..7; nullptr = none
    const uint8_t* qt_xtab;        // [qt_w + 1], 16-byte aligned, padded to a multiple of 16
    const uint8_t* qt_ytab;        // [qt_h + 1]
    int32_t qt_rbound[8];
    // round 4: the FAST kernel computes every candidate's geometric key itself and leaves, per (image, level), the HISTOGRAM of the keys (u16
    // counters packed in u32) and the BEST candidate of every deepest cell (u64 keys) in global memory: the quadtree kernel starts from them
    // instead of gathering the candidates.  Offsets of this level inside one image's arrays (entries); 0xFFFFFFFF = the level has no tables.
    uint32_t qt_hist_off;          // in u32 (= 2 cells)
    uint32_t qt_best_off;          // in u64 (= 1 cell)
    // candidate / selection storage (entries, per image)
    int32_t cand_cap;
    int32_t sel_cap;
    uint64_t cand_off;             // offset of this level inside one image's candidate arrays
    int32_t sel_off;               // offset inside one image's selection arrays
    // resize tables (cv::resize INTER_LINEAR, fixed point), device pointers; level >= 1
    int32_t xmax;
    const int16_t* xofs;           // [w]
    const int16_t* ialpha;         // [w][2]
    const int16_t* yofs;           // [h]
    const int16_t* ibeta;          // [h][2]
    float scale;                   // mvScaleFactor[level]
    float kp_size;                 // (int)(31*scale)
    // two-levels-per-launch pyramid kernel (kernels_pyramid.hip): this level and the next one are produced by one workgroup per tile of the NEXT
    // level; 0 = this pair is not fused.  Tile width of the next level, LDS rows for this level's region / the source rectangle, source pitch.
    int32_t fuse_tbx, fuse_ar, fuse_sr, fuse_pitch;
    int32_t chain_n;               // levels produced by a k_resize_chain launch that starts at this level (0: none)
    int32_t _r1;
};

// One work item of the FAST kernel (kernels_fast.hip): `ncell` horizontally adjacent cells of one cell row.  Everything that does not
// depend on the image index is precomputed on the host, so a wave fetches an item's geometry with a single 64-byte scalar load.
struct HsFastItem {
    const uint8_t* base;           // level buffer; nullptr = level 0 (the caller's frame, HsImg0)
    uint64_t img_stride;           // bytes between images of the level buffer
    int32_t pitch;                 // row pitch of the level buffer (level 0: HsImg0::row_stride)
    int32_t gcell0;                // HsLevel::cell_begin + c0: first entry of the item in cell_count
    int32_t c0;                    // cell index (row-major within the level) of the item's first cell
    uint32_t slot0;                // HsLevel::cand_off + c0 * ccap: the first cell's slots in the candidate arrays
    int32_t ccap;                  // slots per cell
    int32_t inv_w, inv_w1;         // ceil(65536 / wcell), ceil(65536 / (wcell + 1))
    uint16_t iniY, a0;             // first tile row; first tile column rounded down to a dword
    uint16_t th, iw;               // tile rows (0: the item yields nothing), interior width
    uint8_t off, ndw, ncell, level;// tile column of x is off + (x - iniX); dwords per tile row
    uint16_t xoff, yoff;           // j0 * wCell, i * hCell (ORBExtractor.cpp:463-464)
    uint32_t _pad;
};
static_assert(sizeof(HsFastItem) == 64, "HsFastItem is fetched as one 64-byte record");

// Per-level record of the FAST kernel's key computation (one 32-byte scalar load per work item): u16 tables with the x part of a candidate's
// geometric key per pixel column — root << 2 DH | the column's cell index at depth DH with its bits spread to the even positions — and the y part
// per pixel row (bits spread to the odd positions), so that key = xkey[x] | ykey[y] (k_quadtree's geo_key, kernels_quadtree.hip); both padded
// by 512 entries so that a work item's slice can be fetched without clamping.
struct HsFastQt { const uint16_t* xkey; const uint16_t* ykey; uint32_t hist_off, best_off; int32_t enabled, _r; };
static_assert(sizeof(HsFastQt) == 32, "scalar-load record");

// candidate slots per FAST cell: 3x3 NMS leaves at most one survivor per 2x2 block; rounded to 4 so that a cell's records start on
// a 16-byte boundary (cand_off is a multiple of 4 entries)
__host__ __device__ inline int hs_cell_cap(int wcell, int hcell) { return ((((wcell + 1) >> 1) * ((hcell + 1) >> 1)) + 3) & ~3; }

struct HsImg0 {                    // level 0 = the caller's frames; images [0,split) from base, the rest from base2
    const uint8_t* base;           // (left / right frames of a stereo batch go through one launch sequence)
    const uint8_t* base2;
    int32_t split;
    uint64_t row_stride;
    uint64_t img_stride;
};
__host__ __device__ inline const uint8_t* hs_img0_ptr(const HsImg0& I, int img)
{
    return img < I.split ? I.base + (size_t)img * I.img_stride : I.base2 + (size_t)(img - I.split) * I.img_stride;
}

// Pointers that reach a kernel through a struct in memory (HsLevel::base, the tables) have no provable address space and compile to
// FLAT loads, which count on lgkmcnt as well as vmcnt: every LDS wait then also waits for them.  These helpers assert "global".
#define HS_GLOBAL __attribute__((address_space(1)))
typedef uint32_t hs_u32x4 __attribute__((ext_vector_type(4)));
template <typename T> __device__ __forceinline__ T hs_gload(const void* p) { return *(const HS_GLOBAL T*)(uintptr_t)p; }
// read-only tables at a wave-uniform address: the constant address space lets the compiler use scalar loads (SGPR result, no VALU/VMEM slot)
#define HS_CONSTANT __attribute__((address_space(4)))
template <typename T> __device__ __forceinline__ T hs_cload(const void* p) { return *(const HS_CONSTANT T*)(uintptr_t)p; }
// element `idx` of a read-only int16 table (4-byte aligned base) at a wave-uniform index: there are no sub-dword scalar loads, so the
// containing dword is fetched and the half picked with SALU (a plain int16 load would be a VECTOR load and put the value in a VGPR)
__device__ __forceinline__ int hs_cload_i16(const int16_t* base, int idx)
{
    const uint32_t w = hs_cload<uint32_t>(reinterpret_cast<const uint8_t*>(base) + 4 * (size_t)(idx >> 1));
    return (int)(int16_t)((idx & 1) ? (w >> 16) : (w & 0xFFFFu));
}
// a pointer the compiler must keep in SGPRs (so that `uniform base + 32-bit lane offset` becomes the saddr form of global_load)
__device__ __forceinline__ const uint8_t* hs_uniform_ptr(const uint8_t* p)
{
    const uint64_t v = (uint64_t)(uintptr_t)p;
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
    return (const uint8_t*)(uintptr_t)(((uint64_t)hi << 32) | lo);
}
template <typename T> __device__ __forceinline__ T hs_gload_off(const uint8_t* uniform_base, uint32_t lane_off)
{
    return *(const HS_GLOBAL T*)((const HS_GLOBAL uint8_t*)(uintptr_t)uniform_base + lane_off);
}
template <typename T> __device__ __forceinline__ void hs_gstore(void* p, T v) { *(HS_GLOBAL T*)(uintptr_t)p = v; }

struct HsOut {                     // extractor outputs, split the same way
    hs_keypoint* kps; uint8_t* desc; int32_t* n;
    hs_keypoint* kps2; uint8_t* desc2; int32_t* n2;
    int32_t split;
    int32_t cap;
};

// Two-levels-per-launch pyramid kernel (kernels_pyramid.hip): everything a workgroup used to derive from the level and resize tables with a
// chain of dependent scalar loads is precomputed on the host — the tile geometry factorises into one record per tile column and one per tile
// row, the vertical pass gets one 8-byte record per destination row (clamped source rows + weights), and the level descriptions travel as a
// kernel argument instead of through the HsLevel array.
struct HsXTab { int16_t sx, a0, a1, pad; };                       // x table of cv::resize: first source column and the two 11-bit weights
struct HsPyrXTile { int32_t ax0, own_x1, col0, nvec, _r[4]; };    // first column / end of the owned columns of the level-A region; first source column (16-aligned), 16-byte vectors per source row
struct HsPyrYTile { int32_t ay0, own_y1, ay_last, sy_first, n_sr, _r[3]; };   // level-A rows [ay0, ay_last], owned up to own_y1; source rows from sy_first, n_sr of them
struct HsPyrRow { uint16_t off0, off1, b0, b1; };                 // a destination row OF ONE TILE: byte offsets of its two source rows in the tile's sums buffer ((row - first row held) * 512) and the vertical weights
static_assert(sizeof(HsPyrXTile) == 32 && sizeof(HsPyrYTile) == 32 && sizeof(HsPyrRow) == 8, "scalar-load records");
struct HsPyrFuse {
    const uint8_t* sbase;          // source level S = A - 1 (nullptr: level 0 = the caller's frames, HsImg0)
    uint64_t s_img_stride; int32_t spitch; int32_t _r0;
    uint8_t* abase; uint64_t a_img_stride; int32_t apitch, aw, ah, slotA;   // slotA: records per y tile in rowA
    uint8_t* bbase; uint64_t b_img_stride; int32_t bpitch, bw, bh, _r2;
    const HsXTab* xtA; const HsXTab* xtB;
    const HsPyrRow* rowA; const HsPyrRow* rowB;                    // per y tile: [grid.y][slotA] rows ay0.. of the tile's level-A region, [grid.y][FZ_ROWS + 8] rows of its level-B tile (padded with copies of the last row)
    const HsPyrXTile* xt; const HsPyrYTile* yt;                    // [grid.x], [grid.y]
    int32_t tbx, sr, lds_pitch, valid;                             // tile width of level B, LDS rows / pitch of the source rectangle; valid = this pair is fused
};

// A CHAIN of 2 or 3 levels per launch (k_resize_chain: the two-level scheme applied once more, so that the last three levels of an 8-level
// pyramid are one launch instead of two).  Stage i produces level first+i from the LDS copy of the level before it (stage 0: from global
// memory); a workgroup owns one tile of the LAST level and, of every other level, the part induced by the tiles' first source columns / rows.
struct HsPyrStageX { int32_t x0, own_x1, ncols, src_x0, nvec, _r[3]; };      // region [x0, x0 + ncols) of the stage's level, owned up to own_x1; first column the source buffer holds; stage 0: 16-byte vectors per source row
struct HsPyrStageY { int32_t y0, own_y1, y_last, src_y0, n_src, _r[3]; };    // region rows [y0, y_last], owned up to own_y1; first row / number of rows the source buffer holds
static_assert(sizeof(HsPyrStageX) == 32 && sizeof(HsPyrStageY) == 32, "scalar-load records");
struct HsPyrStage {
    uint8_t* base; uint64_t img_stride; int32_t pitch, w, h, slot;           // the level the stage produces; slot = row records per y tile
    const HsXTab* xt; const HsPyrRow* rows;                                    // its x table and row records [grid.y][slot]: rows y0.. of the tile's region (padded with copies of the last row)
    const HsPyrStageX* tx; const HsPyrStageY* ty;                              // [grid.x], [grid.y]
};
#define HS_PYR_DEEP_LDS (120 * 1024)  // LDS a deep chain may plan with (hs_pyramid_plan_chain's lds_max for the small-batch plan)
#define HS_PYR_CHAIN_MAX 7            // levels one k_resize_chain launch can produce (small batches: the whole pyramid of 8 levels in ONE launch)
struct HsPyrChain {
    const uint8_t* sbase; uint64_t s_img_stride; int32_t spitch, nstage;       // the source level (sbase == nullptr: the caller's frames)
    HsPyrStage st[HS_PYR_CHAIN_MAX];
    int32_t tbx, lds_pitch, x_bytes, h_rows;                                   // tile width of the last level; LDS: pitch of the source rectangle, bytes of the level buffer, rows of the sums buffer
    int32_t grid_x, grid_y, valid, _r;
};

#define HS_STRIP_ENTRY_BYTES 16
#define HS_STRIP_SHIFT 5           // right keypoints are binned into strips of 32 rows (kernels_stereo.hip)
// A strip entry carries everything the candidate test needs, so that the matcher's chain of dependent loads is entry -> descriptor instead of
// index -> keypoint record -> descriptor: the right keypoint's u, its octave, its index and its row band CUT TO THE STRIP (two 5-bit row numbers:
// a left keypoint that scans strip s has its row in [32 s, 32 s + 31], so the cut band decides exactly what the whole band decides).
struct __attribute__((aligned(16))) HsStripEntry { float uR; int32_t octave; uint32_t idx_band; uint32_t _pad; };   // idx_band = iR | lo << 16 | hi << 24
static_assert(sizeof(HsStripEntry) == HS_STRIP_ENTRY_BYTES, "hs_api.hip sizes the strip lists with HS_STRIP_ENTRY_BYTES");
#define HS_STRIPS_MAX 2048         // 65536 rows / 32
// The stereo front end (extract L + R, then match; hs_stereo_frontend_batch_device): the describe launch carries one extra workgroup per RIGHT
// image that bins the image's selected keypoints into the strips (position, level and list index are known since the quadtree kernel; the
// describe workgroups next to it fill in angle and descriptor) — no k_stereo_strips launch.  enabled = 0: a plain describe launch.
struct HsStripFuse { int32_t enabled, n_rows, n_strips, _r; float size_ref; int32_t* strip_count; HsStripEntry* strip_list; };
inline int hs_stereo_strips(int n_rows) { return (n_rows > 0 ? ((n_rows - 1) >> 5) : 0) + 1; }

// kernels_*.hip launchers (all asynchronous on `s`)
int  hs_launch_pyramid(const HsLevel* d_lv, const HsLevel* h_lv, const HsPyrFuse* fuse /*[nlevels], host*/, const HsPyrChain* chain /*[nlevels], host*/, int nlevels, HsImg0 img0, int batch, hipStream_t s,
                       const HsPyrChain* deep = nullptr /*[nlevels], host: the small-batch plan (long chains); used where deep[l].valid*/);      // returns the kernel launches it enqueued
// plans a chain over levels [first, first + n) (2 <= n <= HS_PYR_CHAIN_MAX): tile tables appended to `blob` (offsets until relocated); C.valid = 0 when the
// geometry does not fit `lds_max` bytes of LDS
void hs_pyramid_plan_chain(const HsLevel* h_lv, int first, int n, const int16_t* const* xtab, const int16_t* const* yofs, const int16_t* const* ibeta,
                           std::vector<uint64_t>& blob, HsPyrChain& C, size_t lds_max = 60 * 1024, int tile_rows = 0 /*rows of the last level per tile; 0 = the two-level kernel's 16*/);
// host side of HsPyrFuse for every fused pair: records appended to `blob` (device pointers are blob offsets until hs_api.hip relocates them)
void hs_pyramid_build_tables(const HsLevel* h_lv, int nlevels, const int16_t* const* xtab, const int16_t* const* yofs, const int16_t* const* ibeta,
                             std::vector<uint64_t>& blob, std::vector<HsPyrFuse>& fuse);
// decides, from the host copies of the resize tables, whether levels (l, l+1) can be produced by the fused kernel and with which tile geometry
void hs_pyramid_plan_fusion(HsLevel* h_lv, int nlevels, const int16_t* const* xtab /*[level] {sx,a0,a1,-} per column*/, const int16_t* const* yofs /*[level]*/, int tbx_max = 0 /*> 0: cap on the level-B tile width*/);
int hs_pyramid_launch_count(const HsLevel* h_lv, int nlevels);          // kernel launches of one hs_launch_pyramid call
int hs_fast_group_cells(int wcell, int ncols, int lc);   // cells per FAST work item for a level (0 when the level has no cells); lc = 6 / 5: wide / narrow tiles
int hs_fast_max_cell_w(int lc);                          // widest FAST cell the kernel's tile holds at any offset (247 px wide tiles, 119 px narrow ones)
void hs_fast_build_items(const HsLevel* h_lv, int nlevels, HsFastItem* out /*[sum ngroups*nrows]*/);
bool hs_fast_item_fits(const HsFastItem& it, int lc);    // the item against the tile k_fast_rows<lc, ...> stages it into
#define HS_FAST_NQ_MAX 32          // work queues of the FAST kernel: 8, 16 or 32 (kernels_fast.hip: FastSched)
#define HS_FAST_QUEUE_DWORDS (32 * HS_FAST_NQ_MAX)   // head of the FAST overflow buffer: FOUR rotating sets of up to 32 work-queue counters on 128-byte lines of their own
struct HsFastKnobs { int pcap, small_lists, wg_per_cu, force_scan_b, image_major, nq, cols, narrow_max, no_fold, list_min; };   // HS_FAST_* test / tuning knobs, read once per handle
HsFastKnobs hs_fast_read_knobs();
bool hs_launch_fast(const HsLevel* d_lv, const HsFastItem* d_items, int nlevels, HsImg0 img0, int batch, int total_cells, int items_per_img, int fast_th,
                    uint2* cand /*{y<<16|x, score<<24|cell} per slot*/, int32_t* cell_count, uint64_t cand_img_stride,
                    int max_wcell, int max_hcell, uint32_t* overflow /*hs_fast_overflow_bytes(), zero-initialised*/, uint32_t epoch /*launch counter of the handle*/,
                    const HsFastKnobs& knobs, int item_first, int item_count /*the launch covers items [first, first + count) of every image*/,
                    int spill_slot /*0 / 1: which half of the spill areas (two launches may be in flight)*/, int lc /*6 / 5: the tile width `d_items` was built for*/,
                    const HsFastQt* d_qt /*[nlevels]; nullptr: no keys*/, uint32_t* qhist, unsigned long long* qbest, uint32_t qhist_img_stride, uint32_t qbest_img_stride, hipStream_t s);
size_t hs_fast_overflow_bytes(int max_hcell, int total_work_max, const HsFastKnobs& knobs);   // per-wave spill areas of the FAST kernel for launches over <= total_work_max items
// host side of the geometric-key tables: appends level `V`'s tables to `blob` (16-byte granules) and returns their offsets; false = the level does
// not use them (more than 8 roots or a level wider than the tables)
bool hs_quadtree_build_tables(HsLevel& V, std::vector<uint8_t>& blob, size_t& xoff, size_t& yoff);
void hs_launch_quadtree(const HsLevel* d_lv, int nlevels, int batch, int total_cells,
                        const uint2* cand, const int32_t* cell_count, uint64_t cand_img_stride,
                        uint32_t* pts_xy, uint32_t* pts_sk, uint16_t* pt_node, int32_t* cand_count,
                        uint32_t* sel_xys, int32_t* sel_count, int sel_img_stride, uint16_t* sel_perm /*spatial order per (image, level)*/, int force_point_domain, int level_first, int level_count,
                        uint32_t* qhist /*nullptr: the FAST launch left no keys — gather*/, unsigned long long* qbest, uint32_t qhist_img_stride, uint32_t qbest_img_stride,
                        int keep_points /*debug: gather the candidates into pts_* even when the keys make it unnecessary*/,
                        int list_mode /*0: the general instance; 1: every level's quota + 8 <= hs_quadtree_small_nodes(), launches of > 256 workgroups may use the two-per-CU instance;
                                        2: a quota + 8 > HS_QT_MAX_NODES: the large-list instance (needs rect_scratch)*/,
                        uint8_t* rect_scratch /*list_mode 2: batch * nlevels * hs_quadtree_large_scratch_bytes(), indexed by (image, level)*/, hipStream_t s);
// kernels_preprocess.hip: ImageProcessing::PreProcessImg (camera scale + grey) between the upload and the pyramid
void hs_preprocess_out_size(int w, int h, float fscale, int* ow, int* oh);
int  hs_preprocess_mode(int w, int h, int ow, int oh, float fscale);      // 0 copy, 1 the 2x2 area path (scale exactly 0.5), 2 bilinear
void hs_launch_preprocess(const uint8_t* d_src, int sw, int sh, size_t src_row_stride, size_t src_img_stride, int channels, int rgb, float fscale,
                          uint8_t* d_dst, int dw, int dh, size_t dst_row_stride, size_t dst_img_stride, int dst_may_pad /*rows may be written up to the next multiple of 4 columns*/,
                          int batch, hipStream_t s);
int hs_quadtree_small_nodes();
size_t hs_quadtree_large_scratch_bytes();
void hs_launch_describe(const HsLevel* d_lv, int nlevels, HsImg0 img0, int batch,
                        const uint32_t* sel_xys, const int32_t* sel_count, const uint16_t* sel_perm, int sel_img_stride, int max_sel,
                        const uint16_t* taps7, HsOut out, hipStream_t s, bool fast_taps, HsStripFuse strips);
void hs_launch_stereo(const hs_keypoint* kpsL, const uint8_t* descL, const int32_t* nL,
                      const hs_keypoint* kpsR, const uint8_t* descR, const int32_t* nR,
                      int pairs, int cap, hs_stereo_params sp, float* uRight, float* depth,
                      int32_t* best_dist /*[pairs][cap] scratch*/,
                      int32_t* strip_count /*[pairs][strips]*/, void* strip_list /*[pairs][strips][cap] entries of HS_STRIP_ENTRY_BYTES*/, hipStream_t s);
void hs_launch_stereo_median(const int32_t* nL, int pairs, int cap, float* uRight, float* depth, const int32_t* best_dist,
                             int32_t* strip_count /*zero on entry of hs_launch_stereo; zeroed again here*/, int n_rows, hipStream_t s);
// the matcher alone, on strips that the describe launch of the stereo front end has already binned (HsStripFuse)
void hs_launch_stereo_match_only(const hs_keypoint* kpsL, const uint8_t* descL, const int32_t* nL,
                                 const hs_keypoint* kpsR, const uint8_t* descR, const int32_t* nR,
                                 int pairs, int cap, hs_stereo_params sp, float* uRight, float* depth, int32_t* best_dist,
                                 const int32_t* strip_count, const void* strip_list, hipStream_t s);

// kernels_match.hip
size_t hs_frame_grid_bytes(int n);               // cells + cell lists of n keypoints
void hs_launch_frame_grid(const hs_frame_view& F, const hs_keypoint* d_kps, int8_t* d_cell /*hs_frame_grid_bytes(n)*/, bool with_lists, hipStream_t s);
void hs_launch_search_projection(const hs_frame_view& F, const hs_keypoint* d_kps, const uint8_t* d_desc, const float* d_uR,
                                 const int32_t* d_obs, const int8_t* d_cell, const hs_landmark* d_lms, int L, const hs_proj_params& pp,
                                 int32_t* d_match_idx, float* d_match_dist, int32_t* d_winner, float* d_prev_angle_scratch,
                                 int32_t* d_n_matches, hipStream_t s);
void hs_launch_bow(const int32_t* d_pair_a, const int32_t* d_pair_b, int n_pairs,
                   const int32_t* d_ptr1, const int32_t* d_idx1, const int32_t* d_ptr2, const int32_t* d_idx2,
                   const uint8_t* d_desc1, const uint8_t* d_desc2, const uint8_t* d_keep1, const uint8_t* d_keep2,
                   const float* F12, float size_ref, float sigma_ref, float thr, float ratio,
                   int32_t* d_match12, int n1, const hs_keypoint* d_kps1, const hs_keypoint* d_kps2, float* d_angle2_scratch,
                   int check_rotation, int32_t* d_self_scratch, int32_t* d_n_matches, hipStream_t s);
void hs_launch_knn2(const uint8_t* d_q, int nq, const uint8_t* d_t, int nt, int32_t* d_bi, int32_t* d_bd, int32_t* d_sd, hipStream_t s);
void hs_launch_sim3_projection(const hs_frame_view& F, const hs_keypoint* d_kps, const uint8_t* d_desc, const int8_t* d_cell,
                               const float* R9, const float* t3, const float* Ow3, const hs_landmark* d_lms, int L, float th, float th_low,
                               float* d_geo, uint8_t* d_kp_matched, int32_t* d_match_idx, int32_t* d_n_matches, hipStream_t s);
void hs_launch_sim3_search(const hs_frame_view& F1, const hs_keypoint* d_kps1, const uint8_t* d_desc1, const int8_t* d_cell1,
                           const hs_frame_view& F2, const hs_keypoint* d_kps2, const uint8_t* d_desc2, const int8_t* d_cell2,
                           const hs_landmark* d_lms1, const hs_landmark* d_lms2, const float* sR21, const float* t21, const float* sR12, const float* t12,
                           float th, float th_high, int32_t* d_m1, int32_t* d_m2, int32_t* d_match12, int32_t* d_n_found, hipStream_t s);
void hs_launch_knn2_records(const uint8_t* d_recs, size_t stride, int world, int rank, int cap, size_t off_desc,
                            int32_t* d_bi, int32_t* d_bd, int32_t* d_sd, hipStream_t s);
void hs_launch_stream_copy(void* d_dst, const void* d_src, size_t bytes, int width, hipStream_t s);
void hs_launch_bow_legacy(const int32_t* d_pair_a, const int32_t* d_pair_b, int n_pairs,
                          const int32_t* d_ptr1, const int32_t* d_idx1, const int32_t* d_ptr2, const int32_t* d_idx2,
                          const uint8_t* d_desc1, const uint8_t* d_desc2, const uint8_t* d_keep1, const uint8_t* d_keep2, float thr, float ratio,
                          int32_t* d_match12, int n1, int n2, const hs_keypoint* d_kps1, const hs_keypoint* d_kps2, float* d_angle1_scratch,
                          int check_orientation, int32_t* d_self_scratch, uint32_t* d_taken2, int32_t* d_n_matches, hipStream_t s);
void hs_launch_bow_transform(int n, const uint8_t* d_desc, const int32_t* d_cb, const int32_t* d_cc, const uint8_t* d_ndesc, const int32_t* d_word,
                             const float* d_weight, int levels, int levelsup, int32_t* d_out_word, float* d_out_weight, int32_t* d_out_node, hipStream_t s);
void hs_launch_search_init(const hs_frame_view& F2, const hs_keypoint* d_kps2, const uint8_t* d_desc2, const int8_t* d_cell2,
                           const hs_keypoint* d_kps1, const uint8_t* d_desc1, int n1, const float* d_prev_xy, float window, float th_low, float nnratio,
                           int32_t* d_owner, int32_t* d_odist, float* d_angle_scratch, int32_t* d_self_scratch, int32_t* d_n_matches, hipStream_t s);
