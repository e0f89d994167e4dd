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
// MI355X mapping (k_fast_rows).  The unit of work is a ROW GROUP: up to 8 horizontally adjacent cells of one cell row (<= 250 px of
// interior), one single-wave workgroup per item, persistent launch, 12 workgroups per CU (wide tiles of <= 40 rows: 12 800 bytes of LDS each = the
// item's tile + its list; narrow tiles 16 per CU).  With ~1 % corners a single 31x31 cell leaves a wave's lanes mostly idle after the first pass and
// pays the per-cell bookkeeping 6342 times per frame; measured on 32 frames of 1080p the first design (one wave per cell: 0.48 ms) became 0.24 ms in
// round 2 and 0.143 ms in round 5.
//   stage   the item's tile (interior + 3 px apron, <= 256 x 70 px) is fetched with 16-byte loads into registers while the PREVIOUS
//           item is processed, then written to LDS; the only HBM traffic of the kernel is this one read of each level
//   scan A  quick reject on the four compass points, 4 pixel columns x 8 rows per lane and block: 14 tile rows in registers, reduced to 6
//           bits so that four pixels fit one plain 32-bit ALU operation; the horizontal ring pixels come from the neighbouring lanes (DPP
//           wave shifts), the vertical differences are shared between the pixels three rows apart; the masks of a block are byte-
//           transposed across lanes (a cluster of hits would otherwise keep one lane busy for 18 rounds) and their bits go to an LDS
//           list as codes (wave prefix sum of popcounts, one short bit loop per lane)
//   corners maybe pixels (decoded on dense lanes): the exact compass test picks the ONE polarity a pixel can still be a corner of, and
//           the corner-score network of that polarity (three-input min / max over the 16 arcs of 9) is the segment test: score >= t
//           decides, the score is kept.  Corners are compacted in place (score tile coordinates + score); the few pixels that pass both
//           compass tests without being a bright corner are re-queued at the list end for the dark test
//   tile    once every corner is scored the pixel tile is dead: its LDS is zeroed and becomes the dense score tile, in which every
//           cell owns its columns plus one zero separator column, so the 3x3 NMS needs no cell-boundary logic
//   NMS     over the corner list; survivors take consecutive slots of their ITEM (a wave-uniform running count).
// The list has a fixed capacity; when a block would overflow it the scan runs the corner passes on what it has and spills the scored
// corners to a per-wave area in global memory; they come back into the score tile before the NMS, which then walks the non-zero
// bytes of the tile instead of the list.  Saturated images stay correct and merely lose batching.  Work distribution, geometry table
// and LDS budget: see the kernel and its launcher.
// Output: each ITEM owns a fixed slot range (the slots of its cells, contiguous; no global atomics), filled from its first slot upwards in NO
// particular order — every order-dependent decision downstream uses the (cell, y, x) key of the reference's candidate order:
//   cand[slot] = { y<<16 | x,  score<<24 | cell }   (coordinates relative to (16,16), as in vToDistributeKeys; cell = row-major cell index
//                                                    of the level); one 8-byte store per survivor, ~20 records = two cache lines per item
//   cell_count[image][first global cell of the item] = number of slots used (the entries of the item's other cells are not written).
// Bound (rounds 5-6): half by SIMD throughput and half by what ONE wave can issue, at 3 waves per SIMD — 12 single-wave workgroups per CU
// is what BOTH the LDS (12 800 bytes) and the registers (142 VGPRs) allow; t = 0.240 + 3.37 / n ms per 128 frames for n workgroups per CU, so occupancy is
// the lever and every added instruction costs.  VALU 0.87 per busy CU cycle of a mix-weighted ceiling ~1.1.  LDS: 49 % of the busy cycles active, 18.5 % in
// bank conflicts, 78 % of THOSE in the corner pass (tools/fast_lds_conflicts.sh: the 34 ds_read_u8 ring gathers of two candidates per lane at random tile
// positions — 2.0 conflict cycles per LDS instruction, the birthday rate of 32 lanes on 32 banks; that phase keeps the LDS busy 74 % of its time).  Wider
// reads do not help: gfx950 executes a ds_read_b32 / b64 at a byte-unaligned address ~6 x slower than an aligned one (tools/micro/lds_gather.hip: 7
// unaligned ds_read_b64 per ring 2 250 ns against 17 ds_read_u8 600 ns per wave-iteration at 12 waves per CU), and aligned chunks need two reads + a
// per-lane byte shift per ring row.  HBM bytes = P per frame, read once (measured 1.05-1.06 x).
#include "hs_internal.h"
#include <algorithm>
#include <cstdlib>
#include <cstring>

// The workgroup is one wave: its LDS operations execute in program order, so a hand-off through LDS only needs the LDS queue
// drained (no s_barrier, and no vmcnt wait that would expose the latency of the next item's prefetch).
#define WAVE_LDS_FENCE() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")

// append: lanes with `flag` get consecutive slots after `base` (wave-uniform); returns the lane's slot, advances base
__device__ __forceinline__ int wave_append(bool flag, int& base)
{
    const unsigned long long m = __ballot(flag);
    const int slot = base + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
    base += (int)__popcll(m);
    return slot;
}

// Corner score of one polarity: cv::FAST's cornerScore<16> is max(t, max_arc min(v-ring), max_arc min(ring-v)) - 1; for a corner of one
// polarity the other polarity's term cannot exceed t (no 9-arc passes it) while its own term does, so score = max_arc min(d) - 1 with
// d[k] = v - ring[k] (dark) or ring[k] - v (bright), and the pixel IS a corner of that polarity exactly when max_arc min(d) > t.
// Three-input min / max: lo3[k] = min(d[k..k+2]), arc9[k] = min(lo3[k], lo3[k+3], lo3[k+6]), a max3 tree over the 16 arcs — 16 + 16 + 8.
// Round 5: the network runs on the RAW bytes.  min / max commute with subtracting a constant, so max_arc min(ring - v) = max_arc min(ring) - v,
// and the dark polarity is the bright one on complemented bytes: v - ring = (255 - ring) - (255 - v) = (ring ^ 0xFF) - (v ^ 0xFF).  With the per-lane
// mask m = 0 (bright) / 0xFF (dark) the 16 differences are 16 v_xor_b32 — a plain VOP2 op of the class this chip issues at up to 1.4-1.6 wave-
// instructions per cycle and CU — instead of the 16 v_mad_i32_i24 of round 3 (VOP3: 0.85; profiles/r04_valu_issue_rates.txt), and one subtraction at the end.
__device__ __forceinline__ int fast_corner_score3(const int (&r)[16], int v, bool bright)
{
    const int m = bright ? 0 : 0xFF;
    int d[16], lo3[16], arc[16];
#pragma unroll
    for (int k = 0; k < 16; k++) asm("v_xor_b32 %0, %1, %2" : "=v"(d[k]) : "v"(m), "v"(r[k]));   // as asm: the compiler turns r ^ m into a select of r and ~r & 0xFF
    // (as asm as well: left to itself the compiler shares min(d[k+1], d[k+2]) between neighbours and ends up with 32 two-input minima)
    auto min3 = [](int x, int y, int z) { int o; asm("v_min3_i32 %0, %1, %2, %3" : "=v"(o) : "v"(x), "v"(y), "v"(z)); return o; };
    auto max3 = [](int x, int y, int z) { int o; asm("v_max3_i32 %0, %1, %2, %3" : "=v"(o) : "v"(x), "v"(y), "v"(z)); return o; };
#pragma unroll
    for (int k = 0; k < 16; k++) lo3[k] = min3(d[k], d[(k + 1) & 15], d[(k + 2) & 15]);
#pragma unroll
    for (int k = 0; k < 16; k++) arc[k] = min3(lo3[k], lo3[(k + 3) & 15], lo3[(k + 6) & 15]);
    int m6[6];
#pragma unroll
    for (int k = 0; k < 5; k++) m6[k] = max3(arc[3 * k], arc[3 * k + 1], arc[3 * k + 2]);
    m6[5] = arc[15];
    return max(max3(m6[0], m6[1], m6[2]), max3(m6[3], m6[4], m6[5])) - (v ^ m) - 1;
}

// TWO candidates per lane (round 5): gfx950 has a packed three-input minimum / maximum, but only for f16 (v_pk_minimum3_f16 / v_pk_maximum3_f16).
// A byte b under the bit pattern 0x6400 | b IS the f16 number 1024 + b (exponent 2^10, mantissa step 1): exact, ordered like b, never a NaN or a
// denormal — so the score network above runs on (candidate A, candidate B) pairs of such halves with the same 40 instructions one candidate needed.
// P[k] = ring_A[k] | ring_B[k] << 16 (bytes); mx = polarity masks (0 / 0xFF per half, as in fast_corner_score3) | 0x64006400: one v_xor_b32 per
// ring position complements AND biases both halves.  Returns the two halves 0x6400 + max_arc min(ring ^ m).
__device__ __forceinline__ uint32_t fast_arc_max_pk(const uint32_t (&P)[16], uint32_t mx)
{
    uint32_t d[16], lo3[16], arc[16];
#pragma unroll
    for (int k = 0; k < 16; k++) asm("v_xor_b32 %0, %1, %2" : "=v"(d[k]) : "v"(mx), "v"(P[k]));
    auto min3 = [](uint32_t x, uint32_t y, uint32_t z) { uint32_t o; asm("v_pk_minimum3_f16 %0, %1, %2, %3" : "=v"(o) : "v"(x), "v"(y), "v"(z)); return o; };
    auto max3 = [](uint32_t x, uint32_t y, uint32_t z) { uint32_t o; asm("v_pk_maximum3_f16 %0, %1, %2, %3" : "=v"(o) : "v"(x), "v"(y), "v"(z)); return o; };
#pragma unroll
    for (int k = 0; k < 16; k++) lo3[k] = min3(d[k], d[(k + 1) & 15], d[(k + 2) & 15]);
#pragma unroll
    for (int k = 0; k < 16; k++) arc[k] = min3(lo3[k], lo3[(k + 3) & 15], lo3[(k + 6) & 15]);
    uint32_t m6[6];
#pragma unroll
    for (int k = 0; k < 5; k++) m6[k] = max3(arc[3 * k], arc[3 * k + 1], arc[3 * k + 2]);
    m6[5] = arc[15];
    const uint32_t a = max3(m6[0], m6[1], m6[2]);
    return max3(a, m6[3], max3(m6[4], m6[5], m6[5]));
}

typedef unsigned short us2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ us2 as_us2(uint32_t x) { return __builtin_bit_cast(us2, x); }
__device__ __forceinline__ uint32_t as_u32(us2 x) { return __builtin_bit_cast(uint32_t, x); }
// bound_ctrl: the lane without a source (lane 0 / lane 63) reads 0 — a single v_mov_b32_dpp, no separate zero-initialisation of the result
// (those lanes' outer pixels are apron columns and are masked out of the scan anyway)
__device__ __forceinline__ uint32_t lane_from_below(uint32_t v) { return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0x138, 0xF, 0xF, true); }  // wave_shr:1
__device__ __forceinline__ uint32_t lane_from_above(uint32_t v) { return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0x130, 0xF, 0xF, true); }  // wave_shl:1

// Quick reject of 2 pixels held in the HIGH bytes of the two 16-bit fields (low bytes: anything).  Exact condition:
// (r0 or r8 darker than v-t) and (r4 or r12 darker), or the same for brighter.  Garbage in the low bytes can only add
// false positives: a true difference >= t+1 is >= 256t+1 in field units.  Returns non-zero fields for pixels that may be corners.
__device__ __forceinline__ us2 fr_pretest(us2 T, us2 C, us2 B, us2 L, us2 R, us2 t16)
{
    const us2 dk = __builtin_elementwise_max(__builtin_elementwise_min(T, B), __builtin_elementwise_min(L, R));
    const us2 br = __builtin_elementwise_min(__builtin_elementwise_max(T, B), __builtin_elementwise_max(L, R));
    const us2 z = __builtin_elementwise_max(__builtin_elementwise_sub_sat(C, dk), __builtin_elementwise_sub_sat(br, C));
    return __builtin_elementwise_sub_sat(z, t16);
}
// 1 in every 16-bit field that is non-zero (v_pk_min_u16; written as asm so that it is not turned back into compares)
__device__ __forceinline__ uint32_t fr_field_flags(us2 s, uint32_t ones)
{
    uint32_t r;
    asm("v_pk_min_u16 %0, %1, %2" : "=v"(r) : "v"(as_u32(s)), "v"(ones));
    return r;
}

// LDS of a workgroup: [0, th * PITCH) the item's pixel tile (later its score tile) | the pixel list of THIS item — 2 bytes of position + 1 byte of score per
// entry, as many entries as fit before the counters: an item of fewer rows than the tallest one of the launch has the longer list (round 5) | [off_cnt, total)
// the per-cell counters.  pcap_max: tuning / test cap on the list length (HS_FAST_PCAP, HS_FAST_TEST_SMALL_LISTS).
struct FastRowsLds { int32_t off_cnt, pcap_max, pcap_min, total; };

struct RowGeom {            // wave-uniform description of one work item in one image
    int img;
    int ncell, c0, gcell0;  // cells in this item, cell index of its first cell (within the level / within the image)
    uint32_t slot0; int ccap;
    int inv_w, inv_w1;
    int xoff, yoff;         // j0*wCell, i*hCell
    int th, ih, iw;         // tile rows, interior rows, interior width
    int off, ndw;           // tile column of level x is off + (x - iniX); dwords per tile row
    const uint8_t* rows;    // address of (a0, iniY): first dword of the first tile row
    size_t pitch;
    bool valid, aligned;
    int level;
};

// How the work units of a launch are spread over the work queues (host: fast_sched()).  The item list of an image is ordered expensive items
// first (the reduced levels, deepest first; level 0 last: hs_api.hip); a queue hands out its units in order, so its LAST units should be cheap —
// every wave still holds its current and its pre-grabbed next item when the queue runs dry, and with heavy items among them the launch ended with
// a third of its time spent draining (63 of 192 us at 32 frames; `tools/fast_wave_timeline.py`).  And there should be MANY queues: a queue is
// one counter, a counter serves same-address atomics one after the other, and with 8 of them for 30 000 items the grabs themselves were late
// (8 -> 32 queues: 0.175 -> 0.158 ms per 32 frames).
//   mode 1  the batch is a multiple of the queue count: a queue owns `par` whole images and walks them ITEM-major (item 0 of its images, item 1,
//           ...): its expensive items all go out early, its last units are the level-0 items of all its images
//   mode 2  the queue count is a multiple of the batch: an image is dealt round-robin to `par` queues (item i -> queue i % par), every one of
//           which sees the same mix of levels in the same order
//   mode 3  anything else: the item-major list of ALL images (item 0 of every image, item 1, ...) dealt round-robin to the queues
//   mode 0  contiguous ranges of the image-major order (HS_FAST_IMAGE_MAJOR=1: the scheme until the end of round 3, kept as a parity variant)
//   mode 4  FOLDED STATIC schedule for small launches (at most two units per workgroup; round 4): workgroup b takes unit b of the item-major
//           list of all images and then unit 2 * grid - 1 - b — the list is ordered expensive first, so the workgroups whose first item is the
//           cheapest get the (cheap) second items and the heaviest items run alone; no counter, no look-ahead grab.  At one stereo pair per
//           call the launch lasts as long as its slowest wave: with the look-ahead of the work queues the waves that started on the HEAVIEST
//           items were the first to grab a second one (`tools/fast_b1_timeline.py`: 28.8 us span, the slowest waves all "level 7 + another")
struct FastSched { int32_t nq_log, mode, par, par_log, per_q; uint32_t m_per_q, m_par, m_items; };     // m_x = ceil(2^32 / x): fast_div()
// floor(a / d) for 0 <= a < 2^31 and d >= 1 with m = ceil(2^32 / d): the high product is floor(a / d) or one more (its error a * (m * d - 2^32) / (d * 2^32)
// is below a / 2^32 < 1/2), one compare corrects it.  Wave-uniform operands: two scalar multiplies instead of the ~20-instruction v_rcp_iflag sequence
// of an integer division, twice per work item on the critical path of every wave.
__device__ __forceinline__ int fast_div(int a, int d, uint32_t m)
{
    int q = (int)__umulhi((uint32_t)a, m);
    if (d == 1) q = a;                                          // (ceil(2^32 / 1) does not fit 32 bits)
    if (q * d > a) q--;
    return q;
}
__device__ __forceinline__ int fast_queue_size(const FastSched& S, int qq, int total_work, int items_per_img)
{
    if (S.mode == 1) return S.per_q;
    if (S.mode == 2) return (items_per_img - (qq & (S.par - 1)) + S.par - 1) >> S.par_log;
    if (S.mode == 3) return (total_work - qq + (1 << S.nq_log) - 1) >> S.nq_log;
    return min(max(total_work - qq * S.per_q, 0), S.per_q);
}
// work unit w -> (item of the launch's item list, image)
__device__ __forceinline__ int unit_decode(int items_per_img, const FastSched& S, int w, int* img)
{
    struct { int img; } g;
    int item;
    if (S.mode == 1) {
        const int q = fast_div(w, S.per_q, S.m_per_q), u = w - q * S.per_q;
        item = fast_div(u, S.par, S.m_par);
        g.img = q * S.par + (u - item * S.par);
    } else if (S.mode == 2) {
        const int q = fast_div(w, S.per_q, S.m_per_q), u = w - q * S.per_q;
        g.img = q >> S.par_log;
        item = (u << S.par_log) + (q & (S.par - 1));
    } else if (S.mode == 3) {
        const int q = fast_div(w, S.per_q, S.m_per_q), u = w - q * S.per_q, gidx = (u << S.nq_log) + q;      // position in the item-major list of all S.par images
        item = fast_div(gidx, S.par, S.m_par);
        g.img = gidx - item * S.par;
    } else if (S.mode == 4) {
        item = fast_div(w, S.par, S.m_par);                          // w = position in the item-major list of all S.par images
        g.img = w - item * S.par;
    } else {
        g.img = fast_div(w, items_per_img, S.m_items);
        item = w - g.img * items_per_img;
    }
    *img = g.img;
    return item;
}
// An item's record as ONE 64-byte scalar load.  Read field by field (`items[i]`) the compiler fetches the dword-sized fields with scalar loads and the
// byte / half-word ones (off, ndw, ...) with a VECTOR load whose round trip through L2 the wave then waits for — once per work item, on its critical path.
typedef uint32_t hs_u32x16 __attribute__((ext_vector_type(16)));
__device__ __forceinline__ HsFastItem fast_item_load(const HsFastItem* p) { return __builtin_bit_cast(HsFastItem, hs_cload<hs_u32x16>(p)); }
// the item's record (`items` already points at the launch's first item) + the image -> geometry
__device__ __forceinline__ RowGeom geom_of(const HsFastItem& it, const HsImg0& img0, int img)
{
    RowGeom g;
    g.img = img;
    g.ncell = it.ncell; g.c0 = it.c0; g.gcell0 = it.gcell0; g.slot0 = it.slot0; g.ccap = it.ccap;
    g.inv_w = it.inv_w; g.inv_w1 = it.inv_w1;
    g.xoff = it.xoff; g.yoff = it.yoff;
    g.th = it.th; g.ih = g.th - 6; g.iw = it.iw;
    g.off = it.off; g.ndw = it.ndw;
    g.valid = it.th != 0;
    g.level = it.level;
    const uint8_t* base;
    if (it.base == nullptr) { base = hs_img0_ptr(img0, g.img); g.pitch = img0.row_stride; }
    else { base = it.base + (size_t)g.img * it.img_stride; g.pitch = (size_t)it.pitch; }
    g.aligned = (((uintptr_t)base | g.pitch) & 3) == 0;
    g.rows = base + (size_t)it.iniY * g.pitch + it.a0;
    return g;
}
__device__ __forceinline__ RowGeom row_geom(const HsFastItem* __restrict__ items, const HsImg0& img0, int items_per_img, const FastSched& S, int w)
{
    int img;
    const int item = unit_decode(items_per_img, S, w, &img);
    return geom_of(fast_item_load(items + item), img0, img);
}

// One wave per workgroup: its LDS instructions execute in issue order, so a later read sees an earlier write without any wait;
// FR_FENCE marks the hand-over points (define it as WAVE_LDS_FENCE() to wait for the LDS queue there).
#ifndef FR_FENCE
#define FR_FENCE() do {} while (0)
#endif
#define FR_MAXG 8            // cells per item (one per-cell counter each)
#ifndef FR_PAD
#define FR_PAD 16            // row padding of the LDS tiles, bytes (multiple of 16)
#endif

// Instruction-budget builds (tools/fast_instr_breakdown.sh): make EXTRA=-DFR_STOP=n cuts the item short after phase n — 1 scan A masks,
// 2 + list expansion, 3 + segment test, 4 + scores; results are meaningless (no candidates come out), only the SQ counters are read.
#ifndef FR_STOP
#define FR_STOP 99
#endif
#if defined(HS_FAST_PROFILE) || defined(HS_FAST_WAVES)   // make EXTRA=-DHS_FAST_WAVES: two real-time stamps per item and nothing else (tools/fast_wave_timeline.py)
__device__ unsigned long long g_fr_wave[4096 * 16];          // per workgroup: first stamp, stamp after the first item, last stamp, items, longest item (10 ns ticks), its work index,
                                                             // level << 32 | corners of the FIRST item, then its phase stamps: tile staged, next item prefetched, scan A done, corners scored, NMS done
extern "C" void hs_debug_fast_waves(unsigned long long* out /*[4096 * 16]*/)
{
    (void)hipDeviceSynchronize();
    (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_fr_wave), sizeof(unsigned long long) * 4096 * 16);
}
#endif
#ifdef HS_FAST_PROFILE       // make EXTRA=-DHS_FAST_PROFILE: per-phase cycle totals over all waves (tools/fast_phase_profile.py)
__device__ unsigned long long g_fr_prof[16];
#define FR_T(var) const unsigned long long var = __builtin_amdgcn_s_memtime()
#define FR_ACC(i, a, b) fr_acc[i] += (b) - (a)
extern "C" void hs_debug_fast_profile(unsigned long long* out16)
{
    (void)hipDeviceSynchronize();
    (void)hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_fr_prof), sizeof(unsigned long long) * 16);
    unsigned long long z[16] = {};
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_fr_prof), z, sizeof(z));
}
#else
#define FR_T(var)
#define FR_ACC(i, a, b)
#endif
#if defined(HS_FAST_WAVES)
#define FR_W(i) do { if (fr_items == 0) fr_ph[i] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define FR_W(i) do {} while (0)
#endif

// inclusive prefix sum over the 64 lanes (DPP row shifts + row broadcasts, no LDS)
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

// TR = tile rows held in LDS (6 + 8*RS*blocks); the item's tile (th <= TR rows) is fetched with 16-byte loads, 64/LPR rows per load
// KEYS: the quadtree's geometric keys are produced by this launch (calls of <= 16 frames: hs_api.hip).  A template parameter, not a runtime
// flag: the large-batch instantiation pays neither registers nor instructions for a path it never runs (round 4 carried it as a runtime
// flag: 139 -> 147 VGPRs and +1 M wave-instructions per 32 frames for nothing).
// Launch bounds: the narrow instances of the standard tile heights run 16 workgroups per CU, which 4 waves per SIMD = 128 VGPRs must allow (the
// per-item list placement of round 5 took the keyed one to 132 without the hint: 12 per CU, +20 % at one pair per call); the others are bound by LDS.
template <int LC, int TR, bool KEYS>
__global__ __launch_bounds__(64, (LC == 5 && TR <= 44 ? 4 : 1)) void k_fast_rows(const HsFastItem* __restrict__ items, HsImg0 img0, int fast_th,
                                                  uint2* __restrict__ cand,
                                                  int32_t* __restrict__ cell_count, uint64_t cand_img_stride,
                                                  int total_cells, int items_per_img, int total_work, FastRowsLds lds, int force_scan_b,
                                                  uint32_t* __restrict__ overflow, uint32_t overflow_stride, uint32_t epoch, int item_first, uint32_t spill_base, FastSched S,
                                                  const HsFastQt* __restrict__ qt, uint32_t* __restrict__ qhist, unsigned long long* __restrict__ qbest,
                                                  uint32_t qhist_img_stride, uint32_t qbest_img_stride)
{
    constexpr int COLS = 1 << LC;            // dwords per tile row
    constexpr int RS = 64 / COLS;            // half-waves working on different rows in the scans
    constexpr int PITCH = 4 * COLS + FR_PAD; // bytes per row of the pixel tile and of the score tile (padded: vertical neighbours in different banks)
    constexpr int PD = PITCH / 4;
    constexpr int LPR = COLS / 4;            // lanes per tile row in the 16-byte staging loads
    constexpr int RPL = 64 / LPR;            // tile rows per staging load
    constexpr int NL = (TR + RPL - 1) / RPL; // staging loads per lane
    constexpr int BR = 8;                    // rows per lane and scan block
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    uint8_t* const tile = smem;
    uint8_t* const score = smem;                                                 // the score tile REUSES the pixel tile once every corner is scored
    uint32_t* const cellcnt = reinterpret_cast<uint32_t*>(smem + lds.off_cnt);   // survivors per cell of the item
    uint32_t* const queue = overflow + (size_t)(epoch & 3) * HS_FAST_QUEUE_DWORDS;   // 8 work counters, 128 bytes apart
    if (blockIdx.x == 0 && threadIdx.x < HS_FAST_NQ_MAX) overflow[(size_t)((epoch + 2) & 3) * HS_FAST_QUEUE_DWORDS + threadIdx.x * 32] = 0u;
    uint32_t* const ovf = overflow + 4 * HS_FAST_QUEUE_DWORDS + spill_base + (size_t)blockIdx.x * overflow_stride;   // this wave's spill area for scored corners (list overflow only)
    const uint32_t* const tile32 = reinterpret_cast<const uint32_t*>(tile);
    const uint32_t* const score32 = reinterpret_cast<const uint32_t*>(score);

    items += item_first;                                         // this launch covers items [item_first, item_first + items_per_img) of every image
    const int tid = threadIdx.x;
    const int col = tid & (COLS - 1), sub = tid >> LC;
    const int ld_row = tid / LPR, ld_c16 = tid % LPR;
    const int t = min(max(fast_th, 0), 255);
    const uint32_t kbias = (0x80u - (uint32_t)((t + 1) >> 2)) * 0x01010101u;     // see scan A
    constexpr int RO[16] = { 3 * PITCH + 0, 3 * PITCH + 1, 2 * PITCH + 2, 1 * PITCH + 3, 0 * PITCH + 3, -1 * PITCH + 3, -2 * PITCH + 2, -3 * PITCH + 1,
                             -3 * PITCH + 0, -3 * PITCH - 1, -2 * PITCH - 2, -1 * PITCH - 3, 0 * PITCH - 3, 1 * PITCH - 3, 2 * PITCH - 2, 3 * PITCH - 1 };

    // Work distribution: the item list is cut into 8 contiguous ranges, one per queue (home queue = blockIdx % 8, i.e. the waves that share
    // an XCD under round-robin placement, so that neighbouring items — shared apron lines — meet in one L2; placement is a speed
    // matter only).  A wave takes its FIRST item statically (range start + its rank among the queue's waves) and every further item from
    // the queue's atomic counter: corner-rich items cost several times more than flat ones, and with static strides the slowest of the
    // 2816 waves (5 items each) set the kernel time while the average wave idled for a third of it (SQ counters: 68 % VALU issue, 27 % of
    // wave cycles waiting).  The grab for the item after next is issued right after the next tile's prefetch and consumed an item later,
    // so its latency is never exposed; 8 counters on lines of their own see ~11 grabs/us each.  When a queue runs dry its waves steal
    // from the other queues, and leave when all eight are empty (every wave reaches that exit: the counters only grow).
    // Four counter sets rotate between launches: a launch uses set `epoch & 3` and its first workgroup zeroes set `(epoch + 2) & 3` for the
    // launch after next — no memset launch in the chain.  Two consecutive launches of a handle may run CONCURRENTLY (level 0 beside the
    // pyramid, the other levels after it: hs_api.hip), which is why the set a launch zeroes is not the next launch's.
    const int nq = 1 << S.nq_log, per_x = S.per_q, wpx = gridDim.x >> S.nq_log;
    int q = (int)(blockIdx.x & (nq - 1));                        // current queue (wave-uniform); blockIdx % 8 = the XCD under round-robin placement
    auto q_size = [&](int qq) { return fast_queue_size(S, qq, total_work, items_per_img); };
    // one lane asks; the value comes back in a VGPR and is only read (readfirstlane) an item later
    auto grab_async = [&](int qq) -> uint32_t {
        uint32_t v = 0;
        if (tid == 0) v = atomicAdd(&queue[qq * 32], 1u);
        return v;
    };
    // raw counter value -> work unit, stealing from the other queues when the home queue is exhausted; -1 = nothing left anywhere.
    // A thief first LOOKS at all eight counters with one load (lanes 0..7; plain loads do not serialise like atomics) and only then grabs
    // from the fullest queue: at the end of a launch thousands of waves would otherwise queue up failing atomics on eight addresses.
    const bool fold = S.mode == 4;                               // folded static schedule: no counters at all
    int fold_next = -1;
    if (fold) { const int u2 = 2 * (int)gridDim.x - 1 - (int)blockIdx.x; if (u2 < total_work) fold_next = u2; }
    const bool dynamic = !fold && wpx < per_x;                   // fewer items than waves: the static first items are the whole job
    auto resolve = [&](uint32_t raw) -> int {
        if (fold) { const int r = fold_next; fold_next = -1; return r; }
        if (!dynamic) return -1;
        int idx = (int)__builtin_amdgcn_readfirstlane(raw) + wpx;
        if (idx < q_size(q)) return q * per_x + idx;
        for (int tries = 0; tries < 8; tries++) {
            int rem = 0;
            if (tid < nq) rem = q_size(tid) - wpx - (int)__hip_atomic_load(&queue[tid * 32], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            int best = 0, best_q = -1;
            for (int i = 0; i < nq; i++) { const int r = __builtin_amdgcn_readlane(rem, i); if (r > best) { best = r; best_q = i; } }
            if (best_q < 0) return -1;
            q = best_q;
            idx = (int)__builtin_amdgcn_readfirstlane(grab_async(q)) + wpx;      // synchronous: only at the very end of the launch
            if (idx < q_size(q)) return q * per_x + idx;
        }
        return -1;
    };
    int w = fold ? ((int)blockIdx.x < total_work ? (int)blockIdx.x : -1)
                 : (int)(blockIdx.x >> S.nq_log) < q_size(q) ? q * per_x + (int)(blockIdx.x >> S.nq_log) : (dynamic ? resolve(grab_async(q)) : -1);
    if (w < 0) return;
    uint32_t raw_next = dynamic ? grab_async(q) : 0u;            // the second item

    if (tid < FR_MAXG) cellcnt[tid] = 0;

#ifdef HS_FAST_PROFILE
    unsigned long long fr_acc[10] = {};
    FR_T(t_kernel0);
#endif
#if defined(HS_FAST_PROFILE) || defined(HS_FAST_WAVES)
    const unsigned long long fr_real0 = __builtin_amdgcn_s_memrealtime();     // the 100 MHz real-time counter: the same on every CU (s_memtime is not)
    unsigned long long fr_prev = fr_real0, fr_first = 0, fr_long = 0, fr_long_w = 0, fr_items = 0;
    unsigned fr_codes = 0;
    unsigned long long fr_ph[6] = {};                         // first item: level << 32 | corners, phase stamps
#endif
    hs_u32x4 pre[NL];
    RowGeom g = row_geom(items, img0, items_per_img, S, w);
    // All NL loads are issued unconditionally (rows beyond the tile re-read its last row, columns beyond it the last needed
    // 16 bytes): predicating them makes the compiler copy the whole register array at every merge point.
    auto prefetch = [&](const RowGeom& q) {
        const int rlast = max(q.th - 1, 0);
        const uint32_t coff = 16u * (uint32_t)min(ld_c16, (q.ndw - 1) >> 2);
        const uint8_t* base = hs_uniform_ptr(q.rows);
        if (q.aligned) {
#pragma unroll
            for (int k = 0; k < NL; k++)
                pre[k] = hs_gload_off<hs_u32x4>(base, (uint32_t)min(RPL * k + ld_row, rlast) * (uint32_t)q.pitch + coff);
        } else {                                                 // caller's frame with an odd base or stride: unaligned dword loads
            struct __attribute__((packed, aligned(1))) U32 { uint32_t v; };
#pragma unroll
            for (int k = 0; k < NL; k++) {
                const HS_GLOBAL U32* p = (const HS_GLOBAL U32*)((const HS_GLOBAL uint8_t*)(uintptr_t)base + ((uint32_t)min(RPL * k + ld_row, rlast) * (uint32_t)q.pitch + coff));
                pre[k] = hs_u32x4{p[0].v, p[1].v, p[2].v, p[3].v};
            }
        }
    };
    prefetch(g);

    for (; w >= 0;) {
        int32_t* const cnt_out = &cell_count[(size_t)g.img * total_cells + g.gcell0];
        if (!g.valid) {
            if (tid == 0) cnt_out[0] = 0;
            w = resolve(raw_next);
            if (w >= 0) { g = row_geom(items, img0, items_per_img, S, w); prefetch(g); if (dynamic) raw_next = grab_async(q); }
            continue;
        }
        const RowGeom cur = g;
        // The next work unit: its queue counter value was requested an item ago and arrives with the tile loads the staging below waits for anyway;
        // its item record (one 64-byte scalar load) is requested BEFORE the staging stores, so that it is there when they are done.
        FR_T(t0);
        const int w_next = resolve(raw_next);
        int img_next = 0;
        const int item_next = w_next >= 0 ? unit_decode(items_per_img, S, w_next, &img_next) : 0;
        const HsFastItem it_next = fast_item_load(items + item_next);      // (no next unit: item 0's record, unused)
        // ---- stage the tile
#pragma unroll
        for (int k = 0; k < NL; k++)
            if (RPL * (k + 1) <= TR || RPL * k + ld_row < TR)
                *reinterpret_cast<hs_u32x4*>(&tile[(RPL * k + ld_row) * PITCH + 16 * ld_c16]) = pre[k];
        FR_FENCE();
        FR_T(t1);
        FR_ACC(0, t0, t1);
        FR_W(1);
        // the item's pixel list starts where its tile ends (rows th .. TR of the tile region are not used by this item)
        const int list_off = cur.th * PITCH;
        const int pcap = min(lds.pcap_max, (int)((((uint32_t)(lds.off_cnt - list_off) * 43691u) >> 17) & ~15u));      // floor(bytes / 3) entries, a multiple of 16
        uint16_t* const plist = reinterpret_cast<uint16_t*>(smem + list_off);        // pixel entries: row<<8 | column (| 0x8000)
        uint8_t* const pscore = smem + list_off + 2 * pcap;                          // score of corner i of the list
        const int inv_w = cur.inv_w, inv_w1 = cur.inv_w1;
        const size_t slot_base = (size_t)cur.img * cand_img_stride + cur.slot0;
        const RowGeom g_next = geom_of(it_next, img0, img_next);       // (unconditional: a use inside the branch would sink the record's load below the staging stores)
        if (w_next >= 0) { g = g_next; prefetch(g); if (dynamic) raw_next = grab_async(q); }   // the next tile: in flight during the passes
        FR_T(t2);
        FR_ACC(1, t1, t2);
        FR_W(2);

        // ---- the quadtree's geometric keys (round 4): a survivor's key is xkey[x] | ykey[y] (HsFastQt).  The item's slices of the two tables are
        //      fetched NOW into registers — lane j holds the keys of interior columns 2 j, 2 j + 1 (kx0) and 128 + 2 j, 129 + 2 j (kx1; wide tiles
        //      only) and of interior rows 2 j, 2 j + 1 (ky0; a cell has at most 125 rows) — and read per survivor with ds_bpermute at the very end
        //      of the item, so their latency is never exposed.  The survivors' key histogram and the best candidate per key go to global memory
        //      with fire-and-forget atomics: the quadtree kernel starts from them instead of gathering the candidates.
        uint32_t kx0 = 0, kx1 = 0, ky0 = 0;
        bool keys_on = false;
        uint32_t* khist = nullptr; unsigned long long* kbest = nullptr;
        if constexpr (KEYS) {
            typedef uint32_t u32x8 __attribute__((ext_vector_type(8)));
            const HsFastQt Q = __builtin_bit_cast(HsFastQt, hs_cload<u32x8>(qt + cur.level));      // one s_load_dwordx8
            keys_on = Q.enabled != 0;
            if (keys_on) {
                struct __attribute__((packed, aligned(2))) U32 { uint32_t v; };
                typedef const HS_GLOBAL U32* gu32;
                const uint8_t* xb = hs_uniform_ptr(reinterpret_cast<const uint8_t*>(Q.xkey + (cur.xoff + 3)));
                const uint8_t* yb = hs_uniform_ptr(reinterpret_cast<const uint8_t*>(Q.ykey + (cur.yoff + 3)));
                kx0 = ((gu32)((const HS_GLOBAL uint8_t*)(uintptr_t)xb + 4u * (uint32_t)tid))->v;
                if (COLS > 32) kx1 = ((gu32)((const HS_GLOBAL uint8_t*)(uintptr_t)xb + (256u + 4u * (uint32_t)tid)))->v;
                ky0 = ((gu32)((const HS_GLOBAL uint8_t*)(uintptr_t)yb + 4u * (uint32_t)tid))->v;
                khist = qhist + (size_t)cur.img * qhist_img_stride + Q.hist_off;
                kbest = qbest + (size_t)cur.img * qbest_img_stride + Q.best_off;
            }
        }
        const int c_first = cur.off + 3;                         // tile column of the first interior pixel
        uint32_t vmask8 = 0;                                     // this lane's pixels that are interior columns: byte j = pixel j, one bit per row of a scan block
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int px = 4 * col + j - c_first;
            if (px >= 0 && px < cur.iw) vmask8 |= 0xFFu << (8 * j);
        }
        int npx = 0;                                             // wave-uniform list length
        int n_done = 0, n_ovf = 0;                               // scored corners at the head of the list (entries [n_done, npx) are scan codes); corners spilled
        int n_emit = 0;                                          // survivors of the item so far (wave-uniform)

        // ---- pixel list -> corners -> scores
        // ---- scan codes [n_done, npx) of the list -> corners -> scores; the scored corners stay at the head of the list
        auto corners_and_scores = [&]() {
            if (FR_STOP <= 2) { npx = n_done = 0; return; }
            FR_FENCE();
            FR_T(tc0);
#if defined(HS_FAST_WAVES)
            const int fr_n0 = n_done;
#endif
            int n_corner = n_done;
            int nend = npx;                                  // list end including the re-queued pixels (below)
            // Two candidates per lane and iteration: A = entry i0 + lane, B = entry i0 + 64 + lane (neighbouring lanes keep neighbouring entries: the
            // ring reads of a wave-instruction stay spatially close).  Decoding, ring reads and the exact compass test are per candidate; the score
            // network — 40 of the ~110 instructions a candidate cost — serves both (fast_arc_max_pk).
            for (int i0 = n_done; i0 < nend; i0 += 128) {
                const int iA = i0 + tid, iB = iA + 64;
                const bool actA = iA < nend, actB = iB < nend;
                const int codeA = plist[actA ? iA : 0], codeB = plist[actB ? iB : 0];
                int pyA, pxA, pyB, pxB;
                auto decode = [&](int code, int& py, int& px) {
                    const int eb = code & 31, ew = (code >> 5) & 63;                 // bit of the scan mask (8 * pixel + row), lane that held the word
                    const int el = (ew & 48) | ((ew - ((code & 24) >> 1)) & 15);     // lane that found it: byte j sits 4 j lanes up in its row of 16
                    py = BR * (RS * ((code >> 11) & 15) + (el >> LC)) + (eb & 7);
                    px = 4 * (el & (COLS - 1)) + (eb >> 3) - c_first;
                };
                decode(codeA, pyA, pxA); decode(codeB, pyB, pxB);
                const bool redo_darkA = codeA & 0x8000, redo_darkB = codeB & 0x8000;   // re-queued: passed both compass tests and is no bright corner
                const uint8_t* ctrA = &tile[(pyA + 3) * PITCH + c_first + pxA];
                const uint8_t* ctrB = &tile[(pyB + 3) * PITCH + c_first + pxB];
                const int vA = ctrA[0], vB = ctrB[0];
                int rA[16], rB[16];
#pragma unroll
                for (int k = 0; k < 16; k++) { rA[k] = ctrA[RO[k]]; rB[k] = ctrB[RO[k]]; }
                // ONE polarity per pixel, and the corner SCORE as the segment test.  A 9-arc contains two adjacent compass points (ring 0, 4,
                // 8, 12), so a dark corner passes the exact compass test "(r0 or r8 darker) and (r4 or r12 darker)" and a bright corner its
                // mirror image; the pixel goes through the score network of the polarity that can still succeed: it is a corner of that
                // polarity exactly when max_arc min(d) > t, i.e. score >= t (cornerScore's own definition).
                const bool dark_okA = max(min(rA[0], rA[8]), min(rA[4], rA[12])) < vA - t, bright_okA = min(max(rA[0], rA[8]), max(rA[4], rA[12])) > vA + t;
                const bool dark_okB = max(min(rB[0], rB[8]), min(rB[4], rB[12])) < vB - t, bright_okB = min(max(rB[0], rB[8]), max(rB[4], rB[12])) > vB + t;
                const bool brightA = bright_okA && !redo_darkA, brightB = bright_okB && !redo_darkB;
                uint32_t P[16];
#pragma unroll
                for (int k = 0; k < 16; k++) P[k] = (uint32_t)rA[k] | ((uint32_t)rB[k] << 16);
                const int mA = brightA ? 0 : 0xFF, mB = brightB ? 0 : 0xFF;
                const uint32_t R = fast_arc_max_pk(P, 0x64006400u | (uint32_t)mA | ((uint32_t)mB << 16));
                int scA = (int)(R & 0x3FFu) - (vA ^ mA) - 1, scB = (int)((R >> 16) & 0x3FFu) - (vB ^ mB) - 1;
                bool cornerA = scA >= t && actA && (dark_okA || bright_okA), cornerB = scB >= t && actB && (dark_okB || bright_okB);
                // Pixels that pass BOTH compass tests (6-8 % of the candidates on the reduced levels, 0.1 % on level 0) and are no bright
                // corner need the dark test as well: they go back to the END of the list with bit 15 set and fill the lanes of the last,
                // partly empty iteration (running the second test in place would double the cost of nearly every iteration).  The last
                // iteration itself (whose lanes already cover the list end) and a full list run the second test in place.
                const bool redoA = actA && dark_okA && bright_okA && !cornerA && !redo_darkA, redoB = actB && dark_okB && bright_okB && !cornerB && !redo_darkB;
                if (__any(redoA || redoB)) {
                    if (i0 + 128 < nend && nend + 128 <= pcap) {
                        const int slotA = wave_append(redoA, nend);
                        if (redoA) plist[slotA] = (uint16_t)(codeA | 0x8000);
                        const int slotB = wave_append(redoB, nend);
                        if (redoB) plist[slotB] = (uint16_t)(codeB | 0x8000);
                    } else {
                        const uint32_t R2 = fast_arc_max_pk(P, 0x64FF64FFu);
                        const int s2A = (int)(R2 & 0x3FFu) - (vA ^ 0xFF) - 1, s2B = (int)((R2 >> 16) & 0x3FFu) - (vB ^ 0xFF) - 1;
                        if (redoA && s2A >= t) { cornerA = true; scA = s2A; }
                        if (redoB && s2B >= t) { cornerB = true; scB = s2B; }
                    }
                }
                FR_FENCE();                                // this iteration's reads precede the in-place compaction writes
                const int slotA = wave_append(cornerA, n_corner);
                if (cornerA) {
                    const int gc = (pxA * inv_w) >> 16;          // cell of the item; its columns start at 1 + gc*(wcell+1)
                    plist[slotA] = (uint16_t)(((pyA + 1) << 8) | (1 + pxA + gc));     // score tile coordinates: what nms_and_emit reads
                    pscore[slotA] = (uint8_t)scA;
                }
                const int slotB = wave_append(cornerB, n_corner);
                if (cornerB) {
                    const int gc = (pxB * inv_w) >> 16;
                    plist[slotB] = (uint16_t)(((pyB + 1) << 8) | (1 + pxB + gc));
                    pscore[slotB] = (uint8_t)scB;
                }
            }
            FR_FENCE();
            FR_T(tc1);
            FR_ACC(4, tc0, tc1);
            if (FR_STOP <= 4 && FR_STOP >= 3) { npx = n_done = 0; return; }
#if defined(HS_FAST_WAVES)
            if (fr_items == 0) fr_codes += (unsigned)(nend - fr_n0);
#endif
            npx = n_done = n_corner;
            FR_FENCE();
        };
        // the list is full of scored corners (saturated image): spill them (row, column, score) to this wave's global area
        auto spill_corners = [&]() {
            for (int i = tid; i < n_done; i += 64) ovf[n_ovf + i] = ((uint32_t)plist[i] << 8) | pscore[i];
            n_ovf += n_done;
            npx = n_done = 0;
        };
        // make room for `need` more scan codes: score what is pending (corners are ~1/3 of the codes); spill only if that is not enough
        auto make_room = [&](int need) {
            corners_and_scores();
            if (npx + need > pcap) spill_corners();
        };

        // ---- pixel list (score tile coordinates) -> strict 3x3 NMS -> the cells' slots, in no particular order (the quadtree kernel
        //      orders a cell's records by (y, x) when it gathers them)
        auto nms_and_emit = [&]() {
            FR_FENCE();
            for (int i0 = 0; i0 < npx; i0 += 64) {
                const int i = i0 + tid;
                const bool act = i < npx;
                const int pos = plist[act ? i : 0];
                const int r = pos >> 8, sc = pos & 255;
                const uint8_t* p = &score[r * PITCH + sc];
                const int s = p[0];
                // all eight neighbours are read before any is compared (&&-chains compile to eight dependent LDS round trips)
                const int n0 = p[-PITCH - 1], n1 = p[-PITCH], n2 = p[-PITCH + 1], n3 = p[-1], n4 = p[1], n5 = p[PITCH - 1], n6 = p[PITCH], n7 = p[PITCH + 1];
                const int nmax = max(max(max(n0, n1), max(n2, n3)), max(max(n4, n5), max(n6, n7)));
                const bool keep = act & (s > nmax);
                const int slot = wave_append(keep, n_emit);     // one running count per ITEM: its records are contiguous from slot0
                const int gc = ((sc - 1) * inv_w1) >> 16;
                const int px = sc - 1 - gc;                      // interior column within the item
                uint32_t gk = 0;
                if (KEYS && keys_on) {                           // (wave-uniform; every lane takes part in the permutes: inactive lanes read lane 0)
                    const int pxc = act ? px : 0, ryc = act ? r - 1 : 0;
                    uint32_t vx = (uint32_t)__builtin_amdgcn_ds_bpermute(((pxc & 127) >> 1) << 2, (int)kx0);
                    if (COLS > 32) { const uint32_t v1 = (uint32_t)__builtin_amdgcn_ds_bpermute(((pxc & 127) >> 1) << 2, (int)kx1); if (pxc >= 128) vx = v1; }
                    const uint32_t vy = (uint32_t)__builtin_amdgcn_ds_bpermute(((ryc & 127) >> 1) << 2, (int)ky0);
                    gk = ((vx >> (16 * (pxc & 1))) & 0xFFFFu) | ((vy >> (16 * (ryc & 1))) & 0xFFFFu);
                }
                if (keep) {
                    const uint32_t xy = ((uint32_t)(r - 1 + 3 + cur.yoff) << 16) | (uint32_t)(px + 3 + cur.xoff), sk = ((uint32_t)s << 24) | (uint32_t)(cur.c0 + gc);
                    cand[slot_base + (size_t)slot] = make_uint2(xy, sk);
                    if (KEYS && keys_on) {
                        // k_quadtree's `offer`: maximum response, first in (cell, y, x) order on ties (ORBExtractor.cpp:381-400)
                        const unsigned long long order = ((unsigned long long)(sk & 0xFFFFFFu) << 32) | xy;
                        const unsigned long long key = ((unsigned long long)(sk >> 24) << 56) | (0x00FFFFFFFFFFFFFFull - order);
                        __hip_atomic_fetch_add(&khist[gk >> 1], 1u << ((gk & 1) * 16), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        __hip_atomic_fetch_max(&kbest[gk], key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                }
            }
            npx = 0;
        };

        // ---- the set bits of M (one per pixel of this lane's scan block) -> pixel list.  SCORES: bit 4r+j, entries in score tile
        //      coordinates; otherwise bit 8j + r, entries relative to the interior.  `row0`: tile/score row of r = 0.
        // Scan A entries are CODES (block<<11 | lane<<5 | bit of M), decoded on dense lanes by corners_and_scores: the per-lane bit loop
        // below runs for as many rounds as the busiest lane has hits, so every instruction in it counts ~10x.
        auto emit_mask = [&](uint32_t M, int row0, bool scores, int blk) {
            const uint32_t code0 = ((uint32_t)blk << 11) | ((uint32_t)tid << 5);
            const int incl = wave_scan_incl((int)__popc(M));
            const int total = __builtin_amdgcn_readlane(incl, 63);
            if (total == 0) return;
            if (npx + total > pcap) { if (scores) nms_and_emit(); else make_room(min(total, pcap)); }
            if (total <= pcap) {
                int pos = npx + incl - (int)__popc(M);
                while (M) {
                    const int b = __ffs((int)M) - 1;
                    M &= M - 1;
                    plist[pos++] = scores ? (uint16_t)(((row0 + (b >> 2)) << 8) | (4 * col + (b & 3))) : (uint16_t)(code0 | (uint32_t)b);
                }
                npx += total;
            } else {                                             // a block alone overflows the list (saturated image): one row at a time
                for (int r = 0; r < BR; r++) {
                    uint32_t Mr = M & (scores ? (0xFu << (4 * r)) : (0x01010101u << r));
                    const int incl_r = wave_scan_incl((int)__popc(Mr));
                    const int total_r = __builtin_amdgcn_readlane(incl_r, 63);      // <= 256 <= pcap
                    if (npx + total_r > pcap) { if (scores) nms_and_emit(); else make_room(total_r); }
                    int pos = npx + incl_r - (int)__popc(Mr);
                    while (Mr) {
                        const int b = __ffs((int)Mr) - 1;
                        Mr &= Mr - 1;
                        plist[pos++] = scores ? (uint16_t)(((row0 + r) << 8) | (4 * col + (b & 3))) : (uint16_t)(code0 | (uint32_t)b);
                    }
                    npx += total_r;
                }
            }
        };

        // ---- scan A: quick reject.  A lane takes 4 pixel columns x 8 rows per block: 14 tile rows in registers, the horizontal ring
        //      pixels from the neighbouring lanes (DPP wave shifts), all compares as v_pk_*_u16 on two pixels at a time.
        FR_T(t3);
        {
            const int yend = 3 + cur.ih;
            const int nblock = (cur.ih + BR * RS - 1) / (BR * RS);               // uniform trip count (the half-waves take different rows)
            for (int b = 0; b < nblock; b++) {
                const int y0 = 3 + BR * (RS * b + sub);
                uint32_t M = 0;
                {
                    // Four pixels per 32-bit operation: every byte holds a pixel reduced to 6 bits (q = v >> 2), which leaves two guard bits per
                    // byte, so byte-wise differences never borrow across bytes.  With bias = 0x80 - k per byte, bit 7 of (q_a + bias - q_b) is
                    // set exactly when q_a - q_b >= k (the byte stays within [0x80 - k - 0x3F, 0x80 - k + 0x3F]).  With k = (t + 1) / 4, "b darker than a by more than t" implies q_a - q_b >= k (floor((a - m) / 4)
                    // <= floor(a / 4) - floor(m / 4)), so the byte test is a NECESSARY condition for the exact compass test of the 16-bit version it
                    // replaces (measured: 3.7 % of the pixels pass instead of 3.6 %); the segment test on the full bytes follows as before.
                    // ~26 plain 32-bit ALU operations per row of 4 pixels instead of ~34 packed-16-bit ones, which also issue slower.
                    const uint32_t* tp = tile32 + (y0 - 3) * PD + col;
                    uint32_t Q[BR + 6], G[BR + 6];
#pragma unroll
                    for (int k = 0; k < BR + 6; k++) { Q[k] = (tp[k * PD] >> 2) & 0x3F3F3F3Fu; G[k] = Q[k] + kbias; }
                    // "x darker than C": bit 7 of (q_C + bias) - q_x; "x brighter": bit 7 of q_x + (bias - q_C).  Vertically the four differences
                    // of a pixel are shared with the pixels three rows above and below it: with D1[y] = G[y] - Q[y+3] and D2[y] = G[y+3] - Q[y],
                    // row c sees "top or bottom darker" = D2[c-3] | D1[c] and "top or bottom brighter" = D1[c-3] | D2[c] — two subtractions per
                    // tile row instead of four per centre row (the same 32-bit values as the direct form, so the masks are the same bits).
                    uint32_t D1[BR + 3], D2[BR + 3];
#pragma unroll
                    for (int y = 0; y < BR + 3; y++) { D1[y] = G[y] - Q[y + 3]; D2[y] = G[y + 3] - Q[y]; }
#pragma unroll
                    for (int r = 0; r < BR; r++) {
                        const uint32_t qC = Q[r + 3];
                        const uint32_t gC = G[r + 3], nC = kbias - qC;
                        const uint32_t qCm = lane_from_below(qC), qCp = lane_from_above(qC);
                        const uint32_t qL = __builtin_amdgcn_alignbyte(qC, qCm, 1);          // the pixels 3 columns to the left / right of this lane's four
                        const uint32_t qR = __builtin_amdgcn_alignbyte(qCp, qC, 3);
                        const uint32_t dark = (D2[r] | D1[r + 3]) & ((gC - qL) | (gC - qR));           // (top or bottom darker) and (left or right darker)
                        const uint32_t bright = (D1[r] | D2[r + 3]) & ((qL + nC) | (qR + nC));
                        M = (M >> 1) | ((dark | bright) & 0x80808080u);                      // row r ends up at bit r of its pixel's byte
                    }
                    const int nrow = min(max(yend - y0, 0), BR);                     // rows of this block inside the interior
                    M &= vmask8 & (((1u << nrow) - 1u) * 0x01010101u);
                }
                // The maybe-pixels come in clusters (both sides of a slanted edge): on the bench scene a block has its ~120 hits in ~14 of
                // its 64 lanes, and the bit loop of emit_mask runs as long as the busiest lane has hits (17.7 rounds on average).  Each lane
                // therefore keeps byte 0 (its first pixel column) and takes byte j from the lane 4 j to its left in its row of 16 lanes (DPP
                // row rotate): the four columns of a lane's word are 16 px apart and a cluster's columns land in different lanes (9.3
                // rounds; 8.4 with the lanes 16 j apart, but then neighbouring list entries come from areas 64 px apart and the ring reads
                // of the segment test collide in the LDS banks).  corners_and_scores undoes it when it decodes a code.
                {
                    const uint32_t t1 = (uint32_t)__builtin_amdgcn_mov_dpp((int)M, 0x124, 0xF, 0xF, false), t2 = (uint32_t)__builtin_amdgcn_mov_dpp((int)M, 0x128, 0xF, 0xF, false),
                                   t3 = (uint32_t)__builtin_amdgcn_mov_dpp((int)M, 0x12C, 0xF, 0xF, false);      // row_ror:4 / 8 / 12
                    M = (M & 0xFFu) | (t1 & 0xFF00u) | (t2 & 0xFF0000u) | (t3 & 0xFF000000u);
                }
                if (FR_STOP <= 1) { if (M == 0x12345u) plist[tid] = 1; continue; }      // keeps M live
                emit_mask(M, y0, false, b);
            }
            FR_W(3);
            corners_and_scores();
            FR_W(4);
        }
        FR_T(t4);
        FR_ACC(2, t3, t4);                                       // scan A including corners_and_scores (4, 5 are subsets)
        if (FR_STOP <= 4) n_done = 0;
        // ---- the pixel tile is dead: its LDS becomes the dense score tile (rows 0..ih+1, zero except at the corners)
        if (FR_STOP > 4) {
            if constexpr (TR <= 54) {                            // rows 0 .. ih+1 = th-5 in chunks of 1 KB at compile-time offsets; never into the list, which starts at
                const int zbytes = (cur.ih + 2) * PITCH;         // row th and is live here (wide tiles: a chunk is shorter than the four rows in between)
#pragma unroll
                for (int k = 0; k < ((TR - 4) * PITCH + 1023) / 1024; k++) {
                    if constexpr (4 * PITCH >= 1024) { if (1024 * k < zbytes) *reinterpret_cast<uint4*>(score + 1024 * k + 16 * tid) = make_uint4(0, 0, 0, 0); }          // wave-uniform test
                    else { if (1024 * k + 16 * tid < zbytes) *reinterpret_cast<uint4*>(score + 1024 * k + 16 * tid) = make_uint4(0, 0, 0, 0); }                          // narrow tiles: four rows are less than a chunk
                }
            } else {
                for (int i = tid * 16; i < (cur.ih + 2) * PITCH; i += 64 * 16) *reinterpret_cast<uint4*>(score + i) = make_uint4(0, 0, 0, 0);
            }
        }
        for (int i = tid; i < n_done; i += 64) {
            const int pos = plist[i];
            score[(pos >> 8) * PITCH + (pos & 255)] = pscore[i];
        }
        if (n_ovf > 0) {                                         // spilled corners come back through L2 (the wave's own stores: wait, then bypass L1)
            __builtin_amdgcn_s_waitcnt(0x0f70);                  // vmcnt(0)
            for (int i = tid; i < n_ovf; i += 64) {
                const uint32_t rec = __hip_atomic_load(&ovf[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                score[(rec >> 16) * PITCH + ((rec >> 8) & 255)] = (uint8_t)rec;
            }
        }
        // ---- NMS.  Usual case: every corner of the item is still in the list (score coordinates).
        if (n_ovf == 0 && !force_scan_b) {
            npx = n_done;
            nms_and_emit();
        } else {
            // ---- scan B (the list overflowed and corners were spilled): find the corners again as the non-zero bytes of the score tile
            FR_FENCE();
            npx = 0;                                             // the list is scratch from here on (its corners are in the tile)
            const int nblock = (cur.ih + BR * RS - 1) / (BR * RS);
            for (int b = 0; b < nblock; b++) {
                const int r0 = 1 + BR * (RS * b + sub);
                uint32_t M = 0;
#pragma unroll
                for (int r = 0; r < BR; r++) {
                    const uint32_t S = (r0 + r <= cur.ih) ? score32[(r0 + r) * PD + col] : 0u;
                    const uint32_t nib = ((S & 0xFFu) ? 1u : 0u) | ((S & 0xFF00u) ? 2u : 0u) | ((S & 0xFF0000u) ? 4u : 0u) | ((S & 0xFF000000u) ? 8u : 0u);
                    M |= nib << (4 * r);
                }
                emit_mask(M, r0, true, b);
            }
            if (npx > 0) nms_and_emit();
        }
        FR_FENCE();
        FR_T(t5);
        FR_ACC(6, t4, t5);
        // ---- the item's count (kept at its first cell's entry); reset the counter
        if (tid == 0) cnt_out[0] = (int32_t)n_emit;
        FR_FENCE();
        if (tid < FR_MAXG) cellcnt[tid] = 0;
#ifdef HS_FAST_PROFILE
        FR_FENCE();
        FR_T(t6);
        FR_ACC(7, t5, t6);
        fr_acc[8] += 1;
        fr_acc[3] += n_ovf > 0;
#endif
#if defined(HS_FAST_PROFILE) || defined(HS_FAST_WAVES)
        {
            const unsigned long long now = __builtin_amdgcn_s_memrealtime();
            if (fr_items == 0) { fr_first = now;
#if defined(HS_FAST_WAVES)
                fr_ph[0] = ((unsigned long long)(unsigned)cur.level << 32) | ((unsigned long long)(fr_codes & 0xFFFFu) << 16) | (unsigned)(n_done & 0xFFFF); fr_ph[5] = now;
#endif
            }
            if (now - fr_prev > fr_long) { fr_long = now - fr_prev; fr_long_w = (unsigned long long)(unsigned)w | ((unsigned long long)(n_ovf > 0) << 32); }
            fr_prev = now; fr_items++;
        }
#endif
        w = w_next;
    }
#ifdef HS_FAST_PROFILE
    FR_T(t_kernel1);
    fr_acc[9] = t_kernel1 - t_kernel0;
    if (tid == 0) for (int i = 0; i < 10; i++) atomicAdd(&g_fr_prof[i], fr_acc[i]);
    if (tid == 0) atomicAdd(&g_fr_prof[10], 1ull);
#endif
#if defined(HS_FAST_PROFILE) || defined(HS_FAST_WAVES)
    if (tid == 0 && blockIdx.x < 4096) {
        unsigned long long* o = &g_fr_wave[blockIdx.x * 16];
        o[0] = fr_real0; o[1] = fr_first; o[2] = __builtin_amdgcn_s_memrealtime(); o[3] = fr_items; o[4] = fr_long; o[5] = fr_long_w;
#if defined(HS_FAST_WAVES)
        for (int i = 0; i < 6; i++) o[8 + i] = fr_ph[i];
#endif
    }
#endif
}

// Cells per work item for a level: as many as fit the tile (<= FR_MAXG), spread evenly over the row.  A tile row holds 4*COLS bytes starting at the
// item's first column rounded DOWN to a dword (a0 = iniX & ~3: the staging loads are dword-aligned), so an item of n cells fits when
//     (iniX & 3) + n * wcell + 6 <= 4*COLS        for every item of the row (iniX = HS_BORDER + first cell * wcell)
// and its corners fit the score tile's 8-bit column (one spare column per cell: column = 1 + px + cell <= 255).  Round 5: the test is made with
// the items' REAL offsets instead of the worst one (3): the standard geometry (cells of 31 px) takes 8 cells per item instead of 7 — 8 * 31 is
// a multiple of 4, every item starts on a dword — which is 12 % fewer items per frame (861 instead of 975 at 1080p / 1.2), each scanning the
// same 256 columns: the lanes of a scan block that hold interior pixels go from 86 % to 97 %.
// Tile width: LC = 6 (64 dwords: the throughput shape) or LC = 5 (32 dwords, two half-waves on different rows).  A handle keeps BOTH item lists;
// the launcher picks per call (hs_api.hip: narrow items for small batches, where the launch lasts as long as its slowest wave and twice as many,
// half as long items are what shortens it).
int hs_fast_max_cell_w(int lc) { return 4 * (1 << lc) - 9; }      // a single cell at the worst offset
int hs_fast_group_cells(int wcell, int ncols, int lc)
{
    if (wcell <= 0 || ncols <= 0) return 0;
    const int tile_w = 4 * (1 << lc);
    auto fits = [&](int g) {
        if (g * wcell + g > 256) return false;                                   // score tile column 1 + px + cell of the last corner
        for (int j0 = 0; j0 < ncols; j0 += g) {
            const int n = std::min(g, ncols - j0);
            if (((HS_BORDER + j0 * wcell) & 3) + n * wcell + 6 > tile_w) return false;
        }
        return true;
    };
    const int gcap = std::max(1, std::min(FR_MAXG, (tile_w - 6) / wcell));
    for (int ngroups = (ncols + gcap - 1) / gcap; ngroups <= ncols; ngroups++) {
        const int g = (ncols + ngroups - 1) / ngroups;                              // even spread over `ngroups` items
        if (fits(g)) return g;
    }
    return 1;                                                                       // configure() rejects wcell > hs_fast_max_cell_w(6) and builds no narrow list for wcell > hs_fast_max_cell_w(5)
}

// what k_fast_rows assumes of an item: its tile row (offset + interior + 6 px of apron) inside the 4 << lc bytes of an LDS row, at most FR_MAXG cells, the last
// corner's score-tile column (1 + px + cell) in 8 bits, at most 64 dwords per row.  Checked for every item when a geometry is configured (hs_api.hip).
bool hs_fast_item_fits(const HsFastItem& it, int lc)
{
    if (it.th == 0) return true;                                                 // yields nothing; staged from (0, 0)
    const int tile_w = 4 << lc;
    return it.off < 4 && (int)it.off + (int)it.iw + 6 <= tile_w && (int)it.ndw * 4 <= tile_w && it.ncell >= 1 && it.ncell <= FR_MAXG && (int)it.iw + (int)it.ncell <= 256;
}

void hs_fast_build_items(const HsLevel* h_lv, int nlevels, HsFastItem* out)
{
    for (int l = 0; l < nlevels; l++) {
        const HsLevel& L = h_lv[l];
        for (int ci = 0; ci < L.nrows; ci++)
            for (int gj = 0; gj < L.ngroups; gj++) {
                HsFastItem& it = out[L.item_begin + ci * L.ngroups + gj];
                memset(&it, 0, sizeof(it));
                const int j0 = gj * L.grp_cells, ncell = std::min(L.grp_cells, L.ncols - j0);
                const int xoff = j0 * L.wcell, yoff = ci * L.hcell;
                const int iniX = HS_BORDER + xoff, iniY = HS_BORDER + yoff;
                const int maxX = std::min(iniX + ncell * L.wcell + 6, L.w - HS_BORDER), maxY = std::min(iniY + L.hcell + 6, L.h - HS_BORDER);
                const int tw = maxX - iniX, th = maxY - iniY;
                const bool valid = tw >= 7 && th >= 7;            // reference skip rules (:435,444) / cv::FAST on < 7 rows or columns
                const int a0 = iniX & ~3;
                it.base = l == 0 ? nullptr : L.base; it.img_stride = L.img_stride; it.pitch = L.pitch;
                it.c0 = ci * L.ncols + j0; it.gcell0 = L.cell_begin + it.c0;
                it.ccap = hs_cell_cap(L.wcell, L.hcell);
                it.slot0 = (uint32_t)(L.cand_off + (uint64_t)it.c0 * it.ccap);
                it.inv_w = L.inv_wcell; it.inv_w1 = L.inv_wcell1;
                it.iniY = valid ? (uint16_t)iniY : 0; it.a0 = valid ? (uint16_t)a0 : 0;   // invalid items still prefetch (harmlessly) from (0,0)
                it.th = valid ? (uint16_t)th : 0; it.iw = valid ? (uint16_t)(tw - 6) : 0;
                it.off = (uint8_t)(iniX - a0); it.ndw = valid ? (uint8_t)((iniX - a0 + tw + 3) >> 2) : 1;
                it.ncell = (uint8_t)ncell; it.level = (uint8_t)l;
                it.xoff = (uint16_t)xoff; it.yoff = (uint16_t)yoff;
            }
    }
}

// test / tuning knobs: read from the environment ONCE per handle (hs_orb_create), never on the launch path
HsFastKnobs hs_fast_read_knobs()
{
    HsFastKnobs k{};
    if (const char* e = getenv("HS_FAST_PCAP")) k.pcap = atoi(e);                         // tuning: cap on the list length
    if (const char* e = getenv("HS_FAST_LIST_MIN")) k.list_min = atoi(e);                 // tuning: list entries the tallest item must keep (decides the workgroups per CU; 1024 = round 4's layout)
    if (const char* e = getenv("HS_FAST_TEST_SMALL_LISTS")) k.small_lists = atoi(e) != 0; // parity tests: force the spill paths
    if (const char* e = getenv("HS_FAST_WG_PER_CU")) k.wg_per_cu = atoi(e);               // tuning: workgroups per CU
    if (const char* e = getenv("HS_FAST_TEST_SCAN_B")) k.force_scan_b = atoi(e) != 0;     // parity tests: NMS from the score tile
    if (const char* e = getenv("HS_FAST_IMAGE_MAJOR")) k.image_major = atoi(e) != 0;      // tuning / parity tests: the work units image-major whatever the batch
    if (const char* e = getenv("HS_FAST_NQ")) k.nq = atoi(e);                             // tuning / parity tests: at most this many work queues (8, 16, 32)
    if (const char* e = getenv("HS_FAST_COLS")) k.cols = atoi(e);                         // tuning / parity tests: 32 / 64 = narrow / wide items whatever the batch (default: by batch)
    if (const char* e = getenv("HS_FAST_NARROW_MAX")) k.narrow_max = atoi(e);             // tuning: narrow items for launches of at most this many (narrow) work items
    if (const char* e = getenv("HS_FAST_NO_FOLD")) k.no_fold = atoi(e) != 0;              // tuning / parity tests: work queues even for launches of <= 2 units per workgroup
    return k;
}

// launch configuration shared by the launcher and by hs_fast_overflow_bytes()
struct FastRowsCfg { int lc, tr, per_cu; uint32_t ovf_stride; FastRowsLds lds; };
static FastRowsCfg fast_rows_cfg(int max_hcell, const HsFastKnobs& knobs, int lc)
{
    FastRowsCfg c;
    c.lc = lc;
    const int cols = 1 << c.lc, pitch = 4 * cols + FR_PAD;
    // Tile rows = the tallest tile of the launch (template instances below).
    const int th_max = max_hcell + 6;
    c.tr = th_max <= 38 ? 38 : th_max <= 40 ? 40 : th_max <= 44 ? 44 : th_max <= 54 ? 54 : th_max <= 70 ? 70 : th_max <= 102 ? 102 : 134;
    FastRowsLds& L = c.lds;
    // LDS is granted in 1280-byte granules on gfx950 (160 KB / 128), and a persistent grid sized for more workgroups per CU than really fit runs its
    // surplus in a second round (measured: +25 % kernel time).  The kernel is latency-bound — measured at 6 / 8 / 11 workgroups per CU (wide items, 128
    // frames): 0.801 / 0.661 / 0.546 ms = 0.240 + 3.37 / n — so a workgroup more per CU is worth more than a long list as long as the list of the
    // TALLEST item keeps `min_entries` (a list of 624 instead of 1056 entries cost 4 % at equal occupancy, one of 800 1.6 %: shorter items, whose
    // list starts where their tile ends, keep 800-900).  Wide tiles of 40 rows: 10 granules = 12 800 bytes = 12 per CU (round 4: 14 080 = 11).
    constexpr int GRAN = 1280, LDS_CU = 160 * 1024;
    const int cnt_bytes = 4 * FR_MAXG;
    const int rs = 64 / cols, blk_rows = 8 * rs;
    const int over_rows = blk_rows * ((c.tr - 6 + blk_rows - 1) / blk_rows) + 6;      // the last scan block of a lane reads whole blocks: up to this many tile rows (the extra ones are masked out)
    // floor: one row step of a scan block (64 lanes x 4 pixels, whatever the tile width) must fit an empty list — the scored corners leave
    // their scores in `pscore` while the rest of the list is still being read, so the list may never run past its end
    const int min_entries = std::max(256, knobs.list_min > 0 ? knobs.list_min : 608);
    const int floor_bytes = std::max(c.tr * pitch + 3 * (min_entries + 16) + cnt_bytes, over_rows * pitch);
    int granules = (floor_bytes + GRAN - 1) / GRAN;
    while (LDS_CU / (granules * GRAN) > 16 && LDS_CU / ((granules + 1) * GRAN) >= 16) granules++;      // at most 16 workgroups per CU anyway (narrow tiles: 4 waves per SIMD by registers): the list takes the LDS that would stay unused
    L.total = granules * GRAN;
    L.off_cnt = L.total - cnt_bytes;
    L.pcap_min = ((L.off_cnt - c.tr * pitch) / 3) & ~15;       // list entries of the tallest item (2 bytes position + 1 byte score each)
    L.pcap_max = knobs.small_lists ? 256 : knobs.pcap > 0 ? std::max(256, knobs.pcap & ~15) : (1 << 20);
    c.per_cu = std::max(1, std::min(16, LDS_CU / ((L.total + GRAN - 1) / GRAN * GRAN)));
    if (knobs.wg_per_cu > 0) c.per_cu = std::max(1, std::min(c.per_cu, knobs.wg_per_cu));
    c.ovf_stride = (uint32_t)((4 * cols - 6) * std::max(max_hcell, 1));       // every interior pixel of an item a corner
    return c;
}
// workgroups of a launch over `total_work` items (non-decreasing in total_work)
static int fast_rows_grid(const FastRowsCfg& c, int total_work)
{
    int nblk = (256 * c.per_cu) & ~(HS_FAST_NQ_MAX - 1);         // a multiple of every queue count
    while (nblk >= 2 * HS_FAST_NQ_MAX && (nblk / 2) % HS_FAST_NQ_MAX == 0 && nblk / 2 >= total_work) nblk /= 2;   // tiny jobs: fewer idle workgroups
    return nblk;
}
// queue count and unit order of a launch over `batch` images with `nblk` workgroups (see FastSched)
static FastSched fast_sched_plan(int batch, int items_per_img, const HsFastKnobs& knobs, int nblk)
{
    FastSched S{};
    if (!knobs.no_fold && !knobs.image_major && batch > 0 && (long long)items_per_img * batch <= 2LL * nblk) {
        S.nq_log = 3; S.mode = 4; S.par = batch; S.per_q = items_per_img * batch;
        return S;
    }
    int nq_log = 3;
    const int want = knobs.nq > 0 ? knobs.nq : HS_FAST_NQ_MAX;
    while ((2 << nq_log) <= want) nq_log++;
    const int nq = 1 << nq_log;
    S.nq_log = nq_log;
    if (!knobs.image_major && batch >= nq && batch % nq == 0) { S.mode = 1; S.par = batch / nq; S.per_q = S.par * items_per_img; }
    else if (!knobs.image_major && batch > 0 && nq % batch == 0) {
        S.mode = 2; S.par = nq / batch;
        while ((1 << S.par_log) < S.par) S.par_log++;
        S.per_q = (items_per_img + S.par - 1) / S.par;
    } else if (!knobs.image_major && batch > 0) { S.mode = 3; S.par = batch; S.per_q = (items_per_img * batch + nq - 1) / nq; }
    else { S.mode = 0; S.per_q = (items_per_img * batch + nq - 1) / nq; }
    return S;
}
static FastSched fast_sched(int batch, int items_per_img, const HsFastKnobs& knobs, int nblk)
{
    FastSched S = fast_sched_plan(batch, items_per_img, knobs, nblk);
    auto magic = [](int d) { return d > 1 ? (uint32_t)(((1ull << 32) + (uint64_t)d - 1) / (uint64_t)d) : 0u; };       // ceil(2^32 / d); d <= 1: unused (fast_div)
    S.m_per_q = magic(S.per_q); S.m_par = magic(S.par); S.m_items = magic(items_per_img);
    return S;
}
size_t hs_fast_overflow_bytes(int max_hcell, int total_work_max, const HsFastKnobs& knobs)
{
    size_t spill = 0;                                             // either tile width may be launched on this workspace
    for (int lc = 5; lc <= 6; lc++) {
        const FastRowsCfg c = fast_rows_cfg(max_hcell, knobs, lc);
        spill = std::max(spill, (size_t)2 * fast_rows_grid(c, total_work_max) * c.ovf_stride * 4);      // two launches may be in flight
    }
    return (size_t)4 * HS_FAST_QUEUE_DWORDS * 4 + spill;
}

static bool launch_fast_rows(const HsFastItem* d_items, HsImg0 img0, int batch, int total_cells, int items_per_img, int fast_th,
                             uint2* cand, int32_t* cell_count, uint64_t cand_img_stride,
                             int max_wcell, int max_hcell, uint32_t* overflow, uint32_t epoch, const HsFastKnobs& knobs, int item_first, int spill_slot, int items_all, int lc_in,
                             const HsFastQt* d_qt, uint32_t* qhist, unsigned long long* qbest, uint32_t qhist_img_stride, uint32_t qbest_img_stride, hipStream_t s)
{
    (void)max_wcell;
    const FastRowsCfg c = fast_rows_cfg(max_hcell, knobs, lc_in);
    const FastRowsLds& L = c.lds;
    const int lc = c.lc, tr = c.tr;
    const int total_work = items_per_img * batch;
    if (total_work <= 0) return false;
    const int nblk = fast_rows_grid(c, total_work);
    const int force_scan_b = knobs.force_scan_b;
    const FastSched S = fast_sched(batch, items_per_img, knobs, nblk);
    const uint32_t spill_base = (uint32_t)spill_slot * (uint32_t)fast_rows_grid(c, items_all * batch) * c.ovf_stride;      // the second spill half starts after a full-size first one
#define FR_LAUNCH_K(LC_, TR_, K_) hipLaunchKernelGGL((k_fast_rows<LC_, TR_, K_>), dim3(nblk), dim3(64), L.total, s, d_items, img0, fast_th, cand, \
                                               cell_count, cand_img_stride, total_cells, items_per_img, total_work, L, force_scan_b, overflow, c.ovf_stride, epoch, item_first, spill_base, S, \
                                               d_qt, qhist, qbest, qhist_img_stride, qbest_img_stride)
#define FR_LAUNCH(LC_, TR_) do { if (d_qt != nullptr) FR_LAUNCH_K(LC_, TR_, true); else FR_LAUNCH_K(LC_, TR_, false); } while (0)
    if (lc == 6) { if (tr == 38) FR_LAUNCH(6, 38); else if (tr == 40) FR_LAUNCH(6, 40); else if (tr == 44) FR_LAUNCH(6, 44); else if (tr == 54) FR_LAUNCH(6, 54); else if (tr == 70) FR_LAUNCH(6, 70); else if (tr == 102) FR_LAUNCH(6, 102); else FR_LAUNCH(6, 134); }
    else         { if (tr == 38) FR_LAUNCH(5, 38); else if (tr == 40) FR_LAUNCH(5, 40); else if (tr == 44) FR_LAUNCH(5, 44); else if (tr == 54) FR_LAUNCH(5, 54); else if (tr == 70) FR_LAUNCH(5, 70); else if (tr == 102) FR_LAUNCH(5, 102); else FR_LAUNCH(5, 134); }
#undef FR_LAUNCH
#undef FR_LAUNCH_K
    return true;
}

// returns whether a kernel was enqueued: a launch consumes one work-queue counter set (`epoch` & 1) and zeroes it for the launch after next,
// so the caller advances its epoch only for launches that happened
bool hs_launch_fast(const HsLevel* d_lv, const HsFastItem* d_items, int nlevels, HsImg0 img0, int batch, int total_cells, int items_per_img, int fast_th,
                    uint2* cand, int32_t* cell_count, uint64_t cand_img_stride,
                    int max_wcell, int max_hcell, uint32_t* overflow, uint32_t epoch, const HsFastKnobs& knobs, int item_first, int item_count, int spill_slot, int lc,
                    const HsFastQt* d_qt, uint32_t* qhist, unsigned long long* qbest, uint32_t qhist_img_stride, uint32_t qbest_img_stride, hipStream_t s)
{
    (void)d_lv; (void)nlevels;
    if (total_cells <= 0 || item_count <= 0) return false;
    return launch_fast_rows(d_items, img0, batch, total_cells, item_count, fast_th, cand, cell_count, cand_img_stride, max_wcell, max_hcell, overflow, epoch, knobs,
                            item_first, spill_slot, items_per_img, lc, d_qt, qhist, qbest, qhist_img_stride, qbest_img_stride, s);
}
