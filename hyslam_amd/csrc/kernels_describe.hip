// kernels_describe.hip — K5+K6+K7 fused: 7x7 Gaussian blur (patch-local), intensity-centroid orientation,
// 256-bit rotated BRIEF descriptor, final keypoint record.  One wavefront per keypoint.
//
// Replaces, per level (src/features/ORBExtractor.cpp:535-555):
//     workingMat = level.clone(); GaussianBlur(workingMat, workingMat, Size(7,7), 2, 2, BORDER_REFLECT_101);
//     feature_finder->compute(workingMat, keypoints, desc);   // ORBFinder.cpp:70-129
//     keypoint->pt *= scale
// The reference blurs whole levels (6.4 Mpx read + write per 1080p frame) although only the 37x37
// neighbourhood of ~2000 keypoints is ever sampled.  The blur is a pure function of the 43x43 raw
// neighbourhood (reflect-101 at the level border), so it is evaluated per keypoint in LDS: 1849 B read per
// keypoint instead of 2*P/K = 6419 B, and no blurred image ever goes to HBM.  Results are identical.
//
// Arithmetic:
//   blur   : unsigned 8.8 fixed-point taps, 16-bit saturating row sums, 32-bit column sums, (acc+0x8000)>>16
//   angle  : integer moments over the 749-px disc (umax table), cv::fastAtan2 polynomial in fp32, no FMA
//   rBRIEF : a=(float)cos(theta), b=(float)sin(theta) evaluated in double; offsets = round-half-even of
//            separately rounded fp32 products; bit i of byte j = test 8j+i -> one ballot per 64 tests.
//
// Fast path (FT = true: every tap <= 255 and 255*sum(taps) <= 65535, i.e. the row sums cannot saturate — true for the
// default taps): the patch is fetched with unaligned 16-byte loads straight into an LDS tile aligned to the patch; the row pass makes four
// adjacent outputs from three dwords with TEN v_dot4_u32_u8 against shifted tap words (the taps are shifted, not the data) and stores the
// sums of a row PAIR transposed, one dword per column; the column pass makes 8 vertically adjacent outputs from 7 dwords with four
// v_dot2_u32_u16 each (odd rows use a tap packing shifted by one element) and writes them with one 8-byte store into a COLUMN-major
// blurred tile.  Integer sums are exact, so this is bit-identical to the generic path, which stays for exotic taps; keypoints whose patch
// crosses the level border fetch their bytes one by one with BORDER_REFLECT_101.
// The kernel is bound by VALU issue (0.80 instructions per busy CU cycle) with the LDS pipe second (46 % busy): what is here is what
// survived counting instructions per keypoint (915 -> 566) and LDS instructions (118 -> 50); DESIGN.md §4 lists what was tried.
#include "hs_internal.h"
#define HS_HD __host__ __device__
#include "lean_sincos.h"
#include "../../include/hyslam_orb_pattern.h"

// the 256 test pairs as floats, 16 bytes per test: one 16-byte load per lane and round, no byte extraction / conversion
struct alignas(16) PatternF { float v[HS_ORB_PATTERN_INTS]; };
static constexpr PatternF make_pattern_f()
{
    constexpr int8_t src[HS_ORB_PATTERN_INTS] = HS_ORB_PATTERN_INIT;
    PatternF t{};
    // stored as (x0, x1, y0, y1): the two points' x and y are then register PAIRS as loaded, ready for v_pk_mul_f32 (no v_mov shuffles per round)
    for (int i = 0; i < HS_ORB_PATTERN_INTS; i += 4) { t.v[i] = (float)src[i]; t.v[i + 1] = (float)src[i + 2]; t.v[i + 2] = (float)src[i + 1]; t.v[i + 3] = (float)src[i + 3]; }
    return t;
}
__constant__ PatternF c_pattern = make_pattern_f();

// intensity-centroid weights per aligned dword of the blurred tile (ORBFinder.cpp:131-149): rows v = -15..15 of the tile (row 18 + v),
// dwords q = 0..9 (columns 4q..4q+3, u = column - 18): .x bytes = u + 15, .y bytes = 1 inside the 749-pixel disc, 0 outside.
// A compile-time table in constant memory: every workgroup used to rebuild it in LDS (70 instructions per keypoint, 2.4 KB per workgroup).
struct MomW { uint2 w[31 * 10]; };
static constexpr MomW make_momw()
{
    constexpr int umax[16] = { 15, 15, 15, 15, 14, 14, 14, 13, 13, 12, 11, 10, 9, 8, 6, 3 };
    MomW t{};
    for (int i = 0; i < 31 * 10; i++) {
        const int r = i / 10, q = i - r * 10;
        const int av = r < 15 ? 15 - r : r - 15, d = umax[av];
        unsigned wu = 0, w1 = 0;
        for (int j = 0; j < 4; j++) {
            const int uu = 4 * q + j - 18;
            if ((uu < 0 ? -uu : uu) <= d) { wu |= (unsigned)(uu + 15) << (8 * j); w1 |= 1u << (8 * j); }
        }
        t.w[i].x = wu; t.w[i].y = w1;
    }
    return t;
}
__constant__ MomW c_momw = make_momw();

#define RAW_N 43
#define RAW_P 48                 // raw tile pitch: 12 dwords hold 43 bytes at any source misalignment (3 + 43 <= 48)
#define RAW_BYTES (RAW_N * RAW_P + 16)
#define BL_N 37
#define H_P 38                   // generic path: row-major u16 row sums
#ifndef HT_P
#define HT_P 46
#endif                           // fast path: transposed u16 row sums, 43 rows + pad; 46 u16 = 23 dwords (odd) keeps column-strided stores off the same banks
#define H_ELEMS (40 * HT_P)       // fast path: 40 columns are written (37 used); 1840 >= RAW_N * H_P = 1634
#ifndef BL_P
#define BL_P 40
#endif                           // blurred tile: COLUMN-major, 40 bytes per column (8-byte aligned columns for the 8-byte stores of the column pass)
#define KP_PER_BLOCK 4

__device__ __forceinline__ int reflect101(int p, int len)
{
    if (p < 0) p = -p;
    if (p >= len) p = 2 * (len - 1) - p;
    return p < 0 ? 0 : p;
}

__device__ __forceinline__ float fast_atan2_deg(float y, float x)
{
    // cv::fastAtan2 (OpenCV 3.4 mathfuncs_core atan_f32); every operation individually rounded
    const float p1 = 0.9997878412794807f * (float)(180 / 3.1415926535897932384626433832795);
    const float p3 = -0.3258083974640975f * (float)(180 / 3.1415926535897932384626433832795);
    const float p5 = 0.1555786518463281f * (float)(180 / 3.1415926535897932384626433832795);
    const float p7 = -0.04432655554792128f * (float)(180 / 3.1415926535897932384626433832795);
    float ax = fabsf(x), ay = fabsf(y);
    // both branches of the reference divide the smaller magnitude by (the larger + eps) and run the same polynomial: one division, one
    // polynomial, and the branch only decides between p and 90 - p
    const bool xmajor = ax >= ay;
    const float c = __fdiv_rn(xmajor ? ay : ax, __fadd_rn(xmajor ? ax : ay, (float)2.2204460492503131e-16));
    const float c2 = __fmul_rn(c, c);
    const float p = __fmul_rn(__fadd_rn(__fmul_rn(__fadd_rn(__fmul_rn(__fadd_rn(__fmul_rn(p7, c2), p5), c2), p3), c2), p1), c);
    float a = xmajor ? p : __fsub_rn(90.f, p);
    if (x < 0) a = __fsub_rn(180.f, a);
    if (y < 0) a = __fsub_rn(360.f, a);
    return a;
}

typedef unsigned short hs_ushort2 __attribute__((ext_vector_type(2)));
typedef uint32_t hs_u32x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ uint32_t udot2(uint32_t a, uint32_t b, uint32_t c)      // v_dot2_u32_u16: a.lo*b.lo + a.hi*b.hi + c
{
    return __builtin_amdgcn_udot2(__builtin_bit_cast(hs_ushort2, a), __builtin_bit_cast(hs_ushort2, b), c, false);
}

__device__ __forceinline__ uint32_t udot2c(uint32_t a, uint32_t b, uint32_t c)     // the same with the clamp bit: the sum saturates at 0xFFFFFFFF
{
    return __builtin_amdgcn_udot2(__builtin_bit_cast(hs_ushort2, a), __builtin_bit_cast(hs_ushort2, b), c, true);
}

#ifdef HS_DESC_PROFILE      // make EXTRA=-DHS_DESC_PROFILE: cycle stamps per phase and wave (tools/describe_phase_profile.py)
#define DP_WAVES (1 << 17)
__device__ unsigned int g_desc_prof[DP_WAVES * 8];      // a slot of 8 deltas per wave: plain stores (atomics on a few hot addresses would BE the profile)
#define DP_T(var) const unsigned long long var = __builtin_amdgcn_s_memtime()
#define DP_ACC(i, a, b) do { if (lane == 0 && dp_slot < DP_WAVES) g_desc_prof[dp_slot * 8 + (i)] = (unsigned int)((b) - (a)); } while (0)
extern "C" void hs_debug_describe_profile(unsigned int* out, int waves)      // out[waves * 8]; clears the buffer
{
    (void)hipDeviceSynchronize();
    (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_desc_prof), sizeof(unsigned int) * 8 * (size_t)(waves < DP_WAVES ? waves : DP_WAVES));
    void* p = nullptr;
    (void)hipGetSymbolAddress(&p, HIP_SYMBOL(g_desc_prof));
    (void)hipMemset(p, 0, sizeof(unsigned int) * 8 * DP_WAVES);
}
#else
#define DP_T(var)
#define DP_ACC(i, a, b)
#endif
// sum over the 64 lanes, the same value in every lane's result (DPP row shifts + row broadcasts, then lane 63)
__device__ __forceinline__ int wave_sum(int x)
{
    x += __builtin_amdgcn_update_dpp(0, x, 0x111, 0xF, 0xF, true);      // row_shr:1
    x += __builtin_amdgcn_update_dpp(0, x, 0x112, 0xF, 0xF, true);      // row_shr:2
    x += __builtin_amdgcn_update_dpp(0, x, 0x114, 0xF, 0xF, true);      // row_shr:4
    x += __builtin_amdgcn_update_dpp(0, x, 0x118, 0xF, 0xF, true);      // row_shr:8   -> lane 15 of each row holds the row's sum
    x += __builtin_amdgcn_update_dpp(0, x, 0x142, 0xA, 0xF, false);     // row_bcast:15 into rows 1 and 3
    x += __builtin_amdgcn_update_dpp(0, x, 0x143, 0xC, 0xF, false);     // row_bcast:31 into rows 2 and 3
    return __builtin_amdgcn_readlane(x, 63);
}
#define WAVE_LDS_SYNC() do { __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_s_waitcnt(0xc07f); } while (0)   // lgkmcnt(0)

// The extra workgroup of a RIGHT image in the stereo front end's describe launch (HsStripFuse): bins the image's selected keypoints into the
// 32-row strips the stereo matcher scans (k_stereo_strips' job, kernels_stereo.hip; Stereomatcher.cpp:46-63).  Everything an entry needs is known
// since the quadtree kernel: position (scaled to level 0 with the same single fp32 multiplication the keypoint record gets, ORBExtractor.cpp:546-552),
// level, size and the keypoint's slot in the output list (levels concatenated, list order).  Eight keypoints per thread in flight.
__device__ __forceinline__ void describe_strips_block(const HsLevel* __restrict__ lv, int nlevels, const uint32_t* __restrict__ sel_xys,
                                                      const int32_t* __restrict__ sel_count, int sel_img_stride, int img, int pair, int cap,
                                                      const HsStripFuse& SF, int* s_cnt)
{
    const int tid = threadIdx.x;
    constexpr int T = 64 * KP_PER_BLOCK, KPT = 8;
    for (int s = tid; s < SF.n_strips; s += T) s_cnt[s] = 0;
    // per-level counts, list offsets, scales and sizes: wave-uniform, fetched once (scalar loads), selected per keypoint with compare chains
    int cnt[HS_MAX_LEVELS], soff[HS_MAX_LEVELS], total = 0; float lsc[HS_MAX_LEVELS], lsz[HS_MAX_LEVELS];
#pragma unroll
    for (int l = 0; l < HS_MAX_LEVELS; l++) {
        const bool on = l < nlevels;
        cnt[l] = on ? hs_cload<int32_t>(sel_count + img * nlevels + l) : 0;
        soff[l] = on ? lv[l].sel_off : 0; lsc[l] = on ? lv[l].scale : 1.f; lsz[l] = on ? lv[l].kp_size : 0.f;
        total += cnt[l];
    }
    total = min(total, cap);
    __syncthreads();
    for (int i0 = 0; i0 < total; i0 += T * KPT) {
        uint32_t cx[KPT], cy[KPT]; int lvl[KPT];
#pragma unroll
        for (int k = 0; k < KPT; k++) {
            const int g = min(i0 + tid + T * k, total - 1);
            int l = 0, adj = soff[0], acc = 0;                  // entry of list slot g: adj + g, adj = the level's offset minus the slots of the levels before it
#pragma unroll
            for (int q = 0; q + 1 < HS_MAX_LEVELS; q++) { acc += cnt[q]; if (g >= acc) { l = q + 1; adj = soff[q + 1] - acc; } }
            const uint32_t* sel = sel_xys + ((size_t)img * sel_img_stride + (size_t)(adj + g)) * 3;
            cx[k] = hs_gload<uint32_t>(sel); cy[k] = hs_gload<uint32_t>(sel + 1); lvl[k] = l;
        }
#pragma unroll
        for (int k = 0; k < KPT; k++) {
            const int g = i0 + tid + T * k;
            if (g >= total) continue;
            const int l = lvl[k];
            float sc = lsc[0], ksz = lsz[0];
#pragma unroll
            for (int q = 1; q < HS_MAX_LEVELS; q++) if (l == q) { sc = lsc[q]; ksz = lsz[q]; }
            const float kx = l ? __fmul_rn((float)(int)cx[k], sc) : (float)(int)cx[k];
            const float ky = l ? __fmul_rn((float)(int)cy[k], sc) : (float)(int)cy[k];
            const float r = 2.0f * ksz / SF.size_ref;         // Stereomatcher.cpp:56
            int maxr = (int)ceilf(ky + r), minr = (int)floorf(ky - r);
            if (maxr < 0 || minr >= SF.n_rows) continue;      // rows outside [0, nRows) do not exist (D2)
            minr = max(minr, 0); maxr = min(maxr, SF.n_rows - 1);
            for (int s = minr >> HS_STRIP_SHIFT; s <= (maxr >> HS_STRIP_SHIFT); s++) {
                const int slot = atomicAdd(&s_cnt[s], 1);
                const int lo = max(minr - (s << HS_STRIP_SHIFT), 0), hi = min(maxr - (s << HS_STRIP_SHIFT), 31);
                HsStripEntry e; e.uR = kx; e.octave = l; e.idx_band = (uint32_t)g | ((uint32_t)lo << 16) | ((uint32_t)hi << 24); e._pad = 0;
                SF.strip_list[((size_t)pair * SF.n_strips + s) * cap + slot] = e;
            }
        }
    }
    __syncthreads();
    for (int s = tid; s < SF.n_strips; s += T) SF.strip_count[(size_t)pair * SF.n_strips + s] = s_cnt[s];
}

template <bool FT>
__global__ __launch_bounds__(64 * KP_PER_BLOCK) void k_describe(const HsLevel* __restrict__ lv, int nlevels, HsImg0 img0,
                                                                 const uint32_t* __restrict__ sel_xys, const int32_t* __restrict__ sel_count,
                                                                 const uint16_t* __restrict__ sel_perm, int sel_img_stride, const uint16_t* __restrict__ taps7,
                                                                 HsOut O, HsStripFuse SF)
{
    __shared__ __attribute__((aligned(16))) uint8_t s_raw[KP_PER_BLOCK][RAW_BYTES];
    __shared__ __attribute__((aligned(16))) uint16_t s_h[KP_PER_BLOCK][H_ELEMS];
    static_assert(sizeof(s_raw) >= HS_STRIPS_MAX * 4, "the strip counters of the extra workgroup live in the raw tiles");
    const int gx = (int)gridDim.x - (SF.enabled ? 1 : 0);    // describe workgroups per image
    if (SF.enabled && (int)blockIdx.x == gx) {               // the extra workgroup (uniform per workgroup)
        if ((int)blockIdx.y >= O.split) describe_strips_block(lv, nlevels, sel_xys, sel_count, sel_img_stride, blockIdx.y, (int)blockIdx.y - O.split, O.cap, SF, reinterpret_cast<int*>(&s_raw[0][0]));
        return;
    }
    DP_T(dp0);
#ifdef HS_DESC_PROFILE
    const int dp_slot = (blockIdx.y * gridDim.x + blockIdx.x) * KP_PER_BLOCK + (threadIdx.x >> 6);
#endif
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // wave index: uniform, so everything derived from the
    const int img = blockIdx.y;                                                                // keypoint record stays in SGPRs / scalar loads
    // Which keypoint: the image's keypoints are walked in SPATIAL order (level, 64-px tile; sel_perm from the quadtree kernel), and the
    // block -> position map gives each residue class of blockIdx.x % 8 — the blocks that share an XCD under round-robin placement — one
    // contiguous eighth of that order, so patches that overlap are fetched into one L2 at about the same time (k_describe used to read
    // 3.9x its compulsory bytes).  Results still go to the keypoint's list-order slot.  Placement only affects speed.
    const int cls = blockIdx.x & 7, jcls = blockIdx.x >> 3;
    const int before = cls * (gx >> 3) + min(cls, gx & 7);   // blocks in the classes below `cls`: class c has (gx >> 3) + (c < (gx & 7)) of them
    const int gs = (before + jcls) * KP_PER_BLOCK + wv;    // position in the spatial order (levels concatenated)

    // locate the level: prefix over the per-level selection counts (wave-uniform scalar loop)
    // With 8 levels (the reference's configuration) the counts arrive as ONE 8-dword scalar load and the search is branch-free — level = number of
    // inclusive prefix sums <= gs — instead of a chain of 8 dependent loads: the header was a quarter of a wave's life.
    int level = -1, first = 0, total = 0;
    if (nlevels == 8) {
        const hs_u32x8 c8 = hs_cload<hs_u32x8>(sel_count + img * 8);
        int nle = 0;
#pragma unroll
        for (int l = 0; l < 8; l++) {
            total += (int)c8[l];
            const int m = (total - 1 - gs) >> 31;          // -1 where gs >= total (integer form: a bool here comes back as a lane mask + v_cndmask)
            nle -= m; first += (int)c8[l] & m;             // the prefix sums rise: the counts of the levels wholly below gs add up to `first`
        }
        level = nle < 8 ? nle : -1;
    } else {
        for (int l = 0; l < nlevels; l++) {
            int c = hs_cload<int32_t>(sel_count + img * nlevels + l);
            if (level < 0 && gs < total + c) { level = l; first = total; }
            total += c;
        }
    }
    const int cap = O.cap;
    const bool second = img >= O.split;
    const int oimg = second ? img - O.split : img;
    hs_keypoint* kps = second ? O.kps2 : O.kps;
    uint8_t* desc = second ? O.desc2 : O.desc;
    if (blockIdx.x == 0 && threadIdx.x == 0) (second ? O.n2 : O.n)[oimg] = min(total, cap);
    if (level < 0) return;                                  // wave-uniform
    const HsLevel& L = lv[level];
    const int li = (int)hs_gload<uint16_t>(sel_perm + (size_t)img * sel_img_stride + L.sel_off + (gs - first));      // index in the level's list
    const int g = first + __builtin_amdgcn_readfirstlane(li);                                                          // list-order slot of the keypoint
    if (g >= cap) return;
    const uint32_t* sel = sel_xys + ((size_t)img * sel_img_stride + L.sel_off + (g - first)) * 3;
    const int cx = (int)hs_cload<uint32_t>(sel), cy = (int)hs_cload<uint32_t>(sel + 1);
    const int score = (int)hs_cload<uint32_t>(sel + 2);

    const uint8_t* base; size_t pitch;
    if (level == 0) { base = hs_img0_ptr(img0, img); pitch = img0.row_stride; }
    else { base = L.base + (size_t)img * L.img_stride; pitch = L.pitch; }

    uint8_t* raw = s_raw[wv]; uint16_t* hb = s_h[wv];
    uint8_t* bl = s_raw[wv];      // the blurred 37x37 tile reuses the raw tile's LDS: the raw bytes are dead once the row pass has run
    uint32_t tp[7];                                         // seven 16-bit taps in four dwords (the buffer holds 16 bytes): scalar loads
#pragma unroll
    for (int k = 0; k < 7; k++) { const uint32_t w = hs_cload<uint32_t>(reinterpret_cast<const uint8_t*>(taps7) + 4 * (k >> 1)); tp[k] = (k & 1) ? (w >> 16) : (w & 0xFFFFu); }

    DP_T(dp1);
    DP_ACC(0, dp0, dp1);
    // ---- raw 43x43 neighbourhood.  Interior keypoints: aligned dword rows, the tile keeps the source misalignment `sh`.
    //      Patches that touch the level border: byte loads with BORDER_REFLECT_101.
    const int x0 = cx - 21, y0 = cy - 21;
    const bool interior = x0 >= 0 && y0 >= 0 && cy + 21 < L.h && cx + 27 < L.w;     // 48 bytes are fetched per row: x0 + 47 stays inside it
    if (FT && interior) {
        // three 16-byte loads per row starting at the patch's own first byte: gfx9 global loads take unaligned addresses, so the LDS tile
        // is aligned to the patch and the row pass below needs no per-keypoint byte shifts
        struct __attribute__((packed, aligned(1))) U128 { hs_u32x4 v; };
        typedef const HS_GLOBAL U128* gu128;
        // lane = (row r0 of 21, 16-byte piece q of 3); rows r0, r0 + 21, r0 + 42: the per-iteration address arithmetic is one add
        const int r0 = lane / 3, q = lane - r0 * 3;
        const uint8_t* src = base + (size_t)(y0 + r0) * pitch + x0 + 16 * q;
        uint8_t* dst = &raw[r0 * RAW_P + 16 * q];
        if (lane < 63) {
#pragma unroll
            for (int k = 0; k < 3; k++)
                if (r0 + 21 * k < RAW_N)
                    *reinterpret_cast<hs_u32x4*>(dst + 21 * k * RAW_P) = ((gu128)(uintptr_t)(src + (size_t)(21 * k) * pitch))->v;
        }
    } else {
        for (int i = lane; i < RAW_N * RAW_N; i += 64) {
            int r = i / RAW_N, q = i - r * RAW_N;
            int y = reflect101(y0 + r, L.h), x = reflect101(x0 + q, L.w);
            raw[r * RAW_P + q] = hs_gload<uint8_t>(base + (size_t)y * pitch + x);
        }
    }
    WAVE_LDS_SYNC();
    DP_T(dp2);
    DP_ACC(1, dp1, dp2);

    if (FT) {
        // ---- row pass: H[r][c..c+3] from three/four aligned dwords, two dot4 per output; stored transposed HT[c][r]
        // four adjacent outputs from the dwords d0 (bytes 0..3), d1 (4..7), d2 (8..11) of the row: output j = sum_k t[k] * byte[j + k].  The
        // TAPS are shifted, not the data: ten v_dot4 with constant tap words per four outputs instead of six v_alignbyte + eight v_dot4
        auto tw = [&](int k0) -> uint32_t {      // tap word whose byte i is t[k0 + i] (0 outside 0..6)
            uint32_t w = 0;
            for (int i = 0; i < 4; i++) { const int k = k0 + i; if (k >= 0 && k < 7) w |= tp[k] << (8 * i); }
            return w;
        };
        const uint32_t ta0 = tw(0), ta1 = tw(4), tb0 = tw(-1), tb1 = tw(3), tc0 = tw(-2), tc1 = tw(2), tc2 = tw(6), td0 = tw(-3), td1 = tw(1), td2 = tw(5);
        // lane = (row r0 of 6, column group gq of 10), rows r0 + 6k: every address below is the lane's base plus a compile-time offset
        // (a flat index over the 430 (row, group) tasks cost a division and two multiplications per iteration)
        {
            // lane = (row PAIR p0 of 6, column group gq of 10), pairs p0 + 6k: the two rows' sums of a column are neighbours in the
            // transposed store, so they leave as one dword — an LDS store costs 4 cycles whatever its width (address + data transfer), and
            // stores, not reads, fill this kernel's LDS time.  Pair 21 = (row 42, a row that does not exist): its second half lands in the
            // pad element H[c][43], which only ever meets a zero tap.
            const int p0 = lane / 10, gq = lane - p0 * 10;
            const uint32_t* row0 = reinterpret_cast<const uint32_t*>(&raw[2 * p0 * RAW_P + 4 * gq]);
            uint32_t* out0 = reinterpret_cast<uint32_t*>(&hb[4 * gq * HT_P + 2 * p0]);          // columns 37..39 of the last group are written too (storage exists, never read)
            auto hrow = [&](const uint32_t* row, uint32_t (&h)[4]) {
                const uint32_t d0 = row[0], d1 = row[1], d2 = row[2];   // bytes beyond the row only meet zero taps
                h[0] = __builtin_amdgcn_udot4(d1, ta1, __builtin_amdgcn_udot4(d0, ta0, 0u, false), false);
                h[1] = __builtin_amdgcn_udot4(d1, tb1, __builtin_amdgcn_udot4(d0, tb0, 0u, false), false);
                h[2] = __builtin_amdgcn_udot4(d2, tc2, __builtin_amdgcn_udot4(d1, tc1, __builtin_amdgcn_udot4(d0, tc0, 0u, false), false), false);
                h[3] = __builtin_amdgcn_udot4(d2, td2, __builtin_amdgcn_udot4(d1, td1, __builtin_amdgcn_udot4(d0, td0, 0u, false), false), false);
            };
            if (lane < 60) {
                constexpr int NP = (RAW_N + 1) / 2;            // 22 row pairs
#pragma unroll
                for (int k = 0; k < (NP + 5) / 6; k++) {
                    if (6 * k + 5 < NP || p0 + 6 * k < NP) {
                        uint32_t ha[4], hb2[4];
                        hrow(row0 + 12 * k * (RAW_P / 4), ha);
                        hrow(row0 + (12 * k + 1) * (RAW_P / 4), hb2);
#pragma unroll
                        for (int j = 0; j < 4; j++) out0[6 * k + j * (HT_P / 2)] = ha[j] | (hb2[j] << 16);      // sums are < 65536
                    }
                }
            }
        }
        WAVE_LDS_SYNC();
        DP_T(dp3);
        DP_ACC(2, dp2, dp3);
        // ---- column pass: lane = (column c, even row r): dwords (r,r+1)..(r+6,r+7) of HT[c]; even rows use taps (t0,t1)(t2,t3)(t4,t5)(t6,0),
        //      the odd row r+1 uses (0,t0)(t1,t2)(t3,t4)(t5,t6) on the SAME dwords
        //      The taps enter multiplied by 256 (a byte-sized tap * 256 fits 16 bits) and every v_dot2 CLAMPS: the sum is then 256 * (sum + 0x8000),
        //      which leaves the 32-bit range exactly when the blurred value would leave the byte range, so byte 3 of the clamped sum is the
        //      saturated result (0xFFFFFFFF -> 255) and the eight v_min of the unscaled form are gone
        const uint32_t e0 = (tp[0] | (tp[1] << 16)) << 8, e1 = (tp[2] | (tp[3] << 16)) << 8, e2 = (tp[4] | (tp[5] << 16)) << 8, e3 = tp[6] << 8;
        const uint32_t o0 = tp[0] << 24, o1 = (tp[1] | (tp[2] << 16)) << 8, o2 = (tp[3] | (tp[4] << 16)) << 8, o3 = (tp[5] | (tp[6] << 16)) << 8;
        // a lane makes 8 vertically adjacent outputs (rows r..r+7, r = 8*rg) from the seven dwords H[r..r+13] of its column: 185 (column,
        // group) tasks in three rounds
        // lane -> (column c, row group rg) of task i = lane + 64 k: 64 = 12 * 5 + 4, so a step adds (12, 4) with one carry — no division per round
        int c = lane / 5, rg = lane - c * 5;
#pragma unroll
        for (int k = 0; k < (BL_N * 5 + 63) / 64; k++) {
            if (64 * k + 63 < BL_N * 5 || lane + 64 * k < BL_N * 5) {
                const int r = 8 * rg;
                const uint32_t* col = reinterpret_cast<const uint32_t*>(&hb[c * HT_P + r]);
                uint32_t a[7];                                     // rows >= 43 (last group): pad / next column, only the discarded outputs see them
#pragma unroll
                for (int m = 0; m < 7; m++) a[m] = col[m];
                // (sum + 0x8000) >> 16 saturated to 255 = byte 3 of the clamped 256-fold sum; the rounding constant starts the accumulator.  The
                // blurred tile is COLUMN-major (BL(row, col) = bl[col * BL_P + row]), so the lane's eight results are eight consecutive bytes: one
                // 8-byte store instead of eight byte stores
                uint32_t w[8];
#pragma unroll
                for (int m = 0; m < 4; m++) {
                    w[2 * m] = udot2c(a[m + 3], e3, udot2c(a[m + 2], e2, udot2c(a[m + 1], e1, udot2c(a[m], e0, 0x800000u))));
                    w[2 * m + 1] = udot2c(a[m + 3], o3, udot2c(a[m + 2], o2, udot2c(a[m + 1], o1, udot2c(a[m], o0, 0x800000u))));
                }
                // v_perm_b32: byte 3 of each of four values -> one dword (selector bytes: 0-3 = second operand, 4-7 = first, 0x0c = zero)
                const uint32_t lo = __builtin_amdgcn_perm(w[1], w[0], 0x0c0c0703u) | __builtin_amdgcn_perm(w[3], w[2], 0x07030c0cu);
                const uint32_t hi = __builtin_amdgcn_perm(w[5], w[4], 0x0c0c0703u) | __builtin_amdgcn_perm(w[7], w[6], 0x07030c0cu);
                *reinterpret_cast<uint2*>(&bl[c * BL_P + r]) = make_uint2(lo, hi);      // rows 37..39 of the last group: the column's unused tail
            }
            c += 12; rg += 4;
            if (rg >= 5) { rg -= 5; c++; }
        }
    } else {
        // ---- generic taps: horizontal pass with ufixedpoint16 saturating sums
        for (int i = lane; i < RAW_N * BL_N; i += 64) {
            int r = i / BL_N, c = i - r * BL_N;
            uint32_t acc = 0;
#pragma unroll
            for (int k = 0; k < 7; k++) {
                uint32_t t = min(tp[k] * (uint32_t)raw[r * RAW_P + c + k], 0xFFFFu);
                acc = min(acc + t, 0xFFFFu);
            }
            hb[r * H_P + c] = (uint16_t)acc;
        }
        WAVE_LDS_SYNC();
        // ---- vertical pass: ufixedpoint32 saturating sums, round, saturate to u8
        for (int i = lane; i < BL_N * BL_N; i += 64) {
            int r = i / BL_N, c = i - r * BL_N;
            unsigned long long acc = 0;
#pragma unroll
            for (int k = 0; k < 7; k++) {
                acc += (unsigned long long)tp[k] * hb[(r + k) * H_P + c];
                acc = acc > 0xFFFFFFFFull ? 0xFFFFFFFFull : acc;
            }
            unsigned long long v = (acc + 0x8000ull) >> 16;
            bl[c * BL_P + r] = (uint8_t)(v > 255 ? 255 : v);
        }
    }
    WAVE_LDS_SYNC();
    DP_T(dp4);
    DP_ACC(3, dp2, dp4);                                         // row + column pass (2 = row pass alone)

    // ---- intensity centroid (ORBFinder.cpp:16-43): integer moments over the umax disc
    //      m10 = sum u*I, m01 = sum v*I as v_dot4_u32_u8 over aligned dwords: sum (u+15)*I - 15*sum I, and v * (row sum)
    int m10 = 0, m01 = 0;
    {
        const uint32_t* bl32 = reinterpret_cast<const uint32_t*>(bl);
        // lane = (row r0 of 6, dword q of 10), rows r0 + 6k: the table index is lane + 60k
        const int r0 = lane / 10, q = lane - r0 * 10;
        if (lane < 60) {
            const uint32_t* w0 = bl32 + (3 + r0) * (BL_P / 4) + q;
            int sv = 0;                                        // sum over k of (r0 + 6k - 15) * s1
#pragma unroll
            for (int k = 0; k < 6; k++) {
                if (6 * k + 5 < 31 || r0 + 6 * k < 31) {
                    const uint32_t W = w0[6 * k * (BL_P / 4)];
                    const uint2 wt = c_momw.w[lane + 60 * k];
                    const int s1 = (int)__builtin_amdgcn_udot4(W, wt.y, 0u, false), su = (int)__builtin_amdgcn_udot4(W, wt.x, 0u, false);
                    // column-major tile: the lane's dword holds four ROWS of column r0 + 6k, so the weighted sum is the v moment and the
                    // column index weights the u moment (the disc is symmetric: the same table serves both orientations)
                    m01 += su - 15 * s1; sv += (6 * k - 15) * s1; m10 += s1;       // m10 holds sum s1 until the line below
                }
            }
            m10 = r0 * m10 + sv;
        }
    }
    m10 = wave_sum(m10); m01 = wave_sum(m01);
    const float angle = fast_atan2_deg((float)m01, (float)m10);
    DP_T(dp5);
    DP_ACC(4, dp4, dp5);

    // ---- rBRIEF (ORBFinder.cpp:89-129)
    const float factorPI = (float)(3.1415926535897932384626433832795 / 180.f);
    const float theta = __fmul_rn(angle, factorPI);
    double sin_t, cos_t;
    hs_lean_sincos((double)theta, &sin_t, &cos_t);              // equal to libm's after the rounding to float for every float theta (lean_sincos.h)
    const float a = (float)cos_t, b = (float)sin_t;
    uint8_t* dout = desc + ((size_t)oimg * cap + g) * HS_DESC_BYTES;
    unsigned long long bits[4];
#pragma unroll
    for (int r = 0; r < 4; r++) {
        int t = 64 * r + lane;
        const float4 pt = *reinterpret_cast<const float4*>(&c_pattern.v[4 * t]);
        typedef float hs_f2 __attribute__((ext_vector_type(2)));   // both points of the test at once: v_pk_mul_f32 / v_pk_add_f32 (IEEE per component)
        const hs_f2 pxv = {pt.x, pt.y}, pyv = {pt.z, pt.w};      // table order (x0, x1, y0, y1)
        // cvRound = round half to even: x + 1.5 * 2^23 has the rounded integer in its low mantissa bits (|x| < 27); the tile offset
        // (18 + dx) * BL_P + 18 + dy (column-major tile) comes out of one 24-bit multiply-add on those bits
        const float M = 12582912.0f;
        const uint32_t K = 0x400000u * BL_P + 0x4B400000u - (18 * BL_P + 18);      // the biases of the two encodings minus the tile centre
        const hs_f2 fyv = (pxv * b + pyv * a) + M, fxv = (pxv * a - pyv * b) + M;      // -ffp-contract=off: separately rounded products and sums
        const uint32_t fy0 = __float_as_uint(fyv.x), fy1 = __float_as_uint(fyv.y), fx0 = __float_as_uint(fxv.x), fx1 = __float_as_uint(fxv.y);
        int t0 = bl[__umul24(fx0, BL_P) + fy0 - K];         // column-major tile; the multiply takes fx's low 24 bits (0x400000 + dx), fy enters whole
        int t1 = bl[__umul24(fx1, BL_P) + fy1 - K];
        bits[r] = __ballot(t0 < t1);
    }
    if (lane == 0) {                                            // the four ballots (wave-uniform) leave with the keypoint record: one predicated block
        typedef unsigned long long hs_u64x2 __attribute__((ext_vector_type(2)));
        *reinterpret_cast<hs_u64x2*>(dout) = hs_u64x2{bits[0], bits[1]};
        *reinterpret_cast<hs_u64x2*>(dout + 16) = hs_u64x2{bits[2], bits[3]};
        hs_keypoint k;
        // keypoint->pt *= scale for level != 0 (ORBExtractor.cpp:546-552)
        k.x = level ? __fmul_rn((float)cx, L.scale) : (float)cx;
        k.y = level ? __fmul_rn((float)cy, L.scale) : (float)cy;
        k.size = L.kp_size; k.angle = angle; k.response = (float)score; k.octave = level;
        kps[(size_t)oimg * cap + g] = k;
    }
    DP_T(dp6);
    DP_ACC(5, dp5, dp6);
    DP_ACC(6, dp0, dp6);
    DP_ACC(7, 0ull, 1ull);
}

void hs_launch_describe(const HsLevel* d_lv, int nlevels, HsImg0 img0, int batch,
                        const uint32_t* sel_xys, const int32_t* sel_count, const uint16_t* sel_perm, int sel_img_stride, int max_sel,
                        const uint16_t* taps7, HsOut out, hipStream_t s, bool fast_taps, HsStripFuse strips)
{
    int per_img = max_sel < out.cap ? max_sel : out.cap;
    dim3 grid((per_img + KP_PER_BLOCK - 1) / KP_PER_BLOCK, batch, 1);
    if (grid.x == 0) grid.x = 1;
    if (strips.enabled) grid.x += 1;                           // the strips workgroup of every image (acts for the right images only)
    if (fast_taps)
        hipLaunchKernelGGL(k_describe<true>, grid, dim3(64 * KP_PER_BLOCK), 0, s, d_lv, nlevels, img0, sel_xys, sel_count, sel_perm, sel_img_stride, taps7, out, strips);
    else
        hipLaunchKernelGGL(k_describe<false>, grid, dim3(64 * KP_PER_BLOCK), 0, s, d_lv, nlevels, img0, sel_xys, sel_count, sel_perm, sel_img_stride, taps7, out, strips);
}
