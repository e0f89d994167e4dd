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
// Each lane produces 4 consecutive destination pixels and stores one dword (256 B per wave row).  The source
// pixels of those 4 outputs span <= 11 bytes (scale <= 2), so each of the two source rows is fetched as up to
// four aligned dwords and the taps are picked out with v_alignbyte instead of 16 byte loads; the x table is one
// 8-byte record {sx, a0, a1, -} per output.
#include "hs_internal.h"
#include <atomic>
#include <algorithm>
#include <cstring>
#include <cstdlib>

template <bool ALIGNED>
__global__ __launch_bounds__(256) void k_resize_level(const HsLevel* __restrict__ lv, int level, HsImg0 img0)
{
    const HsLevel& D = lv[level];
    const int img = blockIdx.z;
    const int dy = blockIdx.y * 4 + threadIdx.y;
    const int dx0 = (blockIdx.x * 64 + threadIdx.x) * 4;
    if (dy >= D.h || dx0 >= D.w) return;

    const uint8_t* sbase; size_t spitch;
    if (level == 1) { sbase = hs_img0_ptr(img0, img); spitch = img0.row_stride; }
    else { const HsLevel& S = lv[level - 1]; sbase = S.base + (size_t)img * S.img_stride; spitch = S.pitch; }
    const int sw = lv[level - 1].w, sh = lv[level - 1].h;

    const int sy = D.yofs[dy];
    const int b0 = D.ibeta[2 * dy], b1 = D.ibeta[2 * dy + 1];
    const int sy0 = sy < 0 ? 0 : (sy >= sh ? sh - 1 : sy);
    const int sy1 = sy + 1 < 0 ? 0 : (sy + 1 >= sh ? sh - 1 : sy + 1);
    const uint8_t* S0 = sbase + (size_t)sy0 * spitch;
    const uint8_t* S1 = sbase + (size_t)sy1 * spitch;

    // x table records for the 4 outputs (dx beyond the row reuse the last valid one; their bytes land in the row padding)
    const HsXTab* xt = reinterpret_cast<const HsXTab*>(D.xofs);
    HsXTab t[4];
#pragma unroll
    for (int i = 0; i < 4; i++) t[i] = xt[min(dx0 + i, D.w - 1)];

    uint32_t packed = 0;
    const int base = t[0].sx & ~3;
    const int last = min(t[3].sx + 1, sw - 1);                 // last source byte any of the 4 outputs reads
    if (ALIGNED && last - base < 12) {
        uint32_t r0[4], r1[4];
        const uint32_t* p0 = reinterpret_cast<const uint32_t*>(S0 + base);
        const uint32_t* p1 = reinterpret_cast<const uint32_t*>(S1 + base);
#pragma unroll
        for (int j = 0; j < 4; j++) {
            bool need = base + 4 * j <= last;
            r0[j] = need ? p0[j] : 0u;
            r1[j] = need ? p1[j] : 0u;
        }
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int k = t[i].sx - base;                       // 0..9
            const int q = k >> 2;
            uint32_t lo0 = q == 0 ? r0[0] : (q == 1 ? r0[1] : r0[2]), hi0 = q == 0 ? r0[1] : (q == 1 ? r0[2] : r0[3]);
            uint32_t lo1 = q == 0 ? r1[0] : (q == 1 ? r1[1] : r1[2]), hi1 = q == 0 ? r1[1] : (q == 1 ? r1[2] : r1[3]);
            uint32_t w0 = __builtin_amdgcn_alignbyte(hi0, lo0, k & 3);     // bytes k, k+1 in the low half
            uint32_t w1 = __builtin_amdgcn_alignbyte(hi1, lo1, k & 3);
            int s00 = w0 & 0xFF, s01 = (w0 >> 8) & 0xFF, s10 = w1 & 0xFF, s11 = (w1 >> 8) & 0xFF;
            int h0, h1;
            if (dx0 + i < D.xmax) { h0 = s00 * t[i].a0 + s01 * t[i].a1; h1 = s10 * t[i].a0 + s11 * t[i].a1; }
            else { h0 = s00 * 2048; h1 = s10 * 2048; }          // S[xofs]*ONE past xmax (right tap not read)
            int v = (((b0 * (h0 >> 4)) >> 16) + ((b1 * (h1 >> 4)) >> 16) + 2) >> 2;
            packed |= (uint32_t)(v & 0xFF) << (8 * i);
        }
    } else {
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int sx = t[i].sx;
            const int sx1 = sx + 1 < sw ? sx + 1 : sw - 1;
            int h0, h1;
            if (dx0 + i < D.xmax) { h0 = S0[sx] * t[i].a0 + S0[sx1] * t[i].a1; h1 = S1[sx] * t[i].a0 + S1[sx1] * t[i].a1; }
            else { h0 = S0[sx] * 2048; h1 = S1[sx] * 2048; }
            int v = (((b0 * (h0 >> 4)) >> 16) + ((b1 * (h1 >> 4)) >> 16) + 2) >> 2;
            packed |= (uint32_t)(v & 0xFF) << (8 * i);
        }
    }
    uint8_t* drow = D.base + (size_t)img * D.img_stride + (size_t)dy * D.pitch;
    *reinterpret_cast<uint32_t*>(drow + dx0) = packed;   // pitch is a multiple of 64: padding bytes may be written
}

// ---- the two passes of the LDS-staged kernels, written for the VALU (the kernels are bound by instruction issue, not by HBM):
//   horizontal  H = S[sx]*a0 + S[sx+1]*a1 is needed as (H >> 4) in 16 bits.  The coefficient pair is pre-multiplied by 16 (a <= 2048, so
//               16 a fits 16 bits), v_dot2_u32_u16 then yields H << 4 whose bytes 1..2 ARE (H >> 4) & 0xFFFF: one v_perm_b32 packs two of them.
//               Per value: one v_perm (byte pair -> two 16-bit fields) + one v_dot2; per pair one more v_perm.  (Before: + a shift per value
//               and a shift-or per pair.)
//   vertical    ((b0*H0) >> 16) + ((b1*H1) >> 16) + 2) >> 2: two 24-bit multiplies, one v_perm that takes the high halves of both
//               products, one v_dot2 against (1,1) with the rounding constant as its accumulator, one shift.
typedef unsigned short hs_us2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pyr_hpair(uint32_t whi, uint32_t wlo, uint32_t sel0, uint32_t sel1, uint32_t coef0, uint32_t coef1)
{
    const uint32_t h0 = __builtin_amdgcn_udot2(__builtin_bit_cast(hs_us2, __builtin_amdgcn_perm(whi, wlo, sel0)), __builtin_bit_cast(hs_us2, coef0), 0u, false);
    const uint32_t h1 = __builtin_amdgcn_udot2(__builtin_bit_cast(hs_us2, __builtin_amdgcn_perm(whi, wlo, sel1)), __builtin_bit_cast(hs_us2, coef1), 0u, false);
    return __builtin_amdgcn_perm(h1, h0, 0x06050201u);         // (H1 >> 4) << 16 | (H0 >> 4)
}
__device__ __forceinline__ uint32_t pyr_vpix4(uint32_t h0, uint32_t h1, uint32_t b0, uint32_t b1)     // h0, h1: 16-bit values; returns 4 * pixel + (0..3)
{
    const uint32_t p0 = __umul24(h0, b0), p1 = __umul24(h1, b1);
    const uint32_t hi = __builtin_amdgcn_perm(p1, p0, 0x07060302u);                                    // (p1 >> 16) << 16 | (p0 >> 16)
    return __builtin_amdgcn_udot2(__builtin_bit_cast(hs_us2, hi), __builtin_bit_cast(hs_us2, 0x00010001u), 2u, false);
}
__device__ __forceinline__ uint32_t pyr_vquad(uint2 H0, uint2 H1, uint32_t b0, uint32_t b1)
{
    const uint32_t v0 = pyr_vpix4(H0.x & 0xFFFFu, H1.x & 0xFFFFu, b0, b1), v1 = pyr_vpix4(H0.x >> 16, H1.x >> 16, b0, b1);
    const uint32_t v2 = pyr_vpix4(H0.y & 0xFFFFu, H1.y & 0xFFFFu, b0, b1), v3 = pyr_vpix4(H0.y >> 16, H1.y >> 16, b0, b1);
    // every value is < 1024: two per register as 16-bit fields, one packed shift for both, one byte permute for all four
    uint32_t a = v0 | (v1 << 16), c = v2 | (v3 << 16), a2, c2;
    const uint32_t two = 0x00020002u;                           // a shift count per 16-bit field (an inline constant would only reach the low one)
    asm("v_pk_lshrrev_b16 %0, %2, %1" : "=v"(a2) : "v"(a), "v"(two));
    asm("v_pk_lshrrev_b16 %0, %2, %1" : "=v"(c2) : "v"(c), "v"(two));
    return __builtin_amdgcn_perm(c2, a2, 0x06040200u);                                                  // bytes 0 and 2 of both
}

// Source rectangle -> LDS with 16-byte vectors: a wave takes whole source rows, floor(64 / nvec) at a time (no per-lane division by the runtime nvec).
// FOUR rows per lane are in flight before the first LDS write (round 4): written as `load; store` per row the loop waited for every load on the spot —
// three dependent HBM round trips at the head of every workgroup (7 k of the 22 k cycles a workgroup of the levels 5-7 launch lives).  Rows past the
// end are clamped to the last row: a duplicate load and an identical write instead of a branch.
template <int NW>
__device__ __forceinline__ void pyr_stage_source(const uint8_t* src0 /*wave-uniform*/, uint32_t spitch, uint8_t* s_dst, uint32_t dpitch, int nvec, int nrows, int tx, int wave)
{
    const int rpw = nvec <= 16 ? 4 : (nvec <= 21 ? 3 : (nvec <= 32 ? 2 : 1));          // rows per wave step
    const int rl = (tx >= nvec) + (tx >= 2 * nvec) + (tx >= 3 * nvec), q = tx - rl * nvec;
    if (tx >= rpw * nvec) return;
#ifdef HS_PYR_NOLOAD          // diagnostic build (results are garbage): how much of a launch is the wait for its source rows?
    return;
#endif
    const int step = NW * rpw, last = nrows - 1;
    for (int r = wave * rpw + rl; r < nrows; r += 4 * step) {
        const int r1 = min(r + step, last), r2 = min(r + 2 * step, last), r3 = min(r + 3 * step, last);
        const hs_u32x4 v0 = hs_gload_off<hs_u32x4>(src0, (uint32_t)r * spitch + 16u * (uint32_t)q);
        const hs_u32x4 v1 = hs_gload_off<hs_u32x4>(src0, (uint32_t)r1 * spitch + 16u * (uint32_t)q);
        const hs_u32x4 v2 = hs_gload_off<hs_u32x4>(src0, (uint32_t)r2 * spitch + 16u * (uint32_t)q);
        const hs_u32x4 v3 = hs_gload_off<hs_u32x4>(src0, (uint32_t)r3 * spitch + 16u * (uint32_t)q);
        *reinterpret_cast<hs_u32x4*>(&s_dst[(uint32_t)r * dpitch + 16 * q]) = v0;
        *reinterpret_cast<hs_u32x4*>(&s_dst[(uint32_t)r1 * dpitch + 16 * q]) = v1;
        *reinterpret_cast<hs_u32x4*>(&s_dst[(uint32_t)r2 * dpitch + 16 * q]) = v2;
        *reinterpret_cast<hs_u32x4*>(&s_dst[(uint32_t)r3 * dpitch + 16 * q]) = v3;
    }
}

// LDS-staged variant (the fast path): a workgroup produces a 256 x LT_ROWS destination tile in three steps.
//   A  the source rectangle it needs is fetched once with 16-byte coalesced loads into LDS (each source row is read from HBM/L2 once
//      per tile instead of once per destination row)
//   B  horizontal pass, once per SOURCE row: H = S[sx]*a0 + S[sx+1]*a1 for the tile's 256 columns, stored as (H >> 4) in 16 bits
//      (the only form the vertical pass uses).  A lane makes 4 adjacent columns from an 8-byte window of the source row: one
//      v_perm_b32 (byte pair -> two 16-bit fields, selectors precomputed per lane) + one v_dot2_u32_u16 per value.  Vertically adjacent
//      destination rows share their source rows, so this pass runs ~1.4x per destination pixel instead of 2x.
//   C  vertical pass: ((b0*H0)>>16) + ((b1*H1)>>16) + 2) >> 2 from two 8-byte LDS reads per 4 pixels.
// ~20 VALU instructions per destination pixel instead of ~43 for the direct form.  (A persistent, software-pipelined variant that
// prefetched the next tile's vectors into registers was measured slower: 0.164 vs 0.146 ms for the 7 levels of 32 frames.)
// Needs 16-byte aligned source rows and a scale <= 2 (the 4 columns of a lane then span <= 8 source bytes).
#define LT_ROWS 16
__global__ __launch_bounds__(256) void k_resize_level_lds(const HsLevel* __restrict__ lv, int level, HsImg0 img0, int lds_pitch, int lds_rows)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t s_src[];
    uint16_t* const s_h = reinterpret_cast<uint16_t*>(s_src + (size_t)lds_pitch * lds_rows);      // [lds_rows][256] (H >> 4)
    const HsLevel& D = lv[level];
    const int img = blockIdx.z;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;        // 64 x 4 lanes
    const int dx_tile = blockIdx.x * 256, dy_tile = blockIdx.y * LT_ROWS;

    const uint8_t* sbase; size_t spitch;
    if (level == 1) { sbase = hs_img0_ptr(img0, img); spitch = img0.row_stride; }
    else { const HsLevel& S = lv[level - 1]; sbase = S.base + (size_t)img * S.img_stride; spitch = S.pitch; }
    const int sw = lv[level - 1].w, sh = lv[level - 1].h;
    const HsXTab* xt = reinterpret_cast<const HsXTab*>(D.xofs);

    // source rectangle of this tile (wave-uniform)
    const int dy_last = min(dy_tile + LT_ROWS, D.h) - 1, dx_last = min(dx_tile + 256, D.w) - 1;
    const int sy_first = min(max(hs_cload_i16(D.yofs, dy_tile), 0), sh - 1);
    const int sy_last = min(max(hs_cload_i16(D.yofs, dy_last) + 1, 0), sh - 1);
    const int col0 = (int)(int16_t)hs_cload<uint32_t>(&xt[dx_tile]) & ~15;      // .sx = low half of the record's first dword
    const int col_last = min((int)(int16_t)hs_cload<uint32_t>(&xt[dx_last]) + 1, sw - 1);
    const int nvec = ((col_last - col0) >> 4) + 1, nrow = sy_last - sy_first + 1;      // host guarantees nvec*16 <= lds_pitch - 16, nrow <= lds_rows
    // ---- A: a wave takes whole source rows, floor(64 / nvec) at a time (no per-lane division by the runtime nvec)
    pyr_stage_source<4>(hs_uniform_ptr(sbase + (size_t)sy_first * spitch + col0), (uint32_t)spitch, s_src, (uint32_t)lds_pitch, nvec, nrow, threadIdx.x & 63, __builtin_amdgcn_readfirstlane(threadIdx.x >> 6));
    // per-lane column data (independent of the row): 8-byte window position, byte-pair selectors, coefficient pairs
    const int dx0 = dx_tile + 4 * tx;
    HsXTab t[4];
#pragma unroll
    for (int i = 0; i < 4; i++) t[i] = __builtin_bit_cast(HsXTab, hs_gload<uint64_t>(&xt[min(dx0 + i, D.w - 1)]));
    const int o0 = t[0].sx - col0;                              // byte offset of the window in a staged row
    const int wbase = o0 & ~3, wshift = o0 & 3;
    uint32_t sel[4], coef[4];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const uint32_t q = (uint32_t)(t[i].sx - t[0].sx);       // 0..6: bytes q, q+1 of the window (a1 == 0 wherever sx+1 is past the row)
        sel[i] = q | 0x0c00u | ((q + 1) << 16) | 0x0c000000u;
        coef[i] = ((uint32_t)(uint16_t)t[i].a0 << 4) | ((uint32_t)(uint16_t)t[i].a1 << 20);      // 16 a0 | 16 a1 << 16
    }
    __syncthreads();
    // ---- B
    for (int r = __builtin_amdgcn_readfirstlane(ty); r < nrow; r += 4) {
        const uint32_t* w = reinterpret_cast<const uint32_t*>(&s_src[r * lds_pitch + wbase]);
        const uint32_t d0 = w[0], d1 = w[1], d2 = w[2];
        const uint32_t wlo = __builtin_amdgcn_alignbyte(d1, d0, wshift), whi = __builtin_amdgcn_alignbyte(d2, d1, wshift);
        *reinterpret_cast<uint2*>(&s_h[r * 256 + 4 * tx]) = make_uint2(pyr_hpair(whi, wlo, sel[0], sel[1], coef[0], coef[1]), pyr_hpair(whi, wlo, sel[2], sel[3], coef[2], coef[3]));
    }
    __syncthreads();
    // ---- C: the destination row of a wave is uniform, so its source rows and weights come from scalar loads
    if (dx0 >= D.w) return;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    uint8_t* const dimg = D.base + (size_t)img * D.img_stride;
    int sy4[LT_ROWS / 4]; uint32_t b4[LT_ROWS / 4];                    // the four rows' parameters (scalar loads) before the first use
#pragma unroll
    for (int rr = 0; rr < LT_ROWS / 4; rr++) { const int dy = min(dy_tile + wave + 4 * rr, D.h - 1); sy4[rr] = hs_cload_i16(D.yofs, dy); b4[rr] = hs_cload<uint32_t>(D.ibeta + 2 * dy); }
#pragma unroll
    for (int rr = 0; rr < LT_ROWS / 4; rr++) {
        const int dy = dy_tile + wave + 4 * rr;
        if (dy >= D.h) break;
        const int sy = sy4[rr];
        const uint32_t b01 = b4[rr];
        const uint32_t b0 = b01 & 0xFFFFu, b1 = b01 >> 16;
        const int r0 = min(max(sy, 0), sh - 1) - sy_first, r1 = min(max(sy + 1, 0), sh - 1) - sy_first;
        const uint2 H0 = *reinterpret_cast<const uint2*>(&s_h[r0 * 256 + 4 * tx]);
        const uint2 H1 = *reinterpret_cast<const uint2*>(&s_h[r1 * 256 + 4 * tx]);
        hs_gstore<uint32_t>(dimg + (size_t)dy * D.pitch + dx0, pyr_vquad(H0, H1, b0, b1));   // pitch is a multiple of 64: padding bytes may be written
    }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// Two levels per launch.  A workgroup owns one tile of level B = l+1 (TBX x 16 px) and everything above it: it stages the rectangle of level
// S = l-1 that the tile's region of level A = l needs, makes that region of A (horizontal sums per source row, vertical combine), keeps it in
// LDS, writes the part of it that the workgroup OWNS to HBM (ownership = the partition of A induced by the first source column / row of
// every B tile, so every pixel of A is written exactly once; the 1-2 px halo a tile needs from its neighbours' parts is recomputed, with
// identical bits), and then makes its tile of B from the LDS copy.  The chain of launches shrinks from L-1 to ceil((L-1)/2), level A is
// never re-read from memory, and the source rectangle is shared by both levels.  Same arithmetic as k_resize_level_lds.
// Round 3: TABLE-DRIVEN.  The kernel used to derive its geometry from the HsLevel array and the resize tables with a chain of ~8 dependent
// scalar loads before its first vector load, and every row of a vertical pass cost ~38 scalar instructions (64-bit table addresses, clamps, a
// re-load of the pitch, a 64-bit multiply for the store address): 548 scalar against 518 vector instructions per wave.  Now the level
// descriptions are a kernel argument, the tile geometry is two 32-byte records (HsPyrXTile / HsPyrYTile), a destination row is one 8-byte
// record (HsPyrRow: clamped source rows + weights, fetched one row ahead) and stores use a scalar row base + a 32-bit lane offset.
typedef uint32_t hs_u32x8 __attribute__((ext_vector_type(8)));
typedef uint32_t hs_u32x2 __attribute__((ext_vector_type(2)));
#ifndef FZ_ROWS
#define FZ_ROWS 16
#endif
#define FZ_APITCH 272             // LDS pitch of the level-A region: 256 columns + the 3-dword window over-read of the last lane
template <typename T> __device__ __forceinline__ void hs_gstore_off(uint8_t* uniform_base, uint32_t lane_off, T v)
{
    *(HS_GLOBAL T*)((HS_GLOBAL uint8_t*)(uintptr_t)uniform_base + lane_off) = v;
}
// NW = wavefronts per workgroup: 4, or 8 when the launch has few workgroups per CU (small batches) — the same tile, half the rows per wave, so a
// workgroup's life (the launch's critical path when every workgroup is resident at once) is shorter.
template <int NW>
__global__ __launch_bounds__(64 * NW) void k_resize_two_levels(HsPyrFuse F, HsImg0 img0)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const int TBX = F.tbx, lds_pitch = F.lds_pitch;
    uint8_t* const s_src = smem;                                                         // [SR][lds_pitch] source rectangle, later the level-A region
    uint8_t* const s_a = smem;                                                           // [AR][FZ_APITCH]
    uint16_t* const s_h = reinterpret_cast<uint16_t*>(smem + (size_t)lds_pitch * F.sr);  // [SR][256] (H >> 4) for level A, later [AR][256] for level B
    const int img = blockIdx.z;
    const int tx = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const HsPyrXTile X = __builtin_bit_cast(HsPyrXTile, hs_cload<hs_u32x8>(&F.xt[blockIdx.x]));
    const HsPyrYTile Y = __builtin_bit_cast(HsPyrYTile, hs_cload<hs_u32x8>(&F.yt[blockIdx.y]));
    const uint8_t* sbase; uint32_t spitch;
    if (F.sbase == nullptr) { sbase = hs_img0_ptr(img0, img); spitch = (uint32_t)img0.row_stride; }
    else { sbase = F.sbase + (size_t)img * F.s_img_stride; spitch = (uint32_t)F.spitch; }
    const int bx0 = blockIdx.x * TBX, by0 = blockIdx.y * FZ_ROWS;
    const int ax0 = X.ax0, ay0 = Y.ay0, nAr = Y.ay_last - Y.ay0 + 1, nSr = Y.n_sr, nvec = X.nvec;

    // ---- 1: the source rectangle (a wave takes whole source rows, floor(64 / nvec) at a time)
    pyr_stage_source<NW>(hs_uniform_ptr(sbase + (size_t)Y.sy_first * spitch + X.col0), spitch, s_src, (uint32_t)lds_pitch, nvec, nSr, tx, wave);
    // per-lane column data of one horizontal pass: window position, byte-pair selectors, coefficient pairs
    struct ColData { int wbase, wshift; uint32_t sel[4], coef[4]; };
    auto col_data = [&](const HsXTab* xt, int dx0, int wmax, int origin) {
        HsXTab t[4];
#pragma unroll
        for (int i = 0; i < 4; i++) t[i] = __builtin_bit_cast(HsXTab, hs_gload<uint64_t>(&xt[min(dx0 + i, wmax)]));
        ColData c;
        const int o0 = t[0].sx - origin;
        c.wbase = o0 & ~3; c.wshift = o0 & 3;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const uint32_t q = (uint32_t)(t[i].sx - t[0].sx);
            c.sel[i] = q | 0x0c00u | ((q + 1) << 16) | 0x0c000000u;
            c.coef[i] = ((uint32_t)(uint16_t)t[i].a0 << 4) | ((uint32_t)(uint16_t)t[i].a1 << 20);  // 16 a0 | 16 a1 << 16
        }
        return c;
    };
    auto h_pass = [&](const uint8_t* src, int pitch, int nrow, const ColData& c) {
        for (int r = wave; r < nrow; r += NW) {
            const uint32_t* w = reinterpret_cast<const uint32_t*>(&src[r * pitch + c.wbase]);
            const uint32_t d0 = w[0], d1 = w[1], d2 = w[2];
            const uint32_t wlo = __builtin_amdgcn_alignbyte(d1, d0, c.wshift), whi = __builtin_amdgcn_alignbyte(d2, d1, c.wshift);
            *reinterpret_cast<uint2*>(&s_h[r * 256 + 4 * tx]) = make_uint2(pyr_hpair(whi, wlo, c.sel[0], c.sel[1], c.coef[0], c.coef[1]), pyr_hpair(whi, wlo, c.sel[2], c.sel[3], c.coef[2], c.coef[3]));
        }
    };
    // a destination row = one 8-byte record of the TILE's row table (scalar load): byte offsets of its two source rows in the sums buffer + weights
    const uint8_t* const h_lane = reinterpret_cast<const uint8_t*>(s_h) + 8 * tx;
    auto v_combine = [&](const HsPyrRow& rec) -> uint32_t {
        const uint2 H0 = *reinterpret_cast<const uint2*>(h_lane + rec.off0);
        const uint2 H1 = *reinterpret_cast<const uint2*>(h_lane + rec.off1);
        return pyr_vquad(H0, H1, rec.b0, rec.b1);
    };
    const ColData cA = col_data(F.xtA, ax0 + 4 * tx, F.aw - 1, X.col0);
    __syncthreads();
    // ---- 2: horizontal sums of the source rows for the level-A columns
    h_pass(s_src, lds_pitch, nSr, cA);
    __syncthreads();
    // ---- 3: level-A region -> LDS (it overlays the source rectangle, which is dead now) and, for the owned part, HBM
    {
        uint8_t* const aimg = F.abase + (size_t)img * F.a_img_stride;
        const uint32_t acol = (uint32_t)(ax0 + 4 * tx);
        const bool own_col = (int)acol < X.own_x1;
        // the row records are scalar loads: fetched ONE ROW AHEAD, so that their latency hides behind the current row's arithmetic; the table is
        // per tile and padded, so the walk is `pointer += NW` with no clamp, and the store base advances by NW rows (no 64-bit multiply per row)
        const HsPyrRow* recp = F.rowA + ((size_t)blockIdx.y * (uint32_t)F.slotA + (uint32_t)wave);
        HsPyrRow nxt = __builtin_bit_cast(HsPyrRow, hs_cload<hs_u32x2>(recp));
        uint8_t* rowp = aimg + (size_t)(ay0 + wave) * (uint32_t)F.apitch;
        const uint32_t rstep = (uint32_t)NW * (uint32_t)F.apitch;
        uint32_t* arow = reinterpret_cast<uint32_t*>(&s_a[wave * FZ_APITCH + 4 * tx]);
        const int own_k1 = Y.own_y1 - ay0;
        for (int k = wave; k < nAr; k += NW) {
            const HsPyrRow rec = nxt;
            recp += NW;
            nxt = __builtin_bit_cast(HsPyrRow, hs_cload<hs_u32x2>(recp));
            const uint32_t px = v_combine(rec);
            *arow = px;
            if (own_col && k < own_k1) hs_gstore_off<uint32_t>(rowp, acol, px);
            arow += NW * (FZ_APITCH / 4); rowp += rstep;
        }
    }
    const ColData cB = col_data(F.xtB, bx0 + 4 * tx, F.bw - 1, ax0);
    __syncthreads();
    // ---- 4: horizontal sums of the level-A rows for the tile's level-B columns (the sums overlay the level-A sums)
    if (4 * tx < TBX) h_pass(s_a, FZ_APITCH, nAr, cB);
    __syncthreads();
    // ---- 5: the level-B tile
    if (4 * tx < TBX && bx0 + 4 * tx < F.bw) {
        uint8_t* const bimg = F.bbase + (size_t)img * F.b_img_stride;
        const HsPyrRow* recp = F.rowB + ((size_t)blockIdx.y * (FZ_ROWS + 8) + (uint32_t)wave);
        HsPyrRow rec4[FZ_ROWS / NW];                                    // the rows' records (scalar loads at constant offsets) before the first use
#pragma unroll
        for (int rr = 0; rr < FZ_ROWS / NW; rr++) rec4[rr] = __builtin_bit_cast(HsPyrRow, hs_cload<hs_u32x2>(recp + NW * rr));
        uint8_t* rowp = bimg + (size_t)(by0 + wave) * (uint32_t)F.bpitch;
        const uint32_t rstep = (uint32_t)NW * (uint32_t)F.bpitch;
#pragma unroll
        for (int rr = 0; rr < FZ_ROWS / NW; rr++) {
            const int by = by0 + wave + NW * rr;
            if (by >= F.bh) break;
            hs_gstore_off<uint32_t>(rowp, (uint32_t)(bx0 + 4 * tx), v_combine(rec4[rr]));
            rowp += rstep;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// A chain of 2 or 3 levels per launch: the scheme of k_resize_two_levels as a loop over stages.  Stage i stages (i = 0: from global memory)
// or finds (i > 0: the region the previous stage left in LDS) its source, makes the horizontal sums of the source rows for its region's
// columns, combines them vertically into its region, stores the part of the region the workgroup OWNS and keeps the region in LDS for the next
// stage.  Everything wave-uniform comes from two 32-byte records per stage (HsPyrStageX / HsPyrStageY, hs_pyramid_plan_chain).  Used for
// the last THREE levels of a pyramid with an odd number of levels to make (levels 5, 6, 7 of 8: one launch instead of two).
#ifdef HS_PYR_PROFILE      // make EXTRA=-DHS_PYR_PROFILE: clock stamps of workgroup (0, 0, 0) of every k_resize_chain launch (tools/pyramid_phase_profile.py)
__device__ unsigned long long g_pyr_prof[64];
__device__ unsigned long long g_pyr_wg[4096 * 2];            // real-time (100 MHz) start / end stamp of every workgroup of the last k_resize_chain launch
extern "C" void hs_debug_pyr_profile(unsigned long long* out64) { (void)hipDeviceSynchronize(); (void)hipMemcpyFromSymbol(out64, HIP_SYMBOL(g_pyr_prof), sizeof(unsigned long long) * 64); }
extern "C" void hs_debug_pyr_workgroups(unsigned long long* out8192) { (void)hipDeviceSynchronize(); (void)hipMemcpyFromSymbol(out8192, HIP_SYMBOL(g_pyr_wg), sizeof(unsigned long long) * 8192); }
#define PYR_MARK() do { if (threadIdx.x == 0 && blockIdx.x == gridDim.x / 2 && blockIdx.y == gridDim.y / 2 && blockIdx.z == 0 && pk < 62) g_pyr_prof[pk++] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define PYR_MARK()
#endif
template <int NW>
__global__ __launch_bounds__(64 * NW) void k_resize_chain(HsPyrChain F, HsImg0 img0)
{
#ifdef HS_PYR_PROFILE
    int pk = 0;
    const unsigned pyr_wg = (blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
    if (threadIdx.x == 0 && pyr_wg < 4096) g_pyr_wg[2 * pyr_wg] = __builtin_amdgcn_s_memrealtime();
#endif
    PYR_MARK();
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    uint8_t* const s_x = smem;                                                           // source rectangle (stage 0) / region of the previous stage
    uint16_t* const s_h = reinterpret_cast<uint16_t*>(smem + F.x_bytes);                 // [h_rows][256] (H >> 4)
    const int img = blockIdx.z;
    const int tx = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint8_t* sbase; uint32_t spitch;
    if (F.sbase == nullptr) { sbase = hs_img0_ptr(img0, img); spitch = (uint32_t)img0.row_stride; }
    else { sbase = F.sbase + (size_t)img * F.s_img_stride; spitch = (uint32_t)F.spitch; }
    int src_pitch = F.lds_pitch;
    // The stages are a dependent sequence inside the workgroup, and every stage starts with two dependent fetches: its tile records (scalar loads) and,
    // addressed by them, the four x-table entries of every lane.  At one workgroup per CU (small batches: the deep chains) nothing hides them, so
    // they are software-pipelined: the records of stage st + 1 are requested at the top of stage st, its x-table entries between the two passes.
    auto ld_x = [&](int st) { return __builtin_bit_cast(HsPyrStageX, hs_cload<hs_u32x8>(&F.st[st].tx[blockIdx.x])); };
    auto ld_y = [&](int st) { return __builtin_bit_cast(HsPyrStageY, hs_cload<hs_u32x8>(&F.st[st].ty[blockIdx.y])); };
    auto ld_t = [&](int st, const HsPyrStageX& X, uint64_t (&t)[4]) {
        const HsPyrStage& S = F.st[st];
#pragma unroll
        for (int i = 0; i < 4; i++) t[i] = hs_gload<uint64_t>(&S.xt[min(X.x0 + 4 * tx + i, S.w - 1)]);
    };
    HsPyrStageX X = ld_x(0); HsPyrStageY Y = ld_y(0);
    uint64_t traw[4]; ld_t(0, X, traw);
    for (int st = 0; st < F.nstage; st++) {
        const HsPyrStage& S = F.st[st];
        const int stn = min(st + 1, F.nstage - 1);                   // (the last stage re-requests its own records: harmless, keeps the loads unconditional)
        const HsPyrStageX Xn = ld_x(stn); const HsPyrStageY Yn = ld_y(stn);
        if (st == 0) pyr_stage_source<NW>(hs_uniform_ptr(sbase + (size_t)Y.src_y0 * spitch + X.src_x0), spitch, s_x, (uint32_t)src_pitch, X.nvec, Y.n_src, tx, wave);
        // per-lane column data of the stage's horizontal pass: window position, byte-pair selectors, coefficient pairs
        int wbase, wshift; uint32_t sel[4], coef[4];
        {
            HsXTab t[4];
#pragma unroll
            for (int i = 0; i < 4; i++) t[i] = __builtin_bit_cast(HsXTab, traw[i]);
            const int o0 = t[0].sx - X.src_x0;
            wbase = o0 & ~3; wshift = o0 & 3;
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const uint32_t q = (uint32_t)(t[i].sx - t[0].sx);
                sel[i] = q | 0x0c00u | ((q + 1) << 16) | 0x0c000000u;
                coef[i] = ((uint32_t)(uint16_t)t[i].a0 << 4) | ((uint32_t)(uint16_t)t[i].a1 << 20);
            }
        }
        __syncthreads();                                             // the source is in LDS (staged, or written by the previous stage)
        PYR_MARK();
        if (4 * tx < X.ncols) {
            // two source rows per iteration, all six LDS reads issued before the first use: with one or two waves per SIMD (small batches) the LDS
            // round trip of every row was exposed
            for (int r = wave; r < Y.n_src; r += 2 * NW) {
                const bool two = r + NW < Y.n_src;                    // uniform
                const uint32_t* wa = reinterpret_cast<const uint32_t*>(&s_x[r * src_pitch + wbase]);
                const uint32_t* wb = reinterpret_cast<const uint32_t*>(&s_x[(two ? r + NW : r) * src_pitch + wbase]);
                const uint32_t a0 = wa[0], a1 = wa[1], a2 = wa[2], b0 = wb[0], b1 = wb[1], b2 = wb[2];
                const uint32_t alo = __builtin_amdgcn_alignbyte(a1, a0, wshift), ahi = __builtin_amdgcn_alignbyte(a2, a1, wshift);
                const uint32_t blo = __builtin_amdgcn_alignbyte(b1, b0, wshift), bhi = __builtin_amdgcn_alignbyte(b2, b1, wshift);
                *reinterpret_cast<uint2*>(&s_h[r * 256 + 4 * tx]) = make_uint2(pyr_hpair(ahi, alo, sel[0], sel[1], coef[0], coef[1]), pyr_hpair(ahi, alo, sel[2], sel[3], coef[2], coef[3]));
                if (two) *reinterpret_cast<uint2*>(&s_h[(r + NW) * 256 + 4 * tx]) = make_uint2(pyr_hpair(bhi, blo, sel[0], sel[1], coef[0], coef[1]), pyr_hpair(bhi, blo, sel[2], sel[3], coef[2], coef[3]));
            }
        }
        ld_t(stn, Xn, traw);                                         // in flight during the vertical pass
        __syncthreads();                                             // the sums are complete, the source is dead: the region overlays it
        PYR_MARK();
        {
            uint8_t* const dimg = S.base + (size_t)img * S.img_stride;
            const uint32_t col = (uint32_t)(X.x0 + 4 * tx);
            const bool lane_on = 4 * tx < X.ncols, own_col = (int)col < X.own_x1;
            const bool keep = st + 1 < F.nstage;                     // uniform: a later stage reads the region from LDS
            const HsPyrRow* recp = S.rows + ((size_t)blockIdx.y * (uint32_t)S.slot + (uint32_t)wave);      // the tile's padded row table: `pointer += NW`, no clamp
            HsPyrRow nxt0 = __builtin_bit_cast(HsPyrRow, hs_cload<hs_u32x2>(recp)), nxt1 = __builtin_bit_cast(HsPyrRow, hs_cload<hs_u32x2>(recp + NW));
            const uint8_t* const h_lane = reinterpret_cast<const uint8_t*>(s_h) + 8 * tx;
            uint8_t* rowp = dimg + (size_t)(Y.y0 + wave) * (uint32_t)S.pitch;
            const uint32_t rstep = (uint32_t)NW * (uint32_t)S.pitch;
            uint32_t* xrow = reinterpret_cast<uint32_t*>(&s_x[wave * FZ_APITCH + 4 * tx]);
            // two destination rows per iteration (their four LDS reads in flight together); the row table is padded with copies of the tile's last row,
            // so the second row of the last iteration reads valid sums and is simply not stored
            for (int y = Y.y0 + wave; y <= Y.y_last; y += 2 * NW) {
                const HsPyrRow rec0 = nxt0, rec1 = nxt1;
                recp += 2 * NW;
                nxt0 = __builtin_bit_cast(HsPyrRow, hs_cload<hs_u32x2>(recp)); nxt1 = __builtin_bit_cast(HsPyrRow, hs_cload<hs_u32x2>(recp + NW));
                const bool two = y + NW <= Y.y_last;                  // uniform
                if (lane_on) {
                    const uint2 A0 = *reinterpret_cast<const uint2*>(h_lane + rec0.off0), A1 = *reinterpret_cast<const uint2*>(h_lane + rec0.off1);
                    const uint2 B0 = *reinterpret_cast<const uint2*>(h_lane + rec1.off0), B1 = *reinterpret_cast<const uint2*>(h_lane + rec1.off1);
                    const uint32_t pa = pyr_vquad(A0, A1, rec0.b0, rec0.b1), pb = pyr_vquad(B0, B1, rec1.b0, rec1.b1);
                    if (keep) { *xrow = pa; if (two) xrow[NW * (FZ_APITCH / 4)] = pb; }
                    if (own_col && y < Y.own_y1) hs_gstore_off<uint32_t>(rowp, col, pa);
                    if (own_col && two && y + NW < Y.own_y1) hs_gstore_off<uint32_t>(rowp + rstep, col, pb);
                }
                xrow += 2 * NW * (FZ_APITCH / 4); rowp += 2 * rstep;
            }
        }
        src_pitch = FZ_APITCH;
        X = Xn; Y = Yn;
        PYR_MARK();
    }
#ifdef HS_PYR_PROFILE
    if (threadIdx.x == 0 && blockIdx.x == gridDim.x / 2 && blockIdx.y == gridDim.y / 2 && blockIdx.z == 0) g_pyr_prof[63] = pk;
    __syncthreads();
    if (threadIdx.x == 0 && pyr_wg < 4096) g_pyr_wg[2 * pyr_wg + 1] = __builtin_amdgcn_s_memrealtime();
#endif
}

// Host side of k_resize_chain for the levels [first, first + n): walks the tiles of the LAST level from the largest tile width downwards until every
// stage's region fits 256 columns and the LDS buffers, with the expressions of hs_pyramid_build_tables / the two-level kernel.
void hs_pyramid_plan_chain(const HsLevel* h_lv, int first, int n, const int16_t* const* xtab, const int16_t* const* yofs, const int16_t* const* ibeta,
                           std::vector<uint64_t>& blob, HsPyrChain& C, size_t lds_max, int tile_rows)
{
    C = HsPyrChain{};
    if (tile_rows <= 0) tile_rows = FZ_ROWS;
    if (n < 2 || n > HS_PYR_CHAIN_MAX || first < 1) return;
    auto clampi = [](int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); };
    for (int i = 0; i < n; i++) {
        const HsLevel& D = h_lv[first + i]; const HsLevel& S = h_lv[first + i - 1];
        if ((double)S.w / D.w > 2.0 || (double)S.h / D.h > 2.0 || D.w < 8 || D.h < 1) return;
    }
    const HsLevel& LAST = h_lv[first + n - 1];
    const int nby = (LAST.h + tile_rows - 1) / tile_rows;
    // ---- rows (independent of the tile width): part starts, owned ends, regions and source spans — the same three steps as for the columns below
    std::vector<std::vector<HsPyrStageY>> ty(n, std::vector<HsPyrStageY>(nby));
    int h_rows = 0, x_rows0 = 0, x_rows = 0;
    for (int by = 0; by < nby; by++) {
        int y0 = by * tile_rows;
        for (int i = n - 1; i >= 0; i--) {
            ty[i][by] = HsPyrStageY{};
            ty[i][by].y0 = y0;
            if (i > 0) y0 = by == 0 ? 0 : clampi(yofs[first + i][y0], 0, h_lv[first + i - 1].h - 1);
        }
    }
    for (int by = 0; by < nby; by++)
        for (int i = n - 1; i >= 0; i--) {
            HsPyrStageY& t = ty[i][by];
            t.own_y1 = by + 1 < nby ? ty[i][by + 1].y0 : h_lv[first + i].h;
            if (t.own_y1 < t.y0) return;
        }
    for (int by = 0; by < nby; by++) {
        int need_last = std::min(by * tile_rows + tile_rows, LAST.h) - 1;          // last row of the stage's level that is really read
        for (int i = n - 1; i >= 0; i--) {
            const HsLevel& S = h_lv[first + i - 1];
            const int16_t* yo = yofs[first + i];
            HsPyrStageY& t = ty[i][by];
            t.y_last = std::max(need_last, t.own_y1 - 1);
            const int src_first = clampi(yo[t.y0], 0, S.h - 1), src_last = clampi(yo[t.y_last] + 1, 0, S.h - 1);
            if (i > 0) {
                if (src_first < ty[i - 1][by].y0) return;
                t.src_y0 = ty[i - 1][by].y0;
                need_last = src_last;
            } else {
                t.src_y0 = src_first; t.n_src = src_last - src_first + 1;
                x_rows0 = std::max(x_rows0, t.n_src);
                h_rows = std::max(h_rows, t.n_src);
            }
        }
        for (int i = n - 1; i >= 1; i--) {                                          // the source of stage i is the whole region of stage i - 1
            ty[i][by].n_src = ty[i - 1][by].y_last - ty[i - 1][by].y0 + 1;
            x_rows = std::max(x_rows, ty[i][by].n_src);
            h_rows = std::max(h_rows, ty[i][by].n_src);
        }
    }
    // ---- columns: the largest tile width whose regions fit
    std::vector<std::vector<HsPyrStageX>> best_txs; int best_tbx = 0, best_pitch = 0; size_t best_xbytes = 0;
    for (int tbx = 256; tbx >= 64; tbx -= 4) {
        const int nbx = (LAST.w + tbx - 1) / tbx;
        std::vector<std::vector<HsPyrStageX>> txs(n, std::vector<HsPyrStageX>(nbx));
        bool ok = true; int pitch = 0;
        // first columns of every level's parts, last level first (a part starts at the first source column of the tile's part of the level below it)
        for (int bx = 0; bx < nbx && ok; bx++) {
            int x0 = bx * tbx;
            for (int i = n - 1; i >= 0; i--) {
                txs[i][bx] = HsPyrStageX{};
                txs[i][bx].x0 = x0;
                if (i > 0) x0 = bx == 0 ? 0 : (xtab[first + i][4 * x0] & ~3);
            }
        }
        for (int bx = 0; bx < nbx && ok; bx++)
            for (int i = n - 1; i >= 0; i--) {
                const HsLevel& D = h_lv[first + i];
                HsPyrStageX& t = txs[i][bx];
                t.own_x1 = bx + 1 < nbx ? txs[i][bx + 1].x0 : ((D.w + 3) & ~3);
                if (t.own_x1 < t.x0) ok = false;
            }
        // regions: last level = the tile; the level below it must hold the columns the region's last column reads
        for (int bx = 0; bx < nbx && ok; bx++) {
            int need_last = std::min(bx * tbx + tbx, LAST.w) - 1;                  // last column of the region of stage i that is really read
            for (int i = n - 1; i >= 0; i--) {
                const HsLevel& D = h_lv[first + i]; const HsLevel& S = h_lv[first + i - 1];
                HsPyrStageX& t = txs[i][bx];
                const int reg_last = std::max(need_last, std::min(t.own_x1, D.w) - 1);
                t.ncols = ((std::max(reg_last + 1, t.own_x1) - t.x0) + 3) & ~3;
                if (t.ncols > 256 || t.ncols <= 0) { ok = false; break; }
                const int16_t* xt = xtab[first + i];
                const int src_need_first = xt[4 * t.x0], src_need_last = std::min(xt[4 * std::min(reg_last, D.w - 1)] + 1, S.w - 1);
                if (i > 0) {
                    const HsPyrStageX& u = txs[i - 1][bx];
                    if (src_need_first < u.x0) { ok = false; break; }
                    t.src_x0 = u.x0;
                    // the window of the last active lane (offset of its first column + 12 bytes) must stay inside the LDS row of the region below
                    const int lastlane_col = std::min(t.x0 + t.ncols - 4, D.w - 1);
                    if (xt[4 * lastlane_col] - u.x0 + 12 > FZ_APITCH) { ok = false; break; }
                    need_last = src_need_last;
                } else {
                    t.src_x0 = src_need_first & ~15;
                    t.nvec = ((src_need_last - t.src_x0) >> 4) + 1;
                    pitch = std::max(pitch, t.nvec * 16 + 16);
                    const int lastlane_col = std::min(t.x0 + t.ncols - 4, D.w - 1);
                    if (xt[4 * lastlane_col] - t.src_x0 + 12 > t.nvec * 16 + 16) { ok = false; break; }
                }
            }
        }
        if (!ok) continue;
        pitch = (pitch + 15) & ~15;
        const size_t x_bytes = std::max((size_t)pitch * x_rows0, (size_t)FZ_APITCH * x_rows);
        const size_t lds = ((x_bytes + 15) & ~(size_t)15) + (size_t)h_rows * 256 * 2;
        if (lds > lds_max || h_rows > 127 || x_rows > 127) continue;               // (row records address the sums buffer with 16-bit byte offsets: 512 B per row)
        best_tbx = tbx; best_pitch = pitch; best_xbytes = x_bytes; best_txs = txs;
        break;
    }
    if (best_tbx == 0) return;
    {
        const int tbx = best_tbx, pitch = best_pitch, nbx = (LAST.w + tbx - 1) / tbx;
        const size_t x_bytes = best_xbytes;
        const std::vector<std::vector<HsPyrStageX>>& txs = best_txs;
        // ---- commit
        if (blob.empty()) blob.push_back(0);
        C.sbase = first == 1 ? nullptr : h_lv[first - 1].base; C.s_img_stride = h_lv[first - 1].img_stride; C.spitch = h_lv[first - 1].pitch; C.nstage = n;
        for (int i = 0; i < n; i++) {
            const HsLevel& D = h_lv[first + i];
            HsPyrStage& S = C.st[i];
            S.base = D.base; S.img_stride = D.img_stride; S.pitch = D.pitch; S.w = D.w; S.h = D.h;
            S.xt = reinterpret_cast<const HsXTab*>(D.xofs);
            // row records per y tile: rows y0 .. y_last of the tile's region, offsets relative to the first row the tile's source buffer holds;
            // every slot is padded with copies of its last row (the kernel prefetches one step of <= 8 rows past the end)
            int slot = 0;
            for (int by = 0; by < nby; by++) slot = std::max(slot, ty[i][by].y_last - ty[i][by].y0 + 1);
            slot += 4 * 8;                                                           // (the vertical pass walks 2 NW rows per step and prefetches one step ahead: 4 NW rows past the end at most)
            const size_t orow = blob.size(); blob.resize(orow + (size_t)slot * nby);
            const int sh = h_lv[first + i - 1].h;
            for (int by = 0; by < nby; by++) {
                const HsPyrStageY& t = ty[i][by];
                for (int k = 0; k < slot; k++) {
                    const int dy = std::min(t.y0 + k, t.y_last);
                    HsPyrRow r;
                    r.off0 = (uint16_t)((clampi(yofs[first + i][dy], 0, sh - 1) - t.src_y0) * 512);
                    r.off1 = (uint16_t)((clampi(yofs[first + i][dy] + 1, 0, sh - 1) - t.src_y0) * 512);
                    r.b0 = (uint16_t)ibeta[first + i][2 * dy]; r.b1 = (uint16_t)ibeta[first + i][2 * dy + 1];
                    memcpy(&blob[orow + (size_t)by * slot + k], &r, 8);
                }
            }
            S.rows = reinterpret_cast<const HsPyrRow*>(orow * 8); S.slot = slot;
            const size_t ox = blob.size(); blob.resize(ox + 4 * (size_t)nbx);
            memcpy(&blob[ox], txs[i].data(), 32 * (size_t)nbx);
            const size_t oy = blob.size(); blob.resize(oy + 4 * (size_t)nby);
            memcpy(&blob[oy], ty[i].data(), 32 * (size_t)nby);
            S.tx = reinterpret_cast<const HsPyrStageX*>(ox * 8); S.ty = reinterpret_cast<const HsPyrStageY*>(oy * 8);
        }
        C.tbx = tbx; C.lds_pitch = pitch; C.x_bytes = (int32_t)((x_bytes + 15) & ~(size_t)15); C.h_rows = h_rows; C.grid_x = nbx; C.grid_y = nby; C.valid = 1;
    }
}

// Host side of the table-driven kernel: for every fused pair the tile records (the geometry the kernel used to compute itself, same
// expressions) and for both levels of the pair the row records; everything is appended to `blob`, pointers into it are byte offsets.
void hs_pyramid_build_tables(const HsLevel* h_lv, int nlevels, const int16_t* const* xtab, const int16_t* const* yofs, const int16_t* const* ibeta,
                             std::vector<uint64_t>& blob, std::vector<HsPyrFuse>& fuse)
{
    fuse.assign(nlevels, HsPyrFuse{});
    auto clampi = [](int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); };
    if (blob.empty()) blob.push_back(0);
    for (int l = 1; l + 1 < nlevels; l++) {
        const HsLevel& A = h_lv[l];
        if (A.fuse_tbx <= 0) continue;
        const HsLevel& B = h_lv[l + 1]; const HsLevel& S = h_lv[l - 1];
        const int16_t* xA = xtab[l]; const int16_t* xB = xtab[l + 1]; const int16_t* yA = yofs[l]; const int16_t* yB = yofs[l + 1];
        const int tbx = A.fuse_tbx, nbx = (B.w + tbx - 1) / tbx, nby = (B.h + FZ_ROWS - 1) / FZ_ROWS;
        HsPyrFuse& F = fuse[l];
        F.sbase = l == 1 ? nullptr : S.base; F.s_img_stride = S.img_stride; F.spitch = S.pitch;
        F.abase = A.base; F.a_img_stride = A.img_stride; F.apitch = A.pitch; F.aw = A.w; F.ah = A.h;
        F.bbase = B.base; F.b_img_stride = B.img_stride; F.bpitch = B.pitch; F.bw = B.w; F.bh = B.h;
        F.xtA = reinterpret_cast<const HsXTab*>(A.xofs); F.xtB = reinterpret_cast<const HsXTab*>(B.xofs);
        F.tbx = tbx; F.sr = A.fuse_sr; F.lds_pitch = A.fuse_pitch; F.valid = 1;
        const size_t ox = blob.size(); blob.resize(ox + 4 * (size_t)nbx);
        for (int bx = 0; bx < nbx; bx++) {
            const int bx0 = bx * tbx; const bool last_x = bx0 + tbx >= B.w;
            HsPyrXTile t{};
            t.ax0 = bx == 0 ? 0 : (xB[4 * bx0] & ~3);
            t.own_x1 = last_x ? ((A.w + 3) & ~3) : (xB[4 * (bx0 + tbx)] & ~3);
            const int ax_lastcol = std::min(t.ax0 + 255, A.w - 1);
            t.col0 = xA[4 * t.ax0] & ~15;
            const int col_last = std::min(xA[4 * ax_lastcol] + 1, S.w - 1);
            t.nvec = ((col_last - t.col0) >> 4) + 1;
            memcpy(&blob[ox + 4 * (size_t)bx], &t, 32);
        }
        const size_t oy = blob.size(); blob.resize(oy + 4 * (size_t)nby);
        for (int by = 0; by < nby; by++) {
            const int by0 = by * FZ_ROWS, by_last = std::min(by0 + FZ_ROWS, B.h) - 1; const bool last_y = by0 + FZ_ROWS >= B.h;
            HsPyrYTile t{};
            t.ay0 = by == 0 ? 0 : clampi(yB[by0], 0, A.h - 1);
            t.own_y1 = last_y ? A.h : clampi(yB[by0 + FZ_ROWS], 0, A.h - 1);
            t.ay_last = std::max(clampi(yB[by_last] + 1, 0, A.h - 1), t.own_y1 - 1);
            t.sy_first = clampi(yA[t.ay0], 0, S.h - 1);
            t.n_sr = clampi(yA[t.ay_last] + 1, 0, S.h - 1) - t.sy_first + 1;
            memcpy(&blob[oy + 4 * (size_t)by], &t, 32);
        }
        F.xt = reinterpret_cast<const HsPyrXTile*>(ox * 8); F.yt = reinterpret_cast<const HsPyrYTile*>(oy * 8);
        // row records per y tile (offsets relative to the first row the tile's sums buffer holds), padded with copies of the tile's last row: the
        // kernel prefetches one step of <= 8 rows past the end
        const int slotA = A.fuse_ar + 8, slotB = FZ_ROWS + 8;
        const size_t oa = blob.size(); blob.resize(oa + (size_t)slotA * nby);
        const size_t ob = blob.size(); blob.resize(ob + (size_t)slotB * nby);
        for (int by = 0; by < nby; by++) {
            HsPyrYTile t; memcpy(&t, &blob[oy + 4 * (size_t)by], 32);
            const int by0 = by * FZ_ROWS;
            for (int k = 0; k < slotA; k++) {
                const int ay = std::min(t.ay0 + k, t.ay_last);
                HsPyrRow r;
                r.off0 = (uint16_t)((clampi(yA[ay], 0, S.h - 1) - t.sy_first) * 512); r.off1 = (uint16_t)((clampi(yA[ay] + 1, 0, S.h - 1) - t.sy_first) * 512);
                r.b0 = (uint16_t)ibeta[l][2 * ay]; r.b1 = (uint16_t)ibeta[l][2 * ay + 1];
                memcpy(&blob[oa + (size_t)by * slotA + k], &r, 8);
            }
            for (int k = 0; k < slotB; k++) {
                const int y = std::min(by0 + k, B.h - 1);
                HsPyrRow r;
                r.off0 = (uint16_t)((clampi(yB[y], 0, A.h - 1) - t.ay0) * 512); r.off1 = (uint16_t)((clampi(yB[y] + 1, 0, A.h - 1) - t.ay0) * 512);
                r.b0 = (uint16_t)ibeta[l + 1][2 * y]; r.b1 = (uint16_t)ibeta[l + 1][2 * y + 1];
                memcpy(&blob[ob + (size_t)by * slotB + k], &r, 8);
            }
        }
        F.rowA = reinterpret_cast<const HsPyrRow*>(oa * 8); F.rowB = reinterpret_cast<const HsPyrRow*>(ob * 8); F.slotA = slotA;
    }
}

// Can levels (l, l+1) be fused?  Walks every tile with the host copies of the tables and checks what the kernel assumes: the level-A region of
// a tile (owned part + halo) fits 256 columns / fuse_ar rows, its source rectangle fits the LDS rectangle, all LDS fits.
void hs_pyramid_plan_fusion(HsLevel* h_lv, int nlevels, const int16_t* const* xtab, const int16_t* const* yofs, int tbx_max)
{
    for (int l = 1; l < nlevels; l++) h_lv[l].fuse_tbx = h_lv[l].fuse_ar = h_lv[l].fuse_sr = h_lv[l].fuse_pitch = 0;
    for (int l = 1; l + 1 < nlevels; l += 2) {
        HsLevel& A = h_lv[l]; const HsLevel& B = h_lv[l + 1]; const HsLevel& S = h_lv[l - 1];
        const double scxA = (double)S.w / A.w, scyA = (double)S.h / A.h, scxB = (double)A.w / B.w, scyB = (double)A.h / B.h;
        if (scxA > 2.0 || scxB > 2.0 || scyA > 2.0 || scyB > 2.0 || B.w < 8 || B.h < 1) continue;
        int tbx = std::min(256, ((int)(250.0 / scxB)) & ~3);
        if (tbx_max >= 64) tbx = std::min(tbx, tbx_max & ~3);      // tuning / experiment knob (HS_PYRAMID_TBX_MAX): narrower level-B tiles
        if (tbx < 64) continue;
        const int16_t* xA = xtab[l]; const int16_t* xB = xtab[l + 1]; const int16_t* yA = yofs[l]; const int16_t* yB = yofs[l + 1];
        auto clampi = [](int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); };
        int ar = 0, sr = 0, pitch = 0; bool ok = true;
        for (int by0 = 0; by0 < B.h && ok; by0 += FZ_ROWS) {
            const int by_last = std::min(by0 + FZ_ROWS, B.h) - 1; const bool last_y = by0 + FZ_ROWS >= B.h;
            const int ay0 = by0 == 0 ? 0 : clampi(yB[by0], 0, A.h - 1);
            const int own_y1 = last_y ? A.h : clampi(yB[by0 + FZ_ROWS], 0, A.h - 1);
            const int ay_last = std::max(clampi(yB[by_last] + 1, 0, A.h - 1), own_y1 - 1);
            if (own_y1 <= ay0 && !last_y) { /* empty owned range is fine */ }
            if (clampi(yB[by0], 0, A.h - 1) < ay0) ok = false;                    // rows the tile needs lie at or below its first row
            ar = std::max(ar, ay_last - ay0 + 1);
            const int sy_first = clampi(yA[ay0], 0, S.h - 1), sy_last = clampi(yA[ay_last] + 1, 0, S.h - 1);
            sr = std::max(sr, sy_last - sy_first + 1);
        }
        for (int bx0 = 0; bx0 < B.w && ok; bx0 += tbx) {
            const int bx_last = std::min(bx0 + tbx, B.w) - 1; const bool last_x = bx0 + tbx >= B.w;
            const int ax0 = bx0 == 0 ? 0 : (xB[4 * bx0] & ~3);
            const int own_x1 = last_x ? ((A.w + 3) & ~3) : (xB[4 * (bx0 + tbx)] & ~3);
            const int need_last = std::min(xB[4 * bx_last] + 1, A.w - 1);
            if (xB[4 * bx0] < ax0 || own_x1 - ax0 > 256 || need_last - ax0 > 255 || own_x1 < ax0) ok = false;
            // window of the last lane: offset of its first column + 11 bytes must stay inside the LDS row
            if (xB[4 * std::min(bx0 + tbx - 4, B.w - 1)] - ax0 + 12 > FZ_APITCH) ok = false;
            const int ax_lastcol = std::min(ax0 + 255, A.w - 1);
            const int col0 = xA[4 * ax0] & ~15, col_last = std::min(xA[4 * ax_lastcol] + 1, S.w - 1);
            pitch = std::max(pitch, ((col_last - col0) >> 4) * 16 + 16 + 16);
            if (xA[4 * ax0] < col0) ok = false;
        }
        if (!ok) continue;
        pitch = (pitch + 15) & ~15;
        const size_t lds = (size_t)pitch * sr + (size_t)std::max(sr, ar) * 256 * 2;
        if ((size_t)FZ_APITCH * ar > (size_t)pitch * sr || lds > 60 * 1024) continue;
        A.fuse_tbx = tbx; A.fuse_ar = ar; A.fuse_sr = sr; A.fuse_pitch = pitch;
    }
}

// launches with at most this many workgroups per CU use 8 waves per workgroup (HS_PYRAMID_NW8 = threshold; 0 = never; tuning / parity knob, read once)
static int pyr_nw8_wg_per_cu()
{
    static int v = [] { const char* e = getenv("HS_PYRAMID_NW8"); return e ? atoi(e) : 3; }();
    return v;
}
int hs_launch_pyramid(const HsLevel* d_lv, const HsLevel* h_lv, const HsPyrFuse* fuse, const HsPyrChain* chain, int nlevels, HsImg0 img0, int batch, hipStream_t s, const HsPyrChain* deep)
{
    // the deep chains of small batches may want more than the default 64 KB of dynamic LDS (gfx950: 160 KB per CU).  Function attributes are PER DEVICE:
    // the raise is done once for every device a launch sequence is enqueued on (a handle on a second GPU of the process gets its own), and a device
    // where it failed only ever runs deep chains of <= 64 KB (the standard plan otherwise)
    const bool big_lds = deep != nullptr && [] {
        int dev = -1;
        if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) { (void)hipGetLastError(); return false; }
        static std::atomic<int8_t> state[64];       // 0 = not tried on this device, 1 = raised, 2 = failed
        const int8_t st = state[dev].load(std::memory_order_acquire);
        if (st) return st == 1;
        const bool ok = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_resize_chain<8>), hipFuncAttributeMaxDynamicSharedMemorySize, HS_PYR_DEEP_LDS) == hipSuccess &&
                        hipFuncSetAttribute(reinterpret_cast<const void*>(&k_resize_chain<4>), hipFuncAttributeMaxDynamicSharedMemorySize, HS_PYR_DEEP_LDS) == hipSuccess;
        if (!ok) (void)hipGetLastError();
        state[dev].store(ok ? 1 : 2, std::memory_order_release);
        return ok;
    }();
    int n_launches = 0;      // what this call really enqueued (the plan can fall back per level: caller-frame alignment, LDS attribute refused)
    for (int l = 1; l < nlevels; l++, n_launches++) {
        const HsLevel& D = h_lv[l];
        if (deep && deep[l].valid && l + deep[l].nstage <= nlevels) {
            bool vec16 = true;
            if (l == 1) vec16 = (((uintptr_t)img0.base | (uintptr_t)img0.base2 | img0.row_stride | img0.img_stride) & 15) == 0 && ((h_lv[0].w + 15) & ~15) <= (int)img0.row_stride;
            const HsPyrChain& C = deep[l];
            const size_t lds = (size_t)C.x_bytes + (size_t)C.h_rows * 256 * 2;
            if (vec16 && (lds <= 64 * 1024 || big_lds)) {
                dim3 grid(C.grid_x, C.grid_y, batch);
                // (measured at one 1080p pair per call, levels 1-7 in one launch of 304 workgroups: 27.6 us with 8 waves per workgroup, 28.8 us with 16 — the
                //  stages are a dependent sequence inside the workgroup, more waves per stage do not shorten it; three launches of the standard plan: 28.6 us)
                if ((size_t)grid.x * grid.y * grid.z <= (size_t)256 * pyr_nw8_wg_per_cu()) hipLaunchKernelGGL(k_resize_chain<8>, grid, dim3(512), lds, s, C, img0);
                else hipLaunchKernelGGL(k_resize_chain<4>, grid, dim3(256), lds, s, C, img0);
                l += C.nstage - 1;
                continue;
            }
        }
        if (chain && D.chain_n > 0 && chain[l].valid && l + D.chain_n <= nlevels) {
            bool vec16 = true;
            if (l == 1) vec16 = (((uintptr_t)img0.base | (uintptr_t)img0.base2 | img0.row_stride | img0.img_stride) & 15) == 0 && ((h_lv[0].w + 15) & ~15) <= (int)img0.row_stride;
            if (vec16) {
                const HsPyrChain& C = chain[l];
                const size_t lds = (size_t)C.x_bytes + (size_t)C.h_rows * 256 * 2;
                dim3 grid(C.grid_x, C.grid_y, batch);
                if ((size_t)grid.x * grid.y * grid.z <= (size_t)256 * pyr_nw8_wg_per_cu()) hipLaunchKernelGGL(k_resize_chain<8>, grid, dim3(512), lds, s, C, img0);
                else hipLaunchKernelGGL(k_resize_chain<4>, grid, dim3(256), lds, s, C, img0);
                l += C.nstage - 1;
                continue;
            }
        }
        if (D.fuse_tbx > 0 && l + 1 < nlevels && fuse && fuse[l].valid) {
            // 16-byte source vectors: always fine for our own levels (pitch % 64 == 0), checked for the caller's frames
            bool vec16 = true;
            if (l == 1) vec16 = (((uintptr_t)img0.base | (uintptr_t)img0.base2 | img0.row_stride | img0.img_stride) & 15) == 0 && ((h_lv[0].w + 15) & ~15) <= (int)img0.row_stride;
            if (vec16) {
                const HsLevel& B = h_lv[l + 1];
                const size_t lds = (size_t)D.fuse_pitch * D.fuse_sr + (size_t)std::max(D.fuse_sr, D.fuse_ar) * 256 * 2;
                dim3 grid((B.w + D.fuse_tbx - 1) / D.fuse_tbx, (B.h + FZ_ROWS - 1) / FZ_ROWS, batch);
                // few workgroups per CU (small batches): 8 waves per workgroup shorten the workgroup's life, which is the launch's duration then
                if ((size_t)grid.x * grid.y * grid.z <= (size_t)256 * pyr_nw8_wg_per_cu()) hipLaunchKernelGGL(k_resize_two_levels<8>, grid, dim3(512), lds, s, fuse[l], img0);
                else hipLaunchKernelGGL(k_resize_two_levels<4>, grid, dim3(256), lds, s, fuse[l], img0);
                l++;                                           // level l+1 is done too
                continue;
            }
        }
        const int sw = h_lv[l - 1].w, sh = h_lv[l - 1].h;
        // 16-byte source vectors: always fine for our own levels (pitch % 64 == 0), checked for the caller's frames;
        // a vector may run past the last source column only inside the row's own pitch
        bool vec16 = true;
        if (l == 1) vec16 = (((uintptr_t)img0.base | (uintptr_t)img0.base2 | img0.row_stride | img0.img_stride) & 15) == 0 && ((sw + 15) & ~15) <= (int)img0.row_stride;
        // worst-case source rectangle of a 256 x LT_ROWS tile
        const double scx = (double)sw / D.w, scy = (double)sh / D.h;
        const int lds_pitch = (((int)(256 * scx) + 2 + 15 + 15) & ~15) + 16;
        const int lds_rows = (int)(LT_ROWS * scy) + 4;
        const size_t lds_bytes = (size_t)lds_pitch * lds_rows + (size_t)lds_rows * 256 * 2;      // source rectangle + 16-bit horizontal sums
        if (vec16 && lds_bytes <= 60 * 1024 && scx <= 2.0) {
            dim3 grid((D.w + 255) / 256, (D.h + LT_ROWS - 1) / LT_ROWS, batch);
            hipLaunchKernelGGL(k_resize_level_lds, grid, dim3(256), lds_bytes, s, d_lv, l, img0, lds_pitch, lds_rows);
            continue;
        }
        dim3 block(64, 4, 1);
        dim3 grid((D.w + 255) / 256, (D.h + 3) / 4, batch);
        // dword source fetches need 4-byte aligned rows
        bool aligned = true;
        if (l == 1) aligned = (((uintptr_t)img0.base | (uintptr_t)img0.base2 | img0.row_stride | img0.img_stride) & 3) == 0;
        if (aligned) hipLaunchKernelGGL(k_resize_level<true>, grid, block, 0, s, d_lv, l, img0);
        else hipLaunchKernelGGL(k_resize_level<false>, grid, block, 0, s, d_lv, l, img0);
    }
    return n_launches;
}

// launches of one hs_launch_pyramid call when every eligible pair is fused (the caller-frame alignment fallback adds one)
int hs_pyramid_launch_count(const HsLevel* h_lv, int nlevels)
{
    int n = 0;
    for (int l = 1; l < nlevels; l++) { n++; if (h_lv[l].chain_n > 0 && l + h_lv[l].chain_n <= nlevels) l += h_lv[l].chain_n - 1; else if (h_lv[l].fuse_tbx > 0 && l + 1 < nlevels) l++; }
    return n;
}

// ---- calibration kernels of known HBM traffic (hs_debug_stream_copy): one dword / one 16-byte vector per lane, grid-stride
__global__ __launch_bounds__(256) void k_copy_u32(uint32_t* __restrict__ d, const uint32_t* __restrict__ s, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) d[i] = s[i];
}
__global__ __launch_bounds__(256) void k_copy_u128(uint4* __restrict__ d, const uint4* __restrict__ s, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) d[i] = s[i];
}
// four 16-byte vectors per lane in flight (each wave-instruction still covers 1 KiB of contiguous bytes): what a streaming copy needs to approach the
// memory system's rate — the figure bench.py prints as the measured copy peak
__global__ __launch_bounds__(256) void k_copy_u128x4(uint4* __restrict__ d_, const uint4* __restrict__ s_, size_t n)
{
    hs_u32x4* const d = reinterpret_cast<hs_u32x4*>(d_);
    const hs_u32x4* const s = reinterpret_cast<const hs_u32x4*>(s_);
    const size_t stride = (size_t)gridDim.x * 1024;
    size_t i = (size_t)blockIdx.x * 1024 + threadIdx.x;
    for (; i + 768 < n; i += stride) {
        const hs_u32x4 a = __builtin_nontemporal_load(s + i), b = __builtin_nontemporal_load(s + i + 256), c = __builtin_nontemporal_load(s + i + 512), e = __builtin_nontemporal_load(s + i + 768);
        __builtin_nontemporal_store(a, d + i); __builtin_nontemporal_store(b, d + i + 256); __builtin_nontemporal_store(c, d + i + 512); __builtin_nontemporal_store(e, d + i + 768);
    }
    for (int k = 0; k < 4; k++) if (i + 256 * k < n) d[i + 256 * k] = s[i + 256 * k];      // the ragged tail of the last pass
}
void hs_launch_stream_copy(void* d_dst, const void* d_src, size_t bytes, int width, hipStream_t s)
{
    if (width == 64) { hipLaunchKernelGGL(k_copy_u128x4, dim3(4096), dim3(256), 0, s, (uint4*)d_dst, (const uint4*)d_src, bytes / 16); return; }
    if (width == 4) hipLaunchKernelGGL(k_copy_u32, dim3(2048), dim3(256), 0, s, (uint32_t*)d_dst, (const uint32_t*)d_src, bytes / 4);
    else hipLaunchKernelGGL(k_copy_u128, dim3(2048), dim3(256), 0, s, (uint4*)d_dst, (const uint4*)d_src, bytes / 16);
}
