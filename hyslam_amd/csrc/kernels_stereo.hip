// kernels_stereo.hip — K8a: left<->right stereo matching by Hamming distance.
//
// Replaces Stereomatcher::computeStereoMatches (src/features/Stereomatcher.cpp:36-156) and the
// ORBDistance bit-hack it calls per pair (src/features/low_level/DescriptorDistance.cpp:9-25).
// The reference builds a per-row table of right keypoints (iR listed in rows floor(y-r)..ceil(y+r),
// r = 2*size/size_ref) and scans vRowIndices[(int)vL] in ascending iR.  Here the right keypoints are binned once per pair into 32-row
// STRIPS (k_stereo_strips: the reference's table at 1/32 of its size); one wavefront owns one left keypoint, scans the strip of its row
// (~100 candidates instead of all 2000) and re-applies the exact predicate (row band, |octave diff| <= 1, uL-maxD <= uR <= uL) to every
// candidate.  The running minimum uses the key dist<<16 | iR, which reproduces the reference's strict `dist < bestDist` over ascending iR
// (first minimum wins) whatever the order inside a strip.  Descriptors are XORed as 4 x u64 and counted with v_bcnt (__popcll) — equal
// to the reference's 32-bit parallel bit count.
//
// Second kernel: the reference's sort + median + 2.1*median rejection (:142-155) is a histogram of the
// accepted integer distances, one workgroup per pair.
// Documented deviation D2 (reference UB): rows outside [0,nRows) are ignored, no match => no filtering.
#include "hs_internal.h"

// Right keypoints binned into 32-row strips once per pair: a right keypoint lists itself in every strip its row band
// [floor(y-r), ceil(y+r)] touches (1-2 strips), so a left keypoint at row v only scans strip v>>5 — ~100 candidates instead of all
// 2000 (the reference's per-row table, Stereomatcher.cpp:46-63, at 1/32 of its size).  The exact band test is repeated per candidate.
#define STRIP_SHIFT HS_STRIP_SHIFT

// One workgroup per pair: the strip counters live in LDS (a slot is an LDS atomic away, not a round trip to L2), every right keypoint's
// record is loaded up front, and the counters are written out whole, so nothing has to be zeroed between calls.
#define STRIPS_T 1024
#define STRIPS_MAX HS_STRIPS_MAX   // 65536 rows / 32 (HsStripEntry: hs_internal.h)
__global__ __launch_bounds__(STRIPS_T) void k_stereo_strips(const hs_keypoint* __restrict__ kpsR, const int32_t* __restrict__ nRs, int cap,
                                                            float size_ref, int n_rows, int n_strips,
                                                            int32_t* __restrict__ strip_count, HsStripEntry* __restrict__ strip_list)
{
    __shared__ int s_cnt[STRIPS_MAX];
    const int pair = blockIdx.x, tid = threadIdx.x;
    const int nR = min(nRs[pair], cap);
    for (int s = tid; s < n_strips; s += STRIPS_T) s_cnt[s] = 0;
    constexpr int KPT = 4;                                    // keypoints per thread whose records are in flight together
    __syncthreads();
    for (int i0 = 0; i0 < nR; i0 += STRIPS_T * KPT) {
        float kx[KPT], ky[KPT], ks[KPT]; int ko[KPT];
#pragma unroll
        for (int k = 0; k < KPT; k++) {
            const int i = min(i0 + tid + STRIPS_T * k, nR - 1);
            const hs_keypoint* kr = &kpsR[(size_t)pair * cap + i];
            kx[k] = kr->x; ky[k] = kr->y; ks[k] = kr->size; ko[k] = kr->octave;
        }
#pragma unroll
        for (int k = 0; k < KPT; k++) {
            const int i = i0 + tid + STRIPS_T * k;
            if (i >= nR) continue;
            const float r = 2.0f * ks[k] / size_ref;         // :56
            int maxr = (int)ceilf(ky[k] + r), minr = (int)floorf(ky[k] - r);
            if (maxr < 0 || minr >= n_rows) continue;         // rows outside [0, nRows) do not exist (D2)
            minr = max(minr, 0); maxr = min(maxr, n_rows - 1);
            for (int s = minr >> STRIP_SHIFT; s <= (maxr >> STRIP_SHIFT); s++) {
                const int slot = atomicAdd(&s_cnt[s], 1);
                const int lo = max(minr - (s << STRIP_SHIFT), 0), hi = min(maxr - (s << STRIP_SHIFT), 31);
                HsStripEntry e; e.uR = kx[k]; e.octave = ko[k]; e.idx_band = (uint32_t)i | ((uint32_t)lo << 16) | ((uint32_t)hi << 24); e._pad = 0;
                strip_list[((size_t)pair * n_strips + s) * cap + slot] = e;
            }
        }
    }
    __syncthreads();
    for (int s = tid; s < n_strips; s += STRIPS_T) strip_count[(size_t)pair * n_strips + s] = s_cnt[s];
}

// One wavefront owns TWO left keypoints and walks both strips together (the kernel is a chain of dependent loads — strip entry, then the
// descriptors of the candidates that pass — and nothing else: twice the loads in flight per wave, half the waves).
template <int SM_KPW>              // left keypoints per wavefront: 2, or 1 when the launch has too few workgroups to fill the chip anyway (single pairs)
__global__ __launch_bounds__(256) void k_stereo_match(const int32_t* __restrict__ strip_count, const HsStripEntry* __restrict__ strip_list, int n_strips,
                                                      const hs_keypoint* __restrict__ kpsL, const uint8_t* __restrict__ descL,
                                                      const int32_t* __restrict__ nLs,
                                                      const hs_keypoint* __restrict__ kpsR, const uint8_t* __restrict__ descR,
                                                      const int32_t* __restrict__ nRs,
                                                      int cap, hs_stereo_params sp,
                                                      float* __restrict__ uRight, float* __restrict__ depth, int32_t* __restrict__ best_dist)
{
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // uniform: the left keypoints' records come through scalar loads
    const int pair = blockIdx.y;
    const int iL0 = (blockIdx.x * 4 + wv) * SM_KPW;
    const int nL = min(nLs[pair], cap);
    if (iL0 >= cap) return;
    const size_t o = (size_t)pair * cap;
    const float mbf = sp.mbf, mb = sp.mbf / sp.fx;
    const float minD = 0.f, maxD = mbf / mb;                 // :66-68
    const float th_high = sp.th_high;

    float uL[SM_KPW], minU[SM_KPW], maxU[SM_KPW]; int levelL[SM_KPW], rowIn[SM_KPW], nc[SM_KPW];
    const HsStripEntry* cl[SM_KPW]; const unsigned long long* dl[SM_KPW];
    bool live[SM_KPW];
    int ncmax = 0;
#pragma unroll
    for (int t = 0; t < SM_KPW; t++) {
        const int iL = iL0 + t;
        live[t] = iL < nL;
        const hs_keypoint kl = kpsL[o + min(iL, max(nL - 1, 0))];
        const float vL = kl.y;
        uL[t] = kl.x; levelL[t] = kl.octave;
        minU[t] = uL[t] - maxD; maxU[t] = uL[t] - minD;
        const int rowL = (int)vL;                             // vRowIndices[vL]
        const bool ok = live[t] && (vL >= 0.f) && rowL < sp.n_rows && !(maxU[t] < 0.f);
        const size_t sb = (size_t)pair * n_strips + (ok ? (rowL >> STRIP_SHIFT) : 0);
        nc[t] = ok ? strip_count[sb] : 0;
        cl[t] = strip_list + sb * cap;
        rowIn[t] = rowL & 31;
        dl[t] = reinterpret_cast<const unsigned long long*>(descL + (o + min(iL, max(nL - 1, 0))) * 32);
        ncmax = max(ncmax, nc[t]);
    }
    // bestDist starts at TH_HIGH and only strictly smaller distances replace it (:92,114)
    uint32_t best[SM_KPW];
#pragma unroll
    for (int t = 0; t < SM_KPW; t++) best[t] = 0xFFFFFFFFu;
    // two candidates per lane, keypoint and round (a strip holds ~100).  Lanes without a candidate re-read the strip's first entry and are masked.
    for (int c0 = 0; c0 < ncmax; c0 += 128) {
        HsStripEntry e[SM_KPW][2]; bool pass[SM_KPW][2];
#pragma unroll
        for (int t = 0; t < SM_KPW; t++)
#pragma unroll
            for (int j = 0; j < 2; j++) {
                const int c = c0 + 64 * j + lane;
                pass[t][j] = c < nc[t];
                e[t][j] = cl[t][pass[t][j] ? c : 0];
            }
#pragma unroll
        for (int t = 0; t < SM_KPW; t++)
#pragma unroll
            for (int j = 0; j < 2; j++) {
                const int lo = (int)((e[t][j].idx_band >> 16) & 0xFFu), hi = (int)(e[t][j].idx_band >> 24);
                pass[t][j] = pass[t][j] && !(rowIn[t] < lo || rowIn[t] > hi);
                pass[t][j] = pass[t][j] && !(e[t][j].octave < levelL[t] - 1 || e[t][j].octave > levelL[t] + 1);
                pass[t][j] = pass[t][j] && (e[t][j].uR >= minU[t] && e[t][j].uR <= maxU[t]);
            }
        unsigned long long dr[SM_KPW][2][4];                  // descriptors only of the candidates that passed (a fifth of them)
#pragma unroll
        for (int t = 0; t < SM_KPW; t++)
#pragma unroll
            for (int j = 0; j < 2; j++) {
                const unsigned long long* p = reinterpret_cast<const unsigned long long*>(descR + (o + (e[t][j].idx_band & 0xFFFFu)) * 32);
                if (pass[t][j]) { dr[t][j][0] = p[0]; dr[t][j][1] = p[1]; dr[t][j][2] = p[2]; dr[t][j][3] = p[3]; }
                else { dr[t][j][0] = dr[t][j][1] = dr[t][j][2] = dr[t][j][3] = 0ull; }
            }
#pragma unroll
        for (int t = 0; t < SM_KPW; t++) {
            const unsigned long long l0 = dl[t][0], l1 = dl[t][1], l2 = dl[t][2], l3 = dl[t][3];      // uniform: scalar loads
#pragma unroll
            for (int j = 0; j < 2; j++) {
                const int d = __popcll(l0 ^ dr[t][j][0]) + __popcll(l1 ^ dr[t][j][1]) + __popcll(l2 ^ dr[t][j][2]) + __popcll(l3 ^ dr[t][j][3]);
                if (pass[t][j] && (float)d < th_high) best[t] = min(best[t], ((uint32_t)d << 16) | (e[t][j].idx_band & 0xFFFFu));
            }
        }
    }
#pragma unroll
    for (int t = 0; t < SM_KPW; t++) {
#pragma unroll
        for (int s = 32; s > 0; s >>= 1) best[t] = min(best[t], (uint32_t)__shfl_xor((int)best[t], s, 64));
        const int iL = iL0 + t;
        if (iL >= cap) break;
        if (lane == 0) {
            float ur_out = -1.0f, depth_out = -1.0f; int bd = -1;
            if (live[t] && best[t] != 0xFFFFFFFFu) {
                const float bestDist = (float)(best[t] >> 16);
                const int bestIdxR = best[t] & 0xFFFF;
                const float dist_threshold = (sp.th_high + sp.th_low) / 2;          // :41
                if (bestDist < dist_threshold) {
                    float uR0 = kpsR[o + bestIdxR].x;
                    float disparity = uL[t] - uR0;
                    if (disparity >= minD && disparity < maxD) {
                        if (disparity <= 0) { disparity = 0.01; uR0 = uL[t] - 0.01; }    // double constants, as in the reference (:130-131)
                        depth_out = mbf / disparity;
                        ur_out = uR0;
                        bd = (int)(best[t] >> 16);
                    }
                }
            }
            uRight[o + iL] = ur_out; depth[o + iL] = depth_out; best_dist[o + iL] = bd;
        }
    }
}

__global__ __launch_bounds__(256) void k_stereo_median(const int32_t* __restrict__ nLs, int cap,
                                                       float* __restrict__ uRight, float* __restrict__ depth,
                                                       const int32_t* __restrict__ best_dist,
                                                       int32_t* __restrict__ strip_count, int n_strips)
{
    __shared__ int hist[257];
    __shared__ float s_th;
    const int pair = blockIdx.x;
    const int nL = min(nLs[pair], cap);
    const size_t o = (size_t)pair * cap;
    for (int i = threadIdx.x; i < 257; i += 256) hist[i] = 0;
    __syncthreads();
    // the distances stay in registers between the histogram and the rejection sweep (8 per thread cover cap <= 2048; more loop again below)
    constexpr int DPT = 8;
    int dreg[DPT];
#pragma unroll
    for (int k = 0; k < DPT; k++) { const int i = threadIdx.x + 256 * k; dreg[k] = i < nL ? best_dist[o + i] : -1; }
#pragma unroll
    for (int k = 0; k < DPT; k++) if (dreg[k] >= 0) atomicAdd(&hist[min(dreg[k], 256)], 1);
    for (int i = threadIdx.x + 256 * DPT; i < nL; i += 256) {
        int d = best_dist[o + i];
        if (d >= 0) atomicAdd(&hist[min(d, 256)], 1);
    }
    __syncthreads();
    if (threadIdx.x < 64) {
        // sorted (dist, iL) pairs: element total/2 carries the median distance (:142-144) = the first bin at which the running count exceeds
        // total/2.  One wavefront: four bins per lane (+ bin 256), inclusive scan over the lanes, the crossing lane finishes inside its bins.
        const int lane = threadIdx.x;
        const int b0 = hist[4 * lane], b1 = hist[4 * lane + 1], b2 = hist[4 * lane + 2], b3 = hist[4 * lane + 3];
        int incl = b0 + b1 + b2 + b3;
#pragma unroll
        for (int s = 1; s < 64; s <<= 1) { const int v = __shfl_up(incl, s, 64); if (lane >= s) incl += v; }
        const int total = __shfl(incl, 63, 64) + hist[256];
        const int target = total / 2, excl = incl - (b0 + b1 + b2 + b3);
        int med = -1;
        if (excl <= target && incl > target) med = 4 * lane + (excl + b0 > target ? 0 : excl + b0 + b1 > target ? 1 : excl + b0 + b1 + b2 > target ? 2 : 3);
        const unsigned long long hit = __ballot(med >= 0);
        if (total == 0) { if (lane == 0) s_th = -1.f; }
        else if (hit == 0) { if (lane == 0) s_th = 1.5f * 1.4f * 256.f; }            // the crossing is in bin 256
        else if (med >= 0) s_th = 1.5f * 1.4f * (float)med;                          // exactly one lane
    }
    __syncthreads();
    const float th = s_th;
    if (th < 0.f) return;
#pragma unroll
    for (int k = 0; k < DPT; k++) {
        const int i = threadIdx.x + 256 * k, d = dreg[k];
        if (d >= 0 && !((float)d < th)) { uRight[o + i] = -1.f; depth[o + i] = -1.f; }      // :146-155
    }
    for (int i = threadIdx.x + 256 * DPT; i < nL; i += 256) {
        int d = best_dist[o + i];
        if (d >= 0 && !((float)d < th)) { uRight[o + i] = -1.f; depth[o + i] = -1.f; }      // :146-155
    }
}

void hs_launch_stereo(const hs_keypoint* kpsL, const uint8_t* descL, const int32_t* nL,
                      const hs_keypoint* kpsR, const uint8_t* descR, const int32_t* nR,
                      int pairs, int cap, hs_stereo_params sp, float* uRight, float* depth, int32_t* best_dist,
                      int32_t* strip_count, void* strip_list, hipStream_t s)
{
    if (pairs <= 0 || cap <= 0) return;
    HsStripEntry* const sl = reinterpret_cast<HsStripEntry*>(strip_list);
    const int n_strips = hs_stereo_strips(sp.n_rows);
    hipLaunchKernelGGL(k_stereo_strips, dim3(pairs), dim3(STRIPS_T), 0, s, kpsR, nR, cap, sp.size_ref, sp.n_rows, n_strips, strip_count, sl);
    if ((size_t)pairs * cap >= (size_t)8 * 2048) {
        dim3 grid((cap + 7) / 8, pairs, 1);
        hipLaunchKernelGGL(k_stereo_match<2>, grid, dim3(256), 0, s, strip_count, sl, n_strips, kpsL, descL, nL, kpsR, descR, nR, cap, sp, uRight, depth, best_dist);
    } else {
        dim3 grid((cap + 3) / 4, pairs, 1);
        hipLaunchKernelGGL(k_stereo_match<1>, grid, dim3(256), 0, s, strip_count, sl, n_strips, kpsL, descL, nL, kpsR, descR, nR, cap, sp, uRight, depth, best_dist);
    }
}

void hs_launch_stereo_median(const int32_t* nL, int pairs, int cap, float* uRight, float* depth, const int32_t* best_dist,
                             int32_t* strip_count, int n_rows, hipStream_t s)
{
    if (pairs <= 0 || cap <= 0) return;
    hipLaunchKernelGGL(k_stereo_median, dim3(pairs), dim3(256), 0, s, nL, cap, uRight, depth, best_dist, strip_count, hs_stereo_strips(n_rows));
}

// the matcher alone, on strips that the describe launch of the stereo front end has already binned (HsStripFuse)
void hs_launch_stereo_match_only(const hs_keypoint* kpsL, const uint8_t* descL, const int32_t* nL,
                                 const hs_keypoint* kpsR, const uint8_t* descR, const int32_t* nR,
                                 int pairs, int cap, hs_stereo_params sp, float* uRight, float* depth, int32_t* best_dist,
                                 const int32_t* strip_count, const void* strip_list, hipStream_t s)
{
    if (pairs <= 0 || cap <= 0) return;
    const HsStripEntry* const sl = reinterpret_cast<const HsStripEntry*>(strip_list);
    const int n_strips = hs_stereo_strips(sp.n_rows);
    if ((size_t)pairs * cap >= (size_t)8 * 2048) {
        dim3 grid((cap + 7) / 8, pairs, 1);
        hipLaunchKernelGGL(k_stereo_match<2>, grid, dim3(256), 0, s, strip_count, sl, n_strips, kpsL, descL, nL, kpsR, descR, nR, cap, sp, uRight, depth, best_dist);
    } else {
        dim3 grid((cap + 3) / 4, pairs, 1);
        hipLaunchKernelGGL(k_stereo_match<1>, grid, dim3(256), 0, s, strip_count, sl, n_strips, kpsL, descL, nL, kpsR, descR, nR, cap, sp, uRight, depth, best_dist);
    }
}
