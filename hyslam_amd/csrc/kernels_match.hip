// kernels_match.hip — K8b/K8c/K8d: the Hamming matchers of HYSLAM::FeatureMatcher on flat arrays.
//
//   k_frame_grid          Frame::AssignFeaturesToGrid / PosInGrid          src/core/Frame.cc:137-153,459-469
//   k_search_projection   FeatureMatcher::_SearchByProjection_             src/features/FeatureMatcher.cc:57-121
//                         + Frame::ProjectLandMark / Camera::Project       src/core/Frame.cc:170-180, src/core/Camera.cpp:116-153
//                         + landMarkSizePixels, GetFeaturesInAreaNEW       src/core/Frame.cc:296-317,416-457
//                         + the view criteria                              src/features/MatchCriteria.cpp:113-360
//   k_rotation_filter     RotationConsistency + ComputeThreeMaxima         src/features/MatchCriteria.cpp:684-767
//   k_bow_match           BestMatchBoWCriterion inside _SearchByBoW_       src/features/FeatureMatcher.cc:281-345, MatchCriteria.cpp:601-635
//   k_knn2                brute-force Hamming 2-NN (no reference call site; config 5)
//
// One wavefront owns one landmark (or one query descriptor).  The reference walks the 64x48 frame grid and filters
// the few keypoints in range; here every lane tests keypoints against the SAME predicate (cell range of the reference's
// query, |dx|<r, |dy|<r, then the view criteria): 2000-3000 keypoints per landmark are 32-47 lane iterations, L2-resident.
// The reference's first-minimum-wins over its candidate order (grid column, then row, then insertion order) is the key
//   dist<<32 | cellx<<22 | celly<<16 | keypoint index;
// the second-best distance is order independent (second smallest of the multiset).
// cv::Mat products in the reference are OpenCV gemm calls (float data, double accumulation, one rounding): restated in fp64.
#include "hs_internal.h"
#include <cfloat>
#include <cstddef>

#define GRID_COLS 64   // FRAME_GRID_COLS, src/core/Frame.h
#define GRID_ROWS 48

struct HsFrameDev {            // hs_frame_view with device pointers
    float Rcw[9], tcw[3], Ow[3];
    float fx, fy, cx, cy, mbf;
    int32_t sensor;
    float min_x, max_x, min_y, max_y, size_ref;
    int32_t n;
    const hs_keypoint* kps; const uint8_t* desc; const float* uR; const int32_t* kp_lm_obs;
    const int8_t* cell;        // [n][2] grid cell of each keypoint, -1 = outside
    const int32_t* cell_start; // [GRID_ROWS*GRID_COLS + 1] first entry of cell cy*GRID_COLS + cx in cell_items (nullptr: no cell lists)
    const uint16_t* cell_items;// [n] keypoint indices cell by cell
};

__global__ void k_frame_grid(HsFrameDev F, int8_t* __restrict__ cell)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= F.n) return;
    const float invW = (float)GRID_COLS / (F.max_x - F.min_x), invH = (float)GRID_ROWS / (F.max_y - F.min_y);
    int px = (int)roundf((F.kps[i].x - F.min_x) * invW);
    int py = (int)roundf((F.kps[i].y - F.min_y) * invH);
    bool ok = !(px < 0 || px >= GRID_COLS || py < 0 || py >= GRID_ROWS);
    cell[2 * i] = ok ? (int8_t)px : (int8_t)-1;
    cell[2 * i + 1] = ok ? (int8_t)py : (int8_t)-1;
}

// Cell lists of the frame grid (the reference's mGrid[col][row] vectors, Frame.cc:137-153), rows of cells contiguous: one workgroup
// counts, scans and scatters.  The order inside a cell is arbitrary: the matcher's candidate key carries the keypoint index.
__global__ __launch_bounds__(1024) void k_frame_grid_lists(int n, const int8_t* __restrict__ cell, int32_t* __restrict__ cell_start, uint16_t* __restrict__ cell_items)
{
    constexpr int NC = GRID_ROWS * GRID_COLS, PER = (NC + 1023) / 1024;
    __shared__ uint32_t cnt[NC];
    __shared__ uint32_t s_wave[16];
    const int tid = threadIdx.x;
    for (int c = tid; c < NC; c += 1024) cnt[c] = 0;
    __syncthreads();
    for (int i = tid; i < n; i += 1024) { const int cx = cell[2 * i], cy = cell[2 * i + 1]; if (cx >= 0) atomicAdd(&cnt[cy * GRID_COLS + cx], 1u); }
    __syncthreads();
    uint32_t loc[PER], sum = 0;
#pragma unroll
    for (int k = 0; k < PER; k++) { const int c = tid * PER + k; loc[k] = c < NC ? cnt[c] : 0; sum += loc[k]; }
    uint32_t incl = sum;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const uint32_t v = __shfl_up(incl, o, 64); if ((tid & 63) >= o) incl += v; }
    if ((tid & 63) == 63) s_wave[tid >> 6] = incl;
    __syncthreads();
    uint32_t base = 0, total = 0;
#pragma unroll
    for (int w = 0; w < 16; w++) { const uint32_t x = s_wave[w]; if (w < (tid >> 6)) base += x; total += x; }
    uint32_t run = base + incl - sum;
#pragma unroll
    for (int k = 0; k < PER; k++) { const int c = tid * PER + k; if (c < NC) { cell_start[c] = (int32_t)run; cnt[c] = run; run += loc[k]; } }
    if (tid == 0) cell_start[NC] = (int32_t)total;
    __syncthreads();
    for (int i = tid; i < n; i += 1024) {
        const int cx = cell[2 * i], cy = cell[2 * i + 1];
        if (cx >= 0) cell_items[atomicAdd(&cnt[cy * GRID_COLS + cx], 1u)] = (uint16_t)i;
    }
}

// one row of a cv::Mat product A*B (+ C): float inputs, double accumulation, alpha applied in double, one rounding to float (cv::gemm)
__device__ __forceinline__ float gemm3(const float* A, float b0, float b1, float b2, float c, double alpha = 1.0)
{
    double s = 0.0;
    s = __dadd_rn(s, __dmul_rn((double)A[0], (double)b0));
    s = __dadd_rn(s, __dmul_rn((double)A[1], (double)b1));
    s = __dadd_rn(s, __dmul_rn((double)A[2], (double)b2));
    return (float)__dadd_rn(__dmul_rn(alpha, s), (double)c);
}
// Camera::Project(Pc, uv) (Camera.cpp:116-153) on a point in camera coordinates.  uv = (u, v, ur); returns validity.
__device__ __forceinline__ bool cam_project(const HsFrameDev& F, const float* Pc, float& u, float& v, float& ur);
// Frame::ProjectLandMark + Camera::Project.
__device__ __forceinline__ bool project(const HsFrameDev& F, float px, float py, float pz, float& u, float& v, float& ur)
{
    float Pc[3];
#pragma unroll
    for (int i = 0; i < 3; i++) {
        double s = 0.0;
        s = __dadd_rn(s, __dmul_rn((double)F.Rcw[3 * i + 0], (double)px));
        s = __dadd_rn(s, __dmul_rn((double)F.Rcw[3 * i + 1], (double)py));
        s = __dadd_rn(s, __dmul_rn((double)F.Rcw[3 * i + 2], (double)pz));
        Pc[i] = (float)__dadd_rn(s, (double)F.tcw[i]);
    }
    return cam_project(F, Pc, u, v, ur);
}
__device__ __forceinline__ bool cam_project(const HsFrameDev& F, const float* Pc, float& u, float& v, float& ur)
{
    const float PcZ = Pc[2];
    const float invz = __fdiv_rn(1.0f, PcZ);
    const float hx = __fdiv_rn(Pc[0], PcZ), hy = __fdiv_rn(Pc[1], PcZ), hz = __fdiv_rn(Pc[2], PcZ);
    u = (float)__dadd_rn(__dadd_rn(__dmul_rn((double)F.fx, (double)hx), __dmul_rn(0.0, (double)hy)), __dmul_rn((double)F.cx, (double)hz));
    v = (float)__dadd_rn(__dadd_rn(__dmul_rn(0.0, (double)hx), __dmul_rn((double)F.fy, (double)hy)), __dmul_rn((double)F.cy, (double)hz));
    ur = F.sensor == 1 ? __fsub_rn(u, __fmul_rn(F.mbf, invz)) : -1.0f;
    return PcZ > 0.0f && u >= F.min_x && u <= F.max_x && v >= F.min_y && v <= F.max_y;
}

__device__ __forceinline__ int hamming256(const unsigned long long* a, const unsigned long long* b)
{
    return __popcll(a[0] ^ b[0]) + __popcll(a[1] ^ b[1]) + __popcll(a[2] ^ b[2]) + __popcll(a[3] ^ b[3]);
}

// wave-wide merge of (best key, second-best distance)
__device__ __forceinline__ void wave_best2(unsigned long long& best, int& second)
{
#pragma unroll
    for (int s = 32; s > 0; s >>= 1) {
        unsigned long long ob = __shfl_xor(best, s, 64);
        int os = __shfl_xor(second, s, 64);
        int db = (int)(best >> 32), dob = (int)(ob >> 32);
        int worse = max(db, dob);          // distance of whichever best loses (0x7FFFFFFF when a side is empty)
        best = min(best, ob);
        second = min(min(second, os), worse);
    }
}
#define NO_KEY 0x7FFFFFFFFFFFFFFFull
#define NO_DIST 0x7FFFFFFF

struct HsProjDev : hs_proj_params { float cos_view_angle; };   // cosf(max_view_angle), evaluated on the host like the reference does

// Fuse: fuse_matches.insert(idx, lm) — the first landmark (array order) that matched a keypoint keeps it
__global__ __launch_bounds__(1024) void k_first_wins(int n, int32_t* __restrict__ match, int32_t* __restrict__ winner, int n_keys, int32_t* __restrict__ n_out)
{
    __shared__ int total;
    const int tid = threadIdx.x;
    if (tid == 0) total = 0;
    for (int k = tid; k < n_keys; k += 1024) winner[k] = 0x7FFFFFFF;
    __syncthreads();
    for (int i = tid; i < n; i += 1024) if (match[i] >= 0) atomicMin(&winner[match[i]], i);
    __syncthreads();
    int kept = 0;
    for (int i = tid; i < n; i += 1024) {
        if (match[i] < 0) continue;
        if (winner[match[i]] != i) match[i] = -1; else kept++;
    }
    atomicAdd(&total, kept);
    __syncthreads();
    if (tid == 0) *n_out = total;
}

__global__ __launch_bounds__(256) void k_search_projection(HsFrameDev F, const hs_landmark* __restrict__ lms, int L, HsProjDev pp,
                                                           int32_t* __restrict__ match_idx, float* __restrict__ match_dist)
{
    const int lane = threadIdx.x & 63;
    const int li = blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);       // wave index: uniform
    if (li >= L) return;
    const hs_landmark& lm = lms[li];
    int out_idx = -1; float out_dist = -1.f;
    float u, v, ur;
    bool ok = !lm.skip && project(F, lm.pos[0], lm.pos[1], lm.pos[2], u, v, ur);                  // ProjectionCriterion
    if (ok && pp.use_distance) {                                                                    // DistanceCriterionCore
        const float ox = __fsub_rn(lm.pos[0], F.Ow[0]), oy = __fsub_rn(lm.pos[1], F.Ow[1]), oz = __fsub_rn(lm.pos[2], F.Ow[2]);
        const double n2 = __dadd_rn(__dadd_rn(__dmul_rn((double)ox, (double)ox), __dmul_rn((double)oy, (double)oy)), __dmul_rn((double)oz, (double)oz));
        const float dist = (float)sqrt(n2);
        const float lo = pp.dist_is_invariance_range ? lm.min_dist : __fmul_rn(0.8f, lm.min_dist);
        const float hi = pp.dist_is_invariance_range ? lm.max_dist : __fmul_rn(1.2f, lm.max_dist);
        if (dist < lo || dist > hi) ok = false;
    }
    if (ok && pp.use_viewing_angle) {                                                               // ViewingAngleCriterionCore
        const float ox = __fsub_rn(lm.pos[0], F.Ow[0]), oy = __fsub_rn(lm.pos[1], F.Ow[1]), oz = __fsub_rn(lm.pos[2], F.Ow[2]);
        const double n2 = __dadd_rn(__dadd_rn(__dmul_rn((double)ox, (double)ox), __dmul_rn((double)oy, (double)oy)), __dmul_rn((double)oz, (double)oz));
        const float distance = (float)sqrt(n2);
        const double alpha = __ddiv_rn(1.0, (double)distance);
        double dot = 0.0;
        dot = __dadd_rn(dot, __dmul_rn((double)(float)__dmul_rn((double)ox, alpha), (double)lm.normal[0]));
        dot = __dadd_rn(dot, __dmul_rn((double)(float)__dmul_rn((double)oy, alpha), (double)lm.normal[1]));
        dot = __dadd_rn(dot, __dmul_rn((double)(float)__dmul_rn((double)oz, alpha), (double)lm.normal[2]));
        if (!(dot > (double)pp.cos_view_angle)) ok = false;
    }
    if (ok) {
        // landMarkSizePixels
        float sizePx;
        if (lm.assoc_kp >= 0) sizePx = F.kps[lm.assoc_kp].size;
        else {
            const float half = __fdiv_rn(lm.size, 2.0f);
            float ul, vl, url, u2, v2, ur2;
            project(F, __fsub_rn(lm.pos[0], half), lm.pos[1], lm.pos[2], ul, vl, url);
            project(F, __fadd_rn(lm.pos[0], half), lm.pos[1], lm.pos[2], u2, v2, ur2);
            sizePx = __fsub_rn(u2, ul);
        }
        const float r = __fdiv_rn(__fmul_rn(pp.th, sizePx), F.size_ref);
        // GetFeaturesInAreaNEW cell range (with its early returns)
        const float invW = (float)GRID_COLS / (F.max_x - F.min_x), invH = (float)GRID_ROWS / (F.max_y - F.min_y);
        const int minCX = max(0, (int)floorf(__fmul_rn(__fsub_rn(__fsub_rn(u, F.min_x), r), invW)));
        const int maxCX = min(GRID_COLS - 1, (int)ceilf(__fmul_rn(__fadd_rn(__fsub_rn(u, F.min_x), r), invW)));
        const int minCY = max(0, (int)floorf(__fmul_rn(__fsub_rn(__fsub_rn(v, F.min_y), r), invH)));
        const int maxCY = min(GRID_ROWS - 1, (int)ceilf(__fmul_rn(__fadd_rn(__fsub_rn(v, F.min_y), r), invH)));
        const bool any = !(minCX >= GRID_COLS || maxCX < 0 || minCY >= GRID_ROWS || maxCY < 0);
        const float smin = __fmul_rn(pp.frac_smaller, sizePx), smax = __fmul_rn(pp.frac_larger, sizePx);
        const bool stereo = pp.use_stereo && F.sensor != 0;
        const unsigned long long* dl = reinterpret_cast<const unsigned long long*>(lm.desc);
        const unsigned long long l0 = dl[0], l1 = dl[1], l2 = dl[2], l3 = dl[3];
        unsigned long long best = NO_KEY; int second = NO_DIST;
        // candidates: the keypoints of the grid cells [minCX,maxCX] x [minCY,maxCY] (GetFeaturesInAreaNEW); a row of cells is one
        // contiguous span of the cell lists.  Without lists (cell_start == nullptr) every keypoint is tested against the cell range.
        auto consider = [&](int i, int cx, int cy) {
            const hs_keypoint kp = F.kps[i];
            if (!(fabsf(__fsub_rn(kp.x, u)) < r && fabsf(__fsub_rn(kp.y, v)) < r)) return;
            if (pp.use_prev_matched && F.kp_lm_obs && F.kp_lm_obs[i] > 0) return;        // PreviouslyMatchedCriterionCore
            if (!(kp.size > smin && kp.size < smax)) return;                              // FeatureSizeCriterionCore
            if (stereo) { const float urv = F.uR[i]; if (!(fabsf(__fsub_rn(ur, urv)) < r && urv > 0.f)) return; }
            if (pp.use_reprojection) {                                                      // ProjectionViewCriterion + KeyFrame::ReprojectionError
                const float ex = __fsub_rn(u, kp.x), ey = __fsub_rn(v, kp.y);
                const float urv = F.uR ? F.uR[i] : -1.f;
                const float er = urv >= 0.0f ? __fsub_rn(ur, urv) : 0.0f;
                const float err = __fadd_rn(__fadd_rn(__fmul_rn(ex, ex), __fmul_rn(ey, ey)), __fmul_rn(er, er));
                const float sf = __fdiv_rn(kp.size, F.size_ref);
                const float sigma = __fmul_rn(pp.sigma_ref, __fmul_rn(sf, sf));
                const float stereo_factor = urv > 0.f ? 1.30f : 1.00f;
                if (!(__fdiv_rn(err, sigma) < __fmul_rn(stereo_factor, pp.reproj_threshold))) return;
            }
            const unsigned long long* dk = reinterpret_cast<const unsigned long long*>(F.desc + (size_t)i * 32);
            const int d = __popcll(l0 ^ dk[0]) + __popcll(l1 ^ dk[1]) + __popcll(l2 ^ dk[2]) + __popcll(l3 ^ dk[3]);
            const unsigned long long key = ((unsigned long long)d << 32) | ((unsigned long long)cx << 22) | ((unsigned long long)cy << 16) | (unsigned)i;
            if (key < best) { second = min(second, (int)(best >> 32)); best = key; }
            else second = min(second, d);
        };
        if (any && F.cell_start) {
            for (int cy = minCY; cy <= maxCY; cy++) {
                const int a = hs_cload<int32_t>(F.cell_start + cy * GRID_COLS + minCX), b = hs_cload<int32_t>(F.cell_start + cy * GRID_COLS + maxCX + 1);
                for (int t = a + lane; t < b; t += 64) {
                    const int i = F.cell_items[t];
                    consider(i, F.cell[2 * i], cy);
                }
            }
        } else if (any) {
            for (int i = lane; i < F.n; i += 64) {
                const int cx = F.cell[2 * i], cy = F.cell[2 * i + 1];
                if (cx < minCX || cx > maxCX || cy < minCY || cy > maxCY) continue;           // also drops cx == -1
                consider(i, cx, cy);
            }
        }
        wave_best2(best, second);
        if (best != NO_KEY) {                                                                   // BestScoreCriterion accept rule
            const float bestDist = (float)(int)(best >> 32);
            const float bestDist2 = second == NO_DIST ? FLT_MAX : (float)second;
            if (bestDist <= pp.score_threshold && !(bestDist > __fmul_rn(pp.second_best_ratio, bestDist2))) {
                out_idx = (int)(best & 0xFFFF); out_dist = bestDist;
            }
        }
    }
    if (lane == 0) { match_idx[li] = out_idx; match_dist[li] = out_dist; }
}

// RotationConsistency on a match list a[i] -> b: keep only pairs whose rotation bin is one of the three most populated.
// DEDUPE: several entries may share the same key_idx (landmarks that matched the same keypoint); the reference collects them
// in a std::map keyed by that index, so only the LAST entry survives, the others are dropped.
__global__ __launch_bounds__(1024) void k_rotation_filter(int n, int32_t* __restrict__ match /*[n] key idx or -1, in/out*/,
                                                          const float* __restrict__ angle_prev /*[n] per entry*/,
                                                          const hs_keypoint* __restrict__ kps_curr /*indexed by match[i]*/,
                                                          int32_t* __restrict__ winner /*[n_keys] scratch*/, int n_keys, int dedupe,
                                                          int32_t* __restrict__ n_out)
{
    __shared__ int hist[30];
    __shared__ int ind[3];
    __shared__ int total;
    const int tid = threadIdx.x;
    if (tid < 30) hist[tid] = 0;
    if (tid == 0) total = 0;
    if (dedupe) {
        for (int k = tid; k < n_keys; k += 1024) winner[k] = -1;
        __syncthreads();
        for (int i = tid; i < n; i += 1024) if (match[i] >= 0) atomicMax(&winner[match[i]], i);
    }
    __syncthreads();
    auto bin_of = [&](int i) {
        float rot = __fsub_rn(angle_prev[i], kps_curr[match[i]].angle);
        if (rot < 0.0f) rot = __fadd_rn(rot, 360.0f);
        int b = (int)roundf(__fmul_rn(rot, 1.0f / 30));
        return b == 30 ? 0 : b;
    };
    for (int i = tid; i < n; i += 1024) {
        if (match[i] < 0) continue;
        if (dedupe && winner[match[i]] != i) continue;
        int b = bin_of(i);
        if (b >= 0 && b < 30) atomicAdd(&hist[b], 1);
    }
    __syncthreads();
    if (tid == 0) {   // ComputeThreeMaxima
        int max1 = 0, max2 = 0, max3 = 0, i1 = -1, i2 = -1, i3 = -1;
        for (int i = 0; i < 30; i++) {
            const int s = hist[i];
            if (s > max1) { max3 = max2; max2 = max1; max1 = s; i3 = i2; i2 = i1; i1 = i; }
            else if (s > max2) { max3 = max2; max2 = s; i3 = i2; i2 = i; }
            else if (s > max3) { max3 = s; i3 = i; }
        }
        if ((float)max2 < 0.1f * (float)max1) { i2 = -1; i3 = -1; }
        else if ((float)max3 < 0.1f * (float)max1) { i3 = -1; }
        ind[0] = i1; ind[1] = i2; ind[2] = i3;
    }
    __syncthreads();
    int kept = 0;
    for (int i = tid; i < n; i += 1024) {
        if (match[i] < 0) continue;
        bool keep = !(dedupe && winner[match[i]] != i);
        if (keep) { int b = bin_of(i); keep = (b == ind[0] || b == ind[1] || b == ind[2]); }
        if (!keep) match[i] = -1; else kept++;
    }
    atomicAdd(&total, kept);
    __syncthreads();
    if (tid == 0) *n_out = total;
}

__global__ __launch_bounds__(1024) void k_count_matches(int n, const int32_t* __restrict__ match, int32_t* __restrict__ n_out)
{
    __shared__ int total;
    if (threadIdx.x == 0) total = 0;
    __syncthreads();
    int c = 0;
    // 16-byte loads, several in flight per thread (one workgroup: the count is a 200 KB read for 50 k landmarks)
    const int n4 = ((reinterpret_cast<uintptr_t>(match) & 15) == 0) ? (n >> 2) : 0;
    const int4* m4 = reinterpret_cast<const int4*>(match);
#pragma unroll 4
    for (int i = threadIdx.x; i < n4; i += 1024) { const int4 v = m4[i]; c += (v.x >= 0) + (v.y >= 0) + (v.z >= 0) + (v.w >= 0); }
    for (int i = 4 * n4 + threadIdx.x; i < n; i += 1024) c += match[i] >= 0;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o, 64);
    if ((threadIdx.x & 63) == 0) atomicAdd(&total, c);
    __syncthreads();
    if (threadIdx.x == 0) *n_out = total;
}

struct HsEpi { int32_t on; float F[9]; float size_ref, sigma_ref; };   // EpipolarConsistencyBoWCriterion (SearchForTriangulation)

// One workgroup per vocabulary node shared by both feature vectors; one wavefront per side-1 index, lanes over the node's side-2 list.
__global__ __launch_bounds__(256) void k_bow_match(const int32_t* __restrict__ pair_a, const int32_t* __restrict__ pair_b,
                                                   const int32_t* __restrict__ ptr1, const int32_t* __restrict__ idx1,
                                                   const int32_t* __restrict__ ptr2, const int32_t* __restrict__ idx2,
                                                   const uint8_t* __restrict__ desc1, const uint8_t* __restrict__ desc2,
                                                   const uint8_t* __restrict__ keep1, const uint8_t* __restrict__ keep2,
                                                   HsEpi epi, const hs_keypoint* __restrict__ kps1, const hs_keypoint* __restrict__ kps2,
                                                   float score_threshold, float ratio, int32_t* __restrict__ match12)
{
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int a = pair_a[blockIdx.x], b = pair_b[blockIdx.x];
    const int p0 = ptr1[a], p1 = ptr1[a + 1], q0 = ptr2[b], q1 = ptr2[b + 1];
    for (int p = p0 + wv; p < p1; p += 4) {
        const int i1 = idx1[p];
        if (keep1 && !keep1[i1]) continue;                                   // PreviouslyMatchedIndexCriterion
        const unsigned long long* d1 = reinterpret_cast<const unsigned long long*>(desc1 + (size_t)i1 * 32);
        unsigned long long best = NO_KEY; int second = NO_DIST;
        float ea = 0.f, eb = 0.f, ec = 0.f;
        if (epi.on) {   // epipolar line of kp1 in image 2: l = x1' F12 (MatchCriteria.cpp:661-663)
            const float x1 = kps1[i1].x, y1 = kps1[i1].y;
            ea = __fadd_rn(__fadd_rn(__fmul_rn(x1, epi.F[0]), __fmul_rn(y1, epi.F[3])), epi.F[6]);
            eb = __fadd_rn(__fadd_rn(__fmul_rn(x1, epi.F[1]), __fmul_rn(y1, epi.F[4])), epi.F[7]);
            ec = __fadd_rn(__fadd_rn(__fmul_rn(x1, epi.F[2]), __fmul_rn(y1, epi.F[5])), epi.F[8]);
        }
        for (int q = q0 + lane; q < q1; q += 64) {
            const int i2 = idx2[q];
            if (keep2 && !keep2[i2]) continue;
            if (epi.on) {
                const hs_keypoint k2 = kps2[i2];
                const float num = __fadd_rn(__fadd_rn(__fmul_rn(ea, k2.x), __fmul_rn(eb, k2.y)), ec);
                const float den = __fadd_rn(__fmul_rn(ea, ea), __fmul_rn(eb, eb));
                if (den == 0.f) continue;
                const float dsqr = __fdiv_rn(__fmul_rn(num, num), den);
                const float sf = __fdiv_rn(k2.size, epi.size_ref);
                const float sigma2 = __fmul_rn(epi.sigma_ref, __fmul_rn(sf, sf));
                if (!((double)dsqr < __dmul_rn(3.84, (double)sigma2))) continue;
            }
            const int d = hamming256(d1, reinterpret_cast<const unsigned long long*>(desc2 + (size_t)i2 * 32));
            const unsigned long long key = ((unsigned long long)d << 32) | (unsigned)(q - q0);     // list order breaks ties
            if (key < best) { second = min(second, (int)(best >> 32)); best = key; }
            else second = min(second, d);
        }
        wave_best2(best, second);
        if (lane == 0 && best != NO_KEY) {
            const float bd1 = (float)(int)(best >> 32), bd2 = second == NO_DIST ? FLT_MAX : (float)second;
            if (bd1 < score_threshold && bd1 < __fmul_rn(ratio, bd2)) match12[i1] = idx2[q0 + (int)(best & 0xFFFFFFFFu)];
        }
    }
}

// The legacy SearchByBoW(pKF1, pKF2, vpMatches12) (FeatureMatcher.cc:938-1077): like k_bow_match, but a side-2 feature that an earlier side-1
// feature matched is no longer a candidate (vbMatched2).  That makes the side-1 loop of a node SEQUENTIAL; the nodes stay independent because a
// feature belongs to exactly one node of the feature vector.  One wavefront per shared node walks the node's side-1 list in order, its lanes
// cover the side-2 list; `taken2` (zeroed by the launcher) is written and read by this wave only — through L2 (atomic load / store at agent scope):
// the vector L1 is not coherent with a wave's own earlier stores.
__global__ __launch_bounds__(64) void k_bow_match_exclusive(const int32_t* __restrict__ pair_a, const int32_t* __restrict__ pair_b,
                                                            const int32_t* __restrict__ ptr1, const int32_t* __restrict__ idx1,
                                                            const int32_t* __restrict__ ptr2, const int32_t* __restrict__ idx2,
                                                            const uint8_t* __restrict__ desc1, const uint8_t* __restrict__ desc2,
                                                            const uint8_t* __restrict__ keep1, const uint8_t* __restrict__ keep2,
                                                            float score_threshold, float ratio, int32_t* __restrict__ match12, uint32_t* __restrict__ taken2)
{
    const int lane = threadIdx.x;
    const int a = pair_a[blockIdx.x], b = pair_b[blockIdx.x];
    const int p0 = ptr1[a], p1 = ptr1[a + 1], q0 = ptr2[b], q1 = ptr2[b + 1];
    for (int p = p0; p < p1; p++) {
        const int i1 = idx1[p];
        if (keep1 && !keep1[i1]) continue;                                   // !pMP1 || pMP1->isBad() (:985-990)
        const unsigned long long* d1 = reinterpret_cast<const unsigned long long*>(desc1 + (size_t)i1 * 32);
        unsigned long long best = NO_KEY; int second = NO_DIST;
        for (int q = q0 + lane; q < q1; q += 64) {
            const int i2 = idx2[q];
            if (keep2 && !keep2[i2]) continue;
            if (__hip_atomic_load(&taken2[i2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) continue;      // vbMatched2[idx2] (:999)
            const int d = hamming256(d1, reinterpret_cast<const unsigned long long*>(desc2 + (size_t)i2 * 32));
            const unsigned long long key = ((unsigned long long)d << 32) | (unsigned)(q - q0);     // list order breaks ties
            if (key < best) { second = min(second, (int)(best >> 32)); best = key; }
            else second = min(second, d);
        }
        wave_best2(best, second);                                            // every lane holds the result
        if (best != NO_KEY) {
            const float bd1 = (float)(int)(best >> 32), bd2 = second == NO_DIST ? FLT_MAX : (float)second;
            if (bd1 < score_threshold && bd1 < __fmul_rn(ratio, bd2)) {
                const int i2 = idx2[q0 + (int)(best & 0xFFFFFFFFu)];
                if (lane == 0) { match12[i1] = i2; __hip_atomic_store(&taken2[i2], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
                __builtin_amdgcn_s_waitcnt(0x0f70);                          // vmcnt(0): the flag is in L2 before the next feature's loads
            }
        }
    }
}

// one wavefront: 2-NN of query descriptor i over the train set (first minimum wins; second = second smallest of the multiset)
__device__ __forceinline__ void knn2_wave(const uint8_t* __restrict__ q, int i, const uint8_t* __restrict__ t, int nt,
                                          int32_t* __restrict__ best_idx, int32_t* __restrict__ best_dist, int32_t* __restrict__ second_dist)
{
    const int lane = threadIdx.x & 63;
    const unsigned long long* dq = reinterpret_cast<const unsigned long long*>(q + (size_t)i * 32);
    const unsigned long long a0 = dq[0], a1 = dq[1], a2 = dq[2], a3 = dq[3];
    unsigned long long best = NO_KEY; int second = NO_DIST;
    for (int j = lane; j < nt; j += 64) {
        const unsigned long long* dt = reinterpret_cast<const unsigned long long*>(t + (size_t)j * 32);
        const int d = __popcll(a0 ^ dt[0]) + __popcll(a1 ^ dt[1]) + __popcll(a2 ^ dt[2]) + __popcll(a3 ^ dt[3]);
        const unsigned long long key = ((unsigned long long)d << 32) | (unsigned)j;
        if (key < best) { second = min(second, (int)(best >> 32)); best = key; }
        else second = min(second, d);
    }
    wave_best2(best, second);
    if (lane == 0) {
        best_idx[i] = best == NO_KEY ? -1 : (int)(best & 0xFFFFFFFFu);
        best_dist[i] = best == NO_KEY ? -1 : (int)(best >> 32);
        second_dist[i] = second == NO_DIST ? -1 : second;
    }
}

__global__ __launch_bounds__(256) void k_knn2(const uint8_t* __restrict__ q, int nq, const uint8_t* __restrict__ t, int nt,
                                              int32_t* __restrict__ best_idx, int32_t* __restrict__ best_dist, int32_t* __restrict__ second_dist)
{
    const int i = blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (i >= nq) return;
    knn2_wave(q, i, t, nt, best_idx, best_dist, second_dist);
}

// Cross-camera 2-NN over gathered frame records (hs_records_knn2_device): blockIdx.y = peer record; the counts come from the record
// headers on the device, so the step needs no host round trip between the all-gather and the matcher.
__global__ __launch_bounds__(256) void k_knn2_records(const uint8_t* __restrict__ recs, size_t stride, int rank, int cap, size_t off_desc,
                                                      int32_t* __restrict__ best_idx, int32_t* __restrict__ best_dist, int32_t* __restrict__ second_dist)
{
    const int peer = blockIdx.y;
    if (peer == rank) return;
    const int nq = min(max(hs_cload<int32_t>(recs + (size_t)rank * stride), 0), cap);
    const int nt = min(max(hs_cload<int32_t>(recs + (size_t)peer * stride), 0), cap);
    const int i = blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (i >= nq) return;
    const size_t o = (size_t)peer * cap;
    knn2_wave(recs + (size_t)rank * stride + off_desc, i, recs + (size_t)peer * stride + off_desc, nt, best_idx + o, best_dist + o, second_dist + o);
}

// ---------------------------------------------------------------- legacy loop-closing matchers (FeatureMatcher.cc:628-934)
// landMarkSizePixels of a landmark in frame F (projects with F's OWN pose — also in the Sim3 variants, KeyFrame.cc:258-279)
__device__ __forceinline__ float landmark_size_px(const HsFrameDev& F, const hs_landmark& lm)
{
    if (lm.assoc_kp >= 0) return F.kps[lm.assoc_kp].size;
    const float half = __fdiv_rn(lm.size, 2.0f);
    float ul, vl, url, u2, v2, ur2;
    project(F, __fsub_rn(lm.pos[0], half), lm.pos[1], lm.pos[2], ul, vl, url);
    project(F, __fadd_rn(lm.pos[0], half), lm.pos[1], lm.pos[2], u2, v2, ur2);
    return __fsub_rn(u2, ul);
}
// best Hamming over GetFeaturesInArea(u, v, r) in the reference's candidate order (first minimum wins), skipping keypoints whose bit is set in
// `taken` (LDS bitmask, may be null).  Wave-wide; returns the key dist<<32 | cellx<<22 | celly<<16 | idx or NO_KEY.
__device__ __forceinline__ unsigned long long best_in_area(const HsFrameDev& F, float u, float v, float r, const uint8_t* desc32, const uint32_t* taken)
{
    const int lane = threadIdx.x & 63;
    const float invW = (float)GRID_COLS / (F.max_x - F.min_x), invH = (float)GRID_ROWS / (F.max_y - F.min_y);
    const int minCX = max(0, (int)floorf(__fmul_rn(__fsub_rn(__fsub_rn(u, F.min_x), r), invW)));
    const int maxCX = min(GRID_COLS - 1, (int)ceilf(__fmul_rn(__fadd_rn(__fsub_rn(u, F.min_x), r), invW)));
    const int minCY = max(0, (int)floorf(__fmul_rn(__fsub_rn(__fsub_rn(v, F.min_y), r), invH)));
    const int maxCY = min(GRID_ROWS - 1, (int)ceilf(__fmul_rn(__fadd_rn(__fsub_rn(v, F.min_y), r), invH)));
    unsigned long long best = NO_KEY;
    if (minCX >= GRID_COLS || maxCX < 0 || minCY >= GRID_ROWS || maxCY < 0) return best;
    const unsigned long long* dl = reinterpret_cast<const unsigned long long*>(desc32);
    const unsigned long long l0 = dl[0], l1 = dl[1], l2 = dl[2], l3 = dl[3];
    for (int cy = minCY; cy <= maxCY; cy++) {
        const int a = F.cell_start[cy * GRID_COLS + minCX], b = F.cell_start[cy * GRID_COLS + maxCX + 1];
        for (int t = a + lane; t < b; t += 64) {
            const int i = F.cell_items[t];
            const hs_keypoint kp = F.kps[i];
            if (!(fabsf(__fsub_rn(kp.x, u)) < r && fabsf(__fsub_rn(kp.y, v)) < r)) continue;
            if (taken && ((taken[i >> 5] >> (i & 31)) & 1u)) continue;
            const unsigned long long* dk = reinterpret_cast<const unsigned long long*>(F.desc + (size_t)i * 32);
            const int d = __popcll(l0 ^ dk[0]) + __popcll(l1 ^ dk[1]) + __popcll(l2 ^ dk[2]) + __popcll(l3 ^ dk[3]);
            const unsigned long long key = ((unsigned long long)d << 32) | ((unsigned long long)F.cell[2 * i] << 22) | ((unsigned long long)cy << 16) | (unsigned)i;
            best = min(best, key);
        }
    }
#pragma unroll
    for (int s = 32; s > 0; s >>= 1) best = min(best, (unsigned long long)__shfl_xor(best, s, 64));
    return best;
}

struct HsSim3 { float R[9], t[3], Ow[3]; };        // decomposed on the host exactly like the reference does (launcher)

// SearchByProjection(pKF, Scw, vpPoints, vpMatched, th), phase A (parallel): everything that does not depend on vpMatched.
// geo[li] = (u, v, radius) or radius < 0 when the landmark is rejected before the candidate search.
__global__ __launch_bounds__(256) void k_sim3_project(HsFrameDev F, HsSim3 S, const hs_landmark* __restrict__ lms, int L, float th, float* __restrict__ geo)
{
    const int li = blockIdx.x * 256 + threadIdx.x;
    if (li >= L) return;
    const hs_landmark& lm = lms[li];
    float u = 0.f, v = 0.f, radius = -1.f;
    if (!lm.skip) {
        const float z = gemm3(&S.R[6], lm.pos[0], lm.pos[1], lm.pos[2], S.t[2]);
        if (!(z < 0.0f)) {
            const float invz = __fdiv_rn(1.0f, z);
            const float x = __fmul_rn(gemm3(&S.R[0], lm.pos[0], lm.pos[1], lm.pos[2], S.t[0]), invz);
            const float y = __fmul_rn(gemm3(&S.R[3], lm.pos[0], lm.pos[1], lm.pos[2], S.t[1]), invz);
            u = __fadd_rn(__fmul_rn(F.fx, x), F.cx); v = __fadd_rn(__fmul_rn(F.fy, y), F.cy);
            if (u >= F.min_x && u < F.max_x && v >= F.min_y && v < F.max_y) {                   // KeyFrame::IsInImage
                const float ox = __fsub_rn(lm.pos[0], S.Ow[0]), oy = __fsub_rn(lm.pos[1], S.Ow[1]), oz = __fsub_rn(lm.pos[2], S.Ow[2]);
                const float dist = (float)sqrt(__dadd_rn(__dadd_rn(__dmul_rn((double)ox, (double)ox), __dmul_rn((double)oy, (double)oy)), __dmul_rn((double)oz, (double)oz)));
                if (!(dist < lm.min_dist || dist > lm.max_dist)) {                                  // invariance range
                    const double dot = __dadd_rn(__dadd_rn(__dmul_rn((double)ox, (double)lm.normal[0]), __dmul_rn((double)oy, (double)lm.normal[1])), __dmul_rn((double)oz, (double)lm.normal[2]));
                    if (!(dot < __dmul_rn(0.5, (double)dist)))
                        radius = __fdiv_rn(__fmul_rn(th, landmark_size_px(F, lm)), F.size_ref);
                }
            }
        }
    }
    geo[3 * li] = u; geo[3 * li + 1] = v; geo[3 * li + 2] = radius;
}
// phase B (one wave, landmarks in order): best untaken keypoint in the area; a match takes its keypoint (vpMatched[bestIdx] = pMP, :731)
__global__ __launch_bounds__(64) void k_sim3_assign(HsFrameDev F, const hs_landmark* __restrict__ lms, int L, const float* __restrict__ geo, float th_low,
                                                    uint8_t* __restrict__ kp_matched, int32_t* __restrict__ match_idx, int32_t* __restrict__ n_matches)
{
    __shared__ uint32_t taken[2048];                                     // 65536 keypoints
    const int lane = threadIdx.x;
    for (int i = lane; i < 2048; i += 64) taken[i] = 0;
    __syncthreads();
    for (int i = lane; i < F.n; i += 64) if (kp_matched[i]) atomicOr(&taken[i >> 5], 1u << (i & 31));
    __syncthreads();
    int n = 0;
    for (int li = 0; li < L; li++) {
        const float r = geo[3 * li + 2];
        int out = -1;
        if (r >= 0.f) {                                                      // a negative or NaN radius finds no candidate in the reference either
            const unsigned long long best = best_in_area(F, geo[3 * li], geo[3 * li + 1], r, lms[li].desc, taken);
            if (best != NO_KEY && (float)(int)(best >> 32) <= th_low) {
                out = (int)(best & 0xFFFF);
                if (lane == 0) { taken[out >> 5] |= 1u << (out & 31); kp_matched[out] = 1; }
                n++;
            }
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_s_waitcnt(0xc07f);                              // the LDS write above is visible to the next landmark's reads
        }
        if (lane == 0) match_idx[li] = out;
    }
    if (lane == 0) *n_matches = n;
}

// SearchBySim3, one direction: landmark of source keypoint i -> best keypoint of the destination keyframe (no exclusion, <= th_high)
__global__ __launch_bounds__(256) void k_sim3_direction(HsFrameDev Fsrc, HsFrameDev Fdst, const hs_landmark* __restrict__ lms, int n, HsSim3 S /*sR, t*/,
                                                        float th, float th_high, int32_t* __restrict__ out)
{
    const int i = blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (i >= n) return;
    const hs_landmark& lm = lms[i];
    int res = -1;
    if (!lm.skip) {
        float ps[3], pd[3];
#pragma unroll
        for (int k = 0; k < 3; k++) ps[k] = gemm3(&Fsrc.Rcw[3 * k], lm.pos[0], lm.pos[1], lm.pos[2], Fsrc.tcw[k]);
#pragma unroll
        for (int k = 0; k < 3; k++) pd[k] = gemm3(&S.R[3 * k], ps[0], ps[1], ps[2], S.t[k]);
        float u, v, ur;
        if (cam_project(Fdst, pd, u, v, ur)) {
            const float d3 = (float)sqrt(__dadd_rn(__dadd_rn(__dmul_rn((double)pd[0], (double)pd[0]), __dmul_rn((double)pd[1], (double)pd[1])), __dmul_rn((double)pd[2], (double)pd[2])));
            if (!(d3 < lm.min_dist || d3 > lm.max_dist)) {
                const float radius = __fdiv_rn(__fmul_rn(th, landmark_size_px(Fdst, lm)), Fdst.size_ref);
                const unsigned long long best = best_in_area(Fdst, u, v, radius, lm.desc, nullptr);
                if (best != NO_KEY && (float)(int)(best >> 32) <= th_high) res = (int)(best & 0xFFFF);
            }
        }
    }
    if ((threadIdx.x & 63) == 0) out[i] = res;
}
__global__ void k_sim3_agree(int n1, const int32_t* __restrict__ m1, const int32_t* __restrict__ m2, int32_t* __restrict__ match12, int32_t* __restrict__ n_found)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n1) return;
    const int idx2 = m1[i];
    const bool ok = idx2 >= 0 && m2[idx2] == i;
    match12[i] = ok ? idx2 : -1;
    if (ok) atomicAdd(n_found, 1);
}

// ---------------------------------------------------------------- launchers (declared in hs_internal.h)
// d_cell: hs_frame_grid_bytes(n) bytes: [n][2] cells, then (with_lists) the cell lists read by hs_launch_search_projection
size_t hs_frame_grid_bytes(int n) { return (((size_t)n * 2 + 15) & ~(size_t)15) + ((size_t)GRID_ROWS * GRID_COLS + 1) * 4 + (size_t)n * 2 + 16; }
static int32_t* grid_lists_start(int8_t* d_cell, int n) { return reinterpret_cast<int32_t*>(d_cell + (((size_t)n * 2 + 15) & ~(size_t)15)); }
void hs_launch_frame_grid(const hs_frame_view& F, const hs_keypoint* d_kps, int8_t* d_cell, bool with_lists, hipStream_t s)
{
    if (F.n <= 0) return;
    HsFrameDev D{}; D.min_x = F.min_x; D.max_x = F.max_x; D.min_y = F.min_y; D.max_y = F.max_y; D.n = F.n; D.kps = d_kps;
    hipLaunchKernelGGL(k_frame_grid, dim3((F.n + 255) / 256), dim3(256), 0, s, D, d_cell);
    if (with_lists) {
        int32_t* start = grid_lists_start(d_cell, F.n);
        hipLaunchKernelGGL(k_frame_grid_lists, dim3(1), dim3(1024), 0, s, F.n, d_cell, start, reinterpret_cast<uint16_t*>(start + GRID_ROWS * GRID_COLS + 1));
    }
}

void hs_launch_search_projection(const hs_frame_view& F, const hs_keypoint* d_kps, const uint8_t* d_desc, const float* d_uR,
                                 const int32_t* d_obs, const int8_t* d_cell, const hs_landmark* d_lms, int L, const hs_proj_params& pp,
                                 int32_t* d_match_idx, float* d_match_dist, int32_t* d_winner, float* d_prev_angle_scratch,
                                 int32_t* d_n_matches, hipStream_t s)
{
    HsFrameDev D{};
    for (int i = 0; i < 9; i++) D.Rcw[i] = F.Rcw[i];
    for (int i = 0; i < 3; i++) { D.tcw[i] = F.tcw[i]; D.Ow[i] = F.Ow[i]; }
    D.fx = F.fx; D.fy = F.fy; D.cx = F.cx; D.cy = F.cy; D.mbf = F.mbf; D.sensor = F.sensor;
    D.min_x = F.min_x; D.max_x = F.max_x; D.min_y = F.min_y; D.max_y = F.max_y; D.size_ref = F.size_ref; D.n = F.n;
    D.kps = d_kps; D.desc = d_desc; D.uR = d_uR; D.kp_lm_obs = d_obs; D.cell = d_cell;
    if (F.n > 0) {                                             // the cell lists follow the cells (hs_launch_frame_grid with_lists)
        const int32_t* start = grid_lists_start(const_cast<int8_t*>(d_cell), F.n);
        D.cell_start = start; D.cell_items = reinterpret_cast<const uint16_t*>(start + GRID_ROWS * GRID_COLS + 1);
    }
    HsProjDev P; static_cast<hs_proj_params&>(P) = pp; P.cos_view_angle = cosf(pp.max_view_angle);
    hipLaunchKernelGGL(k_search_projection, dim3((L + 3) / 4), dim3(256), 0, s, D, d_lms, L, P, d_match_idx, d_match_dist);
    if (pp.first_wins) {
        hipLaunchKernelGGL(k_first_wins, dim3(1), dim3(1024), 0, s, L, d_match_idx, d_winner, F.n, d_n_matches);
    } else if (pp.check_rotation) {
        // prev_angle lives inside the landmark records; the filter wants a flat float array per entry
        hipMemcpy2DAsync(d_prev_angle_scratch, sizeof(float), reinterpret_cast<const uint8_t*>(d_lms) + offsetof(hs_landmark, prev_angle),
                         sizeof(hs_landmark), sizeof(float), L, hipMemcpyDeviceToDevice, s);
        hipLaunchKernelGGL(k_rotation_filter, dim3(1), dim3(1024), 0, s, L, d_match_idx, d_prev_angle_scratch, d_kps, d_winner, F.n, 1, d_n_matches);
    } else {
        hipLaunchKernelGGL(k_count_matches, dim3(1), dim3(1024), 0, s, L, d_match_idx, d_n_matches);
    }
}

__global__ void k_iota_all(int n, int32_t* __restrict__ out) { const int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) out[i] = i; }
__global__ void k_gather_angle(int n, const int32_t* __restrict__ match, const hs_keypoint* __restrict__ kps2, float* __restrict__ out)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = match[i] >= 0 ? kps2[match[i]].angle : 0.f;
}
__global__ void k_iota_where(int n, const int32_t* __restrict__ match, int32_t* __restrict__ self)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) self[i] = match[i] >= 0 ? i : -1;
}
__global__ void k_mask_by(int n, const int32_t* __restrict__ self, int32_t* __restrict__ match)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n && self[i] < 0) match[i] = -1;
}

void hs_launch_bow(const int32_t* d_pair_a, const int32_t* d_pair_b, int n_pairs,
                   const int32_t* d_ptr1, const int32_t* d_idx1, const int32_t* d_ptr2, const int32_t* d_idx2,
                   const uint8_t* d_desc1, const uint8_t* d_desc2, const uint8_t* d_keep1, const uint8_t* d_keep2,
                   const float* F12, float size_ref, float sigma_ref, float thr, float ratio,
                   int32_t* d_match12, int n1, const hs_keypoint* d_kps1, const hs_keypoint* d_kps2, float* d_angle2_scratch,
                   int check_rotation, int32_t* d_self_scratch, int32_t* d_n_matches, hipStream_t s)
{
    HsEpi epi{}; epi.on = F12 != nullptr; epi.size_ref = size_ref; epi.sigma_ref = sigma_ref;
    if (F12) for (int i = 0; i < 9; i++) epi.F[i] = F12[i];
    hipMemsetAsync(d_match12, 0xFF, (size_t)n1 * 4, s);
    if (n_pairs > 0)
        hipLaunchKernelGGL(k_bow_match, dim3(n_pairs), dim3(256), 0, s, d_pair_a, d_pair_b, d_ptr1, d_idx1, d_ptr2, d_idx2, d_desc1, d_desc2,
                           d_keep1, d_keep2, epi, d_kps1, d_kps2, thr, ratio, d_match12);
    if (check_rotation && n1 > 0) {
        // RotationConsistencyBoW::apply(matches, views1, views2): rot = angle2[match] - angle1[i]; entries are keyed by side-1 index (unique)
        const int g = (n1 + 255) / 256;
        hipLaunchKernelGGL(k_gather_angle, dim3(g), dim3(256), 0, s, n1, d_match12, d_kps2, d_angle2_scratch);
        hipLaunchKernelGGL(k_iota_where, dim3(g), dim3(256), 0, s, n1, d_match12, d_self_scratch);
        hipLaunchKernelGGL(k_rotation_filter, dim3(1), dim3(1024), 0, s, n1, d_self_scratch, d_angle2_scratch, d_kps1, (int32_t*)nullptr, 0, 0, d_n_matches);
        hipLaunchKernelGGL(k_mask_by, dim3(g), dim3(256), 0, s, n1, d_self_scratch, d_match12);
    } else {
        hipLaunchKernelGGL(k_count_matches, dim3(1), dim3(1024), 0, s, n1, d_match12, d_n_matches);
    }
}

// legacy SearchByBoW(KF, KF): exclusive matching, then the rotation histogram on angle1 - angle2 (FeatureMatcher.cc:1031: the opposite sign of
// RotationConsistencyBoW), entries of the minority bins removed from match12
void hs_launch_bow_legacy(const int32_t* d_pair_a, const int32_t* d_pair_b, int n_pairs,
                          const int32_t* d_ptr1, const int32_t* d_idx1, const int32_t* d_ptr2, const int32_t* d_idx2,
                          const uint8_t* d_desc1, const uint8_t* d_desc2, const uint8_t* d_keep1, const uint8_t* d_keep2, float thr, float ratio,
                          int32_t* d_match12, int n1, int n2, const hs_keypoint* d_kps1, const hs_keypoint* d_kps2, float* d_angle1_scratch,
                          int check_orientation, int32_t* d_self_scratch, uint32_t* d_taken2, int32_t* d_n_matches, hipStream_t s)
{
    hipMemsetAsync(d_match12, 0xFF, (size_t)n1 * 4, s);
    hipMemsetAsync(d_taken2, 0, (size_t)n2 * 4, s);
    if (n_pairs > 0)
        hipLaunchKernelGGL(k_bow_match_exclusive, dim3(n_pairs), dim3(64), 0, s, d_pair_a, d_pair_b, d_ptr1, d_idx1, d_ptr2, d_idx2, d_desc1, d_desc2,
                           d_keep1, d_keep2, thr, ratio, d_match12, d_taken2);
    if (check_orientation && n1 > 0) {
        const int g = (n1 + 255) / 256;
        hipLaunchKernelGGL(k_iota_all, dim3(g), dim3(256), 0, s, n1, d_self_scratch);
        hipLaunchKernelGGL(k_gather_angle, dim3(g), dim3(256), 0, s, n1, d_self_scratch, d_kps1, d_angle1_scratch);      // angle of side-1 feature i
        hipLaunchKernelGGL(k_rotation_filter, dim3(1), dim3(1024), 0, s, n1, d_match12, d_angle1_scratch, d_kps2, (int32_t*)nullptr, 0, 0, d_n_matches);
    } else {
        hipLaunchKernelGGL(k_count_matches, dim3(1), dim3(1024), 0, s, n1, d_match12, d_n_matches);
    }
}

void hs_launch_knn2(const uint8_t* d_q, int nq, const uint8_t* d_t, int nt, int32_t* d_bi, int32_t* d_bd, int32_t* d_sd, hipStream_t s)
{
    if (nq <= 0) return;
    hipLaunchKernelGGL(k_knn2, dim3((nq + 3) / 4), dim3(256), 0, s, d_q, nq, d_t, nt, d_bi, d_bd, d_sd);
}

static HsFrameDev frame_dev(const hs_frame_view& F, const hs_keypoint* d_kps, const uint8_t* d_desc, const float* d_uR, const int32_t* d_obs, const int8_t* d_cell)
{
    HsFrameDev D{};
    for (int i = 0; i < 9; i++) D.Rcw[i] = F.Rcw[i];
    for (int i = 0; i < 3; i++) { D.tcw[i] = F.tcw[i]; D.Ow[i] = F.Ow[i]; }
    D.fx = F.fx; D.fy = F.fy; D.cx = F.cx; D.cy = F.cy; D.mbf = F.mbf; D.sensor = F.sensor;
    D.min_x = F.min_x; D.max_x = F.max_x; D.min_y = F.min_y; D.max_y = F.max_y; D.size_ref = F.size_ref; D.n = F.n;
    D.kps = d_kps; D.desc = d_desc; D.uR = d_uR; D.kp_lm_obs = d_obs; D.cell = d_cell;
    if (F.n > 0) {
        const int32_t* start = grid_lists_start(const_cast<int8_t*>(d_cell), F.n);
        D.cell_start = start; D.cell_items = reinterpret_cast<const uint16_t*>(start + GRID_ROWS * GRID_COLS + 1);
    }
    return D;
}

void hs_launch_sim3_projection(const hs_frame_view& F, const hs_keypoint* d_kps, const uint8_t* d_desc, const int8_t* d_cell,
                               const float* R9, const float* t3, const float* Ow3, const hs_landmark* d_lms, int L, float th, float th_low,
                               float* d_geo, uint8_t* d_kp_matched, int32_t* d_match_idx, int32_t* d_n_matches, hipStream_t s)
{
    const HsFrameDev D = frame_dev(F, d_kps, d_desc, nullptr, nullptr, d_cell);
    HsSim3 S; for (int i = 0; i < 9; i++) S.R[i] = R9[i]; for (int i = 0; i < 3; i++) { S.t[i] = t3[i]; S.Ow[i] = Ow3[i]; }
    hipLaunchKernelGGL(k_sim3_project, dim3((L + 255) / 256), dim3(256), 0, s, D, S, d_lms, L, th, d_geo);
    hipLaunchKernelGGL(k_sim3_assign, dim3(1), dim3(64), 0, s, D, d_lms, L, d_geo, th_low, d_kp_matched, d_match_idx, d_n_matches);
}

void hs_launch_sim3_search(const hs_frame_view& F1, const hs_keypoint* d_kps1, const uint8_t* d_desc1, const int8_t* d_cell1,
                           const hs_frame_view& F2, const hs_keypoint* d_kps2, const uint8_t* d_desc2, const int8_t* d_cell2,
                           const hs_landmark* d_lms1, const hs_landmark* d_lms2, const float* sR21, const float* t21, const float* sR12, const float* t12,
                           float th, float th_high, int32_t* d_m1, int32_t* d_m2, int32_t* d_match12, int32_t* d_n_found, hipStream_t s)
{
    const HsFrameDev D1 = frame_dev(F1, d_kps1, d_desc1, nullptr, nullptr, d_cell1), D2 = frame_dev(F2, d_kps2, d_desc2, nullptr, nullptr, d_cell2);
    HsSim3 S21{}, S12{};
    for (int i = 0; i < 9; i++) { S21.R[i] = sR21[i]; S12.R[i] = sR12[i]; }
    for (int i = 0; i < 3; i++) { S21.t[i] = t21[i]; S12.t[i] = t12[i]; }
    hipMemsetAsync(d_n_found, 0, 4, s);
    if (F1.n > 0) hipLaunchKernelGGL(k_sim3_direction, dim3((F1.n + 3) / 4), dim3(256), 0, s, D1, D2, d_lms1, F1.n, S21, th, th_high, d_m1);
    if (F2.n > 0) hipLaunchKernelGGL(k_sim3_direction, dim3((F2.n + 3) / 4), dim3(256), 0, s, D2, D1, d_lms2, F2.n, S12, th, th_high, d_m2);
    if (F1.n > 0) hipLaunchKernelGGL(k_sim3_agree, dim3((F1.n + 255) / 256), dim3(256), 0, s, F1.n, d_m1, d_m2, d_match12, d_n_found);
}

void hs_launch_knn2_records(const uint8_t* d_recs, size_t stride, int world, int rank, int cap, size_t off_desc,
                            int32_t* d_bi, int32_t* d_bd, int32_t* d_sd, hipStream_t s)
{
    if (world <= 0 || cap <= 0) return;
    hipLaunchKernelGGL(k_knn2_records, dim3((cap + 3) / 4, world), dim3(256), 0, s, d_recs, stride, rank, cap, off_desc, d_bi, d_bd, d_sd);
}

// DBoW2 transform: one lane per descriptor walks the tree (k children x L levels Hamming distances; the tree is L2-resident)
__global__ __launch_bounds__(256) void k_bow_transform(int n, const uint8_t* __restrict__ desc, const int32_t* __restrict__ child_begin,
                                                       const int32_t* __restrict__ child_count, const uint8_t* __restrict__ ndesc,
                                                       const int32_t* __restrict__ nword, const float* __restrict__ nweight, int levels, int levelsup,
                                                       int32_t* __restrict__ word_id, float* __restrict__ weight, int32_t* __restrict__ node_id)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const unsigned long long* f = reinterpret_cast<const unsigned long long*>(desc + (size_t)i * 32);
    const unsigned long long f0 = f[0], f1 = f[1], f2 = f[2], f3 = f[3];
    const int nid_level = levels - levelsup;
    int final_id = 0, level = 0, nid = 0;
    do {
        ++level;
        const int cb = child_begin[final_id], cc = child_count[final_id];
        int best = 0x7FFFFFFF;
        for (int c = cb; c < cb + cc; c++) {
            const unsigned long long* d = reinterpret_cast<const unsigned long long*>(ndesc + (size_t)c * 32);
            const int dist = __popcll(f0 ^ d[0]) + __popcll(f1 ^ d[1]) + __popcll(f2 ^ d[2]) + __popcll(f3 ^ d[3]);
            if (dist < best) { best = dist; final_id = c; }      // strict: the first minimum wins
        }
        if (level == nid_level) nid = final_id;
    } while (child_count[final_id] != 0 && level < 64);
    word_id[i] = nword[final_id]; weight[i] = nweight[final_id]; node_id[i] = nid;
}

void hs_launch_bow_transform(int n, const uint8_t* d_desc, const int32_t* d_cb, const int32_t* d_cc, const uint8_t* d_ndesc, const int32_t* d_word,
                             const float* d_weight, int levels, int levelsup, int32_t* d_out_word, float* d_out_weight, int32_t* d_out_node, hipStream_t s)
{
    if (n <= 0) return;
    hipLaunchKernelGGL(k_bow_transform, dim3((n + 255) / 256), dim3(256), 0, s, n, d_desc, d_cb, d_cc, d_ndesc, d_word, d_weight, levels, levelsup,
                       d_out_word, d_out_weight, d_out_node);
}

// SearchForInitialization: ONE workgroup, sequential over frame-1 keypoints, parallel inside each step.
// owner[i2] = frame-1 index currently matched to frame-2 keypoint i2 (-1 none), odist[i2] = its distance (global scratch, L2-resident).
__global__ __launch_bounds__(1024) void k_search_init(HsFrameDev F2, const uint8_t* __restrict__ desc1, int n1, const float* __restrict__ prev_xy,
                                                      float window, float th_low, float nnratio, int32_t* __restrict__ owner, int32_t* __restrict__ odist)
{
    __shared__ unsigned long long s_best[16];
    __shared__ int s_second[16];
    const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int i = tid; i < F2.n; i += 1024) { owner[i] = -1; odist[i] = -1; }
    __syncthreads();
    const float invW = (float)GRID_COLS / (F2.max_x - F2.min_x), invH = (float)GRID_ROWS / (F2.max_y - F2.min_y);
    const float r = window;
    for (int i1 = 0; i1 < n1; i1++) {
        const float x = prev_xy[2 * i1], y = prev_xy[2 * i1 + 1];
        const int minCX = max(0, (int)floorf(__fmul_rn(__fsub_rn(__fsub_rn(x, F2.min_x), r), invW)));
        const int maxCX = min(GRID_COLS - 1, (int)ceilf(__fmul_rn(__fadd_rn(__fsub_rn(x, F2.min_x), r), invW)));
        const int minCY = max(0, (int)floorf(__fmul_rn(__fsub_rn(__fsub_rn(y, F2.min_y), r), invH)));
        const int maxCY = min(GRID_ROWS - 1, (int)ceilf(__fmul_rn(__fadd_rn(__fsub_rn(y, F2.min_y), r), invH)));
        unsigned long long best = NO_KEY; int second = NO_DIST;
        if (!(minCX >= GRID_COLS || maxCX < 0 || minCY >= GRID_ROWS || maxCY < 0)) {
            const unsigned long long* d1 = reinterpret_cast<const unsigned long long*>(desc1 + (size_t)i1 * 32);
            const unsigned long long a0 = d1[0], a1 = d1[1], a2 = d1[2], a3 = d1[3];
            for (int i2 = tid; i2 < F2.n; i2 += 1024) {
                const int cx = F2.cell[2 * i2], cy = F2.cell[2 * i2 + 1];
                if (cx < minCX || cx > maxCX || cy < minCY || cy > maxCY) continue;
                const hs_keypoint kp = F2.kps[i2];
                if (!(fabsf(__fsub_rn(kp.x, x)) < r && fabsf(__fsub_rn(kp.y, y)) < r)) continue;
                const unsigned long long* dk = reinterpret_cast<const unsigned long long*>(F2.desc + (size_t)i2 * 32);
                const int d = __popcll(a0 ^ dk[0]) + __popcll(a1 ^ dk[1]) + __popcll(a2 ^ dk[2]) + __popcll(a3 ^ dk[3]);
                const int dp = odist[i2];
                if (dp >= 0 && !(d < dp)) continue;                         // MonoInitScoreExceedsPrevious
                const unsigned long long key = ((unsigned long long)d << 32) | ((unsigned long long)cx << 22) | ((unsigned long long)cy << 16) | (unsigned)i2;
                if (key < best) { second = min(second, (int)(best >> 32)); best = key; }
                else second = min(second, d);
            }
        }
        wave_best2(best, second);
        if (lane == 0) { s_best[wv] = best; s_second[wv] = second; }
        __syncthreads();
        if (tid == 0) {
            unsigned long long b = NO_KEY; int s2 = NO_DIST;
            for (int w = 0; w < 16; w++) {
                const unsigned long long ob = s_best[w]; const int os = s_second[w];
                const int worse = max((int)(b >> 32), (int)(ob >> 32));
                b = min(b, ob); s2 = min(min(s2, os), worse);
            }
            if (b != NO_KEY) {                                               // MonoInitBestScore accept rule
                const float bd = (float)(int)(b >> 32), bd2 = s2 == NO_DIST ? FLT_MAX : (float)s2;
                if (bd <= th_low && bd < __fmul_rn(bd2, nnratio)) { const int i2 = (int)(b & 0xFFFF); owner[i2] = i1; odist[i2] = (int)(b >> 32); }
            }
        }
        __syncthreads();
    }
}

void hs_launch_search_init(const hs_frame_view& F2, const hs_keypoint* d_kps2, const uint8_t* d_desc2, const int8_t* d_cell2,
                           const hs_keypoint* d_kps1, const uint8_t* d_desc1, int n1, const float* d_prev_xy, float window, float th_low, float nnratio,
                           int32_t* d_owner, int32_t* d_odist, float* d_angle_scratch, int32_t* d_self_scratch, int32_t* d_n_matches, hipStream_t s)
{
    HsFrameDev D{}; D.min_x = F2.min_x; D.max_x = F2.max_x; D.min_y = F2.min_y; D.max_y = F2.max_y; D.n = F2.n;
    D.kps = d_kps2; D.desc = d_desc2; D.cell = d_cell2;
    hipLaunchKernelGGL(k_search_init, dim3(1), dim3(1024), 0, s, D, d_desc1, n1, d_prev_xy, window, th_low, nnratio, d_owner, d_odist);
    // RotationConsistency(matches, views2, views1): entries keyed by the frame-2 index, rot = angle1[owner] - angle2[i2]
    const int g = (F2.n + 255) / 256;
    hipLaunchKernelGGL(k_gather_angle, dim3(g), dim3(256), 0, s, F2.n, d_owner, d_kps1, d_angle_scratch);
    hipLaunchKernelGGL(k_iota_where, dim3(g), dim3(256), 0, s, F2.n, d_owner, d_self_scratch);
    hipLaunchKernelGGL(k_rotation_filter, dim3(1), dim3(1024), 0, s, F2.n, d_self_scratch, d_angle_scratch, d_kps2, (int32_t*)nullptr, 0, 0, d_n_matches);
    hipLaunchKernelGGL(k_mask_by, dim3(g), dim3(256), 0, s, F2.n, d_self_scratch, d_owner);
}
