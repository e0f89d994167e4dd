// kernels_bow.hip — device-resident vocabulary: Frame::ComputeBoW and the BoW-grouped matcher without host round trips.
//
//   hs_vocab_upload              the flat vocabulary tree (hs_vocab_tree) copied to HBM once, plus the dense rank ("group") of every node at
//                                the feature-vector level L - levelsup (Frame.cc:472-479 calls transform(..., levelsup = 4))
//   hs_bow_transform_device      DBoW2 transform of descriptors that already live in HBM (the extractor's outputs)
//   hs_records_bow_match_device  BASELINE config 5 "cross-camera BoW match": for every gathered frame record p != rank the matching core of
//                                SearchByBoW / _SearchByBoW_ (src/features/FeatureMatcher.cc:216-345, BestMatchBoWCriterion MatchCriteria.cpp:601-635,
//                                RotationConsistencyBoW :679-726) between record `rank` (side 1) and record p (side 2)
//
// The reference walks two std::map<NodeId, vector<unsigned>> (DBoW2::FeatureVector) in step.  Here both sides are bucketed by the dense group of
// their feature-vector node (one counting sort per record in LDS), so "node present on both sides" is simply "both buckets non-empty" and the
// merge walk disappears.  Inside a node the reference scans side 2 in ascending index order and the first minimum wins: the candidate key is
// dist<<32 | side-2 index, which makes the bucket's internal order irrelevant.  Features whose word weight is not positive are left out of the
// feature vector, as in DBoW2 (`if (w > 0) fv.addFeature(nid, i)`).
#include "hs_internal.h"
#include <algorithm>
#include <cfloat>
#include <vector>

#define BOW_NO_KEY 0x7FFFFFFFFFFFFFFFull
#define BOW_NO_DIST 0x7FFFFFFF
#define BOW_MAX_GROUPS 8192

struct hs_vocab_dev {
    int device = 0, n_nodes = 0, levels = 0, levelsup = 0, groups = 0;
    int32_t *d_cb = nullptr, *d_cc = nullptr, *d_word = nullptr, *d_group = nullptr, *d_report = nullptr; float* d_weight = nullptr; uint8_t* d_desc = nullptr;
    // scratch of hs_records_bow_match_device (grow-only): group of every feature, bucket starts and bucket items per record
    int32_t* d_fgroup = nullptr; int32_t* d_start = nullptr; uint16_t* d_items = nullptr; size_t cap_feats = 0, cap_records = 0;
};

struct BowTree { const int32_t* cb; const int32_t* cc; const uint8_t* desc; const int32_t* word; const float* weight; const int32_t* group; const int32_t* report; int nid_level; };

// DBoW2 transform of one descriptor: word (leaf) and the node passed at level nid_level (0 = root when nid_level <= 0)
__device__ __forceinline__ void bow_descend(const BowTree& T, const uint8_t* f32, int& leaf, int& nid)
{
    const unsigned long long* f = reinterpret_cast<const unsigned long long*>(f32);
    const unsigned long long f0 = f[0], f1 = f[1], f2 = f[2], f3 = f[3];
    int final_id = 0, level = 0; nid = 0;
    do {
        ++level;
        const int cb = T.cb[final_id], cc = T.cc[final_id];
        int best = 0x7FFFFFFF;
        for (int c = cb; c < cb + cc; c++) {
            const unsigned long long* d = reinterpret_cast<const unsigned long long*>(T.desc + (size_t)c * 32);
            const int dist = __popcll(f0 ^ d[0]) + __popcll(f1 ^ d[1]) + __popcll(f2 ^ d[2]) + __popcll(f3 ^ d[3]);
            if (dist < best) { best = dist; final_id = c; }      // strict: the first minimum wins
        }
        if (level == T.nid_level) nid = final_id;
    } while (T.cc[final_id] != 0 && level < 64);
    leaf = final_id;
}

__global__ __launch_bounds__(256) void k_bow_transform_dev(BowTree T, const uint8_t* __restrict__ desc, const int32_t* __restrict__ d_n, int n_max,
                                                           int32_t* __restrict__ word_id, float* __restrict__ weight, int32_t* __restrict__ node_id)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    const int n = d_n ? min(max(*d_n, 0), n_max) : n_max;
    if (i >= n) return;
    int leaf, nid;
    bow_descend(T, desc + (size_t)i * 32, leaf, nid);
    word_id[i] = T.word[leaf]; weight[i] = T.weight[leaf]; node_id[i] = T.report ? T.report[nid] : nid;
}

// group of every feature of every record (-1: not in the feature vector)
__global__ __launch_bounds__(256) void k_records_groups(BowTree T, const uint8_t* __restrict__ recs, size_t stride, int cap, size_t off_desc, int32_t* __restrict__ fgroup)
{
    const int r = blockIdx.y, i = blockIdx.x * 256 + threadIdx.x;
    const int n = min(max(hs_gload<int32_t>(recs + (size_t)r * stride), 0), cap);
    if (i >= n) return;
    int leaf, nid;
    bow_descend(T, recs + (size_t)r * stride + off_desc + (size_t)i * 32, leaf, nid);
    fgroup[(size_t)r * cap + i] = T.weight[leaf] > 0.f ? T.group[nid] : -1;
}

// one workgroup per record: bucket the features by group (counting sort in LDS)
__global__ __launch_bounds__(1024) void k_records_buckets(const uint8_t* __restrict__ recs, size_t stride, int cap, int groups, const int32_t* __restrict__ fgroup,
                                                          int32_t* __restrict__ start /*[world][groups+1]*/, uint16_t* __restrict__ items /*[world][cap]*/)
{
    __shared__ uint32_t cnt[BOW_MAX_GROUPS];
    __shared__ uint32_t s_wave[16];
    const int r = blockIdx.x, tid = threadIdx.x;
    const int n = min(max(hs_gload<int32_t>(recs + (size_t)r * stride), 0), cap);
    const int32_t* fg = fgroup + (size_t)r * cap;
    for (int g = tid; g < groups; g += 1024) cnt[g] = 0;
    __syncthreads();
    for (int i = tid; i < n; i += 1024) { const int g = fg[i]; if (g >= 0) atomicAdd(&cnt[g], 1u); }
    __syncthreads();
    // exclusive scan of cnt[0..groups): PER consecutive groups per thread
    const int PER = (groups + 1023) / 1024;
    uint32_t sum = 0;
    for (int k = 0; k < PER; k++) { const int g = tid * PER + k; if (g < groups) sum += cnt[g]; }
    uint32_t incl = sum;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const uint32_t v = __shfl_up(incl, o, 64); if ((tid & 63) >= o) incl += v; }
    if ((tid & 63) == 63) s_wave[tid >> 6] = incl;
    __syncthreads();
    uint32_t base = 0, total = 0;
#pragma unroll
    for (int w = 0; w < 16; w++) { const uint32_t x = s_wave[w]; if (w < (tid >> 6)) base += x; total += x; }
    uint32_t run = base + incl - sum;
    int32_t* st = start + (size_t)r * (groups + 1);
    for (int k = 0; k < PER; k++) { const int g = tid * PER + k; if (g < groups) { const uint32_t c = cnt[g]; st[g] = (int32_t)run; cnt[g] = run; run += c; } }
    if (tid == 0) st[groups] = (int32_t)total;
    __syncthreads();
    uint16_t* it = items + (size_t)r * cap;
    for (int i = tid; i < n; i += 1024) { const int g = fg[i]; if (g >= 0) it[atomicAdd(&cnt[g], 1u)] = (uint16_t)i; }
}

__device__ __forceinline__ void bow_wave_best2(unsigned long long& best, int& second)
{
#pragma unroll
    for (int s = 32; s > 0; s >>= 1) {
        const unsigned long long ob = __shfl_xor(best, s, 64);
        const int os = __shfl_xor(second, s, 64);
        const int worse = max((int)(best >> 32), (int)(ob >> 32));
        best = min(best, ob);
        second = min(min(second, os), worse);
    }
}

// blockIdx.x = group, blockIdx.y = peer record: every side-1 feature of the group against the peer's features of the same group
__global__ __launch_bounds__(256) void k_records_bow_match(const uint8_t* __restrict__ recs, size_t stride, int rank, int cap, size_t off_desc, int groups,
                                                           const int32_t* __restrict__ start, const uint16_t* __restrict__ items,
                                                           float score_threshold, float ratio, int32_t* __restrict__ match12)
{
    const int g = blockIdx.x, peer = blockIdx.y;
    if (peer == rank) return;
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int32_t* s1 = start + (size_t)rank * (groups + 1); const int32_t* s2 = start + (size_t)peer * (groups + 1);
    const int p0 = s1[g], p1 = s1[g + 1], q0 = s2[g], q1 = s2[g + 1];
    if (p0 == p1 || q0 == q1) return;
    const uint16_t* it1 = items + (size_t)rank * cap; const uint16_t* it2 = items + (size_t)peer * cap;
    const uint8_t* d1 = recs + (size_t)rank * stride + off_desc; const uint8_t* d2 = recs + (size_t)peer * stride + off_desc;
    for (int p = p0 + wv; p < p1; p += 4) {
        const int i1 = it1[p];
        const unsigned long long* a = reinterpret_cast<const unsigned long long*>(d1 + (size_t)i1 * 32);
        const unsigned long long a0 = a[0], a1 = a[1], a2 = a[2], a3 = a[3];
        unsigned long long best = BOW_NO_KEY; int second = BOW_NO_DIST;
        for (int q = q0 + lane; q < q1; q += 64) {
            const int i2 = it2[q];
            const unsigned long long* b = reinterpret_cast<const unsigned long long*>(d2 + (size_t)i2 * 32);
            const int d = __popcll(a0 ^ b[0]) + __popcll(a1 ^ b[1]) + __popcll(a2 ^ b[2]) + __popcll(a3 ^ b[3]);
            const unsigned long long key = ((unsigned long long)d << 32) | (unsigned)i2;      // ascending side-2 index breaks ties, like the reference's scan
            if (key < best) { second = min(second, (int)(best >> 32)); best = key; }
            else second = min(second, d);
        }
        bow_wave_best2(best, second);
        if (lane == 0 && best != BOW_NO_KEY) {
            const float bd1 = (float)(int)(best >> 32), bd2 = second == BOW_NO_DIST ? FLT_MAX : (float)second;
            if (bd1 < score_threshold && bd1 < __fmul_rn(ratio, bd2)) match12[(size_t)peer * cap + i1] = (int)(best & 0xFFFFFFFFu);
        }
    }
}

// one workgroup per peer: RotationConsistencyBoW (MatchCriteria.cpp:679-767) + the number of surviving matches
__global__ __launch_bounds__(1024) void k_records_rotation(const uint8_t* __restrict__ recs, size_t stride, int rank, int cap, size_t off_kps, int check_rotation,
                                                           int32_t* __restrict__ match12, int32_t* __restrict__ n_matches)
{
    __shared__ int hist[30];
    __shared__ int ind[3];
    __shared__ int total;
    const int peer = blockIdx.x, tid = threadIdx.x;
    if (peer == rank) { if (tid == 0) n_matches[peer] = 0; return; }
    const int n = min(max(hs_gload<int32_t>(recs + (size_t)rank * stride), 0), cap);
    const hs_keypoint* k1 = reinterpret_cast<const hs_keypoint*>(recs + (size_t)rank * stride + off_kps);
    const hs_keypoint* k2 = reinterpret_cast<const hs_keypoint*>(recs + (size_t)peer * stride + off_kps);
    int32_t* m = match12 + (size_t)peer * cap;
    if (tid < 30) hist[tid] = 0;
    if (tid == 0) total = 0;
    __syncthreads();
    auto bin_of = [&](int i) {
        float rot = __fsub_rn(k2[m[i]].angle, k1[i].angle);
        if (rot < 0.0f) rot = __fadd_rn(rot, 360.0f);
        const int b = (int)roundf(__fmul_rn(rot, 1.0f / 30));
        return b == 30 ? 0 : b;
    };
    if (check_rotation) {
        for (int i = tid; i < n; i += 1024) if (m[i] >= 0) { const int b = bin_of(i); if (b >= 0 && b < 30) atomicAdd(&hist[b], 1); }
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
    }
    int kept = 0;
    for (int i = tid; i < n; i += 1024) {
        if (m[i] < 0) continue;
        bool keep = true;
        if (check_rotation) { const int b = bin_of(i); keep = (b == ind[0] || b == ind[1] || b == ind[2]); }
        if (!keep) m[i] = -1; else kept++;
    }
    atomicAdd(&total, kept);
    __syncthreads();
    if (tid == 0) n_matches[peer] = total;
}

namespace {
BowTree tree_of(const hs_vocab_dev* v)
{
    BowTree T; T.cb = v->d_cb; T.cc = v->d_cc; T.desc = v->d_desc; T.word = v->d_word; T.weight = v->d_weight; T.group = v->d_group; T.report = v->d_report;
    T.nid_level = v->levels - v->levelsup;
    return T;
}
}

// ---- host side (declared in include/hyslam_amd.h); error text goes through hs_api.hip's handle via hs_set_error
void hs_set_error(hs_orb* h, const char* msg);       // hs_api.hip
int hs_orb_device_of(const hs_orb* h);              // hs_api.hip
hipStream_t hs_orb_stream_of(const hs_orb* h);      // hs_api.hip

extern "C" {

int hs_vocab_upload(hs_orb* h, const hs_vocab_tree* T, int levelsup, hs_vocab_dev** out)
{
    if (!h || !T || !out || T->n_nodes < 2 || T->levels < 1 || !T->child_begin || !T->child_count || !T->desc || !T->word_id || !T->weight) return HS_ERR_INVALID;
    *out = nullptr;
    const int n = T->n_nodes;
    if (T->child_count[0] < 1) { hs_set_error(h, "vocabulary root has no children"); return HS_ERR_INVALID; }
    std::vector<int> level(n, 0);
    for (int i = 0; i < n; i++) {
        const long cb = T->child_begin[i], cc = T->child_count[i];
        if (cc < 0 || (cc > 0 && (cb <= i || cb + cc > n))) { hs_set_error(h, "vocabulary tree is not a forward-linked flat tree"); return HS_ERR_INVALID; }
        for (long c = cb; c < cb + cc; c++) level[c] = level[i] + 1;
    }
    // dense ranks of the feature-vector nodes, ascending in the id DBoW2 reports (the order of its std::map)
    const int nid_level = T->levels - levelsup;
    std::vector<std::pair<int, int>> nodes;          // (reported id, flat index)
    if (nid_level <= 0) nodes.push_back({ T->orig_id ? T->orig_id[0] : 0, 0 });
    else for (int i = 0; i < n; i++) if (level[i] == nid_level) nodes.push_back({ T->orig_id ? T->orig_id[i] : i, i });
    // a walk that reaches a leaf above nid_level reports node 0 (DBoW2 leaves *nid untouched = 0): the root gets a group too
    bool shallow = false;
    for (int i = 1; i < n; i++) if (T->child_count[i] == 0 && level[i] < nid_level) shallow = true;
    if (shallow && nid_level > 0) nodes.push_back({ T->orig_id ? T->orig_id[0] : 0, 0 });
    std::sort(nodes.begin(), nodes.end());
    if ((int)nodes.size() > BOW_MAX_GROUPS) { hs_set_error(h, "too many feature-vector nodes at this level (levelsup too small)"); return HS_ERR_INVALID; }
    std::vector<int32_t> group(n, -1);
    for (size_t g = 0; g < nodes.size(); g++) group[nodes[g].second] = (int32_t)g;
    hs_vocab_dev* v = new hs_vocab_dev();
    v->device = hs_orb_device_of(h); v->n_nodes = n; v->levels = T->levels; v->levelsup = levelsup; v->groups = (int)nodes.size();
    hipError_t e = hipSetDevice(v->device);
    auto up = [&](void** d, const void* src, size_t bytes) { if (e == hipSuccess) e = hipMalloc(d, bytes); if (e == hipSuccess) e = hipMemcpy(*d, src, bytes, hipMemcpyHostToDevice); };
    up((void**)&v->d_cb, T->child_begin, (size_t)n * 4); up((void**)&v->d_cc, T->child_count, (size_t)n * 4); up((void**)&v->d_word, T->word_id, (size_t)n * 4);
    up((void**)&v->d_weight, T->weight, (size_t)n * 4); up((void**)&v->d_desc, T->desc, (size_t)n * 32); up((void**)&v->d_group, group.data(), (size_t)n * 4);
    if (T->orig_id) up((void**)&v->d_report, T->orig_id, (size_t)n * 4);
    if (e != hipSuccess) { (void)hipGetLastError(); hs_set_error(h, hipGetErrorString(e)); hs_vocab_dev_destroy(v); return HS_ERR_HIP; }
    *out = v;
    return HS_OK;
}

void hs_vocab_dev_destroy(hs_vocab_dev* v)
{
    if (!v) return;
    (void)hipSetDevice(v->device);
    hipFree(v->d_cb); hipFree(v->d_cc); hipFree(v->d_word); hipFree(v->d_weight); hipFree(v->d_desc); hipFree(v->d_group); hipFree(v->d_report);
    hipFree(v->d_fgroup); hipFree(v->d_start); hipFree(v->d_items);
    delete v;
}

int hs_vocab_dev_groups(const hs_vocab_dev* v) { return v ? v->groups : 0; }

int hs_bow_transform_device(hs_orb* h, const hs_vocab_dev* v, const uint8_t* d_desc, const int32_t* d_n, int n_max,
                            int32_t* d_word, float* d_weight, int32_t* d_node, void* stream)
{
    if (!h || !v) return HS_ERR_INVALID;
    if (n_max < 0 || (n_max > 0 && (!d_desc || !d_word || !d_weight || !d_node)) || v->device != hs_orb_device_of(h)) { hs_set_error(h, "bad argument"); return HS_ERR_INVALID; }
    if (n_max == 0) return HS_OK;
    if (hipSetDevice(v->device) != hipSuccess) return HS_ERR_HIP;
    hipStream_t s = stream ? (hipStream_t)stream : hs_orb_stream_of(h);
    hipLaunchKernelGGL(k_bow_transform_dev, dim3((n_max + 255) / 256), dim3(256), 0, s, tree_of(v), d_desc, d_n, n_max, d_word, d_weight, d_node);
    if (hipGetLastError() != hipSuccess) { hs_set_error(h, "k_bow_transform_dev launch failed"); return HS_ERR_HIP; }
    return HS_OK;
}

int hs_records_bow_match_device(hs_orb* h, hs_vocab_dev* v, const uint8_t* d_records, size_t record_stride, int world, int rank, int cap,
                                float score_threshold, float second_best_ratio, int check_rotation,
                                int32_t* d_match12, int32_t* d_n_matches, void* stream)
{
    if (!h || !v) return HS_ERR_INVALID;
    if (!d_records || world < 1 || world > 65535 || rank < 0 || rank >= world || cap < 1 || cap > 65535 || record_stride < hs_record_bytes(cap) ||
        (record_stride & 3) || ((uintptr_t)d_records & 15) || !d_match12 || !d_n_matches || v->device != hs_orb_device_of(h)) { hs_set_error(h, "bad argument"); return HS_ERR_INVALID; }
    if (hipSetDevice(v->device) != hipSuccess) return HS_ERR_HIP;
    hipStream_t s = stream ? (hipStream_t)stream : hs_orb_stream_of(h);
    const size_t feats = (size_t)world * cap;
    if (feats > v->cap_feats || (size_t)world > v->cap_records) {             // scratch grows; steady state allocates nothing
        // the scratch belongs to the vocabulary object, which callers may have used on another stream before: drain the whole device (rare path)
        if (hipDeviceSynchronize() != hipSuccess) return HS_ERR_HIP;
        hipFree(v->d_fgroup); hipFree(v->d_start); hipFree(v->d_items); v->d_fgroup = nullptr; v->d_start = nullptr; v->d_items = nullptr; v->cap_feats = v->cap_records = 0;
        if (hipMalloc(&v->d_fgroup, feats * 4) != hipSuccess || hipMalloc(&v->d_items, feats * 2) != hipSuccess ||
            hipMalloc(&v->d_start, (size_t)world * (v->groups + 1) * 4) != hipSuccess) { (void)hipGetLastError(); hs_set_error(h, "out of device memory"); return HS_ERR_HIP; }
        v->cap_feats = feats; v->cap_records = world;
    }
    size_t off_kps, off_desc; hs_record_offsets(cap, nullptr, &off_kps, &off_desc);
    const BowTree T = tree_of(v);
    hipLaunchKernelGGL(k_records_groups, dim3((cap + 255) / 256, world), dim3(256), 0, s, T, d_records, record_stride, cap, off_desc, v->d_fgroup);
    hipLaunchKernelGGL(k_records_buckets, dim3(world), dim3(1024), 0, s, d_records, record_stride, cap, v->groups, v->d_fgroup, v->d_start, v->d_items);
    (void)hipMemsetAsync(d_match12, 0xFF, feats * 4, s);
    hipLaunchKernelGGL(k_records_bow_match, dim3(v->groups, world), dim3(256), 0, s, d_records, record_stride, rank, cap, off_desc, v->groups, v->d_start, v->d_items,
                       score_threshold, second_best_ratio, d_match12);
    hipLaunchKernelGGL(k_records_rotation, dim3(world), dim3(1024), 0, s, d_records, record_stride, rank, cap, off_kps, check_rotation, d_match12, d_n_matches);
    if (hipGetLastError() != hipSuccess) { hs_set_error(h, "BoW record matcher launch failed"); return HS_ERR_HIP; }
    return HS_OK;
}

} // extern "C"
