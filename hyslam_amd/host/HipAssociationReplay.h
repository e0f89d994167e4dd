// HipAssociationReplay.h — the association replay of _SearchByProjection_ (src/features/FeatureMatcher.cc:113-118) with as few
// Frame::associateLandMark calls as give the SAME LandMarkMatches.
//
// The reference ends a projection search with
//     for (auto it = matches.begin(); it != matches.end(); ++it) F.associateLandMark(idx, lm, true);      // std::map<MapPoint*, ...>: address order
// and every associateLandMark call scans the frame's whole view -> landmark map for the landmark (LandMarkMatches::hasAssociation(MapPoint*),
// src/core/LandMarkMatches.cpp:17-24).  TrackLocalMap's search at BASELINE config 4 (50 000 landmarks, 2 000 keypoints) returns ~14 000 matches
// onto ~2 000 views: 14 000 scans of a 2 000-entry std::map = 23.5 of the call's 29 ms (profiles/r04_bench_lines.json -> adaptor), although at
// most 2 000 views can end up associated.  The adaptor cannot change LandMarkMatches, but it chooses how many calls it makes.
//
// What one call (view i, landmark p, replace = true) does (LandMarkMatches.cpp:26-51), with A = the landmark at view i before the call and
// j = the FIRST view (ascending) that holds p, or -1:
//     fresh    A == null and j < 0:   insert (i, p); outliers.insert({i, false}) — no overwrite of a stale entry —; ++n_matches
//     replace  otherwise:             views[i] = p; outliers[i] = false; if j >= 0 and j != i: erase view j ("moves": its outliers entry stays,
//                                     n_matches is NOT decremented)
// So the final state (views_to_landmarks, outliers, n_matches) depends on the path, not only on "last writer per view".  plan_replay():
//   1. runs the FULL replay on a model of that state (dense arrays by view index; every landmark occurs in at most one op, so "where is p now"
//      is "which of p's INITIAL views still holds it": O(1) per op, ~20 us for 14 000 ops);
//   2. selects a subsequence of the ops:
//        kept always      the LAST op of every view (it decides views[i]); every op that MOVES its landmark (it erases another view)
//        kept when needed the FIRST op of a view: it is what makes the later ops of that view "replace" calls.  It can go when the view was
//                         empty, it was a fresh insert, no kept op of the view moves its landmark (the last op then becomes the fresh insert:
//                         the same ++n_matches) and no stale outliers entry is in the way; or when the view held a landmark X whose own op
//                         (if any) does not fall between the first and the last op of the view;
//        dropped          everything in between: such an op finds the view taken (replace), writes a landmark that the last op overwrites, and
//                         its landmark — which no other op mentions — was nowhere else;
//   3. runs the SELECTED ops on a second copy of the model and compares the two final states field by field.  Equal -> the plan is used.  Not equal
//      (cannot happen for the rule "first + last + moving ops"; the lean rule for the first op is what the check is for) -> the next safer rule,
//      at the end the full replay.  The reduction is therefore self-verifying: a plan that is executed has been shown, on the model, to end in
//      the state the reference's loop ends in.
// tests/cpp/test_replay.cpp drives it against the real LandMarkMatches on randomised states (stale outliers entries, landmarks that sit on several
// views, landmarks already on their target view, matched landmarks that move) and checks views_to_landmarks, outliers and n_matches after
// plan + execute against the full replay.
#pragma once
#ifdef HYSLAM_AMD_WITH_HYSLAM
#include <LandMarkMatches.h>
#include <MapPoint.h>
#else
#include "cv_compat.h"
#endif
#include <algorithm>
#include <cstring>
#include <cstdint>
#include <vector>

namespace HYSLAM {
namespace hip_detail {

struct AssocOp { int32_t view; uint32_t lm; };           // associateLandMark(view, lms[lm], true)

struct AssocState {                                       // LandMarkMatches as dense arrays by view index
    std::vector<MapPoint*> lm;                            // views_to_landmarks: nullptr = no entry
    std::vector<uint8_t> outl;                            // outliers: 0 = no entry, 1 = false, 2 = true
    long n_matches = 0;
    bool operator==(const AssocState& o) const { return n_matches == o.n_matches && lm == o.lm && outl == o.outl; }
};

enum : unsigned { ASSOC_FRESH = 1u, ASSOC_MOVES = 2u };

// one associateLandMark(i, p, true) on the model; init_views = the views that held p BEFORE the replay, ascending (p is placed only by its own op,
// and no op after that one mentions p: "the first view that holds p" is the first initial view that still does)
inline unsigned assoc_apply(AssocState& S, int i, MapPoint* p, const int32_t* init_views, int n_init)
{
    int j = -1;
    for (int q = 0; q < n_init; q++) if (S.lm[init_views[q]] == p) { j = init_views[q]; break; }
    if (!S.lm[i] && j < 0) { S.lm[i] = p; if (!S.outl[i]) S.outl[i] = 1; ++S.n_matches; return ASSOC_FRESH; }
    S.lm[i] = p; S.outl[i] = 1;
    if (j >= 0 && j != i) { S.lm[j] = nullptr; return ASSOC_MOVES; }
    return 0u;
}

struct ReplayPlan {
    std::vector<AssocOp> ops;          // the calls to make, in order
    int rule = 0;                      // 2 = lean (last + moving + needed first ops), 1 = first + last + moving ops, 0 = the full replay
    size_t full_ops = 0;               // calls of the full replay
};

// lms: the landmarks in replay order (sorted by address, unique: the reference's std::map keys); midx[k] = the view landmark k matched, or -1.
// M = the frame's LandMarkMatches BEFORE the replay.
template <class Matches>
inline ReplayPlan plan_replay(const Matches& M, const std::vector<MapPoint*>& lms, const std::vector<int32_t>& midx)
{
    ReplayPlan plan;
    std::vector<AssocOp> all;
    int max_view = -1;
    for (size_t k = 0; k < lms.size(); k++)
        if (midx[k] >= 0) { all.push_back(AssocOp{ midx[k], (uint32_t)k }); max_view = std::max(max_view, (int)midx[k]); }
    plan.full_ops = all.size();
    auto full = [&]() { plan.ops = all; plan.rule = 0; return plan; };
    if (all.size() < 2) return full();
    // ---- the model's start state; states the dense model does not cover (entries with a null landmark, absurd view indices) -> full replay
    bool modelled = true;
    for (const auto& kv : M.views_to_landmarks) { if (kv.first < 0 || kv.first > (1 << 22) || !kv.second) modelled = false; max_view = std::max(max_view, kv.first); }
    for (const auto& kv : M.outliers) { if (kv.first < 0 || kv.first > (1 << 22)) modelled = false; max_view = std::max(max_view, kv.first); }
    if (!modelled) return full();
    AssocState S0;
    S0.lm.assign((size_t)max_view + 1, nullptr); S0.outl.assign((size_t)max_view + 1, 0); S0.n_matches = M.n_matches;
    for (const auto& kv : M.outliers) S0.outl[kv.first] = kv.second ? 2 : 1;
    // initial views of every op landmark, CSR over the landmark index (views ascending: std::map order)
    std::vector<int32_t> iv_ptr(lms.size() + 1, 0), iv;
    {
        std::vector<std::pair<uint32_t, int32_t>> found;      // (landmark index, view)
        for (const auto& kv : M.views_to_landmarks) {
            S0.lm[kv.first] = kv.second;
            auto it = std::lower_bound(lms.begin(), lms.end(), kv.second);
            if (it != lms.end() && *it == kv.second) found.push_back({ (uint32_t)(it - lms.begin()), kv.first });
        }
        std::stable_sort(found.begin(), found.end(), [](const std::pair<uint32_t, int32_t>& a, const std::pair<uint32_t, int32_t>& b) { return a.first < b.first; });
        for (const auto& f : found) iv_ptr[f.first + 1]++;
        for (size_t k = 0; k < lms.size(); k++) iv_ptr[k + 1] += iv_ptr[k];
        iv.resize(found.size());
        for (size_t q = 0; q < found.size(); q++) iv[q] = found[q].second;      // already grouped by landmark, views ascending within a group
    }
    auto apply = [&](AssocState& S, const AssocOp& op) { return assoc_apply(S, op.view, lms[op.lm], iv.data() + iv_ptr[op.lm], iv_ptr[op.lm + 1] - iv_ptr[op.lm]); };
    // ---- 1. the full replay on the model
    const int32_t NONE = -1;
    std::vector<int32_t> first_op((size_t)max_view + 1, NONE), last_op((size_t)max_view + 1, NONE);
    std::vector<uint8_t> flags(all.size(), 0);
    std::vector<MapPoint*> held_at_first((size_t)max_view + 1, nullptr);      // A0: the landmark at the view when its first op ran
    std::vector<uint8_t> stale_true((size_t)max_view + 1, 0);                 // outliers[view] == true with no landmark at the view, when its first op ran
    std::vector<uint8_t> has_moves((size_t)max_view + 1, 0);                  // some op of the view moves its landmark (such an op is kept, and it is a "replace" call whatever it finds)
    AssocState Sf = S0;
    for (size_t o = 0; o < all.size(); o++) {
        const int v = all[o].view;
        if (first_op[v] == NONE) { first_op[v] = (int32_t)o; held_at_first[v] = Sf.lm[v]; stale_true[v] = !Sf.lm[v] && Sf.outl[v] == 2; }
        last_op[v] = (int32_t)o;
        flags[o] = (uint8_t)apply(Sf, all[o]);
        if (flags[o] & ASSOC_MOVES) has_moves[v] = 1;
    }
    // op index of a landmark (by address), or -1
    std::vector<int32_t> op_of_lm(lms.size(), NONE);
    for (size_t o = 0; o < all.size(); o++) op_of_lm[all[o].lm] = (int32_t)o;
    auto op_of = [&](MapPoint* x) -> int32_t {
        auto it = std::lower_bound(lms.begin(), lms.end(), x);
        return (it != lms.end() && *it == x) ? op_of_lm[it - lms.begin()] : NONE;
    };
    // ---- 2 + 3. select, simulate, compare
    for (int rule = 2; rule >= 1; rule--) {
        std::vector<AssocOp> sel;
        for (size_t o = 0; o < all.size(); o++) {
            const int v = all[o].view;
            bool keep = last_op[v] == (int32_t)o || (flags[o] & ASSOC_MOVES);
            if (!keep && first_op[v] == (int32_t)o) {
                if (rule == 1) keep = true;
                else if (!held_at_first[v]) keep = !((flags[o] & ASSOC_FRESH) && !has_moves[v] && !stale_true[v]);
                else { const int32_t ox = op_of(held_at_first[v]); keep = ox > (int32_t)o && ox < last_op[v]; }
            }
            if (keep) sel.push_back(all[o]);
        }
        AssocState Sr = S0;
        for (const AssocOp& op : sel) apply(Sr, op);
        if (Sr == Sf) { plan.ops.swap(sel); plan.rule = rule; return plan; }
    }
    return full();
}

// plan + execute on the real frame
template <class FrameT>
inline ReplayPlan replay_associations(FrameT& F, const std::vector<MapPoint*>& lms, const std::vector<int32_t>& midx)
{
    ReplayPlan plan = plan_replay(F.getLandMarkMatches(), lms, midx);
    for (const AssocOp& op : plan.ops) F.associateLandMark(op.view, lms[op.lm], true);
    return plan;
}

}  // namespace hip_detail
}  // namespace HYSLAM
