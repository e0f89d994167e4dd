// test_replay — hyslam_amd/host/HipAssociationReplay.h against host/cv_compat.h's RESTATEMENT of LandMarkMatches (src/core/LandMarkMatches.cpp:6-51 restated
// in cv_compat.h:160-183, not the reference's own translation unit: OpenCV is absent here; inside hySLAM the adaptor meets the reference's own struct): after plan_replay + the selected associateLandMark calls, views_to_landmarks, outliers and
// n_matches must equal what the reference's loop (FeatureMatcher.cc:113-118: one call per match, address order) leaves behind.
// Host only: no GPU, no C ABI.   usage: test_replay [cases] [seed]     prints "REPLAY OK ..." on success
#include <cstdio>
#include <cstdlib>
#include <memory>
#include <random>
#include <vector>
#include "../../hyslam_amd/host/HipAssociationReplay.h"

using namespace HYSLAM;

struct Holder {                                   // Frame's association surface (src/core/Frame.h:158, Frame.cc:216-219)
    LandMarkMatches matches;
    const LandMarkMatches& getLandMarkMatches() { return matches; }
    int associateLandMark(int i, MapPoint* p, bool replace) { ++calls; return matches.associateLandMark(i, p, replace); }
    long calls = 0;
};

static bool same(const LandMarkMatches& a, const LandMarkMatches& b)
{
    return a.views_to_landmarks == b.views_to_landmarks && a.outliers == b.outliers && a.n_matches == b.n_matches;
}

int main(int argc, char** argv)
{
    const int cases = argc > 1 ? atoi(argv[1]) : 3000;
    std::mt19937 rng(argc > 2 ? (unsigned)atoi(argv[2]) : 20251004u);
    auto rnd = [&](int n) { return n <= 0 ? 0 : (int)(rng() % (unsigned)n); };
    long by_rule[3] = { 0, 0, 0 }, calls_full = 0, calls_plan = 0, moved_cases = 0;
    for (int c = 0; c < cases; c++) {
        const int N = 8 + rnd(c % 7 == 0 ? 1500 : 120);                          // views
        const int M = 1 + rnd(c % 5 == 0 ? 8 * N : 2 * N);                         // landmarks offered to the search
        const int others = rnd(N / 2 + 1);                                         // landmarks that only sit in the frame
        std::vector<std::unique_ptr<MapPoint>> pool;
        for (int i = 0; i < M + others; i++) pool.emplace_back(new MapPoint());
        std::vector<MapPoint*> lms, rest;
        for (int i = 0; i < M + others; i++) (i < M ? lms : rest).push_back(pool[(size_t)i].get());
        std::shuffle(lms.begin(), lms.end(), rng);                                 // (allocation order is address order: which landmark gets which role must not follow it)
        std::sort(lms.begin(), lms.end());
        // ---- the frame before the search
        LandMarkMatches init;
        const int mode = c % 4;                                                    // 0: fresh frame (TrackLocalMap on a new frame), 1: sparse, 2: dense, 3: anything goes
        const int p_assoc = mode == 0 ? 0 : mode == 1 ? 15 : mode == 2 ? 70 : rnd(100);
        for (int v = 0; v < N; v++) {
            if (rnd(100) < p_assoc) {
                MapPoint* p = (!rest.empty() && rnd(100) < 50) ? rest[(size_t)rnd((int)rest.size())] : lms[(size_t)rnd(M)];      // a landmark may sit on several views
                init.views_to_landmarks[v] = p;
                if (rnd(100) < 90) init.outliers[v] = rnd(100) < 25;              // (an association without an outliers entry is possible too)
            } else if (mode != 0 && rnd(100) < 12) init.outliers[v] = rnd(100) < 60;   // a stale entry: what a "moves" erase leaves behind
        }
        init.n_matches = rnd(3 * N);
        // ---- the search's result: landmark k matched view midx[k]; popular views collect many landmarks
        std::vector<int32_t> midx((size_t)M, -1);
        const int hot = 1 + rnd(N);
        for (int k = 0; k < M; k++) {
            const int r = rnd(100);
            if (r < 55) midx[(size_t)k] = rnd(100) < 70 ? rnd(hot) : rnd(N);
            else if (r < 65) {                                                     // its own current view, or the view of another landmark
                for (const auto& kv : init.views_to_landmarks) if (kv.second == lms[(size_t)k] || rnd(40) == 0) { midx[(size_t)k] = kv.first; break; }
            }
        }
        if (c % 11 == 0) { init.views_to_landmarks[N + 5] = lms[0]; }              // a view index beyond the keypoints
        if (c % 97 == 0) { init.views_to_landmarks[-3] = lms[0]; }                 // a state the dense model refuses: full replay
        // ---- reference loop against plan + execute
        Holder full, lean;
        full.matches = init; lean.matches = init;
        for (int k = 0; k < M; k++) if (midx[(size_t)k] >= 0) full.associateLandMark(midx[(size_t)k], lms[(size_t)k], true);
        const hip_detail::ReplayPlan plan = hip_detail::replay_associations(lean, lms, midx);
        if (!same(full.matches, lean.matches)) {
            printf("MISMATCH in case %d (N %d, M %d, mode %d, rule %d): %zu views vs %zu, n_matches %d vs %d\n", c, N, M, mode, plan.rule,
                   full.matches.views_to_landmarks.size(), lean.matches.views_to_landmarks.size(), full.matches.n_matches, lean.matches.n_matches);
            return 1;
        }
        if ((long)plan.full_ops != full.calls || (long)plan.ops.size() != lean.calls || lean.calls > full.calls) { printf("call counts inconsistent in case %d\n", c); return 1; }
        by_rule[plan.rule]++; calls_full += full.calls; calls_plan += lean.calls;
        if (mode == 0 && c % 97 != 0 && c % 11 != 0 && plan.full_ops >= 2) {
            // a fresh frame: exactly one call per view that ends up associated
            if (plan.rule != 2 || plan.ops.size() != lean.matches.views_to_landmarks.size()) { printf("fresh frame: %zu calls for %zu views (rule %d), case %d\n", plan.ops.size(), lean.matches.views_to_landmarks.size(), plan.rule, c); return 1; }
        }
        for (const auto& op : plan.ops) (void)op;
        moved_cases += full.matches.views_to_landmarks.size() != init.views_to_landmarks.size();
    }
    // BASELINE config 4's shape: 50 000 landmarks, 14 000 matches onto 2 000 views of a fresh frame
    {
        const int N = 2000, M = 50000;
        std::vector<std::unique_ptr<MapPoint>> pool;
        for (int i = 0; i < M; i++) pool.emplace_back(new MapPoint());
        std::vector<MapPoint*> lms;
        for (auto& p : pool) lms.push_back(p.get());
        std::sort(lms.begin(), lms.end());
        std::vector<int32_t> midx((size_t)M, -1);
        for (int k = 0; k < M; k++) if (rnd(100) < 28) midx[(size_t)k] = rnd(N);
        Holder full, lean;
        for (int k = 0; k < M; k++) if (midx[(size_t)k] >= 0) full.associateLandMark(midx[(size_t)k], lms[(size_t)k], true);
        const hip_detail::ReplayPlan plan = hip_detail::replay_associations(lean, lms, midx);
        if (!same(full.matches, lean.matches) || plan.rule != 2 || plan.ops.size() != lean.matches.views_to_landmarks.size()) { printf("config-4 shape: mismatch or %zu calls (rule %d)\n", plan.ops.size(), plan.rule); return 1; }
        printf("config-4 shape: %ld calls of the full replay -> %ld\n", full.calls, lean.calls);
    }
    printf("REPLAY OK: %d cases, rules lean/safe/full = %ld/%ld/%ld, associateLandMark calls %ld -> %ld, %ld cases changed the set of associated views\n",
           cases, by_rule[2], by_rule[1], by_rule[0], calls_full, calls_plan, moved_cases);
    return 0;
}
