// HipFeatureMatcher.h — HYSLAM::FeatureMatcher's search entry points over the C ABI (include/hyslam_amd.h).
//
//   HYSLAM::HipMatcherCore                                 the twelve searches as a plain class (settings + a C-ABI handle)
//   HYSLAM::HipFeatureMatcher : HYSLAM::FeatureMatcher     (src/features/FeatureMatcher.h:105-176) — overrides that forward to the core
//
// Two ways into hySLAM:
//   (a) the `virtual` patch: `virtual` on FeatureFactory::getFeatureMatcher and on the FeatureMatcher search methods — the reference obtains
//       matchers through the NON-virtual FeatureFactory::getFeatureMatcher() (src/features/FeatureFactory.cpp:7-9) and calls non-virtual methods, so
//       without the patch a subclass is never reached — and HipORBFactory hands out HipFeatureMatcher;
//   (b) NO header edit: host/replace/FeatureMatcher.cc replaces src/features/FeatureMatcher.cc in hySLAM's source list and defines the
//       reference's own FeatureMatcher member functions as calls into HipMatcherCore (define HYSLAM_AMD_UNPATCHED_MATCHER: HipFeatureMatcher
//       cannot exist then — nothing to override).
// Every search does three things:
//   gather   Frame / KeyFrame / MapPoint fields -> the flat arrays of hs_frame_view / hs_landmark (one pass, no per-landmark map copies);
//            landmarks are passed SORTED BY ADDRESS, which reproduces the iteration order of the reference's std::map<MapPoint*, ...>
//            (FeatureMatcher.cc:64,113-118; deviation D6 of DESIGN.md)
//   search   one C-ABI call (GPU)
//   replay   Frame::associateLandMark(idx, lm, true) in that same address order (FeatureMatcher.cc:113-118) — a later landmark that picked the same
//            keypoint overwrites the earlier one, exactly as in the reference — but only the calls that matter: HipAssociationReplay.h plans the
//            shortest subsequence that leaves views_to_landmarks, outliers and n_matches as the full loop would (checked on a model per call).
// Overridden: all twelve search entry points of FeatureMatcher (src/features/FeatureMatcher.h:114-150) — the four SearchByProjection overloads,
// SearchByBoW(KF, Frame), SearchByBoW(KF, KF) (legacy: no call site in hySLAM, "aim to replace this with SearchByBoW2", FeatureMatcher.cc:939),
// SearchByBoW2, SearchForTriangulation, SearchForInitialization, Fuse(pKF, points, ...), Fuse(pKF, Scw, ...) (an empty body in the reference,
// :523-624) and SearchBySim3.
#pragma once
#ifdef HYSLAM_AMD_WITH_HYSLAM
#include <FeatureMatcher.h>
#include <Frame.h>
#include <KeyFrame.h>
#include <MapPoint.h>
#else
#include "cv_compat.h"
#endif
#include <algorithm>
#include <cstring>
#include <cstdlib>
#include <future>
#include <set>
#include <stdexcept>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>
#include "../../include/hyslam_amd.h"
#include "HipORBExtractor.h"
#include "HipAssociationReplay.h"

namespace HYSLAM {

class HipMatcherCore {
public:
    // `handle`: any hs_orb on the device the matcher should run on (e.g. HipORBExtractor::handle()); it only lends its stream and scratch.
    // A handle is thread-compatible: give each thread that matches concurrently (Tracking, Mapping jobs) its own.
    // The four settings are FeatureMatcher's protected members of the same names (FeatureMatcher.h:158-161).
    HipMatcherCore(float nnratio, bool checkOri, float th_low, float th_high, hs_orb* handle)
        : mfNNratio(nnratio), mbCheckOrientation(checkOri), TH_LOW(th_low), TH_HIGH(th_high), h(handle) {}
    HipMatcherCore(FeatureMatcherSettings s, hs_orb* handle) : HipMatcherCore(s.nnratio, s.checkOri, s.TH_LOW, s.TH_HIGH, handle) {}

    // SearchByProjection(Frame&, vector<MapPoint*>&, th) — TrackLocalMap (FeatureMatcher.cc:123-143)
    int SearchByProjection(Frame& F, const std::vector<MapPoint*>& vpMapPoints, const float th = 3) {
        hs_proj_params pp = base_params(th, TH_HIGH, mfNNratio);
        pp.use_distance = 1; pp.use_stereo = 1; pp.check_rotation = 0;
        return project_and_associate(F, vpMapPoints, nullptr, pp);
    }

    // SearchByProjection(CurrentFrame, LastFrame, th, bMono) — TrackMotionModel (FeatureMatcher.cc:145-176)
    int SearchByProjection(Frame& CurrentFrame, const Frame& LastFrame, const float th, const bool /*bMono*/) {
        hs_proj_params pp = base_params(th, TH_HIGH, mfNNratio);
        pp.use_distance = 0; pp.use_stereo = 1; pp.check_rotation = 1;
        return project_and_associate(CurrentFrame, LastFrame.replicatemvpMapPoints(), &LastFrame, pp);
    }

    // SearchByProjection(CurrentFrame, pKF, sAlreadyFound, th, ORBdist) — relocalisation (FeatureMatcher.cc:180-212).  The reference's
    // RotationConsistencyCriterion is a no-op here (no previous frame is set, MatchCriteria.cpp:368-371).
    int SearchByProjection(Frame& CurrentFrame, KeyFrame* pKF, const std::set<MapPoint*>& sAlreadyFound, const float th, const int ORBdist) {
        std::vector<MapPoint*> lms;
        for (const auto& kv : pKF->getLandMarkMatches()) if (kv.second && !sAlreadyFound.count(kv.second)) lms.push_back(kv.second);
        hs_proj_params pp = base_params(th, (float)ORBdist, 1.0f);
        pp.use_distance = 1; pp.use_stereo = 0; pp.check_rotation = 0;
        return project_and_associate(CurrentFrame, lms, nullptr, pp);
    }

    // SearchByBoW(pKF, F, matches) — TrackReferenceKeyFrame / relocalisation (FeatureMatcher.cc:216-278)
    int SearchByBoW(KeyFrame* pKF, Frame& F, std::map<size_t, MapPoint*>& matches) {
        const FeatureViews& v1 = pKF->getViews(); const FeatureViews& v2 = F.getViews();
        Views a = gather_views(v1), b = gather_views(v2);
        Csr f1 = gather_featvec(pKF->mFeatVec), f2 = gather_featvec(F.mFeatVec);
        // PreviouslyMatchedIndexCriterion(true): keep key-frame indices that HAVE a (good) landmark (MatchCriteria.cpp:555-574)
        std::vector<uint8_t> keep1(a.kps.size(), 0);
        for (const auto& kv : pKF->getLandMarkMatches())
            if (kv.second && !kv.second->isBad() && kv.first >= 0 && kv.first < (int)keep1.size()) keep1[kv.first] = 1;
        std::vector<int32_t> m12(std::max<size_t>(a.kps.size(), 1), -1); int32_t n = 0;
        check(hs_search_by_bow(h, a.kps.data(), a.desc.data(), (int)a.kps.size(), f1.id.data(), f1.ptr.data(), f1.idx.data(), (int)f1.ptr.size() - 1,
                               b.kps.data(), b.desc.data(), (int)b.kps.size(), f2.id.data(), f2.ptr.data(), f2.idx.data(), (int)f2.ptr.size() - 1,
                               keep1.data(), TH_LOW, mfNNratio, 1, m12.data(), &n), "SearchByBoW");
        for (size_t i = 0; i < a.kps.size(); i++)          // ascending key-frame index = the order of the reference's std::map (:267-272)
            if (m12[i] >= 0) matches[(size_t)m12[i]] = pKF->hasAssociation((int)i);
        return n;
    }

    // SearchForInitialization(F1, F2, vbPrevMatched, vnMatches12, windowSize) — MonoInitializer (FeatureMatcher.cc:404-462)
    int SearchForInitialization(Frame& F1, Frame& F2, std::vector<cv::Point2f>& vbPrevMatched, std::vector<int>& vnMatches12, int windowSize = 10) {
        Views a = gather_views(F1.getViews());
        FrameArrays fb; hs_frame_view V = gather_frame(F2, fb);
        std::vector<float> prev(2 * std::max<size_t>(a.kps.size(), 1));
        for (size_t i = 0; i < a.kps.size(); i++) { prev[2 * i] = vbPrevMatched[i].x; prev[2 * i + 1] = vbPrevMatched[i].y; }
        std::vector<int32_t> m(std::max<size_t>(a.kps.size(), 1), -1); int32_t n = 0;
        check(hs_search_for_initialization(h, a.kps.data(), a.desc.data(), (int)a.kps.size(), &V, prev.data(), windowSize, TH_LOW, mfNNratio, m.data(), &n),
              "SearchForInitialization");
        vnMatches12.assign(a.kps.size(), -1);
        for (size_t i = 0; i < a.kps.size(); i++) { vnMatches12[i] = m[i]; vbPrevMatched[i] = cv::Point2f(prev[2 * i], prev[2 * i + 1]); }
        return n;
    }

    // Fuse(pKF, vpMapPoints, fuse_matches, th, reprojection_err) — LandMarkFuser (FeatureMatcher.cc:464-521)
    int Fuse(KeyFrame* pKF, const std::vector<MapPoint*>& vpMapPoints, std::map<std::size_t, MapPoint*>& fuse_matches, const float th = 3.0,
             const float reprojection_err = 5.99) {
        hs_proj_params pp = base_params(th, TH_LOW, 1.0f);
        pp.use_distance = 1; pp.use_stereo = 0; pp.check_rotation = 0; pp.use_prev_matched = 0;
        pp.use_viewing_angle = 1; pp.max_view_angle = 1.047f; pp.use_reprojection = 1; pp.reproj_threshold = reprojection_err; pp.first_wins = 1;
        FrameArrays fa; hs_frame_view V = gather_frame(*pKF, fa);
        // the reference walks vpMapPoints in VECTOR order and std::map::insert keeps the first landmark per keypoint (:515): array order here
        std::vector<MapPoint*> lms = vpMapPoints;
        std::vector<hs_landmark> L = gather_landmarks(lms, *pKF, nullptr);
        for (size_t i = 0; i < lms.size(); i++)
            if (lms[i] && (lms[i]->isBad() || lms[i]->IsInKeyFrame(pKF) || lms[i]->Protected())) L[i].skip = 1;      // pre-screen (:480-485)
        std::vector<int32_t> midx(std::max<size_t>(lms.size(), 1), -1); std::vector<float> mdist(midx.size(), -1.f); int32_t n = 0;
        check(hs_search_by_projection(h, &V, L.data(), (int)lms.size(), &pp, midx.data(), mdist.data(), &n), "Fuse");
        for (size_t i = 0; i < lms.size(); i++) if (midx[i] >= 0) fuse_matches.insert(std::make_pair((size_t)midx[i], lms[i]));
        return (int)fuse_matches.size();
    }

    // SearchByBoW(pKF1, pKF2, vpMatches12) — the legacy key-frame matcher (FeatureMatcher.cc:938-1077; no call site in hySLAM): a KF2 view is matched at
    // most once (vbMatched2), the orientation histogram takes angle1 - angle2; views take part when they have a landmark that is not bad.
    int SearchByBoW(KeyFrame* pKF1, KeyFrame* pKF2, std::vector<MapPoint*>& vpMatches12) {
        const FeatureViews& v1 = pKF1->getViews(); const FeatureViews& v2 = pKF2->getViews();
        Views a = gather_views(v1), b = gather_views(v2);
        Csr f1 = gather_featvec(pKF1->mFeatVec), f2 = gather_featvec(pKF2->mFeatVec);
        const std::vector<MapPoint*> mp1 = pKF1->GetMapPointMatches(), mp2 = pKF2->GetMapPointMatches();
        std::vector<uint8_t> keep1(std::max<size_t>(mp1.size(), 1), 0), keep2(std::max<size_t>(mp2.size(), 1), 0);
        for (size_t i = 0; i < mp1.size(); i++) keep1[i] = mp1[i] && !mp1[i]->isBad();
        for (size_t i = 0; i < mp2.size(); i++) keep2[i] = mp2[i] && !mp2[i]->isBad();
        std::vector<int32_t> m12(std::max<size_t>(a.kps.size(), 1), -1); int32_t n = 0;
        check(hs_search_by_bow_legacy(h, a.kps.data(), a.desc.data(), (int)a.kps.size(), f1.id.data(), f1.ptr.data(), f1.idx.data(), (int)f1.ptr.size() - 1,
                                      b.kps.data(), b.desc.data(), (int)b.kps.size(), f2.id.data(), f2.ptr.data(), f2.idx.data(), (int)f2.ptr.size() - 1,
                                      keep1.data(), keep2.data(), TH_LOW, mfNNratio, mbCheckOrientation ? 1 : 0, m12.data(), &n), "SearchByBoW(KF, KF)");
        vpMatches12 = std::vector<MapPoint*>(mp1.size(), static_cast<MapPoint*>(NULL));          // :957
        for (size_t i = 0; i < mp1.size() && i < a.kps.size(); i++) if (m12[i] >= 0) vpMatches12[i] = mp2[m12[i]];
        return n;
    }

    // SearchByBoW2(pKF1, pKF2, vpMatches12) — LoopClosing.cc:275 (FeatureMatcher.cc:346-371): _SearchByBoW_ with PreviouslyMatchedIndexCriterion(true) on
    // BOTH key frames (:306-309), BestMatchBoWCriterion(TH_LOW, mfNNratio), RotationConsistencyBoW; vpMatches12[i] = KF2's landmark at the matched view.
    int SearchByBoW2(KeyFrame* pKF1, KeyFrame* pKF2, std::vector<MapPoint*>& vpMatches12) {
        std::vector<int32_t> m12; const int n = bow_between_keyframes(pKF1, pKF2, true, false, nullptr, mfNNratio, m12, "SearchByBoW2");
        const size_t N1 = pKF1->GetMapPointMatches().size();
        vpMatches12 = std::vector<MapPoint*>(N1, static_cast<MapPoint*>(NULL));
        for (size_t i = 0; i < N1 && i < m12.size(); i++) if (m12[i] >= 0) vpMatches12[i] = pKF2->hasAssociation((int)m12[i]);
        return n;
    }

    // SearchForTriangulation(pKF1, pKF2, F12, vMatchedPairs, bOnlyStereo) — LandMarkTriangulator.cpp:81 (FeatureMatcher.cc:373-402):
    // PreviouslyMatchedIndexCriterion(false) (+ StereoIndexCriterion when bOnlyStereo) on both key frames, EpipolarConsistencyBoWCriterion(F12)
    // — the epipole (ex, ey) the reference computes is stored by the criterion and never read (MatchCriteria.cpp:637-676) —
    // BestMatchBoWCriterion(TH_LOW, 1.0), RotationConsistencyBoW.  Pairs are appended in ascending idx1 order (std::map iteration, :396-398).
    int SearchForTriangulation(KeyFrame* pKF1, KeyFrame* pKF2, cv::Mat F12, std::vector<std::pair<size_t, size_t>>& vMatchedPairs, const bool bOnlyStereo) {
        float F[9];
        for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) F[3 * r + c] = F12.at<float>(r, c);
        std::vector<int32_t> m12; const int n = bow_between_keyframes(pKF1, pKF2, false, bOnlyStereo, F, 1.0f, m12, "SearchForTriangulation");
        for (size_t i = 0; i < m12.size(); i++) if (m12[i] >= 0) vMatchedPairs.push_back(std::make_pair(i, (size_t)m12[i]));
        return n;
    }

    // SearchByProjection(pKF, Scw, vpPoints, vpMatched, th) — LoopClosing.cc:389 (FeatureMatcher.cc:628-737).  Landmarks stay in vpPoints order (the
    // search is sequential: a keypoint taken by an earlier landmark is invisible to later ones, :713,731).
    int SearchByProjection(KeyFrame* pKF, cv::Mat Scw, const std::vector<MapPoint*>& vpPoints, std::vector<MapPoint*>& vpMatched, int th) {
        FrameArrays fa; hs_frame_view V = gather_frame(*pKF, fa);
        float S[16];
        for (int r = 0; r < 4; r++) for (int c = 0; c < 4; c++) S[4 * r + c] = Scw.at<float>(r, c);
        std::set<MapPoint*> spAlreadyFound(vpMatched.begin(), vpMatched.end());                               // :649-650
        spAlreadyFound.erase(static_cast<MapPoint*>(NULL));
        std::vector<hs_landmark> L = gather_landmarks(vpPoints, *pKF, nullptr);
        for (size_t i = 0; i < vpPoints.size(); i++)
            if (vpPoints[i] && (vpPoints[i]->isBad() || spAlreadyFound.count(vpPoints[i]))) L[i].skip = 1;   // :660-661
        std::vector<uint8_t> taken(std::max<size_t>(V.n, 1), 0);
        for (int i = 0; i < V.n && i < (int)vpMatched.size(); i++) taken[i] = vpMatched[i] != nullptr;
        std::vector<int32_t> midx(std::max<size_t>(vpPoints.size(), 1), -1); int32_t n = 0;
        check(hs_search_by_projection_sim3(h, &V, S, L.data(), (int)vpPoints.size(), th, TH_LOW, taken.data(), midx.data(), &n), "SearchByProjection(Scw)");
        for (size_t i = 0; i < vpPoints.size(); i++) if (midx[i] >= 0) vpMatched[midx[i]] = vpPoints[i];     // :731
        return n;
    }

    // Fuse(pKF, Scw, vpPoints, th, vpReplacePoint) — LoopClosing.cc:632.  The reference's body is commented out in full (FeatureMatcher.cc:523-624):
    // it computes nothing and falls off the end of an int function.  Nothing to run on the GPU; the override leaves vpReplacePoint untouched and returns 0.
    int Fuse(KeyFrame* /*pKF*/, cv::Mat /*Scw*/, const std::vector<MapPoint*>& /*vpPoints*/, float /*th*/, std::vector<MapPoint*>& /*vpReplacePoint*/) { return 0; }

    // SearchBySim3(pKF1, pKF2, vpMatches12, s12, R12, t12, th) — LoopClosing.cc:333 (FeatureMatcher.cc:739-934): both projection directions, then the
    // agreement check; vpMatches12[i1] = KF2's landmark at the agreed view.
    int SearchBySim3(KeyFrame* pKF1, KeyFrame* pKF2, std::vector<MapPoint*>& vpMatches12, const float& s12, const cv::Mat& R12, const cv::Mat& t12, const float th) {
        FrameArrays fa1, fa2; hs_frame_view V1 = gather_frame(*pKF1, fa1), V2 = gather_frame(*pKF2, fa2);
        const std::vector<MapPoint*> mp1 = pKF1->GetMapPointMatches(), mp2 = pKF2->GetMapPointMatches();
        const int N1 = (int)mp1.size(), N2 = (int)mp2.size();
        std::vector<bool> done1(N1, false), done2(N2, false);                                                 // vbAlreadyMatched1 / 2 (:768-781)
        for (int i = 0; i < N1 && i < (int)vpMatches12.size(); i++)
            if (MapPoint* pMP = vpMatches12[i]) { done1[i] = true; const int idx2 = pMP->GetIndexInKeyFrame(pKF2); if (idx2 >= 0 && idx2 < N2) done2[idx2] = true; }
        // the landmark of each keypoint, sized in the OTHER key frame (landMarkSizePixels of pKF2 / pKF1, :823,883)
        std::vector<hs_landmark> L1 = gather_landmarks(mp1, *pKF2, nullptr), L2 = gather_landmarks(mp2, *pKF1, nullptr);
        for (int i = 0; i < N1; i++) if (mp1[i] && (done1[i] || mp1[i]->isBad())) L1[i].skip = 1;
        for (int i = 0; i < N2; i++) if (mp2[i] && (done2[i] || mp2[i]->isBad())) L2[i].skip = 1;
        float R[9], t[3];
        for (int r = 0; r < 3; r++) { for (int c = 0; c < 3; c++) R[3 * r + c] = R12.at<float>(r, c); t[r] = t12.at<float>(r); }
        std::vector<int32_t> m12(std::max(N1, 1), -1); int32_t n = 0;
        check(hs_search_by_sim3(h, &V1, L1.data(), &V2, L2.data(), s12, R, t, th, TH_HIGH, m12.data(), &n), "SearchBySim3");
        if ((int)vpMatches12.size() < N1) vpMatches12.resize(N1, static_cast<MapPoint*>(NULL));
        for (int i = 0; i < N1; i++) if (m12[i] >= 0) vpMatches12[i] = mp2[m12[i]];                            // :925
        return n;
    }

    hs_orb* handle() const { return h; }
    HipCallTiming timing;             // of the last SearchByProjection(Frame...) / Fuse / key-frame BoW call (HipORBExtractor.h)
    size_t replay_calls = 0, replay_full = 0; int replay_rule = 0;      // of the last SearchByProjection(Frame...): associateLandMark calls made / of the reference's loop
    bool frame_on_device = false;     // the last SearchByProjection(Frame...) took the frame's keypoints / descriptors from the device's frame cache

private:
    struct Views { std::vector<hs_keypoint> kps; std::vector<uint8_t> desc; std::vector<float> uR; };
    struct FrameArrays { Views v; std::vector<int32_t> obs; hs_frame_token token = 0; };
    struct Csr { std::vector<int32_t> id, ptr, idx; };

    void check(int st, const char* what) const {
        if (st != HS_OK) throw std::runtime_error(std::string("HipMatcherCore::") + what + ": " + hs_status_string(st) + ": " + hs_orb_last_error(h));
    }
    hs_proj_params base_params(float th, float score_threshold, float ratio) const {
        hs_proj_params pp; std::memset(&pp, 0, sizeof(pp));
        pp.th = th; pp.score_threshold = score_threshold; pp.second_best_ratio = ratio; pp.frac_smaller = 0.5f; pp.frac_larger = 1.5f;
        pp.use_prev_matched = 1; pp.max_view_angle = 1.047f; pp.reproj_threshold = 5.99f; pp.sigma_ref = 1.0f;
        pp.dist_is_invariance_range = 1;       // MapPoint only exposes GetMin/MaxDistanceInvariance() (= 0.8f*min, 1.2f*max), MapPoint.cc:139-149
        return pp;
    }
    static Views gather_views(const FeatureViews& views, bool with_descriptors = true) {
        Views o; const int n = views.numViews();
        o.kps.resize(std::max(n, 1)); o.uR.resize(std::max(n, 1), -1.f);
        for (int i = 0; i < n; i++) {
            const cv::KeyPoint k = views.keypt(i);
            o.kps[i] = hs_keypoint{ k.pt.x, k.pt.y, k.size, k.angle, k.response, k.octave };
            o.uR[i] = views.uR(i);
        }
        o.kps.resize(n); o.uR.resize(n);
        if (with_descriptors) gather_descriptors(views, o);
        return o;
    }
    // the expensive half of a frame gather: FeatureDescriptor::rawDescriptor() is the only accessor and clones a cv::Mat per keypoint
    static void gather_descriptors(const FeatureViews& views, Views& o) {
        const int n = views.numViews();
        o.desc.resize((size_t)std::max(n, 1) * HS_DESC_BYTES);
        for (int i = 0; i < n; i++) {
            const cv::Mat row = views.descriptor(i).rawDescriptor();
            std::memcpy(o.desc.data() + (size_t)i * HS_DESC_BYTES, row.ptr(0), HS_DESC_BYTES);
        }
    }
    static Csr gather_featvec(const DBoW2::FeatureVector& fv) {     // std::map: node ids ascending, indices in insertion (ascending) order
        Csr c; c.ptr.push_back(0);
        for (const auto& kv : fv) { c.id.push_back((int32_t)kv.first); for (unsigned i : kv.second) c.idx.push_back((int32_t)i); c.ptr.push_back((int32_t)c.idx.size()); }
        if (c.idx.empty()) c.idx.push_back(0);
        if (c.id.empty()) c.id.push_back(0);
        return c;
    }
    // Frame / KeyFrame -> hs_frame_view (Frame.cc:45-72,137-180; Camera.cpp:116-153).  `T` needs mTcw-like pose access: see pose_of().
    static cv::Mat pose_of(Frame& F) { return F.mTcw; }
    static cv::Mat pose_of(KeyFrame& K) { return K.GetPose(); }
    // device >= 0: look the frame up in the device's frame cache first (hs_frame_find: the extractor adaptors publish what they extract, include/hyslam_amd.h
    // "device-resident frames").  Found -> a.token is set, the descriptors are NOT gathered (V.desc = NULL) and the caller uses a *_frame entry point.
    template <class T> static hs_frame_view gather_frame(T& F, FrameArrays& a, int device = -1) {
        hs_frame_view V; std::memset(&V, 0, sizeof(V));
        const cv::Mat Tcw = pose_of(F);
        if (!Tcw.empty()) {
            for (int r = 0; r < 3; r++) { for (int c = 0; c < 3; c++) V.Rcw[3 * r + c] = Tcw.at<float>(r, c); V.tcw[r] = Tcw.at<float>(r, 3); }
            const cv::Mat Ow = F.GetCameraCenter();
            for (int r = 0; r < 3; r++) V.Ow[r] = Ow.at<float>(r);
        }
        const Camera& cam = F.getCamera();
        V.fx = cam.fx(); V.fy = cam.fy(); V.cx = cam.cx(); V.cy = cam.cy(); V.mbf = cam.mbf; V.sensor = cam.sensor;
        V.min_x = F.mnMinX; V.max_x = F.mnMaxX; V.min_y = F.mnMinY; V.max_y = F.mnMaxY;
        const FeatureViews& views = F.getViews();
        V.size_ref = views.orbParams().size_ref;
        a.v = gather_views(views, false);
        V.n = (int)a.v.kps.size();
        a.token = 0;
        if (device >= 0 && V.n > 0 && hs_frame_find(device, a.v.kps.data(), V.n, &a.token) != HS_OK) a.token = 0;
        if (!a.token) gather_descriptors(views, a.v);
        a.obs.assign(std::max(V.n, 1), -1);
        for (const auto& kv : F.getLandMarkMatches())       // PreviouslyMatchedCriterionCore: Observations() of the keypoint's landmark (MatchCriteria.cpp:124-144)
            if (kv.second && kv.first >= 0 && kv.first < V.n) a.obs[kv.first] = kv.second->Observations();
        V.kps = a.v.kps.data(); V.desc = a.token ? nullptr : a.v.desc.data(); V.uR = a.v.uR.data(); V.kp_lm_obs = a.obs.data();
        return V;
    }
    // MapPoints -> hs_landmark records.  `F` supplies hasAssociation(lm) (landMarkSizePixels, Frame.cc:296-300) through ONE reverse map instead
    // of the reference's linear scan per landmark; `prev` (may be null) supplies the previous frame's keypoint angle (rotation check).
    template <class T> static std::vector<hs_landmark> gather_landmarks(const std::vector<MapPoint*>& lms, T& F, const Frame* prev, bool need_normals = true) {
        std::unordered_map<MapPoint*, int> assoc, assoc_prev;
        for (const auto& kv : F.getLandMarkMatches()) assoc.insert({ kv.second, kv.first });                 // ascending view index: the first view wins, like the scan
        if (prev) for (const auto& kv : const_cast<Frame*>(prev)->getLandMarkMatches()) assoc_prev.insert({ kv.second, kv.first });
        std::vector<hs_landmark> out(std::max<size_t>(lms.size(), 1));
        std::memset(out.data(), 0, out.size() * sizeof(hs_landmark));
        // geometry of landmarks [a, b): position (+ normal: only Fuse's viewing-angle criterion reads it), size, distance range, associations
        // The landmarks are separate heap objects visited in address order of their POINTERS' set, i.e. in no order the hardware prefetcher can follow:
        // the first cache lines of the object a few entries ahead are requested by hand (a MapPoint's position, distances and descriptor header sit in
        // its first few hundred bytes; nothing is assumed about its layout beyond that).
        constexpr size_t PF_AHEAD = 8;
        auto prefetch_object = [](const void* p) {
            if (!p) return;
            const char* c = static_cast<const char*>(p);
            __builtin_prefetch(c); __builtin_prefetch(c + 64); __builtin_prefetch(c + 128); __builtin_prefetch(c + 192);
        };
        auto fill_geometry = [&](size_t a, size_t b) {
            for (size_t i = a; i < b; i++) {
                if (i + PF_AHEAD < b) prefetch_object(lms[i + PF_AHEAD]);
                hs_landmark& L = out[i];
                MapPoint* lm = lms[i];
                L.assoc_kp = -1;
                if (!lm) { L.skip = 1; continue; }
                const cv::Mat P = lm->GetWorldPos();
                for (int k = 0; k < 3; k++) L.pos[k] = P.at<float>(k);
                if (need_normals) { const cv::Mat Nn = lm->GetNormal(); for (int k = 0; k < 3; k++) L.normal[k] = Nn.at<float>(k); }
                L.size = lm->getSize();
                L.min_dist = lm->GetMinDistanceInvariance(); L.max_dist = lm->GetMaxDistanceInvariance();         // with dist_is_invariance_range = 1
                auto it = assoc.find(lm);
                if (it != assoc.end()) L.assoc_kp = it->second;
                if (prev) { auto ip = assoc_prev.find(lm); if (ip != assoc_prev.end()) L.prev_angle = prev->getViews().keypt(ip->second).angle; }
            }
        };
        auto fill_descriptors = [&](size_t a, size_t b) {      // walks DOWN: the geometry helpers walk up through the same records (one cache line each) — meet once, not at every record
            for (size_t i = b; i-- > a;) {
                if (i >= a + PF_AHEAD) prefetch_object(lms[i - PF_AHEAD]);
                if (!lms[i]) continue;
                const cv::Mat row = lms[i]->GetDescriptor().rawDescriptor();
                std::memcpy(out[i].desc, row.ptr(0), HS_DESC_BYTES);
            }
        };
        // Every accessor MapPoint offers clones (GetWorldPos / GetNormal: a cv::Mat each; GetDescriptor(): a FeatureDescriptor copy, then
        // rawDescriptor(): another clone), each under the MapPoint's own mutex: ~100 ns per landmark, 5.2 ms for TrackLocalMap's 50 000 on one
        // thread; the normals (a third of it) are only fetched for the one search that reads them.  Splitting the LANDMARKS over threads makes it
        // slower, not faster (5.2 -> 5.7 ms with 8 threads on the GPU box): every FeatureDescriptor copy increments and decrements the reference count
        // of the ONE DescriptorDistance object all descriptors share (FeatureDescriptor.h:31-33), and that cache line then bounces between the cores.
        // Splitting by FIELD does not: the calling thread takes all descriptors (the shared count stays in its cache), two helper threads take the
        // geometry, whose clones share nothing (GPU box, 50 000 landmarks, 0 / 1 / 2 / 4 helpers: 3.3 / 3.8 / 2.2 / 2.9 ms).  Disjoint fields of the
        // output records; the two association maps are only read.
        const size_t n = lms.size();
        static const unsigned max_helpers = [] { const char* e = std::getenv("HYSLAM_AMD_GATHER_THREADS"); const int v = e ? std::atoi(e) : -1; return v >= 0 ? (unsigned)std::min(v, 2) : 2u; }();
        const unsigned helpers_n = n >= 4096 ? std::min<unsigned>(std::max(1u, std::thread::hardware_concurrency()) - 1u, max_helpers) : 0u;
        if (helpers_n == 0) { fill_geometry(0, n); fill_descriptors(0, n); return out; }
        hip_detail::Worker* const helpers = hip_detail::thread_workers();      // persistent per calling thread (HipORBExtractor.h)
        const size_t chunk = (n + helpers_n - 1) / helpers_n;
        // a helper that has started references this frame's locals: whatever throws from here on (thread creation inside run() included) first waits for
        // every helper that was started
        unsigned started = 0;
        try {
            for (; started < helpers_n; started++) { const size_t a = std::min(n, started * chunk), b = std::min(n, (started + 1) * chunk); helpers[started].run([&fill_geometry, a, b] { fill_geometry(a, b); }); }
            fill_descriptors(0, n);
        } catch (...) { for (unsigned w = 0; w < started; w++) { try { helpers[w].wait(); } catch (...) {} } throw; }
        for (unsigned w = 0; w < helpers_n; w++) helpers[w].wait();
        return out;
    }
    // _SearchByBoW_ between two key frames (FeatureMatcher.cc:281-345): the index criteria apply to BOTH sides (:306-309).
    // keep_matched = PreviouslyMatchedIndexCriterion's flag (MatchCriteria.cpp:554-574); only_stereo adds StereoIndexCriterion (:577-595: a mono key
    // frame passes everything, otherwise uR >= 0).  m12[i] = KF2 view matched to KF1 view i, or -1.
    int bow_between_keyframes(KeyFrame* pKF1, KeyFrame* pKF2, bool keep_matched, bool only_stereo, const float* F12, float ratio, std::vector<int32_t>& m12, const char* what) {
        const auto t0 = std::chrono::steady_clock::now();
        const FeatureViews& v1 = pKF1->getViews(); const FeatureViews& v2 = pKF2->getViews();
        Views a = gather_views(v1), b = gather_views(v2);
        Csr f1 = gather_featvec(pKF1->mFeatVec), f2 = gather_featvec(pKF2->mFeatVec);
        auto index_mask = [&](KeyFrame* pKF, const Views& v) {
            std::vector<uint8_t> has(std::max<size_t>(v.kps.size(), 1), 0), keep(has.size(), 0);
            for (const auto& kv : pKF->getLandMarkMatches())
                if (kv.second && !kv.second->isBad() && kv.first >= 0 && kv.first < (int)v.kps.size()) has[kv.first] = 1;
            const bool stereo_gate = only_stereo && pKF->getCamera().sensor != 0;
            for (size_t i = 0; i < v.kps.size(); i++) keep[i] = (has[i] != 0) == keep_matched && (!stereo_gate || v.uR[i] >= 0);
            return keep;
        };
        std::vector<uint8_t> keep1 = index_mask(pKF1, a), keep2 = index_mask(pKF2, b);
        const FeatureExtractorSettings orb2 = v2.orbParams();
        m12.assign(std::max<size_t>(a.kps.size(), 1), -1); int32_t n = 0;
        timing.gather_ms = hip_detail::ms_since(t0);
        const auto t1 = std::chrono::steady_clock::now();
        check(hs_search_by_bow_ex(h, a.kps.data(), a.desc.data(), (int)a.kps.size(), f1.id.data(), f1.ptr.data(), f1.idx.data(), (int)f1.ptr.size() - 1,
                                  b.kps.data(), b.desc.data(), (int)b.kps.size(), f2.id.data(), f2.ptr.data(), f2.idx.data(), (int)f2.ptr.size() - 1,
                                  keep1.data(), keep2.data(), F12, orb2.size_ref, orb2.sigma_ref, TH_LOW, ratio, 1, m12.data(), &n), what);
        timing.abi_ms = hip_detail::ms_since(t1); timing.scatter_ms = 0;
        m12.resize(a.kps.size());
        return n;
    }
    int project_and_associate(Frame& F, const std::vector<MapPoint*>& landmarks, const Frame* prev, const hs_proj_params& pp) {
        // address order == iteration order of the reference's std::map<MapPoint*, SingleMatchData> (FeatureMatcher.cc:64); duplicates collapse like map keys
        const auto t0 = std::chrono::steady_clock::now();
        std::vector<MapPoint*> lms;
        // TrackLocalMap hands over the contents of a std::set<MapPoint*> (TrackLocalMap.cpp:73): already ascending and unique — one pass instead of a sort
        bool ascending = true;
        for (size_t i = 0; i < landmarks.size() && ascending; i++) ascending = landmarks[i] != nullptr && (i == 0 || landmarks[i - 1] < landmarks[i]);
        if (ascending) lms = landmarks;
        else {
            lms.reserve(landmarks.size());
            for (MapPoint* p : landmarks) if (p) lms.push_back(p);
            std::sort(lms.begin(), lms.end());
            lms.erase(std::unique(lms.begin(), lms.end()), lms.end());
        }
        if (lms.empty()) return 0;
        FrameArrays fa; hs_frame_view V = gather_frame(F, fa, hs_orb_get_device(h));
        std::vector<hs_landmark> L = gather_landmarks(lms, F, prev, pp.use_viewing_angle != 0);
        std::vector<int32_t> midx(lms.size(), -1); std::vector<float> mdist(lms.size(), -1.f); int32_t n = 0;
        timing.gather_ms = hip_detail::ms_since(t0);
        const auto t1 = std::chrono::steady_clock::now();
        frame_on_device = false;
        if (fa.token) {      // the frame's keypoints and descriptors are still on the device (published by the extractor adaptor): nothing of them is uploaded
            const int st = hs_search_by_projection_frame(h, fa.token, &V, L.data(), (int)lms.size(), &pp, midx.data(), mdist.data(), &n);
            if (st == HS_OK) frame_on_device = true;
            else if (st != HS_ERR_INVALID) check(st, "SearchByProjection(frame on device)");
            else { gather_descriptors(F.getViews(), fa.v); V.desc = fa.v.desc.data(); }      // the slot was reused meanwhile: the host path
        }
        if (!frame_on_device) check(hs_search_by_projection(h, &V, L.data(), (int)lms.size(), &pp, midx.data(), mdist.data(), &n), "SearchByProjection");
        timing.abi_ms = hip_detail::ms_since(t1);
        const auto t2 = std::chrono::steady_clock::now();
        // replay (FeatureMatcher.cc:113-118): the calls that decide the frame's final LandMarkMatches, in address order (HipAssociationReplay.h)
        const hip_detail::ReplayPlan plan = hip_detail::replay_associations(F, lms, midx);
        replay_calls = plan.ops.size(); replay_full = plan.full_ops; replay_rule = plan.rule;
        timing.scatter_ms = hip_detail::ms_since(t2);
        return n;
    }

    float mfNNratio; bool mbCheckOrientation; float TH_LOW, TH_HIGH;
    hs_orb* h;
};

#ifndef HYSLAM_AMD_UNPATCHED_MATCHER
// integration (a): FeatureMatcher with virtual search entry points (the patch of INTEGRATION.md §3); every override forwards to the core
class HipFeatureMatcher : public FeatureMatcher {
public:
    HipFeatureMatcher(FeatureMatcherSettings settings, hs_orb* handle) : FeatureMatcher(settings), core(settings, handle), timing(core.timing) {}
    int SearchByProjection(Frame& F, const std::vector<MapPoint*>& vpMapPoints, const float th = 3) override { return core.SearchByProjection(F, vpMapPoints, th); }
    int SearchByProjection(Frame& CurrentFrame, const Frame& LastFrame, const float th, const bool bMono) override { return core.SearchByProjection(CurrentFrame, LastFrame, th, bMono); }
    int SearchByProjection(Frame& CurrentFrame, KeyFrame* pKF, const std::set<MapPoint*>& sAlreadyFound, const float th, const int ORBdist) override { return core.SearchByProjection(CurrentFrame, pKF, sAlreadyFound, th, ORBdist); }
    int SearchByProjection(KeyFrame* pKF, cv::Mat Scw, const std::vector<MapPoint*>& vpPoints, std::vector<MapPoint*>& vpMatched, int th) override { return core.SearchByProjection(pKF, Scw, vpPoints, vpMatched, th); }
    int SearchByBoW(KeyFrame* pKF, Frame& F, std::map<size_t, MapPoint*>& matches) override { return core.SearchByBoW(pKF, F, matches); }
    int SearchByBoW(KeyFrame* pKF1, KeyFrame* pKF2, std::vector<MapPoint*>& vpMatches12) override { return core.SearchByBoW(pKF1, pKF2, vpMatches12); }
    int SearchByBoW2(KeyFrame* pKF1, KeyFrame* pKF2, std::vector<MapPoint*>& vpMatches12) override { return core.SearchByBoW2(pKF1, pKF2, vpMatches12); }
    int SearchForTriangulation(KeyFrame* pKF1, KeyFrame* pKF2, cv::Mat F12, std::vector<std::pair<size_t, size_t>>& vMatchedPairs, const bool bOnlyStereo) override { return core.SearchForTriangulation(pKF1, pKF2, F12, vMatchedPairs, bOnlyStereo); }
    int SearchForInitialization(Frame& F1, Frame& F2, std::vector<cv::Point2f>& vbPrevMatched, std::vector<int>& vnMatches12, int windowSize = 10) override { return core.SearchForInitialization(F1, F2, vbPrevMatched, vnMatches12, windowSize); }
    int Fuse(KeyFrame* pKF, const std::vector<MapPoint*>& vpMapPoints, std::map<std::size_t, MapPoint*>& fuse_matches, const float th = 3.0, const float reprojection_err = 5.99) override { return core.Fuse(pKF, vpMapPoints, fuse_matches, th, reprojection_err); }
    int Fuse(KeyFrame* pKF, cv::Mat Scw, const std::vector<MapPoint*>& vpPoints, float th, std::vector<MapPoint*>& vpReplacePoint) override { return core.Fuse(pKF, Scw, vpPoints, th, vpReplacePoint); }
    int SearchBySim3(KeyFrame* pKF1, KeyFrame* pKF2, std::vector<MapPoint*>& vpMatches12, const float& s12, const cv::Mat& R12, const cv::Mat& t12, const float th) override { return core.SearchBySim3(pKF1, pKF2, vpMatches12, s12, R12, t12, th); }
    hs_orb* handle() const { return core.handle(); }
    const HipMatcherCore& matcherCore() const { return core; }      // replay_calls / replay_full / replay_rule of the last projection search
private:
    HipMatcherCore core;
public:
    HipCallTiming& timing;            // of the last SearchByProjection(Frame...) / Fuse / key-frame BoW call
};
#endif

}  // namespace HYSLAM