// HipFeatureMatcher.h — HYSLAM::FeatureMatcher's search entry points over the C ABI (include/hyslam_amd.h).
//
//   HYSLAM::HipFeatureMatcher : HYSLAM::FeatureMatcher     (src/features/FeatureMatcher.h:105-176)
//
// Needs the two-line patch of INTEGRATION.md §3 in hySLAM (`virtual` on FeatureFactory::getFeatureMatcher and on the FeatureMatcher search
// methods): the reference obtains matchers through the NON-virtual FeatureFactory::getFeatureMatcher() (src/features/FeatureFactory.cpp:7-9)
// and calls non-virtual methods, so without the patch a subclass is never reached.  Every override does three things:
//   gather   Frame / KeyFrame / MapPoint fields -> the flat arrays of hs_frame_view / hs_landmark (one pass, no per-landmark map copies);
//            landmarks are passed SORTED BY ADDRESS, which reproduces the iteration order of the reference's std::map<MapPoint*, ...>
//            (FeatureMatcher.cc:64,113-118; deviation D6 of DESIGN.md)
//   search   one C-ABI call (GPU)
//   replay   Frame::associateLandMark(idx, lm, true) for every match, in that same address order (FeatureMatcher.cc:113-118) — a later
//            landmark that picked the same keypoint overwrites the earlier one, exactly as in the reference.
// Entry points not overridden here (Sim3 / loop-closing legacy, SearchByBoW(KF,KF), SearchForTriangulation) stay the reference's CPU code.
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
#include <stdexcept>
#include <string>
#include <unordered_map>
#include <vector>
#include "../../include/hyslam_amd.h"

namespace HYSLAM {

class HipFeatureMatcher : public FeatureMatcher {
public:
    // `handle`: any hs_orb on the device the matcher should run on (e.g. HipORBExtractor::handle()); it only lends its stream and scratch.
    // A handle is thread-compatible: give each thread that matches concurrently (Tracking, Mapping jobs) its own.
    HipFeatureMatcher(FeatureMatcherSettings settings, hs_orb* handle) : FeatureMatcher(settings), h(handle) {}

    // SearchByProjection(Frame&, vector<MapPoint*>&, th) — TrackLocalMap (FeatureMatcher.cc:123-143)
    int SearchByProjection(Frame& F, const std::vector<MapPoint*>& vpMapPoints, const float th = 3) override {
        hs_proj_params pp = base_params(th, TH_HIGH, mfNNratio);
        pp.use_distance = 1; pp.use_stereo = 1; pp.check_rotation = 0;
        return project_and_associate(F, vpMapPoints, nullptr, pp);
    }

    // SearchByProjection(CurrentFrame, LastFrame, th, bMono) — TrackMotionModel (FeatureMatcher.cc:145-176)
    int SearchByProjection(Frame& CurrentFrame, const Frame& LastFrame, const float th, const bool /*bMono*/) override {
        hs_proj_params pp = base_params(th, TH_HIGH, mfNNratio);
        pp.use_distance = 0; pp.use_stereo = 1; pp.check_rotation = 1;
        return project_and_associate(CurrentFrame, LastFrame.replicatemvpMapPoints(), &LastFrame, pp);
    }

    // SearchByProjection(CurrentFrame, pKF, sAlreadyFound, th, ORBdist) — relocalisation (FeatureMatcher.cc:180-212).  The reference's
    // RotationConsistencyCriterion is a no-op here (no previous frame is set, MatchCriteria.cpp:368-371).
    int SearchByProjection(Frame& CurrentFrame, KeyFrame* pKF, const std::set<MapPoint*>& sAlreadyFound, const float th, const int ORBdist) override {
        std::vector<MapPoint*> lms;
        for (const auto& kv : pKF->getLandMarkMatches()) if (kv.second && !sAlreadyFound.count(kv.second)) lms.push_back(kv.second);
        hs_proj_params pp = base_params(th, (float)ORBdist, 1.0f);
        pp.use_distance = 1; pp.use_stereo = 0; pp.check_rotation = 0;
        return project_and_associate(CurrentFrame, lms, nullptr, pp);
    }

    // SearchByBoW(pKF, F, matches) — TrackReferenceKeyFrame / relocalisation (FeatureMatcher.cc:216-278)
    int SearchByBoW(KeyFrame* pKF, Frame& F, std::map<size_t, MapPoint*>& matches) override {
        const FeatureViews& v1 = pKF->getViews(); const FeatureViews& v2 = F.getViews();
        Views a = gather_views(v1), b = gather_views(v2);
        Csr f1 = gather_featvec(pKF->mFeatVec), f2 = gather_featvec(F.mFeatVec);
        // PreviouslyMatchedIndexCriterion(true): keep key-frame indices that HAVE a (good) landmark (MatchCriteria.cpp:555-574)
        std::vector<uint8_t> keep1(a.kps.size(), 0);
        for (const auto& kv : pKF->getLandMarkMatches())
            if (kv.second && !kv.second->isBad() && kv.first >= 0 && kv.first < (int)keep1.size()) keep1[kv.first] = 1;
        std::vector<int32_t> m12(std::max<size_t>(a.kps.size(), 1), -1); int32_t n = 0;
        check(hs_search_by_bow(h, a.kps.data(), a.desc.data(), (int)a.kps.size(), f1.id.data(), f1.ptr.data(), f1.idx.data(), (int)f1.id.size(),
                               b.kps.data(), b.desc.data(), (int)b.kps.size(), f2.id.data(), f2.ptr.data(), f2.idx.data(), (int)f2.id.size(),
                               keep1.data(), TH_LOW, mfNNratio, 1, m12.data(), &n), "SearchByBoW");
        for (size_t i = 0; i < a.kps.size(); i++)          // ascending key-frame index = the order of the reference's std::map (:267-272)
            if (m12[i] >= 0) matches[(size_t)m12[i]] = pKF->hasAssociation((int)i);
        return n;
    }

    // SearchForInitialization(F1, F2, vbPrevMatched, vnMatches12, windowSize) — MonoInitializer (FeatureMatcher.cc:404-462)
    int SearchForInitialization(Frame& F1, Frame& F2, std::vector<cv::Point2f>& vbPrevMatched, std::vector<int>& vnMatches12, int windowSize = 10) override {
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
             const float reprojection_err = 5.99) override {
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

    hs_orb* handle() const { return h; }

private:
    struct Views { std::vector<hs_keypoint> kps; std::vector<uint8_t> desc; std::vector<float> uR; };
    struct FrameArrays { Views v; std::vector<int32_t> obs; };
    struct Csr { std::vector<int32_t> id, ptr, idx; };

    void check(int st, const char* what) const {
        if (st != HS_OK) throw std::runtime_error(std::string("HipFeatureMatcher::") + what + ": " + hs_status_string(st) + ": " + hs_orb_last_error(h));
    }
    hs_proj_params base_params(float th, float score_threshold, float ratio) const {
        hs_proj_params pp; std::memset(&pp, 0, sizeof(pp));
        pp.th = th; pp.score_threshold = score_threshold; pp.second_best_ratio = ratio; pp.frac_smaller = 0.5f; pp.frac_larger = 1.5f;
        pp.use_prev_matched = 1; pp.max_view_angle = 1.047f; pp.reproj_threshold = 5.99f; pp.sigma_ref = 1.0f;
        pp.dist_is_invariance_range = 1;       // MapPoint only exposes GetMin/MaxDistanceInvariance() (= 0.8f*min, 1.2f*max), MapPoint.cc:139-149
        return pp;
    }
    static Views gather_views(const FeatureViews& views) {
        Views o; const int n = views.numViews();
        o.kps.resize(std::max(n, 1)); o.desc.resize((size_t)std::max(n, 1) * HS_DESC_BYTES); o.uR.resize(std::max(n, 1), -1.f);
        for (int i = 0; i < n; i++) {
            const cv::KeyPoint k = views.keypt(i);
            o.kps[i] = hs_keypoint{ k.pt.x, k.pt.y, k.size, k.angle, k.response, k.octave };
            const cv::Mat row = views.descriptor(i).rawDescriptor();
            std::memcpy(o.desc.data() + (size_t)i * HS_DESC_BYTES, row.ptr(0), HS_DESC_BYTES);
            o.uR[i] = views.uR(i);
        }
        o.kps.resize(n); o.uR.resize(n);
        return o;
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
    template <class T> static hs_frame_view gather_frame(T& F, FrameArrays& a) {
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
        a.v = gather_views(views);
        V.n = (int)a.v.kps.size();
        a.obs.assign(std::max(V.n, 1), -1);
        for (const auto& kv : F.getLandMarkMatches())       // PreviouslyMatchedCriterionCore: Observations() of the keypoint's landmark (MatchCriteria.cpp:124-144)
            if (kv.second && kv.first >= 0 && kv.first < V.n) a.obs[kv.first] = kv.second->Observations();
        V.kps = a.v.kps.data(); V.desc = a.v.desc.data(); V.uR = a.v.uR.data(); V.kp_lm_obs = a.obs.data();
        return V;
    }
    // MapPoints -> hs_landmark records.  `F` supplies hasAssociation(lm) (landMarkSizePixels, Frame.cc:296-300) through ONE reverse map instead
    // of the reference's linear scan per landmark; `prev` (may be null) supplies the previous frame's keypoint angle (rotation check).
    template <class T> static std::vector<hs_landmark> gather_landmarks(const std::vector<MapPoint*>& lms, T& F, const Frame* prev) {
        std::unordered_map<MapPoint*, int> assoc, assoc_prev;
        for (const auto& kv : F.getLandMarkMatches()) assoc.insert({ kv.second, kv.first });                 // ascending view index: the first view wins, like the scan
        if (prev) for (const auto& kv : const_cast<Frame*>(prev)->getLandMarkMatches()) assoc_prev.insert({ kv.second, kv.first });
        std::vector<hs_landmark> out(std::max<size_t>(lms.size(), 1));
        for (size_t i = 0; i < lms.size(); i++) {
            hs_landmark& L = out[i]; std::memset(&L, 0, sizeof(L));
            MapPoint* lm = lms[i];
            L.assoc_kp = -1;
            if (!lm) { L.skip = 1; continue; }
            const cv::Mat P = lm->GetWorldPos(), Nn = lm->GetNormal();
            for (int k = 0; k < 3; k++) { L.pos[k] = P.at<float>(k); L.normal[k] = Nn.at<float>(k); }
            L.size = lm->getSize();
            L.min_dist = lm->GetMinDistanceInvariance(); L.max_dist = lm->GetMaxDistanceInvariance();         // with dist_is_invariance_range = 1
            auto it = assoc.find(lm);
            if (it != assoc.end()) L.assoc_kp = it->second;
            if (prev) { auto ip = assoc_prev.find(lm); if (ip != assoc_prev.end()) L.prev_angle = prev->getViews().keypt(ip->second).angle; }
            const cv::Mat row = lm->GetDescriptor().rawDescriptor();
            std::memcpy(L.desc, row.ptr(0), HS_DESC_BYTES);
        }
        return out;
    }
    int project_and_associate(Frame& F, const std::vector<MapPoint*>& landmarks, const Frame* prev, const hs_proj_params& pp) {
        // address order == iteration order of the reference's std::map<MapPoint*, SingleMatchData> (FeatureMatcher.cc:64); duplicates collapse like map keys
        std::vector<MapPoint*> lms;
        lms.reserve(landmarks.size());
        for (MapPoint* p : landmarks) if (p) lms.push_back(p);
        std::sort(lms.begin(), lms.end());
        lms.erase(std::unique(lms.begin(), lms.end()), lms.end());
        if (lms.empty()) return 0;
        FrameArrays fa; hs_frame_view V = gather_frame(F, fa);
        std::vector<hs_landmark> L = gather_landmarks(lms, F, prev);
        std::vector<int32_t> midx(lms.size(), -1); std::vector<float> mdist(lms.size(), -1.f); int32_t n = 0;
        check(hs_search_by_projection(h, &V, L.data(), (int)lms.size(), &pp, midx.data(), mdist.data(), &n), "SearchByProjection");
        for (size_t i = 0; i < lms.size(); i++)          // replay, FeatureMatcher.cc:113-118
            if (midx[i] >= 0) F.associateLandMark(midx[i], lms[i], true);
        return n;
    }

    hs_orb* h;
};

}  // namespace HYSLAM
