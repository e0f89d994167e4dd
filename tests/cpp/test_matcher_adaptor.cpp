// Exercises hyslam_amd/host/{HipFeatureMatcher,HipORBFactory,HipORBExtractor}.h the way hySLAM's call sites do:
//   TrackLocalMap::SearchLocalPoints   feature_factory->getFeatureMatcher()->SearchByProjection(frame, landmarks, th)      (TrackLocalMap.cpp:72-75)
//   TrackMotionModel::track            ->SearchByProjection(current, last, th, mono)                                       (TrackMotionModel.cpp:44,51)
//   TrackReferenceKeyFrame::track      ->SearchByBoW(pKF, frame, matches)                                                  (TrackReferenceKeyFrame.cpp:23)
//   LandMarkFuser                      ->Fuse(pKF, landmarks, fuse_matches)                                                (LandMarkFuser.cpp:57,83)
//   ImageProcessing::ProcessStereoImage  Stereomatcher(views, camera, settings)                                            (ImageProcessing.cpp:100-103)
// through std::unique_ptr<FeatureMatcher> obtained from the FeatureFactory base class (virtual dispatch after the INTEGRATION.md §3 patch),
// on Frame / KeyFrame / MapPoint objects (host/cv_compat.h) whose MapPoints are heap-allocated in SHUFFLED order, so that address order differs
// from array order (deviation D6).  Expected results: an independent gather written here + the CPU oracle (test infrastructure) + a replay of
// associateLandMark in address order on a copy of the frame's LandMarkMatches.
// usage: test_matcher_adaptor scene.bin          prints "MATCHER ADAPTOR OK ..." on success, "NO DEVICE" without a GPU
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>
#include "../../hyslam_amd/host/HipORBFactory.h"
#include "../../oracle/hs_oracle.h"

using namespace HYSLAM;

struct Scene {
    int n_kp = 0, n_lm = 0, sensor = 1, w = 0, h = 0;
    float pose[17];                                   // Rcw[9], tcw[3], fx, fy, cx, cy, mbf
    std::vector<hso_keypoint> kps; std::vector<uint8_t> desc; std::vector<float> uR; std::vector<int32_t> obs;
    std::vector<hso_landmark> lms;
};

static bool load(const char* path, Scene& s)
{
    FILE* f = fopen(path, "rb");
    if (!f) return false;
    int32_t hdr[8];
    bool ok = fread(hdr, 4, 8, f) == 8 && fread(s.pose, 4, 17, f) == 17;
    s.n_kp = hdr[0]; s.n_lm = hdr[1]; s.sensor = hdr[2]; s.w = hdr[3]; s.h = hdr[4];
    s.kps.resize(s.n_kp); s.desc.resize((size_t)s.n_kp * 32); s.uR.resize(s.n_kp); s.obs.resize(s.n_kp); s.lms.resize(s.n_lm);
    ok = ok && fread(s.kps.data(), sizeof(hso_keypoint), s.n_kp, f) == (size_t)s.n_kp && fread(s.desc.data(), 32, s.n_kp, f) == (size_t)s.n_kp &&
         fread(s.uR.data(), 4, s.n_kp, f) == (size_t)s.n_kp && fread(s.obs.data(), 4, s.n_kp, f) == (size_t)s.n_kp &&
         fread(s.lms.data(), sizeof(hso_landmark), s.n_lm, f) == (size_t)s.n_lm;
    fclose(f);
    return ok;
}

static FeatureViews make_views(const std::vector<hso_keypoint>& k, const std::vector<uint8_t>& d, const std::vector<float>& uR,
                               std::shared_ptr<DescriptorDistance> dist)
{
    std::vector<cv::KeyPoint> keys(k.size()); std::vector<FeatureDescriptor> descs;
    for (size_t i = 0; i < k.size(); i++) {
        keys[i].pt.x = k[i].x; keys[i].pt.y = k[i].y; keys[i].size = k[i].size; keys[i].angle = k[i].angle; keys[i].response = k[i].response; keys[i].octave = k[i].octave;
        descs.push_back(FeatureDescriptor(cv::Mat(1, 32, CV_8UC1, const_cast<uint8_t*>(d.data()) + i * 32, 32), dist));
    }
    FeatureExtractorSettings orb;      // default-constructed, like ImageProcessing.cpp:85,100: size_ref = 31, sigma_ref = 1
    std::vector<float> depth(k.size(), -1.f);
    return FeatureViews(keys, std::vector<cv::KeyPoint>(), uR, depth, descs, std::vector<FeatureDescriptor>(), orb);
}

static Camera make_camera(const Scene& s)
{
    Camera c; c.sensor = s.sensor;
    for (int i = 0; i < 9; i++) c.K.at<float>(i / 3, i % 3) = 0.f;
    c.K.at<float>(0, 0) = s.pose[12]; c.K.at<float>(1, 1) = s.pose[13]; c.K.at<float>(0, 2) = s.pose[14]; c.K.at<float>(1, 2) = s.pose[15]; c.K.at<float>(2, 2) = 1.f;
    c.mbf = s.pose[16]; c.mnMinX = 0; c.mnMaxX = (float)s.w; c.mnMinY = 0; c.mnMaxY = (float)s.h;
    return c;
}

static cv::Mat make_pose(const Scene& s)
{
    cv::Mat T(4, 4, CV_32F);
    for (int r = 0; r < 3; r++) { for (int c = 0; c < 3; c++) T.at<float>(r, c) = s.pose[3 * r + c]; T.at<float>(r, 3) = s.pose[9 + r]; }
    T.at<float>(3, 0) = T.at<float>(3, 1) = T.at<float>(3, 2) = 0.f; T.at<float>(3, 3) = 1.f;
    return T;
}

// independent gather of the frame fields the oracle needs (what HipFeatureMatcher::gather_frame must also arrive at)
template <class T> static hso_frame_view view_of(T& F, const Scene& s, cv::Mat Tcw, std::vector<int32_t>& obs)
{
    hso_frame_view V{};
    for (int r = 0; r < 3; r++) { for (int c = 0; c < 3; c++) V.Rcw[3 * r + c] = Tcw.at<float>(r, c); V.tcw[r] = Tcw.at<float>(r, 3); V.Ow[r] = F.GetCameraCenter().template at<float>(r); }
    V.fx = s.pose[12]; V.fy = s.pose[13]; V.cx = s.pose[14]; V.cy = s.pose[15]; V.mbf = s.pose[16]; V.sensor = s.sensor;
    V.min_x = 0; V.max_x = (float)s.w; V.min_y = 0; V.max_y = (float)s.h; V.size_ref = 31.f; V.n = s.n_kp;
    obs.assign(s.n_kp, -1);
    for (int i = 0; i < s.n_kp; i++) if (MapPoint* m = F.hasAssociation(i)) obs[i] = m->Observations();
    V.kps = s.kps.data(); V.desc = s.desc.data(); V.uR = s.uR.data(); V.kp_lm_obs = obs.data();
    return V;
}

#define FAIL(code, ...) do { printf(__VA_ARGS__); printf("\n"); return code; } while (0)

int main(int argc, char** argv)
{
    if (argc < 2) return 2;
    Scene s;
    if (!load(argv[1], s)) FAIL(3, "cannot read scene %s", argv[1]);

    int ndev = 0;
    std::map<std::string, FeatureExtractorSettings> per_type;
    per_type["SLAM"].nFeatures = 1000; per_type["Imaging"].nFeatures = 3000; per_type["Imaging"].fScaleFactor = 1.4f;
    FeatureMatcherSettings ms; ms.nnratio = 0.8f;
    std::unique_ptr<FeatureFactory> factory = std::make_unique<HipORBFactory>(per_type, ms, 0);       // System.cc:79 uses std::make_unique too
    if (hs_device_count(&ndev) != HS_OK || ndev == 0) {
        try { factory->getExtractor("SLAM"); } catch (const std::exception& e) { printf("NO DEVICE: %s\n", e.what()); return 0; }
        return 4;      // must have thrown: there is no CPU fallback
    }
    // ---- factory: extractors per camera type, settings, virtual matcher creation
    std::shared_ptr<FeatureExtractor> exS = factory->getExtractor("SLAM"), exI = factory->getExtractor("Imaging");
    if (exS->GetLevels() != 8 || exI->GetScaleFactor() != 1.4f || factory->getFeatureExtractorSettings().nFeatures != 3000) FAIL(5, "factory settings");
    std::unique_ptr<FeatureMatcher> matcher = factory->getFeatureMatcher();
    if (!dynamic_cast<HipFeatureMatcher*>(matcher.get())) FAIL(6, "getFeatureMatcher() did not dispatch to the HIP matcher");

    // ---- the scene as hySLAM objects; MapPoints allocated in shuffled order
    std::shared_ptr<DescriptorDistance> dist = factory->getDistanceFunc();
    const Camera cam = make_camera(s);
    const cv::Mat Tcw = make_pose(s);
    std::vector<int> order(s.n_lm);
    for (int i = 0; i < s.n_lm; i++) order[i] = i;
    std::mt19937 rng(12345);
    std::shuffle(order.begin(), order.end(), rng);
    std::vector<MapPoint*> lm(s.n_lm, nullptr);
    std::vector<void*> gaps;
    for (int j : order) {
        if (rng() % 3 == 0) gaps.push_back(malloc(16 + rng() % 200));        // perturb the allocator so addresses are not monotone in time either
        if (s.lms[j].skip) continue;
        MapPoint* m = new MapPoint();
        for (int k = 0; k < 3; k++) { m->mWorldPos.at<float>(k) = s.lms[j].pos[k]; m->mNormalVector.at<float>(k) = s.lms[j].normal[k]; }
        m->size = s.lms[j].size; m->mfMinDistance = s.lms[j].min_dist; m->mfMaxDistance = s.lms[j].max_dist;
        m->mDescriptor = FeatureDescriptor(cv::Mat(1, 32, CV_8UC1, s.lms[j].desc, 32), dist);
        m->nObs = 1 + (int)(rng() % 3);
        lm[j] = m;
    }
    bool monotone = true; for (int j = 1; j < s.n_lm; j++) if (lm[j] && lm[j - 1] && lm[j] < lm[j - 1]) monotone = false;
    if (monotone) FAIL(7, "allocation order did not shuffle the addresses");

    auto build_frame = [&](Frame& F) {
        F = Frame(make_views(s.kps, s.desc, s.uR, dist), cam);
        F.SetPose(Tcw);
        for (int j = 0; j < s.n_lm; j++) if (lm[j] && s.lms[j].assoc_kp >= 0) F.associateLandMark(s.lms[j].assoc_kp, lm[j], true);
        for (int i = 0; i < s.n_kp; i++)
            if (s.obs[i] >= 0 && !F.hasAssociation(i)) { MapPoint* m = new MapPoint(); m->nObs = s.obs[i]; F.associateLandMark(i, m, true); }
    };
    // expected outcome of one projection variant: independent gather + oracle + replay in address order
    auto expected = [&](Frame& F, std::vector<MapPoint*> cands, const Frame* prev, hso_proj_params pp, LandMarkMatches& out, int& n_out) {
        std::vector<MapPoint*> srt;
        for (MapPoint* p : cands) if (p) srt.push_back(p);
        std::sort(srt.begin(), srt.end()); srt.erase(std::unique(srt.begin(), srt.end()), srt.end());
        std::vector<int32_t> obs; hso_frame_view V = view_of(F, s, Tcw, obs);
        std::vector<hso_landmark> L(srt.size());
        for (size_t i = 0; i < srt.size(); i++) {
            const int j = (int)(std::find(lm.begin(), lm.end(), srt[i]) - lm.begin());
            L[i] = s.lms[j];
            L[i].min_dist = 0.8f * s.lms[j].min_dist; L[i].max_dist = 1.2f * s.lms[j].max_dist;      // what GetMin/MaxDistanceInvariance() return
            L[i].assoc_kp = F.hasAssociation(srt[i]); L[i].skip = 0;
            L[i].prev_angle = 0.f;
            if (prev) { int ip = prev->hasAssociation(srt[i]); if (ip >= 0) L[i].prev_angle = prev->getViews().keypt(ip).angle; }
        }
        pp.dist_is_invariance_range = 1;
        std::vector<int32_t> mi(srt.size()); std::vector<float> md(srt.size());
        n_out = hso_search_by_projection(&V, L.data(), (int)L.size(), &pp, mi.data(), md.data());
        out = F.getLandMarkMatches();
        for (size_t i = 0; i < srt.size(); i++) if (mi[i] >= 0) out.associateLandMark(mi[i], srt[i], true);
    };
    hso_proj_params base{}; base.frac_smaller = 0.5f; base.frac_larger = 1.5f; base.use_prev_matched = 1; base.max_view_angle = 1.047f;
    base.reproj_threshold = 5.99f; base.sigma_ref = 1.f; base.score_threshold = 100.f; base.second_best_ratio = 0.8f;

    // ---- TrackLocalMap variant
    int total_matches = 0;
    {
        Frame F; build_frame(F);
        hso_proj_params pp = base; pp.th = 5.f; pp.use_distance = 1; pp.use_stereo = 1;
        LandMarkMatches want; int n_want = 0; expected(F, lm, nullptr, pp, want, n_want);
        const int n_got = matcher->SearchByProjection(F, lm, 5.f);
        if (n_got != n_want || n_want < 50) FAIL(10, "local map: %d matches, expected %d", n_got, n_want);
        if (F.getLandMarkMatches().views_to_landmarks != want.views_to_landmarks) FAIL(11, "local map: associations differ after the replay");
        total_matches += n_got;
    }
    // ---- TrackMotionModel variant: the previous frame holds one keypoint per landmark whose angle is the record's prev_angle
    {
        Frame F; build_frame(F);
        std::vector<hso_keypoint> pk(s.n_lm); std::vector<uint8_t> pd((size_t)s.n_lm * 32, 0); std::vector<float> pu(s.n_lm, -1.f);
        for (int j = 0; j < s.n_lm; j++) { pk[j] = hso_keypoint{ 10.f + j % 600, 10.f + j / 600, 31.f, s.lms[j].prev_angle, 30.f, 0 }; }
        Frame Last(make_views(pk, pd, pu, dist), cam);
        Last.SetPose(Tcw);
        for (int j = 0; j < s.n_lm; j++) if (lm[j]) Last.associateLandMark(j, lm[j], true);
        hso_proj_params pp = base; pp.th = 7.f; pp.use_distance = 0; pp.use_stereo = 1; pp.check_rotation = 1;
        LandMarkMatches want; int n_want = 0; expected(F, Last.replicatemvpMapPoints(), &Last, pp, want, n_want);
        const int n_got = matcher->SearchByProjection(F, Last, 7.f, false);
        if (n_got != n_want || n_want < 50) FAIL(12, "last frame: %d matches, expected %d", n_got, n_want);
        if (F.getLandMarkMatches().views_to_landmarks != want.views_to_landmarks) FAIL(13, "last frame: associations differ after the replay");
        total_matches += n_got;
    }
    // ---- Fuse on a KeyFrame: vector order, first landmark per keypoint wins, pre-screen of bad / protected / already-observed landmarks
    {
        KeyFrame K(make_views(s.kps, s.desc, s.uR, dist), cam);
        K.SetPose(Tcw);
        std::vector<MapPoint*> cands = lm;
        for (int j = 0; j < s.n_lm; j += 9) if (lm[j]) lm[j]->mbBad = true;
        for (int j = 4; j < s.n_lm; j += 13) if (lm[j]) lm[j]->n_protected = 1;
        for (int j = 2; j < s.n_lm; j += 17) if (lm[j]) lm[j]->in_keyframes.insert(&K);
        std::vector<int32_t> obs; hso_frame_view V = view_of(K, s, Tcw, obs);
        std::vector<hso_landmark> L(s.n_lm);
        for (int j = 0; j < s.n_lm; j++) {
            L[j] = s.lms[j];
            L[j].min_dist = 0.8f * s.lms[j].min_dist; L[j].max_dist = 1.2f * s.lms[j].max_dist; L[j].assoc_kp = -1;
            L[j].skip = !lm[j] || lm[j]->isBad() || lm[j]->IsInKeyFrame(&K) || lm[j]->Protected();
        }
        hso_proj_params pp = base; pp.th = 3.f; pp.score_threshold = 50.f; pp.second_best_ratio = 1.f; pp.use_distance = 1; pp.use_prev_matched = 0;
        pp.use_viewing_angle = 1; pp.use_reprojection = 1; pp.first_wins = 1; pp.dist_is_invariance_range = 1;
        std::vector<int32_t> mi(s.n_lm); std::vector<float> md(s.n_lm);
        hso_search_by_projection(&V, L.data(), s.n_lm, &pp, mi.data(), md.data());
        std::map<size_t, MapPoint*> want, got;
        for (int j = 0; j < s.n_lm; j++) if (mi[j] >= 0) want.insert({ (size_t)mi[j], lm[j] });
        const int n_got = matcher->Fuse(&K, cands, got, 3.f, 5.99f);
        if (got != want || n_got != (int)want.size() || want.size() < 30) FAIL(14, "Fuse: %d matches, expected %zu", n_got, want.size());
        for (int j = 0; j < s.n_lm; j++) if (lm[j]) { lm[j]->mbBad = false; lm[j]->n_protected = 0; lm[j]->in_keyframes.clear(); }
        total_matches += n_got;
    }
    // ---- SearchByBoW(KeyFrame, Frame): a hashed stand-in for DBoW2's feature vectors on both sides
    {
        KeyFrame K(make_views(s.kps, s.desc, s.uR, dist), cam);
        std::vector<hso_keypoint> k2 = s.kps; std::vector<uint8_t> d2 = s.desc;
        for (int i = 0; i < s.n_kp; i += 2) d2[(size_t)i * 32 + 5] ^= 0x24;
        std::reverse(k2.begin(), k2.end());
        for (int i = 0; i < s.n_kp / 2; i++) for (int b = 0; b < 32; b++) std::swap(d2[(size_t)i * 32 + b], d2[(size_t)(s.n_kp - 1 - i) * 32 + b]);
        Frame F(make_views(k2, d2, std::vector<float>(s.n_kp, -1.f), dist), cam);
        auto node_of = [](const uint8_t* d) { return 3u + 7u * ((d[0] ^ (d[9] << 1)) % 61u); };
        for (int i = 0; i < s.n_kp; i++) { K.mFeatVec[node_of(&s.desc[(size_t)i * 32])].push_back(i); F.mFeatVec[node_of(&d2[(size_t)i * 32])].push_back(i); }
        for (int i = 0; i < s.n_kp; i += 3) K.associateLandMark(i, lm[std::min(i, s.n_lm - 1)] ? lm[std::min(i, s.n_lm - 1)] : new MapPoint(), true);
        std::vector<uint8_t> keep(s.n_kp, 0);
        for (int i = 0; i < s.n_kp; i++) if (MapPoint* m = K.hasAssociation(i)) keep[i] = !m->isBad();
        auto csr = [](const DBoW2::FeatureVector& fv, std::vector<int32_t>& id, std::vector<int32_t>& ptr, std::vector<int32_t>& idx) {
            ptr.push_back(0);
            for (const auto& kv : fv) { id.push_back((int32_t)kv.first); for (unsigned i : kv.second) idx.push_back((int32_t)i); ptr.push_back((int32_t)idx.size()); }
        };
        std::vector<int32_t> i1, p1, x1, i2, p2, x2; csr(K.mFeatVec, i1, p1, x1); csr(F.mFeatVec, i2, p2, x2);
        std::vector<int32_t> m12(s.n_kp, -1);
        const int n_want = hso_search_by_bow(s.kps.data(), s.desc.data(), s.n_kp, i1.data(), p1.data(), x1.data(), (int)i1.size(),
                                             k2.data(), d2.data(), s.n_kp, i2.data(), p2.data(), x2.data(), (int)i2.size(), keep.data(), 50.f, 0.8f, 1, m12.data());
        std::map<size_t, MapPoint*> want, got;
        for (int i = 0; i < s.n_kp; i++) if (m12[i] >= 0) want[(size_t)m12[i]] = K.hasAssociation(i);
        const int n_got = matcher->SearchByBoW(&K, F, got);
        if (n_got != n_want || got != want || n_want < 20) FAIL(15, "SearchByBoW: %d matches, expected %d", n_got, n_want);
        total_matches += n_got;
    }
    // ---- Stereomatcher with the reference's constructor (FeatureViews, Camera, FeatureMatcherSettings): ImageProcessing.cpp:100-103
    {
        std::vector<cv::KeyPoint> keys(s.n_kp), keysR(s.n_kp); std::vector<FeatureDescriptor> dL, dR;
        std::vector<hso_keypoint> kR = s.kps;
        for (int i = 0; i < s.n_kp; i++) kR[i].x -= 3.0f + (i % 40);
        for (int i = 0; i < s.n_kp; i++) {
            keys[i].pt.x = s.kps[i].x; keys[i].pt.y = s.kps[i].y; keys[i].size = s.kps[i].size; keys[i].octave = s.kps[i].octave;
            keysR[i] = keys[i]; keysR[i].pt.x = kR[i].x;
            dL.push_back(FeatureDescriptor(cv::Mat(1, 32, CV_8UC1, s.desc.data() + (size_t)i * 32, 32), dist)); dR.push_back(dL.back());
        }
        FeatureViews views(keys, keysR, dL, dR, FeatureExtractorSettings());
        HipStereomatcher sm(views, cam, FeatureMatcherSettings());
        sm.computeStereoMatches();
        std::vector<float> uR, depth; sm.getData(uR, depth);
        sm.getData(views);
        hso_stereo_params sp{ cam.fx(), cam.mbf, (int)cam.mnMaxY, 100.f, 50.f, 31.f };
        std::vector<float> ou(s.n_kp), od(s.n_kp);
        hso_stereo_match(s.kps.data(), s.desc.data(), s.n_kp, kR.data(), s.desc.data(), s.n_kp, &sp, ou.data(), od.data(), nullptr, nullptr);
        for (int i = 0; i < s.n_kp; i++) if (uR[i] != ou[i] || depth[i] != od[i] || views.uR(i) != ou[i] || views.depth(i) != od[i]) FAIL(16, "stereo %d differs", i);
    }
    printf("MATCHER ADAPTOR OK %d keypoints, %d landmarks, %d matches over 4 searches\n", s.n_kp, s.n_lm, total_matches);
    return 0;
}
