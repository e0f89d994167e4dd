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
#include <type_traits>
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <thread>
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
#ifdef HYSLAM_AMD_COMPAT_UNPATCHED
    // integration (b): the UNPATCHED reference declarations (non-virtual everywhere) + host/replace/FeatureMatcher.cc in the place of the reference's
    // FeatureMatcher.cc: `matcher` is a plain FeatureMatcher made by the base class's non-virtual getFeatureMatcher(); every search below goes
    // through the replaced member functions on the calling thread's handle.  (Nothing to dynamic_cast to: the subclass does not exist in this mode.)
    static_assert(!std::is_polymorphic<FeatureMatcher>::value, "the unpatched FeatureMatcher has no virtual function");
    {
        hs_orb* mine = hip_detail::thread_handle(hip_detail::default_device().load(), "test");
        hs_orb* theirs = nullptr;
        std::thread t([&] { theirs = hip_detail::thread_handle(hip_detail::default_device().load(), "test"); });
        t.join();
        if (!theirs || theirs == mine) FAIL(9, "two threads share one matcher handle");
    }
#else
    if (!dynamic_cast<HipFeatureMatcher*>(matcher.get())) FAIL(6, "getFeatureMatcher() did not dispatch to the HIP matcher");
    {   // matcher handles are per (calling thread, device): the same handle again on this thread, the factory's device, another handle on another thread
        hs_orb* mine = static_cast<HipFeatureMatcher*>(matcher.get())->handle();
        if (hs_orb_get_device(mine) != 0 || static_cast<HipFeatureMatcher*>(factory->getFeatureMatcher().get())->handle() != mine) FAIL(8, "matcher handle is not the calling thread's handle on the factory's device");
        std::unique_ptr<FeatureFactory> other = std::make_unique<HipORBFactory>(per_type, ms, 0);
        if (static_cast<HipFeatureMatcher*>(other->getFeatureMatcher().get())->handle() != mine) FAIL(8, "a second factory on the same device must reuse the thread's handle");
        hs_orb* theirs = nullptr;
        std::thread t([&] { theirs = static_cast<HipFeatureMatcher*>(factory->getFeatureMatcher().get())->handle(); });
        t.join();
        if (!theirs || theirs == mine) FAIL(9, "two threads share one matcher handle");
    }
#endif

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
        if (F.getLandMarkMatches().outliers != want.outliers || F.getLandMarkMatches().n_matches != want.n_matches) FAIL(11, "local map: outliers / n_matches differ after the (planned) replay");
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
        if (F.getLandMarkMatches().outliers != want.outliers || F.getLandMarkMatches().n_matches != want.n_matches) FAIL(13, "last frame: outliers / n_matches differ after the (planned) replay");
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
    // ================= the key-frame entry points (LandMarkTriangulator.cpp:81, LoopClosing.cc:275,333,389) =================
    // Two key frames that see the same landmarks: every keypoint of KF1 owns a landmark back-projected with KF1's pose at a seeded depth;
    // KF2 = the same keypoints in permuted order with sub-pixel jitter, posed so that x_c1 = s12 R12 x_c2 + t12.
    {
        const int n = s.n_kp;
        const float fx = s.pose[12], fy = s.pose[13], cx = s.pose[14], cy = s.pose[15];
        std::mt19937 r2(777);
        auto unif = [&](double a, double b) { return a + (b - a) * (double)(r2() & 0xFFFFFF) / (double)0x1000000; };
        const double R1[9] = { s.pose[0], s.pose[1], s.pose[2], s.pose[3], s.pose[4], s.pose[5], s.pose[6], s.pose[7], s.pose[8] };
        const double t1[3] = { s.pose[9], s.pose[10], s.pose[11] };
        std::vector<hso_landmark> own1(n);                                   // landmark of KF1's keypoint i (flat record, invariance range in min/max_dist)
        std::vector<MapPoint*> mp1(n, nullptr);
        for (int i = 0; i < n; i++) {
            const double d = unif(3.0, 20.0);
            const double pc[3] = { (s.kps[i].x - cx) * d / fx - t1[0], (s.kps[i].y - cy) * d / fy - t1[1], d - t1[2] };
            double pw[3], ow[3];
            for (int k = 0; k < 3; k++) { pw[k] = R1[k] * pc[0] + R1[3 + k] * pc[1] + R1[6 + k] * pc[2]; ow[k] = -(R1[k] * t1[0] + R1[3 + k] * t1[1] + R1[6 + k] * t1[2]); }
            const double dist = std::sqrt((pw[0] - ow[0]) * (pw[0] - ow[0]) + (pw[1] - ow[1]) * (pw[1] - ow[1]) + (pw[2] - ow[2]) * (pw[2] - ow[2]));
            hso_landmark& L = own1[i]; std::memset(&L, 0, sizeof(L));
            for (int k = 0; k < 3; k++) { L.pos[k] = (float)pw[k]; L.normal[k] = (float)((pw[k] - ow[k]) / dist); }
            L.size = (float)(s.kps[i].size * d / fx);
            L.min_dist = (float)(dist * unif(0.4, 1.15)); L.max_dist = (float)(dist * unif(0.9, 2.5));      // mfMin/MaxDistance: some fall outside 0.8 / 1.2
            std::memcpy(L.desc, &s.desc[(size_t)i * 32], 32);
            L.assoc_kp = -1;
        }
        std::vector<int> alloc_order(n); for (int i = 0; i < n; i++) alloc_order[i] = i;
        std::shuffle(alloc_order.begin(), alloc_order.end(), r2);
        for (int i : alloc_order) {
            MapPoint* m = new MapPoint();
            for (int k = 0; k < 3; k++) { m->mWorldPos.at<float>(k) = own1[i].pos[k]; m->mNormalVector.at<float>(k) = own1[i].normal[k]; }
            m->size = own1[i].size; m->mfMinDistance = own1[i].min_dist; m->mfMaxDistance = own1[i].max_dist;
            m->mDescriptor = FeatureDescriptor(cv::Mat(1, 32, CV_8UC1, own1[i].desc, 32), dist); m->nObs = 2;
            mp1[i] = m;
        }
        auto flat_of = [&](MapPoint* m) {                                    // independent gather of one MapPoint (what gather_landmarks must arrive at)
            hso_landmark L; std::memset(&L, 0, sizeof(L)); L.assoc_kp = -1;
            if (!m) { L.skip = 1; return L; }
            for (int k = 0; k < 3; k++) { L.pos[k] = m->mWorldPos.at<float>(k); L.normal[k] = m->mNormalVector.at<float>(k); }
            L.size = m->size; L.min_dist = 0.8f * m->mfMinDistance; L.max_dist = 1.2f * m->mfMaxDistance;
            std::memcpy(L.desc, m->mDescriptor.rawDescriptor().ptr(0), 32);
            return L;
        };
        // KF2
        std::vector<int> perm(n); for (int i = 0; i < n; i++) perm[i] = i;
        std::shuffle(perm.begin(), perm.end(), r2);
        std::vector<hso_keypoint> k2(n); std::vector<uint8_t> d2((size_t)n * 32); std::vector<float> u2(n);
        for (int i = 0; i < n; i++) {
            k2[i] = s.kps[perm[i]]; k2[i].x += (float)unif(-0.9, 0.9); k2[i].y += (float)unif(-0.9, 0.9);
            std::memcpy(&d2[(size_t)i * 32], &s.desc[(size_t)perm[i] * 32], 32);
            if (i % 3 == 0) d2[(size_t)i * 32 + 11] ^= 0x18;
            u2[i] = (i % 4 == 1) ? -1.f : s.uR[perm[i]];
        }
        const float s12 = 1.0f / 1.1f;
        const double ax = 0.001, ay = 0.002, az = -0.001;
        const double cxr = std::cos(ax), sxr = std::sin(ax), cyr = std::cos(ay), syr = std::sin(ay), czr = std::cos(az), szr = std::sin(az);
        const double Rm[9] = { czr * cyr, czr * syr * sxr - szr * cxr, czr * syr * cxr + szr * sxr, szr * cyr, szr * syr * sxr + czr * cxr, szr * syr * cxr - czr * sxr, -syr, cyr * sxr, cyr * cxr };
        cv::Mat R12(3, 3, CV_32F), t12(3, 1, CV_32F);
        for (int i = 0; i < 9; i++) R12.at<float>(i / 3, i % 3) = (float)Rm[i];
        t12.at<float>(0) = 0.02f; t12.at<float>(1) = -0.01f; t12.at<float>(2) = 0.03f;
        cv::Mat T2(4, 4, CV_32F);                                             // T2w = (1/s12) R12^T (T1w - t12)
        for (int r = 0; r < 3; r++) {
            for (int c = 0; c < 3; c++) { double a = 0; for (int k = 0; k < 3; k++) a += (double)R12.at<float>(k, r) * R1[3 * k + c]; T2.at<float>(r, c) = (float)a; }
            double a = 0; for (int k = 0; k < 3; k++) a += (double)R12.at<float>(k, r) * (t1[k] - (double)t12.at<float>(k));
            T2.at<float>(r, 3) = (float)(a / s12);
        }
        T2.at<float>(3, 0) = T2.at<float>(3, 1) = T2.at<float>(3, 2) = 0.f; T2.at<float>(3, 3) = 1.f;
        Scene s2 = s; s2.kps = k2; s2.desc = d2; s2.uR = u2;
        for (int r = 0; r < 3; r++) { for (int c = 0; c < 3; c++) s2.pose[3 * r + c] = T2.at<float>(r, c); s2.pose[9 + r] = T2.at<float>(r, 3); }

        auto fresh_keyframes = [&](KeyFrame& K1, KeyFrame& K2) {
            K1 = KeyFrame(make_views(s.kps, s.desc, s.uR, dist), cam); K1.SetPose(Tcw);
            K2 = KeyFrame(make_views(k2, d2, u2, dist), cam); K2.SetPose(T2);
        };
        auto csr = [](const DBoW2::FeatureVector& fv, std::vector<int32_t>& id, std::vector<int32_t>& ptr, std::vector<int32_t>& idx) {
            ptr.push_back(0);
            for (const auto& kv : fv) { id.push_back((int32_t)kv.first); for (unsigned i : kv.second) idx.push_back((int32_t)i); ptr.push_back((int32_t)idx.size()); }
        };
        auto node_of = [](const uint8_t* d) { return 5u + 11u * ((d[1] ^ (d[20] << 2)) % 97u); };

        // ---- SearchByProjection(pKF, Scw, vpPoints, vpMatched, th): LoopClosing.cc:389
        {
            KeyFrame K1, K2; fresh_keyframes(K1, K2);
            for (int i = 0; i < n; i += 7) K1.associateLandMark(i, mp1[i], true);                  // landMarkSizePixels then uses the keypoint's size
            cv::Mat Scw(4, 4, CV_32F);
            const float sc = 1.07f, dt[3] = { 0.01f, -0.02f, 0.015f };
            for (int r = 0; r < 3; r++) { for (int c = 0; c < 3; c++) Scw.at<float>(r, c) = sc * Tcw.at<float>(r, c); Scw.at<float>(r, 3) = sc * (Tcw.at<float>(r, 3) + dt[r]); }
            Scw.at<float>(3, 0) = Scw.at<float>(3, 1) = Scw.at<float>(3, 2) = 0.f; Scw.at<float>(3, 3) = 1.f;
            std::vector<MapPoint*> pts = mp1;
            for (int i = 0; i < n; i += 2) pts.push_back(mp1[i]);                                   // duplicates: the second copy must lose its keypoint
            std::shuffle(pts.begin(), pts.end(), r2);
            for (int i = 3; i < n; i += 19) mp1[i]->mbBad = true;
            std::vector<MapPoint*> matched(n, nullptr);
            for (int i = 0; i < n; i += 10) matched[i] = (i % 20 == 0) ? mp1[(i + 1) % n] : new MapPoint();      // pre-matched views; half of them hold candidates (spAlreadyFound)
            std::set<MapPoint*> found(matched.begin(), matched.end()); found.erase(nullptr);
            std::vector<int32_t> obs; hso_frame_view V = view_of(K1, s, Tcw, obs);
            std::vector<hso_landmark> L(pts.size());
            for (size_t i = 0; i < pts.size(); i++) { L[i] = flat_of(pts[i]); L[i].assoc_kp = K1.hasAssociation(pts[i]); L[i].skip = pts[i]->isBad() || found.count(pts[i]); }
            std::vector<uint8_t> taken(n); for (int i = 0; i < n; i++) taken[i] = matched[i] != nullptr;
            std::vector<int32_t> mi(pts.size());
            float S[16]; for (int r = 0; r < 4; r++) for (int c = 0; c < 4; c++) S[4 * r + c] = Scw.at<float>(r, c);
            const int n_want = hso_search_by_projection_sim3(&V, S, L.data(), (int)L.size(), 4, 50.f, taken.data(), mi.data());
            std::vector<MapPoint*> want = matched;
            for (size_t i = 0; i < pts.size(); i++) if (mi[i] >= 0) want[mi[i]] = pts[i];
            const int n_got = matcher->SearchByProjection(&K1, Scw, pts, matched, 4);
            if (n_got != n_want || n_want < 100) FAIL(20, "SearchByProjection(Scw): %d matches, expected %d", n_got, n_want);
            if (matched != want) FAIL(21, "SearchByProjection(Scw): vpMatched differs");
            for (int i = 0; i < n; i++) mp1[i]->mbBad = false;
            total_matches += n_got;
        }
        // ---- SearchBySim3(pKF1, pKF2, vpMatches12, s12, R12, t12, th): LoopClosing.cc:333
        {
            KeyFrame K1, K2; fresh_keyframes(K1, K2);
            for (int i = 0; i < n; i++) if (i % 12 != 5) K1.associateLandMark(i, mp1[i], true);     // a twelfth of KF1's keypoints own no landmark
            for (int i = 0; i < n; i++) if (i % 10 != 3) K2.associateLandMark(i, mp1[perm[i]], true);
            for (int i = 6; i < n; i += 23) mp1[i]->mbBad = true;
            std::vector<MapPoint*> m12(n, nullptr);
            for (int i = 0; i < n; i += 15) if (K1.hasAssociation(i)) {                              // already matched pairs (vbAlreadyMatched1 / 2)
                m12[i] = K1.hasAssociation(i);
                const int i2 = K2.hasAssociation(m12[i]);
                if (i2 >= 0) m12[i]->mObservations[&K2] = (size_t)i2;
            }
            const std::vector<MapPoint*> v1 = K1.GetMapPointMatches(), v2 = K2.GetMapPointMatches();
            std::vector<uint8_t> a1(n, 0), a2(n, 0);
            for (int i = 0; i < n; i++) if (m12[i]) { a1[i] = 1; const int i2 = m12[i]->GetIndexInKeyFrame(&K2); if (i2 >= 0 && i2 < n) a2[i2] = 1; }
            std::vector<int32_t> o1, o2; hso_frame_view V1 = view_of(K1, s, Tcw, o1), V2 = view_of(K2, s2, T2, o2);
            std::vector<hso_landmark> L1(n), L2(n);
            for (int i = 0; i < n; i++) {
                L1[i] = flat_of(v1[i]); if (v1[i]) { L1[i].assoc_kp = K2.hasAssociation(v1[i]); L1[i].skip = a1[i] || v1[i]->isBad(); }
                L2[i] = flat_of(v2[i]); if (v2[i]) { L2[i].assoc_kp = K1.hasAssociation(v2[i]); L2[i].skip = a2[i] || v2[i]->isBad(); }
            }
            float Rf[9], tf[3]; for (int r = 0; r < 3; r++) { for (int c = 0; c < 3; c++) Rf[3 * r + c] = R12.at<float>(r, c); tf[r] = t12.at<float>(r); }
            std::vector<int32_t> om(n);
            const int n_want = hso_search_by_sim3(&V1, L1.data(), &V2, L2.data(), s12, Rf, tf, 7.5f, 100.f, om.data());
            std::vector<MapPoint*> want = m12;
            for (int i = 0; i < n; i++) if (om[i] >= 0) want[i] = v2[om[i]];
            const int n_got = matcher->SearchBySim3(&K1, &K2, m12, s12, R12, t12, 7.5f);
            if (n_got != n_want || n_want < 100) FAIL(22, "SearchBySim3: %d found, expected %d", n_got, n_want);
            if (m12 != want) FAIL(23, "SearchBySim3: vpMatches12 differs");
            for (int i = 0; i < n; i++) { mp1[i]->mbBad = false; mp1[i]->mObservations.clear(); }
            total_matches += n_got;
        }
        // ---- SearchForTriangulation (LandMarkTriangulator.cpp:81) and SearchByBoW2 (LoopClosing.cc:275) on hashed feature vectors
        for (int variant = 0; variant < 3; variant++) {                       // 0: triangulation, all views; 1: triangulation, bOnlyStereo; 2: SearchByBoW2
            KeyFrame K1, K2; fresh_keyframes(K1, K2);
            for (int i = 0; i < n; i++) { K1.mFeatVec[node_of(&s.desc[(size_t)i * 32])].push_back(i); K2.mFeatVec[node_of(&d2[(size_t)i * 32])].push_back(i); }
            const int gap1 = variant == 2 ? 7 : 2, gap2 = variant == 2 ? 9 : 3;                      // BoW2 wants matched views, triangulation un-matched ones
            for (int i = 0; i < n; i++) if ((i % gap1 != 0) == (variant == 2)) K1.associateLandMark(i, mp1[i], true);
            for (int i = 0; i < n; i++) if ((i % gap2 != 0) == (variant == 2)) K2.associateLandMark(i, mp1[perm[i]], true);
            for (int i = 1; i < n; i += 5) mp1[i]->mbBad = true;                                     // a bad landmark counts as "no landmark" (MatchCriteria.cpp:559-563)
            const bool only_stereo = variant == 1;
            auto mask = [&](KeyFrame& K, const std::vector<float>& uR, bool keep_matched) {
                std::vector<uint8_t> keep(n, 0);
                for (int i = 0; i < n; i++) {
                    MapPoint* m = K.hasAssociation(i); const bool has = m && !m->isBad();
                    keep[i] = has == keep_matched && (!only_stereo || K.getCamera().sensor == 0 || uR[i] >= 0);
                }
                return keep;
            };
            std::vector<uint8_t> keep1 = mask(K1, s.uR, variant == 2), keep2 = mask(K2, u2, variant == 2);
            std::vector<int32_t> i1, p1, x1, i2, p2, x2; csr(K1.mFeatVec, i1, p1, x1); csr(K2.mFeatVec, i2, p2, x2);
            // F12 = K^-T [t12]x R12 K^-1 (GenUtils::ComputeF12): x1' F12 x2 = 0 for x_c1 = s12 R12 x_c2 + t12
            cv::Mat F12(3, 3, CV_32F);
            {
                const double tx[9] = { 0, -t12.at<float>(2), t12.at<float>(1), t12.at<float>(2), 0, -t12.at<float>(0), -t12.at<float>(1), t12.at<float>(0), 0 };
                double E[9], Ki[9] = { 1.0 / fx, 0, -cx / fx, 0, 1.0 / fy, -cy / fy, 0, 0, 1 }, A[9];
                for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) { double a = 0; for (int k = 0; k < 3; k++) a += tx[3 * r + k] * Rm[3 * k + c]; E[3 * r + c] = a; }
                for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) { double a = 0; for (int k = 0; k < 3; k++) a += Ki[3 * k + r] * E[3 * k + c]; A[3 * r + c] = a; }
                for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) { double a = 0; for (int k = 0; k < 3; k++) a += A[3 * r + k] * Ki[3 * k + c]; F12.at<float>(r, c) = (float)a; }
            }
            float Ff[9]; for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) Ff[3 * r + c] = F12.at<float>(r, c);
            std::vector<int32_t> om(n, -1);
            const int n_want = hso_search_by_bow_ex(s.kps.data(), s.desc.data(), n, i1.data(), p1.data(), x1.data(), (int)i1.size(),
                                                    k2.data(), d2.data(), n, i2.data(), p2.data(), x2.data(), (int)i2.size(), keep1.data(), keep2.data(),
                                                    variant == 2 ? nullptr : Ff, 31.f, 1.f, 50.f, variant == 2 ? 0.8f : 1.0f, 1, om.data());
            if (variant == 2) {
                {   // the legacy SearchByBoW(KF1, KF2) on the same key frames: exclusive use of KF2's views, angle1 - angle2 histogram
                    std::vector<int32_t> ol(n, -1);
                    const int nl_want = hso_search_by_bow_legacy(s.kps.data(), s.desc.data(), n, i1.data(), p1.data(), x1.data(), (int)i1.size(),
                                                                 k2.data(), d2.data(), n, i2.data(), p2.data(), x2.data(), (int)i2.size(), keep1.data(), keep2.data(), 50.f, 0.8f, 1, ol.data());
                    std::vector<MapPoint*> wl(n, nullptr), gl;
                    for (int i = 0; i < n; i++) if (ol[i] >= 0) wl[i] = K2.hasAssociation(ol[i]);
                    const int nl_got = matcher->SearchByBoW(&K1, &K2, gl);
                    if (nl_got != nl_want || nl_want < 20) FAIL(29, "SearchByBoW(KF, KF): %d matches, expected %d", nl_got, nl_want);
                    if (gl != wl) FAIL(30, "SearchByBoW(KF, KF): vpMatches12 differs");
                    total_matches += nl_got;
                }
                std::vector<MapPoint*> want(n, nullptr), got;
                for (int i = 0; i < n; i++) if (om[i] >= 0) want[i] = K2.hasAssociation(om[i]);
                const int n_got = matcher->SearchByBoW2(&K1, &K2, got);
                if (n_got != n_want || n_want < 20) FAIL(26, "SearchByBoW2: %d matches, expected %d", n_got, n_want);
                if (got != want) FAIL(27, "SearchByBoW2: vpMatches12 differs");
                total_matches += n_got;
            } else {
                std::vector<std::pair<size_t, size_t>> want, got;
                for (int i = 0; i < n; i++) if (om[i] >= 0) want.push_back({ (size_t)i, (size_t)om[i] });
                const int n_got = matcher->SearchForTriangulation(&K1, &K2, F12, got, only_stereo);
                if (n_got != n_want || n_want < 20) FAIL(24, "SearchForTriangulation(%d): %d matches, expected %d", variant, n_got, n_want);
                if (got != want) FAIL(25, "SearchForTriangulation(%d): pairs differ", variant);
                total_matches += n_got;
            }
            for (int i = 0; i < n; i++) mp1[i]->mbBad = false;
        }
        // ---- Fuse(pKF, Scw, ...): an empty body in the reference (FeatureMatcher.cc:523-624); the override must leave its output alone
        {
            KeyFrame K1, K2; fresh_keyframes(K1, K2);
            std::vector<MapPoint*> repl(3, mp1[0]);
            if (matcher->Fuse(&K1, Tcw, mp1, 4.f, repl) != 0 || repl != std::vector<MapPoint*>(3, mp1[0])) FAIL(28, "Fuse(Scw) touched its output");
        }
    }
    printf("MATCHER ADAPTOR OK %d keypoints, %d landmarks, %d matches over 10 searches\n", s.n_kp, s.n_lm, total_matches);
    return 0;
}
