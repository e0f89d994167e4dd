// bench_adaptor — what hySLAM itself would see through the C++ adaptors (hyslam_amd/host/), call by call:
//   stereo      ImageProcessing::ProcessStereoImage (src/main/ImageProcessing.cpp:69-116): left extractor on a spawned std::thread, right on the
//               caller (:82-84), FeatureViews, Stereomatcher(views, camera, settings) + computeStereoMatches + getData (:100-103)
//   localmap    TrackLocalMap::SearchLocalPoints (src/slam/tracking/TrackLocalMap.cpp:55-78): feature_factory->getFeatureMatcher()
//               ->SearchByProjection(frame, v_lmp, th) with a 50 000-landmark local map (BASELINE config 4)
//   triangulate LandMarkTriangulator.cpp:81: SearchForTriangulation between two key frames
// and where the time of a call goes: gather (hySLAM objects -> flat arrays; this is where FeatureDescriptor::rawDescriptor() clones a cv::Mat
// per descriptor and MapPoint::GetWorldPos()/GetNormal()/GetDescriptor() clone per landmark), the C-ABI call (H2D + kernels + D2H, synchronous)
// and scatter (flat results -> cv::KeyPoint / FeatureDescriptor vectors, associateLandMark replay).
// usage: bench_adaptor W H left.raw right.raw [reps] [n_landmarks]      (raw u8 frames)        prints one JSON object
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <set>
#include <thread>
#include <vector>
#include "../../hyslam_amd/host/HipORBFactory.h"

using namespace HYSLAM;
using clk = std::chrono::steady_clock;
static double ms(clk::time_point a, clk::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); }
static double median(std::vector<double> v) { std::sort(v.begin(), v.end()); return v.empty() ? 0.0 : v[v.size() / 2]; }

static bool read_raw(const char* path, size_t bytes, std::vector<uint8_t>& out)
{
    FILE* f = fopen(path, "rb");
    if (!f) return false;
    out.resize(bytes);
    const bool ok = fread(out.data(), 1, bytes, f) == bytes;
    fclose(f);
    return ok;
}

// FeatureUtil::extractFeatures (src/features/FeatureUtil.cpp): what the spawned thread of ImageProcessing.cpp:82 runs
static void extractFeatures(FeatureExtractor* ex, cv::Mat& img, std::vector<cv::KeyPoint>& keys, std::vector<FeatureDescriptor>& descs)
{
    (*ex)(img, cv::Mat(), keys, descs);
}

int main(int argc, char** argv)
{
    if (argc < 5) { printf("usage: bench_adaptor W H left.raw right.raw [reps] [n_landmarks] [cpu-list e.g. 0-7,128-135]\n"); return 2; }
    // optional confinement to a CPU list (one last-level-cache domain): applied HERE, before the first HIP call, so that no launcher that re-execs
    // (taskset, numactl) has to sit between a profiler's preloaded library and this program
    if (argc > 7 && argv[7][0]) {
        cpu_set_t set; CPU_ZERO(&set); int n_set = 0;
        for (const char* p = argv[7]; *p;) {
            char* e; long a = strtol(p, &e, 10), b = a;
            if (e == p) break;
            if (*e == '-') { const char* q = e + 1; b = strtol(q, &e, 10); if (e == q) break; }
            for (long c = a; c <= b && c < CPU_SETSIZE; c++) if (c >= 0) { CPU_SET((int)c, &set); n_set++; }
            p = (*e == ',') ? e + 1 : e;
            if (*e && *e != ',') break;
        }
        if (n_set == 0 || sched_setaffinity(0, sizeof set, &set) != 0) { printf("cannot confine to cpu list '%s'\n", argv[7]); return 4; }
    }
    const int w = atoi(argv[1]), h = atoi(argv[2]);
    const int reps = argc > 5 ? atoi(argv[5]) : 30, n_lm = argc > 6 ? atoi(argv[6]) : 50000;
    std::vector<uint8_t> rawL, rawR;
    if (!read_raw(argv[3], (size_t)w * h, rawL) || !read_raw(argv[4], (size_t)w * h, rawR)) { printf("cannot read frames\n"); return 3; }
    int ndev = 0;
    if (hs_device_count(&ndev) != HS_OK || ndev == 0) { printf("NO DEVICE\n"); return 0; }

    std::map<std::string, FeatureExtractorSettings> per_type;
    per_type["SLAM"].nFeatures = 2000;
    FeatureMatcherSettings ms_; ms_.nnratio = 0.8f;                       // slam_tracking_config.yaml:101-103
    std::unique_ptr<FeatureFactory> factory = std::make_unique<HipORBFactory>(per_type, ms_, 0);
    std::shared_ptr<FeatureExtractor> exL = factory->getExtractor("SLAM"), exR = factory->getExtractor("SLAM");      // ImageProcessing.cpp:31-32
    HipORBExtractor* hxL = static_cast<HipORBExtractor*>(exL.get());
    HipORBExtractor* hxR = static_cast<HipORBExtractor*>(exR.get());

    Camera cam; cam.sensor = 1;
    for (int i = 0; i < 9; i++) cam.K.at<float>(i / 3, i % 3) = 0.f;
    const float fx = 1050.f * w / 1920.f;
    cam.K.at<float>(0, 0) = fx; cam.K.at<float>(1, 1) = fx; cam.K.at<float>(0, 2) = w / 2.f - 0.5f; cam.K.at<float>(1, 2) = h / 2.f - 0.5f; cam.K.at<float>(2, 2) = 1.f;
    cam.mbf = fx * 0.12f; cam.mnMinX = 0; cam.mnMaxX = (float)w; cam.mnMinY = 0; cam.mnMaxY = (float)h;

    cv::Mat imL(h, w, CV_8UC1, rawL.data(), (size_t)w), imR(h, w, CV_8UC1, rawR.data(), (size_t)w);
    FeatureViews last_views;
    std::vector<double> t_total, t_extract, t_exL_abi, t_exL_scatter, t_views, t_sm_gather, t_sm_abi, t_getdata; int stereo_on_device = 0;
    for (int r = 0; r < reps + 3; r++) {
        const auto t0 = clk::now();
        std::vector<cv::KeyPoint> mvKeys, mvKeysRight; std::vector<FeatureDescriptor> mDescriptors, mDescriptorsRight;
        std::thread orb_thread(extractFeatures, exL.get(), std::ref(imL), std::ref(mvKeys), std::ref(mDescriptors));
        (*exR)(imR, cv::Mat(), mvKeysRight, mDescriptorsRight);
        orb_thread.join();
        const auto t1 = clk::now();
        FeatureExtractorSettings orb_params;
        FeatureViews LMviews(mvKeys, mvKeysRight, mDescriptors, mDescriptorsRight, orb_params);
        const auto t2 = clk::now();
        HipStereomatcher stereomatch(LMviews, cam, FeatureMatcherSettings());
        stereomatch.computeStereoMatches();
        const auto t3 = clk::now();
        stereomatch.getData(LMviews);
        const auto t4 = clk::now();
        if (r >= 3) {
            t_total.push_back(ms(t0, t4)); t_extract.push_back(ms(t0, t1)); t_views.push_back(ms(t1, t2)); t_getdata.push_back(ms(t3, t4));
            t_exL_abi.push_back(std::max(hxL->timing.abi_ms, hxR->timing.abi_ms)); t_exL_scatter.push_back(std::max(hxL->timing.scatter_ms, hxR->timing.scatter_ms));
            t_sm_gather.push_back(stereomatch.timing.gather_ms); t_sm_abi.push_back(stereomatch.timing.abi_ms);
            stereo_on_device = stereomatch.frames_on_device;
        }
        last_views = LMviews;
    }
    const int n_kp = last_views.numViews();
    int n_stereo = 0; for (int i = 0; i < n_kp; i++) n_stereo += last_views.depth(i) > 0;

    // ---- the optional one-call front end (HipStereoFrontend): one ticket per pair, synchronous, then two tickets in flight
    std::vector<double> f_total, f_abi, f_scatter, f_pipe;
    {
        HipStereoFrontend fe(factory->getDistanceFunc(), per_type["SLAM"], cam, FeatureMatcherSettings(), 0);
        for (int r = 0; r < reps + 3; r++) {
            const auto t0 = clk::now();
            FeatureViews v = fe.process(imL, imR);
            const auto t1 = clk::now();
            if (r >= 3) { f_total.push_back(ms(t0, t1)); f_abi.push_back(fe.timing.gather_ms + fe.timing.abi_ms); f_scatter.push_back(fe.timing.scatter_ms); }
            if (v.numViews() != n_kp) { printf("front end: %d views, extractors: %d\n", v.numViews(), n_kp); return 5; }
        }
        int32_t tk = fe.submit(imL, imR);
        const auto p0 = clk::now();
        for (int r = 0; r < reps; r++) {                      // submit pair r+1, then collect pair r: the upload of one runs under the kernels of the other
            const int32_t nx = fe.submit(imL, imR);
            FeatureViews v = fe.collect(tk);
            tk = nx;
        }
        f_pipe.push_back(ms(p0, clk::now()) / reps);
        fe.collect(tk);
    }

    // ---- TrackLocalMap: a local map of n_lm landmarks = the frame's keypoints back-projected at their stereo depth (or a seeded one), several noisy copies
    std::shared_ptr<DescriptorDistance> dist = factory->getDistanceFunc();
    cv::Mat Tcw(4, 4, CV_32F);
    for (int r = 0; r < 4; r++) for (int c = 0; c < 4; c++) Tcw.at<float>(r, c) = r == c ? 1.f : 0.f;
    Tcw.at<float>(0, 3) = 0.01f; Tcw.at<float>(1, 3) = -0.005f; Tcw.at<float>(2, 3) = 0.02f;
    std::mt19937 rng(4242);
    auto unif = [&](double a, double b) { return a + (b - a) * (double)(rng() & 0xFFFFFF) / (double)0x1000000; };
    std::vector<MapPoint*> lms(n_lm, nullptr);
    const float cx = cam.cx(), cy = cam.cy();
    for (int j = 0; j < n_lm && n_kp > 0; j++) {
        const int i = j % n_kp;
        const cv::KeyPoint k = last_views.keypt(i);
        double d = last_views.depth(i) > 0 ? last_views.depth(i) : unif(2.0, 25.0);
        d *= unif(0.97, 1.03);
        const double px = k.pt.x + unif(-1.5, 1.5), py = k.pt.y + unif(-1.5, 1.5);
        const double pc[3] = { (px - cx) * d / fx - Tcw.at<float>(0, 3), (py - cy) * d / fx - Tcw.at<float>(1, 3), d - Tcw.at<float>(2, 3) };
        const double dd = std::sqrt(pc[0] * pc[0] + pc[1] * pc[1] + pc[2] * pc[2]);
        MapPoint* m = new MapPoint();
        for (int c = 0; c < 3; c++) { m->mWorldPos.at<float>(c) = (float)pc[c]; m->mNormalVector.at<float>(c) = (float)(pc[c] / dd); }
        m->size = (float)(k.size * d / fx); m->mfMinDistance = (float)(dd * 0.5); m->mfMaxDistance = (float)(dd * 2.0); m->nObs = 2;
        cv::Mat row = last_views.descriptor(i).rawDescriptor();
        for (int b = 0, nb = (int)(rng() % 12); b < nb; b++) { const unsigned bit = rng() % 256; row.ptr(0)[bit >> 3] ^= (uint8_t)(1u << (bit & 7)); }
        m->mDescriptor = FeatureDescriptor(row, dist);
        lms[j] = m;
    }
    // TrackLocalMap keeps its local map in a std::set<MapPoint*> and hands the matcher a vector built from it (TrackLocalMap.cpp:73): address order
    std::set<MapPoint*> local_map_points(lms.begin(), lms.end());
    local_map_points.erase(static_cast<MapPoint*>(nullptr));
    const std::vector<MapPoint*> v_lmp(local_map_points.begin(), local_map_points.end());
    std::vector<double> p_total, p_gather, p_abi, p_scatter; int n_proj = 0; size_t replay_calls = 0, replay_full = 0; int replay_rule = -1, frame_on_device = 0;
    for (int r = 0; r < std::max(reps / 3, 5) + 2; r++) {
        Frame F(last_views, cam); F.SetPose(Tcw);
        const auto t0 = clk::now();
        std::unique_ptr<FeatureMatcher> matcher = factory->getFeatureMatcher();          // TrackLocalMap.cpp:72
        n_proj = matcher->SearchByProjection(F, v_lmp, 5.f);
        const auto t1 = clk::now();
        if (r >= 2) {
            const HipCallTiming& t = static_cast<HipFeatureMatcher*>(matcher.get())->timing;
            p_total.push_back(ms(t0, t1)); p_gather.push_back(t.gather_ms); p_abi.push_back(t.abi_ms); p_scatter.push_back(t.scatter_ms);
            const HipMatcherCore& core = static_cast<HipFeatureMatcher*>(matcher.get())->matcherCore();
            replay_calls = core.replay_calls; replay_full = core.replay_full; replay_rule = core.replay_rule; frame_on_device = core.frame_on_device;
        }
    }
    // ---- LandMarkTriangulator: two key frames with hashed feature vectors (a stand-in for DBoW2's, ~100 nodes like level 2 of ORBvoc on 2000 features)
    std::vector<double> q_total, q_gather, q_abi; int n_tri = 0;
    {
        KeyFrame K1(last_views, cam), K2(last_views, cam); K1.SetPose(Tcw); K2.SetPose(Tcw);
        for (int i = 0; i < n_kp; i++) { const uint8_t* d = last_views.descriptor(i).rawDescriptor().ptr(0); const unsigned node = 5u + 11u * ((d[1] ^ (d[20] << 2)) % 97u); K1.mFeatVec[node].push_back(i); K2.mFeatVec[node].push_back(i); }
        cv::Mat F12(3, 3, CV_32F);
        const float Fv[9] = { 0.f, -1e-6f, 2e-4f, 1e-6f, 0.f, -3e-3f, -2e-4f, 3e-3f, 0.f };      // a pure-translation fundamental matrix
        for (int i = 0; i < 9; i++) F12.at<float>(i / 3, i % 3) = Fv[i];
        for (int r = 0; r < std::max(reps / 3, 5) + 2; r++) {
            const auto t0 = clk::now();
            std::unique_ptr<FeatureMatcher> matcher = factory->getFeatureMatcher();
            std::vector<std::pair<size_t, size_t>> pairs;
            n_tri = matcher->SearchForTriangulation(&K1, &K2, F12, pairs, false);
            const auto t1 = clk::now();
            if (r >= 2) { const HipCallTiming& t = static_cast<HipFeatureMatcher*>(matcher.get())->timing; q_total.push_back(ms(t0, t1)); q_gather.push_back(t.gather_ms); q_abi.push_back(t.abi_ms); }
        }
    }
    printf("{\"frame\": \"%dx%d\", \"keypoints\": %d, \"stereo_matches\": %d, \"reps\": %d,\n"
           " \"HipStereoFrontend_ms\": {\"process_total\": %.3f, \"submit_plus_wait\": %.3f, \"FeatureViews_build\": %.3f, \"pipelined_per_pair\": %.3f},\n"
           " \"ProcessStereoImage_ms\": {\"total\": %.3f, \"extract_LR_threads\": %.3f, \"extract_c_abi\": %.3f, \"extract_scatter\": %.3f, \"FeatureViews_ctor\": %.3f,"
           " \"stereo_gather\": %.3f, \"stereo_c_abi\": %.3f, \"getData\": %.3f, \"stereo_frames_on_device\": %d},\n"
           " \"TrackLocalMap_SearchByProjection_ms\": {\"landmarks\": %d, \"matches\": %d, \"total\": %.3f, \"gather\": %.3f, \"c_abi\": %.3f, \"scatter\": %.3f, \"associateLandMark_calls\": %zu, \"of_full_replay\": %zu, \"replay_rule\": %d, \"frame_on_device\": %d},\n"
           " \"SearchForTriangulation_ms\": {\"matches\": %d, \"total\": %.3f, \"gather\": %.3f, \"c_abi\": %.3f}}\n",
           w, h, n_kp, n_stereo, reps, median(f_total), median(f_abi), median(f_scatter), median(f_pipe), median(t_total), median(t_extract), median(t_exL_abi), median(t_exL_scatter), median(t_views), median(t_sm_gather), median(t_sm_abi),
           median(t_getdata), stereo_on_device, n_lm, n_proj, median(p_total), median(p_gather), median(p_abi), median(p_scatter), replay_calls, replay_full, replay_rule, frame_on_device, n_tri, median(q_total), median(q_gather), median(q_abi));
    return 0;
}
