// Exercises hyslam_amd/host/HipORBExtractor.h the way ImageProcessing::ProcessStereoImage uses an extractor and the
// stereo matcher (src/main/ImageProcessing.cpp:82-103), and compares with the CPU oracle (test infrastructure).
// usage: test_adaptor W H  < left.raw right.raw on stdin        prints "ADAPTOR OK ..." on success
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../../hyslam_amd/host/HipORBExtractor.h"
#include "../../oracle/hs_oracle.h"

int main(int argc, char** argv)
{
    if (argc < 3) return 2;
    const int W = atoi(argv[1]), H = atoi(argv[2]);
    cv::Mat L(H, W), R(H, W);
    if (fread(L.data, 1, (size_t)W * H, stdin) != (size_t)W * H || fread(R.data, 1, (size_t)W * H, stdin) != (size_t)W * H) return 3;
    using namespace HYSLAM;
    FeatureExtractorSettings s; s.nFeatures = 1000;
    int ndev = 0;
    if (hs_device_count(&ndev) != HS_OK || ndev == 0) {
        try { HipORBExtractor ex(std::make_shared<HipORBDistance>(), s); } catch (const std::exception& e) { printf("NO DEVICE: %s\n", e.what()); return 0; }
        return 4;      // must have thrown: there is no CPU fallback
    }
    auto dist = std::make_shared<HipORBDistance>();
    HipORBExtractor exL(dist, s), exR(dist, s);
    std::vector<cv::KeyPoint> kL, kR; std::vector<FeatureDescriptor> dL, dR;
    exL(L, cv::Mat(), kL, dL);
    exR(R, cv::Mat(), kR, dR);
    std::vector<cv::KeyPoint> none; std::vector<FeatureDescriptor> dnone;
    exL(cv::Mat(), cv::Mat(), none, dnone);
    if (!none.empty() || !dnone.empty()) return 5;
    FeatureMatcherSettings ms;
    HipStereomatcher sm(exL.handle(), kL, kR, dL, dR, 500.f, 60.f, (float)H, ms);
    sm.computeStereoMatches();
    std::vector<float> uR, depth; sm.getData(uR, depth);

    hso_orb_params p; hso_default_params(&p); p.nfeatures = 1000;
    const int cap = 1200;
    std::vector<hso_keypoint> okL(cap), okR(cap); std::vector<uint8_t> odL(cap * 32), odR(cap * 32);
    int nL = hso_orb_extract(&p, L.data, W, H, W, okL.data(), odL.data(), cap, nullptr);
    int nR = hso_orb_extract(&p, R.data, W, H, W, okR.data(), odR.data(), cap, nullptr);
    if (nL != (int)kL.size() || nR != (int)kR.size() || dL.size() != kL.size()) { printf("count mismatch %d %zu %d %zu\n", nL, kL.size(), nR, kR.size()); return 6; }
    for (int i = 0; i < nL; i++) {
        if (kL[i].pt.x != okL[i].x || kL[i].pt.y != okL[i].y || kL[i].angle != okL[i].angle || kL[i].octave != okL[i].octave ||
            kL[i].size != okL[i].size || kL[i].response != okL[i].response || kL[i].class_id != -1) { printf("keypoint %d differs\n", i); return 7; }
        cv::Mat row = dL[i].rawDescriptor();
        if (memcmp(row.ptr(0), &odL[i * 32], 32)) { printf("descriptor %d differs\n", i); return 8; }
    }
    if (dL[0].distance(dL[1]) != (float)hso_hamming256(&odL[0], &odL[32])) return 9;
    hso_stereo_params sp{ 500.f, 60.f, H, 100.f, 50.f, 31.f };
    std::vector<float> ouR(nL), odepth(nL);
    hso_stereo_match(okL.data(), odL.data(), nL, okR.data(), odR.data(), nR, &sp, ouR.data(), odepth.data(), nullptr, nullptr);
    int matches = 0;
    for (int i = 0; i < nL; i++) { if (uR[i] != ouR[i] || depth[i] != odepth[i]) { printf("stereo %d differs\n", i); return 10; } matches += depth[i] > 0; }
    std::vector<float> sf = exL.GetScaleFactors();
    if (exL.GetLevels() != 8 || sf.size() != 8 || sf[1] != 1.2f) return 11;
    // ---- HipStereoFrontend: the same pair as ONE ticket (extract L+R + stereo match), synchronously and with two tickets in flight
    {
        Camera cam; cam.sensor = 1;
        for (int i = 0; i < 9; i++) cam.K.at<float>(i / 3, i % 3) = 0.f;
        cam.K.at<float>(0, 0) = 500.f; cam.K.at<float>(1, 1) = 500.f; cam.K.at<float>(2, 2) = 1.f;
        cam.mbf = 60.f; cam.mnMaxX = (float)W; cam.mnMaxY = (float)H;
        HipStereoFrontend fe(dist, s, cam, ms);
        auto same = [&](const FeatureViews& v, const char* what) {
            if (v.numViews() != nL || (int)v.getKeysR().size() != nR || !v.isStereo()) { printf("%s: counts differ\n", what); return false; }
            const std::vector<cv::KeyPoint> kr = v.getKeysR(); const std::vector<FeatureDescriptor> dr = v.getDescriptorsR();
            for (int i = 0; i < nL; i++) {
                const cv::KeyPoint k = v.keypt(i);
                if (k.pt.x != okL[i].x || k.pt.y != okL[i].y || k.angle != okL[i].angle || k.octave != okL[i].octave || k.size != okL[i].size || k.response != okL[i].response ||
                    memcmp(v.descriptor(i).rawDescriptor().ptr(0), &odL[i * 32], 32) || v.uR(i) != ouR[i] || v.depth(i) != odepth[i]) { printf("%s: left view %d differs\n", what, i); return false; }
            }
            for (int i = 0; i < nR; i++)
                if (kr[i].pt.x != okR[i].x || kr[i].pt.y != okR[i].y || kr[i].angle != okR[i].angle || memcmp(dr[i].rawDescriptor().ptr(0), &odR[i * 32], 32)) { printf("%s: right view %d differs\n", what, i); return false; }
            return true;
        };
        if (!same(fe.process(L, R), "frontend process")) return 12;
        const int32_t t1 = fe.submit(L, R), t2 = fe.submit(L, R);            // two tickets in flight
        bool threw = false;
        try { fe.submit(L, R); } catch (const std::exception&) { threw = true; }
        if (!threw) { printf("a third ticket was accepted\n"); return 13; }
        if (!same(fe.collect(t2), "frontend ticket 2") || !same(fe.collect(t1), "frontend ticket 1")) return 14;
    }
    printf("ADAPTOR OK %d %d keypoints, %d stereo matches\n", nL, nR, matches);
    return 0;
}
