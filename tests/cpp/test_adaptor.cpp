// Exercises hyslam_amd/host/HipORBExtractor.h the way ImageProcessing::ProcessStereoImage uses an extractor and the
// stereo matcher (src/main/ImageProcessing.cpp:82-103), and compares with the CPU oracle (test infrastructure).
// usage: test_adaptor W H  < left.raw right.raw on stdin        prints "ADAPTOR OK ..." on success
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../../hyslam_amd/host/HipORBExtractor.h"
#include "../../hyslam_amd/host/HipFeatureMatcher.h"
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
    // ---- device-resident frames (include/hyslam_amd.h): the extractors published what they extracted; the reference-signature
    //      Stereomatcher and the projection matcher find the frames by their keypoints and run on the device copies
    {
        Camera cam; cam.sensor = 1;
        for (int i = 0; i < 9; i++) cam.K.at<float>(i / 3, i % 3) = 0.f;
        cam.K.at<float>(0, 0) = 500.f; cam.K.at<float>(1, 1) = 500.f; cam.K.at<float>(0, 2) = W / 2.f; cam.K.at<float>(1, 2) = H / 2.f; cam.K.at<float>(2, 2) = 1.f;
        cam.mbf = 60.f; cam.mnMaxX = (float)W; cam.mnMaxY = (float)H;
        exL(L, cv::Mat(), kL, dL = std::vector<FeatureDescriptor>());       // (the empty-image call above was the handle's last extraction)
        exR(R, cv::Mat(), kR, dR = std::vector<FeatureDescriptor>());
        const hs_frame_token tokL = exL.lastFrameToken(), tokR = exR.lastFrameToken();
        int32_t n_info = 0;
        if (!tokL || !tokR || hs_frame_info(0, tokL, &n_info) != HS_OK || n_info != nL) { printf("extractors did not publish their frames\n"); return 15; }
        FeatureViews views(kL, kR, dL, dR, FeatureExtractorSettings());
        HipStereomatcher sm2(views, cam, ms);
        sm2.computeStereoMatches();
        std::vector<float> uR2, depth2; sm2.getData(uR2, depth2);
        if (!sm2.frames_on_device) { printf("Stereomatcher(views, cam, settings) did not find the frames on the device\n"); return 16; }
        for (int i = 0; i < nL; i++) if (uR2[i] != ouR[i] || depth2[i] != odepth[i]) { printf("stereo on device frames: %d differs\n", i); return 17; }
        sm2.getData(views);
        // a frame is recognised by ALL its keypoints: one changed field -> unknown
        std::vector<hs_keypoint> probe(nL);
        for (int i = 0; i < nL; i++) probe[i] = hs_keypoint{ kL[i].pt.x, kL[i].pt.y, kL[i].size, kL[i].angle, kL[i].response, kL[i].octave };
        hs_frame_token found = 0;
        if (hs_frame_find(0, probe.data(), nL, &found) != HS_OK || found != tokL) { printf("hs_frame_find missed the left frame\n"); return 18; }
        const float angle_saved = probe[nL / 2].angle;
        probe[nL / 2].angle += 1.f;
        if (hs_frame_find(0, probe.data(), nL, &found) == HS_OK || found != 0) { printf("hs_frame_find accepted a changed frame\n"); return 19; }
        // TrackLocalMap-like search on the left frame: once from the device copy, once (token released) from the host objects — same associations
        cv::Mat Tcw(4, 4, CV_32F);
        for (int r = 0; r < 4; r++) for (int c = 0; c < 4; c++) Tcw.at<float>(r, c) = r == c ? 1.f : 0.f;
        Tcw.at<float>(0, 3) = 0.01f; Tcw.at<float>(2, 3) = 0.02f;
        std::vector<MapPoint*> lms;
        for (int j = 0; j < 3 * nL; j++) {
            const int i = j % nL; const cv::KeyPoint k = views.keypt(i);
            const double d = (views.depth(i) > 0 ? views.depth(i) : 4.0 + (j % 17)) * (1.0 + 0.01 * (j % 5 - 2));
            const double pc[3] = { (k.pt.x + (j % 3 - 1) - cam.cx()) * d / 500.0 - 0.01, (k.pt.y - cam.cy()) * d / 500.0, d - 0.02 };
            const double dd = std::sqrt(pc[0] * pc[0] + pc[1] * pc[1] + pc[2] * pc[2]);
            MapPoint* m = new MapPoint();
            for (int c = 0; c < 3; c++) { m->mWorldPos.at<float>(c) = (float)pc[c]; m->mNormalVector.at<float>(c) = (float)(pc[c] / dd); }
            m->size = (float)(k.size * d / 500.0); m->mfMinDistance = (float)(dd * 0.5); m->mfMaxDistance = (float)(dd * 2.0); m->nObs = 2;
            cv::Mat row = views.descriptor(i).rawDescriptor();
            if (j >= nL) row.ptr(0)[j % 32] ^= (uint8_t)(1u << (j % 8));
            m->mDescriptor = FeatureDescriptor(row, dist);
            lms.push_back(m);
        }
        FeatureMatcherSettings fms; fms.nnratio = 0.8f;
        HipMatcherCore core(fms, exL.handle());
        Frame F1(views, cam); F1.SetPose(Tcw);
        const int n1 = core.SearchByProjection(F1, lms, 5.f);
        if (!core.frame_on_device) { printf("SearchByProjection did not find the frame on the device\n"); return 20; }
        if (hs_frame_release(0, tokL) != HS_OK || hs_frame_release(0, tokL) == HS_OK) { printf("hs_frame_release\n"); return 21; }
        probe[nL / 2].angle = angle_saved;      // (the same frame was published several times above — the front end's tickets, the first extraction: every copy goes)
        for (int guard = 0; guard < 32 && hs_frame_find(0, probe.data(), nL, &found) == HS_OK; guard++) hs_frame_release(0, found);
        Frame F2(views, cam); F2.SetPose(Tcw);
        const int n2 = core.SearchByProjection(F2, lms, 5.f);
        if (core.frame_on_device) { printf("a released frame was still found\n"); return 22; }
        if (n1 != n2 || n1 < nL / 4 || F1.getLandMarkMatches().views_to_landmarks != F2.getLandMarkMatches().views_to_landmarks ||
            F1.getLandMarkMatches().n_matches != F2.getLandMarkMatches().n_matches) { printf("device frame vs host frame: %d vs %d matches, associations differ\n", n1, n2); return 23; }
        // slot reuse: 16 more publishes push the right frame out; its token is unknown afterwards and the stereo matcher falls back to the host path
        for (int r = 0; r < 17; r++) { std::vector<cv::KeyPoint> kk; std::vector<FeatureDescriptor> ddd; exL(L, cv::Mat(), kk, ddd); }
        if (hs_frame_info(0, tokR, &n_info) == HS_OK) { printf("a slot survived 17 publishes\n"); return 24; }
        HipStereomatcher sm3(views, cam, ms);      // left: found again (just re-published), right: gone -> host path
        sm3.computeStereoMatches();
        std::vector<float> uR3, depth3; sm3.getData(uR3, depth3);
        if (sm3.frames_on_device) { printf("stereo matcher used a frame that is not cached\n"); return 25; }
        for (int i = 0; i < nL; i++) if (uR3[i] != ouR[i] || depth3[i] != odepth[i]) { printf("stereo host fallback: %d differs\n", i); return 26; }
        printf("device frames: stereo on cached frames ok, projection search %d matches (device == host), slot reuse ok\n", n1);
    }
    // ---- HipStereoFrontend::processCamera: the pair as BGR frames of twice the size at the camera scale 0.5 is the grey pair again (see below): the views must
    //      equal the oracle's for (L, R)
    {
        Camera cam2; cam2.sensor = 1;
        for (int i = 0; i < 9; i++) cam2.K.at<float>(i / 3, i % 3) = 0.f;
        cam2.K.at<float>(0, 0) = 500.f; cam2.K.at<float>(1, 1) = 500.f; cam2.K.at<float>(2, 2) = 1.f;
        cam2.mbf = 60.f; cam2.mnMinX = 0; cam2.mnMaxX = (float)W; cam2.mnMinY = 0; cam2.mnMaxY = (float)H;
        HipStereoFrontend fe(dist, s, cam2, ms);
        cv::Mat bigL(2 * H, 2 * W, CV_8UC3), bigR(2 * H, 2 * W, CV_8UC3);
        for (int y = 0; y < 2 * H; y++) for (int x = 0; x < 2 * W; x++) for (int k = 0; k < 3; k++) { bigL.ptr(y)[3 * x + k] = L.ptr(y / 2)[x / 2]; bigR.ptr(y)[3 * x + k] = R.ptr(y / 2)[x / 2]; }
        FeatureViews v = fe.processCamera(bigL, bigR, false, 0.5f);
        if (v.numViews() != nL) { printf("processCamera: %d views, expected %d\n", v.numViews(), nL); return 34; }
        for (int i = 0; i < nL; i++)
            if (v.keypt(i).pt.x != okL[i].x || v.keypt(i).pt.y != okL[i].y || v.uR(i) != ouR[i] || v.depth(i) != odepth[i] || memcmp(v.descriptor(i).rawDescriptor().ptr(0), &odL[i * 32], 32)) {
                printf("processCamera: view %d differs\n", i); return 35; }
        printf("processCamera: BGR pair 2x at scale 0.5 == the oracle's stereo views (%d)\n", nL);
    }
    // ---- extractFromCamera (ImageProcessing::PreProcessImg on the device): the left frame as a BGR frame of twice the size (every pixel a 2x2 block of
    //      equal channels) at the camera scale 0.5 is the grey frame again — the rounded 2x2 mean of four equal bytes and 4899 + 9617 + 1868 = 2^14 —, so
    //      features and the grey frame handed back must equal the oracle's for L; then against the oracle's PreProcessImg on a frame with three different channels
    {
        cv::Mat big(2 * H, 2 * W, CV_8UC3);
        for (int y = 0; y < 2 * H; y++) for (int x = 0; x < 2 * W; x++) for (int k = 0; k < 3; k++) big.ptr(y)[3 * x + k] = L.ptr(y / 2)[x / 2];
        std::vector<cv::KeyPoint> kc; std::vector<FeatureDescriptor> dc; cv::Mat grey;
        exL.extractFromCamera(big, false, 0.5f, &grey, kc, dc);
        if (grey.rows != H || grey.cols != W || memcmp(grey.ptr(0), L.ptr(0), (size_t)W * H)) { printf("extractFromCamera: the grey frame differs\n"); return 27; }
        if ((int)kc.size() != nL) { printf("extractFromCamera: %zu keypoints, expected %d\n", kc.size(), nL); return 28; }
        for (int i = 0; i < nL; i++) {
            cv::Mat row = dc[i].rawDescriptor();
            if (kc[i].pt.x != okL[i].x || kc[i].pt.y != okL[i].y || kc[i].angle != okL[i].angle || memcmp(row.ptr(0), &odL[i * 32], 32)) { printf("extractFromCamera: feature %d differs\n", i); return 29; }
        }
        cv::Mat col(H, W, CV_8UC3);
        for (int y = 0; y < H; y++) for (int x = 0; x < W; x++) { col.ptr(y)[3 * x] = L.ptr(y)[x]; col.ptr(y)[3 * x + 1] = R.ptr(y)[x]; col.ptr(y)[3 * x + 2] = (uint8_t)(255 - L.ptr(y)[x]); }
        std::vector<uint8_t> og((size_t)W * H);
        if (hso_preprocess(col.ptr(0), W, H, (int)col.step, 3, 1, 1.0f, og.data(), W) != 0) return 30;
        dc.clear();                                              // (descriptors are APPENDED, like the reference: ORBExtractor.cpp:558-561)
        exL.extractFromCamera(col, true, 1.0f, &grey, kc, dc);
        if (memcmp(grey.ptr(0), og.data(), (size_t)W * H)) { printf("extractFromCamera: RGB -> grey differs from the oracle\n"); return 31; }
        std::vector<hso_keypoint> okc(cap); std::vector<uint8_t> odc(cap * 32);
        const int nc = hso_orb_extract(&p, og.data(), W, H, W, okc.data(), odc.data(), cap, nullptr);
        if (nc != (int)kc.size()) { printf("extractFromCamera: %zu keypoints on the RGB frame, oracle %d\n", kc.size(), nc); return 32; }
        for (int i = 0; i < nc; i++) if (kc[i].pt.x != okc[i].x || kc[i].pt.y != okc[i].y || memcmp(dc[i].rawDescriptor().ptr(0), &odc[i * 32], 32)) { printf("extractFromCamera: RGB feature %d differs\n", i); return 33; }
        printf("extractFromCamera: BGR 2x at scale 0.5 == the grey frame (%d keypoints), RGB at scale 1.0 == oracle (%d keypoints)\n", nL, nc);
    }
    printf("ADAPTOR OK %d %d keypoints, %d stereo matches\n", nL, nR, matches);
    return 0;
}
