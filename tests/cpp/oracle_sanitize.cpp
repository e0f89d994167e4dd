// Runs the CPU oracle (test infrastructure) under AddressSanitizer + UndefinedBehaviorSanitizer on a small synthetic workload: extraction
// (all stages), stereo matching, frame grid, projection search, brute-force 2-NN, Sim3 matchers.  GPU sanitizers are not available on the pool,
// so the CPU restatement — which every GPU parity test trusts — is the part that can be checked.
// build: g++ -O1 -g -fsanitize=address,undefined -fno-sanitize-recover=all oracle_sanitize.cpp ../../oracle/hs_oracle.cpp ../../oracle/hs_oracle_match.cpp -pthread
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "../../oracle/hs_oracle.h"

static uint64_t s = 0x9E3779B97F4A7C15ull;
static uint32_t rnd() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return (uint32_t)(s >> 32); }

int main()
{
    const int W = 320, H = 240;
    std::vector<uint8_t> L(W * H), R(W * H);
    for (int y = 0; y < H; y++) for (int x = 0; x < W; x++) L[y * W + x] = (uint8_t)(60 + (x * 90) / W + ((x / 23 + y / 17) & 1) * 70 + rnd() % 5);
    for (int i = 0; i < 150; i++) { int x0 = rnd() % (W - 20), y0 = rnd() % (H - 20), w = 6 + rnd() % 30, h = 6 + rnd() % 30, g = rnd() % 256; for (int y = y0; y < y0 + h && y < H; y++) for (int x = x0; x < x0 + w && x < W; x++) L[y * W + x] = (uint8_t)g; }
    for (int y = 0; y < H; y++) for (int x = 0; x < W; x++) R[y * W + x] = L[y * W + (x + 6 < W ? x + 6 : W - 1)];
    hso_orb_params p; hso_default_params(&p); p.nfeatures = 400;
    const int cap = 600;
    std::vector<hso_keypoint> kL(cap), kR(cap); std::vector<uint8_t> dL(cap * 32), dR(cap * 32);
    const int nL = hso_orb_extract(&p, L.data(), W, H, W, kL.data(), dL.data(), cap, nullptr);
    const int nR = hso_orb_extract(&p, R.data(), W, H, W, kR.data(), dR.data(), cap, nullptr);
    if (nL < 50 || nR < 50) { printf("too few keypoints %d %d\n", nL, nR); return 1; }
    hso_stereo_params sp{ 300.f, 36.f, H, 100.f, 50.f, 31.f };
    std::vector<float> uR(nL), depth(nL);
    hso_stereo_match(kL.data(), dL.data(), nL, kR.data(), dR.data(), nR, &sp, uR.data(), depth.data(), nullptr, nullptr);
    // a frame view + landmarks back-projected from its own keypoints
    hso_frame_view F{}; F.Rcw[0] = F.Rcw[4] = F.Rcw[8] = 1.f; F.fx = F.fy = 300.f; F.cx = 159.5f; F.cy = 119.5f; F.mbf = 36.f; F.sensor = 1;
    F.max_x = (float)W; F.max_y = (float)H; F.size_ref = 31.f; F.n = nL; F.kps = kL.data(); F.desc = dL.data(); F.uR = uR.data();
    std::vector<int32_t> obs(nL, -1); F.kp_lm_obs = obs.data();
    std::vector<hso_landmark> lms(nL * 2);
    for (int i = 0; i < nL * 2; i++) {
        const int k = i % nL; hso_landmark& m = lms[i]; memset(&m, 0, sizeof(m));
        const float d = 4.f + (rnd() % 100) * 0.1f;
        m.pos[0] = (kL[k].x - F.cx) * d / F.fx; m.pos[1] = (kL[k].y - F.cy) * d / F.fy; m.pos[2] = d;
        m.size = kL[k].size * d / F.fx; m.min_dist = d * 0.5f; m.max_dist = d * 2.f; m.normal[2] = 1.f; m.assoc_kp = -1; m.prev_angle = kL[k].angle;
        memcpy(m.desc, &dL[k * 32], 32); m.skip = (i % 37) == 0;
    }
    hso_proj_params pp{}; pp.th = 5.f; pp.score_threshold = 100.f; pp.second_best_ratio = 0.8f; pp.frac_smaller = 0.5f; pp.frac_larger = 1.5f;
    pp.use_distance = 1; pp.use_stereo = 1; pp.use_prev_matched = 1; pp.check_rotation = 1; pp.max_view_angle = 1.047f; pp.reproj_threshold = 5.99f; pp.sigma_ref = 1.f;
    std::vector<int32_t> mi(lms.size()); std::vector<float> md(lms.size());
    const int nm = hso_search_by_projection(&F, lms.data(), (int)lms.size(), &pp, mi.data(), md.data());
    std::vector<int32_t> cell(2 * nL); hso_frame_grid(&F, cell.data());
    std::vector<int32_t> bi(nL), bd(nL), sd(nL); hso_hamming_knn2(dL.data(), nL, dR.data(), nR, bi.data(), bd.data(), sd.data());
    float Scw[16] = { 1.05f, 0, 0, 0.01f, 0, 1.05f, 0, 0, 0, 0, 1.05f, 0.02f, 0, 0, 0, 1 };
    std::vector<uint8_t> taken(nL, 0); std::vector<int32_t> m3(lms.size());
    for (auto& m : lms) { m.min_dist *= 0.8f; m.max_dist *= 1.2f; }
    const int n3 = hso_search_by_projection_sim3(&F, Scw, lms.data(), (int)lms.size(), 4, 50.f, taken.data(), m3.data());
    float R12[9] = { 1, 0, 0, 0, 1, 0, 0, 0, 1 }, t12[3] = { 0.01f, 0, 0 };
    std::vector<int32_t> m12(nL);
    const int n4 = hso_search_by_sim3(&F, lms.data(), &F, lms.data(), 1.0f, R12, t12, 7.5f, 100.f, m12.data());
    printf("ORACLE SANITIZE OK %d %d keypoints, %d projection, %d sim3-projection, %d sim3 matches\n", nL, nR, nm, n3, n4);
    return 0;
}
