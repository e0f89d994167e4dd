// host_sanitize.cpp — drives the PRODUCT's host code (everything in hyslam_amd/csrc that does not run on the GPU) under AddressSanitizer +
// UndefinedBehaviorSanitizer, CPU only: the library's translation units compiled host-only and linked against hip_stub.cpp (device memory =
// host memory, launches = no-ops).  Covered: hs_orb_create's tables, configure_impl's geometry for a random sweep of frame sizes / scale factors /
// level counts / cell sizes / feature counts (the sweep of tests/test_gpu_parity.py plus extremes): pyramid fusion, chain and deep-chain planners,
// FAST work items (wide and narrow), quadtree key tables, workspace sizing and uploads; the host-pointer entry points' staging (extract, batch,
// stereo, the matchers' CSR / scratch handling), the ingest ticket state machine (submit / wait, errors, both slots busy), and the vocabulary
// loaders on valid, truncated and randomly corrupted text / binary files (hs_vocab_last_error instead of stderr).
// usage: host_sanitize [seed] [geometries] [vocab_cases]      prints "HOST SANITIZE OK ..." (any sanitizer report aborts with a non-zero status)
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <string>
#include <vector>
#include <thread>
#include <atomic>
#include <unistd.h>
#include "../../../include/hyslam_amd.h"
void hs_orb_borrow(hs_orb* h, int delta);      // hs_api.hip (internal: what hs_comm_create / hs_comm_destroy call)

extern "C" long hip_stub_launches();
void hs_debug_plan_summary(const hs_orb* h, int32_t* out /*[8]*/);        // hs_api.hip: launches of the pyramid's two plans, item counts (host-side facts of the last configuration)

#define CHECK(c) do { if (!(c)) { printf("FAILED %s:%d: %s\n", __FILE__, __LINE__, #c); return 1; } } while (0)

static std::mt19937_64 rng;
static int rnd(int lo, int hi) { return lo + (int)(rng() % (uint64_t)(hi - lo + 1)); }

static int geometry_case(int w, int h, int nfeat, float scale, int levels, int cell, int batch, bool exercise)
{
    hs_orb_params p; hs_orb_default_params(&p);
    p.nfeatures = nfeat; p.scale_factor = scale; p.nlevels = levels; p.cell_px = cell;
    hs_orb* ex = nullptr;
    int st = hs_orb_create(&p, 0, &ex);
    if (st != HS_OK) return 0;                                   // rejected parameter combinations (quota > LDS list, ...) are fine: no crash is the point
    st = hs_orb_reserve(ex, w, h, batch);
    int ok = 0;
    if (st != HS_OK && std::strstr(hs_orb_last_error(ex), "internal:")) { printf("reserve %dx%d: %s\n", w, h, hs_orb_last_error(ex)); return -1; }   // a refusal is fine, a failed self-check is not
    if (st == HS_OK) {
        ok = 1;
        int32_t plan[8]; hs_debug_plan_summary(ex, plan);
        if (plan[0] <= 0 && levels > 1) { printf("no pyramid launches for %dx%d levels %d\n", w, h, levels); return -1; }
        const int cap = hs_orb_max_keypoints(ex);
        if (exercise && cap > 0) {
            std::vector<uint8_t> img((size_t)w * h);
            for (auto& v : img) v = (uint8_t)rng();
            std::vector<hs_keypoint> k((size_t)batch * cap); std::vector<uint8_t> d((size_t)batch * cap * 32); std::vector<int32_t> n(batch, -1);
            std::vector<const uint8_t*> ptrs(batch, img.data());
            st = hs_orb_extract_batch(ex, ptrs.data(), batch, w, h, w, k.data(), d.data(), cap, n.data());
            if (st != HS_OK) { printf("extract_batch %dx%d: %s\n", w, h, hs_orb_last_error(ex)); return -1; }
            {   // the camera-frame entry point (PreProcessImg on the device): a 3-channel frame of TWICE the size at scale 0.5 lands on the same level-0 geometry,
                // a 4-channel frame at scale 1 too; the staging of the raw frames (row padding to 4 bytes, odd strides) and the grey read-back run under the sanitizers
                hs_preprocess_params pp{ 3, 1, 0.5f, 0 };
                int32_t ow = 0, oh = 0; hs_preprocess_size(2 * w, 2 * h, pp.scale, &ow, &oh);
                if (ow != w || oh != h) { printf("hs_preprocess_size(%d, %d, 0.5) = %d x %d\n", 2 * w, 2 * h, ow, oh); return -1; }
                std::vector<uint8_t> col((size_t)2 * w * 3 * 2 * h + 16), grey((size_t)batch * w * h);
                for (auto& v : col) v = (uint8_t)rng();
                std::vector<const uint8_t*> cptrs(batch, col.data());
                st = hs_orb_extract_camera_batch(ex, cptrs.data(), batch, 2 * w, 2 * h, (size_t)2 * w * 3, &pp, k.data(), d.data(), cap, n.data(), grey.data());
                if (st != HS_OK) { printf("extract_camera_batch %dx%d: %s\n", w, h, hs_orb_last_error(ex)); return -1; }
                hs_preprocess_params p4{ 4, 0, 1.0f, 0 };
                std::vector<uint8_t> c4((size_t)(w * 4 + 3) * h);
                std::vector<const uint8_t*> c4p(batch, c4.data());
                st = hs_orb_extract_camera_batch(ex, c4p.data(), batch, w, h, (size_t)w * 4 + 3, &p4, k.data(), d.data(), cap, n.data(), nullptr);
                if (st != HS_OK) { printf("extract_camera_batch (4 channels) %dx%d: %s\n", w, h, hs_orb_last_error(ex)); return -1; }
                hs_preprocess_params bad{ 2, 0, 1.0f, 0 };
                if (hs_orb_extract_camera_batch(ex, c4p.data(), batch, w, h, (size_t)w * 4 + 3, &bad, k.data(), d.data(), cap, n.data(), nullptr) == HS_OK) { printf("two channels were accepted\n"); return -1; }
            }
            // the ingest tickets: two in flight, a third is refused, waits in both orders, a wait with too small a capacity keeps the ticket
            hs_stereo_params sp{ 500.f, 60.f, h, 100.f, 50.f, 31.f };
            if (batch % 2 == 0) {
                int32_t t1 = 0, t2 = 0, t3 = 0;
                if (hs_orb_submit_batch(ex, ptrs.data(), batch, w, h, w, &sp, &t1) != HS_OK) { printf("submit: %s\n", hs_orb_last_error(ex)); return -1; }
                if (hs_orb_submit_batch(ex, ptrs.data(), batch, w, h, w, &sp, &t2) != HS_OK) return -1;
                if (hs_orb_submit_batch(ex, ptrs.data(), batch, w, h, w, &sp, &t3) == HS_OK) { printf("a third ticket was accepted\n"); return -1; }
                std::vector<float> ur((size_t)batch / 2 * cap), dz((size_t)batch / 2 * cap);
                if (hs_orb_wait(ex, t2, k.data(), d.data(), n.data(), 1, ur.data(), dz.data()) == HS_OK) { printf("a wait with cap 1 passed\n"); return -1; }      // (the ticket stays valid)
                if (hs_orb_wait(ex, t2, k.data(), d.data(), n.data(), cap, ur.data(), dz.data()) != HS_OK) { printf("wait t2: %s\n", hs_orb_last_error(ex)); return -1; }
                if (hs_orb_wait(ex, t1, k.data(), d.data(), n.data(), cap, ur.data(), dz.data()) != HS_OK) return -1;
                if (hs_orb_wait(ex, t1, k.data(), d.data(), n.data(), cap, ur.data(), dz.data()) == HS_OK) { printf("a ticket was waited for twice\n"); return -1; }
            }
            // stereo matcher + 2-NN on host arrays (staging sizes follow nL / nR)
            const int nL = rnd(0, 300), nR = rnd(0, 300);
            std::vector<hs_keypoint> kl(std::max(nL, 1)), kr(std::max(nR, 1)); std::vector<uint8_t> dl((size_t)std::max(nL, 1) * 32), dr((size_t)std::max(nR, 1) * 32);
            for (auto& q : kl) { q.x = (float)rnd(0, w); q.y = (float)rnd(-5, h + 5); q.size = 31.f; q.octave = rnd(0, 7); }
            for (auto& q : kr) { q.x = (float)rnd(0, w); q.y = (float)rnd(-5, h + 5); q.size = 31.f; q.octave = rnd(0, 7); }
            std::vector<float> ur(std::max(nL, 1)), dz(std::max(nL, 1));
            if (hs_stereo_match(ex, kl.data(), dl.data(), nL, kr.data(), dr.data(), nR, &sp, ur.data(), dz.data()) != HS_OK) { printf("stereo: %s\n", hs_orb_last_error(ex)); return -1; }
            std::vector<int32_t> bi(std::max(nL, 1)), bd(std::max(nL, 1)), sd(std::max(nL, 1));
            if (nL > 0 && nR > 0 && hs_hamming_knn2(ex, dl.data(), nL, dr.data(), nR, bi.data(), bd.data(), sd.data()) != HS_OK) return -1;
        }
    }
    hs_orb_destroy(ex);
    return ok;
}

// a small valid vocabulary (k-ary tree of `levels` levels), saved as text and binary through the library's own writer
static hs_vocab* make_vocab(int k, int levels, std::vector<int32_t>& cb, std::vector<int32_t>& cc, std::vector<uint8_t>& desc, std::vector<int32_t>& word, std::vector<float>& weight)
{
    int n = 1, width = 1;
    for (int l = 1; l <= levels; l++) { width *= k; n += width; }
    cb.assign(n, 0); cc.assign(n, 0); desc.resize((size_t)n * 32); word.assign(n, -1); weight.assign(n, 0.f);
    for (auto& v : desc) v = (uint8_t)rng();
    int next = 1, words = 0;
    std::vector<int> level(n, 0);
    for (int i = 0; i < n; i++) {
        if (level[i] < levels && next + k <= n) { cb[i] = next; cc[i] = k; for (int c = 0; c < k; c++) level[next + c] = level[i] + 1; next += k; }
        else { word[i] = words++; weight[i] = 0.5f + (float)(rng() % 100) / 100.f; }
    }
    hs_vocab_tree T{ n, levels, cb.data(), cc.data(), desc.data(), word.data(), weight.data(), nullptr };
    hs_vocab* v = nullptr;
    return hs_vocab_from_tree(&T, k, &v) == HS_OK ? v : nullptr;
}

static std::vector<uint8_t> slurp(const std::string& p) { std::vector<uint8_t> b; FILE* f = fopen(p.c_str(), "rb"); if (!f) return b; int c; while ((c = fgetc(f)) != EOF) b.push_back((uint8_t)c); fclose(f); return b; }
static void spit(const std::string& p, const std::vector<uint8_t>& b) { FILE* f = fopen(p.c_str(), "wb"); if (f) { if (!b.empty()) fwrite(b.data(), 1, b.size(), f); fclose(f); } }

int main(int argc, char** argv)
{
    if (argc > 3 && !strcmp(argv[1], "plan")) {                 // host_sanitize plan W H [nfeat scale levels]: the host-side plan of a geometry
        hs_orb_params p; hs_orb_default_params(&p);
        p.nfeatures = argc > 4 ? atoi(argv[4]) : 2000; if (argc > 5) p.scale_factor = (float)atof(argv[5]); if (argc > 6) p.nlevels = atoi(argv[6]);
        hs_orb* ex = nullptr;
        CHECK(hs_orb_create(&p, 0, &ex) == HS_OK && hs_orb_reserve(ex, atoi(argv[2]), atoi(argv[3]), 2) == HS_OK);
        int32_t s[8]; hs_debug_plan_summary(ex, s);
        printf("pyramid launches: standard plan %d, small-batch plan %d (longest chain %d levels, %d B of LDS, %d workgroups per frame); FAST items per frame: %d wide, %d narrow; levels with quadtree keys: %d\n",
               s[0], s[1], s[5], s[6], s[7], s[2], s[3], s[4]);
        hs_orb_destroy(ex);
        return 0;
    }
    const uint64_t seed = argc > 1 ? strtoull(argv[1], nullptr, 10) : 1;
    const int n_geo = argc > 2 ? atoi(argv[2]) : 120, n_voc = argc > 3 ? atoi(argv[3]) : 400;
    rng.seed(seed);
    int ndev = 0;
    CHECK(hs_device_count(&ndev) == HS_OK && ndev == 1);

    // ---- geometry: the fixed shapes of the GPU suites, then the random sweep
    int accepted = 0, tried = 0;
    const int fixed[][7] = { {640, 480, 1000, 120, 8, 30, 2}, {1920, 1080, 2000, 120, 8, 30, 2}, {1920, 1080, 2000, 120, 8, 30, 32}, {4000, 3000, 3000, 140, 8, 30, 1},
                             {643, 481, 700, 120, 8, 30, 1}, {3840, 2160, 2000, 120, 8, 30, 1}, {2560, 40, 500, 120, 3, 30, 1}, {64, 64, 100, 120, 8, 30, 1}, {33, 33, 50, 200, 12, 12, 4},
                             {1200, 300, 1500, 120, 6, 30, 1}, {1900, 200, 1200, 120, 3, 30, 1}, {16384, 64, 100, 110, 2, 62, 1}, {1920, 1080, 9000, 120, 8, 30, 1}, {800, 600, 1500, 140, 8, 62, 2} };
    for (const auto& f : fixed) {
        const int r = geometry_case(f[0], f[1], f[2], f[3] / 100.f, f[4], f[5], f[6], (size_t)f[0] * f[1] * f[6] < 6000000);
        CHECK(r >= 0); accepted += r; tried++;
    }
    for (int i = 0; i < n_geo; i++) {
        // the tuning knobs that are PARSED (read once per handle): explicit pyramid plans incl. malformed ones, chain / deep-plan switches
        static const char* const plans[] = { "3,4", "2,5", "1,2,3,1", "7", "9,9", "1,1,1,1,1,1,1", "", "a,b", "2,,3", "0,0,0", "2,2,2,2,2,2,2,2,2,2", "-3,4" };
        if (i % 4 == 1) setenv("HS_PYRAMID_PLAN", plans[rnd(0, (int)(sizeof(plans) / sizeof(plans[0])) - 1)], 1); else unsetenv("HS_PYRAMID_PLAN");
        if (i % 7 == 2) setenv("HS_PYRAMID_CHAIN", rnd(0, 1) ? "2" : "0", 1); else unsetenv("HS_PYRAMID_CHAIN");
        if (i % 5 == 3) setenv("HS_PYRAMID_DEEP_MAX", rnd(0, 1) ? "0" : "100000", 1); else unsetenv("HS_PYRAMID_DEEP_MAX");
        const int w = rnd(1, 10) == 1 ? rnd(20, 120) : rnd(64, 2600), h = rnd(1, 10) == 1 ? rnd(20, 120) : rnd(64, 1600);
        const int r = geometry_case(w, h, rnd(20, 4000), 1.1f + 0.01f * (float)rnd(0, 90), rnd(1, 12), rnd(1, 8) == 1 ? rnd(8, 70) : 30, rnd(1, 5), (size_t)w * h < 1500000);
        CHECK(r >= 0); accepted += r; tried++;
    }
    unsetenv("HS_PYRAMID_PLAN"); unsetenv("HS_PYRAMID_CHAIN"); unsetenv("HS_PYRAMID_DEEP_MAX");
    CHECK(accepted > tried / 2);

    // ---- a handle outliving its communicators: hs_orb_destroy and the last borrower's release race on two threads; exactly one of them frees
    // the handle (ASan: a double free or a leak fails the run), in either order and with several borrowers
    for (int i = 0; i < 200; i++) {
        hs_orb_params p; hs_orb_default_params(&p);
        hs_orb* ex = nullptr;
        CHECK(hs_orb_create(&p, 0, &ex) == HS_OK);
        const int nb = 1 + i % 3;
        for (int b = 0; b < nb; b++) hs_orb_borrow(ex, +1);
        CHECK(hs_orb_borrowers(ex) == nb);
        std::atomic<int> go{0};
        std::thread td([&] { while (!go.load()) {} hs_orb_destroy(ex); });
        std::thread tb([&] { while (!go.load()) {} for (int b = 0; b < nb; b++) hs_orb_borrow(ex, -1); });
        go.store(1);
        td.join(); tb.join();
    }

    // ---- the frame cache (hs_frame_*): publish after an extraction, find by keypoints, the *_frame(s) entry points, release, slot reuse after 16
    //      publishes, refusals (no extraction to publish, unknown tokens, a count that disagrees); then publishers, finders and consumers on four threads
    {
        hs_orb_params p; hs_orb_default_params(&p); p.nfeatures = 300;
        hs_orb *ex = nullptr, *mh = nullptr;
        CHECK(hs_orb_create(&p, 0, &ex) == HS_OK && hs_orb_create(&p, 0, &mh) == HS_OK);
        const int w = 320, h = 240;
        std::vector<uint8_t> img((size_t)w * h);
        for (auto& v : img) v = (uint8_t)rng();
        CHECK(hs_orb_reserve(ex, w, h, 1) == HS_OK);
        const int cap = hs_orb_max_keypoints(ex);
        std::vector<hs_keypoint> k(cap); std::vector<uint8_t> d((size_t)cap * 32); int32_t n = 0;
        hs_frame_token tok = 0;
        CHECK(hs_frame_publish(ex, 0, k.data(), 10, &tok) != HS_OK && tok == 0);                 // nothing extracted yet
        CHECK(hs_orb_extract(ex, img.data(), w, h, w, k.data(), d.data(), cap, &n) == HS_OK);
        // (the stub runs no kernels: fabricate a result of 40 keypoints to publish)
        const int nk = 40;
        for (int i = 0; i < nk; i++) k[i] = hs_keypoint{ 20.f + i, 30.f + 2 * i, 31.f, (float)i, 50.f, 0 };
        CHECK(hs_frame_publish(ex, 1, k.data(), nk, &tok) != HS_OK);                              // image index beyond the call's batch
        CHECK(hs_frame_publish(ex, 0, k.data(), cap + 1, &tok) != HS_OK);                         // more keypoints than the call could have produced
        CHECK(hs_frame_publish(ex, 0, k.data(), nk, &tok) == HS_OK && tok != 0);
        int32_t ninfo = 0; hs_frame_token found = 0;
        CHECK(hs_frame_info(0, tok, &ninfo) == HS_OK && ninfo == nk);
        CHECK(hs_frame_find(0, k.data(), nk, &found) == HS_OK && found == tok);
        CHECK(hs_frame_find(0, k.data(), nk - 1, &found) != HS_OK && found == 0);
        CHECK(hs_frame_find(1, k.data(), nk, &found) != HS_OK);                                    // another device's cache
        { std::vector<hs_keypoint> k2(k.begin(), k.begin() + nk); k2[7].response += 1.f; CHECK(hs_frame_find(0, k2.data(), nk, &found) != HS_OK); }
        hs_frame_view F; memset(&F, 0, sizeof(F));
        F.fx = F.fy = 300.f; F.cx = 160.f; F.cy = 120.f; F.max_x = (float)w; F.max_y = (float)h; F.size_ref = 31.f; F.n = nk; F.Rcw[0] = F.Rcw[4] = F.Rcw[8] = 1.f;
        std::vector<hs_landmark> lm(25); memset(lm.data(), 0, lm.size() * sizeof(hs_landmark));
        for (size_t i = 0; i < lm.size(); i++) { lm[i].pos[2] = 5.f; lm[i].size = 0.5f; lm[i].max_dist = 100.f; lm[i].assoc_kp = -1; }
        hs_proj_params pp; memset(&pp, 0, sizeof(pp)); pp.th = 3.f; pp.score_threshold = 100.f; pp.second_best_ratio = 0.8f; pp.frac_smaller = 0.5f; pp.frac_larger = 1.5f;
        std::vector<int32_t> mi(lm.size()); std::vector<float> md(lm.size()); int32_t nm = -1;
        CHECK(hs_search_by_projection_frame(mh, tok, &F, lm.data(), (int)lm.size(), &pp, mi.data(), md.data(), &nm) == HS_OK);
        F.n = nk - 1;
        CHECK(hs_search_by_projection_frame(mh, tok, &F, lm.data(), (int)lm.size(), &pp, mi.data(), md.data(), &nm) != HS_OK);      // count disagrees with the published frame
        F.n = nk;
        CHECK(hs_search_by_projection_frame(mh, tok + 1000, &F, lm.data(), (int)lm.size(), &pp, mi.data(), md.data(), &nm) != HS_OK);
        hs_stereo_params sp{ 300.f, 36.f, h, 100.f, 50.f, 31.f };
        std::vector<float> ur(nk), dz(nk);
        CHECK(hs_stereo_match_frames(mh, tok, tok, &sp, ur.data(), dz.data()) == HS_OK);
        CHECK(hs_stereo_match_frames(mh, tok, 0, &sp, ur.data(), dz.data()) != HS_OK);
        CHECK(hs_frame_release(0, tok) == HS_OK && hs_frame_release(0, tok) != HS_OK && hs_frame_info(0, tok, &ninfo) != HS_OK);
        // slot reuse: 17 more publishes of frames of different sizes (the slots' buffers grow) push the oldest out
        CHECK(hs_frame_publish(ex, 0, k.data(), nk, &tok) == HS_OK);
        for (int r = 0; r < 17; r++) { hs_frame_token t2 = 0; k[0].x = 500.f + r; CHECK(hs_frame_publish(ex, 0, k.data(), 1 + (r * 37) % cap, &t2) == HS_OK && t2 > tok); }
        CHECK(hs_frame_info(0, tok, &ninfo) != HS_OK);
        // four threads: two extractors publishing, one finder, one consumer (ASan: a slot freed or refilled under a reader fails the run)
        hs_orb* ex2 = nullptr;
        CHECK(hs_orb_create(&p, 0, &ex2) == HS_OK && hs_orb_reserve(ex2, w, h, 1) == HS_OK);
        std::vector<hs_keypoint> kb(cap); std::vector<uint8_t> db((size_t)cap * 32); int32_t nb2 = 0;
        CHECK(hs_orb_extract(ex2, img.data(), w, h, w, kb.data(), db.data(), cap, &nb2) == HS_OK);
        std::atomic<int> stop{0}, bad{0}; std::atomic<uint64_t> latest{0};
        auto publisher = [&](hs_orb* hx, int salt) {
            std::vector<hs_keypoint> kk(nk);
            for (int it = 0; it < 400; it++) {
                for (int i = 0; i < nk; i++) kk[i] = hs_keypoint{ (float)(salt + it), (float)i, 31.f, 0.f, 1.f, 0 };
                hs_frame_token t = 0;
                const int st = hs_frame_publish(hx, 0, kk.data(), 1 + (it * 7 + salt) % nk, &t);
                if (st == HS_OK) latest.store(t); else if (st != HS_ERR_CAPACITY) bad++;
            }
        };
        std::thread t1(publisher, ex, 1000), t2(publisher, ex2, 5000);
        std::thread t3([&] { std::vector<hs_keypoint> kk(nk); while (!stop.load()) { for (int i = 0; i < nk; i++) kk[i] = hs_keypoint{ 1000.f + (float)(rng() % 400), (float)i, 31.f, 0.f, 1.f, 0 }; hs_frame_token t = 0; (void)hs_frame_find(0, kk.data(), 1 + (int)(rng() % nk), &t); } });
        std::thread t4([&] {
            hs_frame_view G = F; std::vector<int32_t> mi2(lm.size()); std::vector<float> md2(lm.size()); int32_t nm2 = 0;
            while (!stop.load()) {
                const hs_frame_token t = latest.load(); int32_t cnt = 0;
                if (t && hs_frame_info(0, t, &cnt) == HS_OK) { G.n = cnt; const int st = hs_search_by_projection_frame(mh, t, &G, lm.data(), (int)lm.size(), &pp, mi2.data(), md2.data(), &nm2); if (st != HS_OK && st != HS_ERR_INVALID) bad++; }
            }
        });
        t1.join(); t2.join(); stop.store(1); t3.join(); t4.join();
        CHECK(bad.load() == 0);
        hs_orb_destroy(ex); hs_orb_destroy(ex2); hs_orb_destroy(mh);
        CHECK(hs_frame_cache_clear(0) == HS_OK && hs_frame_cache_clear(0) == HS_OK && hs_frame_cache_clear(3) == HS_OK);      // (LeakSanitizer: nothing of the cache may outlive this)
        CHECK(hs_frame_info(0, latest.load(), &ninfo) != HS_OK);
    }

    // ---- vocabulary files: valid round trips, then truncations and random corruptions of both formats
    char dir[] = "/tmp/hs_host_sanitize_XXXXXX";
    CHECK(mkdtemp(dir) != nullptr);
    const std::string ptxt = std::string(dir) + "/v.txt", pbin = std::string(dir) + "/v.bin", pbad_t = std::string(dir) + "/bad.txt", pbad_b = std::string(dir) + "/bad.bin";
    std::vector<int32_t> cb, cc, word; std::vector<uint8_t> desc; std::vector<float> weight;
    hs_vocab* v = make_vocab(4, 3, cb, cc, desc, word, weight);
    CHECK(v != nullptr);
    CHECK(hs_vocab_save(v, ptxt.c_str()) == HS_OK && hs_vocab_save(v, pbin.c_str()) == HS_OK);
    hs_vocab_destroy(v);
    int loaded = 0, refused = 0;
    for (const std::string& p : { ptxt, pbin }) {
        hs_vocab* a = nullptr;
        CHECK(hs_vocab_load(p.c_str(), &a) == HS_OK && a && hs_vocab_last_error()[0] == 0);
        int32_t k, L, nn, nw, sc, wt;
        CHECK(hs_vocab_info(a, &k, &L, &nn, &nw, &sc, &wt) == HS_OK && k == 4 && L == 3 && nw == 64);
        hs_vocab_destroy(a);
    }
    CHECK(hs_vocab_load((std::string(dir) + "/missing.txt").c_str(), &v) != HS_OK && hs_vocab_last_error()[0] != 0);
    const std::vector<uint8_t> good_t = slurp(ptxt), good_b = slurp(pbin);
    CHECK(!good_t.empty() && !good_b.empty());
    for (int i = 0; i < n_voc; i++) {
        const bool text = i & 1;
        std::vector<uint8_t> b = text ? good_t : good_b;
        const int kind = rnd(0, 3);
        if (kind == 0) b.resize((size_t)rnd(0, (int)b.size()));                                                      // truncated
        else if (kind == 1) for (int j = 0, m = rnd(1, 8); j < m; j++) b[(size_t)rnd(0, (int)b.size() - 1)] = (uint8_t)rng();         // a few bytes
        else if (kind == 2) { const size_t at = (size_t)rnd(0, (int)b.size() - 1); for (size_t j = at; j < b.size() && j < at + 16; j++) b[j] = text ? (uint8_t)"9-.e 7"[rng() % 6] : (uint8_t)0xFF; }   // a run (huge numbers / counts)
        else { const size_t at = (size_t)rnd(0, (int)b.size()); b.insert(b.begin() + at, (size_t)rnd(1, 64), text ? (uint8_t)' ' : (uint8_t)rng()); }               // inserted bytes
        const std::string& p = text ? pbad_t : pbad_b;
        spit(p, b);
        hs_vocab* a = nullptr;
        const int st = hs_vocab_load(p.c_str(), &a);
        if (st == HS_OK) {                                                                                            // still well-formed: it must be usable
            CHECK(a != nullptr);
            hs_vocab_tree T; CHECK(hs_vocab_get_tree(a, &T) == HS_OK && T.n_nodes >= 2);
            hs_vocab_destroy(a); loaded++;
        } else { CHECK(a == nullptr && hs_vocab_last_error()[0] != 0); refused++; }
    }
    unlink(ptxt.c_str()); unlink(pbin.c_str()); unlink(pbad_t.c_str()); unlink(pbad_b.c_str()); rmdir(dir);
    printf("HOST SANITIZE OK seed %llu: %d of %d geometries configured, %ld kernel launches issued into the stub, %d corrupted vocabularies refused, %d still well-formed\n",
           (unsigned long long)seed, accepted, tried, hip_stub_launches(), refused, loaded);
    return 0;
}
