// HipORBExtractor.h — header-only C++ adaptors that put the C ABI (include/hyslam_amd.h) behind hySLAM's own
// interfaces, so Tracking / Mapping see a drop-in:
//   HYSLAM::HipORBExtractor : HYSLAM::FeatureExtractor        (src/features/FeatureExtractor.h:25-37; replaces
//                                                              ORBExtractor, src/features/ORBExtractor.h:62-130)
//   HYSLAM::HipORBDistance  : HYSLAM::DescriptorDistance      (src/features/low_level/DescriptorDistance.h)
//   HYSLAM::HipStereomatcher                                   (src/features/Stereomatcher.h:25-51)
// Build inside hySLAM with -DHYSLAM_AMD_WITH_HYSLAM (real OpenCV + hySLAM headers); in this repository the same
// code is compiled against host/cv_compat.h so its gather/scatter logic is unit-tested without OpenCV.
#pragma once
#ifdef HYSLAM_AMD_WITH_HYSLAM
#include <FeatureExtractor.h>
#include <FeatureMatcher.h>
#include <FeatureViews.h>
#include <Camera.h>
#include <opencv2/core/core.hpp>
#else
#include "cv_compat.h"
#endif
#include <atomic>
#include <cstring>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <functional>
#include <future>
#include <mutex>
#include <thread>
#include <map>
#include <stdexcept>
#include <string>
#include <vector>
#include "../../include/hyslam_amd.h"
#if defined(__linux__)
#include <pthread.h>
#include <sched.h>
#endif

// helper threads of an extractor's descriptor scatter (0 = the calling thread alone; the environment variable HYSLAM_AMD_SCATTER_THREADS overrides).
// Every FeatureDescriptor construction copies the ONE shared_ptr<DescriptorDistance> all descriptors of an extractor share (FeatureDescriptor.cpp:6-10),
// so with more than one thread that reference count bounces between cores.  With helpers wherever the scheduler put them (two sockets) that cost what
// the split saved: 0 / 1 / 2 helpers = 0.145 / 0.144 / 0.145 ms (2 x 2000 descriptors, two extractors side by side).  With the helper in its caller's
// L3 domain (Worker::follow_caller) the bounce is an L3 hit: 0.095 -> 0.070 ms with one helper when hySLAM runs confined to one domain
// (ProcessStereoImage 0.535 -> 0.50 ms), nothing lost with free placement (0.588 / 0.589); a second helper adds nothing.  Default 1.
#ifndef HYSLAM_AMD_SCATTER_HELPERS
#define HYSLAM_AMD_SCATTER_HELPERS 1
#endif

namespace HYSLAM {

// Wall-clock split of the last call of an adaptor: gather = hySLAM objects -> flat arrays, abi = the C-ABI call (H2D + kernels + D2H),
// scatter = flat results -> hySLAM objects.  Three clock reads per call; read by tests/cpp/bench_adaptor.cpp.
struct HipCallTiming { double gather_ms = 0, abi_ms = 0, scatter_ms = 0; };
namespace hip_detail {
inline double ms_since(std::chrono::steady_clock::time_point t0) { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(); }
// One matcher / stereo-matcher handle per (calling thread, device), created on first use and destroyed when the thread exits.  These handles are
// created with default parameters and only lend a stream and scratch memory, so the device is the whole key: two factories on two devices used from
// one thread get two handles, two threads never share one (a handle is thread-compatible, include/hyslam_amd.h).
inline hs_orb* thread_handle(int device, const char* who) {
    struct Owner { std::map<int, hs_orb*> by_device; ~Owner() { for (auto& kv : by_device) if (kv.second) hs_orb_destroy(kv.second); } };
    static thread_local Owner o;
    auto it = o.by_device.find(device);
    if (it != o.by_device.end()) return it->second;
    hs_orb_params p; hs_orb_default_params(&p);
    hs_orb* h = nullptr;
    const int st = hs_orb_create(&p, device, &h);
    if (st != HS_OK) throw std::runtime_error(std::string(who) + ": " + hs_status_string(st));
    o.by_device[device] = h;
    return h;
}
// The CPUs that share a last-level cache with `cpu` (Linux sysfs), restricted to the CPUs this process may use; empty when unknown.  A helper that the
// scheduler drops on the other socket of a two-socket host reads every MapPoint the caller allocated across the socket link: on the GPU box (2 x EPYC
// 9575F, 256 CPUs, no affinity set) TrackLocalMap's landmark gather took 3.7-4.4 ms with free placement and 2.1 ms with caller and helpers in one L3
// domain (tools/experiments/r5_adaptor_runs.sh).  A helper therefore follows its caller: before a job it is confined to the caller's L3 domain
// (re-done only when the caller has moved to another domain).  Best effort: any failure leaves the thread where it is; HYSLAM_AMD_PIN_HELPERS=0 disables it.
inline bool l3_domain_read(int cpu, cpu_set_t* out);
// cached per CPU (the topology does not change; the process's affinity mask is taken as it is at the first question about a CPU)
inline bool l3_domain_of(int cpu, cpu_set_t* out) {
#if defined(__linux__)
    struct Cache { std::mutex mu; std::map<int, cpu_set_t> by_cpu; };
    static Cache c;
    std::lock_guard<std::mutex> g(c.mu);
    auto it = c.by_cpu.find(cpu);
    if (it == c.by_cpu.end()) {
        cpu_set_t d; CPU_ZERO(&d);
        if (!l3_domain_read(cpu, &d)) CPU_ZERO(&d);
        for (int k = 0; k < CPU_SETSIZE; k++) if (CPU_ISSET(k, &d)) c.by_cpu[k] = d;      // every CPU of the domain at once
        it = c.by_cpu.insert({ cpu, d }).first;
    }
    *out = it->second;
    return CPU_COUNT(out) > 0;
#else
    (void)cpu; (void)out; return false;
#endif
}
inline bool l3_domain_read(int cpu, cpu_set_t* out) {
#if defined(__linux__)
    char path[128];
    std::snprintf(path, sizeof(path), "/sys/devices/system/cpu/cpu%d/cache/index3/shared_cpu_list", cpu);
    std::FILE* f = std::fopen(path, "r");
    if (!f) return false;
    char buf[1024];
    const bool got = std::fgets(buf, sizeof(buf), f) != nullptr;
    std::fclose(f);
    if (!got) return false;
    cpu_set_t allowed;
    if (sched_getaffinity(0, sizeof(allowed), &allowed) != 0) return false;
    CPU_ZERO(out);
    for (const char* q = buf; *q && *q != '\n';) {              // "0-7,128-135"
        char* e = nullptr;
        const long a = std::strtol(q, &e, 10);
        if (e == q) break;
        long b = a;
        q = e;
        if (*q == '-') { b = std::strtol(q + 1, &e, 10); if (e == q + 1) break; q = e; }
        for (long c = a; c <= b && c < CPU_SETSIZE; c++) if (c >= 0 && CPU_ISSET((int)c, &allowed)) CPU_SET((int)c, out);
        if (*q == ',') q++;
    }
    return CPU_COUNT(out) > 0;
#else
    (void)cpu; (void)out; return false;
#endif
}
// A helper thread that lives as long as its owner: run(f) hands it one job, wait() blocks until that job is done.  (std::async starts a new thread per
// call: ~35 us to create it, and a fresh thread allocates from a fresh malloc arena whose pages are touched for the first time — for the ~60 us of
// FeatureDescriptor constructions an extractor call hands out, that overhead was most of the helper's time.)
class Worker {
public:
    Worker() = default;
    Worker(const Worker&) = delete;
    Worker& operator=(const Worker&) = delete;
    ~Worker() {
        { std::lock_guard<std::mutex> g(mu); quit = true; }
        cv.notify_all();
        if (th.joinable()) th.join();
    }
    template <class F> void run(F&& f) {
        std::unique_lock<std::mutex> g(mu);
        if (!th.joinable()) th = std::thread([this] { loop(); });
        follow_caller();
        job = std::forward<F>(f); busy = true;
        g.unlock();
        cv.notify_all();
    }
    void wait() { std::unique_lock<std::mutex> g(mu); done_cv.wait(g, [this] { return !busy; }); if (err) { std::exception_ptr e = err; err = nullptr; std::rethrow_exception(e); } }
private:
    // keep the helper in the calling thread's L3 domain (see l3_domain_of): one sched_getcpu() per job, a sysfs read + setaffinity only when the domain changed
    void follow_caller() {
#if defined(__linux__)
        static const bool enabled = [] { const char* e = std::getenv("HYSLAM_AMD_PIN_HELPERS"); return !e || std::atoi(e) != 0; }();
        if (!enabled) return;
        const int cpu = sched_getcpu();
        if (cpu < 0 || (have_domain && cpu < CPU_SETSIZE && CPU_ISSET(cpu, &domain))) return;
        cpu_set_t d;
        if (!l3_domain_of(cpu, &d)) return;
        // a caller pinned to ONE CPU (domain ∩ its affinity mask = that CPU): a helper pinned beside it would run serially with it — leave the helper where it is
        if (CPU_COUNT(&d) < 2) return;
        if (pthread_setaffinity_np(th.native_handle(), sizeof(d), &d) == 0) { domain = d; have_domain = true; }
#endif
    }
    void loop() {
        std::unique_lock<std::mutex> g(mu);
        for (;;) {
            cv.wait(g, [this] { return quit || (busy && job); });
            if (quit) return;
            std::function<void()> f; f.swap(job);
            g.unlock();
            std::exception_ptr e;
            try { f(); } catch (...) { e = std::current_exception(); }
            g.lock();
            err = e; busy = false;
            done_cv.notify_all();
        }
    }
    std::mutex mu; std::condition_variable cv, done_cv; std::thread th; std::function<void()> job; bool busy = false, quit = false; std::exception_ptr err;
#if defined(__linux__)
    cpu_set_t domain; bool have_domain = false;
#endif
};
// two helpers per CALLING thread, for adaptors that are short-lived objects themselves (a FeatureMatcher is made per call: FeatureFactory.cpp:7-9)
inline Worker* thread_workers() { static thread_local Worker w[2]; return w; }
// the GPU of the per-thread handles that have no factory to ask (HipStereomatcher's reference constructor, replace/FeatureMatcher.cc)
inline std::atomic<int>& default_device() { static std::atomic<int> d{ 0 }; return d; }
}  // namespace hip_detail

// ORBDistance (DescriptorDistance.cpp:9-25) kept on the host for the callers that still ask per-pair distances.
class HipORBDistance : public DescriptorDistance {
public:
    float distance(const cv::Mat& D1, const cv::Mat& D2) override {
        const uint64_t* a = reinterpret_cast<const uint64_t*>(D1.ptr(0));
        const uint64_t* b = reinterpret_cast<const uint64_t*>(D2.ptr(0));
        int d = 0;
        for (int i = 0; i < 4; i++) d += __builtin_popcountll(a[i] ^ b[i]);
        return static_cast<float>(d);
    }
};

class HipORBExtractor : public FeatureExtractor {
public:
    HipORBExtractor(std::shared_ptr<DescriptorDistance> dist_func_, FeatureExtractorSettings settings, int device = 0)
        : dist_func(dist_func_) {
        hs_orb_params p;
        hs_orb_default_params(&p);
        p.nfeatures = settings.nFeatures; p.scale_factor = settings.fScaleFactor; p.nlevels = settings.nLevels;
        p.cell_px = settings.N_CELLS; p.ini_th_fast = settings.init_threshold; p.min_th_fast = settings.min_threshold;
        int st = hs_orb_create(&p, device, &h);
        if (st != HS_OK) throw std::runtime_error(std::string("HipORBExtractor: ") + hs_status_string(st));
        cap = hs_orb_max_keypoints(h);
        kps.resize(cap); desc.resize((size_t)cap * HS_DESC_BYTES);
    }
    // HYSLAM::FeatureExtractor has no virtual destructor (src/features/FeatureExtractor.h:25-37): own a HipORBExtractor through
    // std::make_shared<HipORBExtractor>(...) (the control block remembers the concrete type, as HipORBFactory::getExtractor does) and never
    // `delete` it through a FeatureExtractor*.
    ~HipORBExtractor() { hs_orb_destroy(h); }
    HipORBExtractor(const HipORBExtractor&) = delete;
    HipORBExtractor& operator=(const HipORBExtractor&) = delete;

    // ORBExtractor::operator(), src/features/ORBExtractor.cpp:496-562: keypoints are cleared, descriptors appended.
    void operator()(cv::InputArray _image, cv::InputArray /*mask*/, std::vector<cv::KeyPoint>& _keypoints,
                    std::vector<FeatureDescriptor>& descriptors) override {
        cv::Mat image = _image.getMat();
        if (image.empty()) return;
        if (image.type() != CV_8UC1) throw std::runtime_error("HipORBExtractor: image must be CV_8UC1");
        int32_t n = 0;
        const auto t0 = std::chrono::steady_clock::now();
        int st = hs_orb_extract(h, image.ptr(0), image.cols, image.rows, (int)image.step, kps.data(), desc.data(), cap, &n);
        if (st != HS_OK) throw std::runtime_error(std::string("HipORBExtractor: ") + hs_status_string(st) + ": " + hs_orb_last_error(h));
        deliver(n, t0, _keypoints, descriptors);
    }
    // ImageProcessing::PreProcessImg + the extractor call in ONE call (src/main/ImageProcessing.cpp:44,55 / :76-77,82-83): `raw` is the frame as the camera
    // delivers it (CV_8UC1 / CV_8UC3 / CV_8UC4), `rgb` and `scale` are the camera's RGB and scale keys (Camera::RGB, Camera::scale).  The frame crosses PCIe
    // as it is, cv::resize by the scale and cvtColor to grey run on the device in front of the pyramid (hs_orb_extract_camera_batch) — no CPU resize of a
    // 2704 x 2028 x 3 frame in front of a 0.1 ms extraction.  `grey` (may be nullptr) receives what the reference keeps as mImGray / track_data.image.
    // Optional edit in hySLAM: replace `mImGray = PreProcessImg(mImGray, RGB, scale); (*extractor)(mImGray, cv::Mat(), keys, descs);` by this.
    void extractFromCamera(const cv::Mat& raw, bool rgb, float scale, cv::Mat* grey, std::vector<cv::KeyPoint>& _keypoints, std::vector<FeatureDescriptor>& descriptors) {
        if (raw.empty()) return;
        const int cn = raw.channels();
        if ((raw.type() & 7) != CV_8U || !(cn == 1 || cn == 3 || cn == 4)) throw std::runtime_error("HipORBExtractor: camera frames must be CV_8UC1, CV_8UC3 or CV_8UC4");
        hs_preprocess_params pp; pp.channels = cn; pp.rgb = rgb ? 1 : 0; pp.scale = scale; pp._pad = 0;
        int32_t ow = 0, oh = 0, n = 0;
        hs_preprocess_size(raw.cols, raw.rows, scale, &ow, &oh);
        if (ow < 1 || oh < 1) throw std::runtime_error("HipORBExtractor: the camera scale reduces the frame to nothing");
        if (grey) *grey = cv::Mat(oh, ow, CV_8UC1);
        const uint8_t* one[1] = { raw.ptr(0) };
        const auto t0 = std::chrono::steady_clock::now();
        const int st = hs_orb_extract_camera_batch(h, one, 1, raw.cols, raw.rows, (size_t)raw.step, &pp, kps.data(), desc.data(), cap, &n, grey ? grey->ptr(0) : nullptr);
        if (st != HS_OK) throw std::runtime_error(std::string("HipORBExtractor: ") + hs_status_string(st) + ": " + hs_orb_last_error(h));
        deliver(n, t0, _keypoints, descriptors);
    }
private:
    // the flat results of the call that started at t0 -> hySLAM's objects (keypoints cleared, descriptors appended: ORBExtractor.cpp:523,558-561)
    void deliver(int32_t n, std::chrono::steady_clock::time_point t0, std::vector<cv::KeyPoint>& _keypoints, std::vector<FeatureDescriptor>& descriptors) {
        // keep what was just extracted on the device for the matchers (device-to-device; a full cache or a failed publish only means the host path later)
        last_token = 0;
        if (publish_frames && n > 0 && hs_frame_publish(h, 0, kps.data(), n, &last_token) != HS_OK) last_token = 0;
        timing.gather_ms = 0; timing.abi_ms = hip_detail::ms_since(t0);
        const auto t1 = std::chrono::steady_clock::now();
        _keypoints.clear();
        _keypoints.resize(n);
        const size_t d0 = descriptors.size();
        descriptors.resize(d0 + n);                          // appended, like the reference (ORBExtractor.cpp:558-561)
        // Building 2000 FeatureDescriptors (a cv::Mat clone each: the reference's own object model) is 40 % of this call's wall time; one helper thread
        // kept in the caller's L3 domain takes half of them (HYSLAM_AMD_SCATTER_HELPERS: see the note at that macro).
        auto fill = [&](int a, int b) {
            for (int i = a; i < b; i++) {
                cv::KeyPoint& k = _keypoints[i];
                k.pt.x = kps[i].x; k.pt.y = kps[i].y; k.size = kps[i].size; k.angle = kps[i].angle;
                k.response = kps[i].response; k.octave = kps[i].octave; k.class_id = -1;
                descriptors[d0 + i] = FeatureDescriptor(cv::Mat(1, HS_DESC_BYTES, CV_8UC1, desc.data() + (size_t)i * HS_DESC_BYTES, HS_DESC_BYTES), dist_func);
            }
        };
        static const int n_helpers = [] { const char* e = std::getenv("HYSLAM_AMD_SCATTER_THREADS"); const int v = e ? std::atoi(e) : -1; return v >= 0 ? std::min(v, 2) : HYSLAM_AMD_SCATTER_HELPERS; }();
        if (n >= 512 && n_helpers > 0) {
            const int parts = n_helpers + 1, a = n / parts, b = n_helpers > 1 ? 2 * n / parts : n;
            int started = 0;
            try {
                helpers[0].run([&fill, a, b] { fill(a, b); }); started = 1;
                if (n_helpers > 1) { helpers[1].run([&fill, b, n] { fill(b, n); }); started = 2; }
                fill(0, a);
            } catch (...) { for (int w = 0; w < started; w++) { try { helpers[w].wait(); } catch (...) {} } throw; }      // the helpers hold references to this frame (run() itself can throw: thread creation)
            helpers[0].wait(); if (n_helpers > 1) helpers[1].wait();
        } else fill(0, n);
        timing.scatter_ms = hip_detail::ms_since(t1);
    }
public:
    int GetLevels() override { return hs_orb_get_levels(h); }
    float GetScaleFactor() override { return hs_orb_get_scale_factor(h); }
    std::vector<float> GetScaleFactors() override { return table(0); }
    std::vector<float> GetInverseScaleFactors() override { return table(1); }
    std::vector<float> GetScaleSigmaSquares() override { return table(2); }
    std::vector<float> GetInverseScaleSigmaSquares() override { return table(3); }
    hs_orb* handle() { return h; }
    HipCallTiming timing;             // of the last operator() call
    bool publish_frames = true;       // keep every extracted frame in the device's frame cache (hs_frame_publish) for the stereo matcher and the projection matchers
    hs_frame_token lastFrameToken() const { return last_token; }

private:
    hs_frame_token last_token = 0;
    hip_detail::Worker helpers[2];
    std::vector<float> table(int which) {
        std::vector<float> t[4];
        for (auto& v : t) v.resize(GetLevels());
        hs_orb_get_scale_tables(h, t[0].data(), t[1].data(), t[2].data(), t[3].data(), nullptr);
        return t[which];
    }
    hs_orb* h = nullptr;
    int cap = 0;
    std::vector<hs_keypoint> kps;
    std::vector<uint8_t> desc;
    std::shared_ptr<DescriptorDistance> dist_func;
};

// Stereomatcher (src/features/Stereomatcher.h:25-51): the reference's constructor (FeatureViews, Camera, FeatureMatcherSettings),
// computeStereoMatches(), getData(...) — src/main/ImageProcessing.cpp:100-103 compiles unchanged against it.
// The reference constructor carries no device handle, so the matcher runs on a handle of the CALLING THREAD (created on first use on device 0 or
// HipStereomatcher::setDefaultDevice, destroyed with the thread): stereo matchers of different cameras / threads never wait for each other.
class HipStereomatcher {
public:
    HipStereomatcher(FeatureViews views, Camera cam_data, FeatureMatcherSettings settings) : h(nullptr) {
        // what Stereomatcher::Stereomatcher reads (src/features/Stereomatcher.cpp:7-24)
        const auto t0 = std::chrono::steady_clock::now();
        const FeatureExtractorSettings orb_params = views.orbParams();
        sp.fx = cam_data.fx(); sp.mbf = cam_data.mbf; sp.n_rows = (int)cam_data.mnMaxY;
        sp.th_high = settings.TH_HIGH; sp.th_low = settings.TH_LOW; sp.size_ref = orb_params.size_ref;
        // Both views were extracted a moment ago by HipORBExtractor instances, which keep their results on the device (hs_frame_publish): when the
        // frame cache still holds both, only the keypoints are read here (to recognise the frames) and the matcher runs on the device copies —
        // no 2 x 2000 rawDescriptor() clones (the only accessor the reference's FeatureDescriptor offers: 0.16 ms), no upload.
        gather_keys(views.getKeys(), kL); gather_keys(views.getKeysR(), kR);
        const int device = hip_detail::default_device().load();
        if (kL.empty() || kR.empty() || hs_frame_find(device, kL.data(), (int)kL.size(), &tokL) != HS_OK || hs_frame_find(device, kR.data(), (int)kR.size(), &tokR) != HS_OK) tokL = tokR = 0;
        if (tokL) held = std::move(views);                 // (the by-value parameter: no copy) kept for the host path should a slot be reused before computeStereoMatches
        else { gather_descs(views.getDescriptors(), dL); gather_descs(views.getDescriptorsR(), dR); }
        timing.gather_ms = hip_detail::ms_since(t0);
    }
    // explicit-handle form (tests, callers that own an extractor on another device)
    HipStereomatcher(hs_orb* handle, const std::vector<cv::KeyPoint>& keys, const std::vector<cv::KeyPoint>& keysR,
                     const std::vector<FeatureDescriptor>& descs, const std::vector<FeatureDescriptor>& descsR,
                     float fx, float mbf, float mnMaxY, FeatureMatcherSettings settings, float size_ref = 31.f)
        : h(handle) {
        const auto t0 = std::chrono::steady_clock::now();
        sp.fx = fx; sp.mbf = mbf; sp.n_rows = (int)mnMaxY; sp.th_high = settings.TH_HIGH; sp.th_low = settings.TH_LOW; sp.size_ref = size_ref;
        gather(keys, descs, kL, dL); gather(keysR, descsR, kR, dR);
        timing.gather_ms = hip_detail::ms_since(t0);
    }
    void computeStereoMatches() {
        const auto t0 = std::chrono::steady_clock::now();
        mvuRight.assign(kL.size(), -1.0f); mvDepth.assign(kL.size(), -1.0f);
        hs_orb* use = h ? h : hip_detail::thread_handle(hip_detail::default_device().load(), "HipStereomatcher");
        frames_on_device = false;
        if (tokL && tokR) {
            const int st = hs_stereo_match_frames(use, tokL, tokR, &sp, mvuRight.data(), mvDepth.data());
            if (st == HS_OK) frames_on_device = true;
            else if (st != HS_ERR_INVALID) throw std::runtime_error(std::string("HipStereomatcher: ") + hs_status_string(st) + ": " + hs_orb_last_error(use));
            else { gather_descs(held.getDescriptors(), dL); gather_descs(held.getDescriptorsR(), dR); }      // a cache slot was reused meanwhile: the host path
        }
        if (!frames_on_device) {
            int st = hs_stereo_match(use, kL.data(), dL.data(), (int)kL.size(), kR.data(), dR.data(), (int)kR.size(), &sp, mvuRight.data(), mvDepth.data());
            if (st != HS_OK) throw std::runtime_error(std::string("HipStereomatcher: ") + hs_status_string(st));
        }
        timing.abi_ms = hip_detail::ms_since(t0);
    }
    void getData(std::vector<float>& mvuRight_, std::vector<float>& mvDepth_) { mvuRight_ = mvuRight; mvDepth_ = mvDepth; }
    void getData(FeatureViews& views) { views.setuRs(mvuRight); views.setDepths(mvDepth); }       // Stereomatcher.cpp:31-34
    static void setDefaultDevice(int device) { hip_detail::default_device().store(device); }
    HipCallTiming timing;             // gather = constructor, abi = computeStereoMatches
    bool frames_on_device = false;    // computeStereoMatches ran on the extractors' device copies (frame cache), nothing was gathered or uploaded

private:
    static void gather_keys(const std::vector<cv::KeyPoint>& k, std::vector<hs_keypoint>& ok) {
        ok.resize(k.size());
        for (size_t i = 0; i < k.size(); i++) ok[i] = hs_keypoint{ k[i].pt.x, k[i].pt.y, k[i].size, k[i].angle, k[i].response, k[i].octave };
    }
    static void gather_descs(const std::vector<FeatureDescriptor>& d, std::vector<uint8_t>& od) {
        od.resize(d.size() * HS_DESC_BYTES);
        for (size_t i = 0; i < d.size(); i++) {
            cv::Mat row = d[i].rawDescriptor();
            std::memcpy(od.data() + i * HS_DESC_BYTES, row.ptr(0), HS_DESC_BYTES);
        }
    }
    static void gather(const std::vector<cv::KeyPoint>& k, const std::vector<FeatureDescriptor>& d, std::vector<hs_keypoint>& ok, std::vector<uint8_t>& od) { gather_keys(k, ok); gather_descs(d, od); }
    hs_frame_token tokL = 0, tokR = 0; FeatureViews held;
    hs_orb* h; hs_stereo_params sp;
    std::vector<hs_keypoint> kL, kR; std::vector<uint8_t> dL, dR;
    std::vector<float> mvuRight, mvDepth;
};

// HipStereoFrontend — OPTIONAL one-call replacement for the body of ImageProcessing::ProcessStereoImage (src/main/ImageProcessing.cpp:80-103): left and
// right extraction and the stereo match as ONE ticket of the pipelined ingest (hs_orb_submit_batch / hs_orb_wait): both frames go through one launch
// sequence, the features never return to the device a second time (the drop-in path extracts, copies out, builds FeatureViews, and HipStereomatcher
// gathers the descriptors again and uploads them), and with submit() / collect() the upload of pair i+1 runs under the kernels of pair i — the
// shape of the reference's own bounded queue (System.cc:194-196).  Results are the same bits as HipORBExtractor x 2 + HipStereomatcher.
//   FeatureViews views = frontend.process(imLeft, imRight);          // keys, keysR, uRight, depth, descriptors, descriptorsR
// A maintainer swaps it in with ~10 lines; nothing else in hySLAM changes.
class HipStereoFrontend {
public:
    HipStereoFrontend(std::shared_ptr<DescriptorDistance> dist_func_, FeatureExtractorSettings settings, Camera cam_data, FeatureMatcherSettings matcher, int device = 0)
        : dist_func(dist_func_), orb_params(settings) {
        hs_orb_params p;
        hs_orb_default_params(&p);
        p.nfeatures = settings.nFeatures; p.scale_factor = settings.fScaleFactor; p.nlevels = settings.nLevels;
        p.cell_px = settings.N_CELLS; p.ini_th_fast = settings.init_threshold; p.min_th_fast = settings.min_threshold;
        int st = hs_orb_create(&p, device, &h);
        if (st != HS_OK) throw std::runtime_error(std::string("HipStereoFrontend: ") + hs_status_string(st));
        // what Stereomatcher::Stereomatcher reads (Stereomatcher.cpp:7-24); ImageProcessing passes default-constructed FeatureExtractorSettings to
        // FeatureViews, so size_ref is 31 there (ImageProcessing.cpp:85,100) — `views_params` reproduces that
        sp.fx = cam_data.fx(); sp.mbf = cam_data.mbf; sp.n_rows = (int)cam_data.mnMaxY;
        sp.th_high = matcher.TH_HIGH; sp.th_low = matcher.TH_LOW; sp.size_ref = views_params.size_ref;
    }
    ~HipStereoFrontend() { hs_orb_destroy(h); }
    HipStereoFrontend(const HipStereoFrontend&) = delete;
    HipStereoFrontend& operator=(const HipStereoFrontend&) = delete;

    // enqueue one pair (the images must stay valid until collect); at most two tickets in flight
    int32_t submit(const cv::Mat& imLeft, const cv::Mat& imRight) {
        if (imLeft.empty() || imRight.empty() || imLeft.type() != CV_8UC1 || imRight.type() != CV_8UC1 || imLeft.cols != imRight.cols || imLeft.rows != imRight.rows ||
            imLeft.step != imRight.step)
            throw std::runtime_error("HipStereoFrontend: two CV_8UC1 images of one size and row step");
        const uint8_t* imgs[2] = { imLeft.ptr(0), imRight.ptr(0) };
        int32_t t = 0;
        const auto t0 = std::chrono::steady_clock::now();
        int st = hs_orb_submit_batch(h, imgs, 2, imLeft.cols, imLeft.rows, (int)imLeft.step, &sp, &t);
        if (st != HS_OK) throw std::runtime_error(std::string("HipStereoFrontend: ") + hs_status_string(st) + ": " + hs_orb_last_error(h));
        timing.gather_ms = hip_detail::ms_since(t0);
        return t;
    }
    // the same with the frames as the camera delivers them (CV_8UC1 / CV_8UC3 / CV_8UC4; Camera::RGB, Camera::scale): ImageProcessing::PreProcessImg for both
    // eyes (ImageProcessing.cpp:76-77) runs on the device inside the ticket (hs_orb_submit_camera_batch)
    int32_t submitCamera(const cv::Mat& rawLeft, const cv::Mat& rawRight, bool rgb, float scale) {
        const int cn = rawLeft.empty() ? 0 : rawLeft.channels();
        if (rawLeft.empty() || rawRight.empty() || rawLeft.type() != rawRight.type() || (rawLeft.type() & 7) != CV_8U || !(cn == 1 || cn == 3 || cn == 4) ||
            rawLeft.cols != rawRight.cols || rawLeft.rows != rawRight.rows || rawLeft.step != rawRight.step)
            throw std::runtime_error("HipStereoFrontend: two 8-bit frames of 1, 3 or 4 channels, one size and one row step");
        hs_preprocess_params pp; pp.channels = cn; pp.rgb = rgb ? 1 : 0; pp.scale = scale; pp._pad = 0;
        const uint8_t* imgs[2] = { rawLeft.ptr(0), rawRight.ptr(0) };
        int32_t t = 0;
        const auto t0 = std::chrono::steady_clock::now();
        int st = hs_orb_submit_camera_batch(h, imgs, 2, rawLeft.cols, rawLeft.rows, (size_t)rawLeft.step, &pp, &sp, &t);
        if (st != HS_OK) throw std::runtime_error(std::string("HipStereoFrontend: ") + hs_status_string(st) + ": " + hs_orb_last_error(h));
        timing.gather_ms = hip_detail::ms_since(t0);
        return t;
    }
    // wait for a ticket and build the stereo FeatureViews (keys, keysR, uRight, depth, descriptors, descriptorsR — FeatureViews.h:20-81)
    FeatureViews collect(int32_t ticket) {
        const int cap = hs_orb_max_keypoints(h);
        if ((int)kps.size() < 2 * cap) { kps.resize(2 * (size_t)cap); desc.resize(2 * (size_t)cap * HS_DESC_BYTES); uR.resize(cap); depth.resize(cap); }
        int32_t n[2] = { 0, 0 };
        const auto t0 = std::chrono::steady_clock::now();
        int st = hs_orb_wait(h, ticket, kps.data(), desc.data(), n, cap, uR.data(), depth.data());
        if (st != HS_OK) throw std::runtime_error(std::string("HipStereoFrontend: ") + hs_status_string(st) + ": " + hs_orb_last_error(h));
        hs_frame_token tok = 0;
        if (n[0] > 0) (void)hs_frame_publish(h, 0, kps.data(), n[0], &tok);      // the left view stays on the device for Tracking's projection searches
        timing.abi_ms = hip_detail::ms_since(t0);
        const auto t1 = std::chrono::steady_clock::now();
        std::vector<cv::KeyPoint> keys[2]; std::vector<FeatureDescriptor> descs[2];
        // The right view is filled by a helper thread kept in this thread's L3 domain (Worker::follow_caller), into vectors sized here.  (Measured twice
        // before the helpers followed their caller: growing vectors inside the helper 0.32 -> 1.08 ms — a fresh thread's allocator arena, freed by the
        // caller —, pre-sized vectors filled by a helper wherever the scheduler had put it 0.32-0.40 -> 0.38 ms: no gain.)
        auto fill = [&](int s) {
            for (int i = 0; i < n[s]; i++) {
                const hs_keypoint& q = kps[(size_t)s * cap + i];
                cv::KeyPoint& k = keys[s][i];
                k.pt.x = q.x; k.pt.y = q.y; k.size = q.size; k.angle = q.angle; k.response = q.response; k.octave = q.octave; k.class_id = -1;
                descs[s][i] = FeatureDescriptor(cv::Mat(1, HS_DESC_BYTES, CV_8UC1, desc.data() + ((size_t)s * cap + i) * HS_DESC_BYTES, HS_DESC_BYTES), dist_func);
            }
        };
        for (int s = 0; s < 2; s++) { keys[s].resize(n[s]); descs[s].resize(n[s]); }
        static const bool parallel = [] { const char* e = std::getenv("HYSLAM_AMD_SCATTER_THREADS"); return !e || std::atoi(e) != 0; }();
        if (parallel && n[1] >= 512) {
            helper.run([&fill] { fill(1); });
            try { fill(0); } catch (...) { try { helper.wait(); } catch (...) {} throw; }      // the helper holds references to this frame
            helper.wait();
        } else { fill(0); fill(1); }
        FeatureViews views(keys[0], keys[1], std::vector<float>(uR.begin(), uR.begin() + n[0]), std::vector<float>(depth.begin(), depth.begin() + n[0]),
                           descs[0], descs[1], views_params);
        timing.scatter_ms = hip_detail::ms_since(t1);
        return views;
    }
    FeatureViews process(const cv::Mat& imLeft, const cv::Mat& imRight) { return collect(submit(imLeft, imRight)); }
    FeatureViews processCamera(const cv::Mat& rawLeft, const cv::Mat& rawRight, bool rgb, float scale) { return collect(submitCamera(rawLeft, rawRight, rgb, scale)); }
    hs_orb* handle() { return h; }
    HipCallTiming timing;             // gather = submit (H2D enqueue), abi = wait, scatter = FeatureViews construction

private:
    hs_orb* h = nullptr; hs_stereo_params sp;
    std::shared_ptr<DescriptorDistance> dist_func;
    FeatureExtractorSettings orb_params, views_params;     // views_params: default-constructed, like ImageProcessing.cpp:85
    std::vector<hs_keypoint> kps; std::vector<uint8_t> desc; std::vector<float> uR, depth;
    hip_detail::Worker helper;        // fills the right view's objects in collect()
};

}  // namespace HYSLAM
