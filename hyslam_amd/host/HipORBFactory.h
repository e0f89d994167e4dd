// HipORBFactory.h — HYSLAM::FeatureFactory for the HIP path: the plug-in seam of the reference (src/features/FeatureFactory.h:21-33,
// chosen by the YAML key `Features:` in System::System, src/main/System.cc:77-85).
//
//   HYSLAM::HipORBFactory : HYSLAM::FeatureFactory      replaces ORBFactory (src/features/ORBFactory.h:20-41, ORBFactory.cpp:13-85)
//
// getExtractor()      -> std::make_shared<HipORBExtractor>          (ORBFactory.cpp:32-40)
// getDistanceFunc()   -> HipORBDistance                             (ORBFactory.cpp:47-49)
// getFeatureMatcher() -> std::make_unique<HipFeatureMatcher>        (FeatureFactory.cpp:7-9; needs the `virtual` patch of INTEGRATION.md §3)
// getVocabulary()     -> inside hySLAM: the reference's own ORBVocabulary (DBoW2 on the CPU; ORBFactory.cpp:42-45); here: nullptr
// Settings: inside hySLAM the YAML is read exactly like ORBFactory::LoadSettings (ORBFactory.cpp:55-85); without OpenCV the per-camera
// settings are handed to the constructor.
#pragma once
#include "HipORBExtractor.h"
#include "HipFeatureMatcher.h"
#ifdef HYSLAM_AMD_WITH_HYSLAM
#include <FeatureFactory.h>
#include <ORBVocabulary.h>
#include <opencv2/core/core.hpp>
#endif
#include <map>
#include <memory>
#include <string>

namespace HYSLAM {

class HipORBFactory : public FeatureFactory {
public:
    // defaults of ORBFactory::ORBFactory() (ORBFactory.cpp:13-25)
    explicit HipORBFactory(int device_ = 0) : device(device_) {
        extractor_settings.nFeatures = 1000; extractor_settings.fScaleFactor = 1.2f; extractor_settings.nLevels = 8; extractor_settings.N_CELLS = 30;
        extractor_settings.init_threshold = 20; extractor_settings.min_threshold = 4;
        matcher_settings.TH_HIGH = 100.0f; matcher_settings.TH_LOW = 50.0f;
    }
#ifdef HYSLAM_AMD_WITH_HYSLAM
    HipORBFactory(std::string settings_path_, int device_ = 0) : device(device_), settings_path(settings_path_) { LoadSettings(settings_path, "SLAM"); }
#endif
    // per-camera-type settings without a YAML file (type = "SLAM", "Imaging", ...: config/slam_feature_config.yaml:6-35)
    HipORBFactory(std::map<std::string, FeatureExtractorSettings> per_type, FeatureMatcherSettings matcher, int device_ = 0)
        : device(device_), settings_by_type(per_type) {
        matcher_settings = matcher;
        auto it = settings_by_type.find("SLAM");
        if (it != settings_by_type.end()) extractor_settings = it->second;
    }

    std::shared_ptr<FeatureExtractor> getExtractor(std::string type) override {
        LoadSettings(settings_path, type);
        return getExtractor(extractor_settings);
    }
    std::shared_ptr<FeatureExtractor> getExtractor(FeatureExtractorSettings settings) override {
        return std::make_shared<HipORBExtractor>(getDistanceFunc(), settings, device);      // make_shared: see ~HipORBExtractor
    }
    FeatureVocabulary* getVocabulary(std::string type) override {
        LoadSettings(settings_path, type);
#ifdef HYSLAM_AMD_WITH_HYSLAM
        return new ORBVocabulary(vocab_path);
#else
        return nullptr;
#endif
    }
    std::shared_ptr<DescriptorDistance> getDistanceFunc() override { return std::make_shared<HipORBDistance>(); }
    FeatureExtractorSettings getFeatureExtractorSettings() override { return extractor_settings; }

    // Every call site builds a short-lived matcher through this function (TrackLocalMap.cpp:72-75, TrackMotionModel.cpp:24, ...).  Matchers
    // created on different threads must not share a handle, and a factory must hand out handles of ITS device: the handle is looked up per
    // (calling thread, device) — hip_detail::thread_handle in HipORBExtractor.h.
#ifndef HYSLAM_AMD_UNPATCHED_MATCHER
    std::unique_ptr<FeatureMatcher> getFeatureMatcher() override {
        return std::make_unique<HipFeatureMatcher>(matcher_settings, hip_detail::thread_handle(device, "HipORBFactory"));
    }
#else
    // integration (b), no header edit: the base class's non-virtual getFeatureMatcher() hands out the reference's own FeatureMatcher type, whose
    // member functions host/replace/FeatureMatcher.cc defines over the C ABI (on the calling thread's handle on hip_detail::default_device());
    // nothing to override here, the factory only makes that device its own
    void useDeviceForMatchers() const { hip_detail::default_device().store(device); }
#endif
    int deviceIndex() const { return device; }

private:
    void LoadSettings(const std::string& path, const std::string& type) {
#ifdef HYSLAM_AMD_WITH_HYSLAM
        if (!path.empty()) {                                   // ORBFactory::LoadSettings, ORBFactory.cpp:55-85
            cv::FileStorage fSettings(path, cv::FileStorage::READ);
            cv::FileNode extract = fSettings["ORB"][type]["Extractor"];
            extractor_settings.nFeatures = extract["N_Features"]; extractor_settings.fScaleFactor = extract["scale_factor"];
            extractor_settings.nLevels = extract["N_Levels"]; extractor_settings.N_CELLS = extract["N_Cells"];
            extractor_settings.init_threshold = extract["threshold_init"]; extractor_settings.min_threshold = extract["threshold_min"];
            cv::FileNode match = fSettings["ORB"][type]["Matcher"];
            matcher_settings.TH_HIGH = match["threshold_high"]; matcher_settings.TH_LOW = match["threshold_low"];
            vocab_path = fSettings["ORB"][type]["Vocabulary"].string();
            fSettings.release();
            return;
        }
#else
        (void)path;
#endif
        auto it = settings_by_type.find(type);
        if (it != settings_by_type.end()) extractor_settings = it->second;
    }

    int device;
    FeatureExtractorSettings extractor_settings;
    std::string vocab_path, settings_path;
    std::map<std::string, FeatureExtractorSettings> settings_by_type;
};

}  // namespace HYSLAM
