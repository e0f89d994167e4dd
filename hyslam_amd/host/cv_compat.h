// cv_compat.h — the few OpenCV / hySLAM types the adaptor's signatures mention, for builds WITHOUT OpenCV.
// Only used when HYSLAM_AMD_WITH_HYSLAM is not defined (unit-testing the adaptor's gather/scatter logic in this
// repository, where OpenCV 3.4 and the hySLAM headers are absent).  Inside hySLAM the real headers are used instead.
// These are not stand-ins for building the reference: nothing of the reference is compiled against them.
#pragma once
#include <cstdint>
#include <cstring>
#include <memory>
#include <vector>

namespace cv {
struct Point2f { float x = 0, y = 0; };
struct KeyPoint {                      // field names of cv::KeyPoint (opencv2/core/types.hpp)
    Point2f pt; float size = 0, angle = -1, response = 0; int octave = 0, class_id = -1;
};
class Mat {                            // 8-bit single-channel only
public:
    int rows = 0, cols = 0; size_t step = 0; uint8_t* data = nullptr;
    Mat() {}
    Mat(int r, int c) : rows(r), cols(c), step((size_t)c), buf_(std::make_shared<std::vector<uint8_t>>((size_t)r * c)) { data = buf_->data(); }
    Mat(int r, int c, uint8_t* ext, size_t st) : rows(r), cols(c), step(st), data(ext) {}
    bool empty() const { return !data || rows == 0 || cols == 0; }
    int type() const { return 0; }     // CV_8UC1
    Mat clone() const { Mat m(rows, cols); for (int y = 0; y < rows; y++) std::memcpy(m.data + (size_t)y * m.step, data + (size_t)y * step, cols); return m; }
    uint8_t* ptr(int y) { return data + (size_t)y * step; }
    const uint8_t* ptr(int y) const { return data + (size_t)y * step; }
    Mat getMat() const { return *this; }
private:
    std::shared_ptr<std::vector<uint8_t>> buf_;
};
typedef const Mat& InputArray;
}  // namespace cv

namespace HYSLAM {
// src/core/FeatureExtractorSettings.h:19-32
class FeatureExtractorSettings {
public:
    int nFeatures = 1000; float fScaleFactor = 1.2f; int nLevels = 8; int init_threshold = 20; int min_threshold = 4;
    float size_ref = 31; float sigma_ref = 1.0; int N_CELLS = 30;
};
// src/features/FeatureMatcher.h:98-103
struct FeatureMatcherSettings { float nnratio = 0.6f; float TH_HIGH = 100.0f; float TH_LOW = 50.0f; bool checkOri = true; };
// src/features/low_level/DescriptorDistance.h:22-26
class DescriptorDistance { public: virtual ~DescriptorDistance() {} virtual float distance(const cv::Mat& a, const cv::Mat& b) = 0; };
// src/features/low_level/FeatureDescriptor.h:26-38 (a 1 x 32 CV_8U row + its distance functor)
class FeatureDescriptor {
public:
    FeatureDescriptor() {}
    FeatureDescriptor(cv::Mat row, std::shared_ptr<DescriptorDistance> d) : descriptor(row.clone()), dist_func(d) {}
    cv::Mat rawDescriptor() const { return descriptor.clone(); }
    float distance(const FeatureDescriptor& o) const { return dist_func->distance(descriptor, o.descriptor); }
private:
    cv::Mat descriptor; std::shared_ptr<DescriptorDistance> dist_func;
};
// src/features/FeatureExtractor.h:25-37
class FeatureExtractor {
public:
    virtual ~FeatureExtractor() {}
    virtual void operator()(cv::InputArray image, cv::InputArray mask, std::vector<cv::KeyPoint>& keypoints,
                            std::vector<FeatureDescriptor>& descriptors) = 0;
    virtual int GetLevels() = 0;
    virtual float GetScaleFactor() = 0;
    virtual std::vector<float> GetScaleFactors() = 0;
    virtual std::vector<float> GetInverseScaleFactors() = 0;
    virtual std::vector<float> GetScaleSigmaSquares() = 0;
    virtual std::vector<float> GetInverseScaleSigmaSquares() = 0;
};
}  // namespace HYSLAM
