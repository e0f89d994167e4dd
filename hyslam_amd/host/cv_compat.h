// cv_compat.h — the OpenCV / hySLAM types the adaptors' signatures mention, for builds WITHOUT OpenCV and hySLAM.
// Only used when HYSLAM_AMD_WITH_HYSLAM is not defined: unit-testing the adaptors' gather / scatter / replay logic in this repository,
// where OpenCV 3.4 and the hySLAM headers are absent.  Inside hySLAM the real headers are used instead.  Signatures mirror the reference
// (file:line below) so that the unit tests compile what hySLAM would compile; only the members the adaptors touch exist.
// These are not stand-ins for building the reference: nothing of the reference is compiled against them.
#pragma once
#include <cstdint>
#include <cstring>
#include <map>
#include <memory>
#include <set>
#include <string>
#include <vector>

#define CV_8U 0
#define CV_32F 5
#define CV_8UC1 0
#define CV_8UC3 16                     // CV_MAKETYPE(CV_8U, 3): the camera frames HipORBExtractor::extractFromCamera takes
#define CV_8UC4 24
#define CV_32FC1 5

namespace cv {
struct Point2f { float x = 0, y = 0; Point2f() {} Point2f(float x_, float y_) : x(x_), y(y_) {} };
struct KeyPoint {                      // field names of cv::KeyPoint (opencv2/core/types.hpp)
    Point2f pt; float size = 0, angle = -1, response = 0; int octave = 0, class_id = -1;
};
class Mat {                            // CV_8UC1 / CV_8UC3 / CV_8UC4 and CV_32FC1, 2-D, continuous or external rows
public:
    int rows = 0, cols = 0; size_t step = 0; uint8_t* data = nullptr;
    Mat() {}
    Mat(int r, int c, int type = CV_8UC1) : rows(r), cols(c), step((size_t)c * esz(type)), type_(type),
        buf_(std::make_shared<std::vector<uint8_t>>((size_t)r * c * esz(type))) { data = buf_->data(); }
    Mat(int r, int c, int type, const void* ext, size_t st = 0) : rows(r), cols(c), step(st ? st : (size_t)c * esz(type)), data((uint8_t*)ext), type_(type) {}
    bool empty() const { return !data || rows == 0 || cols == 0; }
    int type() const { return type_; }
    int channels() const { return (type_ >> 3) + 1; }
    Mat clone() const { Mat m(rows, cols, type_); for (int y = 0; y < rows; y++) std::memcpy(m.data + (size_t)y * m.step, data + (size_t)y * step, (size_t)cols * esz(type_)); return m; }
    uint8_t* ptr(int y = 0) { return data + (size_t)y * step; }
    const uint8_t* ptr(int y = 0) const { return data + (size_t)y * step; }
    template <class T> T& at(int r, int c) { return *reinterpret_cast<T*>(data + (size_t)r * step + (size_t)c * sizeof(T)); }
    template <class T> const T& at(int r, int c) const { return *reinterpret_cast<const T*>(data + (size_t)r * step + (size_t)c * sizeof(T)); }
    template <class T> T& at(int i) { return rows == 1 ? at<T>(0, i) : at<T>(i, 0); }
    template <class T> const T& at(int i) const { return rows == 1 ? at<T>(0, i) : at<T>(i, 0); }
    Mat getMat() const { return *this; }
private:
    static size_t esz(int t) { return (size_t)((t & 7) == CV_32F ? 4 : 1) * (size_t)((t >> 3) + 1); }
    int type_ = CV_8UC1;
    std::shared_ptr<std::vector<uint8_t>> buf_;
};
typedef const Mat& InputArray;
}  // namespace cv

namespace DBoW2 {                      // DBoW2/FeatureVector.h: node id -> indices of the local features
typedef unsigned int NodeId;
class FeatureVector : public std::map<NodeId, std::vector<unsigned int>> {};
}  // namespace DBoW2

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
// src/features/FeatureExtractor.h:25-37 — like the reference: NO virtual destructor
class FeatureExtractor {
public:
    virtual void operator()(cv::InputArray image, cv::InputArray mask, std::vector<cv::KeyPoint>& keypoints,
                            std::vector<FeatureDescriptor>& descriptors) = 0;
    virtual int GetLevels() = 0;
    virtual float GetScaleFactor() = 0;
    virtual std::vector<float> GetScaleFactors() = 0;
    virtual std::vector<float> GetInverseScaleFactors() = 0;
    virtual std::vector<float> GetScaleSigmaSquares() = 0;
    virtual std::vector<float> GetInverseScaleSigmaSquares() = 0;
};
// src/features/low_level/FeatureVocabulary.h (opaque here)
class FeatureVocabulary { public: virtual ~FeatureVocabulary() {} };

// src/core/FeatureViews.h:20-81
class FeatureViews {
public:
    FeatureViews() {}
    FeatureViews(std::vector<cv::KeyPoint> k, std::vector<FeatureDescriptor> d, FeatureExtractorSettings p)
        : N((int)k.size()), mvKeys(k), mDescriptors(d), orb_params(p) { is_empty = false; }
    FeatureViews(std::vector<cv::KeyPoint> k, std::vector<cv::KeyPoint> kR, std::vector<FeatureDescriptor> d, std::vector<FeatureDescriptor> dR,
                 FeatureExtractorSettings p) : is_stereo(true), N((int)k.size()), mvKeys(k), mvKeysRight(kR), mDescriptors(d), mDescriptorsRight(dR), orb_params(p) { is_empty = false; }
    FeatureViews(std::vector<cv::KeyPoint> k, std::vector<cv::KeyPoint> kR, std::vector<float> uR, std::vector<float> depth,
                 std::vector<FeatureDescriptor> d, std::vector<FeatureDescriptor> dR, FeatureExtractorSettings p)
        : is_stereo(true), N((int)k.size()), mvKeys(k), mvKeysRight(kR), mvuRight(uR), mvDepth(depth), mDescriptors(d), mDescriptorsRight(dR), orb_params(p) { is_empty = false; }
    bool empty() const { return is_empty; }
    bool isStereo() const { return is_stereo; }
    int numViews() const { return N; }
    cv::KeyPoint keypt(int i) const { return mvKeys[i]; }
    const FeatureDescriptor& descriptor(int i) const { return mDescriptors[i]; }
    float uR(int i) const { return (is_stereo && i < (int)mvuRight.size()) ? mvuRight[i] : -1.0f; }
    float depth(int i) const { return (is_stereo && i < (int)mvDepth.size()) ? mvDepth[i] : -1.0f; }
    FeatureExtractorSettings orbParams() const { return orb_params; }
    std::vector<cv::KeyPoint> getKeys() const { return mvKeys; }
    std::vector<cv::KeyPoint> getKeysR() const { return mvKeysRight; }
    std::vector<float> getuRs() const { return mvuRight; }
    std::vector<float> getDepths() const { return mvDepth; }
    std::vector<FeatureDescriptor> getDescriptors() const { return mDescriptors; }
    std::vector<FeatureDescriptor> getDescriptorsR() const { return mDescriptorsRight; }
    void setuRs(std::vector<float> uRs) { mvuRight = uRs; }
    void setDepths(std::vector<float> depths) { mvDepth = depths; }
private:
    bool is_stereo = false, is_empty = true; int N = 0;
    std::vector<cv::KeyPoint> mvKeys, mvKeysRight; std::vector<float> mvuRight, mvDepth;
    std::vector<FeatureDescriptor> mDescriptors, mDescriptorsRight; FeatureExtractorSettings orb_params;
};

// src/core/Camera.h (fields the matchers and the stereo matcher read)
class Camera {
public:
    int sensor = 0;                    // 0 mono, 1 stereo, 2 RGBD
    cv::Mat K = cv::Mat(3, 3, CV_32F);
    float mbf = 0;
    float mnMinX = 0, mnMaxX = 0, mnMinY = 0, mnMaxY = 0;
    float fx() const { return K.at<float>(0, 0); }
    float fy() const { return K.at<float>(1, 1); }
    float cx() const { return K.at<float>(0, 2); }
    float cy() const { return K.at<float>(1, 2); }
    float mb() const { return mbf / K.at<float>(0, 0); }
};

class KeyFrame;
// src/core/MapPoint.h:54-169 (accessors the matchers call)
class MapPoint {
public:
    cv::Mat GetWorldPos() { return mWorldPos.clone(); }
    cv::Mat GetNormal() { return mNormalVector.clone(); }
    FeatureDescriptor GetDescriptor() { return mDescriptor; }
    int Observations() { return nObs; }
    float getSize() const { return size; }
    float GetMinDistanceInvariance() { return 0.8f * mfMinDistance; }      // MapPoint.cc:139-143
    float GetMaxDistanceInvariance() { return 1.2f * mfMaxDistance; }      // MapPoint.cc:145-149
    bool isBad() { return mbBad; }
    bool Protected() { return n_protected > 0; }
    bool IsInKeyFrame(KeyFrame* pKF) { return in_keyframes.count(pKF) > 0 || mObservations.count(pKF) > 0; }
    int GetIndexInKeyFrame(KeyFrame* pKF) { auto it = mObservations.find(pKF); return it == mObservations.end() ? -1 : (int)it->second; }      // MapPoint.cc:124-131
    // test set-up (the reference fills these through Map / MapPointDB)
    cv::Mat mWorldPos = cv::Mat(3, 1, CV_32F), mNormalVector = cv::Mat(3, 1, CV_32F);
    FeatureDescriptor mDescriptor; int nObs = 0; float size = 0, mfMinDistance = 0, mfMaxDistance = 0; bool mbBad = false; int n_protected = 0;
    std::set<KeyFrame*> in_keyframes;
    std::map<KeyFrame*, size_t> mObservations;
};

// src/core/LandMarkMatches.h:27-66 — same observable behaviour as src/core/LandMarkMatches.cpp:6-51 (own wording)
struct LandMarkMatches {
    using LandMarkMatches_t = std::map<int, MapPoint*>;
    LandMarkMatches_t views_to_landmarks;
    std::map<int, bool> outliers;
    int n_matches = 0;
    MapPoint* hasAssociation(int i) const { auto it = views_to_landmarks.find(i); return it == views_to_landmarks.end() ? nullptr : it->second; }
    int hasAssociation(MapPoint* pMP) const { for (const auto& kv : views_to_landmarks) if (kv.second == pMP) return kv.first; return -1; }
    int associateLandMark(int i, MapPoint* pMP, bool replace) {
        if (!pMP) return -1;
        const bool view_taken = hasAssociation(i) != nullptr;
        const int other_view = hasAssociation(pMP);
        if (!view_taken && other_view < 0) { views_to_landmarks.insert({ i, pMP }); outliers.insert({ i, false }); ++n_matches; return 0; }
        if (!replace) return -1;
        views_to_landmarks[i] = pMP; outliers[i] = false;
        if (other_view >= 0 && other_view != i) views_to_landmarks.erase(other_view);      // the landmark moves to view i
        return 0;
    }
    using const_iterator = LandMarkMatches_t::const_iterator;
    const_iterator begin() const { return views_to_landmarks.begin(); }
    const_iterator end() const { return views_to_landmarks.end(); }
    const_iterator cbegin() const { return views_to_landmarks.cbegin(); }
    const_iterator cend() const { return views_to_landmarks.cend(); }
};

// src/core/Frame.h:69-213 (members the matchers read or call)
class Frame {
public:
    Frame() {}
    Frame(FeatureViews views_, const Camera& cam) : camera(cam), N(views_.numViews()), views(views_) {
        mnMinX = cam.mnMinX; mnMaxX = cam.mnMaxX; mnMinY = cam.mnMinY; mnMaxY = cam.mnMaxY;
    }
    void SetPose(cv::Mat Tcw) {        // Frame.cc: SetPose + UpdatePoseMatrices (mOw = -Rcw^T tcw, float products accumulated in double like cv::gemm)
        mTcw = Tcw.clone();
        mOw = cv::Mat(3, 1, CV_32F);
        for (int i = 0; i < 3; i++) { double s = 0; for (int k = 0; k < 3; k++) s += (double)mTcw.at<float>(k, i) * (double)mTcw.at<float>(k, 3); mOw.at<float>(i) = (float)-s; }
    }
    cv::Mat GetCameraCenter() { return mOw.clone(); }
    MapPoint* hasAssociation(int i) const { return matches.hasAssociation(i); }
    int hasAssociation(MapPoint* pMP) const { return matches.hasAssociation(pMP); }
    int associateLandMark(int i, MapPoint* pMP, bool replace) { return matches.associateLandMark(i, pMP, replace); }
    std::vector<MapPoint*> replicatemvpMapPoints() const { std::vector<MapPoint*> v(N, nullptr); for (const auto& kv : matches) if (kv.first < N) v[kv.first] = kv.second; return v; }
    const Camera& getCamera() const { return camera; }
    const FeatureViews& getViews() const { return views; }
    const LandMarkMatches& getLandMarkMatches() { return matches; }
    Camera camera;
    DBoW2::FeatureVector mFeatVec;
    int N = 0;
    cv::Mat mTcw;
    float mnMinX = 0, mnMaxX = 0, mnMinY = 0, mnMaxY = 0;
private:
    cv::Mat mOw;
    FeatureViews views;
    LandMarkMatches matches;
};

// src/core/KeyFrame.h (members the matchers read or call)
class KeyFrame {
public:
    KeyFrame() {}
    KeyFrame(FeatureViews views_, const Camera& cam) : camera(cam), views(views_) {
        mnMinX = cam.mnMinX; mnMaxX = cam.mnMaxX; mnMinY = cam.mnMinY; mnMaxY = cam.mnMaxY;
    }
    void SetPose(const cv::Mat& Tcw_) {
        Tcw = Tcw_.clone(); Ow = cv::Mat(3, 1, CV_32F);
        for (int i = 0; i < 3; i++) { double s = 0; for (int k = 0; k < 3; k++) s += (double)Tcw.at<float>(k, i) * (double)Tcw.at<float>(k, 3); Ow.at<float>(i) = (float)-s; }
    }
    cv::Mat GetPose() { return Tcw.clone(); }
    cv::Mat GetCameraCenter() { return Ow.clone(); }
    cv::Mat GetRotation() { cv::Mat R(3, 3, CV_32F); for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) R.at<float>(r, c) = Tcw.at<float>(r, c); return R; }      // KeyFrame.cc:126-130
    cv::Mat GetTranslation() { cv::Mat t(3, 1, CV_32F); for (int r = 0; r < 3; r++) t.at<float>(r) = Tcw.at<float>(r, 3); return t; }                                  // KeyFrame.cc:132-136
    MapPoint* hasAssociation(int i) const { return matches.hasAssociation(i); }
    int hasAssociation(MapPoint* pMP) const { return matches.hasAssociation(pMP); }
    int associateLandMark(int i, MapPoint* pMP, bool replace) { return matches.associateLandMark(i, pMP, replace); }
    std::vector<MapPoint*> GetMapPointMatches() { std::vector<MapPoint*> v; const int N = views.numViews(); v.reserve(N); for (int i = 0; i < N; ++i) v.push_back(matches.hasAssociation(i)); return v; }   // KeyFrame.cc:580-593
    bool IsInImage(const float& x, const float& y) const { return x >= mnMinX && x < mnMaxX && y >= mnMinY && y < mnMaxY; }                                             // KeyFrame.cc:371-374
    const Camera& getCamera() const { return camera; }
    const FeatureViews& getViews() const { return views; }
    const LandMarkMatches& getLandMarkMatches() { return matches; }
    Camera camera;
    DBoW2::FeatureVector mFeatVec;
    float mnMinX = 0, mnMaxX = 0, mnMinY = 0, mnMaxY = 0;
private:
    cv::Mat Tcw, Ow;
    FeatureViews views;
    LandMarkMatches matches;
};

#ifdef HYSLAM_AMD_COMPAT_UNPATCHED
// src/features/FeatureMatcher.h:105-176 AS IT IS in the reference (no patch): non-virtual search entry points, declared only — the bodies come from
// host/replace/FeatureMatcher.cc, the translation unit that takes the place of src/features/FeatureMatcher.cc (mode b).
class FeatureMatcher {
public:
    FeatureMatcher(float nnratio = 0.6, bool checkOri = true);
    FeatureMatcher(FeatureMatcherSettings settings);
    int SearchByProjection(Frame& F, const std::vector<MapPoint*>& vpMapPoints, const float th = 3);
    int SearchByProjection(Frame& CurrentFrame, const Frame& LastFrame, const float th, const bool bMono);
    int SearchByProjection(Frame& CurrentFrame, KeyFrame* pKF, const std::set<MapPoint*>& sAlreadyFound, const float th, const int ORBdist);
    int SearchByProjection(KeyFrame* pKF, cv::Mat Scw, const std::vector<MapPoint*>& vpPoints, std::vector<MapPoint*>& vpMatched, int th);
    int SearchByBoW(KeyFrame* pKF, Frame& F, std::map<size_t, MapPoint*>& matches);
    int SearchByBoW(KeyFrame* pKF1, KeyFrame* pKF2, std::vector<MapPoint*>& vpMatches12);
    int SearchByBoW2(KeyFrame* pKF1, KeyFrame* pKF2, std::vector<MapPoint*>& vpMatches12);
    int SearchForTriangulation(KeyFrame* pKF1, KeyFrame* pKF2, cv::Mat F12, std::vector<std::pair<size_t, size_t>>& vMatchedPairs, const bool bOnlyStereo);
    int SearchForInitialization(Frame& F1, Frame& F2, std::vector<cv::Point2f>& vbPrevMatched, std::vector<int>& vnMatches12, int windowSize = 10);
    int Fuse(KeyFrame* pKF, const std::vector<MapPoint*>& vpMapPoints, std::map<std::size_t, MapPoint*>& fuse_matches, const float th = 3.0, const float reprojection_err = 5.99);
    int Fuse(KeyFrame* pKF, cv::Mat Scw, const std::vector<MapPoint*>& vpPoints, float th, std::vector<MapPoint*>& vpReplacePoint);
    int SearchBySim3(KeyFrame* pKF1, KeyFrame* pKF2, std::vector<MapPoint*>& vpMatches12, const float& s12, const cv::Mat& R12, const cv::Mat& t12, const float th);
    static const int HISTO_LENGTH;
protected:
    float mfNNratio; bool mbCheckOrientation; float TH_LOW; float TH_HIGH;
};

// src/features/FeatureFactory.h:21-33 as it is in the reference; getFeatureMatcher() with the body of src/features/FeatureFactory.cpp:7-9
class FeatureFactory {
public:
    virtual ~FeatureFactory() {}
    virtual std::shared_ptr<FeatureExtractor> getExtractor(std::string type) = 0;
    virtual std::shared_ptr<FeatureExtractor> getExtractor(FeatureExtractorSettings settings) = 0;
    virtual FeatureVocabulary* getVocabulary(std::string type) = 0;
    virtual std::shared_ptr<DescriptorDistance> getDistanceFunc() = 0;
    virtual FeatureExtractorSettings getFeatureExtractorSettings() = 0;
    std::unique_ptr<FeatureMatcher> getFeatureMatcher() { return std::make_unique<FeatureMatcher>(matcher_settings); }
    FeatureMatcherSettings getFeatureMatcherSettings() const { return matcher_settings; }
    void setFeatureMatcherSettings(FeatureMatcherSettings fm_settings) { matcher_settings = fm_settings; }
protected:
    FeatureMatcherSettings matcher_settings;
};
#else
// src/features/FeatureMatcher.h:105-176 AFTER the two-line patch of INTEGRATION.md §3 (`virtual` on the search entry points).  The
// reference's own bodies are not restated: in this shim the base class only defines the interface the adaptor overrides.
class FeatureMatcher {
public:
    FeatureMatcher(FeatureMatcherSettings settings) : mfNNratio(settings.nnratio), mbCheckOrientation(settings.checkOri), TH_LOW(settings.TH_LOW), TH_HIGH(settings.TH_HIGH) {}
    virtual ~FeatureMatcher() {}
    virtual int SearchByProjection(Frame& F, const std::vector<MapPoint*>& vpMapPoints, const float th = 3) { (void)F; (void)vpMapPoints; (void)th; return -1; }
    virtual int SearchByProjection(Frame& CurrentFrame, const Frame& LastFrame, const float th, const bool bMono) { (void)CurrentFrame; (void)LastFrame; (void)th; (void)bMono; return -1; }
    virtual int SearchByProjection(Frame& CurrentFrame, KeyFrame* pKF, const std::set<MapPoint*>& sAlreadyFound, const float th, const int ORBdist) { (void)CurrentFrame; (void)pKF; (void)sAlreadyFound; (void)th; (void)ORBdist; return -1; }
    virtual int SearchByBoW(KeyFrame* pKF, Frame& F, std::map<size_t, MapPoint*>& matches) { (void)pKF; (void)F; (void)matches; return -1; }
    virtual int SearchForInitialization(Frame& F1, Frame& F2, std::vector<cv::Point2f>& vbPrevMatched, std::vector<int>& vnMatches12, int windowSize = 10) { (void)F1; (void)F2; (void)vbPrevMatched; (void)vnMatches12; (void)windowSize; return -1; }
    virtual int Fuse(KeyFrame* pKF, const std::vector<MapPoint*>& vpMapPoints, std::map<std::size_t, MapPoint*>& fuse_matches, const float th = 3.0, const float reprojection_err = 5.99) { (void)pKF; (void)vpMapPoints; (void)fuse_matches; (void)th; (void)reprojection_err; return -1; }
    virtual int SearchByProjection(KeyFrame* pKF, cv::Mat Scw, const std::vector<MapPoint*>& vpPoints, std::vector<MapPoint*>& vpMatched, int th) { (void)pKF; (void)Scw; (void)vpPoints; (void)vpMatched; (void)th; return -1; }
    virtual int SearchByBoW(KeyFrame* pKF1, KeyFrame* pKF2, std::vector<MapPoint*>& vpMatches12) { (void)pKF1; (void)pKF2; (void)vpMatches12; return -1; }
    virtual int SearchByBoW2(KeyFrame* pKF1, KeyFrame* pKF2, std::vector<MapPoint*>& vpMatches12) { (void)pKF1; (void)pKF2; (void)vpMatches12; return -1; }
    virtual int SearchForTriangulation(KeyFrame* pKF1, KeyFrame* pKF2, cv::Mat F12, std::vector<std::pair<size_t, size_t>>& vMatchedPairs, const bool bOnlyStereo) { (void)pKF1; (void)pKF2; (void)F12; (void)vMatchedPairs; (void)bOnlyStereo; return -1; }
    virtual int Fuse(KeyFrame* pKF, cv::Mat Scw, const std::vector<MapPoint*>& vpPoints, float th, std::vector<MapPoint*>& vpReplacePoint) { (void)pKF; (void)Scw; (void)vpPoints; (void)th; (void)vpReplacePoint; return -1; }
    virtual int SearchBySim3(KeyFrame* pKF1, KeyFrame* pKF2, std::vector<MapPoint*>& vpMatches12, const float& s12, const cv::Mat& R12, const cv::Mat& t12, const float th) { (void)pKF1; (void)pKF2; (void)vpMatches12; (void)s12; (void)R12; (void)t12; (void)th; return -1; }
protected:
    float mfNNratio; bool mbCheckOrientation; float TH_LOW; float TH_HIGH;
};

// src/features/FeatureFactory.h:21-33 AFTER the patch of INTEGRATION.md §3 (`virtual` on getFeatureMatcher)
class FeatureFactory {
public:
    virtual ~FeatureFactory() {}
    virtual std::shared_ptr<FeatureExtractor> getExtractor(std::string type) = 0;
    virtual std::shared_ptr<FeatureExtractor> getExtractor(FeatureExtractorSettings settings) = 0;
    virtual FeatureVocabulary* getVocabulary(std::string type) = 0;
    virtual std::shared_ptr<DescriptorDistance> getDistanceFunc() = 0;
    virtual FeatureExtractorSettings getFeatureExtractorSettings() = 0;
    virtual std::unique_ptr<FeatureMatcher> getFeatureMatcher() { return std::make_unique<FeatureMatcher>(matcher_settings); }
    FeatureMatcherSettings getFeatureMatcherSettings() const { return matcher_settings; }
    void setFeatureMatcherSettings(FeatureMatcherSettings fm_settings) { matcher_settings = fm_settings; }
protected:
    FeatureMatcherSettings matcher_settings;
};
#endif
}  // namespace HYSLAM
