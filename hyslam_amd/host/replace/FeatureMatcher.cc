// replace/FeatureMatcher.cc — integration WITHOUT a hySLAM header edit (mode b).
//
// Put this file in the place of src/features/FeatureMatcher.cc in hySLAM's source list (src/CMakeLists.txt) — one CMake list edit, no header
// touched.  It defines the member functions that the reference's OWN header declares (src/features/FeatureMatcher.h:105-176: non-virtual,
// called through the std::unique_ptr<FeatureMatcher> that the non-virtual FeatureFactory::getFeatureMatcher() returns,
// src/features/FeatureFactory.cpp:7-9 — that file stays as it is); every body is gather -> one C-ABI call -> replay (HipMatcherCore,
// ../HipFeatureMatcher.h) on the HIP handle of the CALLING THREAD (hip_detail::thread_handle: Tracking, the Mapping jobs and LoopClosing each
// get their own on first use; the device is HipStereomatcher::setDefaultDevice's, default 0).
// The protected helpers of the reference's bodies (_SearchByProjection_, _SearchByBoW_, ComputeThreeMaxima) are declared by the header and
// have no caller outside the file this one replaces: they are not defined.
// Compile with -DHYSLAM_AMD_WITH_HYSLAM -DHYSLAM_AMD_UNPATCHED_MATCHER inside hySLAM; this repository compiles it against host/cv_compat.h with
// -DHYSLAM_AMD_COMPAT_UNPATCHED (the reference's unpatched declarations) for tests/cpp/test_matcher_adaptor.cpp.
#ifndef HYSLAM_AMD_UNPATCHED_MATCHER
#define HYSLAM_AMD_UNPATCHED_MATCHER
#endif
#include "../HipFeatureMatcher.h"

namespace HYSLAM {

const int FeatureMatcher::HISTO_LENGTH = 30;                         // FeatureMatcher.cc:40

// FeatureMatcher.cc:45-47 leaves TH_LOW / TH_HIGH uninitialised in this constructor (the factory never uses it): the struct's
// defaults are used here instead of indeterminate values
FeatureMatcher::FeatureMatcher(float nnratio, bool checkOri) : mfNNratio(nnratio), mbCheckOrientation(checkOri), TH_LOW(FeatureMatcherSettings().TH_LOW), TH_HIGH(FeatureMatcherSettings().TH_HIGH) {}
// FeatureMatcher.cc:50-55
FeatureMatcher::FeatureMatcher(FeatureMatcherSettings settings) : mfNNratio(settings.nnratio), mbCheckOrientation(settings.checkOri), TH_LOW(settings.TH_LOW), TH_HIGH(settings.TH_HIGH) {}

#define HS_CORE() HipMatcherCore(mfNNratio, mbCheckOrientation, TH_LOW, TH_HIGH, hip_detail::thread_handle(hip_detail::default_device().load(), "FeatureMatcher"))

int FeatureMatcher::SearchByProjection(Frame &F, const std::vector<MapPoint*> &vpMapPoints, const float th) { return HS_CORE().SearchByProjection(F, vpMapPoints, th); }
int FeatureMatcher::SearchByProjection(Frame &CurrentFrame, const Frame &LastFrame, const float th, const bool bMono) { return HS_CORE().SearchByProjection(CurrentFrame, LastFrame, th, bMono); }
int FeatureMatcher::SearchByProjection(Frame &CurrentFrame, KeyFrame* pKF, const std::set<MapPoint*> &sAlreadyFound, const float th, const int ORBdist) { return HS_CORE().SearchByProjection(CurrentFrame, pKF, sAlreadyFound, th, ORBdist); }
int FeatureMatcher::SearchByProjection(KeyFrame* pKF, cv::Mat Scw, const std::vector<MapPoint*> &vpPoints, std::vector<MapPoint*> &vpMatched, int th) { return HS_CORE().SearchByProjection(pKF, Scw, vpPoints, vpMatched, th); }
int FeatureMatcher::SearchByBoW(KeyFrame *pKF, Frame &F, std::map<size_t, MapPoint*> &matches) { return HS_CORE().SearchByBoW(pKF, F, matches); }
int FeatureMatcher::SearchByBoW(KeyFrame *pKF1, KeyFrame* pKF2, std::vector<MapPoint*> &vpMatches12) { return HS_CORE().SearchByBoW(pKF1, pKF2, vpMatches12); }
int FeatureMatcher::SearchByBoW2(KeyFrame *pKF1, KeyFrame* pKF2, std::vector<MapPoint*> &vpMatches12) { return HS_CORE().SearchByBoW2(pKF1, pKF2, vpMatches12); }
int FeatureMatcher::SearchForTriangulation(KeyFrame *pKF1, KeyFrame* pKF2, cv::Mat F12, std::vector<std::pair<size_t, size_t> > &vMatchedPairs, const bool bOnlyStereo) { return HS_CORE().SearchForTriangulation(pKF1, pKF2, F12, vMatchedPairs, bOnlyStereo); }
int FeatureMatcher::SearchForInitialization(Frame &F1, Frame &F2, std::vector<cv::Point2f> &vbPrevMatched, std::vector<int> &vnMatches12, int windowSize) { return HS_CORE().SearchForInitialization(F1, F2, vbPrevMatched, vnMatches12, windowSize); }
int FeatureMatcher::Fuse(KeyFrame* pKF, const std::vector<MapPoint *> &vpMapPoints, std::map<std::size_t, MapPoint*> &fuse_matches, const float th, const float reprojection_err) { return HS_CORE().Fuse(pKF, vpMapPoints, fuse_matches, th, reprojection_err); }
int FeatureMatcher::Fuse(KeyFrame* pKF, cv::Mat Scw, const std::vector<MapPoint*> &vpPoints, float th, std::vector<MapPoint *> &vpReplacePoint) { return HS_CORE().Fuse(pKF, Scw, vpPoints, th, vpReplacePoint); }
int FeatureMatcher::SearchBySim3(KeyFrame* pKF1, KeyFrame* pKF2, std::vector<MapPoint *> &vpMatches12, const float &s12, const cv::Mat &R12, const cv::Mat &t12, const float th) { return HS_CORE().SearchBySim3(pKF1, pKF2, vpMatches12, s12, R12, t12, th); }

#undef HS_CORE

}  // namespace HYSLAM
