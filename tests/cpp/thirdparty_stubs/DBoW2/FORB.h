// DECLARATION-ONLY stand-in: DBoW2/FORB.h
#pragma once
#include <opencv2/core/core.hpp>
#include <string>
#include <vector>
namespace DBoW2 {
class FORB { public: typedef cv::Mat TDescriptor; typedef const TDescriptor* pDescriptor; static const int L = 32;
    static void meanValue(const std::vector<pDescriptor>& descriptors, TDescriptor& mean); static int distance(const TDescriptor& a, const TDescriptor& b);
    static std::string toString(const TDescriptor& a); static void fromString(TDescriptor& a, const std::string& s); };
}
