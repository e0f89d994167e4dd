// DECLARATION-ONLY stand-in for DBoW2's public containers (tests/cpp/thirdparty_stubs/opencv2/core/core.hpp explains): DBoW2/BowVector.h
#pragma once
#include <map>
#include <vector>
namespace DBoW2 {
typedef unsigned int WordId; typedef double WordValue; typedef unsigned int NodeId;
enum LNorm { L1, L2 }; enum WeightingType { TF_IDF, TF, IDF, BINARY }; enum ScoringType { L1_NORM, L2_NORM, CHI_SQUARE, KL, BHATTACHARYYA, DOT_PRODUCT };
class BowVector : public std::map<WordId, WordValue> { public: BowVector(); ~BowVector(); void addWeight(WordId id, WordValue v); void addIfNotExist(WordId id, WordValue v); void normalize(LNorm norm_type); };
}
