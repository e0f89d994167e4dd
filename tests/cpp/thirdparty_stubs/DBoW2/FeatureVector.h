// DECLARATION-ONLY stand-in: DBoW2/FeatureVector.h (node id -> indices of the features below that node)
#pragma once
#include "BowVector.h"
namespace DBoW2 {
class FeatureVector : public std::map<NodeId, std::vector<unsigned int> > { public: FeatureVector(); ~FeatureVector(); void addFeature(NodeId id, unsigned int i_feature); };
}
