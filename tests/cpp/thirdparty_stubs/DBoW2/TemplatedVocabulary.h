// DECLARATION-ONLY stand-in: DBoW2/TemplatedVocabulary.h (hySLAM's "modified DBoW2": loadFromTextFile / loadFromBinaryFile as ORB-SLAM2's fork has them)
#pragma once
#include "BowVector.h"
#include "FeatureVector.h"
#include <string>
#include <vector>
namespace DBoW2 {
template <class TDescriptor, class F> class TemplatedVocabulary {
public:
    TemplatedVocabulary(int k = 10, int L = 5, WeightingType weighting = TF_IDF, ScoringType scoring = L1_NORM); TemplatedVocabulary(const std::string& filename); virtual ~TemplatedVocabulary();
    virtual unsigned int size() const; virtual bool empty() const;
    virtual void transform(const std::vector<TDescriptor>& features, BowVector& v) const;
    virtual void transform(const std::vector<TDescriptor>& features, BowVector& v, FeatureVector& fv, int levelsup) const;
    double score(const BowVector& a, const BowVector& b) const;
    bool loadFromTextFile(const std::string& filename); bool loadFromBinaryFile(const std::string& filename); void saveToTextFile(const std::string& filename) const; void saveToBinaryFile(const std::string& filename) const;
    void load(const std::string& filename); void save(const std::string& filename) const;
};
}
