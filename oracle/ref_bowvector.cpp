// ORACLE (test infrastructure) -- C shim over the REFERENCE's own DBoW2::BowVector, compiled from the source where it lies
// (/root/reference/pose_graph/src/ThirdParty/DBoW/BowVector.{h,cpp}: the only arithmetic translation unit of the reference that builds here
// without OpenCV / Eigen / boost / ROS) into oracle/_ref/ by `make -C oracle ref`.  It pins the bag-of-words vector half of oracle/bow.cpp
// (TemplatedVocabulary::transform(features, v): addWeight / addIfNotExist per feature, then BowVector::normalize(L1)) against reference code
// run in this container; tests/test_oracle_bow_cpu.py.  Nothing of the reference is copied: this file only calls its class.
#include "BowVector.h"

extern "C" int oref_bow_vector(int n, const int *word, const double *weight, int weighting, int cap, int *out_id, double *out_value) {
    DBoW2::BowVector v;
    const bool tf = weighting == DBoW2::TF_IDF || weighting == DBoW2::TF;
    for (int i = 0; i < n; i++) {
        if (!(weight[i] > 0)) continue;                       // "not stopped" (TemplatedVocabulary.h:1094, :1113)
        if (tf) v.addWeight((DBoW2::WordId)word[i], weight[i]);
        else v.addIfNotExist((DBoW2::WordId)word[i], weight[i]);
    }
    v.normalize(DBoW2::L1);                                   // L1Scoring::mustNormalize
    int m = 0;
    for (DBoW2::BowVector::const_iterator it = v.begin(); it != v.end(); ++it, ++m)
        if (m < cap) { out_id[m] = (int)it->first; out_value[m] = it->second; }
    return m;
}
