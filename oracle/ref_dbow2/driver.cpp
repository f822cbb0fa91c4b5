// driver.cpp -- flat-array entry into the reference's own DBoW2::BowVector / DBoW2::FeatureVector (compiled in place by
// the Makefile next to this file).  TEST INFRASTRUCTURE ONLY.
//
// The loop around them is TemplatedVocabulary<..>::transform(features, v, fv, levelsup)
// (src/dbow2/include/TemplatedVocabulary.h:1125-1197) with the per-feature descent -- the part that needs OpenCV -- replaced
// by its results handed in as arrays: word id, word weight and node id per feature.  What this pins: addWeight /
// addIfNotExist accumulation order, std::map iteration order, normalize() (L1: sum of fabs, L2: sqrt of the sum of squares,
// division element by element) and addFeature's per-node lists.
#include <cstdint>

#include "BowVector.h"
#include "FeatureVector.h"

extern "C" {

// weighting: DBoW2::WeightingType {TF_IDF, TF, IDF, BINARY}; must_normalize / norm: what ScoringObject::mustNormalize returns
// for the vocabulary's scoring (ScoringObject.h:72-89); returns the number of words, *n_fv = number of nodes
int ref_bow_assemble(int n, const int32_t* word_id, const double* weight, const int32_t* node_id, int weighting,
                     int must_normalize, int norm_l2, int32_t* bow_id, double* bow_val, int32_t* fv_node, int32_t* fv_start,
                     int32_t* fv_idx, int32_t* n_fv) {
  DBoW2::BowVector v;
  DBoW2::FeatureVector fv;
  const bool must = must_normalize != 0;
  const DBoW2::LNorm norm = norm_l2 ? DBoW2::L2 : DBoW2::L1;
  if (weighting == DBoW2::TF || weighting == DBoW2::TF_IDF) {
    for (int i = 0; i < n; ++i) {
      const DBoW2::WordValue w = weight[i];
      if (w > 0) {
        v.addWeight((DBoW2::WordId)word_id[i], w);
        fv.addFeature((DBoW2::NodeId)node_id[i], (unsigned)i);
      }
    }
    if (!v.empty() && !must) {
      const double nd = v.size();
      for (DBoW2::BowVector::iterator vit = v.begin(); vit != v.end(); vit++) vit->second /= nd;
    }
  } else {
    for (int i = 0; i < n; ++i) {
      const DBoW2::WordValue w = weight[i];
      if (w > 0) {
        v.addIfNotExist((DBoW2::WordId)word_id[i], w);
        fv.addFeature((DBoW2::NodeId)node_id[i], (unsigned)i);
      }
    }
  }
  if (must) v.normalize(norm);
  int k = 0;
  for (DBoW2::BowVector::const_iterator it = v.begin(); it != v.end(); ++it, ++k) { bow_id[k] = (int32_t)it->first; bow_val[k] = it->second; }
  int m = 0, pos = 0;
  fv_start[0] = 0;
  for (DBoW2::FeatureVector::const_iterator it = fv.begin(); it != fv.end(); ++it) {
    fv_node[m] = (int32_t)it->first;
    for (size_t f = 0; f < it->second.size(); ++f) fv_idx[pos++] = (int32_t)it->second[f];
    fv_start[++m] = pos;
  }
  *n_fv = m;
  return k;
}

}  // extern "C"
