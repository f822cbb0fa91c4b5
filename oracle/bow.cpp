// oracle/bow.cpp -- CPU restatement of the DBoW2 transform for float descriptors (FSift).
// TEST INFRASTRUCTURE ONLY (see oracle.h).
//
// Follows src/dbow2/include/TemplatedVocabulary.h:1125-1197 (transform(features, BowVector, FeatureVector,
// levelsup)), :1219-1260 (single-feature descent), src/dbow2/DBoW2/FSift.cpp:86-101 (distance: f32 squared
// differences accumulated in a double, ascending index), BowVector.cpp:35-85 (addWeight / addIfNotExist /
// normalize), FeatureVector.cpp:34-48 (addFeature), ScoringObject.h:72-89 (mustNormalize table), as called
// from Frame::ComputeBoW (src/vslam/src/Frame.cc:289-296, levelsup = 4).
// DBoW2 is vendored in the reference; the vocabulary template and FSift need OpenCV to compile (cv::Mat descriptors,
// cv::FileStorage) and the vocabulary file is not part of the tree: the DESCENT is PARITY UNPINNED beyond known-answer
// cases.  BowVector.cpp / FeatureVector.cpp include only the STL: the ASSEMBLY (orc_bow_assemble) is pinned against them
// compiled in place (oracle/ref_dbow2/Makefile -> oracle/_ref/libdbow2_ref.so).
#include <cmath>
#include <cstring>
#include <map>
#include <vector>

#include "oracle.h"

struct orc_vocabulary {
  int k, L, weighting, scoring;  // WeightingType {TF_IDF, TF, IDF, BINARY}, ScoringType {L1, L2, CHI, KL, BHATT, DOT}
  struct Node {
    std::vector<int> children;
    double weight = 0;
    int word_id = -1;
    float desc[128];
    bool isLeaf() const { return children.empty(); }
  };
  std::vector<Node> nodes;
};

namespace {

double fsift_distance(const float* a, const float* b) {  // FSift.cpp:86-101
  double sqd = 0.;
  for (int i = 0; i < 128; i += 4) {
    sqd += (a[i] - b[i]) * (a[i] - b[i]);
    sqd += (a[i + 1] - b[i + 1]) * (a[i + 1] - b[i + 1]);
    sqd += (a[i + 2] - b[i + 2]) * (a[i + 2] - b[i + 2]);
    sqd += (a[i + 3] - b[i + 3]) * (a[i + 3] - b[i + 3]);
  }
  return sqd;
}

// TemplatedVocabulary::transform(feature, word_id, weight, nid, levelsup) (:1219-1260).  The reference leaves
// *nid uninitialised when a leaf is reached above nid_level; that case reports the leaf itself here.
void transform_one(const orc_vocabulary* V, const float* feature, int levelsup, int* word_id, double* weight, int* nid) {
  const int nid_level = V->L - levelsup;
  bool nid_set = false;
  if (nid_level <= 0) { *nid = 0; nid_set = true; }
  int final_id = 0, current_level = 0;
  do {
    ++current_level;
    const std::vector<int>& nodes = V->nodes[final_id].children;
    final_id = nodes[0];
    double best_d = fsift_distance(feature, V->nodes[final_id].desc);
    for (size_t c = 1; c < nodes.size(); ++c) {
      const int id = nodes[c];
      const double d = fsift_distance(feature, V->nodes[id].desc);
      if (d < best_d) { best_d = d; final_id = id; }
    }
    if (current_level == nid_level) { *nid = final_id; nid_set = true; }
  } while (!V->nodes[final_id].isLeaf());
  if (!nid_set) *nid = final_id;
  *word_id = V->nodes[final_id].word_id;
  *weight = V->nodes[final_id].weight;
}

}  // namespace

extern "C" {

orc_vocabulary* orc_voc_create(int n_nodes, int k, int L, int weighting, int scoring, const int32_t* child_start,
                               const int32_t* child_ids, const double* weight, const int32_t* word_id, const float* desc) {
  orc_vocabulary* V = new orc_vocabulary();
  V->k = k; V->L = L; V->weighting = weighting; V->scoring = scoring;
  V->nodes.resize(n_nodes);
  for (int i = 0; i < n_nodes; ++i) {
    auto& N = V->nodes[i];
    N.children.assign(child_ids + child_start[i], child_ids + child_start[i + 1]);
    N.weight = weight[i];
    N.word_id = word_id[i];
    memcpy(N.desc, desc + (size_t)i * 128, sizeof N.desc);
  }
  return V;
}
void orc_voc_destroy(orc_vocabulary* V) { delete V; }

// per-feature descent only
void orc_bow_descend(const orc_vocabulary* V, const float* desc, int n, int levelsup, int32_t* word, int32_t* node, double* weight) {
  for (int i = 0; i < n; ++i) {
    int w, nd; double wt;
    transform_one(V, desc + (size_t)i * 128, levelsup, &w, &wt, &nd);
    word[i] = w; node[i] = nd; weight[i] = wt;
  }
}

// The assembly half of TemplatedVocabulary::transform(features, v, fv, levelsup) (:1141-1197) on the per-feature results
// of the descent (word id, weight, node id): BowVector::addWeight / addIfNotExist (BowVector.cpp:35-60), the !must division
// (:1163-1169), BowVector::normalize (:64-87) and FeatureVector::addFeature (FeatureVector.cpp:34-48).  PINNED against the
// reference's own BowVector.cpp / FeatureVector.cpp compiled in place (oracle/ref_dbow2, tests/golden/bow_golden.npz).
int orc_bow_assemble(int n, const int32_t* word_id, const double* weight, const int32_t* node_id, int weighting, int scoring,
                     int32_t* bow_id, double* bow_val, int32_t* fv_node, int32_t* fv_start, int32_t* fv_idx, int32_t* n_fv) {
  std::map<unsigned, double> v;
  std::map<unsigned, std::vector<unsigned>> fv;
  const bool must = scoring != 5;      // DotProductScoring is the only one that does not normalise (ScoringObject.h:72-89)
  const bool l2 = scoring == 1;        // L2Scoring -> L2, all others L1
  if (weighting == 0 || weighting == 1) {  // TF_IDF, TF
    for (int i = 0; i < n; ++i) {
      const double w = weight[i];
      if (w > 0) {
        const unsigned id = (unsigned)word_id[i];
        auto it = v.lower_bound(id);
        if (it != v.end() && !(v.key_comp()(id, it->first))) it->second += w;
        else v.insert(it, std::make_pair(id, w));
        fv[(unsigned)node_id[i]].push_back(i);
      }
    }
    if (!v.empty() && !must) {
      const double nd = v.size();
      for (auto& e : v) e.second /= nd;
    }
  } else {  // IDF, BINARY
    for (int i = 0; i < n; ++i) {
      const double w = weight[i];
      if (w > 0) {
        const unsigned id = (unsigned)word_id[i];
        if (v.find(id) == v.end()) v[id] = w;
        fv[(unsigned)node_id[i]].push_back(i);
      }
    }
  }
  if (must) {  // BowVector::normalize
    double norm = 0.0;
    if (!l2) for (auto& e : v) norm += fabs(e.second);
    else { for (auto& e : v) norm += e.second * e.second; norm = sqrt(norm); }
    if (norm > 0.0) for (auto& e : v) e.second /= norm;
  }
  int k = 0;
  for (auto& e : v) { bow_id[k] = (int)e.first; bow_val[k] = e.second; ++k; }
  int m = 0, pos = 0;
  fv_start[0] = 0;
  for (auto& e : fv) {
    fv_node[m] = (int)e.first;
    for (unsigned f : e.second) fv_idx[pos++] = (int)f;
    fv_start[++m] = pos;
  }
  *n_fv = m;
  return k;
}

// TemplatedVocabulary::transform(features, v, fv, levelsup) (:1125-1197): descent per feature, then the assembly above.
// Outputs: BowVector as (bow_id ascending, bow_val), FeatureVector as CSR (fv_node ascending, fv_start, fv_idx); returns the
// number of words, *n_fv nodes.
int orc_bow_transform(const orc_vocabulary* V, const float* desc, int n, int levelsup, int32_t* bow_id, double* bow_val,
                      int32_t* fv_node, int32_t* fv_start, int32_t* fv_idx, int32_t* n_fv) {
  if (V->nodes.size() <= 1) { fv_start[0] = 0; *n_fv = 0; return 0; }   // empty(): v and fv stay cleared (:1134-1137)
  std::vector<int32_t> word(n), node(n);
  std::vector<double> weight(n);
  for (int i = 0; i < n; ++i) {
    int id, nid; double w;
    transform_one(V, desc + (size_t)i * 128, levelsup, &id, &w, &nid);
    word[i] = id; node[i] = nid; weight[i] = w;
  }
  return orc_bow_assemble(n, word.data(), weight.data(), node.data(), V->weighting, V->scoring, bow_id, bow_val, fv_node, fv_start,
                          fv_idx, n_fv);
}

}  // extern "C"
