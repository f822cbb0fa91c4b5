// voc_io.hpp -- the DBoW2 vocabulary file of the reference as flat arrays for asd_voc_load (no OpenCV).
//
// System.cc:116-121 loads the vocabulary with TemplatedVocabulary::load(filename) = cv::FileStorage YAML written by
// TemplatedVocabulary::save (src/dbow2/include/TemplatedVocabulary.h:1361-1452):
//   %YAML:1.0
//   vocabulary:
//      k: 10
//      L: 6
//      scoringType: 0
//      weightingType: 0
//      nodes:
//         - { nodeId:1, parentId:0, weight:0., descriptor:"f0 f1 ... f127 " }
//         ...
//      words:
//         - { wordId:0, nodeId:19 }
//         ...
// load() (:1455-1497) sizes m_nodes by the node count, appends every node to its parent's children in FILE order and
// takes word ids from the words list; FSift::fromString (FSift.cpp:117-133) reads 128 floats from the descriptor string.
// ReadVocabulary reproduces exactly that into the layout asd_voc_load takes (children CSR in file order, word_id = -1 for
// inner nodes).  The scanner is tolerant about whitespace and line wraps inside the flow maps, which is all cv::FileStorage
// varies.  WriteVocabulary emits the same shape (tests, synthetic vocabularies).  The vocabulary file itself is not part
// of the reference tree, so this reader is checked by round trips only.
#pragma once
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <sstream>
#include <string>
#include <vector>

namespace asd {

struct VocabularyArrays {
  int k = 0, L = 0, scoring = 0, weighting = 0;
  int n_nodes = 0;                 // including the root (node 0)
  std::vector<int32_t> child_start, child_ids, word_id;
  std::vector<double> weight;
  std::vector<float> desc;         // [n_nodes][128], row 0 unused
};

namespace vocio {
inline bool find_key(const std::string& s, size_t& pos, const char* key, size_t limit = std::string::npos) {
  const size_t p = s.find(key, pos);
  if (p == std::string::npos || p >= limit) return false;
  pos = p + strlen(key);
  return true;
}
inline double number_after(const std::string& s, size_t& pos) {
  while (pos < s.size() && (s[pos] == ' ' || s[pos] == ':' || s[pos] == '\n' || s[pos] == '\r' || s[pos] == '\t')) ++pos;
  char* end = nullptr;
  const double v = strtod(s.c_str() + pos, &end);
  pos = (size_t)(end - s.c_str());
  return v;
}
}  // namespace vocio

inline bool ReadVocabulary(const std::string& path, VocabularyArrays& V, std::string* error = nullptr) {
  auto fail = [&](const char* m) { if (error) *error = m; return false; };
  std::ifstream f(path.c_str(), std::ios::binary);
  if (!f.is_open()) return fail("cannot open the vocabulary file");
  std::stringstream ss;
  ss << f.rdbuf();
  const std::string s = ss.str();
  size_t pos = 0;
  if (!vocio::find_key(s, pos, "vocabulary")) return fail("no `vocabulary` node");
  const size_t nodes_at = s.find("nodes", pos), words_at = s.find("words", pos);
  if (nodes_at == std::string::npos || words_at == std::string::npos || words_at < nodes_at) return fail("no nodes / words lists");
  size_t p = pos;
  if (!vocio::find_key(s, p, "k:", nodes_at)) return fail("no k");
  V.k = (int)vocio::number_after(s, p);
  p = pos;
  if (!vocio::find_key(s, p, "L:", nodes_at)) return fail("no L");
  V.L = (int)vocio::number_after(s, p);
  p = pos;
  if (!vocio::find_key(s, p, "scoringType:", nodes_at)) return fail("no scoringType");
  V.scoring = (int)vocio::number_after(s, p);
  p = pos;
  if (!vocio::find_key(s, p, "weightingType:", nodes_at)) return fail("no weightingType");
  V.weighting = (int)vocio::number_after(s, p);
  // pass 1 over the node list
  struct Rec { int id, parent; double w; size_t d0, d1; };
  std::vector<Rec> recs;
  p = nodes_at;
  int max_id = 0;
  while (vocio::find_key(s, p, "nodeId", words_at)) {
    Rec r;
    r.id = (int)vocio::number_after(s, p);
    if (!vocio::find_key(s, p, "parentId", words_at)) return fail("node without parentId");
    r.parent = (int)vocio::number_after(s, p);
    if (!vocio::find_key(s, p, "weight", words_at)) return fail("node without weight");
    r.w = vocio::number_after(s, p);
    if (!vocio::find_key(s, p, "descriptor", words_at)) return fail("node without descriptor");
    const size_t q0 = s.find('"', p);
    const size_t q1 = q0 == std::string::npos ? q0 : s.find('"', q0 + 1);
    if (q1 == std::string::npos || q1 > words_at) return fail("unterminated descriptor string");
    r.d0 = q0 + 1; r.d1 = q1;
    p = q1 + 1;
    if (r.id <= 0 || r.parent < 0) return fail("node / parent id out of range");
    if (r.id > max_id) max_id = r.id;
    recs.push_back(r);
  }
  const int n = (int)recs.size() + 1;  // m_nodes.resize(fn.size() + 1)
  if (max_id >= n) return fail("node ids are not dense");
  V.n_nodes = n;
  V.weight.assign(n, 0.0);
  V.word_id.assign(n, -1);
  V.desc.assign((size_t)n * 128, 0.f);
  std::vector<int> nchild(n, 0), parent(n, 0);
  for (const Rec& r : recs) {
    if (r.parent >= n) return fail("parent id out of range");
    V.weight[r.id] = r.w;
    parent[r.id] = r.parent;
    ++nchild[r.parent];
    const char* c = s.c_str() + r.d0;
    char* end = nullptr;
    for (int k = 0; k < 128; ++k) {  // FSift::fromString: stringstream >> float, 128 times
      V.desc[(size_t)r.id * 128 + k] = strtof(c, &end);
      if (end == c || (size_t)(end - s.c_str()) > r.d1) return fail("descriptor with fewer than 128 values");
      c = end;
    }
  }
  V.child_start.assign(n + 1, 0);
  for (int i = 0; i < n; ++i) V.child_start[i + 1] = V.child_start[i] + nchild[i];
  V.child_ids.assign(V.child_start[n], 0);
  std::vector<int> cur(V.child_start.begin(), V.child_start.end() - 1);
  for (const Rec& r : recs) V.child_ids[cur[r.parent]++] = r.id;  // children.push_back(nid) in file order
  // words
  p = words_at;
  while (vocio::find_key(s, p, "wordId")) {
    const int wid = (int)vocio::number_after(s, p);
    if (!vocio::find_key(s, p, "nodeId")) return fail("word without nodeId");
    const int nid = (int)vocio::number_after(s, p);
    if (nid <= 0 || nid >= n || wid < 0) return fail("word entry out of range");
    V.word_id[nid] = wid;
  }
  return true;
}

// the shape TemplatedVocabulary::save writes (node order: file order = ascending id here)
inline bool WriteVocabulary(const std::string& path, const VocabularyArrays& V) {
  FILE* f = fopen(path.c_str(), "w");
  if (!f) return false;
  fprintf(f, "%%YAML:1.0\n---\nvocabulary:\n   k: %d\n   L: %d\n   scoringType: %d\n   weightingType: %d\n   nodes:\n", V.k, V.L, V.scoring, V.weighting);
  std::vector<int> parent(V.n_nodes, 0);
  for (int i = 0; i < V.n_nodes; ++i)
    for (int c = V.child_start[i]; c < V.child_start[i + 1]; ++c) parent[V.child_ids[c]] = i;
  // depth-first like save(): a stack of parents, children in stored order
  std::vector<int> stack{0};
  while (!stack.empty()) {
    const int pid = stack.back();
    stack.pop_back();
    for (int c = V.child_start[pid]; c < V.child_start[pid + 1]; ++c) {
      const int id = V.child_ids[c];
      fprintf(f, "      - { nodeId:%d, parentId:%d, weight:%.17g,\n          descriptor:\"", id, pid, V.weight[id]);
      for (int k = 0; k < 128; ++k) fprintf(f, "%.9g ", V.desc[(size_t)id * 128 + k]);
      fprintf(f, "\" }\n");
      if (V.child_start[id + 1] > V.child_start[id]) stack.push_back(id);
    }
  }
  fprintf(f, "   words:\n");
  std::vector<int> node_of_word;
  for (int i = 0; i < V.n_nodes; ++i)
    if (V.word_id[i] >= 0) {
      if ((int)node_of_word.size() <= V.word_id[i]) node_of_word.resize(V.word_id[i] + 1, -1);
      node_of_word[V.word_id[i]] = i;
    }
  for (size_t w = 0; w < node_of_word.size(); ++w) fprintf(f, "      - { wordId:%zu, nodeId:%d }\n", w, node_of_word[w]);
  fclose(f);
  return true;
}

}  // namespace asd
