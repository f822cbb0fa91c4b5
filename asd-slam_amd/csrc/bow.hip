// bow.hip -- DBoW2 vocabulary transform for 128-D float descriptors.
//
// Replaces Frame::ComputeBoW / KeyFrame::ComputeBoW (reference src/vslam/src/Frame.cc:289-296) =
// TemplatedVocabulary<FSift>::transform(features, BowVector, FeatureVector, levelsup = 4)
// (src/dbow2/include/TemplatedVocabulary.h:1125-1197): every descriptor walks the k-ary tree from the root,
// at each level taking the child with the smallest FSift::distance (src/dbow2/DBoW2/FSift.cpp:86-101).
//
// Device layout: the vocabulary is one flat node table resident in HBM (descriptor rows [n_nodes][128] f32,
// CSR child lists, word ids, weights).  The descent is independent per descriptor; a group of G lanes (G = next
// power of two >= the branching factor) owns one descriptor, lane c evaluates child c, and a G-wide butterfly
// picks the winner with the reference's tie rule (first child wins).  FSift::distance accumulates f32 squares
// into a double in ascending index order; the lane keeps exactly that chain, so word and node ids are bit-exact.
// The BowVector / FeatureVector maps are then assembled on the host in std::map order.
#include <algorithm>
#include <cmath>
#include <cstring>
#include <numeric>

#include "ctx.h"

namespace {

struct VocDev {
  const float* desc;        // [n_nodes][128]
  const int* child_start;   // [n_nodes + 1]
  const int* child_ids;
  int max_depth;
};

template <int G>
__global__ __launch_bounds__(256) void k_bow_descend(VocDev V, const float* __restrict__ desc, int n, int nid_level,
                                                    int* __restrict__ leaf_out, int* __restrict__ node_out) {
  constexpr int FPB = 256 / G;  // descriptors per workgroup
  __shared__ float s_feat[FPB][128];
  const int g = threadIdx.x / G, c = threadIdx.x % G;
  const int f0 = blockIdx.x * FPB;
  for (int t = threadIdx.x; t < FPB * 128; t += 256) {
    const int ff = f0 + t / 128;
    s_feat[t / 128][t % 128] = ff < n ? desc[(size_t)ff * 128 + t % 128] : 0.f;
  }
  asd_syncthreads();
  const int f = f0 + g;
  if (f >= n) return;  // whole groups leave together; the butterflies below stay inside one group
  const float* q = s_feat[g];
  int cur = 0, nid = nid_level <= 0 ? 0 : -1;
  for (int level = 1; level <= V.max_depth; ++level) {
    const int cs = V.child_start[cur], nc = V.child_start[cur + 1] - cs;
    if (nc == 0) break;  // leaf
    double d = INFINITY;
    int id = -1, bc = c;
    if (c < nc) {
      id = V.child_ids[cs + c];
      const float4* nd = reinterpret_cast<const float4*>(V.desc + (size_t)id * 128);
      double sqd = 0.;
#pragma unroll 4
      for (int i = 0; i < 32; ++i) {
        const float4 b = nd[i];
        const float4 a = *reinterpret_cast<const float4*>(q + 4 * i);
        const float d0 = a.x - b.x, d1 = a.y - b.y, d2 = a.z - b.z, d3 = a.w - b.w;
        sqd += (double)(d0 * d0);
        sqd += (double)(d1 * d1);
        sqd += (double)(d2 * d2);
        sqd += (double)(d3 * d3);
      }
      // `d < best_d` with a NaN: child 0 seeds best_d and is then never replaced, any other NaN child never wins
      d = (sqd != sqd) ? (c == 0 ? -INFINITY : INFINITY) : sqd;
    }
#pragma unroll
    for (int off = G / 2; off > 0; off >>= 1) {
      const double od = __shfl_xor(d, off, G);
      const int oc = __shfl_xor(bc, off, G);
      const int oid = __shfl_xor(id, off, G);
      if (od < d || (od == d && oc < bc)) { d = od; bc = oc; id = oid; }
    }
    cur = id;
    if (level == nid_level) nid = cur;
  }
  if (c == 0) {
    leaf_out[f] = cur;
    node_out[f] = nid < 0 ? cur : nid;  // leaf above nid_level: the reference leaves *nid unset; report the leaf
  }
}

struct BowState {
  bool loaded = false;
  int n_nodes = 0, k = 0, L = 0, weighting = 0, scoring = 0, max_children = 0, max_depth = 0;
  float* d_desc = nullptr;
  int *d_child_start = nullptr, *d_child_ids = nullptr;
  std::vector<int> word_id;      // host copies: the assembly runs on the host
  std::vector<double> weight;
  // per-call buffers
  int cap = 0;
  int *d_out = nullptr, *h_out = nullptr;  // [2*cap] leaf | node (h_* pinned)
  float *d_q = nullptr, *h_q = nullptr;    // [cap][128] staging for host descriptors
};

BowState* bstate(asd_ctx* ctx) {
  if (!ctx->bow) ctx->bow = new BowState();
  return static_cast<BowState*>(ctx->bow);
}

void free_voc(BowState* b) {
  if (b->d_desc) (void)hipFree(b->d_desc);
  if (b->d_child_start) (void)hipFree(b->d_child_start);
  if (b->d_child_ids) (void)hipFree(b->d_child_ids);
  b->d_desc = nullptr; b->d_child_start = nullptr; b->d_child_ids = nullptr;
  b->loaded = false;
}

int ensure(asd_ctx* ctx, BowState* b, int n, bool need_q) {
  if (n > b->cap) {
    const int cap = std::max(n * 3 / 2, 4096);
    if (b->d_out) { (void)hipFree(b->d_out); (void)hipHostFree(b->h_out); }
    if (b->d_q) { (void)hipFree(b->d_q); (void)hipHostFree(b->h_q); b->d_q = nullptr; b->h_q = nullptr; }
    b->cap = 0;
    ASD_HIP_CHECK(ctx, hipMalloc(&b->d_out, (size_t)2 * cap * sizeof(int)));
    ASD_HIP_CHECK(ctx, hipHostMalloc(&b->h_out, (size_t)2 * cap * sizeof(int)));
    b->cap = cap;
  }
  if (need_q && !b->d_q) {
    ASD_HIP_CHECK(ctx, hipMalloc(&b->d_q, (size_t)b->cap * 128 * sizeof(float)));
    ASD_HIP_CHECK(ctx, hipHostMalloc(&b->h_q, (size_t)b->cap * 128 * sizeof(float)));
  }
  return ASD_OK;
}

template <int G>
void launch(hipStream_t st, const VocDev& V, const float* d_desc, int n, int nid_level, int* d_out) {
  constexpr int FPB = 256 / G;
  hipLaunchKernelGGL(k_bow_descend<G>, dim3((n + FPB - 1) / FPB), dim3(256), 0, st, V, d_desc, n, nid_level, d_out, d_out + n);
}

// leaf / node per descriptor into b->h_out[0..n) / [n..2n)
int descend(asd_ctx* ctx, BowState* b, int32_t slot, const float* desc, int n, int levelsup) {
  const float* d_src = nullptr;
  if (desc) {
    int rc = ensure(ctx, b, n, true);
    if (rc != ASD_OK) return rc;
  } else {
    if (slot < 0 || slot >= ASD_MAX_FRAMES || !ctx->frames[slot].d_desc || ctx->frames[slot].n != n) {
      ctx->set_error("bow: frame slot %d does not hold %d descriptors", slot, n);
      return ASD_ERR_INVALID;
    }
    int rc = ensure(ctx, b, n, false);
    if (rc != ASD_OK) return rc;
    d_src = ctx->frames[slot].d_desc;
  }
  hipStream_t st = ctx->stream;
  if (desc) {
    memcpy(b->h_q, desc, (size_t)n * 128 * sizeof(float));
    ASD_HIP_CHECK(ctx, hipMemcpyAsync(b->d_q, b->h_q, (size_t)n * 128 * sizeof(float), hipMemcpyHostToDevice, st));
    d_src = b->d_q;
  }
  const VocDev V{b->d_desc, b->d_child_start, b->d_child_ids, b->max_depth};
  const int nid_level = b->L - levelsup;
  if (b->max_children <= 4) launch<4>(st, V, d_src, n, nid_level, b->d_out);
  else if (b->max_children <= 8) launch<8>(st, V, d_src, n, nid_level, b->d_out);
  else if (b->max_children <= 16) launch<16>(st, V, d_src, n, nid_level, b->d_out);
  else if (b->max_children <= 32) launch<32>(st, V, d_src, n, nid_level, b->d_out);
  else launch<64>(st, V, d_src, n, nid_level, b->d_out);
  ASD_HIP_CHECK(ctx, hipGetLastError());
  ASD_HIP_CHECK(ctx, hipMemcpyAsync(b->h_out, b->d_out, (size_t)2 * n * sizeof(int), hipMemcpyDeviceToHost, st));
  ASD_HIP_CHECK(ctx, hipStreamSynchronize(st));
  return ASD_OK;
}

}  // namespace

void bow_free(asd_ctx* ctx) {
  if (!ctx->bow) return;
  BowState* b = static_cast<BowState*>(ctx->bow);
  free_voc(b);
  if (b->d_out) { (void)hipFree(b->d_out); (void)hipHostFree(b->h_out); }
  if (b->d_q) { (void)hipFree(b->d_q); (void)hipHostFree(b->h_q); }
  delete b;
  ctx->bow = nullptr;
}

extern "C" {

int asd_voc_load(asd_ctx* ctx, int32_t n_nodes, int32_t k, int32_t L, int32_t weighting, int32_t scoring,
                 const int32_t* child_start, const int32_t* child_ids, const double* weight, const int32_t* word_id,
                 const float* desc) {
  if (!ctx) return ASD_ERR_INVALID;
  if (n_nodes < 2 || !child_start || !child_ids || !weight || !word_id || !desc || weighting < 0 || weighting > 3 || scoring < 0 ||
      scoring > 5 || child_start[0] != 0) {
    ctx->set_error("asd_voc_load: invalid argument");
    return ASD_ERR_INVALID;
  }
  // The descent kernel follows child links until it meets a leaf: insist on a proper tree rooted at node 0
  // (every other node reached exactly once), so that every walk ends after at most max_depth steps.
  std::vector<int> depth(n_nodes, -1);
  std::vector<int> queue;
  queue.reserve(n_nodes);
  queue.push_back(0);
  depth[0] = 0;
  int max_children = 0, max_depth = 0;
  for (size_t qi = 0; qi < queue.size(); ++qi) {
    const int u = queue[qi];
    const int cs = child_start[u], ce = child_start[u + 1];
    if (cs < 0 || ce < cs || ce > n_nodes - 1) { ctx->set_error("asd_voc_load: child list of node %d out of range", u); return ASD_ERR_INVALID; }
    max_children = std::max(max_children, ce - cs);
    if (ce == cs && (word_id[u] < 0 || u == 0)) { ctx->set_error("asd_voc_load: leaf %d has no word id", u); return ASD_ERR_INVALID; }
    for (int t = cs; t < ce; ++t) {
      const int v = child_ids[t];
      if (v <= 0 || v >= n_nodes || depth[v] >= 0) { ctx->set_error("asd_voc_load: node %d is not a tree child of %d", v, u); return ASD_ERR_INVALID; }
      depth[v] = depth[u] + 1;
      max_depth = std::max(max_depth, depth[v]);
      queue.push_back(v);
    }
  }
  if ((int)queue.size() != n_nodes) { ctx->set_error("asd_voc_load: %d of %d nodes unreachable from the root", n_nodes - (int)queue.size(), n_nodes); return ASD_ERR_INVALID; }
  if (max_children > 64) { ctx->set_error("asd_voc_load: branching factor %d > 64", max_children); return ASD_ERR_CAPACITY; }
  (void)hipSetDevice(ctx->cfg.device);
  BowState* b = bstate(ctx);
  free_voc(b);
  const int n_child = child_start[n_nodes];
  ASD_HIP_CHECK(ctx, hipMalloc(&b->d_desc, (size_t)n_nodes * 128 * sizeof(float)));
  ASD_HIP_CHECK(ctx, hipMalloc(&b->d_child_start, (size_t)(n_nodes + 1) * sizeof(int)));
  ASD_HIP_CHECK(ctx, hipMalloc(&b->d_child_ids, (size_t)std::max(n_child, 1) * sizeof(int)));
  ASD_HIP_CHECK(ctx, hipMemcpy(b->d_desc, desc, (size_t)n_nodes * 128 * sizeof(float), hipMemcpyHostToDevice));
  ASD_HIP_CHECK(ctx, hipMemcpy(b->d_child_start, child_start, (size_t)(n_nodes + 1) * sizeof(int), hipMemcpyHostToDevice));
  ASD_HIP_CHECK(ctx, hipMemcpy(b->d_child_ids, child_ids, (size_t)n_child * sizeof(int), hipMemcpyHostToDevice));
  b->word_id.assign(word_id, word_id + n_nodes);
  b->weight.assign(weight, weight + n_nodes);
  b->n_nodes = n_nodes; b->k = k; b->L = L; b->weighting = weighting; b->scoring = scoring;
  b->max_children = max_children; b->max_depth = max_depth;
  b->loaded = true;
  return ASD_OK;
}

int asd_bow_descend(asd_ctx* ctx, int32_t slot, const float* desc, int32_t n, int32_t levelsup, int32_t* word, int32_t* node,
                    double* weight) {
  if (!ctx || n < 0) return ASD_ERR_INVALID;
  BowState* b = bstate(ctx);
  if (!b->loaded) { ctx->set_error("asd_bow_descend: no vocabulary loaded"); return ASD_ERR_NO_WEIGHTS; }
  if (n == 0) return ASD_OK;
  (void)hipSetDevice(ctx->cfg.device);
  int rc = descend(ctx, b, slot, desc, n, levelsup);
  if (rc != ASD_OK) return rc;
  for (int i = 0; i < n; ++i) {
    const int leaf = b->h_out[i];
    if (word) word[i] = b->word_id[leaf];
    if (weight) weight[i] = b->weight[leaf];
    if (node) node[i] = b->h_out[n + i];
  }
  return ASD_OK;
}

int asd_compute_bow(asd_ctx* ctx, int32_t slot, const float* desc, int32_t n, int32_t levelsup, int32_t* bow_id, double* bow_val,
                    int32_t* n_words, int32_t* fv_node, int32_t* fv_start, int32_t* fv_idx, int32_t* n_fv_nodes) {
  if (!ctx || n < 0 || !n_words || !n_fv_nodes || !fv_start || (n > 0 && (!bow_id || !bow_val || !fv_node || !fv_idx)))
    return ASD_ERR_INVALID;
  BowState* b = bstate(ctx);
  if (!b->loaded) { ctx->set_error("asd_compute_bow: no vocabulary loaded"); return ASD_ERR_NO_WEIGHTS; }
  *n_words = 0; *n_fv_nodes = 0; fv_start[0] = 0;
  if (n == 0) return ASD_OK;
  (void)hipSetDevice(ctx->cfg.device);
  int rc = descend(ctx, b, slot, desc, n, levelsup);
  if (rc != ASD_OK) return rc;
  // features that survive the stop-word test (w > 0), in feature order
  std::vector<int> feat;
  feat.reserve(n);
  for (int i = 0; i < n; ++i)
    if (b->weight[b->h_out[i]] > 0) feat.push_back(i);
  // BowVector: std::map<WordId, WordValue> with addWeight (TF_IDF / TF: += in feature order) or addIfNotExist
  std::vector<int> by_word(feat);
  std::stable_sort(by_word.begin(), by_word.end(), [&](int x, int y) { return b->word_id[b->h_out[x]] < b->word_id[b->h_out[y]]; });
  const bool accumulate = b->weighting == 0 || b->weighting == 1;
  int nw = 0;
  for (size_t t = 0; t < by_word.size();) {
    const int leaf = b->h_out[by_word[t]];
    const int wid = b->word_id[leaf];
    double v = b->weight[leaf];
    size_t u = t + 1;
    for (; u < by_word.size() && b->word_id[b->h_out[by_word[u]]] == wid; ++u)
      if (accumulate) v += b->weight[b->h_out[by_word[u]]];
    bow_id[nw] = wid; bow_val[nw] = v; ++nw;
    t = u;
  }
  const bool must = b->scoring != 5;  // ScoringObject.h:72-89: only DotProductScoring skips normalisation
  if (accumulate && nw > 0 && !must) {
    const double nd = nw;
    for (int t = 0; t < nw; ++t) bow_val[t] /= nd;
  }
  if (must) {  // BowVector::normalize, ascending word id
    double norm = 0.0;
    if (b->scoring != 1) { for (int t = 0; t < nw; ++t) norm += fabs(bow_val[t]); }
    else { for (int t = 0; t < nw; ++t) norm += bow_val[t] * bow_val[t]; norm = sqrt(norm); }
    if (norm > 0.0) for (int t = 0; t < nw; ++t) bow_val[t] /= norm;
  }
  *n_words = nw;
  // FeatureVector: std::map<NodeId, vector<feature index>>
  std::vector<int>& by_node = feat;
  std::stable_sort(by_node.begin(), by_node.end(), [&](int x, int y) { return b->h_out[n + x] < b->h_out[n + y]; });
  int nn = 0;
  for (size_t t = 0; t < by_node.size(); ++t) {
    const int nid = b->h_out[n + by_node[t]];
    if (t == 0 || nid != fv_node[nn - 1]) { fv_node[nn] = nid; fv_start[nn] = (int)t; ++nn; }
    fv_idx[t] = by_node[t];
  }
  fv_start[nn] = (int)by_node.size();
  *n_fv_nodes = nn;
  return ASD_OK;
}

}  // extern "C"
