// matcher.hip -- Frame grid (G1) and the per-frame ORBmatcher searches (M0-M2, M4 init, M5).
//
// Reference (src/vslam/src): Frame.cc:123-138,219-286 (grid), Frame.cc:160-217 (isInFrustum),
// ORBmatcher.cc:1629-1650 (DescriptorDistance), :44-122, :1318-1452, :416-531 (searches),
// :1584-1625 (ComputeThreeMaxima), MapPoint.cc:271-338 (ComputeDistinctiveDescriptors).
//
// Split of work.  The reference's searches are order-dependent: a keypoint claimed by an earlier
// map point is skipped by later ones, best/second-best use strict '<' so the first candidate in
// grid order wins ties.  What is data parallel is DescriptorDistance itself, which the
// reference calls ~10^4-10^5 times per frame with two Mat conversions and two heap vectors per
// call.  So: the host walks the 64x48 grid (a few thousand window queries over <1 keypoint per
// cell) and emits (query, candidate) pairs in the reference's visiting order; one kernel
// evaluates every pair's squared L2 in the reference's exact summation order (sequential f32,
// no FMA: -ffp-contract=off), reading descriptors that are already resident in HBM; the host
// then replays the claim / ratio / rotation-histogram logic over the distances.
// The all-pairs matrix (asd_dist_matrix) uses the same exact summation, tiled through LDS.
#include <algorithm>
#include <cmath>
#include <cstring>

#include "ctx.h"

namespace {
constexpr float TH_HIGH = 1.5f, TH_LOW = 0.5f;  // ORBmatcher.cc:37-38
constexpr int HISTO = ASD_HISTO_LENGTH;
constexpr int GC = ASD_GRID_COLS, GR = ASD_GRID_ROWS;

// one lane per (query, candidate) pair; each lane streams two 512-B descriptor rows.
__global__ __launch_bounds__(256) void k_pair_dist(const float* __restrict__ qdesc, const float* __restrict__ cdesc,
                                                   const int2* __restrict__ pairs, int npairs,
                                                   float* __restrict__ out) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= npairs) return;
  const int2 p = pairs[i];
  const float4* a = reinterpret_cast<const float4*>(qdesc + (size_t)p.x * 128);
  const float4* b = reinterpret_cast<const float4*>(cdesc + (size_t)p.y * 128);
  float sqd = 0.f;
#pragma unroll 8
  for (int k = 0; k < 32; ++k) {
    const float4 x = a[k], y = b[k];
    float d;
    d = x.x - y.x; sqd = sqd + d * d;
    d = x.y - y.y; sqd = sqd + d * d;
    d = x.z - y.z; sqd = sqd + d * d;
    d = x.w - y.w; sqd = sqd + d * d;
  }
  out[i] = sqd;
}

// all-pairs: block = 256 columns (b rows) x 16 a rows; a tile broadcast from LDS, b row in VGPRs.
__global__ __launch_bounds__(256) void k_dist_matrix(const float* __restrict__ a, int na, const float* __restrict__ b,
                                                     int nb, float* __restrict__ out) {
  __shared__ float4 sa[16][32];
  const int j = blockIdx.x * 256 + threadIdx.x;
  const int i0 = blockIdx.y * 16;
  for (int idx = threadIdx.x; idx < 16 * 32; idx += 256) {
    const int r = idx >> 5, c = idx & 31;
    const int ia = min(i0 + r, na - 1);
    sa[r][c] = reinterpret_cast<const float4*>(a + (size_t)ia * 128)[c];
  }
  __syncthreads();
  const float4* brow = reinterpret_cast<const float4*>(b + (size_t)min(j, nb - 1) * 128);
  float acc[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  for (int k = 0; k < 32; ++k) {
    const float4 y = brow[k];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float4 x = sa[r][k];
      float d;
      d = x.x - y.x; acc[r] = acc[r] + d * d;
      d = x.y - y.y; acc[r] = acc[r] + d * d;
      d = x.z - y.z; acc[r] = acc[r] + d * d;
      d = x.w - y.w; acc[r] = acc[r] + d * d;
    }
  }
  if (j < nb)
    for (int r = 0; r < 16; ++r)
      if (i0 + r < na) out[(size_t)(i0 + r) * nb + j] = acc[r];
}

// ---- host helpers -------------------------------------------------------------------------
struct Window { int begin, end; };  // range of the pair list that belongs to one query

inline void three_maxima(const int* cnt, int& ind1, int& ind2, int& ind3) {  // ORBmatcher.cc:1584-1625
  int max1 = 0, max2 = 0, max3 = 0;
  ind1 = ind2 = ind3 = -1;
  for (int i = 0; i < HISTO; i++) {
    const int s = cnt[i];
    if (s > max1) { max3 = max2; max2 = max1; max1 = s; ind3 = ind2; ind2 = ind1; ind1 = i; }
    else if (s > max2) { max3 = max2; max2 = s; ind3 = ind2; ind2 = i; }
    else if (s > max3) { max3 = s; ind3 = i; }
  }
  if (max2 < 0.1f * (float)max1) { ind2 = -1; ind3 = -1; }
  else if (max3 < 0.1f * (float)max1) { ind3 = -1; }
}

inline int rot_bin(float a1, float a2) {  // :1419-1425
  float rot = a1 - a2;
  if (rot < 0.0) rot += 360.0f;
  int bin = (int)std::round(rot * (1.0f / HISTO));
  if (bin == HISTO) bin = 0;
  return bin;
}

// Frame::GetFeaturesInArea (Frame.cc:219-274) over the CSR grid; appends (q, idx) pairs.
void features_in_area(const AsdFrameSlot& F, float x, float y, float r, int minLevel, int maxLevel, int q,
                      std::vector<int2>& pairs) {
  const int nMinCellX = std::max(0, (int)std::floor((x - F.min_x - r) * F.inv_w));
  if (nMinCellX >= GC) return;
  const int nMaxCellX = std::min(GC - 1, (int)std::ceil((x - F.min_x + r) * F.inv_w));
  if (nMaxCellX < 0) return;
  const int nMinCellY = std::max(0, (int)std::floor((y - F.min_y - r) * F.inv_h));
  if (nMinCellY >= GR) return;
  const int nMaxCellY = std::min(GR - 1, (int)std::ceil((y - F.min_y + r) * F.inv_h));
  if (nMaxCellY < 0) return;
  const bool check = (minLevel > 0) || (maxLevel >= 0);
  for (int ix = nMinCellX; ix <= nMaxCellX; ix++) {
    // cells (ix, iyMin..iyMax) are contiguous in the CSR (cell = ix*48 + iy)
    const int b = F.cell_start[ix * GR + nMinCellY], e = F.cell_start[ix * GR + nMaxCellY + 1];
    for (int t = b; t < e; ++t) {
      const int idx = F.cell_items[t];
      const asd_keypoint& kp = F.kps[idx];
      if (check) {
        if (kp.octave < minLevel) continue;
        if (maxLevel >= 0 && kp.octave > maxLevel) continue;
      }
      const float dx = kp.x - x, dy = kp.y - y;
      if (std::fabs(dx) < r && std::fabs(dy) < r) pairs.push_back(make_int2(q, idx));
    }
  }
}

// cv::Mat Rcw*x + tcw (gemm 3x3*3x1 small-matrix path, A*B+C folded): left-to-right f32 dot, then + t
inline void transform(const float* T, const float* X, float* out) {
  for (int r = 0; r < 3; ++r) {
    const float t0 = T[r * 4 + 0] * X[0] + T[r * 4 + 1] * X[1] + T[r * 4 + 2] * X[2];
    out[r] = (float)((double)t0 + (double)T[r * 4 + 3]);
  }
}

int ensure_pairs(asd_ctx* ctx, size_t n) {
  if ((int)n <= ctx->pairs_cap) return ASD_OK;
  const int cap = (int)std::max<size_t>(n * 3 / 2, 1 << 16);
  if (ctx->d_pairs) (void)hipFree(ctx->d_pairs);
  if (ctx->d_pair_dist) (void)hipFree(ctx->d_pair_dist);
  if (ctx->h_pair_dist) (void)hipHostFree(ctx->h_pair_dist);
  ctx->pairs_cap = 0;
  ASD_HIP_CHECK(ctx, hipMalloc(&ctx->d_pairs, (size_t)cap * sizeof(int2)));
  ASD_HIP_CHECK(ctx, hipMalloc(&ctx->d_pair_dist, (size_t)cap * sizeof(float)));
  ASD_HIP_CHECK(ctx, hipHostMalloc(&ctx->h_pair_dist, (size_t)cap * sizeof(float)));
  ctx->pairs_cap = cap;
  return ASD_OK;
}

int ensure_qdesc(asd_ctx* ctx, size_t n) {
  if ((int)n <= ctx->qdesc_cap) return ASD_OK;
  const int cap = (int)std::max<size_t>(n * 3 / 2, 4096);
  if (ctx->d_qdesc) (void)hipFree(ctx->d_qdesc);
  ctx->qdesc_cap = 0;
  ASD_HIP_CHECK(ctx, hipMalloc(&ctx->d_qdesc, (size_t)cap * 128 * sizeof(float)));
  ctx->qdesc_cap = cap;
  return ASD_OK;
}

// distances of all pairs: query descriptors on the device (d_q), candidates = slot descriptors
int pair_distances(asd_ctx* ctx, const float* d_q, const AsdFrameSlot& F, const std::vector<int2>& pairs) {
  const int np = (int)pairs.size();
  if (np == 0) return ASD_OK;
  int rc = ensure_pairs(ctx, np);
  if (rc != ASD_OK) return rc;
  hipStream_t st = ctx->stream;
  ASD_HIP_CHECK(ctx, hipMemcpyAsync(ctx->d_pairs, pairs.data(), (size_t)np * sizeof(int2), hipMemcpyHostToDevice, st));
  ASD_HIP_CHECK(ctx, hipEventRecord(ctx->ev0, st));
  hipLaunchKernelGGL(k_pair_dist, dim3((np + 255) / 256), dim3(256), 0, st, d_q, F.d_desc, ctx->d_pairs, np, ctx->d_pair_dist);
  ASD_HIP_CHECK(ctx, hipGetLastError());
  ASD_HIP_CHECK(ctx, hipEventRecord(ctx->ev1, st));
  ASD_HIP_CHECK(ctx, hipMemcpyAsync(ctx->h_pair_dist, ctx->d_pair_dist, (size_t)np * sizeof(float), hipMemcpyDeviceToHost, st));
  ASD_HIP_CHECK(ctx, hipStreamSynchronize(st));
  ASD_HIP_CHECK(ctx, hipEventElapsedTime(&ctx->ms_match, ctx->ev0, ctx->ev1));
  return ASD_OK;
}

AsdFrameSlot* slot_of(asd_ctx* ctx, int s) {
  if (!ctx || s < 0 || s >= ASD_MAX_FRAMES) return nullptr;
  return &ctx->frames[s];
}
}  // namespace

void matcher_free(asd_ctx* ctx) {
  for (auto& f : ctx->frames)
    if (f.d_desc) { (void)hipFree(f.d_desc); f.d_desc = nullptr; }
  if (ctx->d_pairs) (void)hipFree(ctx->d_pairs);
  if (ctx->d_pair_dist) (void)hipFree(ctx->d_pair_dist);
  if (ctx->h_pair_dist) (void)hipHostFree(ctx->h_pair_dist);
  if (ctx->d_qdesc) (void)hipFree(ctx->d_qdesc);
}

extern "C" {

int asd_frame_set(asd_ctx* ctx, int32_t slot, const asd_keypoint* kps, const float* desc, int32_t n, float min_x,
                  float max_x, float min_y, float max_y) {
  AsdFrameSlot* F = slot_of(ctx, slot);
  if (!F || n < 0 || (n > 0 && !kps) || !(max_x > min_x) || !(max_y > min_y)) return ASD_ERR_INVALID;
  if (n > ctx->cfg.max_patches) { ctx->set_error("frame has %d keypoints, capacity %d", n, ctx->cfg.max_patches); return ASD_ERR_CAPACITY; }
  if (!desc && n != ctx->last_n) { ctx->set_error("desc == NULL adopts the last extract (%d keypoints), got n=%d", ctx->last_n, n); return ASD_ERR_INVALID; }
  (void)hipSetDevice(ctx->cfg.device);
  if (!F->d_desc) ASD_HIP_CHECK(ctx, hipMalloc(&F->d_desc, (size_t)ctx->cfg.max_patches * 128 * sizeof(float)));
  if (n > 0) {
    if (desc) ASD_HIP_CHECK(ctx, hipMemcpyAsync(F->d_desc, desc, (size_t)n * 128 * sizeof(float), hipMemcpyHostToDevice, ctx->stream));
    else ASD_HIP_CHECK(ctx, hipMemcpyAsync(F->d_desc, ctx->d_desc, (size_t)n * 128 * sizeof(float), hipMemcpyDeviceToDevice, ctx->stream));
  }
  F->n = n;
  F->min_x = min_x; F->max_x = max_x; F->min_y = min_y; F->max_y = max_y;
  F->inv_w = static_cast<float>(GC) / static_cast<float>(max_x - min_x);  // Frame.cc:106-107
  F->inv_h = static_cast<float>(GR) / static_cast<float>(max_y - min_y);
  F->kps.assign(kps, kps + n);
  // AssignFeaturesToGrid (Frame.cc:123-138): stable counting sort into CSR, cell = ix*48 + iy
  std::vector<int> cell(n);
  F->cell_start.assign(GC * GR + 1, 0);
  for (int i = 0; i < n; ++i) {
    const int px = (int)std::round((kps[i].x - min_x) * F->inv_w);  // PosInGrid uses round (Frame.cc:278-279)
    const int py = (int)std::round((kps[i].y - min_y) * F->inv_h);
    if (px < 0 || px >= GC || py < 0 || py >= GR) { cell[i] = -1; continue; }
    cell[i] = px * GR + py;
    ++F->cell_start[cell[i] + 1];
  }
  for (int c = 0; c < GC * GR; ++c) F->cell_start[c + 1] += F->cell_start[c];
  F->cell_items.assign(F->cell_start[GC * GR], 0);
  std::vector<int> cur(F->cell_start.begin(), F->cell_start.end() - 1);
  for (int i = 0; i < n; ++i)
    if (cell[i] >= 0) F->cell_items[cur[cell[i]]++] = i;
  ASD_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  return ASD_OK;
}

int asd_frame_features_in_area(asd_ctx* ctx, int32_t slot, float x, float y, float r, int32_t min_level,
                               int32_t max_level, int32_t capacity, int32_t* idx_out, int32_t* n_out) {
  AsdFrameSlot* F = slot_of(ctx, slot);
  if (!F || !idx_out || !n_out) return ASD_ERR_INVALID;
  std::vector<int2> pairs;
  features_in_area(*F, x, y, r, min_level, max_level, 0, pairs);
  const int n = std::min((int)pairs.size(), capacity);
  for (int i = 0; i < n; ++i) idx_out[i] = pairs[i].y;
  *n_out = n;
  return ASD_OK;
}

int asd_dist_matrix(asd_ctx* ctx, const float* a, int32_t na, const float* b, int32_t nb, float* out) {
  if (!ctx || na < 0 || nb < 0 || ((na > 0 && nb > 0) && (!a || !b || !out))) return ASD_ERR_INVALID;
  if (na == 0 || nb == 0) return ASD_OK;
  (void)hipSetDevice(ctx->cfg.device);
  float *da = nullptr, *db = nullptr, *dout = nullptr;
  ASD_HIP_CHECK(ctx, hipMalloc(&da, (size_t)na * 512));
  ASD_HIP_CHECK(ctx, hipMalloc(&db, (size_t)nb * 512));
  ASD_HIP_CHECK(ctx, hipMalloc(&dout, (size_t)na * nb * sizeof(float)));
  hipStream_t st = ctx->stream;
  ASD_HIP_CHECK(ctx, hipMemcpyAsync(da, a, (size_t)na * 512, hipMemcpyHostToDevice, st));
  ASD_HIP_CHECK(ctx, hipMemcpyAsync(db, b, (size_t)nb * 512, hipMemcpyHostToDevice, st));
  ASD_HIP_CHECK(ctx, hipEventRecord(ctx->ev0, st));
  hipLaunchKernelGGL(k_dist_matrix, dim3((nb + 255) / 256, (na + 15) / 16), dim3(256), 0, st, da, na, db, nb, dout);
  ASD_HIP_CHECK(ctx, hipGetLastError());
  ASD_HIP_CHECK(ctx, hipEventRecord(ctx->ev1, st));
  ASD_HIP_CHECK(ctx, hipMemcpyAsync(out, dout, (size_t)na * nb * sizeof(float), hipMemcpyDeviceToHost, st));
  ASD_HIP_CHECK(ctx, hipStreamSynchronize(st));
  ASD_HIP_CHECK(ctx, hipEventElapsedTime(&ctx->ms_match, ctx->ev0, ctx->ev1));
  (void)hipFree(da); (void)hipFree(db); (void)hipFree(dout);
  return ASD_OK;
}

int asd_distinctive_descriptor(asd_ctx* ctx, const float* desc, int32_t n, int32_t* best_idx) {
  if (!ctx || !desc || !best_idx || n < 1) return ASD_ERR_INVALID;
  std::vector<float> D((size_t)n * n);
  int rc = asd_dist_matrix(ctx, desc, n, desc, n, D.data());
  if (rc != ASD_OK) return rc;
  // MapPoint.cc:305-331: the reference fills the upper triangle and mirrors it; (a-b)^2 == (b-a)^2
  // term by term, so the full matrix is already symmetric bit for bit.  Diagonal = 0.
  float best_median = 100;
  int best = 0;
  std::vector<float> row(n);
  for (int i = 0; i < n; ++i) {
    std::copy(D.begin() + (size_t)i * n, D.begin() + (size_t)(i + 1) * n, row.begin());
    row[i] = 0;
    std::sort(row.begin(), row.end());
    const float median = row[(size_t)(0.5 * (n - 1))];
    if (median < best_median) { best_median = median; best = i; }
  }
  *best_idx = best;
  return ASD_OK;
}

int asd_match_project_frame(asd_ctx* ctx, int32_t slot_cur, int32_t slot_last, const uint8_t* has_mp, const float* Xw,
                            const float* mp_desc, const float* Tcw, const float* K, float th,
                            int32_t check_orientation, int32_t* match_cur, int32_t* n_matches) {
  AsdFrameSlot *C = slot_of(ctx, slot_cur), *L = slot_of(ctx, slot_last);
  if (!C || !L || !has_mp || !Xw || !mp_desc || !Tcw || !K || !match_cur || !n_matches) return ASD_ERR_INVALID;
  (void)hipSetDevice(ctx->cfg.device);
  const float fx = K[0], fy = K[1], cx = K[2], cy = K[3];
  std::vector<int2> pairs;
  std::vector<Window> win(L->n, Window{0, 0});
  pairs.reserve((size_t)L->n * 24);
  for (int i = 0; i < L->n; ++i) {
    if (!has_mp[i]) continue;
    float Xc[3];
    transform(Tcw, Xw + 3 * i, Xc);
    const float invzc = 1.0 / Xc[2];  // double division, rounded to float (:1352)
    if (invzc < 0) continue;
    const float u = fx * Xc[0] * invzc + cx;
    const float v = fy * Xc[1] * invzc + cy;
    if (u < C->min_x || u > C->max_x) continue;
    if (v < C->min_y || v > C->max_y) continue;
    const int oct = L->kps[i].octave;
    const float radius = th * ctx->scale[oct];
    win[i].begin = (int)pairs.size();
    features_in_area(*C, u, v, radius, oct - 1, oct + 1, i, pairs);
    win[i].end = (int)pairs.size();
  }
  int rc = ensure_qdesc(ctx, L->n);
  if (rc != ASD_OK) return rc;
  if (!pairs.empty()) {
    ASD_HIP_CHECK(ctx, hipMemcpyAsync(ctx->d_qdesc, mp_desc, (size_t)L->n * 512, hipMemcpyHostToDevice, ctx->stream));
    rc = pair_distances(ctx, ctx->d_qdesc, *C, pairs);
    if (rc != ASD_OK) return rc;
  }
  const float* dist = ctx->h_pair_dist;
  std::fill(match_cur, match_cur + C->n, -1);
  int nmatches = 0;
  std::vector<int> hist[HISTO];
  for (int i = 0; i < L->n; ++i) {
    float best = 100;
    int best_idx = -1;
    for (int t = win[i].begin; t < win[i].end; ++t) {
      const int j = pairs[t].y;
      if (match_cur[j] >= 0) continue;  // already holds a map point with Observations() > 0
      if (dist[t] < best) { best = dist[t]; best_idx = j; }
    }
    if (win[i].end > win[i].begin && best <= TH_HIGH) {
      match_cur[best_idx] = i;
      nmatches++;
      if (check_orientation) hist[rot_bin(L->kps[i].angle, C->kps[best_idx].angle)].push_back(best_idx);
    }
  }
  if (check_orientation) {
    int cnt[HISTO], i1, i2, i3;
    for (int b = 0; b < HISTO; ++b) cnt[b] = (int)hist[b].size();
    three_maxima(cnt, i1, i2, i3);
    for (int b = 0; b < HISTO; ++b)
      if (b != i1 && b != i2 && b != i3)
        for (int j : hist[b]) { match_cur[j] = -1; nmatches--; }
  }
  *n_matches = nmatches;
  return ASD_OK;
}

int asd_match_project_points(asd_ctx* ctx, int32_t slot_cur, int32_t n_mp, const uint8_t* in_view, const float* proj,
                             const int32_t* level, const float* view_cos, const float* desc, const uint8_t* occupied,
                             float th, float nn_ratio, int32_t* match_cur, int32_t* n_matches) {
  AsdFrameSlot* F = slot_of(ctx, slot_cur);
  if (!F || n_mp < 0 || !match_cur || !n_matches || (n_mp > 0 && (!in_view || !proj || !level || !view_cos || !desc)) ||
      (F->n > 0 && !occupied))
    return ASD_ERR_INVALID;
  (void)hipSetDevice(ctx->cfg.device);
  const bool bFactor = th != 1.0;
  std::vector<int2> pairs;
  std::vector<Window> win(n_mp, Window{0, 0});
  pairs.reserve((size_t)n_mp * 8);
  for (int m = 0; m < n_mp; ++m) {
    if (!in_view[m]) continue;
    const int lvl = level[m];
    if (lvl < 0 || lvl >= ctx->cfg.n_levels) { ctx->set_error("map point %d: level %d out of range", m, lvl); return ASD_ERR_INVALID; }
    float r = view_cos[m] > 0.998 ? 2.5f : 4.0f;  // RadiusByViewingCos (:126-132)
    if (bFactor) r *= th;
    win[m].begin = (int)pairs.size();
    features_in_area(*F, proj[2 * m], proj[2 * m + 1], r * ctx->scale[lvl], lvl - 1, lvl, m, pairs);
    win[m].end = (int)pairs.size();
  }
  int rc = ensure_qdesc(ctx, n_mp);
  if (rc != ASD_OK) return rc;
  if (!pairs.empty()) {
    ASD_HIP_CHECK(ctx, hipMemcpyAsync(ctx->d_qdesc, desc, (size_t)n_mp * 512, hipMemcpyHostToDevice, ctx->stream));
    rc = pair_distances(ctx, ctx->d_qdesc, *F, pairs);
    if (rc != ASD_OK) return rc;
  }
  const float* dist = ctx->h_pair_dist;
  std::fill(match_cur, match_cur + F->n, -1);
  int nmatches = 0;
  for (int m = 0; m < n_mp; ++m) {
    if (win[m].end == win[m].begin) continue;
    float best = 256, best2 = 256;
    int best_lvl = -1, best_lvl2 = -1, best_idx = -1;
    for (int t = win[m].begin; t < win[m].end; ++t) {
      const int j = pairs[t].y;
      if (occupied[j] || match_cur[j] >= 0) continue;
      const float d = dist[t];
      if (d < best) {
        best2 = best; best = d;
        best_lvl2 = best_lvl; best_lvl = F->kps[j].octave;
        best_idx = j;
      } else if (d < best2) {
        best_lvl2 = F->kps[j].octave;
        best2 = d;
      }
    }
    if (best <= TH_HIGH) {
      if (best_lvl == best_lvl2 && best > nn_ratio * best2) continue;
      match_cur[best_idx] = m;
      nmatches += 2;  // the reference increments twice per match (ORBmatcher.cc:116-117)
    }
  }
  *n_matches = nmatches;
  return ASD_OK;
}

int asd_frustum(asd_ctx* ctx, int32_t slot_cur, int32_t n, const float* Xw, const float* normal, const float* min_dist,
                const float* max_dist, const float* Tcw, const float* K, float cos_limit, uint8_t* in_view,
                float* proj, int32_t* level, float* view_cos) {
  AsdFrameSlot* F = slot_of(ctx, slot_cur);
  if (!F || n < 0 || !Tcw || !K || (n > 0 && (!Xw || !normal || !min_dist || !max_dist || !in_view || !proj || !level || !view_cos)))
    return ASD_ERR_INVALID;
  const float fx = K[0], fy = K[1], cx = K[2], cy = K[3];
  const float log_scale = std::log(ctx->cfg.scale_factor);  // Frame.cc:74 (float log)
  float Ow[3];  // mOw = -mRcw.t()*mtcw (Frame.cc:157): transposed gemm accumulates in double
  for (int i = 0; i < 3; ++i) {
    double s = 0;
    for (int k = 0; k < 3; ++k) s += (double)Tcw[k * 4 + i] * (double)Tcw[k * 4 + 3];
    Ow[i] = (float)(-1.0 * s);
  }
  for (int m = 0; m < n; ++m) {
    in_view[m] = 0; proj[2 * m] = proj[2 * m + 1] = 0.f; level[m] = 0; view_cos[m] = 0.f;
    const float* P = Xw + 3 * m;
    float Pc[3];
    transform(Tcw, P, Pc);
    if (Pc[2] < 0.0f) continue;
    const float invz = 1.0f / Pc[2];
    const float u = fx * Pc[0] * invz + cx, v = fy * Pc[1] * invz + cy;
    if (u < F->min_x || u > F->max_x || v < F->min_y || v > F->max_y) continue;
    const float maxD = 1.2f * max_dist[m], minD = 0.8f * min_dist[m];  // MapPoint.cc:409-419
    const float PO[3] = {P[0] - Ow[0], P[1] - Ow[1], P[2] - Ow[2]};
    const double nn = (double)PO[0] * PO[0] + (double)PO[1] * PO[1] + (double)PO[2] * PO[2];
    const float dist = (float)std::sqrt(nn);
    if (dist < minD || dist > maxD) continue;
    const float* Pn = normal + 3 * m;
    const double dot = (double)PO[0] * Pn[0] + (double)PO[1] * Pn[1] + (double)PO[2] * Pn[2];
    const float vc = (float)(dot / dist);
    if (vc < cos_limit) continue;
    const float ratio = max_dist[m] / dist;  // MapPoint::PredictScale (MapPoint.cc:438-453)
    int s = (int)std::ceil(std::log(ratio) / log_scale);
    if (s < 0) s = 0;
    else if (s >= ctx->cfg.n_levels) s = ctx->cfg.n_levels - 1;
    in_view[m] = 1;
    proj[2 * m] = u; proj[2 * m + 1] = v;
    level[m] = s;
    view_cos[m] = vc;
  }
  return ASD_OK;
}

int asd_match_init(asd_ctx* ctx, int32_t slot1, int32_t slot2, float* prev_matched, int32_t window, float nn_ratio,
                   int32_t check_orientation, int32_t* matches12, int32_t* n_matches) {
  AsdFrameSlot *F1 = slot_of(ctx, slot1), *F2 = slot_of(ctx, slot2);
  if (!F1 || !F2 || !prev_matched || !matches12 || !n_matches) return ASD_ERR_INVALID;
  (void)hipSetDevice(ctx->cfg.device);
  std::vector<int2> pairs;
  std::vector<Window> win(F1->n, Window{0, 0});
  for (int i1 = 0; i1 < F1->n; ++i1) {
    if (F1->kps[i1].octave > 0) continue;  // :434-436
    win[i1].begin = (int)pairs.size();
    features_in_area(*F2, prev_matched[2 * i1], prev_matched[2 * i1 + 1], (float)window, 0, 0, i1, pairs);
    win[i1].end = (int)pairs.size();
  }
  int rc = pair_distances(ctx, F1->d_desc, *F2, pairs);  // query descriptors are frame 1's, resident
  if (rc != ASD_OK) return rc;
  const float* dist = ctx->h_pair_dist;
  std::fill(matches12, matches12 + F1->n, -1);
  std::vector<float> matched_dist(F2->n, 100.f);
  std::vector<int> matches21(F2->n, -1);
  std::vector<int> hist[HISTO];
  int nmatches = 0;
  for (int i1 = 0; i1 < F1->n; ++i1) {
    if (win[i1].end == win[i1].begin) continue;
    float best = 100.f, best2 = 100.f;
    int best_idx = -1;
    for (int t = win[i1].begin; t < win[i1].end; ++t) {
      const int i2 = pairs[t].y;
      const float d = dist[t];
      if (matched_dist[i2] <= d) continue;
      if (d < best) { best2 = best; best = d; best_idx = i2; }
      else if (d < best2) best2 = d;
    }
    if (best <= TH_LOW && best < best2 * nn_ratio) {
      if (matches21[best_idx] >= 0) { matches12[matches21[best_idx]] = -1; nmatches--; }
      matches12[i1] = best_idx;
      matches21[best_idx] = i1;
      matched_dist[best_idx] = best;
      nmatches++;
      if (check_orientation) hist[rot_bin(F1->kps[i1].angle, F2->kps[best_idx].angle)].push_back(i1);
    }
  }
  if (check_orientation) {
    int cnt[HISTO], a, b, c;
    for (int k = 0; k < HISTO; ++k) cnt[k] = (int)hist[k].size();
    three_maxima(cnt, a, b, c);
    for (int k = 0; k < HISTO; ++k) {
      if (k == a || k == b || k == c) continue;
      for (int idx1 : hist[k])
        if (matches12[idx1] >= 0) { matches12[idx1] = -1; nmatches--; }
    }
  }
  for (int i1 = 0; i1 < F1->n; ++i1)
    if (matches12[i1] >= 0) {
      prev_matched[2 * i1] = F2->kps[matches12[i1]].x;
      prev_matched[2 * i1 + 1] = F2->kps[matches12[i1]].y;
    }
  *n_matches = nmatches;
  return ASD_OK;
}

}  // extern "C"
