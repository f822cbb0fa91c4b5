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
// call.  So: the host only projects the queries (u, v, radius, level range); ONE kernel
// (k_window_search, a wave per query) walks the frame's 64x48 CSR grid in the reference's visiting
// order (Frame::GetFeaturesInArea: ix outer, iy inner, insertion order inside a cell), compacts the
// accepted candidates by ballot/prefix in that order and evaluates each candidate's squared L2 in
// the reference's exact summation order (sequential f32, no FMA: -ffp-contract=off) from
// descriptors resident in HBM; the host then replays the claim / ratio / rotation-histogram logic
// over the per-query (index, distance) lists.
// The all-pairs matrix (asd_dist_matrix) uses the same exact summation, tiled through LDS.
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstring>
#include <array>
#include <functional>
#include <memory>

#include "ctx.h"
#include "resolve2.h"

namespace {

// (TH_HIGH, TH_LOW, HISTO, kTop: resolve2.h)
constexpr int GC = ASD_GRID_COLS, GR = ASD_GRID_ROWS;

// A window query = one GetFeaturesInArea call + the descriptor it is matched against.
using WinQuery = AsdWinQuery;   // (ctx.h: the solver's tail makes queries too)
struct GridDev {
  const float4* kp;        // (x, y, octave as int bits, angle) per keypoint
  const int* cell_start;   // [64*48+1], cell = ix*48 + iy
  const int* cell_items;
  float min_x, min_y, inv_w, inv_h;
};

// one wave per query: pass 1 counts the candidates, one atomicAdd reserves the query's segment
// (segments may land in any order, each query's own list is in reference order), pass 2 writes
// (candidate index, distance).  Arithmetic of the cell range / distance tests is the float
// arithmetic of Frame.cc:226-265 verbatim.
//
// SORT = true (the device claim replay, k_resolve; round 4): the query's list leaves the kernel in PREFERENCE order instead -- ascending
// (distance bits, position in Frame::GetFeaturesInArea order), which is the order the reference's strict `<` walks settle ties in
// (ORBmatcher.cc:86-104, :1392-1404: the first candidate in visiting order wins among equal distances) -- with its four best entries
// also in a compact per-query table.  The replay then needs the head of each list only: "best candidate no earlier map point holds"
// is the first entry of the sorted list without such a claim, the second best the next one.  A wave ranks its list by counting
// (rank of p = number of entries with a smaller key; keys are distinct because positions are): O(n^2 / 64) LDS broadcast reads for
// lists of 6 entries on average.  Entries the replay can never pick are left out here: keypoints that hold a map point on entry
// (`occ`, ORBmatcher.cc:84-88) and, with th_cut = TH_HIGH, candidates beyond it (frame-to-frame search: best-only, :1406).
#ifndef ASD_SEARCH_WAVES
#define ASD_SEARCH_WAVES 8
#endif
constexpr int kSearchWaves = ASD_SEARCH_WAVES;   // queries (waves) per workgroup
constexpr int kSearchCols = 16;                  // fast path: windows of up to this many grid columns ...
constexpr int kSearchList = 128;                 // ... and up to this many candidates per query
struct SortArgs {
  const uint8_t* occ;      // [n_cur] or null: keypoints that cannot be matched (occupied on entry)
  float th_cut;            // q_cnt counts the entries with distance <= th_cut (they are the list's head); +inf = all
  uint16_t* top_idx;       // [nq][kTop] keypoint (0xffff = no entry)
};
template <bool SORT>
__global__ __launch_bounds__(64 * kSearchWaves) void k_window_search(GridDev G, const WinQuery* __restrict__ queries, int nq,
                                                       const float* __restrict__ qdesc, const float* __restrict__ cdesc,
                                                       int* __restrict__ q_off, int* __restrict__ q_cnt,
                                                       int* __restrict__ total, int cap, int* __restrict__ out_idx,
                                                       float* __restrict__ out_dist, unsigned* __restrict__ out_meta, SortArgs S) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int q = blockIdx.x * kSearchWaves + wave;
  // the waves of a workgroup reserve their segments with ONE atomicAdd (2000 same-address atomics, one per query, were a
  // serial chain through one L2 channel): every wave stays alive up to the barriers below, a query beyond nq counts as empty
  __shared__ int wg_cnt[kSearchWaves], wg_base;
  __shared__ int cand_l[kSearchWaves][kSearchList];
  __shared__ unsigned dist_l[SORT ? kSearchWaves : 1][kSearchList];
  const bool live = q < nq;
  const WinQuery Q = live ? queries[q] : WinQuery{0.f, 0.f, 0.f, 0, 0, -1};
  int cnt = 0, off = 0;
  bool empty = Q.qrow < 0;
  const int nMinCellX = max(0, (int)floorf((Q.x - G.min_x - Q.r) * G.inv_w));
  const int nMaxCellX = min(ASD_GRID_COLS - 1, (int)ceilf((Q.x - G.min_x + Q.r) * G.inv_w));
  const int nMinCellY = max(0, (int)floorf((Q.y - G.min_y - Q.r) * G.inv_h));
  const int nMaxCellY = min(ASD_GRID_ROWS - 1, (int)ceilf((Q.y - G.min_y + Q.r) * G.inv_h));
  if (nMinCellX >= ASD_GRID_COLS || nMaxCellX < 0 || nMinCellY >= ASD_GRID_ROWS || nMaxCellY < 0) empty = true;
  const bool check = (Q.min_level > 0) || (Q.max_level >= 0);
  auto accept = [&](const float4& kp) {
    const int oct = __float_as_int(kp.z);
    bool ok = true;
    if (check) {
      if (oct < Q.min_level) ok = false;
      if (Q.max_level >= 0 && oct > Q.max_level) ok = false;
    }
    const float dx = kp.x - Q.x, dy = kp.y - Q.y;
    if (!(fabsf(dx) < Q.r && fabsf(dy) < Q.r)) ok = false;
    return ok;
  };
  // the query's row is the same for the whole wave: read through the scalar cache (no vector registers, no address per lane), which leaves
  // room for sixteen candidate quads in flight per lane -- two dependent round trips per distance instead of four (beside the ASDNet
  // grids a dependent read takes 2-3 us, and this kernel's time is its waves' chains of them)
  const int qrow_u = __builtin_amdgcn_readfirstlane(Q.qrow < 0 ? 0 : Q.qrow);
  auto distance = [&](int idx) {   // exact summation order of DescriptorDistance (sequential f32)
    const float4* a = reinterpret_cast<const float4*>(qdesc + (size_t)qrow_u * 128);
    const float4* bb = reinterpret_cast<const float4*>(cdesc + (size_t)idx * 128);
    float sqd = 0.f;
#pragma unroll 16
    for (int k = 0; k < 32; ++k) {
      const float4 x = a[k], y = bb[k];
      float d;
      d = x.x - y.x; sqd = sqd + d * d;
      d = x.y - y.y; sqd = sqd + d * d;
      d = x.z - y.z; sqd = sqd + d * d;
      d = x.w - y.w; sqd = sqd + d * d;
    }
    return sqd;
  };
  auto score = [&](int idx, int p) {   // candidate p of the list, in list order
    const float sqd = distance(idx);
    out_idx[off + p] = idx;
    out_dist[off + p] = sqd;
    if (out_meta) out_meta[off + p] = ((unsigned)p << 16) | (unsigned)q;   // position in the list | query
  };
  // the general walk (any window, any list length): one dependent chain cell range -> items -> keypoints per column and pass
  auto walk = [&](int pass) {
    int pos = 0;
    for (int ix = nMinCellX; ix <= nMaxCellX; ++ix) {
      const int b = G.cell_start[ix * ASD_GRID_ROWS + nMinCellY], e = G.cell_start[ix * ASD_GRID_ROWS + nMaxCellY + 1];
      for (int base = b; base < e; base += 64) {
        const int it = base + lane;
        bool ok = false;
        int idx = 0;
        if (it < e) {
          idx = G.cell_items[it];
          ok = accept(G.kp[idx]);
          if (SORT && ok && S.occ && S.occ[idx]) ok = false;
        }
        const unsigned long long m = __ballot(ok);
        if (pass == 1 && ok) {
          const int p = pos + __popcll(m & ((1ull << lane) - 1));
          if (off + p < cap) score(idx, p);
        }
        pos += __popcll(m);
      }
    }
    return pos;
  };
  // Fast path (wave-uniform choice): the window's columns fit kSearchCols register slots and no column holds more than 64 items.
  // All column ranges are fetched together, then all item lists, then all keypoints -- three round trips instead of three per
  // column -- and the accepted keypoints are parked in LDS in list order, so that the second pass is one distance per lane
  // instead of a second walk.  Same visiting order (columns ascending, items in cell order), same arithmetic.
  const int ncol = nMaxCellX - nMinCellX + 1;
  bool fast = !empty && ncol <= kSearchCols;
  int cb[kSearchCols], ce[kSearchCols];
  if (fast) {
#pragma unroll
    for (int c = 0; c < kSearchCols; ++c) {
      const int ix = min(nMinCellX + c, nMaxCellX);
      cb[c] = G.cell_start[ix * ASD_GRID_ROWS + nMinCellY];
      ce[c] = G.cell_start[ix * ASD_GRID_ROWS + nMaxCellY + 1];
    }
#pragma unroll
    for (int c = 0; c < kSearchCols; ++c) {
      if (c >= ncol) ce[c] = cb[c];                 // unused slot: empty range
      if (ce[c] - cb[c] > 64) fast = false;
    }
  }
  if (fast) {
    int idxc[kSearchCols];
#pragma unroll
    for (int c = 0; c < kSearchCols; ++c) idxc[c] = cb[c] + lane < ce[c] ? G.cell_items[cb[c] + lane] : -1;
    float4 kpc[kSearchCols];
#pragma unroll
    for (int c = 0; c < kSearchCols; ++c) kpc[c] = idxc[c] >= 0 ? G.kp[idxc[c]] : make_float4(0.f, 0.f, 0.f, 0.f);
    if (SORT && S.occ) {   // keypoints occupied on entry never enter a list (one more batch of loads, in flight with the keypoints)
      uint8_t oc[kSearchCols];
#pragma unroll
      for (int c = 0; c < kSearchCols; ++c) oc[c] = idxc[c] >= 0 ? S.occ[idxc[c]] : (uint8_t)0;
#pragma unroll
      for (int c = 0; c < kSearchCols; ++c) if (oc[c]) idxc[c] = -1;
    }
    int pos = 0;
#pragma unroll
    for (int c = 0; c < kSearchCols; ++c) {
      const bool ok = idxc[c] >= 0 && accept(kpc[c]);
      const unsigned long long m = __ballot(ok);
      if (ok) {
        const int p = pos + __popcll(m & ((1ull << lane) - 1));
        if (p < kSearchList) cand_l[wave][p] = idxc[c];
      }
      pos += __popcll(m);
    }
    cnt = pos;
    if (cnt > kSearchList) fast = false;            // (the list did not fit: the second pass walks again)
  } else if (!empty) {
    cnt = walk(0);
  }
  if (lane == 0) wg_cnt[wave] = cnt;
  asd_syncthreads();
  if (threadIdx.x == 0) {
    int sum = 0;
    for (int w = 0; w < kSearchWaves; ++w) sum += wg_cnt[w];
    wg_base = sum ? atomicAdd(total, sum) : 0;
  }
  asd_syncthreads();
  off = wg_base;
  for (int w = 0; w < wave; ++w) off += wg_cnt[w];
  if (!SORT) {
    if (cnt > 0) {
      if (fast) {
        for (int p = lane; p < cnt; p += 64)
          if (off + p < cap) score(cand_l[wave][p], p);
      } else {
        (void)walk(1);
      }
    }
    if (lane == 0 && live) { q_cnt[q] = cnt; q_off[q] = cnt ? off : 0; }
    return;
  }
  // ---- SORT: the list in preference order + the compact head table
  if (!live) return;
  const bool fits = off + cnt <= cap;   // (an overflowing search is run again by the host: nothing of it is read)
  int n_keep = 0, s_off = off;
  auto put = [&](int base, int rank, int idx, float d) {
    out_idx[base + rank] = idx;
    out_dist[base + rank] = d;
    if (rank < kTop) S.top_idx[(size_t)q * kTop + rank] = (uint16_t)idx;
  };
  if (cnt > 0 && fits) {
    if (fast) {   // at most two candidates per lane, every key of the list in LDS
      const bool v0 = lane < cnt, v1 = lane + 64 < cnt;
      const int i0 = v0 ? cand_l[wave][lane] : 0, i1 = v1 ? cand_l[wave][lane + 64] : 0;
      const float d0 = v0 ? distance(i0) : 0.f, d1 = v1 ? distance(i1) : 0.f;
      const unsigned b0 = __float_as_uint(d0), b1 = __float_as_uint(d1);
      if (v0) dist_l[wave][lane] = b0;
      if (v1) dist_l[wave][lane + 64] = b1;
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_wave_barrier();
      int r0 = 0, r1 = 0;
      for (int i = 0; i < cnt; ++i) {
        const unsigned di = dist_l[wave][i];
        r0 += (di < b0) || (di == b0 && i < lane);
        r1 += (di < b1) || (di == b1 && i < lane + 64);
      }
      if (v0) put(off, r0, i0, d0);
      if (v1) put(off, r1, i1, d1);
      n_keep = __popcll(__ballot(v0 && d0 <= S.th_cut)) + __popcll(__ballot(v1 && d1 <= S.th_cut));
    } else {      // any length: the list is written in visiting order first, then ranked from there in chunks of kSearchList keys
      (void)walk(1);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
      s_off = cap + off;   // the sorted copy lives in the buffers' second half
      constexpr int R = 4;
      for (int pb = 0; pb < cnt; pb += 64 * R) {
        int pi[R], ii[R], rk[R]; unsigned bi[R];
#pragma unroll
        for (int r = 0; r < R; ++r) {
          pi[r] = pb + 64 * r + lane;
          const int pc = min(pi[r], cnt - 1);
          ii[r] = __hip_atomic_load(out_idx + off + pc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          bi[r] = __float_as_uint(__hip_atomic_load(out_dist + off + pc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
          rk[r] = 0;
        }
        for (int cbase = 0; cbase < cnt; cbase += kSearchList) {
          __builtin_amdgcn_wave_barrier();
          for (int i = lane; i < kSearchList; i += 64)
            dist_l[wave][i] = cbase + i < cnt ? __float_as_uint(__hip_atomic_load(out_dist + off + cbase + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) : 0xffffffffu;
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          __builtin_amdgcn_wave_barrier();
          const int m = min(kSearchList, cnt - cbase);
          for (int i = 0; i < m; ++i) {
            const unsigned di = dist_l[wave][i];
#pragma unroll
            for (int r = 0; r < R; ++r) rk[r] += (di < bi[r]) || (di == bi[r] && cbase + i < pi[r]);
          }
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
#pragma unroll
        for (int r = 0; r < R; ++r) {
          const bool v = pi[r] < cnt;
          if (v) put(s_off, rk[r], ii[r], __uint_as_float(bi[r]));
          n_keep += __popcll(__ballot(v && __uint_as_float(bi[r]) <= S.th_cut));
        }
      }
    }
  }
  if (lane < kTop && lane >= (fits ? cnt : 0)) S.top_idx[(size_t)q * kTop + lane] = 0xffffu;
  if (lane == 0) { q_cnt[q] = n_keep; q_off[q] = cnt ? s_off : 0; }
}

// Node-restricted search (BoW-guided matchers): query q is matched against the explicit candidate list
// cand[cbeg[q] .. cend[q]) (the frame-2 members of the query's vocabulary node); distances land at
// out[ooff[q] + t] in list order.  One wave per query, lanes over candidates.
struct ListQuery { int qrow, cbeg, cend, ooff; };
__global__ __launch_bounds__(256) void k_list_dist(const ListQuery* __restrict__ queries, int nq,
                                                   const int* __restrict__ cand, const float* __restrict__ qdesc,
                                                   const float* __restrict__ cdesc, float* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  const int q = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (q >= nq) return;
  const ListQuery Q = queries[q];
  const float4* a = reinterpret_cast<const float4*>(qdesc + (size_t)Q.qrow * 128);
  for (int t = Q.cbeg + lane; t < Q.cend; t += 64) {
    const float4* b = reinterpret_cast<const float4*>(cdesc + (size_t)cand[t] * 128);
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
    out[Q.ooff + (t - Q.cbeg)] = sqd;
  }
}

// MapPoint::ComputeDistinctiveDescriptors for many map points at once: one workgroup per map point (<= 64
// observations), descriptors staged in LDS, all-pairs distances in the exact summation order, then the row medians by
// rank counting (the k-th order statistic is the element that k others precede; ties ordered by index, which does not
// change the VALUE found) and the first row with the smallest median (strict `<`, MapPoint.cc:318-331).
constexpr int kDistinctMax = 64;
__global__ __launch_bounds__(256) void k_distinctive(const float* __restrict__ desc, const int* __restrict__ set_start, int* __restrict__ best_out) {
  extern __shared__ __attribute__((aligned(16))) float smem_d[];
  const int s0 = set_start[blockIdx.x], n = set_start[blockIdx.x + 1] - s0;
  float* sd = smem_d;                       // [n][132] (padded rows)
  float* D = smem_d + kDistinctMax * 132;   // [n][n + 1]
  float* med = D + kDistinctMax * (kDistinctMax + 1);
  const int t = threadIdx.x;
  for (int idx = t; idx < n * 32; idx += 256) {
    const int r = idx >> 5, c = idx & 31;
    *reinterpret_cast<float4*>(sd + r * 132 + c * 4) = reinterpret_cast<const float4*>(desc + (size_t)(s0 + r) * 128)[c];
  }
  asd_syncthreads();
  for (int p = t; p < n * n; p += 256) {
    const int i = p / n, j = p % n;
    float acc = 0.f;
    if (i != j) {
      const float4* x = reinterpret_cast<const float4*>(sd + i * 132);
      const float4* y = reinterpret_cast<const float4*>(sd + j * 132);
      for (int k = 0; k < 32; ++k) {
        const float4 a = x[k], b = y[k];
        float d;
        d = a.x - b.x; acc = acc + d * d;
        d = a.y - b.y; acc = acc + d * d;
        d = a.z - b.z; acc = acc + d * d;
        d = a.w - b.w; acc = acc + d * d;
      }
    }
    D[i * (n + 1) + j] = acc;
  }
  asd_syncthreads();
  const int kth = (int)(0.5 * (n - 1));
  for (int p = t; p < n * n; p += 256) {
    const int i = p / n, j = p % n;
    const float v = D[i * (n + 1) + j];
    int rank = 0;
    for (int l = 0; l < n; ++l) {
      const float u = D[i * (n + 1) + l];
      rank += (u < v || (u == v && l < j)) ? 1 : 0;
    }
    if (rank == kth) med[i] = v;  // exactly one j per row has this rank
  }
  asd_syncthreads();
  if (t == 0) {
    float best_median = 100;
    int best = 0;
    for (int i = 0; i < n; ++i)
      if (med[i] < best_median) { best_median = med[i]; best = i; }
    best_out[blockIdx.x] = best;
  }
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
  asd_syncthreads();
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


// device-to-device copy of descriptor rows as a plain kernel: a hipMemcpyAsync costs the host 15-25 us per call on this
// stream, a launch ~5 (frame_set's adoption of the extractor's descriptors, asd_bank_put_from_frame)
__global__ __launch_bounds__(256) void k_copy16(const uint4* __restrict__ src, uint4* __restrict__ dst, size_t n16) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) dst[i] = src[i];
}
hipError_t copy_rows(hipStream_t st, void* dst, const void* src, size_t bytes) {   // bytes % 16 == 0, 16-B aligned
  const size_t n16 = bytes / 16;
  if (!n16) return hipSuccess;
  hipLaunchKernelGGL(k_copy16, dim3((unsigned)std::min<size_t>((n16 + 255) / 256, 1024)), dim3(256), 0, st, static_cast<const uint4*>(src),
                     static_cast<uint4*>(dst), n16);
  return hipGetLastError();
}

// ---- claim / ratio / rotation-histogram replay on the device ------------------------------------------------------
// The searches are order dependent: map point q (in input order) takes the best candidate that no EARLIER map point with
// Observations() > 0 holds (ORBmatcher.cc:86-88, :1392-1395).  Written as a recurrence, pick(q) = g(picks of all q' < q):
// a triangular system with exactly one solution, the serial walk's.  A serial walk is ~100 dependent LDS / L2 round trips
// per map point, so the system is solved by fixed-point iteration instead: every map point recomputes its pick in parallel
// from the claims of the previous iteration ("keypoint j is taken for q iff the smallest claimant of j is < q") until an
// iteration changes nothing.  Correctness propagates upward from q = 0 (after k iterations the first k map points are
// final), and a state that one more iteration leaves unchanged IS the solution of the recurrence; real frames settle in
// 4-13 iterations of one barrier each (13: 2000 frame-to-frame queries with every candidate eligible and random distances).
// Claims live in LDS as two tables used alternately; an entry is (iterations-left << 16 | q) so that atomicMin lets the
// current iteration's posts overrule stale ones and nothing has to be cleared.  A keypoint held by a map point without
// observations stays available (mp_obs_positive): such map points post no claim, and the keypoint's final holder is the
// LAST writer = the largest q picking it.
// KIND 0 = SearchByProjection(frame, frame) (:1318-1452): best only, TH_HIGH, rotation histogram over every write.
// KIND 1 = SearchByProjection(frame, points) (:44-122): best / second best with levels and the ratio test; counts twice.
// Round 2-3 solved the recurrence with every candidate bidding for its query in every iteration (k_resolve: 86-91 us for the frame-to-frame
// search); since round 4 the search hands its lists over in preference order and the replay walks their heads (resolve2.h: 33-40 us) --
// the bidding kernel was removed in round 5 (DESIGN.md section 4.0c keeps its numbers).
// (the claim replay over sorted lists -- Resolve2Args, resolve2_body -- lives in resolve2.h: ba.hip runs it inside k_resolve_pose)
template <int KIND, int QPT>
__global__ __launch_bounds__(kResolve2Threads) void k_resolve2(Resolve2Args a) {
  resolve2_body<KIND, QPT, kResolve2Threads>(a);
}

// Frame::isInFrustum (Frame.cc:160-217) + MapPoint::PredictScale (MapPoint.cc:438-453) + the search window of
// ORBmatcher::SearchByProjection(F, vpMapPoints, th) (:60-70), one thread per map point, straight into the query table of
// k_window_search.  The arithmetic is asd_frustum's, operation for operation (f32 with the two double accumulations of the
// reference; -ffp-contract=off; IEEE division and square root), and the level comes from comparisons with thresholds that the
// host derived from its own logf (ctx.h, level_thr), so the queries are the ones the host would have written.
// The upload of a fused chain rides in its first kernel: the blocks behind the query blocks copy the chain's pinned upload block to its
// device twin (everything but the query table at its head, which the query blocks write), the query blocks read their own inputs from
// the PINNED block.  One launch and ~10 us of dependent copy kernel less per stage; the kernels behind this one read the device twin.
struct UploadTail {
  const uint4* src; uint4* dst;   // pinned block, device twin
  size_t first16, n16;            // 16-B words [first16, n16) are copied
  int q_blocks;                   // blocks [0, q_blocks) make queries, the rest copy
};
__device__ inline bool upload_tail_block(const UploadTail& u) {
  if ((int)blockIdx.x < u.q_blocks) return false;
  const size_t nb = gridDim.x - u.q_blocks;
  for (size_t i = u.first16 + (size_t)(blockIdx.x - u.q_blocks) * 256 + threadIdx.x; i < u.n16; i += nb * 256) u.dst[i] = u.src[i];
  return true;
}
constexpr int kUploadTailBlocks = 64;

struct FrustumArgs {
  int n, n_levels, bfactor;
  const float* Xw; const float* normal; const float* min_dist; const float* max_dist;
  const int* rows;          // bank row per map point, or null: descriptor row = map point index
  float T[16], Ow[3];
  float fx, fy, cx, cy, min_x, max_x, min_y, max_y, cos_limit, th;
  float level_thr[ASD_MAX_LEVELS], scale[ASD_MAX_LEVELS];
  WinQuery* queries;
  UploadTail up;
  // asd_track_frame (the stage behind another one, nothing from the host in between): the frame pose comes from device memory
  // (T_dev[16] + Ow[3], written by k_between), the map points are rows of the attribute bank (attr[row][8] = position, normal,
  // min / max distance), candidates flagged in `skip` are in the frame already (Tracking.cc:811-823) and make no query, and every
  // candidate's position is also written to xw_out[q] for the edges of the pose solver behind the search
  const float* T_dev; const float* attr; const uint8_t* skip; float* xw_out;
};
__global__ __launch_bounds__(256) void k_frustum_queries(FrustumArgs a) {
  if (upload_tail_block(a.up)) return;
  const int q = blockIdx.x * 256 + threadIdx.x;
  if (q >= a.n) return;
  WinQuery Q{0.f, 0.f, 0.f, 0, 0, -1};
  if (a.attr) {   // bank form: the map points are rows of the attribute bank (the per-candidate arithmetic lives in ctx.h)
    if (a.T_dev) asd_frustum_bank_point(a, a.T_dev, a.skip, q);   // asd_track_frame: pose and skip flags from the stage in front
    else {                                                        // asd_track_local_points_rows: pose from the host, by value
      float Tl[19];
      for (int i = 0; i < 16; ++i) Tl[i] = a.T[i];
      for (int i = 0; i < 3; ++i) Tl[16 + i] = a.Ow[i];
      asd_frustum_bank_point(a, Tl, a.skip, q);
    }
    return;
  }
  const float* P = a.Xw + 3 * (size_t)q;
  float Pc[3];
  for (int r = 0; r < 3; ++r) {
    const float t0 = a.T[r * 4 + 0] * P[0] + a.T[r * 4 + 1] * P[1] + a.T[r * 4 + 2] * P[2];
    Pc[r] = (float)((double)t0 + (double)a.T[r * 4 + 3]);
  }
  bool ok = !(Pc[2] < 0.0f);
  const float invz = 1.0f / Pc[2];
  const float u = a.fx * Pc[0] * invz + a.cx, v = a.fy * Pc[1] * invz + a.cy;
  if (u < a.min_x || u > a.max_x || v < a.min_y || v > a.max_y) ok = false;
  const float maxD = 1.2f * a.max_dist[q], minD = 0.8f * a.min_dist[q];
  const float PO[3] = {P[0] - a.Ow[0], P[1] - a.Ow[1], P[2] - a.Ow[2]};
  const double nn = (double)PO[0] * PO[0] + (double)PO[1] * PO[1] + (double)PO[2] * PO[2];
  const float dist = (float)sqrt(nn);
  if (dist < minD || dist > maxD) ok = false;
  const float* Pn = a.normal + 3 * (size_t)q;
  const double dot = (double)PO[0] * Pn[0] + (double)PO[1] * Pn[1] + (double)PO[2] * Pn[2];
  const float vc = (float)(dot / dist);
  if (vc < a.cos_limit) ok = false;
  if (ok) {
    const float ratio = a.max_dist[q] / dist;
    int lvl = 0;
    for (int k = 1; k < a.n_levels; ++k) lvl += ratio >= a.level_thr[k];
    float r = vc > 0.998 ? 2.5f : 4.0f;  // RadiusByViewingCos (:126-132)
    if (a.bfactor) r *= a.th;
    Q = WinQuery{u, v, r * a.scale[lvl], lvl - 1, lvl, a.rows ? a.rows[q] : q};
  }
  a.queries[q] = Q;
}

// The projection loop of ORBmatcher::SearchByProjection(CurrentFrame, LastFrame, th) (ORBmatcher.cc:1343-1368), one thread per
// map point of the last frame, straight into the query table of k_window_search: the same f32 operations in the same order as
// the host loop of match_project_frame_impl (-ffp-contract=off, the reciprocal as a double division rounded to float), the
// octave from the last frame's device keypoints.
struct ProjectArgs {
  int n;
  const uint8_t* has_mp; const float* Xw; const int* rows; const float4* kp_last;
  float T[16];
  float fx, fy, cx, cy, min_x, max_x, min_y, max_y, th;
  float scale[ASD_MAX_LEVELS];
  WinQuery* queries;
  UploadTail up;
};
__global__ __launch_bounds__(256) void k_project_queries(ProjectArgs a) {
  if (upload_tail_block(a.up)) return;
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= a.n) return;
  WinQuery Q{0.f, 0.f, 0.f, 0, 0, -1};
  if (a.has_mp[i]) {
    const float* X = a.Xw + 3 * (size_t)i;
    float Xc[3];
    for (int r = 0; r < 3; ++r) {
      const float t0 = a.T[r * 4 + 0] * X[0] + a.T[r * 4 + 1] * X[1] + a.T[r * 4 + 2] * X[2];
      Xc[r] = (float)((double)t0 + (double)a.T[r * 4 + 3]);
    }
    const float invzc = (float)(1.0 / (double)Xc[2]);
    if (!(invzc < 0)) {
      const float u = a.fx * Xc[0] * invzc + a.cx;
      const float v = a.fy * Xc[1] * invzc + a.cy;
      if (!(u < a.min_x || u > a.max_x) && !(v < a.min_y || v > a.max_y)) {
        const int oct = __float_as_int(a.kp_last[i].z);
        Q = WinQuery{u, v, a.th * a.scale[oct], oct - 1, oct + 1, a.rows ? a.rows[i] : i};
      }
    }
  }
  a.queries[i] = Q;
}

// (k_between's body lives in ctx.h: asd_between_body -- it runs as the tail of the motion-model stage's k_pose_opt)
// ---- host helpers -------------------------------------------------------------------------
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

// Frame::GetFeaturesInArea (Frame.cc:219-274) on the host CSR mirror (asd_frame_features_in_area only)
void features_in_area(const AsdFrameSlot& F, float x, float y, float r, int minLevel, int maxLevel, std::vector<int>& out) {
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
    const int b = F.cell_start[ix * GR + nMinCellY], e = F.cell_start[ix * GR + nMaxCellY + 1];
    for (int t = b; t < e; ++t) {
      const int idx = F.cell_items[t];
      const asd_keypoint& kp = F.kps[idx];
      if (check) {
        if (kp.octave < minLevel) continue;
        if (maxLevel >= 0 && kp.octave > maxLevel) continue;
      }
      const float dx = kp.x - x, dy = kp.y - y;
      if (std::fabs(dx) < r && std::fabs(dy) < r) out.push_back(idx);
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

struct MatcherState {
  int q_cap = 0, cand_cap = 0, h_cand_cap = 0, qdesc_cap = 0;   // cand_cap: device candidate buffers; h_cand_cap: their pinned host twins (host replay only)
  int last_total[4] = {1 << 15, 1 << 15, 1 << 15, 1 << 15};  // candidates the previous search of each kind produced
  WinQuery *d_queries = nullptr, *h_queries = nullptr;   // h_* pinned
  int *d_q_off = nullptr, *d_q_cnt = nullptr, *d_total = nullptr, *d_idx = nullptr;
  unsigned* d_meta = nullptr;  // per candidate: position in its list << 16 | query (k_resolve)
  float* d_dist = nullptr;
  int *h_q = nullptr;      // [2*q_cap + 1]: off, cnt, total
  int* h_idx = nullptr;
  float* h_dist = nullptr;
  float *d_qdesc = nullptr, *h_qdesc = nullptr;
  float* d_bank = nullptr;  // device-resident descriptor bank (MapPoint::mDescriptor rows)
  int bank_cap = 0;
  // map-point attribute bank, same row ids: [row][8] = mWorldPos, mNormalVector, mfMinDistance, mfMaxDistance (asd_mpbank_put)
  float* d_attr = nullptr;
  int attr_cap = 0;
  float* d_cxw = nullptr;     // asd_track_local_points_rows: the candidates' positions as k_frustum_queries read them from the bank (the solver's table)
  size_t cxw_cap = 0;
  float* h_attr[2] = {nullptr, nullptr};   // pinned staging, used alternately
  size_t h_attr_cap[2] = {0, 0};
  hipEvent_t ev_attr[2] = {nullptr, nullptr};
  int attr_turn = 0;
  // device-side replay (k_resolve): pinned staging for flags in / matches out
  int* h_res = nullptr;      // [res_cap + 4] ints, then res_cap bytes of flags
  int res_cap = 0;
  bool replay_host = false;  // ASD_MATCH_REPLAY=host at asd_ctx_create
};

MatcherState* mstate(asd_ctx* ctx) {
  if (!ctx->matcher) {
    MatcherState* m = new MatcherState();
    m->replay_host = ctx->match_replay_host;
    ctx->matcher = m;
  }
  return static_cast<MatcherState*>(ctx->matcher);
}

int ensure_queries(asd_ctx* ctx, MatcherState* m, int nq) {
  if (nq <= m->q_cap) return ASD_OK;
  const int cap = std::max(nq * 3 / 2, 8192);
  if (m->d_queries) { (void)hipFree(m->d_queries); (void)hipHostFree(m->h_queries); (void)hipFree(m->d_q_off); (void)hipHostFree(m->h_q); }
  m->q_cap = 0;
  ASD_HIP_CHECK(ctx, hipMalloc(&m->d_queries, (size_t)cap * sizeof(WinQuery)));
  ASD_HIP_CHECK(ctx, hipHostMalloc(&m->h_queries, (size_t)cap * sizeof(WinQuery)));
  ASD_HIP_CHECK(ctx, hipMalloc(&m->d_q_off, ((size_t)2 * cap + 4) * sizeof(int)));  // off | cnt | total
  ASD_HIP_CHECK(ctx, hipHostMalloc(&m->h_q, ((size_t)2 * cap + 4) * sizeof(int)));
  m->q_cap = cap;
  return ASD_OK;
}
// device candidate buffers (k_window_search's output, k_resolve's input): 12 B per candidate
int ensure_cands_dev(asd_ctx* ctx, MatcherState* m, size_t n) {
  if (n <= (size_t)m->cand_cap) return ASD_OK;
  if (n > ((size_t)1 << 30)) { ctx->set_error("candidate buffers: %zu entries asked for", n); return ASD_ERR_CAPACITY; }
  const size_t cap = std::max(n + n / 2, (size_t)1 << 18);
  if (m->d_idx) { (void)hipFree(m->d_idx); (void)hipFree(m->d_dist); (void)hipFree(m->d_meta); }
  m->d_idx = nullptr; m->d_dist = nullptr; m->d_meta = nullptr;
  m->cand_cap = 0;
  // idx / dist twice over: k_window_search<true> writes the sorted copy of a list that took its slow path into the second half
  ASD_HIP_CHECK(ctx, hipMalloc(&m->d_idx, 2 * cap * sizeof(int)));
  ASD_HIP_CHECK(ctx, hipMalloc(&m->d_dist, 2 * cap * sizeof(float)));
  ASD_HIP_CHECK(ctx, hipMalloc(&m->d_meta, cap * sizeof(unsigned)));
  m->cand_cap = (int)std::min(cap, (size_t)0x7fffffff);
  return ASD_OK;
}
// ... and their pinned host twins, which only the host replay (window_search) reads
int ensure_cands(asd_ctx* ctx, MatcherState* m, int n) {
  int rc = ensure_cands_dev(ctx, m, (size_t)n);
  if (rc != ASD_OK) return rc;
  if (m->cand_cap <= m->h_cand_cap) return ASD_OK;
  if (m->h_idx) { (void)hipHostFree(m->h_idx); (void)hipHostFree(m->h_dist); }
  m->h_idx = nullptr; m->h_dist = nullptr;
  m->h_cand_cap = 0;
  ASD_HIP_CHECK(ctx, hipHostMalloc(&m->h_idx, (size_t)m->cand_cap * sizeof(int)));
  ASD_HIP_CHECK(ctx, hipHostMalloc(&m->h_dist, (size_t)m->cand_cap * sizeof(float)));
  m->h_cand_cap = m->cand_cap;
  return ASD_OK;
}
int ensure_qdesc(asd_ctx* ctx, MatcherState* m, int n) {
  if (n <= m->qdesc_cap) return ASD_OK;
  const int cap = std::max(n * 3 / 2, 8192);
  if (m->d_qdesc) { (void)hipFree(m->d_qdesc); (void)hipHostFree(m->h_qdesc); }
  m->qdesc_cap = 0;
  ASD_HIP_CHECK(ctx, hipMalloc(&m->d_qdesc, (size_t)cap * 512));
  ASD_HIP_CHECK(ctx, hipHostMalloc(&m->h_qdesc, (size_t)cap * 512));
  m->qdesc_cap = cap;
  return ASD_OK;
}

int ensure_bank(asd_ctx* ctx, MatcherState* m, int rows) {
  if (rows <= m->bank_cap) return ASD_OK;
  const int cap = std::max(rows * 3 / 2, 16384);
  float* nb = nullptr;
  ASD_HIP_CHECK(ctx, hipMalloc(&nb, (size_t)cap * 512));
  if (m->d_bank) {
    ASD_HIP_CHECK(ctx, hipMemcpyAsync(nb, m->d_bank, (size_t)m->bank_cap * 512, hipMemcpyDeviceToDevice, ctx->stream));
    ASD_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    (void)hipFree(m->d_bank);
  }
  m->d_bank = nb;
  m->bank_cap = cap;
  return ASD_OK;
}

// result of one batched window search: per query q the candidates idx[off[q] .. off[q]+cnt[q]) in
// Frame::GetFeaturesInArea order with their DescriptorDistance to the query's descriptor
struct SearchResult { const int* off; const int* cnt; const int* idx; const float* dist; };

// queries are already in m->h_queries[0..nq); d_q = query descriptor table on the device
// kind: 0 frame-to-frame, 1 local map, 2 fuse, 3 other -- only sizes the speculative read-back
int window_search(asd_ctx* ctx, MatcherState* m, const AsdFrameSlot& F, int nq, const float* d_q, SearchResult* res, int kind = 3) {
  if (asd_track_busy(ctx, "matcher call")) return ASD_ERR_INVALID;
  static const bool timing = getenv("ASD_TIMING") != nullptr;   // per-kind host-side split, printed every 200 searches
  static double tacc[4][3]; static long tcalls[4];
  const auto tw0 = std::chrono::steady_clock::now();
  auto tw1 = tw0;
  int rc = ensure_cands(ctx, m, 1);
  if (rc != ASD_OK) return rc;
  hipStream_t st = ctx->stream;
  // off | cnt | total packed for THIS query count, so one small copy brings all three back
  int* d_off = m->d_q_off;
  int* d_cnt = m->d_q_off + nq;
  int* d_total = m->d_q_off + 2 * nq;
  // candidates copied back together with the counts: a little more than the last search produced
  const int optimistic = std::min(m->cand_cap, std::max(4096, m->last_total[kind] + m->last_total[kind] / 4));
  for (int attempt = 0; attempt < 2; ++attempt) {
    ASD_HIP_CHECK(ctx, hipMemcpyAsync(m->d_queries, m->h_queries, (size_t)nq * sizeof(WinQuery), hipMemcpyHostToDevice, st));
    ASD_HIP_CHECK(ctx, hipMemsetAsync(d_total, 0, sizeof(int), st));
    GridDev G{F.d_kp, F.d_cell_start, F.d_cell_items, F.min_x, F.min_y, F.inv_w, F.inv_h};
    ASD_HIP_CHECK(ctx, hipEventRecord(ctx->ev0, st));
    hipLaunchKernelGGL(k_window_search<false>, dim3((nq + kSearchWaves - 1) / kSearchWaves), dim3(64 * kSearchWaves), 0, st, G, m->d_queries, nq, d_q, F.d_desc, d_off, d_cnt,
                       d_total, m->cand_cap, m->d_idx, m->d_dist, (unsigned*)nullptr, SortArgs{});
    ASD_HIP_CHECK(ctx, hipGetLastError());
    ASD_HIP_CHECK(ctx, hipEventRecord(ctx->ev1, st));
    ASD_HIP_CHECK(ctx, hipMemcpyAsync(m->h_q, d_off, ((size_t)2 * nq + 1) * sizeof(int), hipMemcpyDeviceToHost, st));
    ASD_HIP_CHECK(ctx, hipMemcpyAsync(m->h_idx, m->d_idx, (size_t)optimistic * sizeof(int), hipMemcpyDeviceToHost, st));
    ASD_HIP_CHECK(ctx, hipMemcpyAsync(m->h_dist, m->d_dist, (size_t)optimistic * sizeof(float), hipMemcpyDeviceToHost, st));
    tw1 = std::chrono::steady_clock::now();
    ASD_HIP_CHECK(ctx, hipStreamSynchronize(st));
    const int total = m->h_q[2 * nq];
    m->last_total[kind] = total;
    if (total > m->cand_cap) {  // segment reservation overflowed the buffers: grow and run again
      if ((rc = ensure_cands(ctx, m, total)) != ASD_OK) return rc;
      continue;
    }
    if (total > optimistic) {
      ASD_HIP_CHECK(ctx, hipMemcpyAsync(m->h_idx + optimistic, m->d_idx + optimistic, (size_t)(total - optimistic) * sizeof(int), hipMemcpyDeviceToHost, st));
      ASD_HIP_CHECK(ctx, hipMemcpyAsync(m->h_dist + optimistic, m->d_dist + optimistic, (size_t)(total - optimistic) * sizeof(float), hipMemcpyDeviceToHost, st));
      ASD_HIP_CHECK(ctx, hipStreamSynchronize(st));
    }
    break;
  }
  ASD_HIP_CHECK(ctx, hipEventElapsedTime(&ctx->ms_match, ctx->ev0, ctx->ev1));
  if (timing) {
    const auto tw2 = std::chrono::steady_clock::now();
    auto ms = [](auto a, auto b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
    tacc[kind][0] += ms(tw0, tw1); tacc[kind][1] += ms(tw1, tw2); tacc[kind][2] += ctx->ms_match;
    if (++tcalls[kind] % 200 == 0)
      fprintf(stderr, "[window_search kind %d] enqueue %.3f sync %.3f kernel %.3f ms, %d candidates copied (last total %d)\n", kind,
              tacc[kind][0] / tcalls[kind], tacc[kind][1] / tcalls[kind], tacc[kind][2] / tcalls[kind], optimistic, m->last_total[kind]);
  }
  res->off = m->h_q;
  res->cnt = m->h_q + nq;
  res->idx = m->h_idx;
  res->dist = m->h_dist;
  return ASD_OK;
}


// where the claim / ratio / histogram replay runs: on the device (k_resolve, default) or on the host over the copied-back
// candidate lists (ASD_MATCH_REPLAY=host in the environment of asd_ctx_create; also taken when the tables would
// not fit the workgroup's LDS or there are more than 4096 queries)
size_t resolve_lds_bytes(int kind, int n_cur, int nq) {
  // k_resolve2: the two claim tables, angle / octave table, last-writer table, (local map) the compacted list of map points with candidates
  return resolve2_fixed_lds(kind, n_cur) + (size_t)n_cur * 4 + (kind == 1 ? ((size_t)nq * 2 + 15) / 16 * 16 : 0);
}
bool replay_on_device(const MatcherState* m, int kind, int n_cur, int nq) {
  // kind 0 (frame to frame): every last-frame point is a slot of the replay workgroup; kind 1 (local map): the map points that HAVE candidates are
  // replayed in chunks (resolve2.h), so the count of candidates is bounded only by the compaction's rounds and the 16-bit indices
  const int max_q = kind == 0 ? kResolve2Threads * 4 : std::min(kResolve2MaxRounds * 512, 65535);
  return !m->replay_host && n_cur > 0 && n_cur < 65536 && nq <= max_q && resolve_lds_bytes(kind, n_cur, nq) <= 96 * 1024;
}

// k_window_search + k_resolve<KIND> behind one synchronisation: queries are in m->h_queries[0..nq), flags (may be null) are
// copied through pinned staging, match_cur[n_cur] / *n_matches come back.  kp_last = last frame's device keypoints (KIND 0).
// what a fused chain adds to a search: tables that travel in the search's upload block, and room for its results in the
// search's result block
constexpr int kChainTabs = 8;
struct ChainHook {
  const void* src[kChainTabs] = {};   // host tables that travel in the upload block (world positions, flags, ...)
  size_t bytes[kChainTabs] = {};
  size_t result_bytes = 0;            // room wanted in the result block
  // optional: the queries are made on the device from the uploaded tables (frustum test + window of every map point)
  // instead of being copied from m->h_queries; called after the upload, before the search
  // (h_tab: the same tables inside the pinned upload block; tail: the copy the kernel carries, see UploadTail)
  std::function<int(WinQuery* d_queries, void* const* d_tab, void* const* h_tab, const UploadTail& tail)> prepare;
  // enqueue the chain's kernels: match table, the uploaded tables, where the results go (all device pointers)
  // fused (may be null): the stage's claim replay, to run in FRONT of the solver inside its workgroup (k_resolve_pose) instead of as a kernel of its own
  std::function<int(const int* d_match, void* const* d_tab, void* d_result, const AsdFusedReplay* fused)> enqueue;
  const void* h_result = nullptr;     // out: the chain's results on the host after the call
  bool kp_flags = false;              // out: the form pose_chain_enqueue gave the flags in that block (per keypoint + edge count, or per edge) --
                                      // carried with the chain: the context-wide pose_chain_kp_flags belongs to whichever chain was enqueued last
};

template <int KIND>
int search_and_resolve(asd_ctx* ctx, MatcherState* m, const AsdFrameSlot& F, int nq, const float* d_q, const float4* kp_last,
                       const uint8_t* obs_pos, const uint8_t* occupied, int check_ori, float nn_ratio, int32_t* match_cur,
                       int32_t* n_matches, ChainHook* chain = nullptr, std::function<int()>* defer = nullptr) {
  // One upload block (queries, the zeroed candidate counter, flags, the chain's tables), the kernels back to back, one result
  // block (match table, counters, the chain's results), one synchronisation: a copy costs 15-25 us on this stream whatever
  // its size (rocprof timeline), five of them per call were a third of the call.
  // Split in two: `attempt` enqueues everything (inputs are consumed: they live in the pinned upload block from then on),
  // `complete` synchronises, handles a candidate-buffer overflow (grow, enqueue again) and unpacks.  With `defer` the caller gets
  // `complete` back instead of having it run (asd_track_async / asd_track_finish); `chain` must then outlive it.
  if (asd_track_busy(ctx, "matcher call")) return ASD_ERR_INVALID;
  const int n_cur = F.n;
  // A deferred completion (asd_track_async) runs after the caller may have rewritten bank rows and other frame slots
  // (include/asd_slam.h allows asd_bank_put* / asd_frame_set in between), so it must never have to search again: the candidate
  // buffers are sized for the worst case up front -- a query's list holds at most every keypoint of the frame -- and an
  // overflow at completion is an error, not a retry.  (12 B x nq x n_cur: 96 MB at 4000 x 2000; 288 GB of HBM make that a non-issue.)
  const bool deferred = defer != nullptr;
  int rc = ensure_cands_dev(ctx, m, deferred ? std::max((size_t)nq * (size_t)std::max(n_cur, 1), (size_t)1) : (size_t)1);
  if (rc != ASD_OK) return rc;
  hipStream_t st = ctx->stream;
  const AsdFrameSlot* Fp = &F;
  {
    AsdXfer &up = ctx->up, &down = ctx->down;
    size_t extra = 0;
    if (chain) for (int i = 0; i < kChainTabs; ++i) extra += chain->bytes[i] + 256;
    ASD_HIP_CHECK(ctx, up.begin(st, (size_t)nq * sizeof(WinQuery) + 256 + (size_t)nq + (size_t)n_cur + 1024 + extra));
    ASD_HIP_CHECK(ctx, down.begin(st, ((size_t)n_cur + 16) * sizeof(int) + 256 + (chain ? chain->result_bytes : 0)));
    ASD_HIP_CHECK(ctx, ctx->scratch.reserve(AsdDevBuf::padded((size_t)nq * 4) + AsdDevBuf::padded((size_t)nq * kTop * 2)));
  }
  int* d_pick = ctx->scratch.carve<int>(nq);
  uint16_t* d_top_idx = ctx->scratch.carve<uint16_t>((size_t)nq * kTop);
  const bool dev_queries = chain && chain->prepare;
  const size_t o_q = dev_queries ? ctx->up.reserve((size_t)nq * sizeof(WinQuery)) : ctx->up.add(m->h_queries, (size_t)nq * sizeof(WinQuery));
  const size_t o_total = ctx->up.zeros(sizeof(int));
  const bool has_obs = obs_pos != nullptr;
  const size_t o_obs = has_obs ? ctx->up.add(obs_pos, nq) : 0;
  const size_t o_occ = KIND == 1 ? ctx->up.add(occupied, n_cur) : 0;
  std::array<size_t, kChainTabs> o_tab{};
  std::array<bool, kChainTabs> has_tab{};
  if (chain) for (int i = 0; i < kChainTabs; ++i) if (chain->src[i]) { o_tab[i] = ctx->up.add(chain->src[i], chain->bytes[i]); has_tab[i] = true; }
  const size_t o_out = ctx->down.reserve(((size_t)n_cur + 16) * sizeof(int));
  const size_t o_res = chain ? ctx->down.reserve(chain->result_bytes) : 0;

  auto attempt = [=]() -> int {
    AsdXfer &up = ctx->up, &down = ctx->down;
    const AsdFrameSlot& F = *Fp;
    int* d_out = down.dev<int>(o_out);
    int* d_off = m->d_q_off;
    int* d_cnt = m->d_q_off + nq;
    int* d_total = up.dev<int>(o_total);
    int rc;
    // ASD_UPLOAD_COPY=1 (copy commands instead of copy kernels) makes the upload a command of its own
    static const bool tail_upload = getenv("ASD_UPLOAD_COPY") == nullptr;
    const bool carried = dev_queries && tail_upload;
    if (!carried) ASD_HIP_CHECK(ctx, up.upload(st));
    void *d_tab[kChainTabs], *h_tab[kChainTabs];
    for (int i = 0; i < kChainTabs; ++i) {
      d_tab[i] = has_tab[i] ? up.dev<void>(o_tab[i]) : nullptr;
      h_tab[i] = has_tab[i] ? (carried ? up.host<void>(o_tab[i]) : up.dev<void>(o_tab[i])) : nullptr;
    }
    if (dev_queries) {
      // the query table is the block's first reservation: the copy starts behind it
      const UploadTail tail{reinterpret_cast<const uint4*>(up.h), reinterpret_cast<uint4*>(up.d), carried ? ((size_t)nq * sizeof(WinQuery) + 255) / 256 * 16 : 0,
                            carried ? (up.used + 15) / 16 : 0, 0};
      if ((rc = chain->prepare(up.dev<WinQuery>(o_q), d_tab, h_tab, tail)) != ASD_OK) return rc;
    }
    GridDev G{F.d_kp, F.d_cell_start, F.d_cell_items, F.min_x, F.min_y, F.inv_w, F.inv_h};
    // (the "match" stage clock of asd_last_stage_ms: two timed event records per chain, each a barrier packet on the stream -- only
    // when somebody asked for timings)
    static const bool stage_timing = getenv("ASD_TIMING") != nullptr;
    if (stage_timing) ASD_HIP_CHECK(ctx, hipEventRecord(ctx->ev0, st));
    constexpr bool zero_copy = true;   // results stored by the kernels straight into the pinned block
    {
      // the lists in preference order (k_window_search<true>), the replay over their heads (k_resolve2)
      SortArgs sa{KIND == 1 ? up.dev<uint8_t>(o_occ) : nullptr, KIND == 0 ? TH_HIGH : __builtin_huge_valf(), d_top_idx};
      hipLaunchKernelGGL(k_window_search<true>, dim3((nq + kSearchWaves - 1) / kSearchWaves), dim3(64 * kSearchWaves), 0, st, G, up.dev<WinQuery>(o_q), nq, d_q, F.d_desc, d_off,
                         d_cnt, d_total, m->cand_cap, m->d_idx, m->d_dist, (unsigned*)nullptr, sa);
      ASD_HIP_CHECK(ctx, hipGetLastError());
      Resolve2Args a{};
      a.nq = nq; a.n_cur = n_cur;
      a.q_off = d_off; a.q_cnt = d_cnt; a.idx = m->d_idx; a.dist = m->d_dist; a.top_idx = d_top_idx;
      a.total = d_total; a.cap = m->cand_cap;
      a.obs_pos = has_obs ? up.dev<uint8_t>(o_obs) : nullptr;
      a.kp_cur = F.d_kp; a.kp_last = kp_last;
      a.check_ori = check_ori; a.nn_ratio = nn_ratio;
      a.match_cur = d_out; a.n_matches = d_out + n_cur;
      a.mirror = zero_copy ? down.host<int>(o_out) : nullptr;
      // the sorted lists' copy in LDS (for queries whose four best are all held): what the previous search of this kind produced and
      // a quarter on top, as far as the 96 KB this kernel may ask for allow; entries beyond it are read from global memory
      const size_t fixed = resolve_lds_bytes(KIND, n_cur, nq), per = KIND == 1 ? 6 : 2;
      a.stage_cap = (int)std::min<size_t>(((size_t)m->last_total[KIND] * 5 / 4 + 1023) / 1024 * 1024, ((size_t)96 * 1024 - fixed) / per / 8 * 8);
      const size_t lds = fixed + (size_t)a.stage_cap * per;
      auto launch = [&](auto kern) -> hipError_t {
        static AsdPerDeviceOnce attr_set;   // per instantiation and device: more than 64 KB of dynamic LDS has to be asked for once
        if (attr_set.need(ctx->cfg.device)) {
          const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
          if (e != hipSuccess) return e;
          attr_set.done(ctx->cfg.device);
        }
        hipLaunchKernelGGL(kern, dim3(1), dim3(kResolve2Threads), lds, st, a);
        return hipGetLastError();
      };
      // With a PoseOptimization chained behind the search, replay and solver are ONE workgroup (k_resolve_pose: resolve2_body on the solver's
      // threads, then the solver) where the stage's tables fit: as two kernels the solver waits 35-55 us for a CU of its own behind the replay,
      // beside the extractor's ASDNet workgroups (round 4 had this form in asd_track_frame only; the two-call form is the headline since round 5)
      const bool fuse = chain && pose_chain_fused_ok(ctx, KIND, nq, n_cur, lds);
      if (fuse) {
        const AsdFusedReplay fr{&a, KIND, nq, lds};
        if ((rc = chain->enqueue(d_out, d_tab, down.host<void>(o_res), &fr)) != ASD_OK) return rc;
        chain->h_result = down.host<void>(o_res);
        chain->kp_flags = ctx->pose_chain_kp_flags;
        if (stage_timing) ASD_HIP_CHECK(ctx, hipEventRecord(ctx->ev1, st));   // (the "match" stage clock then covers the solver too)
      } else {
        if (nq <= 2 * kResolve2Threads) ASD_HIP_CHECK(ctx, launch(k_resolve2<KIND, 2>));
        else ASD_HIP_CHECK(ctx, launch(k_resolve2<KIND, 4>));
        if (stage_timing) ASD_HIP_CHECK(ctx, hipEventRecord(ctx->ev1, st));
        if (chain) {
          if ((rc = chain->enqueue(d_out, d_tab, zero_copy ? down.host<void>(o_res) : down.dev<void>(o_res), nullptr)) != ASD_OK) return rc;
          chain->h_result = down.host<void>(o_res);
          chain->kp_flags = ctx->pose_chain_kp_flags;
        }
      }
    }
    if (!zero_copy) ASD_HIP_CHECK(ctx, down.download(st));
    // the completion waits for THIS point of the stream, not for the stream: a split-phase caller enqueues the next frame's grid
    // and descriptor copies behind the chain, and they are not part of its result
    if (!ctx->ev_chain) ASD_HIP_CHECK(ctx, hipEventCreateWithFlags(&ctx->ev_chain, hipEventDisableTiming));
    ASD_HIP_CHECK(ctx, hipEventRecord(ctx->ev_chain, st));
    return ASD_OK;
  };

  auto complete = [=]() -> int {
    const int* h_out = ctx->down.host<int>(o_out);
    int rc;
    for (int round = 0;; ++round) {
      ASD_HIP_CHECK(ctx, hipEventSynchronize(ctx->ev_chain));
      const int total = h_out[n_cur + 1];
      m->last_total[KIND] = total;
      if (total > m->cand_cap && deferred) {   // cannot happen with the worst-case sizing above; never search again over state the caller may have changed
        ctx->set_error("asd_track_finish: %d candidates overflowed the %d-entry buffers of a deferred stage", total, m->cand_cap);
        return ASD_ERR_CAPACITY;
      }
      if (total > m->cand_cap && round == 0) {  // the candidate buffers overflowed (k_resolve did not run): grow and search again
        if ((rc = ensure_cands_dev(ctx, m, (size_t)total)) != ASD_OK) return rc;
        *ctx->up.host<int>(o_total) = 0;
        if ((rc = attempt()) != ASD_OK) return rc;
        continue;
      }
      break;
    }
    static const bool stage_timing = getenv("ASD_TIMING") != nullptr;
    if (stage_timing) ASD_HIP_CHECK(ctx, hipEventElapsedTime(&ctx->ms_match, ctx->ev0, ctx->ev1));
    else ctx->ms_match = 0.f;
    memcpy(match_cur, h_out, (size_t)n_cur * sizeof(int));
    *n_matches = h_out[n_cur];
    static const bool timing = getenv("ASD_TIMING") != nullptr;
    if (timing) {
      static double acc[2]; static long calls[2]; static long rounds[2]; static double st_us[2][8];
      acc[KIND] += ctx->ms_match; rounds[KIND] += h_out[n_cur + 2];
      for (int i = 0; i < 8; ++i) st_us[KIND][i] += 0.01 * h_out[n_cur + 3 + i];
      if (++calls[KIND] % 200 == 0)
        fprintf(stderr, "[search+resolve kind %d] device %.3f ms, %.1f iterations, %d candidates; replay: staging %.1f us, iterations %.1f us (the first %.1f; thread 0 work %.1f, barrier + verdict %.1f), outputs %.1f us\n", KIND,
                acc[KIND] / calls[KIND], (double)rounds[KIND] / calls[KIND], h_out[n_cur + 1], st_us[KIND][0] / calls[KIND], st_us[KIND][1] / calls[KIND],
                st_us[KIND][3] / calls[KIND], st_us[KIND][4] / calls[KIND], st_us[KIND][5] / calls[KIND], st_us[KIND][2] / calls[KIND]);
    }
    return ASD_OK;
  };

  if ((rc = attempt()) != ASD_OK) return rc;
  if (defer) { *defer = complete; return ASD_OK; }
  return complete();
}

// query descriptors: host table -> pinned staging -> device (one async copy)
int upload_qdesc(asd_ctx* ctx, MatcherState* m, const float* desc, int n) {
  int rc = ensure_qdesc(ctx, m, n);
  if (rc != ASD_OK) return rc;
  memcpy(m->h_qdesc, desc, (size_t)n * 512);
  ASD_HIP_CHECK(ctx, hipMemcpyAsync(m->d_qdesc, m->h_qdesc, (size_t)n * 512, hipMemcpyHostToDevice, ctx->stream));
  return ASD_OK;
}

AsdFrameSlot* slot_of(asd_ctx* ctx, int s) {
  if (!ctx || s < 0 || s >= ASD_MAX_FRAMES) return nullptr;
  return &ctx->frames[s];
}
}  // namespace

hipError_t asd_copy_rows(hipStream_t st, void* dst, const void* src, size_t bytes) { return copy_rows(st, dst, src, bytes); }

void matcher_free(asd_ctx* ctx) {
  if (ctx->matcher && static_cast<MatcherState*>(ctx->matcher)->d_cxw) (void)hipFree(static_cast<MatcherState*>(ctx->matcher)->d_cxw);
  for (auto& f : ctx->frames) {
    if (f.d_desc) (void)hipFree(f.d_desc);
    if (f.d_fv) (void)hipFree(f.d_fv);
    f.d_fv = nullptr;
    if (f.d_kp) (void)hipFree(f.d_kp);   // one block: keypoints, cell offsets, cell items
    if (f.h_stage) (void)hipHostFree(f.h_stage);
    if (f.ev_staged) (void)hipEventDestroy(f.ev_staged);
    f.ev_staged = nullptr;
    f.d_desc = nullptr; f.d_kp = nullptr; f.d_cell_start = nullptr; f.d_cell_items = nullptr; f.h_stage = nullptr;
  }
  if (ctx->matcher) {
    MatcherState* m = static_cast<MatcherState*>(ctx->matcher);
    void* dev[] = {m->d_queries, m->d_q_off, m->d_idx, m->d_dist, m->d_meta, m->d_qdesc, m->d_bank, m->d_attr};
    for (void* p : dev) if (p) (void)hipFree(p);
    void* host[] = {m->h_queries, m->h_q, m->h_idx, m->h_dist, m->h_qdesc, m->h_res, m->h_attr[0], m->h_attr[1]};
    for (void* p : host) if (p) (void)hipHostFree(p);
    for (hipEvent_t e : m->ev_attr) if (e) (void)hipEventDestroy(e);
    delete m;
    ctx->matcher = nullptr;
  }
}

extern "C" {

static int frame_set_impl(asd_ctx* ctx, int32_t slot, const asd_keypoint* kps, const float* desc, int32_t n, float min_x,
                          float max_x, float min_y, float max_y, asd_ctx* src);
int asd_frame_set(asd_ctx* ctx, int32_t slot, const asd_keypoint* kps, const float* desc, int32_t n, float min_x,
                  float max_x, float min_y, float max_y) {
  return frame_set_impl(ctx, slot, kps, desc, n, min_x, max_x, min_y, max_y, ctx);
}
// the frame's descriptors adopted from ANOTHER context's last extraction (same device): the right image of a stereo pair is extracted by
// its own context (the reference keeps two extractors) and matched inside the left one's
int asd_frame_set_from_ctx(asd_ctx* ctx, int32_t slot, const asd_keypoint* kps, int32_t n, float min_x, float max_x, float min_y, float max_y,
                           asd_ctx* src) {
  if (!ctx || !src || src->cfg.device != ctx->cfg.device) return ASD_ERR_INVALID;
  return frame_set_impl(ctx, slot, kps, nullptr, n, min_x, max_x, min_y, max_y, src);
}
static int frame_set_impl(asd_ctx* ctx, int32_t slot, const asd_keypoint* kps, const float* desc, int32_t n, float min_x,
                          float max_x, float min_y, float max_y, asd_ctx* src) {
  AsdFrameSlot* F = slot_of(ctx, slot);
  if (!F || n < 0 || (n > 0 && !kps) || !(max_x > min_x) || !(max_y > min_y)) return ASD_ERR_INVALID;
  if (n > ctx->cfg.max_patches) { ctx->set_error("frame has %d keypoints, capacity %d", n, ctx->cfg.max_patches); return ASD_ERR_CAPACITY; }
  if (!desc && !src->d_desc_last) { ctx->set_error("desc == NULL: no extraction has completed on the source context yet"); return ASD_ERR_INVALID; }
  if (!desc && n != src->last_n) { ctx->set_error("desc == NULL adopts the last extract (%d keypoints), got n=%d", src->last_n, n); return ASD_ERR_INVALID; }
  (void)hipSetDevice(ctx->cfg.device);
  const size_t cap = ctx->cfg.max_patches;
  if (!F->d_desc) {
    ASD_HIP_CHECK(ctx, hipMalloc(&F->d_desc, cap * 128 * sizeof(float)));
    // keypoints, cell offsets and cell items in ONE device block laid out like the pinned staging buffer: one copy per frame
    char* blk = nullptr;
    ASD_HIP_CHECK(ctx, hipMalloc(&blk, cap * (sizeof(float4) + sizeof(int)) + (GC * GR + 1) * sizeof(int) + 16));
    F->d_kp = reinterpret_cast<float4*>(blk);
    F->d_cell_start = reinterpret_cast<int32_t*>(blk + cap * sizeof(float4));
    F->d_cell_items = F->d_cell_start + (GC * GR + 1);
    ASD_HIP_CHECK(ctx, hipHostMalloc(&F->h_stage, cap * (sizeof(float4) + sizeof(int)) + (GC * GR + 1) * sizeof(int) + 16));
    ASD_HIP_CHECK(ctx, hipEventCreateWithFlags(&F->ev_staged, hipEventDisableTiming));
  } else {
    // the slot's pinned staging buffer is about to be rewritten: its previous copies (a frame or more ago) must have left it
    ASD_HIP_CHECK(ctx, hipEventSynchronize(F->ev_staged));
  }
  hipStream_t st = asd_prep_stream(ctx);
  if (n > 0) {
    if (desc) ASD_HIP_CHECK(ctx, hipMemcpyAsync(F->d_desc, desc, (size_t)n * 128 * sizeof(float), hipMemcpyHostToDevice, st));
    else ASD_HIP_CHECK(ctx, copy_rows(st, F->d_desc, src->d_desc_last, (size_t)n * 128 * sizeof(float)));
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
  // device mirror of keypoints + grid (pinned staging, three small async copies)
  float4* hk = reinterpret_cast<float4*>(F->h_stage);
  int* hs = reinterpret_cast<int*>(F->h_stage + cap * sizeof(float4));
  int* hi = hs + (GC * GR + 1);
  for (int i = 0; i < n; ++i) {
    float ob;
    memcpy(&ob, &kps[i].octave, sizeof ob);  // octave travels as raw int bits in .z
    hk[i] = make_float4(kps[i].x, kps[i].y, ob, kps[i].angle);  // .w = angle (rotation histogram of k_resolve)
  }
  memcpy(hs, F->cell_start.data(), (GC * GR + 1) * sizeof(int));
  if (!F->cell_items.empty()) memcpy(hi, F->cell_items.data(), F->cell_items.size() * sizeof(int));
  // one copy for the three arrays (the block up to the last cell item in use; unused keypoint rows travel along: 16 B each)
  {
    static const bool by_kernel = getenv("ASD_UPLOAD_COPY") == nullptr;   // ASD_UPLOAD_COPY=1: copy commands instead
    const size_t bytes = cap * sizeof(float4) + (GC * GR + 1 + F->cell_items.size()) * sizeof(int);
    if (by_kernel) ASD_HIP_CHECK(ctx, copy_rows(st, F->d_kp, hk, (bytes + 15) / 16 * 16));
    else ASD_HIP_CHECK(ctx, hipMemcpyAsync(F->d_kp, hk, bytes, hipMemcpyHostToDevice, st));
  }
  // desc == NULL (the per-frame path): no synchronisation -- every consumer of the slot is enqueued on this stream behind the
  // copies; ev_staged guards the slot's pinned staging buffer, ev_adopt the extraction buffer the descriptors are copied
  // out of (asd_extract_submit waits for it before the worker may reuse that buffer on its own streams).
  // desc != NULL: the caller's buffer is ordinary host memory and may be freed or rewritten as soon as we return.
  ASD_HIP_CHECK(ctx, hipEventRecord(F->ev_staged, st));
  if (!desc && n > 0) {   // (the SOURCE context's extractor must not reuse that buffer before the copy has left it)
    if (!src->ev_adopt) ASD_HIP_CHECK(ctx, hipEventCreateWithFlags(&src->ev_adopt, hipEventDisableTiming));
    ASD_HIP_CHECK(ctx, hipEventRecord(src->ev_adopt, st));
    src->adopt_pending = true;
  } else {
    ASD_HIP_CHECK(ctx, hipStreamSynchronize(st));
  }
  return ASD_OK;
}

int asd_frame_features_in_area(asd_ctx* ctx, int32_t slot, float x, float y, float r, int32_t min_level,
                               int32_t max_level, int32_t capacity, int32_t* idx_out, int32_t* n_out) {
  AsdFrameSlot* F = slot_of(ctx, slot);
  if (!F || !idx_out || !n_out) return ASD_ERR_INVALID;
  // single query through the same kernel the matchers use (so the test of this entry point is a
  // test of the device grid walk); the host mirror cross-checks it
  MatcherState* m = mstate(ctx);
  int rc = ensure_queries(ctx, m, 1);
  if (rc != ASD_OK) return rc;
  if (F->n == 0) { *n_out = 0; return ASD_OK; }
  m->h_queries[0] = WinQuery{x, y, r, min_level, max_level, 0};
  SearchResult res;
  if ((rc = window_search(ctx, m, *F, 1, F->d_desc, &res)) != ASD_OK) return rc;
  std::vector<int> host;
  features_in_area(*F, x, y, r, min_level, max_level, host);
  if ((int)host.size() != res.cnt[0] || !std::equal(host.begin(), host.end(), res.idx + res.off[0])) {
    ctx->set_error("device grid walk disagrees with the host mirror");
    return ASD_ERR_HIP;
  }
  const int n = std::min(res.cnt[0], capacity);
  for (int i = 0; i < n; ++i) idx_out[i] = res.idx[res.off[0] + i];
  *n_out = n;
  return ASD_OK;
}

int asd_dist_matrix(asd_ctx* ctx, const float* a, int32_t na, const float* b, int32_t nb, float* out) {
  if (!ctx || na < 0 || nb < 0 || ((na > 0 && nb > 0) && (!a || !b || !out))) return ASD_ERR_INVALID;
  if (na == 0 || nb == 0) return ASD_OK;
  if (asd_track_busy(ctx, "asd_dist_matrix")) return ASD_ERR_INVALID;
  (void)hipSetDevice(ctx->cfg.device);
  float *da = nullptr, *db = nullptr, *dout = nullptr;
  hipStream_t st = ctx->stream;
  ASD_HIP_CHECK(ctx, ctx->scratch.reserve(AsdDevBuf::padded((size_t)na * 512) + AsdDevBuf::padded((size_t)nb * 512) +
                                          AsdDevBuf::padded((size_t)na * nb * sizeof(float))));
  da = ctx->scratch.carve<float>((size_t)na * 128);
  db = ctx->scratch.carve<float>((size_t)nb * 128);
  dout = ctx->scratch.carve<float>((size_t)na * nb);
  ASD_HIP_CHECK(ctx, hipMemcpyAsync(da, a, (size_t)na * 512, hipMemcpyHostToDevice, st));
  ASD_HIP_CHECK(ctx, hipMemcpyAsync(db, b, (size_t)nb * 512, hipMemcpyHostToDevice, st));
  ASD_HIP_CHECK(ctx, hipEventRecord(ctx->ev0, st));
  hipLaunchKernelGGL(k_dist_matrix, dim3((nb + 255) / 256, (na + 15) / 16), dim3(256), 0, st, da, na, db, nb, dout);
  ASD_HIP_CHECK(ctx, hipGetLastError());
  ASD_HIP_CHECK(ctx, hipEventRecord(ctx->ev1, st));
  ASD_HIP_CHECK(ctx, hipMemcpyAsync(out, dout, (size_t)na * nb * sizeof(float), hipMemcpyDeviceToHost, st));
  ASD_HIP_CHECK(ctx, hipStreamSynchronize(st));
  ASD_HIP_CHECK(ctx, hipEventElapsedTime(&ctx->ms_match, ctx->ev0, ctx->ev1));
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

int asd_distinctive_descriptor_batch(asd_ctx* ctx, int32_t n_sets, const int32_t* set_start, const float* desc, int32_t* best_idx) {
  if (!ctx || n_sets < 0 || (n_sets > 0 && (!set_start || !desc || !best_idx))) return ASD_ERR_INVALID;
  if (n_sets == 0) return ASD_OK;
  if (set_start[0] != 0) { ctx->set_error("asd_distinctive_descriptor_batch: set_start[0] must be 0"); return ASD_ERR_INVALID; }
  bool any_big = false;
  for (int s = 0; s < n_sets; ++s) {
    const int n = set_start[s + 1] - set_start[s];
    if (n < 1) { ctx->set_error("asd_distinctive_descriptor_batch: set %d is empty", s); return ASD_ERR_INVALID; }
    any_big |= n > kDistinctMax;
  }
  (void)hipSetDevice(ctx->cfg.device);
  hipStream_t st = ctx->stream;
  const int total = set_start[n_sets];
  hipError_t e = ctx->scratch.reserve(AsdDevBuf::padded((size_t)total * 512) + 2 * AsdDevBuf::padded((size_t)(n_sets + 1) * sizeof(int)));
  float* dd = ctx->scratch.carve<float>((size_t)total * 128);
  int* ds = ctx->scratch.carve<int>((size_t)n_sets + 1);
  int* db = ctx->scratch.carve<int>((size_t)n_sets + 1);
  if (e == hipSuccess) e = hipMemcpyAsync(dd, desc, (size_t)total * 512, hipMemcpyHostToDevice, st);
  if (e == hipSuccess) e = hipMemcpyAsync(ds, set_start, (size_t)(n_sets + 1) * sizeof(int), hipMemcpyHostToDevice, st);
  if (e == hipSuccess) e = hipMemsetAsync(db, 0, (size_t)n_sets * sizeof(int), st);
  if (e == hipSuccess) {
    static AsdPerDeviceOnce attr;
    const size_t lds = (size_t)(kDistinctMax * 132 + kDistinctMax * (kDistinctMax + 1) + kDistinctMax) * sizeof(float);
    if (attr.need(ctx->cfg.device)) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_distinctive), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); attr.done(ctx->cfg.device); }
    // one launch per run of map points that fit a workgroup (normally a single launch over all of them)
    for (int s = 0; s < n_sets;) {
      if (set_start[s + 1] - set_start[s] > kDistinctMax) { ++s; continue; }
      int r = s;
      while (r < n_sets && set_start[r + 1] - set_start[r] <= kDistinctMax) ++r;
      hipLaunchKernelGGL(k_distinctive, dim3(r - s), dim3(256), lds, st, dd, ds + s, db + s);
      s = r;
    }
    e = hipGetLastError();
  }
  if (e == hipSuccess) e = hipMemcpyAsync(best_idx, db, (size_t)n_sets * sizeof(int), hipMemcpyDeviceToHost, st);
  if (e == hipSuccess) e = hipStreamSynchronize(st);
  if (e != hipSuccess) { ctx->set_error("asd_distinctive_descriptor_batch: %s", hipGetErrorString(e)); return ASD_ERR_HIP; }
  if (any_big)  // map points with more observations than one workgroup stages: through the all-pairs kernel, one by one
    for (int s = 0; s < n_sets; ++s) {
      const int n = set_start[s + 1] - set_start[s];
      if (n <= kDistinctMax) continue;
      const int rc = asd_distinctive_descriptor(ctx, desc + (size_t)set_start[s] * 128, n, best_idx + s);
      if (rc != ASD_OK) return rc;
    }
  return ASD_OK;
}

}  // extern "C"

// mp_desc: host table indexed like the last frame's keypoints, or NULL with mp_rows = bank rows
static int match_project_frame_impl(asd_ctx* ctx, int32_t slot_cur, int32_t slot_last, const uint8_t* has_mp, const float* Xw,
                            const float* mp_desc, const int32_t* mp_rows, const float* Tcw, const float* K, float th,
                            int32_t check_orientation, int32_t* match_cur, int32_t* n_matches, const uint8_t* obs_pos,
                            ChainHook* chain = nullptr, bool* chained = nullptr, std::function<int()>* defer = nullptr) {
  if (chained) *chained = false;
  AsdFrameSlot *C = slot_of(ctx, slot_cur), *L = slot_of(ctx, slot_last);
  if (!C || !L || !has_mp || !Xw || (!mp_desc && !mp_rows) || !Tcw || !K || !match_cur || !n_matches) return ASD_ERR_INVALID;
  (void)hipSetDevice(ctx->cfg.device);
  MatcherState* m = mstate(ctx);
  std::fill(match_cur, match_cur + C->n, -1);
  *n_matches = 0;
  if (L->n == 0 || C->n == 0) return ASD_OK;
  int rc = ensure_queries(ctx, m, L->n);
  if (rc != ASD_OK) return rc;
  if (mp_desc) { if ((rc = upload_qdesc(ctx, m, mp_desc, L->n)) != ASD_OK) return rc; }
  else {
    for (int i = 0; i < L->n; ++i)
      if (has_mp[i] && (mp_rows[i] < 0 || mp_rows[i] >= m->bank_cap)) { ctx->set_error("bank row %d out of range", mp_rows[i]); return ASD_ERR_INVALID; }
  }
  static const bool timing = getenv("ASD_TIMING") != nullptr;
  static double tacc[3]; static long tcalls;
  const auto tm0 = std::chrono::steady_clock::now();
  const float fx = K[0], fy = K[1], cx = K[2], cy = K[3];
  const bool on_device = replay_on_device(m, 0, C->n, L->n);
  if (chain && !on_device) chain->prepare = nullptr;   // host replay: the queries are needed here
  if (!(chain && chain->prepare))                      // (else: k_project_queries writes them on the device, asd_track_motion_model)
  for (int i = 0; i < L->n; ++i) {  // projection, ORBmatcher.cc:1343-1368
    WinQuery& Q = m->h_queries[i];
    Q = WinQuery{0.f, 0.f, 0.f, 0, 0, -1};
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
    Q = WinQuery{u, v, th * ctx->scale[oct], oct - 1, oct + 1, mp_desc ? i : mp_rows[i]};
  }
  if (on_device)   // search + claims + rotation histogram on the device, one synchronisation, 4 B per keypoint back
  {
    if (chained) *chained = chain != nullptr;
    return search_and_resolve<0>(ctx, m, *C, L->n, mp_desc ? m->d_qdesc : m->d_bank, L->d_kp, obs_pos, nullptr, check_orientation, 0.f,
                                 match_cur, n_matches, chain, defer);
  }
  SearchResult R;
  const auto tm1 = std::chrono::steady_clock::now();
  if ((rc = window_search(ctx, m, *C, L->n, mp_desc ? m->d_qdesc : m->d_bank, &R, 0)) != ASD_OK) return rc;
  const auto tm2 = std::chrono::steady_clock::now();
  int nmatches = 0;
  // rotation histogram (ORBmatcher.cc:1419-1425, 1437-1450) without per-bin vectors: a current keypoint is matched at most
  // once, so its bin is kept per keypoint and the bins only need their counts
  static thread_local std::vector<int8_t> bin_of;
  // with map points that have no observations a keypoint can be written -- and enter the histogram -- more than once
  // (ORBmatcher.cc:1392-1395): those calls keep the reference's per-write entries (keypoint, bin)
  static thread_local std::vector<std::pair<int, int8_t>> writes;
  if (obs_pos) writes.clear();
  int cnt[HISTO];
  for (int b = 0; b < HISTO; ++b) cnt[b] = 0;
  if (check_orientation) bin_of.assign(C->n, -1);
  constexpr int kAhead = 12;  // the lists were just written by DMA: every line is a DRAM miss, and segments sit in launch order, not query order
  for (int i = 0; i < L->n; ++i) {
    if (i + kAhead < L->n && R.cnt[i + kAhead] > 0) {
      __builtin_prefetch(R.idx + R.off[i + kAhead]);
      __builtin_prefetch(R.dist + R.off[i + kAhead]);
    }
    if (R.cnt[i] == 0) continue;
    float best = 100;
    int best_idx = -1;
    const int* idx = R.idx + R.off[i];
    const float* dist = R.dist + R.off[i];
    for (int t = 0, e = R.cnt[i]; t < e; ++t) {
      const int j = idx[t];
      if (match_cur[j] >= 0 && (!obs_pos || obs_pos[match_cur[j]])) continue;  // holds a map point with Observations() > 0
      if (dist[t] < best) { best = dist[t]; best_idx = j; }
    }
    if (best <= TH_HIGH) {
      match_cur[best_idx] = i;
      nmatches++;
      if (check_orientation) {
        const int b = rot_bin(L->kps[i].angle, C->kps[best_idx].angle);
        if (obs_pos) writes.emplace_back(best_idx, (int8_t)b);
        else bin_of[best_idx] = (int8_t)b;
        ++cnt[b];
      }
    }
  }
  if (check_orientation) {
    int i1, i2, i3;
    three_maxima(cnt, i1, i2, i3);
    if (obs_pos) {
      for (const auto& w : writes)
        if (w.second != i1 && w.second != i2 && w.second != i3) { match_cur[w.first] = -1; nmatches--; }
    } else {
      for (int j = 0; j < C->n; ++j) {
        const int b = bin_of[j];
        if (b >= 0 && b != i1 && b != i2 && b != i3) { match_cur[j] = -1; nmatches--; }
      }
    }
  }
  *n_matches = nmatches;
  if (timing) {
    const auto tm3 = std::chrono::steady_clock::now();
    auto ms = [](auto a, auto b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
    tacc[0] += ms(tm0, tm1); tacc[1] += ms(tm1, tm2); tacc[2] += ms(tm2, tm3);
    if (++tcalls % 200 == 0)
      fprintf(stderr, "[match_project_frame] project %.3f search %.3f select %.3f ms\n", tacc[0] / tcalls, tacc[1] / tcalls, tacc[2] / tcalls);
  }
  return ASD_OK;
}

static int match_project_points_impl(asd_ctx* ctx, int32_t slot_cur, int32_t n_mp, const uint8_t* in_view, const float* proj,
                             const int32_t* level, const float* view_cos, const float* desc, const int32_t* rows,
                             const uint8_t* occupied, float th, float nn_ratio, int32_t* match_cur, int32_t* n_matches,
                             const uint8_t* obs_pos, ChainHook* chain = nullptr, bool* chained = nullptr, std::function<int()>* defer = nullptr) {
  if (chained) *chained = false;
  AsdFrameSlot* F = slot_of(ctx, slot_cur);
  if (!F || n_mp < 0 || !match_cur || !n_matches || (n_mp > 0 && (!in_view || !proj || !level || !view_cos || (!desc && !rows))) ||
      (F->n > 0 && !occupied))
    return ASD_ERR_INVALID;
  (void)hipSetDevice(ctx->cfg.device);
  MatcherState* m = mstate(ctx);
  std::fill(match_cur, match_cur + F->n, -1);
  *n_matches = 0;
  if (n_mp == 0 || F->n == 0) return ASD_OK;
  int rc = ensure_queries(ctx, m, n_mp);
  if (rc != ASD_OK) return rc;
  const bool bFactor = th != 1.0;
  for (int q = 0; q < n_mp; ++q) {
    WinQuery& Q = m->h_queries[q];
    Q = WinQuery{0.f, 0.f, 0.f, 0, 0, -1};
    if (!in_view[q]) continue;
    const int lvl = level[q];
    if (lvl < 0 || lvl >= ctx->cfg.n_levels) { ctx->set_error("map point %d: level %d out of range", q, lvl); return ASD_ERR_INVALID; }
    float r = view_cos[q] > 0.998 ? 2.5f : 4.0f;  // RadiusByViewingCos (:126-132)
    if (bFactor) r *= th;
    if (!desc && (rows[q] < 0 || rows[q] >= m->bank_cap)) { ctx->set_error("bank row %d out of range", rows[q]); return ASD_ERR_INVALID; }
    Q = WinQuery{proj[2 * q], proj[2 * q + 1], r * ctx->scale[lvl], lvl - 1, lvl, desc ? q : rows[q]};
  }
  if (desc && (rc = upload_qdesc(ctx, m, desc, n_mp)) != ASD_OK) return rc;
  if (replay_on_device(m, 1, F->n, n_mp)) {
    if (chained) *chained = chain != nullptr;
    return search_and_resolve<1>(ctx, m, *F, n_mp, desc ? m->d_qdesc : m->d_bank, nullptr, obs_pos, occupied, 0, nn_ratio, match_cur,
                                 n_matches, chain, defer);
  }
  SearchResult R;
  if ((rc = window_search(ctx, m, *F, n_mp, desc ? m->d_qdesc : m->d_bank, &R, 1)) != ASD_OK) return rc;
  int nmatches = 0;
  for (int q = 0; q < n_mp; ++q) {
    if (R.cnt[q] == 0) continue;
    float best = 256, best2 = 256;
    int best_lvl = -1, best_lvl2 = -1, best_idx = -1;
    for (int t = R.off[q]; t < R.off[q] + R.cnt[q]; ++t) {
      const int j = R.idx[t];
      if (occupied[j] || (match_cur[j] >= 0 && (!obs_pos || obs_pos[match_cur[j]]))) continue;
      const float d = R.dist[t];
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
      match_cur[best_idx] = q;
      nmatches += 2;  // the reference increments twice per match (ORBmatcher.cc:116-117)
    }
  }
  *n_matches = nmatches;
  return ASD_OK;
}

extern "C" {

int asd_match_project_frame(asd_ctx* ctx, int32_t slot_cur, int32_t slot_last, const uint8_t* has_mp, const float* Xw,
                            const float* mp_desc, const float* Tcw, const float* K, float th,
                            int32_t check_orientation, int32_t* match_cur, int32_t* n_matches, const uint8_t* mp_obs_positive) {
  if (!mp_desc) return ASD_ERR_INVALID;
  return match_project_frame_impl(ctx, slot_cur, slot_last, has_mp, Xw, mp_desc, nullptr, Tcw, K, th, check_orientation, match_cur, n_matches, mp_obs_positive);
}

int asd_match_project_frame_bank(asd_ctx* ctx, int32_t slot_cur, int32_t slot_last, const uint8_t* has_mp, const float* Xw,
                                 const int32_t* mp_rows, const float* Tcw, const float* K, float th,
                                 int32_t check_orientation, int32_t* match_cur, int32_t* n_matches, const uint8_t* mp_obs_positive) {
  if (!mp_rows) return ASD_ERR_INVALID;
  return match_project_frame_impl(ctx, slot_cur, slot_last, has_mp, Xw, nullptr, mp_rows, Tcw, K, th, check_orientation, match_cur, n_matches, mp_obs_positive);
}

int asd_match_project_points(asd_ctx* ctx, int32_t slot_cur, int32_t n_mp, const uint8_t* in_view, const float* proj,
                             const int32_t* level, const float* view_cos, const float* desc, const uint8_t* occupied,
                             float th, float nn_ratio, int32_t* match_cur, int32_t* n_matches, const uint8_t* mp_obs_positive) {
  if (n_mp > 0 && !desc) return ASD_ERR_INVALID;
  return match_project_points_impl(ctx, slot_cur, n_mp, in_view, proj, level, view_cos, desc, nullptr, occupied, th, nn_ratio, match_cur, n_matches, mp_obs_positive);
}

int asd_match_project_points_bank(asd_ctx* ctx, int32_t slot_cur, int32_t n_mp, const uint8_t* in_view, const float* proj,
                                  const int32_t* level, const float* view_cos, const int32_t* rows, const uint8_t* occupied,
                                  float th, float nn_ratio, int32_t* match_cur, int32_t* n_matches, const uint8_t* mp_obs_positive) {
  if (n_mp > 0 && !rows) return ASD_ERR_INVALID;
  return match_project_points_impl(ctx, slot_cur, n_mp, in_view, proj, level, view_cos, nullptr, rows, occupied, th, nn_ratio, match_cur, n_matches, mp_obs_positive);
}

// ---- fused tracking chains -------------------------------------------------------------------------------------------
// The numeric bodies of Tracking::TrackWithMotionModel (Tracking.cc:664-723: SearchByProjection(cur, last) then
// Optimizer::PoseOptimization) and Tracking::TrackLocalMap (:725-736 with SearchLocalPoints' matcher call, :803-851) as ONE
// submission each: search, claim replay, edge assembly and the pose solver are enqueued back to back on the context's stream
// and the host synchronises once, instead of search -> host -> solver with two round trips.  Same kernels, same edge order
// (keypoint order, Optimizer.cc:281) as the separate calls, so the results are the same bits (tests/test_track_chain.py).
namespace {
// after the chain: outlier flags per KEYPOINT from the per-edge bytes, or the whole PoseOptimization through the separate
// entry point when the matches were replayed on the host (ASD_MATCH_REPLAY=host, very large inputs)
int finish_pose_chain(asd_ctx* ctx, const AsdFrameSlot& C, const std::function<const float*(int)>& point_of, bool chained, const double* h_io,
                      const double* Kd, double* pose7, uint8_t* outlier, int32_t* n_inliers, bool kp_flags) {
  const int n_cur = C.n;
  if (chained && kp_flags) {
    // the gather form of k_pose_opt hands the flags over per keypoint with the edge count behind them: no walk over the keypoints
    const int ne = (int)(h_io[8 + (n_cur + 7) / 8] + 0.5);
    memcpy(outlier, h_io + 8, (size_t)n_cur);
    *n_inliers = 0;
    if (ne < 3) return ASD_OK;   // Optimizer.cc:323-324 (the kernel left the pose as it was and cleared the flags)
    memcpy(pose7, h_io, 56);
    *n_inliers = ne - (int)(h_io[7] + 0.5);
    return ASD_OK;
  }
  std::vector<int> kp_of_edge;
  for (int j = 0; j < n_cur; ++j) { outlier[j] = 0; if (point_of(j)) kp_of_edge.push_back(j); }
  const int ne = (int)kp_of_edge.size();
  *n_inliers = 0;
  if (ne < 3) return ASD_OK;   // Optimizer.cc:323-324
  if (chained) {
    memcpy(pose7, h_io, 56);
    const uint8_t* flags = reinterpret_cast<const uint8_t*>(h_io + 8);
    for (int e = 0; e < ne; ++e) outlier[kp_of_edge[e]] = flags[e];
    *n_inliers = ne - (int)(h_io[7] + 0.5);
    return ASD_OK;
  }
  std::vector<double> Xd((size_t)3 * ne), obs((size_t)2 * ne), info(ne);
  std::vector<uint8_t> out(ne);
  for (int e = 0; e < ne; ++e) {
    const int jk = kp_of_edge[e];
    const float* X = point_of(jk);
    for (int k = 0; k < 3; ++k) Xd[3 * e + k] = (double)X[k];
    obs[2 * e] = (double)C.kps[jk].x; obs[2 * e + 1] = (double)C.kps[jk].y;
    info[e] = (double)ctx->inv_sigma2[C.kps[jk].octave];
  }
  const int rc = asd_pose_optimize(ctx, pose7, ne, Xd.data(), obs.data(), info.data(), Kd, out.data(), n_inliers);
  if (rc != ASD_OK) return rc;
  for (int e = 0; e < ne; ++e) outlier[kp_of_edge[e]] = out[e];
  return ASD_OK;
}

// The three chains share one shape: build the hook (everything its callbacks need captured BY VALUE: they may run again from
// the completion when the candidate buffers overflowed), enqueue, then either run the completion or hand it to the caller
// (`defer`, asd_track_async).  The completion reads the caller's OUTPUT arrays and, of the inputs, only copies made here.
int track_motion_model_impl(asd_ctx* ctx, int32_t slot_cur, int32_t slot_last, const uint8_t* has_mp, const float* Xw, const float* mp_desc,
                            const int32_t* mp_rows, const float* Tcw, const float* K, float th, int32_t check_orientation,
                            const uint8_t* mp_obs_positive, double* pose7, int32_t* match_cur, int32_t* n_matches, uint8_t* outlier,
                            int32_t* n_inliers, std::function<int()>* defer) {
  AsdFrameSlot *C = slot_of(ctx, slot_cur), *L = slot_of(ctx, slot_last);
  if (!C || !L || !pose7 || !outlier || !n_inliers || !K) return ASD_ERR_INVALID;
  const std::array<double, 4> Kd = {(double)K[0], (double)K[1], (double)K[2], (double)K[3]};
  std::array<double, 7> p0;
  memcpy(p0.data(), pose7, sizeof(double) * 7);
  auto chain = std::make_shared<ChainHook>();
  chain->src[0] = Xw; chain->bytes[0] = (size_t)L->n * 12;
  chain->result_bytes = pose_chain_io_bytes(C->n);
  chain->enqueue = [ctx, C, Kd, p0](const int* d_match, void* const* d_tab, void* d_result, const AsdFusedReplay* fused) {
    return pose_chain_enqueue(ctx, C->n, d_match, C->d_kp, static_cast<const float*>(d_tab[0]), nullptr, nullptr, p0.data(), Kd.data(), static_cast<double*>(d_result),
                              nullptr, nullptr, nullptr, fused);
  };
  if (has_mp && Xw && Tcw && L->n > 0) {   // the projection loop on the device too (2000 points: ~45 us of host time otherwise)
    chain->src[1] = has_mp; chain->bytes[1] = (size_t)L->n;
    if (mp_rows) { chain->src[2] = mp_rows; chain->bytes[2] = (size_t)L->n * 4; }
    ProjectArgs pa{};
    pa.n = L->n; pa.kp_last = L->d_kp;
    memcpy(pa.T, Tcw, sizeof pa.T);
    pa.fx = K[0]; pa.fy = K[1]; pa.cx = K[2]; pa.cy = K[3];
    pa.min_x = C->min_x; pa.max_x = C->max_x; pa.min_y = C->min_y; pa.max_y = C->max_y; pa.th = th;
    for (int l = 0; l < ASD_MAX_LEVELS; ++l) pa.scale[l] = l < ctx->cfg.n_levels ? ctx->scale[l] : 0.f;
    const bool by_rows = mp_rows != nullptr;
    chain->prepare = [ctx, pa, by_rows](WinQuery* d_queries, void* const*, void* const* h_tab, const UploadTail& tail) -> int {
      ProjectArgs a = pa;
      a.Xw = static_cast<const float*>(h_tab[0]); a.has_mp = static_cast<const uint8_t*>(h_tab[1]);
      a.rows = by_rows ? static_cast<const int*>(h_tab[2]) : nullptr;
      a.queries = d_queries;
      a.up = tail;
      a.up.q_blocks = (a.n + 255) / 256;
      hipLaunchKernelGGL(k_project_queries, dim3(a.up.q_blocks + (tail.n16 ? kUploadTailBlocks : 0)), dim3(256), 0, ctx->stream, a);
      ASD_HIP_CHECK(ctx, hipGetLastError());
      return ASD_OK;
    };
  }
  bool chained = false;
  std::function<int()> search_done;
  int rc = match_project_frame_impl(ctx, slot_cur, slot_last, has_mp, Xw, mp_desc, mp_rows, Tcw, K, th, check_orientation, match_cur, n_matches,
                                    mp_obs_positive, chain.get(), &chained, defer ? &search_done : nullptr);
  if (rc != ASD_OK) return rc;
  auto fin = [=]() -> int {
    if (search_done) { const int r = search_done(); if (r != ASD_OK) return r; }
    return finish_pose_chain(ctx, *C, [=](int j) -> const float* { return match_cur[j] >= 0 ? Xw + 3 * (size_t)match_cur[j] : nullptr; }, chained,
                             static_cast<const double*>(chain->h_result), Kd.data(), pose7, outlier, n_inliers, chain->kp_flags);
  };
  if (defer && search_done) { *defer = fin; return ASD_OK; }   // (a search that finished on the host is complete already)
  return fin();
}

int track_local_map_impl(asd_ctx* ctx, int32_t slot_cur, int32_t n_mp, const uint8_t* in_view, const float* proj, const int32_t* level,
                         const float* view_cos, const float* desc, const int32_t* rows, const float* mp_Xw, const uint8_t* occupied,
                         const float* cur_Xw, float th, float nn_ratio, const uint8_t* mp_obs_positive, const float* K, double* pose7,
                         int32_t* match_cur, int32_t* n_matches, uint8_t* outlier, int32_t* n_inliers, std::function<int()>* defer) {
  AsdFrameSlot* F = slot_of(ctx, slot_cur);
  if (!F || !pose7 || !outlier || !n_inliers || !K || (n_mp > 0 && !mp_Xw) || (F->n > 0 && (!occupied || !cur_Xw))) return ASD_ERR_INVALID;
  const std::array<double, 4> Kd = {(double)K[0], (double)K[1], (double)K[2], (double)K[3]};
  std::array<double, 7> p0;
  memcpy(p0.data(), pose7, sizeof(double) * 7);
  auto chain = std::make_shared<ChainHook>();
  chain->src[0] = mp_Xw; chain->bytes[0] = (size_t)n_mp * 12;
  chain->src[1] = cur_Xw; chain->bytes[1] = (size_t)F->n * 12;
  chain->src[2] = occupied; chain->bytes[2] = (size_t)F->n;
  chain->result_bytes = pose_chain_io_bytes(F->n);
  chain->enqueue = [ctx, F, Kd, p0](const int* d_match, void* const* d_tab, void* d_result, const AsdFusedReplay* fused) {
    return pose_chain_enqueue(ctx, F->n, d_match, F->d_kp, static_cast<const float*>(d_tab[0]), static_cast<const uint8_t*>(d_tab[2]),
                              static_cast<const float*>(d_tab[1]), p0.data(), Kd.data(), static_cast<double*>(d_result), nullptr, nullptr, nullptr, fused);
  };
  bool chained = false;
  std::function<int()> search_done;
  int rc = match_project_points_impl(ctx, slot_cur, n_mp, in_view, proj, level, view_cos, desc, rows, occupied, th, nn_ratio, match_cur, n_matches,
                                     mp_obs_positive, chain.get(), &chained, defer ? &search_done : nullptr);
  if (rc != ASD_OK) return rc;
  if (defer && search_done) {
    auto occ = std::make_shared<std::vector<uint8_t>>(occupied, occupied + F->n);   // the completion must not read the caller's inputs
    *defer = [=]() -> int {
      const int r = search_done();
      if (r != ASD_OK) return r;
      const uint8_t* oc = occ->data();
      return finish_pose_chain(ctx, *F, [=](int j) -> const float* { return (oc[j] || match_cur[j] >= 0) ? reinterpret_cast<const float*>(oc) : nullptr; },
                               true, static_cast<const double*>(chain->h_result), Kd.data(), pose7, outlier, n_inliers, chain->kp_flags);
    };
    return ASD_OK;
  }
  return finish_pose_chain(ctx, *F, [&](int j) -> const float* {
    return occupied[j] ? cur_Xw + 3 * (size_t)j : (match_cur[j] >= 0 ? mp_Xw + 3 * (size_t)match_cur[j] : nullptr); }, chained,
    static_cast<const double*>(chain->h_result), Kd.data(), pose7, outlier, n_inliers, chain->kp_flags);
}

// Tracking::SearchLocalPoints (Tracking.cc:803-851: isInFrustum for every local map point, then SearchByProjection) +
// Optimizer::PoseOptimization = Tracking::TrackLocalMap's numeric body (:725-736) with the frustum test and the search windows
// made on the device too: the host uploads the map points (position, normal, distance bounds) and gets matches and pose back.
int track_local_points_impl(asd_ctx* ctx, int32_t slot_cur, int32_t n_mp, const float* Xw, const float* normal, const float* min_dist,
                            const float* max_dist, const float* desc, const int32_t* rows, const float* Tcw, const float* K, float cos_limit,
                            const uint8_t* occupied, const float* cur_Xw, float th, float nn_ratio, const uint8_t* mp_obs_positive,
                            double* pose7, int32_t* match_cur, int32_t* n_matches, uint8_t* outlier, int32_t* n_inliers,
                            std::function<int()>* defer) {
  AsdFrameSlot* F = slot_of(ctx, slot_cur);
  if (!F || n_mp < 0 || !Tcw || !K || !pose7 || !match_cur || !n_matches || !outlier || !n_inliers ||
      (n_mp > 0 && (!Xw || !normal || !min_dist || !max_dist || (!desc && !rows))) || (F->n > 0 && (!occupied || !cur_Xw)))
    return ASD_ERR_INVALID;
  (void)hipSetDevice(ctx->cfg.device);
  MatcherState* m = mstate(ctx);
  if (n_mp == 0 || F->n == 0 || !replay_on_device(m, 1, F->n, n_mp)) {
    // host frustum + the chain on its outputs (also the fallback for very large maps)
    std::vector<uint8_t> in_view(std::max(n_mp, 1));
    std::vector<float> proj((size_t)2 * std::max(n_mp, 1)), vc(std::max(n_mp, 1));
    std::vector<int32_t> level(std::max(n_mp, 1));
    int rc = asd_frustum(ctx, slot_cur, n_mp, Xw, normal, min_dist, max_dist, Tcw, K, cos_limit, in_view.data(), proj.data(), level.data(), vc.data());
    if (rc != ASD_OK) return rc;
    return track_local_map_impl(ctx, slot_cur, n_mp, in_view.data(), proj.data(), level.data(), vc.data(), desc, rows, Xw, occupied, cur_Xw, th,
                                nn_ratio, mp_obs_positive, K, pose7, match_cur, n_matches, outlier, n_inliers, defer);
  }
  std::fill(match_cur, match_cur + F->n, -1);
  *n_matches = 0;
  int rc = ensure_queries(ctx, m, n_mp);
  if (rc != ASD_OK) return rc;
  if (!desc)
    for (int q = 0; q < n_mp; ++q)
      if (rows[q] < 0 || rows[q] >= m->bank_cap) { ctx->set_error("bank row %d out of range", rows[q]); return ASD_ERR_INVALID; }
  if (desc && (rc = upload_qdesc(ctx, m, desc, n_mp)) != ASD_OK) return rc;
  const std::array<double, 4> Kd = {(double)K[0], (double)K[1], (double)K[2], (double)K[3]};
  std::array<double, 7> p0;
  memcpy(p0.data(), pose7, sizeof(double) * 7);
  FrustumArgs fa{};
  fa.n = n_mp; fa.n_levels = ctx->cfg.n_levels; fa.bfactor = th != 1.0;
  memcpy(fa.T, Tcw, sizeof fa.T);
  for (int i = 0; i < 3; ++i) {  // mOw = -mRcw.t()*mtcw (Frame.cc:157): transposed gemm accumulates in double
    double sum = 0;
    for (int k = 0; k < 3; ++k) sum += (double)Tcw[k * 4 + i] * (double)Tcw[k * 4 + 3];
    fa.Ow[i] = (float)(-1.0 * sum);
  }
  fa.fx = K[0]; fa.fy = K[1]; fa.cx = K[2]; fa.cy = K[3];
  fa.min_x = F->min_x; fa.max_x = F->max_x; fa.min_y = F->min_y; fa.max_y = F->max_y;
  fa.cos_limit = cos_limit; fa.th = th;
  for (int l = 0; l < ASD_MAX_LEVELS; ++l) { fa.level_thr[l] = ctx->level_thr[l]; fa.scale[l] = l < ctx->cfg.n_levels ? ctx->scale[l] : 0.f; }
  auto chain = std::make_shared<ChainHook>();
  chain->src[0] = Xw; chain->bytes[0] = (size_t)n_mp * 12;
  chain->src[1] = cur_Xw; chain->bytes[1] = (size_t)F->n * 12;
  chain->src[2] = occupied; chain->bytes[2] = (size_t)F->n;
  chain->src[3] = normal; chain->bytes[3] = (size_t)n_mp * 12;
  chain->src[4] = min_dist; chain->bytes[4] = (size_t)n_mp * 4;
  chain->src[5] = max_dist; chain->bytes[5] = (size_t)n_mp * 4;
  const bool by_rows = desc == nullptr;
  if (by_rows) { chain->src[6] = rows; chain->bytes[6] = (size_t)n_mp * 4; }
  chain->result_bytes = pose_chain_io_bytes(F->n);
  chain->prepare = [ctx, fa, by_rows, n_mp](WinQuery* d_queries, void* const*, void* const* h_tab, const UploadTail& tail) -> int {
    FrustumArgs a = fa;
    a.Xw = static_cast<const float*>(h_tab[0]); a.normal = static_cast<const float*>(h_tab[3]);
    a.min_dist = static_cast<const float*>(h_tab[4]); a.max_dist = static_cast<const float*>(h_tab[5]);
    a.rows = by_rows ? static_cast<const int*>(h_tab[6]) : nullptr;
    a.queries = d_queries;
    a.up = tail;
    a.up.q_blocks = (n_mp + 255) / 256;
    hipLaunchKernelGGL(k_frustum_queries, dim3(a.up.q_blocks + (tail.n16 ? kUploadTailBlocks : 0)), dim3(256), 0, ctx->stream, a);
    ASD_HIP_CHECK(ctx, hipGetLastError());
    return ASD_OK;
  };
  chain->enqueue = [ctx, F, Kd, p0](const int* d_match, void* const* d_tab, void* d_result, const AsdFusedReplay* fused) {
    return pose_chain_enqueue(ctx, F->n, d_match, F->d_kp, static_cast<const float*>(d_tab[0]), static_cast<const uint8_t*>(d_tab[2]),
                              static_cast<const float*>(d_tab[1]), p0.data(), Kd.data(), static_cast<double*>(d_result), nullptr, nullptr, nullptr, fused);
  };
  std::function<int()> search_done;
  rc = search_and_resolve<1>(ctx, m, *F, n_mp, desc ? m->d_qdesc : m->d_bank, nullptr, mp_obs_positive, occupied, 0, nn_ratio, match_cur, n_matches,
                             chain.get(), defer ? &search_done : nullptr);
  if (rc != ASD_OK) return rc;
  auto occ = std::make_shared<std::vector<uint8_t>>(occupied, occupied + F->n);   // the completion must not read the caller's inputs
  auto fin = [=]() -> int {
    if (search_done) { const int r = search_done(); if (r != ASD_OK) return r; }
    const uint8_t* oc = occ->data();
    // chained: finish_pose_chain only asks which keypoints carry an edge (the positions went to the device in the upload block)
    return finish_pose_chain(ctx, *F, [=](int j) -> const float* { return (oc[j] || match_cur[j] >= 0) ? reinterpret_cast<const float*>(oc) : nullptr; },
                             true, static_cast<const double*>(chain->h_result), Kd.data(), pose7, outlier, n_inliers, chain->kp_flags);
  };
  if (defer && search_done) { *defer = fin; return ASD_OK; }
  return fin();
}

// The same with the map points named by ROW of the banks (descriptor bank: MapPoint::mDescriptor, attribute bank: position, normal, distance
// range -- asd_bank_put*, asd_mpbank_put): what the host uploads per candidate shrinks from 36 bytes (position, normal, distances, row) to 4,
// and it gathers nothing.  Same kernels as asd_track_frame's local-map stage (k_frustum_queries' bank form), same results as
// asd_track_local_points_bank given the same attributes.
int track_local_points_rows_impl(asd_ctx* ctx, int32_t slot_cur, int32_t n_mp, const int32_t* rows, const float* Tcw, const float* K, float cos_limit,
                                 const uint8_t* occupied, const float* cur_Xw, float th, float nn_ratio, const uint8_t* mp_obs_positive,
                                 double* pose7, int32_t* match_cur, int32_t* n_matches, uint8_t* outlier, int32_t* n_inliers, std::function<int()>* defer) {
  AsdFrameSlot* F = slot_of(ctx, slot_cur);
  if (!F || n_mp < 1 || !rows || !Tcw || !K || !pose7 || !match_cur || !n_matches || !outlier || !n_inliers || F->n < 1 || !occupied || !cur_Xw) return ASD_ERR_INVALID;
  (void)hipSetDevice(ctx->cfg.device);
  MatcherState* m = mstate(ctx);
  if (!replay_on_device(m, 1, F->n, n_mp)) {
    ctx->set_error("asd_track_local_points_rows: %d map points / %d keypoints are outside the device replay (use asd_track_local_points_bank)", n_mp, F->n);
    return ASD_ERR_CAPACITY;
  }
  for (int q = 0; q < n_mp; ++q)
    if (rows[q] < 0 || rows[q] >= m->bank_cap || rows[q] >= m->attr_cap) { ctx->set_error("row %d out of range (descriptor bank %d rows, attribute bank %d)", rows[q], m->bank_cap, m->attr_cap); return ASD_ERR_INVALID; }
  std::fill(match_cur, match_cur + F->n, -1);
  *n_matches = 0;
  int rc = ensure_queries(ctx, m, n_mp);
  if (rc != ASD_OK) return rc;
  if (m->cxw_cap < (size_t)n_mp * 3) {
    if (m->d_cxw) { ASD_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream)); (void)hipFree(m->d_cxw); m->d_cxw = nullptr; m->cxw_cap = 0; }
    const size_t want = (size_t)n_mp * 3 + (size_t)n_mp / 2 * 3 + 3072;
    ASD_HIP_CHECK(ctx, hipMalloc(&m->d_cxw, want * sizeof(float)));
    m->cxw_cap = want;
  }
  const std::array<double, 4> Kd = {(double)K[0], (double)K[1], (double)K[2], (double)K[3]};
  std::array<double, 7> p0;
  memcpy(p0.data(), pose7, sizeof(double) * 7);
  FrustumArgs fa{};
  fa.n = n_mp; fa.n_levels = ctx->cfg.n_levels; fa.bfactor = th != 1.0;
  memcpy(fa.T, Tcw, sizeof fa.T);
  for (int i = 0; i < 3; ++i) {  // mOw = -mRcw.t()*mtcw (Frame.cc:157): transposed gemm accumulates in double
    double sum = 0;
    for (int k = 0; k < 3; ++k) sum += (double)Tcw[k * 4 + i] * (double)Tcw[k * 4 + 3];
    fa.Ow[i] = (float)(-1.0 * sum);
  }
  fa.fx = K[0]; fa.fy = K[1]; fa.cx = K[2]; fa.cy = K[3];
  fa.min_x = F->min_x; fa.max_x = F->max_x; fa.min_y = F->min_y; fa.max_y = F->max_y;
  fa.cos_limit = cos_limit; fa.th = th;
  for (int l = 0; l < ASD_MAX_LEVELS; ++l) { fa.level_thr[l] = ctx->level_thr[l]; fa.scale[l] = l < ctx->cfg.n_levels ? ctx->scale[l] : 0.f; }
  fa.attr = m->d_attr; fa.T_dev = nullptr; fa.skip = nullptr; fa.xw_out = m->d_cxw;
  auto chain = std::make_shared<ChainHook>();
  chain->src[1] = cur_Xw; chain->bytes[1] = (size_t)F->n * 12;
  chain->src[2] = occupied; chain->bytes[2] = (size_t)F->n;
  chain->src[6] = rows; chain->bytes[6] = (size_t)n_mp * 4;
  chain->result_bytes = pose_chain_io_bytes(F->n);
  chain->prepare = [ctx, fa, n_mp](WinQuery* d_queries, void* const*, void* const* h_tab, const UploadTail& tail) -> int {
    FrustumArgs a = fa;
    a.rows = static_cast<const int*>(h_tab[6]);
    a.queries = d_queries;
    a.up = tail;
    a.up.q_blocks = (n_mp + 255) / 256;
    hipLaunchKernelGGL(k_frustum_queries, dim3(a.up.q_blocks + (tail.n16 ? kUploadTailBlocks : 0)), dim3(256), 0, ctx->stream, a);
    ASD_HIP_CHECK(ctx, hipGetLastError());
    return ASD_OK;
  };
  float* d_cxw = m->d_cxw;
  chain->enqueue = [ctx, F, Kd, p0, d_cxw](const int* d_match, void* const* d_tab, void* d_result, const AsdFusedReplay* fused) {
    return pose_chain_enqueue(ctx, F->n, d_match, F->d_kp, d_cxw, static_cast<const uint8_t*>(d_tab[2]), static_cast<const float*>(d_tab[1]), p0.data(), Kd.data(),
                              static_cast<double*>(d_result), nullptr, nullptr, nullptr, fused);
  };
  std::function<int()> search_done;
  rc = search_and_resolve<1>(ctx, m, *F, n_mp, m->d_bank, nullptr, mp_obs_positive, occupied, 0, nn_ratio, match_cur, n_matches, chain.get(), defer ? &search_done : nullptr);
  if (rc != ASD_OK) return rc;
  auto occ = std::make_shared<std::vector<uint8_t>>(occupied, occupied + F->n);   // the completion must not read the caller's inputs
  auto fin = [=]() -> int {
    if (search_done) { const int r = search_done(); if (r != ASD_OK) return r; }
    const uint8_t* oc = occ->data();
    return finish_pose_chain(ctx, *F, [=](int j) -> const float* { return (oc[j] || match_cur[j] >= 0) ? reinterpret_cast<const float*>(oc) : nullptr; },
                             true, static_cast<const double*>(chain->h_result), Kd.data(), pose7, outlier, n_inliers, chain->kp_flags);
  };
  if (defer && search_done) { *defer = fin; return ASD_OK; }
  return fin();
}

// asd_track_async / asd_track_finish: run an asd_track_* entry point split in two
extern "C++" {
template <typename Impl>
int run_track(asd_ctx* ctx, Impl&& impl) {
  if (!ctx) return ASD_ERR_INVALID;
  if (asd_track_busy(ctx, "asd_track_*")) return ASD_ERR_INVALID;
  const bool async = ctx->track_async_armed;
  ctx->track_async_armed = false;
  std::function<int()> fin;
  const int rc = impl(async ? &fin : nullptr);
  if (!async || rc != ASD_OK) return rc;
  ctx->track_pending = fin ? std::move(fin) : std::function<int()>([] { return (int)ASD_OK; });   // finished on the host already
  ctx->track_has_pending = true;
  return ASD_OK;
}
}  // extern "C++"
}  // namespace

int asd_track_motion_model(asd_ctx* ctx, int32_t slot_cur, int32_t slot_last, const uint8_t* has_mp, const float* Xw, const float* mp_desc,
                           const float* Tcw, const float* K, float th, int32_t check_orientation, const uint8_t* mp_obs_positive, double* pose7,
                           int32_t* match_cur, int32_t* n_matches, uint8_t* outlier, int32_t* n_inliers) {
  if (!mp_desc) return ASD_ERR_INVALID;
  return run_track(ctx, [&](std::function<int()>* defer) {
    return track_motion_model_impl(ctx, slot_cur, slot_last, has_mp, Xw, mp_desc, nullptr, Tcw, K, th, check_orientation, mp_obs_positive, pose7,
                                   match_cur, n_matches, outlier, n_inliers, defer); });
}
int asd_track_motion_model_bank(asd_ctx* ctx, int32_t slot_cur, int32_t slot_last, const uint8_t* has_mp, const float* Xw, const int32_t* mp_rows,
                                const float* Tcw, const float* K, float th, int32_t check_orientation, const uint8_t* mp_obs_positive,
                                double* pose7, int32_t* match_cur, int32_t* n_matches, uint8_t* outlier, int32_t* n_inliers) {
  if (!mp_rows) return ASD_ERR_INVALID;
  return run_track(ctx, [&](std::function<int()>* defer) {
    return track_motion_model_impl(ctx, slot_cur, slot_last, has_mp, Xw, nullptr, mp_rows, Tcw, K, th, check_orientation, mp_obs_positive, pose7,
                                   match_cur, n_matches, outlier, n_inliers, defer); });
}
int asd_track_local_map(asd_ctx* ctx, int32_t slot_cur, int32_t n_mp, const uint8_t* in_view, const float* proj, const int32_t* level,
                        const float* view_cos, const float* desc, const float* mp_Xw, const uint8_t* occupied, const float* cur_Xw, float th,
                        float nn_ratio, const uint8_t* mp_obs_positive, const float* K, double* pose7, int32_t* match_cur, int32_t* n_matches,
                        uint8_t* outlier, int32_t* n_inliers) {
  if (n_mp > 0 && !desc) return ASD_ERR_INVALID;
  return run_track(ctx, [&](std::function<int()>* defer) {
    return track_local_map_impl(ctx, slot_cur, n_mp, in_view, proj, level, view_cos, desc, nullptr, mp_Xw, occupied, cur_Xw, th, nn_ratio,
                                mp_obs_positive, K, pose7, match_cur, n_matches, outlier, n_inliers, defer); });
}
int asd_track_local_map_bank(asd_ctx* ctx, int32_t slot_cur, int32_t n_mp, const uint8_t* in_view, const float* proj, const int32_t* level,
                             const float* view_cos, const int32_t* rows, const float* mp_Xw, const uint8_t* occupied, const float* cur_Xw,
                             float th, float nn_ratio, const uint8_t* mp_obs_positive, const float* K, double* pose7, int32_t* match_cur,
                             int32_t* n_matches, uint8_t* outlier, int32_t* n_inliers) {
  if (n_mp > 0 && !rows) return ASD_ERR_INVALID;
  return run_track(ctx, [&](std::function<int()>* defer) {
    return track_local_map_impl(ctx, slot_cur, n_mp, in_view, proj, level, view_cos, nullptr, rows, mp_Xw, occupied, cur_Xw, th, nn_ratio,
                                mp_obs_positive, K, pose7, match_cur, n_matches, outlier, n_inliers, defer); });
}
int asd_track_local_points(asd_ctx* ctx, int32_t slot_cur, int32_t n_mp, const float* Xw, const float* normal, const float* min_dist,
                           const float* max_dist, const float* desc, const float* Tcw, const float* K, float viewing_cos_limit,
                           const uint8_t* occupied, const float* cur_Xw, float th, float nn_ratio, const uint8_t* mp_obs_positive, double* pose7,
                           int32_t* match_cur, int32_t* n_matches, uint8_t* outlier, int32_t* n_inliers) {
  if (n_mp > 0 && !desc) return ASD_ERR_INVALID;
  return run_track(ctx, [&](std::function<int()>* defer) {
    return track_local_points_impl(ctx, slot_cur, n_mp, Xw, normal, min_dist, max_dist, desc, nullptr, Tcw, K, viewing_cos_limit, occupied, cur_Xw, th,
                                   nn_ratio, mp_obs_positive, pose7, match_cur, n_matches, outlier, n_inliers, defer); });
}
int asd_track_local_points_bank(asd_ctx* ctx, int32_t slot_cur, int32_t n_mp, const float* Xw, const float* normal, const float* min_dist,
                                const float* max_dist, const int32_t* rows, const float* Tcw, const float* K, float viewing_cos_limit,
                                const uint8_t* occupied, const float* cur_Xw, float th, float nn_ratio, const uint8_t* mp_obs_positive,
                                double* pose7, int32_t* match_cur, int32_t* n_matches, uint8_t* outlier, int32_t* n_inliers) {
  if (n_mp > 0 && !rows) return ASD_ERR_INVALID;
  return run_track(ctx, [&](std::function<int()>* defer) {
    return track_local_points_impl(ctx, slot_cur, n_mp, Xw, normal, min_dist, max_dist, nullptr, rows, Tcw, K, viewing_cos_limit, occupied, cur_Xw, th,
                                   nn_ratio, mp_obs_positive, pose7, match_cur, n_matches, outlier, n_inliers, defer); });

}

int asd_track_local_points_rows(asd_ctx* ctx, int32_t slot_cur, int32_t n_mp, const int32_t* rows, const float* Tcw, const float* K, float viewing_cos_limit,
                                const uint8_t* occupied, const float* cur_Xw, float th, float nn_ratio, const uint8_t* mp_obs_positive,
                                double* pose7, int32_t* match_cur, int32_t* n_matches, uint8_t* outlier, int32_t* n_inliers) {
  return run_track(ctx, [&](std::function<int()>* defer) {
    return track_local_points_rows_impl(ctx, slot_cur, n_mp, rows, Tcw, K, viewing_cos_limit, occupied, cur_Xw, th, nn_ratio, mp_obs_positive, pose7, match_cur,
                                        n_matches, outlier, n_inliers, defer); });
}


namespace {
int track_frame_impl(asd_ctx* ctx, const asd_track_frame_args& A, std::function<int()>* defer) {
  AsdFrameSlot *C = slot_of(ctx, A.slot_cur), *L = slot_of(ctx, A.slot_last);
  if (!C || !L || !A.has_mp || !A.Xw_last || !A.last_rows || !A.Tcw || !A.K || !A.cand_rows || A.n_cand < 1 || !A.pose7 || !A.pose1 || !A.match1 ||
      !A.n_matches1 || !A.outlier1 || !A.n_inliers1 || !A.match2 || !A.n_matches2 || !A.outlier2 || !A.n_inliers2)
    return ASD_ERR_INVALID;
  if (asd_track_busy(ctx, "asd_track_frame")) return ASD_ERR_INVALID;
  (void)hipSetDevice(ctx->cfg.device);
  MatcherState* m = mstate(ctx);
  const int nl = L->n, nc = C->n, ncand = A.n_cand;
  if (nl < 1 || nc < 1 || !replay_on_device(m, 0, nc, nl) || !replay_on_device(m, 1, nc, ncand) || !pose_chain_lds_form(ctx, nc)) {
    ctx->set_error("asd_track_frame: %d / %d keypoints, %d candidates are outside what the one-submission form handles (empty frame, more than "
                   "%d last-frame points or %d candidates, or tables beyond the workgroup's LDS): use asd_track_motion_model_bank + asd_track_local_points_bank", nc, nl, ncand,
                   kResolve2Threads * 4, kResolve2MaxRounds * 512);
    return ASD_ERR_CAPACITY;
  }
  for (int i = 0; i < nl; ++i)
    if (A.has_mp[i] && (A.last_rows[i] < 0 || A.last_rows[i] >= m->bank_cap)) { ctx->set_error("bank row %d out of range", A.last_rows[i]); return ASD_ERR_INVALID; }
  for (int c = 0; c < ncand; ++c)
    if (A.cand_rows[c] < 0 || A.cand_rows[c] >= m->bank_cap || A.cand_rows[c] >= m->attr_cap) { ctx->set_error("candidate row %d out of range (descriptor bank %d rows, attribute bank %d)", A.cand_rows[c], m->bank_cap, m->attr_cap); return ASD_ERR_INVALID; }
  // nothing is searched twice here: the candidate buffers take the worst case up front (a list holds at most every keypoint)
  int rc = ensure_cands_dev(ctx, m, (size_t)std::max(nl, ncand) * (size_t)nc);
  if (rc != ASD_OK) return rc;
  hipStream_t st = ctx->stream;
  AsdXfer &up = ctx->up, &down = ctx->down;
  const size_t io_bytes = pose_chain_io_bytes(nc);
  ASD_HIP_CHECK(ctx, up.begin(st, (size_t)nl * sizeof(WinQuery) + (size_t)nl * (1 + 1 + 12 + 4 + 4) + (size_t)ncand * 5 + 16 * 256));
  ASD_HIP_CHECK(ctx, down.begin(st, 2 * (((size_t)nc + 16) * sizeof(int) + 256 + io_bytes + 256)));
  size_t need = 0;
  auto pad = [](size_t b) { return AsdDevBuf::padded(b); };
  need += pad((size_t)ncand * sizeof(WinQuery)) + 2 * pad((size_t)nl * 4) + 2 * pad((size_t)ncand * 4) + pad((size_t)nl * kTop * 2) + pad((size_t)ncand * kTop * 2) +
          pad(nc) + pad((size_t)nc * 12) + pad(ncand) + pad((size_t)ncand * 12) + pad(32 * 4) + pad(io_bytes);
  ASD_HIP_CHECK(ctx, ctx->scratch.reserve(need));
  WinQuery* d_q2 = ctx->scratch.carve<WinQuery>(ncand);
  int *d_off1 = ctx->scratch.carve<int>(nl), *d_cnt1 = ctx->scratch.carve<int>(nl), *d_off2 = ctx->scratch.carve<int>(ncand), *d_cnt2 = ctx->scratch.carve<int>(ncand);
  uint16_t *d_top1 = ctx->scratch.carve<uint16_t>((size_t)nl * kTop), *d_top2 = ctx->scratch.carve<uint16_t>((size_t)ncand * kTop);
  uint8_t* d_occ = ctx->scratch.carve<uint8_t>(nc);
  float* d_curXw = ctx->scratch.carve<float>((size_t)nc * 3);
  uint8_t* d_skip = ctx->scratch.carve<uint8_t>(ncand);
  float* d_cXw = ctx->scratch.carve<float>((size_t)ncand * 3);
  float* d_T1 = ctx->scratch.carve<float>(32);
  double* d_io1 = reinterpret_cast<double*>(ctx->scratch.carve<char>(io_bytes));
  // one upload block: the motion-model stage's query table (written on the device) first, the copy the kernel carries behind it
  const size_t o_q1 = up.reserve((size_t)nl * sizeof(WinQuery));
  const size_t o_tot1 = up.zeros(sizeof(int)), o_tot2 = up.zeros(sizeof(int));
  const size_t o_obs1 = A.last_obs_positive ? up.add(A.last_obs_positive, nl) : 0, o_obs2 = A.cand_obs_positive ? up.add(A.cand_obs_positive, ncand) : 0;
  const size_t o_has = up.add(A.has_mp, nl), o_Xw = up.add(A.Xw_last, (size_t)nl * 12), o_rows1 = up.add(A.last_rows, (size_t)nl * 4);
  const size_t o_lc = A.last_cand ? up.add(A.last_cand, (size_t)nl * 4) : 0, o_crows = up.add(A.cand_rows, (size_t)ncand * 4);
  const size_t o_btw = up.reserve(sizeof(AsdBetweenArgs));   // (filled below, before the launch whose tail blocks copy the block to the device)
  const size_t o_out1 = down.reserve(((size_t)nc + 16) * sizeof(int)), o_res1 = down.reserve(io_bytes);
  const size_t o_out2 = down.reserve(((size_t)nc + 16) * sizeof(int)), o_res2 = down.reserve(io_bytes);
  const bool has_obs1 = A.last_obs_positive != nullptr, has_obs2 = A.cand_obs_positive != nullptr, has_lc = A.last_cand != nullptr;
  const std::array<double, 4> Kd = {(double)A.K[0], (double)A.K[1], (double)A.K[2], (double)A.K[3]};

  {
    AsdBetweenArgs b{};
    b.n_cur = nc; b.n_last = nl; b.n_cand = ncand;
    b.match1 = down.dev<int>(o_out1); b.io1 = d_io1; b.Xw_last = up.dev<float>(o_Xw); b.last_cand = has_lc ? up.dev<int>(o_lc) : nullptr;
    memcpy(b.T_pred, A.Tcw, sizeof b.T_pred);
    b.occ = d_occ; b.cur_Xw = d_curXw; b.skip = d_skip; b.T1 = d_T1;
    memcpy(up.host<char>(o_btw), &b, sizeof b);
  }
  {
    // every runtime request of the chain that is not a launch happens here, before the first kernel goes out
    if ((rc = pose_chain_reserve(ctx, nc)) != ASD_OK) return rc;
    static AsdPerDeviceOnce attrs;
    if (attrs.need(ctx->cfg.device)) {
      const void* ks[] = {reinterpret_cast<const void*>(k_resolve2<0, 2>), reinterpret_cast<const void*>(k_resolve2<0, 4>),
                          reinterpret_cast<const void*>(k_resolve2<1, 2>), reinterpret_cast<const void*>(k_resolve2<1, 4>)};
      for (const void* k : ks) ASD_HIP_CHECK(ctx, hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
      attrs.done(ctx->cfg.device);
    }
  }
  // ---- motion-model stage: projection arguments (launched below)
  ProjectArgs pa{};
  {
    pa.n = nl; pa.kp_last = L->d_kp;
    memcpy(pa.T, A.Tcw, sizeof pa.T);
    pa.fx = A.K[0]; pa.fy = A.K[1]; pa.cx = A.K[2]; pa.cy = A.K[3];
    pa.min_x = C->min_x; pa.max_x = C->max_x; pa.min_y = C->min_y; pa.max_y = C->max_y; pa.th = A.th;
    for (int l = 0; l < ASD_MAX_LEVELS; ++l) pa.scale[l] = l < ctx->cfg.n_levels ? ctx->scale[l] : 0.f;
    pa.Xw = up.host<float>(o_Xw); pa.has_mp = up.host<uint8_t>(o_has); pa.rows = up.host<int>(o_rows1);   // the query blocks read the pinned block
    pa.queries = up.dev<WinQuery>(o_q1);
    pa.up = UploadTail{reinterpret_cast<const uint4*>(up.h), reinterpret_cast<uint4*>(up.d), ((size_t)nl * sizeof(WinQuery) + 255) / 256 * 16, (up.used + 15) / 16,
                       (nl + 255) / 256};
  }
  auto resolve_launch = [&](auto kern, const Resolve2Args& a, size_t lds) -> hipError_t {   // (attributes: set above)
    hipLaunchKernelGGL(kern, dim3(1), dim3(kResolve2Threads), lds, st, a);
    return hipGetLastError();
  };
  // the replay's arguments of a stage (every pointer is known before anything is launched)
  auto replay_args = [&](int KIND, int nq, int* d_off, int* d_cnt, int* d_total, uint16_t* d_top, const uint8_t* d_obs, const float4* kp_last, int check_ori,
                         float nn_ratio, int* d_out, int* h_out, size_t* lds_out) {
    Resolve2Args a{};
    a.nq = nq; a.n_cur = nc;
    a.q_off = d_off; a.q_cnt = d_cnt; a.idx = m->d_idx; a.dist = m->d_dist; a.top_idx = d_top;
    a.total = d_total; a.cap = m->cand_cap;
    a.obs_pos = d_obs;
    a.kp_cur = C->d_kp; a.kp_last = kp_last;
    a.check_ori = check_ori; a.nn_ratio = nn_ratio;
    a.match_cur = d_out; a.n_matches = d_out + nc; a.mirror = h_out;
    const size_t fixed = resolve_lds_bytes(KIND, nc, nq), per = KIND == 1 ? 6 : 2;
    a.stage_cap = (int)std::min<size_t>(((size_t)m->last_total[KIND] * 5 / 4 + 1023) / 1024 * 1024, ((size_t)96 * 1024 - fixed) / per / 8 * 8);
    *lds_out = fixed + (size_t)a.stage_cap * per;
    return a;
  };
  auto search = [&](int KIND, int nq, const WinQuery* d_q, int* d_off, int* d_cnt, int* d_total, uint16_t* d_top, const uint8_t* d_occ_in) -> int {
    GridDev G{C->d_kp, C->d_cell_start, C->d_cell_items, C->min_x, C->min_y, C->inv_w, C->inv_h};
    SortArgs sa{d_occ_in, KIND == 0 ? TH_HIGH : __builtin_huge_valf(), d_top};
    hipLaunchKernelGGL(k_window_search<true>, dim3((nq + kSearchWaves - 1) / kSearchWaves), dim3(64 * kSearchWaves), 0, st, G, d_q, nq, m->d_bank, C->d_desc, d_off, d_cnt,
                       d_total, m->cand_cap, m->d_idx, m->d_dist, (unsigned*)nullptr, sa);
    ASD_HIP_CHECK(ctx, hipGetLastError());
    return ASD_OK;
  };
  size_t lds1 = 0, lds2 = 0;
  Resolve2Args ra1 = replay_args(0, nl, d_off1, d_cnt1, up.dev<int>(o_tot1), d_top1, has_obs1 ? up.dev<uint8_t>(o_obs1) : nullptr, L->d_kp, A.check_orientation, 0.f,
                                 down.dev<int>(o_out1), down.host<int>(o_out1), &lds1);
  Resolve2Args ra2 = replay_args(1, ncand, d_off2, d_cnt2, up.dev<int>(o_tot2), d_top2, has_obs2 ? up.dev<uint8_t>(o_obs2) : nullptr, nullptr, 0, A.nn_ratio,
                                 down.dev<int>(o_out2), down.host<int>(o_out2), &lds2);
  // replay + solver of a stage as ONE workgroup where the stage's tables fit (k_resolve_pose); two kernels otherwise
  const bool fuse_now = pose_chain_fused_ok(ctx, 0, nl, nc, lds1) && pose_chain_fused_ok(ctx, 1, ncand, nc, lds2);
  AsdFusedReplay fr1{&ra1, 0, nl, lds1}, fr2{&ra2, 1, ncand, lds2};
  FrustumArgs fa{};
  fa.n = ncand; fa.n_levels = ctx->cfg.n_levels; fa.bfactor = A.th_local != 1.0;
  fa.fx = A.K[0]; fa.fy = A.K[1]; fa.cx = A.K[2]; fa.cy = A.K[3];
  fa.min_x = C->min_x; fa.max_x = C->max_x; fa.min_y = C->min_y; fa.max_y = C->max_y;
  fa.cos_limit = A.viewing_cos_limit; fa.th = A.th_local;
  for (int l = 0; l < ASD_MAX_LEVELS; ++l) { fa.level_thr[l] = ctx->level_thr[l]; fa.scale[l] = l < ctx->cfg.n_levels ? ctx->scale[l] : 0.f; }
  fa.rows = up.dev<int>(o_crows); fa.queries = d_q2;
  fa.up = UploadTail{nullptr, nullptr, 0, 0, (ncand + 255) / 256};
  fa.T_dev = d_T1; fa.attr = m->d_attr; fa.skip = d_skip; fa.xw_out = d_cXw;
  auto project = [&]() -> int {
    hipLaunchKernelGGL(k_project_queries, dim3(pa.up.q_blocks + kUploadTailBlocks), dim3(256), 0, st, pa);
    ASD_HIP_CHECK(ctx, hipGetLastError());
    return ASD_OK;
  };
  auto frustum = [&]() -> int {
    hipLaunchKernelGGL(k_frustum_queries, dim3((ncand + 255) / 256), dim3(256), 0, st, fa);
    ASD_HIP_CHECK(ctx, hipGetLastError());
    return ASD_OK;
  };
  {
    if ((rc = project()) != ASD_OK || (rc = search(0, nl, up.dev<WinQuery>(o_q1), d_off1, d_cnt1, up.dev<int>(o_tot1), d_top1, nullptr)) != ASD_OK) return rc;
    if (!fuse_now) {
      if (nl <= 2 * kResolve2Threads) ASD_HIP_CHECK(ctx, resolve_launch(k_resolve2<0, 2>, ra1, lds1));
      else ASD_HIP_CHECK(ctx, resolve_launch(k_resolve2<0, 4>, ra1, lds1));
    }
    // ... what happens between the stages is the tail of the stage's PoseOptimization kernel (the workgroup that has just written the
    // flags and the pose: no launch, no second read of them)
    if ((rc = pose_chain_enqueue(ctx, nc, down.dev<int>(o_out1), C->d_kp, up.dev<float>(o_Xw), nullptr, nullptr, A.pose7, Kd.data(), down.host<double>(o_res1), nullptr,
                                 d_io1, up.dev<AsdBetweenArgs>(o_btw), fuse_now ? &fr1 : nullptr)) != ASD_OK)
      return rc;
    if ((rc = frustum()) != ASD_OK || (rc = search(1, ncand, d_q2, d_off2, d_cnt2, up.dev<int>(o_tot2), d_top2, d_occ)) != ASD_OK) return rc;
    if (!fuse_now) {
      if (ncand <= 2 * kResolve2Threads) ASD_HIP_CHECK(ctx, resolve_launch(k_resolve2<1, 2>, ra2, lds2));
      else ASD_HIP_CHECK(ctx, resolve_launch(k_resolve2<1, 4>, ra2, lds2));
    }
    if ((rc = pose_chain_enqueue(ctx, nc, down.dev<int>(o_out2), C->d_kp, d_cXw, d_occ, d_curXw, nullptr, Kd.data(), down.host<double>(o_res2), d_io1, nullptr, nullptr,
                                 fuse_now ? &fr2 : nullptr)) != ASD_OK)
      return rc;
  }
  if (!ctx->ev_chain) ASD_HIP_CHECK(ctx, hipEventCreateWithFlags(&ctx->ev_chain, hipEventDisableTiming));
  ASD_HIP_CHECK(ctx, hipEventRecord(ctx->ev_chain, st));

  asd_track_frame_args O = A;   // (only the output pointers are used below)
  std::array<double, 7> p_in;
  memcpy(p_in.data(), A.pose7, 56);
  auto complete = [=]() -> int {
    ASD_HIP_CHECK(ctx, hipEventSynchronize(ctx->ev_chain));
    const int *h1 = ctx->down.host<int>(o_out1), *h2 = ctx->down.host<int>(o_out2);
    m->last_total[0] = h1[nc + 1]; m->last_total[1] = h2[nc + 1];
    if (h1[nc + 1] > m->cand_cap || h2[nc + 1] > m->cand_cap) {   // cannot happen with the worst-case sizing above
      ctx->set_error("asd_track_frame: %d / %d candidates overflowed the %d-entry buffers", h1[nc + 1], h2[nc + 1], m->cand_cap);
      return ASD_ERR_CAPACITY;
    }
    auto unpack = [&](const int* h_out, const double* h_io, const double* p_start, int32_t* match, int32_t* n_matches, uint8_t* outlier, double* pose, int32_t* n_inl) {
      memcpy(match, h_out, (size_t)nc * sizeof(int));
      *n_matches = h_out[nc];
      const int ne = (int)(h_io[8 + (nc + 7) / 8] + 0.5);
      memcpy(outlier, h_io + 8, (size_t)nc);
      memcpy(pose, ne < 3 ? p_start : h_io, 56);   // Optimizer.cc:323-324: fewer than 3 correspondences leave the pose alone
      *n_inl = ne < 3 ? 0 : ne - (int)(h_io[7] + 0.5);
    };
    unpack(h1, ctx->down.host<double>(o_res1), p_in.data(), O.match1, O.n_matches1, O.outlier1, O.pose1, O.n_inliers1);
    double p1[7];
    memcpy(p1, O.pose1, 56);
    unpack(h2, ctx->down.host<double>(o_res2), p1, O.match2, O.n_matches2, O.outlier2, O.pose7, O.n_inliers2);
    ctx->ms_match = 0.f;
    static const bool timing = getenv("ASD_TIMING") != nullptr;
    if (timing) {   // k_resolve2's own stamps (10 ns units), averaged over 200 frames
      static double acc[2][5], tl[8]; static long calls;
      const int* hh[2] = {h1, h2};
      for (int k = 0; k < 2; ++k) { acc[k][0] += hh[k][nc + 2]; for (int i = 0; i < 4; ++i) acc[k][1 + i] += 0.01 * hh[k][nc + 3 + i]; }
      // the chain on the device's own 100 MHz clock: resolve 1 [start, end], pose 1 [start, end], resolve 2, pose 2
      const double* io[2] = {ctx->down.host<double>(o_res1), ctx->down.host<double>(o_res2)};
      const int nw = (nc + 7) / 8;
      double st[8];
      for (int k = 0; k < 2; ++k) {
        st[4 * k] = hh[k][nc + 9]; st[4 * k + 1] = hh[k][nc + 10];
        st[4 * k + 2] = (double)((long long)io[k][8 + nw + 1] & 0x7fffffff); st[4 * k + 3] = (double)((long long)io[k][8 + nw + 2] & 0x7fffffff);
      }
      auto d = [](double a, double b) { double x = b - a; if (x < 0) x += 2147483648.0; return 0.01 * x; };
      tl[0] += d(st[0], st[1]); tl[1] += d(st[1], st[2]); tl[2] += d(st[2], st[3]); tl[3] += d(st[3], st[4]);
      tl[4] += d(st[4], st[5]); tl[5] += d(st[5], st[6]); tl[6] += d(st[6], st[7]); tl[7] += d(st[0], st[7]);
      if (++calls % 200 == 0) {
        for (int k = 0; k < 2; ++k)
          fprintf(stderr, "[track_frame resolve kind %d] %.1f iterations, %d candidates; staging %.1f us, iterations %.1f us (the first %.1f), outputs %.1f us; last frame: thread 0 work %.1f us, barrier + verdict %.1f us\n", k,
                  acc[k][0] / calls, hh[k][nc + 1], acc[k][1] / calls, acc[k][2] / calls, acc[k][4] / calls, acc[k][3] / calls, 0.01 * hh[k][nc + 7], 0.01 * hh[k][nc + 8]);
        fprintf(stderr, "[track_frame device clock] resolve1 %.1f | -> pose1 %.1f | pose1 %.1f | -> (frustum, search) resolve2 %.1f | resolve2 %.1f | -> pose2 %.1f | pose2 %.1f | resolve1 start -> pose2 end %.1f us\n",
                tl[0] / calls, tl[1] / calls, tl[2] / calls, tl[3] / calls, tl[4] / calls, tl[5] / calls, tl[6] / calls, tl[7] / calls);
      }
    }
    return ASD_OK;
  };
  if (defer) { *defer = complete; return ASD_OK; }
  return complete();
}
}  // namespace

int asd_track_frame(asd_ctx* ctx, const asd_track_frame_args* args) {
  if (!args) return ASD_ERR_INVALID;
  return run_track(ctx, [&](std::function<int()>* defer) { return track_frame_impl(ctx, *args, defer); });
}

int asd_track_async(asd_ctx* ctx) {
  if (!ctx) return ASD_ERR_INVALID;
  if (asd_track_busy(ctx, "asd_track_async")) return ASD_ERR_INVALID;
  ctx->track_async_armed = true;
  return ASD_OK;
}

int asd_track_finish(asd_ctx* ctx) {
  if (!ctx) return ASD_ERR_INVALID;
  if (!ctx->track_has_pending) { ctx->set_error("asd_track_finish: no asd_track_* call is outstanding (asd_track_async arms the next one)"); return ASD_ERR_INVALID; }
  (void)hipSetDevice(ctx->cfg.device);
  std::function<int()> fin = std::move(ctx->track_pending);
  ctx->track_pending = nullptr;
  ctx->track_has_pending = false;
  return fin();
}

// ORBmatcher::Fuse, search half (ORBmatcher.cc:825-936): the Replace / AddObservation side effects
// (:938-956) do not feed back into the search, so every map point is an independent window query.
int asd_fuse_search(asd_ctx* ctx, int32_t slot_kf, int32_t n_mp, const uint8_t* valid, const float* Xw, const float* normal,
                    const float* min_dist, const float* max_dist, const float* desc, const float* Tcw, const float* K,
                    float th, int32_t* best_idx, float* best_dist) {
  AsdFrameSlot* KF = slot_of(ctx, slot_kf);
  if (!KF || n_mp < 0 || !Tcw || !K || (n_mp > 0 && (!valid || !Xw || !normal || !min_dist || !max_dist || !desc || !best_idx || !best_dist)))
    return ASD_ERR_INVALID;
  (void)hipSetDevice(ctx->cfg.device);
  MatcherState* m = mstate(ctx);
  for (int i = 0; i < n_mp; ++i) { best_idx[i] = -1; best_dist[i] = 256.f; }
  if (n_mp == 0 || KF->n == 0) return ASD_OK;
  int rc = ensure_queries(ctx, m, n_mp);
  if (rc != ASD_OK) return rc;
  const float fx = K[0], fy = K[1], cx = K[2], cy = K[3];
  const float log_scale = std::log(ctx->cfg.scale_factor);
  float Ow[3];  // KeyFrame::SetPose: Ow = -Rwc*tcw with Rwc materialised (gemm small-matrix path, f32 sums)
  for (int i = 0; i < 3; ++i) {
    const float t0 = Tcw[0 * 4 + i] * Tcw[3] + Tcw[1 * 4 + i] * Tcw[7] + Tcw[2 * 4 + i] * Tcw[11];
    Ow[i] = (float)((double)t0 * -1.0);
  }
  std::vector<int> pred(n_mp, 0);
  std::vector<float> pu(n_mp), pv(n_mp);
  for (int i = 0; i < n_mp; ++i) {
    WinQuery& Q = m->h_queries[i];
    Q = WinQuery{0.f, 0.f, 0.f, 0, 0, -1};
    if (!valid[i]) continue;
    const float* P = Xw + 3 * i;
    float Pc[3];
    transform(Tcw, P, Pc);
    if (Pc[2] < 0.0f) continue;
    const float invz = 1 / Pc[2];
    const float u = fx * (Pc[0] * invz) + cx, v = fy * (Pc[1] * invz) + cy;
    if (!(u >= KF->min_x && u < KF->max_x && v >= KF->min_y && v < KF->max_y)) continue;  // KeyFrame::IsInImage
    const float maxD = 1.2f * max_dist[i], minD = 0.8f * min_dist[i];
    const float PO[3] = {P[0] - Ow[0], P[1] - Ow[1], P[2] - Ow[2]};
    const double nn = (double)PO[0] * PO[0] + (double)PO[1] * PO[1] + (double)PO[2] * PO[2];
    const float dist3D = (float)std::sqrt(nn);
    if (dist3D < minD || dist3D > maxD) continue;
    const float* Pn = normal + 3 * i;
    const double dot = (double)PO[0] * Pn[0] + (double)PO[1] * Pn[1] + (double)PO[2] * Pn[2];
    if (dot < 0.5 * dist3D) continue;
    int lvl = (int)std::ceil(std::log(max_dist[i] / dist3D) / log_scale);
    if (lvl < 0) lvl = 0;
    else if (lvl >= ctx->cfg.n_levels) lvl = ctx->cfg.n_levels - 1;
    pred[i] = lvl; pu[i] = u; pv[i] = v;
    Q = WinQuery{u, v, th * ctx->scale[lvl], -1, -1, i};  // KeyFrame::GetFeaturesInArea: no level filter
  }
  if ((rc = upload_qdesc(ctx, m, desc, n_mp)) != ASD_OK) return rc;
  SearchResult R;
  if ((rc = window_search(ctx, m, *KF, n_mp, m->d_qdesc, &R, 2)) != ASD_OK) return rc;
  for (int i = 0; i < n_mp; ++i) {
    if (R.cnt[i] == 0) continue;
    float best = 256;
    int bi = -1;
    for (int t = R.off[i]; t < R.off[i] + R.cnt[i]; ++t) {
      const asd_keypoint& kp = KF->kps[R.idx[t]];
      if (kp.octave < pred[i] - 1 || kp.octave > pred[i]) continue;
      const float ex = pu[i] - kp.x, ey = pv[i] - kp.y;
      const float e2 = ex * ex + ey * ey;
      if (e2 * ctx->inv_sigma2[kp.octave] > 5.99) continue;
      if (R.dist[t] < best) { best = R.dist[t]; bi = R.idx[t]; }
    }
    if (best <= TH_LOW) { best_idx[i] = bi; best_dist[i] = best; }
  }
  return ASD_OK;
}

// ---- asd_fuse_search_batch: every (keyframe, map point list) pair of SearchInNeighbors in one launch ------------------------------
// One wave per candidate map point.  The gates of ORBmatcher::Fuse (ORBmatcher.cc:850-895) are evaluated by the wave (uniformly: every
// lane the same f32 / f64 operations asd_fuse_search's host loop performs, -ffp-contract=off; the predicted level from the thresholds the
// host derived from its own logf), the window is walked in KeyFrame::GetFeaturesInArea order (no level filter), a lane takes a
// candidate -- level gate (:905-906), 5.99 chi2 gate (:909-916), DescriptorDistance in the reference's summation order -- and the wave
// keeps the smallest distance, the FIRST such candidate among equals (`dist < bestDist`, :927).
struct FuseCallDev {
  GridDev G;
  const float* desc;         // the keyframe's descriptors
  float T[16], Ow[3], K[4];
  float min_x, max_x, min_y, max_y;
  int first, n;
};
struct FuseBatchArgs {
  const FuseCallDev* calls; const int* call_of;   // [n_total]
  int n_total, n_levels;
  const uint8_t* valid; const float* Xw; const float* normal; const float* min_dist; const float* max_dist; const float* desc;
  const int* desc_rows;      // null: desc is [n_total][128]; else desc = the descriptor bank and desc_rows[i] the map point's row
  float th, level_thr[ASD_MAX_LEVELS], scale[ASD_MAX_LEVELS], inv_sigma2[ASD_MAX_LEVELS];
  int* best_idx; float* best_dist;
};
__global__ __launch_bounds__(256) void k_fuse_batch(FuseBatchArgs a) {
  const int lane = threadIdx.x & 63;
  const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (i >= a.n_total) return;
  int out_idx = -1;
  float out_dist = 256.f;
  do {
    if (!a.valid[i]) break;
    const FuseCallDev& Cc = a.calls[a.call_of[i]];
    const float* P = a.Xw + 3 * (size_t)i;
    float Pc[3];
    for (int r = 0; r < 3; ++r) {
      const float t0 = Cc.T[r * 4 + 0] * P[0] + Cc.T[r * 4 + 1] * P[1] + Cc.T[r * 4 + 2] * P[2];
      Pc[r] = (float)((double)t0 + (double)Cc.T[r * 4 + 3]);
    }
    if (Pc[2] < 0.0f) break;
    const float invz = 1 / Pc[2];
    const float u = Cc.K[0] * (Pc[0] * invz) + Cc.K[2], v = Cc.K[1] * (Pc[1] * invz) + Cc.K[3];
    if (!(u >= Cc.min_x && u < Cc.max_x && v >= Cc.min_y && v < Cc.max_y)) break;  // KeyFrame::IsInImage
    const float maxD = 1.2f * a.max_dist[i], minD = 0.8f * a.min_dist[i];
    const float PO[3] = {P[0] - Cc.Ow[0], P[1] - Cc.Ow[1], P[2] - Cc.Ow[2]};
    const double nn = (double)PO[0] * PO[0] + (double)PO[1] * PO[1] + (double)PO[2] * PO[2];
    const float dist3D = (float)sqrt(nn);
    if (dist3D < minD || dist3D > maxD) break;
    const float* Pn = a.normal + 3 * (size_t)i;
    const double dot = (double)PO[0] * Pn[0] + (double)PO[1] * Pn[1] + (double)PO[2] * Pn[2];
    if (dot < 0.5 * dist3D) break;
    const float ratio = a.max_dist[i] / dist3D;
    int lvl = 0;
    for (int k = 1; k < a.n_levels; ++k) lvl += ratio >= a.level_thr[k];
    const float r = a.th * a.scale[lvl];
    const GridDev& G = Cc.G;
    const int nMinCellX = max(0, (int)floorf((u - G.min_x - r) * G.inv_w));
    const int nMaxCellX = min(ASD_GRID_COLS - 1, (int)ceilf((u - G.min_x + r) * G.inv_w));
    const int nMinCellY = max(0, (int)floorf((v - G.min_y - r) * G.inv_h));
    const int nMaxCellY = min(ASD_GRID_ROWS - 1, (int)ceilf((v - G.min_y + r) * G.inv_h));
    if (nMinCellX >= ASD_GRID_COLS || nMaxCellX < 0 || nMinCellY >= ASD_GRID_ROWS || nMaxCellY < 0) break;
    const float4* qa = reinterpret_cast<const float4*>(a.desc + (size_t)(a.desc_rows ? a.desc_rows[i] : i) * 128);
    unsigned long long best = ~0ull;
    int pos = 0;   // position in GetFeaturesInArea order (ties: the first wins)
    for (int ix = nMinCellX; ix <= nMaxCellX; ++ix) {
      const int b = G.cell_start[ix * ASD_GRID_ROWS + nMinCellY], e = G.cell_start[ix * ASD_GRID_ROWS + nMaxCellY + 1];
      for (int base = b; base < e; base += 64) {
        const int it = base + lane;
        bool in = false;
        int idx = 0;
        float4 kp = make_float4(0.f, 0.f, 0.f, 0.f);
        if (it < e) {
          idx = G.cell_items[it];
          kp = G.kp[idx];
          const float dx = kp.x - u, dy = kp.y - v;
          in = fabsf(dx) < r && fabsf(dy) < r;
        }
        const unsigned long long m = __ballot(in);
        unsigned long long key = ~0ull;
        if (in) {
          const int p = pos + __popcll(m & ((1ull << lane) - 1));
          const int oct = __float_as_int(kp.z);
          bool ok = !(oct < lvl - 1 || oct > lvl);
          const float ex = u - kp.x, ey = v - kp.y;
          const float e2 = ex * ex + ey * ey;
          if ((double)(e2 * a.inv_sigma2[oct]) > 5.99) ok = false;
          if (ok) {
            const float4* qb = reinterpret_cast<const float4*>(Cc.desc + (size_t)idx * 128);
            float sqd = 0.f;
#pragma unroll 8
            for (int k = 0; k < 32; ++k) {
              const float4 x = qa[k], y = qb[k];
              float d;
              d = x.x - y.x; sqd = sqd + d * d;
              d = x.y - y.y; sqd = sqd + d * d;
              d = x.z - y.z; sqd = sqd + d * d;
              d = x.w - y.w; sqd = sqd + d * d;
            }
            if (sqd < 256.f) key = ((unsigned long long)__float_as_uint(sqd) << 32) | ((unsigned long long)(unsigned)p << 16) | (unsigned)idx;
          }
        }
        for (int off = 32; off >= 1; off >>= 1) {
          const unsigned long long o = __shfl_xor(key, off);
          key = o < key ? o : key;
        }
        best = key < best ? key : best;
        pos += __popcll(m);
      }
    }
    if (best != ~0ull) {
      const float d = __uint_as_float((unsigned)(best >> 32));
      if (d <= TH_LOW) { out_idx = (int)(best & 0xffffu); out_dist = d; }
    }
  } while (false);
  if (lane == 0) { a.best_idx[i] = out_idx; a.best_dist[i] = out_dist; }
}

int asd_fuse_search_batch(asd_ctx* ctx, int32_t n_calls, const asd_fuse_call* calls, int32_t n_total, const uint8_t* valid, const float* Xw,
                          const float* normal, const float* min_dist, const float* max_dist, const float* desc, const int32_t* desc_rows, float th,
                          int32_t* best_idx, float* best_dist) {
  if (!ctx || n_calls < 0 || n_total < 0 || (n_calls > 0 && !calls) ||
      (n_total > 0 && (!valid || !Xw || !normal || !min_dist || !max_dist || (!desc && !desc_rows) || !best_idx || !best_dist)))
    return ASD_ERR_INVALID;
  if (asd_track_busy(ctx, "asd_fuse_search_batch")) return ASD_ERR_INVALID;
  for (int i = 0; i < n_total; ++i) { best_idx[i] = -1; best_dist[i] = 256.f; }
  if (n_calls == 0 || n_total == 0) return ASD_OK;
  (void)hipSetDevice(ctx->cfg.device);
  std::vector<FuseCallDev> cd(n_calls);
  std::vector<int> call_of(n_total, -1);
  for (int c = 0; c < n_calls; ++c) {
    const asd_fuse_call& A = calls[c];
    AsdFrameSlot* KF = slot_of(ctx, A.slot_kf);
    if (!KF || !KF->d_kp || A.first < 0 || A.n < 0 || A.first + A.n > n_total) { ctx->set_error("asd_fuse_search_batch: call %d: slot %d / rows [%d, %d) of %d", c, A.slot_kf, A.first, A.first + A.n, n_total); return ASD_ERR_INVALID; }
    if (KF->n >= 65536) { ctx->set_error("asd_fuse_search_batch: keyframe of %d keypoints", KF->n); return ASD_ERR_CAPACITY; }
    FuseCallDev& D = cd[c];
    D.G = GridDev{KF->d_kp, KF->d_cell_start, KF->d_cell_items, KF->min_x, KF->min_y, KF->inv_w, KF->inv_h};
    D.desc = KF->d_desc;
    memcpy(D.T, A.Tcw, sizeof D.T);
    for (int i = 0; i < 3; ++i) {  // KeyFrame::SetPose: Ow = -Rwc*tcw with Rwc materialised (gemm small-matrix path, f32 sums) -- as asd_fuse_search
      const float t0 = A.Tcw[0 * 4 + i] * A.Tcw[3] + A.Tcw[1 * 4 + i] * A.Tcw[7] + A.Tcw[2 * 4 + i] * A.Tcw[11];
      D.Ow[i] = (float)((double)t0 * -1.0);
    }
    memcpy(D.K, A.K, sizeof D.K);
    D.min_x = KF->min_x; D.max_x = KF->max_x; D.min_y = KF->min_y; D.max_y = KF->max_y;
    D.first = A.first; D.n = A.n;
    for (int i = A.first; i < A.first + A.n; ++i) {
      if (call_of[i] >= 0) { ctx->set_error("asd_fuse_search_batch: rows of calls %d and %d overlap", call_of[i], c); return ASD_ERR_INVALID; }
      call_of[i] = c;
    }
  }
  std::vector<uint8_t> val(valid, valid + n_total);
  for (int i = 0; i < n_total; ++i) if (call_of[i] < 0) { val[i] = 0; call_of[i] = 0; }   // rows no call covers: nothing to search
  MatcherState* mst = mstate(ctx);
  if (!desc)
    for (int i = 0; i < n_total; ++i)
      if (val[i] && (desc_rows[i] < 0 || desc_rows[i] >= mst->bank_cap)) { ctx->set_error("asd_fuse_search_batch: bank row %d out of range", desc_rows[i]); return ASD_ERR_INVALID; }
  hipStream_t st = ctx->stream;
  AsdXfer &up = ctx->up, &down = ctx->down;
  ASD_HIP_CHECK(ctx, up.begin(st, (size_t)n_total * (1 + 12 + 12 + 4 + 4 + 512 + 4) + (size_t)n_calls * sizeof(FuseCallDev) + 16 * 256));
  ASD_HIP_CHECK(ctx, down.begin(st, (size_t)n_total * 8 + 1024));
  FuseBatchArgs a{};
  a.calls = up.dev<FuseCallDev>(up.add(cd.data(), cd.size() * sizeof(FuseCallDev)));
  a.call_of = up.dev<int>(up.add(call_of.data(), (size_t)n_total * 4));
  a.n_total = n_total; a.n_levels = ctx->cfg.n_levels;
  a.valid = up.dev<uint8_t>(up.add(val.data(), n_total));
  a.Xw = up.dev<float>(up.add(Xw, (size_t)n_total * 12));
  a.normal = up.dev<float>(up.add(normal, (size_t)n_total * 12));
  a.min_dist = up.dev<float>(up.add(min_dist, (size_t)n_total * 4));
  a.max_dist = up.dev<float>(up.add(max_dist, (size_t)n_total * 4));
  if (desc) { a.desc = up.dev<float>(up.add(desc, (size_t)n_total * 512)); a.desc_rows = nullptr; }
  else { a.desc = mst->d_bank; a.desc_rows = up.dev<int>(up.add(desc_rows, (size_t)n_total * 4)); }
  a.th = th;
  for (int l = 0; l < ASD_MAX_LEVELS; ++l) {
    a.level_thr[l] = ctx->level_thr[l];
    a.scale[l] = l < ctx->cfg.n_levels ? ctx->scale[l] : 0.f;
    a.inv_sigma2[l] = l < ctx->cfg.n_levels ? ctx->inv_sigma2[l] : 0.f;
  }
  const size_t o_i = down.reserve((size_t)n_total * 4), o_d = down.reserve((size_t)n_total * 4);
  a.best_idx = down.dev<int>(o_i); a.best_dist = down.dev<float>(o_d);
  ASD_HIP_CHECK(ctx, up.upload(st));
  hipLaunchKernelGGL(k_fuse_batch, dim3((n_total + 3) / 4), dim3(256), 0, st, a);
  ASD_HIP_CHECK(ctx, hipGetLastError());
  ASD_HIP_CHECK(ctx, down.download(st));
  ASD_HIP_CHECK(ctx, hipStreamSynchronize(st));
  memcpy(best_idx, down.host<int>(o_i), (size_t)n_total * 4);
  memcpy(best_dist, down.host<float>(o_d), (size_t)n_total * 4);
  return ASD_OK;
}

// ---- relocalisation / loop-closing variants of the projection searches -----------------------
// Same split as M1 / M2: the host evaluates the per-point gates in the reference's f32 arithmetic, k_window_search
// walks the keyframe grid and evaluates the candidates' distances, the host replays best / claim / histogram logic.
namespace {
// -Rcw.t()*tcw: gemm with the transpose flag = general path, double accumulation (as Frame.cc:157)
void centre_gemm_t(const float* R, const float* t, float* Ow) {
  for (int i = 0; i < 3; ++i) {
    double s = 0;
    for (int k = 0; k < 3; ++k) s += (double)R[k * 3 + i] * (double)t[k];
    Ow[i] = (float)(-1.0 * s);
  }
}
// Scw -> Rcw = sRcw / scw, tcw = Scw.col(3) / scw (ORBmatcher.cc:310-313; Mat / scalar = convertTo with a float scale)
void decompose_sim3(const float* Scw, float* Rcw, float* tcw) {
  double d = 0;
  for (int k = 0; k < 3; ++k) d += (double)Scw[k] * (double)Scw[k];
  const float scw = (float)std::sqrt(d);
  const float inv = (float)(1.0 / (double)scw);
  for (int r = 0; r < 3; ++r) {
    for (int c = 0; c < 3; ++c) Rcw[r * 3 + c] = Scw[r * 4 + c] * inv + 0.0f;
    tcw[r] = Scw[r * 4 + 3] * inv + 0.0f;
  }
}
inline void rot_add(const float* R, const float* t, const float* X, float* out) {
  for (int r = 0; r < 3; ++r) {
    const float t0 = R[r * 3 + 0] * X[0] + R[r * 3 + 1] * X[1] + R[r * 3 + 2] * X[2];
    out[r] = (float)((double)t0 + (double)t[r]);
  }
}
inline float norm3f(const float* p) {
  return (float)std::sqrt((double)p[0] * p[0] + (double)p[1] * p[1] + (double)p[2] * p[2]);
}
inline int predict_scale(const asd_ctx* ctx, float maxd_raw, float dist) {  // MapPoint::PredictScale (MapPoint.cc:438-453)
  int s = (int)std::ceil(std::log(maxd_raw / dist) / std::log(ctx->cfg.scale_factor));
  if (s < 0) s = 0;
  else if (s >= ctx->cfg.n_levels) s = ctx->cfg.n_levels - 1;
  return s;
}
// front half shared by the two Scw searches (:300-366, :963-1010)
bool sim3_project(const asd_ctx* ctx, const AsdFrameSlot& KF, const float* Rcw, const float* tcw, const float* Ow, const float* K,
                  const float* P, const float* Pn, float mind_raw, float maxd_raw, bool double_invz, float* u, float* v, int* level) {
  float Pc[3];
  rot_add(Rcw, tcw, P, Pc);
  if (Pc[2] < 0.0f) return false;
  const float invz = double_invz ? (float)(1.0 / Pc[2]) : 1 / Pc[2];
  *u = K[0] * (Pc[0] * invz) + K[2];
  *v = K[1] * (Pc[1] * invz) + K[3];
  if (!(*u >= KF.min_x && *u < KF.max_x && *v >= KF.min_y && *v < KF.max_y)) return false;
  const float PO[3] = {P[0] - Ow[0], P[1] - Ow[1], P[2] - Ow[2]};
  const float dist = norm3f(PO);
  if (dist < 0.8f * mind_raw || dist > 1.2f * maxd_raw) return false;
  const double dot = (double)PO[0] * Pn[0] + (double)PO[1] * Pn[1] + (double)PO[2] * Pn[2];
  if (dot < 0.5 * dist) return false;
  *level = predict_scale(ctx, maxd_raw, dist);
  return true;
}
}  // namespace

extern "C" {

// ORBmatcher::SearchByProjection(Frame&, KeyFrame*, const set<MapPoint*>&, float th, float ORBdist) (:1455-1582)
int asd_match_project_keyframe(asd_ctx* ctx, int32_t slot_cur, int32_t n_kf, const uint8_t* valid, const float* Xw, const float* min_dist,
                               const float* max_dist, const float* desc, const float* kf_angle, const uint8_t* occupied, const float* Tcw,
                               const float* K, float th, float orb_dist, int32_t check_orientation, int32_t* match_cur,
                               int32_t* n_matches) {
  AsdFrameSlot* C = slot_of(ctx, slot_cur);
  if (!C || n_kf < 0 || !Tcw || !K || !match_cur || !n_matches || (C->n > 0 && !occupied) ||
      (n_kf > 0 && (!valid || !Xw || !min_dist || !max_dist || !desc || !kf_angle)))
    return ASD_ERR_INVALID;
  (void)hipSetDevice(ctx->cfg.device);
  MatcherState* m = mstate(ctx);
  std::fill(match_cur, match_cur + C->n, -1);
  *n_matches = 0;
  if (n_kf == 0 || C->n == 0) return ASD_OK;
  int rc = ensure_queries(ctx, m, n_kf);
  if (rc != ASD_OK) return rc;
  float Rcw[9], tcw[3], Ow[3];
  for (int r = 0; r < 3; ++r) { for (int c = 0; c < 3; ++c) Rcw[r * 3 + c] = Tcw[r * 4 + c]; tcw[r] = Tcw[r * 4 + 3]; }
  centre_gemm_t(Rcw, tcw, Ow);
  for (int i = 0; i < n_kf; ++i) {
    WinQuery& Q = m->h_queries[i];
    Q = WinQuery{0.f, 0.f, 0.f, 0, 0, -1};
    if (!valid[i]) continue;
    const float* P = Xw + 3 * i;
    float Pc[3];
    rot_add(Rcw, tcw, P, Pc);
    const float invzc = 1.0 / Pc[2];
    const float u = K[0] * Pc[0] * invzc + K[2];
    const float v = K[1] * Pc[1] * invzc + K[3];
    if (!(u >= C->min_x && u <= C->max_x && v >= C->min_y && v <= C->max_y)) continue;  // also drops NaN (z == 0)
    const float PO[3] = {P[0] - Ow[0], P[1] - Ow[1], P[2] - Ow[2]};
    const float dist3D = norm3f(PO);
    if (dist3D < 0.8f * min_dist[i] || dist3D > 1.2f * max_dist[i]) continue;
    const int lvl = predict_scale(ctx, max_dist[i], dist3D);
    Q = WinQuery{u, v, th * ctx->scale[lvl], lvl - 1, lvl + 1, i};
  }
  if ((rc = upload_qdesc(ctx, m, desc, n_kf)) != ASD_OK) return rc;
  SearchResult R;
  if ((rc = window_search(ctx, m, *C, n_kf, m->d_qdesc, &R)) != ASD_OK) return rc;
  std::vector<uint8_t> occ(occupied, occupied + C->n);
  std::vector<int> hist[HISTO];
  int nmatches = 0;
  for (int i = 0; i < n_kf; ++i) {
    if (R.cnt[i] == 0) continue;
    float best = 256;
    int best_idx = -1;
    for (int t = R.off[i]; t < R.off[i] + R.cnt[i]; ++t) {
      const int j = R.idx[t];
      if (occ[j]) continue;
      if (R.dist[t] < best) { best = R.dist[t]; best_idx = j; }
    }
    if (best <= orb_dist) {
      occ[best_idx] = 1;
      match_cur[best_idx] = i;
      nmatches++;
      if (check_orientation) hist[rot_bin(kf_angle[i], C->kps[best_idx].angle)].push_back(best_idx);
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

// ORBmatcher::SearchByProjection(KeyFrame*, cv::Mat Scw, const vector<MapPoint*>&, vector<MapPoint*>& vpMatched, int th) (:300-413)
int asd_match_project_sim3(asd_ctx* ctx, int32_t slot_kf, const float* Scw, int32_t n_mp, const uint8_t* valid, const float* Xw,
                           const float* normal, const float* min_dist, const float* max_dist, const float* desc, const float* K,
                           int32_t th, int32_t* matched_kp, int32_t* n_matches) {
  AsdFrameSlot* KF = slot_of(ctx, slot_kf);
  if (!KF || !Scw || n_mp < 0 || !K || !n_matches || (KF->n > 0 && !matched_kp) ||
      (n_mp > 0 && (!valid || !Xw || !normal || !min_dist || !max_dist || !desc)))
    return ASD_ERR_INVALID;
  (void)hipSetDevice(ctx->cfg.device);
  MatcherState* m = mstate(ctx);
  *n_matches = 0;
  if (n_mp == 0 || KF->n == 0) return ASD_OK;
  int rc = ensure_queries(ctx, m, n_mp);
  if (rc != ASD_OK) return rc;
  float Rcw[9], tcw[3], Ow[3];
  decompose_sim3(Scw, Rcw, tcw);
  centre_gemm_t(Rcw, tcw, Ow);
  std::vector<int> pred(n_mp, 0);
  for (int i = 0; i < n_mp; ++i) {
    WinQuery& Q = m->h_queries[i];
    Q = WinQuery{0.f, 0.f, 0.f, 0, 0, -1};
    if (!valid[i]) continue;
    float u, v;
    if (!sim3_project(ctx, *KF, Rcw, tcw, Ow, K, Xw + 3 * i, normal + 3 * i, min_dist[i], max_dist[i], false, &u, &v, &pred[i])) continue;
    Q = WinQuery{u, v, (float)th * ctx->scale[pred[i]], -1, -1, i};
  }
  if ((rc = upload_qdesc(ctx, m, desc, n_mp)) != ASD_OK) return rc;
  SearchResult R;
  if ((rc = window_search(ctx, m, *KF, n_mp, m->d_qdesc, &R)) != ASD_OK) return rc;
  int nmatches = 0;
  for (int i = 0; i < n_mp; ++i) {
    if (R.cnt[i] == 0) continue;
    float best = 256;
    int best_idx = -1;
    for (int t = R.off[i]; t < R.off[i] + R.cnt[i]; ++t) {
      const int j = R.idx[t];
      if (matched_kp[j] != -1) continue;  // vpMatched[idx] already holds a map point
      const int lv = KF->kps[j].octave;
      if (lv < pred[i] - 1 || lv > pred[i]) continue;
      if (R.dist[t] < best) { best = R.dist[t]; best_idx = j; }
    }
    if (best <= TH_LOW) { matched_kp[best_idx] = i; nmatches++; }
  }
  *n_matches = nmatches;
  return ASD_OK;
}

// ORBmatcher::Fuse(KeyFrame*, cv::Mat Scw, const vector<MapPoint*>&, float th, vector<MapPoint*>& vpReplacePoint) (:963-1086)
int asd_fuse_search_sim3(asd_ctx* ctx, int32_t slot_kf, const float* Scw, int32_t n_mp, const uint8_t* valid, const float* Xw,
                         const float* normal, const float* min_dist, const float* max_dist, const float* desc, const float* K, float th,
                         int32_t* best_idx, float* best_dist) {
  AsdFrameSlot* KF = slot_of(ctx, slot_kf);
  if (!KF || !Scw || n_mp < 0 || !K || (n_mp > 0 && (!valid || !Xw || !normal || !min_dist || !max_dist || !desc || !best_idx || !best_dist)))
    return ASD_ERR_INVALID;
  (void)hipSetDevice(ctx->cfg.device);
  MatcherState* m = mstate(ctx);
  for (int i = 0; i < n_mp; ++i) { best_idx[i] = -1; best_dist[i] = 100.f; }
  if (n_mp == 0 || KF->n == 0) return ASD_OK;
  int rc = ensure_queries(ctx, m, n_mp);
  if (rc != ASD_OK) return rc;
  float Rcw[9], tcw[3], Ow[3];
  decompose_sim3(Scw, Rcw, tcw);
  centre_gemm_t(Rcw, tcw, Ow);
  std::vector<int> pred(n_mp, 0);
  for (int i = 0; i < n_mp; ++i) {
    WinQuery& Q = m->h_queries[i];
    Q = WinQuery{0.f, 0.f, 0.f, 0, 0, -1};
    if (!valid[i]) continue;
    float u, v;
    if (!sim3_project(ctx, *KF, Rcw, tcw, Ow, K, Xw + 3 * i, normal + 3 * i, min_dist[i], max_dist[i], true, &u, &v, &pred[i])) continue;
    Q = WinQuery{u, v, th * ctx->scale[pred[i]], -1, -1, i};
  }
  if ((rc = upload_qdesc(ctx, m, desc, n_mp)) != ASD_OK) return rc;
  SearchResult R;
  if ((rc = window_search(ctx, m, *KF, n_mp, m->d_qdesc, &R, 2)) != ASD_OK) return rc;
  for (int i = 0; i < n_mp; ++i) {
    if (R.cnt[i] == 0) continue;
    float best = 100;
    int bi = -1;
    for (int t = R.off[i]; t < R.off[i] + R.cnt[i]; ++t) {
      const int lv = KF->kps[R.idx[t]].octave;
      if (lv < pred[i] - 1 || lv > pred[i]) continue;
      if (R.dist[t] < best) { best = R.dist[t]; bi = R.idx[t]; }
    }
    if (best <= TH_LOW) { best_idx[i] = bi; best_dist[i] = best; }
  }
  return ASD_OK;
}

// ORBmatcher::SearchBySim3 (:1090-1314)
int asd_match_sim3(asd_ctx* ctx, int32_t slot1, int32_t slot2, const uint8_t* has1, const uint8_t* has2, const float* Xw1, const float* Xw2,
                   const float* min_dist1, const float* max_dist1, const float* min_dist2, const float* max_dist2, const float* desc1,
                   const float* desc2, const float* T1w, const float* T2w, float s12, const float* R12, const float* t12, const float* K,
                   float th, int32_t* match12, int32_t* n_matches) {
  AsdFrameSlot *KF1 = slot_of(ctx, slot1), *KF2 = slot_of(ctx, slot2);
  if (!KF1 || !KF2 || !T1w || !T2w || !R12 || !t12 || !K || !n_matches || (KF1->n > 0 && (!has1 || !Xw1 || !min_dist1 || !max_dist1 || !desc1 || !match12)) ||
      (KF2->n > 0 && (!has2 || !Xw2 || !min_dist2 || !max_dist2 || !desc2)))
    return ASD_ERR_INVALID;
  (void)hipSetDevice(ctx->cfg.device);
  MatcherState* m = mstate(ctx);
  *n_matches = 0;
  std::fill(match12, match12 + KF1->n, -1);
  if (KF1->n == 0 || KF2->n == 0) return ASD_OK;
  float R1w[9], t1w[3], R2w[9], t2w[3], sR12[9], sR21[9], t21[3];
  for (int r = 0; r < 3; ++r) {
    for (int c = 0; c < 3; ++c) { R1w[r * 3 + c] = T1w[r * 4 + c]; R2w[r * 3 + c] = T2w[r * 4 + c]; }
    t1w[r] = T1w[r * 4 + 3]; t2w[r] = T2w[r * 4 + 3];
  }
  const float a12 = (float)(double)s12, a21 = (float)(1.0 / (double)s12);  // scaled Mat = convertTo with a float scale
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 3; ++c) { sR12[r * 3 + c] = R12[r * 3 + c] * a12 + 0.0f; sR21[r * 3 + c] = R12[c * 3 + r] * a21 + 0.0f; }
  for (int r = 0; r < 3; ++r) {
    const float t0 = sR21[r * 3 + 0] * t12[0] + sR21[r * 3 + 1] * t12[1] + sR21[r * 3 + 2] * t12[2];
    t21[r] = (float)((double)t0 * -1.0);
  }
  auto one_way = [&](AsdFrameSlot& A, AsdFrameSlot& B, const uint8_t* has, const float* Xw, const float* mind, const float* maxd,
                     const float* desc, const float* Raw, const float* taw, const float* sRba, const float* tba, std::vector<int>& out) -> int {
    int rc = ensure_queries(ctx, m, A.n);
    if (rc != ASD_OK) return rc;
    std::vector<int> pred(A.n, 0);
    for (int i = 0; i < A.n; ++i) {
      WinQuery& Q = m->h_queries[i];
      Q = WinQuery{0.f, 0.f, 0.f, 0, 0, -1};
      if (!has[i]) continue;
      float pA[3], pB[3];
      rot_add(Raw, taw, Xw + 3 * i, pA);
      rot_add(sRba, tba, pA, pB);
      if (pB[2] < 0.0) continue;
      const float invz = 1.0 / pB[2];
      const float u = K[0] * (pB[0] * invz) + K[2], v = K[1] * (pB[1] * invz) + K[3];
      if (!(u >= B.min_x && u < B.max_x && v >= B.min_y && v < B.max_y)) continue;
      const float dist3D = norm3f(pB);
      if (dist3D < 0.8f * mind[i] || dist3D > 1.2f * maxd[i]) continue;
      pred[i] = predict_scale(ctx, maxd[i], dist3D);
      Q = WinQuery{u, v, th * ctx->scale[pred[i]], -1, -1, i};
    }
    if ((rc = upload_qdesc(ctx, m, desc, A.n)) != ASD_OK) return rc;
    SearchResult R;
    if ((rc = window_search(ctx, m, B, A.n, m->d_qdesc, &R)) != ASD_OK) return rc;
    out.assign(A.n, -1);
    for (int i = 0; i < A.n; ++i) {
      if (R.cnt[i] == 0) continue;
      float best = 100;
      int bi = -1;
      for (int t = R.off[i]; t < R.off[i] + R.cnt[i]; ++t) {
        const int lv = B.kps[R.idx[t]].octave;
        if (lv < pred[i] - 1 || lv > pred[i]) continue;
        if (R.dist[t] < best) { best = R.dist[t]; bi = R.idx[t]; }
      }
      if (best <= TH_HIGH) out[i] = bi;
    }
    return ASD_OK;
  };
  std::vector<int> m1, m2;
  int rc = one_way(*KF1, *KF2, has1, Xw1, min_dist1, max_dist1, desc1, R1w, t1w, sR21, t21, m1);
  if (rc != ASD_OK) return rc;
  if ((rc = one_way(*KF2, *KF1, has2, Xw2, min_dist2, max_dist2, desc2, R2w, t2w, sR12, t12, m2)) != ASD_OK) return rc;
  int found = 0;
  for (int i1 = 0; i1 < KF1->n; ++i1) {
    const int i2 = m1[i1];
    if (i2 >= 0 && m2[i2] == i1) { match12[i1] = i2; ++found; }
  }
  *n_matches = found;
  return ASD_OK;
}

}  // extern "C"

// MapPoint::mDescriptor rows kept on the device: written when a map point's descriptor changes
// (MapPoint::ComputeDistinctiveDescriptors, once per keyframe), read by the matchers every frame.
int asd_bank_put(asd_ctx* ctx, int32_t first_row, int32_t n, const float* desc) {
  if (!ctx || first_row < 0 || n < 0 || (n > 0 && !desc)) return ASD_ERR_INVALID;
  (void)hipSetDevice(ctx->cfg.device);
  MatcherState* m = mstate(ctx);
  int rc = ensure_bank(ctx, m, first_row + n);
  if (rc != ASD_OK || n == 0) return rc;
  if ((rc = ensure_qdesc(ctx, m, n)) != ASD_OK) return rc;
  memcpy(m->h_qdesc, desc, (size_t)n * 512);
  ASD_HIP_CHECK(ctx, hipMemcpyAsync(m->d_bank + (size_t)first_row * 128, m->h_qdesc, (size_t)n * 512, hipMemcpyHostToDevice, ctx->stream));
  ASD_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  return ASD_OK;
}

// bank[first_row + i] = descriptor of keypoint i of frame `slot` (device to device), i in [0, n)
int asd_bank_put_from_frame(asd_ctx* ctx, int32_t slot, int32_t first_row, int32_t n) {
  AsdFrameSlot* F = slot_of(ctx, slot);
  if (!F || first_row < 0 || n < 0 || n > F->n) return ASD_ERR_INVALID;
  (void)hipSetDevice(ctx->cfg.device);
  MatcherState* m = mstate(ctx);
  int rc = ensure_bank(ctx, m, first_row + n);
  if (rc != ASD_OK || n == 0) return rc;
  ASD_HIP_CHECK(ctx, copy_rows(asd_prep_stream(ctx), m->d_bank + (size_t)first_row * 128, F->d_desc, (size_t)n * 512));
  return ASD_OK;
}

// MapPoint::{mWorldPos, mNormalVector, mfMinDistance, mfMaxDistance} of rows [first_row, first_row + n) of the bank
int asd_mpbank_put(asd_ctx* ctx, int32_t first_row, int32_t n, const float* Xw, const float* normal, const float* min_dist, const float* max_dist) {
  if (!ctx || first_row < 0 || n < 0 || (n > 0 && (!Xw || !normal || !min_dist || !max_dist))) return ASD_ERR_INVALID;
  (void)hipSetDevice(ctx->cfg.device);
  MatcherState* m = mstate(ctx);
  if (first_row + n > m->attr_cap) {
    const int cap = std::max((first_row + n) * 3 / 2, 16384);
    float* nb = nullptr;
    ASD_HIP_CHECK(ctx, hipMalloc(&nb, (size_t)cap * 32));
    ASD_HIP_CHECK(ctx, hipDeviceSynchronize());   // (growth is rare: everything that reads or writes the old table has finished)
    if (m->d_attr) { ASD_HIP_CHECK(ctx, hipMemcpy(nb, m->d_attr, (size_t)m->attr_cap * 32, hipMemcpyDeviceToDevice)); (void)hipFree(m->d_attr); }
    m->d_attr = nb;
    m->attr_cap = cap;
  }
  if (n == 0) return ASD_OK;
  const int k = m->attr_turn ^= 1;
  if (!m->ev_attr[k]) ASD_HIP_CHECK(ctx, hipEventCreateWithFlags(&m->ev_attr[k], hipEventDisableTiming));
  else ASD_HIP_CHECK(ctx, hipEventSynchronize(m->ev_attr[k]));   // this staging buffer's previous upload has left it
  if (m->h_attr_cap[k] < (size_t)n * 32) {
    if (m->h_attr[k]) (void)hipHostFree(m->h_attr[k]);
    m->h_attr[k] = nullptr; m->h_attr_cap[k] = 0;
    ASD_HIP_CHECK(ctx, hipHostMalloc(&m->h_attr[k], (size_t)n * 48));
    m->h_attr_cap[k] = (size_t)n * 48;
  }
  float* h = m->h_attr[k];
  for (int i = 0; i < n; ++i) {
    float* r = h + 8 * (size_t)i;
    r[0] = Xw[3 * i]; r[1] = Xw[3 * i + 1]; r[2] = Xw[3 * i + 2];
    r[3] = normal[3 * i]; r[4] = normal[3 * i + 1]; r[5] = normal[3 * i + 2];
    r[6] = min_dist[i]; r[7] = max_dist[i];
  }
  hipStream_t st = asd_prep_stream(ctx);
  ASD_HIP_CHECK(ctx, copy_rows(st, m->d_attr + 8 * (size_t)first_row, h, (size_t)n * 32));
  ASD_HIP_CHECK(ctx, hipEventRecord(m->ev_attr[k], st));
  return ASD_OK;
}

// Frame construction beside the stages in flight: between asd_prep_async(ctx, 1) and asd_prep_async(ctx, 0) the device work of
// asd_frame_set, asd_bank_put_from_frame and asd_mpbank_put goes to a second stream of the context instead of queueing behind the
// outstanding asd_track_* stage; closing the bracket makes everything enqueued on the context's stream AFTERWARDS wait for it.
int asd_prep_async(asd_ctx* ctx, int32_t on) {
  if (!ctx) return ASD_ERR_INVALID;
  (void)hipSetDevice(ctx->cfg.device);
  if (on) {
    if (!ctx->stream_prep) {
      int lo = 0, hi = 0;
      (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
      // highest priority, like the main stream
      ASD_HIP_CHECK(ctx, hipStreamCreateWithPriority(&ctx->stream_prep, hipStreamNonBlocking, hi));
      ASD_HIP_CHECK(ctx, hipEventCreateWithFlags(&ctx->ev_prep, hipEventDisableTiming));
    }
    ctx->prep_on = true;
    return ASD_OK;
  }
  if (!ctx->prep_on) return ASD_OK;
  ctx->prep_on = false;
  ASD_HIP_CHECK(ctx, hipEventRecord(ctx->ev_prep, ctx->stream_prep));
  ASD_HIP_CHECK(ctx, hipStreamWaitEvent(ctx->stream, ctx->ev_prep, 0));
  return ASD_OK;
}

}  // extern "C"

// merge-join of two DBoW2 feature vectors (std::map iteration + lower_bound, ORBmatcher.cc:183-278):
// calls fn(a, b) for every node present in both
template <typename Fn>
static void node_join(const asd_feature_vector* A, const asd_feature_vector* B, Fn fn) {
  int a = 0, b = 0;
  while (a < A->n_nodes && b < B->n_nodes) {
    if (A->node_id[a] == B->node_id[b]) { fn(a, b); ++a; ++b; }
    else if (A->node_id[a] < B->node_id[b]) a = (int)(std::lower_bound(A->node_id, A->node_id + A->n_nodes, B->node_id[b]) - A->node_id);
    else b = (int)(std::lower_bound(B->node_id, B->node_id + B->n_nodes, A->node_id[a]) - B->node_id);
  }
}

static bool fv_valid(const asd_feature_vector* fv, int n) {
  if (!fv || fv->n_nodes < 0 || (fv->n_nodes > 0 && (!fv->node_id || !fv->start || !fv->idx))) return false;
  for (int i = 0; i < fv->n_nodes; ++i) {
    if (i > 0 && fv->node_id[i] <= fv->node_id[i - 1]) return false;
    if (fv->start[i + 1] < fv->start[i]) return false;
  }
  for (int t = fv->n_nodes ? fv->start[0] : 0; t < (fv->n_nodes ? fv->start[fv->n_nodes] : 0); ++t)
    if (fv->idx[t] < 0 || fv->idx[t] >= n) return false;
  return true;
}

// distances of every (query keypoint of A, same-node keypoint of B) pair; queries in reference visiting order.
// On return lq[k] describes query k (qrow = keypoint in A, candidates fvB->idx[cbeg..cend), distances at dist[ooff..]).
static int list_search(asd_ctx* ctx, MatcherState* m, const AsdFrameSlot& A, const AsdFrameSlot& B,
                       const asd_feature_vector* fvA, const asd_feature_vector* fvB, const uint8_t* skipA, bool skip_if_set,
                       std::vector<ListQuery>& lq, const float** dist_out) {
  lq.clear();
  int total = 0;
  node_join(fvA, fvB, [&](int a, int b) {
    const int cb = fvB->start[b], ce = fvB->start[b + 1];
    for (int t = fvA->start[a]; t < fvA->start[a + 1]; ++t) {
      const int i = fvA->idx[t];
      if ((skipA[i] != 0) == skip_if_set) continue;
      lq.push_back(ListQuery{i, cb, ce, total});
      total += ce - cb;
    }
  });
  *dist_out = nullptr;
  if (lq.empty() || total == 0) return ASD_OK;
  int rc;
  if ((rc = ensure_cands(ctx, m, std::max(total, fvB->start[fvB->n_nodes]))) != ASD_OK) return rc;
  if ((rc = ensure_queries(ctx, m, (int)lq.size())) != ASD_OK) return rc;
  static_assert(sizeof(ListQuery) <= sizeof(WinQuery), "query staging is shared");
  hipStream_t st = ctx->stream;
  memcpy(m->h_queries, lq.data(), lq.size() * sizeof(ListQuery));
  const int ncand = fvB->start[fvB->n_nodes];
  memcpy(m->h_idx, fvB->idx, (size_t)ncand * sizeof(int));
  ASD_HIP_CHECK(ctx, hipMemcpyAsync(m->d_queries, m->h_queries, lq.size() * sizeof(ListQuery), hipMemcpyHostToDevice, st));
  ASD_HIP_CHECK(ctx, hipMemcpyAsync(m->d_idx, m->h_idx, (size_t)ncand * sizeof(int), hipMemcpyHostToDevice, st));
  ASD_HIP_CHECK(ctx, hipEventRecord(ctx->ev0, st));
  hipLaunchKernelGGL(k_list_dist, dim3(((int)lq.size() + 3) / 4), dim3(256), 0, st, reinterpret_cast<const ListQuery*>(m->d_queries),
                     (int)lq.size(), m->d_idx, A.d_desc, B.d_desc, m->d_dist);
  ASD_HIP_CHECK(ctx, hipGetLastError());
  ASD_HIP_CHECK(ctx, hipEventRecord(ctx->ev1, st));
  ASD_HIP_CHECK(ctx, hipMemcpyAsync(m->h_dist, m->d_dist, (size_t)total * sizeof(float), hipMemcpyDeviceToHost, st));
  ASD_HIP_CHECK(ctx, hipStreamSynchronize(st));
  ASD_HIP_CHECK(ctx, hipEventElapsedTime(&ctx->ms_match, ctx->ev0, ctx->ev1));
  *dist_out = m->h_dist;
  return ASD_OK;
}

extern "C" {

int asd_match_bow(asd_ctx* ctx, int32_t slot_kf, int32_t slot_f, const asd_feature_vector* fv_kf,
                  const asd_feature_vector* fv_f, const uint8_t* has_mp_kf, float nn_ratio, int32_t check_orientation,
                  int32_t* match_f, int32_t* n_matches) {
  AsdFrameSlot *KF = slot_of(ctx, slot_kf), *F = slot_of(ctx, slot_f);
  if (!KF || !F || !has_mp_kf || !match_f || !n_matches || !fv_valid(fv_kf, KF->n) || !fv_valid(fv_f, F->n)) return ASD_ERR_INVALID;
  (void)hipSetDevice(ctx->cfg.device);
  MatcherState* m = mstate(ctx);
  std::fill(match_f, match_f + F->n, -1);
  *n_matches = 0;
  std::vector<ListQuery> lq;
  const float* dist = nullptr;
  int rc = list_search(ctx, m, *KF, *F, fv_kf, fv_f, has_mp_kf, /*skip_if_set=*/false, lq, &dist);
  if (rc != ASD_OK) return rc;
  int nmatches = 0;
  std::vector<int> hist[HISTO];
  for (const ListQuery& Q : lq) {  // reference order: node by node, KF keypoints of the node in order (:194-262)
    float best1 = 256, best2 = 256;
    int best_idx = -1;
    for (int t = Q.cbeg; t < Q.cend; ++t) {
      const int j = fv_f->idx[t];
      if (match_f[j] >= 0) continue;
      const float d = dist[Q.ooff + (t - Q.cbeg)];
      if (d < best1) { best2 = best1; best1 = d; best_idx = j; }
      else if (d < best2) best2 = d;
    }
    if (best1 <= TH_LOW && best1 < nn_ratio * best2) {
      match_f[best_idx] = Q.qrow;
      if (check_orientation) hist[rot_bin(KF->kps[Q.qrow].angle, F->kps[best_idx].angle)].push_back(best_idx);
      nmatches++;
    }
  }
  if (check_orientation) {
    int cnt[HISTO], a, b, c;
    for (int k = 0; k < HISTO; ++k) cnt[k] = (int)hist[k].size();
    three_maxima(cnt, a, b, c);
    for (int k = 0; k < HISTO; ++k) {
      if (k == a || k == b || k == c) continue;
      for (int j : hist[k]) { match_f[j] = -1; nmatches--; }
    }
  }
  *n_matches = nmatches;
  return ASD_OK;
}

int asd_match_triangulate(asd_ctx* ctx, int32_t slot1, int32_t slot2, const asd_feature_vector* fv1,
                          const asd_feature_vector* fv2, const uint8_t* has_mp1, const uint8_t* has_mp2, const float* F12,
                          float ex, float ey, int32_t check_orientation, int32_t* matches12, int32_t* n_matches) {
  AsdFrameSlot *K1 = slot_of(ctx, slot1), *K2 = slot_of(ctx, slot2);
  if (!K1 || !K2 || !has_mp1 || !has_mp2 || !F12 || !matches12 || !n_matches || !fv_valid(fv1, K1->n) || !fv_valid(fv2, K2->n))
    return ASD_ERR_INVALID;
  (void)hipSetDevice(ctx->cfg.device);
  MatcherState* m = mstate(ctx);
  std::fill(matches12, matches12 + K1->n, -1);
  *n_matches = 0;
  std::vector<ListQuery> lq;
  const float* dist = nullptr;
  int rc = list_search(ctx, m, *K1, *K2, fv1, fv2, has_mp1, /*skip_if_set=*/true, lq, &dist);
  if (rc != ASD_OK) return rc;
  int nmatches = 0;
  std::vector<int> hist[HISTO];
  for (const ListQuery& Q : lq) {
    const asd_keypoint& kp1 = K1->kps[Q.qrow];
    // epipolar line of kp1 in image 2 (CheckDistEpipolarLine, ORBmatcher.cc:136-153)
    const float la = kp1.x * F12[0] + kp1.y * F12[3] + F12[6];
    const float lb = kp1.x * F12[1] + kp1.y * F12[4] + F12[7];
    const float lc = kp1.x * F12[2] + kp1.y * F12[5] + F12[8];
    const float den = la * la + lb * lb;
    float best = TH_LOW;
    int best_idx = -1;
    for (int t = Q.cbeg; t < Q.cend; ++t) {
      const int idx2 = fv2->idx[t];
      if (has_mp2[idx2]) continue;  // vbMatched2 is never set by the reference (:688, :729)
      const float d = dist[Q.ooff + (t - Q.cbeg)];
      if (d > TH_LOW || d > best) continue;
      const asd_keypoint& kp2 = K2->kps[idx2];
      const float dex = ex - kp2.x, dey = ey - kp2.y;
      if (dex * dex + dey * dey < 100 * ctx->scale[kp2.octave]) continue;
      const float num = la * kp2.x + lb * kp2.y + lc;
      if (den == 0) continue;
      const float dsqr = num * num / den;
      if (dsqr < 3.84 * ctx->sigma2[kp2.octave]) { best_idx = idx2; best = d; }
    }
    if (best_idx >= 0) {
      matches12[Q.qrow] = best_idx;
      nmatches++;
      if (check_orientation) hist[rot_bin(kp1.angle, K2->kps[best_idx].angle)].push_back(Q.qrow);
    }
  }
  if (check_orientation) {
    int cnt[HISTO], a, b, c;
    for (int k = 0; k < HISTO; ++k) cnt[k] = (int)hist[k].size();
    three_maxima(cnt, a, b, c);
    for (int k = 0; k < HISTO; ++k) {
      if (k == a || k == b || k == c) continue;
      for (int i1 : hist[k]) { matches12[i1] = -1; nmatches--; }
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
  for (int q = 0; q < n; ++q) {
    in_view[q] = 0; proj[2 * q] = proj[2 * q + 1] = 0.f; level[q] = 0; view_cos[q] = 0.f;
    const float* P = Xw + 3 * q;
    float Pc[3];
    transform(Tcw, P, Pc);
    if (Pc[2] < 0.0f) continue;
    const float invz = 1.0f / Pc[2];
    const float u = fx * Pc[0] * invz + cx, v = fy * Pc[1] * invz + cy;
    if (u < F->min_x || u > F->max_x || v < F->min_y || v > F->max_y) continue;
    const float maxD = 1.2f * max_dist[q], minD = 0.8f * min_dist[q];  // MapPoint.cc:409-419
    const float PO[3] = {P[0] - Ow[0], P[1] - Ow[1], P[2] - Ow[2]};
    const double nn = (double)PO[0] * PO[0] + (double)PO[1] * PO[1] + (double)PO[2] * PO[2];
    const float dist = (float)std::sqrt(nn);
    if (dist < minD || dist > maxD) continue;
    const float* Pn = normal + 3 * q;
    const double dot = (double)PO[0] * Pn[0] + (double)PO[1] * Pn[1] + (double)PO[2] * Pn[2];
    const float vc = (float)(dot / dist);
    if (vc < cos_limit) continue;
    const float ratio = max_dist[q] / dist;  // MapPoint::PredictScale (MapPoint.cc:438-453)
    int s = (int)std::ceil(std::log(ratio) / log_scale);
    if (s < 0) s = 0;
    else if (s >= ctx->cfg.n_levels) s = ctx->cfg.n_levels - 1;
    in_view[q] = 1;
    proj[2 * q] = u; proj[2 * q + 1] = v;
    level[q] = s;
    view_cos[q] = vc;
  }
  return ASD_OK;
}

int asd_match_init(asd_ctx* ctx, int32_t slot1, int32_t slot2, float* prev_matched, int32_t window, float nn_ratio,
                   int32_t check_orientation, int32_t* matches12, int32_t* n_matches) {
  AsdFrameSlot *F1 = slot_of(ctx, slot1), *F2 = slot_of(ctx, slot2);
  if (!F1 || !F2 || !prev_matched || !matches12 || !n_matches) return ASD_ERR_INVALID;
  (void)hipSetDevice(ctx->cfg.device);
  MatcherState* m = mstate(ctx);
  std::fill(matches12, matches12 + F1->n, -1);
  *n_matches = 0;
  if (F1->n == 0 || F2->n == 0) return ASD_OK;
  int rc = ensure_queries(ctx, m, F1->n);
  if (rc != ASD_OK) return rc;
  for (int i1 = 0; i1 < F1->n; ++i1) {
    m->h_queries[i1] = WinQuery{0.f, 0.f, 0.f, 0, 0, -1};
    if (F1->kps[i1].octave > 0) continue;  // :434-436
    m->h_queries[i1] = WinQuery{prev_matched[2 * i1], prev_matched[2 * i1 + 1], (float)window, 0, 0, i1};
  }
  SearchResult R;
  if ((rc = window_search(ctx, m, *F2, F1->n, F1->d_desc, &R)) != ASD_OK) return rc;  // query descriptors: frame 1's, resident
  std::vector<float> matched_dist(F2->n, 100.f);
  std::vector<int> matches21(F2->n, -1);
  std::vector<int> hist[HISTO];
  int nmatches = 0;
  for (int i1 = 0; i1 < F1->n; ++i1) {
    if (R.cnt[i1] == 0) continue;
    float best = 100.f, best2 = 100.f;
    int best_idx = -1;
    for (int t = R.off[i1]; t < R.off[i1] + R.cnt[i1]; ++t) {
      const int i2 = R.idx[t];
      const float d = R.dist[t];
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
