// mapping.hip -- new-map-point triangulation for the local-mapping thread.
//
// Replaces the per-match body of LocalMapping::CreateNewMapPoints (reference
// src/vslam/src/LocalMapping.cc:386-519, monocular branch): parallax gate, linear triangulation through the
// 4x4 SVD, cheirality, reprojection chi2 in both keyframes and the scale-consistency gate.  Every match is
// independent, so one lane handles one match; the keyframes' keypoints are already resident in the frame
// slots (asd_frame_set).  The arithmetic follows the reference's OpenCV 3.2.0 evaluation order (f32 data,
// f64 accumulators where cv::Mat::dot / cv::norm / addWeighted / JacobiSVDImpl_ use them), compiled with
// -ffp-contract=off, so accept flags and coordinates are reproducible against the CPU restatement.
#include <cfloat>
#include <cmath>
#include <cstring>

#include "ctx.h"

namespace {

struct TriParams {
  float T1[12], T2[12];      // rows 0..2 of Tcw
  float Rwc1[9], Rwc2[9];    // materialised transposes (KeyFrame::SetPose)
  float Ow1[3], Ow2[3];
  float fx1, fy1, cx1, cy1, ifx1, ify1;
  float fx2, fy2, cx2, cy2, ifx2, ify2;
  float ratio_factor;
  float sf[ASD_MAX_LEVELS], sigma2[ASD_MAX_LEVELS];
};

__device__ __forceinline__ double dot3d(const float* a, const float* b) {
  double r = 0;
  r += (double)a[0] * b[0];
  r += (double)a[1] * b[1];
  r += (double)a[2] * b[2];
  return r;
}
__device__ __forceinline__ float scaled_minus(float alpha, float a, float b) {
  if (alpha == 1.0f) return a - b;  // MatOp_AddEx::assign: plain subtract when the scale is 1
  return (float)((double)a * (double)alpha + (double)b * -1.0 + 0.0);
}

// One-sided Jacobi on the rows of At (= columns of A), rotations accumulated in Vt; returns the row of Vt
// belonging to the smallest singular value.  Registers only: every index is a compile-time constant.
struct Row4 { float v[4]; };

__device__ __forceinline__ void rot_pair(Row4& Ai, Row4& Aj, Row4& Vi, Row4& Vj, double& Wi, double& Wj, bool& changed) {
  const float eps = FLT_EPSILON * 2;
  double a = Wi, p = 0, b = Wj;
#pragma unroll
  for (int k = 0; k < 4; ++k) p += (double)Ai.v[k] * Aj.v[k];
  if (fabs(p) <= eps * sqrt(a * b)) return;
  p *= 2;
  const double beta = a - b, gamma = hypot(p, beta);
  float c, s;
  if (beta < 0) {
    const double delta = (gamma - beta) * 0.5;
    s = (float)sqrt(delta / gamma);
    c = (float)(p / (gamma * s * 2));
  } else {
    c = (float)sqrt((gamma + beta) / (gamma * 2));
    s = (float)(p / (gamma * c * 2));
  }
  a = b = 0;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const float t0 = c * Ai.v[k] + s * Aj.v[k];
    const float t1 = -s * Ai.v[k] + c * Aj.v[k];
    Ai.v[k] = t0; Aj.v[k] = t1;
    a += (double)t0 * t0; b += (double)t1 * t1;
  }
  Wi = a; Wj = b;
  changed = true;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const float t0 = c * Vi.v[k] + s * Vj.v[k];
    const float t1 = -s * Vi.v[k] + c * Vj.v[k];
    Vi.v[k] = t0; Vj.v[k] = t1;
  }
}

__device__ __forceinline__ double rownorm2(const Row4& r) {
  double sd = 0;
#pragma unroll
  for (int k = 0; k < 4; ++k) sd += (double)r.v[k] * r.v[k];
  return sd;
}

__device__ Row4 svd4_last_vt(Row4 A0, Row4 A1, Row4 A2, Row4 A3) {
  Row4 V0{{1, 0, 0, 0}}, V1{{0, 1, 0, 0}}, V2{{0, 0, 1, 0}}, V3{{0, 0, 0, 1}};
  double W0 = rownorm2(A0), W1 = rownorm2(A1), W2 = rownorm2(A2), W3 = rownorm2(A3);
  for (int iter = 0; iter < 30; ++iter) {
    bool changed = false;
    rot_pair(A0, A1, V0, V1, W0, W1, changed);
    rot_pair(A0, A2, V0, V2, W0, W2, changed);
    rot_pair(A0, A3, V0, V3, W0, W3, changed);
    rot_pair(A1, A2, V1, V2, W1, W2, changed);
    rot_pair(A1, A3, V1, V3, W1, W3, changed);
    rot_pair(A2, A3, V2, V3, W2, W3, changed);
    if (!changed) break;
  }
  W0 = sqrt(rownorm2(A0)); W1 = sqrt(rownorm2(A1)); W2 = sqrt(rownorm2(A2)); W3 = sqrt(rownorm2(A3));
  // selection sort, descending, first maximum wins (lapack.cpp JacobiSVDImpl_): only the last slot is needed.
  // Replay it on (W, V) pairs.
#define ASD_SWAP_IF(Wa, Va, Wb, Vb) { const double tw = Wa; Wa = Wb; Wb = tw; const Row4 tv = Va; Va = Vb; Vb = tv; }
  {  // i = 0
    int j = 0; double wj = W0;
    if (wj < W1) { j = 1; wj = W1; }
    if (wj < W2) { j = 2; wj = W2; }
    if (wj < W3) { j = 3; wj = W3; }
    if (j == 1) ASD_SWAP_IF(W0, V0, W1, V1) else if (j == 2) ASD_SWAP_IF(W0, V0, W2, V2) else if (j == 3) ASD_SWAP_IF(W0, V0, W3, V3)
  }
  {  // i = 1
    int j = 1; double wj = W1;
    if (wj < W2) { j = 2; wj = W2; }
    if (wj < W3) { j = 3; wj = W3; }
    if (j == 2) ASD_SWAP_IF(W1, V1, W2, V2) else if (j == 3) ASD_SWAP_IF(W1, V1, W3, V3)
  }
  if (W2 < W3) ASD_SWAP_IF(W2, V2, W3, V3)
#undef ASD_SWAP_IF
  return V3;
}

// the per-match body of LocalMapping::CreateNewMapPoints (LocalMapping.cc:386-519) for one pair of keypoints
__device__ __forceinline__ void triangulate_one(const TriParams& P, const float4 a, const float4 b, float (&X)[3], unsigned char& ok_out) {
  const int o1 = __float_as_int(a.z), o2 = __float_as_int(b.z);
  unsigned char ok = 0;
  X[0] = X[1] = X[2] = 0.f;
  do {
    const float xn1[3] = {(a.x - P.cx1) * P.ifx1, (a.y - P.cy1) * P.ify1, 1.0f};
    const float xn2[3] = {(b.x - P.cx2) * P.ifx2, (b.y - P.cy2) * P.ify2, 1.0f};
    float ray1[3], ray2[3];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      ray1[r] = P.Rwc1[r * 3 + 0] * xn1[0] + P.Rwc1[r * 3 + 1] * xn1[1] + P.Rwc1[r * 3 + 2] * xn1[2];
      ray2[r] = P.Rwc2[r * 3 + 0] * xn2[0] + P.Rwc2[r * 3 + 1] * xn2[1] + P.Rwc2[r * 3 + 2] * xn2[2];
    }
    const float cosr = (float)(dot3d(ray1, ray2) / (sqrt(dot3d(ray1, ray1)) * sqrt(dot3d(ray2, ray2))));
    if (!(cosr < cosr + 1 && cosr > 0 && cosr < 0.9998)) break;
    // rows of A; the SVD runs on the columns (cv::transpose into temp_a)
    float A[4][4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      A[0][k] = scaled_minus(xn1[0], P.T1[8 + k], P.T1[0 + k]);
      A[1][k] = scaled_minus(xn1[1], P.T1[8 + k], P.T1[4 + k]);
      A[2][k] = scaled_minus(xn2[0], P.T2[8 + k], P.T2[0 + k]);
      A[3][k] = scaled_minus(xn2[1], P.T2[8 + k], P.T2[4 + k]);
    }
    const Row4 c0{{A[0][0], A[1][0], A[2][0], A[3][0]}}, c1{{A[0][1], A[1][1], A[2][1], A[3][1]}},
        c2{{A[0][2], A[1][2], A[2][2], A[3][2]}}, c3{{A[0][3], A[1][3], A[2][3], A[3][3]}};
    const Row4 v = svd4_last_vt(c0, c1, c2, c3);
    if (v.v[3] == 0) break;
    const float inv = (float)(1.0 / (double)v.v[3]);
    X[0] = v.v[0] * inv + 0.0f; X[1] = v.v[1] * inv + 0.0f; X[2] = v.v[2] * inv + 0.0f;
    const float z1 = (float)(dot3d(P.T1 + 8, X) + P.T1[11]);
    if (z1 <= 0) break;
    const float z2 = (float)(dot3d(P.T2 + 8, X) + P.T2[11]);
    if (z2 <= 0) break;
    {
      const float x1 = (float)(dot3d(P.T1 + 0, X) + P.T1[3]);
      const float y1 = (float)(dot3d(P.T1 + 4, X) + P.T1[7]);
      const float invz1 = (float)(1.0 / z1);
      const float u1 = P.fx1 * x1 * invz1 + P.cx1, v1 = P.fy1 * y1 * invz1 + P.cy1;
      const float ex = u1 - a.x, ey = v1 - a.y;
      if ((ex * ex + ey * ey) > 5.991 * P.sigma2[o1]) break;
    }
    {
      const float x2 = (float)(dot3d(P.T2 + 0, X) + P.T2[3]);
      const float y2 = (float)(dot3d(P.T2 + 4, X) + P.T2[7]);
      const float invz2 = (float)(1.0 / z2);
      const float u2 = P.fx2 * x2 * invz2 + P.cx2, v2 = P.fy2 * y2 * invz2 + P.cy2;
      const float ex = u2 - b.x, ey = v2 - b.y;
      if ((ex * ex + ey * ey) > 5.991 * P.sigma2[o2]) break;
    }
    const float n1[3] = {X[0] - P.Ow1[0], X[1] - P.Ow1[1], X[2] - P.Ow1[2]};
    const float n2[3] = {X[0] - P.Ow2[0], X[1] - P.Ow2[1], X[2] - P.Ow2[2]};
    const float dist1 = (float)sqrt(dot3d(n1, n1)), dist2 = (float)sqrt(dot3d(n2, n2));
    if (dist1 == 0 || dist2 == 0) break;
    const float ratioDist = dist2 / dist1;
    const float ratioOctave = P.sf[o1] / P.sf[o2];
    if (ratioDist * P.ratio_factor < ratioOctave || ratioDist > ratioOctave * P.ratio_factor) break;
    ok = 1;
  } while (false);
  ok_out = ok;
}

__global__ __launch_bounds__(64) void k_triangulate(TriParams P, const float4* __restrict__ kp1, const float4* __restrict__ kp2,
                                                   const int* __restrict__ idx1, const int* __restrict__ idx2, int n,
                                                   float* __restrict__ x3d_out, unsigned char* __restrict__ ok_out) {
  const int i = blockIdx.x * 64 + threadIdx.x;
  if (i >= n) return;
  float X[3];
  unsigned char ok;
  triangulate_one(P, kp1[idx1[i]], kp2[idx2[i]], X, ok);
  ok_out[i] = ok;
  x3d_out[3 * i + 0] = ok ? X[0] : 0.f;
  x3d_out[3 * i + 1] = ok ? X[1] : 0.f;
  x3d_out[3 * i + 2] = ok ? X[2] : 0.f;
}

// ---- the batched per-keyframe stage (asd_create_map_points_batch): every neighbour of the current keyframe in one launch each ----
struct NbDev {
  const float4* kp; const float* desc; const int* fv; const uint8_t* has;   // fv block: node_id [cap] | start [cap + 1] | idx [cap] | kp_node [cap]
  int n, n_nodes, cap;
  float F12[9], ex, ey;
};
// ORBmatcher::SearchForTriangulation (ORBmatcher.cc:669-822, bOnlyStereo = false, no orientation check) for keypoint i of the current
// keyframe against neighbour b: one wave.  The keypoint's vocabulary node is looked up in the neighbour's FeatureVector (node ids
// ascending: the std::map walk of :694-701 meets equal ids exactly once), its members are the candidates; a lane takes a candidate --
// map point already there: skip (:729), DescriptorDistance in the reference's summation order, d > TH_LOW or d > bestDist: skip (:737),
// too close to the epipole (:745-750), CheckDistEpipolarLine (:136-153) -- and the wave keeps the smallest distance, the LATEST such
// candidate among equals (`d > bestDist` lets an equal distance through, so a later candidate of equal distance replaces an earlier one).
constexpr float kTriThLow = 0.5f;   // TH_LOW
__global__ __launch_bounds__(256) void k_tri_match_batch(const float4* __restrict__ kp_cur, const float* __restrict__ desc_cur, const int* __restrict__ kp_node_cur,
                                                        const uint8_t* __restrict__ has_cur, int n_cur, const NbDev* __restrict__ nbs,
                                                        const float* __restrict__ scale, const float* __restrict__ sigma2, int* __restrict__ matches) {
  const int lane = threadIdx.x & 63;
  const int i = blockIdx.x * 4 + (threadIdx.x >> 6), b = blockIdx.y;
  if (i >= n_cur) return;
  int result = -1;
  const int node = kp_node_cur[i];
  if (!has_cur[i] && node >= 0) {
    const NbDev N = nbs[b];
    const int* node_id = N.fv;
    const int* start = N.fv + N.cap;
    const int* idx = N.fv + 2 * N.cap + 1;
    int pos = -1;
    for (int base = 0; base < N.n_nodes && pos < 0; base += 64) {
      const bool hit = base + lane < N.n_nodes && node_id[base + lane] == node;
      const unsigned long long m = __ballot(hit);
      if (m) pos = base + (__ffsll((long long)m) - 1);
    }
    if (pos >= 0) {
      const int cb = start[pos], ce = start[pos + 1];
      const float4 k1 = kp_cur[i];
      const float la = k1.x * N.F12[0] + k1.y * N.F12[3] + N.F12[6];
      const float lb = k1.x * N.F12[1] + k1.y * N.F12[4] + N.F12[7];
      const float lc = k1.x * N.F12[2] + k1.y * N.F12[5] + N.F12[8];
      const float den = la * la + lb * lb;
      const float4* qa = reinterpret_cast<const float4*>(desc_cur + (size_t)i * 128);
      unsigned long long best = ~0ull;
      for (int t0 = cb; t0 < ce; t0 += 64) {
        const int t = t0 + lane;
        unsigned long long key = ~0ull;
        if (t < ce) {
          const int i2 = idx[t];
          if (!N.has[i2]) {
            const float4* qb = reinterpret_cast<const float4*>(N.desc + (size_t)i2 * 128);
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
            bool ok = !(sqd > kTriThLow);
            const float4 k2 = N.kp[i2];
            const int o2 = __float_as_int(k2.z);
            const float dex = N.ex - k2.x, dey = N.ey - k2.y;
            if (dex * dex + dey * dey < 100 * scale[o2]) ok = false;
            const float num = la * k2.x + lb * k2.y + lc;
            if (den == 0) ok = false;
            const float dsqr = num * num / den;
            if (!((double)dsqr < 3.84 * (double)sigma2[o2])) ok = false;
            // smallest distance, among equals the latest candidate: key = distance bits (non-negative: ordered like the value) | ~t
            if (ok) key = ((unsigned long long)__float_as_uint(sqd) << 32) | (unsigned long long)(0xffffffffu - (unsigned)t);
          }
        }
        for (int off = 32; off >= 1; off >>= 1) {
          const unsigned long long o = __shfl_xor(key, off);
          key = o < key ? o : key;
        }
        best = key < best ? key : best;
      }
      if (best != ~0ull) result = idx[0xffffffffu - (unsigned)(best & 0xffffffffull)];
    }
  }
  if (lane == 0) matches[(size_t)b * n_cur + i] = result;
}
// the triangulation body for every match of every neighbour: one lane per (neighbour, keypoint of the current keyframe)
__global__ __launch_bounds__(64) void k_triangulate_batch(const TriParams* __restrict__ P, const float4* __restrict__ kp_cur, const NbDev* __restrict__ nbs,
                                                         const int* __restrict__ matches, int n_cur, float* __restrict__ x3d, unsigned char* __restrict__ ok_out) {
  const int i = blockIdx.x * 64 + threadIdx.x, b = blockIdx.y;
  if (i >= n_cur) return;
  const size_t o = (size_t)b * n_cur + i;
  const int j = matches[o];
  float X[3] = {0.f, 0.f, 0.f};
  unsigned char ok = 0;
  if (j >= 0) triangulate_one(P[b], kp_cur[i], nbs[b].kp[j], X, ok);
  ok_out[o] = ok;
  x3d[3 * o + 0] = ok ? X[0] : 0.f;
  x3d[3 * o + 1] = ok ? X[1] : 0.f;
  x3d[3 * o + 2] = ok ? X[2] : 0.f;
}

__global__ void k_svd4(const float* __restrict__ A, int n, float* __restrict__ v_out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float* M = A + 16 * i;
  const Row4 c0{{M[0], M[4], M[8], M[12]}}, c1{{M[1], M[5], M[9], M[13]}}, c2{{M[2], M[6], M[10], M[14]}}, c3{{M[3], M[7], M[11], M[15]}};
  const Row4 v = svd4_last_vt(c0, c1, c2, c3);
  for (int k = 0; k < 4; ++k) v_out[4 * i + k] = v.v[k];
}

struct MappingState {
  int cap = 0;
  int *d_idx = nullptr, *h_idx = nullptr;         // [2*cap] idx1 | idx2 (h_* pinned)
  float *d_x = nullptr, *h_x = nullptr;           // [3*cap]
  unsigned char *d_ok = nullptr, *h_ok = nullptr; // [cap]
};

MappingState* mapstate(asd_ctx* ctx) {
  if (!ctx->mapping) ctx->mapping = new MappingState();
  return static_cast<MappingState*>(ctx->mapping);
}

int ensure(asd_ctx* ctx, MappingState* m, int n) {
  if (n <= m->cap) return ASD_OK;
  const int cap = std::max(n * 3 / 2, 4096);
  if (m->d_idx) {
    (void)hipFree(m->d_idx); (void)hipHostFree(m->h_idx); (void)hipFree(m->d_x); (void)hipHostFree(m->h_x);
    (void)hipFree(m->d_ok); (void)hipHostFree(m->h_ok);
  }
  m->cap = 0;
  ASD_HIP_CHECK(ctx, hipMalloc(&m->d_idx, (size_t)2 * cap * sizeof(int)));
  ASD_HIP_CHECK(ctx, hipHostMalloc(&m->h_idx, (size_t)2 * cap * sizeof(int)));
  ASD_HIP_CHECK(ctx, hipMalloc(&m->d_x, (size_t)4 * cap * sizeof(float)));
  ASD_HIP_CHECK(ctx, hipHostMalloc(&m->h_x, (size_t)4 * cap * sizeof(float)));
  ASD_HIP_CHECK(ctx, hipMalloc(&m->d_ok, (size_t)cap));
  ASD_HIP_CHECK(ctx, hipHostMalloc(&m->h_ok, (size_t)cap));
  m->cap = cap;
  return ASD_OK;
}

// KeyFrame::SetPose (KeyFrame.cc:221-228): Rwc = Rcw.t() materialised; Ow = -Rwc*tcw (gemm small-matrix path, alpha = -1)
void camera_centre(const float* T, float* Rwc, float* Ow) {
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) Rwc[i * 3 + j] = T[j * 4 + i];
  for (int i = 0; i < 3; ++i) {
    const float t0 = Rwc[i * 3 + 0] * T[3] + Rwc[i * 3 + 1] * T[7] + Rwc[i * 3 + 2] * T[11];
    Ow[i] = (float)((double)t0 * -1.0);
  }
}

}  // namespace

void mapping_free(asd_ctx* ctx) {
  if (!ctx->mapping) return;
  MappingState* m = static_cast<MappingState*>(ctx->mapping);
  if (m->d_idx) {
    (void)hipFree(m->d_idx); (void)hipHostFree(m->h_idx); (void)hipFree(m->d_x); (void)hipHostFree(m->h_x);
    (void)hipFree(m->d_ok); (void)hipHostFree(m->h_ok);
  }
  delete m;
  ctx->mapping = nullptr;
}

extern "C" {

int asd_triangulate_pairs(asd_ctx* ctx, int32_t slot1, int32_t slot2, int32_t n_pairs, const int32_t* idx1, const int32_t* idx2,
                          const float* Tcw1, const float* Tcw2, const float* K1, const float* K2, float* x3D, uint8_t* ok,
                          int32_t* n_ok) {
  if (!ctx) return ASD_ERR_INVALID;
  if (slot1 < 0 || slot1 >= ASD_MAX_FRAMES || slot2 < 0 || slot2 >= ASD_MAX_FRAMES || n_pairs < 0 || !Tcw1 || !Tcw2 || !K1 || !K2 ||
      (n_pairs > 0 && (!idx1 || !idx2 || !x3D || !ok))) {
    ctx->set_error("asd_triangulate_pairs: invalid argument");
    return ASD_ERR_INVALID;
  }
  const AsdFrameSlot& F1 = ctx->frames[slot1];
  const AsdFrameSlot& F2 = ctx->frames[slot2];
  if (n_ok) *n_ok = 0;
  if (n_pairs == 0) return ASD_OK;
  if (!F1.d_kp || !F2.d_kp) { ctx->set_error("asd_triangulate_pairs: frame slot not set"); return ASD_ERR_INVALID; }
  for (int i = 0; i < n_pairs; ++i)
    if (idx1[i] < 0 || idx1[i] >= F1.n || idx2[i] < 0 || idx2[i] >= F2.n) {
      ctx->set_error("asd_triangulate_pairs: pair %d (%d,%d) out of range (%d,%d keypoints)", i, idx1[i], idx2[i], F1.n, F2.n);
      return ASD_ERR_INVALID;
    }
  (void)hipSetDevice(ctx->cfg.device);
  MappingState* m = mapstate(ctx);
  int rc = ensure(ctx, m, n_pairs);
  if (rc != ASD_OK) return rc;
  TriParams P;
  memcpy(P.T1, Tcw1, sizeof P.T1);
  memcpy(P.T2, Tcw2, sizeof P.T2);
  camera_centre(Tcw1, P.Rwc1, P.Ow1);
  camera_centre(Tcw2, P.Rwc2, P.Ow2);
  P.fx1 = K1[0]; P.fy1 = K1[1]; P.cx1 = K1[2]; P.cy1 = K1[3]; P.ifx1 = 1.0f / K1[0]; P.ify1 = 1.0f / K1[1];
  P.fx2 = K2[0]; P.fy2 = K2[1]; P.cx2 = K2[2]; P.cy2 = K2[3]; P.ifx2 = 1.0f / K2[0]; P.ify2 = 1.0f / K2[1];
  P.ratio_factor = 1.5f * ctx->cfg.scale_factor;
  for (int l = 0; l < ASD_MAX_LEVELS; ++l) { P.sf[l] = ctx->scale[l]; P.sigma2[l] = ctx->sigma2[l]; }
  memcpy(m->h_idx, idx1, (size_t)n_pairs * sizeof(int));
  memcpy(m->h_idx + n_pairs, idx2, (size_t)n_pairs * sizeof(int));
  hipStream_t st = ctx->stream;
  ASD_HIP_CHECK(ctx, hipMemcpyAsync(m->d_idx, m->h_idx, (size_t)2 * n_pairs * sizeof(int), hipMemcpyHostToDevice, st));
  hipLaunchKernelGGL(k_triangulate, dim3((n_pairs + 63) / 64), dim3(64), 0, st, P, F1.d_kp, F2.d_kp, m->d_idx, m->d_idx + n_pairs,
                     n_pairs, m->d_x, m->d_ok);
  ASD_HIP_CHECK(ctx, hipGetLastError());
  ASD_HIP_CHECK(ctx, hipMemcpyAsync(m->h_x, m->d_x, (size_t)3 * n_pairs * sizeof(float), hipMemcpyDeviceToHost, st));
  ASD_HIP_CHECK(ctx, hipMemcpyAsync(m->h_ok, m->d_ok, (size_t)n_pairs, hipMemcpyDeviceToHost, st));
  ASD_HIP_CHECK(ctx, hipStreamSynchronize(st));
  memcpy(x3D, m->h_x, (size_t)3 * n_pairs * sizeof(float));
  memcpy(ok, m->h_ok, (size_t)n_pairs);
  if (n_ok) {
    int c = 0;
    for (int i = 0; i < n_pairs; ++i) c += ok[i];
    *n_ok = c;
  }
  return ASD_OK;
}

// right singular vector of the smallest singular value of n row-major 4x4 matrices (vt.row(3) of cv::SVD::compute)
int asd_frame_set_bow(asd_ctx* ctx, int32_t slot, const asd_feature_vector* fv) {
  if (!ctx || slot < 0 || slot >= ASD_MAX_FRAMES || !fv || fv->n_nodes < 0 || (fv->n_nodes > 0 && (!fv->node_id || !fv->start || !fv->idx))) return ASD_ERR_INVALID;
  AsdFrameSlot& F = ctx->frames[slot];
  if (!F.d_kp) { ctx->set_error("asd_frame_set_bow: frame slot %d not set", slot); return ASD_ERR_INVALID; }
  const int cap = ctx->cfg.max_patches, nn = fv->n_nodes;
  if (nn > cap || (nn > 0 && fv->start[nn] > F.n)) { ctx->set_error("asd_frame_set_bow: %d nodes / %d entries for a frame of %d keypoints", nn, nn ? fv->start[nn] : 0, F.n); return ASD_ERR_INVALID; }
  (void)hipSetDevice(ctx->cfg.device);
  if (!F.d_fv) ASD_HIP_CHECK(ctx, hipMalloc(&F.d_fv, ((size_t)4 * cap + 8) * sizeof(int)));
  std::vector<int> blk((size_t)4 * cap + 8, -1);
  int* node_id = blk.data(); int* start = node_id + cap; int* idx = start + cap + 1; int* kp_node = idx + cap;
  for (int k = 0; k < nn; ++k) {
    if (k > 0 && fv->node_id[k] <= fv->node_id[k - 1]) { ctx->set_error("asd_frame_set_bow: node ids must ascend"); return ASD_ERR_INVALID; }
    node_id[k] = fv->node_id[k];
    start[k] = fv->start[k];
    for (int t = fv->start[k]; t < fv->start[k + 1]; ++t) {
      const int i = fv->idx[t];
      if (i < 0 || i >= F.n) { ctx->set_error("asd_frame_set_bow: keypoint index %d out of range", i); return ASD_ERR_INVALID; }
      idx[t] = i;
      kp_node[i] = fv->node_id[k];
    }
  }
  start[nn] = nn ? fv->start[nn] : 0;
  ASD_HIP_CHECK(ctx, hipMemcpyAsync(F.d_fv, blk.data(), blk.size() * sizeof(int), hipMemcpyHostToDevice, ctx->stream));
  ASD_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  F.fv_nodes = nn;
  return ASD_OK;
}

int asd_create_map_points_batch(asd_ctx* ctx, int32_t slot_cur, const uint8_t* has_mp_cur, const float* Tcw_cur, const float* K_cur, int32_t n_nb,
                                const asd_kf_neighbor* nb, int32_t* matches12, int32_t* n_matches, float* x3D, uint8_t* ok) {
  if (!ctx || slot_cur < 0 || slot_cur >= ASD_MAX_FRAMES || !has_mp_cur || !Tcw_cur || !K_cur || n_nb < 0 ||
      (n_nb > 0 && (!nb || !matches12 || !n_matches || !x3D || !ok)))
    return ASD_ERR_INVALID;
  if (asd_track_busy(ctx, "asd_create_map_points_batch")) return ASD_ERR_INVALID;
  const AsdFrameSlot& C = ctx->frames[slot_cur];
  if (!C.d_kp || !C.d_fv) { ctx->set_error("asd_create_map_points_batch: slot %d needs asd_frame_set and asd_frame_set_bow", slot_cur); return ASD_ERR_INVALID; }
  if (n_nb == 0 || C.n == 0) return ASD_OK;
  (void)hipSetDevice(ctx->cfg.device);
  const int nc = C.n, cap = ctx->cfg.max_patches;
  size_t has_bytes = AsdDevBuf::padded(nc);
  for (int b = 0; b < n_nb; ++b) {
    if (nb[b].slot < 0 || nb[b].slot >= ASD_MAX_FRAMES || !nb[b].has_mp) return ASD_ERR_INVALID;
    const AsdFrameSlot& N = ctx->frames[nb[b].slot];
    if (!N.d_kp || !N.d_fv) { ctx->set_error("asd_create_map_points_batch: neighbour slot %d needs asd_frame_set and asd_frame_set_bow", nb[b].slot); return ASD_ERR_INVALID; }
    has_bytes += AsdDevBuf::padded(N.n);
  }
  // one upload block (flags of every keyframe, the neighbours' records, the triangulation parameters, the level tables), one result block
  hipStream_t st = ctx->stream;
  AsdXfer &up = ctx->up, &down = ctx->down;
  ASD_HIP_CHECK(ctx, up.begin(st, has_bytes + (size_t)n_nb * (sizeof(NbDev) + sizeof(TriParams) + 512) + 4096));
  ASD_HIP_CHECK(ctx, down.begin(st, (size_t)n_nb * nc * (4 + 12 + 1) + 4096));
  const size_t o_hc = up.add(has_mp_cur, nc);
  std::vector<NbDev> nd(n_nb);
  std::vector<TriParams> tp(n_nb);
  for (int b = 0; b < n_nb; ++b) {
    const AsdFrameSlot& N = ctx->frames[nb[b].slot];
    const size_t o = up.add(nb[b].has_mp, N.n);
    nd[b] = NbDev{N.d_kp, N.d_desc, N.d_fv, up.dev<uint8_t>(o), N.n, N.fv_nodes, cap, {}, nb[b].ex, nb[b].ey};
    memcpy(nd[b].F12, nb[b].F12, sizeof nd[b].F12);
    TriParams& P = tp[b];
    memcpy(P.T1, Tcw_cur, sizeof P.T1);
    memcpy(P.T2, nb[b].Tcw, sizeof P.T2);
    camera_centre(Tcw_cur, P.Rwc1, P.Ow1);
    camera_centre(nb[b].Tcw, P.Rwc2, P.Ow2);
    const float* K1 = K_cur; const float* K2 = nb[b].K;
    P.fx1 = K1[0]; P.fy1 = K1[1]; P.cx1 = K1[2]; P.cy1 = K1[3]; P.ifx1 = 1.0f / K1[0]; P.ify1 = 1.0f / K1[1];
    P.fx2 = K2[0]; P.fy2 = K2[1]; P.cx2 = K2[2]; P.cy2 = K2[3]; P.ifx2 = 1.0f / K2[0]; P.ify2 = 1.0f / K2[1];
    P.ratio_factor = 1.5f * ctx->cfg.scale_factor;
    for (int l = 0; l < ASD_MAX_LEVELS; ++l) { P.sf[l] = ctx->scale[l]; P.sigma2[l] = ctx->sigma2[l]; }
  }
  const size_t o_nb = up.add(nd.data(), nd.size() * sizeof(NbDev)), o_tp = up.add(tp.data(), tp.size() * sizeof(TriParams));
  float lv[2 * ASD_MAX_LEVELS];
  for (int l = 0; l < ASD_MAX_LEVELS; ++l) { lv[l] = ctx->scale[l]; lv[ASD_MAX_LEVELS + l] = ctx->sigma2[l]; }
  const size_t o_lv = up.add(lv, sizeof lv);
  const size_t o_m = down.reserve((size_t)n_nb * nc * 4), o_x = down.reserve((size_t)n_nb * nc * 12), o_ok = down.reserve((size_t)n_nb * nc);
  ASD_HIP_CHECK(ctx, up.upload(st));
  const int* fvc = C.d_fv;
  hipLaunchKernelGGL(k_tri_match_batch, dim3((nc + 3) / 4, n_nb), dim3(256), 0, st, C.d_kp, C.d_desc, fvc + 3 * (size_t)cap + 1, up.dev<uint8_t>(o_hc), nc,
                     up.dev<NbDev>(o_nb), up.dev<float>(o_lv), up.dev<float>(o_lv) + ASD_MAX_LEVELS, down.dev<int>(o_m));
  hipLaunchKernelGGL(k_triangulate_batch, dim3((nc + 63) / 64, n_nb), dim3(64), 0, st, up.dev<TriParams>(o_tp), C.d_kp, up.dev<NbDev>(o_nb), down.dev<int>(o_m), nc,
                     down.dev<float>(o_x), down.dev<unsigned char>(o_ok));
  ASD_HIP_CHECK(ctx, hipGetLastError());
  ASD_HIP_CHECK(ctx, down.download(st));
  ASD_HIP_CHECK(ctx, hipStreamSynchronize(st));
  memcpy(matches12, down.host<int>(o_m), (size_t)n_nb * nc * 4);
  memcpy(x3D, down.host<float>(o_x), (size_t)n_nb * nc * 12);
  memcpy(ok, down.host<unsigned char>(o_ok), (size_t)n_nb * nc);
  for (int b = 0; b < n_nb; ++b) {
    int c = 0;
    for (int i = 0; i < nc; ++i) c += matches12[(size_t)b * nc + i] >= 0;
    n_matches[b] = c;
  }
  return ASD_OK;
}

int asd_svd4_null(asd_ctx* ctx, int32_t n, const float* A, float* v) {
  if (!ctx || n < 0 || (n > 0 && (!A || !v))) return ASD_ERR_INVALID;
  if (n == 0) return ASD_OK;
  (void)hipSetDevice(ctx->cfg.device);
  ASD_HIP_CHECK(ctx, ctx->scratch.reserve(AsdDevBuf::padded((size_t)n * 64) + AsdDevBuf::padded((size_t)n * 16)));
  float* dA = ctx->scratch.carve<float>((size_t)n * 16);
  float* dv = ctx->scratch.carve<float>((size_t)n * 4);
  hipStream_t st = ctx->stream;
  hipError_t e = hipMemcpyAsync(dA, A, (size_t)n * 16 * sizeof(float), hipMemcpyHostToDevice, st);
  if (e == hipSuccess) {
    hipLaunchKernelGGL(k_svd4, dim3((n + 63) / 64), dim3(64), 0, st, dA, n, dv);
    e = hipMemcpyAsync(v, dv, (size_t)n * 4 * sizeof(float), hipMemcpyDeviceToHost, st);
  }
  if (e == hipSuccess) e = hipStreamSynchronize(st);
  if (e != hipSuccess) { ctx->set_error("asd_svd4_null: %s", hipGetErrorString(e)); return ASD_ERR_HIP; }
  return ASD_OK;
}

}  // extern "C"
