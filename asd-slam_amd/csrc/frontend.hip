// frontend.hip -- image front-end of ORBextractor::ExtractDesc on gfx950.
//
// Reference: src/vslam/src/ORBextractor.cc
//   ComputePyramid :1251-1276 (cv::resize INTER_LINEAR chained level to level)
//   ComputeKeyPointsOctTree :813-904 (cv::FAST per 30x30 cell, threshold 20 then 7)
//   IC_Angle :80-107, GaussianBlur + patch gather :1217-1231 / :1099-1126, tail :1234-1245
// All of this is byte / integer work bounded by HBM and launch latency, so the kernels are
// plain coalesced loads with wave-level ballot compaction; nothing here is reshaped into a GEMM.
//
// Observations that shape the GPU formulation (each justified in DESIGN.md):
//  * the 19 px pyramid border (copyMakeBorder) is never read by the monocular path: FAST runs on
//    [16, dim-16), the orientation disc (r = 15) and the 32x32 patch stay inside the image for
//    corners in [19, dim-20], and GaussianBlur runs on a border-less clone with its own
//    BORDER_REFLECT_101 -- so levels are stored without border;
//  * cv::FAST's score (fast_score.cpp cornerScore<16>) of a pixel that IS a corner at threshold t
//    does not depend on t, and 3x3 non-max suppression of pixels that pass t is unaffected by
//    neighbours that fail t; so one score map per level (at minThFAST) serves both the
//    iniThFAST pass and the per-cell minThFAST retry of ORBextractor.cc:858-866;
//  * the 6 px cell overlap makes the cells' FAST-valid interiors tile the level exactly, so
//    each pixel belongs to one cell and NMS only looks at neighbours of the same cell.
#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstring>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <functional>
#include <map>
#include <mutex>
#include <thread>

#include "ctx.h"
#include "quadtree.h"

namespace {

constexpr int kEdge = 19;         // EDGE_THRESHOLD (ORBextractor.cc:77)
constexpr int kMinBorder = 16;    // EDGE_THRESHOLD - 3 (:822)
constexpr int kPitchAlign = 64;

inline int cv_round(double v) { return (int)std::lrint(v); }
inline int cv_floor(double v) { return (int)std::floor(v); }
inline short sat_short(float v) { int i = cv_round(v); return (short)std::min(std::max(i, -32768), 32767); }

struct LevelDev {
  int w, h, pitch;
  int off;        // byte offset of the level in the pyramid / blur / score buffers
  int tile_start; // first 64x16 tile of this level in a whole-pyramid launch
  int tiles_x;
};
struct PyrDev {
  int nlevels;
  int total_tiles;
  LevelDev lv[ASD_MAX_LEVELS];
};
struct CellDev {  // FAST-valid interior of one 30x30 cell (image coordinates)
  short level, x0, x1, y0, y1, pad;
};

__constant__ int c_gauss[7];
__constant__ short2 c_disc[768];  // (u, v) offsets of the radius-15 orientation disc (umax table, :497-512)
__constant__ int c_ndisc;

// ---------------------------------------------------------------- E1: cv::resize 8U INTER_LINEAR
// imgwarp.cpp (3.2.0): 11-bit fixed-point coefficients, VResizeLinear<uchar,int,short,...>:
// dst = ((b0*(r0>>4))>>16 + (b1*(r1>>4))>>16 + 2) >> 2 with r = S[sx]*a0 + S[sx+1]*a1.
__global__ __launch_bounds__(256) void k_resize(const uint8_t* __restrict__ src, int sw, int sh, int spitch,
                                                uint8_t* __restrict__ dst, int dw, int dh, int dpitch,
                                                const short* __restrict__ xofs, const short* __restrict__ ialpha,
                                                const short* __restrict__ yofs, const short* __restrict__ ibeta) {
  const int x4 = (blockIdx.x * 64 + (threadIdx.x & 63)) * 4;
  const int dy = blockIdx.y * 4 + (threadIdx.x >> 6);
  if (dy >= dh || x4 >= dw) return;
  int sy0 = yofs[dy], sy1 = sy0 + 1;
  sy0 = min(max(sy0, 0), sh - 1);
  sy1 = min(max(sy1, 0), sh - 1);
  const int b0 = ibeta[dy * 2], b1 = ibeta[dy * 2 + 1];
  const uint8_t* S0 = src + (size_t)sy0 * spitch;
  const uint8_t* S1 = src + (size_t)sy1 * spitch;
  uint32_t packed = 0;
  for (int k = 0; k < 4; ++k) {
    const int dx = min(x4 + k, dw - 1);
    const int sx = xofs[dx], sx1 = min(sx + 1, sw - 1);
    const int a0 = ialpha[dx * 2], a1 = ialpha[dx * 2 + 1];
    const int r0 = S0[sx] * a0 + S0[sx1] * a1;
    const int r1 = S1[sx] * a0 + S1[sx1] * a1;
    const int v = (((b0 * (r0 >> 4)) >> 16) + ((b1 * (r1 >> 4)) >> 16) + 2) >> 2;
    packed |= (uint32_t)(v & 255) << (8 * k);
  }
  *reinterpret_cast<uint32_t*>(dst + (size_t)dy * dpitch + x4) = packed;
}

__device__ inline const LevelDev& find_level(const PyrDev& P, int tile, int& local) {
  int l = 0;
  while (l + 1 < P.nlevels && tile >= P.lv[l + 1].tile_start) ++l;
  local = tile - P.lv[l].tile_start;
  return P.lv[l];
}

// ---------------------------------------------------------------- E2a: FAST-9/16 score map
// score = (max over the 16 arcs of 9 contiguous ring pixels of min |v - p|, one-sided) - 1,
// exactly cornerScore<16> for a pixel that passes the segment test; 0 if score < min_th.
// Level 0 of the pyramid = the input image, copied row by row into the pitched level buffer by a kernel: the source is device memory
// or PAGE-LOCKED host memory read over PCIe (467 KB per KITTI frame).  hipMemcpy2DAsync from host memory made the front half's stream wait
// milliseconds per frame (round 4: 347 frames/s against 1187 with the frames resident), and a copy command costs this stream more than a
// launch anyway.  A thread moves four destination bytes (the level pitch is a multiple of 64, so they are aligned); the source row may
// start anywhere: two aligned dwords around it, funnel-shifted -- never a byte load over PCIe, never a read beyond the buffer's last dword.
__global__ __launch_bounds__(256) void k_copy_image(const uint8_t* __restrict__ src, int stride, int width, int height, uint8_t* __restrict__ dst, int pitch) {
  const int x4 = (blockIdx.x * 256 + threadIdx.x) * 4, y = blockIdx.y;
  if (x4 >= width || y >= height) return;
  const uintptr_t a = reinterpret_cast<uintptr_t>(src) + (size_t)y * stride + x4;
  const uintptr_t last = (reinterpret_cast<uintptr_t>(src) + (size_t)(height - 1) * stride + width - 1) & ~(uintptr_t)3;   // the last dword that holds image bytes
  const uintptr_t a0 = a & ~(uintptr_t)3;
  const unsigned sh = (unsigned)(a & 3) * 8;
  const uint32_t w0 = *reinterpret_cast<const uint32_t*>(a0);
  const uint32_t w1 = (sh && a0 + 4 <= last) ? *reinterpret_cast<const uint32_t*>(a0 + 4) : 0u;
  const uint32_t v = sh ? (w0 >> sh) | (w1 << (32 - sh)) : w0;
  *reinterpret_cast<uint32_t*>(dst + (size_t)y * pitch + x4) = v;   // (bytes beyond `width` inside the pitch are never read as image)
}

__global__ __launch_bounds__(256) void k_fast_score(PyrDev P, const uint8_t* __restrict__ pyr,
                                                    uint8_t* __restrict__ score, int min_th) {
  int local;
  const LevelDev& L = find_level(P, blockIdx.x, local);
  const int x = (local % L.tiles_x) * 64 + (threadIdx.x & 63);
  const int y = (local / L.tiles_x) * 16 + (threadIdx.x >> 6) * 4;
  const uint8_t* img = pyr + L.off;
  for (int r = 0; r < 4; ++r) {
    const int yy = y + r;
    if (x < kEdge || x >= L.w - kEdge || yy < kEdge || yy >= L.h - kEdge) continue;
    const uint8_t* p = img + (size_t)yy * L.pitch + x;
    const int pt = L.pitch;
    const int v = p[0];
    int d[16];
    d[0] = v - p[3 * pt];       d[1] = v - p[3 * pt + 1];   d[2] = v - p[2 * pt + 2];   d[3] = v - p[pt + 3];
    d[4] = v - p[3];            d[5] = v - p[-pt + 3];      d[6] = v - p[-2 * pt + 2];  d[7] = v - p[-3 * pt + 1];
    d[8] = v - p[-3 * pt];      d[9] = v - p[-3 * pt - 1];  d[10] = v - p[-2 * pt - 2]; d[11] = v - p[-pt - 3];
    d[12] = v - p[-3];          d[13] = v - p[pt - 3];      d[14] = v - p[2 * pt - 2];  d[15] = v - p[3 * pt - 1];
    // circular sliding-window min / max of width 9 by doubling
    int mn2[16], mx2[16], mn4[16], mx4[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) { mn2[k] = min(d[k], d[(k + 1) & 15]); mx2[k] = max(d[k], d[(k + 1) & 15]); }
#pragma unroll
    for (int k = 0; k < 16; ++k) { mn4[k] = min(mn2[k], mn2[(k + 2) & 15]); mx4[k] = max(mx2[k], mx2[(k + 2) & 15]); }
    int best_pos = -256, best_neg = 256;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      const int mn9 = min(min(mn4[k], mn4[(k + 4) & 15]), d[(k + 8) & 15]);
      const int mx9 = max(max(mx4[k], mx4[(k + 4) & 15]), d[(k + 8) & 15]);
      best_pos = max(best_pos, mn9);
      best_neg = min(best_neg, mx9);
    }
    const int s = max(best_pos, -best_neg) - 1;
    score[L.off + (size_t)yy * L.pitch + x] = (uint8_t)(s >= min_th ? s : 0);
  }
}

// ---------------------------------------------------------------- E2b: per-cell NMS + threshold retry
// One wave per cell.  pass 0: count survivors at ini_th / min_th; pass 1: write them compacted in
// cv::FAST's order (row-major inside the cell).  packed = x_rel | y_rel << 12 | score << 24 with
// coordinates relative to minBorder like vToDistributeKeys (:872-874).
template <bool WRITE>
__global__ __launch_bounds__(64) void k_cell_nms(const CellDev* __restrict__ cells, PyrDev P,
                                                 const uint8_t* __restrict__ score, int ini_th,
                                                 int* __restrict__ cell_count, const int* __restrict__ cell_off,
                                                 uint32_t* __restrict__ out, int out_cap) {
  const CellDev c = cells[blockIdx.x];
  const LevelDev& L = P.lv[c.level];
  const uint8_t* S = score + L.off;
  const int cw = c.x1 - c.x0, chh = c.y1 - c.y0, area = cw * chh;
  const int lane = threadIdx.x;
  int n20 = 0, n7 = 0;
  bool use20 = false;
  int base = 0;
  if (WRITE) {
    const int cnt = cell_count[blockIdx.x];
    use20 = cnt < 0;  // sign bit marks "ini_th produced corners"
    base = cell_off[blockIdx.x];
  }
  for (int i0 = 0; i0 < area; i0 += 64) {
    const int i = i0 + lane;
    bool keep = false, is20 = false;
    int x = 0, y = 0, s = 0;
    if (i < area) {
      x = c.x0 + i % cw;
      y = c.y0 + i / cw;
      s = S[(size_t)y * L.pitch + x];
      if (s > 0) {
        keep = true;
        for (int dy = -1; dy <= 1; ++dy)
          for (int dx = -1; dx <= 1; ++dx) {
            if (dx == 0 && dy == 0) continue;
            const int xx = x + dx, yy = y + dy;
            if (xx < c.x0 || xx >= c.x1 || yy < c.y0 || yy >= c.y1) continue;
            if (S[(size_t)yy * L.pitch + xx] >= s) keep = false;
          }
        is20 = keep && s >= ini_th;
      }
    }
    if (!WRITE) {
      n20 += __popcll(__ballot(is20));
      n7 += __popcll(__ballot(keep));
    } else {
      const bool w = use20 ? is20 : keep;
      const unsigned long long m = __ballot(w);
      if (w) {
        const int pos = base + __popcll(m & ((1ull << lane) - 1));
        // `out` is pinned host memory (the host reads the list right after this kernel): never past its end; the host sees
        // total > capacity in level_start and reports the overflow
        if (pos < out_cap) out[pos] = (uint32_t)(x - kMinBorder) | ((uint32_t)(y - kMinBorder) << 12) | ((uint32_t)s << 24);
      }
      base += __popcll(m);
    }
  }
  if (!WRITE && lane == 0) cell_count[blockIdx.x] = n20 > 0 ? (n20 | 0x80000000) : n7;
}

// exclusive scan of the per-cell counts (all levels) + per-level start offsets.  One workgroup.
__global__ __launch_bounds__(1024) void k_cell_scan(const int* __restrict__ cell_count, int ncells,
                                                    const int* __restrict__ level_cell_start, int nlevels,
                                                    int* __restrict__ cell_off, int* __restrict__ level_start,
                                                    int* __restrict__ level_start_host) {
  __shared__ int wsum[16];
  __shared__ int carry_s;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  if (t == 0) carry_s = 0;
  asd_syncthreads();
  for (int base = 0; base < ncells; base += 1024) {
    const int i = base + t;
    const int v = i < ncells ? (cell_count[i] & 0x7fffffff) : 0;
    int incl = v;
    for (int off = 1; off < 64; off <<= 1) {
      const int o = __shfl_up(incl, off);
      if (lane >= off) incl += o;
    }
    if (lane == 63) wsum[wave] = incl;
    asd_syncthreads();
    int wbase = 0;
    for (int w = 0; w < wave; ++w) wbase += wsum[w];
    const int carry = carry_s;
    if (i < ncells) cell_off[i] = carry + wbase + incl - v;
    asd_syncthreads();
    if (t == 1023) carry_s = carry + wbase + incl;
    asd_syncthreads();
  }
  if (t <= nlevels) {
    // level_start[l] = offset of the level's first cell; level_start[nlevels] = total
    int v;
    if (t == nlevels) v = carry_s;
    else {
      const int c = level_cell_start[t];
      v = c < ncells ? cell_off[c] : carry_s;
    }
    level_start[t] = v;
    level_start_host[t] = v;   // pinned: the host needs the counts together with the corner list, one synchronisation
  }
}

// ---------------------------------------------------------------- E5a: GaussianBlur 7x7 sigma 2, 8U
// filter.cpp fixed-point separable filter: 8-bit kernel (sums to 257/256), row pass in int,
// column pass (sum + 2^15) >> 16, BORDER_REFLECT_101 on the border-less level image.
__device__ inline int reflect101(int p, int len) {
  if (p < 0) p = -p;
  if (p >= len) p = 2 * len - 2 - p;
  return p;
}
__global__ __launch_bounds__(256) void k_blur7(PyrDev P, const uint8_t* __restrict__ pyr, uint8_t* __restrict__ blur) {
  __shared__ int rows[22][64];
  int local;
  const LevelDev& L = find_level(P, blockIdx.x, local);
  const int x = (local % L.tiles_x) * 64 + (threadIdx.x & 63);
  const int y0 = (local / L.tiles_x) * 16;
  const uint8_t* img = pyr + L.off;
  const int wq = threadIdx.x >> 6;
  const int xc = min(x, L.w - 1);
  int xs[7];
  for (int i = 0; i < 7; ++i) xs[i] = reflect101(xc + i - 3, L.w);
  for (int r = wq; r < 22; r += 4) {
    const int yy = reflect101(min(y0 + r - 3, L.h + 2), L.h);
    const uint8_t* row = img + (size_t)yy * L.pitch;
    int s = 0;
    for (int i = 0; i < 7; ++i) s += c_gauss[i] * row[xs[i]];
    rows[r][threadIdx.x & 63] = s;
  }
  asd_syncthreads();
  for (int r = wq; r < 16; r += 4) {
    const int yy = y0 + r;
    if (yy >= L.h || x >= L.w) continue;
    int s = 0;
    for (int i = 0; i < 7; ++i) s += c_gauss[i] * rows[r + i][threadIdx.x & 63];
    const int v = (s + (1 << 15)) >> 16;
    blur[L.off + (size_t)yy * L.pitch + x] = (uint8_t)min(max(v, 0), 255);
  }
}

// ---------------------------------------------------------------- E4 + E5b: orientation + patch gather
// cv::fastAtan2 (mathfuncs.cpp atanImpl<float>), un-fused f32 exactly as written there.
__device__ inline float fast_atan2_deg(float y, float x) {
  const float p1 = 0.9997878412794807f * (float)(180 / M_PI);
  const float p3 = -0.3258083974640975f * (float)(180 / M_PI);
  const float p5 = 0.1555786518463281f * (float)(180 / M_PI);
  const float p7 = -0.04432655554792128f * (float)(180 / M_PI);
  const float ax = fabsf(x), ay = fabsf(y);
  float a, c, c2;
  if (ax >= ay) {
    c = ay / (ax + (float)DBL_EPSILON);
    c2 = c * c;
    a = (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
  } else {
    c = ax / (ay + (float)DBL_EPSILON);
    c2 = c * c;
    a = 90.f - (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
  }
  if (x < 0) a = 180.f - a;
  if (y < 0) a = 360.f - a;
  return a;
}

// one wave per keypoint: kp = (x, y, level, -) in level pixel coordinates
__global__ __launch_bounds__(256) void k_angle_patch(PyrDev P, const uint8_t* __restrict__ pyr,
                                                     const uint8_t* __restrict__ blur, const short4* __restrict__ kps,
                                                     int n, float* __restrict__ angles, uint8_t* __restrict__ patches) {
  const int lane = threadIdx.x & 63;
  const int k = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (k >= n) return;
  const short4 kp = kps[k];
  const LevelDev& L = P.lv[kp.z];
  // IC_Angle (:80-107): integer moments over the disc, order-independent
  const uint8_t* center = pyr + L.off + (size_t)kp.y * L.pitch + kp.x;
  int m01 = 0, m10 = 0;
  for (int i = lane; i < c_ndisc; i += 64) {
    const short2 o = c_disc[i];
    const int val = center[o.y * L.pitch + o.x];
    m10 += o.x * val;
    m01 += o.y * val;
  }
  for (int off = 32; off >= 1; off >>= 1) {
    m01 += __shfl_xor(m01, off);
    m10 += __shfl_xor(m10, off);
  }
  if (lane == 0) angles[k] = fast_atan2_deg((float)m01, (float)m10);
  // 32x32 patch of the blurred level with the keypoint at [16][16] (:1113-1115)
  const int row = lane >> 1, half = lane & 1;
  const uint8_t* src = blur + L.off + (size_t)(kp.y - 16 + row) * L.pitch + (kp.x - 16 + half * 16);
  uint32_t w[4];
  for (int q = 0; q < 4; ++q)
    w[q] = (uint32_t)src[q * 4] | ((uint32_t)src[q * 4 + 1] << 8) | ((uint32_t)src[q * 4 + 2] << 16) |
           ((uint32_t)src[q * 4 + 3] << 24);
  *reinterpret_cast<uint4*>(patches + (size_t)k * 1024 + row * 32 + half * 16) = make_uint4(w[0], w[1], w[2], w[3]);
}


// Small persistent worker pool: the per-level quadtrees are independent, level 0 holds ~40 % of the
// corners, so 3 helpers + the calling thread bring the host part down to about the largest level.
class LevelPool {
 public:
  explicit LevelPool(int nworkers) {
    for (int i = 0; i < nworkers; ++i) th_.emplace_back([this] { worker(); });
  }
  ~LevelPool() {
    { std::lock_guard<std::mutex> l(m_); stop_ = true; ++gen_; }
    cv_.notify_all();
    for (auto& t : th_) t.join();
  }
  // runs job(0..ntasks-1), tasks handed out dynamically; returns when all are done
  void run(int ntasks, const std::function<void(int)>& job) {
    {
      std::lock_guard<std::mutex> l(m_);
      job_ = &job; ntasks_ = ntasks; next_.store(0); active_ = (int)th_.size(); ++gen_;
    }
    cv_.notify_all();
    drain();
    std::unique_lock<std::mutex> l(m_);
    done_cv_.wait(l, [this] { return active_ == 0; });
    job_ = nullptr;
  }
 private:
  void drain() {
    for (;;) {
      const int i = next_.fetch_add(1);
      if (i >= ntasks_) break;
      (*job_)(i);
    }
  }
  void worker() {
    unsigned long long seen = 0;
    for (;;) {
      {
        std::unique_lock<std::mutex> l(m_);
        cv_.wait(l, [&] { return gen_ != seen; });
        seen = gen_;
        if (stop_) return;
      }
      drain();
      { std::lock_guard<std::mutex> l(m_); --active_; }
      done_cv_.notify_one();
    }
  }
  std::vector<std::thread> th_;
  std::mutex m_;
  std::condition_variable cv_, done_cv_;
  const std::function<void(int)>* job_ = nullptr;
  std::atomic<int> next_{0};
  int ntasks_ = 0, active_ = 0;
  unsigned long long gen_ = 0;
  bool stop_ = false;
};

}  // namespace

// ------------------------------------------------------------------------------------------
struct ExtractSlot {
  uint8_t* d_patches = nullptr;
  float* d_desc = nullptr;
  float *d_angles = nullptr, *h_angles = nullptr, *h_desc = nullptr;  // h_* pinned
  int* h_range = nullptr;   // pinned, behind the angles: the forward's fp16x2 range flag (asdnet.hip, k_l2norm)
  hipEvent_t ev_begin = nullptr, ev_front = nullptr, ev_end = nullptr;
  bool owned = false;  // slot 0 aliases the ctx / FrontendState buffers
  // asd_extract_keep_pyramid: the submission's own copy of its pyramid (Frame::ComputeStereoMatches reads the level images around
  // the keypoints, Frame.cc:442-505, after both extractions -- by then the shared pyramid holds a later frame of the read-ahead)
  uint8_t* d_pyr_keep = nullptr;
  size_t pyr_keep_cap = 0;
};

struct FrontendState {
  int cfg_w = 0, cfg_h = 0;  // size the tables are currently built for
  PyrDev pyr{};
  size_t buf_bytes = 0;      // capacity of d_pyr / d_blur / d_score
  uint8_t *d_pyr = nullptr, *d_blur = nullptr, *d_score = nullptr;
  // resize tables, all levels concatenated
  short *d_xofs = nullptr, *d_ialpha = nullptr, *d_yofs = nullptr, *d_ibeta = nullptr;
  int tab_x_off[ASD_MAX_LEVELS] = {}, tab_y_off[ASD_MAX_LEVELS] = {};
  size_t tab_x_cap = 0, tab_y_cap = 0;
  // cells
  std::vector<CellDev> h_cells;
  int level_cell_start[ASD_MAX_LEVELS + 1] = {};
  CellDev* d_cells = nullptr;
  int cells_cap = 0;
  int *d_cell_count = nullptr, *d_cell_off = nullptr, *d_level_cell_start = nullptr, *d_level_start = nullptr;
  size_t corners_cap = 0;
  uint32_t* h_corners = nullptr;  // pinned
  int* h_level_start = nullptr;   // pinned
  short4 *d_kps = nullptr, *h_kps = nullptr;
  float *d_angles = nullptr, *h_angles = nullptr, *h_desc = nullptr;
  // last extract, host side
  std::vector<float> raw_x[ASD_MAX_LEVELS], raw_y[ASD_MAX_LEVELS], raw_r[ASD_MAX_LEVELS];
  std::vector<int> sel[ASD_MAX_LEVELS];
  LevelPool* pool = nullptr;
  ExtractSlot slot0;  // buffers of the synchronous asd_extract (aliases ctx->d_patches / d_desc and the arrays above)
};

static void level_dims(const asd_ctx* ctx, int w, int h, int level, int* lw, int* lh) {
  const float s = ctx->inv_scale[level];
  *lw = cv_round((float)w * s);  // ORBextractor.cc:1255-1256
  *lh = cv_round((float)h * s);
}

// one complete set of front-half state (pyramid, score map, blurred pyramid, tables, corner staging, quadtree scratch): the context
// owns one (ctx->fe: the synchronous entry points and the first extraction worker), the pipelined extractor a second one for its
// second worker, so that two front halves can be in flight
static int fe_alloc(asd_ctx* ctx, FrontendState** out) {
  FrontendState* fe = new FrontendState();
  *out = fe;
  fe->pool = new LevelPool(3);
  const int nl = ctx->cfg.n_levels, W = ctx->cfg.max_width, H = ctx->cfg.max_height;
  size_t bytes = 0, tx = 0, ty = 0;
  int ncell_max = 0;
  for (int l = 0; l < nl; ++l) {
    int lw, lh;
    level_dims(ctx, W, H, l, &lw, &lh);
    const int pitch = (lw + kPitchAlign - 1) / kPitchAlign * kPitchAlign;
    bytes += (size_t)pitch * (lh + 1);
    tx += lw; ty += lh;
    ncell_max += (lw / 30 + 2) * (lh / 30 + 2);
  }
  fe->buf_bytes = bytes + 4096;
  ASD_HIP_CHECK(ctx, hipMalloc(&fe->d_pyr, fe->buf_bytes));
  ASD_HIP_CHECK(ctx, hipMalloc(&fe->d_blur, fe->buf_bytes));
  ASD_HIP_CHECK(ctx, hipMalloc(&fe->d_score, fe->buf_bytes));
  fe->tab_x_cap = tx + 64; fe->tab_y_cap = ty + 64;
  ASD_HIP_CHECK(ctx, hipMalloc(&fe->d_xofs, fe->tab_x_cap * sizeof(short)));
  ASD_HIP_CHECK(ctx, hipMalloc(&fe->d_ialpha, fe->tab_x_cap * 2 * sizeof(short)));
  ASD_HIP_CHECK(ctx, hipMalloc(&fe->d_yofs, fe->tab_y_cap * sizeof(short)));
  ASD_HIP_CHECK(ctx, hipMalloc(&fe->d_ibeta, fe->tab_y_cap * 2 * sizeof(short)));
  fe->cells_cap = ncell_max;
  ASD_HIP_CHECK(ctx, hipMalloc(&fe->d_cells, ncell_max * sizeof(CellDev)));
  ASD_HIP_CHECK(ctx, hipMalloc(&fe->d_cell_count, ncell_max * sizeof(int)));
  ASD_HIP_CHECK(ctx, hipMalloc(&fe->d_cell_off, ncell_max * sizeof(int)));
  ASD_HIP_CHECK(ctx, hipMalloc(&fe->d_level_cell_start, (ASD_MAX_LEVELS + 1) * sizeof(int)));
  ASD_HIP_CHECK(ctx, hipMalloc(&fe->d_level_start, (ASD_MAX_LEVELS + 1) * sizeof(int)));
  // 3x3 strict NMS keeps at most one pixel per 2x2 block
  fe->corners_cap = bytes / 4 + 1024;
  ASD_HIP_CHECK(ctx, hipHostMalloc(&fe->h_corners, fe->corners_cap * sizeof(uint32_t)));
  ASD_HIP_CHECK(ctx, hipHostMalloc(&fe->h_level_start, (ASD_MAX_LEVELS + 1) * sizeof(int)));
  const size_t np = ctx->cfg.max_patches;
  ASD_HIP_CHECK(ctx, hipMalloc(&fe->d_kps, np * sizeof(short4)));
  ASD_HIP_CHECK(ctx, hipHostMalloc(&fe->h_kps, np * sizeof(short4)));
  ASD_HIP_CHECK(ctx, hipMalloc(&fe->d_angles, np * sizeof(float)));
  ASD_HIP_CHECK(ctx, hipHostMalloc(&fe->h_angles, (np + 16) * sizeof(float)));
  ASD_HIP_CHECK(ctx, hipHostMalloc(&fe->h_desc, np * 128 * sizeof(float)));
  return ASD_OK;
}
int frontend_alloc(asd_ctx* ctx) { return fe_alloc(ctx, &ctx->fe); }

static void fe_free(FrontendState* fe) {
  if (!fe) return;
  void* dev[] = {fe->d_pyr, fe->d_blur, fe->d_score, fe->d_xofs, fe->d_ialpha, fe->d_yofs, fe->d_ibeta, fe->d_cells,
                 fe->d_cell_count, fe->d_cell_off, fe->d_level_cell_start, fe->d_level_start,
                 fe->d_kps, fe->d_angles};
  for (void* p : dev) if (p) (void)hipFree(p);
  void* host[] = {fe->h_corners, fe->h_level_start, fe->h_kps, fe->h_angles, fe->h_desc};
  for (void* p : host) if (p) (void)hipHostFree(p);
  for (hipEvent_t* e : {&fe->slot0.ev_begin, &fe->slot0.ev_front, &fe->slot0.ev_end})
    if (*e) (void)hipEventDestroy(*e);
  delete fe->pool;
  delete fe;
}
void frontend_free(asd_ctx* ctx) { fe_free(ctx->fe); ctx->fe = nullptr; }

// Tables that depend on the image size: level geometry, resize coefficients, FAST cells.
static int configure_size(asd_ctx* ctx, FrontendState* fe, int w, int h) {
  if (fe->cfg_w == w && fe->cfg_h == h) return ASD_OK;
  const int nl = ctx->cfg.n_levels;
  // The constant-memory tables belong to the DEVICE (one copy per code object and device), not to a front-end state: uploaded once per
  // device under a lock.  (Per front-end state -- round 4 -- the second extraction worker's first job ran hipMemcpyToSymbol, a null-stream
  // synchronising call, while the first worker's kernels were reading the same symbols.  Every context computes the same umax table.)
  static std::mutex consts_mu;
  static AsdPerDeviceOnce consts_once;
  std::lock_guard<std::mutex> consts_lock(consts_mu);
  if (consts_once.need(ctx->cfg.device)) {
    // getGaussianKernel(7, 2, CV_32F) -> convertTo(CV_32S, 256) (smooth.cpp / filter.cpp, bits = 8)
    int gk[7];
    float cf[7];
    double sum = 0;
    for (int i = 0; i < 7; i++) { const double x = i - 3.0; cf[i] = (float)std::exp(-0.5 / 4.0 * x * x); sum += cf[i]; }
    sum = 1. / sum;
    for (int i = 0; i < 7; i++) { cf[i] = (float)(cf[i] * sum); gk[i] = cv_round(cf[i] * 256.f); }
    ASD_HIP_CHECK(ctx, hipMemcpyToSymbol(HIP_SYMBOL(c_gauss), gk, sizeof gk));
    std::vector<short2> disc;
    for (int v = -15; v <= 15; ++v) {
      const int d = ctx->umax[v < 0 ? -v : v];
      for (int u = -d; u <= d; ++u) disc.push_back(make_short2((short)u, (short)v));
    }
    const int nd = (int)disc.size();
    if (nd > 768) { ctx->set_error("orientation disc has %d pixels", nd); return ASD_ERR_INVALID; }
    ASD_HIP_CHECK(ctx, hipMemcpyToSymbol(HIP_SYMBOL(c_disc), disc.data(), nd * sizeof(short2)));
    ASD_HIP_CHECK(ctx, hipMemcpyToSymbol(HIP_SYMBOL(c_ndisc), &nd, sizeof nd));
    consts_once.done(ctx->cfg.device);
  }
  PyrDev& P = fe->pyr;
  P.nlevels = nl;
  int off = 0, tiles = 0;
  size_t tx = 0, ty = 0;
  for (int l = 0; l < nl; ++l) {
    LevelDev& L = P.lv[l];
    level_dims(ctx, w, h, l, &L.w, &L.h);
    if (L.w < 2 * kEdge + 8 || L.h < 2 * kEdge + 8) { ctx->set_error("image %dx%d too small for level %d", w, h, l); return ASD_ERR_INVALID; }
    L.pitch = (L.w + kPitchAlign - 1) / kPitchAlign * kPitchAlign;
    L.off = off;
    off += L.pitch * L.h;
    L.tiles_x = (L.w + 63) / 64;
    L.tile_start = tiles;
    tiles += L.tiles_x * ((L.h + 15) / 16);
    fe->tab_x_off[l] = (int)tx; fe->tab_y_off[l] = (int)ty;
    tx += L.w; ty += L.h;
  }
  P.total_tiles = tiles;
  if ((size_t)off > fe->buf_bytes || tx > fe->tab_x_cap || ty > fe->tab_y_cap) { ctx->set_error("image %dx%d exceeds ctx capacity", w, h); return ASD_ERR_CAPACITY; }
  // resize coefficient tables (imgwarp.cpp resize(), INTER_LINEAR, 8U)
  std::vector<short> xofs(tx), ialpha(tx * 2), yofs(ty), ibeta(ty * 2);
  for (int l = 1; l < nl; ++l) {
    const int sw = P.lv[l - 1].w, sh = P.lv[l - 1].h, dw = P.lv[l].w, dh = P.lv[l].h;
    const double inv_scale_x = (double)dw / sw, inv_scale_y = (double)dh / sh;
    const double scale_x = 1. / inv_scale_x, scale_y = 1. / inv_scale_y;
    for (int dx = 0; dx < dw; dx++) {
      float fx = (float)((dx + 0.5) * scale_x - 0.5);
      int sx = cv_floor(fx);
      fx -= sx;
      if (sx < 0) { fx = 0; sx = 0; }
      if (sx >= sw - 1) { fx = 0; sx = sw - 1; }
      const size_t o = fe->tab_x_off[l] + dx;
      xofs[o] = (short)sx;
      ialpha[o * 2] = sat_short((1.f - fx) * 2048);
      ialpha[o * 2 + 1] = sat_short(fx * 2048);
    }
    for (int dy = 0; dy < dh; dy++) {
      float fy = (float)((dy + 0.5) * scale_y - 0.5);
      const int sy = cv_floor(fy);
      fy -= sy;
      const size_t o = fe->tab_y_off[l] + dy;
      yofs[o] = (short)sy;
      ibeta[o * 2] = sat_short((1.f - fy) * 2048);
      ibeta[o * 2 + 1] = sat_short(fy * 2048);
    }
  }
  ASD_HIP_CHECK(ctx, hipMemcpy(fe->d_xofs, xofs.data(), tx * sizeof(short), hipMemcpyHostToDevice));
  ASD_HIP_CHECK(ctx, hipMemcpy(fe->d_ialpha, ialpha.data(), tx * 2 * sizeof(short), hipMemcpyHostToDevice));
  ASD_HIP_CHECK(ctx, hipMemcpy(fe->d_yofs, yofs.data(), ty * sizeof(short), hipMemcpyHostToDevice));
  ASD_HIP_CHECK(ctx, hipMemcpy(fe->d_ibeta, ibeta.data(), ty * 2 * sizeof(short), hipMemcpyHostToDevice));
  // FAST cells (ORBextractor.cc:817-854): FAST-valid interior of each cell window
  fe->h_cells.clear();
  for (int l = 0; l < nl; ++l) {
    fe->level_cell_start[l] = (int)fe->h_cells.size();
    const LevelDev& L = P.lv[l];
    const int minBX = kMinBorder, minBY = kMinBorder, maxBX = L.w - kEdge + 3, maxBY = L.h - kEdge + 3;
    const float width = (float)(maxBX - minBX), height = (float)(maxBY - minBY);
    const int nCols = (int)(width / 30.f), nRows = (int)(height / 30.f);
    if (nCols < 1 || nRows < 1) continue;
    const int wCell = (int)std::ceil(width / nCols), hCell = (int)std::ceil(height / nRows);
    for (int i = 0; i < nRows; i++) {
      const float iniY = (float)(minBY + i * hCell);
      float maxY = iniY + hCell + 6;
      if (iniY >= maxBY - 3) continue;
      if (maxY > maxBY) maxY = (float)maxBY;
      for (int j = 0; j < nCols; j++) {
        const float iniX = (float)(minBX + j * wCell);
        float maxX = iniX + wCell + 6;
        if (iniX >= maxBX - 6) continue;
        if (maxX > maxBX) maxX = (float)maxBX;
        const int cw = (int)maxX - (int)iniX, ch = (int)maxY - (int)iniY;
        if (cw < 7 || ch < 7) continue;  // cv::FAST finds nothing in such a window
        CellDev c;
        c.level = (short)l;
        c.x0 = (short)((int)iniX + 3); c.x1 = (short)((int)maxX - 3);
        c.y0 = (short)((int)iniY + 3); c.y1 = (short)((int)maxY - 3);
        c.pad = 0;
        fe->h_cells.push_back(c);
      }
    }
  }
  fe->level_cell_start[nl] = (int)fe->h_cells.size();
  for (int l = nl + 1; l <= ASD_MAX_LEVELS; ++l) fe->level_cell_start[l] = (int)fe->h_cells.size();
  if ((int)fe->h_cells.size() > fe->cells_cap) { ctx->set_error("cell table overflow"); return ASD_ERR_CAPACITY; }
  ASD_HIP_CHECK(ctx, hipMemcpy(fe->d_cells, fe->h_cells.data(), fe->h_cells.size() * sizeof(CellDev), hipMemcpyHostToDevice));
  ASD_HIP_CHECK(ctx, hipMemcpy(fe->d_level_cell_start, fe->level_cell_start, (ASD_MAX_LEVELS + 1) * sizeof(int), hipMemcpyHostToDevice));
  fe->cfg_w = w; fe->cfg_h = h;
  return ASD_OK;
}

// ---- one extraction = front half (pyramid .. patch gather) + back half (ASDNet + read-back) -----------
// The halves only share an ExtractSlot, so the front half of frame t+1 can run (own stream, host quadtree)
// while the back half of frame t keeps the matrix cores busy.
struct ExtractJob {
  const uint8_t* image = nullptr;
  bool on_device = false;
  int w = 0, h = 0, stride = 0, nfeat = 0;
  int n = 0, rc = ASD_OK;
};

static int slot_alloc(asd_ctx* ctx, ExtractSlot& S) {
  const size_t np = ctx->cfg.max_patches;
  ASD_HIP_CHECK(ctx, hipMalloc(&S.d_patches, np * 1024));
  ASD_HIP_CHECK(ctx, hipMalloc(&S.d_desc, np * 128 * sizeof(float)));
  ASD_HIP_CHECK(ctx, hipMalloc(&S.d_angles, np * sizeof(float)));
  ASD_HIP_CHECK(ctx, hipHostMalloc(&S.h_angles, (np + 16) * sizeof(float)));
  S.h_range = reinterpret_cast<int*>(S.h_angles + np);
  ASD_HIP_CHECK(ctx, hipHostMalloc(&S.h_desc, np * 128 * sizeof(float)));
  S.owned = true;
  return ASD_OK;
}
static int slot_events(asd_ctx* ctx, ExtractSlot& S) {
  if (!S.ev_begin) ASD_HIP_CHECK(ctx, hipEventCreate(&S.ev_begin));
  if (!S.ev_front) ASD_HIP_CHECK(ctx, hipEventCreate(&S.ev_front));
  if (!S.ev_end) ASD_HIP_CHECK(ctx, hipEventCreate(&S.ev_end));
  return ASD_OK;
}
static void slot_free(ExtractSlot& S) {
  if (S.owned) {
    if (S.d_patches) (void)hipFree(S.d_patches);
    if (S.d_desc) (void)hipFree(S.d_desc);
    if (S.d_angles) (void)hipFree(S.d_angles);
    if (S.h_angles) (void)hipHostFree(S.h_angles);
    if (S.h_desc) (void)hipHostFree(S.h_desc);
  }
  for (hipEvent_t* e : {&S.ev_begin, &S.ev_front, &S.ev_end})
    if (*e) { (void)hipEventDestroy(*e); *e = nullptr; }
  if (S.d_pyr_keep) (void)hipFree(S.d_pyr_keep);
  S = ExtractSlot();
}

// E1-E5: pyramid, FAST, quadtree, orientation, blur, patch gather.  Fills kps (angle still 0) and leaves the
// patches + angles in the slot; records S.ev_front on `st` after the last kernel.  Blocks the calling host
// thread twice (corner counts, corner list) but never waits for anything outside `st`.
static int extract_front(asd_ctx* ctx, FrontendState* fe, const ExtractJob& J, ExtractSlot& S, hipStream_t st, hipEvent_t ev_corners,
                         asd_keypoint* kps, int32_t* n_out) {
  const int width = J.w, height = J.h, stride = J.stride;
  int rc = configure_size(ctx, fe, width, height);
  if (rc != ASD_OK) return rc;
  const PyrDev& P = fe->pyr;
  const int nl = P.nlevels;
  int quota[ASD_MAX_LEVELS];
  const int nfeat = J.nfeat > 0 ? J.nfeat : ctx->cfg.n_features;
  if (nfeat > ctx->cfg.max_patches) { ctx->set_error("n_features %d exceeds max_patches", nfeat); return ASD_ERR_CAPACITY; }
  if (J.nfeat > 0) asd_compute_quotas(nfeat, ctx->cfg.scale_factor, nl, quota);
  else for (int l = 0; l < nl; ++l) quota[l] = ctx->features_per_level[l];

  const bool timing = getenv("ASD_TIMING") != nullptr;
  auto now = [] { return std::chrono::steady_clock::now(); };
  auto us = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double, std::micro>(b - a).count(); };
  const auto t_start = now();
  ASD_HIP_CHECK(ctx, hipEventRecord(S.ev_begin, st));
  // E1 pyramid
  {
    // device memory and page-locked host memory go through the copy kernel; pageable host memory (asd_extract with an ordinary buffer)
    // needs the runtime's staging copy
    bool by_kernel = J.on_device;
    const uint8_t* src_dev = J.image;
    if (!by_kernel) {
      hipPointerAttribute_t at{};
      by_kernel = hipPointerGetAttributes(&at, J.image) == hipSuccess && at.type == hipMemoryTypeHost && at.devicePointer != nullptr;
      if (!by_kernel) (void)hipGetLastError();   // (an unregistered pointer is reported as an error: not ours)
      // the device-side address of the mapping: equal to the host address for hipHostMalloc memory, not necessarily for hipHostRegister'ed memory
      else src_dev = static_cast<const uint8_t*>(at.devicePointer) + (J.image - static_cast<const uint8_t*>(at.hostPointer ? at.hostPointer : J.image));
    }
    if (by_kernel) {
      hipLaunchKernelGGL(k_copy_image, dim3((width + 1023) / 1024, height), dim3(256), 0, st, src_dev, stride, width, height, fe->d_pyr + P.lv[0].off, P.lv[0].pitch);
      ASD_HIP_CHECK(ctx, hipGetLastError());
    } else {
      ASD_HIP_CHECK(ctx, hipMemcpy2DAsync(fe->d_pyr + P.lv[0].off, P.lv[0].pitch, J.image, stride, width, height, hipMemcpyHostToDevice, st));
    }
  }
  for (int l = 1; l < nl; ++l) {
    const LevelDev &Sl = P.lv[l - 1], &D = P.lv[l];
    hipLaunchKernelGGL(k_resize, dim3((D.w + 255) / 256, (D.h + 3) / 4), dim3(256), 0, st, fe->d_pyr + Sl.off, Sl.w, Sl.h,
                       Sl.pitch, fe->d_pyr + D.off, D.w, D.h, D.pitch, fe->d_xofs + fe->tab_x_off[l],
                       fe->d_ialpha + 2 * fe->tab_x_off[l], fe->d_yofs + fe->tab_y_off[l],
                       fe->d_ibeta + 2 * fe->tab_y_off[l]);
  }
  if (ctx->keep_pyramid) {   // the submission's own copy (device to device, ~1.4 MB at KITTI size)
    size_t bytes = 0;
    for (int l = 0; l < nl; ++l) bytes = std::max(bytes, (size_t)P.lv[l].off + (size_t)P.lv[l].pitch * P.lv[l].h);
    bytes = (bytes + 15) / 16 * 16;
    if (S.pyr_keep_cap < fe->buf_bytes) {
      if (S.d_pyr_keep) (void)hipFree(S.d_pyr_keep);
      S.d_pyr_keep = nullptr; S.pyr_keep_cap = 0;
      ASD_HIP_CHECK(ctx, hipMalloc(&S.d_pyr_keep, fe->buf_bytes));
      S.pyr_keep_cap = fe->buf_bytes;
    }
    ASD_HIP_CHECK(ctx, asd_copy_rows(st, S.d_pyr_keep, fe->d_pyr, bytes));
  }
  // E2 FAST score, per-cell NMS, compaction
  const int ncells = (int)fe->h_cells.size();
  hipLaunchKernelGGL(k_fast_score, dim3(P.total_tiles), dim3(256), 0, st, P, fe->d_pyr, fe->d_score, ctx->cfg.min_th_fast);
  // the corner list and the per-level counts are written straight into pinned host memory by the kernels that produce them:
  // one synchronisation instead of counts D2H -> sync -> list D2H -> sync (the second round trip took 0.23 ms beside ASDNet)
  hipLaunchKernelGGL(k_cell_nms<false>, dim3(ncells), dim3(64), 0, st, fe->d_cells, P, fe->d_score, ctx->cfg.ini_th_fast,
                     fe->d_cell_count, fe->d_cell_off, fe->h_corners, 0);
  hipLaunchKernelGGL(k_cell_scan, dim3(1), dim3(1024), 0, st, fe->d_cell_count, ncells, fe->d_level_cell_start, nl,
                     fe->d_cell_off, fe->d_level_start, fe->h_level_start);
  hipLaunchKernelGGL(k_cell_nms<true>, dim3(ncells), dim3(64), 0, st, fe->d_cells, P, fe->d_score, ctx->cfg.ini_th_fast,
                     fe->d_cell_count, fe->d_cell_off, fe->h_corners, (int)fe->corners_cap);
  ASD_HIP_CHECK(ctx, hipGetLastError());
  ASD_HIP_CHECK(ctx, hipEventRecord(ev_corners, st));
  // E5a blur runs on the GPU while the host does the quadtree
  hipLaunchKernelGGL(k_blur7, dim3(P.total_tiles), dim3(256), 0, st, P, fe->d_pyr, fe->d_blur);
  ASD_HIP_CHECK(ctx, hipGetLastError());
  // E3 quadtree per level on the host (DistributeOctTree): wait for the corner list only
  ASD_HIP_CHECK(ctx, hipEventSynchronize(ev_corners));
  const auto t_counts = now();
  const int total = fe->h_level_start[nl];
  if ((size_t)total > fe->corners_cap) { ctx->set_error("corner buffer overflow"); return ASD_ERR_CAPACITY; }
  const auto t_corners = now();
  // unpack + quadtree per level in parallel (levels are independent), then assemble in level order
  const std::function<void(int)> level_job = [&](int l) {
    const int b = fe->h_level_start[l], e = fe->h_level_start[l + 1], cnt = e - b;
    auto &rx = fe->raw_x[l], &ry = fe->raw_y[l], &rr = fe->raw_r[l];
    rx.resize(cnt); ry.resize(cnt); rr.resize(cnt);
    for (int i = 0; i < cnt; ++i) {
      const uint32_t pk = fe->h_corners[b + i];
      rx[i] = (float)(pk & 0xfff);
      ry[i] = (float)((pk >> 12) & 0xfff);
      rr[i] = (float)(pk >> 24);
    }
    const LevelDev& L = P.lv[l];
    asd_distribute_octtree(rx.data(), ry.data(), rr.data(), cnt, kMinBorder, L.w - kEdge + 3, kMinBorder,
                           L.h - kEdge + 3, quota[l], fe->sel[l]);
  };
  fe->pool->run(nl, level_job);
  int n = 0;
  for (int l = 0; l < nl; ++l) {
    auto &rx = fe->raw_x[l], &ry = fe->raw_y[l], &rr = fe->raw_r[l];
    const int scaledPatchSize = (int)(31 * ctx->scale[l]);  // :887
    for (int idx : fe->sel[l]) {
      if (n >= ctx->cfg.max_patches) { ctx->set_error("more keypoints than max_patches"); return ASD_ERR_CAPACITY; }
      const float px = rx[idx] + kMinBorder, py = ry[idx] + kMinBorder;  // :894-895
      fe->h_kps[n] = make_short4((short)px, (short)py, (short)l, 0);
      asd_keypoint& k = kps[n];
      k.x = px; k.y = py;
      if (l != 0) { k.x *= ctx->scale[l]; k.y *= ctx->scale[l]; }  // :1236-1242
      k.size = (float)scaledPatchSize;
      k.response = rr[idx];
      k.octave = l;
      k.angle = 0.f;
      ++n;
    }
  }
  *n_out = n;
  const auto t_quad = now();
  if (n > 0) {
    // E4 + E5b: orientation + patch gather
    ASD_HIP_CHECK(ctx, hipMemcpyAsync(fe->d_kps, fe->h_kps, (size_t)n * sizeof(short4), hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(k_angle_patch, dim3((n + 3) / 4), dim3(256), 0, st, P, fe->d_pyr, fe->d_blur, fe->d_kps, n,
                       S.d_angles, S.d_patches);
    ASD_HIP_CHECK(ctx, hipGetLastError());
    // the angles go back to the host from HERE (the stream that made them), not behind the ASDNet forward: one copy command less between two
    // forwards on the ASDNet stream, whose every command boundary is 10-15 us of idle matrix cores (ev_end still covers it: the forward waits for ev_front)
    ASD_HIP_CHECK(ctx, hipMemcpyAsync(S.h_angles, S.d_angles, (size_t)n * sizeof(float), hipMemcpyDeviceToHost, st));
  }
  ASD_HIP_CHECK(ctx, hipEventRecord(S.ev_front, st));
  if (timing) fprintf(stderr, "[extract front] launch+counts %.0f us, corners D2H %.0f us (%d), quadtree %.0f us\n", us(t_start, t_counts), us(t_counts, t_corners), total, us(t_corners, t_quad));
  return ASD_OK;
}

// E6: ASDNet on the slot's patches + read-back of angles and descriptors, all enqueued on `st` behind ev_front
static int extract_back_enqueue(asd_ctx* ctx, ExtractSlot& S, int n, hipStream_t st) {
  ASD_HIP_CHECK(ctx, hipStreamWaitEvent(st, S.ev_front, 0));
  if (S.h_range) *S.h_range = 0;   // (the slot's previous user was waited for before the slot came round again)
  const int rc = asdnet_forward_device(ctx, S.d_patches, n, S.d_desc, st, S.h_range);
  if (rc != ASD_OK) return rc;
  ASD_HIP_CHECK(ctx, hipMemcpyAsync(S.h_desc, S.d_desc, (size_t)n * 128 * sizeof(float), hipMemcpyDeviceToHost, st));
  ASD_HIP_CHECK(ctx, hipEventRecord(S.ev_end, st));
  return ASD_OK;
}

// E7: wait for the back half, hand the results over (pinned staging -> caller's pageable buffers)
static int extract_finish(asd_ctx* ctx, ExtractSlot& S, int n, asd_keypoint* kps, float* desc) {
  ASD_HIP_CHECK(ctx, hipEventSynchronize(S.ev_end));
  if (desc) memcpy(desc, S.h_desc, (size_t)n * 128 * sizeof(float));  // else: the consumer reads the pinned staging itself
  for (int i = 0; i < n; ++i) kps[i].angle = S.h_angles[i];
  ASD_HIP_CHECK(ctx, hipEventElapsedTime(&ctx->ms_extract, S.ev_begin, S.ev_end));
  return asd_range_status(ctx, S.h_range, "asd_extract");
}

static int extract_check(asd_ctx* ctx, const uint8_t* image, int32_t width, int32_t height, int32_t stride) {
  if (!ctx || !image || stride < width) return ASD_ERR_INVALID;
  if (width > ctx->cfg.max_width || height > ctx->cfg.max_height) { ctx->set_error("image %dx%d exceeds ctx capacity %dx%d", width, height, ctx->cfg.max_width, ctx->cfg.max_height); return ASD_ERR_CAPACITY; }
  if (!ctx->weights_loaded) { ctx->set_error("asd_load_weights has not been called"); return ASD_ERR_NO_WEIGHTS; }
  return ASD_OK;
}

extern "C" {

static int extract_impl(asd_ctx* ctx, const uint8_t* image, bool image_on_device, int32_t width, int32_t height,
                        int32_t stride, int32_t n_features_override, asd_keypoint* kps, float* desc, int32_t* n_out) {
  if (!kps || !desc || !n_out) return ASD_ERR_INVALID;
  int rc = extract_check(ctx, image, width, height, stride);
  if (rc != ASD_OK) return rc;
  if (asd_extractor_busy(ctx, "asd_extract")) return ASD_ERR_INVALID;
  (void)hipSetDevice(ctx->cfg.device);
  ctx->adopt_pending = false;   // same stream as the adoption copy: ordered behind it
  FrontendState* fe = ctx->fe;
  ExtractSlot& S = fe->slot0;
  if (!S.d_patches) {  // slot 0 = the buffers asd_describe uses as well
    S.d_patches = ctx->d_patches; S.d_desc = ctx->d_desc;
    S.d_angles = fe->d_angles; S.h_angles = fe->h_angles; S.h_desc = fe->h_desc;
    S.h_range = reinterpret_cast<int*>(fe->h_angles + ctx->cfg.max_patches);
  }
  if ((rc = slot_events(ctx, S)) != ASD_OK) return rc;
  ExtractJob J;
  J.image = image; J.on_device = image_on_device; J.w = width; J.h = height; J.stride = stride; J.nfeat = n_features_override;
  int32_t n = 0;
  if ((rc = extract_front(ctx, fe, J, S, ctx->stream, ctx->ev2, kps, &n)) != ASD_OK) return rc;
  *n_out = n;
  ctx->last_n = n;
  ctx->d_desc_last = S.d_desc;
  ctx->d_pyr_view = nullptr;   // the shared pyramid holds this frame
  if (n == 0) return ASD_OK;
  if ((rc = extract_back_enqueue(ctx, S, n, ctx->stream)) != ASD_OK) return rc;
  return extract_finish(ctx, S, n, kps, desc);
}

int asd_extract(asd_ctx* ctx, const uint8_t* image, int32_t width, int32_t height, int32_t stride,
                int32_t n_features_override, asd_keypoint* kps, float* desc, int32_t* n_out) {
  return extract_impl(ctx, image, false, width, height, stride, n_features_override, kps, desc, n_out);
}

int asd_extract_device(asd_ctx* ctx, const uint8_t* d_image, int32_t width, int32_t height, int32_t stride,
                       int32_t n_features_override, asd_keypoint* kps, float* desc, int32_t* n_out) {
  return extract_impl(ctx, d_image, true, width, height, stride, n_features_override, kps, desc, n_out);
}

}  // extern "C"

// ---- pipelined extraction ---------------------------------------------------------------------------
// A worker thread owns two low-priority streams: front halves run on stream_f, back halves on stream_x.  Up to
// kQueueDepth submissions may be outstanding; the worker starts the front half of the next queued frame before it
// waits for the back half (ASDNet) of the previous one, so the GPU never idles during the host-side quadtree.
constexpr int kQueueDepth = ASD_EXTRACT_QUEUE;
constexpr int kSlots = kQueueDepth + 2;  // a waited frame's device descriptors stay valid for two more submissions

struct AsyncJob {
  ExtractJob job;
  int slot = 0;
  enum State { QUEUED, FRONT, BACK, DONE } state = QUEUED;
  std::vector<asd_keypoint> kps;
};

// Two workers (round 4): a front half is ~0.5 ms of one host thread -- fifteen small launches that queue beside ASDNet, one wait for
// the corner list, the quadtrees -- and enqueueing the back half another ~0.2 ms, so ONE worker turned a frame around every ~0.72 ms
// while the ASDNet kernels of a frame take 0.56: the matrix cores idled 0.12 ms per frame waiting for the next patches (rocprofv3,
// tools/extractor_timeline.py).  With a second set of front-half buffers two front halves are in flight; back halves are still
// enqueued in submission order (back_next) on the one ASDNet stream.
constexpr int kWorkers = 2;
struct AsyncExtract {
  std::thread th[kWorkers];
  std::mutex m;
  std::condition_variable cv;
  bool stop = false;
  hipStream_t stream_f[kWorkers] = {};
  hipEvent_t ev_corners[kWorkers] = {};
  FrontendState* fe[kWorkers] = {};   // fe[0] = ctx->fe, fe[1] owned here
  ExtractSlot slots[kSlots];
  AsyncJob jobs[kSlots];     // ring: job of submission s lives in jobs[s % kSlots]
  uint64_t submitted = 0, started = 0, waited = 0;  // counters: submitted >= started >= waited
  uint64_t back_next = 0;                           // submission whose back half is enqueued next
  bool hold_back = false;                           // asd_extract_hold: no further back half (ASDNet forward) is enqueued while set
  uint64_t last_view = ~0ull;                       // submission index of the most recently waited job
};

static void async_worker(asd_ctx* ctx, int wk) {
  AsyncExtract* ax = ctx->ax;
  (void)hipSetDevice(ctx->cfg.device);
  // The worker only ENQUEUES: front half of the next queued frame (two host syncs + the host quadtree), then its back half
  // (ASDNet + read-back) behind the previous frame's on the ASDNet stream -- and on to the next job.  It never waits for a back
  // half: the thread that calls asd_extract_wait synchronises on the frame's end event itself (wait_oldest).  Waiting here put
  // the hand-over (event wait, wake-up, angle copy) between one frame's ASDNet and the NEXT BUT ONE frame's front half, which
  // then started ~0.4 ms into the next ASDNet and no longer fitted under it (rocprofv3: the ASDNet queue idled 0.3 ms per frame).
  static const bool timing = getenv("ASD_TIMING") != nullptr;
  double t_idle = 0, t_front = 0, t_back = 0;
  long jobs = 0;
  for (;;) {
    AsyncJob* next = nullptr;
    uint64_t seq = 0;
    const auto w0 = std::chrono::steady_clock::now();
    {
      std::unique_lock<std::mutex> l(ax->m);
      ax->cv.wait(l, [&] { return ax->stop || ax->started < ax->submitted; });
      if (ax->stop) return;
      seq = ax->started;
      next = &ax->jobs[seq % kSlots];
      ++ax->started;
      next->state = AsyncJob::FRONT;
    }
    const auto w1 = std::chrono::steady_clock::now();
    ExtractSlot& S = ax->slots[next->slot];
    int32_t n = 0;
    int rc = extract_front(ctx, ax->fe[wk], next->job, S, ax->stream_f[wk], ax->ev_corners[wk], next->kps.data(), &n);
    const auto w2 = std::chrono::steady_clock::now();
    next->job.n = n;
    {   // back halves go onto the ASDNet stream in submission order, one thread at a time (asdnet_forward_device uses per-context state)
      std::unique_lock<std::mutex> l(ax->m);
      ax->cv.wait(l, [&] { return ax->stop || (ax->back_next == seq && !ax->hold_back); });
      if (ax->stop) return;
    }
    if (rc == ASD_OK && n > 0) rc = extract_back_enqueue(ctx, S, n, ctx->stream_x);
    if (timing) {
      const auto w3 = std::chrono::steady_clock::now();
      auto us = [](auto a, auto b) { return std::chrono::duration<double, std::micro>(b - a).count(); };
      t_idle += us(w0, w1); t_front += us(w1, w2); t_back += us(w2, w3);
      if (++jobs % 100 == 0) fprintf(stderr, "[extract worker %d] per frame: waiting for a job %.0f us, front half %.0f us, back half (turn + enqueue) %.0f us\n", wk, t_idle / jobs, t_front / jobs, t_back / jobs);
    }
    {
      std::lock_guard<std::mutex> l(ax->m);
      next->job.rc = rc;
      next->state = (rc != ASD_OK || n == 0) ? AsyncJob::DONE : AsyncJob::BACK;   // BACK: enqueued, the waiter synchronises
      ++ax->back_next;
    }
    ax->cv.notify_all();
  }
}

void frontend_async_shutdown(asd_ctx* ctx) {
  if (!ctx->ax) return;
  AsyncExtract* ax = ctx->ax;
  { std::lock_guard<std::mutex> l(ax->m); ax->stop = true; }
  ax->cv.notify_all();
  for (auto& th : ax->th) if (th.joinable()) th.join();
  if (ctx->stream_x) (void)hipStreamSynchronize(ctx->stream_x);
  for (hipStream_t st : ax->stream_f) if (st) (void)hipStreamSynchronize(st);
  for (auto& S : ax->slots) slot_free(S);
  for (hipEvent_t e : ax->ev_corners) if (e) (void)hipEventDestroy(e);
  for (hipStream_t st : ax->stream_f) if (st) (void)hipStreamDestroy(st);
  for (int w = 1; w < kWorkers; ++w) fe_free(ax->fe[w]);
  delete ax;
  ctx->ax = nullptr;
  if (ctx->stream_x) (void)hipStreamDestroy(ctx->stream_x);
  ctx->stream_x = nullptr;
}

bool asd_extractor_busy(asd_ctx* ctx, const char* who) {
  AsyncExtract* ax = ctx->ax;
  if (!ax) return false;
  std::lock_guard<std::mutex> l(ax->m);
  if (ax->submitted == ax->waited) return false;
  ctx->set_error("%s: %llu submission(s) of asd_extract_submit are outstanding -- the worker thread owns the pyramid, score and "
                 "activation buffers until asd_extract_wait has returned them all", who, (unsigned long long)(ax->submitted - ax->waited));
  return true;
}

extern "C" {

// While set, the workers enqueue no further ASDNet forward (front halves go on, forwards already enqueued finish): for a caller that wants
// LocalBundleAdjustment in line to run beside as little of the extractor as possible.  asd_extract_wait on a held job ends the hold.
// (Measured in the bench, round 4: LocalBA 2.67 instead of 2.73-2.92 ms, the step unchanged -- the extractor has to make the time up.)
int asd_extract_hold(asd_ctx* ctx, int32_t on) {
  if (!ctx) return ASD_ERR_INVALID;
  AsyncExtract* ax = ctx->ax;
  if (!ax) return ASD_OK;   // no read-ahead extraction has been started: nothing to hold
  { std::lock_guard<std::mutex> l(ax->m); ax->hold_back = on != 0; }
  ax->cv.notify_all();
  return ASD_OK;
}

int asd_extract_submit(asd_ctx* ctx, const uint8_t* image, int32_t device_resident, int32_t width, int32_t height,
                       int32_t stride, int32_t n_features_override) {
  int rc = extract_check(ctx, image, width, height, stride);
  if (rc != ASD_OK) return rc;
  (void)hipSetDevice(ctx->cfg.device);
  if (!ctx->ax) {
    // Streams of the extractor: ASDNet (back half) at the LOWEST priority -- the latency-critical tracking kernels on ctx->stream
    // go first --, the front half's fifteen small kernels at the HIGHEST: beside ASDNet at equal priority every one of them waited
    // 20-60 us for its turn (rocprofv3: k_resize 4 -> 20-66 us), the front half stretched from 0.35 to 0.75 ms and the ASDNet
    // queue idled waiting for it.
    // NO CU-masked stream (round 3).  Rounds 1-2 ran ASDNet on a stream made with hipExtStreamCreateWithCUMask that left 32 CUs to
    // the tracking stream (+6 % frames/s when re-measured in round 3: 980-1006 against 931-942).  ROCm 7.2 cannot tear such a stream
    // down: destroyed, the process deadlocks at exit (main thread in __hip_module_dtor on a HIP mutex, an HSA event thread in an
    // ioctl; round 2 also saw hipStreamDestroy itself hang); leaked, rocprofiler-sdk's static destructor segfaults inside
    // libhsa-runtime64 when the process was profiled.  Both reproduce without this library (tools/ubench/masked_stream_exit.hip,
    // tools/diag/masked_stream_py.py; profiles/r03_teardown_diagnostics.txt), so the mechanism is gone rather than worked around;
    // ASD_EXTRACT_RESERVE_CUS is ignored.  Looked at again in round 5 (the masked stream created once per process and leaked, A/B on one box):
    // with 16 CUs kept out of the ASDNet stream the tracking chain drops from 0.62 to 0.45 ms per frame -- and the tracking thread then
    // waits 0.15 ms per frame for the extractor (forward 0.59 -> 0.62 ms on 240 CUs plus its 50-70 us of read-back between forwards):
    // 1228-1248 frames/s against 1235-1238, nothing gained (1284 with the descriptor read-back skipped, i.e. +4 % is the bound of what
    // a cheaper read-back could add).  The read-back on a stream of its own was tried for that and is a trap: ONE MORE STREAM in the
    // process changes how the runtime spreads the streams over its hardware queues -- with the stream merely existing the ASDNet forward
    // read 0.72 instead of 0.59 ms (1030-1100 frames/s).  Six streams (tracking, frame construction, ASDNet, two front halves, the null
    // stream's copies) map to six queues today (rocprofv3 kernel trace: Queue_Id 1-6); a seventh does not get its own.
    // With a read-ahead of five frames instead of three beside the 16 reserved CUs the waiting goes away (tracking 0.555 ms per frame) but
    // LocalBA, which then never has the chip to itself, takes 3.3 instead of 2.75 ms: 1285-1289 against 1240-1257 frames/s at K = 300 and
    // nothing at the driver's K = 20 (two LocalBAs in twenty frames) -- not worth a stream that cannot be destroyed.
    // The extractor is built completely -- streams, events, slots, worker thread -- before ctx->ax publishes it: a failure on
    // the way leaves ctx->ax null and everything released, so the next call starts over instead of queueing a job no worker
    // will ever take (asd_extract_wait would block forever).
    int prio_least = 0, prio_greatest = 0;
    (void)hipDeviceGetStreamPriorityRange(&prio_least, &prio_greatest);
    AsyncExtract* ax = new AsyncExtract();
    hipStream_t sx = nullptr;
    auto build = [&]() -> int {
      ASD_HIP_CHECK(ctx, hipStreamCreateWithPriority(&sx, hipStreamDefault, prio_least));
      ax->fe[0] = ctx->fe;
      for (int w = 0; w < kWorkers; ++w) {
        ASD_HIP_CHECK(ctx, hipStreamCreateWithPriority(&ax->stream_f[w], hipStreamDefault, prio_greatest));
        ASD_HIP_CHECK(ctx, hipEventCreate(&ax->ev_corners[w]));
        int r;
        if (w > 0 && (r = fe_alloc(ctx, &ax->fe[w])) != ASD_OK) return r;
      }
      for (int i = 0; i < kSlots; ++i) {
        int r;
        if ((r = slot_alloc(ctx, ax->slots[i])) != ASD_OK) return r;
        if ((r = slot_events(ctx, ax->slots[i])) != ASD_OK) return r;
        ax->jobs[i].slot = i;
        ax->jobs[i].kps.resize(ctx->cfg.max_patches);
      }
      return ASD_OK;
    };
    if ((rc = build()) != ASD_OK) {
      for (auto& S : ax->slots) slot_free(S);
      for (hipEvent_t e : ax->ev_corners) if (e) (void)hipEventDestroy(e);
      for (hipStream_t st : ax->stream_f) if (st) { (void)hipStreamDestroy(st); }
      for (int w = 1; w < kWorkers; ++w) fe_free(ax->fe[w]);
      if (sx) { (void)hipStreamDestroy(sx); }
      delete ax;
      return rc;
    }
    ctx->stream_x = sx;
    ctx->ax = ax;
    static const int n_workers = [] { const char* e = getenv("ASD_EXTRACT_WORKERS"); const int v = e ? atoi(e) : kWorkers; return v < 1 ? 1 : (v > kWorkers ? kWorkers : v); }();
    for (int w = 0; w < n_workers; ++w) ax->th[w] = std::thread(async_worker, ctx, w);
  }
  AsyncExtract* ax = ctx->ax;
  if (ctx->adopt_pending) {
    // an asd_frame_set(desc == NULL) may still be copying out of an extraction buffer: the ASDNet stream -- the only writer of
    // those buffers -- waits for that copy before anything enqueued from now on runs (no host wait: a stream-side dependency)
    ASD_HIP_CHECK(ctx, hipStreamWaitEvent(ctx->stream_x, ctx->ev_adopt, 0));
    ctx->adopt_pending = false;
  }
  {
    std::lock_guard<std::mutex> l(ax->m);
    if (ax->submitted - ax->waited >= (uint64_t)kQueueDepth) {
      ctx->set_error("asd_extract_submit: %d submissions are already outstanding", kQueueDepth);
      return ASD_ERR_CAPACITY;
    }
    AsyncJob& a = ax->jobs[ax->submitted % kSlots];
    a.job = ExtractJob();
    a.job.image = image; a.job.on_device = device_resident != 0; a.job.w = width; a.job.h = height; a.job.stride = stride;
    a.job.nfeat = n_features_override;
    a.state = AsyncJob::QUEUED;
    ++ax->submitted;
  }
  ax->cv.notify_all();
  return ASD_OK;
}

// oldest outstanding submission: blocks until its back half has been enqueued, synchronises on its end event in THIS thread and
// hands the job over (nullptr + error code on failure)
static AsyncJob* wait_oldest(asd_ctx* ctx, int* rc) {
  AsyncExtract* ax = ctx->ax;
  AsyncJob* a;
  bool back;
  {
    std::unique_lock<std::mutex> l(ax->m);
    if (ax->waited == ax->submitted) { ctx->set_error("asd_extract_wait: nothing was submitted"); *rc = ASD_ERR_INVALID; return nullptr; }
    a = &ax->jobs[ax->waited % kSlots];
    if (ax->hold_back && a->state != AsyncJob::BACK && a->state != AsyncJob::DONE) { ax->hold_back = false; ax->cv.notify_all(); }   // (a wait ends a hold: it could never return otherwise)
    ax->cv.wait(l, [&] { return a->state == AsyncJob::BACK || a->state == AsyncJob::DONE; });
    back = a->state == AsyncJob::BACK;   // (read under the lock: the worker writes the state under it)
  }
  if (back) {   // only this thread moves a job from BACK to DONE
    (void)hipSetDevice(ctx->cfg.device);
    const int frc = extract_finish(ctx, ax->slots[a->slot], a->job.n, a->kps.data(), nullptr);
    std::lock_guard<std::mutex> l(ax->m);
    if (frc != ASD_OK) a->job.rc = frc;
    a->state = AsyncJob::DONE;
  }
  {
    std::lock_guard<std::mutex> l(ax->m);
    ++ax->waited;
    if (a->job.rc == ASD_OK) ax->last_view = ax->waited - 1;   // submission index of the job handed over (asd_extract_last_view reads it under the lock)
  }
  if (a->job.rc != ASD_OK) { *rc = a->job.rc; return nullptr; }
  ctx->last_n = a->job.n;
  ctx->d_desc_last = ax->slots[a->slot].d_desc;
  ctx->d_pyr_view = ctx->keep_pyramid ? ax->slots[a->slot].d_pyr_keep : nullptr;
  *rc = ASD_OK;
  return a;
}

int asd_extract_wait(asd_ctx* ctx, asd_keypoint* kps, float* desc, int32_t* n_out) {
  if (!ctx || !ctx->ax || !kps || !desc || !n_out) return ASD_ERR_INVALID;
  int rc;
  AsyncJob* a = wait_oldest(ctx, &rc);
  if (!a) return rc;
  const int n = a->job.n;
  memcpy(kps, a->kps.data(), (size_t)n * sizeof(asd_keypoint));
  memcpy(desc, ctx->ax->slots[a->slot].h_desc, (size_t)n * 128 * sizeof(float));
  *n_out = n;
  return ASD_OK;
}

int asd_extract_wait_view(asd_ctx* ctx, const asd_keypoint** kps, const float** desc, int32_t* n_out) {
  if (!ctx || !ctx->ax || !kps || !desc || !n_out) return ASD_ERR_INVALID;
  int rc;
  AsyncJob* a = wait_oldest(ctx, &rc);
  if (!a) return rc;
  *kps = a->kps.data();
  *desc = ctx->ax->slots[a->slot].h_desc;
  *n_out = a->job.n;
  return ASD_OK;
}

uint64_t asd_extract_last_view(const asd_ctx* ctx) {
  if (!ctx || !ctx->ax) return ~0ull;
  std::lock_guard<std::mutex> l(ctx->ax->m);
  return ctx->ax->last_view;
}

int32_t asd_extract_view_valid(asd_ctx* ctx, uint64_t view_id) {
  if (!ctx || !ctx->ax) return 0;
  AsyncExtract* ax = ctx->ax;
  std::lock_guard<std::mutex> l(ax->m);
  // submission view_id lives in ring entry view_id % kSlots; submission view_id + kSlots takes that entry (keypoints) and its slot
  // (descriptors, angles)
  return view_id < ax->waited && ax->submitted <= view_id + (uint64_t)kSlots ? 1 : 0;
}

int asd_get_level_size(const asd_ctx* ctx, int32_t level, int32_t* width, int32_t* height) {
  if (!ctx || !ctx->fe || !width || !height || level < 0 || level >= ctx->fe->pyr.nlevels) return ASD_ERR_INVALID;
  *width = ctx->fe->pyr.lv[level].w;
  *height = ctx->fe->pyr.lv[level].h;
  return ASD_OK;
}

int asd_get_level_image(asd_ctx* ctx, int32_t level, int32_t blurred, uint8_t* out) {
  if (!ctx || !ctx->fe || !out || level < 0 || level >= ctx->fe->pyr.nlevels) return ASD_ERR_INVALID;
  if (asd_extractor_busy(ctx, "asd_get_level_image")) return ASD_ERR_INVALID;
  const LevelDev& L = ctx->fe->pyr.lv[level];
  const uint8_t* src = (blurred ? ctx->fe->d_blur : ctx->fe->d_pyr) + L.off;
  ASD_HIP_CHECK(ctx, hipMemcpy2DAsync(out, L.w, src, L.pitch, L.w, L.h, hipMemcpyDeviceToHost, ctx->stream));
  ASD_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  return ASD_OK;
}

int asd_get_raw_corners(asd_ctx* ctx, int32_t level, int32_t capacity, float* x, float* y, float* response, int32_t* n_out) {
  if (!ctx || !ctx->fe || !n_out || level < 0 || level >= ctx->fe->pyr.nlevels) return ASD_ERR_INVALID;
  if (asd_extractor_busy(ctx, "asd_get_raw_corners")) return ASD_ERR_INVALID;
  const FrontendState* fe = ctx->fe;
  const int n = std::min((int)fe->raw_x[level].size(), capacity);
  for (int i = 0; i < n; ++i) {
    if (x) x[i] = fe->raw_x[level][i];
    if (y) y[i] = fe->raw_y[level][i];
    if (response) response[i] = fe->raw_r[level][i];
  }
  *n_out = n;
  return ASD_OK;
}

}  // extern "C"

// ---- stereo association -----------------------------------------------------------------------
// Frame::ComputeStereoMatches (reference Frame.cc:360-535).  Every left keypoint is independent (no claims), so one
// wave owns one left keypoint: lanes stride over the right keypoints of its row band, evaluate the exact-order
// descriptor distance, the wave picks the first minimum; then the 11 SAD values of the 11x11 windows on the keypoint's
// pyramid level are computed by the lanes together (integer arithmetic: exact in any order) and lane 0 does the
// parabola fit in the reference's f32 operation order.  The median filter over the accepted matches (:517-531) is a
// sort of <= N pairs and stays on the host.
namespace {

struct StereoArgs {
  PyrDev P;
  const uint8_t *pyr_l, *pyr_r;
  const float4 *kp_l, *kp_r;     // (x, y, octave bits, -)
  const float *desc_l, *desc_r;
  const int *row_start, *row_items;
  int n_l, n_rows;
  float maxD, mbf;
  float scale[ASD_MAX_LEVELS], inv_scale[ASD_MAX_LEVELS];
  float* u_right; float* depth; int* sad;  // outputs per left keypoint
};

__global__ __launch_bounds__(256) void k_stereo_match(StereoArgs a) {
  const int lane = threadIdx.x & 63;
  const int iL = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (iL >= a.n_l) return;
  const float4 kl = a.kp_l[iL];
  const float uL = kl.x, vL = kl.y;
  const int levelL = __float_as_int(kl.z);
  float out_u = -1.0f, out_d = -1.0f;
  int out_sad = -1;
  const int row = (int)vL;
  do {
    if (row < 0 || row >= a.n_rows) break;
    const int c0 = a.row_start[row], c1 = a.row_start[row + 1];
    if (c0 == c1) break;
    const float minU = uL - a.maxD, maxU = uL;  // minD = 0
    if (maxU < 0) break;
    // best right keypoint: strict `<` from TH_HIGH in candidate order == smallest (distance, position)
    float best = 1.5f;
    int best_pos = 0x7fffffff;
    const float4* dl = reinterpret_cast<const float4*>(a.desc_l + (size_t)iL * 128);
    for (int p = c0 + lane; p < c1; p += 64) {
      const int iR = a.row_items[p];
      const float4 kr = a.kp_r[iR];
      const int oR = __float_as_int(kr.z);
      if (oR < levelL - 1 || oR > levelL + 1) continue;
      if (!(kr.x >= minU && kr.x <= maxU)) continue;
      const float4* dr = reinterpret_cast<const float4*>(a.desc_r + (size_t)iR * 128);
      float acc = 0.f;
      for (int k = 0; k < 32; ++k) {
        const float4 x = dl[k], y = dr[k];
        float d;
        d = x.x - y.x; acc = acc + d * d;
        d = x.y - y.y; acc = acc + d * d;
        d = x.z - y.z; acc = acc + d * d;
        d = x.w - y.w; acc = acc + d * d;
      }
      if (acc < best) { best = acc; best_pos = p; }  // positions ascend within a lane: first minimum kept
    }
    for (int off = 32; off >= 1; off >>= 1) {
      const float ob = __shfl_xor(best, off);
      const int op = __shfl_xor(best_pos, off);
      if (ob < best || (ob == best && op < best_pos)) { best = ob; best_pos = op; }
    }
    if (!(best < 1.0f) || best_pos == 0x7fffffff) break;  // thOrbDist = int((TH_HIGH + TH_LOW) / 2) = 1 (:366)
    const int iR = a.row_items[best_pos];
    const float uR0 = a.kp_r[iR].x;
    const float sf = a.inv_scale[levelL];
    const float scaleduL = roundf(uL * sf), scaledvL = roundf(vL * sf), scaleduR0 = roundf(uR0 * sf);
    const LevelDev Lv = a.P.lv[levelL];
    const float iniu = scaleduR0 + 5 - 5, endu = scaleduR0 + 5 + 5 + 1;
    if (iniu < 0 || endu >= Lv.w) break;
    const int xl = (int)scaleduL, yl = (int)scaledvL, xr = (int)scaleduR0;
    // 11 x 11 windows: the library's own bounds (the reference relies on cv::Mat range checks throwing)
    if (xl - 5 < 0 || xl + 5 >= Lv.w || yl - 5 < 0 || yl + 5 >= Lv.h || xr - 10 < 0 || xr + 10 >= Lv.w) break;
    const uint8_t* L0 = a.pyr_l + Lv.off;
    const uint8_t* R0 = a.pyr_r + Lv.off;
    const int cl = L0[yl * Lv.pitch + xl];
    int sad[11];
#pragma unroll
    for (int s = 0; s < 11; ++s) sad[s] = 0;
    for (int q = lane; q < 121; q += 64) {
      const int dy = q / 11 - 5, dx = q % 11 - 5;
      const int il = (int)L0[(yl + dy) * Lv.pitch + xl + dx] - cl;
      const uint8_t* rrow = R0 + (yl + dy) * Lv.pitch + xr + dx;
#pragma unroll
      for (int s = 0; s < 11; ++s) {
        const int cr = R0[yl * Lv.pitch + xr + (s - 5)];
        const int ir = (int)rrow[s - 5] - cr;
        sad[s] += abs(il - ir);
      }
    }
#pragma unroll
    for (int s = 0; s < 11; ++s)
      for (int off = 32; off >= 1; off >>= 1) sad[s] += __shfl_xor(sad[s], off);
    float bestS = 2147483648.0f;  // (float)INT_MAX
    int bestinc = 0;
    float vd[11];
#pragma unroll
    for (int s = 0; s < 11; ++s) {
      vd[s] = (float)sad[s];
      if (vd[s] < bestS) { bestS = vd[s]; bestinc = s - 5; }
    }
    if (bestinc == -5 || bestinc == 5) break;
    float d1 = 0.f, d2 = 0.f, d3 = 0.f;
#pragma unroll
    for (int s = 1; s < 10; ++s)
      if (s - 5 == bestinc) { d1 = vd[s - 1]; d2 = vd[s]; d3 = vd[s + 1]; }
    const float deltaR = (d1 - d3) / (2.0f * (d1 + d3 - 2.0f * d2));
    if (deltaR < -1 || deltaR > 1) break;
    float bestuR = a.scale[levelL] * ((float)scaleduR0 + (float)bestinc + deltaR);
    float disparity = uL - bestuR;
    if (disparity >= 0.f && disparity < a.maxD) {
      if (disparity <= 0) { disparity = 0.01; bestuR = uL - 0.01; }
      out_d = a.mbf / disparity;
      out_u = bestuR;
      out_sad = (int)bestS;
    }
  } while (false);
  if (lane == 0) { a.u_right[iL] = out_u; a.depth[iL] = out_d; a.sad[iL] = out_sad; }
}

}  // namespace

extern "C" int asd_stereo_match(asd_ctx* ctx_left, asd_ctx* ctx_right, int32_t slot_left, int32_t slot_right, float mb, float mbf,
                                float* u_right, float* depth, int32_t* n_matched) {
  if (!ctx_left || !ctx_right || !u_right || !depth || !n_matched || slot_left < 0 || slot_left >= ASD_MAX_FRAMES || slot_right < 0 ||
      slot_right >= ASD_MAX_FRAMES || !(mb > 0) || !(mbf > 0))
    return ASD_ERR_INVALID;
  asd_ctx* ctx = ctx_left;
  if (ctx_left->cfg.device != ctx_right->cfg.device) { ctx->set_error("asd_stereo_match: both contexts must live on the same device"); return ASD_ERR_INVALID; }
  FrontendState *fl = ctx_left->fe, *fr = ctx_right->fe;
  // (the pipelined extractor's second worker has a front-half state of its own: whichever is configured holds the level geometry)
  if (fl && fl->cfg_w == 0 && ctx_left->ax && ctx_left->ax->fe[1]) fl = ctx_left->ax->fe[1];
  if (fr && fr->cfg_w == 0 && ctx_right->ax && ctx_right->ax->fe[1]) fr = ctx_right->ax->fe[1];
  if (!fl || !fr || fl->cfg_w == 0 || fl->cfg_w != fr->cfg_w || fl->cfg_h != fr->cfg_h) {
    ctx->set_error("asd_stereo_match: extract the left and the right image (same size) on the two contexts first");
    return ASD_ERR_INVALID;
  }
  const AsdFrameSlot &FL = ctx->frames[slot_left], &FR = ctx->frames[slot_right];
  const int N = FL.n, Nr = FR.n;
  *n_matched = 0;
  for (int i = 0; i < N; ++i) { u_right[i] = -1.0f; depth[i] = -1.0f; }
  if (N == 0 || Nr == 0) return ASD_OK;
  if (!FL.d_kp || !FR.d_kp) { ctx->set_error("asd_stereo_match: frame slot not set"); return ASD_ERR_INVALID; }
  // the SAD windows are read from both contexts' pyramids: those must still hold the images the two frame slots came from -- either
  // the shared pyramids of two synchronous extractions (nothing submitted since), or the kept copies of the two submissions waited for
  // last (asd_extract_keep_pyramid: valid like that submission's descriptors, for two further submissions)
  const bool views = ctx_left->d_pyr_view && ctx_right->d_pyr_view;
  if (!views && (asd_extractor_busy(ctx_left, "asd_stereo_match") || asd_extractor_busy(ctx_right, "asd_stereo_match"))) {
    if (ctx != ctx_right) ctx->set_error("%s", ctx_right->last_error());
    return ASD_ERR_INVALID;
  }
  (void)hipSetDevice(ctx->cfg.device);
  // inside an asd_prep_async bracket (frame construction beside the tracking stages in flight) the call runs on the context's second
  // stream; its device buffers are its own either way (the stages in flight own ctx->scratch)
  hipStream_t st = asd_prep_stream(ctx);
  AsdDevBuf& scratch = ctx->stereo_scratch;
  // row table (:370-387): right keypoint iR is a candidate for every row within 2 * scale[octave] of its y
  const int nRows = fl->pyr.lv[0].h;
  std::vector<int> row_start(nRows + 1, 0);
  std::vector<int> lo(Nr), hi(Nr);
  for (int iR = 0; iR < Nr; ++iR) {
    const float kpY = FR.kps[iR].y, r = 2.0f * ctx->scale[FR.kps[iR].octave];
    lo[iR] = std::max((int)std::floor(kpY - r), 0);
    hi[iR] = std::min((int)std::ceil(kpY + r), nRows - 1);
    for (int y = lo[iR]; y <= hi[iR]; ++y) ++row_start[y + 1];
  }
  for (int y = 0; y < nRows; ++y) row_start[y + 1] += row_start[y];
  std::vector<int> row_items(std::max(row_start[nRows], 1));
  {
    std::vector<int> cur(row_start.begin(), row_start.end() - 1);
    for (int iR = 0; iR < Nr; ++iR)
      for (int y = lo[iR]; y <= hi[iR]; ++y) row_items[cur[y]++] = iR;  // ascending iR inside a row, like push_back order
  }
  hipError_t e = scratch.reserve(AsdDevBuf::padded(row_start.size() * sizeof(int)) + AsdDevBuf::padded(row_items.size() * sizeof(int)) +
                                 3 * AsdDevBuf::padded((size_t)N * sizeof(float)));
  int* d_rs = scratch.carve<int>(row_start.size());
  int* d_ri = scratch.carve<int>(row_items.size());
  int* d_sad = scratch.carve<int>(N);
  float* d_u = scratch.carve<float>(N);
  float* d_d = scratch.carve<float>(N);
  if (e == hipSuccess) e = hipMemcpyAsync(d_rs, row_start.data(), row_start.size() * sizeof(int), hipMemcpyHostToDevice, st);
  if (e == hipSuccess) e = hipMemcpyAsync(d_ri, row_items.data(), row_items.size() * sizeof(int), hipMemcpyHostToDevice, st);
  std::vector<int> sad(N);
  if (e == hipSuccess) {
    StereoArgs a;
    a.P = fl->pyr;
    a.pyr_l = views ? ctx_left->d_pyr_view : fl->d_pyr; a.pyr_r = views ? ctx_right->d_pyr_view : fr->d_pyr;
    a.kp_l = FL.d_kp; a.kp_r = FR.d_kp;
    a.desc_l = FL.d_desc; a.desc_r = FR.d_desc;
    a.row_start = d_rs; a.row_items = d_ri;
    a.n_l = N; a.n_rows = nRows;
    a.maxD = mbf / mb; a.mbf = mbf;
    for (int l = 0; l < ASD_MAX_LEVELS; ++l) { a.scale[l] = ctx->scale[l]; a.inv_scale[l] = ctx->inv_scale[l]; }
    a.u_right = d_u; a.depth = d_d; a.sad = d_sad;
    // the right pyramid was written on ctx_right's stream: make sure it is complete (a kept copy is: its submission was waited for)
    if (!views) e = hipStreamSynchronize(ctx_right->stream);
    if (e == hipSuccess) {
      hipLaunchKernelGGL(k_stereo_match, dim3((N + 3) / 4), dim3(256), 0, st, a);
      e = hipGetLastError();
    }
  }
  if (e == hipSuccess) e = hipMemcpyAsync(u_right, d_u, (size_t)N * sizeof(float), hipMemcpyDeviceToHost, st);
  if (e == hipSuccess) e = hipMemcpyAsync(depth, d_d, (size_t)N * sizeof(float), hipMemcpyDeviceToHost, st);
  if (e == hipSuccess) e = hipMemcpyAsync(sad.data(), d_sad, (size_t)N * sizeof(int), hipMemcpyDeviceToHost, st);
  if (e == hipSuccess) e = hipStreamSynchronize(st);
  if (e != hipSuccess) { ctx->set_error("asd_stereo_match: %s", hipGetErrorString(e)); return ASD_ERR_HIP; }
  // median filter on the SAD values (:517-531)
  std::vector<std::pair<int, int>> vDistIdx;
  for (int i = 0; i < N; ++i)
    if (sad[i] >= 0) vDistIdx.emplace_back(sad[i], i);
  if (vDistIdx.empty()) return ASD_OK;
  std::sort(vDistIdx.begin(), vDistIdx.end());
  const float median = (float)vDistIdx[vDistIdx.size() / 2].first;
  const float thDist = 1.5f * 1.4f * median;
  int kept = (int)vDistIdx.size();
  for (int i = (int)vDistIdx.size() - 1; i >= 0; --i) {
    if (vDistIdx[i].first < thDist) break;
    u_right[vDistIdx[i].second] = -1;
    depth[vDistIdx[i].second] = -1;
    --kept;
  }
  *n_matched = kept;
  return ASD_OK;
}
