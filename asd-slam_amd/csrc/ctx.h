// ctx.h -- internal context of libasdhip (not part of the C ABI).
#pragma once
#include <hip/hip_runtime.h>

#include <atomic>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/asd_slam.h"

#define ASD_HIP_CHECK(ctx, expr)                                                          \
  do {                                                                                    \
    hipError_t _e = (expr);                                                               \
    if (_e != hipSuccess) {                                                               \
      (ctx)->set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
      return ASD_ERR_HIP;                                                                 \
    }                                                                                     \
  } while (0)

// hipFuncSetAttribute (dynamic LDS above 64 KB) applies to the CURRENT device only: one bit per device ordinal says whether this
// process has asked for it there (contexts on different devices of one process launch the same kernels)
struct AsdPerDeviceOnce {
  std::atomic<uint64_t> bits{0};
  bool need(int dev) const { return !((bits.load(std::memory_order_acquire) >> (dev & 63)) & 1); }
  void done(int dev) { bits.fetch_or(1ull << (dev & 63), std::memory_order_release); }
};

// Grow-only device buffer.  asd_ctx::scratch is the per-call workspace of entry points that need temporary device
// arrays (one API call at a time per context, so one arena is enough); carve() hands out 256-B aligned pieces.
struct AsdDevBuf {
  void* p = nullptr;
  size_t cap = 0, used = 0;
  hipError_t reserve(size_t bytes) {
    used = 0;
    if (bytes <= cap) return hipSuccess;
    if (p) (void)hipFree(p);
    p = nullptr; cap = 0;
    const size_t want = bytes + bytes / 4 + 4096;
    const hipError_t e = hipMalloc(&p, want);
    if (e == hipSuccess) cap = want;
    return e;
  }
  template <typename T> T* carve(size_t count) {
    char* q = static_cast<char*>(p) + used;
    used += (count * sizeof(T) + 255) / 256 * 256;
    return reinterpret_cast<T*>(q);
  }
  static size_t padded(size_t bytes) { return (bytes + 255) / 256 * 256; }
  void release() { if (p) (void)hipFree(p); p = nullptr; cap = used = 0; }
};


// One pinned host block + its device twin.  A hipMemcpyAsync costs ~15-25 us on the tracking stream (host call + the hop to
// the copy engine), whatever its size, so everything a call uploads -- queries, flags, tables -- is packed into ONE block and
// travels in one copy; results come back the same way (AsdXfer used in the other direction).
// device-side copy kernel (matcher.hip): dst / src 16-B aligned, bytes % 16 == 0; src may be pinned host memory
hipError_t asd_copy_rows(hipStream_t st, void* dst, const void* src, size_t bytes);

// Workgroup barrier with the wait for this wave's own LDS operations spelled out.  hipcc (ROCm 7.2) left the `s_waitcnt lgkmcnt(0)`
// of __syncthreads() out in front of ONE barrier of k_pose_opt -- the head of the Levenberg loop, behind thread 0's store of the
// "round over" flag: the other waves could pass the barrier and read the flag (or the trial pose) before the store had landed.
// Invisible alone; beside the extractor's ASDNet workgroups, which keep the CU's LDS queues full, 1-3 PoseOptimization calls in a
// thousand came back with a pose 1e-14..1e-12 off or a garbage inlier count (tools/diag/pose_determinism.py; found through a
// flaky tests/test_bench_host.py).  Every barrier in the library is written through these two, and `make check-isa` checks that
// every s_barrier in the device code has the wait directly in front of it.
#if defined(__HIPCC__)
#define asd_syncthreads() do { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __syncthreads(); } while (0)
__device__ inline int asd_syncthreads_or(int pred) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); return __syncthreads_or(pred); }
#endif


// What Tracking does between its two stages, on the device (asd_track_frame): the matches PoseOptimization marked as outliers are
// dropped (Tracking.cc:695-714), the optimised pose becomes the frame's pose (Optimizer.cc:405-407 -> Frame::SetPose: Tcw as
// Converter::toCvMat(SE3Quat) gives it, mOw = -Rcw^T tcw, Frame.cc:150-158), and the map points the motion-model stage matched -- kept
// or dropped as outliers (mnLastFrameSeen, :705-707 and :811-823) -- are marked so that SearchLocalPoints does not project them again.  Runs as the tail of the motion-model stage's k_pose_opt (the
// workgroup that has just written the flags and the pose); outputs feed k_frustum_queries, k_window_search and k_pose_opt of the
// local-map stage.  The conversions are asd_pose7_to_tcw's and track_local_points_impl's expressions, operation for operation (the
// caller compiles this with -ffp-contract=off semantics: every product below is rounded before it is added -- see the pragma), so the
// stage behind sees the bits a host in between would have handed it.
// A window query = one GetFeaturesInArea call + the descriptor it is matched against (k_window_search's input; matcher.hip)
struct AsdWinQuery { float x, y, r; int min_level, max_level, qrow; };
struct AsdBetweenArgs {
  int n_cur, n_last, n_cand;
  const int* match1;        // [n_cur] last-frame keypoint or -1
  const double* io1;        // the motion-model stage's result block (device copy): pose[7], n_bad, outlier byte per keypoint, edge count
  const float* Xw_last;     // [n_last][3]
  const int* last_cand;     // [n_last] or null: candidate index of the map point last keypoint i holds
  float T_pred[16];         // the pose the motion-model search projected with (kept when it made fewer than 3 matches)
  uint8_t* occ;             // out [n_cur]: the keypoint keeps its map point
  float* cur_Xw;            // out [n_cur][3]: that map point's position
  uint8_t* skip;            // out [n_cand]
  float* T1;                // out [19]: Tcw (row major 4x4), Ow
};
#if defined(__HIPCC__)
// one candidate of the bank form: the arithmetic of asd_frustum, operation for operation (f32 with the reference's two double accumulations;
// no contraction; IEEE division and square root), the level from comparisons with thresholds the host derived from its own logf
// (F = k_frustum_queries' argument block: no private copy of the level tables)
template <class F>
__device__ inline void asd_frustum_bank_point(const F& a, const float* T_dev, const uint8_t* skip, const int q) {
#pragma clang fp contract(off)
  AsdWinQuery Q{0.f, 0.f, 0.f, 0, 0, -1};
  const int row = a.rows[q];
  const float4 A0 = reinterpret_cast<const float4*>(a.attr)[2 * (size_t)row], A1 = reinterpret_cast<const float4*>(a.attr)[2 * (size_t)row + 1];
  a.xw_out[3 * (size_t)q] = A0.x; a.xw_out[3 * (size_t)q + 1] = A0.y; a.xw_out[3 * (size_t)q + 2] = A0.z;
  if (!skip || !skip[q]) {   // (skip == null: every candidate is looked at)
    float T[16], Ow[3];
    for (int i = 0; i < 16; ++i) T[i] = T_dev[i];
    for (int i = 0; i < 3; ++i) Ow[i] = T_dev[16 + i];
    const float P[3] = {A0.x, A0.y, A0.z}, Pn[3] = {A0.w, A1.x, A1.y};
    const float min_dist = A1.z, max_dist = A1.w;
    float Pc[3];
    for (int r = 0; r < 3; ++r) {
      const float t0 = T[r * 4 + 0] * P[0] + T[r * 4 + 1] * P[1] + T[r * 4 + 2] * P[2];
      Pc[r] = (float)((double)t0 + (double)T[r * 4 + 3]);
    }
    bool ok = !(Pc[2] < 0.0f);
    const float invz = 1.0f / Pc[2];
    const float u = a.fx * Pc[0] * invz + a.cx, v = a.fy * Pc[1] * invz + a.cy;
    if (u < a.min_x || u > a.max_x || v < a.min_y || v > a.max_y) ok = false;
    const float maxD = 1.2f * max_dist, minD = 0.8f * min_dist;
    const float PO[3] = {P[0] - Ow[0], P[1] - Ow[1], P[2] - Ow[2]};
    const double nn = (double)PO[0] * PO[0] + (double)PO[1] * PO[1] + (double)PO[2] * PO[2];
    const float dist = (float)sqrt(nn);
    if (dist < minD || dist > maxD) ok = false;
    const double dot = (double)PO[0] * Pn[0] + (double)PO[1] * Pn[1] + (double)PO[2] * Pn[2];
    const float vc = (float)(dot / dist);
    if (vc < a.cos_limit) ok = false;
    if (ok) {
      const float ratio = max_dist / dist;
      int lvl = 0;
      for (int k = 1; k < a.n_levels; ++k) lvl += ratio >= a.level_thr[k];
      float r = vc > 0.998 ? 2.5f : 4.0f;  // RadiusByViewingCos (ORBmatcher.cc:126-132)
      if (a.bfactor) r *= a.th;
      Q = AsdWinQuery{u, v, r * a.scale[lvl], lvl - 1, lvl, row};
    }
  }
  a.queries[q] = Q;
}
// every thread of ONE workgroup (nt threads) calls this behind a barrier that follows the workgroup's stores to io1
__device__ inline void asd_between_body(const AsdBetweenArgs& a_dev, const int t, const int nt) {
#pragma clang fp contract(off)
  const AsdBetweenArgs a = a_dev;
  __shared__ int s_nmatch;
  if (t == 0) s_nmatch = 0;
  for (int c = t; c < a.n_cand; c += nt) a.skip[c] = 0;
  asd_syncthreads();
  const uint8_t* outl = reinterpret_cast<const uint8_t*>(a.io1 + 8);
  int mine = 0;
  for (int j = t; j < a.n_cur; j += nt) {
    const int i = a.match1[j];
    mine += i >= 0;
    const int src = i >= 0 ? i : 0;
    for (int k = 0; k < 3; ++k) a.cur_Xw[3 * (size_t)j + k] = a.n_last > 0 ? a.Xw_last[3 * (size_t)src + k] : 0.f;
    const bool keep = i >= 0 && !outl[j];
    a.occ[j] = keep ? 1 : 0;
    // the map point of EVERY motion-model match is out of the local-map search: a kept one because the frame holds it (Tracking.cc:811-823),
    // an outlier's because TrackWithMotionModel stamps it with the frame's id when it drops the match (pMP->mnLastFrameSeen, :705-707) --
    // the keypoint itself is free again (occ = 0)
    if (i >= 0 && a.last_cand) { const int c = a.last_cand[i]; if (c >= 0 && c < a.n_cand) a.skip[c] = 1; }
  }
  if (mine) atomicAdd(&s_nmatch, mine);
  asd_syncthreads();
  if (t == 0) {
    float T[16];
    if (s_nmatch >= 3) {   // asd_pose7_to_tcw
      const double* p = a.io1;
      const double x = p[0], y = p[1], z = p[2], w = p[3];
      const double tx = 2 * x, ty = 2 * y, tz = 2 * z;
      const double twx = tx * w, twy = ty * w, twz = tz * w, txx = tx * x, txy = ty * x, txz = tz * x, tyy = ty * y, tyz = tz * y, tzz = tz * z;
      const double R[9] = {1 - (tyy + tzz), txy - twz, txz + twy, txy + twz, 1 - (txx + tzz), tyz - twx, txz - twy, tyz + twx, 1 - (txx + tyy)};
      for (int i = 0; i < 3; ++i) {
        for (int j = 0; j < 3; ++j) T[i * 4 + j] = (float)R[i * 3 + j];
        T[i * 4 + 3] = (float)p[4 + i];
      }
      T[12] = T[13] = T[14] = 0.f;
      T[15] = 1.f;
    } else {
      for (int i = 0; i < 16; ++i) T[i] = a.T_pred[i];
    }
    for (int i = 0; i < 16; ++i) a.T1[i] = T[i];
    for (int i = 0; i < 3; ++i) {  // mOw = -mRcw.t()*mtcw (Frame.cc:157): transposed gemm accumulates in double
      double sum = 0;
      for (int k = 0; k < 3; ++k) sum += (double)T[k * 4 + i] * (double)T[k * 4 + 3];
      a.T1[16 + i] = (float)(-1.0 * sum);
    }
  }
}
#endif

struct AsdXfer {
  char* h = nullptr;
  char* d = nullptr;
  size_t cap = 0, used = 0;
  hipError_t begin(hipStream_t st, size_t bytes) {   // room for `bytes` (+ alignment slack); contents undefined
    used = 0;
    bytes += 4096;
    if (bytes <= cap) return hipSuccess;
    if (h) { (void)hipStreamSynchronize(st); (void)hipHostFree(h); (void)hipFree(d); }
    h = nullptr; d = nullptr; cap = 0;
    const size_t want = bytes + bytes / 2;
    hipError_t e = hipHostMalloc(&h, want);
    if (e == hipSuccess) e = hipMalloc(&d, want);
    if (e == hipSuccess) cap = want;
    return e;
  }
  size_t reserve(size_t bytes) { const size_t off = used; used += (bytes + 255) / 256 * 256; return off; }
  size_t add(const void* src, size_t bytes) { const size_t off = reserve(bytes); memcpy(h + off, src, bytes); return off; }
  size_t zeros(size_t bytes) { const size_t off = reserve(bytes); memset(h + off, 0, bytes); return off; }
  // a KERNEL reads the pinned block over PCIe and writes the device twin: an async H2D copy command costs this stream 50-60 us
  // of latency whatever its size (rocprofv3: the gap in front of every chain's first kernel), a 150 KB kernel copy ~10
  hipError_t upload(hipStream_t st) {
    if (!used) return hipSuccess;
    static const bool by_kernel = getenv("ASD_UPLOAD_COPY") == nullptr;   // ASD_UPLOAD_COPY=1: copy commands instead
    return by_kernel ? asd_copy_rows(st, d, h, (used + 15) / 16 * 16) : hipMemcpyAsync(d, h, used, hipMemcpyHostToDevice, st);
  }
  hipError_t download(hipStream_t st) { return used ? hipMemcpyAsync(h, d, used, hipMemcpyDeviceToHost, st) : hipSuccess; }
  template <typename T> T* dev(size_t off) { return reinterpret_cast<T*>(d + off); }
  template <typename T> T* host(size_t off) { return reinterpret_cast<T*>(h + off); }
  void release() { if (h) (void)hipHostFree(h); if (d) (void)hipFree(d); h = d = nullptr; cap = used = 0; }
};

struct AsdFrameSlot {
  int n = 0;
  float min_x = 0, max_x = 0, min_y = 0, max_y = 0;
  float inv_w = 0, inv_h = 0;
  // host mirrors (the order-dependent parts of the matchers run on the host)
  std::vector<asd_keypoint> kps;
  std::vector<int32_t> cell_start;  // [64*48+1] CSR, cell index = ix*48+iy (reference loop order)
  std::vector<int32_t> cell_items;
  // device mirrors (descriptors, keypoints as (x, y, octave bits, -), CSR grid) + pinned staging
  float* d_desc = nullptr;
  float4* d_kp = nullptr;
  int32_t* d_cell_start = nullptr;
  int32_t* d_cell_items = nullptr;
  // DBoW2::FeatureVector of a keyframe (asd_frame_set_bow): node ids ascending, CSR over keypoint indices, node id per keypoint (-1 none)
  int32_t* d_fv = nullptr;       // one block: node_id [n] | start [n + 1] | idx [n] | kp_node [n] (capacity max_patches each)
  int32_t fv_nodes = 0;
  char* h_stage = nullptr;
  hipEvent_t ev_staged = nullptr;  // recorded behind the slot's H2D copies: h_stage may be rewritten once it has completed
};

struct asd_ctx {
  asd_config cfg{};
  hipStream_t stream = nullptr;
  hipStream_t stream_x = nullptr;    // back halves (ASDNet) of the pipelined extractor
  // every stream this context's entry points or worker threads launch into (registered when created)
  hipStream_t stream_prep = nullptr; // asd_prep_async: frame construction (grid / descriptor / bank copies) beside the stages in flight
  hipEvent_t ev_prep = nullptr;
  bool prep_on = false;
  struct AsyncExtract* ax = nullptr;
  int num_cu = 256;
  std::string err;

  // ---- extractor tables (ORBextractor.cc:459-512)
  float scale[ASD_MAX_LEVELS], inv_scale[ASD_MAX_LEVELS], sigma2[ASD_MAX_LEVELS], inv_sigma2[ASD_MAX_LEVELS];
  int features_per_level[ASD_MAX_LEVELS];
  // MapPoint::PredictScale without a logarithm: level_thr[k] (k = 1 .. n_levels-1) is the smallest float ratio r with
  // ceil(logf(r) / logf(scaleFactor)) >= k, found by bisection over float bit patterns with the host's own logf at
  // asd_ctx_create, so the device predicts the level by comparisons and gets libm's answer bit for bit
  float level_thr[ASD_MAX_LEVELS];
  int umax[16];

  // ---- ASDNet
  bool weights_loaded = false;
  float* d_w1 = nullptr;        // [32][9] folded
  float* d_bias[7] = {};        // folded BN bias per layer
  float* d_wimg[7] = {};        // MFMA B-operand images, layers 2..7 (index 1..6)
  void* d_wx3[7] = {};          // the same weights split into three bf16 terms (asdnet.hip, split-operand kernels)
  bool match_replay_host = false;  // ASD_MATCH_REPLAY=host (read at asd_ctx_create): matcher claim replay on the host
  void* d_wx2[7] = {};          // layers 1..5 split into two fp16 terms of (weight * wx2_scale[l]) (ASD_ASDNET_MATH=f16x2)
  float wx2_scale[7] = {1, 1, 1, 1, 1, 1, 1};
  int net_pieces_req = 2;       // the form asked for at asd_ctx_create (ASD_ASDNET_MATH); asd_load_weights starts from it every time
  int net_pieces = 2;           // 3 = bf16x3 (six products, exact operands), 2 = fp16x2 (three products, 22-bit operands)
  int net_split = 1;            // ASD_ASDNET_MATH: 1 = split-bf16 kernels where a layer has one, 0 = f32 MFMA everywhere
  float* d_act[2] = {};         // ping-pong NHWC activations
  float* d_part = nullptr;      // split-K partials of the last layer
  int* h_range = nullptr;       // pinned: set by k_l2norm when a descriptor of an asd_describe* call came out non-finite (fp16x2 range)
  std::string calib_note;       // what asd_load_weights' calibration found (empty: nothing to report)
  unsigned* d_calib = nullptr;  // calibration only: per-layer max |activation| as float bits (asdnet_forward_device fills it when set)
  bool pose_chain_kp_flags = false;   // the last pose_chain_enqueue wrote its outlier flags per keypoint (gather form of k_pose_opt)
  bool net_pair = true;         // two-piece form: activations between the layers as the fp16 piece pairs themselves (ASD_ASDNET_PAIR=0: f32 NHWC)
  float* d_act6 = nullptr;      // where the last forward left conv6's output (d_act[0] or d_act[1]; asd_debug_act6)
  int ring_mask = 0;            // ASD_ASDNET_RING: layers on the LDS-image / weight-ring kernels of asdnet_ring.hip (bit 0 conv4, bit 1 conv6, bit 2 conv4+conv5 fused)
  uint8_t* d_patches = nullptr; // [max_patches][1024]
  float* d_desc = nullptr;      // [max_patches][128] descriptors of the last asd_extract / asd_describe
  bool keep_pyramid = false;    // asd_extract_keep_pyramid: every submission keeps a copy of its pyramid for asd_stereo_match
  const uint8_t* d_pyr_view = nullptr;   // that copy of the submission waited for last (null: the shared pyramid of a synchronous extraction)
  AsdDevBuf stereo_scratch;     // asd_stereo_match's own device buffers (it may run beside tracking stages that own `scratch`)
  float* d_desc_last = nullptr; // device descriptors asd_frame_set(desc == NULL) adopts: last asd_extract or last waited submission
  hipEvent_t ev_adopt = nullptr;  // behind the last adoption copy out of d_desc_last (see asd_frame_set / asd_extract_submit)
  bool adopt_pending = false;

  // ---- front-end (state private to frontend.hip)
  struct FrontendState* fe = nullptr;
  int last_n = 0;   // keypoints of the last asd_extract (descriptors in d_desc)

  // ---- frames
  AsdFrameSlot frames[ASD_MAX_FRAMES];

  // ---- matcher scratch (state private to matcher.hip)
  void* matcher = nullptr;

  // ---- BA scratch (lazily grown)
  void* ba = nullptr;

  // ---- per-call device workspace (see AsdDevBuf)
  AsdDevBuf scratch;
  AsdXfer up, down;   // per-call upload / result blocks of the tracking entry points (one copy each way)
  // asd_track_async / asd_track_finish: the completion of an asd_track_* call that returned after enqueueing its work
  bool track_async_armed = false, track_has_pending = false;
  hipEvent_t ev_chain = nullptr;   // end of the last search (+ chain) enqueued by search_and_resolve
  std::function<int()> track_pending;

  // ---- local-mapping scratch (state private to mapping.hip)
  void* mapping = nullptr;

  // ---- vocabulary + BoW scratch (state private to bow.hip)
  void* bow = nullptr;

  // ---- per-layer profiling (asd_profile_enable).  The extraction worker enqueues forwards while the caller enables /
  // reads the profile: every access to the prof_* fields below happens under prof_mutex.
  std::mutex prof_mutex;
  bool prof_on = false;
  hipEvent_t prof_ev[2][9] = {};   // two sets: a forward is enqueued while the previous one is still running
  bool prof_pending[2] = {false, false};
  int prof_pending_n[2] = {0, 0};
  int prof_cur = 0;
  double prof_ms[8] = {};
  int prof_calls[8] = {};
  long long prof_patches[8] = {};

  // ---- timing
  hipEvent_t ev0 = nullptr, ev1 = nullptr, ev2 = nullptr;
  float ms_asdnet = 0, ms_extract = 0, ms_match = 0, ms_ba = 0;

  // the extraction worker thread reports errors too: the message is written under a lock, and asd_last_error hands
  // out a copy that stays put until the next call of asd_last_error on this context
  std::mutex err_mutex;
  std::string err_returned;
  void set_error(const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    std::lock_guard<std::mutex> l(err_mutex);
    err = buf;
  }
  const char* last_error() {
    std::lock_guard<std::mutex> l(err_mutex);
    err_returned = err;
    return err_returned.c_str();
  }
};

inline hipStream_t asd_prep_stream(asd_ctx* ctx) { return ctx->prep_on ? ctx->stream_prep : ctx->stream; }

// frontend.hip
int frontend_alloc(asd_ctx* ctx);
void frontend_free(asd_ctx* ctx);
void frontend_async_shutdown(asd_ctx* ctx);
// true while submissions of asd_extract_submit have not been waited for: the worker thread owns the shared pyramid / score /
// blur buffers and the ASDNet activations then, and the synchronous entry points that use them must refuse to run
bool asd_extractor_busy(asd_ctx* ctx, const char* who);
// true (+ error message) while an asd_track_* call armed with asd_track_async has not been finished: its upload / result blocks,
// the matcher's candidate buffers and the pose solver's staging are in use
bool asd_track_busy(asd_ctx* ctx, const char* who);
// matcher.hip / ba.hip
void matcher_free(asd_ctx* ctx);
void ba_free(asd_ctx* ctx);
void mapping_free(asd_ctx* ctx);
void bow_free(asd_ctx* ctx);
// The claim replay that makes d_src, to run in FRONT of the solver inside its workgroup (k_resolve_pose, ba.hip) instead of as a kernel of
// its own: args = the Resolve2Args of resolve2.h (both translation units include it), kind 0 / 1, nq = its query count, lds = the
// dynamic LDS the replay needs.  pose_chain_fused_ok says whether that form exists for (kind, nq, n_cur).
struct AsdFusedReplay { const void* args; int kind; int nq; size_t lds; };
bool pose_chain_fused_ok(const asd_ctx* ctx, int kind, int nq, int n_cur, size_t lds);
// ba.hip: PoseOptimization enqueued behind device-resident matches (fused tracking chains; see the definition)
// d_pose0 (optional): the start pose on the device (the previous stage's result block), pose7 is then ignored; d_io_dev (optional): a
// second copy of the result block in device memory for the kernels of a following stage (asd_track_frame)
// between (optional, with d_io_dev; DEVICE memory): the work between the two tracking stages as the kernel's tail (asd_between_body)
int pose_chain_enqueue(asd_ctx* ctx, int n_cur, const int* d_src, const float4* d_kp, const float* d_tab, const uint8_t* d_hold,
                       const float* d_own, const double* pose7, const double* K, double* d_io, const double* d_pose0 = nullptr,
                       double* d_io_dev = nullptr, const AsdBetweenArgs* between = nullptr, const AsdFusedReplay* fused = nullptr);
int pose_chain_reserve(asd_ctx* ctx, int n_cur);   // its allocations and kernel attributes, ahead of time (see the definition)
// true when pose_chain_enqueue will take the LDS (gather) form for a frame of n_cur keypoints
inline bool pose_chain_lds_form(const asd_ctx* ctx, int n_cur) { return ctx->cfg.n_levels <= 16 && (size_t)n_cur * 35 + 16 <= 150 * 1024; }
inline size_t pose_chain_io_bytes(int n_cur) { return 64 + ((size_t)n_cur + 7) / 8 * 8 + 64; }   // (+ two 100 MHz stamps behind the edge count: kernel entry, exit)   // pose[7], n_bad, flags (8-B words), edge count
// capi.cpp
void asd_compute_quotas(int nfeatures, float scaleFactor, int nl, int* out);

// asdnet.hip
// the fp16x2 range flag of a finished forward -> status (clears the flag)
inline int asd_range_status(asd_ctx* ctx, int* flag, const char* who) {
  if (!flag || !*flag) return ASD_OK;
  *flag = 0;
  ctx->set_error("%s: an ASDNet activation left the range of the two-piece fp16 operand form (|x| > 4094): the descriptors of this call are "
                 "not valid.  ASD_ASDNET_MATH=bf16x3 (or f32) has no range limit", who);
  return ASD_ERR_RANGE;
}
int asdnet_alloc(asd_ctx* ctx);
void asdnet_free(asd_ctx* ctx);
int asdnet_load_weights(asd_ctx* ctx, const float* const conv_w[7], const float* const bn_mean[7],
                        const float* const bn_var[7], float eps);
// enqueues one forward on `st` (the caller's stream or the extraction worker's); d_act / d_part are one set per context, so
// forwards of one context must be ordered among themselves: asd_extractor_busy() guards the caller-side entry points
// range_flag: pinned host int the L2-norm kernel sets to 1 when a descriptor row is not finite (null = no report)
int asdnet_forward_device(asd_ctx* ctx, const uint8_t* d_patches, int n, float* d_desc, hipStream_t st, int* range_flag = nullptr);
int asdnet_profile_collect(asd_ctx* ctx);  // folds pending layer events into the totals (needs a synced stream)
// asdnet_ring.hip: conv layers with both MFMA operands from LDS (whole-patch images, weights through an LDS-DMA ring); two-piece pair form only
int asdnet_ring_conv(asd_ctx* ctx, int layer, const void* in, void* out, int n, hipStream_t st);
int asdnet_ring_conv45(asd_ctx* ctx, const void* in, void* out, int n, hipStream_t st);
