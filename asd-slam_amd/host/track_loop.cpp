// track_loop.cpp -- the bench's per-frame tracking step as C++ host code over the C ABI.
//
// The reference's host side is C++ (Tracking.cc / LocalMapping.cc call ORBextractor, ORBmatcher, Optimizer); bench.py's
// Python loop adds ~0.25 ms of interpreter and numpy time to a ~2 ms step that a C++ caller would not pay.  This file is
// the same synthetic tracker step as bench.py's track_step -- extract (read-ahead queue) -> grid / frame slot ->
// SearchByProjection(frame) -> PoseOptimization -> isInFrustum + SearchByProjection(local map) -> PoseOptimization ->
// LocalBA every kf_interval frames -- with the same f32 / f64 arithmetic, so both hosts produce identical statistics
// (tests/test_bench_host.py).  It touches the library only through include/asd_slam.h.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <chrono>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <vector>

#include "track_loop.h"

extern "C" {

struct asd_track_handle {
  asd_ctx* ctx;
  std::vector<const uint8_t*> d_frames;  // frames resident in HBM (device pointers), cyclic
  int W, H;
  float K32[4], T[16], scale32[8];
  double K64[4], pose0[7], inv_sigma2[8];
  asd_ba_problem ba;  // pristine problem (host arrays owned by the caller), copied per LocalBA call
  int kf_interval, lookahead;
  bool fused = true;  // asd_track_motion_model / asd_track_local_map (one submission per stage) instead of matcher + solver calls
  // LocalBA in the reference's order (default): Tracking::CreateNewKeyFrame calls LocalMapping::DoMapping IN LINE (Tracking.cc:797 ->
  // LocalMapping.cc:59-113, LocalBundleAdjustment at :89; this fork starts no mapping thread, LocalMapping::Run is dead code), so
  // frame t+1 is tracked against the map the keyframe's LocalBA has already rewritten: asd_local_ba at the keyframe, before the next
  // frame's stages.  async_ba = the library's optional lane (asd_local_ba_submit / _wait): the run goes on beside the next frames,
  // which then read the PRE-BA map -- a different data dependency from the reference, measured only as a variant.
  bool async_ba = false;
  // split-phase stages (asd_track_async / asd_track_finish): while a stage's kernels run, the host does the work that does not
  // depend on its result -- the local-map tables under the motion-model stage, and under the local-map stage the NEXT frame's
  // construction (wait for its extraction, AssignFeaturesToGrid + descriptor adoption, read-ahead submission, descriptor-bank
  // rows, projected points), which the reference's Frame constructor does before tracking that frame
  bool split = true;
  // both stages as ONE submission (asd_track_frame): the between-stage work (outlier drop, pose hand-over, local-map selection) runs on
  // the device, the local map's attributes live in the bank, and the next frame is constructed on the context's second stream
  // (asd_prep_async) beside the stages instead of behind them.  Needs three frame slots (the next frame's grid is written while the
  // motion-model stage may still read the last frame's) and two alternating bank regions.
  // OFF by default since round 5: the reference selects its local map between the stages (Tracking::UpdateLocalMap, Tracking.cc:730), which
  // needs the host there -- the split-phase two-call form above is the one Tracking can bind; this one is a measured variant
  bool chain = false;
  int tables_for = -1;      // the frame whose candidate tables (Xw2 ...) prepare_frame has built and stored in the attribute bank
  int map_copies = 2;       // the stand-in local map: every point of the last frame and map_copies - 1 displaced copies of it (asd_track_set_map_copies)
  const asd_do_mapping_inputs* dm = nullptr;   // the batched per-keyframe stage in front of LocalBA (asd_track_set_do_mapping)
  std::vector<int32_t> dm_matches, dm_nmatch, dm_best, dm_distinct;
  std::vector<float> dm_x3d, dm_bdist;
  std::vector<uint8_t> dm_ok;
  double dm_ms = 0.0;
  long dm_calls = 0;
  // stereo mode (asd_track_set_stereo)
  asd_ctx* ctx_r = nullptr;
  std::vector<const uint8_t*> d_frames_r;
  float mb = 0.f, mbf = 0.f;
  std::vector<float> u_right, depth_r;
  std::vector<asd_keypoint> kps_r;
  std::vector<float> desc_sync_r;
  int32_t prep_stereo = -1;   // stereo matches of the prepared frame
  int frames_on_host = 0;   // the frame pointers are (pinned) host memory: asd_extract_submit(device_resident = 0), the image crosses PCIe per frame
  int bank_base = 0;        // first bank row of the map the prepared frame is tracked against
  std::vector<int32_t> last_cand, cand_rows, sel_rows;
  asd_track_frame_args fa;
  int32_t f_n1 = 0, f_inl1 = 0, f_n2 = 0, f_inl2 = 0;
  double f_pose[7], f_pose1[7];
  float drift_cx = 620.5f, drift_cy = 188.0f, drift_z = 1.003f, drift_dx = 3.0f, drift_dy = 0.2f;   // asd_track_set_drift
  int stop_after = -1;      // no read-ahead beyond this frame (-1 = unbounded)
  int prep_t = -1;          // the frame prepare_frame() has made ready (its grid sits in slot `slot`), -1 = none
  const asd_keypoint* prep_kps = nullptr;
  int32_t prep_n = 0;
  int32_t c2_n2 = 0, c2_ninl = 0;   // outputs of an outstanding local-map stage (stable addresses)
  double c2_pose[7];
  std::vector<uint8_t> outl2;
  bool ba_out = false;      // a submission is outstanding
  long ba_step = -1;        // the step that submitted it
  asd_ba_problem ba_p;
  asd_ba_result ba_r;
  // state
  int slot = 0;
  std::deque<int> pending;  // frame indices of outstanding submissions, oldest first
  bool have_last = false;
  std::vector<asd_keypoint> last_kps;  // copies: the previous frame's keypoints (16 B each)
  int last_slot = 0;
  // work buffers
  std::vector<asd_keypoint> kps;
  std::vector<float> desc_sync;
  std::vector<float> uv, Xw, Xw2, nrm, dist, maxd, mind, proj, vc;
  std::vector<uint8_t> has, in_view, occ, outl;
  std::vector<float> cur_Xw, Xs, ns, mind_s, maxd_s;   // the selected local points' tables (UpdateLocalMap stand-in)
  std::vector<int32_t> sel;
  std::vector<uint8_t> keep, in_frame;
  std::vector<double> ba_obs;   // the keyframe's observations (nominal + deterministic per-keyframe displacement)
  float T1[16];
  std::vector<int32_t> rows, m1, m2, level;
  std::vector<double> Xd, obs, info;
  std::vector<double> ba_poses, ba_points, ba_chi2;
  std::vector<uint8_t> ba_dpos, ba_out1;
  double wait_ms = 0.0, ba_ms = 0.0;  // ASD_TIMING: time blocked on the extractor / inside LocalBA
  double ba_prep_ms = 0.0;            // ... and putting the keyframe's LocalBA problem together on the host (the stand-in for Optimizer.cc:415-600's graph assembly)
  double seg_ms[8] = {};              // frame_set, submit, bank+M1, pose1, frustum, M2, pose2, host glue
  double kern_ms[4] = {};             // device time of M1, pose1, M2, pose2 (asd_last_stage_ms)
  long steps = 0;
};

asd_track_handle* asd_track_create(asd_ctx* ctx, int32_t n_frames, const void* const* d_frames, int32_t W, int32_t H, const float* K32,
                                   const float* T, const double* pose0, const double* inv_sigma2, const float* scale32,
                                   const asd_ba_problem* ba, int32_t kf_interval, int32_t lookahead) {
  if (!ctx || n_frames < 1 || !d_frames || !K32 || !T || !pose0 || !inv_sigma2 || !scale32 || !ba || kf_interval < 1 || lookahead < 0 ||
      lookahead > ASD_EXTRACT_QUEUE)   // a step waits for its own frame before it submits: never more than `lookahead` outstanding
    return nullptr;
  asd_track_handle* h = new asd_track_handle();
  h->ctx = ctx;
  for (int i = 0; i < n_frames; ++i) h->d_frames.push_back(static_cast<const uint8_t*>(d_frames[i]));
  h->W = W; h->H = H;
  for (int i = 0; i < 4; ++i) { h->K32[i] = K32[i]; h->K64[i] = (double)K32[i]; }
  memcpy(h->T, T, sizeof h->T);
  memcpy(h->pose0, pose0, sizeof h->pose0);
  memcpy(h->inv_sigma2, inv_sigma2, sizeof h->inv_sigma2);
  memcpy(h->scale32, scale32, sizeof h->scale32);
  h->ba = *ba;
  h->kf_interval = kf_interval; h->lookahead = lookahead;
  h->kps.resize(1 << 13);
  h->desc_sync.resize((size_t)(1 << 13) * 128);
  return h;
}

void asd_track_set_fused(asd_track_handle* h, int32_t on) { if (h) h->fused = on != 0; }
void asd_track_set_async_ba(asd_track_handle* h, int32_t on) { if (h) h->async_ba = on != 0; }
void asd_track_set_split(asd_track_handle* h, int32_t on) { if (h) h->split = on != 0; }
void asd_track_set_chain(asd_track_handle* h, int32_t on) { if (h) h->chain = on != 0; }
void asd_track_set_map_copies(asd_track_handle* h, int32_t copies) { if (h && copies >= 1 && copies <= 16) h->map_copies = copies; }
// hands every read-ahead submission of this handle back to the library (another handle of the same context can then start)
int asd_track_drain(asd_track_handle* h) {
  if (!h) return ASD_ERR_INVALID;
  int rc = ASD_OK;
  while (!h->pending.empty()) {
    const asd_keypoint* k; const float* d; int32_t n;
    const int r = asd_extract_wait_view(h->ctx, &k, &d, &n);
    if (r != ASD_OK) rc = r;
    if (h->ctx_r) { const int r2 = asd_extract_wait_view(h->ctx_r, &k, &d, &n); if (r2 != ASD_OK) rc = r2; }
    h->pending.pop_front();
  }
  h->prep_t = -1;
  h->have_last = false;
  return rc;
}
void asd_track_set_frames_on_host(asd_track_handle* h, int32_t on) { if (h) h->frames_on_host = on != 0; }
int asd_track_set_stereo(asd_track_handle* h, asd_ctx* ctx_right, const void* const* d_frames_right, float mb, float mbf) {
  if (!h || !ctx_right || !d_frames_right || !(mb > 0) || !(mbf > 0) || !h->pending.empty()) return ASD_ERR_INVALID;
  int rc;
  if ((rc = asd_extract_keep_pyramid(h->ctx, 1)) != ASD_OK || (rc = asd_extract_keep_pyramid(ctx_right, 1)) != ASD_OK) return rc;
  h->ctx_r = ctx_right;
  h->d_frames_r.clear();
  for (size_t i = 0; i < h->d_frames.size(); ++i) h->d_frames_r.push_back(static_cast<const uint8_t*>(d_frames_right[i]));
  h->mb = mb; h->mbf = mbf;
  return ASD_OK;
}
void asd_track_set_do_mapping(asd_track_handle* h, const asd_do_mapping_inputs* in, int32_t n_cur) {
  if (!h) return;
  h->dm = in;
  if (!in) return;
  h->dm_matches.assign((size_t)in->n_nb * n_cur, -1); h->dm_nmatch.assign(std::max(in->n_nb, 1), 0);
  h->dm_x3d.assign((size_t)in->n_nb * n_cur * 3, 0.f); h->dm_ok.assign((size_t)in->n_nb * n_cur, 0);
  h->dm_best.assign(std::max(in->n_fuse_total, 1), -1); h->dm_bdist.assign(std::max(in->n_fuse_total, 1), 0.f);
  h->dm_distinct.assign(std::max(in->n_sets, 1), 0);
}
void asd_track_get_do_mapping_times(const asd_track_handle* h, double* ms, int64_t* calls) {
  if (!h) return;
  if (ms) *ms = h->dm_ms;
  if (calls) *calls = h->dm_calls;
}
void asd_track_get_times(const asd_track_handle* h, double* ba_ms, double* extract_wait_ms, int64_t* steps) {
  if (!h) return;
  if (ba_ms) *ba_ms = h->ba_ms;
  if (extract_wait_ms) *extract_wait_ms = h->wait_ms;
  if (steps) *steps = h->steps;
}
void asd_track_set_drift(asd_track_handle* h, float cx, float cy, float z, float dx, float dy) {
  if (h) { h->drift_cx = cx; h->drift_cy = cy; h->drift_z = z; h->drift_dx = dx; h->drift_dy = dy; }
}

// collect the outstanding LocalBA; its chi2 goes into *st only when the step that submitted it is the one st describes
static int collect_ba(asd_track_handle* h, asd_track_stats* st, long st_step) {
  if (!h->ba_out) return ASD_OK;
  const auto b0 = std::chrono::steady_clock::now();
  const int rc = asd_local_ba_wait(h->ctx);
  h->ba_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - b0).count();
  h->ba_out = false;
  if (rc != ASD_OK) return rc;
  if (st && st_step == h->ba_step) { st->ba_chi2 = h->ba_r.chi2_second; st->has_ba = 1; }
  return ASD_OK;
}

void asd_track_destroy(asd_track_handle* h) {
  if (!h) return;
  while (!h->pending.empty()) {  // drain: the library still owns those submissions
    const asd_keypoint* k; const float* d; int32_t n;
    (void)asd_extract_wait_view(h->ctx, &k, &d, &n);
    if (h->ctx_r) (void)asd_extract_wait_view(h->ctx_r, &k, &d, &n);
    h->pending.pop_front();
  }
  (void)collect_ba(h, nullptr, -1);
  if (getenv("ASD_TIMING") && h->steps)
  {
    fprintf(stderr, "[track_loop] steps %ld  extract wait %.3f ms/step  local BA %.3f ms/step (+ %.3f ms/step putting its problem together on the host)\n", h->steps,
            h->wait_ms / h->steps, h->ba_ms / h->steps, h->ba_prep_ms / h->steps);
    static const char* nm[8] = {"frame_set", "submit", "bank+M1", "pose1", "frustum", "M2", "pose2", "host glue"};
    for (int i = 0; i < 8; ++i) fprintf(stderr, "[track_loop]   %-10s %.3f ms/step\n", nm[i], h->seg_ms[i] / h->steps);
    fprintf(stderr, "[track_loop]   device time: M1 %.3f pose1 %.3f M2 %.3f pose2 %.3f ms/step\n", h->kern_ms[0] / h->steps, h->kern_ms[1] / h->steps,
            h->kern_ms[2] / h->steps, h->kern_ms[3] / h->steps);
  }
  delete h;
}

// ---- split-phase form of the fused step --------------------------------------------------------------------------------
// Frame construction for frame t: take its extraction (read-ahead result or a synchronous one), AssignFeaturesToGrid + adopt the
// descriptors, keep the read-ahead queue full, and -- when there is a previous frame -- the inputs of the motion-model stage
// (projected points of the previous frame, its descriptors as bank rows).  Everything here is enqueued on the context's stream
// behind whatever stage is still in flight and reads none of its results.
// the stand-in local map's attribute tables: candidate c * nl + i = the last frame's point i displaced by c * 0.02 m in every coordinate
// (c = 0: the point itself), with its normal, distance and the distance range of the point's pyramid level -- numpy f32 arithmetic, mirrored
// term by term in bench.py's track_step (two copies there)
static void build_candidate_tables(asd_track_handle* h, int nl) {
  const std::vector<asd_keypoint>& lk = h->last_kps;
  const int nc = h->map_copies * nl;
  h->Xw2.resize((size_t)3 * nc); h->nrm.resize((size_t)3 * nc); h->dist.resize(nc); h->maxd.resize(nc); h->mind.resize(nc);
  for (int c = 0; c < h->map_copies; ++c)
    for (int i = 0; i < nl; ++i)
      for (int k = 0; k < 3; ++k) h->Xw2[3 * ((size_t)c * nl + i) + k] = c == 0 ? h->Xw[3 * i + k] : h->Xw[3 * i + k] + (c == 1 ? 0.02f : 0.02f * (float)c);
  for (int i = 0; i < nc; ++i) {
    const float* P = &h->Xw2[3 * (size_t)i];
    const float nn = std::sqrt((P[0] * P[0] + P[1] * P[1]) + P[2] * P[2]);  // numpy: sqrt(add.reduce(x * x)), float32
    for (int k = 0; k < 3; ++k) h->nrm[3 * (size_t)i + k] = P[k] / nn;
    h->dist[i] = nn;
    const int lv = lk[i % nl].octave;
    h->maxd[i] = nn * h->scale32[lv];
    h->mind[i] = h->maxd[i] / h->scale32[7];
  }
}

static int prepare_frame(asd_track_handle* h, int t, const std::vector<int>& next) {
  asd_ctx* ctx = h->ctx;
  asd_ctx* ctxr = h->ctx_r;   // stereo mode: the right image's extractor, its queue in lockstep with the left one's
  const int nf = (int)h->d_frames.size();
  int rc;
  auto tp = std::chrono::steady_clock::now();
  auto seg = [&](int i) {
    const auto now = std::chrono::steady_clock::now();
    h->seg_ms[i] += std::chrono::duration<double, std::milli>(now - tp).count();
    tp = now;
  };
  auto drop_pending = [&]() -> int {
    while (!h->pending.empty()) {
      const asd_keypoint* k; const float* d; int32_t nn;
      int r = asd_extract_wait_view(ctx, &k, &d, &nn);
      if (r == ASD_OK && ctxr) r = asd_extract_wait_view(ctxr, &k, &d, &nn);
      if (r != ASD_OK) return r;
      h->pending.pop_front();
    }
    return ASD_OK;
  };
  const asd_keypoint *kps = nullptr, *kps_r = nullptr;
  int32_t n = 0, n_r = 0;
  if (!h->pending.empty() && h->pending.front() == t) {
    const float* d = nullptr;
    const auto w0 = std::chrono::steady_clock::now();
    if ((rc = asd_extract_wait_view(ctx, &kps, &d, &n)) != ASD_OK) return rc;
    if (ctxr && (rc = asd_extract_wait_view(ctxr, &kps_r, &d, &n_r)) != ASD_OK) return rc;
    h->wait_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - w0).count();
    h->pending.pop_front();
  } else {
    if ((rc = drop_pending()) != ASD_OK) return rc;
    if ((rc = (h->frames_on_host ? asd_extract : asd_extract_device)(ctx, h->d_frames[t % nf], h->W, h->H, h->W, 0, h->kps.data(), h->desc_sync.data(), &n)) != ASD_OK) return rc;
    kps = h->kps.data();
    if (ctxr) {
      h->kps_r.resize(1 << 13); h->desc_sync_r.resize((size_t)(1 << 13) * 128);
      if ((rc = asd_extract_device(ctxr, h->d_frames_r[t % nf], h->W, h->H, h->W, 0, h->kps_r.data(), h->desc_sync_r.data(), &n_r)) != ASD_OK) return rc;
      kps_r = h->kps_r.data();
    }
  }
  // split-phase forms (two calls or one submission): the next frame is constructed on the context's second stream BESIDE the stage in flight
  // (asd_prep_async), so it writes a third frame slot and the other bank region -- nothing the stage in flight reads
  const bool chain = h->fused && h->split;
  if (chain) h->slot = (h->slot + 1) % 3; else h->slot ^= 1;
  seg(7);
  if ((rc = asd_frame_set(ctx, h->slot, kps, nullptr, n, 0.f, (float)h->W, 0.f, (float)h->H)) != ASD_OK) return rc;
  seg(0);
  if (ctxr) {
    // Frame::ComputeStereoMatches (Frame.cc:360-535): the right frame in a slot of the left context (its descriptors device to device
    // from the right extractor's buffer), then the association over both submissions' own pyramids
    const int slot_r = 3 + h->slot;
    if ((rc = asd_frame_set_from_ctx(ctx, slot_r, kps_r, n_r, 0.f, (float)h->W, 0.f, (float)h->H, ctxr)) != ASD_OK) return rc;
    h->u_right.resize(std::max(n, 1)); h->depth_r.resize(std::max(n, 1));
    int32_t nm = 0;
    if ((rc = asd_stereo_match(ctx, ctxr, h->slot, slot_r, h->mb, h->mbf, h->u_right.data(), h->depth_r.data(), &nm)) != ASD_OK) return rc;
    h->prep_stereo = nm;
    seg(4);
  }
  if (h->lookahead > 0) {
    bool same = h->pending.size() <= next.size();
    for (size_t i = 0; same && i < h->pending.size(); ++i) same = h->pending[i] == next[i];
    if (!same && (rc = drop_pending()) != ASD_OK) return rc;
    for (size_t i = h->pending.size(); i < next.size(); ++i) {
      if ((rc = asd_extract_submit(ctx, h->d_frames[next[i] % nf], h->frames_on_host ? 0 : 1, h->W, h->H, h->W, 0)) != ASD_OK) return rc;
      if (ctxr && (rc = asd_extract_submit(ctxr, h->d_frames_r[next[i] % nf], 1, h->W, h->H, h->W, 0)) != ASD_OK) return rc;
      h->pending.push_back(next[i]);
    }
  }
  seg(1);
  if (h->have_last && chain) {
    // the map this frame is tracked against, written beside the previous frame's stages: the other bank region (rows the stages in
    // flight do not read) takes the last frame's points -- descriptors device to device, attributes from the host -- and a displaced copy
    const std::vector<asd_keypoint>& lk = h->last_kps;
    const int nl = (int)lk.size();
    const float fx = h->K32[0], fy = h->K32[1], cx = h->K32[2], cy = h->K32[3];
    const float z = h->drift_z, c3 = (float)((double)h->drift_dx * (double)(h->drift_z == 1.003f ? 1.003 : (double)h->drift_z)),
                c02 = (float)((double)(h->drift_dy == 0.2f ? 0.2 : (double)h->drift_dy) * (double)(h->drift_z == 1.003f ? 1.003 : (double)h->drift_z)), depth = 20.0f;
    h->Xw.resize((size_t)3 * nl);
    for (int i = 0; i < nl; ++i) {
      const float u = (lk[i].x - h->drift_cx) * z + h->drift_cx - c3;
      const float v = (lk[i].y - h->drift_cy) * z + h->drift_cy - c02;
      h->Xw[3 * i + 0] = (u - cx) / fx * depth;
      h->Xw[3 * i + 1] = (v - cy) / fy * depth;
      h->Xw[3 * i + 2] = depth;
    }
    const int n2p = h->map_copies * nl;
    build_candidate_tables(h, nl);
    seg(7);
    h->bank_base = h->bank_base ? 0 : 65536;
    const int base = h->bank_base;
    if (n2p > 65536) return ASD_ERR_CAPACITY;
    for (int c = 0; c < h->map_copies; ++c)
      if ((rc = asd_bank_put_from_frame(ctx, h->last_slot, base + c * nl, nl)) != ASD_OK) return rc;
    if ((rc = asd_mpbank_put(ctx, base, n2p, h->Xw2.data(), h->nrm.data(), h->mind.data(), h->maxd.data())) != ASD_OK) return rc;
    h->rows.resize(nl); h->last_cand.resize(nl); h->cand_rows.resize(n2p);
    for (int i = 0; i < nl; ++i) { h->rows[i] = base + i; h->last_cand[i] = i; }
    for (int i = 0; i < n2p; ++i) h->cand_rows[i] = base + i;
    h->tables_for = t;
    seg(2);
  } else if (h->have_last) {
    const std::vector<asd_keypoint>& lk = h->last_kps;
    const int nl = (int)lk.size();
    const float fx = h->K32[0], fy = h->K32[1], cx = h->K32[2], cy = h->K32[3];
    // (default drift: exactly bench.py's predicted_uv -- z = 1.003f, c3 = (float)(3 * 1.003), c02 = (float)(0.2 * 1.003))
    const float z = h->drift_z, c3 = (float)((double)h->drift_dx * (double)(h->drift_z == 1.003f ? 1.003 : (double)h->drift_z)),
                c02 = (float)((double)(h->drift_dy == 0.2f ? 0.2 : (double)h->drift_dy) * (double)(h->drift_z == 1.003f ? 1.003 : (double)h->drift_z)), depth = 20.0f;
    h->Xw.resize((size_t)3 * nl);
    for (int i = 0; i < nl; ++i) {
      const float u = (lk[i].x - h->drift_cx) * z + h->drift_cx - c3;
      const float v = (lk[i].y - h->drift_cy) * z + h->drift_cy - c02;
      h->Xw[3 * i + 0] = (u - cx) / fx * depth;
      h->Xw[3 * i + 1] = (v - cy) / fy * depth;
      h->Xw[3 * i + 2] = depth;
    }
    seg(7);
    for (int c = 0; c < h->map_copies; ++c)
      if ((rc = asd_bank_put_from_frame(ctx, h->last_slot, c * nl, nl)) != ASD_OK) return rc;
    h->rows.resize((size_t)h->map_copies * nl);
    for (int i = 0; i < h->map_copies * nl; ++i) h->rows[i] = i;
    // the candidates' attributes (MapPoint::mWorldPos, mNormalVector, mfMin/MaxDistance: they exist before the frame is tracked) into the
    // attribute bank, rows as the descriptors': the local-map stage then names its points by row (asd_track_local_points_rows)
    build_candidate_tables(h, nl);
    if ((rc = asd_mpbank_put(ctx, 0, h->map_copies * nl, h->Xw2.data(), h->nrm.data(), h->mind.data(), h->maxd.data())) != ASD_OK) return rc;
    h->tables_for = t;
    seg(2);
  }
  h->prep_t = t; h->prep_kps = kps; h->prep_n = n;
  return ASD_OK;
}

static int submit_ba(asd_track_handle* h, asd_track_stats* st, int t);

// What Tracking does between its two stages (Tracking.cc:695-714, 725-726, 811-823): drop the matches PoseOptimization marked as
// outliers, take the optimised pose as the frame's pose, and put the local map together from what is not in the frame already --
// here: the candidates (the last frame's points + their displaced copies) that no kept match refers to.  Fills keep / occ / cur_Xw,
// sel and the selected tables, T1; returns the number of selected points.  Mirrors bench.py's track_step line by line.
static int select_local_points(asd_track_handle* h, int n, int nl, const uint8_t* outl1, const double* pose1, bool gather = true) {
  h->keep.assign(n, 0); h->occ.assign(n, 0);
  h->in_frame.assign((size_t)h->map_copies * nl, 0);
  h->cur_Xw.assign((size_t)3 * n, 0.f);
  int nmatch = 0;
  for (int j = 0; j < n; ++j) {
    const int i = h->m1[j];
    nmatch += i >= 0;
    const int src = i >= 0 ? i : 0;
    for (int k = 0; k < 3; ++k) h->cur_Xw[3 * j + k] = h->Xw[3 * src + k];
    if (i >= 0 && !outl1[j]) { h->keep[j] = 1; h->occ[j] = 1; }
    if (i >= 0) h->in_frame[i] = 1;   // kept: the frame holds it (:811-823); outlier: mnLastFrameSeen = this frame (:705-707) -- neither is searched again
  }
  if (nmatch >= 3) (void)asd_pose7_to_tcw(pose1, h->T1);
  else memcpy(h->T1, h->T, sizeof h->T1);
  h->sel.clear();
  for (int i = 0; i < h->map_copies * nl; ++i) if (!h->in_frame[i]) h->sel.push_back(i);
  const int ns = (int)h->sel.size();
  if (!gather) return ns;   // (the stage names its points by bank row: no tables to put together)
  h->Xs.resize((size_t)3 * ns); h->ns.resize((size_t)3 * ns); h->mind_s.resize(ns); h->maxd_s.resize(ns);
  for (int q = 0; q < ns; ++q) {
    const int i = h->sel[q];
    for (int k = 0; k < 3; ++k) { h->Xs[3 * q + k] = h->Xw2[3 * i + k]; h->ns[3 * q + k] = h->nrm[3 * i + k]; }
    h->mind_s[q] = h->mind[i]; h->maxd_s[q] = h->maxd[i];
  }
  return ns;
}

// The map points frame t holds when it becomes mLastFrame (Tracking.cc:296-349): a keypoint keeps the map point of a kept motion-model
// match or of a local-map match, unless the local-map stage's PoseOptimization marked it as an outlier (:345-349).  These flags are
// the next frame's has_mp[] -- derived from this frame's RESULTS, after its last stage has finished.  (The stand-in map gives every
// keypoint of a frame a position, see prepare_frame; which of them the next frame projects is decided here.)
static void final_matches_to_has(asd_track_handle* h, int n, bool tracked, const uint8_t* keep1, const int32_t* m2, const uint8_t* outl2) {
  h->has.resize(n);
  if (!tracked) { std::fill(h->has.begin(), h->has.end(), (uint8_t)1); return; }   // bootstrap frame: as after initialisation, every keypoint holds a point
  for (int j = 0; j < n; ++j) h->has[j] = ((keep1[j] || m2[j] >= 0) && !outl2[j]) ? 1 : 0;
}

static int track_step_split(asd_track_handle* h, int t, bool do_ba, const std::vector<int>& next, asd_track_stats* st) {
  asd_ctx* ctx = h->ctx;
  int rc;
  auto tp = std::chrono::steady_clock::now();
  auto seg = [&](int i) {
    const auto now = std::chrono::steady_clock::now();
    h->seg_ms[i] += std::chrono::duration<double, std::milli>(now - tp).count();
    tp = now;
  };
  if (h->prep_t != t && (rc = prepare_frame(h, t, next)) != ASD_OK) return rc;
  tp = std::chrono::steady_clock::now();
  const asd_keypoint* kps = h->prep_kps;
  const int32_t n = h->prep_n;
  const int cur = h->slot;
  const int base = h->bank_base;     // the bank region that holds THIS frame's map (the next frame's goes to the other one)
  h->prep_t = -1;
  memset(st, 0, sizeof *st);
  st->n_kp = n;
  if (h->ctx_r) { st->stereo_matched = h->prep_stereo; st->has_stereo = 1; }
  const bool had_last = h->have_last;
  if (had_last) {
    const std::vector<asd_keypoint>& lk = h->last_kps;
    const int nl = (int)lk.size();
    // ---- Tracking::TrackWithMotionModel's numeric body, enqueued (Tracking.cc:664-723)
    h->m1.assign(n, -1);
    int32_t n1 = 0, ninl1 = 0;
    double pose[7];
    memcpy(pose, h->pose0, sizeof pose);
    h->outl.resize(n);
    if ((rc = asd_track_async(ctx)) != ASD_OK) return rc;
    if ((rc = asd_track_motion_model_bank(ctx, cur, h->last_slot, h->has.data(), h->Xw.data(), h->rows.data(), h->T, h->K32, 15.0f, 1, nullptr,
                                          pose, h->m1.data(), &n1, h->outl.data(), &ninl1)) != ASD_OK)
      return rc;
    seg(2);
    // ---- under it: the local map's tables (the last frame's points plus a jittered copy; nothing here needs the stage's result)
    if (h->tables_for != t) build_candidate_tables(h, nl);   // (prepare_frame built them and stored them in the attribute bank)
    seg(7);
    if ((rc = asd_track_finish(ctx)) != ASD_OK) return rc;
    st->m1 = n1; st->has_m1 = 1;
    seg(3);
    // ---- between the stages: outlier matches dropped, the optimised pose becomes the frame's pose, the local map is put together
    const int nsel = select_local_points(h, n, nl, h->outl.data(), pose, h->tables_for != t);
    // ---- Tracking::TrackLocalMap's numeric body, enqueued (Tracking.cc:725-736, 803-851), from the motion-model stage's pose
    h->m2.assign(n, -1);
    h->outl2.resize(n);
    memcpy(h->c2_pose, pose, sizeof h->c2_pose);
    h->c2_n2 = 0; h->c2_ninl = 0;
    seg(7);
    if ((rc = asd_track_async(ctx)) != ASD_OK) return rc;
    // the selected points by ROW of the banks (their attributes were stored when the frame was prepared): 4 bytes per point go up
    h->sel_rows.resize(nsel);
    for (int q = 0; q < nsel; ++q) h->sel_rows[q] = base + h->sel[q];
    if (nsel > 0 && h->tables_for == t) {
      rc = asd_track_local_points_rows(ctx, cur, nsel, h->sel_rows.data(), h->T1, h->K32, 0.5f, h->occ.data(), h->cur_Xw.data(), 1.0f, 0.8f, nullptr, h->c2_pose,
                                       h->m2.data(), &h->c2_n2, h->outl2.data(), &h->c2_ninl);
    }
    else
      rc = asd_track_local_points_bank(ctx, cur, nsel, h->Xs.data(), h->ns.data(), h->mind_s.data(), h->maxd_s.data(), h->sel_rows.data(), h->T1,
                                       h->K32, 0.5f, h->occ.data(), h->cur_Xw.data(), 1.0f, 0.8f, nullptr, h->c2_pose, h->m2.data(), &h->c2_n2,
                                       h->outl2.data(), &h->c2_ninl);
    if (rc != ASD_OK) return rc;
    seg(5);
  }
  // ---- under the local-map stage: this frame becomes the last frame, LocalBA goes to its lane, the next frame is constructed
  h->last_kps.assign(kps, kps + n);
  h->last_slot = cur;
  h->have_last = true;
  if (do_ba && h->async_ba && (rc = submit_ba(h, st, t)) != ASD_OK) return rc;   // (in line: after the stage has finished, below)
  seg(7);
  if (!next.empty() && next[0] == t + 1) {
    std::vector<int> after(next.begin() + 1, next.end());
    after.push_back(next.back() + 1);   // the queue of frame t+1: t+2 .. t+1+lookahead (asd_track_run trims it at the end of a run)
    if ((int)after.size() > h->lookahead) after.resize(h->lookahead);
    if (h->stop_after >= 0) while (!after.empty() && after.back() > h->stop_after) after.pop_back();
    // the next frame's construction (grid, descriptor adoption, the map's bank rows) on the context's second stream, beside the local-map stage:
    // enqueued behind it on the tracking stream, its five copy kernels (16-60 us each beside ASDNet) sat between this frame's solver and the
    // next frame's projection -- 150 us of every frame (profiles/r05_tracking_timeline_before.txt)
    if ((rc = asd_prep_async(ctx, 1)) != ASD_OK) { if (had_last) (void)asd_track_finish(ctx); return rc; }
    rc = prepare_frame(h, t + 1, after);
    const int rc2 = asd_prep_async(ctx, 0);
    if (rc != ASD_OK || rc2 != ASD_OK) { if (had_last) (void)asd_track_finish(ctx); return rc != ASD_OK ? rc : rc2; }
    tp = std::chrono::steady_clock::now();
  }
  if (had_last) {
    if ((rc = asd_track_finish(ctx)) != ASD_OK) return rc;
    st->m2 = h->c2_n2; st->has_m2 = 1;
    int nedge = 0;
    for (int j = 0; j < n; ++j) nedge += h->keep[j] || h->m2[j] >= 0;
    if (nedge >= 3) { st->inliers = h->c2_ninl; st->has_inliers = 1; } else std::fill(h->outl2.begin(), h->outl2.end(), (uint8_t)0);
    seg(6);
  }
  final_matches_to_has(h, n, had_last, h->keep.data(), h->m2.data(), h->outl2.data());
  if (do_ba && !h->async_ba && (rc = submit_ba(h, st, t)) != ASD_OK) return rc;   // asd_local_ba uses the context's stream: no stage outstanding
  ++h->steps;
  return ASD_OK;
}

// One frame with both stages as one submission (asd_track_frame) and the next frame constructed beside them (asd_prep_async).
static int track_step_chain(asd_track_handle* h, int t, bool do_ba, const std::vector<int>& next, asd_track_stats* st) {
  asd_ctx* ctx = h->ctx;
  int rc;
  auto tp = std::chrono::steady_clock::now();
  auto seg = [&](int i) {
    const auto now = std::chrono::steady_clock::now();
    h->seg_ms[i] += std::chrono::duration<double, std::milli>(now - tp).count();
    tp = now;
  };
  if (h->prep_t != t && (rc = prepare_frame(h, t, next)) != ASD_OK) return rc;
  tp = std::chrono::steady_clock::now();
  const asd_keypoint* kps = h->prep_kps;
  const int32_t n = h->prep_n;
  const int cur = h->slot;
  h->prep_t = -1;
  memset(st, 0, sizeof *st);
  st->n_kp = n;
  if (h->ctx_r) { st->stereo_matched = h->prep_stereo; st->has_stereo = 1; }
  const bool had_last = h->have_last;
  if (had_last) {
    h->m1.assign(n, -1); h->m2.assign(n, -1);
    h->outl.assign(n, 0); h->outl2.assign(n, 0);
    memcpy(h->f_pose, h->pose0, sizeof h->f_pose);
    asd_track_frame_args& a = h->fa;
    memset(&a, 0, sizeof a);
    a.slot_cur = cur; a.slot_last = h->last_slot;
    a.has_mp = h->has.data(); a.Xw_last = h->Xw.data(); a.last_rows = h->rows.data(); a.last_cand = h->last_cand.data();
    a.Tcw = h->T; a.th = 15.0f; a.check_orientation = 1;
    a.n_cand = (int32_t)h->cand_rows.size(); a.cand_rows = h->cand_rows.data();
    a.viewing_cos_limit = 0.5f; a.th_local = 1.0f; a.nn_ratio = 0.8f; a.K = h->K32;
    a.pose7 = h->f_pose; a.pose1 = h->f_pose1;
    a.match1 = h->m1.data(); a.n_matches1 = &h->f_n1; a.outlier1 = h->outl.data(); a.n_inliers1 = &h->f_inl1;
    a.match2 = h->m2.data(); a.n_matches2 = &h->f_n2; a.outlier2 = h->outl2.data(); a.n_inliers2 = &h->f_inl2;
    if ((rc = asd_track_async(ctx)) != ASD_OK) return rc;
    if ((rc = asd_track_frame(ctx, &a)) != ASD_OK) return rc;
    seg(2);
  }
  // ---- beside the stages: this frame becomes the last frame, the next one is constructed on the second stream
  h->last_kps.assign(kps, kps + n);
  h->last_slot = cur;
  h->have_last = true;
  if (do_ba && h->async_ba && (rc = submit_ba(h, st, t)) != ASD_OK) return rc;
  seg(7);
  if (!next.empty() && next[0] == t + 1) {
    std::vector<int> after(next.begin() + 1, next.end());
    after.push_back(next.back() + 1);
    if ((int)after.size() > h->lookahead) after.resize(h->lookahead);
    if (h->stop_after >= 0) while (!after.empty() && after.back() > h->stop_after) after.pop_back();
    // (on an error the stage submitted above is finished before the error is returned: its completion holds pointers into this handle's
    // vectors, and the context stays busy until asd_track_finish)
    if ((rc = asd_prep_async(ctx, 1)) != ASD_OK) { if (had_last) (void)asd_track_finish(ctx); return rc; }
    rc = prepare_frame(h, t + 1, after);
    const int rc2 = asd_prep_async(ctx, 0);
    if (rc != ASD_OK || rc2 != ASD_OK) { if (had_last) (void)asd_track_finish(ctx); return rc != ASD_OK ? rc : rc2; }
    tp = std::chrono::steady_clock::now();
  }
  if (had_last) {
    if ((rc = asd_track_finish(ctx)) != ASD_OK) return rc;
    st->m1 = h->f_n1; st->has_m1 = 1;
    st->m2 = h->f_n2; st->has_m2 = 1;
    int nedge = 0;
    for (int j = 0; j < n; ++j) nedge += (h->m1[j] >= 0 && !h->outl[j]) || h->m2[j] >= 0;
    if (nedge >= 3) { st->inliers = h->f_inl2; st->has_inliers = 1; } else std::fill(h->outl2.begin(), h->outl2.end(), (uint8_t)0);
    h->keep.resize(n);
    for (int j = 0; j < n; ++j) h->keep[j] = h->m1[j] >= 0 && !h->outl[j];
    seg(6);
  }
  final_matches_to_has(h, n, had_last, h->keep.data(), h->m2.data(), h->outl2.data());
  if (do_ba && !h->async_ba && (rc = submit_ba(h, st, t)) != ASD_OK) return rc;
  ++h->steps;
  return ASD_OK;
}

static int track_step(asd_track_handle* h, int t, bool do_ba, const std::vector<int>& next, asd_track_stats* st) {
  asd_ctx* ctx = h->ctx;
  const int nf = (int)h->d_frames.size();
  int rc;
  auto tp = std::chrono::steady_clock::now();
  auto seg = [&](int i) {  // charge the time since the previous mark to segment i
    const auto now = std::chrono::steady_clock::now();
    h->seg_ms[i] += std::chrono::duration<double, std::milli>(now - tp).count();
    tp = now;
  };
  // ---- ExtractDesc: take the read-ahead result if this frame is the oldest submission, else extract now
  const asd_keypoint* kps = nullptr;
  int32_t n = 0;
  if (!h->pending.empty() && h->pending.front() == t) {
    const float* d = nullptr;
    const auto w0 = std::chrono::steady_clock::now();
    if ((rc = asd_extract_wait_view(ctx, &kps, &d, &n)) != ASD_OK) return rc;
    h->wait_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - w0).count();
    h->pending.pop_front();
  } else {
    while (!h->pending.empty()) {
      const asd_keypoint* k; const float* d; int32_t nn;
      if ((rc = asd_extract_wait_view(ctx, &k, &d, &nn)) != ASD_OK) return rc;
      h->pending.pop_front();
    }
    if ((rc = (h->frames_on_host ? asd_extract : asd_extract_device)(ctx, h->d_frames[t % nf], h->W, h->H, h->W, 0, h->kps.data(), h->desc_sync.data(), &n)) != ASD_OK) return rc;
    kps = h->kps.data();
  }
  // ---- Frame::AssignFeaturesToGrid + adopt the device-resident descriptors
  h->slot ^= 1;
  const int cur = h->slot;
  seg(7);
  if ((rc = asd_frame_set(ctx, cur, kps, nullptr, n, 0.f, (float)h->W, 0.f, (float)h->H)) != ASD_OK) return rc;
  seg(0);
  // ---- read ahead
  if (h->lookahead > 0) {
    bool same = h->pending.size() <= next.size();
    for (size_t i = 0; same && i < h->pending.size(); ++i) same = h->pending[i] == next[i];
    if (!same) {
      while (!h->pending.empty()) {
        const asd_keypoint* k; const float* d; int32_t nn;
        if ((rc = asd_extract_wait_view(ctx, &k, &d, &nn)) != ASD_OK) return rc;
        h->pending.pop_front();
      }
    }
    for (size_t i = h->pending.size(); i < next.size(); ++i) {
      if ((rc = asd_extract_submit(ctx, h->d_frames[next[i] % nf], h->frames_on_host ? 0 : 1, h->W, h->H, h->W, 0)) != ASD_OK) return rc;
      h->pending.push_back(next[i]);
    }
  }
  seg(1);
  memset(st, 0, sizeof *st);
  st->n_kp = n;
  if (h->have_last) {
    const std::vector<asd_keypoint>& lk = h->last_kps;
    const int nl = (int)lk.size();
    const float fx = h->K32[0], fy = h->K32[1], cx = h->K32[2], cy = h->K32[3];
    // predicted_uv + backproject_identity (depth 20), float32 like the numpy expressions
    // (default drift: exactly bench.py's predicted_uv -- z = 1.003f, c3 = (float)(3 * 1.003), c02 = (float)(0.2 * 1.003))
    const float z = h->drift_z, c3 = (float)((double)h->drift_dx * (double)(h->drift_z == 1.003f ? 1.003 : (double)h->drift_z)),
                c02 = (float)((double)(h->drift_dy == 0.2f ? 0.2 : (double)h->drift_dy) * (double)(h->drift_z == 1.003f ? 1.003 : (double)h->drift_z)), depth = 20.0f;
    h->Xw.resize((size_t)3 * nl);
    for (int i = 0; i < nl; ++i) {
      const float u = (lk[i].x - h->drift_cx) * z + h->drift_cx - c3;
      const float v = (lk[i].y - h->drift_cy) * z + h->drift_cy - c02;
      h->Xw[3 * i + 0] = (u - cx) / fx * depth;
      h->Xw[3 * i + 1] = (v - cy) / fy * depth;
      h->Xw[3 * i + 2] = depth;
    }
    if ((int)h->has.size() != nl) h->has.assign(nl, 1);   // (first tracked frame: every keypoint of the bootstrap frame holds a map point)
    // map point descriptors: rows 0..nl-1 = the last frame's descriptors, rows nl..2nl-1 the same again
    seg(7);
    for (int c = 0; c < h->map_copies; ++c)
      if ((rc = asd_bank_put_from_frame(ctx, h->last_slot, c * nl, nl)) != ASD_OK) return rc;
    h->rows.resize((size_t)h->map_copies * nl);
    for (int i = 0; i < h->map_copies * nl; ++i) h->rows[i] = i;
    h->m1.assign(n, -1);
    int32_t n1 = 0;
    auto dev = [&](int i, const char* stage) { float ms = 0.f; if (asd_last_stage_ms(ctx, stage, &ms) == ASD_OK) h->kern_ms[i] += ms; };
    // PoseOptimization over the keypoints `sel` from `pose_io` (in: start, out: optimum); outl_k[j] per KEYPOINT
    auto pose_opt = [&](const std::vector<int>& sel, auto point_of, double* pose_io, uint8_t* outl_k, int32_t* ninl) -> int {
      const int m = (int)sel.size();
      h->Xd.resize((size_t)3 * m); h->obs.resize((size_t)2 * m); h->info.resize(m); h->outl.resize(m);
      for (int q = 0; q < m; ++q) {
        const int j = sel[q];
        const float* P = point_of(j);
        for (int k = 0; k < 3; ++k) h->Xd[3 * q + k] = (double)P[k];
        h->obs[2 * q] = (double)kps[j].x; h->obs[2 * q + 1] = (double)kps[j].y;
        h->info[q] = h->inv_sigma2[kps[j].octave];
      }
      const int r = asd_pose_optimize(ctx, pose_io, m, h->Xd.data(), h->obs.data(), h->info.data(), h->K64, h->outl.data(), ninl);
      if (r == ASD_OK && outl_k) for (int q = 0; q < m; ++q) outl_k[sel[q]] = h->outl[q];
      return r;
    };
    // attributes of the candidate map points (they exist before the frame is tracked): the last frame's points plus a displaced copy
    build_candidate_tables(h, nl);
    std::vector<int> sel;
    std::vector<uint8_t> outl1(n, 0);
    double pose1[7];
    memcpy(pose1, h->pose0, sizeof pose1);
    seg(7);
    if (h->fused) {
      // Tracking::TrackWithMotionModel's numeric body in one submission (Tracking.cc:664-723)
      int32_t ninl = 0;
      if ((rc = asd_track_motion_model_bank(ctx, cur, h->last_slot, h->has.data(), h->Xw.data(), h->rows.data(), h->T, h->K32, 15.0f, 1, nullptr,
                                            pose1, h->m1.data(), &n1, outl1.data(), &ninl)) != ASD_OK)
        return rc;
      st->m1 = n1; st->has_m1 = 1;
      seg(2);
      dev(0, "match");
      seg(3);
    } else {
      if ((rc = asd_match_project_frame_bank(ctx, cur, h->last_slot, h->has.data(), h->Xw.data(), h->rows.data(), h->T, h->K32, 15.0f, 1,
                                             h->m1.data(), &n1, nullptr)) != ASD_OK)
        return rc;
      st->m1 = n1; st->has_m1 = 1;
      seg(2);
      dev(0, "match");
      for (int j = 0; j < n; ++j) if (h->m1[j] >= 0) sel.push_back(j);
      if (sel.size() >= 3) {
        int32_t ninl = 0;
        if ((rc = pose_opt(sel, [&](int j) { return &h->Xw[3 * h->m1[j]]; }, pose1, outl1.data(), &ninl)) != ASD_OK) return rc;
      }
      dev(1, "ba");
      seg(3);
    }
    // between the stages: outlier matches dropped, the optimised pose becomes the frame's pose, the local map is put together
    const int nsel = select_local_points(h, n, nl, outl1.data(), pose1);
    h->in_view.resize(nsel); h->proj.resize((size_t)2 * nsel); h->level.resize(nsel); h->vc.resize(nsel);
    seg(7);
    if (!h->fused) {   // fused: isInFrustum / PredictScale / search windows are made on the device inside asd_track_local_points
      if ((rc = asd_frustum(ctx, cur, nsel, h->Xs.data(), h->ns.data(), h->mind_s.data(), h->maxd_s.data(), h->T1, h->K32, 0.5f, h->in_view.data(),
                            h->proj.data(), h->level.data(), h->vc.data())) != ASD_OK)
        return rc;
    }
    seg(4);
    h->m2.assign(n, -1);
    int32_t n2 = 0;
    if (h->fused) {
      // Tracking::TrackLocalMap's numeric body (SearchLocalPoints: frustum loop + matcher, then PoseOptimization; Tracking.cc:725-736, 803-851)
      int nedge = 0;
      double pose[7];
      memcpy(pose, pose1, sizeof pose);
      int32_t ninl = 0;
      h->outl.resize(n);
      if ((rc = asd_track_local_points_bank(ctx, cur, nsel, h->Xs.data(), h->ns.data(), h->mind_s.data(), h->maxd_s.data(), h->sel.data(), h->T1,
                                            h->K32, 0.5f, h->occ.data(), h->cur_Xw.data(), 1.0f, 0.8f, nullptr, pose, h->m2.data(), &n2,
                                            h->outl.data(), &ninl)) != ASD_OK)
        return rc;
      st->m2 = n2; st->has_m2 = 1;
      dev(2, "match");
      seg(5);
      for (int j = 0; j < n; ++j) nedge += h->keep[j] || h->m2[j] >= 0;
      if (nedge >= 3) { st->inliers = ninl; st->has_inliers = 1; h->outl2.assign(h->outl.begin(), h->outl.begin() + n); } else h->outl2.assign(n, 0);
      seg(6);
    } else {
      if ((rc = asd_match_project_points_bank(ctx, cur, nsel, h->in_view.data(), h->proj.data(), h->level.data(), h->vc.data(), h->sel.data(),
                                              h->occ.data(), 1.0f, 0.8f, h->m2.data(), &n2, nullptr)) != ASD_OK)
        return rc;
      st->m2 = n2; st->has_m2 = 1;
      dev(2, "match");
      seg(5);
      sel.clear();
      for (int j = 0; j < n; ++j) if (h->keep[j] || h->m2[j] >= 0) sel.push_back(j);
      h->outl2.assign(n, 0);
      if (sel.size() >= 3) {
        int32_t ninl = 0;
        double pose[7];
        memcpy(pose, pose1, sizeof pose);
        if ((rc = pose_opt(sel, [&](int j) { return h->keep[j] ? &h->Xw[3 * h->m1[j]] : &h->Xw2[3 * h->sel[std::max(h->m2[j], 0)]]; }, pose, h->outl2.data(), &ninl)) != ASD_OK)
          return rc;
        st->inliers = ninl; st->has_inliers = 1;
      }
      dev(3, "ba");
      seg(6);
    }
  }
  final_matches_to_has(h, n, h->have_last, h->keep.data(), h->m2.data(), h->outl2.data());
  if (do_ba && (rc = submit_ba(h, st, t)) != ASD_OK) return rc;
  seg(7);
  h->last_kps.assign(kps, kps + n);
  h->last_slot = cur;
  h->have_last = true;
  ++h->steps;
  return ASD_OK;
}


// LocalBA at a keyframe: in line (the reference's order) or on the optional lane (collected later).  The problem of keyframe number
// t / kf_interval = the nominal problem with every observation displaced by a deterministic +-0.1 px (bench.py,
// ba_problem_for_keyframe: same integer hash, same double arithmetic): a real map changes between keyframes.
// LocalMapping::DoMapping in front of its LocalBundleAdjustment call (LocalMapping.cc:66-88): new map points against the covisible
// keyframes, fusion with the neighbours, distinctive descriptors of the touched points -- three batched submissions
static int do_mapping_stage(asd_track_handle* h) {
  const asd_do_mapping_inputs& D = *h->dm;
  asd_ctx* ctx = h->ctx;
  const auto t0 = std::chrono::steady_clock::now();
  int rc;
  if (D.n_nb > 0 && (rc = asd_create_map_points_batch(ctx, D.slot_cur, D.has_mp_cur, D.Tcw_cur, D.K_cur, D.n_nb, D.nb, h->dm_matches.data(), h->dm_nmatch.data(),
                                                      h->dm_x3d.data(), h->dm_ok.data())) != ASD_OK)
    return rc;
  if (D.n_fuse_calls > 0 && (rc = asd_fuse_search_batch(ctx, D.n_fuse_calls, D.fuse_calls, D.n_fuse_total, D.valid, D.Xw, D.normal, D.min_dist, D.max_dist, nullptr,
                                                        D.desc_rows, D.th, h->dm_best.data(), h->dm_bdist.data())) != ASD_OK)
    return rc;
  if (D.n_sets > 0 && (rc = asd_distinctive_descriptor_batch(ctx, D.n_sets, D.set_start, D.set_desc, h->dm_distinct.data())) != ASD_OK) return rc;
  h->dm_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  ++h->dm_calls;
  return ASD_OK;
}

static int submit_ba(asd_track_handle* h, asd_track_stats* st, int t) {
  asd_ctx* ctx = h->ctx;
  int rc;
  if ((rc = collect_ba(h, nullptr, -1)) != ASD_OK) return rc;   // the previous keyframe's run (its buffers are reused below)
  const auto p0 = std::chrono::steady_clock::now();
  const asd_ba_problem& B = h->ba;
  h->ba_poses.assign(B.poses, B.poses + (size_t)7 * B.n_poses);
  h->ba_points.assign(B.points, B.points + (size_t)3 * B.n_points);
  h->ba_chi2.assign(B.n_edges, 0.0); h->ba_dpos.assign(B.n_edges, 0); h->ba_out1.assign(B.n_edges, 0);
  h->ba_obs.resize((size_t)2 * B.n_edges);
  const uint32_t kf = (uint32_t)(t / h->kf_interval);
  for (uint32_t i = 0; i < (uint32_t)(2 * B.n_edges); ++i) {
    uint32_t x = i * 2654435761u + kf * 40503u + 12345u;
    x ^= x >> 15;
    x *= 2246822519u;
    x ^= x >> 13;
    const double u = (double)((x >> 8) & 0xFFFFu);
    h->ba_obs[i] = B.e_obs[i] + (u / 65536.0 - 0.5) * 0.2;
  }
  h->ba_p = B;
  h->ba_p.poses = h->ba_poses.data(); h->ba_p.points = h->ba_points.data(); h->ba_p.e_obs = h->ba_obs.data();
  memset(&h->ba_r, 0, sizeof h->ba_r);
  h->ba_r.edge_chi2 = h->ba_chi2.data(); h->ba_r.edge_depth_pos = h->ba_dpos.data(); h->ba_r.edge_outlier1 = h->ba_out1.data();
  h->ba_prep_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - p0).count();
  // No further ASDNet forward is ENQUEUED while the per-keyframe stage and LocalBA run in line (asd_extract_hold; forwards already on the
  // device finish, front halves continue): the reference does nothing else during DoMapping either, and the tracking thread is blocked for as
  // long as it takes.  Round 4 (extractor and tracking thread balanced): no gain.  Round 5 (the tracking thread is the longer side, the
  // extractor has ~12 % of slack to catch up with): 2.3-2.5 instead of 2.75-2.85 ms per LocalBA, tracking 0.64 instead of 0.62 ms per frame:
  // +2.2 % at the driver's K = 20 (median of eight alternating pairs on one box: 1097 against 1073 frames/s), +0.7 % at K = 300.
  struct ExtractHold {
    asd_ctx *a, *b;
    ExtractHold(asd_ctx* a_, asd_ctx* b_, bool on) : a(on ? a_ : nullptr), b(on ? b_ : nullptr) { if (a) (void)asd_extract_hold(a, 1); if (b) (void)asd_extract_hold(b, 1); }
    ~ExtractHold() { if (a) (void)asd_extract_hold(a, 0); if (b) (void)asd_extract_hold(b, 0); }
  } hold(ctx, h->ctx_r, !h->async_ba);
  if (h->dm && !h->async_ba && (rc = do_mapping_stage(h)) != ASD_OK) return rc;   // (in line only: with the lane a tracking stage is outstanding here)
  const auto b0 = std::chrono::steady_clock::now();
  if (h->async_ba) {
    if ((rc = asd_local_ba_submit(ctx, &h->ba_p, &h->ba_r)) != ASD_OK) return rc;
    h->ba_out = true;
    h->ba_step = h->steps;
  } else {
    rc = asd_local_ba(ctx, &h->ba_p, &h->ba_r);
    if (rc != ASD_OK) return rc;
    st->ba_chi2 = h->ba_r.chi2_second; st->has_ba = 1;
  }
  h->ba_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - b0).count();
  return ASD_OK;
}

// n frames t0 .. t0+n-1; frames after the last one are read ahead only when the replay continues (prefetch_beyond)
int asd_track_run(asd_track_handle* h, int32_t t0, int32_t n, int32_t prefetch_beyond, asd_track_stats* stats) {
  if (!h || n < 0 || !stats) return ASD_ERR_INVALID;
  std::vector<int> next;
  for (int i = 0; i < n; ++i) {
    const int t = t0 + i;
    next.clear();
    for (int k = 1; k <= h->lookahead; ++k)
      if (i + k < n || prefetch_beyond) next.push_back(t + k);
    h->stop_after = prefetch_beyond ? -1 : t0 + n - 1;   // the last frame a read-ahead submission may be made for
    const bool split = h->fused && h->split;
    if (!split && h->prep_t >= 0) { h->prep_t = -1; }    // (a frame prepared by the split step is simply prepared again)
    const bool ba_now = t % h->kf_interval == h->kf_interval - 1;
    const int rc = split ? (h->chain ? track_step_chain(h, t, ba_now, next, stats) : track_step_split(h, t, ba_now, next, stats))
                         : track_step(h, t, ba_now, next, stats);
    if (rc != ASD_OK) return rc;
  }
  return collect_ba(h, stats, h->steps - 1);   // the run ends with its LocalBA finished (and reported if the last step started it)
}

}  // extern "C"
