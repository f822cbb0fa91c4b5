// asd_replay.cpp -- headless replay of an image sequence through the front-end and the frame-to-frame matchers
// (front-end half).  The loop is the reference's Examples/Monocular/kitti.cc:116-155 -- read frame ni,
// hand it to the tracker -- with the tracker reduced to what this library implements: ExtractDesc (read-ahead through
// asd_extract_submit / asd_extract_wait_view: the replay knows its next images), Frame's grid, then between consecutive
// frames either ORBmatcher::SearchForInitialization (no poses given: what Tracking::MonocularInitialization runs,
// Tracking.cc:571-600) or ORBmatcher::SearchByProjection(cur, last) with CALLER-SUPPLIED poses (--poses, KITTI ground-truth
// format) and the last frame's keypoints back-projected at a fixed depth as stand-in map points.  No Tracking state
// machine, no Initializer, no map: those stay the reference's own code.
//
//   asd_replay <sequence_dir> <camera.txt> <weights.bin> [--ext pgm] [--features 2000] [--max-frames N] [--lookahead 2]
//              [--poses poses.txt] [--depth 20] [--stats stats.csv] [--tum trajectory.txt]
//   <sequence_dir>/times.txt + image_0/%06d.<ext> (kitti.cc:56-84; binary PGM frames: no PNG decoder here),
//   camera.txt as cameraconfig/KITTI/*.txt (read_write.cpp:27-60), weights.bin from tools/convert_weights.py.
// Output: one CSV line per frame -- index, timestamp, keypoints, matches to the previous frame, ms waiting for the
// extraction, ms in the matcher, ms for the whole frame -- and, with --tum, System::SaveTrajectoryTUM lines of the supplied poses.
//
//   asd_replay <sequence_dir> <camera.txt> <weights.bin> --chain [--max_step_KF 15] [--warmup W] [--max-frames N] [--lookahead 2]
// runs the per-frame chain the metric is quoted on (bench.py's step, host/track_loop.cpp) over the sequence's images: ExtractDesc with
// read-ahead, grid, TrackWithMotionModel body (SearchByProjection(frame, frame) + PoseOptimization), the local-map selection,
// TrackLocalMap body (isInFrustum + SearchByProjection(frame, points) + PoseOptimization from the first stage's pose), and
// LocalBundleAdjustment IN LINE every --max_step_KF frames (run_vslam_kitti.sh:7: 15).  There is no Tracking / Initializer / map
// logic here (that stays the reference's): the map is a stand-in -- the previous frame's keypoints at 20 m, identity motion
// prediction, the SURVEY 8(d) nominal LocalBA problem with the sequence's intrinsics -- so the numbers are the hot path's cost on
// real frames, not a trajectory.  Prints ONE JSON line with bench.py's keys (metric, value, unit, steps, ms_per_step, config, roofline ...).
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iterator>
#include <string>
#include <vector>

#include "../../include/asd_slam.h"
#include "replay_io.hpp"
#include "ba_nominal.hpp"
#include "track_loop.h"

namespace {

struct Args {
  std::string seq, cam, weights, ext = "pgm", poses, stats, tum;
  int features = 2000, max_frames = -1, lookahead = 2;
  float depth = 20.f;
  bool chain = false;
  int kf_interval = 15, warmup = 0;
};

bool parse(int argc, char** argv, Args& a) {
  if (argc < 4) return false;
  a.seq = argv[1]; a.cam = argv[2]; a.weights = argv[3];
  for (int i = 4; i < argc; ++i) {
    const std::string k = argv[i];
    auto val = [&]() -> const char* { return i + 1 < argc ? argv[++i] : nullptr; };
    const char* v = nullptr;
    if (k == "--ext" && (v = val())) a.ext = v;
    else if (k == "--features" && (v = val())) a.features = atoi(v);
    else if (k == "--max-frames" && (v = val())) a.max_frames = atoi(v);
    else if (k == "--lookahead" && (v = val())) a.lookahead = atoi(v);
    else if (k == "--poses" && (v = val())) a.poses = v;
    else if (k == "--depth" && (v = val())) a.depth = (float)atof(v);
    else if (k == "--stats" && (v = val())) a.stats = v;
    else if (k == "--tum" && (v = val())) a.tum = v;
    else if (k == "--chain") a.chain = true;
    else if (k == "--max_step_KF" && (v = val())) a.kf_interval = atoi(v);
    else if (k == "--warmup" && (v = val())) a.warmup = atoi(v);
    else return false;
  }
  return a.features > 0 && a.lookahead >= 0 && a.lookahead < ASD_EXTRACT_QUEUE && a.kf_interval >= 1 && a.warmup >= 0;
}

// KITTI odometry ground truth: one row-major 3x4 camera-to-world matrix per line -> Tcw (row-major 4x4, f32)
bool read_kitti_poses(const std::string& path, std::vector<std::vector<float>>& Tcw) {
  std::ifstream f(path.c_str());
  if (!f.is_open()) return false;
  double m[12];
  while (f >> m[0]) {
    for (int k = 1; k < 12; ++k) if (!(f >> m[k])) return false;
    std::vector<float> T(16, 0.f);
    for (int r = 0; r < 3; ++r) {
      for (int c = 0; c < 3; ++c) T[r * 4 + c] = (float)m[c * 4 + r];  // Rcw = Rwc^T
      T[r * 4 + 3] = (float)-(m[0 * 4 + r] * m[3] + m[1 * 4 + r] * m[7] + m[2 * 4 + r] * m[11]);
    }
    T[15] = 1.f;
    Tcw.push_back(T);
  }
  return !Tcw.empty();
}

double ms_since(std::chrono::steady_clock::time_point t0) {
  return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
}

// --chain: the metric's per-frame chain over the sequence's frames (all resident in HBM before the timed region, like bench.py)
int run_chain(const Args& a, asd_ctx* ctx, const std::vector<std::string>& files, int nframes, int W, int H, const asd::CamInfo& cam,
              std::vector<uint8_t>& first) {
  auto fail = [&](const char* what) { fprintf(stderr, "%s: %s\n", what, asd_last_error(ctx)); return 1; };
  std::vector<void*> d_frames(nframes, nullptr);
  std::vector<uint8_t> buf;
  for (int t = 0; t < nframes; ++t) {
    int w = W, h = H;
    const std::vector<uint8_t>* src = &first;
    if (t > 0) {
      if (!asd::ReadPGM(files[t], buf, w, h) || w != W || h != H) { fprintf(stderr, "cannot read %s (%dx%d expected)\n", files[t].c_str(), W, H); return 2; }
      src = &buf;
    }
    if (asd_device_alloc(ctx, (uint64_t)W * H, &d_frames[t]) != ASD_OK || asd_memcpy_h2d(ctx, d_frames[t], src->data(), (uint64_t)W * H) != ASD_OK)
      return fail("frame upload");
  }
  const float K32[4] = {(float)cam.fx, (float)cam.fy, (float)cam.cx, (float)cam.cy};
  const double K64[4] = {(double)K32[0], (double)K32[1], (double)K32[2], (double)K32[3]};
  const float T[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
  double pose0[7] = {0.002, -0.001, 0.0015, 1.0, 0.01, -0.02, 0.03};   // the prediction error PoseOptimization has to take out (bench.py)
  {
    const double nq = std::sqrt(pose0[0] * pose0[0] + pose0[1] * pose0[1] + pose0[2] * pose0[2] + pose0[3] * pose0[3]);
    for (int k = 0; k < 4; ++k) pose0[k] /= nq;
  }
  float scale32[ASD_MAX_LEVELS], isg32[ASD_MAX_LEVELS];
  if (asd_get_scale_tables(ctx, scale32, nullptr, nullptr, isg32, nullptr) != ASD_OK) return fail("asd_get_scale_tables");
  double inv_sigma2[8];
  for (int l = 0; l < 8; ++l) inv_sigma2[l] = (double)isg32[l];
  asd::NominalBa ba;
  asd::MakeNominalBa(ba, K64, W, H);
  asd_track_handle* h = asd_track_create(ctx, nframes, d_frames.data(), W, H, K32, T, pose0, inv_sigma2, scale32, &ba.problem, a.kf_interval, a.lookahead);
  if (!h) { fprintf(stderr, "asd_track_create failed\n"); return 1; }
  asd_track_set_drift(h, 0.f, 0.f, 1.f, 0.f, 0.f);   // a real sequence: the stand-in map point projects where its keypoint was
  asd_track_stats st{};
  const int warm = std::min(a.warmup, nframes - 1);
  int rc = ASD_OK;
  if (warm > 0) rc = asd_track_run(h, 0, warm, 1, &st);
  if (rc == ASD_OK) rc = asd_sync(ctx);
  if (rc != ASD_OK) { asd_track_destroy(h); return fail("warm-up"); }
  (void)asd_profile_enable(ctx, 1);
  const int steps = nframes - warm;
  const auto t0 = std::chrono::steady_clock::now();
  rc = asd_track_run(h, warm, steps, 0, &st);
  if (rc == ASD_OK) rc = asd_sync(ctx);
  const double ms = ms_since(t0);
  if (rc != ASD_OK) { asd_track_destroy(h); return fail("asd_track_run"); }
  double l_ms[8] = {}; int32_t l_calls[8] = {}; int64_t l_patches[8] = {};
  for (int l = 0; l < 8; ++l) (void)asd_profile_get(ctx, l, &l_ms[l], &l_calls[l], &l_patches[l]);
  asd_track_destroy(h);
  for (void* p : d_frames) (void)asd_device_free(ctx, p);
  // roofline of the dominant kernel (ASDNet conv2), as bench.py computes it: algorithmic f32 FLOP / hipEvent time, ceiling = dense
  // 16-bit MFMA peak / products per multiply-add of the operand form in use
  const int pieces = asd_asdnet_pieces(ctx);
  const double nprod = pieces == 2 ? 3.0 : 6.0, peak = (asd_asdnet_split_mask(ctx) & 1) ? 2516.6 / nprod : 157.3;
  const double achieved = l_ms[1] > 0 ? 2.0 * 9437184.0 * (double)l_patches[1] / (l_ms[1] * 1e-3) / 1e12 : 0.0;
  double asdnet_ms = 0;
  for (int l = 0; l < 8; ++l) asdnet_ms += l_ms[l];
  asdnet_ms /= std::max(l_calls[1], 1);
  printf("{\"metric\": \"frames/sec end-to-end tracking+LocalBA, KITTI 00 mono @2000 keypoints\", \"value\": %.3f, \"unit\": \"frames/s\", \"n_gpus\": 1, "
         "\"steps\": %d, \"warmup\": %d, \"ms_per_step\": %.6f, \"higher_is_better\": true, \"scaling\": \"weak\", \"vs_baseline\": null, \"dtype\": \"f32\", "
         "\"data\": \"image sequence from disk (%s), stand-in map\", \"config\": {\"workload\": \"asd_replay --chain: %d frames %dx%d, %d features, "
         "extract(E1-E7)+grid+SearchByProjection(frame)+PoseOptimization+isInFrustum+SearchByProjection(map)+PoseOptimization per frame, LocalBA "
         "(%d+%d KF, %d MP, %d edges) every %d frames\", \"keypoints\": %d, \"kf_interval\": %d, \"local_ba\": \"in line (reference order): asd_local_ba at the "
         "keyframe, before the next frame is tracked (Tracking.cc:797 -> LocalMapping.cc:89)\", \"host\": \"C++ (host/asd_replay.cpp + host/track_loop.cpp)\", "
         "\"map\": \"stand-in: previous frame's keypoints at 20 m, identity motion prediction, nominal LocalBA problem\"}, "
         "\"roofline\": {\"bound\": \"mfma\", \"kernel\": \"k_conv_x3 (ASDNet input_norm+conv1+conv2)\", \"achieved\": %.3f, \"peak\": %.3f, \"unit\": \"TFLOP/s\", "
         "\"frac\": %.4f, \"traffic\": null, \"avg_launch_us\": %.2f, \"asdnet_forward_ms\": %.4f}, \"cpu_baseline\": null, "
         "\"last_step\": {\"n_kp\": %d, \"m1\": %d, \"m2\": %d, \"inliers\": %d}}\n",
         1e3 * steps / ms, steps, warm, ms / steps, a.seq.c_str(), nframes, W, H, a.features, 24, 12, ba.problem.n_points, ba.problem.n_edges, a.kf_interval,
         st.n_kp, a.kf_interval, achieved, peak, peak > 0 ? achieved / peak : 0.0, l_calls[1] ? 1e3 * l_ms[1] / l_calls[1] : 0.0, asdnet_ms, st.n_kp, st.m1,
         st.m2, st.inliers);
  return 0;
}

}  // namespace

int main(int argc, char** argv) {
  Args a;
  if (!parse(argc, argv, a)) {
    fprintf(stderr, "usage: %s <sequence_dir> <camera.txt> <weights.bin> [--ext pgm] [--features 2000] [--max-frames N] [--lookahead 0..%d]\n"
                    "       [--poses kitti_poses.txt] [--depth 20] [--stats stats.csv] [--tum trajectory.txt]\n"
                    "       [--chain [--max_step_KF 15] [--warmup W]]   the metric's per-frame chain + in-line LocalBA, one JSON line\n", argv[0], ASD_EXTRACT_QUEUE - 1);
    return 2;
  }
  std::vector<std::string> files;
  std::vector<double> stamps;
  if (!asd::LoadImages(a.seq, files, stamps, a.ext) || files.empty()) { fprintf(stderr, "no times.txt / images under %s\n", a.seq.c_str()); return 2; }
  asd::CamInfo cam;
  if (!asd::ReadCamInfo(a.cam, cam)) { fprintf(stderr, "cannot read camera file %s\n", a.cam.c_str()); return 2; }
  std::vector<std::vector<float>> poses;
  if (!a.poses.empty() && !read_kitti_poses(a.poses, poses)) { fprintf(stderr, "cannot read poses %s\n", a.poses.c_str()); return 2; }
  int nframes = (int)files.size();
  if (a.max_frames > 0 && a.max_frames < nframes) nframes = a.max_frames;
  if (!poses.empty() && (int)poses.size() < nframes) nframes = (int)poses.size();

  // first frame fixes the image size (the reference's configs do the same through ReadImageInfo)
  const int ring = a.lookahead + 1;
  std::vector<std::vector<uint8_t>> img(ring);
  int W = 0, H = 0;
  if (!asd::ReadPGM(files[0], img[0], W, H)) { fprintf(stderr, "cannot read %s (binary PGM expected)\n", files[0].c_str()); return 2; }

  std::vector<uint8_t> wb;
  {
    std::ifstream f(a.weights.c_str(), std::ios::binary);
    wb.assign((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
  }
  const int cout_[7] = {32, 32, 64, 64, 128, 128, 128}, cin_[7] = {1, 32, 32, 64, 64, 128, 128}, ks_[7] = {3, 3, 3, 3, 3, 3, 8};
  const float *cw[7], *bm[7], *bv[7];
  {
    const float* p = reinterpret_cast<const float*>(wb.data());
    size_t need = 0;
    for (int l = 0; l < 7; ++l) need += ((size_t)cout_[l] * cin_[l] * ks_[l] * ks_[l] + 2 * (size_t)cout_[l]) * 4;
    if (wb.size() != need) { fprintf(stderr, "%s: %zu bytes, expected %zu (tools/convert_weights.py layout)\n", a.weights.c_str(), wb.size(), need); return 2; }
    for (int l = 0; l < 7; ++l) { cw[l] = p; p += (size_t)cout_[l] * cin_[l] * ks_[l] * ks_[l]; bm[l] = p; p += cout_[l]; bv[l] = p; p += cout_[l]; }
  }

  asd_config cfg{a.features, 1.2f, 8, 20, 7, W, H, std::max(2 * a.features, 4096), 0};
  asd_ctx* ctx = nullptr;
  if (asd_ctx_create(&cfg, &ctx) != ASD_OK || !ctx) { fprintf(stderr, "asd_ctx_create failed: no usable HIP device (there is no CPU fallback)\n"); return 1; }
  auto fail = [&](const char* what) { fprintf(stderr, "%s: %s\n", what, asd_last_error(ctx)); asd_ctx_destroy(ctx); return 1; };
  if (asd_load_weights(ctx, cw, bm, bv, 1e-5f) != ASD_OK) return fail("asd_load_weights");

  if (a.chain) {
    const int rc = run_chain(a, ctx, files, nframes, W, H, cam, img[0]);
    asd_ctx_destroy(ctx);
    return rc;
  }
  FILE* stats = a.stats.empty() ? stdout : fopen(a.stats.c_str(), "w");
  if (!stats) { perror(a.stats.c_str()); return 2; }
  FILE* tum = a.tum.empty() ? nullptr : fopen(a.tum.c_str(), "w");
  fprintf(stats, "frame,timestamp,keypoints,matches,wait_ms,match_ms,frame_ms\n");

  const float K[4] = {(float)cam.fx, (float)cam.fy, (float)cam.cx, (float)cam.cy};
  const float I4[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
  auto submit = [&](int t) -> int {   // frame t into its ring slot (frame 0 is already loaded), queued for extraction
    std::vector<uint8_t>& buf = img[t % ring];
    int w = 0, h = 0;
    if (t > 0 && (!asd::ReadPGM(files[t], buf, w, h) || w != W || h != H)) { fprintf(stderr, "cannot read %s (%dx%d expected)\n", files[t].c_str(), W, H); return -1; }
    return asd_extract_submit(ctx, buf.data(), 0, W, H, W, 0);
  };
  int submitted = 0;
  for (; submitted < nframes && submitted <= a.lookahead; ++submitted)
    if (submit(submitted) != ASD_OK) return fail("asd_extract_submit");

  std::vector<asd_keypoint> last_kps;
  std::vector<float> Xw, prev_matched;
  std::vector<uint8_t> has;
  std::vector<int32_t> rows, match;
  double sum_frame = 0;
  for (int t = 0; t < nframes; ++t) {
    const auto t0 = std::chrono::steady_clock::now();
    const asd_keypoint* kps = nullptr;
    const float* desc = nullptr;
    int32_t n = 0;
    if (asd_extract_wait_view(ctx, &kps, &desc, &n) != ASD_OK) return fail("asd_extract_wait_view");
    const double wait_ms = ms_since(t0);
    const int cur = t & 1, last = cur ^ 1;
    if (asd_frame_set(ctx, cur, kps, nullptr, n, 0.f, (float)W, 0.f, (float)H) != ASD_OK) return fail("asd_frame_set");
    // the ring slot of frame t is free again: read ahead (the view of frame t stays valid for two further submissions)
    if (submitted < nframes) { if (submit(submitted) != ASD_OK) return fail("asd_extract_submit"); ++submitted; }
    int32_t nmatches = 0;
    const auto tm = std::chrono::steady_clock::now();
    if (t > 0 && !last_kps.empty() && n > 0) {
      const int nl = (int)last_kps.size();
      match.assign(std::max(n, nl), -1);
      if (poses.empty()) {
        // SearchForInitialization(F1 = last, F2 = cur, vbPrevMatched = last keypoints, windowSize 100) (Tracking.cc:585-588)
        prev_matched.resize((size_t)2 * nl);
        for (int i = 0; i < nl; ++i) { prev_matched[2 * i] = last_kps[i].x; prev_matched[2 * i + 1] = last_kps[i].y; }
        if (asd_match_init(ctx, last, cur, prev_matched.data(), 100, 0.9f, 1, match.data(), &nmatches) != ASD_OK) return fail("asd_match_init");
      } else {
        // SearchByProjection(cur, last, th = 15) with the supplied poses; stand-in map points = last keypoints at --depth
        const float* Tl = poses[t - 1].data();
        Xw.resize((size_t)3 * nl); has.assign(nl, 1); rows.resize(nl);
        for (int i = 0; i < nl; ++i) {
          const float xc = (last_kps[i].x - K[2]) / K[0] * a.depth, yc = (last_kps[i].y - K[3]) / K[1] * a.depth, zc = a.depth;
          const float dx = xc - Tl[3], dy = yc - Tl[7], dz = zc - Tl[11];  // Xw = Rcw^T (Xc - tcw)
          for (int c = 0; c < 3; ++c) Xw[3 * i + c] = Tl[0 * 4 + c] * dx + Tl[1 * 4 + c] * dy + Tl[2 * 4 + c] * dz;
          rows[i] = i;
        }
        if (asd_bank_put_from_frame(ctx, last, 0, nl) != ASD_OK) return fail("asd_bank_put_from_frame");
        if (asd_match_project_frame_bank(ctx, cur, last, has.data(), Xw.data(), rows.data(), poses[t].data(), K, 15.f, 1, match.data(), &nmatches,
                                         nullptr) != ASD_OK)
          return fail("asd_match_project_frame_bank");
      }
    }
    const double match_ms = ms_since(tm);
    last_kps.assign(kps, kps + n);
    const double frame_ms = ms_since(t0);
    sum_frame += frame_ms;
    fprintf(stats, "%d,%.6f,%d,%d,%.3f,%.3f,%.3f\n", t, stamps[t], n, nmatches, wait_ms, match_ms, frame_ms);
    if (tum) {
      float twc[3], q[4];
      asd::TcwToTumPose(poses.empty() ? I4 : poses[t].data(), twc, q);
      fprintf(tum, "%s\n", asd::TumLine(stamps[t], twc, q).c_str());
    }
  }
  fprintf(stderr, "asd_replay: %d frames, %.3f ms per frame (%.1f frames/s)\n", nframes, sum_frame / nframes, 1e3 * nframes / sum_frame);
  if (stats != stdout) fclose(stats);
  if (tum) fclose(tum);
  asd_ctx_destroy(ctx);
  return 0;
}
