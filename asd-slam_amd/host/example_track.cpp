// example_track.cpp -- two-frame tracking with the host-side mirror classes (asd_adapters.hpp):
// what Tracking::GrabImageMonocular + TrackWithMotionModel do with the hot path (Tracking.cc:102-110,
// 664-723), minus the map.  Usage: example_track weights.bin frame0.raw frame1.raw W H
// weights.bin = 7 x (conv_w, bn_mean, bn_var) f32 blobs in ASDNet.py:334-356 order.
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iterator>
#include <utility>
#include <vector>

#include "asd_adapters.hpp"

static std::vector<uint8_t> slurp(const char* path) {
  std::ifstream f(path, std::ios::binary);
  return std::vector<uint8_t>((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
}

int main(int argc, char** argv) {
  if (argc < 6) { fprintf(stderr, "usage: %s weights.bin frame0.raw frame1.raw W H\n", argv[0]); return 2; }
  const int W = atoi(argv[4]), H = atoi(argv[5]);
  const int cout_[7] = {32, 32, 64, 64, 128, 128, 128}, cin_[7] = {1, 32, 32, 64, 64, 128, 128}, k_[7] = {3, 3, 3, 3, 3, 3, 8};
  std::vector<uint8_t> wb = slurp(argv[1]);
  const float* p = reinterpret_cast<const float*>(wb.data());
  const float *cw[7], *bm[7], *bv[7];
  for (int l = 0; l < 7; ++l) {
    cw[l] = p; p += (size_t)cout_[l] * cin_[l] * k_[l] * k_[l];
    bm[l] = p; p += cout_[l];
    bv[l] = p; p += cout_[l];
  }
  if ((const uint8_t*)p != wb.data() + wb.size()) { fprintf(stderr, "weights.bin has the wrong size\n"); return 2; }
  try {
    asd::Context ctx(2000, 1.2f, 8, 20, 7, W, H);
    asd::ORBextractor extractor(ctx, cw, bm, bv);
    const asd::Camera K{718.856f, 718.856f, 607.1928f, 185.2157f};
    asd::FrameView F[2];
    for (int t = 0; t < 2; ++t) {
      std::vector<uint8_t> img = slurp(argv[2 + t]);
      if ((int)img.size() != W * H) { fprintf(stderr, "frame %d has the wrong size\n", t); return 2; }
      F[t].slot = t;
      F[t].mnMinX = 0; F[t].mnMaxX = (float)W; F[t].mnMinY = 0; F[t].mnMaxY = (float)H;
      if (extractor.ExtractDesc(img.data(), W, H, W, F[t].mvKeysUn, F[t].mDescriptors) < 0) { fprintf(stderr, "%s\n", ctx.error()); return 1; }
      if (asd::FrameAssignFeaturesToGrid(ctx, F[t]) != ASD_OK) { fprintf(stderr, "%s\n", ctx.error()); return 1; }
    }
    // every keypoint of frame 0 carries a map point 20 m in front of the camera (planar stand-in for the map)
    std::vector<asd::MapPointView> mps(F[0].N());
    std::vector<const asd::MapPointView*> pmp(F[0].N());
    F[0].mvpMapPoints.resize(F[0].N());
    for (int i = 0; i < F[0].N(); ++i) {
      const float z = 20.f, u = (F[0].mvKeysUn[i].x - 620.5f) * 1.003f + 620.5f - 3.009f, v = (F[0].mvKeysUn[i].y - 188.f) * 1.003f + 188.f - 0.2006f;
      mps[i].Xw[0] = (u - K.cx) / K.fx * z; mps[i].Xw[1] = (v - K.cy) / K.fy * z; mps[i].Xw[2] = z;
      mps[i].normal[0] = 0; mps[i].normal[1] = 0; mps[i].normal[2] = 1;
      mps[i].mfMinDistance = 1; mps[i].mfMaxDistance = 200;
      mps[i].descriptor = &F[0].mDescriptors[(size_t)i * ASD_DESC_DIM];
      pmp[i] = &mps[i];
      F[0].mvpMapPoints[i] = i;
    }
    asd::ORBmatcher matcher(ctx, 0.8f, true);                                   // Tracking.cc:666
    const int nmatches = matcher.SearchByProjection(F[1], F[0], pmp, K, 15.f);  // Tracking.cc:679
    std::vector<const asd::MapPointView*> cur_pts(F[1].N(), nullptr);
    for (int j = 0; j < F[1].N(); ++j) if (F[1].mvpMapPoints[j] >= 0) cur_pts[j] = &mps[F[1].mvpMapPoints[j]];
    const int ninl = asd::Optimizer::PoseOptimization(ctx, &F[1], cur_pts, K, extractor.GetInverseScaleSigmaSquares());  // :693
    // the same stage as ONE submission (asd::Tracking::TrackWithMotionModel = asd_track_motion_model): same matches, same pose bits
    int chain_ok = 0;
    {
      asd::FrameView G = F[1];
      G.mvpMapPoints.clear(); G.mvbOutlier.clear();
      for (int k = 0; k < 16; ++k) G.mTcw[k] = (k % 5 == 0) ? 1.f : 0.f;      // the pose the separate calls started from
      int nm2 = 0, ninl2 = 0;
      const bool ok = asd::Tracking::TrackWithMotionModel(ctx, G, F[0], pmp, K, true, &nm2, &ninl2);
      bool same = ok && nm2 == nmatches && ninl2 == ninl;
      for (int k = 0; k < 16 && same; ++k) same = G.mTcw[k] == F[1].mTcw[k];
      for (int j = 0; j < G.N() && same; ++j)   // the frame keeps its inlier matches (Tracking.cc:704-719)
        same = G.mvpMapPoints[j] == (F[1].mvbOutlier[j] ? -1 : F[1].mvpMapPoints[j]);
      chain_ok = same;
      // ... and TrackLocalMap's body on the points the frame did not take, with LocalBundleAdjustment running on its lane meanwhile
      std::vector<asd::MapPointView> local;
      std::vector<int32_t> ids;
      std::vector<uint8_t> taken(mps.size(), 0);
      std::vector<const asd::MapPointView*> curp(G.N(), nullptr);
      for (int j = 0; j < G.N(); ++j) if (G.mvpMapPoints[j] >= 0) { taken[G.mvpMapPoints[j]] = 1; curp[j] = &mps[G.mvpMapPoints[j]]; }
      for (size_t i = 0; i < mps.size(); ++i) if (!taken[i]) { local.push_back(mps[i]); ids.push_back((int32_t)i); }
      // a small bundle: two keyframes (the first fixed) observing the frame's inlier points
      std::vector<double> poses = {0, 0, 0, 1, 0, 0, 0, 0, 0, 0, 1, 0, 0, 0}, pts, eobs, einfo;
      std::vector<uint8_t> fixed = {1, 0};
      std::vector<int32_t> ept, eps;
      asd_tcw_to_pose7(G.mTcw, &poses[7]);
      for (int j = 0; j < G.N() && pts.size() < 3 * 400; ++j)
        if (curp[j]) {
          const int m = G.mvpMapPoints[j], pid = (int)pts.size() / 3;
          int j0 = -1;
          for (int q = 0; q < F[0].N(); ++q) if (F[0].mvpMapPoints[q] == m) { j0 = q; break; }
          if (j0 < 0) continue;
          for (int k = 0; k < 3; ++k) pts.push_back(mps[m].Xw[k]);
          ept.push_back(pid); eps.push_back(0); eobs.push_back(F[0].mvKeysUn[j0].x); eobs.push_back(F[0].mvKeysUn[j0].y); einfo.push_back(1.0);
          ept.push_back(pid); eps.push_back(1); eobs.push_back(G.mvKeysUn[j].x); eobs.push_back(G.mvKeysUn[j].y); einfo.push_back(1.0);
        }
      asd_ba_problem pr{};
      pr.n_poses = 2; pr.n_points = (int32_t)pts.size() / 3; pr.n_edges = (int32_t)ept.size();
      pr.poses = poses.data(); pr.fixed = fixed.data(); pr.points = pts.data(); pr.e_point = ept.data(); pr.e_pose = eps.data();
      pr.e_obs = eobs.data(); pr.e_info = einfo.data(); pr.K[0] = K.fx; pr.K[1] = K.fy; pr.K[2] = K.cx; pr.K[3] = K.cy; pr.its_first = 5; pr.its_second = 10;
      std::vector<double> chi2(ept.size()); std::vector<uint8_t> dpos(ept.size()), out1(ept.size());
      asd_ba_result rs{};
      rs.edge_chi2 = chi2.data(); rs.edge_depth_pos = dpos.data(); rs.edge_outlier1 = out1.data();
      const bool ba_sub = pr.n_points >= 10 && asd::LocalMapping::LocalBundleAdjustmentSubmit(ctx, &pr, &rs) == ASD_OK;
      const int ninl3 = asd::Tracking::TrackLocalMap(ctx, G, local, ids, curp, K);
      const bool ba_ok = ba_sub && asd::LocalMapping::LocalBundleAdjustmentWait(ctx) == ASD_OK && rs.iters_second > 0;
      printf("chain: same=%d local-map inliers=%d lane BA ok=%d (%d points, chi2 %.3f)\n", (int)same, ninl3, (int)ba_ok, pr.n_points, rs.chi2_second);
      chain_ok = chain_ok && ninl3 > 50 && ba_ok;
    }
    // what LocalMapping::SearchInNeighbors does with the same points once frame 1 became a keyframe (LocalMapping.cc:557-636)
    asd::ORBmatcher fuser(ctx);
    std::vector<int32_t> bestIdx;
    const int nfused = fuser.Fuse(F[1], mps, std::vector<uint8_t>(mps.size(), 1), K, bestIdx);
    // Frame::ComputeBoW + SearchByBoW(KeyFrame*, Frame&, ...) (Tracking::TrackReferenceKeyFrame, Tracking.cc:607-620) with a
    // small random vocabulary standing in for the missing vocabulary file: k = 8, L = 2, node ids breadth-first
    int nbow = 0;
    {
      const int k = 8, L = 2, n_nodes = 1 + k + k * k;
      std::vector<int32_t> start(n_nodes + 1, 0), kids, word(n_nodes, -1);
      std::vector<double> weight(n_nodes, 0.0);
      std::vector<float> d((size_t)n_nodes * 128, 0.f);
      for (int i = 0; i < n_nodes; ++i) {
        if (i < 1 + k) for (int c = 0; c < k; ++c) kids.push_back(1 + i * k + c);
        start[i + 1] = (int32_t)kids.size();
      }
      uint32_t seed = 12345u;
      for (int i = 1; i < n_nodes; ++i) {
        if (i > k) { word[i] = i - 1 - k; weight[i] = 1.0 + (i % 7); }
        // children scatter around a real descriptor of frame 0 so that the words are populated
        const float* base = &F[0].mDescriptors[(size_t)((i * 37) % F[0].N()) * ASD_DESC_DIM];
        for (int q = 0; q < 128; ++q) { seed = seed * 1664525u + 1013904223u; d[(size_t)i * 128 + q] = base[q] + ((seed >> 8) & 0xffff) * (0.2f / 65536.f) - 0.1f; }
      }
      if (asd_voc_load(ctx.get(), n_nodes, k, L, 0, 0, start.data(), kids.data(), weight.data(), word.data(), d.data()) != ASD_OK) { fprintf(stderr, "%s\n", ctx.error()); return 1; }
      std::vector<std::pair<int32_t, double>> bow0, bow1;
      asd::ORBmatcher::FeatVec fv0, fv1;
      if (asd::ComputeBoW(ctx, F[0], bow0, fv0, 1) != ASD_OK || asd::ComputeBoW(ctx, F[1], bow1, fv1, 1) != ASD_OK) { fprintf(stderr, "%s\n", ctx.error()); return 1; }
      asd::ORBmatcher bowmatcher(ctx, 0.7f, true);                               // Tracking.cc:616
      std::vector<int32_t> vpMapPointMatches;
      nbow = bowmatcher.SearchByBoW(F[0], fv0, F[1], fv1, vpMapPointMatches);
    }
    printf("kp0=%d kp1=%d matches=%d inliers=%d fused=%d bow=%d t=(%.4f %.4f %.4f)\n", F[0].N(), F[1].N(), nmatches, ninl, nfused, nbow,
           F[1].mTcw[3], F[1].mTcw[7], F[1].mTcw[11]);
    return (nmatches > 100 && ninl > 50 && nfused > 0 && nbow > 0 && chain_ok) ? 0 : 1;
  } catch (const std::exception& e) {
    fprintf(stderr, "error: %s\n", e.what());
    return 3;
  }
}
