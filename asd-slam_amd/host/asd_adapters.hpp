// asd_adapters.hpp -- C++ host-side mirror of the reference's call surface for the hot path.
//
// The reference reaches the hot path through three classes of libvslam:
//   ORB_SLAM2::ORBextractor  src/vslam/include/ORBextractor.h:54-87
//   ORB_SLAM2::ORBmatcher    src/vslam/include/ORBmatcher.h:48-90
//   ORB_SLAM2::Optimizer     src/vslam/include/Optimizer.h:45-46
// Their signatures carry cv::Mat / Frame / MapPoint*; OpenCV and the vslam headers are not
// available here, so these mirrors keep the METHOD NAMES, ARGUMENT MEANING and RETURN VALUES but
// take the flat views the methods actually read (asd::FrameView, asd::MapPointView).  Each
// method documents the reference line it stands for.  Header only, depends on include/asd_slam.h
// and the C++ standard library; link with -lasdhip.  INTEGRATION.md shows the few lines that wrap
// these into the reference's own classes inside the catkin workspace.
//
// Error behaviour follows the reference (no exceptions on the data path, "return 0 / -1" on
// failure) except construction, which throws std::runtime_error if no HIP device is usable.
#pragma once
#include <cstdint>
#include <stdexcept>
#include <string>
#include <utility>
#include <algorithm>
#include <vector>

#include "../../include/asd_slam.h"

namespace asd {

struct Camera { float fx, fy, cx, cy; };

// What the matchers and the optimizer read from a Frame (Frame.h): undistorted keypoints,
// descriptors (row-major N x 128 f32), image bounds, pose Tcw (row-major 4x4 f32).
struct FrameView {
  int slot = 0;                         // asd_frame_set slot holding mvKeysUn / mDescriptors / mGrid
  std::vector<asd_keypoint> mvKeysUn;
  std::vector<float> mDescriptors;
  float mnMinX = 0, mnMaxX = 0, mnMinY = 0, mnMaxY = 0;
  float mTcw[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
  std::vector<int32_t> mvpMapPoints;    // index into the caller's map point table, -1 = NULL
  std::vector<uint8_t> mvbOutlier;
  int N() const { return (int)mvKeysUn.size(); }
};

// What SearchByProjection / isInFrustum / PoseOptimization read from MapPoint (MapPoint.h).
struct MapPointView {
  float Xw[3];          // GetWorldPos
  float normal[3];      // GetNormal
  float mfMinDistance, mfMaxDistance;
  const float* descriptor;  // GetDescriptor, 128 f32
  int nObs = 1;             // Observations(): a keypoint holding a map point with nObs == 0 is not skipped (ORBmatcher.cc:86-88)
};

class Context {
 public:
  Context(int nfeatures, float scaleFactor, int nlevels, int iniThFAST, int minThFAST, int maxWidth, int maxHeight,
          int device = 0) {
    asd_config cfg{nfeatures, scaleFactor, nlevels, iniThFAST, minThFAST, maxWidth, maxHeight, 2 * nfeatures, device};
    const int rc = asd_ctx_create(&cfg, &ctx_);
    if (rc != ASD_OK) throw std::runtime_error("asd_ctx_create failed (" + std::to_string(rc) + "): no usable HIP device");
  }
  ~Context() { if (ctx_) asd_ctx_destroy(ctx_); }
  Context(const Context&) = delete;
  Context& operator=(const Context&) = delete;
  asd_ctx* get() const { return ctx_; }
  const char* error() const { return asd_last_error(ctx_); }
 private:
  asd_ctx* ctx_ = nullptr;
};

// ---- ORBextractor (ORBextractor.h:54-87; ctor ORBextractor.cc:452-516) -------------------------
class ORBextractor {
 public:
  // weights replace torch::jit::load("...bestmodel_c.pt") (ORBextractor.cc:457)
  ORBextractor(Context& c, const float* const conv_w[7], const float* const bn_mean[7], const float* const bn_var[7],
               float bn_eps = 1e-5f)
      : c_(c) {
    if (asd_load_weights(c_.get(), conv_w, bn_mean, bn_var, bn_eps) != ASD_OK) throw std::runtime_error(c_.error());
    asd_config dummy{};
    (void)dummy;
  }
  // void ExtractDesc(InputArray image, InputArray mask, vector<KeyPoint>&, OutputArray desc, bool use_orb)
  // (ORBextractor.cc:1137).  mask is ignored by the reference; use_orb=false is the ASD path.
  // Returns the number of keypoints, -1 on error (the reference returns void and asserts).
  int ExtractDesc(const uint8_t* image, int width, int height, int stride, std::vector<asd_keypoint>& keypoints,
                  std::vector<float>& descriptors, int nfeatures_override = 0) {
    const int cap = capacity();
    keypoints.resize(cap);
    descriptors.resize((size_t)cap * ASD_DESC_DIM);
    int32_t n = 0;
    if (asd_extract(c_.get(), image, width, height, stride, nfeatures_override, keypoints.data(), descriptors.data(), &n) != ASD_OK)
      return -1;
    keypoints.resize(n);
    descriptors.resize((size_t)n * ASD_DESC_DIM);
    return n;
  }
  int GetLevels() const { return levels(); }
  std::vector<float> GetScaleFactors() const { return table(0); }
  std::vector<float> GetInverseScaleFactors() const { return table(1); }
  std::vector<float> GetScaleSigmaSquares() const { return table(2); }
  std::vector<float> GetInverseScaleSigmaSquares() const { return table(3); }
  // mvImagePyramid[level] without the 19 px border (ORBextractor.h:87)
  std::vector<uint8_t> ImagePyramidLevel(int level, int* w, int* h) const {
    int32_t ww = 0, hh = 0;
    if (asd_get_level_size(c_.get(), level, &ww, &hh) != ASD_OK) return {};
    std::vector<uint8_t> img((size_t)ww * hh);
    asd_get_level_image(c_.get(), level, 0, img.data());
    *w = ww; *h = hh;
    return img;
  }
 private:
  int levels() const { int n = 0; float s[ASD_MAX_LEVELS]; int32_t f[ASD_MAX_LEVELS]; asd_get_scale_tables(c_.get(), s, nullptr, nullptr, nullptr, f); for (; n < ASD_MAX_LEVELS && f[n] > 0; ++n) {} return n; }
  int capacity() const { return 1 << 13; }
  std::vector<float> table(int which) const {
    float t[4][ASD_MAX_LEVELS] = {};
    asd_get_scale_tables(c_.get(), t[0], t[1], t[2], t[3], nullptr);
    return std::vector<float>(t[which], t[which] + levels());
  }
  Context& c_;
};

// ---- Frame bookkeeping (Frame.cc:64-138): call after ExtractDesc ------------------------------
inline int FrameAssignFeaturesToGrid(Context& c, FrameView& F, bool adopt_last_extract = true) {
  return asd_frame_set(c.get(), F.slot, F.mvKeysUn.data(), adopt_last_extract ? nullptr : F.mDescriptors.data(), F.N(),
                       F.mnMinX, F.mnMaxX, F.mnMinY, F.mnMaxY);
}

// ---- ORBmatcher (ORBmatcher.h:48-90) -----------------------------------------------------------
class ORBmatcher {
 public:
  ORBmatcher(Context& c, float nnratio = 0.6f, bool checkOri = true) : c_(c), mfNNratio(nnratio), mbCheckOrientation(checkOri) {}

  // static float DescriptorDistance(const cv::Mat&, const cv::Mat&)  (ORBmatcher.cc:1629)
  float DescriptorDistance(const float* a, const float* b) {
    float d = -1.f;
    asd_dist_matrix(c_.get(), a, 1, b, 1, &d);
    return d;
  }

  // int SearchByProjection(Frame& CurrentFrame, const Frame& LastFrame, float th, bool bMono)  (:1318)
  // `points` are LastFrame's map points indexed like LastFrame.mvKeysUn (null where mvpMapPoints[i]==NULL
  // or mvbOutlier[i]).  Writes CurrentFrame.mvpMapPoints like the reference.
  int SearchByProjection(FrameView& Cur, const FrameView& Last, const std::vector<const MapPointView*>& points,
                         const Camera& K, float th, bool /*bMono: the reference is monocular only*/ = true) {
    const int nl = Last.N();
    std::vector<uint8_t> has(nl, 0), obs(nl, 1);
    std::vector<float> Xw((size_t)nl * 3, 0.f), desc((size_t)nl * ASD_DESC_DIM, 0.f);
    for (int i = 0; i < nl; ++i)
      if (points[i]) {
        has[i] = 1;
        obs[i] = points[i]->nObs > 0;
        for (int k = 0; k < 3; ++k) Xw[3 * i + k] = points[i]->Xw[k];
        for (int k = 0; k < ASD_DESC_DIM; ++k) desc[(size_t)i * ASD_DESC_DIM + k] = points[i]->descriptor[k];
      }
    std::vector<int32_t> match(Cur.N(), -1);
    int32_t n = 0;
    const float Kv[4] = {K.fx, K.fy, K.cx, K.cy};
    if (asd_match_project_frame(c_.get(), Cur.slot, Last.slot, has.data(), Xw.data(), desc.data(), Cur.mTcw, Kv, th,
                                mbCheckOrientation, match.data(), &n, obs.data()) != ASD_OK)
      return 0;
    Cur.mvpMapPoints.assign(Cur.N(), -1);
    for (int j = 0; j < Cur.N(); ++j)
      if (match[j] >= 0) Cur.mvpMapPoints[j] = Last.mvpMapPoints[match[j]];
    return n;
  }

  // int SearchByProjection(Frame& F, const vector<MapPoint*>& vpMapPoints, float th)  (:44), preceded by
  // Frame::isInFrustum for every point as Tracking::SearchLocalPoints does (Tracking.cc:803-851).
  int SearchByProjection(FrameView& F, const std::vector<MapPointView>& vpMapPoints, const std::vector<int32_t>& ids,
                         const Camera& K, float th = 3.f) {
    const int n = (int)vpMapPoints.size();
    std::vector<float> Xw((size_t)n * 3), nrm((size_t)n * 3), mind(n), maxd(n), desc((size_t)n * ASD_DESC_DIM);
    for (int m = 0; m < n; ++m) {
      for (int k = 0; k < 3; ++k) { Xw[3 * m + k] = vpMapPoints[m].Xw[k]; nrm[3 * m + k] = vpMapPoints[m].normal[k]; }
      mind[m] = vpMapPoints[m].mfMinDistance; maxd[m] = vpMapPoints[m].mfMaxDistance;
      for (int k = 0; k < ASD_DESC_DIM; ++k) desc[(size_t)m * ASD_DESC_DIM + k] = vpMapPoints[m].descriptor[k];
    }
    std::vector<uint8_t> in_view(n), occupied(F.N(), 0), obs(n, 1);
    for (int m = 0; m < n; ++m) obs[m] = vpMapPoints[m].nObs > 0;
    std::vector<float> proj((size_t)n * 2), vc(n);
    std::vector<int32_t> level(n), match(F.N(), -1);
    const float Kv[4] = {K.fx, K.fy, K.cx, K.cy};
    if (asd_frustum(c_.get(), F.slot, n, Xw.data(), nrm.data(), mind.data(), maxd.data(), F.mTcw, Kv, 0.5f,
                    in_view.data(), proj.data(), level.data(), vc.data()) != ASD_OK)
      return 0;
    for (int j = 0; j < F.N(); ++j) occupied[j] = F.mvpMapPoints.size() == (size_t)F.N() && F.mvpMapPoints[j] >= 0;
    int32_t nm = 0;
    if (asd_match_project_points(c_.get(), F.slot, n, in_view.data(), proj.data(), level.data(), vc.data(), desc.data(),
                                 occupied.data(), th, mfNNratio, match.data(), &nm, obs.data()) != ASD_OK)
      return 0;
    F.mvpMapPoints.resize(F.N(), -1);
    for (int j = 0; j < F.N(); ++j)
      if (match[j] >= 0) F.mvpMapPoints[j] = ids[match[j]];
    return nm;
  }

  // int SearchForInitialization(Frame& F1, Frame& F2, vector<Point2f>& vbPrevMatched, vector<int>& vnMatches12,
  //                             int windowSize)  (:416)
  int SearchForInitialization(const FrameView& F1, const FrameView& F2, std::vector<float>& vbPrevMatched,
                              std::vector<int>& vnMatches12, int windowSize = 10) {
    vnMatches12.assign(F1.N(), -1);
    int32_t n = 0;
    if (asd_match_init(c_.get(), F1.slot, F2.slot, vbPrevMatched.data(), windowSize, mfNNratio, mbCheckOrientation,
                       vnMatches12.data(), &n) != ASD_OK)
      return 0;
    return n;
  }

  // DBoW2::FeatureVector as the matchers read it (node id -> keypoint indices), CSR over ascending node ids
  struct FeatVec {
    std::vector<int32_t> node, start{0}, idx;
    asd_feature_vector view() const { return asd_feature_vector{(int32_t)node.size(), node.data(), start.data(), idx.data()}; }
  };

  // int SearchByBoW(KeyFrame* pKF, Frame& F, vector<MapPoint*>& vpMapPointMatches)  (:156)
  // vpMapPointMatches[j] = pKF's map point id matched to F's keypoint j, -1 = NULL
  int SearchByBoW(const FrameView& KF, const FeatVec& fvKF, const FrameView& F, const FeatVec& fvF,
                  std::vector<int32_t>& vpMapPointMatches) {
    std::vector<uint8_t> has(KF.N(), 0);
    for (int i = 0; i < KF.N(); ++i) has[i] = KF.mvpMapPoints.size() == (size_t)KF.N() && KF.mvpMapPoints[i] >= 0;
    std::vector<int32_t> m(F.N(), -1);
    int32_t n = 0;
    const asd_feature_vector a = fvKF.view(), b = fvF.view();
    if (asd_match_bow(c_.get(), KF.slot, F.slot, &a, &b, has.data(), mfNNratio, mbCheckOrientation, m.data(), &n) != ASD_OK) return 0;
    vpMapPointMatches.assign(F.N(), -1);
    for (int j = 0; j < F.N(); ++j)
      if (m[j] >= 0) vpMapPointMatches[j] = KF.mvpMapPoints[m[j]];
    return n;
  }

  // int SearchForTriangulation(KeyFrame* pKF1, KeyFrame* pKF2, cv::Mat F12, vector<pair<size_t,size_t>>& vMatchedPairs,
  //                            bool bOnlyStereo)  (:669).  (ex, ey) = epipole of camera 1 in image 2 (:675-683).
  int SearchForTriangulation(const FrameView& KF1, const FeatVec& fv1, const FrameView& KF2, const FeatVec& fv2, const float F12[9],
                             float ex, float ey, std::vector<std::pair<size_t, size_t>>& vMatchedPairs) {
    std::vector<uint8_t> h1(KF1.N(), 0), h2(KF2.N(), 0);
    for (int i = 0; i < KF1.N(); ++i) h1[i] = KF1.mvpMapPoints.size() == (size_t)KF1.N() && KF1.mvpMapPoints[i] >= 0;
    for (int i = 0; i < KF2.N(); ++i) h2[i] = KF2.mvpMapPoints.size() == (size_t)KF2.N() && KF2.mvpMapPoints[i] >= 0;
    std::vector<int32_t> m(KF1.N(), -1);
    int32_t n = 0;
    const asd_feature_vector a = fv1.view(), b = fv2.view();
    vMatchedPairs.clear();
    if (asd_match_triangulate(c_.get(), KF1.slot, KF2.slot, &a, &b, h1.data(), h2.data(), F12, ex, ey, mbCheckOrientation, m.data(), &n) != ASD_OK)
      return 0;
    for (int i = 0; i < KF1.N(); ++i)
      if (m[i] >= 0) vMatchedPairs.emplace_back((size_t)i, (size_t)m[i]);   // :809-816: ascending idx1
    return n;
  }

  // int Fuse(KeyFrame* pKF, const vector<MapPoint*>& vpMapPoints, float th)  (:825): search half.  bestIdx[i] = keypoint
  // of pKF that map point i should be fused into (-1 = none); the caller then runs the Replace / AddObservation loop
  // (:938-956).  valid[i] = pMP && !pMP->isBad() && !pMP->IsInKeyFrame(pKF).
  int Fuse(const FrameView& KF, const std::vector<MapPointView>& vpMapPoints, const std::vector<uint8_t>& valid, const Camera& K,
           std::vector<int32_t>& bestIdx, float th = 3.f) {
    const int n = (int)vpMapPoints.size();
    std::vector<float> Xw((size_t)n * 3), nrm((size_t)n * 3), mind(n), maxd(n), desc((size_t)n * ASD_DESC_DIM), bd(n);
    for (int m = 0; m < n; ++m) {
      for (int k = 0; k < 3; ++k) { Xw[3 * m + k] = vpMapPoints[m].Xw[k]; nrm[3 * m + k] = vpMapPoints[m].normal[k]; }
      mind[m] = vpMapPoints[m].mfMinDistance; maxd[m] = vpMapPoints[m].mfMaxDistance;
      for (int k = 0; k < ASD_DESC_DIM; ++k) desc[(size_t)m * ASD_DESC_DIM + k] = vpMapPoints[m].descriptor[k];
    }
    bestIdx.assign(n, -1);
    const float Kv[4] = {K.fx, K.fy, K.cx, K.cy};
    if (asd_fuse_search(c_.get(), KF.slot, n, valid.data(), Xw.data(), nrm.data(), mind.data(), maxd.data(), desc.data(), KF.mTcw, Kv, th,
                        bestIdx.data(), bd.data()) != ASD_OK)
      return 0;
    int nFused = 0;
    for (int m = 0; m < n; ++m) nFused += bestIdx[m] >= 0;
    return nFused;
  }

 private:
  Context& c_;
  float mfNNratio;
  bool mbCheckOrientation;
};

// ---- Optimizer (Optimizer.h:45-46) -------------------------------------------------------------
struct Optimizer {
  // int static PoseOptimization(Frame* pFrame)  (Optimizer.cc:239): returns the number of inliers, writes
  // pFrame->mTcw and mvbOutlier.  inv_level_sigma2 = pFrame->mvInvLevelSigma2.
  static int PoseOptimization(Context& c, FrameView* pFrame, const std::vector<const MapPointView*>& points,
                              const Camera& K, const std::vector<float>& inv_level_sigma2) {
    std::vector<int> idx;
    std::vector<double> Xw, obs, info;
    for (int i = 0; i < pFrame->N(); ++i)
      if (points[i]) {
        idx.push_back(i);
        for (int k = 0; k < 3; ++k) Xw.push_back(points[i]->Xw[k]);
        obs.push_back(pFrame->mvKeysUn[i].x); obs.push_back(pFrame->mvKeysUn[i].y);
        info.push_back(inv_level_sigma2[pFrame->mvKeysUn[i].octave]);
      }
    pFrame->mvbOutlier.assign(pFrame->N(), 0);
    double pose[7];
    asd_tcw_to_pose7(pFrame->mTcw, pose);
    std::vector<uint8_t> out(idx.size());
    const double Kd[4] = {K.fx, K.fy, K.cx, K.cy};
    int32_t ninl = 0;
    if (asd_pose_optimize(c.get(), pose, (int)idx.size(), Xw.data(), obs.data(), info.data(), Kd, out.data(), &ninl) != ASD_OK)
      return 0;
    for (size_t k = 0; k < idx.size(); ++k) pFrame->mvbOutlier[idx[k]] = out[k];
    asd_pose7_to_tcw(pose, pFrame->mTcw);
    return ninl;
  }

  // void static LocalBundleAdjustment(KeyFrame* pKF, bool* pbStopFlag, Map* pMap)  (Optimizer.cc:415): the
  // caller gathers local / fixed keyframes and map points exactly as :423-467 and hands them over flat;
  // on return it applies the erase policy of :652-700 from edge_chi2 / edge_depth_pos.
  static int LocalBundleAdjustment(Context& c, asd_ba_problem* problem, asd_ba_result* result, const bool* pbStopFlag = nullptr) {
    if (pbStopFlag && *pbStopFlag) return 0;  // :595-597
    return asd_local_ba(c.get(), problem, result);
  }
};

// ---- Tracking (Tracking.cc): the two per-frame stages as one submission each ------------------------------------------------
// The reference runs matcher.SearchByProjection and Optimizer::PoseOptimization back to back in TrackWithMotionModel (:664-723)
// and in TrackLocalMap (:725-736 with SearchLocalPoints :803-851); asd_track_motion_model / asd_track_local_points do the pair
// behind one synchronisation and return the same bits as the two calls (tests/test_track_chain.py).
struct Tracking {
  // bool Tracking::TrackWithMotionModel(): Cur.mTcw holds the motion-model prediction on entry (mVelocity * mLastFrame.mTcw, :677).
  // `points` = LastFrame's map points indexed like LastFrame.mvKeysUn (null where there is none).  Writes Cur.mvpMapPoints,
  // Cur.mvbOutlier and Cur.mTcw, discards outliers like :704-719 and returns nmatchesMap >= 10 (:723); *nmatches_out gets the
  // matcher's count of the accepted search.
  static bool TrackWithMotionModel(Context& c, FrameView& Cur, const FrameView& Last, const std::vector<const MapPointView*>& points,
                                   const Camera& K, bool checkOrientation = true, int* nmatches_out = nullptr, int* ninliers_out = nullptr) {
    const int nl = Last.N(), nc = Cur.N();
    std::vector<uint8_t> has(nl, 0), obs(nl, 1);
    std::vector<float> Xw((size_t)nl * 3, 0.f), desc((size_t)nl * ASD_DESC_DIM, 0.f);
    for (int i = 0; i < nl; ++i)
      if (points[i]) {
        has[i] = 1;
        obs[i] = points[i]->nObs > 0;
        for (int k = 0; k < 3; ++k) Xw[3 * i + k] = points[i]->Xw[k];
        for (int k = 0; k < ASD_DESC_DIM; ++k) desc[(size_t)i * ASD_DESC_DIM + k] = points[i]->descriptor[k];
      }
    const float Kv[4] = {K.fx, K.fy, K.cx, K.cy};
    std::vector<int32_t> match(nc, -1);
    std::vector<uint8_t> outl(std::max(nc, 1), 0);
    double pose[7], pose_in[7];
    asd_tcw_to_pose7(Cur.mTcw, pose_in);
    int32_t n = 0, ninl = 0;
    float th = 15.f;                                                   // :673-675 (monocular)
    for (int attempt = 0; attempt < 2; ++attempt, th *= 2) {           // :679-687: a second search with 2*th when fewer than 20 matches
      for (int k = 0; k < 7; ++k) pose[k] = pose_in[k];
      if (asd_track_motion_model(c.get(), Cur.slot, Last.slot, has.data(), Xw.data(), desc.data(), Cur.mTcw, Kv, th, checkOrientation, obs.data(),
                                 pose, match.data(), &n, outl.data(), &ninl) != ASD_OK)
        return false;
      if (n >= 20) break;
    }
    if (nmatches_out) *nmatches_out = n;
    if (ninliers_out) *ninliers_out = ninl;
    Cur.mvpMapPoints.assign(nc, -1);
    Cur.mvbOutlier.assign(nc, 0);
    if (n < 20) return false;                                          // :689-690
    asd_pose7_to_tcw(pose, Cur.mTcw);
    int nmatchesMap = 0;
    for (int j = 0; j < nc; ++j) {
      if (match[j] < 0) continue;
      if (outl[j]) continue;                                           // :708-716: the map point is dropped from the frame
      Cur.mvpMapPoints[j] = Last.mvpMapPoints[match[j]];
      if (points[match[j]]->nObs > 0) ++nmatchesMap;                   // :717-718
    }
    return nmatchesMap >= 10;
  }

  // Tracking::SearchLocalPoints + Optimizer::PoseOptimization (TrackLocalMap's numeric body): `local` = mvpLocalMapPoints not
  // already in the frame, `ids` their map-point ids, `cur_points` the points the frame already holds (indexed like mvKeysUn,
  // null where none).  Returns the inlier count; writes Cur.mvpMapPoints (new matches), mvbOutlier and mTcw.
  static int TrackLocalMap(Context& c, FrameView& Cur, const std::vector<MapPointView>& local, const std::vector<int32_t>& ids,
                           const std::vector<const MapPointView*>& cur_points, const Camera& K, float th = 1.f, float nnratio = 0.8f) {
    const int n = (int)local.size(), nc = Cur.N();
    std::vector<float> Xw((size_t)n * 3), nrm((size_t)n * 3), mind(n), maxd(n), desc((size_t)n * ASD_DESC_DIM), curX((size_t)nc * 3, 0.f);
    std::vector<uint8_t> obs(n, 1), occupied(nc, 0), outl(std::max(nc, 1), 0);
    for (int m = 0; m < n; ++m) {
      for (int k = 0; k < 3; ++k) { Xw[3 * m + k] = local[m].Xw[k]; nrm[3 * m + k] = local[m].normal[k]; }
      mind[m] = local[m].mfMinDistance; maxd[m] = local[m].mfMaxDistance;
      obs[m] = local[m].nObs > 0;
      for (int k = 0; k < ASD_DESC_DIM; ++k) desc[(size_t)m * ASD_DESC_DIM + k] = local[m].descriptor[k];
    }
    for (int j = 0; j < nc; ++j)
      if (cur_points[j]) { occupied[j] = 1; for (int k = 0; k < 3; ++k) curX[3 * j + k] = cur_points[j]->Xw[k]; }
    const float Kv[4] = {K.fx, K.fy, K.cx, K.cy};
    std::vector<int32_t> match(nc, -1);
    double pose[7];
    asd_tcw_to_pose7(Cur.mTcw, pose);
    int32_t nm = 0, ninl = 0;
    if (asd_track_local_points(c.get(), Cur.slot, n, Xw.data(), nrm.data(), mind.data(), maxd.data(), desc.data(), Cur.mTcw, Kv, 0.5f,
                               occupied.data(), curX.data(), th, nnratio, obs.data(), pose, match.data(), &nm, outl.data(), &ninl) != ASD_OK)
      return 0;
    asd_pose7_to_tcw(pose, Cur.mTcw);
    Cur.mvpMapPoints.resize(nc, -1);
    Cur.mvbOutlier.assign(nc, 0);
    for (int j = 0; j < nc; ++j) {
      if (match[j] >= 0) Cur.mvpMapPoints[j] = ids[match[j]];
      Cur.mvbOutlier[j] = outl[j];
    }
    return ninl;
  }
};

// OPTIONAL: Optimizer::LocalBundleAdjustment on the library's lane.  The reference calls it IN LINE (Tracking.cc:797 ->
// LocalMapping::DoMapping, LocalMapping.cc:89; no mapping thread exists in this fork) -- that is asd::Optimizer::LocalBundleAdjustment /
// asd_local_ba.  Submit returns at once (problem / result stay owned by the library), Tracking goes on against the PRE-BA map, Wait
// returns the run's status: upstream ORB-SLAM2's concurrency, a different data dependency from this reference.
struct LocalMapping {
  static int LocalBundleAdjustmentSubmit(Context& c, asd_ba_problem* problem, asd_ba_result* result) { return asd_local_ba_submit(c.get(), problem, result); }
  static int LocalBundleAdjustmentWait(Context& c) { return asd_local_ba_wait(c.get()); }
  static bool Busy(Context& c) { return asd_local_ba_poll(c.get()) == 1; }
};

// MapPoint::ComputeDistinctiveDescriptors for a batch of map points (MapPoint.cc:271-338): observations[s] = the
// descriptors (128 f32 each) of map point s's observations; returns the chosen observation index per map point
inline int ComputeDistinctiveDescriptors(Context& c, const std::vector<std::vector<const float*>>& observations, std::vector<int32_t>& best) {
  std::vector<int32_t> start(observations.size() + 1, 0);
  for (size_t s = 0; s < observations.size(); ++s) start[s + 1] = start[s] + (int32_t)observations[s].size();
  std::vector<float> desc((size_t)start.back() * ASD_DESC_DIM);
  for (size_t s = 0; s < observations.size(); ++s)
    for (size_t k = 0; k < observations[s].size(); ++k)
      for (int q = 0; q < ASD_DESC_DIM; ++q) desc[((size_t)start[s] + k) * ASD_DESC_DIM + q] = observations[s][k][q];
  best.assign(observations.size(), 0);
  return asd_distinctive_descriptor_batch(c.get(), (int32_t)observations.size(), start.data(), desc.data(), best.data());
}

// ---- vocabulary / local mapping (ORBVocabulary.h:34, Frame.cc:289-296, LocalMapping.cc:299-519) ---
// void Frame::ComputeBoW(): BowVector (word id -> weight) and FeatureVector (node id at levelsup = 4 -> keypoints)
inline int ComputeBoW(Context& c, const FrameView& F, std::vector<std::pair<int32_t, double>>& mBowVec, ORBmatcher::FeatVec& mFeatVec,
                      int levelsup = 4) {
  const int N = F.N();
  std::vector<int32_t> bid(N + 1);
  std::vector<double> bval(N + 1);
  mFeatVec.node.assign(N + 1, 0); mFeatVec.start.assign(N + 1, 0); mFeatVec.idx.assign(N + 1, 0);
  int32_t nw = 0, nn = 0;
  const int rc = asd_compute_bow(c.get(), F.slot, nullptr, N, levelsup, bid.data(), bval.data(), &nw, mFeatVec.node.data(),
                                 mFeatVec.start.data(), mFeatVec.idx.data(), &nn);
  if (rc != ASD_OK) return rc;
  mBowVec.clear();
  for (int k = 0; k < nw; ++k) mBowVec.emplace_back(bid[k], bval[k]);
  mFeatVec.node.resize(nn);
  mFeatVec.start.resize(nn + 1);
  mFeatVec.idx.resize(mFeatVec.start[nn]);
  return ASD_OK;
}

// LocalMapping::CreateNewMapPoints, per-match body (:386-519): x3D[3*i..] and ok[i] for every matched pair
inline int TriangulateMatches(Context& c, const FrameView& KF1, const FrameView& KF2, const Camera& K1, const Camera& K2,
                              const std::vector<std::pair<size_t, size_t>>& vMatchedIndices, std::vector<float>& x3D,
                              std::vector<uint8_t>& ok) {
  const int n = (int)vMatchedIndices.size();
  std::vector<int32_t> i1(n), i2(n);
  for (int i = 0; i < n; ++i) { i1[i] = (int32_t)vMatchedIndices[i].first; i2[i] = (int32_t)vMatchedIndices[i].second; }
  x3D.assign((size_t)3 * n, 0.f);
  ok.assign(n, 0);
  int32_t nnew = 0;
  const float k1[4] = {K1.fx, K1.fy, K1.cx, K1.cy}, k2[4] = {K2.fx, K2.fy, K2.cx, K2.cy};
  if (asd_triangulate_pairs(c.get(), KF1.slot, KF2.slot, n, i1.data(), i2.data(), KF1.mTcw, KF2.mTcw, k1, k2, x3D.data(), ok.data(), &nnew) != ASD_OK)
    return 0;
  return nnew;
}

}  // namespace asd
