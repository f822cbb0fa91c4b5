// driver.cpp -- flat-array entry points over the REFERENCE's vendored g2o (compiled in place by the
// Makefile next to this file).  TEST INFRASTRUCTURE ONLY: produces tests/golden/ba_*.npz and serves as
// the "reference" CPU baseline in bench.py.  Graph construction and schedules follow
// /root/reference/src/vslam/src/Optimizer.cc:241-404 (PoseOptimization) and :470-671
// (LocalBundleAdjustment) with the pointer graph replaced by the flat arrays of include/asd_slam.h.
// Signatures match oracle/oracle.h (orc_pose_optimize / orc_local_ba) so one ctypes wrapper drives both.
#include <cmath>
#include <cstdint>
#include <vector>

#include "g2o/core/block_solver.h"
#include "g2o/core/optimization_algorithm_levenberg.h"
#include "g2o/core/robust_kernel_impl.h"
#include "g2o/solvers/linear_solver_dense.h"
#include "g2o/types/types_six_dof_expmap.h"

struct ref_ba_out { double chi2_first, chi2_second; int32_t iters_first, iters_second; };

static g2o::SE3Quat from7(const double* p) {
  return g2o::SE3Quat(Eigen::Quaterniond(p[3], p[0], p[1], p[2]), Eigen::Vector3d(p[4], p[5], p[6]));
}
static void to7(const g2o::SE3Quat& T, double* p) {
  const Eigen::Quaterniond& q = T.rotation();
  p[0] = q.x(); p[1] = q.y(); p[2] = q.z(); p[3] = q.w();
  p[4] = T.translation()[0]; p[5] = T.translation()[1]; p[6] = T.translation()[2];
}

extern "C" int ref_pose_optimize(double* pose7, int n, const double* Xw, const double* obs, const double* inv_sigma2,
                                 const double* K, uint8_t* outlier) {
  g2o::SparseOptimizer optimizer;
  g2o::OptimizationAlgorithmLevenberg* solver = new g2o::OptimizationAlgorithmLevenberg(
      new g2o::BlockSolver_6_3(new g2o::LinearSolverDense<g2o::BlockSolver_6_3::PoseMatrixType>()));
  optimizer.setAlgorithm(solver);
  int nInitialCorrespondences = 0;
  const g2o::SE3Quat T0 = from7(pose7);
  g2o::VertexSE3Expmap* vSE3 = new g2o::VertexSE3Expmap();
  vSE3->setEstimate(T0);
  vSE3->setId(0);
  vSE3->setFixed(false);
  optimizer.addVertex(vSE3);
  std::vector<g2o::EdgeSE3ProjectXYZOnlyPose*> vpEdgesMono;
  const float deltaMono = sqrt(5.991);
  for (int i = 0; i < n; i++) {
    nInitialCorrespondences++;
    outlier[i] = false;
    Eigen::Matrix<double, 2, 1> o;
    o << obs[2 * i], obs[2 * i + 1];
    g2o::EdgeSE3ProjectXYZOnlyPose* e = new g2o::EdgeSE3ProjectXYZOnlyPose();
    e->setVertex(0, dynamic_cast<g2o::OptimizableGraph::Vertex*>(optimizer.vertex(0)));
    e->setMeasurement(o);
    e->setInformation(Eigen::Matrix2d::Identity() * inv_sigma2[i]);
    g2o::RobustKernelHuber* rk = new g2o::RobustKernelHuber;
    e->setRobustKernel(rk);
    rk->setDelta(deltaMono);
    e->fx = K[0]; e->fy = K[1]; e->cx = K[2]; e->cy = K[3];
    e->Xw[0] = Xw[3 * i]; e->Xw[1] = Xw[3 * i + 1]; e->Xw[2] = Xw[3 * i + 2];
    optimizer.addEdge(e);
    vpEdgesMono.push_back(e);
  }
  if (nInitialCorrespondences < 3) return 0;
  const float chi2Mono[4] = {5.991, 5.991, 5.991, 5.991};
  const int its[4] = {10, 10, 10, 10};
  int nBad = 0;
  for (size_t it = 0; it < 4; it++) {
    vSE3->setEstimate(T0);
    optimizer.initializeOptimization(0);
    optimizer.optimize(its[it]);
    nBad = 0;
    for (size_t i = 0, iend = vpEdgesMono.size(); i < iend; i++) {
      g2o::EdgeSE3ProjectXYZOnlyPose* e = vpEdgesMono[i];
      if (outlier[i]) e->computeError();
      const float chi2 = e->chi2();
      if (chi2 > chi2Mono[it]) { outlier[i] = true; e->setLevel(1); nBad++; }
      else { outlier[i] = false; e->setLevel(0); }
      if (it == 2) e->setRobustKernel(0);
    }
    if (optimizer.edges().size() < 10) break;
  }
  g2o::VertexSE3Expmap* vr = static_cast<g2o::VertexSE3Expmap*>(optimizer.vertex(0));
  to7(vr->estimate(), pose7);
  return nInitialCorrespondences - nBad;
}

extern "C" int ref_local_ba(int n_poses, int n_points, int n_edges, double* poses, const uint8_t* fixed, double* points,
                            const int32_t* e_point, const int32_t* e_pose, const double* e_obs, const double* e_info,
                            const double* K, int its1, int its2, double* edge_chi2, uint8_t* edge_depth_pos,
                            uint8_t* edge_outlier1, ref_ba_out* out) {
  g2o::SparseOptimizer optimizer;
  g2o::OptimizationAlgorithmLevenberg* solver = new g2o::OptimizationAlgorithmLevenberg(
      new g2o::BlockSolver_6_3(new g2o::LinearSolverDense<g2o::BlockSolver_6_3::PoseMatrixType>()));
  optimizer.setAlgorithm(solver);
  // keyframe vertices: id = pose index (poses arrive ordered by ascending KF id)
  for (int p = 0; p < n_poses; ++p) {
    g2o::VertexSE3Expmap* vSE3 = new g2o::VertexSE3Expmap();
    vSE3->setEstimate(from7(poses + 7 * p));
    vSE3->setId(p);
    vSE3->setFixed(fixed[p] != 0);
    optimizer.addVertex(vSE3);
  }
  const int maxKFid = n_poses - 1;
  const float thHuberMono = sqrt(5.991);
  std::vector<g2o::EdgeSE3ProjectXYZ*> vpEdgesMono(n_edges);
  for (int l = 0; l < n_points; ++l) {
    g2o::VertexSBAPointXYZ* vPoint = new g2o::VertexSBAPointXYZ();
    vPoint->setEstimate(Eigen::Vector3d(points[3 * l], points[3 * l + 1], points[3 * l + 2]));
    vPoint->setId(l + maxKFid + 1);
    vPoint->setMarginalized(true);
    vPoint->setFixed(false);
    optimizer.addVertex(vPoint);
  }
  for (int k = 0; k < n_edges; ++k) {
    Eigen::Matrix<double, 2, 1> o;
    o << e_obs[2 * k], e_obs[2 * k + 1];
    g2o::EdgeSE3ProjectXYZ* e = new g2o::EdgeSE3ProjectXYZ();
    e->setVertex(0, dynamic_cast<g2o::OptimizableGraph::Vertex*>(optimizer.vertex(e_point[k] + maxKFid + 1)));
    e->setVertex(1, dynamic_cast<g2o::OptimizableGraph::Vertex*>(optimizer.vertex(e_pose[k])));
    e->setMeasurement(o);
    e->setInformation(Eigen::Matrix2d::Identity() * e_info[k]);
    g2o::RobustKernelHuber* rk = new g2o::RobustKernelHuber;
    e->setRobustKernel(rk);
    rk->setDelta(thHuberMono);
    e->fx = K[0]; e->fy = K[1]; e->cx = K[2]; e->cy = K[3];
    optimizer.addEdge(e);
    vpEdgesMono[k] = e;
  }
  optimizer.initializeOptimization();
  out->iters_first = optimizer.optimize(its1);
  out->chi2_first = optimizer.activeRobustChi2();  // from the edges' stored errors, nothing recomputed
  for (int k = 0; k < n_edges; ++k) {
    g2o::EdgeSE3ProjectXYZ* e = vpEdgesMono[k];
    const bool bad = e->chi2() > 5.991 || !e->isDepthPositive();
    edge_outlier1[k] = bad;
    if (bad) e->setLevel(1);
    e->setRobustKernel(0);
  }
  optimizer.initializeOptimization(0);
  out->iters_second = optimizer.optimize(its2);
  for (int k = 0; k < n_edges; ++k) {
    g2o::EdgeSE3ProjectXYZ* e = vpEdgesMono[k];
    edge_chi2[k] = e->chi2();
    edge_depth_pos[k] = e->isDepthPositive();
  }
  out->chi2_second = optimizer.activeChi2();
  for (int p = 0; p < n_poses; ++p)
    to7(static_cast<g2o::VertexSE3Expmap*>(optimizer.vertex(p))->estimate(), poses + 7 * p);
  for (int l = 0; l < n_points; ++l) {
    const Eigen::Vector3d& x = static_cast<g2o::VertexSBAPointXYZ*>(optimizer.vertex(l + maxKFid + 1))->estimate();
    points[3 * l] = x[0]; points[3 * l + 1] = x[1]; points[3 * l + 2] = x[2];
  }
  return 0;
}
