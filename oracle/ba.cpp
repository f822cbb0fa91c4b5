// oracle/ba.cpp -- CPU restatement of Optimizer::PoseOptimization and the numeric core of
// Optimizer::LocalBundleAdjustment.  TEST INFRASTRUCTURE ONLY (see oracle.h).
//
// Reference:
//   src/vslam/src/Optimizer.cc:239-413 (PoseOptimization), :415-734 (LocalBundleAdjustment)
//   src/vslam/src/Converter.cc:37-71 (toSE3Quat / toCvMat)
//   src/g2o_catkin:
//     src/core/optimization_algorithm_levenberg.cpp:61-189   LM loop, lambda init, gain ratio
//     src/core/sparse_optimizer.cpp:206-267,340-435          active sets, optimize(), update()
//     include/g2o/core/block_solver.hpp:354-604              Schur complement, buildSystem, lambda
//     include/g2o/core/base_binary_edge.hpp:55-120, base_unary_edge.hpp:43-72  quadratic form
//     src/core/robust_kernel_impl.cpp:78-91                  Huber
//     src/types/types_six_dof_expmap.cpp:115-151,372-394     Jacobians
//     include/g2o/types/se3quat.h:58-60,223-257,280-285      SE3Quat ctor / exp / normalizeRotation
//     include/g2o/solvers/linear_solver_dense.h:65-113       dense LDLT (here: Cholesky)
// Pinned by tests/golden/ba_*.npz, produced by the reference's own g2o compiled in place
// (oracle/ref_g2o).  Summation order differs from g2o's hash/pointer-ordered containers, so
// agreement is to ~1e-9, not bitwise.
#include "oracle.h"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <limits>
#include <vector>

namespace {

struct Quat { double x, y, z, w; };
struct Pose { Quat q; double t[3]; };

void normalize_rotation(Quat& q) {  // se3quat.h:280-285
  if (q.w < 0) { q.x = -q.x; q.y = -q.y; q.z = -q.z; q.w = -q.w; }
  const double n = std::sqrt(q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w);
  q.x /= n; q.y /= n; q.z /= n; q.w /= n;
}
void quat_to_rot(const Quat& q, double R[9]) {  // Eigen QuaternionBase::toRotationMatrix
  const double tx = 2 * q.x, ty = 2 * q.y, tz = 2 * q.z;
  const double twx = tx * q.w, twy = ty * q.w, twz = tz * q.w;
  const double txx = tx * q.x, txy = ty * q.x, txz = tz * q.x, tyy = ty * q.y, tyz = tz * q.y, tzz = tz * q.z;
  R[0] = 1 - (tyy + tzz); R[1] = txy - twz; R[2] = txz + twy;
  R[3] = txy + twz; R[4] = 1 - (txx + tzz); R[5] = tyz - twx;
  R[6] = txz - twy; R[7] = tyz + twx; R[8] = 1 - (txx + tyy);
}
Quat rot_to_quat(const double R[9]) {  // Eigen quaternionbase_assign_impl<Matrix3d>
  Quat q;
  const double t = R[0] + R[4] + R[8];
  if (t > 0) {
    double s = std::sqrt(t + 1.0);
    q.w = 0.5 * s;
    s = 0.5 / s;
    q.x = (R[7] - R[5]) * s; q.y = (R[2] - R[6]) * s; q.z = (R[3] - R[1]) * s;
  } else {
    int i = 0;
    if (R[4] > R[0]) i = 1;
    if (R[8] > R[i * 4]) i = 2;
    const int j = (i + 1) % 3, k = (j + 1) % 3;
    double s = std::sqrt(R[i * 4] - R[j * 4] - R[k * 4] + 1.0);
    double v[3];
    v[i] = 0.5 * s;
    s = 0.5 / s;
    q.w = (R[k * 3 + j] - R[j * 3 + k]) * s;
    v[j] = (R[j * 3 + i] + R[i * 3 + j]) * s;
    v[k] = (R[k * 3 + i] + R[i * 3 + k]) * s;
    q.x = v[0]; q.y = v[1]; q.z = v[2];
  }
  return q;
}
Quat quat_mul(const Quat& a, const Quat& b) {
  return Quat{a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y, a.w * b.y + a.y * b.w + a.z * b.x - a.x * b.z,
              a.w * b.z + a.z * b.w + a.x * b.y - a.y * b.x, a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z};
}
void quat_rotate(const Quat& q, const double v[3], double out[3]) {  // Eigen _transformVector
  double uv[3] = {q.y * v[2] - q.z * v[1], q.z * v[0] - q.x * v[2], q.x * v[1] - q.y * v[0]};
  uv[0] += uv[0]; uv[1] += uv[1]; uv[2] += uv[2];
  out[0] = v[0] + q.w * uv[0] + (q.y * uv[2] - q.z * uv[1]);
  out[1] = v[1] + q.w * uv[1] + (q.z * uv[0] - q.x * uv[2]);
  out[2] = v[2] + q.w * uv[2] + (q.x * uv[1] - q.y * uv[0]);
}
void pose_map(const Pose& T, const double X[3], double out[3]) {  // SE3Quat::map
  quat_rotate(T.q, X, out);
  out[0] += T.t[0]; out[1] += T.t[1]; out[2] += T.t[2];
}
// VertexSE3Expmap::oplusImpl: estimate = SE3Quat::exp(update) * estimate (se3quat.h:223-257, operator*)
Pose pose_oplus(const Pose& T, const double u[6]) {
  const double omega[3] = {u[0], u[1], u[2]}, ups[3] = {u[3], u[4], u[5]};
  const double theta = std::sqrt(omega[0] * omega[0] + omega[1] * omega[1] + omega[2] * omega[2]);
  const double O[9] = {0, -omega[2], omega[1], omega[2], 0, -omega[0], -omega[1], omega[0], 0};
  double O2[9];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) O2[i * 3 + j] = O[i * 3] * O[j] + O[i * 3 + 1] * O[3 + j] + O[i * 3 + 2] * O[6 + j];
  double R[9], V[9];
  if (theta < 0.00001) {
    for (int i = 0; i < 9; ++i) { R[i] = (i % 4 == 0 ? 1.0 : 0.0) + O[i] + O2[i]; V[i] = R[i]; }
  } else {
    const double a = std::sin(theta) / theta, b = (1 - std::cos(theta)) / (theta * theta);
    const double c = (theta - std::sin(theta)) / std::pow(theta, 3);
    for (int i = 0; i < 9; ++i) {
      const double I = (i % 4 == 0 ? 1.0 : 0.0);
      R[i] = I + a * O[i] + b * O2[i];
      V[i] = I + b * O[i] + c * O2[i];
    }
  }
  Pose E;
  E.q = rot_to_quat(R);
  for (int i = 0; i < 3; ++i) E.t[i] = V[i * 3] * ups[0] + V[i * 3 + 1] * ups[1] + V[i * 3 + 2] * ups[2];
  normalize_rotation(E.q);  // SE3Quat(Quaterniond, Vector3d) ctor
  Pose out;  // operator*: t = E.t + E.r * T.t ; r = E.r * T.r ; normalizeRotation
  double rt[3];
  quat_rotate(E.q, T.t, rt);
  for (int i = 0; i < 3; ++i) out.t[i] = E.t[i] + rt[i];
  out.q = quat_mul(E.q, T.q);
  normalize_rotation(out.q);
  return out;
}

Pose pose_from7(const double* p) {
  Pose T;
  T.q = Quat{p[0], p[1], p[2], p[3]};
  normalize_rotation(T.q);
  T.t[0] = p[4]; T.t[1] = p[5]; T.t[2] = p[6];
  return T;
}
void pose_to7(const Pose& T, double* p) {
  p[0] = T.q.x; p[1] = T.q.y; p[2] = T.q.z; p[3] = T.q.w;
  p[4] = T.t[0]; p[5] = T.t[1]; p[6] = T.t[2];
}

const double kHuberDelta = (double)(float)std::sqrt(5.991);  // "const float deltaMono = sqrt(5.991)" (Optimizer.cc:271, :530)
struct Rho { double r0, r1; };
inline Rho huber(double e2, double delta) {  // robust_kernel_impl.cpp:78-91
  const double dsqr = delta * delta;
  if (e2 <= dsqr) return Rho{e2, 1.0};
  const double sqrte = std::sqrt(e2);
  return Rho{2 * sqrte * delta - dsqr, delta / sqrte};
}

// in-place Cholesky solve of the symmetric n x n system (stand-in for Eigen::LDLT + isPositive)
bool chol_solve(std::vector<double>& A, int n, const double* b, double* x) {
  for (int j = 0; j < n; ++j) {
    double d = A[(size_t)j * n + j];
    for (int k = 0; k < j; ++k) d -= A[(size_t)j * n + k] * A[(size_t)j * n + k];
    if (!(d > 0)) return false;
    d = std::sqrt(d);
    A[(size_t)j * n + j] = d;
    for (int i = j + 1; i < n; ++i) {
      double s = A[(size_t)i * n + j];
      for (int k = 0; k < j; ++k) s -= A[(size_t)i * n + k] * A[(size_t)j * n + k];
      A[(size_t)i * n + j] = s / d;
    }
  }
  for (int i = 0; i < n; ++i) {
    double s = b[i];
    for (int k = 0; k < i; ++k) s -= A[(size_t)i * n + k] * x[k];
    x[i] = s / A[(size_t)i * n + i];
  }
  for (int i = n - 1; i >= 0; --i) {
    double s = x[i];
    for (int k = i + 1; k < n; ++k) s -= A[(size_t)k * n + i] * x[k];
    x[i] = s / A[(size_t)i * n + i];
  }
  return true;
}

// pose Jacobian rows (types_six_dof_expmap.cpp:136-150 / :382-394)
inline void jac_pose(double x, double y, double z, double fx, double fy, double J[12]) {
  const double z_2 = z * z;
  J[0] = x * y / z_2 * fx; J[1] = -(1 + (x * x / z_2)) * fx; J[2] = y / z * fx;
  J[3] = -1. / z * fx; J[4] = 0; J[5] = x / z_2 * fx;
  J[6] = (1 + y * y / z_2) * fy; J[7] = -x * y / z_2 * fy; J[8] = -x / z * fy;
  J[9] = 0; J[10] = -1. / z * fy; J[11] = y / z_2 * fy;
}

const float kChi2Mono = 5.991f;  // compared as float in the reference (Optimizer.cc:329, 361-363)

// ---------------------------------------------------------------- PoseOptimization
struct PoseProblem {
  int n;
  const double *Xw, *obs, *info;
  double fx, fy, cx, cy;
  std::vector<double> err;      // last computed error per edge (edge->_error)
  std::vector<uint8_t> level;   // 0 active, 1 outlier
  bool robust;

  void compute_error(const Pose& T, int i) {
    double Xc[3];
    pose_map(T, Xw + 3 * i, Xc);
    const double invz = 1.0 / Xc[2];  // EdgeSE3ProjectXYZOnlyPose::cam_project -> project2d = v / v[2]
    err[2 * i] = obs[2 * i] - (Xc[0] / Xc[2] * fx + cx);
    err[2 * i + 1] = obs[2 * i + 1] - (Xc[1] / Xc[2] * fy + cy);
    (void)invz;
  }
  double chi2(int i) const { return (err[2 * i] * err[2 * i] + err[2 * i + 1] * err[2 * i + 1]) * info[i]; }
  void compute_active_errors(const Pose& T) {
    for (int i = 0; i < n; ++i) if (!level[i]) compute_error(T, i);
  }
  double active_robust_chi2() const {
    double s = 0;
    for (int i = 0; i < n; ++i)
      if (!level[i]) { const double c = chi2(i); s += robust ? huber(c, kHuberDelta).r0 : c; }
    return s;
  }
  void build(const Pose& T, double H[36], double b[6]) const {
    memset(H, 0, 36 * sizeof(double));
    memset(b, 0, 6 * sizeof(double));
    for (int i = 0; i < n; ++i) {
      if (level[i]) continue;
      double Xc[3], J[12];
      pose_map(T, Xw + 3 * i, Xc);
      const double invz = 1.0 / Xc[2], invz_2 = invz * invz, x = Xc[0], y = Xc[1];
      J[0] = x * y * invz_2 * fx; J[1] = -(1 + (x * x * invz_2)) * fx; J[2] = y * invz * fx;
      J[3] = -invz * fx; J[4] = 0; J[5] = x * invz_2 * fx;
      J[6] = (1 + y * y * invz_2) * fy; J[7] = -x * y * invz_2 * fy; J[8] = -x * invz * fy;
      J[9] = 0; J[10] = -invz * fy; J[11] = y * invz_2 * fy;
      const double w = robust ? huber(chi2(i), kHuberDelta).r1 : 1.0;
      const double om = info[i] * w;
      for (int r = 0; r < 6; ++r) {
        b[r] -= om * (J[r] * err[2 * i] + J[6 + r] * err[2 * i + 1]);  // b += J^T * (-w*Omega*e)
        for (int c = 0; c < 6; ++c) H[r * 6 + c] += om * (J[r] * J[c] + J[6 + r] * J[6 + c]);
      }
    }
  }
};

// one g2o optimize(iterations) call on the single-pose graph
int pose_lm(PoseProblem& P, Pose& T, int iterations) {
  double lambda = -1, ni = 2;
  int nBad = 0, done = 0;
  for (int it = 0; it < iterations; ++it) {
    P.compute_active_errors(T);
    double currentChi = P.active_robust_chi2(), tempChi = currentChi;
    const double iniChi = currentChi;
    double H[36], b[6];
    P.build(T, H, b);
    if (it == 0) {
      double maxDiag = 0;
      for (int j = 0; j < 6; ++j) maxDiag = std::max(std::fabs(H[j * 6 + j]), maxDiag);
      lambda = 1e-5 * maxDiag;
      ni = 2;
      nBad = 0;
    }
    double rho = 0;
    int qmax = 0;
    do {
      const Pose backup = T;
      std::vector<double> A(H, H + 36);
      for (int j = 0; j < 6; ++j) A[j * 6 + j] += lambda;
      double x[6] = {0, 0, 0, 0, 0, 0};
      const bool ok2 = chol_solve(A, 6, b, x);
      T = pose_oplus(T, x);
      P.compute_active_errors(T);
      tempChi = P.active_robust_chi2();
      if (!ok2) tempChi = std::numeric_limits<double>::max();
      rho = currentChi - tempChi;
      double scale = 0;
      for (int j = 0; j < 6; ++j) scale += x[j] * (lambda * x[j] + b[j]);
      scale += 1e-3;
      rho /= scale;
      if (rho > 0 && std::isfinite(tempChi)) {
        double alpha = 1. - std::pow((2 * rho - 1), 3);
        alpha = std::min(alpha, 2. / 3.);
        const double scaleFactor = std::max(1. / 3., alpha);
        lambda *= scaleFactor;
        ni = 2;
        currentChi = tempChi;
      } else {
        lambda *= ni;
        ni *= 2;
        T = backup;
      }
      qmax++;
    } while (rho < 0 && qmax < 10);
    ++done;
    if (qmax == 10 || rho == 0) break;
    if ((iniChi - currentChi) * 1e3 < iniChi) nBad++; else nBad = 0;
    if (nBad >= 3) break;
  }
  return done;
}

// ---------------------------------------------------------------- LocalBundleAdjustment
struct BA {
  int P, L, E;
  std::vector<Pose> poses;
  std::vector<double> pts;
  const uint8_t* fixed;
  const int32_t *e_pt, *e_ps;
  const double *obs, *info;
  double fx, fy, cx, cy;
  std::vector<double> err;
  std::vector<uint8_t> level;
  bool robust = true;
  // active structure
  std::vector<int> act_edges, pose_h, pt_h;  // hessian index or -1
  int nPf = 0, nLa = 0;
  std::vector<int> pose_of_h, pt_of_h;
  // system
  std::vector<double> Hpp, Hll, bp, bl, B;  // Hpp [nPf][36], Hll [nLa][9], B per active edge [18] (6x3), zero for fixed pose

  void project_error(int e) {
    double Xc[3];
    pose_map(poses[e_ps[e]], &pts[3 * e_pt[e]], Xc);
    err[2 * e] = obs[2 * e] - (Xc[0] / Xc[2] * fx + cx);
    err[2 * e + 1] = obs[2 * e + 1] - (Xc[1] / Xc[2] * fy + cy);
  }
  double chi2(int e) const { return (err[2 * e] * err[2 * e] + err[2 * e + 1] * err[2 * e + 1]) * info[e]; }
  bool depth_positive(int e) const {
    double Xc[3];
    pose_map(poses[e_ps[e]], &pts[3 * e_pt[e]], Xc);
    return Xc[2] > 0.0;
  }
  // SparseOptimizer::initializeOptimization(level 0) + buildIndexMapping (sparse_optimizer.cpp:206-267,166-190)
  void init_active() {
    act_edges.clear();
    std::vector<uint8_t> pa(P, 0), la(L, 0);
    for (int e = 0; e < E; ++e)
      if (!level[e]) { act_edges.push_back(e); pa[e_ps[e]] = 1; la[e_pt[e]] = 1; }
    pose_h.assign(P, -1); pt_h.assign(L, -1);
    pose_of_h.clear(); pt_of_h.clear();
    for (int p = 0; p < P; ++p) if (pa[p] && !fixed[p]) { pose_h[p] = (int)pose_of_h.size(); pose_of_h.push_back(p); }
    for (int l = 0; l < L; ++l) if (la[l]) { pt_h[l] = (int)pt_of_h.size(); pt_of_h.push_back(l); }
    nPf = (int)pose_of_h.size(); nLa = (int)pt_of_h.size();
  }
  void compute_active_errors() { for (int e : act_edges) project_error(e); }
  double active_robust_chi2() const {
    double s = 0;
    for (int e : act_edges) { const double c = chi2(e); s += robust ? huber(c, kHuberDelta).r0 : c; }
    return s;
  }
  void build_system() {  // block_solver.hpp:502-560 + base_binary_edge.hpp:55-120
    Hpp.assign((size_t)nPf * 36, 0); Hll.assign((size_t)nLa * 9, 0);
    bp.assign((size_t)nPf * 6, 0); bl.assign((size_t)nLa * 3, 0);
    B.assign(act_edges.size() * 18, 0);
    for (size_t k = 0; k < act_edges.size(); ++k) {
      const int e = act_edges[k];
      const Pose& T = poses[e_ps[e]];
      double Xc[3], R[9], Jc[12], Jp[6];
      pose_map(T, &pts[3 * e_pt[e]], Xc);
      quat_to_rot(T.q, R);
      const double x = Xc[0], y = Xc[1], z = Xc[2];
      // _jacobianOplusXi = -1/z * tmp * R  (types_six_dof_expmap.cpp:124-134)
      const double tmp[6] = {fx, 0, -x / z * fx, 0, fy, -y / z * fy};
      for (int r = 0; r < 2; ++r)
        for (int c = 0; c < 3; ++c)
          Jp[r * 3 + c] = -1. / z * (tmp[r * 3] * R[c] + tmp[r * 3 + 1] * R[3 + c] + tmp[r * 3 + 2] * R[6 + c]);
      jac_pose(x, y, z, fx, fy, Jc);
      const double w = robust ? huber(chi2(e), kHuberDelta).r1 : 1.0;
      const double om = info[e] * w;
      const int hl = pt_h[e_pt[e]], hp = pose_h[e_ps[e]];
      const double e0 = err[2 * e], e1 = err[2 * e + 1];
      for (int r = 0; r < 3; ++r) {
        bl[hl * 3 + r] -= om * (Jp[r] * e0 + Jp[3 + r] * e1);
        for (int c = 0; c < 3; ++c) Hll[hl * 9 + r * 3 + c] += om * (Jp[r] * Jp[c] + Jp[3 + r] * Jp[3 + c]);
      }
      if (hp >= 0) {
        for (int r = 0; r < 6; ++r) {
          bp[hp * 6 + r] -= om * (Jc[r] * e0 + Jc[6 + r] * e1);
          for (int c = 0; c < 6; ++c) Hpp[hp * 36 + r * 6 + c] += om * (Jc[r] * Jc[c] + Jc[6 + r] * Jc[6 + c]);
          for (int c = 0; c < 3; ++c) B[k * 18 + r * 3 + c] = om * (Jc[r] * Jp[c] + Jc[6 + r] * Jp[3 + c]);
        }
      }
    }
  }
  // BlockSolver::solve with Schur (block_solver.hpp:354-486); x = [xp | xl]
  bool solve(double lambda, std::vector<double>& x) {
    const int n = 6 * nPf;
    x.assign((size_t)n + 3 * nLa, 0);
    std::vector<double> Dinv((size_t)nLa * 9), db((size_t)nLa * 3);
    for (int l = 0; l < nLa; ++l) {
      double D[9];
      for (int i = 0; i < 9; ++i) D[i] = Hll[l * 9 + i] + (i % 4 == 0 ? lambda : 0.0);
      const double c00 = D[4] * D[8] - D[5] * D[7], c01 = D[5] * D[6] - D[3] * D[8], c02 = D[3] * D[7] - D[4] * D[6];
      const double det = D[0] * c00 + D[1] * c01 + D[2] * c02, id = 1.0 / det;
      double* I = &Dinv[l * 9];
      I[0] = c00 * id; I[1] = (D[2] * D[7] - D[1] * D[8]) * id; I[2] = (D[1] * D[5] - D[2] * D[4]) * id;
      I[3] = c01 * id; I[4] = (D[0] * D[8] - D[2] * D[6]) * id; I[5] = (D[2] * D[3] - D[0] * D[5]) * id;
      I[6] = c02 * id; I[7] = (D[1] * D[6] - D[0] * D[7]) * id; I[8] = (D[0] * D[4] - D[1] * D[3]) * id;
      for (int r = 0; r < 3; ++r) db[l * 3 + r] = I[r * 3] * bl[l * 3] + I[r * 3 + 1] * bl[l * 3 + 1] + I[r * 3 + 2] * bl[l * 3 + 2];
    }
    if (n == 0) {  // no free pose: landmarks only
      for (int l = 0; l < nLa; ++l) for (int r = 0; r < 3; ++r) x[3 * l + r] = db[l * 3 + r];
      return true;
    }
    std::vector<double> S((size_t)n * n, 0), bs(bp);
    for (int p = 0; p < nPf; ++p)
      for (int r = 0; r < 6; ++r)
        for (int c = 0; c < 6; ++c) S[(size_t)(p * 6 + r) * n + p * 6 + c] = Hpp[p * 36 + r * 6 + c] + (r == c ? lambda : 0.0);
    // group active edges (with a free pose) by landmark
    std::vector<std::vector<int>> by_l(nLa);
    for (size_t k = 0; k < act_edges.size(); ++k)
      if (pose_h[e_ps[act_edges[k]]] >= 0) by_l[pt_h[e_pt[act_edges[k]]]].push_back((int)k);
    for (int l = 0; l < nLa; ++l) {
      const double* I = &Dinv[l * 9];
      for (int ki : by_l[l]) {
        const int pi = pose_h[e_ps[act_edges[ki]]];
        const double* Bi = &B[(size_t)ki * 18];
        double BD[18];
        for (int r = 0; r < 6; ++r)
          for (int c = 0; c < 3; ++c) BD[r * 3 + c] = Bi[r * 3] * I[c] + Bi[r * 3 + 1] * I[3 + c] + Bi[r * 3 + 2] * I[6 + c];
        for (int r = 0; r < 6; ++r) bs[pi * 6 + r] -= Bi[r * 3] * db[l * 3] + Bi[r * 3 + 1] * db[l * 3 + 1] + Bi[r * 3 + 2] * db[l * 3 + 2];
        for (int kj : by_l[l]) {
          const int pj = pose_h[e_ps[act_edges[kj]]];
          const double* Bj = &B[(size_t)kj * 18];
          for (int r = 0; r < 6; ++r)
            for (int c = 0; c < 6; ++c)
              S[(size_t)(pi * 6 + r) * n + pj * 6 + c] -= BD[r * 3] * Bj[c * 3] + BD[r * 3 + 1] * Bj[c * 3 + 1] + BD[r * 3 + 2] * Bj[c * 3 + 2];
        }
      }
    }
    if (!chol_solve(S, n, bs.data(), x.data())) return false;
    // xl = Dinv * (bl - B^T xp)
    std::vector<double> cl(bl);
    for (size_t k = 0; k < act_edges.size(); ++k) {
      const int pi = pose_h[e_ps[act_edges[k]]];
      if (pi < 0) continue;
      const int l = pt_h[e_pt[act_edges[k]]];
      const double* Bk = &B[k * 18];
      for (int c = 0; c < 3; ++c)
        for (int r = 0; r < 6; ++r) cl[l * 3 + c] -= Bk[r * 3 + c] * x[pi * 6 + r];
    }
    for (int l = 0; l < nLa; ++l) {
      const double* I = &Dinv[l * 9];
      for (int r = 0; r < 3; ++r) x[n + 3 * l + r] = I[r * 3] * cl[l * 3] + I[r * 3 + 1] * cl[l * 3 + 1] + I[r * 3 + 2] * cl[l * 3 + 2];
    }
    return true;
  }
  int optimize(int iterations, double* final_chi) {
    init_active();
    double lambda = -1, ni = 2;
    int nBad = 0, done = 0;
    std::vector<double> x;
    for (int it = 0; it < iterations; ++it) {
      compute_active_errors();
      double currentChi = active_robust_chi2(), tempChi = currentChi;
      const double iniChi = currentChi;
      build_system();
      if (it == 0) {
        double maxDiag = 0;
        for (int p = 0; p < nPf; ++p) for (int j = 0; j < 6; ++j) maxDiag = std::max(std::fabs(Hpp[p * 36 + j * 7]), maxDiag);
        for (int l = 0; l < nLa; ++l) for (int j = 0; j < 3; ++j) maxDiag = std::max(std::fabs(Hll[l * 9 + j * 4]), maxDiag);
        lambda = 1e-5 * maxDiag;
        ni = 2; nBad = 0;
      }
      double rho = 0;
      int qmax = 0;
      do {
        const std::vector<Pose> bposes = poses;
        const std::vector<double> bpts = pts;
        const bool ok2 = solve(lambda, x);
        if (ok2) {
          for (int h = 0; h < nPf; ++h) poses[pose_of_h[h]] = pose_oplus(poses[pose_of_h[h]], &x[h * 6]);
          for (int h = 0; h < nLa; ++h) for (int r = 0; r < 3; ++r) pts[3 * pt_of_h[h] + r] += x[6 * nPf + 3 * h + r];
        }
        compute_active_errors();
        tempChi = active_robust_chi2();
        if (!ok2) tempChi = std::numeric_limits<double>::max();
        rho = currentChi - tempChi;
        double scale = 0;
        for (int j = 0; j < 6 * nPf; ++j) scale += x[j] * (lambda * x[j] + bp[j]);
        for (int j = 0; j < 3 * nLa; ++j) scale += x[6 * nPf + j] * (lambda * x[6 * nPf + j] + bl[j]);
        scale += 1e-3;
        rho /= scale;
        if (rho > 0 && std::isfinite(tempChi)) {
          double alpha = 1. - std::pow((2 * rho - 1), 3);
          alpha = std::min(alpha, 2. / 3.);
          lambda *= std::max(1. / 3., alpha);
          ni = 2;
          currentChi = tempChi;
        } else {
          lambda *= ni;
          ni *= 2;
          poses = bposes;
          pts = bpts;
        }
        qmax++;
      } while (rho < 0 && qmax < 10);
      ++done;
      if (final_chi) *final_chi = currentChi;
      if (qmax == 10 || rho == 0) break;
      if ((iniChi - currentChi) * 1e3 < iniChi) nBad++; else nBad = 0;
      if (nBad >= 3) break;
    }
    return done;
  }
};

}  // namespace

extern "C" {

int orc_pose_optimize(double* pose7, int n, const double* Xw, const double* obs, const double* inv_sigma2,
                      const double* K, uint8_t* outlier) {
  // Optimizer.cc:239-413
  int nInitialCorrespondences = n;
  for (int i = 0; i < n; ++i) outlier[i] = 0;
  if (nInitialCorrespondences < 3) return 0;
  PoseProblem P;
  P.n = n; P.Xw = Xw; P.obs = obs; P.info = inv_sigma2;
  P.fx = K[0]; P.fy = K[1]; P.cx = K[2]; P.cy = K[3];
  P.err.assign(2 * (size_t)n, 0); P.level.assign(n, 0);
  std::vector<uint8_t> robust_edge(n, 1);
  P.robust = true;
  const Pose T0 = pose_from7(pose7);
  Pose T = T0;
  int nBad = 0;
  for (int it = 0; it < 4; it++) {
    T = T0;  // vSE3->setEstimate(Converter::toSE3Quat(pFrame->mTcw)) : every round restarts from the input pose
    int nact = 0;
    for (int i = 0; i < n; ++i) nact += !P.level[i];
    if (nact > 0) pose_lm(P, T, 10);
    nBad = 0;
    for (int i = 0; i < n; i++) {
      if (outlier[i]) P.compute_error(T, i);
      const float chi2 = (float)P.chi2(i);
      if (chi2 > kChi2Mono) { outlier[i] = 1; P.level[i] = 1; nBad++; }
      else { outlier[i] = 0; P.level[i] = 0; }
    }
    if (it == 2) P.robust = false;  // e->setRobustKernel(0) on every edge
    if (n < 10) break;              // optimizer.edges().size() < 10
  }
  pose_to7(T, pose7);
  return nInitialCorrespondences - nBad;
}

int orc_local_ba(int n_poses, int n_points, int n_edges, double* poses, const uint8_t* fixed, double* points,
                 const int32_t* e_point, const int32_t* e_pose, const double* e_obs, const double* e_info,
                 const double* K, int its1, int its2, double* edge_chi2, uint8_t* edge_depth_pos,
                 uint8_t* edge_outlier1, orc_ba_out* out) {
  BA ba;
  ba.P = n_poses; ba.L = n_points; ba.E = n_edges;
  ba.poses.resize(n_poses);
  for (int p = 0; p < n_poses; ++p) ba.poses[p] = pose_from7(poses + 7 * p);
  ba.pts.assign(points, points + 3 * (size_t)n_points);
  ba.fixed = fixed; ba.e_pt = e_point; ba.e_ps = e_pose; ba.obs = e_obs; ba.info = e_info;
  ba.fx = K[0]; ba.fy = K[1]; ba.cx = K[2]; ba.cy = K[3];
  ba.err.assign(2 * (size_t)n_edges, 0);
  ba.level.assign(n_edges, 0);
  ba.robust = true;
  double chi = 0;
  out->iters_first = ba.optimize(its1, &chi);   // optimizer.initializeOptimization(); optimize(5)
  out->chi2_first = ba.active_robust_chi2();    // from the stored edge errors
  for (int e = 0; e < n_edges; ++e) {           // Optimizer.cc:616-631
    const bool bad = ba.chi2(e) > 5.991 || !ba.depth_positive(e);
    edge_outlier1[e] = bad;
    if (bad) ba.level[e] = 1;
  }
  ba.robust = false;                            // e->setRobustKernel(0)
  chi = 0;
  out->iters_second = ba.optimize(its2, &chi);  // initializeOptimization(0); optimize(10)
  out->chi2_second = ba.active_robust_chi2();
  for (int e = 0; e < n_edges; ++e) {           // Optimizer.cc:657-671 (level-1 edges keep their stale error)
    edge_chi2[e] = ba.chi2(e);
    edge_depth_pos[e] = ba.depth_positive(e);
  }
  for (int p = 0; p < n_poses; ++p) pose_to7(ba.poses[p], poses + 7 * p);
  memcpy(points, ba.pts.data(), 3 * (size_t)n_points * sizeof(double));
  return 0;
}

void orc_tcw_to_pose7(const float* T, double* p) {  // Converter::toSE3Quat (Converter.cc:37-47)
  double R[9];
  for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) R[i * 3 + j] = (double)T[i * 4 + j];
  Quat q = rot_to_quat(R);
  normalize_rotation(q);
  p[0] = q.x; p[1] = q.y; p[2] = q.z; p[3] = q.w;
  p[4] = T[3]; p[5] = T[7]; p[6] = T[11];
}
void orc_pose7_to_tcw(const double* p, float* T) {  // Converter::toCvMat(SE3Quat) (Converter.cc:57-71)
  double R[9];
  quat_to_rot(Quat{p[0], p[1], p[2], p[3]}, R);
  for (int i = 0; i < 3; ++i) {
    for (int j = 0; j < 3; ++j) T[i * 4 + j] = (float)R[i * 3 + j];
    T[i * 4 + 3] = (float)p[4 + i];
  }
  T[12] = T[13] = T[14] = 0.f;
  T[15] = 1.f;
}

}  // extern "C"
