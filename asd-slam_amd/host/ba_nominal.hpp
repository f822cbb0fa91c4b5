// ba_nominal.hpp -- the nominal LocalBA problem of SURVEY 8(d) for host programs that have no map of their own (asd_replay --chain):
// 24 free + 12 fixed keyframes 0.8 m apart on a forward track, 6000 map points each seen by five consecutive keyframes, +-0.3 px
// observation noise, 2 % gross outliers, perturbed free poses and points; KITTI-like intrinsics supplied by the caller.  Same
// construction as asd-slam_amd/synth.py::ba_problem with a generator of its own (the numbers differ, the statistics do not).
#pragma once
#include <cmath>
#include <cstdint>
#include <vector>

#include "../../include/asd_slam.h"

namespace asd {

struct NominalBa {
  std::vector<double> poses, points, obs, info;
  std::vector<uint8_t> fixed;
  std::vector<int32_t> e_point, e_pose;
  asd_ba_problem problem{};
};

namespace ba_detail {
struct Rng {
  uint64_t s;
  explicit Rng(uint64_t seed) : s(seed * 6364136223846793005ull + 1442695040888963407ull) {}
  double uniform() { s = s * 6364136223846793005ull + 1442695040888963407ull; return (double)(s >> 11) * (1.0 / 9007199254740992.0); }
  double uniform(double a, double b) { return a + (b - a) * uniform(); }
  double normal() { const double u1 = uniform() + 1e-300, u2 = uniform(); return std::sqrt(-2.0 * std::log(u1)) * std::cos(6.283185307179586 * u2); }
};
inline void rodrigues(const double w[3], double R[9]) {
  const double th = std::sqrt(w[0] * w[0] + w[1] * w[1] + w[2] * w[2]);
  const double a = th < 1e-12 ? 1.0 : std::sin(th) / th, b = th < 1e-12 ? 0.5 : (1 - std::cos(th)) / (th * th);
  const double K[9] = {0, -w[2], w[1], w[2], 0, -w[0], -w[1], w[0], 0};
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) {
      double k2 = 0;
      for (int k = 0; k < 3; ++k) k2 += K[i * 3 + k] * K[k * 3 + j];
      R[i * 3 + j] = (i == j) + a * K[i * 3 + j] + b * k2;
    }
}
inline void rot_to_quat(const double R[9], double q[4]) {   // x y z w, w >= 0
  const double tr = R[0] + R[4] + R[8];
  double w = std::sqrt(std::fmax(0.0, 1 + tr)) / 2;
  if (w < 1e-6) w = 1e-6;
  q[0] = (R[7] - R[5]) / (4 * w); q[1] = (R[2] - R[6]) / (4 * w); q[2] = (R[3] - R[1]) / (4 * w); q[3] = w;
  const double n = std::sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
  for (int k = 0; k < 4; ++k) q[k] /= n;
}
}  // namespace ba_detail

inline void MakeNominalBa(NominalBa& B, const double K[4], int width, int height, uint64_t seed = 1, int n_free = 24, int n_fixed = 12,
                          int n_points = 6000, int obs_per_point = 5) {
  using namespace ba_detail;
  Rng rng(seed);
  const int P = n_free + n_fixed;
  std::vector<double> Rs((size_t)9 * P), ts((size_t)3 * P);
  for (int i = 0; i < P; ++i) {
    const double w[3] = {rng.normal() * 0.01, rng.normal() * 0.01, rng.normal() * 0.01};
    rodrigues(w, &Rs[9 * i]);
    const double c[3] = {rng.normal() * 0.05, rng.normal() * 0.02, 0.8 * i};
    for (int r = 0; r < 3; ++r) ts[3 * i + r] = -(Rs[9 * i + 3 * r] * c[0] + Rs[9 * i + 3 * r + 1] * c[1] + Rs[9 * i + 3 * r + 2] * c[2]);
  }
  B = NominalBa();
  int l = 0;
  for (long attempts = 0; l < n_points && attempts < 50L * n_points; ++attempts) {
    const int first = (int)(rng.uniform() * (P - obs_per_point + 1));
    const double X[3] = {rng.uniform(-10, 10), rng.uniform(-3, 3), 0.8 * first + rng.uniform(8, 28)};
    int pi[16]; double pu[16], pv[16]; int no = 0;
    for (int p = first; p < first + obs_per_point && no < 16; ++p) {
      double Xc[3];
      for (int r = 0; r < 3; ++r) Xc[r] = Rs[9 * p + 3 * r] * X[0] + Rs[9 * p + 3 * r + 1] * X[1] + Rs[9 * p + 3 * r + 2] * X[2] + ts[3 * p + r];
      if (Xc[2] < 1.0) break;
      const double u = K[0] * Xc[0] / Xc[2] + K[2], v = K[1] * Xc[1] / Xc[2] + K[3];
      if (!(u >= 0 && u < width && v >= 0 && v < height)) break;
      pi[no] = p; pu[no] = u; pv[no] = v; ++no;
    }
    if (no < 2) continue;
    for (int k = 0; k < no; ++k) {
      double du = rng.uniform(-0.3, 0.3), dv = rng.uniform(-0.3, 0.3);
      if (rng.uniform() < 0.02) { du = rng.uniform() < 0.5 ? -20.0 : 20.0; dv = rng.uniform() < 0.5 ? -20.0 : 20.0; }
      const int lvl = (int)(rng.uniform() * 8);
      B.e_point.push_back(l); B.e_pose.push_back(pi[k]);
      B.obs.push_back((double)(float)(pu[k] + du)); B.obs.push_back((double)(float)(pv[k] + dv));   // keypoints are f32 (cv::KeyPoint)
      const float s = std::pow(1.2f, (float)lvl);
      B.info.push_back((double)(1.0f / (s * s)));
    }
    for (int r = 0; r < 3; ++r) B.points.push_back(X[r] + rng.normal() * 0.05);
    ++l;
  }
  B.fixed.assign(P, 0);
  B.poses.resize((size_t)7 * P);
  for (int i = 0; i < P; ++i) {
    B.fixed[i] = i < n_fixed;
    double R[9], t[3];
    for (int k = 0; k < 9; ++k) R[k] = Rs[9 * i + k];
    for (int k = 0; k < 3; ++k) t[k] = ts[3 * i + k];
    if (!B.fixed[i]) {
      const double w[3] = {rng.normal() * 0.01, rng.normal() * 0.01, rng.normal() * 0.01};
      double D[9], R2[9];
      rodrigues(w, D);
      for (int a = 0; a < 3; ++a)
        for (int b = 0; b < 3; ++b) R2[a * 3 + b] = D[a * 3] * R[b] + D[a * 3 + 1] * R[3 + b] + D[a * 3 + 2] * R[6 + b];
      for (int k = 0; k < 9; ++k) R[k] = R2[k];
      for (int k = 0; k < 3; ++k) t[k] += rng.normal() * 0.01;
    }
    rot_to_quat(R, &B.poses[7 * i]);
    for (int k = 0; k < 3; ++k) B.poses[7 * i + 4 + k] = t[k];
  }
  asd_ba_problem& p = B.problem;
  p.n_poses = P; p.n_points = l; p.n_edges = (int32_t)B.e_point.size();
  p.poses = B.poses.data(); p.fixed = B.fixed.data(); p.points = B.points.data();
  p.e_point = B.e_point.data(); p.e_pose = B.e_pose.data(); p.e_obs = B.obs.data(); p.e_info = B.info.data();
  for (int k = 0; k < 4; ++k) p.K[k] = K[k];
  p.its_first = 5; p.its_second = 10;
}

}  // namespace asd
