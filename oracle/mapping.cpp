// oracle/mapping.cpp -- CPU restatement of LocalMapping::CreateNewMapPoints' per-match triangulation
// (src/vslam/src/LocalMapping.cc:386-519).  TEST INFRASTRUCTURE ONLY (see oracle.h).
//
// The reference leans on OpenCV 3.2.0 (absent third-party blob, src/3rd_party/opencv3_catkin) for
//   - MatExpr `s*row - row`  -> cv::addWeighted (float data, double weights)        [matop.cpp, arithm]
//   - cv::SVD::compute 4x4   -> JacobiSVDImpl_<float> (one-sided Jacobi, eps 2*FLT_EPSILON, <=30 sweeps)
//                               on the transposed matrix                             [lapack.cpp]
//   - Mat::dot / cv::norm    -> double accumulation
//   - Mat / scalar           -> convertTo with float scale 1/s
// restated here from the published 3.2.0 algorithm: PARITY UNPINNED for those pieces.
#include <cfloat>
#include <cmath>
#include <cstring>
#include <vector>

#include "oracle.h"

namespace {

// JacobiSVDImpl_<float>(At, W, Vt, m=4, n=4): rows of At are the columns of A; returns Vt sorted by
// descending singular value (the U-normalisation tail does not touch Vt and is omitted).
void jacobi_svd4(float At[4][4], float Vt[4][4]) {
  const int m = 4, n = 4;
  const float eps = FLT_EPSILON * 2;
  double W[4];
  for (int i = 0; i < n; i++) {
    double sd = 0;
    for (int k = 0; k < m; k++) { const float t = At[i][k]; sd += (double)t * t; }
    W[i] = sd;
    for (int k = 0; k < n; k++) Vt[i][k] = 0;
    Vt[i][i] = 1;
  }
  const int max_iter = 30;  // std::max(m, 30)
  for (int iter = 0; iter < max_iter; iter++) {
    bool changed = false;
    for (int i = 0; i < n - 1; i++)
      for (int j = i + 1; j < n; j++) {
        float *Ai = At[i], *Aj = At[j];
        double a = W[i], p = 0, b = W[j];
        for (int k = 0; k < m; k++) p += (double)Ai[k] * Aj[k];
        if (std::abs(p) <= eps * std::sqrt((double)a * b)) continue;
        p *= 2;
        const double beta = a - b, gamma = hypot((double)p, beta);
        float c, s;
        if (beta < 0) {
          const double delta = (gamma - beta) * 0.5;
          s = (float)std::sqrt(delta / gamma);
          c = (float)(p / (gamma * s * 2));
        } else {
          c = (float)std::sqrt((gamma + beta) / (gamma * 2));
          s = (float)(p / (gamma * c * 2));
        }
        a = b = 0;
        for (int k = 0; k < m; k++) {
          const float t0 = c * Ai[k] + s * Aj[k];
          const float t1 = -s * Ai[k] + c * Aj[k];
          Ai[k] = t0; Aj[k] = t1;
          a += (double)t0 * t0; b += (double)t1 * t1;
        }
        W[i] = a; W[j] = b;
        changed = true;
        float *Vi = Vt[i], *Vj = Vt[j];
        for (int k = 0; k < n; k++) {
          const float t0 = c * Vi[k] + s * Vj[k];
          const float t1 = -s * Vi[k] + c * Vj[k];
          Vi[k] = t0; Vj[k] = t1;
        }
      }
    if (!changed) break;
  }
  for (int i = 0; i < n; i++) {
    double sd = 0;
    for (int k = 0; k < m; k++) { const float t = At[i][k]; sd += (double)t * t; }
    W[i] = std::sqrt(sd);
  }
  for (int i = 0; i < n - 1; i++) {
    int j = i;
    for (int k = i + 1; k < n; k++)
      if (W[j] < W[k]) j = k;
    if (i != j) {
      std::swap(W[i], W[j]);
      for (int k = 0; k < m; k++) std::swap(At[i][k], At[j][k]);
      for (int k = 0; k < n; k++) std::swap(Vt[i][k], Vt[j][k]);
    }
  }
}

// (alpha*a - b) element as MatOp_AddEx::assign evaluates it: addWeighted<float, double>, or cv::subtract if alpha == 1
inline float scaled_minus(float alpha, float a, float b) {
  if (alpha == 1.0f) return a - b;
  return (float)((double)a * (double)alpha + (double)b * -1.0 + 0.0);
}
inline double dot3d(const float* a, const float* b) {
  double r = 0;
  for (int k = 0; k < 3; ++k) r += (double)a[k] * b[k];
  return r;
}
inline double norm3d(const float* a) { return std::sqrt(dot3d(a, a)); }
// KeyFrame::SetPose (KeyFrame.cc:63-80): Rwc = Rcw.t() (materialised), Ow = -Rwc*tcw
void camera_centre(const float* T, float Rwc[9], float Ow[3]) {
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) Rwc[i * 3 + j] = T[j * 4 + i];
  for (int i = 0; i < 3; ++i) {
    const float t0 = Rwc[i * 3 + 0] * T[3] + Rwc[i * 3 + 1] * T[7] + Rwc[i * 3 + 2] * T[11];
    Ow[i] = (float)((double)t0 * -1.0);
  }
}
inline void mat3_vec(const float* R, const float* x, float* out) {  // gemm small-matrix path, f32
  for (int i = 0; i < 3; ++i) out[i] = R[i * 3 + 0] * x[0] + R[i * 3 + 1] * x[1] + R[i * 3 + 2] * x[2];
}

}  // namespace

extern "C" {

void orc_svd4_vt(const float* A, float* vt) {
  float At[4][4], Vt[4][4];
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 4; ++j) At[i][j] = A[j * 4 + i];  // cv::transpose(src, temp_a)
  jacobi_svd4(At, Vt);
  memcpy(vt, Vt, sizeof Vt);
}

// LocalMapping.cc:386-519, monocular branch (bStereo1 = bStereo2 = false)
int orc_triangulate_pairs(const orc_keypoint* kps1, const orc_keypoint* kps2, int n_pairs, const int32_t* idx1,
                          const int32_t* idx2, const float* Tcw1, const float* Tcw2, const float* K1, const float* K2,
                          float scale_factor, int nlevels, float* x3D_out, uint8_t* ok) {
  std::vector<float> sf(nlevels), sigma2(nlevels);
  sf[0] = 1.0f; sigma2[0] = 1.0f;
  for (int i = 1; i < nlevels; i++) {  // ORBextractor.cc:464-468 (double factor, float table)
    sf[i] = (float)(sf[i - 1] * (double)scale_factor);
    sigma2[i] = sf[i] * sf[i];
  }
  const float fx1 = K1[0], fy1 = K1[1], cx1 = K1[2], cy1 = K1[3], invfx1 = 1.0f / fx1, invfy1 = 1.0f / fy1;
  const float fx2 = K2[0], fy2 = K2[1], cx2 = K2[2], cy2 = K2[3], invfx2 = 1.0f / fx2, invfy2 = 1.0f / fy2;
  float Rwc1[9], Rwc2[9], Ow1[3], Ow2[3];
  camera_centre(Tcw1, Rwc1, Ow1);
  camera_centre(Tcw2, Rwc2, Ow2);
  const float ratioFactor = 1.5f * scale_factor;
  int nnew = 0;
  for (int ikp = 0; ikp < n_pairs; ikp++) {
    ok[ikp] = 0;
    x3D_out[3 * ikp] = x3D_out[3 * ikp + 1] = x3D_out[3 * ikp + 2] = 0.f;
    const orc_keypoint& kp1 = kps1[idx1[ikp]];
    const orc_keypoint& kp2 = kps2[idx2[ikp]];
    const float xn1[3] = {(kp1.x - cx1) * invfx1, (kp1.y - cy1) * invfy1, 1.0f};
    const float xn2[3] = {(kp2.x - cx2) * invfx2, (kp2.y - cy2) * invfy2, 1.0f};
    float ray1[3], ray2[3];
    mat3_vec(Rwc1, xn1, ray1);
    mat3_vec(Rwc2, xn2, ray2);
    const float cosParallaxRays = (float)(dot3d(ray1, ray2) / (norm3d(ray1) * norm3d(ray2)));
    const float cosParallaxStereo = cosParallaxRays + 1;
    if (!(cosParallaxRays < cosParallaxStereo && cosParallaxRays > 0 && cosParallaxRays < 0.9998)) continue;
    float A[4][4];
    for (int k = 0; k < 4; ++k) {
      A[0][k] = scaled_minus(xn1[0], Tcw1[8 + k], Tcw1[0 + k]);
      A[1][k] = scaled_minus(xn1[1], Tcw1[8 + k], Tcw1[4 + k]);
      A[2][k] = scaled_minus(xn2[0], Tcw2[8 + k], Tcw2[0 + k]);
      A[3][k] = scaled_minus(xn2[1], Tcw2[8 + k], Tcw2[4 + k]);
    }
    float vt[16];
    orc_svd4_vt(&A[0][0], vt);
    const float* v = vt + 12;
    if (v[3] == 0) continue;
    const float inv = (float)(1.0 / (double)v[3]);  // Mat / double -> convertTo(alpha = 1/s), float scale
    const float x3D[3] = {v[0] * inv + 0.0f, v[1] * inv + 0.0f, v[2] * inv + 0.0f};
    const float z1 = (float)(dot3d(Tcw1 + 8, x3D) + Tcw1[11]);
    if (z1 <= 0) continue;
    const float z2 = (float)(dot3d(Tcw2 + 8, x3D) + Tcw2[11]);
    if (z2 <= 0) continue;
    const float sigmaSquare1 = sigma2[kp1.octave];
    const float x1 = (float)(dot3d(Tcw1 + 0, x3D) + Tcw1[3]);
    const float y1 = (float)(dot3d(Tcw1 + 4, x3D) + Tcw1[7]);
    const float invz1 = (float)(1.0 / z1);
    {
      const float u1 = fx1 * x1 * invz1 + cx1;
      const float v1 = fy1 * y1 * invz1 + cy1;
      const float errX1 = u1 - kp1.x;
      const float errY1 = v1 - kp1.y;
      if ((errX1 * errX1 + errY1 * errY1) > 5.991 * sigmaSquare1) continue;
    }
    const float sigmaSquare2 = sigma2[kp2.octave];
    const float x2 = (float)(dot3d(Tcw2 + 0, x3D) + Tcw2[3]);
    const float y2 = (float)(dot3d(Tcw2 + 4, x3D) + Tcw2[7]);
    const float invz2 = (float)(1.0 / z2);
    {
      const float u2 = fx2 * x2 * invz2 + cx2;
      const float v2 = fy2 * y2 * invz2 + cy2;
      const float errX2 = u2 - kp2.x;
      const float errY2 = v2 - kp2.y;
      if ((errX2 * errX2 + errY2 * errY2) > 5.991 * sigmaSquare2) continue;
    }
    const float n1[3] = {x3D[0] - Ow1[0], x3D[1] - Ow1[1], x3D[2] - Ow1[2]};
    const float dist1 = (float)norm3d(n1);
    const float n2[3] = {x3D[0] - Ow2[0], x3D[1] - Ow2[1], x3D[2] - Ow2[2]};
    const float dist2 = (float)norm3d(n2);
    if (dist1 == 0 || dist2 == 0) continue;
    const float ratioDist = dist2 / dist1;
    const float ratioOctave = sf[kp1.octave] / sf[kp2.octave];
    if (ratioDist * ratioFactor < ratioOctave || ratioDist > ratioOctave * ratioFactor) continue;
    ok[ikp] = 1;
    memcpy(x3D_out + 3 * ikp, x3D, 12);
    nnew++;
  }
  return nnew;
}

}  // extern "C"
