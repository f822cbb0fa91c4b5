// oracle/asdnet.cpp -- CPU restatement of ASDNet.forward.  TEST INFRASTRUCTURE ONLY (see oracle.h).
//
// Follows /root/reference/ASDNet/ASDNet/ASDNet.py:
//   input_norm  :360-365  per-patch (x - mean) / (std_unbiased + 1e-7)
//   features    :334-356  7 x Conv2d(bias=False) + BatchNorm2d(affine=False) [+ ReLU]; Dropout is
//                         the identity in eval mode
//   forward     :367-370  flatten, L2Norm (Utils.py:15-22: x / sqrt(sum x^2 + 1e-10))
// and the caller's patch preparation ORBextractor.cc:1125 (u8 -> f32 * 1/255).
// Pinned by tests/golden/asdnet_golden.npz (outputs of the reference class itself).
#include "oracle.h"

#include <cmath>
#include <vector>

namespace {
struct LayerSpec { int cout, cin, k, stride, pad; bool relu; };
const LayerSpec kLayers[7] = {
    {32, 1, 3, 1, 1, true},   {32, 32, 3, 1, 1, true},   {64, 32, 3, 2, 1, true},  {64, 64, 3, 1, 1, true},
    {128, 64, 3, 2, 1, true}, {128, 128, 3, 1, 1, true}, {128, 128, 8, 1, 0, false},
};

// NCHW single-image conv + BN(eval, affine=False) + optional ReLU
void conv_bn(const LayerSpec& L, const float* w, const float* mean, const float* var, float eps,
             const std::vector<float>& in, int hin, std::vector<float>& out, int& hout) {
  hout = (hin + 2 * L.pad - L.k) / L.stride + 1;
  out.assign((size_t)L.cout * hout * hout, 0.f);
  for (int co = 0; co < L.cout; ++co) {
    const float inv = 1.0f / std::sqrt(var[co] + eps);
    for (int oy = 0; oy < hout; ++oy)
      for (int ox = 0; ox < hout; ++ox) {
        float acc = 0.f;
        for (int ci = 0; ci < L.cin; ++ci)
          for (int ky = 0; ky < L.k; ++ky) {
            const int iy = oy * L.stride - L.pad + ky;
            if (iy < 0 || iy >= hin) continue;
            for (int kx = 0; kx < L.k; ++kx) {
              const int ix = ox * L.stride - L.pad + kx;
              if (ix < 0 || ix >= hin) continue;
              acc += in[((size_t)ci * hin + iy) * hin + ix] * w[(((size_t)co * L.cin + ci) * L.k + ky) * L.k + kx];
            }
          }
        float v = (acc - mean[co]) * inv;
        if (L.relu && v < 0.f) v = 0.f;
        out[((size_t)co * hout + oy) * hout + ox] = v;
      }
  }
}
}  // namespace

extern "C" void orc_asdnet_forward(const float* const conv_w[7], const float* const bn_mean[7],
                                   const float* const bn_var[7], float bn_eps, const uint8_t* patches,
                                   int32_t n, float* desc, float* act_l6) {
  const float inv255 = (float)(1.0 / 255);  // ORBextractor.cc:1125 convertTo(CV_32F, 1.0/255)
#pragma omp parallel for schedule(dynamic, 4)
  for (int p = 0; p < n; ++p) {
    std::vector<float> a, b;
    const uint8_t* src = patches + (size_t)p * 1024;
    a.assign(1024, 0.f);
    double s = 0.0;
    for (int i = 0; i < 1024; ++i) { a[i] = (float)src[i] * inv255; s += a[i]; }
    const float mean = (float)(s / 1024.0);
    double ss = 0.0;
    for (int i = 0; i < 1024; ++i) { const double d = (double)a[i] - mean; ss += d * d; }
    const float sd = (float)std::sqrt(ss / 1023.0) + 1e-7f;  // torch.std is unbiased
    for (int i = 0; i < 1024; ++i) a[i] = (a[i] - mean) / sd;
    int h = 32;
    for (int l = 0; l < 7; ++l) {
      int ho;
      conv_bn(kLayers[l], conv_w[l], bn_mean[l], bn_var[l], bn_eps, a, h, b, ho);
      a.swap(b);
      h = ho;
      if (l == 5 && act_l6)
        for (size_t i = 0; i < a.size(); ++i) act_l6[(size_t)p * 8192 + i] = a[i];
    }
    double nn = 0.0;
    for (int i = 0; i < 128; ++i) nn += (double)a[i] * a[i];
    const float norm = std::sqrt((float)nn + 1e-10f);
    for (int i = 0; i < 128; ++i) desc[(size_t)p * 128 + i] = a[i] / norm;
  }
}
