// oracle/frontend.cpp -- CPU restatement of ORBextractor::ExtractDesc up to the patch gather.
// TEST INFRASTRUCTURE ONLY (see oracle.h).
//
// Reference: src/vslam/src/ORBextractor.cc
//   ctor tables            :452-512      ComputePyramid          :1251-1276
//   ComputeKeyPointsOctTree :813-904      DistributeOctTree       :587-811
//   DivideNode             :529-585      IC_Angle                :80-107
//   ExtractDesc            :1137-1249    computeSIFTDescriptors  :1099-1126 (gather only)
// The OpenCV calls inside (resize, FAST, GaussianBlur, fastAtan2, cvRound) are restated from
// the published OpenCV 3.2.0 algorithms (the version pinned by
// src/3rd_party/opencv3_catkin/CMakeLists.txt:25; its source archive is a missing blob):
//   imgproc/imgwarp.cpp  resize / HResizeLinear / VResizeLinear<uchar,int,short,...>
//   features2d/fast.cpp FAST_t<16>, fast_score.cpp cornerScore<16>
//   imgproc/smooth.cpp  getGaussianKernel / createGaussianKernels, filter.cpp fixed-point
//                        separable filter (8-bit kernels, FixedPtCastEx<int,uchar>)
//   core/mathfuncs.cpp  fastAtan2
// PARITY UNPINNED for these (no OpenCV in the container, no reference test vectors).
// Where OpenCV's SIMD column filter computes in float (SymmColumnVec_32s8u) the scalar,
// exact-integer definition is followed.
#include "oracle.h"

#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstring>
#include <list>
#include <vector>

namespace {

inline int cvRound(double v) { return (int)std::lrint(v); }
inline int cvFloor(double v) { return (int)std::floor(v); }
inline int cvCeil(double v) { return (int)std::ceil(v); }
inline short sat_short(float v) { int i = cvRound(v); return (short)std::min(std::max(i, -32768), 32767); }
inline int reflect101(int p, int len) {
  if (len == 1) return 0;
  while (p < 0 || p >= len) p = p < 0 ? -p : 2 * len - 2 - p;
  return p;
}

const int PATCH_SIZE = 31, HALF_PATCH_SIZE = 15, EDGE_THRESHOLD = 19;

struct Img {
  int w = 0, h = 0;
  std::vector<uint8_t> d;
  uint8_t* row(int y) { return d.data() + (size_t)y * w; }
  const uint8_t* row(int y) const { return d.data() + (size_t)y * w; }
};

struct KP { float x, y, size, angle, response; int octave; };

// ---------------- cv::resize 8UC1 INTER_LINEAR (imgwarp.cpp, 3.2.0) ----------------
void resize_linear(const uint8_t* src, int sw, int sh, int sstride, uint8_t* dst, int dw, int dh, int dstride) {
  const int COEF_BITS = 11, COEF_SCALE = 1 << COEF_BITS;
  const double inv_scale_x = (double)dw / sw, inv_scale_y = (double)dh / sh;
  const double scale_x = 1. / inv_scale_x, scale_y = 1. / inv_scale_y;
  std::vector<int> xofs(dw), yofs(dh);
  std::vector<short> ialpha(dw * 2), ibeta(dh * 2);
  for (int dx = 0; dx < dw; dx++) {
    float fx = (float)((dx + 0.5) * scale_x - 0.5);
    int sx = cvFloor(fx);
    fx -= sx;
    if (sx < 0) { fx = 0; sx = 0; }
    if (sx >= sw - 1) { fx = 0; sx = sw - 1; }
    xofs[dx] = sx;
    const float c0 = 1.f - fx, c1 = fx;
    ialpha[dx * 2] = sat_short(c0 * COEF_SCALE);
    ialpha[dx * 2 + 1] = sat_short(c1 * COEF_SCALE);
  }
  for (int dy = 0; dy < dh; dy++) {
    float fy = (float)((dy + 0.5) * scale_y - 0.5);
    int sy = cvFloor(fy);
    fy -= sy;
    yofs[dy] = sy;
    const float c0 = 1.f - fy, c1 = fy;
    ibeta[dy * 2] = sat_short(c0 * COEF_SCALE);
    ibeta[dy * 2 + 1] = sat_short(c1 * COEF_SCALE);
  }
  std::vector<int> r0(dw), r1(dw);
  auto hrow = [&](int sy, std::vector<int>& out) {
    sy = std::min(std::max(sy, 0), sh - 1);
    const uint8_t* S = src + (size_t)sy * sstride;
    for (int dx = 0; dx < dw; dx++) {
      const int sx = xofs[dx];
      const int s1 = sx + 1 < sw ? S[sx + 1] : 0;  // coefficient is 0 there
      out[dx] = S[sx] * ialpha[dx * 2] + s1 * ialpha[dx * 2 + 1];
    }
  };
  for (int dy = 0; dy < dh; dy++) {
    hrow(yofs[dy], r0);
    hrow(yofs[dy] + 1, r1);
    const int b0 = ibeta[dy * 2], b1 = ibeta[dy * 2 + 1];
    uint8_t* D = dst + (size_t)dy * dstride;
    for (int x = 0; x < dw; x++)
      D[x] = (uint8_t)((((b0 * (r0[x] >> 4)) >> 16) + ((b1 * (r1[x] >> 4)) >> 16) + 2) >> 2);
  }
}

// ---------------- cv::GaussianBlur 8U, 7x7, sigma 2, BORDER_REFLECT_101 ----------------
void gaussian_kernel7_fixed(int k[7]) {
  // getGaussianKernel(7, 2.0, CV_32F) then convertTo(CV_32S, 256) (filter.cpp createSeparableLinearFilter, bits=8)
  const int n = 7;
  const double sigmaX = 2.0, scale2X = -0.5 / (sigmaX * sigmaX);
  float cf[7];
  double sum = 0;
  for (int i = 0; i < n; i++) {
    const double x = i - (n - 1) * 0.5;
    const double t = std::exp(scale2X * x * x);
    cf[i] = (float)t;
    sum += cf[i];
  }
  sum = 1. / sum;
  for (int i = 0; i < n; i++) {
    cf[i] = (float)(cf[i] * sum);
    k[i] = cvRound(cf[i] * 256.f);
  }
}

void gaussian_blur7(const uint8_t* src, int w, int h, int sstride, uint8_t* dst, int dstride) {
  int k[7];
  gaussian_kernel7_fixed(k);
  std::vector<int> tmp((size_t)w * h);
  for (int y = 0; y < h; y++) {
    const uint8_t* S = src + (size_t)y * sstride;
    for (int x = 0; x < w; x++) {
      int s = 0;
      for (int i = 0; i < 7; i++) s += k[i] * S[reflect101(x + i - 3, w)];
      tmp[(size_t)y * w + x] = s;
    }
  }
  for (int y = 0; y < h; y++)
    for (int x = 0; x < w; x++) {
      int s = 0;
      for (int i = 0; i < 7; i++) s += k[i] * tmp[(size_t)reflect101(y + i - 3, h) * w + x];
      const int v = (s + (1 << 15)) >> 16;  // FixedPtCastEx<int, uchar>(16)
      dst[(size_t)y * dstride + x] = (uint8_t)std::min(std::max(v, 0), 255);
    }
}

// ---------------- cv::FAST TYPE_9_16 (fast.cpp / fast_score.cpp) ----------------
const int kOff16[16][2] = {{0, 3},  {1, 3},   {2, 2},   {3, 1},   {3, 0},  {3, -1}, {2, -2}, {1, -3},
                           {0, -3}, {-1, -3}, {-2, -2}, {-3, -1}, {-3, 0}, {-3, 1}, {-2, 2}, {-1, 3}};

void make_offsets(int pixel[25], int stride) {
  int k = 0;
  for (; k < 16; k++) pixel[k] = kOff16[k][0] + kOff16[k][1] * stride;
  for (; k < 25; k++) pixel[k] = pixel[k - 16];
}

int corner_score16(const uint8_t* ptr, const int pixel[], int threshold) {
  const int K = 8, N = K * 3 + 1;
  int k, v = ptr[0];
  short d[N];
  for (k = 0; k < N; k++) d[k] = (short)(v - ptr[pixel[k]]);
  int a0 = threshold;
  for (k = 0; k < 16; k += 2) {
    int a = std::min((int)d[k + 1], (int)d[k + 2]);
    a = std::min(a, (int)d[k + 3]);
    if (a <= a0) continue;
    a = std::min(a, (int)d[k + 4]);
    a = std::min(a, (int)d[k + 5]);
    a = std::min(a, (int)d[k + 6]);
    a = std::min(a, (int)d[k + 7]);
    a = std::min(a, (int)d[k + 8]);
    a0 = std::max(a0, std::min(a, (int)d[k]));
    a0 = std::max(a0, std::min(a, (int)d[k + 9]));
  }
  int b0 = -a0;
  for (k = 0; k < 16; k += 2) {
    int b = std::max((int)d[k + 1], (int)d[k + 2]);
    b = std::max(b, (int)d[k + 3]);
    b = std::max(b, (int)d[k + 4]);
    b = std::max(b, (int)d[k + 5]);
    if (b >= b0) continue;
    b = std::max(b, (int)d[k + 6]);
    b = std::max(b, (int)d[k + 7]);
    b = std::max(b, (int)d[k + 8]);
    b0 = std::min(b0, std::max(b, (int)d[k]));
    b0 = std::min(b0, std::max(b, (int)d[k + 9]));
  }
  threshold = -b0 - 1;
  return threshold;
}

bool is_corner16(const uint8_t* ptr, const int pixel[], int threshold) {
  const int K = 8, N = 25;
  const int v = ptr[0];
  {
    const int vt = v - threshold;
    int count = 0;
    for (int k = 0; k < N; k++) {
      if (ptr[pixel[k]] < vt) { if (++count > K) return true; } else count = 0;
    }
  }
  {
    const int vt = v + threshold;
    int count = 0;
    for (int k = 0; k < N; k++) {
      if (ptr[pixel[k]] > vt) { if (++count > K) return true; } else count = 0;
    }
  }
  return false;
}

// FAST_t<16>(img, keypoints, threshold, nonmax_suppression = true)
void fast_detect(const uint8_t* img, int cols, int rows, int stride, int threshold, std::vector<KP>& out) {
  out.clear();
  if (cols < 7 || rows < 7) return;
  int pixel[25];
  make_offsets(pixel, stride);
  threshold = std::min(std::max(threshold, 0), 255);
  std::vector<uint8_t> bufv((size_t)cols * 3, 0);
  uint8_t* buf[3] = {bufv.data(), bufv.data() + cols, bufv.data() + 2 * cols};
  std::vector<int> cpv((size_t)(cols + 1) * 3, 0);
  int* cpbuf[3] = {cpv.data() + 1, cpv.data() + (cols + 1) + 1, cpv.data() + 2 * (cols + 1) + 1};
  for (int i = 3; i < rows - 2; i++) {
    const uint8_t* ptr = img + (size_t)i * stride + 3;
    uint8_t* curr = buf[(i - 3) % 3];
    int* cornerpos = cpbuf[(i - 3) % 3];
    memset(curr, 0, cols);
    int ncorners = 0;
    if (i < rows - 3) {
      for (int j = 3; j < cols - 3; j++, ptr++) {
        if (is_corner16(ptr, pixel, threshold)) {
          cornerpos[ncorners++] = j;
          curr[j] = (uint8_t)corner_score16(ptr, pixel, threshold);
        }
      }
    }
    cornerpos[-1] = ncorners;
    if (i == 3) continue;
    const uint8_t* prev = buf[(i - 4 + 3) % 3];
    const uint8_t* pprev = buf[(i - 5 + 3) % 3];
    cornerpos = cpbuf[(i - 4 + 3) % 3];
    ncorners = cornerpos[-1];
    for (int k = 0; k < ncorners; k++) {
      const int j = cornerpos[k];
      const int score = prev[j];
      if (score > prev[j + 1] && score > prev[j - 1] && score > pprev[j - 1] && score > pprev[j] &&
          score > pprev[j + 1] && score > curr[j - 1] && score > curr[j] && score > curr[j + 1]) {
        out.push_back(KP{(float)j, (float)(i - 1), 7.f, -1.f, (float)score, 0});
      }
    }
  }
}

// ---------------- cv::fastAtan2 (scalar, 3.2.0 mathfuncs.cpp atanImpl<float>) ----------------
const float atan2_p1 = 0.9997878412794807f * (float)(180 / M_PI);
const float atan2_p3 = -0.3258083974640975f * (float)(180 / M_PI);
const float atan2_p5 = 0.1555786518463281f * (float)(180 / M_PI);
const float atan2_p7 = -0.04432655554792128f * (float)(180 / M_PI);

float fast_atan2(float y, float x) {
  float ax = std::abs(x), ay = std::abs(y);
  float a, c, c2;
  if (ax >= ay) {
    c = ay / (ax + (float)DBL_EPSILON);
    c2 = c * c;
    a = (((atan2_p7 * c2 + atan2_p5) * c2 + atan2_p3) * c2 + atan2_p1) * c;
  } else {
    c = ax / (ay + (float)DBL_EPSILON);
    c2 = c * c;
    a = 90.f - (((atan2_p7 * c2 + atan2_p5) * c2 + atan2_p3) * c2 + atan2_p1) * c;
  }
  if (x < 0) a = 180.f - a;
  if (y < 0) a = 360.f - a;
  return a;
}

// IC_Angle, ORBextractor.cc:80-107 (image = un-blurred level, no border needed: kp >= 19 px inside)
float ic_angle(const uint8_t* img, int step, int px, int py, const int* u_max) {
  int m_01 = 0, m_10 = 0;
  const uint8_t* center = img + (size_t)py * step + px;
  for (int u = -HALF_PATCH_SIZE; u <= HALF_PATCH_SIZE; ++u) m_10 += u * center[u];
  for (int v = 1; v <= HALF_PATCH_SIZE; ++v) {
    int v_sum = 0;
    const int d = u_max[v];
    for (int u = -d; u <= d; ++u) {
      const int val_plus = center[u + v * step], val_minus = center[u - v * step];
      v_sum += (val_plus - val_minus);
      m_10 += u * (val_plus + val_minus);
    }
    m_01 += v * v_sum;
  }
  return fast_atan2((float)m_01, (float)m_10);
}

// ---------------- quadtree (ORBextractor.cc:529-811) ----------------
struct Pt2i { int x = 0, y = 0; };
struct Node {
  std::vector<KP> vKeys;
  Pt2i UL, UR, BL, BR;
  std::list<Node>::iterator lit;
  bool bNoMore = false;
  long seq = 0;  // creation order: stands in for the pointer value in the reference's sort tie-break
  void Divide(Node& n1, Node& n2, Node& n3, Node& n4) const {
    const int halfX = (int)std::ceil(static_cast<float>(UR.x - UL.x) / 2);
    const int halfY = (int)std::ceil(static_cast<float>(BR.y - UL.y) / 2);
    n1.UL = UL; n1.UR = {UL.x + halfX, UL.y}; n1.BL = {UL.x, UL.y + halfY}; n1.BR = {UL.x + halfX, UL.y + halfY};
    n2.UL = n1.UR; n2.UR = UR; n2.BL = n1.BR; n2.BR = {UR.x, UL.y + halfY};
    n3.UL = n1.BL; n3.UR = n1.BR; n3.BL = BL; n3.BR = {n1.BR.x, BL.y};
    n4.UL = n3.UR; n4.UR = n2.BR; n4.BL = n3.BR; n4.BR = BR;
    for (size_t i = 0; i < vKeys.size(); i++) {
      const KP& kp = vKeys[i];
      if (kp.x < n1.UR.x) {
        if (kp.y < n1.BR.y) n1.vKeys.push_back(kp); else n3.vKeys.push_back(kp);
      } else if (kp.y < n1.BR.y) n2.vKeys.push_back(kp);
      else n4.vKeys.push_back(kp);
    }
    if (n1.vKeys.size() == 1) n1.bNoMore = true;
    if (n2.vKeys.size() == 1) n2.bNoMore = true;
    if (n3.vKeys.size() == 1) n3.bNoMore = true;
    if (n4.vKeys.size() == 1) n4.bNoMore = true;
  }
};

std::vector<KP> distribute_octtree(const std::vector<KP>& keys, int minX, int maxX, int minY, int maxY, int N) {
  std::vector<KP> result;
  const int nIni = (int)std::round(static_cast<float>(maxX - minX) / (maxY - minY));
  if (nIni < 1) return result;  // (reference would divide by zero; never happens at supported sizes)
  const float hX = static_cast<float>(maxX - minX) / nIni;
  std::list<Node> lNodes;
  std::vector<Node*> vpIniNodes(nIni);
  long seq = 0;
  for (int i = 0; i < nIni; i++) {
    Node ni;
    ni.UL = {(int)(hX * static_cast<float>(i)), 0};
    ni.UR = {(int)(hX * static_cast<float>(i + 1)), 0};
    ni.BL = {ni.UL.x, maxY - minY};
    ni.BR = {ni.UR.x, maxY - minY};
    ni.seq = seq++;
    lNodes.push_back(ni);
    vpIniNodes[i] = &lNodes.back();
  }
  for (size_t i = 0; i < keys.size(); i++) {
    const KP& kp = keys[i];
    int idx = (int)(kp.x / hX);
    if (idx >= nIni) idx = nIni - 1;  // guard (reference indexes out of bounds here; cannot happen for x < maxX-minX)
    vpIniNodes[idx]->vKeys.push_back(kp);
  }
  auto lit = lNodes.begin();
  while (lit != lNodes.end()) {
    if (lit->vKeys.size() == 1) { lit->bNoMore = true; lit++; }
    else if (lit->vKeys.empty()) lit = lNodes.erase(lit);
    else lit++;
  }
  bool bFinish = false;
  typedef std::pair<int, Node*> SP;
  auto sp_less = [](const SP& a, const SP& b) { return a.first != b.first ? a.first < b.first : a.second->seq < b.second->seq; };
  std::vector<SP> vSizeAndPointerToNode;
  auto add_child = [&](Node& n, int* nToExpand) {
    if (n.vKeys.size() > 0) {
      n.seq = seq++;
      lNodes.push_front(n);
      if (n.vKeys.size() > 1) {
        if (nToExpand) (*nToExpand)++;
        vSizeAndPointerToNode.push_back(std::make_pair((int)n.vKeys.size(), &lNodes.front()));
        lNodes.front().lit = lNodes.begin();
      }
    }
  };
  while (!bFinish) {
    int prevSize = (int)lNodes.size();
    lit = lNodes.begin();
    int nToExpand = 0;
    vSizeAndPointerToNode.clear();
    while (lit != lNodes.end()) {
      if (lit->bNoMore) { lit++; continue; }
      Node n1, n2, n3, n4;
      lit->Divide(n1, n2, n3, n4);
      add_child(n1, &nToExpand);
      add_child(n2, &nToExpand);
      add_child(n3, &nToExpand);
      add_child(n4, &nToExpand);
      lit = lNodes.erase(lit);
    }
    if ((int)lNodes.size() >= N || (int)lNodes.size() == prevSize) {
      bFinish = true;
    } else if (((int)lNodes.size() + nToExpand * 3) > N) {
      while (!bFinish) {
        prevSize = (int)lNodes.size();
        std::vector<SP> vPrev = vSizeAndPointerToNode;
        vSizeAndPointerToNode.clear();
        std::sort(vPrev.begin(), vPrev.end(), sp_less);
        for (int j = (int)vPrev.size() - 1; j >= 0; j--) {
          Node n1, n2, n3, n4;
          vPrev[j].second->Divide(n1, n2, n3, n4);
          add_child(n1, nullptr);
          add_child(n2, nullptr);
          add_child(n3, nullptr);
          add_child(n4, nullptr);
          lNodes.erase(vPrev[j].second->lit);
          if ((int)lNodes.size() >= N) break;
        }
        if ((int)lNodes.size() >= N || (int)lNodes.size() == prevSize) bFinish = true;
      }
    }
  }
  for (auto it = lNodes.begin(); it != lNodes.end(); it++) {
    const std::vector<KP>& v = it->vKeys;
    const KP* p = &v[0];
    float maxResponse = p->response;
    for (size_t k = 1; k < v.size(); k++)
      if (v[k].response > maxResponse) { p = &v[k]; maxResponse = v[k].response; }
    result.push_back(*p);
  }
  return result;
}

}  // namespace

struct orc_extractor {
  int nfeatures, nlevels, iniTh, minTh;
  double scaleFactor;
  std::vector<float> scale, sigma2, inv_scale, inv_sigma2;
  std::vector<int> nfeat, umax;
  std::vector<Img> pyr, blur;
  std::vector<std::vector<KP>> raw;
};

extern "C" {

orc_extractor* orc_extractor_create(int nfeatures, float scale_factor, int nlevels, int ini_th, int min_th) {
  orc_extractor* e = new orc_extractor();
  e->nfeatures = nfeatures; e->nlevels = nlevels; e->iniTh = ini_th; e->minTh = min_th;
  e->scaleFactor = scale_factor;  // member is a double initialised from the float argument (ORBextractor.h:102)
  e->scale.resize(nlevels); e->sigma2.resize(nlevels); e->inv_scale.resize(nlevels); e->inv_sigma2.resize(nlevels);
  e->scale[0] = 1.0f; e->sigma2[0] = 1.0f;
  for (int i = 1; i < nlevels; i++) {
    e->scale[i] = (float)(e->scale[i - 1] * e->scaleFactor);
    e->sigma2[i] = e->scale[i] * e->scale[i];
  }
  for (int i = 0; i < nlevels; i++) { e->inv_scale[i] = 1.0f / e->scale[i]; e->inv_sigma2[i] = 1.0f / e->sigma2[i]; }
  e->nfeat.resize(nlevels);
  float factor = (float)(1.0f / e->scaleFactor);
  float nDesired = nfeatures * (1 - factor) / (1 - (float)std::pow((double)factor, (double)nlevels));
  int sum = 0;
  for (int level = 0; level < nlevels - 1; level++) {
    e->nfeat[level] = cvRound(nDesired);
    sum += e->nfeat[level];
    nDesired *= factor;
  }
  e->nfeat[nlevels - 1] = std::max(nfeatures - sum, 0);
  e->umax.resize(HALF_PATCH_SIZE + 1);
  int v, v0, vmax = cvFloor(HALF_PATCH_SIZE * std::sqrt(2.f) / 2 + 1);
  int vmin = cvCeil(HALF_PATCH_SIZE * std::sqrt(2.f) / 2);
  const double hp2 = HALF_PATCH_SIZE * HALF_PATCH_SIZE;
  for (v = 0; v <= vmax; ++v) e->umax[v] = cvRound(std::sqrt(hp2 - v * v));
  for (v = HALF_PATCH_SIZE, v0 = 0; v >= vmin; --v) {
    while (e->umax[v0] == e->umax[v0 + 1]) ++v0;
    e->umax[v] = v0;
    ++v0;
  }
  return e;
}

void orc_extractor_destroy(orc_extractor* e) { delete e; }

void orc_extractor_tables(const orc_extractor* e, float* scale, float* inv_scale, float* sigma2, float* inv_sigma2,
                          int32_t* fpl, int32_t* umax16) {
  for (int i = 0; i < e->nlevels; ++i) {
    if (scale) scale[i] = e->scale[i];
    if (inv_scale) inv_scale[i] = e->inv_scale[i];
    if (sigma2) sigma2[i] = e->sigma2[i];
    if (inv_sigma2) inv_sigma2[i] = e->inv_sigma2[i];
    if (fpl) fpl[i] = e->nfeat[i];
  }
  if (umax16) for (int i = 0; i < 16; ++i) umax16[i] = e->umax[i];
}

int orc_extract_keypoints(orc_extractor* e, const uint8_t* image, int width, int height, int stride,
                          orc_keypoint* kps_out, uint8_t* patches, int cap) {
  const int nl = e->nlevels;
  // ---- ComputePyramid (:1251-1276); borders are never read by the mono path, so they are not materialised
  e->pyr.assign(nl, Img());
  e->blur.assign(nl, Img());
  for (int level = 0; level < nl; ++level) {
    const float scale = e->inv_scale[level];
    Img& im = e->pyr[level];
    im.w = cvRound((float)width * scale);
    im.h = cvRound((float)height * scale);
    im.d.resize((size_t)im.w * im.h);
    if (level == 0) {
      for (int y = 0; y < height; ++y) memcpy(im.row(y), image + (size_t)y * stride, width);
    } else {
      const Img& p = e->pyr[level - 1];
      resize_linear(p.d.data(), p.w, p.h, p.w, im.d.data(), im.w, im.h, im.w);
    }
  }
  // ---- ComputeKeyPointsOctTree (:813-904)
  std::vector<std::vector<KP>> all(nl);
  e->raw.assign(nl, std::vector<KP>());
  const float W = 30;
  for (int level = 0; level < nl; ++level) {
    const Img& im = e->pyr[level];
    const int minBorderX = EDGE_THRESHOLD - 3, minBorderY = minBorderX;
    const int maxBorderX = im.w - EDGE_THRESHOLD + 3, maxBorderY = im.h - EDGE_THRESHOLD + 3;
    std::vector<KP> vToDistributeKeys;
    const float width_f = (float)(maxBorderX - minBorderX), height_f = (float)(maxBorderY - minBorderY);
    const int nCols = (int)(width_f / W), nRows = (int)(height_f / W);
    if (nCols < 1 || nRows < 1) continue;
    const int wCell = (int)std::ceil(width_f / nCols), hCell = (int)std::ceil(height_f / nRows);
    for (int i = 0; i < nRows; i++) {
      const float iniY = (float)(minBorderY + i * hCell);
      float maxY = iniY + hCell + 6;
      if (iniY >= maxBorderY - 3) continue;
      if (maxY > maxBorderY) maxY = (float)maxBorderY;
      for (int j = 0; j < nCols; j++) {
        const float iniX = (float)(minBorderX + j * wCell);
        float maxX = iniX + wCell + 6;
        if (iniX >= maxBorderX - 6) continue;
        if (maxX > maxBorderX) maxX = (float)maxBorderX;
        std::vector<KP> cell;
        const uint8_t* sub = im.row((int)iniY) + (int)iniX;
        const int cw = (int)maxX - (int)iniX, ch = (int)maxY - (int)iniY;
        fast_detect(sub, cw, ch, im.w, e->iniTh, cell);
        if (cell.empty()) fast_detect(sub, cw, ch, im.w, e->minTh, cell);
        for (auto& kp : cell) {
          kp.x += j * wCell;
          kp.y += i * hCell;
          vToDistributeKeys.push_back(kp);
        }
      }
    }
    e->raw[level] = vToDistributeKeys;
    std::vector<KP>& keypoints = all[level];
    keypoints = distribute_octtree(vToDistributeKeys, minBorderX, maxBorderX, minBorderY, maxBorderY, e->nfeat[level]);
    const int scaledPatchSize = (int)(PATCH_SIZE * e->scale[level]);
    for (auto& kp : keypoints) {
      kp.x += minBorderX;
      kp.y += minBorderY;
      kp.octave = level;
      kp.size = (float)scaledPatchSize;
    }
  }
  for (int level = 0; level < nl; ++level) {
    const Img& im = e->pyr[level];
    for (auto& kp : all[level]) kp.angle = ic_angle(im.d.data(), im.w, cvRound(kp.x), cvRound(kp.y), e->umax.data());
  }
  // ---- ExtractDesc tail (:1196-1245): blur, patch gather, rescale, concatenate
  int n = 0;
  for (int level = 0; level < nl; ++level) {
    std::vector<KP>& keypoints = all[level];
    if (keypoints.empty()) continue;
    const Img& im = e->pyr[level];
    Img& bl = e->blur[level];
    bl.w = im.w; bl.h = im.h; bl.d.resize(im.d.size());
    gaussian_blur7(im.d.data(), im.w, im.h, im.w, bl.d.data(), bl.w);
    for (auto& kp : keypoints) {
      if (n >= cap) return n;
      const int x = cvRound(kp.x), y = cvRound(kp.y);
      // :1113 -- the reference silently skips a keypoint failing this test (rows then shift);
      // it cannot fire for FAST corners (x in [19, W-20]); treated as a hard error here.
      if (!(x - 16 > 0 && x + 16 < bl.w && y - 16 > 0 && y + 16 < bl.h)) return -1;
      if (patches)
        for (int r = 0; r < 32; ++r) memcpy(patches + (size_t)n * 1024 + r * 32, bl.row(y - 16 + r) + x - 16, 32);
      KP o = kp;
      if (level != 0) { const float s = e->scale[level]; o.x *= s; o.y *= s; }
      kps_out[n] = orc_keypoint{o.x, o.y, o.size, o.angle, o.response, o.octave};
      ++n;
    }
  }
  return n;
}

int orc_level_size(const orc_extractor* e, int level, int* w, int* h) {
  if (level < 0 || level >= (int)e->pyr.size()) return -1;
  *w = e->pyr[level].w; *h = e->pyr[level].h;
  return 0;
}
void orc_level_image(const orc_extractor* e, int level, int blurred, uint8_t* out) {
  const Img& im = blurred ? e->blur[level] : e->pyr[level];
  memcpy(out, im.d.data(), im.d.size());
}
int orc_raw_corners(const orc_extractor* e, int level, int cap, float* x, float* y, float* resp) {
  const auto& v = e->raw[level];
  const int n = std::min((int)v.size(), cap);
  for (int i = 0; i < n; ++i) { x[i] = v[i].x; y[i] = v[i].y; resp[i] = v[i].response; }
  return n;
}

void orc_resize_linear_u8(const uint8_t* src, int sw, int sh, int sstride, uint8_t* dst, int dw, int dh, int dstride) {
  resize_linear(src, sw, sh, sstride, dst, dw, dh, dstride);
}
void orc_gaussian_blur7_u8(const uint8_t* src, int w, int h, int sstride, uint8_t* dst, int dstride) {
  gaussian_blur7(src, w, h, sstride, dst, dstride);
}
int orc_fast_score(const uint8_t* img, int stride, int x, int y, int threshold) {
  int pixel[25];
  make_offsets(pixel, stride);
  const uint8_t* p = img + (size_t)y * stride + x;
  return is_corner16(p, pixel, threshold) ? corner_score16(p, pixel, threshold) : 0;
}
int orc_fast_detect(const uint8_t* img, int w, int h, int stride, int threshold, int cap, int32_t* xs, int32_t* ys,
                    int32_t* scores) {
  std::vector<KP> v;
  fast_detect(img, w, h, stride, threshold, v);
  const int n = std::min((int)v.size(), cap);
  for (int i = 0; i < n; ++i) { xs[i] = (int)v[i].x; ys[i] = (int)v[i].y; scores[i] = (int)v[i].response; }
  return n;
}
float orc_fast_atan2(float y, float x) { return fast_atan2(y, x); }
float orc_ic_angle(const uint8_t* img, int stride, int x, int y, const int32_t* umax16) {
  return ic_angle(img, stride, x, y, umax16);
}

}  // extern "C"
