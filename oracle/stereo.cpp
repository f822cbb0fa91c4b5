// oracle/stereo.cpp -- CPU restatement of Frame::ComputeStereoMatches (src/vslam/src/Frame.cc:360-535).
// TEST INFRASTRUCTURE ONLY (see oracle.h).
//
// The function is dead code in the reference (no stereo Frame constructor survives), so there is nothing
// to run it against: PARITY UNPINNED, restated from the source text.  Kept quirks: `thOrbDist` is an int that
// truncates (TH_HIGH + TH_LOW) / 2 = 1.0 (:366); the row table is indexed with the truncated float row (:404); the SAD
// values are exact integers in float; the parabola fit may produce NaN, which passes the [-1, 1] test (:491).
// Guards added where the reference has undefined behaviour: row indices are clamped to the image, and an empty match
// list skips the median filter (:519 reads vDistIdx[0]).
#include <algorithm>
#include <climits>
#include <cmath>
#include <utility>
#include <vector>

#include "oracle.h"

extern "C" int orc_stereo_match(const orc_extractor* exL, const orc_extractor* exR, const orc_keypoint* kpsL, const float* descL, int N,
                                const orc_keypoint* kpsR, const float* descR, int Nr, float mb, float mbf, float* mvuRight,
                                float* mvDepth) {
  int nlevels = 0, w0 = 0, h0 = 0;
  while (orc_level_size(exL, nlevels, &w0, &h0) == 0) ++nlevels;
  std::vector<float> scale(nlevels), inv_scale(nlevels);
  orc_extractor_tables(exL, scale.data(), inv_scale.data(), nullptr, nullptr, nullptr, nullptr);
  struct Lvl { int w, h; std::vector<uint8_t> l, r; };
  std::vector<Lvl> pyr(nlevels);
  for (int l = 0; l < nlevels; ++l) {
    orc_level_size(exL, l, &pyr[l].w, &pyr[l].h);
    pyr[l].l.resize((size_t)pyr[l].w * pyr[l].h);
    pyr[l].r.resize((size_t)pyr[l].w * pyr[l].h);
    orc_level_image(exL, l, 0, pyr[l].l.data());
    orc_level_image(exR, l, 0, pyr[l].r.data());
  }
  for (int i = 0; i < N; ++i) { mvuRight[i] = -1.0f; mvDepth[i] = -1.0f; }
  const float TH_HIGH = 1.5f, TH_LOW = 0.5f;
  const int thOrbDist = (TH_HIGH + TH_LOW) / 2;
  const int nRows = pyr[0].h;
  std::vector<std::vector<size_t>> vRowIndices(nRows);
  for (int iR = 0; iR < Nr; iR++) {
    const float kpY = kpsR[iR].y;
    const float r = 2.0f * scale[kpsR[iR].octave];
    const int maxr = (int)std::ceil(kpY + r);
    const int minr = (int)std::floor(kpY - r);
    for (int yi = std::max(minr, 0); yi <= std::min(maxr, nRows - 1); yi++) vRowIndices[yi].push_back(iR);
  }
  const float minZ = mb;
  const float minD = 0;
  const float maxD = mbf / minZ;
  std::vector<std::pair<int, int>> vDistIdx;
  for (int iL = 0; iL < N; iL++) {
    const orc_keypoint& kpL = kpsL[iL];
    const int levelL = kpL.octave;
    const float vL = kpL.y, uL = kpL.x;
    const int row = (int)vL;
    if (row < 0 || row >= nRows) continue;
    const std::vector<size_t>& vCandidates = vRowIndices[row];
    if (vCandidates.empty()) continue;
    const float minU = uL - maxD;
    const float maxU = uL - minD;
    if (maxU < 0) continue;
    float bestDist = TH_HIGH;
    size_t bestIdxR = 0;
    const float* dL = descL + (size_t)iL * 128;
    for (size_t iC = 0; iC < vCandidates.size(); iC++) {
      const size_t iR = vCandidates[iC];
      const orc_keypoint& kpR = kpsR[iR];
      if (kpR.octave < levelL - 1 || kpR.octave > levelL + 1) continue;
      const float uR = kpR.x;
      if (uR >= minU && uR <= maxU) {
        const float dist = orc_descriptor_distance(dL, descR + iR * 128);
        if (dist < bestDist) { bestDist = dist; bestIdxR = iR; }
      }
    }
    if (bestDist < thOrbDist) {
      const float uR0 = kpsR[bestIdxR].x;
      const float scaleFactor = inv_scale[kpL.octave];
      const float scaleduL = std::round(kpL.x * scaleFactor);
      const float scaledvL = std::round(kpL.y * scaleFactor);
      const float scaleduR0 = std::round(uR0 * scaleFactor);
      const int w = 5;
      const Lvl& P = pyr[kpL.octave];
      auto patch = [&](const std::vector<uint8_t>& img, int cx, int cy, float* out) {  // 11x11, float, centre subtracted
        const float c = (float)img[(size_t)cy * P.w + cx];
        for (int dy = -w; dy <= w; ++dy)
          for (int dx = -w; dx <= w; ++dx) out[(dy + w) * 11 + dx + w] = (float)img[(size_t)(cy + dy) * P.w + cx + dx] - c * 1.0f;
      };
      float IL[121], IR[121];
      patch(P.l, (int)scaleduL, (int)scaledvL, IL);
      float bestDistS = (float)INT_MAX;
      int bestincR = 0;
      const int L = 5;
      std::vector<float> vDists(2 * L + 1);
      const float iniu = scaleduR0 + L - w;
      const float endu = scaleduR0 + L + w + 1;
      if (iniu < 0 || endu >= P.w) continue;
      for (int incR = -L; incR <= +L; incR++) {
        patch(P.r, (int)scaleduR0 + incR, (int)scaledvL, IR);
        double s = 0;
        for (int k = 0; k < 121; ++k) s += std::fabs((double)(IL[k] - IR[k]));  // cv::norm(IL, IR, NORM_L1)
        const float dist = (float)s;
        if (dist < bestDistS) { bestDistS = dist; bestincR = incR; }
        vDists[L + incR] = dist;
      }
      if (bestincR == -L || bestincR == L) continue;
      const float dist1 = vDists[L + bestincR - 1];
      const float dist2 = vDists[L + bestincR];
      const float dist3 = vDists[L + bestincR + 1];
      const float deltaR = (dist1 - dist3) / (2.0f * (dist1 + dist3 - 2.0f * dist2));
      if (deltaR < -1 || deltaR > 1) continue;
      float bestuR = scale[kpL.octave] * ((float)scaleduR0 + (float)bestincR + deltaR);
      float disparity = (uL - bestuR);
      if (disparity >= minD && disparity < maxD) {
        if (disparity <= 0) { disparity = 0.01; bestuR = uL - 0.01; }
        mvDepth[iL] = mbf / disparity;
        mvuRight[iL] = bestuR;
        vDistIdx.push_back(std::pair<int, int>(bestDistS, iL));
      }
    }
  }
  if (vDistIdx.empty()) return 0;
  std::sort(vDistIdx.begin(), vDistIdx.end());
  const float median = vDistIdx[vDistIdx.size() / 2].first;
  const float thDist = 1.5f * 1.4f * median;
  int kept = (int)vDistIdx.size();
  for (int i = (int)vDistIdx.size() - 1; i >= 0; i--) {
    if (vDistIdx[i].first < thDist) break;
    mvuRight[vDistIdx[i].second] = -1;
    mvDepth[vDistIdx[i].second] = -1;
    --kept;
  }
  return kept;
}
