// oracle/matcher.cpp -- CPU restatement of the Frame grid and the ORBmatcher searches on the
// per-frame path.  TEST INFRASTRUCTURE ONLY (see oracle.h).
//
// Reference (src/vslam/src):
//   Frame::AssignFeaturesToGrid / PosInGrid / GetFeaturesInArea   Frame.cc:123-138, 276-286, 219-274
//   Frame::isInFrustum                                            Frame.cc:160-217
//   MapPoint::PredictScale                                        MapPoint.cc:438-453
//   ORBmatcher::DescriptorDistance                                ORBmatcher.cc:1629-1650
//   ORBmatcher::SearchByProjection(Frame&, vector<MapPoint*>&)    ORBmatcher.cc:44-122
//   ORBmatcher::SearchByProjection(Frame&, const Frame&, ...)     ORBmatcher.cc:1318-1452
//   ORBmatcher::SearchForInitialization                           ORBmatcher.cc:416-531
//   ORBmatcher::ComputeThreeMaxima                                ORBmatcher.cc:1584-1625
//   MapPoint::ComputeDistinctiveDescriptors                       MapPoint.cc:271-338
// The source is in the tree but needs OpenCV types to compile, so it is restated as text:
// PARITY UNPINNED beyond self-consistency.  cv::Mat float products are restated as OpenCV 3.2.0
// computes them (gemm small-matrix path: left-to-right f32 sums; norm / dot / transposed gemm
// accumulate in double).
#include "oracle.h"

#include <algorithm>
#include <cmath>
#include <vector>

namespace {
const float TH_HIGH = 1.5f, TH_LOW = 0.5f;
const int HISTO_LENGTH = 30;
const int GRID_COLS = 64, GRID_ROWS = 48;

void ComputeThreeMaxima(std::vector<int>* histo, const int L, int& ind1, int& ind2, int& ind3) {
  int max1 = 0, max2 = 0, max3 = 0;
  for (int i = 0; i < L; i++) {
    const int s = (int)histo[i].size();
    if (s > max1) { max3 = max2; max2 = max1; max1 = s; ind3 = ind2; ind2 = ind1; ind1 = i; }
    else if (s > max2) { max3 = max2; max2 = s; ind3 = ind2; ind2 = i; }
    else if (s > max3) { max3 = s; ind3 = i; }
  }
  if (max2 < 0.1f * (float)max1) { ind2 = -1; ind3 = -1; }
  else if (max3 < 0.1f * (float)max1) { ind3 = -1; }
}

float DescriptorDistance(const float* a, const float* b) {
  float sqd = 0.;
  for (int i = 0; i < 128; i += 4) {
    sqd += (a[i] - b[i]) * (a[i] - b[i]);
    sqd += (a[i + 1] - b[i + 1]) * (a[i + 1] - b[i + 1]);
    sqd += (a[i + 2] - b[i + 2]) * (a[i + 2] - b[i + 2]);
    sqd += (a[i + 3] - b[i + 3]) * (a[i + 3] - b[i + 3]);
  }
  return sqd;
}

// Rcw*x + tcw as cv::gemm's 3x3 * 3x1 small-matrix path (A*B + C folded into one gemm)
inline void transform(const float* T, const float* X, float* out) {
  for (int r = 0; r < 3; ++r) {
    const float t0 = T[r * 4 + 0] * X[0] + T[r * 4 + 1] * X[1] + T[r * 4 + 2] * X[2];
    out[r] = (float)((double)t0 + (double)T[r * 4 + 3]);
  }
}
}  // namespace

struct orc_frame {
  int N;
  std::vector<orc_keypoint> kps;
  std::vector<float> desc;
  float mnMinX, mnMaxX, mnMinY, mnMaxY, mfGridElementWidthInv, mfGridElementHeightInv;
  std::vector<int> mGrid[GRID_COLS][GRID_ROWS];
  std::vector<float> mvScaleFactors;
  int mnScaleLevels;
  float mfLogScaleFactor;

  std::vector<size_t> GetFeaturesInArea(const float& x, const float& y, const float& r, const int minLevel,
                                        const int maxLevel) const {
    std::vector<size_t> vIndices;
    const int nMinCellX = std::max(0, (int)std::floor((x - mnMinX - r) * mfGridElementWidthInv));
    if (nMinCellX >= GRID_COLS) return vIndices;
    const int nMaxCellX = std::min((int)GRID_COLS - 1, (int)std::ceil((x - mnMinX + r) * mfGridElementWidthInv));
    if (nMaxCellX < 0) return vIndices;
    const int nMinCellY = std::max(0, (int)std::floor((y - mnMinY - r) * mfGridElementHeightInv));
    if (nMinCellY >= GRID_ROWS) return vIndices;
    const int nMaxCellY = std::min((int)GRID_ROWS - 1, (int)std::ceil((y - mnMinY + r) * mfGridElementHeightInv));
    if (nMaxCellY < 0) return vIndices;
    const bool bCheckLevels = (minLevel > 0) || (maxLevel >= 0);
    for (int ix = nMinCellX; ix <= nMaxCellX; ix++)
      for (int iy = nMinCellY; iy <= nMaxCellY; iy++) {
        const std::vector<int>& vCell = mGrid[ix][iy];
        for (size_t j = 0, jend = vCell.size(); j < jend; j++) {
          const orc_keypoint& kpUn = kps[vCell[j]];
          if (bCheckLevels) {
            if (kpUn.octave < minLevel) continue;
            if (maxLevel >= 0)
              if (kpUn.octave > maxLevel) continue;
          }
          const float distx = kpUn.x - x;
          const float disty = kpUn.y - y;
          if (std::fabs(distx) < r && std::fabs(disty) < r) vIndices.push_back(vCell[j]);
        }
      }
    return vIndices;
  }
};

extern "C" {

orc_frame* orc_frame_create(const orc_keypoint* kps, const float* desc, int n, float minx, float maxx, float miny,
                            float maxy, int nlevels, float scale_factor) {
  orc_frame* f = new orc_frame();
  f->N = n;
  f->kps.assign(kps, kps + n);
  f->desc.assign(desc, desc + (size_t)n * 128);
  f->mnMinX = minx; f->mnMaxX = maxx; f->mnMinY = miny; f->mnMaxY = maxy;
  f->mfGridElementWidthInv = static_cast<float>(GRID_COLS) / static_cast<float>(maxx - minx);   // Frame.cc:106
  f->mfGridElementHeightInv = static_cast<float>(GRID_ROWS) / static_cast<float>(maxy - miny);  // Frame.cc:107
  for (int i = 0; i < n; i++) {  // AssignFeaturesToGrid + PosInGrid
    const int posX = (int)std::round((kps[i].x - minx) * f->mfGridElementWidthInv);
    const int posY = (int)std::round((kps[i].y - miny) * f->mfGridElementHeightInv);
    if (posX < 0 || posX >= GRID_COLS || posY < 0 || posY >= GRID_ROWS) continue;
    f->mGrid[posX][posY].push_back(i);
  }
  // ORBextractor scale tables as Frame copies them (Frame.cc:72-79)
  f->mnScaleLevels = nlevels;
  f->mvScaleFactors.resize(nlevels);
  f->mvScaleFactors[0] = 1.0f;
  const double sf = scale_factor;
  for (int i = 1; i < nlevels; i++) f->mvScaleFactors[i] = (float)(f->mvScaleFactors[i - 1] * sf);
  f->mfLogScaleFactor = std::log((float)scale_factor);  // Frame.cc:74 log(mfScaleFactor), float
  return f;
}
void orc_frame_destroy(orc_frame* f) { delete f; }

int orc_features_in_area(const orc_frame* f, float x, float y, float r, int minlevel, int maxlevel, int cap,
                         int32_t* out) {
  const std::vector<size_t> v = f->GetFeaturesInArea(x, y, r, minlevel, maxlevel);
  const int n = std::min((int)v.size(), cap);
  for (int i = 0; i < n; ++i) out[i] = (int32_t)v[i];
  return n;
}

float orc_descriptor_distance(const float* a, const float* b) { return DescriptorDistance(a, b); }

void orc_dist_matrix(const float* a, int na, const float* b, int nb, float* out) {
  for (int i = 0; i < na; ++i)
    for (int j = 0; j < nb; ++j) out[(size_t)i * nb + j] = DescriptorDistance(a + (size_t)i * 128, b + (size_t)j * 128);
}

int orc_distinctive_descriptor(const float* desc, int n) {
  const size_t N = n;
  std::vector<float> Distances(N * N);
  for (size_t i = 0; i < N; i++) {
    Distances[i * N + i] = 0;
    for (size_t j = i + 1; j < N; j++) {
      const float distij = DescriptorDistance(desc + i * 128, desc + j * 128);
      Distances[i * N + j] = distij;
      Distances[j * N + i] = distij;
    }
  }
  float BestMedian = 100;
  int BestIdx = 0;
  for (size_t i = 0; i < N; i++) {
    std::vector<float> vDists(Distances.begin() + i * N, Distances.begin() + (i + 1) * N);
    std::sort(vDists.begin(), vDists.end());
    const float median = vDists[(size_t)(0.5 * (N - 1))];
    if (median < BestMedian) { BestMedian = median; BestIdx = (int)i; }
  }
  return BestIdx;
}

int orc_match_project_frame(const orc_frame* cur, const orc_frame* last, const uint8_t* has_mp, const float* Xw,
                            const float* mp_desc, const float* Tcw, const float* K, float th, int check_ori,
                            int32_t* match_cur, const uint8_t* obs_pos) {
  // obs_pos[i] = pMP->Observations() > 0 of last-frame map point i (NULL: all true)
  int nmatches = 0;
  std::vector<int> rotHist[HISTO_LENGTH];
  const float factor = 1.0f / HISTO_LENGTH;
  const float fx = K[0], fy = K[1], cx = K[2], cy = K[3];
  for (int j = 0; j < cur->N; ++j) match_cur[j] = -1;  // CurrentFrame.mvpMapPoints all NULL (Tracking.cc:670)
  for (int i = 0; i < last->N; i++) {
    if (!has_mp[i]) continue;  // pMP && !mvbOutlier[i]
    float x3Dc[3];
    transform(Tcw, Xw + 3 * i, x3Dc);
    const float xc = x3Dc[0], yc = x3Dc[1];
    const float invzc = 1.0 / x3Dc[2];
    if (invzc < 0) continue;
    float u = fx * xc * invzc + cx;
    float v = fy * yc * invzc + cy;
    if (u < cur->mnMinX || u > cur->mnMaxX) continue;
    if (v < cur->mnMinY || v > cur->mnMaxY) continue;
    const int nLastOctave = last->kps[i].octave;
    const float radius = th * cur->mvScaleFactors[nLastOctave];
    const std::vector<size_t> vIndices2 = cur->GetFeaturesInArea(u, v, radius, nLastOctave - 1, nLastOctave + 1);
    if (vIndices2.empty()) continue;
    const float* dMP = mp_desc + (size_t)i * 128;
    float bestDist = 100;
    int bestIdx2 = -1;
    for (size_t k = 0; k < vIndices2.size(); ++k) {
      const size_t i2 = vIndices2[k];
      if (match_cur[i2] >= 0)                                  // if(CurrentFrame.mvpMapPoints[i2])
        if (!obs_pos || obs_pos[match_cur[i2]]) continue;      //   if(...->Observations()>0) continue;  (:1392-1395)
      const float dist = DescriptorDistance(dMP, cur->desc.data() + i2 * 128);
      if (dist < bestDist) { bestDist = dist; bestIdx2 = (int)i2; }
    }
    if (bestDist <= TH_HIGH) {
      match_cur[bestIdx2] = i;
      nmatches++;
      if (check_ori) {
        float rot = last->kps[i].angle - cur->kps[bestIdx2].angle;
        if (rot < 0.0) rot += 360.0f;
        int bin = (int)std::round(rot * factor);
        if (bin == HISTO_LENGTH) bin = 0;
        rotHist[bin].push_back(bestIdx2);
      }
    }
  }
  if (check_ori) {
    int ind1 = -1, ind2 = -1, ind3 = -1;
    ComputeThreeMaxima(rotHist, HISTO_LENGTH, ind1, ind2, ind3);
    for (int i = 0; i < HISTO_LENGTH; i++)
      if (i != ind1 && i != ind2 && i != ind3)
        for (size_t j = 0, jend = rotHist[i].size(); j < jend; j++) {
          match_cur[rotHist[i][j]] = -1;
          nmatches--;
        }
  }
  return nmatches;
}

int orc_match_project_points(const orc_frame* F, int n_mp, const uint8_t* in_view, const float* proj,
                             const int32_t* level, const float* view_cos, const float* desc, const uint8_t* occupied,
                             float th, float nn_ratio, int32_t* match_cur, const uint8_t* obs_pos) {
  int nmatches = 0;
  const bool bFactor = th != 1.0;
  for (int j = 0; j < F->N; ++j) match_cur[j] = -1;
  for (int iMP = 0; iMP < n_mp; iMP++) {
    if (!in_view[iMP]) continue;  // mbTrackInView && !isBad()
    const int nPredictedLevel = level[iMP];
    float r = view_cos[iMP] > 0.998 ? 2.5f : 4.0f;  // RadiusByViewingCos (:126-132)
    if (bFactor) r *= th;
    const std::vector<size_t> vIndices = F->GetFeaturesInArea(proj[2 * iMP], proj[2 * iMP + 1],
                                                              r * F->mvScaleFactors[nPredictedLevel],
                                                              nPredictedLevel - 1, nPredictedLevel);
    if (vIndices.empty()) continue;
    const float* MPdescriptor = desc + (size_t)iMP * 128;
    float bestDist = 256;
    int bestLevel = -1;
    float bestDist2 = 256;
    int bestLevel2 = -1;
    int bestIdx = -1;
    for (size_t k = 0; k < vIndices.size(); ++k) {
      const size_t idx = vIndices[k];
      if (occupied[idx]) continue;                                  // F.mvpMapPoints[idx] with Observations() > 0 on entry
      if (match_cur[idx] >= 0 && (!obs_pos || obs_pos[match_cur[idx]])) continue;  // given one in this call (:86-88)
      const float dist = DescriptorDistance(MPdescriptor, F->desc.data() + idx * 128);
      if (dist < bestDist) {
        bestDist2 = bestDist;
        bestDist = dist;
        bestLevel2 = bestLevel;
        bestLevel = F->kps[idx].octave;
        bestIdx = (int)idx;
      } else if (dist < bestDist2) {
        bestLevel2 = F->kps[idx].octave;
        bestDist2 = dist;
      }
    }
    if (bestDist <= TH_HIGH) {
      if (bestLevel == bestLevel2 && bestDist > nn_ratio * bestDist2) continue;
      match_cur[bestIdx] = iMP;
      nmatches++;
      nmatches++;
    }
  }
  return nmatches;
}

void orc_frustum(const orc_frame* F, int n, const float* Xw, const float* normal, const float* min_dist,
                 const float* max_dist, const float* Tcw, const float* K, float viewingCosLimit, uint8_t* in_view,
                 float* proj, int32_t* level, float* view_cos) {
  const float fx = K[0], fy = K[1], cx = K[2], cy = K[3];
  // mOw = -mRcw.t()*mtcw (Frame.cc:157): general gemm path, double accumulation, one rounding
  float Ow[3];
  for (int i = 0; i < 3; ++i) {
    double s = 0;
    for (int k = 0; k < 3; ++k) s += (double)Tcw[k * 4 + i] * (double)Tcw[k * 4 + 3];
    Ow[i] = (float)(-1.0 * s);
  }
  for (int m = 0; m < n; ++m) {
    in_view[m] = 0; proj[2 * m] = proj[2 * m + 1] = 0; level[m] = 0; view_cos[m] = 0;
    const float* P = Xw + 3 * m;
    float Pc[3];
    transform(Tcw, P, Pc);
    const float PcX = Pc[0], PcY = Pc[1], PcZ = Pc[2];
    if (PcZ < 0.0f) continue;
    const float invz = 1.0f / PcZ;
    const float u = fx * PcX * invz + cx;
    const float v = fy * PcY * invz + cy;
    if (u < F->mnMinX || u > F->mnMaxX) continue;
    if (v < F->mnMinY || v > F->mnMaxY) continue;
    const float maxDistance = 1.2f * max_dist[m];  // GetMaxDistanceInvariance (MapPoint.cc:415-419)
    const float minDistance = 0.8f * min_dist[m];  // GetMinDistanceInvariance
    const float PO[3] = {P[0] - Ow[0], P[1] - Ow[1], P[2] - Ow[2]};
    double nn = 0;
    for (int k = 0; k < 3; ++k) nn += (double)PO[k] * (double)PO[k];
    const float dist = (float)std::sqrt(nn);  // cv::norm
    if (dist < minDistance || dist > maxDistance) continue;
    const float* Pn = normal + 3 * m;
    double dot = 0;
    for (int k = 0; k < 3; ++k) dot += (double)PO[k] * (double)Pn[k];
    const float viewCos = (float)(dot / dist);
    if (viewCos < viewingCosLimit) continue;
    // MapPoint::PredictScale (MapPoint.cc:438-453)
    const float ratio = max_dist[m] / dist;
    int nScale = (int)std::ceil(std::log(ratio) / F->mfLogScaleFactor);
    if (nScale < 0) nScale = 0;
    else if (nScale >= F->mnScaleLevels) nScale = F->mnScaleLevels - 1;
    in_view[m] = 1;
    proj[2 * m] = u; proj[2 * m + 1] = v;
    level[m] = nScale;
    view_cos[m] = viewCos;
  }
}

int orc_match_init(const orc_frame* F1, const orc_frame* F2, float* vbPrevMatched, int windowSize, float nn_ratio,
                   int check_ori, int32_t* vnMatches12) {
  int nmatches = 0;
  for (int i = 0; i < F1->N; ++i) vnMatches12[i] = -1;
  std::vector<int> rotHist[HISTO_LENGTH];
  const float factor = 1.0f / HISTO_LENGTH;
  std::vector<float> vMatchedDistance(F2->N, 100);
  std::vector<int> vnMatches21(F2->N, -1);
  for (int i1 = 0; i1 < F1->N; i1++) {
    const orc_keypoint kp1 = F1->kps[i1];
    const int level1 = kp1.octave;
    if (level1 > 0) continue;
    const std::vector<size_t> vIndices2 =
        F2->GetFeaturesInArea(vbPrevMatched[2 * i1], vbPrevMatched[2 * i1 + 1], (float)windowSize, level1, level1);
    if (vIndices2.empty()) continue;
    const float* d1 = F1->desc.data() + (size_t)i1 * 128;
    float bestDist = 100.0, bestDist2 = 100.0;
    int bestIdx2 = -1;
    for (size_t k = 0; k < vIndices2.size(); ++k) {
      const size_t i2 = vIndices2[k];
      const float dist = DescriptorDistance(d1, F2->desc.data() + i2 * 128);
      if (vMatchedDistance[i2] <= dist) continue;
      if (dist < bestDist) { bestDist2 = bestDist; bestDist = dist; bestIdx2 = (int)i2; }
      else if (dist < bestDist2) bestDist2 = dist;
    }
    if (bestDist <= TH_LOW) {
      if (bestDist < (float)bestDist2 * nn_ratio) {
        if (vnMatches21[bestIdx2] >= 0) { vnMatches12[vnMatches21[bestIdx2]] = -1; nmatches--; }
        vnMatches12[i1] = bestIdx2;
        vnMatches21[bestIdx2] = i1;
        vMatchedDistance[bestIdx2] = bestDist;
        nmatches++;
        if (check_ori) {
          float rot = F1->kps[i1].angle - F2->kps[bestIdx2].angle;
          if (rot < 0.0) rot += 360.0f;
          int bin = (int)std::round(rot * factor);
          if (bin == HISTO_LENGTH) bin = 0;
          rotHist[bin].push_back(i1);
        }
      }
    }
  }
  if (check_ori) {
    int ind1 = -1, ind2 = -1, ind3 = -1;
    ComputeThreeMaxima(rotHist, HISTO_LENGTH, ind1, ind2, ind3);
    for (int i = 0; i < HISTO_LENGTH; i++) {
      if (i == ind1 || i == ind2 || i == ind3) continue;
      for (size_t j = 0, jend = rotHist[i].size(); j < jend; j++) {
        const int idx1 = rotHist[i][j];
        if (vnMatches12[idx1] >= 0) { vnMatches12[idx1] = -1; nmatches--; }
      }
    }
  }
  for (int i1 = 0; i1 < F1->N; i1++)
    if (vnMatches12[i1] >= 0) {
      vbPrevMatched[2 * i1] = F2->kps[vnMatches12[i1]].x;
      vbPrevMatched[2 * i1 + 1] = F2->kps[vnMatches12[i1]].y;
    }
  return nmatches;
}


// ORBmatcher::SearchByBoW(KeyFrame*, Frame&, vector<MapPoint*>&)  (ORBmatcher.cc:156-297)
int orc_match_bow(const orc_frame* KF, const orc_frame* F, int nn_kf, const int32_t* node_kf, const int32_t* start_kf,
                  const int32_t* idx_kf, int nn_f, const int32_t* node_f, const int32_t* start_f, const int32_t* idx_f,
                  const uint8_t* has_mp_kf, float nn_ratio, int check_ori, int32_t* match_f) {
  for (int j = 0; j < F->N; ++j) match_f[j] = -1;
  int nmatches = 0;
  std::vector<int> rotHist[HISTO_LENGTH];
  const float factor = 1.0f / HISTO_LENGTH;
  int a = 0, b = 0;
  while (a < nn_kf && b < nn_f) {
    if (node_kf[a] == node_f[b]) {
      for (int iKF = start_kf[a]; iKF < start_kf[a + 1]; iKF++) {
        const int realIdxKF = idx_kf[iKF];
        if (!has_mp_kf[realIdxKF]) continue;  // !pMP || pMP->isBad()
        const float* dKF = KF->desc.data() + (size_t)realIdxKF * 128;
        float bestDist1 = 256, bestDist2 = 256;
        int bestIdxF = -1;
        for (int iF = start_f[b]; iF < start_f[b + 1]; iF++) {
          const int realIdxF = idx_f[iF];
          if (match_f[realIdxF] >= 0) continue;
          const float dist = DescriptorDistance(dKF, F->desc.data() + (size_t)realIdxF * 128);
          if (dist < bestDist1) { bestDist2 = bestDist1; bestDist1 = dist; bestIdxF = realIdxF; }
          else if (dist < bestDist2) bestDist2 = dist;
        }
        if (bestDist1 <= TH_LOW) {
          if (static_cast<float>(bestDist1) < nn_ratio * static_cast<float>(bestDist2)) {
            match_f[bestIdxF] = realIdxKF;
            if (check_ori) {
              float rot = KF->kps[realIdxKF].angle - F->kps[bestIdxF].angle;
              if (rot < 0.0) rot += 360.0f;
              int bin = (int)std::round(rot * factor);
              if (bin == HISTO_LENGTH) bin = 0;
              rotHist[bin].push_back(bestIdxF);
            }
            nmatches++;
          }
        }
      }
      a++; b++;
    } else if (node_kf[a] < node_f[b]) {
      a = (int)(std::lower_bound(node_kf, node_kf + nn_kf, node_f[b]) - node_kf);
    } else {
      b = (int)(std::lower_bound(node_f, node_f + nn_f, node_kf[a]) - node_f);
    }
  }
  if (check_ori) {
    int ind1 = -1, ind2 = -1, ind3 = -1;
    ComputeThreeMaxima(rotHist, HISTO_LENGTH, ind1, ind2, ind3);
    for (int i = 0; i < HISTO_LENGTH; i++) {
      if (i == ind1 || i == ind2 || i == ind3) continue;
      for (size_t j = 0, jend = rotHist[i].size(); j < jend; j++) { match_f[rotHist[i][j]] = -1; nmatches--; }
    }
  }
  return nmatches;
}

// ORBmatcher::SearchForTriangulation (ORBmatcher.cc:669-822) + CheckDistEpipolarLine (:136-153)
int orc_match_triangulate(const orc_frame* KF1, const orc_frame* KF2, int nn1, const int32_t* node1, const int32_t* start1,
                          const int32_t* idx1v, int nn2, const int32_t* node2, const int32_t* start2, const int32_t* idx2v,
                          const uint8_t* has_mp1, const uint8_t* has_mp2, const float* F12, float ex, float ey,
                          int check_ori, int32_t* vMatches12) {
  int nmatches = 0;
  std::vector<bool> vbMatched2(KF2->N, false);  // never set by the reference either
  for (int i = 0; i < KF1->N; ++i) vMatches12[i] = -1;
  std::vector<int> rotHist[HISTO_LENGTH];
  const float factor = 1.0f / HISTO_LENGTH;
  std::vector<float> sigma2(KF2->mnScaleLevels);
  for (int l = 0; l < KF2->mnScaleLevels; ++l) sigma2[l] = KF2->mvScaleFactors[l] * KF2->mvScaleFactors[l];
  int a = 0, b = 0;
  while (a < nn1 && b < nn2) {
    if (node1[a] == node2[b]) {
      for (int i1 = start1[a]; i1 < start1[a + 1]; i1++) {
        const int idx1 = idx1v[i1];
        if (has_mp1[idx1]) continue;
        const orc_keypoint& kp1 = KF1->kps[idx1];
        const float* d1 = KF1->desc.data() + (size_t)idx1 * 128;
        float bestDist = TH_LOW;
        int bestIdx2 = -1;
        for (int i2 = start2[b]; i2 < start2[b + 1]; i2++) {
          const int idx2 = idx2v[i2];
          if (vbMatched2[idx2] || has_mp2[idx2]) continue;
          const float dist = DescriptorDistance(d1, KF2->desc.data() + (size_t)idx2 * 128);
          if (dist > TH_LOW || dist > bestDist) continue;
          const orc_keypoint& kp2 = KF2->kps[idx2];
          {
            const float distex = ex - kp2.x;
            const float distey = ey - kp2.y;
            if (distex * distex + distey * distey < 100 * KF2->mvScaleFactors[kp2.octave]) continue;
          }
          // CheckDistEpipolarLine
          const float la = kp1.x * F12[0] + kp1.y * F12[3] + F12[6];
          const float lb = kp1.x * F12[1] + kp1.y * F12[4] + F12[7];
          const float lc = kp1.x * F12[2] + kp1.y * F12[5] + F12[8];
          const float num = la * kp2.x + lb * kp2.y + lc;
          const float den = la * la + lb * lb;
          if (den == 0) continue;
          const float dsqr = num * num / den;
          if (dsqr < 3.84 * sigma2[kp2.octave]) { bestIdx2 = idx2; bestDist = dist; }
        }
        if (bestIdx2 >= 0) {
          const orc_keypoint& kp2 = KF2->kps[bestIdx2];
          vMatches12[idx1] = bestIdx2;
          nmatches++;
          if (check_ori) {
            float rot = kp1.angle - kp2.angle;
            if (rot < 0.0) rot += 360.0f;
            int bin = (int)std::round(rot * factor);
            if (bin == HISTO_LENGTH) bin = 0;
            rotHist[bin].push_back(idx1);
          }
        }
      }
      a++; b++;
    } else if (node1[a] < node2[b]) {
      a = (int)(std::lower_bound(node1, node1 + nn1, node2[b]) - node1);
    } else {
      b = (int)(std::lower_bound(node2, node2 + nn2, node1[a]) - node2);
    }
  }
  if (check_ori) {
    int ind1 = -1, ind2 = -1, ind3 = -1;
    ComputeThreeMaxima(rotHist, HISTO_LENGTH, ind1, ind2, ind3);
    for (int i = 0; i < HISTO_LENGTH; i++) {
      if (i == ind1 || i == ind2 || i == ind3) continue;
      for (size_t j = 0, jend = rotHist[i].size(); j < jend; j++) { vMatches12[rotHist[i][j]] = -1; nmatches--; }
    }
  }
  return nmatches;
}


// ORBmatcher::Fuse(KeyFrame*, const vector<MapPoint*>&, float th), search half (ORBmatcher.cc:825-936)
void orc_fuse_search(const orc_frame* KF, int n_mp, const uint8_t* valid, const float* Xw, const float* normal,
                     const float* min_dist, const float* max_dist, const float* desc, const float* Tcw, const float* K,
                     float th, int32_t* best_idx, float* best_dist) {
  const float fx = K[0], fy = K[1], cx = K[2], cy = K[3];
  // KeyFrame::SetPose (KeyFrame.cc): Rwc = Rcw.t() materialised, Ow = -Rwc*tcw through gemm's small-matrix path
  float Ow[3];
  for (int i = 0; i < 3; ++i) {
    const float t0 = Tcw[0 * 4 + i] * Tcw[3] + Tcw[1 * 4 + i] * Tcw[7] + Tcw[2 * 4 + i] * Tcw[11];
    Ow[i] = (float)((double)t0 * -1.0);
  }
  std::vector<float> invSigma2(KF->mnScaleLevels);
  for (int l = 0; l < KF->mnScaleLevels; ++l) invSigma2[l] = 1.0f / (KF->mvScaleFactors[l] * KF->mvScaleFactors[l]);
  for (int i = 0; i < n_mp; i++) {
    best_idx[i] = -1;
    best_dist[i] = 256;
    if (!valid[i]) continue;
    const float* p3Dw = Xw + 3 * i;
    float p3Dc[3];
    transform(Tcw, p3Dw, p3Dc);
    if (p3Dc[2] < 0.0f) continue;
    const float invz = 1 / p3Dc[2];
    const float x = p3Dc[0] * invz;
    const float y = p3Dc[1] * invz;
    const float u = fx * x + cx;
    const float v = fy * y + cy;
    if (!(u >= KF->mnMinX && u < KF->mnMaxX && v >= KF->mnMinY && v < KF->mnMaxY)) continue;  // IsInImage
    const float maxDistance = 1.2f * max_dist[i];
    const float minDistance = 0.8f * min_dist[i];
    const float PO[3] = {p3Dw[0] - Ow[0], p3Dw[1] - Ow[1], p3Dw[2] - Ow[2]};
    double nn = 0;
    for (int k = 0; k < 3; ++k) nn += (double)PO[k] * (double)PO[k];
    const float dist3D = (float)std::sqrt(nn);
    if (dist3D < minDistance || dist3D > maxDistance) continue;
    const float* Pn = normal + 3 * i;
    double dot = 0;
    for (int k = 0; k < 3; ++k) dot += (double)PO[k] * (double)Pn[k];
    if (dot < 0.5 * dist3D) continue;
    const float ratio = max_dist[i] / dist3D;  // PredictScale(dist3D, pKF)
    int nPredictedLevel = (int)std::ceil(std::log(ratio) / KF->mfLogScaleFactor);
    if (nPredictedLevel < 0) nPredictedLevel = 0;
    else if (nPredictedLevel >= KF->mnScaleLevels) nPredictedLevel = KF->mnScaleLevels - 1;
    const float radius = th * KF->mvScaleFactors[nPredictedLevel];
    const std::vector<size_t> vIndices = KF->GetFeaturesInArea(u, v, radius, -1, -1);  // KeyFrame.cc:839-878: no level filter
    if (vIndices.empty()) continue;
    const float* dMP = desc + (size_t)i * 128;
    float bestDist = 256;
    int bestIdx = -1;
    for (size_t k = 0; k < vIndices.size(); ++k) {
      const size_t idx = vIndices[k];
      const orc_keypoint& kp = KF->kps[idx];
      const int kpLevel = kp.octave;
      if (kpLevel < nPredictedLevel - 1 || kpLevel > nPredictedLevel) continue;
      {
        const float ex = u - kp.x;
        const float ey = v - kp.y;
        const float e2 = ex * ex + ey * ey;
        if (e2 * invSigma2[kpLevel] > 5.99) continue;
      }
      const float dist = DescriptorDistance(dMP, KF->desc.data() + idx * 128);
      if (dist < bestDist) { bestDist = dist; bestIdx = (int)idx; }
    }
    if (bestDist <= TH_LOW) { best_idx[i] = bestIdx; best_dist[i] = bestDist; }
  }
}


// ---- relocalisation / loop-closing variants (row M4 of SURVEY 8(a)) ---------------------------------------------

namespace {
// -Rcw.t()*tcw with the transpose flag on the gemm: general path, double accumulation (Frame.cc:157 idiom)
void centre_gemm_t(const float* R /*[3][3]*/, const float* t, float* Ow) {
  for (int i = 0; i < 3; ++i) {
    double s = 0;
    for (int k = 0; k < 3; ++k) s += (double)R[k * 3 + i] * (double)t[k];
    Ow[i] = (float)(-1.0 * s);
  }
}
// Scw -> Rcw = sRcw / scw, tcw = Scw.col(3) / scw (ORBmatcher.cc:310-313): Mat / scalar = convertTo with float scale
void decompose_sim3(const float* Scw, float* Rcw, float* tcw) {
  double d = 0;
  for (int k = 0; k < 3; ++k) d += (double)Scw[k] * (double)Scw[k];
  const float scw = (float)std::sqrt(d);
  const float inv = (float)(1.0 / (double)scw);
  for (int r = 0; r < 3; ++r) {
    for (int c = 0; c < 3; ++c) Rcw[r * 3 + c] = Scw[r * 4 + c] * inv + 0.0f;
    tcw[r] = Scw[r * 4 + 3] * inv + 0.0f;
  }
}
inline void rot_add(const float* R, const float* t, const float* X, float* out) {  // R*X + t, gemm small path + MatExpr add
  for (int r = 0; r < 3; ++r) {
    const float t0 = R[r * 3 + 0] * X[0] + R[r * 3 + 1] * X[1] + R[r * 3 + 2] * X[2];
    out[r] = (float)((double)t0 + (double)t[r]);
  }
}
inline int predict_scale(float maxd_raw, float dist, float logsf, int nlevels) {  // MapPoint::PredictScale
  const float ratio = maxd_raw / dist;
  int s = (int)std::ceil(std::log(ratio) / logsf);
  if (s < 0) s = 0;
  else if (s >= nlevels) s = nlevels - 1;
  return s;
}
inline float norm3f(const float* p) {
  double nn = 0;
  for (int k = 0; k < 3; ++k) nn += (double)p[k] * (double)p[k];
  return (float)std::sqrt(nn);
}
}  // namespace

// ORBmatcher::SearchByProjection(Frame& CurrentFrame, KeyFrame* pKF, const set<MapPoint*>& sAlreadyFound, float th,
// float ORBdist) (ORBmatcher.cc:1455-1582).  valid[i] = pMP && !isBad() && !sAlreadyFound.count(pMP);
// occupied_in[j] = CurrentFrame.mvpMapPoints[j] != NULL on entry.  match_cur[j] = index i into pKF's map points.
int orc_match_project_keyframe(const orc_frame* cur, int n_kf, const uint8_t* valid, const float* Xw, const float* min_dist,
                               const float* max_dist, const float* desc, const float* kf_angle, const uint8_t* occupied_in,
                               const float* Tcw, const float* K, float th, float ORBdist, int check_ori, int32_t* match_cur) {
  int nmatches = 0;
  float Rcw[9], tcw[3], Ow[3];
  for (int r = 0; r < 3; ++r) { for (int c = 0; c < 3; ++c) Rcw[r * 3 + c] = Tcw[r * 4 + c]; tcw[r] = Tcw[r * 4 + 3]; }
  centre_gemm_t(Rcw, tcw, Ow);
  std::vector<int> rotHist[HISTO_LENGTH];
  const float factor = 1.0f / HISTO_LENGTH;
  const float fx = K[0], fy = K[1], cx = K[2], cy = K[3];
  std::vector<uint8_t> occ(occupied_in, occupied_in + cur->N);
  for (int j = 0; j < cur->N; ++j) match_cur[j] = -1;
  for (int i = 0; i < n_kf; i++) {
    if (!valid[i]) continue;
    const float* x3Dw = Xw + 3 * i;
    float x3Dc[3];
    rot_add(Rcw, tcw, x3Dw, x3Dc);
    const float xc = x3Dc[0], yc = x3Dc[1];
    const float invzc = 1.0 / x3Dc[2];
    const float u = fx * xc * invzc + cx;
    const float v = fy * yc * invzc + cy;
    if (u < cur->mnMinX || u > cur->mnMaxX) continue;
    if (v < cur->mnMinY || v > cur->mnMaxY) continue;
    const float PO[3] = {x3Dw[0] - Ow[0], x3Dw[1] - Ow[1], x3Dw[2] - Ow[2]};
    const float dist3D = norm3f(PO);
    const float maxDistance = 1.2f * max_dist[i], minDistance = 0.8f * min_dist[i];
    if (dist3D < minDistance || dist3D > maxDistance) continue;
    const int nPredictedLevel = predict_scale(max_dist[i], dist3D, cur->mfLogScaleFactor, cur->mnScaleLevels);
    const float radius = th * cur->mvScaleFactors[nPredictedLevel];
    const std::vector<size_t> vIndices2 = cur->GetFeaturesInArea(u, v, radius, nPredictedLevel - 1, nPredictedLevel + 1);
    if (vIndices2.empty()) continue;
    const float* dMP = desc + (size_t)i * 128;
    float bestDist = 256;
    int bestIdx2 = -1;
    for (size_t k = 0; k < vIndices2.size(); ++k) {
      const size_t i2 = vIndices2[k];
      if (occ[i2]) continue;
      const float dist = DescriptorDistance(dMP, cur->desc.data() + i2 * 128);
      if (dist < bestDist) { bestDist = dist; bestIdx2 = (int)i2; }
    }
    if (bestDist <= ORBdist) {
      occ[bestIdx2] = 1;
      match_cur[bestIdx2] = i;
      nmatches++;
      if (check_ori) {
        float rot = kf_angle[i] - cur->kps[bestIdx2].angle;
        if (rot < 0.0) rot += 360.0f;
        int bin = (int)std::round(rot * factor);
        if (bin == HISTO_LENGTH) bin = 0;
        rotHist[bin].push_back(bestIdx2);
      }
    }
  }
  if (check_ori) {
    int ind1 = -1, ind2 = -1, ind3 = -1;
    ComputeThreeMaxima(rotHist, HISTO_LENGTH, ind1, ind2, ind3);
    for (int i = 0; i < HISTO_LENGTH; i++)
      if (i != ind1 && i != ind2 && i != ind3)
        for (size_t j = 0, jend = rotHist[i].size(); j < jend; j++) { match_cur[rotHist[i][j]] = -1; nmatches--; }
  }
  return nmatches;
}

// shared front part of the two Scw searches (ORBmatcher.cc:300-366 and :963-1010): returns false if the point is gated out
static bool sim3_project(const orc_frame* KF, const float* Rcw, const float* tcw, const float* Ow, const float* K, const float* p3Dw,
                         const float* Pn, float mind_raw, float maxd_raw, bool double_invz, float* u, float* v, int* level) {
  float p3Dc[3];
  rot_add(Rcw, tcw, p3Dw, p3Dc);
  if (p3Dc[2] < 0.0f) return false;
  const float invz = double_invz ? (float)(1.0 / p3Dc[2]) : 1 / p3Dc[2];  // :993 `1.0/z`, :331 `1/z`
  const float x = p3Dc[0] * invz, y = p3Dc[1] * invz;
  *u = K[0] * x + K[2];
  *v = K[1] * y + K[3];
  if (!(*u >= KF->mnMinX && *u < KF->mnMaxX && *v >= KF->mnMinY && *v < KF->mnMaxY)) return false;  // IsInImage
  const float maxDistance = 1.2f * maxd_raw, minDistance = 0.8f * mind_raw;
  const float PO[3] = {p3Dw[0] - Ow[0], p3Dw[1] - Ow[1], p3Dw[2] - Ow[2]};
  const float dist = norm3f(PO);
  if (dist < minDistance || dist > maxDistance) return false;
  double dot = 0;
  for (int k = 0; k < 3; ++k) dot += (double)PO[k] * (double)Pn[k];
  if (dot < 0.5 * dist) return false;
  *level = predict_scale(maxd_raw, dist, KF->mfLogScaleFactor, KF->mnScaleLevels);
  return true;
}

// ORBmatcher::SearchByProjection(KeyFrame* pKF, cv::Mat Scw, const vector<MapPoint*>& vpPoints,
// vector<MapPoint*>& vpMatched, int th) (:300-413).  valid[i] = !isBad() && !spAlreadyFound.count(pMP);
// matched_kp[j] in/out: -1 = vpMatched[j] NULL, >= 0 = index into vpPoints written by this call, -2 = occupied on entry.
int orc_match_project_sim3(const orc_frame* KF, const float* Scw, int n_mp, const uint8_t* valid, const float* Xw, const float* normal,
                           const float* min_dist, const float* max_dist, const float* desc, const float* K, int th,
                           int32_t* matched_kp) {
  float Rcw[9], tcw[3], Ow[3];
  decompose_sim3(Scw, Rcw, tcw);
  centre_gemm_t(Rcw, tcw, Ow);
  int nmatches = 0;
  for (int iMP = 0; iMP < n_mp; iMP++) {
    if (!valid[iMP]) continue;
    float u, v;
    int nPredictedLevel;
    if (!sim3_project(KF, Rcw, tcw, Ow, K, Xw + 3 * iMP, normal + 3 * iMP, min_dist[iMP], max_dist[iMP], false, &u, &v, &nPredictedLevel))
      continue;
    const float radius = th * KF->mvScaleFactors[nPredictedLevel];
    const std::vector<size_t> vIndices = KF->GetFeaturesInArea(u, v, radius, -1, -1);
    if (vIndices.empty()) continue;
    const float* dMP = desc + (size_t)iMP * 128;
    float bestDist = 256;
    int bestIdx = -1;
    for (size_t k = 0; k < vIndices.size(); ++k) {
      const size_t idx = vIndices[k];
      if (matched_kp[idx] != -1) continue;
      const int kpLevel = KF->kps[idx].octave;
      if (kpLevel < nPredictedLevel - 1 || kpLevel > nPredictedLevel) continue;
      const float dist = DescriptorDistance(dMP, KF->desc.data() + idx * 128);
      if (dist < bestDist) { bestDist = dist; bestIdx = (int)idx; }
    }
    if (bestDist <= TH_LOW) { matched_kp[bestIdx] = iMP; nmatches++; }
  }
  return nmatches;
}

// ORBmatcher::Fuse(KeyFrame* pKF, cv::Mat Scw, const vector<MapPoint*>& vpPoints, float th, vector<MapPoint*>& vpReplacePoint)
// (:963-1086), search half: best_idx[i] = keypoint of pKF to fuse point i into, -1 = none
void orc_fuse_search_sim3(const orc_frame* KF, const float* Scw, int n_mp, const uint8_t* valid, const float* Xw, const float* normal,
                          const float* min_dist, const float* max_dist, const float* desc, const float* K, float th, int32_t* best_idx,
                          float* best_dist) {
  float Rcw[9], tcw[3], Ow[3];
  decompose_sim3(Scw, Rcw, tcw);
  centre_gemm_t(Rcw, tcw, Ow);
  for (int iMP = 0; iMP < n_mp; iMP++) {
    best_idx[iMP] = -1;
    best_dist[iMP] = 100;
    if (!valid[iMP]) continue;
    float u, v;
    int nPredictedLevel;
    if (!sim3_project(KF, Rcw, tcw, Ow, K, Xw + 3 * iMP, normal + 3 * iMP, min_dist[iMP], max_dist[iMP], true, &u, &v, &nPredictedLevel))
      continue;
    const float radius = th * KF->mvScaleFactors[nPredictedLevel];
    const std::vector<size_t> vIndices = KF->GetFeaturesInArea(u, v, radius, -1, -1);
    if (vIndices.empty()) continue;
    const float* dMP = desc + (size_t)iMP * 128;
    float bestDist = 100;
    int bestIdx = -1;
    for (size_t k = 0; k < vIndices.size(); ++k) {
      const size_t idx = vIndices[k];
      const int kpLevel = KF->kps[idx].octave;
      if (kpLevel < nPredictedLevel - 1 || kpLevel > nPredictedLevel) continue;
      const float dist = DescriptorDistance(dMP, KF->desc.data() + idx * 128);
      if (dist < bestDist) { bestDist = dist; bestIdx = (int)idx; }
    }
    if (bestDist <= TH_LOW) { best_idx[iMP] = bestIdx; best_dist[iMP] = bestDist; }
  }
}

// ORBmatcher::SearchBySim3 (:1090-1314).  has1[i] / has2[i]: map point present, not bad, not already matched
// (vbAlreadyMatched); Xw1 / Xw2 world positions; match12[i1] = i2 of a mutually consistent pair, else -1.
int orc_match_sim3(const orc_frame* KF1, const orc_frame* KF2, const uint8_t* has1, const uint8_t* has2, const float* Xw1,
                   const float* Xw2, const float* mind1, const float* maxd1, const float* mind2, const float* maxd2, const float* desc1,
                   const float* desc2, const float* T1w, const float* T2w, float s12, const float* R12, const float* t12, const float* K,
                   float th, int32_t* match12) {
  float R1w[9], t1w[3], R2w[9], t2w[3];
  for (int r = 0; r < 3; ++r) {
    for (int c = 0; c < 3; ++c) { R1w[r * 3 + c] = T1w[r * 4 + c]; R2w[r * 3 + c] = T2w[r * 4 + c]; }
    t1w[r] = T1w[r * 4 + 3]; t2w[r] = T2w[r * 4 + 3];
  }
  // sR12 = s12*R12 ; sR21 = (1.0/s12)*R12.t() ; t21 = -sR21*t12    (:1105-1107): scaled Mat = convertTo with float scale
  float sR12[9], sR21[9], t21[3];
  const float a12 = (float)(double)s12, a21 = (float)(1.0 / (double)s12);
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 3; ++c) { sR12[r * 3 + c] = R12[r * 3 + c] * a12 + 0.0f; sR21[r * 3 + c] = R12[c * 3 + r] * a21 + 0.0f; }
  for (int r = 0; r < 3; ++r) {
    const float t0 = sR21[r * 3 + 0] * t12[0] + sR21[r * 3 + 1] * t12[1] + sR21[r * 3 + 2] * t12[2];
    t21[r] = (float)((double)t0 * -1.0);
  }
  const int N1 = KF1->N, N2 = KF2->N;
  std::vector<int> vnMatch1(N1, -1), vnMatch2(N2, -1);
  auto one_way = [&](const orc_frame* A, const orc_frame* B, const uint8_t* has, const float* Xw, const float* mind, const float* maxd,
                     const float* desc, const float* Raw, const float* taw, const float* sRba, const float* tba, std::vector<int>& out) {
    for (int i = 0; i < A->N; i++) {
      if (!has[i]) continue;
      float pA[3], pB[3];
      rot_add(Raw, taw, Xw + 3 * i, pA);
      rot_add(sRba, tba, pA, pB);
      if (pB[2] < 0.0) continue;
      const float invz = 1.0 / pB[2];
      const float x = pB[0] * invz, y = pB[1] * invz;
      const float u = K[0] * x + K[2], v = K[1] * y + K[3];
      if (!(u >= B->mnMinX && u < B->mnMaxX && v >= B->mnMinY && v < B->mnMaxY)) continue;
      const float maxDistance = 1.2f * maxd[i], minDistance = 0.8f * mind[i];
      const float dist3D = norm3f(pB);
      if (dist3D < minDistance || dist3D > maxDistance) continue;
      const int nPredictedLevel = predict_scale(maxd[i], dist3D, B->mfLogScaleFactor, B->mnScaleLevels);
      const float radius = th * B->mvScaleFactors[nPredictedLevel];
      const std::vector<size_t> vIndices = B->GetFeaturesInArea(u, v, radius, -1, -1);
      if (vIndices.empty()) continue;
      const float* dMP = desc + (size_t)i * 128;
      float bestDist = 100;
      int bestIdx = -1;
      for (size_t k = 0; k < vIndices.size(); ++k) {
        const size_t idx = vIndices[k];
        const orc_keypoint& kp = B->kps[idx];
        if (kp.octave < nPredictedLevel - 1 || kp.octave > nPredictedLevel) continue;
        const float dist = DescriptorDistance(dMP, B->desc.data() + idx * 128);
        if (dist < bestDist) { bestDist = dist; bestIdx = (int)idx; }
      }
      if (bestDist <= TH_HIGH) out[i] = bestIdx;
    }
  };
  one_way(KF1, KF2, has1, Xw1, mind1, maxd1, desc1, R1w, t1w, sR21, t21, vnMatch1);
  one_way(KF2, KF1, has2, Xw2, mind2, maxd2, desc2, R2w, t2w, sR12, t12, vnMatch2);
  int nFound = 0;
  for (int i1 = 0; i1 < N1; i1++) {
    match12[i1] = -1;
    const int idx2 = vnMatch1[i1];
    if (idx2 >= 0 && vnMatch2[idx2] == i1) { match12[i1] = idx2; nFound++; }
  }
  return nFound;
}

}  // extern "C"
