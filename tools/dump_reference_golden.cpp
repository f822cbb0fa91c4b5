// dump_reference_golden.cpp -- run ON THE REFERENCE SIDE (a catkin workspace of mataiyuan/ASD-SLAM with OpenCV 3.2 and
// libtorch, which this repository's build container does not have) to turn "parity unpinned" for the image front-end
// into a one-command check.  For one grey image it writes, in the ASDG1 container of tests/golden/asdg.py, everything
// the front-end parity tests compare: the pyramid levels, the per-cell FAST corners before the quadtree, the blurred
// levels, ORBextractor's final keypoints and a table of cv::fastAtan2 values.
//
//   build (inside the reference's vslam package, e.g. as an extra executable in its CMakeLists.txt):
//     add_executable(dump_reference_golden <path>/dump_reference_golden.cpp)
//     target_link_libraries(dump_reference_golden ${PROJECT_NAME} ${OpenCV_LIBS} ${TORCH_LIBRARIES})
//   run:    dump_reference_golden frame.png 2000 out.asdg        (nfeatures; 1.2 / 8 / 20 / 7 as in the reference's configs)
//   check:  copy out.asdg to this repository's tests/golden/reference/ and run
//             python -m pytest tests/test_reference_golden.py            (oracle vs the dump, CPU)
//             python -m pytest tests/test_reference_golden.py -m gpu     (HIP vs the dump, on an MI355X)
//
// It only calls the reference's own code and OpenCV: ORBextractor::ComputePyramid / ComputeKeyPointsOctTree through a
// subclass (they are protected, ORBextractor.h:92-95), cv::FAST per cell exactly as ComputeKeyPointsOctTree does
// (ORBextractor.cc:813-876), cv::GaussianBlur as ExtractDesc does (:1226-1227) and cv::fastAtan2 (IC_Angle, :80-107).
// The ORBextractor constructor loads the TorchScript descriptor model the reference is configured with; descriptors are
// not dumped (tests/golden/make_asdnet_golden.py pins the network separately).
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include <opencv2/opencv.hpp>

#include "ORBextractor.h"

namespace {

const int EDGE_THRESHOLD = 19, PATCH_SIZE = 31;  // ORBextractor.cc:75-77

struct Writer {
  FILE* f;
  uint32_t n = 0;
  explicit Writer(const char* path) : f(fopen(path, "wb")) {
    if (!f) { perror(path); exit(2); }
    fwrite("ASDG", 1, 4, f);
    const uint32_t hdr[2] = {1, 0};
    fwrite(hdr, 4, 2, f);
  }
  void array(const std::string& name, uint8_t dtype, const std::vector<uint32_t>& dims, const void* data, size_t bytes) {
    const uint16_t ln = (uint16_t)name.size();
    fwrite(&ln, 2, 1, f);
    fwrite(name.data(), 1, ln, f);
    const uint8_t hd[2] = {dtype, (uint8_t)dims.size()};
    fwrite(hd, 1, 2, f);
    fwrite(dims.data(), 4, dims.size(), f);
    fwrite(data, 1, bytes, f);
    ++n;
  }
  void image(const std::string& name, const cv::Mat& m) {  // u8, possibly a ROI: row by row
    std::vector<uint8_t> buf((size_t)m.rows * m.cols);
    for (int r = 0; r < m.rows; ++r) memcpy(&buf[(size_t)r * m.cols], m.ptr<uint8_t>(r), m.cols);
    array(name, 0, {(uint32_t)m.rows, (uint32_t)m.cols}, buf.data(), buf.size());
  }
  void f32(const std::string& name, const std::vector<float>& v, uint32_t cols) {
    array(name, 2, {(uint32_t)(v.size() / cols), cols}, v.data(), v.size() * 4);
  }
  ~Writer() {
    fseek(f, 8, SEEK_SET);
    fwrite(&n, 4, 1, f);
    fclose(f);
  }
};

struct Dumper : ORB_SLAM2::ORBextractor {
  using ORB_SLAM2::ORBextractor::ORBextractor;
  void pyramid(const cv::Mat& im) { ComputePyramid(im); }
  void keypoints(std::vector<std::vector<cv::KeyPoint>>& all) { ComputeKeyPointsOctTree(all); }
  std::vector<float> scales() { return mvScaleFactor; }
};

}  // namespace

int main(int argc, char** argv) {
  if (argc < 4) { fprintf(stderr, "usage: %s image nfeatures out.asdg [iniThFAST=20 minThFAST=7]\n", argv[0]); return 1; }
  const int nfeatures = atoi(argv[2]), nlevels = 8, iniTh = argc > 4 ? atoi(argv[4]) : 20, minTh = argc > 5 ? atoi(argv[5]) : 7;
  const float scaleFactor = 1.2f;
  cv::Mat im = cv::imread(argv[1], cv::IMREAD_GRAYSCALE);
  if (im.empty()) { fprintf(stderr, "cannot read %s\n", argv[1]); return 1; }
  Dumper ex(nfeatures, scaleFactor, nlevels, iniTh, minTh);
  Writer w(argv[3]);
  const int32_t params[5] = {nfeatures, nlevels, iniTh, minTh, (int32_t)std::lround(scaleFactor * 1000)};
  w.array("params", 1, {5}, params, sizeof params);
  w.image("image", im);

  ex.pyramid(im);
  for (int L = 0; L < nlevels; ++L) {
    const cv::Mat& lv = ex.mvImagePyramid[L];  // the level WITHOUT border: a view into the padded buffer
    w.image("pyr_" + std::to_string(L), lv);
    cv::Mat blur = lv.clone();
    cv::GaussianBlur(blur, blur, cv::Size(7, 7), 2, 2, cv::BORDER_REFLECT_101);
    w.image("blur_" + std::to_string(L), blur);
    // the cell loop of ComputeKeyPointsOctTree, verbatim arithmetic
    const int minBorderX = EDGE_THRESHOLD - 3, minBorderY = minBorderX;
    const int maxBorderX = lv.cols - EDGE_THRESHOLD + 3, maxBorderY = lv.rows - EDGE_THRESHOLD + 3;
    const float width = (maxBorderX - minBorderX), height = (maxBorderY - minBorderY), W = 30;
    const int nCols = width / W, nRows = height / W;
    const int wCell = ceil(width / nCols), hCell = ceil(height / nRows);
    std::vector<float> raw;
    for (int i = 0; i < nRows; i++) {
      const float iniY = minBorderY + i * hCell;
      float maxY = iniY + hCell + 6;
      if (iniY >= maxBorderY - 3) continue;
      if (maxY > maxBorderY) maxY = maxBorderY;
      for (int j = 0; j < nCols; j++) {
        const float iniX = minBorderX + j * wCell;
        float maxX = iniX + wCell + 6;
        if (iniX >= maxBorderX - 6) continue;
        if (maxX > maxBorderX) maxX = maxBorderX;
        std::vector<cv::KeyPoint> cell;
        cv::FAST(lv.rowRange(iniY, maxY).colRange(iniX, maxX), cell, iniTh, true);
        if (cell.empty()) cv::FAST(lv.rowRange(iniY, maxY).colRange(iniX, maxX), cell, minTh, true);
        for (const cv::KeyPoint& k : cell) { raw.push_back(k.pt.x + j * wCell); raw.push_back(k.pt.y + i * hCell); raw.push_back(k.response); }
      }
    }
    w.f32("raw_" + std::to_string(L), raw, 3);
  }

  // final keypoints: ComputeKeyPointsOctTree + the scale-back of ExtractDesc (:1234-1245)
  std::vector<std::vector<cv::KeyPoint>> all;
  ex.keypoints(all);
  const std::vector<float> sc = ex.scales();
  std::vector<float> kps;
  for (int L = 0; L < nlevels; ++L)
    for (cv::KeyPoint k : all[L]) {
      if (L != 0) k.pt *= sc[L];
      kps.insert(kps.end(), {k.pt.x, k.pt.y, k.size, k.angle, k.response, (float)k.octave});
    }
  w.f32("keypoints", kps, 6);

  // cv::fastAtan2 on the moments IC_Angle can produce (integers up to 15 * 255 * ~700 px) plus a fine sweep
  std::vector<float> ain, aout;
  uint32_t s = 12345;
  for (int i = 0; i < 20000; ++i) {
    s = s * 1664525u + 1013904223u; const float y = (float)((int)(s >> 8) % 2000001 - 1000000);
    s = s * 1664525u + 1013904223u; const float x = (float)((int)(s >> 8) % 2000001 - 1000000);
    ain.push_back(y); ain.push_back(x); aout.push_back(cv::fastAtan2(y, x));
  }
  for (int i = 0; i < 3600; ++i) {
    const float a = (float)(i * 0.1 * CV_PI / 180.0), y = 1000.f * sinf(a), x = 1000.f * cosf(a);
    ain.push_back(y); ain.push_back(x); aout.push_back(cv::fastAtan2(y, x));
  }
  w.f32("atan_in", ain, 2);
  w.f32("atan_out", aout, 1);
  fprintf(stderr, "wrote %s: %zu keypoints\n", argv[3], kps.size() / 6);
  return 0;
}
