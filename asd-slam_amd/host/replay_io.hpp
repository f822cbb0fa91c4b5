// replay_io.hpp -- file formats of the reference's headless replay (SURVEY 8(f) rank 3, I/O half).
//
// What Examples/Monocular/kitti.cc needs around System::TrackMonocular, without ROS / OpenCV / glog:
//   ReadCamInfo     camera-config reader   (src/read_write_data_lib/src/read_write.cpp:27-60, files cameraconfig/KITTI/*.txt)
//   ReadImageInfo   image / descriptor config (read_write.cpp:62-89)
//   LoadImages      times.txt + image_0/%06d.png naming (src/vslam/Examples/Monocular/kitti.cc:56-84)
//   TumLine / TumKeyFrameLine   one line of System::SaveTrajectoryTUM / SaveKeyFrameTrajectoryTUM (System.cc:446-535)
//   RotToQuaternion Converter::toQuaternion (Converter.cc:147-159: Eigen::Quaterniond(R), cast to float, order x y z w)
//   ReadPGM         8-bit binary PGM (P5) frames for replays without a PNG decoder
// The Tracking / LocalMapping state machine that would sit between these and the C ABI is the reference's control
// plane and stays the reference's own code (INTEGRATION.md 4b shows the replay loop with read-ahead extraction).
#pragma once
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <fstream>
#include <iomanip>
#include <sstream>
#include <string>
#include <vector>

namespace asd {

inline std::vector<std::string> SplitNonEmpty(const std::string& str, const std::string& delim) {  // read_write.cpp:11-25
  std::vector<std::string> tokens;
  size_t prev = 0, pos = 0;
  do {
    pos = str.find(delim, prev);
    if (pos == std::string::npos) pos = str.length();
    std::string token = str.substr(prev, pos - prev);
    if (!token.empty()) tokens.push_back(token);
    prev = pos + delim.length();
  } while (pos < str.length() && prev < str.length());
  return tokens;
}

struct CamInfo {
  double fx = 0, fy = 0, cx = 0, cy = 0;
  double distort[4] = {0, 0, 0, 0};  // k1 k2 p1 p2
  bool has_Tbc = false;
  double Tbc[12] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0};  // rows 0..2 of the camera-to-body transform
};

// line 1: fx, fy, cx, cy, k1, k2, p1, p2 ; optional line 2: the 12 entries of Tbc (read_write.cpp:27-60)
inline bool ReadCamInfo(const std::string& path, CamInfo& out) {
  std::ifstream f(path.c_str());
  if (!f.is_open()) return false;
  std::string line;
  std::getline(f, line);
  std::vector<std::string> s = SplitNonEmpty(line, ",");
  if (s.size() < 8) return false;
  out.fx = atof(s[0].c_str()); out.fy = atof(s[1].c_str()); out.cx = atof(s[2].c_str()); out.cy = atof(s[3].c_str());
  for (int k = 0; k < 4; ++k) out.distort[k] = atof(s[4 + k].c_str());
  out.has_Tbc = false;
  if (!std::getline(f, line) || line.empty()) return true;  // "Not use camera to imu transformation!"
  s = SplitNonEmpty(line, ",");
  if (s.size() < 12) return false;
  for (int k = 0; k < 12; ++k) out.Tbc[k] = atof(s[k].c_str());
  out.has_Tbc = true;
  return true;
}

// line 1: width, height ; line 2: descriptor count, scale factor, pyramid levels (read_write.cpp:62-89)
inline bool ReadImageInfo(const std::string& path, int& width, int& height, float& desc_scale, int& desc_level, int& desc_count) {
  std::ifstream f(path.c_str());
  if (!f.is_open()) return false;
  std::string line;
  std::getline(f, line);
  std::vector<std::string> s = SplitNonEmpty(line, ",");
  if (s.size() != 2) return false;
  width = atoi(s[0].c_str()); height = atoi(s[1].c_str());
  std::getline(f, line);
  s = SplitNonEmpty(line, ",");
  if (s.size() != 3) return false;
  desc_count = atoi(s[0].c_str()); desc_scale = (float)atof(s[1].c_str()); desc_level = atoi(s[2].c_str());
  return true;
}

// kitti.cc:56-84: one timestamp per non-empty line of <sequence>/times.txt, images <sequence>/image_0/%06d.<ext>
inline bool LoadImages(const std::string& sequence, std::vector<std::string>& filenames, std::vector<double>& timestamps,
                       const std::string& ext = "png") {
  std::ifstream f((sequence + "/times.txt").c_str());
  if (!f.is_open()) return false;
  timestamps.clear();
  std::string s;
  while (std::getline(f, s)) {
    if (s.empty()) continue;
    std::stringstream ss;
    ss << s;
    double t;
    ss >> t;
    timestamps.push_back(t);
  }
  filenames.resize(timestamps.size());
  for (size_t i = 0; i < timestamps.size(); ++i) {
    std::stringstream ss;
    ss << std::setfill('0') << std::setw(6) << i;
    filenames[i] = sequence + "/image_0/" + ss.str() + "." + ext;
  }
  return true;
}

// Eigen::Quaterniond(Matrix3d) as Converter::toQuaternion uses it (no normalisation, no sign convention), R row-major
inline void RotToQuaternion(const float Rf[9], float q[4]) {
  double R[9];
  for (int k = 0; k < 9; ++k) R[k] = Rf[k];
  double x, y, z, w;
  const double t = R[0] + R[4] + R[8];
  if (t > 0) {
    double s = std::sqrt(t + 1.0);
    w = 0.5 * s;
    s = 0.5 / s;
    x = (R[7] - R[5]) * s; y = (R[2] - R[6]) * s; z = (R[3] - R[1]) * s;
  } else {
    int i = 0;
    if (R[4] > R[0]) i = 1;
    if (R[8] > R[i * 4]) i = 2;
    const int j = (i + 1) % 3, k = (j + 1) % 3;
    double s = std::sqrt(R[i * 4] - R[j * 4] - R[k * 4] + 1.0);
    double v[3];
    v[i] = 0.5 * s;
    s = 0.5 / s;
    w = (R[k * 3 + j] - R[j * 3 + k]) * s;
    v[j] = (R[j * 3 + i] + R[i * 3 + j]) * s;
    v[k] = (R[k * 3 + i] + R[i * 3 + k]) * s;
    x = v[0]; y = v[1]; z = v[2];
  }
  q[0] = (float)x; q[1] = (float)y; q[2] = (float)z; q[3] = (float)w;
}

// System::SaveTrajectoryTUM line (System.cc:529): `fixed`, time with 6 decimals, camera centre and quaternion (floats) with 9
inline std::string TumLine(double timestamp, const float twc[3], const float q[4]) {
  std::ostringstream f;
  f << std::fixed;
  f << std::setprecision(6) << timestamp << " " << std::setprecision(9) << twc[0] << " " << twc[1] << " " << twc[2] << " " << q[0] << " "
    << q[1] << " " << q[2] << " " << q[3];
  return f.str();
}
// System::SaveKeyFrameTrajectoryTUM line (System.cc:470-471): time with 6 decimals, the rest with 10
inline std::string TumKeyFrameLine(double timestamp, const float t[3], const float q[4]) {
  std::ostringstream f;
  f << std::fixed;
  f << std::setprecision(6) << timestamp << std::setprecision(10) << " " << t[0] << " " << t[1] << " " << t[2] << " " << q[0] << " " << q[1]
    << " " << q[2] << " " << q[3];
  return f.str();
}
// pose of a frame as the trajectory files store it: Twc from Tcw (row-major 4x4), i.e. Rwc = Rcw^T, twc = -Rwc * tcw
inline void TcwToTumPose(const float Tcw[16], float twc[3], float q[4]) {
  float Rwc[9];
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 3; ++c) Rwc[r * 3 + c] = Tcw[c * 4 + r];
  for (int r = 0; r < 3; ++r) twc[r] = -(Rwc[r * 3] * Tcw[3] + Rwc[r * 3 + 1] * Tcw[7] + Rwc[r * 3 + 2] * Tcw[11]);
  RotToQuaternion(Rwc, q);
}

// binary PGM (P5, maxval <= 255)
inline bool ReadPGM(const std::string& path, std::vector<uint8_t>& pixels, int& width, int& height) {
  std::ifstream f(path.c_str(), std::ios::binary);
  if (!f.is_open()) return false;
  std::string magic;
  f >> magic;
  if (magic != "P5") return false;
  int vals[3], got = 0;
  while (got < 3) {
    f >> std::ws;
    if (f.peek() == '#') { std::string c; std::getline(f, c); continue; }
    if (!(f >> vals[got])) return false;
    ++got;
  }
  if (vals[2] > 255 || vals[0] <= 0 || vals[1] <= 0) return false;
  f.get();  // the single whitespace byte after maxval
  width = vals[0]; height = vals[1];
  pixels.resize((size_t)width * height);
  f.read(reinterpret_cast<char*>(pixels.data()), (std::streamsize)pixels.size());
  return (size_t)f.gcount() == pixels.size();
}

}  // namespace asd
