// test_replay_io.cpp -- command-line probe for replay_io.hpp, driven by tests/test_replay_io.py (CPU only).
//   test_replay_io vocdump <yaml> <out.bin> | vocwrite <in.bin> <yaml> |
//   test_replay_io cam <file> | imginfo <file> | images <sequence dir> | tum <t> <16 Tcw floats> | kftum <t> <16 Tcw floats> | pgm <file>
#include <cstdio>
#include <cstring>

#include "replay_io.hpp"
#include "voc_io.hpp"

int main(int argc, char** argv) {
  if (argc < 3) return 2;
  const std::string cmd = argv[1];
  if (cmd == "cam") {
    asd::CamInfo c;
    if (!asd::ReadCamInfo(argv[2], c)) return 1;
    printf("%.10g %.10g %.10g %.10g %.10g %.10g %.10g %.10g %d", c.fx, c.fy, c.cx, c.cy, c.distort[0], c.distort[1], c.distort[2], c.distort[3], (int)c.has_Tbc);
    for (int k = 0; k < 12; ++k) printf(" %.10g", c.Tbc[k]);
    printf("\n");
  } else if (cmd == "imginfo") {
    int w, h, lv, cnt; float sc;
    if (!asd::ReadImageInfo(argv[2], w, h, sc, lv, cnt)) return 1;
    printf("%d %d %d %.6g %d\n", w, h, cnt, sc, lv);
  } else if (cmd == "images") {
    std::vector<std::string> files; std::vector<double> times;
    if (!asd::LoadImages(argv[2], files, times)) return 1;
    for (size_t i = 0; i < files.size(); ++i) printf("%.9f %s\n", times[i], files[i].c_str());
  } else if ((cmd == "tum" || cmd == "kftum") && argc == 19) {
    float T[16], t[3], q[4];
    for (int k = 0; k < 16; ++k) T[k] = (float)atof(argv[3 + k]);
    asd::TcwToTumPose(T, t, q);
    printf("%s\n", (cmd == "tum" ? asd::TumLine(atof(argv[2]), t, q) : asd::TumKeyFrameLine(atof(argv[2]), t, q)).c_str());
  } else if (cmd == "pgm") {
    std::vector<uint8_t> px; int w, h;
    if (!asd::ReadPGM(argv[2], px, w, h)) return 1;
    unsigned long long s = 0;
    for (uint8_t v : px) s += v;
    printf("%d %d %llu\n", w, h, s);
  } else if (cmd == "vocdump" && argc == 4) {   // YAML -> flat binary: header ints, then the arrays
    asd::VocabularyArrays V;
    std::string err;
    if (!asd::ReadVocabulary(argv[2], V, &err)) { fprintf(stderr, "%s\n", err.c_str()); return 1; }
    FILE* o = fopen(argv[3], "wb");
    if (!o) return 1;
    const int32_t hdr[6] = {V.k, V.L, V.scoring, V.weighting, V.n_nodes, (int32_t)V.child_ids.size()};
    fwrite(hdr, 4, 6, o);
    fwrite(V.child_start.data(), 4, V.child_start.size(), o);
    fwrite(V.child_ids.data(), 4, V.child_ids.size(), o);
    fwrite(V.word_id.data(), 4, V.word_id.size(), o);
    fwrite(V.weight.data(), 8, V.weight.size(), o);
    fwrite(V.desc.data(), 4, V.desc.size(), o);
    fclose(o);
  } else if (cmd == "vocwrite" && argc == 4) {  // flat binary -> YAML
    FILE* in = fopen(argv[2], "rb");
    if (!in) return 1;
    int32_t hdr[6];
    if (fread(hdr, 4, 6, in) != 6) return 1;
    asd::VocabularyArrays V;
    V.k = hdr[0]; V.L = hdr[1]; V.scoring = hdr[2]; V.weighting = hdr[3]; V.n_nodes = hdr[4];
    V.child_start.resize(V.n_nodes + 1); V.child_ids.resize(hdr[5]); V.word_id.resize(V.n_nodes); V.weight.resize(V.n_nodes);
    V.desc.resize((size_t)V.n_nodes * 128);
    bool ok = fread(V.child_start.data(), 4, V.child_start.size(), in) == V.child_start.size() &&
              fread(V.child_ids.data(), 4, V.child_ids.size(), in) == V.child_ids.size() &&
              fread(V.word_id.data(), 4, V.word_id.size(), in) == V.word_id.size() &&
              fread(V.weight.data(), 8, V.weight.size(), in) == V.weight.size() &&
              fread(V.desc.data(), 4, V.desc.size(), in) == V.desc.size();
    fclose(in);
    if (!ok || !asd::WriteVocabulary(argv[3], V)) return 1;
  } else {
    return 2;
  }
  return 0;
}
