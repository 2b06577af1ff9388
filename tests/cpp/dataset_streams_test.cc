// tests/cpp/dataset_streams_test.cc -- drives include/flame_ros/dataset_streams.h on the fixtures
// written by tests/test_dataset_streams.py and prints every frame, one line each:
//   tum <id> <time> <qw qx qy qz> <tx ty tz> <has_depth> <rgb path> [<depth path>]
//   asl <id> <time> <qw qx qy qz> <tx ty tz> <rgb path> [<depth path>]
// usage: dataset_streams_test tum <index file> <frame> | asl <pose dir> <rgb dir> <depth dir|-> <frame>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "flame_ros/dataset_streams.h"

namespace ds = flame_ros::datasets;

static ds::Frame frameOf(const char* s) {
  const char* names[] = {"RDF", "FLU", "FRD", "RDF_IN_FLU", "RDF_IN_FRD", "RFU"};
  for (int k = 0; k < 6; ++k)
    if (!std::strcmp(s, names[k])) return static_cast<ds::Frame>(k);
  return ds::RDF;
}

int main(int argc, char** argv) {
  if (argc >= 4 && !std::strcmp(argv[1], "tum")) {
    ds::TumIndex idx(argv[2], frameOf(argv[3]));
    std::printf("frames %zu scale %.1f\n", idx.size(), idx.depthScaleFactor());
    uint32_t id;
    ds::TumFrame f;
    while (idx.get(&id, &f))
      std::printf("tum %u %.9f %.17g %.17g %.17g %.17g %.17g %.17g %.17g %d %s %s\n", id, f.time, f.pose_optical.q.w,
                  f.pose_optical.q.x, f.pose_optical.q.y, f.pose_optical.q.z, f.pose_optical.t[0], f.pose_optical.t[1],
                  f.pose_optical.t[2], f.has_depth ? 1 : 0, f.rgb_file.c_str(), f.depth_file.c_str());
    return idx.empty() ? 0 : 2;
  }
  if (argc >= 5 && !std::strcmp(argv[1], "tumpix")) {
    // pixels of every frame of a TUM sequence: <index> <frame> <out.bin> [fx fy cx cy k1 k2 p1 p2 k3]
    ds::TumIndex idx(argv[2], frameOf(argv[3]));
    flame_ros::images::PlumbBob cam;
    const bool rect = argc >= 14;
    if (rect) {
      cam.fx = std::atof(argv[5]); cam.fy = std::atof(argv[6]); cam.cx = std::atof(argv[7]); cam.cy = std::atof(argv[8]);
      cam.k1 = std::atof(argv[9]); cam.k2 = std::atof(argv[10]); cam.p1 = std::atof(argv[11]); cam.p2 = std::atof(argv[12]);
      cam.k3 = std::atof(argv[13]);
    }
    FILE* f = std::fopen(argv[4], "wb");
    if (!f) return 4;
    uint32_t id;
    ds::TumFrame fr;
    while (idx.get(&id, &fr)) {
      int w = 0, h = 0;
      std::vector<uint8_t> gray;
      std::vector<float> depth;
      std::string err;
      if (!ds::loadFramePixels(fr.rgb_file, fr.has_depth ? fr.depth_file : std::string(), idx.depthScaleFactor(),
                               rect ? &cam : nullptr, true, &w, &h, &gray, &depth, &err)) {
        std::fprintf(stderr, "%s\n", err.c_str());
        std::fclose(f);
        return 5;
      }
      const int32_t hdr[3] = {w, h, depth.empty() ? 0 : 1};
      std::fwrite(hdr, 4, 3, f);
      std::fwrite(gray.data(), 1, gray.size(), f);
      std::fwrite(depth.data(), 4, depth.size(), f);
    }
    std::fclose(f);
    return 0;
  }
  if (argc >= 6 && !std::strcmp(argv[1], "asl")) {
    ds::AslDataset d(argv[2], argv[3], std::strcmp(argv[4], "-") ? argv[4] : "", frameOf(argv[5]));
    if (!d.ok()) return 3;
    std::printf("frames %zu w %d h %d K %.9g %.9g %.9g %.9g D %.9g %.9g %.9g %.9g %.9g scale %.9g\n", d.size(), d.width(),
                d.height(), d.K()[0], d.K()[4], d.K()[2], d.K()[5], d.D()[0], d.D()[1], d.D()[2], d.D()[3], d.D()[4],
                d.depthScaleFactor());
    uint32_t id;
    ds::AslFrame f;
    while (d.get(&id, &f))
      std::printf("asl %u %.9f %.17g %.17g %.17g %.17g %.17g %.17g %.17g %s %s\n", id, f.time, f.pose_optical.q.w,
                  f.pose_optical.q.x, f.pose_optical.q.y, f.pose_optical.q.z, f.pose_optical.t[0], f.pose_optical.t[1],
                  f.pose_optical.t[2], f.rgb_file.c_str(), f.depth_file.c_str());
    return 0;
  }
  return 1;
}
