// tests/cpp/image_io_test.cc -- drives include/flame_ros/image_io.h for tests/test_image_io.py.
//   image_io_test decode  <image> <out.bin>            header (w h channels bit_depth as int32) + samples
//   image_io_test gray    <image> <out.bin>            8-bit gray as update() receives it
//   image_io_test rectify <image> <out.bin> fx fy cx cy k1 k2 p1 p2 k3   undistorted image, same layout as decode
//   image_io_test depth   <image> <out.bin> scale      float32 metres
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "flame_ros/image_io.h"

using namespace flame_ros::images;

static bool dump(const char* path, const Image& im) {
  FILE* f = std::fopen(path, "wb");
  if (!f) return false;
  const int32_t hdr[4] = {im.width, im.height, im.channels, im.bit_depth};
  std::fwrite(hdr, 4, 4, f);
  if (im.bit_depth == 8) std::fwrite(im.u8.data(), 1, im.u8.size(), f);
  else std::fwrite(im.u16.data(), 2, im.u16.size(), f);
  std::fclose(f);
  return true;
}

int main(int argc, char** argv) {
  if (argc < 4) return 2;
  if (std::string(argv[1]) == "inflate") {  // <zlib stream file> <out.bin>: raw inflate of a whole file
    std::vector<uint8_t> in, out;
    if (!readFile(argv[2], &in)) return 3;
    if (!inflate(in.data(), in.size(), &out)) return 6;
    FILE* f = std::fopen(argv[3], "wb");
    if (!f) return 4;
    std::fwrite(out.data(), 1, out.size(), f);
    std::fclose(f);
    return 0;
  }
  Image im;
  std::string err;
  if (!readImage(argv[2], &im, &err)) { std::fprintf(stderr, "%s\n", err.c_str()); return 3; }
  const std::string mode = argv[1];
  if (mode == "decode") return dump(argv[3], im) ? 0 : 4;
  if (mode == "gray") {
    std::vector<uint8_t> g;
    if (!toGray8(im, &g)) return 5;
    Image o; o.width = im.width; o.height = im.height; o.channels = 1; o.bit_depth = 8; o.u8.swap(g);
    return dump(argv[3], o) ? 0 : 4;
  }
  if (mode == "rectify" && argc == 13) {
    PlumbBob c;
    c.fx = std::atof(argv[4]); c.fy = std::atof(argv[5]); c.cx = std::atof(argv[6]); c.cy = std::atof(argv[7]);
    c.k1 = std::atof(argv[8]); c.k2 = std::atof(argv[9]); c.p1 = std::atof(argv[10]); c.p2 = std::atof(argv[11]);
    c.k3 = std::atof(argv[12]);
    Image o = im;
    if (im.bit_depth == 8) undistort<uint8_t>(im.u8.data(), im.width, im.height, im.channels, c, o.u8.data());
    else undistort<uint16_t>(im.u16.data(), im.width, im.height, im.channels, c, o.u16.data());
    return dump(argv[3], o) ? 0 : 4;
  }
  if (mode == "depth" && argc == 5 && im.bit_depth == 16 && im.channels == 1) {
    std::vector<float> d(im.u16.size());
    depthToFloat(im.u16.data(), d.size(), static_cast<float>(std::atof(argv[4])), d.data());
    FILE* f = std::fopen(argv[3], "wb");
    if (!f) return 4;
    std::fwrite(d.data(), 4, d.size(), f);
    std::fclose(f);
    return 0;
  }
  return 2;
}
