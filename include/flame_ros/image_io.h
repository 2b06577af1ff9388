// include/flame_ros/image_io.h -- the pixel side of the ROS-free dataset harness (SURVEY.md 8f row
// f4): what flame_ros' offline streams do with OpenCV between the file on disk and
// flame::Flame::update(), without OpenCV / libpng / zlib (none of them is part of this build):
//   * decode   cv::imread of the dataset images -- TUM: 8-bit RGB + 16-bit depth PNGs, EuRoC: 8-bit
//              gray PNGs (reference src/ros_sensor_streams/tum_rgbd_offline_stream.cc:248-300 parses
//              the file names, asl_rgbd_offline_stream.cc:282-296 reads them): a PNG reader (all five
//              row filters, colour types 0/2/4/6, bit depths 8/16, non-interlaced) over a small
//              inflate (RFC 1950/1951: stored, fixed and dynamic Huffman blocks), plus binary PGM/PPM
//   * gray     the BGR -> gray conversion the frontends apply before update() (OpenCV's fixed-point
//              weights: (4899 R + 9617 G + 1868 B + 8192) >> 14)
//   * rectify  model_.rectifyImage / cv::undistort (tum_rgbd_offline_stream.cc:196-200,
//              asl_rgbd_offline_stream.cc:285-287): plumb-bob (k1, k2, p1, p2, k3) undistortion onto
//              the same camera matrix, bilinear, zero border.  The build's own statement: float32
//              coordinates and weights (OpenCV's remap quantises the weights to 1/32 pixel, so its
//              output may differ by one gray level)
//   * depth    raw uint16 / depth_scale_factor -> float metres (tum_rgbd_offline_stream.cc:203-209,
//              asl_rgbd_offline_stream.cc:303-308)
// Header-only, C++11.  Nothing here is on the GPU hot path: it feeds the FrontEnd callbacks.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

namespace flame_ros {
namespace images {

constexpr long kMaxImageDim = 1 << 15;  // per side: anything larger is refused before sizes are computed from it

struct Image {
  int width = 0, height = 0, channels = 0, bit_depth = 0;  // 8: u8 holds the samples, 16: u16 (host order)
  std::vector<uint8_t> u8;
  std::vector<uint16_t> u16;
  size_t samples() const { return static_cast<size_t>(width) * height * channels; }
};

// ------------------------------------------------------------------------------------------
// inflate (zlib container, RFC 1950 / deflate RFC 1951); canonical-Huffman decoding by counts
// ------------------------------------------------------------------------------------------
namespace detail {
struct BitReader {
  const uint8_t* p; size_t n, pos; uint32_t buf; int cnt; bool bad;
  BitReader(const uint8_t* p_, size_t n_) : p(p_), n(n_), pos(0), buf(0), cnt(0), bad(false) {}
  int bits(int need) {
    uint32_t v = buf;
    while (cnt < need) {
      if (pos >= n) { bad = true; return 0; }
      v |= static_cast<uint32_t>(p[pos++]) << cnt;
      cnt += 8;
    }
    buf = need < 32 ? (v >> need) : 0;
    cnt -= need;
    return static_cast<int>(v & ((1u << need) - 1u));
  }
};
struct Huffman { short count[16]; short symbol[288]; };
inline bool buildHuffman(Huffman* h, const short* length, int n) {
  for (int l = 0; l <= 15; ++l) h->count[l] = 0;
  for (int s = 0; s < n; ++s) h->count[length[s]]++;
  int left = 1;
  for (int l = 1; l <= 15; ++l) { left <<= 1; left -= h->count[l]; if (left < 0) return false; }
  short offs[16];
  offs[1] = 0;
  for (int l = 1; l < 15; ++l) offs[l + 1] = static_cast<short>(offs[l] + h->count[l]);
  for (int s = 0; s < n; ++s) if (length[s]) h->symbol[offs[length[s]]++] = static_cast<short>(s);
  return true;
}
inline int decodeSymbol(BitReader* br, const Huffman& h) {
  int code = 0, first = 0, index = 0;
  for (int l = 1; l <= 15; ++l) {
    code |= br->bits(1);
    if (br->bad) return -1;
    const int c = h.count[l];
    if (code - c < first) return h.symbol[index + (code - first)];
    index += c; first += c; first <<= 1; code <<= 1;
  }
  return -1;
}
inline bool inflateCodes(BitReader* br, std::vector<uint8_t>* out, const Huffman& lit, const Huffman& dist, size_t max_out) {
  static const short lbase[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
  static const short lext[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
  static const short dbase[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
  static const short dext[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};
  for (;;) {
    int sym = decodeSymbol(br, lit);
    if (sym < 0) return false;
    if (out->size() > max_out) return false;  // (a stream that inflates past what the caller can use: refused, not followed)
    if (sym < 256) { out->push_back(static_cast<uint8_t>(sym)); continue; }
    if (sym == 256) return true;
    sym -= 257;
    if (sym >= 29) return false;
    const int len = lbase[sym] + br->bits(lext[sym]);
    const int ds = decodeSymbol(br, dist);
    if (ds < 0 || ds >= 30) return false;
    const size_t d = static_cast<size_t>(dbase[ds]) + static_cast<size_t>(br->bits(dext[ds]));
    if (br->bad || d > out->size()) return false;
    const size_t from = out->size() - d;
    for (int k = 0; k < len; ++k) out->push_back((*out)[from + k]);
  }
}
}  // namespace detail

// max_out: the output is refused once it grows past this many bytes (no unbounded inflate)
inline bool inflate(const uint8_t* src, size_t n, std::vector<uint8_t>* out, size_t max_out = static_cast<size_t>(-1)) {
  using namespace detail;
  if (n < 6 || (src[0] & 0x0f) != 8 || ((src[0] << 8) | src[1]) % 31 != 0 || (src[1] & 0x20)) return false;
  BitReader br(src + 2, n - 2);
  out->clear();
  int last;
  do {
    last = br.bits(1);
    const int type = br.bits(2);
    if (br.bad) return false;
    if (type == 0) {
      br.buf = 0; br.cnt = 0;  // to the byte boundary
      if (br.pos + 4 > br.n) return false;
      const unsigned len = br.p[br.pos] | (br.p[br.pos + 1] << 8), nlen = br.p[br.pos + 2] | (br.p[br.pos + 3] << 8);
      br.pos += 4;
      if ((len ^ 0xffffu) != nlen || br.pos + len > br.n) return false;
      if (out->size() + len > max_out) return false;
      out->insert(out->end(), br.p + br.pos, br.p + br.pos + len);
      br.pos += len;
    } else if (type == 1) {
      Huffman lit, dist;
      short l[288];
      for (int s = 0; s < 144; ++s) l[s] = 8;
      for (int s = 144; s < 256; ++s) l[s] = 9;
      for (int s = 256; s < 280; ++s) l[s] = 7;
      for (int s = 280; s < 288; ++s) l[s] = 8;
      buildHuffman(&lit, l, 288);
      for (int s = 0; s < 30; ++s) l[s] = 5;
      buildHuffman(&dist, l, 30);
      if (!inflateCodes(&br, out, lit, dist, max_out)) return false;
    } else if (type == 2) {
      static const short order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
      const int nlen = br.bits(5) + 257, ndist = br.bits(5) + 1, ncode = br.bits(4) + 4;
      if (br.bad || nlen > 286 || ndist > 30) return false;
      short l[320];
      for (int k = 0; k < 19; ++k) l[order[k]] = 0;
      for (int k = 0; k < ncode; ++k) l[order[k]] = static_cast<short>(br.bits(3));
      Huffman lencode;
      if (!buildHuffman(&lencode, l, 19)) return false;
      int idx = 0;
      while (idx < nlen + ndist) {
        int sym = decodeSymbol(&br, lencode);
        if (sym < 0) return false;
        if (sym < 16) { l[idx++] = static_cast<short>(sym); continue; }
        int rep, val = 0;
        if (sym == 16) { if (idx == 0) return false; val = l[idx - 1]; rep = 3 + br.bits(2); }
        else if (sym == 17) rep = 3 + br.bits(3);
        else rep = 11 + br.bits(7);
        if (br.bad || idx + rep > nlen + ndist) return false;
        while (rep--) l[idx++] = static_cast<short>(val);
      }
      if (l[256] == 0) return false;
      Huffman lit, dist;
      if (!buildHuffman(&lit, l, nlen)) return false;
      buildHuffman(&dist, l + nlen, ndist);  // (an incomplete distance code is legal)
      if (!inflateCodes(&br, out, lit, dist, max_out)) return false;
    } else {
      return false;
    }
  } while (!last);
  return !br.bad;
}

// ------------------------------------------------------------------------------------------
// PNG / PNM readers
// ------------------------------------------------------------------------------------------
inline bool readFile(const std::string& path, std::vector<uint8_t>* buf) {
  FILE* f = std::fopen(path.c_str(), "rb");
  if (!f) return false;
  std::fseek(f, 0, SEEK_END);
  const long n = std::ftell(f);
  std::fseek(f, 0, SEEK_SET);
  buf->resize(n > 0 ? static_cast<size_t>(n) : 0);
  const bool ok = n >= 0 && std::fread(buf->data(), 1, buf->size(), f) == buf->size();
  std::fclose(f);
  return ok;
}

inline bool decodePNG(const uint8_t* d, size_t n, Image* img, std::string* err) {
  static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
  auto fail = [&](const char* m) { if (err) *err = m; return false; };
  if (n < 8 || std::memcmp(d, sig, 8) != 0) return fail("not a PNG file");
  auto be32 = [&](size_t o) { return (static_cast<uint32_t>(d[o]) << 24) | (d[o + 1] << 16) | (d[o + 2] << 8) | d[o + 3]; };
  size_t off = 8;
  int ctype = -1, interlace = 0;
  std::vector<uint8_t> z;
  while (off + 12 <= n) {
    const uint32_t len = be32(off);
    if (off + 12 + len > n) return fail("truncated chunk");
    const char* type = reinterpret_cast<const char*>(d + off + 4);
    const uint8_t* body = d + off + 8;
    if (!std::memcmp(type, "IHDR", 4)) {
      if (len < 13) return fail("bad IHDR");
      img->width = static_cast<int>(be32(off + 8)); img->height = static_cast<int>(be32(off + 12));
      img->bit_depth = body[8]; ctype = body[9]; interlace = body[12];
    } else if (!std::memcmp(type, "IDAT", 4)) {
      z.insert(z.end(), body, body + len);
    } else if (!std::memcmp(type, "IEND", 4)) {
      break;
    }
    off += 12 + len;
  }
  if (ctype < 0 || img->width <= 0 || img->height <= 0) return fail("no IHDR");
  if (interlace) return fail("interlaced PNGs are not supported");
  if (img->bit_depth != 8 && img->bit_depth != 16) return fail("only 8- and 16-bit samples are supported");
  img->channels = ctype == 0 ? 1 : ctype == 2 ? 3 : ctype == 4 ? 2 : ctype == 6 ? 4 : 0;
  if (!img->channels) return fail("palette PNGs are not supported");
  // (IHDR is untrusted -- dimensions are bounded before any size is computed from them, and the inflate
  // output is bounded by what these dimensions can hold, plus slack for the 258-byte match that crosses the bound)
  if (img->width > kMaxImageDim || img->height > kMaxImageDim) return fail("image dimensions out of range");
  const size_t bpp = static_cast<size_t>(img->channels) * img->bit_depth / 8, stride = bpp * img->width;
  std::vector<uint8_t> raw;
  if (!inflate(z.data(), z.size(), &raw, (stride + 1) * img->height + 512)) return fail("corrupt zlib stream");
  if (raw.size() < (stride + 1) * img->height) return fail("short image data");
  std::vector<uint8_t> pix(stride * img->height);
  for (int y = 0; y < img->height; ++y) {  // undo the row filters (PNG spec 9.2)
    const uint8_t ft = raw[(stride + 1) * y];
    const uint8_t* in = raw.data() + (stride + 1) * y + 1;
    uint8_t* cur = pix.data() + stride * y;
    const uint8_t* up = y ? cur - stride : nullptr;
    for (size_t i = 0; i < stride; ++i) {
      const int a = i >= bpp ? cur[i - bpp] : 0, b = up ? up[i] : 0, c = (up && i >= bpp) ? up[i - bpp] : 0;
      int pred = 0;
      switch (ft) {
        case 0: pred = 0; break;
        case 1: pred = a; break;
        case 2: pred = b; break;
        case 3: pred = (a + b) >> 1; break;
        case 4: { const int p = a + b - c, pa = std::abs(p - a), pb = std::abs(p - b), pc = std::abs(p - c);
                  pred = (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c); break; }
        default: return fail("bad row filter");
      }
      cur[i] = static_cast<uint8_t>(in[i] + pred);
    }
  }
  if (img->bit_depth == 8) { img->u8.swap(pix); img->u16.clear(); }
  else {
    img->u16.resize(img->samples());
    for (size_t k = 0; k < img->u16.size(); ++k) img->u16[k] = static_cast<uint16_t>((pix[2 * k] << 8) | pix[2 * k + 1]);
    img->u8.clear();
  }
  return true;
}

inline bool decodePNM(const uint8_t* d, size_t n, Image* img, std::string* err) {  // binary P5 / P6
  auto fail = [&](const char* m) { if (err) *err = m; return false; };
  if (n < 2 || d[0] != 'P' || (d[1] != '5' && d[1] != '6')) return fail("not a binary PGM/PPM file");
  size_t off = 2;
  long v[3];
  for (int k = 0; k < 3; ++k) {
    for (;;) {  // whitespace and comments
      while (off < n && (d[off] == ' ' || d[off] == '\n' || d[off] == '\r' || d[off] == '\t')) ++off;
      if (off < n && d[off] == '#') { while (off < n && d[off] != '\n') ++off; continue; }
      break;
    }
    long x = 0; bool any = false;
    while (off < n && d[off] >= '0' && d[off] <= '9') { x = x * 10 + (d[off++] - '0'); any = true; if (x > (1l << 30)) return fail("bad PNM header"); }
    if (!any) return fail("bad PNM header");
    v[k] = x;
  }
  ++off;  // the single whitespace behind maxval
  if (v[0] < 1 || v[1] < 1 || v[0] > kMaxImageDim || v[1] > kMaxImageDim || v[2] < 1 || v[2] > 65535) return fail("bad PNM header");
  img->width = static_cast<int>(v[0]); img->height = static_cast<int>(v[1]);
  img->channels = d[1] == '5' ? 1 : 3;
  img->bit_depth = v[2] < 256 ? 8 : 16;
  const size_t need = img->samples() * (img->bit_depth / 8);
  if (off + need > n) return fail("short PNM data");
  if (img->bit_depth == 8) { img->u8.assign(d + off, d + off + need); img->u16.clear(); }
  else {
    img->u16.resize(img->samples());
    for (size_t k = 0; k < img->u16.size(); ++k) img->u16[k] = static_cast<uint16_t>((d[off + 2 * k] << 8) | d[off + 2 * k + 1]);
    img->u8.clear();
  }
  return true;
}

// cv::imread stand-in: PNG or binary PGM/PPM by magic number
inline bool readImage(const std::string& path, Image* img, std::string* err = nullptr) {
  std::vector<uint8_t> buf;
  if (!readFile(path, &buf)) { if (err) *err = "cannot read " + path; return false; }
  if (buf.size() >= 2 && buf[0] == 'P') return decodePNM(buf.data(), buf.size(), img, err);
  return decodePNG(buf.data(), buf.size(), img, err);
}

// 8-bit gray image as the frontends hand it to update() (cv::Mat1b): RGB(A) through OpenCV's
// fixed-point BGR2GRAY weights, gray(+alpha) as is, 16-bit samples by their high byte
inline bool toGray8(const Image& im, std::vector<uint8_t>* gray) {
  const size_t npx = static_cast<size_t>(im.width) * im.height;
  gray->resize(npx);
  auto s8 = [&](size_t k) -> int { return im.bit_depth == 8 ? im.u8[k] : (im.u16[k] >> 8); };
  if (im.channels == 1 || im.channels == 2) {
    for (size_t p = 0; p < npx; ++p) (*gray)[p] = static_cast<uint8_t>(s8(p * im.channels));
  } else if (im.channels == 3 || im.channels == 4) {
    for (size_t p = 0; p < npx; ++p) {
      const int r = s8(p * im.channels), g = s8(p * im.channels + 1), b = s8(p * im.channels + 2);
      (*gray)[p] = static_cast<uint8_t>((4899 * r + 9617 * g + 1868 * b + 8192) >> 14);
    }
  } else {
    return false;
  }
  return true;
}

// ------------------------------------------------------------------------------------------
// rectification: plumb-bob undistortion onto the same camera matrix (cv::undistort(src, dst, K, D))
// ------------------------------------------------------------------------------------------
struct PlumbBob {
  float fx = 1.f, fy = 1.f, cx = 0.f, cy = 0.f;          // K
  float k1 = 0.f, k2 = 0.f, p1 = 0.f, p2 = 0.f, k3 = 0.f;  // D (OpenCV order k1 k2 p1 p2 k3)
};

// source position (in the distorted image) of the undistorted pixel (u, v)
inline void distortPoint(const PlumbBob& c, float u, float v, float* su, float* sv) {
  const float x = (u - c.cx) / c.fx, y = (v - c.cy) / c.fy;
  const float r2 = x * x + y * y;
  const float radial = 1.0f + r2 * (c.k1 + r2 * (c.k2 + r2 * c.k3));
  const float xd = x * radial + 2.0f * c.p1 * x * y + c.p2 * (r2 + 2.0f * x * x);
  const float yd = y * radial + c.p1 * (r2 + 2.0f * y * y) + 2.0f * c.p2 * x * y;
  *su = c.fx * xd + c.cx;
  *sv = c.fy * yd + c.cy;
}

// dst(v, u) = bilinear sample of src at distortPoint(u, v); samples outside the image read 0;
// T = uint8_t (rounded to nearest) / uint16_t (rounded) / float; interleaved channels
template <class T>
inline void undistort(const T* src, int width, int height, int channels, const PlumbBob& cam, T* dst) {
  for (int v = 0; v < height; ++v)
    for (int u = 0; u < width; ++u) {
      float su, sv;
      distortPoint(cam, static_cast<float>(u), static_cast<float>(v), &su, &sv);
      const float fx0 = std::floor(su), fy0 = std::floor(sv);
      const int x0 = static_cast<int>(fx0), y0 = static_cast<int>(fy0);
      const float ax = su - fx0, ay = sv - fy0;
      for (int ch = 0; ch < channels; ++ch) {
        auto at = [&](int xx, int yy) -> float {
          return (xx < 0 || yy < 0 || xx >= width || yy >= height)
                     ? 0.0f : static_cast<float>(src[(static_cast<size_t>(yy) * width + xx) * channels + ch]);
        };
        const float top = at(x0, y0) + ax * (at(x0 + 1, y0) - at(x0, y0));
        const float bot = at(x0, y0 + 1) + ax * (at(x0 + 1, y0 + 1) - at(x0, y0 + 1));
        const float val = top + ay * (bot - top);
        T* o = dst + (static_cast<size_t>(v) * width + u) * channels + ch;
        if (static_cast<T>(0.5f) == static_cast<T>(0)) *o = static_cast<T>(val + 0.5f);  // integer sample types
        else *o = static_cast<T>(val);
      }
    }
}

// raw depth image (uint16) -> metres: reference tum_rgbd_offline_stream.cc:203-209
inline void depthToFloat(const uint16_t* raw, size_t n, float depth_scale_factor, float* out) {
  for (size_t k = 0; k < n; ++k) out[k] = static_cast<float>(raw[k]) / depth_scale_factor;
}

}  // namespace images
}  // namespace flame_ros
