// include/flame/types.h -- value types crossing the flame::Flame boundary.
//
// flame_ros passes OpenCV / Eigen / Sophus types (reference src/flame_offline_tum.cc:565-567,
// 578-579, 628-635, 643: Sophus::SE3f, cv::Mat1b, cv::Mat1f, std::vector<cv::Point2f>,
// std::vector<Eigen::Vector3f>, std::vector<flame::Triangle>, std::vector<flame::Edge>).  When
// those libraries are on the include path the aliases below ARE those types, so flame_ros compiles
// against this header unchanged; otherwise layout-compatible minimal structs are used (this image
// has none of the three libraries; tests/cpp/standins/ holds API stand-ins that exercise the
// library branch).
#pragma once
#include <array>
#include <cstddef>
#include <cstdint>
#include <vector>

#if defined(__has_include)
#if __has_include(<opencv2/core/core.hpp>)
#include <opencv2/core/core.hpp>
#define FLAME_HAVE_OPENCV 1
#endif
#if __has_include(<Eigen/Core>)
#include <Eigen/Core>
#define FLAME_HAVE_EIGEN 1
#endif
#if __has_include(<sophus/se3.hpp>)
#include <sophus/se3.hpp>
#define FLAME_HAVE_SOPHUS 1
#endif
#endif

namespace flame {

#ifdef FLAME_HAVE_OPENCV
using Point2f = cv::Point2f;
using Triangle = cv::Vec3i;  // indexable [0..2] -> vertex index (reference src/utils.cc:224-226)
using Edge = cv::Vec2i;
using Vec3b = cv::Vec3b;
using Image1b = cv::Mat1b;
using Image1f = cv::Mat1f;
using Image3b = cv::Mat3b;
#else
struct Point2f {
  float x = 0.f, y = 0.f;
  Point2f() = default;
  Point2f(float x_, float y_) : x(x_), y(y_) {}
};
struct Triangle {
  int32_t v[3] = {0, 0, 0};
  Triangle() = default;
  Triangle(int32_t a, int32_t b, int32_t c) : v{a, b, c} {}
  int32_t& operator[](int i) { return v[i]; }
  const int32_t& operator[](int i) const { return v[i]; }
};
struct Edge {
  int32_t v[2] = {0, 0};
  Edge() = default;
  Edge(int32_t a, int32_t b) : v{a, b} {}
  int32_t& operator[](int i) { return v[i]; }
  const int32_t& operator[](int i) const { return v[i]; }
};
struct Vec3b {
  uint8_t v[3] = {0, 0, 0};
  Vec3b() = default;
  Vec3b(uint8_t a, uint8_t b, uint8_t c) : v{a, b, c} {}
  uint8_t& operator[](int i) { return v[i]; }
  const uint8_t& operator[](int i) const { return v[i]; }
};
// row-major dense image with the subset of the cv::Mat_ interface the boundary needs
template <class T>
struct ImageT {
  int rows = 0, cols = 0;
  std::vector<T> data;
  ImageT() = default;
  ImageT(int r, int c) : rows(r), cols(c), data(static_cast<size_t>(r) * c) {}
  ImageT(int r, int c, const T& v) : rows(r), cols(c), data(static_cast<size_t>(r) * c, v) {}
  void create(int r, int c) { rows = r; cols = c; data.resize(static_cast<size_t>(r) * c); }
  bool empty() const { return data.empty(); }
  T& operator()(int r, int c) { return data[static_cast<size_t>(r) * cols + c]; }
  const T& operator()(int r, int c) const { return data[static_cast<size_t>(r) * cols + c]; }
  template <class U> U* ptr(int r = 0) { return reinterpret_cast<U*>(data.data() + static_cast<size_t>(r) * cols); }
  template <class U> const U* ptr(int r = 0) const { return reinterpret_cast<const U*>(data.data() + static_cast<size_t>(r) * cols); }
};
using Image1b = ImageT<uint8_t>;
using Image1f = ImageT<float>;
using Image3b = ImageT<Vec3b>;
#endif

#ifdef FLAME_HAVE_EIGEN
using Vector3f = Eigen::Vector3f;
using Matrix3f = Eigen::Matrix3f;
inline void toRowMajor(const Matrix3f& M, float out[9]) {
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 3; ++c) out[3 * r + c] = M(r, c);
}
#else
struct Vector3f {
  float v[3] = {0.f, 0.f, 0.f};
  float& operator()(int i) { return v[i]; }
  const float& operator()(int i) const { return v[i]; }
};
// row-major 3x3, operator()(r, c) like Eigen
struct Matrix3f {
  float m[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
  float& operator()(int r, int c) { return m[3 * r + c]; }
  const float& operator()(int r, int c) const { return m[3 * r + c]; }
};
inline void toRowMajor(const Matrix3f& M, float out[9]) {
  for (int k = 0; k < 9; ++k) out[k] = M.m[k];
}
#endif

#ifdef FLAME_HAVE_SOPHUS
using SE3f = Sophus::SE3f;
#else
// camera pose: unit quaternion (x, y, z, w) + translation; only carried to the feature front end
struct SE3f {
  float q[4] = {0.f, 0.f, 0.f, 1.f};
  float t[3] = {0.f, 0.f, 0.f};
};
#endif

}  // namespace flame
