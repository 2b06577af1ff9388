// TEST STAND-IN for <opencv2/core/core.hpp> (see ../../README.md): the interface subset that
// flame_ros' call sites and include/flame/ use.
#pragma once
#include <cstddef>
#include <cstdint>
#include <memory>
#include <vector>

namespace cv {

template <class T> struct Point_ {
  T x, y;
  Point_() : x(0), y(0) {}
  Point_(T x_, T y_) : x(x_), y(y_) {}
};
typedef Point_<float> Point2f;

template <class T, int N> struct Vec {
  T val[N];
  Vec() { for (int i = 0; i < N; ++i) val[i] = T(0); }
  Vec(T a, T b) { static_assert(N == 2, "Vec2"); val[0] = a; val[1] = b; }
  Vec(T a, T b, T c) { static_assert(N == 3, "Vec3"); val[0] = a; val[1] = b; val[2] = c; }
  T& operator[](int i) { return val[i]; }
  const T& operator[](int i) const { return val[i]; }
};
typedef Vec<unsigned char, 3> Vec3b;
typedef Vec<int, 3> Vec3i;
typedef Vec<int, 2> Vec2i;

// dense row-major matrix with shared storage (copies share the buffer, like cv::Mat)
template <class T> class Mat_ {
 public:
  int rows, cols;
  Mat_() : rows(0), cols(0) {}
  Mat_(int r, int c) : rows(0), cols(0) { create(r, c); }
  Mat_(int r, int c, const T& v) : rows(0), cols(0) { create(r, c); for (size_t k = 0; k < buf_->size(); ++k) (*buf_)[k] = v; }
  void create(int r, int c) {
    if (r == rows && c == cols && buf_) return;
    rows = r; cols = c;
    buf_ = std::make_shared<std::vector<T> >(static_cast<size_t>(r) * c);
  }
  bool empty() const { return !buf_ || buf_->empty(); }
  T& operator()(int r, int c) { return (*buf_)[static_cast<size_t>(r) * cols + c]; }
  const T& operator()(int r, int c) const { return (*buf_)[static_cast<size_t>(r) * cols + c]; }
  template <class U> U* ptr(int r = 0) { return reinterpret_cast<U*>(buf_->data() + static_cast<size_t>(r) * cols); }
  template <class U> const U* ptr(int r = 0) const { return reinterpret_cast<const U*>(buf_->data() + static_cast<size_t>(r) * cols); }
  const void* data() const { return buf_ ? buf_->data() : nullptr; }

 private:
  std::shared_ptr<std::vector<T> > buf_;
};
typedef Mat_<unsigned char> Mat1b;
typedef Mat_<float> Mat1f;
typedef Mat_<Vec3b> Mat3b;

// type-erased view, what cv_bridge::CvImage(hdr, "bgr8", mat) takes
class Mat {
 public:
  int rows, cols;
  Mat() : rows(0), cols(0), elem_(0), data_(nullptr) {}
  template <class T> Mat(const Mat_<T>& m) : rows(m.rows), cols(m.cols), elem_(sizeof(T)), data_(m.data()) {}  // NOLINT
  size_t elemSize() const { return elem_; }
  const void* data() const { return data_; }

 private:
  size_t elem_;
  const void* data_;
};

}  // namespace cv
