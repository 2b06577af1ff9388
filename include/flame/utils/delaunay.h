// include/flame/utils/delaunay.h -- Delaunay triangulation of the tracked features, the step of
// flame::Flame::update() right in front of the graph sync (upstream keeps a wrapper around
// Shewchuk's Triangle in flame/utils/delaunay.h; reference evidence: stat key `triangulate`,
// msg/FlameStats.msg:44, src/utils.cc; the mesh leaves as vector<flame::Triangle>,
// src/flame_offline_tum.cc:628-635).  It is NOT on the regulariser path and runs on the host: this
// header exists so that flame::Flame::FrontEnd::triangulate has a dependency-free default.
//
// Divide and conquer over the points sorted by (x, y) (Guibas & Stolfi 1985) on a quad-edge structure
// kept in flat arrays; orientation and in-circle tests are EXACT: the coordinates are snapped to a
// 2^-16 pixel lattice (float pixel coordinates >= 128 are on it already) and the determinants are
// evaluated in 128-bit integers (|x|, |y| < 2^13 pixels: differences < 2^30, the in-circle sum
// < 2^124).  Collinear and cocircular inputs are therefore handled like any other: the result is always a
// triangulation of the convex hull in which no vertex lies strictly inside a circumcircle.  Points that
// coincide after snapping are triangulated once; the later copies are not referenced by any triangle.
//
// Output: counter-clockwise triangles in the (u right, v down) image frame's coordinates, i.e.
// orient(a, b, c) > 0 with orient = (b - a) x (c - a); sorted by (min vertex, ...) for a stable order.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <numeric>
#include <utility>
#include <vector>

#include "../types.h"

namespace flame {
namespace utils {

class DelaunayTriangulator {
 public:
  // false: fewer than 3 distinct points, all points collinear, or a coordinate that is not finite /
  // beyond 2^13 pixels
  bool triangulate(const std::vector<Point2f>& pts, std::vector<Triangle>* out) {
    out->clear();
    const int32_t n_in = static_cast<int32_t>(pts.size());
    px_.resize(n_in); py_.resize(n_in);
    for (int32_t i = 0; i < n_in; ++i) {
      const double x = static_cast<double>(pts[i].x) * 65536.0, y = static_cast<double>(pts[i].y) * 65536.0;
      if (!(std::fabs(x) < 536870912.0) || !(std::fabs(y) < 536870912.0)) return false;  // 2^29
      px_[i] = static_cast<int64_t>(std::llround(x));
      py_[i] = static_cast<int64_t>(std::llround(y));
    }
    order_.resize(n_in);
    std::iota(order_.begin(), order_.end(), 0);
    std::sort(order_.begin(), order_.end(), [&](int32_t a, int32_t b) {
      if (px_[a] != px_[b]) return px_[a] < px_[b];
      if (py_[a] != py_[b]) return py_[a] < py_[b];
      return a < b;
    });
    // distinct points only (the first of every group of coinciding ones)
    int32_t n = 0;
    for (int32_t k = 0; k < n_in; ++k)
      if (k == 0 || px_[order_[k]] != px_[order_[n - 1]] || py_[order_[k]] != py_[order_[n - 1]]) order_[n++] = order_[k];
    order_.resize(n);
    if (n < 3) return false;
    next_.clear(); org_.clear(); dead_.clear();
    next_.reserve(static_cast<size_t>(12) * n); org_.reserve(static_cast<size_t>(12) * n);
    int32_t le, re;
    build(0, n, &le, &re);
    // ---- faces: every counter-clockwise 3-cycle of Lnext, once ----
    const int32_t ne = static_cast<int32_t>(next_.size());
    for (int32_t e = 0; e < ne; e += 2) {  // directed edges are the even slots
      if (dead_[e >> 2]) continue;
      const int32_t e1 = lnext(e), e2 = lnext(e1);
      if (lnext(e2) != e) continue;
      const int32_t a = org_[e], b = org_[e1], c = org_[e2];
      if (!(a < b && a < c)) continue;  // the rotation that starts at the smallest vertex
      if (orient(a, b, c) <= 0) continue;  // (the outer face of a 3-point hull)
      out->push_back(Triangle(a, b, c));
    }
    std::sort(out->begin(), out->end(), [](const Triangle& s, const Triangle& t) {
      if (s[0] != t[0]) return s[0] < t[0];
      if (s[1] != t[1]) return s[1] < t[1];
      return s[2] < t[2];
    });
    return !out->empty();
  }

 private:
  typedef __int128 i128;
  // quad-edge: edge q occupies slots 4q .. 4q + 3 (rotations); next_ = Onext, org_ on the even slots
  std::vector<int32_t> next_, org_;
  std::vector<uint8_t> dead_;
  std::vector<int64_t> px_, py_;
  std::vector<int32_t> order_;

  static int32_t rot(int32_t e) { return (e & ~3) | ((e + 1) & 3); }
  static int32_t sym(int32_t e) { return (e & ~3) | ((e + 2) & 3); }
  static int32_t invrot(int32_t e) { return (e & ~3) | ((e + 3) & 3); }
  int32_t onext(int32_t e) const { return next_[e]; }
  int32_t oprev(int32_t e) const { return rot(next_[rot(e)]); }
  int32_t lnext(int32_t e) const { return rot(next_[invrot(e)]); }
  int32_t lprev(int32_t e) const { return sym(next_[e]); }
  int32_t rprev(int32_t e) const { return next_[sym(e)]; }
  int32_t dest(int32_t e) const { return org_[sym(e)]; }

  int32_t make_edge(int32_t a, int32_t b) {
    const int32_t e = static_cast<int32_t>(next_.size());
    next_.push_back(e); next_.push_back(e + 3); next_.push_back(e + 2); next_.push_back(e + 1);
    org_.push_back(a); org_.push_back(-1); org_.push_back(b); org_.push_back(-1);
    dead_.push_back(0);
    return e;
  }
  void splice(int32_t a, int32_t b) {
    const int32_t alpha = rot(next_[a]), beta = rot(next_[b]);
    std::swap(next_[a], next_[b]);
    std::swap(next_[alpha], next_[beta]);
  }
  int32_t connect(int32_t a, int32_t b) {
    const int32_t e = make_edge(dest(a), org_[b]);
    splice(e, lnext(a));
    splice(sym(e), b);
    return e;
  }
  void remove(int32_t e) {
    splice(e, oprev(e));
    splice(sym(e), oprev(sym(e)));
    dead_[e >> 2] = 1;
  }

  // > 0: a, b, c counter-clockwise ((b - a) x (c - a)).  Both tests try double precision first: the
  // differences are exact there (integers below 2^31), the rounding of the products is bounded by a few
  // ulps of the sum of their magnitudes (Shewchuk's static filter, constants rounded up); only a
  // determinant inside that bound -- a (nearly) degenerate configuration -- is re-evaluated exactly.
  int orient(int32_t a, int32_t b, int32_t c) const {
    {
      const double bax = static_cast<double>(px_[b] - px_[a]), bay = static_cast<double>(py_[b] - py_[a]);
      const double cax = static_cast<double>(px_[c] - px_[a]), cay = static_cast<double>(py_[c] - py_[a]);
      const double p1 = bax * cay, p2 = bay * cax, det = p1 - p2;
      const double bound = 4.0e-16 * (std::fabs(p1) + std::fabs(p2));
      if (det > bound) return 1;
      if (det < -bound) return -1;
    }
    const i128 d = static_cast<i128>(px_[b] - px_[a]) * (py_[c] - py_[a]) - static_cast<i128>(py_[b] - py_[a]) * (px_[c] - px_[a]);
    return d > 0 ? 1 : (d < 0 ? -1 : 0);
  }
  // d strictly inside the circle through the counter-clockwise a, b, c
  bool in_circle(int32_t a, int32_t b, int32_t c, int32_t d) const {
    {
      const double ax = static_cast<double>(px_[a] - px_[d]), ay = static_cast<double>(py_[a] - py_[d]);
      const double bx = static_cast<double>(px_[b] - px_[d]), by = static_cast<double>(py_[b] - py_[d]);
      const double cx = static_cast<double>(px_[c] - px_[d]), cy = static_cast<double>(py_[c] - py_[d]);
      const double bc1 = bx * cy, bc2 = by * cx, ac1 = ax * cy, ac2 = ay * cx, ab1 = ax * by, ab2 = ay * bx;
      const double a2 = ax * ax + ay * ay, b2 = bx * bx + by * by, c2 = cx * cx + cy * cy;
      const double det = a2 * (bc1 - bc2) - b2 * (ac1 - ac2) + c2 * (ab1 - ab2);
      const double perm = a2 * (std::fabs(bc1) + std::fabs(bc2)) + b2 * (std::fabs(ac1) + std::fabs(ac2)) +
                          c2 * (std::fabs(ab1) + std::fabs(ab2));
      const double bound = 2.0e-15 * perm;  // (Shewchuk's iccerrboundA is 1.11e-15 x the same permanent)
      if (det > bound) return true;
      if (det < -bound) return false;
    }
    const i128 ax = px_[a] - px_[d], ay = py_[a] - py_[d], bx = px_[b] - px_[d], by = py_[b] - py_[d];
    const i128 cx = px_[c] - px_[d], cy = py_[c] - py_[d];
    const i128 a2 = ax * ax + ay * ay, b2 = bx * bx + by * by, c2 = cx * cx + cy * cy;  // < 2^61
    // (2 x 2 minors < 2^61, products < 2^122, their sum < 2^124)
    const i128 det = a2 * (bx * cy - by * cx) - b2 * (ax * cy - ay * cx) + c2 * (ax * by - ay * bx);
    return det > 0;
  }
  bool right_of(int32_t p, int32_t e) const { return orient(p, dest(e), org_[e]) > 0; }
  bool left_of(int32_t p, int32_t e) const { return orient(p, org_[e], dest(e)) > 0; }
  bool valid(int32_t e, int32_t basel) const { return right_of(dest(e), basel); }

  // triangulation of order_[lo .. hi): *le = the counter-clockwise hull edge out of the leftmost
  // vertex, *re = the clockwise hull edge out of the rightmost vertex
  void build(int32_t lo, int32_t hi, int32_t* le, int32_t* re) {
    const int32_t n = hi - lo;
    if (n == 2) {
      const int32_t a = make_edge(order_[lo], order_[lo + 1]);
      *le = a; *re = sym(a);
      return;
    }
    if (n == 3) {
      const int32_t s1 = order_[lo], s2 = order_[lo + 1], s3 = order_[lo + 2];
      const int32_t a = make_edge(s1, s2), b = make_edge(s2, s3);
      splice(sym(a), b);
      const int o = orient(s1, s2, s3);
      if (o > 0) { connect(b, a); *le = a; *re = sym(b); }
      else if (o < 0) { const int32_t c = connect(b, a); *le = sym(c); *re = c; }
      else { *le = a; *re = sym(b); }
      return;
    }
    const int32_t mid = lo + n / 2;
    int32_t ldo, ldi, rdi, rdo;
    build(lo, mid, &ldo, &ldi);
    build(mid, hi, &rdi, &rdo);
    // lower common tangent
    for (;;) {
      if (left_of(org_[rdi], ldi)) ldi = lnext(ldi);
      else if (right_of(org_[ldi], rdi)) rdi = rprev(rdi);
      else break;
    }
    int32_t basel = connect(sym(rdi), ldi);
    if (org_[ldi] == org_[ldo]) ldo = sym(basel);
    if (org_[rdi] == org_[rdo]) rdo = basel;
    for (;;) {  // merge upwards
      int32_t lcand = onext(sym(basel));
      if (valid(lcand, basel))
        while (in_circle(dest(basel), org_[basel], dest(lcand), dest(onext(lcand)))) {
          const int32_t t = onext(lcand);
          remove(lcand);
          lcand = t;
        }
      int32_t rcand = oprev(basel);
      if (valid(rcand, basel))
        while (in_circle(dest(basel), org_[basel], dest(rcand), dest(oprev(rcand)))) {
          const int32_t t = oprev(rcand);
          remove(rcand);
          rcand = t;
        }
      const bool lv = valid(lcand, basel), rv = valid(rcand, basel);
      if (!lv && !rv) break;
      if (!lv || (rv && in_circle(dest(lcand), org_[lcand], org_[rcand], dest(rcand)))) basel = connect(rcand, sym(basel));
      else basel = connect(sym(basel), sym(lcand));
    }
    *le = ldo; *re = rdo;
  }
};

// Convenience: flame::utils::delaunay(points, &triangles)
inline bool delaunay(const std::vector<Point2f>& pts, std::vector<Triangle>* out) {
  DelaunayTriangulator t;
  return t.triangulate(pts, out);
}

}  // namespace utils
}  // namespace flame
