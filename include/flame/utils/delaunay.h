// include/flame/utils/delaunay.h -- Delaunay triangulation of the tracked features, the step of
// flame::Flame::update() right in front of the graph sync (upstream keeps a wrapper around
// Shewchuk's Triangle in flame/utils/delaunay.h; reference evidence: stat key `triangulate`,
// msg/FlameStats.msg:44, src/utils.cc; the mesh leaves as vector<flame::Triangle>,
// src/flame_offline_tum.cc:628-635).  It is NOT on the regulariser path and runs on the host: this
// header exists so that flame::Flame::FrontEnd::triangulate has a dependency-free default.
//
// Divide and conquer (Guibas & Stolfi 1985) with alternating vertical / horizontal cuts (Dwyer 1987) on a
// quad-edge structure kept in flat arrays; orientation and in-circle tests are EXACT: the coordinates are snapped to a
// 2^-16 pixel lattice (float pixel coordinates >= 128 are on it already) and the determinants are
// evaluated in 128-bit integers (|x|, |y| < 2^13 pixels: differences < 2^30, the in-circle sum
// < 2^124).  Collinear and cocircular inputs are therefore handled like any other: the result is always a
// triangulation of the convex hull in which no vertex lies strictly inside a circumcircle.  Points that
// coincide after snapping are triangulated once; the later copies are not referenced by any triangle.
//
// threads > 1 (r04): the recursion tree is cut at a level with ~2 x threads subtrees.  A persistent pool of worker
// threads (kept by the object between calls: starting a thread costs ~0.1 ms on a 256-core host, as much as a whole
// 1.2 k-point triangulation) first makes the cuts of the top levels (the same std::nth_element calls the serial recursion
// makes, level by level, the nodes of a level in parallel), then triangulates the subtrees -- every one allocates its
// quad-edges from a range of its own in the ONE shared array, so nothing is copied afterwards --, then merges
// pairs level by level, then reads the faces of every range.  The sort in front (snap, bucket by x, sort the buckets)
// and the face scan behind run on the same pool.  The triangle SET does not depend on the thread count; the order of the
// list does (deterministic for a given input and thread count).
//
// Cost on the GPU box's EPYC 9575F: see DESIGN.md (1 thread 0.33 / 3.6 / 19 ms at 1.2 k / 10 k / 50 k uniform points).
//
// Output: counter-clockwise triangles in the (u right, v down) image frame's coordinates, i.e.
// orient(a, b, c) > 0 with orient = (b - a) x (c - a), each starting at its smallest vertex.
#pragma once
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <condition_variable>
#include <cstdint>
#include <functional>
#include <memory>
#include <mutex>
#include <numeric>
#include <thread>
#include <utility>
#include <vector>

#include "../types.h"

namespace flame {
namespace utils {

#ifndef FLAME_DT_LEAF_FACTOR
#define FLAME_DT_LEAF_FACTOR 2
#endif
#ifndef FLAME_DT_SPIN_US
#define FLAME_DT_SPIN_US 2500  /* the workers spin this long between runs (they outlast the GPU tail and the getters of a frame of a
                                  back-to-back stream: waking 15 sleepers costs 0.1-0.2 ms), then sleep; at camera rate they sleep */
#endif
#if defined(__x86_64__) || defined(__i386__)
#define FLAME_DT_RELAX() __builtin_ia32_pause()
#else
#define FLAME_DT_RELAX() std::this_thread::yield()
#endif
namespace detail {
// A small persistent pool: run(n, fn) executes fn(0) .. fn(n - 1) on the workers and the calling thread and returns
// when all are done.  Workers spin for a short while between runs (the phases of one triangulation follow each other
// within microseconds) and sleep on a condition variable otherwise.
class SpinPool {
 public:
  explicit SpinPool(int workers) {
    for (int k = 0; k < workers; ++k) {
      bool ok = true;
#if defined(__cpp_exceptions)
      try { th_.emplace_back([this]() { loop(); }); } catch (...) { ok = false; }
#else
      th_.emplace_back([this]() { loop(); });
#endif
      if (!ok) break;
    }
  }
  ~SpinPool() {
    {
      std::lock_guard<std::mutex> lk(m_);
      stop_.store(true, std::memory_order_relaxed);
      gen_.fetch_add(1, std::memory_order_release);
    }
    cv_.notify_all();
    for (std::thread& t : th_) t.join();
  }
  SpinPool(const SpinPool&) = delete;
  SpinPool& operator=(const SpinPool&) = delete;
  int workers() const { return static_cast<int>(th_.size()); }

  // Every worker passes through every run exactly once (it takes tasks while there are any, then checks out), and run()
  // returns only when all have checked out: no worker can still be looking at fn_ / n_ / next_ when the next run
  // rewrites them, and what it reads was published by the release on gen_.
  void run(int n, const std::function<void(int)>& fn) {
    if (n <= 0) return;
    if (n == 1 || th_.empty()) { for (int i = 0; i < n; ++i) fn(i); return; }
    fn_ = &fn; n_ = n;
    next_.store(0, std::memory_order_relaxed);
    out_.store(0, std::memory_order_relaxed);
    {
      std::lock_guard<std::mutex> lk(m_);  // (a worker between its last spin and its wait must not miss this)
      gen_.fetch_add(1, std::memory_order_release);
    }
    if (sleepers_.load(std::memory_order_acquire) > 0) cv_.notify_all();
    work();
    const int w = static_cast<int>(th_.size());
    while (out_.load(std::memory_order_acquire) < w) FLAME_DT_RELAX();
  }

 private:
  void work() {
    for (;;) {
      const int i = next_.fetch_add(1, std::memory_order_relaxed);
      if (i >= n_) return;
      (*fn_)(i);
    }
  }
  void loop() {
    unsigned seen = 0;  // (gen_ at construction: a worker that starts late must still see the first run's bump)
    for (;;) {
      int spins = 0;
      std::chrono::steady_clock::time_point t0;
      while (gen_.load(std::memory_order_acquire) == seen) {
        FLAME_DT_RELAX();  // (the phases of one call are microseconds apart)
        if ((++spins & 255) != 0) continue;
        if (spins == 256) { t0 = std::chrono::steady_clock::now(); continue; }
        if (std::chrono::steady_clock::now() - t0 < std::chrono::microseconds(FLAME_DT_SPIN_US)) continue;
        std::unique_lock<std::mutex> lk(m_);
        sleepers_.fetch_add(1, std::memory_order_acq_rel);
        cv_.wait(lk, [&]() { return gen_.load(std::memory_order_acquire) != seen; });
        sleepers_.fetch_sub(1, std::memory_order_acq_rel);
      }
      seen = gen_.load(std::memory_order_acquire);
      if (stop_.load(std::memory_order_relaxed)) return;
      work();
      out_.fetch_add(1, std::memory_order_release);  // (everything this worker wrote is visible to run()'s caller)
    }
  }
  std::vector<std::thread> th_;
  std::mutex m_;
  std::condition_variable cv_;
  std::atomic<unsigned> gen_{0};
  std::atomic<int> next_{0}, out_{0}, sleepers_{0};
  std::atomic<bool> stop_{false};
  const std::function<void(int)>* fn_ = nullptr;
  int n_ = 0;
};
}  // namespace detail

class DelaunayTriangulator {
 public:
  // false: fewer than 3 distinct points, all points collinear, or a coordinate that is not finite /
  // beyond 2^13 pixels
#ifdef FLAME_DELAUNAY_TIMING
  double t_ms_[12] = {0};  // phase laps of the last call (tools/exp/delaunay_time.cc)
  std::chrono::steady_clock::time_point lap_;
#define FLAME_DT_LAP(k) { const auto now_ = std::chrono::steady_clock::now(); t_ms_[k] = std::chrono::duration<double, std::milli>(now_ - lap_).count(); lap_ = now_; }
#else
#define FLAME_DT_LAP(k)
#endif
  bool triangulate(const std::vector<Point2f>& pts, std::vector<Triangle>* out, int threads = 1) {
#ifdef FLAME_DELAUNAY_TIMING
    lap_ = std::chrono::steady_clock::now();
#endif
    out->clear();
    const int32_t n_in = static_cast<int32_t>(pts.size());
    const bool par = threads > 1 && n_in >= 1024;
    if (par && (!pool_ || pool_->workers() + 1 != threads)) pool_.reset(new detail::SpinPool(threads - 1));
    const int T = par ? pool_->workers() + 1 : 1;
    px_.resize(n_in); py_.resize(n_in);
    order_.resize(n_in);
    // ---- snap to the lattice ----
    std::atomic<bool> bad(false);
    const int chunks = par ? 4 * T : 1;
    auto snap = [&](int c) {
      const int32_t i0 = static_cast<int32_t>(static_cast<int64_t>(n_in) * c / chunks), i1 = static_cast<int32_t>(static_cast<int64_t>(n_in) * (c + 1) / chunks);
      for (int32_t i = i0; i < i1; ++i) {
        const double x = static_cast<double>(pts[i].x) * 65536.0, y = static_cast<double>(pts[i].y) * 65536.0;
        if (!(std::fabs(x) < 536870912.0) || !(std::fabs(y) < 536870912.0)) { bad.store(true); return; }  // 2^29
        px_[i] = static_cast<int64_t>(std::llround(x));
        py_[i] = static_cast<int64_t>(std::llround(y));
      }
    };
    if (par) pool_->run(chunks, snap); else snap(0);
    if (bad.load()) return false;
    FLAME_DT_LAP(0)
    // ---- the total order (x, y, id): one sort, or buckets along x sorted in parallel ----
    auto less = [&](int32_t a, int32_t b) {
      if (px_[a] != px_[b]) return px_[a] < px_[b];
      if (py_[a] != py_[b]) return py_[a] < py_[b];
      return a < b;
    };
    if (!par) {
      std::iota(order_.begin(), order_.end(), 0);
      std::sort(order_.begin(), order_.end(), less);
    } else {
      int64_t mn = px_[0], mx = px_[0];
      for (int32_t i = 1; i < n_in; ++i) { mn = std::min(mn, px_[i]); mx = std::max(mx, px_[i]); }
      const int nb = 8 * T;
      const int64_t span = mx - mn + 1;
      bstart_.assign(static_cast<size_t>(nb) + 1, 0);
      bucket_.resize(n_in);
      for (int32_t i = 0; i < n_in; ++i) {
        // (monotone in x: a bucket holds a range of x, so the buckets in order are the sorted sequence)
        const int b = static_cast<int>((static_cast<__int128>(px_[i] - mn) * nb) / span);
        bucket_[i] = b;
        ++bstart_[static_cast<size_t>(b) + 1];
      }
      for (int b = 0; b < nb; ++b) bstart_[static_cast<size_t>(b) + 1] += bstart_[static_cast<size_t>(b)];
      bcur_.assign(bstart_.begin(), bstart_.end() - 1);
      for (int32_t i = 0; i < n_in; ++i) order_[static_cast<size_t>(bcur_[static_cast<size_t>(bucket_[i])]++)] = i;
      pool_->run(nb, [&](int b) { std::sort(order_.begin() + bstart_[static_cast<size_t>(b)], order_.begin() + bstart_[static_cast<size_t>(b) + 1], less); });
    }
    FLAME_DT_LAP(1)
    // distinct points only (the first of every group of coinciding ones)
    int32_t n = 0;
    for (int32_t k = 0; k < n_in; ++k)
      if (k == 0 || px_[order_[k]] != px_[order_[n - 1]] || py_[order_[k]] != py_[order_[n - 1]]) order_[n++] = order_[k];
    order_.resize(n);
    if (n < 3) return false;
    // From here on a vertex is its RANK in the sorted order (neighbours in the plane are neighbours in
    // memory; the tests below walk one 16-byte record per vertex), mapped back when the faces are read.
    xy_.resize(static_cast<size_t>(2) * n);
    idx_.resize(n);
    auto ranks = [&](int c) {
      const int32_t k0 = static_cast<int32_t>(static_cast<int64_t>(n) * c / chunks), k1 = static_cast<int32_t>(static_cast<int64_t>(n) * (c + 1) / chunks);
      for (int32_t k = k0; k < k1; ++k) {
        xy_[2 * k] = static_cast<double>(px_[order_[k]]);  // (exact: integers below 2^30)
        xy_[2 * k + 1] = static_cast<double>(py_[order_[k]]);
        idx_[k] = k;
      }
    };
    if (par) pool_->run(chunks, ranks); else ranks(0);
    xyp_ = xy_.data();
    FLAME_DT_LAP(2)
    // levels of the recursion that are cut open: ~2 x threads subtrees of at least 256 points
    int levels = 0;
    if (par) while ((1 << levels) < FLAME_DT_LEAF_FACTOR * T && (n >> (levels + 1)) >= 256) ++levels;
    bool done = false;
    if (levels > 0) done = build_parallel(n, levels, out);
    if (!done) {  // serial (also the way out when a subtree outgrew its range of the edge array: never seen)
      out->clear();
      reserve_edges(static_cast<size_t>(8) * n + 64);
      Arena A;
      A.next = 0; A.end = static_cast<int32_t>(next_.size() / 4); A.growable = true;
      int32_t le, re;
      build(0, n, 0, &le, &re, &A, true);
      faces(0, A.next, out);
    }
    return !out->empty();
  }

 private:
  __extension__ typedef __int128 i128;
  // a range of quad-edge numbers one thread allocates from
  struct Arena {  // (a cache line of its own: `next` is written by its thread at every new edge)
    int32_t next = 0, end = 0;
    bool growable = false;
    char pad[64 - 2 * sizeof(int32_t) - sizeof(bool)];
  };
  struct Node {  // a node of the recursion tree's top levels
    int32_t lo = 0, hi = 0, le = 0, re = 0;
    int axis = 0;
    Arena* arena = nullptr;  // where its merge allocates: its leftmost leaf's range
  };
  // quad-edge: edge q occupies slots 4q .. 4q + 3 (rotations); next_ = Onext, org_ on the even slots
  std::vector<int32_t> next_, org_;
  std::vector<uint8_t> dead_;
  std::vector<int64_t> px_, py_;
  std::vector<int32_t> order_, bucket_;
  std::vector<int64_t> bstart_, bcur_;
  std::vector<double> xy_;  // (x, y) of the distinct points by rank
  const double* xyp_ = nullptr;
  std::vector<int32_t> idx_;  // the vertices (ranks) in the order of the recursion's cuts
  std::unique_ptr<detail::SpinPool> pool_;
  std::vector<Arena> arenas_;
  std::vector<Node> nodes_;
  std::vector<std::vector<Triangle> > pieces_;
  std::atomic<int32_t> spill_next_{0};
  int32_t spill_end_ = 0;
  std::atomic<bool> overflow_{false};

  void reserve_edges(size_t quads) {
    if (next_.size() < 4 * quads) { next_.resize(4 * quads); org_.resize(next_.size()); dead_.resize(next_.size() / 4); }
  }

  bool build_parallel(int32_t n, int levels, std::vector<Triangle>* out) {
    const int leaves = 1 << levels;
    // ~7.6 quad-edges per point are made (merges delete some again): 10 per point and subtree, a shared spill range behind
    std::vector<int32_t> base(static_cast<size_t>(leaves) + 1, 0);
    nodes_.assign(static_cast<size_t>(2) * leaves, Node());  // heap order: node 1 = root, children 2k, 2k + 1
    nodes_[1].lo = 0; nodes_[1].hi = n; nodes_[1].axis = 0;
    // ---- the cuts of the top levels: what the serial recursion does at these nodes, a level at a time ----
    for (int l = 0; l < levels; ++l) {
      const int first = 1 << l;
      pool_->run(first, [&](int k) {
        Node& nd = nodes_[static_cast<size_t>(first + k)];
        const int32_t mid = nd.lo + (nd.hi - nd.lo) / 2;
        const int axis = nd.axis;
        if (l > 0)  // (the root's cut is there already: idx_ starts as the ranks, i.e. in frame 0's order)
          std::nth_element(idx_.begin() + nd.lo, idx_.begin() + mid, idx_.begin() + nd.hi,
                           [&](int32_t a, int32_t b) { return before(a, b, axis); });
        Node& c0 = nodes_[static_cast<size_t>(2 * (first + k))];
        Node& c1 = nodes_[static_cast<size_t>(2 * (first + k) + 1)];
        c0.lo = nd.lo; c0.hi = mid; c0.axis = 1 - axis;
        c1.lo = mid; c1.hi = nd.hi; c1.axis = 1 - axis;
      });
    }
    FLAME_DT_LAP(3)
    arenas_.assign(static_cast<size_t>(leaves), Arena());
    for (int k = 0; k < leaves; ++k) {
      const Node& lf = nodes_[static_cast<size_t>(leaves + k)];
      base[static_cast<size_t>(k) + 1] = base[static_cast<size_t>(k)] + 10 * (lf.hi - lf.lo) + 64;
    }
    const int32_t spill = 3 * n + 1024;
    reserve_edges(static_cast<size_t>(base[static_cast<size_t>(leaves)]) + static_cast<size_t>(spill));
    spill_next_.store(base[static_cast<size_t>(leaves)]);
    spill_end_ = base[static_cast<size_t>(leaves)] + spill;
    overflow_.store(false);
    for (int k = 0; k < leaves; ++k) {
      arenas_[static_cast<size_t>(k)].next = base[static_cast<size_t>(k)];
      arenas_[static_cast<size_t>(k)].end = base[static_cast<size_t>(k) + 1];
      arenas_[static_cast<size_t>(k)].growable = false;
    }
    // every node merges into the range of its leftmost leaf
    for (int k = 1; k < 2 * leaves; ++k) {
      int lf = k;
      while (lf < leaves) lf *= 2;
      nodes_[static_cast<size_t>(k)].arena = &arenas_[static_cast<size_t>(lf - leaves)];
    }
    FLAME_DT_LAP(4)
    // ---- the subtrees ----
    pool_->run(leaves, [&](int k) {
      Node& lf = nodes_[static_cast<size_t>(leaves + k)];
      build(lf.lo, lf.hi, lf.axis, &lf.le, &lf.re, lf.arena);
    });
    FLAME_DT_LAP(5)
    // ---- merges, bottom-up ----
    for (int l = levels - 1; l >= 0; --l) {
      const int first = 1 << l;
      pool_->run(first, [&](int k) {
        Node& nd = nodes_[static_cast<size_t>(first + k)];
        merge(nd.axis, nodes_[static_cast<size_t>(2 * (first + k))].le, nodes_[static_cast<size_t>(2 * (first + k) + 1)].le, &nd.le, &nd.re, nd.arena);
      });
    }
    FLAME_DT_LAP(6)
    if (overflow_.load()) return false;
    // ---- faces of every range (the spill range last), put together in range order ----
    pieces_.resize(static_cast<size_t>(leaves) + 1);
    pool_->run(leaves + 1, [&](int k) {
      // (filled through a local vector: the headers of neighbouring pieces share cache lines, and push_back writes them)
      std::vector<Triangle> dst;
      dst.swap(pieces_[static_cast<size_t>(k)]);
      dst.clear();
      if (k < leaves) faces(base[static_cast<size_t>(k)], arenas_[static_cast<size_t>(k)].next, &dst);
      else faces(base[static_cast<size_t>(leaves)], std::min(spill_next_.load(), spill_end_), &dst);
      dst.swap(pieces_[static_cast<size_t>(k)]);
    });
    FLAME_DT_LAP(7)
    size_t total = 0;
    for (const auto& p : pieces_) total += p.size();
    out->reserve(total);
    for (const auto& p : pieces_) out->insert(out->end(), p.begin(), p.end());
    FLAME_DT_LAP(8)
    return true;
  }

  // every counter-clockwise 3-cycle of Lnext among the quad-edges [q0, q1), once
  void faces(int32_t q0, int32_t q1, std::vector<Triangle>* out) const {
    for (int32_t e = 4 * q0; e < 4 * q1; e += 2) {  // directed edges are the even slots
      if (dead_[e >> 2]) continue;
      const int32_t e1 = lnext(e), e2 = lnext(e1);
      if (lnext(e2) != e) continue;
      if (orient(org_[e], org_[e1], org_[e2]) <= 0) continue;  // (the outer face of a 3-point hull)
      const int32_t a = order_[org_[e]], b = order_[org_[e1]], c = order_[org_[e2]];
      if (!(a < b && a < c)) continue;  // the rotation that starts at the smallest vertex
      out->push_back(Triangle(a, b, c));
    }
  }

  static int32_t rot(int32_t e) { return (e & ~3) | ((e + 1) & 3); }
  static int32_t sym(int32_t e) { return (e & ~3) | ((e + 2) & 3); }
  static int32_t invrot(int32_t e) { return (e & ~3) | ((e + 3) & 3); }
  int32_t onext(int32_t e) const { return next_[e]; }
  int32_t oprev(int32_t e) const { return rot(next_[rot(e)]); }
  int32_t lnext(int32_t e) const { return rot(next_[invrot(e)]); }
  int32_t lprev(int32_t e) const { return sym(next_[e]); }
  int32_t rprev(int32_t e) const { return next_[sym(e)]; }
  int32_t dest(int32_t e) const { return org_[sym(e)]; }

  int32_t make_edge(int32_t a, int32_t b, Arena* A) {
    int32_t q;
    if (A->next < A->end) {
      q = A->next++;
    } else if (A->growable) {  // (serial: the one range is the whole array and grows with it)
      reserve_edges(2 * (next_.size() / 4) + 64);
      A->end = static_cast<int32_t>(next_.size() / 4);
      q = A->next++;
    } else {  // a subtree outgrew its range: the shared spill range; past that the call starts over serially
      q = spill_next_.fetch_add(1);
      if (q >= spill_end_) { overflow_.store(true); q = spill_end_ - 1; }
    }
    const int32_t e = 4 * q;
    next_[e] = e; next_[e + 1] = e + 3; next_[e + 2] = e + 2; next_[e + 3] = e + 1;
    org_[e] = a; org_[e + 1] = -1; org_[e + 2] = b; org_[e + 3] = -1;
    dead_[e >> 2] = 0;
    return e;
  }
  void splice(int32_t a, int32_t b) {
    const int32_t alpha = rot(next_[a]), beta = rot(next_[b]);
    std::swap(next_[a], next_[b]);
    std::swap(next_[alpha], next_[beta]);
  }
  int32_t connect(int32_t a, int32_t b, Arena* A) {
    const int32_t e = make_edge(dest(a), org_[b], A);
    splice(e, lnext(a));
    splice(sym(e), b);
    return e;
  }
  void remove(int32_t e) {
    splice(e, oprev(e));
    splice(sym(e), oprev(sym(e)));
    dead_[e >> 2] = 1;
  }

  int64_t ix(int32_t v) const { return static_cast<int64_t>(xyp_[2 * v]); }
  int64_t iy(int32_t v) const { return static_cast<int64_t>(xyp_[2 * v + 1]); }
  // > 0: a, b, c counter-clockwise ((b - a) x (c - a)).  Both tests try double precision first: the
  // differences are exact there (integers below 2^31), the rounding of the products is bounded by a few
  // ulps of the sum of their magnitudes (Shewchuk's static filter, constants rounded up); only a
  // determinant inside that bound -- a (nearly) degenerate configuration -- is re-evaluated exactly.
  int orient(int32_t a, int32_t b, int32_t c) const {
    {
      const double bax = xyp_[2 * b] - xyp_[2 * a], bay = xyp_[2 * b + 1] - xyp_[2 * a + 1];
      const double cax = xyp_[2 * c] - xyp_[2 * a], cay = xyp_[2 * c + 1] - xyp_[2 * a + 1];
      const double p1 = bax * cay, p2 = bay * cax, det = p1 - p2;
      const double bound = 4.0e-16 * (std::fabs(p1) + std::fabs(p2));
      if (det > bound) return 1;
      if (det < -bound) return -1;
    }
    const i128 d = static_cast<i128>(ix(b) - ix(a)) * (iy(c) - iy(a)) - static_cast<i128>(iy(b) - iy(a)) * (ix(c) - ix(a));
    return d > 0 ? 1 : (d < 0 ? -1 : 0);
  }
  // d strictly inside the circle through the counter-clockwise a, b, c
  bool in_circle(int32_t a, int32_t b, int32_t c, int32_t d) const {
    {
      const double dx = xyp_[2 * d], dy = xyp_[2 * d + 1];
      const double ax = xyp_[2 * a] - dx, ay = xyp_[2 * a + 1] - dy, bx = xyp_[2 * b] - dx, by = xyp_[2 * b + 1] - dy;
      const double cx = xyp_[2 * c] - dx, cy = xyp_[2 * c + 1] - dy;
      const double bc1 = bx * cy, bc2 = by * cx, ac1 = ax * cy, ac2 = ay * cx, ab1 = ax * by, ab2 = ay * bx;
      const double a2 = ax * ax + ay * ay, b2 = bx * bx + by * by, c2 = cx * cx + cy * cy;
      const double det = a2 * (bc1 - bc2) - b2 * (ac1 - ac2) + c2 * (ab1 - ab2);
      const double perm = a2 * (std::fabs(bc1) + std::fabs(bc2)) + b2 * (std::fabs(ac1) + std::fabs(ac2)) +
                          c2 * (std::fabs(ab1) + std::fabs(ab2));
      const double bound = 2.0e-15 * perm;  // (Shewchuk's iccerrboundA is 1.11e-15 x the same permanent)
      if (det > bound) return true;
      if (det < -bound) return false;
    }
    const i128 ax = ix(a) - ix(d), ay = iy(a) - iy(d), bx = ix(b) - ix(d), by = iy(b) - iy(d);
    const i128 cx = ix(c) - ix(d), cy = iy(c) - iy(d);
    const i128 a2 = ax * ax + ay * ay, b2 = bx * bx + by * by, c2 = cx * cx + cy * cy;  // < 2^61
    // (2 x 2 minors < 2^61, products < 2^122, their sum < 2^124)
    const i128 det = a2 * (bx * cy - by * cx) - b2 * (ax * cy - ay * cx) + c2 * (ax * by - ay * bx);
    return det > 0;
  }
  bool right_of(int32_t p, int32_t e) const { return orient(p, dest(e), org_[e]) > 0; }
  bool left_of(int32_t p, int32_t e) const { return orient(p, org_[e], dest(e)) > 0; }
  bool valid(int32_t e, int32_t basel) const { return right_of(dest(e), basel); }

  // The cuts alternate between vertical and horizontal (Dwyer 1987): strips that are cut one way only get
  // long thin hulls whose merges make and delete many more edges (measured: 1.4 x the time).  Frame 0 orders the plane by (x, y); frame 1 is the same plane turned by a quarter: (y, -x).
  // The merge below only uses orientation and in-circle tests, which a rotation does not change: all it
  // needs in frame f is the set L before the set R in f's order and the right handles -- the
  // counter-clockwise hull edge out of a set's FIRST vertex in f's order, the clockwise hull edge out of
  // its LAST one.  The children come back with the handles of the other frame; a walk around each hull
  // finds these.
  bool before(int32_t a, int32_t b, int axis) const {
    const double ax = xyp_[2 * a], ay = xyp_[2 * a + 1], bx = xyp_[2 * b], by = xyp_[2 * b + 1];
    if (axis == 0) return ax != bx ? ax < bx : ay < by;
    return ay != by ? ay < by : ax > bx;
  }
  // hull handles of frame `axis` from any counter-clockwise hull edge
  void handles(int32_t start, int axis, int32_t* le, int32_t* re) const {
    int32_t best_lo = start, best_hi = sym(start);
    int32_t e = start;
    do {
      if (before(org_[e], org_[best_lo], axis)) best_lo = e;
      if (before(org_[best_hi], dest(e), axis)) best_hi = sym(e);
      e = rprev(e);  // the next hull edge counter-clockwise (same outer face on the right)
    } while (e != start);
    *le = best_lo; *re = best_hi;
  }

  // triangulation of idx_[lo .. hi), cut along frame `axis`: *le = the counter-clockwise hull edge out of
  // the first vertex in that frame's order, *re = the clockwise hull edge out of the last one
  // (sorted: idx_[lo .. hi) is in frame `axis`' order already -- the root, whose vertices are the ranks)
  void build(int32_t lo, int32_t hi, int axis, int32_t* le, int32_t* re, Arena* A, bool sorted = false) {
    const int32_t n = hi - lo;
    if (n <= 3) std::sort(idx_.begin() + lo, idx_.begin() + hi, [&](int32_t a, int32_t b) { return before(a, b, axis); });
    if (n == 2) {
      const int32_t a = make_edge(idx_[lo], idx_[lo + 1], A);
      *le = a; *re = sym(a);
      return;
    }
    if (n == 3) {
      const int32_t s1 = idx_[lo], s2 = idx_[lo + 1], s3 = idx_[lo + 2];
      const int32_t a = make_edge(s1, s2, A), b = make_edge(s2, s3, A);
      splice(sym(a), b);
      const int o = orient(s1, s2, s3);
      if (o > 0) { connect(b, a, A); *le = a; *re = sym(b); }
      else if (o < 0) { const int32_t c = connect(b, a, A); *le = sym(c); *re = c; }
      else { *le = a; *re = sym(b); }
      return;
    }
    const int32_t mid = lo + n / 2;
    if (!sorted)
      std::nth_element(idx_.begin() + lo, idx_.begin() + mid, idx_.begin() + hi,
                       [&](int32_t a, int32_t b) { return before(a, b, axis); });
    int32_t l_le, l_re, r_le, r_re;
    build(lo, mid, 1 - axis, &l_le, &l_re, A);
    build(mid, hi, 1 - axis, &r_le, &r_re, A);
    merge(axis, l_le, r_le, le, re, A);
  }

  // joins the triangulations of a left and a right set of frame `axis` (given by their `le` handles of the OTHER frame)
  void merge(int axis, int32_t left_le, int32_t right_le, int32_t* le, int32_t* re, Arena* A) {
    int32_t ldo, ldi, rdi, rdo;
    handles(left_le, axis, &ldo, &ldi);
    handles(right_le, axis, &rdi, &rdo);
    // lower common tangent
    for (;;) {
      if (left_of(org_[rdi], ldi)) ldi = lnext(ldi);
      else if (right_of(org_[ldi], rdi)) rdi = rprev(rdi);
      else break;
    }
    int32_t basel = connect(sym(rdi), ldi, A);
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
      if (!lv || (rv && in_circle(dest(lcand), org_[lcand], org_[rcand], dest(rcand)))) basel = connect(rcand, sym(basel), A);
      else basel = connect(sym(basel), sym(lcand), A);
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
