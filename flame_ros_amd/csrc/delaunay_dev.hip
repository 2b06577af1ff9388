// flame_ros_amd/csrc/delaunay_dev.hip -- Delaunay triangulation of a frame's features on the GPU (SURVEY.md 8 row
// f3's first leg; the reference budgets it as `triangulate` beside `sync_graph`, msg/FlameStats.msg:43-44; upstream
// calls Shewchuk's Triangle on the host).  include/flame/utils/delaunay.h is the host triangulator of the same contract
// (exact predicates on a 2^-16 pixel lattice, |u|, |v| < 2^13); this file is the device one.
//
// One thread per point builds that point's STAR -- its Delaunay neighbours in angular order -- by gift wrapping, from
// exact predicates only:
//   * the nearest neighbour q0 of p is a Delaunay neighbour of p in every Delaunay triangulation;
//   * given a Delaunay edge p -> q, the third vertex of the triangle on its left is the point r strictly left of p -> q
//     whose circle (p, q, r) holds no other point on that side: one pass over the candidates, "r' beats r when r' is
//     strictly inside (p, q, r)";
//   * when nothing is strictly left of p -> q the edge is on the convex hull: the star is open, and is completed by
//     wrapping the other way round from q0.
// Cocircular points (pixel lattices are full of them) are resolved by ONE rule every star applies alike: the polygon of
// the points on an empty circle is triangulated as a fan from its smallest vertex id.  A star sees the polygon either
// whole (from one of its boundary edges) or as the part left of one of the fan's diagonals, in which the smallest id is
// still a vertex: the rule restricted to the part is the same fan, so the stars agree and every triangle appears in the
// stars of its three vertices.  It is written once, by the star of its smallest vertex.
//
// Candidates come from a uniform grid (about two points per cell).  Floating point is used for PRUNING only, always
// conservatively: a cell row / cell is skipped when it cannot meet the current cap (the part of the current circle's
// disk left of p -> q; every later cap is inside it), with the disk's centre and radius padded by their rounding
// bounds.  Hull edges query a whole half-plane: per grid row the x extent of its points makes that one test per row.
// Every accept / reject of a candidate is exact: orientation in 64-bit integers (differences < 2^30), in-circle by a
// double-precision filter and 128-bit integers behind it (sum < 2^124), exactly as the host triangulator.
//
// Output: counter-clockwise triangles (orient = (b - a) x (c - a) > 0), each starting at its smallest vertex, ordered by
// that vertex, a star's triangles in wrapping order -- a function of the input alone (not of the order in which the
// scatter filled the cells).  Points that coincide after snapping are triangulated once (smallest id).
// The launcher checks Euler's relation T = 2 n - 2 - h (n live points, h boundary vertices) before it hands
// the list out.
#include "delaunay_dev.h"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstring>

#include "../../include/flame_hip.h"

namespace flamehip {
namespace {

__extension__ typedef __int128 i128;

#define DT_HIPCHK(expr)                                          \
  do {                                                           \
    hipError_t e__ = (expr);                                     \
    if (e__ != hipSuccess) return FLAME_HIP_ERR_HIP - (int)e__;  \
  } while (0)

// flags[0] error bits, [1] boundary vertices, [2] later copies of a point, [3] triangles,
// [4] min x, [5] min y, [6] max x, [7] max y (lattice)
constexpr int kFlagWords = 8;
constexpr int kErrRange = 1, kErrWrap = 2, kErrCap = 4, kErrCount = 8;

struct DtView {
  const int2* sxy;       // lattice coordinates, cell by cell
  const int32_t* sid;    // slot -> point id; ~id for a later copy of another point
  const int32_t* start;  // G*G + 1 cell offsets (row-major: a grid row's cells are contiguous)
  const int2* rowx;      // per grid row {min x, max x} of its live points ({1, 0}: none)
  const int32_t* flags;
  int32_t G, V;
};

struct DtBox {
  int32_t minx, miny;
  int64_t spanx, spany;
};

__device__ inline DtBox dt_box(const int32_t* flags) {
  DtBox b;
  b.minx = flags[4]; b.miny = flags[5];
  b.spanx = (int64_t)flags[6] - flags[4] + 1; b.spany = (int64_t)flags[7] - flags[5] + 1;
  return b;
}

__device__ inline int64_t orient64(int2 a, int2 b, int2 c) {
  return (int64_t)(b.x - a.x) * (int64_t)(c.y - a.y) - (int64_t)(b.y - a.y) * (int64_t)(c.x - a.x);
}
__device__ inline int sgn64(int64_t v) { return v > 0 ? 1 : (v < 0 ? -1 : 0); }

// > 0: d strictly inside the circle through a, b, c when those are counter-clockwise (the sign flips with their
// orientation); 0: on it.  Filter and exact evaluation as include/flame/utils/delaunay.h in_circle.
__device__ inline int incircle_sign(int2 a, int2 b, int2 c, int2 d) {
  {
    const double ax = (double)(a.x - d.x), ay = (double)(a.y - d.y), bx = (double)(b.x - d.x), by = (double)(b.y - d.y);
    const double cx = (double)(c.x - d.x), cy = (double)(c.y - d.y);
    const double bc1 = bx * cy, bc2 = by * cx, ac1 = ax * cy, ac2 = ay * cx, ab1 = ax * by, ab2 = ay * bx;
    const double a2 = ax * ax + ay * ay, b2 = bx * bx + by * by, c2 = cx * cx + cy * cy;
    const double det = a2 * (bc1 - bc2) - b2 * (ac1 - ac2) + c2 * (ab1 - ab2);
    const double perm = a2 * (fabs(bc1) + fabs(bc2)) + b2 * (fabs(ac1) + fabs(ac2)) + c2 * (fabs(ab1) + fabs(ab2));
    const double bound = 2.0e-15 * perm;
    if (det > bound) return 1;
    if (det < -bound) return -1;
  }
  const i128 ax = a.x - d.x, ay = a.y - d.y, bx = b.x - d.x, by = b.y - d.y, cx = c.x - d.x, cy = c.y - d.y;
  const i128 a2 = ax * ax + ay * ay, b2 = bx * bx + by * by, c2 = cx * cx + cy * cy;
  const i128 det = a2 * (bx * cy - by * cx) - b2 * (ax * cy - ay * cx) + c2 * (ax * by - ay * bx);
  return det > 0 ? 1 : (det < 0 ? -1 : 0);
}

// Disk through p, q, r: centre and a radius padded by the rounding bounds of the centre (pruning only).  A triangle
// too flat for double precision gets an infinite radius (= "the whole half-plane").
__device__ inline void circle_of(int2 p, int2 q, int2 r, double* cx, double* cy, double* rad) {
  const double ax = (double)(q.x - p.x), ay = (double)(q.y - p.y), bx = (double)(r.x - p.x), by = (double)(r.y - p.y);
  const double p1 = ax * by, p2 = ay * bx, D = p1 - p2;
  const double relD = 4.0e-16 * (fabs(p1) + fabs(p2)) / fabs(D);
  *cx = (double)p.x; *cy = (double)p.y; *rad = INFINITY;
  if (!(relD < 0.25)) return;
  const double a2 = ax * ax + ay * ay, b2 = bx * bx + by * by;
  const double nx = a2 * by - b2 * ay, ny = b2 * ax - a2 * bx;
  const double nxm = a2 * fabs(by) + b2 * fabs(ay), nym = b2 * fabs(ax) + a2 * fabs(bx);
  const double inv = 0.5 / D, ainv = fabs(inv);
  const double ux = nx * inv, uy = ny * inv;
  const double ex = (1.0e-15 * nxm + 1.4 * relD * fabs(nx)) * ainv, ey = (1.0e-15 * nym + 1.4 * relD * fabs(ny)) * ainv;
  const double rr = sqrt(ux * ux + uy * uy) * (1.0 + 1.0e-12) + ex + ey + 2.0;
  if (!(rr < 1.0e12)) return;  // (also NaN)
  *cx += ux; *cy += uy; *rad = rr;
}

__device__ inline int32_t clampi(double v, int32_t lo, int32_t hi) {
  return v <= (double)lo ? lo : (v >= (double)hi ? hi : (int32_t)v);
}

struct Star {
  const DtView& g;
  const DtBox bx;
  const int32_t ps, ip, pcx, pcy;
  const int2 p;
  __device__ Star(const DtView& g_, DtBox b_, int32_t ps_, int32_t ip_, int2 p_)
      : g(g_), bx(b_), ps(ps_), ip(ip_),
        pcx((int32_t)(((int64_t)(p_.x - b_.minx) * g_.G) / b_.spanx)), pcy((int32_t)(((int64_t)(p_.y - b_.miny) * g_.G) / b_.spany)),
        p(p_) {}

  // ---- the nearest live point (ties: smallest id): ring by ring around p's cell ----
  __device__ int32_t nearest() const {
    const int32_t G = g.G;
    int64_t bd = INT64_MAX;
    int32_t bs = -1, bid = INT32_MAX;
    const double cmin = fmin((double)bx.spanx / G, (double)bx.spany / G);
    for (int32_t k = 0; k < G; ++k) {
      // after rings < k every unscanned point is at least (k - 1) cells away from p
      if (bs >= 0) {
        const double reach = (double)(k - 1) * cmin - 2.0;
        if (reach > 0.0 && (double)bd * (1.0 + 1.0e-9) <= reach * reach) break;
      }
      const int32_t y0 = pcy - k, y1 = pcy + k, x0 = pcx - k, x1 = pcx + k;
      if (y0 < 0 && y1 >= G && x0 < 0 && x1 >= G) break;
      for (int32_t cy = max(y0, 0); cy <= min(y1, G - 1); ++cy) {
        const bool edge_row = (cy == y0 || cy == y1);
        const int32_t step = edge_row ? 1 : max(2 * k, 1);
        for (int32_t cx = x0; cx <= x1; cx += step) {
          if (cx < 0 || cx >= G) continue;
          const int32_t c = cy * G + cx;
          for (int32_t sl = g.start[c], se = g.start[c + 1]; sl < se; ++sl) {
            const int32_t id = g.sid[sl];
            if (id < 0 || sl == ps) continue;
            const int2 r = g.sxy[sl];
            const int64_t dx = r.x - p.x, dy = r.y - p.y, d2 = dx * dx + dy * dy;
            if (d2 < bd || (d2 == bd && id < bid)) { bd = d2; bs = sl; bid = id; }
          }
        }
      }
    }
    return bs;
  }

  // ---- the third vertex of the triangle on side s of the Delaunay edge p -> q (s = +1: left), or -1: hull edge ----
  __device__ int32_t next(int32_t qs, int2 q, int32_t iq, int s) const {
    const int32_t G = g.G;
    bool have = false;
    int2 b = p, f = p, l = p;
    int32_t fs = -1, ls = -1, ms = -1, mid = INT32_MAX;
    double ccx = 0.0, ccy = 0.0, rad = INFINITY;
    auto consider = [&](int32_t sl) {
      const int32_t id = g.sid[sl];
      if (id < 0 || sl == ps || sl == qs) return;
      const int2 r = g.sxy[sl];
      const int64_t o = orient64(p, q, r);
      if (s > 0 ? o <= 0 : o >= 0) return;
      const int t = have ? s * incircle_sign(p, q, b, r) : 1;
      if (t > 0) {
        have = true; b = r; f = r; l = r; fs = ls = ms = sl; mid = id;
        circle_of(p, q, r, &ccx, &ccy, &rad);
      } else if (t == 0) {  // on the current circle: angular order as seen from p, turning towards side s
        if (s * sgn64(orient64(p, r, f)) > 0) { f = r; fs = sl; }
        if (s * sgn64(orient64(p, l, r)) > 0) { l = r; ls = sl; }
        if (id < mid) { mid = id; ms = sl; }
      }
    };
    auto scan_cell = [&](int32_t c) {
      for (int32_t sl = g.start[c], se = g.start[c + 1]; sl < se; ++sl) consider(sl);
    };
    // the cells around p first: they hold the answer for an interior point and bound the cap for the rows below
    for (int32_t cy = max(pcy - 1, 0); cy <= min(pcy + 1, G - 1); ++cy)
      for (int32_t cx = max(pcx - 1, 0); cx <= min(pcx + 1, G - 1); ++cx) scan_cell(cy * G + cx);
    // every grid row the current cap can reach, outwards from p's row; false: the row is past the disk
    const double A = (double)s * (double)(q.x - p.x), B = (double)s * (double)(q.y - p.y);  // side(x, y) = A (y - py) - B (x - px) > 0
    auto do_row = [&](int32_t j) -> bool {
      const double ylo = (double)bx.miny + (double)(((int64_t)j * bx.spany) / G);
      const double yhi = (double)bx.miny + (double)(((int64_t)(j + 1) * bx.spany) / G) + 1.0;
      double xa = -INFINITY, xb = INFINITY;
      if (rad < INFINITY) {
        const double d = ylo > ccy ? ylo - ccy : (yhi < ccy ? ccy - yhi : 0.0);
        if (d > rad) return false;
        const double w = sqrt(rad * rad - d * d) * (1.0 + 1.0e-12) + 1.0;
        xa = ccx - w; xb = ccx + w;
      }
      const int2 rx = g.rowx[j];
      if (rx.x > rx.y) return true;
      xa = fmax(xa, (double)rx.x); xb = fmin(xb, (double)rx.y);
      const double h = A * ((A > 0.0 ? yhi : ylo) - (double)p.y);  // max of A (y - py) over the row
      if (B > 0.0) {
        const double t = h / B;
        xb = fmin(xb, (double)p.x + t + 1.0 + 1.0e-12 * fabs(t));
      } else if (B < 0.0) {
        const double t = h / B;
        xa = fmax(xa, (double)p.x + t - 1.0 - 1.0e-12 * fabs(t));
      } else if (!(h > 0.0)) {
        return true;
      }
      if (!(xa <= xb)) return true;
      const double sc = (double)G / (double)bx.spanx;
      const int32_t ca = clampi(floor((xa - (double)bx.minx) * sc - 1.0e-6), 0, G - 1);
      const int32_t cb = clampi(floor((xb - (double)bx.minx) * sc + 1.0e-6), 0, G - 1);
      const bool near_row = j >= pcy - 1 && j <= pcy + 1;
      for (int32_t cx = ca; cx <= cb; ++cx) {
        if (near_row && cx >= pcx - 1 && cx <= pcx + 1) continue;  // scanned above
        scan_cell(j * G + cx);
      }
      return true;
    };
    for (int32_t j = pcy; j < G; ++j)
      if (!do_row(j)) break;
    for (int32_t j = pcy - 1; j >= 0; --j)
      if (!do_row(j)) break;
    if (!have) return -1;
    // the points on the empty circle: a fan from the smallest id of the polygon p, q, first .. last
    if (ip < iq && ip < mid) return fs;
    if (iq < mid) return ls;
    return ms;
  }
};

// ---------------------------------------------------------------- kernels
__global__ void k_dt_init(int32_t* cnt, int32_t n, int32_t* flags) {
  const int32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) cnt[i] = 0;
  if (i == 0) {
    flags[0] = flags[1] = flags[2] = flags[3] = 0;
    flags[4] = flags[5] = INT32_MAX;
    flags[6] = flags[7] = INT32_MIN;
  }
}

__global__ void k_dt_snap(const float2* __restrict__ pos, int32_t V, int2* __restrict__ ixy, int32_t* flags) {
  const int32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  int32_t x0 = INT32_MAX, y0 = INT32_MAX, x1 = INT32_MIN, y1 = INT32_MIN;
  if (i < V) {
    const float2 v = pos[i];
    const double x = (double)v.x * 65536.0, y = (double)v.y * 65536.0;
    if (!(fabs(x) < 536870912.0) || !(fabs(y) < 536870912.0)) {  // 2^29, NaN
      atomicOr(&flags[0], kErrRange);
      ixy[i] = make_int2(0, 0);
      x0 = x1 = y0 = y1 = 0;
    } else {
      const int2 q = make_int2((int32_t)llround(x), (int32_t)llround(y));
      ixy[i] = q;
      x0 = x1 = q.x; y0 = y1 = q.y;
    }
  }
  for (int o = 32; o > 0; o >>= 1) {
    x0 = min(x0, __shfl_xor(x0, o)); y0 = min(y0, __shfl_xor(y0, o));
    x1 = max(x1, __shfl_xor(x1, o)); y1 = max(y1, __shfl_xor(y1, o));
  }
  if ((threadIdx.x & 63) == 0 && x0 <= x1) {
    atomicMin(&flags[4], x0); atomicMin(&flags[5], y0);
    atomicMax(&flags[6], x1); atomicMax(&flags[7], y1);
  }
}

__global__ void k_dt_count(const int2* __restrict__ ixy, int32_t V, int32_t G, const int32_t* __restrict__ flags,
                           int32_t* __restrict__ cell_of, int32_t* cnt) {
  const int32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= V) return;
  const DtBox b = dt_box(flags);
  const int2 q = ixy[i];
  const int32_t cx = (int32_t)(((int64_t)(q.x - b.minx) * G) / b.spanx), cy = (int32_t)(((int64_t)(q.y - b.miny) * G) / b.spany);
  const int32_t c = cy * G + cx;
  cell_of[i] = c;
  atomicAdd(&cnt[c], 1);
}

// exclusive scan of in[0..n) into out[0..n], out[n] = total, by ONE workgroup; optionally clears in[]
__global__ void __launch_bounds__(1024) k_dt_scan(int32_t* in, int32_t n, int32_t* out, int32_t clear_in, int32_t* total) {
  __shared__ int32_t part[1024];
  const int32_t t = threadIdx.x, chunk = (n + 1023) / 1024;
  const int32_t i0 = min(t * chunk, n), i1 = min(i0 + chunk, n);
  int32_t s = 0;
  for (int32_t i = i0; i < i1; ++i) s += in[i];
  part[t] = s;
  __syncthreads();
  for (int o = 1; o < 1024; o <<= 1) {
    const int32_t v = t >= o ? part[t - o] : 0;
    __syncthreads();
    part[t] += v;
    __syncthreads();
  }
  int32_t run = part[t] - s;
  for (int32_t i = i0; i < i1; ++i) {
    const int32_t v = in[i];
    out[i] = run;
    run += v;
    if (clear_in) in[i] = 0;
  }
  if (t == 1023) {
    out[n] = part[1023];
    if (total) *total = part[1023];
  }
}

__global__ void k_dt_scatter(const int2* __restrict__ ixy, const int32_t* __restrict__ cell_of, int32_t V,
                             const int32_t* __restrict__ start, int32_t* fill, int2* __restrict__ sxy, int32_t* __restrict__ sid) {
  const int32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= V) return;
  const int32_t c = cell_of[i];
  const int32_t sl = start[c] + atomicAdd(&fill[c], 1);
  sxy[sl] = ixy[i];
  sid[sl] = i;
}

// later copies of a point (same lattice coordinates, larger id) leave the triangulation: sid = ~id
__global__ void k_dt_dups(const int2* __restrict__ sxy, const int32_t* __restrict__ sid_in, int32_t* __restrict__ sid_out,
                          const int32_t* __restrict__ cell_of, const int32_t* __restrict__ start, int32_t V, int32_t* flags) {
  const int32_t sl = blockIdx.x * blockDim.x + threadIdx.x;
  if (sl >= V) return;
  const int32_t id = sid_in[sl];
  const int2 p = sxy[sl];
  const int32_t c = cell_of[id];
  bool dup = false;
  for (int32_t o = start[c], e = start[c + 1]; o < e; ++o) {
    const int2 r = sxy[o];
    if (r.x == p.x && r.y == p.y && sid_in[o] < id) { dup = true; break; }
  }
  sid_out[sl] = dup ? ~id : id;
  if (dup) atomicAdd(&flags[2], 1);
}

__global__ void k_dt_rows(const int2* __restrict__ sxy, const int32_t* __restrict__ sid, const int32_t* __restrict__ start,
                          int32_t G, int2* __restrict__ rowx) {
  const int32_t j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= G) return;
  int32_t lo = INT32_MAX, hi = INT32_MIN;
  for (int32_t sl = start[j * G], e = start[(j + 1) * G]; sl < e; ++sl) {
    if (sid[sl] < 0) continue;
    const int32_t x = sxy[sl].x;
    lo = min(lo, x); hi = max(hi, x);
  }
  rowx[j] = lo <= hi ? make_int2(lo, hi) : make_int2(1, 0);
}

template <bool WRITE>
__global__ void __launch_bounds__(64) k_dt_star(DtView g, int32_t* flags, int32_t* tcnt, const int32_t* __restrict__ toff,
                                                int32_t* __restrict__ tris, int32_t tri_cap) {
  const int32_t sl = blockIdx.x * blockDim.x + threadIdx.x;
  if (sl >= g.V) return;
  const int32_t ip = g.sid[sl];
  if (ip < 0) {
    if (!WRITE) tcnt[~ip] = 0;
    return;
  }
  const int2 p = g.sxy[sl];
  const Star st(g, dt_box(g.flags), sl, ip, p);
  int32_t n = 0;
  int32_t* out = WRITE ? tris + 3 * (size_t)toff[ip] : nullptr;
  const int32_t room = WRITE ? min(toff[ip + 1], tri_cap) - toff[ip] : 0;
  auto emit = [&](int32_t a, int32_t b) {  // triangle (p, a, b), counter-clockwise
    if (ip < a && ip < b) {
      if (WRITE) {
        if (n < room) { out[3 * n] = ip; out[3 * n + 1] = a; out[3 * n + 2] = b; }
        else atomicOr(&flags[0], kErrCap);
      }
      ++n;
    }
  };
  const int32_t q0 = st.nearest();
  if (q0 >= 0) {
    int32_t cur = q0, steps = 0;
    bool open = false;
    for (;;) {
      const int32_t r = st.next(cur, g.sxy[cur], g.sid[cur], +1);
      if (r < 0) { open = true; break; }
      emit(g.sid[cur], g.sid[r]);
      cur = r;
      if (r == q0) break;
      if (++steps > g.V) { atomicOr(&flags[0], kErrWrap); break; }
    }
    if (open) {
      cur = q0;
      for (;;) {
        const int32_t r = st.next(cur, g.sxy[cur], g.sid[cur], -1);
        if (r < 0) break;
        emit(g.sid[r], g.sid[cur]);
        cur = r;
        if (r == q0 || ++steps > g.V) { atomicOr(&flags[0], kErrWrap); break; }
      }
      if (!WRITE) atomicAdd(&flags[1], 1);
    }
  }
  if (!WRITE) tcnt[ip] = n;
  else if (n != toff[ip + 1] - toff[ip]) atomicOr(&flags[0], kErrCount);
}

inline size_t al256(size_t v) { return (v + 255) & ~(size_t)255; }

}  // namespace

void DelaunayScratch::release() {
  if (dev) (void)hipFree(dev);
  if (pin) (void)hipHostFree(pin);
  dev = nullptr; pin = nullptr; dev_cap = pin_cap = 0;
}

int delaunay_device(hipStream_t s, DelaunayScratch* sc, int32_t V, const float* pos, int32_t tri_cap, int32_t* tris_out,
                    int32_t* T_out) {
  if (!sc || V < 0 || tri_cap < 0 || !T_out || (V > 0 && !pos) || (tri_cap > 0 && !tris_out)) return FLAME_HIP_ERR_ARG;
  *T_out = 0;
  sc->last_hull = 0; sc->last_live = V;
  if (V < 3) return 0;
  const auto t0 = std::chrono::steady_clock::now();
  const int32_t G = std::max(1, std::min(256, (int32_t)std::ceil(std::sqrt(0.5 * (double)V))));
  const int32_t ncell = G * G;
  const int32_t tmax = 2 * V;  // T <= 2 V - 5
  // ---- arenas ----
  size_t off = 0;
  auto take = [&](size_t bytes) { const size_t o = off; off += al256(bytes); return o; };
  const size_t o_flags = take(sizeof(int32_t) * kFlagWords), o_pos = take(sizeof(float2) * (size_t)V), o_ixy = take(sizeof(int2) * (size_t)V);
  const size_t o_cell = take(sizeof(int32_t) * (size_t)V), o_cnt = take(sizeof(int32_t) * ((size_t)ncell + 1));
  const size_t o_start = take(sizeof(int32_t) * ((size_t)ncell + 1)), o_sxy = take(sizeof(int2) * (size_t)V);
  const size_t o_sid0 = take(sizeof(int32_t) * (size_t)V), o_sid = take(sizeof(int32_t) * (size_t)V), o_rowx = take(sizeof(int2) * (size_t)G);
  const size_t o_tcnt = take(sizeof(int32_t) * ((size_t)V + 1)), o_toff = take(sizeof(int32_t) * ((size_t)V + 1));
  const size_t o_tris = take(sizeof(int32_t) * 3 * (size_t)tmax);
  if (off > sc->dev_cap) {
    DT_HIPCHK(hipStreamSynchronize(s));
    if (sc->dev) (void)hipFree(sc->dev);
    sc->dev = nullptr; sc->dev_cap = 0;
    const size_t want = off + off / 2;
    if (hipMalloc(reinterpret_cast<void**>(&sc->dev), want) != hipSuccess) { (void)hipGetLastError(); return FLAME_HIP_ERR_ALLOC; }
    sc->dev_cap = want;
  }
  const size_t p_pos = 0, p_flags = al256(sizeof(float2) * (size_t)V), p_tris = p_flags + al256(sizeof(int32_t) * kFlagWords);
  const size_t pin_total = p_tris + al256(sizeof(int32_t) * 3 * (size_t)tmax);
  if (pin_total > sc->pin_cap) {
    DT_HIPCHK(hipStreamSynchronize(s));
    if (sc->pin) (void)hipHostFree(sc->pin);
    sc->pin = nullptr; sc->pin_cap = 0;
    const size_t want = pin_total + pin_total / 2;
    if (hipHostMalloc(reinterpret_cast<void**>(&sc->pin), want, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); return FLAME_HIP_ERR_ALLOC; }
    sc->pin_cap = want;
  }
  char* d = sc->dev;
  int32_t* flags = reinterpret_cast<int32_t*>(d + o_flags);
  float2* dpos = reinterpret_cast<float2*>(d + o_pos);
  int2* ixy = reinterpret_cast<int2*>(d + o_ixy);
  int32_t* cell_of = reinterpret_cast<int32_t*>(d + o_cell);
  int32_t* cnt = reinterpret_cast<int32_t*>(d + o_cnt);
  int32_t* start = reinterpret_cast<int32_t*>(d + o_start);
  int2* sxy = reinterpret_cast<int2*>(d + o_sxy);
  int32_t* sid0 = reinterpret_cast<int32_t*>(d + o_sid0);
  int32_t* sid = reinterpret_cast<int32_t*>(d + o_sid);
  int2* rowx = reinterpret_cast<int2*>(d + o_rowx);
  int32_t* tcnt = reinterpret_cast<int32_t*>(d + o_tcnt);
  int32_t* toff = reinterpret_cast<int32_t*>(d + o_toff);
  int32_t* dtris = reinterpret_cast<int32_t*>(d + o_tris);
  int32_t* hflags = reinterpret_cast<int32_t*>(sc->pin + p_flags);
  int32_t* htris = reinterpret_cast<int32_t*>(sc->pin + p_tris);

  std::memcpy(sc->pin + p_pos, pos, sizeof(float2) * (size_t)V);
  DT_HIPCHK(hipMemcpyAsync(dpos, sc->pin + p_pos, sizeof(float2) * (size_t)V, hipMemcpyHostToDevice, s));
  const int B = 256;
  const unsigned gv = (unsigned)((V + B - 1) / B);
  hipLaunchKernelGGL(k_dt_init, dim3((unsigned)((ncell + 1 + B - 1) / B)), dim3(B), 0, s, cnt, ncell + 1, flags);
  hipLaunchKernelGGL(k_dt_snap, dim3(gv), dim3(B), 0, s, dpos, V, ixy, flags);
  hipLaunchKernelGGL(k_dt_count, dim3(gv), dim3(B), 0, s, ixy, V, G, flags, cell_of, cnt);
  hipLaunchKernelGGL(k_dt_scan, dim3(1), dim3(1024), 0, s, cnt, ncell, start, 1, (int32_t*)nullptr);
  hipLaunchKernelGGL(k_dt_scatter, dim3(gv), dim3(B), 0, s, ixy, cell_of, V, start, cnt, sxy, sid0);
  hipLaunchKernelGGL(k_dt_dups, dim3(gv), dim3(B), 0, s, sxy, sid0, sid, cell_of, start, V, flags);
  hipLaunchKernelGGL(k_dt_rows, dim3((unsigned)((G + 63) / 64)), dim3(64), 0, s, sxy, sid, start, G, rowx);
  DtView view;
  view.sxy = sxy; view.sid = sid; view.start = start; view.rowx = rowx; view.flags = flags; view.G = G; view.V = V;
  const unsigned gs = (unsigned)((V + 63) / 64);
  hipLaunchKernelGGL(k_dt_star<false>, dim3(gs), dim3(64), 0, s, view, flags, tcnt, toff, dtris, tmax);
  hipLaunchKernelGGL(k_dt_scan, dim3(1), dim3(1024), 0, s, tcnt, V, toff, 0, flags + 3);
  hipLaunchKernelGGL(k_dt_star<true>, dim3(gs), dim3(64), 0, s, view, flags, tcnt, toff, dtris, tmax);
  DT_HIPCHK(hipGetLastError());
  // the count first (it bounds the copy of the list), then the list itself
  DT_HIPCHK(hipMemcpyAsync(hflags, flags, sizeof(int32_t) * kFlagWords, hipMemcpyDeviceToHost, s));
  DT_HIPCHK(hipStreamSynchronize(s));
  const int32_t err = hflags[0], hull = hflags[1], live = V - hflags[2], T = hflags[3];
  sc->last_hull = hull; sc->last_live = live;
  if (err & kErrRange) return FLAME_HIP_ERR_ARG;  // a coordinate outside |u|, |v| < 2^13 pixels, or not finite
  if (err || T < 0 || T > tmax || (T > 0 && T != 2 * live - 2 - hull)) return FLAME_HIP_ERR_STATE;
  if (T > tri_cap) return FLAME_HIP_ERR_ARG;
  if (T > 0) {
    DT_HIPCHK(hipMemcpyAsync(htris, dtris, sizeof(int32_t) * 3 * (size_t)T, hipMemcpyDeviceToHost, s));
    DT_HIPCHK(hipStreamSynchronize(s));
    std::memcpy(tris_out, htris, sizeof(int32_t) * 3 * (size_t)T);
  }
  *T_out = T;
  sc->last_ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count();
  return 0;
}

}  // namespace flamehip
