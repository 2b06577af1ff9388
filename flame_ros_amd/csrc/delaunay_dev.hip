// flame_ros_amd/csrc/delaunay_dev.hip -- Delaunay triangulation of a frame's features on the GPU (SURVEY.md 8 row
// f3's first leg; the reference budgets it as `triangulate` beside `sync_graph`, msg/FlameStats.msg:43-44; upstream
// calls Shewchuk's Triangle on the host).  include/flame/utils/delaunay.h is the host triangulator of the same contract
// (exact predicates on a 2^-16 pixel lattice, |u|, |v| < 2^13); this file is the device one.
//
// One WAVEFRONT per point builds that point's STAR -- its Delaunay neighbours in angular order -- by gift wrapping, from
// exact predicates only:
//   * the nearest neighbour q0 of p is a Delaunay neighbour of p in every Delaunay triangulation;
//   * given a Delaunay edge p -> q, the third vertex of the triangle on its left is the point r strictly left of p -> q
//     whose circle (p, q, r) holds no other point on that side ("r' beats r when r' is strictly inside (p, q, r)": a
//     total preorder -- the angle under which p q is seen);
//   * when nothing is strictly left of p -> q the edge is on the convex hull: the star is open, and is completed by
//     wrapping the other way round from q0.
// The 64 lanes test 64 candidates at a time (one coalesced 1 KB load of 16-byte records, the predicates in parallel, a
// ballot; the star's own state -- current best, circle, ties -- is wave-uniform, so the control flow never diverges).
// Cocircular points (pixel lattices are full of them) are resolved by ONE rule every star applies alike: the polygon of
// the points on an empty circle is triangulated as a fan from its smallest vertex id.  A star sees the polygon either
// whole (from one of its boundary edges) or as the part left of one of the fan's diagonals, in which the smallest id is
// still a vertex: the rule restricted to the part is the same fan, so the stars agree and every triangle appears in the
// stars of its three vertices.  It is written once, by the star of its smallest vertex.
//
// Candidates come from a uniform grid (about two points per cell, cells row-major: a run of cells in a grid row is ONE
// run of records).  The 3 x 3 cells around p are loaded once per star, packed into the 64 lanes: they answer nearly
// every step of an interior star.  Floating point is used for PRUNING only, always conservatively: a grid row, or the
// part of it outside an x interval, is skipped when it cannot meet the current cap (the part of the current circle's
// disk left of p -> q; every later cap is inside it), with the disk's centre and radius padded by their rounding
// bounds; 64 rows are judged at a time, one per lane.  Hull edges query a whole half-plane: per grid row the x extent of
// its points makes that one test per row.  Every accept / reject of a candidate is exact: orientation in 64-bit
// integers (differences < 2^30), in-circle by a double-precision filter and 128-bit integers behind it (sum < 2^124),
// exactly as the host triangulator.
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
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

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
constexpr int kStash = 20;  // triangles a star keeps beside its count in the first pass (more: the star is rebuilt in the second)

struct DtRow {
  double ylo, yhi;       // every point of the row has ylo <= y < yhi (lattice)
  int32_t xlo, xhi;      // min / max x of its live points (xlo > xhi: none)
};

struct DtView {
  const int4* rec;       // {x, y (lattice), id (~id: a later copy of another point), cell}, cell by cell
  const int32_t* start;  // G*G + 1 cell offsets (row-major: a grid row's cells are contiguous)
  const DtRow* row;      // per grid row: its y extent and the x extent of its live points
  const int32_t* flags;
  int32_t G, V;
  const int32_t* order;  // wavefront -> slot: the stars of the grid's boundary cells first (NULL: slot order)
  int32_t* dbg;          // dev aid (FLAME_HIP_DT_STATS): per point {own triangles, chunks, row batches, row scans}
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

// the exact evaluation behind the filter: differences < 2^30: squared norms and 2 x 2 minors < 2^61 fit 64 bits; three
// 64 x 64 -> 128-bit products, sum < 2^124.  NOT inlined (r05 experiment): it is rare, and its temporaries inflate the
// star kernel's register allocation
__device__ __attribute__((noinline)) int incircle_exact(int2 a, int2 b, int2 c, int2 d) {
  const int64_t ax = a.x - d.x, ay = a.y - d.y, bx = b.x - d.x, by = b.y - d.y, cx = c.x - d.x, cy = c.y - d.y;
  const int64_t a2 = ax * ax + ay * ay, b2 = bx * bx + by * by, c2 = cx * cx + cy * cy;
  const int64_t mbc = bx * cy - by * cx, mac = ax * cy - ay * cx, mab = ax * by - ay * bx;
  const i128 det = (i128)a2 * mbc - (i128)b2 * mac + (i128)c2 * mab;
  return det > 0 ? 1 : (det < 0 ? -1 : 0);
}

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
  return incircle_exact(a, b, c, d);
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

// A star is built by kSW lanes: 64 = one star per wavefront (the default); 32 = two stars per wavefront, each on its own
// half -- a step of the wrapping has ~18 candidates, so 64 lanes are mostly idle -- measured (-DFLAME_DT_STAR_LANES=32):
// the star kernel 193 -> 158 us at 10 k points and 690 -> 600 at 50 k, but 58 -> 72 at 1.2 k and 2 170 -> 2 680 at 200 k
// (the two stars of a wavefront run each other's loop iterations; nothing of the star's state stays scalar): not kept.
// "Uniform" below means: the same on every lane of the star.
#ifndef FLAME_DT_STAR_LANES
#define FLAME_DT_STAR_LANES 64
#endif
constexpr int kSW = FLAME_DT_STAR_LANES;
constexpr unsigned long long kStarMask = kSW == 64 ? ~0ull : ((1ull << (kSW & 63)) - 1ull);
static_assert(kSW == 32 || kSW == 64, "a star is a wavefront or half of one");

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

// snap to the lattice; bounding box: grid-stride, one set of atomics per workgroup
__global__ void __launch_bounds__(256) k_dt_snap(const float2* __restrict__ pos, int32_t V, int2* __restrict__ ixy, int32_t* flags) {
  __shared__ int32_t red[4][4];
  int32_t x0 = INT32_MAX, y0 = INT32_MAX, x1 = INT32_MIN, y1 = INT32_MIN;
  for (int32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < V; i += gridDim.x * blockDim.x) {
    const float2 v = pos[i];
    const double x = (double)v.x * 65536.0, y = (double)v.y * 65536.0;
    int2 q = make_int2(0, 0);
    if (!(fabs(x) < 536870912.0) || !(fabs(y) < 536870912.0)) atomicOr(&flags[0], kErrRange);  // 2^29, NaN
    else q = make_int2((int32_t)llround(x), (int32_t)llround(y));
    ixy[i] = q;
    x0 = min(x0, q.x); y0 = min(y0, q.y); x1 = max(x1, q.x); y1 = max(y1, q.y);
  }
  for (int o = 32; o > 0; o >>= 1) {
    x0 = min(x0, __shfl_xor(x0, o)); y0 = min(y0, __shfl_xor(y0, o));
    x1 = max(x1, __shfl_xor(x1, o)); y1 = max(y1, __shfl_xor(y1, o));
  }
  const int w = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) { red[w][0] = x0; red[w][1] = y0; red[w][2] = x1; red[w][3] = y1; }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int k = 1; k < 4; ++k) { x0 = min(x0, red[k][0]); y0 = min(y0, red[k][1]); x1 = max(x1, red[k][2]); y1 = max(y1, red[k][3]); }
    if (x0 <= x1) {
      atomicMin(&flags[4], x0); atomicMin(&flags[5], y0);
      atomicMax(&flags[6], x1); atomicMax(&flags[7], y1);
    }
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

// ---- exclusive scan of n ints in three launches: sums of 1024-element blocks, scan of the sums, blocks ----
__device__ inline int32_t block_excl_scan_1024(int32_t v, int32_t* total) {  // 1024 threads; returns the exclusive prefix of v
  __shared__ int32_t wsum[16];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  int32_t inc = v;
  for (int o = 1; o < 64; o <<= 1) {
    const int32_t u = __shfl_up(inc, o);
    if (lane >= o) inc += u;
  }
  if (lane == 63) wsum[w] = inc;
  __syncthreads();
  if (w == 0) {
    int32_t s = lane < 16 ? wsum[lane] : 0;
    for (int o = 1; o < 16; o <<= 1) {
      const int32_t u = __shfl_up(s, o);
      if (lane >= o) s += u;
    }
    if (lane < 16) wsum[lane] = s;
  }
  __syncthreads();
  const int32_t before = w ? wsum[w - 1] : 0;
  *total = wsum[15];
  return before + inc - v;
}
__global__ void __launch_bounds__(1024) k_dt_scan_sums(const int32_t* __restrict__ in, int32_t n, int32_t* __restrict__ sums) {
  const int32_t i = blockIdx.x * 1024 + threadIdx.x;
  int32_t tot;
  (void)block_excl_scan_1024(i < n ? in[i] : 0, &tot);
  if (threadIdx.x == 0) sums[blockIdx.x] = tot;
}
__global__ void __launch_bounds__(1024) k_dt_scan_top(int32_t* sums, int32_t nb, int32_t* out_n, int32_t* total) {
  // nb <= 1024 blocks (n <= 2^20); sums become the blocks' offsets
  int32_t tot;
  const int32_t v = (int32_t)threadIdx.x < nb ? sums[threadIdx.x] : 0;
  const int32_t ex = block_excl_scan_1024(v, &tot);
  if ((int32_t)threadIdx.x < nb) sums[threadIdx.x] = ex;
  if (threadIdx.x == 0) {
    *out_n = tot;
    if (total) *total = tot;
  }
}
__global__ void __launch_bounds__(1024) k_dt_scan_blocks(int32_t* in, int32_t n, const int32_t* __restrict__ sums,
                                                         int32_t* __restrict__ out, int32_t clear_in) {
  const int32_t i = blockIdx.x * 1024 + threadIdx.x;
  int32_t tot;
  const int32_t ex = block_excl_scan_1024(i < n ? in[i] : 0, &tot);
  if (i < n) {
    out[i] = sums[blockIdx.x] + ex;
    if (clear_in) in[i] = 0;
  }
}

__global__ void k_dt_scatter(const int2* __restrict__ ixy, const int32_t* __restrict__ cell_of, int32_t V,
                             const int32_t* __restrict__ start, int32_t* fill, int4* __restrict__ rec) {
  const int32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= V) return;
  const int32_t c = cell_of[i];
  const int32_t sl = start[c] + atomicAdd(&fill[c], 1);
  const int2 q = ixy[i];
  rec[sl] = make_int4(q.x, q.y, i, c);
}

// later copies of a point (same lattice coordinates, larger id) leave the triangulation: id -> ~id
__global__ void k_dt_dups(const int4* __restrict__ rec_in, int4* __restrict__ rec_out, const int32_t* __restrict__ start,
                          int32_t V, int32_t* flags) {
  const int32_t sl = blockIdx.x * blockDim.x + threadIdx.x;
  if (sl >= V) return;
  int4 p = rec_in[sl];
  const int32_t c = p.w;
  bool dup = false;
  for (int32_t o = start[c], e = start[c + 1]; o < e; ++o) {
    const int4 r = rec_in[o];
    if (r.x == p.x && r.y == p.y && r.z < p.z) { dup = true; break; }
  }
  if (dup) { p.z = ~p.z; atomicAdd(&flags[2], 1); }
  rec_out[sl] = p;
}

// per grid row its y extent and the x extent of its live points: one wavefront per row
__global__ void __launch_bounds__(64) k_dt_rows(const int4* __restrict__ rec, const int32_t* __restrict__ start, int32_t G,
                                                const int32_t* __restrict__ flags, DtRow* __restrict__ row) {
  const int32_t j = blockIdx.x;
  int32_t lo = INT32_MAX, hi = INT32_MIN;
  for (int32_t sl = start[j * G] + threadIdx.x, e = start[(j + 1) * G]; sl < e; sl += 64) {
    const int4 r = rec[sl];
    if (r.z >= 0) { lo = min(lo, r.x); hi = max(hi, r.x); }
  }
  for (int o = 32; o > 0; o >>= 1) { lo = min(lo, __shfl_xor(lo, o)); hi = max(hi, __shfl_xor(hi, o)); }
  if (threadIdx.x == 0) {
    const DtBox b = dt_box(flags);
    DtRow rw;
    // row j holds the points with j spany / G <= y - miny < (j + 1) spany / G
    rw.ylo = (double)b.miny + (double)(((int64_t)j * b.spany) / G);
    rw.yhi = (double)b.miny + (double)(((int64_t)(j + 1) * b.spany) / G) + 1.0;
    rw.xlo = lo <= hi ? lo : 1; rw.xhi = lo <= hi ? hi : 0;
    row[j] = rw;
  }
}

// ---- small frames (V <= kSmallV: the reference's own 640 x 480 frames have ~1.2 k features): everything in front of the
// star kernel in ONE launch of one workgroup -- snap + bounding box, cell counts, their scan, the scatter, the copies and the
// row extents -- with counts, offsets and records in LDS; the positions are read straight from the caller's page-locked
// copy (no DMA in front).  Same arrays as the kernels above produce (the order inside a cell may differ: nothing
// depends on it).
constexpr int kSmallV = 2048, kSmallG = 32;
__global__ void __launch_bounds__(1024) k_dt_prep_small(const float2* __restrict__ pos, int32_t V, int32_t G, int32_t* flags,
                                                        int32_t* __restrict__ start, int4* __restrict__ rec, DtRow* __restrict__ row) {
  __shared__ int32_t s_cnt[kSmallG * kSmallG], s_start[kSmallG * kSmallG + 1];
  __shared__ int4 s_rec[kSmallV];
  __shared__ int32_t s_red[16][4], s_box[4], s_dups;
  const int32_t t = threadIdx.x, lane = t & 63, w = t >> 6, ncell = G * G;
  if (t < ncell) s_cnt[t] = 0;
  if (t == 0) s_dups = 0;
  int2 q[2];
  bool bad = false;
  int32_t x0 = INT32_MAX, y0 = INT32_MAX, x1 = INT32_MIN, y1 = INT32_MIN;
  for (int k = 0; k < 2; ++k) {
    const int32_t i = t + 1024 * k;
    q[k] = make_int2(0, 0);
    if (i < V) {
      const float2 v = pos[i];
      const double x = (double)v.x * 65536.0, y = (double)v.y * 65536.0;
      if (!(fabs(x) < 536870912.0) || !(fabs(y) < 536870912.0)) bad = true;
      else q[k] = make_int2((int32_t)llround(x), (int32_t)llround(y));
      x0 = min(x0, q[k].x); y0 = min(y0, q[k].y); x1 = max(x1, q[k].x); y1 = max(y1, q[k].y);
    }
  }
  for (int o = 32; o > 0; o >>= 1) {
    x0 = min(x0, __shfl_xor(x0, o)); y0 = min(y0, __shfl_xor(y0, o));
    x1 = max(x1, __shfl_xor(x1, o)); y1 = max(y1, __shfl_xor(y1, o));
  }
  if (lane == 0) { s_red[w][0] = x0; s_red[w][1] = y0; s_red[w][2] = x1; s_red[w][3] = y1; }
  const unsigned long long anybad = __ballot(bad);
  __syncthreads();
  if (t == 0) {
    for (int k = 1; k < 16; ++k) { x0 = min(x0, s_red[k][0]); y0 = min(y0, s_red[k][1]); x1 = max(x1, s_red[k][2]); y1 = max(y1, s_red[k][3]); }
    s_box[0] = x0; s_box[1] = y0; s_box[2] = x1; s_box[3] = y1;
    flags[0] = 0; flags[1] = 0; flags[2] = 0; flags[3] = 0;
    flags[4] = x0; flags[5] = y0; flags[6] = x1; flags[7] = y1;
  }
  __syncthreads();
  if (anybad && lane == 0) atomicOr(&flags[0], kErrRange);  // (behind thread 0's reset)
  DtBox b;
  b.minx = s_box[0]; b.miny = s_box[1];
  b.spanx = (int64_t)s_box[2] - s_box[0] + 1; b.spany = (int64_t)s_box[3] - s_box[1] + 1;
  int32_t c[2] = {0, 0};
  for (int k = 0; k < 2; ++k)
    if (t + 1024 * k < V) {
      const int32_t cx = (int32_t)(((int64_t)(q[k].x - b.minx) * G) / b.spanx), cy = (int32_t)(((int64_t)(q[k].y - b.miny) * G) / b.spany);
      c[k] = cy * G + cx;
      atomicAdd(&s_cnt[c[k]], 1);
    }
  __syncthreads();
  {
    int32_t tot;
    const int32_t ex = block_excl_scan_1024(t < ncell ? s_cnt[t] : 0, &tot);
    if (t < ncell) { s_start[t] = ex; start[t] = ex; s_cnt[t] = 0; }
    if (t == 0) { s_start[ncell] = V; start[ncell] = V; }
  }
  __syncthreads();
  for (int k = 0; k < 2; ++k)
    if (t + 1024 * k < V) {
      const int32_t sl = s_start[c[k]] + atomicAdd(&s_cnt[c[k]], 1);
      s_rec[sl] = make_int4(q[k].x, q[k].y, t + 1024 * k, c[k]);
    }
  __syncthreads();
  bool dup[2] = {false, false};
  for (int k = 0; k < 2; ++k) {
    const int32_t sl = t + 1024 * k;
    if (sl < V) {
      const int4 p = s_rec[sl];
      for (int32_t o = s_start[p.w], e = s_start[p.w + 1]; o < e; ++o) {
        const int4 r = s_rec[o];
        if (r.x == p.x && r.y == p.y && r.z < p.z) { dup[k] = true; break; }
      }
    }
  }
  __syncthreads();
  for (int k = 0; k < 2; ++k) {
    const int32_t sl = t + 1024 * k;
    if (sl < V) {
      int4 p = s_rec[sl];
      if (dup[k]) { p.z = ~p.z; s_rec[sl] = p; atomicAdd(&s_dups, 1); }
      rec[sl] = p;
    }
  }
  __syncthreads();
  if (t == 0) flags[2] = s_dups;
  for (int32_t j = w; j < G; j += 16) {
    int32_t lo = INT32_MAX, hi = INT32_MIN;
    for (int32_t sl = s_start[j * G] + lane, e = s_start[(j + 1) * G]; sl < e; sl += 64) {
      const int4 r = s_rec[sl];
      if (r.z >= 0) { lo = min(lo, r.x); hi = max(hi, r.x); }
    }
    for (int o = 32; o > 0; o >>= 1) { lo = min(lo, __shfl_xor(lo, o)); hi = max(hi, __shfl_xor(hi, o)); }
    if (lane == 0) {
      DtRow rw;
      rw.ylo = (double)b.miny + (double)(((int64_t)j * b.spany) / G);
      rw.yhi = (double)b.miny + (double)(((int64_t)(j + 1) * b.spany) / G) + 1.0;
      rw.xlo = lo <= hi ? lo : 1; rw.xhi = lo <= hi ? hi : 0;
      row[j] = rw;
    }
  }
}

// exclusive scan of n <= 1024 * kScanOne ints by ONE workgroup (out[n] = total, also *total): one launch where the general
// scan takes three
constexpr int kScanOne = 16;
__global__ void __launch_bounds__(1024) k_dt_scan_one(int32_t* in, int32_t n, int32_t* __restrict__ out, int32_t clear_in, int32_t* total) {
  const int32_t t = threadIdx.x, per = (n + 1023) / 1024, i0 = min(t * per, n), i1 = min(i0 + per, n);
  int32_t v[kScanOne], sum = 0;
  for (int k = 0; k < kScanOne; ++k) {
    v[k] = i0 + k < i1 ? in[i0 + k] : 0;
    sum += v[k];
  }
  int32_t tot;
  int32_t run = block_excl_scan_1024(sum, &tot);
  for (int k = 0; k < kScanOne; ++k)
    if (i0 + k < i1) {
      out[i0 + k] = run;
      run += v[k];
      if (clear_in) in[i0 + k] = 0;
    }
  if (t == 0) { out[n] = tot; if (total) *total = tot; }
}

// ---- launch order of the stars.  A star of a boundary cell of the grid is a hull star (or next to one): long thin triangles,
// big circles, three to five times the life of an interior star.  In slot order the last grid row -- all of them -- starts
// last and IS the tail of the launch (10 k points: 96 us of work on 2 048 wavefront slots, 170 us of launch).  So: the
// boundary cells first (row 0, row G - 1, then columns 0 and G - 1 of the rows between), the interior in slot order
// behind them (neighbouring wavefronts still share their cells).
__global__ void __launch_bounds__(256) k_dt_order_rows(const int32_t* __restrict__ start, int32_t G, int32_t* __restrict__ rowb,
                                                       int32_t* __restrict__ rowi) {
  // rows j = 1 .. G - 2 (G >= 3): exclusive prefix sums of their boundary slots (columns 0, G - 1) and interior slots
  __shared__ int32_t sb[256], si[256];
  const int32_t t = threadIdx.x, j = t + 1;
  int32_t cb = 0, ci = 0;
  if (j <= G - 2) {
    cb = (start[j * G + 1] - start[j * G]) + (start[j * G + G] - start[j * G + G - 1]);
    ci = start[j * G + G - 1] - start[j * G + 1];
  }
  sb[t] = cb; si[t] = ci;
  __syncthreads();
  for (int o = 1; o < 256; o <<= 1) {
    const int32_t vb = t >= o ? sb[t - o] : 0, vi = t >= o ? si[t - o] : 0;
    __syncthreads();
    sb[t] += vb; si[t] += vi;
    __syncthreads();
  }
  if (t <= G - 2) { rowb[t] = sb[t] - cb; rowi[t] = si[t] - ci; }  // (entry G - 2: the totals)
}
__global__ void k_dt_order_fill(const int32_t* __restrict__ start, const int32_t* __restrict__ rowb, const int32_t* __restrict__ rowi,
                                int32_t G, int32_t V, int32_t* __restrict__ order) {
  const int32_t w = blockIdx.x * blockDim.x + threadIdx.x;
  if (w >= V) return;
  const int32_t nr = G - 2, nb0 = start[G], s1 = start[(G - 1) * G], nb1 = V - s1, nb = nb0 + nb1 + rowb[nr];
  auto row_of = [&](const int32_t* pre, int32_t k) {  // the last of the rows r with pre[r] <= k
    int32_t lo = 0, hi = nr - 1;
    while (lo < hi) {
      const int32_t mid = (lo + hi + 1) >> 1;
      if (pre[mid] <= k) lo = mid; else hi = mid - 1;
    }
    return lo;
  };
  int32_t slot;
  if (w < nb0) {
    slot = w;
  } else if (w < nb0 + nb1) {
    slot = s1 + (w - nb0);
  } else if (w < nb) {
    const int32_t k = w - nb0 - nb1, r = row_of(rowb, k), j = r + 1, off = k - rowb[r], c0 = start[j * G + 1] - start[j * G];
    slot = off < c0 ? start[j * G] + off : start[j * G + G - 1] + (off - c0);
  } else {
    const int32_t m = w - nb, r = row_of(rowi, m), j = r + 1;
    slot = start[j * G + 1] + (m - rowi[r]);
  }
  order[w] = slot;
}

// First pass (WRITE = false): every star counts its triangles and keeps the first kStash of them in `stash`.
// Second pass: the stashed triangles are copied to their place in the list; a star with more is built again, writing.
#ifdef FLAME_DT_WAVES  /* dev A/B: cap the registers for this many waves per SIMD */
#define FLAME_DT_WAVES_ATTR __attribute__((amdgpu_waves_per_eu(FLAME_DT_WAVES)))
#else
#define FLAME_DT_WAVES_ATTR
#endif
// One wavefront per workgroup: a wavefront's slot is free for the next star the moment ITS star is done (with four stars per
// workgroup a new workgroup waits for a free slot on each of the CU's four SIMDs, i.e. for the slowest of four stars).
#ifndef FLAME_DT_STAR_WG
#define FLAME_DT_STAR_WG 64
#endif
constexpr int kStarWG = FLAME_DT_STAR_WG;
template <bool WRITE>
__device__ __forceinline__ void dt_star_body(const DtView& g, int32_t* flags, int32_t* tcnt, const int32_t* __restrict__ toff,
                                             int32_t* __restrict__ stash, int32_t* __restrict__ tris, int32_t tri_cap) {
  const int32_t lane = threadIdx.x & (kSW - 1);       // lane within the star
  const int32_t shift = (threadIdx.x & 63) & ~(kSW - 1);  // first lane of the star within its wavefront
  const int32_t w0 = blockIdx.x * (blockDim.x / kSW) + threadIdx.x / kSW;
  const int32_t sl = (g.order && w0 < g.V) ? g.order[w0] : w0;
  // ballot over the star's lanes; the value lane w of the star holds (w: the same on every lane of the star)
  auto sballot = [&](bool pr_) __attribute__((always_inline)) -> unsigned long long { return (__ballot(pr_) >> shift) & kStarMask; };
  auto bcast = [&](int32_t v, int w) __attribute__((always_inline)) -> int32_t {
    if constexpr (kSW == 64) return __builtin_amdgcn_readlane(v, __builtin_amdgcn_readfirstlane(w));  // (w is wave-uniform)
    else return __shfl(v, shift + w);
  };
  auto bcastd = [&](double v, int w) __attribute__((always_inline)) -> double {
    return __hiloint2double(bcast(__double2hiint(v), w), bcast(__double2loint(v), w));
  };
  if (sl >= g.V) return;
  const int4 pr = g.rec[sl];
  const int32_t ip = pr.z;
  if (ip < 0) {
    if (!WRITE && lane == 0) tcnt[~ip] = 0;
    return;
  }
  int32_t* out = nullptr;
  int32_t room = 0;
  if (WRITE) {
    const int32_t o0 = toff[ip], n0 = toff[ip + 1] - o0;
    if (o0 + n0 > tri_cap) {
      if (lane == 0) atomicOr(&flags[0], kErrCap);
      return;
    }
    if (n0 <= kStash) {
      for (int32_t k = lane; k < 3 * n0; k += kSW) tris[3 * (size_t)o0 + k] = stash[(size_t)ip * (3 * kStash) + k];
      return;
    }
    out = tris + 3 * (size_t)o0;
    room = n0;
  } else {
    out = stash + (size_t)ip * (3 * kStash);
    room = kStash;
  }
  // ---- one wavefront = one star.  Its state is plain locals (an object holding them, or a reference to the kernel
  // argument, ends up in scratch memory: measured 250 scratch accesses per star, half of the kernel's time); everything
  // below is wave-uniform except `lane` and what is loaded per lane ----
  const int4* const g_rec = g.rec;
  const int32_t* const g_start = g.start;
  const DtRow* const g_row = g.row;
  const int32_t G = g.G, ps = sl;
  const DtBox bx = dt_box(g.flags);
  const int2 p = make_int2(pr.x, pr.y);
  const int32_t pcx = pr.w % G, pcy = pr.w / G;
  // the 3 x 3 cells around p, loaded ONCE per star (every step of the wrapping starts there, and an interior star
  // rarely looks further): the three runs of records (one per grid row) packed into the 64 lanes, if they fit
  int32_t sxa, sxb;            // the block's cell columns
  int32_t cslot;               // this lane's slot
  int4 crec;                   // ... and record ({.., id = -1}: none)
  bool cached;
  double ix0, ix1, iy0, iy1;   // a disk inside [ix0, ix1) x [iy0, iy1) holds points of the block only
  int32_t n_chunks = 0, n_rows = 0, n_scans = 0;
  {
    sxa = max(pcx - 1, 0); sxb = min(pcx + 1, G - 1);
    const int32_t sya = max(pcy - 1, 0), syb = min(pcy + 1, G - 1);
    int32_t cbase[3], cn[3], tot = 0;
    for (int k = 0; k < 3; ++k) {
      const int32_t cy = pcy - 1 + k;
      cbase[k] = 0; cn[k] = 0;
      if (cy >= 0 && cy < G) {
        cbase[k] = g_start[cy * G + sxa];
        cn[k] = g_start[cy * G + sxb + 1] - cbase[k];
      }
      tot += cn[k];
    }
    cached = tot <= kSW;
    crec = make_int4(0, 0, -1, 0);
    cslot = -1;
    if (cached && lane < tot) {
      cslot = lane < cn[0] ? cbase[0] + lane : (lane < cn[0] + cn[1] ? cbase[1] + lane - cn[0] : cbase[2] + lane - cn[0] - cn[1]);
      crec = g_rec[cslot];
    }
    // a point left of column sxa has x < minx + ceil(sxa spanx / G); one right of column sxb has x >= minx + (sxb + 1) spanx / G
    ix0 = sxa == 0 ? -INFINITY : (double)bx.minx + (double)(((int64_t)sxa * bx.spanx) / G) + 1.0;
    ix1 = sxb == G - 1 ? INFINITY : (double)bx.minx + (double)(((int64_t)(sxb + 1) * bx.spanx) / G);
    iy0 = sya == 0 ? -INFINITY : (double)bx.miny + (double)(((int64_t)sya * bx.spany) / G) + 1.0;
    iy1 = syb == G - 1 ? INFINITY : (double)bx.miny + (double)(((int64_t)(syb + 1) * bx.spany) / G);
  }

  // ---- the nearest live point (ties: smallest id): ring by ring around p's cell ----
  auto nearest = [&]() __attribute__((always_inline)) -> int32_t {
    int64_t bd = INT64_MAX;   // per lane, reduced at the end of every ring
    int32_t bs = -1, bid = INT32_MAX;
    const double cmin = fmin((double)bx.spanx / G, (double)bx.spany / G);
    auto scan = [&](int32_t s0, int32_t s1) __attribute__((always_inline)) {
      for (int32_t base = s0; base < s1; base += kSW) {
        const int32_t sl = base + lane;
        if (sl < s1) {
          const int4 r = g_rec[sl];
          if (r.z >= 0 && sl != ps) {
            const int64_t dx = r.x - p.x, dy = r.y - p.y, d2 = dx * dx + dy * dy;
            if (d2 < bd || (d2 == bd && r.z < bid)) { bd = d2; bs = sl; bid = r.z; }
          }
        }
      }
    };
    auto reduce = [&]() __attribute__((always_inline)) {  // the result so far, on every lane
      for (int o = kSW / 2; o > 0; o >>= 1) {
        const int64_t od = __shfl_xor(bd, o);
        const int32_t os = __shfl_xor(bs, o), oi = __shfl_xor(bid, o);
        if (od < bd || (od == bd && oi < bid)) { bd = od; bs = os; bid = oi; }
      }
    };
    int32_t k0 = 0;
    if (cached) {  // rings 0 and 1 are the cached block
      if (crec.z >= 0 && cslot != ps) {
        const int64_t dx = crec.x - p.x, dy = crec.y - p.y;
        bd = dx * dx + dy * dy; bs = cslot; bid = crec.z;
      }
      reduce();
      k0 = 2;
    }
    for (int32_t k = k0; k < G; ++k) {
      // after rings < k every unscanned point is at least (k - 1) cells away from p
      if (bs >= 0) {
        const double reach = (double)(k - 1) * cmin - 2.0;
        if (reach > 0.0 && (double)bd * (1.0 + 1.0e-9) <= reach * reach) break;
      }
      const int32_t y0 = pcy - k, y1 = pcy + k, x0 = pcx - k, x1 = pcx + k;
      if (y0 < 0 && y1 >= G && x0 < 0 && x1 >= G) break;
      const int32_t xa = max(x0, 0), xb = min(x1, G - 1);
      for (int32_t cy = max(y0, 0); cy <= min(y1, G - 1); ++cy) {
        if (cy == y0 || cy == y1) {
          scan(g_start[cy * G + xa], g_start[cy * G + xb + 1]);
        } else {
          if (x0 >= 0) scan(g_start[cy * G + x0], g_start[cy * G + x0 + 1]);
          if (x1 < G) scan(g_start[cy * G + x1], g_start[cy * G + x1 + 1]);
        }
      }
      reduce();
    }
    return bs;
  };

  // ---- the third vertex of the triangle on side s of the Delaunay edge p -> q (s = +1: left), or -1: hull edge ----
  auto next = [&](int32_t qs, int2 q, int32_t iq, int s, int4* nrec) __attribute__((always_inline)) -> int32_t {
    bool have = false;
    int2 b = p, f = p, l = p;
    int32_t bsl = -1, fs = -1, ls = -1, ms = -1, mid = INT32_MAX, fid = -1, lid = -1;
    int2 mxy = p;  // (first / last / smallest id on the circle: slot, coordinates and id, so the winner needs no load)
    double ccx = 0.0, ccy = 0.0, rad = INFINITY;
    auto chunk = [&](int32_t sl, int4 r) __attribute__((always_inline)) {  // kSW candidates: this lane's slot and record (r.z < 0: none)
      ++n_chunks;
      const int2 rp = make_int2(r.x, r.y);
      bool ok = r.z >= 0 && sl != ps && sl != qs;
      if (ok) {
        const int64_t o = orient64(p, q, rp);
        ok = s > 0 ? o > 0 : o < 0;
      }
      if (!sballot(ok)) return;
      int t;
      bool moved = false;
      for (;;) {  // candidates that beat the current best: one of them becomes the best, the others are asked again
        t = !ok ? -1 : (!have ? 1 : (sl == bsl ? 0 : s * incircle_sign(p, q, b, rp)));  // (the best against itself: on the circle)
        const unsigned long long m = sballot(t > 0);
        if (!m) break;
        const int w = __ffsll(m) - 1;
        b = make_int2(bcast(r.x, w), bcast(r.y, w));
        have = true; moved = true; f = b; l = b; mxy = b; bsl = fs = ls = ms = bcast(sl, w); mid = fid = lid = bcast(r.z, w);
      }
      if (moved) circle_of(p, q, b, &ccx, &ccy, &rad);
      // on the current circle: angular order as seen from p, turning towards side s (the best itself is among them already)
      unsigned long long m0 = sballot(t == 0);
      m0 &= ~sballot(sl == bsl);
      while (m0) {
        const int w = __ffsll(m0) - 1;
        m0 &= m0 - 1;
        const int2 rt = make_int2(bcast(r.x, w), bcast(r.y, w));
        const int32_t it = bcast(r.z, w), st = bcast(sl, w);
        if (s * sgn64(orient64(p, rt, f)) > 0) { f = rt; fs = st; fid = it; }
        if (s * sgn64(orient64(p, l, rt)) > 0) { l = rt; ls = st; lid = it; }
        if (it < mid) { mid = it; ms = st; mxy = rt; }
      }
    };
    auto scan = [&](int32_t s0, int32_t s1) __attribute__((always_inline)) {
      for (int32_t base = s0; base < s1; base += kSW) {
        int4 r = make_int4(0, 0, -1, 0);
        if (base + lane < s1) r = g_rec[base + lane];
        chunk(base + lane, r);
      }
    };
    auto result = [&]() __attribute__((always_inline)) -> int32_t {  // the slot; *nrec = its record {x, y, id}
      if (!have) return -1;
      // the points on the empty circle: a fan from the smallest id of the polygon p, q, first .. last
      if (ip < iq && ip < mid) { *nrec = make_int4(f.x, f.y, fid, 0); return fs; }
      if (iq < mid) { *nrec = make_int4(l.x, l.y, lid, 0); return ls; }
      *nrec = make_int4(mxy.x, mxy.y, mid, 0);
      return ms;
    };
    // the cells around p first: they hold the answer for an interior point and bound the cap for the rows below
    if (cached) {
      chunk(cslot, crec);
      // the whole disk inside the block: nothing else can be in the cap
      if (have && ccx - rad >= ix0 && ccx + rad < ix1 && ccy - rad >= iy0 && ccy + rad < iy1) return result();
    } else {
      for (int32_t cy = max(pcy - 1, 0); cy <= min(pcy + 1, G - 1); ++cy) scan(g_start[cy * G + sxa], g_start[cy * G + sxb + 1]);
    }
    // every grid row the current cap can reach, outwards from p's row
    const double A = (double)s * (double)(q.x - p.x), B = (double)s * (double)(q.y - p.y);  // side(x, y) = A (y - py) - B (x - px) > 0
    // kSW grid rows at a time, one per lane: past the disk / cannot hold a candidate / the run of cells to scan
    auto sweep = [&](int dir) __attribute__((always_inline)) {  // +1: rows pcy, pcy + 1, ...; -1: rows pcy - 1, pcy - 2, ...
      for (int32_t j0 = dir > 0 ? pcy : pcy - 1; j0 >= 0 && j0 < G; j0 += kSW * dir) {
        const int32_t j = j0 + dir * lane;
        const bool in = j >= 0 && j < G;
        bool stop = false, keep = false;
        int32_t ca = 0, cb = -1;
        double ylo = 0.0, yhi = 0.0;
        if (in) {
          const DtRow rw = g_row[j];
          ylo = rw.ylo; yhi = rw.yhi;
          double xa = -INFINITY, xb = INFINITY;
          if (rad < INFINITY) {
            const double d = ylo > ccy ? ylo - ccy : (yhi < ccy ? ccy - yhi : 0.0);
            if (d > rad) {
              stop = true;
            } else {
              const double w = sqrt(rad * rad - d * d) * (1.0 + 1.0e-12) + 1.0;
              xa = ccx - w; xb = ccx + w;
            }
          }
          if (!stop && rw.xlo <= rw.xhi) {
            xa = fmax(xa, (double)rw.xlo); xb = fmin(xb, (double)rw.xhi);
            const double h = A * ((A > 0.0 ? yhi : ylo) - (double)p.y);  // max of A (y - py) over the row
            // the corner of the row's rectangle that is furthest on side s: not on it (candidates have side >= 1) -> none
            const double tb = B * ((B > 0.0 ? xa : xb) - (double)p.x);
            bool may = xa <= xb && !(h - tb + 1.0e-15 * (fabs(h) + fabs(tb)) < 0.5);
            if (may) {
              if (B > 0.0) {
                const double t = h / B;
                xb = fmin(xb, (double)p.x + t + 1.0 + 1.0e-12 * fabs(t));
              } else if (B < 0.0) {
                const double t = h / B;
                xa = fmax(xa, (double)p.x + t - 1.0 - 1.0e-12 * fabs(t));
              }
              may = xa <= xb;
            }
            if (may) {
              const double sc = (double)G / (double)bx.spanx;
              ca = clampi(floor((xa - (double)bx.minx) * sc - 1.0e-6), 0, G - 1);
              cb = clampi(floor((xb - (double)bx.minx) * sc + 1.0e-6), 0, G - 1);
              // (a run that reaches beyond the block meets the block's cells again: a candidate seen twice neither
              // beats the best nor changes the ties)
              keep = !(cached && j >= pcy - 1 && j <= pcy + 1 && ca >= sxa && cb <= sxb);
            }
          }
        }
        ++n_rows;
        const unsigned long long mstop = sballot(in && stop);
        unsigned long long mkeep = sballot(keep);
        if (mstop) mkeep &= (1ull << (__ffsll(mstop) - 1)) - 1;  // (lanes are in visiting order: rows before the first one past the disk)
        while (mkeep) {
          const int w = __ffsll(mkeep) - 1;
          mkeep &= mkeep - 1;
          if (rad < INFINITY) {  // has the disk shrunk past this row meanwhile?
            const double yl = bcastd(ylo, w), yh = bcastd(yhi, w);
            const double d = yl > ccy ? yl - ccy : (yh < ccy ? ccy - yh : 0.0);
            if (d > rad) return;
          }
          const int32_t jj = j0 + dir * w;
          ++n_scans;
          scan(g_start[jj * G + bcast(ca, w)], g_start[jj * G + bcast(cb, w) + 1]);
        }
        if (mstop) return;
      }
    };
    sweep(+1);
    sweep(-1);
    return result();
  };

  int32_t n = 0;
  auto emit = [&](int32_t a, int32_t b) __attribute__((always_inline)) {  // triangle (p, a, b), counter-clockwise
    if (ip < a && ip < b) {
      if (n < room && lane == 0) { out[3 * n] = ip; out[3 * n + 1] = a; out[3 * n + 2] = b; }
      ++n;
    }
  };
  const int32_t q0 = nearest();
  if (q0 >= 0) {
    // counter-clockwise from q0 until the star closes; at a hull edge: back to q0 and clockwise to the other hull edge
    int32_t cur = q0, steps = 0;
    const int4 q0rec = g_rec[q0];
    int4 cr = q0rec;
    int dir = +1;
    for (;;) {
      int4 rr = cr;
      const int32_t r = next(cur, make_int2(cr.x, cr.y), cr.z, dir, &rr);
      if (r < 0) {
        if (dir < 0) break;
        dir = -1; cur = q0; cr = q0rec;
        if (!WRITE && lane == 0) atomicAdd(&flags[1], 1);  // an open star: a boundary vertex
        continue;
      }
      if (dir > 0) emit(cr.z, rr.z); else emit(rr.z, cr.z);
      cur = r; cr = rr;
      if (r == q0) {
        if (dir < 0 && lane == 0) atomicOr(&flags[0], kErrWrap);  // (an open star cannot close)
        break;
      }
      if (++steps > g.V) { if (lane == 0) atomicOr(&flags[0], kErrWrap); break; }
    }
  }
  if (lane == 0 && g.dbg && !WRITE) {
    g.dbg[4 * ip] = n; g.dbg[4 * ip + 1] = n_chunks; g.dbg[4 * ip + 2] = n_rows; g.dbg[4 * ip + 3] = n_scans;
  }
  if (lane == 0) {
    if (!WRITE) tcnt[ip] = n;
    else if (n != room) atomicOr(&flags[0], kErrCount);
  }
}

// The star kernel twice (r05): as the compiler allocates it (197 VGPRs, 2 waves per SIMD: the shortest star, best while the
// wavefront slots are not saturated -- frames below ~30 k features) and with the registers capped for 3 waves per SIMD
// (168 VGPRs + 112 B of scratch: each star a little slower, half as many again in flight -- 50 k points 0.78 -> 0.74 ms,
// 200 k 2.54 -> 2.41, but 1.2 k 0.105 -> 0.109: profiles/r05_delaunay_variants.txt).
template <bool WRITE>
__global__ void FLAME_DT_WAVES_ATTR __launch_bounds__(kStarWG) k_dt_star(DtView g, int32_t* flags, int32_t* tcnt, const int32_t* __restrict__ toff,
                                                 int32_t* __restrict__ stash, int32_t* __restrict__ tris, int32_t tri_cap) {
  dt_star_body<WRITE>(g, flags, tcnt, toff, stash, tris, tri_cap);
}
template <bool WRITE>
__global__ void __attribute__((amdgpu_waves_per_eu(3))) __launch_bounds__(kStarWG) k_dt_star_dense(DtView g, int32_t* flags, int32_t* tcnt, const int32_t* __restrict__ toff,
                                                 int32_t* __restrict__ stash, int32_t* __restrict__ tris, int32_t tri_cap) {
  dt_star_body<WRITE>(g, flags, tcnt, toff, stash, tris, tri_cap);
}
constexpr int32_t kStarDenseFrom = 30000;  // points from which the 3-waves variant runs


inline size_t al256(size_t v) { return (v + 255) & ~(size_t)255; }

// out[0..n) = exclusive scan of in[0..n), out[n] = total (also *total when given); n <= 2^20
void scan_ints(hipStream_t s, int32_t* in, int32_t n, int32_t* out, int32_t* sums, int32_t clear_in, int32_t* total) {
  if (n <= 1024 * kScanOne) {
    hipLaunchKernelGGL(k_dt_scan_one, dim3(1), dim3(1024), 0, s, in, n, out, clear_in, total);
    return;
  }
  const unsigned nb = (unsigned)((n + 1023) / 1024);
  hipLaunchKernelGGL(k_dt_scan_sums, dim3(nb), dim3(1024), 0, s, in, n, sums);
  hipLaunchKernelGGL(k_dt_scan_top, dim3(1), dim3(1024), 0, s, sums, (int32_t)nb, out + n, total);
  hipLaunchKernelGGL(k_dt_scan_blocks, dim3(nb), dim3(1024), 0, s, in, n, sums, out, clear_in);
}

}  // namespace

void DelaunayScratch::release() {
  if (list_pending) (void)host_list();
  if (dev) (void)hipFree(dev);
  if (pin) (void)hipHostFree(pin);
  if (ev_list) (void)hipEventDestroy(ev_list);
  if (ev_done) (void)hipEventDestroy(ev_done);
  if (s_list) (void)hipStreamDestroy(s_list);
  ev_done = nullptr; s_list = nullptr;
  dev = nullptr; pin = nullptr; dev_cap = pin_cap = 0;
  last_V = last_T = -1; last_list = nullptr; last_dev = nullptr; ev_list = nullptr; list_pending = false;
}

const int32_t* DelaunayScratch::host_list() {
  if (list_pending) {
    list_pending = false;
    if (hipEventSynchronize(ev_list) != hipSuccess) { (void)hipGetLastError(); last_list = nullptr; last_V = last_T = -1; }
  }
  return last_list;
}

int delaunay_device(hipStream_t s, DelaunayScratch* sc, int32_t V, const float* pos, int32_t tri_cap, int32_t* tris_out,
                    int32_t* T_out) {
  if (!sc || V < 0 || tri_cap < 0 || !T_out || (V > 0 && !pos) || (tri_cap > 0 && !tris_out)) return FLAME_HIP_ERR_ARG;
  *T_out = 0;
  const bool keep = tri_cap == 0 && tris_out == nullptr;  // the list stays in the library (device + asynchronous host copy)
  if (sc->list_pending) (void)sc->host_list();           // (the arena is about to be rewritten)
  sc->last_hull = 0; sc->last_live = V;
  sc->last_V = sc->last_T = -1; sc->last_list = nullptr; sc->last_dev = nullptr;
  if (V < 3) return 0;
  if (V > (1 << 20)) return FLAME_HIP_ERR_ARG;  // (the scans; a frame has 10^3..10^5 features)
  const auto t0 = std::chrono::steady_clock::now();
  const int32_t G = std::max(1, std::min(256, (int32_t)std::ceil(std::sqrt(0.5 * (double)V))));
  const int32_t ncell = G * G;
  const int32_t tmax = 2 * V;  // T <= 2 V - 5
  // ---- arenas ----
  size_t off = 0;
  auto take = [&](size_t bytes) { const size_t o = off; off += al256(bytes); return o; };
  const size_t o_flags = take(sizeof(int32_t) * kFlagWords), o_pos = take(sizeof(float2) * (size_t)V), o_ixy = take(sizeof(int2) * (size_t)V);
  const size_t o_cell = take(sizeof(int32_t) * (size_t)V), o_cnt = take(sizeof(int32_t) * ((size_t)ncell + 1));
  const size_t o_start = take(sizeof(int32_t) * ((size_t)ncell + 1)), o_rec0 = take(sizeof(int4) * (size_t)V), o_rec = take(sizeof(int4) * (size_t)V);
  const size_t o_rows = take(sizeof(DtRow) * (size_t)G), o_sums = take(sizeof(int32_t) * 1024);
  const size_t o_tcnt = take(sizeof(int32_t) * ((size_t)V + 1)), o_toff = take(sizeof(int32_t) * ((size_t)V + 1));
  const size_t o_order = take(sizeof(int32_t) * (size_t)V), o_rowb = take(sizeof(int32_t) * 2 * 256);
  const size_t o_stash = take(sizeof(int32_t) * 3 * kStash * (size_t)V), o_tris = take(sizeof(int32_t) * 3 * (size_t)tmax);
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
  int4* rec0 = reinterpret_cast<int4*>(d + o_rec0);
  int4* rec = reinterpret_cast<int4*>(d + o_rec);
  DtRow* rows = reinterpret_cast<DtRow*>(d + o_rows);
  int32_t* sums = reinterpret_cast<int32_t*>(d + o_sums);
  int32_t* tcnt = reinterpret_cast<int32_t*>(d + o_tcnt);
  int32_t* toff = reinterpret_cast<int32_t*>(d + o_toff);
  int32_t* order = reinterpret_cast<int32_t*>(d + o_order);
  int32_t* rowb = reinterpret_cast<int32_t*>(d + o_rowb);
  int32_t* stash = reinterpret_cast<int32_t*>(d + o_stash);
  int32_t* dtris = reinterpret_cast<int32_t*>(d + o_tris);
  int32_t* hflags = reinterpret_cast<int32_t*>(sc->pin + p_flags);
  int32_t* htris = reinterpret_cast<int32_t*>(sc->pin + p_tris);

  std::memcpy(sc->pin + p_pos, pos, sizeof(float2) * (size_t)V);
  const bool small = V <= kSmallV && G <= kSmallG;
  if (small) {
    hipLaunchKernelGGL(k_dt_prep_small, dim3(1), dim3(1024), 0, s, reinterpret_cast<const float2*>(sc->pin + p_pos), V, G, flags, start, rec, rows);
  } else {
    DT_HIPCHK(hipMemcpyAsync(dpos, sc->pin + p_pos, sizeof(float2) * (size_t)V, hipMemcpyHostToDevice, s));
    const int B = 256;
    const unsigned gv = (unsigned)((V + B - 1) / B);
    hipLaunchKernelGGL(k_dt_init, dim3((unsigned)((ncell + 1 + B - 1) / B)), dim3(B), 0, s, cnt, ncell + 1, flags);
    hipLaunchKernelGGL(k_dt_snap, dim3(std::min(gv, 256u)), dim3(B), 0, s, dpos, V, ixy, flags);
    hipLaunchKernelGGL(k_dt_count, dim3(gv), dim3(B), 0, s, ixy, V, G, flags, cell_of, cnt);
    scan_ints(s, cnt, ncell, start, sums, 1, nullptr);
    hipLaunchKernelGGL(k_dt_scatter, dim3(gv), dim3(B), 0, s, ixy, cell_of, V, start, cnt, rec0);
    hipLaunchKernelGGL(k_dt_dups, dim3(gv), dim3(B), 0, s, rec0, rec, start, V, flags);
    hipLaunchKernelGGL(k_dt_rows, dim3((unsigned)G), dim3(64), 0, s, rec, start, G, flags, rows);
    hipLaunchKernelGGL(k_dt_order_rows, dim3(1), dim3(256), 0, s, start, G, rowb, rowb + 256);
    hipLaunchKernelGGL(k_dt_order_fill, dim3(gv), dim3(B), 0, s, start, rowb, rowb + 256, G, V, order);
  }
#ifndef FLAME_DT_STATS  // (dev aid, tools/exp/delaunay_stats.py: work per star -- a variant build, -DFLAME_DT_STATS=1)
#define FLAME_DT_STATS 0
#endif
  const bool dt_stats = FLAME_DT_STATS != 0;
  int32_t* dbg = nullptr;
  if (dt_stats) {
    (void)hipMalloc(reinterpret_cast<void**>(&dbg), sizeof(int32_t) * 4 * (size_t)V);
    (void)hipMemsetAsync(dbg, 0, sizeof(int32_t) * 4 * (size_t)V, s);
  }
  DtView view;
  view.rec = rec; view.start = start; view.row = rows; view.flags = flags; view.G = G; view.V = V; view.dbg = dbg;
  view.order = small ? nullptr : order;  // (a small frame's stars all run at once)
  const int per_wg = kStarWG / kSW;  // stars per workgroup
  const unsigned gs = (unsigned)((V + per_wg - 1) / per_wg);
  const bool dense = V >= kStarDenseFrom;
  if (dense) hipLaunchKernelGGL(k_dt_star_dense<false>, dim3(gs), dim3(kStarWG), 0, s, view, flags, tcnt, toff, stash, dtris, tmax);
  else hipLaunchKernelGGL(k_dt_star<false>, dim3(gs), dim3(kStarWG), 0, s, view, flags, tcnt, toff, stash, dtris, tmax);
  scan_ints(s, tcnt, V, toff, sums, 0, flags + 3);
  if (dense) hipLaunchKernelGGL(k_dt_star_dense<true>, dim3(gs), dim3(kStarWG), 0, s, view, flags, tcnt, toff, stash, dtris, tmax);
  else hipLaunchKernelGGL(k_dt_star<true>, dim3(gs), dim3(kStarWG), 0, s, view, flags, tcnt, toff, stash, dtris, tmax);
  DT_HIPCHK(hipGetLastError());
  // flags and the list leave together (T = 2 n - 2 - h is within a few triangles of the 2 V the buffer holds: copying
  // the whole buffer costs nothing over copying T triangles, and saves the round trip that would bring T first)
  // (keep mode: only the flags come back now -- one round trip for T --; the list follows behind the caller's back)
  DT_HIPCHK(hipMemcpyAsync(hflags, flags, sizeof(int32_t) * kFlagWords, hipMemcpyDeviceToHost, s));
  if (!keep) DT_HIPCHK(hipMemcpyAsync(htris, dtris, sizeof(int32_t) * 3 * (size_t)tmax, hipMemcpyDeviceToHost, s));
  DT_HIPCHK(hipStreamSynchronize(s));
  if (dbg) {
    std::vector<int32_t> h(4 * (size_t)V);
    (void)hipMemcpy(h.data(), dbg, sizeof(int32_t) * h.size(), hipMemcpyDeviceToHost);
    (void)hipFree(dbg);
    int64_t sum[4] = {0, 0, 0, 0};
    int32_t mx[4] = {0, 0, 0, 0}, arg[4] = {0, 0, 0, 0};
    for (int32_t v = 0; v < V; ++v)
      for (int k = 0; k < 4; ++k) {
        sum[k] += h[4 * (size_t)v + k];
        if (h[4 * (size_t)v + k] > mx[k]) { mx[k] = h[4 * (size_t)v + k]; arg[k] = v; }
      }
    std::fprintf(stderr, "[dt] V %d G %d: own triangles mean %.2f max %d | chunks mean %.1f max %d (point %d at %.1f, %.1f) | row batches mean %.1f max %d | row scans mean %.1f max %d\n",
                 V, G, (double)sum[0] / V, mx[0], (double)sum[1] / V, mx[1], arg[1], pos[2 * arg[1]], pos[2 * arg[1] + 1],
                 (double)sum[2] / V, mx[2], (double)sum[3] / V, mx[3]);
  }
  const int32_t err = hflags[0], hull = hflags[1], live = V - hflags[2], T = hflags[3];
  sc->last_hull = hull; sc->last_live = live;
  if (err & kErrRange) return FLAME_HIP_ERR_ARG;  // a coordinate outside |u|, |v| < 2^13 pixels, or not finite
  if (err || T < 0 || T > tmax || (T > 0 && T != 2 * live - 2 - hull)) return FLAME_HIP_ERR_STATE;
  if (keep) {
    if (T > 0) {
      if (!sc->ev_list) DT_HIPCHK(hipEventCreateWithFlags(&sc->ev_list, hipEventDisableTiming));
      if (!sc->s_list) DT_HIPCHK(hipStreamCreateWithFlags(&sc->s_list, hipStreamNonBlocking));
      // (`s` has been synchronised: the list is complete; its copy-out runs beside whatever the caller stages in next)
      DT_HIPCHK(hipMemcpyAsync(htris, dtris, sizeof(int32_t) * 3 * (size_t)T, hipMemcpyDeviceToHost, sc->s_list));
      DT_HIPCHK(hipEventRecord(sc->ev_list, sc->s_list));
      sc->list_pending = true;
    }
  } else {
    if (T > tri_cap) return FLAME_HIP_ERR_ARG;
    if (T > 0) std::memcpy(tris_out, htris, sizeof(int32_t) * 3 * (size_t)T);
  }
  *T_out = T;
  sc->last_V = V; sc->last_T = T; sc->last_list = htris; sc->last_dev = dtris;
  sc->last_ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count();
  return 0;
}

}  // namespace flamehip
