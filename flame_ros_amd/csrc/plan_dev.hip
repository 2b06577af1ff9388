// flame_ros_amd/csrc/plan_dev.hip -- see plan_dev.h.  The graph plan built by HIP kernels
// (SURVEY.md 8f row f3).  Stage by stage the same rules as plan.cpp, so the arrays are identical:
//
//   A  partition     recursive coordinate bisection on two PRESORTED lists (the vertex ids along x and
//                    along y, total order (coordinate, id)): per level the box of a segment is the two
//                    ends of its lists, an int64 prefix sum of the integer cost weights along the
//                    chosen axis' list and the split rule of plan.cpp split_range() evaluated at every
//                    position, then a stable partition of both lists (one 64-bit flag scan); deep
//                    levels: one workgroup per subtree, the same in LDS (k_rcb_subtree)
//   B  vertex order  one workgroup per tile: (Morton code of the pixel position, vertex id) sorted in LDS
//   C  edge order    by counting: buckets (source, cross-tile flag) in tile order, scan, cursor fill,
//                    every bucket sorted by original edge id by its own thread
//   D  incidence CSR by counting: per-vertex rows, every row sorted by original edge id
//   E  triangle CSR  the same over the 3T corners, on a second stream
//   F  tiles, pass 1 one workgroup per tile: breadth-first halo rings over the CSR with an LDS bitmap
//                    of the vertices, rings sorted by internal id, local edge count
//   G  tiles, pass 2 the local edge list in the order (level, owned, source, edge id) by bucket counting
//                    (no keys, no sort), gather lists, incidence slots (odd pitch per 64-vertex group),
//                    local edge records
// Frame streams take shortcuts through these stages that leave the same kind of plan behind: the
// partition of a frame from the previous frame's tile map (k_reuse_*, "Partition REUSE"), the counting
// passes' atomics doubling as ranks, fused launches (k_he_unique, k_edge_rows_gather, k_tile_fused,
// k_publish) and, for frames of up to 2 k vertices, everything in front of the tile pass in one
// launch of one workgroup (k_mini_plan).
// Library code: rocprim's device radix sort (the two entry sorts), device scan, block radix sort, called directly (r05:
// no hipcub veneer).  rocprim
// sorts fewer than ~1 M keys by block sort + log2 merge passes (7 launches for 50 k keys, 19 for
// 300 k), which is why every sort that could be a counting pass is one.
#include "plan_dev.h"

#include <rocprim/block/block_radix_sort.hpp>
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>
#include <rocprim/functional.hpp>

#include <algorithm>
#include <chrono>
#include <climits>
#include <cmath>
#include <cstdio>
#include <cstdlib>

namespace flamehip {
namespace {

constexpr int kSegCap = 1024;     // segments (= tiles) per level at most
constexpr int kIdBits = 22;       // vertex ids in sort keys
constexpr int kEdgeBits = 24;     // edge / triangle ids in sort keys
constexpr int kCapExt = 2048;     // local vertices per tile (largest kernel configuration)
constexpr int kCapEdge = 6144;    // local edges per tile (largest kernel configuration)
constexpr int kSortPad = 8192;    // bound of a tile's local edge count in the packed look-back totals
constexpr int kHash = 4096;       // LDS hash slots (global -> local vertex id)
constexpr int kMetaWords = 32;    // per tile: 0 n_ext, 1 e_loc, 2 n_upd, 3 fail, 4..20 ring_end, 21..23 offsets
constexpr int kP1Threads = 512, kP2Threads = 1024;

#define HIPRET(expr)                   \
  do {                                 \
    hipError_t e__ = (expr);           \
    if (e__ != hipSuccess) return e__; \
  } while (0)

__device__ __forceinline__ uint32_t ord_f(float f) {  // order-preserving float -> uint
  const uint32_t b = __float_as_uint(f);
  return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float unord_f(uint32_t u) {
  return __uint_as_float((u & 0x80000000u) ? (u & 0x7fffffffu) : ~u);
}
__device__ __forceinline__ uint32_t spread16(uint32_t x) {
  x &= 0xffff; x = (x | (x << 8)) & 0x00ff00ff; x = (x | (x << 4)) & 0x0f0f0f0f;
  x = (x | (x << 2)) & 0x33333333; x = (x | (x << 1)) & 0x55555555;
  return x;
}

template <int NTB, class T>
__device__ void bitonic_sort(T* a, int m) {  // m a power of two, ascending
  const int tid = threadIdx.x;
  for (int k = 2; k <= m; k <<= 1)
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int p = tid; p < (m >> 1); p += NTB) {  // pair p: i = lower index of the pair
        const int i = ((p & ~(j - 1)) << 1) | (p & (j - 1));
        const int ixj = i | j;
        const T x = a[i], y = a[ixj];
        const bool asc = (i & k) == 0;
        if ((x > y) == asc) { a[i] = y; a[ixj] = x; }
      }
      __syncthreads();
    }
}

__device__ __forceinline__ int next_pow2(int n) { int m = 1; while (m < n) m <<= 1; return m; }

// ------------------------------------------------------------------------------------------
// Stage A: recursive coordinate bisection, one level per round of kernels.
// seg tables (int32, kSegCap each): lo, hi, leaves, first; two tables ping-pong per level.
// ------------------------------------------------------------------------------------------
struct SegTab { int32_t *lo, *hi, *leaves, *first; };

__global__ void k_rcb_init(int32_t V, int32_t ntiles, int32_t* perm, int32_t* seg_pos, SegTab t,
                           int32_t* nseg, uint32_t* bbox, int32_t* mid_raw) {
  const int32_t p = blockIdx.x * 256 + threadIdx.x;
  if (p < V) { perm[p] = p; seg_pos[p] = 0; }
  if (p == 0) {
    t.lo[0] = 0; t.hi[0] = V; t.leaves[0] = ntiles; t.first[0] = 0;
    nseg[0] = 1;
    bbox[0] = bbox[1] = 0xffffffffu; bbox[2] = bbox[3] = 0u;
    mid_raw[0] = V;
  }
}

// bounding box of the whole graph for a lone subtree (one workgroup walks the range; same-address
// atomics serialise: an atomic version of the per-level boxes once took ~100 us per level)
__global__ __launch_bounds__(1024) void k_rcb_bbox(const int32_t* __restrict__ nseg, SegTab t,
                                                   const int32_t* __restrict__ perm,
                                                   const float2* __restrict__ pos, uint32_t* bbox) {
  __shared__ uint32_t s_red[4][16];
  const int s = blockIdx.x;
  if (s >= nseg[0] || t.leaves[s] <= 1) return;
  const int32_t lo = t.lo[s], hi = t.hi[s];
  uint32_t mnx = 0xffffffffu, mny = 0xffffffffu, mxx = 0u, mxy = 0u;
  for (int32_t p = lo + threadIdx.x; p < hi; p += 1024) {
    const float2 q = pos[perm[p]];
    const uint32_t ux = ord_f(q.x), uy = ord_f(q.y);
    mnx = min(mnx, ux); mny = min(mny, uy); mxx = max(mxx, ux); mxy = max(mxy, uy);
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    mnx = min(mnx, (uint32_t)__shfl_xor((int)mnx, off, 64));
    mny = min(mny, (uint32_t)__shfl_xor((int)mny, off, 64));
    mxx = max(mxx, (uint32_t)__shfl_xor((int)mxx, off, 64));
    mxy = max(mxy, (uint32_t)__shfl_xor((int)mxy, off, 64));
  }
  const int w = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) { s_red[0][w] = mnx; s_red[1][w] = mny; s_red[2][w] = mxx; s_red[3][w] = mxy; }
  __syncthreads();
  if (threadIdx.x < 4) {
    const int c = threadIdx.x;
    uint32_t v = s_red[c][0];
    for (int k = 1; k < 16; ++k) v = c < 2 ? min(v, s_red[c][k]) : max(v, s_red[c][k]);
    bbox[4 * s + c] = v;
  }
}


__global__ void k_save_gbbox(const uint32_t* bbox, float* gbbox) {
  if (threadIdx.x < 4) gbbox[threadIdx.x] = unord_f(bbox[threadIdx.x]);
}

// Sort keys of the two entry sorts: the vertex ids sorted along x and along y in the total order
// (coordinate, id) are the two lists every bisection level partitions.
__global__ __launch_bounds__(256) void k_rank_keys(int32_t V, const float2* __restrict__ pos, int axis,
                                                   uint32_t* keys, uint32_t* vals) {
  const int32_t v = blockIdx.x * 256 + threadIdx.x;
  if (v >= V) return;
  const float2 q = pos[v];
  keys[v] = ord_f(axis ? q.y : q.x);  // the (stable) radix sort breaks ties by input order = id
  vals[v] = (uint32_t)v;
}




// first position m of the segment with (2 acc(m) + w_m) leaves >= 2 total l1 (plan.cpp split_range)
__global__ __launch_bounds__(256) void k_rcb_mid(int32_t V, const int32_t* __restrict__ seg_pos, SegTab t,
                                                 const long long* __restrict__ wsort,
                                                 const long long* __restrict__ wscan, int32_t* mid_raw) {
  const int32_t p = blockIdx.x * 256 + threadIdx.x;
  if (p >= V) return;
  const int32_t s = seg_pos[p];
  const int32_t L = t.leaves[s];
  if (L <= 1) return;
  const int32_t lo = t.lo[s], hi = t.hi[s], l1 = L / 2;
  const long long base = lo > 0 ? wscan[lo - 1] : 0;
  const long long total = wscan[hi - 1] - base;
  const long long rhs = 2 * total * l1;
  const long long w = wsort[p];
  const long long before = wscan[p] - w - base;
  const bool c = (2 * before + w) * L >= rhs;
  bool cprev = false;
  if (p > lo) {
    const long long wp = wsort[p - 1];
    const long long bp = wscan[p - 1] - wp - base;
    cprev = (2 * bp + wp) * L >= rhs;
  }
  if (c && !cprev) mid_raw[s] = p;
}

// one block: children of every segment, next level's table, reset of the next level's bbox / mid
__global__ __launch_bounds__(kSegCap) void k_rcb_split(const int32_t* nseg_cur, int32_t* nseg_next, SegTab cur,
                                                        SegTab nxt, const int32_t* mid_raw_cur,
                                                        int32_t* mid_raw_next, int weighted, int32_t* child_base,
                                                        int32_t* mid_out, uint32_t* bbox) {
  __shared__ int32_t sc[kSegCap];
  const int s = threadIdx.x;
  const int n = nseg_cur[0];
  int32_t lo = 0, hi = 0, L = 0, first = 0, mid = 0, nchild = 0;
  if (s < n) {
    lo = cur.lo[s]; hi = cur.hi[s]; L = cur.leaves[s]; first = cur.first[s];
    if (L > 1) {
      const int32_t l1 = L / 2;
      if (weighted) {
        mid = mid_raw_cur[s];
        mid = max(lo + l1, min(mid, hi - (L - l1)));
        mid = max(lo, min(mid, hi));
      } else {
        mid = lo + (int32_t)(((long long)(hi - lo) * l1) / L);
      }
      nchild = 2;
    } else {
      nchild = 1;
    }
  }
  sc[s] = nchild;
  __syncthreads();
  for (int off = 1; off < kSegCap; off <<= 1) {  // inclusive scan
    const int32_t v = s >= off ? sc[s - off] : 0;
    __syncthreads();
    sc[s] += v;
    __syncthreads();
  }
  const int32_t base = sc[s] - nchild;
  if (s == kSegCap - 1) nseg_next[0] = min(sc[s], kSegCap);
  __syncthreads();
  if (s < n) {
    child_base[s] = base;
    mid_out[s] = mid;
    if (base + nchild <= kSegCap) {
      if (nchild == 2) {
        const int32_t l1 = L / 2;
        nxt.lo[base] = lo; nxt.hi[base] = mid; nxt.leaves[base] = l1; nxt.first[base] = first;
        nxt.lo[base + 1] = mid; nxt.hi[base + 1] = hi; nxt.leaves[base + 1] = L - l1; nxt.first[base + 1] = first + l1;
        mid_raw_next[base] = mid; mid_raw_next[base + 1] = hi;
      } else {
        nxt.lo[base] = lo; nxt.hi[base] = hi; nxt.leaves[base] = L; nxt.first[base] = first;
        mid_raw_next[base] = hi;
      }
    }
  }
  // next level's boxes
  bbox[4 * s] = bbox[4 * s + 1] = 0xffffffffu;
  bbox[4 * s + 2] = bbox[4 * s + 3] = 0u;
}



// several scratch arrays zeroed by ONE launch (a build had a dozen separate memsets)
constexpr int kZeroJobs = 5;
struct ZeroJob { int32_t* p[kZeroJobs]; int64_t n[kZeroJobs]; };
__global__ __launch_bounds__(256) void k_zero4(ZeroJob j) {
  const int64_t stride = (int64_t)gridDim.x * 256;
  for (int a = 0; a < kZeroJobs; ++a)
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < j.n[a]; i += stride) j.p[a][i] = 0;
}
inline void zero4(hipStream_t s, int32_t* a, int64_t na, int32_t* b = nullptr, int64_t nb = 0, int32_t* c = nullptr,
                  int64_t nc = 0, int32_t* d = nullptr, int64_t nd = 0, int32_t* e = nullptr, int64_t ne = 0) {
  ZeroJob j = {{a, b, c, d, e}, {a ? na : 0, b ? nb : 0, c ? nc : 0, d ? nd : 0, e ? ne : 0}};
  int64_t m = 0;
  for (int k = 0; k < kZeroJobs; ++k) m = std::max(m, j.n[k]);
  hipLaunchKernelGGL(k_zero4, dim3((unsigned)std::min<int64_t>(1024, std::max<int64_t>(1, (m + 255) / 256))), dim3(256), 0, s, j);
}

// exclusive prefix of one value per thread over the workgroup (a[] = scratch of >= 16 entries):
// shuffle scan inside the wave, the 16 wave totals through LDS -- two barriers
template <class T>
__device__ __forceinline__ T block_exclusive(T v, T* a) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  T inc = v;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const T u = __shfl_up(inc, off, 64);
    if (lane >= off) inc += u;
  }
  if (lane == 63) a[wave] = inc;
  __syncthreads();
  T base = 0;
  for (int w = 0; w < wave; ++w) base += a[w];
  __syncthreads();
  return base + inc - v;
}


// ------------------------------------------------------------------------------------------
// Single-launch prefix sums.  The library's device scan is two launches (look-back state init + scan) and
// a plan build has a dozen scans of a few 10^4..10^5 items on its critical path, every dependent
// launch ~4-5 us of host-bound latency: here a scan is ONE launch.  A block scans its tile of
// kScanTile items, publishes its total {value, then -- after s_waitcnt vmcnt(0) -- the launch's epoch
// as the flag; agent-scope (sc1) stores} and reads the totals of the blocks before it (sc1 loads,
// polling the flag).  No reset between launches (a flag only ever matches the launch that wrote it);
// all blocks are co-resident (grid <= kScanMaxBlocks <= 2 blocks per CU), so polling cannot starve
// the producer; the poll is bounded all the same (flags word, bit 32: the build is then rejected and
// redone by the host builder).  The input of a tile comes through a functor, so the level loop's
// gather / flag kernels are fused into their scans.
// ------------------------------------------------------------------------------------------
constexpr int kScanThreads = 256, kScanItems = 4, kScanTile = kScanThreads * kScanItems;
constexpr int kScanMaxBlocks = 448;
struct ScanState {
  unsigned long long* agg;  // kScanMaxBlocks block totals
  uint32_t* flag;           // kScanMaxBlocks: epoch of the launch that published agg[b]
  uint32_t epoch;
  int32_t* err;             // builder flags word
};

template <class T>
__device__ __forceinline__ T scan_lookback(T block_total, const ScanState& st, T* sh /* >= 1 entry */) {
  const int b = blockIdx.x, tid = threadIdx.x;
  if (tid == 0) {
    __hip_atomic_store(&st.agg[b], (unsigned long long)block_total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (the compiler may drop its own wait: MI355X guide, hand-offs)
    __hip_atomic_store(&st.flag[b], st.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  if (tid < 64) {
    T acc = 0;
    for (int j = tid; j < b; j += 64) {
      int spin = 0;
      while (__hip_atomic_load(&st.flag[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != st.epoch) {
        __builtin_amdgcn_s_sleep(1);
        if (++spin > (1 << 22)) { atomicOr(st.err, 32); break; }
      }
      acc += (T)__hip_atomic_load(&st.agg[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 64);
    if (tid == 0) sh[0] = acc;
  }
  __syncthreads();
  return sh[0];
}

// out[i] = sum of load(j) for j < i (exclusive) or j <= i (inclusive); store(i, item, prefix)
template <class T, bool INCLUSIVE, class Load, class Store>
__device__ __forceinline__ void scan_tile(int64_t n, const ScanState& st, Load load, Store store) {
  __shared__ T sh_a[16];
  __shared__ T sh_p[2];
  const int tid = threadIdx.x;
  const int64_t base = (int64_t)blockIdx.x * kScanTile + (int64_t)tid * kScanItems;
  T v[kScanItems];
  T sum = 0;
#pragma unroll
  for (int k = 0; k < kScanItems; ++k) {
    v[k] = (base + k < n) ? load(base + k) : (T)0;
    sum += v[k];
  }
  const T ex = block_exclusive<T>(sum, sh_a);
  if (tid == kScanThreads - 1) sh_p[1] = ex + sum;
  __syncthreads();
  const T before = scan_lookback<T>(sh_p[1], st, sh_p);
  T run = before + ex;
#pragma unroll
  for (int k = 0; k < kScanItems; ++k) {
    if (base + k < n) store(base + k, v[k], INCLUSIVE ? run + v[k] : run);
    run += v[k];
  }
}

template <bool INCLUSIVE>
__global__ __launch_bounds__(kScanThreads) void k_scan_i32(const int32_t* __restrict__ in, int32_t* __restrict__ out,
                                                           int64_t n, ScanState st) {
  scan_tile<int32_t, INCLUSIVE>(n, st, [&](int64_t i) { return in[i]; },
                                [&](int64_t i, int32_t, int32_t p) { out[i] = p; });
}

// ---- global bisection levels on two presorted lists ----
// lx / ly = the vertex ids sorted along x / along y in the total order (coordinate, id), both grouped
// by segment (a segment is the same position range [lo, hi) in both).  A level never sorts: the box
// of a segment is the two ends of its lists, the order along the chosen axis is that axis' list, and
// both lists are PARTITIONED stably into the children (one scan of the side flags, x flags in the
// low and y flags in the high half of a 64-bit counter).
__global__ __launch_bounds__(256) void k_lvl_axis(const int32_t* __restrict__ nseg, SegTab t,
                                                  const uint32_t* __restrict__ lx, const uint32_t* __restrict__ ly,
                                                  const float2* __restrict__ pos, int32_t* axis, float* gbbox) {
  const int s = blockIdx.x * 256 + threadIdx.x;
  if (s >= nseg[0]) return;
  const int32_t lo = t.lo[s], hi = t.hi[s];
  int a = 0;
  if (hi > lo) {
    const float x0 = pos[lx[lo]].x, x1 = pos[lx[hi - 1]].x, y0 = pos[ly[lo]].y, y1 = pos[ly[hi - 1]].y;
    if (t.leaves[s] > 1) a = (y1 - y0) > (x1 - x0) ? 1 : 0;
    if (gbbox && s == 0) { gbbox[0] = x0; gbbox[1] = y0; gbbox[2] = x1; gbbox[3] = y1; }  // level 0: the whole frame
  }
  axis[s] = a;
}

__global__ __launch_bounds__(256) void k_lvl_wgather(int32_t V, const int32_t* __restrict__ seg_pos,
                                                     const int32_t* __restrict__ axis,
                                                     const uint32_t* __restrict__ lx, const uint32_t* __restrict__ ly,
                                                     const int32_t* __restrict__ w_int, long long* wsort) {
  const int32_t p = blockIdx.x * 256 + threadIdx.x;
  if (p >= V) return;
  wsort[p] = w_int[axis[seg_pos[p]] ? ly[p] : lx[p]];
}

// the same fused into its inclusive scan (one launch instead of gather + two scan launches)
__global__ __launch_bounds__(kScanThreads) void k_lvl_wscan(int32_t V, const int32_t* __restrict__ seg_pos,
                                                            const int32_t* __restrict__ axis,
                                                            const uint32_t* __restrict__ lx, const uint32_t* __restrict__ ly,
                                                            const int32_t* __restrict__ w_int, long long* wsort,
                                                            long long* wscan, ScanState st) {
  scan_tile<long long, true>(V, st, [&](int64_t p) { return (long long)w_int[axis[seg_pos[p]] ? ly[p] : lx[p]]; },
                             [&](int64_t p, long long w, long long inc) { wsort[p] = w; wscan[p] = inc; });
}

// side of every vertex (1 = right child) = its position along its segment's axis against the split
__global__ __launch_bounds__(256) void k_lvl_side(int32_t V, const int32_t* __restrict__ seg_pos,
                                                  const int32_t* __restrict__ axis,
                                                  const int32_t* __restrict__ leaves,
                                                  const int32_t* __restrict__ mid,
                                                  const uint32_t* __restrict__ lx, const uint32_t* __restrict__ ly,
                                                  int32_t* side) {
  const int32_t p = blockIdx.x * 256 + threadIdx.x;
  if (p >= V) return;
  const int32_t s = seg_pos[p];
  side[axis[s] ? ly[p] : lx[p]] = (leaves[s] > 1 && p >= mid[s]) ? 1 : 0;
}

__global__ __launch_bounds__(256) void k_lvl_flags(int32_t V, const uint32_t* __restrict__ lx,
                                                   const uint32_t* __restrict__ ly, const int32_t* __restrict__ side,
                                                   long long* flags) {
  const int32_t p = blockIdx.x * 256 + threadIdx.x;
  if (p >= V) return;
  flags[p] = (long long)side[lx[p]] | ((long long)side[ly[p]] << 32);
}

// the flags fused into their inclusive scan
__global__ __launch_bounds__(kScanThreads) void k_lvl_fscan(int32_t V, const uint32_t* __restrict__ lx,
                                                            const uint32_t* __restrict__ ly,
                                                            const int32_t* __restrict__ side, long long* flags,
                                                            long long* scan, ScanState st) {
  scan_tile<long long, true>(V, st, [&](int64_t p) { return (long long)side[lx[p]] | ((long long)side[ly[p]] << 32); },
                             [&](int64_t p, long long f, long long inc) { flags[p] = f; scan[p] = inc; });
}

// stable partition of both lists + the segment of every position on the next level
__global__ __launch_bounds__(256) void k_lvl_scatter(int32_t V, int32_t* seg_pos, SegTab t,
                                                     const int32_t* __restrict__ mid,
                                                     const int32_t* __restrict__ child_base,
                                                     const long long* __restrict__ flags,
                                                     const long long* __restrict__ scan,  // inclusive
                                                     const uint32_t* __restrict__ lx, const uint32_t* __restrict__ ly,
                                                     uint32_t* lx2, uint32_t* ly2) {
  const int32_t p = blockIdx.x * 256 + threadIdx.x;
  if (p >= V) return;
  const int32_t s = seg_pos[p];
  const int32_t lo = t.lo[s], m = mid[s];
  const bool split = t.leaves[s] > 1;
  const long long f = flags[p], ex = scan[p] - f, base = lo > 0 ? scan[lo - 1] : 0;
  const int32_t fx = (int32_t)(f & 1), fy = (int32_t)((f >> 32) & 1);
  const int32_t rx = (int32_t)((ex & 0xffffffffll) - (base & 0xffffffffll));  // right-flags before p in the segment
  const int32_t ry = (int32_t)((ex >> 32) - (base >> 32));
  lx2[fx ? m + rx : p - rx] = lx[p];
  ly2[fy ? m + ry : p - ry] = ly[p];
  seg_pos[p] = child_base[s] + ((split && p >= m) ? 1 : 0);
}

__global__ __launch_bounds__(256) void k_lvl_posx(int32_t V, const uint32_t* __restrict__ lx, uint32_t* posx,
                                                  int32_t* perm) {
  const int32_t p = blockIdx.x * 256 + threadIdx.x;
  if (p >= V) return;
  posx[lx[p]] = (uint32_t)p;
  perm[p] = (int32_t)lx[p];
}

// ------------------------------------------------------------------------------------------
// Stage A, deep levels: once a segment holds <= kSubCap vertices its whole remaining bisection
// subtree is finished by ONE workgroup in LDS, instead of ~12 dependent launches per level.
// The subtree's vertices are sorted ONCE along x and once along y (global ranks = the total order
// (coordinate, id); a lone subtree that is the whole graph sorts the coordinates themselves, stably
// from id order).  Every level then only PARTITIONS the two lists stably (one block scan of the
// side flags per list): both stay sorted inside every segment, so the bounding box of a segment is
// the two ends of its lists and the order along the chosen axis is simply that axis' list.  The
// split rule (integer weight prefix along the chosen axis) is the host builder's.
// ------------------------------------------------------------------------------------------
constexpr int kSubCap = 8192;    // vertices of a subtree
constexpr int kSubLeaves = 256;  // tiles of a subtree
constexpr int kSubThreads = 1024;
constexpr int kSubItems = kSubCap / kSubThreads;
typedef rocprim::block_radix_sort<uint32_t, kSubThreads, kSubItems, uint32_t> SubPairSort;
// (after the entry sorts the same LDS holds the subtree's weights by local index)
constexpr size_t kSubSortBytes =
    sizeof(SubPairSort::storage_type) > (size_t)kSubCap * 4 ? sizeof(SubPairSort::storage_type) : (size_t)kSubCap * 4;
// lists (2 x u16), global ids (u32), side + segment of position (u8), thread partials (i64),
// 2 segment tables x 4 + 5 per-segment words + 2 per-segment i64 prefixes, sort scratch
constexpr size_t kSubLdsBytes = (size_t)kSubCap * (2 + 2 + 4 + 1 + 1) + (size_t)kSubThreads * 8 +
                                (size_t)kSubLeaves * (8 + 5) * 4 + (size_t)kSubLeaves * 2 * 8 + kSubSortBytes + 64;

static_assert(kSubLdsBytes <= 160 * 1024 - 256, "subtree kernel LDS");

__global__ __launch_bounds__(kSubThreads) void k_rcb_subtree(const int32_t* nseg_cur, SegTab cur, SegTab out,
                                                             int32_t* nseg_out, int32_t ntiles, int32_t* perm,
                                                             int32_t* seg_pos, const float2* __restrict__ pos,
                                                             const uint32_t* __restrict__ glx,
                                                             const uint32_t* __restrict__ gly,
                                                             const uint32_t* __restrict__ posx,
                                                             const int32_t* __restrict__ w_int, int weighted, int vb,
                                                             int direct, int cap, int32_t* flags) {
  // direct != 0: the subtree is the whole graph in id order (perm = identity) and sorts itself along
  // both axes; else its two sorted lists are the ranges [glo, ghi) of the global lists glx / gly
  // (posx = position of a vertex in glx), and perm is written only at the end
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int sidx = blockIdx.x, tid = threadIdx.x;
  if (sidx == 0 && tid == 0) nseg_out[0] = ntiles;
  if (sidx >= nseg_cur[0]) return;
  const int32_t glo = cur.lo[sidx], ghi = cur.hi[sidx], gleaves = cur.leaves[sidx], gfirst = cur.first[sidx];
  const int n = ghi - glo;
  if (n > cap || gleaves > kSubLeaves) {
    // Overflow: the host sees the flag at the builder's first sync and rebuilds one level later.
    // Until then the later stages still run on this partition, so leave a VALID one behind (the
    // segment cut into equal runs of the current order) -- stale table contents here sent the
    // tile kernels out of bounds.
    if (tid == 0) atomicOr(&flags[0], 16);
    for (int j = tid; j < gleaves; j += kSubThreads) {
      const int32_t a = glo + (int32_t)(((long long)n * j) / gleaves);
      const int32_t b = glo + (int32_t)(((long long)n * (j + 1)) / gleaves);
      out.lo[gfirst + j] = a; out.hi[gfirst + j] = b; out.leaves[gfirst + j] = 1; out.first[gfirst + j] = gfirst + j;
      for (int32_t q = a; q < b; ++q) { seg_pos[q] = gfirst + j; if (!direct) perm[q] = (int32_t)glx[q]; }
    }
    return;
  }
  uint16_t* lx = reinterpret_cast<uint16_t*>(smem);                      // local index at x-order position p
  uint16_t* ly = lx + kSubCap;                                           // ... at y-order position p
  uint32_t* gid = reinterpret_cast<uint32_t*>(ly + kSubCap);             // vertex id of local index l
  uint8_t* side = reinterpret_cast<uint8_t*>(gid + kSubCap);             // by local index: 1 = right child
  uint8_t* segof = side + kSubCap;                                       // by position: local segment
  long long* part = reinterpret_cast<long long*>(segof + kSubCap);       // kSubThreads
  int32_t* tb = reinterpret_cast<int32_t*>(part + kSubThreads);          // 2 tables x 4 x kSubLeaves
  int32_t* mid_raw = tb + 8 * kSubLeaves;
  int32_t* mid_fin = mid_raw + kSubLeaves;
  int32_t* cbase = mid_fin + kSubLeaves;
  int32_t* axis = cbase + kSubLeaves;                                    // 1: split along y
  int32_t* rlo = axis + kSubLeaves;                                      // right-flags before the segment
  long long* pre_lo = reinterpret_cast<long long*>(rlo + kSubLeaves);    // weight prefix before the segment
  long long* pre_hi = pre_lo + kSubLeaves;                               // ... through its last position
  char* sort_tmp = reinterpret_cast<char*>(((uintptr_t)(pre_hi + kSubLeaves) + 15) & ~(uintptr_t)15);
  __shared__ int s_nloc, s_more, s_more_next;
  int32_t *lo = tb, *hi = tb + kSubLeaves, *lv = tb + 2 * kSubLeaves, *fi = tb + 3 * kSubLeaves;
  int32_t *lo2 = tb + 4 * kSubLeaves, *hi2 = tb + 5 * kSubLeaves, *lv2 = tb + 6 * kSubLeaves, *fi2 = tb + 7 * kSubLeaves;
  for (int p = tid; p < n; p += kSubThreads) {
    gid[p] = direct ? (uint32_t)perm[glo + p] : glx[glo + p];
    segof[p] = 0;
    if (!direct) { lx[p] = (uint16_t)p; ly[p] = (uint16_t)(posx[gly[glo + p]] - (uint32_t)glo); }
  }
  if (tid == 0) { lo[0] = 0; hi[0] = n; lv[0] = gleaves; fi[0] = gfirst; s_nloc = 1; s_more = gleaves > 1; s_more_next = 0; }
  __syncthreads();
  // ---- a lone subtree sorts itself: stable from id order, so the coordinate alone gives the
  // (coordinate, id) order ----
  const int m = next_pow2(max(n, 1));
  for (int ax = 0; ax < 2 && direct; ++ax) {
    uint16_t* list = ax ? ly : lx;
    if (m <= 2048) {  // small windows: the bitonic network beats the fixed-size radix sort
      uint64_t* packed = reinterpret_cast<uint64_t*>(sort_tmp);
      for (int p = tid; p < m; p += kSubThreads) {
        if (p >= n) { packed[p] = ~0ull; continue; }
        const float2 q = pos[gid[p]];
        packed[p] = ((uint64_t)ord_f(ax ? q.y : q.x) << 32) | (uint32_t)p;
      }
      __syncthreads();
      bitonic_sort<kSubThreads, uint64_t>(packed, m);
      for (int p = tid; p < n; p += kSubThreads) list[p] = (uint16_t)(packed[p] & 0x1fffu);
      __syncthreads();
    } else {
      SubPairSort::storage_type& tmp = *reinterpret_cast<SubPairSort::storage_type*>(sort_tmp);
      uint32_t keys[kSubItems], vals[kSubItems];
#pragma unroll
      for (int i = 0; i < kSubItems; ++i) {
        const int p = tid * kSubItems + i;
        vals[i] = (uint32_t)p;
        keys[i] = 0xffffffffu;  // padding stays behind every real key (stable)
        if (p < n) { const float2 q = pos[gid[p]]; keys[i] = ord_f(ax ? q.y : q.x); }
      }
      SubPairSort().sort(keys, vals, tmp, 0, 32);
      __syncthreads();
#pragma unroll
      for (int i = 0; i < kSubItems; ++i) {
        const int p = tid * kSubItems + i;
        if (p < n) list[p] = (uint16_t)vals[i];
      }
      __syncthreads();
    }
  }
  int32_t* wl = reinterpret_cast<int32_t*>(sort_tmp);  // weights by local index (the sort scratch is free now)
  if (weighted)
    for (int p = tid; p < n; p += kSubThreads) wl[p] = w_int[gid[p]];
  __syncthreads();
  // ---- levels ----
  while (s_more) {
    const int nloc = s_nloc;
    for (int k = tid; k < nloc; k += kSubThreads) {
      mid_raw[k] = hi[k];
      axis[k] = 0;
      if (lv[k] > 1 && hi[k] > lo[k]) {  // the lists are sorted inside the segment: its box is their two ends
        const float ex = pos[gid[lx[hi[k] - 1]]].x - pos[gid[lx[lo[k]]]].x;
        const float ey = pos[gid[ly[hi[k] - 1]]].y - pos[gid[ly[lo[k]]]].y;
        axis[k] = ey > ex ? 1 : 0;
      }
    }
    __syncthreads();
    if (weighted) {  // integer weight prefix along every segment's chosen axis, split rule per position
      long long pref[kSubItems];
      int32_t wv[kSubItems];
      long long acc = 0;
      // (branch-free, so that the eight dependent LDS chains of a thread overlap)
#pragma unroll
      for (int i = 0; i < kSubItems; ++i) {
        const int pc = min(tid * kSubItems + i, n - 1);
        wv[i] = wl[axis[segof[pc]] ? ly[pc] : lx[pc]];
      }
#pragma unroll
      for (int i = 0; i < kSubItems; ++i) {
        if (tid * kSubItems + i >= n) wv[i] = 0;
        acc += wv[i];
        pref[i] = acc;
      }
      const long long base = block_exclusive<long long>(acc, part);
#pragma unroll
      for (int i = 0; i < kSubItems; ++i) {
        const int p = tid * kSubItems + i;
        if (p >= n) continue;
        pref[i] += base;
        const int k = segof[p];
        if (p == lo[k]) pre_lo[k] = pref[i] - wv[i];
        if (p == hi[k] - 1) pre_hi[k] = pref[i];
      }
      __syncthreads();
#pragma unroll
      for (int i = 0; i < kSubItems; ++i) {
        const int p = tid * kSubItems + i;
        if (p >= n) continue;
        const int k = segof[p];
        const int32_t L = lv[k];
        if (L <= 1) continue;
        const int32_t l1 = L / 2, slo = lo[k];
        const long long sbase = pre_lo[k];
        const long long rhs = 2 * (pre_hi[k] - sbase) * l1;
        const long long w = wv[i];
        const bool c = (2 * (pref[i] - w - sbase) + w) * L >= rhs;
        bool cprev = false;
        if (p > slo) {
          long long wp, pp;
          if (i > 0) { wp = wv[i - 1]; pp = pref[i - 1]; }
          else { wp = wl[axis[k] ? ly[p - 1] : lx[p - 1]]; pp = base; }
          cprev = (2 * (pp - wp - sbase) + wp) * L >= rhs;
        }
        if (c && !cprev) mid_raw[k] = p;
      }
      __syncthreads();
    }
    {  // children of every local segment: one thread per segment, child slots by a block scan
      int32_t L = 0, slo = 0, shi = 0, mid = 0, nch = 0;
      if (tid < nloc) {
        L = lv[tid]; slo = lo[tid]; shi = hi[tid];
        if (L > 1) {
          const int32_t l1 = L / 2;
          if (weighted) {
            mid = mid_raw[tid];
            mid = max(slo + l1, min(mid, shi - (L - l1)));
            mid = max(slo, min(mid, shi));
          } else {
            mid = slo + (int32_t)(((long long)(shi - slo) * l1) / L);
          }
          nch = 2;
        } else {
          mid = shi;
          nch = 1;
        }
      }
      const int32_t nn = block_exclusive<int32_t>(nch, reinterpret_cast<int32_t*>(part));
      if (tid < nloc) {
        cbase[tid] = nn;
        mid_fin[tid] = mid;
        if (nch == 2) {
          const int32_t l1 = L / 2;
          lo2[nn] = slo; hi2[nn] = mid; lv2[nn] = l1; fi2[nn] = fi[tid];
          lo2[nn + 1] = mid; hi2[nn + 1] = shi; lv2[nn + 1] = L - l1; fi2[nn + 1] = fi[tid] + l1;
          if (l1 > 1 || L - l1 > 1) s_more_next = 1;
        } else {
          lo2[nn] = slo; hi2[nn] = shi; lv2[nn] = L; fi2[nn] = fi[tid];
        }
        if (tid == nloc - 1) s_nloc = nn + nch;
      }
    }
    __syncthreads();
    if (tid == 0) { s_more = s_more_next; s_more_next = 0; }
    __syncthreads();
    // side of every vertex = its position along the chosen axis against the split
    for (int p = tid; p < n; p += kSubThreads) {
      const int k = segof[p];
      side[axis[k] ? ly[p] : lx[p]] = (lv[k] > 1 && p >= mid_fin[k]) ? 1 : 0;
    }
    __syncthreads();
    // stable partition of both lists: left children keep their order in front, right behind
    for (int ax = 0; ax < 2; ++ax) {
      uint16_t* list = ax ? ly : lx;
      uint16_t it[kSubItems];
      int32_t ex[kSubItems];
      int32_t cnt = 0;
      int32_t fl[kSubItems];
#pragma unroll
      for (int i = 0; i < kSubItems; ++i) it[i] = list[min(tid * kSubItems + i, n - 1)];
#pragma unroll
      for (int i = 0; i < kSubItems; ++i) fl[i] = side[it[i]];
#pragma unroll
      for (int i = 0; i < kSubItems; ++i) {
        ex[i] = cnt;
        if (tid * kSubItems + i < n) cnt += fl[i];
      }
      const int32_t base = block_exclusive<int32_t>(cnt, reinterpret_cast<int32_t*>(part));
#pragma unroll
      for (int i = 0; i < kSubItems; ++i) {
        const int p = tid * kSubItems + i;
        if (p >= n) continue;
        ex[i] += base;
        if (p == lo[segof[p]]) rlo[segof[p]] = ex[i];
      }
      __syncthreads();
#pragma unroll
      for (int i = 0; i < kSubItems; ++i) {
        const int p = tid * kSubItems + i;
        if (p >= n) continue;
        const int k = segof[p];
        const int32_t r = ex[i] - rlo[k];  // right-flags before p inside the segment
        list[fl[i] ? mid_fin[k] + r : p - r] = it[i];
      }
      __syncthreads();
    }
    for (int p = tid; p < n; p += kSubThreads) {
      const int k = segof[p];
      segof[p] = (uint8_t)(cbase[k] + ((lv[k] > 1 && p >= mid_fin[k]) ? 1 : 0));
    }
    __syncthreads();
    { int32_t* t; t = lo; lo = lo2; lo2 = t; t = hi; hi = hi2; hi2 = t; t = lv; lv = lv2; lv2 = t; t = fi; fi = fi2; fi2 = t; }
  }
  // every local segment is one tile now
  for (int p = tid; p < n; p += kSubThreads) {
    perm[glo + p] = (int32_t)gid[lx[p]];
    seg_pos[glo + p] = fi[segof[p]];
  }
  for (int k = tid; k < s_nloc; k += kSubThreads) {
    const int t = fi[k];
    out.lo[t] = glo + lo[k]; out.hi[t] = glo + hi[k]; out.leaves[t] = lv[k]; out.first[t] = t;
  }
}

// after the last level: every segment is one tile, in tile order
__global__ __launch_bounds__(kSegCap) void k_rcb_check(const int32_t* nseg, SegTab t, int32_t ntiles, int32_t* flags) {
  const int s = threadIdx.x;
  if (s == 0 && nseg[0] != ntiles) atomicOr(&flags[0], 1);
  if (s < ntiles && (t.leaves[s] != 1 || t.first[s] != s)) atomicOr(&flags[0], 1);
}

// ------------------------------------------------------------------------------------------
// Stage B: order inside tiles = Morton code of the pixel position (plan.cpp, order_mode 1)
// ------------------------------------------------------------------------------------------
// One workgroup per tile sorts the tile's own vertices by (Morton code, id) in LDS (a tile holds a
// few hundred vertices; a library sort of all V 64-bit keys was seven launches).
constexpr int kOrderCap = 2048;
__global__ __launch_bounds__(256) void k_tile_order(int32_t V, int32_t ntiles, const int32_t* __restrict__ tlo,
                                                    const int32_t* __restrict__ thi,
                                                    const float2* __restrict__ pos, const float* __restrict__ gb,
                                                    int32_t* v_i2o, int32_t* v_o2i, int32_t* tile_of_int,
                                                    int32_t* flags) {
  __shared__ uint64_t keys[kOrderCap];
  const int t = blockIdx.x, tid = threadIdx.x;
  const int32_t lo = max(0, min(tlo[t], V)), hi = max(lo, min(thi[t], V));  // (tables of a rejected
  const int n = hi - lo;                                                    //  partition stay in range)
  if (n > kOrderCap) {  // cannot be a tile of any kernel configuration: keep the order, flag the plan
    if (tid == 0) atomicOr(&flags[0], 4);
    for (int p = tid; p < n; p += 256) { const int32_t v = v_i2o[lo + p]; v_o2i[v] = lo + p; tile_of_int[lo + p] = t; }
    return;
  }
  const float mnx = gb[0], mny = gb[1], mxx = gb[2], mxy = gb[3];
  const int m = next_pow2(max(n, 1));
  for (int p = tid; p < m; p += 256) {
    uint64_t k = ~0ull;
    if (p < n) {
      const int32_t v = v_i2o[lo + p];
      const float2 q = pos[v];
      // (clamped: identity for a box that is this frame's own; a reused partition keeps the box of
      // the frame it was made from)
      const uint32_t qx = (uint32_t)(65535.0f * (fminf(fmaxf(q.x, mnx), mxx) - mnx) / fmaxf(mxx - mnx, 1e-20f));
      const uint32_t qy = (uint32_t)(65535.0f * (fminf(fmaxf(q.y, mny), mxy) - mny) / fmaxf(mxy - mny, 1e-20f));
      k = ((uint64_t)(spread16(qx) | (spread16(qy) << 1)) << 32) | (uint32_t)v;
    }
    keys[p] = k;
  }
  __syncthreads();
  bitonic_sort<256, uint64_t>(keys, m);
  for (int p = tid; p < n; p += 256) {
    const int32_t v = (int32_t)(uint32_t)keys[p];
    v_i2o[lo + p] = v;
    v_o2i[v] = lo + p;
    tile_of_int[lo + p] = t;
  }
}

// ------------------------------------------------------------------------------------------
// Stage C: edge order (owner tile of the source, level 0 before level 1, source, original id)
// ------------------------------------------------------------------------------------------
// Edge order by counting: bucket (source vertex u, cross-tile flag) in the order of the edge array --
// per tile first the buckets (u, 0) of its vertices in internal order, then the buckets (u, 1) --
// i.e. index 2 lo_t + (cross ? n_t : 0) + (u - lo_t).  Count, exclusive scan, fill through a cursor,
// every bucket sorted by original edge id (k_csr_rows): the order the stable sort on
// (tile, cross, source) gave, in 7 short launches instead of a 9-launch merge sort of E keys.
__device__ __forceinline__ int32_t edge_bucket(const int2 ij, int32_t V, const int32_t* __restrict__ v_o2i,
                                               const int32_t* __restrict__ tile_of_int,
                                               const int32_t* __restrict__ tlo, const int32_t* __restrict__ thi,
                                               int32_t* flags) {
  if (ij.x < 0 || ij.y < 0 || ij.x >= V || ij.y >= V || ij.x == ij.y) {  // build_plan's index check
    if (flags) atomicOr(&flags[0], 2);
    return 0;
  }
  const int32_t si = v_o2i[ij.x], sj = v_o2i[ij.y];
  const int32_t ti = tile_of_int[si], tj = tile_of_int[sj];
  const int32_t lo = tlo[ti], n = thi[ti] - lo;
  return 2 * lo + (ti == tj ? 0 : n) + (si - lo);
}

__global__ __launch_bounds__(256) void k_edge_count(int32_t E, int32_t V, const int2* __restrict__ edges,
                                                    const int32_t* __restrict__ v_o2i,
                                                    const int32_t* __restrict__ tile_of_int,
                                                    const int32_t* __restrict__ tlo, const int32_t* __restrict__ thi,
                                                    int32_t* cnt, int32_t* rank, int32_t* flags, int32_t* zero_me) {
  const int32_t e = blockIdx.x * 256 + threadIdx.x;
  if (e <= V) zero_me[e] = 0;  // stage D's degree counts (V + 1), used two launches later (grid >= V + 1)
  if (e >= E) return;
  // the value the count returns is the entry's place in its bucket: the fill needs no second atomic
  // (the order inside a bucket is arbitrary either way; k_csr_rows sorts it)
  rank[e] = atomicAdd(&cnt[edge_bucket(edges[e], V, v_o2i, tile_of_int, tlo, thi, flags)], 1);
}

__global__ __launch_bounds__(256) void k_edge_fill(int32_t E, int32_t V, const int2* __restrict__ edges,
                                                   const int32_t* __restrict__ v_o2i,
                                                   const int32_t* __restrict__ tile_of_int,
                                                   const int32_t* __restrict__ tlo, const int32_t* __restrict__ thi,
                                                   const int32_t* __restrict__ off,
                                                   const int32_t* __restrict__ rank, uint32_t* out) {
  const int32_t e = blockIdx.x * 256 + threadIdx.x;
  if (e >= E) return;
  const int32_t b = edge_bucket(edges[e], V, v_o2i, tile_of_int, tlo, thi, nullptr);
  out[off[b] + rank[e]] = (uint32_t)e;
}




// ------------------------------------------------------------------------------------------
// Stage D / E: incidence CSR (ascending ORIGINAL edge id per vertex), triangle CSR
// ------------------------------------------------------------------------------------------
// ---- CSRs by counting (stages D and E): degree count -> exclusive scan -> fill through a cursor
// per row (the order inside a row is whatever the atomics give) -> every row sorted by its own
// thread.  The result is the one the stable sort by vertex gave (rows in ascending original edge /
// triangle id), in 7 short launches instead of the ~20 of a library merge sort of 2E or 3T keys. ----

// entry = (original edge id << 1) | role (1: the vertex is the target): ascending entry = ascending
// original id, the summation order of the arithmetic contract
__global__ __launch_bounds__(256) void k_csr_fill(int32_t E, const int2* __restrict__ eij,
                                                  const int32_t* __restrict__ e_o2i,
                                                  const int32_t* __restrict__ row,
                                                  const int2* __restrict__ rank, uint32_t* out) {
  const int32_t e = blockIdx.x * 256 + threadIdx.x;
  if (e >= E) return;
  const int2 ij = eij[e_o2i[e]];
  const int2 r = rank[e];
  out[row[ij.x] + r.x] = (uint32_t)e << 1;
  out[row[ij.y] + r.y] = ((uint32_t)e << 1) | 1u;
}

__global__ __launch_bounds__(256) void k_tri_count(int32_t n3, int32_t V, const int32_t* __restrict__ tris,
                                                   const int32_t* __restrict__ v_o2i, int32_t* tris_int,
                                                   int32_t* cnt, int32_t* rank, int32_t* flags) {
  const int32_t k = blockIdx.x * 256 + threadIdx.x;
  if (k >= n3) return;
  const int32_t vo = tris[k];
  if (vo < 0 || vo >= V) { atomicOr(&flags[0], 2); tris_int[k] = 0; return; }
  const int32_t v = v_o2i[vo];
  tris_int[k] = v;
  rank[k] = atomicAdd(&cnt[v], 1);
}

__global__ __launch_bounds__(256) void k_tri_fill(int32_t n3, int32_t V, const int32_t* __restrict__ tris,
                                                  const int32_t* __restrict__ tris_int,
                                                  const int32_t* __restrict__ row,
                                                  const int32_t* __restrict__ rank, uint32_t* out) {
  const int32_t k = blockIdx.x * 256 + threadIdx.x;
  if (k >= n3) return;
  const int32_t vo = tris[k];
  if (vo < 0 || vo >= V) return;  // (flagged by k_tri_count: the plan is rejected after the next sync)
  const int32_t v = tris_int[k];
  out[row[v] + rank[k]] = (uint32_t)(k / 3);
}

// one thread per row: ascending order (insertion sort; heap sort for long rows, so that a vertex of
// huge degree costs d log d, not d^2), the rows of a block staged in LDS when they fit; CONVERT: entries (original edge id << 1 | role) become
// (internal edge id | role << 31)
__device__ __forceinline__ void sort_row(uint32_t* a, int d) {
  if (d <= 24) {
    for (int i = 1; i < d; ++i) {
      const uint32_t x = a[i];
      int j = i - 1;
      while (j >= 0 && a[j] > x) { a[j + 1] = a[j]; --j; }
      a[j + 1] = x;
    }
  } else {
    auto sift = [&](int root, int end) {
      for (;;) {
        int child = 2 * root + 1;
        if (child >= end) break;
        if (child + 1 < end && a[child] < a[child + 1]) ++child;
        if (a[root] >= a[child]) break;
        const uint32_t t = a[root]; a[root] = a[child]; a[child] = t;
        root = child;
      }
    };
    for (int i = d / 2 - 1; i >= 0; --i) sift(i, d);
    for (int end = d - 1; end > 0; --end) {
      const uint32_t t = a[0]; a[0] = a[end]; a[end] = t;
      sift(0, end);
    }
  }
}

constexpr int kRowsLds = 6144;  // entries of 256 consecutive rows staged in LDS (else sorted in place)
// CONVERT also leaves, for the tile passes: gadj[s] = the vertex at the other end of incidence s (the
// ring search then needs two dependent loads per step, not three) and ipos[2k + role] = the place of
// internal edge k in the row of its source (role 0) / target (role 1) (the slot of an incidence
// without a search).
template <bool CONVERT>
__global__ __launch_bounds__(256) void k_csr_rows(int32_t V, const int32_t* __restrict__ row,
                                                  const int32_t* __restrict__ e_o2i, uint32_t* out,
                                                  const int2* __restrict__ eij = nullptr, int32_t* gadj = nullptr,
                                                  int32_t* ipos = nullptr) {
  __shared__ uint32_t s_a[kRowsLds];
  const int32_t v0 = blockIdx.x * 256, v1 = min(v0 + 256, V);
  const int32_t r0 = row[v0], r1 = row[v1];
  const int32_t v = v0 + threadIdx.x;
  const bool lds = r1 - r0 <= kRowsLds;  // the block's rows are one contiguous range of the array
  if (lds) {
    for (int32_t i = threadIdx.x; i < r1 - r0; i += 256) s_a[i] = out[r0 + i];
    __syncthreads();
  }
  if (v < V) {
    const int d = row[v + 1] - row[v];
    uint32_t* a = lds ? s_a + (row[v] - r0) : out + row[v];
    sort_row(a, d);
    if (CONVERT && gadj) {
      for (int i = 0; i < d; ++i) {
        const uint32_t x = a[i];
        const int32_t k = e_o2i[x >> 1];
        const int2 ij = eij[k];
        gadj[row[v] + i] = (x & 1u) ? ij.x : ij.y;
        ipos[2 * k + (int32_t)(x & 1u)] = i;
      }
    }
  }
  if (lds) {
    __syncthreads();
    for (int32_t i = threadIdx.x; i < r1 - r0; i += 256) {
      const uint32_t x = s_a[i];
      out[r0 + i] = CONVERT ? ((uint32_t)e_o2i[x >> 1] | ((x & 1u) << 31)) : x;
    }
  } else if (CONVERT && v < V) {
    uint32_t* a = out + row[v];
    const int d = row[v + 1] - row[v];
    for (int i = 0; i < d; ++i) {
      const uint32_t x = a[i];
      a[i] = (uint32_t)e_o2i[x >> 1] | ((x & 1u) << 31);
    }
  }
}


// Stage C's last two launches and stage D's first in one: a block sorts its 256 buckets in LDS
// (as k_csr_rows does), then writes the edge records of exactly those entries and counts the degrees of
// their endpoints for the incidence CSR (the count is the rank again).
__global__ __launch_bounds__(256) void k_edge_rows_gather(int32_t nrows, const int32_t* __restrict__ off, uint32_t* sorted_e,
                                                          const int2* __restrict__ edges,
                                                          const float* __restrict__ alpha,
                                                          const float* __restrict__ beta,
                                                          const float2* __restrict__ pos,
                                                          const int32_t* __restrict__ v_o2i, int32_t* e_i2o,
                                                          int32_t* e_o2i, int2* eij, float4* ew, int32_t V, int32_t E,
                                                          int32_t ntiles, const int32_t* __restrict__ tlo,
                                                          int32_t* estart, float dsign, int32_t* deg_cnt, int2* rank2) {
  __shared__ uint32_t s_a[kRowsLds];
  const int32_t v0 = blockIdx.x * 256, v1 = min(v0 + 256, nrows);
  const int32_t r0 = off[v0], r1 = off[v1];
  const int32_t v = v0 + threadIdx.x;
  const bool lds = r1 - r0 <= kRowsLds;  // the block's rows are one contiguous range of the array
  if (lds) {
    for (int32_t i = threadIdx.x; i < r1 - r0; i += 256) s_a[i] = sorted_e[r0 + i];
    __syncthreads();
  }
  if (v < nrows) sort_row(lds ? s_a + (off[v] - r0) : sorted_e + off[v], off[v + 1] - off[v]);
  __syncthreads();
  // estart[t] = first internal edge owned by tile t = where its first bucket starts
  const int32_t g = blockIdx.x * 256 + threadIdx.x;
  if (g < ntiles) estart[g] = off[2 * max(0, min(tlo[g], V))];
  if (g == ntiles) estart[g] = E;
  for (int32_t k = r0 + threadIdx.x; k < r1; k += 256) {
    const int32_t e = (int32_t)(lds ? s_a[k - r0] : sorted_e[k]);
    if (lds) sorted_e[k] = (uint32_t)e;
    const int2 ij = edges[e];
    e_i2o[k] = e;
    e_o2i[e] = k;
    if (ij.x < 0 || ij.y < 0 || ij.x >= V || ij.y >= V || ij.x == ij.y) {
      // flagged by the bucket count (the plan is rejected after the next sync); keep every index in range
      eij[k] = make_int2(0, 0);
      ew[k] = make_float4(0.f, 0.f, 0.f, 0.f);
      int2 r0v;  // (counted at vertex 0 like any (0, 0) record, so that the fill stays inside the array)
      r0v.x = atomicAdd(&deg_cnt[0], 1);
      r0v.y = atomicAdd(&deg_cnt[0], 1);
      rank2[e] = r0v;
      continue;
    }
    const int32_t si = v_o2i[ij.x], sj = v_o2i[ij.y];
    const float2 pi = pos[ij.x], pj = pos[ij.y];
    eij[k] = make_int2(si, sj);
    ew[k] = make_float4(alpha[e], beta[e], dsign * (pi.x - pj.x), dsign * (pi.y - pj.y));
    int2 r;
    r.x = atomicAdd(&deg_cnt[si], 1);
    r.y = atomicAdd(&deg_cnt[sj], 1);
    rank2[e] = r;
  }
}

// ------------------------------------------------------------------------------------------
// Stage F / G: tiles.  One workgroup per tile.
// ------------------------------------------------------------------------------------------
struct TileGraph {
  int32_t V, depth;
  const int32_t* grow; const int32_t* ginc; const int2* eij;
  const int32_t* e_i2o; const int32_t* e_o2i;
  const int32_t* gadj; const int32_t* ipos;  // k_csr_rows<true>
};

struct TileLds {
  uint32_t* bitmap;  // (V + 31) / 32
  int32_t* ext;      // kCapExt
  int32_t* hkey;     // kHash   (also the ring-sort window before the hash is built)
  int32_t* hval;     // kHash
};

__device__ __forceinline__ uint32_t hash_slot(int32_t gid) { return ((uint32_t)gid * 2654435761u) >> 20; }  // 12 bits

__device__ __forceinline__ int32_t hash_lookup(const TileLds& L, int32_t gid) {
  uint32_t h = hash_slot(gid);
  for (int probe = 0; probe < kHash; ++probe) {
    const int32_t k = L.hkey[h];
    if (k == gid) return L.hval[h];
    if (k == -1) return -1;
    h = (h + 1) & (kHash - 1);
  }
  return -1;
}


constexpr int kRowLanes = 4;  // threads per incidence row in the tile passes

// breadth-first halo rings; returns through s_n / s_ring_end (shared), sets *fail on overflow
template <int NTB>
__device__ void tile_rings(const TileGraph& G, const TileLds& L, int32_t vstart, int32_t n_own, int* s_n,
                           int32_t* s_ring_end, int* s_fail) {
  const int tid = threadIdx.x;
  const int bm_words = (G.V + 31) >> 5;
  for (int i = tid; i < bm_words; i += NTB) L.bitmap[i] = 0u;
  __syncthreads();
  if (n_own > kCapExt) { if (tid == 0) *s_fail = 1; n_own = kCapExt; }
  for (int l = tid; l < n_own; l += NTB) {
    const int32_t v = vstart + l;
    L.ext[l] = v;
    atomicOr(&L.bitmap[v >> 5], 1u << (v & 31));
  }
  if (tid == 0) { *s_n = n_own; s_ring_end[0] = n_own; }
  __syncthreads();
  int prev_lo = 0, prev_hi = n_own;
  for (int r = 1; r <= kMaxDepth; ++r) {
    if (r <= G.depth) {
      // (kRowLanes threads share a vertex's incidence row: the dependent loads row -> entry -> edge of a
      // thread that walks a whole row alone are the latency of these passes)
      for (int it = tid; it < (prev_hi - prev_lo) * kRowLanes; it += NTB) {
        const int32_t v = L.ext[prev_lo + it / kRowLanes];
        for (int32_t s = G.grow[v] + it % kRowLanes; s < G.grow[v + 1]; s += kRowLanes) {
          const int32_t u = G.gadj[s];
          const uint32_t bit = 1u << (u & 31);
          const uint32_t old = atomicOr(&L.bitmap[u >> 5], bit);
          if (!(old & bit)) {
            const int pos = atomicAdd(s_n, 1);
            if (pos < kCapExt) L.ext[pos] = u;
          }
        }
      }
      __syncthreads();
      int n = *s_n;
      if (n > kCapExt) { if (tid == 0) { *s_fail = 1; *s_n = kCapExt; } n = kCapExt; }
      const int cnt = n - prev_hi;
      if (cnt > 1) {  // ring vertices in ascending internal id (any append order -> same list): every vertex
        // counts the smaller ones -- cnt broadcast reads and two barriers, where a bitonic sort of a few
        // hundred ids is ~45 barrier-separated stages (half of this function's time, measured)
        int32_t* win = L.hkey;  // 2 * kHash ints contiguous (hkey, hval)
        for (int i = tid; i < cnt; i += NTB) win[i] = L.ext[prev_hi + i];
        __syncthreads();
        for (int i = tid; i < cnt; i += NTB) {
          const int32_t v = win[i];
          int r = 0;
          for (int j = 0; j < cnt; ++j) r += win[j] < v ? 1 : 0;
          L.ext[prev_hi + r] = v;
        }
      }
      __syncthreads();
      prev_lo = prev_hi; prev_hi = n;
    }
    if (tid == 0) s_ring_end[r] = prev_hi;
  }
  __syncthreads();
}

template <int NTB>
__device__ void tile_hash_build(const TileLds& L, int n_ext) {
  const int tid = threadIdx.x;
  for (int i = tid; i < kHash; i += NTB) L.hkey[i] = -1;
  __syncthreads();
  for (int l = tid; l < n_ext; l += NTB) {
    const int32_t gid = L.ext[l];
    uint32_t h = hash_slot(gid);
    for (;;) {
      const int32_t old = atomicCAS(&L.hkey[h], -1, gid);
      if (old == -1) { L.hval[h] = l; break; }
      h = (h + 1) & (kHash - 1);
    }
  }
  __syncthreads();
}

__device__ __forceinline__ int ring_of(const int32_t* s_ring_end, int lv) {
  int r = 0;
  while (r < kMaxDepth && lv >= s_ring_end[r]) ++r;
  return r;
}

// key of a local edge: [level:5][not owned:1][source local id:16][original edge id:32]; 0 = not local
__device__ __forceinline__ bool local_edge_key(const TileGraph& G, const TileLds& L, const int32_t* s_ring_end,
                                               int32_t k, int lsrc, int rsrc, int32_t dst, uint64_t* key) {
  if (!((L.bitmap[dst >> 5] >> (dst & 31)) & 1u)) return false;
  const int ldst = hash_lookup(L, dst);
  const int rdst = ring_of(s_ring_end, ldst);
  if (G.depth > 0 && min(rsrc, rdst) >= G.depth) return false;  // feeds no updated vertex
  const uint64_t lvl = (uint64_t)max(rsrc, rdst);
  const uint64_t notown = (lvl <= 1 && rsrc != 0) ? 1 : 0;
  *key = (lvl << 49) | (notown << 48) | ((uint64_t)(uint32_t)lsrc << 32) | (uint64_t)(uint32_t)G.e_i2o[k];
  return true;
}

__global__ __launch_bounds__(kP1Threads) void k_tile_pass1(TileGraph G, const int32_t* __restrict__ vstart_tab,
                                                           const int32_t* __restrict__ vend_tab,
                                                           int32_t* tile_ext, int32_t* meta) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  __shared__ int s_n, s_fail, s_ecnt;
  __shared__ int32_t s_ring_end[kMaxDepth + 1];
  const int t = blockIdx.x, tid = threadIdx.x;
  TileLds L;
  L.bitmap = reinterpret_cast<uint32_t*>(smem);
  L.ext = reinterpret_cast<int32_t*>(L.bitmap + ((G.V + 31) >> 5));
  L.hkey = L.ext + kCapExt;
  L.hval = L.hkey + kHash;
  if (tid == 0) { s_fail = 0; s_ecnt = 0; }
  __syncthreads();
  const int32_t vstart = vstart_tab[t], n_own = vend_tab[t] - vstart;
  tile_rings<kP1Threads>(G, L, vstart, n_own, &s_n, s_ring_end, &s_fail);
  const int n_ext = s_n;
  tile_hash_build<kP1Threads>(L, n_ext);
  int cnt = 0;
  for (int it = tid; it < n_ext * kRowLanes; it += kP1Threads) {
    const int lv = it / kRowLanes, sub = it % kRowLanes;
    const int32_t v = L.ext[lv];
    const int rv = ring_of(s_ring_end, lv);
    for (int32_t s = G.grow[v] + sub; s < G.grow[v + 1]; s += kRowLanes) {
      const int32_t ent = G.ginc[s];
      if (ent < 0) continue;  // v is the target; the source adds the edge
      uint64_t key;
      if (local_edge_key(G, L, s_ring_end, ent, lv, rv, G.eij[ent].y, &key)) ++cnt;
    }
    if (sub == 0) tile_ext[(size_t)t * kCapExt + lv] = v;
  }
  if (cnt) atomicAdd(&s_ecnt, cnt);
  __syncthreads();
  if (tid == 0) {
    int32_t* m = meta + (size_t)t * kMetaWords;
    m[0] = n_ext;
    m[1] = s_ecnt;
    m[2] = G.depth == 0 ? n_ext : s_ring_end[G.depth - 1];
    m[3] = (s_fail || s_ecnt > kCapEdge) ? 1 : 0;
  }
  if (tid <= kMaxDepth) meta[(size_t)t * kMetaWords + 4 + tid] = s_ring_end[tid];
}

// exclusive scans of (n_ext, e_loc, n_upd) over the tiles; totals and the fail flag
__global__ __launch_bounds__(kSegCap) void k_tile_offsets(int ntiles, int32_t* meta, int32_t* flags, int32_t cap_nv,
                                                          int32_t cap_ne, int32_t cap_ns) {
  __shared__ int32_t sc[3][kSegCap];
  const int t = threadIdx.x;
  int32_t v[3] = {0, 0, 0};
  if (t < ntiles) {
    const int32_t* m = meta + (size_t)t * kMetaWords;
    v[0] = m[0]; v[1] = m[1]; v[2] = m[2];
    if (m[3]) atomicOr(&flags[0], 4);
  }
  for (int c = 0; c < 3; ++c) sc[c][t] = v[c];
  __syncthreads();
  for (int off = 1; off < kSegCap; off <<= 1) {
    int32_t u[3];
    for (int c = 0; c < 3; ++c) u[c] = t >= off ? sc[c][t - off] : 0;
    __syncthreads();
    for (int c = 0; c < 3; ++c) sc[c][t] += u[c];
    __syncthreads();
  }
  if (t < ntiles) {
    int32_t* m = meta + (size_t)t * kMetaWords;
    for (int c = 0; c < 3; ++c) m[21 + c] = sc[c][t] - v[c];
  }
  if (t == kSegCap - 1) {
    flags[1] = sc[0][t]; flags[2] = sc[1][t]; flags[3] = sc[2][t];
    // speculative tile arrays (sized from the previous frame, no host round trip before pass 2): too small?
    if (cap_nv > 0 && (sc[0][t] > cap_nv || sc[1][t] > cap_ne || sc[2][t] > cap_ns)) atomicOr(&flags[0], 128);
  }
}

// Lane order inside one block of 64 local edges (plan.cpp assign_lanes(), the identical greedy),
// run by ONE wave; lane l is candidate position l.  Per step the k-th edge of the sorted order is
// broadcast, every free lane prices itself from the wave's tables, the cheapest lowest lane takes
// the edge.  The tables live in registers, one entry per lane: lane g*16+c holds the first address
// (+1) gathered in read group g from bank class c (source and target tables), lane w*8+c the
// number of stores of write group w into class c (source- and target-slot tables); lookups are
// lane reads, the winner is found with ballots -- no LDS, no barriers.
// S12 (r05, the deferred application on a plan with 12-byte slots only -- the build-time order of both builders keeps the model
// above): a slot store is a ds_write_b64 of the pair array (4 groups of 16 contiguous lanes, the pair of slot s covers banks
// 2 s, 2 s + 1 of 32: two slots collide when s = s' mod 16) and a ds_write_b32 of the cx array (2 groups of 32 lanes, bank
// s mod 32).  The b96 model above is worth 0.5 % at 200 k vertices where it is worth 5 % on 16-byte slots at 100 k
// (profiles/r05_lane_order_sens.txt).
template <bool S12>
__device__ __forceinline__ void assign_lanes_block(int lane, int b0, int e_loc, int32_t eoff, int32_t nslots,
                                                   uint2* t_eij, float4* t_ew, int32_t* t_emap) {
  const int hl = lane & 31;
  const int rg = ((hl < 4 || (hl >= 12 && hl < 16) || (hl >= 20 && hl < 28)) ? 0 : 1) + 2 * (lane >> 5);
  const int wg = lane >> 3;
  const int c = min(64, e_loc - b0);
  const int src_i = eoff + b0 + min(lane, c - 1);
  const uint2 rec = t_eij[src_i];
  const float4 w = t_ew[src_i];
  const int32_t mp = t_emap[src_i];
  int32_t T_rs = 0, T_rt = 0, T_ws = 0, T_wd = 0;
  int32_t T_ws1 = 0, T_wd1 = 0;  // (S12: the cx array's stores; T_ws / T_wd are the pair array's)
  const int wg16 = lane >> 4, wg32 = lane >> 5;
  bool used = lane >= c;
  int got = lane;
  for (int k = 0; k < c; ++k) {
    const uint32_t ex = (uint32_t)__builtin_amdgcn_readlane((int)rec.x, k);
    const uint32_t ey = (uint32_t)__builtin_amdgcn_readlane((int)rec.y, k);
    const int32_t li = (int32_t)(ex & 0xffffu), lj = (int32_t)(ex >> 16);
    const uint32_t ss = ey & 0xffffu, sd = ey >> 16;
    // a lane without a slot stores into its own trash slot (nslots + lane)
    const uint32_t s1 = ss != 0xffffu ? ss : (uint32_t)(nslots + lane);
    const uint32_t s2 = sd != 0xffffu ? sd : (uint32_t)(nslots + lane);
    const int32_t fs = __shfl(T_rs, rg * 16 + (li & 15), 64), ft = __shfl(T_rt, rg * 16 + (lj & 15), 64);
    int cost;
    if (S12) {
      cost = __shfl(T_ws, wg16 * 16 + (int)(s1 & 15), 64) + __shfl(T_wd, wg16 * 16 + (int)(s2 & 15), 64) +
             __shfl(T_ws1, wg32 * 32 + (int)(s1 & 31), 64) + __shfl(T_wd1, wg32 * 32 + (int)(s2 & 31), 64);
    } else {
      cost = __shfl(T_ws, wg * 8 + (int)(s1 & 7), 64) + __shfl(T_wd, wg * 8 + (int)(s2 & 7), 64);
    }
    if (fs != 0 && fs != li + 1) ++cost;
    if (ft != 0 && ft != lj + 1) ++cost;
    int best = -1;
    for (int cc = 0; best < 0; ++cc) {  // the cheapest free lane, lowest first (costs are small)
      const unsigned long long m = __ballot(!used && cost == cc);
      if (m) best = (int)__builtin_ctzll(m);
    }
    const int hb = best & 31;
    const int rgb = ((hb < 4 || (hb >= 12 && hb < 16) || (hb >= 20 && hb < 28)) ? 0 : 1) + 2 * (best >> 5);
    const int wgb = best >> 3;
    const uint32_t s1b = ss != 0xffffu ? ss : (uint32_t)(nslots + best);
    const uint32_t s2b = sd != 0xffffu ? sd : (uint32_t)(nslots + best);
    if (lane == best) { used = true; got = k; }
    if (lane == rgb * 16 + (li & 15) && T_rs == 0) T_rs = li + 1;
    if (lane == rgb * 16 + (lj & 15) && T_rt == 0) T_rt = lj + 1;
    if (S12) {
      const int g16 = best >> 4, g32 = best >> 5;
      if (lane == g16 * 16 + (int)(s1b & 15)) ++T_ws;
      if (lane == g16 * 16 + (int)(s2b & 15)) ++T_wd;
      if (lane == g32 * 32 + (int)(s1b & 31)) ++T_ws1;
      if (lane == g32 * 32 + (int)(s2b & 31)) ++T_wd1;
    } else {
      if (lane == wgb * 8 + (int)(s1b & 7)) ++T_ws;
      if (lane == wgb * 8 + (int)(s2b & 7)) ++T_wd;
    }
  }
  const uint32_t nx = (uint32_t)__shfl((int)rec.x, got, 64), ny = (uint32_t)__shfl((int)rec.y, got, 64);
  const float wx = __shfl(w.x, got, 64), wy = __shfl(w.y, got, 64), wz = __shfl(w.z, got, 64), ww = __shfl(w.w, got, 64);
  const int32_t nm = __shfl(mp, got, 64);
  if (lane < c) {
    t_eij[eoff + b0 + lane] = make_uint2(nx, ny);
    t_ew[eoff + b0 + lane] = make_float4(wx, wy, wz, ww);
    t_emap[eoff + b0 + lane] = nm;
  }
}

// The same on a finished plan (lane_order = 1: when a plan is solved a second time, see
// flame_hip.cpp): grid (tiles, blocks of 4 edge blocks), one wave per 64-edge block.
template <bool S12>
__global__ __launch_bounds__(256) void k_assign_lanes(const TileDesc* __restrict__ tiles, uint2* t_eij, float4* t_ew,
                                                      int32_t* t_emap) {
  const TileDesc D = tiles[blockIdx.x];
  const int b0 = (blockIdx.y * 4 + (threadIdx.x >> 6)) * 64;
  if (D.n_ext <= 0 || b0 >= D.e_loc) return;
  assign_lanes_block<S12>(threadIdx.x & 63, b0, D.e_loc, D.erec_off, D.nslots, t_eij, t_ew, t_emap);
}

// what a tile's workgroup shares while it is being built
struct TileShared {
  int fail, ecnt, n;
  int32_t ring_end[kMaxDepth + 1], level_end[kMaxDepth + 1];
  int32_t gw[kCapExt / 64], gbase[kCapExt / 64 + 1];
};

struct TileOut {
  const float4* ew; TileDesc* tiles; int32_t* t_vmap; int32_t* t_emap; uint2* t_eij; float4* t_ew; uint32_t* t_srow;
  int32_t* flags; int lane_order;
  long long* prof;  // dev aid (option "plan_timing" = 5): phase stamps of tile 0, or null
};
#define TILE_STAMP(O, n) do { if ((O).prof && blockIdx.x == 0 && threadIdx.x == 0) (O).prof[n] = (long long)__builtin_amdgcn_s_memtime(); } while (0)

// ---- the local edge list of a tile without a sort ----
// The order of the list (plan.cpp: level, owned before not owned, source, original edge id) is, with the
// local vertices numbered ring by ring: for every ring r, first the edges of its vertices that stay
// inside the rings 0..r ("same": level r), then those that lead to ring r + 1 ("next": level r + 1), each
// block by source vertex, then by original id.  (Level 1's owned edges are ring 0's "next" block, its
// not-owned ones ring 1's "same" block.)  That is a bucket order: bucket (source, kind) at
// 2 * ring_start + (kind ? ring_size : 0) + (source - ring_start), and inside a bucket the order of the
// source's incidence row, which is ascending original id already.  So: one walk over the rows counts the
// buckets, a scan makes offsets, a second walk writes every edge's records straight to its place --
// the 64-bit keys and their bitonic sort (~45 of ~180 k ticks per tile at 50 k vertices) are gone.
constexpr int kBucketInts = 2 * kCapExt + 8;

__device__ __forceinline__ int tile_bucket(const int32_t* ring_end, int lv, int rv, int kind) {
  const int rs = rv == 0 ? 0 : ring_end[rv - 1];
  return 2 * rs + (kind ? ring_end[rv] - rs : 0) + (lv - rs);
}

template <int NTB>
struct TileEmit {  // what the second walk needs beside the graph
  int32_t es, e_own, n_upd, eoff;
  const TileOut* O;
};

// kRowLanes (4) lanes walk one local vertex's incidence row together, four entries per step; what a
// lane needs of its neighbours -- how many qualifying entries of its kind sit in front of it -- comes by
// shuffles inside the group, so the place of every edge inside its bucket is known without atomics.
// EMIT = false: bucket sizes into cnt[]; EMIT = true: cnt[] holds the offsets, the records are written.
template <int NTB, bool EMIT>
__device__ __forceinline__ int tile_walk(const TileGraph& G, const TileLds& L, TileShared& S, int32_t* cnt, int n_ext,
                                         const TileEmit<NTB>* E) {
  static_assert(kRowLanes == 4 && NTB % 4 == 0, "groups of four lanes");
  const int tid = threadIdx.x, sub = tid & 3, lane0 = (tid & 63) & ~3;
  int found = 0;
  for (int it = tid; it < n_ext * 4; it += NTB) {
    const int lv = it >> 2;
    const int32_t v = L.ext[lv];
    const int rv = ring_of(S.ring_end, lv);
    const int32_t g0 = G.grow[v], g1 = G.grow[v + 1];
    const int b_same = tile_bucket(S.ring_end, lv, rv, 0), b_next = tile_bucket(S.ring_end, lv, rv, 1);
    int run_same = 0, run_next = 0;
    for (int32_t base = g0; base < g1; base += 4) {
      const int32_t s = base + sub;
      int f = 0;  // 1: same, 2: next
      int32_t ent = 0, ldst = 0;
      if (s < g1) {
        ent = G.ginc[s];
        if (ent >= 0) {  // v is the source (the target's row does not add the edge)
          const int32_t u = G.gadj[s];
          if ((L.bitmap[u >> 5] >> (u & 31)) & 1u) {
            ldst = hash_lookup(L, u);
            const int rd = ring_of(S.ring_end, ldst);
            if (!(G.depth > 0 && min(rv, rd) >= G.depth)) f = rd > rv ? 2 : 1;  // (else: feeds no updated vertex)
          }
        }
      }
      int before = 0, tot_same = 0, tot_next = 0;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int fj = __shfl(f, lane0 + j, 64);
        tot_same += fj == 1; tot_next += fj == 2;
        before += (j < sub && fj == f) ? 1 : 0;
      }
      if (EMIT && f) {
        const int32_t le = cnt[f == 2 ? b_next : b_same] + (f == 2 ? run_next : run_same) + before;
        const int32_t k = ent;
        uint32_t ss = 0xffffu, sd = 0xffffu;
        if (lv < E->n_upd) { const int g = lv >> 6; ss = (uint32_t)(slot_row_start(S.gbase[g], lv, S.gw[g]) + G.ipos[2 * k]); ++found; }
        if (ldst < E->n_upd) { const int g = ldst >> 6; sd = (uint32_t)(slot_row_start(S.gbase[g], ldst, S.gw[g]) + G.ipos[2 * k + 1]); ++found; }
        if (le < E->e_own && k != E->es + le) S.fail = 1;  // owned edges = the prefix, in internal order
        E->O->t_emap[E->eoff + le] = k;
        E->O->t_eij[E->eoff + le] = make_uint2((uint32_t)lv | ((uint32_t)ldst << 16), (ss & 0xffffu) | (sd << 16));
        E->O->t_ew[E->eoff + le] = E->O->ew[k];
      }
      run_same += tot_same; run_next += tot_next;
    }
    if (!EMIT && sub == 0) { cnt[b_same] = run_same; cnt[b_next] = run_next; }
  }
  return found;
}

// exclusive scan of a[0 .. n) in place (LDS), n <= 2 * kCapExt + 1; returns the total.  (Ends with a barrier.)
template <int NTB>
__device__ __forceinline__ int32_t tile_scan(int32_t* a, int n, int32_t* sh /* >= 17 */) {
  const int tid = threadIdx.x;
  const int per = (n + NTB - 1) / NTB;
  const int i0 = tid * per, i1 = min(i0 + per, n);
  int32_t sum = 0;
  for (int i = i0; i < i1; ++i) sum += a[i];
  int32_t run = block_exclusive<int32_t>(sum, sh);
  if (tid == NTB - 1) sh[16] = run + sum;
  for (int i = i0; i < i1; ++i) { const int32_t v = a[i]; a[i] = run; run += v; }
  __syncthreads();
  return sh[16];
}

// bucket counts -> offsets; returns the number of local edges
template <int NTB>
__device__ __forceinline__ int tile_count_edges(const TileGraph& G, const TileLds& L, TileShared& S, int32_t* cnt, int n_ext,
                                                int32_t* sh) {
  if (threadIdx.x == 0) cnt[2 * n_ext] = 0;
  (void)tile_walk<NTB, false>(G, L, S, cnt, n_ext, nullptr);
  __syncthreads();
  return tile_scan<NTB>(cnt, 2 * n_ext + 1, sh);
}

// everything behind the bucket offsets: vertex map, incidence slot rows, the local edge records with
// their slots, level ends, descriptor.  Expects L (ext, bitmap, hash), S.ring_end, S.fail, S.gw[] = 1.
template <int NTB>
__device__ __forceinline__ void tile_emit(const TileGraph& G, const TileLds& L, TileShared& S, int32_t* off, int t,
                                          int32_t vstart, int32_t n_own, int32_t es, int32_t e_own, int n_ext,
                                          int e_loc, int n_upd, int32_t voff, int32_t eoff, int32_t soff,
                                          const TileOut& O) {
  const int tid = threadIdx.x;
  for (int l = tid; l < n_ext; l += NTB) O.t_vmap[voff + l] = L.ext[l];
  if (tid == 0) S.n = 0;  // (spent as the ring counter: now the balance of the halo closure check below)
  // level l ends where ring l's "next" block begins
  if (tid <= kMaxDepth) S.level_end[tid] = off[(tid == 0 ? 0 : S.ring_end[tid - 1]) + S.ring_end[tid]];
  if (e_own > e_loc && tid == 0) S.fail = 1;
  __syncthreads();
  TILE_STAMP(O, 6);
  // ---- incidence slots: one row per updated vertex, odd pitch per 64-vertex group ----
  for (int lv = tid; lv < n_upd; lv += NTB) {
    const int32_t v = L.ext[lv];
    const int32_t deg = G.grow[v + 1] - G.grow[v];
    atomicMax(&S.gw[lv >> 6], deg);
    atomicAdd(&S.n, -deg);  // (back at 0 when every incidence of an updated vertex finds its slot below)
  }
  __syncthreads();
  if (tid == 0) {
    int32_t base = 0;
    const int ng = (n_upd + 63) >> 6;
    for (int g = 0; g < ng; ++g) {
      const int32_t w = slot_group_pitch(S.gw[g]);  // (common.h: the layout's one statement)
      S.gw[g] = w;
      S.gbase[g] = base;
      if (!slot_group_fits(base, w)) { S.fail = 1; break; }
      base += slot_group_span(w);
    }
    S.gbase[kCapExt / 64] = base;
  }
  __syncthreads();
  for (int lv = tid; lv < n_upd; lv += NTB) {
    const int32_t v = L.ext[lv];
    const int g = lv >> 6;
    O.t_srow[soff + lv] = slot_row_word(slot_row_start(S.gbase[g], lv, S.gw[g]), G.grow[v + 1] - G.grow[v]);
  }
  TILE_STAMP(O, 7);
  // ---- local edge records with their two slots (the slot of an incidence = its vertex's row start + its
  // place in the vertex's incidence row, which k_csr_rows<true> left in ipos) ----
  TileEmit<NTB> E;
  E.es = es; E.e_own = e_own; E.n_upd = n_upd; E.eoff = eoff; E.O = &O;
  const int found = tile_walk<NTB, true>(G, L, S, off, n_ext, &E);
  if (found) atomicAdd(&S.n, found);
  __syncthreads();
  // halo closure invariant: every incidence of an updated vertex is a local edge
  if (tid == 0 && S.n != 0) S.fail = 1;
  __syncthreads();
  TILE_STAMP(O, 8);
  // ---- lane order at build time (lane_order = 2): plan.cpp assign_lanes(), the identical greedy ----
  if (O.lane_order && !S.fail) {
    const int32_t nslots = S.gbase[kCapExt / 64];
    for (int b0 = (tid >> 6) * 64; b0 < e_loc; b0 += (NTB / 64) * 64)
      assign_lanes_block<false>(tid & 63, b0, e_loc, eoff, nslots, O.t_eij, O.t_ew, O.t_emap);
  }
  if (tid == 0) {
    TileDesc D = {};
    D.vstart = vstart; D.n_own = n_own; D.n_ext = n_ext;
    D.estart = es; D.e_own = e_own; D.e_loc = e_loc;
    D.n_upd = n_upd; D.depth = G.depth;
    D.vmap_off = voff; D.emap_off = eoff; D.erec_off = eoff; D.srow_off = soff;
    D.nslots = S.gbase[kCapExt / 64];
    for (int r = 0; r <= kMaxDepth; ++r) { D.ring_end[r] = S.ring_end[r]; D.level_end[r] = S.level_end[r]; }
    if (S.fail) { D.n_ext = -1; atomicOr(&O.flags[0], 8); }
    O.tiles[t] = D;
  }
}

__global__ __launch_bounds__(kP2Threads) void k_tile_pass2(TileGraph G, const int32_t* __restrict__ vstart_tab,
                                                           const int32_t* __restrict__ vend_tab,
                                                           const int32_t* __restrict__ estart,
                                                           const int32_t* __restrict__ tile_ext,
                                                           const int32_t* __restrict__ meta, TileOut O) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  __shared__ TileShared S;
  __shared__ int32_t s_sc[20];
  const int t = blockIdx.x, tid = threadIdx.x;
  // a build that has already failed (bad indices, a tile that does not fit, a rejected partition, tile
  // arrays too small for a speculative launch: every bit but pass 2's own 8) must not be continued:
  // the host used to stop before this launch; without the round trip the kernel stops itself
  if (__builtin_amdgcn_readfirstlane(O.flags[0]) & ~8) return;
  TileLds L;
  int32_t* cnt = reinterpret_cast<int32_t*>(smem);               // kBucketInts
  L.bitmap = reinterpret_cast<uint32_t*>(cnt + kBucketInts);
  L.ext = reinterpret_cast<int32_t*>(L.bitmap + ((G.V + 31) >> 5));
  L.hkey = L.ext + kCapExt;
  L.hval = L.hkey + kHash;
  const int32_t* m = meta + (size_t)t * kMetaWords;
  const int n_ext = m[0], e_loc = m[1], n_upd = m[2];
  const int32_t voff = m[21], eoff = m[22], soff = m[23];
  const int32_t vstart = vstart_tab[t], n_own = vend_tab[t] - vstart;
  const int32_t es = estart[t], e_own = estart[t + 1] - es;
  if (tid == 0) { S.fail = m[3]; S.ecnt = 0; }
  if (tid <= kMaxDepth) { S.ring_end[tid] = m[4 + tid]; S.level_end[tid] = 0; }
  if (tid < kCapExt / 64) S.gw[tid] = 1;
  const int bm_words = (G.V + 31) >> 5;
  for (int i = tid; i < bm_words; i += kP2Threads) L.bitmap[i] = 0u;
  __syncthreads();
  if (m[3]) {  // pass 1 already failed this tile: leave a descriptor that says so
    if (tid == 0) { TileDesc D = {}; D.n_ext = -1; O.tiles[t] = D; }
    return;
  }
  for (int l = tid; l < n_ext; l += kP2Threads) {
    const int32_t v = tile_ext[(size_t)t * kCapExt + l];
    L.ext[l] = v;
    atomicOr(&L.bitmap[v >> 5], 1u << (v & 31));
  }
  __syncthreads();
  tile_hash_build<kP2Threads>(L, n_ext);
  const int e_cnt = tile_count_edges<kP2Threads>(G, L, S, cnt, n_ext, s_sc);
  if (e_cnt != e_loc) { if (tid == 0) S.fail = 1; }
  tile_emit<kP2Threads>(G, L, S, cnt, t, vstart, n_own, es, e_own, n_ext, e_loc, n_upd, voff, eoff, soff, O);
}

// Pass 1 + offsets + pass 2 in ONE launch (a frame stream with speculative tile arrays, tiles <=
// kScanMaxBlocks): the rings and the hash are built once instead of twice, the three running totals
// (local vertices, local edges, slot rows) come by look-back over the tiles before this one, packed
// 22 | 22 | 20 bits.  Tiles are dispatched in index order and a tile only waits for lower indices, so
// the grid may exceed the resident set.  tile_ext / meta are written all the same: when the speculative
// arrays turn out too small (bit 128) the host re-runs plain pass 2 from them.
__global__ __launch_bounds__(kP2Threads) void k_tile_fused(TileGraph G, const int32_t* __restrict__ vstart_tab,
                                                           const int32_t* __restrict__ vend_tab,
                                                           const int32_t* __restrict__ estart, int32_t* tile_ext,
                                                           int32_t* meta, TileOut O, ScanState st, int32_t cap_nv,
                                                           int32_t cap_ne, int32_t cap_ns) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  __shared__ TileShared S;
  __shared__ unsigned long long sh_lb[2];
  __shared__ int32_t s_sc[20];
  const int t = blockIdx.x, tid = threadIdx.x, ntiles = gridDim.x;
  // failed before this launch (bits 4, 8, 128 are the launch's own: a tile must not leave on them,
  // the tiles behind it wait for its totals)
  // (it still publishes: a flag raised by the second stream while this launch runs is seen by some
  // tiles and not by others)
  if (__builtin_amdgcn_readfirstlane(O.flags[0]) & ~(4 | 8 | 128)) {
    if (tid == 0) {
      __hip_atomic_store(&st.agg[t], 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __hip_atomic_store(&st.flag[t], st.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    return;
  }
  TileLds L;
  int32_t* cnt = reinterpret_cast<int32_t*>(smem);               // kBucketInts
  L.bitmap = reinterpret_cast<uint32_t*>(cnt + kBucketInts);
  L.ext = reinterpret_cast<int32_t*>(L.bitmap + ((G.V + 31) >> 5));
  L.hkey = L.ext + kCapExt;
  L.hval = L.hkey + kHash;
  if (tid == 0) { S.fail = 0; S.ecnt = 0; }
  if (tid <= kMaxDepth) S.level_end[tid] = 0;
  if (tid < kCapExt / 64) S.gw[tid] = 1;
  __syncthreads();
  const int32_t vstart = vstart_tab[t], n_own = vend_tab[t] - vstart;
  const int32_t es = estart[t], e_own = estart[t + 1] - es;
  TILE_STAMP(O, 1);
  tile_rings<kP2Threads>(G, L, vstart, n_own, &S.n, S.ring_end, &S.fail);
  const int n_ext = S.n;
  TILE_STAMP(O, 2);
  tile_hash_build<kP2Threads>(L, n_ext);
  TILE_STAMP(O, 3);
  const int e_cnt = tile_count_edges<kP2Threads>(G, L, S, cnt, n_ext, s_sc);
  TILE_STAMP(O, 4);
  const int n_upd = G.depth == 0 ? n_ext : S.ring_end[G.depth - 1];
  const bool bad = S.fail || e_cnt > kCapEdge;
  __syncthreads();
  const int e_loc = min(e_cnt, kSortPad);
  const unsigned long long mine = (unsigned long long)n_ext | ((unsigned long long)e_loc << 22) | ((unsigned long long)n_upd << 44);
  const unsigned long long before = scan_lookback<unsigned long long>(mine, st, sh_lb);
  TILE_STAMP(O, 5);
  const int32_t voff = (int32_t)(before & 0x3fffffu), eoff = (int32_t)((before >> 22) & 0x3fffffu),
                soff = (int32_t)(before >> 44);
  int32_t* m = meta + (size_t)t * kMetaWords;
  for (int l = tid; l < n_ext; l += kP2Threads) tile_ext[(size_t)t * kCapExt + l] = L.ext[l];
  if (tid <= kMaxDepth) m[4 + tid] = S.ring_end[tid];
  if (tid == 0) {
    m[0] = n_ext; m[1] = e_cnt; m[2] = n_upd; m[3] = bad ? 1 : 0;
    m[21] = voff; m[22] = eoff; m[23] = soff;
    if (bad) atomicOr(&O.flags[0], 4);
    if (t == ntiles - 1) { O.flags[1] = voff + n_ext; O.flags[2] = eoff + e_loc; O.flags[3] = soff + n_upd; }
  }
  const bool over = voff + n_ext > cap_nv || eoff + e_loc > cap_ne || soff + n_upd > cap_ns;
  if (over && tid == 0) atomicOr(&O.flags[0], 128);
  if (bad || over) {
    if (tid == 0 && bad && !over) { TileDesc D = {}; D.n_ext = -1; O.tiles[t] = D; }
    return;
  }
  tile_emit<kP2Threads>(G, L, S, cnt, t, vstart, n_own, es, e_own, n_ext, e_loc, n_upd, voff, eoff, soff, O);
}

// What the host reads after a build -- the flags word, the caller's check word, the tile descriptors --
// written straight into page-locked host memory by one launch (three small D2H copies were three
// dependent blit launches, ~20 us of the frame's critical path)
__global__ __launch_bounds__(256) void k_publish(const int32_t* __restrict__ flags, const int32_t* __restrict__ user,
                                                 const int32_t* __restrict__ tiles, int32_t tile_words, int32_t* hflags,
                                                 int32_t* huser, int32_t* htiles) {
  const int32_t i = blockIdx.x * 256 + threadIdx.x;
  if (i < 8) hflags[i] = flags[i];
  if (user && i < 4) huser[i] = user[i];
  if (tiles && i < tile_words) htiles[i] = tiles[i];
}

// ------------------------------------------------------------------------------------------
// Cost weights: from the tiles of the previous pass, or from the cost-density grid of the
// previous frame (plan.cpp: tile_weight(), Plan::wgrid).  All integer.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ int32_t tile_weight_dev(const TileDesc& D) {
  const long long cost = tile_cost(D);
  const long long w = cost * 1024 / max(D.n_own, 1);
  return (int32_t)max(1ll, w);
}

__global__ __launch_bounds__(256) void k_weights_from_tiles(int32_t V, const int32_t* __restrict__ v_i2o,
                                                            const int32_t* __restrict__ tile_of_int,
                                                            const TileDesc* __restrict__ tiles, int32_t* w_int) {
  const int32_t k = blockIdx.x * 256 + threadIdx.x;
  if (k < V) w_int[v_i2o[k]] = tile_weight_dev(tiles[tile_of_int[k]]);
}

// refinement pass: w_v <- w_v * cost(tile of v) * ntiles / total cost (plan.cpp, "refine weights")
__global__ __launch_bounds__(256) void k_weights_scale(int32_t V, const int32_t* __restrict__ v_i2o,
                                                       const int32_t* __restrict__ tile_of_int,
                                                       const TileDesc* __restrict__ tiles, int32_t ntiles,
                                                       long long total, int32_t* w_int) {
  const int32_t k = blockIdx.x * 256 + threadIdx.x;
  if (k >= V) return;
  const TileDesc& D = tiles[tile_of_int[k]];
  const long long cost = tile_cost(D);
  const int32_t v = v_i2o[k];
  const long long w = (long long)w_int[v] * cost * ntiles / max(total, 1ll);
  w_int[v] = (int32_t)min(1ll << 28, max(1ll, w));
}

__device__ __forceinline__ int grid_cell_dev(const float* b, float2 q) {
  const float fx = (q.x - b[0]) / fmaxf(b[2] - b[0], 1e-20f);
  const float fy = (q.y - b[1]) / fmaxf(b[3] - b[1], 1e-20f);
  const int cx = max(0, min(Plan::kGrid - 1, (int)(fx * Plan::kGrid)));
  const int cy = max(0, min(Plan::kGrid - 1, (int)(fy * Plan::kGrid)));
  return cy * Plan::kGrid + cx;
}

// ------------------------------------------------------------------------------------------
// Partition REUSE on a frame stream.  The solver's results do not depend on the partition (every
// path produces the oracle's bits), the partition only has to be spatially compact and balanced --
// and consecutive frames of a camera see almost the same feature distribution.  After every build the
// tile of each vertex is rasterised into a pyramid of spatial cells (256^2 ... 8^2, 1 cells over the
// frame's bounding box; value = tile + 1, the larger id wins a shared cell: integer atomicMax on the
// three finest levels, order-free; the coarser ones are reduced from those).  The NEXT frame looks its vertices up in that pyramid (finest non-empty
// level), counts the tiles, and a stable counting pass groups the vertices by tile: three short
// kernels instead of two device sorts, the level loop and the subtree kernel (~0.3 ms at 50 k).
// Inside a tile the Morton order of stage B applies as ever.  A frame whose tiles come out empty or
// much larger than planned (scene change) is rebuilt by exact bisection.
// ------------------------------------------------------------------------------------------
constexpr int kCntStride = 32;  // ints between two tile counters of the reuse pass (one cache line each)
constexpr int kPyrLevels = 7;
constexpr int kPyrAtomicLevels = 3;  // written per vertex (>= 4096 cells: no contended atomics); the coarser
                                     // levels are reduced from level 2 by k_grid_final
__device__ constexpr int kPyrDims[kPyrLevels] = {256, 128, 64, 32, 16, 8, 1};
__device__ constexpr int kPyrOff[kPyrLevels] = {0, 65536, 65536 + 16384, 65536 + 16384 + 4096, 65536 + 16384 + 4096 + 1024,
                                                65536 + 16384 + 4096 + 1024 + 256, 65536 + 16384 + 4096 + 1024 + 256 + 64};
constexpr int kPyrCells = 65536 + 16384 + 4096 + 1024 + 256 + 64 + 1;
constexpr int kPyrAtomicCells = 65536 + 16384 + 4096;

__device__ __forceinline__ int pyr_cell(const float* b, float2 q, int level) {
  const int d = kPyrDims[level];
  const float fx = (q.x - b[0]) / fmaxf(b[2] - b[0], 1e-20f);
  const float fy = (q.y - b[1]) / fmaxf(b[3] - b[1], 1e-20f);
  const int cx = max(0, min(d - 1, (int)(fx * (float)d)));
  const int cy = max(0, min(d - 1, (int)(fy * (float)d)));
  return kPyrOff[level] + cy * d + cx;
}

// tile of every vertex of the NEW frame from the previous frame's pyramid; tile counts
__global__ __launch_bounds__(256) void k_reuse_assign(int32_t V, int32_t ntiles, const float2* __restrict__ pos,
                                                      const float* __restrict__ bounds,
                                                      const int32_t* __restrict__ pyr, int32_t* vt, int32_t* vrank,
                                                      int32_t* tile_cnt) {
  const int32_t v = blockIdx.x * 256 + threadIdx.x;
  if (v >= V) return;
  const float2 q = pos[v];
  int32_t t = 0;
  for (int l = 0; l < kPyrLevels; ++l) {
    const int32_t c = pyr[pyr_cell(bounds, q, l)];
    if (c > 0) { t = c - 1; break; }
  }
  t = max(0, min(t, ntiles - 1));
  // the count doubles as the vertex's rank inside its tile (any order will do: stage B sorts the tile),
  // so the scatter needs no second round of contended atomics.  One counter per 128-byte line: a
  // few hundred counters packed into 8 lines serialise every atomic of the launch on 8 L2 channels
  vt[v] = t;
  vrank[v] = atomicAdd(&tile_cnt[t * kCntStride], 1);
}

// Tile ranges from the counts + the counting scatter, one launch: every block scans the (<= 1024) tile
// counts itself -- a few hundred LDS operations instead of a dependent single-block launch -- and
// places its 256 vertices; block 0 also writes the segment table the later stages read (what the last
// bisection level leaves behind: one leaf per segment, in tile order).  A tile that is empty or above
// `cap` vertices rejects the reuse (flag 64).
__global__ __launch_bounds__(256) void k_reuse_scatter(int32_t V, int32_t ntiles, int32_t cap,
                                                       const int32_t* __restrict__ vt,
                                                       const int32_t* __restrict__ vrank,
                                                       const int32_t* __restrict__ tile_cnt, SegTab t, int32_t* nseg,
                                                       int32_t* flags, int32_t* perm, int32_t* seg_pos) {
  __shared__ int32_t s_lo[kSegCap];
  __shared__ int32_t s_part[16];
  static_assert(kSegCap == 4 * 256, "four counters per thread");
  const int tid = threadIdx.x;
  int32_t c[4], sum = 0;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int i = 4 * tid + k;
    c[k] = i < ntiles ? tile_cnt[i * kCntStride] : 0;
    sum += c[k];
  }
  int32_t run = block_exclusive<int32_t>(sum, s_part);
#pragma unroll
  for (int k = 0; k < 4; ++k) { s_lo[4 * tid + k] = run; run += c[k]; }
  if (blockIdx.x == 0) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int i = 4 * tid + k;
      if (i < ntiles) {
        const int32_t lo = s_lo[i];
        t.lo[i] = lo; t.hi[i] = lo + c[k]; t.leaves[i] = 1; t.first[i] = i;
        if (c[k] < 1 || c[k] > cap) atomicOr(&flags[0], 64);
      }
    }
    if (tid == 0) nseg[0] = ntiles;
    if (tid == 255 && run != V) atomicOr(&flags[0], 64);
  }
  __syncthreads();
  const int32_t v = blockIdx.x * 256 + tid;
  if (v >= V) return;
  const int32_t tl = vt[v];
  const int32_t p = s_lo[tl] + vrank[v];  // (order inside a tile is set by stage B)
  perm[p] = v;
  seg_pos[p] = tl;
}

__global__ __launch_bounds__(256) void k_weights_from_grid(int32_t V, const float2* __restrict__ pos,
                                                           const float* __restrict__ bounds,
                                                           const int32_t* __restrict__ grid_w, int32_t* w_int) {
  const int32_t v = blockIdx.x * 256 + threadIdx.x;
  if (v < V) w_int[v] = grid_w[grid_cell_dev(bounds, pos[v])];
}

__global__ __launch_bounds__(256) void k_grid_accum(int32_t V, const float2* __restrict__ pos,
                                                    const int32_t* __restrict__ v_i2o,
                                                    const int32_t* __restrict__ tile_of_int,
                                                    const TileDesc* __restrict__ tiles,
                                                    const float* __restrict__ bounds,
                                                    unsigned long long* sum, int32_t* cnt, int32_t* pyr) {
  // vertices that are neighbours in internal order (same tile, Morton order) fall into the same
  // cell: accumulate per workgroup in LDS, flush the touched cells once (integer sums: any order)
  __shared__ unsigned long long s_sum[Plan::kGrid * Plan::kGrid];
  __shared__ int32_t s_cnt[Plan::kGrid * Plan::kGrid];
  for (int c = threadIdx.x; c < Plan::kGrid * Plan::kGrid; c += 256) { s_sum[c] = 0ull; s_cnt[c] = 0; }
  __syncthreads();
  const int32_t k = blockIdx.x * 256 + threadIdx.x;
  if (k < V) {
    const float2 q = pos[v_i2o[k]];
    const int c = grid_cell_dev(bounds, q);
    atomicAdd(&s_sum[c], (unsigned long long)tile_weight_dev(tiles[tile_of_int[k]]));
    atomicAdd(&s_cnt[c], 1);
    if (pyr) {  // the tile map the next frame's partition is read from (partition reuse)
      const int32_t tv = tile_of_int[k] + 1;
#pragma unroll
      for (int l = 0; l < kPyrAtomicLevels; ++l) atomicMax(&pyr[pyr_cell(bounds, q, l)], tv);
    }
  }
  __syncthreads();
  for (int c = threadIdx.x; c < Plan::kGrid * Plan::kGrid; c += 256)
    if (s_cnt[c]) { atomicAdd(&sum[c], s_sum[c]); atomicAdd(&cnt[c], s_cnt[c]); }
}

__global__ __launch_bounds__(1024) void k_grid_final(int32_t V, const unsigned long long* sum,
                                                     const int32_t* cnt, int32_t* grid_w, const float* gbbox,
                                                     float* bounds, int32_t* pyr) {
  if (threadIdx.x < 4) bounds[threadIdx.x] = gbbox[threadIdx.x];  // the frame the grid was made from
  if (pyr) {  // coarse levels of the tile map: max over the 2 x 2 (last step 8 x 8) children, level by level
    for (int l = kPyrAtomicLevels; l < kPyrLevels; ++l) {
      const int d = kPyrDims[l], dc = kPyrDims[l - 1], r = dc / d;
      for (int c = threadIdx.x; c < d * d; c += 1024) {
        const int cx = c % d, cy = c / d;
        int32_t m = 0;
        for (int yy = 0; yy < r; ++yy)
          for (int xx = 0; xx < r; ++xx) m = max(m, pyr[kPyrOff[l - 1] + (cy * r + yy) * dc + cx * r + xx]);
        pyr[kPyrOff[l] + c] = m;
      }
      __syncthreads();
    }
  }
  __shared__ unsigned long long s_tot[1024];
  const int c = threadIdx.x;  // kGrid * kGrid == 1024
  s_tot[c] = sum[c];
  __syncthreads();
  for (int off = 512; off > 0; off >>= 1) {
    if (c < off) s_tot[c] += s_tot[c + off];
    __syncthreads();
  }
  const long long total = (long long)s_tot[0];
  const int32_t mean = V > 0 ? (int32_t)max(1ll, total / V) : 1024;
  grid_w[c] = cnt[c] > 0 ? (int32_t)((long long)sum[c] / cnt[c]) : mean;
}


// ------------------------------------------------------------------------------------------
// Graph sync on the device (row a7 / f3): the unique undirected edges of a triangulation, i < j, in
// lexicographic order, with alpha = 1 / |pos_i - pos_j| (statement: oracle nltgv2_graph_sync).
// ------------------------------------------------------------------------------------------
// Edges of a triangulation by counting: every triangle side (a, b) goes to the row of min(a, b)
// holding max(a, b); a row is sorted by its own thread, which also marks the first of every run of
// equal entries (an interior edge appears twice); after a scan of the marks the same thread writes
// its row's edges.  Result: the unique sides in ascending (min, max) order -- what the sort of the
// 3T 64-bit keys gave, in 9 short launches instead of ~24.
__device__ __forceinline__ bool half_edge(int32_t k, int32_t V, const int32_t* __restrict__ tris, int32_t* a, int32_t* b) {
  const int32_t t = k / 3, c = k - 3 * t;
  const int32_t u = tris[3 * t + c], w = tris[3 * t + (c == 2 ? 0 : c + 1)];
  *a = min(u, w); *b = max(u, w);
  return !(u < 0 || w < 0 || u >= V || w >= V || u == w);
}

__global__ __launch_bounds__(256) void k_he_count(int32_t n3, int32_t V, const int32_t* __restrict__ tris, int32_t* cnt,
                                                  int32_t* rank, int32_t* flags) {
  const int32_t k = blockIdx.x * 256 + threadIdx.x;
  if (k >= n3) return;
  int32_t a, b;
  if (!half_edge(k, V, tris, &a, &b)) { atomicOr(&flags[0], 2); return; }
  rank[k] = atomicAdd(&cnt[a], 1);
}

__global__ __launch_bounds__(256) void k_he_fill(int32_t n3, int32_t V, const int32_t* __restrict__ tris,
                                                 const int32_t* __restrict__ off,
                                                 const int32_t* __restrict__ rank, uint32_t* out) {
  const int32_t k = blockIdx.x * 256 + threadIdx.x;
  if (k >= n3) return;
  int32_t a, b;
  if (!half_edge(k, V, tris, &a, &b)) return;
  out[off[a] + rank[k]] = (uint32_t)b;
}

__global__ __launch_bounds__(256) void k_he_mark(int32_t V, const int32_t* __restrict__ off,
                                                 const uint32_t* __restrict__ out, int32_t* f) {
  const int32_t v = blockIdx.x * 256 + threadIdx.x;
  if (v >= V) return;
  for (int32_t k = off[v]; k < off[v + 1]; ++k) f[k] = (k == off[v] || out[k] != out[k - 1]) ? 1 : 0;
}

__global__ __launch_bounds__(256) void k_he_compact(int32_t V, const int32_t* __restrict__ off,
                                                    const uint32_t* __restrict__ out, const int32_t* __restrict__ f,
                                                    const int32_t* __restrict__ idx, const float2* __restrict__ pos,
                                                    int2* edges, float* alpha, int32_t* total, int32_t* nan_flag) {
  const int32_t v = blockIdx.x * 256 + threadIdx.x;
  if (v >= V) return;
  if (v == V - 1) { const int32_t n = off[V]; total[0] = n > 0 ? idx[n - 1] + f[n - 1] : 0; }
  const float2 pi = pos[v];
  for (int32_t k = off[v]; k < off[v + 1]; ++k) {
    if (!f[k]) continue;
    const int32_t j = (int32_t)out[k];
    const float2 pj = pos[j];
    const float dx = pi.x - pj.x, dy = pi.y - pj.y;
    edges[idx[k]] = make_int2(v, j);
    const float a = 1.0f / sqrtf(dx * dx + dy * dy);  // -ffp-contract=off: two roundings, as the oracle
    alpha[idx[k]] = a;
    if (nan_flag && !isfinite(a)) atomicOr(nan_flag, 1);  // (two features on one pixel)
  }
}

// rows + mark + scan + compact in ONE launch (graphs of up to kScanMaxBlocks * 256 vertices): a block
// sorts its 256 rows in LDS, counts the distinct entries, gets the number of edges before it by
// look-back and writes its edges -- the sorted rows never travel back to memory
__global__ __launch_bounds__(256) void k_he_unique(int32_t V, const int32_t* __restrict__ off, uint32_t* out,
                                                   const float2* __restrict__ pos, int2* edges, float* alpha,
                                                   int32_t* total, int32_t* nan_flag, ScanState st) {
  __shared__ uint32_t s_a[kRowsLds];
  __shared__ int32_t sh_a[16];
  __shared__ int32_t sh_p[2];
  const int32_t v0 = blockIdx.x * 256, v1 = min(v0 + 256, V);
  const int32_t r0 = off[v0], r1 = off[v1];
  const int32_t v = v0 + threadIdx.x;
  const bool lds = r1 - r0 <= kRowsLds;
  if (lds) {
    for (int32_t i = threadIdx.x; i < r1 - r0; i += 256) s_a[i] = out[r0 + i];
    __syncthreads();
  }
  int d = 0, uniq = 0;
  uint32_t* a = nullptr;
  if (v < V) {
    d = off[v + 1] - off[v];
    a = lds ? s_a + (off[v] - r0) : out + off[v];
    sort_row(a, d);
    for (int k = 0; k < d; ++k) uniq += (k == 0 || a[k] != a[k - 1]) ? 1 : 0;
  }
  const int32_t ex = block_exclusive<int32_t>(uniq, sh_a);
  if (threadIdx.x == 255) sh_p[1] = ex + uniq;
  __syncthreads();
  const int32_t before = scan_lookback<int32_t>(sh_p[1], st, sh_p);
  if (v >= V) return;
  int32_t idx = before + ex;
  const float2 pi = pos[v];
  for (int k = 0; k < d; ++k) {
    if (k > 0 && a[k] == a[k - 1]) continue;
    const int32_t j = (int32_t)a[k];
    const float2 pj = pos[j];
    const float dx = pi.x - pj.x, dy = pi.y - pj.y;
    edges[idx] = make_int2(v, j);
    const float al = 1.0f / sqrtf(dx * dx + dy * dy);  // -ffp-contract=off: two roundings, as the oracle
    alpha[idx] = al;
    if (nan_flag && !isfinite(al)) atomicOr(nan_flag, 1);  // (two features on one pixel)
    ++idx;
  }
  if (v == V - 1) total[0] = idx;
}

__global__ __launch_bounds__(256) void k_sync_data(int32_t V, const float* __restrict__ mu,
                                                   const float* __restrict__ var, const float* __restrict__ pred,
                                                   float scale, int adaptive, int init_pred, float* z, float* wgt,
                                                   float* x0, int32_t* nan_flag) {
  const int32_t v = blockIdx.x * 256 + threadIdx.x;
  if (v >= V) return;
  const float zi = mu[v] / scale;
  const float wi = adaptive ? 1.0f / var[v] : 1.0f;
  const float xi = (init_pred && pred && isfinite(pred[v])) ? pred[v] / scale : zi;
  z[v] = zi;
  wgt[v] = wi;
  x0[v] = xi;
  // the non-finite-input check of the upload rides here (it was three more launches per frame)
  if (nan_flag && !(isfinite(zi) && isfinite(wi) && isfinite(xi))) atomicOr(nan_flag, 1);
}

// ------------------------------------------------------------------------------------------
// SMALL frames (<= kMiniV vertices, reused partition, predicted edge count): everything in front of
// the tile pass -- edges of the triangulation, data terms, partition from the previous frame's tile
// map, vertex order, triangle CSR, edge order, incidence CSR -- in ONE launch of ONE workgroup.
// At 1.2 k vertices those stages were 24 dependent launches whose kernels run 2-5 us each: ~0.15 ms
// of a 0.55 ms frame was the dependency latency between them.  Same rules, same arrays as the
// stages above (k_he_*, k_sync_data, k_reuse_*, k_tile_order, k_tri_*, k_edge_*, k_csr_*): the phases
// below are those kernels' bodies as loops over one workgroup, counters and offsets in LDS,
// __syncthreads() between them.  tests/test_gpu_plan_device.py compares the two paths array for array.
// ------------------------------------------------------------------------------------------
constexpr int kMiniV = 2048;          // vertices
constexpr int kMiniT = 4096;          // triangles (3T half edges <= 12288)
constexpr int kMiniE = 6144;          // edges
constexpr int kMiniThreads = 1024;
constexpr int kMiniTileCap = 128;     // own vertices of one tile (two keys per lane of the wave that orders it)
constexpr int kMiniRows = 2 * kMiniE; // entries of the largest counting pass (3T = 2E = 12288)
constexpr size_t kMiniLds = sizeof(uint32_t) * kMiniRows + sizeof(int32_t) * (2 * kMiniV + kMiniE) + sizeof(float2) * kMiniV;
static_assert(3 * kMiniT <= kMiniRows, "half-edge and triangle rows fit the row buffer");
static_assert(kMiniV == 2048 && kMiniT == 4096 && kMiniE == 6144, "DevPlanner::mini_eligible (plan_dev.h) states these limits");

struct MiniArgs {
  int32_t V, T, E_expect, ntiles, cap;
  // inputs (device copies of the caller's arrays)
  const int32_t* tris; const float2* pos; const float* mu; const float* var; const float* pred;
  float2* pos_out;               // the builder's own input arrays (the launch's inputs may sit in a staging arena;
  int32_t* tris_out;             //  a retry that bisects reads these)
  float scale; int32_t adaptive, init_pred; float dsign;
  // graph sync outputs
  int2* edges; float* alpha; float* z; float* wgt; float* x0;
  int32_t* dflags;               // caller's words: [0] non-finite, [1] derived edge count, [2] bit 2 bad index
  // scratch in global memory (L1 / L2 resident at these sizes)
  int32_t* rank;                 // max(3T, 2E): ranks of the counting passes
  int32_t* vt;                   // V
  // partition
  const float* gbbox; const int32_t* pyr; SegTab tab; int32_t* nseg; int32_t* flags;
  int32_t* seg_pos;
  // plan arrays
  int32_t* v_i2o; int32_t* v_o2i; int32_t* tile_of_int;
  int32_t* tris_int; int32_t* trow; uint32_t* tinc;
  int32_t* e_i2o; int32_t* e_o2i; int2* eij; float4* ew; int32_t* estart;
  int32_t* grow; uint32_t* ginc; int32_t* gadj; int32_t* ipos;
  long long* prof;               // dev aid (option "plan_timing" = 4): 20 phase stamps, or null
};

// exclusive scan of a[0 .. n) in place (LDS), 1024 threads, n <= 5 * 1024; returns the total.  Every
// thread owns `per` consecutive entries: one block scan (three barriers) whatever n is.
__device__ __forceinline__ int32_t mini_scan(int32_t* a, int n, int32_t* sh /* >= 17 */) {
  const int tid = threadIdx.x;
  const int per = (n + kMiniThreads - 1) / kMiniThreads;
  const int i0 = tid * per;
  int32_t v[5], sum = 0;
#pragma unroll
  for (int k = 0; k < 5; ++k) {
    v[k] = (k < per && i0 + k < n) ? a[i0 + k] : 0;
    sum += v[k];
  }
  int32_t run = block_exclusive<int32_t>(sum, sh);
  if (tid == kMiniThreads - 1) sh[16] = run + sum;
#pragma unroll
  for (int k = 0; k < 5; ++k) {
    if (k < per && i0 + k < n) a[i0 + k] = run;
    run += v[k];
  }
  __syncthreads();
  return sh[16];
}

// A row of <= N entries sorted through registers: every entry's place is the number of entries in front
// of it (ties by position), N^2 register compares and no dependent LDS round trips (the insertion
// sort of sort_row() on an LDS row is ~200 cycles per move: 25 k ticks for the half-edge rows of a
// 1.2 k-vertex frame, this: 4 k).  *dups (optional) = entries that repeat an earlier one.
template <int N>
__device__ __forceinline__ void sort_row_reg(uint32_t* row, int d, int* dups) {
  uint32_t r[N];
#pragma unroll
  for (int i = 0; i < N; ++i) r[i] = i < d ? row[i] : 0xffffffffu;
  int nd = 0;
#pragma unroll
  for (int i = 0; i < N; ++i) {
    int rank = 0, dup = 0;
#pragma unroll
    for (int j = 0; j < N; ++j) {
      if (j == i) continue;
      const bool eq_before = r[j] == r[i] && j < i;
      rank += (r[j] < r[i] || eq_before) ? 1 : 0;
      dup |= eq_before ? 1 : 0;
    }
    if (i < d) { row[rank] = r[i]; nd += dup; }
  }
  if (dups) *dups = nd;
}

// sorts row[0 .. d) ascending; returns the number of distinct entries
__device__ __forceinline__ int mini_sort_row(uint32_t* row, int d) {
  int nd = 0;
  if (d <= 8) { sort_row_reg<8>(row, d, &nd); return d - nd; }
  if (d <= 16) { sort_row_reg<16>(row, d, &nd); return d - nd; }
  sort_row(row, d);
  int uq = 0;
  for (int k = 0; k < d; ++k) uq += (k == 0 || row[k] != row[k - 1]) ? 1 : 0;
  return uq;
}

// dev aid: phase boundaries of the launch (shader-clock ticks) into a.prof when it is set
#define MINI_STAMP(n) do { if (a.prof && tid == 0) a.prof[n] = (long long)__builtin_amdgcn_s_memtime(); } while (0)
// k_mini_plan gives up early in two places (the predicted edge count was wrong; the reused partition does not suit the
// frame).  The host only learns that at the builder's synchronisation, AFTER the stages behind this launch (initial
// state, tile passes) have run on whatever the plan arrays hold -- the previous frame's tables, or, after the arrays
// grew, nothing at all.  So an early exit leaves a VALID plan behind: every tile empty, no incidences, no
// triangles, an identity vertex order (unless the partition phase already wrote a permutation).
__device__ __forceinline__ void mini_leave_empty_plan(const MiniArgs& a, bool keep_perm) {
  const int tid = threadIdx.x, V = a.V, ntiles = a.ntiles;
  for (int v = tid; v < V; v += kMiniThreads) {
    if (!keep_perm) a.v_i2o[v] = v;
    a.v_o2i[v] = keep_perm ? 0 : v;  // (only ever an index into [0, V))
    a.tile_of_int[v] = 0;
    a.grow[v] = 0;
    a.trow[v] = 0;
  }
  if (keep_perm) {
    __syncthreads();
    for (int p = tid; p < V; p += kMiniThreads) a.v_o2i[a.v_i2o[p]] = p;
  }
  for (int t = tid; t <= ntiles; t += kMiniThreads) {
    a.estart[t] = 0;
    if (t < ntiles) { a.tab.lo[t] = 0; a.tab.hi[t] = 0; a.tab.leaves[t] = 1; a.tab.first[t] = t; }
  }
  if (tid == 0) { a.grow[V] = 0; a.trow[V] = 0; a.nseg[0] = ntiles; }
}

__global__ __launch_bounds__(kMiniThreads) void k_mini_plan(MiniArgs a) {
  __shared__ int32_t s_a[2 * kMiniV + 4];         // he offsets | triangle rows | edge buckets (2V + 1) | degrees
  __shared__ int32_t s_b[kMiniV + 4];             // distinct entries per half-edge row -> first edge id of the row
  __shared__ int32_t s_tl[kSegCap + 1];           // tile counts -> tile starts
  __shared__ uint64_t s_key[16][kMiniTileCap];    // one tile's (Morton code, id) keys per wave
  __shared__ int32_t s_sc[20];
  // rows of the counting passes (half edges, triangle rows, edge buckets, incidence rows) are filled and
  // sorted HERE and written out once, coalesced: a thread that sorts its row in global memory pays a
  // round trip per element (the first version of this kernel: 125 us at 1.2 k vertices, 80 of them there)
  extern __shared__ __attribute__((aligned(16))) uint32_t s_buf[];  // kMiniRows entries, then:
  // what the later phases gather from, on chip (a dependent gather from global memory is a 1-2 us round
  // trip for the one workgroup; the phases chain three or four of them)
  int32_t* s_vo2i = reinterpret_cast<int32_t*>(s_buf + kMiniRows);    // V: original -> internal vertex id
  int32_t* s_tile = s_vo2i + kMiniV;                                  // V: tile of an internal vertex id
  int32_t* s_eo2i = s_tile + kMiniV;                                  // E: original -> internal edge id
  float2* s_pos = reinterpret_cast<float2*>(s_eo2i + kMiniE);          // V
  const int tid = threadIdx.x, NT = kMiniThreads;
  const int32_t V = a.V, ntiles = a.ntiles;
  for (int v = tid; v < V; v += NT) {
    const float2 q = a.pos[v];
    s_pos[v] = q;
    if (a.pos_out != a.pos) a.pos_out[v] = q;
  }
  MINI_STAMP(1);
  // ---- flags of the build, counters ----
  if (tid < 8) a.flags[tid] = 0;
  for (int i = tid; i <= V; i += NT) { s_a[i] = 0; s_b[i] = 0; }
  for (int i = tid; i <= ntiles; i += NT) s_tl[i] = 0;
  __syncthreads();
  MINI_STAMP(2);
  // ---- edges of the triangulation (k_he_count / fill / unique) ----
  // (a thread keeps its <= 4 triangles -- corner ids and the ranks its counting atomics returned -- in
  // registers from here to the triangle CSR: fixed trip counts, so that the loads of a phase leave
  // back to back instead of one round trip per loop iteration)
  constexpr int kTri = kMiniT / kMiniThreads;  // 4
  int32_t tv[kTri][3], tr[kTri][3];
#pragma unroll
  for (int i = 0; i < kTri; ++i) {
    const int32_t t = tid + i * NT;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      tv[i][c] = t < a.T ? a.tris[3 * t + c] : -1;
      if (t < a.T && a.tris_out != a.tris) a.tris_out[3 * t + c] = tv[i][c];
    }
  }
  if (a.prof) { __syncthreads(); asm volatile("" :: "v"(tv[0][0])); MINI_STAMP(3); }
#pragma unroll
  for (int i = 0; i < kTri; ++i) {
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const int32_t u = tv[i][c], w = tv[i][c == 2 ? 0 : c + 1];
      tr[i][c] = -1;
      if (tid + i * NT >= a.T) continue;
      if (u < 0 || w < 0 || u >= V || w >= V || u == w) { atomicOr(&a.dflags[2], 2); continue; }  // = half_edge()
      tr[i][c] = atomicAdd(&s_a[min(u, w)], 1);
    }
  }
  __syncthreads();
  MINI_STAMP(4);
  mini_scan(s_a, V + 1, s_sc);
  MINI_STAMP(5);
#pragma unroll
  for (int i = 0; i < kTri; ++i) {
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const int32_t u = tv[i][c], w = tv[i][c == 2 ? 0 : c + 1];
      if (tr[i][c] >= 0) s_buf[s_a[min(u, w)] + tr[i][c]] = (uint32_t)max(u, w);
    }
  }
  __syncthreads();
  MINI_STAMP(6);
  for (int v = tid; v < V; v += NT) {
    uint32_t* row = s_buf + s_a[v];
    const int d = s_a[v + 1] - s_a[v];
    s_b[v] = mini_sort_row(row, d);
  }
  __syncthreads();
  MINI_STAMP(7);
  const int32_t E = mini_scan(s_b, V + 1, s_sc);
  MINI_STAMP(8);
  if (tid == 0) a.dflags[1] = E;
  if (E != a.E_expect || E > kMiniE) {  // (uniform) the host's prediction was wrong: it builds again
    mini_leave_empty_plan(a, false);
    return;
  }
  for (int v = tid; v < V; v += NT) {
    const uint32_t* row = s_buf + s_a[v];
    const int d = s_a[v + 1] - s_a[v];
    int32_t idx = s_b[v];
    const float2 pi = s_pos[v];
    for (int k = 0; k < d; ++k) {
      if (k > 0 && row[k] == row[k - 1]) continue;
      const int32_t j = (int32_t)row[k];
      const float2 pj = s_pos[j];
      const float dx = pi.x - pj.x, dy = pi.y - pj.y;
      a.edges[idx] = make_int2(v, j);
      const float al = 1.0f / sqrtf(dx * dx + dy * dy);  // -ffp-contract=off: two roundings, as the oracle
      a.alpha[idx] = al;
      if (!isfinite(al)) atomicOr(&a.dflags[0], 1);
      ++idx;
    }
  }
  MINI_STAMP(9);
  // ---- data terms (k_sync_data) ----
  for (int v = tid; v < V; v += NT) {
    const float zi = a.mu[v] / a.scale;
    const float wi = a.adaptive ? 1.0f / a.var[v] : 1.0f;
    const float xi = (a.init_pred && a.pred && isfinite(a.pred[v])) ? a.pred[v] / a.scale : zi;
    a.z[v] = zi; a.wgt[v] = wi; a.x0[v] = xi;
    if (!(isfinite(zi) && isfinite(wi) && isfinite(xi))) atomicOr(&a.dflags[0], 1);
  }
  MINI_STAMP(10);
  // ---- partition from the previous frame's tile map (k_reuse_assign / scatter) ----
  for (int v = tid; v < V; v += NT) {
    const float2 q = s_pos[v];
    int32_t t = 0;
    for (int l = 0; l < kPyrLevels; ++l) {
      const int32_t c = a.pyr[pyr_cell(a.gbbox, q, l)];
      if (c > 0) { t = c - 1; break; }
    }
    t = max(0, min(t, ntiles - 1));
    a.vt[v] = t;
    a.rank[v] = atomicAdd(&s_tl[t], 1);  // (the half-edge ranks are spent)
  }
  __syncthreads();
  {
    const int32_t c0 = tid < ntiles ? s_tl[tid] : 0;  // (ntiles <= kSegCap = 1024 = NT)
    __syncthreads();
    const int32_t total = mini_scan(s_tl, ntiles + 1, s_sc);
    if (tid < ntiles) {
      const int32_t lo = s_tl[tid];
      a.tab.lo[tid] = lo; a.tab.hi[tid] = lo + c0; a.tab.leaves[tid] = 1; a.tab.first[tid] = tid;
      if (c0 < 1 || c0 > a.cap || c0 > kMiniTileCap) atomicOr(&a.flags[0], 64);
    }
    if (tid == 0) { a.nseg[0] = ntiles; if (total != V) atomicOr(&a.flags[0], 64); }
  }
  for (int v = tid; v < V; v += NT) {
    const int32_t t = a.vt[v];
    const int32_t p = s_tl[t] + a.rank[v];
    a.v_i2o[p] = v;  // (= perm; ordered inside the tile below)
    a.seg_pos[p] = t;
  }
  __syncthreads();
  if (__builtin_amdgcn_readfirstlane(a.flags[0]) & 64) {  // (uniform after the barrier) rejected: bisection next
    mini_leave_empty_plan(a, true);
    return;
  }
  MINI_STAMP(11);
  // ---- order inside tiles: (Morton code, id), one wave per tile, rank by counting (k_tile_order) ----
  {
    const int wave = tid >> 6, lane = tid & 63;
    const float mnx = a.gbbox[0], mny = a.gbbox[1], mxx = a.gbbox[2], mxy = a.gbbox[3];
    uint64_t* keys = s_key[wave];
    for (int t = wave; t < ntiles; t += NT / 64) {
      const int32_t lo = s_tl[t], n = s_tl[t + 1] - lo;
      uint64_t mine[2];
      for (int h = 0; h < 2; ++h) {
        const int p = lane + 64 * h;
        uint64_t k = ~0ull;
        if (p < n) {
          const int32_t v = a.v_i2o[lo + p];
          const float2 q = s_pos[v];
          const uint32_t qx = (uint32_t)(65535.0f * (fminf(fmaxf(q.x, mnx), mxx) - mnx) / fmaxf(mxx - mnx, 1e-20f));
          const uint32_t qy = (uint32_t)(65535.0f * (fminf(fmaxf(q.y, mny), mxy) - mny) / fmaxf(mxy - mny, 1e-20f));
          k = ((uint64_t)(spread16(qx) | (spread16(qy) << 1)) << 32) | (uint32_t)v;
        }
        mine[h] = k;
        keys[p] = k;
      }
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0): the wave's keys are in LDS
      int r[2] = {0, 0};
      for (int j = 0; j < n; ++j) {
        const uint64_t kj = keys[j];
        r[0] += kj < mine[0] ? 1 : 0;
        r[1] += kj < mine[1] ? 1 : 0;
      }
      __builtin_amdgcn_wave_barrier();
      for (int h = 0; h < 2; ++h) {
        const int p = lane + 64 * h;
        if (p < n) {
          const int32_t v = (int32_t)(uint32_t)mine[h];
          a.v_o2i[v] = lo + r[h]; s_vo2i[v] = lo + r[h];
          a.tile_of_int[lo + r[h]] = t; s_tile[lo + r[h]] = t;
        }
      }
      __builtin_amdgcn_wave_barrier();
    }
  }
  __syncthreads();
  for (int v = tid; v < V; v += NT) a.v_i2o[s_vo2i[v]] = v;  // (the inverse, once every vertex has its place)
  for (int i = tid; i <= V; i += NT) s_a[i] = 0;
  __syncthreads();
  MINI_STAMP(12);
  // ---- triangle CSR (k_tri_count / fill / rows) ----
#pragma unroll
  for (int i = 0; i < kTri; ++i) {
    const int32_t t = tid + i * NT;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const int32_t vo = tv[i][c];
      tr[i][c] = -1;
      if (t >= a.T) continue;
      if (vo < 0 || vo >= V) { atomicOr(&a.flags[0], 2); a.tris_int[3 * t + c] = 0; continue; }
      const int32_t v = s_vo2i[vo];
      tv[i][c] = v;  // (internal id from here on)
      a.tris_int[3 * t + c] = v;
      tr[i][c] = atomicAdd(&s_a[v], 1);
    }
  }
  __syncthreads();
  mini_scan(s_a, V + 1, s_sc);
  for (int i = tid; i <= V; i += NT) a.trow[i] = s_a[i];
#pragma unroll
  for (int i = 0; i < kTri; ++i) {
#pragma unroll
    for (int c = 0; c < 3; ++c)
      if (tr[i][c] >= 0) s_buf[s_a[tv[i][c]] + tr[i][c]] = (uint32_t)(tid + i * NT);
  }
  __syncthreads();
  for (int v = tid; v < V; v += NT) (void)mini_sort_row(s_buf + s_a[v], s_a[v + 1] - s_a[v]);
  __syncthreads();
  for (int i = tid; i < s_a[V]; i += NT) a.tinc[i] = s_buf[i];
  __syncthreads();
  MINI_STAMP(13);
  // ---- edge order (k_edge_count / fill / rows / gather) ----
  for (int i = tid; i <= 2 * V; i += NT) s_a[i] = 0;
  for (int i = tid; i <= V; i += NT) s_b[i] = 0;  // degrees of the incidence CSR
  __syncthreads();
  constexpr int kEdg = kMiniE / kMiniThreads;  // 6 edges per thread, in registers from the count to the CSR fill
  int2 eij_o[kEdg];
  int32_t eb[kEdg], er[kEdg];
#pragma unroll
  for (int i = 0; i < kEdg; ++i) {
    const int32_t e = tid + i * NT;
    eij_o[i] = e < E ? a.edges[e] : make_int2(0, 0);
  }
#pragma unroll
  for (int i = 0; i < kEdg; ++i) {  // (derived edges: both ends valid and distinct)
    if (tid + i * NT >= E) continue;
    const int32_t si = s_vo2i[eij_o[i].x], sj = s_vo2i[eij_o[i].y];
    const int32_t ti = s_tile[si], tj = s_tile[sj];
    const int32_t lo = s_tl[ti], n = s_tl[ti + 1] - lo;
    eb[i] = 2 * lo + (ti == tj ? 0 : n) + (si - lo);  // = edge_bucket()
    er[i] = atomicAdd(&s_a[eb[i]], 1);
    eij_o[i] = make_int2(si, sj);                      // (internal ids from here on)
  }
  __syncthreads();
  mini_scan(s_a, 2 * V + 1, s_sc);
#pragma unroll
  for (int i = 0; i < kEdg; ++i)
    if (tid + i * NT < E) s_buf[s_a[eb[i]] + er[i]] = (uint32_t)(tid + i * NT);
  __syncthreads();
  for (int b = tid; b < 2 * V; b += NT) {
    const int d = s_a[b + 1] - s_a[b];
    if (d > 1) (void)mini_sort_row(s_buf + s_a[b], d);
  }
  for (int t = tid; t <= ntiles; t += NT) a.estart[t] = t < ntiles ? s_a[2 * max(0, min(s_tl[t], V))] : E;
  __syncthreads();
  int2* rank2 = reinterpret_cast<int2*>(a.rank);  // (2E ints; the edge ranks are spent)
  {
    int32_t ge[kEdg];
    int2 gij[kEdg];
    float gal[kEdg];
#pragma unroll
    for (int i = 0; i < kEdg; ++i) ge[i] = tid + i * NT < E ? (int32_t)s_buf[tid + i * NT] : 0;
#pragma unroll
    for (int i = 0; i < kEdg; ++i) { gij[i] = a.edges[ge[i]]; gal[i] = a.alpha[ge[i]]; }  // (E >= 1: index 0 is valid)
#pragma unroll
    for (int i = 0; i < kEdg; ++i) {
      const int32_t kk = tid + i * NT;
      if (kk >= E) continue;
      const int32_t e = ge[i];
      a.e_i2o[kk] = e;
      a.e_o2i[e] = kk; s_eo2i[e] = kk;
      const int32_t si = s_vo2i[gij[i].x], sj = s_vo2i[gij[i].y];
      const float2 pi = s_pos[gij[i].x], pj = s_pos[gij[i].y];
      a.eij[kk] = make_int2(si, sj);
      a.ew[kk] = make_float4(gal[i], gal[i], a.dsign * (pi.x - pj.x), a.dsign * (pi.y - pj.y));
      int2 r;
      r.x = atomicAdd(&s_b[si], 1);
      r.y = atomicAdd(&s_b[sj], 1);
      rank2[e] = r;
    }
  }
  __syncthreads();
  MINI_STAMP(14);
  // ---- incidence CSR (k_csr_fill / rows) ----
  mini_scan(s_b, V + 1, s_sc);
  for (int i = tid; i <= V; i += NT) a.grow[i] = s_b[i];
  {
    int2 fr[kEdg];
#pragma unroll
    for (int i = 0; i < kEdg; ++i) fr[i] = tid + i * NT < E ? rank2[tid + i * NT] : make_int2(0, 0);
#pragma unroll
    for (int i = 0; i < kEdg; ++i) {
      const int32_t e = tid + i * NT;
      if (e >= E) continue;
      s_buf[s_b[eij_o[i].x] + fr[i].x] = (uint32_t)e << 1;
      s_buf[s_b[eij_o[i].y] + fr[i].y] = ((uint32_t)e << 1) | 1u;
    }
  }
  __syncthreads();
  for (int v = tid; v < V; v += NT) {
    uint32_t* row = s_buf + s_b[v];
    const int d = s_b[v + 1] - s_b[v];
    (void)mini_sort_row(row, d);
    for (int i = 0; i < d; ++i) {
      const uint32_t x = row[i];
      const int32_t kk = s_eo2i[x >> 1];
      const int2 oe = a.edges[x >> 1];
      row[i] = (uint32_t)kk | ((x & 1u) << 31);
      a.gadj[s_b[v] + i] = s_vo2i[(x & 1u) ? oe.x : oe.y];  // (what k_csr_rows<true> leaves for the tile passes)
      a.ipos[2 * kk + (int32_t)(x & 1u)] = i;
    }
  }
  __syncthreads();
  for (int i = tid; i < 2 * E; i += NT) a.ginc[i] = s_buf[i];
  MINI_STAMP(15);
}

template <class T>
hipError_t dalloc(T** p, size_t n) {
  if (*p) (void)hipFree(*p);
  *p = nullptr;
  hipError_t e = hipMalloc(reinterpret_cast<void**>(p), std::max<size_t>(n, 1) * sizeof(T));
  const int fill = test_alloc_fill();  // (-1 in the product library; flame_hip.cpp, FLAME_HIP_TEST_HOOKS)
  if (e == hipSuccess && fill >= 0) { e = hipMemset(*p, fill, std::max<size_t>(n, 1) * sizeof(T)); }
  return e;
}

inline dim3 grid1(int64_t n) { return dim3((unsigned)std::max<int64_t>(1, (n + 255) / 256)); }

inline int bits_for(int64_t n) { int b = 1; while ((1ll << b) < n) ++b; return b; }

}  // namespace

static_assert(Plan::kGrid * Plan::kGrid == 1024, "k_grid_final assumes a 32 x 32 grid");

DevPlanner::~DevPlanner() { release(); }

void DevPlanner::release() {
  grid_deferred_ = false;  // (maps nobody will read)
  (void)wait_maps();
  void* ptrs[] = {cub_tmp_, keys_a_, keys_b_, vals_a_, vals_b_, seg_pos_, tile_of_int_, w_int_, wsort_, wscan_,
                  counts_, seg_tab_, estart_, tile_ext_, tile_meta_, flags_, grid_sum_, grid_cnt_, grid_w_,
                  grid_bounds_, gbbox_, tcub_tmp_, tcnt_, scan_agg_[0], scan_agg_[1], scan_flag_[0], scan_flag_[1],
                  cell_pyr_, rank_, reuse_cnt_, gadj_, ipos_};
  for (void* p : ptrs)
    if (p) (void)hipFree(p);
  scan_agg_[0] = scan_agg_[1] = nullptr; scan_flag_[0] = scan_flag_[1] = nullptr;
  cell_pyr_ = nullptr; map_tiles_ = 0; map_V_ = 0;
  rank_ = rank_tri_ = nullptr; reuse_cnt_ = nullptr; gadj_ = ipos_ = nullptr;
  if (hpin_) (void)hipHostFree(hpin_);
  hpin_ = nullptr; hpin_bytes_ = 0;
  if (s2_) (void)hipStreamDestroy(s2_);
  if (ev_fork_) (void)hipEventDestroy(ev_fork_);
  if (ev_join_) (void)hipEventDestroy(ev_join_);
  if (ev_grid_) (void)hipEventDestroy(ev_grid_);
  s2_ = nullptr; ev_fork_ = ev_join_ = ev_grid_ = nullptr;
  grid_pending_ = false;
  tcub_tmp_ = nullptr; tcnt_ = nullptr;
  capV2_ = 0; tcub_bytes_ = 0;
  cub_tmp_ = nullptr; keys_a_ = keys_b_ = nullptr; vals_a_ = vals_b_ = nullptr;
  seg_pos_ = tile_of_int_ = w_int_ = counts_ = seg_tab_ = estart_ = tile_ext_ = tile_meta_ = flags_ = nullptr;
  wsort_ = wscan_ = nullptr; grid_sum_ = nullptr; grid_cnt_ = grid_w_ = nullptr;
  grid_bounds_ = gbbox_ = nullptr;
  capV_ = capE_ = capT_ = capTiles_ = 0;
  cub_bytes_ = 0;
}

bool DevPlanner::eligible(const PlanOptions& opt, int32_t V, int32_t E, int32_t T, int tile_own, int depth,
                          bool single, int64_t lds_bytes) {
  if (single || !opt.batch_voff.empty() || opt.order_mode != 1 || depth < 1) return false;
  if (V < 2 || (int64_t)V >= (1ll << kIdBits) || (int64_t)E >= (1ll << kEdgeBits) || (int64_t)T >= (1ll << kEdgeBits))
    return false;
  const int ntiles = (V + tile_own - 1) / std::max(tile_own, 1);
  if (ntiles < 2 || ntiles > kSegCap) return false;
  const int64_t lds2 = (int64_t)kBucketInts * 4 + ((V + 31) / 32) * 4ll + kCapExt * 4 + kHash * 8;
  // the tile passes and the subtree kernel opt in to (almost) all of gfx950's 160 KiB of LDS: a
  // device / option set with less is the host builder's
  if ((int64_t)kSubLdsBytes > lds_bytes || 160 * 1024 - 512 > lds_bytes) return false;
  return lds2 <= lds_bytes;
}

// the look-back state of the next scan launch on a lane (epochs never repeat between resets)
static hipError_t scan_state(hipStream_t s, unsigned long long* agg, uint32_t* flag, uint32_t* epoch, int32_t* err,
                             ScanState* st) {
  if (*epoch == 0xfffffff0u) {  // wrap: forget every flag written so far
    hipError_t e = hipMemsetAsync(flag, 0, sizeof(uint32_t) * kScanMaxBlocks, s);
    if (e != hipSuccess) return e;
    *epoch = 0;
  }
  st->agg = agg; st->flag = flag; st->epoch = ++*epoch; st->err = err;
  return hipSuccess;
}

// One-launch exclusive / inclusive prefix sum of n int32 (falls back to rocprim's two-launch scan beyond the co-resident
// grid).  `lane` selects the look-back state: 0 = the builder's main stream, 1 = its second stream
// (two scans may be in flight at once, one per stream).
hipError_t DevPlanner::scan_i32(hipStream_t s, int lane, const int32_t* in, int32_t* out, int64_t n, bool inclusive,
                                void* cub_tmp, size_t cub_bytes) {
  if (n <= 0) return hipSuccess;
  const int64_t nb = (n + kScanTile - 1) / kScanTile;
  if (nb > kScanMaxBlocks || !scan_agg_[lane]) {
    size_t tb = cub_bytes;
    return inclusive ? rocprim::inclusive_scan(cub_tmp, tb, in, out, (size_t)((int)n), rocprim::plus<int32_t>(), s)
                     : rocprim::exclusive_scan(cub_tmp, tb, in, out, (int32_t)0, (size_t)((int)n), rocprim::plus<int32_t>(), s);
  }
  ScanState st;
  HIPRET(scan_state(s, scan_agg_[lane], scan_flag_[lane], &scan_epoch_[lane], flags_, &st));
  if (inclusive) hipLaunchKernelGGL(k_scan_i32<true>, dim3((unsigned)nb), dim3(kScanThreads), 0, s, in, out, n, st);
  else hipLaunchKernelGGL(k_scan_i32<false>, dim3((unsigned)nb), dim3(kScanThreads), 0, s, in, out, n, st);
  return hipGetLastError();
}

hipError_t DevPlanner::wait_maps() {
  HIPRET(flush_grid());
  if (!grid_pending_) return hipSuccess;
  grid_pending_ = false;
  return hipEventSynchronize(ev_grid_);
}

hipError_t DevPlanner::reserve(int32_t V, int32_t E, int32_t T, int ntiles) {
  HIPRET(wait_maps());  // (buffers may be re-allocated or rewritten from here on)
  const int64_t nk = std::max<int64_t>(std::max<int64_t>(V, 2 * (int64_t)E), 3 * (int64_t)T);
  if (V > capV_ || E > capE_ || T > capT_) {
    const int64_t v = std::max<int64_t>(V + V / 4, capV_), e = std::max<int64_t>(E + E / 4, capE_),
                  t = std::max<int64_t>(T + T / 4, capT_);
    const int64_t n = std::max<int64_t>(std::max<int64_t>(2 * v + 1, 2 * e), 3 * t);  // (stage C: 4V + 1 ints in keys_b_)
    HIPRET(dalloc(&keys_a_, (size_t)n)); HIPRET(dalloc(&keys_b_, (size_t)n));
    HIPRET(dalloc(&vals_a_, (size_t)std::max(std::max(2 * e, 2 * v), 3 * t))); HIPRET(dalloc(&vals_b_, (size_t)std::max(std::max(2 * e, 2 * v), 3 * t)));
    HIPRET(dalloc(&seg_pos_, (size_t)v)); HIPRET(dalloc(&tile_of_int_, (size_t)v));
    HIPRET(dalloc(&w_int_, (size_t)v)); HIPRET(dalloc(&wsort_, (size_t)v)); HIPRET(dalloc(&wscan_, (size_t)v));
    HIPRET(dalloc(&counts_, (size_t)v + 2));
    // places inside the counting CSRs' rows: [0, 2e) the edge stages (C: one per edge, D: one per edge
    // end), [2e, 2e + 3t) the triangle stages (half edges, then stage E on the second stream)
    HIPRET(dalloc(&rank_, (size_t)(2 * e + 3 * t) + 2));
    HIPRET(dalloc(&gadj_, (size_t)(2 * e) + 2));
    HIPRET(dalloc(&ipos_, (size_t)(2 * e) + 2));
    rank_tri_ = rank_ + 2 * e;
    capV_ = v; capE_ = e; capT_ = t;
    // temp storage of the library sorts / scans at the largest sizes
    size_t need = 0, b = 0;
    HIPRET(rocprim::radix_sort_keys(nullptr, b, keys_a_, keys_b_, (int)n, 0, 64, nullptr)); need = std::max(need, b);
    HIPRET(rocprim::radix_sort_pairs(nullptr, b, keys_a_, keys_b_, vals_a_, vals_b_, (int)n, 0, 64, nullptr)); need = std::max(need, b);
    HIPRET(rocprim::radix_sort_pairs(nullptr, b, vals_a_, vals_b_, vals_a_, vals_b_, (int)v, 0, 32, nullptr)); need = std::max(need, b);
    HIPRET(rocprim::inclusive_scan(nullptr, b, wsort_, wscan_, (size_t)((int)v), rocprim::plus<int32_t>(), nullptr)); need = std::max(need, b);
    HIPRET(rocprim::exclusive_scan(nullptr, b, counts_, counts_, (int32_t)0, (size_t)(2 * (int)v + 2), rocprim::plus<int32_t>(), nullptr)); need = std::max(need, b);  // (edge buckets: 2V + 1)
    if (need > cub_bytes_) { HIPRET(dalloc(reinterpret_cast<char**>(&cub_tmp_), need)); cub_bytes_ = need; }
  }
  (void)nk;
  if (V > capV2_) {
    const int64_t v = std::max<int64_t>(V + V / 4, 64);
    HIPRET(dalloc(&tcnt_, 2 * (size_t)v + 2));
    size_t b = 0;
    HIPRET(rocprim::exclusive_scan(nullptr, b, tcnt_, tcnt_, (int32_t)0, (size_t)((int)v + 1), rocprim::plus<int32_t>(), nullptr));
    size_t b2 = 0;  // the second stream also sorts the y list (stage A)
    HIPRET(rocprim::radix_sort_pairs(nullptr, b2, reinterpret_cast<uint32_t*>(tcnt_), reinterpret_cast<uint32_t*>(tcnt_),
                                              reinterpret_cast<uint32_t*>(tcnt_), reinterpret_cast<uint32_t*>(tcnt_), (int)v, 0, 32,
                                              nullptr));
    b = std::max(b, b2);
    if (b > tcub_bytes_) { HIPRET(dalloc(reinterpret_cast<char**>(&tcub_tmp_), b)); tcub_bytes_ = b; }
    capV2_ = v;
  }
  if (!s2_) {
    HIPRET(hipStreamCreateWithFlags(&s2_, hipStreamNonBlocking));
    HIPRET(hipEventCreateWithFlags(&ev_fork_, hipEventDisableTiming));
    HIPRET(hipEventCreateWithFlags(&ev_join_, hipEventDisableTiming));
    HIPRET(hipEventCreateWithFlags(&ev_grid_, hipEventDisableTiming));
  }
  if (!seg_tab_) {
    // 2 tables x 4 arrays, bbox (4), mid_raw x 2, child_base, mid_out, nseg x 2 (+ pad), axis
    HIPRET(dalloc(&seg_tab_, (size_t)kSegCap * 17 + 32));
    HIPRET(dalloc(&flags_, 8));
    for (int l = 0; l < 2; ++l) {
      HIPRET(dalloc(&scan_agg_[l], (size_t)kScanMaxBlocks));
      HIPRET(dalloc(&scan_flag_[l], (size_t)kScanMaxBlocks));
      scan_epoch_[l] = 0xfffffff0u;  // = "wrapped": the first scan on the lane zeroes the flags on ITS stream
                                     // (a legacy-stream hipMemset here collides with another thread's capture)
    }
    HIPRET(dalloc(&grid_sum_, (size_t)Plan::kGrid * Plan::kGrid));
    HIPRET(dalloc(&grid_cnt_, (size_t)Plan::kGrid * Plan::kGrid));
    HIPRET(dalloc(&grid_w_, (size_t)Plan::kGrid * Plan::kGrid));
    HIPRET(dalloc(&grid_bounds_, 4)); HIPRET(dalloc(&gbbox_, 4));
    HIPRET(dalloc(&cell_pyr_, (size_t)kPyrCells));
    HIPRET(dalloc(&reuse_cnt_, (size_t)kSegCap * kCntStride));
  }
  {  // D2H copies land in page-locked memory (a copy into pageable memory is staged and waited for)
    const size_t need = 256 + sizeof(TileDesc) * (size_t)std::max<int64_t>(ntiles + ntiles / 4, 64);
    if (need > hpin_bytes_) {
      if (hpin_) (void)hipHostFree(hpin_);
      hpin_ = nullptr; hpin_bytes_ = 0;
      HIPRET(hipHostMalloc(reinterpret_cast<void**>(&hpin_), need, hipHostMallocDefault));
      hpin_bytes_ = need;
    }
  }
  if (ntiles > capTiles_) {
    const int64_t n = std::max<int64_t>(ntiles + ntiles / 4, 64);
    HIPRET(dalloc(&estart_, (size_t)n + 2));
    HIPRET(dalloc(&tile_ext_, (size_t)n * kCapExt));
    HIPRET(dalloc(&tile_meta_, (size_t)n * kMetaWords));
    capTiles_ = n;
  }
  return hipSuccess;
}

hipError_t DevPlanner::build(hipStream_t s, const PlanOptions& opt, int32_t V, int32_t E, int32_t T, int ntiles,
                             int depth, const DevPlanInputs& in, DevPlanArrays* A, AllocTilesFn alloc_tiles,
                             void* alloc_ctx, std::vector<TileDesc>* tiles_host, bool* ok, bool* index_error,
                             const int32_t* user_flags_dev, int32_t* user_flags_host,
                             const std::function<hipError_t()>& after_partition) {
  *ok = false;
  *index_error = false;
  // option "plan_timing" (diagnostic; PlanOptions::timing): 1 synchronise after every stage and print its wall time; 2 host
  // enqueue time only; 3 device time of the build beside its host wall time; 4 / 5 the in-kernel stamps of k_mini_plan / of tile 0
  const int tlevel = opt.timing;
  const bool timing = tlevel > 0;
  auto tprev = std::chrono::steady_clock::now();
  const bool timing_nosync = tlevel >= 2;  // host enqueue time only
  auto lap = [&](const char* what) {
    if (!timing) return;
    if (!timing_nosync) (void)hipStreamSynchronize(s);
    const auto now = std::chrono::steady_clock::now();
    std::fprintf(stderr, "[plan_dev] %-14s %7.3f ms\n", what, std::chrono::duration<double, std::milli>(now - tprev).count());
    tprev = now;
  };
  HIPRET(reserve(V, E, T, ntiles));
  lap("reserve");
  const bool timing_dev = tlevel == 3;  // device time of the build's launches (events on `s`) beside its host wall time
  static hipEvent_t tev[2] = {nullptr, nullptr};
  const auto t_build0 = std::chrono::steady_clock::now();
  if (timing_dev) {
    if (!tev[0]) { (void)hipEventCreate(&tev[0]); (void)hipEventCreate(&tev[1]); }
    (void)hipEventRecord(tev[0], s);
  }
  const size_t lds1 = ((size_t)(V + 31) / 32) * 4 + kCapExt * 4 + kHash * 8;
  const size_t lds2 = lds1 + (size_t)kBucketInts * 4;  // + the bucket counts / offsets of the local edge list
  if (!attr_set_) {  // per planner (= per handle = per device), not per process
    HIPRET(hipFuncSetAttribute(reinterpret_cast<const void*>(k_tile_pass1), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 512));
    HIPRET(hipFuncSetAttribute(reinterpret_cast<const void*>(k_tile_pass2), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 2048));
    HIPRET(hipFuncSetAttribute(reinterpret_cast<const void*>(k_tile_fused), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 2048));
    HIPRET(hipFuncSetAttribute(reinterpret_cast<const void*>(k_mini_plan), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kMiniLds));
    attr_set_ = true;
  }
  // segment tables
  int32_t* st = seg_tab_;
  SegTab tab[2] = {{st, st + kSegCap, st + 2 * kSegCap, st + 3 * kSegCap},
                   {st + 4 * kSegCap, st + 5 * kSegCap, st + 6 * kSegCap, st + 7 * kSegCap}};
  uint32_t* bbox = reinterpret_cast<uint32_t*>(st + 8 * kSegCap);  // 4 * kSegCap
  int32_t* mid_raw[2] = {st + 12 * kSegCap, st + 13 * kSegCap};
  int32_t* child_base = st + 14 * kSegCap;
  int32_t* mid_out = st + 15 * kSegCap;
  int32_t* nseg = st + 16 * kSegCap;  // [2]
  int32_t* perm = A->v_i2o;
  const bool weighted = weight_mode_ != 0;
  const int vb = bits_for(V);
  // flags; the vertex order outputs (every entry is an index for the later stages, whatever the
  // partition); the triangle stage's counts and cursors
  const bool reuse = reuse_next_;  // the caller asked for the previous frame's partition (map_usable())
  // small frame of a graph sync: stages A-E (and the edge derivation + data terms in front of them)
  // in one launch of one workgroup (k_mini_plan)
  const bool mini = mini_set_ && reuse && mini_eligible(V, T, E) && ntiles <= kSegCap && expect_E_ == E;
  mini_used_ = mini_used_ || mini;  // (sticky until the next offer: a retry after a rejected partition keeps
  mini_set_ = false;                //  what the mini launch derived -- edges and data terms come first in it)
  // (the reuse path's tile counters and stage C's bucket counts ride along: their scratch is free from
  // the start there, while the bisection stages still use it)
  int32_t* ecnt0 = reinterpret_cast<int32_t*>(keys_b_);
  if (!mini) zero4(s, flags_, 8, A->v_o2i, V, tile_of_int_, V, reuse ? reuse_cnt_ : nullptr, (int64_t)kCntStride * ntiles,
        (reuse && E > 0) ? ecnt0 : nullptr, 2 * (int64_t)V + 1);
  reuse_next_ = false;
  last_reused_ = false;
  if (weight_mode_ == 2 && !reuse)  // (mode 1: w_int_ already holds the weights, see weights_from_tiles / _scale_)
    hipLaunchKernelGGL(k_weights_from_grid, grid1(V), dim3(256), 0, s, V, in.pos, grid_bounds_, grid_w_, w_int_);

  // ---- stage A ----
  int cur = 0;
  if (mini) {
    MiniArgs a;
    a.V = V; a.T = T; a.E_expect = E; a.ntiles = ntiles;
    a.cap = (int32_t)std::min<int64_t>(kOrderCap, ((int64_t)V * 3) / ntiles + 16);
    a.tris = mini_.tris ? mini_.tris : in.tris; a.pos = mini_.pos ? mini_.pos : in.pos; a.pos_out = const_cast<float2*>(in.pos);
    a.tris_out = const_cast<int32_t*>(in.tris);
    a.mu = mini_.mu; a.var = mini_.var; a.pred = mini_.pred;
    a.scale = mini_.scale; a.adaptive = mini_.adaptive; a.init_pred = mini_.init_pred;
    a.dsign = opt.d_sign < 0 ? -1.0f : 1.0f;
    a.edges = mini_.edges; a.alpha = mini_.alpha; a.z = mini_.z; a.wgt = mini_.wgt; a.x0 = mini_.x0;
    a.dflags = mini_.dflags;
    a.rank = rank_;
    a.vt = w_int_;
    a.gbbox = gbbox_; a.pyr = cell_pyr_; a.tab = tab[0]; a.nseg = nseg; a.flags = flags_;
    a.seg_pos = seg_pos_;
    a.v_i2o = A->v_i2o; a.v_o2i = A->v_o2i; a.tile_of_int = tile_of_int_;
    a.tris_int = A->tris; a.trow = A->trow; a.tinc = reinterpret_cast<uint32_t*>(A->tinc);
    a.e_i2o = A->e_i2o; a.e_o2i = A->e_o2i; a.eij = A->eij; a.ew = A->ew; a.estart = estart_;
    a.grow = A->grow; a.ginc = reinterpret_cast<uint32_t*>(A->ginc); a.gadj = gadj_; a.ipos = ipos_;
    const bool mini_prof = tlevel == 4;
    a.prof = mini_prof ? reinterpret_cast<long long*>(wscan_) : nullptr;  // (free in this path)
    hipLaunchKernelGGL(k_mini_plan, dim3(1), dim3(kMiniThreads), kMiniLds, s, a);
    if (mini_prof) {
      long long h[20] = {0};
      (void)hipStreamSynchronize(s);
      (void)hipMemcpy(h, wscan_, sizeof(h), hipMemcpyDeviceToHost);
      std::fprintf(stderr, "[mini] ticks per phase:");
      int last = 1;
      for (int q = 2; q < 20 && h[q]; ++q) { std::fprintf(stderr, " %lld", h[q] - h[q - 1]); last = q; }
      std::fprintf(stderr, "  total %lld\n", h[last] - h[1]);
    }
    last_reused_ = true;
  } else if (reuse) {
    // partition from the previous frame's tile map: lookup + counts, ranges, counting scatter
    int32_t* tile_cnt = reuse_cnt_;          // one counter per cache line
    int32_t* vt = w_int_;                    // (no weights in this path)
    int32_t* vrank = reinterpret_cast<int32_t*>(wscan_);  // (the weight scratch of the bisection path)
    hipLaunchKernelGGL(k_reuse_assign, grid1(V), dim3(256), 0, s, V, ntiles, in.pos, gbbox_, cell_pyr_, vt, vrank, tile_cnt);
    // (cost-balanced partitions hold 0.5..1.8 x the mean on purpose; a tile that is too LARGE for LDS or
    // a kernel configuration is found by the fit check of the caller like on any other partition)
    const int32_t cap = (int32_t)std::min<int64_t>(kOrderCap, ((int64_t)V * 3) / ntiles + 16);
    hipLaunchKernelGGL(k_reuse_scatter, grid1(V), dim3(256), 0, s, V, ntiles, cap, vt, vrank, tile_cnt, tab[0], nseg, flags_,
                       perm, seg_pos_);
    last_reused_ = true;
  } else {
  int levels = 0;
  while ((1 << levels) < ntiles) ++levels;
  // deep levels in LDS: from the first level whose segments hold <= kSubCap vertices (1.5 x margin
  // for uneven weighted splits) and <= kSubLeaves tiles
  int sub_level = levels;
  if (use_subtree_)
    for (int L = 0; L < levels; ++L)
      if (((int64_t)V >> L) * 5 / 4 + 2 <= kSubCap && ((ntiles >> L) + 1) * 2 <= kSubLeaves) {
        sub_level = std::min(levels, L + sub_extra_levels_);  // (+1 per overflow seen on this handle)
        break;
      }
  const bool lists = sub_level > 0;  // (a lone subtree sorts itself, on the coordinates)
  // the two sorted lists, double-buffered; scratch of the two entry sorts
  uint32_t* LX[2] = {vals_a_, vals_a_ + capV_};
  uint32_t* LY[2] = {vals_b_, vals_b_ + capV_};
  uint32_t* key_in = reinterpret_cast<uint32_t*>(keys_a_);
  uint32_t* key_out = reinterpret_cast<uint32_t*>(keys_a_) + capV_;
  uint32_t* val_in = reinterpret_cast<uint32_t*>(keys_b_);
  uint32_t* posx = reinterpret_cast<uint32_t*>(keys_b_) + capV_;
  int32_t* side = counts_;
  int32_t* axis = st + 16 * kSegCap + 16;
  if (lists) {  // ids sorted along x (this stream) and along y (second stream): (coordinate, id)
    HIPRET(hipEventRecord(ev_fork_, s));
    HIPRET(hipStreamWaitEvent(s2_, ev_fork_, 0));
    hipLaunchKernelGGL(k_rank_keys, grid1(V), dim3(256), 0, s, V, in.pos, 0, key_in, val_in);
    size_t tb2 = cub_bytes_;
    HIPRET(rocprim::radix_sort_pairs(cub_tmp_, tb2, key_in, key_out, val_in, LX[0], V, 0, 32, s));
    uint32_t* key_in2 = reinterpret_cast<uint32_t*>(wsort_);  // (the weight scratch is not in use yet)
    uint32_t* key_out2 = reinterpret_cast<uint32_t*>(wsort_) + capV_;
    uint32_t* val_in2 = reinterpret_cast<uint32_t*>(wscan_);
    hipLaunchKernelGGL(k_rank_keys, grid1(V), dim3(256), 0, s2_, V, in.pos, 1, key_in2, val_in2);
    size_t tb3 = tcub_bytes_;
    HIPRET(rocprim::radix_sort_pairs(tcub_tmp_, tb3, key_in2, key_out2, val_in2, LY[0], V, 0, 32, s2_));
    HIPRET(hipEventRecord(ev_join_, s2_));
    HIPRET(hipStreamWaitEvent(s, ev_join_, 0));
  }
  hipLaunchKernelGGL(k_rcb_init, grid1(V), dim3(256), 0, s, V, ntiles, perm, seg_pos_, tab[0], nseg, bbox, mid_raw[0]);
  int lb = 0;
  for (int lev = 0; lev < sub_level; ++lev, cur ^= 1, lb ^= 1) {
    hipLaunchKernelGGL(k_lvl_axis, dim3((unsigned)(((1 << lev) + 255) / 256)), dim3(256), 0, s, nseg + cur, tab[cur], LX[lb], LY[lb],
                       in.pos, axis, lev == 0 ? gbbox_ : nullptr);
    const int64_t scan_blocks = ((int64_t)V + kScanTile - 1) / kScanTile;
    const bool one_launch_scans = scan_blocks <= kScanMaxBlocks;
    if (weighted) {
      if (one_launch_scans) {  // gather + inclusive scan in one launch
        ScanState st;
        HIPRET(scan_state(s, scan_agg_[0], scan_flag_[0], &scan_epoch_[0], flags_, &st));
        hipLaunchKernelGGL(k_lvl_wscan, dim3((unsigned)scan_blocks), dim3(kScanThreads), 0, s, V, seg_pos_, axis, LX[lb],
                           LY[lb], w_int_, wsort_, wscan_, st);
      } else {
        hipLaunchKernelGGL(k_lvl_wgather, grid1(V), dim3(256), 0, s, V, seg_pos_, axis, LX[lb], LY[lb], w_int_, wsort_);
        size_t tb = cub_bytes_;
        HIPRET(rocprim::inclusive_scan(cub_tmp_, tb, wsort_, wscan_, (size_t)(V), rocprim::plus<int32_t>(), s));
      }
      hipLaunchKernelGGL(k_rcb_mid, grid1(V), dim3(256), 0, s, V, seg_pos_, tab[cur], wsort_, wscan_, mid_raw[cur]);
    }
    hipLaunchKernelGGL(k_rcb_split, dim3(1), dim3(kSegCap), 0, s, nseg + cur, nseg + (cur ^ 1), tab[cur], tab[cur ^ 1],
                       mid_raw[cur], mid_raw[cur ^ 1], weighted ? 1 : 0, child_base, mid_out, bbox);
    hipLaunchKernelGGL(k_lvl_side, grid1(V), dim3(256), 0, s, V, seg_pos_, axis, tab[cur].leaves, mid_out, LX[lb], LY[lb], side);
    if (one_launch_scans) {  // side flags + inclusive scan in one launch
      ScanState st;
      HIPRET(scan_state(s, scan_agg_[0], scan_flag_[0], &scan_epoch_[0], flags_, &st));
      hipLaunchKernelGGL(k_lvl_fscan, dim3((unsigned)scan_blocks), dim3(kScanThreads), 0, s, V, LX[lb], LY[lb], side, wsort_,
                         wscan_, st);
    } else {
      hipLaunchKernelGGL(k_lvl_flags, grid1(V), dim3(256), 0, s, V, LX[lb], LY[lb], side, wsort_);
      size_t tb = cub_bytes_;
      HIPRET(rocprim::inclusive_scan(cub_tmp_, tb, wsort_, wscan_, (size_t)(V), rocprim::plus<int32_t>(), s));
    }
    hipLaunchKernelGGL(k_lvl_scatter, grid1(V), dim3(256), 0, s, V, seg_pos_, tab[cur], mid_out, child_base, wsort_, wscan_,
                       LX[lb], LY[lb], LX[lb ^ 1], LY[lb ^ 1]);
  }
  // perm = the x list (any order inside a segment serves the later stages); posx for the subtrees
  if (lists) hipLaunchKernelGGL(k_lvl_posx, grid1(V), dim3(256), 0, s, V, LX[lb], posx, perm);
  if (sub_level < levels) {
    const size_t lds_sub = kSubLdsBytes;
    if (!sub_attr_set_) {
      HIPRET(hipFuncSetAttribute(reinterpret_cast<const void*>(k_rcb_subtree), hipFuncAttributeMaxDynamicSharedMemorySize,
                                 (int)lds_sub));
      sub_attr_set_ = true;
    }
    if (sub_level == 0) hipLaunchKernelGGL(k_rcb_bbox, dim3(1), dim3(1024), 0, s, nseg + cur, tab[cur], perm, in.pos, bbox);
    if (sub_level == 0) hipLaunchKernelGGL(k_save_gbbox, dim3(1), dim3(64), 0, s, bbox, gbbox_);
    hipLaunchKernelGGL(k_rcb_subtree, dim3(1 << sub_level), dim3(kSubThreads), lds_sub, s, nseg + cur, tab[cur], tab[cur ^ 1],
                       nseg + (cur ^ 1), ntiles, perm, seg_pos_, in.pos, LX[lb], LY[lb], posx, w_int_, weighted ? 1 : 0, vb,
                       lists ? 0 : 1, (opt.debug_sub_cap > 0 && sub_extra_levels_ == 0) ? std::min(opt.debug_sub_cap, kSubCap) : kSubCap,
                       flags_);
    cur ^= 1;
  }
  }  // (exact bisection)
  if (!reuse)  // (the reuse path writes the table of a finished bisection by construction)
    hipLaunchKernelGGL(k_rcb_check, dim3(1), dim3(kSegCap), 0, s, nseg + cur, tab[cur], ntiles, flags_);
  const SegTab leaf = tab[cur];  // lo = vstart, hi = vstart + n_own per tile
  lap("A rcb");

  // ---- stage B ----
  if (!mini)
    hipLaunchKernelGGL(k_tile_order, dim3((unsigned)ntiles), dim3(256), 0, s, V, ntiles, leaf.lo, leaf.hi, in.pos, gbbox_, perm,
                       A->v_o2i, tile_of_int_, flags_);

  lap("B morton");
  if (after_partition) HIPRET(after_partition());  // the caller's edge / data arrays arrive now
  // ---- stage E: vertex -> triangle CSR; needs only the vertex order, so it runs beside the edge
  // stages (C, D, tile pass 1) on a second stream and joins before the flags are read ----
  const bool tri_stage = T > 0 && in.tris && !mini;
  if (tri_stage) {
    HIPRET(hipEventRecord(ev_fork_, s));
    HIPRET(hipStreamWaitEvent(s2_, ev_fork_, 0));
    zero4(s2_, tcnt_, (int64_t)V + 2);  // (on the second stream: not one more job of the build's first launch)
    hipLaunchKernelGGL(k_tri_count, grid1(3 * (int64_t)T), dim3(256), 0, s2_, 3 * T, V, in.tris, A->v_o2i, A->tris, tcnt_,
                       rank_tri_, flags_);
    HIPRET(scan_i32(s2_, 1, tcnt_, A->trow, (int64_t)V + 1, false, tcub_tmp_, tcub_bytes_));
    hipLaunchKernelGGL(k_tri_fill, grid1(3 * (int64_t)T), dim3(256), 0, s2_, 3 * T, V, in.tris, A->tris, A->trow, rank_tri_,
                       reinterpret_cast<uint32_t*>(A->tinc));
    hipLaunchKernelGGL(k_csr_rows<false>, grid1(V), dim3(256), 0, s2_, V, A->trow, nullptr,
                       reinterpret_cast<uint32_t*>(A->tinc));
    HIPRET(hipEventRecord(ev_join_, s2_));
  }
  // ---- stage C ----
  if (mini) {
    // (stages C and D are part of k_mini_plan)
  } else if (E > 0) {
    int32_t* ecnt = reinterpret_cast<int32_t*>(keys_b_);     // 2V + 1 bucket counts
    int32_t* eoff = reinterpret_cast<int32_t*>(vals_b_);     // (the lists of stage A are dead)
    uint32_t* esorted = reinterpret_cast<uint32_t*>(keys_a_);
    // stage C's bucket counts (the reuse path zeroed them with its first launch); stage D's degree
    // counts are zeroed by k_edge_count
    if (!reuse) zero4(s, ecnt, 2 * (int64_t)V + 1);
    hipLaunchKernelGGL(k_edge_count, grid1(std::max<int64_t>(E, (int64_t)V + 1)), dim3(256), 0, s, E, V, in.edges, A->v_o2i,
                       tile_of_int_, leaf.lo, leaf.hi, ecnt, rank_, flags_, counts_);
    HIPRET(scan_i32(s, 0, ecnt, eoff, 2 * (int64_t)V + 1, false, cub_tmp_, cub_bytes_));
    hipLaunchKernelGGL(k_edge_fill, grid1(E), dim3(256), 0, s, E, V, in.edges, A->v_o2i, tile_of_int_, leaf.lo, leaf.hi, eoff,
                       rank_, esorted);
    // bucket sort + edge records + degree counts of stage D (kSegCap tiles <= 2V rows / 256 blocks * 256: the
    // grid covers estart[0 .. ntiles] as long as 2V >= ntiles + 1, i.e. always: a tile owns a vertex)
    hipLaunchKernelGGL(k_edge_rows_gather, grid1(std::max<int64_t>(2 * (int64_t)V, ntiles + 1)), dim3(256), 0, s, 2 * V, eoff,
                       esorted, in.edges, in.alpha, in.beta, in.pos, A->v_o2i, A->e_i2o, A->e_o2i, A->eij, A->ew, V, E, ntiles,
                       leaf.lo, estart_, opt.d_sign < 0 ? -1.0f : 1.0f, counts_, reinterpret_cast<int2*>(rank_));
  } else {
    HIPRET(hipMemsetAsync(estart_, 0, sizeof(int32_t) * (size_t)(ntiles + 2), s));
  }
  lap("C edges");
  // ---- stage D ----
  if (mini) {
  } else if (E > 0) {
    int2* rank2 = reinterpret_cast<int2*>(rank_);  // (degree counts and ranks: k_edge_rows_gather)
    HIPRET(scan_i32(s, 0, counts_, A->grow, (int64_t)V + 1, false, cub_tmp_, cub_bytes_));
    hipLaunchKernelGGL(k_csr_fill, grid1(E), dim3(256), 0, s, E, A->eij, A->e_o2i, A->grow, rank2,
                       reinterpret_cast<uint32_t*>(A->ginc));
    hipLaunchKernelGGL(k_csr_rows<true>, grid1(V), dim3(256), 0, s, V, A->grow, A->e_o2i,
                       reinterpret_cast<uint32_t*>(A->ginc), A->eij, gadj_, ipos_);
  } else {
    HIPRET(hipMemsetAsync(A->grow, 0, sizeof(int32_t) * ((size_t)V + 1), s));
  }
  lap("D csr");
  // ---- stage F ----
  TileGraph G;
  G.V = V; G.depth = depth; G.grow = A->grow; G.ginc = A->ginc; G.eij = A->eij; G.e_i2o = A->e_i2o; G.e_o2i = A->e_o2i;
  G.gadj = gadj_; G.ipos = ipos_;
  // The tile arrays are sized from the totals pass 1 counts.  A frame stream does not wait for them:
  // the arrays are allocated for 9/8 of the PREVIOUS build's totals, pass 2 is launched right behind
  // pass 1 and the one synchronisation at the end tells whether the guess held (flags bit 128: the
  // exact sizes are known by then and pass 2 is simply run again).  First build on a handle: one more
  // round trip, as before.
  const bool spec = spec_nv_ > 0 && spec_tiles_ >= ntiles;
  if (spec && alloc_tiles(alloc_ctx, (size_t)spec_tiles_, (size_t)spec_nv_, (size_t)spec_ne_, (size_t)spec_ns_) != 0)
    return hipErrorOutOfMemory;
  const bool fused = spec && ntiles <= kScanMaxBlocks && scan_agg_[0];
  int32_t* hflags = reinterpret_cast<int32_t*>(hpin_);
  int32_t* huser = reinterpret_cast<int32_t*>(hpin_ + 64);
  TileDesc* htiles = reinterpret_cast<TileDesc*>(hpin_ + 256);
  auto tile_out = [&]() {
    TileOut O;
    O.ew = A->ew; O.tiles = A->tiles; O.t_vmap = A->t_vmap; O.t_emap = A->t_emap; O.t_eij = A->t_eij; O.t_ew = A->t_ew;
    O.t_srow = A->t_srow; O.flags = flags_; O.lane_order = opt.lane_order == 2 ? 1 : 0;
    const bool tile_prof = tlevel == 5;
    O.prof = tile_prof ? reinterpret_cast<long long*>(wscan_) : nullptr;  // (free by now)
    return O;
  };
  auto launch_pass2 = [&]() -> hipError_t {
    hipLaunchKernelGGL(k_tile_pass2, dim3(ntiles), dim3(kP2Threads), lds2, s, G, leaf.lo, leaf.hi, estart_, tile_ext_,
                       tile_meta_, tile_out());
    return hipGetLastError();
  };
  static_assert(sizeof(TileDesc) % 4 == 0, "descriptors are published as words");
  bool tiles_built = false;
  auto publish = [&]() -> hipError_t {
    const int32_t words = tiles_built ? (int32_t)(sizeof(TileDesc) / 4) * ntiles : 0;
    hipLaunchKernelGGL(k_publish, grid1(std::max(words, 8)), dim3(256), 0, s, flags_,
                       (user_flags_dev && user_flags_host) ? user_flags_dev : nullptr,
                       tiles_built ? reinterpret_cast<const int32_t*>(A->tiles) : nullptr, words, hflags, huser,
                       reinterpret_cast<int32_t*>(htiles));
    return hipGetLastError();
  };
  if (fused) {
    // (the triangle stage on the second stream raises its flags on the same word: join first, so that
    // every tile of the launch sees the same word at its start)
    if (tri_stage) HIPRET(hipStreamWaitEvent(s, ev_join_, 0));
    ScanState st;
    HIPRET(scan_state(s, scan_agg_[0], scan_flag_[0], &scan_epoch_[0], flags_, &st));
    hipLaunchKernelGGL(k_tile_fused, dim3(ntiles), dim3(kP2Threads), lds2, s, G, leaf.lo, leaf.hi, estart_, tile_ext_,
                       tile_meta_, tile_out(), st, spec_nv_, spec_ne_, spec_ns_);
    tiles_built = true;
    if (tlevel == 5) {
      long long h[10] = {0};
      (void)hipStreamSynchronize(s);
      (void)hipMemcpy(h, wscan_, sizeof(h), hipMemcpyDeviceToHost);
      std::fprintf(stderr, "[tile0] ticks: rings %lld hash %lld keys %lld lookback %lld sort %lld slot rows %lld records %lld  total %lld\n",
                   h[2] - h[1], h[3] - h[2], h[4] - h[3], h[5] - h[4], h[6] - h[5], h[7] - h[6], h[8] - h[7], h[8] - h[1]);
    }
  } else {
    hipLaunchKernelGGL(k_tile_pass1, dim3(ntiles), dim3(kP1Threads), lds1, s, G, leaf.lo, leaf.hi, tile_ext_, tile_meta_);
    hipLaunchKernelGGL(k_tile_offsets, dim3(1), dim3(kSegCap), 0, s, ntiles, tile_meta_, flags_, spec ? spec_nv_ : 0,
                       spec ? spec_ne_ : 0, spec ? spec_ns_ : 0);
    if (tri_stage) HIPRET(hipStreamWaitEvent(s, ev_join_, 0));
    if (spec) { HIPRET(launch_pass2()); tiles_built = true; }
  }
  lap("E tris (joined)");
  HIPRET(publish());  // (the caller's own check word rides on the same sync)
  if (timing_dev) (void)hipEventRecord(tev[1], s);
  const auto t_enq = std::chrono::steady_clock::now();
  HIPRET(hipStreamSynchronize(s));
  HIPRET(hipGetLastError());
  if (timing_dev) {
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, tev[0], tev[1]);
    const auto now = std::chrono::steady_clock::now();
    std::fprintf(stderr, "[plan_dev] device %.3f ms; host: enqueued after %.3f ms, synchronised after %.3f ms\n", ms,
                 std::chrono::duration<double, std::milli>(t_enq - t_build0).count(),
                 std::chrono::duration<double, std::milli>(now - t_build0).count());
  }
  if (user_flags_dev && user_flags_host) {
    for (int i = 0; i < 4; ++i) user_flags_host[i] = huser[i];
    // word 2: the flags of the edge derivation in front of this build (edges_from_tris without a round trip)
    if (user_flags_host[2] & 32) return hipSuccess;                          // look-back timeout: not ok
    if (user_flags_host[2] & 2) { *index_error = true; return hipSuccess; }  // triangle index out of range
  }
  lap(spec ? "F+G pass1+2+sync" : "F pass1+sync");
  // word 0: the caller's failure flag; word 1: the number of edges the graph sync derived -- a build
  // that was launched on a PREDICTED edge count and got it wrong is worthless whatever else it says
  if (user_flags_dev && user_flags_host && (user_flags_host[0] || (expect_E_ >= 0 && user_flags_host[1] != expect_E_)))
    return hipSuccess;
  if (hflags[0] & 32) return hipSuccess;  // a scan's look-back timed out (never seen): not ok -> host builder
  if (hflags[0] & 64) { map_tiles_ = 0; return hipSuccess; }  // the reused partition does not suit this frame: not ok
  if (hflags[0] & 2) { *index_error = true; return hipSuccess; }
  if (hflags[0] & 16) {  // a subtree outgrew its workgroup (uneven weighted splits): hand over one
    // level later from now on; after three such steps every level goes through the global kernels
    if (++sub_extra_levels_ > 3) use_subtree_ = false;
    return build(s, opt, V, E, T, ntiles, depth, in, A, alloc_tiles, alloc_ctx, tiles_host, ok, index_error,
                 user_flags_dev, user_flags_host, nullptr);  // (the caller's arrays are staged by now)
  }
  if (hflags[0] & 5) return hipSuccess;  // a tile does not fit (or the partition is inconsistent): not ok
  // the next build on this handle speculates on these totals
  spec_tiles_ = std::max(ntiles, spec_tiles_);
  spec_nv_ = hflags[1] + hflags[1] / 8 + 1024; spec_ne_ = hflags[2] + hflags[2] / 8 + 1024; spec_ns_ = hflags[3] + hflags[3] / 8 + 1024;
  if (!spec || (hflags[0] & 128)) {
    if (alloc_tiles(alloc_ctx, (size_t)ntiles, (size_t)hflags[1], (size_t)hflags[2], (size_t)hflags[3]) != 0)
      return hipErrorOutOfMemory;
    // ---- stage G (first build on a handle, or the speculative arrays were too small) ----
    if (hflags[0] & 128) HIPRET(hipMemsetAsync(flags_, 0, sizeof(int32_t), s));  // (nothing else was set: checked above)
    HIPRET(launch_pass2());
    tiles_built = true;
    HIPRET(publish());
    HIPRET(hipStreamSynchronize(s));
    HIPRET(hipGetLastError());
    lap("G pass2+sync");
  }
  tiles_host->assign(htiles, htiles + ntiles);
  if (hflags[0] & (8 | 32)) return hipSuccess;
  *ok = true;
  return hipGetLastError();
}

hipError_t DevPlanner::weights_from_tiles(hipStream_t s, int32_t V, const DevPlanArrays& A) {
  hipLaunchKernelGGL(k_weights_from_tiles, grid1(V), dim3(256), 0, s, V, A.v_i2o, tile_of_int_, A.tiles, w_int_);
  weight_mode_ = 1;
  return hipGetLastError();
}

hipError_t DevPlanner::weights_scale_by_tiles(hipStream_t s, int32_t V, int ntiles, long long total_cost,
                                              const DevPlanArrays& A) {
  hipLaunchKernelGGL(k_weights_scale, grid1(V), dim3(256), 0, s, V, A.v_i2o, tile_of_int_, A.tiles, ntiles, total_cost,
                     w_int_);
  weight_mode_ = 1;
  return hipGetLastError();
}

hipError_t DevPlanner::edges_from_tris(hipStream_t s, int32_t V, int32_t T, const int32_t* tris, const float2* pos,
                                       int2* edges, float* alpha, int32_t* E_out, bool* index_error, int32_t* nan_flag,
                                       const std::function<void()>& while_running, int32_t expected_E,
                                       const std::function<hipError_t()>& before_positions) {
  *E_out = 0;
  *index_error = false;
  if (T <= 0) return hipSuccess;
  const int32_t n = 3 * T;
  HIPRET(reserve(V, n, T, 1));  // E <= 3T
  const bool fused = ((int64_t)V + 255) / 256 <= kScanMaxBlocks && scan_agg_[0];
  // no round trip: the chain's flags (bit 2 bad index, bit 32 look-back timeout) go to the caller's word
  // nan_flag[2] -- the build() that follows zeroes flags_ with its first launch and reads the caller's
  // words at its first synchronisation
  const bool own_flags = expected_E >= 0 && nan_flag;
  int32_t* he_flags = own_flags ? nan_flag + 2 : flags_;
  int32_t* f = reinterpret_cast<int32_t*>(vals_a_);
  int32_t* idx = reinterpret_cast<int32_t*>(vals_b_);
  int32_t* cnt = tcnt_;                                     // V + 1 counts
  int32_t* off = counts_;                                   // V + 1 row offsets
  uint32_t* out = reinterpret_cast<uint32_t*>(keys_a_);     // 3T entries
  if (!own_flags) HIPRET(hipMemsetAsync(flags_, 0, 8 * sizeof(int32_t), s));
  HIPRET(hipMemsetAsync(cnt, 0, sizeof(int32_t) * ((size_t)V + 1), s));
  hipLaunchKernelGGL(k_he_count, grid1(n), dim3(256), 0, s, n, V, tris, cnt, rank_tri_, he_flags);
  HIPRET(scan_i32(s, 0, cnt, off, (int64_t)V + 1, false, cub_tmp_, cub_bytes_));
  hipLaunchKernelGGL(k_he_fill, grid1(n), dim3(256), 0, s, n, V, tris, off, rank_tri_, out);
  // (the half-edge kernels above read the triangles only: the caller stages the positions now, its
  // host-synchronous copy runs beside them)
  if (before_positions) HIPRET(before_positions());
  int32_t* total = (expected_E >= 0 && nan_flag) ? nan_flag + 1 : flags_ + 4;
  if (fused) {
    ScanState st;
    HIPRET(scan_state(s, scan_agg_[0], scan_flag_[0], &scan_epoch_[0], he_flags, &st));
    hipLaunchKernelGGL(k_he_unique, grid1(V), dim3(256), 0, s, V, off, out, pos, edges, alpha, total, nan_flag, st);
  } else {
    hipLaunchKernelGGL(k_csr_rows<false>, grid1(V), dim3(256), 0, s, V, off, nullptr, out);
    hipLaunchKernelGGL(k_he_mark, grid1(V), dim3(256), 0, s, V, off, out, f);
    HIPRET(scan_i32(s, 0, f, idx, n, false, cub_tmp_, cub_bytes_));
    hipLaunchKernelGGL(k_he_compact, grid1(V), dim3(256), 0, s, V, off, out, f, idx, pos, edges, alpha, total, nan_flag);
  }
  if (expected_E >= 0 && nan_flag) {
    // No round trip: the caller goes on with the edge count it predicted (Euler: E = V + T - 1 for a
    // triangulated disk); the true count lands in nan_flag[1] and comes back with the plan builder's
    // first synchronisation, which rejects the build when the prediction was wrong.
    if (while_running) while_running();
    *E_out = expected_E;
    return hipGetLastError();
  }
  int32_t* h = reinterpret_cast<int32_t*>(hpin_);  // (page-locked, see reserve())
  HIPRET(hipMemcpyAsync(h, flags_, 8 * sizeof(int32_t), hipMemcpyDeviceToHost, s));
  if (while_running) while_running();  // host work of the caller overlaps the kernels above
  HIPRET(hipStreamSynchronize(s));
  HIPRET(hipGetLastError());
  if (h[0] & 32) return hipErrorUnknown;  // a scan's look-back timed out (never seen)
  if (h[0] & 2) { *index_error = true; return hipSuccess; }
  *E_out = h[4];
  return hipSuccess;
}

// flags[0] |= 1 when any value of up to five arrays is not finite (the upload path's input check: one
// launch instead of one per array)
struct FiniteJob { const float* p[5]; int64_t n[5]; };
__global__ __launch_bounds__(256) void k_check_finite5(FiniteJob j, int32_t* flags) {
  const int64_t stride = (int64_t)gridDim.x * 256;
  bool bad = false;
  for (int a = 0; a < 5; ++a)
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < j.n[a]; i += stride) bad |= !isfinite(j.p[a][i]);
  if (bad) atomicOr(flags, 1);
}

hipError_t DevPlanner::check_finite(hipStream_t s, int32_t* flags, const float* a, int64_t na, const float* b, int64_t nb,
                                    const float* c, int64_t nc, const float* d, int64_t nd, const float* e, int64_t ne) {
  FiniteJob j = {{a, b, c, d, e}, {a ? na : 0, b ? nb : 0, c ? nc : 0, d ? nd : 0, e ? ne : 0}};
  int64_t m = 0;
  for (int k = 0; k < 5; ++k) m = std::max(m, j.n[k]);
  if (m <= 0) return hipSuccess;
  hipLaunchKernelGGL(k_check_finite5, dim3((unsigned)std::min<int64_t>(2048, (m + 255) / 256)), dim3(256), 0, s, j, flags);
  return hipGetLastError();
}

hipError_t DevPlanner::sync_data(hipStream_t s, int32_t V, const float* mu, const float* var, const float* pred,
                                 float scale, int adaptive, int init_pred, float* z, float* wgt, float* x0,
                                 int32_t* nan_flag) {
  if (V <= 0) return hipSuccess;
  hipLaunchKernelGGL(k_sync_data, grid1(V), dim3(256), 0, s, V, mu, var, pred, scale, adaptive, init_pred, z, wgt, x0,
                     nan_flag);
  return hipGetLastError();
}

hipError_t DevPlanner::update_grid(hipStream_t s, int32_t V, int ntiles, const DevPlanInputs& in,
                                   const DevPlanArrays& A) {
  // Only the NEXT build reads these maps: they are made on the second stream, beside the iterations
  // of this frame instead of in front of them -- and their launches are ENQUEUED behind the first
  // iterations too (flush_grid(), called by the solve once its own launches are out; the host's
  // enqueue time of three launches was in front of the first iteration otherwise).  Every entry point
  // that touches what they read or write (the next build, the next edge derivation, the caller's next
  // upload) goes through wait_maps(), which enqueues them if nobody has and waits for ev_grid_.
  if (s2_) {
    HIPRET(hipEventRecord(ev_fork_, s));
    HIPRET(hipStreamWaitEvent(s2_, ev_fork_, 0));
  }
  grid_job_.stream = s2_ ? s2_ : s;
  grid_job_.V = V; grid_job_.pos = in.pos; grid_job_.v_i2o = A.v_i2o; grid_job_.tiles = A.tiles;
  grid_deferred_ = true;
  map_tiles_ = ntiles; map_V_ = V;
  grid_tiles_ = ntiles;
  return hipSuccess;
}

hipError_t DevPlanner::flush_grid(hipEvent_t after) {
  if (!grid_deferred_) return hipSuccess;
  grid_deferred_ = false;
  const int n = Plan::kGrid * Plan::kGrid;
  hipStream_t g2 = grid_job_.stream;
  if (after && s2_ && g2 == s2_) HIPRET(hipStreamWaitEvent(g2, after, 0));
  const int32_t V = grid_job_.V;
  zero4(g2, reinterpret_cast<int32_t*>(grid_sum_), 2 * (int64_t)n, grid_cnt_, n, cell_pyr_, kPyrAtomicCells);
  hipLaunchKernelGGL(k_grid_accum, grid1(V), dim3(256), 0, g2, V, grid_job_.pos, grid_job_.v_i2o, tile_of_int_,
                     grid_job_.tiles, gbbox_, reinterpret_cast<unsigned long long*>(grid_sum_), grid_cnt_, cell_pyr_);
  hipLaunchKernelGGL(k_grid_final, dim3(1), dim3(1024), 0, g2, V, reinterpret_cast<unsigned long long*>(grid_sum_),
                     grid_cnt_, grid_w_, gbbox_, grid_bounds_, cell_pyr_);
  if (s2_) {
    HIPRET(hipEventRecord(ev_grid_, s2_));
    grid_pending_ = true;
  }
  return hipGetLastError();
}

hipError_t launch_assign_lanes(hipStream_t s, int32_t ntiles, int32_t e_max, const TileDesc* tiles, uint2* t_eij,
                               float4* t_ew, int32_t* t_emap, bool slot12) {
  if (ntiles <= 0 || e_max <= 0) return hipSuccess;
  const unsigned by = (unsigned)((e_max + 255) / 256);
  if (slot12) hipLaunchKernelGGL(k_assign_lanes<true>, dim3((unsigned)ntiles, by), dim3(256), 0, s, tiles, t_eij, t_ew, t_emap);
  else hipLaunchKernelGGL(k_assign_lanes<false>, dim3((unsigned)ntiles, by), dim3(256), 0, s, tiles, t_eij, t_ew, t_emap);
  return hipGetLastError();
}

}  // namespace flamehip
