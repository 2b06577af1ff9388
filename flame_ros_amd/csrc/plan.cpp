// flame_ros_amd/csrc/plan.cpp -- see plan.h.
#include "plan.h"

#include <algorithm>
#include <cmath>
#include <numeric>
#include <thread>
#include <chrono>
#include <cstdio>
#include <cstdlib>

#include "../../include/flame_hip.h"

namespace flamehip {
namespace {

// Split position of idx[lo,hi) along `axis` for l1 of `leaves` parts.  The order along the axis is
// the total order (coordinate, original id).  Unweighted: the first (hi-lo) l1 / leaves vertices
// go left.  Weighted (w != nullptr, INTEGER weights so that every implementation -- this one and the
// device builder in plan_dev.hip -- computes the same sums whatever the summation order): with
// acc(m) = weight before position m in that order, the split is the first m with
// (2 acc(m) + w_m) leaves >= 2 total l1, clamped so that both sides keep one vertex per leaf.
int split_range(const float* pos, const int32_t* w, std::vector<int32_t>& idx, int lo, int hi,
                int axis, int l1, int leaves) {
  auto less = [&](int32_t a, int32_t b) {
    const float pa = pos[2 * a + axis], pb = pos[2 * b + axis];
    return pa < pb || (pa == pb && a < b);
  };
  if (!w) {
    const int mid = lo + (int)(((int64_t)(hi - lo) * l1) / leaves);
    std::nth_element(idx.begin() + lo, idx.begin() + mid, idx.begin() + hi, less);
    return mid;
  }
  // weighted quickselect (expected O(n)): narrow [a,b) keeping `before` = weight left of a
  int64_t total = 0;
  for (int k = lo; k < hi; ++k) total += w[idx[k]];
  const int64_t rhs = 2 * total * l1;  // compare (2 acc + w) * leaves against this
  int64_t before = 0;
  int a = lo, b = hi;
  while (b - a > 32) {
    const int32_t c0 = idx[a], c1 = idx[a + (b - a) / 2], c2 = idx[b - 1];
    const int32_t piv = less(c0, c1) ? (less(c1, c2) ? c1 : (less(c0, c2) ? c2 : c0))
                                     : (less(c0, c2) ? c0 : (less(c1, c2) ? c2 : c1));
    const int mi = (int)(std::partition(idx.begin() + a, idx.begin() + b,
                                        [&](int32_t v) { return less(v, piv); }) - idx.begin());
    if (mi == a) {  // the pivot is the minimum of the range: settle it at position a
      std::iter_swap(idx.begin() + a, std::find(idx.begin() + a, idx.begin() + b, piv));
      if ((2 * before + w[piv]) * leaves >= rhs) { b = a; break; }
      before += w[piv];
      ++a;
      continue;
    }
    int64_t lw = 0;
    for (int k = a; k < mi; ++k) lw += w[idx[k]];
    // the answer is <= mi iff acc(mi) = before + lw already reaches the target
    if (2 * (before + lw) * leaves >= rhs) b = mi;
    else { before += lw; a = mi; }
  }
  std::sort(idx.begin() + a, idx.begin() + b, less);
  int mid = a;
  {
    int64_t acc = before;
    while (mid < b && (2 * acc + w[idx[mid]]) * leaves < rhs) acc += w[idx[mid++]];
  }
  // every part keeps at least one vertex per leaf it must still produce
  mid = std::max(lo + l1, std::min(mid, hi - (leaves - l1)));
  return std::max(lo, std::min(mid, hi));
}

// Recursive coordinate bisection of idx[lo,hi) into `leaves` parts of near-equal size (or cost,
// when weights are given); parts are emitted in recursion order, which keeps spatial neighbours
// close in the tile order.
void rcb(const float* pos, const int32_t* w, std::vector<int32_t>& idx, int lo, int hi, int leaves,
         std::vector<int32_t>* leaf_start) {
  if (leaves <= 1 || hi - lo <= 1) {
    leaf_start->push_back(lo);
    for (int k = 1; k < leaves; ++k) leaf_start->push_back(hi);  // empty extra leaves
    return;
  }
  float mn[2] = {INFINITY, INFINITY}, mx[2] = {-INFINITY, -INFINITY};
  for (int k = lo; k < hi; ++k)
    for (int a = 0; a < 2; ++a) {
      const float p = pos[2 * idx[k] + a];
      mn[a] = std::min(mn[a], p);
      mx[a] = std::max(mx[a], p);
    }
  const int axis = (mx[1] - mn[1] > mx[0] - mn[0]) ? 1 : 0;
  const int l1 = leaves / 2, l2 = leaves - l1;
  const int mid = split_range(pos, w, idx, lo, hi, axis, l1, leaves);
  rcb(pos, w, idx, lo, mid, l1, leaf_start);
  rcb(pos, w, idx, mid, hi, l2, leaf_start);
}

// Same bisection, but the two halves of the top `par_levels` levels run on separate threads; each
// half appends its leaves to its own list, concatenated in order (identical result).
void rcb_par(const float* pos, const int32_t* w, std::vector<int32_t>& idx, int lo, int hi,
             int leaves, std::vector<int32_t>* leaf_start, int par_levels) {
  if (par_levels <= 0 || leaves <= 1 || hi - lo <= 4096) {
    rcb(pos, w, idx, lo, hi, leaves, leaf_start);
    return;
  }
  float mn[2] = {INFINITY, INFINITY}, mx[2] = {-INFINITY, -INFINITY};
  for (int k = lo; k < hi; ++k)
    for (int a = 0; a < 2; ++a) {
      const float p = pos[2 * idx[k] + a];
      mn[a] = std::min(mn[a], p);
      mx[a] = std::max(mx[a], p);
    }
  const int axis = (mx[1] - mn[1] > mx[0] - mn[0]) ? 1 : 0;
  const int l1 = leaves / 2, l2 = leaves - l1;
  const int mid = split_range(pos, w, idx, lo, hi, axis, l1, leaves);
  std::vector<int32_t> right;
  std::thread th([&] { rcb_par(pos, w, idx, mid, hi, l2, &right, par_levels - 1); });
  rcb_par(pos, w, idx, lo, mid, l1, leaf_start, par_levels - 1);
  th.join();
  leaf_start->insert(leaf_start->end(), right.begin(), right.end());
}

// Integer cost density of a tile (x 1024): what one own vertex of the tile "costs" a launch --
// local edges + 2 x local vertices, per own vertex.  Integer so that sums of it are exact.
int32_t tile_weight(const TileDesc& D) {
  const int64_t cost = tile_cost(D);
  return (int32_t)std::max<int64_t>(1, cost * 1024 / std::max(D.n_own, 1));
}

struct TileCfg { int nt, ept, vpt; };
// instantiated kernel configurations (must match kernels.hip)
const TileCfg kCfgs[] = {
    {256, 2, 1},  {256, 3, 1},  {256, 4, 1},  {256, 6, 1},  {256, 4, 2},  {256, 6, 2},
    {512, 2, 1},  {512, 3, 1},  {512, 4, 1},  {512, 6, 1},  {512, 4, 2},  {512, 6, 2},
    {1024, 2, 1}, {1024, 3, 1}, {1024, 4, 1}, {1024, 6, 1}, {1024, 4, 2}, {1024, 6, 2},
};

// ... of the resident kernels, and of the 12-byte-slot kernels (FLAME_PERSIST_CFGS / FLAME_S12_CFGS in kernels.hip)
bool tile_persist_cfg(int nt, int ept, int vpt) { return vpt == 1 && (ept == 2 || ept == 3) && (nt == 256 || nt == 512 || nt == 1024); }
bool tile_slot12_cfg(int nt, int ept, int vpt) { return vpt == 1 && (ept == 2 || ept == 3) && nt == 1024; }

// Lane order inside every block of 64 local edges (the edges one wave handles together in phase D).
// Which edges share a block is fixed by the level / source order; WHICH LANE takes which edge is
// free, and it decides the LDS bank conflicts of the four accesses of phase D (MI355X guide, LDS):
//   ds_read_b128  bar[source], bar[target]: 4 groups of 16 lanes {0-3,12-15,20-27}, {4-11,16-19,
//                 28-31} (+32); float4 index mod 16 is the bank class; equal addresses broadcast;
//   ds_write_b96  source slot, target slot: 8 groups of 8 consecutive lanes; slot index mod 8.
// Greedy, in the sorted order of the block: each edge takes the free lane that adds the fewest
// extra LDS cycles (ties: the lowest lane).  Integer rules only -- the device builder
// (plan_dev.hip, k_tile_pass2) makes the identical choice.
inline int lane_read_group(int lane) {
  const int l = lane & 31;
  const int g = (l < 4 || (l >= 12 && l < 16) || (l >= 20 && l < 28)) ? 0 : 1;
  return g + 2 * (lane >> 5);
}

void assign_lanes(int32_t e_loc, int32_t nslots, std::vector<UInt2>& eij, std::vector<Float4>& ew,
                  std::vector<int32_t>& emap) {
  UInt2 r_eij[64];
  Float4 r_ew[64];
  int32_t r_map[64];
  for (int32_t b0 = 0; b0 < e_loc; b0 += 64) {
    const int c = std::min<int32_t>(64, e_loc - b0);
    // per read group and class: first address seen (+1), per write group and class: stores so far
    int32_t rs_first[4][16] = {}, rt_first[4][16] = {};
    uint8_t ws_cnt[8][8] = {}, wd_cnt[8][8] = {};
    bool used[64] = {};
    for (int k = 0; k < c; ++k) {
      const UInt2 rec = eij[b0 + k];
      const int32_t li = (int32_t)(rec.x & 0xffffu), lj = (int32_t)(rec.x >> 16);
      const uint32_t ss = rec.y & 0xffffu, sd = rec.y >> 16;
      int best = -1, best_cost = 1 << 30;
      for (int lane = 0; lane < c; ++lane) {
        if (used[lane]) continue;
        const int rg = lane_read_group(lane), wg = lane >> 3;
        int cost = 0;
        const int32_t fs = rs_first[rg][li & 15], ft = rt_first[rg][lj & 15];
        if (fs != 0 && fs != li + 1) ++cost;
        if (ft != 0 && ft != lj + 1) ++cost;
        // a lane without a slot stores into its own trash slot (nslots + lane)
        const uint32_t s1 = ss != 0xffffu ? ss : (uint32_t)(nslots + lane);
        const uint32_t s2 = sd != 0xffffu ? sd : (uint32_t)(nslots + lane);
        cost += ws_cnt[wg][s1 & 7] + wd_cnt[wg][s2 & 7];
        if (cost < best_cost) { best_cost = cost; best = lane; if (cost == 0) break; }
      }
      used[best] = true;
      const int rg = lane_read_group(best), wg = best >> 3;
      if (rs_first[rg][li & 15] == 0) rs_first[rg][li & 15] = li + 1;
      if (rt_first[rg][lj & 15] == 0) rt_first[rg][lj & 15] = lj + 1;
      const uint32_t s1 = ss != 0xffffu ? ss : (uint32_t)(nslots + best);
      const uint32_t s2 = sd != 0xffffu ? sd : (uint32_t)(nslots + best);
      ++ws_cnt[wg][s1 & 7];
      ++wd_cnt[wg][s2 & 7];
      r_eij[best] = rec; r_ew[best] = ew[b0 + k]; r_map[best] = emap[b0 + k];
    }
    for (int lane = 0; lane < c; ++lane) { eij[b0 + lane] = r_eij[lane]; ew[b0 + lane] = r_ew[lane]; emap[b0 + lane] = r_map[lane]; }
  }
}

bool pick_cfg(int want_nt, int e_max, int upd_max, TileCfg* out) {
  // pass 0: one vertex and <= 3 edges per thread (most waves to hide LDS latency; measured:
  // 1024 x 2 x 1 beats 512 x 4 x 2 by 4 % at 50 k); pass 1: <= 4 edges per thread (no spills);
  // pass 2: anything that fits.  Within a pass the smallest workgroup wins.
  for (int pass = 0; pass < 3; ++pass)
    for (const TileCfg& c : kCfgs) {
      if (want_nt && c.nt != want_nt) continue;
      if (pass == 0 && (c.ept > 3 || c.vpt > 1)) continue;
      if (pass == 1 && c.ept > 4) continue;
      if ((int64_t)c.nt * c.ept >= e_max && (int64_t)c.nt * c.vpt >= upd_max) {
        *out = c;
        return true;
      }
    }
  return false;
}

}  // namespace

PlanSizing plan_sizing(const PlanOptions& opt, int32_t V, int32_t E) {
  const int64_t lds_cap = opt.lds_bytes;
  // an isolated single tile holds the whole graph when it fits the largest kernel config
  // (LDS of the tile: 16 B per vertex + 16 B per incidence slot; the slot rows of a 64-vertex group
  // share the group's largest degree as pitch, measured 1.1-1.2 x the 2E incidences in degree order:
  // priced at 1.19 x -- a tile that turns out too large after all is rebuilt as a halo'd partition)
  const bool single_fits = V <= 2048 && E <= 6144 && ((int64_t)V * 16 + (int64_t)E * 38 + 1024) <= lds_cap;
  // Auto sizing (measured on MI355X, DESIGN.md "Tile sizing"): one tile per CU when the graph
  // allows it (256 CUs), never below 32 own vertices (halo overhead) or above 196 (LDS / threads);
  // when there are more tiles than CUs prefer the shallower halo whose tiles co-reside on a CU.
  // Beyond one tile per CU (V > 256 * 196) two rounds of fat depth-3 tiles beat four rounds of
  // small ones (200 k vertices: 392 own / depth 3 = 98 k it/s vs 196 / 3 = 71 k it/s).
  const bool one_round = V <= 256 * 196;
  // r05: beyond that, a caller that solves by resident tiles keeps one tile per CU as long as the FAT tiles fit at some depth
  // (measured: 200 k vertices 5.9 us per iteration by 167 launches of two rounds of tiles -> resident)
  // (fat tiles are 256 of them, one per CU -- a device with fewer CUs could never make them resident and would
  // run the shallow fat plan launch by launch: it keeps the r04 partition)
  const bool fat = !one_round && opt.resident && opt.tile_own <= 0 && V <= 256 * 940 && opt.batch_voff.empty() && opt.num_cus >= 256;
  const int fat_own = (V + 255) / 256;
  // (r05, resident tiles: 24 instead of 32 own vertices at least -- more CUs at work, 1-2 % per iteration below 6 k vertices and
  // on a TUM-sized frame: profiles/r05_min_own_ab.txt; tiles that small take the deeper halo)
  const int min_own = opt.resident ? 24 : 32;
  // r06: a resident graph of up to 32 x 40 vertices takes 32 tiles -- all on ONE XCD, hand-offs through its L2 (flame_hip.cpp
  // "one_xcd"; 1.2 k vertices: 50 tiles of 24 over all XCDs 0.89 us per iteration, 32 tiles of 38 on one 0.81)
  const bool one_xcd = opt.one_xcd && opt.resident && opt.tile_own <= 0 && opt.batch_voff.empty() && V > kOneXcdTiles * min_own &&
                       V <= kOneXcdTiles * kOneXcdMaxOwn;
  const int auto_own = one_xcd ? (V + kOneXcdTiles - 1) / kOneXcdTiles
                       : one_round ? std::max(min_own, std::min(196, (V + 255) / 256))
                       : fat     ? fat_own
                                 : std::max(196, std::min(400, (V + 511) / 512));
  // few tiles (a small lone graph): CUs are idle anyway, so redundant halo work is free and deeper
  // halos amortise the per-launch load (1.2 k vertices: depth 8 = 652 k it/s vs depth 4 = 573 k)
  const int auto_tiles = (V + auto_own - 1) / std::max(auto_own, 1);
  // (resident tiles, r04: a round's hand-off costs ~2 us where a launch cost ~3.5: shallower halos win -- depth 5 up to
  // ~100 tiles, 4 above; tools/exp/xpersist_bench.py: 1.2 k / 38 tiles depth 5 / 8 = 0.92 / 1.00 us per iteration, 5 k /
  // 157 tiles depth 4 / 5 = 1.01 / 1.04, 4 k / 125 tiles 1.00 / 1.01)
  // fat tiles, measured (profiles/r05_fat_tiles.txt): two edges per thread on 16-byte slots beat three edges / 12-byte slots
  // at one level deeper (80 k: depth 3 1.62 vs depth 4 1.74 us per iteration; 130 k: depth 2 2.22 vs depth 3 2.49); from ~540
  // own vertices on only the 12-byte layout fits and the deepest halo that does wins (160 k: depth 3 2.79, 2 3.10, 1 3.12)
  const int auto_depth = fat ? (fat_own <= 280 ? 4 : (fat_own <= 420 ? 3 : (fat_own <= 600 ? 2 : 1)))  // (16-byte slots first)
                         : !one_round ? 3
                         : opt.resident ? ((auto_tiles <= 100 || auto_own < 32) ? 5 : 4)
                                        : (auto_tiles <= 64 ? 8 : (auto_tiles <= 160 ? 5 : 4));
  int tile_own = opt.tile_own > 0 ? opt.tile_own : auto_own;
  int depth = opt.tile_depth > 0 ? std::min(opt.tile_depth, kMaxDepth) : auto_depth;
  // Auto: a lone graph is one isolated tile only when it is small (<= single_max = 512 vertices):
  // above that a few dozen depth-4 tiles on as many CUs finish sooner than one CU iterating alone
  // (TUM-sized 1.2 k vertices: 0.32 ms vs 0.44 ms per 200 iterations).  A frame STREAM of such graphs
  // is better off with the single tile all the same (its plan is trivial: 0.88 ms vs 1.00 ms per
  // frame at 1.2 k vertices): option "tile_single_max" up to 2048.  tile_own >= V forces the single
  // tile; batch frames are always single tiles (throughput, one CU per frame).
  bool single = single_fits && (opt.tile_own >= V || (opt.tile_own <= 0 && V <= opt.single_max));
  if (single) { tile_own = std::max(V, 1); depth = 0; }
  PlanSizing sz;
  sz.auto_own = auto_own; sz.auto_depth = auto_depth;
  sz.tile_own = tile_own; sz.depth = depth; sz.single = single;
  sz.fat = fat && !single;
  sz.fat_s12_depth = opt.tile_depth > 0 ? depth : (fat_own <= 640 ? 3 : (fat_own <= 800 ? 2 : 1));
  sz.fallback_own = std::max(196, std::min(400, (V + 511) / 512));
  sz.fallback_depth = opt.tile_depth > 0 ? depth : 3;
  return sz;
}

bool fat_next_attempt(const PlanOptions& opt, const PlanSizing& sz, int* depth, bool* allow_slot12) {
  if (!*allow_slot12) {  // 16-byte slots: one level shallower, then the 12-byte layout from its own first depth
    if (*depth > 1 && opt.tile_depth <= 0) { --*depth; return true; }
    *allow_slot12 = true;
    *depth = sz.fat_s12_depth;
    return true;
  }
  if (*depth > 1 && opt.tile_depth <= 0) { --*depth; return true; }
  return false;
}

bool regular_next_attempt(const PlanOptions& opt, const PlanSizing& sz, int32_t V, int* tile_own, int* depth) {
  const int own = std::max(*tile_own, 1);
  const int64_t ntiles = ((int64_t)V + own - 1) / own;
  const bool sized_for_resident = opt.resident && opt.tile_own <= 0 && opt.tile_depth <= 0 && !sz.single && *tile_own == sz.auto_own &&
                                  ntiles >= 2 && ntiles <= std::min(256, opt.num_cus) && 2 * ntiles > std::min(256, opt.num_cus);
  if (sized_for_resident && *depth > 2) { --*depth; return true; }
  if (sized_for_resident) *depth = sz.depth;  // (no depth fits: smaller tiles by launches, at the automatic depth again)
  return false;
}

TileFit tile_fit(const PlanOptions& opt, bool fat, const std::vector<TileDesc>& tiles, bool allow_slot12) {
  TileFit f;
  int e_max = 0, ext_max = 0, upd_max = 0, hv_max = 0;
  int64_t lds16 = 0, lds12 = 0, stage = 0;
  for (const TileDesc& D : tiles) {
    e_max = std::max(e_max, D.e_loc);
    ext_max = std::max(ext_max, D.n_ext);  // every local vertex gets a register slot ...
    upd_max = std::max(upd_max, D.n_upd);
    hv_max = std::max(hv_max, D.n_ext - D.n_own);
    lds16 = std::max(lds16, tile_lds_bytes(D.n_ext, D.nslots, false));
    lds12 = std::max(lds12, tile_lds_bytes(D.n_ext, D.nslots, true));
    stage = std::max<int64_t>(stage, 16 * (int64_t)std::max(D.n_upd - D.n_own, 0));
  }
  TileCfg c{};
  if (!fat) {
    f.ok = lds16 + kTileLdsReserve <= opt.lds_bytes && pick_cfg(opt.tile_threads, e_max, ext_max, &c);
    // (r06 / ADVICE r05: tiles that are meant to be resident must fit the way the resident launch needs them -- with its
    // staging area, in a configuration that has a resident kernel -- or the caller tries a shallower halo: regular_next_attempt())
    if (f.ok && wants_resident(opt, tiles.size()) && opt.tile_own <= 0 && opt.tile_depth <= 0 && opt.batch_voff.empty() && tiles[0].depth > 0 &&
        (lds16 + stage + kTileLdsReserve > opt.lds_bytes || !tile_persist_cfg(c.nt, c.ept, c.vpt)))
      f.ok = false;
    f.lds_bytes = lds16;
  } else {
    // ... except in fat tiles: a thread per UPDATED vertex and per halo vertex of the poll list is enough (the outermost
    // ring only needs its x_bar in LDS, kernels.hip), the staging area of the resident launch must fit too (a fat partition
    // that cannot be resident is pointless), and 12-byte slots take over where 16 do not fit
    bool cfg_ok = pick_cfg(opt.tile_threads, e_max, ext_max, &c) && tile_persist_cfg(c.nt, c.ept, c.vpt);
    if (!cfg_ok)  // (the lane-less outermost ring is the FAT kernel variants': 1 024 threads)
      cfg_ok = pick_cfg(opt.tile_threads, e_max, std::max(upd_max, hv_max), &c) && tile_slot12_cfg(c.nt, c.ept, c.vpt);
    f.fat = cfg_ok && tile_slot12_cfg(c.nt, c.ept, c.vpt);
    const int64_t margin16 = (!tiles.empty() && tiles[0].depth == 1) ? kFatLdsMarginDepth1 : kFatLdsMargin;
    if (cfg_ok && tile_persist_cfg(c.nt, c.ept, c.vpt) && lds16 + stage + margin16 <= opt.lds_bytes) {
      f.ok = true; f.lds_bytes = lds16;
    } else if (allow_slot12 && cfg_ok && tile_slot12_cfg(c.nt, c.ept, c.vpt) && lds12 + stage + kFatLdsMargin <= opt.lds_bytes) {
      f.ok = true; f.slot12 = true; f.lds_bytes = lds12;
    }
  }
  f.nt = c.nt; f.ept = c.ept; f.vpt = c.vpt;
  return f;
}

bool pick_tile_config(int want_nt, int e_max, int upd_max, int* nt, int* ept, int* vpt) {
  TileCfg c{};
  if (!pick_cfg(want_nt, e_max, upd_max, &c)) return false;
  *nt = c.nt; *ept = c.ept; *vpt = c.vpt;
  return true;
}

int build_plan(const PlanOptions& opt, int32_t V, int32_t E, int32_t T, const float* pos,
               const int32_t* edges, const float* alpha, const float* beta, const int32_t* tris,
               Plan* out) {
  Plan& P = *out;
  P.has_tiles = false;
  P.tile_threads = P.tile_ept = P.tile_vpt = P.tile_depth = 0;
  P.tile_lds_bytes = 0;
  P.tile_slot12 = false;
  P.tile_fat = false;
  P.note.clear();
  P.tiles.clear();
  P.t_vmap.clear(); P.t_emap.clear(); P.t_eij.clear(); P.t_ew.clear(); P.t_srow.clear();
  const bool timing = opt.timing != 0;  // (flame_hip option "plan_timing": the stages on stderr)
  auto tprev = std::chrono::steady_clock::now();
  auto lap = [&](const char* what) {
    if (!timing) return;
    auto now = std::chrono::steady_clock::now();
    std::fprintf(stderr, "[plan] %-14s %7.2f ms\n", what, std::chrono::duration<double, std::milli>(now - tprev).count());
    tprev = now;
  };
  P.V = V; P.E = E; P.T = tris ? T : 0;
  for (int32_t e = 0; e < E; ++e) {
    const int32_t i = edges[2 * e], j = edges[2 * e + 1];
    if (i < 0 || j < 0 || i >= V || j >= V || i == j) return FLAME_HIP_ERR_ARG;
  }
  for (int32_t t = 0; t < P.T; ++t)
    for (int k = 0; k < 3; ++k)
      if (tris[3 * t + k] < 0 || tris[3 * t + k] >= V) return FLAME_HIP_ERR_ARG;

  // ---- tile sizing ----
  const PlanSizing sz = plan_sizing(opt, V, E);
  const int auto_own = sz.auto_own, auto_depth = sz.auto_depth;
  int tile_own = sz.tile_own, depth = sz.depth;
  bool single = sz.single, fat = sz.fat;
  bool fat_s12 = false;  // (fat tiles: the attempts with 16-byte slots come first, fat_next_attempt())
  const bool batch = !opt.batch_voff.empty();
  if (batch) {  // every graph of the batch is one isolated tile; edges must not cross graphs
    const std::vector<int32_t>& vo = opt.batch_voff;
    if (vo.front() != 0 || vo.back() != V) return FLAME_HIP_ERR_ARG;
    for (size_t b = 0; b + 1 < vo.size(); ++b)
      if (vo[b + 1] < vo[b]) return FLAME_HIP_ERR_ARG;
    for (int32_t e = 0; e < E; ++e) {
      const int32_t i = edges[2 * e], j = edges[2 * e + 1];
      const size_t b = std::upper_bound(vo.begin(), vo.end(), i) - vo.begin() - 1;
      if (j < vo[b] || j >= vo[b + 1]) return FLAME_HIP_ERR_ARG;
    }
    single = false;
    depth = 0;
  }

  std::vector<int32_t> deg_o(V, 0);
  for (int32_t e = 0; e < E; ++e) { deg_o[edges[2 * e]]++; deg_o[edges[2 * e + 1]]++; }

  // Cost balancing (second pass): tiles at the image border have much larger halos (long hull
  // edges of the triangulation), and a launch lasts as long as its slowest tile.  After a first
  // unweighted partition every vertex gets the cost density of its tile and the bisection is
  // redone on cost instead of count.
  std::vector<int32_t> vweight;  // integer cost density per vertex (x 1024), see tile_weight()
  bool balanced = false;
  int refine_left = 0;  // set when the balance starts from the tiles (not from a grid)
  auto grid_cell = [&](const float* pp) {
    int c[2];
    for (int a = 0; a < 2; ++a) {
      const float f = (pp[a] - P.wgrid_mn[a]) / std::max(P.wgrid_mx[a] - P.wgrid_mn[a], 1e-20f);
      c[a] = std::max(0, std::min(Plan::kGrid - 1, (int)(f * Plan::kGrid)));
    }
    return c[1] * Plan::kGrid + c[0];
  };
  {
    // frame stream: reuse the cost-density field of the previous frame (same tile count) so the
    // weighted bisection is the only pass
    const int want_tiles = (V + tile_own - 1) / std::max(tile_own, 1);
    if (opt.balance && !batch && !single && !P.wgrid.empty() && P.wgrid_tiles == want_tiles && want_tiles >= 16) {
      vweight.resize(V);
      for (int32_t v = 0; v < V; ++v) vweight[v] = P.wgrid[grid_cell(pos + 2 * v)];
      balanced = true;
    }
  }
  // (fat tiles: every halo depth that does not fit costs a balance sequence of its own before the next one is tried)
  const int max_attempts = batch ? 1 : 9 + kBalanceRefinePasses + (sz.fat ? 7 * (2 + kBalanceRefinePasses) : 0);  // (+ 2: regular_next_attempt())
  for (int attempt = 0; attempt < max_attempts; ++attempt) {
    const int ntiles = batch ? (int)opt.batch_voff.size() - 1 : (V == 0 ? 0 : (V + tile_own - 1) / tile_own);
    // ---- vertex order: RCB leaves = tiles ----
    std::vector<int32_t> idx(V);
    std::iota(idx.begin(), idx.end(), 0);
    std::vector<int32_t> leaf_start;
    if (batch) leaf_start.assign(opt.batch_voff.begin(), opt.batch_voff.end() - 1);
    else if (V > 0) rcb_par(pos, vweight.empty() ? nullptr : vweight.data(), idx, 0, V, ntiles, &leaf_start,
                            opt.host_threads == 1 ? 0 : 3);
    leaf_start.push_back(V);
    // Inside a tile the order is free.  order_mode 0: by degree (lanes of a wave walk incidence
    // lists of similar length); 1: Morton order of the pixel position, so the vertices another
    // tile needs as halo (a strip along the shared border) are contiguous runs in memory and its
    // 16-byte gathers share 64-byte sectors.
    // Isolated tiles (single-tile graphs, batch frames) gather nothing from other tiles: they keep
    // the degree order, whose tighter incidence rows let ~1.2 k-vertex graphs fit one tile's LDS.
    const bool spatial = opt.order_mode == 1 && !single && !batch;
    if (spatial && V > 0) {
      float mn[2] = {INFINITY, INFINITY}, mx[2] = {-INFINITY, -INFINITY};
      for (int32_t v = 0; v < V; ++v)
        for (int a = 0; a < 2; ++a) { mn[a] = std::min(mn[a], pos[2 * v + a]); mx[a] = std::max(mx[a], pos[2 * v + a]); }
      std::vector<uint32_t>& code = P.b_code;
      code.resize(V);
      auto spread = [](uint32_t x) { x &= 0xffff; x = (x | (x << 8)) & 0x00ff00ff; x = (x | (x << 4)) & 0x0f0f0f0f;
                                     x = (x | (x << 2)) & 0x33333333; x = (x | (x << 1)) & 0x55555555; return x; };
      for (int32_t v = 0; v < V; ++v) {
        const uint32_t qx = (uint32_t)(65535.0f * (pos[2 * v] - mn[0]) / std::max(mx[0] - mn[0], 1e-20f));
        const uint32_t qy = (uint32_t)(65535.0f * (pos[2 * v + 1] - mn[1]) / std::max(mx[1] - mn[1], 1e-20f));
        code[v] = spread(qx) | (spread(qy) << 1);
      }
      for (int t = 0; t < ntiles; ++t)
        std::sort(idx.begin() + leaf_start[t], idx.begin() + leaf_start[t + 1],
                  [&](int32_t a, int32_t b) { return code[a] != code[b] ? code[a] < code[b] : a < b; });
    } else {
      // (degree descending, id ascending) by a stable counting pass per tile: the same order as a
      // comparison sort on that key, without its log factor (a frame-per-frame single-tile plan
      // spent a third of its time here)
      int32_t dmax = 0;
      for (int32_t v = 0; v < V; ++v) dmax = std::max(dmax, deg_o[v]);
      std::vector<int32_t>& cnt = P.b_fill;
      std::vector<int32_t>& tmp = P.b_idx;
      tmp.resize(V);
      for (int t = 0; t < ntiles; ++t) {
        const int lo = leaf_start[t], hi = leaf_start[t + 1];
        if (hi - lo < 2) continue;
        if (!std::is_sorted(idx.begin() + lo, idx.begin() + hi))  // ids ascending (a bisection leaf is not; a lone
          std::sort(idx.begin() + lo, idx.begin() + hi);          //  tile / a batch frame already is)
        cnt.assign((size_t)dmax + 2, 0);
        for (int k = lo; k < hi; ++k) cnt[dmax - deg_o[idx[k]] + 1]++;   // bucket 0 = highest degree
        for (int32_t d = 0; d <= dmax; ++d) cnt[d + 1] += cnt[d];
        for (int k = lo; k < hi; ++k) tmp[lo + cnt[dmax - deg_o[idx[k]]]++] = idx[k];
        std::copy(tmp.begin() + lo, tmp.begin() + hi, idx.begin() + lo);
      }
    }
    lap("rcb+degsort");
    P.v_i2o = idx;
    P.v_o2i.assign(V, 0);
    for (int32_t k = 0; k < V; ++k) P.v_o2i[idx[k]] = k;
    std::vector<int32_t> tile_of(V);
    for (int t = 0; t < ntiles; ++t)
      for (int k = leaf_start[t]; k < leaf_start[t + 1]; ++k) tile_of[k] = t;

    // ---- edge order: (owner tile of the source, level 0 before level 1, source vertex,
    // original id).  Edges of one source are adjacent, so the lanes of a wave gather the same or
    // consecutive `bar` entries (broadcast / conflict-free) and write consecutive incidence slots.
    // Two stable counting sorts: by source internal id, then by the 2 * ntiles (tile, level)
    // buckets.
    std::vector<int32_t> eorder(E);
    {
      std::vector<int32_t>& cnt = P.b_fill;
      std::vector<int32_t>& by_src = P.b_estart;
      by_src.resize(E);
      cnt.assign((size_t)V + 1, 0);
      for (int32_t e = 0; e < E; ++e) cnt[P.v_o2i[edges[2 * e]] + 1]++;
      for (int32_t v = 0; v < V; ++v) cnt[v + 1] += cnt[v];
      for (int32_t e = 0; e < E; ++e) by_src[cnt[P.v_o2i[edges[2 * e]]]++] = e;
      cnt.assign(2 * (size_t)ntiles + 1, 0);
      auto bucket = [&](int32_t e) {
        const int32_t ti = tile_of[P.v_o2i[edges[2 * e]]], tj = tile_of[P.v_o2i[edges[2 * e + 1]]];
        return 2 * ti + (ti == tj ? 0 : 1);
      };
      for (int32_t e = 0; e < E; ++e) cnt[bucket(e) + 1]++;
      for (size_t b = 0; b + 1 < cnt.size(); ++b) cnt[b + 1] += cnt[b];
      for (int32_t k = 0; k < E; ++k) { const int32_t e = by_src[k]; eorder[cnt[bucket(e)]++] = e; }
    }
    P.e_i2o = eorder;
    P.e_o2i.assign(E, 0);
    for (int32_t k = 0; k < E; ++k) P.e_o2i[eorder[k]] = k;
    P.eij.resize(E);
    P.ew.resize(E);
    for (int32_t k = 0; k < E; ++k) {
      const int32_t e = eorder[k];
      const int32_t io = edges[2 * e], jo = edges[2 * e + 1];
      P.eij[k] = {P.v_o2i[io], P.v_o2i[jo]};
      const float ds = opt.d_sign < 0 ? -1.0f : 1.0f;  // (a multiplication by +-1 is exact)
      P.ew[k] = {alpha[e], beta[e], ds * (pos[2 * io] - pos[2 * jo]), ds * (pos[2 * io + 1] - pos[2 * jo + 1])};
    }
    lap("edge order");
    // ---- incidence CSR, ascending ORIGINAL edge id per vertex ----
    P.grow.assign(V + 1, 0);
    for (int32_t e = 0; e < E; ++e) { P.grow[P.eij[e].x + 1]++; P.grow[P.eij[e].y + 1]++; }
    for (int32_t v = 0; v < V; ++v) P.grow[v + 1] += P.grow[v];
    P.ginc.assign(2 * (size_t)E, 0);
    {
      std::vector<int32_t> fill(P.grow.begin(), P.grow.end() - 1);
      for (int32_t eo = 0; eo < E; ++eo) {  // original order => each list ascending in original id
        const int32_t k = P.e_o2i[eo];
        P.ginc[fill[P.eij[k].x]++] = k;
        P.ginc[fill[P.eij[k].y]++] = k | (int32_t)0x80000000;
      }
    }
    lap("csr");
    // ---- triangles ----
    P.tris.clear(); P.trow.clear(); P.tinc.clear();
    if (P.T > 0) {
      P.tris.resize(3 * (size_t)P.T);
      for (size_t k = 0; k < P.tris.size(); ++k) P.tris[k] = P.v_o2i[tris[k]];
      P.trow.assign(V + 1, 0);
      for (size_t k = 0; k < P.tris.size(); ++k) P.trow[P.tris[k] + 1]++;
      for (int32_t v = 0; v < V; ++v) P.trow[v + 1] += P.trow[v];
      P.tinc.assign(P.tris.size(), 0);
      std::vector<int32_t> fill(P.trow.begin(), P.trow.end() - 1);
      for (int32_t t = 0; t < P.T; ++t)
        for (int k = 0; k < 3; ++k) P.tinc[fill[P.tris[3 * t + k]]++] = t;
    }
    if (opt.path == FLAME_HIP_PATH_GLOBAL) { P.note = "global path requested"; return 0; }

    lap("triangles");
    // ---- tiles ----
    P.tiles.assign(ntiles, TileDesc());
    P.t_vmap.clear(); P.t_emap.clear(); P.t_eij.clear(); P.t_ew.clear(); P.t_srow.clear();
    bool ok = true;
    // internal edge ranges per owner tile
    std::vector<int32_t> estart(ntiles + 1, 0);
    for (int32_t k = 0; k < E; ++k) estart[tile_of[P.eij[k].x] + 1]++;
    for (int t = 0; t < ntiles; ++t) estart[t + 1] += estart[t];

    // Tiles are independent: build them on a few host threads, each with its own scratch, then
    // concatenate in tile order (so the result does not depend on the thread count).
    typedef TileBuild TileOut;
    std::vector<TileOut>& outs = P.tile_build;
    if ((int)outs.size() < ntiles) outs.resize(ntiles);
    for (int t = 0; t < ntiles; ++t) { outs[t].ok = true; outs[t].note = nullptr; }
    const int nthreads = (int)std::min<int64_t>(opt.host_threads > 0 ? opt.host_threads : 8,
                                                std::max(1, ntiles / 8));
    if ((int)P.scratch.size() < nthreads) P.scratch.resize(nthreads);
    auto build_range = [&](int tid_, int t0, int t1) {
      ThreadScratch& S = P.scratch[tid_];
      S.stamp.assign(V, -1); S.ring.resize(V); S.lidx.resize(V); S.estamp.assign(E, -1); S.eloc.resize(E);
      std::vector<int32_t>&stamp = S.stamp, &ring = S.ring, &lidx = S.lidx, &estamp = S.estamp, &eloc = S.eloc;
      std::vector<int32_t>&ext = S.ext, &frontier = S.frontier, &next = S.next, &lk = S.lk;
      std::vector<uint64_t>& keys = S.keys;
      std::vector<uint16_t>&slot_src = S.slot_src, &slot_dst = S.slot_dst;
      for (int t = t0; t < t1; ++t) {
        TileDesc& D = P.tiles[t];
        TileOut& O = outs[t];
        D.vstart = leaf_start[t];
        D.n_own = leaf_start[t + 1] - leaf_start[t];
        D.depth = depth;
        ext.clear();
        frontier.clear();
        for (int32_t v = D.vstart; v < D.vstart + D.n_own; ++v) {
          stamp[v] = t; ring[v] = 0; lidx[v] = (int32_t)ext.size(); ext.push_back(v); frontier.push_back(v);
        }
        D.ring_end[0] = (int32_t)ext.size();
        for (int r = 1; r <= kMaxDepth; ++r) {
          if (r <= depth) {
            next.clear();
            for (int32_t v : frontier)
              for (int32_t s = P.grow[v]; s < P.grow[v + 1]; ++s) {
                const int32_t k = P.ginc[s] & 0x7fffffff;
                const int32_t u = (P.ginc[s] < 0) ? P.eij[k].x : P.eij[k].y;
                if (stamp[u] != t) { stamp[u] = t; ring[u] = r; next.push_back(u); }
              }
            // inside a ring the order is free: by degree (lanes of a wave walk similar lists),
            // or by internal id (monotone gather addresses: adjacent lanes share sectors)
            if (spatial) std::sort(next.begin(), next.end());
            else
              std::sort(next.begin(), next.end(), [&](int32_t a, int32_t b) {
                const int32_t da = P.grow[a + 1] - P.grow[a], db = P.grow[b + 1] - P.grow[b];
                return da != db ? da > db : a < b;
              });
            for (int32_t u : next) { lidx[u] = (int32_t)ext.size(); ext.push_back(u); }
            frontier.swap(next);
          }
          D.ring_end[r] = (int32_t)ext.size();
        }
        D.n_ext = (int32_t)ext.size();
        D.n_upd = depth == 0 ? D.n_ext : D.ring_end[depth - 1];
        if (D.n_ext > 65535) { O.ok = false; continue; }
        O.vmap.assign(ext.begin(), ext.end());
        // local edges: each ext vertex contributes its outgoing (source-role) incidences;
        // sort key = (level, not-owned, source local id, original id) packed in 64 bits
        keys.clear();
        for (int32_t lv = 0; lv < D.n_ext; ++lv) {
          const int32_t v = ext[lv];
          for (int32_t s = P.grow[v]; s < P.grow[v + 1]; ++s) {
            if (P.ginc[s] < 0) continue;  // v is the target; the source adds it
            const int32_t k = P.ginc[s];
            const int32_t u = P.eij[k].y;
            if (stamp[u] != t) continue;
            const int32_t lvl = std::max(ring[v], ring[u]);
            if (depth > 0 && std::min(ring[v], ring[u]) >= depth) continue;  // feeds no updated vertex
            const uint64_t notown = (lvl <= 1 && ring[v] != 0) ? 1 : 0;
            // [level:5][not owned:1][source local id:16][original id:32]
            keys.push_back(((uint64_t)lvl << 49) | (notown << 48) | ((uint64_t)(uint32_t)lv << 32) |
                           (uint64_t)(uint32_t)P.e_i2o[k]);
          }
        }
        if (!std::is_sorted(keys.begin(), keys.end())) std::sort(keys.begin(), keys.end());  // (an isolated tile emits them in order)
        D.e_loc = (int32_t)keys.size();
        D.estart = estart[t];
        D.e_own = estart[t + 1] - estart[t];
        lk.resize(D.e_loc);
        for (int32_t le = 0; le < D.e_loc; ++le) {
          const int32_t k = P.e_o2i[(int32_t)(keys[le] & 0xffffffffu)];
          lk[le] = k;
          estamp[k] = t;
          eloc[k] = le;
        }
        // owned edges must be exactly the prefix and in internal order
        for (int32_t le = 0; le < D.e_own; ++le)
          if (le >= D.e_loc || lk[le] != D.estart + le) { O.ok = false; O.note = "edge order invariant"; }
        if (!O.ok) continue;
        {
          int32_t le = 0;
          for (int l = 0; l <= kMaxDepth; ++l) {
            while (le < D.e_loc && (int)(keys[le] >> 49) <= l) ++le;
            D.level_end[l] = le;
          }
        }
        O.emap.assign(lk.begin(), lk.end());
        // Incidence slots, one row per vertex with an ODD pitch W per 64-vertex group (the 64
        // vertices a wave updates together): slot(lv, j) = base + (lv - g0) W + j.  Phase P reads
        // column j across lanes (stride W, odd => conflict-free ds_read_b128); phase D writes the
        // consecutive incidences of one source from consecutive lanes.  Within a vertex the
        // incidences keep ascending original edge id (ginc order).
        slot_src.assign(D.e_loc, 0xffff);
        slot_dst.assign(D.e_loc, 0xffff);
        O.srow.clear();
        int32_t base = 0;
        for (int32_t g0 = 0; g0 < D.n_upd && O.ok; g0 += 64) {
          const int32_t g1 = std::min(g0 + 64, D.n_upd);
          int32_t width = 1;
          for (int32_t lv = g0; lv < g1; ++lv) width = std::max(width, P.grow[ext[lv] + 1] - P.grow[ext[lv]]);
          width = slot_group_pitch(width);  // (common.h: the layout's one statement)
          if (!slot_group_fits(base, width)) { O.ok = false; break; }
          for (int32_t lv = g0; lv < g1 && O.ok; ++lv) {
            const int32_t v = ext[lv];
            const int32_t deg = P.grow[v + 1] - P.grow[v];
            const int32_t s0 = slot_row_start(base, lv, width);
            O.srow.push_back(slot_row_word(s0, deg));
            int32_t j = 0;
            for (int32_t s = P.grow[v]; s < P.grow[v + 1]; ++s, ++j) {
              const int32_t k = P.ginc[s] & 0x7fffffff;
              if (estamp[k] != t) { O.ok = false; O.note = "halo closure invariant"; break; }
              const uint16_t slot = (uint16_t)(s0 + j);
              if (P.ginc[s] < 0) slot_dst[eloc[k]] = slot; else slot_src[eloc[k]] = slot;
            }
          }
          base += slot_group_span(width);
        }
        D.nslots = base;
        if (!O.ok) continue;
        O.eij.resize(D.e_loc);
        O.ew.resize(D.e_loc);
        for (int32_t le = 0; le < D.e_loc; ++le) {
          const int32_t k = lk[le];
          const uint32_t li = (uint32_t)lidx[P.eij[k].x], lj = (uint32_t)lidx[P.eij[k].y];
          O.eij[le] = {li | (lj << 16), (uint32_t)slot_src[le] | ((uint32_t)slot_dst[le] << 16)};
          O.ew[le] = P.ew[k];
        }
        if (opt.lane_order == 2) assign_lanes(D.e_loc, D.nslots, O.eij, O.ew, O.emap);
      }
    };
    lap("tiles setup");
    if (nthreads <= 1) {
      build_range(0, 0, ntiles);
    } else {
      std::vector<std::thread> th;
      for (int i = 0; i < nthreads; ++i)
        th.emplace_back(build_range, i, (int)((int64_t)ntiles * i / nthreads),
                        (int)((int64_t)ntiles * (i + 1) / nthreads));
      for (auto& x : th) x.join();
    }
    lap("tiles(par)");
    {
      size_t nv = 0, ne = 0, ns = 0;
      for (int t = 0; t < ntiles && ok; ++t) {
        TileDesc& D = P.tiles[t];
        TileOut& O = outs[t];
        if (!O.ok) { ok = false; if (O.note) P.note = O.note; break; }
        D.vmap_off = (int32_t)nv; D.emap_off = (int32_t)ne; D.erec_off = (int32_t)ne; D.srow_off = (int32_t)ns;
        nv += O.vmap.size(); ne += O.emap.size(); ns += O.srow.size();
      }
      if (ok) {
        P.t_vmap.resize(nv); P.t_emap.resize(ne); P.t_eij.resize(ne); P.t_ew.resize(ne); P.t_srow.resize(ns);
        for (int t = 0; t < ntiles; ++t) {
          const TileDesc& D = P.tiles[t];
          const TileOut& O = outs[t];
          std::copy(O.vmap.begin(), O.vmap.end(), P.t_vmap.begin() + D.vmap_off);
          std::copy(O.emap.begin(), O.emap.end(), P.t_emap.begin() + D.emap_off);
          std::copy(O.eij.begin(), O.eij.end(), P.t_eij.begin() + D.erec_off);
          std::copy(O.ew.begin(), O.ew.end(), P.t_ew.begin() + D.erec_off);
          std::copy(O.srow.begin(), O.srow.end(), P.t_srow.begin() + D.srow_off);
        }
      }
    }
    lap("concat");
    const bool tiles_valid = ok;
    TileFit fit;
    if (ok) { fit = tile_fit(opt, fat, P.tiles, fat_s12); ok = fit.ok; }
    // only the LARGEST tile decides whether a partition fits, and before the cost balance that is a
    // border tile: balance first, shrink only if the balanced partition does not fit either
    // (same rule in flame_hip.cpp upload_device_plan)
    if (!ok && tiles_valid && opt.balance && !balanced && !batch && !single && ntiles >= 16) {
      balanced = true;
      refine_left = kBalanceRefinePasses;
      vweight.assign(V, 1);
      for (int t = 0; t < ntiles; ++t) {
        const TileDesc& D = P.tiles[t];
        const int32_t wt = tile_weight(D);
        for (int32_t k = D.vstart; k < D.vstart + D.n_own; ++k) vweight[P.v_i2o[k]] = wt;
      }
      continue;
    }
    // Refinement passes of the cost balance (first upload only; a frame stream refines through the
    // cost-density grid from frame to frame): every vertex weight is scaled by its tile's cost over
    // the mean tile cost and the bisection is redone.  Integer arithmetic, see plan_dev.hip
    // k_weights_scale.  Border tiles are halo-dominated (long hull edges), so the slowest tile stays
    // ~1.2 x the mean; two passes take max/mean from 1.35 to 1.20 at 50 k vertices (+3.5 % it/s).
    if (ok && balanced && refine_left > 0 && !vweight.empty() && !batch && !single && ntiles >= 16) {
      --refine_left;
      int64_t total = 0;
      for (int t = 0; t < ntiles; ++t) total += tile_cost(P.tiles[t]);
      for (int t = 0; t < ntiles; ++t) {
        const TileDesc& D = P.tiles[t];
        const int64_t cost = tile_cost(D);
        for (int32_t k = D.vstart; k < D.vstart + D.n_own; ++k) {
          int32_t& w = vweight[P.v_i2o[k]];
          w = (int32_t)std::min<int64_t>(1 << 28, std::max<int64_t>(1, (int64_t)w * cost * ntiles / std::max<int64_t>(total, 1)));
        }
      }
      lap("refine weights");
      continue;
    }
    if (ok && opt.balance && !balanced && !batch && !single && ntiles >= 16) {
      balanced = true;
      refine_left = kBalanceRefinePasses;
      vweight.assign(V, 1);
      for (int t = 0; t < ntiles; ++t) {
        const TileDesc& D = P.tiles[t];
        const int32_t wt = tile_weight(D);
        for (int32_t k = D.vstart; k < D.vstart + D.n_own; ++k) vweight[P.v_i2o[k]] = wt;
      }
      lap("balance weights");
      continue;  // rebuild with weighted bisection (same tile count)
    }
    if (ok) {
      P.has_tiles = true;
      P.tile_threads = fit.nt; P.tile_ept = fit.ept; P.tile_vpt = fit.vpt;
      P.tile_depth = depth;
      P.tile_lds_bytes = fit.lds_bytes;
      P.tile_slot12 = fit.slot12;
      P.tile_fat = fit.fat;
      if (opt.balance && !batch && !single && ntiles >= 16) {  // remember the cost-density field
        for (int a = 0; a < 2; ++a) { P.wgrid_mn[a] = INFINITY; P.wgrid_mx[a] = -INFINITY; }
        for (int32_t v = 0; v < V; ++v)
          for (int a = 0; a < 2; ++a) {
            P.wgrid_mn[a] = std::min(P.wgrid_mn[a], pos[2 * v + a]);
            P.wgrid_mx[a] = std::max(P.wgrid_mx[a], pos[2 * v + a]);
          }
        // integer sums: the same field whatever the accumulation order (host loop / device atomics)
        std::vector<int64_t> sum(Plan::kGrid * Plan::kGrid, 0), cnt(Plan::kGrid * Plan::kGrid, 0);
        int64_t total = 0;
        for (int t = 0; t < ntiles; ++t) {
          const TileDesc& D = P.tiles[t];
          const int32_t wt = tile_weight(D);
          for (int32_t k = D.vstart; k < D.vstart + D.n_own; ++k) {
            const int c = grid_cell(pos + 2 * P.v_i2o[k]);
            sum[c] += wt; cnt[c] += 1;
          }
          total += (int64_t)wt * D.n_own;
        }
        const int32_t mean = V > 0 ? (int32_t)std::max<int64_t>(1, total / V) : 1024;
        P.wgrid.resize(sum.size());
        for (size_t c = 0; c < sum.size(); ++c) P.wgrid[c] = cnt[c] > 0 ? (int32_t)(sum[c] / cnt[c]) : mean;
        P.wgrid_tiles = ntiles;
      }
      return 0;
    }
    vweight.clear();  // a failed weighted pass falls back to plain bisection with smaller tiles
    balanced = false;
    refine_left = 0;
    P.wgrid.clear();
    // did not fit: shrink the tiles (a single tile becomes a halo'd partition) and retry
    P.tiles.clear();
    if (single && opt.single_only) { P.has_tiles = false; return kPlanSingleNoFit; }
    if (single) { single = false; tile_own = opt.tile_own > 0 ? opt.tile_own : auto_own; depth = opt.tile_depth > 0 ? std::min(opt.tile_depth, kMaxDepth) : auto_depth; }
    else if (fat && fat_next_attempt(opt, sz, &depth, &fat_s12)) {}  // fat tiles: shallower / 12-byte slots first ...
    else if (fat) { fat = false; tile_own = sz.fallback_own; depth = sz.fallback_depth; }  // ... then two rounds of smaller tiles
    else if (regular_next_attempt(opt, sz, V, &tile_own, &depth)) {}  // resident-sized regular tiles: a shallower halo first
    else tile_own = std::max(16, tile_own / 2);
  }
  P.has_tiles = false;
  if (P.note.empty()) P.note = "no tile configuration fits";
  return 0;
}

}  // namespace flamehip
