// flame_ros_amd/csrc/plan.h -- host-side graph plan: locality reordering, CSR incidence lists and
// the partition of the Delaunay vertex graph into LDS-resident tiles with depth-D halos.
//
// Upstream builds a Boost adjacency_list per frame inside Flame::update (SURVEY.md 8a rows a1,
// a7; results visible at reference src/flame_offline_tum.cc:628-635).  Here the same graph is
// flattened into the arrays of common.h.  Pure C++ (no HIP) so it is testable without a GPU.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

#include "common.h"

namespace flamehip {

struct PlanOptions {
  int path = 0;          // FLAME_HIP_PATH_*
  int tile_own = 0;      // target own vertices per tile (0 = auto)
  int tile_depth = 0;    // halo depth (0 = auto)
  int tile_threads = 0;  // workgroup size (0 = auto)
  int64_t lds_bytes = 160 * 1024;
  int host_threads = 0;  // plan-build threads (0 = up to 8)
  int balance = 1;       // second, cost-weighted bisection pass (equalises tile cost)
  int order_mode = 1;    // vertex order inside tiles / rings: 0 by degree, 1 spatial (gather locality)
  // lanes inside every 64-edge block assigned against LDS bank conflicts (+2 % iterations/s, but
  // 60 us of GPU time at 50 k vertices, more on the host): 0 never; 1 when a plan is solved a
  // SECOND time (a frame stream that solves every graph once never pays for it); 2 at build time
  int lane_order = 1;
  int d_sign = 1;        // [UPSTREAM-RECALL] switch: the edge vector is d_sign * (pos_i - pos_j)
  bool resident = false; // the caller solves by ONE launch of resident tiles (flame_hip option "persist"): the automatic halo
                         // depth is tuned for round hand-offs (~2 us per round) instead of kernel boundaries + reloads
  int single_max = 512;  // auto: a lone graph up to this many vertices becomes ONE isolated tile
  bool single_only = false;  // build_plan(): return kPlanSingleNoFit instead of falling back to a
                             // halo'd partition when the isolated tile does not fit after all
  bool one_xcd = false;  // the caller keeps graphs of <= kOneXcdTiles resident tiles on ONE XCD (flame_hip option "one_xcd"): sized for it
  int num_cus = 256;     // compute units of the device the plan will run on (a host-only plan: MI355X)
  int timing = 0;        // diagnostic (option "plan_timing"): the builders print their stages' times to stderr (levels: plan_dev.hip)
  int debug_sub_cap = 0; // test hook: the device builder's subtree kernel reports an overflow above
                         // this many vertices on its first try (exercises the recovery path)
  // batch of independent graphs (frames axis): nb + 1 vertex offsets; graph b = one isolated tile
  std::vector<int32_t> batch_voff;
};

constexpr int kOneXcdTiles = 32;     // an XCD has 32 CUs: one resident tile each
constexpr int kOneXcdMaxOwn = 40;    // ... worth it up to 32 x 40 vertices (1.5 k vertices on 32 tiles of 47: +-0 against 63 tiles of 24)
constexpr int kPlanSingleNoFit = 1;  // build_plan() with single_only: the caller partitions elsewhere

struct Float4 { float x, y, z, w; };
struct Int2 { int32_t x, y; };
struct UInt2 { uint32_t x, y; };

// Per-tile and per-thread build buffers.  They live in the Plan so that a handle re-uploaded every
// frame keeps their capacity (no page faults / allocator traffic after the first frame).
struct TileBuild {
  std::vector<int32_t> vmap, emap;
  std::vector<UInt2> eij;
  std::vector<Float4> ew;
  std::vector<uint32_t> srow;
  bool ok = true;
  const char* note = nullptr;
};
struct ThreadScratch {
  std::vector<int32_t> stamp, ring, lidx, estamp, eloc, ext, frontier, next, lk;
  std::vector<uint64_t> keys;
  std::vector<uint16_t> slot_src, slot_dst;
};

struct Plan {
  int32_t V = 0, E = 0, T = 0;
  // permutations
  std::vector<int32_t> v_o2i, v_i2o, e_o2i, e_i2o;
  // global-path arrays (internal order)
  std::vector<Int2> eij;
  std::vector<Float4> ew;
  std::vector<int32_t> grow, ginc;
  // triangle stage: internal vertex ids per triangle (original triangle order) + vertex->tri CSR
  std::vector<int32_t> tris;
  std::vector<int32_t> trow, tinc;
  // tile plan
  bool has_tiles = false;
  int tile_threads = 0, tile_ept = 0, tile_vpt = 0, tile_depth = 0;
  int64_t tile_lds_bytes = 0;
  bool tile_fat = false;     // fat tiles on a 1 024-thread configuration: the resident launch takes the FAT kernel variants
  bool tile_slot12 = false;  // 12-byte incidence slots (fat tiles; kernels.hip SlotMem<true>): tile_lds_bytes is priced that way
  std::vector<TileDesc> tiles;
  std::vector<int32_t> t_vmap, t_emap;
  std::vector<UInt2> t_eij;
  std::vector<Float4> t_ew;
  std::vector<uint32_t> t_srow;
  std::string note;  // why the tile path was not built, if so
  // built by the device builder (plan_dev.hip): the arrays above live on the GPU only (the handle
  // copies them back on demand); `tiles` and the scalars are valid on the host
  bool on_device = false;
  // build buffers (persistent capacity)
  std::vector<TileBuild> tile_build;
  std::vector<ThreadScratch> scratch;
  std::vector<int32_t> b_idx, b_leaf, b_tile_of, b_deg, b_estart, b_fill;
  std::vector<int64_t> b_keys;
  std::vector<uint32_t> b_code;
  // cost-density field of the last balanced partition on a coarse pixel grid: a handle that is
  // re-uploaded every frame balances the next frame in ONE weighted pass instead of two
  static constexpr int kGrid = 32;
  std::vector<int32_t> wgrid;  // kGrid * kGrid integer cost densities (x 1024), empty = none yet
  float wgrid_mn[2] = {0.f, 0.f}, wgrid_mx[2] = {1.f, 1.f};
  int32_t wgrid_tiles = 0;   // tile count the field was built for
};

// cost-balance refinement passes after the first weighted bisection (first upload of a handle only)
constexpr int kBalanceRefinePasses = 3;  // (r05 sweep, resident tiles: 10 k 1.054 -> 1.019 us per iteration at 3, 50 k +-0; profiles/r05_refine_passes.txt)

// Tile sizing shared by the host and the device builder (measured on MI355X).
struct PlanSizing {
  int auto_own = 0, auto_depth = 0;  // what "auto" resolves to
  int tile_own = 0, depth = 0;       // first attempt (a single isolated tile: tile_own = V, depth 0)
  bool single = false;
  // r05 FAT resident tiles: a graph beyond 256 x 196 vertices still gets ONE tile per CU (V / 256 own vertices each, up to
  // ~940) so that it can be solved by one launch of resident tiles; the halo gets shallower as the tiles grow and the
  // incidence slots shrink to 12 bytes where 16 do not fit (tile_fit()).  A partition that fits at no depth falls back to
  // fallback_own / fallback_depth (two rounds of smaller tiles, solved by launches).
  bool fat = false;
  int fallback_own = 0, fallback_depth = 0;
  // the attempts of a fat partition, in this order: 16-byte slots from `depth` down to 1 (measured: they beat the 12-byte
  // layout even one or two levels shallower -- 145 k vertices depth 2 / 16 B 2.27 us per iteration vs depth 3 / 12 B 2.66,
  // 160 k depth 1 / 16 B 2.61 vs 2.75, profiles/r05_fat_tiles.txt), then 12-byte slots from fat_s12_depth down to 1
  int fat_s12_depth = 0;
};
// what one build attempt produced, priced: does it fit (LDS, a kernel configuration), and how
struct TileFit {
  bool ok = false;
  int nt = 0, ept = 0, vpt = 0;
  bool slot12 = false;
  bool fat = false;  // fat sizing and a 1 024-thread configuration (the FAT kernel variants exist for those)
  int64_t lds_bytes = 0;
};
// e_max / ext_max / upd_max / hv_max: the largest tile's local edges, local vertices, updated vertices, halo vertices;
// lds16 / lds12: max over the tiles of tile_lds_bytes() (+ the resident tiles' staging area when `resident`)
TileFit tile_fit(const PlanOptions& opt, bool fat, const std::vector<TileDesc>& tiles, bool allow_slot12 = true);
// what a fat partition that did not fit tries next (both plan builders): false = nothing left, take the fallback
bool fat_next_attempt(const PlanOptions& opt, const PlanSizing& sz, int* depth, bool* allow_slot12);
// r06, regular tiles sized for ONE resident launch (automatic sizes, at most one tile per CU) that do not fit -- LDS with the
// resident launch's staging area, a kernel configuration that has a resident variant --: halving the tiles would put more
// tiles than CUs on the chip and the solve on launches (4 x slower), so the same tiles are first tried one halo level
// shallower, down to depth 2 (the subdomains of a partitioned 200 k graph: 46-48 k local vertices with ragged halo bands,
// half of them fell back to 511-515 tiles or to four edges per thread, profiles/r06_partition_parts_resident.txt).  Returns
// true when *depth was lowered; else the caller halves *tile_own (and this call has restored the automatic depth).
bool regular_next_attempt(const PlanOptions& opt, const PlanSizing& sz, int32_t V, int* tile_own, int* depth);
// the tiles of a partition are meant to be resident: the caller solves that way and there is at most one tile per CU
inline bool wants_resident(const PlanOptions& opt, size_t ntiles) { return opt.resident && ntiles >= 2 && (int64_t)ntiles <= std::min(256, opt.num_cus); }
PlanSizing plan_sizing(const PlanOptions& opt, int32_t V, int32_t E);
// smallest instantiated kernel configuration that holds e_max local edges / upd_max local vertices
bool pick_tile_config(int want_nt, int e_max, int upd_max, int* nt, int* ept, int* vpt);

// Builds the plan.  Returns 0 or a FLAME_HIP_ERR_* code (bad indices).
int build_plan(const PlanOptions& opt, int32_t V, int32_t E, int32_t T, const float* pos,
               const int32_t* edges, const float* alpha, const float* beta, const int32_t* tris,
               Plan* out);

}  // namespace flamehip
