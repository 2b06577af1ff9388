// flame_ros_amd/csrc/plan_dev.h -- the graph plan built ON THE GPU (graph
// construction upstream of the regulariser; the reference budgets it per frame, stat keys
// sync_graph / triangulate, reference msg/FlameStats.msg:43-44, because every frame is a new graph,
// reference src/flame_offline_tum.cc:578).
//
// Same contract as the host builder (plan.h / plan.cpp): identical arrays, element for element --
// the partition (recursive coordinate bisection on the total order (coordinate, id), integer cost
// weights), the Morton order inside tiles, the edge order, the incidence CSR in ascending original
// edge id, the per-tile halo rings, local edge order, incidence slots.  Every rule is either
// integer arithmetic or single IEEE float operations, so host and device agree bit for bit
// (tests/test_gpu_plan_device.py compares them array by array).
//
// Covers the product case: halo tiles (depth >= 1) in spatial order.  Isolated single tiles, batches
// of frames, the degree order and graphs beyond the LDS bitmap are built by the host builder.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <functional>
#include <vector>

#include "common.h"
#include "plan.h"

namespace flamehip {

int test_alloc_fill();  // flame_hip.cpp: the byte new device allocations are filled with in the hooks library, else -1

// Device arrays of a plan.  Owned by the caller (the handle), sized by the caller:
// V / E / T sized arrays before build, the tile arrays through `alloc_tiles` once their sizes are
// known (after the first per-tile pass).
struct DevPlanArrays {
  int32_t* v_o2i = nullptr;   // V
  int32_t* v_i2o = nullptr;   // V
  int32_t* e_o2i = nullptr;   // E
  int32_t* e_i2o = nullptr;   // E
  int2* eij = nullptr;        // E
  float4* ew = nullptr;       // E
  int32_t* grow = nullptr;    // V + 1
  int32_t* ginc = nullptr;    // 2E
  int32_t* tris = nullptr;    // 3T internal vertex ids
  int32_t* trow = nullptr;    // V + 1
  int32_t* tinc = nullptr;    // 3T
  TileDesc* tiles = nullptr;  // ntiles
  int32_t* t_vmap = nullptr;
  int32_t* t_emap = nullptr;
  uint2* t_eij = nullptr;
  float4* t_ew = nullptr;
  uint32_t* t_srow = nullptr;
};

struct DevPlanInputs {  // device copies of the caller's arrays, ORIGINAL order
  const float2* pos = nullptr;   // V
  const int2* edges = nullptr;   // E
  const float* alpha = nullptr;  // E
  const float* beta = nullptr;   // E
  const int32_t* tris = nullptr; // 3T or nullptr
};

class DevPlanner {
 public:
  DevPlanner() = default;
  ~DevPlanner();
  DevPlanner(const DevPlanner&) = delete;
  DevPlanner& operator=(const DevPlanner&) = delete;

  // true when this graph / option set is handled on the device
  static bool eligible(const PlanOptions& opt, int32_t V, int32_t E, int32_t T, int tile_own, int depth,
                       bool single, int64_t lds_bytes);

  // Builds the plan for `ntiles` tiles of halo depth `depth`.  `arrays` must hold the V/E/T sized
  // buffers; alloc_tiles(nv, ne, ns) is called once (after a stream sync) and must fill the tile
  // array pointers of `arrays` for those element counts.  On success *tiles_host receives the tile
  // descriptors and *ok tells whether every tile could be built (false = the caller retries with
  // smaller tiles or falls back to the host builder).  user_flags_dev (optional): TWO device words of
  // the caller (a non-finite-input flag; the derived edge count) copied to user_flags_host[0..1] with the builder's first
  // sync; a non-zero word ends the build early (ok = false).  after_partition (optional) is called
  // once stages A and B -- which read only `in.pos` -- are enqueued: the caller stages its other
  // arrays there, so the host-side copies overlap the partition kernels.  Returns a hipError_t.
  typedef int (*AllocTilesFn)(void* ctx, size_t ntiles, size_t nv, size_t ne, size_t ns);
  hipError_t build(hipStream_t s, const PlanOptions& opt, int32_t V, int32_t E, int32_t T, int ntiles,
                   int depth, const DevPlanInputs& in, DevPlanArrays* arrays, AllocTilesFn alloc_tiles,
                   void* alloc_ctx, std::vector<TileDesc>* tiles_host, bool* ok, bool* index_error,
                   const int32_t* user_flags_dev = nullptr, int32_t* user_flags_host = nullptr,
                   const std::function<hipError_t()>& after_partition = nullptr);

  // Graph sync on the device (row a7): unique undirected edges (i < j, lexicographic) of a
  // triangulation + alpha = 1 / |pos_i - pos_j|; edges / alpha need 3T entries.  Synchronises (E).
  // nan_flag (optional device word): bit 0 is set when a derived value is not finite (the upload's
  // non-finite-input check, done where the values are made instead of by launches of its own)
  hipError_t edges_from_tris(hipStream_t s, int32_t V, int32_t T, const int32_t* tris, const float2* pos,
                             int2* edges, float* alpha, int32_t* E_out, bool* index_error, int32_t* nan_flag = nullptr,
                             const std::function<void()>& while_running = nullptr, int32_t expected_E = -1,
                             const std::function<hipError_t()>& before_positions = nullptr);
  // before_positions: called once the kernels that read ONLY the triangles are enqueued and before the
  // first one that reads `pos` -- a caller that still has to stage the positions does it there.
  // With expected_E >= 0 the chain's own flags (bit 2 bad index, bit 32 look-back timeout) go to
  // nan_flag[2], which build() reads through its 4 user-flag words.
  // expected_E >= 0 (needs nan_flag): do not wait for the count -- *E_out = expected_E, the true count is
  // written to nan_flag[1]; expect_edges(E) makes the NEXT build() check it at its first synchronisation
  // (user_flags_host[1] then holds the true count; a mismatch ends the build with ok = false)
  void expect_edges(int32_t E) { expect_E_ = E; }
  // Small frames of a graph sync (mini_eligible(), a reused partition, an expected edge count): the NEXT
  // build() derives the edges and data terms itself and does everything in front of its tile pass in
  // one launch of one workgroup (k_mini_plan) -- the caller stages the inputs and does NOT call
  // edges_from_tris / sync_data.  mini_used() tells whether that build took the offer; if not (no
  // reuse after all), nothing was derived and the caller goes the usual way.
  struct MiniSync {
    const float* mu; const float* var; const float* pred; float scale; int adaptive, init_pred;
    const int32_t* tris; const float2* pos;  // where the launch reads triangles / positions (null: the build's inputs;
                                             // the positions are copied to the build's input array either way)
    float* z; float* wgt; float* x0; int2* edges; float* alpha; int32_t* dflags;
  };
  static bool mini_eligible(int32_t V, int32_t T, int32_t E) { return V >= 2 && V <= 2048 && T >= 1 && T <= 4096 && E >= 1 && E <= 6144; }
  void offer_mini(const MiniSync& m) { mini_ = m; mini_set_ = true; mini_used_ = false; }
  void withdraw_mini() { mini_set_ = false; }
  bool mini_used() const { return mini_used_; }
  // flags[0] |= 1 when any value of the (up to five) arrays is not finite; null arrays are skipped
  static hipError_t check_finite(hipStream_t s, int32_t* flags, const float* a, int64_t na, const float* b = nullptr,
                                 int64_t nb = 0, const float* c = nullptr, int64_t nc = 0, const float* d = nullptr,
                                 int64_t nd = 0, const float* e = nullptr, int64_t ne = 0);
  // z = mu / scale, wgt = 1 or 1 / var, x0 = prediction / scale where finite (else z)
  hipError_t sync_data(hipStream_t s, int32_t V, const float* mu, const float* var, const float* pred,
                       float scale, int adaptive, int init_pred, float* z, float* wgt, float* x0,
                       int32_t* nan_flag = nullptr);

  // weights for the next build: none / from the tiles of the last build / from the cost-density grid
  void set_weights_none() { weight_mode_ = 0; }
  void set_weights_from_grid() { weight_mode_ = 2; }
  // weights of the next build from the tiles of the last one: their cost density (second pass),
  // or the current weights scaled by tile cost / mean tile cost (refinement passes)
  hipError_t weights_from_tiles(hipStream_t s, int32_t V, const DevPlanArrays& arrays);
  hipError_t weights_scale_by_tiles(hipStream_t s, int32_t V, int ntiles, long long total_cost,
                                    const DevPlanArrays& arrays);
  // records the cost-density grid of the last build (device side, integer); grid_tiles() then
  // returns the tile count it was made for
  hipError_t update_grid(hipStream_t s, int32_t V, int ntiles, const DevPlanInputs& in, const DevPlanArrays& arrays);
  // update_grid() works on the builder's second stream, beside the caller's iterations: host-side wait
  // before anything it reads (positions, tile descriptors, vertex order) or writes is touched again
  hipError_t wait_maps();
  // after (optional): the maps' launches wait for this event -- the end of a solve by RESIDENT tiles, which owns every CU:
  // beside it the map kernels crawl (k_grid_accum 80 us instead of 9 at 50 k) and slow the tiles they share CUs with
  hipError_t flush_grid(hipEvent_t after = nullptr);  // enqueue update_grid()'s launches now (no-op when they are out already)
  int grid_tiles() const { return grid_tiles_; }
  void drop_grid() { grid_tiles_ = 0; }

  // Partition reuse on a frame stream (plan_dev.hip "Partition REUSE"): update_grid() also records
  // the tile of every vertex in a pyramid of spatial cells; while the next frames hold about as many
  // vertices (within 1/8) at the same halo depth, reuse_partition() makes the NEXT build() read its
  // partition from that map (map_tiles() tiles) instead of bisecting.  A build that rejects the
  // reused partition drops the map (the caller then bisects as usual).
  int map_tiles() const { return map_tiles_; }
  bool map_usable(int32_t V, int depth) const {
    return map_tiles_ >= 2 && depth == map_depth_ && V >= 2 * map_tiles_ &&
           8ll * (V > map_V_ ? V - map_V_ : map_V_ - V) <= map_V_;
  }
  void reuse_partition() { reuse_next_ = true; }
  void set_map_depth(int depth) { map_depth_ = depth; }
  void drop_map() { map_tiles_ = 0; }
  bool last_build_reused() const { return last_reused_; }

 private:
  hipError_t reserve(int32_t V, int32_t E, int32_t T, int ntiles);
  hipError_t scan_i32(hipStream_t s, int lane, const int32_t* in, int32_t* out, int64_t n, bool inclusive,
                      void* cub_tmp, size_t cub_bytes);
  void release();

  int weight_mode_ = 0;
  int grid_tiles_ = 0;
  int map_tiles_ = 0, map_depth_ = 0;
  int32_t map_V_ = 0;
  bool reuse_next_ = false, last_reused_ = false;
  int32_t expect_E_ = -1;  // >= 0: the build was launched on a predicted edge count (checked at its first sync)
  int32_t spec_nv_ = 0, spec_ne_ = 0, spec_ns_ = 0;  // tile-array totals (+1/8) of the previous build: the next
  int spec_tiles_ = 0;                                //   build allocates for them and skips a host round trip
  int32_t* cell_pyr_ = nullptr;    // tile + 1 of the vertices in every cell of a 128/32/8/1 pyramid
  bool use_subtree_ = true;  // deep bisection levels in one LDS kernel
  int sub_extra_levels_ = 0; // hand-over level pushed down after a subtree overflow
  bool attr_set_ = false, sub_attr_set_ = false;  // dynamic-LDS opt-in done on this handle's device
  // capacities
  int64_t capV_ = 0, capE_ = 0, capT_ = 0, capTiles_ = 0;
  size_t cub_bytes_ = 0;
  // scratch (device)
  void* cub_tmp_ = nullptr;
  uint64_t* keys_a_ = nullptr;   // max(V, 2E, 3T)
  uint64_t* keys_b_ = nullptr;
  uint32_t* vals_a_ = nullptr;   // 2E
  uint32_t* vals_b_ = nullptr;
  int32_t* seg_pos_ = nullptr;   // V
  int32_t* tile_of_int_ = nullptr;  // V (by internal id)
  int32_t* w_int_ = nullptr;     // V (by original id)
  long long* wsort_ = nullptr;   // V
  long long* wscan_ = nullptr;   // V
  int32_t* counts_ = nullptr;    // V + 2 (degree / triangle counts)
  int32_t* rank_ = nullptr;      // 2E + 3T places inside the counting CSRs' rows (what the counting atomics returned)
  int32_t* rank_tri_ = nullptr;  // = rank_ + 2 capE_
  int32_t* reuse_cnt_ = nullptr; // tile counters of the partition-reuse pass, one per 128-byte line
  int32_t* gadj_ = nullptr;      // 2E: the vertex at the other end of every incidence (ring search of the tile passes)
  int32_t* ipos_ = nullptr;      // 2E: place of (internal edge, role) in its vertex's incidence row (slots)
  int32_t* seg_tab_ = nullptr;   // segment tables + bbox + mids (see plan_dev.hip)
  int32_t* estart_ = nullptr;    // ntiles + 1
  int32_t* tile_ext_ = nullptr;  // ntiles * kCapExt (pass-1 vertex lists)
  int32_t* tile_meta_ = nullptr; // ntiles * kMetaWords (pass-1 counts, ring ends, offsets)
  int32_t* flags_ = nullptr;     // error / totals words
  // look-back state of the single-launch scans (plan_dev.hip "Single-launch prefix sums"): one set per
  // stream the builder scans on; a flag matches only the launch (epoch) that wrote it
  unsigned long long* scan_agg_[2] = {nullptr, nullptr};
  uint32_t* scan_flag_[2] = {nullptr, nullptr};
  uint32_t scan_epoch_[2] = {0, 0};
  long long* grid_sum_ = nullptr;  // kGrid^2
  int32_t* grid_cnt_ = nullptr;    // kGrid^2
  int32_t* grid_w_ = nullptr;      // kGrid^2
  float* grid_bounds_ = nullptr;   // mn.x mn.y mx.x mx.y of the frame the grid was made from
  float* gbbox_ = nullptr;         // global bbox of the current frame (4 floats)
  // stage E (vertex -> triangle CSR) runs beside the edge stages on a stream of its own, with its
  // own sort scratch
  char* hpin_ = nullptr;         // page-locked landing area of the builder's D2H copies (flags, descriptors)
  size_t hpin_bytes_ = 0;
  hipStream_t s2_ = nullptr;
  hipEvent_t ev_fork_ = nullptr, ev_join_ = nullptr, ev_grid_ = nullptr;
  bool grid_pending_ = false;    // update_grid()'s kernels may still run on s2_
  bool grid_deferred_ = false;   // ... or have not been enqueued yet (flush_grid())
  struct GridJob { hipStream_t stream = nullptr; int32_t V = 0; const float2* pos = nullptr; const int32_t* v_i2o = nullptr;
                   const TileDesc* tiles = nullptr; } grid_job_;
  MiniSync mini_{};
  bool mini_set_ = false, mini_used_ = false;
  int64_t capV2_ = 0;
  size_t tcub_bytes_ = 0;
  void* tcub_tmp_ = nullptr;     // scan scratch of the triangle stage
  int32_t* tcnt_ = nullptr;      // V + 1 counts (2V + 2 allocated)
};

// Conflict-avoiding lane order applied to a finished plan on the device (plan.h PlanOptions::
// lane_order = 1): every 64-edge block of every tile, in place.
hipError_t launch_assign_lanes(hipStream_t s, int32_t ntiles, int32_t e_max, const TileDesc* tiles, uint2* t_eij,
                               float4* t_ew, int32_t* t_emap, bool slot12 = false);

}  // namespace flamehip
