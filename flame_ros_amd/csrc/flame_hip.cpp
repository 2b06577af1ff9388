// flame_ros_amd/csrc/flame_hip.cpp -- C ABI of libflame_hip.so (include/flame_hip.h): handle
// management, upload (plan build + H2D), the PD-iteration driver (global path / tile path,
// optional hipGraph replay), costs, triangle stage, download.
//
// Boundary: replaces the regulariser loop and per-triangle stage inside flame::Flame::update()
// (reference call sites src/flame_offline_tum.cc:578-579, results out :628-635, stats
// src/utils.cc:131-136).  No CPU fallback exists here: without a device every compute entry
// point returns FLAME_HIP_ERR_NODEVICE.
#include "../../include/flame_hip.h"

#include <hip/hip_runtime.h>
#include <roctracer/roctx.h>

#include <atomic>
#include <chrono>
#include <climits>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <unordered_map>
#include <mutex>
#include <vector>

#include "delaunay_dev.h"
#include "kernels.h"
#include "plan.h"
#include "plan_dev.h"
#include "sync.h"

using namespace flamehip;

#define HIPCHK(expr)                                             \
  do {                                                           \
    hipError_t e__ = (expr);                                     \
    if (e__ != hipSuccess) return FLAME_HIP_ERR_HIP - (int)e__;  \
  } while (0)

// Fault injection for the tests, compiled ONLY into flame_ros_amd/libflame_hip_hooks.so (-DFLAME_HIP_TEST_HOOKS=1, built by
// flame_ros_amd/build.py beside the product library; tests ask for it by name, tests/util.py hooks_env()).  The product
// library reads no environment variable and has no such switch.  flame_hip_test_hook(key, value), process-wide:
//   "persist_fail" 1      every launch of resident tiles counts as having given up (the recovery paths)
//   "fill_alloc"   0..255 new device allocations are filled with that byte (stale-read detection), -1 off
//   "persist_stall_us" n  tile 0 is REALLY late by n us behind its first hand-off (needs kernels.hip built with
//                         -DFLAME_PERSIST_STALL_HOOK=1: tools/exp/build_variant.sh)
#ifndef FLAME_HIP_TEST_HOOKS
#define FLAME_HIP_TEST_HOOKS 0
#endif
#if FLAME_HIP_TEST_HOOKS
namespace {
struct TestHooks { std::atomic<int> persist_fail{0}, fill_alloc{-1}, persist_stall_us{0}; };
TestHooks& test_hooks() { static TestHooks h; return h; }
}  // namespace
extern "C" int flame_hip_test_hook(const char* key, int32_t value) {
  if (!key) return FLAME_HIP_ERR_ARG;
  const std::string k(key);
  if (k == "persist_fail") test_hooks().persist_fail = value;
  else if (k == "fill_alloc") test_hooks().fill_alloc = value;
  else if (k == "persist_stall_us") test_hooks().persist_stall_us = value;
  else return FLAME_HIP_ERR_ARG;
  return 0;
}
#define TEST_HOOK(name, off) (test_hooks().name.load())
#else
#define TEST_HOOK(name, off) (off)
#endif
int flamehip::test_alloc_fill() { return TEST_HOOK(fill_alloc, -1); }

namespace {

// roctx range over one ABI call (visible in rocprofv3 --marker-trace; SURVEY.md 5 "tracing")
struct RoctxRange {
  explicit RoctxRange(const char* name) { roctxRangePush(name); }
  ~RoctxRange() { roctxRangePop(); }
};

struct GraphExecEntry {
  int32_t iters;
  int cur;
  SolveParams p;
  hipGraphExec_t exec;
  int launches;
};

// Device buffers are owned by the handle and keep their capacity across uploads (a frame stream
// re-uploads a similar graph every frame: no hipMalloc/hipFree after the first few frames).
// `caps` maps the address of the pointer member to its capacity in bytes.
typedef std::unordered_map<void*, size_t> CapMap;

template <class T>
int dev_alloc(CapMap& caps, T** p, size_t n) {
  n += 1;  // one pad element: kernels clamp indices of empty lists to element 0
  const size_t bytes = n * sizeof(T);
  auto it = caps.find((void*)p);
  if (*p && it != caps.end() && it->second >= bytes) return 0;
  if (*p && it != caps.end()) (void)hipFree(*p);  // (no entry: the pointer lives inside the upload arena)
  *p = nullptr;
  const size_t want = bytes + bytes / 4;  // slack for the next, slightly larger frame
  hipError_t e = hipMalloc(reinterpret_cast<void**>(p), want);
  if (e == hipErrorOutOfMemory) return FLAME_HIP_ERR_ALLOC;
  if (e != hipSuccess) return FLAME_HIP_ERR_HIP - (int)e;
  const int fill = flamehip::test_alloc_fill();  // (-1 in the product library)
  if (fill >= 0) (void)hipMemset(*p, fill, want);
  caps[(void*)p] = want;
  return 0;
}

// ... in UNCACHED device memory (hipDeviceMallocUncached: no L2 holds it, so stores and loads of different XCDs meet in
// memory while a kernel runs -- the hand-off copies of the resident tiles, kernels.hip PersistArgs).  Uncached blocks
// are never handed back to the runtime: on this stack (ROCm 7.2) a later ordinary hipMalloc that recycles such a
// block misbehaves (tools/exp/fault_repro.py: the plan builder's kernels fault on it), so freed blocks wait in a
// process-wide pool, per device, for the next uncached request (bounded by the largest graph the process has seen).
struct UncachedPool {
  std::mutex m;
  struct Block { void* p; size_t bytes; int device; };
  std::vector<Block> free_blocks;
  void* take(int device, size_t bytes, size_t* got) {
    std::lock_guard<std::mutex> lk(m);
    size_t best = free_blocks.size();
    for (size_t i = 0; i < free_blocks.size(); ++i)
      if (free_blocks[i].device == device && free_blocks[i].bytes >= bytes &&
          (best == free_blocks.size() || free_blocks[i].bytes < free_blocks[best].bytes))
        best = i;
    if (best == free_blocks.size()) return nullptr;
    void* p = free_blocks[best].p;
    *got = free_blocks[best].bytes;
    free_blocks.erase(free_blocks.begin() + (long)best);
    return p;
  }
  void give(int device, void* p, size_t bytes) {
    std::lock_guard<std::mutex> lk(m);
    free_blocks.push_back({p, bytes, device});
  }
};
static UncachedPool& uncached_pool() {
  static UncachedPool* pool = new UncachedPool();  // (never destroyed: handles may outlive static destruction order)
  return *pool;
}
// (the blocks are NOT registered in the handle's CapMap: free_device() must not hipFree them; *cap = bytes of *p)
template <class T>
int dev_alloc_uncached(int device, T** p, size_t* cap, size_t n, bool* fresh) {
  n += 1;
  const size_t bytes = n * sizeof(T);
  *fresh = false;
  if (*p && *cap >= bytes) return 0;
  if (*p) uncached_pool().give(device, *p, *cap);
  *p = nullptr; *cap = 0;
  *fresh = true;  // (contents unknown: the caller zeroes it)
  size_t got = 0;
  if (void* q = uncached_pool().take(device, bytes, &got)) {
    *p = static_cast<T*>(q); *cap = got;
    return 0;
  }
  const size_t want = bytes + bytes / 4;
  hipError_t e = hipExtMallocWithFlags(reinterpret_cast<void**>(p), want, hipDeviceMallocUncached);
  if (e == hipErrorOutOfMemory) return FLAME_HIP_ERR_ALLOC;
  if (e != hipSuccess) return FLAME_HIP_ERR_HIP - (int)e;
  *cap = want;
  return 0;
}

// All copies go through the handle's own non-blocking stream (hipMemcpyAsync + stream sync),
// never the legacy stream: a legacy-stream copy in one host thread collides with a stream capture
// running in another thread (hipErrorStreamCaptureImplicit), and distinct handles must be usable
// from distinct threads.
inline hipError_t memcpy_sync(hipStream_t s, void* dst, const void* src, size_t bytes, hipMemcpyKind kind) {
  hipError_t e = hipMemcpyAsync(dst, src, bytes, kind, s);
  if (e != hipSuccess) return e;
  return hipStreamSynchronize(s);
}

template <class T>
int h2d(hipStream_t s, T* dst, const std::vector<T>& src) {
  if (src.empty()) return 0;
  HIPCHK(memcpy_sync(s, dst, src.data(), src.size() * sizeof(T), hipMemcpyHostToDevice));
  return 0;
}

// Asynchronous variant for the upload path: the source vectors outlive the call (plan members or
// locals of a function that synchronises the stream before it returns), so a whole upload pays one
// stream synchronisation instead of one per array.
// Page-locked staging area of a handle: host arrays are copied into it once and leave through
// truly asynchronous DMAs (a hipMemcpyAsync from pageable memory is staged and waited for by the
// runtime, call by call -- ~10 us each, the larger part of a small frame's upload).
// frame_results: outputs up to this size are written to host memory by the kernels themselves (measured on
// the facade stream, direct vs one copy: 1.2 k -14 us, 5 k -10 us, 10 k (185 KB) -15 us, 50 k (925 KB) +65 us)
constexpr int kPollDelayDefault = 2;  // x 256 clocks.  r05, address-sorted poll: a pass is short enough to sample memory before the neighbours' stores have landed (passes per round 1.35 at 0, 1.10 at 2, 1.01 at 3; 50 k 1.316 / 1.319 / 1.304 / 1.303 / 1.357 us per iteration at 0 / 1 / 2 / 3 / 4, profiles/r05_sorted_poll_delay.txt; swept again in r06 behind the one-block poll: 1.313 / 1.334 / 1.309 / 1.329 / 1.350); option "poll_delay" overrides
constexpr int kMapMinTiles = 6;  // partitions of at least this many tiles leave a tile map for the next frame (partition reuse)
constexpr size_t kDirectOutBytes = 256 * 1024;

struct PinnedArena {
  char* base = nullptr;
  size_t cap = 0, used = 0;
  // room for `bytes` more (only ever grows between uploads: call with nothing in flight)
  hipError_t reserve(size_t bytes) {
    used = 0;
    if (bytes <= cap) return hipSuccess;
    if (base) (void)hipHostFree(base);
    base = nullptr; cap = 0;
    const size_t want = bytes + bytes / 2 + 4096;
    hipError_t e = hipHostMalloc(reinterpret_cast<void**>(&base), want, hipHostMallocDefault);
    if (e == hipSuccess) cap = want;
    return e;
  }
  void* put(const void* src, size_t bytes) {
    used = (used + 63) & ~(size_t)63;
    if (used + bytes > cap) return nullptr;
    void* p = base + used;
    std::memcpy(p, src, bytes);
    used += bytes;
    return p;
  }
  void release() { if (base) (void)hipHostFree(base); base = nullptr; cap = used = 0; }
};

// H2D of a host vector through the arena (falls back to the pageable source if it does not fit)
template <class T>
int h2d_pinned(hipStream_t s, PinnedArena& ar, T* dst, const T* src, size_t n) {
  if (n == 0) return 0;
  const void* p = ar.put(src, n * sizeof(T));
  HIPCHK(hipMemcpyAsync(dst, p ? p : (const void*)src, n * sizeof(T), hipMemcpyHostToDevice, s));
  return 0;
}
template <class T>
int h2d_pinned(hipStream_t s, PinnedArena& ar, T* dst, const std::vector<T>& src) {
  return h2d_pinned(s, ar, dst, src.data(), src.size());
}

// Results leave the same way: device -> page-locked arena (asynchronous DMAs), ONE synchronisation,
// then plain memcpys into the caller's (pageable) buffers.
struct D2HBatch {
  PinnedArena& ar;
  hipStream_t s;
  struct Item { void* dst; const void* tmp; size_t bytes; };
  std::vector<Item> items;
  D2HBatch(PinnedArena& a, hipStream_t st) : ar(a), s(st) {}
  hipError_t add(void* dst, const void* dev, size_t bytes) {
    if (!bytes || !dst) return hipSuccess;
    ar.used = (ar.used + 63) & ~(size_t)63;
    if (ar.used + bytes > ar.cap)  // (reserve() was sized for every piece: not expected)
      return hipMemcpyAsync(dst, dev, bytes, hipMemcpyDeviceToHost, s);
    void* tmp = ar.base + ar.used;
    ar.used += bytes;
    items.push_back({dst, tmp, bytes});
    return hipMemcpyAsync(tmp, dev, bytes, hipMemcpyDeviceToHost, s);
  }
  hipError_t finish() {
    hipError_t e = hipStreamSynchronize(s);
    if (e != hipSuccess) return e;
    for (const Item& it : items) std::memcpy(it.dst, it.tmp, it.bytes);
    items.clear();
    return hipSuccess;
  }
};

template <class T>
int h2d_async(hipStream_t s, T* dst, const std::vector<T>& src) {
  if (src.empty()) return 0;
  HIPCHK(hipMemcpyAsync(dst, src.data(), src.size() * sizeof(T), hipMemcpyHostToDevice, s));
  return 0;
}

bool all_finite(const float* p, size_t n) {
  for (size_t k = 0; k < n; ++k)
    if (!std::isfinite(p[k])) return false;
  return true;
}

}  // namespace

struct flame_hip_graph {
  int device = -1;
  int32_t V = 0, E = 0, T = 0;
  bool uploaded = false;
  PlanOptions opt;
  int num_cus = 0;
  int use_graph = 1;
  Plan plan;
  SyncOut sync;          // inputs derived by flame_hip_graph_sync (kept for flame_hip_graph_edges)
  bool synced = false;   // the current graph came from flame_hip_graph_sync
  int path = 0;  // resolved path after upload

  hipStream_t stream = nullptr;
  hipStream_t stream_in = nullptr;  // input staging (H2D of a frame overlaps the partition kernels)
  DelaunayScratch dt;               // flame_hip_delaunay: arenas kept between frames
  hipEvent_t ev0 = nullptr, ev1 = nullptr, ev_in = nullptr;
  hipEvent_t ev_state = nullptr;  // last state-writing work on `stream` (upload, scale, filter, results)
  bool state_pending = false;     // ev_state was recorded since the last full synchronisation
  float state_scale = 1.0f;       // factor applied to the resident state since its upload (rescale_data epilogue)
  bool timed = false;
  int last_launches = 0;

  float4* A[2] = {nullptr, nullptr};
  float4* B[2] = {nullptr, nullptr};
  float4* q[2] = {nullptr, nullptr};
  int cur = 0;
  int2* eij = nullptr;
  float4* ew = nullptr;
  int32_t* grow = nullptr;
  int32_t* ginc = nullptr;
  float2* pos = nullptr;
  // tiles
  TileDesc* tiles = nullptr;
  int32_t* t_vmap = nullptr;
  int32_t* t_emap = nullptr;
  uint2* t_eij = nullptr;
  float4* t_ew = nullptr;
  uint32_t* t_srow = nullptr;
  // triangles
  int32_t* tris = nullptr;
  int32_t* trow = nullptr;
  int32_t* tinc = nullptr;
  float4* tri_normals = nullptr;
  float4* vtx_normals = nullptr;
  uint8_t* tri_valid = nullptr;
  // dense maps (row f2), allocated on first use for the requested image size
  int64_t map_pixels = 0;
  uint32_t* map_owner = nullptr;
  float* map_idm = nullptr;
  float* map_dm = nullptr;
  float* map_cloud = nullptr;
  uint32_t* map_cov = nullptr;   // per-block covered-pixel counts of the last raster (stat key coverage)
  char* fr_dev = nullptr;        // output arena of flame_hip_frame_results (one D2H per frame)
  uint32_t* dbg_key = nullptr;   // debug images: winning-primitive key map, BGR8 output, staged features
  uint8_t* dbg_bgr = nullptr;
  float* dbg_feat = nullptr;
  // which raster the map buffers hold: the solver state it was made from (state_serial), the
  // filter parameters, filtered or not -- dense maps, coverage and the debug images share it
  uint64_t state_serial = 1, raster_serial = 0;
  int raster_filtered = -1;
  flame_hip_tri_params raster_tp;
  float raster_kinv[9];
  // graph filter scratch (row a9)
  float* filter_tmp = nullptr;
  // mesh output (row f1)
  float4* mesh_pts = nullptr;
  // permutations on the device (results leave in the caller's order without a host pass)
  int32_t* v_i2o_dev = nullptr;
  int32_t* v_o2i_dev = nullptr;
  int32_t* e_i2o_dev = nullptr;
  int32_t* e_o2i_dev = nullptr;
  float* dl_v = nullptr;  // download staging: 3V floats
  float* dl_q = nullptr;  // 3E floats
  float* dl_n = nullptr;  // 3V floats (vertex normals of flame_hip_frame_results)
  // device plan builder (row f3) and the staged inputs it reads (caller's order)
  int plan_device = 1;
  int plan_reuse = 1;          // frame streams: partition from the previous frame's tile map
  int plan_mini = 1;           // option "plan_mini": small frames of a graph sync planned by one launch (k_mini_plan)
  bool plan_mini_used = false; // ... the current plan was
  // option "persist" (default 1): graphs of 2 .. min(kPersistMaxTiles, CUs) halo tiles are solved by ONE launch of
  // RESIDENT tiles (kernels.hip k_tile_persist: round-tagged hand-offs through uncached memory instead of a kernel
  // boundary per `depth` iterations); 0: launches
  bool persist = true, persist_used = false;
  float4 *snapA = nullptr, *snapB = nullptr, *snapq = nullptr;  // flame_hip_state_snapshot / _rollback
  bool snap_valid = false;
  int64_t persist_launches = 0;     // launches of resident tiles so far (info "persist_launches")
  float persist_round_us = 0.f;     // device time of a round of the last resident solve that was looked at (0 = none yet)
  int32_t persist_round_V = 0;      // ... and the size of the graph it was measured on
  int32_t last_rounds = 0;          // rounds of the last resident launch
  int32_t persist_timeout_us = 0;   // what the last resident launch was given (info "persist_timeout_us")
  int32_t snap_V = 0, snap_E = 0;
  PersistBufs xp;                   // hand-off buffers (uncached, from the process-wide pool: NOT in caps) + dev-aid words
  size_t xp_cap[6] = {0, 0, 0, 0, 0, 0};
  // r06, option "one_xcd" (default 1): a graph of up to 32 resident tiles keeps them on ONE XCD -- the launch has 8 x ntiles
  // workgroups of which every 8th carries a tile -- and hands over through ORDINARY memory, i.e. that XCD's L2 (0.5 us per
  // load instead of 0.9 through uncached memory: 1.2 k vertices 0.89 -> 0.81 us per iteration).  Which XCD a workgroup lands on
  // is the dispatcher's habit, not a guarantee: tiles on different XCDs would never see each other's tags, time out, and the
  // solve is repeated by launches like any give-up -- after which the device's lease never tries the mode again.
  bool one_xcd_opt = true, one_xcd_used = false;
  float4* cA[2] = {nullptr, nullptr};  // the hand-off copies in ordinary memory
  float4* cB[2] = {nullptr, nullptr};
  float4* cq[2] = {nullptr, nullptr};
  void* c_zeroed[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};  // the allocations that have been zeroed (tag 0 is never a
  size_t c_zeroed_cap[6] = {0, 0, 0, 0, 0, 0};                                 // round's): address AND capacity, i.e. no reallocation since
  int poll_delay_opt = -1;          // option "poll_delay" (-1 = automatic)
  int persist_timeout_opt = 0;      // option "persist_timeout_us" (0 = automatic)
  bool need_marks = true;           // option "need_marks"
  int persist_prof_want = 0, persist_prof_set = 0;  // option "persist_prof": tile + 1 that records its round split (0 = none)
  bool persist_unchecked = false;   // a resident launch is in flight / finished and nobody has looked at persist_err yet
  int persist_unchecked_n = 0;      // ... how many of them (the error word does not say WHICH launch gave up: only when it
                                    // can only have been the last solve is that solve repeated)
  bool persist_skip_once = false;   // the next enqueue goes by launches (the repeat of a solve that gave up)
  int last_src = 0;                 // the buffer the last solve started from
  int32_t* persist_err = nullptr;   // page-locked: raised by a launch whose wait timed out
  int32_t persist_base = 0;         // the hand-off tags used so far (they only grow)
  SolveParams last_sp{};            // the last solve (a persistent launch that gave up is repeated by launches)
  int32_t last_iters = 0;
  hipStream_t last_stream = nullptr;
  bool init_have_x0 = false;        // the device-built plan's initial state came with an x0 array
  int persist_recovered = 0;        // how many solves were repeated that way (flame_hip_get_info)
  // the last give-up of this handle, for whoever has to find out why (info "persist_gave_up_tile" / _round / _front_round /
  // _not_started / _rounds / _tiles / _one_xcd / _timeout_us, persist_note_give_up()): the tile that had handed over the fewest
  // rounds and how many, the most any tile had, the tiles that had handed over none, the launch's rounds
  struct GiveUp { int32_t tile = -1, round = -1, front_round = 0, not_started = 0, rounds = 0, tiles = 0, one_xcd = 0, timeout_us = 0; } give_up;
  int32_t give_up_base = 0;         // persist_base of the last resident launch (its tags are give_up_base + 1 ..)
  uint64_t solve_serial = 0;        // state_serial right behind the last solve: later state-writing calls move on from it
  // r05: SEVERAL solves queued behind each other without a synchronising call are repeatable too (a bench window, a caller
  // that pipelines solves): every solve from the first unchecked resident one on is logged, and when a second one is queued
  // the source buffers of the first -- which it would overwrite -- are copied aside first (device to device, once per
  // synchronising call).  A give-up then restores that state and repeats the whole queue by launches.
  struct QueuedSolve { SolveParams sp; int32_t iters; };
  std::vector<QueuedSolve> queued;  // since the last look, the first unchecked resident solve first
  int queued_src = 0;               // the buffer that solve started from
  uint64_t queued_serial = 0;       // state_serial in front of it
  float4 *qsnapA = nullptr, *qsnapB = nullptr, *qsnapq = nullptr;
  bool qsnap_valid = false;
  int stream_depth = 0;        // option "stream_depth": halo depth of small graphs (<= 64 tiles) instead of the
                               // auto depth 8, which is tuned for a RESIDENT graph (fewest launches); a graph
                               // that is solved once pays for its plan, and that is cheapest at depth 4-5
  int tile_imbalance_pct = 100; // info "tile_imbalance_pct": 100 x max / mean of the tiles' cost (e_loc + 2 n_ext)
  int single_cap = 1 << 30;    // an isolated tile did not fit a graph of this many vertices + 1 on this handle
  bool plan_reused = false;    // the current plan's partition came from the map
  int reuse_tile_own_opt = 0;  // the "tile_own" option the map was made with
  int reuse_backoff = 0, reuse_skip = 0;  // frames to sit out after a rejected reuse (doubles, <= 16)
  // graph sync without waiting for the edge count: E is predicted (V + T + euler_off, -1 for a
  // triangulated disk) and checked at the plan builder's first synchronisation
  bool spec_edges = false;     // the current upload_device_plan() runs on a predicted count
  int32_t true_edges = 0;      // ... and this is what the device counted when the prediction failed
  int32_t euler_off = -1;      // E - V - T of the last frame
  int euler_skip = 0, euler_backoff = 0;
  DevPlanner planner;
  float2* in_pos = nullptr;
  int2* in_edges = nullptr;
  float* in_alpha = nullptr;
  float* in_beta = nullptr;
  float* in_z = nullptr;
  float* in_wgt = nullptr;
  float* in_x0 = nullptr;
  int32_t* in_tris = nullptr;
  float* in_mu = nullptr;    // graph sync on the device: measured idepths, variances, predictions
  float* in_var = nullptr;
  float* in_pred = nullptr;
  bool sync_on_device = false;  // the derived edge list lives in in_edges (sync.edges is stale)
  bool beta_is_alpha = false;   // staged graph sync: beta = alpha, one buffer
  int32_t* dflags = nullptr;
  bool host_perms = true;  // plan.v_i2o / v_o2i / e_i2o / e_o2i / tris are valid on the host
  // costs
  double* partials = nullptr;
  uint8_t* cost_mask = nullptr;  // flame_hip_costs_masked: V vertex flags then E edge flags, internal order
  // halo exchange lists (internal ids), see flame_hip_halo_register
  int32_t n_send_v = 0, n_send_e = 0, n_recv_v = 0, n_recv_e = 0;
  int32_t* halo_send_v = nullptr;
  int32_t* halo_send_e = nullptr;
  int32_t* halo_recv_v = nullptr;
  int32_t* halo_recv_e = nullptr;
  // debug timeline of the tile kernel
  int profile = 0;
  unsigned long long* prof = nullptr;

  std::vector<GraphExecEntry> execs;
  CapMap caps;
  int solves_since_upload = 0;
  PinnedArena pin;             // page-locked staging of the host arrays of an upload
  PinnedArena pin_in;          // ... of the inputs of a small graph sync (one DMA; k_mini_plan reads the device copy)
  char* in_stage = nullptr;
  PinnedArena pout;            // page-locked landing area of the results (frame_results, download)
  char* harena = nullptr;      // device arena the host-built plan of the current upload lives in
  bool lanes_applied = false;  // lane_order = 1: the conflict-avoiding lane order is in the device arrays
  bool poll_valid = false;     // the resident tiles' poll lists (xp.poll_*) belong to the current tile arrays
  bool poll_sorted = false;    // ... and are the address-sorted ones (from a plan's second solve on)

  void drop_execs() {
    for (auto& e : execs) (void)hipGraphExecDestroy(e.exec);
    execs.clear();
  }

  void free_device() {
    if (device < 0) return;
    (void)hipSetDevice(device);
    drop_execs();
    // every device buffer of the handle is registered in `caps` (key = address of the member)
    for (auto& kv : caps) {
      void** pp = reinterpret_cast<void**>(kv.first);
      if (*pp) (void)hipFree(*pp);
      *pp = nullptr;
    }
    caps.clear();
    for (int b = 0; b < 2; ++b) {  // (uncached blocks go back to the pool, never to the runtime)
      float4** pp[3] = {&xp.hA[b], &xp.hB[b], &xp.hq[b]};
      for (int k = 0; k < 3; ++k) {
        if (*pp[k]) uncached_pool().give(device, *pp[k], xp_cap[3 * b + k]);
        *pp[k] = nullptr; xp_cap[3 * b + k] = 0;
      }
    }
    xp.prof = nullptr;  // (was in caps)
    xp.poll_v = nullptr; xp.poll_e = nullptr; xp.poll_ne = nullptr;
    xp.need_v = nullptr; xp.need_e = nullptr; xp.need_valid = false;
    pin.release();
    pin_in.release();
    pout.release();
    map_pixels = 0;
    n_send_v = n_send_e = n_recv_v = n_recv_e = 0;
  }
};

extern "C" {

int flame_hip_version(void) { return 402; }  // (402, r06: halo view, peer transport, local communicator, handle options instead of environment switches)

const char* flame_hip_strerror(int code) {
  switch (code) {
    case FLAME_HIP_OK: return "ok";
    case FLAME_HIP_ERR_ARG: return "invalid argument";
    case FLAME_HIP_ERR_STATE: return "invalid call order (graph not uploaded?)";
    case FLAME_HIP_ERR_NAN: return "non-finite input";
    case FLAME_HIP_ERR_ALLOC: return "allocation failed";
    case FLAME_HIP_ERR_NODEVICE: return "no HIP device (plan-only handle or no GPU present)";
    case FLAME_HIP_ERR_NORCCL: return "librccl.so could not be loaded";
    default: break;
  }
  if (code <= FLAME_HIP_ERR_RCCL) return "RCCL error (code = -(3000 + ncclResult_t))";
  if (code <= FLAME_HIP_ERR_HIP) return hipGetErrorString((hipError_t)(FLAME_HIP_ERR_HIP - code));
  return "unknown error";
}

int flame_hip_graph_create(flame_hip_graph** out, int device, int32_t V, int32_t E, int32_t T) {
  if (!out) return FLAME_HIP_ERR_ARG;
  *out = nullptr;
  if (V < 0 || E < 0 || T < 0) return FLAME_HIP_ERR_ARG;
  flame_hip_graph* g = new (std::nothrow) flame_hip_graph();
  if (!g) return FLAME_HIP_ERR_ALLOC;
  g->V = V; g->E = E; g->T = T;
  g->device = device;
  if (device >= 0) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || device >= n) { delete g; return FLAME_HIP_ERR_NODEVICE; }
    hipDeviceProp_t prop;
    if (hipSetDevice(device) != hipSuccess || hipGetDeviceProperties(&prop, device) != hipSuccess) {
      delete g;
      return FLAME_HIP_ERR_NODEVICE;
    }
    g->opt.lds_bytes = (int64_t)prop.sharedMemPerBlock > 0 ? (int64_t)prop.sharedMemPerBlock : 64 * 1024;
    g->num_cus = prop.multiProcessorCount;
    g->opt.num_cus = g->num_cus;
    if (hipStreamCreateWithFlags(&g->stream, hipStreamNonBlocking) != hipSuccess ||
        hipStreamCreateWithFlags(&g->stream_in, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreate(&g->ev0) != hipSuccess || hipEventCreate(&g->ev1) != hipSuccess ||
        hipEventCreateWithFlags(&g->ev_in, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&g->ev_state, hipEventDisableTiming) != hipSuccess) {
      delete g;
      return FLAME_HIP_ERR_NODEVICE;
    }
  }
  *out = g;
  return 0;
}

static void persist_lease_drop(flame_hip_graph* g, bool gave_up);
static int persist_gave_up_count(int device);
static int persist_backoff(int device);
static bool one_xcd_allowed(int device);

void flame_hip_graph_destroy(flame_hip_graph* g) {
  if (!g) return;
  if (g->device >= 0) {
    (void)hipSetDevice(g->device);
    if (g->stream) (void)hipStreamSynchronize(g->stream);
    if (g->timed) (void)hipEventSynchronize(g->ev1);
    persist_lease_drop(g, false);
    (void)g->planner.wait_maps();
    g->free_device();
    g->dt.release();
    if (g->persist_err) (void)hipHostFree(g->persist_err);
    if (g->ev0) (void)hipEventDestroy(g->ev0);
    if (g->ev1) (void)hipEventDestroy(g->ev1);
    if (g->ev_in) (void)hipEventDestroy(g->ev_in);
    if (g->ev_state) (void)hipEventDestroy(g->ev_state);
    if (g->stream_in) { (void)hipStreamSynchronize(g->stream_in); (void)hipStreamDestroy(g->stream_in); }
    if (g->stream) (void)hipStreamDestroy(g->stream);
  }
  delete g;
}

// A solve may have been enqueued on a caller-supplied stream: everything that rewrites the solver
// state from the host waits for its end event first (the handle's own stream is synchronised by
// the callers themselves).
static hipError_t wait_last_solve(flame_hip_graph* g) {
  if (g->device < 0 || !g->timed) return hipSuccess;
  return hipEventSynchronize(g->ev1);
}

// A launch of resident tiles gave up: say how far its tiles got (cold path, nothing in the kernel -- a per-tile record written at
// the time-out cost every size 3-5 % through the poll loop's registers and code layout, profiles/r06_giveup_record_ab.txt).
// The x_bar hand-off copies of both parities still hold what the LAST launch's tiles stored: a tile's highest tag above the
// launch's base is the last round it handed over.  The tile with the lowest one is the late one (0 rounds: it never started) or,
// when the late one caught up after its neighbours had stopped, stood next to it.  analyse = false: the plan is no longer the
// launch's (a new upload), only the launch's own numbers are kept.
static void persist_note_give_up(flame_hip_graph* g, bool analyse = true) {
  flame_hip_graph::GiveUp u;
  u.tiles = (int32_t)g->plan.tiles.size();
  u.one_xcd = g->one_xcd_used ? 1 : 0;
  u.timeout_us = g->persist_timeout_us;
  u.rounds = g->last_rounds;
  const int32_t V = g->V, base = g->give_up_base;
  float4* const* hB = g->one_xcd_used ? g->cB : g->xp.hB;
  if (analyse && V > 0 && hB[0] && hB[1] && g->persist_used && *g->persist_err == 1) {  // (a forced failure of the hooks library: nothing to look at)
    std::vector<float4> e0((size_t)V), e1((size_t)V);
    if (hipMemcpy(e0.data(), hB[0], sizeof(float4) * (size_t)V, hipMemcpyDeviceToHost) == hipSuccess &&
        hipMemcpy(e1.data(), hB[1], sizeof(float4) * (size_t)V, hipMemcpyDeviceToHost) == hipSuccess) {
      for (int t = 0; t < u.tiles; ++t) {
        const TileDesc& T = g->plan.tiles[t];
        int32_t last = 0;
        for (int32_t v = T.vstart; v < T.vstart + T.n_own && v < V; ++v)
          for (const float4* e : {&e0[(size_t)v], &e1[(size_t)v]}) {
            int32_t tag;
            std::memcpy(&tag, &e->w, sizeof(tag));
            if (tag > base && tag - base <= u.rounds) last = std::max(last, tag - base);
          }
        if (last == 0) ++u.not_started;
        if (u.tile < 0 || last < u.round) { u.tile = t; u.round = last; }
        u.front_round = std::max(u.front_round, last);
      }
    } else (void)hipGetLastError();
  }
  g->give_up = u;
}

// A new upload throws the state of the solves before it away: whether one of their resident launches gave up no longer
// matters for the results -- only for the lease and the back-off.  (Called behind the upload's own wait for those solves.)
static void persist_discard(flame_hip_graph* g) {
  g->queued.clear(); g->qsnap_valid = false;
  if (!g->persist_unchecked) return;
  g->persist_unchecked = false;
  g->persist_unchecked_n = 0;
  if (g->persist_err && *g->persist_err != 0) {
    persist_note_give_up(g, false);
    *g->persist_err = 0;
    persist_lease_drop(g, true);
  }
}

// State-writing work enqueued on the handle's own stream (upload, un-scaling, filters) is marked
// with an event; a solve / halo pack / unpack on a CALLER's stream orders itself behind it
// (the device-plan upload returns with k_init_state still in flight).
static hipError_t mark_state(flame_hip_graph* g) {
  g->state_pending = true;
  g->state_serial++;
  return hipEventRecord(g->ev_state, g->stream);
}
static hipError_t order_after_state(flame_hip_graph* g, hipStream_t s) {
  if (s == g->stream || !g->state_pending) return hipSuccess;
  return hipStreamWaitEvent(s, g->ev_state, 0);
}

int flame_hip_graph_resize(flame_hip_graph* g, int32_t V, int32_t E, int32_t T) {
  if (!g || V < 0 || E < 0 || T < 0) return FLAME_HIP_ERR_ARG;
  if (g->device >= 0) {
    HIPCHK(hipSetDevice(g->device));
    HIPCHK(wait_last_solve(g));
    HIPCHK(hipStreamSynchronize(g->stream));
    persist_discard(g);
  }
  g->V = V; g->E = E; g->T = T;
  g->uploaded = false;
  g->n_send_v = g->n_send_e = g->n_recv_v = g->n_recv_e = 0;  // halo lists index the old graph
  return 0;
}

int flame_hip_set_option(flame_hip_graph* g, const char* key, int32_t value) {
  if (!g || !key) return FLAME_HIP_ERR_ARG;
  const std::string k(key);
  if (k == "path") {
    if (value < 0 || value > 2) return FLAME_HIP_ERR_ARG;
    g->opt.path = value;
  } else if (k == "tile_own") {
    if (value < 0) return FLAME_HIP_ERR_ARG;
    g->opt.tile_own = value;
  } else if (k == "tile_depth") {
    if (value < 0 || value > kMaxDepth) return FLAME_HIP_ERR_ARG;
    g->opt.tile_depth = value;
  } else if (k == "tile_threads") {
    if (value != 0 && value != 256 && value != 512 && value != 1024) return FLAME_HIP_ERR_ARG;
    g->opt.tile_threads = value;
  } else if (k == "use_graph") {
    g->use_graph = value != 0;
  } else if (k == "order_mode") {
    if (value < 0 || value > 1) return FLAME_HIP_ERR_ARG;
    g->opt.order_mode = value;
  } else if (k == "lane_order") {
    if (value < 0 || value > 2) return FLAME_HIP_ERR_ARG;
    g->opt.lane_order = value;
  } else if (k == "balance") {
    g->opt.balance = value != 0;
  } else if (k == "host_threads") {
    if (value < 0) return FLAME_HIP_ERR_ARG;
    g->opt.host_threads = value;
  } else if (k == "tile_single_max") {
    if (value < 0 || value > 2048) return FLAME_HIP_ERR_ARG;
    g->opt.single_max = value;
  } else if (k == "debug_sub_cap") {
    g->opt.debug_sub_cap = value;
  } else if (k == "d_sign") {
    if (value != 1 && value != -1) return FLAME_HIP_ERR_ARG;
    g->opt.d_sign = value;
  } else if (k == "plan_device") {
    g->plan_device = value != 0;
  } else if (k == "plan_reuse") {
    g->plan_reuse = value != 0;
  } else if (k == "plan_mini") {
    g->plan_mini = value != 0;
  } else if (k == "stream_depth") {
    if (value < 0 || value > kMaxDepth) return FLAME_HIP_ERR_ARG;
    g->stream_depth = value;
  } else if (k == "persist") {
    g->persist = value != 0;  // (2 was r03's "also size small frames for it": the automatic tiles do as well now)
  } else if (k == "persist_prof") {
    if (value < 0) return FLAME_HIP_ERR_ARG;
    g->persist_prof_want = value;
  } else if (k == "one_xcd") {
    g->one_xcd_opt = value != 0;
  } else if (k == "poll_delay") {  // x 256 clocks between a round's stores and its first poll pass; -1 = automatic
    if (value < -1 || value > 255) return FLAME_HIP_ERR_ARG;
    g->poll_delay_opt = value;
  } else if (k == "persist_timeout_us") {  // what a poll may wait before the launch gives up; 0 = automatic
    if (value < 0) return FLAME_HIP_ERR_ARG;
    g->persist_timeout_opt = value;
  } else if (k == "need_marks") {  // fat tiles hand over only what somebody polls (default 1)
    g->need_marks = value != 0;
  } else if (k == "plan_timing") {  // diagnostic: the plan builders' stages on stderr (levels: plan_dev.hip)
    if (value < 0 || value > 5) return FLAME_HIP_ERR_ARG;
    g->opt.timing = value;
  } else if (k == "profile") {
    g->profile = value != 0;
  } else if (k == "lds_bytes") {
    if (value < 1024) return FLAME_HIP_ERR_ARG;
    g->opt.lds_bytes = value;
  } else {
    return FLAME_HIP_ERR_ARG;
  }
  return 0;
}

int flame_hip_get_info(const flame_hip_graph* g, const char* key, int64_t* value) {
  if (!g || !key || !value) return FLAME_HIP_ERR_ARG;
  const std::string k(key);
  const Plan& P = g->plan;
  if (k == "V") *value = g->V;
  else if (k == "E") *value = g->E;
  else if (k == "T") *value = g->T;
  else if (k == "path") *value = g->path;
  else if (k == "num_tiles") *value = (int64_t)P.tiles.size();
  else if (k == "tile_threads") *value = P.tile_threads;
  else if (k == "tile_ept") *value = P.tile_ept;
  else if (k == "tile_vpt") *value = P.tile_vpt;
  else if (k == "tile_depth") *value = P.tile_depth;
  else if (k == "tile_lds_bytes") *value = P.tile_lds_bytes;
  else if (k == "tile_slot12") *value = P.tile_slot12 ? 1 : 0;
  else if (k == "tile_fat") *value = P.tile_fat ? 1 : 0;
  else if (k == "stall_hook_build") *value = tile_stall_hook_build() ? 1 : 0;
  else if (k == "tile_ext_vertices") { int64_t s = 0; for (auto& t : P.tiles) s += t.n_ext; *value = s; }
  else if (k == "tile_loc_edges") { int64_t s = 0; for (auto& t : P.tiles) s += t.e_loc; *value = s; }
  else if (k == "device") *value = g->device;
  else if (k == "plan_on_device") *value = P.on_device ? 1 : 0;
  else if (k == "plan_reused") *value = (P.on_device && g->plan_reused) ? 1 : 0;
  else if (k == "single_cap") *value = g->single_cap;
  else if (k == "tile_imbalance_pct") *value = P.on_device ? g->tile_imbalance_pct : 0;
  else if (k == "plan_mini") *value = (P.on_device && g->plan_mini_used) ? 1 : 0;
  else if (k == "stream_depth") *value = g->stream_depth;
  else if (k == "persist") *value = g->persist ? 1 : 0;
  else if (k == "persist_used") *value = g->persist_used ? 1 : 0;
  else if (k == "one_xcd_used") *value = (g->persist_used && g->one_xcd_used) ? 1 : 0;
  else if (k == "one_xcd") *value = g->one_xcd_opt ? 1 : 0;
  else if (k == "persist_recovered") *value = g->persist_recovered;
  else if (k == "persist_launches") *value = g->persist_launches;
  else if (k == "persist_timeout_us") *value = g->persist_timeout_us;
  else if (k == "persist_torn") {  // torn hand-off entries seen so far (only a FLAME_TORN_CHECK build counts; synchronise first)
    int32_t v = 0;
    if (g->xp.prof && hipMemcpy(&v, g->xp.prof + 8, sizeof(v), hipMemcpyDeviceToHost) != hipSuccess) return FLAME_HIP_ERR_HIP;
    *value = v;
  }
  else if (k == "torn_check_build") *value = tile_torn_check_build() ? 1 : 0;
  else if (k == "persist_round_ns") *value = (int64_t)(g->persist_round_us * 1e3f);
  else if (k == "persist_wait_us_max") {  // the longest a poll of any wave waited since the last look (read and cleared; synchronise first)
    int32_t v = 0;
    if (!g->xp.prof) { *value = 0; return 0; }
    if (hipMemcpy(&v, g->xp.prof + 7, sizeof(v), hipMemcpyDeviceToHost) != hipSuccess ||
        hipMemset(g->xp.prof + 7, 0, sizeof(v)) != hipSuccess) return FLAME_HIP_ERR_HIP;
    *value = v / 100;
  }
  else if (k == "persist_gave_up") *value = persist_gave_up_count(g->device);  // (give-ups of any handle on this device)
  else if (k == "persist_backoff") *value = g->device >= 0 ? persist_backoff(g->device) : 0;  // solves the device's lease still sits out (plans uploaded meanwhile are sized for launches)
  else if (k == "one_xcd_allowed") *value = one_xcd_allowed(g->device) ? 1 : 0;  // 0: a one-XCD launch gave up in this process, the mode is off
  else if (k == "persist_gave_up_tile") *value = g->give_up.tile;        // (this handle's last give-up, see GiveUp)
  else if (k == "persist_gave_up_round") *value = g->give_up.round;
  else if (k == "persist_gave_up_front_round") *value = g->give_up.front_round;
  else if (k == "persist_gave_up_not_started") *value = g->give_up.not_started;
  else if (k == "persist_gave_up_rounds") *value = g->give_up.rounds;
  else if (k == "persist_gave_up_tiles") *value = g->give_up.tiles;
  else if (k == "persist_gave_up_one_xcd") *value = g->give_up.one_xcd;
  else if (k == "persist_gave_up_timeout_us") *value = g->give_up.timeout_us;
  else if (k.rfind("persist_prof_", 0) == 0) {  // dev aid (option "persist_prof" = <tile + 1>): 10 ns ticks of that tile, summed over rounds
    const int i = std::atoi(k.c_str() + 13);
    int32_t v = 0;
    if (i < 0 || i > 4 || !g->xp.prof) return FLAME_HIP_ERR_ARG;
    if (hipMemcpy(&v, g->xp.prof + 2 + i, sizeof(v), hipMemcpyDeviceToHost) != hipSuccess) return FLAME_HIP_ERR_HIP;
    *value = v;
  }
  else if (k == "delaunay_hull") *value = g->dt.last_hull;  // flame_hip_delaunay: boundary vertices / live points / host microseconds of the last call
  else if (k == "delaunay_live") *value = g->dt.last_live;
  else if (k == "delaunay_us") *value = (int64_t)(g->dt.last_ms * 1000.0f);
  else if (k == "lds_bytes") *value = g->opt.lds_bytes;
  else if (k == "num_cus") *value = g->num_cus;
  else if (k == "clock_khz") {  // peak engine clock of the handle's device (timeline cycles -> time)
    int khz = 0;
    if (g->device < 0 || hipDeviceGetAttribute(&khz, hipDeviceAttributeClockRate, g->device) != hipSuccess) return FLAME_HIP_ERR_NODEVICE;
    *value = khz;
  }
  else return FLAME_HIP_ERR_ARG;
  return 0;
}

int flame_hip_graph_upload_batch(flame_hip_graph* g, int32_t num_graphs, const int32_t* voff,
                                 const float* pos, const int32_t* edges, const float* alpha,
                                 const float* beta, const float* z, const float* wgt,
                                 const float* x0, const int32_t* tris) {
  if (!g || num_graphs < 1 || !voff) return FLAME_HIP_ERR_ARG;
  g->opt.batch_voff.assign(voff, voff + num_graphs + 1);
  const int saved_path = g->opt.path;
  g->opt.path = FLAME_HIP_PATH_TILE;
  int rc = flame_hip_graph_upload(g, pos, edges, alpha, beta, z, wgt, x0, tris);
  g->opt.path = saved_path;
  g->opt.batch_voff.clear();
  return rc;
}

static int finish_upload(flame_hip_graph* g);

// ---- device plan path (row f3): inputs are staged in the caller's order, the plan is built by
// plan_dev.hip, the initial state is permuted by a kernel.  Returns 1 = done, 0 = not eligible /
// could not be built there (the caller falls back to the host builder), < 0 = error. ----
namespace {
struct TileAllocCtx { flame_hip_graph* g; DevPlanArrays* A; };

int alloc_tile_arrays(void* ctx, size_t ntiles, size_t nv, size_t ne, size_t ns) {
  TileAllocCtx* c = static_cast<TileAllocCtx*>(ctx);
  flame_hip_graph* g = c->g;
  int rc;
  if ((rc = dev_alloc(g->caps, &g->tiles, ntiles)) || (rc = dev_alloc(g->caps, &g->t_vmap, nv)) ||
      (rc = dev_alloc(g->caps, &g->t_emap, ne)) || (rc = dev_alloc(g->caps, &g->t_eij, ne)) ||
      (rc = dev_alloc(g->caps, &g->t_ew, ne)) || (rc = dev_alloc(g->caps, &g->t_srow, ns)))
    return rc;
  c->A->tiles = g->tiles; c->A->t_vmap = g->t_vmap; c->A->t_emap = g->t_emap;
  c->A->t_eij = g->t_eij; c->A->t_ew = g->t_ew; c->A->t_srow = g->t_srow;
  return 0;
}
}  // namespace

// Plan options as they apply to ONE graph of V vertices on this handle (entry points: upload, graph
// sync): the isolated-tile limit capped by what failed to fit before, the stream depth for small
// graphs.  Restores the handle's options when the entry point returns.
struct GraphOptScope {
  flame_hip_graph* g;
  int saved_single_max, saved_depth;
  GraphOptScope(flame_hip_graph* g_, int32_t V) : g(g_), saved_single_max(g_->opt.single_max), saved_depth(g_->opt.tile_depth) {
    g->opt.single_max = std::min(g->opt.single_max, g->single_cap);
    // (plan-only handles size their tiles the same way: they are the device plans' reference.  ADVICE r4: the depth follows
    // whether resident tiles can actually be taken -- not while the device's lease sits out a back-off after a give-up, not
    // under the in-kernel timeline -- because depth 5 by ordinary launches is the slower configuration)
    g->opt.resident = g->persist && !g->prof && (g->device < 0 || persist_backoff(g->device) == 0);
    g->opt.one_xcd = g->opt.resident && g->one_xcd_opt && g->opt.num_cus >= 256 && one_xcd_allowed(g->device);
    if (g->stream_depth > 0 && g->opt.tile_depth == 0 && g->opt.tile_own <= 0 && g->opt.batch_voff.empty() && V <= 64 * 32)
      g->opt.tile_depth = g->stream_depth;
  }
  ~GraphOptScope() { g->opt.single_max = saved_single_max; g->opt.tile_depth = saved_depth; }
  GraphOptScope(const GraphOptScope&) = delete;
  GraphOptScope& operator=(const GraphOptScope&) = delete;
};

// LDS the resident tiles' poll delivery needs behind a tile's incidence slots: the primal state of the halo vertices it updates
static size_t persist_stage_bytes(const TileDesc& D) { return sizeof(float4) * (size_t)std::max(D.n_upd - D.n_own, 0); }

static int upload_device_plan(flame_hip_graph* g, const float* pos, const int32_t* edges,
                              const float* alpha, const float* beta, const float* z, const float* wgt,
                              const float* x0, const int32_t* tris, bool staged = false, bool have_x0 = false) {
  // staged: the inputs are already in the in_* device buffers (graph sync on the device)
  const int32_t V = g->V, E = g->E, T = (tris || (staged && g->T > 0)) ? g->T : 0;
  if (!staged) have_x0 = x0 != nullptr;
  g->init_have_x0 = have_x0;
  Plan& P = g->plan;
  PlanSizing sz = plan_sizing(g->opt, V, E);
  if (!g->plan_device || g->opt.path == FLAME_HIP_PATH_GLOBAL ||
      !DevPlanner::eligible(g->opt, V, E, T, sz.tile_own, sz.depth, sz.single, g->opt.lds_bytes))
    return 0;
  hipStream_t s = g->stream;
  int rc;
  const bool timing = g->opt.timing != 0;  // (option "plan_timing")
  auto tprev = std::chrono::steady_clock::now();
  auto lap = [&](const char* what) {
    if (!timing) return;
    (void)hipStreamSynchronize(s);
    const auto now = std::chrono::steady_clock::now();
    std::fprintf(stderr, "[upload] %-14s %7.3f ms\n", what, std::chrono::duration<double, std::milli>(now - tprev).count());
    tprev = now;
  };
  // ---- stage the caller's arrays ----
  if (staged) {
    // (g->dflags was zeroed by flame_hip_graph_sync and carries the non-finite check of the kernels
    // that derived z / wgt / x0 / alpha there)
  } else {
  if ((rc = dev_alloc(g->caps, &g->in_pos, (size_t)V)) || (rc = dev_alloc(g->caps, &g->in_edges, (size_t)E)) ||
      (rc = dev_alloc(g->caps, &g->in_alpha, (size_t)E)) || (rc = dev_alloc(g->caps, &g->in_beta, (size_t)E)) ||
      (rc = dev_alloc(g->caps, &g->in_z, (size_t)V)) || (rc = dev_alloc(g->caps, &g->in_wgt, (size_t)V)) ||
      (rc = dev_alloc(g->caps, &g->in_x0, (size_t)V)) || (rc = dev_alloc(g->caps, &g->in_tris, 3 * (size_t)T)) ||
      (rc = dev_alloc(g->caps, &g->dflags, 8)))
    return rc;
  // Only the positions are needed by the partition stages: they go first; the other arrays are
  // staged by `stage_rest` once those stages are enqueued (plan_dev.hip "after_partition"), so the
  // host-side staging copies overlap the bisection kernels.
  HIPCHK(hipMemsetAsync(g->dflags, 0, 8 * sizeof(int32_t), s));
  HIPCHK(hipMemcpyAsync(g->in_pos, pos, sizeof(float2) * (size_t)V, hipMemcpyHostToDevice, s));
  HIPCHK(launch_check_finite(s, 2 * (int64_t)V, reinterpret_cast<const float*>(g->in_pos), g->dflags));
  }
  lap("stage inputs");
  // ---- plan arrays ----
  if ((rc = dev_alloc(g->caps, &g->v_i2o_dev, (size_t)V)) || (rc = dev_alloc(g->caps, &g->v_o2i_dev, (size_t)V)) ||
      (rc = dev_alloc(g->caps, &g->e_i2o_dev, (size_t)E)) || (rc = dev_alloc(g->caps, &g->e_o2i_dev, (size_t)E)) ||
      (rc = dev_alloc(g->caps, &g->eij, (size_t)E)) || (rc = dev_alloc(g->caps, &g->ew, (size_t)E)) ||
      (rc = dev_alloc(g->caps, &g->grow, (size_t)V + 1)) || (rc = dev_alloc(g->caps, &g->ginc, 2 * (size_t)E)) ||
      (rc = dev_alloc(g->caps, &g->tris, 3 * (size_t)T)) || (rc = dev_alloc(g->caps, &g->trow, (size_t)V + 1)) ||
      (rc = dev_alloc(g->caps, &g->tinc, 3 * (size_t)T)))
    return rc;
  DevPlanInputs in;
  in.pos = g->in_pos; in.edges = g->in_edges; in.alpha = g->in_alpha;
  in.beta = (staged && g->beta_is_alpha) ? g->in_alpha : g->in_beta;
  in.tris = T > 0 ? g->in_tris : nullptr;
  DevPlanArrays A;
  A.v_o2i = g->v_o2i_dev; A.v_i2o = g->v_i2o_dev; A.e_o2i = g->e_o2i_dev; A.e_i2o = g->e_i2o_dev;
  A.eij = g->eij; A.ew = g->ew; A.grow = g->grow; A.ginc = g->ginc;
  A.tris = g->tris; A.trow = g->trow; A.tinc = g->tinc;
  TileAllocCtx ctx{g, &A};

  bool rest_staged = staged;
  auto stage_rest = [&]() -> hipError_t {
    if (rest_staged) return hipSuccess;
    rest_staged = true;
    hipError_t e;
#define STG(expr) do { e = (expr); if (e != hipSuccess) return e; } while (0)
    // on the staging stream: the (host-synchronous) pageable copies do not wait for the partition
    // kernels queued on the solve stream; that stream waits for the event instead
    if (E > 0) {
      STG(hipMemcpyAsync(g->in_edges, edges, sizeof(int2) * (size_t)E, hipMemcpyHostToDevice, g->stream_in));
      STG(hipMemcpyAsync(g->in_alpha, alpha, sizeof(float) * (size_t)E, hipMemcpyHostToDevice, g->stream_in));
      STG(hipMemcpyAsync(g->in_beta, beta, sizeof(float) * (size_t)E, hipMemcpyHostToDevice, g->stream_in));
    }
    STG(hipMemcpyAsync(g->in_z, z, sizeof(float) * (size_t)V, hipMemcpyHostToDevice, g->stream_in));
    STG(hipMemcpyAsync(g->in_wgt, wgt, sizeof(float) * (size_t)V, hipMemcpyHostToDevice, g->stream_in));
    if (x0) STG(hipMemcpyAsync(g->in_x0, x0, sizeof(float) * (size_t)V, hipMemcpyHostToDevice, g->stream_in));
    if (T > 0) STG(hipMemcpyAsync(g->in_tris, tris, sizeof(int32_t) * 3 * (size_t)T, hipMemcpyHostToDevice, g->stream_in));
    STG(hipEventRecord(g->ev_in, g->stream_in));
    STG(hipStreamWaitEvent(s, g->ev_in, 0));
    // non-finite inputs are found on the device (the flag is read with the builder's first sync)
    STG(DevPlanner::check_finite(s, g->dflags, g->in_z, V, g->in_wgt, V, g->in_alpha, E, g->in_beta, E,
                                 x0 ? g->in_x0 : nullptr, V));
#undef STG
    return hipSuccess;
  };
  // ---- state: written INSIDE the build, right behind the vertex order (stage B) -- the launch is then off
  // the path between the builder's round trip and the first iteration (a retry writes it again) ----
  for (int b = 0; b < 2; ++b) {
    if ((rc = dev_alloc(g->caps, &g->A[b], V)) || (rc = dev_alloc(g->caps, &g->B[b], V)) ||
        (rc = dev_alloc(g->caps, &g->q[b], E)))
      return rc;
  }
  if ((rc = dev_alloc(g->caps, &g->pos, V))) return rc;
  auto after_order = [&]() -> hipError_t {
    hipError_t e = stage_rest();
    if (e != hipSuccess) return e;
    return launch_init_state(s, V, g->v_i2o_dev, g->in_pos, g->in_z, g->in_wgt, have_x0 ? g->in_x0 : nullptr, g->A[0],
                             g->B[0], g->pos, E > 0 ? E : 1, g->q[0], g->q[1]);
  };
  int tile_own = sz.tile_own;
  int depth = sz.depth;
  bool fat = sz.fat;
  bool slot12 = false, fat_cfg = false, fat_s12 = false;
  bool balanced = false, built = false;
  int refine_left = 0;
  int ntiles = 0, cfg_nt = 0, cfg_ept = 0, cfg_vpt = 0;
  int64_t lds_max = 0;
  std::vector<TileDesc>& tiles = P.tiles;
  // A frame stream re-uses the previous frame's PARTITION (the tile of every spatial cell) while the
  // frames stay alike: no sorts, no bisection (option "plan_reuse", default on; results are the
  // oracle's bits on any partition).  One try; a frame it does not suit is bisected as usual.
  bool try_reuse = g->plan_reuse && g->opt.balance && g->planner.map_usable(V, depth) &&
                   g->opt.tile_own == g->reuse_tile_own_opt;
  if (try_reuse && g->reuse_skip > 0) { --g->reuse_skip; try_reuse = false; }  // back-off after rejections
  g->plan_reused = false;
  const int max_attempts = 10 + kBalanceRefinePasses + (sz.fat ? 7 * (2 + kBalanceRefinePasses) : 0);  // (plan.cpp)
  for (int attempt = 0; attempt < max_attempts && !built; ++attempt) {
    const bool reusing = try_reuse;
    try_reuse = false;
    ntiles = reusing ? g->planner.map_tiles() : (V + tile_own - 1) / tile_own;
    if (ntiles < 2) return 0;
    // every retry halves tile_own: the tile count may have outgrown the builder's segment tables
    // (kSegCap) -- the host builder shrinks safely in that case
    if (attempt > 0 && !reusing && !DevPlanner::eligible(g->opt, V, E, T, tile_own, depth, false, g->opt.lds_bytes)) return 0;
    if (reusing) {
      g->planner.reuse_partition();
    } else if (!balanced) {
      if (g->opt.balance && ntiles >= 16 && g->planner.grid_tiles() == ntiles) {
        g->planner.set_weights_from_grid();  // a frame stream balances in ONE pass
        balanced = true;
      } else {
        g->planner.set_weights_none();
      }
    }
    bool ok = false, index_error = false;
    int32_t uflags[4] = {0, 0, 0, 0};  // the finite-check word, the derived edge count (and the flags of an edge
                                       // derivation beside the build) ride on the builder's first sync
    g->planner.expect_edges(g->spec_edges ? E : -1);
    HIPCHK(g->planner.build(s, g->opt, V, E, T, ntiles, depth, in, &A, alloc_tile_arrays, &ctx, &tiles, &ok,
                            &index_error, g->dflags, uflags, after_order));
    g->planner.expect_edges(-1);
    if (g->spec_edges && uflags[1] != E) { g->true_edges = uflags[1]; return 2; }  // the predicted edge count was wrong
    if (uflags[0] & 1) return FLAME_HIP_ERR_NAN;
    if (index_error) return FLAME_HIP_ERR_ARG;
    const bool tiles_valid = ok;  // every tile was built (it may still be too large for LDS / a kernel config)
    if (ok) {
      const TileFit fit = tile_fit(g->opt, fat, tiles, fat_s12);  // (LDS, kernel configuration, 12-byte slots: plan.cpp)
      ok = fit.ok;
      lds_max = fit.lds_bytes; slot12 = fit.slot12; fat_cfg = fit.fat;
      cfg_nt = fit.nt; cfg_ept = fit.ept; cfg_vpt = fit.vpt;
      // A stream that solves by ONE launch of resident tiles takes over the previous frame's partition only if this frame
      // can be resident on it.  One hull tile of THIS frame with a halo twice the others' (the old partition knows nothing
      // of its long edges) is enough for a configuration the resident kernels do not have, and the whole solve went by
      // launches (r05: every 4th frame of the 50 k stream, 2.8 ms instead of 1.3): such a frame is bisected anew.
      if (ok && reusing && g->opt.resident && ntiles >= 2 && ntiles <= std::min(kPersistMaxTiles, g->num_cus)) {
        size_t stage = 0;
        for (const TileDesc& D : tiles) stage = std::max(stage, persist_stage_bytes(D));
        if (!(slot12 ? tile_slot12_exists(cfg_nt, cfg_ept, cfg_vpt) : tile_persist_exists(cfg_nt, cfg_ept, cfg_vpt)) ||
            (size_t)lds_max + stage + (size_t)kTileLdsReserve > (size_t)g->opt.lds_bytes)
          ok = false;
      }
    }
    if (reusing) {
      if (ok) { built = true; g->plan_reused = true; g->reuse_backoff = 0; break; }
      g->planner.drop_map();  // scene change / does not fit: exact bisection from here on
      g->reuse_backoff = std::min(16, std::max(1, 2 * g->reuse_backoff));  // a wasted attempt costs a build:
      g->reuse_skip = g->reuse_backoff;                                     // try again 1, 2, 4 ... 16 frames later
      balanced = false;
      continue;
    }
    // Only the LARGEST tile decides whether a partition fits, and before the cost balance that is a
    // border tile (long hull edges => a halo up to 1.5 x the median): balance first, shrink only
    // if the balanced partition does not fit either.
    if (!ok && tiles_valid && g->opt.balance && !balanced && ntiles >= 16) {
      balanced = true;
      refine_left = kBalanceRefinePasses;
      HIPCHK(g->planner.weights_from_tiles(s, V, A));
      continue;
    }
    if (ok && g->opt.balance && !balanced && ntiles >= 16) {
      balanced = true;
      refine_left = kBalanceRefinePasses;
      HIPCHK(g->planner.weights_from_tiles(s, V, A));  // second, cost-weighted pass, same tile count
      continue;
    }
    if (ok && balanced && refine_left > 0 && ntiles >= 16) {  // refinement passes (plan.cpp)
      --refine_left;
      long long total = 0;
      for (const TileDesc& D : tiles) total += tile_cost(D);
      HIPCHK(g->planner.weights_scale_by_tiles(s, V, ntiles, total, A));
      continue;
    }
    if (ok) { built = true; break; }
    // did not fit: smaller tiles, plain bisection again
    g->planner.drop_grid();
    balanced = false;
    refine_left = 0;
    if (fat && fat_next_attempt(g->opt, sz, &depth, &fat_s12)) {}  // fat tiles: shallower / 12-byte slots first (plan.cpp)
    else if (fat) { fat = false; tile_own = sz.fallback_own; depth = sz.fallback_depth; }
    else if (regular_next_attempt(g->opt, sz, V, &tile_own, &depth)) {}  // (plan.cpp)
    else tile_own = std::max(16, tile_own / 2);
  }
  if (!built) return 0;
  lap("plan build");
  {  // how even the tiles are (a launch lasts as long as its slowest tile): max / mean of the cost model
    long long sum = 0, mx = 0;
    for (const TileDesc& D : tiles) {
      const long long c = tile_cost(D);
      sum += c; mx = std::max(mx, c);
    }
    g->tile_imbalance_pct = sum > 0 ? (int)(100 * mx * (long long)tiles.size() / sum) : 100;
  }
  if (g->opt.balance && ntiles >= kMapMinTiles) {  // (the cost grid only serves >= 16 tiles, the tile map any count)
    HIPCHK(g->planner.update_grid(s, V, ntiles, in, A));  // cost-density grid + the tile map of this frame
    g->planner.set_map_depth(depth);
    g->reuse_tile_own_opt = g->opt.tile_own;
  } else {
    g->planner.drop_map();
  }
  // ---- host-side description of the plan ----
  P.V = V; P.E = E; P.T = T;
  P.on_device = true;
  P.has_tiles = true;
  P.tile_threads = cfg_nt; P.tile_ept = cfg_ept; P.tile_vpt = cfg_vpt;
  P.tile_depth = depth;
  P.tile_lds_bytes = lds_max;
  P.tile_slot12 = slot12;
  P.tile_fat = fat_cfg;
  P.note.clear();
  P.v_o2i.clear(); P.v_i2o.clear(); P.e_o2i.clear(); P.e_i2o.clear();
  P.eij.clear(); P.ew.clear(); P.grow.clear(); P.ginc.clear(); P.tris.clear(); P.trow.clear(); P.tinc.clear();
  P.t_vmap.clear(); P.t_emap.clear(); P.t_eij.clear(); P.t_ew.clear(); P.t_srow.clear();
  g->host_perms = false;
  g->path = FLAME_HIP_PATH_TILE;
  if (!(slot12 ? tile_slot12_exists(cfg_nt, cfg_ept, cfg_vpt) : tile_config_exists(cfg_nt, cfg_ept, cfg_vpt))) return FLAME_HIP_ERR_STATE;
  HIPCHK(prepare_tile(cfg_nt, cfg_ept, cfg_vpt, (size_t)lds_max, slot12));
  g->cur = 0;  // (the state was written inside the build: after_order)
  if ((rc = dev_alloc(g->caps, &g->tri_normals, (size_t)T)) || (rc = dev_alloc(g->caps, &g->tri_valid, (size_t)T)))
    return rc;
  if (T == 0) HIPCHK(hipMemsetAsync(g->trow, 0, sizeof(int32_t) * ((size_t)V + 1), s));
  lap("grid+state");
  return 1;
}

// Host copies of the permutations of a device-built plan, fetched on demand (set_state,
// update_data, halo lists, mesh faces, the debug hook; the per-frame path does not need them).
static int ensure_host_perms(flame_hip_graph* g) {
  if (g->host_perms || !g->plan.on_device) return 0;
  Plan& P = g->plan;
  const size_t V = (size_t)g->V, E = (size_t)g->E, T3 = 3 * (size_t)P.T;
  P.v_i2o.resize(V); P.v_o2i.resize(V); P.e_i2o.resize(E); P.e_o2i.resize(E); P.tris.resize(T3);
  hipStream_t s = g->stream;
  if (V) {
    HIPCHK(hipMemcpyAsync(P.v_i2o.data(), g->v_i2o_dev, 4 * V, hipMemcpyDeviceToHost, s));
    HIPCHK(hipMemcpyAsync(P.v_o2i.data(), g->v_o2i_dev, 4 * V, hipMemcpyDeviceToHost, s));
  }
  if (E) {
    HIPCHK(hipMemcpyAsync(P.e_i2o.data(), g->e_i2o_dev, 4 * E, hipMemcpyDeviceToHost, s));
    HIPCHK(hipMemcpyAsync(P.e_o2i.data(), g->e_o2i_dev, 4 * E, hipMemcpyDeviceToHost, s));
  }
  if (T3) HIPCHK(hipMemcpyAsync(P.tris.data(), g->tris, 4 * T3, hipMemcpyDeviceToHost, s));
  HIPCHK(hipStreamSynchronize(s));
  g->host_perms = true;
  return 0;
}

int flame_hip_graph_upload(flame_hip_graph* g, const float* pos, const int32_t* edges,
                           const float* alpha, const float* beta, const float* z,
                           const float* wgt, const float* x0, const int32_t* tris) {
  RoctxRange roctx_("flame_hip_graph_upload");
  if (!g) return FLAME_HIP_ERR_ARG;
  const int32_t V = g->V, E = g->E;
  if (edges != g->sync.edges.data()) g->synced = false;
  g->sync_on_device = false;
  g->beta_is_alpha = false;
  if ((V > 0 && (!pos || !z || !wgt)) || (E > 0 && (!edges || !alpha || !beta)))
    return FLAME_HIP_ERR_ARG;
  g->uploaded = false;
  const GraphOptScope opt_scope(g, V);
  int rc;
  auto device_plan = [&]() -> int {
    int r = upload_device_plan(g, pos, edges, alpha, beta, z, wgt, x0, tris);
    if (r <= 0) {  // not built there: copies from the caller's arrays may still be in flight
      (void)hipStreamSynchronize(g->stream);
      (void)hipStreamSynchronize(g->stream_in);
    }
    return r;
  };
  if (g->device >= 0) {
    HIPCHK(hipSetDevice(g->device));
    HIPCHK(wait_last_solve(g));
    HIPCHK(hipStreamSynchronize(g->stream));
    persist_discard(g);
    HIPCHK(hipStreamSynchronize(g->stream_in));
    HIPCHK(g->planner.wait_maps());  // (the previous frame's maps read its positions / tiles on the builder's stream)
    g->drop_execs();  // captured launches hold the old grid / pointers
    g->solves_since_upload = 0;
    g->lanes_applied = false;
    rc = device_plan();
    if (rc < 0) return rc;
  } else {
    rc = 0;
  }
  bool dev_plan = rc == 1;
  if (!dev_plan) {
    if (!all_finite(pos, 2 * (size_t)V) || !all_finite(z, V) || !all_finite(wgt, V) ||
        !all_finite(alpha, E) || !all_finite(beta, E) || (x0 && !all_finite(x0, V)))
      return FLAME_HIP_ERR_NAN;
    g->plan.on_device = false;
    g->host_perms = true;
    // An isolated tile that was sized from (V, E) alone may not fit after all (its slot rows follow
    // the degrees).  The host builder's own answer is a halo'd partition built on the host -- 4 ms at
    // 1.3 k vertices; when the choice was automatic the graph goes to the device builder instead, and
    // the handle remembers the size (flame_hip_get_info "single_cap").
    const bool auto_single = g->device >= 0 && g->plan_device && g->opt.tile_own <= 0 && g->opt.batch_voff.empty() &&
                             plan_sizing(g->opt, V, E).single;
    g->opt.single_only = auto_single;
    rc = build_plan(g->opt, V, E, g->T, pos, edges, alpha, beta, tris, &g->plan);
    g->opt.single_only = false;
    if (rc == kPlanSingleNoFit) {
      g->single_cap = std::min(g->single_cap, std::max(V - 1, 0));
      g->opt.single_max = std::min(g->opt.single_max, g->single_cap);  // (restored by opt_scope)
      rc = device_plan();
      if (rc < 0) return rc;
      dev_plan = rc == 1;
      if (!dev_plan) {
        g->plan.on_device = false;
        g->host_perms = true;
        rc = build_plan(g->opt, V, E, g->T, pos, edges, alpha, beta, tris, &g->plan);
      }
    }
    if (!dev_plan && rc != 0) return rc;
  }
  const Plan& P = g->plan;
  if (!dev_plan) {
    g->path = (P.has_tiles && g->opt.path != FLAME_HIP_PATH_GLOBAL) ? FLAME_HIP_PATH_TILE
                                                                     : FLAME_HIP_PATH_GLOBAL;
    if (g->opt.path == FLAME_HIP_PATH_TILE && !P.has_tiles) return FLAME_HIP_ERR_ARG;
  }
  if (g->device < 0) {  // plan-only handle (host-logic tests): no device work
    g->uploaded = true;
    return 0;
  }
  if (!dev_plan) {
    if (P.has_tiles) {
      if (!(P.tile_slot12 ? tile_slot12_exists(P.tile_threads, P.tile_ept, P.tile_vpt)
                          : tile_config_exists(P.tile_threads, P.tile_ept, P.tile_vpt)))
        return FLAME_HIP_ERR_STATE;
      HIPCHK(prepare_tile(P.tile_threads, P.tile_ept, P.tile_vpt, (size_t)P.tile_lds_bytes, P.tile_slot12));
    }
    // initial state in internal order
    std::vector<float4> hA(V), hB(V);
    std::vector<float2> hpos(V);
    for (int32_t k = 0; k < V; ++k) {
      const int32_t o = P.v_i2o[k];
      const float xi = x0 ? x0[o] : z[o];
      hA[k] = make_float4(xi, 0.f, 0.f, z[o]);
      hB[k] = make_float4(xi, 0.f, 0.f, wgt[o]);
      hpos[k] = make_float2(pos[2 * o], pos[2 * o + 1]);
    }
    static_assert(sizeof(Int2) == sizeof(int2) && sizeof(Float4) == sizeof(float4) &&
                      sizeof(UInt2) == sizeof(uint2), "layout");
    // Everything the host built goes to the GPU as ONE transfer: the arrays are laid out back to
    // back in the handle's page-locked arena, copied with a single DMA into one device arena, and
    // the handle's array pointers point into it.  (Nineteen separate small copies cost the GPU
    // ~0.1 ms and the host ~0.05 ms of a 0.8 ms frame at TUM size.)
    struct Piece { void** member; const void* src; size_t bytes; size_t off; };
    std::vector<Piece> pieces;
    size_t total = 0;
    auto add = [&](auto** member, const void* src, size_t bytes) {
      pieces.push_back({reinterpret_cast<void**>(member), src, bytes, total});
      total += (bytes + 16 + 255) & ~(size_t)255;  // one pad element behind every array, 256-byte aligned
    };
    add(&g->A[0], hA.data(), sizeof(float4) * (size_t)V);
    add(&g->B[0], hB.data(), sizeof(float4) * (size_t)V);
    add(&g->pos, hpos.data(), sizeof(float2) * (size_t)V);
    add(&g->eij, P.eij.data(), sizeof(int2) * (size_t)E);
    add(&g->ew, P.ew.data(), sizeof(float4) * (size_t)E);
    add(&g->ginc, P.ginc.data(), sizeof(int32_t) * 2 * (size_t)E);
    add(&g->grow, P.grow.data(), sizeof(int32_t) * ((size_t)V + 1));
    if (P.has_tiles) {
      add(&g->tiles, P.tiles.data(), sizeof(TileDesc) * P.tiles.size());
      add(&g->t_vmap, P.t_vmap.data(), sizeof(int32_t) * P.t_vmap.size());
      add(&g->t_emap, P.t_emap.data(), sizeof(int32_t) * P.t_emap.size());
      add(&g->t_srow, P.t_srow.data(), sizeof(uint32_t) * P.t_srow.size());
      add(&g->t_eij, P.t_eij.data(), sizeof(uint2) * P.t_eij.size());
      add(&g->t_ew, P.t_ew.data(), sizeof(float4) * P.t_ew.size());
    }
    // the triangle arrays always exist: with T == 0 the vertex -> triangle CSR is all-empty rows, so
    // the triangle stage, mesh and dense maps run (degenerate normals, nothing covered)
    std::vector<int32_t> zero_rows;
    if (P.T <= 0) zero_rows.assign((size_t)V + 1, 0);
    add(&g->tris, P.tris.data(), sizeof(int32_t) * P.tris.size());
    add(&g->tinc, P.tinc.data(), sizeof(int32_t) * P.tinc.size());
    add(&g->trow, P.T > 0 ? P.trow.data() : zero_rows.data(), sizeof(int32_t) * ((size_t)V + 1));
    // permutations for the device-side result paths
    add(&g->v_i2o_dev, P.v_i2o.data(), sizeof(int32_t) * (size_t)V);
    add(&g->v_o2i_dev, P.v_o2i.data(), sizeof(int32_t) * (size_t)V);
    add(&g->e_o2i_dev, P.e_o2i.data(), sizeof(int32_t) * (size_t)E);
    HIPCHK(g->pin.reserve(total));
    if ((rc = dev_alloc(g->caps, &g->harena, total))) return rc;
    for (const Piece& pc : pieces) {
      if (pc.bytes) std::memcpy(g->pin.base + pc.off, pc.src, pc.bytes);
      auto it = g->caps.find((void*)pc.member);  // a buffer of its own from an earlier (device-built) frame
      if (it != g->caps.end()) { if (*pc.member) (void)hipFree(*pc.member); g->caps.erase(it); }
      *pc.member = g->harena + pc.off;
    }
    HIPCHK(hipMemcpyAsync(g->harena, g->pin.base, total, hipMemcpyHostToDevice, g->stream));
    if ((rc = dev_alloc(g->caps, &g->A[1], V)) || (rc = dev_alloc(g->caps, &g->B[1], V))) return rc;
    for (int b = 0; b < 2; ++b) {
      if ((rc = dev_alloc(g->caps, &g->q[b], E))) return rc;
      HIPCHK(hipMemsetAsync(g->q[b], 0, sizeof(float4) * (size_t)(E > 0 ? E : 1), g->stream));
    }
    g->cur = 0;
    if ((rc = dev_alloc(g->caps, &g->tri_normals, (size_t)P.T)) || (rc = dev_alloc(g->caps, &g->tri_valid, (size_t)P.T)))
      return rc;
    // (no synchronisation here: the transfer reads the page-locked arena, which is only rewritten by
    // the next upload -- behind that upload's own stream synchronisation; the solve is ordered behind
    // the copy on the stream, a caller's stream through the state event of finish_upload)
  }
  return finish_upload(g);
}

// buffers every uploaded graph needs, whichever builder made its plan
static int finish_upload(flame_hip_graph* g) {
  g->poll_valid = false;  // (new tile arrays: the resident tiles' poll lists are rebuilt at their next launch)
  const int32_t V = g->V, E = g->E;
  // (the round time the resident launches' time-out follows was measured on ANOTHER graph -- a frame stream's
  // next frame is about as large and keeps it, a graph of another size starts again from the 4 ms of a first launch)
  if (g->persist_round_V > 0 && (V > 2 * g->persist_round_V || 2 * V < g->persist_round_V)) { g->persist_round_us = 0.f; g->persist_round_V = 0; }
  const Plan& P = g->plan;
  int rc;
  if ((rc = dev_alloc(g->caps, &g->vtx_normals, (size_t)V))) return rc;
  if ((rc = dev_alloc(g->caps, &g->mesh_pts, 3 * (size_t)V))) return rc;
  if ((rc = dev_alloc(g->caps, &g->dl_v, 3 * (size_t)V)) || (rc = dev_alloc(g->caps, &g->dl_q, 3 * (size_t)E)) ||
      (rc = dev_alloc(g->caps, &g->dl_n, 3 * (size_t)V)))
    return rc;
  if ((rc = dev_alloc(g->caps, &g->partials, 2 * (size_t)costs_num_blocks(V, E)))) return rc;
  if (g->profile && P.has_tiles) {
    if ((rc = dev_alloc(g->caps, &g->prof, P.tiles.size() * kProfWords))) return rc;
    HIPCHK(hipMemsetAsync(g->prof, 0, sizeof(unsigned long long) * P.tiles.size() * kProfWords, g->stream));
  }
  g->uploaded = true;
  g->timed = false;
  g->state_scale = 1.0f;
  HIPCHK(mark_state(g));
  return 0;
}

static int require_device(const flame_hip_graph* g);

int32_t flame_hip_feature_gate(int32_t n, const float* idepth_var, float var_max, uint8_t* keep) {
  if (n < 0 || (n > 0 && (!idepth_var || !keep))) return FLAME_HIP_ERR_ARG;
  int32_t cnt = 0;
  for (int32_t v = 0; v < n; ++v) { keep[v] = idepth_var[v] < var_max ? 1 : 0; cnt += keep[v]; }
  return cnt;
}

int flame_hip_graph_sync(flame_hip_graph* g, const flame_hip_sync_params* sp, int32_t V, int32_t T,
                         const float* pos, const float* idepth_mu, const float* idepth_var,
                         const int32_t* tris, const float* prediction, float* scale) {
  RoctxRange roctx_("flame_hip_graph_sync");
  if (!g || !sp || V < 0 || T < 0) return FLAME_HIP_ERR_ARG;
  const int32_t* tris_dev = nullptr;  // ... the same list on the device (r05: the device paths never take it through the host)
  if (T > 0 && !tris) {  // the triangulation flame_hip_delaunay made last on this handle, read where the library left it
    if (g->dt.last_T != T || g->dt.last_V != V || !g->dt.last_list) return FLAME_HIP_ERR_ARG;
    tris = g->dt.last_list;  // (page-locked; in keep mode its copy may still be in flight on the staging stream: DMAs queued
    tris_dev = g->dt.last_dev;  // there are ordered behind it, a HOST read has to wait -- dt.host_list())
  }
  if (V > 0 && (!pos || !idepth_mu || !idepth_var)) return FLAME_HIP_ERR_ARG;
  const auto t_entry = std::chrono::steady_clock::now();
  const GraphOptScope opt_scope(g, V);
  // input validation on the host: non-finite positions / idepths, NaN variances, the variance gate.
  // On the device path it runs WHILE the GPU already derives the edges (every kernel there is safe on
  // unvalidated input: indices are range-checked, non-finite values only set a flag).
  auto validate = [&]() -> int {
    if (!all_finite(pos, 2 * (size_t)V) || !all_finite(idepth_mu, V)) return FLAME_HIP_ERR_NAN;
    for (int32_t v = 0; v < V; ++v) {
      if (std::isnan(idepth_var[v])) return FLAME_HIP_ERR_NAN;
      if (!(idepth_var[v] < sp->idepth_var_max_graph)) return FLAME_HIP_ERR_ARG;  // fails the gate
    }
    return 0;
  };
  int rc;
  g->sync_on_device = false;
  // graphs that become one isolated tile (host plan) are synced on the host as well (E <= 3V bounds
  // the edge count of a triangulation before it is known)
  const bool lone_tile = plan_sizing(g->opt, V, (int32_t)std::min<int64_t>(3ll * V, INT32_MAX)).single;
  if (sp->edge_weight_rule < 0 || sp->edge_weight_rule > 3 || !std::isfinite(sp->alpha_gain) || !std::isfinite(sp->beta_gain))
    return FLAME_HIP_ERR_ARG;
  if (g->device >= 0 && g->plan_device && V >= 2 && T > 0 && !lone_tile && g->opt.path != FLAME_HIP_PATH_GLOBAL &&
      !sync_weights_custom(*sp)) {
    // ---- on the device: edges of the triangulation, alpha, data terms; then the device plan ----
    HIPCHK(hipSetDevice(g->device));
    HIPCHK(wait_last_solve(g));
    hipStream_t s = g->stream;
    HIPCHK(hipStreamSynchronize(s));
    persist_discard(g);
    HIPCHK(g->planner.wait_maps());
    if ((rc = dev_alloc(g->caps, &g->in_pos, (size_t)V)) || (rc = dev_alloc(g->caps, &g->in_tris, 3 * (size_t)T)) ||
        (rc = dev_alloc(g->caps, &g->in_mu, (size_t)V)) || (rc = dev_alloc(g->caps, &g->in_var, (size_t)V)) ||
        (rc = dev_alloc(g->caps, &g->in_pred, (size_t)V)) || (rc = dev_alloc(g->caps, &g->in_z, (size_t)V)) ||
        (rc = dev_alloc(g->caps, &g->in_wgt, (size_t)V)) || (rc = dev_alloc(g->caps, &g->in_x0, (size_t)V)) ||
        (rc = dev_alloc(g->caps, &g->in_edges, 3 * (size_t)T)) || (rc = dev_alloc(g->caps, &g->in_alpha, 3 * (size_t)T)))
      return rc;
    const bool timing = g->opt.timing != 0;  // (option "plan_timing")
    auto t_prev = t_entry;
    auto lap = [&](const char* what) {
      if (!timing) return;
      const auto now = std::chrono::steady_clock::now();
      std::fprintf(stderr, "[sync] %-16s %7.3f ms\n", what, std::chrono::duration<double, std::milli>(now - t_prev).count());
      t_prev = now;
    };
    lap("checks+allocs");
    if ((rc = dev_alloc(g->caps, &g->dflags, 8))) return rc;
    HIPCHK(hipMemsetAsync(g->dflags, 0, 8 * sizeof(int32_t), s));
    // The copies from the caller's (pageable) arrays block the host, so they are issued in the order
    // the kernels need them, each batch of kernels enqueued before the next copy starts: triangles ->
    // half-edge count / scan / fill; positions -> unique edges + alpha; idepths -> data terms.
    // They go through the staging stream (a copy on the solve stream would wait for the kernels queued
    // there, and the host with it); the solve stream waits for an event behind each batch.
    hipStream_t sin = g->stream_in;
    auto stage = [&](void* dst, const void* src, size_t bytes) { return hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, sin); };
    auto staged_for = [&]() {
      hipError_t e = hipEventRecord(g->ev_in, sin);
      return e != hipSuccess ? e : hipStreamWaitEvent(s, g->ev_in, 0);
    };
    // (Copying a small frame's inputs into a page-locked arena first and sending them by asynchronous
    // DMAs was measured: 0.585 vs 0.535 ms per 1.2 k frame -- the first kernel then waits for ALL the
    // inputs, while the pipelined pageable copies below keep the GPU busy from the triangles on.)
    hipError_t stage_err = hipSuccess;
    int32_t E = 0;
    bool index_error = false;
    int vrc = 0;
    float sc = 1.0f;
    // The edge count of a triangulation is known before the device has derived the edges: Euler's
    // formula, E = V + T - 1 for a triangulated disk (what a Delaunay triangulation is); for meshes
    // with holes / several components the offset of the previous frame.  The frame goes on without
    // the round trip; the plan builder's first synchronisation brings the true count and a wrong
    // guess only costs a second build (then the handle waits 1, 2, 4 ... frames before guessing again).
    int32_t expected_E = -1;
    if (g->euler_skip > 0) --g->euler_skip;
    else if ((int64_t)V + T + g->euler_off > 0 && (int64_t)V + T + g->euler_off <= 3ll * T) expected_E = V + T + g->euler_off;
    const bool use_pred = sp->init_with_prediction && prediction;
    // ---- small frames: one launch for everything in front of the tile pass (plan_dev.hip k_mini_plan) ----
    g->plan_mini_used = false;
    if (g->plan_mini && expected_E >= 0 && DevPlanner::mini_eligible(V, T, expected_E) && g->plan_reuse && g->opt.balance &&
        g->reuse_skip == 0 && g->opt.tile_own == g->reuse_tile_own_opt &&
        g->planner.map_usable(V, plan_sizing(g->opt, V, expected_E).depth)) {
      if ((vrc = validate())) return vrc;  // (a few microseconds at this size)
      if (sp->rescale_data) {  // mean in the oracle's order (sequential, float64)
        double acc = 0.0;
        for (int32_t v = 0; v < V; ++v) acc += (double)idepth_mu[v];
        sc = (float)(acc / (double)V);
        if (!(sc > 0.0f)) sc = 1.0f;
      }
      // The one launch needs every input before it starts: five copies from pageable memory would be
      // ~60 us of host time in front of it.  The arrays (<= 100 KB here) are packed into a page-locked
      // arena and leave as ONE asynchronous DMA into a device staging buffer the launch reads from.
      auto al = [](size_t v) { return (v + 63) & ~(size_t)63; };
      const size_t o_tris = 0, o_pos = al(sizeof(int32_t) * 3 * (size_t)T), o_mu = o_pos + al(sizeof(float2) * (size_t)V);
      const size_t o_var = o_mu + al(sizeof(float) * (size_t)V), o_pred = o_var + al(sizeof(float) * (size_t)V);
      const size_t in_total = o_pred + (prediction ? al(sizeof(float) * (size_t)V) : 0);
      if ((rc = dev_alloc(g->caps, &g->in_stage, in_total))) return rc;
      HIPCHK(g->pin_in.reserve(in_total));
      if (!tris_dev) std::memcpy(g->pin_in.base + o_tris, tris, sizeof(int32_t) * 3 * (size_t)T);
      std::memcpy(g->pin_in.base + o_pos, pos, sizeof(float2) * (size_t)V);
      std::memcpy(g->pin_in.base + o_mu, idepth_mu, sizeof(float) * (size_t)V);
      std::memcpy(g->pin_in.base + o_var, idepth_var, sizeof(float) * (size_t)V);
      if (prediction) std::memcpy(g->pin_in.base + o_pred, prediction, sizeof(float) * (size_t)V);
      // ... or, measured faster still (1.2 k frame 0.452 -> 0.435 ms): no copy command at all, the launch
      // reads the arena through its device mapping (every input is read once, coalesced: ~100 KB over the
      // host link inside the kernel's first phase instead of a DMA + its dependency in front of it)
      const char* src = g->pin_in.base;
      lap("H2D all");
      DevPlanner::MiniSync ms;
      ms.tris = tris_dev ? tris_dev : reinterpret_cast<const int32_t*>(src + o_tris);  // (the library's own list: read where it lies)
      ms.pos = reinterpret_cast<const float2*>(src + o_pos);
      ms.mu = reinterpret_cast<const float*>(src + o_mu);
      ms.var = reinterpret_cast<const float*>(src + o_var);
      ms.pred = (use_pred) ? reinterpret_cast<const float*>(src + o_pred) : nullptr;
      ms.scale = sc;
      ms.adaptive = sp->adaptive_data_weights; ms.init_pred = sp->init_with_prediction;
      ms.z = g->in_z; ms.wgt = g->in_wgt; ms.x0 = g->in_x0; ms.edges = g->in_edges; ms.alpha = g->in_alpha;
      ms.dflags = g->dflags;
      g->planner.offer_mini(ms);
      const int32_t V0 = g->V, E0 = g->E, T0 = g->T;
      g->V = V; g->E = expected_E; g->T = T;
      g->uploaded = false;
      g->n_send_v = g->n_send_e = g->n_recv_v = g->n_recv_e = 0;
      g->drop_execs();
      g->solves_since_upload = 0;
      g->lanes_applied = false;
      g->beta_is_alpha = true;
      g->spec_edges = true;
      rc = upload_device_plan(g, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, true, true);
      g->spec_edges = false;
      g->planner.withdraw_mini();
      if (g->planner.mini_used() && rc == 1) {
        g->plan_mini_used = true;
        g->euler_backoff = 0;
        g->euler_off = (int32_t)((int64_t)g->E - V - T);
        if ((rc = finish_upload(g))) return rc;
        g->synced = true;
        g->sync_on_device = true;
        g->sync.scale = sc;
        if (scale) *scale = sc;
        return 0;
      }
      if (rc < 0) return rc;
      // not taken (the build did not reuse the partition after all), a wrong edge-count prediction or a
      // plan that does not fit: the usual way from the start (the inputs are staged again in its order)
      (void)hipStreamSynchronize(s);
      if (rc == 2 && g->planner.mini_used()) {  // (only the one-launch path derives the edges: without it rc == 2 says
        g->euler_backoff = std::min(16, std::max(1, 2 * g->euler_backoff));  // nothing about the prediction -- ADVICE r3)
        g->euler_skip = g->euler_backoff;
        expected_E = -1;
      }
      g->V = V0; g->E = E0; g->T = T0;
      HIPCHK(hipMemsetAsync(g->dflags, 0, 8 * sizeof(int32_t), s));
    }
    // The copies from the caller's (pageable) arrays block the host, so they are issued in the order the
    // kernels need them, each batch of kernels enqueued before the next copy starts: triangles ->
    // half-edge count / scan / fill; positions -> unique edges + alpha; idepths -> data terms.
    if (tris_dev) {  // the library's own list is complete on the device (flame_hip_delaunay returned behind its last kernel):
      // a device-to-device copy on the build's stream, nothing through the host link and nothing to wait for
      HIPCHK(hipMemcpyAsync(g->in_tris, tris_dev, sizeof(int32_t) * 3 * (size_t)T, hipMemcpyDeviceToDevice, s));
    } else {
      HIPCHK(stage(g->in_tris, tris, sizeof(int32_t) * 3 * (size_t)T));
      HIPCHK(staged_for());
    }
    lap("H2D triangles");
    HIPCHK(g->planner.edges_from_tris(s, V, T, g->in_tris, g->in_pos, g->in_edges, g->in_alpha, &E, &index_error, g->dflags,
                                      [&]() {
                                        stage_err = stage(g->in_mu, idepth_mu, sizeof(float) * (size_t)V);
                                        if (stage_err == hipSuccess) stage_err = stage(g->in_var, idepth_var, sizeof(float) * (size_t)V);
                                        if (stage_err == hipSuccess && prediction)
                                          stage_err = stage(g->in_pred, prediction, sizeof(float) * (size_t)V);
                                        if (stage_err == hipSuccess) stage_err = staged_for();
                                        vrc = validate();
                                        if (vrc == 0 && sp->rescale_data) {  // mean in the oracle's order (sequential, float64)
                                          double acc = 0.0;
                                          for (int32_t v = 0; v < V; ++v) acc += (double)idepth_mu[v];
                                          sc = (float)(acc / (double)V);
                                          if (!(sc > 0.0f)) sc = 1.0f;
                                        }
                                      }, expected_E,
                                      [&]() {
                                        hipError_t e = stage(g->in_pos, pos, sizeof(float2) * (size_t)V);
                                        return e != hipSuccess ? e : staged_for();
                                      }));
    HIPCHK(stage_err);
    lap("edges_from_tris");
    if (vrc || index_error) g->uploaded = false;  // (the staged inputs of the previous graph are gone)
    if (vrc) return vrc;
    if (index_error) return FLAME_HIP_ERR_ARG;
    HIPCHK(g->planner.sync_data(s, V, g->in_mu, g->in_var, use_pred ? g->in_pred : nullptr, sc, sp->adaptive_data_weights,
                                sp->init_with_prediction, g->in_z, g->in_wgt, g->in_x0, g->dflags));
    g->V = V; g->E = E; g->T = T;
    g->uploaded = false;
    g->n_send_v = g->n_send_e = g->n_recv_v = g->n_recv_e = 0;
    g->drop_execs();
    g->solves_since_upload = 0;
    g->lanes_applied = false;
    g->beta_is_alpha = true;
    g->spec_edges = expected_E >= 0;
    rc = upload_device_plan(g, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, true, true);
    g->spec_edges = false;
    if (rc == 2) {  // the predicted edge count was wrong: the device has the true one by now -- build again
      E = g->true_edges;
      g->euler_backoff = std::min(16, std::max(1, 2 * g->euler_backoff));
      g->euler_skip = g->euler_backoff;
      if (E < 0 || E > 3 * (int64_t)T) return FLAME_HIP_ERR_ARG;
      g->E = E;
      rc = upload_device_plan(g, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, true, true);
    } else if (expected_E >= 0) {
      g->euler_backoff = 0;
    }
    if (rc == 1 || rc == 0) g->euler_off = (int32_t)((int64_t)g->E - V - T);
    if (rc <= 0) (void)hipStreamSynchronize(s);  // the H2D copies of the caller's arrays end here
    if (rc < 0) return rc;
    if (rc == 1) {
      if ((rc = finish_upload(g))) return rc;
      g->synced = true;
      g->sync_on_device = true;
      g->sync.scale = sc;
      if (scale) *scale = sc;
      return 0;
    }
    // not eligible for the device plan (e.g. one isolated tile): fall through to the host builder
  }
  if ((rc = validate())) return rc;
  if (tris_dev && !(tris = g->dt.host_list())) return FLAME_HIP_ERR_HIP;  // (the host builder reads the list: its copy-out must be complete)
  rc = graph_sync_host(*sp, V, T, pos, idepth_mu, idepth_var, tris, prediction, &g->sync);
  if (rc) return rc;
  const int32_t E = (int32_t)(g->sync.edges.size() / 2);
  if ((rc = flame_hip_graph_resize(g, V, E, T))) return rc;
  const SyncOut& S = g->sync;
  rc = flame_hip_graph_upload(g, pos, S.edges.data(), S.alpha.data(), S.beta.empty() ? S.alpha.data() : S.beta.data(),
                              S.z.data(), S.wgt.data(), S.x0.data(), T > 0 ? tris : nullptr);
  if (rc) return rc;
  g->synced = true;
  if (scale) *scale = S.scale;
  return 0;
}

int flame_hip_delaunay(flame_hip_graph* g, int32_t V, const float* pos, int32_t tri_cap, int32_t* tris, int32_t* T) {
  RoctxRange roctx_("flame_hip_delaunay");
  if (!g || V < 0 || tri_cap < 0 || !T || (V > 0 && !pos) || (tri_cap > 0 && !tris)) return FLAME_HIP_ERR_ARG;
  if (g->device < 0) return FLAME_HIP_ERR_STATE;  // a plan-only handle has no GPU (the host triangulator is include/flame/utils/delaunay.h)
  if (V > 0 && !all_finite(pos, 2 * (size_t)V)) return FLAME_HIP_ERR_NAN;
  HIPCHK(hipSetDevice(g->device));
  return delaunay_device(g->stream_in, &g->dt, V, pos, tri_cap, tris, T);
}

// The list of the last flame_hip_delaunay on this handle (tri_cap >= its T).  After a call in keep mode (tri_cap = 0, tris =
// NULL) this is where the host copy is waited for: it travelled on a stream of its own while the caller went on.
int flame_hip_delaunay_list(flame_hip_graph* g, int32_t tri_cap, int32_t* tris) {
  if (!g || tri_cap < 0 || (tri_cap > 0 && !tris)) return FLAME_HIP_ERR_ARG;
  if (g->device < 0 || g->dt.last_T < 0) return FLAME_HIP_ERR_STATE;
  if (g->dt.last_T > tri_cap) return FLAME_HIP_ERR_ARG;
  HIPCHK(hipSetDevice(g->device));
  const int32_t* h = g->dt.host_list();
  if (!h) return FLAME_HIP_ERR_HIP;
  if (g->dt.last_T > 0) std::memcpy(tris, h, sizeof(int32_t) * 3 * (size_t)g->dt.last_T);
  return 0;
}

int flame_hip_graph_edges(const flame_hip_graph* g, int32_t* edges) {
  if (!g || !g->uploaded) return FLAME_HIP_ERR_STATE;
  if (!g->synced) return FLAME_HIP_ERR_STATE;  // the caller supplied the edge list itself
  if (g->E > 0 && !edges) return FLAME_HIP_ERR_ARG;
  if (g->sync_on_device) {
    if (hipSetDevice(g->device) != hipSuccess ||
        memcpy_sync(g->stream, edges, g->in_edges, sizeof(int2) * (size_t)g->E, hipMemcpyDeviceToHost) != hipSuccess)
      return FLAME_HIP_ERR_HIP;
    return 0;
  }
  std::memcpy(edges, g->sync.edges.data(), sizeof(int32_t) * 2 * (size_t)g->E);
  return 0;
}

// New data terms on an unchanged topology: resets the solver state (x = x0 or z, w = 0,
// x_bar = x, q = 0) without rebuilding the host plan or touching the graph arrays.
int flame_hip_sync(flame_hip_graph* g);  // (defined below)

int flame_hip_graph_update_data(flame_hip_graph* g, const float* z, const float* wgt, const float* x0) {
  int rc = require_device(g);
  if (rc) return rc;
  const int32_t V = g->V, E = g->E;
  if (V > 0 && (!z || !wgt)) return FLAME_HIP_ERR_ARG;
  if (!all_finite(z, V) || !all_finite(wgt, V) || (x0 && !all_finite(x0, V))) return FLAME_HIP_ERR_NAN;
  HIPCHK(hipSetDevice(g->device));
  HIPCHK(wait_last_solve(g));
  HIPCHK(hipStreamSynchronize(g->stream));
  if (g->persist_unchecked && (rc = flame_hip_sync(g))) return rc;  // (a resident launch that gave up is repeated first)
  if ((rc = dev_alloc(g->caps, &g->in_z, (size_t)V)) || (rc = dev_alloc(g->caps, &g->in_wgt, (size_t)V)) ||
      (rc = dev_alloc(g->caps, &g->in_x0, (size_t)V)))
    return rc;
  HIPCHK(hipMemcpyAsync(g->in_z, z, sizeof(float) * (size_t)V, hipMemcpyHostToDevice, g->stream));
  HIPCHK(hipMemcpyAsync(g->in_wgt, wgt, sizeof(float) * (size_t)V, hipMemcpyHostToDevice, g->stream));
  if (x0) HIPCHK(hipMemcpyAsync(g->in_x0, x0, sizeof(float) * (size_t)V, hipMemcpyHostToDevice, g->stream));
  HIPCHK(launch_init_state(g->stream, V, g->v_i2o_dev, nullptr, g->in_z, g->in_wgt, x0 ? g->in_x0 : nullptr,
                           g->A[g->cur], g->B[g->cur], nullptr, E > 0 ? E : 1, g->q[g->cur], nullptr));
  HIPCHK(hipStreamSynchronize(g->stream));
  g->state_scale = 1.0f;
  g->state_serial++;
  return 0;
}

static int require_device(const flame_hip_graph* g) {
  if (!g) return FLAME_HIP_ERR_ARG;
  if (!g->uploaded) return FLAME_HIP_ERR_STATE;
  if (g->device < 0) return FLAME_HIP_ERR_NODEVICE;
  return 0;
}

int flame_hip_set_state(flame_hip_graph* g, const float* x, const float* w1, const float* w2,
                        const float* xb, const float* w1b, const float* w2b, const float* q) {
  int rc = require_device(g);
  if (rc) return rc;
  const int32_t V = g->V, E = g->E;
  const Plan& P = g->plan;
  HIPCHK(hipSetDevice(g->device));
  HIPCHK(wait_last_solve(g));
  HIPCHK(hipStreamSynchronize(g->stream));
  if (g->persist_unchecked && (rc = flame_hip_sync(g))) return rc;  // (a resident launch that gave up is repeated first)
  if ((rc = ensure_host_perms(g))) return rc;
  g->state_serial++;
  if (x || w1 || w2 || xb || w1b || w2b) {
    std::vector<float4> hA(V), hB(V);
    if (V > 0) {
      HIPCHK(memcpy_sync(g->stream, hA.data(), g->A[g->cur], sizeof(float4) * (size_t)V, hipMemcpyDeviceToHost));
      HIPCHK(memcpy_sync(g->stream, hB.data(), g->B[g->cur], sizeof(float4) * (size_t)V, hipMemcpyDeviceToHost));
    }
    for (int32_t k = 0; k < V; ++k) {
      const int32_t o = P.v_i2o[k];
      if (x) hA[k].x = x[o];
      if (w1) hA[k].y = w1[o];
      if (w2) hA[k].z = w2[o];
      if (xb) hB[k].x = xb[o];
      if (w1b) hB[k].y = w1b[o];
      if (w2b) hB[k].z = w2b[o];
    }
    if ((rc = h2d(g->stream, g->A[g->cur], hA)) || (rc = h2d(g->stream, g->B[g->cur], hB))) return rc;
  }
  if (q && E > 0) {
    std::vector<float4> hq(E);
    for (int32_t k = 0; k < E; ++k) {
      const int32_t o = P.e_i2o[k];
      hq[k] = make_float4(q[3 * o], q[3 * o + 1], q[3 * o + 2], 0.f);
    }
    if ((rc = h2d(g->stream, g->q[g->cur], hq))) return rc;
  }
  return 0;
}

// Enqueue the launches of `num_iters` PD iterations on stream s, starting from buffer `cur`.
// Returns the buffer index holding the result through *cur_out.
// one launch of resident tiles instead of ceil(num_iters / depth) launches?
// one launch of resident tiles instead of ceil(num_iters / depth) launches?
static bool persist_applies(const flame_hip_graph* g, int32_t num_iters) {
  const Plan& P = g->plan;
  if (!g->persist || g->persist_skip_once || g->path != FLAME_HIP_PATH_TILE || g->prof || P.tile_depth <= 0 ||
      num_iters <= P.tile_depth)
    return false;
  const size_t nt = P.tiles.size();
  if (nt < 2 || nt > (size_t)kPersistMaxTiles || (int)nt > g->num_cus) return false;  // (one workgroup per CU is always resident)
  if (!(P.tile_slot12 ? tile_slot12_exists(P.tile_threads, P.tile_ept, P.tile_vpt)
                      : tile_persist_exists(P.tile_threads, P.tile_ept, P.tile_vpt)))
    return false;
  size_t stage = 0;
  for (const TileDesc& D : P.tiles) {
    if (D.n_ext <= 0) return false;  // (an empty tile has nothing to hand over, but its neighbours would not know)
    stage = std::max(stage, persist_stage_bytes(D));
  }
  // (r05: what a tile polls is delivered through an LDS staging area behind its incidence slots)
  if ((size_t)P.tile_lds_bytes + stage + (size_t)kTileLdsReserve > (size_t)g->opt.lds_bytes) return false;
  return true;
}

// Resident tiles need every workgroup of their launch on the chip at once.  Two such launches from two handles of one
// process can each get part of the CUs and starve each other until both time out, so a device has ONE lease: a handle
// takes it for a solve and holds it until that solve's end event has fired; a handle that finds it taken solves by
// ordinary launches.  After a launch gave up for any other reason (a foreign kernel holding CUs, another process) the
// whole process stays off resident tiles for a while: 16 solves, doubling up to 4096 with every further give-up.
struct PersistLease {
  std::mutex m;
  flame_hip_graph* holder = nullptr;
  hipEvent_t holder_done = nullptr;  // (the holder's ev1)
  hipStream_t holder_stream = nullptr;  // the stream its launch went to: a launch queued BEHIND it on the same stream can
                                     // never be co-resident with it (the partition mode's parts of one rank)
  bool recorded = false;             // the holder's end event has been recorded behind its launch (until then: busy)
  bool one_xcd_broken = false;       // a one-XCD launch gave up on this device: the mode is off for the rest of the process
  int backoff = 0;                   // solves (of any handle) to sit out
  int backoff_next = 16;
  int gave_up = 0;                   // give-ups seen on this device (info "persist_gave_up")
};
static PersistLease& persist_lease(int device) {
  static PersistLease leases[64];
  return leases[(unsigned)device & 63];
}
// wait_for_holder: a FAT plan -- halo depth 1 or 2, sized for ONE resident launch -- that finds the lease taken
// would otherwise run as one 256-workgroup launch per 1-2 iterations, several times slower than waiting for the other
// handle's solve to end; so it waits for the holder's end event (host side, bounded by that solve) and takes the lease then
static bool persist_lease_take(flame_hip_graph* g, hipStream_t s, bool wait_for_holder = false) {
  PersistLease& L = persist_lease(g->device);
  std::unique_lock<std::mutex> lk(L.m);
  if (L.backoff > 0) { --L.backoff; return false; }
  if (L.holder && L.holder != g && (!L.recorded || (L.holder_stream != s && hipEventQuery(L.holder_done) != hipSuccess))) {
    if (!wait_for_holder || !L.recorded || !L.holder_done) return false;
    hipEvent_t ev = L.holder_done;
    lk.unlock();
    const hipError_t e = hipEventSynchronize(ev);  // (the event belongs to a live handle of this process: handles outlive their solves)
    lk.lock();
    if (e != hipSuccess || (L.holder && L.holder != g && hipEventQuery(L.holder_done) != hipSuccess)) return false;
  }
  L.holder = g;
  L.holder_done = g->ev1;
  L.holder_stream = s;
  L.recorded = false;  // (flame_hip_solve records ev1 behind the launch, then persist_lease_recorded())
  return true;
}
static void persist_lease_recorded(flame_hip_graph* g) {
  PersistLease& L = persist_lease(g->device);
  std::lock_guard<std::mutex> lk(L.m);
  if (L.holder == g) L.recorded = true;
}
static int persist_gave_up_count(int device) {
  PersistLease& L = persist_lease(device);
  std::lock_guard<std::mutex> lk(L.m);
  return L.gave_up;
}
static int persist_backoff(int device) {
  PersistLease& L = persist_lease(device);
  std::lock_guard<std::mutex> lk(L.m);
  return L.backoff;
}
static bool one_xcd_allowed(int device) {
  if (device < 0) return true;  // (a host-only plan: sized as the MI355X would)
  PersistLease& L = persist_lease(device);
  std::lock_guard<std::mutex> lk(L.m);
  return !L.one_xcd_broken;
}
static void persist_lease_drop(flame_hip_graph* g, bool gave_up) {
  PersistLease& L = persist_lease(g->device);
  std::lock_guard<std::mutex> lk(L.m);
  if (L.holder == g) { L.holder = nullptr; L.holder_done = nullptr; L.holder_stream = nullptr; }
  if (gave_up && g->one_xcd_used) L.one_xcd_broken = true;
  if (gave_up) {
    ++L.gave_up;
    L.backoff = L.backoff_next;
    L.backoff_next = std::min(L.backoff_next * 2, 4096);
  }
}

static int enqueue_iterations(flame_hip_graph* g, const SolveParams& sp, int32_t num_iters, hipStream_t s, int cur,
                              int* cur_out, int* launches);

// After a synchronisation: did a launch of resident tiles give up (a wait timed out: not all of its workgroups were
// on the chip)?  Returns 0 when there was nothing, 1 when the solve was REPEATED by ordinary launches (enqueued, not
// yet waited for: the caller synchronises again and redoes what it had queued behind the solve), an error code when it
// cannot be.  The launch only reads its source buffers, so it can be repeated whenever nothing else wrote the state
// since.  own_marks: state-writing steps the CALLER itself queued behind the solve and will redo (frame_results'
// un-scaling); anything else that wrote the state since (a graph filter, new data terms) cannot be replayed here
static int persist_check(flame_hip_graph* g, int own_marks = 0) {
  const bool force_fail = TEST_HOOK(persist_fail, 0) != 0;  // (hooks library only: the recovery path)
  if (force_fail && g->persist_used && g->persist_err) *g->persist_err = 3;
  g->persist_unchecked = false;  // (every caller has synchronised the solve's stream)
  const int unchecked = g->persist_unchecked_n;
  g->persist_unchecked_n = 0;
  if (!g->persist_err || *g->persist_err == 0) {
    if (g->persist_used && g->last_rounds > 0 && unchecked == 1) {  // the round time the next launch's time-out follows
      float ms = 0.f;
      if (hipEventElapsedTime(&ms, g->ev0, g->ev1) == hipSuccess && ms > 0.f) { g->persist_round_us = ms * 1e3f / (float)g->last_rounds; g->persist_round_V = g->V; }
      else (void)hipGetLastError();
    }
    g->queued.clear(); g->qsnap_valid = false;
    return 0;
  }
  persist_note_give_up(g);
  *g->persist_err = 0;
  persist_lease_drop(g, true);
  // r05: a queue of solves is repeated as a whole from the state copied aside in front of its second member
  if (g->queued.size() > 1) {
    std::vector<flame_hip_graph::QueuedSolve> q;
    q.swap(g->queued);
    const bool ok = g->qsnap_valid && g->state_serial == g->queued_serial + (uint64_t)q.size() + (uint64_t)own_marks;
    g->qsnap_valid = false;
    g->persist_used = false;
    if (!ok) { g->uploaded = false; return FLAME_HIP_ERR_STATE; }
    hipStream_t s = g->last_stream ? g->last_stream : g->stream;
    const int src = g->queued_src;
    HIPCHK(hipMemcpyAsync(g->A[src], g->qsnapA, sizeof(float4) * (size_t)g->V, hipMemcpyDeviceToDevice, s));
    HIPCHK(hipMemcpyAsync(g->B[src], g->qsnapB, sizeof(float4) * (size_t)g->V, hipMemcpyDeviceToDevice, s));
    if (g->E > 0) HIPCHK(hipMemcpyAsync(g->q[src], g->qsnapq, sizeof(float4) * (size_t)g->E, hipMemcpyDeviceToDevice, s));
    HIPCHK(hipEventRecord(g->ev0, s));
    int cur = src, total_launches = 0;
    g->persist_skip_once = true;
    for (const auto& e : q) {
      int cur_out = cur, launches = 0;
      const int rc = enqueue_iterations(g, e.sp, e.iters, s, cur, &cur_out, &launches);
      if (rc) { g->persist_skip_once = false; return rc; }
      cur = cur_out; total_launches += launches;
    }
    g->persist_skip_once = false;
    g->cur = cur;
    g->state_serial++;
    g->last_launches = total_launches;
    HIPCHK(hipEventRecord(g->ev1, s));
    ++g->persist_recovered;
    return 1;
  }
  g->queued.clear(); g->qsnap_valid = false;
  // (several solves queued without a synchronisation in between: an earlier one may have been the one that gave up, and
  // the later ones started from its unfinished result -- nothing to repeat from)
  const bool can = g->persist_used && unchecked <= 1 && g->state_serial == g->solve_serial + (uint64_t)own_marks && g->last_iters > 0;
  g->persist_used = false;
  if (!can) {
    g->uploaded = false;  // the state is that of an unfinished solve
    return FLAME_HIP_ERR_STATE;
  }
  hipStream_t s = g->last_stream ? g->last_stream : g->stream;
  int cur_out = 0, launches = 0;
  HIPCHK(hipEventRecord(g->ev0, s));
  g->persist_skip_once = true;
  int rc = enqueue_iterations(g, g->last_sp, g->last_iters, s, g->last_src, &cur_out, &launches);
  g->persist_skip_once = false;
  if (rc) return rc;
  g->cur = cur_out;
  g->state_serial++;
  g->last_launches = launches;
  HIPCHK(hipEventRecord(g->ev1, s));
  ++g->persist_recovered;
  return 1;
}

static int enqueue_iterations(flame_hip_graph* g, const SolveParams& sp, int32_t num_iters,
                              hipStream_t s, int cur, int* cur_out, int* launches) {
  const Plan& P = g->plan;
  *launches = 0;
  g->persist_used = false;
  if (g->path == FLAME_HIP_PATH_TILE) {
    TileArgs a;
    a.tiles = g->tiles; a.t_vmap = g->t_vmap; a.t_emap = g->t_emap; a.t_eij = g->t_eij;
    a.t_ew = g->t_ew; a.t_srow = g->t_srow; a.p = sp; a.ntiles = (int32_t)P.tiles.size();
    a.prof = g->prof;
    a.slot12 = P.tile_slot12 ? 1 : 0;
    a.fat = P.tile_fat ? 1 : 0;
    if (persist_applies(g, num_iters) && persist_lease_take(g, s, P.tile_fat && P.tile_depth <= 2)) {
      PersistBufs& x = g->xp;
      int rc;
      if (!g->persist_err) {
        HIPCHK(hipHostMalloc(reinterpret_cast<void**>(&g->persist_err), 64, hipHostMallocDefault));
        *g->persist_err = 0;
      }
      if (!x.prof || g->persist_prof_set != g->persist_prof_want) {  // dev aid: which tile (if any) splits its rounds' time
        if ((rc = dev_alloc(g->caps, &x.prof, 16))) return rc;
        int32_t w[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        int want = g->persist_prof_want;
        if (want > 0) { w[0] = 1; w[1] = want - 1; }
        HIPCHK(memcpy_sync(s, x.prof, w, sizeof(w), hipMemcpyHostToDevice));
        g->persist_prof_set = g->persist_prof_want;
      }
      const int rounds = (num_iters + P.tile_depth - 1) / P.tile_depth;
      bool rezero = g->persist_base > (1 << 23);  // (the tags only grow; 2^23: the torn-read debug build compares 24 bits)
      if (rezero) g->persist_base = 0;
      // the tiles' poll lists: in local order for the FIRST solve of a plan (a frame of a stream is solved once and would
      // not earn the sorts back), address-sorted from the second solve on (a resident graph; the lane order, applied then,
      // invalidates them anyway)
      const bool want_sorted = g->solves_since_upload > 0;
      if (!g->poll_valid || (want_sorted && !g->poll_sorted)) {
        size_t nv_loc = 0, ne_loc = 0;
        for (const TileDesc& D : P.tiles) { nv_loc += (size_t)D.n_ext; ne_loc += (size_t)D.e_loc; }
        // (+ 1: a tile without local edges still reads the list's entry at its own offset -- the last tile's is one past the end)
        if ((rc = dev_alloc(g->caps, &x.poll_v, nv_loc + 1)) || (rc = dev_alloc(g->caps, &x.poll_e, ne_loc + 1)) ||
            (rc = dev_alloc(g->caps, &x.poll_ne, P.tiles.size())))
          return rc;
        // (r05: the sorted lists' kernel also marks what anybody polls; the rest of a tile's own entries is not handed over.
        // A plan's first solve -- a frame of a stream -- stores everything and pays neither the marks nor their memsets)
        x.need_valid = false;
        if (want_sorted && g->need_marks && P.tile_fat) {  // (the FAT kernel variants read the marks)
          if ((rc = dev_alloc(g->caps, &x.need_v, std::max<size_t>((size_t)g->V, 1))) || (rc = dev_alloc(g->caps, &x.need_e, std::max<size_t>((size_t)g->E, 1))))
            return rc;
          HIPCHK(hipMemsetAsync(x.need_v, 0, sizeof(int32_t) * std::max<size_t>((size_t)g->V, 1), s));
          HIPCHK(hipMemsetAsync(x.need_e, 0, sizeof(int32_t) * std::max<size_t>((size_t)g->E, 1), s));
          x.need_valid = true;
        }
        HIPCHK(launch_poll_lists(s, (int32_t)P.tiles.size(), g->tiles, g->t_vmap, g->t_emap, g->t_eij, x.poll_v, x.poll_e, x.poll_ne, want_sorted,
                                 x.need_valid ? x.need_v : nullptr, x.need_valid ? x.need_e : nullptr));
        g->poll_sorted = want_sorted;
        x.stage_bytes = 0;
        for (const TileDesc& D : P.tiles) x.stage_bytes = std::max(x.stage_bytes, persist_stage_bytes(D));
        g->poll_valid = true;
      }
      for (int b = 0; b < 2; ++b) {  // (a new or recycled buffer is zeroed: tag 0 is never a round's)
        float4** pp[3] = {&x.hA[b], &x.hB[b], &x.hq[b]};
        const size_t nn[3] = {(size_t)g->V, (size_t)g->V, (size_t)std::max(g->E, 1)};
        for (int k = 0; k < 3; ++k) {
          bool fresh = false;
          if ((rc = dev_alloc_uncached(g->device, pp[k], &g->xp_cap[3 * b + k], nn[k], &fresh))) return rc;
          if (fresh || rezero) HIPCHK(hipMemsetAsync(*pp[k], 0, g->xp_cap[3 * b + k], s));
        }
      }
      PersistBufs xl = x;  // (what this launch gets)
      g->one_xcd_used = false;
      if (g->one_xcd_opt && P.tiles.size() <= (size_t)kOneXcdTiles && g->num_cus >= 256 && one_xcd_allowed(g->device)) {
        for (int b = 0; b < 2; ++b) {
          float4** pp[3] = {&g->cA[b], &g->cB[b], &g->cq[b]};
          const size_t nn[3] = {(size_t)g->V, (size_t)g->V, (size_t)std::max(g->E, 1)};
          for (int k = 0; k < 3; ++k) {
            if ((rc = dev_alloc(g->caps, pp[k], nn[k]))) return rc;
            const size_t cap_now = g->caps[(void*)pp[k]];
            if (g->c_zeroed[3 * b + k] != (void*)*pp[k] || g->c_zeroed_cap[3 * b + k] != cap_now || rezero) {  // (a new or re-grown
              HIPCHK(hipMemsetAsync(*pp[k], 0, cap_now, s));                                                   // allocation: garbage could pass for a tag)
              g->c_zeroed[3 * b + k] = (void*)*pp[k];
              g->c_zeroed_cap[3 * b + k] = cap_now;
            }
          }
          xl.hA[b] = g->cA[b]; xl.hB[b] = g->cB[b]; xl.hq[b] = g->cq[b];
        }
        xl.one_xcd = 1;
        g->one_xcd_used = true;
      }
      a.A_src = g->A[cur]; a.B_src = g->B[cur]; a.q_src = g->q[cur];
      a.A_dst = g->A[cur ^ 1]; a.B_dst = g->B[cur ^ 1]; a.q_dst = g->q[cur ^ 1];
      a.iters = num_iters;
      {
        // (the lists in local order: a pass is long enough, 0; fat tiles with a 1- or 2-iteration round: the hand-off is the
        // larger part of the round and the first pass should not wait -- 200 k 3.02 / 3.05 / 3.13 / 3.15 / 3.23 us per
        // iteration at 0 .. 4, 160 k and 100 k (depth 3) flat: profiles/r05_fat_poll_delay.txt)
        const bool short_fat_round = g->V > 256 * 196 && P.tile_depth <= 2;
        x.poll_delay = g->poll_delay_opt >= 0 ? g->poll_delay_opt : ((g->poll_sorted && !short_fat_round) ? kPollDelayDefault : 0);
        // A poll waits at most max(0.5 ms, 8 x the handle's last measured round) -- r04's flat 4 ms was 5.5 headline
        // solves; 4 ms while nothing has been measured.  Option "persist_timeout_us" overrides.
        const float us = g->persist_timeout_opt > 0 ? (float)g->persist_timeout_opt : (g->persist_round_us > 0.f ? std::max(500.f, 8.f * g->persist_round_us) : 4000.f);
        g->persist_timeout_us = (int32_t)std::min(us, 1.0e6f);
        x.timeout_ticks = g->persist_timeout_us * 100;
        const int stall_us = TEST_HOOK(persist_stall_us, 0);  // (hooks library: a REAL late tile, tests/test_gpu_persist.py)
        if (stall_us > 0 && tile_stall_hook_build()) x.poll_delay = (x.poll_delay & 0xff) | (stall_us << 8);  // (microseconds, bits 8.. of poll_delay; debug kernels only)
      }
      g->last_rounds = rounds;
      // (one XCD: a pass through the L2 is short and cheap -- no pause in front of the first one, profiles/r06_one_xcd_ab.txt)
      xl.poll_delay = (xl.one_xcd && g->poll_delay_opt < 0) ? 0 : x.poll_delay;
      xl.timeout_ticks = x.timeout_ticks;
      HIPCHK(launch_tile_persist(s, P.tile_threads, P.tile_ept, P.tile_vpt, (size_t)P.tile_lds_bytes, a, xl, g->persist_err,
                                 g->persist_base));
      g->give_up_base = g->persist_base;
      g->persist_base += rounds - 1;
      g->persist_used = true;
      g->persist_unchecked = true;
      ++g->persist_unchecked_n;
      ++g->persist_launches;
      g->last_src = cur;
      *launches = 1;
      *cur_out = cur ^ 1;  // (written once, by the last round; the source buffers are only read)
      return 0;
    }
    const int per = P.tile_depth > 0 ? P.tile_depth : num_iters;
    for (int32_t done = 0; done < num_iters;) {
      const int32_t n = std::min<int32_t>(per, num_iters - done);
      a.A_src = g->A[cur]; a.B_src = g->B[cur]; a.q_src = g->q[cur];
      a.A_dst = g->A[cur ^ 1]; a.B_dst = g->B[cur ^ 1]; a.q_dst = g->q[cur ^ 1];
      a.iters = n;
      HIPCHK(launch_tile(s, P.tile_threads, P.tile_ept, P.tile_vpt, (size_t)P.tile_lds_bytes, a));
      cur ^= 1;
      done += n;
      ++*launches;
    }
  } else {
    for (int32_t it = 0; it < num_iters; ++it) {
      HIPCHK(launch_dual(s, g->E, g->eij, g->ew, g->B[cur], g->q[cur], sp.sigma));
      HIPCHK(launch_primal(s, g->V, g->grow, g->ginc, g->ew, g->q[cur], g->A[cur], g->B[cur], sp));
      *launches += 2;
    }
  }
  *cur_out = cur;
  return 0;
}

int flame_hip_solve(flame_hip_graph* g, const flame_hip_params* p, int32_t num_iters,
                    void* stream) {
  RoctxRange roctx_("flame_hip_solve");
  int rc = require_device(g);
  if (rc) return rc;
  if (!p || num_iters < 0) return FLAME_HIP_ERR_ARG;
  if (!std::isfinite(p->data_factor) || !std::isfinite(p->step_x) || !std::isfinite(p->step_q) ||
      !std::isfinite(p->theta) || std::isnan(p->x_min) || std::isnan(p->x_max))
    return FLAME_HIP_ERR_NAN;
  HIPCHK(hipSetDevice(g->device));
  hipStream_t s = stream ? (hipStream_t)stream : g->stream;
  SolveParams sp;
  sp.lambda = p->data_factor; sp.tau = p->step_x; sp.sigma = p->step_q; sp.theta = p->theta;
  sp.x_min = p->x_min; sp.x_max = p->x_max;
  sp.tl = p->step_x * p->data_factor;
  // The plan is being solved a second time, i.e. it is reused (a resident graph, the subdomain
  // solver, a bench): only now the lanes of every 64-edge block are re-assigned against LDS bank
  // conflicts (plan.h lane_order = 1) -- a frame stream that solves each graph once never pays.
  HIPCHK(order_after_state(g, s));
  if (g->path == FLAME_HIP_PATH_TILE && g->opt.lane_order == 1 && !g->lanes_applied &&
      g->solves_since_upload > 0 && num_iters > 0 && g->V > 0) {
    int e_max = 0;
    for (const TileDesc& D : g->plan.tiles) e_max = std::max(e_max, D.e_loc);
    if (g->timed) HIPCHK(hipStreamWaitEvent(s, g->ev1, 0));  // (the first solve may have run on another stream)
    HIPCHK(launch_assign_lanes(s, (int32_t)g->plan.tiles.size(), e_max, g->tiles, g->t_eij, g->t_ew, g->t_emap, g->plan.tile_slot12));
    g->lanes_applied = true;
    g->poll_valid = false;  // (edges moved inside their 64-blocks: a poll record names a position)
  }
  // r05: a solve queued behind an unchecked resident one -- before it overwrites that one's source, the source is set aside
  if (g->persist_unchecked_n == 0) { g->queued.clear(); g->qsnap_valid = false; }
  if (g->queued.size() == 1 && !g->qsnap_valid && g->state_serial == g->queued_serial + 1 && num_iters > 0) {
    int rc2;
    if ((rc2 = dev_alloc(g->caps, &g->qsnapA, (size_t)g->V)) || (rc2 = dev_alloc(g->caps, &g->qsnapB, (size_t)g->V)) ||
        (rc2 = dev_alloc(g->caps, &g->qsnapq, (size_t)std::max(g->E, 1))))
      return rc2;
    const int src = g->queued_src;
    HIPCHK(hipMemcpyAsync(g->qsnapA, g->A[src], sizeof(float4) * (size_t)g->V, hipMemcpyDeviceToDevice, s));
    HIPCHK(hipMemcpyAsync(g->qsnapB, g->B[src], sizeof(float4) * (size_t)g->V, hipMemcpyDeviceToDevice, s));
    if (g->E > 0) HIPCHK(hipMemcpyAsync(g->qsnapq, g->q[src], sizeof(float4) * (size_t)g->E, hipMemcpyDeviceToDevice, s));
    g->qsnap_valid = true;
  }
  const int cur_before = g->cur;
  const uint64_t serial_before = g->state_serial;
  HIPCHK(hipEventRecord(g->ev0, s));
  g->last_sp = sp; g->last_iters = num_iters; g->last_stream = s;
  // (a hipGraph replay or a solve of 0 iterations does not pass through enqueue_iterations(); without this a
  // stale `persist_used` made persist_check() repeat the SHORT solve from buffers the replay had already overwritten)
  g->persist_used = false;
  int launches = 0, cur_out = g->cur;
  if (num_iters > 0 && g->V > 0) {
    if (g->use_graph && g->solves_since_upload > 0 && !persist_applies(g, num_iters)) {  // a frame stream that re-uploads before
      // every solve never pays capture + instantiate; the captured launches replay on any stream
      // (the subdomain solver of the multi-GPU path passes its own)
      GraphExecEntry* hit = nullptr;
      for (auto& e : g->execs)
        if (e.iters == num_iters && e.cur == g->cur && std::memcmp(&e.p, &sp, sizeof(sp)) == 0) hit = &e;
      if (!hit) {
        hipGraph_t graph = nullptr;
        HIPCHK(hipStreamBeginCapture(s, hipStreamCaptureModeRelaxed));
        rc = enqueue_iterations(g, sp, num_iters, s, g->cur, &cur_out, &launches);
        hipError_t ee = hipStreamEndCapture(s, &graph);
        if (rc) { if (graph) (void)hipGraphDestroy(graph); return rc; }
        HIPCHK(ee);
        hipGraphExec_t exec = nullptr;
        ee = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
        (void)hipGraphDestroy(graph);
        HIPCHK(ee);
        if (g->execs.size() >= 8) {
          (void)hipGraphExecDestroy(g->execs.front().exec);
          g->execs.erase(g->execs.begin());
        }
        g->execs.push_back({num_iters, g->cur, sp, exec, launches});
        hit = &g->execs.back();
      }
      HIPCHK(hipGraphLaunch(hit->exec, s));
      launches = hit->launches;
      // parity of the result buffer: tile path flips once per launch, global path never
      cur_out = (g->path == FLAME_HIP_PATH_TILE) ? (g->cur ^ (launches & 1)) : g->cur;
    } else {
      rc = enqueue_iterations(g, sp, num_iters, s, g->cur, &cur_out, &launches);
      if (rc) return rc;
    }
  }
  g->cur = cur_out;
  g->state_serial++;
  g->solve_serial = g->state_serial;
  if (num_iters > 0 && g->V > 0) {  // (the log of what a give-up would have to repeat)
    if (g->queued.empty()) {
      if (g->persist_used) { g->queued.push_back({sp, num_iters}); g->queued_src = cur_before; g->queued_serial = serial_before; }
    } else if (g->qsnap_valid) {
      g->queued.push_back({sp, num_iters});
    } else {
      g->queued.push_back({sp, num_iters});  // (no copy was taken -- something else wrote the state in between: the check refuses)
    }
  }
  g->last_launches = launches;
  g->solves_since_upload++;
  HIPCHK(hipEventRecord(g->ev1, s));
  if (g->persist_used) persist_lease_recorded(g);
  g->timed = true;
  // the maps of the NEXT frame's builder: enqueued behind this frame's iterations -- and, when those run as ONE launch of
  // resident tiles, ordered behind it too (they then overlap the results stage instead of fighting the tiles for CUs)
  HIPCHK(g->planner.flush_grid(g->persist_used ? g->ev1 : nullptr));
  return 0;
}

int flame_hip_sync(flame_hip_graph* g) {
  int rc = require_device(g);
  if (rc) return rc;
  HIPCHK(hipSetDevice(g->device));
  if (g->timed) HIPCHK(hipEventSynchronize(g->ev1));
  HIPCHK(hipStreamSynchronize(g->stream));
  rc = persist_check(g);
  if (rc == 1) {  // the solve was repeated by launches: wait for those
    HIPCHK(hipEventSynchronize(g->ev1));
    HIPCHK(hipStreamSynchronize(g->stream));
    rc = 0;
  }
  return rc;
}

int flame_hip_last_solve_ms(flame_hip_graph* g, float* ms, int32_t* launches) {
  int rc = require_device(g);
  if (rc) return rc;
  if (!g->timed) return FLAME_HIP_ERR_STATE;
  HIPCHK(hipSetDevice(g->device));
  HIPCHK(hipEventSynchronize(g->ev1));
  float t = 0.f;
  HIPCHK(hipEventElapsedTime(&t, g->ev0, g->ev1));
  if (ms) *ms = t;
  if (launches) *launches = g->last_launches;
  return 0;
}

int flame_hip_costs(flame_hip_graph* g, const flame_hip_params* p, double* smooth, double* data) {
  return flame_hip_costs_masked(g, p, nullptr, nullptr, smooth, data);
}

int flame_hip_costs_masked(flame_hip_graph* g, const flame_hip_params* p, const uint8_t* vmask,
                           const uint8_t* emask, double* smooth, double* data) {
  RoctxRange roctx_("flame_hip_costs");
  int rc = require_device(g);
  if (rc) return rc;
  if (!p) return FLAME_HIP_ERR_ARG;
  if ((rc = flame_hip_sync(g))) return rc;
  const int nb = costs_num_blocks(g->V, g->E);
  uint8_t *dvm = nullptr, *dem = nullptr;
  if (vmask || emask) {  // masks arrive in the caller's order: permuted to the internal one here
    if ((rc = ensure_host_perms(g))) return rc;
    const Plan& P = g->plan;
    if ((rc = dev_alloc(g->caps, &g->cost_mask, (size_t)g->V + (size_t)g->E + 64))) return rc;
    std::vector<uint8_t> h((size_t)g->V + (size_t)g->E, 1);
    if (vmask) for (int32_t k = 0; k < g->V; ++k) h[k] = vmask[P.v_i2o[k]] ? 1 : 0;
    if (emask) for (int32_t k = 0; k < g->E; ++k) h[(size_t)g->V + k] = emask[P.e_i2o[k]] ? 1 : 0;
    if (!h.empty()) HIPCHK(memcpy_sync(g->stream, g->cost_mask, h.data(), h.size(), hipMemcpyHostToDevice));
    if (vmask) dvm = g->cost_mask;
    if (emask) dem = g->cost_mask + g->V;
  }
  HIPCHK(launch_costs(g->stream, g->V, g->E, g->eij, g->ew, g->A[g->cur], g->B[g->cur],
                      p->data_factor, g->partials, dem, dvm));
  std::vector<double> h(2 * (size_t)nb);
  HIPCHK(hipMemcpyAsync(h.data(), g->partials, sizeof(double) * h.size(), hipMemcpyDeviceToHost, g->stream));
  HIPCHK(hipStreamSynchronize(g->stream));
  double s = 0.0, d = 0.0;
  for (int b = 0; b < nb; ++b) { s += h[2 * b]; d += h[2 * b + 1]; }
  if (smooth) *smooth = s;
  if (data) *data = d;
  return 0;
}

static void fill_tri_params(const float Kinv[9], const flame_hip_tri_params* tp, TriParamsDev* d);

int flame_hip_triangles(flame_hip_graph* g, const float Kinv[9], const flame_hip_tri_params* tp,
                        float* vtx_normals, uint8_t* tri_valid, float* tri_normals) {
  RoctxRange roctx_("flame_hip_triangles");
  int rc = require_device(g);
  if (rc) return rc;
  if (!Kinv || !tp) return FLAME_HIP_ERR_ARG;
  if (g->plan.T <= 0 && g->T > 0) return FLAME_HIP_ERR_STATE;  // created with T but uploaded without tris
  if ((rc = flame_hip_sync(g))) return rc;
  const int32_t V = g->V, T = g->plan.T;
  TriParamsDev d;
  fill_tri_params(Kinv, tp, &d);
  HIPCHK(launch_triangles(g->stream, V, T, g->pos, g->A[g->cur], g->tris, g->trow, g->tinc, d,
                          g->tri_normals, g->tri_valid, g->vtx_normals));
  g->raster_serial = 0;  // tri_valid / vtx_normals were rewritten (maybe with other filter parameters)
  HIPCHK(hipStreamSynchronize(g->stream));
  if (vtx_normals && V > 0) {
    HIPCHK(launch_download_rows3(g->stream, V, g->v_o2i_dev, g->vtx_normals, g->dl_v));
    HIPCHK(memcpy_sync(g->stream, vtx_normals, g->dl_v, sizeof(float) * 3 * (size_t)V, hipMemcpyDeviceToHost));
  }
  if (tri_valid && T > 0)
    HIPCHK(memcpy_sync(g->stream, tri_valid, g->tri_valid, (size_t)T, hipMemcpyDeviceToHost));
  if (tri_normals && T > 0) {
    std::vector<float4> h(T);
    HIPCHK(memcpy_sync(g->stream, h.data(), g->tri_normals, sizeof(float4) * (size_t)T, hipMemcpyDeviceToHost));
    for (int32_t t = 0; t < T; ++t) {
      tri_normals[3 * t] = h[t].x; tri_normals[3 * t + 1] = h[t].y; tri_normals[3 * t + 2] = h[t].z;
    }
  }
  return 0;
}

static void fill_tri_params(const float Kinv[9], const flame_hip_tri_params* tp, TriParamsDev* d) {
  d->do_oblique = tp->do_oblique_triangle_filter;
  d->do_edge = tp->do_edge_length_filter;
  d->do_idepth = tp->do_idepth_triangle_filter;
  d->cos_thresh = (float)std::cos((double)tp->oblique_normal_thresh);
  d->diff_factor = tp->oblique_idepth_diff_factor;
  d->diff_abs = tp->oblique_idepth_diff_abs;
  const float max_len = tp->edge_length_thresh * (float)tp->width;
  d->max_len2 = max_len * max_len;
  d->min_idepth = tp->min_triangle_idepth;
  for (int k = 0; k < 9; ++k) d->Kinv[k] = Kinv[k];
}

// Triangle stage + dense raster of the CURRENT solver state into the handle's map buffers
// (owner, idepthmap; depth map / cloud on request), enqueued on the handle's stream; skipped when
// the buffers already hold exactly that raster (same state, parameters, filtered flag) -- the stat
// key `coverage`, the dense-map getters and the debug images all come from one rasterisation.
// fo (optional): the frame's per-vertex / per-triangle outputs, written by the triangle stage's own
// two launches.  The per-block covered-pixel counts land in g->map_cov (raster_num_blocks() words).
// cost_lambda / cost_partials (optional, with fo and cov_dst = the frame's results stage): the cost
// partials of the current state ride in the first launch (launch_frame_stage).
static int ensure_raster(flame_hip_graph* g, const float Kinv[9], const flame_hip_tri_params* tp, int filtered,
                         float min_depth, float max_depth, bool want_dm, bool want_cloud, const FrameOut* fo = nullptr,
                         uint32_t* cov_dst = nullptr, float cost_lambda = 0.f, double* cost_partials = nullptr) {
  int rc;
  const int32_t V = g->V, T = g->plan.T;
  const int64_t npix = (int64_t)tp->width * tp->height;
  if (npix != g->map_pixels) {
    if ((rc = dev_alloc(g->caps, &g->map_owner, (size_t)npix)) || (rc = dev_alloc(g->caps, &g->map_idm, (size_t)npix)) ||
        (rc = dev_alloc(g->caps, &g->map_dm, (size_t)npix)) || (rc = dev_alloc(g->caps, &g->map_cloud, 3 * (size_t)npix)) ||
        (rc = dev_alloc(g->caps, &g->map_cov, (size_t)raster_num_blocks(tp->width, tp->height))))
      return rc;
    g->map_pixels = npix;
    g->raster_serial = 0;
  }
  // (cov_dst: the caller wants the covered-pixel counts in its own buffer => always rasterise)
  const bool cached = !cov_dst && g->raster_serial == g->state_serial && g->raster_filtered == filtered &&
                      std::memcmp(&g->raster_tp, tp, sizeof(*tp)) == 0 && std::memcmp(g->raster_kinv, Kinv, 36) == 0;
  if (cached && !want_dm && !want_cloud && !fo) return 0;
  TriParamsDev d;
  fill_tri_params(Kinv, tp, &d);
  hipStream_t s = g->stream;
  if (!cached && V > 0 && T > 0 && npix > 0) {  // everything is due: three launches instead of five or six
    HIPCHK(launch_frame_stage(s, V, g->E, T, tp->width, tp->height, g->pos, g->A[g->cur], g->B[g->cur], g->eij, g->ew, g->tris,
                              g->trow, g->tinc, d, g->tri_normals, g->tri_valid, g->vtx_normals, fo, cost_lambda,
                              cost_partials, filtered, min_depth, max_depth, g->map_owner, g->map_idm,
                              (want_dm || want_cloud) ? g->map_dm : nullptr, want_cloud ? g->map_cloud : nullptr,
                              cov_dst ? cov_dst : g->map_cov));
    g->raster_serial = g->state_serial;
    g->raster_filtered = filtered;
    g->raster_tp = *tp;
    std::memcpy(g->raster_kinv, Kinv, 36);
    return 0;
  }
  if (cost_partials)
    HIPCHK(launch_costs(s, V, g->E, g->eij, g->ew, g->A[g->cur], g->B[g->cur], cost_lambda, cost_partials));
  HIPCHK(launch_triangles(s, V, T, g->pos, g->A[g->cur], g->tris, g->trow, g->tinc, d, g->tri_normals, g->tri_valid,
                          g->vtx_normals, fo));
  if (cached && !want_dm && !want_cloud) return 0;
  HIPCHK(launch_raster(s, T, tp->width, tp->height, g->pos, g->A[g->cur], g->tris, g->tri_valid, filtered, d, min_depth,
                       max_depth, g->map_owner, g->map_idm, (want_dm || want_cloud) ? g->map_dm : nullptr,
                       want_cloud ? g->map_cloud : nullptr, cov_dst ? cov_dst : g->map_cov));
  g->raster_serial = g->state_serial;
  g->raster_filtered = filtered;
  g->raster_tp = *tp;
  std::memcpy(g->raster_kinv, Kinv, 36);
  return 0;
}

// Everything flame::Flame::update() reads back after the solve, in ONE call with ONE stream
// synchronisation (each of costs / download / triangles / graph_edges alone pays its own).  r03: the
// kernels write their results straight into ONE device arena in the caller's order (cost partials |
// x | vertex normals | triangle validity) that leaves with ONE copy; the edge list (constant since
// the graph sync) is fetched on the staging stream while the iterations still run.
int flame_hip_frame_results(flame_hip_graph* g, const flame_hip_params* p, float scale_back,
                            const float Kinv[9], const flame_hip_tri_params* tp, double* smooth, double* data,
                            float* x, float* vtx_normals, uint8_t* tri_valid, int32_t* edges, float* coverage) {
  RoctxRange roctx_("flame_hip_frame_results");
  const float scale_back_arg = scale_back;
  int rc = require_device(g);
  if (rc) return rc;
  if (((smooth || data) && !p) || ((vtx_normals || tri_valid || coverage) && (!Kinv || !tp)) || !std::isfinite(scale_back))
    return FLAME_HIP_ERR_ARG;
  if (coverage && (tp->width < 1 || tp->height < 1)) return FLAME_HIP_ERR_ARG;
  if (g->plan.T <= 0 && g->T > 0 && (vtx_normals || tri_valid || coverage)) return FLAME_HIP_ERR_STATE;
  if (edges && !g->synced) return FLAME_HIP_ERR_STATE;
  // The un-scaling is applied to the resident state in place, ONCE per upload: a second call on the
  // same frame does not scale again, and cannot report costs any more (they are defined in the
  // solver's units).
  if (g->state_scale != 1.0f) {
    if (smooth || data) return FLAME_HIP_ERR_STATE;
    scale_back = 1.0f;
  }
  HIPCHK(hipSetDevice(g->device));
  hipStream_t s = g->stream;
  if (g->timed) HIPCHK(hipStreamWaitEvent(s, g->ev1, 0));  // a solve on a caller's stream
  const int32_t V = g->V, E = g->E, T = g->plan.T;
  const int nb = costs_num_blocks(V, E);
  const bool tri_stage = (vtx_normals || tri_valid || coverage) || (x && Kinv && tp);
  const int ncb = coverage ? raster_num_blocks(tp->width, tp->height) : 0;
  // ---- arena layout (device and page-locked mirror) ----
  auto al = [](size_t v) { return (v + 63) & ~(size_t)63; };
  const size_t off_part = 0, off_x = al(sizeof(double) * 2 * (size_t)nb);
  const size_t off_n = off_x + al(sizeof(float) * (tri_stage ? 1 : 3) * (size_t)V);
  const size_t off_tv = off_n + al(sizeof(float) * 3 * (size_t)V);
  const size_t off_cov = off_tv + al((size_t)std::max(T, 0));
  const size_t off_edges = off_cov + al(sizeof(uint32_t) * (size_t)ncb);
  const size_t dev_bytes = off_edges, total = off_edges + al(sizeof(int2) * (size_t)E);
  if ((rc = dev_alloc(g->caps, &g->fr_dev, dev_bytes))) return rc;
  HIPCHK(g->pout.reserve(total + 64));
  char* host = g->pout.base;
  // Small frames: the kernels write their outputs straight into the page-locked arena (it is mapped into
  // the device's address space) -- no copy command behind the last kernel, which costs more than these
  // few posted writes.  Larger frames go through the device arena and ONE copy.
  const bool direct = dev_bytes <= kDirectOutBytes;
  char* arena = direct ? host : g->fr_dev;
  const bool dev_edges = edges && E > 0 && g->sync_on_device;
  if (dev_edges) {  // not ordered behind the solve: in_edges has been final since the graph sync
    HIPCHK(hipMemcpyAsync(host + off_edges, g->in_edges, sizeof(int2) * (size_t)E, hipMemcpyDeviceToHost, g->stream_in));
  }
  // costs are taken BEFORE the state goes back to the caller's units; with no scaling pending and the
  // raster due anyway they ride in the triangle stage's launch (ensure_raster)
  const bool costs_ride = (smooth || data) && scale_back == 1.0f && tri_stage && coverage != nullptr;
  if ((smooth || data) && !costs_ride)
    HIPCHK(launch_costs(s, V, E, g->eij, g->ew, g->A[g->cur], g->B[g->cur], p->data_factor,
                        reinterpret_cast<double*>(arena + off_part)));
  if (scale_back != 1.0f) {
    HIPCHK(launch_scale_state(s, V, g->A[g->cur], g->B[g->cur], scale_back));
    g->state_scale *= scale_back;
    HIPCHK(mark_state(g));
  }
  if (tri_stage) {
    FrameOut fo;
    fo.v_i2o = g->v_i2o_dev;
    fo.x = x ? reinterpret_cast<float*>(arena + off_x) : nullptr;
    fo.normals = vtx_normals ? reinterpret_cast<float*>(arena + off_n) : nullptr;
    fo.tri_valid = tri_valid ? reinterpret_cast<uint8_t*>(arena + off_tv) : nullptr;
    if (coverage) {  // + the filtered dense raster (kept for the map getters / debug images)
      if ((rc = ensure_raster(g, Kinv, tp, 1, 0.f, 0.f, false, false, &fo, reinterpret_cast<uint32_t*>(arena + off_cov),
                              costs_ride ? p->data_factor : 0.f,
                              costs_ride ? reinterpret_cast<double*>(arena + off_part) : nullptr)))
        return rc;
    } else {
      TriParamsDev d;
      fill_tri_params(Kinv, tp, &d);
      HIPCHK(launch_triangles(s, V, T, g->pos, g->A[g->cur], g->tris, g->trow, g->tinc, d, g->tri_normals,
                              g->tri_valid, g->vtx_normals, &fo));
      g->raster_serial = 0;
    }
  } else if (x && V > 0) {  // no camera / filter parameters given: plain permuted download (3 planes, x first)
    HIPCHK(launch_download_vertex(s, V, g->v_o2i_dev, g->A[g->cur], reinterpret_cast<float*>(arena + off_x)));
  }
  if (!direct) HIPCHK(hipMemcpyAsync(host, g->fr_dev, dev_bytes, hipMemcpyDeviceToHost, s));
  if (edges && E > 0) {  // the edge list is copied out WHILE the iterations still run on the main stream
    if (dev_edges) {
      HIPCHK(hipStreamSynchronize(g->stream_in));
      std::memcpy(edges, host + off_edges, sizeof(int2) * (size_t)E);
    } else {
      std::memcpy(edges, g->sync.edges.data(), sizeof(int32_t) * 2 * (size_t)E);
    }
  }
  HIPCHK(hipStreamSynchronize(s));
  if ((rc = persist_check(g, scale_back != 1.0f ? 1 : 0)) != 0) {
    if (rc != 1) return rc;
    // the solve was repeated by launches (its state is the initial one again, un-scaled): the whole call once more
    g->state_scale = 1.0f;
    g->raster_serial = 0;
    return flame_hip_frame_results(g, p, scale_back_arg, Kinv, tp, smooth, data, x, vtx_normals, tri_valid, edges, coverage);
  }
  if (smooth || data) {
    const double* h = reinterpret_cast<const double*>(host + off_part);
    double sm = 0.0, da = 0.0;
    for (int b = 0; b < nb; ++b) { sm += h[2 * b]; da += h[2 * b + 1]; }
    if (smooth) *smooth = sm;
    if (data) *data = da;
  }
  if (x && V > 0) std::memcpy(x, host + off_x, sizeof(float) * (size_t)V);
  if (vtx_normals && V > 0) std::memcpy(vtx_normals, host + off_n, sizeof(float) * 3 * (size_t)V);
  if (tri_valid && T > 0) std::memcpy(tri_valid, host + off_tv, (size_t)T);
  if (coverage) {
    const uint32_t* c = reinterpret_cast<const uint32_t*>(host + off_cov);
    uint64_t covered = 0;
    for (int b = 0; b < ncb; ++b) covered += c[b];
    *coverage = (float)covered / (float)((int64_t)tp->width * tp->height);
  }
  return 0;
}

int flame_hip_debug_image(flame_hip_graph* g, int32_t kind, const float Kinv[9], const flame_hip_tri_params* tp,
                          float scene_color_scale, int32_t n_feat, const float* feat_pos, const float* feat_mu,
                          uint8_t* bgr) {
  RoctxRange roctx_("flame_hip_debug_image");
  int rc = require_device(g);
  if (rc) return rc;
  if (kind < 0 || kind > 3 || !Kinv || !tp || tp->width < 1 || tp->height < 1 || !bgr || n_feat < 0 ||
      (kind == FLAME_HIP_IMG_FEATURES && n_feat > 0 && (!feat_pos || !feat_mu)))
    return FLAME_HIP_ERR_ARG;
  if (g->plan.T <= 0 && g->T > 0) return FLAME_HIP_ERR_STATE;
  HIPCHK(hipSetDevice(g->device));
  // (a launch of resident tiles that gave up is noticed -- and repeated -- at a synchronising call: an image must not
  // be drawn from an unfinished solve)
  if (g->persist_unchecked && (rc = flame_hip_sync(g))) return rc;
  hipStream_t s = g->stream;
  if (g->timed) HIPCHK(hipStreamWaitEvent(s, g->ev1, 0));
  if ((rc = ensure_raster(g, Kinv, tp, 1, 0.f, 0.f, false, false))) return rc;
  const int64_t npix = (int64_t)tp->width * tp->height;
  if ((rc = dev_alloc(g->caps, &g->dbg_key, (size_t)npix)) || (rc = dev_alloc(g->caps, &g->dbg_bgr, 3 * (size_t)npix)))
    return rc;
  if (kind != FLAME_HIP_IMG_FEATURES) n_feat = 0;
  HIPCHK(g->pout.reserve(3 * (size_t)npix + 12 * (size_t)n_feat + 256));
  if (n_feat > 0) {  // {u, v, mu} records through the page-locked arena
    if ((rc = dev_alloc(g->caps, &g->dbg_feat, 3 * (size_t)n_feat))) return rc;
    float* st = reinterpret_cast<float*>(g->pout.base);
    for (int32_t f = 0; f < n_feat; ++f) { st[3 * f] = feat_pos[2 * f]; st[3 * f + 1] = feat_pos[2 * f + 1]; st[3 * f + 2] = feat_mu[f]; }
    g->pout.used = 12 * (size_t)n_feat;
    HIPCHK(hipMemcpyAsync(g->dbg_feat, st, 12 * (size_t)n_feat, hipMemcpyHostToDevice, s));
  }
  HIPCHK(launch_debug_image(s, kind, g->plan.T, tp->width, tp->height, g->pos, g->A[g->cur], g->tris, g->tri_valid,
                            g->map_owner, g->map_idm, g->vtx_normals, n_feat, g->dbg_feat, scene_color_scale, g->dbg_key,
                            g->dbg_bgr));
  D2HBatch out(g->pout, s);
  HIPCHK(out.add(bgr, g->dbg_bgr, 3 * (size_t)npix));
  HIPCHK(out.finish());
  return 0;
}

int flame_hip_mesh(flame_hip_graph* g, const float Kinv[9], const flame_hip_tri_params* tp,
                   float* points, int32_t* faces, int32_t* num_faces) {
  int rc = require_device(g);
  if (rc) return rc;
  if (!Kinv || !tp || tp->width < 2 || tp->height < 2) return FLAME_HIP_ERR_ARG;
  if (g->plan.T <= 0 && g->T > 0) return FLAME_HIP_ERR_STATE;
  if ((rc = flame_hip_sync(g))) return rc;
  const int32_t V = g->V, T = g->plan.T;
  const Plan& P = g->plan;
  TriParamsDev d;
  fill_tri_params(Kinv, tp, &d);
  HIPCHK(launch_triangles(g->stream, V, T, g->pos, g->A[g->cur], g->tris, g->trow, g->tinc, d,
                          g->tri_normals, g->tri_valid, g->vtx_normals));
  g->raster_serial = 0;  // (as in flame_hip_triangles)
  HIPCHK(launch_mesh(g->stream, V, g->pos, g->A[g->cur], g->vtx_normals, g->v_i2o_dev, d, tp->width,
                     tp->height, g->mesh_pts));
  HIPCHK(hipStreamSynchronize(g->stream));
  if (points && V > 0)
    HIPCHK(memcpy_sync(g->stream, points, g->mesh_pts, sizeof(float4) * 3 * (size_t)V, hipMemcpyDeviceToHost));
  int32_t nf = 0;
  if (T > 0 && faces && (rc = ensure_host_perms(g))) return rc;
  if (T > 0 && (faces || num_faces)) {
    std::vector<uint8_t> valid(T);
    HIPCHK(memcpy_sync(g->stream, valid.data(), g->tri_valid, (size_t)T, hipMemcpyDeviceToHost));
    for (int32_t t = 0; t < T; ++t)
      if (valid[t]) {
        if (faces) {  // reversed winding, caller's vertex ids (reference src/utils.cc:224-226)
          faces[3 * nf] = P.v_i2o[P.tris[3 * t + 2]];
          faces[3 * nf + 1] = P.v_i2o[P.tris[3 * t + 1]];
          faces[3 * nf + 2] = P.v_i2o[P.tris[3 * t]];
        }
        ++nf;
      }
  }
  if (num_faces) *num_faces = nf;
  return 0;
}

int flame_hip_graph_filter(flame_hip_graph* g, int32_t kind, int32_t passes) {
  int rc = require_device(g);
  if (rc) return rc;
  if ((kind != 0 && kind != 1) || passes < 0) return FLAME_HIP_ERR_ARG;
  HIPCHK(hipSetDevice(g->device));
  if (g->persist_unchecked && passes > 0 && (rc = flame_hip_sync(g))) return rc;  // (the filter rewrites what a repeat would need)
  if ((rc = dev_alloc(g->caps, &g->filter_tmp, (size_t)g->V))) return rc;
  if (g->timed) HIPCHK(hipStreamWaitEvent(g->stream, g->ev1, 0));
  for (int32_t k = 0; k < passes; ++k)
    HIPCHK(launch_graph_filter(g->stream, g->V, kind, g->grow, g->ginc, g->eij, g->A[g->cur],
                               g->B[g->cur], g->filter_tmp));
  if (passes > 0) HIPCHK(mark_state(g));
  return 0;
}

int flame_hip_scale_state(flame_hip_graph* g, float s) {
  int rc = require_device(g);
  if (rc) return rc;
  if (!std::isfinite(s)) return FLAME_HIP_ERR_NAN;
  HIPCHK(hipSetDevice(g->device));
  if (g->persist_unchecked && (rc = flame_hip_sync(g))) return rc;
  if (g->timed) HIPCHK(hipStreamWaitEvent(g->stream, g->ev1, 0));
  HIPCHK(launch_scale_state(g->stream, g->V, g->A[g->cur], g->B[g->cur], s));
  g->state_scale *= s;
  HIPCHK(mark_state(g));
  return 0;
}

int flame_hip_depthmaps(flame_hip_graph* g, const float Kinv[9], const flame_hip_tri_params* tp,
                        int32_t filtered, float min_depth, float max_depth, float* idepthmap,
                        float* depthmap, float* cloud) {
  int rc = require_device(g);
  if (rc) return rc;
  if (!Kinv || !tp || tp->width < 1 || tp->height < 1) return FLAME_HIP_ERR_ARG;
  if (g->plan.T <= 0 && g->T > 0) return FLAME_HIP_ERR_STATE;
  if ((rc = flame_hip_sync(g))) return rc;
  const int64_t npix = (int64_t)tp->width * tp->height;
  if ((rc = ensure_raster(g, Kinv, tp, filtered ? 1 : 0, min_depth, max_depth, depthmap != nullptr, cloud != nullptr)))
    return rc;
  HIPCHK(hipStreamSynchronize(g->stream));
  if (idepthmap) HIPCHK(memcpy_sync(g->stream, idepthmap, g->map_idm, sizeof(float) * (size_t)npix, hipMemcpyDeviceToHost));
  if (depthmap) HIPCHK(memcpy_sync(g->stream, depthmap, g->map_dm, sizeof(float) * (size_t)npix, hipMemcpyDeviceToHost));
  if (cloud) HIPCHK(memcpy_sync(g->stream, cloud, g->map_cloud, sizeof(float) * 3 * (size_t)npix, hipMemcpyDeviceToHost));
  return 0;
}

static int download_impl(flame_hip_graph* g, bool bar, float* a0, float* a1, float* a2, float* q) {
  RoctxRange roctx_("flame_hip_download");
  int rc = require_device(g);
  if (rc) return rc;
  if ((rc = flame_hip_sync(g))) return rc;
  const int32_t V = g->V, E = g->E;
  hipStream_t s = g->stream;
  // results leave in the caller's order: permuted by a kernel into a staging buffer, then copied
  HIPCHK(g->pout.reserve(sizeof(float) * 3 * ((size_t)V + (size_t)E) + 64 * 6));
  D2HBatch out(g->pout, s);
  if ((a0 || a1 || a2) && V > 0) {
    HIPCHK(launch_download_vertex(s, V, g->v_o2i_dev, bar ? g->B[g->cur] : g->A[g->cur], g->dl_v));
    HIPCHK(out.add(a0, g->dl_v, sizeof(float) * (size_t)V));
    HIPCHK(out.add(a1, g->dl_v + V, sizeof(float) * (size_t)V));
    HIPCHK(out.add(a2, g->dl_v + 2 * (size_t)V, sizeof(float) * (size_t)V));
  }
  if (q && E > 0) {
    HIPCHK(launch_download_rows3(s, E, g->e_o2i_dev, g->q[g->cur], g->dl_q));
    HIPCHK(out.add(q, g->dl_q, sizeof(float) * 3 * (size_t)E));
  }
  HIPCHK(out.finish());
  return 0;
}

int flame_hip_download(flame_hip_graph* g, float* x, float* w1, float* w2, float* q) {
  return download_impl(g, false, x, w1, w2, q);
}

int flame_hip_download_bar(flame_hip_graph* g, float* xb, float* w1b, float* w2b) {
  return download_impl(g, true, xb, w1b, w2b, nullptr);
}

static int upload_index_list(hipStream_t s, CapMap& caps, const std::vector<int32_t>& map,
                             int32_t limit, int32_t n, const int32_t* ids, int32_t** dev) {
  std::vector<int32_t> h((size_t)(n > 0 ? n : 0));
  for (int32_t k = 0; k < n; ++k) {
    if (ids[k] < 0 || ids[k] >= limit) return FLAME_HIP_ERR_ARG;
    h[k] = map[ids[k]];
  }
  int rc = dev_alloc(caps, dev, h.size());
  if (rc) return rc;
  return h2d(s, *dev, h);
}

int flame_hip_halo_register(flame_hip_graph* g, int32_t n_send_v, const int32_t* send_v,
                            int32_t n_send_e, const int32_t* send_e, int32_t n_recv_v,
                            const int32_t* recv_v, int32_t n_recv_e, const int32_t* recv_e) {
  int rc = require_device(g);
  if (rc) return rc;
  if (n_send_v < 0 || n_send_e < 0 || n_recv_v < 0 || n_recv_e < 0) return FLAME_HIP_ERR_ARG;
  if ((n_send_v && !send_v) || (n_send_e && !send_e) || (n_recv_v && !recv_v) || (n_recv_e && !recv_e))
    return FLAME_HIP_ERR_ARG;
  HIPCHK(hipSetDevice(g->device));
  HIPCHK(wait_last_solve(g));
  HIPCHK(hipStreamSynchronize(g->stream));
  if ((rc = ensure_host_perms(g))) return rc;
  const Plan& P = g->plan;
  if ((rc = upload_index_list(g->stream, g->caps, P.v_o2i, g->V, n_send_v, send_v, &g->halo_send_v)) ||
      (rc = upload_index_list(g->stream, g->caps, P.e_o2i, g->E, n_send_e, send_e, &g->halo_send_e)) ||
      (rc = upload_index_list(g->stream, g->caps, P.v_o2i, g->V, n_recv_v, recv_v, &g->halo_recv_v)) ||
      (rc = upload_index_list(g->stream, g->caps, P.e_o2i, g->E, n_recv_e, recv_e, &g->halo_recv_e)))
    return rc;
  g->n_send_v = n_send_v; g->n_send_e = n_send_e; g->n_recv_v = n_recv_v; g->n_recv_e = n_recv_e;
  return 0;
}

int flame_hip_halo_bytes(const flame_hip_graph* g, int64_t* send_bytes, int64_t* recv_bytes) {
  if (!g) return FLAME_HIP_ERR_ARG;
  if (send_bytes) *send_bytes = 4 * (6 * (int64_t)g->n_send_v + 3 * (int64_t)g->n_send_e);
  if (recv_bytes) *recv_bytes = 4 * (6 * (int64_t)g->n_recv_v + 3 * (int64_t)g->n_recv_e);
  return 0;
}

int flame_hip_halo_pack(flame_hip_graph* g, void* send_buf_dev, void* stream) {
  int rc = require_device(g);
  if (rc) return rc;
  if (!send_buf_dev && (g->n_send_v + g->n_send_e) > 0) return FLAME_HIP_ERR_ARG;
  HIPCHK(hipSetDevice(g->device));
  hipStream_t s = stream ? (hipStream_t)stream : g->stream;
  HIPCHK(order_after_state(g, s));
  HIPCHK(launch_halo_pack(s, g->n_send_v, g->n_send_e, g->halo_send_v, g->halo_send_e,
                          g->A[g->cur], g->B[g->cur], g->q[g->cur], (float*)send_buf_dev));
  return 0;
}

int flame_hip_halo_unpack(flame_hip_graph* g, const void* recv_buf_dev, void* stream) {
  int rc = require_device(g);
  if (rc) return rc;
  if (!recv_buf_dev && (g->n_recv_v + g->n_recv_e) > 0) return FLAME_HIP_ERR_ARG;
  HIPCHK(hipSetDevice(g->device));
  hipStream_t s = stream ? (hipStream_t)stream : g->stream;
  HIPCHK(order_after_state(g, s));
  HIPCHK(launch_halo_unpack(s, g->n_recv_v, g->n_recv_e, g->halo_recv_v, g->halo_recv_e,
                            (const float*)recv_buf_dev, g->A[g->cur], g->B[g->cur], g->q[g->cur]));
  g->state_serial++;
  return 0;
}

int flame_hip_halo_view_get(flame_hip_graph* g, void* stream, flame_hip_halo_view* out) {
  int rc = require_device(g);
  if (rc) return rc;
  if (!out) return FLAME_HIP_ERR_ARG;
  HIPCHK(hipSetDevice(g->device));
  hipStream_t s = stream ? (hipStream_t)stream : g->stream;
  HIPCHK(order_after_state(g, s));
  for (int b = 0; b < 2; ++b) { out->A[b] = g->A[b]; out->B[b] = g->B[b]; out->q[b] = g->q[b]; }
  out->cur = g->cur;
  out->n_send_v = g->n_send_v; out->n_send_e = g->n_send_e; out->n_recv_v = g->n_recv_v; out->n_recv_e = g->n_recv_e;
  out->send_v = g->halo_send_v; out->send_e = g->halo_send_e; out->recv_v = g->halo_recv_v; out->recv_e = g->halo_recv_e;
  return 0;
}

int flame_hip_halo_written(flame_hip_graph* g) {
  int rc = require_device(g);
  if (rc) return rc;
  g->state_serial++;
  return 0;
}

// ---- state snapshot / rollback: what makes a give-up of resident tiles recoverable where the handle alone cannot repeat
// the solve (partition mode: by the time the host looks, the halo unpack has rewritten the state and peers have received
// records of the unfinished solve).  The partition layer snapshots every part in front of the solves it queues between
// two synchronising calls, and, when ANY rank reports a give-up at the next one, rolls all parts back and repeats those
// solves by launches (csrc/part.cpp).  Device-to-device copies on the caller's stream; a few MB, once per synchronising
// call -- not per exchange period.
int flame_hip_state_snapshot(flame_hip_graph* g, void* stream) {
  int rc = require_device(g);
  if (rc) return rc;
  HIPCHK(hipSetDevice(g->device));
  hipStream_t s = stream ? (hipStream_t)stream : g->stream;
  const size_t V = (size_t)g->V, E = (size_t)std::max(g->E, 1);
  if ((rc = dev_alloc(g->caps, &g->snapA, V)) || (rc = dev_alloc(g->caps, &g->snapB, V)) || (rc = dev_alloc(g->caps, &g->snapq, E)))
    return rc;
  HIPCHK(order_after_state(g, s));
  if (g->timed && g->last_stream && g->last_stream != s) HIPCHK(hipStreamWaitEvent(s, g->ev1, 0));
  HIPCHK(hipMemcpyAsync(g->snapA, g->A[g->cur], sizeof(float4) * V, hipMemcpyDeviceToDevice, s));
  HIPCHK(hipMemcpyAsync(g->snapB, g->B[g->cur], sizeof(float4) * V, hipMemcpyDeviceToDevice, s));
  if (g->E > 0) HIPCHK(hipMemcpyAsync(g->snapq, g->q[g->cur], sizeof(float4) * (size_t)g->E, hipMemcpyDeviceToDevice, s));
  g->snap_valid = true;
  g->snap_V = g->V; g->snap_E = g->E;
  return 0;
}

int flame_hip_state_rollback(flame_hip_graph* g, void* stream) {
  int rc = require_device(g);
  if (rc) return rc;
  if (!g->snap_valid || g->snap_V != g->V || g->snap_E != g->E) return FLAME_HIP_ERR_STATE;
  HIPCHK(hipSetDevice(g->device));
  hipStream_t s = stream ? (hipStream_t)stream : g->stream;
  HIPCHK(order_after_state(g, s));
  if (g->timed && g->last_stream && g->last_stream != s) HIPCHK(hipStreamWaitEvent(s, g->ev1, 0));
  const size_t V = (size_t)g->V;
  HIPCHK(hipMemcpyAsync(g->A[g->cur], g->snapA, sizeof(float4) * V, hipMemcpyDeviceToDevice, s));
  HIPCHK(hipMemcpyAsync(g->B[g->cur], g->snapB, sizeof(float4) * V, hipMemcpyDeviceToDevice, s));
  if (g->E > 0) HIPCHK(hipMemcpyAsync(g->q[g->cur], g->snapq, sizeof(float4) * (size_t)g->E, hipMemcpyDeviceToDevice, s));
  g->state_serial++;
  g->uploaded = true;  // (a give-up the handle could not repeat had marked the state unfinished)
  return 0;
}

// Did a launch of resident tiles of this handle give up since the last look?  For a caller that has synchronised the
// stream(s) the solves ran on and owns the recovery (the partition layer): reads and clears the error word, puts the
// device's lease into its back-off, and does NOT try to repeat anything on this handle alone.
int flame_hip_persist_take_error(flame_hip_graph* g, int32_t* gave_up) {
  int rc = require_device(g);
  if (rc) return rc;
  if (!gave_up) return FLAME_HIP_ERR_ARG;
  const bool force_fail = TEST_HOOK(persist_fail, 0) != 0;  // (hooks library only: the recovery path)
  *gave_up = 0;
  const bool had = g->persist_unchecked;
  g->persist_unchecked = false;
  g->persist_unchecked_n = 0;
  g->queued.clear(); g->qsnap_valid = false;  // (the caller -- the partition mode -- keeps snapshots of its own)
  if (had && g->persist_err && (*g->persist_err != 0 || force_fail)) {
    persist_note_give_up(g);
    *g->persist_err = 0;
    persist_lease_drop(g, true);
    g->persist_used = false;
    *gave_up = 1;
  }
  return 0;
}

// Debug/test hook: copy a named plan array to the caller (host logic tests, no device needed).
// Returns the element count (>= 0) or a negative error; copies min(count, cap) elements.
int64_t flame_hip_debug_plan_array(const flame_hip_graph* g, const char* name, void* buf,
                                   int64_t cap_bytes) {
  if (!g || !name || !g->uploaded) return FLAME_HIP_ERR_ARG;
  const Plan& P = g->plan;
  const std::string k(name);
  const void* src = nullptr;
  int64_t n = 0, esz = 4;
  if (P.on_device && k != "tiles" && k != "profile" && k.compare(0, 5, "sync_") != 0) {
    // device-built plan: the arrays live on the GPU, copied out on request
    int64_t nv = 0, ne = 0, ns = 0;
    for (const TileDesc& D : P.tiles) { nv += D.n_ext; ne += D.e_loc; ns += D.n_upd; }
    const void* dev = nullptr;
    const int64_t V = g->V, E = g->E;
    if (k == "v_o2i") { dev = g->v_o2i_dev; n = V; }
    else if (k == "v_i2o") { dev = g->v_i2o_dev; n = V; }
    else if (k == "e_o2i") { dev = g->e_o2i_dev; n = E; }
    else if (k == "e_i2o") { dev = g->e_i2o_dev; n = E; }
    else if (k == "grow") { dev = g->grow; n = V + 1; }
    else if (k == "ginc") { dev = g->ginc; n = 2 * E; }
    else if (k == "eij") { dev = g->eij; n = E; esz = 8; }
    else if (k == "ew") { dev = g->ew; n = E; esz = 16; }
    else if (k == "tris") { dev = g->tris; n = 3 * (int64_t)P.T; }
    else if (k == "trow") { dev = g->trow; n = P.T > 0 ? V + 1 : 0; }
    else if (k == "tinc") { dev = g->tinc; n = 3 * (int64_t)P.T; }
    else if (k == "t_vmap") { dev = g->t_vmap; n = nv; }
    else if (k == "t_emap") { dev = g->t_emap; n = ne; }
    else if (k == "t_eij") { dev = g->t_eij; n = ne; esz = 8; }
    else if (k == "t_ew") { dev = g->t_ew; n = ne; esz = 16; }
    else if (k == "t_srow") { dev = g->t_srow; n = ns; }
    else return FLAME_HIP_ERR_ARG;
    if (buf && cap_bytes > 0 && n > 0) {
      if (hipSetDevice(g->device) != hipSuccess ||
          memcpy_sync(g->stream, buf, dev, (size_t)std::min<int64_t>(cap_bytes, n * esz), hipMemcpyDeviceToHost) != hipSuccess)
        return FLAME_HIP_ERR_HIP;
    }
    return n;
  }
  if (g->lanes_applied && g->device >= 0 && (k == "t_emap" || k == "t_eij" || k == "t_ew")) {
    // host-built plan whose lane order was applied on the device afterwards: the device holds it
    const void* dev = k == "t_emap" ? (const void*)g->t_emap : k == "t_eij" ? (const void*)g->t_eij : (const void*)g->t_ew;
    n = (int64_t)P.t_emap.size();
    esz = k == "t_emap" ? 4 : k == "t_eij" ? 8 : 16;
    if (buf && cap_bytes > 0 && n > 0) {
      if (hipSetDevice(g->device) != hipSuccess ||
          memcpy_sync(g->stream, buf, dev, (size_t)std::min<int64_t>(cap_bytes, n * esz), hipMemcpyDeviceToHost) != hipSuccess)
        return FLAME_HIP_ERR_HIP;
    }
    return n;
  }
  if (k == "v_o2i") { src = P.v_o2i.data(); n = (int64_t)P.v_o2i.size(); }
  else if (k == "v_i2o") { src = P.v_i2o.data(); n = (int64_t)P.v_i2o.size(); }
  else if (k == "e_o2i") { src = P.e_o2i.data(); n = (int64_t)P.e_o2i.size(); }
  else if (k == "e_i2o") { src = P.e_i2o.data(); n = (int64_t)P.e_i2o.size(); }
  else if (k == "grow") { src = P.grow.data(); n = (int64_t)P.grow.size(); }
  else if (k == "ginc") { src = P.ginc.data(); n = (int64_t)P.ginc.size(); }
  else if (k == "eij") { src = P.eij.data(); n = (int64_t)P.eij.size(); esz = 8; }
  else if (k == "tiles") { src = P.tiles.data(); n = (int64_t)P.tiles.size(); esz = sizeof(TileDesc); }
  else if (k == "t_vmap") { src = P.t_vmap.data(); n = (int64_t)P.t_vmap.size(); }
  else if (k == "t_emap") { src = P.t_emap.data(); n = (int64_t)P.t_emap.size(); }
  else if (k == "t_eij") { src = P.t_eij.data(); n = (int64_t)P.t_eij.size(); esz = 8; }
  else if (k == "t_srow") { src = P.t_srow.data(); n = (int64_t)P.t_srow.size(); }
  else if (k == "ew") { src = P.ew.data(); n = (int64_t)P.ew.size(); esz = 16; }
  else if (k == "t_ew") { src = P.t_ew.data(); n = (int64_t)P.t_ew.size(); esz = 16; }
  else if (k == "tris") { src = P.tris.data(); n = (int64_t)P.tris.size(); }
  else if (k == "trow") { src = P.trow.data(); n = (int64_t)P.trow.size(); }
  else if (k == "tinc") { src = P.tinc.data(); n = (int64_t)P.tinc.size(); }
  else if (k == "sync_edges") { src = g->sync.edges.data(); n = (int64_t)g->sync.edges.size(); }
  else if (k == "sync_alpha") { src = g->sync.alpha.data(); n = (int64_t)g->sync.alpha.size(); }
  else if (k == "sync_beta") {
    const std::vector<float>& b = g->sync.beta.empty() ? g->sync.alpha : g->sync.beta;
    src = b.data(); n = (int64_t)b.size();
  }
  else if (k == "sync_z") { src = g->sync.z.data(); n = (int64_t)g->sync.z.size(); }
  else if (k == "sync_wgt") { src = g->sync.wgt.data(); n = (int64_t)g->sync.wgt.size(); }
  else if (k == "sync_x0") { src = g->sync.x0.data(); n = (int64_t)g->sync.x0.size(); }
  else if (k == "profile") {  // device timeline of the LAST tile launch (set_option profile=1)
    if (!g->prof) return FLAME_HIP_ERR_STATE;
    n = (int64_t)P.tiles.size() * kProfWords; esz = 8;
    if (buf && cap_bytes > 0) {
      if (hipDeviceSynchronize() != hipSuccess) return FLAME_HIP_ERR_HIP;
      if (memcpy_sync(g->stream, buf, g->prof, (size_t)std::min<int64_t>(cap_bytes, n * esz), hipMemcpyDeviceToHost) != hipSuccess)
        return FLAME_HIP_ERR_HIP;
    }
    return n;
  }
  else return FLAME_HIP_ERR_ARG;
  if (buf && cap_bytes > 0 && n > 0) std::memcpy(buf, src, (size_t)std::min<int64_t>(cap_bytes, n * esz));
  return n;
}

}  // extern "C"
