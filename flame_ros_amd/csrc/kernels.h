// flame_ros_amd/csrc/kernels.h -- launch wrappers of the HIP kernels (kernels.hip).
#pragma once
#include <hip/hip_runtime.h>

#include "common.h"

namespace flamehip {

struct TileArgs {
  const TileDesc* tiles;
  const int32_t* t_vmap;
  const int32_t* t_emap;
  const uint2* t_eij;
  const float4* t_ew;
  const uint32_t* t_srow;
  const float4* A_src;
  const float4* B_src;
  const float4* q_src;
  float4* A_dst;
  float4* B_dst;
  float4* q_dst;
  SolveParams p;
  int32_t iters;   // iterations in this launch (<= tile depth unless depth == 0)
  int32_t ntiles;
  unsigned long long* prof;  // debug timeline, kProfWords words per tile, or nullptr
  int32_t slot12 = 0;  // 12-byte incidence slots (fat tiles: Plan::tile_slot12; lds_bytes is then tile_lds_bytes(.., true))
  int32_t fat = 0;     // fat tiles (Plan::tile_fat): the resident launch takes the FAT kernel variants (tile_slot12_exists())
};

// ---- global path: one dual + one primal kernel per PD iteration ----
hipError_t launch_dual(hipStream_t s, int32_t E, const int2* eij, const float4* ew,
                       const float4* B, float4* q, float sigma);
hipError_t launch_primal(hipStream_t s, int32_t V, const int32_t* grow, const int32_t* ginc,
                         const float4* ew, const float4* q, float4* A, float4* B, SolveParams p);

// ---- tile path: `iters` PD iterations per launch on LDS-resident subdomains ----
hipError_t launch_tile(hipStream_t s, int nt, int ept, int vpt, size_t lds_bytes,
                       const TileArgs& a);
bool tile_config_exists(int nt, int ept, int vpt);
bool tile_slot12_exists(int nt, int ept, int vpt);  // ... with 12-byte incidence slots (resident or by launches)
// one-time per configuration: opt in to > 48 KiB of dynamic LDS (not capturable in a hipGraph)
hipError_t prepare_tile(int nt, int ept, int vpt, size_t lds_bytes, bool slot12 = false);
// resident tiles (ONE launch for the whole solve; graphs of 2 .. kPersistMaxTiles tiles, at most one per CU: every
// workgroup of the launch must be on the chip at once).  a.iters = the TOTAL iteration count, rounds of `depth`
// iterations inside; neighbours hand their results over through uncached, round-tagged copies of the state arrays
// (kernels.hip PersistArgs).  The caller owns the buffers: hA / hB / hq [2] (V / V / E float4, same indices as the
// state arrays) in hipDeviceMallocUncached memory, zeroed once; prof: 8 ints of device memory (dev aid, zero = off);
// err_host: a page-locked word.  `base` grows by rounds - 1 per launch (tags base + 1 .. base + rounds - 1 never repeat).
// The result lands in a.A_dst / B_dst / q_dst whatever the number of rounds; the source buffers are only read.
constexpr int kPersistMaxTiles = 256;
struct PersistBufs {
  float4* hA[2] = {nullptr, nullptr};
  float4* hB[2] = {nullptr, nullptr};
  float4* hq[2] = {nullptr, nullptr};
  // r05: the tiles' address-sorted poll lists (launch_poll_lists): slot vmap_off + j / emap_off + j = the tile's j-th halo
  // vertex / halo edge in ascending global id, {global id, local id (| bit 31: needs the primal state too)}
  uint2* poll_v = nullptr;   // sum of n_ext (capacity)
  uint2* poll_e = nullptr;   // sum of e_loc
  int32_t* poll_ne = nullptr;  // per tile: its halo edges
  // r05: which owned entries anybody polls (V / E ints, marked by the SORTED lists' kernel after the caller zeroed them):
  // the rest is not handed over.  need_valid: the marks belong to the current lists
  int32_t* need_v = nullptr;
  int32_t* need_e = nullptr;
  bool need_valid = false;
  size_t stage_bytes = 0;    // LDS behind the incidence slots: (n_upd - n_own) x 16 B of the largest tile
  int32_t* prof = nullptr;
  int32_t poll_delay = 0;  // x 256 clocks between a round's stores and the first poll pass
  int32_t timeout_ticks = 0;  // 10 ns ticks a poll may wait (0 = 4 ms)
  int32_t one_xcd = 0;     // r06 one-XCD mode (kernels.hip PersistArgs::one_xcd; graphs of <= 32 tiles): hA / hB / hq are ordinary memory then
};
bool tile_persist_exists(int nt, int ept, int vpt);
hipError_t launch_poll_lists(hipStream_t s, int32_t ntiles, const TileDesc* tiles, const int32_t* t_vmap, const int32_t* t_emap,
                             const uint2* t_eij, uint2* poll_v, uint2* poll_e, int32_t* poll_ne, bool sorted,
                             int32_t* need_v = nullptr, int32_t* need_e = nullptr);
bool tile_stall_hook_build();  // compiled with FLAME_PERSIST_STALL_HOOK (debug: FLAME_HIP_PERSIST_STALL_US makes tile 0 late)
bool tile_torn_check_build();  // compiled with FLAME_TORN_CHECK (debug: hashed hand-off tags, torn entries counted)
hipError_t launch_tile_persist(hipStream_t s, int nt, int ept, int vpt, size_t lds_bytes, const TileArgs& a, const PersistBufs& x,
                               int32_t* err_host, int32_t base);

// ---- costs: per-block float64 partial sums (partials[2*b] smooth, [2*b+1] data) ----
int costs_num_blocks(int32_t V, int32_t E);
// emask / vmask (optional, INTERNAL order): only the flagged edges / vertices are summed
hipError_t launch_costs(hipStream_t s, int32_t V, int32_t E, const int2* eij, const float4* ew,
                        const float4* A, const float4* B, float lambda, double* partials,
                        const uint8_t* emask = nullptr, const uint8_t* vmask = nullptr);

// ---- halo exchange pack / unpack (multi-GPU subdomains) ----
hipError_t launch_halo_pack(hipStream_t s, int32_t nv, int32_t ne, const int32_t* vidx,
                            const int32_t* eidx, const float4* A, const float4* B, const float4* q,
                            float* out);
hipError_t launch_halo_unpack(hipStream_t s, int32_t nv, int32_t ne, const int32_t* vidx,
                              const int32_t* eidx, const float* in, float4* A, float4* B, float4* q);

// ---- r06, the PEER transport of the partition mode (csrc/part.cpp): halo records written straight into the receiving part's
// inbox -- memory of the same GPU, of another GPU of the node (peer access) or of another process (hipIpc) -- instead of
// pack -> ncclSend / ncclRecv -> unpack.  One push and one pull launch per exchange for ALL the parts of a rank.  A segment =
// the vertex (kind 0, kPeerVRec floats each) or edge (kind 1, kPeerERec floats) records of one message; a message's flag word, in the
// RECEIVER's inbox, takes the exchange's epoch once every record of the exchange has left (release, system scope); the pull
// waits for the flags of its segments (bounded: timeout_ticks x 10 ns, then *err = 1) before it reads.  Inboxes are uncached
// memory, two record buffers by epoch parity: a sender can only be one exchange ahead of a receiver (the relation is symmetric).
constexpr int kPeerVRec = 8, kPeerERec = 4;  // floats per vertex / edge record of the peer transport (whole 16-byte words)
struct HaloSegDev {
  const int32_t* idx;  // local vertex / edge ids of the part, `count` of them
  float* buf[2];       // push: the segment in the receiver's inbox; pull: in this rank's own inbox ([epoch & 1])
  int32_t* flag;       // the message's flag word (receiver's inbox)
  const int32_t* didx; // push, the receiving part is a part of THIS rank: the records go straight into its state arrays at
                       // these local ids (no inbox, no flag, no pull segment: one stream orders the parts of a rank); else null
  int32_t dpart;
  int32_t first;       // thread range [first, first + count) of the launch
  int32_t count;
  int32_t kind;        // 0 vertex records, 1 edge records
  int32_t part;        // local part (index into HaloPartDev)
};
struct HaloPartDev { float4* A[2]; float4* B[2]; float4* q[2]; };
struct HaloXArgs {
  const HaloSegDev* segs;
  const HaloPartDev* parts;
  int32_t nsegs, total;   // total = threads that carry a record
  uint32_t cur_mask;      // bit i: part i's current state buffer
  int32_t epoch;
  int32_t* counter;       // push: blocks done (device word, zero between launches)
  int32_t* err;           // pull: set when a wait timed out
  int32_t timeout_ticks;
};
hipError_t launch_halo_push(hipStream_t s, const HaloXArgs& a);
hipError_t launch_halo_pull(hipStream_t s, const HaloXArgs& a);

// ---- per-triangle stage ----
struct TriParamsDev {
  int32_t do_oblique, do_edge, do_idepth;
  float cos_thresh, diff_factor, diff_abs, max_len2, min_idepth;
  float Kinv[9];
};
// fo (optional): the frame's outputs in the CALLER's vertex order (x V floats, normals 3V floats,
// any may be null) and a copy of tri_valid, written by the same two launches
struct FrameOut {
  const int32_t* v_i2o;
  float* x;
  float* normals;
  uint8_t* tri_valid;
};
hipError_t launch_triangles(hipStream_t s, int32_t V, int32_t T, const float2* pos,
                            const float4* A, const int32_t* tris, const int32_t* trow,
                            const int32_t* tinc, TriParamsDev tp, float4* tri_normals,
                            uint8_t* tri_valid, float4* vtx_normals, const FrameOut* fo = nullptr);

// ---- row a9: graph median (kind 0) / low-pass (kind 1) filter, one Jacobi pass ----
hipError_t launch_graph_filter(hipStream_t s, int32_t V, int32_t kind, const int32_t* grow,
                               const int32_t* ginc, const int2* eij, float4* A, float4* B, float* tmp);

// ---- host boundary with a device-resident plan: state init / results out in the caller's order ----
hipError_t launch_init_state(hipStream_t s, int32_t V, const int32_t* v_i2o, const float2* pos_o, const float* z,
                             const float* wgt, const float* x0, float4* A, float4* B, float2* pos_i, int32_t nq,
                             float4* q0, float4* q1);
// out = 3 planes of V floats {x | w1 | w2} in the caller's vertex order
hipError_t launch_download_vertex(hipStream_t s, int32_t V, const int32_t* v_o2i, const float4* S, float* out);
// out[3 o .. 3 o + 2] = S[o2i[o]].xyz (edge duals, vertex normals)
hipError_t launch_download_rows3(hipStream_t s, int32_t n, const int32_t* o2i, const float4* S, float* out);
hipError_t launch_check_finite(hipStream_t s, int64_t n, const float* p, int32_t* flags);

// ---- row a7 epilogue: x, w, x_bar, w_bar, z *= scale (state back in the caller's units) ----
hipError_t launch_scale_state(hipStream_t s, int32_t V, float4* A, float4* B, float scale);

// ---- row f1: mesh vertices in PointNormalUV layout (3 float4 per vertex, caller's order) ----
hipError_t launch_mesh(hipStream_t s, int32_t V, const float2* pos, const float4* A,
                       const float4* vtx_normals, const int32_t* i2o, TriParamsDev tp, int32_t width,
                       int32_t height, float4* out);

// ---- row f2: dense idepthmap (owner pass + fill pass), depth map, point cloud ----
hipError_t launch_raster(hipStream_t s, int32_t T, int32_t width, int32_t height, const float2* pos,
                         const float4* A, const int32_t* tris, const uint8_t* tri_valid,
                         int32_t filtered, TriParamsDev tp, float min_depth, float max_depth,
                         uint32_t* owner, float* idm, float* dm, float* cloud, uint32_t* covered = nullptr);
// covered (optional): raster_num_blocks() per-block counts of the pixels that are not NaN
int raster_num_blocks(int32_t width, int32_t height);
// One frame's results stage in three launches instead of six: {triangle stage | cost partials (partials
// may be null) | owner map cleared}, {vertex normals + the frame's per-vertex outputs | owner pass},
// fill pass.  Same results as launch_costs + launch_triangles + launch_raster.  Needs V, T, pixels > 0.
hipError_t launch_frame_stage(hipStream_t s, int32_t V, int32_t E, int32_t T, int32_t width, int32_t height,
                              const float2* pos, const float4* A, const float4* B, const int2* eij, const float4* ew,
                              const int32_t* tris, const int32_t* trow, const int32_t* tinc, TriParamsDev tp,
                              float4* tri_normals, uint8_t* tri_valid, float4* vtx_normals, const FrameOut* fo,
                              float lambda, double* partials, int32_t filtered, float min_depth, float max_depth,
                              uint32_t* owner, float* idm, float* dm, float* cloud, uint32_t* covered);

// ---- debug images (BGR8) rendered on the device: kind 0 wireframe, 1 features, 2 normals, 3 idepthmap;
// owner / idm = the FILTERED raster of launch_raster; feat = n_feat x {u, v, mu}; key = W x H scratch ----
hipError_t launch_debug_image(hipStream_t s, int32_t kind, int32_t T, int32_t width, int32_t height, const float2* pos,
                              const float4* A, const int32_t* tris, const uint8_t* tri_valid, const uint32_t* owner,
                              const float* idm, const float4* vtx_normals, int32_t n_feat, const float* feat,
                              float scale, uint32_t* key, uint8_t* bgr);

}  // namespace flamehip
