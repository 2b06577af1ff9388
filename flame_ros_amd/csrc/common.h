// flame_ros_amd/csrc/common.h -- structures shared by the host plan builder and the HIP kernels.
//
// Device data layout (all "internal" order = tile-major: the own vertices / own edges of tile t
// are contiguous, tiles follow a recursive-coordinate-bisection order so neighbours in the image
// are neighbours in memory):
//   vtxA[V] float4 {x, w1, w2, z}        primal state + data term      (ping-pong x2)
//   vtxB[V] float4 {xb, w1b, w2b, wgt}   extrapolated state + weight   (ping-pong x2)
//   q[E]    float4 {q1, q2, q3, 0}       dual state                    (ping-pong x2)
//   eij[E]  int2   {i, j}                internal endpoints, i = source (global path)
//   ew[E]   float4 {alpha, beta, dx, dy} d = pos_i - pos_j
//   grow[V+1], ginc[2E]                  vertex -> incident edges, ascending ORIGINAL edge id,
//                                        bit 31 set when the vertex is the edge's target
// Tile plan (tile path): per tile a TileDesc + slices of the flat arrays t_vmap / t_eij /
// t_ew / t_emap / t_srow (see TileDesc).
#pragma once
#include <stdint.h>

namespace flamehip {

constexpr int kMaxDepth = 16;    // max halo depth (iterations per tile launch)
constexpr int kProfWords = 2 * kMaxDepth + 4;  // debug timeline words per tile
#ifndef FLAME_SLOT_ROUND
#define FLAME_SLOT_ROUND 6
#endif
constexpr int kSlotRound = FLAME_SLOT_ROUND;    // incidence slots summed per round of phase P
constexpr int kDummySlots = 64;  // per-lane trash slots behind the incidence slots (inert writes);
                                 // one more, always +0, follows them (reads past a vertex's degree)

struct SolveParams {
  float lambda, tau, sigma, theta, x_min, x_max;
  float tl;  // tau * lambda, rounded once on the host exactly like the oracle does
};

// One LDS-resident subdomain.  Local vertices are ordered by ring (graph distance from the own
// set): ring 0 = own = internal ids [vstart, vstart+n_own); t_vmap lists all n_ext ids.  Local
// edges are ordered by level = max(ring_i, ring_j); the first e_own of them are owned
// (source vertex is own) = internal edge ids [estart, estart+e_own); t_emap lists all e_loc ids.
struct TileDesc {
  int32_t vstart, n_own, n_ext;
  int32_t estart, e_own, e_loc;
  int32_t n_upd;      // local vertices that are ever updated (ring <= depth-1, or all if depth 0)
  int32_t depth;      // halo depth D (0: isolated tile, any number of iterations per launch)
  int32_t vmap_off;   // into t_vmap: n_ext internal vertex ids (gather list, own first)
  int32_t emap_off;   // into t_emap: e_loc internal edge ids (gather list of q, owned first)
  int32_t erec_off;   // into t_eij / t_ew: e_loc records
  int32_t srow_off;   // into t_srow: n_upd packed {slot of incidence 0 | degree << 16}
  int32_t nslots;     // incidence slots (one row per vertex, odd pitch per 64-vertex group)
  int32_t ring_end[kMaxDepth + 1];   // ring_end[r] = #local vertices with ring <= r
  int32_t level_end[kMaxDepth + 1];  // level_end[l] = #local edges with level <= l
};

// LDS of a tile's workgroup: bar[n_ext] (16 B) + the incidence slots, 16 B each -- or, slot12 (fat tiles), 12 B each as a
// 4-byte and an 8-byte array, the slot count rounded up to 4 (kernels.hip SlotMem).  Resident tiles add their staging area.
constexpr int64_t kTileLdsReserve = 64;  // the tile kernels' static LDS (two words) + alignment: a dynamic request of exactly
                                         // the CU's 160 KiB does not launch (found by the fat-size frame soak, r05)
// fat tiles keep a margin to the CU's LDS, and 16-byte slots at depth 1 -- the last configuration tried before the 12-byte
// layout, i.e. the one that ends up closest to 160 KiB -- a larger one: in frame streams of 185-200 k-vertex frames four
// solves in ~1 900 on such plans (156-160 KiB per tile) took 11-27 ms instead of 0.3 (correct bits, no give-up: some tile
// started that late; not reproduced on a resident graph, nor with 12-byte slots at 146-150 KiB).  Cause not found; avoided.
constexpr int64_t kFatLdsMargin = 4096;
constexpr int64_t kFatLdsMarginDepth1 = 12288;
inline int64_t tile_lds_bytes(int32_t n_ext, int32_t nslots, bool slot12) {
  const int64_t n = (int64_t)nslots + kDummySlots + 1;
  return slot12 ? (int64_t)n_ext * 16 + ((n + 3) & ~(int64_t)3) * 12 : (int64_t)n_ext * 16 + n * 16;
}

// ---- incidence-slot layout of a tile: ONE statement for every plan builder (plan.cpp on the host; plan_dev.hip's fused tile pass,
// pass 2 and the one-launch mini plan on the device) and for what the kernels assume (kernels.hip phase P).  One row per
// UPDATED local vertex; the 64 vertices a wavefront updates together (local ids 64 g .. 64 g + 63) share one pitch = their
// largest degree made ODD, so that a column read -- lane l reads row_l[j] -- touches 64 different 16-byte slots, a multiple of
// an odd number apart: conflict-free for ds_read_b128 (and for the split 12-byte layout's b64 / b32 reads); rows are
// zero-padded up to the pitch (phase P sums the wave's longest row from every row: fmaf(-tau, +0, x) == x).  A group takes 64
// pitches even when it holds fewer vertices (the last group of a tile).  Slot numbers are 16 bits (edge records pack two).
#ifdef __HIPCC__
#define FLAME_HD __host__ __device__
#else
#define FLAME_HD
#endif
FLAME_HD inline int32_t slot_group_pitch(int32_t max_degree_of_group) { return (max_degree_of_group < 1 ? 1 : max_degree_of_group) | 1; }
FLAME_HD inline int32_t slot_group_span(int32_t pitch) { return 64 * pitch; }
FLAME_HD inline bool slot_group_fits(int32_t base, int32_t pitch) { return base + slot_group_span(pitch) + kDummySlots <= 65535; }
FLAME_HD inline int32_t slot_row_start(int32_t group_base, int32_t lv, int32_t pitch) { return group_base + (lv & 63) * pitch; }
FLAME_HD inline uint32_t slot_row_word(int32_t row_start, int32_t degree) { return (uint32_t)row_start | ((uint32_t)degree << 16); }  // t_srow

// Cost model of a tile for the balance of the partition (both plan builders, integer): a launch / a round lasts as long as its
// slowest tile -- local edges + 2 x local vertices (r02).  (r05 tried local edges + 2 x UPDATED vertices, what a resident tile's
// iterate time correlates with best at 50 k: -1 % there, +6...18 % at 20 k / 100 k / 200 k, profiles/r05_cost_model_ab.txt.)
#ifdef __HIPCC__
__host__ __device__
#endif
inline long long tile_cost(const TileDesc& D) { return (long long)D.e_loc + 2 * (long long)D.n_ext; }

// Local edge record halves.
//   t_eij[e] = {li | lj << 16, slot_src | slot_dst << 16}   (local vertex ids / incidence slots)
//   t_ew[e]  = {alpha, beta, dx, dy}

}  // namespace flamehip
