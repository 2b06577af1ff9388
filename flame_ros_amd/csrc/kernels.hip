// flame_ros_amd/csrc/kernels.hip -- hand-written HIP kernels for gfx950 (MI355X, CDNA4).
//
// The hot path is FLaME's NLTGV2-L1 primal-dual iteration on the Delaunay vertex graph (upstream
// optimizers::nltgv2_l1_graph_regularizer::step(), driven from flame::Flame::update(), called at
// reference src/flame_offline_tum.cc:578).  It is a sparse-graph
// stencil at ~0.4 flop/byte: no MFMA, the levers are coalescing, LDS residency and launch count.
//
// Arithmetic contract (bit-exact against oracle/nltgv2_oracle.c): float32, explicit fmaf only
// (this file is compiled with -ffp-contract=off), per-vertex accumulation of -tau K^T q in
// ascending ORIGINAL edge id.  The projection v / max(1,|v|) equals clamp(v,-1,1) bit-for-bit
// for every non-NaN v (v/1 = v; v/|v| = +-1), so it is one v_med3_f32.
#include <algorithm>
#include <type_traits>

#include "kernels.h"

namespace flamehip {
namespace {

__device__ __forceinline__ float proj_unit(float v) {
  return __builtin_amdgcn_fmed3f(v, -1.0f, 1.0f);
}

// dual ascent of one edge (row a2).  bi/bj = {xb, w1b, w2b, *} of source/target, w = {alpha,
// beta, dx, dy}.  Updates q in place.
__device__ __forceinline__ void dual_edge(const float4& bi, const float4& bj, const float4& w,
                                          float sigma, float& q1, float& q2, float& q3) {
  float t = bi.x - bj.x;
  t = fmaf(-bi.y, w.z, t);
  t = fmaf(-bi.z, w.w, t);
  const float K1 = w.x * t;
  const float K2 = w.y * (bi.y - bj.y);
  const float K3 = w.y * (bi.z - bj.z);
  q1 = proj_unit(fmaf(sigma, K1, q1));
  q2 = proj_unit(fmaf(sigma, K2, q2));
  q3 = proj_unit(fmaf(sigma, K3, q3));
}

// L1 prox toward the data term + idepth clamp (row a3, proxL1)
__device__ __forceinline__ float prox_l1(float x, float z, float t, float x_min, float x_max) {
  const float r = x - z;
  float xn = (r > t) ? (x - t) : ((r < -t) ? (x + t) : z);
  return fminf(fmaxf(xn, x_min), x_max);
}

// ------------------------------------------------------------------------------------------
// Global path.  Edge-parallel dual, vertex-parallel CSR primal (+prox +extra-gradient fused).
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_dual(int32_t E, const int2* __restrict__ eij,
                                              const float4* __restrict__ ew,
                                              const float4* __restrict__ B,
                                              float4* __restrict__ q, float sigma) {
  const int32_t e = blockIdx.x * 256 + threadIdx.x;
  if (e >= E) return;
  const int2 ij = eij[e];
  const float4 w = ew[e];
  const float4 bi = B[ij.x], bj = B[ij.y];
  float4 qq = q[e];
  dual_edge(bi, bj, w, sigma, qq.x, qq.y, qq.z);
  q[e] = qq;
}

__global__ __launch_bounds__(256) void k_primal(int32_t V, const int32_t* __restrict__ grow,
                                                const int32_t* __restrict__ ginc,
                                                const float4* __restrict__ ew,
                                                const float4* __restrict__ q,
                                                float4* __restrict__ A, float4* __restrict__ B,
                                                SolveParams p) {
  const int32_t v = blockIdx.x * 256 + threadIdx.x;
  if (v >= V) return;
  const float4 a = A[v];
  const float wgt = B[v].w;
  float x = a.x, w1 = a.y, w2 = a.z;
  const int32_t s0 = grow[v], s1 = grow[v + 1];
  const float ntau = -p.tau;
  for (int32_t s = s0; s < s1; ++s) {
    const int32_t ent = ginc[s];
    const int32_t k = ent & 0x7fffffff;
    const float4 qq = q[k];
    const float4 w = ew[k];
    const float aq = w.x * qq.x, b2 = w.y * qq.y, b3 = w.y * qq.z;
    float cx, c1, c2;
    if (ent >= 0) {  // v is the source
      cx = aq; c1 = fmaf(-w.z, aq, b2); c2 = fmaf(-w.w, aq, b3);
    } else {
      cx = -aq; c1 = -b2; c2 = -b3;
    }
    x = fmaf(ntau, cx, x);
    w1 = fmaf(ntau, c1, w1);
    w2 = fmaf(ntau, c2, w2);
  }
  x = prox_l1(x, a.w, p.tl * wgt, p.x_min, p.x_max);
  const float xb = fmaf(p.theta, x - a.x, x);
  const float w1b = fmaf(p.theta, w1 - a.y, w1);
  const float w2b = fmaf(p.theta, w2 - a.z, w2);
  A[v] = make_float4(x, w1, w2, a.w);
  B[v] = make_float4(xb, w1b, w2b, wgt);
}

// ------------------------------------------------------------------------------------------
// Tile path.  One workgroup = one subdomain (own vertices + depth-D halo) resident in LDS and
// registers for `iters` PD iterations; see common.h TileDesc and DESIGN.md.
//   registers : per-thread EPT edges {ids, slots, alpha, beta, dx, dy, q1..3}, VPT vertices
//   LDS       : bar[n_ext] float4 {xb,w1b,w2b,-}; cs[nslots] float4 per-incidence -K^T q terms
// Phase D (edge threads) gathers bar[] of both endpoints, ascends q, scatters the two endpoint
// contributions into the endpoints' incidence slots; phase P (vertex threads) sums its slots in
// slot order (= ascending original edge id => deterministic and oracle-exact), prox,
// extra-gradient, publishes the new bar[].  Two workgroup barriers per iteration, no atomics.
// ------------------------------------------------------------------------------------------
// Keeps the unused 4th component of an LDS float4 read live, so the compiler emits ds_read_b128
// (4 LDS cycles per wave) instead of ds_read_b96 (8 cycles) when only x,y,z are consumed.  Call
// it AFTER a whole batch of reads has been issued: the empty asm makes the compiler wait for its
// operand, so touching each value right after its own load would serialise the batch.
__device__ __forceinline__ void keep_w(const float4& v) { asm volatile("" ::"v"(v.w)); }

// 12-byte store into a 16-byte LDS slot: ds_write_b96 moves 4 source dwords (address + 3 data)
// instead of ds_write_b128's 5 -- the store's cost is that transfer (MI355X guide, LDS table: 10 vs
// 13 cycles per wave-instruction); the unused 4th word of the slot is never read as data.
typedef float f3v __attribute__((ext_vector_type(3)));
__device__ __forceinline__ void lds_store3(float4* slot, float a, float b, float c) {
  f3v v = {a, b, c};
  *reinterpret_cast<f3v*>(slot) = v;
}

#ifndef FLAME_PERSIST_STALL_HOOK
#define FLAME_PERSIST_STALL_HOOK 0
#endif
// Write-back of a tile's results (the launch-by-launch kernels): write-through (sc0 sc1), so the lines drain
// while slower tiles still compute instead of at the end-of-kernel release (guide, "boundary":
// dirty bytes / 6 TB/s are added to the kernel boundary).
typedef float f4v __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void store_result(float4* p, float a, float b, float c, float d) {
  f4v v = {a, b, c, d};
  // s_nop 1: a store of more than 64 bits followed by a VALU write of its data VGPRs needs two
  // wait states (CDNA3 ISA, data hazards); the compiler pads its own stores, not inline asm --
  // without it the next edge's address arithmetic landed in this store's first data register.
  asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
}

__device__ __forceinline__ int wave_max(int v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = max(v, __shfl_xor(v, off, 64));
  return __builtin_amdgcn_readfirstlane(v);
}

// Packed fp32: the (w1, w2) halves of every update are the same operation on two floats, which
// CDNA3+ issues as ONE v_pk_{add,mul,fma}_f32 (IEEE, the same rounding as the scalar op).  The
// iteration phases are VALU-issue bound (4 cycles per wave64 instruction, 4 waves per SIMD), so
// bar[] and the incidence slots keep the pair first: bar = {w1b, w2b, xb, -}, slot = {c1, c2, cx, -}.
typedef float f2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f2v pk_fma(f2v a, f2v b, f2v c) { return __builtin_elementwise_fma(a, b, c); }

// The tile phases pick one of several straight-line bodies by scalar compares (phase P: "read n slots, then the ordered fma
// chain" for n = 1 .. kSlotRound).  Left alone, SimplifyCFG sinks the bodies' common tails into shared blocks joined by PHIs:
// register shuffles and a full lgkmcnt(0) drain in front of the shared tail, on the critical path of every iteration.  A
// body that ends in an asm statement no other body has (the constant differs) has no common tail to sink; the results pass
// through it, so nothing of the chain can move below it either.  (r06; -mllvm -simplifycfg-sink-common=false does the same
// for the whole file and costs the spilling fat-tile kernel 4 %: profiles/r06_tail_marks_ab.txt)
#ifndef FLAME_TAIL_MARKS
#define FLAME_TAIL_MARKS 1
#endif
template <int ID>
__device__ __forceinline__ void tail_mark(f2v& w, float& x) {
#if FLAME_TAIL_MARKS
  asm volatile("; body %c2" : "+v"(w), "+v"(x) : "n"(ID));
#endif
}

// Incidence slots come in two layouts.  S12 = false (the default): one float4 per slot {c1, c2, cx, -}, a slot is named by its
// LDS address.  S12 = true (r05, FAT tiles: one tile per CU beyond 196 own vertices, where 16 bytes per slot do not fit 160 KiB):
// 12 bytes per slot, split so that every access stays naturally aligned (a 12-byte slot read as ds_read_b96 off its 16-byte
// alignment is replayed at 64 cycles, MI355X guide, LDS) -- the (c1, c2) pairs in one array of 8-byte entries, the cx words
// in another of 4-byte entries that sits at the START of the dynamic LDS: a slot is named by 4 x its index, which IS the
// address of its cx word, and 2 x that + the pair array's offset (one v_lshl_add_u32) is the address of its pair.  Same LDS
// cycles as the 16-byte layout by the guide's table (ds_write_b64 + ds_write_b32 = 6 + 4 vs ds_write_b96 = 10; ds_read_b64 +
// ds_read_b32 = 2 + 2 vs ds_read_b128 = 4), twice the DS instructions; odd row pitches are conflict-free for both reads.
template <bool S12> struct SlotT { typedef float4* ref; };
template <> struct SlotT<true> { typedef uint32_t ref; };
template <bool S12> struct SlotMem { float4* cs; };
template <> struct SlotMem<true> { char* f1; char* f2; };
template <bool S12>
__device__ __forceinline__ void slot_store(const SlotMem<S12>& m, typename SlotT<S12>::ref r, float a, float b, float c) {
  if constexpr (S12) {
    *reinterpret_cast<f2v*>(m.f2 + 2u * r) = f2v{a, b};
    *reinterpret_cast<float*>(m.f1 + r) = c;
  } else {
    lds_store3(r, a, b, c);
  }
}
__device__ __forceinline__ char* sm_f2_end(float4* cs, int n_slot_pad) { return reinterpret_cast<char*>(cs) + 8 * n_slot_pad; }
template <bool S12>
__device__ __forceinline__ float4 slot_load(const SlotMem<S12>& m, typename SlotT<S12>::ref r) {
  if constexpr (S12) {
    const f2v p = *reinterpret_cast<const f2v*>(m.f2 + 2u * r);
    return make_float4(p.x, p.y, *reinterpret_cast<const float*>(m.f1 + r), 0.f);
  } else {
    return *r;
  }
}

// Phase D on K of this thread's edges, the O-th on: every gather is issued before the first use.
// ew = {alpha, beta, dx, dy}; es/ed = the source / target incidence slot.
template <int O, int K, int EPT, bool S12, bool MARK>
__device__ __forceinline__ void tile_phase_d(const float4* bar, const SlotMem<S12>& sm, const uint32_t (&eij)[EPT],
                                             const typename SlotT<S12>::ref (&es)[EPT], const typename SlotT<S12>::ref (&ed)[EPT],
                                             const float4 (&ew)[EPT], float (&q1)[EPT],
                                             f2v (&q23)[EPT], float sigma) {
  float4 bi[K], bj[K];
#pragma unroll
  for (int k = 0; k < K; ++k) {
    bi[k] = bar[eij[O + k] & 0xffffu];
    bj[k] = bar[eij[O + k] >> 16];
  }
#pragma unroll
  for (int k = 0; k < K; ++k) { keep_w(bi[k]); keep_w(bj[k]); }
  const f2v sg = {sigma, sigma};
#pragma unroll
  for (int k = 0; k < K; ++k) {
    const int e = O + k;
    // same expressions as dual_edge(), the (w1, w2) pair packed
    const f2v wi = {bi[k].x, bi[k].y}, wj = {bj[k].x, bj[k].y};
    const f2v be = {ew[e].y, ew[e].y}, nd = {-ew[e].z, -ew[e].w};
    float t = bi[k].z - bj[k].z;
    t = fmaf(-bi[k].x, ew[e].z, t);
    t = fmaf(-bi[k].y, ew[e].w, t);
    const float K1 = ew[e].x * t;
    const f2v K23 = be * (wi - wj);
    q1[e] = proj_unit(fmaf(sigma, K1, q1[e]));
    f2v u = pk_fma(sg, K23, q23[e]);
    u.x = proj_unit(u.x);
    u.y = proj_unit(u.y);
    q23[e] = u;
    const float aq = ew[e].x * q1[e];
    const f2v aq2 = {aq, aq};
    const f2v b23 = be * u;
    const f2v s23 = pk_fma(nd, aq2, b23);  // {fmaf(-dx, aq, b2), fmaf(-dy, aq, b3)}
    // one 12-byte store per endpoint; -(a*b) == (-a)*b bit-for-bit, so the target side is two
    // multiplies with a negated operand instead of three sign flips
    slot_store<S12>(sm, es[e], s23.x, s23.y, aq);
    const f2v n23 = (-be) * u;
    slot_store<S12>(sm, ed[e], n23.x, n23.y, -ew[e].x * q1[e]);
  }
#if FLAME_TAIL_MARKS
  // (the last slot store stays in THIS body, see tail_mark())
  if constexpr (MARK) asm volatile("; phase D body %c0" : : "n"(16 * O + K) : "memory");
#endif
}

// blocks [O, O + n) of this wave's edge blocks, n (wave-uniform) <= K: dispatch to the matching unrolled body
template <int O, int K, int EPT, bool S12, bool MARK>
struct PhaseD {
  static __device__ __forceinline__ void run(int n, const float4* bar, const SlotMem<S12>& sm, const uint32_t (&eij)[EPT],
                                             const typename SlotT<S12>::ref (&es)[EPT], const typename SlotT<S12>::ref (&ed)[EPT],
                                             const float4 (&ew)[EPT], float (&q1)[EPT],
                                             f2v (&q23)[EPT], float sigma) {
    if (n == K) tile_phase_d<O, K, EPT, S12, MARK>(bar, sm, eij, es, ed, ew, q1, q23, sigma);
    else PhaseD<O, K - 1, EPT, S12, MARK>::run(n, bar, sm, eij, es, ed, ew, q1, q23, sigma);
  }
};
template <int O, int EPT, bool S12, bool MARK>
struct PhaseD<O, 0, EPT, S12, MARK> {
  static __device__ __forceinline__ void run(int, const float4*, const SlotMem<S12>&, const uint32_t (&)[EPT],
                                             const typename SlotT<S12>::ref (&)[EPT], const typename SlotT<S12>::ref (&)[EPT],
                                             const float4 (&)[EPT], float (&)[EPT], f2v (&)[EPT],
                                             float) {}
};

constexpr int kPRound = kSlotRound;  // incidence slots read per batch of phase P (12-byte slots)
// ... 16-byte slots: 7 -- one batch covers more of the usual longest rows (8-10) and the 1 024-thread kernels still fit 128
// VGPRs: 1.2 k and 50 k -1 % each against 6; the fat-tile kernels with 12-byte slots lose 4.5 % with 7, 8 spills
// (profiles/r06_slot_round_sweep.txt)
constexpr int kPRound16 = kSlotRound + 1;

// Phase P: K consecutive incidence slots of this lane's row, all reads issued before the first use
// (constant offsets: no address arithmetic), then the dependent fma chain in slot order.
// (12-byte slots: `row` is 4 x the row's first slot index)
template <int K>
__device__ __forceinline__ void sum_slots12(const SlotMem<true>& m, uint32_t row, f2v nt2, float ntau, f2v& w, float& x) {
  const f2v* p2 = reinterpret_cast<const f2v*>(m.f2 + 2u * row);
  const float* p1 = reinterpret_cast<const float*>(m.f1 + row);
  f2v c[K];
  float cx[K];
#pragma unroll
  for (int u = 0; u < K; ++u) c[u] = p2[u];
#pragma unroll
  for (int u = 0; u < K; ++u) cx[u] = p1[u];
#pragma unroll
  for (int u = 0; u < K; ++u) {
    w = pk_fma(nt2, c[u], w);
    x = fmaf(ntau, cx[u], x);
  }
}
// n in [LO, HI] (wave-uniform) slots, picked by a balanced tree of scalar compares (r06: a linear chain of compares used to sit
// in front of the reads, on the critical path of every iteration)
template <int LO, int HI>
struct SlotSel12 {
  static __device__ __forceinline__ void run(int n, const SlotMem<true>& m, uint32_t row, f2v nt2, float ntau, f2v& w, float& x) {
    if constexpr (LO == HI) {
      if constexpr (LO > 0) sum_slots12<LO>(m, row, nt2, ntau, w, x);
    } else {
      constexpr int MID = (LO + HI) / 2;
      if (n <= MID) SlotSel12<LO, MID>::run(n, m, row, nt2, ntau, w, x);
      else SlotSel12<MID + 1, HI>::run(n, m, row, nt2, ntau, w, x);
    }
  }
};

template <int K, bool MARK>
__device__ __forceinline__ void sum_slots(const float4* row, f2v nt2, float ntau, f2v& w, float& x) {
  float4 t[K];
#pragma unroll
  for (int u = 0; u < K; ++u) t[u] = row[u];
#pragma unroll
  for (int u = 0; u < K; ++u) keep_w(t[u]);
#pragma unroll
  for (int u = 0; u < K; ++u) {
    const f2v c = {t[u].x, t[u].y};
    w = pk_fma(nt2, c, w);
    x = fmaf(ntau, t[u].z, x);
  }
  if constexpr (MARK) tail_mark<K>(w, x);
}
template <int LO, int HI, bool MARK>
struct SlotSel {
  static __device__ __forceinline__ void run(int n, const float4* row, f2v nt2, float ntau, f2v& w, float& x) {
    if constexpr (LO == HI) {
      if constexpr (LO > 0) sum_slots<LO, MARK>(row, nt2, ntau, w, x);
    } else {
      constexpr int MID = (LO + HI) / 2;
      if (n <= MID) SlotSel<LO, MID, MARK>::run(n, row, nt2, ntau, w, x);
      else SlotSel<MID + 1, HI, MARK>::run(n, row, nt2, ntau, w, x);
    }
  }
};

// Explicit parameters, hottest first: the first 16 dwords of the kernel arguments are preloaded into
// SGPRs at wave launch (-mllvm -amdgpu-kernarg-preload-count=16, flame_ros_amd/build.py), so the
// tile descriptor's address is known at cycle 0 instead of two scalar round trips later.
//
// PERSIST (k_tile_persist, graphs of 2 .. one-tile-per-CU tiles): the tiles stay RESIDENT for the whole solve, one
// workgroup each, on whatever CU / XCD the dispatcher gives them.  A round (= what a launch is otherwise: `depth`
// iterations) ends with a hand-off between neighbouring tiles instead of a kernel boundary: no reload of the constants
// or the own state, no end-of-kernel release.  The L2s of different XCDs are not coherent and a CU's L1 is never
// refreshed, so what a tile hands over travels through UNCACHED copies of the state arrays (hipDeviceMallocUncached:
// no cache holds them, stores and loads meet in memory).  There are no flags, no drain and no barrier over the tiles:
// every 16-byte entry carries the round it belongs to in its 4th word (z, the data weight and the dual's padding
// are constants a reader already has), an owner stores {value, base + round} and a reader polls the entries of its
// halo until their tags are this round's.  The buffers alternate with the round's parity: an owner can only reach
// round r + 2 after every reader of its round-r entries has published round r + 1 (the halo relation is symmetric),
// so two buffers are enough.  A 16-byte aligned store / load is one request inside one 32-byte sector (observed
// untorn on gfx950: MI355X guide, "R2"; the parity tests compare every bit of ~10^8 hand-offs per solve).  The state
// arrays proper are read once and written once, by the last round, into the OTHER buffer: a launch that gave up
// (a wait is bounded: 4 ms) leaves its source intact and is repeated by ordinary launches.  Same arithmetic in the
// same order as the launches it replaces.  What it relies on: all workgroups of the launch being resident at once
// (the host keeps ntiles <= the CU count and gives one such launch per device the chip at a time).
// History: r03 kept the tiles of <= 32-tile graphs on ONE XCD (L2 hand-offs, a flag barrier; it relied on block b
// running on XCD b % 8); r04's first attempt mixed L2 hand-offs inside an XCD with uncached mirrors across and
// neighbour flags (tools/exp/xpersist_mixed_flags.patch: flag latency 3-5 us per round, slower than launches).
struct PersistArgs {
  float4* hA[2];      // uncached, zeroed once: hand-off copies of vtxA / vtxB / q (same indices), [round & 1]
  float4* hB[2];
  float4* hq[2];
  // r05, the ADDRESS-SORTED poll: the poll is bound by the CU's request rate (~one 64-byte line request per clock), and a
  // wave whose lanes poll the entries of THEIR OWN local vertices / edges (ring-major, level-major) touches 0.63 lines per
  // entry.  So which lane polls what is decoupled from who needs it: slot j of a tile's poll list (vmap_off + j /
  // emap_off + j; k_poll_lists) is its j-th halo entry in ascending GLOBAL id -- owner by owner, in the owner's order: 0.36
  // lines per entry --, {global id, local id (| bit 31: the tile updates the vertex: needs A too)}; what arrives is
  // delivered through LDS (x_bar straight into bar[], the duals through the polled edge's own incidence slot -- dead between
  // rounds --, the primal state through a small staging area behind the slots; picked up by the lanes that hold them after the
  // round's barrier).  Edge records name that slot instead of a local id.
  const uint2* poll_v;
  const uint2* poll_e;
  const int32_t* poll_ne;  // halo edges per tile
  // r05: what NOBODY polls is not handed over.  need_v[v] bit 0: some tile holds v in its halo (polls its x_bar entry), bit 1:
  // some tile updates it (polls the primal entry too); need_e[e] != 0: some tile holds e as a halo edge.  Marked by
  // k_poll_lists<true> (the lists of a plan's second solve on); nullptr = every owned entry is stored (a plan's first solve).
  // A fat tile hands over a third of what it owns (200 k vertices: 4.0 GB of hand-off stores per 500 iterations before).
  const int32_t* need_v;
  const int32_t* need_e;
  int32_t* err_host;  // page-locked: set when a wait timed out
  int32_t base;       // tags of this launch are base + 1 .. base + rounds - 1 (they only grow)
  int32_t* prof;      // device memory (16 words), dev aid: [0] != 0: tile [1] sums where its rounds' time goes into [2..6]; [7] the
                      // longest poll wait (10 ns ticks); [8] torn entries seen (FLAME_TORN_CHECK builds)
  int32_t poll_delay; // units of 256 clocks between a round's stores and its first poll pass
  int32_t one_xcd;    // r06 one-XCD mode: the grid is 8 x ntiles blocks and only every 8th -- the ones the dispatcher puts on XCD 0 --
                      // carries a tile: the hand-off copies are then ORDINARY memory, met in that XCD's L2
  int32_t timeout_ticks;  // 10 ns ticks a poll may wait before the launch gives up (the host: max(0.5 ms, 8 x the handle's
                          // last measured round), 4 ms while nothing has been measured)
  // (test hook of a DEBUG build, -DFLAME_PERSIST_STALL_HOOK=1 + FLAME_HIP_PERSIST_STALL_US: bits 8.. of poll_delay =
  // microseconds tile 0 sleeps behind its first hand-off -- a REAL late tile for the time-out path.  Not in the product
  // build: compiled in it cost the 50 k headline 1.7-3 %, never taken (profiles/r05_persist_guards.txt has its runs))
};

__device__ __forceinline__ float4 load_agent(const float4* p) {  // misses the CU's L1, served by the XCD's L2
  const unsigned long long* q = reinterpret_cast<const unsigned long long*>(p);
  const unsigned long long lo = __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  const unsigned long long hi = __hip_atomic_load(q + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return make_float4(__uint_as_float((uint32_t)lo), __uint_as_float((uint32_t)(lo >> 32)), __uint_as_float((uint32_t)hi),
                     __uint_as_float((uint32_t)(hi >> 32)));
}

// Hand-off tags.  Product build: the 4th word of an entry is the round's tag.  FLAME_TORN_CHECK (a debug build,
// tools/exp/build_variant.sh torn -DFLAME_TORN_CHECK=1; tools/stream_soak.py): the low 24 bits are the tag and the top
// 8 a hash of the entry's three payload words, so a reader that ever sees this round's tag beside payload words of
// another round -- a TORN 16-byte access, which the protocol assumes does not happen (MI355X guide: "observed untorn,
// not an architectural guarantee") -- counts it (PersistArgs::prof[8], info "persist_torn") and polls again: "observed
// untorn" becomes a counted zero instead of an absence of failures.
#ifndef FLAME_TORN_CHECK
#define FLAME_TORN_CHECK 0
#endif
#if FLAME_TORN_CHECK
__device__ __forceinline__ uint32_t payload_hash8(float a, float b, float c) {
  uint32_t h = __float_as_uint(a) * 0x9e3779b1u ^ __float_as_uint(b) * 0x85ebca6bu ^ __float_as_uint(c) * 0xc2b2ae35u;
  h ^= h >> 16;
  h ^= h >> 8;
  return h & 0xffu;
}
#endif
__device__ __forceinline__ float tag_word(int32_t tag, float a, float b, float c) {
#if FLAME_TORN_CHECK
  return __uint_as_float(((uint32_t)tag & 0xffffffu) | (payload_hash8(a, b, c) << 24));
#else
  (void)a; (void)b; (void)c;
  return __int_as_float(tag);
#endif
}
// does the entry carry `target`?  (debug build: ... and do its payload words belong to that tag)
__device__ __forceinline__ bool tag_ok(const float4& v, int32_t target, int32_t* torn_counter) {
#if FLAME_TORN_CHECK
  const uint32_t w = __float_as_uint(v.w);
  if ((w & 0xffffffu) != ((uint32_t)target & 0xffffffu)) return false;
  if ((w >> 24) != payload_hash8(v.x, v.y, v.z)) { atomicAdd(torn_counter, 1); return false; }
  return true;
#else
  (void)torn_counter;
  return __float_as_int(v.w) == target;
#endif
}

// FAT (resident kernels only; the launch-by-launch kernels always carry it): the variant fat tiles get -- the lane-less
// outermost ring's loads and the hand-off filtered by what somebody polls.  Kept out of the kernels of graphs up to one
// regular tile per CU: measured +1.2 % per iteration at 50 k with both compiled in (profiles/r05_fat_variant_ab.txt).
template <int NT, int EPT, int VPT, bool PERSIST, bool S12, bool FAT>
__device__ __forceinline__ void tile_body(TileArgs& a, const PersistArgs& pa) {
  typedef typename SlotT<S12>::ref slot_t;
  // tail_mark(): measured -2.6 / -1.2 / -1.3 % per iteration at 1.2 k / 5 k / 10 k vertices (512 threads), +1 % at 50 k and with
  // three edges per thread more scratch in the fat-tile kernels (1 024 threads): profiles/r06_tail_marks_ab.txt
  constexpr bool MARK = FLAME_TAIL_MARKS && NT <= 512 && EPT <= 2;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  // XCD-aware block -> tile map (speed only): block b runs on XCD b % 8, tiles are numbered in
  // bisection order (neighbours adjacent), so giving XCD k the k-th contiguous eighth of the tiles
  // makes tiles that share halo vertices / edges share one L2.  Bijective for any tile count.
  const int nt_all = a.ntiles, xq = nt_all >> 3, xr = nt_all & 7, xcd = blockIdx.x & 7;
  if (PERSIST && pa.one_xcd && xcd != 0) return;
  const int tile_id = (PERSIST && pa.one_xcd) ? (int)(blockIdx.x >> 3)
                                              : (xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + (int)(blockIdx.x >> 3);
  const TileDesc& D = a.tiles[tile_id];
  const int tid = threadIdx.x;
  // the whole descriptor header up front, before anything with side effects: the compiler then
  // fetches it with scalar loads in one go (a uniform *vector* load per use, each followed by a
  // wait, was three serial round trips in front of the index loads)
  const int n_own = D.n_own, n_ext = D.n_ext, n_upd = D.n_upd;
  const int e_own = D.e_own, e_loc = D.e_loc, depth = D.depth;
  const int vstart = D.vstart, estart = D.estart, nslots = D.nslots;
  const int vmap_off = D.vmap_off, emap_off = D.emap_off, erec_off = D.erec_off, srow_off = D.srow_off;
  // (every header word is "used" here, so none of its loads sinks below the early return: one
  // scalar round trip for the whole header)
  asm volatile("" ::"s"(vstart), "s"(estart), "s"(nslots), "s"(vmap_off), "s"(emap_off), "s"(erec_off),
               "s"(srow_off), "s"(n_own), "s"(n_upd), "s"(e_own), "s"(e_loc), "s"(depth));
  if (n_ext == 0) return;  // empty tile (more tiles than vertices)
  // LDS: bar[n_ext] | cs[nslots + kDummySlots + 1] (| the resident tiles' staging area); 12-byte slots: the cx words of the
  // slots FIRST (a slot's name is that word's offset), then bar[], then the (c1, c2) pairs -- tile_lds_bytes() in common.h
  const int n_slot_all = nslots + kDummySlots + 1;
  const int n_slot_pad = (n_slot_all + 3) & ~3;  // (12-byte slots: keeps bar[] 16-byte and the pairs 8-byte aligned)
  float4* bar = reinterpret_cast<float4*>(S12 ? smem + 4 * n_slot_pad : smem);
  float4* cs = bar + n_ext;  // nslots + kDummySlots incidence slots
  SlotMem<S12> sm;
  if constexpr (S12) { sm.f1 = smem; sm.f2 = reinterpret_cast<char*>(cs); }
  else sm.cs = cs;
  const int lane = tid & 63;
  const int wbase = __builtin_amdgcn_readfirstlane(tid & ~63);
  const uint32_t dummy = (uint32_t)(nslots + lane);  // per-lane trash slot: inert writes/reads

  // active-set cutoffs live in lanes: lane l < 32 holds ring_end[l], lane 32+l holds level_end[l]
  int cut = 0;
  if ((lane & 31) <= kMaxDepth) cut = (lane < 32) ? D.ring_end[lane & 31] : D.level_end[lane & 31];

  // ---- global loads: index lists first, then every dependent gather, nothing waited between ----
  int gi[VPT];
  uint32_t vs[VPT];
#pragma unroll
  for (int k = 0; k < VPT; ++k) {
    const int lv = k * NT + tid;
    gi[k] = a.t_vmap[vmap_off + min(lv, n_ext - 1)];
    vs[k] = a.t_srow[srow_off + min(lv, n_upd - 1)];
  }
  uint2 er[EPT];
  int qi[EPT];
#pragma unroll
  for (int k = 0; k < EPT; ++k) {
    const int lec = min(k * NT + tid, max(e_loc - 1, 0));  // arrays carry one pad element
    er[k] = a.t_eij[erec_off + lec];
    qi[k] = a.t_emap[emap_off + lec];
  }
  // (debug timeline start: taken here so that the index loads above do not wait for its kernel
  // argument, which is not among the preloaded ones)
  // resident tiles: this thread's slots of the tile's address-sorted poll list
  uint2 pvr[VPT], per[EPT];
  int n_hv = 0, n_he = 0;
  if (PERSIST) {
    n_hv = n_ext - n_own;
    n_he = pa.poll_ne[tile_id];
#pragma unroll
    for (int k = 0; k < VPT; ++k) pvr[k] = pa.poll_v[vmap_off + min(k * NT + tid, max(n_hv - 1, 0))];
#pragma unroll
    for (int k = 0; k < EPT; ++k) per[k] = pa.poll_e[emap_off + min(k * NT + tid, max(n_he - 1, 0))];
  }
  int nbits[VPT], nedge[EPT];
#pragma unroll
  for (int k = 0; k < VPT; ++k) nbits[k] = 3;
#pragma unroll
  for (int k = 0; k < EPT; ++k) nedge[k] = 1;
  if constexpr (PERSIST && FAT) {
#pragma unroll
    for (int k = 0; k < VPT; ++k) nbits[k] = (pa.need_v && k * NT + tid < n_own) ? pa.need_v[gi[k]] : 3;
#pragma unroll
    for (int k = 0; k < EPT; ++k) nedge[k] = (pa.need_e && e_loc > 0) ? pa.need_e[qi[k]] : 1;
  }
  const unsigned long long t_start = a.prof ? __builtin_readcyclecounter() : 0ull;
  // Loads return in issue order, so they are issued in the order of first need: x_bar (B) of every
  // local vertex fills bar[] and is all the first workgroup barrier waits for; the edge constants,
  // q and the primal state (A) are still in flight across that barrier and are waited for by the
  // wave that needs them (phase D: ew, q; phase P: A).
  float4 vA[VPT], vB[VPT];
#pragma unroll
  for (int k = 0; k < VPT; ++k) vB[k] = a.B_src[gi[k]];
  float4 ew[EPT];
#pragma unroll
  for (int k = 0; k < EPT; ++k) {
    const int lec = min(k * NT + tid, max(e_loc - 1, 0));
    ew[k] = a.t_ew[erec_off + lec];
  }
  float q1[EPT], q2[EPT], q3[EPT];
#pragma unroll
  for (int k = 0; k < EPT; ++k) {
    const float4 qq = a.q_src[e_loc > 0 ? qi[k] : 0];
    q1[k] = qq.x; q2[k] = qq.y; q3[k] = qq.z;
  }
#pragma unroll
  for (int k = 0; k < VPT; ++k) {
    // the outermost ring is read-only: it needs x_bar (B) but not the primal state (A).  Its lanes
    // all read the tile's first own vertex instead (one cache line for the vector-memory unit to
    // walk, the value is never used); a select on the address, NOT a branch around the load -- a
    // predicated load makes the compiler wait for it inside the branch.
    vA[k] = a.A_src[(k * NT + tid < n_upd) ? gi[k] : vstart];
  }

  // Every incidence slot starts at +0 and the padding of a row (slots past the vertex's degree, up
  // to the group's pitch) is never written: phase P sums a wave-uniform number of slots per row
  // without a per-lane bound, because fmaf(-tau, +0, x) == x bit-for-bit.  The stores go out while
  // the global loads above are in flight.
  if constexpr (S12) {
    for (int i = tid; i < n_slot_all; i += NT) {
      reinterpret_cast<float*>(sm.f1)[i] = 0.f;
      reinterpret_cast<f2v*>(sm.f2)[i] = f2v{0.f, 0.f};
    }
  } else {
    for (int i = tid; i < nslots; i += NT) cs[i] = make_float4(0.f, 0.f, 0.f, 0.f);
  }
  // fat tiles: local vertices beyond one per thread lie in the outermost ring (the configuration holds every UPDATED vertex,
  // pick_cfg()): they have no lane, only their x_bar in bar[] -- loaded here, refreshed by the poll's deliveries
  if constexpr (!PERSIST || FAT) {
    for (int lv = VPT * NT + tid; lv < n_ext; lv += NT) {
      const float4 b = a.B_src[a.t_vmap[vmap_off + lv]];
      bar[lv] = make_float4(b.y, b.z, b.x, 0.f);
    }
  }

  float vx[VPT], vz[VPT], vt[VPT], vwgt[VPT], vxb[VPT];
  f2v vw[VPT], vwb[VPT];
  int wdeg[VPT];
  typename std::conditional<S12, uint32_t, const float4*>::type vrow[VPT];
  const float tl = a.p.tl;
#pragma unroll
  for (int k = 0; k < VPT; ++k) {
    const int lv = k * NT + tid;
    vx[k] = vA[k].x; vw[k].x = vA[k].y; vw[k].y = vA[k].z; vz[k] = vA[k].w;
    vxb[k] = vB[k].x; vwb[k].x = vB[k].y; vwb[k].y = vB[k].z; vwgt[k] = vB[k].w;
    vt[k] = tl * vwgt[k];
    if (lv >= n_upd) vs[k] = 0;  // outermost ring / padding lanes: no row of their own (they sum
                                 // somebody else's slots into a value that is never published)
    if (lv < n_ext) bar[lv] = make_float4(vB[k].y, vB[k].z, vB[k].x, 0.f);
    wdeg[k] = wave_max((int)(vs[k] >> 16));
    if constexpr (S12) vrow[k] = 4u * (vs[k] & 0xffffu);
    else vrow[k] = cs + (vs[k] & 0xffffu);
  }
  bool hand_a[VPT], hand_b[VPT], hand_q[EPT];  // resident tiles: is this lane's own entry polled by anybody?
#pragma unroll
  for (int k = 0; k < VPT; ++k) { hand_a[k] = PERSIST && (nbits[k] & 2); hand_b[k] = PERSIST && (nbits[k] & 1); }
#pragma unroll
  for (int k = 0; k < EPT; ++k) hand_q[k] = PERSIST && nedge[k] != 0;
  uint32_t eij[EPT];
  slot_t es[EPT], ed[EPT];
  f2v q23[EPT];
#pragma unroll
  for (int k = 0; k < EPT; ++k) {
    const bool real = (k * NT + tid) < e_loc;
    eij[k] = real ? er[k].x : 0u;  // padding edges gather local vertex 0 and write trash slots
    const uint32_t ss = er[k].y & 0xffffu, sd = er[k].y >> 16;
    if constexpr (S12) {
      es[k] = 4u * ((real && ss != 0xffffu) ? ss : dummy);
      ed[k] = 4u * ((real && sd != 0xffffu) ? sd : dummy);
    } else {
      es[k] = cs + ((real && ss != 0xffffu) ? ss : dummy);
      ed[k] = cs + ((real && sd != 0xffffu) ? sd : dummy);
    }
    q23[k].x = q2[k]; q23[k].y = q3[k];
  }
  __syncthreads();
  // optional in-kernel timeline (debug): [tile][0]=start, [1]=loaded, [2it]=after phase D of
  // iteration it, [2it+1]=after phase P, [kProfWords-1]=end
  unsigned long long* prof = a.prof ? a.prof + (size_t)tile_id * kProfWords : nullptr;
  if (prof && tid == 0) { prof[0] = t_start; prof[1] = __builtin_readcyclecounter(); }

  const float sigma = a.p.sigma, ntau = -a.p.tau, theta = a.p.theta;
  const f2v nt2 = {ntau, ntau}, th2 = {theta, theta};
  const float x_min = a.p.x_min, x_max = a.p.x_max;
  __shared__ int s_abort, s_wait;
  if (PERSIST && tid == 0) { s_abort = 0; s_wait = 0; }
  const bool pprof = PERSIST && pa.prof[0] != 0 && tile_id == pa.prof[1];  // (dev aid, see the end of the round)
  unsigned long long pround = pprof ? wall_clock64() : 0ull;
  int32_t pacc[4] = {0, 0, 0, 0};
  int32_t wait_max = 0;
  int done = 0, round = 0;
  // r06: what this WAVE does in an iteration depends only on r = min(iterations left in the round, depth): its number of active
  // edge blocks (3 bits per r) and which of its vertex blocks are active (2 bits per r), tabulated once in two SGPR pairs --
  // an iteration used to derive them from two v_readlane, a rounding division and a VALU clamp (16 instructions on the
  // critical path in front of phase D's gathers, tools/exp/iter_prof.py)
  unsigned long long nk_tab = 0ull, pa_tab = 0ull;
  for (int r = 0; r <= depth; ++r) {
    // Active sets are prefixes (vertices by ring, edges by level).  Lanes past the cutoff inside
    // an active block keep computing on stale data: by construction that garbage only reaches
    // vertices/edges that are themselves past the cutoff, and nothing past it is written back.
    const int v_act = __builtin_amdgcn_readlane(cut, r);
    const int e_act = __builtin_amdgcn_readlane(cut, 32 + min(r + 1, depth));
    // wave-uniform: this wave's k-th edge block covers local edges [k NT + wbase, +64)
    const int nk = max(0, min(EPT, (e_act - wbase + NT - 1) / NT));
    int pa = 0;
#pragma unroll
    for (int k = 0; k < VPT; ++k) pa |= (k * NT + wbase < v_act) ? (1 << k) : 0;
    nk_tab |= (unsigned long long)nk << (3 * r);
    pa_tab |= (unsigned long long)pa << (2 * r);
  }
  static_assert(EPT < 8 && VPT <= 2 && 3 * (kMaxDepth + 1) <= 64, "the per-wave activity tables hold 3 / 2 bits per entry");
  for (;;) {  // (one pass unless PERSIST: a round = the iterations of one launch)
  const int iters = PERSIST ? min(depth > 0 ? depth : a.iters, a.iters - done) : a.iters;
  for (int it = 1; it <= iters; ++it) {
    const int ri = min(iters - it, depth);
    const int nk = (int)(nk_tab >> (3 * ri)) & 7;
    const int pact = (int)(pa_tab >> (2 * ri)) & 3;
    // ---- phase D: dual ascent + scatter of the -K^T q terms into incidence slots ----
    PhaseD<0, EPT, EPT, S12, MARK>::run(nk, bar, sm, eij, es, ed, ew, q1, q23, sigma);
    // resident tiles: the duals of a round are final after its last phase D -- their hand-off entries (58 % of what
    // a tile hands over) leave now and travel while phase P still runs
    if (PERSIST && it == iters && done + iters < a.iters) {
      float4* const oq_ = pa.hq[(round + 1) & 1];
      const int32_t tagq = pa.base + round + 1;
#pragma unroll
      for (int k = 0; k < EPT; ++k) {
        const int le = k * NT + tid;
        if (le < e_loc && (uint32_t)(qi[k] - estart) < (uint32_t)e_own && hand_q[k])
          oq_[qi[k]] = make_float4(q1[k], q23[k].x, q23[k].y, tag_word(tagq, q1[k], q23[k].x, q23[k].y));
      }
    }
    __syncthreads();
    if (prof && tid == 0 && it <= kMaxDepth) prof[2 * it] = __builtin_readcyclecounter();
    // ---- phase P: primal descent (slot order = ascending original edge id), prox, extra-grad ----
#pragma unroll
    for (int k = 0; k < VPT; ++k) {
      if (pact & (1 << k)) {  // wave-uniform
        const int lv = k * NT + tid;
        const float xp = vx[k];
        const f2v wp = vw[k];
        float x = xp;
        f2v w = wp;
        // incidence j of this lane is at row[j] (rows have an odd pitch: a column read is
        // conflict-free across lanes); the wave sums the longest row's length from every row
        int j = wdeg[k];  // wave-uniform
        if constexpr (S12) {
          uint32_t row = vrow[k];
          for (; j > kPRound; j -= kPRound, row += 4u * kPRound) sum_slots12<kPRound>(sm, row, nt2, ntau, w, x);
          SlotSel12<0, kPRound>::run(j, sm, row, nt2, ntau, w, x);
        } else {
          // (batches of kPRound: 8 or 12 slots per batch -- one LDS round trip for the usual longest row of a 64-vertex
          // group -- measured equal at 1.2 k ... 10 k vertices, r06 profiles/r06_slot_batch_ab.txt: the segment is bound by
          // the LDS return path, three waves x 9 KB per phase P, not by the round trips)
          const float4* row = vrow[k];
          for (; j > kPRound16; j -= kPRound16, row += kPRound16) sum_slots<kPRound16, MARK>(row, nt2, ntau, w, x);
          SlotSel<0, kPRound16, MARK>::run(j, row, nt2, ntau, w, x);
        }
        x = prox_l1(x, vz[k], vt[k], x_min, x_max);
        vxb[k] = fmaf(theta, x - xp, x);
        vwb[k] = pk_fma(th2, w - wp, w);
        vx[k] = x; vw[k] = w;
        if (lv < n_upd) lds_store3(&bar[lv], vwb[k].x, vwb[k].y, vxb[k]);
        // (resident tiles: the own vertices' hand-off entries leave as soon as the round's last phase P has them, in
        // front of the workgroup barrier)
        if (PERSIST && it == iters && done + iters < a.iters && lv < n_own) {
          const int32_t tagv = pa.base + round + 1;
          if (hand_a[k]) pa.hA[(round + 1) & 1][vstart + lv] = make_float4(x, w.x, w.y, tag_word(tagv, x, w.x, w.y));
          if (hand_b[k]) pa.hB[(round + 1) & 1][vstart + lv] = make_float4(vxb[k], vwb[k].x, vwb[k].y, tag_word(tagv, vxb[k], vwb[k].x, vwb[k].y));
        }
      }
    }
    __syncthreads();
    if (prof && tid == 0 && it <= kMaxDepth) prof[2 * it + 1] = __builtin_readcyclecounter();
  }

  // ---- write back what this tile owns ----
  // (resident tiles, every round but the last: not the state arrays but the uncached hand-off copies, tagged with the round)
  const bool handoff = PERSIST && done + iters < a.iters;
  const int32_t tagf = pa.base + round + 1;
  float4* const oA = handoff ? pa.hA[(round + 1) & 1] : a.A_dst;
  float4* const oB = handoff ? pa.hB[(round + 1) & 1] : a.B_dst;
  float4* const oq = handoff ? pa.hq[(round + 1) & 1] : a.q_dst;
#pragma unroll
  for (int k = 0; k < VPT; ++k) {
    const int lv = k * NT + tid;
    if (lv < n_own) {
      if (PERSIST) {
        if (!handoff) {  // (the hand-off entries left inside the last phase P)
          oA[vstart + lv] = make_float4(vx[k], vw[k].x, vw[k].y, handoff ? tag_word(tagf, vx[k], vw[k].x, vw[k].y) : vz[k]);
          oB[vstart + lv] = make_float4(vxb[k], vwb[k].x, vwb[k].y, handoff ? tag_word(tagf, vxb[k], vwb[k].x, vwb[k].y) : vwgt[k]);
        }
      } else {
        store_result(&a.A_dst[vstart + lv], vx[k], vw[k].x, vw[k].y, vz[k]);
        store_result(&a.B_dst[vstart + lv], vxb[k], vwb[k].x, vwb[k].y, vwgt[k]);
      }
    }
  }
#pragma unroll
  for (int k = 0; k < EPT; ++k) {
    const int le = k * NT + tid;
    // owned <=> the edge's internal id lies in the tile's range (lanes inside a 64-edge block are
    // assigned by the plan's conflict-avoiding lane order, not by internal id)
    if (le < e_loc && (uint32_t)(qi[k] - estart) < (uint32_t)e_own) {
      if (PERSIST) { if (!handoff) oq[qi[k]] = make_float4(q1[k], q23[k].x, q23[k].y, handoff ? tag_word(tagf, q1[k], q23[k].x, q23[k].y) : 0.0f); }
      else store_result(&a.q_dst[qi[k]], q1[k], q23[k].x, q23[k].y, 0.0f);
    }
  }
  if (!PERSIST) break;
  done += iters;
  ++round;
  if (done >= a.iters) break;
  // ---- end of a round: the halo state of the next one = the owners' hand-off entries, polled until they carry this
  // round's tag.  A lane re-issues its loads until all of ITS entries are there; every load of a pass goes out before
  // the first one is looked at; entries a lane does not need point at one address per tile (one request per wave) ----
  const unsigned long long pt0 = pprof ? wall_clock64() : 0ull;
  float4 nb[VPT], na[VPT], nq[EPT];
  {
    const int32_t target = pa.base + round;
    const float4* const hA = pa.hA[round & 1];
    const float4* const hB = pa.hB[round & 1];
    const float4* const hq = pa.hq[round & 1];
    bool needv[VPT], needa[VPT], neede[EPT], want = false;
#pragma unroll
    for (int k = 0; k < VPT; ++k) {
      needv[k] = (k * NT + tid) < n_hv;          // (poll slot j = k NT + tid of this tile's list, not local vertex j)
      needa[k] = needv[k] && (pvr[k].y >> 31);
      want = want || needv[k];
    }
#pragma unroll
    for (int k = 0; k < EPT; ++k) {
      neede[k] = (k * NT + tid) < n_he && per[k].x != 0xffffffffu;  // (an unsorted list marks the owned edges invalid)
      want = want || neede[k];
    }
    // (the neighbours finish their round at about the same time and their stores take ~0.7 us to land: a poll pass
    // issued at once samples memory too early and costs a second round trip)
#if FLAME_PERSIST_STALL_HOOK
    for (int w = 0; w < (pa.poll_delay & 0xff); ++w) __builtin_amdgcn_s_sleep(4);
#else
    for (int w = 0; w < pa.poll_delay; ++w) __builtin_amdgcn_s_sleep(4);
#endif
    // (r05, tried and dropped -- a give-up is cheap now that whole queues of solves are repeated (flame_hip.cpp), and each of
    // these cost the 50 k headline 1-2 % through the code around the poll: a bound on the number of passes beside the clock
    // (a queue the driver switches out has not "waited"; but passes are slow exactly when a wait is long: one ran to 21 ms),
    // discounting gaps of more than 50 us between passes, one 100 us grace period after the first expiry)
    const unsigned long long w0 = wall_clock64();
    bool stale = want;
    for (;;) {
      if (stale) {
        if (VPT == 1 && EPT <= 3) {  // one 16-byte request per lane and array (an atomic load is at most 8 bytes: twice
          const float4* pb = &hB[needv[0] ? (int)pvr[0].x : vstart];  // the requests, and the poll is bound by their number)
          const float4* pv = &hA[needa[0] ? (int)pvr[0].x : vstart];
          const float4* p0 = &hq[neede[0] ? (int)per[0].x : estart];
          const float4* p1 = &hq[neede[EPT > 1 ? 1 : 0] ? (int)per[EPT > 1 ? 1 : 0].x : estart];
          const float4* p2 = &hq[neede[EPT - 1] ? (int)per[EPT - 1].x : estart];
          // a request goes out only when a lane of the wave (still) needs that array (r05: every wave used to issue all five
          // each pass, most of them for the placeholder address -- own vertices have no halo entries, the last lanes no
          // vertices, EPT = 2 no third edge; the poll is bound by the CU's request rate).  Loads, their scalar predicates and
          // the wait are ONE asm block: the compiler does not track vmcnt for loads issued in inline asm, so nothing of its
          // own (a copy, a spill) may land between a load and the wait.
          f4v r0 = {0.f, 0.f, 0.f, 0.f}, r1 = r0, r2 = r0, r3 = r0, r4 = r0;
          // (ballots: wave-uniform by construction, so the compiler keeps them in SGPR pairs as the "s" constraints ask)
          const unsigned long long c0 = __builtin_amdgcn_ballot_w64(needv[0]), c1 = __builtin_amdgcn_ballot_w64(needa[0]);
          const unsigned long long c2 = __builtin_amdgcn_ballot_w64(neede[0]);
          const unsigned long long c3 = EPT > 1 ? __builtin_amdgcn_ballot_w64(neede[EPT > 1 ? 1 : 0]) : 0ull;
          const unsigned long long c4 = EPT > 2 ? __builtin_amdgcn_ballot_w64(neede[EPT - 1]) : 0ull;
          asm volatile(
              "s_cmp_eq_u64 %10, 0\n\ts_cbranch_scc1 .Lpoll_a%=\n\tglobal_load_dwordx4 %0, %5, off sc1\n.Lpoll_a%=:\n\t"
              "s_cmp_eq_u64 %11, 0\n\ts_cbranch_scc1 .Lpoll_b%=\n\tglobal_load_dwordx4 %1, %6, off sc1\n.Lpoll_b%=:\n\t"
              "s_cmp_eq_u64 %12, 0\n\ts_cbranch_scc1 .Lpoll_c%=\n\tglobal_load_dwordx4 %2, %7, off sc1\n.Lpoll_c%=:\n\t"
              "s_cmp_eq_u64 %13, 0\n\ts_cbranch_scc1 .Lpoll_d%=\n\tglobal_load_dwordx4 %3, %8, off sc1\n.Lpoll_d%=:\n\t"
              "s_cmp_eq_u64 %14, 0\n\ts_cbranch_scc1 .Lpoll_e%=\n\tglobal_load_dwordx4 %4, %9, off sc1\n.Lpoll_e%=:\n\t"
              "s_waitcnt vmcnt(0)"
              : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4)
              : "v"(pb), "v"(pv), "v"(p0), "v"(p1), "v"(p2), "s"(c0), "s"(c1), "s"(c2), "s"(c3), "s"(c4)
              : "memory", "scc");
          nb[0] = make_float4(r0.x, r0.y, r0.z, r0.w);
          na[0] = make_float4(r1.x, r1.y, r1.z, r1.w);
          nq[0] = make_float4(r2.x, r2.y, r2.z, r2.w);
          if (EPT > 1) nq[EPT > 1 ? 1 : 0] = make_float4(r3.x, r3.y, r3.z, r3.w);
          if (EPT > 2) nq[EPT - 1] = make_float4(r4.x, r4.y, r4.z, r4.w);
        } else {
#pragma unroll
          for (int k = 0; k < VPT; ++k) {
            nb[k] = load_agent(&hB[needv[k] ? (int)pvr[k].x : vstart]);
            na[k] = load_agent(&hA[needa[k] ? (int)pvr[k].x : vstart]);
          }
#pragma unroll
          for (int k = 0; k < EPT; ++k) nq[k] = load_agent(&hq[neede[k] ? (int)per[k].x : estart]);
        }
        bool ok = true;
#pragma unroll
        for (int k = 0; k < VPT; ++k) {
          ok = ok && (!needv[k] || tag_ok(nb[k], target, pa.prof + 8));
          ok = ok && (!needa[k] || tag_ok(na[k], target, pa.prof + 8));
        }
#pragma unroll
        for (int k = 0; k < EPT; ++k) ok = ok && (!neede[k] || tag_ok(nq[k], target, pa.prof + 8));
        stale = !ok;
        if (pprof) ++pacc[3];  // (dev aid: poll passes of the profiled wave)
      }
      if (!__any(stale)) break;
      if (wall_clock64() - w0 > (unsigned long long)pa.timeout_ticks) {  // give up, never hang (the host repeats the solve by launches)
        s_abort = 1;
        *pa.err_host = 1;
        break;
      }
    }
    wait_max = max(wait_max, __builtin_amdgcn_readfirstlane((int32_t)(wall_clock64() - w0)));  // (scalar: no VGPR)  // (the longest any poll of this wave waited: info "persist_wait_us_max")
  }
  const unsigned long long pt1 = pprof ? wall_clock64() : 0ull;
  // what this lane polled goes where it is needed: x_bar into bar[] (phase D gathers it there); the duals into the polled
  // edge's OWN incidence slot (dead between rounds: the first phase D of the next round rewrites every real slot before a
  // phase P reads one; the zero padding of the rows is never touched); the primal state into a small staging area behind
  // the slots ((n_upd - n_own) x 16 B)
  float4* const stA = S12 ? reinterpret_cast<float4*>(sm_f2_end(cs, n_slot_pad)) : cs + nslots + kDummySlots + 1;
#pragma unroll
  for (int k = 0; k < VPT; ++k) {
    if ((k * NT + tid) < n_hv) {
      const int lv = (int)(pvr[k].y & 0x7fffffffu);
      bar[lv] = make_float4(nb[k].y, nb[k].z, nb[k].x, 0.f);
      if (pvr[k].y >> 31) stA[lv - n_own] = na[k];
    }
  }
#pragma unroll
  for (int k = 0; k < EPT; ++k)
    if ((k * NT + tid) < n_he && per[k].x != 0xffffffffu) {
      if constexpr (S12) slot_store<S12>(sm, 4u * per[k].y, nq[k].x, nq[k].y, nq[k].z);
      else cs[per[k].y] = nq[k];
    }
  __syncthreads();
  if (s_abort) break;
  // ... and the lanes that hold a halo vertex / halo edge pick their state up
#pragma unroll
  for (int k = 0; k < VPT; ++k) {
    const int lv = k * NT + tid;
    if (lv >= n_own && lv < n_upd) {
      const float4 t = stA[lv - n_own];
      vx[k] = t.x; vw[k].x = t.y; vw[k].y = t.z;
    }
  }
#pragma unroll
  for (int k = 0; k < EPT; ++k) {
    const int le = k * NT + tid;
    if (le < e_loc && !((uint32_t)(qi[k] - estart) < (uint32_t)e_own)) {  // halo edges: staged in the source's slot, else the target's
      float4 t;
      if constexpr (S12) t = slot_load<S12>(sm, (es[k] != 4u * dummy) ? es[k] : ed[k]);
      else t = *((es[k] != cs + dummy) ? es[k] : ed[k]);
      q1[k] = t.x; q23[k].x = t.y; q23[k].y = t.z;
    }
  }
#if FLAME_PERSIST_STALL_HOOK
  if ((pa.poll_delay >> 8) != 0 && tile_id == 0 && round == 1) {  // (test hook: tile 0 is late from here on, see PersistArgs)
    const unsigned long long ts0 = wall_clock64();
    while (wall_clock64() - ts0 < 100ull * (unsigned long long)(pa.poll_delay >> 8)) __builtin_amdgcn_s_sleep(64);
  }
#endif
  if (pprof) {  // dev aid: where a round's time goes (10 ns ticks, summed over the rounds of one tile)
    const unsigned long long pt2 = wall_clock64();
    pacc[0] += (int32_t)(pt0 - pround); pacc[1] += (int32_t)(pt1 - pt0); pacc[2] += (int32_t)(pt2 - pt1);
    pround = pt2;
  }
  }  // rounds
  if (PERSIST) {  // the tile's longest poll wait: ONE global atomic per tile (a same-address atomic per wave -- 4 096 of them at
    // the end of a 50 k solve -- cost 46 us, measured: the word takes ~88 of them per microsecond)
    if (lane == 0 && wait_max > 0) atomicMax(&s_wait, wait_max);  // (LDS)
    __syncthreads();
    if (tid == 0 && s_wait > 0) __hip_atomic_fetch_max(&pa.prof[7], s_wait, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  if (PERSIST && pprof && tid == 0) {
    pa.prof[2] = pacc[0]; pa.prof[3] = pacc[1]; pa.prof[4] = pacc[2]; pa.prof[5] = round; pa.prof[6] = pacc[3];
  }
  if (!PERSIST && prof && tid == 0) prof[kProfWords - 1] = __builtin_readcyclecounter();
}

template <int NT, int EPT, int VPT, bool S12 = false>
__global__ __launch_bounds__(NT) void k_tile(const TileDesc* __restrict__ tiles, int32_t ntiles, int32_t iters_arg,
                                             const int32_t* __restrict__ t_vmap,
                                             const uint32_t* __restrict__ t_srow,
                                             const uint2* __restrict__ t_eij,
                                             const int32_t* __restrict__ t_emap,
                                             const float4* __restrict__ t_ew,
                                             const float4* __restrict__ B_src,
                                             const float4* __restrict__ A_src,
                                             const float4* __restrict__ q_src, float4* __restrict__ A_dst,
                                             float4* __restrict__ B_dst, float4* __restrict__ q_dst,
                                             unsigned long long* prof_arg, const SolveParams p_arg) {
  TileArgs a;
  a.tiles = tiles; a.t_vmap = t_vmap; a.t_emap = t_emap; a.t_eij = t_eij; a.t_ew = t_ew; a.t_srow = t_srow;
  a.A_src = A_src; a.B_src = B_src; a.q_src = q_src; a.A_dst = A_dst; a.B_dst = B_dst; a.q_dst = q_dst;
  a.p = p_arg; a.iters = iters_arg; a.ntiles = ntiles; a.prof = prof_arg;
  tile_body<NT, EPT, VPT, false, S12, true>(a, PersistArgs{});
}

template <int NT, int EPT, int VPT, bool S12 = false, bool FAT = false>
__global__ __launch_bounds__(NT) void k_tile_persist(const TileDesc* __restrict__ tiles, int32_t ntiles, int32_t iters_total,
                                                     const int32_t* __restrict__ t_vmap, const uint32_t* __restrict__ t_srow,
                                                     const uint2* __restrict__ t_eij, const int32_t* __restrict__ t_emap,
                                                     const float4* __restrict__ t_ew, const float4* __restrict__ B_src,
                                                     const float4* __restrict__ A_src, const float4* __restrict__ q_src,
                                                     float4* __restrict__ A_dst, float4* __restrict__ B_dst,
                                                     float4* __restrict__ q_dst, PersistArgs pa, const SolveParams p_arg) {
  TileArgs a;
  a.tiles = tiles; a.t_vmap = t_vmap; a.t_emap = t_emap; a.t_eij = t_eij; a.t_ew = t_ew; a.t_srow = t_srow;
  a.A_src = A_src; a.B_src = B_src; a.q_src = q_src; a.A_dst = A_dst; a.B_dst = B_dst; a.q_dst = q_dst;
  a.p = p_arg; a.iters = iters_total; a.ntiles = ntiles; a.prof = nullptr;
  tile_body<NT, EPT, VPT, true, S12, FAT>(a, pa);
}

template <int NT, int EPT, int VPT, bool S12 = false>
hipError_t launch_tile_t(hipStream_t s, size_t lds, const TileArgs& a) {
  hipLaunchKernelGGL((k_tile<NT, EPT, VPT, S12>), dim3(a.ntiles), dim3(NT), lds, s, a.tiles, a.ntiles, a.iters, a.t_vmap,
                     a.t_srow, a.t_eij, a.t_emap, a.t_ew, a.B_src, a.A_src, a.q_src, a.A_dst, a.B_dst, a.q_dst, a.prof,
                     a.p);
  return hipGetLastError();
}

template <int NT, int EPT, int VPT, bool S12 = false>
hipError_t prepare_tile_t(size_t lds) {
  if (lds <= 48 * 1024) return hipSuccess;
  return hipFuncSetAttribute(reinterpret_cast<const void*>(k_tile<NT, EPT, VPT, S12>),
                             hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
}

// ------------------------------------------------------------------------------------------
// Costs (row a6): float32 terms, float64 block partials, summed on the host in block order.
// ------------------------------------------------------------------------------------------
constexpr int kCostBlocks = 256;

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  return v;
}

// (bid of nblocks: the kernel's own grid, or a range of blocks inside a fused launch)
__device__ __forceinline__ void costs_body(int bid, int nblocks, int32_t V, int32_t E, const int2* __restrict__ eij,
                                           const float4* __restrict__ ew, const float4* __restrict__ A,
                                           const float4* __restrict__ B, float lambda, double* __restrict__ partials,
                                           const uint8_t* __restrict__ emask, const uint8_t* __restrict__ vmask) {
  __shared__ double red[2][4];
  double s = 0.0, d = 0.0;
  for (int32_t e = bid * 256 + threadIdx.x; e < E; e += nblocks * 256) {
    if (emask && !emask[e]) continue;  // multi-GPU subdomain: only the edges this rank owns
    const int2 ij = eij[e];
    const float4 w = ew[e];
    const float4 ai = A[ij.x], aj = A[ij.y];
    float t = ai.x - aj.x;
    t = fmaf(-ai.y, w.z, t);
    t = fmaf(-ai.z, w.w, t);
    const float t1 = w.x * fabsf(t);
    const float t2 = w.y * fabsf(ai.y - aj.y);
    const float t3 = w.y * fabsf(ai.z - aj.z);
    s += (double)t1 + (double)t2 + (double)t3;
  }
  for (int32_t v = bid * 256 + threadIdx.x; v < V; v += nblocks * 256) {
    if (vmask && !vmask[v]) continue;
    const float4 a = A[v];
    const float c = (lambda * B[v].w) * fabsf(a.x - a.w);
    d += (double)c;
  }
  s = wave_sum(s);
  d = wave_sum(d);
  const int wid = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) { red[0][wid] = s; red[1][wid] = d; }
  __syncthreads();
  if (threadIdx.x == 0) {
    partials[2 * bid] = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
    partials[2 * bid + 1] = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
  }
}

__global__ __launch_bounds__(256) void k_costs(int32_t V, int32_t E, const int2* __restrict__ eij,
                                               const float4* __restrict__ ew,
                                               const float4* __restrict__ A,
                                               const float4* __restrict__ B, float lambda,
                                               double* __restrict__ partials,
                                               const uint8_t* __restrict__ emask,
                                               const uint8_t* __restrict__ vmask) {
  costs_body((int)blockIdx.x, (int)gridDim.x, V, E, eij, ew, A, B, lambda, partials, emask, vmask);
}

// ------------------------------------------------------------------------------------------
// Per-triangle stage (row a8): plane normal via Kinv back-projection, validity filters, vertex
// normals as the normalised sum of incident triangle normals in ascending triangle id.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ float dot3(float ax, float ay, float az, float bx, float by, float bz) {
  return fmaf(az, bz, fmaf(ay, by, ax * bx));
}

__device__ __forceinline__ void backproject(const float* K, float2 uv, float x, float& X,
                                            float& Y, float& Z) {
  const float depth = 1.0f / x;
  X = fmaf(K[0], uv.x, fmaf(K[1], uv.y, K[2])) * depth;
  Y = fmaf(K[3], uv.x, fmaf(K[4], uv.y, K[5])) * depth;
  Z = fmaf(K[6], uv.x, fmaf(K[7], uv.y, K[8])) * depth;
}

__device__ __forceinline__ void tri_body(int32_t t, int32_t T, const float2* __restrict__ pos,
                                         const float4* __restrict__ A, const int32_t* __restrict__ tris,
                                         const TriParamsDev& tp, float4* __restrict__ tri_normals,
                                         uint8_t* __restrict__ tri_valid, uint8_t* __restrict__ valid_out) {
  if (t >= T) return;
  const int32_t a = tris[3 * t], b = tris[3 * t + 1], c = tris[3 * t + 2];
  const float xa = A[a].x, xb = A[b].x, xc = A[c].x;
  const bool ok = isfinite(xa) && isfinite(xb) && isfinite(xc) && xa > 0.0f && xb > 0.0f && xc > 0.0f;
  if (!ok) {
    tri_normals[t] = make_float4(0.f, 0.f, 0.f, 0.f);
    tri_valid[t] = 0;
    if (valid_out) valid_out[t] = 0;
    return;
  }
  const float2 pa = pos[a], pb = pos[b], pc = pos[c];
  float Pax, Pay, Paz, Pbx, Pby, Pbz, Pcx, Pcy, Pcz;
  backproject(tp.Kinv, pa, xa, Pax, Pay, Paz);
  backproject(tp.Kinv, pb, xb, Pbx, Pby, Pbz);
  backproject(tp.Kinv, pc, xc, Pcx, Pcy, Pcz);
  const float e1x = Pbx - Pax, e1y = Pby - Pay, e1z = Pbz - Paz;
  const float e2x = Pcx - Pax, e2y = Pcy - Pay, e2z = Pcz - Paz;
  float nx = fmaf(e1y, e2z, -(e1z * e2y));
  float ny = fmaf(e1z, e2x, -(e1x * e2z));
  float nz = fmaf(e1x, e2y, -(e1y * e2x));
  const float len = sqrtf(dot3(nx, ny, nz, nx, ny, nz));
  if (len > 0.0f) { nx /= len; ny /= len; nz /= len; } else { nx = 0.f; ny = 0.f; nz = -1.f; }
  if (dot3(nx, ny, nz, Pax, Pay, Paz) > 0.0f) { nx = -nx; ny = -ny; nz = -nz; }
  uint8_t valid = 1;
  const float xmin = fminf(xa, fminf(xb, xc)), xmax = fmaxf(xa, fmaxf(xb, xc));
  if (tp.do_idepth && xmin < tp.min_idepth) valid = 0;
  if (tp.do_edge) {
    const float ux0 = pa.x - pb.x, ux1 = pb.x - pc.x, ux2 = pc.x - pa.x;
    const float uy0 = pa.y - pb.y, uy1 = pb.y - pc.y, uy2 = pc.y - pa.y;
    if (fmaf(ux0, ux0, uy0 * uy0) > tp.max_len2) valid = 0;
    if (fmaf(ux1, ux1, uy1 * uy1) > tp.max_len2) valid = 0;
    if (fmaf(ux2, ux2, uy2 * uy2) > tp.max_len2) valid = 0;
  }
  if (tp.do_oblique) {
    float rx = (Pax + Pbx) + Pcx, ry = (Pay + Pby) + Pcy, rz = (Paz + Pbz) + Pcz;
    const float rl = sqrtf(dot3(rx, ry, rz, rx, ry, rz));
    if (rl > 0.0f) {
      rx /= rl; ry /= rl; rz /= rl;
      const float cosang = -dot3(nx, ny, nz, rx, ry, rz);
      if (cosang < tp.cos_thresh) valid = 0;
    }
    const float diff = xmax - xmin;
    if (diff > tp.diff_abs && diff > tp.diff_factor * xmax) valid = 0;
  }
  tri_normals[t] = make_float4(nx, ny, nz, 0.f);
  tri_valid[t] = valid;
  if (valid_out) valid_out[t] = valid;
}

__global__ __launch_bounds__(256) void k_tri(int32_t T, const float2* __restrict__ pos,
                                             const float4* __restrict__ A,
                                             const int32_t* __restrict__ tris, TriParamsDev tp,
                                             float4* __restrict__ tri_normals,
                                             uint8_t* __restrict__ tri_valid, uint8_t* __restrict__ valid_out) {
  tri_body(blockIdx.x * 256 + threadIdx.x, T, pos, A, tris, tp, tri_normals, tri_valid, valid_out);
}

// The first launch of a frame's results stage (flame_hip_frame_results): blocks [0, nbt) run the
// triangle stage, the next ncb blocks the cost reduction (state not yet un-scaled: the caller fuses
// only when no scaling is pending), and every thread clears its share of the raster's owner map --
// three dependent launches of a small frame (~7 us of chain time each) in one.
__global__ __launch_bounds__(256) void k_frame_a(int32_t T, int nbt, const float2* __restrict__ pos,
                                                 const float4* __restrict__ A,
                                                 const int32_t* __restrict__ tris, TriParamsDev tp,
                                                 float4* __restrict__ tri_normals,
                                                 uint8_t* __restrict__ tri_valid, uint8_t* __restrict__ valid_out,
                                                 int ncb, int32_t V, int32_t E, const int2* __restrict__ eij,
                                                 const float4* __restrict__ ew, const float4* __restrict__ B,
                                                 float lambda, double* __restrict__ partials,
                                                 uint32_t* __restrict__ owner, int64_t npix) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < npix; i += (int64_t)gridDim.x * 256) owner[i] = 0xffffffffu;
  const int b = (int)blockIdx.x;
  if (b < nbt) tri_body(b * 256 + threadIdx.x, T, pos, A, tris, tp, tri_normals, tri_valid, valid_out);
  else costs_body(b - nbt, ncb, V, E, eij, ew, A, B, lambda, partials, nullptr, nullptr);
}

__device__ __forceinline__ void vtx_normals_body(int32_t v, int32_t V, const int32_t* __restrict__ trow,
                                                 const int32_t* __restrict__ tinc,
                                                 const float4* __restrict__ tri_normals,
                                                 float4* __restrict__ vtx_normals,
                                                 const int32_t* __restrict__ i2o, const float4* __restrict__ A,
                                                 float* __restrict__ out_x, float* __restrict__ out_n) {
  if (v >= V) return;
  float nx = 0.f, ny = 0.f, nz = 0.f;
  for (int32_t s = trow[v]; s < trow[v + 1]; ++s) {
    const float4 n = tri_normals[tinc[s]];
    nx += n.x; ny += n.y; nz += n.z;
  }
  const float len = sqrtf(dot3(nx, ny, nz, nx, ny, nz));
  if (len > 0.0f) { nx /= len; ny /= len; nz /= len; } else { nx = 0.f; ny = 0.f; nz = -1.f; }
  vtx_normals[v] = make_float4(nx, ny, nz, 0.f);
  // the frame's results leave in the CALLER's vertex order, written here instead of by two more
  // launches (flame_hip_frame_results: every dependent launch of a small frame costs ~5 us)
  if (i2o) {
    const int32_t o = i2o[v];
    if (out_x) out_x[o] = A[v].x;
    if (out_n) { out_n[3 * (size_t)o] = nx; out_n[3 * (size_t)o + 1] = ny; out_n[3 * (size_t)o + 2] = nz; }
  }
}

__global__ __launch_bounds__(256) void k_vtx_normals(int32_t V, const int32_t* __restrict__ trow,
                                                     const int32_t* __restrict__ tinc,
                                                     const float4* __restrict__ tri_normals,
                                                     float4* __restrict__ vtx_normals,
                                                     const int32_t* __restrict__ i2o, const float4* __restrict__ A,
                                                     float* __restrict__ out_x, float* __restrict__ out_n) {
  vtx_normals_body(blockIdx.x * 256 + threadIdx.x, V, trow, tinc, tri_normals, vtx_normals, i2o, A, out_x, out_n);
}

// ------------------------------------------------------------------------------------------
// Row a9: optional graph median / low-pass filter of the vertex idepths (Jacobi pass over the
// incidence CSR; see oracle/nltgv2_oracle.c nltgv2_graph_filter for the exact rule).  The median
// is found by rank counting (O(deg^2) compares, no per-thread arrays).
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_graph_filter(int32_t V, int32_t kind,
                                                      const int32_t* __restrict__ grow,
                                                      const int32_t* __restrict__ ginc,
                                                      const int2* __restrict__ eij,
                                                      const float4* __restrict__ A,
                                                      float* __restrict__ out) {
  const int32_t v = blockIdx.x * 256 + threadIdx.x;
  if (v >= V) return;
  const int32_t s0 = grow[v], n = grow[v + 1] - s0 + 1;
  auto val = [&](int32_t k) -> float {
    if (k == 0) return A[v].x;
    const int32_t ent = ginc[s0 + k - 1];
    const int2 ij = eij[ent & 0x7fffffff];
    return A[ent < 0 ? ij.x : ij.y].x;
  };
  if (kind == 0) {
    float med = A[v].x;
    for (int32_t i = 0; i < n; ++i) {
      const float xi = val(i);
      int32_t rank = 0;
      for (int32_t j = 0; j < n; ++j) {
        const float xj = val(j);
        rank += (xj < xi) || (xj == xi && j < i);
      }
      if (rank == (n - 1) / 2) med = xi;
    }
    out[v] = med;
  } else {
    float sum = A[v].x;
    for (int32_t k = 1; k < n; ++k) sum += val(k);
    out[v] = sum / (float)n;
  }
}

__global__ __launch_bounds__(256) void k_graph_filter_commit(int32_t V, const float* __restrict__ in,
                                                             float4* __restrict__ A,
                                                             float4* __restrict__ B) {
  const int32_t v = blockIdx.x * 256 + threadIdx.x;
  if (v >= V) return;
  A[v].x = in[v];
  B[v].x = in[v];
}

// ------------------------------------------------------------------------------------------
// Host boundary helpers for a plan that lives on the device (plan_dev.hip): the caller's arrays
// arrive and leave in the CALLER's vertex / edge order, permuted here instead of on the host.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_init_state(int32_t V, const int32_t* __restrict__ v_i2o,
                                                    const float2* __restrict__ pos_o,
                                                    const float* __restrict__ z, const float* __restrict__ wgt,
                                                    const float* __restrict__ x0, float4* __restrict__ A,
                                                    float4* __restrict__ B, float2* __restrict__ pos_i,
                                                    int32_t nq, float4* __restrict__ q0, float4* __restrict__ q1) {
  const int32_t k = blockIdx.x * 256 + threadIdx.x;
  // q = 0 in the buffers given (the dual state of a fresh upload), by the same launch
  for (int32_t e = k; e < nq; e += gridDim.x * 256) {
    if (q0) q0[e] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (q1) q1[e] = make_float4(0.f, 0.f, 0.f, 0.f);
  }
  if (k >= V) return;
  const int32_t o = v_i2o[k];
  const float zi = z[o];
  const float xi = x0 ? x0[o] : zi;
  A[k] = make_float4(xi, 0.f, 0.f, zi);
  B[k] = make_float4(xi, 0.f, 0.f, wgt[o]);
  if (pos_i) pos_i[k] = pos_o[o];
}

__global__ __launch_bounds__(256) void k_download_vertex(int32_t V, const int32_t* __restrict__ v_o2i,
                                                         const float4* __restrict__ S, float* __restrict__ out) {
  const int32_t o = blockIdx.x * 256 + threadIdx.x;
  if (o >= V) return;
  const float4 a = S[v_o2i[o]];
  out[o] = a.x; out[V + o] = a.y; out[2 * (size_t)V + o] = a.z;
}

__global__ __launch_bounds__(256) void k_download_rows3(int32_t n, const int32_t* __restrict__ o2i,
                                                        const float4* __restrict__ S, float* __restrict__ out) {
  const int32_t o = blockIdx.x * 256 + threadIdx.x;
  if (o >= n) return;
  const float4 a = S[o2i[o]];
  out[3 * (size_t)o] = a.x; out[3 * (size_t)o + 1] = a.y; out[3 * (size_t)o + 2] = a.z;
}

// flags |= 1 when any value is not finite
__global__ __launch_bounds__(256) void k_check_finite(int64_t n, const float* __restrict__ p, int32_t* flags) {
  const int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (k < n && !isfinite(p[k])) atomicOr(flags, 1);
}

// Row a7 epilogue: back to the caller's units after a solve on rescaled data (rescale_data,
// reference cfg/flame_offline_tum.yaml:90): primal state and data term times s.
__global__ __launch_bounds__(256) void k_scale_state(int32_t V, float4* __restrict__ A,
                                                     float4* __restrict__ B, float s) {
  const int32_t v = blockIdx.x * 256 + threadIdx.x;
  if (v >= V) return;
  float4 a = A[v], b = B[v];
  a.x *= s; a.y *= s; a.z *= s; a.w *= s;
  b.x *= s; b.y *= s; b.z *= s;
  A[v] = a; B[v] = b;
}

// ------------------------------------------------------------------------------------------
// "Next" row f1: mesh vertices in flame_ros::PointNormalUV layout (reference src/utils.h:47-53,
// packed at src/utils.cc:184-209): 3 float4 per vertex {p,0 | n,0 | u,v,0,0}; NaN xyz when the
// idepth is not a positive finite number.  Written in the CALLER's vertex order (i2o).
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_mesh(int32_t V, const float2* __restrict__ pos,
                                              const float4* __restrict__ A,
                                              const float4* __restrict__ vtx_normals,
                                              const int32_t* __restrict__ i2o, TriParamsDev tp,
                                              float wm1, float hm1, float4* __restrict__ out) {
  const int32_t v = blockIdx.x * 256 + threadIdx.x;
  if (v >= V) return;
  const float id = A[v].x;
  const float2 uv = pos[v];
  float4 o0 = make_float4(0.f, 0.f, 0.f, 0.f), o1 = o0, o2 = o0;
  if (!isnan(id) && id > 0.0f) {
    const float q0 = uv.x / id, q1 = uv.y / id, q2 = 1.0f / id;
    o0.x = (tp.Kinv[0] * q0 + tp.Kinv[1] * q1) + tp.Kinv[2] * q2;
    o0.y = (tp.Kinv[3] * q0 + tp.Kinv[4] * q1) + tp.Kinv[5] * q2;
    o0.z = (tp.Kinv[6] * q0 + tp.Kinv[7] * q1) + tp.Kinv[8] * q2;
    const float4 n = vtx_normals[v];
    o1 = make_float4(n.x, n.y, n.z, 0.f);
    o2.x = uv.x / wm1;
    o2.y = uv.y / hm1;
  } else {
    o0.x = o0.y = o0.z = __builtin_nanf("");
  }
  float4* o = out + 3 * (size_t)i2o[v];
  o[0] = o0; o[1] = o1; o[2] = o2;
}

// ------------------------------------------------------------------------------------------
// "Next" row f2: dense idepthmap / depthmap / point cloud.  Pass 1 (one wave per triangle, lanes
// stride over the bounding box): atomicMin of the triangle id into a per-pixel owner map, so the
// LOWEST covering triangle wins deterministically.  Pass 2 (one thread per pixel): barycentric
// idepth of the owner triangle, depth = 1/idepth (reference src/flame_offline_tum.cc:650-661),
// cloud = Kinv (jj d, ii d, d) within [min_depth, max_depth] (reference src/utils.cc:290-312).
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ float edge_fn(float ax, float ay, float bx, float by, float px, float py) {
  return fmaf(bx - ax, py - ay, -((by - ay) * (px - ax)));
}

// LPT lanes per triangle: 64 for meshes of large triangles (1.2 k vertices on 640 x 480: ~130 pixels
// each), 8 for dense ones (a 50 k-vertex mesh has 3-pixel triangles: a wave per triangle kept 9 lanes
// of 64 busy); with 8, a triangle whose bounding box exceeds 256 pixels (hull slivers) is rasterised by
// the whole wave afterwards.  The owner of a pixel is the smallest triangle id that covers it
// (atomicMin): the same map in any order.
__device__ __forceinline__ void raster_cover(float2 A, float2 B, float2 Cc, int x0, int y0, int bw, int n, int first,
                                             int step, int32_t width, uint32_t t, uint32_t* __restrict__ owner) {
  for (int k = first; k < n; k += step) {
    const int jj = x0 + k % bw, ii = y0 + k / bw;
    const float px = (float)jj, py = (float)ii;
    const float wa = edge_fn(B.x, B.y, Cc.x, Cc.y, px, py);
    const float wb = edge_fn(Cc.x, Cc.y, A.x, A.y, px, py);
    const float wc = edge_fn(A.x, A.y, B.x, B.y, px, py);
    const bool in = (wa >= 0.f && wb >= 0.f && wc >= 0.f) || (wa <= 0.f && wb <= 0.f && wc <= 0.f);
    if (in) atomicMin(owner + (size_t)ii * width + jj, t);
  }
}

template <int LPT>
__device__ __forceinline__ void raster_owner_body(int bid, int32_t T, int32_t width, int32_t height,
                                                  const float2* __restrict__ pos, const int32_t* __restrict__ tris,
                                                  const uint8_t* __restrict__ tri_valid, int32_t filtered,
                                                  uint32_t* __restrict__ owner) {
  const int lane = threadIdx.x & 63, sub = lane % LPT;
  const int32_t t = (bid * 4 + (threadIdx.x >> 6)) * (64 / LPT) + lane / LPT;
  bool live = t < T && !(filtered && !tri_valid[t]);
  float2 A = make_float2(0.f, 0.f), B = A, Cc = A;
  int x0 = 0, y0 = 0, bw = 0, bh = 0;
  if (live) {
    A = pos[tris[3 * t]]; B = pos[tris[3 * t + 1]]; Cc = pos[tris[3 * t + 2]];
    const float area = edge_fn(A.x, A.y, B.x, B.y, Cc.x, Cc.y);
    x0 = (int)ceilf(fminf(A.x, fminf(B.x, Cc.x)));
    y0 = (int)ceilf(fminf(A.y, fminf(B.y, Cc.y)));
    int x1 = (int)floorf(fmaxf(A.x, fmaxf(B.x, Cc.x))), y1 = (int)floorf(fmaxf(A.y, fmaxf(B.y, Cc.y)));
    x0 = max(x0, 0); y0 = max(y0, 0); x1 = min(x1, width - 1); y1 = min(y1, height - 1);
    bw = x1 - x0 + 1; bh = y1 - y0 + 1;
    live = area != 0.0f && bw > 0 && bh > 0;
  }
  const int n = live ? bw * bh : 0;
  if (LPT == 64 || n <= 256) raster_cover(A, B, Cc, x0, y0, bw, n, sub, LPT, width, (uint32_t)t, owner);
  if (LPT == 64) return;
  unsigned long long big = __ballot(n > 256 && sub == 0);
  while (big) {  // wave-uniform: every lane takes the triangle of lane l
    const int l = (int)__builtin_ctzll(big);
    big &= big - 1;
    const float2 a = make_float2(__shfl(A.x, l, 64), __shfl(A.y, l, 64));
    const float2 b = make_float2(__shfl(B.x, l, 64), __shfl(B.y, l, 64));
    const float2 c = make_float2(__shfl(Cc.x, l, 64), __shfl(Cc.y, l, 64));
    raster_cover(a, b, c, __shfl(x0, l, 64), __shfl(y0, l, 64), __shfl(bw, l, 64), __shfl(n, l, 64), lane, 64, width,
                 (uint32_t)__shfl(t, l, 64), owner);
  }
}

template <int LPT>
__global__ __launch_bounds__(256) void k_raster_owner(int32_t T, int32_t width, int32_t height,
                                                      const float2* __restrict__ pos,
                                                      const int32_t* __restrict__ tris,
                                                      const uint8_t* __restrict__ tri_valid,
                                                      int32_t filtered, uint32_t* __restrict__ owner) {
  raster_owner_body<LPT>((int)blockIdx.x, T, width, height, pos, tris, tri_valid, filtered, owner);
}

// The second launch of a frame's results stage: both halves only need the triangle stage's outputs --
// blocks [0, nbv) make the vertex normals (and the frame's per-vertex outputs), the rest rasterise the
// owner map.
template <int LPT>
__global__ __launch_bounds__(256) void k_frame_b(int32_t V, int nbv, const int32_t* __restrict__ trow,
                                                 const int32_t* __restrict__ tinc,
                                                 const float4* __restrict__ tri_normals,
                                                 float4* __restrict__ vtx_normals,
                                                 const int32_t* __restrict__ i2o, const float4* __restrict__ A,
                                                 float* __restrict__ out_x, float* __restrict__ out_n, int32_t T,
                                                 int32_t width, int32_t height, const float2* __restrict__ pos,
                                                 const int32_t* __restrict__ tris,
                                                 const uint8_t* __restrict__ tri_valid, int32_t filtered,
                                                 uint32_t* __restrict__ owner) {
  const int b = (int)blockIdx.x;
  if (b < nbv) vtx_normals_body(b * 256 + threadIdx.x, V, trow, tinc, tri_normals, vtx_normals, i2o, A, out_x, out_n);
  else raster_owner_body<LPT>(b - nbv, T, width, height, pos, tris, tri_valid, filtered, owner);
}

__global__ __launch_bounds__(256) void k_raster_fill(int32_t width, int32_t height,
                                                     const float2* __restrict__ pos,
                                                     const float4* __restrict__ A,
                                                     const int32_t* __restrict__ tris,
                                                     const uint32_t* __restrict__ owner,
                                                     TriParamsDev tp, float min_depth, float max_depth,
                                                     float* __restrict__ idm, float* __restrict__ dm,
                                                     float* __restrict__ cloud, uint32_t* __restrict__ covered) {
  const int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const bool inside = k < (int64_t)width * height;
  const int jj = inside ? (int)(k % width) : 0, ii = inside ? (int)(k / width) : 0;
  const uint32_t t = inside ? owner[k] : 0xffffffffu;
  float id = __builtin_nanf("");
  if (t != 0xffffffffu) {
    const int32_t a = tris[3 * t], b = tris[3 * t + 1], c = tris[3 * t + 2];
    const float2 Pa = pos[a], Pb = pos[b], Pc = pos[c];
    const float px = (float)jj, py = (float)ii;
    const float wa = edge_fn(Pb.x, Pb.y, Pc.x, Pc.y, px, py);
    const float wb = edge_fn(Pc.x, Pc.y, Pa.x, Pa.y, px, py);
    const float wc = edge_fn(Pa.x, Pa.y, Pb.x, Pb.y, px, py);
    const float num = fmaf(wc, A[c].x, fmaf(wb, A[b].x, wa * A[a].x));
    id = num / ((wa + wb) + wc);
  }
  if (covered) {  // stat key `coverage`: pixels of the map that are not NaN, one count per block
    // (summed on the host; a same-address atomic per wave serialised this kernel to ~60 us)
    __shared__ uint32_t wcnt[4];
    const unsigned long long m = __ballot(inside && !isnan(id));
    if ((threadIdx.x & 63) == 0) wcnt[threadIdx.x >> 6] = (uint32_t)__popcll(m);
    __syncthreads();
    if (threadIdx.x == 0) covered[blockIdx.x] = (wcnt[0] + wcnt[1]) + (wcnt[2] + wcnt[3]);
  }
  if (!inside) return;
  idm[k] = id;
  float depth = __builtin_nanf("");
  if (!isnan(id) && id > 0.0f) depth = 1.0f / id;
  if (dm) dm[k] = depth;
  if (cloud) {
    float ox, oy, oz;
    if (isnan(depth) || depth < min_depth || depth > max_depth) {
      ox = oy = oz = __builtin_nanf("");
    } else {
      const float q0 = (float)jj * depth, q1 = (float)ii * depth, q2 = depth;
      ox = (tp.Kinv[0] * q0 + tp.Kinv[1] * q1) + tp.Kinv[2] * q2;
      oy = (tp.Kinv[3] * q0 + tp.Kinv[4] * q1) + tp.Kinv[5] * q2;
      oz = (tp.Kinv[6] * q0 + tp.Kinv[7] * q1) + tp.Kinv[8] * q2;
    }
    cloud[3 * k] = ox; cloud[3 * k + 1] = oy; cloud[3 * k + 2] = oz;
  }
}

// ------------------------------------------------------------------------------------------
// Debug images of flame::Flame (reference src/flame_offline_tum.cc:731-766; what each shows:
// cfg/flame_offline_tum.yaml:58-64), rendered on the device into a BGR8 buffer so that update()
// never draws on the host; the rules are stated in oracle/nltgv2_oracle.c (nltgv2_debug_image).
// "Later primitive overwrites earlier ones" is made order-free by a key map: every primitive
// atomicMax-es its index + 1 into the pixels it covers, a second pass colours each pixel from the
// winning primitive.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ float ramp01(float v) { return fmaxf(0.0f, fminf(1.0f, v)); }
__device__ __forceinline__ void put_jet(uint8_t* o, float v) {  // jet(v, 0, 2), BGR
  float t = (v - 0.0f) / (2.0f - 0.0f);
  t = ramp01(t);
  const float r = ramp01(1.5f - fabsf(4.0f * t - 3.0f));
  const float g = ramp01(1.5f - fabsf(4.0f * t - 2.0f));
  const float b = ramp01(1.5f - fabsf(4.0f * t - 1.0f));
  o[0] = (uint8_t)(255.0f * b + 0.5f);
  o[1] = (uint8_t)(255.0f * g + 0.5f);
  o[2] = (uint8_t)(255.0f * r + 0.5f);
}
__device__ __forceinline__ int round_px(float v) { return (int)(v + (v >= 0.0f ? 0.5f : -0.5f)); }

// one thread per triangle side: Bresenham between the rounded end points
__global__ __launch_bounds__(256) void k_dbg_lines(int32_t T, int32_t W, int32_t H, const float2* __restrict__ pos,
                                                   const int32_t* __restrict__ tris,
                                                   const uint8_t* __restrict__ tri_valid, uint32_t* __restrict__ key) {
  const int32_t i = blockIdx.x * 256 + threadIdx.x;
  if (i >= 3 * T) return;
  const int32_t t = i / 3, k = i - 3 * t;
  if (!tri_valid[t]) return;
  const float2 pa = pos[tris[3 * t + k]], pb = pos[tris[3 * t + (k + 1) % 3]];
  int x0 = round_px(pa.x), y0 = round_px(pa.y);
  const int x1 = round_px(pb.x), y1 = round_px(pb.y);
  const int dx = abs(x1 - x0), dy = -abs(y1 - y0);
  const int sx = x0 < x1 ? 1 : -1, sy = y0 < y1 ? 1 : -1;
  int err = dx + dy;
  for (int guard = 0; guard < 4 * (W + H); ++guard) {
    if (x0 >= 0 && y0 >= 0 && x0 < W && y0 < H) atomicMax(key + (size_t)y0 * W + x0, (uint32_t)i + 1u);
    if (x0 == x1 && y0 == y1) break;
    const int e2 = 2 * err;
    if (e2 >= dy) { err += dy; x0 += sx; }
    if (e2 <= dx) { err += dx; y0 += sy; }
  }
}

__global__ __launch_bounds__(256) void k_dbg_wire_color(int64_t npix, const uint32_t* __restrict__ key,
                                                        const int32_t* __restrict__ tris,
                                                        const float4* __restrict__ A, float scale,
                                                        uint8_t* __restrict__ bgr) {
  const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (p >= npix) return;
  uint8_t* o = bgr + 3 * p;
  const uint32_t kk = key[p];
  if (kk == 0) { o[0] = 0; o[1] = 0; o[2] = 0; return; }
  const int32_t i = (int32_t)(kk - 1), t = i / 3, k = i - 3 * t;
  const float xa = A[tris[3 * t + k]].x, xb = A[tris[3 * t + (k + 1) % 3]].x;
  put_jet(o, (0.5f * (xa + xb)) * scale);
}

// feat = n x {u, v, mu}
__global__ __launch_bounds__(256) void k_dbg_feat_mark(int32_t n, int32_t W, int32_t H, const float* __restrict__ feat,
                                                       uint32_t* __restrict__ key) {
  const int32_t f = blockIdx.x * 256 + threadIdx.x;
  if (f >= n) return;
  const int px = round_px(feat[3 * f]), py = round_px(feat[3 * f + 1]);
  for (int dy = -1; dy <= 1; ++dy)
    for (int dx = -1; dx <= 1; ++dx) {
      const int x = px + dx, y = py + dy;
      if (x >= 0 && y >= 0 && x < W && y < H) atomicMax(key + (size_t)y * W + x, (uint32_t)f + 1u);
    }
}

__global__ __launch_bounds__(256) void k_dbg_feat_color(int64_t npix, const uint32_t* __restrict__ key,
                                                        const float* __restrict__ feat, float scale,
                                                        uint8_t* __restrict__ bgr) {
  const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (p >= npix) return;
  uint8_t* o = bgr + 3 * p;
  const uint32_t kk = key[p];
  if (kk == 0) { o[0] = 0; o[1] = 0; o[2] = 0; return; }
  put_jet(o, feat[3 * (size_t)(kk - 1) + 2] * scale);
}

__global__ __launch_bounds__(256) void k_dbg_idm_color(int64_t npix, const float* __restrict__ idm, float scale,
                                                       uint8_t* __restrict__ bgr) {
  const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (p >= npix) return;
  uint8_t* o = bgr + 3 * p;
  const float id = idm[p];
  if (isnan(id)) { o[0] = 0; o[1] = 0; o[2] = 0; return; }
  put_jet(o, id * scale);
}

// "Image colored by interpolated normal vectors" (cfg/flame_offline_tum.yaml:62): barycentric blend
// of the three vertex normals of the pixel's owner triangle (owner map of the FILTERED raster)
__global__ __launch_bounds__(256) void k_dbg_normals(int32_t W, int32_t H, const float2* __restrict__ pos,
                                                     const int32_t* __restrict__ tris,
                                                     const uint32_t* __restrict__ owner,
                                                     const float4* __restrict__ vn, uint8_t* __restrict__ bgr) {
  const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (p >= (int64_t)W * H) return;
  uint8_t* o = bgr + 3 * p;
  const uint32_t t = owner[p];
  if (t == 0xffffffffu) { o[0] = 0; o[1] = 0; o[2] = 0; return; }
  const int jj = (int)(p % W), ii = (int)(p / W);
  const int32_t a = tris[3 * t], b = tris[3 * t + 1], c = tris[3 * t + 2];
  const float2 Pa = pos[a], Pb = pos[b], Pc = pos[c];
  const float px = (float)jj, py = (float)ii;
  const float wa = edge_fn(Pb.x, Pb.y, Pc.x, Pc.y, px, py);
  const float wb = edge_fn(Pc.x, Pc.y, Pa.x, Pa.y, px, py);
  const float wc = edge_fn(Pa.x, Pa.y, Pb.x, Pb.y, px, py);
  const float s = (wa + wb) + wc;
  const float4 na = vn[a], nb = vn[b], nc = vn[c];
  float nx = fmaf(wc, nc.x, fmaf(wb, nb.x, wa * na.x)) / s;
  float ny = fmaf(wc, nc.y, fmaf(wb, nb.y, wa * na.y)) / s;
  float nz = fmaf(wc, nc.z, fmaf(wb, nb.z, wa * na.z)) / s;
  const float len = sqrtf(fmaf(nz, nz, fmaf(ny, ny, nx * nx)));
  if (len > 0.0f) { nx /= len; ny /= len; nz /= len; } else { nx = 0.f; ny = 0.f; nz = -1.f; }
  o[0] = (uint8_t)(255.0f * (0.5f * nz + 0.5f) + 0.5f);
  o[1] = (uint8_t)(255.0f * (0.5f * ny + 0.5f) + 0.5f);
  o[2] = (uint8_t)(255.0f * (0.5f * nx + 0.5f) + 0.5f);
}

// ------------------------------------------------------------------------------------------
// Halo exchange (multi-GPU subdomains): gather the state that CHANGES of the listed
// own vertices / edges into a contiguous send buffer, scatter a received buffer into halo entries.
// Layout: nv x {x, w1, w2, xb, w1b, w2b} (24 B), then ne x {q1, q2, q3} (12 B).
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_halo_pack(int32_t nv, int32_t ne,
                                                   const int32_t* __restrict__ vidx,
                                                   const int32_t* __restrict__ eidx,
                                                   const float4* __restrict__ A,
                                                   const float4* __restrict__ B,
                                                   const float4* __restrict__ q,
                                                   float* __restrict__ out) {
  const int32_t t = blockIdx.x * 256 + threadIdx.x;
  if (t < nv) {
    const int32_t v = vidx[t];
    const float4 a = A[v], b = B[v];
    float2* o = reinterpret_cast<float2*>(out + 6 * (size_t)t);  // 24-byte records: 8-byte aligned
    o[0] = make_float2(a.x, a.y);
    o[1] = make_float2(a.z, b.x);
    o[2] = make_float2(b.y, b.z);
  } else if (t < nv + ne) {
    const float4 qq = q[eidx[t - nv]];
    float* o = out + 6 * (size_t)nv + 3 * (size_t)(t - nv);
    o[0] = qq.x; o[1] = qq.y; o[2] = qq.z;
  }
}

__global__ __launch_bounds__(256) void k_halo_unpack(int32_t nv, int32_t ne,
                                                     const int32_t* __restrict__ vidx,
                                                     const int32_t* __restrict__ eidx,
                                                     const float* __restrict__ in,
                                                     float4* __restrict__ A, float4* __restrict__ B,
                                                     float4* __restrict__ q) {
  const int32_t t = blockIdx.x * 256 + threadIdx.x;
  if (t < nv) {
    const int32_t v = vidx[t];
    const float2* r = reinterpret_cast<const float2*>(in + 6 * (size_t)t);
    const float2 r0 = r[0], r1 = r[1], r2 = r[2];
    // the data term z (A.w) and the data weight (B.w) are constants the receiver already holds
    A[v].x = r0.x; A[v].y = r0.y; A[v].z = r1.x;
    B[v].x = r1.y; B[v].y = r2.x; B[v].z = r2.y;
  } else if (t < nv + ne) {
    const float* r = in + 6 * (size_t)nv + 3 * (size_t)(t - nv);
    q[eidx[t - nv]] = make_float4(r[0], r[1], r[2], 0.0f);
  }
}

// ---- peer transport (kernels.h HaloXArgs) ----
__device__ __forceinline__ int halo_seg_of(const HaloSegDev* segs, int nsegs, int t) {
  int lo = 0, hi = nsegs - 1;  // the last segment whose first <= t
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (segs[mid].first <= t) lo = mid; else hi = mid - 1;
  }
  return lo;
}

__global__ __launch_bounds__(256) void k_halo_push(HaloXArgs a) {
  const int32_t t = blockIdx.x * 256 + threadIdx.x;
  if (t < a.total) {
    const HaloSegDev& S = a.segs[halo_seg_of(a.segs, a.nsegs, t)];
    const HaloPartDev& P = a.parts[S.part];
    const int c = (a.cur_mask >> S.part) & 1u, r = t - S.first;
    // records of whole 16-byte words (vertex 2, edge 1: kPeerVRec / kPeerERec floats): the inboxes are UNCACHED memory, where a
    // 4- or 8-byte store is a memory transaction of its own -- the packed 24 / 12-byte records of the RCCL path took 88 us per
    // exchange of 0.7 MB this way (profiles/r06_peer_transport_ab.txt)
    float4* out = reinterpret_cast<float4*>(S.buf[a.epoch & 1]);
    if (S.didx) {  // a part of this rank: from state to state (what it sends it owns, what the other receives is halo: disjoint)
      const HaloPartDev& D = a.parts[S.dpart];
      const int dc = (a.cur_mask >> S.dpart) & 1u;
      if (S.kind == 0) {
        const int32_t v = S.idx[r], u = S.didx[r];
        const float4 x = P.A[c][v], b = P.B[c][v];
        D.A[dc][u].x = x.x; D.A[dc][u].y = x.y; D.A[dc][u].z = x.z;  // (z and the data weight are the receiver's own constants)
        D.B[dc][u].x = b.x; D.B[dc][u].y = b.y; D.B[dc][u].z = b.z;
      } else {
        const float4 qq = P.q[c][S.idx[r]];
        D.q[dc][S.didx[r]] = make_float4(qq.x, qq.y, qq.z, 0.f);
      }
    } else if (S.kind == 0) {
      const int32_t v = S.idx[r];
      const float4 x = P.A[c][v], b = P.B[c][v];
      out[2 * (size_t)r] = make_float4(x.x, x.y, x.z, b.x);
      out[2 * (size_t)r + 1] = make_float4(b.y, b.z, 0.f, 0.f);
    } else {
      const float4 qq = P.q[c][S.idx[r]];
      out[r] = make_float4(qq.x, qq.y, qq.z, 0.f);
    }
  }
  // Every record of this block has been ACKNOWLEDGED before the block counts itself done (an inbox that other ranks read while
  // this kernel runs is uncached memory: an acknowledged store is in memory; the single-process inbox is ordered by the kernel
  // boundary anyway); the LAST block raises the flags behind ONE system-scope fence.  (A system- or agent-scope release per
  // block is an L2 write-back each -- buffer_wbl2 -- and cost 76 us over the 755 blocks of a 4 MB exchange: r06_peer_trace.)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __shared__ int s_last;
  __syncthreads();
  if (threadIdx.x == 0) s_last = (__hip_atomic_fetch_add(a.counter, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (int)gridDim.x - 1);
  __syncthreads();
  if (s_last) {
    __threadfence_system();
    for (int s = threadIdx.x; s < a.nsegs; s += 256)
      if (a.segs[s].flag) __hip_atomic_store(a.segs[s].flag, a.epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    if (threadIdx.x == 0) __hip_atomic_store(a.counter, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

__global__ __launch_bounds__(256) void k_halo_pull(HaloXArgs a) {
  // every block waits for every incoming message of the rank (a few dozen words at most), then unpacks its records
  const unsigned long long w0 = wall_clock64();
  for (int s = threadIdx.x; s < a.nsegs; s += 256) {
    while (__hip_atomic_load(a.segs[s].flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) < a.epoch) {
      if (wall_clock64() - w0 > (unsigned long long)a.timeout_ticks) { *a.err = 1; break; }  // never hang: the host reports it
      __builtin_amdgcn_s_sleep(2);
    }
  }
  __atomic_thread_fence(__ATOMIC_ACQUIRE);  // (one acquire per block behind the polls, not one cache invalidate per poll)
  __syncthreads();
  const int32_t t = blockIdx.x * 256 + threadIdx.x;
  if (t >= a.total) return;
  const HaloSegDev& S = a.segs[halo_seg_of(a.segs, a.nsegs, t)];
  const HaloPartDev& P = a.parts[S.part];
  const int c = (a.cur_mask >> S.part) & 1u, r = t - S.first;
  const float4* in = reinterpret_cast<const float4*>(S.buf[a.epoch & 1]);
  if (S.kind == 0) {
    const int32_t v = S.idx[r];
    const float4 r0 = in[2 * (size_t)r], r1 = in[2 * (size_t)r + 1];
    // (the data term z (A.w) and the data weight (B.w) are constants the receiver already holds)
    P.A[c][v].x = r0.x; P.A[c][v].y = r0.y; P.A[c][v].z = r0.z;
    P.B[c][v].x = r0.w; P.B[c][v].y = r1.x; P.B[c][v].z = r1.y;
  } else {
    const float4 t4 = in[r];
    P.q[c][S.idx[r]] = make_float4(t4.x, t4.y, t4.z, 0.0f);
  }
}

}  // namespace

hipError_t launch_halo_push(hipStream_t s, const HaloXArgs& a) {
  if (a.nsegs <= 0) return hipSuccess;
  hipLaunchKernelGGL(k_halo_push, dim3((unsigned)std::max(1, (a.total + 255) / 256)), dim3(256), 0, s, a);
  return hipGetLastError();
}

hipError_t launch_halo_pull(hipStream_t s, const HaloXArgs& a) {
  if (a.nsegs <= 0) return hipSuccess;
  hipLaunchKernelGGL(k_halo_pull, dim3((unsigned)std::max(1, (a.total + 255) / 256)), dim3(256), 0, s, a);
  return hipGetLastError();
}

hipError_t launch_graph_filter(hipStream_t s, int32_t V, int32_t kind, const int32_t* grow,
                               const int32_t* ginc, const int2* eij, float4* A, float4* B, float* tmp) {
  if (V <= 0) return hipSuccess;
  hipLaunchKernelGGL(k_graph_filter, dim3((V + 255) / 256), dim3(256), 0, s, V, kind, grow, ginc, eij, A, tmp);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(k_graph_filter_commit, dim3((V + 255) / 256), dim3(256), 0, s, V, tmp, A, B);
  return hipGetLastError();
}

hipError_t launch_init_state(hipStream_t s, int32_t V, const int32_t* v_i2o, const float2* pos_o, const float* z,
                             const float* wgt, const float* x0, float4* A, float4* B, float2* pos_i, int32_t nq,
                             float4* q0, float4* q1) {
  if (V <= 0 && nq <= 0) return hipSuccess;
  hipLaunchKernelGGL(k_init_state, dim3((std::max(V, 1) + 255) / 256), dim3(256), 0, s, V, v_i2o, pos_o, z, wgt, x0, A, B,
                     pos_i, nq, q0, q1);
  return hipGetLastError();
}

hipError_t launch_download_vertex(hipStream_t s, int32_t V, const int32_t* v_o2i, const float4* S, float* out) {
  if (V <= 0) return hipSuccess;
  hipLaunchKernelGGL(k_download_vertex, dim3((V + 255) / 256), dim3(256), 0, s, V, v_o2i, S, out);
  return hipGetLastError();
}

hipError_t launch_download_rows3(hipStream_t s, int32_t n, const int32_t* o2i, const float4* S, float* out) {
  if (n <= 0) return hipSuccess;
  hipLaunchKernelGGL(k_download_rows3, dim3((n + 255) / 256), dim3(256), 0, s, n, o2i, S, out);
  return hipGetLastError();
}

hipError_t launch_check_finite(hipStream_t s, int64_t n, const float* p, int32_t* flags) {
  if (n <= 0 || !p) return hipSuccess;
  hipLaunchKernelGGL(k_check_finite, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, n, p, flags);
  return hipGetLastError();
}

hipError_t launch_scale_state(hipStream_t s, int32_t V, float4* A, float4* B, float scale) {
  if (V <= 0) return hipSuccess;
  hipLaunchKernelGGL(k_scale_state, dim3((V + 255) / 256), dim3(256), 0, s, V, A, B, scale);
  return hipGetLastError();
}

hipError_t launch_mesh(hipStream_t s, int32_t V, const float2* pos, const float4* A,
                       const float4* vtx_normals, const int32_t* i2o, TriParamsDev tp, int32_t width,
                       int32_t height, float4* out) {
  if (V <= 0) return hipSuccess;
  hipLaunchKernelGGL(k_mesh, dim3((V + 255) / 256), dim3(256), 0, s, V, pos, A, vtx_normals, i2o, tp,
                     (float)(width - 1), (float)(height - 1), out);
  return hipGetLastError();
}

hipError_t launch_raster(hipStream_t s, int32_t T, int32_t width, int32_t height, const float2* pos,
                         const float4* A, const int32_t* tris, const uint8_t* tri_valid,
                         int32_t filtered, TriParamsDev tp, float min_depth, float max_depth,
                         uint32_t* owner, float* idm, float* dm, float* cloud, uint32_t* covered) {
  const int64_t npix = (int64_t)width * height;
  if (npix <= 0) return hipSuccess;
  hipError_t e = hipMemsetAsync(owner, 0xff, sizeof(uint32_t) * (size_t)npix, s);
  if (e != hipSuccess) return e;
  if (T > 0) {
    if (npix / T >= 64)  // mean triangle area in pixels
      hipLaunchKernelGGL(k_raster_owner<64>, dim3((T + 3) / 4), dim3(256), 0, s, T, width, height, pos, tris, tri_valid,
                         filtered, owner);
    else
      hipLaunchKernelGGL(k_raster_owner<8>, dim3((T + 31) / 32), dim3(256), 0, s, T, width, height, pos, tris, tri_valid,
                         filtered, owner);
    if ((e = hipGetLastError()) != hipSuccess) return e;
  }
  hipLaunchKernelGGL(k_raster_fill, dim3((unsigned)((npix + 255) / 256)), dim3(256), 0, s, width, height,
                     pos, A, tris, owner, tp, min_depth, max_depth, idm, dm, cloud, covered);
  return hipGetLastError();
}

// triangle stage + costs + dense raster of one frame in three launches (k_frame_a, k_frame_b, k_raster_fill)
hipError_t launch_frame_stage(hipStream_t s, int32_t V, int32_t E, int32_t T, int32_t width, int32_t height,
                              const float2* pos, const float4* A, const float4* B, const int2* eij, const float4* ew,
                              const int32_t* tris, const int32_t* trow, const int32_t* tinc, TriParamsDev tp,
                              float4* tri_normals, uint8_t* tri_valid, float4* vtx_normals, const FrameOut* fo,
                              float lambda, double* partials, int32_t filtered, float min_depth, float max_depth,
                              uint32_t* owner, float* idm, float* dm, float* cloud, uint32_t* covered) {
  const int64_t npix = (int64_t)width * height;
  const int nbt = (T + 255) / 256, ncb = partials ? kCostBlocks : 0, nbv = (V + 255) / 256;
  if (T <= 0 || V <= 0 || npix <= 0) return hipErrorInvalidValue;
  hipLaunchKernelGGL(k_frame_a, dim3(nbt + ncb), dim3(256), 0, s, T, nbt, pos, A, tris, tp, tri_normals, tri_valid,
                     fo ? fo->tri_valid : nullptr, ncb, V, E, eij, ew, B, lambda, partials, owner, npix);
  if (npix / T >= 64)  // mean triangle area in pixels
    hipLaunchKernelGGL(k_frame_b<64>, dim3(nbv + (T + 3) / 4), dim3(256), 0, s, V, nbv, trow, tinc, tri_normals, vtx_normals,
                       fo ? fo->v_i2o : nullptr, A, fo ? fo->x : nullptr, fo ? fo->normals : nullptr, T, width, height, pos,
                       tris, tri_valid, filtered, owner);
  else
    hipLaunchKernelGGL(k_frame_b<8>, dim3(nbv + (T + 31) / 32), dim3(256), 0, s, V, nbv, trow, tinc, tri_normals, vtx_normals,
                       fo ? fo->v_i2o : nullptr, A, fo ? fo->x : nullptr, fo ? fo->normals : nullptr, T, width, height, pos,
                       tris, tri_valid, filtered, owner);
  hipLaunchKernelGGL(k_raster_fill, dim3((unsigned)((npix + 255) / 256)), dim3(256), 0, s, width, height, pos, A, tris, owner,
                     tp, min_depth, max_depth, idm, dm, cloud, covered);
  return hipGetLastError();
}

hipError_t launch_debug_image(hipStream_t s, int32_t kind, int32_t T, int32_t width, int32_t height, const float2* pos,
                              const float4* A, const int32_t* tris, const uint8_t* tri_valid, const uint32_t* owner,
                              const float* idm, const float4* vtx_normals, int32_t n_feat, const float* feat,
                              float scale, uint32_t* key, uint8_t* bgr) {
  const int64_t npix = (int64_t)width * height;
  if (npix <= 0) return hipSuccess;
  const dim3 gp((unsigned)((npix + 255) / 256)), b(256);
  hipError_t e;
  if (kind == 0 || kind == 1) {
    if ((e = hipMemsetAsync(key, 0, sizeof(uint32_t) * (size_t)npix, s)) != hipSuccess) return e;
    if (kind == 0) {
      if (T > 0) hipLaunchKernelGGL(k_dbg_lines, dim3((3 * T + 255) / 256), b, 0, s, T, width, height, pos, tris, tri_valid, key);
      hipLaunchKernelGGL(k_dbg_wire_color, gp, b, 0, s, npix, key, tris, A, scale, bgr);
    } else {
      if (n_feat > 0) hipLaunchKernelGGL(k_dbg_feat_mark, dim3((n_feat + 255) / 256), b, 0, s, n_feat, width, height, feat, key);
      hipLaunchKernelGGL(k_dbg_feat_color, gp, b, 0, s, npix, key, feat, scale, bgr);
    }
  } else if (kind == 2) {
    hipLaunchKernelGGL(k_dbg_normals, gp, b, 0, s, width, height, pos, tris, owner, vtx_normals, bgr);
  } else {
    hipLaunchKernelGGL(k_dbg_idm_color, gp, b, 0, s, npix, idm, scale, bgr);
  }
  return hipGetLastError();
}

hipError_t launch_halo_pack(hipStream_t s, int32_t nv, int32_t ne, const int32_t* vidx,
                            const int32_t* eidx, const float4* A, const float4* B, const float4* q,
                            float* out) {
  if (nv + ne <= 0) return hipSuccess;
  hipLaunchKernelGGL(k_halo_pack, dim3((nv + ne + 255) / 256), dim3(256), 0, s, nv, ne, vidx, eidx, A, B, q, out);
  return hipGetLastError();
}

hipError_t launch_halo_unpack(hipStream_t s, int32_t nv, int32_t ne, const int32_t* vidx,
                              const int32_t* eidx, const float* in, float4* A, float4* B, float4* q) {
  if (nv + ne <= 0) return hipSuccess;
  hipLaunchKernelGGL(k_halo_unpack, dim3((nv + ne + 255) / 256), dim3(256), 0, s, nv, ne, vidx, eidx, in, A, B, q);
  return hipGetLastError();
}

hipError_t launch_dual(hipStream_t s, int32_t E, const int2* eij, const float4* ew,
                       const float4* B, float4* q, float sigma) {
  if (E <= 0) return hipSuccess;
  hipLaunchKernelGGL(k_dual, dim3((E + 255) / 256), dim3(256), 0, s, E, eij, ew, B, q, sigma);
  return hipGetLastError();
}

hipError_t launch_primal(hipStream_t s, int32_t V, const int32_t* grow, const int32_t* ginc,
                         const float4* ew, const float4* q, float4* A, float4* B, SolveParams p) {
  if (V <= 0) return hipSuccess;
  hipLaunchKernelGGL(k_primal, dim3((V + 255) / 256), dim3(256), 0, s, V, grow, ginc, ew, q, A, B, p);
  return hipGetLastError();
}

#define FLAME_TILE_CFGS(X)                                                             \
  X(256, 2, 1) X(256, 3, 1) X(256, 4, 1) X(256, 6, 1) X(256, 4, 2) X(256, 6, 2)         \
  X(512, 2, 1) X(512, 3, 1) X(512, 4, 1) X(512, 6, 1) X(512, 4, 2) X(512, 6, 2)         \
  X(1024, 2, 1) X(1024, 3, 1) X(1024, 4, 1) X(1024, 6, 1) X(1024, 4, 2) X(1024, 6, 2)

// 12-byte incidence slots (fat tiles, SlotMem<true>): the configurations a one-tile-per-CU partition of a graph beyond
// 256 x 196 vertices gets
#define FLAME_S12_CFGS(X) X(1024, 2, 1) X(1024, 3, 1)

bool tile_config_exists(int nt, int ept, int vpt) {
#define X(N, Ep, Vp) if (nt == N && ept == Ep && vpt == Vp) return true;
  FLAME_TILE_CFGS(X)
#undef X
  return false;
}
bool tile_slot12_exists(int nt, int ept, int vpt) {
#define X(N, Ep, Vp) if (nt == N && ept == Ep && vpt == Vp) return true;
  FLAME_S12_CFGS(X)
#undef X
  return false;
}

hipError_t launch_tile(hipStream_t s, int nt, int ept, int vpt, size_t lds_bytes,
                       const TileArgs& a) {
  if (a.ntiles <= 0 || a.iters <= 0) return hipSuccess;
  if (a.slot12) {
#define X(N, Ep, Vp) if (nt == N && ept == Ep && vpt == Vp) return launch_tile_t<N, Ep, Vp, true>(s, lds_bytes, a);
    FLAME_S12_CFGS(X)
#undef X
    return hipErrorInvalidConfiguration;
  }
#define X(N, Ep, Vp) if (nt == N && ept == Ep && vpt == Vp) return launch_tile_t<N, Ep, Vp>(s, lds_bytes, a);
  FLAME_TILE_CFGS(X)
#undef X
  return hipErrorInvalidConfiguration;
}

// persistent variant: the configurations small graphs get
#define FLAME_PERSIST_CFGS(X) X(256, 2, 1) X(256, 3, 1) X(512, 2, 1) X(512, 3, 1) X(1024, 2, 1) X(1024, 3, 1)

// ---- the resident tiles' address-sorted poll lists (PersistArgs::poll_*): per tile, its halo vertices and its halo edges in
// ascending GLOBAL id.  Derived from the finished tile arrays whoever made them (host builder, device builder, one-launch
// plan), after the lane order has been applied (it moves edges inside their 64-blocks; a record names a position).  One
// workgroup per tile, bitonic sort of <= 4 096 64-bit keys {global id, local id} in LDS.
constexpr int kPollCap = 4096, kPollThreads = 1024;
// thread t handles elements t, t + 1024, ...: partners i ^ j with j < 64 lie in the same wavefront's 64-block, and a
// wavefront's LDS operations complete in order -- only the steps with j >= 64, and the step in front of one, need the
// workgroup barrier (27 instead of 78 at 4 096 keys)
__device__ __forceinline__ void bitonic_sort_u64(unsigned long long* key, int m, int tid) {
  for (int k = 2; k <= m; k <<= 1)
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int i = tid; i < m; i += kPollThreads) {
        const int l = i ^ j;
        if (l > i) {
          const unsigned long long a = key[i], b = key[l];
          const bool up = (i & k) == 0;
          if ((a > b) == up) { key[i] = b; key[l] = a; }
        }
      }
      // (the workgroup barrier behind every step that paired elements of different wavefronts, j >= 64, and behind the last
      // step of a merge when the next merge starts with such a step, j = k >= 64)
      if (j >= 64 || (j == 1 && k >= 64)) __syncthreads();
      else { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup"); __builtin_amdgcn_wave_barrier(); }
    }
  __syncthreads();
}

// SORT = false: the lists in LOCAL order, unsorted -- a plan that is solved once (a frame of a stream) does not pay for the
// sorts (28 us at 1.2 k vertices against the ~5 us its one solve would gain); every local edge gets a record, the owned
// ones marked invalid (global id ~0), so a lane polls what it holds, as before r05.
template <bool SORT>
__global__ __launch_bounds__(kPollThreads) void k_poll_lists(const TileDesc* __restrict__ tiles, const int32_t* __restrict__ t_vmap,
                                                             const int32_t* __restrict__ t_emap, const uint2* __restrict__ t_eij,
                                                             uint2* __restrict__ poll_v, uint2* __restrict__ poll_e,
                                                             int32_t* __restrict__ poll_ne, int32_t* __restrict__ need_v,
                                                             int32_t* __restrict__ need_e) {
  __shared__ unsigned long long key[kPollCap];
  __shared__ int s_n;
  const TileDesc& D = tiles[blockIdx.x];
  const int tid = threadIdx.x;
  if (D.n_ext <= 0) { if (tid == 0) poll_ne[blockIdx.x] = 0; return; }
  // halo vertices
  const int nhv = D.n_ext - D.n_own;
  if (nhv > kPollCap || D.e_loc > kPollCap) { if (tid == 0) poll_ne[blockIdx.x] = -1; return; }  // (never with the resident configurations)
  if (!SORT) {
    for (int i = tid; i < nhv; i += kPollThreads) {
      const int lv = D.n_own + i;
      poll_v[D.vmap_off + i] = make_uint2((uint32_t)t_vmap[D.vmap_off + lv], (uint32_t)lv | (lv < D.n_upd ? 0x80000000u : 0u));
    }
    for (int le = tid; le < D.e_loc; le += kPollThreads) {
      const int32_t eid = t_emap[D.emap_off + le];
      const bool owned = (uint32_t)(eid - D.estart) < (uint32_t)D.e_own;
      const uint32_t sl = t_eij[D.erec_off + le].y, ss = sl & 0xffffu, sd = sl >> 16;
      const uint32_t slot = ss != 0xffffu ? ss : (sd != 0xffffu ? sd : (uint32_t)(D.nslots + (le & 63)));
      poll_e[D.emap_off + le] = make_uint2(owned ? 0xffffffffu : (uint32_t)eid, slot);
    }
    if (tid == 0) poll_ne[blockIdx.x] = D.e_loc;
    return;
  }
  int m = 64;
  while (m < nhv) m <<= 1;
  for (int i = tid; i < m; i += kPollThreads) {
    unsigned long long kk = ~0ull;
    if (i < nhv) {
      const int lv = D.n_own + i;
      kk = ((unsigned long long)(uint32_t)t_vmap[D.vmap_off + lv] << 32) | (uint32_t)lv | (lv < D.n_upd ? 0x80000000u : 0u);
    }
    key[i] = kk;
  }
  __syncthreads();
  bitonic_sort_u64(key, m, tid);
  for (int i = tid; i < nhv; i += kPollThreads) {
    poll_v[D.vmap_off + i] = make_uint2((uint32_t)(key[i] >> 32), (uint32_t)key[i]);
    if (need_v) atomicOr(&need_v[(uint32_t)(key[i] >> 32)], ((uint32_t)key[i] >> 31) ? 3 : 1);  // (zeroed by the caller)
  }
  __syncthreads();
  // halo edges (the owned ones are the internal ids [estart, estart + e_own)); the record names the incidence slot the dual
  // is staged in: the source's, else the target's, else (no endpoint is ever updated) the lane's trash slot
  if (tid == 0) s_n = 0;
  __syncthreads();
  for (int le = tid; le < D.e_loc; le += kPollThreads) {
    const int32_t eid = t_emap[D.emap_off + le];
    if ((uint32_t)(eid - D.estart) < (uint32_t)D.e_own) continue;
    const uint32_t sl = t_eij[D.erec_off + le].y, ss = sl & 0xffffu, sd = sl >> 16;
    const uint32_t slot = ss != 0xffffu ? ss : (sd != 0xffffu ? sd : (uint32_t)(D.nslots + (le & 63)));
    const int p = atomicAdd(&s_n, 1);
    key[p] = ((unsigned long long)(uint32_t)eid << 32) | slot;
  }
  __syncthreads();
  const int nhe = s_n;
  m = 64;
  while (m < nhe) m <<= 1;
  for (int i = nhe + tid; i < m; i += kPollThreads) key[i] = ~0ull;
  __syncthreads();
  bitonic_sort_u64(key, m, tid);
  for (int i = tid; i < nhe; i += kPollThreads) {
    poll_e[D.emap_off + i] = make_uint2((uint32_t)(key[i] >> 32), (uint32_t)key[i]);
    if (need_e) need_e[(uint32_t)(key[i] >> 32)] = 1;
  }
  if (tid == 0) poll_ne[blockIdx.x] = nhe;
}

hipError_t launch_poll_lists(hipStream_t s, int32_t ntiles, const TileDesc* tiles, const int32_t* t_vmap, const int32_t* t_emap,
                             const uint2* t_eij, uint2* poll_v, uint2* poll_e, int32_t* poll_ne, bool sorted, int32_t* need_v,
                             int32_t* need_e) {
  if (ntiles <= 0) return hipSuccess;
  if (sorted) hipLaunchKernelGGL(k_poll_lists<true>, dim3(ntiles), dim3(kPollThreads), 0, s, tiles, t_vmap, t_emap, t_eij, poll_v, poll_e, poll_ne, need_v, need_e);
  else hipLaunchKernelGGL(k_poll_lists<false>, dim3(ntiles), dim3(kPollThreads), 0, s, tiles, t_vmap, t_emap, t_eij, poll_v, poll_e, poll_ne, (int32_t*)nullptr, (int32_t*)nullptr);
  return hipGetLastError();
}

bool tile_torn_check_build() { return FLAME_TORN_CHECK != 0; }
bool tile_stall_hook_build() { return FLAME_PERSIST_STALL_HOOK != 0; }

bool tile_persist_exists(int nt, int ept, int vpt) {
#define X(N, Ep, Vp) if (nt == N && ept == Ep && vpt == Vp) return true;
  FLAME_PERSIST_CFGS(X)
#undef X
  return false;
}

template <int NT, int EPT, int VPT, bool S12 = false, bool FAT = false>
hipError_t launch_tile_persist_t(hipStream_t s, size_t lds, const TileArgs& a, const PersistArgs& pa) {
  if (lds > 48 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_tile_persist<NT, EPT, VPT, S12, FAT>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
  }
  // the launches' grid: one workgroup per tile, all of them resident (the caller keeps ntiles <= the number of CUs)
  hipLaunchKernelGGL((k_tile_persist<NT, EPT, VPT, S12, FAT>), dim3(pa.one_xcd ? 8 * a.ntiles : a.ntiles), dim3(NT), lds, s, a.tiles, a.ntiles, a.iters, a.t_vmap,
                     a.t_srow, a.t_eij, a.t_emap, a.t_ew, a.B_src, a.A_src, a.q_src, a.A_dst, a.B_dst, a.q_dst, pa, a.p);
  return hipGetLastError();
}

hipError_t launch_tile_persist(hipStream_t s, int nt, int ept, int vpt, size_t lds_bytes, const TileArgs& a, const PersistBufs& x,
                               int32_t* err_host, int32_t base) {
  if (a.ntiles <= 0 || a.iters <= 0) return hipSuccess;
  PersistArgs pa{};
  pa.err_host = err_host; pa.base = base; pa.prof = x.prof; pa.poll_delay = x.poll_delay;
  pa.timeout_ticks = x.timeout_ticks > 0 ? x.timeout_ticks : 400000;
  pa.one_xcd = x.one_xcd;
  for (int b = 0; b < 2; ++b) { pa.hA[b] = x.hA[b]; pa.hB[b] = x.hB[b]; pa.hq[b] = x.hq[b]; }
  pa.poll_v = x.poll_v; pa.poll_e = x.poll_e; pa.poll_ne = x.poll_ne;
  pa.need_v = (x.need_valid && a.fat) ? x.need_v : nullptr; pa.need_e = (x.need_valid && a.fat) ? x.need_e : nullptr;
  if (a.slot12 || a.fat) {  // the fat variants: 1 024 threads, 2 or 3 edges per thread, either slot layout
#define X(N, Ep, Vp) if (nt == N && ept == Ep && vpt == Vp) return a.slot12 ? launch_tile_persist_t<N, Ep, Vp, true, true>(s, lds_bytes + x.stage_bytes, a, pa) \
                                                                            : launch_tile_persist_t<N, Ep, Vp, false, true>(s, lds_bytes + x.stage_bytes, a, pa);
    FLAME_S12_CFGS(X)
#undef X
    return hipErrorInvalidConfiguration;
  }
#define X(N, Ep, Vp) if (nt == N && ept == Ep && vpt == Vp) return launch_tile_persist_t<N, Ep, Vp>(s, lds_bytes + x.stage_bytes, a, pa);
  FLAME_PERSIST_CFGS(X)
#undef X
  return hipErrorInvalidConfiguration;
}

hipError_t prepare_tile(int nt, int ept, int vpt, size_t lds_bytes, bool slot12) {
  if (slot12) {
#define X(N, Ep, Vp) if (nt == N && ept == Ep && vpt == Vp) return prepare_tile_t<N, Ep, Vp, true>(lds_bytes);
    FLAME_S12_CFGS(X)
#undef X
    return hipErrorInvalidConfiguration;
  }
#define X(N, Ep, Vp) if (nt == N && ept == Ep && vpt == Vp) return prepare_tile_t<N, Ep, Vp>(lds_bytes);
  FLAME_TILE_CFGS(X)
#undef X
  return hipErrorInvalidConfiguration;
}

int costs_num_blocks(int32_t, int32_t) { return kCostBlocks; }

hipError_t launch_costs(hipStream_t s, int32_t V, int32_t E, const int2* eij, const float4* ew,
                        const float4* A, const float4* B, float lambda, double* partials,
                        const uint8_t* emask, const uint8_t* vmask) {
  hipLaunchKernelGGL(k_costs, dim3(kCostBlocks), dim3(256), 0, s, V, E, eij, ew, A, B, lambda, partials, emask, vmask);
  return hipGetLastError();
}

hipError_t launch_triangles(hipStream_t s, int32_t V, int32_t T, const float2* pos,
                            const float4* A, const int32_t* tris, const int32_t* trow,
                            const int32_t* tinc, TriParamsDev tp, float4* tri_normals,
                            uint8_t* tri_valid, float4* vtx_normals, const FrameOut* fo) {
  if (T > 0) {
    hipLaunchKernelGGL(k_tri, dim3((T + 255) / 256), dim3(256), 0, s, T, pos, A, tris, tp, tri_normals, tri_valid,
                       fo ? fo->tri_valid : nullptr);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
  }
  if (V > 0) {
    hipLaunchKernelGGL(k_vtx_normals, dim3((V + 255) / 256), dim3(256), 0, s, V, trow, tinc, tri_normals, vtx_normals,
                       fo ? fo->v_i2o : nullptr, A, fo ? fo->x : nullptr, fo ? fo->normals : nullptr);
    return hipGetLastError();
  }
  return hipSuccess;
}

int raster_num_blocks(int32_t width, int32_t height) { return (int)(((int64_t)width * height + 255) / 256); }

}  // namespace flamehip
