// include/flame/optimizers/nltgv2_l1_graph_regularizer.h -- upstream's regulariser interface
// ([UPSTREAM-RECALL] robustrobotics/flame src/flame/optimizers/nltgv2_l1_graph_regularizer.h:
// Params{data_factor, step_x, step_q, theta}, step(params, &graph), smoothnessCost, dataCost;
// the parameters are pinned by reference src/flame_offline_tum.cc:242-245) over the HIP library.
//
// Upstream's Graph is a Boost adjacency_list walked on the CPU; here Graph is a handle to the
// same graph resident in MI355X HBM (include/flame_hip.h).  Header-only, links libflame_hip.so.
#pragma once
#include <cstdint>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../flame_hip.h"
#include "../params.h"

namespace flame {
namespace optimizers {
namespace nltgv2_l1_graph_regularizer {

inline flame_hip_params toC(const Params& p) {
  return flame_hip_params{p.data_factor, p.step_x, p.step_q, p.theta, p.x_min, p.x_max};
}

// Device-resident Delaunay vertex graph.  Not copyable; one Graph is not thread-safe.
class Graph {
 public:
  Graph() = default;
  Graph(const Graph&) = delete;
  Graph& operator=(const Graph&) = delete;
  ~Graph() { reset(); }

  void reset() {
    if (g_) flame_hip_graph_destroy(g_);
    g_ = nullptr;
    device_ = -1;
    d_sign_ = 1;
    V_ = E_ = T_ = 0;
    resident_ = false;
  }
  // Returns a flame_hip error code (0 = ok).  pos 2V, edges 2E (i -> j), tris 3T or nullptr.
  // The handle is created once and then RESIZED frame after frame: its stream, events, device
  // buffers, host plan buffers and tile cost-density grid survive (flame_hip_graph_resize); it is
  // only re-created when the device changes.
  int build(int device, int32_t V, int32_t E, int32_t T, const float* pos, const int32_t* edges,
            const float* alpha, const float* beta, const float* z, const float* wgt,
            const float* x0, const int32_t* tris) {
    int rc = acquire(device);
    if (rc) return rc;
    if ((rc = flame_hip_graph_resize(g_, V, E, T))) return rc;
    V_ = E_ = T_ = 0;
    resident_ = false;
    if ((rc = flame_hip_graph_upload(g_, pos, edges, alpha, beta, z, wgt, x0, tris))) return rc;
    V_ = V; E_ = E; T_ = T;
    resident_ = true;
    return 0;
  }
  // Graph sync (row a7) in the library: features + triangulation in, graph resident on the GPU.
  int sync(int device, const flame_hip_sync_params& sp, int32_t V, int32_t T, const float* pos,
           const float* idepth_mu, const float* idepth_var, const int32_t* tris,
           const float* prediction, float* scale, int d_sign = 1) {
    int rc = acquire(device);
    if (rc) return rc;
    if (d_sign != d_sign_) {
      if ((rc = flame_hip_set_option(g_, "d_sign", d_sign))) return rc;
      d_sign_ = d_sign;
    }
    V_ = E_ = T_ = 0;
    resident_ = false;
    if ((rc = flame_hip_graph_sync(g_, &sp, V, T, pos, idepth_mu, idepth_var, tris, prediction, scale)))
      return rc;
    int64_t e = 0;
    if ((rc = flame_hip_get_info(g_, "E", &e))) return rc;
    V_ = V; E_ = static_cast<int32_t>(e); T_ = T;
    resident_ = true;
    return 0;
  }
  // Row f3's first leg in the library (flame_hip_delaunay): the Delaunay triangulation of the features on the GPU of
  // `device`.  tris (3 ints per triangle, e.g. a vector<cv::Vec3i>'s storage) needs room for 2V triangles.
  int triangulate(int device, int32_t V, const float* pos, int32_t tri_cap, int32_t* tris, int32_t* T) {
    const int rc = acquire(device);
    return rc ? rc : flame_hip_delaunay(g_, V, pos, tri_cap, tris, T);
  }
  // ... in KEEP mode (tri_cap = 0, tris = NULL above): the list stays on the device for the sync() that follows, its host
  // copy travels meanwhile; this hands it out (and waits for it) -- flame::Flame calls it while the GPU iterates
  int triangleList(int32_t tri_cap, int32_t* tris) { return g_ ? flame_hip_delaunay_list(g_, tri_cap, tris) : FLAME_HIP_ERR_STATE; }
  // a graph is resident (the last build / sync succeeded)
  bool valid() const { return g_ != nullptr && resident_; }
  int32_t numVertices() const { return V_; }
  int32_t numEdges() const { return E_; }
  int32_t numTriangles() const { return T_; }
  flame_hip_graph* handle() const { return g_; }

 private:
  int acquire(int device) {
    if (g_ && device_ == device) return 0;
    reset();
    const int rc = flame_hip_graph_create(&g_, device, 0, 0, 0);
    if (rc) { g_ = nullptr; return rc; }
    // A Graph is re-built every frame, so the plan is paid by ONE solve.  Measured per frame (graph
    // sync + 200 iterations + results, tools/exp/small_frame_sweep.py): up to ~900 vertices ONE
    // isolated tile wins (trivial host plan: 0.44 ms at 770 vertices against 0.49 on halo tiles, a tie
    // at 910); above, a few dozen halo tiles planned on the GPU do (1 200 vertices: 0.51 vs 0.64 ms),
    // and with halo depth 5 rather than the 8 a resident graph of <= 64 tiles gets (1.2-2 k vertices:
    // 0.51-0.58 vs 0.56-0.66 ms).
    // (r03 / r04, resident tiles -- one launch per solve, the library's default -- moved the crossover down again:
    // 700 vertices 0.428 ms on one tile vs 0.412 on 14 resident tiles, 850: 0.479 vs 0.400, 500: 0.354 vs 0.370,
    // tools/exp/persist_small.py; r04: the automatic 32-vertex tiles do as well as any special tile size on frames of
    // 0.8-3 k vertices, tools/exp/facade_small_sweep.py)
    (void)flame_hip_set_option(g_, "tile_single_max", 640);
    (void)flame_hip_set_option(g_, "stream_depth", 5);
    device_ = device;
    return 0;
  }
  flame_hip_graph* g_ = nullptr;
  int device_ = -1;
  int d_sign_ = 1;
  int32_t V_ = 0, E_ = 0, T_ = 0;
  bool resident_ = false;
};

// num_iters x (dualStep; primalStep; extraGradientStep).  Returns 0 or a flame_hip error code.
// wait = false leaves the iterations running on the handle's stream (the next library call orders
// itself behind them).
inline int step(const Params& params, Graph* graph, int num_iters = 1, bool wait = true) {
  if (!graph || !graph->valid()) return FLAME_HIP_ERR_STATE;
  const flame_hip_params p = toC(params);
  int rc = flame_hip_solve(graph->handle(), &p, num_iters, nullptr);
  return (rc || !wait) ? rc : flame_hip_sync(graph->handle());
}

inline float smoothnessCost(const Params& params, const Graph& graph) {
  double s = 0.0, d = 0.0;
  const flame_hip_params p = toC(params);
  if (!graph.valid() || flame_hip_costs(graph.handle(), &p, &s, &d)) return -1.0f;
  return static_cast<float>(s);
}

inline float dataCost(const Params& params, const Graph& graph) {
  double s = 0.0, d = 0.0;
  const flame_hip_params p = toC(params);
  if (!graph.valid() || flame_hip_costs(graph.handle(), &p, &s, &d)) return -1.0f;
  return static_cast<float>(d);
}

// ---- the same regulariser on ONE graph cut over several GPUs (one process per GPU; include/flame_hip.h
// flame_hip_comm_* / flame_hip_part_*: RCB subdomains, halo records by ncclSend / ncclRecv inside the library).
// Upstream has no counterpart (a single CPU process); the interface mirrors Graph / step() / the costs above.
//   rank 0: Communicator::uniqueId(&id); hand the 128 bytes to every rank; every rank:
//   Communicator comm; comm.init(device, rank, world, id);
//   PartitionedGraph g; g.build(comm, 1, 16, V, E, pos, edges, alpha, beta, z, wgt, nullptr);   // the WHOLE graph
//   step(params, &g, 500); g.gather(x.data(), nullptr, nullptr, nullptr);
class Communicator {
 public:
  Communicator() = default;
  Communicator(const Communicator&) = delete;
  Communicator& operator=(const Communicator&) = delete;
  ~Communicator() { reset(); }
  static int uniqueId(char id[FLAME_HIP_COMM_ID_BYTES]) { return flame_hip_comm_get_unique_id(id); }
  int init(int device, int rank, int world, const char id[FLAME_HIP_COMM_ID_BYTES]) {
    reset();
    return flame_hip_comm_create(&c_, device, rank, world, id);
  }
  // a communicator WITHOUT RCCL: its graphs exchange through the peer transport only (PartitionedGraph::peerBlob / peerConnect)
  int initLocal(int device, int rank, int world) {
    reset();
    return flame_hip_comm_create_local(&c_, device, rank, world);
  }
  void reset() { if (c_) flame_hip_comm_destroy(c_); c_ = nullptr; }
  flame_hip_comm* handle() const { return c_; }

 private:
  flame_hip_comm* c_ = nullptr;
};

class PartitionedGraph {
 public:
  PartitionedGraph() = default;
  PartitionedGraph(const PartitionedGraph&) = delete;
  PartitionedGraph& operator=(const PartitionedGraph&) = delete;
  ~PartitionedGraph() { reset(); }
  void reset() { if (p_) flame_hip_part_destroy(p_); p_ = nullptr; V_ = E_ = 0; }
  // every rank passes the WHOLE graph (caller's order); halo_depth iterations run between two exchanges
  int build(const Communicator& comm, int parts_per_rank, int halo_depth, int32_t V, int32_t E, const float* pos,
            const int32_t* edges, const float* alpha, const float* beta, const float* z, const float* wgt, const float* x0) {
    reset();
    const int rc = flame_hip_part_create(&p_, comm.handle(), 0, 0, parts_per_rank, halo_depth, V, E, pos, edges, alpha, beta, z,
                                         wgt, x0);
    if (!rc) { V_ = V; E_ = E; }
    return rc;
  }
  bool valid() const { return p_ != nullptr; }
  int32_t numVertices() const { return V_; }
  int32_t numEdges() const { return E_; }
  // the whole solution on every rank (any pointer may be null; q is 3E interleaved)
  int gather(float* x, float* w1, float* w2, float* q) { return p_ ? flame_hip_part_gather(p_, x, w1, w2, q) : FLAME_HIP_ERR_STATE; }
  // Transport of the halo records: RCCL send / receive (default), or the peer transport -- one kernel writes them straight into
  // the receiving parts' inboxes (same GPU, hipIpc, xGMI peer access), one kernel waits for the messages' flags and unpacks.
  // Collective over RCCL when the communicator has one; between processes without RCCL: peerBlob() of every rank, gathered in
  // rank order by the application, to peerConnect(), then setPeerTransport(true).
  int setPeerTransport(bool on) { return p_ ? flame_hip_part_set_option(p_, "transport", on ? 1 : 0) : FLAME_HIP_ERR_STATE; }
  int peerBlob(char blob[FLAME_HIP_PEER_BLOB_BYTES]) { return p_ ? flame_hip_part_peer_blob(p_, blob) : FLAME_HIP_ERR_STATE; }
  int peerConnect(const char* blobs) { return p_ ? flame_hip_part_peer_connect(p_, blobs) : FLAME_HIP_ERR_STATE; }
  flame_hip_part* handle() const { return p_; }

 private:
  flame_hip_part* p_ = nullptr;
  int32_t V_ = 0, E_ = 0;
};

inline int step(const Params& params, PartitionedGraph* graph, int num_iters = 1, bool wait = true) {
  if (!graph || !graph->valid()) return FLAME_HIP_ERR_STATE;
  const flame_hip_params p = toC(params);
  int rc = flame_hip_part_solve(graph->handle(), &p, num_iters);
  return (rc || !wait) ? rc : flame_hip_part_sync(graph->handle());
}

// whole-graph costs (owned sums of every part + one ncclAllReduce); both through one call to save a reduction
inline int costs(const Params& params, const PartitionedGraph& graph, double* smooth, double* data) {
  if (!graph.valid()) return FLAME_HIP_ERR_STATE;
  const flame_hip_params p = toC(params);
  return flame_hip_part_costs(graph.handle(), &p, smooth, data);
}

}  // namespace nltgv2_l1_graph_regularizer
}  // namespace optimizers
}  // namespace flame
