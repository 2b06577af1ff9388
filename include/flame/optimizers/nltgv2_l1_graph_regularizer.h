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
    V_ = E_ = T_ = 0;
  }
  // Returns a flame_hip error code (0 = ok).  pos 2V, edges 2E (i -> j), tris 3T or nullptr.
  int build(int device, int32_t V, int32_t E, int32_t T, const float* pos, const int32_t* edges,
            const float* alpha, const float* beta, const float* z, const float* wgt,
            const float* x0, const int32_t* tris) {
    reset();
    int rc = flame_hip_graph_create(&g_, device, V, E, T);
    if (rc) return rc;
    rc = flame_hip_graph_upload(g_, pos, edges, alpha, beta, z, wgt, x0, tris);
    if (rc) { reset(); return rc; }
    V_ = V; E_ = E; T_ = T;
    return 0;
  }
  bool valid() const { return g_ != nullptr; }
  int32_t numVertices() const { return V_; }
  int32_t numEdges() const { return E_; }
  int32_t numTriangles() const { return T_; }
  flame_hip_graph* handle() const { return g_; }

 private:
  flame_hip_graph* g_ = nullptr;
  int32_t V_ = 0, E_ = 0, T_ = 0;
};

// num_iters x (dualStep; primalStep; extraGradientStep).  Returns 0 or a flame_hip error code.
inline int step(const Params& params, Graph* graph, int num_iters = 1) {
  if (!graph || !graph->valid()) return FLAME_HIP_ERR_STATE;
  const flame_hip_params p = toC(params);
  int rc = flame_hip_solve(graph->handle(), &p, num_iters, nullptr);
  return rc ? rc : flame_hip_sync(graph->handle());
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

}  // namespace nltgv2_l1_graph_regularizer
}  // namespace optimizers
}  // namespace flame
