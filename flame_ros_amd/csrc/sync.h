// flame_ros_amd/csrc/sync.h -- graph sync: from the tracked features and
// their Delaunay triangulation to the regulariser's inputs.  Upstream does this inside
// Flame::update() (stat key sync_graph, reference msg/FlameStats.msg:43; parameters reference
// src/flame_offline_tum.cc:234-249, cfg/flame_offline_tum.yaml:87-92).  The precise rule is
// stated at oracle/nltgv2_oracle.c nltgv2_graph_sync; this is the product's implementation.
#pragma once
#include <cstdint>
#include <vector>

#include "../../include/flame_hip.h"

namespace flamehip {

struct SyncOut {
  std::vector<int32_t> edges;  // 2E, i < j, lexicographic
  std::vector<float> alpha;    // E (= beta unless a non-default edge_weight_rule / gain is set)
  std::vector<float> beta;     // E, only filled when it differs from alpha
  std::vector<float> z, wgt, x0;
  float scale = 1.0f;
  std::vector<int32_t> scratch_cnt, scratch_hi, scratch_fill;  // persistent capacity
};

// a non-default [UPSTREAM-RECALL] edge-weight switch is set (flame_hip_sync_params tail)
inline bool sync_weights_custom(const flame_hip_sync_params& sp) {
  return sp.edge_weight_rule != 0 || (sp.alpha_gain != 0.0f && sp.alpha_gain != 1.0f) ||
         (sp.beta_gain != 0.0f && sp.beta_gain != 1.0f);
}

// Returns 0 or FLAME_HIP_ERR_ARG (bad triangle index).
int graph_sync_host(const flame_hip_sync_params& sp, int32_t V, int32_t T, const float* pos,
                    const float* mu, const float* var, const int32_t* tris, const float* prediction,
                    SyncOut* out);

}  // namespace flamehip
