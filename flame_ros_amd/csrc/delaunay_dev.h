// flame_ros_amd/csrc/delaunay_dev.h -- Delaunay triangulation of a frame's features on the GPU (SURVEY.md 8 row f3's
// first leg: "Delaunay -> edge list"; reference evidence: stat key `triangulate`, msg/FlameStats.msg:44).  See
// delaunay_dev.hip for the algorithm; include/flame_hip.h (flame_hip_delaunay) for the contract.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace flamehip {

struct DelaunayScratch {
  char* dev = nullptr;       // one device arena, grown geometrically, reused frame after frame
  size_t dev_cap = 0;
  char* pin = nullptr;       // page-locked host arena: positions in, flags + triangles out
  size_t pin_cap = 0;
  float last_ms = 0.f;       // host time of the last call (copies included)
  int32_t last_hull = 0;     // boundary vertices of the last triangulation
  int32_t last_live = 0;     // points that are not later copies of another point
  // the list of the last successful call, still in the page-locked arena (flame_hip_graph_sync reads it from there when the
  // caller passes no triangle pointer: an asynchronous DMA instead of a staged copy from pageable memory)
  int32_t last_V = -1, last_T = -1;
  const int32_t* last_list = nullptr;
  // r05 "keep" mode (tri_cap = 0, tris_out = NULL): the list STAYS on the device -- flame_hip_graph_sync reads it there --
  // and its host copy travels asynchronously (behind `ev_list`) while the frame goes on; host_list() waits for it
  const int32_t* last_dev = nullptr;
  hipEvent_t ev_list = nullptr, ev_done = nullptr;
  hipStream_t s_list = nullptr;  // the list's copy-out: a stream of its own, beside the next inputs' copy-in
  bool list_pending = false;
  const int32_t* host_list();  // the page-locked copy, complete (NULL when there is none)
  void release();
};

// Triangulates V points (host array of {u, v} floats) on stream s of the current device.  tris_out (host) receives
// *T_out <= tri_cap counter-clockwise triangles (orient = (b - a) x (c - a) > 0 in the image frame's coordinates), each
// starting at its smallest vertex, ordered by that vertex.  tri_cap = 0 and tris_out = NULL: keep mode (see DelaunayScratch).
// Returns 0, or a FLAME_HIP_ERR_* code.
int delaunay_device(hipStream_t s, DelaunayScratch* sc, int32_t V, const float* pos, int32_t tri_cap, int32_t* tris_out,
                    int32_t* T_out);

}  // namespace flamehip
