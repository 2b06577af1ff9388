// tools/pin_upstream/dump_upstream.cc -- run on a machine where robustrobotics/flame is BUILT
// (it is not available where this repository was written: reference CMakeLists.txt:57, README.md:73).
// Builds upstream's regulariser graph from a plain-text scene, calls upstream's own step() N times
// and dumps the graph and the solver state after selected iteration counts.  The dump pins this
// repository's oracle and HIP path to upstream (tests/test_upstream_pin.py).
//
//   g++ -std=c++11 -I<flame>/src -I<eigen> -I<boost> dump_upstream.cc -o dump_upstream
//   ./dump_upstream scene.txt upstream_tum.fldump 1 10 200
//   python tools/pin_upstream/convert_dump.py upstream_tum.fldump tests/golden/upstream_tum.npz
//
// scene.txt (written by `python tools/pin_upstream/make_scene.py`): V E / V lines "u v z wgt x0" /
// E lines "i j alpha beta".  Pass alpha = beta = -1 to let UPSTREAM's graph sync choose the weights
// if you drive it through Flame::update instead (then dump them from the graph as done below).
//
// [UPSTREAM-RECALL] The member names below (VertexData::{pos,x,x_bar,x_prev,w1,w1_bar,w1_prev,w2,
// w2_bar,w2_prev,data_term,data_weight}, EdgeData::{alpha,beta,q1,q2,q3,valid}, Params::{data_factor,
// step_x,step_q,theta,x_min,x_max}, step(params,&graph)) are recalled, not verified: adjust them to
// the checked-out header flame/optimizers/nltgv2_l1_graph_regularizer.h -- nothing else changes.
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "flame/optimizers/nltgv2_l1_graph_regularizer.h"
#include "fldump.h"

namespace reg = flame::optimizers::nltgv2_l1_graph_regularizer;

int main(int argc, char** argv) {
  if (argc < 4) return 2;
  FILE* f = std::fopen(argv[1], "r");
  if (!f) return 3;
  int V = 0, E = 0;
  if (std::fscanf(f, "%d %d", &V, &E) != 2) return 3;
  reg::Graph graph;
  std::vector<reg::Graph::vertex_descriptor> vd(V);
  std::vector<reg::Graph::edge_descriptor> ed(E);
  std::vector<float> pos(2 * V), z(V), wgt(V), x0(V), alpha(E), beta(E);
  std::vector<int32_t> edges(2 * E);
  for (int v = 0; v < V; ++v) {
    if (std::fscanf(f, "%f %f %f %f %f", &pos[2 * v], &pos[2 * v + 1], &z[v], &wgt[v], &x0[v]) != 5) return 3;
    vd[v] = boost::add_vertex(graph);
    auto& d = graph[vd[v]];
    d.pos = cv::Point2f(pos[2 * v], pos[2 * v + 1]);
    d.data_term = z[v]; d.data_weight = wgt[v];
    d.x = d.x_bar = d.x_prev = x0[v];
    d.w1 = d.w1_bar = d.w1_prev = 0.f;
    d.w2 = d.w2_bar = d.w2_prev = 0.f;
  }
  for (int e = 0; e < E; ++e) {
    if (std::fscanf(f, "%d %d %f %f", &edges[2 * e], &edges[2 * e + 1], &alpha[e], &beta[e]) != 4) return 3;
    ed[e] = boost::add_edge(vd[edges[2 * e]], vd[edges[2 * e + 1]], graph).first;
    auto& d = graph[ed[e]];
    d.alpha = alpha[e]; d.beta = beta[e];
    d.q1 = d.q2 = d.q3 = 0.f;
    d.valid = true;
  }
  std::fclose(f);
  reg::Params params;  // upstream defaults unless the YAML of flame_ros overrides them
  params.data_factor = 0.15f; params.step_x = 0.001f; params.step_q = 125.0f; params.theta = 0.25f;

  fldump::Writer w(argv[2]);
  if (!w.ok()) return 4;
  w.floats("pos", pos, V, 2); w.ints("edges", edges, E, 2);
  for (int e = 0; e < E; ++e) { alpha[e] = graph[ed[e]].alpha; beta[e] = graph[ed[e]].beta; }  // as upstream holds them
  w.floats("alpha", alpha, E); w.floats("beta", beta, E);
  w.floats("z", z, V); w.floats("wgt", wgt, V); w.floats("x0", x0, V);
  w.floats("params", {params.data_factor, params.step_x, params.step_q, params.theta, params.x_min, params.x_max}, 6);
  std::vector<int32_t> iters;
  for (int a = 3; a < argc; ++a) iters.push_back(std::atoi(argv[a]));
  w.ints("iters", iters, static_cast<uint32_t>(iters.size()));
  int done = 0;
  for (size_t k = 0; k < iters.size(); ++k) {
    for (; done < iters[k]; ++done) reg::step(params, &graph);
    std::vector<float> x(V), w1(V), w2(V), q(3 * E);
    for (int v = 0; v < V; ++v) { x[v] = graph[vd[v]].x; w1[v] = graph[vd[v]].w1; w2[v] = graph[vd[v]].w2; }
    for (int e = 0; e < E; ++e) { q[3 * e] = graph[ed[e]].q1; q[3 * e + 1] = graph[ed[e]].q2; q[3 * e + 2] = graph[ed[e]].q3; }
    const std::string tag = "_after_" + std::to_string(iters[k]);
    w.floats("x" + tag, x, V); w.floats("w1" + tag, w1, V); w.floats("w2" + tag, w2, V); w.floats("q" + tag, q, E, 3);
  }
  return 0;
}
