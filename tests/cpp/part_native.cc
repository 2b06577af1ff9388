// tests/cpp/part_native.cc -- partition mode from C++, no Python in the data path: the same random
// graph solved (i) on one handle and (ii) cut into `parts` subdomains on ONE rank of an RCCL communicator of world size 1
// (every halo record = ncclSend / ncclRecv of the rank with itself), through include/flame/optimizers/
// nltgv2_l1_graph_regularizer.h.  Every bit of x, w1, w2, q and the costs must agree.  argv: device parts depth iters V [peer transport 0 / 1].
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>

#include "flame/optimizers/nltgv2_l1_graph_regularizer.h"
#include "flame/utils/delaunay.h"

namespace reg = flame::optimizers::nltgv2_l1_graph_regularizer;

int main(int argc, char** argv) {
  const int device = argc > 1 ? std::atoi(argv[1]) : 0, parts = argc > 2 ? std::atoi(argv[2]) : 2;
  const int depth = argc > 3 ? std::atoi(argv[3]) : 8, iters = argc > 4 ? std::atoi(argv[4]) : 60;
  const int V = argc > 5 ? std::atoi(argv[5]) : 8000;
  const bool peer = argc > 6 && std::atoi(argv[6]) != 0;  // the r06 peer transport instead of RCCL's send / receive
  std::mt19937 rng(7);
  std::uniform_real_distribution<float> ux(0.f, 640.f), uy(0.f, 480.f);
  std::normal_distribution<float> noise(0.f, 0.02f);
  std::vector<flame::Point2f> pts(static_cast<size_t>(V));
  for (auto& p : pts) p = flame::Point2f(ux(rng), uy(rng));
  std::vector<flame::Triangle> tris;
  if (!flame::utils::delaunay(pts, &tris)) return 2;
  // unique undirected edges i < j of the triangulation
  std::vector<std::pair<int32_t, int32_t> > es;
  for (const auto& t : tris)
    for (int k = 0; k < 3; ++k) {
      const int32_t a = t[k], b = t[(k + 1) % 3];
      es.push_back(a < b ? std::make_pair(a, b) : std::make_pair(b, a));
    }
  std::sort(es.begin(), es.end());
  es.erase(std::unique(es.begin(), es.end()), es.end());
  const int32_t E = static_cast<int32_t>(es.size());
  std::vector<float> pos(2 * static_cast<size_t>(V)), z(static_cast<size_t>(V)), wgt(static_cast<size_t>(V), 1.0f), alpha(static_cast<size_t>(E)), beta;
  std::vector<int32_t> edges(2 * static_cast<size_t>(E));
  for (int v = 0; v < V; ++v) {
    pos[2 * v] = pts[v].x; pos[2 * v + 1] = pts[v].y;
    z[v] = std::max(0.01f, 0.5f + 0.001f * pts[v].x - 0.0005f * pts[v].y + (pts[v].x > 320.f ? 0.3f : 0.f) + noise(rng));
  }
  for (int32_t e = 0; e < E; ++e) {
    edges[2 * e] = es[e].first; edges[2 * e + 1] = es[e].second;
    const float dx = pos[2 * es[e].first] - pos[2 * es[e].second], dy = pos[2 * es[e].first + 1] - pos[2 * es[e].second + 1];
    alpha[e] = 1.0f / std::sqrt(dx * dx + dy * dy);
  }
  beta = alpha;
  reg::Params prm;  // cfg/flame_offline_tum.yaml:93-96 defaults
  // (i) one handle
  reg::Graph g;
  int rc = g.build(device, V, E, 0, pos.data(), edges.data(), alpha.data(), beta.data(), z.data(), wgt.data(), nullptr, nullptr);
  if (rc) { std::fprintf(stderr, "build: %s\n", flame_hip_strerror(rc)); return 3; }
  if ((rc = reg::step(prm, &g, iters))) { std::fprintf(stderr, "step: %s\n", flame_hip_strerror(rc)); return 3; }
  std::vector<float> x(V), w1(V), w2(V), q(3 * static_cast<size_t>(E)), px(V), p1(V), p2(V), pq(3 * static_cast<size_t>(E));
  if ((rc = flame_hip_download(g.handle(), x.data(), w1.data(), w2.data(), q.data()))) return 3;
  const float s1 = reg::smoothnessCost(prm, g), d1 = reg::dataCost(prm, g);
  // (ii) `parts` subdomains on rank 0 of a world of 1
  char id[FLAME_HIP_COMM_ID_BYTES];
  if ((rc = reg::Communicator::uniqueId(id))) { std::fprintf(stderr, "unique id: %s\n", flame_hip_strerror(rc)); return 4; }
  reg::Communicator comm;
  if ((rc = comm.init(device, 0, 1, id))) { std::fprintf(stderr, "comm: %s\n", flame_hip_strerror(rc)); return 4; }
  reg::PartitionedGraph pg;
  if ((rc = pg.build(comm, parts, depth, V, E, pos.data(), edges.data(), alpha.data(), beta.data(), z.data(), wgt.data(), nullptr))) {
    std::fprintf(stderr, "part build: %s\n", flame_hip_strerror(rc));
    return 4;
  }
  if (peer && (rc = pg.setPeerTransport(true))) { std::fprintf(stderr, "peer transport: %s\n", flame_hip_strerror(rc)); return 4; }
  if ((rc = reg::step(prm, &pg, iters / 3)) || (rc = reg::step(prm, &pg, iters - iters / 3))) {
    std::fprintf(stderr, "part step: %s\n", flame_hip_strerror(rc));
    return 4;
  }
  if ((rc = pg.gather(px.data(), p1.data(), p2.data(), pq.data()))) return 4;
  double s2 = 0, d2 = 0;
  if ((rc = reg::costs(prm, pg, &s2, &d2))) return 4;
  int64_t ex = 0, ops = 0;
  flame_hip_part_info(pg.handle(), "exchanges", 0, &ex);
  flame_hip_part_info(pg.handle(), "p2p_ops", 0, &ops);
  const bool same = !std::memcmp(x.data(), px.data(), sizeof(float) * V) && !std::memcmp(w1.data(), p1.data(), sizeof(float) * V) &&
                    !std::memcmp(w2.data(), p2.data(), sizeof(float) * V) && !std::memcmp(q.data(), pq.data(), sizeof(float) * 3 * E);
  const bool costs_ok = std::fabs(s2 - s1) <= 1e-5 * s1 && std::fabs(d2 - d1) <= 1e-5 * d1;
  std::printf("V %d E %d parts %d depth %d iters %d exchanges %lld p2p_ops %lld bit_exact %d costs_ok %d (%.6f %.6f | %.6f %.6f)\n", V, E,
              parts, depth, iters, static_cast<long long>(ex), static_cast<long long>(ops), same ? 1 : 0, costs_ok ? 1 : 0, s1, d1, s2, d2);
  return (same && costs_ok && ex > 0) ? 0 : 5;
}
