// tests/cpp/pin_standin_dump.cc -- stand-in for tools/pin_upstream/dump_upstream.cc on a machine
// WITHOUT robustrobotics/flame: the same scene reader and the same fldump.h writer, with this
// repository's oracle (oracle/nltgv2_oracle.h, test infrastructure) playing upstream's step().  It
// exists so that the pinning pipeline (dump -> convert_dump.py -> tests/test_upstream_pin.py) is
// exercised end to end; a dump it writes pins nothing.  d_sign = -1 emulates an upstream whose edge
// vector is pos_j - pos_i (the oracle is handed negated positions: exactly -d).
//   pin_standin_dump scene.txt out.fldump d_sign x_max iters...
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#include "../../oracle/nltgv2_oracle.h"
#include "../../tools/pin_upstream/fldump.h"

int main(int argc, char** argv) {
  if (argc < 6) return 2;
  FILE* f = std::fopen(argv[1], "r");
  if (!f) return 3;
  int V = 0, E = 0;
  if (std::fscanf(f, "%d %d", &V, &E) != 2) return 3;
  std::vector<float> pos(2 * V), z(V), wgt(V), x0(V), alpha(E), beta(E);
  std::vector<int32_t> edges(2 * E);
  for (int v = 0; v < V; ++v)
    if (std::fscanf(f, "%f %f %f %f %f", &pos[2 * v], &pos[2 * v + 1], &z[v], &wgt[v], &x0[v]) != 5) return 3;
  for (int e = 0; e < E; ++e)
    if (std::fscanf(f, "%d %d %f %f", &edges[2 * e], &edges[2 * e + 1], &alpha[e], &beta[e]) != 4) return 3;
  std::fclose(f);
  const int d_sign = std::atoi(argv[3]);
  nltgv2_params params = {0.15f, 0.001f, 125.0f, 0.25f, 0.0f, static_cast<float>(std::atof(argv[4]))};
  std::vector<float> spos(pos);
  if (d_sign < 0) for (size_t k = 0; k < spos.size(); ++k) spos[k] = -spos[k];
  std::vector<float> x(x0), w1(V, 0.f), w2(V, 0.f), xb(x0), w1b(V, 0.f), w2b(V, 0.f), q(3 * static_cast<size_t>(E), 0.f);
  nltgv2_graph g = {V, E, spos.data(), edges.data(), alpha.data(), beta.data(), z.data(), wgt.data(),
                    x.data(), w1.data(), w2.data(), xb.data(), w1b.data(), w2b.data(), q.data()};
  fldump::Writer w(argv[2]);
  if (!w.ok()) return 4;
  w.floats("pos", pos, V, 2); w.ints("edges", edges, E, 2);
  w.floats("alpha", alpha, E); w.floats("beta", beta, E);
  w.floats("z", z, V); w.floats("wgt", wgt, V); w.floats("x0", x0, V);
  w.floats("params", {params.data_factor, params.step_x, params.step_q, params.theta, params.x_min, params.x_max}, 6);
  std::vector<int32_t> iters;
  for (int a = 5; a < argc; ++a) iters.push_back(std::atoi(argv[a]));
  w.ints("iters", iters, static_cast<uint32_t>(iters.size()));
  int done = 0;
  for (size_t k = 0; k < iters.size(); ++k) {
    if (nltgv2_solve(&params, &g, iters[k] - done)) return 5;
    done = iters[k];
    const std::string tag = "_after_" + std::to_string(iters[k]);
    w.floats("x" + tag, x, V); w.floats("w1" + tag, w1, V); w.floats("w2" + tag, w2, V); w.floats("q" + tag, q, E, 3);
  }
  return 0;
}
