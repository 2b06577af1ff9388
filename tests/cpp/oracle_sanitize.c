/* tests/cpp/oracle_sanitize.c -- every oracle entry point once on a small graph, built with
 * -fsanitize=address,undefined (the reference has no sanitizer build; the oracle is the
 * checker everything else is compared with, so it gets one).  Exit code 0 = clean. */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

#include "../../oracle/nltgv2_oracle.h"

int main(void) {
  enum { NX = 9, NY = 7, V = NX * NY, T = 2 * (NX - 1) * (NY - 1) };
  float pos[2 * V], mu[V], var[V], pred[V];
  int32_t tris[3 * T];
  unsigned s = 12345u;
  for (int y = 0; y < NY; ++y)
    for (int x = 0; x < NX; ++x) {
      const int v = y * NX + x;
      s = s * 1664525u + 1013904223u;
      pos[2 * v] = 40.0f * x + (float)(s >> 28);
      pos[2 * v + 1] = 40.0f * y + (float)((s >> 20) & 7);
      mu[v] = 0.5f + 0.001f * pos[2 * v] + ((v % 5) ? 0.0f : 0.1f);
      var[v] = 1e-4f * (1 + v % 3);
      pred[v] = (v % 2) ? mu[v] : NAN;
    }
  int t = 0;
  for (int y = 0; y + 1 < NY; ++y)
    for (int x = 0; x + 1 < NX; ++x) {
      const int a = y * NX + x, b = a + 1, c = a + NX, d = c + 1;
      tris[3 * t] = a; tris[3 * t + 1] = b; tris[3 * t + 2] = c; ++t;
      tris[3 * t] = b; tris[3 * t + 1] = d; tris[3 * t + 2] = c; ++t;
    }
  int32_t* edges = malloc(sizeof(int32_t) * 2 * 3 * T);
  float *alpha = malloc(sizeof(float) * 3 * T), *beta = malloc(sizeof(float) * 3 * T);
  float z[V], wgt[V], x0[V], scale = 0.f;
  nltgv2_sync_params sp = {1, 1, 1, 0.01f, 0, 0.0f, 0.0f};
  uint8_t keep[V];
  if (nltgv2_feature_gate(V, var, sp.idepth_var_max_graph, keep) != V) return 2;
  const int32_t E = nltgv2_graph_sync(&sp, V, T, pos, mu, var, tris, pred, edges, alpha, beta, z, wgt, x0, &scale);
  float x[V], w1[V], w2[V], xb[V], w1b[V], w2b[V];
  float* q = calloc((size_t)3 * E, sizeof(float));
  for (int v = 0; v < V; ++v) { x[v] = xb[v] = x0[v]; w1[v] = w2[v] = w1b[v] = w2b[v] = 0.f; }
  nltgv2_graph g = {V, E, pos, edges, alpha, beta, z, wgt, x, w1, w2, xb, w1b, w2b, q};
  nltgv2_params p = {0.15f, 1e-3f, 125.0f, 0.25f, 0.0f, 10.0f};
  if (nltgv2_solve(&p, &g, 25)) return 3;
  int32_t* row = malloc(sizeof(int32_t) * (V + 1));
  int32_t* inc = malloc(sizeof(int32_t) * 2 * E);
  nltgv2_build_incidence(&g, row, inc);
  nltgv2_solve_omp(&p, &g, row, inc, 5, 2);
  float scratch[V];
  nltgv2_graph_filter(&g, row, inc, 0, scratch);
  nltgv2_graph_filter(&g, row, inc, 1, scratch);
  double sm, da;
  nltgv2_costs(&p, &g, &sm, &da);
  float* Ku = malloc(sizeof(float) * 3 * E);
  nltgv2_apply_K(&g, x, w1, w2, Ku);
  float kx[V], k1[V], k2[V];
  nltgv2_apply_KT(&g, q, kx, k1, k2);
  nltgv2_scale_state(&g, z, scale);
  nltgv2_tri_params tp = {1, 1.57f, 0.35f, 0.1f, 1, 0.333f, 1, 0.01f, 360, 280};
  const float Kinv[9] = {1.f / 300, 0, -180.f / 300, 0, 1.f / 300, -140.f / 300, 0, 0, 1};
  float tn[3 * T], vn[3 * V];
  uint8_t tv[T];
  nltgv2_triangles(&tp, Kinv, V, T, pos, x, tris, tn, tv, vn);
  float pts[12 * V];
  int32_t faces[3 * T];
  nltgv2_mesh_points(Kinv, V, pos, x, vn, tp.width, tp.height, pts);
  const int32_t nf = nltgv2_mesh_faces(T, tris, tv, faces);
  float* idm = malloc(sizeof(float) * tp.width * tp.height);
  float* dm = malloc(sizeof(float) * tp.width * tp.height);
  float* cl = malloc(sizeof(float) * 3 * tp.width * tp.height);
  nltgv2_idepthmap(tp.width, tp.height, T, pos, x, tris, tv, 1, idm);
  nltgv2_depth_and_cloud(tp.width, tp.height, idm, Kinv, 0.1f, 100.0f, dm, cl);
  printf("E=%d faces=%d smooth=%.6f data=%.6f scale=%.6f\n", E, nf, sm, da, scale);
  free(edges); free(alpha); free(beta); free(q); free(row); free(inc); free(Ku); free(idm); free(dm); free(cl);
  return (E > 0 && isfinite(sm) && isfinite(da)) ? 0 : 4;
}
