/*
 * oracle/nltgv2_oracle.c -- see nltgv2_oracle.h.  TEST INFRASTRUCTURE ONLY; PARITY UNPINNED
 * (the solver source, robustrobotics/flame, is absent from /root/reference and un-pinned:
 * reference README.md:73, CMakeLists.txt:57).  Build: gcc -O3 -march=x86-64-v3 -ffp-contract=off.
 *
 * Every function cites the SURVEY.md section-8a row it restates and the reference lines that
 * show the quantity crossing the flame::Flame boundary.
 */
#include "nltgv2_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

/* projection of one dual component onto [-1,1]: v / max(1,|v|)   */
static inline float proj_unit(float v) { return v / fmaxf(1.0f, fabsf(v)); }

/* Row a2: internal::dualStep.  step_q crosses the boundary at reference
 * src/flame_offline_tum.cc:244 (cfg/flame_offline_tum.yaml:95). */
void nltgv2_dual_step(const nltgv2_params* p, nltgv2_graph* g) {
  const float sigma = p->step_q;
  for (int32_t e = 0; e < g->E; ++e) {
    const int32_t i = g->edges[2 * e], j = g->edges[2 * e + 1];
    const float dx = g->pos[2 * i] - g->pos[2 * j];
    const float dy = g->pos[2 * i + 1] - g->pos[2 * j + 1];
    float t = g->xb[i] - g->xb[j];
    t = fmaf(-g->w1b[i], dx, t);
    t = fmaf(-g->w2b[i], dy, t);
    const float K1 = g->alpha[e] * t;
    const float K2 = g->beta[e] * (g->w1b[i] - g->w1b[j]);
    const float K3 = g->beta[e] * (g->w2b[i] - g->w2b[j]);
    float* q = g->q + 3 * e;
    q[0] = proj_unit(fmaf(sigma, K1, q[0]));
    q[1] = proj_unit(fmaf(sigma, K2, q[1]));
    q[2] = proj_unit(fmaf(sigma, K3, q[2]));
  }
}

/* Row a3: internal::primalStep incl. proxL1.  step_x and data_factor cross the boundary at
 * reference src/flame_offline_tum.cc:242-243 (yaml :93-94). */
void nltgv2_primal_step(const nltgv2_params* p, nltgv2_graph* g, float* xp, float* w1p,
                        float* w2p) {
  const float tau = p->step_x;
  memcpy(xp, g->x, sizeof(float) * g->V);
  memcpy(w1p, g->w1, sizeof(float) * g->V);
  memcpy(w2p, g->w2, sizeof(float) * g->V);
  /* u <- u - tau K^T q, scattered in ascending edge order */
  for (int32_t e = 0; e < g->E; ++e) {
    const int32_t i = g->edges[2 * e], j = g->edges[2 * e + 1];
    const float dx = g->pos[2 * i] - g->pos[2 * j];
    const float dy = g->pos[2 * i + 1] - g->pos[2 * j + 1];
    const float* q = g->q + 3 * e;
    const float aq = g->alpha[e] * q[0];
    const float b2 = g->beta[e] * q[1];
    const float b3 = g->beta[e] * q[2];
    const float s1 = fmaf(-dx, aq, b2); /* -alpha dx q1 + beta q2 */
    const float s2 = fmaf(-dy, aq, b3);
    g->x[i] = fmaf(-tau, aq, g->x[i]);
    g->w1[i] = fmaf(-tau, s1, g->w1[i]);
    g->w2[i] = fmaf(-tau, s2, g->w2[i]);
    g->x[j] = fmaf(-tau, -aq, g->x[j]);
    g->w1[j] = fmaf(-tau, -b2, g->w1[j]);
    g->w2[j] = fmaf(-tau, -b3, g->w2[j]);
  }
  /* L1 prox toward the data term, then idepth clamp */
  const float tl = tau * p->data_factor;
  for (int32_t v = 0; v < g->V; ++v) {
    const float t = tl * g->wgt[v];
    const float x = g->x[v], z = g->z[v];
    const float r = x - z;
    float xn = (r > t) ? (x - t) : ((r < -t) ? (x + t) : z);
    xn = fminf(fmaxf(xn, p->x_min), p->x_max);
    g->x[v] = xn;
  }
}

/* Row a4: internal::extraGradientStep.  theta crosses at src/flame_offline_tum.cc:245. */
void nltgv2_extragradient_step(const nltgv2_params* p, nltgv2_graph* g, const float* xp,
                               const float* w1p, const float* w2p) {
  const float th = p->theta;
  for (int32_t v = 0; v < g->V; ++v) {
    g->xb[v] = fmaf(th, g->x[v] - xp[v], g->x[v]);
    g->w1b[v] = fmaf(th, g->w1[v] - w1p[v], g->w1[v]);
    g->w2b[v] = fmaf(th, g->w2[v] - w2p[v], g->w2[v]);
  }
}

/* Row a5: step() = dual; primal; extra-gradient (gated by do_nltgv2, reference
 * src/flame_offline_tum.cc:234, and driven from Flame::update, :578). */
void nltgv2_step(const nltgv2_params* p, nltgv2_graph* g, float* scratch) {
  float* xp = scratch;
  float* w1p = scratch + g->V;
  float* w2p = scratch + 2 * (size_t)g->V;
  nltgv2_dual_step(p, g);
  nltgv2_primal_step(p, g, xp, w1p, w2p);
  nltgv2_extragradient_step(p, g, xp, w1p, w2p);
}

int nltgv2_solve(const nltgv2_params* p, nltgv2_graph* g, int num_iters) {
  float* scratch = (float*)malloc(sizeof(float) * 3 * (size_t)(g->V > 0 ? g->V : 1));
  if (!scratch) return -1;
  for (int it = 0; it < num_iters; ++it) nltgv2_step(p, g, scratch);
  free(scratch);
  return 0;
}

/* Row a6: smoothnessCost / dataCost behind the stat keys nltgv2_total_smoothness_cost and
 * nltgv2_total_data_cost (reference src/utils.cc:131-136, msg/FlameStats.msg:22-25). */
void nltgv2_costs(const nltgv2_params* p, const nltgv2_graph* g, double* smooth, double* data) {
  double s = 0.0, d = 0.0;
  for (int32_t e = 0; e < g->E; ++e) {
    const int32_t i = g->edges[2 * e], j = g->edges[2 * e + 1];
    const float dx = g->pos[2 * i] - g->pos[2 * j];
    const float dy = g->pos[2 * i + 1] - g->pos[2 * j + 1];
    float t = g->x[i] - g->x[j];
    t = fmaf(-g->w1[i], dx, t);
    t = fmaf(-g->w2[i], dy, t);
    const float c1 = g->alpha[e] * fabsf(t);
    const float c2 = g->beta[e] * fabsf(g->w1[i] - g->w1[j]);
    const float c3 = g->beta[e] * fabsf(g->w2[i] - g->w2[j]);
    s += (double)c1 + (double)c2 + (double)c3;
  }
  for (int32_t v = 0; v < g->V; ++v) {
    const float c = (p->data_factor * g->wgt[v]) * fabsf(g->x[v] - g->z[v]);
    d += (double)c;
  }
  *smooth = s;
  *data = d;
}

void nltgv2_apply_K(const nltgv2_graph* g, const float* x, const float* w1, const float* w2,
                    float* Ku) {
  for (int32_t e = 0; e < g->E; ++e) {
    const int32_t i = g->edges[2 * e], j = g->edges[2 * e + 1];
    const float dx = g->pos[2 * i] - g->pos[2 * j];
    const float dy = g->pos[2 * i + 1] - g->pos[2 * j + 1];
    float t = x[i] - x[j];
    t = fmaf(-w1[i], dx, t);
    t = fmaf(-w2[i], dy, t);
    Ku[3 * e] = g->alpha[e] * t;
    Ku[3 * e + 1] = g->beta[e] * (w1[i] - w1[j]);
    Ku[3 * e + 2] = g->beta[e] * (w2[i] - w2[j]);
  }
}

void nltgv2_apply_KT(const nltgv2_graph* g, const float* q, float* kx, float* kw1, float* kw2) {
  memset(kx, 0, sizeof(float) * g->V);
  memset(kw1, 0, sizeof(float) * g->V);
  memset(kw2, 0, sizeof(float) * g->V);
  for (int32_t e = 0; e < g->E; ++e) {
    const int32_t i = g->edges[2 * e], j = g->edges[2 * e + 1];
    const float dx = g->pos[2 * i] - g->pos[2 * j];
    const float dy = g->pos[2 * i + 1] - g->pos[2 * j + 1];
    const float aq = g->alpha[e] * q[3 * e];
    const float b2 = g->beta[e] * q[3 * e + 1];
    const float b3 = g->beta[e] * q[3 * e + 2];
    kx[i] += aq;
    kw1[i] += fmaf(-dx, aq, b2);
    kw2[i] += fmaf(-dy, aq, b3);
    kx[j] -= aq;
    kw1[j] -= b2;
    kw2[j] -= b3;
  }
}

/* ---- Row a8: per-triangle stage.  Outputs feed getInverseDepthMesh(&vtx,&idepths,&normals,
 * &triangles,&tri_validity,&edges) (reference src/flame_offline_tum.cc:628-635); filter
 * parameters are loaded at :168-192 (yaml :38-53).  The upstream arithmetic is not in the
 * reference tree, so this is the build's own precise statement of those filters. ---- */
static inline void backproject(const float K[9], float u, float v, float x, float P[3]) {
  const float depth = 1.0f / x;
  const float r0 = fmaf(K[0], u, fmaf(K[1], v, K[2]));
  const float r1 = fmaf(K[3], u, fmaf(K[4], v, K[5]));
  const float r2 = fmaf(K[6], u, fmaf(K[7], v, K[8]));
  P[0] = r0 * depth;
  P[1] = r1 * depth;
  P[2] = r2 * depth;
}

static inline float dot3(const float a[3], const float b[3]) {
  return fmaf(a[2], b[2], fmaf(a[1], b[1], a[0] * b[0]));
}

void nltgv2_triangles(const nltgv2_tri_params* tp, const float Kinv[9], int32_t V, int32_t T,
                      const float* pos, const float* x, const int32_t* tris, float* tri_normals,
                      uint8_t* tri_valid, float* vtx_normals) {
  const float cos_thresh = (float)cos((double)tp->oblique_normal_thresh);
  const float max_len = tp->edge_length_thresh * (float)tp->width;
  const float max_len2 = max_len * max_len;
  memset(vtx_normals, 0, sizeof(float) * 3 * (size_t)V);
  for (int32_t t = 0; t < T; ++t) {
    const int32_t a = tris[3 * t], b = tris[3 * t + 1], c = tris[3 * t + 2];
    const float xa = x[a], xb = x[b], xc = x[c];
    float* n = tri_normals + 3 * (size_t)t;
    const int ok = isfinite(xa) && isfinite(xb) && isfinite(xc) && xa > 0.0f && xb > 0.0f &&
                   xc > 0.0f;
    if (!ok) {
      n[0] = n[1] = n[2] = 0.0f;
      tri_valid[t] = 0;
      continue;
    }
    float Pa[3], Pb[3], Pc[3];
    backproject(Kinv, pos[2 * a], pos[2 * a + 1], xa, Pa);
    backproject(Kinv, pos[2 * b], pos[2 * b + 1], xb, Pb);
    backproject(Kinv, pos[2 * c], pos[2 * c + 1], xc, Pc);
    const float e1[3] = {Pb[0] - Pa[0], Pb[1] - Pa[1], Pb[2] - Pa[2]};
    const float e2[3] = {Pc[0] - Pa[0], Pc[1] - Pa[1], Pc[2] - Pa[2]};
    float nn[3] = {fmaf(e1[1], e2[2], -(e1[2] * e2[1])), fmaf(e1[2], e2[0], -(e1[0] * e2[2])),
                   fmaf(e1[0], e2[1], -(e1[1] * e2[0]))};
    const float len = sqrtf(dot3(nn, nn));
    if (len > 0.0f) {
      nn[0] /= len; nn[1] /= len; nn[2] /= len;
    } else {
      nn[0] = 0.0f; nn[1] = 0.0f; nn[2] = -1.0f;
    }
    /* orient toward the camera (origin): n . P_a <= 0 */
    if (dot3(nn, Pa) > 0.0f) { nn[0] = -nn[0]; nn[1] = -nn[1]; nn[2] = -nn[2]; }
    n[0] = nn[0]; n[1] = nn[1]; n[2] = nn[2];

    uint8_t valid = 1;
    const float xmin = fminf(xa, fminf(xb, xc)), xmax = fmaxf(xa, fmaxf(xb, xc));
    if (tp->do_idepth_triangle_filter && xmin < tp->min_triangle_idepth) valid = 0;
    if (tp->do_edge_length_filter) {
      const float ux[3] = {pos[2 * a] - pos[2 * b], pos[2 * b] - pos[2 * c], pos[2 * c] - pos[2 * a]};
      const float uy[3] = {pos[2 * a + 1] - pos[2 * b + 1], pos[2 * b + 1] - pos[2 * c + 1],
                           pos[2 * c + 1] - pos[2 * a + 1]};
      for (int k = 0; k < 3; ++k)
        if (fmaf(ux[k], ux[k], uy[k] * uy[k]) > max_len2) valid = 0;
    }
    if (tp->do_oblique_triangle_filter) {
      /* viewing ray through the centroid */
      float ray[3] = {(Pa[0] + Pb[0]) + Pc[0], (Pa[1] + Pb[1]) + Pc[1], (Pa[2] + Pb[2]) + Pc[2]};
      const float rl = sqrtf(dot3(ray, ray));
      if (rl > 0.0f) {
        ray[0] /= rl; ray[1] /= rl; ray[2] /= rl;
        const float cosang = -dot3(nn, ray); /* angle between normal and the ray back to camera */
        if (cosang < cos_thresh) valid = 0;
      }
      const float diff = xmax - xmin;
      if (diff > tp->oblique_idepth_diff_abs && diff > tp->oblique_idepth_diff_factor * xmax)
        valid = 0;
    }
    tri_valid[t] = valid;
    /* vertex normals: sum of incident triangle normals in ascending triangle order */
    const int32_t vs[3] = {a, b, c};
    for (int k = 0; k < 3; ++k) {
      float* vn = vtx_normals + 3 * (size_t)vs[k];
      vn[0] += nn[0]; vn[1] += nn[1]; vn[2] += nn[2];
    }
  }
  for (int32_t v = 0; v < V; ++v) {
    float* vn = vtx_normals + 3 * (size_t)v;
    const float len = sqrtf(dot3(vn, vn));
    if (len > 0.0f) {
      vn[0] /= len; vn[1] /= len; vn[2] /= len;
    } else {
      vn[0] = 0.0f; vn[1] = 0.0f; vn[2] = -1.0f;
    }
  }
}

/* ---- multi-threaded CPU variant (for the cpu_baseline legs (ii)/(iii) of BASELINE.md only):
 * edge-parallel dual + vertex-parallel CSR primal under OpenMP, the shape the reference's own
 * OpenMP loops have (reference cfg/flame_offline_tum.yaml:70-71: 4 threads, chunk 1024).  Same
 * arithmetic and the same per-vertex summation order as the sequential functions above, so its
 * result is bit-identical; `inc` lists, per vertex, the incident edges in ascending edge id with
 * bit 31 set when the vertex is the target (built by nltgv2_build_incidence). ---- */
void nltgv2_build_incidence(const nltgv2_graph* g, int32_t* row /* V+1 */, int32_t* inc /* 2E */) {
  memset(row, 0, sizeof(int32_t) * ((size_t)g->V + 1));
  for (int32_t e = 0; e < g->E; ++e) { row[g->edges[2 * e] + 1]++; row[g->edges[2 * e + 1] + 1]++; }
  for (int32_t v = 0; v < g->V; ++v) row[v + 1] += row[v];
  int32_t* fill = (int32_t*)malloc(sizeof(int32_t) * (size_t)(g->V > 0 ? g->V : 1));
  memcpy(fill, row, sizeof(int32_t) * (size_t)g->V);
  for (int32_t e = 0; e < g->E; ++e) {
    inc[fill[g->edges[2 * e]]++] = e;
    inc[fill[g->edges[2 * e + 1]]++] = e | (int32_t)0x80000000;
  }
  free(fill);
}

int nltgv2_solve_omp(const nltgv2_params* p, nltgv2_graph* g, const int32_t* row,
                     const int32_t* inc, int num_iters, int num_threads) {
  const float sigma = p->step_q, tau = p->step_x, th = p->theta, tl = p->step_x * p->data_factor;
#ifdef _OPENMP
#pragma omp parallel num_threads(num_threads)
#endif
  for (int it = 0; it < num_iters; ++it) {
#ifdef _OPENMP
#pragma omp for schedule(static)
#endif
    for (int32_t e = 0; e < g->E; ++e) {
      const int32_t i = g->edges[2 * e], j = g->edges[2 * e + 1];
      const float dx = g->pos[2 * i] - g->pos[2 * j], dy = g->pos[2 * i + 1] - g->pos[2 * j + 1];
      float t = g->xb[i] - g->xb[j];
      t = fmaf(-g->w1b[i], dx, t);
      t = fmaf(-g->w2b[i], dy, t);
      float* q = g->q + 3 * e;
      q[0] = proj_unit(fmaf(sigma, g->alpha[e] * t, q[0]));
      q[1] = proj_unit(fmaf(sigma, g->beta[e] * (g->w1b[i] - g->w1b[j]), q[1]));
      q[2] = proj_unit(fmaf(sigma, g->beta[e] * (g->w2b[i] - g->w2b[j]), q[2]));
    }
#ifdef _OPENMP
#pragma omp for schedule(static)
#endif
    for (int32_t v = 0; v < g->V; ++v) {
      const float xp = g->x[v], w1p = g->w1[v], w2p = g->w2[v];
      float x = xp, w1 = w1p, w2 = w2p;
      for (int32_t s = row[v]; s < row[v + 1]; ++s) {
        const int32_t e = inc[s] & 0x7fffffff;
        const int32_t i = g->edges[2 * e], j = g->edges[2 * e + 1];
        const float dx = g->pos[2 * i] - g->pos[2 * j], dy = g->pos[2 * i + 1] - g->pos[2 * j + 1];
        const float* q = g->q + 3 * e;
        const float aq = g->alpha[e] * q[0], b2 = g->beta[e] * q[1], b3 = g->beta[e] * q[2];
        if (inc[s] >= 0) {
          x = fmaf(-tau, aq, x);
          w1 = fmaf(-tau, fmaf(-dx, aq, b2), w1);
          w2 = fmaf(-tau, fmaf(-dy, aq, b3), w2);
        } else {
          x = fmaf(-tau, -aq, x);
          w1 = fmaf(-tau, -b2, w1);
          w2 = fmaf(-tau, -b3, w2);
        }
      }
      const float t = tl * g->wgt[v], z = g->z[v], r = x - z;
      float xn = (r > t) ? (x - t) : ((r < -t) ? (x + t) : z);
      xn = fminf(fmaxf(xn, p->x_min), p->x_max);
      g->x[v] = xn; g->w1[v] = w1; g->w2[v] = w2;
      g->xb[v] = fmaf(th, xn - xp, xn);
      g->w1b[v] = fmaf(th, w1 - w1p, w1);
      g->w2b[v] = fmaf(th, w2 - w2p, w2);
    }
  }
  (void)num_threads;
  return 0;
}

/* ---- "next" row f1: mesh vertices as flame_ros packs them for /flame/mesh.
 * Restates reference src/utils.cc:184-209 (publishDepthMesh): valid iff idepth is not NaN and
 * > 0; uhom = (u, v, 1) / id; p = Kinv * uhom; texture coords u / (cols-1), v / (rows-1); invalid
 * vertices get NaN xyz and nothing else.  Layout = flame_ros::PointNormalUV (reference
 * src/utils.h:47-53: PCL_ADD_POINT4D, PCL_ADD_NORMAL4D, u, v, 16-byte aligned = 12 floats).  Faces
 * (reference src/utils.cc:216-230): valid triangles, winding reversed.  The matrix-vector product
 * is evaluated (K0*q0 + K1*q1) + K2*q2 without contraction (Eigen's own order is not pinned:
 * PARITY UNPINNED applies to the last ulp here too). */
void nltgv2_mesh_points(const float Kinv[9], int32_t V, const float* pos, const float* x,
                        const float* vtx_normals, int32_t width, int32_t height, float* out12) {
  const float wm1 = (float)(width - 1), hm1 = (float)(height - 1);
  for (int32_t v = 0; v < V; ++v) {
    float* o = out12 + 12 * (size_t)v;
    for (int k = 0; k < 12; ++k) o[k] = 0.0f;
    const float id = x[v];
    if (!isnan(id) && id > 0.0f) {
      const float q0 = pos[2 * v] / id, q1 = pos[2 * v + 1] / id, q2 = 1.0f / id;
      for (int r = 0; r < 3; ++r) o[r] = (Kinv[3 * r] * q0 + Kinv[3 * r + 1] * q1) + Kinv[3 * r + 2] * q2;
      o[4] = vtx_normals[3 * v]; o[5] = vtx_normals[3 * v + 1]; o[6] = vtx_normals[3 * v + 2];
      o[8] = pos[2 * v] / wm1;
      o[9] = pos[2 * v + 1] / hm1;
    } else {
      o[0] = o[1] = o[2] = NAN;
    }
  }
}

int32_t nltgv2_mesh_faces(int32_t T, const int32_t* tris, const uint8_t* tri_valid, int32_t* faces) {
  int32_t n = 0;
  for (int32_t t = 0; t < T; ++t)
    if (tri_valid[t]) {
      faces[3 * n] = tris[3 * t + 2]; faces[3 * n + 1] = tris[3 * t + 1]; faces[3 * n + 2] = tris[3 * t];
      ++n;
    }
  return n;
}

/* ---- "next" row f2: dense inverse-depth map, depth map and point cloud.
 * (i) rasterisation of the mesh (upstream getInverseDepthMap / getFilteredInverseDepthMap,
 * reference src/flame_offline_tum.cc:643, src/flame_nodelet.cc:688; the upstream rasteriser is
 * not in the reference tree, so this is the build's own precise rule): a pixel centre (jj, ii)
 * belongs to the LOWEST-index triangle whose three edge functions have one sign (zero included);
 * its idepth is the barycentric interpolation of the vertex idepths; uncovered pixels are NaN;
 * `filtered` keeps only triangles with tri_valid != 0.  (ii) depth = 1/idepth where idepth is not
 * NaN and > 0, else NaN: reference src/flame_offline_tum.cc:650-661.  (iii) cloud: NaN if depth is
 * NaN or outside [min_depth, max_depth], else Kinv * (jj*depth, ii*depth, depth): reference
 * src/utils.cc:290-312. ---- */
static inline float edge_fn(float ax, float ay, float bx, float by, float px, float py) {
  return fmaf(bx - ax, py - ay, -((by - ay) * (px - ax)));
}

void nltgv2_idepthmap(int32_t width, int32_t height, int32_t T, const float* pos, const float* x,
                      const int32_t* tris, const uint8_t* tri_valid, int32_t filtered,
                      float* idepthmap) {
  for (int64_t k = 0; k < (int64_t)width * height; ++k) idepthmap[k] = NAN;
  for (int32_t t = T - 1; t >= 0; --t) { /* descending, so the lowest index wins overlaps */
    if (filtered && !tri_valid[t]) continue;
    const int32_t a = tris[3 * t], b = tris[3 * t + 1], c = tris[3 * t + 2];
    const float ax = pos[2 * a], ay = pos[2 * a + 1], bx = pos[2 * b], by = pos[2 * b + 1];
    const float cx = pos[2 * c], cy = pos[2 * c + 1];
    const float area = edge_fn(ax, ay, bx, by, cx, cy);
    if (!(area != 0.0f)) continue;
    int32_t x0 = (int32_t)ceilf(fminf(ax, fminf(bx, cx))), x1 = (int32_t)floorf(fmaxf(ax, fmaxf(bx, cx)));
    int32_t y0 = (int32_t)ceilf(fminf(ay, fminf(by, cy))), y1 = (int32_t)floorf(fmaxf(ay, fmaxf(by, cy)));
    if (x0 < 0) x0 = 0;
    if (y0 < 0) y0 = 0;
    if (x1 > width - 1) x1 = width - 1;
    if (y1 > height - 1) y1 = height - 1;
    for (int32_t ii = y0; ii <= y1; ++ii)
      for (int32_t jj = x0; jj <= x1; ++jj) {
        const float px = (float)jj, py = (float)ii;
        const float wa = edge_fn(bx, by, cx, cy, px, py);
        const float wb = edge_fn(cx, cy, ax, ay, px, py);
        const float wc = edge_fn(ax, ay, bx, by, px, py);
        const int in = (wa >= 0.0f && wb >= 0.0f && wc >= 0.0f) || (wa <= 0.0f && wb <= 0.0f && wc <= 0.0f);
        if (!in) continue;
        const float num = fmaf(wc, x[c], fmaf(wb, x[b], wa * x[a]));
        idepthmap[(int64_t)ii * width + jj] = num / ((wa + wb) + wc);
      }
  }
}

void nltgv2_depth_and_cloud(int32_t width, int32_t height, const float* idepthmap,
                            const float Kinv[9], float min_depth, float max_depth,
                            float* depthmap, float* cloud /* 3 per pixel or NULL */) {
  for (int32_t ii = 0; ii < height; ++ii)
    for (int32_t jj = 0; jj < width; ++jj) {
      const int64_t k = (int64_t)ii * width + jj;
      const float id = idepthmap[k];
      float depth = NAN;
      if (!isnan(id) && id > 0.0f) depth = 1.0f / id;
      depthmap[k] = depth;
      if (!cloud) continue;
      float* o = cloud + 3 * k;
      if (isnan(depth) || depth < min_depth || depth > max_depth) {
        o[0] = o[1] = o[2] = NAN;
      } else {
        const float q0 = (float)jj * depth, q1 = (float)ii * depth, q2 = depth;
        for (int r = 0; r < 3; ++r) o[r] = (Kinv[3 * r] * q0 + Kinv[3 * r + 1] * q1) + Kinv[3 * r + 2] * q2;
      }
    }
}

/* ---- stat key `coverage` (read at reference src/utils.cc:122, msg/FlameStats.msg:11) and the
 * debug images flame_ros publishes (reference src/flame_offline_tum.cc:731-766; what each shows is
 * stated at cfg/flame_offline_tum.yaml:58-64: "Mesh wireframe colored by idepth", "Features colored
 * by idepth", "Image colored by interpolated normal vectors", "Colored idepthmap").  Upstream's
 * drawing code is not in the reference tree; this is the build's precise statement.  All images
 * are BGR8 (published as "bgr8", src/flame_offline_tum.cc:730), drawn on black (the input image
 * is not kept), W x H, row-major.
 *   coverage  : (number of pixels of the FILTERED dense idepthmap that are not NaN) / (W * H),
 *               one float32 division of the two integers converted to float32
 *   colour(id): jet(id * scene_color_scale, 0, 2) -- include/flame/utils/visualization.h jet()
 *   wireframe : for every valid triangle t in ascending t, sides k = 0,1,2 (vertices tris[3t+k],
 *               tris[3t+(k+1)%3]): Bresenham line between the rounded (half away from zero)
 *               vertex positions, clipped per pixel, colour(0.5 * (x_a + x_b)); later lines
 *               overwrite earlier ones
 *   features  : for every raw feature f in ascending f a 3x3 square around its rounded position,
 *               colour(mu_f); later features overwrite earlier ones
 *   idepthmap : colour(idepthmap_filtered(p)) where that is not NaN, else black
 *   normals   : for every pixel covered by the filtered dense map (same owner rule: the lowest-index
 *               valid covering triangle), n = sum_k w_k * vtx_normal_k / ((wa + wb) + wc) with the
 *               barycentric edge functions of nltgv2_idepthmap, normalised (a zero vector reads
 *               (0, 0, -1)); channel c = (uint8)(255 * (0.5 * n_c + 0.5) + 0.5), stored B = n_z,
 *               G = n_y, R = n_x; uncovered pixels black. ---- */
float nltgv2_coverage(int32_t width, int32_t height, const float* idepthmap_filtered) {
  int64_t cnt = 0;
  for (int64_t k = 0; k < (int64_t)width * height; ++k) cnt += !isnan(idepthmap_filtered[k]);
  return (float)cnt / (float)((int64_t)width * height);
}

static inline float ramp01(float v) { return fmaxf(0.0f, fminf(1.0f, v)); }
static void jet_bgr(float v, float vmin, float vmax, uint8_t* bgr) {
  float t = (vmax > vmin) ? (v - vmin) / (vmax - vmin) : 0.0f;
  t = ramp01(t);
  const float r = ramp01(1.5f - fabsf(4.0f * t - 3.0f));
  const float g = ramp01(1.5f - fabsf(4.0f * t - 2.0f));
  const float b = ramp01(1.5f - fabsf(4.0f * t - 1.0f));
  bgr[0] = (uint8_t)(255.0f * b + 0.5f);
  bgr[1] = (uint8_t)(255.0f * g + 0.5f);
  bgr[2] = (uint8_t)(255.0f * r + 0.5f);
}
static inline int32_t round_px(float v) { return (int32_t)(v + (v >= 0.0f ? 0.5f : -0.5f)); }
static inline void put_px(uint8_t* img, int32_t W, int32_t H, int32_t x, int32_t y, const uint8_t* c) {
  if (x < 0 || y < 0 || x >= W || y >= H) return;
  uint8_t* o = img + 3 * ((int64_t)y * W + x);
  o[0] = c[0]; o[1] = c[1]; o[2] = c[2];
}

void nltgv2_debug_image(int32_t kind, int32_t W, int32_t H, float scene_color_scale, int32_t V,
                        const float* pos, const float* x, int32_t T, const int32_t* tris,
                        const uint8_t* tri_valid, const float* vtx_normals,
                        const float* idepthmap_filtered, int32_t n_feat, const float* feat_pos,
                        const float* feat_mu, uint8_t* bgr) {
  (void)V;
  memset(bgr, 0, 3 * (size_t)W * H);
  uint8_t c[3];
  if (kind == 0) { /* wireframe */
    for (int32_t t = 0; t < T; ++t) {
      if (!tri_valid[t]) continue;
      for (int k = 0; k < 3; ++k) {
        const int32_t a = tris[3 * t + k], b = tris[3 * t + (k + 1) % 3];
        jet_bgr((0.5f * (x[a] + x[b])) * scene_color_scale, 0.0f, 2.0f, c);
        int32_t x0 = round_px(pos[2 * a]), y0 = round_px(pos[2 * a + 1]);
        const int32_t x1 = round_px(pos[2 * b]), y1 = round_px(pos[2 * b + 1]);
        const int32_t dx = abs(x1 - x0), dy = -abs(y1 - y0);
        const int32_t sx = x0 < x1 ? 1 : -1, sy = y0 < y1 ? 1 : -1;
        int32_t err = dx + dy;
        for (int32_t guard = 0; guard < 4 * (W + H); ++guard) {
          put_px(bgr, W, H, x0, y0, c);
          if (x0 == x1 && y0 == y1) break;
          const int32_t e2 = 2 * err;
          if (e2 >= dy) { err += dy; x0 += sx; }
          if (e2 <= dx) { err += dx; y0 += sy; }
        }
      }
    }
  } else if (kind == 1) { /* features */
    for (int32_t f = 0; f < n_feat; ++f) {
      jet_bgr(feat_mu[f] * scene_color_scale, 0.0f, 2.0f, c);
      const int32_t px = round_px(feat_pos[2 * f]), py = round_px(feat_pos[2 * f + 1]);
      for (int32_t dy = -1; dy <= 1; ++dy)
        for (int32_t dx = -1; dx <= 1; ++dx) put_px(bgr, W, H, px + dx, py + dy, c);
    }
  } else if (kind == 3) { /* idepthmap */
    for (int64_t k = 0; k < (int64_t)W * H; ++k) {
      const float id = idepthmap_filtered[k];
      if (isnan(id)) continue;
      jet_bgr(id * scene_color_scale, 0.0f, 2.0f, bgr + 3 * k);
    }
  } else if (kind == 2) { /* normals: the raster rule of nltgv2_idepthmap, filtered */
    for (int32_t t = T - 1; t >= 0; --t) {
      if (!tri_valid[t]) continue;
      const int32_t a = tris[3 * t], b = tris[3 * t + 1], cc = tris[3 * t + 2];
      const float ax = pos[2 * a], ay = pos[2 * a + 1], bx = pos[2 * b], by = pos[2 * b + 1];
      const float cx = pos[2 * cc], cy = pos[2 * cc + 1];
      const float area = edge_fn(ax, ay, bx, by, cx, cy);
      if (!(area != 0.0f)) continue;
      int32_t x0 = (int32_t)ceilf(fminf(ax, fminf(bx, cx))), x1 = (int32_t)floorf(fmaxf(ax, fmaxf(bx, cx)));
      int32_t y0 = (int32_t)ceilf(fminf(ay, fminf(by, cy))), y1 = (int32_t)floorf(fmaxf(ay, fmaxf(by, cy)));
      if (x0 < 0) x0 = 0;
      if (y0 < 0) y0 = 0;
      if (x1 > W - 1) x1 = W - 1;
      if (y1 > H - 1) y1 = H - 1;
      for (int32_t ii = y0; ii <= y1; ++ii)
        for (int32_t jj = x0; jj <= x1; ++jj) {
          const float px = (float)jj, py = (float)ii;
          const float wa = edge_fn(bx, by, cx, cy, px, py);
          const float wb = edge_fn(cx, cy, ax, ay, px, py);
          const float wc = edge_fn(ax, ay, bx, by, px, py);
          const int in = (wa >= 0.0f && wb >= 0.0f && wc >= 0.0f) || (wa <= 0.0f && wb <= 0.0f && wc <= 0.0f);
          if (!in) continue;
          const float s = (wa + wb) + wc;
          float n[3];
          for (int r = 0; r < 3; ++r)
            n[r] = fmaf(wc, vtx_normals[3 * cc + r], fmaf(wb, vtx_normals[3 * b + r], wa * vtx_normals[3 * a + r])) / s;
          const float len = sqrtf(fmaf(n[2], n[2], fmaf(n[1], n[1], n[0] * n[0])));
          if (len > 0.0f) { n[0] /= len; n[1] /= len; n[2] /= len; } else { n[0] = 0.f; n[1] = 0.f; n[2] = -1.f; }
          uint8_t* o = bgr + 3 * ((int64_t)ii * W + jj);
          o[0] = (uint8_t)(255.0f * (0.5f * n[2] + 0.5f) + 0.5f);
          o[1] = (uint8_t)(255.0f * (0.5f * n[1] + 0.5f) + 0.5f);
          o[2] = (uint8_t)(255.0f * (0.5f * n[0] + 0.5f) + 0.5f);
        }
    }
  }
}

/* ---- row a9: optional graph median / low-pass filters of the vertex idepths.
 * Evidence that they exist: timing stat keys median_filter / lowpass_filter (reference
 * msg/FlameStats.msg:45-46, src/utils.cc:155-156) and YAML keys regularization/do_median_filter,
 * do_lowpass_filter (cfg/flame_offline_tum.yaml:85-86, never read by flame_ros => upstream default,
 * off).  Upstream's arithmetic is not in the reference tree; this is the build's precise rule,
 * Jacobi style (all vertices read the pre-filter values):
 *   median : x_v <- lower median of {x_v} U {x_u : u adjacent to v}  (element (n-1)/2 of the
 *            ascending order; ties broken by position: self first, then incidence order)
 *   lowpass: x_v <- (x_v + sum_u x_u) / (1 + deg v), summed self first then in ascending edge id
 * Afterwards the extrapolated value x_bar is set to the filtered x. ---- */
void nltgv2_graph_filter(nltgv2_graph* g, const int32_t* row, const int32_t* inc, int32_t kind,
                         float* scratch /* V */) {
  for (int32_t v = 0; v < g->V; ++v) {
    const int32_t n = row[v + 1] - row[v] + 1;
    /* value k of the multiset: k = 0 is the vertex itself, k >= 1 its (k-1)-th incidence */
#define NB_VAL(k) ((k) == 0 ? g->x[v] : g->x[(inc[row[v] + (k) - 1] < 0) ? g->edges[2 * (inc[row[v] + (k) - 1] & 0x7fffffff)] \
                                                                           : g->edges[2 * inc[row[v] + (k) - 1] + 1]])
    if (kind == 0) {
      float med = g->x[v];
      for (int32_t i = 0; i < n; ++i) {
        const float xi = NB_VAL(i);
        int32_t rank = 0;
        for (int32_t j = 0; j < n; ++j) {
          const float xj = NB_VAL(j);
          rank += (xj < xi) || (xj == xi && j < i);
        }
        if (rank == (n - 1) / 2) med = xi;
      }
      scratch[v] = med;
    } else {
      float sum = g->x[v];
      for (int32_t k = 1; k < n; ++k) sum += NB_VAL(k);
      scratch[v] = sum / (float)n;
    }
#undef NB_VAL
  }
  for (int32_t v = 0; v < g->V; ++v) { g->x[v] = scratch[v]; g->xb[v] = scratch[v]; }
}

/* ---- row a7: graph sync -- what Flame::update() does between the Delaunay
 * triangulation and the first regulariser step.  The parameters cross the boundary at reference
 * src/flame_offline_tum.cc:234-249 with the meanings stated in cfg/flame_offline_tum.yaml:87-92:
 *   idepth_var_max_graph  "Maximum idepth var before feature can be added to graph" (:92)
 *   adaptive_data_weights "Set vertex data weights to inverse idepth variance"       (:89)
 *   rescale_data          "Rescale data to have mean 1"                              (:90)
 *   init_with_prediction  "Initialize vertex idepths with predicted value from dense idepthmap" (:91)
 * Upstream's code for it is not in the reference tree; this is the build's precise statement:
 *   gate    : feature v may enter the graph iff var_v < idepth_var_max_graph (strict)
 *   edges   : the unique undirected edges of the triangulation, oriented i < j, in lexicographic
 *             order of (i, j); alpha_e = beta_e = 1 / sqrt(dx^2 + dy^2) in float32, dx^2 + dy^2 NOT
 *             fused ([UPSTREAM-RECALL] reciprocal pixel edge length; edge_weight_rule / alpha_gain /
 *             beta_gain select the alternatives listed in nltgv2_oracle.h)
 *   scale   : rescale_data ? (float)(sum_v (double)mu_v / V) : 1; a scale that is not > 0 reads 1
 *   z_v     = mu_v / scale;  wgt_v = adaptive ? 1 / var_v : 1
 *   x0_v    = (init_with_prediction && prediction && isfinite(prediction_v)) ? prediction_v / scale
 *                                                                             : z_v
 * Returns E. ---- */
int32_t nltgv2_feature_gate(int32_t n, const float* var, float var_max, uint8_t* keep) {
  int32_t cnt = 0;
  for (int32_t v = 0; v < n; ++v) { keep[v] = var[v] < var_max ? 1 : 0; cnt += keep[v]; }
  return cnt;
}

static int cmp_pair(const void* a, const void* b) {
  const int32_t* x = (const int32_t*)a; const int32_t* y = (const int32_t*)b;
  if (x[0] != y[0]) return x[0] < y[0] ? -1 : 1;
  if (x[1] != y[1]) return x[1] < y[1] ? -1 : 1;
  return 0;
}

int32_t nltgv2_graph_sync(const nltgv2_sync_params* sp, int32_t V, int32_t T, const float* pos,
                          const float* mu, const float* var, const int32_t* tris,
                          const float* prediction, int32_t* edges /* cap 2*3T */, float* alpha,
                          float* beta, float* z, float* wgt, float* x0, float* scale_out) {
  int32_t n = 0;
  for (int32_t t = 0; t < T; ++t)
    for (int k = 0; k < 3; ++k) {
      const int32_t a = tris[3 * t + k], b = tris[3 * t + (k + 1) % 3];
      edges[2 * n] = a < b ? a : b;
      edges[2 * n + 1] = a < b ? b : a;
      ++n;
    }
  qsort(edges, (size_t)n, 2 * sizeof(int32_t), cmp_pair);
  int32_t E = 0;
  for (int32_t k = 0; k < n; ++k)
    if (E == 0 || edges[2 * k] != edges[2 * E - 2] || edges[2 * k + 1] != edges[2 * E - 1]) {
      edges[2 * E] = edges[2 * k]; edges[2 * E + 1] = edges[2 * k + 1]; ++E;
    }
  for (int32_t e = 0; e < E; ++e) {
    const int32_t i = edges[2 * e], j = edges[2 * e + 1];
    const float dx = pos[2 * i] - pos[2 * j], dy = pos[2 * i + 1] - pos[2 * j + 1];
    const float len = sqrtf(dx * dx + dy * dy);
    const float inv = 1.0f / len;
    const int32_t rule = sp->edge_weight_rule;
    float a = (rule == 1 || rule == 3) ? 1.0f : inv;
    float b = (rule == 1 || rule == 2) ? 1.0f : inv;
    if (sp->alpha_gain != 0.0f) a *= sp->alpha_gain;
    if (sp->beta_gain != 0.0f) b *= sp->beta_gain;
    alpha[e] = a;
    beta[e] = b;
  }
  float scale = 1.0f;
  if (sp->rescale_data && V > 0) {
    double s = 0.0;
    for (int32_t v = 0; v < V; ++v) s += (double)mu[v];
    scale = (float)(s / (double)V);
    if (!(scale > 0.0f)) scale = 1.0f;
  }
  for (int32_t v = 0; v < V; ++v) {
    z[v] = mu[v] / scale;
    wgt[v] = sp->adaptive_data_weights ? 1.0f / var[v] : 1.0f;
    x0[v] = (sp->init_with_prediction && prediction && isfinite(prediction[v])) ? prediction[v] / scale : z[v];
  }
  if (scale_out) *scale_out = scale;
  return E;
}

/* Back to the caller's units after a rescaled solve: primal state and data term times s (the dual
 * state is scale free).  One float32 multiply per value. */
void nltgv2_scale_state(nltgv2_graph* g, float* z_mut, float s) {
  for (int32_t v = 0; v < g->V; ++v) {
    g->x[v] *= s; g->w1[v] *= s; g->w2[v] *= s;
    g->xb[v] *= s; g->w1b[v] *= s; g->w2b[v] *= s;
    if (z_mut) z_mut[v] *= s;
  }
}
