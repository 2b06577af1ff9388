/*
 * oracle/nltgv2_oracle.h -- CPU restatement of FLaME's NLTGV2-L1 graph regulariser.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under flame_ros_amd/ or include/ may include, link or
 * call this file; only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg do.
 *
 * PARITY UNPINNED.  The algorithm lives in the un-vendored, un-pinned dependency
 * robustrobotics/flame (reference: CMakeLists.txt:57 find_package(flame), README.md:12-13,73
 * clones master HEAD).  Its source is not under /root/reference, the reference holds no tests or
 * golden vectors (CMakeLists.txt:271-281 is a commented template), so this file restates the
 * published algorithm (Greene & Roy, ICCV'17, cited at reference README.md:21-23; SURVEY.md
 * section 8a rows a2-a6) and is pinned only by analytic known-answer tests and an independent
 * float64 NumPy restatement (oracle/nltgv2_np.py).  What the reference DOES pin is the parameter
 * set that crosses the boundary: rparams.{data_factor,step_x,step_q,theta}
 * (reference src/flame_offline_tum.cc:242-245, defaults cfg/flame_offline_tum.yaml:93-96) and the
 * cost stat keys nltgv2_{total,avg}_{smoothness,data}_cost (reference src/utils.cc:131-136,
 * msg/FlameStats.msg:22-25).
 *
 * Arithmetic contract (what "bit-exact vs the oracle" means for the HIP path): float32, every
 * fused multiply-add is an explicit fmaf(), nothing else may be contracted (build with
 * -ffp-contract=off), and the primal scatter visits edges in ascending edge index.
 */
#ifndef NLTGV2_ORACLE_H_
#define NLTGV2_ORACLE_H_
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* mirrors flame::Params::rparams (reference src/flame_offline_tum.cc:242-245) + idepth clamp */
typedef struct {
  float data_factor; /* lambda, cfg/flame_offline_tum.yaml:93 */
  float step_x;      /* tau,    :94 */
  float step_q;      /* sigma,  :95 */
  float theta;       /* :96 */
  float x_min, x_max; /* idepth clamp after the prox (upstream default recalled 0..10) */
} nltgv2_params;

/* Graph in plain arrays.  Edge e is oriented edges[2e] -> edges[2e+1] (source i, target j). */
typedef struct {
  int32_t V, E;
  const float* pos;     /* 2V pixel coords (u,v) */
  const int32_t* edges; /* 2E */
  const float* alpha;   /* E */
  const float* beta;    /* E */
  const float* z;       /* V data term (measured idepth) */
  const float* wgt;     /* V data weight */
  float* x;  float* w1;  float* w2;    /* V primal */
  float* xb; float* w1b; float* w2b;   /* V extrapolated primal */
  float* q;                            /* 3E dual, interleaved q1,q2,q3 per edge */
} nltgv2_graph;

void nltgv2_dual_step(const nltgv2_params* p, nltgv2_graph* g);
/* primal step; xp/w1p/w2p (each V floats) receive the pre-step values */
void nltgv2_primal_step(const nltgv2_params* p, nltgv2_graph* g, float* xp, float* w1p, float* w2p);
void nltgv2_extragradient_step(const nltgv2_params* p, nltgv2_graph* g, const float* xp,
                               const float* w1p, const float* w2p);
/* one PD iteration = dual; primal; extra-gradient.  scratch = 3V floats */
void nltgv2_step(const nltgv2_params* p, nltgv2_graph* g, float* scratch);
/* num_iters iterations; returns 0, or -1 on allocation failure */
int nltgv2_solve(const nltgv2_params* p, nltgv2_graph* g, int num_iters);
/* smoothness and data cost (float32 terms, float64 accumulation) */
void nltgv2_costs(const nltgv2_params* p, const nltgv2_graph* g, double* smooth, double* data);

/* K u and K^T q, exposed for the adjointness known-answer test.  u=(x,w1,w2) 3 arrays of V,
 * Ku = 3E interleaved; KTq = 3 arrays of V. */
void nltgv2_apply_K(const nltgv2_graph* g, const float* x, const float* w1, const float* w2,
                    float* Ku);
void nltgv2_apply_KT(const nltgv2_graph* g, const float* q, float* kx, float* kw1, float* kw2);

/* Multi-threaded variant for the threaded cpu_baseline legs:
 * bit-identical to nltgv2_solve.  row V+1 / inc 2E from nltgv2_build_incidence. */
void nltgv2_build_incidence(const nltgv2_graph* g, int32_t* row, int32_t* inc);
int nltgv2_solve_omp(const nltgv2_params* p, nltgv2_graph* g, const int32_t* row,
                     const int32_t* inc, int num_iters, int num_threads);

/* ---- per-triangle stage ---- */
typedef struct {
  int32_t do_oblique_triangle_filter; /* cfg/flame_offline_tum.yaml:40 */
  float oblique_normal_thresh;        /* :41 rad */
  float oblique_idepth_diff_factor;   /* :42 */
  float oblique_idepth_diff_abs;      /* :43 */
  int32_t do_edge_length_filter;      /* :47 */
  float edge_length_thresh;           /* :48 fraction of image width */
  int32_t do_idepth_triangle_filter;  /* :52 */
  float min_triangle_idepth;          /* :53 */
  int32_t width, height;
} nltgv2_tri_params;

/* tris = 3T vertex ids.  Outputs: tri_normals 3T, tri_valid T, vtx_normals 3V. */
void nltgv2_triangles(const nltgv2_tri_params* tp, const float Kinv[9], int32_t V, int32_t T,
                      const float* pos, const float* x, const int32_t* tris, float* tri_normals,
                      uint8_t* tri_valid, float* vtx_normals);

/* row a9: graph median (kind 0) / low-pass (kind 1) filter of the vertex idepths, one Jacobi pass;
 * row/inc from nltgv2_build_incidence, scratch V floats.  Sets x and x_bar. */
void nltgv2_graph_filter(nltgv2_graph* g, const int32_t* row, const int32_t* inc, int32_t kind,
                         float* scratch);

/* ---- row a7: graph sync (gate, edges of the triangulation, alpha/beta, rescale, adaptive weights,
 * prediction init); see the statement above nltgv2_graph_sync in nltgv2_oracle.c ---- */
typedef struct {
  int32_t adaptive_data_weights; /* cfg/flame_offline_tum.yaml:89 */
  int32_t rescale_data;          /* :90 */
  int32_t init_with_prediction;  /* :91 */
  float idepth_var_max_graph;    /* :92 */
  /* [UPSTREAM-RECALL] switches (all-zero = the build's default statement), so that a mismatch
   * against a state dump of a real robustrobotics/flame build is a flag flip, not a rewrite:
   * edge_weight_rule 0: alpha = beta = 1/len; 1: alpha = beta = 1; 2: alpha = 1/len, beta = 1;
   * 3: alpha = 1, beta = 1/len.  alpha_gain / beta_gain multiply the rule's value (0 reads 1). */
  int32_t edge_weight_rule;
  float alpha_gain, beta_gain;
} nltgv2_sync_params;
int32_t nltgv2_feature_gate(int32_t n, const float* var, float var_max, uint8_t* keep);
/* edges capacity 2*3T ints, alpha/beta capacity 3T; returns E */
int32_t nltgv2_graph_sync(const nltgv2_sync_params* sp, int32_t V, int32_t T, const float* pos,
                          const float* mu, const float* var, const int32_t* tris,
                          const float* prediction, int32_t* edges, float* alpha, float* beta,
                          float* z, float* wgt, float* x0, float* scale_out);
void nltgv2_scale_state(nltgv2_graph* g, float* z_mut, float s);

/* "next" row f1: mesh vertices in flame_ros::PointNormalUV layout (12 floats) and faces with
 * reversed winding (reference src/utils.cc:184-230).  Returns the number of faces. */
void nltgv2_mesh_points(const float Kinv[9], int32_t V, const float* pos, const float* x,
                        const float* vtx_normals, int32_t width, int32_t height, float* out12);
int32_t nltgv2_mesh_faces(int32_t T, const int32_t* tris, const uint8_t* tri_valid, int32_t* faces);

/* "next" row f2: dense idepthmap (lowest-index covering triangle, barycentric), depth = 1/idepth
 * (reference src/flame_offline_tum.cc:650-661) and point cloud (reference src/utils.cc:290-312). */
void nltgv2_idepthmap(int32_t width, int32_t height, int32_t T, const float* pos, const float* x,
                      const int32_t* tris, const uint8_t* tri_valid, int32_t filtered,
                      float* idepthmap);
void nltgv2_depth_and_cloud(int32_t width, int32_t height, const float* idepthmap,
                            const float Kinv[9], float min_depth, float max_depth,
                            float* depthmap, float* cloud);

/* stat key `coverage` (reference src/utils.cc:122) and the debug images (reference
 * src/flame_offline_tum.cc:731-766); kind 0 wireframe, 1 features, 2 normals, 3 idepthmap; bgr =
 * 3 W H bytes.  The rules are stated above nltgv2_coverage in nltgv2_oracle.c. */
float nltgv2_coverage(int32_t width, int32_t height, const float* idepthmap_filtered);
void nltgv2_debug_image(int32_t kind, int32_t W, int32_t H, float scene_color_scale, int32_t V,
                        const float* pos, const float* x, int32_t T, const int32_t* tris,
                        const uint8_t* tri_valid, const float* vtx_normals,
                        const float* idepthmap_filtered, int32_t n_feat, const float* feat_pos,
                        const float* feat_mu, uint8_t* bgr);

#ifdef __cplusplus
}
#endif
#endif
