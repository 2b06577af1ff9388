/*
 * include/flame_hip.h -- C ABI of libflame_hip.so: FLaME's NLTGV2-L1 graph regulariser and
 * per-triangle stage as hand-written HIP kernels for MI355X (gfx950).
 *
 * This is the drop-in boundary.  Upstream, flame::Flame::update() (called at reference
 * src/flame_offline_tum.cc:578-579,593-594, src/flame_offline_asl.cc:520,535,
 * src/flame_nodelet.cc:634-635) runs N x optimizers::nltgv2_l1_graph_regularizer::step() on the
 * Delaunay vertex graph and then the per-triangle stage whose results leave through
 * getInverseDepthMesh() (src/flame_offline_tum.cc:628-635).  A maintainer replaces that inner
 * loop by the calls below (INTEGRATION.md shows the binding); the C++ facade in include/flame/
 * does exactly that.
 *
 * Conventions: return 0 = ok, negative = error (flame_hip_strerror()); no exceptions, no global
 * state.  All pointers are HOST pointers owned by the caller for the duration of the call unless
 * the name says _dev.  One handle is not thread-safe; distinct handles are.  Edge e is oriented
 * edges[2e] -> edges[2e+1]; the orientation matters (the source vertex's plane slopes enter K1).
 * All arithmetic is float32 and bit-reproducible: identical inputs give identical bits on every
 * solver path and partitioning (see DESIGN.md "Arithmetic contract").
 */
#ifndef FLAME_HIP_H_
#define FLAME_HIP_H_
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* flame::Params::rparams as loaded at reference src/flame_offline_tum.cc:242-245
 * (defaults cfg/flame_offline_tum.yaml:93-96) + the idepth clamp applied after the prox. */
typedef struct {
  float data_factor; /* lambda  (regularization/nltgv2/data_factor) */
  float step_x;      /* tau     (.../step_x) */
  float step_q;      /* sigma   (.../step_q) */
  float theta;       /* extra-gradient weight (.../theta) */
  float x_min, x_max;
} flame_hip_params;

/* Triangle filter parameters as loaded at reference src/flame_offline_tum.cc:168-192
 * (cfg/flame_offline_tum.yaml:38-53). */
typedef struct {
  int32_t do_oblique_triangle_filter;
  float oblique_normal_thresh;      /* rad */
  float oblique_idepth_diff_factor; /* relative max/min idepth difference */
  float oblique_idepth_diff_abs;    /* absolute max/min idepth difference */
  int32_t do_edge_length_filter;
  float edge_length_thresh; /* fraction of image width */
  int32_t do_idepth_triangle_filter;
  float min_triangle_idepth;
  int32_t width, height;
} flame_hip_tri_params;

typedef struct flame_hip_graph flame_hip_graph; /* opaque; owns device buffers */

enum {
  FLAME_HIP_OK = 0,
  FLAME_HIP_ERR_ARG = -1,     /* bad argument (NULL, negative size, index out of range) */
  FLAME_HIP_ERR_STATE = -2,   /* call order (e.g. solve before upload) */
  FLAME_HIP_ERR_NAN = -3,     /* non-finite input */
  FLAME_HIP_ERR_ALLOC = -4,   /* host or device allocation failed */
  FLAME_HIP_ERR_NODEVICE = -5,
  FLAME_HIP_ERR_NORCCL = -6,  /* librccl.so could not be loaded (flame_hip_comm_* / flame_hip_part_*) */
  FLAME_HIP_ERR_HIP = -1000,  /* -(1000 + hipError_t) */
  FLAME_HIP_ERR_RCCL = -3000  /* -(3000 + ncclResult_t) */
};

/* Solver paths (flame_hip_set_option key "path"). */
enum {
  FLAME_HIP_PATH_AUTO = 0,
  FLAME_HIP_PATH_GLOBAL = 1, /* two kernels per iteration over global arrays */
  FLAME_HIP_PATH_TILE = 2    /* LDS-resident tiles, several iterations per launch */
};

/* Replaces: construction of the regulariser graph inside Flame::update. */
int flame_hip_graph_create(flame_hip_graph** out, int device, int32_t V, int32_t E, int32_t T);
void flame_hip_graph_destroy(flame_hip_graph* g);

/* One handle serves a frame STREAM: every frame of FLaME is a new graph (reference
 * src/flame_offline_tum.cc:578: update() re-triangulates per image).  Sets the sizes of the NEXT
 * flame_hip_graph_upload*; the handle keeps its stream, events, device buffers (capacity), host
 * plan buffers and the tile cost-density grid.  Registered halo lists are dropped.  Waits for any
 * solve still in flight. */
int flame_hip_graph_resize(flame_hip_graph* g, int32_t V, int32_t E, int32_t T);

/* Options are set BEFORE upload.  Keys: "path" (enum above), "tile_own" (target own vertices
 * per tile), "tile_depth" (halo depth = iterations per launch), "tile_threads", "use_graph"
 * (replay launches from a hipGraph), "plan_device" (1 = build halo-tile plans on the GPU, default),
 * "plan_reuse" (1 = default: on a frame stream the device builder takes the PARTITION of a frame from
 * the previous frame's tile map while the frames hold about as many vertices, instead of sorting and
 * bisecting again; results do not depend on the partition; flame_hip_get_info "plan_reused"),
 * "plan_mini" (1 = default: a graph sync of a small frame -- up to 2048 vertices / 4096 triangles, a
 * reused partition, a predicted edge count -- derives the edges and builds everything in front of the
 * tile pass in ONE launch of one workgroup instead of two dozen dependent ones; flame_hip_get_info
 * "plan_mini" tells whether the current plan was made that way),
 * (flame_hip_get_info "tile_imbalance_pct": 100 x max / mean of the tiles' modelled cost, local edges + 2 x
 * local vertices -- a launch lasts as long as its slowest tile; 125-155 on 50 k-vertex frames, the border
 * tiles' long hull edges)
 * "tile_single_max" (auto: graphs up to this many vertices become ONE LDS-resident tile, default
 * 512, up to 2048; a graph whose tile does not fit after all is partitioned, and the handle stops
 * trying at that size: flame_hip_get_info "single_cap"), "stream_depth" (0 = off, default; > 0: halo
 * depth of graphs of up to 2048 vertices in place of the automatic 8 -- for handles that solve every
 * graph ONCE, where the plan of shallow tiles is cheaper than the launches deep tiles save), "persist"
 * (default 1: a graph of 2 .. 256 halo tiles -- at most one per CU -- is solved by ONE launch of RESIDENT tiles:
 * neighbours hand their results over through uncached, round-tagged copies of the state arrays every `depth`
 * iterations instead of meeting at a kernel boundary; same bits, any placement of the tiles on the chip; 0: one
 * launch per `depth` iterations; flame_hip_get_info "persist_used" tells whether the last solve ran that way.  r05: with
 * automatic tile sizes that covers every graph up to ~240 000 vertices -- beyond 256 x 196 the tiles grow FAT (one per CU,
 * shallower halos, 12-byte incidence slots where 16 do not fit the LDS: flame_hip_get_info "tile_fat", "tile_slot12"); a handle
 * with "persist" 0 keeps the partition of two rounds of smaller tiles there.  The
 * launch ASSUMES that all its workgroups are on the chip at once.  The library keeps the tile count within the
 * CU count and lets one handle per device and process run such a launch at a time (another handle solving at the
 * same moment uses ordinary launches), but a foreign kernel that holds CUs for long -- another process, another
 * library -- can still keep tiles from starting: every wait inside the launch is bounded (4 ms until the handle has measured
 * a round of the current graph, then max(0.5 ms, 8 x that round); option "persist_timeout_us" > 0 fixes it; flame_hip_get_info
 * "persist_timeout_us" = what the last launch was given), a launch that
 * gave up is noticed at the next synchronising call and REPEATED by ordinary launches from its untouched source
 * buffers ("persist_recovered" counts those; r05: SEVERAL solves queued without a synchronising call between them are
 * repeated as a whole -- the error word does not say which one gave up, so the source of the first is copied aside when
 * the second is queued and every solve since is logged; FLAME_HIP_ERR_STATE only when something other than solves wrote
 * the state in between), and the whole process then stays off resident tiles for 16 solves, doubling with every further
 * give-up ("persist_gave_up").  How far the tiles of a handle's last give-up got, read from the hand-off copies afterwards:
 * "persist_gave_up_tile" / "_round" = the tile that had handed over the fewest rounds (0: it never started; otherwise it is the
 * late one or stood next to it) and how many, "_front_round" the most any tile had, "_not_started" tiles that had handed over
 * none, "_rounds" / "_tiles" / "_one_xcd" / "_timeout_us" the launch's own numbers; "persist_backoff" = solves the device's lease still
 * sits out (plans uploaded meanwhile are sized for launches), "one_xcd_allowed" = 0 once a one-XCD launch has given up.  "one_xcd" (r06, default 1): a graph of up to 32 resident tiles -- automatic sizes give 32 tiles to
 * graphs of 770 .. 1 280 vertices -- keeps them on ONE XCD (the launch has 8 x ntiles workgroups, every 8th carries a tile) and hands
 * over through ordinary memory, i.e. that XCD's L2, instead of uncached memory: 1.2 k vertices 0.89 -> 0.82 us per iteration.  Which
 * XCD a workgroup lands on is the dispatcher's habit, not a guarantee; the round tags and the bounded polls keep the result right
 * regardless: tiles that cannot see each other time out, the solve is repeated by launches and the device's lease drops the mode for
 * the rest of the process (flame_hip_get_info "one_xcd_used").  Tuning: "poll_delay" (x 256 clocks between a round's stores and its first poll pass, -1 =
 * automatic), "need_marks" (1 = default: fat tiles hand over only the entries somebody polls).  Diagnostics: "plan_timing" (1-5:
 * the plan builders print their stages' times to stderr); with option "persist_prof" = <tile + 1>
 * "persist_prof_0".."persist_prof_3" return that tile's time split of the last solve's rounds in 10 ns ticks:
 * iterations + stores, poll of the halo entries, halo applied + barrier, and the number of rounds),
 * "lane_order" (lanes of the tile plan re-assigned against LDS bank conflicts:
 * 0 never, 1 = when an uploaded graph is solved a second time (default; a frame stream never pays),
 * 2 = while the plan is built), "balance", "order_mode", "host_threads", "lds_bytes", "profile",
 * "d_sign" ([UPSTREAM-RECALL] switch: +1 (default) the edge vector entering K1 is d = pos_i - pos_j,
 * -1 it is pos_j - pos_i).  Unknown key -> FLAME_HIP_ERR_ARG.  The library reads NO environment variable: everything that
 * changes its behaviour is an option of a handle (fault injection for the tests exists only in a library of its own,
 * libflame_hip_hooks.so, flame_ros_amd/build.py). */
int flame_hip_set_option(flame_hip_graph* g, const char* key, int32_t value);
int flame_hip_get_info(const flame_hip_graph* g, const char* key, int64_t* value);

/* Replaces: graph sync inside Flame::update (row a7): vertex positions, oriented edge list,
 * edge weights, data terms.  pos 2V, edges 2E, alpha/beta E, z/wgt V, x0 V or NULL (= z),
 * tris 3T or NULL.  State is reset to x = x0, w = 0, x_bar = x, w_bar = 0, q = 0. */
int flame_hip_graph_upload(flame_hip_graph* g, const float* pos, const int32_t* edges,
                           const float* alpha, const float* beta, const float* z,
                           const float* wgt, const float* x0, const int32_t* tris);

/* Replaces: graph sync inside Flame::update (row a7) when the caller has the tracked features and
 * their Delaunay triangulation rather than a ready edge list: derives the unique undirected edges
 * (i < j, lexicographic), alpha = beta = 1/|pos_i - pos_j|, the data terms z = mu / scale with
 * scale = mean(mu) under rescale_data, the data weights (1 or 1/var) and the initial x (prediction
 * / scale where init_with_prediction and the prediction is finite, else z), resizes the handle to
 * (V, E, T) and uploads.  Parameters: reference src/flame_offline_tum.cc:234-249,
 * cfg/flame_offline_tum.yaml:87-92.  Every feature must pass the gate var < idepth_var_max_graph
 * (FLAME_HIP_ERR_ARG otherwise; flame_hip_feature_gate selects them BEFORE triangulation).
 * The solve then runs in rescaled units; flame_hip_scale_state(g, scale) brings the state back. */
typedef struct {
  int32_t adaptive_data_weights; /* regularization/nltgv2/adaptive_data_weights */
  int32_t rescale_data;          /* .../rescale_data */
  int32_t init_with_prediction;  /* .../init_with_prediction */
  float idepth_var_max_graph;    /* .../idepth_var_max */
  /* [UPSTREAM-RECALL] switches; an all-zero tail is the default.  edge_weight_rule 0: alpha = beta =
   * 1/|pos_i - pos_j|; 1: alpha = beta = 1; 2: alpha = 1/len, beta = 1; 3: alpha = 1, beta = 1/len.
   * alpha_gain / beta_gain multiply the rule's value (0 reads 1).  Any non-default value takes the
   * host sync path.  (Caller-supplied weights: flame_hip_graph_upload.) */
  int32_t edge_weight_rule;
  float alpha_gain, beta_gain;
} flame_hip_sync_params;
int flame_hip_graph_sync(flame_hip_graph* g, const flame_hip_sync_params* sp, int32_t V, int32_t T,
                         const float* pos, const float* idepth_mu, const float* idepth_var,
                         const int32_t* tris, const float* prediction, float* scale);
/* Row f3's first leg -- the Delaunay triangulation of a frame's features, on the GPU (the reference budgets it as
 * `triangulate` beside `sync_graph`, msg/FlameStats.msg:43-44; upstream calls Shewchuk's Triangle on the host; the
 * host triangulator of the same contract is include/flame/utils/delaunay.h).  pos: V x {u, v} pixel coordinates
 * (|u|, |v| < 2^13, snapped to a 2^-16 pixel lattice; exact predicates, so pixel lattices and collinear runs are
 * ordinary inputs).  tris receives *T <= tri_cap (2V always suffices) counter-clockwise triangles -- (b - a) x (c - a)
 * > 0 in these coordinates --, each starting at its smallest vertex, ordered by that vertex; cocircular points are cut
 * as a fan from their smallest id, points that coincide after snapping are triangulated once (smallest id).  The
 * list is a function of the input alone.  Independent of the graph held by g (own scratch, the handle's staging
 * stream); returns when the list is in tris.  Errors: ARG (coordinate out of range, tri_cap too small), NAN, STATE
 * (no device; or the result failed Euler's check T = 2 n - 2 - h: never seen, reported rather than handed out).
 * flame_hip_get_info: "delaunay_hull" (h), "delaunay_live" (n), "delaunay_us" (host time of the call).
 * A flame_hip_graph_sync on the same handle with the same V and T and tris = NULL uses this list as the library still
 * holds it -- on the DEVICE (r05): nothing of it crosses the host link on the frame path.
 * KEEP mode, tri_cap = 0 and tris = NULL: only *T comes back (one round trip behind the last kernel); the list stays on the
 * device for that graph sync and its host copy travels on a stream of its own meanwhile -- flame_hip_delaunay_list hands
 * it out (and waits for it) when the caller wants the triangles, e.g. while the GPU iterates. */
int flame_hip_delaunay(flame_hip_graph* g, int32_t V, const float* pos, int32_t tri_cap, int32_t* tris, int32_t* T);
int flame_hip_delaunay_list(flame_hip_graph* g, int32_t tri_cap, int32_t* tris);
/* keep[v] = var[v] < var_max; returns the number kept (or a negative error).  No device needed. */
int32_t flame_hip_feature_gate(int32_t n, const float* idepth_var, float var_max, uint8_t* keep);
/* The edge list flame_hip_graph_sync derived (2E ints; E from flame_hip_get_info "E"). */
int flame_hip_graph_edges(const flame_hip_graph* g, int32_t* edges);
/* x, w, x_bar, w_bar and z times s (asynchronous on the handle's stream). */
int flame_hip_scale_state(flame_hip_graph* g, float s);

/* New frame on an UNCHANGED topology (same vertices, edges, weights): only the data terms, data
 * weights and the initial x change.  Resets the state exactly like flame_hip_graph_upload but
 * keeps the host plan, the device graph arrays and the captured launch graphs.  z/wgt V, x0 V or
 * NULL (= z). */
int flame_hip_graph_update_data(flame_hip_graph* g, const float* z, const float* wgt, const float* x0);

/* Frames axis: a batch of `num_graphs` INDEPENDENT graphs in one handle (block-diagonal): graph b
 * owns the vertices [voff[b], voff[b+1]) (voff has num_graphs+1 entries, voff[0] = 0,
 * voff[num_graphs] = V); edge endpoints are vertex ids of the concatenated arrays and must stay
 * inside one graph.  Every graph becomes one LDS-resident tile, so one launch runs ALL iterations
 * of ALL frames (one workgroup per frame); a graph too large for one tile -> FLAME_HIP_ERR_ARG.
 * Same array conventions as flame_hip_graph_upload. */
int flame_hip_graph_upload_batch(flame_hip_graph* g, int32_t num_graphs, const int32_t* voff,
                                 const float* pos, const int32_t* edges, const float* alpha,
                                 const float* beta, const float* z, const float* wgt,
                                 const float* x0, const int32_t* tris);

/* Overwrite solver state (any pointer may be NULL = keep).  q is 3E interleaved. */
int flame_hip_set_state(flame_hip_graph* g, const float* x, const float* w1, const float* w2,
                        const float* xb, const float* w1b, const float* w2b, const float* q);

/* Replaces: the loop of nltgv2_l1_graph_regularizer::step() calls (rows a2-a5).  Asynchronous on
 * the handle's stream unless stream != NULL (a hipStream_t), in which case it runs there. */
int flame_hip_solve(flame_hip_graph* g, const flame_hip_params* p, int32_t num_iters,
                    void* stream);
int flame_hip_sync(flame_hip_graph* g);
/* Device time of the last flame_hip_solve in ms (HIP events on the solve's stream) and the
 * number of kernel launches it made.  Synchronises. */
int flame_hip_last_solve_ms(flame_hip_graph* g, float* ms, int32_t* launches);

/* Row a9: the optional graph filters applied to the vertex idepths (upstream options
 * regularization/do_median_filter, do_lowpass_filter, reference cfg/flame_offline_tum.yaml:85-86;
 * timing keys median_filter / lowpass_filter, msg/FlameStats.msg:45-46).  kind 0 = median of the
 * vertex and its neighbours (lower median), kind 1 = plain average of the vertex and its
 * neighbours; `passes` Jacobi passes; sets x and x_bar.  Asynchronous on the handle's stream. */
int flame_hip_graph_filter(flame_hip_graph* g, int32_t kind, int32_t passes);

/* Replaces: smoothnessCost()/dataCost() behind the stat keys nltgv2_total_smoothness_cost and
 * nltgv2_total_data_cost (reference src/utils.cc:131-136).  Synchronises. */
int flame_hip_costs(flame_hip_graph* g, const flame_hip_params* p, double* smooth, double* data);
/* The same sums restricted to flagged vertices (vmask, V bytes, caller's order, NULL = all) and
 * edges (emask, E bytes, NULL = all): a multi-GPU subdomain sums what it OWNS and the
 * ranks all-reduce the two doubles ("final cost reduction: one ncclAllReduce of 2 doubles"). */
int flame_hip_costs_masked(flame_hip_graph* g, const flame_hip_params* p, const uint8_t* vmask,
                           const uint8_t* emask, double* smooth, double* data);

/* Replaces: the per-triangle stage feeding getInverseDepthMesh (row a8).  Kinv row-major 3x3.
 * Outputs (any may be NULL): vtx_normals 3V, tri_valid T, tri_normals 3T.  Synchronises. */
int flame_hip_triangles(flame_hip_graph* g, const float Kinv[9], const flame_hip_tri_params* tp,
                        float* vtx_normals, uint8_t* tri_valid, float* tri_normals);

/* Everything flame::Flame::update() reads back after flame_hip_solve, in one call with ONE stream
 * synchronisation: the two costs (in the solver's units, i.e. before scale_back is applied), then
 * the state times scale_back (1 = leave it; rescale_data: the scale flame_hip_graph_sync returned),
 * the idepths x (V), the per-triangle stage (vtx_normals 3V, tri_valid T) and the edge list derived
 * by flame_hip_graph_sync (2E), and the stat key `coverage` (reference src/utils.cc:122): the share
 * of the tp->width x tp->height pixels the FILTERED dense idepthmap covers (not NaN).  Any output
 * pointer may be NULL.  Synchronises.  The un-scaling is
 * applied to the resident state in place and only ONCE per upload: on a state that is already back
 * in the caller's units (an earlier call, or flame_hip_scale_state) scale_back is ignored and asking
 * for the costs returns FLAME_HIP_ERR_STATE. */
int flame_hip_frame_results(flame_hip_graph* g, const flame_hip_params* p, float scale_back,
                            const float Kinv[9], const flame_hip_tri_params* tp, double* smooth,
                            double* data, float* x, float* vtx_normals, uint8_t* tri_valid,
                            int32_t* edges, float* coverage);

/* Debug images of flame::Flame rendered ON THE DEVICE (reference src/flame_offline_tum.cc:731-766;
 * what each shows: cfg/flame_offline_tum.yaml:58-64): BGR8, tp->width x tp->height, row-major, on
 * black.  WIREFRAME: sides of the valid triangles coloured by idepth; FEATURES: 3x3 squares at the
 * n_feat raw features (feat_pos 2 n_feat, feat_mu n_feat; ignored by the other kinds) coloured by
 * idepth; NORMALS: "image colored by interpolated normal vectors" over the filtered dense map;
 * IDEPTHMAP: jet of the filtered dense idepthmap.  Colour = jet(idepth * scene_color_scale, 0, 2)
 * (output/scene_color_scale, cfg/flame_offline_tum.yaml:35).  The exact rules are stated in
 * oracle/nltgv2_oracle.c (nltgv2_debug_image).  Nothing is drawn on the host; the dense raster is
 * shared with flame_hip_frame_results' coverage and flame_hip_depthmaps.  Synchronises. */
enum { FLAME_HIP_IMG_WIREFRAME = 0, FLAME_HIP_IMG_FEATURES = 1, FLAME_HIP_IMG_NORMALS = 2, FLAME_HIP_IMG_IDEPTHMAP = 3 };
int flame_hip_debug_image(flame_hip_graph* g, int32_t kind, const float Kinv[9],
                          const flame_hip_tri_params* tp, float scene_color_scale, int32_t n_feat,
                          const float* feat_pos, const float* feat_mu, uint8_t* bgr);

/* "Next" row f1: the mesh as flame_ros publishes it on /flame/mesh.  Replaces the
 * vertex loop and face loop of publishDepthMesh (reference src/utils.cc:184-230): points = V x 12
 * floats in flame_ros::PointNormalUV layout {x,y,z,0 | nx,ny,nz,0 | u/(W-1), v/(H-1), 0, 0}
 * (reference src/utils.h:47-53), NaN xyz for vertices whose idepth is NaN or <= 0; faces = valid
 * triangles with reversed winding, 3 vertex ids each (capacity 3T), *num_faces of them.  Runs the
 * triangle stage first.  Any output pointer may be NULL.  Synchronises. */
int flame_hip_mesh(flame_hip_graph* g, const float Kinv[9], const flame_hip_tri_params* tp,
                   float* points, int32_t* faces, int32_t* num_faces);

/* "Next" row f2: dense maps at tp->width x tp->height (row-major, any output may
 * be NULL).  idepthmap replaces getInverseDepthMap() (filtered = 0, reference
 * src/flame_nodelet.cc:688) / getFilteredInverseDepthMap() (filtered = 1: only valid triangles,
 * reference src/flame_offline_tum.cc:643): barycentric idepth of the lowest-index covering
 * triangle, NaN where uncovered.  depthmap replaces the OpenMP inversion loop (reference
 * src/flame_offline_tum.cc:650-661): 1/idepth where idepth is not NaN and > 0, else NaN.  cloud
 * (3 floats per pixel) replaces publishPointCloud's loop (reference src/utils.cc:290-312): NaN if
 * depth is NaN or outside [min_depth, max_depth], else Kinv (jj d, ii d, d).  Synchronises. */
int flame_hip_depthmaps(flame_hip_graph* g, const float Kinv[9], const flame_hip_tri_params* tp,
                        int32_t filtered, float min_depth, float max_depth, float* idepthmap,
                        float* depthmap, float* cloud);

/* Results out (any pointer may be NULL).  Caller's vertex/edge order.  Synchronises. */
int flame_hip_download(flame_hip_graph* g, float* x, float* w1, float* w2, float* q);
int flame_hip_download_bar(flame_hip_graph* g, float* xb, float* w1b, float* w2b);

/* ---- multi-GPU subdomains: halo exchange of the solver state every D
 * iterations.  The handle holds one subdomain (own vertices + D halo rings); the caller registers
 * which of its vertices/edges (caller's ids) other ranks need (send) and which halo entries other
 * ranks own (recv).  pack/unpack move them between the solver state and a contiguous DEVICE buffer
 * (layout: n_v x {x,w1,w2,xb,w1b,w2b} then n_e x {q1,q2,q3}, float32: only what changes -- the
 * receiver holds the data terms and weights of its halo vertices since its upload) that the caller
 * hands to RCCL (ncclSend/ncclRecv, e.g. torch.distributed P2P ops).  stream: hipStream_t or NULL
 * (= the handle's stream). */
int flame_hip_halo_register(flame_hip_graph* g, int32_t n_send_v, const int32_t* send_v,
                            int32_t n_send_e, const int32_t* send_e, int32_t n_recv_v,
                            const int32_t* recv_v, int32_t n_recv_e, const int32_t* recv_e);
int flame_hip_halo_bytes(const flame_hip_graph* g, int64_t* send_bytes, int64_t* recv_bytes);
int flame_hip_halo_pack(flame_hip_graph* g, void* send_buf_dev, void* stream);
int flame_hip_halo_unpack(flame_hip_graph* g, const void* recv_buf_dev, void* stream);
/* State snapshot / rollback (device-to-device, on `stream` or the handle's) and the resident tiles' give-up word, for a
 * caller that owns the recovery of several handles at once -- the partition layer below: a give-up of ONE part cannot
 * be repeated by that handle alone (its peers already hold records of the unfinished solve), so flame_hip_part_solve
 * snapshots every part in front of the solves it queues, and flame_hip_part_sync, when any rank reports a give-up, rolls
 * every part back and repeats those solves by launches.  take_error: call after synchronising the solve's stream;
 * *gave_up = 1 when a launch of resident tiles of this handle timed out since the last look (the word is cleared, the
 * device's lease backs off; nothing is repeated here). */
/* r06, for a transport that moves the halo records itself (the partition mode's PEER transport below): the handle's state
 * arrays and registered lists as DEVICE pointers -- A / B / q [2]: {x, w1, w2, z} / {x_bar, w1_bar, w2_bar, wgt} / {q1, q2, q3, -}
 * float4 per vertex / vertex / edge, indexed by the handle's INTERNAL ids; the registered lists come in those ids too (the
 * library translated them at flame_hip_halo_register), so a transport never needs the permutation; `cur` = the buffer that
 * holds the state now.  The call orders `stream` behind any state-writing work
 * of the handle's own stream; the pointers stay valid until the next upload / resize.  flame_hip_halo_written: the caller's
 * kernels on that stream have written halo state into the current buffers (what flame_hip_halo_unpack does itself). */
typedef struct {
  void* A[2];
  void* B[2];
  void* q[2];
  int32_t cur;
  int32_t n_send_v, n_send_e, n_recv_v, n_recv_e;
  const int32_t* send_v;
  const int32_t* send_e;
  const int32_t* recv_v;
  const int32_t* recv_e;
} flame_hip_halo_view;
int flame_hip_halo_view_get(flame_hip_graph* g, void* stream, flame_hip_halo_view* out);
int flame_hip_halo_written(flame_hip_graph* g);
int flame_hip_state_snapshot(flame_hip_graph* g, void* stream);
int flame_hip_state_rollback(flame_hip_graph* g, void* stream);
int flame_hip_persist_take_error(flame_hip_graph* g, int32_t* gave_up);

/* ---- partition mode without Python (BASELINE.json configs 4 / 5): ONE graph cut into world x
 * parts_per_rank subdomains by recursive coordinate bisection (METIS is not in the image), every rank solves its parts
 * with ordinary handles (resident tiles), and every halo_depth iterations the parts swap their halo records through
 * RCCL on one stream: flame_hip_halo_pack -> ncclGroupStart / ncclSend + ncclRecv per neighbouring part (a part of the
 * same rank is a send / receive of the rank with itself) / ncclGroupEnd -> flame_hip_halo_unpack.  Bit-identical to
 * the single-GPU result.  The reference has no counterpart (one CPU process, reference src/flame_offline_tum.cc:
 * 403-563).  One process per GPU; librccl.so is loaded at the first call (no link-time dependency).
 *   rank 0: flame_hip_comm_get_unique_id(id); hand `id` to every rank (MPI, a file, a socket ...);
 *   every rank: flame_hip_comm_create(&c, device, rank, world, id);
 *               flame_hip_part_create(&p, c, 0, 0, parts_per_rank, halo_depth, V, E, ... the WHOLE graph ...);
 *               flame_hip_part_solve(p, &params, iters);  flame_hip_part_costs(...);  flame_hip_part_gather(...).
 * Every rank derives all subdomains from the whole graph, so no request lists travel.  comm == NULL makes a
 * host-only plan of rank plan_rank of plan_world (tests: flame_hip_part_info / _array on it, no device, no RCCL). */
typedef struct flame_hip_comm flame_hip_comm;
typedef struct flame_hip_part flame_hip_part;
#define FLAME_HIP_COMM_ID_BYTES 128
int flame_hip_rccl_available(void); /* 1: librccl.so loads and exports every entry point this file needs */
int flame_hip_comm_get_unique_id(char id[FLAME_HIP_COMM_ID_BYTES]);
int flame_hip_comm_create(flame_hip_comm** out, int device, int rank, int world, const char id[FLAME_HIP_COMM_ID_BYTES]);
void flame_hip_comm_destroy(flame_hip_comm* c); /* destroy the parts built on it FIRST (a part keeps the pointer) */
void* flame_hip_comm_stream(flame_hip_comm* c); /* the hipStream_t everything of this communicator is ordered on */
int flame_hip_part_create(flame_hip_part** out, flame_hip_comm* comm, int32_t plan_rank, int32_t plan_world,
                          int32_t parts_per_rank, int32_t halo_depth, int32_t V, int32_t E, const float* pos,
                          const int32_t* edges, const float* alpha, const float* beta, const float* z,
                          const float* wgt, const float* x0 /* or NULL */);
void flame_hip_part_destroy(flame_hip_part* p);
/* num_iters more PD iterations, an exchange whenever the halo rings are used up; asynchronous on the communicator's
 * stream (no host synchronisation inside); successive calls continue on the rings the last one left */
int flame_hip_part_solve(flame_hip_part* p, const flame_hip_params* params, int32_t num_iters);
/* Waits for everything queued.  COLLECTIVE when world > 1 (one 4-byte ncclAllReduce): the ranks agree whether any launch
 * of resident tiles gave up since the last synchronising call; if so EVERY rank rolls its parts back to the snapshot taken
 * in front of the first solve since then and repeats those solves by ordinary launches (info "recovered" counts them) --
 * a give-up costs time, never the result.  _costs, _gather and _update_data synchronise through this call. */
int flame_hip_part_sync(flame_hip_part* p);
/* "rank", "world", "rccl_ranks" (what RCCL itself reports: ncclCommCount), "device" */
int flame_hip_comm_info(const flame_hip_comm* c, const char* key, int64_t* value);
/* new frame on the unchanged topology: data terms, weights, initial x of the WHOLE graph (x0 NULL = z); state reset */
int flame_hip_part_update_data(flame_hip_part* p, const float* z, const float* wgt, const float* x0);
/* the whole graph's cost terms: owned sums of every part + one ncclAllReduce of 2 doubles.  Synchronises. */
int flame_hip_part_costs(flame_hip_part* p, const flame_hip_params* params, double* smooth, double* data);
/* the whole solution on every rank (x, w1, w2: V; q: 3E interleaved; any may be NULL).  Synchronises. */
int flame_hip_part_gather(flame_hip_part* p, float* x, float* w1, float* w2, float* q);
/* r06 PEER TRANSPORT ("transport" 1; 0 = RCCL, the default and the contract's path): the records of an exchange are written
 * by ONE kernel of the sending rank straight into the receiving parts' inboxes -- uncached device memory of the same GPU, of
 * another process (hipIpc) or of another GPU of the node (peer access over xGMI) --, a per-message flag word takes the exchange's
 * epoch behind them (release, system scope), and ONE kernel of the receiving rank waits for its flags (bounded) and unpacks:
 * two launches per exchange whatever the number of parts and neighbours, no host synchronisation, no ncclGroup (35-70 us on
 * one GPU against ~20 us of resident compute per 16 iterations: profiles/r05_part_halo_depth.txt).  Two record buffers by
 * epoch parity (a sender can be at most one exchange ahead).  Same bits.  Ranks of ONE process / world 1 need nothing else;
 * ranks in different processes exchange their inbox handles once: every rank calls flame_hip_part_peer_blob, the application
 * gathers the blobs of all ranks in rank order (the library does it itself over RCCL when the communicator has one:
 * flame_hip_part_set_option "transport" 1 is then collective) and hands them to flame_hip_part_peer_connect.
 * flame_hip_comm_create_local: a communicator WITHOUT RCCL for that case (peer transport only; flame_hip_part_costs then
 * returns this rank's owned sums, and the parts solve by launches -- the give-up agreement of resident tiles is an all-reduce). */
#define FLAME_HIP_PEER_BLOB_BYTES 128
int flame_hip_comm_create_local(flame_hip_comm** out, int device, int rank, int world);
int flame_hip_part_peer_blob(flame_hip_part* p, char blob[FLAME_HIP_PEER_BLOB_BYTES]);
int flame_hip_part_peer_connect(flame_hip_part* p, const char* blobs /* world x FLAME_HIP_PEER_BLOB_BYTES */);
/* "transport" 0 / 1 (above); "time_exchanges" 0 / 1: HIP events around the next (up to 64) exchanges -- pack, the group of sends / receives, unpack;
 * "pipeline" (-1 = automatic, the default: on from 4 parts per rank; 0 / 1 force it; acts with parts_per_rank >= 2): inside a solve call the halo records of part i leave -- an ncclGroup of
 * their own on the communicator's second stream -- while part i + 1 iterates (overlap compute with the exchange, by
 * over-decomposition); info "exchanges_pipelined" counts them.  Same bits either way. */
int flame_hip_part_set_option(flame_hip_part* p, const char* key, int32_t value);
/* keys: "transport", "peer_connected", "num_parts", "parts_per_rank", "exchanges", "p2p_ops", "rings_left", "exchanges_timed", "exchange_ns" (mean device
 * time of the timed exchanges; synchronise first), "recovered" (solves repeated by launches after
 * a give-up of resident tiles on any rank), "persist" (the parts solve with resident tiles); per local part: "part_id",
 * "n_own", "n_ext", "e_loc", "num_peers", "send_bytes", "recv_bytes", "persist_used" (the part's LAST local solve was one launch
 * of resident tiles), "persist_launches" (how many were, so far) */
int flame_hip_part_info(const flame_hip_part* p, const char* key, int32_t local_part, int64_t* value);
/* int32 arrays: "part" (V: the part of every vertex), per local part "vid", "eid", "edges", "e_owned", "peers",
 * "send_v", "send_e", "recv_v", "recv_e", "send_cnt", "recv_cnt"; returns the element count or a negative error */
int64_t flame_hip_part_array(const flame_hip_part* p, const char* key, int32_t local_part, int32_t* out, int64_t cap);

/* Debug/test hook (no device needed; works on a handle created with device = -1): copies the
 * named host-side plan array ("v_o2i", "e_o2i", "grow", "ginc", "eij", "tiles", "t_vmap",
 * "t_emap", "t_eij", "t_srow", ...) and returns its element count, or a negative error. */
int64_t flame_hip_debug_plan_array(const flame_hip_graph* g, const char* name, void* buf,
                                   int64_t cap_bytes);

const char* flame_hip_strerror(int code);
/* Library/ABI version: major*10000 + minor*100 + patch. */
int flame_hip_version(void);

#ifdef __cplusplus
}
#endif
#endif
