// include/flame/params.h -- flame::Params as flame_ros fills it.
//
// Every field below is assigned from a rosparam by the reference frontends (reference
// src/flame_offline_tum.cc:158-249, src/flame_offline_asl.cc, src/flame_nodelet.cc:220-330);
// defaults are the values of cfg/flame_offline_tum.yaml.  Only the regulariser / triangle-filter
// fields drive the HIP path; the rest are carried so the frontends compile and can be forwarded to
// the upstream feature pipeline unchanged.
#pragma once

namespace flame {

namespace optimizers {
namespace nltgv2_l1_graph_regularizer {
struct Params {
  float data_factor = 0.15f;  // cfg/flame_offline_tum.yaml:93
  float step_x = 0.001f;      // :94
  float step_q = 125.0f;      // :95
  float theta = 0.25f;        // :96
  float x_min = 0.0f;         // idepth clamp after the prox ([UPSTREAM-RECALL] 0..10)
  float x_max = 10.0f;
};
}  // namespace nltgv2_l1_graph_regularizer
}  // namespace optimizers

struct InverseDepthFilterParams {  // params_.zparams (reference src/flame_offline_tum.cc:224,231)
  int win_size = 5;
  float epipolar_line_var = 4.0f;
};
struct FeatureDetectionParams {  // params_.fparams (:214,225)
  float min_grad_mag = 5.0f;
  int win_size = 5;
};

struct Params {
  // output / display filters (reference src/flame_offline_tum.cc:158-192, yaml :19-57)
  bool debug_quiet = false;
  float scene_color_scale = 1.0f;
  bool do_oblique_triangle_filter = true;
  float oblique_normal_thresh = 1.57f;
  float oblique_idepth_diff_factor = 0.35f;
  float oblique_idepth_diff_abs = 0.1f;
  bool do_edge_length_filter = true;
  float edge_length_thresh = 0.333f;
  bool do_idepth_triangle_filter = true;
  float min_triangle_idepth = 0.01f;
  // debug images (:195-202)
  bool debug_draw_wireframe = true, debug_draw_features = true, debug_draw_detections = false;
  bool debug_draw_matches = false, debug_draw_normals = false, debug_draw_idepthmap = true;
  bool debug_draw_text_overlay = true, debug_flip_images = false;
  // threading (:205-206)
  int omp_num_threads = 4, omp_chunk_size = 1024;
  // (this build's own, not a flame_ros key) host threads of the built-in Delaunay triangulator, the one CPU stage in
  // front of the GPU tail: a persistent pool (flame/utils/delaunay.h); 0 = omp_num_threads.  A host that feeds an
  // MI355X has the cores: 10 k points take 2.3 ms on 4 threads of the old fork-per-level scheme, 0.9 ms on 16 of the pool.
  int triangulate_threads = 16;
  // (this build's own) the built-in triangulation runs on the GPU (flame_hip_delaunay: one wavefront per feature builds the
  // feature's star from exact predicates; same contract as the host triangulator, which stays the choice when false).
  // A registered FrontEnd::triangulate takes precedence over both.
  bool triangulate_on_gpu = true;
  // features (:209-231)
  bool do_letterbox = false;
  float min_grad_mag = 5.0f;
  float min_error = 100.0f;
  int detection_win_size = 16;
  int max_dropouts = 5;
  InverseDepthFilterParams zparams;
  FeatureDetectionParams fparams;
  // regulariser (:234-249, yaml :84-99)
  // do_median_filter / do_lowpass_filter: YAML keys regularization/do_median_filter, do_lowpass_filter
  // (cfg/flame_offline_tum.yaml:85-86) that flame_ros never reads => upstream defaults (off); one
  // Jacobi pass of the graph filter on the regularised idepths (row a9)
  bool do_median_filter = false;
  bool do_lowpass_filter = false;
  bool do_nltgv2 = true;
  bool adaptive_data_weights = false;
  bool rescale_data = false;
  bool init_with_prediction = true;
  float idepth_var_max_graph = 0.01f;
  optimizers::nltgv2_l1_graph_regularizer::Params rparams;
  // min_height / max_height ("height of features that are added to graph", yaml :97-98) and
  // check_sticky_obstacles (:99) gate FEATURES, upstream of the variance gate, in the feature
  // pipeline that is not rebuilt here: carried for the frontends and a registered FrontEnd,
  // IGNORED by the GPU tail (the reference defaults +-1e14 / false disable them anyway).
  float min_height = -1e14f, max_height = 1e14f;
  bool check_sticky_obstacles = false;
  // [UPSTREAM-RECALL] switches (not in the reference; defaults = the build's statement, see
  // include/flame_hip.h flame_hip_sync_params and option "d_sign"): flipped when a state dump of a
  // real robustrobotics/flame build disagrees (tools/pin_upstream/)
  int edge_weight_rule = 0;
  float edge_alpha_gain = 0.0f, edge_beta_gain = 0.0f;
  int edge_d_sign = 1;
  // not in the reference: regulariser iterations per update (an upstream constant that no
  // flame_ros YAML key exposes) and the GPU to run on
  int nltgv2_iterations = 200;
  int hip_device = 0;
};

}  // namespace flame
