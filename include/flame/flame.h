// include/flame/flame.h -- flame::Flame as flame_ros consumes it, with the regulariser path on
// MI355X (libflame_hip.so).
//
// Every member flame_ros calls is declared here with the signature of its call site:
//   constructor                              reference src/flame_offline_tum.cc:408-412, src/flame_nodelet.cc:523-527
//   update(time, img_id, pose, gray, is_pf)  src/flame_offline_tum.cc:578-579, src/flame_nodelet.cc:634-635
//   update(..., idepths_true)                src/flame_offline_tum.cc:593-594
//   getInverseDepthMesh                      :628-635
//   getFilteredInverseDepthMap(cv::Mat1f*)   :643;  getInverseDepthMap()  src/flame_nodelet.cc:688
//   getRawIDepths                            :680-682
//   getDebugImage{Wireframe,Features,Detections,Matches,Normals,InverseDepthMap}   :731-766
//   stats()                                  :706-707
//   updatePoseFramePoses / prunePoseFrames   src/flame_nodelet.cc:474-475
//
// Upstream's update() = feature detection + epipolar idepth filtering + Delaunay triangulation
// (OpenCV/Sophus code upstream of the hot path, SURVEY.md 2 "OUT OF SCOPE") followed by the part
// this class runs on the GPU: graph sync (row a7), N x nltgv2 step (a2-a5), costs (a6), the
// per-triangle stage (a8), dense maps / mesh (f1, f2).  The feature pipeline plugs in through
// FrontEnd (tracked features of the frame; optionally the triangulation of the gated features, whose
// default is the library's exact Delaunay triangulation on the GPU, flame_hip_delaunay -- row f3's first leg --, or,
// with Params::triangulate_on_gpu = false, the exact host triangulator of utils/delaunay.h);
// update() returns false when no `track` is registered, exactly like any other failed update (the
// frontends warn and skip the frame, src/flame_offline_tum.cc:597-601).  updateGraph() is the GPU
// tail on its own, for callers that already hold features + triangulation.
//
// Conventions kept from the reference: update*() returns false on failure; every other method is
// void / returns an image with caller-owned outputs; nothing throws; an internal mutex serialises
// update against the pose-frame mutators the nodelet calls from another thread
// (src/flame_nodelet.cc:474-475; stat key update_locking, msg/FlameStats.msg:34).
#pragma once
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <functional>
#include <limits>
#include <mutex>
#include <utility>
#include <vector>

#include "../flame_hip.h"
#include "optimizers/nltgv2_l1_graph_regularizer.h"
#include "params.h"
#include "types.h"
#include "utils/assert.h"
#include "utils/delaunay.h"
#include "utils/image_utils.h"
#include "utils/stats_tracker.h"
#include "utils/visualization.h"

namespace flame {

// One camera frame as update() receives it.
struct FrameInput {
  double time = 0.0;
  uint32_t img_id = 0;
  SE3f pose;
  const Image1b* img = nullptr;
  bool is_poseframe = false;
  const Image1f* idepths_true = nullptr;  // analysis/pass_in_truth (src/flame_offline_tum.cc:577-595)
};

// Everything the feature pipeline knows about the tracked features of a frame.
struct FeatureSet {
  std::vector<Point2f> vtx;
  std::vector<float> idepth_mu, idepth_var;
  std::vector<float> prediction;  // empty, or one value per feature (NaN = none)
};

// The part of upstream's Flame::update() that is NOT on the hot path, as callbacks.
struct FrontEnd {
  // detection + tracking + epipolar idepth filtering: all features tracked in this frame
  std::function<bool(const FrameInput&, FeatureSet*)> track;
  // Delaunay triangulation of the features that passed the variance gate (optional: without it the
  // built-in exact divide-and-conquer triangulator of flame/utils/delaunay.h is used)
  std::function<bool(const std::vector<Point2f>&, std::vector<Triangle>*)> triangulate;
  // pose-frame bookkeeping (optional)
  std::function<void(const std::vector<uint32_t>&, const std::vector<SE3f>&)> updatePoseFramePoses;
  std::function<void(const std::vector<uint32_t>&)> prunePoseFrames;
};

class Flame {
 public:
  Flame(int width, int height, const Matrix3f& K, const Matrix3f& Kinv,
        const Params& params = Params())
      : width_(width), height_(height), params_(params) {
    toRowMajor(K, K_);
    toRowMajor(Kinv, Kinv_);
    const Vec3b black(0, 0, 0);
    debug_wireframe_.img = Image3b(height, width, black);
    debug_features_.img = Image3b(height, width, black);
    debug_detections_ = Image3b(height, width, black);
    debug_matches_ = Image3b(height, width, black);
    debug_normals_.img = Image3b(height, width, black);
    debug_idepthmap_.img = Image3b(height, width, black);
  }
  Flame(const Flame&) = delete;
  Flame& operator=(const Flame&) = delete;

  void setFrontEnd(const FrontEnd& fe) {
    std::lock_guard<std::mutex> lock(mtx_);
    frontend_ = fe;
  }

  // reference src/flame_offline_tum.cc:578-579, src/flame_nodelet.cc:634-635
  bool update(double time, uint32_t img_id, const SE3f& pose, const Image1b& img, bool is_poseframe) {
    FrameInput in;
    in.time = time; in.img_id = img_id; in.pose = pose; in.img = &img; in.is_poseframe = is_poseframe;
    return updateFrame(in);
  }
  // reference src/flame_offline_tum.cc:593-594 (ground-truth idepths passed in)
  bool update(double time, uint32_t img_id, const SE3f& pose, const Image1b& img, bool is_poseframe,
              const Image1f& idepths_true) {
    FrameInput in;
    in.time = time; in.img_id = img_id; in.pose = pose; in.img = &img; in.is_poseframe = is_poseframe;
    in.idepths_true = &idepths_true;
    return updateFrame(in);
  }

  // The GPU tail of update(): `vtx` are the features that passed the variance gate
  // (idepth_var_max_graph; a feature that fails it makes the update fail), `idepth_mu` /
  // `idepth_var` their filtered inverse depths, `triangles` their Delaunay triangulation;
  // `prediction` (optional, NaN = none) initialises x when init_with_prediction is set.  Returns
  // false on any error (stats key "hip_error" holds the flame_hip code); the results of the last
  // successful update stay readable.
  bool updateGraph(double time, uint32_t img_id, const std::vector<Point2f>& vtx,
                   const std::vector<float>& idepth_mu, const std::vector<float>& idepth_var,
                   const std::vector<Triangle>& triangles,
                   const std::vector<float>* prediction = nullptr) {
    stats_.tick("update_locking");
    std::lock_guard<std::mutex> lock(mtx_);
    stats_.tock("update_locking");
    stats_.tick("update");
    const bool ok = updateGraphLocked(time, img_id, vtx, idepth_mu, idepth_var, triangles, prediction, nullptr);
    stats_.tock("update");
    return ok;
  }

  // Caller owns the vectors (reference src/flame_offline_tum.cc:628-635).
  void getInverseDepthMesh(std::vector<Point2f>* vtx, std::vector<float>* idepths,
                           std::vector<Vector3f>* normals, std::vector<Triangle>* triangles,
                           std::vector<bool>* tri_validity, std::vector<Edge>* edges) const {
    std::lock_guard<std::mutex> lock(mtx_);
    if (vtx) *vtx = vtx_;
    if (idepths) *idepths = idepths_;
    if (normals) {
      normals->resize(vtx_.size());
      for (size_t v = 0; v < vtx_.size(); ++v)
        for (int k = 0; k < 3; ++k) (*normals)[v](k) = normals_flat_[3 * v + k];
    }
    if (triangles) *triangles = tris_;
    if (tri_validity) tri_validity->assign(tri_valid_.begin(), tri_valid_.end());
    if (edges) *edges = edges_;
  }

  // reference src/flame_offline_tum.cc:680-682: every tracked feature, before the variance gate
  void getRawIDepths(std::vector<Point2f>* vtx, std::vector<float>* mu, std::vector<float>* var) const {
    std::lock_guard<std::mutex> lock(mtx_);
    if (vtx) *vtx = raw_vtx_;
    if (mu) *mu = raw_mu_;
    if (var) *var = raw_var_;
  }

  // Dense maps (row f2), NaN where the mesh does not cover the pixel.  The Image1f forms are the
  // reference's (cv::Mat1f when OpenCV is present: src/flame_offline_tum.cc:643,
  // src/flame_nodelet.cc:688); on failure the image is all NaN.
  void getFilteredInverseDepthMap(Image1f* idepthmap) const { mapImage(1, idepthmap); }
  Image1f getInverseDepthMap() const {
    Image1f m;
    mapImage(0, &m);
    return m;
  }
  bool getFilteredInverseDepthMap(std::vector<float>* idepthmap) const { return maps(1, idepthmap, nullptr, nullptr, 0.f, 0.f); }
  bool getInverseDepthMap(std::vector<float>* idepthmap) const { return maps(0, idepthmap, nullptr, nullptr, 0.f, 0.f); }
  // idepth -> depth inversion and point cloud of flame_ros (reference
  // src/flame_offline_tum.cc:650-661, src/utils.cc:290-312), computed on the GPU as well.
  bool getDepthMapAndCloud(std::vector<float>* depthmap, std::vector<float>* cloud_xyz,
                           float min_depth, float max_depth) const {
    return maps(1, nullptr, depthmap, cloud_xyz, min_depth, max_depth);
  }
  // Mesh as flame_ros publishes it (row f1; reference src/utils.cc:184-230): 12 floats per
  // vertex in flame_ros::PointNormalUV layout and reversed-winding faces of the valid triangles.
  bool getMeshPointNormalUV(std::vector<float>* points, std::vector<int32_t>* faces) const {
    std::lock_guard<std::mutex> lock(mtx_);
    if (!device_frame_valid_) return false;
    const flame_hip_tri_params tp = triParams();
    points->assign(12 * vtx_.size(), 0.f);
    faces->assign(3 * tris_.size(), 0);
    int32_t nf = 0;
    if (flame_hip_mesh(graph_.handle(), Kinv_, &tp, points->data(), faces->data(), &nf)) return false;
    faces->resize(3 * static_cast<size_t>(nf));
    return true;
  }

  // Debug images, BGR8, width x height (reference src/flame_offline_tum.cc:731-766; what each
  // shows: cfg/flame_offline_tum.yaml:58-64).  Wireframe: sides of the valid triangles coloured by
  // idepth; Features: the raw features coloured by idepth; Normals: "image colored by interpolated
  // normal vectors"; InverseDepthMap: jet colormap of the filtered dense idepthmap; all on black
  // (the input image is not kept).  They are rendered ON THE GPU (flame_hip_debug_image) and only
  // when a getter asks: update() itself draws nothing -- the first call of a getter after an
  // update renders that image from the frame's device state and copies it out, later calls return
  // the cached image.  A disabled image (Params::debug_draw_*) stays black.  Detections / Matches
  // belong to the feature pipeline: black images of the right size.  debug_flip_images rotates the
  // rendered images by 180 degrees (cfg/flame_offline_tum.yaml:65); debug_draw_text_overlay is not
  // applied (no font rendering here).
  const Image3b& getDebugImageWireframe() const {
    return debugImage(FLAME_HIP_IMG_WIREFRAME, params_.debug_draw_wireframe, &debug_wireframe_);
  }
  const Image3b& getDebugImageFeatures() const {
    return debugImage(FLAME_HIP_IMG_FEATURES, params_.debug_draw_features, &debug_features_);
  }
  const Image3b& getDebugImageDetections() const { return debug_detections_; }
  const Image3b& getDebugImageMatches() const { return debug_matches_; }
  const Image3b& getDebugImageNormals() const {
    return debugImage(FLAME_HIP_IMG_NORMALS, params_.debug_draw_normals, &debug_normals_);
  }
  const Image3b& getDebugImageInverseDepthMap() const {
    return debugImage(FLAME_HIP_IMG_IDEPTHMAP, params_.debug_draw_idepthmap, &debug_idepthmap_);
  }

  // reference src/flame_nodelet.cc:474-475 (called from the ROS callback thread): forwarded to the
  // front end under the same mutex update() holds.
  void updatePoseFramePoses(const std::vector<uint32_t>& pf_ids, const std::vector<SE3f>& pf_poses) {
    std::lock_guard<std::mutex> lock(mtx_);
    if (frontend_.updatePoseFramePoses) frontend_.updatePoseFramePoses(pf_ids, pf_poses);
  }
  void prunePoseFrames(const std::vector<uint32_t>& pf_ids) {
    std::lock_guard<std::mutex> lock(mtx_);
    if (frontend_.prunePoseFrames) frontend_.prunePoseFrames(pf_ids);
  }

  const utils::StatsTracker& stats() const { return stats_; }
  const Params& params() const { return params_; }
  int width() const { return width_; }
  int height() const { return height_; }

 private:
  bool updateFrame(const FrameInput& in) {
    stats_.tick("update_locking");
    std::lock_guard<std::mutex> lock(mtx_);
    stats_.tock("update_locking");
    stats_.tick("update");
    bool ok = false;
    if (!frontend_.track) {
      stats_.set("hip_error", FLAME_HIP_ERR_STATE);  // no feature pipeline registered
    } else {
      FeatureSet fs;
      stats_.tick("update_idepths");
      ok = frontend_.track(in, &fs) && fs.idepth_mu.size() == fs.vtx.size() &&
           fs.idepth_var.size() == fs.vtx.size() &&
           (fs.prediction.empty() || fs.prediction.size() == fs.vtx.size());
      stats_.tock("update_idepths");
      if (ok) {
        // variance gate (row a7): "Maximum idepth var before feature can be added to graph"
        // (reference cfg/flame_offline_tum.yaml:92)
        const int32_t n = static_cast<int32_t>(fs.vtx.size());
        std::vector<uint8_t> keep(fs.vtx.size());
        const int32_t nk = flame_hip_feature_gate(n, fs.idepth_var.data(), params_.idepth_var_max_graph, keep.data());
        ok = nk >= 0;
        FeatureSet gated;
        if (ok && nk != n) {  // (every feature through the gate -- the usual frame -- is used where it lies)
          gated.vtx.reserve(nk); gated.idepth_mu.reserve(nk); gated.idepth_var.reserve(nk);
          if (!fs.prediction.empty()) gated.prediction.reserve(nk);
          for (int32_t v = 0; v < n; ++v)
            if (keep[v]) {
              gated.vtx.push_back(fs.vtx[v]);
              gated.idepth_mu.push_back(fs.idepth_mu[v]);
              gated.idepth_var.push_back(fs.idepth_var[v]);
              if (!fs.prediction.empty()) gated.prediction.push_back(fs.prediction[v]);
            }
        }
        const FeatureSet& g = (ok && nk != n) ? gated : fs;
        std::vector<Triangle> tris;
        tris.swap(tri_buf_);  // (last frame's list: its storage -- and, on the GPU branch, its elements -- are reused:
                              // resizing a fresh vector to 2 V triangles would clear 2 V triangles first, every frame)
        if (ok) {
          stats_.tick("triangulate");
          // (the reference's `omp_num_threads`, cfg/flame_offline_tum.yaml:70, is what its CPU stages run on)
          if (frontend_.triangulate) {
            tris.clear();
            ok = frontend_.triangulate(g.vtx, &tris);
          } else if (params_.triangulate_on_gpu) {  // row f3's first leg in the library
            static_assert(sizeof(Point2f) == 2 * sizeof(float) && sizeof(Triangle) == 3 * sizeof(int32_t), "boundary types are packed");
            const int32_t nv = static_cast<int32_t>(g.vtx.size());
            int32_t nt = 0;
            // KEEP mode (r05): only the triangle count comes back; the list stays on the device for the graph sync and is
            // fetched while the GPU iterates (updateGraphLocked).  `tris` only carries the count until then.
            const int rc = graph_.triangulate(params_.hip_device, nv, nv ? reinterpret_cast<const float*>(g.vtx.data()) : nullptr,
                                              0, nullptr, &nt);
            tris.resize(rc ? 0 : nt);
            ok = rc == 0 && nt > 0;
            tris_in_library_ = ok;  // (nothing to triangulate -- fewer than three distinct points, or all on a line -- fails the
                                     // frame, as the host triangulator's `false` does)
            if (rc) stats_.set("hip_error", rc);
          } else {
            tris.clear();
            ok = delaunay_.triangulate(g.vtx, &tris, params_.triangulate_threads > 0 ? params_.triangulate_threads : params_.omp_num_threads);
          }
          stats_.tock("triangulate");
        } else {
          tris.clear();
        }
        if (ok)
          ok = updateGraphLocked(in.time, in.img_id, g.vtx, g.idepth_mu, g.idepth_var, tris,
                                 g.prediction.empty() ? nullptr : &g.prediction, &fs);
        tris_in_library_ = false;  // (also when the graph update was not reached)
        tri_buf_.swap(tris);
      }
    }
    stats_.tock("update");
    return ok;
  }

  bool updateGraphLocked(double time, uint32_t img_id, const std::vector<Point2f>& vtx,
                         const std::vector<float>& idepth_mu, const std::vector<float>& idepth_var,
                         const std::vector<Triangle>& triangles, const std::vector<float>* prediction,
                         const FeatureSet* raw) {
    namespace reg = optimizers::nltgv2_l1_graph_regularizer;
    (void)time;
    (void)img_id;
    const int32_t V = static_cast<int32_t>(vtx.size()), T = static_cast<int32_t>(triangles.size());
    // (read and cleared before anything can return -- a frame that fails its size check must not leave the flag
    // set for a later updateGraph() with caller-supplied triangles, which would then silently use the library's old list)
    const bool tris_in_library = tris_in_library_;
    tris_in_library_ = false;
    if (idepth_mu.size() != vtx.size() || idepth_var.size() != vtx.size() ||
        (prediction && prediction->size() != vtx.size()))
      return fail(FLAME_HIP_ERR_ARG);
    device_frame_valid_ = false;  // the device state follows this frame from here on

    // ---- graph sync (row a7), in the library ----
    stats_.tick("sync_graph");
    // cv::Point2f / cv::Vec3i (and the fallback structs) ARE the library's input layout -- {float u, v}
    // per vertex, three int32 vertex ids per triangle: no conversion pass over the frame
    static_assert(sizeof(Point2f) == 2 * sizeof(float) && sizeof(Triangle) == 3 * sizeof(int32_t) &&
                      sizeof(Edge) == 2 * sizeof(int32_t), "boundary types are packed");
    const float* pos = V ? reinterpret_cast<const float*>(vtx.data()) : nullptr;
    // (a list flame_hip_delaunay made for this very frame is read where the library still holds it)
    const int32_t* tidx = (T && !tris_in_library) ? reinterpret_cast<const int32_t*>(triangles.data()) : nullptr;
    flame_hip_sync_params sp;
    sp.adaptive_data_weights = params_.adaptive_data_weights;
    sp.rescale_data = params_.rescale_data;
    sp.init_with_prediction = params_.init_with_prediction;
    sp.idepth_var_max_graph = params_.idepth_var_max_graph;
    sp.edge_weight_rule = params_.edge_weight_rule;
    sp.alpha_gain = params_.edge_alpha_gain;
    sp.beta_gain = params_.edge_beta_gain;
    float scale = 1.0f;
    int rc = graph_.sync(params_.hip_device, sp, V, T, pos, idepth_mu.data(), idepth_var.data(), tidx,
                         prediction ? prediction->data() : nullptr, &scale, params_.edge_d_sign);
    if (rc) return fail(rc);
    const int32_t E = graph_.numEdges();
    stats_.tock("sync_graph");

    // ---- regulariser (rows a2-a6) ----
    stats_.tick("nltgv2");
    if (params_.do_nltgv2) {
      rc = reg::step(params_.rparams, &graph_, params_.nltgv2_iterations, /*wait=*/false);
      if (rc) return fail(rc);
    }
    // ---- optional graph filters (row a9; regularization/do_median_filter, do_lowpass_filter,
    // cfg/flame_offline_tum.yaml:85-86; timing keys src/utils.cc:155-156), on the regularised idepths ----
    if (params_.do_median_filter) {
      stats_.tick("median_filter");
      rc = flame_hip_graph_filter(graph_.handle(), 0, 1);
      if (!rc) rc = flame_hip_sync(graph_.handle());
      stats_.tock("median_filter");
      if (rc) return fail(rc);
    }
    if (params_.do_lowpass_filter) {
      stats_.tick("lowpass_filter");
      rc = flame_hip_graph_filter(graph_.handle(), 1, 1);
      if (!rc) rc = flame_hip_sync(graph_.handle());
      stats_.tock("lowpass_filter");
      if (rc) return fail(rc);
    }
    // ---- costs (a6), idepths back in the caller's units, per-triangle stage (a8), edge list: one
    // library call, one synchronisation.  Everything downstream (triangle filters, mesh, maps) works
    // on un-scaled inverse depths with the un-scaled thresholds. ----
    const flame_hip_params cp = reg::toC(params_.rparams);
    double smooth = 0.0, data = 0.0;
    // Host-side copies of the frame (what the getters hand out) are made WHILE the GPU iterates: the
    // solve above is asynchronous, the call below is the frame's one synchronisation.  They go into
    // staging members and are swapped in only when the frame succeeded (atomic commit).
    st_vtx_ = vtx;
    if (tris_in_library) {  // the list flame_hip_delaunay kept: its host copy has been travelling since; taken over here
      st_tris_.resize(triangles.size());
      rc = graph_.triangleList(T, T ? reinterpret_cast<int32_t*>(st_tris_.data()) : nullptr);
      if (rc) return fail(rc);
    } else {
      st_tris_ = triangles;
    }
    if (raw) { st_raw_vtx_ = raw->vtx; st_raw_mu_ = raw->idepth_mu; st_raw_var_ = raw->idepth_var; }
    else { st_raw_vtx_ = vtx; st_raw_mu_ = idepth_mu; st_raw_var_ = idepth_var; }
    st_edges_.resize(E);
    st_idepths_.resize(V);
    st_normals_.resize(3 * static_cast<size_t>(V));
    st_tri_valid_.resize(T);
    const flame_hip_tri_params tp = triParams();
    float coverage = 0.0f;
    rc = flame_hip_frame_results(graph_.handle(), &cp, scale, Kinv_, &tp, &smooth, &data, st_idepths_.data(),
                                 st_normals_.data(), st_tri_valid_.data(),
                                 E ? reinterpret_cast<int32_t*>(st_edges_.data()) : nullptr, &coverage);
    if (rc) return fail(rc);
    stats_.tock("nltgv2");
    stats_.setTiming("interpolate", 0.0);  // folded into the call above (device time: see nltgv2_device)

    // ---- commit: every cached output changes together ----
    vtx_.swap(st_vtx_);
    tris_.swap(st_tris_);
    idepths_.swap(st_idepths_);
    normals_flat_.swap(st_normals_);
    tri_valid_.swap(st_tri_valid_);
    edges_.swap(st_edges_);
    raw_vtx_.swap(st_raw_vtx_); raw_mu_.swap(st_raw_mu_); raw_var_.swap(st_raw_var_);
    device_frame_valid_ = true;
    ++frame_serial_;  // the debug images of earlier frames are stale (rendered on demand, see debugImage)

    // ---- stats (keys read at reference src/utils.cc:117-156) ----
    stats_.set("num_feats", static_cast<double>(raw_vtx_.size()));
    stats_.set("num_vtx", V);
    stats_.set("num_tris", T);
    stats_.set("num_edges", E);
    stats_.set("coverage", coverage);  // share of the image the filtered dense map covers (src/utils.cc:122)
    stats_.set("nltgv2_total_smoothness_cost", smooth);
    stats_.set("nltgv2_avg_smoothness_cost", V ? smooth / V : 0.0);
    stats_.set("nltgv2_total_data_cost", data);
    stats_.set("nltgv2_avg_data_cost", V ? data / V : 0.0);
    stats_.set("nltgv2_iters", params_.do_nltgv2 ? params_.nltgv2_iterations : 0);
    stats_.set("nltgv2_data_scale", scale);
    stats_.set("hip_error", 0);
    float ms = 0.f;
    int32_t launches = 0;
    if (params_.do_nltgv2 && flame_hip_last_solve_ms(graph_.handle(), &ms, &launches) == 0)
      stats_.setTiming("nltgv2_device", ms);
    // resident tiles (include/flame_hip.h, option "persist"): did this frame's solve run as one launch, and how many
    // solves of this object had to be repeated by launches because a launch gave up (contention for the chip)
    int64_t iv = 0;
    if (flame_hip_get_info(graph_.handle(), "persist_used", &iv) == 0) stats_.set("persist_used", static_cast<double>(iv));
    if (flame_hip_get_info(graph_.handle(), "persist_recovered", &iv) == 0) stats_.set("persist_recovered", static_cast<double>(iv));
    return true;
  }

  flame_hip_tri_params triParams() const {
    flame_hip_tri_params tp;
    tp.do_oblique_triangle_filter = params_.do_oblique_triangle_filter;
    tp.oblique_normal_thresh = params_.oblique_normal_thresh;
    tp.oblique_idepth_diff_factor = params_.oblique_idepth_diff_factor;
    tp.oblique_idepth_diff_abs = params_.oblique_idepth_diff_abs;
    tp.do_edge_length_filter = params_.do_edge_length_filter;
    tp.edge_length_thresh = params_.edge_length_thresh;
    tp.do_idepth_triangle_filter = params_.do_idepth_triangle_filter;
    tp.min_triangle_idepth = params_.min_triangle_idepth;
    tp.width = width_;
    tp.height = height_;
    return tp;
  }
  bool maps(int filtered, std::vector<float>* idm, std::vector<float>* dm, std::vector<float>* cloud,
            float min_depth, float max_depth) const {
    std::lock_guard<std::mutex> lock(mtx_);
    return mapsLocked(filtered, idm, dm, cloud, min_depth, max_depth);
  }
  bool mapsLocked(int filtered, std::vector<float>* idm, std::vector<float>* dm, std::vector<float>* cloud,
                  float min_depth, float max_depth) const {
    if (!device_frame_valid_) return false;
    const flame_hip_tri_params tp = triParams();
    const size_t n = static_cast<size_t>(width_) * height_;
    if (idm) idm->assign(n, 0.f);
    if (dm) dm->assign(n, 0.f);
    if (cloud) cloud->assign(3 * n, 0.f);
    return flame_hip_depthmaps(graph_.handle(), Kinv_, &tp, filtered, min_depth, max_depth,
                               idm ? idm->data() : nullptr, dm ? dm->data() : nullptr,
                               cloud ? cloud->data() : nullptr) == 0;
  }
  void mapImage(int filtered, Image1f* out) const {
    std::vector<float> v;
    const bool ok = maps(filtered, &v, nullptr, nullptr, 0.f, 0.f);
    *out = Image1f(height_, width_, std::numeric_limits<float>::quiet_NaN());
    if (!ok) return;
    for (int i = 0; i < height_; ++i)  // (rows of both image types are contiguous)
      std::memcpy(static_cast<void*>(&(*out)(i, 0)), v.data() + static_cast<size_t>(i) * width_, sizeof(float) * width_);
  }
  bool fail(int code) {
    stats_.set("hip_error", code);
    stats_.tock("sync_graph");  // close whatever timer the failed stage left open (no-op otherwise)
    stats_.tock("nltgv2");
    stats_.tock("interpolate");
    return false;
  }

  // ---- debug images: rendered by the library on the device, on demand ----
  struct DebugImage {
    Image3b img;
    uint64_t serial = 0;  // frame_serial_ the image was rendered for
  };
  const Image3b& debugImage(int kind, bool enabled, DebugImage* d) const {
    std::lock_guard<std::mutex> lock(mtx_);
    if (!enabled || !device_frame_valid_ || d->serial == frame_serial_) return d->img;
    const flame_hip_tri_params tp = triParams();
    dbg_buf_.resize(3 * static_cast<size_t>(width_) * height_);
    const bool feats = kind == FLAME_HIP_IMG_FEATURES;
    if (feats) {
      dbg_fpos_.resize(2 * raw_vtx_.size());
      for (size_t v = 0; v < raw_vtx_.size(); ++v) { dbg_fpos_[2 * v] = raw_vtx_[v].x; dbg_fpos_[2 * v + 1] = raw_vtx_[v].y; }
    }
    if (flame_hip_debug_image(graph_.handle(), kind, Kinv_, &tp, params_.scene_color_scale,
                              feats ? static_cast<int32_t>(raw_vtx_.size()) : 0, feats ? dbg_fpos_.data() : nullptr,
                              feats ? raw_mu_.data() : nullptr, dbg_buf_.data()))
      return d->img;  // the previous image stays
    static_assert(sizeof(Vec3b) == 3, "BGR8 pixels are packed");
    if (params_.debug_flip_images) {  // "Rotated debug images by 180 degrees for display" (yaml :65)
      uint8_t* b = dbg_buf_.data();
      const size_t n = static_cast<size_t>(width_) * height_;
      for (size_t lo = 0, hi = n - 1; lo < hi; ++lo, --hi)
        for (int c = 0; c < 3; ++c) std::swap(b[3 * lo + c], b[3 * hi + c]);
    }
    for (int i = 0; i < height_; ++i)  // rows of both image types are contiguous
      std::memcpy(static_cast<void*>(&d->img(i, 0)), dbg_buf_.data() + 3 * static_cast<size_t>(i) * width_,
                  3 * static_cast<size_t>(width_));
    d->serial = frame_serial_;
    return d->img;
  }

  int width_, height_;
  Params params_;
  float K_[9], Kinv_[9];
  mutable std::mutex mtx_;
  utils::StatsTracker stats_;
  FrontEnd frontend_;
  std::vector<Triangle> tri_buf_;         // storage of the frame's triangle list, kept across frames
  bool tris_in_library_ = false;          // the list in hand is the one flame_hip_delaunay just made on graph_'s handle
  utils::DelaunayTriangulator delaunay_;  // the built-in triangulation on the host (Params::triangulate_on_gpu = false; scratch kept across frames)
  optimizers::nltgv2_l1_graph_regularizer::Graph graph_;
  bool device_frame_valid_ = false;  // the device state belongs to the committed frame
  std::vector<Point2f> vtx_, raw_vtx_;
  std::vector<float> raw_mu_, raw_var_, idepths_, normals_flat_;
  std::vector<Triangle> tris_;
  std::vector<Edge> edges_;
  std::vector<uint8_t> tri_valid_;
  // staging of the frame in flight (filled while the GPU iterates, swapped in on success)
  std::vector<Point2f> st_vtx_, st_raw_vtx_;
  std::vector<float> st_raw_mu_, st_raw_var_, st_idepths_, st_normals_;
  std::vector<Triangle> st_tris_;
  std::vector<Edge> st_edges_;
  std::vector<uint8_t> st_tri_valid_;
  uint64_t frame_serial_ = 0;        // successful updates so far
  Image3b debug_detections_, debug_matches_;
  mutable DebugImage debug_wireframe_, debug_features_, debug_normals_, debug_idepthmap_;
  mutable std::vector<uint8_t> dbg_buf_;
  mutable std::vector<float> dbg_fpos_;
};

}  // namespace flame
