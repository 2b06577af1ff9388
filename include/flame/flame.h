// include/flame/flame.h -- flame::Flame, reduced to the regulariser path, on MI355X.
//
// What flame_ros calls: the constructor (reference src/flame_offline_tum.cc:
// 408-412), update() (:578-579), getInverseDepthMesh() (:628-635), getRawIDepths() (:680-682),
// stats() (:706-707).  Upstream's update() = feature detection + epipolar idepth filtering +
// Delaunay triangulation (OpenCV/Sophus code, out of scope here) followed by the part this class
// implements on the GPU: graph sync (row a7), N x nltgv2 step (a2-a5), costs (a6), per-triangle
// stage (a8).  That tail is exposed as updateGraph(); INTEGRATION.md shows the three-line change
// that makes upstream's update() call it.
//
// Conventions kept from the reference: update*() returns false on failure and the caller skips
// the frame (src/flame_offline_tum.cc:597-601); every other method is void with caller-owned
// output vectors; nothing throws; an internal mutex serialises update against the pose-frame
// mutators the nodelet calls from another thread (src/flame_nodelet.cc:474-475).
#pragma once
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <mutex>
#include <utility>
#include <vector>

#include "../flame_hip.h"
#include "optimizers/nltgv2_l1_graph_regularizer.h"
#include "params.h"
#include "types.h"
#include "utils/stats_tracker.h"

namespace flame {

class Flame {
 public:
  Flame(int width, int height, const Matrix3f& K, const Matrix3f& Kinv,
        const Params& params = Params())
      : width_(width), height_(height), params_(params) {
    toRowMajor(K, K_);
    toRowMajor(Kinv, Kinv_);
  }
  Flame(const Flame&) = delete;
  Flame& operator=(const Flame&) = delete;

  // The tail of upstream's update(): `vtx` are the tracked features that passed the variance gate
  // (idepth_var_max_graph), `idepth_mu` / `idepth_var` their filtered inverse depths,
  // `triangles` their Delaunay triangulation; `prediction` (optional) initialises x when
  // init_with_prediction is set.  Returns false on any error (stats key "hip_error" holds the
  // flame_hip code).
  bool updateGraph(double time, uint32_t img_id, const std::vector<Point2f>& vtx,
                   const std::vector<float>& idepth_mu, const std::vector<float>& idepth_var,
                   const std::vector<Triangle>& triangles,
                   const std::vector<float>* prediction = nullptr) {
    namespace reg = optimizers::nltgv2_l1_graph_regularizer;
    stats_.tick("update_locking");
    std::lock_guard<std::mutex> lock(mtx_);
    stats_.tock("update_locking");
    stats_.tick("update");
    (void)time;
    (void)img_id;
    const int32_t V = static_cast<int32_t>(vtx.size()), T = static_cast<int32_t>(triangles.size());
    if (idepth_mu.size() != vtx.size() || idepth_var.size() != vtx.size() ||
        (prediction && prediction->size() != vtx.size()))
      return fail(FLAME_HIP_ERR_ARG);

    // ---- graph sync (row a7) ----
    stats_.tick("sync_graph");
    vtx_ = vtx;
    mu_ = idepth_mu;
    var_ = idepth_var;
    tris_ = triangles;
    std::vector<std::pair<int32_t, int32_t> > und;
    und.reserve(3 * static_cast<size_t>(T));
    for (int32_t t = 0; t < T; ++t)
      for (int k = 0; k < 3; ++k) {
        int32_t a = triangles[t][k], b = triangles[t][(k + 1) % 3];
        if (a < 0 || b < 0 || a >= V || b >= V || a == b) return fail(FLAME_HIP_ERR_ARG);
        und.push_back(a < b ? std::make_pair(a, b) : std::make_pair(b, a));
      }
    std::sort(und.begin(), und.end());
    und.erase(std::unique(und.begin(), und.end()), und.end());
    const int32_t E = static_cast<int32_t>(und.size());
    edges_.resize(E);
    std::vector<float> pos(2 * static_cast<size_t>(V)), alpha(E), z(V), wgt(V), x0(V);
    std::vector<int32_t> eidx(2 * static_cast<size_t>(E)), tidx(3 * static_cast<size_t>(T));
    for (int32_t v = 0; v < V; ++v) { pos[2 * v] = vtx[v].x; pos[2 * v + 1] = vtx[v].y; }
    for (int32_t e = 0; e < E; ++e) {
      edges_[e] = Edge(und[e].first, und[e].second);
      eidx[2 * e] = und[e].first;
      eidx[2 * e + 1] = und[e].second;
      const float dx = pos[2 * und[e].first] - pos[2 * und[e].second];
      const float dy = pos[2 * und[e].first + 1] - pos[2 * und[e].second + 1];
      alpha[e] = 1.0f / std::sqrt(dx * dx + dy * dy);  // [UPSTREAM-RECALL] reciprocal edge length
    }
    for (int32_t t = 0; t < T; ++t)
      for (int k = 0; k < 3; ++k) tidx[3 * t + k] = triangles[t][k];
    float scale = 1.0f;
    if (params_.rescale_data && V > 0) {  // "Rescale data to have mean 1" (yaml :90)
      double s = 0.0;
      for (int32_t v = 0; v < V; ++v) s += idepth_mu[v];
      scale = static_cast<float>(s / V);
      if (!(scale > 0.0f)) scale = 1.0f;
    }
    for (int32_t v = 0; v < V; ++v) {
      z[v] = idepth_mu[v] / scale;
      wgt[v] = params_.adaptive_data_weights ? 1.0f / idepth_var[v] : 1.0f;  // yaml :89
      x0[v] = (params_.init_with_prediction && prediction) ? (*prediction)[v] / scale : z[v];
    }
    int rc = graph_.build(params_.hip_device, V, E, T, pos.data(), eidx.data(), alpha.data(),
                          alpha.data(), z.data(), wgt.data(), x0.data(), T ? tidx.data() : nullptr);
    stats_.tock("sync_graph");
    if (rc) return fail(rc);

    // ---- regulariser (rows a2-a6) ----
    stats_.tick("nltgv2");
    if (params_.do_nltgv2) {
      rc = reg::step(params_.rparams, &graph_, params_.nltgv2_iterations);
      if (rc) return fail(rc);
    }
    stats_.tock("nltgv2");
    idepths_.assign(V, 0.0f);
    rc = flame_hip_download(graph_.handle(), idepths_.data(), nullptr, nullptr, nullptr);
    if (rc) return fail(rc);
    const flame_hip_params cp = reg::toC(params_.rparams);
    double smooth = 0.0, data = 0.0;
    rc = flame_hip_costs(graph_.handle(), &cp, &smooth, &data);
    if (rc) return fail(rc);

    // ---- per-triangle stage (row a8) ----
    stats_.tick("interpolate");
    normals_flat_.assign(3 * static_cast<size_t>(V), 0.0f);
    tri_valid_.assign(T, 0);
    flame_hip_tri_params tp;
    tp.do_oblique_triangle_filter = params_.do_oblique_triangle_filter;
    tp.oblique_normal_thresh = params_.oblique_normal_thresh;
    tp.oblique_idepth_diff_factor = params_.oblique_idepth_diff_factor;
    tp.oblique_idepth_diff_abs = params_.oblique_idepth_diff_abs;
    tp.do_edge_length_filter = params_.do_edge_length_filter;
    tp.edge_length_thresh = params_.edge_length_thresh;
    tp.do_idepth_triangle_filter = params_.do_idepth_triangle_filter;
    tp.min_triangle_idepth = params_.min_triangle_idepth / scale;
    tp.width = width_;
    tp.height = height_;
    if (T > 0) {
      rc = flame_hip_triangles(graph_.handle(), Kinv_, &tp, normals_flat_.data(), tri_valid_.data(), nullptr);
      if (rc) return fail(rc);
    }
    stats_.tock("interpolate");
    for (int32_t v = 0; v < V; ++v) idepths_[v] *= scale;

    // ---- stats (keys read at reference src/utils.cc:117-156) ----
    stats_.set("num_feats", V);
    stats_.set("num_vtx", V);
    stats_.set("num_tris", T);
    stats_.set("num_edges", E);
    stats_.set("nltgv2_total_smoothness_cost", smooth);
    stats_.set("nltgv2_avg_smoothness_cost", V ? smooth / V : 0.0);
    stats_.set("nltgv2_total_data_cost", data);
    stats_.set("nltgv2_avg_data_cost", V ? data / V : 0.0);
    stats_.set("nltgv2_iters", params_.do_nltgv2 ? params_.nltgv2_iterations : 0);
    stats_.set("hip_error", 0);
    float ms = 0.f;
    int32_t launches = 0;
    if (params_.do_nltgv2 && flame_hip_last_solve_ms(graph_.handle(), &ms, &launches) == 0)
      stats_.setTiming("nltgv2_device", ms);
    stats_.tock("update");
    return true;
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

  // reference src/flame_offline_tum.cc:680-682
  void getRawIDepths(std::vector<Point2f>* vtx, std::vector<float>* mu, std::vector<float>* var) const {
    std::lock_guard<std::mutex> lock(mtx_);
    if (vtx) *vtx = vtx_;
    if (mu) *mu = mu_;
    if (var) *var = var_;
  }

  // Dense maps (row f2).  Upstream returns cv::Mat1f (reference src/flame_offline_tum.cc:643,
  // src/flame_nodelet.cc:688); here a row-major width() x height() float vector, NaN where the
  // mesh does not cover the pixel (cv::Mat1f overloads below when OpenCV is present).
  bool getFilteredInverseDepthMap(std::vector<float>* idepthmap) const { return maps(1, idepthmap, nullptr, nullptr, 0.f, 0.f); }
  bool getInverseDepthMap(std::vector<float>* idepthmap) const { return maps(0, idepthmap, nullptr, nullptr, 0.f, 0.f); }
  // idepth -> depth inversion and point cloud of flame_ros (reference
  // src/flame_offline_tum.cc:650-661, src/utils.cc:290-312), computed on the GPU as well.
  bool getDepthMapAndCloud(std::vector<float>* depthmap, std::vector<float>* cloud_xyz,
                           float min_depth, float max_depth) const {
    return maps(1, nullptr, depthmap, cloud_xyz, min_depth, max_depth);
  }
#ifdef FLAME_HAVE_OPENCV
  void getFilteredInverseDepthMap(cv::Mat1f* idepthmap) const {
    std::vector<float> v;
    idepthmap->create(height_, width_);
    if (getFilteredInverseDepthMap(&v)) std::copy(v.begin(), v.end(), idepthmap->ptr<float>());
  }
  cv::Mat1f getInverseDepthMap() const {
    std::vector<float> v;
    cv::Mat1f m(height_, width_);
    if (getInverseDepthMap(&v)) std::copy(v.begin(), v.end(), m.ptr<float>());
    return m;
  }
#endif
  // Mesh as flame_ros publishes it (row f1; reference src/utils.cc:184-230): 12 floats per
  // vertex in flame_ros::PointNormalUV layout and reversed-winding faces of the valid triangles.
  bool getMeshPointNormalUV(std::vector<float>* points, std::vector<int32_t>* faces) const {
    std::lock_guard<std::mutex> lock(mtx_);
    if (!graph_.valid()) return false;
    const flame_hip_tri_params tp = triParams();
    points->assign(12 * vtx_.size(), 0.f);
    faces->assign(3 * tris_.size(), 0);
    int32_t nf = 0;
    if (flame_hip_mesh(graph_.handle(), Kinv_, &tp, points->data(), faces->data(), &nf)) return false;
    faces->resize(3 * static_cast<size_t>(nf));
    return true;
  }

  const utils::StatsTracker& stats() const { return stats_; }
  const Params& params() const { return params_; }
  int width() const { return width_; }
  int height() const { return height_; }

 private:
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
  // NOTE: with rescale_data the device state is in rescaled units; the dense maps are only
  // offered for rescale_data = false (the reference default, cfg/flame_offline_tum.yaml:90).
  bool maps(int filtered, std::vector<float>* idm, std::vector<float>* dm, std::vector<float>* cloud,
            float min_depth, float max_depth) const {
    std::lock_guard<std::mutex> lock(mtx_);
    if (!graph_.valid() || params_.rescale_data) return false;
    const flame_hip_tri_params tp = triParams();
    const size_t n = static_cast<size_t>(width_) * height_;
    if (idm) idm->assign(n, 0.f);
    if (dm) dm->assign(n, 0.f);
    if (cloud) cloud->assign(3 * n, 0.f);
    return flame_hip_depthmaps(graph_.handle(), Kinv_, &tp, filtered, min_depth, max_depth,
                               idm ? idm->data() : nullptr, dm ? dm->data() : nullptr,
                               cloud ? cloud->data() : nullptr) == 0;
  }
  bool fail(int code) {
    stats_.set("hip_error", code);
    stats_.tock("update");
    return false;
  }

  int width_, height_;
  Params params_;
  float K_[9], Kinv_[9];
  mutable std::mutex mtx_;
  utils::StatsTracker stats_;
  optimizers::nltgv2_l1_graph_regularizer::Graph graph_;
  std::vector<Point2f> vtx_;
  std::vector<float> mu_, var_, idepths_, normals_flat_;
  std::vector<Triangle> tris_;
  std::vector<Edge> edges_;
  std::vector<uint8_t> tri_valid_;
};

}  // namespace flame
