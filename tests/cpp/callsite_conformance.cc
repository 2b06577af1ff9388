// tests/cpp/callsite_conformance.cc -- flame_ros' own call sites of the flame:: API, reproduced
// against include/flame/ with the OpenCV / Eigen / Sophus types the reference passes (API stand-ins
// under tests/cpp/standins/, since none of the three libraries is in this image).  Compiled with
// -std=c++11 -Wall -Wextra -Werror (the reference's standard, reference CMakeLists.txt:26).
//
// Each block cites the reference lines it mirrors.  Usage: callsite_conformance <device>
// Exit code 0 = every update succeeded (GPU present), 3 = update() returned false the way the
// reference expects on failure (no device), anything else = harness error.
#include <unistd.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <limits>
#include <memory>
#include <vector>

// reference src/flame_offline_tum.cc:37-41, 58-61; src/utils.h:28-36; src/utils.cc:29-30
#include <Eigen/Core>
#include <sophus/se3.hpp>
#include <opencv2/core/core.hpp>

#include <flame/flame.h>
#include <flame/utils/image_utils.h>
#include <flame/utils/stats_tracker.h>
#include <flame/utils/load_tracker.h>
#include <flame/utils/triangulator.h>
#include <flame/utils/visualization.h>

namespace fu = flame::utils;  // reference src/flame_offline_tum.cc:70

// reference src/flame_offline_tum.cc:74-77
void crash_handler(int sig) {
  (void)sig;
  FLAME_ASSERT(false);
  return;
}

// reference src/utils.h:72-83 (publishFlameStats takes the maps by const reference)
static int countKeys(const std::unordered_map<std::string, double>& stats,
                     const std::unordered_map<std::string, double>& timings) {
  return static_cast<int>(stats.size() + timings.size());
}

// reference src/utils.cc:163-237 publishDepthMesh signature
static int meshTouch(const Eigen::Matrix3f& Kinv, const std::vector<cv::Point2f>& vertices,
                     const std::vector<float>& idepths, const std::vector<Eigen::Vector3f>& normals,
                     const std::vector<flame::Triangle>& triangles, const std::vector<bool>& tri_validity) {
  int faces = 0;
  for (size_t ii = 0; ii < vertices.size(); ++ii) {
    float id = idepths[ii];
    if (!std::isnan(id) && (id > 0)) {  // src/utils.cc:189
      Eigen::Vector3f p(vertices[ii].x / id, vertices[ii].y / id, 1.0f / id);
      (void)p; (void)Kinv; (void)normals[ii](0);
    }
  }
  for (size_t ii = 0; ii < triangles.size(); ++ii)
    if (tri_validity[ii]) {  // src/utils.cc:221-227: reversed winding
      unsigned a = triangles[ii][2], b = triangles[ii][1], c = triangles[ii][0];
      (void)a; (void)b; (void)c;
      ++faces;
    }
  return faces;
}

class Frontend {
 public:
  // reference src/flame_offline_tum.cc:98-101: stats_(), load_(getpid())
  Frontend() : stats_(), load_(getpid()), num_imgs_(0), poseframe_subsample_factor_(6), pass_in_truth_(false) {
    // reference src/flame_offline_tum.cc:158-249: every flame::Params field the frontends assign
    params_.debug_quiet = true;
    params_.scene_color_scale = 1.0f;
    params_.do_oblique_triangle_filter = true;
    double oblique_normal_thresh = 1.57; params_.oblique_normal_thresh = oblique_normal_thresh;
    params_.oblique_idepth_diff_factor = 0.35f;
    params_.oblique_idepth_diff_abs = 0.1f;
    params_.do_edge_length_filter = true;
    double edge_length_thresh = 0.333; params_.edge_length_thresh = edge_length_thresh;
    params_.do_idepth_triangle_filter = true;
    double min_triangle_idepth = 0.01; params_.min_triangle_idepth = min_triangle_idepth;
    params_.debug_draw_wireframe = true; params_.debug_draw_features = true;
    params_.debug_draw_detections = true; params_.debug_draw_matches = true;
    params_.debug_draw_normals = true; params_.debug_draw_idepthmap = true;
    params_.debug_draw_text_overlay = true; params_.debug_flip_images = false;
    params_.omp_num_threads = 4; params_.omp_chunk_size = 1024;
    params_.do_letterbox = false;
    params_.min_grad_mag = 5.0f; params_.fparams.min_grad_mag = params_.min_grad_mag;
    double min_error = 100.0; params_.min_error = min_error;
    params_.detection_win_size = 16;
    int win_size = 5; params_.zparams.win_size = win_size; params_.fparams.win_size = win_size;
    params_.max_dropouts = 5;
    double epipolar_line_var = 4.0; params_.zparams.epipolar_line_var = epipolar_line_var;
    params_.do_nltgv2 = true;
    params_.adaptive_data_weights = false;
    params_.rescale_data = false;
    params_.init_with_prediction = true;
    params_.idepth_var_max_graph = 0.01f;
    params_.rparams.data_factor = 0.15f; params_.rparams.step_x = 0.001f;
    params_.rparams.step_q = 125.0f; params_.rparams.theta = 0.25f;
    params_.min_height = -100000000000000.0f; params_.max_height = 100000000000000.0f;
    params_.check_sticky_obstacles = false;
    // reference src/flame_nodelet.cc:153
    load_ = std::move(fu::LoadTracker(getpid()));
  }

  int run(int device) {
    params_.hip_device = device;
    params_.nltgv2_iterations = 60;
    const int width = 640, height = 480;
    Eigen::Matrix3f K = Eigen::Matrix3f::Identity();  // cfg/kinect.yaml
    K(0, 0) = 525.f; K(1, 1) = 525.f; K(0, 2) = 319.5f; K(1, 2) = 239.5f;
    // reference src/flame_offline_tum.cc:404-412
    Kinv_ = K.inverse();
    sensor_ = std::make_shared<flame::Flame>(width, height, K, Kinv_, params_);

    // Without a feature pipeline update() reports failure the reference way.
    cv::Mat1b img_gray(height, width, static_cast<unsigned char>(0));
    Sophus::SE3f pose(Eigen::Quaternionf(1.f, 0.f, 0.f, 0.f), Eigen::Vector3f(0.f, 0.f, 0.f));
    if (sensor_->update(0.0, 0, pose, img_gray, true)) return 20;
    if (static_cast<int>(sensor_->stats().stats("hip_error")) != FLAME_HIP_ERR_STATE) return 21;

    // The feature pipeline of upstream's update() (detection / tracking / idepth filtering and
    // Delaunay) enters through FrontEnd; here: a regular feature grid split into two triangles
    // per cell, a planar idepth field, one feature with too large a variance (must be gated out
    // before triangulation) and predictions for half of the features.
    const int nx = 20, ny = 15;
    flame::FrontEnd fe;
    fe.track = [=](const flame::FrameInput& in, flame::FeatureSet* fs) {
      for (int y = 0; y < ny; ++y)
        for (int x = 0; x < nx; ++x) {
          const float u = 16.f + 32.f * x + ((x * 7 + y * 3) % 5), v = 16.f + 32.f * y + ((x * 5 + y) % 7);
          fs->vtx.push_back(cv::Point2f(u, v));
          float id = 0.4f + 0.0005f * u + 0.0002f * v + (((x + y) % 4) ? 0.f : 0.03f);
          if (in.idepths_true) id = (*in.idepths_true)(static_cast<int>(v), static_cast<int>(u));
          fs->idepth_mu.push_back(id);
          fs->idepth_var.push_back(1e-4f);
          fs->prediction.push_back((x % 2) ? id : std::numeric_limits<float>::quiet_NaN());
        }
      fs->vtx.push_back(cv::Point2f(5.f, 5.f));  // fails the gate (idepth_var_max_graph = 0.01)
      fs->idepth_mu.push_back(3.0f);
      fs->idepth_var.push_back(0.5f);
      fs->prediction.push_back(std::numeric_limits<float>::quiet_NaN());
      return in.img != nullptr && in.img->rows == 480;
    };
    fe.triangulate = [=](const std::vector<cv::Point2f>& pts, std::vector<flame::Triangle>* tris) {
      if (static_cast<int>(pts.size()) != nx * ny) return false;  // the gated feature must be gone
      for (int y = 0; y + 1 < ny; ++y)
        for (int x = 0; x + 1 < nx; ++x) {
          const int a = y * nx + x, b = a + 1, c = a + nx, d = c + 1;
          tris->push_back(flame::Triangle(a, b, c));
          tris->push_back(flame::Triangle(b, d, c));
        }
      return true;
    };
    int pf_calls = 0;
    fe.updatePoseFramePoses = [&pf_calls](const std::vector<uint32_t>& ids, const std::vector<Sophus::SE3f>& poses) {
      pf_calls += static_cast<int>(ids.size() == poses.size());
    };
    fe.prunePoseFrames = [&pf_calls](const std::vector<uint32_t>&) { ++pf_calls; };
    sensor_->setFrontEnd(fe);

    int failures = 0;
    for (uint32_t img_id = 0; img_id < 3; ++img_id) {
      pass_in_truth_ = (img_id == 2);
      if (!processFrame(img_id, 0.033 * img_id, pose, img_gray)) ++failures;
      ++num_imgs_;
    }

    // reference src/flame_nodelet.cc:456-475
    std::vector<uint32_t> pf_ids(2);
    std::vector<Sophus::SE3f> pf_poses(2);
    pf_ids[0] = 0; pf_ids[1] = 6;
    sensor_->updatePoseFramePoses(pf_ids, pf_poses);
    sensor_->prunePoseFrames(pf_ids);
    if (pf_calls != 2) return 22;

    // reference src/flame_offline_tum.cc:529-543
    fu::Load max_load, sys_load, pid_load;
    load_.get(&max_load, &sys_load, &pid_load);
    stats_.set("max_load_cpu", max_load.cpu);
    stats_.set("max_load_mem", max_load.mem);
    stats_.set("max_load_swap", max_load.swap);
    stats_.set("sys_load_cpu", sys_load.cpu);
    stats_.set("pid_load_mem", pid_load.mem);
    stats_.set("pid", getpid());
    if (!(pid_load.mem > 0.0f) || !(sys_load.mem > 0.0f)) return 23;

    // reference src/flame_offline_tum.cc:337-342 (idepth error colormap)
    cv::Mat1f idepth_error(height, width, 0.1f);
    idepth_error(3, 4) = std::numeric_limits<float>::quiet_NaN();
    cv::Mat3b debug_img(height, width, cv::Vec3b(9, 9, 9));
    cv::Mat3b* debug_img_idepth_error = &debug_img;
    auto colormap = [this](float v, cv::Vec3b c) {
      return std::isnan(v) ? c : flame::utils::jet(v, 0.0f, 0.35f);
    };
    flame::utils::applyColorMap<float>(idepth_error, colormap, debug_img_idepth_error);
    if (debug_img(3, 4)[0] != 9 || debug_img(0, 0)[0] == 9) return 24;
    cv::Vec3b lo = fu::jet(0.0f, 0.0f, 1.0f), hi = fu::jet(1.0f, 0.0f, 1.0f);
    if (!(lo[0] > lo[2]) || !(hi[2] > hi[0])) return 25;  // BGR: low = blue, high = red
    if (fu::fast_roundf(2.5f) != 3 || fu::fast_roundf(-2.5f) != -3 || fu::fast_abs(-1.5f) != 1.5f) return 26;

    std::printf("frames_failed=%d hip_error=%d\n", failures, static_cast<int>(sensor_->stats().stats("hip_error")));
    return failures ? 3 : 0;
  }

 private:
  // reference src/flame_offline_tum.cc:565-782
  bool processFrame(const uint32_t img_id, const double time, const Sophus::SE3f& pose,
                    const cv::Mat1b& img_gray) {
    stats_.tick("process_frame");
    bool is_poseframe = (img_id % poseframe_subsample_factor_) == 0;
    bool update_success = false;
    if (!pass_in_truth_) {
      update_success = sensor_->update(time, img_id, pose, img_gray, is_poseframe);  // :578-579
    } else {
      cv::Mat1f depth(img_gray.rows, img_gray.cols, 2.0f);
      cv::Mat1f idepths_true(img_gray.rows, img_gray.cols, std::numeric_limits<float>::quiet_NaN());
      for (int ii = 0; ii < depth.rows; ++ii)
        for (int jj = 0; jj < depth.cols; ++jj)
          if (!std::isnan(depth(ii, jj)) && (depth(ii, jj) > 0)) idepths_true(ii, jj) = 1.0f / depth(ii, jj);
      update_success = sensor_->update(time, img_id, pose, img_gray, is_poseframe, idepths_true);  // :593-594
    }
    if (!update_success) {  // :597-601
      stats_.tock("process_frame");
      return false;
    }
    // :603-621 angular-rate gate
    Eigen::Quaternionf q_delta = pose.unit_quaternion() * prev_pose_.unit_quaternion().inverse();
    float angle_delta = fu::fast_abs(Eigen::AngleAxisf(q_delta).angle());
    if (angle_delta > 1.0f) return false;
    prev_pose_ = pose;

    // :628-636
    std::vector<cv::Point2f> vtx;
    std::vector<float> idepths;
    std::vector<Eigen::Vector3f> normals;
    std::vector<flame::Triangle> triangles;
    std::vector<flame::Edge> edges;
    std::vector<bool> tri_validity;
    sensor_->getInverseDepthMesh(&vtx, &idepths, &normals, &triangles, &tri_validity, &edges);
    const int faces = meshTouch(Kinv_, vtx, idepths, normals, triangles, tri_validity);
    if (vtx.size() != 300u || triangles.size() != 2u * 19u * 14u || edges.empty() || faces == 0) std::exit(30);

    // :640-661
    cv::Mat1f idepthmap;
    sensor_->getFilteredInverseDepthMap(&idepthmap);
    cv::Mat1f depth_est(idepthmap.rows, idepthmap.cols, std::numeric_limits<float>::quiet_NaN());
    int covered = 0;
    for (int ii = 0; ii < depth_est.rows; ++ii)
      for (int jj = 0; jj < depth_est.cols; ++jj) {
        float idepth = idepthmap(ii, jj);
        if (!std::isnan(idepth) && (idepth > 0)) { depth_est(ii, jj) = 1.0f / idepth; ++covered; }
      }
    if (idepthmap.rows != 480 || idepthmap.cols != 640 || covered < 100000) std::exit(31);
    // src/flame_nodelet.cc:688
    cv::Mat1f full = sensor_->getInverseDepthMap();
    if (full.rows != 480) std::exit(32);
    // :664-667
    float max_depth = (params_.do_idepth_triangle_filter) ? 1.0f / params_.min_triangle_idepth
                                                          : std::numeric_limits<float>::max();
    (void)max_depth;

    // :676-698 raw features scattered into an image
    cv::Mat1f depth_raw(img_gray.rows, img_gray.cols, std::numeric_limits<float>::quiet_NaN());
    std::vector<cv::Point2f> vertices;
    std::vector<float> idepths_mu, idepths_var;
    sensor_->getRawIDepths(&vertices, &idepths_mu, &idepths_var);
    if (vertices.size() != 301u) std::exit(33);  // raw = before the variance gate
    for (size_t ii = 0; ii < vertices.size(); ++ii) {
      float id = idepths_mu[ii];
      if (!std::isnan(id) && (id > 0)) {
        int x = fu::fast_roundf(vertices[ii].x);
        int y = fu::fast_roundf(vertices[ii].y);
        FLAME_ASSERT(x >= 0);
        FLAME_ASSERT(x < depth_raw.cols);
        FLAME_ASSERT(y >= 0);
        FLAME_ASSERT(y < depth_raw.rows);
        depth_raw(y, x) = 1.0f / id;
      }
    }

    // :704-708
    auto stats = sensor_->stats().stats();
    auto timings = sensor_->stats().timings();
    if (countKeys(stats, timings) < 10 || stats.find("nltgv2_total_smoothness_cost") == stats.end() ||
        timings.find("update") == timings.end() || timings.find("sync_graph") == timings.end())
      std::exit(34);
    if (stats["num_vtx"] != 300.0 || stats["num_feats"] != 301.0) std::exit(35);

    // :728-768 debug images handed to cv_bridge as cv::Mat
    int lit = 0;
    if (params_.debug_draw_wireframe) { cv::Mat m(sensor_->getDebugImageWireframe()); lit += m.rows; }
    if (params_.debug_draw_features) { cv::Mat m(sensor_->getDebugImageFeatures()); lit += m.rows; }
    if (params_.debug_draw_detections) { cv::Mat m(sensor_->getDebugImageDetections()); lit += m.rows; }
    if (params_.debug_draw_matches) { cv::Mat m(sensor_->getDebugImageMatches()); lit += m.rows; }
    if (params_.debug_draw_normals) { cv::Mat m(sensor_->getDebugImageNormals()); lit += m.rows; }
    if (params_.debug_draw_idepthmap) { cv::Mat m(sensor_->getDebugImageInverseDepthMap()); lit += m.rows; }
    if (lit != 6 * 480) std::exit(36);
    const cv::Mat3b& wf = sensor_->getDebugImageWireframe();
    const cv::Mat3b& dm = sensor_->getDebugImageInverseDepthMap();
    int wf_px = 0, dm_px = 0;
    for (int ii = 0; ii < wf.rows; ++ii)
      for (int jj = 0; jj < wf.cols; ++jj) {
        wf_px += (wf(ii, jj)[0] | wf(ii, jj)[1] | wf(ii, jj)[2]) != 0;
        dm_px += (dm(ii, jj)[0] | dm(ii, jj)[1] | dm(ii, jj)[2]) != 0;
      }
    if (wf_px < 1000 || dm_px < covered / 2) std::exit(37);

    // :502-522 fps bookkeeping on the wrapper's own tracker (missing key reads as <= 0)
    if (stats_.stats("fps_max") <= 0.0f) stats_.set("fps_max", 30.0);
    stats_.tock("process_frame");
    return stats_.timings("process_frame") >= 0.0;
  }

  fu::StatsTracker stats_;
  fu::LoadTracker load_;
  int num_imgs_;
  int poseframe_subsample_factor_;
  bool pass_in_truth_;
  flame::Params params_;
  Eigen::Matrix3f Kinv_;
  Sophus::SE3f prev_pose_;
  std::shared_ptr<flame::Flame> sensor_;
};

int main(int argc, char** argv) {
  Frontend f;
  return f.run(argc > 1 ? std::atoi(argv[1]) : 0);
}
