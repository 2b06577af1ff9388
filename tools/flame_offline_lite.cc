// tools/flame_offline_lite.cc -- a ROS-free miniature of flame_offline_tum (reference
// src/flame_offline_tum.cc:404-412 construct, :565-601 the per-image loop, :628-635 mesh out,
// :706-707 stats): TUM index -> image files -> pixels (include/flame_ros/dataset_streams.h,
// image_io.h) -> flame::Flame::update() with a registered FrontEnd -> idepth mesh + stats per frame.
//
// BASELINE config 1 ("flame_offline_tum ..., single frame-pair (plumbing)") needs upstream's feature
// pipeline (detection, epipolar tracking, Delaunay), which is not part of this build and plugs in
// through flame::FrontEnd.  The stand-in used here is deliberately simple and says so: one feature
// per detection_win_size cell (cfg/flame_offline_tum.yaml:78) where the dataset's DEPTH image is
// valid, idepth = 1 / depth at that pixel (what analysis/pass_in_truth feeds, src/flame_offline_tum.cc:
// 577-595).  The kept features are triangulated by the facade's built-in Delaunay triangulator
// (flame/utils/delaunay.h); everything behind the FrontEnd is the product path.
//
//   flame_offline_lite <index.txt> <frame RDF|FLU|...> fx fy cx cy [iters] -> one line per frame:
//   frame <id> time <t> ok <0|1> feats <n> vtx <n> tris <n> edges <n> coverage <c> cost_smooth <s> cost_data <d> rms_vs_truth <r> update_ms <ms>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <string>
#include <vector>

#include "flame/flame.h"
#include "flame_ros/dataset_streams.h"

namespace ds = flame_ros::datasets;

int main(int argc, char** argv) {
  if (argc < 7) { std::fprintf(stderr, "usage: %s index.txt frame fx fy cx cy [iters]\n", argv[0]); return 2; }
  const char* names[] = {"RDF", "FLU", "FRD", "RDF_IN_FLU", "RDF_IN_FRD", "RFU"};
  ds::Frame in_frame = ds::RDF;
  for (int k = 0; k < 6; ++k) if (!std::strcmp(argv[2], names[k])) in_frame = static_cast<ds::Frame>(k);
  const float fx = std::atof(argv[3]), fy = std::atof(argv[4]), cx = std::atof(argv[5]), cy = std::atof(argv[6]);
  ds::TumIndex index(argv[1], in_frame);
  if (index.size() == 0) return 3;

  flame::Params params;  // cfg/flame_offline_tum.yaml defaults
  if (argc > 7) params.nltgv2_iterations = std::atoi(argv[7]);
  const int win = params.detection_win_size;

  std::shared_ptr<flame::Flame> sensor;
  std::vector<float> depth;  // the current frame's depth image in metres (shared with the front end)
  int W = 0, H = 0, cols = 0, rows = 0;

  flame::FrontEnd fe;
  fe.track = [&](const flame::FrameInput&, flame::FeatureSet* fs) {
    cols = W / win; rows = H / win;
    for (int r = 0; r < rows; ++r)
      for (int c = 0; c < cols; ++c) {
        const int u = c * win + win / 2, v = r * win + win / 2;
        const float d = depth.empty() ? 0.f : depth[static_cast<size_t>(v) * W + u];
        if (!(d > 0.f)) continue;  // no depth measurement in this cell: no feature
        fs->vtx.push_back(flame::Point2f(static_cast<float>(u), static_cast<float>(v)));
        fs->idepth_mu.push_back(1.0f / d);
        fs->idepth_var.push_back(1e-4f);
      }
    return !fs->vtx.empty();
  };
  // (no fe.triangulate: the facade's built-in Delaunay triangulator, flame/utils/delaunay.h)

  uint32_t id = 0;
  ds::TumFrame fr;
  int failed = 0;
  while (index.get(&id, &fr)) {
    std::vector<uint8_t> gray;
    std::string err;
    if (!ds::loadFramePixels(fr.rgb_file, fr.has_depth ? fr.depth_file : std::string(), index.depthScaleFactor(), nullptr,
                             false, &W, &H, &gray, &depth, &err)) {
      std::fprintf(stderr, "%s\n", err.c_str());
      return 4;
    }
    if (!sensor) {
      flame::Matrix3f K, Kinv;
      K(0, 0) = fx; K(0, 1) = 0.f; K(0, 2) = cx; K(1, 0) = 0.f; K(1, 1) = fy; K(1, 2) = cy; K(2, 0) = 0.f; K(2, 1) = 0.f; K(2, 2) = 1.f;
      Kinv(0, 0) = 1.f / fx; Kinv(0, 1) = 0.f; Kinv(0, 2) = -cx / fx; Kinv(1, 0) = 0.f; Kinv(1, 1) = 1.f / fy; Kinv(1, 2) = -cy / fy;
      Kinv(2, 0) = 0.f; Kinv(2, 1) = 0.f; Kinv(2, 2) = 1.f;
      sensor = std::make_shared<flame::Flame>(W, H, K, Kinv, params);
      sensor->setFrontEnd(fe);
    }
    flame::Image1b img(H, W);
    std::memcpy(static_cast<void*>(&img(0, 0)), gray.data(), gray.size());
    flame::SE3f pose;
    pose.q[0] = static_cast<float>(fr.pose_optical.q.x); pose.q[1] = static_cast<float>(fr.pose_optical.q.y);
    pose.q[2] = static_cast<float>(fr.pose_optical.q.z); pose.q[3] = static_cast<float>(fr.pose_optical.q.w);
    for (int k = 0; k < 3; ++k) pose.t[k] = static_cast<float>(fr.pose_optical.t[k]);
    const bool ok = sensor->update(fr.time, id, pose, img, (id % 10) == 0);
    if (!ok) ++failed;
    std::vector<flame::Point2f> vtx;
    std::vector<float> idepths;
    std::vector<flame::Vector3f> normals;
    std::vector<flame::Triangle> tris;
    std::vector<bool> valid;
    std::vector<flame::Edge> edges;
    sensor->getInverseDepthMesh(&vtx, &idepths, &normals, &tris, &valid, &edges);
    double se = 0.0;
    size_t n = 0;
    for (size_t v = 0; ok && v < vtx.size(); ++v) {
      const float d = depth[static_cast<size_t>(vtx[v].y) * W + static_cast<size_t>(vtx[v].x)];
      if (d > 0.f) { const double e = idepths[v] - 1.0 / d; se += e * e; ++n; }
    }
    const flame::utils::StatsTracker& st = sensor->stats();
    std::printf("frame %u time %.6f ok %d feats %d vtx %zu tris %zu edges %zu coverage %.4f cost_smooth %.6g cost_data %.6g rms_vs_truth %.6g update_ms %.3f hip_error %d\n",
                id, fr.time, ok ? 1 : 0, static_cast<int>(st.stats("num_feats")), vtx.size(), tris.size(), edges.size(),
                st.stats("coverage"), st.stats("nltgv2_total_smoothness_cost"), st.stats("nltgv2_total_data_cost"),
                n ? std::sqrt(se / n) : 0.0, st.timings("update"), static_cast<int>(st.stats("hip_error")));
  }
  return failed ? 3 : 0;
}
