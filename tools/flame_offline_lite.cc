// tools/flame_offline_lite.cc -- a ROS-free miniature of the reference's two offline frontends on the facade:
//   flame_offline_tum (reference src/flame_offline_tum.cc:404-412 construct, :565-601 the per-image loop, :628-635
//                      mesh out, :706-707 stats)           = BASELINE config 1's plumbing
//   flame_offline_asl (reference src/flame_offline_asl.cc:398-407 construct with the dataset's K, :423-435 the loop:
//                      pose as Eigen doubles, CAST TO FLOAT for Sophus::SE3f, colour image rectified by the stream,
//                      src/ros_sensor_streams/asl_rgbd_offline_stream.cc:152-345) = BASELINE config 3's plumbing
// dataset index -> image files -> pixels (include/flame_ros/dataset_streams.h, image_io.h) -> flame::Flame::update()
// with a registered FrontEnd -> idepth mesh + stats per frame.
//
// Upstream's feature pipeline (detection, epipolar tracking) is not part of this build and plugs in through
// flame::FrontEnd.  The stand-in used here is deliberately simple and says so: one feature per detection_win_size
// cell (cfg/flame_offline_tum.yaml:78) where the dataset's DEPTH image is valid, idepth = 1 / depth at that pixel
// (what analysis/pass_in_truth feeds, src/flame_offline_tum.cc:577-595).  The kept features are triangulated by the
// facade's built-in Delaunay triangulator (flame/utils/delaunay.h); everything behind the FrontEnd is the product path.
//
//   flame_offline_lite [tum] <index.txt> <frame RDF|FLU|...> fx fy cx cy [iters] [--dump dir]
//   flame_offline_lite asl <pose_dir> <rgb_dir> <depth_dir> <world frame RDF|FLU|FRD|RFU> [iters] [--dump dir]
//     (K, the distortion coefficients and the depth scale come from the sensor.yaml files, as in the reference)
// -> one line per frame:
//   frame <id> time <t> ok <0|1> feats <n> vtx <n> tris <n> edges <n> coverage <c> cost_smooth <s> cost_data <d> rms_vs_truth <r> update_ms <ms> ...
// --dump dir: frame_<id>.bin = {int32 V, T; float pos[2V], idepth_mu[V], idepth_var[V]; int32 tris[3T]; float idepth[V]}:
// what went into the regulariser and what came out, for a bit-for-bit comparison with the oracle (tests).
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

namespace {

ds::Frame parseFrame(const char* s) {
  const char* names[] = {"RDF", "FLU", "FRD", "RDF_IN_FLU", "RDF_IN_FRD", "RFU"};
  for (int k = 0; k < 6; ++k) if (!std::strcmp(s, names[k])) return static_cast<ds::Frame>(k);
  return ds::RDF;
}

struct Lite {
  flame::Params params;  // cfg/flame_offline_tum.yaml defaults (= cfg/flame_offline_asl.yaml's for everything used here)
  std::shared_ptr<flame::Flame> sensor;
  std::vector<float> depth;  // the current frame's depth image in metres (shared with the front end)
  std::vector<float> fmu, fvar;  // the features of the current frame as the front end handed them over
  int W = 0, H = 0;
  std::string dump_dir;
  int failed = 0;

  flame::FrontEnd frontEnd() {
    flame::FrontEnd fe;
    fe.track = [this](const flame::FrameInput&, flame::FeatureSet* fs) {
      const int win = params.detection_win_size, cols = W / win, rows = H / win;
      for (int r = 0; r < rows; ++r)
        for (int c = 0; c < cols; ++c) {
          const int u = c * win + win / 2, v = r * win + win / 2;
          const float d = depth.empty() ? 0.f : depth[static_cast<size_t>(v) * W + u];
          if (!(d > 0.f)) continue;  // no depth measurement in this cell: no feature
          fs->vtx.push_back(flame::Point2f(static_cast<float>(u), static_cast<float>(v)));
          fs->idepth_mu.push_back(1.0f / d);
          fs->idepth_var.push_back(1e-4f);
        }
      fmu = fs->idepth_mu; fvar = fs->idepth_var;
      return !fs->vtx.empty();
    };
    // (no fe.triangulate: the facade's built-in Delaunay triangulator, flame/utils/delaunay.h)
    return fe;
  }

  void construct(float fx, float fy, float cx, float cy) {
    flame::Matrix3f K, Kinv;
    K(0, 0) = fx; K(0, 1) = 0.f; K(0, 2) = cx; K(1, 0) = 0.f; K(1, 1) = fy; K(1, 2) = cy; K(2, 0) = 0.f; K(2, 1) = 0.f; K(2, 2) = 1.f;
    Kinv(0, 0) = 1.f / fx; Kinv(0, 1) = 0.f; Kinv(0, 2) = -cx / fx; Kinv(1, 0) = 0.f; Kinv(1, 1) = 1.f / fy; Kinv(1, 2) = -cy / fy;
    Kinv(2, 0) = 0.f; Kinv(2, 1) = 0.f; Kinv(2, 2) = 1.f;
    sensor = std::make_shared<flame::Flame>(W, H, K, Kinv, params);
    sensor->setFrontEnd(frontEnd());
  }

  // one frame: pose in DOUBLE precision as the dataset streams deliver it, cast to float where the reference
  // builds its Sophus::SE3f (src/flame_offline_asl.cc:431, src/flame_offline_tum.cc:566-572)
  void frame(uint32_t id, double time, const ds::Pose& pose_optical, const std::vector<uint8_t>& gray) {
    flame::Image1b img(H, W);
    std::memcpy(static_cast<void*>(&img(0, 0)), gray.data(), gray.size());
    flame::SE3f pose;
    pose.q[0] = static_cast<float>(pose_optical.q.x); pose.q[1] = static_cast<float>(pose_optical.q.y);
    pose.q[2] = static_cast<float>(pose_optical.q.z); pose.q[3] = static_cast<float>(pose_optical.q.w);
    for (int k = 0; k < 3; ++k) pose.t[k] = static_cast<float>(pose_optical.t[k]);
    const bool ok = sensor->update(time, id, pose, img, (id % 10) == 0);
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
    if (ok && !dump_dir.empty() && fmu.size() == vtx.size()) {  // (every feature passes the variance gate here)
      const std::string path = dump_dir + "/frame_" + std::to_string(id) + ".bin";
      if (FILE* f = std::fopen(path.c_str(), "wb")) {
        const int32_t hdr[2] = {static_cast<int32_t>(vtx.size()), static_cast<int32_t>(tris.size())};
        std::fwrite(hdr, sizeof(hdr), 1, f);
        for (size_t v = 0; v < vtx.size(); ++v) { const float p[2] = {vtx[v].x, vtx[v].y}; std::fwrite(p, sizeof(p), 1, f); }
        std::fwrite(fmu.data(), sizeof(float), fmu.size(), f);
        std::fwrite(fvar.data(), sizeof(float), fvar.size(), f);
        for (size_t t = 0; t < tris.size(); ++t) { const int32_t q[3] = {tris[t][0], tris[t][1], tris[t][2]}; std::fwrite(q, sizeof(q), 1, f); }
        std::fwrite(idepths.data(), sizeof(float), idepths.size(), f);
        std::fclose(f);
      }
    }
    const flame::utils::StatsTracker& st = sensor->stats();
    std::printf("frame %u time %.6f ok %d feats %d vtx %zu tris %zu edges %zu coverage %.4f cost_smooth %.6g cost_data %.6g rms_vs_truth %.6g update_ms %.3f hip_error %d persist_used %d pose_t %.9g %.9g %.9g pose_q %.9g %.9g %.9g %.9g\n",
                id, time, ok ? 1 : 0, static_cast<int>(st.stats("num_feats")), vtx.size(), tris.size(), edges.size(),
                st.stats("coverage"), st.stats("nltgv2_total_smoothness_cost"), st.stats("nltgv2_total_data_cost"),
                n ? std::sqrt(se / n) : 0.0, st.timings("update"), static_cast<int>(st.stats("hip_error")),
                static_cast<int>(st.stats("persist_used")), pose.t[0], pose.t[1], pose.t[2], pose.q[0], pose.q[1], pose.q[2], pose.q[3]);
  }
};

}  // namespace

int main(int argc, char** argv) {
  std::vector<char*> args;
  Lite L;
  for (int k = 1; k < argc; ++k) {
    if (!std::strcmp(argv[k], "--dump") && k + 1 < argc) L.dump_dir = argv[++k];
    else args.push_back(argv[k]);
  }
  const bool asl = !args.empty() && !std::strcmp(args[0], "asl");
  if (!args.empty() && (!std::strcmp(args[0], "tum") || asl)) args.erase(args.begin());
  if ((asl && args.size() < 4) || (!asl && args.size() < 6)) {
    std::fprintf(stderr, "usage: %s [tum] index.txt frame fx fy cx cy [iters] [--dump dir]\n       %s asl pose_dir rgb_dir depth_dir world_frame [iters] [--dump dir]\n",
                 argv[0], argv[0]);
    return 2;
  }
  std::string err;
  if (asl) {
    // ---- flame_offline_asl: K / D / depth scale from the sensor.yaml files, colour image rectified, depth not ----
    ds::AslDataset data(args[0], args[1], std::strcmp(args[2], "-") ? args[2] : "", parseFrame(args[3]));
    if (!data.ok() || data.size() == 0) { std::fprintf(stderr, "cannot read the ASL folders\n"); return 3; }
    if (args.size() > 4) L.params.nltgv2_iterations = std::atoi(args[4]);
    flame_ros::images::PlumbBob cam;
    cam.fx = static_cast<float>(data.K()[0]); cam.fy = static_cast<float>(data.K()[4]);
    cam.cx = static_cast<float>(data.K()[2]); cam.cy = static_cast<float>(data.K()[5]);
    cam.k1 = static_cast<float>(data.D()[0]); cam.k2 = static_cast<float>(data.D()[1]);
    cam.p1 = static_cast<float>(data.D()[2]); cam.p2 = static_cast<float>(data.D()[3]); cam.k3 = static_cast<float>(data.D()[4]);
    const bool distorted = cam.k1 != 0.f || cam.k2 != 0.f || cam.p1 != 0.f || cam.p2 != 0.f || cam.k3 != 0.f;
    uint32_t id = 0;
    ds::AslFrame fr;
    while (data.get(&id, &fr)) {
      std::vector<uint8_t> gray;
      if (!ds::loadFramePixels(fr.rgb_file, fr.has_depth ? fr.depth_file : std::string(), static_cast<float>(data.depthScaleFactor()),
                               distorted ? &cam : nullptr, false, &L.W, &L.H, &gray, &L.depth, &err)) {
        std::fprintf(stderr, "%s\n", err.c_str());
        return 4;
      }
      if (L.W != data.width() || L.H != data.height()) { std::fprintf(stderr, "image size differs from sensor.yaml's resolution\n"); return 4; }
      if (!L.sensor) L.construct(cam.fx, cam.fy, cam.cx, cam.cy);
      L.frame(id, fr.time, fr.pose_optical, gray);
    }
  } else {
    // ---- flame_offline_tum ----
    const float fx = std::atof(args[2]), fy = std::atof(args[3]), cx = std::atof(args[4]), cy = std::atof(args[5]);
    ds::TumIndex index(args[0], parseFrame(args[1]));
    if (index.size() == 0) return 3;
    if (args.size() > 6) L.params.nltgv2_iterations = std::atoi(args[6]);
    uint32_t id = 0;
    ds::TumFrame fr;
    while (index.get(&id, &fr)) {
      std::vector<uint8_t> gray;
      if (!ds::loadFramePixels(fr.rgb_file, fr.has_depth ? fr.depth_file : std::string(), index.depthScaleFactor(), nullptr,
                               false, &L.W, &L.H, &gray, &L.depth, &err)) {
        std::fprintf(stderr, "%s\n", err.c_str());
        return 4;
      }
      if (!L.sensor) L.construct(fx, fy, cx, cy);
      L.frame(id, fr.time, fr.pose_optical, gray);
    }
  }
  return L.failed ? 3 : 0;
}
