// tests/cpp/facade_threads.cc -- the nodelet's threading pattern (reference src/flame_nodelet.cc:
// 474-475 call updatePoseFramePoses / prunePoseFrames from the ROS callback thread while update()
// runs on the worker thread, :634-635): one thread streams frames through update(), a second one
// hammers the pose-frame mutators and every getter.  Checks: nothing crashes or deadlocks, every
// mesh read is internally consistent (all vectors belong to ONE frame), the front end callbacks
// never run concurrently with each other (the facade's mutex serialises them).
// Usage: facade_threads <device> <frames>; exit 0 = ok, 3 = update() failed (no device).
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <memory>
#include <thread>
#include <vector>

#include "flame/flame.h"

int main(int argc, char** argv) {
  const int device = argc > 1 ? std::atoi(argv[1]) : 0;
  const int nframes = argc > 2 ? std::atoi(argv[2]) : 60;
  flame::Params params;
  params.hip_device = device;
  params.nltgv2_iterations = 40;
  flame::Matrix3f K, Kinv;
  K(0, 0) = 525.f; K(0, 1) = 0.f; K(0, 2) = 319.5f; K(1, 0) = 0.f; K(1, 1) = 525.f; K(1, 2) = 239.5f;
  K(2, 0) = 0.f; K(2, 1) = 0.f; K(2, 2) = 1.f;
  Kinv(0, 0) = 1.f / 525.f; Kinv(0, 1) = 0.f; Kinv(0, 2) = -319.5f / 525.f;
  Kinv(1, 0) = 0.f; Kinv(1, 1) = 1.f / 525.f; Kinv(1, 2) = -239.5f / 525.f;
  Kinv(2, 0) = 0.f; Kinv(2, 1) = 0.f; Kinv(2, 2) = 1.f;
  std::shared_ptr<flame::Flame> sensor = std::make_shared<flame::Flame>(640, 480, K, Kinv, params);

  // front end: a jittered grid of features whose size changes every frame (V = cols * rows), two
  // triangles per cell; `inside` detects overlapping callbacks
  std::atomic<int> inside(0), overlaps(0), pf_calls(0);
  flame::FrontEnd fe;
  fe.track = [&](const flame::FrameInput& in, flame::FeatureSet* fs) {
    if (inside.fetch_add(1) != 0) overlaps.fetch_add(1);
    const int cols = 20 + static_cast<int>(in.img_id % 7), rows = 15 + static_cast<int>(in.img_id % 5);
    for (int r = 0; r < rows; ++r)
      for (int c = 0; c < cols; ++c) {
        const float jx = static_cast<float>((r * 31 + c * 17 + in.img_id) % 9) * 0.3f;
        const float jy = static_cast<float>((r * 13 + c * 29 + in.img_id) % 7) * 0.3f;
        fs->vtx.push_back(flame::Point2f(8.f + 600.f * c / cols + jx, 8.f + 440.f * r / rows + jy));
        fs->idepth_mu.push_back(0.5f + 0.001f * c + 0.0005f * r);
        fs->idepth_var.push_back(1e-4f);
      }
    fs->prediction.clear();
    inside.fetch_sub(1);
    return true;
  };
  fe.triangulate = [&](const std::vector<flame::Point2f>& vtx, std::vector<flame::Triangle>* tris) {
    if (inside.fetch_add(1) != 0) overlaps.fetch_add(1);
    // the grid shape is recovered from the vertex count of this frame's track()
    int cols = 0, rows = 0;
    for (int c = 20; c < 27 && !cols; ++c)
      for (int r = 15; r < 20; ++r)
        if (c * r == static_cast<int>(vtx.size())) { cols = c; rows = r; break; }
    tris->clear();
    for (int r = 0; r + 1 < rows; ++r)
      for (int c = 0; c + 1 < cols; ++c) {
        const int a = r * cols + c;
        tris->push_back(flame::Triangle(a, a + 1, a + cols));
        tris->push_back(flame::Triangle(a + 1, a + cols + 1, a + cols));
      }
    inside.fetch_sub(1);
    return cols > 0;
  };
  fe.updatePoseFramePoses = [&](const std::vector<uint32_t>&, const std::vector<flame::SE3f>&) {
    if (inside.fetch_add(1) != 0) overlaps.fetch_add(1);
    pf_calls.fetch_add(1);
    inside.fetch_sub(1);
  };
  fe.prunePoseFrames = [&](const std::vector<uint32_t>&) {
    if (inside.fetch_add(1) != 0) overlaps.fetch_add(1);
    pf_calls.fetch_add(1);
    inside.fetch_sub(1);
  };
  sensor->setFrontEnd(fe);

  std::atomic<bool> done(false);
  std::atomic<int> bad(0), reads(0);
  std::thread callback([&] {  // the ROS callback thread
    std::vector<flame::Point2f> vtx;
    std::vector<float> id;
    std::vector<flame::Vector3f> nrm;
    std::vector<flame::Triangle> tris;
    std::vector<bool> valid;
    std::vector<flame::Edge> edges;
    std::vector<uint32_t> ids(3, 1);
    std::vector<flame::SE3f> poses(3);
    while (!done.load()) {
      sensor->updatePoseFramePoses(ids, poses);
      sensor->prunePoseFrames(ids);
      sensor->getInverseDepthMesh(&vtx, &id, &nrm, &tris, &valid, &edges);
      reads.fetch_add(1);
      if (vtx.size() != id.size() || vtx.size() != nrm.size() || tris.size() != valid.size()) bad.fetch_add(1);
      for (size_t t = 0; t < tris.size(); ++t)
        for (int k = 0; k < 3; ++k)
          if (tris[t][k] < 0 || tris[t][k] >= static_cast<int>(vtx.size())) { bad.fetch_add(1); break; }
      for (size_t e = 0; e < edges.size(); ++e)
        if (edges[e][0] >= static_cast<int>(vtx.size()) || edges[e][1] >= static_cast<int>(vtx.size())) { bad.fetch_add(1); break; }
      (void)sensor->stats().stats("num_vtx");
      (void)sensor->getDebugImageWireframe();
    }
  });
  flame::Image1b img(480, 640);
  int failed = 0;
  for (int k = 0; k < nframes; ++k) {
    if (!sensor->update(0.033 * k, static_cast<uint32_t>(k), flame::SE3f(), img, (k % 10) == 0)) ++failed;
    // frames arrive at camera rate, not back to back (std::mutex is not fair: a worker that re-locks
    // at once would starve the callback thread and the test would exercise nothing)
    std::this_thread::sleep_for(std::chrono::microseconds(500));
  }
  done.store(true);
  callback.join();
  std::printf("frames=%d failed=%d reads=%d inconsistent=%d overlaps=%d pf_calls=%d hip_error=%d\n", nframes, failed,
              reads.load(), bad.load(), overlaps.load(), pf_calls.load(), static_cast<int>(sensor->stats().stats("hip_error")));
  if (bad.load() || overlaps.load()) return 21;
  return failed ? 3 : 0;
}
