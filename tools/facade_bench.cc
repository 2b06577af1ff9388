// tools/facade_bench.cc -- latency of the REAL integration path: a frame stream through
// flame::Flame::updateGraph (include/flame/flame.h) with the reference's default parameters
// (reference cfg/flame_offline_tum.yaml:19-99: wireframe / features / idepthmap debug draws ON,
// all triangle filters ON), the way flame_offline_tum drives it (reference
// src/flame_offline_tum.cc:565-782): update, then the mesh getter, stats, and -- optionally -- the
// debug image getters a frontend with debug publishing enabled would call.
//
// Usage: facade_bench <width> <height> <iters> <repeats> <getters> frame0.bin [frame1.bin ...]
//   frame files: the format of tests/cpp/facade_conformance.cc (header V, T, iters, device, flags;
//   pos, mu, [var], tris), written by tools/facade_frames.py.  The frames are fed round-robin
//   `repeats` times to ONE flame::Flame (every frame is a new graph for the library: nothing of a
//   previous frame's topology is reused).  getters: 0 = update only, 1 = + getInverseDepthMesh,
//   2 = + the three default debug images.  Prints one JSON line.
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <memory>
#include <string>
#include <vector>

#include "flame/flame.h"

namespace {
struct Frame {
  std::vector<flame::Point2f> vtx;
  std::vector<float> mu, var;
  std::vector<flame::Triangle> tris;
};

bool load(const char* path, Frame* f) {
  FILE* fp = std::fopen(path, "rb");
  if (!fp) return false;
  std::fseek(fp, 0, SEEK_END);
  const long n = std::ftell(fp);
  std::fseek(fp, 0, SEEK_SET);
  std::vector<char> buf(n);
  const bool ok = std::fread(buf.data(), 1, n, fp) == static_cast<size_t>(n);
  std::fclose(fp);
  if (!ok) return false;
  const int32_t* hdr = reinterpret_cast<const int32_t*>(buf.data());
  const int V = hdr[0], T = hdr[1];
  const float* pos = reinterpret_cast<const float*>(hdr + 5);
  const float* mu = pos + 2 * V;
  const float* varp = (hdr[4] & 4) ? mu + V : nullptr;
  const int32_t* tri = reinterpret_cast<const int32_t*>(mu + V + (varp ? V : 0));
  f->vtx.resize(V); f->mu.assign(mu, mu + V); f->var.assign(V, 1e-4f); f->tris.resize(T);
  for (int v = 0; v < V; ++v) { f->vtx[v] = flame::Point2f(pos[2 * v], pos[2 * v + 1]); if (varp) f->var[v] = varp[v]; }
  for (int t = 0; t < T; ++t) f->tris[t] = flame::Triangle(tri[3 * t], tri[3 * t + 1], tri[3 * t + 2]);
  return true;
}

double pct(std::vector<double> v, double p) {
  if (v.empty()) return 0.0;
  std::sort(v.begin(), v.end());
  return v[std::min(v.size() - 1, static_cast<size_t>(p * (v.size() - 1) + 0.5))];
}
}  // namespace

int main(int argc, char** argv) {
  if (argc < 7) { std::fprintf(stderr, "usage: %s W H iters repeats getters frame.bin...\n", argv[0]); return 10; }
  const int W = std::atoi(argv[1]), H = std::atoi(argv[2]), iters = std::atoi(argv[3]);
  const int repeats = std::atoi(argv[4]), getters = std::atoi(argv[5]);
  std::vector<Frame> frames(argc - 6);
  for (int a = 6; a < argc; ++a)
    if (!load(argv[a], &frames[a - 6])) return 11;

  flame::Params params;  // defaults = reference cfg/flame_offline_tum.yaml
  params.nltgv2_iterations = iters;
  if (const char* e = std::getenv("FLAME_BENCH_NO_DEBUG")) {  // A/B: the debug draws off
    if (e[0] == '1') params.debug_draw_wireframe = params.debug_draw_features = params.debug_draw_idepthmap = false;
  }
  if (const char* e = std::getenv("FLAME_BENCH_TRI_GPU")) params.triangulate_on_gpu = e[0] == '1';  // A/B: built-in triangulation on the GPU / on the host pool
  if (const char* e = std::getenv("FLAME_BENCH_TRI_THREADS")) params.triangulate_threads = std::atoi(e);
  flame::Matrix3f K, Kinv;  // cfg/kinect.yaml: 525/525/319.5/239.5
  K(0, 0) = 525.f; K(0, 1) = 0.f; K(0, 2) = 319.5f; K(1, 0) = 0.f; K(1, 1) = 525.f; K(1, 2) = 239.5f;
  K(2, 0) = 0.f; K(2, 1) = 0.f; K(2, 2) = 1.f;
  Kinv(0, 0) = 1.f / 525.f; Kinv(0, 1) = 0.f; Kinv(0, 2) = -319.5f / 525.f;
  Kinv(1, 0) = 0.f; Kinv(1, 1) = 1.f / 525.f; Kinv(1, 2) = -239.5f / 525.f;
  Kinv(2, 0) = 0.f; Kinv(2, 1) = 0.f; Kinv(2, 2) = 1.f;
  std::shared_ptr<flame::Flame> sensor = std::make_shared<flame::Flame>(W, H, K, Kinv, params);

  // FLAME_BENCH_FRONTEND=1: the frames go through update() with a FrontEnd whose track() hands over the
  // frame's features and NO triangulate(): the built-in triangulator (on the GPU, flame_hip_delaunay, or flame/utils/delaunay.h on the host) runs
  // inside update(), as it does for a caller that only brings features
  const char* fe_env = std::getenv("FLAME_BENCH_FRONTEND");
  const bool with_frontend = fe_env && fe_env[0] == '1';
  const Frame* cur = nullptr;
  if (with_frontend) {
    flame::FrontEnd fe;
    fe.track = [&](const flame::FrameInput&, flame::FeatureSet* fs) {
      fs->vtx = cur->vtx; fs->idepth_mu = cur->mu; fs->idepth_var = cur->var;
      return true;
    };
    sensor->setFrontEnd(fe);
  }
  flame::Image1b gray(H, W);
  flame::SE3f pose;
  pose.q[0] = pose.q[1] = pose.q[2] = 0.f; pose.q[3] = 1.f;
  pose.t[0] = pose.t[1] = pose.t[2] = 0.f;
  std::vector<double> upd, sync, solve, wall, get, tri;
  std::vector<flame::Point2f> ovtx;
  std::vector<float> oid;
  std::vector<flame::Vector3f> normals;
  std::vector<flame::Triangle> otris;
  std::vector<bool> validity;
  std::vector<flame::Edge> edges;
  const int warm = 3;
  int n = 0;
  unsigned long checksum = 0;
  int resident_frames = 0, recovered_frames = 0;  // frames solved by one launch of resident tiles; of those, repeated after a give-up
  unsigned long long x_hash = 1469598103934665603ull;
  for (int r = 0; r < repeats + warm; ++r)
    for (size_t k = 0; k < frames.size(); ++k, ++n) {
      const Frame& f = frames[k];
      const auto t0 = std::chrono::steady_clock::now();
      cur = &f;
      const bool ok = with_frontend ? sensor->update(0.033 * n, static_cast<uint32_t>(n), pose, gray, false)
                                    : sensor->updateGraph(0.033 * n, static_cast<uint32_t>(n), f.vtx, f.mu, f.var, f.tris);
      const auto t1 = std::chrono::steady_clock::now();
      if (!ok) {
        std::printf("{\"error\": \"update failed\", \"hip_error\": %d}\n", static_cast<int>(sensor->stats().stats("hip_error")));
        return 3;
      }
      if (getters >= 1) {
        sensor->getInverseDepthMesh(&ovtx, &oid, &normals, &otris, &validity, &edges);
        // FNV-1a over the bits of the regularised idepths (vertex order = feature order, whatever triangulated them) and
        // the edge list of the frame: equal for two triangulators that return the same triangle SET
        if (r == repeats + warm - 1) {
          for (size_t v = 0; v < oid.size(); ++v) {
            uint32_t bits;
            std::memcpy(&bits, &oid[v], 4);
            for (int b8 = 0; b8 < 4; ++b8) { x_hash ^= (bits >> (8 * b8)) & 0xffu; x_hash *= 1099511628211ull; }
          }
          for (size_t e = 0; e < edges.size(); ++e) {
            const int32_t ij[2] = {static_cast<int32_t>(edges[e][0]), static_cast<int32_t>(edges[e][1])};
            for (int w = 0; w < 2; ++w)
              for (int b8 = 0; b8 < 4; ++b8) { x_hash ^= (static_cast<uint32_t>(ij[w]) >> (8 * b8)) & 0xffu; x_hash *= 1099511628211ull; }
          }
        }
      }
      if (getters >= 2) {
        const flame::Image3b& a = sensor->getDebugImageWireframe();
        const flame::Image3b& b = sensor->getDebugImageFeatures();
        const flame::Image3b& c = sensor->getDebugImageInverseDepthMap();
        checksum += a(H / 2, W / 2)[0] + b(H / 2, W / 2)[1] + c(H / 2, W / 2)[2];
      }
      const auto t2 = std::chrono::steady_clock::now();
      if (r < warm) continue;
      resident_frames += sensor->stats().stats("persist_used") > 0.5;
      recovered_frames = static_cast<int>(sensor->stats().stats("persist_recovered"));  // (cumulative in the library)
      upd.push_back(sensor->stats().timings("update"));
      sync.push_back(sensor->stats().timings("sync_graph"));
      solve.push_back(sensor->stats().timings("nltgv2"));
      tri.push_back(with_frontend ? sensor->stats().timings("triangulate") : 0.0);
      wall.push_back(std::chrono::duration<double, std::milli>(t1 - t0).count());
      get.push_back(std::chrono::duration<double, std::milli>(t2 - t1).count());
    }
  if (std::getenv("FLAME_BENCH_SERIES")) {  // dev: update() time of every measured frame, in stream order
    std::fprintf(stderr, "update_ms series:");
    for (size_t i = 0; i < upd.size(); ++i) std::fprintf(stderr, " %.2f", upd[i]);
    std::fprintf(stderr, "\n");
  }
  std::printf(
      "{\"V\": %d, \"T\": %d, \"E\": %d, \"iters\": %d, \"frames\": %d, \"getters\": %d, "
      "\"update_ms\": {\"p50\": %.4f, \"p10\": %.4f, \"p90\": %.4f, \"max\": %.4f}, "
      "\"update_wall_ms_p50\": %.4f, \"sync_graph_ms_p50\": %.4f, \"nltgv2_ms_p50\": %.4f, "
      "\"nltgv2_device_ms\": %.4f, \"getters_ms_p50\": %.4f, \"coverage\": %.6f, \"checksum\": %lu, \"x_hash\": \"%016llx\", "
      "\"triangulate_ms_p50\": %.4f, \"resident_frames\": %d, \"recovered_frames\": %d}\n",
      static_cast<int>(frames[0].vtx.size()), static_cast<int>(frames[0].tris.size()),
      static_cast<int>(sensor->stats().stats("num_edges")), iters, static_cast<int>(upd.size()), getters,
      pct(upd, 0.5), pct(upd, 0.1), pct(upd, 0.9), pct(upd, 1.0), pct(wall, 0.5), pct(sync, 0.5), pct(solve, 0.5),
      sensor->stats().timings("nltgv2_device"), pct(get, 0.5), sensor->stats().stats("coverage"), checksum, x_hash, pct(tri, 0.5), resident_frames, recovered_frames);
  return 0;
}
