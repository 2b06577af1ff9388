// tests/cpp/facade_conformance.cc -- exercises include/flame/flame.h the way flame_ros does
// (reference src/flame_offline_tum.cc:404-412 construct, :578 update, :628-635 mesh out,
// :706-707 stats, src/utils.cc:117-136 stat keys), WITHOUT OpenCV/Eigen (the fallback types of
// include/flame/types.h).  Usage: facade_conformance out.bin out.txt in1.bin [in2.bin ...]: every
// input is one frame (header V, T, iters, device, flags; pos, mu, [var], tris) fed to the SAME
// flame::Flame object in turn (a frame stream on one GPU handle); idepths / validity / normals /
// costs of the LAST frame are written back.  flags: 1 = adaptive_data_weights, 2 = rescale_data,
// 4 = a var array follows mu, 8 = debug_flip_images.  Exit code 0 = ok, 3 = update returned false (no device).
#include <cstdio>
#include <cstdlib>
#include <memory>
#include <string>
#include <vector>

#include "flame/flame.h"

static bool read_all(const char* path, std::vector<char>* buf) {
  FILE* f = std::fopen(path, "rb");
  if (!f) return false;
  std::fseek(f, 0, SEEK_END);
  long n = std::ftell(f);
  std::fseek(f, 0, SEEK_SET);
  buf->resize(n);
  bool ok = std::fread(buf->data(), 1, n, f) == static_cast<size_t>(n);
  std::fclose(f);
  return ok;
}

int main(int argc, char** argv) {
  if (argc < 4) return 10;
  std::vector<char> buf;
  if (!read_all(argv[3], &buf)) return 11;
  const int32_t* hdr0 = reinterpret_cast<const int32_t*>(buf.data());

  flame::Params params;  // defaults = cfg/flame_offline_tum.yaml
  params.nltgv2_iterations = hdr0[2];
  params.hip_device = hdr0[3];
  params.adaptive_data_weights = (hdr0[4] & 1) != 0;
  params.rescale_data = (hdr0[4] & 2) != 0;
  params.rparams.data_factor = 0.15f;
  params.rparams.step_x = 0.001f;
  params.rparams.step_q = 125.0f;
  params.rparams.theta = 0.25f;
  params.debug_draw_normals = true;  // (reference default false, cfg/flame_offline_tum.yaml:62)
  params.debug_flip_images = (hdr0[4] & 8) != 0;  // debug/flip_images (yaml :65)
  flame::Matrix3f K, Kinv;  // cfg/kinect.yaml: 525/525/319.5/239.5
  K(0, 0) = 525.f; K(0, 1) = 0.f; K(0, 2) = 319.5f; K(1, 0) = 0.f; K(1, 1) = 525.f; K(1, 2) = 239.5f;
  K(2, 0) = 0.f; K(2, 1) = 0.f; K(2, 2) = 1.f;
  Kinv(0, 0) = 1.f / 525.f; Kinv(0, 1) = 0.f; Kinv(0, 2) = -319.5f / 525.f;
  Kinv(1, 0) = 0.f; Kinv(1, 1) = 1.f / 525.f; Kinv(1, 2) = -239.5f / 525.f;
  Kinv(2, 0) = 0.f; Kinv(2, 1) = 0.f; Kinv(2, 2) = 1.f;
  std::shared_ptr<flame::Flame> sensor = std::make_shared<flame::Flame>(640, 480, K, Kinv, params);
  const bool first = sensor->stats().stats("fps_max") <= 0.0f;  // missing key reads as <= 0
  if (!first) return 12;

  int32_t V = 0, T = 0;
  for (int a = 3; a < argc; ++a) {
    if (!read_all(argv[a], &buf)) return 11;
    const int32_t* hdr = reinterpret_cast<const int32_t*>(buf.data());
    V = hdr[0]; T = hdr[1];
    const float* pos = reinterpret_cast<const float*>(hdr + 5);
    const float* mu = pos + 2 * V;
    const float* varp = (hdr[4] & 4) ? mu + V : nullptr;
    const int32_t* tri = reinterpret_cast<const int32_t*>(mu + V + (varp ? V : 0));
    std::vector<flame::Point2f> vtx(V);
    std::vector<float> idepth(V), var(V, 1e-4f);
    std::vector<flame::Triangle> tris(T);
    for (int v = 0; v < V; ++v) {
      vtx[v] = flame::Point2f(pos[2 * v], pos[2 * v + 1]);
      idepth[v] = mu[v];
      if (varp) var[v] = varp[v];
    }
    for (int t = 0; t < T; ++t) tris[t] = flame::Triangle(tri[3 * t], tri[3 * t + 1], tri[3 * t + 2]);
    bool ok = sensor->updateGraph(0.033 * a, a, vtx, idepth, var, tris);
    std::printf("update=%d hip_error=%d update_ms=%.3f sync_graph_ms=%.3f nltgv2_ms=%.3f\n", ok ? 1 : 0,
                static_cast<int>(sensor->stats().stats("hip_error")), sensor->stats().timings("update"),
                sensor->stats().timings("sync_graph"), sensor->stats().timings("nltgv2"));
    if (!ok) return 3;
  }

  std::vector<flame::Point2f> ovtx;
  std::vector<float> oid;
  std::vector<flame::Vector3f> normals;
  std::vector<flame::Triangle> otris;
  std::vector<bool> validity;
  std::vector<flame::Edge> edges;
  sensor->getInverseDepthMesh(&ovtx, &oid, &normals, &otris, &validity, &edges);
  std::vector<float> rmu, rvar;
  sensor->getRawIDepths(&ovtx, &rmu, &rvar);
  if (static_cast<int>(oid.size()) != V || static_cast<int>(validity.size()) != T) return 13;
  const auto& st = sensor->stats().stats();
  if (st.find("nltgv2_total_smoothness_cost") == st.end() || st.find("num_edges") == st.end()) return 14;

  // rows f1 / f2 through the facade (reference src/flame_offline_tum.cc:636-661)
  std::vector<float> idm, dm, cloud, pts;
  std::vector<int32_t> faces;
  if (!sensor->getFilteredInverseDepthMap(&idm) || !sensor->getDepthMapAndCloud(&dm, &cloud, 0.1f, 100.f) ||
      !sensor->getMeshPointNormalUV(&pts, &faces))
    return 17;
  if (idm.size() != 640u * 480u || cloud.size() != 3u * idm.size() || pts.size() != 12u * V) return 18;
  size_t nvalid = 0;
  for (size_t k = 0; k < validity.size(); ++k) nvalid += validity[k] ? 1 : 0;
  if (faces.size() != 3 * nvalid) return 19;

  FILE* f = std::fopen(argv[1], "wb");
  if (!f) return 15;
  std::fwrite(oid.data(), 4, V, f);
  for (int v = 0; v < V; ++v) { float n[3] = {normals[v](0), normals[v](1), normals[v](2)}; std::fwrite(n, 4, 3, f); }
  for (int t = 0; t < T; ++t) { unsigned char b = validity[t] ? 1 : 0; std::fwrite(&b, 1, 1, f); }
  std::fclose(f);
  f = std::fopen(argv[2], "w");
  if (!f) return 16;
  std::fprintf(f, "%d %.17g %.17g %.17g %.17g %.9g\n", static_cast<int>(edges.size()),
               sensor->stats().stats("nltgv2_total_smoothness_cost"),
               sensor->stats().stats("nltgv2_total_data_cost"),
               sensor->stats().stats("nltgv2_avg_smoothness_cost"),
               sensor->stats().timings("update"), sensor->stats().stats("coverage"));
  std::fclose(f);
  // the debug images (reference src/flame_offline_tum.cc:731-766), rendered on demand: wireframe,
  // features, normals, idepthmap as 4 x H x W x 3 bytes behind the binary outputs
  f = std::fopen((std::string(argv[1]) + ".img").c_str(), "wb");
  if (!f) return 15;
  const flame::Image3b* imgs[4] = {&sensor->getDebugImageWireframe(), &sensor->getDebugImageFeatures(),
                                   &sensor->getDebugImageNormals(), &sensor->getDebugImageInverseDepthMap()};
  for (int k = 0; k < 4; ++k)
    for (int i = 0; i < 480; ++i)
      for (int j = 0; j < 640; ++j) {
        const flame::Vec3b c = (*imgs[k])(i, j);
        unsigned char b[3] = {c[0], c[1], c[2]};
        std::fwrite(b, 1, 3, f);
      }
  std::fclose(f);
  if (sensor->getDebugImageDetections().rows != 480 || sensor->getDebugImageMatches().cols != 640) return 20;
  return 0;
}
