#include <chrono>
#include <cstdio>
#include <random>
#include <vector>
#include "flame/utils/delaunay.h"
int main() {
  for (int n : {1200, 10000, 50000}) {
    std::mt19937 rng(1);
    std::uniform_real_distribution<float> ux(0.f, 640.f), uy(0.f, 480.f);
    std::vector<flame::Point2f> p(n);
    for (auto& q : p) q = flame::Point2f(ux(rng), uy(rng));
    flame::utils::DelaunayTriangulator dt;
    std::vector<flame::Triangle> t, t1;
    for (int th : {1, 2, 4, 8, 16, 32}) {
      double best = 1e9;
      for (int r = 0; r < 12; ++r) {
        auto t0 = std::chrono::steady_clock::now();
        dt.triangulate(p, &t, th);
        best = std::min(best, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
      }
      if (th == 1) t1 = t;
      bool same = t.size() == t1.size();
      std::printf("n %d threads %d tris %zu  %.3f ms same-count %d", n, th, t.size(), best, (int)same);
#ifdef FLAME_DELAUNAY_TIMING
      std::printf("  last: snap %.3f sort %.3f ranks %.3f cuts %.3f setup %.3f subtrees %.3f merges %.3f faces %.3f join %.3f", dt.t_ms_[0], dt.t_ms_[1],
                  dt.t_ms_[2], dt.t_ms_[3], dt.t_ms_[4], dt.t_ms_[5], dt.t_ms_[6], dt.t_ms_[7], dt.t_ms_[8]);
#endif
      std::printf("\n");
    }
  }
}
