// tests/cpp/delaunay_test.cc -- reads "n\n x y\n..." from stdin, prints the triangles of
// flame::utils::DelaunayTriangulator (argv[1] = threads) (one "a b c" per line) or "FAIL".  Built with g++ by tests/test_delaunay.py.
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "flame/utils/delaunay.h"

int main(int argc, char** argv) {
  const int threads = argc > 1 ? std::atoi(argv[1]) : 1;
  int n = 0;
  if (std::scanf("%d", &n) != 1 || n < 0) return 2;
  std::vector<flame::Point2f> pts(static_cast<size_t>(n));
  for (int i = 0; i < n; ++i)
    if (std::scanf("%f %f", &pts[i].x, &pts[i].y) != 2) return 2;
  std::vector<flame::Triangle> tris;
  flame::utils::DelaunayTriangulator dt;
  if (!dt.triangulate(pts, &tris, threads)) { std::puts("FAIL"); return 0; }
  for (const flame::Triangle& t : tris) std::printf("%d %d %d\n", t[0], t[1], t[2]);
  return 0;
}
