// tools/exp/vc_microbench.hip -- EXPERIMENT (not product code): cost of one PD iteration in a
// "vertex-centric" tile kernel that keeps every incidence of a vertex in registers and recomputes
// the dual of each edge on both endpoints, so that an iteration is: gather x_bar of the neighbours
// from LDS, dual ascent + contributions + primal chain in registers, one bar write, ONE barrier
// (no per-incidence slots, no scatter).  Synthetic tiles with the statistics of the 50 k graph
// (573 local vertices, degree 3..14, mean 6).  Prints cycles per iteration.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/exp/vc_microbench.hip -o /tmp/vcmb && /tmp/vcmb
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

constexpr int MAXD = 12;
struct Tile { int n_ext; };

__device__ __forceinline__ float proj_unit(float v) { return __builtin_amdgcn_fmed3f(v, -1.0f, 1.0f); }

template <int NT>
__global__ __launch_bounds__(NT) void k_vc(int n_ext, const int* __restrict__ nbr_g, const float4* __restrict__ ew_g,
                                          const int* __restrict__ deg_g, const unsigned* __restrict__ role_g,
                                          float4* state, int iters, long long* cycles) {
  extern __shared__ float4 bar[];  // 2 * n_ext
  const int tid = threadIdx.x, tile = blockIdx.x;
  const int lv = tid;
  const bool act = lv < n_ext;
  int nbr[MAXD];
  float al[MAXD], be[MAXD], dx[MAXD], dy[MAXD], q1[MAXD], q2[MAXD], q3[MAXD];
  const int deg = act ? deg_g[tile * NT + lv] : 0;
  const unsigned role = act ? role_g[tile * NT + lv] : 0u;
#pragma unroll
  for (int j = 0; j < MAXD; ++j) {
    const size_t o = ((size_t)tile * MAXD + j) * NT + lv;
    nbr[j] = act ? nbr_g[o] : 0;
    const float4 w = act ? ew_g[o] : make_float4(0, 0, 0, 0);
    al[j] = w.x; be[j] = w.y; dx[j] = w.z; dy[j] = w.w;
    q1[j] = q2[j] = q3[j] = 0.f;
  }
  float4 A = act ? state[(size_t)tile * NT + lv] : make_float4(0, 0, 0, 0);
  float x = A.x, w1 = A.y, w2 = A.z;
  const float z = A.w;
  float xb = x, w1b = w1, w2b = w2;
  if (act) bar[lv] = make_float4(xb, w1b, w2b, 0.f);
  int wdeg = deg;
  for (int off = 32; off > 0; off >>= 1) wdeg = max(wdeg, __shfl_xor(wdeg, off, 64));
  wdeg = __builtin_amdgcn_readfirstlane(wdeg);
  __syncthreads();
  const float sigma = 125.f, ntau = -1e-3f, theta = 0.25f, tl = 1e-3f * 0.15f;
  const long long t0 = __builtin_readcyclecounter();
  int cur = 0;
  for (int it = 0; it < iters; ++it) {
    const float4* b = bar + cur * n_ext;
    const float xp = x, w1p = w1, w2p = w2;
#pragma unroll
    for (int j0 = 0; j0 < MAXD; j0 += 6) {
      if (j0 >= wdeg) break;
      float4 nb[6];
#pragma unroll
      for (int u = 0; u < 6; ++u) nb[u] = b[nbr[j0 + u]];
#pragma unroll
      for (int u = 0; u < 6; ++u) asm volatile("" ::"v"(nb[u].w));
#pragma unroll
      for (int u = 0; u < 6; ++u) {
        const int j = j0 + u;
        const bool tgt = (role >> j) & 1u;  // this vertex is the edge's target
        const float s = tgt ? -1.f : 1.f;
        const float Sy = tgt ? nb[u].y : w1b, Sz = tgt ? nb[u].z : w2b;
        float t = s * (xb - nb[u].x);
        t = fmaf(-Sy, dx[j], t);
        t = fmaf(-Sz, dy[j], t);
        const float K1 = al[j] * t;
        const float K2 = be[j] * (s * (w1b - nb[u].y));
        const float K3 = be[j] * (s * (w2b - nb[u].z));
        q1[j] = proj_unit(fmaf(sigma, K1, q1[j]));
        q2[j] = proj_unit(fmaf(sigma, K2, q2[j]));
        q3[j] = proj_unit(fmaf(sigma, K3, q3[j]));
        const float aq = al[j] * q1[j], b2 = be[j] * q2[j], b3 = be[j] * q3[j];
        const float cx = s * aq;
        const float c1 = tgt ? -b2 : fmaf(-dx[j], aq, b2);
        const float c2 = tgt ? -b3 : fmaf(-dy[j], aq, b3);
        const bool on = j < deg;
        x = on ? fmaf(ntau, cx, x) : x;
        w1 = on ? fmaf(ntau, c1, w1) : w1;
        w2 = on ? fmaf(ntau, c2, w2) : w2;
      }
    }
    const float r = x - z;
    float xn = (r > tl) ? (x - tl) : ((r < -tl) ? (x + tl) : z);
    x = fminf(fmaxf(xn, 0.f), 10.f);
    xb = fmaf(theta, x - xp, x);
    w1b = fmaf(theta, w1 - w1p, w1);
    w2b = fmaf(theta, w2 - w2p, w2);
    cur ^= 1;
    if (act) {
      typedef float f3v __attribute__((ext_vector_type(3)));
      f3v v = {xb, w1b, w2b};
      *reinterpret_cast<f3v*>(&bar[cur * n_ext + lv]) = v;
    }
    __syncthreads();
  }
  const long long t1 = __builtin_readcyclecounter();
  if (act) state[(size_t)tile * NT + lv] = make_float4(x, w1, w2, q1[0] + q2[1] + q3[2]);
  if (tid == 0) cycles[tile] = t1 - t0;
}

int main(int argc, char** argv) {
  const int NT = 640, ntiles = 256, n_ext = 573;
  const int sorted = argc > 1 ? atoi(argv[1]) : 0;
  std::vector<int> nbr((size_t)ntiles * MAXD * NT), deg((size_t)ntiles * NT);
  std::vector<unsigned> role((size_t)ntiles * NT);
  std::vector<float4> ew((size_t)ntiles * MAXD * NT), st((size_t)ntiles * NT);
  srand(1);
  for (int t = 0; t < ntiles; ++t) {
    std::vector<int> d(n_ext);
    for (int v = 0; v < n_ext; ++v) {  // degree distribution ~ Delaunay: mean 6, 3..12
      int s = 0; for (int k = 0; k < 6; ++k) s += rand() % 3;  // 0..12 mean 6
      d[v] = std::min(MAXD, std::max(3, s));
    }
    if (sorted) std::sort(d.begin(), d.end(), std::greater<int>());
    for (int v = 0; v < n_ext; ++v) {
      deg[(size_t)t * NT + v] = d[v];
      role[(size_t)t * NT + v] = rand();
      st[(size_t)t * NT + v] = make_float4(0.5f + 0.001f * (rand() % 100), 0, 0, 0.5f);
      for (int j = 0; j < MAXD; ++j) {
        // neighbours: mostly near in local order (spatial locality), some far
        int u = (rand() % 4) ? std::min(n_ext - 1, std::max(0, v + (rand() % 61) - 30)) : rand() % n_ext;
        nbr[((size_t)t * MAXD + j) * NT + v] = u;
        ew[((size_t)t * MAXD + j) * NT + v] = make_float4(0.1f, 0.1f, (rand() % 20) - 10.f, (rand() % 20) - 10.f);
      }
    }
  }
  int *d_nbr, *d_deg; unsigned* d_role; float4 *d_ew, *d_st; long long* d_cyc;
  hipMalloc(&d_nbr, nbr.size() * 4); hipMalloc(&d_deg, deg.size() * 4); hipMalloc(&d_role, role.size() * 4);
  hipMalloc(&d_ew, ew.size() * 16); hipMalloc(&d_st, st.size() * 16); hipMalloc(&d_cyc, ntiles * 8);
  hipMemcpy(d_nbr, nbr.data(), nbr.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(d_deg, deg.data(), deg.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(d_role, role.data(), role.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(d_ew, ew.data(), ew.size() * 16, hipMemcpyHostToDevice);
  hipMemcpy(d_st, st.data(), st.size() * 16, hipMemcpyHostToDevice);
  const size_t lds = 2 * n_ext * sizeof(float4);
  for (int iters : {4, 64, 256}) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k_vc<NT>, dim3(ntiles), dim3(NT), lds, 0, n_ext, d_nbr, d_ew, d_deg, d_role, d_st, iters, d_cyc);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    const int reps = 20;
    for (int r = 0; r < reps; ++r)
      hipLaunchKernelGGL(k_vc<NT>, dim3(ntiles), dim3(NT), lds, 0, n_ext, d_nbr, d_ew, d_deg, d_role, d_st, iters, d_cyc);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<long long> cyc(ntiles);
    hipMemcpy(cyc.data(), d_cyc, ntiles * 8, hipMemcpyDeviceToHost);
    std::sort(cyc.begin(), cyc.end());
    printf("sorted=%d iters=%3d: launch %.2f us, in-kernel iteration loop p50 %lld cycles = %.0f cycles/iteration, %.3f us/iteration (wall)\n",
           sorted, iters, ms * 1e3 / reps, cyc[ntiles / 2], (double)cyc[ntiles / 2] / iters, ms * 1e3 / reps / iters);
  }
  return 0;
}
