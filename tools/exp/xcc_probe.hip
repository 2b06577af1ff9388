// r06 probe: is "block b of a launch runs on XCD b % 8" still true when ANOTHER queue dispatches workgroups at the same time?
// The one-XCD mode of the resident kernel takes tiles only in the blocks with (b & 7) == 0 and relies on them
// sharing one L2.  This launches a grid of 8 x 32 blocks that record the hardware's XCC_ID, alone and beside a second stream
// of short 157-block launches, and counts the launches whose (b & 7) == 0 blocks did NOT all report one XCD.
//   hipcc --offload-arch=gfx950 -O3 tools/exp/xcc_probe.hip -o /tmp/xcc_probe && /tmp/xcc_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ void k_rec(int* out) {
  extern __shared__ char lds[];
  unsigned x;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
  if (threadIdx.x == 0) out[blockIdx.x] = (int)(x & 0xf);
  if ((blockIdx.x & 7) == 0) {  // the tile carriers stay a few microseconds, like a short solve
    long long t0 = clock64();
    while (clock64() - t0 < 20000) lds[threadIdx.x] = 1;
  }
}

__global__ void k_noise(float* p, int spin) {
  extern __shared__ char lds[];
  long long t0 = clock64();
  while (clock64() - t0 < spin) lds[threadIdx.x] = 2;
  if (threadIdx.x == 0) p[blockIdx.x] = 1.f;
}

int main() {
  int* d; float* n;
  const int G = 256;
  hipMalloc(&d, G * sizeof(int)); hipMalloc(&n, 4096 * sizeof(float));
  hipStream_t s0, s1; hipStreamCreate(&s0); hipStreamCreate(&s1);
  std::vector<int> h(G);
  for (int mode = 0; mode < 3; ++mode) {  // 0 alone, 1 beside 157-block launches, 2 beside 1024-block launches
    int bad = 0, spread_max = 0; const int reps = 400;
    int hist[9] = {0};
    for (int r = 0; r < reps; ++r) {
      if (mode) for (int q = 0; q < 6; ++q) k_noise<<<mode == 1 ? 157 : 1024, 512, 60 * 1024, s1>>>(n, mode == 1 ? 6000 : 3000);
      k_rec<<<G, 512, 44 * 1024, s0>>>(d);
      hipMemcpyAsync(h.data(), d, G * sizeof(int), hipMemcpyDeviceToHost, s0);
      hipStreamSynchronize(s0);
      int seen = 0;
      for (int b = 0; b < G; b += 8) seen |= 1 << h[b];
      const int k = __builtin_popcount(seen);
      ++hist[k > 8 ? 8 : k];
      if (k != 1) ++bad;
      if (k > spread_max) spread_max = k;
    }
    hipDeviceSynchronize();
    printf("mode %d: %d of %d launches had their (b&7)==0 blocks on more than one XCD (max %d XCDs); histogram of XCD counts:", mode, bad, reps, spread_max);
    for (int k = 1; k <= 8; ++k) printf(" %d:%d", k, hist[k]);
    printf("\n");
  }
  return 0;
}
