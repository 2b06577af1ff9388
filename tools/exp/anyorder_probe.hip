// tools/exp/anyorder_probe.hip -- does a launch without the AQL barrier bit (hipExtAnyOrderLaunch)
// overlap with its predecessor in the same stream on gfx950?  Each block stamps wall_clock64()
// (100 MHz constant clock) at start and end and spins `spin_ticks` in between.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <vector>
#include <algorithm>

__global__ void probe(unsigned long long* out, int k, int spin_ticks) {
  const unsigned long long t0 = wall_clock64();
  while (wall_clock64() - t0 < (unsigned long long)spin_ticks) __builtin_amdgcn_s_sleep(8);
  if (threadIdx.x == 0) {
    out[(size_t)(k * gridDim.x + blockIdx.x) * 2] = t0;
    out[(size_t)(k * gridDim.x + blockIdx.x) * 2 + 1] = wall_clock64();
  }
}

int main() {
  const int N = 12, G = 256, spin = 500;  // 5 us per kernel
  unsigned long long* d;
  hipMalloc(&d, sizeof(unsigned long long) * 2 * N * G);
  hipStream_t s, s2;
  hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
  hipStreamCreateWithFlags(&s2, hipStreamNonBlocking);
  std::vector<unsigned long long> h(2 * N * G);
  for (int mode = 0; mode < 4; ++mode) {
    for (int rep = 0; rep < 2; ++rep) {
      hipMemsetAsync(d, 0, sizeof(unsigned long long) * 2 * N * G, s);
      hipStreamSynchronize(s);
      for (int k = 0; k < N; ++k) {
        if (mode == 0) hipLaunchKernelGGL(probe, dim3(G), dim3(256), 0, s, d, k, spin);
        else if (mode == 1) hipExtLaunchKernelGGL(probe, dim3(G), dim3(256), 0, s, nullptr, nullptr, hipExtAnyOrderLaunch, d, k, spin);
        else if (mode == 2) hipLaunchKernelGGL(probe, dim3(G), dim3(256), 0, (k & 1) ? s2 : s, d, k, spin);
        else hipExtLaunchKernelGGL(probe, dim3(G), dim3(256), 0, s, nullptr, nullptr, (k % 3) ? hipExtAnyOrderLaunch : 0, d, k, spin);
      }
      hipStreamSynchronize(s); hipStreamSynchronize(s2);
    }
    hipMemcpy(h.data(), d, sizeof(unsigned long long) * h.size(), hipMemcpyDeviceToHost);
    unsigned long long base = ~0ull;
    for (int i = 0; i < N * G; ++i) base = std::min(base, h[2 * i]);
    printf("mode %d (%s): per kernel [first start, last start, first end, last end] in us\n", mode,
           mode == 0 ? "normal" : mode == 1 ? "anyorder" : mode == 2 ? "two streams" : "anyorder 2 of 3");
    for (int k = 0; k < N; ++k) {
      unsigned long long s0 = ~0ull, s1 = 0, e0 = ~0ull, e1 = 0;
      for (int b = 0; b < G; ++b) {
        s0 = std::min(s0, h[2 * (k * G + b)]); s1 = std::max(s1, h[2 * (k * G + b)]);
        e0 = std::min(e0, h[2 * (k * G + b) + 1]); e1 = std::max(e1, h[2 * (k * G + b) + 1]);
      }
      printf("  k%02d  %7.2f %7.2f %7.2f %7.2f\n", k, (s0 - base) / 100.0, (s1 - base) / 100.0, (e0 - base) / 100.0, (e1 - base) / 100.0);
    }
  }
  return 0;
}
