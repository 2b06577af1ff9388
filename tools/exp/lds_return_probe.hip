// r06 probe: what does ONE wave pay per ds_read_* it issues and waits for -- and does a wave with half of its lanes off pay
// half?  (tools/exp/iter_prof.py: the nine ds_read_b128 of a phase P land after ~400 cycles even with ONE active wave: the
// segment is not bound by the LDS array -- 4 cycles per b128 by the guide -- but by something per wave.)
//   hipcc --offload-arch=gfx950 -O3 tools/exp/lds_return_probe.hip -o tools/exp/lds_return_probe && tools/exp/lds_return_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float f4v __attribute__((ext_vector_type(4)));
typedef float f3v __attribute__((ext_vector_type(3)));
typedef float f2v __attribute__((ext_vector_type(2)));

template <int MODE, int reads>  // 0 b128, 1 b96, 2 b64, 3 b32, 4 b64 + b32 (the split 12-byte slot), 5 b128 without the keep-alive of the 4th word
__global__ void k_probe(long long* out, int active_lanes, int reps) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < 16384; i += blockDim.x) reinterpret_cast<float*>(lds)[i] = (float)i;
  __syncthreads();
  // row per lane, odd pitch (13 slots of 16 bytes): a column read is conflict-free, like the incidence slots
  const char* row = lds + ((wave * 64 + lane) % 64) * 13 * 16 + wave * 16;
  float acc = 0.f;
  long long t0 = 0, t1 = 0;
  if (lane < active_lanes) {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0) : : "memory");
    for (int r = 0; r < reps; ++r) {
      if (MODE == 0 || MODE == 5) {
        f4v v[12];
#pragma unroll
        for (int u = 0; u < 12; ++u) if (u < reads) v[u] = *reinterpret_cast<const f4v*>(row + 16 * u);
#pragma unroll
        for (int u = 0; u < 12; ++u) if (u < reads) { if (MODE == 0) asm volatile("" ::"v"(v[u].w)); acc += v[u].x + v[u].y + v[u].z; }
      } else if (MODE == 1) {
        f3v v[12];
#pragma unroll
        for (int u = 0; u < 12; ++u) if (u < reads) v[u] = *reinterpret_cast<const f3v*>(row + 16 * u);
#pragma unroll
        for (int u = 0; u < 12; ++u) if (u < reads) acc += v[u].x + v[u].y + v[u].z;
      } else if (MODE == 2) {
        f2v v[12];
#pragma unroll
        for (int u = 0; u < 12; ++u) if (u < reads) v[u] = *reinterpret_cast<const f2v*>(row + 16 * u);
#pragma unroll
        for (int u = 0; u < 12; ++u) if (u < reads) acc += v[u].x + v[u].y;
      } else if (MODE == 3) {
        float v[12];
#pragma unroll
        for (int u = 0; u < 12; ++u) if (u < reads) v[u] = *reinterpret_cast<const float*>(row + 16 * u);
#pragma unroll
        for (int u = 0; u < 12; ++u) if (u < reads) acc += v[u];
      } else {
        f2v v[12]; float c[12];
#pragma unroll
        for (int u = 0; u < 12; ++u) if (u < reads) { v[u] = *reinterpret_cast<const f2v*>(row + 16 * u); c[u] = *reinterpret_cast<const float*>(row + 16 * u + 8); }
#pragma unroll
        for (int u = 0; u < 12; ++u) if (u < reads) acc += v[u].x + v[u].y + c[u];
      }
      asm volatile("" : "+v"(acc));
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1) : "v"(acc) : "memory");
  }
  if (lane == 0) { out[2 * wave] = t1 - t0; out[2 * wave + 1] = (long long)acc; }
}

template <int MODE, int READS>
double run(int threads, int active, int reps, long long* d) {
  hipFuncSetAttribute(reinterpret_cast<const void*>(k_probe<MODE, READS>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  double best = 1e30;
  for (int t = 0; t < 3; ++t) {  // (the first run warms the instruction cache)
    hipLaunchKernelGGL((k_probe<MODE, READS>), dim3(1), dim3(threads), 65536, 0, d, active, reps);
    (void)hipDeviceSynchronize();
    std::vector<long long> h(32);
    (void)hipMemcpy(h.data(), d, sizeof(long long) * 32, hipMemcpyDeviceToHost);
    long long mx = 0;
    for (int w = 0; w < threads / 64; ++w) mx = std::max(mx, h[2 * w]);
    best = std::min(best, (double)mx / reps);
  }
  return best;
}

template <int MODE>
void sweep(const char* name, long long* d) {
  const int reps = 200;
  for (int threads : {64, 256, 512})
    for (int active : {64, 32}) {
      std::printf("%-18s waves %d active lanes %2d: reads 1 / 3 / 6 / 9 / 12: %6.1f %6.1f %6.1f %6.1f %6.1f cycles per batch\n", name, threads / 64, active,
                  run<MODE, 1>(threads, active, reps, d), run<MODE, 3>(threads, active, reps, d), run<MODE, 6>(threads, active, reps, d),
                  run<MODE, 9>(threads, active, reps, d), run<MODE, 12>(threads, active, reps, d));
    }
}

int main() {
  long long* d;
  (void)hipMalloc(&d, 32 * sizeof(long long));
  sweep<0>("ds_read_b128 (kept)", d);
  sweep<5>("b128 -> b96 by use", d);
  sweep<1>("ds_read_b96", d);
  sweep<2>("ds_read_b64", d);
  sweep<3>("ds_read_b32", d);
  sweep<4>("ds_read_b64+b32", d);
  return 0;
}
