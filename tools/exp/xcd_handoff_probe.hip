// tools/exp/xcd_handoff_probe.hip -- what would a hand-off between PERSISTENT tiles cost if all tiles of a
// small graph sat on ONE XCD (one L2)?  The r03 pricing of persistent tiles used agent-scope
// release / acquire fences (L2 write-back + invalidate, ~1.7 us each) because tiles of one launch are spread
// over the 8 XCDs.  Inside one XCD the L2 is the point of coherence: payload stores are written through the
// CU's L1, the flag is a relaxed agent-scope atomic, the neighbours' payload is read with sc1 loads (L1 miss,
// L2 hit) -- no fences.  Block b runs on XCD b % 8 (the tile kernel's mapping relies on the same rule), so a
// grid of 8 n blocks in which only b % 8 == 0 works puts n workgroups on XCD 0.
//
// Every workgroup: `rounds` x { spin `work` ticks (the iterations), publish 4 KB, raise its flag, wait for both
// ring neighbours' flags, read their 4 KB }.  Reported: median / max over tiles of (time per round - work), and
// how many payload words arrived stale (the all-XCD mode is expected to show some: it is NOT coherent).
// Every wait is bounded (2 ms): a protocol error ends the launch instead of hanging the GPU.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>

constexpr int kPay = 256;  // float4 per tile per round

__device__ __forceinline__ float4 load_l2(const float4* p) {
  float4 v;
  asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
  return v;
}

__global__ __launch_bounds__(512) void probe(int* flags, float4* pay, unsigned long long* out, int rounds, int work_ticks,
                                             int single_xcd, int ntiles, int exchange) {
  const int b = blockIdx.x;
  int tile;
  if (single_xcd) { if (b & 7) return; tile = b >> 3; } else tile = b;
  if (tile >= ntiles) return;
  const int tid = threadIdx.x;
  const int left = (tile + ntiles - 1) % ntiles, right = (tile + 1) % ntiles;
  int* err = flags + (ntiles + 1) * 32;
  __shared__ int s_abort;
  if (tid == 0) s_abort = 0;
  __syncthreads();
  float acc = 0.f;
  const unsigned long long t_begin = wall_clock64();
  for (int r = 1; r <= rounds; ++r) {
    const unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < (unsigned long long)work_ticks) __builtin_amdgcn_s_sleep(2);
    if (!exchange) continue;
    float4* mine = pay + ((size_t)(r & 1) * ntiles + tile) * kPay;
    if (tid < kPay) mine[tid] = make_float4((float)r, (float)tile, (float)tid, acc);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the stores are in L2
    __syncthreads();
    if (tid == 0) __hip_atomic_store(&flags[tile * 32], r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (tid < 2) {
      const int nb = tid ? right : left;
      const unsigned long long w0 = wall_clock64();
      while (__hip_atomic_load(&flags[nb * 32], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < r) {
        if (wall_clock64() - w0 > 200000ull) { s_abort = 1; atomicAdd(&err[1], 1); break; }  // 2 ms
        __builtin_amdgcn_s_sleep(1);
      }
    }
    __syncthreads();
    if (s_abort) break;
    const float4* theirs = pay + ((size_t)(r & 1) * ntiles + (tid < kPay ? left : right)) * kPay;
    const float4 v = load_l2(theirs + (tid & (kPay - 1)));
    if (v.x != (float)r) atomicAdd(&err[0], 1);
    acc += v.z;
  }
  if (tid == 0) out[tile] = wall_clock64() - t_begin;
  if (acc == -1.f) out[tile] = 0;
}

// block b -> XCC id (HW_REG_XCC_ID, hwreg 20, bits 3:0): the rule the single-XCD placement relies on
__global__ void xcc_of_block(int* out) {
  if (threadIdx.x == 0) out[blockIdx.x] = (int)(__builtin_amdgcn_s_getreg((31 << 11) | 20) & 15);
}

int main() {
  {
    int* d; int h[64];
    hipMalloc(&d, sizeof(h));
    hipLaunchKernelGGL(xcc_of_block, dim3(64), dim3(64), 0, 0, d);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    std::printf("XCC id of blocks 0..23:");
    for (int b = 0; b < 24; ++b) std::printf(" %d", h[b]);
    bool rr = true;
    for (int b = 8; b < 64; ++b) rr = rr && h[b] == h[b - 8];
    std::printf("   (block b and b + 8 on the same XCC for all 64 blocks: %s)\n", rr ? "yes" : "NO");
    hipFree(d);
  }
  const int rounds = 200, work = 300;  // 3 us of "iterations" per round
  const int max_tiles = 256;
  // memory kinds: ordinary device memory (cached in every XCD's L2), and the two kinds the runtime offers for
  // data shared while kernels run -- what tiles spread over ALL XCDs would need
  for (int kind = 0; kind < 3; ++kind) {
  int* flags; float4* pay; unsigned long long* out;
  const unsigned mflag = kind == 1 ? hipDeviceMallocUncached : hipDeviceMallocFinegrained;
  const size_t fb = sizeof(int) * 32 * (max_tiles + 2), pb = sizeof(float4) * 2 * max_tiles * kPay;
  if (kind == 0) { hipMalloc(&flags, fb); hipMalloc(&pay, pb); }
  else if (hipExtMallocWithFlags((void**)&flags, fb, mflag) != hipSuccess || hipExtMallocWithFlags((void**)&pay, pb, mflag) != hipSuccess) {
    std::printf("memory kind %d not available\n", kind);
    continue;
  }
  std::printf("---- flags and payload in %s ----\n", kind == 0 ? "ordinary device memory" : kind == 1 ? "uncached device memory" : "fine-grained device memory");
  hipMalloc(&out, sizeof(unsigned long long) * max_tiles);
  std::vector<unsigned long long> h(max_tiles);
  struct Cfg { const char* name; int single, ntiles, exchange; };
  const Cfg cfgs[] = {{"no exchange, 38 tiles on XCD 0", 1, 38, 0},       {"exchange, 38 tiles on XCD 0", 1, 38, 1},
                      {"exchange, 32 tiles on XCD 0", 1, 32, 1},          {"exchange, 16 tiles on XCD 0", 1, 16, 1},
                      {"exchange, 38 tiles over all XCDs", 0, 38, 1},     {"exchange, 128 tiles over all XCDs", 0, 128, 1},
                      {"exchange, 256 tiles over all XCDs", 0, 256, 1}};
  for (const Cfg& c : cfgs) {
    if (kind != 0 && c.single && c.ntiles != 38) continue;
    for (int rep = 0; rep < 2; ++rep) {
      hipMemset(flags, 0, sizeof(int) * 32 * (max_tiles + 2));
      hipMemset(out, 0, sizeof(unsigned long long) * max_tiles);
      hipLaunchKernelGGL(probe, dim3(c.single ? 8 * c.ntiles : c.ntiles), dim3(512), 0, 0, flags, pay, out, rounds, work, c.single,
                         c.ntiles, c.exchange);
      if (hipDeviceSynchronize() != hipSuccess) { std::printf("launch failed\n"); return 1; }
    }
    int herr[2];
    hipMemcpy(herr, flags + (c.ntiles + 1) * 32, sizeof(herr), hipMemcpyDeviceToHost);
    hipMemcpy(h.data(), out, sizeof(unsigned long long) * c.ntiles, hipMemcpyDeviceToHost);
    std::vector<double> per(c.ntiles);
    for (int t = 0; t < c.ntiles; ++t) per[t] = (double)h[t] / rounds * 0.01 - work * 0.01;  // us per round beyond the work
    std::sort(per.begin(), per.end());
    std::printf("%-36s per round beyond the %.1f us of work: median %.2f us, max %.2f us; stale payload words %d, timeouts %d\n", c.name,
                work * 0.01, per[c.ntiles / 2], per[c.ntiles - 1], herr[0], herr[1]);
  }
  hipFree(flags); hipFree(pay); hipFree(out);
  }
  return 0;
}
