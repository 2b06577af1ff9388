"""r06 dev aid: where does ONE PD iteration of a resident tile go?  Builds flame_ros_amd/libflame_hip_itp.so from a PATCHED COPY
of csrc/kernels.hip (the product source carries no instrumentation): every wavefront of one chosen tile stamps s_memtime at
seven points of every iteration of one chosen round --
  0 top of the iteration | 1 phase D's gathers landed | 2 phase D done (dual ascent, slot stores drained) | 3 behind barrier 1 |
  4 phase P's slot sums done | 5 prox, extrapolation, bar[] store drained | 6 behind barrier 2
(each stamp drains lgkmcnt: the stamps perturb what they measure, +10 % or so; guide, "s_memtime").
Run here (hipcc cross-compiles); tools/exp/iter_prof.py reads the stamps on the GPU box."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
C = os.path.join(ROOT, "flame_ros_amd", "csrc")
s = open(os.path.join(C, "kernels.hip")).read()

def rep(a, b, cnt=1):
    global s
    assert s.count(a) == cnt, (a, s.count(a))
    s = s.replace(a, b)

rep("namespace flamehip {\nnamespace {\n", '''namespace flamehip {
namespace {
__device__ unsigned long long g_itp[16 * 8 * 8];  // [wave][iteration of the round][stamp]
__device__ int g_itp_cfg[2] = {-1, 2};             // tile, round
__device__ __forceinline__ void itp_stamp(int base, int n) {
  if (base >= 0) {
    unsigned long long t;
    asm volatile("s_waitcnt lgkmcnt(0)\\n\\ts_memtime %0\\n\\ts_waitcnt lgkmcnt(0)" : "=s"(t) : : "memory");
    if ((threadIdx.x & 63) == 0) g_itp[base + n] = t;
  }
}
''')
# phase D: the stamp behind the gathers needs the wave's base index
rep("                                             f2v (&q23)[EPT], float sigma) {\n  float4 bi[K], bj[K];",
    "                                             f2v (&q23)[EPT], float sigma, int itp = -1) {\n  float4 bi[K], bj[K];")
rep("  for (int k = 0; k < K; ++k) { keep_w(bi[k]); keep_w(bj[k]); }\n  const f2v sg = {sigma, sigma};",
    "  for (int k = 0; k < K; ++k) { keep_w(bi[k]); keep_w(bj[k]); }\n  itp_stamp(itp, 1);\n  const f2v sg = {sigma, sigma};")
rep("                                             f2v (&q23)[EPT], float sigma) {\n    if (n == K) tile_phase_d<O, K, EPT, S12, MARK>(bar, sm, eij, es, ed, ew, q1, q23, sigma);\n    else PhaseD<O, K - 1, EPT, S12, MARK>::run(n, bar, sm, eij, es, ed, ew, q1, q23, sigma);",
    "                                             f2v (&q23)[EPT], float sigma, int itp = -1) {\n    if (n == K) tile_phase_d<O, K, EPT, S12, MARK>(bar, sm, eij, es, ed, ew, q1, q23, sigma, itp);\n    else PhaseD<O, K - 1, EPT, S12, MARK>::run(n, bar, sm, eij, es, ed, ew, q1, q23, sigma, itp);")
rep("                                             const float4 (&)[EPT], float (&)[EPT], f2v (&)[EPT],\n                                             float) {}",
    "                                             const float4 (&)[EPT], float (&)[EPT], f2v (&)[EPT],\n                                             float, int = -1) {}")
rep("  int done = 0, round = 0;\n", "  int done = 0, round = 0;\n  const bool itp_tile = PERSIST && tile_id == g_itp_cfg[0];\n  const int itp_round = g_itp_cfg[1];\n")
rep("    const int ri = min(iters - it, depth);\n", "    const int ri = min(iters - it, depth);\n    const int itp = (itp_tile && round == itp_round && it <= 8) ? ((tid >> 6) * 8 + (it - 1)) * 8 : -1;\n    itp_stamp(itp, 0);\n")
rep("    PhaseD<0, EPT, EPT, S12, MARK>::run(nk, bar, sm, eij, es, ed, ew, q1, q23, sigma);\n", "    PhaseD<0, EPT, EPT, S12, MARK>::run(nk, bar, sm, eij, es, ed, ew, q1, q23, sigma, itp);\n    itp_stamp(itp, 2);\n")
rep("    __syncthreads();\n    if (prof && tid == 0 && it <= kMaxDepth) prof[2 * it] = __builtin_readcyclecounter();\n",
    "    __syncthreads();\n    itp_stamp(itp, 3);\n")
# phase P: a stamp behind the landed slot reads (7), in front of the ordered chain
rep("__device__ __forceinline__ void sum_slots(const float4* row, f2v nt2, float ntau, f2v& w, float& x) {",
    "__device__ __forceinline__ void sum_slots(const float4* row, f2v nt2, float ntau, f2v& w, float& x, int itp = -1) {")
rep("  for (int u = 0; u < K; ++u) keep_w(t[u]);\n#pragma unroll\n  for (int u = 0; u < K; ++u) {\n    const f2v c = {t[u].x, t[u].y};",
    "  for (int u = 0; u < K; ++u) keep_w(t[u]);\n  itp_stamp(itp, 7);\n#pragma unroll\n  for (int u = 0; u < K; ++u) {\n    const f2v c = {t[u].x, t[u].y};")
rep("  static __device__ __forceinline__ void run(int n, const float4* row, f2v nt2, float ntau, f2v& w, float& x) {\n    if constexpr (LO == HI) {\n      if constexpr (LO > 0) sum_slots<LO, MARK>(row, nt2, ntau, w, x);",
    "  static __device__ __forceinline__ void run(int n, const float4* row, f2v nt2, float ntau, f2v& w, float& x, int itp = -1) {\n    if constexpr (LO == HI) {\n      if constexpr (LO > 0) sum_slots<LO, MARK>(row, nt2, ntau, w, x, itp);")
rep("      if (n <= MID) SlotSel<LO, MID, MARK>::run(n, row, nt2, ntau, w, x);\n      else SlotSel<MID + 1, HI, MARK>::run(n, row, nt2, ntau, w, x);",
    "      if (n <= MID) SlotSel<LO, MID, MARK>::run(n, row, nt2, ntau, w, x, itp);\n      else SlotSel<MID + 1, HI, MARK>::run(n, row, nt2, ntau, w, x, itp);")
rep("          SlotSel<0, kPRound, MARK>::run(j, row, nt2, ntau, w, x);", "          SlotSel<0, kPRound, MARK>::run(j, row, nt2, ntau, w, x, k == 0 ? itp : -1);")
rep("        x = prox_l1(x, vz[k], vt[k], x_min, x_max);\n", "        asm volatile(\"\" : \"+v\"(x), \"+v\"(w));\n        if (k == 0) itp_stamp(itp, 4);\n        x = prox_l1(x, vz[k], vt[k], x_min, x_max);\n")
rep("        if (lv < n_upd) lds_store3(&bar[lv], vwb[k].x, vwb[k].y, vxb[k]);\n", "        if (lv < n_upd) lds_store3(&bar[lv], vwb[k].x, vwb[k].y, vxb[k]);\n        if (k == 0) itp_stamp(itp, 5);\n")
rep("    __syncthreads();\n    if (prof && tid == 0 && it <= kMaxDepth) prof[2 * it + 1] = __builtin_readcyclecounter();\n",
    "    __syncthreads();\n    itp_stamp(itp, 6);\n")
s += '''
// (variant build only) set the stamped tile / round, or read the stamps back
extern "C" int flame_hip_exp_itp(int tile, int round, unsigned long long* out, int read) {
  if (read) return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(flamehip::g_itp), sizeof(unsigned long long) * 16 * 8 * 8);
  int cfg[2] = {tile, round};
  hipError_t e = hipMemset(nullptr, 0, 0);
  (void)e;
  unsigned long long z[16 * 8 * 8] = {};
  if (hipMemcpyToSymbol(HIP_SYMBOL(flamehip::g_itp), z, sizeof(z)) != hipSuccess) return -1;
  return (int)hipMemcpyToSymbol(HIP_SYMBOL(flamehip::g_itp_cfg), cfg, sizeof(cfg));
}
'''
out = os.path.join(C, "kernels_itp.hip")
open(out, "w").write(s)
subprocess.check_call([sys.executable, "-c", "from flame_ros_amd import build; build.build()"], cwd=ROOT)
sys.path.insert(0, ROOT)
from flame_ros_amd import build as _b  # noqa: E402
flags = _b.FLAGS + _b.FLAGS_FOR.get("kernels.hip", [])
subprocess.check_call(["/opt/rocm/bin/hipcc"] + flags + ["-c", out, "-o", os.path.join(C, "kernels_itp.o")])
objs = [os.path.join(C, o) for o in ("kernels_itp.o", "plan_dev.o", "delaunay_dev.o", "flame_hip.o", "plan.o", "sync.o", "part.o")]
lib = os.path.join(ROOT, "flame_ros_amd", "libflame_hip_itp.so")
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-pthread", "-o", lib] + objs + ["-L/opt/rocm/lib", "-lroctx64", "-ldl"])
os.remove(out)
print("built", lib)
