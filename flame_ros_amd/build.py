"""Builds flame_ros_amd/libflame_hip.so in-tree with hipcc for gfx950 (MI355X).

hipcc cross-compiles without a GPU.  -ffp-contract=off is part of the arithmetic contract (only
explicit fmaf() is fused; see csrc/kernels.hip and oracle/nltgv2_oracle.h).
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libflame_hip.so")
SOURCES = ["kernels.hip", "plan_dev.hip", "delaunay_dev.hip", "flame_hip.cpp", "plan.cpp", "sync.cpp", "part.cpp"]
HEADERS = ["common.h", "kernels.h", "plan.h", "plan_dev.h", "delaunay_dev.h", "sync.h", os.path.join("..", "..", "include", "flame_hip.h")]
# -amdgpu-kernarg-preload-count: the first 16 dwords of a kernel's arguments arrive in SGPRs at wave
# launch (gfx950) instead of through a scalar load -- the tile kernel's argument order relies on it.
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-Wall",
         "-Wno-unused-result", "-mllvm", "-amdgpu-kernarg-preload-count=16"]
FLAGS_FOR = {}  # extra flags per source file (none today)


def hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    return "hipcc"


# The tests' fault-injection library: flame_hip.cpp compiled with -DFLAME_HIP_TEST_HOOKS=1 (flame_hip_test_hook():
# forced give-ups of resident launches, filled allocations), every other object the product's.  The product library has
# no such switch and reads no environment variable.
HOOKS_LIB = os.path.join(HERE, "libflame_hip_hooks.so")


def needs_build():
    if not os.path.exists(LIB) or not os.path.exists(HOOKS_LIB):
        return True
    t = min(os.path.getmtime(LIB), os.path.getmtime(HOOKS_LIB))
    deps = [os.path.join(CSRC, s) for s in SOURCES + HEADERS] + [os.path.abspath(__file__)]
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


def build(force=False, verbose=False):
    extra = os.environ.get("FLAME_EXTRA_HIPCC_FLAGS", "").split()  # dev aid (A/B of kernel variants)
    if not force and not needs_build() and not extra:
        return LIB
    objs = []
    for s in SOURCES:
        o = os.path.join(CSRC, os.path.splitext(s)[0] + ".o")
        cmd = [hipcc()] + FLAGS + FLAGS_FOR.get(s, []) + extra + ["-c", os.path.join(CSRC, s), "-o", o]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
        objs.append(o)
    cmd = [hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-pthread", "-o", LIB] + objs + ["-L/opt/rocm/lib", "-lroctx64", "-ldl"]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    ho = os.path.join(CSRC, "flame_hip_hooks.o")
    cmd = [hipcc()] + FLAGS + extra + ["-DFLAME_HIP_TEST_HOOKS=1", "-c", os.path.join(CSRC, "flame_hip.cpp"), "-o", ho]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    hobjs = [ho if o.endswith(os.sep + "flame_hip.o") else o for o in objs]
    cmd = [hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-pthread", "-o", HOOKS_LIB] + hobjs + ["-L/opt/rocm/lib", "-lroctx64", "-ldl"]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
