#!/bin/bash
# tools/exp/build_variant.sh <tag> [-DFLAG=V ...] -- dev aid: builds flame_ros_amd/libflame_hip_<tag>.so with kernels.hip
# compiled with the extra flags (the other objects are the default build's), for A/B runs in ONE gpurun call:
#   FLAME_HIP_LIB=flame_ros_amd/libflame_hip_<tag>.so python tools/exp/resident_ab.py
# HOOKS=1: link the host side of the hooks library (flame_hip_test_hook) -- the late-tile tests want both:
#   HOOKS=1 tools/exp/build_variant.sh stall -DFLAME_PERSIST_STALL_HOOK=1
#   FLAME_HIP_LIB=$PWD/flame_ros_amd/libflame_hip_stall.so FLAME_HIP_HOOKS_IN_LIB=1 python -m pytest tests/test_gpu_persist.py -m gpu -k really_late
set -e
cd "$(dirname "$0")/../.."
tag=$1; shift
python -c "from flame_ros_amd import build; build.build()"
C=flame_ros_amd/csrc
/opt/rocm/bin/hipcc $(python -c "from flame_ros_amd import build as b; print(' '.join(b.FLAGS + b.FLAGS_FOR.get('kernels.hip', [])))") "$@" -c $C/kernels.hip -o $C/kernels_$tag.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -pthread -o flame_ros_amd/libflame_hip_$tag.so $C/kernels_$tag.o $C/plan_dev.o $C/delaunay_dev.o $C/flame_hip${HOOKS:+_hooks}.o $C/plan.o $C/sync.o $C/part.o -L/opt/rocm/lib -lroctx64 -ldl
echo built flame_ros_amd/libflame_hip_$tag.so
