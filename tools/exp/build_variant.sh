#!/bin/bash
# tools/exp/build_variant.sh <tag> [-DFLAG=V ...] -- dev aid: builds flame_ros_amd/libflame_hip_<tag>.so with kernels.hip
# compiled with the extra flags (the other objects are the default build's), for A/B runs in ONE gpurun call:
#   FLAME_HIP_LIB=flame_ros_amd/libflame_hip_<tag>.so python tools/exp/resident_ab.py
set -e
cd "$(dirname "$0")/../.."
tag=$1; shift
python -c "from flame_ros_amd import build; build.build()"
C=flame_ros_amd/csrc
/opt/rocm/bin/hipcc $(python -c "from flame_ros_amd import build as b; print(' '.join(b.FLAGS + b.FLAGS_FOR.get('kernels.hip', [])))") "$@" -c $C/kernels.hip -o $C/kernels_$tag.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -pthread -o flame_ros_amd/libflame_hip_$tag.so $C/kernels_$tag.o $C/plan_dev.o $C/delaunay_dev.o $C/flame_hip.o $C/plan.o $C/sync.o $C/part.o -L/opt/rocm/lib -lroctx64 -ldl
echo built flame_ros_amd/libflame_hip_$tag.so
