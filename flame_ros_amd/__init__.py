"""flame_ros_amd -- MI355X-native NLTGV2-L1 graph regulariser for FLaME (hot path only).

csrc/ holds the HIP kernels and the C ABI (include/flame_hip.h -> libflame_hip.so);
`regularizer` mirrors upstream's regulariser interface over that ABI; `graphgen` makes the
synthetic Delaunay graphs of BASELINE.json; `dist` shards frames / subdomains over GPUs.
"""
__all__ = ["lib", "regularizer", "graphgen"]
