"""oracle/ -- CPU restatement of FLaME's NLTGV2-L1 graph regulariser (TEST INFRASTRUCTURE).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package.
PARITY UNPINNED: the reference tree (/root/reference = flame_ros) does not contain the solver
(robustrobotics/flame, un-vendored, un-pinned: reference README.md:73, CMakeLists.txt:57) nor any
test or golden vector for it; see nltgv2_oracle.h and DESIGN.md.

`COracle` wraps the float32 C restatement (nltgv2_oracle.c, the checker the HIP path is compared
with bit-for-bit); `nltgv2_np` is an independent float64 NumPy restatement used to pin the C one.
"""
from .cbind import COracle, OracleParams, TriParams, build_oracle, oracle_lib_path  # noqa: F401
