"""r06: the partition mode's two transports on ONE GPU (every record travels from a part of rank 0 to a part of rank 0):
us per iteration, device time of an exchange, bits against one handle.
  python tools/exp/peer_transport_ab.py [name parts depth] ..."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402,F401

import bench  # noqa: E402
from flame_ros_amd import partition  # noqa: E402

cases = [("50k", 2, 16), ("50k", 2, 48), ("200k", 8, 16), ("200k", 8, 32)]
if len(sys.argv) > 3:
    cases = [(sys.argv[i], int(sys.argv[i + 1]), int(sys.argv[i + 2])) for i in range(1, len(sys.argv) - 2, 3)]
for name, parts, depth in cases:
    for transport in (0, 1, 0, 1):
        d = bench.library_partition(0, 1, 0, partition.unique_id(), torch.cuda.synchronize, lambda v: v, workload=name,
                                    parts_per_rank=parts, halo_depth=depth, steps=3, pipeline=0, transport=transport)
        print("%-5s %d parts depth %2d %-4s: %7.3f us/it, exchange %6.1f us x %d per step (share %.2f), resident %s, bit-exact %s" % (
            name, parts, depth, d["transport"], d["us_per_iteration"], d["exchange_us"], d["exchanges_per_step"], d["exchange_share"],
            d["resident_tiles"], d.get("bit_exact_vs_one_gpu")), flush=True)
