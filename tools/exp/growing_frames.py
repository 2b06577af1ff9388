"""r06 (the 11-27 ms solves of the fat-size frame soak): is the "device time" of a slow solve device time?
flame_hip_last_solve_ms brackets everything flame_hip_solve enqueues with two events -- poll-list kernel, mark memsets, the
resident launch -- so a HOST stall between two of those enqueues (a buffer that has to grow: hipFree + hipMalloc inside the
solve path) reads as device time.  One handle, frames whose size keeps reaching new maxima; per frame the host time of the
upload, of the first solve, its "device" time, and the device memory the process holds (growth = a reallocation happened)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

from flame_ros_amd import graphgen  # noqa: E402
from flame_ros_amd.regularizer import GraphRegularizer, default_params  # noqa: E402

p = default_params()
rng = np.random.default_rng(5)
sizes = []
v = 60000
while v < 235000:
    sizes += [int(v * f) for f in (1.0, 0.93, 0.97)]
    v = int(v * 1.07)
frames = {}
r = GraphRegularizer.empty(device=0)
rows = []
for i, V in enumerate(sizes):
    g = graphgen.synthetic(V, 1280, 1024, seed=100 + i)
    free0 = torch.cuda.mem_get_info()[0]
    t0 = time.perf_counter()
    r.reupload(g.pos, g.edges, g.alpha, g.beta, g.z, g.wgt, tris=g.tris)
    t1 = time.perf_counter()
    r.step(p, 40)
    t2 = time.perf_counter()
    dev1 = r.last_solve_ms()[0]
    free1 = torch.cuda.mem_get_info()[0]
    r.step(p, 40)
    dev2 = r.last_solve_ms()[0]
    free2 = torch.cuda.mem_get_info()[0]
    rows.append((V, (t1 - t0) * 1e3, (t2 - t1) * 1e3, dev1, dev2, (free0 - free1) / 1e6, (free1 - free2) / 1e6, r.info("tile_lds_bytes"), r.info("tile_depth"), r.info("tile_slot12"), r.info("persist_used")))
    print("frame %2d V %6d: upload %7.2f ms | first solve host %7.3f ms 'device' %7.3f ms | second solve 'device' %6.3f ms | device memory taken during upload+first solve %7.1f MB, during the second %6.1f MB | LDS %6d depth %d s12 %d resident %d" % ((i,) + rows[-1]), flush=True)
r.close()
