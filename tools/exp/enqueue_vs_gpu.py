"""dev helper (gpurun): host time to ENQUEUE the launches of one solve (first solve of a plan: direct launches)
against the device time of the same solve -- is a frame's solve bound by the host's launch rate?"""
import os, sys, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from flame_ros_amd import graphgen
from flame_ros_amd.regularizer import GraphRegularizer, default_params
p = default_params()
for name, opts in (("tum", dict(tile_single_max=1, tile_depth=5)), ("euroc", {}), ("50k", {})):
    g, iters = graphgen.named(name)
    enq, dev = [], []
    for rep in range(30):
        r = GraphRegularizer(g.pos, g.edges, g.alpha, g.beta, g.z, g.wgt, tris=g.tris, device=0, lane_order=0, **opts)
        r.sync()
        t0 = time.perf_counter()
        r.step(p, iters, sync=False)
        t1 = time.perf_counter()
        r.sync()
        enq.append((t1 - t0) * 1e3); dev.append(r.last_solve_ms()[0])
        n = r.last_solve_ms()[1]
        r.close()
    enq, dev = np.median(enq[10:]), np.median(dev[10:])
    print("%-6s launches %3d  host enqueue %.4f ms (%.2f us each)  device %.4f ms (%.2f us each)" % (name, n, enq, enq * 1e3 / n, dev, dev * 1e3 / n), flush=True)
