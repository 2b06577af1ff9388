"""r06 (the 11-27 ms solves): what does the FIRST resident launch of a kernel configuration cost on a
stream -- by whether the kernel uses scratch memory (the <1024,3,1> fat-tile kernels spill 60-116 bytes per lane)?
New handle (= new stream) per line; device ms (HIP events around the launch) of its first four solves."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from flame_ros_amd import graphgen  # noqa: E402
from flame_ros_amd.regularizer import GraphRegularizer, default_params  # noqa: E402

p = default_params()
graphs = {}
for V in (50000, 100000, 190000, 160000):
    graphs[V] = graphgen.synthetic(V, 1280 if V > 60000 else 640, 1024 if V > 60000 else 480, seed=4)
order = [(50000, {}), (100000, {}), (190000, {}), (190000, {}), (160000, {}), (100000, {}), (190000, dict(lds_bytes=172 * 1024)), (50000, {}), (190000, {})]
for rep in range(2):
    for V, opts in order:
        g = graphs[V]
        with GraphRegularizer(g.pos, g.edges, g.alpha, g.beta, g.z, g.wgt, device=0, **opts) as r:
            ms = []
            for _ in range(4):
                r.step(p, 60)
                ms.append(r.last_solve_ms()[0])
            print("pass %d V %6d: %d tiles depth %d threads %d ept %d slot12 %d fat %d LDS %6d B resident %d | first solves (device ms): %s" % (
                rep, V, r.info("num_tiles"), r.info("tile_depth"), r.info("tile_threads"), r.info("tile_ept"), r.info("tile_slot12"), r.info("tile_fat"),
                r.info("tile_lds_bytes"), r.info("persist_used"), " ".join("%.3f" % m for m in ms)), flush=True)
