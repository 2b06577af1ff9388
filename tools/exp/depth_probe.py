"""dev experiment: does a deeper halo pay when it fits?  it/s of V-vertex graphs on 256 tiles at depth 4/5/6."""
import os, sys, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from flame_ros_amd import graphgen
from flame_ros_amd.regularizer import GraphRegularizer, default_params
p = default_params()
for V in (30000, 36000, 42000, 50000):
    g = graphgen.synthetic(V, seed=0)
    for depth in (4, 5, 6, 8):
        try:
            r = GraphRegularizer(g.pos, g.edges, g.alpha, g.beta, g.z, g.wgt, tile_depth=depth)
        except Exception as e:
            print(V, depth, "fail", str(e)[:60]); continue
        if r.info("path") != 2 or r.info("tile_depth") != depth:
            print(V, depth, "not built as asked: tiles", r.info("num_tiles"), "depth", r.info("tile_depth")); r.close(); continue
        for _ in range(3): r.step(p, 480, sync=True)
        t0 = time.perf_counter()
        for _ in range(20): r.step(p, 480, sync=False)
        r.sync()
        dt = time.perf_counter() - t0
        print("V %d depth %d tiles %d nt %d ept %d vpt %d lds %d KB  %.0f it/s  %.3f us/it" % (V, depth, r.info("num_tiles"), r.info("tile_threads"),
              r.info("tile_ept"), r.info("tile_vpt"), r.info("tile_lds_bytes") // 1024, 20 * 480 / dt, dt / (20 * 480) * 1e6))
        r.close()
