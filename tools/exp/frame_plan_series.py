"""dev (gpurun): the plan every frame of a 4-graph stream gets (tiles, threads x ept x vpt, depth, LDS, imbalance, resident or not)."""
import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from flame_ros_amd import graphgen
from flame_ros_amd.regularizer import GraphRegularizer, default_params, default_sync_params
name = sys.argv[1] if len(sys.argv) > 1 else "50k"
frames = [graphgen.named(name, seed=k) for k in range(4)]
iters = frames[0][1]
r = GraphRegularizer.empty(device=0)
p, sp = default_params(), default_sync_params()
keys = ("num_tiles", "tile_threads", "tile_ept", "tile_vpt", "tile_depth", "tile_lds_bytes", "tile_imbalance_pct", "plan_reused",
        "plan_on_device", "tile_ext_vertices", "tile_loc_edges", "persist_used")
for k in range(12):
    g = frames[k & 3][0]
    var = np.full(g.V, 1e-4, np.float32)
    r.sync_features(g.pos, g.z, var, g.tris, sp)
    r.step(p, iters, sync=True)
    print("frame %2d (graph %d, E %d):" % (k, k & 3, g.E), " ".join("%s=%d" % (q.replace("tile_", ""), r.info(q)) for q in keys))
