import os, sys
import numpy as np
sys.path.insert(0, os.getcwd())
from flame_ros_amd import graphgen
from flame_ros_amd.regularizer import GraphRegularizer, default_params, default_sync_params
sp, p = default_sync_params(), default_params()
lvl = int(sys.argv[1])
r = GraphRegularizer.empty(device=0, tile_single_max=640, stream_depth=5)
for k in range(8):
    g = graphgen.synthetic(1200, seed=100 + k)
    if k == 6: r.set_option("plan_timing", lvl)
    r.sync_features(g.pos, g.z, np.full(g.V, 1e-4, np.float32), g.tris, sp)
    r.step(p, 60)
