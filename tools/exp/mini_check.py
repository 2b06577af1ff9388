import numpy as np, sys, os
sys.path.insert(0,'/root/repo')
from flame_ros_amd import graphgen
from flame_ros_amd.regularizer import GraphRegularizer, default_params, default_sync_params
from oracle import COracle
from oracle.cbind import default_params as oparams
p, sp = default_params(), default_sync_params()
r = GraphRegularizer.empty(device=0, tile_single_max=1, stream_depth=5)
bad = 0
for k in range(12):
    name = ["tum","g14","g13","v2000","g18","g16"][k % 6]
    g = graphgen.named(name, seed=k)[0]
    var = np.full(g.V, 1e-4, np.float32)
    scale = r.sync_features(g.pos, g.z, var, g.tris, sp)
    r.step(p, 30)
    x = r.download()[0]
    o = COracle(g.pos, g.edges, g.alpha, g.beta, g.z, g.wgt)
    o.solve(oparams(), 30)
    ok = np.array_equal(np.asarray(x,np.float32).view(np.uint32), np.asarray(o.x,np.float32).view(np.uint32))
    print(name, g.V, g.E, "tiles", r.info("num_tiles"), "reused", r.info("plan_reused"), "exact", ok)
    bad += not ok
print("bad", bad)
