import os, sys, time, numpy as np
sys.path.insert(0, '/root/repo')
from flame_ros_amd import graphgen
from flame_ros_amd.regularizer import GraphRegularizer, default_params, default_sync_params, default_tri_params
import sys as _s
name = _s.argv[1] if len(_s.argv) > 1 else "tum"
frames = [graphgen.named(name, seed=k) for k in range(4)]
r = GraphRegularizer.empty(device=0, tile_single_max=896, stream_depth=5)
p, sp = default_params(), default_sync_params()
Kinv = np.linalg.inv(np.array([[525.0, 0, 319.5], [0, 525.0, 239.5], [0, 0, 1]])).astype(np.float32)
for k in range(10):
    g = frames[k & 3][0]
    tp = default_tri_params(g.width, g.height)
    var = np.full(g.V, 1e-4, np.float32)
    t0 = time.perf_counter()
    scale = r.sync_features(g.pos, g.z, var, g.tris, sp)
    t1 = time.perf_counter()
    r.step(p, frames[0][1], sync=False)
    out = r.frame_results(p, Kinv, tp, scale_back=scale, with_edges=True, with_coverage=True)
    t2 = time.perf_counter()
    print("sync %.3f total %.3f  mini %d reused %d tiles %d depth %d imbalance %d%% threads %d ept %d" % ((t1-t0)*1e3, (t2-t0)*1e3, r.info("plan_mini"), r.info("plan_reused"), r.info("num_tiles"), r.info("tile_depth"), r.info("tile_imbalance_pct"), r.info("tile_threads"), r.info("tile_ept")))
